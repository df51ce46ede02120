#!/bin/bash
# round 6, GPU call 8: in-kernel timelines of the per-frame 128-channel conv (ResnetBlock conv2 at 17x512^2, residual + statistics):
# the four-wave instance (CFG 9: two workgroups per CU) and the 8-wave 16-row tile (CFG 8), with and without their stores
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
L=$O/r6_probe_c2d128_timelines.log
: > $L
for c in 9 8 0; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCVVAE_CONV_PROBE -DCFG=$c -Icvvae_amd/csrc tools/probes/conv_probe.hip -o /tmp/conv_probe_$c 2>/dev/null
  for env in "PROBE_RES=1" "PROBE_RES=1 PROBE_NOSTORE=1" ""; do
    echo "=== CFG $c $env" >> $L
    env $env timeout 60 /tmp/conv_probe_$c >> $L 2>&1
  done
done
cat $L
