#!/bin/bash
# round 6, GPU call 2: in-kernel timelines (s_memtime stamps, tools/probes/conv_probe.hip) of the 128 -> 128 3x3x3 layer at 17x512^2:
# bf16 two-frame tile (CFG 3), fast fp32 fp6 on the 256-pixel tile (CFG 12), on the planar 512-pixel tile (CFG 15), the same without
# prologue (CFG 16), and the 256-channel layer on the planar all-waves-in-N tile (CFG 17) beside its bf16 twin (CFG 2)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
: > $O/r6_probe_fast_fp32_timelines.log
for c in 3 12 15 16 2 17; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCVVAE_CONV_PROBE -DCFG=$c -Icvvae_amd/csrc tools/probes/conv_probe.hip -o /tmp/conv_probe_$c 2>/dev/null
  echo "=== CFG $c" >> $O/r6_probe_fast_fp32_timelines.log
  timeout 60 /tmp/conv_probe_$c >> $O/r6_probe_fast_fp32_timelines.log 2>&1
done
cat $O/r6_probe_fast_fp32_timelines.log
timeout 300 python tools/conv_bench.py c2d128res c2d256res c2d512res --dtype f32q6 --iters 5 --rounds 3 > $O/r6_ab_planar_perframe_select.log 2>&1
cat $O/r6_ab_planar_perframe_select.log
