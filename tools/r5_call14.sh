#!/bin/bash
# round 5, GPU call 14: wgrad_dma_kernel with the wave-loads dealt statically (row / chunk of a load known at compile time);
# FAST on / off, 768 vs 240 workgroups; scratch builds without the wave-load instruction (ab2) / without the request code (ab4)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_grad3d.py -q -x -k "wgrad" 2>&1 | tail -2
for rep in 1 2; do
  for v in "1 768 default" "0 768 default" "1 240 default" "1 768 ab2" "1 768 ab4"; do
    set -- $v
    lib=cvvae_amd/libcvvae_hip.so; [ $3 != default ] && lib=gpurun_in/libcvvae_hip_$3.so
    CVVAE_LIB=$GRAFT_REPO_ROOT/$lib CVVAE_WGRAD_FAST=$1 CVVAE_WGRAD_WGS=$2 timeout 300 python tools/train_step_bench.py --dtype bf16 --wgrad-only > gpurun_out/r5_call14_f$1_w$2_$3_$rep.json 2> gpurun_out/r5_call14_f$1_w$2_$3_$rep.err
    python -c "
import json
d=json.loads(open('gpurun_out/r5_call14_f$1_w$2_$3_$rep.json').read().strip().splitlines()[-1])
print('fast=$1 wgs=$2 $3 rep$rep', [w['ms'] for w in d['wgrad']])"
  done
done
timeout 900 python -m pytest tests/test_gpu_grad3d.py tests/test_gpu_round5.py -q -x 2>&1 | tail -2
