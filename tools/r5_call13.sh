#!/bin/bash
# round 5, GPU call 13: wgrad_dma_kernel with scalar-base wave-loads (FAST) against the per-lane 64-bit form (CVVAE_WGRAD_FAST=0), and
# the number of workgroups the pixel range is cut into (CVVAE_WGRAD_WGS: 768 = three rounds per CU, 256 = one)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_grad3d.py -q -x -k "wgrad" 2>&1 | tail -2
for rep in 1 2; do
  for v in "1 768" "0 768" "1 256" "1 512" "1 1024"; do
    set -- $v
    CVVAE_WGRAD_FAST=$1 CVVAE_WGRAD_WGS=$2 timeout 300 python tools/train_step_bench.py --dtype bf16 --wgrad-only > gpurun_out/r5_call13_f$1_w$2_$rep.json 2> gpurun_out/r5_call13_f$1_w$2_$rep.err
    python -c "
import json
d=json.loads(open('gpurun_out/r5_call13_f$1_w$2_$rep.json').read().strip().splitlines()[-1])
print('fast=$1 wgs=$2 rep$rep', [w['ms'] for w in d['wgrad']])"
  done
done
timeout 900 python -m pytest tests/test_gpu_grad3d.py tests/test_gpu_round5.py -q -x 2>&1 | tail -2
