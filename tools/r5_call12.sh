#!/bin/bash
# round 5, GPU call 12: wgrad_dma_kernel with a co-fragment PAIR x four taps (+ tap 8) per wave (default) against one fragment pair x
# nine taps (old), and what each part of the panel loop costs -- scratch builds with parts left out
# (CVVAE_WGRAD_ABLATE: 1 no transpose reads, 2 no wave-load instruction, 4 no request code, 8 no MFMAs; 5 = MFMAs + barrier only,
#  12 = transpose reads + barrier only); the numbers of the ablated builds are times, not results
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_grad3d.py -q -x -k "wgrad" 2>&1 | tail -2
for rep in 1 2; do
  for v in old default ab1 ab4 ab8 ab5 ab12; do
    lib=cvvae_amd/libcvvae_hip.so; [ $v != default ] && lib=gpurun_in/libcvvae_hip_$v.so
    CVVAE_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python tools/train_step_bench.py --dtype bf16 --wgrad-only > gpurun_out/r5_call12_${v}_${rep}.json 2> gpurun_out/r5_call12_${v}_${rep}.err
    python -c "
import json
d=json.loads(open('gpurun_out/r5_call12_${v}_${rep}.json').read().strip().splitlines()[-1])
print('$v rep$rep', [w['ms'] for w in d['wgrad']])"
  done
done
