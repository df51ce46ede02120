#!/bin/bash
# round 5, GPU call 9: operand-prefetch depth of the DMA + transpose-read weight-gradient kernel (library variants under gpurun_in/)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for rep in 1 2; do
  for v in d2 default d6 d9; do
    lib=cvvae_amd/libcvvae_hip.so; [ $v != default ] && lib=gpurun_in/libcvvae_hip_$v.so
    CVVAE_LIB=$GRAFT_REPO_ROOT/$lib timeout 600 python tools/train_step_bench.py --dtype bf16 --no-golden > gpurun_out/r5_call9_train_${v}_${rep}.json 2> gpurun_out/r5_call9_train_${v}_${rep}.err
    python -c "
import json
d=json.loads(open('gpurun_out/r5_call9_train_${v}_${rep}.json').read().strip().splitlines()[-1])
print('depth $v rep$rep', [(w['ms'], w['frac_of_mfma_peak']) for w in d['wgrad']], 'backward', d['train_step']['backward_ms'])"
  done
done
timeout 300 python -m pytest tests/test_gpu_grad3d.py -q -x -k "wgrad" 2>&1 | tail -2
