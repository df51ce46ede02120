#!/bin/bash
# round 5, GPU call 17: partial tiles stored contiguously per workgroup, zero page cleared inside the kernel (no memset launch):
# tests, the layers' times, the fixed cost (ab13: empty panel loop), the training step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_grad3d.py tests/test_gpu_round5.py -q -x 2>&1 | tail -2
for rep in 1 2; do
  for v in default ab13; do
    lib=cvvae_amd/libcvvae_hip.so; [ $v != default ] && lib=gpurun_in/libcvvae_hip_$v.so
    CVVAE_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python tools/train_step_bench.py --dtype bf16 --wgrad-only > gpurun_out/r5_call17_${v}_$rep.json 2> gpurun_out/r5_call17_${v}_$rep.err
    python -c "
import json
d=json.loads(open('gpurun_out/r5_call17_${v}_$rep.json').read().strip().splitlines()[-1])
print('$v rep$rep', [w['ms'] for w in d['wgrad']])"
  done
done
timeout 600 python tools/train_step_bench.py --dtype bf16 > gpurun_out/r5_call17_train.json 2> gpurun_out/r5_call17_train.err
python -c "
import json
d=json.loads(open('gpurun_out/r5_call17_train.json').read().strip().splitlines()[-1])
print('train', [(w['ms'], w['frac_of_mfma_peak']) for w in d['wgrad']], d['train_step'], d.get('gradient_parity_vs_reference_modules'))"
