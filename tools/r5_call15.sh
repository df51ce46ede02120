#!/bin/bash
# round 5, GPU call 15: rounds rule of the weight-gradient plan (default) against 768 workgroups; scratch builds: both waves of a SIMD
# request first (stag0), input fragments 1 / 5 taps ahead (d1, d5), and nothing but the barrier in the panel loop (ab13 = the fixed
# cost: launch, accumulator clear, partial-tile stores, reduction)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_grad3d.py -q -x -k "wgrad" 2>&1 | tail -2
for rep in 1 2; do
  for v in "0 default" "768 default" "0 stag0" "0 stag0d5" "0 d5" "0 d1" "0 ab13"; do
    set -- $v
    lib=cvvae_amd/libcvvae_hip.so; [ $2 != default ] && lib=gpurun_in/libcvvae_hip_$2.so
    CVVAE_LIB=$GRAFT_REPO_ROOT/$lib CVVAE_WGRAD_WGS=$1 timeout 300 python tools/train_step_bench.py --dtype bf16 --wgrad-only > gpurun_out/r5_call15_w$1_$2_$rep.json 2> gpurun_out/r5_call15_w$1_$2_$rep.err
    python -c "
import json
d=json.loads(open('gpurun_out/r5_call15_w$1_$2_$rep.json').read().strip().splitlines()[-1])
print('wgs=$1 $2 rep$rep', [w['ms'] for w in d['wgrad']])"
  done
done
timeout 600 python tools/train_step_bench.py --dtype bf16 --no-golden > gpurun_out/r5_call15_train.json 2> gpurun_out/r5_call15_train.err
python -c "
import json
d=json.loads(open('gpurun_out/r5_call15_train.json').read().strip().splitlines()[-1])
print('train', [(w['ms'], w['frac_of_mfma_peak']) for w in d['wgrad']], d['train_step'])"
