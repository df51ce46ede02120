#!/bin/bash
# round 5, GPU call 22: bias gradient fused into the weight-gradient launch (ABI 12) + multi-tensor gradient conversions: tests, step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_grad3d.py tests/test_gpu_round5.py -q -x 2>&1 | tail -3
timeout 300 python tools/train_step_bench.py --dtype bf16 --wgrad-only > gpurun_out/r5_call22_wgrad.json 2> gpurun_out/r5_call22_wgrad.err
python -c "
import json
d=json.loads(open('gpurun_out/r5_call22_wgrad.json').read().strip().splitlines()[-1])
print('wgrad-only', [(w['ms'], w['frac_of_mfma_peak']) for w in d['wgrad']])"
timeout 600 python tools/train_step_bench.py --dtype bf16 > gpurun_out/r5_call22_train.json 2> gpurun_out/r5_call22_train.err
python -c "
import json
d=json.loads(open('gpurun_out/r5_call22_train.json').read().strip().splitlines()[-1])
print('train', d['train_step'], d.get('gradient_parity_vs_reference_modules'))"
