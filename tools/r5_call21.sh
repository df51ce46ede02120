#!/bin/bash
# round 5, GPU call 21: partial tiles [co][tap][ci] + one-block-per-(co, ci block) reduction with a transposed contiguous store;
# scalar-base wave-loads without the zero-page window for all-replicate layers: tests, layers, the training step and its breakdown
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_grad3d.py tests/test_gpu_round5.py -q -x 2>&1 | tail -2
timeout 300 python tools/train_step_bench.py --dtype bf16 --wgrad-only > gpurun_out/r5_call21_wgrad.json 2> gpurun_out/r5_call21_wgrad.err
python -c "
import json
d=json.loads(open('gpurun_out/r5_call21_wgrad.json').read().strip().splitlines()[-1])
print('wgrad-only', [(w['ms'], w['frac_of_mfma_peak']) for w in d['wgrad']])"
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train/trace -o t -- python $R/tools/train_step_bench.py --dtype bf16 > $R/gpurun_out/r5_call21_train.json 2> $R/gpurun_out/prof_train.log
DB=$(find $R/gpurun_out/prof_train/trace -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/step_breakdown.py $DB 150 > $R/gpurun_out/r5_train_step_kernel_breakdown_v2.txt 2>&1
rm -rf $R/gpurun_out/prof_train
cd $R
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r5_call21_train.json').read().strip().splitlines() if l.startswith('{')][-1])
print('train (under rocprof)', d['train_step'], d.get('gradient_parity_vs_reference_modules',{}).get('dec'))"
head -16 gpurun_out/r5_train_step_kernel_breakdown_v2.txt | cut -c1-150
timeout 600 python tools/train_step_bench.py --dtype bf16 --no-golden > gpurun_out/r5_call21_train2.json 2> gpurun_out/r5_call21_train2.err
python -c "
import json
d=json.loads(open('gpurun_out/r5_call21_train2.json').read().strip().splitlines()[-1])
print('train', d['train_step'])"
