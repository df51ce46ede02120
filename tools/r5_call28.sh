#!/bin/bash
# round 5, GPU call 28: fp32 master weights under torch.autocast(bf16), and a training step at the inference bench's size (17x512^2)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 120 python tools/train_step_bench.py --dtype f32 --autocast bf16 --no-golden > gpurun_out/r5_train_step_f32_autocast_bf16.json 2> gpurun_out/r5_train_step_f32_autocast_bf16.err
timeout 150 python tools/train_step_bench.py --dtype bf16 --H 512 --W 512 --no-golden > gpurun_out/r5_train_step_bf16_17x512.json 2> gpurun_out/r5_train_step_bf16_17x512.err
for f in r5_train_step_f32_autocast_bf16 r5_train_step_bf16_17x512; do python -c "
import json
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1])
print('$f', [(w['ms'], w['frac_of_mfma_peak']) for w in d['wgrad']], d['train_step'])"; done
