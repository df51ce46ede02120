#!/bin/bash
# round 5, GPU call 27: the training step in the other dtypes on the final tree (fp16; fp32 = the register-staged weight-gradient kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for dt in f16 f32; do
  timeout 170 python tools/train_step_bench.py --dtype $dt > gpurun_out/r5_train_step_$dt.json 2> gpurun_out/r5_train_step_$dt.err
  python -c "
import json
d=json.loads(open('gpurun_out/r5_train_step_$dt.json').read().strip().splitlines()[-1])
g=d.get('gradient_parity_vs_reference_modules') or {}
print('$dt', [(w['ms'], w['frac_of_mfma_peak']) for w in d['wgrad']], d['train_step']['backward_ms'], d['train_step']['inference_forward_ms'], {k: (g[k]['input_grad_rel'], g[k]['param_worst_rel']) for k in ('enc','dec') if k in g})"
done
