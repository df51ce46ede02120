#!/bin/bash
# round 5, GPU call 10: weight-gradient kernel with staggered request / multiply phases and cached row bases
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_grad3d.py -q -x -k "wgrad" 2>&1 | tail -2
for rep in 1 2; do
  for f in 0 1; do
    CVVAE_WGRAD_DMA=$f timeout 600 python tools/train_step_bench.py --dtype bf16 --no-golden > gpurun_out/r5_call10_train_dma${f}_${rep}.json 2> gpurun_out/r5_call10_train_dma${f}_${rep}.err
    python -c "
import json
d=json.loads(open('gpurun_out/r5_call10_train_dma${f}_${rep}.json').read().strip().splitlines()[-1])
print('wgrad_dma=$f rep$rep', [(w['ms'], w['frac_of_mfma_peak']) for w in d['wgrad']], 'backward', d['train_step']['backward_ms'])"
  done
done
timeout 900 python -m pytest tests/test_gpu_grad3d.py -q 2>&1 | tail -2
