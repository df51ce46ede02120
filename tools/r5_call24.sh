#!/bin/bash
# round 5, GPU call 24: wgrad_ring_kernel (input rows kept in LDS across consecutive output rows, column-major panels) on / off
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_grad3d.py -q -x -k "wgrad" 2>&1 | tail -3
for rep in 1 2; do
  for r in 1 0; do
    CVVAE_WGRAD_RING=$r timeout 300 python tools/train_step_bench.py --dtype bf16 --wgrad-only > gpurun_out/r5_call24_ring${r}_$rep.json 2> gpurun_out/r5_call24_ring${r}_$rep.err
    python -c "
import json
d=json.loads(open('gpurun_out/r5_call24_ring${r}_$rep.json').read().strip().splitlines()[-1])
print('ring=$r rep$rep', [(w['ms'], w['frac_of_mfma_peak']) for w in d['wgrad']])"
  done
done
timeout 900 python -m pytest tests/test_gpu_grad3d.py tests/test_gpu_round5.py -q -x 2>&1 | tail -2
