#!/bin/bash
# round 5, GPU call 30: image mode (cfg 1) eager and replayed from hipGraphs on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for g in "" "--hip-graphs"; do
  timeout 100 python bench.py --workload cfg1_vae3d_T1_256 --steps 20 --warmup 5 --no-cpu-baseline --no-tolerance-mode $g > gpurun_out/r5_bench_cfg1${g:+_graphs}.json 2> gpurun_out/r5_bench_cfg1${g:+_graphs}.err
  python -c "
import json
d=json.loads(open('gpurun_out/r5_bench_cfg1${g:+_graphs}.json').read().strip().splitlines()[-1]); print('cfg1 $g', d['value'], d['ms_per_step'], d['config'].get('hip_graphs'))"
done
