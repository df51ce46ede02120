#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round2.py -k "gn_silu_apply" -q --maxfail=10 -p no:cacheprovider > gpurun_out/r2d_pytest.log 2>&1
for m in 0 k333 1 0 k333 1; do
  CVVAE_PREPASS=$m timeout 300 python bench.py --no-cpu-baseline --no-parity > gpurun_out/r2d_bench_$m.json 2> gpurun_out/r2d_bench_$m.err
  python -c "
import json
d=json.load(open('gpurun_out/r2d_bench_$m.json')); print('$m', d['value'], d['ms_per_step'], d['encode_ms'], d['decode_ms'])
for k,v in list(d['kernels'].items())[:6]: print('   ',k,v)
"
done
tail -n 3 gpurun_out/r2d_pytest.log
