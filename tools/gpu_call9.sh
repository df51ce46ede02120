#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_gpu.txt
timeout 1800 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/r2h_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2h_pytest.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2h_bench.json 2> gpurun_out/r2h_bench.err
timeout 300 python bench.py --dtype f32 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r2h_bench_f32.json 2>/dev/null
tail -n 4 gpurun_out/r2h_pytest.log
python - <<'PY'
import json
for f in ('r2h_bench.json','r2h_bench_f32.json'):
    d=json.load(open('gpurun_out/'+f)); print(f, d['value'], d['ms_per_step'], d.get('encode_ms'), d.get('decode_ms'), d.get('encode_frac_of_mfma_peak'), d['parity']['latent_max_abs'])
d=json.load(open('gpurun_out/r2h_bench.json'))
for k,v in d['kernels'].items():
    if 'k111' in k: print(k, v)
PY
