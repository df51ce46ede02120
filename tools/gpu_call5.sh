#!/bin/bash
# full GPU suite + bench (default line) + other workloads kept as logs + profiles
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_gpu.txt
timeout 1800 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/r2e_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2e_pytest.log
timeout 600 python bench.py > gpurun_out/r2e_bench.json 2> gpurun_out/r2e_bench.err
timeout 300 python bench.py --dtype f16 --no-cpu-baseline --no-roofline > gpurun_out/r2e_bench_f16.json 2>/dev/null
for w in cfg2_vae3d_T17_256 cfg5_sd3_B8_T33_512_encode cfg4_sd3_T129_720x1280; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r2e_bench_$w.json 2> gpurun_out/r2e_bench_$w.err
done
timeout 300 python bench.py --workload cfg1_vae3d_T1_256 --dtype f16 --hip-graphs --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/r2e_bench_cfg1_graphs.json 2>/dev/null
timeout 300 python bench.py --workload cfg1_vae3d_T1_256 --dtype f16 --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/r2e_bench_cfg1.json 2>/dev/null
timeout 900 bash tools/profile_bench.sh r2e > gpurun_out/r2e_profile.log 2>&1
tail -n 5 gpurun_out/r2e_pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2e_bench*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['unit'], d['ms_per_step'], d.get('encode_frac_of_mfma_peak'), d.get('roofline',{}).get('frac'), d.get('parity',{}).get('latent_max_abs'))
    except Exception as e: print(f, 'ERR', e)
PY
