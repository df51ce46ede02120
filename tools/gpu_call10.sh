#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2f_smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err
timeout 900 bash tools/profile_bench.sh r2f > gpurun_out/r2f_profile.log 2>&1
timeout 600 python tools/tune_instances.py > gpurun_out/r2f_tune.log 2>&1
tail -3 gpurun_out/r2f_smoke.log
tail -40 gpurun_out/r2f_tune.log
python -c "
import json; d=json.load(open('gpurun_out/r2f_bench.json')); print(d['value'], d['ms_per_step'], d['encode_frac_of_mfma_peak'], d['roofline']['frac'], d['parity']['latent_max_abs'])"
