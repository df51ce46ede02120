#!/bin/bash
# round 5, GPU call 18 / 23 (final kernel sources): smoke, rocprofv3 kernel stats + PMC passes of the bench command, the default bench line
# (in-run CPU baseline), the training step with gradient parity
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r5_smoke.log
bash tools/profile_bench.sh r5 > gpurun_out/r5_profile_bench.log 2>&1
cd "$GRAFT_REPO_ROOT"
python -c "
import json
d=json.load(open('gpurun_out/prof_r5/pmc_traffic.json')); print(d['library_source_fingerprint'], list(d['kernels'].keys())[:6])"
cp gpurun_out/prof_r5/pmc_traffic.json gpurun_out/prof_r5/pmc_sq.json profiles/   # (so that the bench line below reads this run's counters)
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_cfg3_final.json 2> gpurun_out/r5_bench_cfg3_final.err
cp gpurun_out/bench_full.json gpurun_out/r5_bench_cfg3_final_full.json 2>/dev/null
timeout 600 python tools/train_step_bench.py --dtype bf16 > gpurun_out/r5_train_step_bf16.json 2> gpurun_out/r5_train_step_bf16.err
tail -2 gpurun_out/r5_smoke.log
cat gpurun_out/r5_bench_cfg3_final.json | cut -c1-3000
head -c 1500 gpurun_out/r5_train_step_bf16.json
