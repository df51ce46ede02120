#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_gpu.txt
timeout 345 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/r2p_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2p_pytest.log
tail -n 3 gpurun_out/r2p_pytest.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2p_smoke.log 2>&1
tail -2 gpurun_out/r2p_smoke.log
timeout 60 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity > gpurun_out/r2p_bench.json 2> gpurun_out/r2p_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r2p_bench.json')); print(d['value'], d['ms_per_step'])"
