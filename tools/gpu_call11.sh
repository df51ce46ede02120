#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_gpu.txt
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/r2o_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2o_pytest.log
timeout 120 python -m pytest tests/test_gpu_grad.py -q -s -p no:cacheprovider -k "input_gradient or resnet_backward or attention_backward" 2>&1 | grep -E "^\[" > gpurun_out/r2o_grad_values.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2o_smoke.log 2>&1
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2o_bench.json 2> gpurun_out/r2o_bench.err
timeout 200 python tools/constraint_bench.py --grad > gpurun_out/r2o_constraint_bench.log 2>&1
tail -n 3 gpurun_out/r2o_pytest.log; tail -2 gpurun_out/r2o_smoke.log; cat gpurun_out/r2o_constraint_bench.log | tail -8
python -c "
import json; d=json.load(open('gpurun_out/r2o_bench.json')); print(d['value'], d['ms_per_step'], d['encode_frac_of_mfma_peak'], d['roofline']['frac'], d['parity']['latent_max_abs'])"
