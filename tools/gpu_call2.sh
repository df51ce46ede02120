#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_gpu.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -k dtype2 -q --maxfail=40 -p no:cacheprovider > gpurun_out/r2b_pytest_ops_f32.log 2>&1
timeout 900 python -m pytest tests/test_gpu_round2.py -k "fp32 or four_wave" -q -s --maxfail=10 -p no:cacheprovider > gpurun_out/r2b_pytest_round2.log 2>&1
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -k dtype2 -q -s --maxfail=10 -p no:cacheprovider > gpurun_out/r2b_pytest_base_f32.log 2>&1
for c in 8 9 3; do PROBE_RES=1 PROBE_STATS=1 timeout 60 ./ab/probe$c > gpurun_out/r2b_probe$c.log 2>&1; done
timeout 60 ./ab/probe3 > gpurun_out/r2b_probe3_nores.log 2>&1
timeout 300 python bench.py --dtype f32 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r2b_bench_f32.json 2> gpurun_out/r2b_bench_f32.err
timeout 600 python bench.py --cpu-baseline-full --steps 3 --warmup 1 --no-roofline > gpurun_out/r2b_bench_cpufull.json 2> gpurun_out/r2b_bench_cpufull.err
tail -3 gpurun_out/r2b_pytest_ops_f32.log gpurun_out/r2b_pytest_round2.log gpurun_out/r2b_pytest_base_f32.log
cat gpurun_out/parity_gpu.txt
