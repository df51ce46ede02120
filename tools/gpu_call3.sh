#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PS='!CVVAE_CONV_PHASE_SYNC=1'
timeout 600 python tools/conv_bench.py c2d128res c2d256res c2d512res enc256 enc512 upfold256to512 --force "" --force "$PS" --rounds 3 > gpurun_out/r2c_ab_phase.log 2>&1
timeout 300 python tools/conv_bench.py enc128 dec256to128 --tfolds --force "" --force "$PS" --rounds 3 >> gpurun_out/r2c_ab_phase.log 2>&1
timeout 300 python tools/conv_bench.py c2d128res --force "1x8x32:1x4x1:2" --force "1x8x32:1x4x1:2$PS" --rounds 3 >> gpurun_out/r2c_ab_phase.log 2>&1
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2c_bench_default.json 2> gpurun_out/r2c_bench_default.err
CVVAE_CONV_PHASE_SYNC=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2c_bench_phasesync.json 2> gpurun_out/r2c_bench_phasesync.err
timeout 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_ops.py -k "four_wave or fused_shortcut or fp32" -q --maxfail=10 -p no:cacheprovider > gpurun_out/r2c_pytest.log 2>&1
grep median gpurun_out/r2c_ab_phase.log
tail -n 3 gpurun_out/r2c_pytest.log
python -c "
import json
for n in ('default','phasesync'):
    d=json.load(open('gpurun_out/r2c_bench_%s.json'%n)); print(n, d['value'], d['ms_per_step'], d['encode_ms'], d['decode_ms'])
"
