#!/bin/bash
# round-2 GPU call 1: full GPU test suite, bench, 4-wave A/B, profiles
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_gpu.txt
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -s -p no:cacheprovider > gpurun_out/r2a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a_pytest.log
timeout 600 python bench.py > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
timeout 300 python tools/conv_bench.py c2d128res c2d128 --force "" --force 1x8x32:1x4x1:2 --rounds 3 > gpurun_out/r2a_ab_c2d128.log 2>&1
timeout 300 python tools/conv_bench.py enc128 dec256to128 --tfolds --force "" --force 2x8x16:1x4x1:1 --rounds 3 > gpurun_out/r2a_ab_enc128.log 2>&1
timeout 900 bash tools/profile_bench.sh r2a > gpurun_out/r2a_profile.log 2>&1
tail -5 gpurun_out/r2a_pytest.log
cat gpurun_out/r2a_ab_c2d128.log gpurun_out/r2a_ab_enc128.log | grep median
