#!/bin/bash
# round 6, GPU call 4: UPPER BOUND on what cutting the halo staging can buy (the march-along-time proposal): the library built with
# -DCVVAE_ABLATE_STAGE=1 (every other staging pass: half the loads, half the GroupNorm + SiLU arithmetic, half the LDS writes) and =2
# (no staging inside the K loop at all), against the product library, interleaved processes on one box.  Results of the ablated
# builds are meaningless; their times are not.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round2.py tests/test_c_abi.py "tests/test_gpu_round5.py::test_weights_written_through_data_and_refresh_weights" "tests/test_gpu_round5.py::test_parameter_checksum_is_accumulated_in_fp64_and_sees_sign_flips" -x -q > $O/r6_call4_tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/r6_call4_tests.log
L=$O/r6_ab_stage_ablation.log
: > $L
for rnd in 1 2; do
  for lib in "" gpurun_in/libcvvae_abl_stage1.so gpurun_in/libcvvae_abl_stage2.so; do
    echo "=== round $rnd lib=${lib:-product}" >> $L
    CVVAE_LIB=${lib:+$PWD/$lib} timeout 300 python tools/conv_bench.py enc128 dec256to128 enc256 c2d128res --dtype bf16 --tfolds --iters 5 --rounds 2 2>&1 | grep -v amdgpu.ids >> $L
    CVVAE_LIB=${lib:+$PWD/$lib} timeout 300 python tools/conv_bench.py enc128 --dtype f32q6 --tfolds --iters 5 --rounds 2 2>&1 | grep -v amdgpu.ids >> $L
  done
done
cat $L
