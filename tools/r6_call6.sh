#!/bin/bash
# round 6, GPU call 6: planar fast-fp32 K loop, two tuning variants against the product build (interleaved processes, one box):
# two pairs of taps in flight (-DCVVAE_XQ_DEPTH8=2) / all staging passes in one batch (-DCVVAE_PL_SBATCH_FULL=1)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
L=$O/r6_ab_planar_variants.log
: > $L
for rnd in 1 2; do
  for lib in "" gpurun_in/libcvvae_pl_xqd2.so gpurun_in/libcvvae_pl_sbfull.so; do
    echo "=== round $rnd lib=${lib:-product}" >> $L
    CVVAE_LIB=${lib:+$PWD/$lib} timeout 300 python tools/conv_bench.py enc128 dec256to128 enc256 enc512 --dtype f32q6 --tfolds --iters 5 --rounds 2 2>&1 | grep "+tf" >> $L
  done
done
cat $L
