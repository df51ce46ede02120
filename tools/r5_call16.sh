#!/bin/bash
# round 5, GPU call 16: per-kernel split of a weight-gradient call (zero-page memset, wgrad_dma_kernel, wgrad_reduce_kernel) for the
# library and for the scratch build with an empty panel loop (ab13)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for v in default ab13; do
  lib=$R/cvvae_amd/libcvvae_hip.so; [ $v != default ] && lib=$R/gpurun_in/libcvvae_hip_$v.so
  CVVAE_LIB=$lib rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_wg_$v/trace -o w -- python $R/tools/train_step_bench.py --dtype bf16 --wgrad-only > $R/gpurun_out/prof_wg_$v.log 2>&1
  DB=$(find $R/gpurun_out/prof_wg_$v/trace -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/r5_wgrad_only_kernel_stats_$v.txt 2>&1
  rm -rf $R/gpurun_out/prof_wg_$v
  head -16 $R/gpurun_out/r5_wgrad_only_kernel_stats_$v.txt | cut -c1-150
done
