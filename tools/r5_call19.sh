#!/bin/bash
# round 5, GPU call 19: per-kernel breakdown of one training step (rocprofv3 --kernel-trace of tools/train_step_bench.py) on the final sources
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train/trace -o t -- python $R/tools/train_step_bench.py --dtype bf16 --no-golden > $R/gpurun_out/prof_train.log 2>&1
DB=$(find $R/gpurun_out/prof_train/trace -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/step_breakdown.py $DB 150 > $R/gpurun_out/r5_train_step_kernel_breakdown.txt 2>&1
rm -rf $R/gpurun_out/prof_train
head -50 $R/gpurun_out/r5_train_step_kernel_breakdown.txt | cut -c1-170
