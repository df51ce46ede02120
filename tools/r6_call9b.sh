#!/bin/bash
# round 6, GPU call 9b: the default bench line (as the driver runs it) + rocprofv3 kernel stats and the separate PMC passes of the same
# command (bf16), and the same for --dtype f32q
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python bench.py --full-json $O/r6_bench_cfg3_default_full.json > $O/r6_bench_cfg3_default.json 2> $O/r6_bench_cfg3_default.err
tail -c 2800 $O/r6_bench_cfg3_default.json
bash tools/profile_bench.sh r6 > $O/r6_profile_bench.log 2>&1
ls $O/prof_r6 | head -20
bash tools/profile_bench.sh r6_f32q --dtype f32q > $O/r6_profile_bench_f32q.log 2>&1
ls $O/prof_r6_f32q | head
