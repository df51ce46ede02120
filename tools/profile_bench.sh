#!/bin/bash
# Profiles `python bench.py` on the GPU box: (1) rocprofv3 --kernel-trace --stats, (2) separate --pmc passes for HBM traffic,
# (3) separate --pmc passes for the SQ counters (MFMA busy, wait buckets).  Counters are never combined with the hip/hsa trace
# domains.  Writes text summaries under gpurun_out/prof_<tag>/ ; copy the ones to keep into profiles/.
#   usage (on the GPU box, from the repo root):  bash tools/profile_bench.sh <tag> [extra bench args]
#   PROFILE_CMD="python $GRAFT_REPO_ROOT/tools/conv_bench.py ..." profiles that command instead of bench.py (PROFILE_STEPS=1)
TAG=${1:-run}
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1 warm-up + 3 timed steps + the 2 passes of the encode/decode split = 6 steps per run
BENCH=${PROFILE_CMD:-"python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity $*"}
STEPS=${PROFILE_STEPS:-6}
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- $BENCH > $O/trace.log 2>&1
DB=$(find $O/trace -name "*.db" | head -1)
if [ -n "$DB" ]; then python $R/tools/rocpd_summary.py $DB > $O/kernel_stats.txt 2>&1; fi
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/sq_counters_available.txt
MOPS=$(grep -E "^SQ_INSTS_VALU_MFMA_MOPS" $O/sq_counters_available.txt | head -4 | tr '\n' ' ')
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" GRBM_GUI_ACTIVE \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
         "$MOPS"; do
  [ -z "$(echo $c | tr -d ' ')" ] && continue
  tag=$(echo $c | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$tag -- $BENCH > $O/pmc_$tag.log 2>&1
done
python $R/tools/pmc_summary.py $O --json=$O/pmc_traffic.json --sq=$O/pmc_sq.json --steps=$STEPS > $O/pmc_summary.txt 2>&1
# (the per-dispatch counter tables are small: kept compressed, so that the summaries can be re-derived off the box)
(cd $O && find . -name "*counter_collection.csv" | tar czf raw_counters.tgz -T - 2>/dev/null)
rm -rf $O/trace $O/pmc_*/   # raw traces are large; the summaries are what is kept
ls -la $O
