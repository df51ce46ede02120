#!/bin/bash
# Profiles `python bench.py` on the GPU box: (1) rocprofv3 --kernel-trace --stats, (2) separate --pmc passes for HBM traffic.
# Writes text summaries under gpurun_out/prof_<tag>/ ; copy the ones to keep into profiles/.
#   usage (on the GPU box, from the repo root):  bash tools/profile_bench.sh <tag>
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- $BENCH > $O/trace.log 2>&1
DB=$(find $O/trace -name "*.db" | head -1)
if [ -n "$DB" ]; then python $R/tools/rocpd_summary.py $DB > $O/kernel_stats.txt 2>&1; fi
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" GRBM_GUI_ACTIVE; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$tag -- $BENCH > $O/pmc_$tag.log 2>&1
done
python $R/tools/pmc_summary.py $O --json=$O/pmc_traffic.json > $O/pmc_summary.txt 2>&1
rm -rf $O/trace $O/pmc_*/   # raw traces are large; the summaries are what is kept
ls -la $O
