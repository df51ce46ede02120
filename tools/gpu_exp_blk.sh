#!/bin/bash
# experiment: the conv's main input read as [C/16][pixels][16] (CVVAE_EXP_INBLK=1) vs NDHWC -- time and L2-side fetch traffic
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
LOG=$R/gpurun_out/r2k_blk.log
: > $LOG
timeout 600 python tools/conv_bench.py enc128 enc256 enc512 c2d128res c2d256res dec256to128 down128 down256 upfold256to512 --force "" --force '!CVVAE_EXP_INBLK=1' --rounds 3 2>&1 | grep median >> $LOG
cd /tmp && export TMPDIR=/tmp
for v in nhwc blk; do
  if [ $v = blk ]; then F='!CVVAE_EXP_INBLK=1'; else F=''; fi
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $c | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/blk_${v}_$tag -- python $R/tools/conv_bench.py enc128 c2d128res down128 enc256 --force "$F" --rounds 1 --iters 3 > /dev/null 2>&1
  done
done
cd $R
python - >> $LOG <<'PY'
import csv, glob, collections
for v in ("nhwc", "blk"):
    for tag in ("FETCH_SIZE", "TCC_HIT_sum_TCC_MISS_sum"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(f"gpurun_out/blk_{v}_{tag}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "conv_fwd" not in k: continue
                acc[k.split("<")[1][:60] if "<" in k else k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            print(v, tag, k, {c: round(sum(x) / len(x), 1) for c, x in d.items()}, "n", len(next(iter(d.values()))))
PY
rm -rf gpurun_out/blk_*
cat $LOG
