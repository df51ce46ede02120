#!/bin/bash
# round 5, GPU call 8: counters of the DMA + transpose-read weight-gradient kernel (LDS bank conflicts, matrix pipe, waits)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r5_wgrad
mkdir -p $O
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$tag -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py --dtype bf16 --no-golden --steps 1 > $O/pmc_$tag.log 2>&1
  python - $O/pmc_$tag <<'PY' > $O/summary_$tag.txt 2>&1
import csv, glob, sys, collections
d = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "wgrad" not in k:
            continue
        short = "wgrad_dma" if "wgrad_dma" in k else ("wgrad_reduce" if "reduce" in k else "wgrad_regs")
        short += "_sw2" if "Li2EE" in k or ", 2>" in k else ""
        rows[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(rows.items()):
    print(k, {c: (len(v), sum(v) / len(v)) for c, v in cs.items()})
PY
  cat $O/summary_$tag.txt
  rm -rf $O/pmc_$tag
done
