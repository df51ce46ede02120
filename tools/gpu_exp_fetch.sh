#!/bin/bash
# one question: does the packed-weight form with time-fold slots (2x the weight bytes) explain why the dominant kernel fetches 4.5 GB per
# launch inside a step but 1.9 GB in the isolated microbenchmark?  FETCH_SIZE per dispatch of enc128, plain weights then +tf weights.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 70 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fx -- python $R/tools/conv_bench.py enc128 --tfolds --rounds 1 --iters 2 > /tmp/fx.log 2>&1
python - > $R/gpurun_out/r2r_fetch.log <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/fx/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_fwd" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
rows.sort()
print("dispatch order, FETCH_SIZE (KiB; x2 = bytes on gfx950):")
for d, v in rows:
    print(d, round(v), f"{2 * v * 1024 / 1e9:.2f} GB")
PY
grep median /tmp/fx.log >> $R/gpurun_out/r2r_fetch.log
cat $R/gpurun_out/r2r_fetch.log
