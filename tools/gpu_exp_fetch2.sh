#!/bin/bash
# follow-up: is the extra fetch of the time-fold weight form tied to the short-tiles-last order (CVVAE_CONV_LPT=0 disables it)?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
: > $R/gpurun_out/r2s_fetch.log
for lpt in 1 0; do
  rm -rf /tmp/fx
  CVVAE_CONV_LPT=$lpt timeout 60 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fx -- python $R/tools/conv_bench.py enc128 enc256 --tfolds --rounds 1 --iters 2 > /tmp/fx.log 2>&1
  echo "== CVVAE_CONV_LPT=$lpt" >> $R/gpurun_out/r2s_fetch.log
  python - >> $R/gpurun_out/r2s_fetch.log <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/fx/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_fwd" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), r["Kernel_Name"].split("Li")[7:10]))
rows.sort()
print(" ".join(f"{2 * v * 1024 / 1e9:.2f}" for d, v, k in rows), "(GB per dispatch, in order)")
PY
  grep median /tmp/fx.log >> $R/gpurun_out/r2s_fetch.log
done
cat $R/gpurun_out/r2s_fetch.log
