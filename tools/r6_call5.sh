#!/bin/bash
# round 6, GPU call 5: the whole GPU suite on the tree with ABI 13 / the per-pass weight guard / the planar fast-fp32 instances;
# per-layer times of cfg 2 and cfg 1; a quick bf16 bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > $O/r6_gpu_suite_call5.log 2>&1
echo "suite rc=$?"; tail -5 $O/r6_gpu_suite_call5.log
timeout 200 python tools/layer_times.py --family vae3d --shape 1,3,17,256,256 > $O/r6_layer_times_cfg2.log 2>&1
timeout 200 python tools/layer_times.py --family vae3d --shape 1,3,1,256,256 > $O/r6_layer_times_cfg1.log 2>&1
tail -3 $O/r6_layer_times_cfg2.log $O/r6_layer_times_cfg1.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-tolerance-mode --full-json $O/r6_bench_cfg3_call5_full.json > $O/r6_bench_cfg3_call5.json 2> $O/r6_bench_cfg3_call5.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_bench_cfg3_call5.json').read().strip().splitlines()[-1])
print('bf16', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('encode_ms'), d.get('roofline',{}).get('encode_frac_of_mfma_peak'))
PY
