#!/bin/bash
# round 6, GPU call 7: fp6 corrections for the folded upsample convs of a fast-fp32 model (device-side bound): parity, per-layer A/B
# against the bf8 form, the cfg 3 step in f32q
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_fast_fp32.py tests/test_c_abi.py -x -q -s > $O/r6_call7_tests.log 2>&1
echo "tests rc=$?"; grep -E "folded upsample|passed|failed" $O/r6_call7_tests.log | tail -8
for ups in 1 0; do
  CVVAE_F32_FP6_UPS=$ups timeout 600 python bench.py --dtype f32q --steps 5 --warmup 2 --no-cpu-baseline --no-tolerance-mode --full-json $O/r6_bench_cfg3_f32q_ups$ups.full.json > $O/r6_bench_cfg3_f32q_ups$ups.json 2> $O/r6_bench_cfg3_f32q_ups$ups.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6_bench_cfg3_f32q_ups$ups.json').read().strip().splitlines()[-1])
print('f32q fp6-ups=$ups', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('encode_ms'), d.get('parity'))
f=json.load(open('gpurun_out/r6_bench_cfg3_f32q_ups$ups.full.json'))
for n,v in list(f['kernels'].items())[:8]: print('   ', n, v.get('ms'), v.get('tflops'))
PY
done
