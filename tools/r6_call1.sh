#!/bin/bash
# round 6, GPU call 1: the planar fast-fp32 instances (Geo::PL, eight fragments per wave) -- parity, per-layer A/B against the
# four-fragment tiles, and the cfg 3 step in f32q
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_fast_fp32.py -x -q -m gpu > $O/r6_call1_tests.log 2>&1
echo "tests rc=$?"; tail -5 $O/r6_call1_tests.log
{
timeout 300 python tools/conv_bench.py enc128 dec256to128 --dtype f32q6 --tfolds --iters 5 --rounds 3 --force "" --force 2x4x32:2x4x1:1
timeout 300 python tools/conv_bench.py enc256 enc512 --dtype f32q6 --tfolds --iters 5 --rounds 3 --force "" --force 1x4x32:1x8x1:1
timeout 300 python tools/conv_bench.py c2d128res --dtype f32q6 --iters 5 --rounds 3 --force "" --force 1x8x32:2x4x1:2
} > $O/r6_ab_planar_fast_fp32.log 2>&1
cat $O/r6_ab_planar_fast_fp32.log
timeout 600 python bench.py --dtype f32q --steps 5 --warmup 2 --no-cpu-baseline --no-tolerance-mode --full-json $O/r6_bench_cfg3_f32q_call1_full.json > $O/r6_bench_cfg3_f32q_call1.json 2> $O/r6_bench_cfg3_f32q_call1.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_bench_cfg3_f32q_call1.json').read().strip().splitlines()[-1])
print('f32q', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('encode_ms'), d.get('parity'))
PY
