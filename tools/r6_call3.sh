#!/bin/bash
# round 6, GPU call 3: fast-fp32 staging with ALL loads of a batch requested up front (the per-pass branches had serialised them):
# parity, per-layer A/B, timelines, the cfg 3 step in f32q
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_fast_fp32.py -x -q -m gpu > $O/r6_call3_tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/r6_call3_tests.log
{
timeout 300 python tools/conv_bench.py enc128 dec256to128 --dtype f32q6 --tfolds --iters 5 --rounds 3 --force "" --force 2x4x32:2x4x1:1
timeout 300 python tools/conv_bench.py enc256 enc512 --dtype f32q6 --tfolds --iters 5 --rounds 3 --force "" --force 1x4x32:1x8x1:1
timeout 300 python tools/conv_bench.py c2d128res --dtype f32q6 --iters 5 --rounds 3 --force "" --force 1x8x32:2x4x1:2
timeout 300 python tools/conv_bench.py c2d256res c2d512res upfold256to512 upfold512 down128 --dtype f32q6 --iters 5 --rounds 3
} > $O/r6_ab_planar_fast_fp32_v2.log 2>&1
cat $O/r6_ab_planar_fast_fp32_v2.log
: > $O/r6_probe_fast_fp32_timelines_v2.log
for c in 12 15 17; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCVVAE_CONV_PROBE -DCFG=$c -Icvvae_amd/csrc tools/probes/conv_probe.hip -o /tmp/conv_probe_$c 2>/dev/null
  echo "=== CFG $c" >> $O/r6_probe_fast_fp32_timelines_v2.log
  timeout 60 /tmp/conv_probe_$c >> $O/r6_probe_fast_fp32_timelines_v2.log 2>&1
done
cat $O/r6_probe_fast_fp32_timelines_v2.log
timeout 600 python bench.py --dtype f32q --steps 5 --warmup 2 --no-cpu-baseline --no-tolerance-mode --full-json $O/r6_bench_cfg3_f32q_call3_full.json > $O/r6_bench_cfg3_f32q_call3.json 2> $O/r6_bench_cfg3_f32q_call3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_bench_cfg3_f32q_call3.json').read().strip().splitlines()[-1])
print('f32q', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('encode_ms'), d.get('parity'))
f=json.load(open('gpurun_out/r6_bench_cfg3_f32q_call3_full.json'))
for n,v in list(f['kernels'].items())[:12]: print(n, v.get('ms'), v.get('tflops'))
PY
