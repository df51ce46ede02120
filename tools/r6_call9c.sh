#!/bin/bash
# round 6, GPU call 9c: the other BASELINE workloads and dtypes on the final tree (one bench line each)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-tolerance-mode "$@" > $O/r6_bench_$tag.json 2> $O/r6_bench_$tag.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r6_bench_$tag.json').read().strip().splitlines()[-1])
    r=d.get('roofline',{})
    print('$tag', d['value'], d['ms_per_step'], 'frac', r.get('frac'), 'enc_ms', r.get('encode_ms'), 'enc_frac', r.get('encode_frac_of_mfma_peak'), (d.get('parity') or {}).get('latent_max_abs'))
except Exception as e:
    print('$tag FAILED', e)
PY
}
run cfg3_f16 --dtype f16 --steps 10 --warmup 3
run cfg3_f32 --dtype f32 --steps 5 --warmup 2
run cfg3_f32q --dtype f32q --steps 5 --warmup 2
run cfg1_vae3d_T1_256 --workload cfg1_vae3d_T1_256 --steps 20 --warmup 5
run cfg2_vae3d_T17_256 --workload cfg2_vae3d_T17_256 --steps 10 --warmup 3
run cfg5_sd3_B8_T33_512_encode --workload cfg5_sd3_B8_T33_512_encode --steps 3 --warmup 1
run cfg4_sd3_T129_720x1280 --workload cfg4_sd3_T129_720x1280 --steps 2 --warmup 1
