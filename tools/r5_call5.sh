#!/bin/bash
# round 5, GPU call 5: four-wave instance restricted to where it pays, starved-grid term of the cost model, 1x1 out of the DMA list:
# A/B on cfg 3, benches of cfg 2 / cfg 1, tuner of cfg 2 with the four-wave instance eligible, round-5 tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round2.py -q -k "round5 or four_wave" > gpurun_out/r5_call5_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_call5_tests.log
rm -f gpurun_out/r5_call5_ab.log
for rep in 1 2 3; do
  for cfg in "four_wave_off:CVVAE_FOUR_WAVE=0" "four_wave_check:CVVAE_FOUR_WAVE=check"; do
    tag=${cfg%%:*}; envs=${cfg#*:}
    env $envs timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-tolerance-mode --verbose > gpurun_out/r5_call5_bench_${tag}_${rep}.json 2> gpurun_out/r5_call5_bench_${tag}_${rep}.err
    python - "$tag" "$rep" <<'PY' >> gpurun_out/r5_call5_ab.log
import json, sys
tag, rep = sys.argv[1:3]
try:
    d = json.loads(open(f"gpurun_out/r5_call5_bench_{tag}_{rep}.json").read().strip().splitlines()[-1])
    ks = d.get("kernels", {})
    print(f"{tag} rep{rep}: {d['value']} frames/s {d['ms_per_step']} ms; encode {d.get('encode_ms')} ({d.get('encode_frac_of_mfma_peak')}) decode {d.get('decode_ms')}; parity {d.get('parity',{}).get('latent_max_abs')}")
    for k, v in list(ks.items())[:9]:
        print(f"    {k:62s} {v['ms']:8.3f} ms x{v['launches']:3d} {v['tflops']:7.1f} TF (executed {v['executed_tflops']:7.1f})")
except Exception as e:
    print(tag, rep, "FAILED", e)
PY
  done
done
for w in cfg2_vae3d_T17_256 cfg1_vae3d_T1_256; do
  for fw in 0 check; do
    CVVAE_FOUR_WAVE=$fw timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-tolerance-mode --verbose > gpurun_out/r5_call5_bench_${w}_fw$fw.json 2> gpurun_out/r5_call5_bench_${w}_fw$fw.err
    python -c "
import json
d=json.loads(open('gpurun_out/r5_call5_bench_${w}_fw$fw.json').read().strip().splitlines()[-1]); print('$w four_wave=$fw', d['value'], d['ms_per_step'], d.get('encode_ms'), d.get('encode_frac_of_mfma_peak'), d.get('parity',{}).get('latent_max_abs'))" >> gpurun_out/r5_call5_ab.log
  done
done
CVVAE_FOUR_WAVE=1 timeout 600 python tools/tune_instances.py --family vae3d --shape 1,3,17,256,256 > gpurun_out/r5_tune_instances_cfg2_v2.log 2>&1
tail -4 gpurun_out/r5_call5_tests.log
grep -v "^    " gpurun_out/r5_call5_ab.log
tail -1 gpurun_out/r5_tune_instances_cfg2_v2.log
