#!/bin/bash
# round 5, GPU call 4: round-5 tests again (four-wave switch, DMA list without the strided instances, checksum guard at mode
# transitions); A/B of the four-wave per-frame instance on the bench workload; instance tuning of cfg 2 / cfg 1; train step timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round2.py -q -k "round5 or four_wave" > gpurun_out/r5_call4_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_call4_tests.log
rm -f gpurun_out/r5_call4_ab.log
for rep in 1 2 3; do
  for cfg in "four_wave_off:CVVAE_FOUR_WAVE=0" "four_wave_check:CVVAE_FOUR_WAVE=check"; do
    tag=${cfg%%:*}; envs=${cfg#*:}
    env $envs timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-tolerance-mode --verbose > gpurun_out/r5_call4_bench_${tag}_${rep}.json 2> gpurun_out/r5_call4_bench_${tag}_${rep}.err
    python - "$tag" "$rep" <<'PY' >> gpurun_out/r5_call4_ab.log
import json, sys
tag, rep = sys.argv[1:3]
try:
    d = json.loads(open(f"gpurun_out/r5_call4_bench_{tag}_{rep}.json").read().strip().splitlines()[-1])
    ks = d.get("kernels", {})
    print(f"{tag} rep{rep}: {d['value']} frames/s {d['ms_per_step']} ms; encode {d.get('encode_ms')} decode {d.get('decode_ms')}; parity {d.get('parity',{}).get('latent_max_abs')}")
    for k, v in list(ks.items())[:9]:
        print(f"    {k:62s} {v['ms']:8.3f} ms x{v['launches']:3d} {v['tflops']:7.1f} TF (executed {v['executed_tflops']:7.1f})")
except Exception as e:
    print(tag, rep, "FAILED", e)
PY
  done
done
for w in cfg2_vae3d_T17_256 cfg1_vae3d_T1_256; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-tolerance-mode --verbose > gpurun_out/r5_call4_bench_$w.json 2> gpurun_out/r5_call4_bench_$w.err
done
timeout 600 python tools/tune_instances.py --family vae3d --shape 1,3,17,256,256 > gpurun_out/r5_tune_instances_cfg2.log 2>&1
timeout 300 python tools/tune_instances.py --family vae3d --shape 1,3,1,256,256 > gpurun_out/r5_tune_instances_cfg1.log 2>&1
timeout 600 python tools/train_step_bench.py --dtype bf16 --no-golden > gpurun_out/r5_call4_train_step_bf16.json 2> gpurun_out/r5_call4_train_step_bf16.err
tail -4 gpurun_out/r5_call4_tests.log
grep -v "^    " gpurun_out/r5_call4_ab.log
grep -h "four-wave" gpurun_out/round5_parity.txt gpurun_out/*.err 2>/dev/null | head -5
for w in cfg2_vae3d_T17_256 cfg1_vae3d_T1_256; do python -c "
import json,sys
d=json.loads(open('gpurun_out/r5_call4_bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d.get('encode_ms'), d.get('encode_frac_of_mfma_peak'))"; done
tail -2 gpurun_out/r5_tune_instances_cfg2.log; tail -2 gpurun_out/r5_tune_instances_cfg1.log
cut -c1-700 gpurun_out/r5_call4_train_step_bf16.json | tail -c 500
