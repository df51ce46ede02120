#!/bin/bash
# round 5, GPU call 1: new tests, then interleaved A/B of the DMA-staged conv instances (and the GroupNorm + SiLU pass that lets
# every conv take them) on the bench workload
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round5.py -x -q > gpurun_out/r5_call1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_call1_tests.log
for rep in 1 2; do
  for cfg in "dma0:CVVAE_CONV_DMA=0" "dma1:CVVAE_CONV_DMA=1" "dma1_prepass_k333:CVVAE_CONV_DMA=1 CVVAE_PREPASS=k333" "dma1_prepass_all:CVVAE_CONV_DMA=1 CVVAE_PREPASS=1" "dma0_prepass_all:CVVAE_CONV_DMA=0 CVVAE_PREPASS=1"; do
    tag=${cfg%%:*}; envs=${cfg#*:}
    env $envs timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-tolerance-mode --verbose > gpurun_out/r5_call1_bench_${tag}_${rep}.json 2> gpurun_out/r5_call1_bench_${tag}_${rep}.err
    python - "$tag" "$rep" <<'PY' >> gpurun_out/r5_call1_ab.log
import json, sys
tag, rep = sys.argv[1:3]
try:
    d = json.loads(open(f"gpurun_out/r5_call1_bench_{tag}_{rep}.json").read().strip().splitlines()[-1])
    ks = d.get("kernels", {})
    print(f"{tag} rep{rep}: {d['value']} frames/s {d['ms_per_step']} ms; encode {d.get('encode_ms')} decode {d.get('decode_ms')}; parity {d.get('parity',{}).get('latent_max_abs')}")
    for k, v in list(ks.items())[:14]:
        print(f"    {k:62s} {v['ms']:8.3f} ms x{v['launches']:3d} {v['tflops']:7.1f} TF (executed {v['executed_tflops']:7.1f})")
except Exception as e:
    print(tag, rep, "FAILED", e)
PY
  done
done
tail -5 gpurun_out/r5_call1_tests.log
cat gpurun_out/r5_call1_ab.log | grep -v "^    "
