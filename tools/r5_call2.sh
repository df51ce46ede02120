#!/bin/bash
# round 5, GPU call 2: round-5 tests; interleaved A/B of the DMA-staged conv instances with the SQ counters of both forms;
# one default bench run (compact record, mixed tolerance mode, in-run CPU baseline)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round5.py -q > gpurun_out/r5_call2_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_call2_tests.log
rm -f gpurun_out/r5_call2_ab.log
for rep in 1 2 3; do
  for cfg in "fused_dma0:CVVAE_CONV_DMA=0" "fused_dma1:CVVAE_CONV_DMA=1" "prepassk333_dma0:CVVAE_CONV_DMA=0 CVVAE_PREPASS=k333" "prepassk333_dma1:CVVAE_CONV_DMA=1 CVVAE_PREPASS=k333" "prepassall_dma1:CVVAE_CONV_DMA=1 CVVAE_PREPASS=1"; do
    tag=${cfg%%:*}; envs=${cfg#*:}
    env $envs timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-tolerance-mode --verbose > gpurun_out/r5_call2_bench_${tag}_${rep}.json 2> gpurun_out/r5_call2_bench_${tag}_${rep}.err
    python - "$tag" "$rep" <<'PY' >> gpurun_out/r5_call2_ab.log
import json, sys
tag, rep = sys.argv[1:3]
try:
    d = json.loads(open(f"gpurun_out/r5_call2_bench_{tag}_{rep}.json").read().strip().splitlines()[-1])
    ks = d.get("kernels", {})
    print(f"{tag} rep{rep}: {d['value']} frames/s {d['ms_per_step']} ms; encode {d.get('encode_ms')} decode {d.get('decode_ms')}; parity {d.get('parity',{}).get('latent_max_abs')}")
    for k, v in list(ks.items())[:9]:
        print(f"    {k:62s} {v['ms']:8.3f} ms x{v['launches']:3d} {v['tflops']:7.1f} TF (executed {v['executed_tflops']:7.1f})")
except Exception as e:
    print(tag, rep, "FAILED", e)
PY
  done
done
# SQ counters of the two forms of the big convs (same launches, GroupNorm + SiLU by the pass in both)
cd /tmp
for form in 0 1; do
  O=$GRAFT_REPO_ROOT/gpurun_out/prof_r5_dma$form
  mkdir -p $O
  CVVAE_CONV_DMA=$form CVVAE_PREPASS=k333 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES_SQ_BUSY_CYCLES_SQ_WAVE_CYCLES_GRBM_GUI_ACTIVE -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-tolerance-mode > $O/pmc_sq.log 2>&1
  CVVAE_CONV_DMA=$form CVVAE_PREPASS=k333 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_SQ_WAIT_ANY_SQ_WAIT_INST_ANY_SQ_ACTIVE_INST_ANY_SQ_WAIT_INST_LDS_SQ_WAVE_CYCLES -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-tolerance-mode > $O/pmc_wait.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $O --json=$O/pmc_traffic.json --sq=$O/pmc_sq.json --steps=6 > $O/pmc_summary.txt 2>&1
  rm -rf $O/pmc_SQ_*
done
cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_call2_bench_default.json 2> gpurun_out/r5_call2_bench_default.err
cp gpurun_out/bench_full.json gpurun_out/r5_call2_bench_default_full.json 2>/dev/null
tail -4 gpurun_out/r5_call2_tests.log
grep -v "^    " gpurun_out/r5_call2_ab.log
wc -c gpurun_out/r5_call2_bench_default.json
