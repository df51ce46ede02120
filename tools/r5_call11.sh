#!/bin/bash
# round 5, GPU call 11 (final measurements): whole GPU suite + smoke; rocprofv3 kernel stats and PMC passes of the bench command;
# the default bench line (in-run CPU baseline); the other BASELINE workloads; f32q kernel stats; training step with gradient parity
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r5_gpu_suite.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r5_smoke.log
bash tools/profile_bench.sh r5 > gpurun_out/r5_profile_bench.log 2>&1
cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_cfg3_final.json 2> gpurun_out/r5_bench_cfg3_final.err
cp gpurun_out/bench_full.json gpurun_out/r5_bench_cfg3_final_full.json 2>/dev/null
for w in cfg2_vae3d_T17_256 cfg1_vae3d_T1_256 cfg4_sd3_T129_720x1280 cfg5_sd3_B8_T33_512_encode; do
  timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-tolerance-mode > gpurun_out/r5_bench_$w.json 2> gpurun_out/r5_bench_$w.err
done
for dt in f16 f32q f32; do
  timeout 600 python bench.py --dtype $dt --steps 5 --warmup 2 --no-cpu-baseline --no-tolerance-mode > gpurun_out/r5_bench_cfg3_$dt.json 2> gpurun_out/r5_bench_cfg3_$dt.err
done
timeout 600 python tools/train_step_bench.py --dtype bf16 > gpurun_out/r5_train_step_bf16.json 2> gpurun_out/r5_train_step_bf16.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r5_f32q/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --dtype f32q --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-tolerance-mode > $GRAFT_REPO_ROOT/gpurun_out/prof_r5_f32q_trace.log 2>&1
DB=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_r5_f32q/trace -name "*.db" | head -1)
[ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB > $GRAFT_REPO_ROOT/gpurun_out/r5_bench_cfg3_f32q_kernel_stats.txt 2>&1
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r5_f32q
cd "$GRAFT_REPO_ROOT"
tail -3 gpurun_out/r5_gpu_suite.log
tail -2 gpurun_out/r5_smoke.log
cat gpurun_out/r5_bench_cfg3_final.json | cut -c1-2900
for f in gpurun_out/r5_bench_cfg*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['unit'], d['ms_per_step'], d.get('roofline',{}).get('encode_frac_of_mfma_peak'), (d.get('parity') or {}).get('latent_max_abs'))
except Exception as e: print('$f', 'FAILED', e)"; done
ls gpurun_out/prof_r5/
