#!/bin/bash
# round 5, GPU call 7: the DMA + transpose-read weight-gradient kernel: semantics probe, wgrad / training tests, train step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/tr_probe.hip -o /tmp/tr_probe > gpurun_out/r5_tr_probe.txt 2>&1 && timeout 60 /tmp/tr_probe >> gpurun_out/r5_tr_probe.txt 2>&1
tail -2 gpurun_out/r5_tr_probe.txt
timeout 900 python -m pytest tests/test_gpu_grad3d.py -q -x -k "wgrad" > gpurun_out/r5_call7_tests_wgrad.log 2>&1
echo "wgrad tests rc=$?" >> gpurun_out/r5_call7_tests_wgrad.log
tail -6 gpurun_out/r5_call7_tests_wgrad.log
for rep in 1 2; do
  for f in 0 1; do
    CVVAE_WGRAD_DMA=$f timeout 600 python tools/train_step_bench.py --dtype bf16 --no-golden > gpurun_out/r5_call7_train_dma${f}_${rep}.json 2> gpurun_out/r5_call7_train_dma${f}_${rep}.err
    python -c "
import json
d=json.loads(open('gpurun_out/r5_call7_train_dma${f}_${rep}.json').read().strip().splitlines()[-1])
print('wgrad_dma=$f rep$rep', [(w['layer'], w['ms'], w['frac_of_mfma_peak']) for w in d['wgrad']], d['train_step']['backward_ms'], d['train_step']['encoder_forward_taped_ms'], d['train_step']['host_launch_ms'])"
  done
done
timeout 1200 python -m pytest tests/test_gpu_grad3d.py tests/test_gpu_round5.py -q > gpurun_out/r5_call7_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_call7_tests.log
tail -5 gpurun_out/r5_call7_tests.log
timeout 600 python tools/train_step_bench.py --dtype bf16 > gpurun_out/r5_train_step_bf16_dma.json 2>/dev/null
