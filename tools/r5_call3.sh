#!/bin/bash
# round 5, GPU call 3: the WHOLE GPU suite with the DMA-staged instances on by default; the training step with the two-panel-deep
# weight-gradient prefetch (+ gradient parity against the reference-generated fixture)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r5_call3_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_call3_tests.log
timeout 600 python tools/train_step_bench.py --dtype bf16 > gpurun_out/r5_train_step_bf16.json 2> gpurun_out/r5_train_step_bf16.err
timeout 600 python tools/train_step_bench.py --dtype f16 --no-golden > gpurun_out/r5_train_step_f16.json 2> gpurun_out/r5_train_step_f16.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_call3_smoke.log 2>&1
tail -4 gpurun_out/r5_call3_tests.log
cat gpurun_out/r5_train_step_bf16.json | cut -c1-3000
tail -3 gpurun_out/r5_call3_smoke.log
