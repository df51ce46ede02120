#!/bin/bash
# round 5, GPU call 20: every convolution of one TRAINING step (17x256^2 crop; taped forward + input-gradient convs) under every
# instance that can run it vs the library's choice
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 800 python tools/tune_instances.py --train --shape 1,3,17,256,256 > gpurun_out/r5_tune_instances_train.log 2>&1
grep -c "" gpurun_out/r5_tune_instances_train.log
grep "faster\|sum over" gpurun_out/r5_tune_instances_train.log | cut -c1-220
