#!/bin/bash
# round 5, GPU call 26: the two address forms of the weight-gradient wave-loads compared bit for bit (CVVAE_WGRAD_FAST read per call)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_round5.py -q -x -k "scalar_base or backward_golden" 2>&1 | tail -3
timeout 200 python -m pytest tests/test_gpu_grad3d.py -q -x -k "wgrad" 2>&1 | tail -2
