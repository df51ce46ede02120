#!/bin/bash
# round 5, GPU call 29: the full-size backward of BOTH families against the reference's own modules (new fixture: vae3d)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_round5.py -q -k "backward_golden" -s 2>&1 | grep "full-size backward\|passed\|failed\|Error\|assert" | cut -c1-330
