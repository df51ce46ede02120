#!/bin/bash
# round 5, GPU call 25: the GPU test files outside the training side that share code with this round's last changes, on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_grad.py tests/test_gpu_grad3d.py -q -x 2>&1 | tail -3
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
