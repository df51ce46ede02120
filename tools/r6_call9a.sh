#!/bin/bash
# round 6, GPU call 9a: whole GPU suite + smoke on the tree with the fp6 upsample form; the 300-repetition four-wave stress on this box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > $O/r6_gpu_suite.log 2>&1
echo "suite rc=$?"; tail -4 $O/r6_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r6_smoke.log 2>&1; tail -2 $O/r6_smoke.log
{ echo "# round 6: tools/probes/nw4_stress.py REPS=300 on this round's box (every record of every run bit-compared with the first run)"; /opt/rocm/bin/rocm-smi --showuniqueid 2>/dev/null | grep -i "unique"; REPS=300 timeout 600 python tools/probes/nw4_stress.py; } > $O/r6_nw4_stress.log 2>&1
cat $O/r6_nw4_stress.log | grep -v amdgpu.ids
