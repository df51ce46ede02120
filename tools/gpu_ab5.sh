#!/bin/bash
# A/B: 1024-thread statistics merge (ab/new.so vs ab/old.so) and GroupNorm+SiLU pass in front of conv_out only (CVVAE_PREPASS=out)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LOG=gpurun_out/r2l_ab.log
: > $LOG
for rnd in 1 2; do
for v in old new new_out; do
  lib=${v%%_*}; pp=auto; [ $v = new_out ] && pp=out
  CVVAE_PREPASS=$pp CVVAE_LIB=$PWD/ab/$lib.so timeout 300 python bench.py --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$v', d['value'], d['ms_per_step'], d['encode_ms'], d['decode_ms'], {n[5:]:v['ms'] for n,v in k.items() if 'w8x1x1' in n})" >> $LOG; done
done
cd /tmp && export TMPDIR=/tmp
for lib in old new; do
  CVVAE_LIB=$GRAFT_REPO_ROOT/ab/$lib.so rocprofv3 --kernel-trace --stats -d /tmp/tr_$lib -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity > /dev/null 2>&1
  DB=$(find /tmp/tr_$lib -name "*.db" | head -1)
  echo "== $lib" >> $GRAFT_REPO_ROOT/$LOG
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB 2>/dev/null | grep -i "finalize\|softmax\|gn_partial" >> $GRAFT_REPO_ROOT/$LOG
done
cd $GRAFT_REPO_ROOT
CVVAE_PREPASS=out timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_baseline_shapes.py -q -x -p no:cacheprovider 2>&1 | tail -3 >> $LOG
cat $LOG
