#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r2j_ab.log
: > $L
for rnd in 1 2; do
for lib in old new; do
  echo "== $lib (round $rnd)" >> $L
  CVVAE_LIB=$PWD/ab/$lib.so timeout 300 python tools/conv_bench.py enc128 enc256 c2d128res c2d256res upfold256to512 dec256to128 --tfolds --rounds 2 2>&1 | grep median | grep -v "force=-  .*k333\|upfold.*force=- " >> $L
done
done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -x -p no:cacheprovider -k "not full_size_720" 2>&1 | tail -3 >> $L
for lib in old new old new; do CVVAE_LIB=$PWD/ab/$lib.so timeout 300 python bench.py --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['encode_ms'], d['decode_ms'])" >> $L; done
cat $L
