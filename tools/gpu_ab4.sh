#!/bin/bash
# A/B: packed/staged GN+SiLU prologue arithmetic + 1024-thread statistics merge (ab/new.so) vs HEAD (ab/old.so); G14 K-chunk variants
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
LOG=gpurun_out/r2j_ab.log
: > $LOG
for rnd in 1 2; do
for lib in old new; do
  echo "== $lib (round $rnd)" >> $LOG
  CVVAE_LIB=$PWD/ab/$lib.so timeout 300 python tools/conv_bench.py enc128 enc256 enc512 c2d128res c2d256res c2d512res dec256to128 --tfolds --rounds 2 2>&1 | grep median >> $LOG
done
done
echo "== G14 variants (new lib)" >> $LOG
CVVAE_LIB=$PWD/ab/new.so timeout 300 python tools/conv_bench.py c2d128res --force "" --force 1x16x32:2x4x1:1 --force 1x8x32:2x4x1:4 --force 1x8x32:2x4x1:2 --rounds 3 2>&1 | grep median >> $LOG
for rnd in 1 2; do
for lib in old new; do CVVAE_LIB=$PWD/ab/$lib.so timeout 300 python bench.py --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['encode_ms'], d['decode_ms'])" >> $LOG; done
done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_round2.py -q -x -p no:cacheprovider -k "not full_size_720" 2>&1 | tail -4 >> $LOG
cat $LOG
