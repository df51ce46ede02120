#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r2i_ab_ilv.log
for rnd in 1 2; do
for lib in old new; do
  echo "== $lib (round $rnd)" >> gpurun_out/r2i_ab_ilv.log
  CVVAE_LIB=$PWD/ab/$lib.so timeout 300 python tools/conv_bench.py enc128 enc256 enc512 c2d128res c2d256res c2d512res upfold256to512 dec256to128 --tfolds --rounds 2 2>&1 | grep median >> gpurun_out/r2i_ab_ilv.log
done
done
CVVAE_LIB=$PWD/ab/new.so timeout 300 python tools/conv_bench.py c2d128res --force "" --force 1x8x32:2x4x1:2 --rounds 3 2>&1 | grep median >> gpurun_out/r2i_ab_ilv.log
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -x -p no:cacheprovider -k "not full_size_720" 2>&1 | tail -4 >> gpurun_out/r2i_ab_ilv.log
for lib in old new; do CVVAE_LIB=$PWD/ab/$lib.so timeout 300 python bench.py --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['encode_ms'], d['decode_ms'])" >> gpurun_out/r2i_ab_ilv.log; done
cat gpurun_out/r2i_ab_ilv.log
