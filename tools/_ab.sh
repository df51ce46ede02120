python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -3
python tools/conv_bench.py c2d128 --force "@1" --force "1x16x32:2x4x1:2@1" --force "1x8x32:1x4x2:2@1"
python tools/conv_bench.py out128to3 --force "@1" --force "2x8x32:8x1x1:1@1" --force "1x8x32:8x1x1:1@1"
