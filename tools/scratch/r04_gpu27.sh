#!/bin/bash
# Round 4, GPU call 27: k_cull_tile<F = 0>: frusta with identical plane normals (the cascades of one light) share a sphere's dot products; against the same
# build without the sharing; cull / exchange / adapter tests
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== tests"; timeout 1500 python -m pytest tests/test_gpu_cull.py tests/test_gpu_exchange.py tests/test_gpu_adapter.py -m gpu -q -x > $OUT/gpu_call27_tests.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error" $OUT/gpu_call27_tests.log | tail -3
for v in base cull8_no_share; do
  LIB=$ROOT/tools/_build/variants/$v/liblumix_mi355.so; [ $v = base ] && LIB=$ROOT/lumixengine_amd/liblumix_mi355.so
  echo "== $v"; LMX_CULL8_WIDTHS=8 LMX_LIB_PATH=$LIB timeout 300 python tools/scratch/cull8_time.py 2>&1 | grep "width 8"
done | tee $OUT/cull8_share_dots.txt
