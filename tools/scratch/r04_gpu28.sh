#!/bin/bash
# Round 4, GPU call 28: k_cull_tile<F = 0> with all LDS reads of a frustum iteration in one batch; shared dot products again (5 waves with spills / 4 waves)
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
for v in base cull8_w8c2; do
  LIB=$ROOT/tools/_build/variants/$v/liblumix_mi355.so; [ $v = base ] && LIB=$ROOT/lumixengine_amd/liblumix_mi355.so
  echo "== $v"; LMX_CULL8_WIDTHS=8 LMX_LIB_PATH=$LIB timeout 300 python tools/scratch/cull8_time.py 2>&1 | grep "width 8"
done | tee $OUT/cull8_shape2.txt
