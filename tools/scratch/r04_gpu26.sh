#!/bin/bash
# Round 4, GPU call 26: k_xform_subtree at 5 waves per SIMD (96 VGPRs) against 4 (108)
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT
for v in base xform_w5 base xform_w5; do
  LIB=$ROOT/tools/_build/variants/$v/liblumix_mi355.so; [ $v = base ] && LIB=$ROOT/lumixengine_amd/liblumix_mi355.so
  echo "== $v"; LMX_LIB_PATH=$LIB timeout 300 python tools/scratch/xform_time.py 2>&1 | grep "fused 1" | awk 'NR%2==1'
done | tee $OUT/xform_waves.txt
