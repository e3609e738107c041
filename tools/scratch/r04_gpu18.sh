#!/bin/bash
# Round 4, GPU call 18: what k_cull_tile<F = 0>'s 196-204 us are made of - timing probes (results wrong): 1 = tile-level tests only, 2 = no cell classification,
# 4 = no sphere tests, 6 = neither
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp LMX_CULL8_WIDTHS=8
for v in base cull8_probe2 cull8_probe4 cull8_probe6 cull8_probe14 cull8_probe30; do
  LIB=$ROOT/tools/_build/variants/$v/liblumix_mi355.so; [ $v = base ] && LIB=$ROOT/lumixengine_amd/liblumix_mi355.so
  echo "== $v"; LMX_LIB_PATH=$LIB timeout 300 python tools/scratch/cull8_time.py 2>&1 | grep "width 8"
done | tee $OUT/cull8_probes2.txt
