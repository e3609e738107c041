#!/bin/bash
# Round 4, GPU call 12: k_cull_tile<F = 0> with all eight loads of a group in flight + packed operands by op_sel (72 VGPRs); k_keys_mesh with the model's
# LOD table in one round trip; k_xform_subtree's wave-consecutive moved list. Cull / keys / world tests first, then the three timings.
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== tests"; timeout 1500 python -m pytest tests/test_gpu_cull.py tests/test_sort_keys.py tests/test_gpu_bridges.py -m gpu -q -x > $OUT/gpu_call12_tests.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error" $OUT/gpu_call12_tests.log | tail -3
echo "=== cull8"; timeout 400 python tools/scratch/cull8_time.py > $OUT/cull8_pass_widths2.txt 2>&1; cat $OUT/cull8_pass_widths2.txt | tail -14
echo "=== keys"; timeout 300 bash tools/scratch/keys_prof.sh 2>&1 | tail -16
echo "=== xform"; timeout 300 python tools/scratch/xform_time.py > $OUT/xform_subtree2.txt 2>&1; cat $OUT/xform_subtree2.txt
echo "=== cull8, 8 waves x 2 chunks"; LMX_LIB_PATH=$ROOT/tools/_build/variants/cull8_w8c2/liblumix_mi355.so timeout 400 python tools/scratch/cull8_time.py > $OUT/cull8_pass_widths2_w8c2.txt 2>&1; grep "width 8" $OUT/cull8_pass_widths2_w8c2.txt
