#!/bin/bash
# Round 4, GPU call 2: skin A/B (block size), keys A/B (look-back x register budgets), the whole GPU suite.
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== skin A/B"; timeout 400 bash tools/scratch/skin_ab.sh > $OUT/skin_ab2.txt 2>&1; cat $OUT/skin_ab2.txt
echo "=== keys A/B"; timeout 900 bash tools/scratch/keys_ab.sh keys_w6_regs4 keys_w4_regs4 keys_w4_regs6 > $OUT/keys_ab2.txt 2>&1; cat $OUT/keys_ab2.txt
echo "=== GPU suite"; timeout 900 python -m pytest tests -m gpu -q > $OUT/gpu_suite2.log 2>&1; echo "suite rc=$?"; tail -8 $OUT/gpu_suite2.log
