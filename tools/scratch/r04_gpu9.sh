#!/bin/bash
# Round 4, GPU call 9: k_pose_wave with 4 / 8 instances per wave, plain and non-temporal stores, phases switched off in turn
OUT=gpurun_out/r04; mkdir -p $OUT
for n in 100000 20000; do for p in pose_probe pose_probe_nt; do echo "=== $p $n"; timeout 120 tools/_build/$p $n 2>&1 | tee -a $OUT/pose_probe2.txt; done; done
