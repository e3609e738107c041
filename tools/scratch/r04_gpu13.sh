#!/bin/bash
# Round 4, GPU call 13: k_keys_mesh with the tile reservations issued ahead of the emit (three barriers per tile), next tile's ids prefetched
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== tests"; timeout 900 python -m pytest tests/test_sort_keys.py tests/test_gpu_bridges.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
echo "=== keys"; timeout 300 bash tools/scratch/keys_prof.sh 2>&1 | grep "k_keys\|k_cull" | head -12; cp gpurun_out/keysprof/k/p_kernel_stats.csv $OUT/keys_kernel_stats_call13.csv
