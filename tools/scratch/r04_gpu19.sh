#!/bin/bash
# Round 4, GPU call 19: k_cull_tile<F = 0> with per-cell class words (one LDS read per chunk instead of one per chunk and frustum; frusta / chunks with
# nothing to test skipped wave-uniformly): cull + exchange + adapter tests, the 8-frusta timings
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== tests"; timeout 1500 python -m pytest tests/test_gpu_cull.py tests/test_gpu_exchange.py tests/test_gpu_adapter.py -m gpu -q -x > $OUT/gpu_call22_tests.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error" $OUT/gpu_call22_tests.log | tail -3
echo "=== cull8"; LMX_CULL8_WIDTHS=8 timeout 400 python tools/scratch/cull8_time.py 2>&1 | grep width | tee $OUT/cull8_pass_widths5.txt
