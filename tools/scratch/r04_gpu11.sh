#!/bin/bash
# Round 4, GPU call 11: multi-frustum pack + automatic pass width: cull / exchange / adapter / real-header tests, the harness's 6-view timings
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== tests"; timeout 1200 python -m pytest tests/test_gpu_cull.py tests/test_gpu_exchange.py tests/test_gpu_adapter.py tests/test_gpu_real_headers.py tests/test_gpu_bench_ranks.py tests/test_gpu_bridges.py -m gpu -q -x 2>&1 | tail -4
echo "=== harness"; for i in 1 2; do LD_LIBRARY_PATH=$ROOT/lumixengine_amd timeout 300 ./oracle/_ref/real_header_harness 2>&1 | tee $OUT/real_header_harness2.txt | grep -i "views of a frame\|FAIL" | cut -c1-400; done
