#!/bin/bash
# Round 4, GPU call 7: the whole GPU suite, the real-header harness's concurrency timings, the default bench run
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== GPU suite"; timeout 900 python -m pytest tests -m gpu -q > $OUT/gpu_suite3.log 2>&1; echo "suite rc=$?"; tail -4 $OUT/gpu_suite3.log
echo "=== harness"; LD_LIBRARY_PATH=$ROOT/lumixengine_amd timeout 300 ./oracle/_ref/real_header_harness 2>&1 | tee $OUT/real_header_harness.txt | grep -i "views of a frame\|OK\|FAIL" | cut -c1-400
echo "=== bench"; timeout 700 python bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$? bytes=$(wc -c < $OUT/bench_line.json)"; cat $OUT/bench_line.json; cp bench_extra.json $OUT/ 2>/dev/null
grep -E "^\[extra (target_kernel|target_frames|target_char|keys_kernels_ms\]|xform|transform_ms|pose_pal|config3_frame|config3_kernel|skin_vertices_kernel_avg|cull8)" $OUT/bench.err | cut -c1-330
