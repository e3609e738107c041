#!/bin/bash
# Round 4, GPU call 8: k_pose_wave (a wave per 4 instances, memory-order I/O) against round 3's k_pose_palette, phases switched off in turn;
# the skin / animation GPU tests on the real device; the target frame's kernel times
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
for n in 20000 100000; do echo "=== pose_probe $n"; timeout 120 tools/_build/pose_probe $n 2>&1 | tee -a $OUT/pose_probe.txt; done
echo "=== tests"; timeout 900 python -m pytest tests/test_gpu_world_skin.py tests/test_animation.py -m gpu -q -x 2>&1 | tail -4
echo "=== bench"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line8.json 2> $OUT/bench8.err; echo "bench rc=$? bytes=$(wc -c < $OUT/bench_line8.json)"; cat $OUT/bench_line8.json; cp bench_extra.json $OUT/bench_extra8.json 2>/dev/null
grep -E "^\[extra (target_kernel|target_frames|target_char|pose_pal|config3_frame|config3_kernel|skin_vertices_kernel_avg)" $OUT/bench8.err | cut -c1-330
