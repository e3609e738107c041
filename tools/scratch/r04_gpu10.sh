#!/bin/bash
# Round 4, GPU call 10: the rebuilt k_pose_palette inside the library: skin / animation GPU tests, the target frames' kernel times
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== tests"; timeout 900 python -m pytest tests/test_gpu_world_skin.py tests/test_animation.py -m gpu -q -x 2>&1 | tail -4
echo "=== bench"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line10.json 2> $OUT/bench10.err; echo "bench rc=$? bytes=$(wc -c < $OUT/bench_line10.json)"; cp bench_extra.json $OUT/bench_extra10.json 2>/dev/null
grep -E "^\[extra (target_kernel|target_frames|target_char|pose_pal|config3_frame|config3_kernel|skin_vertices_kernel_avg)" $OUT/bench10.err | cut -c1-330
