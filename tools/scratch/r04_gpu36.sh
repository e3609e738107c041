#!/bin/bash
# Round 4, GPU call 36: palette rows 0 and 1 interleaved by column (the blended {row 0, row 1} components are register pairs: packed transform, no v_mov):
# skin / animation / bridge tests incl. the full-size digests, k_skin_multi timings (both meshes), k_skin_vertices through the bench's distinct-mesh leg later
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== tests"; timeout 1200 python -m pytest tests/test_gpu_world_skin.py tests/test_animation.py tests/test_gpu_bridges.py -m gpu -q -x > $OUT/gpu_call36_tests.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error" $OUT/gpu_call36_tests.log | tail -2
{
for mesh in 0 1; do for r in 1 2; do echo "== mesh $mesh"; PROBE_REPEAT=20 ./tools/_build/skin_probe_base 100000 2 2 $mesh 0 64 | grep "I= 2 splits=1"; done; done
} 2>&1 | tee $OUT/skin_interleaved_rows.txt
