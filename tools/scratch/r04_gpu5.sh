#!/bin/bash
# Round 4, GPU call 5: what bounds k_skin_multi at 100 k instances; the transform step with k_xform_subtree against per-level launches; world / skin GPU tests
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== skin decomposition"; timeout 600 bash tools/scratch/skin_ab.sh > $OUT/skin_ab5.txt 2>&1; cat $OUT/skin_ab5.txt
echo "=== xform"; timeout 300 python tools/scratch/xform_time.py > $OUT/xform_subtree.txt 2>&1; cat $OUT/xform_subtree.txt
echo "=== GPU tests (world, skin, bridges, adapter)"; timeout 900 python -m pytest tests/test_gpu_world_skin.py tests/test_gpu_bridges.py tests/test_gpu_adapter.py tests/test_gpu_real_headers.py tests/test_world_blob.py -m gpu -q > $OUT/gpu_world.log 2>&1; echo "rc=$?"; tail -5 $OUT/gpu_world.log
