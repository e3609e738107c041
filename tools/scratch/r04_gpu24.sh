#!/bin/bash
# Round 4, GPU call 24: k_skin_multi with its LDS sized by the bones the mesh references: the character-like mesh (53 of 64 bones) staged whole (48 KiB, 3 blocks
# per CU) against 52 bones staged (39 KiB, 4 blocks per CU); skin GPU tests
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== tests"; timeout 900 python -m pytest tests/test_gpu_world_skin.py -m gpu -q -x -k "skin" 2>&1 | grep -E "passed|failed|error" | tail -2
{
for st in 64 52 40; do echo "== character-like mesh, $st bones staged"; ./tools/_build/skin_probe_base 100000 2 2 1 0 $st | grep "I= 2 splits=1"; done
echo "== worst-case mesh, 64 bones staged"; ./tools/_build/skin_probe_base 100000 2 2 0 0 64 | grep "I= 2 splits=1"
} 2>&1 | tee $OUT/skin_lds_by_bones.txt
