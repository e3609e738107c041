#!/bin/bash
# Round 4, GPU call 1: skin A/B (k_skin_multi vs k_skin_shared), keys A/B (single material walk), the GPU suite, the default bench run.
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== skin A/B"; timeout 400 bash tools/scratch/skin_ab.sh > $OUT/skin_ab.txt 2>&1; cat $OUT/skin_ab.txt
echo "=== keys A/B"; timeout 600 bash tools/scratch/keys_ab.sh keys_r03_staged keys_regs3 keys_regs4 keys_regs6 keys_regs6_w5 > $OUT/keys_ab.txt 2>&1; cat $OUT/keys_ab.txt
echo "=== GPU suite"; timeout 700 python -m pytest tests -m gpu -x -q > $OUT/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -4 $OUT/gpu_suite.log
echo "=== bench"; timeout 700 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$? bytes=$(wc -c < $OUT/bench_line.json)"; cat $OUT/bench_line.json; cp bench_extra.json $OUT/ 2>/dev/null
grep -E "^\[extra (target|skin|keys|xform|pose|transform|config3)" $OUT/bench.err | cut -c1-400
