#!/bin/bash
# Round 4, GPU call 38: more fuzzer seeds on the MI355X with the round's final tree (every result against the oracle)
OUT=gpurun_out/r04final; mkdir -p $OUT
(S=$(date +%s); timeout 400 python -m tests.fuzz_cull --seeds 30-109 --steps 300; echo "fuzz_cull rc=$? seconds=$(( $(date +%s) - S ))"; S=$(date +%s); timeout 200 python -m tests.fuzz_skin --seeds 20-59; echo "fuzz_skin rc=$? seconds=$(( $(date +%s) - S ))"; S=$(date +%s); timeout 200 python -m tests.fuzz_world --seeds 20-59; echo "fuzz_world rc=$? seconds=$(( $(date +%s) - S ))") > $OUT/fuzz_on_gpu_more_seeds.log 2>&1
grep -E "rc=|Error|error|mismatch|FAIL" $OUT/fuzz_on_gpu_more_seeds.log | head; grep -c "^seed" $OUT/fuzz_on_gpu_more_seeds.log
