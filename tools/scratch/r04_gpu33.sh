#!/bin/bash
# Round 4, GPU call 33 (the tree at the end of the round): the whole GPU suite, the fuzzers, smoke(), the default bench run, rocprofv3 stats + PMC passes
ROOT=$(pwd); OUT=gpurun_out/r04final; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== GPU suite"; timeout 1200 python -m pytest tests -m gpu -q > $OUT/gpu_suite.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|error" $OUT/gpu_suite.log | tail -3
echo "=== fuzzers"; (timeout 200 python -m tests.fuzz_cull --seeds 0-29 --steps 200; timeout 120 python -m tests.fuzz_skin --seeds 0-19; timeout 120 python -m tests.fuzz_world --seeds 0-19) > $OUT/fuzz_on_gpu.log 2>&1; echo "fuzz rc=$?"; tail -3 $OUT/fuzz_on_gpu.log | cut -c1-200
echo "=== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "=== bench"; S=$(date +%s); timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$? bytes=$(wc -c < $OUT/bench_line.json) seconds=$(( $(date +%s) - S ))"; cat $OUT/bench_line.json; cp bench_extra.json $OUT/ 2>/dev/null
echo "=== profiles"; S=$(date +%s); bash tools/collect_r04.sh $OUT all > $OUT/collect.log 2>&1; echo "collect seconds=$(( $(date +%s) - S ))"; ls $OUT | wc -l
