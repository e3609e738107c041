#!/bin/bash
# Round 4, GPU call 17 (checkpoint of the tree): the whole GPU suite, smoke(), the default bench run, rocprofv3 stats + PMC passes (tools/collect_r04.sh)
ROOT=$(pwd); OUT=gpurun_out/r04final; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== GPU suite"; timeout 1200 python -m pytest tests -m gpu -q > $OUT/gpu_suite.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed|error" $OUT/gpu_suite.log | tail -3
echo "=== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "=== bench"; S=$(date +%s); timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc=$? bytes=$(wc -c < $OUT/bench_line.json) seconds=$(( $(date +%s) - S ))"; cat $OUT/bench_line.json; cp bench_extra.json $OUT/ 2>/dev/null
echo "=== profiles"; S=$(date +%s); bash tools/collect_r04.sh $OUT all > $OUT/collect.log 2>&1; echo "collect seconds=$(( $(date +%s) - S ))"; ls $OUT | head -40
