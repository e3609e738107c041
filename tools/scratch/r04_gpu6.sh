#!/bin/bash
# Round 4, GPU call 6: the 8-frusta all-test leg (k_cull_tile<F = 0>) - time by the bench's leg + SQ counters of the same workload
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-live-traffic --steps 20 --warmup 5 > $OUT/bench_legs.json 2> $OUT/bench_legs.err; grep "^\[leg all_test" $OUT/bench_legs.err | cut -c1-600
timeout 900 bash tools/collect_counters.sh $OUT/cull8_counters "cull8_all_test" > $OUT/cull8_counters.log 2>&1; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04/cull8_counters/summary.json'))
for k,v in d.items():
    if 'k_cull_tile' in k or isinstance(v,dict):
        print(k, json.dumps(v)[:1500])
PY
