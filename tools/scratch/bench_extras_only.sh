#!/bin/bash
# the bench's extras (incl. the add-stream legs) without the CPU baseline / live traffic passes
python bench.py --no-cpu-baseline --no-live-traffic --big-entities 0 > gpurun_out/bench_extras.json 2> gpurun_out/bench_extras.err
python - <<'PY'
import json
b=json.load(open('gpurun_out/bench_extras.json'))
print(json.dumps(b["extra"]["add_stream"], indent=1))
print(json.dumps(b["extra"]["add_stream_async_compaction"], indent=1))
print("value", b["value"], "frac", b["roofline"]["frac"])
PY
tail -3 gpurun_out/bench_extras.err
