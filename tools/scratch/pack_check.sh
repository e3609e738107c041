#!/bin/bash
python -m pytest tests/test_gpu_adapter.py tests/test_gpu_exchange.py tests/test_gpu_real_headers.py "tests/test_gpu_cull.py::test_cull_matches_golden" -m gpu -x -q 2>&1 | tail -3
ROOT=$(pwd); OUT=gpurun_out/packprof; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT -o p -- python $ROOT/bench.py --headline-only --no-extras --no-cpu-baseline > $ROOT/$OUT/bench.log 2>&1 < /dev/null)
python - <<'PY'
import csv,re
for r in list(csv.DictReader(open('gpurun_out/packprof/p_kernel_stats.csv')))[:3]:
    m=re.search(r"(k_\w+)(<[^>]*>)?",r["Name"]); print("%-40s calls %5s avg %9.1f ns min %s" % ((m.group(0) if m else r["Name"][:38]), r["Calls"], float(r["AverageNs"]), r["MinNs"]))
PY
grep -h '^{' gpurun_out/packprof/bench.log | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('step', b['ms_per_step']*1e3, 'us; cull only', b['ms_per_step_cull_only']*1e3)"
