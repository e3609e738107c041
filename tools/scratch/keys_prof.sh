#!/bin/bash
ROOT=$(pwd); OUT=gpurun_out/keysprof; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/k -o p -- python $ROOT/tools/run_workload.py --workload keys --steps 12 > $ROOT/$OUT/keys.log 2>&1 < /dev/null)
python - <<'PY'
import csv,re
for r in list(csv.DictReader(open('gpurun_out/keysprof/k/p_kernel_stats.csv')))[:12]:
    m=re.search(r"(k_\w+)(<[^>]*>)?",r["Name"]); print("%-44s calls %4s avg %10.1f ns" % ((m.group(0) if m else r["Name"][:40]), r["Calls"], float(r["AverageNs"])))
PY
tail -3 $OUT/keys.log
