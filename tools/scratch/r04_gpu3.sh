#!/bin/bash
# Round 4, GPU call 3: skin A/B (L2 touch of the next block's palettes, block size, 20 k vs 100 k instances), keys default under rocprofv3
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
echo "=== skin A/B"; timeout 600 bash tools/scratch/skin_ab.sh > $OUT/skin_ab3.txt 2>&1; cat $OUT/skin_ab3.txt
echo "=== keys (default build)"; mkdir -p $OUT/keys_default; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/keys_default -o p -- python $ROOT/tools/run_workload.py --workload keys --steps 12 > $ROOT/$OUT/keys_default/log.txt 2>&1 < /dev/null)
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r04/keys_default/p_kernel_stats.csv')):
    if 'k_keys' in r["Name"] or 'k_cull' in r["Name"]: print("%-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
