#!/bin/bash
# A/B of k_keys_mesh build variants: kernel time under rocprofv3 (keys workload: dense 10 M scene, 1.0 M visible)
ROOT=$(pwd); export TMPDIR=/tmp
for v in "$@"; do
  OUT=gpurun_out/keysab_$v; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && LMX_LIB_PATH=$ROOT/tools/_build/variants/$v/liblumix_mi355.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT -o p -- python $ROOT/tools/run_workload.py --workload keys --steps 12 > $ROOT/$OUT/log.txt 2>&1 < /dev/null)
  python - "$v" <<'PY'
import csv,re,sys
v=sys.argv[1]
for r in csv.DictReader(open(f'gpurun_out/keysab_{v}/p_kernel_stats.csv')):
    if 'k_keys_mesh' in r["Name"]: print("%-8s k_keys_mesh calls %s avg %.1f us min %.1f max %.1f" % (v, r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
