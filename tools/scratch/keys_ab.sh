#!/bin/bash
# A/B of k_keys_mesh build variants x tile reservation (look-back 1 / atomics 0): kernel time under rocprofv3
# (keys workload: dense 10 M scene, 1.0 M visible).   bash tools/scratch/keys_ab.sh <variant>...
ROOT=$(pwd); export TMPDIR=/tmp
for v in "$@"; do
 for lb in 1 0; do
  OUT=gpurun_out/keysab_${v}_lb$lb; rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && LMX_WORKLOAD_KEYS_LOOK_BACK=$lb LMX_LIB_PATH=$ROOT/tools/_build/variants/$v/liblumix_mi355.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT -o p -- python $ROOT/tools/run_workload.py --workload keys --steps 12 > $ROOT/$OUT/log.txt 2>&1 < /dev/null)
  python - "$v" "$lb" <<'PY'
import csv,re,sys
v,lb=sys.argv[1],sys.argv[2]
tot=0.0
for r in csv.DictReader(open(f'gpurun_out/keysab_{v}_lb{lb}/p_kernel_stats.csv')):
    if 'k_keys' in r["Name"]: tot+=float(r["AverageNs"])/1e3
    if 'k_keys_mesh' in r["Name"]: print("%-16s look_back=%s k_keys_mesh calls %s avg %.1f us min %.1f max %.1f" % (v, lb, r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
print("%-16s look_back=%s all k_keys_* kernels: %.1f us per frame" % (v, lb, tot))
PY
  tail -1 $OUT/log.txt
 done
done
