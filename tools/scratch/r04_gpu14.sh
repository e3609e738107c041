#!/bin/bash
# Round 4, GPU call 14 / 15: what k_keys_mesh is made of - timing probes (results wrong): 1 = no group histogram atomics, 4 = no tile reservations; call 15: plain pose stamp, 256 copies of the group counters against 64
# 4 = no tile reservations, 7 = none of the three
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
for v in base keys_no_lds_hist base; do
  D=gpurun_out/keysprobe_$v; rm -rf $D; mkdir -p $D
  LIB=$ROOT/tools/_build/variants/$v/liblumix_mi355.so; [ $v = base ] && LIB=$ROOT/lumixengine_amd/liblumix_mi355.so
  (cd /tmp && LMX_LIB_PATH=$LIB timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$D -o p -- python $ROOT/tools/run_workload.py --workload keys --steps 12 > $ROOT/$D/log.txt 2>&1 < /dev/null)
  python - "$v" <<'PY'
import csv,sys
v=sys.argv[1]
for r in csv.DictReader(open(f'gpurun_out/keysprobe_{v}/p_kernel_stats.csv')):
    if 'k_keys_mesh' in r["Name"] or 'k_keys_scatter' in r["Name"]: print("%-12s %-16s calls %s avg %.1f us min %.1f max %.1f" % (v, 'k_keys_mesh' if 'k_keys_mesh' in r["Name"] else 'k_keys_scatter', r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done 2>&1 | tee $OUT/keys_probes3.txt
