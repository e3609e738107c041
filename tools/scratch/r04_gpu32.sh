#!/bin/bash
# Round 4, GPU call 32: the 1-frustum launch's floor: camera that sees nothing (every tile rejected by the tile-level test) and the default camera, per kernel
ROOT=$(pwd); OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
for cam in nothing default; do
  D=gpurun_out/floor_$cam; rm -rf $D
  (cd /tmp && LMX_WORKLOAD_CAMERA=$cam timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$D -o p -- python $ROOT/tools/run_workload.py --workload cull_default --steps 200 > $ROOT/$D.log 2>&1 < /dev/null)
  echo "== camera: $cam"; python - $D <<'PY'
import csv,sys,re
for r in list(csv.DictReader(open(sys.argv[1]+'/p_kernel_stats.csv')))[:5]:
    m=re.search(r"(k_\w+)(<[^>]*>)?",r["Name"]); print("  %-36s calls %5s avg %9.1f ns min %9.1f" % ((m.group(0) if m else r["Name"][:36]), r["Calls"], float(r["AverageNs"]), float(r["MinNs"])))
PY
  tail -1 $D.log
done | tee $OUT/cull_floor.txt
