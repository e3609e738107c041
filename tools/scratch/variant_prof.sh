#!/bin/bash
# kernel durations (rocprofv3) of the cull for tile variants given as arguments, cameras: nothing / default; legs: all_test, all_accept (warm + cold)
OUT=gpurun_out/vprof
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
prof() { local name=$1; shift
	(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$name" -o p -- "$@" > "$ROOT/$OUT/$name.log" 2>&1 < /dev/null)
	[ -f "$OUT/$name/p_kernel_stats.csv" ] && cp "$OUT/$name/p_kernel_stats.csv" "$OUT/${name}_kernel_stats.csv"; rm -rf "$OUT/$name"; }
W="python $ROOT/tools/run_workload.py"
for variant in "$@"; do
	export LMX_TILE_VARIANT=$variant
	LMX_WORKLOAD_CAMERA= prof default_v${variant} $W --workload cull_default --steps 300
	LMX_WORKLOAD_CAMERA=nothing prof nothing_v${variant} $W --workload cull_default --steps 300
	prof alltest_v${variant} $W --workload cull_all_test --steps 60
	prof alltestcold_v${variant} $W --workload cull_all_test --steps 60 --cold read
	prof accept_v${variant} $W --workload cull_stream --steps 60
	prof acceptcold_v${variant} $W --workload cull_stream --steps 60 --cold read
done
python - <<'PY'
import csv, glob, os
for f in sorted(glob.glob("gpurun_out/vprof/*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "k_cull_tile" in r["Name"]:
            print(os.path.basename(f)[:-17], r["Calls"], "avg_us %.2f min_us %.2f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
