#!/bin/bash
# kernel durations (rocprofv3 kernel trace) of the headline cull: cameras x tile-test modes x variants
OUT=gpurun_out/floor
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
prof() { local name=$1; shift
	(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$name" -o p -- "$@" > "$ROOT/$OUT/$name.log" 2>&1 < /dev/null)
	[ -f "$OUT/$name/p_kernel_stats.csv" ] && cp "$OUT/$name/p_kernel_stats.csv" "$OUT/${name}_kernel_stats.csv"; rm -rf "$OUT/$name"; }
W="python $ROOT/tools/run_workload.py"
for mode in 1 2; do for variant in 1 4; do
	export LMX_TILE_TEST_MODE=$mode LMX_TILE_VARIANT=$variant
	LMX_WORKLOAD_CAMERA= prof default_m${mode}_v${variant} $W --workload cull_default --steps 300
	LMX_WORKLOAD_CAMERA=nothing prof nothing_m${mode}_v${variant} $W --workload cull_default --steps 300
done; done
export LMX_TILE_VARIANT=1
for mode in 1 2; do export LMX_TILE_TEST_MODE=$mode
	prof alltest_m${mode} $W --workload cull_all_test --steps 60
	prof accept_m${mode} $W --workload cull_stream --steps 60
done
python - <<'PY'
import csv, glob, os
for f in sorted(glob.glob("gpurun_out/floor/*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "k_cull_tile" in r["Name"]:
            print(os.path.basename(f)[:-17], r["Calls"], "avg_us %.2f min_us %.2f" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
