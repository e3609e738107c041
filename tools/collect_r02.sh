#!/bin/bash
# Round-2 profile collection (GPU box, repo root):   bash tools/collect_r02.sh gpurun_out/r02
#   * rocprofv3 --kernel-trace --stats of the headline bench command and of each roofline leg (10 M warm / cache-cold, 100 M)
#   * HBM traffic of each leg: separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (they cannot share a pass), kernel trace only
# Every step runs under `timeout` with stdin closed.
OUT=${1:-gpurun_out/r02}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
prof() { # name, command...
	local name=$1; shift
	(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$name" -o p -- "$@" > "$ROOT/$OUT/$name.log" 2>&1 < /dev/null)
	if [ -f "$OUT/$name/p_kernel_stats.csv" ]; then cp "$OUT/$name/p_kernel_stats.csv" "$OUT/${name}_kernel_stats.csv"; else echo "no stats for $name"; fi
	rm -rf "$OUT/$name"
}
pmc() { # name, command...
	local name=$1; shift
	for c in FETCH_SIZE WRITE_SIZE; do
		(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$ROOT/$OUT/pmc_${name}_$c" -o p -- "$@" > "$ROOT/$OUT/pmc_${name}_$c.log" 2>&1 < /dev/null)
	done
	python "$ROOT/tools/pmc_summary.py" "$OUT/pmc_${name}_FETCH_SIZE" "$OUT/pmc_${name}_WRITE_SIZE" > "$OUT/traffic_$name.json" 2> "$OUT/traffic_$name.err" < /dev/null
	rm -rf "$OUT/pmc_${name}_FETCH_SIZE" "$OUT/pmc_${name}_WRITE_SIZE"
}
W="python $ROOT/tools/run_workload.py"
prof bench_headline python "$ROOT/bench.py" --headline-only --no-extras --no-cpu-baseline
grep -h '^{' "$OUT/bench_headline.log" > "$OUT/bench_headline_under_rocprof.json" 2>/dev/null
LMX_WORKLOAD_CAMERA=nothing prof cull_nothing_visible $W --workload cull_default --steps 200
prof cull_default $W --workload cull_default --steps 200
prof cull_all_test_warm $W --workload cull_all_test --steps 40
prof cull_all_test_cold $W --workload cull_all_test --steps 40 --cold read
prof cull_all_test_coldw $W --workload cull_all_test --steps 40 --cold write
prof cull_all_accept_warm $W --workload cull_stream --steps 40
prof cull_all_accept_cold $W --workload cull_stream --steps 40 --cold read
prof cull_all_test_100m $W --workload cull_all_test --entities 100000000 --steps 10
prof cull_all_accept_100m $W --workload cull_stream --entities 100000000 --steps 10
pmc all_test_cold $W --workload cull_all_test --steps 30 --cold read
pmc all_accept_cold $W --workload cull_stream --steps 30 --cold read
pmc default $W --workload cull_default --steps 30
pmc all_test_100m $W --workload cull_all_test --entities 100000000 --steps 8
for w in xform skin keys target; do prof "$w" $W --workload "$w" --steps 12; done
prof skin_distinct $W --workload skin_distinct --instances 1500 --steps 12
# SQ counters (separate --pmc passes, kernel trace only) of the all-test leg and of the keys kernels
bash "$ROOT/tools/collect_counters.sh" "$OUT/cnt" "cull_all_test keys" > "$OUT/cnt.log" 2>&1
cp "$OUT/cnt/summary.json" "$OUT/cull_all_test_and_keys_counters.json" 2>/dev/null
rm -rf "$OUT/cnt"
"$ROOT/tools/_build/launch_floor_probe" > "$OUT/launch_floor.json" 2> /dev/null
ls -la "$OUT"
