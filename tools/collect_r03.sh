#!/bin/bash
# Round-3 profile collection on the GPU box (everything lands in $OUT; the summaries are copied to profiles/r03/ afterwards):
#   bash tools/collect_r03.sh gpurun_out/r03 [stats|counters|bench|all]
OUT=${1:-gpurun_out/r03}
WHAT=${2:-all}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
prof() { # name, command...: rocprofv3 --kernel-trace --stats summary of one command
	local name=$1; shift
	(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$name" -o p -- "$@" > "$ROOT/$OUT/$name.log" 2>&1 < /dev/null)
	if [ -f "$OUT/$name/p_kernel_stats.csv" ]; then cp "$OUT/$name/p_kernel_stats.csv" "$OUT/${name}_kernel_stats.csv"; else echo "no stats for $name"; tail -n 5 "$OUT/$name.log"; fi
	rm -rf "$OUT/$name"
}
W="python $ROOT/tools/run_workload.py"
if [ "$WHAT" = "stats" ] || [ "$WHAT" = "all" ]; then
	prof bench_headline python "$ROOT/bench.py" --headline-only --no-extras --no-cpu-baseline
	grep -h '^{' "$OUT/bench_headline.log" > "$OUT/bench_headline_under_rocprof.json" 2>/dev/null
	prof cull_all_test_warm $W --workload cull_all_test --steps 40
	prof cull_all_test_cold $W --workload cull_all_test --steps 40 --cold read
	prof cull_all_test_100m $W --workload cull_all_test --steps 12 --entities 100000000
	prof cull_slab_warm $W --workload cull_slab --steps 40
	prof cull_slab_cold $W --workload cull_slab --steps 40 --cold read
	prof keys $W --workload keys --steps 12
	prof cull8 $W --workload cull8 --steps 20
	prof target $W --workload target --steps 12
	prof skin $W --workload skin --steps 12
	prof xform $W --workload xform --steps 12
fi
if [ "$WHAT" = "counters" ] || [ "$WHAT" = "all" ]; then
	bash "$ROOT/tools/collect_counters.sh" "$OUT/counters" "target cull_all_test" > "$OUT/counters.log" 2>&1
	cp "$OUT/counters/summary.json" "$OUT/target_and_cull_all_test_counters.json" 2>/dev/null
	rm -rf "$OUT/counters"
fi
if [ "$WHAT" = "bench" ] || [ "$WHAT" = "all" ]; then
	python "$ROOT/bench.py" > "$OUT/bench_full_run.json" 2> "$OUT/bench_full_run.err"
	python "$ROOT/bench.py" --force-collective --no-extras --no-cpu-baseline --headline-only > "$OUT/bench_force_collective_weak.json" 2> "$OUT/bench_force_collective_weak.err"
	python "$ROOT/bench.py" --force-collective --scaling strong --no-extras --no-cpu-baseline --headline-only > "$OUT/bench_force_collective_strong.json" 2> "$OUT/bench_force_collective_strong.err"
fi
ls -la "$OUT"
