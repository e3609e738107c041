#!/bin/bash
# End-of-round refresh of the profiles whose kernels changed late in round 2 (skinning kernels, dense-tile copy):
#   bash tools/collect_r02_final.sh gpurun_out/r02f
OUT=${1:-gpurun_out/r02f}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
prof() { # name, command...
	local name=$1; shift
	(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$name" -o p -- "$@" > "$ROOT/$OUT/$name.log" 2>&1 < /dev/null)
	if [ -f "$OUT/$name/p_kernel_stats.csv" ]; then cp "$OUT/$name/p_kernel_stats.csv" "$OUT/${name}_kernel_stats.csv"; else echo "no stats for $name"; fi
	rm -rf "$OUT/$name"
}
W="python $ROOT/tools/run_workload.py"
prof bench_headline python "$ROOT/bench.py" --headline-only --no-extras --no-cpu-baseline
grep -h '^{' "$OUT/bench_headline.log" > "$OUT/bench_headline_under_rocprof.json" 2>/dev/null
prof cull_all_accept_warm $W --workload cull_stream --steps 40
prof cull_all_accept_cold $W --workload cull_stream --steps 40 --cold read
for w in skin target; do prof "$w" $W --workload "$w" --steps 12; done
prof skin_distinct $W --workload skin_distinct --instances 1500 --steps 12
ls -la "$OUT"
