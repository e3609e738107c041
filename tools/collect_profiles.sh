#!/bin/bash
# Collects the rocprofv3 summaries kept under profiles/: run on the GPU box from the repo root.
#   bash tools/collect_profiles.sh gpurun_out/r01b
# Every step runs under `timeout` with stdin closed; nothing here reads from a pipe that may be empty.
OUT=${1:-gpurun_out/profiles}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
prof() { # name, command...
	local name=$1; shift
	(cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$name" -o p -- "$@" > "$ROOT/$OUT/$name.log" 2>&1 < /dev/null)
	if [ -f "$OUT/$name/p_kernel_stats.csv" ]; then cp "$OUT/$name/p_kernel_stats.csv" "$OUT/${name}_kernel_stats.csv"; else echo "no stats for $name"; fi
	rm -rf "$OUT/$name"
}
timeout 400 python bench.py > "$OUT/bench_full_run.json" 2> "$OUT/bench_full_run.err" < /dev/null
prof bench_headline python "$ROOT/bench.py" --headline-only --no-extras --no-cpu-baseline
grep -h '^{' "$OUT/bench_headline.log" > "$OUT/bench_headline_under_rocprof.json" 2>/dev/null
for w in cull_stream xform skin keys target; do prof "$w" python "$ROOT/tools/run_workload.py" --workload "$w" --steps 12; done
prof skin_distinct python "$ROOT/tools/run_workload.py" --workload skin_distinct --instances 1500 --steps 12
ls -la "$OUT"
