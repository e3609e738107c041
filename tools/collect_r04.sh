#!/bin/bash
# Round-4 profile collection on the GPU box (everything lands in $OUT; the summaries are copied to profiles/r04/ afterwards):
#   bash tools/collect_r04.sh gpurun_out/r04final [stats|counters|all]
OUT=${1:-gpurun_out/r04final}
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
pmc() { # name, counters..., then -- command: one rocprofv3 --pmc pass (kernel trace only), the per-kernel CSV is kept
	local name=$1; shift
	local counters=()
	while [ "$1" != "--" ]; do counters+=("$1"); shift; done
	shift
	(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "${counters[@]}" --output-format csv -d "$ROOT/$OUT/pmc/$name" -o p -- "$@" > "$ROOT/$OUT/pmc_$name.log" 2>&1 < /dev/null)
}
W="python $ROOT/tools/run_workload.py"
if [ "$WHAT" = "stats" ] || [ "$WHAT" = "all" ]; then
	prof bench_headline python "$ROOT/bench.py" --headline-only --no-extras --no-cpu-baseline
	grep -h '^{' "$OUT/bench_headline.log" > "$OUT/bench_headline_under_rocprof.json" 2>/dev/null
	prof cull_all_test_cold $W --workload cull_all_test --steps 40 --cold read
	prof cull_all_test_100m $W --workload cull_all_test --steps 12 --entities 100000000
	prof cull8_all_test $W --workload cull8_all_test --steps 20
	prof keys $W --workload keys --steps 12
	prof target $W --workload target --steps 6
	prof xform $W --workload xform --steps 12
fi
if [ "$WHAT" = "counters" ] || [ "$WHAT" = "all" ]; then
	for w in keys xform cull8_all_test; do
		pmc ${w}_fetch FETCH_SIZE -- $W --workload $w --steps 4
		pmc ${w}_write WRITE_SIZE -- $W --workload $w --steps 4
	done
	pmc cull8_all_test_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -- $W --workload cull8_all_test --steps 4
	pmc cull8_all_test_sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS -- $W --workload cull8_all_test --steps 4
	python "$ROOT/tools/pmc_summary.py" "$OUT"/pmc/* > "$OUT/counters_summary.json" 2> "$OUT/counters_summary.err" < /dev/null
	rm -rf "$OUT/pmc"
fi
ls -la "$OUT"
