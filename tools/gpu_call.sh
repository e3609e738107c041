#!/bin/bash
# One GPU call of the round, by name (replaces the per-call scripts of earlier rounds):
#   gpurun --timeout 900 -- 'bash tools/gpu_call.sh <case> [args...]'
# Everything lands under gpurun_out/r06/<case>/; what is worth keeping is copied to profiles/r06/ by hand afterwards.
CASE=${1:-help}; shift
ROOT=$(pwd)
OUT=gpurun_out/r06/$CASE
mkdir -p "$OUT"
export TMPDIR=/tmp
W="python $ROOT/tools/run_workload.py"

prof() { # name, command...: rocprofv3 --kernel-trace --stats summary of one command
	local name=$1; shift
	(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$name" -o p -- "$@" > "$ROOT/$OUT/$name.log" 2>&1 < /dev/null)
	if [ -f "$OUT/$name/p_kernel_stats.csv" ]; then cp "$OUT/$name/p_kernel_stats.csv" "$OUT/${name}_kernel_stats.csv"; else echo "no stats for $name"; tail -n 5 "$OUT/$name.log"; fi
	rm -rf "$OUT/$name"
}
pmc() { # name, counters..., then -- command: one rocprofv3 --pmc pass (kernel trace only)
	local name=$1; shift
	local counters=()
	while [ "$1" != "--" ]; do counters+=("$1"); shift; done
	shift
	(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "${counters[@]}" --output-format csv -d "$ROOT/$OUT/pmc/$name" -o p -- "$@" > "$ROOT/$OUT/pmc_$name.log" 2>&1 < /dev/null)
}
pmc_summary() { python "$ROOT/tools/pmc_summary.py" "$OUT"/pmc/* > "$OUT/counters_summary.json" 2> "$OUT/counters_summary.err" < /dev/null; rm -rf "$OUT/pmc"; }
sq() { # workload name, extra args: the two SQ passes of a workload
	local w=$1; shift
	pmc ${w}_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -- $W --workload $w --steps 4 "$@"
	pmc ${w}_sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS -- $W --workload $w --steps 4 "$@"
}
variant_time() { # library path (or "" for the product's), workload, extra args: wall time per cull with that library, as run_workload prints it
	local lib=$1; shift
	LMX_LIB_PATH=$lib timeout 300 $W "$@" 2>&1 | grep -v amdgpu.ids | tail -n 3
}

case $CASE in
baseline) # the round's first call: hardware contracts of the planned MFMA pre-test, SQ counters of the 1-frustum all-test launch, today's box
	./tools/_build/mfma_contract_probe > "$OUT/mfma_contract_probe.txt" 2>&1; echo "probe rc=$?" >> "$OUT/mfma_contract_probe.txt"; cat "$OUT/mfma_contract_probe.txt"
	prof cull_all_test_cold $W --workload cull_all_test --steps 40 --cold read
	prof cull8_all_test $W --workload cull8_all_test --steps 20
	sq cull_all_test --cold read
	pmc_summary
	;;
suite) # the GPU suite + smoke of the current tree
	timeout 2400 python -m pytest tests -m gpu -q > "$OUT/gpu_suite.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/gpu_suite.log"; tail -n 5 "$OUT/gpu_suite.log"
	timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/smoke.log"; tail -n 2 "$OUT/smoke.log"
	;;
*) # any other case: a script of that name under tools/gpu_cases/ (kept short; one per experiment family, arguments instead of copies)
	if [ -f "tools/gpu_cases/$CASE.sh" ]; then source "tools/gpu_cases/$CASE.sh" "$@"; else echo "unknown case $CASE"; exit 2; fi
	;;
esac
ls -la "$OUT" | head -40
