#!/bin/bash
# SQ / memory counter passes (separate --pmc runs, kernel trace only) for the workloads of tools/run_workload.py.
#   bash tools/collect_counters.sh gpurun_out/r01c "skin keys"
OUT=${1:-gpurun_out/counters}
WORKLOADS=${2:-"skin keys"}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
pass() { # dir name, counters..., then -- command
	local name=$1; shift
	local counters=()
	while [ "$1" != "--" ]; do counters+=("$1"); shift; done
	shift
	(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "${counters[@]}" --output-format csv -d "$ROOT/$OUT/$name" -o p -- "$@" > "$ROOT/$OUT/$name.log" 2>&1 < /dev/null)
	ls "$OUT/$name" 2>/dev/null | head -3
}
for w in $WORKLOADS; do
	cmd=(python "$ROOT/tools/run_workload.py" --workload "$w" --steps 4)
	pass "${w}_sq1" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -- "${cmd[@]}"
	pass "${w}_sq2" SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS -- "${cmd[@]}"
	pass "${w}_fetch" FETCH_SIZE -- "${cmd[@]}"
	pass "${w}_write" WRITE_SIZE -- "${cmd[@]}"
done
python "$ROOT/tools/pmc_summary.py" "$OUT"/*_sq1 "$OUT"/*_sq2 "$OUT"/*_fetch "$OUT"/*_write > "$OUT/summary.json" 2> "$OUT/summary.err" < /dev/null
ls -la "$OUT" | head -30
