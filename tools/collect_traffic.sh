#!/bin/bash
# HBM traffic of the headline kernel: two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the headline bench loop.
#   bash tools/collect_traffic.sh gpurun_out/r01e   ->  gpurun_out/r01e/traffic_summary.json (pmc_summary format)
OUT=${1:-gpurun_out/traffic}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
	(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$ROOT/$OUT/$c" -o p -- python "$ROOT/bench.py" --headline-only --no-extras --no-cpu-baseline --steps 30 --warmup 3 > "$ROOT/$OUT/$c.log" 2>&1 < /dev/null)
done
python "$ROOT/tools/pmc_summary.py" "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE" > "$OUT/traffic_summary.json" 2> "$OUT/traffic_summary.err" < /dev/null
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"
head -c 1500 "$OUT/traffic_summary.json"
