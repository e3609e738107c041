# the default bench run as the driver runs it (one JSON line on stdout), plus its stderr and bench_extra.json
#   bash tools/gpu_call.sh bench [bench.py args]
S=$(date +%s)
timeout 900 python bench.py "$@" > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.txt"; echo "bench rc=$? seconds=$(( $(date +%s) - S ))" | tee "$OUT/bench_rc.txt"
cp bench_extra.json "$OUT/bench_extra.json" 2>/dev/null
cat "$OUT/bench_line.json"; tail -n 5 "$OUT/bench_stderr.txt"
