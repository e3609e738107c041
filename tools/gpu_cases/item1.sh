# round 6, item 1: the N > 1 readiness work on one GPU - exchange tests in every mode (incl. P2P over real hipIpc mappings), bench.py with two ranks both ways
# (torchrun and self-launched), then the whole GPU suite + smoke and the driver's N = 1 bench line on today's box.
#   bash tools/gpu_call.sh item1 [exchange|suite|bench|all]
WHAT=${1:-all}
if [ "$WHAT" = exchange ] || [ "$WHAT" = all ]; then
	timeout 1500 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_bench_ranks.py -m gpu -q -x > "$OUT/exchange_tests.log" 2>&1; echo "exchange pytest rc=$?" | tee -a "$OUT/exchange_tests.log"; tail -n 6 "$OUT/exchange_tests.log"
	cp bench_extra.json "$OUT/bench_two_ranks_one_gpu_extra.json" 2>/dev/null
fi
if [ "$WHAT" = suite ] || [ "$WHAT" = all ]; then
	timeout 2400 python -m pytest tests -m gpu -q > "$OUT/gpu_suite.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/gpu_suite.log"; tail -n 4 "$OUT/gpu_suite.log"
	timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> "$OUT/gpu_suite.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/gpu_suite.log"
fi
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
	S=$(date +%s)
	timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.txt"; echo "bench rc=$? seconds=$(( $(date +%s) - S ))" | tee "$OUT/bench_rc.txt"
	cp bench_extra.json "$OUT/bench_extra.json" 2>/dev/null
	cat "$OUT/bench_line.json"
fi
