# the round's final evidence, one call: the driver's bench line, rocprofv3 kernel stats of the workloads the numbers in DESIGN.md come from, SQ / traffic
# counters of the several-frusta launch, the GPU suite + smoke, fuzzers. Everything under gpurun_out/r06/final -> profiles/r06/final.
#   bash tools/gpu_call.sh final [stats|counters|suite|bench|all]
WHAT=${1:-all}
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
	S=$(date +%s)
	timeout 900 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_stderr_legs_and_extras.txt"; echo "bench rc=$? seconds=$(( $(date +%s) - S ))" | tee "$OUT/bench_rc.txt"
	cp bench_extra.json "$OUT/bench_extra.json" 2>/dev/null
fi
if [ "$WHAT" = stats ] || [ "$WHAT" = all ]; then
	prof bench_headline python "$ROOT/bench.py" --headline-only --no-extras --no-cpu-baseline
	grep -h '^{' "$OUT/bench_headline.log" > "$OUT/bench_headline_under_rocprof.json" 2>/dev/null
	prof cull_all_test_cold $W --workload cull_all_test --steps 40 --cold read
	prof cull_all_test_100m $W --workload cull_all_test --steps 12 --entities 100000000
	prof cull_all_test_dirty $W --workload cull_all_test --steps 30 --cold write
	prof cull_slab_cold $W --workload cull_slab --steps 40 --cold read
	prof cull8_all_test $W --workload cull8_all_test --steps 20
	prof cull8_all_test_cold $W --workload cull8_all_test --steps 20 --cold read
	prof keys $W --workload keys --steps 12
	prof target $W --workload target --steps 24
	prof xform $W --workload xform --steps 12
fi
if [ "$WHAT" = counters ] || [ "$WHAT" = all ]; then
	pmc cull_all_test_fetch FETCH_SIZE -- $W --workload cull_all_test --steps 8 --cold read
	pmc cull_all_test_write WRITE_SIZE -- $W --workload cull_all_test --steps 8 --cold read
	sq cull_all_test --cold read
	pmc cull8_all_test_fetch FETCH_SIZE -- $W --workload cull8_all_test --steps 4
	pmc cull8_all_test_write WRITE_SIZE -- $W --workload cull8_all_test --steps 4
	sq cull8_all_test
	pmc cull8_all_test_mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS -- $W --workload cull8_all_test --steps 4
	pmc keys_fetch FETCH_SIZE -- $W --workload keys --steps 4
	pmc keys_write WRITE_SIZE -- $W --workload keys --steps 4
	pmc_summary
fi
if [ "$WHAT" = suite ] || [ "$WHAT" = all ]; then
	timeout 2400 python -m pytest tests -m gpu -q > "$OUT/gpu_suite.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/gpu_suite.log"; tail -n 4 "$OUT/gpu_suite.log"
	timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> "$OUT/gpu_suite.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/gpu_suite.log"
	(S=$(date +%s); timeout 300 python -m tests.fuzz_cull --seeds 30-69 --steps 300; echo "fuzz_cull rc=$? seconds=$(( $(date +%s) - S ))"; timeout 150 python -m tests.fuzz_skin --seeds 20-39; echo "fuzz_skin rc=$?"; timeout 150 python -m tests.fuzz_world --seeds 20-39; echo "fuzz_world rc=$?"; timeout 300 python -m tests.fuzz_keys --seeds 6-65 --oracle reference; echo "fuzz_keys rc=$?") > "$OUT/fuzz_on_gpu.log" 2>&1
	grep -E "rc=" "$OUT/fuzz_on_gpu.log"
	./tools/_build/mfma_contract_probe > "$OUT/mfma_contract_probe.txt" 2>&1; ./tools/_build/mfma_overlap_probe > "$OUT/mfma_overlap_probe.txt" 2>&1
	./tools/_build/read_probe > "$OUT/read_probe_10m.txt" 2>&1; ./tools/_build/read_probe 100001792 > "$OUT/read_probe_100m.txt" 2>&1
fi
