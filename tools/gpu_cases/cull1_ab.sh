# 1-frustum cull: parity on the GPU, then rocprofv3 kernel stats of the all-test (cache-cold), streaming (far camera) and dense launches, product against variants
#   bash tools/gpu_call.sh cull1_ab [variant names...]
timeout 900 python -m pytest tests/test_gpu_cull.py -m gpu -x -q -k "not 100m" > "$OUT/cull_tests.log" 2>&1; echo "cull tests rc=$?" | tee -a "$OUT/cull_tests.log"; tail -n 3 "$OUT/cull_tests.log"
for v in product "$@"; do
	if [ "$v" = product ]; then unset LMX_LIB_PATH; else export LMX_LIB_PATH=$ROOT/tools/_build/variants/$v/liblumix_mi355.so; fi
	prof ${v}_all_test_cold $W --workload cull_all_test --steps 40 --cold read
	prof ${v}_stream_cold $W --workload cull_stream --steps 40 --cold read
	prof ${v}_dense_far_cold $W --workload cull_dense --steps 40 --cold read
	for w in all_test_cold stream_cold dense_far_cold; do echo "$v $w: $(grep k_cull_tile "$OUT/${v}_${w}_kernel_stats.csv" | awk -F, '{print $(NF-5), $(NF-4), $(NF-2), $(NF-1)}' | head -2 | tr '\n' ' ')"; done
done 2>&1 | tee "$OUT/cull1_ab.txt"
unset LMX_LIB_PATH
