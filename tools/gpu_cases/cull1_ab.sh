# 1-frustum cull: rocprofv3 kernel stats of the all-test and slab (cache-cold), default-camera (warm: the headline step's kernel), streaming and dense launches,
# the product against build variants (tools/_build/variants/<name>), all in ONE call (boxes differ by a few per cent).
#   bash tools/gpu_call.sh cull1_ab [variant names...]
for v in product "$@"; do
	if [ "$v" = product ]; then unset LMX_LIB_PATH; else export LMX_LIB_PATH=$ROOT/tools/_build/variants/$v/liblumix_mi355.so; fi
	prof ${v}_all_test_cold $W --workload cull_all_test --steps 40 --cold read
	prof ${v}_all_test_dirty $W --workload cull_all_test --steps 30 --cold write
	prof ${v}_slab_cold $W --workload cull_slab --steps 40 --cold read
	prof ${v}_default_warm $W --workload cull_default --steps 200
	prof ${v}_stream_cold $W --workload cull_stream --steps 40 --cold read
	for w in all_test_cold all_test_dirty slab_cold default_warm stream_cold; do echo "$v $w: $(grep k_cull_tile "$OUT/${v}_${w}_kernel_stats.csv" | awk -F, '{print "calls", $(NF-6), "avg_ns", $(NF-4), "min", $(NF-2), "max", $(NF-1)}' | head -2 | tr '\n' ' ')"; done
done 2>&1 | tee "$OUT/cull1_ab.txt"
unset LMX_LIB_PATH
# the several-frusta kernel shares the header batch: config 5's pass (8 cascades x 10 M all-test), cache-cold and back to back
for v in product "$@"; do
	if [ "$v" = product ]; then unset LMX_LIB_PATH; else export LMX_LIB_PATH=$ROOT/tools/_build/variants/$v/liblumix_mi355.so; fi
	prof ${v}_cull8_cold $W --workload cull8_all_test --steps 20 --cold read
	prof ${v}_cull8_warm $W --workload cull8_all_test --steps 20
	LMX_WORKLOAD_PASS_WIDTH=4 prof ${v}_cull4x2_cold $W --workload cull8_all_test --steps 20 --cold read # the 2..4-frusta shape: the same cascades as two passes of four
	for w in cull8_cold cull8_warm cull4x2_cold; do echo "$v $w: $(grep k_cull_tile "$OUT/${v}_${w}_kernel_stats.csv" | awk -F, '{print "calls", $(NF-6), "avg_ns", $(NF-4), "min", $(NF-2), "max", $(NF-1)}' | head -2 | tr '\n' ' ')"; done
done 2>&1 | tee -a "$OUT/cull1_ab.txt"
unset LMX_LIB_PATH
