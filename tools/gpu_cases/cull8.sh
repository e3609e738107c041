# several-frusta cull: parity on the GPU, then kernel time of the product against build variants (tools/_build/variants/<name>)
#   bash tools/gpu_call.sh cull8 [variant names...]
timeout 900 python -m pytest tests/test_gpu_cull.py -m gpu -x -q -k "not 100m" > "$OUT/cull_tests.log" 2>&1; echo "cull tests rc=$?" | tee -a "$OUT/cull_tests.log"; tail -n 3 "$OUT/cull_tests.log"
{
echo "== product"; LMX_CULL8_WIDTHS=8 timeout 300 python tools/cull8_time.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do
	echo "== $v"; LMX_CULL8_WIDTHS=8 LMX_LIB_PATH=tools/_build/variants/$v/liblumix_mi355.so timeout 300 python tools/cull8_time.py 2>&1 | grep -v amdgpu.ids
done
} | tee "$OUT/cull8_times.txt"
