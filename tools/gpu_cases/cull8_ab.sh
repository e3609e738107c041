# several-frusta cull: kernel time of the product against build variants, no tests (tools/_build/variants/<name>); first the MFMA / VALU overlap probe
#   bash tools/gpu_call.sh cull8_ab [variant names...]
./tools/_build/mfma_overlap_probe 2>&1 | tee "$OUT/mfma_overlap_probe.txt"
{
echo "== product"; LMX_CULL8_WIDTHS=8 timeout 300 python tools/cull8_time.py 2>&1 | grep -v amdgpu.ids
for v in "$@"; do
	echo "== $v"; LMX_CULL8_WIDTHS=8 LMX_LIB_PATH=tools/_build/variants/$v/liblumix_mi355.so timeout 300 python tools/cull8_time.py 2>&1 | grep -v amdgpu.ids
done
} | tee "$OUT/cull8_times.txt"
