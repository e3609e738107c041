# createSortKeys, product against build variants (tools/build_variant.py): kernel stats + the chain's span of the keys workload
#   bash tools/gpu_call.sh keys_ab [variant names...]
for v in product "$@"; do
	if [ "$v" = product ]; then unset LMX_LIB_PATH; else export LMX_LIB_PATH=$ROOT/tools/_build/variants/$v/liblumix_mi355.so; fi
	prof ${v}_keys $W --workload keys --steps 12
	echo "$v: $(grep -E 'k_keys_(mesh|scatter|reduce)' "$OUT/${v}_keys_kernel_stats.csv" | sed -E 's/.*(k_keys_[a-z_]+).*\),([0-9]+),[0-9]+,([0-9.]+),.*/\1 \3/' | tr '\n' ' ')"
	grep "createSortKeys span" "$OUT/${v}_keys.log" | sed "s/^/$v: /"
done 2>&1 | tee "$OUT/keys_ab.txt"
unset LMX_LIB_PATH
