# skinning: parity on the GPU, then k_skin_multi on the north-star load for the three meshes (tools/skin_time.py), optionally under rocprofv3 (arg: prof)
timeout 1200 python -m pytest tests/test_gpu_world_skin.py -m gpu -x -q -k "skin" > "$OUT/skin_tests.log" 2>&1; echo "skin tests rc=$?" | tee -a "$OUT/skin_tests.log"; tail -n 2 "$OUT/skin_tests.log"
timeout 600 python tools/skin_time.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/skin_time.txt"
if [ "$1" = prof ]; then
	prof target24 $W --workload target --steps 24
	grep -E "k_skin_multi|k_pose_palette" "$OUT/target24_kernel_stats.csv" | awk -F, '{print $1, $(NF-6), $(NF-5), $(NF-4), $(NF-2), $(NF-1)}' | cut -c1-60,180- | tee -a "$OUT/skin_time.txt"
fi
