# why rocprofv3 sees k_skin_multi at 3.5 ms where bench.py sees 3.0: the same workload, its own timestamps, plain and under the profiler
{
echo "== plain"; timeout 300 $W --workload target --steps 6 2>&1 | grep "target frame"
echo "== under rocprofv3 --kernel-trace --stats"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/t" -o p -- $W --workload target --steps 6 2>&1 | grep "target frame")
grep -E "k_skin_multi|k_pose_palette|k_cull_tile" "$OUT/t/p_kernel_stats.csv" | cut -d, -f1-6 | cut -c1-60,200-
cp "$OUT/t/p_kernel_stats.csv" "$OUT/target_kernel_stats.csv"; rm -rf "$OUT/t"
} 2>&1 | tee "$OUT/skin_rocprof_gap.txt"
