# createSortKeys: parity on the GPU, then rocprofv3 kernel stats of the keys workload (cull + createSortKeys on the dense 10 M scene, ~1 M visible)
timeout 900 python -m pytest tests/test_sort_keys.py -m gpu -x -q > "$OUT/keys_tests.log" 2>&1; echo "keys tests rc=$?" | tee -a "$OUT/keys_tests.log"; tail -n 2 "$OUT/keys_tests.log"
prof keys $W --workload keys --steps 12
python - "$OUT/keys_kernel_stats.csv" <<'PY' | tee "$OUT/keys_chain.txt"
import csv, re, sys
tot, runs = 0.0, 1
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_keys_\w+|k_cull_\w+|__amd_rocclr_\w+)", r["Name"])
    if not m: continue
    rows.append((m.group(1), int(r["Calls"]), float(r["AverageNs"])))
    if m.group(1) == "k_keys_mesh": runs = int(r["Calls"])
for name, calls, avg in rows:
    print(f"{name:28s} calls {calls:3d}  avg {avg / 1e3:8.2f} us")
    if name.startswith("k_keys") and "mirror" not in name and calls >= 12: tot += avg * calls / runs
print(f"createSortKeys kernels per run ({runs} runs; a third of them gather the shard windows first): {tot / 1e3:.1f} us")
PY
grep "createSortKeys span" "$OUT/keys.log" | tee -a "$OUT/keys_chain.txt"
