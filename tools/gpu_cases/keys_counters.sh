# createSortKeys: SQ counters + HBM traffic of the chain's kernels (keys workload: cull + createSortKeys on the dense 10 M scene)
sq keys
pmc keys_fetch FETCH_SIZE -- $W --workload keys --steps 4
pmc keys_write WRITE_SIZE -- $W --workload keys --steps 4
pmc_summary
python - "$OUT/counters_summary.json" <<'PY' | tee "$OUT/keys_counters.txt"
import json, sys
c = json.load(open(sys.argv[1]))
for run, v in c.items():
    for kernel, counters in v["counters_mean_per_launch"].items():
        if kernel.startswith("k_keys") and "mirror" not in kernel:
            print(f"{run:12s} {kernel:24s} " + " ".join(f"{k}={val:.0f}" for k, val in counters.items()))
PY
