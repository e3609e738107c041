#!/usr/bin/env python
"""Summarises rocprofv3 CSV output directories (kernel stats + counter collection) into one JSON per round.

    python tools/pmc_summary.py gpurun_out/<dir> [...] > profiles/rNN/<name>.json
"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    for k in ("k_cull_fused", "k_cull_spheres", "k_cull_classify", "k_xform_level", "k_xform_scatter", "k_sphere_refresh", "k_pose_palette", "k_skin_vertices", "k_patch_spheres"):
        if k in name:
            return k
    return name[:48]


def main():
    out = {}
    for d in sys.argv[1:]:
        entry = {}
        for f in glob.glob(os.path.join(d, "*kernel_stats.csv")):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    entry.setdefault("kernel_stats", {})[short(row["Name"])] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]), "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"])}
        for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
            acc = collections.defaultdict(list)
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    acc[(short(row["Kernel_Name"]), row["Counter_Name"])].append(float(row["Counter_Value"]))
            for (k, c), v in sorted(acc.items()):
                v = v[len(v) // 4 :] if len(v) >= 8 else v  # drop warm-up launches
                entry.setdefault("counters_mean_per_launch", {}).setdefault(k, {})[c] = sum(v) / len(v)
        out[os.path.basename(d.rstrip("/"))] = entry
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
