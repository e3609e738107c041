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
    """kernel name + its template arguments, e.g. k_cull_fused<1, 8, 8>"""
    import re

    for k in ("k_cull_tile", "k_cull_finalize", "k_cull_consolidate", "k_apply_patches", "k_cull_fused", "k_cull_spheres", "k_cull_classify", "k_cull_dynamic", "k_xform_level", "k_xform_scatter", "k_xform_export", "k_sphere_refresh",
              "k_pose_palette", "k_skin_vertices", "k_skin_shared", "k_patch_spheres", "k_keys_mesh", "k_keys_decal", "k_keys_offsets", "k_keys_scatter",
              "k_keys_reduce_rows", "k_keys_reduce_copies", "k_anim_update", "k_anim_blend_stack", "k_bone_attach", "k_palette_expand", "k_xform_subtree", "k_skin_multi", "k_cull_pack"):
        if k in name:
            m = re.search(re.escape(k) + r"(<[^>(]*>)?", name)
            return m.group(0) if m else k
    return name[:64]


def main():
    out = {}
    for d in sys.argv[1:]:
        entry = {}
        for f in glob.glob(os.path.join(d, "*kernel_stats.csv")):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    ks = entry.setdefault("kernel_stats", {})
                    key, calls, avg = short(row["Name"]), int(row["Calls"]), float(row["AverageNs"])
                    if key in ks:  # several instantiations behind one short name: merge
                        old = ks[key]
                        tot = old["calls"] + calls
                        ks[key] = {"calls": tot, "avg_ns": (old["avg_ns"] * old["calls"] + avg * calls) / tot, "min_ns": min(old["min_ns"], float(row["MinNs"])),
                                   "max_ns": max(old["max_ns"], float(row["MaxNs"]))}
                    else:
                        ks[key] = {"calls": calls, "avg_ns": avg, "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"])}
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
