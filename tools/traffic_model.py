#!/usr/bin/env python
"""Memory-traffic model of the hot kernels WITHOUT a GPU (tests/hostsim, build mode "traffic"): the kernel sources are executed on the
simulated device with an instrumentation call in front of every load and store; per kernel the tool prints what the lanes requested from
device memory (= the algorithmic bytes DESIGN.md section 4 prices the kernels with), how many 64-byte sectors / 128-byte lines the
wave-instructions touch (what the L1 / TA path processes; requested / (64 x sectors) = coalescing efficiency) and the launch's footprint
(distinct lines: the least HBM can move). No statement about time.

    python tools/traffic_model.py [--entities 1000000] [--instances 64] [--out profiles/r03/traffic_model_hostsim.json]
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--entities", type=int, default=1_000_000)
    ap.add_argument("--instances", type=int, default=64)
    ap.add_argument("--out", default="")
    ap.add_argument("--by-line", default="", metavar="KERNEL", help="also print the source lines of kernels whose label contains KERNEL, by the bytes their accesses request")
    args = ap.parse_args()
    from tests.hostsim import build as hostsim_build

    if args.by_line:
        os.environ["HOSTSIM_TRAFFIC_BY_PC"] = "1"
    lib_path = hostsim_build.build(sanitize="traffic")
    os.environ["LMX_LIB_PATH"] = lib_path
    os.environ["LMX_HOSTSIM"] = "1"
    from lumixengine_amd import api, scenes

    sim = ctypes.CDLL(lib_path)
    tmp = tempfile.mkdtemp()

    def measure(tag, fn, units, unit_name):
        sim.hostsim_traffic_reset()
        fn()
        path = os.path.join(tmp, "t.json")
        sim.hostsim_traffic_dump(path.encode())
        d = json.load(open(path))
        rows = {}
        for k, v in d.items():
            req = v["read_bytes"] + v["write_bytes"] + v["uniform_read_bytes"]
            rows[k] = dict(v, requested_bytes=req, bytes_per_unit=req / units,
                           read_coalescing=v["read_bytes"] / max(1, 64 * v["read_sector64_requests"]), write_coalescing=v["write_bytes"] / max(1, 64 * v["write_sector64_requests"]),
                           footprint_bytes=v["footprint_read_bytes"] + v["footprint_write_bytes"], footprint_per_unit=(v["footprint_read_bytes"] + v["footprint_write_bytes"]) / units)
        if args.by_line:
            by_line(tag, units, unit_name)
        print(f"== {tag}  ({units} {unit_name})")
        for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["requested_bytes"]):
            print(f"   {k[:58]:58s} x{r['launches']:<2d} requested {r['requested_bytes'] / 1e6:8.2f} MB = {r['bytes_per_unit']:7.2f} B/{unit_name} ({r['read_bytes'] / 1e6:.2f} R + {r['write_bytes'] / 1e6:.2f} W + "
                  f"{r['uniform_read_bytes'] / 1e6:.2f} scalar)  sector use R {r['read_coalescing']:.2f} W {r['write_coalescing']:.2f}  footprint {r['footprint_bytes'] / 1e6:8.2f} MB = {r['footprint_per_unit']:6.2f} B/{unit_name}")
        return {"units": units, "unit": unit_name, "kernels": rows}

    def by_line(tag, units, unit_name):
        import subprocess

        path = os.path.join(tmp, "pc.tsv")
        sim.hostsim_traffic_dump_by_pc(path.encode())
        rows = [l.rstrip("\n").split("\t") for l in open(path)]
        rows = [r for r in rows if args.by_line in r[0]]
        if not rows:
            return
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-symbolizer", "--obj=" + lib_path, "--inlines", "--output-style=JSON"] + ["0x" + r[1] for r in rows],
                             capture_output=True, text=True).stdout
        agg = {}
        for r, entry in zip(rows, json.loads(out)):
            loc = "?"
            for frame in entry.get("Symbol", []):  # innermost frame first: the first one inside csrc/ is the kernel's own line
                if "csrc/" in frame.get("FileName", ""):
                    loc = f"{frame['FileName'].split('csrc/')[1]}:{frame['Line']}"
                    break
            a = agg.setdefault(loc, [0, 0, 0, 0])
            for i in range(4):
                a[i] += int(r[2 + i])
        print(f"-- {tag}: source lines of *{args.by_line}* by requested bytes")
        for loc, a in sorted(agg.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:25]:
            use = (a[0] + a[1]) / max(1, 64 * a[2])
            print(f"   {loc:34s} {a[0] / units:8.2f} B read {a[1] / units:8.2f} B written per {unit_name}   sector use {use:.2f}   {a[3]} wave-instructions")

    report = {}
    ctx = api.Context(0)
    n = args.entities
    # ---- culling: the all-test leg (every sphere fetched and tested), the default camera, the dense scene
    half = 15000.0 * (n / 1e7) ** (1.0 / 3.0)
    sc = scenes.cull_scene(n, half, seed=2)
    cs = api.CullingSystem(ctx)
    cs.setOption(api.CULL_OPT_TILE_VARIANT, 1)
    fr = api.viewport_frustum()
    big = dict(sc)
    big["radius"] = scenes.all_test_radii(n)
    cs.build(big["entity"], big["type"], big["pos"], big["radius"])
    cs.cull(fr)  # (first cull: result buffers)
    report["cull_all_test"] = measure("cull, all-test scene (roofline leg), 1 frustum, tile variant 1", lambda: cs.cull(fr), n, "entity")
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    cs.setOption(api.CULL_OPT_TILE_VARIANT, -1)
    cs.cull(fr)
    report["cull_default_camera"] = measure("cull, sparse scene, default camera (headline step: cull + pack)", lambda: (cs.cull(fr), cs.cull(fr).map_all(0)), n, "entity")
    # ---- world hierarchy: 4-deep fans, all roots moved (config 3's shape)
    roots = max(1, n // 1111)
    h = scenes.hierarchy_fans(roots, 10, 4, seed=3)
    w = api.World(ctx)
    tr = h["local"].copy()
    w.build(h["parent"], tr)
    rng = np.random.default_rng(1)
    rt = np.flatnonzero(h["parent"] < 0).astype(np.int32)
    w.setTransforms(rt, scenes.random_transforms(rng, len(rt), 3000.0))
    w.propagate()

    def xform():
        w.setTransforms(rt, scenes.random_transforms(rng, len(rt), 3000.0))
        w.propagate()

    report["xform"] = measure("hierarchy: every root moved, 4 levels", xform, len(h["parent"]), "node")
    # ---- skinning: instances x 64 bones x 10 k vertices of one shared mesh (north-star frame shape), FUSED mode
    s = scenes.skeleton(64, seed=4)
    verts, skin = scenes.skinned_mesh(10_000, 64, seed=6)
    sk = api.Skinning(ctx)
    model = sk.addModel(s["parents"], s["bind"], s["first_nonroot"])
    mesh = sk.addMesh(verts, skin)
    k = args.instances
    sk.setInstances([model] * k, [mesh] * k)
    pos, rot = scenes.relative_poses(k, 64, seed=5)

    def skin_frame():
        sk.uploadPoses(pos, rot)
        sk.run()

    skin_frame()
    report["skin"] = measure("skinning frame: pose palette + vertex kernel (shared mesh, worst-case bone indices)", skin_frame, k * 10_000, "vertex")
    # ---- createSortKeys on what a cull of the dense scene leaves (slot-ordered instance tables)
    nk = min(n, 400_000)
    dsc = scenes.cull_scene(nk, 5000.0 * (nk / 1e7) ** (1.0 / 3.0), seed=2)
    cs.build(dsc["entity"], dsc["type"], dsc["pos"], dsc["radius"])
    ks = scenes.keys_scene(nk, dsc["type"], seed=12, max_sort_key=255)
    sk2 = api.SortKeys(ctx)
    sk2.setModels(ks["models"], ks["mesh_types"])
    sk2.setInstances(ks["model"], ks["material_offset"], ks["mesh_materials"], ks["lod"], ks["flags"], ks["dirty"], ks["pose_frame"])
    sk2.setPositions(dsc["pos"])
    frame = [100]

    def keys_frame():
        frame[0] += 1
        cs.cull(fr)
        sk2.run(api.keys_view(layer_to_bucket=ks["layer_to_bucket"], bucket_depth_sorted=ks["bucket_depth_sorted"], frame_number=frame[0]), 255)

    sk2.setOption(api.KEYS_OPT_SPLIT_STATE, 0)  # rounds 2 / 3's AoS mirror first; the structure of arrays (the default since round 4) last
    keys_frame()
    visible = int(cs.cull(fr).count(0))
    report["keys"] = measure(f"cull + createSortKeys, dense scene, default camera ({visible} visible)", keys_frame, max(visible, 1), "visible entity")
    sk2.setOption(api.KEYS_OPT_SPLIT_STATE, 1)
    keys_frame()
    report["keys_split_state"] = measure("the same with LMX_KEYS_OPT_SPLIT_STATE (lod / Pose::frame in a dense per-slot array)", keys_frame, max(visible, 1), "visible entity")
    sk2.setOption(api.KEYS_OPT_SPLIT_STATE, 2)
    keys_frame()
    report["keys_soa_mirror"] = measure("the same with the mirror as a structure of arrays (LMX_KEYS_OPT_SPLIT_STATE = 2, the default)", keys_frame, max(visible, 1), "visible entity")
    ctx.close()
    if args.out:
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1, sort_keys=True)
        print("written", args.out)


if __name__ == "__main__":
    main()
