"""Writes tests/golden/cull_bench_scenes.json: digests of the reference CPU path on the EXACT scenes bench.py times, so that the
bench line can assert the identity of what it measured (sha256 of the visible ids, not just a count) and the `-m gpu` tests can
check BASELINE config 5's size without a 100 M oracle run on the GPU box.

    python tests/golden/make_golden_bench_scenes.py [--skip-100m]      # needs /root/reference (oracle/_ref); ~40 minutes, <= 25 GB

Scenes (lumixengine_amd/scenes.py, all seeded):
  sparse_10m      cull_scene(10 M, +-15000, seed 2), one renderable type        bench.py's headline scene (BASELINE config 2)
  all_test_10m    the same positions, radii = all_test_radii()                  bench.py's roofline leg
  config5_100m    cull_scene(100 M, +-32317, seed 2, mixed types 90/5/5)        BASELINE config 5's size: default camera + 8 cascades
  all_test_100m   the 100 M positions with all_test_radii(), one type           bench.py's HBM-cold-by-size roofline extra
  slab_10m        slab_scene(10 M): normal radii, one layer of cells, ortho slab camera    bench.py's all-CELL_TEST leg with the AABB pre-tests

Generator = the reference's own CullingSystemImpl object code (oracle/_ref: renderer/culling_system.cpp compiled in place). The
reference keeps one 4 KiB page per (cell, type) and pushes one 4 KiB result page per visited cell: ~70 GB for the 100 M scene in one
piece. An entity's visibility depends only on the frustum and its own cell (culling_system.cpp:321-369 - the property the multi-GPU
partition rests on), so the 100 M scenes are culled in SHARDS of whole cells (the reference's CellIndicesHasher mod K,
lumixengine_amd/distributed.py) and the visible sets are united: every verdict is still computed by reference object code.
`all_types_sha256` is the digest of the visible ids regardless of type: renderable types only key the cell pages
(culling_system.cpp:23-40), they do not enter any test, so it is also the digest of the same scene with a single type
(bench.py's 100 M legs)."""
import argparse
import gc
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from lumixengine_amd import distributed, scenes  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tests import helpers as H  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "cull_bench_scenes.json")


def digest(ids, types):
    counts, sha = H.visible_digest(ids, types)
    return {"counts": counts, "sha256": sha, "all_types_sha256": hashlib.sha256(np.sort(ids).astype(np.int32).tobytes()).hexdigest()}


def cull_sharded(o, sc, frusta, n_shards, t0):
    """[(ids, types)] per frustum: the scene culled shard by shard (whole cells per shard) by the reference, results united."""
    got = [([], []) for _ in range(len(frusta))]
    cells = 0
    h = distributed.cell_hash(sc["pos"]) % np.uint32(n_shards) if n_shards > 1 else None
    for s in range(n_shards):
        m = slice(None) if h is None else np.flatnonzero(h == s)
        cs = o.culling_system()
        cs.add_bulk(sc["entity"][m], sc["type"][m], sc["pos"][m], sc["radius"][m])
        cells += cs.cell_count()
        for f in range(len(frusta)):
            ids, types, _ = cs.cull(frusta[f : f + 1], n_threads=1)
            got[f][0].append(ids)
            got[f][1].append(types)
        del cs
        gc.collect()
        print(f"   shard {s + 1}/{n_shards} done, {time.time() - t0:.0f} s", flush=True)
    return [(np.concatenate(a), np.concatenate(b)) for a, b in got], cells


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-100m", action="store_true")
    ap.add_argument("--only", default="", help="comma-separated scene names to (re)generate; the others are kept from the existing file")
    args = ap.parse_args()
    po.build()
    assert po.have_reference(), "needs oracle/_ref (the reference's object code)"
    o = po.Oracle("reference")
    out = json.load(open(OUT)) if os.path.exists(OUT) else {"scenes": {}}
    out["generator"] = o.describe()
    out["kind"] = "reference"
    only = set(x for x in args.only.split(",") if x)
    t0 = time.time()

    def want(name):
        return (not only or name in only) and not (args.skip_100m and name.endswith("100m"))

    def run(name, n, mixed, all_test, cams, n_shards, slab=False):
        if not want(name):
            return
        half = scenes.scaled_half_extent(n)
        if slab:
            sc = scenes.slab_scene(n, seed=2)
            half = sc["half"]
        else:
            sc = scenes.cull_scene(n, half, seed=2, mixed_types=mixed)
        if all_test:
            sc["radius"] = scenes.all_test_radii(n)
        rec = {"n": n, "half": half, "seed": 2, "mixed": mixed, "all_test_radii": all_test, "shards_used_by_generator": n_shards,
               "scene_sha": H.array_digest(sc["entity"], sc["type"], sc["pos"], sc["radius"]), "cameras": {}}
        print(name, "scene ready", round(time.time() - t0), "s", flush=True)
        frusta = np.concatenate([fr for _, fr in cams])
        res, cells = cull_sharded(o, sc, frusta, n_shards, t0)
        rec["cells"] = cells
        for (cam, _), (ids, types) in zip(cams, res):
            rec["cameras"][cam] = digest(ids, types)
            print("  ", cam, rec["cameras"][cam]["counts"], flush=True)
        out["scenes"][name] = rec
        with open(OUT, "w") as fh:
            json.dump(out, fh, indent=1)
        del sc, res
        gc.collect()

    default = [("default", o.viewport_frustum())]
    cascades = [(f"cascade{k}", o.viewport_frustum(**kw)) for k, kw in enumerate(scenes.config5_cascade_kwargs())]
    run("sparse_10m", 10_000_000, False, False, H.config2_cameras(o), 1)
    run("all_test_10m", 10_000_000, False, True, default, 1)
    run("config5_100m", 100_000_000, True, False, default + cascades, 16)
    run("all_test_100m", 100_000_000, False, True, default, 16)
    slab_half = scenes.slab_half_extent(10_000_000)
    run("slab_10m", 10_000_000, False, False, [("slab", o.viewport_frustum(**scenes.slab_frustum_kwargs(slab_half)))], 4, slab=True)
    print("written", OUT, round(time.time() - t0), "s")


if __name__ == "__main__":
    main()
