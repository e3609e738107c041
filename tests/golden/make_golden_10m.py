"""Writes tests/golden/cull_10m.json: digests (per-type counts + sha256 of the sorted id lists) of the reference CPU path on
BASELINE config 2 at full size (10 M entities), so that the `-m gpu` parity test needs no 10 M oracle run on the GPU box
(the reference allocates one 4 KiB result page per visited cell: 4 GB and ~50 s for the first cull of the sparse scene).

    python tests/golden/make_golden_10m.py          # needs /root/reference (oracle/_ref is built from it); ~15 minutes

Generator = oracle/_ref (the reference's own math.cpp / geometry.cpp object code + the restated CullingSystemImpl driver). The
scenes are regenerated from their seeds by the test; a digest of the scene arrays guards against a different numpy stream."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from lumixengine_amd import scenes  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tests import helpers as H  # noqa: E402

N = 10_000_000
CHURN_FRAMES, CHURN_PER_FRAME, CHURN_CHECK = 20, 1000, (0, 9, 19)


def main():
    po.build()
    kind = "reference" if po.have_reference() else "port"
    o = po.Oracle(kind)
    out = {"generator": o.describe(), "kind": kind, "n": N, "scenes": {}}
    for name, (half, mixed) in H.CONFIG2_SCENES.items():
        t0 = time.time()
        sc = scenes.cull_scene(N, half, seed=2, mixed_types=mixed)
        cs = o.culling_system()
        cs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
        rec = {"half": half, "mixed": mixed, "scene_sha": H.array_digest(sc["entity"], sc["type"], sc["pos"], sc["radius"]), "cells": cs.cell_count(), "cameras": {}}
        print(name, "built", round(time.time() - t0, 1), "s", flush=True)
        for cam, fr in H.config2_cameras(o):
            ids, types, _ = cs.cull(fr, n_threads=8)
            counts, sha = H.visible_digest(ids, types)
            rec["cameras"][cam] = {"counts": counts, "sha256": sha}
            print(" ", cam, counts, round(time.time() - t0, 1), "s", flush=True)
        if mixed:
            fr8 = H.cascade_frusta(o, 8)
            rec["cascades"] = []
            for k in range(8):
                ids, types, _ = cs.cull(fr8[k : k + 1], n_threads=8)
                counts, sha = H.visible_digest(ids, types)
                rec["cascades"].append({"counts": counts, "sha256": sha})
            # the update stream: removals, adds, in-cell and cross-cell sets, 20 frames x 1000
            rec["churn"] = {}
            fr = H.config2_cameras(o)[0][1]
            for f, ops in enumerate(H.churn_stream(sc["pos"], half, CHURN_FRAMES, CHURN_PER_FRAME)):
                for e in ops["remove"]:
                    cs.remove(int(e))
                for i in range(len(ops["add_ids"])):
                    cs.add(int(ops["add_ids"][i]), int(ops["add_type"][i]), ops["add_pos"][i], float(ops["add_radius"][i]))
                for i in range(len(ops["set_ids"])):
                    cs.set(int(ops["set_ids"][i]), ops["set_pos"][i], float(ops["set_radius"][i]))
                if f in CHURN_CHECK:
                    ids, types, _ = cs.cull(fr, n_threads=8)
                    counts, sha = H.visible_digest(ids, types)
                    rec["churn"][str(f)] = {"counts": counts, "sha256": sha}
                    print("  churn frame", f, counts, flush=True)
        out["scenes"][name] = rec
        del cs
    with open(os.path.join(ROOT, "tests", "golden", "cull_10m.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("written")


if __name__ == "__main__":
    main()
