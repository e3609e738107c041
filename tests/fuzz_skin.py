"""Randomised skinning tables against the CPU oracle (Pose::computeAbsolute, computeSkinMatrices, evaluateSkin:
src/renderer/pose.cpp:63-134, src/renderer/model.cpp:103-137): skeletons of 1..196 bones (Model::Bone::MAX_COUNT), meshes whose
vertex counts sit on and around every tile size of the two vertex kernels, meshes that reference all bones of a model or only a
few (tile-local bone subsets of k_skin_shared), instance tables made of runs (the register-resident path) and singles (the streaming
path). Palettes bit-exact; vertices bit-exact in LMX_SKIN_EXACT, within 1e-5 per vertex otherwise.

    python -m tests.fuzz_skin [--seeds 0-9]

`tests/test_gpu_world_skin.py::test_skin_fuzz` runs a few seeds; under `pytest --hostsim address,undefined` every table index the
kernels form is bounds-checked."""
from __future__ import annotations

import argparse
import sys

import numpy as np

from lumixengine_amd import api, scenes
from tests import helpers as H

BONES = [1, 2, 3, 17, 52, 63, 64, 65, 100, 127, 128, 129, 195, 196]
VERTS = [1, 2, 63, 64, 65, 511, 512, 513, 1023, 1025, 2048, 5119, 5120, 5121, 7000, 10241]


def close_1e5(got, want):
    got = np.asarray(got, np.float64).reshape(-1, 3)
    want = np.asarray(want, np.float64).reshape(-1, 3)
    scale = np.maximum(np.abs(want).max(axis=1), 1e-2 * float(np.abs(want).max()))
    return bool((np.abs(got - want).max(axis=1) <= 1e-5 * scale).all())


def run(seed: int, oracle, ctx=None, verbose: bool = False) -> dict:
    rng = np.random.default_rng(7000 + seed)
    own = ctx is None
    if own:
        ctx = api.Context(0)
    try:
        sk = api.Skinning(ctx)
        exact = bool(rng.random() < 0.5)
        sk.setMode(exact)
        n_models = int(rng.integers(1, 5))
        skel = [scenes.skeleton(int(rng.choice(BONES)), seed=int(rng.integers(1 << 30))) for _ in range(n_models)]
        models = [sk.addModel(s["parents"], s["bind"], s["first_nonroot"] if len(s["parents"]) > 1 else -1) for s in skel]
        meshes, mesh_ids, mesh_bones = [], [], []
        for _ in range(int(rng.integers(1, 6))):
            nb_model = len(skel[int(rng.integers(0, n_models))]["parents"])
            nv = int(rng.choice(VERTS))
            kind = rng.random()
            if kind < 0.4:  # every vertex picks 4 of all the model's bones
                nb = nb_model
                v, s = scenes.skinned_mesh(nv, nb, seed=int(rng.integers(1 << 30)))
            elif kind < 0.7:  # only a few bones, anywhere in the skeleton
                nb = nb_model
                v, s = scenes.skinned_mesh(nv, nb, seed=int(rng.integers(1 << 30)))
                used = rng.choice(nb, size=min(nb, int(rng.integers(1, 9))), replace=False)
                s["indices"] = used[rng.integers(0, len(used), size=(nv, 4))].astype(np.int16)
            else:  # limb by limb
                nb = nb_model
                v, s = scenes.skinned_mesh_character(nv, nb, seed=int(rng.integers(1 << 30)), run=int(rng.choice([7, 190, 900])))
            meshes.append((v, s))
            mesh_bones.append(int(s["indices"].max()) + 1)
            mesh_ids.append(sk.addMesh(v, s))
        # instance table: runs and singles of compatible (model, mesh) pairs
        pick = []
        while len(pick) < int(rng.integers(1, 60)):
            g = int(rng.integers(0, len(meshes)))
            ok = [m for m in range(n_models) if len(skel[m]["parents"]) >= mesh_bones[g]]
            if not ok:
                continue
            m = int(rng.choice(ok))
            pick += [(m, g)] * int(rng.choice([1, 1, 2, 3, 9, 20]))
        sk.setInstances([models[m] for m, _ in pick], [mesh_ids[g] for _, g in pick])
        per_block = int(np.random.default_rng(9000 + seed).choice([0, 0, 1, 2, 4, 4, 8, 16]))  # (a stream of its own: the scenes of a seed stay what they were)
        sk.setOption(api.SKIN_OPT_INSTANCES_PER_BLOCK, per_block)
        poses = [scenes.relative_poses(1, len(skel[m]["parents"]), seed=int(rng.integers(1 << 30))) for m, _ in pick]
        sk.uploadPoses(np.concatenate([p[0].reshape(-1, 3) for p in poses]), np.concatenate([p[1].reshape(-1, 4) for p in poses]))
        sk.run()
        inv = [oracle.invert_bind(s["bind"]) for s in skel]
        checked = 0
        for i in sorted(set(rng.integers(0, len(pick), size=min(len(pick), 12)).tolist() + [0, len(pick) - 1])):
            m, g = pick[i]
            s = skel[m]
            fn = s["first_nonroot"] if len(s["parents"]) > 1 else 1
            apos, arot = oracle.pose_compute_absolute(poses[i][0], poses[i][1], s["parents"], fn)
            pal = oracle.skin_matrices(apos, arot, inv[m])
            assert H.bits_equal(sk.readPalette(i), pal[0]), f"seed {seed}: instance {i} palette"
            want = oracle.evaluate_skin(meshes[g][0], meshes[g][1], pal)[0]
            got = sk.readVertices(i)
            assert close_1e5(got, want), f"seed {seed}: instance {i} (model {m}: {len(s['parents'])} bones, mesh {g}: {len(meshes[g][0])} vertices) vertices"
            if exact:
                assert H.bits_equal(got, want), f"seed {seed}: instance {i} vertices (exact mode)"
            checked += 1
        sk.setMode(False)
        sk.setOption(api.SKIN_OPT_INSTANCES_PER_BLOCK, api.SKIN_INSTANCES_PER_BLOCK_DEFAULT)
        st = {"instances": len(pick), "checked": checked, "exact": exact, "per_block": per_block, "bones": [len(s["parents"]) for s in skel], "verts": [len(v) for v, _ in meshes]}
        if verbose:
            print(f"seed {seed}: {st}")
        return st
    finally:
        if own:
            ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0-9")
    a = ap.parse_args()
    lo, _, hi = a.seeds.partition("-")
    from oracle import pyoracle

    oracle = pyoracle.Oracle("port")
    for seed in range(int(lo), int(hi or lo) + 1):
        run(seed, oracle, verbose=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
