"""Parity of the HIP cull path (through the C ABI) against the CPU oracle and the golden fixtures. Needs an MI355X."""
import json
import os
import time

import numpy as np
import pytest

from lumixengine_amd import api, scenes
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gpu_visible(res, frustum):
    ids, types = res.all_ids(frustum)
    return H.sorted_by_type(ids, types)


def oracle_visible(cs, frustum, type_=0xFF):
    ids, types, _ = cs.cull(frustum, type_, n_threads=8)
    return H.sorted_by_type(ids, types)


@pytest.mark.parametrize("fixture", ["cull_edge.npz", "cull_mixed.npz"])
def test_cull_matches_golden(gpu_ctx, fixture):
    g = np.load(os.path.join(G, fixture))
    cs = api.CullingSystem(gpu_ctx)
    cs.build(g["entity"], g["type"], g["pos"], g["radius"])
    frusta = g["frusta"]
    # one frustum per call, then all of them in passes of up to 8 frusta
    for f in range(len(frusta)):
        res = cs.cull(np.ascontiguousarray(frusta[f : f + 1]))
        H.assert_same_visible(gpu_visible(res, 0), H.sorted_by_type(g[f"vis_ids_{f}"], g[f"vis_types_{f}"]), f"{fixture} single {f}")
    for start in range(0, len(frusta), 8):
        batch = np.ascontiguousarray(frusta[start : start + 8])
        res = cs.cull(batch, view=1)
        for k in range(len(batch)):
            f = start + k
            H.assert_same_visible(gpu_visible(res, k), H.sorted_by_type(g[f"vis_ids_{f}"], g[f"vis_types_{f}"]), f"{fixture} batch {f}")
    # every pass width 1..8 is its own kernel instantiation (the default is one frustum per pass)
    try:
        for width in range(1, min(8, len(frusta)) + 1):
            cs.setPassWidth(width)
            res = cs.cull(np.ascontiguousarray(frusta[:8]), view=2)
            for k in range(min(8, len(frusta))):
                H.assert_same_visible(gpu_visible(res, k), H.sorted_by_type(g[f"vis_ids_{k}"], g[f"vis_types_{k}"]), f"width {width} frustum {k}")
    finally:
        cs.setPassWidth(0)  # back to the default (automatic)


def test_cull_config1_golden(gpu_ctx):
    """BASELINE config 1: 100 k static entities, bit-exact visible list vs the reference CPU path."""
    g = np.load(os.path.join(G, "cull_config1.npz"))
    sc = scenes.cull_scene(100_000, 3000.0, seed=1)
    cs = api.CullingSystem(gpu_ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    assert cs.stats()["cells"] == int(g["n_cells"][0])
    frusta = g["frusta"]
    res = cs.cull(np.ascontiguousarray(frusta))
    for f in range(len(frusta)):
        H.assert_same_visible(gpu_visible(res, f), H.sorted_by_type(g[f"vis_ids_{f}"], g[f"vis_types_{f}"]), H.CAMERAS[f][0])
        # page split of the adapter: 1020 ids per CullResult page
        pages = res.pages(f, 0)
        assert all(len(p) == 1020 for p in pages[:-1]) and sum(len(p) for p in pages) == len(g[f"vis_ids_{f}"])


def test_cull_both_kernel_forms_match_golden(gpu_ctx):
    """The two forms of the 1-frustum kernel (LMX_CULL_OPT_TILE_VARIANT 1 = streaming, 4 = all loads in flight; -1 picks by how much of the set
    the frustum covers) on the config-1 fixture: rejected, accepted, dense and mixed tiles all occur under its cameras. The tile shapes of rounds
    2-5 that no rule ever selected are gone: asking for one is an error."""
    g = np.load(os.path.join(G, "cull_config1.npz"))
    sc = scenes.cull_scene(100_000, 3000.0, seed=1)
    cs = api.CullingSystem(gpu_ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    frusta = g["frusta"]
    try:
        for variant in (1, 4, -1):
            cs.setOption(api.CULL_OPT_TILE_VARIANT, variant)
            for f in range(len(frusta)):
                res = cs.cull(np.ascontiguousarray(frusta[f : f + 1]))
                H.assert_same_visible(gpu_visible(res, 0), H.sorted_by_type(g[f"vis_ids_{f}"], g[f"vis_types_{f}"]), f"variant {variant} camera {f}")
    finally:
        cs.setOption(api.CULL_OPT_TILE_VARIANT, -1)
    for gone in (0, 2, 3, 5):
        with pytest.raises(api.LumixError):
            cs.setOption(api.CULL_OPT_TILE_VARIANT, gone)
    with pytest.raises(api.LumixError):
        cs.setOption(1, 0)  # (the retired tile-level-test option)


def test_cull_cell_key_formats(gpu_ctx, oracle_port, monkeypatch):
    """The per-tile cell tables carry 8-byte keys relative to the tile's box wherever every tile spans <= 65535 cell indices per axis, 16-byte keys
    otherwise (the 1-frustum kernels: separate instantiations; the several-frusta kernels: a launch-uniform branch). The same scene under both
    (LMX_CULL_WIDE_KEYS forces the wide form at build time) against the config-1 fixture, one frustum and eight in one pass, both 1-frustum forms; and a
    scene that is wide by itself - two clusters 4 x 10^7 units apart whose spheres share tiles - against the oracle."""
    g = np.load(os.path.join(G, "cull_config1.npz"))
    sc = scenes.cull_scene(100_000, 3000.0, seed=1)
    frusta = g["frusta"]
    for wide in (False, True):
        if wide:
            monkeypatch.setenv("LMX_CULL_WIDE_KEYS", "1")
        cs = api.CullingSystem(gpu_ctx)
        cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
        info = cs.layoutInfo()
        assert info["cell_key_bytes"] == (16 if wide else 8) and info["table_bytes"] > 0
        try:
            for variant in (1, 4):
                cs.setOption(api.CULL_OPT_TILE_VARIANT, variant)
                for f in range(len(frusta)):
                    res = cs.cull(np.ascontiguousarray(frusta[f : f + 1]))
                    H.assert_same_visible(gpu_visible(res, 0), H.sorted_by_type(g[f"vis_ids_{f}"], g[f"vis_types_{f}"]), f"wide {wide} variant {variant} camera {f}")
            cs.setPassWidth(8)
            res = cs.cull(np.ascontiguousarray(frusta[:8]), view=1)
            for k in range(min(8, len(frusta))):
                H.assert_same_visible(gpu_visible(res, k), H.sorted_by_type(g[f"vis_ids_{k}"], g[f"vis_types_{k}"]), f"wide {wide} 8 frusta, frustum {k}")
        finally:
            cs.setOption(api.CULL_OPT_TILE_VARIANT, -1)
            cs.setPassWidth(0)
    monkeypatch.delenv("LMX_CULL_WIDE_KEYS")
    # wide by itself: 3000 spheres around x = -2e7 and 3000 around x = +2e7 (cell indices +-66 667: one 2048-sphere tile holds cells of both clusters)
    rng = np.random.default_rng(5)
    n = 6000
    pos = rng.uniform(-900.0, 900.0, size=(n, 3))
    pos[: n // 2, 0] -= 2.0e7
    pos[n // 2 :, 0] += 2.0e7
    radius = rng.uniform(0.5, 40.0, n).astype(np.float32)
    ent, types = np.arange(n, dtype=np.int32), np.zeros(n, np.uint8)
    cs = api.CullingSystem(gpu_ctx)
    cs.build(ent, types, pos, radius)
    assert cs.layoutInfo()["cell_key_bytes"] == 16
    ocs = oracle_port.culling_system()
    ocs.add_bulk(ent, types, pos, radius)
    cams = np.concatenate([api.viewport_frustum(pos=(-2.0e7, 0.0, 2500.0), far=6000.0), api.viewport_frustum(pos=(2.0e7 + 300.0, 100.0, 2000.0), far=4000.0),
                           api.viewport_frustum(pos=(0.0, 0.0, 0.0), far=5.0e7, rot=(0.0, 0.70710678, 0.0, 0.70710678))])
    seen = 0
    for f in range(len(cams)):
        res = cs.cull(cams[f : f + 1])
        want = oracle_visible(ocs, cams[f : f + 1])
        H.assert_same_visible(gpu_visible(res, 0), want, f"wide scene camera {f}")
        seen += sum(len(v) for v in want.values()) if isinstance(want, dict) else len(want)
    assert seen > 500
    res = cs.cull(cams, view=1)
    for f in range(len(cams)):
        H.assert_same_visible(gpu_visible(res, f), oracle_visible(ocs, cams[f : f + 1]), f"wide scene, one pass, camera {f}")


def test_cull_type_filter_and_views(gpu_ctx, oracle_port):
    sc = H.mixed_scene()
    cs = api.CullingSystem(gpu_ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    ocs = oracle_port.culling_system()
    ocs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    fr = H.frusta(api)
    results = [cs.cull(fr[f : f + 1], view=f) for f in range(len(fr))]  # 7 views in flight, read afterwards
    for f, res in enumerate(results):
        H.assert_same_visible(gpu_visible(res, 0), oracle_visible(ocs, fr[f : f + 1]), f"view {f}")
    for t in (0, 1, 2):
        res = cs.cull(fr[:2], type_=t)
        for f in range(2):
            H.assert_same_visible(gpu_visible(res, f), oracle_visible(ocs, fr[f : f + 1], t), f"type {t}")
            assert res.counts()[f].sum() == res.counts()[f, t]


def test_cull_empty_and_tiny(gpu_ctx):
    cs = api.CullingSystem(gpu_ctx)
    cs.build(np.zeros(0, np.int32), np.zeros(0, np.uint8), np.zeros((0, 3)), np.zeros(0, np.float32))
    fr = H.frusta(api)[:1]
    assert cs.cull(fr).count(0) == 0  # the reference returns nullptr (culling_system.cpp:322)
    cs.add(5, 3, (0.0, 0.0, -10.0), 1.0)
    res = cs.cull(fr)
    assert res.counts()[0, 3] == 1 and res.ids(0, 3)[0] == 5
    assert cs.isAdded(5) and not cs.isAdded(4) and cs.getRadius(5) == 1.0
    cs.remove(5)
    assert cs.cull(fr).count(0) == 0


def test_cull_incremental_ops(gpu_ctx, live_oracle):
    """CullingSystem::add/remove/set/setPosition/setRadius (culling_system.cpp:131-258) against the oracle."""
    oracle_port = live_oracle
    rng = np.random.default_rng(21)
    sc = H.mixed_scene(4000, 1200.0, seed=5)
    cs = api.CullingSystem(gpu_ctx)
    ocs = oracle_port.culling_system()
    n0 = 3000
    cs.build(sc["entity"][:n0], sc["type"][:n0], sc["pos"][:n0], sc["radius"][:n0])
    ocs.add_bulk(sc["entity"][:n0], sc["type"][:n0], sc["pos"][:n0], sc["radius"][:n0])
    alive = set(int(e) for e in sc["entity"][:n0])
    pending = list(range(n0, 4000))
    fr = H.frusta(api, names=["origin_identity", "origin_yaw_pitch"])
    for step in range(1500):
        op = rng.integers(0, 6)
        if op == 0 and pending:
            i = pending.pop()
            for c in (cs, ocs):
                c.add(sc["entity"][i], sc["type"][i], sc["pos"][i], sc["radius"][i])
            alive.add(int(sc["entity"][i]))
        elif op == 1 and len(alive) > 10:
            e = int(rng.choice(sorted(alive)))
            cs.remove(e)
            ocs.remove(e)
            alive.discard(e)
        elif alive:
            e = int(rng.choice(sorted(alive)))
            pos = rng.uniform(-1500, 1500, 3)
            r = float(rng.choice([rng.uniform(0.5, 60.0), rng.uniform(280.0, 330.0), 300.0]))
            if op == 2:
                cs.set(e, pos, r)
                ocs.set(e, pos, r)
            elif op == 3:
                cs.setPosition(e, pos)
                ocs.set_position(e, pos)
            else:
                cs.setRadius(e, r)
                ocs.set_radius(e, r)
            assert cs.getRadius(e) == ocs.get_radius(e)
        if step % 100 == 99:
            res = cs.cull(fr)
            for f in range(len(fr)):
                H.assert_same_visible(gpu_visible(res, f), oracle_visible(ocs, fr[f : f + 1]), f"step {step}")
    assert cs.stats()["cells"] == ocs.cell_count()


@pytest.mark.parametrize("variant", ["sparse", "dense"])
def test_cull_1m_vs_oracle(gpu_ctx, oracle_port, variant):
    """1 M spheres (a tenth of BASELINE config 2) compared id for id with the 8-thread CPU oracle."""
    half = 7000.0 if variant == "sparse" else 2300.0
    sc = scenes.cull_scene(1_000_000, half, seed=3)
    cs = api.CullingSystem(gpu_ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    ocs = oracle_port.culling_system()
    ocs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    fr = H.frusta(api, names=["origin_identity", "far_camera", "ortho_cascade_large"])
    res = cs.cull(fr)
    for f in range(len(fr)):
        H.assert_same_visible(gpu_visible(res, f), oracle_visible(ocs, fr[f : f + 1]), f"{variant}/{f}")


@pytest.mark.parametrize("order", ["grouped", "interleaved"])
def test_cull_cascades_with_shared_normals(gpu_ctx, oracle_port, order):
    """Shadow cascades of one light share its rotation: frusta of a call whose plane normals are bitwise identical share a sphere's dot
    products in k_cull_tile<F = 0> (pipeline.cpp:734-827 builds a light's cascades from one rotation). 8 cascades of 2 lights in one pass,
    the lights' cascades adjacent or interleaved (the shared products must be recomputed at every change of normals), plus big spheres
    (every cell CELL_TEST) - id for id against the oracle, per frustum."""
    sc = scenes.cull_scene(300_000, 3000.0, seed=9, big_fraction=0.05)
    cs = api.CullingSystem(gpu_ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    ocs = oracle_port.culling_system()
    ocs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    kws = scenes.config5_cascade_kwargs()
    for kw in kws:  # the bench's cascades look at a +-15000 cube from 9000 above: the same shapes over this scene
        kw["pos"] = (kw["pos"][0], 2500.0, kw["pos"][2])
        kw["ortho_size"] = kw["ortho_size"] / 4.0
    if order == "interleaved":
        kws = [kws[k] for k in (0, 4, 1, 5, 2, 6, 3, 7)]
    fr = np.concatenate([api.viewport_frustum(**kw) for kw in kws])
    cs.setPassWidth(8)
    res = cs.cull(fr)
    total = 0
    for f in range(8):
        got = gpu_visible(res, f)
        H.assert_same_visible(got, oracle_visible(ocs, fr[f : f + 1]), f"{order}/{f}")
        total += len(res.all_ids(f)[0])
    assert total > 1000
    cs.setPassWidth(1)


@pytest.mark.parametrize("seed", [31, 32])
def test_cull_pretest_undecided_band(gpu_ctx, oracle_port, seed):
    """The several-frusta kernel decides most (sphere, frustum) pairs from a bf16 MFMA evaluation of the six plane expressions and leaves the
    pairs within its error bound of zero to the exact loop (cull_kernels.hip, "sphere x plane pre-test"). Here every sphere is tangent, up
    to a perturbation between 0 and 1e-2, to one plane of one of config 5's eight cascades (culling_system.cpp:283-306: `t + r < 0` culls -
    the sign of a sum within a few ulps of zero): the centre lies outside the plane at a distance rho, the radius is rho + delta. Wrong by
    one rounding anywhere and an id is missing or extra; every frustum is compared with the oracle id for id."""
    rng = np.random.default_rng(seed)
    kws = scenes.config5_cascade_kwargs()
    for kw in kws:
        kw["pos"] = (kw["pos"][0], 2500.0, kw["pos"][2])
        kw["ortho_size"] = kw["ortho_size"] / 4.0
    fr = np.concatenate([api.viewport_frustum(**kw) for kw in kws])
    n = 120_000
    f_of = rng.integers(0, 8, n)
    k_of = rng.integers(0, 6, n)
    nrm = np.stack([fr["xs"][f_of, k_of], fr["ys"][f_of, k_of], fr["zs"][f_of, k_of]], 1).astype(np.float64)
    origin = fr["origin"][f_of].astype(np.float64)
    # a point of the plane: n . p + d = 0 in the frustum's frame (ShiftedFrustum: planes relative to origin)
    on_plane = origin - nrm * (fr["ds"][f_of, k_of].astype(np.float64) / (nrm * nrm).sum(1))[:, None]
    tangent = np.cross(nrm, rng.normal(size=(n, 3)))
    tangent /= np.linalg.norm(tangent, axis=1, keepdims=True)
    rho = np.exp(rng.uniform(np.log(0.5), np.log(60.0), n))
    pos = on_plane + tangent * rng.uniform(-800.0, 800.0, (n, 1)) - nrm * (rho / np.linalg.norm(nrm, axis=1))[:, None]
    delta = rng.choice([0.0, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4, 1e-3, -1e-3, 1e-2, -1e-2], n) * rng.uniform(0.0, 1.0, n)
    radius = (rho + delta).astype(np.float32)
    entity = np.arange(n, dtype=np.int32)
    types = (entity % 3).astype(np.uint8)
    cs = api.CullingSystem(gpu_ctx)
    cs.build(entity, types, pos, radius)
    ocs = oracle_port.culling_system()
    ocs.add_bulk(entity, types, pos, radius)
    cs.setPassWidth(8)
    res = cs.cull(fr)
    seen = 0
    for f in range(8):
        got = gpu_visible(res, f)
        H.assert_same_visible(got, oracle_visible(ocs, fr[f : f + 1]), f"tangent spheres, frustum {f}")
        seen += len(res.all_ids(f)[0])
    assert 0.05 * n < seen < 7.5 * n  # neither everything nor nothing: the tangent plane decides
    cs.setPassWidth(1)


def _digest(res, frustum=0):
    ids, types = res.all_ids(frustum)
    return H.visible_digest(ids, types)


@pytest.mark.parametrize("scene", list(H.CONFIG2_SCENES))
def test_cull_config2_10m_bit_exact(gpu_ctx, scene):
    """BASELINE config 2 at full size (10 M entities, sparse with mixed types / dense): the visible ids of every camera, every tile
    variant of the 1-frustum kernel, 8 cascade frusta in passes of width 1 / 4 / 8 and an update stream of 20 frames x (1000
    removes + 1000 adds + 500 sets) are compared with the reference CPU path through committed digests (per-type counts +
    sha256 of the sorted id lists, tests/golden/make_golden_10m.py: a 10 M oracle run costs minutes and 4 GB of result pages)."""
    g = json.load(open(os.path.join(G, "cull_10m.json")))
    n, rec = g["n"], g["scenes"][scene]
    half, mixed = H.CONFIG2_SCENES[scene]
    sc = scenes.cull_scene(n, half, seed=2, mixed_types=mixed)
    assert H.array_digest(sc["entity"], sc["type"], sc["pos"], sc["radius"]) == rec["scene_sha"], "the scene generator's random stream differs from the one the digests were made with"
    cs = api.CullingSystem(gpu_ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    # the oracle counts 4 KiB cell PAGES (<= 200 spheres each); the device layout counts (cell, type, is_big) groups
    assert cs.stats()["cells"] == rec["cells"] if mixed else cs.stats()["cells"] <= rec["cells"]
    cams = H.config2_cameras(api)

    def check(res, want, what, frustum=0):
        counts, sha = _digest(res, frustum)
        assert counts == want["counts"], f"{scene} {what}: visible counts {counts} vs reference {want['counts']}"
        assert sha == want["sha256"], f"{scene} {what}: same counts, different ids"

    for cam, fr in cams:
        check(cs.cull(fr), rec["cameras"][cam], cam)
    try:
        for variant in (1, 4):  # the two forms of the 1-frustum kernel
            cs.setOption(api.CULL_OPT_TILE_VARIANT, variant)
            for cam, fr in cams:
                if cam in ("default", "narrow", "all_visible"):
                    check(cs.cull(fr, view=1), rec["cameras"][cam], f"{cam} variant {variant}")
        cs.setOption(api.CULL_OPT_TILE_VARIANT, -1)
        for shards, pad in ((1, 1), (7, 16), (64, 32)):
            cs.setOption(api.CULL_OPT_MAX_SHARDS, shards)
            cs.setOption(api.CULL_OPT_COUNTER_PAD, pad)
            check(cs.cull(cams[4][1], view=2), rec["cameras"]["all_visible"], f"{shards} shards, pad {pad}")
            check(cs.cull(cams[0][1], view=2), rec["cameras"]["default"], f"{shards} shards, pad {pad}")
    finally:
        cs.setOption(api.CULL_OPT_TILE_VARIANT, -1)
        cs.setOption(api.CULL_OPT_MAX_SHARDS, 64)
        cs.setOption(api.CULL_OPT_COUNTER_PAD, 32)
    if not mixed:
        return
    fr8 = H.cascade_frusta(api, 8)
    try:
        for width in (1, 4, 8):
            cs.setPassWidth(width)
            res = cs.cull(fr8)
            for k in range(8):
                check(res, rec["cascades"][k], f"cascade {k}, pass width {width}", frustum=k)
    finally:
        cs.setPassWidth(0)  # back to the default (automatic)
    # update stream: O(1) patches, no rebuild of the sorted set (the overflow stays far below the compaction threshold)
    fr = cams[0][1]
    for _ in range(3):
        cs.cull(fr)
    gpu_ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        cs.cull(fr)
    gpu_ctx.synchronize()
    t_cull = (time.perf_counter() - t0) / 20
    t_frames = []
    for f, ops in enumerate(H.churn_stream(sc["pos"], half, 20, 1000)):
        t0 = time.perf_counter()
        cs.removeMany(ops["remove"])
        cs.addMany(ops["add_ids"], ops["add_type"], ops["add_pos"], ops["add_radius"])
        cs.setMany(ops["set_ids"], ops["set_pos"], ops["set_radius"])
        res = cs.cull(fr)
        gpu_ctx.synchronize()
        t_frames.append(time.perf_counter() - t0)
        if str(f) in rec["churn"]:
            check(res, rec["churn"][str(f)], f"update stream frame {f}")
    st = cs.updateStats()
    # nothing was rebuilt: 20 x 1000 adds + the sets that left their cell (every other one by construction, a few more by chance)
    assert 25_000 <= st["overflow"] <= 25_500 and st["tombstones"] == st["overflow"], st
    assert H.TIMELESS or np.median(t_frames) - t_cull < 2e-3, f"2500 updates add {1e6 * (np.median(t_frames) - t_cull):.0f} us to a {1e6 * t_cull:.0f} us cull"
    # a compaction folds everything back into the sorted layout: same visible set
    cs.compact()
    st = cs.updateStats()
    assert st["overflow"] == 0 and st["tombstones"] == 0
    check(cs.cull(fr), rec["churn"]["19"], "after compaction")


def _bench_scene(name):
    """(record, scene arrays) of one of the scenes bench.py times, regenerated from its seed and checked against the digest of the
    arrays the reference culled (tests/golden/make_golden_bench_scenes.py)."""
    rec = json.load(open(os.path.join(G, "cull_bench_scenes.json")))["scenes"][name]
    if name.startswith("slab"):
        sc = scenes.slab_scene(rec["n"], seed=rec["seed"])
    else:
        sc = scenes.cull_scene(rec["n"], scenes.scaled_half_extent(rec["n"]), seed=rec["seed"], mixed_types=rec["mixed"])
    if rec["all_test_radii"]:
        sc["radius"] = scenes.all_test_radii(rec["n"])
    assert H.array_digest(sc["entity"], sc["type"], sc["pos"], sc["radius"]) == rec["scene_sha"], "the scene generator's random stream differs from the one the digests were made with"
    return rec, sc


def _check_digest(res, want, what, frustum=0):
    counts, sha = _digest(res, frustum)
    assert counts == want["counts"], f"{what}: visible counts {counts} vs reference {want['counts']}"
    assert sha == want["sha256"], f"{what}: same counts, different ids"


@pytest.mark.parametrize("name", ["sparse_10m", "all_test_10m"])
def test_cull_bench_scenes_10m_digest(gpu_ctx, name):
    """The EXACT 10 M scenes bench.py times - its headline scene (sparse, one renderable type, seed 2) and its roofline leg
    (the same positions, every sphere "big": every sphere fetched and tested) - against the reference's own CullingSystemImpl
    (digests in tests/golden/cull_bench_scenes.json): the ids, not just the counts bench.py used to print."""
    rec, sc = _bench_scene(name)
    cs = api.CullingSystem(gpu_ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    cams = dict(H.config2_cameras(api))
    for cam, want in rec["cameras"].items():
        _check_digest(cs.cull(cams[cam]), want, f"{name} {cam}")
    # what bench.py itself asserts: the digest regardless of type
    ids, _ = cs.cull(cams["default"]).all_ids(0)
    assert H.ids_digest(ids) == rec["cameras"]["default"]["all_types_sha256"]


def test_cull_slab_all_cell_test_digest(gpu_ctx):
    """bench.py's all-CELL_TEST leg with NORMAL radii: 10 M spheres in one layer of cells under an orthographic slab camera that every
    cell straddles, so every cell is classified CELL_TEST by the AABB pre-tests (culling_system.cpp:342-363) and every sphere is
    tested; 4.34 M of them are visible. Ids against the reference's CullingSystemImpl, for every tile variant of the kernel."""
    rec, sc = _bench_scene("slab_10m")
    cs = api.CullingSystem(gpu_ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    assert cs.stats()["cells"] == rec["cells"]
    fr = api.viewport_frustum(**scenes.slab_frustum_kwargs(sc["half"]))
    try:
        for variant in (-1, 1, 4):
            cs.setOption(api.CULL_OPT_TILE_VARIANT, variant)
            _check_digest(cs.cull(fr), rec["cameras"]["slab"], f"slab variant {variant}")
    finally:
        cs.setOption(api.CULL_OPT_TILE_VARIANT, -1)


@pytest.mark.parametrize("name", ["config5_100m", "all_test_100m"])
def test_cull_config5_100m_digest(gpu_ctx, name):
    """BASELINE config 5's size on one GPU: 100 M entities (2 GB of spheres + ids), mixed renderable types 90 / 5 / 5. The default
    camera and the 8 shadow-cascade frusta of one frame - as one lmx_cull call in passes of width 1, 4 and 8 (one pass over the
    spheres tests all 8 frusta, culling_system.cpp:321-369 run 8 times in the reference) - against digests the reference's own
    CullingSystemImpl produced for the same seeded scene, culled there in 16 shards of whole cells (an entity's visibility depends
    only on its own cell; tests/golden/make_golden_bench_scenes.py says why and how). all_test_100m: the HBM-cold-by-size roofline
    extra of bench.py (every sphere fetched and tested)."""
    rec, sc = _bench_scene(name)
    cs = api.CullingSystem(gpu_ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    del sc
    _check_digest(cs.cull(api.viewport_frustum()), rec["cameras"]["default"], f"{name} default")
    if "cascade0" not in rec["cameras"]:
        return
    fr8 = np.concatenate([api.viewport_frustum(**kw) for kw in scenes.config5_cascade_kwargs()])
    try:
        for width in (1, 4, 8):
            cs.setPassWidth(width)
            res = cs.cull(fr8, view=1)
            for k in range(8):
                _check_digest(res, rec["cameras"][f"cascade{k}"], f"{name} cascade {k}, pass width {width}", frustum=k)
    finally:
        cs.setPassWidth(0)  # back to the default (automatic)


def test_cull_add_stream_never_stalls(gpu_ctx, oracle_port):
    """The reference's add / remove never stall (culling_system.cpp:131-190). With LMX_CULL_OPT_OVERFLOW_RESERVE sized for the churn and
    LMX_CULL_OPT_AUTO_COMPACTION off, a stream of adds that grows the set by 20 % (2 M entities, 400 frames x 1000 adds, every frame
    culled) takes free overflow slots only: no re-layout, no re-upload, no frame beyond a few hundred microseconds - and the visible
    set stays the oracle's (checked at 1/5 of the stream and at its end, when 400 k entities sit in the unsorted overflow set)."""
    n, half, frames, per = 2_000_000, 8800.0, 400, 1000
    sc = scenes.cull_scene(n, half, seed=21)
    cs = api.CullingSystem(gpu_ctx)
    try:
        cs.setOption(api.CULL_OPT_AUTO_COMPACTION, 0)
        cs.setOption(api.CULL_OPT_OVERFLOW_RESERVE, frames * per + 50_000)
        cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
        ocs = oracle_port.culling_system()
        ocs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
        fr = api.viewport_frustum()
        rng = np.random.default_rng(8)
        for _ in range(5):
            cs.cull(fr)
        gpu_ctx.synchronize()
        t_frames = []
        for f in range(frames):
            ids = np.arange(n + f * per, n + (f + 1) * per, dtype=np.int32)
            pos = rng.uniform(-half, half, size=(per, 3))
            rad = np.exp(rng.uniform(np.log(0.5), np.log(50.0), size=per)).astype(np.float32)
            typ = np.zeros(per, np.uint8)
            ocs.add_bulk(ids, typ, pos, rad)
            t0 = time.perf_counter()
            cs.addMany(ids, typ, pos, rad)
            res = cs.cull(fr)
            gpu_ctx.synchronize()
            t_frames.append(time.perf_counter() - t0)
            if f in (frames // 5, frames - 1):
                H.assert_same_visible(gpu_visible(res, 0), oracle_visible(ocs, fr), f"frame {f}")
        st = cs.updateStats()
        assert st["overflow"] == frames * per and st["tombstones"] == 0, st
        t = np.array(t_frames[3:])
        # a re-layout of this set costs > 100 ms (0.5 s at 10 M entities) and a re-upload tens of ms in EVERY frame; one frame of 400 may catch a
        # scheduling hiccup of the host (20 ms seen once on a shared 256-core box): the slowest frame stays far below a re-layout, all others below 2 ms
        ts = np.sort(t)
        assert H.TIMELESS or (ts[-1] < 50e-3 and ts[-2] < 2e-3), f"slowest frames {1e3 * ts[-1]:.2f} / {1e3 * ts[-2]:.2f} ms (median {1e6 * np.median(t):.0f} us): an add stalled"
    finally:
        cs.setOption(api.CULL_OPT_OVERFLOW_RESERVE, 0)
        cs.setOption(api.CULL_OPT_AUTO_COMPACTION, 1)


def test_cull_async_compaction_stream(gpu_ctx, oracle_port):
    """LMX_CULL_OPT_ASYNC_COMPACTION: the re-sort of the static set on a worker thread, against a second copy of the sets, traded with
    the live one inside a flush. A 600 k-entity scene takes 2000 adds + 500 removes + 500 moves per frame (the thresholds of the
    automatic compaction are crossed every ~40 frames) with a cull per frame; the visible set is compared with the oracle every few
    frames - in particular right after every swap - several sets must have been traded, the update thread must have replayed only a
    small tail of operations at each swap, and no frame may take anywhere near what a synchronous re-sort of this set costs."""
    n, half = 600_000, 6000.0
    sc = scenes.cull_scene(n, half, seed=31, mixed_types=True)
    cs = api.CullingSystem(gpu_ctx)
    try:
        cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
        t0 = time.perf_counter()
        cs.compact()  # nothing to do: measures nothing; the synchronous cost is measured below on a dirty set
        cs.setOption(api.CULL_OPT_ASYNC_COMPACTION, 1)
        assert cs.asyncStats()["state"] == 0
        ocs = oracle_port.culling_system()
        ocs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
        alive = list(range(n))
        pos_of = {}
        next_id = n
        rng = np.random.default_rng(5)
        cams = H.frusta(api, names=["origin_identity", "origin_yaw_pitch", "narrow_fov"])
        t_frames, swaps_seen, checked_after_swap = [], 0, 0
        frame = 0
        deadline = time.time() + 240
        while (swaps_seen < 3 or frame < 150) and time.time() < deadline:
            k_add, k_rm, k_mv = 2000, 500, 500
            ids = np.arange(next_id, next_id + k_add, dtype=np.int32)
            next_id += k_add
            p_add = rng.uniform(-half, half, size=(k_add, 3))
            r_add = np.exp(rng.uniform(np.log(0.5), np.log(60.0), size=k_add)).astype(np.float32)
            t_add = rng.integers(0, 3, k_add).astype(np.uint8)
            pick = rng.choice(len(alive), size=k_rm + k_mv, replace=False)
            rm = np.array([alive[i] for i in pick[:k_rm]], np.int32)
            mv = np.array([alive[i] for i in pick[k_rm:]], np.int32)
            p_mv = rng.uniform(-half, half, size=(k_mv, 3))
            r_mv = np.exp(rng.uniform(np.log(0.5), np.log(400.0), size=k_mv)).astype(np.float32)  # some cross the big-sphere threshold
            for i in sorted(pick[:k_rm].tolist(), reverse=True):
                alive[i] = alive[-1]
                alive.pop()
            alive.extend(ids.tolist())
            ocs.add_bulk(ids, t_add, p_add, r_add)
            for e in rm.tolist():
                ocs.remove(e)
            for j, e in enumerate(mv.tolist()):
                ocs.set(e, p_mv[j], float(r_mv[j]))
            fr = cams[frame % len(cams) : frame % len(cams) + 1]
            t1 = time.perf_counter()
            cs.addMany(ids, t_add, p_add, r_add)
            cs.removeMany(rm)
            cs.setMany(mv, p_mv, r_mv)
            res = cs.cull(fr)
            gpu_ctx.synchronize()
            t_frames.append(time.perf_counter() - t1)
            st = cs.asyncStats()
            assert st["state"] != 4, "the asynchronous compaction failed"
            swapped = st["swaps"] > swaps_seen
            swaps_seen = st["swaps"]
            if swapped or frame % 25 == 0:
                H.assert_same_visible(gpu_visible(res, 0), oracle_visible(ocs, fr), f"frame {frame} (swaps {swaps_seen})")
                checked_after_swap += 1 if swapped else 0
            frame += 1
            time.sleep(0.002)  # a frame of a real engine lasts milliseconds: the worker's catch-up must be able to outrun the update stream
        st = cs.asyncStats()
        assert st["swaps"] >= 3 and checked_after_swap >= 3, (st, frame)
        assert st["ops_replayed_at_swaps"] <= st["swaps"] * 40_000, st  # a few frames' worth per swap, not the build's whole backlog
        us = cs.updateStats()
        assert us["static"] + us["overflow"] == len(alive), (us, len(alive))
        # the synchronous re-sort of the same set, for scale
        cs.setOption(api.CULL_OPT_ASYNC_COMPACTION, 0)
        assert cs.asyncStats()["state"] == -1
        ids = np.arange(next_id, next_id + 100_000, dtype=np.int32)
        cs.addMany(ids, np.zeros(len(ids), np.uint8), rng.uniform(-half, half, size=(len(ids), 3)), np.ones(len(ids), np.float32))
        t1 = time.perf_counter()
        cs.compact()
        gpu_ctx.synchronize()
        t_sync = time.perf_counter() - t1
        t = np.array(t_frames[3:])
        print(f"async compaction: {frame} frames, {st['swaps']} swaps, {st['ops_replayed_at_swaps']} ops replayed at swaps; slowest frame {1e3 * t.max():.2f} ms, "
              f"median {1e3 * np.median(t):.3f} ms; synchronous re-sort {1e3 * t_sync:.1f} ms")
        assert H.TIMELESS or t.max() < 0.5 * t_sync, f"slowest frame {1e3 * t.max():.2f} ms vs a synchronous re-sort of {1e3 * t_sync:.1f} ms"
    finally:
        cs.setOption(api.CULL_OPT_ASYNC_COMPACTION, 0)


def test_cull_async_compaction_lifecycle(oracle_port):
    """The worker of LMX_CULL_OPT_ASYNC_COMPACTION against everything that can cut across a running job: a synchronous lmx_cull_compact,
    a fresh lmx_cull_build, switching the option off, and tearing the context down - each right after a job was requested. Results stay
    the oracle's, nothing hangs, nothing crashes."""
    ctx = api.Context(0)
    try:
        n, half = 400_000, 5000.0
        sc = scenes.cull_scene(n, half, seed=37)
        cs = api.CullingSystem(ctx)
        cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
        ocs = oracle_port.culling_system()
        ocs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
        fr = api.viewport_frustum()
        rng = np.random.default_rng(3)
        next_id = [n]

        def burst(k):  # enough adds to cross the compaction threshold (max(65536, n / 8)) in one go
            ids = np.arange(next_id[0], next_id[0] + k, dtype=np.int32)
            next_id[0] += k
            p = rng.uniform(-half, half, size=(k, 3))
            r = np.exp(rng.uniform(np.log(0.5), np.log(60.0), size=k)).astype(np.float32)
            ocs.add_bulk(ids, np.zeros(k, np.uint8), p, r)
            cs.addMany(ids, np.zeros(k, np.uint8), p, r)

        def same(what):
            H.assert_same_visible(gpu_visible(cs.cull(fr), 0), oracle_visible(ocs, fr), what)

        cs.setOption(api.CULL_OPT_ASYNC_COMPACTION, 1)
        burst(80_000)
        same("job requested")  # the flush inside this cull asks for the job
        assert cs.asyncStats()["state"] in (1, 2, 3)
        cs.compact()  # synchronous re-sort while the job runs: waits for it, discards it, re-seeds the shadow set
        st = cs.updateStats()
        assert st["overflow"] == 0 and cs.asyncStats()["state"] == 0
        same("after a synchronous compaction")
        burst(80_000)
        same("second job requested")
        sc2 = scenes.cull_scene(150_000, 3000.0, seed=38)  # a new scene while the job runs
        cs.build(sc2["entity"], sc2["type"], sc2["pos"], sc2["radius"])
        ocs = oracle_port.culling_system()
        ocs.add_bulk(sc2["entity"], sc2["type"], sc2["pos"], sc2["radius"])
        next_id[0] = 150_000
        same("after lmx_cull_build")
        burst(70_000)
        same("third job requested")
        deadline = time.time() + 60
        while cs.asyncStats()["swaps"] == 0 and time.time() < deadline:  # let this one finish: the swap happens inside a cull's flush
            same("waiting for the swap")
            time.sleep(0.005)
        assert cs.asyncStats()["swaps"] >= 1
        same("after the swap")
        burst(70_000)
        same("fourth job requested")
        # in-cell moves never make a re-sort due (they patch the sorted set in place): the operation log is bounded by drain jobs
        deadline = time.time() + 60
        while cs.asyncStats()["state"] != 0 and time.time() < deadline:
            same("waiting for the fourth job")
            time.sleep(0.005)
        drains0 = cs.asyncStats()["log_drains"]
        some = sc2["entity"][:50_000]
        base = sc2["pos"][:50_000]
        for k in range(30):  # 1.5 M set calls, every one a tiny move inside the entity's cell (or across its border: both are fine)
            p = base + rng.uniform(-0.01, 0.01, size=base.shape)
            cs.setMany(some, p, sc2["radius"][:50_000])
            cs.cull(fr)
            time.sleep(0.002)
        for j in range(len(some)):  # the oracle takes the final positions
            ocs.set(int(some[j]), p[j], float(sc2["radius"][j]))
        deadline = time.time() + 30
        while cs.asyncStats()["log_drains"] == drains0 and time.time() < deadline:
            cs.cull(fr)
            time.sleep(0.005)
        assert cs.asyncStats()["log_drains"] > drains0, cs.asyncStats()
        cs.setOption(api.CULL_OPT_ASYNC_COMPACTION, 0)  # off while a job may run
        assert cs.asyncStats()["state"] == -1
        same("option off")
        cs.setOption(api.CULL_OPT_ASYNC_COMPACTION, 1)
        burst(70_000)
        same("fifth job requested")
    finally:
        ctx.close()  # with a job possibly still running: the teardown joins the worker


def test_cull_10m_properties(gpu_ctx):
    """BASELINE config 2 size (10 M): size-independent properties next to the digest comparison above.

    * a frustum containing the whole scene returns every id exactly once;
    * culling is idempotent and independent of the batch a frustum is evaluated in;
    * visible(narrow frustum) is a subset of visible(the same camera with a wider fov and a longer far plane).
    """
    n = 10_000_000
    sc = scenes.cull_scene(n, 15000.0, seed=2)
    cs = api.CullingSystem(gpu_ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    everything = api.viewport_frustum(is_ortho=True, ortho_size=40000.0, w=1024, h=1024, near=0.0, far=100000.0, pos=(0.0, 0.0, 50000.0))
    narrow = api.viewport_frustum(pos=(100.0, 20.0, 300.0), rot=H.quat_from_yaw_pitch(0.5, -0.2), fov=float(np.deg2rad(40.0)), far=8000.0)
    wide = api.viewport_frustum(pos=(100.0, 20.0, 300.0), rot=H.quat_from_yaw_pitch(0.5, -0.2), fov=float(np.deg2rad(90.0)), far=20000.0)
    res = cs.cull(np.concatenate([everything, narrow, wide]))
    all_ids = res.ids(0, 0)
    assert len(all_ids) == n and np.array_equal(np.sort(all_ids), np.arange(n, dtype=np.int32))
    a, b = np.sort(res.ids(1, 0)), np.sort(res.ids(2, 0))
    assert 0 < len(a) < len(b) < n
    assert np.all(np.isin(a, b, assume_unique=True))
    again = cs.cull(narrow, view=1)
    assert np.array_equal(np.sort(again.ids(0, 0)), a)
    assert len(np.unique(a)) == len(a)


def test_cull_update_stream_vs_oracle(gpu_ctx, live_oracle):
    """The O(1) update path against the live oracle at 300 k entities: removals (tombstones), adds and cross-cell sets (overflow in
    the dynamic set), in-cell sets (sphere patches), a growing overflow region (slot reassignment), re-adding removed entities,
    and an explicit compaction in the middle of the stream."""
    oracle_port = live_oracle
    n, half = 300_000, 4500.0
    sc = scenes.cull_scene(n, half, seed=9, mixed_types=True)
    cs = api.CullingSystem(gpu_ctx)
    ocs = oracle_port.culling_system()
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    ocs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    # the last camera sees the whole scene: its tiles are accepted as a whole, and the ones that received a tombstone must have
    # lost their "straight copy" (TILE_DENSE) status
    fr = np.concatenate([H.frusta(api, names=["origin_identity", "origin_yaw_pitch", "ortho_cascade_large"]), api.viewport_frustum(pos=(0.0, 0.0, 30000.0), far=100000.0)])
    removed = []
    for f, ops in enumerate(H.churn_stream(sc["pos"], half, 12, 3000, seed=5)):
        cs.removeMany(ops["remove"])
        cs.addMany(ops["add_ids"], ops["add_type"], ops["add_pos"], ops["add_radius"])
        cs.setMany(ops["set_ids"], ops["set_pos"], ops["set_radius"])
        for e in ops["remove"]:
            ocs.remove(int(e))
        for i in range(len(ops["add_ids"])):
            ocs.add(int(ops["add_ids"][i]), int(ops["add_type"][i]), ops["add_pos"][i], float(ops["add_radius"][i]))
        for i in range(len(ops["set_ids"])):
            ocs.set(int(ops["set_ids"][i]), ops["set_pos"][i], float(ops["set_radius"][i]))
        removed.extend(int(e) for e in ops["remove"][:50])
        if f % 3 == 2:  # bring some removed entities back (they now live in the overflow) and touch overflow entities again
            back = removed[-100:]
            for e in back:
                cs.add(e, 0, sc["pos"][e], 7.5)
                ocs.add(e, 0, sc["pos"][e], 7.5)
            for e in ops["add_ids"][:40]:
                cs.setRadius(int(e), 301.0)
                ocs.set_radius(int(e), 301.0)
                cs.setPosition(int(e), (12.0, 3.0, -40.0))
                ocs.set_position(int(e), (12.0, 3.0, -40.0))
            for e in ops["add_ids"][40:60]:
                cs.remove(int(e))
                ocs.remove(int(e))
            removed = removed[:-100]
        if f == 6:
            cs.compact()
        res = cs.cull(fr)
        for k in range(len(fr)):
            H.assert_same_visible(gpu_visible(res, k), oracle_visible(ocs, fr[k : k + 1]), f"frame {f} frustum {k}")
    assert cs.stats()["cells"] == ocs.cell_count()


def test_cull_one_sphere_per_cell_layout_padding(gpu_ctx, oracle_port):
    """One sphere per cell: the layout spreads such runs over more 1024-slot blocks (at most 240 cells per block, dead
    padding in between) so the fused kernel's per-tile cell table always fits LDS. 8 frusta and 1 frustum vs the oracle."""
    g = np.arange(0, 28)  # non-negative: int() truncation makes cells straddle zero (two lattice points in cell 0)
    xx, yy, zz = np.meshgrid(g, g, g, indexing="ij")
    pos = np.stack([xx.ravel(), yy.ravel(), zz.ravel()], axis=1).astype(np.float64) * 300.0 + 150.0
    n = len(pos)
    rng = np.random.default_rng(8)
    sc = {"entity": np.arange(n, dtype=np.int32), "type": np.zeros(n, np.uint8), "pos": pos, "radius": rng.uniform(1.0, 200.0, n).astype(np.float32)}
    cs = api.CullingSystem(gpu_ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    st = cs.stats()
    assert st["cells"] == n and st["entities"] == n
    ocs = oracle_port.culling_system()
    ocs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    fr8 = H.cascade_frusta(api, 8)
    res = cs.cull(fr8)
    for f in range(8):
        H.assert_same_visible(gpu_visible(res, f), oracle_visible(ocs, fr8[f : f + 1]), f"8 frusta, frustum {f}")
    fr1 = H.frusta(api, names=["origin_yaw_pitch"])
    H.assert_same_visible(gpu_visible(cs.cull(fr1), 0), oracle_visible(ocs, fr1), "fused path")


def test_cull_auto_compaction_switch(gpu_ctx, oracle_port):
    """LMX_CULL_OPT_AUTO_COMPACTION = 0: the overflow set grows past the automatic threshold (max(65536, n / 8)) without a re-sort until
    the host asks for one; results are the oracle's before and after."""
    n, extra = 100_000, 70_000
    sc = scenes.cull_scene(n + extra, 3000.0, seed=13)
    cs = api.CullingSystem(gpu_ctx)
    ocs = oracle_port.culling_system()
    cs.build(sc["entity"][:n], sc["type"][:n], sc["pos"][:n], sc["radius"][:n])
    ocs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    fr = H.frusta(api, names=["origin_identity", "origin_yaw_pitch"])
    try:
        cs.setOption(api.CULL_OPT_AUTO_COMPACTION, 0)
        cs.addMany(sc["entity"][n:], sc["type"][n:], sc["pos"][n:], sc["radius"][n:])
        res = cs.cull(fr)
        assert cs.updateStats()["overflow"] == extra  # above the threshold, still unsorted
        for k in range(len(fr)):
            H.assert_same_visible(gpu_visible(res, k), oracle_visible(ocs, fr[k : k + 1]), f"overflow, frustum {k}")
        cs.compact()
        res = cs.cull(fr)
        assert cs.updateStats()["overflow"] == 0 and cs.updateStats()["static"] == n + extra
        for k in range(len(fr)):
            H.assert_same_visible(gpu_visible(res, k), oracle_visible(ocs, fr[k : k + 1]), f"compacted, frustum {k}")
        cs.setOption(api.CULL_OPT_AUTO_COMPACTION, 1)
        cs.removeMany(sc["entity"][:10])
        for e in sc["entity"][:10]:
            ocs.remove(int(e))
        res = cs.cull(fr)
        H.assert_same_visible(gpu_visible(res, 0), oracle_visible(ocs, fr[0:1]), "after re-enabling")
    finally:
        cs.setOption(api.CULL_OPT_AUTO_COMPACTION, 1)


def test_cull_map_all_equals_read_all(gpu_ctx):
    """lmx_cull_map_all (gather kernels write into pinned mapped host memory, one host wait) returns what lmx_cull_read_all returns,
    per frustum, types in order; the mapped view survives reading another frustum's list only until the next map on that view."""
    sc = H.mixed_scene(60_000, 2500.0, seed=17)
    cs = api.CullingSystem(gpu_ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    fr = H.frusta(api, names=["origin_identity", "origin_yaw_pitch", "ortho_cascade_large"])
    res = cs.cull(fr)
    for f in range(len(fr)):
        want_ids, want_types = res.all_ids(f)
        ids, types = res.map_all(f)
        assert np.array_equal(types, want_types) and len(ids) == len(want_ids)
        for t in range(api.MAX_TYPES):
            assert np.array_equal(np.sort(ids[types == t]), np.sort(want_ids[want_types == t]))
    empty = cs.cull(api.viewport_frustum(pos=(1.0e6, 0.0, 0.0)))
    ids, types = empty.map_all(0)
    assert len(ids) == 0 and len(types) == 0


@pytest.mark.parametrize("seed", [2, 4, 8, 16, 19, 27])
def test_cull_fuzz(oracle_port, seed):
    """tests/fuzz_cull.py: random interleavings of single / batched add, remove, set*, explicit and automatic re-sorts (synchronous and
    on the worker: these seeds see up to 16 swaps), scene reloads, entity indices re-used, NaN / inf / negative radii, positions on
    cell boundaries and at 1e6 - every cull (random cameras, 1-8 frusta, type filters, views, tile variants, pass widths, all three
    read paths) against the oracle (culling_system.cpp:131-369)."""
    from tests import fuzz_cull

    st = fuzz_cull.run(seed, 300, oracle_port)
    assert st["culls"] > 30
