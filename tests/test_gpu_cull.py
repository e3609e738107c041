"""Parity of the HIP cull path (through the C ABI) against the CPU oracle and the golden fixtures. Needs an MI355X."""
import os

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
        cs.setPassWidth(1)


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


def test_cull_incremental_ops(gpu_ctx, oracle_port):
    """CullingSystem::add/remove/set/setPosition/setRadius (culling_system.cpp:131-258) against the oracle."""
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


def test_cull_10m_properties(gpu_ctx):
    """BASELINE config 2 size (10 M): size-independent properties instead of a CPU comparison.

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
    assert st["cells"] == n and st["chunks"] * 64 >= 4 * n  # 239 spheres per 1024-slot block
    ocs = oracle_port.culling_system()
    ocs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    fr8 = H.cascade_frusta(api, 8)
    res = cs.cull(fr8)
    for f in range(8):
        H.assert_same_visible(gpu_visible(res, f), oracle_visible(ocs, fr8[f : f + 1]), f"8 frusta, frustum {f}")
    fr1 = H.frusta(api, names=["origin_yaw_pitch"])
    H.assert_same_visible(gpu_visible(cs.cull(fr1), 0), oracle_visible(ocs, fr1), "fused path")
