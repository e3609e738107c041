"""Parity of the HIP transform-hierarchy and skinning paths against the oracle / golden fixtures. Needs an MI355X."""
import os

import numpy as np
import pytest

from lumixengine_amd import api, scenes
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def oracle_world(oracle, h):
    n = len(h["parent"])
    w = oracle.world(n)
    roots = np.flatnonzero(h["parent"] < 0).astype(np.int32)
    kids = np.flatnonzero(h["parent"] >= 0).astype(np.int32)
    w.init_transforms(roots, h["local"][roots])
    w.set_parents(h["parent"][kids], kids)
    w.set_local_transforms(kids, h["local"][kids])
    return w, roots, kids


def gpu_inputs(ow, parent, roots):
    """children: the oracle's stored Hierarchy::local_transform; roots: their world transform."""
    tr = ow.get_local_transforms()
    tr[roots] = ow.get_transforms()[roots]
    return tr


def test_world_golden(gpu_ctx):
    g = np.load(os.path.join(G, "transforms.npz"))
    parent = g["parent"]
    roots = np.flatnonzero(parent < 0).astype(np.int32)
    tr = np.ascontiguousarray(g["locals"]).copy()
    tr[roots] = g["world0"][roots]
    w = api.World(gpu_ctx)
    w.build(parent, tr)
    w.setTransforms(roots, g["new_root"])
    w.propagate()
    assert H.transforms_bits_equal(w.getTransforms(), g["world1"])


@pytest.mark.parametrize("kind", ["chains", "fans", "flat"])
def test_world_propagate_bit_exact(gpu_ctx, oracle_port, kind):
    if kind == "chains":
        h = scenes.hierarchy_chains(20000, 4, seed=2)
    elif kind == "fans":
        h = scenes.hierarchy_fans(50, 10, 4, seed=3)  # 50 x (1 + 10 + 100 + 1000)
    else:
        h = scenes.hierarchy_chains(5000, 1, seed=8)  # roots only
    ow, roots, kids = oracle_world(oracle_port, h)
    w = api.World(gpu_ctx)
    w.build(h["parent"], gpu_inputs(ow, h["parent"], roots))
    rng = np.random.default_rng(17)
    for frame in range(3):
        new_root = scenes.random_transforms(rng, len(roots), 4000.0)
        ow.set_transforms(roots, new_root)
        w.setTransforms(roots, new_root)
        w.propagate()
        assert H.transforms_bits_equal(w.getTransforms(), ow.get_transforms()), f"{kind} frame {frame}"


def _depths(parent):
    depth = np.zeros(len(parent), np.int32)
    order = np.argsort(parent, kind="stable")  # not a topological order in general: iterate to a fixed point
    changed = True
    while changed:
        changed = False
        for e in order:
            p = parent[e]
            d = 0 if p < 0 else depth[p] + 1
            if d != depth[e]:
                depth[e] = d
                changed = True
    return depth


@pytest.mark.parametrize("kind", ["chains", "fans"])
def test_world_child_writes_bit_exact(gpu_ctx, live_oracle, kind):
    """World::setLocalTransform and World::setTransform on entities WITH a parent (world.cpp:741-753, :337-342): the reference
    re-derives the stored local with Transform::computeLocal after composing (world.cpp:266-269), which is lossy, and later frames
    compose with that re-derived local. Several frames of interleaved root moves, child local writes and child world-space writes
    (issued ancestors-first, as the batch form defines) must give bit-identical world AND local transforms."""
    oracle_port = live_oracle
    h = scenes.hierarchy_chains(3000, 4, seed=21) if kind == "chains" else scenes.hierarchy_fans(20, 6, 4, seed=22)
    ow, roots, kids = oracle_world(oracle_port, h)
    parent = h["parent"]
    depth = _depths(parent)
    w = api.World(gpu_ctx)
    w.buildWithWorld(parent, ow.get_local_transforms(), ow.get_transforms())
    rng = np.random.default_rng(31)
    n = len(parent)
    for frame in range(5):
        picked = rng.permutation(n)[: n // 3].astype(np.int32)
        how = rng.integers(0, 2, size=len(picked))  # 0: setLocalTransform (roots: setTransform), 1: world-space setTransform
        tr = scenes.random_transforms(rng, len(picked), 50.0)
        tr["pos"][parent[picked] < 0] *= 60.0
        # oracle: eager calls, ancestors first
        for d in range(int(depth.max()) + 1):
            sel = np.flatnonzero(depth[picked] == d)
            for i in sel:
                e = picked[i : i + 1]
                if how[i] == 1 or parent[e[0]] < 0:
                    ow.set_transforms(e, tr[i : i + 1])
                else:
                    ow.set_local_transforms(e, tr[i : i + 1])
        # device: two staged batches, one propagation
        loc = (how == 0) | (parent[picked] < 0)
        w.setTransforms(picked[loc], tr[loc])
        w.setWorldTransforms(picked[~loc], tr[~loc])
        w.propagate()
        assert H.transforms_bits_equal(w.getTransforms(), ow.get_transforms()), f"{kind} frame {frame}: world transforms"
        got_l, want_l = w.getLocalTransforms(), ow.get_local_transforms()
        assert H.transforms_bits_equal(got_l[kids], want_l[kids]), f"{kind} frame {frame}: stored locals"
    # and the re-derived locals are what a plain root move composes with afterwards
    new_root = scenes.random_transforms(rng, len(roots), 4000.0)
    ow.set_transforms(roots, new_root)
    w.setTransforms(roots, new_root)
    w.propagate()
    assert H.transforms_bits_equal(w.getTransforms(), ow.get_transforms())


def test_world_config3_full_size(gpu_ctx, oracle_port):
    """BASELINE config 3's hierarchy at full size: 1 M entities = 250 k roots x chains of depth 4, every root moved every frame,
    bit-exact against the reference's DFS (the CPU walks it in tens of milliseconds)."""
    h = scenes.hierarchy_chains(250_000, 4, seed=2)
    ow, roots, kids = oracle_world(oracle_port, h)
    w = api.World(gpu_ctx)
    w.build(h["parent"], gpu_inputs(ow, h["parent"], roots))
    rng = np.random.default_rng(1)
    for frame in range(2):
        new_root = scenes.random_transforms(rng, len(roots), 4000.0)
        ow.set_transforms(roots, new_root)
        w.setTransforms(roots, new_root)
        w.propagate()
        assert H.transforms_bits_equal(w.getTransforms(), ow.get_transforms()), f"frame {frame}"


def test_world_child_parent_order_independent(gpu_ctx, oracle_port):
    """Entity indices of children may be smaller than their parents': slot order, not entity order, drives levels."""
    h = scenes.hierarchy_fans(30, 3, 5, seed=13)
    n = len(h["parent"])
    rng = np.random.default_rng(5)
    perm = rng.permutation(n).astype(np.int32)  # new entity id of old entity i
    parent = np.full(n, -1, np.int32)
    local = np.zeros(n, api.TRANSFORM)
    for old in range(n):
        p = h["parent"][old]
        parent[perm[old]] = -1 if p < 0 else perm[p]
        local[perm[old]] = h["local"][old]
    hp = {"parent": parent, "local": local}
    n = len(parent)
    ow = oracle_port.world(n)
    roots = np.flatnonzero(parent < 0).astype(np.int32)
    ow.init_transforms(roots, local[roots])
    # parents must exist before children are attached: attach in BFS order
    order, seen = list(roots), set(int(r) for r in roots)
    children = {}
    for e in range(n):
        if parent[e] >= 0:
            children.setdefault(int(parent[e]), []).append(e)
    i = 0
    while i < len(order):
        for c in children.get(int(order[i]), []):
            order.append(c)
        i += 1
    kids = np.array([e for e in order if parent[e] >= 0], np.int32)
    ow.set_parents(parent[kids], kids)
    ow.set_local_transforms(kids, local[kids])
    w = api.World(gpu_ctx)
    w.build(parent, gpu_inputs(ow, parent, roots))
    new_root = scenes.random_transforms(rng, len(roots), 3000.0)
    ow.set_transforms(roots, new_root)
    w.setTransforms(roots, new_root)
    w.propagate()
    assert H.transforms_bits_equal(w.getTransforms(), ow.get_transforms())


def test_world_moves_refresh_culling(gpu_ctx, live_oracle):
    """transform -> sphere refresh -> cull, end to end (render_module.cpp:1544-1554 + culling_system.cpp:225-242)."""
    oracle_port = live_oracle
    h = scenes.hierarchy_chains(3000, 3, seed=4, root_extent=1500.0)
    n = len(h["parent"])
    rng = np.random.default_rng(33)
    model_radius = rng.uniform(0.5, 40.0, n).astype(np.float32)
    model_radius[:5] = 400.0  # a few spheres cross the is_big threshold when scales change
    ow, roots, kids = oracle_world(oracle_port, h)
    ocs = oracle_port.culling_system()
    tr0 = ow.get_transforms()
    ent = np.arange(n, dtype=np.int32)
    r0 = model_radius * np.maximum(tr0["scale"][:, 0], np.maximum(tr0["scale"][:, 1], tr0["scale"][:, 2]))
    ocs.add_bulk(ent, np.zeros(n, np.uint8), tr0["pos"], r0)
    ow.bind_culling(ocs, ent, model_radius)

    w = api.World(gpu_ctx)
    w.build(h["parent"], gpu_inputs(ow, h["parent"], roots))
    cs = api.CullingSystem(gpu_ctx)
    cs.build(ent, np.zeros(n, np.uint8), tr0["pos"], r0)
    w.bindCulling(ent, model_radius)
    fr = np.concatenate([api.viewport_frustum(pos=(0, 0, 2000.0)), api.viewport_frustum(pos=(300.0, 50.0, -100.0), rot=H.quat_from_yaw_pitch(1.0, 0.1))])
    for frame in range(4):
        # small moves keep most entities in their cell (in-place refresh), large ones force re-binning
        extent = 1500.0 if frame % 2 else 40.0
        new_root = ow.get_transforms()[roots]
        new_root["pos"] += rng.uniform(-extent, extent, size=(len(roots), 3))
        new_root["scale"] = rng.uniform(0.5, 2.0, size=(len(roots), 3)).astype(np.float32)
        ow.set_transforms(roots, new_root)
        w.setTransforms(roots, new_root)
        w.propagate()
        assert H.transforms_bits_equal(w.getTransforms(), ow.get_transforms())
        res = cs.cull(fr)
        for f in range(len(fr)):
            ids, types, _ = ocs.cull(fr[f : f + 1])
            got_ids, got_types = res.all_ids(f)
            H.assert_same_visible(H.sorted_by_type(got_ids, got_types), H.sorted_by_type(ids, types), f"frame {frame} frustum {f}")
        for e in (0, 1, 2, 100, n - 1):
            assert cs.getRadius(e) == ocs.get_radius(e)


def test_world_bound_spheres_survive_async_compaction(gpu_ctx, oracle_port):
    """LMX_CULL_OPT_ASYNC_COMPACTION with hierarchy-bound entities: their spheres are refreshed on the device (k_sphere_refresh), never
    by the host, so when the re-sorted copy of the sets trades places with the live one they must be carried over device to device
    (k_dyn_carry_over) - including the entities that did NOT move in the frames around the swap, whose only up-to-date copy is the old
    device set. 9000 bound entities in chains + 150 k static ones; every frame moves a tenth of the roots and adds 3000 static
    entities (the compaction thresholds are crossed every ~22 frames); the visible set is the oracle's after every swap."""
    import time

    h = scenes.hierarchy_chains(3000, 3, seed=41, root_extent=2500.0)
    n = len(h["parent"])
    rng = np.random.default_rng(12)
    model_radius = rng.uniform(0.5, 40.0, n).astype(np.float32)
    ow, roots, kids = oracle_world(oracle_port, h)
    ocs = oracle_port.culling_system()
    tr0 = ow.get_transforms()
    ent = np.arange(n, dtype=np.int32)
    r0 = model_radius * tr0["scale"].max(axis=1)
    ocs.add_bulk(ent, np.zeros(n, np.uint8), tr0["pos"], r0)
    n_static = 150_000
    s_ent = np.arange(n, n + n_static, dtype=np.int32)
    s_pos = rng.uniform(-3000.0, 3000.0, size=(n_static, 3))
    s_rad = np.exp(rng.uniform(np.log(0.5), np.log(60.0), size=n_static)).astype(np.float32)
    ocs.add_bulk(s_ent, np.ones(n_static, np.uint8), s_pos, s_rad)
    ow.bind_culling(ocs, ent, model_radius)

    w = api.World(gpu_ctx)
    w.build(h["parent"], gpu_inputs(ow, h["parent"], roots))
    cs = api.CullingSystem(gpu_ctx)
    cs.build(np.concatenate([ent, s_ent]), np.concatenate([np.zeros(n, np.uint8), np.ones(n_static, np.uint8)]), np.concatenate([tr0["pos"], s_pos]),
             np.concatenate([r0, s_rad]))
    w.bindCulling(ent, model_radius)
    fr = np.concatenate([api.viewport_frustum(pos=(0, 0, 2000.0)), api.viewport_frustum(pos=(300.0, 50.0, -100.0), rot=H.quat_from_yaw_pitch(1.0, 0.1))])
    try:
        cs.setOption(api.CULL_OPT_ASYNC_COMPACTION, 1)
        next_id = n + n_static
        swaps_seen, checked_after_swap, frame = 0, 0, 0
        deadline = time.time() + 240
        while (swaps_seen < 2 or frame < 60) and time.time() < deadline:
            moved = roots if frame < 2 else rng.choice(roots, size=len(roots) // 10, replace=False)
            new_root = ow.get_transforms()[moved]
            new_root["pos"] += rng.uniform(-300.0, 300.0, size=(len(moved), 3))
            new_root["scale"] = rng.uniform(0.5, 2.0, size=(len(moved), 3)).astype(np.float32)
            ow.set_transforms(moved, new_root)
            w.setTransforms(moved, new_root)
            w.propagate()
            k = 3000
            ids = np.arange(next_id, next_id + k, dtype=np.int32)
            next_id += k
            p = rng.uniform(-3000.0, 3000.0, size=(k, 3))
            r = np.exp(rng.uniform(np.log(0.5), np.log(60.0), size=k)).astype(np.float32)
            ocs.add_bulk(ids, np.ones(k, np.uint8), p, r)
            cs.addMany(ids, np.ones(k, np.uint8), p, r)
            res = cs.cull(fr)
            st = cs.asyncStats()
            assert st["state"] != 4, "the asynchronous compaction failed"
            swapped = st["swaps"] > swaps_seen
            swaps_seen = st["swaps"]
            if swapped or frame % 10 == 0:
                for f in range(len(fr)):
                    ids_o, types_o, _ = ocs.cull(fr[f : f + 1])
                    got_ids, got_types = res.all_ids(f)
                    H.assert_same_visible(H.sorted_by_type(got_ids, got_types), H.sorted_by_type(ids_o, types_o), f"frame {frame} frustum {f} (swaps {swaps_seen})")
                for e in (0, 1, 2, 100, n - 1):
                    assert cs.getRadius(e) == ocs.get_radius(e)
                checked_after_swap += 1 if swapped else 0
            frame += 1
            time.sleep(0.002)
        assert swaps_seen >= 2 and checked_after_swap >= 2, (cs.asyncStats(), frame)
        assert H.transforms_bits_equal(w.getTransforms(), ow.get_transforms())
    finally:
        cs.setOption(api.CULL_OPT_ASYNC_COMPACTION, 0)


def test_rebinding_keeps_moved_spheres(gpu_ctx, oracle_port):
    """bind A, propagate (spheres of A move on the device), bind A + B, cull WITHOUT another propagate: the entities of the first
    binding keep their moved spheres (the rebuild of the dynamic set must not re-upload a stale host mirror), also when an entity
    is dropped from the new list (it stays where the last refresh put it)."""
    h = scenes.hierarchy_chains(1200, 3, seed=14, root_extent=1200.0)
    n = len(h["parent"])
    rng = np.random.default_rng(5)
    model_radius = rng.uniform(0.5, 30.0, n).astype(np.float32)
    ow, roots, kids = oracle_world(oracle_port, h)
    ocs = oracle_port.culling_system()
    tr0 = ow.get_transforms()
    ent = np.arange(n, dtype=np.int32)
    r0 = model_radius * tr0["scale"].max(axis=1)
    ocs.add_bulk(ent, np.zeros(n, np.uint8), tr0["pos"], r0)
    first = ent[: n // 2]
    ow.bind_culling(ocs, first, model_radius[first])
    w = api.World(gpu_ctx)
    w.build(h["parent"], gpu_inputs(ow, h["parent"], roots))
    cs = api.CullingSystem(gpu_ctx)
    cs.build(ent, np.zeros(n, np.uint8), tr0["pos"], r0)
    w.bindCulling(first, model_radius[first])
    new_root = ow.get_transforms()[roots]
    new_root["pos"] += rng.uniform(-900.0, 900.0, size=(len(roots), 3))
    ow.set_transforms(roots, new_root)
    w.setTransforms(roots, new_root)
    w.propagate()
    # second binding: everything except entity first[3], plus the other half
    second = np.concatenate([np.delete(first, 3), ent[n // 2:]])
    ow.bind_culling(ocs, second, model_radius[second])
    w.bindCulling(second, model_radius[second])
    fr = np.concatenate([api.viewport_frustum(pos=(0, 0, 1500.0)), api.viewport_frustum(pos=(-200.0, 30.0, 100.0), rot=H.quat_from_yaw_pitch(-0.8, 0.05))])
    res = cs.cull(fr)
    for f in range(len(fr)):
        ids, types, _ = ocs.cull(fr[f : f + 1])
        got_ids, got_types = res.all_ids(f)
        H.assert_same_visible(H.sorted_by_type(got_ids, got_types), H.sorted_by_type(ids, types), f"frustum {f}")
    for e in (0, 3, int(first[3]), n // 2 - 1, n - 1):
        assert cs.getRadius(e) == ocs.get_radius(e)


@pytest.mark.parametrize("depth, fused", [(4, 1), (4, 0), (8, 1), (12, 1)])
def test_world_fused_and_per_level_propagation_bit_exact(gpu_ctx, oracle_port, depth, fused):
    """The one-launch form (every node re-composes down from its topmost written ancestor, hierarchies of <= 8 levels) and the one-launch-
    per-level form (deeper hierarchies, or LMX_WORLD_OPT_FUSED_LEVELS = 0) against World::transformEntity (world.cpp:255-282), bit for bit,
    world transforms and stored locals: moved roots, children written in local space under moved and unmoved parents, children written in
    world space, several levels written in one frame (applied in hierarchy order on the reference side: the semantics of a batch)."""
    h = scenes.hierarchy_chains(120, depth, seed=31)
    n = len(h["parent"])
    ow, roots, kids = oracle_world(oracle_port, h)
    w = api.World(gpu_ctx)
    w.setOption(api.WORLD_OPT_FUSED_LEVELS, fused)
    try:
        w.buildWithWorld(h["parent"], gpu_inputs(ow, h["parent"], roots), ow.get_transforms())  # a mirror of the live World: nothing is recomputed
        w.propagate()
        assert H.transforms_bits_equal(w.getTransforms(), ow.get_transforms())
        rng = np.random.default_rng(depth * 10 + fused)
        level = np.zeros(n, np.int32)
        for e in range(n):
            p, d = h["parent"][e], 0
            while p >= 0:
                p, d = h["parent"][p], d + 1
            level[e] = d
        for frame in range(4):
            picked = rng.choice(n, 90, replace=False)
            picked = picked[np.argsort(level[picked], kind="stable")]  # hierarchy order = what a batch means
            world_space = (level[picked] == 0) | (rng.random(len(picked)) < 0.3)
            tr = scenes.random_transforms(rng, len(picked), 50.0)
            for e, ws, t in zip(picked, world_space, tr):
                (ow.set_transforms if ws else ow.set_local_transforms)(np.array([e], np.int32), t[None])
            w.setWorldTransforms(picked[world_space].astype(np.int32), tr[world_space])
            w.setTransforms(picked[~world_space].astype(np.int32), tr[~world_space])
            w.propagate()
            assert H.transforms_bits_equal(w.getTransforms(), ow.get_transforms()), f"frame {frame}: world transforms"
            got_l, want_l = w.getLocalTransforms(), ow.get_local_transforms()
            assert H.transforms_bits_equal(got_l[kids], want_l[kids]), f"frame {frame}: stored locals"
    finally:
        w.setOption(api.WORLD_OPT_FUSED_LEVELS, 0)


def test_world_duplicate_writes_last_one_wins(gpu_ctx, oracle_port):
    """Two writes to ONE entity in a batch: the reference applies writes one by one, so the last one stands (world.cpp:337-342); the
    batch keeps only the last record of an entity (the scatter kernel runs one thread per record - two records would race)."""
    h = scenes.hierarchy_chains(300, 3, seed=17)
    ow, roots, kids = oracle_world(oracle_port, h)
    w = api.World(gpu_ctx)
    w.buildWithWorld(h["parent"], gpu_inputs(ow, h["parent"], roots), ow.get_transforms())
    w.propagate()
    rng = np.random.default_rng(5)
    ent = np.concatenate([roots[:50], roots[:50][::-1], roots[10:20]]).astype(np.int32)  # every one of the 50 twice, ten of them three times
    tr = scenes.random_transforms(rng, len(ent), 2000.0)
    ow.set_transforms(ent, tr)  # one by one, in order
    w.setTransforms(ent, tr)
    w.propagate()
    assert H.transforms_bits_equal(w.getTransforms(), ow.get_transforms())
    kid = np.concatenate([kids[:40], kids[:40]]).astype(np.int32)
    tk = scenes.random_transforms(rng, len(kid), 5.0)
    ow.set_local_transforms(kid, tk)
    w.setTransforms(kid, tk)
    w.propagate()
    assert H.transforms_bits_equal(w.getTransforms(), ow.get_transforms())


def test_world_moved_list_is_what_the_dfs_visits(gpu_ctx, oracle_port):
    """lmx_world_track_moved / lmx_world_read_moved: the per-frame hand-back lists exactly the entities World::transformEntity visits
    (world.cpp:255-282: the written entity and its whole subtree) with their new world transforms, nothing else; two propagations
    between reads accumulate; the list is empty when nothing was written."""
    h = scenes.hierarchy_fans(40, 4, 4, seed=13)
    n = len(h["parent"])
    ow, roots, kids = oracle_world(oracle_port, h)
    w = api.World(gpu_ctx)
    w.trackMoved(True)
    try:
        w.buildWithWorld(h["parent"], gpu_inputs(ow, h["parent"], roots), ow.get_transforms())
        w.propagate()
        ent, _ = w.readMoved()
        assert len(ent) == 0, "a mirrored World moves nothing until something is written"
        rng = np.random.default_rng(3)
        children = [[] for _ in range(n)]
        for c in kids:
            children[h["parent"][c]].append(int(c))

        def subtree(e):
            out, stack = [], [int(e)]
            while stack:
                x = stack.pop()
                out.append(x)
                stack += children[x]
            return out

        picked = rng.choice(roots, 5, replace=False).astype(np.int32)
        new_root = scenes.random_transforms(rng, len(picked), 3000.0)
        kid = np.array([kids[7], kids[300]], np.int32)  # two children in other subtrees
        kid = kid[[int(k) not in sum((subtree(r) for r in picked), []) for k in kid]]
        new_kid = scenes.random_transforms(rng, len(kid), 5.0)
        ow.set_transforms(picked, new_root)
        ow.set_local_transforms(kid, new_kid)
        w.setTransforms(picked, new_root)
        w.setTransforms(kid, new_kid)
        w.propagate()
        ent, tr = w.readMoved()
        want = sorted(sum((subtree(e) for e in list(picked) + list(kid)), []))
        assert sorted(ent.tolist()) == want
        assert H.transforms_bits_equal(tr, ow.get_transforms()[ent])
        # two propagations before one read: both frames' movers are listed, later entries are newer
        second = scenes.random_transforms(rng, 1, 3000.0)
        for frame_tr in (new_root[:1], second):
            ow.set_transforms(picked[:1], frame_tr)
            w.setTransforms(picked[:1], frame_tr)
            w.propagate()
        ent, tr = w.readMoved()
        sub = sorted(subtree(picked[0]))
        assert sorted(ent.tolist()) == sorted(sub + sub)
        last = {int(e): i for i, e in enumerate(ent)}
        idx = np.array([last[e] for e in sub])
        assert H.transforms_bits_equal(tr[idx], ow.get_transforms()[np.array(sub)])
        assert H.transforms_bits_equal(w.getTransforms(), ow.get_transforms())
    finally:
        w.trackMoved(False)


def close_1e5(got, want):
    """north star: skinned vertex positions within 1e-5 relative fp32 - PER VERTEX: every component within 1e-5 of that vertex's own
    magnitude (max |component|), floored at 1 % of the mesh's extent (a vertex that lands next to the origin is still the sum of
    terms as large as the skeleton: its rounding error does not shrink with it)."""
    got = np.asarray(got, np.float64).reshape(-1, 3)
    want = np.asarray(want, np.float64).reshape(-1, 3)
    scale = np.maximum(np.abs(want).max(axis=1), 1e-2 * float(np.abs(want).max()))
    return bool((np.abs(got - want).max(axis=1) <= 1e-5 * scale).all())


@pytest.mark.parametrize("exact", [True, False])
def test_skin_golden(gpu_ctx, exact):
    g = np.load(os.path.join(G, "skin.npz"))
    sk = api.Skinning(gpu_ctx)
    sk.setMode(exact)
    sk.enableDualQuats(True)
    model = sk.addModel(g["parents"], g["bind"], int(g["first_nonroot"][0]))
    mesh = sk.addMesh(g["verts"], g["skin"])
    n_inst = g["rel_pos"].shape[0]
    sk.setInstances([model] * n_inst, [mesh] * n_inst)
    sk.uploadPoses(g["rel_pos"], g["rel_rot"])
    sk.run()
    for i in range(n_inst):
        pos, rot = sk.readPose(i)
        assert H.bits_equal(pos, g["abs_pos"][i]) and H.bits_equal(rot, g["abs_rot"][i])
        assert H.bits_equal(sk.readPalette(i), g["palette"][i])
        assert H.bits_equal(sk.readDualQuats(i), g["dual_quats"][i])  # computeSkeletonDualQuats, pipeline.cpp:2680-2745
        got, want = sk.readVertices(i), g["skinned"][i]
        assert close_1e5(got, want)
        if exact:  # LMX_SKIN_EXACT is FMA-free and bit-identical to the reference
            assert H.bits_equal(got, want)
    sk.setMode(False)


def test_skin_groups_and_pose_writeback_switch(gpu_ctx, live_oracle):
    """Runs of one model are walked 16 / 8 instances at a time (ragged last group, model change mid-run); with the absolute
    pose store switched off the palettes and vertices are unchanged, readPose fails, and uploaded poses stay relative."""
    oracle_port = live_oracle
    sk = api.Skinning(gpu_ctx)
    sk.setMode(True)
    skel = [scenes.skeleton(64, seed=4), scenes.skeleton(100, seed=24)]
    meshes = [scenes.skinned_mesh(300, 64, seed=6), scenes.skinned_mesh(129, 100, seed=7)]
    models = [sk.addModel(s["parents"], s["bind"], s["first_nonroot"]) for s in skel]
    mesh_ids = [sk.addMesh(v, s) for v, s in meshes]
    pick = [0] * 37 + [1] * 19 + [0] * 3 + [1]
    sk.setInstances([models[k] for k in pick], [mesh_ids[k] for k in pick])
    poses = [scenes.relative_poses(1, len(skel[k]["parents"]), seed=300 + i) for i, k in enumerate(pick)]
    rel_pos = np.concatenate([p[0].reshape(-1, 3) for p in poses])
    rel_rot = np.concatenate([p[1].reshape(-1, 4) for p in poses])
    sk.setPoseWriteback(False)
    sk.uploadPoses(rel_pos, rel_rot)
    want = []
    for i, k in enumerate(pick):
        s = skel[k]
        apos, arot = oracle_port.pose_compute_absolute(poses[i][0], poses[i][1], s["parents"], s["first_nonroot"])
        pal = oracle_port.skin_matrices(apos, arot, oracle_port.invert_bind(s["bind"]))
        want.append((apos[0], arot[0], pal[0], oracle_port.evaluate_skin(meshes[k][0], meshes[k][1], pal)[0]))
    for _ in range(2):  # the second run starts from the same (still relative) uploaded poses
        sk.run()
        for i in range(len(pick)):
            assert H.bits_equal(sk.readPalette(i), want[i][2])
            assert H.bits_equal(sk.readVertices(i), want[i][3])
        with pytest.raises(api.LumixError):
            sk.readPose(0)
    sk.setPoseWriteback(True)
    sk.run()
    for i in range(len(pick)):
        pos, rot = sk.readPose(i)
        assert H.bits_equal(pos, want[i][0]) and H.bits_equal(rot, want[i][1])
        assert H.bits_equal(sk.readPalette(i), want[i][2])
    sk.setMode(False)


def test_pose_blend_golden(gpu_ctx, oracle_port):
    """Pose::blend (pose.cpp:30-41) of the skin golden's relative poses with a second set: bit-exact against the golden written by
    the reference's own nlerp, and the chain into the palette stays bit-exact; thresholds (weight <= 0.001 no-op, clamp to 1)."""
    g, gb = np.load(os.path.join(G, "skin.npz")), np.load(os.path.join(G, "blend.npz"))
    sk = api.Skinning(gpu_ctx)
    sk.setMode(True)
    model = sk.addModel(g["parents"], g["bind"], int(g["first_nonroot"][0]))
    mesh = sk.addMesh(g["verts"], g["skin"])
    n_inst = g["rel_pos"].shape[0]
    sk.setInstances([model] * n_inst, [mesh] * n_inst)
    sk.uploadPoses(g["rel_pos"], g["rel_rot"])
    sk.blendPoses(gb["rhs_pos"], gb["rhs_rot"], 0.0005)  # below the threshold: untouched
    sk.blendPoses(gb["rhs_pos"], gb["rhs_rot"], float(gb["weight"][0]))
    for i in range(n_inst):
        pos, rot = sk.readRelativePose(i)
        assert H.bits_equal(pos, gb["pos"][i]) and H.bits_equal(rot, gb["rot"][i])
    sk.run()
    apos, arot = oracle_port.pose_compute_absolute(gb["pos"], gb["rot"], g["parents"], int(g["first_nonroot"][0]))
    pal = oracle_port.skin_matrices(apos, arot, oracle_port.invert_bind(g["bind"]))
    for i in range(n_inst):
        assert H.bits_equal(sk.readPalette(i), pal[i])
    with pytest.raises(api.LumixError):
        sk.blendPoses(gb["rhs_pos"], gb["rhs_rot"], 0.5)  # the poses are absolute now
    sk.uploadPoses(g["rel_pos"], g["rel_rot"])
    sk.blendPoses(gb["rhs_pos"], gb["rhs_rot"], 3.0)  # clamped to 1
    want = oracle_port.pose_blend(g["rel_pos"], g["rel_rot"], gb["rhs_pos"], gb["rhs_rot"], 1.0)
    pos, rot = sk.readRelativePose(2)
    assert H.bits_equal(pos, want[0][2]) and H.bits_equal(rot, want[1][2])
    sk.setMode(False)


def test_bone_attachments_golden_and_subtrees(gpu_ctx, oracle_port):
    """updateBoneAttachment for 64 attachments at once (render_module.cpp:377-404): attached roots against the golden from the
    reference's object code, their children against the oracle's compose, bit for bit; the pose is the skin golden's."""
    g, ga = np.load(os.path.join(G, "skin.npz")), np.load(os.path.join(G, "attach.npz"))
    sk = api.Skinning(gpu_ctx)
    sk.setMode(True)
    model = sk.addModel(g["parents"], g["bind"], int(g["first_nonroot"][0]))
    mesh = sk.addMesh(g["verts"], g["skin"])
    n_inst = g["rel_pos"].shape[0]
    sk.setInstances([model] * n_inst, [mesh] * n_inst)
    sk.uploadPoses(g["rel_pos"], g["rel_rot"])
    n = len(ga["parent"])
    # entities: [0, n) parents (model instances), [n, 2n) attached roots, [2n, 3n) one child under every attached root
    rng = np.random.default_rng(23)
    tr = np.zeros(3 * n, api.TRANSFORM)
    tr[:n] = ga["parent"]
    tr[n : 2 * n] = scenes.random_transforms(rng, n, 100.0)
    tr["scale"][n : 2 * n] = ga["scale"]
    tr[2 * n :] = scenes.random_transforms(rng, n, 3.0)  # locals of the children
    parent = np.full(3 * n, -1, np.int32)
    parent[2 * n :] = np.arange(n, 2 * n)
    w = api.World(gpu_ctx)
    w.build(parent, tr)
    w.setBoneAttachments(np.arange(n, 2 * n), np.arange(n), ga["instance"], ga["bone"], ga["relative"])
    with pytest.raises(api.LumixError):
        w.updateBoneAttachments()  # the pose is still relative (ASSERT(pose->is_absolute), render_module.cpp:424)
    sk.run()
    w.updateBoneAttachments()
    w.propagate()
    got = w.getTransforms()
    assert H.transforms_bits_equal(got[n : 2 * n], ga["result"])
    assert H.transforms_bits_equal(got[:n], ga["parent"])
    assert H.transforms_bits_equal(got[2 * n :], oracle_port.compose(ga["result"], tr[2 * n :]))
    # a chain (attachment hanging off an attached entity) and an attached non-root are refused
    with pytest.raises(api.LumixError):
        w.setBoneAttachments([n, n + 1], [0, n], [0, 0], [1, 1], ga["relative"][:2])
    with pytest.raises(api.LumixError):
        w.setBoneAttachments([2 * n], [0], [0], [1], ga["relative"][:1])
    sk.setMode(False)


def test_skin_dual_quaternion_blend(gpu_ctx, oracle_port, oracle_ref):
    """LMX_SKIN_DQS: the SKINNED branch of the reference's vertex shader (surface_base.hlsli:196-217, transformByDualQuat
    common.hlsli:632-636) on the bit-exact dual-quaternion palette, through both vertex kernels (a run of 6 instances on a
    2500-vertex mesh -> k_skin_multi, 3 single small ones -> k_skin_vertices), against the reference's OWN shader text compiled as C++
    (oracle/ref/slice_hlsl.py: f3 pinned) and the plain-C restatement. Tolerance 1e-5: HLSL does not pin association."""
    sk = api.Skinning(gpu_ctx)
    skel = [scenes.skeleton(64, seed=4), scenes.skeleton(100, seed=24)]
    meshes = [scenes.skinned_mesh(2500, 64, seed=6), scenes.skinned_mesh(300, 100, seed=7)]
    models = [sk.addModel(s["parents"], s["bind"], s["first_nonroot"]) for s in skel]
    mesh_ids = [sk.addMesh(v, s) for v, s in meshes]
    pick = [0] * 6 + [1, 0, 1]
    sk.setInstances([models[k] for k in pick], [mesh_ids[k] for k in pick])
    poses = [scenes.relative_poses(1, len(skel[k]["parents"]), seed=700 + i) for i, k in enumerate(pick)]
    sk.uploadPoses(np.concatenate([p[0].reshape(-1, 3) for p in poses]), np.concatenate([p[1].reshape(-1, 4) for p in poses]))
    sk.setMode(api.SKIN_DQS)
    sk.run()
    for i, k in enumerate(pick):
        s = skel[k]
        apos, arot = oracle_port.pose_compute_absolute(poses[i][0], poses[i][1], s["parents"], s["first_nonroot"])
        dq = oracle_port.dual_quats(apos, arot, oracle_port.invert_bind(s["bind"]))
        assert H.bits_equal(sk.readDualQuats(i), dq[0])
        want = oracle_ref.evaluate_dq_skin_hlsl(meshes[k][0], meshes[k][1], dq)[0]  # the sliced shader text
        got = sk.readVertices(i)
        assert close_1e5(got, want), f"instance {i}"
        assert H.bits_equal(want, oracle_port.evaluate_dq_skin(meshes[k][0], meshes[k][1], dq)[0])  # ... which the restatement reproduces bit for bit
        # a different deformation than linear blending, but of the same skeleton: same ballpark, not the same numbers
        lbs = oracle_port.evaluate_skin(meshes[k][0], meshes[k][1], oracle_port.skin_matrices(apos, arot, oracle_port.invert_bind(s["bind"])))[0]
        assert not np.allclose(got, lbs, rtol=1e-3, atol=1e-3)
    # rigid case: all four weights on one bone -> DQS and LBS agree (a dual quaternion of one bone is that bone's rigid transform)
    verts, skin = scenes.skinned_mesh(2100, 64, seed=9)
    skin["indices"][:, 1:] = skin["indices"][:, :1]
    one = sk.addMesh(verts, skin)
    sk.setInstances([models[0]] * 2, [one] * 2)
    p2 = scenes.relative_poses(2, 64, seed=31)
    for mode in (api.SKIN_DQS, api.SKIN_EXACT):
        sk.uploadPoses(p2[0].reshape(-1, 3), p2[1].reshape(-1, 4))
        sk.setMode(mode)
        sk.run()
        if mode == api.SKIN_DQS:
            dqs = [sk.readVertices(i) for i in range(2)]
        else:
            for i in range(2):
                assert np.allclose(dqs[i], sk.readVertices(i), rtol=2e-4, atol=2e-4)
    sk.setMode(api.SKIN_FUSED)


def test_skin_dqs_bounded_by_exact_evaluation(gpu_ctx, oracle_port):
    """f3: LMX_SKIN_DQS held to an extended-precision evaluation of the shader's own expressions (surface_base.hlsli:196-217,
    common.hlsli:632-636; tests/dq_exact.py states the bound and why HLSL admits no bit-exact target), on the adversarial case -
    antipodal quaternions, hemisphere tests decided by rounding (both signs admissible), real parts with w ~ 0, nearly cancelling
    blends - and on an ordinary skeleton. tests/test_dq_blend_bound.py holds the plain-C restatement to the same bound, so
    HIP == restatement within 1e-5 (test_skin_dual_quaternion_blend) is no longer the only anchor of this path."""
    from tests import dq_exact as DQ

    pos, rot, verts, skin = DQ.adversarial_case()
    nb = len(pos)
    ident = np.zeros(nb, api.LOCAL_RIGID)
    ident["rot"][:, 3] = 1.0
    sk = api.Skinning(gpu_ctx)
    # every bone a root with an identity bind pose: the absolute pose IS the uploaded pose, the palette its dual quaternion
    flat = sk.addModel(np.full(nb, -1, np.int16), ident, nb)
    s = scenes.skeleton(64, seed=4)
    tree = sk.addModel(s["parents"], s["bind"], s["first_nonroot"])
    rverts, rskin = scenes.skinned_mesh(5000, 64, seed=6)
    m_adv, m_rnd = sk.addMesh(verts, skin), sk.addMesh(rverts, rskin)
    sk.setInstances([flat, tree], [m_adv, m_rnd])
    rpos, rrot = scenes.relative_poses(1, 64, seed=706)
    sk.uploadPoses(np.concatenate([pos, rpos[0]]), np.concatenate([rot, rrot[0]]))
    sk.setMode(api.SKIN_DQS)
    try:
        sk.run()
        dq_adv = oracle_port.dual_quats(pos[None], rot[None], oracle_port.invert_bind(ident))[0]  # (the reference's conj negates w: -q, the same rotation)
        apos, arot = oracle_port.pose_compute_absolute(rpos, rrot, s["parents"], s["first_nonroot"])
        dq_rnd = oracle_port.dual_quats(apos, arot, oracle_port.invert_bind(s["bind"]))[0]
        for i, (v, sn, dq, what) in enumerate(((verts, skin, dq_adv, "adversarial"), (rverts, rskin, dq_rnd, "random"))):
            assert H.bits_equal(sk.readDualQuats(i), dq), what  # the palette itself is pinned bit for bit (a18)
            got = sk.readVertices(i)
            cands, bounds = DQ.dq_skin_candidates(v, sn, dq)
            ok = DQ.within_bound(got, cands, bounds)
            assert ok.all(), f"{what}: {int((~ok).sum())} vertices outside the bound, worst ratio {DQ.worst_ratio(got, cands, bounds):.2f}, first {np.flatnonzero(~ok)[:8]}"
        assert DQ.worst_ratio(sk.readVertices(1), *DQ.dq_skin_candidates(rverts, rskin, dq_rnd)) < 0.5  # well inside on ordinary inputs
    finally:
        sk.setMode(api.SKIN_FUSED)


@pytest.mark.parametrize("per_block", [0, 1, 2, 4, 8, 16])
@pytest.mark.parametrize("exact", [True, False])
def test_skin_shared_mesh_runs(gpu_ctx, live_oracle, exact, per_block):
    """Runs of instances that share a mesh take the register-resident path (k_skin_shared: ragged tiles, 2 tiles per mesh,
    8-copy palettes for 100 bones) or - per_block instances at a time - k_skin_multi (ragged last groups, 8 columns for 100 bones);
    single instances and small meshes the streaming kernel; all in one instance table."""
    oracle_port = live_oracle
    sk = api.Skinning(gpu_ctx)
    sk.setMode(exact)
    sk.setOption(api.SKIN_OPT_INSTANCES_PER_BLOCK, per_block)
    skel = [scenes.skeleton(64, seed=4), scenes.skeleton(100, seed=24), scenes.skeleton(64, seed=34)]
    # mesh 3 references 60 of its model's 100 bones, mesh 4 40 of 64: k_skin_multi stages only the bones a mesh references (and picks its bank
    # columns by that number: 16 for the 100-bone model's 60), the palettes keep their models' strides
    meshes = [scenes.skinned_mesh(5121, 64, seed=6), scenes.skinned_mesh(2049, 100, seed=7), scenes.skinned_mesh(700, 64, seed=8),
              scenes.skinned_mesh(2300, 60, seed=9), scenes.skinned_mesh(1900, 40, seed=10)]
    models = [sk.addModel(s["parents"], s["bind"], s["first_nonroot"]) for s in skel]
    mesh_ids = [sk.addMesh(v, s) for v, s in meshes]
    # (model, mesh): a run of 9 on mesh 0 (two models with 64 bones: still one run), a single, a run of 3 on mesh 1, small meshes
    pick = [(0, 0)] * 5 + [(2, 0)] * 4 + [(1, 1)] + [(0, 2)] * 3 + [(1, 1)] * 3 + [(0, 0)] + [(1, 3)] * 5 + [(2, 4)] * 3
    sk.setInstances([models[m] for m, _ in pick], [mesh_ids[g] for _, g in pick])
    poses = [scenes.relative_poses(1, len(skel[m]["parents"]), seed=500 + i) for i, (m, _) in enumerate(pick)]
    sk.uploadPoses(np.concatenate([p[0].reshape(-1, 3) for p in poses]), np.concatenate([p[1].reshape(-1, 4) for p in poses]))
    sk.run()
    for i, (m, g) in enumerate(pick):
        s = skel[m]
        apos, arot = oracle_port.pose_compute_absolute(poses[i][0], poses[i][1], s["parents"], s["first_nonroot"])
        pal = oracle_port.skin_matrices(apos, arot, oracle_port.invert_bind(s["bind"]))
        want = oracle_port.evaluate_skin(meshes[g][0], meshes[g][1], pal)[0]
        got = sk.readVertices(i)
        assert close_1e5(got, want), f"instance {i} vertices"
        if exact:
            assert H.bits_equal(got, want), f"instance {i} vertices (exact mode)"
    if per_block:  # the dual-quaternion blend through k_skin_multi against the same blend through k_skin_vertices: the same bits
        sk.setMode(api.SKIN_DQS)
        sk.uploadPoses(np.concatenate([p[0].reshape(-1, 3) for p in poses]), np.concatenate([p[1].reshape(-1, 4) for p in poses]))
        sk.run()
        multi = [sk.readVertices(i) for i in range(len(pick))]
        sk.setOption(api.SKIN_OPT_INSTANCES_PER_BLOCK, 0)
        sk.uploadPoses(np.concatenate([p[0].reshape(-1, 3) for p in poses]), np.concatenate([p[1].reshape(-1, 4) for p in poses]))
        sk.run()
        for i in range(len(pick)):
            assert H.bits_equal(multi[i], sk.readVertices(i)), f"instance {i}: DQS through k_skin_multi != k_skin_vertices"
    sk.setMode(False)
    sk.setOption(api.SKIN_OPT_INSTANCES_PER_BLOCK, api.SKIN_INSTANCES_PER_BLOCK_DEFAULT)


@pytest.mark.parametrize("exact", [True, False])
def test_skin_many_instances_vs_oracle(gpu_ctx, live_oracle, exact):
    """Two models (64 and 196 bones = Model::Bone::MAX_COUNT), three meshes with ragged vertex counts."""
    oracle_port = live_oracle
    sk = api.Skinning(gpu_ctx)
    sk.setMode(exact)
    skel = [scenes.skeleton(64, seed=4), scenes.skeleton(196, seed=14), scenes.skeleton(1, seed=15)]
    meshes = [scenes.skinned_mesh(1000, 64, seed=6), scenes.skinned_mesh(257, 196, seed=7), scenes.skinned_mesh(1, 1, seed=8)]
    models = [sk.addModel(s["parents"], s["bind"], s["first_nonroot"] if len(s["parents"]) > 1 else -1) for s in skel]
    mesh_ids = [sk.addMesh(v, s) for v, s in meshes]
    rng = np.random.default_rng(2)
    pick = rng.integers(0, 3, size=37)
    sk.setInstances([models[k] for k in pick], [mesh_ids[k] for k in pick])
    poses = [scenes.relative_poses(1, len(skel[k]["parents"]), seed=100 + i) for i, k in enumerate(pick)]
    sk.uploadPoses(np.concatenate([p[0].reshape(-1, 3) for p in poses]), np.concatenate([p[1].reshape(-1, 4) for p in poses]))
    sk.run()
    for i, k in enumerate(pick):
        s = skel[k]
        fn = s["first_nonroot"] if len(s["parents"]) > 1 else 1
        inv = oracle_port.invert_bind(s["bind"])
        apos, arot = oracle_port.pose_compute_absolute(poses[i][0], poses[i][1], s["parents"], fn)
        pal = oracle_port.skin_matrices(apos, arot, inv)
        want = oracle_port.evaluate_skin(meshes[k][0], meshes[k][1], pal)[0]
        assert H.bits_equal(sk.readPalette(i), pal[0]), f"instance {i} palette"
        got = sk.readVertices(i)
        assert close_1e5(got, want), f"instance {i} vertices"
        if exact:
            assert H.bits_equal(got, want), f"instance {i} vertices (exact mode)"
    sk.setMode(False)


def test_skin_config3_slice_properties(gpu_ctx, oracle_port):
    """A slice of BASELINE config 3 (instances x 64 bones x 10 k verts, shared mesh): spot-checked against the oracle,
    plus a linearity property — an identity pose on an identity-bind model leaves vertices where they were."""
    n_inst, n_verts = 512, 10_000
    s = scenes.skeleton(64, seed=4)
    verts, skin = scenes.skinned_mesh(n_verts, 64, seed=6)
    sk = api.Skinning(gpu_ctx)
    model = sk.addModel(s["parents"], s["bind"], s["first_nonroot"])
    mesh = sk.addMesh(verts, skin)
    sk.setInstances([model] * n_inst, [mesh] * n_inst)
    pos, rot = scenes.relative_poses(n_inst, 64, seed=5)
    inv = oracle_port.invert_bind(s["bind"])
    for exact in (True, False):
        sk.setMode(exact)
        sk.uploadPoses(pos, rot)
        sk.run()
        for i in (0, 1, 255, 511):
            apos, arot = oracle_port.pose_compute_absolute(pos[i : i + 1], rot[i : i + 1], s["parents"], s["first_nonroot"])
            want = oracle_port.evaluate_skin(verts, skin, oracle_port.skin_matrices(apos, arot, inv))[0]
            got = sk.readVertices(i)
            assert close_1e5(got, want)
            if exact:
                assert H.bits_equal(got, want)
    # identity: bind = identity, pose = identity -> palette = identity -> weights sum (u16-quantised) scales the point
    ident = np.zeros(64, api.LOCAL_RIGID)
    ident["rot"][:, 3] = 1.0
    sk2 = api.Skinning(gpu_ctx)
    m2 = sk2.addModel(s["parents"], ident, s["first_nonroot"])
    me2 = sk2.addMesh(verts, skin)
    sk2.setInstances([m2], [me2])
    sk2.uploadPoses(np.zeros((64, 3), np.float32), np.tile(np.array([0, 0, 0, 1], np.float32), (64, 1)))
    sk2.run()
    wsum = skin["weights"].sum(axis=1, dtype=np.float32)[:, None]
    assert np.allclose(sk2.readVertices(0), verts * wsum, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("config", ["config3_10k", "config4_100k"])
def test_skin_full_size_digest(gpu_ctx, oracle_port, config):
    """BASELINE config 3 / 4 at their FULL instance counts (10 k / 100 k instances of one 10 k-vertex mesh, 64 bones: 10^8 / 10^9
    vertices). LMX_SKIN_EXACT: sha256 of every skinned position, read back in blocks of 1000 instances, against the digests the
    reference's own pose / palette / evaluateSkin code (model.cpp:103-137, pose.cpp:63-134) produced for the same seeded inputs
    (tests/golden/make_golden_skin_full.py). LMX_SKIN_FUSED: a strided sample that touches every block of instances, <= 1e-5
    relative against the live oracle (the north star's tolerance for skinned positions)."""
    import hashlib
    import json

    g = json.load(open(os.path.join(G, "skin_full.json")))
    rec, n_verts, block = g["configs"][config], g["n_verts"], g["block"]
    n_inst = rec["instances"]
    s = scenes.skeleton(g["n_bones"], seed=4)
    verts, skin = scenes.skinned_mesh(n_verts, g["n_bones"], seed=6)
    pos, rot = scenes.relative_poses(n_inst, g["n_bones"], seed=5)
    assert H.array_digest(s["parents"], s["bind"], verts, skin, pos, rot) == rec["inputs_sha"], "the generators' random streams differ from the ones the digests were made with"
    sk = api.Skinning(gpu_ctx)
    model = sk.addModel(s["parents"], s["bind"], s["first_nonroot"])
    mesh = sk.addMesh(verts, skin)
    sk.setInstances(np.full(n_inst, model, np.uint32), np.full(n_inst, mesh, np.uint32))
    sk.setPoseWriteback(False)  # uploaded relative poses stay valid for the second run
    try:
        sk.setMode(True)
        sk.uploadPoses(pos, rot)
        sk.run()
        whole = hashlib.sha256()
        for k, b in enumerate(range(0, n_inst, block)):
            raw = sk.readVerticesRange(b, min(block, n_inst - b)).tobytes()
            assert hashlib.sha256(raw).hexdigest()[:16] == rec["block_sha16"][k], f"{config}: instances [{b}, {b + block}) differ from the reference"
            whole.update(raw)
        assert whole.hexdigest() == rec["sha256"]
        sk.setMode(False)
        sk.run()
        inv = oracle_port.invert_bind(s["bind"])
        stride = 97 if n_inst <= 10_000 else 997
        for i in list(range(0, n_inst, stride)) + [n_inst - 1]:
            apos, arot = oracle_port.pose_compute_absolute(pos[i : i + 1], rot[i : i + 1], s["parents"], s["first_nonroot"])
            want = oracle_port.evaluate_skin(verts, skin, oracle_port.skin_matrices(apos, arot, inv))[0]
            assert close_1e5(sk.readVertices(i), want), f"{config}: FUSED instance {i}"
    finally:
        sk.setPoseWriteback(True)
        sk.setMode(api.SKIN_FUSED)


def test_skin_distinct_meshes_sampled_digest(gpu_ctx):
    """BASELINE config 3's distinct-mesh variant (every instance its own 10 k-vertex mesh, scenes.distinct_mesh), 1 500 instances here (bench.py runs the
    stated 10 000 and makes the same comparison): LMX_SKIN_EXACT positions of the sampled instances, bit for bit against the digests the reference's
    own evaluateSkin produced (tests/golden/make_golden_skin_distinct.py); LMX_SKIN_FUSED of the same instances within 1e-5 of the EXACT result."""
    import hashlib
    import json

    g = json.load(open(os.path.join(G, "skin_distinct.json")))
    n_inst = 1500 if os.environ.get("LMX_HOSTSIM") != "1" else 3  # (the simulated device skins three 10 k-vertex meshes in seconds, not 1500)
    s = scenes.skeleton(g["n_bones"], seed=4)
    verts, skin = scenes.skinned_mesh(g["n_verts"], g["n_bones"], seed=6)
    pos, rot = scenes.relative_poses(g["poses_drawn"], g["n_bones"], seed=5)
    assert H.array_digest(s["parents"], s["bind"], verts, skin, pos, rot) == g["inputs_sha"], "the generators' random streams differ from the ones the digests were made with"
    sk = api.Skinning(gpu_ctx)
    model = sk.addModel(s["parents"], s["bind"], s["first_nonroot"])
    meshes = [sk.addMesh(*scenes.distinct_mesh(verts, skin, i)) for i in range(n_inst)]
    sk.setInstances(np.full(n_inst, model, np.uint32), np.array(meshes, np.uint32))
    sk.setPoseWriteback(False)
    sample = [i for i in scenes.DISTINCT_MESH_SAMPLE if i < n_inst]
    assert len(sample) >= min(5, n_inst)
    try:
        sk.setMode(api.SKIN_EXACT)
        sk.uploadPoses(pos[:n_inst], rot[:n_inst])
        sk.run()
        exact = {}
        for i in sample:
            exact[i] = sk.readVertices(i)
            assert hashlib.sha256(np.ascontiguousarray(exact[i], np.float32).tobytes()).hexdigest() == g["instances"][str(i)], f"instance {i} (its own mesh) differs from the reference"
        sk.setMode(api.SKIN_FUSED)
        sk.run()
        for i in sample:
            assert close_1e5(sk.readVertices(i), exact[i]), f"FUSED instance {i}"
    finally:
        sk.setPoseWriteback(True)
        sk.setMode(api.SKIN_FUSED)


@pytest.mark.parametrize("fixture", ["cull_edge.npz", "cull_mixed.npz"])
def test_dynamic_set_matches_golden(gpu_ctx, fixture):
    """Every culling entity bound to a (flat) world: they all live in the dynamic set, get their sphere from the world
    transform on the device, and must produce the reference's visible sets (quirky cells, NaN / inf radii, far cameras,
    several renderable types) without ever being re-binned."""
    g = np.load(os.path.join(G, fixture))
    ent = g["entity"].astype(np.int32)
    n_world = int(ent.max()) + 1
    tr = np.zeros(n_world, api.TRANSFORM)
    tr["rot"][:, 3] = 1.0
    tr["scale"] = 1.0
    start = tr.copy()
    start["pos"][ent] = g["pos"] + 7.0  # culling set is built somewhere else first, then everything "moves"
    w = api.World(gpu_ctx)
    w.build(np.full(n_world, -1, np.int32), start)
    cs = api.CullingSystem(gpu_ctx)
    cs.build(ent, g["type"], start["pos"][ent], np.ones(len(ent), np.float32))
    w.bindCulling(ent, g["radius"])  # radius = model_radius * maximum(1, 1, 1)
    tr["pos"][ent] = g["pos"]
    w.setTransforms(ent, tr[ent])
    w.propagate()
    frusta = np.ascontiguousarray(g["frusta"])
    for start_f in range(0, len(frusta), 8):
        batch = frusta[start_f : start_f + 8]
        res = cs.cull(batch)
        for k in range(len(batch)):
            f = start_f + k
            ids, types = res.all_ids(k)
            H.assert_same_visible(H.sorted_by_type(ids, types), H.sorted_by_type(g[f"vis_ids_{f}"], g[f"vis_types_{f}"]), f"{fixture} frustum {f}")
    # type filter on the dynamic set + host-side reads of device-refreshed values
    res = cs.cull(frusta[:1], type_=0)
    want = H.sorted_by_type(g["vis_ids_0"], g["vis_types_0"])
    assert np.array_equal(np.sort(res.ids(0, 0)), want.get(0, np.zeros(0, np.int32)))
    finite = np.flatnonzero(np.isfinite(g["radius"]))[:5]
    for i in finite:
        assert cs.getRadius(int(ent[i])) == float(g["radius"][i])


def test_world_set_parent_matches_oracle(gpu_ctx, live_oracle):
    """World::setParent (world.cpp:619-701): re-parenting inside and across trees, detaching, cycle rejection; the stored
    locals and the world transforms after the next root move must equal the reference's."""
    oracle_port = live_oracle
    h = scenes.hierarchy_fans(8, 3, 4, seed=21)
    n = len(h["parent"])
    ow, roots, kids = oracle_world(oracle_port, h)
    w = api.World(gpu_ctx)
    w.build(h["parent"], gpu_inputs(ow, h["parent"], roots))
    rng = np.random.default_rng(6)
    # one propagation first, so that the GPU world values of the children are the chain of stored locals (like the oracle's
    # after its own root move)
    first = scenes.random_transforms(rng, len(roots), 2000.0)
    ow.set_transforms(roots, first)
    w.setTransforms(roots, first)
    w.propagate()
    assert H.transforms_bits_equal(w.getTransforms(), ow.get_transforms())
    parent = h["parent"].copy()
    leaves = [e for e in range(n) if e not in set(parent.tolist())]
    moves = [(int(roots[1]), int(leaves[0])), (int(leaves[3]), int(leaves[5])), (-1, int(kids[2])), (int(roots[0]), int(roots[3])), (int(kids[10]), int(leaves[7]))]
    for new_parent, child in moves:
        ow.set_parents(np.array([new_parent], np.int32), np.array([child], np.int32))
        w.setParent(new_parent, child)
        parent[child] = new_parent
        assert H.transforms_bits_equal(w.getTransforms(), ow.get_transforms()), (new_parent, child)
        got, want = w.getLocalTransforms(), ow.get_local_transforms()
        has_parent = parent >= 0
        assert H.transforms_bits_equal(got[has_parent], want[has_parent]), (new_parent, child)
    with pytest.raises(api.LumixError):  # a node cannot become a child of its own descendant
        w.setParent(int(leaves[0]), int(roots[1]))
    cur_roots = np.flatnonzero(parent < 0).astype(np.int32)
    new_root = scenes.random_transforms(rng, len(cur_roots), 3000.0)
    ow.set_transforms(cur_roots, new_root)
    w.setTransforms(cur_roots, new_root)
    w.propagate()
    assert H.transforms_bits_equal(w.getTransforms(), ow.get_transforms())


def test_position_only_binding_keeps_radius(gpu_ctx, oracle_port):
    """Lights / decals: onPointLightMoved / onDecalMoved call CullingSystem::setPosition (render_module.cpp:1568-1592)."""
    n = 3000
    rng = np.random.default_rng(12)
    tr = scenes.random_transforms(rng, n, 1500.0)
    radius = rng.uniform(1.0, 60.0, n).astype(np.float32)
    types = np.full(n, 2, np.uint8)  # LOCAL_LIGHT
    w = api.World(gpu_ctx)
    w.build(np.full(n, -1, np.int32), tr)
    cs = api.CullingSystem(gpu_ctx)
    ent = np.arange(n, dtype=np.int32)
    cs.build(ent, types, tr["pos"], radius)
    ocs = oracle_port.culling_system()
    ocs.add_bulk(ent, types, tr["pos"], radius)
    w.bindCulling(ent, np.full(n, -1.0, np.float32))
    moved = scenes.random_transforms(rng, n, 1500.0)
    w.setTransforms(ent, moved)
    w.propagate()
    for e in range(n):
        ocs.set_position(e, moved["pos"][e])
    fr = H.frusta(api, names=["origin_identity", "origin_yaw_pitch"])
    res = cs.cull(fr)
    for f in range(len(fr)):
        ids, tys, _ = ocs.cull(fr[f : f + 1])
        got_ids, got_types = res.all_ids(f)
        H.assert_same_visible(H.sorted_by_type(got_ids, got_types), H.sorted_by_type(ids, tys), f"frustum {f}")
    assert cs.getRadius(17) == float(radius[17])


@pytest.mark.parametrize("seed", [0, 4, 8, 13, 14, 17])
def test_skin_fuzz(gpu_ctx, oracle_port, seed):
    """tests/fuzz_skin.py: random skeletons (1..196 bones), meshes on and around every tile size of both vertex kernels, meshes that
    use all bones / a handful / one limb at a time, instance tables of runs and singles; palettes bit-exact, vertices bit-exact
    (LMX_SKIN_EXACT) or within 1e-5 (pose.cpp:63-134, model.cpp:103-137)."""
    from tests import fuzz_skin

    st = fuzz_skin.run(seed, oracle_port, ctx=gpu_ctx)
    assert st["checked"] >= 3


@pytest.mark.parametrize("seed", [0, 3, 4, 7, 9, 11])
def test_world_fuzz(gpu_ctx, oracle_port, seed):
    """tests/fuzz_world.py: random forests (depth <= 12), frames of interleaved root / child-local / child-world writes with duplicates,
    re-parenting (incl. rejected cycles) between frames, both propagation forms, the moved-entity hand-back - world and stored local
    transforms bit-exact after every step (world.cpp:255-282,337-342,619-753)."""
    from tests import fuzz_world

    st = fuzz_world.run(seed, 8, oracle_port, ctx=gpu_ctx)
    assert st["entities"] >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("world_size", [2, 4])
@pytest.mark.parametrize("kind", ["chains", "fans"])
def test_world_sharded_by_root_union_equals_unsharded(oracle_port, world_size, kind):
    """SURVEY.md 8e, second row: the hierarchy partitioned by ROOT (distributed.shard_by_root: a subtree never crosses GPUs), every rank
    a compact world + culling set of its own with local entity indices, no exchange for the transforms. Three frames of root moves:
    each rank's world transforms are bit-identical to the unsharded World's for its entities, and the ranks' visible lists - local ids
    translated through the rank's table, the way a consumer of the gathered records does - are disjoint and add up to the unsharded
    visible set (one context per rank on the one device; the all-gather of such records has its own tests)."""
    from lumixengine_amd import distributed as D

    h = scenes.hierarchy_chains(1500, 4, seed=14, root_extent=1500.0) if kind == "chains" else scenes.hierarchy_fans(40, 5, 3, seed=15, root_extent=1500.0)
    parent = h["parent"]
    n = len(parent)
    rng = np.random.default_rng(71)
    model_radius = rng.uniform(0.5, 40.0, n).astype(np.float32)
    ow, roots, kids = oracle_world(oracle_port, h)
    ocs = oracle_port.culling_system()
    tr0 = ow.get_transforms()
    ent = np.arange(n, dtype=np.int32)
    r0 = model_radius * tr0["scale"].max(axis=1)
    ocs.add_bulk(ent, np.zeros(n, np.uint8), tr0["pos"], r0)
    ow.bind_culling(ocs, ent, model_radius)
    inputs = gpu_inputs(ow, parent, roots)
    fr = np.concatenate([api.viewport_frustum(pos=(0, 0, 2000.0)), api.viewport_frustum(pos=(300.0, 50.0, -100.0), rot=H.quat_from_yaw_pitch(1.0, 0.1))])

    ranks = []
    try:
        owned = np.zeros(n, np.int32)
        for r in range(world_size):
            nodes, lparent = D.shard_by_root(parent, world_size, r)
            owned[nodes] += 1
            assert np.all((lparent < 0) == (parent[nodes] < 0)) and np.array_equal(nodes[lparent[lparent >= 0]], parent[nodes][lparent >= 0])
            ctx = api.Context(0)
            w = api.World(ctx)
            w.build(lparent, inputs[nodes])
            cs = api.CullingSystem(ctx)
            local = np.arange(len(nodes), dtype=np.int32)
            cs.build(local, np.zeros(len(nodes), np.uint8), tr0["pos"][nodes], r0[nodes])
            w.bindCulling(local, model_radius[nodes])
            my_roots = np.flatnonzero(lparent < 0).astype(np.int32)
            ranks.append(dict(ctx=ctx, w=w, cs=cs, nodes=nodes, my_roots=my_roots))
        assert np.all(owned == 1), "every entity lives on exactly one rank"
        assert abs(len(ranks[0]["my_roots"]) - len(roots) / world_size) <= 1
        for frame in range(3):
            extent = 1200.0 if frame % 2 else 30.0
            new_root = ow.get_transforms()[roots]
            new_root["pos"] += rng.uniform(-extent, extent, size=(len(roots), 3))
            new_root["scale"] = rng.uniform(0.5, 2.0, size=(len(roots), 3)).astype(np.float32)
            ow.set_transforms(roots, new_root)
            want_tr = ow.get_transforms()
            by_scene_index = np.zeros(n, new_root.dtype)
            by_scene_index[roots] = new_root
            seen = [[] for _ in range(len(fr))]
            for rk in ranks:
                rk["w"].setTransforms(rk["my_roots"], by_scene_index[rk["nodes"][rk["my_roots"]]])
                rk["w"].propagate()
                assert H.transforms_bits_equal(rk["w"].getTransforms(), want_tr[rk["nodes"]]), f"frame {frame}: a rank's world transforms"
                res = rk["cs"].cull(fr)
                for f in range(len(fr)):
                    seen[f].append(rk["nodes"][res.all_ids(f)[0]])
            for f in range(len(fr)):
                want_ids, _, _ = ocs.cull(fr[f : f + 1])
                got = np.concatenate(seen[f])
                assert len(np.unique(got)) == len(got), "ranks' lists overlap"
                assert np.array_equal(np.sort(got), np.sort(want_ids)) and len(got) > 0, f"frame {frame} frustum {f}: union of the ranks' lists != unsharded"
    finally:
        for rk in ranks:
            del rk["w"], rk["cs"]
            rk["ctx"].close()
