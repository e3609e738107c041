"""Randomised hierarchies and write sequences against the CPU oracle's World (World::setTransform / setLocalTransform / setParent /
transformEntity, src/engine/world.cpp:255-282,337-342,619-753): random forests (depth <= 12, ragged fan-out, lone roots), frames of
interleaved root moves, child-local and child-world writes (issued ancestors-first, the order the batch form defines), duplicate writes
of one entity in a frame (last wins), re-parenting between frames (into other trees, to the root level, rejected cycles), both
propagation forms (per level / one launch), the moved-entity hand-back. World AND stored local transforms bit-exact after every frame.

    python -m tests.fuzz_world [--seeds 0-9] [--frames 8]"""
from __future__ import annotations

import argparse
import sys

import numpy as np

from lumixengine_amd import api, scenes
from tests import helpers as H


def random_forest(rng, n):
    """parent[] with parent index < child index is NOT required by the engine: ids are shuffled afterwards."""
    parent = np.full(n, -1, np.int32)
    depth = np.zeros(n, np.int32)
    max_depth = int(rng.choice([1, 2, 4, 8, 12]))
    for e in range(1, n):
        if rng.random() < 0.15:
            continue
        p = int(rng.integers(max(0, e - int(rng.choice([3, 30, 3000]))), e))
        if depth[p] + 1 >= max_depth:
            continue
        parent[e] = p
        depth[e] = depth[p] + 1
    perm = rng.permutation(n).astype(np.int32)  # new id of old id
    new_parent = np.full(n, -1, np.int32)
    for e in range(n):
        new_parent[perm[e]] = perm[parent[e]] if parent[e] >= 0 else -1
    return new_parent


def depths(parent):
    n = len(parent)
    depth = np.full(n, -1, np.int32)
    for e in range(n):
        chain = []
        x = e
        while x >= 0 and depth[x] < 0:
            chain.append(x)
            x = parent[x]
        d = depth[x] if x >= 0 else -1
        for y in reversed(chain):
            d += 1
            depth[y] = d
    return depth


def run(seed: int, frames: int, oracle, ctx=None, verbose: bool = False) -> dict:
    rng = np.random.default_rng(9000 + seed)
    own = ctx is None
    if own:
        ctx = api.Context(0)
    try:
        n = int(rng.choice([1, 2, 70, 900, 6000]))
        parent = random_forest(rng, n)
        local = scenes.random_transforms(rng, n, 10.0)
        roots = np.flatnonzero(parent < 0).astype(np.int32)
        local["pos"][roots] = rng.uniform(-5000, 5000, size=(len(roots), 3))
        ow = oracle.world(n)
        ow.init_transforms(np.arange(n, dtype=np.int32), local)
        order = np.argsort(depths(parent), kind="stable").astype(np.int32)  # parents before children
        kids_in_order = order[parent[order] >= 0]
        if len(kids_in_order):
            ow.set_parents(parent[kids_in_order], kids_in_order)
            ow.set_local_transforms(kids_in_order, local[kids_in_order])
        w = api.World(ctx)
        fused = int(rng.integers(0, 2))
        w.setOption(api.WORLD_OPT_FUSED_LEVELS, fused)
        w.trackMoved(True)
        w.buildWithWorld(parent, ow.get_local_transforms(), ow.get_transforms())
        w.propagate()
        ent, _ = w.readMoved()
        assert len(ent) == 0
        n_reparent = 0

        def same(tag):
            assert H.transforms_bits_equal(w.getTransforms(), ow.get_transforms()), f"seed {seed} {tag}: world transforms"
            kids = np.flatnonzero(parent >= 0)
            if len(kids):
                assert H.transforms_bits_equal(w.getLocalTransforms()[kids], ow.get_local_transforms()[kids]), f"seed {seed} {tag}: stored locals"

        for frame in range(frames):
            depth = depths(parent)
            k = int(rng.choice([0, 1, max(1, n // 50), max(1, n // 3), n]))
            picked = rng.permutation(n)[:k].astype(np.int32)
            if k and rng.random() < 0.3:  # the same entity written twice in one frame: the last write wins
                picked = np.concatenate([picked, rng.choice(picked, size=min(k, 5))]).astype(np.int32)
            how = rng.integers(0, 2, size=len(picked))  # 0 local (roots: world), 1 world
            tr = scenes.random_transforms(rng, len(picked), 50.0)
            tr["pos"][parent[picked] < 0] *= 60.0
            # the batch semantics: writes are applied in hierarchy order and the LAST write of an entity wins - the reference sees the
            # same thing when the calls come ancestors-first with only each entity's final write
            last = {}
            for i, e in enumerate(picked.tolist()):
                last[e] = i
            final = np.array(sorted(last.values()), np.int64)
            for d in range(int(depth.max()) + 1 if n else 0):
                for i in final[depth[picked[final]] == d]:
                    e = picked[i : i + 1]
                    if how[i] == 1 or parent[e[0]] < 0:
                        ow.set_transforms(e, tr[i : i + 1])
                    else:
                        ow.set_local_transforms(e, tr[i : i + 1])
            # the device gets the calls in the order they were "made", duplicates included; all records of an entity use the kind of
            # write of its LAST one (the one the oracle applied)
            kind = {e: bool(how[i] == 0 or parent[e] < 0) for e, i in last.items()}
            how_local = np.array([kind[e] for e in picked.tolist()], bool) if len(picked) else np.zeros(0, bool)
            if how_local.any():
                w.setTransforms(picked[how_local], tr[how_local])
            if (~how_local).any():
                w.setWorldTransforms(picked[~how_local], tr[~how_local])
            w.propagate()
            same(f"frame {frame}")
            # moved list = the written entities and their subtrees
            ent, mtr = w.readMoved()
            children = [[] for _ in range(n)]
            for c in np.flatnonzero(parent >= 0):
                children[parent[c]].append(int(c))
            want, stack = set(), list(set(picked.tolist()))
            while stack:
                x = stack.pop()
                if x in want:
                    continue
                want.add(x)
                stack += children[x]
            assert set(ent.tolist()) == want, f"seed {seed} frame {frame}: moved list"
            cur = ow.get_transforms()
            seen = {}
            for j, e in enumerate(ent.tolist()):
                seen[e] = j  # an entity listed twice: newest last
            idx = np.array(list(seen.values()), np.int64)
            if len(idx):
                assert H.transforms_bits_equal(mtr[idx], cur[ent[idx]]), f"seed {seed} frame {frame}: moved transforms"
            # re-parenting between frames
            for _ in range(int(rng.choice([0, 0, 1, 4]))):
                if n < 2:
                    break
                child = int(rng.integers(0, n))
                new_parent = -1 if rng.random() < 0.2 else int(rng.integers(0, n))
                x, cyc = new_parent, False
                while x >= 0:
                    if x == child:
                        cyc = True
                        break
                    x = parent[x]
                if cyc or new_parent == child:
                    try:
                        w.setParent(new_parent, child)
                        raise AssertionError(f"seed {seed}: a cycle was accepted")
                    except api.LumixError:
                        continue
                if depths(np.where(np.arange(n) == child, new_parent, parent)).max() > 60:
                    continue
                ow.set_parents(np.array([new_parent], np.int32), np.array([child], np.int32))
                w.setParent(new_parent, child)
                parent[child] = new_parent
                n_reparent += 1
                same(f"frame {frame} setParent({new_parent}, {child})")
        w.trackMoved(False)
        w.setOption(api.WORLD_OPT_FUSED_LEVELS, 0)
        st = {"entities": n, "max_depth": int(depths(parent).max()) if n else 0, "reparented": n_reparent, "fused": fused}
        if verbose:
            print(f"seed {seed}: {st}")
        return st
    finally:
        if own:
            ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0-9")
    ap.add_argument("--frames", type=int, default=8)
    a = ap.parse_args()
    lo, _, hi = a.seeds.partition("-")
    from oracle import pyoracle

    oracle = pyoracle.Oracle("port")
    for seed in range(int(lo), int(hi or lo) + 1):
        run(seed, a.frames, oracle, verbose=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
