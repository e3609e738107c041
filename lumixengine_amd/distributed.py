"""Host-side helpers of the multi-GPU path: how a scene is partitioned over ranks and how the gathered records are read.

The exchange itself is native: `lmx_exchange_*` in the C ABI (csrc/lmx_capi_exchange.hip: one ncclAllGather per frame on a side
stream, record = [LMX_MAX_TYPES counts | ids_per_rank ids] per rank, written by the cull's gather kernels). What stays here is
pure numpy: SURVEY.md 8e's partition rules and the record format, shared by bench.py and the CPU tests (tests/test_distributed.py
runs them over gloo with the oracle standing in for the per-rank cull).

* culling shards by CELL (an entity's visibility depends only on the frustum and its own cell): `shard_by_cell`, the reference's
  CellIndicesHasher so that all entities of a cell land on the same rank;
* skinned instances and hierarchy roots shard by index with no exchange: `shard_by_index`;
* the transform hierarchy shards by ROOT (a subtree never crosses GPUs, world.cpp:255-282 walks one subtree per write): `shard_by_root`
  gives a rank its subtrees as a compact world of its own - local entity indices, as every `World` of the engine numbers its own
  entities - and the table back to the scene's indices; a rank's visible list then carries local ids and the record's position in the
  gathered buffer says whose table translates them.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

CELL_SIZE = np.float32(300.0)
MAX_TYPES = 8


def cell_hash(pos: np.ndarray) -> np.ndarray:
    """CellIndicesHasher of the reference (culling_system.cpp:43-50) on IVec3(pos * (1 / 300.f)) — all entities of one
    cell land on the same rank, so per-cell work is never duplicated across GPUs."""
    inv = np.float64(np.float32(1.0) / CELL_SIZE)
    idx = np.trunc(pos * inv).astype(np.int64)
    h = (idx[:, 0] * 73856093 + idx[:, 1] * 19349663 + idx[:, 2] * 83492791) & 0xFFFFFFFF
    return h.astype(np.uint32)


def shard_by_cell(pos: np.ndarray, world_size: int, rank: int) -> np.ndarray:
    """Boolean mask of the entities this rank owns (partition by cell hash mod world_size)."""
    return (cell_hash(pos) % np.uint32(world_size)) == np.uint32(rank)


def shard_by_index(n: int, world_size: int, rank: int) -> np.ndarray:
    """Indices [rank::world_size] — skinned instances / hierarchy roots."""
    return np.arange(rank, n, world_size)


def root_of(parent: np.ndarray) -> np.ndarray:
    """Index of the root above every node (itself for a root): pointer jumping, O(n log depth)."""
    parent = np.asarray(parent, np.int64)
    up = np.where(parent < 0, np.arange(len(parent), dtype=np.int64), parent)
    while True:
        nxt = up[up]
        if np.array_equal(nxt, up):
            return up
        up = nxt


def shard_by_root(parent: np.ndarray, world_size: int, rank: int) -> Tuple[np.ndarray, np.ndarray]:
    """The subtrees this rank owns - roots dealt out by their ordinal among the roots, `shard_by_index` - as a compact hierarchy:
    (nodes, local_parent). `nodes` = the scene indices of the rank's entities in ascending order (local index i is scene entity
    nodes[i]); local_parent[i] = local index of the parent, -1 for a root. No node's parent lives on another rank."""
    parent = np.asarray(parent, np.int64)
    roots = np.flatnonzero(parent < 0)
    ordinal = np.full(len(parent), -1, np.int64)
    ordinal[roots] = np.arange(len(roots))
    mine = (ordinal[root_of(parent)] % world_size) == rank
    nodes = np.flatnonzero(mine)
    local_parent = np.full(len(nodes), -1, np.int32)
    has_parent = parent[nodes] >= 0
    local_parent[has_parent] = np.searchsorted(nodes, parent[nodes][has_parent]).astype(np.int32)
    return nodes.astype(np.int32), local_parent


def make_record(ids_by_type: List[np.ndarray], ids_per_rank: int) -> np.ndarray:
    """One rank's exchange record as lmx_exchange_cull writes it: MAX_TYPES counts (what the rank saw, even beyond the capacity),
    then the ids with the types packed back to back, clipped to `ids_per_rank`."""
    rec = np.zeros(MAX_TYPES + ids_per_rank, np.int32)
    at = 0
    for t in range(MAX_TYPES):
        a = np.asarray(ids_by_type[t], np.int32) if t < len(ids_by_type) else np.zeros(0, np.int32)
        rec[t] = len(a)
        n = max(0, min(len(a), ids_per_rank - at))
        rec[MAX_TYPES + at : MAX_TYPES + at + n] = a[:n]
        at += len(a)
    return rec


def parse_records(records: np.ndarray, ids_per_rank: int) -> Tuple[List[List[np.ndarray]], bool]:
    """records [world, MAX_TYPES + ids_per_rank] (the all-gathered buffer) -> (ids[rank][type], overflowed). `overflowed` is
    True when some rank saw more ids than a record holds (its list is clipped: gather again with a larger capacity)."""
    records = np.asarray(records).reshape(-1, MAX_TYPES + ids_per_rank)
    out, overflowed = [], False
    for rec in records:
        counts = rec[:MAX_TYPES].astype(np.int64)
        overflowed |= int(counts.sum()) > ids_per_rank
        per_type, at = [], 0
        for t in range(MAX_TYPES):
            n = max(0, min(int(counts[t]), ids_per_rank - at))
            per_type.append(rec[MAX_TYPES + at : MAX_TYPES + at + n].copy())
            at += int(counts[t])
        out.append(per_type)
    return out, overflowed


def merge_ranks(parsed: List[List[np.ndarray]]) -> List[np.ndarray]:
    """Per type: concatenation over ranks = the global visible list (compare as a sorted set)."""
    return [np.concatenate([rank[t] for rank in parsed]) if parsed else np.zeros(0, np.int32) for t in range(MAX_TYPES)]


def frame_caps(n_frusta: int, ids_per_rank: int, caps=None) -> List[int]:
    """capacities of a frame's sub-records: as given, else the equal split lmx_exchange_cull_many starts from"""
    return [int(c) for c in caps] if caps is not None else [ids_per_rank // n_frusta] * n_frusta


def cap_for(most_visible: int) -> int:
    """lmx_capi_exchange.hip cap_for: the capacity a list whose largest instance over the ranks holds `most_visible` ids is given -
    20 % head room (at least 64 ids), rounded up to the 256-id grain"""
    m = int(most_visible)
    return (m + max(m // 5, 64) + 255) // 256 * 256


def make_frame_record(ids_by_frustum_and_type: List[List[np.ndarray]], ids_per_rank: int, caps=None) -> np.ndarray:
    """One rank's record of a FRAME of n_frusta views as lmx_exchange_cull_many writes it: n_frusta sub-records
    [MAX_TYPES counts | caps[f] ids] back to back, one collective for all of them (config 5: 8 cascades = one all-gather)."""
    caps = frame_caps(len(ids_by_frustum_and_type), ids_per_rank, caps)
    return np.concatenate([make_record(by_type, c) for by_type, c in zip(ids_by_frustum_and_type, caps)])


def parse_frame_records(records: np.ndarray, n_frusta: int, ids_per_rank: int, caps=None):
    """all-gathered frame records -> (ids[frustum][rank][type], overflowed)."""
    caps = frame_caps(n_frusta, ids_per_rank, caps)
    record = sum(MAX_TYPES + c for c in caps)
    records = np.asarray(records).reshape(-1, record)
    out, overflowed, at = [], False, 0
    for f in range(n_frusta):
        parsed, over = parse_records(records[:, at : at + MAX_TYPES + caps[f]], caps[f])
        out.append(parsed)
        overflowed |= over
        at += MAX_TYPES + caps[f]
    return out, overflowed


def slot_of_last(x) -> int:
    """the slot the exchange's most recent frame went to (frames alternate 0 / 1; lmx_exchange_cull_many returns it - this is for callers that
    only kept the exchange)"""
    return getattr(x, "last_slot", 0)


def config5_frame(api, frusta: np.ndarray, ctx, cs, rank: int, world: int, n_entities_per_rank: int, coll, timed, quiet=None, steps: int = 50) -> dict:
    """BASELINE config 5's frame across GPUs, as bench.py --config5-frame times it: this rank's entities (culling system `cs` of `ctx`)
    under the frame's `frusta` (the 8 shadow cascades) in ONE pass over the spheres (pass width = len(frusta)) and ONE all-gather of
    len(frusta) sub-records (lmx_exchange_cull_many); the reference culls them one after the other (pipeline.cpp:1252-1258).

    `coll` carries the few scalars the ranks must agree on - max_int(x), min_int(x), bcast_bytes(bytes on rank 0 / None elsewhere) - over
    whatever the host has (torch.distributed in bench.py, files in the tests); `timed(fn, steps)` -> ms per call, max over ranks.
    Every rank runs the SAME sequence of collectives whatever happens in between: a failure is carried as a flag and agreed on before the
    timed loop, never raised between two collectives (the peer would wait in the next one for ever)."""
    import contextlib

    n_f = len(frusta)
    out = {"what": f"{n_f} frusta over the rank's entities, pass width {n_f}, lmx_exchange_cull_many: one ncclAllGather of {n_f} sub-records per frame"}
    ok, local, x = 1, None, None
    try:
        cs.setPassWidth(n_f)
        local = cs.cull(frusta, view=2)  # (views 0 / 1 are the exchange's slots)
        per_frustum = local.counts().sum(axis=1)
        out["visible_per_frustum_this_rank"] = [int(v) for v in per_frustum]
    except Exception as e:  # noqa: BLE001
        ok = 0
        out["error"] = repr(e)
    # capacities PER FRUSTUM (round 6): what the largest list of any rank needs, frustum by frustum - config 5's cascades see 535 ... 1.15 M ids on a
    # 10 M-entity rank, one capacity for all of them shipped 4.6x the ids. The exchange keeps them on the lists from here on (regrowth from the
    # gathered counts: no further collective); the buffers are sized for twice the first frame's sum.
    per_frustum = [0] * n_f if local is None else [int(v) for v in local.counts().sum(axis=1)]
    caps = [cap_for(coll.max_int(v)) for v in per_frustum]
    uid = coll.bcast_bytes(api.exchange_unique_id() if rank == 0 else None)
    try:
        with (quiet() if quiet is not None else contextlib.nullcontext()):
            x = api.VisibleExchange(ctx, rank, world, uid, 2 * sum(caps))  # (a collective: every rank gets here)
            x.setCaps(caps)
            slot = x.cullMany(frusta)
            x.wait(slot)
        if ok:
            same = True
            for f in range(n_f):
                _, got = x.readMany(slot, rank, f)
                same = same and np.array_equal(np.sort(got), np.sort(local.all_ids(f)[0]))
            out["own_sub_records_equal_local_cull"] = bool(same)
            out["visible_per_rank_and_frustum"] = [[int(x.readMany(slot, r, f)[0].sum()) for f in range(n_f)] for r in range(world)]
            out["ids_per_rank_and_frustum"] = caps
    except Exception as e:  # noqa: BLE001
        ok = 0
        out["error"] = repr(e)
    if coll.min_int(ok) == 1:
        step = lambda: x.cullMany(frusta)  # noqa: E731
        for _ in range(5):
            step()
        gather_us = None
        if x.info()["mode"] != "p2p":
            gather_us, gather_words = x.timeGather(n_f)  # (a collective, like everything in this branch: the ranks agreed to be here)
            for _ in range(2):
                step()
        ms = timed(step, steps)
        out["ms_per_frame_max_over_ranks"] = ms
        out["entity_frustum_tests_per_sec_all_ranks"] = float(n_f) * n_entities_per_rank * world / (ms * 1e-3)
        st = x.stats(slot_of_last(x))
        out["exchange"] = {"mode": st["mode"], "record_bytes_per_rank": 4 * st["record_words"], "bytes_shipped_per_peer": st["bytes_shipped_per_peer"], "bytes_used_this_rank": st["bytes_used"],
                           "shipped_over_used": round(st["bytes_shipped_per_peer"] / max(st["bytes_used"], 1), 3), "caps": st["caps"], "max_visible_any_rank": st["max_visible"],
                           "overflow_mask": st["overflow_mask"], "gather_us_of_this_record": gather_us,
                           "bytes_arriving_per_rank_per_frame": st["bytes_shipped_per_peer"] * (world - 1) if st["mode"] != "p2p" else None}
    elif "error" not in out:
        out["error"] = "another rank failed"
    if x is not None:
        x.close()
    try:
        cs.setPassWidth(1)
    except Exception:  # noqa: BLE001
        pass
    return out
