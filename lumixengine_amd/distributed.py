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


def make_frame_record(ids_by_frustum_and_type: List[List[np.ndarray]], ids_per_rank: int) -> np.ndarray:
    """One rank's record of a FRAME of n_frusta views as lmx_exchange_cull_many writes it: n_frusta sub-records
    [MAX_TYPES counts | ids_per_rank // n_frusta ids], one collective for all of them (config 5: 8 cascades = one all-gather)."""
    n = len(ids_by_frustum_and_type)
    cap_f = ids_per_rank // n
    return np.concatenate([make_record(by_type, cap_f) for by_type in ids_by_frustum_and_type])


def parse_frame_records(records: np.ndarray, n_frusta: int, ids_per_rank: int):
    """all-gathered frame records -> (ids[frustum][rank][type], overflowed)."""
    cap_f = ids_per_rank // n_frusta
    sub = MAX_TYPES + cap_f
    records = np.asarray(records).reshape(-1, n_frusta, sub)
    out, overflowed = [], False
    for f in range(n_frusta):
        parsed, over = parse_records(records[:, f, :], cap_f)
        out.append(parsed)
        overflowed |= over
    return out, overflowed
