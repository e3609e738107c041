"""Multi-GPU sharding of the hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

SURVEY.md §8e: culling shards naturally (an entity's visibility depends only on the frustum and its own cell), so
every rank owns a disjoint set of entities and culls it with no data-path collective; the only exchange step per
frustum is the all-gather of the per-rank visible-id lists (counts first, then the payload padded to the largest
count). Payloads are a few MB at most, so the collective is latency-bound; one fused gather of every frustum's list
is issued instead of one collective per frustum. Skinned instances and hierarchy roots shard by index with no
exchange at all.

The same code runs on CPU tensors with the gloo backend (tests/test_distributed.py, world_size 2).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch
import torch.distributed as dist

CELL_SIZE = np.float32(300.0)


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


def allgather_visible(ids: torch.Tensor, counts: torch.Tensor, group=None) -> List[List[torch.Tensor]]:
    """All-gather of per-rank visible-id lists.

    ids     [F, cap] int32 — row f holds this rank's visible ids of frustum f in ids[f, :counts[f]]
    counts  [F] int32/int64 (same device as ids)
    returns out[f][r] = tensor of rank r's visible ids for frustum f (views into one gathered buffer)
    """
    world = dist.get_world_size(group)
    F = ids.shape[0]
    counts = counts.to(torch.int32).contiguous()
    all_counts = torch.empty(world * F, dtype=torch.int32, device=ids.device)
    dist.all_gather_into_tensor(all_counts, counts, group=group)
    host_counts = all_counts.view(world, F).cpu()  # the one host sync: payload size depends on it
    m = int(host_counts.max())
    if m == 0:
        return [[ids.new_empty(0) for _ in range(world)] for _ in range(F)]
    send = ids[:, :m].contiguous()  # [F, m]
    recv = torch.empty(world * F * m, dtype=ids.dtype, device=ids.device)
    dist.all_gather_into_tensor(recv, send.view(-1), group=group)
    recv = recv.view(world, F, m)
    return [[recv[r, f, : int(host_counts[r, f])] for r in range(world)] for f in range(F)]


def concat_visible(gathered: Sequence[Sequence[torch.Tensor]]) -> List[torch.Tensor]:
    """Per frustum: concatenation over ranks = the global visible list (compare as a sorted set)."""
    return [torch.cat(list(per_rank)) if len(per_rank) else torch.empty(0, dtype=torch.int32) for per_rank in gathered]
