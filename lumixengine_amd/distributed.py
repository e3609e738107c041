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


class VisibleExchange:
    """Steady-state exchange of visible-id lists with ONE collective per frame and no host synchronisation.

    Each rank's cull writes `[counts (n_counts int32) | ids]` into one contiguous device buffer (lmx_cull_bind_output can
    point the kernel at it), and a frame sends the first `n_counts + cap` words of it with a single all-gather. `cap` is a
    capacity chosen from earlier frames (visible sets change slowly from frame to frame), so the payload size never
    depends on this frame's counts and the host never waits for them; `overflowed()` reports afterwards whether some
    rank had more than `cap` visible ids, in which case the caller re-gathers that frame with `allgather_visible`.
    Two buffers are used alternately and the collective runs asynchronously (`async_op=True`, RCCL's own stream), so the
    next frame's cull overlaps this frame's exchange: throughput is max(cull, exchange) instead of their sum.
    """

    def __init__(self, n_counts: int, row: int, cap: int, device, group=None, dtype=torch.int32):
        self.group = group
        self.world = dist.get_world_size(group)
        self.n_counts, self.row, self.cap = n_counts, row, min(cap, row)
        self.send = [torch.zeros(n_counts + row, dtype=dtype, device=device) for _ in range(2)]
        self.recv = [torch.empty(self.world * (n_counts + self.cap), dtype=dtype, device=device) for _ in range(2)]
        self.send_head = [s[: n_counts + self.cap] for s in self.send]  # what a frame ships: sliced once, not per frame
        self.work = [None, None]
        self.k = 0

    def buffer(self):
        """(index, send buffer) of the frame about to be culled; waits (on the stream, not the host) for the exchange that
        last read this buffer."""
        i = self.k & 1
        if self.work[i] is not None:
            self.work[i].wait()
            self.work[i] = None
        return i, self.send[i]

    def exchange(self, i: int):
        self.work[i] = dist.all_gather_into_tensor(self.recv[i], self.send_head[i], group=self.group, async_op=True)
        self.k += 1

    def finish(self):
        for i in range(2):
            if self.work[i] is not None:
                self.work[i].wait()
                self.work[i] = None

    def gathered(self, i: int):
        """[world, n_counts + cap] view of the last completed exchange of buffer i (counts first, then ids)."""
        return self.recv[i].view(self.world, self.n_counts + self.cap)

    def overflowed(self, i: int, count_index: int = 0) -> bool:
        return bool((self.gathered(i)[:, count_index] > self.cap).any().item())


def concat_visible(gathered: Sequence[Sequence[torch.Tensor]]) -> List[torch.Tensor]:
    """Per frustum: concatenation over ranks = the global visible list (compare as a sorted set)."""
    return [torch.cat(list(per_rank)) if len(per_rank) else torch.empty(0, dtype=torch.int32) for per_rank in gathered]
