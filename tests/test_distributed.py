"""world_size-2 CPU (gloo) test of the multi-GPU path: entities are sharded by cell, every rank culls only its shard
(with the CPU oracle standing in for the per-rank HIP cull, which needs a GPU), the visible lists are all-gathered
exactly as bench.py does over RCCL, and the union must equal the unsharded oracle result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lumixengine_amd import distributed as D
from lumixengine_amd import scenes
from tests import helpers as H


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    from oracle import pyoracle as po

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = po.Oracle("port")
        sc = scenes.cull_scene(60_000, 2500.0, seed=7)
        fr = H.frusta(o, names=["origin_identity", "origin_yaw_pitch", "ortho_cascade_large"])
        mask = D.shard_by_cell(sc["pos"], world, rank)
        cs = o.culling_system()
        cs.add_bulk(sc["entity"][mask], sc["type"][mask], sc["pos"][mask], sc["radius"][mask])
        cap = int(mask.sum())
        ids = torch.zeros((len(fr), cap), dtype=torch.int32)
        counts = torch.zeros(len(fr), dtype=torch.int32)
        for f in range(len(fr)):
            v, _, _ = cs.cull(fr[f : f + 1])
            ids[f, : len(v)] = torch.from_numpy(v)
            counts[f] = len(v)
        gathered = D.allgather_visible(ids, counts)
        merged = D.concat_visible(gathered)
        # steady-state form: one async all-gather of [counts | first cap ids] per frame, double-buffered
        n_counts = len(fr)
        cap_ids = max(int(t.numel()) for per in gathered for t in per) + 5  # the same on every rank
        for f in range(len(fr)):
            x = D.VisibleExchange(n_counts, cap, cap_ids, "cpu")
            for frame in range(3):  # three frames through the two buffers
                i, buf = x.buffer()
                buf[:n_counts] = 0
                buf[f] = counts[f]
                buf[n_counts : n_counts + cap] = ids[f]
                x.exchange(i)
            x.finish()
            g = x.gathered(i)
            assert not x.overflowed(i, f)
            one = torch.cat([g[r, n_counts : n_counts + int(g[r, f])] for r in range(world)])
            assert torch.equal(torch.sort(one).values, torch.sort(merged[f]).values)
            small = D.VisibleExchange(n_counts, cap, 1, "cpu")
            j, buf = small.buffer()
            buf[f] = counts[f]
            small.exchange(j)
            small.finish()
            assert small.overflowed(j, f) == bool(int(torch.stack([t.new_tensor(t.numel()) for t in gathered[f]]).max()) > 1)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{f"f{f}": merged[f].numpy() for f in range(len(fr))}, owned=np.array([cap]))
    finally:
        dist.destroy_process_group()


def test_sharded_cull_allgather_equals_unsharded(tmp_path, oracle_port):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sc = scenes.cull_scene(60_000, 2500.0, seed=7)
    fr = H.frusta(oracle_port, names=["origin_identity", "origin_yaw_pitch", "ortho_cascade_large"])
    cs = oracle_port.culling_system()
    cs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    results = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert sum(int(r["owned"][0]) for r in results) == 60_000  # shards are a partition
    for f in range(len(fr)):
        want, _, _ = cs.cull(fr[f : f + 1])
        for r in results:  # every rank ends up with the full list
            assert np.array_equal(np.sort(r[f"f{f}"]), np.sort(want))


def test_shard_by_cell_keeps_cells_whole():
    sc = scenes.cull_scene(20_000, 1500.0, seed=9)
    masks = [D.shard_by_cell(sc["pos"], 4, r) for r in range(4)]
    assert np.array_equal(np.sum(masks, axis=0), np.ones(20_000))
    cells = np.trunc(sc["pos"] * np.float64(np.float32(1.0) / np.float32(300.0))).astype(np.int64)
    key = cells[:, 0] * 1_000_003 + cells[:, 1] * 1_009 + cells[:, 2]
    owner = np.argmax(np.stack(masks), axis=0)
    for k in np.unique(key)[:200]:
        assert len(np.unique(owner[key == k])) == 1
    idx = [D.shard_by_index(10, 4, r) for r in range(4)]
    assert sorted(np.concatenate(idx).tolist()) == list(range(10))
