"""world_size-2 and -4 CPU (gloo) tests of the multi-GPU path: entities are sharded by cell, every rank culls only its shard
(the CPU oracle stands in for the per-rank HIP cull, which needs a GPU), writes the exchange record of
lmx_exchange_cull ([8 counts | ids, types packed]) and the records are all-gathered as one fixed-size collective per frame -
the same record format, capacity / overflow rule and partition that bench.py runs natively over RCCL
(csrc/lmx_capi_exchange.hip); the union must equal the unsharded oracle result."""
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

N, HALF = 60_000, 2500.0
CAMS = ["origin_identity", "origin_yaw_pitch", "ortho_cascade_large"]


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
        sc = scenes.cull_scene(N, HALF, seed=7, mixed_types=True)
        fr = H.frusta(o, names=CAMS)
        mask = D.shard_by_cell(sc["pos"], world, rank)
        cs = o.culling_system()
        cs.add_bulk(sc["entity"][mask], sc["type"][mask], sc["pos"][mask], sc["radius"][mask])
        out = {"owned": np.array([int(mask.sum())])}
        for f in range(len(fr)):
            v, t, _ = cs.cull(fr[f : f + 1])
            by_type = [v[t == k] for k in range(D.MAX_TYPES)]
            # frame 1: a capacity that is too small on purpose -> every rank must see the overflow and gather again
            small = 4
            send = torch.from_numpy(D.make_record(by_type, small))
            recv = torch.empty(world * len(send), dtype=torch.int32)
            dist.all_gather_into_tensor(recv, send)
            _, overflowed = D.parse_records(recv.numpy(), small)
            counts = recv.view(world, -1)[:, : D.MAX_TYPES].sum(dim=1)
            assert overflowed == bool((counts > small).any())
            # frame 2: capacity from the counts every rank now knows (1.25 x the largest, as bench.py sizes it)
            cap = int(counts.max()) * 5 // 4 + 8
            send = torch.from_numpy(D.make_record(by_type, cap))
            recv = torch.empty(world * len(send), dtype=torch.int32)
            dist.all_gather_into_tensor(recv, send)
            parsed, overflowed = D.parse_records(recv.numpy(), cap)
            assert not overflowed and len(parsed) == world
            merged = D.merge_ranks(parsed)
            for k in range(D.MAX_TYPES):
                out[f"f{f}_t{k}"] = merged[k]
        # the frame's views in ONE collective (lmx_exchange_cull_many): n_frusta sub-records of ids_per_rank // n_frusta ids each
        per_frustum = []
        for f in range(len(fr)):
            v, t, _ = cs.cull(fr[f : f + 1])
            per_frustum.append([v[t == k] for k in range(D.MAX_TYPES)])
        cap = (max(sum(len(a) for a in bt) for bt in per_frustum) * 5 // 4 + 8) * len(fr)
        t_cap = torch.tensor([cap], dtype=torch.int64)
        dist.all_reduce(t_cap, op=dist.ReduceOp.MAX)  # every rank the same capacity (bench.py: all_reduce MAX of the probe counts)
        cap = int(t_cap.item())
        send = torch.from_numpy(D.make_frame_record(per_frustum, cap))
        recv = torch.empty(world * len(send), dtype=torch.int32)
        dist.all_gather_into_tensor(recv, send)
        frame, overflowed = D.parse_frame_records(recv.numpy(), len(fr), cap)
        assert not overflowed and len(frame) == len(fr) and all(len(p) == world for p in frame)
        for f in range(len(fr)):
            merged = D.merge_ranks(frame[f])
            for k in range(D.MAX_TYPES):
                assert np.array_equal(np.sort(merged[k]), np.sort(out[f"f{f}_t{k}"])), "one collective per frame != one collective per view"
        # the NEXT frame's capacities follow the lists, frustum by frustum (lmx_capi_exchange.hip `regrow`): every rank derives them from the
        # counts it has just gathered - the largest list of any rank per frustum, + 20 % - so the ranks agree on the new record size WITHOUT
        # another collective (an all-gather of different sizes would not complete), and the record shrinks to what the lists need
        gathered = recv.view(world, -1).numpy()
        at, caps = 0, []
        for f in range(len(fr)):
            caps.append(D.cap_for(int(gathered[:, at : at + D.MAX_TYPES].sum(axis=1).max())))
            at += D.MAX_TYPES + cap // len(fr)
        send2 = torch.from_numpy(D.make_frame_record(per_frustum, cap, caps))
        assert len(send2) == sum(D.MAX_TYPES + c for c in caps)
        recv2 = torch.empty(world * len(send2), dtype=torch.int32)
        dist.all_gather_into_tensor(recv2, send2)
        frame2, overflowed2 = D.parse_frame_records(recv2.numpy(), len(fr), cap, caps)
        assert not overflowed2
        for f in range(len(fr)):
            for r in range(world):
                for k in range(D.MAX_TYPES):
                    assert np.array_equal(frame2[f][r][k], frame[f][r][k])
        out["caps"] = np.array(caps, np.int64)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_cull_allgather_equals_unsharded(tmp_path, oracle_port, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sc = scenes.cull_scene(N, HALF, seed=7, mixed_types=True)
    fr = H.frusta(oracle_port, names=CAMS)
    cs = oracle_port.culling_system()
    cs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    results = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert sum(int(r["owned"][0]) for r in results) == N  # shards are a partition
    assert all(np.array_equal(r["caps"], results[0]["caps"]) for r in results)  # every rank derived the same per-frustum capacities from the gathered counts
    for f in range(len(fr)):
        want, wt, _ = cs.cull(fr[f : f + 1])
        for r in results:  # every rank ends up with the full list, per type
            for k in range(D.MAX_TYPES):
                assert np.array_equal(np.sort(r[f"f{f}_t{k}"]), np.sort(want[wt == k])), (f, k)


def test_record_roundtrip_and_clipping():
    rng = np.random.default_rng(3)
    by_type = [rng.integers(0, 1 << 30, size=n).astype(np.int32) for n in (7, 0, 3, 0, 0, 5, 0, 1)]
    rec = D.make_record(by_type, 32)
    parsed, over = D.parse_records(rec[None, :], 32)
    assert not over and all(np.array_equal(parsed[0][t], by_type[t]) for t in range(8))
    rec = D.make_record(by_type, 9)  # clipped inside type 2
    parsed, over = D.parse_records(rec[None, :], 9)
    assert over and np.array_equal(parsed[0][0], by_type[0]) and np.array_equal(parsed[0][2], by_type[2][:2]) and len(parsed[0][5]) == 0
    assert list(rec[:8]) == [7, 0, 3, 0, 0, 5, 0, 1]  # the counts still tell what the rank saw


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


def test_shard_by_root_keeps_subtrees_whole(oracle_port):
    """the hierarchy shards by root. Random forest with parents on either side of their children in index order; for world
    sizes 1..5 every entity lands on exactly one rank, no parent link leaves a rank, the roots are dealt out evenly, and a rank that
    propagates ITS compact world (the oracle standing in for the kernels) gets, bit for bit, what the unsharded World holds for its entities."""
    from lumixengine_amd import scenes

    rng = np.random.default_rng(5)
    n = 4000
    order = rng.permutation(n)  # a random topological order: node order[i] may only hang under order[< i]
    parent = np.full(n, -1, np.int32)
    for i in range(1, n):
        if rng.random() < 0.85:
            parent[order[i]] = order[rng.integers(0, i)]
    roots = np.flatnonzero(parent < 0).astype(np.int32)
    kids = np.flatnonzero(parent >= 0).astype(np.int32)
    local = scenes.random_transforms(rng, n, 500.0)
    whole = oracle_port.world(n)
    whole.init_transforms(roots, local[roots])
    whole.set_parents(parent[kids], kids)
    whole.set_local_transforms(kids, local[kids])
    new_root = scenes.random_transforms(rng, len(roots), 800.0)
    whole.set_transforms(roots, new_root)
    want = whole.get_transforms()
    by_scene_index = np.zeros(n, new_root.dtype)
    by_scene_index[roots] = new_root
    up = D.root_of(parent)
    assert np.all(parent[up] < 0)
    for world_size in (1, 2, 3, 5):
        owned = np.zeros(n, np.int32)
        for rank in range(world_size):
            nodes, lparent = D.shard_by_root(parent, world_size, rank)
            owned[nodes] += 1
            assert np.all(np.diff(nodes) > 0)
            has = lparent >= 0
            assert np.array_equal(nodes[lparent[has]], parent[nodes][has]) and np.all(parent[nodes][~has] < 0)
            n_roots = int((~has).sum())
            assert abs(n_roots - len(roots) / world_size) < 1.0 + 1e-9
            lroots, lkids = np.flatnonzero(~has).astype(np.int32), np.flatnonzero(has).astype(np.int32)
            w = oracle_port.world(len(nodes))
            w.init_transforms(lroots, local[nodes][lroots])
            w.set_parents(lparent[lkids], lkids)
            w.set_local_transforms(lkids, local[nodes][lkids])
            w.set_transforms(lroots, by_scene_index[nodes[lroots]])
            got = w.get_transforms()
            for k in ("pos", "rot", "scale"):
                assert np.ascontiguousarray(got[k]).tobytes() == np.ascontiguousarray(want[k][nodes]).tobytes(), (world_size, rank, k)
        assert np.all(owned == 1)


def test_config5_frame_failure_on_one_rank_is_agreed_on_not_hung():
    """distributed.config5_frame (bench.py --config5-frame): whatever happens on one rank, every rank runs the same sequence of collectives.
    Two ranks as threads over stand-ins for the binding; rank 1's cull fails: both return, nobody enters the timed loop, rank 0 reports that
    another rank failed, rank 1 its own error, and the exchange each one created is closed. With no failure both time the loop."""
    import threading

    class Coll:
        def __init__(self, world):
            self.bar = threading.Barrier(world, timeout=20)
            self.vals = [None] * world
            self.ops = [[] for _ in range(world)]

        def _gather(self, rank, v, op):
            self.ops[rank].append(op)
            self.vals[rank] = v
            self.bar.wait()
            got = list(self.vals)
            self.bar.wait()
            return got

    class RankColl:
        def __init__(self, coll, rank):
            self.c, self.r = coll, rank

        def max_int(self, v):
            return max(self.c._gather(self.r, v, "max"))

        def min_int(self, v):
            return min(self.c._gather(self.r, v, "min"))

        def bcast_bytes(self, b):
            return self.c._gather(self.r, b, "bcast")[0]

    class FakeResult:
        def counts(self):
            return np.array([[3, 0, 0, 0, 0, 0, 0, 0], [5, 0, 0, 0, 0, 0, 0, 0]], np.uint32)

        def all_ids(self, f):
            return np.arange(3 if f == 0 else 5, dtype=np.int32), None

    class FakeCs:
        def __init__(self, fail):
            self.fail, self.widths = fail, []

        def setPassWidth(self, w):
            self.widths.append(w)

        def cull(self, frusta, view=0):
            if self.fail:
                raise RuntimeError("device lost")
            return FakeResult()

    class FakeExchange:
        live = []

        def __init__(self, ctx, rank, world, uid, cap):
            # per-frustum capacities: cap_for(3) = cap_for(5) = 256 ids (a healthy rank's lists; the failing one reports 0 - the MAX over the
            # ranks decides), the buffers hold twice their sum
            assert uid == b"U" * 128 and cap == 2 * (256 + 256)
            self.closed, self.steps, self.caps, self.timed = False, 0, None, 0
            FakeExchange.live.append(self)

        def setCaps(self, caps, keep_fixed=False):
            self.caps = list(caps)

        def info(self):
            return {"mode": "inline", "gather_us": None, "why": "fake"}

        def timeGather(self, n_frusta):
            self.timed += 1
            return 7.0, sum(8 + c for c in self.caps)

        def stats(self, slot):
            words = sum(8 + c for c in self.caps)
            return {"mode": "inline", "record_words": words, "bytes_shipped_per_peer": 4 * words, "bytes_used": 4 * (8 + 3 + 8 + 5), "caps": self.caps, "max_visible": [3, 5],
                    "overflow_mask": 0}

        def cullMany(self, frusta):
            self.steps += 1
            self.last_slot = 0
            return 0

        def wait(self, slot):
            pass

        def readMany(self, slot, rank, f):
            n = 3 if f == 0 else 5
            return np.array([n, 0, 0, 0, 0, 0, 0, 0], np.uint32), np.arange(n, dtype=np.int32)[::-1].copy()

        def close(self):
            self.closed = True

    class FakeApi:
        VisibleExchange = FakeExchange

        @staticmethod
        def exchange_unique_id():
            return b"U" * 128

    for failing in (True, False):
        FakeExchange.live = []
        coll = Coll(2)
        out, css = [None, None], [FakeCs(False), FakeCs(failing)]

        def run(rank):
            out[rank] = D.config5_frame(FakeApi, np.zeros(2), None, css[rank], rank, 2, 1000, RankColl(coll, rank), lambda fn, steps: [fn() for _ in range(steps)] and 0.5, steps=3)

        threads = [threading.Thread(target=run, args=(r,)) for r in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(30)
        assert not any(t.is_alive() for t in threads), "a rank hangs"
        assert coll.ops[0] == coll.ops[1], "the ranks ran different sequences of collectives"
        assert all(x.closed for x in FakeExchange.live) and len(FakeExchange.live) == 2
        assert css[0].widths == [2, 1] and css[1].widths == [2, 1]
        if failing:
            assert out[0]["error"] == "another rank failed" and "device lost" in out[1]["error"]
            assert "ms_per_frame_max_over_ranks" not in out[0] and "ms_per_frame_max_over_ranks" not in out[1]
            assert all(x.steps == 1 for x in FakeExchange.live)
        else:
            for r in range(2):
                assert "error" not in out[r] and out[r]["own_sub_records_equal_local_cull"] is True and out[r]["ms_per_frame_max_over_ranks"] == 0.5
                assert out[r]["visible_per_rank_and_frustum"] == [[3, 5], [3, 5]] and out[r]["entity_frustum_tests_per_sec_all_ranks"] == 2.0 * 1000 * 2 / 0.5e-3
                assert out[r]["exchange"]["caps"] == [256, 256] and out[r]["exchange"]["gather_us_of_this_record"] == 7.0 and out[r]["exchange"]["bytes_used_this_rank"] == 4 * 24
            assert all(x.steps == 1 + 5 + 2 + 3 and x.timed == 1 and x.caps == [256, 256] for x in FakeExchange.live)
