"""The native exchange (lmx_exchange_*: csrc/lmx_capi_exchange.hip) on one GPU: a world of one rank runs the full code path - the
cull's gather kernels write the [8 counts | ids] record, ncclAllGather ships it on the side stream, frames alternate between two
slots - and the gathered record must hold exactly the oracle's visible ids per type. (More ranks: tests/test_distributed.py checks
the partition and the record format over gloo; the 8-GPU run is the driver's.)"""
import numpy as np
import pytest

from lumixengine_amd import api, scenes
from lumixengine_amd import distributed as D
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_exchange_single_rank_matches_oracle(gpu_ctx, oracle_port):
    sc = scenes.cull_scene(200_000, 4000.0, seed=13, mixed_types=True)
    cs = api.CullingSystem(gpu_ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    ocs = oracle_port.culling_system()
    ocs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    cams = H.frusta(api, names=["origin_identity", "origin_yaw_pitch", "narrow_fov", "ortho_cascade_large"])
    want = []
    for f in range(len(cams)):
        ids, types, _ = ocs.cull(cams[f : f + 1])
        want.append([np.sort(ids[types == t]) for t in range(8)])
    cap = max(sum(len(a) for a in w) for w in want) + 100
    x = api.VisibleExchange(gpu_ctx, 0, 1, api.exchange_unique_id(), cap)
    try:
        # pipelined: two frames in flight, read back one frame late (what a renderer consuming last frame's list does)
        slots = []
        for frame in range(8):
            slots.append(x.cull(cams[frame % len(cams)]))
            if frame >= 1:
                counts, ids = x.read(slots[frame - 1], 0)
                parsed, over = D.parse_records(np.concatenate([counts.astype(np.int32), np.pad(ids, (0, cap - len(ids)))])[None, :], cap)
                assert not over
                for t in range(8):
                    assert np.array_equal(np.sort(parsed[0][t]), want[(frame - 1) % len(cams)][t]), (frame, t)
        # the frame's views in ONE collective (config 5: the cascades of a frame are one exchange step): n sub-records of cap // n ids
        big = api.VisibleExchange(gpu_ctx, 0, 1, api.exchange_unique_id(), cap * len(cams))
        try:
            for frame in range(3):  # both slots, the third re-uses the first
                slot_m = big.cullMany(cams)
                for f in range(len(cams)):
                    counts, ids = big.readMany(slot_m, 0, f)
                    at = 0
                    for t in range(8):
                        assert int(counts[t]) == len(want[f][t]) and np.array_equal(np.sort(ids[at : at + int(counts[t])]), want[f][t]), (frame, f, t)
                        at += int(counts[t])
        finally:
            big.close()
        # type filter
        slot = x.cull(cams[0], 2)
        counts, ids = x.read(slot, 0)
        assert counts[2] == len(want[0][2]) and counts.sum() == counts[2] and np.array_equal(np.sort(ids), want[0][2])
    finally:
        x.close()
    # a capacity that is too small: the counts still tell what the rank saw, the ids are clipped, nothing is written out of bounds
    small = api.VisibleExchange(gpu_ctx, 0, 1, api.exchange_unique_id(), 64)
    try:
        slot = small.cull(cams[0])
        counts, ids = small.read(slot, 0)
        assert [int(c) for c in counts] == [len(a) for a in want[0]] and len(ids) == 64
        assert np.all(np.isin(ids, np.concatenate(want[0])))
    finally:
        small.close()
    # a rank that owns nothing (a strong split with fewer occupied cells than ranks): its record still reports zero counts
    cs.build(np.zeros(0, np.int32), np.zeros(0, np.uint8), np.zeros((0, 3)), np.zeros(0, np.float32))
    empty = api.VisibleExchange(gpu_ctx, 0, 1, api.exchange_unique_id(), 64)
    try:
        for _ in range(3):  # both slots
            slot = empty.cull(cams[0])
            counts, ids = empty.read(slot, 0)
            assert int(counts.sum()) == 0 and len(ids) == 0
    finally:
        empty.close()
