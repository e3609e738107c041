"""One rank of the several-ranks-on-one-GPU exchange test (tests/test_gpu_exchange.py::test_exchange_ranks_on_one_gpu_loopback): a process of its
own with its own context on cuda:0, the exchange's collective going through tests/cpp/loopback_rccl.cpp (LMX_RCCL_LIBRARY). Writes
what it read from every rank's record into <dir>/rank<r>.npz; the parent test compares with the oracle.

    python -m tests.exchange_rank <rank> <world> <dir>"""
import os
import sys
import time

import numpy as np


def wait_for(path, timeout=120.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise SystemExit(f"timed out waiting for {path}")
        time.sleep(0.01)
    return open(path, "rb").read()


def shared_uid(api, rank, directory, name):
    """rank 0 draws the communicator id and publishes it (atomically: write + rename), the others pick it up"""
    path = os.path.join(directory, name)
    if rank == 0:
        uid = api.exchange_unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(path + ".tmp", path)
        return uid
    return wait_for(path)


def missing_peer(api, ctx, rank, world, directory, cams, cap):
    """tests/test_gpu_exchange.py::test_p2p_exchange_gives_up_on_a_missing_peer: both ranks create the exchange, rank 1 never steps"""
    x = api.VisibleExchange(ctx, rank, world, shared_uid(api, rank, directory, "uid_m"), cap)
    try:
        if rank == 0:
            t0 = time.time()
            slot = x.cull(cams[0])
            try:
                x.wait(slot)
                print("rank 0: the step completed without its peer?!")
                sys.exit(3)
            except api.LumixError as e:
                assert e.code == api.ERR_BUSY, e
                print(f"rank 0: step gave up with LMX_ERR_BUSY after {time.time() - t0:.2f} s: {e}")
            try:
                x.cull(cams[0])
                print("rank 0: a failed P2P exchange accepted another step")
                sys.exit(4)
            except api.LumixError as e:
                assert e.code == api.ERR_BUSY, e
                print("rank 0: the exchange refuses further steps")
        else:
            time.sleep(1.5)  # stays away from the step; leaves after rank 0 has given up
    finally:
        x.close()
    ctx.close()


def main():
    rank, world, directory = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    from lumixengine_amd import api, scenes
    from lumixengine_amd import distributed as D
    from tests import helpers as H

    ctx = api.Context(0)
    sc = scenes.cull_scene(120_000, 4000.0, seed=13, mixed_types=True)
    mine = D.shard_by_cell(sc["pos"], world, rank)
    cs = api.CullingSystem(ctx)
    cs.build(sc["entity"][mine], sc["type"][mine], sc["pos"][mine], sc["radius"][mine])
    cams = H.frusta(api, names=["origin_identity", "origin_yaw_pitch", "narrow_fov", "ortho_cascade_large"])
    out = {"owned": sc["entity"][mine]}
    cap = 120_000  # the whole scene fits: nothing is clipped in the first two exchanges
    if len(sys.argv) > 4 and sys.argv[4] == "missing_peer":
        return missing_peer(api, ctx, rank, world, directory, cams, cap)
    x = api.VisibleExchange(ctx, rank, world, shared_uid(api, rank, directory, "uid_a"), cap)
    try:
        want_mode = os.environ.get("LMX_EXPECT_EXCHANGE_MODE")
        slots = []
        for frame in range(6):  # pipelined: read one frame late, both slots re-used twice
            slots.append(x.cull(cams[frame % len(cams)]))
            if want_mode and frame == 0:  # the parent test says what lmx_exchange_info must report once a frame has run (e.g. "side" when the double's gathers are slow and the mode is "auto")
                assert x.info()["mode"] == want_mode, x.info()
            if frame >= 1:
                for r in range(world):
                    counts, ids = x.read(slots[frame - 1], r)
                    out[f"single_f{frame - 1}_r{r}_counts"], out[f"single_f{frame - 1}_r{r}_ids"] = counts, ids
        slot = x.cull(cams[0], 2)  # type filter
        for r in range(world):
            counts, ids = x.read(slot, r)
            out[f"type2_r{r}_counts"], out[f"type2_r{r}_ids"] = counts, ids
    finally:
        x.close()
    big = api.VisibleExchange(ctx, rank, world, shared_uid(api, rank, directory, "uid_b"), cap * len(cams))
    try:
        for frame in range(3):  # all of a frame's frusta in one collective
            slot = big.cullMany(cams)
            for r in range(world):
                for f in range(len(cams)):
                    counts, ids = big.readMany(slot, r, f)
                    out[f"many_f{frame}_r{r}_c{f}_counts"], out[f"many_f{frame}_r{r}_c{f}_ids"] = counts, ids
    finally:
        big.close()
    # capacities per frustum that follow the lists: frustum 0 starts far too small (256 ids), the others far too large. Frames 0 and 1 (one per
    # slot) are clipped and flagged; frame 2 - the next user of slot 0 - has regrown frustum 0 from what EVERY rank gathered in frame 0 and shrunk
    # the others, with no collective besides the frame's own; all ranks must arrive at the same layout (the parent test compares them).
    grow = api.VisibleExchange(ctx, rank, world, shared_uid(api, rank, directory, "uid_d"), cap * len(cams))
    try:
        grow.setCaps([256] + [cap] * (len(cams) - 1))
        for frame in range(5):
            slot = grow.cullMany(cams)
            st = grow.stats(slot)
            out[f"grow_f{frame}_caps"] = np.array(st["caps"], np.int64)
            out[f"grow_f{frame}_max"] = np.array(st["max_visible"], np.int64)
            out[f"grow_f{frame}_misc"] = np.array([st["overflow_mask"], st["record_words"], st["used_words_own"], st["used_words_max"], st["bytes_shipped_per_peer"], st["bytes_used"]], np.int64)
            for r in range(world):
                for f in range(len(cams)):
                    counts, ids = grow.readMany(slot, r, f)
                    out[f"grow_f{frame}_r{r}_c{f}_counts"], out[f"grow_f{frame}_r{r}_c{f}_ids"] = counts, ids
    finally:
        grow.close()
    small = api.VisibleExchange(ctx, rank, world, shared_uid(api, rank, directory, "uid_c"), 64)  # too small: counts tell, ids are clipped
    try:
        slot = small.cull(cams[0])
        for r in range(world):
            counts, ids = small.read(slot, r)
            out[f"small_r{r}_counts"], out[f"small_r{r}_ids"] = counts, ids
    finally:
        small.close()
    np.savez(os.path.join(directory, f"rank{rank}.npz"), **out)
    ctx.close()


if __name__ == "__main__":
    main()
