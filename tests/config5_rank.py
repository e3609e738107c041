"""One rank of tests/test_gpu_exchange.py::test_config5_frame_two_ranks_loopback: distributed.config5_frame - what bench.py --config5-frame runs -
with a world of several ranks on one device (loopback collective, as tests/exchange_rank.py), the ranks' scalars agreed on through files.

    python -m tests.config5_rank <rank> <world> <dir>"""
import json
import os
import sys
import time

import numpy as np

from tests.exchange_rank import wait_for


class FileColl:
    """max / min of one int per rank and a broadcast of rank 0's bytes through <dir>: one file per (operation, rank), written atomically"""

    def __init__(self, rank, world, directory):
        self.rank, self.world, self.dir, self.seq = rank, world, directory, 0

    def _gather(self, payload: bytes):
        self.seq += 1
        path = os.path.join(self.dir, f"coll{self.seq}_r{self.rank}")
        with open(path + ".tmp", "wb") as f:
            f.write(payload)
        os.rename(path + ".tmp", path)
        return [wait_for(os.path.join(self.dir, f"coll{self.seq}_r{r}")) for r in range(self.world)]

    def max_int(self, v):
        return max(int(b) for b in self._gather(str(int(v)).encode()))

    def min_int(self, v):
        return min(int(b) for b in self._gather(str(int(v)).encode()))

    def bcast_bytes(self, b):
        return self._gather(b if b is not None else b"-")[0]


def main():
    rank, world, directory = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    from lumixengine_amd import api, scenes
    from lumixengine_amd import distributed as D

    ctx = api.Context(0)
    n = 150_000
    sc = scenes.cull_scene(n, 4000.0, seed=21 + rank, mixed_types=True)  # weak scaling: every rank its own entities (ids offset by rank)
    cs = api.CullingSystem(ctx)
    cs.build(sc["entity"] + rank * n, sc["type"], sc["pos"], sc["radius"])
    frusta = np.concatenate([api.viewport_frustum(**kw) for kw in scenes.config5_cascade_kwargs()])
    coll = FileColl(rank, world, directory)

    def timed(fn, steps):
        coll.max_int(0)  # barrier
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        ctx.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        return coll.max_int(int(ms * 1e6)) / 1e6

    out = D.config5_frame(api, frusta, ctx, cs, rank, world, n, coll, timed, steps=4)
    local = cs.cull(frusta, view=2)  # (pass width is back at 1: the lists must not depend on it)
    out["local_sha_per_frustum"] = [hash(np.sort(local.all_ids(f)[0]).tobytes()) & 0xFFFFFFFF for f in range(len(frusta))]
    with open(os.path.join(directory, f"config5_rank{rank}.json"), "w") as f:
        json.dump(out, f)
    ctx.close()


if __name__ == "__main__":
    main()
