#!/usr/bin/env python
"""Host + device cost of the per-frame update stream on the 10 M scene: 1000 removes + 1000 adds (+ 500 sets) per frame through the
batched C-ABI entry points, split into the host-only part (mirror + patch queues), the flush (staging copy + patch kernel enqueue) and
the cull. GPU box only."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from lumixengine_amd import api, scenes

    ctx = api.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    N = 10_000_000
    sc = scenes.cull_scene(N, 15000.0, seed=2)
    cs = api.CullingSystem(ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    fr = api.viewport_frustum()
    rng = np.random.default_rng(3)
    frames = 160
    perm = rng.permutation(N)
    victims = perm[: frames * 1000].astype(np.int32).reshape(frames, 1000)
    movers = perm[-frames * 500:].astype(np.int32).reshape(frames, 500)  # disjoint from the victims
    add_pos = rng.uniform(-15000.0, 15000.0, size=(frames, 1000, 3))
    add_r = np.exp(rng.uniform(np.log(0.5), np.log(50.0), size=(frames, 1000))).astype(np.float32)
    add_t = np.zeros(1000, np.uint8)
    new_ids = [np.arange(N + 1000 * k, N + 1000 * (k + 1), dtype=np.int32) for k in range(frames)]
    mv_pos = sc["pos"][movers] + rng.uniform(-1.0, 1.0, size=(frames, 500, 3))
    mv_r = sc["radius"][movers]
    t = {"remove": 0.0, "add": 0.0, "set": 0.0, "flush": 0.0, "cull": 0.0}
    for k in range(frames):
        timed = k >= 40
        t0 = time.perf_counter(); cs.removeMany(victims[k])
        t1 = time.perf_counter(); cs.addMany(new_ids[k], add_t, add_pos[k], add_r[k])
        t2 = time.perf_counter(); cs.setMany(movers[k], mv_pos[k], mv_r[k])
        t3 = time.perf_counter(); ctx.check(cs.lib.lmx_cull_flush(ctx.h))
        t4 = time.perf_counter(); cs.cull(fr)
        t5 = time.perf_counter()
        if timed:
            for name, d in zip(t, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                t[name] += d
            if os.environ.get("LMX_UPDATE_TRACE") and (t3 - t2) > 200e-6:
                print("frame", k, "remove %.0f add %.0f set %.0f flush %.0f cull %.0f us" % tuple(x * 1e6 for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)), cs.updateStats(), file=sys.stderr)
    ctx.synchronize()
    n = frames - 40
    out = {k: round(v / n * 1e6, 2) for k, v in t.items()}
    out["unit"] = "us per frame (1000 removes, 1000 adds, 500 in-cell sets; host wall clock per call incl. ctypes)"
    out["state"] = cs.updateStats()
    print(json.dumps(out))
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "update_cost.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
