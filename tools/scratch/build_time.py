import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lumixengine_amd import api, scenes
ctx = api.Context(0)
for N in (10_000_000,):
    sc = scenes.cull_scene(N, 15000.0, seed=2)
    cs = api.CullingSystem(ctx)
    t0 = time.perf_counter(); cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"]); ctx.synchronize(); t1 = time.perf_counter()
    print(N, "build s %.3f" % (t1 - t0))
    # churn then forced compaction
    ids = np.arange(N, N + 200000, dtype=np.int32)
    rng = np.random.default_rng(1)
    cs.addMany(ids, np.zeros(len(ids), np.uint8), rng.uniform(-15000, 15000, size=(len(ids), 3)), np.full(len(ids), 2.0, np.float32))
    cs.removeMany(np.arange(0, 200000, dtype=np.int32))
    cs.cull(api.viewport_frustum()); ctx.synchronize()
    t0 = time.perf_counter(); cs.compact(); ctx.synchronize(); t1 = time.perf_counter()
    print(N, "compact s %.3f" % (t1 - t0), cs.updateStats())
