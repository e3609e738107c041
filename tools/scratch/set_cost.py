import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lumixengine_amd import api, scenes
ctx = api.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
N = 10_000_000
sc = scenes.cull_scene(N, 15000.0, seed=2)
cs = api.CullingSystem(ctx)
cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
fr = api.viewport_frustum()
cs.cull(fr)
rng = np.random.default_rng(3)
perm = rng.permutation(N).astype(np.int32)
for n in (500, 5000, 50000):
    for rep in range(3):
        ids = perm[rep * n:(rep + 1) * n]
        pos = np.ascontiguousarray(sc["pos"][ids])
        r = np.ascontiguousarray(sc["radius"][ids])
        t0 = time.perf_counter(); cs.setMany(ids, pos, r); t1 = time.perf_counter()
        ctx.check(cs.lib.lmx_cull_flush(ctx.h)); t2 = time.perf_counter()
        print(n, rep, "set us %.1f (%.0f ns each) flush us %.1f" % ((t1 - t0) * 1e6, (t1 - t0) * 1e9 / n, (t2 - t1) * 1e6), cs.updateStats())
ctx.synchronize()
