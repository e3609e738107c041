"""what a host read of the headline camera's 334 k visible ids costs (cull + lmx_cull_map_all, one host wait): device record + DMA copy (the default above
64 k ids) against the pack kernel writing straight into pinned host memory (LMX_CULL_OPT_MAP_ZERO_COPY with the threshold lifted)"""
import os, sys, time
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
ref = None
for name, value in (("DMA copy (default)", 1), ("zero copy up to 1 M ids", 1 << 20), ("DMA copy (default)", 1), ("zero copy up to 1 M ids", 1 << 20)):
    cs.setOption(api.CULL_OPT_MAP_ZERO_COPY, value)
    for _ in range(20):
        ids, types = cs.cull(fr).map_all(0)
    t0 = time.perf_counter()
    for _ in range(300):
        ids, types = cs.cull(fr).map_all(0)
    dt = (time.perf_counter() - t0) / 300
    key = (np.sort(ids).tobytes(), len(ids))
    if ref is None: ref = key
    print(f"{name:26s}: {1e6 * dt:7.1f} us per cull + host read of {len(ids)} ids, same ids: {key == ref}", flush=True)
