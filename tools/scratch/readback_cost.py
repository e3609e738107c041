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
for _ in range(10):
    cs.cull(fr).all_ids(0)
t = {"cull_enqueue": 0.0, "counts": 0.0, "read_all": 0.0}
n = 100
for _ in range(n):
    t0 = time.perf_counter(); res = cs.cull(fr)
    t1 = time.perf_counter(); c = res.counts()
    t2 = time.perf_counter(); ids, types = res.all_ids(0)
    t3 = time.perf_counter()
    t["cull_enqueue"] += t1 - t0; t["counts"] += t2 - t1; t["read_all"] += t3 - t2
print({k: round(v / n * 1e6, 1) for k, v in t.items()}, "us per frame; visible", len(ids))
import ctypes as C
tt = [0.0, 0.0, 0.0]
p = C.POINTER(C.c_int32)()
cnt = np.zeros(api.MAX_TYPES, np.uint32)
for _ in range(n):
    t0 = time.perf_counter(); res = cs.cull(fr)
    t1 = time.perf_counter(); rc = cs.lib.lmx_cull_map_all(ctx.h, res.view, 0, C.byref(p), api._ptr(cnt))
    t2 = time.perf_counter(); ids, types = res.map_all(0)
    t3 = time.perf_counter()
    tt[0] += t1 - t0; tt[1] += t2 - t1; tt[2] += t3 - t2
print("cull %.1f us, raw lmx_cull_map_all %.1f us, second map_all through the wrapper %.1f us; visible %d" % (tt[0] / n * 1e6, tt[1] / n * 1e6, tt[2] / n * 1e6, len(ids)))
ctx.profile_reset(); ctx.profile_enable(True)
for _ in range(20):
    cs.cull(fr).all_ids(0)
ctx.profile_enable(False)
for k in range(10):
    ms, cnt = ctx.profile_get(k)
    if cnt: print("kernel id", k, "avg us %.1f" % (ms * 1e3 / cnt), "n", cnt)
