"""pass width 1 (what the automatic rule picks for sets above 1 M spheres) against one launch for all frusta, on camera-like workloads where most tiles are
rejected at tile level: 10 M sparse scene under 8 small cascades (tests/helpers.cascade_frusta), under a 6-view frame (main view + 4 cascades + a light query),
and config 5's 100 M scene under its 8 cascades"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lumixengine_amd import api, scenes
from tests import helpers as H
ctx = api.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)


def run(name, cs, fr):
    for width in (1, 4, len(fr)):
        cs.setPassWidth(width)
        for _ in range(5):
            cs.cull(fr)
        vis = int(cs.cull(fr).counts().sum())
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(40):
            cs.cull(fr)
        ctx.synchronize(); ctx.profile_enable(False)
        ms, n = ctx.profile_get(api.K_CULL_SPHERES)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(200):
            cs.cull(fr)
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / 200
        print(f"{name:34s} width {width}: kernels {1e3 * ms / 40:7.2f} us in {n // 40} launches, wall {1e6 * wall:7.2f} us per call, visible {vis}", flush=True)
    cs.setPassWidth(0)


N = 10_000_000
sc = scenes.cull_scene(N, 15000.0, seed=4, mixed_types=True)
cs = api.CullingSystem(ctx)
cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
run("10 M, 8 small cascades", cs, H.cascade_frusta(api, 8))
six = np.concatenate([api.viewport_frustum(), H.cascade_frusta(api, 4), api.viewport_frustum(pos=(300.0, 40.0, 900.0), rot=(0.0, 0.38268343, 0.0, 0.92387953), far=600.0)])
run("10 M, a frame's 6 views", cs, six)
del cs, sc
NB = 100_000_000
half = 15000.0 * (NB / N) ** (1.0 / 3.0)
scb = scenes.cull_scene(NB, half, seed=7, mixed_types=True)
csb = api.CullingSystem(ctx)
csb.build(scb["entity"], scb["type"], scb["pos"], scb["radius"])
del scb
run("100 M, config 5's 8 cascades", csb, np.concatenate([api.viewport_frustum(**kw) for kw in scenes.config5_cascade_kwargs()]))
run("100 M, a frame's 6 views", csb, six)
