"""kernel time of the multi-frustum cull (8 cascade frusta, pass widths 1 / 4 / 8) on the all-test and the sparse mixed 10 M scenes"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lumixengine_amd import api, scenes
ctx = api.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
N = 10_000_000
fr8 = np.concatenate([api.viewport_frustum(**kw) for kw in scenes.config5_cascade_kwargs()])
scrub = torch.zeros(1 << 28, dtype=torch.int32, device="cuda")
for name in ("all_test", "sparse_mixed"):
    sc = scenes.cull_scene(N, 15000.0, seed=2 if name == "all_test" else 4, mixed_types=name != "all_test")
    if name == "all_test":
        sc["radius"] = scenes.all_test_radii(N)
    cs = api.CullingSystem(ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    for width in [int(w) for w in os.environ.get("LMX_CULL8_WIDTHS", "1,4,8").split(",")]:
        cs.setPassWidth(width)
        for _ in range(5):
            cs.cull(fr8)
        vis = cs.cull(fr8).counts().sum(axis=1)
        for cold in (False, True):
            ctx.profile_reset(); ctx.profile_enable(True)
            for _ in range(20):
                if cold:
                    scrub.sum()
                cs.cull(fr8)
            ctx.synchronize(); ctx.profile_enable(False)
            ms, n = ctx.profile_get(api.K_CULL_SPHERES)
            print(f"{name:13s} width {width}  {'cold' if cold else 'warm'}  {1e3 * ms / 20:8.2f} us per 8-frusta call ({n // 20} launches)  visible {int(vis.sum())}", flush=True)
    cs.setPassWidth(1)
    del cs
