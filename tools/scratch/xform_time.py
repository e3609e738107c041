"""transform step (scatter of 250 k roots + propagate of 750 k children, config 3's hierarchy): one launch per propagation (k_xform_subtree,
LMX_WORLD_OPT_FUSED_LEVELS 1) against one per level, with and without the moved list and the culling binding (config 3 binds every entity)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lumixengine_amd import api, scenes
ctx = api.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
h = scenes.hierarchy_chains(250_000, 4, seed=2)
n = len(h["parent"])
roots = np.flatnonzero(h["parent"] < 0).astype(np.int32)
new_root = scenes.random_transforms(np.random.default_rng(1), len(roots), 4000.0)
d_ent = torch.from_numpy(roots).cuda()
d_tr = torch.from_numpy(new_root.view(np.uint8).reshape(len(roots), -1)).cuda()
for track, bind in ((0, 0), (1, 0), (0, 1)):
    for fused in (0, 1, 0, 1):
        w = api.World(ctx)
        w.setOption(api.WORLD_OPT_FUSED_LEVELS, fused)
        w.trackMoved(bool(track))
        w.build(h["parent"], h["local"])
        cs = None
        if bind:
            cs = api.CullingSystem(ctx)
            ent = np.arange(n, dtype=np.int32)
            rng = np.random.default_rng(3)
            cs.build(ent, np.zeros(n, np.uint8), rng.uniform(-6000.0, 6000.0, size=(n, 3)), np.ones(n, np.float32))
            w.bindCulling(ent, rng.uniform(0.5, 20.0, n).astype(np.float32))

        def step():
            w.setTransformsDevice(len(roots), d_ent.data_ptr(), d_tr.data_ptr())
            w.propagate()

        def frames(k):  # (the moved list holds two propagations: read it in between)
            for i in range(k):
                step()
                if track and (i & 1):
                    w.readMoved()

        frames(10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frames(200)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / 200
        ctx.profile_reset(); ctx.profile_enable(True)
        frames(20)
        ctx.synchronize(); ctx.profile_enable(False)
        t_l, n_l = ctx.profile_get(api.K_XFORM_LEVEL)
        t_s, n_s = ctx.profile_get(api.K_SPHERE_REFRESH)
        print(f"fused {fused} track_moved {track} bound {bind}: {ms * 1e3:7.2f} us per step{' (incl. a host read every other step)' if track else ''}, propagation kernel(s) {1e3 * t_l / 20:7.2f} us in {n_l // 20} launch(es)"
              f", sphere refresh kernel {1e3 * t_s / 20:6.2f} us, {156.0 * (n - len(roots)) / ((t_l + t_s) / 20 * 1e-3) / 1e9:7.1f} GB/s algorithmic on the kernels", flush=True)
        w.trackMoved(False)
        w.setOption(api.WORLD_OPT_FUSED_LEVELS, 1)
        del w, cs
