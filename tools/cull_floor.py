#!/usr/bin/env python
"""Where do the 12 us of the headline cull go? Back-to-back culls of the bench scene (10 M, sparse) for cameras that keep
nothing (every tile rejected by the tile-level test: the kernel's floor), the default camera, and a narrow one. GPU box only."""
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
    from tests import helpers as H

    ctx = api.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 10_000_000
    sc = scenes.cull_scene(n, 15000.0, seed=2)
    cs = api.CullingSystem(ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    cams = {
        "nothing": api.viewport_frustum(pos=(1.0e6, 50.0, -1.0e6)),
        "default": api.viewport_frustum(),
        "narrow": api.viewport_frustum(pos=(-800.0, 120.0, 300.0), rot=H.quat_from_yaw_pitch(-1.3, 0.1), fov=float(np.deg2rad(20.0)), far=4000.0),
        "near_only": api.viewport_frustum(far=1000.0),
    }
    out = []
    cams["all_visible"] = api.viewport_frustum(pos=(0.0, 0.0, 60000.0), far=300000.0)
    variants = [int(v) for v in os.environ.get("LMX_FLOOR_VARIANTS", "1").split(",")]
    modes = [int(v) for v in os.environ.get("LMX_FLOOR_MODES", "1,2").split(",")]
    for variant, tpb in [(v, k) for v in variants for k in modes]:
        cs.setOption(api.CULL_OPT_TILE_VARIANT, variant)
        cs.setOption(api.CULL_OPT_LANE_PARALLEL_TILE_TEST, tpb)
        for name, fr in cams.items():
            for _ in range(20):
                res = cs.cull(fr)
            ctx.synchronize()
            reps = 400
            t0 = time.perf_counter()
            for _ in range(reps):
                res = cs.cull(fr)
            ctx.synchronize()
            wall = (time.perf_counter() - t0) * 1e6 / reps
            ctx.profile_reset()
            ctx.profile_enable(True)
            for _ in range(100):
                res = cs.cull(fr)
            ctx.synchronize()
            ctx.profile_enable(False)
            ms, k = ctx.profile_get(api.K_CULL_SPHERES)
            rec = {"variant": variant, "tile_test_mode": tpb, "camera": name, "visible": int(res.counts().sum()), "wall_us": round(wall, 3), "event_kernel_us": round(ms * 1e3 / max(k, 1), 3)}
            print(rec, flush=True)
            out.append(rec)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cull_floor.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
