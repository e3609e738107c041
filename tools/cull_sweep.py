#!/usr/bin/env python
"""Sweep of the cull kernel's tuning knobs over the three regimes bench.py reports (GPU box only).

    python tools/cull_sweep.py [--entities 10000000] [--out gpurun_out/sweep.json] [--quick]

Scenes (BASELINE config 2 geometry, cube +-15000):
  * sparse      radii 0.5..50 (+0.1 % big)        - default camera: hierarchical skip (latency regime)
                                                   - far camera that sees everything: every tile TILE_ACCEPT (8 B/entity moved)
  * all_test    radii in (300, 330]: every cell is a "big" cell (culling_system.cpp:140,342) -> CELL_TEST, every sphere
                is fetched and tested: 20 B/entity moved, the regime SURVEY.md 8d's roofline formula describes
For each knob setting: HIP-event kernel time (lmx_profile_*), warm (back to back) and cold (1 GiB scrub before every cull).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--entities", type=int, default=10_000_000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep.json"))
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--settings", default="all", help="all | best: only variants 0 and 1 with the lane-parallel tile test")
    ap.add_argument("--variants", default="", help="comma-separated tile variants to run with the default tile test (overrides --settings), e.g. 1,4")
    ap.add_argument("--no-coldw", action="store_true", help="skip the dirty-scrub leg")
    ap.add_argument("--tag", default="", help="label copied into every record (e.g. the build variant behind LMX_LIB_PATH)")
    args = ap.parse_args()
    import torch

    from lumixengine_amd import api, scenes

    ctx = api.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    scrub = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")
    N = args.entities
    half = 15000.0 * (N / 1e7) ** (1.0 / 3.0)
    results = []

    scrub32 = scrub.view(torch.int32)

    def kernel_ms(cs, fr, reps, cold):
        """cold: 'w' = 1 GiB read-modify-write before every cull (leaves the Infinity Cache full of DIRTY lines),
        'r' = 1 GiB read-only reduction (clean lines)"""
        ctx.profile_reset()
        ctx.profile_enable(True)
        for _ in range(reps):
            if cold == "w":
                scrub.add_(1)
            elif cold == "r":
                scrub32.sum()
            cs.cull(fr)
        ctx.synchronize()
        ctx.profile_enable(False)
        ms, n = ctx.profile_get(api.K_CULL_SPHERES)
        return ms / max(n, 1)

    def wall_ms(cs, fr, reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            cs.cull(fr)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / reps

    cams = {
        "default": api.viewport_frustum(),
        "all_visible": api.viewport_frustum(pos=(0.0, 0.0, 4.0 * half), far=20.0 * half),
        "nothing": api.viewport_frustum(pos=(0.0, 0.0, -5.0 * half)),  # looks away from the scene: every tile ends at the tile-level test
    }

    for scene_name in ("sparse", "all_test"):
        t0 = time.time()
        sc = scenes.cull_scene(N, half, seed=2)
        if scene_name == "all_test":
            rng = np.random.default_rng(5)
            sc["radius"] = rng.uniform(300.5, 330.0, size=N).astype(np.float32)
        cs = api.CullingSystem(ctx)
        cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
        print(f"[{scene_name}] built {N} in {time.time() - t0:.1f}s", file=sys.stderr, flush=True)
        for _ in range(300 if N <= 20_000_000 else 20):
            cs.cull(cams["default"])
        ctx.synchronize()
        legs = [("default", cams["default"])] + ([("all_visible", cams["all_visible"]), ("nothing", cams["nothing"])] if scene_name == "sparse" else [])
        settings = []
        for variant in ((0, 1, 3, 4, 5) if args.settings == "best" else (0, 1, 2, 3, 4, 5)):
            for lanepar in ((1,) if args.settings == "best" else (1, 0)):
                settings.append(dict(variant=variant, lanepar=lanepar, shards=64, pad=32))
        if args.variants:
            settings = [dict(variant=int(v), lanepar=2, shards=64, pad=32) for v in args.variants.split(",")]
        if not args.quick and not args.variants:
            for shards, pad in ((1, 32), (8, 32), (16, 32), (64, 1), (64, 16)):
                settings.append(dict(variant=0, lanepar=1, shards=shards, pad=pad))
                settings.append(dict(variant=1, lanepar=1, shards=shards, pad=pad))
        for st in settings:
            cs.setOption(api.CULL_OPT_TILE_VARIANT, st["variant"])
            cs.setOption(api.CULL_OPT_LANE_PARALLEL_TILE_TEST, st["lanepar"])
            cs.setOption(api.CULL_OPT_MAX_SHARDS, st["shards"])
            cs.setOption(api.CULL_OPT_COUNTER_PAD, st["pad"])
            for leg, fr in legs:
                for _ in range(5):
                    res = cs.cull(fr)
                vis = int(res.counts()[0].sum())
                rec = dict(tag=args.tag, scene=scene_name, leg=leg, visible=vis, **st)
                rec["warm_kernel_us"] = 1e3 * kernel_ms(cs, fr, args.reps, cold=None)
                rec["cold_kernel_us"] = 1e3 * kernel_ms(cs, fr, max(10, args.reps // 2), cold="r")
                rec["coldw_kernel_us"] = float("nan") if args.no_coldw else 1e3 * kernel_ms(cs, fr, max(10, args.reps // 2), cold="w")
                rec["wall_us"] = 1e3 * wall_ms(cs, fr, args.reps * (5 if N <= 20_000_000 else 1))
                if scene_name == "all_test":
                    moved = 20.0 * N + 4.0 * vis
                elif leg == "all_visible":
                    moved = 8.0 * N
                else:
                    moved = 20.0 * N + 4.0 * vis  # algorithmic (not moved): throughput figure only
                rec["bytes"] = moved
                rec["warm_GBps"] = moved / rec["warm_kernel_us"] / 1e3
                rec["cold_GBps"] = moved / rec["cold_kernel_us"] / 1e3
                rec["coldw_GBps"] = moved / rec["coldw_kernel_us"] / 1e3
                results.append(rec)
                print(json.dumps(rec), file=sys.stderr, flush=True)
        del cs
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(results, f, indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
