#!/usr/bin/env python
"""Runs ONE hot-path workload in a loop so that rocprofv3 (kernel trace / PMC passes) sees only its kernels.

    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python tools/run_workload.py --workload skin --steps 20
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["cull_default", "cull_stream", "cull_all_test", "cull8_all_test", "cull_slab", "cull_dense", "cull8", "xform", "skin", "skin_distinct", "keys", "target"], required=True)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--entities", type=int, default=10_000_000)
    ap.add_argument("--instances", type=int, default=2000)
    ap.add_argument("--cold", choices=["none", "read", "write"], default="none",
                    help="evict the Infinity Cache before every cull: read = 1 GiB read-only reduction (clean lines), write = 1 GiB read-modify-write")
    args = ap.parse_args()
    import torch

    from lumixengine_amd import api, scenes
    from tests import helpers as H

    ctx = api.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.workload == "keys":  # cull + createSortKeys on the dense 10 M scene, default camera (1 M visible)
        sc = scenes.cull_scene(args.entities, 5000.0, seed=2)
        cs = api.CullingSystem(ctx)
        cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
        fr = api.viewport_frustum()
        ks = scenes.keys_scene(args.entities, sc["type"], seed=12, max_sort_key=255)
        sk = api.SortKeys(ctx)
        sk.setModels(ks["models"], ks["mesh_types"])
        sk.setInstances(ks["model"], ks["material_offset"], ks["mesh_materials"], ks["lod"], ks["flags"], ks["dirty"], ks["pose_frame"])
        sk.setPositions(sc["pos"])
        for f in range(args.steps):
            cs.cull(fr)
            sk.run(api.keys_view(layer_to_bucket=ks["layer_to_bucket"], bucket_depth_sorted=ks["bucket_depth_sorted"], frame_number=100 + f), 255)
        ctx.synchronize()
        print(sk.counts())
        # the span of the chain as bench.py reports it (also.keys_kernels_ms: first launch's begin -> last launch's end), both list forms
        kid = api.KERNEL_NAMES.index("sort_keys")
        for walk, ranks in ((1, 1), (1, 0), (0, 0), (1, 1)):
            sk.setOption(api.KEYS_OPT_WALK_SHARDS, walk)
            sk.setOption(api.KEYS_OPT_BLOCK_RANKS, ranks)
            spans = []
            for rep in range(3):
                for f in range(3):
                    cs.cull(fr)
                    sk.run(api.keys_view(layer_to_bucket=ks["layer_to_bucket"], bucket_depth_sorted=ks["bucket_depth_sorted"], frame_number=500 + 10 * rep + f), 255)
                ctx.profile_reset()
                ctx.profile_enable(True)
                for f in range(5):
                    cs.cull(fr)
                    sk.run(api.keys_view(layer_to_bucket=ks["layer_to_bucket"], bucket_depth_sorted=ks["bucket_depth_sorted"], frame_number=600 + 10 * rep + f), 255)
                ctx.synchronize()
                ctx.profile_enable(False)
                spans.append(ctx.profile_get(kid)[0] / 5 * 1e3)
            print(f"createSortKeys span, walk_shards {walk} block_ranks {ranks}: " + " ".join(f"{x:.1f}" for x in spans) + " us", sk.counts())
    elif args.workload.startswith("cull"):
        half = 5000.0 if args.workload == "cull_dense" else 15000.0 * (args.entities / 1e7) ** (1.0 / 3.0)
        sc = scenes.cull_scene(args.entities, half, seed=2, mixed_types=args.workload == "cull8")
        if args.workload in ("cull_all_test", "cull8_all_test"):  # every sphere "big" (radius > 300): every cell CELL_TEST, every sphere fetched and tested
            sc["radius"] = np.random.default_rng(5).uniform(300.5, 330.0, size=args.entities).astype(np.float32)
        if args.workload == "cull_slab":  # normal radii, one layer of cells, ortho slab camera: every cell CELL_TEST through the AABB pre-tests
            sc = scenes.slab_scene(args.entities, seed=2)
        cs = api.CullingSystem(ctx)
        cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
        if "LMX_TILE_VARIANT" in os.environ:
            cs.setOption(api.CULL_OPT_TILE_VARIANT, int(os.environ["LMX_TILE_VARIANT"]))
        if args.workload == "cull_stream" or os.environ.get("LMX_WORKLOAD_CAMERA") == "far":
            fr = api.viewport_frustum(pos=(0.0, 0.0, 60000.0), far=200000.0) if args.workload == "cull_dense" else api.viewport_frustum(pos=(0.0, 0.0, 4.0 * half), far=20.0 * half)
        elif os.environ.get("LMX_WORKLOAD_CAMERA") == "nothing":  # every tile rejected by the tile-level test: the kernel's floor
            fr = api.viewport_frustum(pos=(1.0e6, 50.0, -1.0e6))
        elif args.workload == "cull_slab":
            fr = api.viewport_frustum(**scenes.slab_frustum_kwargs(sc["half"]))
        elif args.workload == "cull8":
            fr = H.cascade_frusta(api, 8)
        elif args.workload == "cull8_all_test":  # bench.py's all_test_8_frusta_one_pass leg: config 5's 8 cascades in ONE pass over the spheres
            fr = np.concatenate([api.viewport_frustum(**kw) for kw in scenes.config5_cascade_kwargs()])
            cs.setPassWidth(int(os.environ.get("LMX_WORKLOAD_PASS_WIDTH", "8")))  # (4: two passes of the 2..4-frusta kernel shape)
        else:
            fr = api.viewport_frustum()
        import time
        scrub = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda") if args.cold != "none" else None
        for _ in range(5):
            cs.cull(fr)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if args.cold == "read":
                scrub.view(torch.int32).sum()
            elif args.cold == "write":
                scrub.add_(1)
            cs.cull(fr)
        ctx.synchronize()
        print(args.workload, "ms per cull %.4f" % ((time.perf_counter() - t0) * 1e3 / args.steps), "visible", cs.cull(fr).counts().sum(axis=1))
    elif args.workload == "xform":
        h = scenes.hierarchy_chains(250_000, 4, seed=2)
        w = api.World(ctx)
        w.build(h["parent"], h["local"])
        roots = np.flatnonzero(h["parent"] < 0).astype(np.int32)
        new_root = scenes.random_transforms(np.random.default_rng(1), len(roots), 4000.0)
        d_ent = torch.from_numpy(roots).cuda()
        d_tr = torch.from_numpy(new_root.view(np.uint8).reshape(len(roots), -1)).cuda()
        for _ in range(args.steps):
            w.setTransformsDevice(len(roots), d_ent.data_ptr(), d_tr.data_ptr())
            w.propagate()
        ctx.synchronize()
    elif args.workload == "target":  # north-star frame on one GPU: 10 M culled + 100 k skinned instances x 10 k verts (shared mesh)
        sc = scenes.cull_scene(args.entities, 15000.0, seed=2)
        cs = api.CullingSystem(ctx)
        cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
        fr = api.viewport_frustum()
        n_inst = 100_000
        s = scenes.skeleton(64, seed=4)
        sk = api.Skinning(ctx)
        model = sk.addModel(s["parents"], s["bind"], s["first_nonroot"])
        mesh = sk.addMesh(*scenes.skinned_mesh(10_000, 64, seed=6))
        sk.setInstances(np.full(n_inst, model, np.uint32), np.full(n_inst, mesh, np.uint32))
        pos, rot = scenes.relative_poses(n_inst, 64, seed=8)
        d_pos, d_rot = torch.from_numpy(pos).cuda(), torch.from_numpy(rot).cuda()
        sk.setPoseSourceDevice(d_pos.data_ptr(), d_rot.data_ptr(), n_inst * 64)
        for _ in range(2):
            cs.cull(fr)
            sk.run()
        ctx.synchronize()
        # the workload's OWN view of its kernels (the launches' begin / end timestamps through hipExtLaunchKernelGGL events, as bench.py reads
        # them) next to its wall time: run once plain and once under rocprofv3 to see what the profiler does to the launches it watches
        import time
        ctx.profile_reset(); ctx.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            cs.cull(fr)
            sk.run()
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / args.steps
        ctx.profile_enable(False)
        own = {api.KERNEL_NAMES[k]: round(ctx.profile_get(k)[0] / args.steps, 4) for k in range(len(api.KERNEL_NAMES)) if ctx.profile_get(k)[1]}
        print(f"target frame: wall {wall * 1e3:.3f} ms per frame; kernel ms per frame by the launches' own timestamps: {own}")
    else:
        n_inst, n_verts = args.instances, 10_000
        s = scenes.skeleton(64, seed=4)
        sk = api.Skinning(ctx)
        model = sk.addModel(s["parents"], s["bind"], s["first_nonroot"])
        if args.workload == "skin":
            verts, skin = scenes.skinned_mesh(n_verts, 64, seed=6)
            mesh = sk.addMesh(verts, skin)
            meshes = np.full(n_inst, mesh, np.uint32)
        else:  # every instance has its own mesh: 48 B of HBM traffic per vertex
            meshes = np.array([sk.addMesh(*scenes.skinned_mesh(n_verts, 64, seed=100 + i)) for i in range(n_inst)], np.uint32)
        sk.setInstances(np.full(n_inst, model, np.uint32), meshes)
        pos, rot = scenes.relative_poses(n_inst, 64, seed=5)
        d_pos, d_rot = torch.from_numpy(pos).cuda(), torch.from_numpy(rot).cuda()
        for _ in range(args.steps):
            sk.uploadPosesDevice(d_pos.data_ptr(), d_rot.data_ptr(), n_inst * 64)
            sk.run()
        ctx.synchronize()
    print("done", args.workload)


if __name__ == "__main__":
    main()
