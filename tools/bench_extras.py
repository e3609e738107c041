"""Side measurements of bench.py (NOT the headline `value`): the dense config-2 variant, createSortKeys, update / add streams, config 5's
single-GPU size, the 8-frusta pass, config 3's transform + skinning slices, the north-star target frame. bench.py writes what this
returns to bench_extra.json (and stderr); none of it is on the one stdout line the driver parses.

`dev` is bench.py's device shim (TorchDev on a GPU, HostsimDev in the CPU self-test): upload(ndarray) -> buffer with .ptr."""
import json
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBPS = 8000.0


def ids_sha256(ids):
    import hashlib

    return hashlib.sha256(np.sort(np.asarray(ids, np.int32)).tobytes()).hexdigest()

def extras(ctx, api, scenes, dev, timed, N, log, check_ids, big_entities=0, out=None, small=False):
    """Side measurements (not the headline `value`): dense config-2 variant, 8-frusta pass, config-3 transform + skin.
    `out`: the caller's dict, filled leg by leg (what was measured before a failing leg survives it). `small`: the CPU self-test's
    sizes (bench.py --selftest-hostsim: the code path, not a measurement)."""
    out = {} if out is None else out
    # sizes: the real ones, or (small) the CPU self-test's - same code path, nothing worth reading in the numbers
    Z = dict(upd=1000, add_frames=2000 if N >= 10_000_000 else 200, async_adds=300_000 if N >= 10_000_000 else 30_000, chains=250_000, skin_inst=2000, verts=10_000,
             c3_inst=10_000, target_inst=100_000, distinct=10_000, c5_skinned=200_000, reps=1.0)
    if small:
        Z = dict(upd=8, add_frames=6, async_adds=200, chains=200, skin_inst=3, verts=700, c3_inst=3, target_inst=5, distinct=2, c5_skinned=3, reps=0.0)
    R = lambda n: max(2, int(n * Z["reps"]))  # noqa: E731 - loop counts (2 in the self-test)
    # dense variant of config 2 (cube +-5000: ~37 k cells, ~270 spheres per cell)
    sc = scenes.cull_scene(N, 5000.0, seed=2)
    cs = api.CullingSystem(ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    fr = api.viewport_frustum()
    for _ in range(10):
        cs.cull(fr)
    ms = timed(lambda: cs.cull(fr), R(100))
    res_d = cs.cull(fr)
    vis = int(res_d.counts()[0].sum())
    if N == 10_000_000:  # tests/golden/cull_10m.json: the reference's digest of this very scene and camera (one renderable type)
        try:
            want_d = json.load(open(os.path.join(ROOT, "tests", "golden", "cull_10m.json")))["scenes"]["dense"]["cameras"]["default"]["sha256"]
            assert ids_sha256(res_d.all_ids(0)[0]) == want_d, "dense scene: visible ids differ from the reference's"
            out["dense_visible_ids"] = "reference"
        except (OSError, KeyError):
            out["dense_visible_ids"] = "unchecked"
    out["dense_entities_culled_per_sec"] = N / (ms * 1e-3)
    out["dense_ms_per_cull"] = ms
    out["dense_visible"] = vis
    # worst case for the hierarchical skip: a frustum that contains no whole cell but touches all of them is not
    # constructible; the closest is the camera far outside looking at the whole cube (every cell intersects or is inside)
    big = api.viewport_frustum(pos=(0.0, 0.0, 60000.0), far=200000.0)
    for _ in range(5):
        cs.cull(big)
    ms_all = timed(lambda: cs.cull(big), R(50))
    vis_all = int(cs.cull(big).counts()[0].sum())
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(10):
        cs.cull(big)
    ctx.synchronize()
    ctx.profile_enable(False)
    out["dense_all_visible_kernel_ms"] = ctx.profile_get(api.K_CULL_SPHERES)[0] / 10
    out["dense_stats"] = cs.stats()
    out["dense_all_visible_ms_per_cull"] = ms_all
    out["dense_all_visible_count"] = vis_all
    out["dense_all_visible_GBps"] = (20.0 * N + 4.0 * vis_all) / (ms_all * 1e-3) / 1e9

    # createSortKeys straight from the device-resident visible list: LOD selection + sort keys +
    # auto-instancer groups for every visible entity of the dense scene; the list never leaves HBM
    # two material populations: 256 distinct mesh sort keys (a scene built from a few hundred mesh / material pairs: the
    # per-wave aggregation of the instancer atomics works) and 4096 uniformly random ones (its worst case: almost every lane
    # of a wave holds a different key and the 16 KB of group counters take ~1.5 M atomics per million visible entities)
    sk = api.SortKeys(ctx)
    frame_no = [100]
    cases = []
    for max_key in (255, 4095):
        ks = scenes.keys_scene(N, sc["type"], seed=12, max_sort_key=max_key)
        tag = "keys" if max_key == 255 else "keys_4096_random_sort_keys"
        cases.append((tag, fr, vis, ks, max_key))
        if max_key == 255:
            cases.append(("keys_all_visible", big, vis_all, ks, max_key))
    current = [None]
    for name, frustum, visible, ks, max_key in cases:
        if current[0] is not ks:
            sk.setModels(ks["models"], ks["mesh_types"])
            sk.setInstances(ks["model"], ks["material_offset"], ks["mesh_materials"], ks["lod"], ks["flags"], ks["dirty"], ks["pose_frame"])
            sk.setPositions(sc["pos"])
            current[0] = ks
        def cull_keys():
            frame_no[0] += 1
            cs.cull(frustum)
            sk.run(api.keys_view(layer_to_bucket=ks["layer_to_bucket"], bucket_depth_sorted=ks["bucket_depth_sorted"], frame_number=frame_no[0]), max_key)
        for _ in range(3):
            cull_keys()
        ms_k = timed(cull_keys, R(20))
        ctx.profile_reset()
        ctx.profile_enable(True)
        for _ in range(5):
            cull_keys()
        ctx.synchronize()
        ctx.profile_enable(False)
        kid = api.KERNEL_NAMES.index("sort_keys")
        k_ms = ctx.profile_get(kid)[0] / 5
        cnt = sk.counts()
        out[name + "_cull_plus_keys_ms"] = ms_k
        out[name + "_kernels_ms"] = k_ms
        out[name + "_visible_per_sec"] = visible / (k_ms * 1e-3) if k_ms else None
        out[name + "_counts"] = cnt
        if name == "keys":  # the same leg through the entity-indexed tables only (LMX_KEYS_OPT_SLOT_ORDER 0: rounds 1 / 2's path)
            sk.setOption(api.KEYS_OPT_SLOT_ORDER, 0)
            for _ in range(3):
                cull_keys()
            ctx.profile_reset()
            ctx.profile_enable(True)
            for _ in range(5):
                cull_keys()
            ctx.synchronize()
            ctx.profile_enable(False)
            out["keys_kernels_ms_entity_indexed_tables"] = ctx.profile_get(kid)[0] / 5
            sk.setOption(api.KEYS_OPT_SLOT_ORDER, 1)
    sk.setOption(api.KEYS_OPT_SLOT_ORDER, 0)  # the legs below have no key tables: their culls should not emit slots
    del cs, sk, ks, cases

    # incremental updates on the headline scene: 1000 removals + 1000 adds per frame are O(1) patches (tombstones + overflow set), no
    # rebuild of the sorted layout. Cost per frame = (updates + cull) - cull, wall clock, host work included.
    sc_u = scenes.cull_scene(N, 15000.0, seed=2)
    cs_u = api.CullingSystem(ctx)
    cs_u.build(sc_u["entity"], sc_u["type"], sc_u["pos"], sc_u["radius"])
    fr_u = api.viewport_frustum()
    for _ in range(20):
        cs_u.cull(fr_u)
    ms_plain = timed(lambda: cs_u.cull(fr_u), R(200))
    rng_u = np.random.default_rng(3)
    U = Z["upd"]
    victims = rng_u.permutation(N)[: 120 * U].astype(np.int32).reshape(120, U)
    add_pos = rng_u.uniform(-15000.0, 15000.0, size=(120, U, 3))
    add_r = np.exp(rng_u.uniform(np.log(0.5), np.log(50.0), size=(120, U))).astype(np.float32)
    add_t = np.zeros(U, np.uint8)
    frame_u = [0]

    def update_frame():
        k = frame_u[0]
        frame_u[0] += 1
        cs_u.removeMany(victims[k])
        cs_u.addMany(np.arange(N + U * k, N + U * (k + 1), dtype=np.int32), add_t, add_pos[k], add_r[k])
        cs_u.cull(fr_u)

    for _ in range(10):
        update_frame()
    ms_upd = timed(update_frame, R(100))
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(5):
        update_frame()
    ctx.synchronize()
    ctx.profile_enable(False)
    t_patch, n_patch = ctx.profile_get(api.K_CULL_PATCH)
    t_dyn, n_dyn = ctx.profile_get(api.K_CULL_DYNAMIC)
    out["update_stream_device_us_per_frame"] = {"patch_copy_plus_kernel": 1e3 * t_patch / max(n_patch, 1), "overflow_set_cull_kernel": 1e3 * t_dyn / max(n_dyn, 1)}
    out["update_stream_plain_cull_ms"] = ms_plain
    out["update_stream_1000_add_1000_remove_plus_cull_ms"] = ms_upd
    out["update_stream_added_us_per_frame"] = (ms_upd - ms_plain) * 1e3
    out["update_stream_state"] = cs_u.updateStats()
    del cs_u, sc_u

    # the reference's add never stalls (culling_system.cpp:131-190): 2 M adds into the 10 M scene, 1000 per frame, every frame culled, with
    # the overflow reserve sized for the stream and no automatic compaction: slowest / median frame, and what the 2 M unsorted
    # overflow entities cost per cull at the end
    sc_s = scenes.cull_scene(N, 15000.0, seed=2)
    cs_s = api.CullingSystem(ctx)
    n_add_frames, per_frame = Z["add_frames"], Z["upd"]
    cs_s.setOption(api.CULL_OPT_AUTO_COMPACTION, 0)
    async_adds = Z["async_adds"]  # the second leg below: adds that arrive while the worker re-sorts
    cs_s.setOption(api.CULL_OPT_OVERFLOW_RESERVE, n_add_frames * per_frame + async_adds + 65536)
    cs_s.build(sc_s["entity"], sc_s["type"], sc_s["pos"], sc_s["radius"])
    fr_s = api.viewport_frustum()
    for _ in range(20):
        cs_s.cull(fr_s)
    ctx.synchronize()
    rng_s = np.random.default_rng(12)
    add_pos = rng_s.uniform(-15000.0, 15000.0, size=(n_add_frames * per_frame, 3))
    add_rad = np.exp(rng_s.uniform(np.log(0.5), np.log(50.0), size=n_add_frames * per_frame)).astype(np.float32)
    add_typ = np.zeros(per_frame, np.uint8)
    t_add = []
    for f in range(n_add_frames):
        a0, a1 = f * per_frame, (f + 1) * per_frame
        ids_f = np.arange(N + a0, N + a1, dtype=np.int32)
        t0 = time.perf_counter()
        cs_s.addMany(ids_f, add_typ, add_pos[a0:a1], add_rad[a0:a1])
        cs_s.cull(fr_s)
        ctx.synchronize()
        t_add.append(time.perf_counter() - t0)
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(10):
        cs_s.cull(fr_s)
    ctx.synchronize()
    ctx.profile_enable(False)
    t_dyn_s, n_dyn_s = ctx.profile_get(api.K_CULL_DYNAMIC)
    ta = np.array(t_add[5:])
    out["add_stream"] = {"adds": n_add_frames * per_frame, "per_frame": per_frame, "frames": n_add_frames, "max_frame_ms": float(ta.max()) * 1e3,
                         "p99_frame_ms": float(np.percentile(ta, 99)) * 1e3, "median_frame_ms": float(np.median(ta)) * 1e3,
                         "overflow_cull_kernel_ms_at_end": t_dyn_s / max(n_dyn_s, 1), "state": cs_s.updateStats(),
                         "note": "frame = addMany(1000) + cull + host wait; LMX_CULL_OPT_AUTO_COMPACTION 0, LMX_CULL_OPT_OVERFLOW_RESERVE = the stream's size: adds take free overflow slots, nothing is re-sorted or re-uploaded"}
    # ... and the re-sort itself off the frame (LMX_CULL_OPT_ASYNC_COMPACTION): the set now holds N sorted + 2 M unsorted entities, well
    # past the compaction threshold (N / 8). With the option on, the next flush asks the worker for a job: it folds the 2 M into the
    # sorted set and re-sorts all of it on a second copy of the sets while the frames go on - 100 adds + a cull each, paced at 1 kHz
    # (a frame of a real engine lasts milliseconds; the worker's catch-up has to outrun the update stream) - until the sets trade places
    t0 = time.perf_counter()
    cs_s.setOption(api.CULL_OPT_ASYNC_COMPACTION, 1)  # copies the host mirror once (O(n))
    t_enable = time.perf_counter() - t0
    cs_s.setOption(api.CULL_OPT_AUTO_COMPACTION, 1)
    next_id = N + n_add_frames * per_frame
    per_async, t_async, t_request, t_swap, frames_after_swap = 100, [], None, None, 0
    pos_a = rng_s.uniform(-15000.0, 15000.0, size=(async_adds, 3))
    rad_a = np.exp(rng_s.uniform(np.log(0.5), np.log(50.0), size=async_adds)).astype(np.float32)
    typ_a = np.zeros(per_async, np.uint8)
    t_start = time.perf_counter()
    f = 0
    while (f + 1) * per_async <= async_adds and time.perf_counter() - t_start < 12.0:
        a0, a1 = f * per_async, (f + 1) * per_async
        ids_f = np.arange(next_id + a0, next_id + a1, dtype=np.int32)
        t0 = time.perf_counter()
        cs_s.addMany(ids_f, typ_a, pos_a[a0:a1], rad_a[a0:a1])
        cs_s.cull(fr_s)
        ctx.synchronize()
        t1 = time.perf_counter()
        t_async.append(t1 - t0)
        st_a = cs_s.asyncStats()
        if t_request is None and st_a["state"] in (1, 2, 3):
            t_request = t1
        if t_swap is None and st_a["swaps"] >= 1:
            t_swap, frames_after_swap = t1, 0
        f += 1
        if t_swap is not None:
            frames_after_swap += 1
            if frames_after_swap > 100:  # a hundred frames on the re-sorted set, then done
                break
        pause = 1e-3 - (time.perf_counter() - t0)
        if pause > 0:
            time.sleep(pause)
    st_a = cs_s.asyncStats()
    # the layout the worker built against the one the synchronous path builds from the same mirror: same visible ids
    sha_async = ids_sha256(cs_s.cull(fr_s).all_ids(0)[0])
    cs_s.setOption(api.CULL_OPT_ASYNC_COMPACTION, 0)
    ids_x = np.arange(next_id + async_adds, next_id + async_adds + 10, dtype=np.int32)  # (something to fold, so that lmx_cull_compact re-sorts)
    cs_s.addMany(ids_x, np.zeros(10, np.uint8), np.full((10, 3), 1.0e7), np.ones(10, np.float32))  # far outside every frustum
    t0 = time.perf_counter()
    cs_s.compact()
    ctx.synchronize()
    t_sync_compact = time.perf_counter() - t0
    sha_sync = ids_sha256(cs_s.cull(fr_s).all_ids(0)[0])
    if sha_async != sha_sync:
        raise SystemExit("bench: visible ids after the asynchronous compaction differ from those after a synchronous one")
    tb = np.array(t_async[2:]) if len(t_async) > 2 else np.array([float("nan")])
    out["add_stream_async_compaction"] = {
        "frames": len(t_async), "adds_per_frame": per_async, "swaps": st_a["swaps"], "ops_replayed_at_swaps": st_a["ops_replayed_at_swaps"],
        "request_to_swap_s": None if (t_request is None or t_swap is None) else t_swap - t_request, "enable_copy_s": t_enable,
        "max_frame_ms": float(tb.max()) * 1e3, "p99_frame_ms": float(np.percentile(tb, 99)) * 1e3, "median_frame_ms": float(np.median(tb)) * 1e3,
        "state_after": cs_s.updateStats(), "synchronous_compaction_of_the_same_set_s": t_sync_compact,
        "visible_ids": "equal to those of the synchronously re-sorted set (sha256)",
        "note": "frame = addMany(100) + cull + host wait while a worker thread folds 2 M overflow entities into the sorted set and re-sorts all 12 M of it on a second copy of the sets; the swap (an O(1) trade + a replay of the last frames' operations) happens inside one of these frames"}
    cs_s.setOption(api.CULL_OPT_ASYNC_COMPACTION, 0)
    cs_s.setOption(api.CULL_OPT_OVERFLOW_RESERVE, 0)
    cs_s.setOption(api.CULL_OPT_AUTO_COMPACTION, 1)
    del cs_s, sc_s, add_pos, add_rad, pos_a, rad_a

    # BASELINE config 5's single-GPU size: 100 M entities (2 GB of spheres + ids, far beyond the 256 MiB Infinity Cache: every pass
    # is HBM-cold by construction, no scrub needed). Same three regimes as the roofline legs + the 8 cascades in one call.
    if big_entities:
        NB = big_entities
        half_b = 15000.0 * (NB / 1e7) ** (1.0 / 3.0)
        t0 = time.time()
        sc_b = scenes.cull_scene(NB, half_b, seed=2)  # the 10 M legs' scene at ten times the size
        cs_b = api.CullingSystem(ctx)
        cs_b.build(sc_b["entity"], sc_b["type"], sc_b["pos"], sc_b["radius"])
        big = {"entities": NB, "half_extent": half_b, "scene_plus_build_s": round(time.time() - t0, 1), "cells": cs_b.stats()["cells"]}

        def leg(csys, fr, reps=10):
            for _ in range(3):
                csys.cull(fr)
            ms_wall = timed(lambda: csys.cull(fr), reps)
            ctx.profile_reset()
            ctx.profile_enable(True)
            for _ in range(reps):
                csys.cull(fr)
            ctx.synchronize()
            ctx.profile_enable(False)
            ms_k, n_k = ctx.profile_get(api.K_CULL_SPHERES)
            return ms_wall, ms_k / max(n_k, 1), csys.cull(fr).counts().sum(axis=1)

        fr_d = api.viewport_frustum()
        w, k, v = leg(cs_b, fr_d)
        at_size = NB == 100_000_000  # digests exist for this size (tests/golden/cull_bench_scenes.json: config5_100m / all_test_100m)
        big["default_camera"] = {"ms_per_cull": w, "kernel_ms": k, "visible": int(v[0]), "entities_per_sec": NB / (w * 1e-3),
                                 "visible_ids": check_ids(cs_b.cull(fr_d), "config5_100m") if at_size else "unchecked"}
        w, k, v = leg(cs_b, api.viewport_frustum(pos=(0.0, 0.0, 4.0 * half_b), far=20.0 * half_b))
        big["all_accept"] = {"ms_per_cull": w, "kernel_ms": k, "visible": int(v[0]), "moved_bytes": 8.0 * NB, "GBps": 8.0 * NB / (k * 1e-3) / 1e9,
                             "frac_of_8TBps": 8.0 * NB / (k * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        fr8b = np.concatenate([api.viewport_frustum(**kw) for kw in scenes.config5_cascade_kwargs()])
        w8, _, v8 = leg(cs_b, fr8b, reps=5)
        big["cascades_8_frusta"] = {"ms_per_call": w8, "visible_per_frustum": [int(x) for x in v8], "entity_frustum_tests_per_sec": 8.0 * NB / (w8 * 1e-3)}
        if at_size:
            res8 = cs_b.cull(fr8b)
            big["cascades_8_frusta"]["visible_ids"] = [check_ids(res8, "config5_100m", f"cascade{k}", frustum=k) for k in range(8)]
        # BASELINE config 5 as ONE single-GPU frame (round 6): the 100 M entities under the frame's 8 cascade frusta + the skinned share. The config's
        # "mixed static/skinned" scene has 1 % of its entities skinned - 1 M instances x 10 k vertices = 120 GB of positions per frame, which one GPU holds
        # but no frame budget does; the share is SCALED to Z["c5_skinned"] instances (200 k = 0.2 %: 2 x 10^9 vertices, 24 GB of positions per frame; the 8-GPU
        # configuration gives each GPU 125 k), shared 10 k-vertex mesh, 64 bones, poses resident in HBM.
        n_c5 = Z["c5_skinned"]
        s5 = scenes.skeleton(64, seed=4)
        verts5, skin5 = scenes.skinned_mesh(10_000 if not small else 700, 64, seed=6)
        sk5 = api.Skinning(ctx)
        model5 = sk5.addModel(s5["parents"], s5["bind"], s5["first_nonroot"])
        mesh5 = sk5.addMesh(verts5, skin5)
        sk5.setInstances(np.full(n_c5, model5, np.uint32), np.full(n_c5, mesh5, np.uint32))
        pos5, rot5 = scenes.relative_poses(n_c5, 64, seed=11)
        d_pos5, d_rot5 = dev.upload(pos5), dev.upload(rot5)
        del pos5, rot5
        sk5.setPoseSourceDevice(d_pos5.ptr, d_rot5.ptr, n_c5 * 64)

        def frame5():
            cs_b.cull(fr8b)
            sk5.run()

        for _ in range(3):
            frame5()
        ms5 = timed(frame5, R(12))
        big["frame_8_cascades_plus_skinned_share"] = {
            "ms_per_frame": ms5, "frames_per_sec": 1e3 / ms5, "entities": NB, "frusta": 8, "skinned_instances": n_c5, "verts_per_instance": len(verts5),
            "skinned_share_of_entities": n_c5 / NB, "entity_frustum_tests_per_sec": 8.0 * NB / (ms5 * 1e-3), "skinned_verts_per_sec": n_c5 * len(verts5) / (ms5 * 1e-3),
            "note": "config 5's 1 % skinned share would be 1 M instances (120 GB of positions per frame): scaled to what a frame can carry; cull of all 8 cascades + pose -> palette -> vertices per frame"}
        out["config5_frame_1gpu_ms"] = ms5
        out["config5_frame_1gpu_is"] = f"{NB} entities x 8 cascade frusta + {n_c5} skinned instances x {len(verts5)} vertices (the config's 1 % skinned share scaled to {100.0 * n_c5 / NB:.2f} %), one GPU"
        del sk5, d_pos5, d_rot5
        del cs_b
        sc_b["radius"] = scenes.all_test_radii(NB)
        cs_b = api.CullingSystem(ctx)
        cs_b.build(sc_b["entity"], sc_b["type"], sc_b["pos"], sc_b["radius"])
        w, k, v = leg(cs_b, fr_d)
        moved = 20.0 * NB + 4.0 * float(v[0])
        big["all_test"] = {"ms_per_cull": w, "kernel_ms": k, "visible": int(v[0]), "moved_bytes": moved, "GBps": moved / (k * 1e-3) / 1e9,
                           "frac_of_8TBps": moved / (k * 1e-3) / 1e9 / HBM_PEAK_GBPS, "visible_ids": check_ids(cs_b.cull(fr_d), "all_test_100m") if at_size else "unchecked"}
        out["config5_size_single_gpu"] = big
        del cs_b, sc_b

    # config 5 flavour on one GPU: mixed renderable types, 8 ortho cascade frusta tested in ONE pass over the spheres
    sc = scenes.cull_scene(N, 15000.0, seed=4, mixed_types=True)
    cs = api.CullingSystem(ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    fr8 = np.concatenate([api.viewport_frustum(**kw) for kw in scenes.config5_cascade_kwargs()])
    for _ in range(5):
        cs.cull(fr8)
    ms8 = timed(lambda: cs.cull(fr8), R(50))
    c8 = cs.cull(fr8).counts()
    out["cull8_ms_per_pass"] = ms8
    out["cull8_entity_frustum_tests_per_sec"] = 8.0 * N / (ms8 * 1e-3)
    out["cull8_visible_per_frustum"] = [int(x) for x in c8.sum(axis=1)]
    out["cull8_GBps_algorithmic"] = (20.0 * N + 4.0 * float(c8.sum())) / (ms8 * 1e-3) / 1e9
    del cs

    # config 3 slice: 1 M entities, depth-4 chains, every root moved each frame (transform inputs resident in HBM)
    h = scenes.hierarchy_chains(Z["chains"], 4, seed=2)
    n = len(h["parent"])
    w = api.World(ctx)
    w.build(h["parent"], h["local"])
    roots = np.flatnonzero(h["parent"] < 0).astype(np.int32)
    new_root = scenes.random_transforms(np.random.default_rng(1), len(roots), 4000.0)
    d_ent = dev.upload(roots)
    d_tr = dev.upload(new_root)

    def xform_step():
        w.setTransformsDevice(len(roots), d_ent.ptr, d_tr.ptr)
        w.propagate()

    for _ in range(10):
        xform_step()
    ms = timed(xform_step, R(100))
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(20):
        xform_step()
    ctx.synchronize()
    ctx.profile_enable(False)
    t_lvl, n_lvl = ctx.profile_get(api.K_XFORM_LEVEL)
    out["xform_level_kernel_avg_ms"] = t_lvl / max(n_lvl, 1)
    out["xform_level_launches_per_frame"] = n_lvl / 20
    n_child = n - len(roots)
    out["transforms_per_sec"] = n_child / (ms * 1e-3)
    out["transform_ms_per_frame"] = ms
    out["transform_GBps_algorithmic"] = 156.0 * n_child / (ms * 1e-3) / 1e9
    # the same propagation with the moved list on (lmx_world_track_moved): what the engine-side module pays every frame, because it replays
    # the `transformed` delegates of the entities a propagation moved (world.cpp:255-282)
    w.trackMoved(True)
    for _ in range(5):
        xform_step()
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(20):
        xform_step()
    ctx.synchronize()
    ctx.profile_enable(False)
    t_mv, n_mv = ctx.profile_get(api.K_XFORM_LEVEL)
    out["xform_with_moved_list_ms"] = t_mv / max(n_mv, 1)
    w.trackMoved(False)
    del w

    # config 3 slice: skinned instances x 64 bones x 10 k verts, shared mesh (2 k instances = 20 M verts per frame)
    n_inst, n_verts = Z["skin_inst"], Z["verts"]
    s = scenes.skeleton(64, seed=4)
    verts, skin = scenes.skinned_mesh(n_verts, 64, seed=6)
    sk = api.Skinning(ctx)
    model = sk.addModel(s["parents"], s["bind"], s["first_nonroot"])
    mesh = sk.addMesh(verts, skin)
    sk.setInstances(np.full(n_inst, model, np.uint32), np.full(n_inst, mesh, np.uint32))
    pos, rot = scenes.relative_poses(n_inst, 64, seed=5)
    d_pos = dev.upload(pos)
    d_rot = dev.upload(rot)

    def skin_step():
        sk.uploadPosesDevice(d_pos.ptr, d_rot.ptr, n_inst * 64)
        sk.run()

    for _ in range(5):
        skin_step()
    ms = timed(skin_step, R(50))
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(20):
        skin_step()
    ctx.synchronize()
    ctx.profile_enable(False)
    t_pp, n_pp = ctx.profile_get(api.K_POSE_PALETTE)
    t_sv, n_sv = ctx.profile_get(api.K_SKIN_VERTICES)
    out["pose_palette_kernel_avg_ms"] = t_pp / max(n_pp, 1)
    out["skin_vertices_kernel_avg_ms"] = t_sv / max(n_sv, 1)
    out["skin_vertices_kernel_verts_per_sec"] = n_inst * n_verts / (t_sv / max(n_sv, 1) * 1e-3)
    out["skinned_verts_per_sec"] = n_inst * n_verts / (ms * 1e-3)
    out["skin_ms_per_frame"] = ms
    out["skin_instances"] = n_inst
    out["skin_GBps_algorithmic_48B"] = 48.0 * n_inst * n_verts / (ms * 1e-3) / 1e9
    out["skin_GBps_shared_mesh_floor_12B"] = 12.0 * n_inst * n_verts / (ms * 1e-3) / 1e9
    # the same slice through the dual-quaternion path (32-byte palette + the shader's DQ vertex blend) and in
    # LMX_SKIN_EXACT (the mode that is bit-identical to evaluateSkin)
    for mode_name, mode in (("dqs", api.SKIN_DQS), ("exact", api.SKIN_EXACT)):
        sk.setMode(mode)
        for _ in range(3):
            skin_step()
        ctx.profile_reset()
        ctx.profile_enable(True)
        for _ in range(10):
            skin_step()
        ctx.synchronize()
        ctx.profile_enable(False)
        t_m, n_m = ctx.profile_get(api.K_SKIN_VERTICES)
        out[f"skin_{mode_name}_vertex_kernel_avg_ms"] = t_m / max(n_m, 1)
        out[f"skin_{mode_name}_verts_per_sec"] = n_inst * n_verts / (t_m / max(n_m, 1) * 1e-3)
    sk.setMode(api.SKIN_FUSED)
    del sk
    # BASELINE config 3 as one simulated frame on one GPU: 1 M entities in depth-4 chains, every root moved, every entity
    # bound to the culling system (dynamic set, refreshed on the device), one camera cull, 10 k skinned instances x 64 bones
    # x 10 k vertices of one shared mesh. Inputs (new root transforms, relative poses) are resident in HBM.
    h3 = scenes.hierarchy_chains(Z["chains"], 4, seed=2, root_extent=6000.0)
    n3 = len(h3["parent"])
    w3 = api.World(ctx)
    w3.build(h3["parent"], h3["local"])
    cs3 = api.CullingSystem(ctx)
    ent3 = np.arange(n3, dtype=np.int32)
    rng3 = np.random.default_rng(3)
    cs3.build(ent3, np.zeros(n3, np.uint8), rng3.uniform(-6000.0, 6000.0, size=(n3, 3)), np.ones(n3, np.float32))
    w3.bindCulling(ent3, rng3.uniform(0.5, 20.0, n3).astype(np.float32))
    roots3 = np.flatnonzero(h3["parent"] < 0).astype(np.int32)
    d_ent3 = dev.upload(roots3)
    d_tr3 = dev.upload(scenes.random_transforms(rng3, len(roots3), 6000.0))
    n_inst3 = Z["c3_inst"]
    sk3 = api.Skinning(ctx)
    model3 = sk3.addModel(s["parents"], s["bind"], s["first_nonroot"])
    mesh3 = sk3.addMesh(verts, skin)
    sk3.setInstances(np.full(n_inst3, model3, np.uint32), np.full(n_inst3, mesh3, np.uint32))
    pos3, rot3 = scenes.relative_poses(n_inst3, 64, seed=7)
    d_pos3, d_rot3 = dev.upload(pos3), dev.upload(rot3)
    fr3 = api.viewport_frustum()

    sk3.setPoseSourceDevice(d_pos3.ptr, d_rot3.ptr, n_inst3 * 64)

    def frame3():
        w3.setTransformsDevice(len(roots3), d_ent3.ptr, d_tr3.ptr)
        w3.propagate()
        cs3.cull(fr3)
        sk3.run()

    for _ in range(5):
        frame3()
    ms3 = timed(frame3, R(50))
    out["config3_frame_ms"] = ms3
    out["config3_frames_per_sec"] = 1e3 / ms3
    out["config3_visible"] = int(cs3.cull(fr3).counts()[0].sum())
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(10):
        frame3()
    ctx.synchronize()
    ctx.profile_enable(False)
    out["config3_kernel_ms"] = {api.KERNEL_NAMES[k]: round(ctx.profile_get(k)[0] / 10, 5) for k in range(len(api.KERNEL_NAMES)) if ctx.profile_get(k)[1]}
    del w3, cs3, sk3
    # North-star target on ONE GPU: 10 M entities culled + 100 k skinned instances (64 bones, 10 k verts of one shared mesh)
    # per simulated frame; >= 240 frames/s asked. 1e9 vertices = 12 GB of skinned positions written per frame.
    cs4 = api.CullingSystem(ctx)
    sc4 = scenes.cull_scene(N, 15000.0, seed=2)
    cs4.build(sc4["entity"], sc4["type"], sc4["pos"], sc4["radius"])
    del sc4
    n_inst4 = Z["target_inst"]
    sk4 = api.Skinning(ctx)
    model4 = sk4.addModel(s["parents"], s["bind"], s["first_nonroot"])
    mesh4 = sk4.addMesh(verts, skin)
    sk4.setInstances(np.full(n_inst4, model4, np.uint32), np.full(n_inst4, mesh4, np.uint32))
    pos4, rot4 = scenes.relative_poses(n_inst4, 64, seed=8)
    d_pos4, d_rot4 = dev.upload(pos4), dev.upload(rot4)
    del pos4, rot4
    fr4 = api.viewport_frustum()

    sk4.setPoseSourceDevice(d_pos4.ptr, d_rot4.ptr, n_inst4 * 64)  # poses are read where the animation system left them

    def frame4():
        cs4.cull(fr4)
        sk4.run()

    for _ in range(4):
        frame4()
    ms4 = timed(frame4, R(30))  # (30 frames behind 4 untimed ones: over 10 behind 2 the first launches after the set-up - clocks ramping, first touches - made the frame 6 % longer than its kernels)
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(5):
        frame4()
    ctx.synchronize()
    ctx.profile_enable(False)
    out["target_kernel_ms"] = {api.KERNEL_NAMES[k]: round(ctx.profile_get(k)[0] / 5, 5) for k in range(len(api.KERNEL_NAMES)) if ctx.profile_get(k)[1]}
    out["target_frame_10M_cull_100k_skinned_ms"] = ms4
    out["target_frames_per_sec_1gpu"] = 1e3 / ms4
    out["target_skinned_verts_per_sec"] = n_inst4 * n_verts / (ms4 * 1e-3)
    out["target_skin_ms_per_1e9_verts"] = out["target_kernel_ms"].get("skin_vertices", float("nan")) * 1e9 / (n_inst4 * n_verts)
    # the same frame with a mesh that has the skinning statistics of a real character (scenes.skinned_mesh_character: the reference's
    # demo character has 52 bones, 1.0-1.2 influences per control point, <= 27 bones per 5120-vertex tile) instead of the worst case
    # above (4 random bones of 64 per vertex): k_skin_shared stages only the palette rows of the bones a tile references
    verts_c, skin_c = scenes.skinned_mesh_character(n_verts, 52, seed=6)
    mesh4c = sk4.addMesh(verts_c, skin_c)
    sk4.setInstances(np.full(n_inst4, model4, np.uint32), np.full(n_inst4, mesh4c, np.uint32))
    sk4.setPoseSourceDevice(d_pos4.ptr, d_rot4.ptr, n_inst4 * 64)
    for _ in range(4):
        frame4()
    ms4c = timed(frame4, R(30))
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(5):
        frame4()
    ctx.synchronize()
    ctx.profile_enable(False)
    k4c = {api.KERNEL_NAMES[k]: round(ctx.profile_get(k)[0] / 5, 5) for k in range(len(api.KERNEL_NAMES)) if ctx.profile_get(k)[1]}
    out["target_character_mesh"] = {"frame_ms": ms4c, "frames_per_sec_1gpu": 1e3 / ms4c, "kernel_ms": k4c,
                                    "skin_ms_per_1e9_verts": k4c.get("skin_vertices", float("nan")) * 1e9 / (n_inst4 * n_verts),
                                    "mesh": "scenes.skinned_mesh_character(10 000 vertices, 52 bones of the 64-bone skeleton): 1.17 influences per vertex, 28 bones per tile"}
    sk4.setInstances(np.full(n_inst4, model4, np.uint32), np.full(n_inst4, mesh4, np.uint32))
    sk4.setPoseSourceDevice(d_pos4.ptr, d_rot4.ptr, n_inst4 * 64)
    # the same frame for a renderer that consumes only palettes / vertices (no absolute-pose store, lmx_skin_set_pose_writeback)
    sk4.setPoseWriteback(False)
    for _ in range(4):
        frame4()
    ms4b = timed(frame4, R(30))
    out["target_no_pose_store_frame_ms"] = ms4b
    out["target_no_pose_store_frames_per_sec_1gpu"] = 1e3 / ms4b
    # ... and with the relative poses sampled on the device every frame (updateAnimable for all 100 k instances, SURVEY.md 8f
    # rank 2) instead of read from a static buffer: animation -> absolute pose -> palette -> vertices never leaves HBM
    sk4.setPoseWriteback(True)
    sk4.setModelPose(model4, s["bind"])
    anim4 = [sk4.addAnimation(scenes.animation(64, 60, 30.0, seed=70 + k)) for k in range(4)]
    rng4 = np.random.default_rng(4)
    sk4.setAnimables(np.array(anim4, np.uint32)[rng4.integers(0, 4, size=n_inst4)], rng4.integers(0, 2 << 15, size=n_inst4).astype(np.uint32))

    def frame4a():
        cs4.cull(fr4)
        sk4.updateAnimables(1.0 / 240.0)
        sk4.run()

    for _ in range(4):
        frame4a()
    ms4a = timed(frame4a, R(30))
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(5):
        frame4a()
    ctx.synchronize()
    ctx.profile_enable(False)
    out["target_animated_frame_ms"] = ms4a
    out["target_animated_frames_per_sec_1gpu"] = 1e3 / ms4a
    out["target_animated_kernel_ms"] = {api.KERNEL_NAMES[k]: round(ctx.profile_get(k)[0] / 5, 5) for k in range(len(api.KERNEL_NAMES)) if ctx.profile_get(k)[1]}
    del cs4, sk4, d_pos4, d_rot4
    # BASELINE config 3's distinct-mesh variant AT ITS STATED SIZE (round 6): 10 000 instances, every one its own 10 k-vertex mesh (scenes.distinct_mesh) - 3.2 GB of vertex
    # records streamed from HBM per frame + 1.2 GB of positions written (44 B per vertex moved; SURVEY.md's algorithmic figure is 48)
    n_inst2 = Z["distinct"]
    sk = api.Skinning(ctx)
    model = sk.addModel(s["parents"], s["bind"], s["first_nonroot"])
    t0 = time.time()
    mesh_ids = [sk.addMesh(*scenes.distinct_mesh(verts, skin, i)) for i in range(n_inst2)]
    sk.setInstances(np.full(n_inst2, model, np.uint32), np.array(mesh_ids, np.uint32))
    pos_d, rot_d = scenes.relative_poses(scenes.DISTINCT_MESH_POSES if not small else n_inst2, 64, seed=5)
    d_pos2, d_rot2 = dev.upload(pos_d[:n_inst2]), dev.upload(rot_d[:n_inst2])

    def skin_step2():
        sk.uploadPosesDevice(d_pos2.ptr, d_rot2.ptr, n_inst2 * 64)
        sk.run()

    skin_step2()
    out["skin_distinct_meshes_setup_s"] = round(time.time() - t0, 1)
    # identity first: LMX_SKIN_EXACT positions of the sampled instances against the digests the reference's evaluateSkin produced for them
    # (tests/golden/skin_distinct.json, written by tests/golden/make_golden_skin_distinct.py; bench.py itself never touches the oracle)
    checked = "unchecked"
    try:
        import hashlib

        g = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "skin_distinct.json")))
        if not small and n_verts == g["n_verts"] and n_inst2 <= g["poses_drawn"]:
            sk.setMode(api.SKIN_EXACT)
            skin_step2()
            sample = [i for i in scenes.DISTINCT_MESH_SAMPLE if i < n_inst2]
            bad = [i for i in sample if hashlib.sha256(np.ascontiguousarray(sk.readVertices(i), np.float32).tobytes()).hexdigest() != g["instances"][str(i)]]
            if bad:
                raise SystemExit(f"bench: skinned positions of the distinct-mesh instances {bad} differ from the reference's")
            checked = f"reference (LMX_SKIN_EXACT sha256 of instances {sample})"
    finally:
        sk.setMode(api.SKIN_FUSED)
    out["skin_distinct_meshes_positions"] = checked
    for _ in range(3):
        skin_step2()
    ms_d = timed(skin_step2, 10 if not small else 1)
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(10 if not small else 1):
        skin_step2()
    ctx.synchronize()
    ctx.profile_enable(False)
    t_sv, n_sv = ctx.profile_get(api.K_SKIN_VERTICES)
    out["skin_distinct_meshes_instances"] = n_inst2
    out["skin_distinct_meshes_ms_per_frame"] = ms_d
    out["skin_distinct_meshes_kernel_avg_ms"] = t_sv / max(n_sv, 1)
    out["skin_distinct_meshes_verts_per_sec"] = n_inst2 * n_verts / (t_sv / max(n_sv, 1) * 1e-3)
    out["skin_distinct_meshes_GBps_48B"] = 48.0 * n_inst2 * n_verts / (t_sv / max(n_sv, 1) * 1e-3) / 1e9
    del sk, d_pos2, d_rot2
    return out
