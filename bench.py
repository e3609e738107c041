#!/usr/bin/env python
"""bench.py — BASELINE.json metric on MI355X: entities culled / s (+ skinned verts / s, transforms / s as extras).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload at N=1: BASELINE config 2 ("10M static entities, 1 frustum, 1xMI355X cull + compaction"), sparse variant
(cube [-15000,15000]^3, ~1 M occupied cells), camera = the reference player's default viewport (fov 60 deg, 1920x1080,
near 0.1, far 10000). One step = one cull of every resident entity INCLUDING the compaction into one packed,
contiguous record [8 per-type counts | ids] left in HBM (k_cull_tile + k_cull_pack: SURVEY.md 8d's "wall time of one cull incl.
compaction"). The JSON line also carries, next to `value` (an EFFECTIVE rate: the default camera rejects ~95 % of the tiles by their
boxes, so most resident bytes are never moved), `value_cull_only` (round 1 / 2's step: k_cull_tile alone, ids left in per-shard
windows), `value_incl_host_readback` (the visible ids in HOST memory: cull + lmx_cull_map_all, one host wait, PCIe included - the
second form of SURVEY.md 8d's metric; never `value`) and `value_streaming` (entities / s of the kernel when every sphere is fetched and
tested, cache-cold: the regime `roofline` describes). The ids the timed camera produces are checked against the reference's sha256 (tests/golden/). For N>1 (one process per GPU) a step is cull + the native exchange (lmx_exchange_*: one
ncclAllGather of [counts | ids] per rank on a side stream, double-buffered). --scaling weak (default): every rank owns its own 10 M
entities. --scaling strong: BASELINE config 4 - ONE 10 M scene partitioned over the ranks by cell hash (+ 100 k skinned instances by
index, timed as an extra); the union of the gathered lists is checked against the unsharded result.

value = entities resident on all ranks x frusta / wall time per step (max over ranks, barrier + synchronize on both
sides of exactly K steps). Inputs are resident in HBM before the timed region starts; the frustum (256 B) is a kernel
argument.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6290 GB/s is the measured copy ceiling


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def ids_sha256(ids):
    import hashlib

    return hashlib.sha256(np.sort(np.asarray(ids, np.int32)).tobytes()).hexdigest()


def golden_sha(scene, camera="default"):
    """all-types sha256 of the reference's visible ids for one of the scenes this bench times (tests/golden/cull_bench_scenes.json,
    written by tests/golden/make_golden_bench_scenes.py from the reference's own CullingSystemImpl), or None."""
    try:
        g = json.load(open(os.path.join(ROOT, "tests", "golden", "cull_bench_scenes.json")))["scenes"][scene]["cameras"][camera]
        return g["all_types_sha256"]
    except Exception:  # noqa: BLE001 - a missing fixture must not break the bench line; the line then says "unchecked"
        return None


def check_ids(res, scene, camera="default", frustum=0):
    """'reference' when the visible ids of (scene, camera) are the reference's (sha256 of the sorted ids), raises when they are not,
    'unchecked' when no digest exists for this scene / size."""
    want = golden_sha(scene, camera)
    if want is None:
        return "unchecked"
    got = ids_sha256(res.all_ids(frustum)[0])
    if got != want:
        raise SystemExit(f"bench: visible ids of {scene}/{camera} differ from the reference's ({got} != {want})")
    return "reference"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--entities", type=int, default=10_000_000, help="entities per GPU")
    ap.add_argument("--variant", choices=["sparse", "dense"], default="sparse")
    ap.add_argument("--camera", choices=["default", "all_visible"], default="default",
                    help="all_visible: camera far outside looking at the whole cube (every sphere is fetched and visible) - the pure streaming case used to calibrate PMC byte counters")
    ap.add_argument("--force-collective", action="store_true", help="run the N>1 code path (RCCL all-gather of visible ids) even with one rank")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N>1: weak = every rank owns --entities entities (default); strong = BASELINE config 4, one --entities scene partitioned by cell hash")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the default-camera loops (warm-up, timed, event-timed): the command whose rocprofv3 --kernel-trace --stats summary is committed under profiles/")
    ap.add_argument("--no-extras", action="store_true", help="skip the dense-variant / transform / skin side measurements")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the two rocprofv3 --pmc child passes; quote the committed profiles/rNN/traffic.json")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--default-stream", action="store_true", help="N > 1: launch on the legacy default stream (the pre-round-2 behaviour, for comparison)")
    ap.add_argument("--skinned-instances", type=int, default=100_000, help="--scaling strong: skinned instances of BASELINE config 4, sharded by index (0 = skip)")
    ap.add_argument("--ranks-share-gpu", action="store_true",
                    help="TEST MODE for the N > 1 code path on a one-GPU box: every rank uses cuda:0, torch.distributed runs on gloo, and the exchange's collective must be carried by LMX_RCCL_LIBRARY = tests/_build/libloopback_rccl.so (RCCL refuses two ranks on one device). Checks that the path runs and what it ships; its timings mean nothing")
    ap.add_argument("--big-entities", type=int, default=100_000_000, help="extras: entity count of the config-5-sized single-GPU legs (0 = skip)")
    ap.add_argument("--config5-frame", action="store_true",
                    help="N > 1 (or --force-collective): also time BASELINE config 5's frame - the rank's entities under the 8 shadow-cascade frusta in ONE pass (pass width 8) and "
                         "ONE collective (lmx_exchange_cull_many) -> config.config5_frame. Off by default: this mode has not run on hardware yet (no GPU budget was left when it was written)")
    ap.add_argument("--no-ab", action="store_true", help="skip extra.ab_variants (tools/ab_variants.py: the not-yet-timed kernel experiments, one child process per leg, after every other measurement)")
    ap.add_argument("--ab-budget", type=float, default=300.0, help="seconds the A/B children may take together")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from lumixengine_amd import api, scenes
    from lumixengine_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if args.ranks_share_gpu:
        if not os.environ.get("LMX_RCCL_LIBRARY"):
            raise SystemExit("--ranks-share-gpu needs LMX_RCCL_LIBRARY (tests/cpp/loopback_rccl.cpp): RCCL refuses two ranks on one device")
        local_rank = 0
    red_dev = "cpu" if args.ranks_share_gpu else "cuda"  # where torch.distributed's few scalars live (gloo in the test mode)
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_collective
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL prints a version banner on the C-level stdout at communicator creation; stdout must carry the JSON line only
        with c_stdout_to_stderr():
            if args.ranks_share_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()

    if use_dist and not args.default_stream:
        # N > 1: launches go to a stream of their own, not the legacy default stream. Every event record / stream wait that involves
        # the default stream makes the runtime look at all other blocking streams, and a step of the exchange path issues four of them
        # (measured with a world of one rank: 41 us of host time per step on the default stream - 15.6 us for ONE record + wait pair).
        torch.cuda.set_stream(torch.cuda.Stream())
    ctx = api.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)  # launches, the events of `timed` and torch.cuda.synchronize() share one stream

    def barrier():
        if use_dist:
            dist.barrier()

    def timed(fn, steps):
        """K steps bracketed by barrier + synchronize. The host may run at most ~48 launches ahead of the GPU (an event every
        16 steps, wait on the one recorded 32-48 steps earlier): the HIP runtime drains the whole queue once every several
        hundred launches, which costs nothing when the GPU keeps up but shows up as one stall of backlog x kernel time (tens of
        ms after ~850 queued 43 us culls, tools/hip_queue_stall_probe.py) when a loop is GPU-paced and unbounded."""
        barrier()
        torch.cuda.synchronize()
        events = []
        t0 = time.perf_counter()
        for i in range(steps):
            fn()
            if (i & 15) == 15:
                e = torch.cuda.Event()
                e.record()
                events.append(e)
                if len(events) > 2:
                    events.pop(0).synchronize()
        torch.cuda.synchronize()
        barrier()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        if use_dist:
            t = torch.tensor([ms], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms = float(t.item())
        return ms

    # ---- the headline workload -----------------------------------------------------------------------------
    N = args.entities
    half = 15000.0 if args.variant == "sparse" else 5000.0
    t0 = time.time()
    strong = args.scaling == "strong" and (world > 1 or args.force_collective)
    sc = scenes.cull_scene(N, half, seed=2 if strong else 2 + rank)
    cs = api.CullingSystem(ctx)
    if strong:
        mine = D.shard_by_cell(sc["pos"], world, rank)  # cells stay whole: no cell is classified on two GPUs
        cs.build(sc["entity"][mine], sc["type"][mine], sc["pos"][mine], sc["radius"][mine])
        log(f"[rank {rank}] owns {int(mine.sum())} of {N} entities")
    else:
        cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    stats = cs.stats()
    log(f"[rank {rank}] scene {args.variant}: {N} entities, {stats['cells']} cells, {stats['chunks']} chunks, build {time.time() - t0:.1f}s")
    frustum = api.viewport_frustum()  # default player viewport at the origin
    baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        baseline = CpuBaseline(scenes, sc, frustum, log)  # builds the CPU-side scene on a background thread while the GPU legs run
    # Process warm-up, outside every timed region: the HIP runtime pays a one-time ~50 ms stall around the 850th kernel launch
    # of a process (measured: tools/hip_queue_stall_probe.py, a 3000-cull loop stalls once in launches 750-1000 and never again in the
    # next 8000). Without this it lands in whichever timed loop crosses that count (a --steps 2000 run read 45 us per step
    # instead of 20). 1200 culls of the real scene = 25 ms.
    for _ in range(1200):
        cs.cull(frustum)
    torch.cuda.synchronize()
    if args.camera == "all_visible":
        frustum = api.viewport_frustum(pos=(0.0, 0.0, 4.0 * half), far=20.0 * half)
    n_frusta = 1

    if use_dist:
        # One exchange per frame, native (csrc/lmx_capi_exchange.hip): the cull's gather kernels write [8 counts | cap ids] into the
        # send buffer, ONE ncclAllGather per frame runs on a side stream, frames alternate between two slots so the next cull
        # overlaps this frame's gather. torch.distributed only carries the 128-byte RCCL id, the barrier and the timing reduction.
        uid = torch.zeros(128, dtype=torch.uint8, device=red_dev)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(api.exchange_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0)
        probe = int(cs.cull(frustum).counts()[0].sum())
        t = torch.tensor([probe], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        cap = (int(t.item()) * 5 // 4 + 1023) // 1024 * 1024  # 1.25 x the largest visible count of any rank
        log(f"[rank {rank}] exchange: creating the RCCL communicator (ids per rank {cap})")
        with c_stdout_to_stderr():
            xchg = api.VisibleExchange(ctx, rank, world, uid.cpu().numpy().tobytes(), cap)
            step_probe = xchg.cull(frustum)  # the first collective creates RCCL's channels (and may print)
            xchg.wait(step_probe)
        log(f"[rank {rank}] exchange ready")
        fr_c = np.ascontiguousarray(frustum, api.SHIFTED_FRUSTUM).reshape(-1)
        fr_ptr, x_cull, xh = api._ptr(fr_c), ctx.lib.lmx_exchange_cull, xchg.h
        import ctypes as C

        slot_c = C.c_uint32(0)

        def step():
            if x_cull(xh, fr_ptr, api.TYPE_ALL, C.byref(slot_c)) != 0:
                raise RuntimeError(ctx.lib.lmx_last_error(ctx.h).decode())
            return slot_c.value
    else:
        import ctypes as C

        fr_c = np.ascontiguousarray(frustum, api.SHIFTED_FRUSTUM).reshape(-1)
        fr_ptr, lib, h = api._ptr(fr_c), ctx.lib, ctx.h
        rec_p, rec_n = C.c_void_p(), C.c_uint32(0)

        def step():  # one cull incl. compaction: k_cull_tile, then k_cull_pack gathers the shard windows into [8 counts | ids] in HBM
            if lib.lmx_cull(h, 0, fr_ptr, 1, api.TYPE_ALL) != 0 or lib.lmx_cull_pack_device(h, 0, 0, C.byref(rec_p), C.byref(rec_n)) != 0:
                raise RuntimeError(lib.lmx_last_error(h).decode())

    for _ in range(args.warmup):
        step()
    ms_per_step = timed(step, args.steps)  # (the closing synchronize of `timed` also drains the side stream's last gathers)
    value = (N if strong else N * world) * n_frusta / (ms_per_step * 1e-3)
    ms_cull_only = ms_host_list = None
    if not use_dist:
        ms_cull_only = timed(lambda: cs.cull(frustum), args.steps)  # rounds 1 / 2's step: the cull kernel alone, ids in shard windows
        # SURVEY.md 8d (i), second form: the list on the HOST - what the CullingSystem adapter does per view: cull + lmx_cull_map_all (packed
        # record copied into the library's pinned buffer, sized from the previous frame; ONE host wait), PCIe included
        ids_p, cnt8 = C.POINTER(C.c_int32)(), (C.c_uint32 * 8)()

        def step_host():
            if lib.lmx_cull(h, 0, fr_ptr, 1, api.TYPE_ALL) != 0 or lib.lmx_cull_map_all(h, 0, 0, C.byref(ids_p), cnt8) != 0:
                raise RuntimeError(lib.lmx_last_error(h).decode())

        for _ in range(5):
            step_host()
        ms_host_list = timed(step_host, min(args.steps, 100))
    res = cs.cull(frustum)
    visible = int(res.counts()[0].sum())
    # identity of what was timed: the ids, not just their number, against the reference's CullingSystemImpl (same seeded scene)
    headline_scene = {("sparse", 10_000_000): "sparse_10m"}.get((args.variant, N)) if (args.camera == "default" and (strong or rank == 0)) else None
    ids_checked = "unchecked"
    if strong and use_dist:
        pass  # the union over ranks is compared with the unsharded cull below, and that one with the reference's digest
    elif headline_scene:
        ids_checked = check_ids(res, headline_scene)
    log(f"[rank {rank}] headline: {ms_per_step * 1e3:.2f} us per step, {visible} visible, ids {ids_checked}")

    dist_info = {}
    if use_dist:
        # sanity of the exchange step: every rank's record arrived, nothing overflowed, own record == local cull; in the strong
        # (config 4) split the union over ranks must be the unsharded visible set
        slot = step()
        xchg.wait(slot)
        torch.cuda.synchronize()
        parsed, seen = [], []
        for r in range(world):
            counts, ids = xchg.read(slot, r)
            seen.append(int(counts.sum()))
            parsed.append(ids)
        dist_info["allgather_visible_counts"] = seen
        dist_info["exchange"] = f"one ncclAllGather per frame of [8 counts | {xchg.cap} ids] per rank, side stream, double-buffered (lmx_exchange_*)"
        dist_info["ranks_seen_by_rccl"] = len(seen)
        assert len(seen) == world and max(seen) <= xchg.cap, (seen, xchg.cap)
        local = cs.cull(frustum, view=2)  # (view 0 / 1 hold exchange frames, `res` is long overwritten by the roofline legs)
        assert seen[rank] == visible and np.array_equal(np.sort(parsed[rank]), np.sort(local.all_ids(0)[0])), "gathered ids differ from the local cull result"
        if strong and rank == 0:
            ctx_whole = api.Context(local_rank)  # a context of its own: a context holds ONE culling set, and this rank's shard stays resident
            whole = api.CullingSystem(ctx_whole)
            whole.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
            res_whole = whole.cull(frustum)
            want = np.sort(res_whole.all_ids(0)[0])
            assert np.array_equal(np.sort(np.concatenate(parsed)), want), "union of the ranks' lists != unsharded cull"
            if args.variant == "sparse" and N == 10_000_000 and args.camera == "default":
                ids_checked = check_ids(res_whole, "sparse_10m")
            dist_info["union_equals_unsharded"] = True
            dist_info["visible_total"] = int(len(want))
            del whole
            ctx_whole.close()
        log(f"[rank {rank}] exchange verified: {dist_info}")
        def config4_frame(ctx_x, step_x, note):
            """The rest of BASELINE config 4 next to a rank's cell shard of the one 10 M scene: 100 k skinned instances (one shared
            10 k-vertex mesh, 64 bones) sharded by index - no exchange, every rank skins its own. One simulated frame = cull of the
            shard + exchange (`step_x`) + pose -> palette -> vertices of the rank's instances; MAX over ranks, like the headline."""
            mine_i = D.shard_by_index(args.skinned_instances, world, rank)
            s_sk = scenes.skeleton(64, seed=4)
            verts_sk, skin_sk = scenes.skinned_mesh(10_000, 64, seed=6)
            sk = api.Skinning(ctx_x)
            model_sk = sk.addModel(s_sk["parents"], s_sk["bind"], s_sk["first_nonroot"])
            mesh_sk = sk.addMesh(verts_sk, skin_sk)
            sk.setInstances(np.full(len(mine_i), model_sk, np.uint32), np.full(len(mine_i), mesh_sk, np.uint32))
            pos_sk, rot_sk = scenes.relative_poses(len(mine_i), 64, seed=8 + rank)
            d_pos_sk, d_rot_sk = torch.from_numpy(pos_sk).cuda(), torch.from_numpy(rot_sk).cuda()
            del pos_sk, rot_sk
            sk.setPoseSourceDevice(d_pos_sk.data_ptr(), d_rot_sk.data_ptr(), len(mine_i) * 64)

            def frame_c4():
                step_x()
                sk.run()

            for _ in range(2):
                frame_c4()
            ms_c4 = timed(frame_c4, 10)
            out = {
                "scaling": "strong", "entities_total": N, "skinned_instances_total": args.skinned_instances,
                "skinned_instances_this_rank": int(len(mine_i)), "verts_per_instance": 10_000,
                "ms_per_frame_max_over_ranks": ms_c4, "frames_per_sec": 1e3 / ms_c4,
                "skinned_verts_per_sec_all_ranks": args.skinned_instances * 10_000 / (ms_c4 * 1e-3),
                "what": "cull of the rank's cell shard + native all-gather + pose/palette/vertex kernels of the rank's instances (shard_by_index)" + note}
            log(f"[rank {rank}] config 4 frame: {out}")
            del sk, d_pos_sk, d_rot_sk
            return out

        if strong and args.skinned_instances:
            dist_info["config4_frame"] = config4_frame(ctx, step, "")
        elif not strong and args.skinned_instances and not args.no_extras and not args.headline_only:
            # weak run (the driver's SCALE line): the strong-scaling frame of BASELINE config 4 rides along as an extra, in a context
            # and a communicator of its own, so one `--gpus N` run yields both curves; N = 1's point is extra.target_frames_per_sec
            try:
                sc4 = sc if rank == 0 else scenes.cull_scene(N, half, seed=2)  # (rank 0's weak scene IS seed 2)
                mine4 = D.shard_by_cell(sc4["pos"], world, rank)
                ctx4 = api.Context(local_rank)
                ctx4.set_stream(torch.cuda.current_stream().cuda_stream)
                cs4 = api.CullingSystem(ctx4)
                cs4.build(sc4["entity"][mine4], sc4["type"][mine4], sc4["pos"][mine4], sc4["radius"][mine4])
                uid4 = torch.zeros(128, dtype=torch.uint8, device=red_dev)
                if rank == 0:
                    uid4.copy_(torch.frombuffer(bytearray(api.exchange_unique_id()), dtype=torch.uint8))
                dist.broadcast(uid4, 0)
                t4 = torch.tensor([int(cs4.cull(frustum).counts()[0].sum())], dtype=torch.int64, device=red_dev)
                t4_sum = t4.clone()
                dist.all_reduce(t4, op=dist.ReduceOp.MAX)
                dist.all_reduce(t4_sum, op=dist.ReduceOp.SUM)
                cap4 = (int(t4.item()) * 5 // 4 + 1023) // 1024 * 1024
                with c_stdout_to_stderr():
                    xchg4 = api.VisibleExchange(ctx4, rank, world, uid4.cpu().numpy().tobytes(), cap4)
                    xchg4.wait(xchg4.cull(frustum))
                xh4, slot4 = xchg4.h, C.c_uint32(0)

                def step4():
                    if x_cull(xh4, fr_ptr, api.TYPE_ALL, C.byref(slot4)) != 0:
                        raise RuntimeError(ctx4.lib.lmx_last_error(ctx4.h).decode())

                c4 = config4_frame(ctx4, step4, "; measured in the weak run as an extra")
                c4["visible_total"] = int(t4_sum.item())
                want4 = None
                if args.variant == "sparse" and N == 10_000_000 and args.camera == "default":
                    try:
                        want4 = sum(json.load(open(os.path.join(ROOT, "tests", "golden", "cull_bench_scenes.json")))["scenes"]["sparse_10m"]["cameras"]["default"]["counts"])
                    except Exception:  # noqa: BLE001
                        want4 = None
                if want4 is not None:
                    assert c4["visible_total"] == want4, f"config 4: the ranks' shards see {c4['visible_total']} ids, the reference {want4}"
                    c4["visible_total_is"] = "the reference's count (tests/golden/cull_bench_scenes.json)"
                dist_info["config4_frame"] = c4
                xchg4.close()
                del cs4
                ctx4.close()
            except AssertionError:
                raise
            except Exception as e:  # noqa: BLE001 - an extra must not take the headline line with it
                dist_info["config4_frame"] = {"error": repr(e)}
                log(f"[rank {rank}] config 4 extra failed: {e!r}")
        if args.config5_frame and not strong:
            # BASELINE config 5 across GPUs: every rank's entities (weak: --entities per rank, config 5 has 12.5 M per GPU) under the frame's 8
            # cascade frusta in one pass over the spheres and ONE all-gather of 8 sub-records (pipeline.cpp:1252-1258 culls them one by one).
            # Every rank runs the same sequence of collectives whatever happens in between: failures are carried as a flag and agreed on
            # (all_reduce MIN) before the timed loop, never raised between two collectives.
            class TorchColl:  # the scalars the ranks agree on travel over torch.distributed (gloo in the --ranks-share-gpu test mode)
                @staticmethod
                def _reduce(v, op):
                    t = torch.tensor([int(v)], dtype=torch.int64, device=red_dev)
                    dist.all_reduce(t, op=op)
                    return int(t.item())

                max_int = staticmethod(lambda v: TorchColl._reduce(v, dist.ReduceOp.MAX))
                min_int = staticmethod(lambda v: TorchColl._reduce(v, dist.ReduceOp.MIN))

                @staticmethod
                def bcast_bytes(b):
                    u = torch.zeros(128, dtype=torch.uint8, device=red_dev)
                    if b is not None:
                        u.copy_(torch.frombuffer(bytearray(b), dtype=torch.uint8))
                    dist.broadcast(u, 0)
                    return u.cpu().numpy().tobytes()

            fr8 = np.concatenate([api.viewport_frustum(**kw) for kw in scenes.config5_cascade_kwargs()])
            c5 = D.config5_frame(api, fr8, ctx, cs, rank, world, N, TorchColl, timed, quiet=c_stdout_to_stderr)
            dist_info["config5_frame"] = c5
            log(f"[rank {rank}] config 5 frame: {c5}")
        xchg.close()
        log(f"[rank {rank}] exchange closed")

    # ---- roofline of the dominant kernel (k_cull_tile): a HIP event pair per launch, on the launch stream, filled by the launch itself
    # (hipExtLaunchKernelGGL: the dispatch's own begin / end timestamps - what rocprofv3's kernel trace reports. Events recorded
    # around the launch also time the command processor's work on the pair: +3 us, 7 % of the 40 us all-test launch) ------------
    # Three regimes of the same kernel on the same 10 M geometry:
    #   default camera   hierarchical skip: ~95 % of the tiles end at the tile-level box test. Latency regime; the algorithmic
    #                    bytes (20 B per RESIDENT entity) are mostly never moved, so it is reported as an effective rate only.
    #   all_accept       a camera that sees the whole cube: every tile is TILE_ACCEPT, ids are copied: 4 B read + 4 B written per entity.
    #   all_test         the same positions with radii in (300, 330]: every cell is a "big" cell (culling_system.cpp:140,342-344) and
    #                    skips the AABB pre-test, so every sphere is fetched and tested: 16 B + 4 B read per entity + 4 B per visible
    #                    id. Here moved bytes == algorithmic bytes: this leg, cache-cold, is `roofline.frac`.
    scrub = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")  # 1 GiB > 256 MiB Infinity Cache

    scrub32 = scrub.view(torch.int32)

    def kernel_times(csys, fr, steps, cold):
        """avg device ms of k_cull_tile over `steps` culls. cold: evict the Infinity Cache before every cull - True / "read": a 1 GiB
        read-only reduction (the cache is left full of CLEAN lines: the cull's misses go to HBM and evict for free); "write": a 1 GiB
        read-modify-write (left full of DIRTY lines: every miss of the cull first writes a victim line back, i.e. the kernel shares
        HBM with ~200 MB of write-backs it did not cause - reported, but not the roofline leg)."""
        ctx.profile_reset()
        ctx.profile_enable(True)
        for _ in range(steps):
            if cold == "write":
                scrub.add_(1)
            elif cold:
                scrub32.sum()
            csys.cull(fr)
        ctx.synchronize()
        ctx.profile_enable(False)
        ms_s, n_s = ctx.profile_get(api.K_CULL_SPHERES)
        return ms_s / max(n_s, 1)

    gbps = lambda b, ms: None if ms != ms else round(b / (ms * 1e-3) / 1e9, 1)
    rnd = lambda x, n: None if x != x else round(x, n)
    frac = lambda b, ms: None if ms != ms else round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
    avg_default_ms = kernel_times(cs, frustum, args.steps, cold=False)
    alg_bytes = 20.0 * N + 4.0 * visible
    legs = {"default_camera": {
        "visible": visible, "warm_avg_launch_ms": rnd(avg_default_ms, 5), "algorithmic_bytes": alg_bytes,
        "effective_GBps": gbps(alg_bytes, avg_default_ms),
        "note": "effective rate: 20 B per resident entity although ~95 % of the tiles are rejected by their box and never fetched; not a roofline fraction",
    }}
    nan = float("nan")
    test_cold_ms = test_warm_ms = nan
    test_bytes = 0.0
    if not args.headline_only:
        legs["default_camera"]["cold_avg_launch_ms"] = rnd(kernel_times(cs, frustum, min(args.steps, 50), cold=True), 5)
        accept_fr = api.viewport_frustum(pos=(0.0, 0.0, 4.0 * half), far=20.0 * half)
        accept_visible = int(cs.cull(accept_fr).counts()[0].sum())
        acc_warm = kernel_times(cs, accept_fr, min(args.steps, 50), cold=False)
        acc_cold = kernel_times(cs, accept_fr, min(args.steps, 50), cold=True)
        acc_coldw = kernel_times(cs, accept_fr, min(args.steps, 50), cold="write")
        acc_bytes = 8.0 * N  # 4 B id read + 4 B id written per entity; cell keys and spheres are not touched
        legs["all_accept"] = {"visible": accept_visible, "moved_bytes": acc_bytes, "warm_avg_launch_ms": rnd(acc_warm, 5), "cold_avg_launch_ms": rnd(acc_cold, 5),
                              "warm_GBps": gbps(acc_bytes, acc_warm), "cold_GBps": gbps(acc_bytes, acc_cold), "warm_frac": frac(acc_bytes, acc_warm), "cold_frac": frac(acc_bytes, acc_cold),
                              "cold_after_dirty_scrub_frac": frac(acc_bytes, acc_coldw), "vs_20B_formula_cold_GBps": gbps(20.0 * N + 4.0 * accept_visible, acc_cold)}
        # all_test: same positions, every sphere "big" -> every cell CELL_TEST
        sc_t = dict(sc)
        sc_t["radius"] = scenes.all_test_radii(N)
        cs_t = api.CullingSystem(ctx)
        cs_t.build(sc_t["entity"], sc_t["type"], sc_t["pos"], sc_t["radius"])
        for _ in range(20):
            cs_t.cull(frustum)
        res_t = cs_t.cull(frustum)
        test_visible = int(res_t.counts()[0].sum())
        test_ids_checked = check_ids(res_t, "all_test_10m") if (N == 10_000_000 and args.variant == "sparse" and args.camera == "default" and rank == 0) else "unchecked"
        test_bytes = 20.0 * N + 4.0 * test_visible
        test_warm_ms = kernel_times(cs_t, frustum, min(args.steps, 50), cold=False)
        test_cold_ms = kernel_times(cs_t, frustum, min(args.steps, 50), cold=True)
        test_coldw_ms = kernel_times(cs_t, frustum, min(args.steps, 50), cold="write")
        legs["all_test"] = {"visible": test_visible, "moved_bytes": test_bytes, "warm_avg_launch_ms": rnd(test_warm_ms, 5), "cold_avg_launch_ms": rnd(test_cold_ms, 5),
                            "warm_GBps": gbps(test_bytes, test_warm_ms), "cold_GBps": gbps(test_bytes, test_cold_ms), "warm_frac": frac(test_bytes, test_warm_ms),
                            "cold_frac": frac(test_bytes, test_cold_ms), "cold_after_dirty_scrub_avg_launch_ms": rnd(test_coldw_ms, 5),
                            "cold_after_dirty_scrub_frac": frac(test_bytes, test_coldw_ms), "cells": cs_t.stats()["cells"], "visible_ids": test_ids_checked,
                            "note": "warm = back-to-back frames (SURVEY.md 8d: 'measure with >= 100 back-to-back frames'; the 200 MB working set stays in the 256 MiB Infinity Cache); cold = after a 1 GiB read-only scrub; the 100 M extra (config5_size_single_gpu.all_test) is HBM-cold by size"}
        # the multi-frustum kernel (k_cull_tile<0, ...>) on the same every-sphere-is-tested scene: the frame's 8 shadow-cascade frusta in
        # ONE pass over the spheres (pass width 8, config 5's form). SURVEY.md 8d asks for both fractions here: with 8 frusta the
        # arithmetic intensity (8 x 56 flop per 20 B) sits at the fp32 ridge.
        fr8_t = np.concatenate([api.viewport_frustum(**kw) for kw in scenes.config5_cascade_kwargs()])
        cs_t.setPassWidth(8)
        try:
            for _ in range(5):
                cs_t.cull(fr8_t, view=1)
            v8_t = cs_t.cull(fr8_t, view=1).counts().sum(axis=1)
            ctx.profile_reset()
            ctx.profile_enable(True)
            for _ in range(20):
                scrub32.sum()
                cs_t.cull(fr8_t, view=1)
            ctx.synchronize()
            ctx.profile_enable(False)
            ms8_k, n8_k = ctx.profile_get(api.K_CULL_SPHERES)
            ms8 = ms8_k / max(n8_k, 1)
            bytes8 = 20.0 * N + 4.0 * float(v8_t.sum())
            legs["all_test_8_frusta_one_pass"] = {
                "visible_per_frustum": [int(x) for x in v8_t], "algorithmic_bytes": bytes8, "cold_avg_launch_ms": rnd(ms8, 5), "cold_GBps": gbps(bytes8, ms8),
                "hbm_frac": frac(bytes8, ms8), "flops": 56.0 * N * 8, "valu_frac_of_157_TFLOPs": None if ms8 != ms8 else round(56.0 * N * 8 / (ms8 * 1e-3) / 157.3e12, 4),
                "entity_frustum_tests_per_sec": None if ms8 != ms8 else 8.0 * N / (ms8 * 1e-3),
                "note": "k_cull_tile<F = 0> (runtime frustum count), pass width 8, cache-cold after a read-only 1 GiB scrub; 56 flop per sphere and frustum (SURVEY.md 8d)"}
        finally:
            cs_t.setPassWidth(1)
        del cs_t, sc_t
        # the same regime - every cell CELL_TEST, every sphere fetched and tested - reached the config-2-faithful way: NORMAL radii, the
        # cells classified by the AABB pre-tests (phase A's full work), under an orthographic slab camera that every cell of a one-layer
        # scene straddles (scenes.slab_scene / slab_frustum_kwargs). ~40 % of the spheres are visible: 4 B per visible id matter here.
        if N >= 1_000_000 and args.variant == "sparse" and args.camera == "default":
            sc_b = scenes.slab_scene(N, seed=2)
            fr_b = api.viewport_frustum(**scenes.slab_frustum_kwargs(sc_b["half"]))
            cs_b = api.CullingSystem(ctx)
            cs_b.build(sc_b["entity"], sc_b["type"], sc_b["pos"], sc_b["radius"])
            for _ in range(20):
                cs_b.cull(fr_b)
            res_b = cs_b.cull(fr_b)
            slab_visible = int(res_b.counts()[0].sum())
            slab_checked = check_ids(res_b, "slab_10m", "slab") if (N == 10_000_000 and rank == 0) else "unchecked"
            slab_bytes = 20.0 * N + 4.0 * slab_visible
            slab_warm = kernel_times(cs_b, fr_b, min(args.steps, 50), cold=False)
            slab_cold = kernel_times(cs_b, fr_b, min(args.steps, 50), cold=True)
            legs["all_cell_test_normal_radii"] = {
                "visible": slab_visible, "moved_bytes": slab_bytes, "warm_avg_launch_ms": rnd(slab_warm, 5), "cold_avg_launch_ms": rnd(slab_cold, 5),
                "warm_GBps": gbps(slab_bytes, slab_warm), "cold_GBps": gbps(slab_bytes, slab_cold), "warm_frac": frac(slab_bytes, slab_warm), "cold_frac": frac(slab_bytes, slab_cold),
                "cells": cs_b.stats()["cells"], "visible_ids": slab_checked,
                "note": "scenes.slab_scene: normal radii, one layer of cells, every cell straddles the ortho slab camera's near and far plane -> CELL_TEST through containsAABB / intersectsAABB (no big-sphere shortcut)"}
            del cs_b, sc_b
    del scrub
    traffic, traffic_note = (None, None)
    if rank == 0 and world == 1 and not args.headline_only and not args.no_live_traffic:
        traffic, traffic_note = measure_traffic_live(log)
    if traffic is None:
        traffic, traffic_note = load_traffic("k_cull_tile:all_test")
    roof_ms = test_cold_ms if test_cold_ms == test_cold_ms else avg_default_ms
    roof_bytes = test_bytes if test_cold_ms == test_cold_ms else alg_bytes
    roofline = {
        "kernel": "k_cull_tile",
        "bound": "hbm",
        "leg": "all_test, cache-cold after a read-only 1 GiB scrub (every sphere fetched and tested: moved bytes == SURVEY.md 8d's 20 B/entity + 4 B/visible id)" if test_cold_ms == test_cold_ms
               else "default camera only (--headline-only): effective rate, NOT a roofline fraction",
        "achieved": round(roof_bytes / (roof_ms * 1e-3) / 1e9, 1),
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": round(roof_bytes / (roof_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        "traffic": traffic,  # HBM bytes per launch of the all_test leg: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, run as child processes of THIS bench (rank 0, N = 1); the committed profiles/rNN/traffic.json only when rocprofv3 is unavailable (traffic_source says which)
        "traffic_source": traffic_note,
        "algorithmic_bytes_per_launch": roof_bytes,
        "avg_launch_ms": round(roof_ms, 5),
        "avg_launch_ms_is": "mean over the leg's launches of hipEventElapsedTime on the event pair the launch itself fills (hipExtLaunchKernelGGL start / stop events on the launch stream): the dispatch's begin -> end, as in rocprofv3's kernel trace",
        "measured_copy_ceiling_GBps": 6290.0,
        "legs": legs,
    }

    result = {
        "metric": "entities_culled_per_sec",
        "value": value,
        "unit": "entities/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"BASELINE config 2: {N} static entities per GPU ({args.variant}, cube +-{half:g}), 1 frustum (fov 60, 16:9, near 0.1, far 10000), cull + compaction",
            "entities_per_gpu": N,
            "frusta": n_frusta,
            "cells_per_gpu": stats["cells"],
            "visible_per_gpu": visible,
            "sharding": ("one scene partitioned by cell hash" if strong else "own entities per rank") + ", native RCCL all-gather of visible ids" if use_dist else "single GPU",
        },
        "roofline": roofline,
    }
    result["value_is"] = ("EFFECTIVE rate of the BASELINE config-2 camera: resident entities / wall time of one cull incl. compaction (k_cull_tile + k_cull_pack); ~95 % of the "
                          "tiles are rejected by their boxes and never fetched, so value x 20 B is NOT a memory rate - `value_streaming` and `roofline` are")
    if ms_cull_only is not None:
        result["value_cull_only"] = N * world / (ms_cull_only * 1e-3)
        result["ms_per_step_cull_only"] = ms_cull_only
        result["value_incl_host_readback"] = N * world / (ms_host_list * 1e-3)
        result["ms_per_step_incl_host_readback"] = ms_host_list
        result["value_incl_host_readback_is"] = "cull + lmx_cull_map_all: the visible ids (and the 8 per-type counts) in host memory, one host wait per cull, PCIe included - what GpuCullingSystem::cull pays per view before it builds the CullResult pages; never `value`"
    if test_cold_ms == test_cold_ms:
        result["value_streaming"] = N / (test_cold_ms * 1e-3)  # every sphere fetched and tested, cache-cold: what roofline.frac is the fraction of
    result["config"]["visible_ids"] = ids_checked  # 'reference': sha256 of the sorted ids == the reference CullingSystemImpl's on the same seeded scene

    result["config"].update(dist_info)
    if args.ranks_share_gpu:
        result["config"]["TEST_MODE"] = "--ranks-share-gpu: all ranks on cuda:0, gloo + shared-memory collective (tests/cpp/loopback_rccl.cpp); timings are meaningless"
    if rank == 0 and world == 1:
        if not args.no_extras:
            result["extra"] = {}
            try:
                extras(ctx, api, scenes, torch, timed, N, log, args.big_entities, out=result["extra"])
            except Exception as e:  # noqa: BLE001 - a side measurement must not take the headline (already measured and digest-checked) with it
                import traceback

                result["extra"]["error"] = f"extras stopped at: {e!r}"
                result["extra"]["error_traceback_tail"] = traceback.format_exc()[-1500:]
                log("extras FAILED (the legs before the failure are kept):\n" + traceback.format_exc())
        if baseline is not None:
            try:
                result["cpu_baseline"] = baseline.measure()
            except Exception as e:  # noqa: BLE001 - the GPU numbers above are measured; say what went wrong instead of losing the line
                result["cpu_baseline"] = {"value": None, "unit": "entities/s", "cores": 0, "kind": "reference", "sample": "not measured", "error": repr(e)}
        if not args.no_extras and not args.no_ab and not args.headline_only and "extra" in result:
            # LAST: every number above is taken. The experiments run in child processes with contexts of their own and a hard timeout; this
            # process makes no further HIP call that a misbehaving variant could hold up before the line is printed.
            try:
                torch.cuda.synchronize()
                # the like-for-like N = 1 point of the `--gpus N` curve: the N > 1 step (cull + pack + ncclAllGather on the side stream, two
                # slots) with a world of ONE rank, in a child process (this one has no process group); `value` above is the step WITHOUT it
                result["extra"]["exchange_path_one_rank"] = exchange_path_one_rank(args, log)
                # the same step with the all-gather on the cull stream (LMX_EXCHANGE_INLINE=1: four API calls instead of seven, no overlap of
                # the next cull with this frame's gather) - an experiment of the step's host cost, same records
                inl = exchange_path_one_rank(args, log, extra_env={"LMX_EXCHANGE_INLINE": "1"})
                inl.pop("what", None)
                result["extra"]["exchange_path_one_rank"]["inline_gather"] = inl
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import ab_variants

                result["extra"]["ab_variants"] = ab_variants.run_all(log, budget_s=args.ab_budget)
            except Exception as e:  # noqa: BLE001 - an extra must not take the line with it
                result["extra"]["ab_variants"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.destroy_process_group()


def exchange_path_one_rank(args, log, timeout_s=150.0, extra_env=None):
    """`bench.py --force-collective --headline-only` as a child: ms per step / entities per second of the exchange path with one rank."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--force-collective", "--headline-only", "--no-cpu-baseline", "--no-live-traffic", "--no-extras",
           "--steps", str(max(args.steps, 500)), "--warmup", str(args.warmup), "--entities", str(args.entities), "--variant", args.variant]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300 + (1 if extra_env else 0)), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    env.update(extra_env or {})
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT, stdin=subprocess.DEVNULL)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"rc {r.returncode}: {r.stderr[-300:]}"}
        c = json.loads(lines[-1])
        out = {"ms_per_step": c["ms_per_step"], "value": c["value"], "unit": c["unit"], "steps": c["steps"], "visible_ids": c["config"].get("visible_ids"),
               "what": "cull + k_cull_pack into the send buffer + ONE ncclAllGather on the side stream per step, two slots in flight, world of one rank (bench.py --force-collective): "
                       "the step `--gpus N` times for N > 1; host-bound (seven API calls), see DESIGN.md section 5"}
        log(f"[exchange path, one rank{', ' + str(extra_env) if extra_env else ''}] {out['ms_per_step'] * 1e3:.2f} us per step")
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


class c_stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block; the C library's own stdout buffer is flushed before fd 1 is restored
    (RCCL 2.26 printf's its banner into that buffer: without the flush it would surface on the real stdout at exit, after the JSON line)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import ctypes

        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def measure_traffic_live(log):
    """HBM bytes per launch of the roofline leg's kernel, measured in THIS run: two rocprofv3 passes (--pmc FETCH_SIZE, then --pmc
    WRITE_SIZE: they cannot share a pass; kernel trace only) over tools/run_workload.py --workload cull_all_test --cold read - the same
    scene, camera and scrub as the all_test cold leg above - in a child process. Returns (bytes, note) or (None, None) when
    rocprofv3 is missing / fails / times out (the committed profiles/rNN/traffic.json is quoted then)."""
    import csv
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, None
    vals = {}
    tmp = tempfile.mkdtemp(prefix="lmx_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable,
                   os.path.join(ROOT, "tools", "run_workload.py"), "--workload", "cull_all_test", "--steps", "16", "--cold", "read"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=180)
            if r.returncode != 0:
                log(f"live traffic: rocprofv3 --pmc {counter} failed with {r.returncode}")
                return None, None
            v = []
            for root, _, files in os.walk(out):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        with open(os.path.join(root, f)) as fh:
                            for row in csv.DictReader(fh):
                                if "k_cull_tile" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                                    v.append(float(row["Counter_Value"]))
            if len(v) < 8:
                log(f"live traffic: only {len(v)} {counter} samples")
                return None, None
            v = v[len(v) // 4:]  # drop the warm-up launches (not scrubbed)
            vals[counter] = sum(v) / len(v)
    except Exception as e:  # noqa: BLE001 - never let the profiler break the bench line
        log(f"live traffic: {e!r}")
        return None, None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch, write = vals["FETCH_SIZE"] * 1024.0 * 2.0, vals["WRITE_SIZE"] * 1024.0  # KiB; FETCH_SIZE doubled per the guide's gfx950 correction
    log(f"live traffic: FETCH_SIZE {vals['FETCH_SIZE']:.0f} KiB (x2), WRITE_SIZE {vals['WRITE_SIZE']:.0f} KiB per launch")
    return int(round(fetch + write)), ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                                       "tools/run_workload.py --workload cull_all_test --cold read (child processes, same scene / camera / scrub as the all_test cold leg); "
                                       "FETCH_SIZE doubled per the guide's gfx950 correction")


def load_traffic(kernel):
    """HBM bytes per launch of the roofline leg, measured by separate rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot
    share a pass; tools/collect_traffic.sh) and COMMITTED as profiles/rNN/traffic.json - a builder-side measurement of the same
    command, not of this run. None when no such file travels with the repo."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
    if not files:
        return None, "no profiles/r*/traffic.json"
    try:
        t = json.load(open(files[-1])).get(kernel)
        if not t:
            return None, f"{os.path.relpath(files[-1], ROOT)} has no entry for {kernel}"
        return t["hbm_bytes_per_launch"], f"committed file {os.path.relpath(files[-1], ROOT)} (not measured in this run): {t['note']}"
    except Exception as e:  # noqa: BLE001 - a malformed profile file must not break the bench line
        return None, f"unreadable traffic file: {e}"


def extras(ctx, api, scenes, torch, timed, N, log, big_entities=0, out=None):
    """Side measurements (not the headline `value`): dense config-2 variant, 8-frusta pass, config-3 transform + skin.
    `out`: the caller's dict, filled leg by leg (what was measured before a failing leg survives it)."""
    out = {} if out is None else out
    # dense variant of config 2 (cube +-5000: ~37 k cells, ~270 spheres per cell)
    sc = scenes.cull_scene(N, 5000.0, seed=2)
    cs = api.CullingSystem(ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    fr = api.viewport_frustum()
    for _ in range(10):
        cs.cull(fr)
    ms = timed(lambda: cs.cull(fr), 100)
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
    ms_all = timed(lambda: cs.cull(big), 50)
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
        ms_k = timed(cull_keys, 20)
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
    ms_plain = timed(lambda: cs_u.cull(fr_u), 200)
    rng_u = np.random.default_rng(3)
    victims = rng_u.permutation(N)[: 120 * 1000].astype(np.int32).reshape(120, 1000)
    add_pos = rng_u.uniform(-15000.0, 15000.0, size=(120, 1000, 3))
    add_r = np.exp(rng_u.uniform(np.log(0.5), np.log(50.0), size=(120, 1000))).astype(np.float32)
    add_t = np.zeros(1000, np.uint8)
    frame_u = [0]

    def update_frame():
        k = frame_u[0]
        frame_u[0] += 1
        cs_u.removeMany(victims[k])
        cs_u.addMany(np.arange(N + 1000 * k, N + 1000 * (k + 1), dtype=np.int32), add_t, add_pos[k], add_r[k])
        cs_u.cull(fr_u)

    for _ in range(10):
        update_frame()
    ms_upd = timed(update_frame, 100)
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
    n_add_frames, per_frame = (2000, 1000) if N >= 10_000_000 else (200, 1000)
    cs_s.setOption(api.CULL_OPT_AUTO_COMPACTION, 0)
    async_adds = 300_000 if N >= 10_000_000 else 30_000  # the second leg below: adds that arrive while the worker re-sorts
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
    ms8 = timed(lambda: cs.cull(fr8), 50)
    c8 = cs.cull(fr8).counts()
    out["cull8_ms_per_pass"] = ms8
    out["cull8_entity_frustum_tests_per_sec"] = 8.0 * N / (ms8 * 1e-3)
    out["cull8_visible_per_frustum"] = [int(x) for x in c8.sum(axis=1)]
    out["cull8_GBps_algorithmic"] = (20.0 * N + 4.0 * float(c8.sum())) / (ms8 * 1e-3) / 1e9
    del cs

    # config 3 slice: 1 M entities, depth-4 chains, every root moved each frame (transform inputs resident in HBM)
    h = scenes.hierarchy_chains(250_000, 4, seed=2)
    n = len(h["parent"])
    w = api.World(ctx)
    w.build(h["parent"], h["local"])
    roots = np.flatnonzero(h["parent"] < 0).astype(np.int32)
    new_root = scenes.random_transforms(np.random.default_rng(1), len(roots), 4000.0)
    d_ent = torch.from_numpy(roots).cuda()
    d_tr = torch.from_numpy(new_root.view(np.uint8).reshape(len(roots), -1)).cuda()

    def xform_step():
        w.setTransformsDevice(len(roots), d_ent.data_ptr(), d_tr.data_ptr())
        w.propagate()

    for _ in range(10):
        xform_step()
    ms = timed(xform_step, 100)
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
    del w

    # config 3 slice: skinned instances x 64 bones x 10 k verts, shared mesh (2 k instances = 20 M verts per frame)
    n_inst, n_verts = 2000, 10_000
    s = scenes.skeleton(64, seed=4)
    verts, skin = scenes.skinned_mesh(n_verts, 64, seed=6)
    sk = api.Skinning(ctx)
    model = sk.addModel(s["parents"], s["bind"], s["first_nonroot"])
    mesh = sk.addMesh(verts, skin)
    sk.setInstances(np.full(n_inst, model, np.uint32), np.full(n_inst, mesh, np.uint32))
    pos, rot = scenes.relative_poses(n_inst, 64, seed=5)
    d_pos = torch.from_numpy(pos).cuda()
    d_rot = torch.from_numpy(rot).cuda()

    def skin_step():
        sk.uploadPosesDevice(d_pos.data_ptr(), d_rot.data_ptr(), n_inst * 64)
        sk.run()

    for _ in range(5):
        skin_step()
    ms = timed(skin_step, 50)
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
    h3 = scenes.hierarchy_chains(250_000, 4, seed=2, root_extent=6000.0)
    n3 = len(h3["parent"])
    w3 = api.World(ctx)
    w3.build(h3["parent"], h3["local"])
    cs3 = api.CullingSystem(ctx)
    ent3 = np.arange(n3, dtype=np.int32)
    rng3 = np.random.default_rng(3)
    cs3.build(ent3, np.zeros(n3, np.uint8), rng3.uniform(-6000.0, 6000.0, size=(n3, 3)), np.ones(n3, np.float32))
    w3.bindCulling(ent3, rng3.uniform(0.5, 20.0, n3).astype(np.float32))
    roots3 = np.flatnonzero(h3["parent"] < 0).astype(np.int32)
    d_ent3 = torch.from_numpy(roots3).cuda()
    d_tr3 = torch.from_numpy(scenes.random_transforms(rng3, len(roots3), 6000.0).view(np.uint8).reshape(len(roots3), -1)).cuda()
    n_inst3 = 10_000
    sk3 = api.Skinning(ctx)
    model3 = sk3.addModel(s["parents"], s["bind"], s["first_nonroot"])
    mesh3 = sk3.addMesh(verts, skin)
    sk3.setInstances(np.full(n_inst3, model3, np.uint32), np.full(n_inst3, mesh3, np.uint32))
    pos3, rot3 = scenes.relative_poses(n_inst3, 64, seed=7)
    d_pos3, d_rot3 = torch.from_numpy(pos3).cuda(), torch.from_numpy(rot3).cuda()
    fr3 = api.viewport_frustum()

    sk3.setPoseSourceDevice(d_pos3.data_ptr(), d_rot3.data_ptr(), n_inst3 * 64)

    def frame3():
        w3.setTransformsDevice(len(roots3), d_ent3.data_ptr(), d_tr3.data_ptr())
        w3.propagate()
        cs3.cull(fr3)
        sk3.run()

    for _ in range(5):
        frame3()
    ms3 = timed(frame3, 50)
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
    n_inst4 = 100_000
    sk4 = api.Skinning(ctx)
    model4 = sk4.addModel(s["parents"], s["bind"], s["first_nonroot"])
    mesh4 = sk4.addMesh(verts, skin)
    sk4.setInstances(np.full(n_inst4, model4, np.uint32), np.full(n_inst4, mesh4, np.uint32))
    pos4, rot4 = scenes.relative_poses(n_inst4, 64, seed=8)
    d_pos4, d_rot4 = torch.from_numpy(pos4).cuda(), torch.from_numpy(rot4).cuda()
    del pos4, rot4
    fr4 = api.viewport_frustum()

    sk4.setPoseSourceDevice(d_pos4.data_ptr(), d_rot4.data_ptr(), n_inst4 * 64)  # poses are read where the animation system left them

    def frame4():
        cs4.cull(fr4)
        sk4.run()

    for _ in range(2):
        frame4()
    ms4 = timed(frame4, 10)
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
    sk4.setPoseSourceDevice(d_pos4.data_ptr(), d_rot4.data_ptr(), n_inst4 * 64)
    for _ in range(2):
        frame4()
    ms4c = timed(frame4, 10)
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
    sk4.setPoseSourceDevice(d_pos4.data_ptr(), d_rot4.data_ptr(), n_inst4 * 64)
    # the same frame for a renderer that consumes only palettes / vertices (no absolute-pose store, lmx_skin_set_pose_writeback)
    sk4.setPoseWriteback(False)
    for _ in range(2):
        frame4()
    ms4b = timed(frame4, 10)
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

    for _ in range(2):
        frame4a()
    ms4a = timed(frame4a, 10)
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
    # distinct meshes: every instance streams its own 32-byte vertex records from HBM (44 B/vertex moved; SURVEY.md's algorithmic figure is 48)
    n_inst2 = 1500  # 540 MB of mesh data + 180 MB of output: well beyond the 256 MiB Infinity Cache
    sk = api.Skinning(ctx)
    model = sk.addModel(s["parents"], s["bind"], s["first_nonroot"])
    rng = np.random.default_rng(9)
    mesh_ids = []
    for i in range(n_inst2):
        v2 = np.roll(verts, i, axis=0)
        mesh_ids.append(sk.addMesh(v2, np.roll(skin, i, axis=0)))
    sk.setInstances(np.full(n_inst2, model, np.uint32), np.array(mesh_ids, np.uint32))
    d_pos2, d_rot2 = d_pos[:n_inst2].contiguous(), d_rot[:n_inst2].contiguous()

    def skin_step2():
        sk.uploadPosesDevice(d_pos2.data_ptr(), d_rot2.data_ptr(), n_inst2 * 64)
        sk.run()

    for _ in range(5):
        skin_step2()
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(20):
        skin_step2()
    ctx.synchronize()
    ctx.profile_enable(False)
    t_sv, n_sv = ctx.profile_get(api.K_SKIN_VERTICES)
    out["skin_distinct_meshes_instances"] = n_inst2
    out["skin_distinct_meshes_kernel_avg_ms"] = t_sv / max(n_sv, 1)
    out["skin_distinct_meshes_verts_per_sec"] = n_inst2 * n_verts / (t_sv / max(n_sv, 1) * 1e-3)
    out["skin_distinct_meshes_GBps_48B"] = 48.0 * n_inst2 * n_verts / (t_sv / max(n_sv, 1) * 1e-3) / 1e9
    log("extras:", json.dumps(out))
    return out


class CpuBaseline:
    """The reference CPU path on this box's host cores, on the FULL headline workload (10 M entities, same scene, same frustum):
    oracle/_ref (the reference's OWN culling_system.cpp + page_allocator.cpp + math / geometry, compiled in place; threads, Mutex and
    os::mem* underneath are stand-ins) when present, else the plain-C port. The scene is added and the page pool warmed up on a
    background thread while the GPU legs run (the reference allocates - and constructs, 1020 EntityRef{-1} each - one 4 KiB result
    page per visited cell page: ~1 M pages for this scene); the timed part is a thread sweep of the reference's jobs::forEach on a
    PERSISTENT worker pool: median + p10 / p90 per thread count. The plain-C port, which does not construct the page bodies, is timed
    next to it ("port"). Baseline only - the GPU / CPU ratio says nothing about kernel quality."""

    THREADS = (1, 8, 16, 32, 64)

    def __init__(self, scenes, sc, frustum, log):
        import threading

        from oracle import pyoracle

        self.log, self.sc, self.fr = log, sc, np.ascontiguousarray(frustum)
        self.kind = "reference" if pyoracle.have_reference() else "port"
        if self.kind == "port" and not os.path.exists(pyoracle.ORACLE_SO):
            pyoracle.build()
        self.o = pyoracle.Oracle(self.kind)
        self.scenes = scenes
        self.error = None
        self.thread = threading.Thread(target=self._prepare, daemon=True)
        self.thread.start()

    def _prepare(self):
        try:
            t0 = time.time()
            self.ocs = self.o.culling_system()
            self.ocs.add_bulk(self.sc["entity"], self.sc["type"], self.sc["pos"], self.sc["radius"])
            self.t_add = time.time() - t0
            t0 = time.time()
            self.ocs.cull(self.fr, n_threads=8, want_ids=False, cap=0)  # fills the page pool
            self.t_first_cull = time.time() - t0
            self.port_cs = None
            import psutil

            if self.kind == "reference" and psutil.virtual_memory().available > (64 << 30):  # the restatement next to it (another ~8 GiB of pages)
                from oracle import pyoracle

                if not os.path.exists(pyoracle.ORACLE_SO):
                    pyoracle.build()
                self.port_cs = pyoracle.Oracle("port").culling_system()
                self.port_cs.add_bulk(self.sc["entity"], self.sc["type"], self.sc["pos"], self.sc["radius"])
                self.port_cs.cull(self.fr, n_threads=8, want_ids=False, cap=0)
        except Exception as e:  # noqa: BLE001 - reported in the JSON line instead of killing the bench
            self.error = repr(e)

    def measure(self):
        self.thread.join()
        if self.error:
            return {"error": self.error, "kind": self.kind}
        n = len(self.sc["entity"])
        host = os.cpu_count() or 1
        sweep = {}
        for threads in self.THREADS:
            if threads > host and threads != 1:
                continue
            # median of >= 20 timed frames at 1 and 8 threads (after warm-ups: the page pool is filled by _prepare);
            # the wider thread counts only show the trend (the reference's one mutex around the result-page push serialises them)
            want, budget = (20, 40.0) if threads in (1, 8) else (5, 8.0)
            times, t_start = [], time.time()
            while len(times) < want and (time.time() - t_start) < budget:
                t0 = time.perf_counter()
                self.ocs.cull(self.fr, n_threads=threads, want_ids=False, cap=0)
                times.append(time.perf_counter() - t0)
            a = np.array(times)
            sweep[threads] = {"median_ms": round(float(np.median(a)) * 1e3, 3), "p10_ms": round(float(np.percentile(a, 10)) * 1e3, 3),
                              "p90_ms": round(float(np.percentile(a, 90)) * 1e3, 3), "culls": len(times), "entities_per_s": n / float(np.median(a))}
        best = min(sweep, key=lambda k: sweep[k]["median_ms"])
        visible, pages = self.ocs.cull(self.fr, n_threads=1, want_ids=False, cap=0)
        self.log(f"cpu baseline ({self.kind}): {n} entities, add {self.t_add:.1f}s, first cull {self.t_first_cull:.1f}s, sweep " +
                 ", ".join(f"{k}t {v['median_ms']:.1f} ms" for k, v in sweep.items()) + f", {visible} visible, {pages} result pages")
        out = {
            "value": sweep[best]["entities_per_s"],
            "unit": "entities/s",
            "cores": best,
            "kind": self.kind,
            "sample": f"the full headline workload: {n} entities, same scene and frustum as the GPU run; median of {sweep[best]['culls']} culls at {best} thread(s) "
                      f"(best of the sweep {list(sweep)}); one CullResult page per visited cell page, constructed and linked under a mutex, as in the reference ({pages} pages per cull)",
            "host_cores": host,
            "thread_sweep": {str(k): v for k, v in sweep.items()},
            "single_thread_value": sweep[1]["entities_per_s"],
            "scene_add_s": round(self.t_add, 2),
            "first_cull_s": round(self.t_first_cull, 2),
            "visible": int(visible),
            "describe": self.o.describe(),
        }
        if getattr(self, "port_cs", None) is not None:
            port = {}
            for threads in (1, 8):
                a = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    self.port_cs.cull(self.fr, n_threads=threads, want_ids=False, cap=0)
                    a.append(time.perf_counter() - t0)
                port[str(threads)] = {"median_ms": round(float(np.median(a)) * 1e3, 3), "entities_per_s": n / float(np.median(a))}
            out["port"] = {"note": "oracle/lmx_oracle.c (plain-C restatement; result pages are linked but their bodies not constructed)", "thread_sweep": port}
            self.port_cs = None
        out.update(self._other())
        return out

    def _other(self):
        """the other two metrics of SURVEY.md 8d, bounded samples, one thread and 8 threads (parallel over instances / roots)"""
        o, scenes, other = self.o, self.scenes, {}
        try:
            sk = scenes.skeleton(64, seed=4)
            verts, skin = scenes.skinned_mesh(10_000, 64, seed=6)
            n_inst = 1000  # a tenth of BASELINE config 3's 10 k instances of the 10 k-vertex mesh: 10^7 vertices per frame
            rp, rr = scenes.relative_poses(n_inst, 64, seed=5)
            inv = o.invert_bind(sk["bind"])
            for threads in (1, 8):
                t_pose, t_skin, t_start = [], [], time.time()
                while len(t_skin) < 20 and (time.time() - t_start) < 25.0:  # >= 20 timed frames, bounded
                    t0 = time.perf_counter()
                    apos, arot = o.pose_compute_absolute(rp, rr, sk["parents"], sk["first_nonroot"], n_threads=threads)
                    pal = o.skin_matrices(apos, arot, inv, n_threads=threads)
                    t1 = time.perf_counter()
                    o.evaluate_skin(verts, skin, pal, n_threads=threads)
                    t2 = time.perf_counter()
                    t_pose.append(t1 - t0)
                    t_skin.append(t2 - t1)
                other[f"skin_verts_per_sec_{threads}thread"] = n_inst * len(verts) / float(np.median(t_skin))
                other[f"pose_palette_bones_per_sec_{threads}thread"] = n_inst * 64 / float(np.median(t_pose))
                other[f"skin_frames_timed_{threads}thread"] = len(t_skin)
            h = scenes.hierarchy_chains(250_000, 4, seed=2)  # BASELINE config 3's hierarchy at full size
            nn = len(h["parent"])
            w = o.world(nn)
            roots = np.flatnonzero(h["parent"] < 0).astype(np.int32)
            kids = np.flatnonzero(h["parent"] >= 0).astype(np.int32)
            w.init_transforms(roots, h["local"][roots])
            w.set_parents(h["parent"][kids], kids)
            w.set_local_transforms(kids, h["local"][kids])
            rng_t = np.random.default_rng(1)
            t_x = []
            for _ in range(20):  # 20 frames, every root moved in each: World::setTransform per root = the DFS of world.cpp:255-282
                new_root = scenes.random_transforms(rng_t, len(roots), 4000.0)
                t0 = time.perf_counter()
                w.set_transforms(roots, new_root)
                t_x.append(time.perf_counter() - t0)
            other["transforms_per_sec_1thread"] = len(kids) / float(np.median(t_x))
            other["transform_frames_timed"] = len(t_x)
            other["other_samples"] = (f"skin: {n_inst} instances x 64 bones x {len(verts)} verts of one mesh (a tenth of config 3), median of >= 20 frames at 1 and 8 threads (parallel over instances); "
                                      f"transforms: {len(roots)} roots x depth-4 chains (config 3 at full size), every root moved per frame, median of 20 frames, 1 thread: "
                                      "World is single-writer by design (add / set arrive on the update thread), there is no multi-threaded reference path to time")
        except Exception as e:  # noqa: BLE001 - the headline baseline must survive a problem in the side measurements
            other["other_error"] = repr(e)
        return other


if __name__ == "__main__":
    main()
