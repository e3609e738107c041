#!/usr/bin/env python
"""bench.py — BASELINE.json metric on MI355X: entities culled / s (+ skinned verts / s and the north-star frame as `also`).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

stdout carries ONE compact JSON line (<= 4 KB: the contract keys + `roofline` + `cpu_baseline` + a handful of scalars under `also`),
printed last. Everything else - the roofline legs, the side measurements (tools/bench_extras.py), the CPU thread sweep - goes to
bench_extra.json (repo root and gpurun_out/) and to stderr. The one-rank exchange-path probe only runs with --exchange-probe.

Workload at N=1: BASELINE config 2 ("10M static entities, 1 frustum, 1xMI355X cull + compaction"), sparse variant (cube
[-15000,15000]^3, ~1 M occupied cells), camera = the reference player's default viewport (fov 60 deg, 1920x1080, near 0.1, far 10000,
SURVEY.md 8d). One step = one cull of every resident entity INCLUDING the compaction into one packed, contiguous record
[8 per-type counts | ids] left in HBM (k_cull_tile + k_cull_pack). `value` is an EFFECTIVE rate (the default camera rejects ~95 % of the
tiles by their boxes); `roofline` describes k_cull_tile on the every-sphere-is-fetched-and-tested scene, cache-cold. The ids the timed
camera produces are checked against the reference's sha256 (tests/golden/). For N>1 (one process per GPU) a step is cull + the native
exchange (lmx_exchange_*: one ncclAllGather of [counts | ids] per rank). --scaling weak (default): every rank owns its own 10 M
entities; --scaling strong: BASELINE config 4 - ONE 10 M scene partitioned over the ranks by cell hash.

Timing: W untimed warm-up steps, then repetitions of EXACTLY K steps, each repetition bracketed by barrier + device synchronize on both
sides and reduced MAX over ranks; `ms_per_step` is the median repetition (K = 20 steps of 15 us are 0.3 ms: one repetition alone is
noise), `config.timed_steps` says how many steps were timed in all. value = entities resident on all ranks x frusta / ms_per_step.
Inputs are resident in HBM before the timed region starts; the frustum (256 B) is a kernel argument.
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
LINE_LIMIT = 4096       # bytes of the one stdout line


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


# ---- the device under the bench: torch.cuda for streams / events / buffers on a GPU; nothing at all on the simulated device ----------
class _Buf:
    def __init__(self, keep, ptr):
        self.keep, self.ptr = keep, ptr


class TorchDev:
    """A real GPU: torch.cuda carries the stream, the synchronisation and the few device buffers the bench owns."""
    hostsim = False

    def __init__(self, local_rank):
        import torch

        self.torch = torch
        torch.cuda.set_device(local_rank)

    def use_side_stream(self):
        self.torch.cuda.set_stream(self.torch.cuda.Stream())

    def stream_handle(self):
        return self.torch.cuda.current_stream().cuda_stream

    def sync(self):
        self.torch.cuda.synchronize()

    def marker(self):
        e = self.torch.cuda.Event()
        e.record()
        return e

    def upload(self, a):
        a = np.ascontiguousarray(a)
        t = self.torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()
        return _Buf(t, t.data_ptr())

    def copy_ceiling_gbps(self):
        """GB/s (read + written bytes) of a 1 GiB device-to-device copy of 16-byte elements on this box, now: the achieved-copy ceiling SURVEY.md
        8d asks the roofline fraction to be quoted against besides the 8 TB/s peak. Timed with events on the current stream, 20 repetitions."""
        torch = self.torch
        n = (1 << 30) // 16
        src = torch.empty((n, 4), dtype=torch.float32, device="cuda").fill_(1.0)
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            dst.copy_(src)
        b.record()
        b.synchronize()
        ms = a.elapsed_time(b) / 20.0
        del src, dst
        return 2.0 * (1 << 30) / (ms * 1e-3) / 1e9

    def scrubber(self):
        """evicts the 256 MiB Infinity Cache: a 1 GiB read-only reduction ("read": the cache is left full of CLEAN lines) or a 1 GiB
        read-modify-write ("write": full of DIRTY lines, every later miss first writes a victim back)"""
        scrub = self.torch.zeros(1 << 30, dtype=self.torch.uint8, device="cuda")
        scrub32 = scrub.view(self.torch.int32)
        return lambda kind: scrub.add_(1) if kind == "write" else scrub32.sum()


class HostsimDev:
    """bench.py --selftest-hostsim (tests/test_bench_contract.py): the library is tests/hostsim's CPU build of the same sources, "device"
    memory is host memory and every call is synchronous. Exercises this file's code path end to end; its numbers mean nothing."""
    hostsim = True

    class _Done:
        def synchronize(self):
            pass

    def use_side_stream(self):
        pass

    def stream_handle(self):
        return 0

    def sync(self):
        pass

    def marker(self):
        return self._Done()

    def upload(self, a):
        a = np.ascontiguousarray(a)
        return _Buf(a, a.ctypes.data)

    def scrubber(self):
        return lambda kind: None


def library_is_the_product(path):
    """weak spot closed: LMX_LIB_PATH may point bench.py at the product or at one of its gfx950 build variants, nothing else"""
    p = os.path.realpath(path)
    return p.startswith(os.path.realpath(os.path.join(ROOT, "lumixengine_amd")) + os.sep) or p.startswith(os.path.realpath(os.path.join(ROOT, "tools", "_build", "variants")) + os.sep)  # (tools/build_variant.py: -D builds of the same sources)


def main(argv=None):
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
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements (tools/bench_extras.py)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the two rocprofv3 --pmc child passes; quote the committed profiles/rNN/traffic.json")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skinned-instances", type=int, default=100_000, help="N>1: skinned instances of BASELINE config 4, sharded by index (0 = skip)")
    ap.add_argument("--ranks-share-gpu", action="store_true",
                    help="TEST MODE for the N > 1 code path on a one-GPU box: every rank uses cuda:0, torch.distributed runs on gloo, and the exchange's collective must be carried by LMX_RCCL_LIBRARY = tests/_build/libloopback_rccl.so (RCCL refuses two ranks on one device). Its timings mean nothing")
    ap.add_argument("--big-entities", type=int, default=100_000_000, help="extras: entity count of the config-5-sized single-GPU legs (0 = skip)")
    ap.add_argument("--no-config5-frame", action="store_true", help="N > 1: skip BASELINE config 5's frame (8 cascade frusta in one pass + ONE collective, lmx_exchange_cull_many)")
    ap.add_argument("--exchange-probe", action="store_true", help="after everything else, besides the default one-rank exchange step: the same with the side-stream gather -> bench_extra.json")
    ap.add_argument("--no-exchange-step", action="store_true", help="skip the N > 1 step with a world of ONE rank (a child process; `also.exchange_step_one_rank_ms`)")
    ap.add_argument("--min-timed-steps", type=int, default=200, help="repeat the K-step timed region until at least this many steps were timed")
    ap.add_argument("--selftest-hostsim", action="store_true", help="TEST MODE (CPU, tests/test_bench_contract.py): run this file's code path against tests/hostsim's build of the library with tiny scenes")
    args = ap.parse_args(argv)

    from lumixengine_amd import api, scenes
    from lumixengine_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            return self_launch(args.gpus, sys.argv[1:] if argv is None else list(argv))  # plain `python bench.py --gpus N`: this process becomes the launcher
        args.gpus = world
    selftest = args.selftest_hostsim
    if selftest:
        if os.environ.get("LMX_HOSTSIM") != "1" or world != 1:
            raise SystemExit("--selftest-hostsim needs LMX_HOSTSIM=1 and LMX_LIB_PATH = tests/hostsim's library, one rank")
        args.no_live_traffic = True
        args.big_entities = 0
        args.entities = min(args.entities, 30_000)
    elif not library_is_the_product(api.LIB_PATH):
        raise SystemExit(f"bench.py refuses to time {api.LIB_PATH}: LMX_LIB_PATH must stay inside lumixengine_amd/ or tools/_build/variants/")
    if args.ranks_share_gpu:
        if not os.environ.get("LMX_RCCL_LIBRARY"):
            raise SystemExit("--ranks-share-gpu needs LMX_RCCL_LIBRARY (tests/cpp/loopback_rccl.cpp): RCCL refuses two ranks on one device")
        local_rank = 0
    red_dev = "cpu" if args.ranks_share_gpu else "cuda"  # where torch.distributed's few scalars live (gloo in the test mode)
    dev = HostsimDev() if selftest else TorchDev(local_rank)
    use_dist = world > 1 or args.force_collective
    dist = torch = None
    if use_dist:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL prints a version banner on the C-level stdout at communicator creation; stdout must carry the JSON line only
        with c_stdout_to_stderr():
            if args.ranks_share_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            dist.barrier()
            dev.sync()
        # N > 1: launches go to a stream of their own, not the legacy default stream. Every event record / stream wait that involves
        # the default stream makes the runtime look at all other blocking streams (measured with a world of one rank: 41 us of host
        # time per step on the default stream - 15.6 us for ONE record + wait pair).
        dev.use_side_stream()
    ctx = api.Context(local_rank)
    ctx.set_stream(dev.stream_handle())  # launches, the markers of `timed` and dev.sync() share one stream

    def barrier():
        if use_dist:
            dist.barrier()

    def reduce_int(v, op):
        t = torch.tensor([int(v)], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t, op=op)
        return int(t.item())

    def timed(fn, steps):
        """K steps bracketed by barrier + synchronize; ms per step, MAX over ranks. The host may run at most ~48 launches ahead of the
        GPU (a marker every 16 steps, wait on the one recorded 32-48 steps earlier): the HIP runtime drains the whole queue once every
        several hundred launches, which shows up as one stall of backlog x kernel time when a loop is GPU-paced and unbounded
        (tools/hip_queue_stall_probe.py)."""
        barrier()
        dev.sync()
        marks = []
        t0 = time.perf_counter()
        for i in range(steps):
            fn()
            if (i & 15) == 15:
                marks.append(dev.marker())
                if len(marks) > 2:
                    marks.pop(0).synchronize()
        dev.sync()
        barrier()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        if use_dist:
            t = torch.tensor([ms], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms = float(t.item())
        return ms

    def timed_repeated(fn, steps):
        """repetitions of exactly `steps` steps until --min-timed-steps were timed: (median ms per step, all repetitions)"""
        reps = max(1, -(-args.min_timed_steps // max(steps, 1)))
        if selftest:
            reps = 2
        all_ms = [timed(fn, steps) for _ in range(reps)]
        return float(np.median(all_ms)), all_ms

    # ---- the headline workload -----------------------------------------------------------------------------
    N = args.entities
    half = 15000.0 if args.variant == "sparse" else 5000.0
    if selftest:
        half = 2000.0
    t0 = time.time()
    strong = args.scaling == "strong" and use_dist
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
        baseline = CpuBaseline(scenes, sc, frustum, log, quick=selftest)  # builds the CPU-side scene on a background thread while the GPU legs run
    # Process warm-up, outside every timed region: the HIP runtime pays a one-time ~50 ms stall around the 850th kernel launch of a
    # process (tools/hip_queue_stall_probe.py). Without this it lands in whichever timed loop crosses that count.
    for _ in range(3 if selftest else 1200):
        cs.cull(frustum)
    dev.sync()
    if args.camera == "all_visible":
        frustum = api.viewport_frustum(pos=(0.0, 0.0, 4.0 * half), far=20.0 * half)
    n_frusta = 1
    import ctypes as C

    fr_c = np.ascontiguousarray(frustum, api.SHIFTED_FRUSTUM).reshape(-1)
    fr_ptr, lib, h = api._ptr(fr_c), ctx.lib, ctx.h
    xchg = None
    if use_dist:
        # One exchange per frame, native (csrc/lmx_capi_exchange.hip): the cull's gather kernels write [8 counts | cap ids] into the
        # send buffer, ONE ncclAllGather per frame. torch.distributed only carries the 128-byte RCCL id, the barrier and the timing
        # reduction.
        uid = torch.zeros(128, dtype=torch.uint8, device=red_dev)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(api.exchange_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0)
        cap = (reduce_int(int(cs.cull(frustum).counts()[0].sum()), dist.ReduceOp.MAX) * 5 // 4 + 1023) // 1024 * 1024  # 1.25 x the largest visible count of any rank
        log(f"[rank {rank}] exchange: creating the RCCL communicator (ids per rank {cap})")
        with c_stdout_to_stderr():
            xchg = api.VisibleExchange(ctx, rank, world, uid.cpu().numpy().tobytes(), cap)
            xchg.wait(xchg.cull(frustum))  # the first collective creates RCCL's channels (and may print)
        exchange_info = xchg.info()  # which of the three step forms this rank's exchange took, and the gather time it took it on
        log(f"[rank {rank}] exchange ready: {exchange_info}")
        x_cull, xh, slot_c = ctx.lib.lmx_exchange_cull, xchg.h, C.c_uint32(0)

        def step():
            if x_cull(xh, fr_ptr, api.TYPE_ALL, C.byref(slot_c)) != 0:
                raise RuntimeError(ctx.lib.lmx_last_error(ctx.h).decode())
            return slot_c.value
    else:
        rec_p, rec_n = C.c_void_p(), C.c_uint32(0)

        def step():  # one cull incl. compaction: k_cull_tile, then k_cull_pack gathers the shard windows into [8 counts | ids] in HBM
            if lib.lmx_cull(h, 0, fr_ptr, 1, api.TYPE_ALL) != 0 or lib.lmx_cull_pack_device(h, 0, 0, C.byref(rec_p), C.byref(rec_n)) != 0:
                raise RuntimeError(lib.lmx_last_error(h).decode())

    for _ in range(args.warmup):
        step()
    gather_probe = None
    if use_dist and exchange_info["mode"] != "p2p":
        # one all-gather of the record THIS step ships, timed over 32 gathers (a collective: every rank is here), outside the timed region
        gather_probe = xchg.timeGather(1)
        for _ in range(max(2, min(args.warmup, 5))):
            step()
    ms_per_step, reps_ms = timed_repeated(step, args.steps)  # (the closing synchronize of `timed` also drains the side stream's last gathers)
    value = (N if strong else N * world) * n_frusta / (ms_per_step * 1e-3)
    ms_cull_only = ms_host_list = None
    if not use_dist:
        ms_cull_only = timed(lambda: cs.cull(frustum), max(args.steps, 2 if selftest else args.min_timed_steps))  # rounds 1 / 2's step: the cull kernel alone
        # SURVEY.md 8d (i), second form: the list on the HOST - what the CullingSystem adapter does per view: cull + lmx_cull_map_all
        # (packed record copied into the library's pinned buffer; ONE host wait), PCIe included. Never `value`.
        ids_p, cnt8 = C.POINTER(C.c_int32)(), (C.c_uint32 * 8)()

        def step_host():
            if lib.lmx_cull(h, 0, fr_ptr, 1, api.TYPE_ALL) != 0 or lib.lmx_cull_map_all(h, 0, 0, C.byref(ids_p), cnt8) != 0:
                raise RuntimeError(lib.lmx_last_error(h).decode())

        for _ in range(5):
            step_host()
        ms_host_list = timed(step_host, 4 if selftest else 100)
    res = cs.cull(frustum)
    visible = int(res.counts()[0].sum())
    # identity of what was timed: the ids, not just their number, against the reference's CullingSystemImpl (same seeded scene)
    at_headline_size = args.variant == "sparse" and N == 10_000_000 and args.camera == "default" and not selftest
    ids_checked = "unchecked"
    if at_headline_size and not (strong and use_dist) and rank == 0:
        ids_checked = check_ids(res, "sparse_10m")  # (strong: the union over ranks is compared below)
    log(f"[rank {rank}] headline: {ms_per_step * 1e3:.2f} us per step (median of {len(reps_ms)} x {args.steps} steps: "
        f"{', '.join(f'{m * 1e3:.2f}' for m in reps_ms[:12])}), {visible} visible, ids {ids_checked}")

    dist_info, headline_exchange = {}, {}
    if use_dist:
        st = xchg.stats(step())  # one more frame of the timed kind: what it ships, what of that is counts + ids
        headline_exchange = {
            "exchange_bytes_shipped_per_rank": st["bytes_shipped_per_peer"],  # towards EACH peer: the fixed-size record (collective forms) or its used part (P2P)
            "exchange_bytes_used_per_rank": st["bytes_used"],                 # counts + the ids this rank has
            "exchange_bytes_arriving_per_rank": st["bytes_shipped_per_peer"] * (world - 1),
            "exchange_overflow_mask": st["overflow_mask"]}
        dist_info, ids_checked = exchange_checks_and_frames(args, api, scenes, D, dev, ctx, cs, sc, xchg, step, frustum, fr_ptr, timed, reduce_int, rank, world, local_rank,
                                                            red_dev, strong, half, visible, at_headline_size, ids_checked, torch, dist)

    # ---- roofline of the dominant kernel (k_cull_tile): a HIP event pair per launch, on the launch stream, filled by the launch itself
    # (hipExtLaunchKernelGGL: the dispatch's own begin / end timestamps - what rocprofv3's kernel trace reports) --------------------
    # Three regimes of the same kernel on the same 10 M geometry:
    #   default camera   hierarchical skip: ~95 % of the tiles end at the tile-level box test. Latency regime; the algorithmic bytes
    #                    (20 B per RESIDENT entity) are mostly never moved: an effective rate only.
    #   all_accept       a camera that sees the whole cube: every tile is TILE_ACCEPT, ids are copied: 4 B read + 4 B written per entity.
    #   all_test         the same positions with radii in (300, 330]: every cell is a "big" cell (culling_system.cpp:140,342-344) and
    #                    skips the AABB pre-test, so every sphere is fetched and tested: 16 B + 4 B read per entity + 4 B per visible
    #                    id. Here moved bytes == algorithmic bytes: this leg, cache-cold, is `roofline.frac`.
    scrub = dev.scrubber()
    kreps = (lambda n: 2) if selftest else (lambda n: n)

    def kernel_times(csys, fr, steps, cold, view=0):
        """avg device ms of k_cull_tile over `steps` culls; cold: evict the Infinity Cache before every cull (True / "read" or "write")"""
        ctx.profile_reset()
        ctx.profile_enable(True)
        for _ in range(steps):
            if cold:
                scrub("write" if cold == "write" else "read")
            csys.cull(fr, view=view)
        ctx.synchronize()
        ctx.profile_enable(False)
        ms_s, n_s = ctx.profile_get(api.K_CULL_SPHERES)
        return ms_s / max(n_s, 1)

    gbps = lambda b, ms: None if (ms != ms or not ms) else round(b / (ms * 1e-3) / 1e9, 1)  # noqa: E731
    rnd = lambda x, n: None if x != x else round(x, n)  # noqa: E731
    frac = lambda b, ms: None if (ms != ms or not ms) else round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)  # noqa: E731
    avg_default_ms = kernel_times(cs, frustum, kreps(max(args.steps, 100)), cold=False)
    alg_bytes = 20.0 * N + 4.0 * visible
    legs = {"default_camera": {
        "visible": visible, "warm_avg_launch_ms": rnd(avg_default_ms, 5), "algorithmic_bytes": alg_bytes, "effective_GBps": gbps(alg_bytes, avg_default_ms),
        "note": "effective rate: 20 B per resident entity although ~95 % of the tiles are rejected by their box and never fetched; not a roofline fraction"}}
    nan = float("nan")
    test_cold_ms = test_warm_ms = nan
    test_bytes = 0.0
    if not args.headline_only:
        legs["default_camera"]["cold_avg_launch_ms"] = rnd(kernel_times(cs, frustum, kreps(50), cold=True), 5)
        accept_fr = api.viewport_frustum(pos=(0.0, 0.0, 4.0 * half), far=20.0 * half)
        accept_visible = int(cs.cull(accept_fr).counts()[0].sum())
        acc_warm = kernel_times(cs, accept_fr, kreps(50), cold=False)
        acc_cold = kernel_times(cs, accept_fr, kreps(50), cold=True)
        acc_bytes = 8.0 * N  # 4 B id read + 4 B id written per entity; cell keys and spheres are not touched
        legs["all_accept"] = {"visible": accept_visible, "moved_bytes": acc_bytes, "warm_avg_launch_ms": rnd(acc_warm, 5), "cold_avg_launch_ms": rnd(acc_cold, 5),
                              "warm_GBps": gbps(acc_bytes, acc_warm), "cold_GBps": gbps(acc_bytes, acc_cold), "warm_frac": frac(acc_bytes, acc_warm), "cold_frac": frac(acc_bytes, acc_cold)}
        # all_test: same positions, every sphere "big" -> every cell CELL_TEST
        sc_t = dict(sc)
        sc_t["radius"] = scenes.all_test_radii(N)
        cs_t = api.CullingSystem(ctx)
        cs_t.build(sc_t["entity"], sc_t["type"], sc_t["pos"], sc_t["radius"])
        for _ in range(kreps(20)):
            cs_t.cull(frustum)
        res_t = cs_t.cull(frustum)
        test_visible = int(res_t.counts()[0].sum())
        test_ids_checked = check_ids(res_t, "all_test_10m") if (at_headline_size and rank == 0) else "unchecked"
        test_bytes = 20.0 * N + 4.0 * test_visible
        test_warm_ms = kernel_times(cs_t, frustum, kreps(50), cold=False)
        test_cold_ms = kernel_times(cs_t, frustum, kreps(100), cold=True)
        test_coldw_ms = kernel_times(cs_t, frustum, kreps(30), cold="write")
        legs["all_test"] = {"visible": test_visible, "moved_bytes": test_bytes, "warm_avg_launch_ms": rnd(test_warm_ms, 5), "cold_avg_launch_ms": rnd(test_cold_ms, 5),
                            "warm_GBps": gbps(test_bytes, test_warm_ms), "cold_GBps": gbps(test_bytes, test_cold_ms), "warm_frac": frac(test_bytes, test_warm_ms),
                            "cold_frac": frac(test_bytes, test_cold_ms), "cold_after_dirty_scrub_avg_launch_ms": rnd(test_coldw_ms, 5),
                            "cold_after_dirty_scrub_frac": frac(test_bytes, test_coldw_ms), "cells": cs_t.stats()["cells"], "visible_ids": test_ids_checked,
                            "note": "warm = back-to-back frames (the 200 MB working set stays in the 256 MiB Infinity Cache); cold = after a 1 GiB read-only scrub; the 100 M extra (config5_size_single_gpu.all_test) is HBM-cold by size"}
        # the multi-frustum kernel on the same every-sphere-is-tested scene: the frame's 8 shadow-cascade frusta in ONE pass over the
        # spheres (pass width 8, config 5's form). SURVEY.md 8d asks for both fractions here (8 x 56 flop per 20 B sits at the fp32 ridge).
        fr8_t = np.concatenate([api.viewport_frustum(**kw) for kw in scenes.config5_cascade_kwargs()])
        cs_t.setPassWidth(8)
        try:
            for _ in range(kreps(5)):
                cs_t.cull(fr8_t, view=1)
            v8_t = cs_t.cull(fr8_t, view=1).counts().sum(axis=1)
            ms8 = kernel_times(cs_t, fr8_t, kreps(20), cold=True, view=1)
            bytes8 = 20.0 * N + 4.0 * float(v8_t.sum())
            legs["all_test_8_frusta_one_pass"] = {
                "visible_per_frustum": [int(x) for x in v8_t], "algorithmic_bytes": bytes8, "cold_avg_launch_ms": rnd(ms8, 5), "cold_GBps": gbps(bytes8, ms8),
                "hbm_frac": frac(bytes8, ms8), "flops": 56.0 * N * 8, "valu_frac_of_157_TFLOPs": None if (ms8 != ms8 or not ms8) else round(56.0 * N * 8 / (ms8 * 1e-3) / 157.3e12, 4),
                "entity_frustum_tests_per_sec": None if (ms8 != ms8 or not ms8) else 8.0 * N / (ms8 * 1e-3),
                "note": "k_cull_tile<F = 0> (runtime frustum count), pass width 8, cache-cold after a read-only 1 GiB scrub; 56 flop per sphere and frustum (SURVEY.md 8d)"}
        finally:
            cs_t.setPassWidth(0)
        del cs_t, sc_t
        # the same regime reached the config-2-faithful way: NORMAL radii, the cells classified by the AABB pre-tests, under an
        # orthographic slab camera that every cell of a one-layer scene straddles (scenes.slab_scene / slab_frustum_kwargs)
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
            slab_warm = kernel_times(cs_b, fr_b, 50, cold=False)
            slab_cold = kernel_times(cs_b, fr_b, 50, cold=True)
            legs["all_cell_test_normal_radii"] = {
                "visible": slab_visible, "moved_bytes": slab_bytes, "warm_avg_launch_ms": rnd(slab_warm, 5), "cold_avg_launch_ms": rnd(slab_cold, 5),
                "warm_GBps": gbps(slab_bytes, slab_warm), "cold_GBps": gbps(slab_bytes, slab_cold), "warm_frac": frac(slab_bytes, slab_warm), "cold_frac": frac(slab_bytes, slab_cold),
                "cells": cs_b.stats()["cells"], "visible_ids": slab_checked,
                "note": "scenes.slab_scene: normal radii, one layer of cells, every cell straddles the ortho slab camera's near and far plane -> CELL_TEST through containsAABB / intersectsAABB"}
            del cs_b, sc_b
    del scrub
    traffic, traffic_note = (None, None)
    if rank == 0 and world == 1 and not args.headline_only and not args.no_live_traffic:
        traffic, traffic_note = measure_traffic_live(log)
    if traffic is None:
        traffic, traffic_note = load_traffic("k_cull_tile:all_test")
    have_test = test_cold_ms == test_cold_ms and test_cold_ms > 0
    copy_gbps = None  # the device-to-device copy this box reaches right now: 1 GiB of float4, 20 repetitions, read + written bytes
    if rank == 0 and not selftest and not args.headline_only:
        try:
            copy_gbps = dev.copy_ceiling_gbps()
            log(f"[copy ceiling] 1 GiB device-to-device copy, 20 repetitions: {copy_gbps:.0f} GB/s (read + written bytes)")
        except Exception as e:  # noqa: BLE001 - a side measurement
            log(f"[copy ceiling] not measured: {e!r}")
    roof_ms = test_cold_ms if have_test else avg_default_ms
    roof_bytes = test_bytes if have_test else alg_bytes
    roof_gbps = roof_bytes / (roof_ms * 1e-3) / 1e9 if roof_ms else 0.0
    roofline = {
        "kernel": "k_cull_tile",
        "bound": "hbm",
        "leg": f"NOT the timed step: a separate launch over the all_test scene ({N} entities, every sphere fetched and tested: 20 B/entity + 4 B/visible id), cache-cold after a read-only 1 GiB scrub" if have_test
               else "NOT the timed step and NOT a roofline fraction: default camera only (--headline-only), effective rate",
        "achieved": round(roof_gbps, 1),
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": round(roof_gbps / HBM_PEAK_GBPS, 4),
        "traffic": traffic,  # HBM bytes per launch of the all_test leg: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of THIS run (rank 0, N = 1); the committed profiles/rNN/traffic.json only when rocprofv3 is unavailable (traffic_source says which)
        "traffic_source": traffic_note,
        "algorithmic_bytes_per_launch": roof_bytes,
        "avg_launch_ms": round(roof_ms, 5),
        "avg_launch_ms_is": "mean over the leg's launches of hipEventElapsedTime on the event pair the launch itself fills (hipExtLaunchKernelGGL start / stop events on the launch stream): the dispatch's begin -> end, as in rocprofv3's kernel trace",
        "measured_copy_GBps": round(copy_gbps, 1) if copy_gbps else None,  # measured in THIS run (not the guide's figure): 1 GiB float4 copy, read + written bytes
        "frac_of_measured_copy": round(roof_gbps / copy_gbps, 4) if copy_gbps else None,
        # the same kernel's weaker relatives, in the record: NORMAL radii, every cell classified CELL_TEST by the AABB pre-tests
        # (slab scene, cache-cold; its algorithmic bytes include 4 B per visible id, 43 % visible), and the all-test launch behind a DIRTY scrub -
        # the state the cache is in behind 12 GB of skinning stores
        "frac_normal_radii": legs.get("all_cell_test_normal_radii", {}).get("cold_frac"),
        "frac_dirty_cache": legs.get("all_test", {}).get("cold_after_dirty_scrub_frac"),
        "read_ceiling_note": "a pure read of the same 20 B per entity in the kernel's access pattern reaches 0.80 of the 8 TB/s peak at 10 M entities and 0.86 at 100 M (tools/read_probe.hip, profiles/r06/read_probe_cull_footprint.txt)",
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
        "data": "synthetic" if not selftest else "SELFTEST on the simulated device: not a measurement",
        "config": {
            "workload": f"BASELINE config 2: {N} static entities per GPU ({args.variant}, cube +-{half:g}), 1 frustum (fov 60, 16:9, near 0.1, far 10000), cull + compaction",
            "entities_per_gpu": N,
            "frusta": n_frusta,
            "cells_per_gpu": stats["cells"],
            "visible_per_gpu": visible,
            "sharding": (("one scene partitioned by cell hash" if strong else "own entities per rank") + ", native RCCL all-gather of visible ids") if use_dist else "single GPU",
            "timed_steps": args.steps * len(reps_ms),
            "repetitions": len(reps_ms),
            "ms_per_step_is": "median over the repetitions of (K steps between barrier + synchronize, MAX over ranks) / K",
            "ms_per_step_min_max": [min(reps_ms), max(reps_ms)],
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
    if have_test:
        result["value_streaming"] = N / (test_cold_ms * 1e-3)  # every sphere fetched and tested, cache-cold: what roofline.frac is the fraction of
    result["config"]["visible_ids"] = ids_checked  # 'reference': sha256 of the sorted ids == the reference CullingSystemImpl's on the same seeded scene
    result["config"].update(dist_info)
    if use_dist:
        result["config"]["exchange_mode"] = exchange_info["mode"]
        result["config"]["exchange_gather_us_first_frame"] = exchange_info["gather_us"]  # what `auto` decided on: the first frame's record
        result["config"]["exchange_mode_why"] = exchange_info["why"]
        if gather_probe is not None:
            result["config"]["exchange_gather_us"] = round(gather_probe[0], 2)  # one gather of the record the timed step ships, 32 back to back on the side stream
            result["config"]["exchange_gather_record_bytes"] = 4 * gather_probe[1]
        result["config"].update(headline_exchange)
        result["config"]["xgmi_curve"] = "ONE point; before this run NO 1/2/4/8 xGMI curve existed (the builder's boxes have one GPU): the driver computes scaling from its own runs"
    if args.ranks_share_gpu:
        result["config"]["TEST_MODE"] = "--ranks-share-gpu: all ranks on cuda:0, gloo + shared-memory collective (tests/cpp/loopback_rccl.cpp); timings are meaningless"
    if rank == 0 and world == 1:
        if not args.no_extras and not args.headline_only:
            result["extra"] = {}
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_extras

                bench_extras.extras(ctx, api, scenes, dev, timed, N, log, check_ids, args.big_entities, out=result["extra"], small=selftest)
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
        if not args.no_exchange_step and not args.headline_only and not selftest:
            # LAST: every number above is taken; a child process with a context of its own and a hard timeout. The like-for-like N = 1 point
            # of the scaling curve: the step `--gpus N` times (cull + pack into the send buffer + ONE ncclAllGather), with a world of one rank
            result.setdefault("extra", {})
            dev.sync()
            result["extra"]["exchange_path_one_rank"] = exchange_path_one_rank(args, log, timeout_s=120.0)
            if args.exchange_probe:
                side = exchange_path_one_rank(args, log, extra_env={"LMX_EXCHANGE_INLINE": "0"})
                side.pop("what", None)
                result["extra"]["exchange_path_one_rank"]["side_stream_gather"] = side
    if rank == 0:
        emit(result, log)
    if use_dist:
        dist.destroy_process_group()
    return 0


def exchange_checks_and_frames(args, api, scenes, D, dev, ctx, cs, sc, xchg, step, frustum, fr_ptr, timed, reduce_int, rank, world, local_rank, red_dev, strong, half,
                               visible, at_headline_size, ids_checked, torch, dist):
    """N > 1 (or --force-collective): sanity of the exchange step, then the frames that ride along - BASELINE config 4's (cell shard +
    exchange + skinning by index) and config 5's (8 cascades, one collective). RULE for everything here: every rank runs the same
    sequence of collectives whatever happens locally. Local failures are carried as a flag and agreed on (all_reduce MIN) before the
    next collective phase; nothing is caught between two collectives."""
    import ctypes as C

    N = args.entities
    info = {}
    slot = step()
    xchg.wait(slot)
    dev.sync()
    parsed, seen = [], []
    for r in range(world):
        counts, ids = xchg.read(slot, r)
        seen.append(int(counts.sum()))
        parsed.append(ids)
    info["allgather_visible_counts"] = seen
    how = {"inline": "one ncclAllGather per frame, in place, on the cull stream behind the pack kernel", "side": "one ncclAllGather per frame, in place, on a side stream (double-buffered: the next cull overlaps it)",
           "p2p": "no collective: the used part of the record stored into every peer through hipIpc mappings + sequence flags"}[xchg.info()["mode"]]
    info["exchange"] = f"[8 counts | {xchg.cap} ids] per rank; {how} (lmx_exchange_*; the mode is lmx_exchange_info's, not an environment guess)"
    info["ranks_seen_by_rccl"] = len(seen)
    assert len(seen) == world and max(seen) <= xchg.cap, (seen, xchg.cap)
    local = cs.cull(frustum, view=2)  # (view 0 / 1 hold exchange frames)
    assert seen[rank] == visible and np.array_equal(np.sort(parsed[rank]), np.sort(local.all_ids(0)[0])), "gathered ids differ from the local cull result"
    if strong and rank == 0:
        ctx_whole = api.Context(local_rank)  # a context of its own: a context holds ONE culling set, and this rank's shard stays resident
        whole = api.CullingSystem(ctx_whole)
        whole.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
        res_whole = whole.cull(frustum)
        want = np.sort(res_whole.all_ids(0)[0])
        assert np.array_equal(np.sort(np.concatenate(parsed)), want), "union of the ranks' lists != unsharded cull"
        if at_headline_size:
            ids_checked = check_ids(res_whole, "sparse_10m")
        info["union_equals_unsharded"] = True
        info["visible_total"] = int(len(want))
        del whole
        ctx_whole.close()
    log(f"[rank {rank}] exchange verified: {info}")

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
        d_pos_sk, d_rot_sk = dev.upload(pos_sk), dev.upload(rot_sk)
        del pos_sk, rot_sk
        sk.setPoseSourceDevice(d_pos_sk.ptr, d_rot_sk.ptr, len(mine_i) * 64)

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
        info["config4_frame"] = config4_frame(ctx, step, "")
    elif not strong and args.skinned_instances and not args.no_extras and not args.headline_only:
        # weak run (the driver's SCALE line): the strong-scaling frame of BASELINE config 4 rides along, in a context and a communicator
        # of its own, so one `--gpus N` run yields both curves.
        ok, err, ctx4, cs4, count4 = 1, None, None, None, 0
        try:  # ---- local phase: may fail on this rank only (memory for a second 10 M-entity context, ...) ----
            sc4 = sc if rank == 0 else scenes.cull_scene(N, half, seed=2)  # (rank 0's weak scene IS seed 2)
            mine4 = D.shard_by_cell(sc4["pos"], world, rank)
            ctx4 = api.Context(local_rank)
            ctx4.set_stream(dev.stream_handle())
            cs4 = api.CullingSystem(ctx4)
            cs4.build(sc4["entity"][mine4], sc4["type"][mine4], sc4["pos"][mine4], sc4["radius"][mine4])
            count4 = int(cs4.cull(frustum).counts()[0].sum())
            uid_bytes = api.exchange_unique_id() if rank == 0 else None
        except Exception as e:  # noqa: BLE001 - carried to the agreement below, never raised between collectives
            ok, err, uid_bytes = 0, repr(e), None
        # ---- collective phase 1: the same four collectives on every rank ----
        uid4 = torch.zeros(128, dtype=torch.uint8, device=red_dev)
        if rank == 0 and uid_bytes is not None:
            uid4.copy_(torch.frombuffer(bytearray(uid_bytes), dtype=torch.uint8))
        dist.broadcast(uid4, 0)
        most4 = reduce_int(count4, dist.ReduceOp.MAX)
        total4 = reduce_int(count4, dist.ReduceOp.SUM)
        all_ok = reduce_int(ok, dist.ReduceOp.MIN)
        if all_ok == 1:
            # ---- collective phase 2: communicator + frames; a failure in here is fatal for the run (raised, not caught) ----
            cap4 = (most4 * 5 // 4 + 1023) // 1024 * 1024
            with c_stdout_to_stderr():
                xchg4 = api.VisibleExchange(ctx4, rank, world, uid4.cpu().numpy().tobytes(), cap4)
                xchg4.wait(xchg4.cull(frustum))
            xh4, slot4, x_cull = xchg4.h, C.c_uint32(0), ctx4.lib.lmx_exchange_cull

            def step4():
                if x_cull(xh4, fr_ptr, api.TYPE_ALL, C.byref(slot4)) != 0:
                    raise RuntimeError(ctx4.lib.lmx_last_error(ctx4.h).decode())

            c4 = config4_frame(ctx4, step4, "; measured in the weak run as an extra")
            c4["visible_total"] = total4
            if at_headline_size:
                try:
                    want4 = sum(json.load(open(os.path.join(ROOT, "tests", "golden", "cull_bench_scenes.json")))["scenes"]["sparse_10m"]["cameras"]["default"]["counts"])
                except Exception:  # noqa: BLE001
                    want4 = None
                if want4 is not None:
                    assert total4 == want4, f"config 4: the ranks' shards see {total4} ids, the reference {want4}"
                    c4["visible_total_is"] = "the reference's count (tests/golden/cull_bench_scenes.json)"
            info["config4_frame"] = c4
            xchg4.close()
        else:
            info["config4_frame"] = {"error": err or "another rank failed in its local phase"}
            log(f"[rank {rank}] config 4 extra skipped on every rank: {info['config4_frame']}")
        if cs4 is not None:
            del cs4
        if ctx4 is not None:
            ctx4.close()
    if not args.no_config5_frame and not strong and not args.headline_only:
        # BASELINE config 5 across GPUs: every rank's entities under the frame's 8 cascade frusta in one pass over the spheres and ONE
        # all-gather of 8 sub-records (pipeline.cpp:1252-1258 culls them one by one); distributed.config5_frame keeps the collective rule.
        class TorchColl:  # the scalars the ranks agree on travel over torch.distributed (gloo in the --ranks-share-gpu test mode)
            max_int = staticmethod(lambda v: reduce_int(v, dist.ReduceOp.MAX))
            min_int = staticmethod(lambda v: reduce_int(v, dist.ReduceOp.MIN))

            @staticmethod
            def bcast_bytes(b):
                u = torch.zeros(128, dtype=torch.uint8, device=red_dev)
                if b is not None:
                    u.copy_(torch.frombuffer(bytearray(b), dtype=torch.uint8))
                dist.broadcast(u, 0)
                return u.cpu().numpy().tobytes()

        fr8 = np.concatenate([api.viewport_frustum(**kw) for kw in scenes.config5_cascade_kwargs()])
        c5 = D.config5_frame(api, fr8, ctx, cs, rank, world, N, TorchColl, timed, quiet=c_stdout_to_stderr)
        info["config5_frame"] = c5
        log(f"[rank {rank}] config 5 frame: {c5}")
    xchg.close()
    log(f"[rank {rank}] exchange closed")
    return info, ids_checked


# ---- the one stdout line -------------------------------------------------------------------------------------------------------
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "measured_copy_GBps", "frac_of_measured_copy", "frac_normal_radii",
                 "frac_dirty_cache", "leg")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "host_cores", "single_thread_value", "O3_avx2_port_value", "error")
CONFIG_KEYS = ("workload", "entities_per_gpu", "frusta", "visible_per_gpu", "visible_ids", "sharding", "timed_steps", "repetitions", "exchange_mode", "exchange_gather_us", "exchange_gather_record_bytes",
               "exchange_bytes_shipped_per_rank", "exchange_bytes_used_per_rank", "exchange_bytes_arriving_per_rank", "exchange_overflow_mask", "ranks_seen_by_rccl",
               "allgather_visible_counts", "union_equals_unsharded", "visible_total", "xgmi_curve", "TEST_MODE")


def _short(v, n=200):
    return v if not isinstance(v, str) or len(v) <= n else v[: n - 3] + "..."


def compact_line(result):
    """The record the driver parses: the contract keys, `config` (workload + a few scalars), `roofline`, `cpu_baseline` and `also` - a
    dozen scalars of the other two north-star components (skinned verts / s, the 10 M + 100 k frame). <= LINE_LIMIT bytes, always."""
    line = {k: result[k] for k in CONTRACT_KEYS}
    cfg = result.get("config", {})
    line["config"] = {k: _short(cfg[k]) for k in CONFIG_KEYS if k in cfg}
    for frame in ("config4_frame", "config5_frame"):
        f = cfg.get(frame)
        if isinstance(f, dict):
            line["config"][frame] = {k: f[k] for k in ("ms_per_frame_max_over_ranks", "frames_per_sec", "skinned_verts_per_sec_all_ranks", "entity_frustum_tests_per_sec_all_ranks", "error") if k in f}
            if isinstance(f.get("exchange"), dict):  # config 5's frame: what its 8-sub-record exchange ships
                line["config"][frame]["exchange"] = {k: f["exchange"][k] for k in ("mode", "bytes_shipped_per_peer", "bytes_used_this_rank", "shipped_over_used", "gather_us_of_this_record", "overflow_mask") if k in f["exchange"]}
    line["roofline"] = {k: _short(result["roofline"].get(k), 160) for k in ROOFLINE_KEYS}
    if "cpu_baseline" in result:
        line["cpu_baseline"] = {k: _short(result["cpu_baseline"][k], 340) for k in CPU_KEYS if k in result["cpu_baseline"]}
    also = {}
    for k in ("value_cull_only", "value_incl_host_readback", "value_streaming"):
        if k in result:
            also[k] = result[k]
    ex = result.get("extra", {})
    for k in ("skinned_verts_per_sec", "transforms_per_sec", "target_frames_per_sec_1gpu", "target_skinned_verts_per_sec", "target_skin_ms_per_1e9_verts", "config3_frames_per_sec",
              "keys_kernels_ms", "xform_level_kernel_avg_ms", "xform_with_moved_list_ms", "pose_palette_kernel_avg_ms", "transform_ms_per_frame",
              "skin_distinct_meshes_instances", "skin_distinct_meshes_verts_per_sec", "skin_distinct_meshes_positions", "config5_frame_1gpu_ms", "config5_frame_1gpu_is"):
        if k in ex:
            also[k] = _short(ex[k], 150)
    leg8 = result.get("roofline", {}).get("legs", {}).get("all_test_8_frusta_one_pass")
    if isinstance(leg8, dict) and leg8.get("cold_avg_launch_ms"):
        also["cull8_all_test_kernel_ms"] = leg8["cold_avg_launch_ms"]  # config 5's pass: 8 cascade frusta x 10 M spheres in ONE launch, every sphere tested, cache-cold
    if isinstance(ex.get("exchange_path_one_rank"), dict) and "ms_per_step" in ex["exchange_path_one_rank"]:
        also["exchange_step_one_rank_ms"] = ex["exchange_path_one_rank"]["ms_per_step"]
        also["exchange_mode"] = ex["exchange_path_one_rank"].get("exchange_mode")
        also["exchange_step_note"] = "N>1 step (cull+pack+ONE ncclAllGather) with a world of ONE rank; no xGMI curve measured by the builder"
    if isinstance(ex.get("target_character_mesh"), dict):
        also["target_character_mesh_skin_ms_per_1e9_verts"] = ex["target_character_mesh"].get("skin_ms_per_1e9_verts")
    if "error" in ex:
        also["extras_error"] = _short(ex["error"], 160)
    also["details"] = "bench_extra.json"
    line["also"] = also
    s = json.dumps(line, separators=(",", ":"))
    for drop in ("also", "cpu_baseline.sample", "roofline.leg", "config"):  # never expected; the limit holds whatever a leg returns
        if len(s) <= LINE_LIMIT:
            break
        if "." in drop:
            a, b = drop.split(".")
            line.get(a, {}).pop(b, None)
        elif drop == "config":
            line["config"] = {"workload": _short(cfg.get("workload", ""), 120)}
        else:
            line.pop(drop, None)
        s = json.dumps(line, separators=(",", ":"))
    return s


def emit(result, log):
    """everything to bench_extra.json + stderr (short lines), then the ONE compact line on stdout, last"""
    for path in (os.path.join(ROOT, "bench_extra.json"), os.path.join(ROOT, "gpurun_out", "bench_extra.json")):
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                json.dump(result, f, indent=1, default=str)
        except OSError as e:
            log(f"bench_extra.json not written to {path}: {e}")
    for k, v in result.get("roofline", {}).get("legs", {}).items():
        log(f"[leg {k}] {json.dumps(v, default=str)[:1500]}")
    for k, v in result.get("extra", {}).items():
        log(f"[extra {k}] {json.dumps(v, default=str)[:1500]}")
    if "cpu_baseline" in result:
        log(f"[cpu_baseline] {json.dumps(result['cpu_baseline'], default=str)[:3000]}")
    s = compact_line(result)
    assert len(s) <= LINE_LIMIT and "\n" not in s
    sys.stderr.flush()
    print(s, flush=True)


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
               "exchange_mode": c["config"].get("exchange_mode"), "exchange_gather_us": c["config"].get("exchange_gather_us"),
               "what": "cull + k_cull_pack into the send buffer + ONE ncclAllGather per step, world of one rank (bench.py --force-collective): the step `--gpus N` times for N > 1"}
        log(f"[exchange path, one rank{', ' + str(extra_env) if extra_env else ''}] {out['ms_per_step'] * 1e3:.2f} us per step")
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher around it (WORLD_SIZE unset): start the N ranks the way the driver's multi-GPU line does -
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <a free port> bench.py <the same
    arguments> - and hand rank 0's ONE JSON line through on stdout. Everything else the ranks (or the launcher) write to stdout goes to
    stderr: stdout carries the line and nothing else, as with N = 1."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    if "--ranks-share-gpu" not in argv:
        try:
            import torch

            have = torch.cuda.device_count()
        except Exception:  # noqa: BLE001
            have = 0
        if have < n:
            raise SystemExit(f"bench.py --gpus {n}: this node shows {have} GPU(s) (one rank per GPU; --ranks-share-gpu is the one-GPU TEST mode)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    log(f"bench.py --gpus {n}: WORLD_SIZE is not set, launching the ranks: {' '.join(cmd)}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (the host driver only supports dmabuf IPC: RCCL needs it across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stdin=subprocess.DEVNULL, text=True, errors="replace")
    line = None
    for out in p.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and out.endswith("}"):
            if line is not None:
                log(line)
            line = out
        elif out:
            log(out)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    if rc != 0:
        log(f"bench.py --gpus {n}: the launcher ended with {rc}")
        return rc
    return 0 if line is not None else 1


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
    return int(round(fetch + write)), ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate child passes over the all_test cold leg's "
                                       "scene / camera / scrub (tools/run_workload.py); FETCH_SIZE doubled per the guide's gfx950 correction")


def load_traffic(kernel):
    """HBM bytes per launch of the roofline leg, measured by separate rocprofv3 PMC passes (tools/collect_traffic.sh; the passes of a whole round: tools/gpu_call.sh final counters) and COMMITTED as
    profiles/rNN/traffic.json - a builder-side measurement of the same command, not of this run. None when no such file exists."""
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


class CpuBaseline:
    """The reference CPU path on this box's host cores, on the FULL headline workload (10 M entities, same scene, same frustum):
    oracle/_ref (the reference's OWN culling_system.cpp + page_allocator.cpp + math / geometry, compiled in place; threads, Mutex and
    os::mem* underneath are stand-ins) when present, else the plain-C port. The scene is added and the page pool warmed up on a
    background thread while the GPU legs run; the timed part is a bounded thread sweep of the reference's jobs::forEach on a PERSISTENT
    worker pool (~25 s in all): median + p10 / p90 per thread count. Baseline only - the GPU / CPU ratio says nothing about kernel
    quality. (The only place bench.py touches oracle/: the checker timed as a baseline, never the thing shipped.)"""

    THREADS = (1, 8, 32)

    def __init__(self, scenes, sc, frustum, log, quick=False):
        import threading

        from oracle import pyoracle

        self.log, self.sc, self.fr, self.quick = log, sc, np.ascontiguousarray(frustum), quick
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
        except Exception as e:  # noqa: BLE001 - reported in the JSON line instead of killing the bench
            self.error = repr(e)

    def measure(self):
        self.thread.join()
        if self.error:
            return {"value": None, "unit": "entities/s", "cores": 0, "kind": self.kind, "sample": "not measured", "error": self.error}
        n = len(self.sc["entity"])
        host = os.cpu_count() or 1
        sweep = {}
        for threads in self.THREADS:
            if threads > host and threads != 1:
                continue
            # median of >= 20 timed frames at 1 and 8 threads (the page pool is filled by _prepare), bounded in time
            want, budget = (20, 10.0) if threads in (1, 8) else (5, 4.0)
            if self.quick:
                want, budget = 2, 2.0
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
            "sample": f"full headline workload ({n} entities, same scene + frustum as the GPU run), median of {sweep[best]['culls']} culls on {best} of {host} cores = best of thread sweep {list(sweep)}: more threads are slower "
                      f"({pages} result pages through ONE mutexed PageAllocator; Mutex / job threads are pthread stand-ins - the sweep's shape may be theirs, not the engine's)",
            "host_cores": host,
            "thread_sweep": {str(k): v for k, v in sweep.items()},
            "single_thread_value": sweep[1]["entities_per_s"],
            "scene_add_s": round(self.t_add, 2),
            "first_cull_s": round(self.t_first_cull, 2),
            "visible": int(visible),
            "describe": self.o.describe(),
        }
        if not self.quick:
            out.update(self._o3_line(n))
            out.update(self._other())
        return out

    def _o3_line(self, n):
        """SURVEY.md 8d's SECOND CPU line: the plain-C restatement built -O3 -march=x86-64-v3 (AVX2 + FMA allowed: NOT an oracle, the rounding may differ),
        one thread, the same scene and frustum - clearly labelled, never `value`."""
        try:
            from oracle import pyoracle

            o3 = pyoracle.Oracle("port_o3")
            ocs = o3.culling_system()
            ocs.add_bulk(self.sc["entity"], self.sc["type"], self.sc["pos"], self.sc["radius"])
            ocs.cull(self.fr, n_threads=1, want_ids=False, cap=0)
            times, t_start = [], time.time()
            while len(times) < 10 and (time.time() - t_start) < 8.0:
                t0 = time.perf_counter()
                ocs.cull(self.fr, n_threads=1, want_ids=False, cap=0)
                times.append(time.perf_counter() - t0)
            med = float(np.median(times))
            return {"O3_avx2_port_value": n / med, "O3_avx2_port_is": f"the plain-C restatement built -O3 -march=x86-64-v3 (FMA allowed: not an oracle), 1 thread, median of {len(times)} culls of the same workload"}
        except Exception as e:  # noqa: BLE001 - an optional line
            return {"O3_avx2_port_error": repr(e)}

    def _other(self):
        """the other two metrics of SURVEY.md 8d, bounded samples, one thread and 8 threads (parallel over instances / roots)"""
        o, scenes, other = self.o, self.scenes, {}
        try:
            sk = scenes.skeleton(64, seed=4)
            verts, skin = scenes.skinned_mesh(10_000, 64, seed=6)
            n_inst = 10_000  # BASELINE config 3 at FULL size: 10 k instances of the 10 k-vertex mesh, 10^8 vertices per frame (1.2 GB of positions)
            rp, rr = scenes.relative_poses(n_inst, 64, seed=5)
            inv = o.invert_bind(sk["bind"])
            for threads in (1, 8):
                t_pose, t_skin, t_start = [], [], time.time()
                while len(t_skin) < (3 if threads == 1 else 8) and (time.time() - t_start) < 12.0:  # a frame is ~3 s at 1 thread: bounded, the count is reported
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
            for _ in range(10):  # every root moved per frame: World::setTransform per root = the DFS of world.cpp:255-282
                new_root = scenes.random_transforms(rng_t, len(roots), 4000.0)
                t0 = time.perf_counter()
                w.set_transforms(roots, new_root)
                t_x.append(time.perf_counter() - t0)
            other["transforms_per_sec_1thread"] = len(kids) / float(np.median(t_x))
            other["transform_frames_timed"] = len(t_x)
            other["other_samples"] = (f"skin: {n_inst} instances x 64 bones x {len(verts)} verts of one mesh (config 3 at full size: 10^8 vertices per frame), median of <= 3 frames at 1 thread and <= 8 at 8 threads; "
                                      f"transforms: {len(roots)} roots x depth-4 chains (config 3 at full size), every root moved per frame, median of 10 frames, 1 thread "
                                      "(World is single-writer by design)")
        except Exception as e:  # noqa: BLE001 - the headline baseline must survive a problem in the side measurements
            other["other_error"] = repr(e)
        return other


if __name__ == "__main__":
    sys.exit(main())
