"""The native exchange (lmx_exchange_*: csrc/lmx_capi_exchange.hip) on one GPU: a world of one rank runs the full code path - the
cull's gather kernels write the [8 counts | ids] record, ncclAllGather ships it on the side stream, frames alternate between two
slots - and the gathered record must hold exactly the oracle's visible ids per type. (More ranks: tests/test_distributed.py checks
the partition and the record format over gloo; the 8-GPU run is the driver's.)"""
import os

import numpy as np
import pytest

from lumixengine_amd import api, scenes
from lumixengine_amd import distributed as D
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["side", "inline", "auto", "p2p"])
def test_exchange_single_rank_matches_oracle(gpu_ctx, oracle_port, mode, monkeypatch):
    if mode == "p2p" and os.environ.get("LMX_HOSTSIM") == "1":
        pytest.skip("hipIpc mappings do not exist on the simulated device")
    monkeypatch.setenv("LMX_EXCHANGE_MODE", mode)  # (read when the exchange is created): where the all-gather runs, or no collective at all
    sc = scenes.cull_scene(200_000, 4000.0, seed=13, mixed_types=True)
    cs = api.CullingSystem(gpu_ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    ocs = oracle_port.culling_system()
    ocs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    cams = H.frusta(api, names=["origin_identity", "origin_yaw_pitch", "narrow_fov", "ortho_cascade_large"])
    want = []
    for f in range(len(cams)):
        ids, types, _ = ocs.cull(cams[f : f + 1])
        want.append([np.sort(ids[types == t]) for t in range(8)])
    cap = max(sum(len(a) for a in w) for w in want) + 100
    x = api.VisibleExchange(gpu_ctx, 0, 1, api.exchange_unique_id(), cap)
    try:
        assert x.info()["gather_us"] is None  # nothing is timed before a frame says what record it ships
        x.wait(x.cull(cams[0]))
        info = x.info()
        assert info["mode"] == (mode if mode != "auto" else info["mode"]) and info["mode"] in ("inline", "side", "p2p")
        assert (info["gather_us"] is not None and info["gather_us"] > 0) == (mode == "auto"), info  # timed only when the choice is made by measurement: on the first frame's record
        # pipelined: two frames in flight, read back one frame late (what a renderer consuming last frame's list does)
        slots = []
        for frame in range(8):
            slots.append(x.cull(cams[frame % len(cams)]))
            if frame >= 1:
                counts, ids = x.read(slots[frame - 1], 0)
                parsed, over = D.parse_records(np.concatenate([counts.astype(np.int32), np.pad(ids, (0, cap - len(ids)))])[None, :], cap)
                assert not over
                for t in range(8):
                    assert np.array_equal(np.sort(parsed[0][t]), want[(frame - 1) % len(cams)][t]), (frame, t)
        # the frame's views in ONE collective (config 5: the cascades of a frame are one exchange step): n sub-records of cap // n ids
        big = api.VisibleExchange(gpu_ctx, 0, 1, api.exchange_unique_id(), cap * len(cams))
        try:
            for frame in range(3):  # both slots, the third re-uses the first
                slot_m = big.cullMany(cams)
                for f in range(len(cams)):
                    counts, ids = big.readMany(slot_m, 0, f)
                    at = 0
                    for t in range(8):
                        assert int(counts[t]) == len(want[f][t]) and np.array_equal(np.sort(ids[at : at + int(counts[t])]), want[f][t]), (frame, f, t)
                        at += int(counts[t])
        finally:
            big.close()
        # type filter
        slot = x.cull(cams[0], 2)
        counts, ids = x.read(slot, 0)
        assert counts[2] == len(want[0][2]) and counts.sum() == counts[2] and np.array_equal(np.sort(ids), want[0][2])
    finally:
        x.close()
    # a capacity that is too small: the counts still tell what the rank saw, the ids are clipped, nothing is written out of bounds
    small = api.VisibleExchange(gpu_ctx, 0, 1, api.exchange_unique_id(), 64)
    try:
        slot = small.cull(cams[0])
        counts, ids = small.read(slot, 0)
        assert [int(c) for c in counts] == [len(a) for a in want[0]] and len(ids) == 64
        assert np.all(np.isin(ids, np.concatenate(want[0])))
    finally:
        small.close()
    # a rank that owns nothing (a strong split with fewer occupied cells than ranks): its record still reports zero counts
    cs.build(np.zeros(0, np.int32), np.zeros(0, np.uint8), np.zeros((0, 3)), np.zeros(0, np.float32))
    empty = api.VisibleExchange(gpu_ctx, 0, 1, api.exchange_unique_id(), 64)
    try:
        for _ in range(3):  # both slots
            slot = empty.cull(cams[0])
            counts, ids = empty.read(slot, 0)
            assert int(counts.sum()) == 0 and len(ids) == 0
    finally:
        empty.close()


def test_exchange_capacities_follow_a_one_frustum_list_when_asked(gpu_ctx, oracle_port, monkeypatch):
    """LMX_EXCHANGE_AUTO_CAPS=1 extends the capacities-follow-the-lists rule to frames of ONE frustum (off by default: the headline step keeps its caller-given
    capacity and pays neither the statistics kernel nor the host wait): created for 200 000 ids, the record shrinks to what the list needs two frames later,
    grows again when the camera sees more, and every frame's ids stay the oracle's; `keep_fixed` pins a shape."""
    monkeypatch.setenv("LMX_EXCHANGE_AUTO_CAPS", "1")
    monkeypatch.setenv("LMX_EXCHANGE_MODE", "inline")
    sc = scenes.cull_scene(200_000, 4000.0, seed=13, mixed_types=True)
    cs = api.CullingSystem(gpu_ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    ocs = oracle_port.culling_system()
    ocs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    cams = H.frusta(api, names=["narrow_fov", "origin_identity"])
    want = [np.sort(ocs.cull(cams[f : f + 1])[0]) for f in range(2)]
    if len(want[0]) > len(want[1]):  # camera 0 = the one that sees less
        cams, want = cams[::-1].copy(), want[::-1]
    assert 0 < len(want[0]) * 2 < len(want[1]) < 150_000
    x = api.VisibleExchange(gpu_ctx, 0, 1, api.exchange_unique_id(), 200_000)
    try:
        caps = []
        for frame, cam in enumerate([0, 0, 0, 0, 1, 1, 1, 1]):
            slot = x.cull(cams[cam])
            st = x.stats(slot)
            counts, ids = x.read(slot, 0)
            caps.append(st["caps"][0])
            if st["overflow_mask"] == 0:
                assert np.array_equal(np.sort(ids), want[cam]), frame
            else:  # the two frames after the camera change: flagged, clipped, never out of bounds
                assert int(counts.sum()) == len(want[cam]) > st["caps"][0] == len(ids) and set(ids.tolist()) <= set(want[cam].tolist()), frame
        assert caps[0] == caps[1] == 200_000 and caps[2] == caps[3] == D.cap_for(len(want[0])), caps
        assert caps[4] == caps[5] == caps[2] and caps[6] == caps[7] == D.cap_for(len(want[1])), caps  # frames 4 / 5 overflow, frame 6 (slot 0 again) has regrown
        x.setCaps([200_000], keep_fixed=True)
        for _ in range(4):
            assert x.layout(x.cull(cams[0]))["caps"] == [200_000]
    finally:
        x.close()


def _loopback_library():
    """tests/cpp/loopback_rccl.cpp -> tests/_build/libloopback_rccl.so (g++ against the HIP runtime; host code only)"""
    import os
    import subprocess

    if os.environ.get("LMX_HOSTSIM") == "1":  # pytest --hostsim: conftest built the stand-in against the simulated device
        return os.environ["LMX_RCCL_LIBRARY"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "cpp", "loopback_rccl.cpp")
    out = os.path.join(root, "tests", "_build", "libloopback_rccl.so")
    if not os.path.exists(out) or os.path.getmtime(src) > os.path.getmtime(out):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", out,
                        "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"], check=True)
    return out


@pytest.mark.parametrize("world,mode", [(2, "side"), (4, "side"), (8, "side"), (2, "inline"), (4, "inline"), (2, "auto_slow_gather"), (2, "p2p"), (4, "p2p")])
def test_exchange_ranks_on_one_gpu_loopback(tmp_path, oracle_port, world, mode):
    """The exchange with a world of 2 / 4 ranks on this box's one GPU: one process per rank, each with its own context and its cell shard of one
    scene, the collective carried by a shared-memory test double of the five RCCL entry points (RCCL itself refuses two ranks on one
    device). Everything around the wire is the product's: the per-rank / per-frustum record layout, the peers' offsets in the receive
    buffer, slot alternation over 6 pipelined frames, the type filter, the F-frusta record, clipping. Every rank must read, for every
    rank r, exactly the oracle's visible ids that live in r's shard."""
    import os
    import subprocess
    import sys

    if mode == "p2p" and os.environ.get("LMX_HOSTSIM") == "1":
        pytest.skip("hipIpc mappings do not exist on the simulated device")
    lib = _loopback_library()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # mode: where the all-gather runs (side stream / cull stream: same records, same pipelining rules), chosen by measurement ("auto": the double is
    # told to take 200 us per gather, so the exchange must pick the side stream - exchange_rank.py asserts what lmx_exchange_info reports), or
    # "p2p": no collective in the step at all - the ranks store into each other's receive buffers through REAL hipIpc mappings (one device, one
    # process per rank) and wait for sequence flags; the double only carries the handles at creation
    env = dict(os.environ, LMX_RCCL_LIBRARY=lib, LMX_EXCHANGE_MODE="auto" if mode.startswith("auto") else mode, LMX_EXPECT_EXCHANGE_MODE="side" if mode.startswith("auto") else mode,
               PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    if mode == "auto_slow_gather":
        env["LOOPBACK_RCCL_DELAY_US"] = "200"
    procs = [subprocess.Popen([sys.executable, "-m", "tests.exchange_rank", str(r), str(world), str(tmp_path)], cwd=root, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=600)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            p.kill()  # (the exact process this test started)
            logs.append(p.communicate()[0].decode(errors="replace") + "\n[timed out]")
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)

    sc = scenes.cull_scene(120_000, 4000.0, seed=13, mixed_types=True)
    ocs = oracle_port.culling_system()
    ocs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    cams = H.frusta(api, names=["origin_identity", "origin_yaw_pitch", "narrow_fov", "ortho_cascade_large"])
    got = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    owned = [set(got[r]["owned"].tolist()) for r in range(world)]
    assert len(set().union(*owned)) == sum(len(o) for o in owned) == len(sc["entity"])  # disjoint, complete
    want = []  # [camera][rank][type] -> sorted ids
    for f in range(len(cams)):
        ids, types, _ = ocs.cull(cams[f : f + 1])
        want.append([[np.sort(np.array([i for i in ids[types == t] if i in owned[r]], np.int32)) for t in range(8)] for r in range(world)])
        assert sum(len(a) for r in range(world) for a in want[f][r]) == len(ids)

    def check(counts, ids, w, what):
        at = 0
        for t in range(8):
            assert int(counts[t]) == len(w[t]), (what, t, int(counts[t]), len(w[t]))
            assert np.array_equal(np.sort(ids[at : at + int(counts[t])]), w[t]), (what, t)
            at += int(counts[t])
        assert at == len(ids)

    for reader in range(world):
        g = got[reader]
        for frame in range(5):
            for r in range(world):
                check(g[f"single_f{frame}_r{r}_counts"], g[f"single_f{frame}_r{r}_ids"], want[frame % len(cams)][r], ("single", reader, frame, r))
        for frame in range(3):
            for r in range(world):
                for f in range(len(cams)):
                    check(g[f"many_f{frame}_r{r}_c{f}_counts"], g[f"many_f{frame}_r{r}_c{f}_ids"], want[f][r], ("many", reader, frame, r, f))
        # capacities that follow the lists (exchange_rank.py "grow"): frames 0 / 1 clip frustum 0 at 256 ids and say so, from frame 2 on every
        # list is whole, the record has shrunk to what the lists need, and every rank computed the same layout from the gathered counts alone
        most = [max(sum(len(a) for a in want[f][r]) for r in range(world)) for f in range(len(cams))]
        for frame in range(5):
            caps, misc = g[f"grow_f{frame}_caps"].tolist(), g[f"grow_f{frame}_misc"].tolist()
            assert caps == got[0][f"grow_f{frame}_caps"].tolist() and g[f"grow_f{frame}_max"].tolist() == most, (reader, frame)
            assert misc[1] == sum(8 + c for c in caps) and misc[4] == (4 * misc[1] if mode != "p2p" else misc[5])
            mine = [sum(len(a) for a in want[f][reader]) for f in range(len(cams))]
            assert misc[2] == sum(8 + min(v, c) for v, c in zip(mine, caps)) and misc[5] == 4 * misc[2]
            if frame < 2:
                assert caps[0] == 256 and misc[0] == (1 if most[0] > 256 else 0)
            else:
                assert misc[0] == 0 and caps == [D.cap_for(m) for m in most], (frame, caps, most)
            for r in range(world):
                for f in range(len(cams)):
                    c, ids = g[f"grow_f{frame}_r{r}_c{f}_counts"], g[f"grow_f{frame}_r{r}_c{f}_ids"]
                    total = sum(len(a) for a in want[f][r])
                    if total <= caps[f]:
                        check(c, ids, want[f][r], ("grow", reader, frame, r, f))
                    else:
                        assert int(c.sum()) == total and len(ids) == caps[f] and set(ids.tolist()) <= set(np.concatenate(want[f][r]).tolist())
        for r in range(world):
            c, ids = g[f"type2_r{r}_counts"], g[f"type2_r{r}_ids"]
            assert int(c[2]) == len(want[0][r][2]) and int(c.sum()) == int(c[2]) and np.array_equal(np.sort(ids), want[0][r][2])
            c, ids = g[f"small_r{r}_counts"], g[f"small_r{r}_ids"]
            total = sum(len(a) for a in want[0][r])
            assert int(c.sum()) == total and len(ids) == min(total, 64)  # the counts tell what the rank saw; the ids are clipped
            assert set(ids.tolist()) <= set(np.concatenate(want[0][r]).tolist())


@pytest.mark.gpu
def test_config5_frame_two_ranks_loopback(tmp_path, oracle_port):
    """distributed.config5_frame - what `bench.py --gpus N --config5-frame` runs - with a world of two ranks on the one device: each rank
    its own 150 k mixed-type entities under the 8 cascade frusta in one pass and ONE collective of 8 sub-records; the ranks agree on the
    capacity and on success through files (torch.distributed in the bench). Both ranks finish the timed loop, each rank's own sub-records
    are its local cull, every rank reads the same per-rank / per-frustum counts, and those are the oracle's."""
    import json
    import os
    import subprocess
    import sys

    world = 2
    lib = _loopback_library()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LMX_RCCL_LIBRARY=lib, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    procs = [subprocess.Popen([sys.executable, "-m", "tests.config5_rank", str(r), str(world), str(tmp_path)], cwd=root, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            p.kill()  # (the exact process this test started)
            logs.append(p.communicate()[0].decode(errors="replace") + "\n[timed out]")
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = [json.load(open(tmp_path / f"config5_rank{r}.json")) for r in range(world)]
    frusta = np.concatenate([api.viewport_frustum(**kw) for kw in scenes.config5_cascade_kwargs()])
    for r in range(world):
        assert "error" not in got[r], got[r]
        assert got[r]["own_sub_records_equal_local_cull"] is True and got[r]["ms_per_frame_max_over_ranks"] > 0
        assert got[r]["visible_per_rank_and_frustum"] == got[0]["visible_per_rank_and_frustum"]
        assert got[r]["ms_per_frame_max_over_ranks"] == got[0]["ms_per_frame_max_over_ranks"]  # the MAX over ranks, agreed on
    for r in range(world):  # the counts every rank read for rank r are the oracle's for r's scene
        sc = scenes.cull_scene(150_000, 4000.0, seed=21 + r, mixed_types=True)
        ocs = oracle_port.culling_system()
        ocs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
        want = [len(ocs.cull(frusta[f : f + 1])[0]) for f in range(len(frusta))]
        assert got[0]["visible_per_rank_and_frustum"][r] == want and got[r]["visible_per_frustum_this_rank"] == want, (r, want)
        assert all(w <= c for w, c in zip(want, got[0]["ids_per_rank_and_frustum"])) and sum(want) > 0
    # what the frame ships: per-frustum capacities that follow the lists - the record of the timed frames is within 1.3x of the bytes the
    # rank with the fullest record uses (round 5's equal split: 4.6x at config 5's size), nothing overflowed, every rank has the same layout
    for r in range(world):
        e = got[r]["exchange"]
        assert e["caps"] == got[0]["exchange"]["caps"] and e["record_bytes_per_rank"] == got[0]["exchange"]["record_bytes_per_rank"]
        assert e["overflow_mask"] == 0 and e["max_visible_any_rank"] == [max(v[f] for v in got[0]["visible_per_rank_and_frustum"]) for f in range(len(frusta))]
        assert all(m <= c for m, c in zip(e["max_visible_any_rank"], e["caps"]))
        assert e["bytes_used_this_rank"] == 4 * sum(8 + v for v in got[0]["visible_per_rank_and_frustum"][r])
    fullest = max(got[r]["exchange"]["bytes_used_this_rank"] for r in range(world))
    assert got[0]["exchange"]["record_bytes_per_rank"] <= 1.3 * fullest, (got[0]["exchange"], fullest)


def test_p2p_exchange_gives_up_on_a_missing_peer(tmp_path):
    """LMX_EXCHANGE_MODE=p2p never waits without a bound: rank 1 of a world of two creates the exchange (the hipIpc mappings exist) and then
    stays away from the step. Rank 0's step must come back - lmx_exchange_wait returns LMX_ERR_BUSY within the configured 300 ms - the
    exchange must refuse further steps, and the process must exit by itself (no hung kernel left on the device)."""
    import subprocess
    import sys
    import time

    if os.environ.get("LMX_HOSTSIM") == "1":
        pytest.skip("hipIpc mappings do not exist on the simulated device")
    lib = _loopback_library()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LMX_RCCL_LIBRARY=lib, LMX_EXCHANGE_MODE="p2p", LMX_EXCHANGE_P2P_TIMEOUT_MS="300", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-m", "tests.exchange_rank", str(r), "2", str(tmp_path), "missing_peer"], cwd=root, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=120)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            p.kill()  # (the exact process this test started)
            logs.append(p.communicate()[0].decode(errors="replace") + "\n[timed out: the bounded wait did not end]")
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    assert "rank 0: step gave up with LMX_ERR_BUSY" in logs[0] and "refuses further steps" in logs[0], logs[0]
    assert time.time() - t0 < 100
