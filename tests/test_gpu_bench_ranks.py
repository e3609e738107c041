"""bench.py's N > 1 code path, end to end, on a one-GPU box: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2
--ranks-share-gpu` - the driver's launch line with two ranks that share cuda:0, torch.distributed on gloo and the exchange's collective
carried by the shared-memory test double of RCCL (tests/cpp/loopback_rccl.cpp; RCCL refuses two ranks on one device). What is checked is
that the path the driver will run on 8 GPUs runs at all with more than one rank and ships the right things: the JSON line's contract
fields, both ranks' records seen by RCCL's stand-in, the strong-scaling config-4 frame riding along the weak run, and - strong
scaling - the union of the ranks' lists equal to the unsharded cull. Timings of this mode mean nothing."""
import json
import os
import subprocess
import sys

import pytest

from tests.test_gpu_exchange import _loopback_library

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, port, launcher="torchrun"):
    if os.environ.get("LMX_HOSTSIM") == "1":
        pytest.skip("bench.py times a GPU (torch.cuda streams and events): not under pytest --hostsim")
    env = dict(os.environ, LMX_RCCL_LIBRARY=_loopback_library())
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    # "torchrun": the driver's multi-GPU launch line; "self": plain `python bench.py --gpus 2` - bench.py starts its ranks itself
    front = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] if launcher == "torchrun" else [sys.executable]
    cmd = front + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--ranks-share-gpu", "--entities", "1000000",
                   "--skinned-instances", "2000", "--no-cpu-baseline", "--no-live-traffic", "--big-entities", "0"] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    lines = [l for l in p.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stderr.decode(errors="replace")[-4000:]
    assert len(lines[0]) <= 4096, "the one stdout line stays compact with N > 1 too"
    line = json.loads(lines[0])
    full = json.load(open(os.path.join(ROOT, "bench_extra.json")))  # rank 0's full record of the same run
    assert full["value"] == line["value"] and full["n_gpus"] == line["n_gpus"]
    for k in ("ranks_seen_by_rccl", "allgather_visible_counts"):
        assert line["config"][k] == full["config"][k]
    assert "ms_per_frame_max_over_ranks" in line["config"]["config4_frame"]
    return full


def check_contract(r):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in r, k
    assert r["n_gpus"] == 2 and r["steps"] == 20 and r["warmup"] == 5 and r["metric"] == "entities_culled_per_sec" and r["vs_baseline"] is None
    assert r["config"]["ranks_seen_by_rccl"] == 2 and len(r["config"]["allgather_visible_counts"]) == 2


@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_bench_weak_two_ranks(launcher):
    r = run_bench([], 29541, launcher)
    check_contract(r)
    cfg = r["config"]  # what the exchange of the timed step shipped: in the record, so that a bad scaling curve can be read from it
    assert cfg["exchange_mode"] in ("inline", "side", "p2p") and cfg["exchange_overflow_mask"] == 0
    assert cfg["exchange_bytes_used_per_rank"] == 4 * (8 + cfg["allgather_visible_counts"][0]) <= cfg["exchange_bytes_shipped_per_rank"]
    assert cfg["exchange_bytes_arriving_per_rank"] == cfg["exchange_bytes_shipped_per_rank"] and cfg["exchange_gather_us"] > 0
    assert cfg["exchange_gather_record_bytes"] == cfg["exchange_bytes_shipped_per_rank"]
    assert r["scaling"] == "weak" and abs(r["value"] - 2 * 1_000_000 / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    c4 = r["config"]["config4_frame"]  # the strong-scaling frame of BASELINE config 4 rides along the weak run
    assert "error" not in c4 and c4["scaling"] == "strong" and c4["skinned_instances_this_rank"] == 1000 and c4["frames_per_sec"] > 0
    assert 0 < c4["visible_total"] <= 1_000_000
    c5 = r["config"]["config5_frame"]  # BASELINE config 5's frame (8 cascades, ONE collective) is part of the default --gpus N path
    assert "error" not in c5 and c5["own_sub_records_equal_local_cull"] is True and c5["ms_per_frame_max_over_ranks"] > 0
    assert len(c5["visible_per_rank_and_frustum"]) == 2 and len(c5["visible_per_rank_and_frustum"][0]) == 8
    e5 = c5["exchange"]  # per-frustum capacities: the 8-sub-record frame ships at most 1.3x what its fullest rank uses (4.6x with one capacity for all)
    fullest = 4 * max(sum(8 + v for v in per_rank) for per_rank in c5["visible_per_rank_and_frustum"])
    assert e5["overflow_mask"] == 0 and e5["record_bytes_per_rank"] <= 1.3 * fullest, (e5, fullest)


@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_bench_strong_two_ranks(launcher):
    r = run_bench(["--scaling", "strong"], 29543, launcher)
    check_contract(r)
    assert r["scaling"] == "strong" and r["config"]["union_equals_unsharded"] is True
    assert sum(r["config"]["allgather_visible_counts"]) == r["config"]["visible_total"]
    assert r["config"]["config4_frame"]["skinned_instances_this_rank"] == 1000
