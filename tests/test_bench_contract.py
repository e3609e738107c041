"""The bench line's contract (keys the driver and the judge read), checked on the committed output of the round's last full run
(profiles/rNN/bench_full_run.json = stdout of `python bench.py` on an MI355X): no GPU needed."""
import glob
import subprocess
import sys
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_line():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_full_run.json")))
    assert files, "no committed bench line"
    lines = [l for l in open(files[-1]).read().splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py prints exactly ONE JSON line on stdout"
    return json.loads(lines[0])


def test_bench_line_has_the_contract_keys():
    d = last_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == "entities_culled_per_sec" and d["unit"] == "entities/s" and "entities culled/sec" in baseline["metric"]  # the metric's first component
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["entities_per_gpu"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6  # whole-job throughput of the timed steps


def test_roofline_and_cpu_baseline_objects():
    d = last_line()
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    # moved bytes ~ algorithmic bytes on the roofline leg (within 15 %): the fraction counts bytes that really travel
    assert r["traffic"] is not None and 1.0 <= r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.15
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0


# ---- the REAL print path of bench.py, end to end, on the simulated device (VERDICT r3 "missing 1": BENCH_r03.json had parsed = null because
# ~20 KB of side measurements rode the one stdout line and nothing here ran bench.py itself) -------------------------------------------
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _check_compact(line: str, steps: int, warmup: int):
    assert "\n" not in line and len(line.encode()) <= 4096, len(line)
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert d["steps"] == steps and d["warmup"] == warmup and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert "legs" not in d["roofline"] and "extra" not in d and "thread_sweep" not in d["cpu_baseline"]  # those live in bench_extra.json
    assert abs(d["value"] - d["config"]["entities_per_gpu"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["config"]["timed_steps"] == steps * d["config"]["repetitions"]
    return d


def test_bench_main_prints_one_compact_line_on_the_simulated_device():
    import subprocess
    import sys

    from tests.hostsim import build as hostsim_build

    lib = hostsim_build.build()
    env = dict(os.environ, LMX_HOSTSIM="1", LMX_LIB_PATH=lib)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-hostsim", "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1 and out.endswith("\n"), "bench.py prints exactly ONE line on stdout"
    d = _check_compact(lines[0], 3, 1)
    assert "SELFTEST" in d["data"]  # a simulated-device run can never pass for a measurement
    assert "target_frames_per_sec_1gpu" in d["also"] and "skinned_verts_per_sec" in d["also"]  # the extras ran to the end
    full = json.load(open(os.path.join(ROOT, "bench_extra.json")))
    assert "legs" in full["roofline"] and "extra" in full and "error" not in full["extra"], full.get("extra", {}).get("error")
    assert max(len(l) for l in p.stderr.decode(errors="replace").splitlines()) < 4000  # stderr stays line-oriented too


def test_compact_line_holds_its_limit_whatever_the_legs_return():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    big = "x" * 20000
    result = {"metric": "entities_culled_per_sec", "value": 1.0e12, "unit": "entities/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.01, "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
              "config": {"workload": big, "entities_per_gpu": 10_000_000, "frusta": 1, "visible_per_gpu": 1, "visible_ids": "reference", "sharding": big, "timed_steps": 200, "repetitions": 10,
                         "config4_frame": {"ms_per_frame_max_over_ranks": 1.0, "what": big}},
              "roofline": {"kernel": "k_cull_tile", "bound": "hbm", "achieved": 4900.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.6125, "traffic": 223905113,
                           "algorithmic_bytes_per_launch": 2.0e8, "avg_launch_ms": 0.041, "leg": big, "legs": {"a": {"note": big}}, "traffic_source": big},
              "cpu_baseline": {"value": 1e8, "unit": "entities/s", "cores": 8, "kind": "reference", "sample": big, "thread_sweep": {str(i): {"x": big} for i in range(5)}},
              "extra": {"ab_variants": {"rows": [big] * 10}, "error": big, "skinned_verts_per_sec": 1.0}}
    line = bench.compact_line(result)
    assert len(line) <= 4096
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert bench.library_is_the_product(os.path.join(ROOT, "lumixengine_amd", "liblumix_mi355.so"))
    assert not bench.library_is_the_product(os.path.join(ROOT, "tests", "_build", "hostsim", "liblumix_hostsim.so"))  # bench.py never times the simulated device


def test_bench_gpus_n_without_a_launcher_starts_its_own_ranks_or_says_why_not():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (how the driver's N = 1 line would look with another N): bench.py becomes the launcher. On a box with
    fewer GPUs than ranks - this container has none - it must say so and exit non-zero at once, not hang in a rendezvous and not print a JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LMX_LIB_PATH", "LMX_HOSTSIM")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s)" in r.stderr and "--gpus 2" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
