#!/usr/bin/env python
"""A/B of kernel experiments that are bit-exact on the simulated device (tests/hostsim) but have no GPU time yet.

Every experiment is a build variant of liblumix_mi355.so (one or two kernel sources re-compiled with a -D flag, linked with the regular
objects) or a run-time option of the regular library. `bench.py` runs `run_all` at the very end of its default run, every measurement in a
CHILD process (its own HIP context, a hard timeout; the bench line's own numbers are all taken before the first child starts), and carries
the table as `extra.ab_variants`: the same workloads as tools/run_workload.py, kernel time by the dispatches' own timestamps (lmx_profile_*),
the results of every variant compared with the base library's and - where a reference digest exists - with the reference's.
Nothing here changes what the product does: a variant only becomes the default after it won here.

    python tools/ab_variants.py --build                     # what __graft_entry__.build() calls: tools/_build/variants/<name>/liblumix_mi355.so
    python tools/ab_variants.py --run [--budget 150]        # the table bench.py embeds (GPU box)
    python tools/ab_variants.py --measure keys [--small]    # one child; LMX_LIB_PATH selects the library
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

VARIANT_DIR = os.path.join(ROOT, "tools", "_build", "variants")

# name -> (group, sources re-compiled, flags, what it is). Run-time options of the base library (the key mirror's forms) are legs of the
# group's child, not build variants.
VARIANTS = {
    "cull_hdr_ahead": ("cull", ["cull_kernels.hip"], ["-DLMX_CULL_HDR_AHEAD=1"],
                       "k_cull_tile phase B resolves header -> cell -> class of all the wave's chunks before the first group's loads"),
    "keys_stage_pairs": ("keys", ["keys_kernels.hip"], ["-DLMX_KEYS_STAGE_PAIRS=1"],
                         "k_keys_mesh: a tile's (key, value) pairs / instancer records leave through LDS in position order"),
    "keys_block_256": ("keys", ["keys_kernels.hip"], ["-DLMX_KEYS_BLOCK=256"], "k_keys_mesh: tiles of 256 entities (twice the same-address reservations per launch, 4-wave blocks)"),
    "keys_block_1024": ("keys", ["keys_kernels.hip"], ["-DLMX_KEYS_BLOCK=1024"], "k_keys_mesh: tiles of 1024 entities (half the same-address reservations per launch, 16-wave blocks)"),
    "pose_stage1": ("pose", ["skin_kernels.hip"], ["-DLMX_POSE_STAGE_OUT=1"], "k_pose_palette: palette rows leave through LDS staging rows"),
    "pose_stage2": ("pose", ["skin_kernels.hip"], ["-DLMX_POSE_STAGE_OUT=2"], "k_pose_palette: palette rows and the absolute pose leave through LDS staging rows"),
}
GROUPS = ("cull", "keys", "pose")


def variant_lib(name: str) -> str:
    return os.path.join(VARIANT_DIR, name, "liblumix_mi355.so")


def build_all(log=print) -> dict:
    """Builds every variant next to the regular library (hipcc cross-compiles: no GPU needed). Returns {name: path}."""
    from concurrent.futures import ThreadPoolExecutor

    from lumixengine_amd import build as B

    B.build()

    def one(name):
        _, sources, flags, _ = VARIANTS[name]
        out_dir = os.path.join(VARIANT_DIR, name)
        os.makedirs(out_dir, exist_ok=True)
        lib = variant_lib(name)
        own = {}
        stale_lib = not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(B.LIB)
        for source in sources:
            obj = os.path.join(out_dir, os.path.splitext(source)[0] + ".o")
            own[source] = obj
            if B._stale(obj, [os.path.join(B.CSRC, source), os.path.abspath(__file__)] + B.HEADERS):
                subprocess.run([B.hipcc()] + B.FLAGS + flags + ["-x", "hip", "-c", os.path.join(B.CSRC, source), "-o", obj], check=True, capture_output=True)
                stale_lib = True
        if stale_lib:
            objs = [own.get(s, os.path.join(B.OBJ, os.path.splitext(s)[0] + ".o")) for s in B.SOURCES]
            subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"], check=True, capture_output=True)
        return name, lib

    with ThreadPoolExecutor(max_workers=len(VARIANTS)) as ex:
        libs = dict(ex.map(one, VARIANTS))
    log(f"[ab_variants] built {len(libs)} variants under {os.path.relpath(VARIANT_DIR, ROOT)}")
    return libs


# ---------------------------------------------------------------------------------------------------------------------------------------
# children: one workload group each, ONE JSON line on stdout
# ---------------------------------------------------------------------------------------------------------------------------------------

def _sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def _scrubber():
    """A 1 GiB read-only reduction between launches (cache-cold legs) when torch sees the GPU; None on the simulated device."""
    try:
        import torch

        if not torch.cuda.is_available():
            return None, None
        scrub = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda").view(torch.int32)
        return (lambda: scrub.sum()), torch.cuda.current_stream().cuda_stream
    except Exception:  # noqa: BLE001
        return None, None


def _golden_sha(scene: str, camera: str = "default"):
    try:
        g = json.load(open(os.path.join(ROOT, "tests", "golden", "cull_bench_scenes.json")))
        return g["scenes"][scene]["cameras"][camera]["all_types_sha256"]
    except Exception:  # noqa: BLE001
        return None


def measure_cull(small: bool) -> dict:
    from lumixengine_amd import api, scenes

    scrub, stream = _scrubber()
    ctx = api.Context(0)
    if stream is not None:
        ctx.set_stream(stream)
    N = 200_000 if small else 10_000_000
    half = 15000.0 * (N / 1e7) ** (1.0 / 3.0)
    reps = 5 if small else 60
    fr = api.viewport_frustum()
    away = api.viewport_frustum(pos=(0.0, 0.0, -5.0 * half))
    out = {}

    def kernel_us(cs, frustum, n, cold):
        ctx.profile_reset()
        ctx.profile_enable(True)
        for _ in range(n):
            if cold and scrub:
                scrub()
            cs.cull(frustum)
        ctx.synchronize()
        ctx.profile_enable(False)
        ms, launches = ctx.profile_get(api.K_CULL_SPHERES)
        return round(1e3 * ms / max(launches, 1), 3)

    sc = scenes.cull_scene(N, half, seed=2)
    for scene_name, golden in (("sparse", "sparse_10m"), ("all_test", "all_test_10m")):
        if scene_name == "all_test":
            sc["radius"] = scenes.all_test_radii(N)
        cs = api.CullingSystem(ctx)
        cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
        for _ in range(20 if small else 300):
            cs.cull(fr)
        res = cs.cull(fr)
        ids = np.sort(res.all_ids(0)[0])
        leg = {"visible": int(len(ids)), "ids_sha": _sha(ids)}
        want = _golden_sha(golden) if N == 10_000_000 else None
        if want is not None:
            leg["ids"] = "reference" if hashlib.sha256(np.ascontiguousarray(ids.astype(np.int32)).tobytes()).hexdigest() == want else "differ from the reference's"
        leg["warm_kernel_us"] = kernel_us(cs, fr, reps, False)
        if scrub:
            leg["cold_kernel_us"] = kernel_us(cs, fr, max(5, reps // 2), True)
        if scene_name == "sparse":
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps * 5):
                cs.cull(fr)
            ctx.synchronize()
            leg["wall_us_per_cull"] = round((time.perf_counter() - t0) * 1e6 / (reps * 5), 3)
            for _ in range(5):
                cs.cull(away)
            leg["nothing_visible_kernel_us"] = kernel_us(cs, away, reps, False)
        out[scene_name] = leg
        del cs
    if os.environ.get("LMX_AB_ROUNDS_PROBE"):
        # Does the launch's LAST, partly filled round of resident blocks cost the 10 M launch its distance to the 100 M one (0.62 vs 0.72 of
        # HBM)? 10 M entities are 4883 tiles on 256 CUs x 7 or 8 resident blocks = 2.72 or 2.38 rounds. The same all-test scene (same
        # density) at sizes that are whole numbers of rounds for either residency, next to sizes that are not: if ns per entity dips at the
        # whole numbers, splitting the tiles of the last round is worth building; if it is flat, it is not.
        probe = {}
        sizes = [200_000, 300_000] if small else [7_340_032, 8_388_608, 9_175_040, 10_000_000, 11_010_048, 12_582_912]
        for n in sizes:
            h = 15000.0 * (n / 1e7) ** (1.0 / 3.0)
            sc_p = scenes.cull_scene(n, h, seed=2)
            sc_p["radius"] = scenes.all_test_radii(n)
            cs = api.CullingSystem(ctx)
            cs.build(sc_p["entity"], sc_p["type"], sc_p["pos"], sc_p["radius"])
            for _ in range(5 if small else 60):
                cs.cull(fr)
            visible = int(cs.cull(fr).counts()[0].sum())
            row = {"chunks_of_64": int(cs.stats()["chunks"]), "tiles": (n + 2047) // 2048, "rounds_at_8_blocks_per_cu": round((n + 2047) // 2048 / 2048.0, 3), "rounds_at_7_blocks_per_cu": round((n + 2047) // 2048 / 1792.0, 3),
                   "visible": visible, "warm_kernel_us": kernel_us(cs, fr, reps, False)}
            row["warm_ns_per_1000_entities"] = round(1e6 * row["warm_kernel_us"] / n, 3)
            if scrub:
                row["cold_kernel_us"] = kernel_us(cs, fr, max(5, reps // 2), True)
                row["cold_ns_per_1000_entities"] = round(1e6 * row["cold_kernel_us"] / n, 3)
                row["cold_frac_of_8TBps"] = round((20.0 * n + 4.0 * visible) / (row["cold_kernel_us"] * 1e-6) / 8e12, 4)
            probe[str(n)] = row
            del cs, sc_p
        out["rounds_probe"] = probe
    ctx.close()
    return out


def measure_keys(small: bool) -> dict:
    from lumixengine_amd import api, scenes

    ctx = api.Context(0)
    N = 200_000 if small else 10_000_000
    half = 5000.0 * (N / 1e7) ** (1.0 / 3.0)
    steps = 4 if small else 16
    sc = scenes.cull_scene(N, half, seed=2)
    cs = api.CullingSystem(ctx)
    cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    fr = api.viewport_frustum()
    ks = scenes.keys_scene(N, sc["type"], seed=12, max_sort_key=255)
    out = {}
    # LMX_KEYS_OPT_SPLIT_STATE: AoS mirror / + dense lod, Pose::frame array / structure of arrays; the base form twice (noise). Build variants
    # run forms 0 and 2 only (LMX_AB_KEYS_FORMS, set by run_all)
    forms = tuple(int(x) for x in os.environ.get("LMX_AB_KEYS_FORMS", "0,1,2,0").split(","))
    for form in forms:
        sk = api.SortKeys(ctx)
        sk.setOption(api.KEYS_OPT_SPLIT_STATE, form)
        sk.setModels(ks["models"], ks["mesh_types"])
        sk.setInstances(ks["model"], ks["material_offset"], ks["mesh_materials"], ks["lod"], ks["flags"], ks["dirty"], ks["pose_frame"])
        sk.setPositions(sc["pos"])
        view = lambda f: api.keys_view(layer_to_bucket=ks["layer_to_bucket"], bucket_depth_sorted=ks["bucket_depth_sorted"], frame_number=100 + f)  # noqa: E731
        for f in range(2):
            cs.cull(fr)
            sk.run(view(f), 255)
        ctx.synchronize()
        ctx.profile_reset()
        ctx.profile_enable(True)
        for f in range(2, 2 + steps):
            cs.cull(fr)
            sk.run(view(f), 255)
        ctx.synchronize()
        ctx.profile_enable(False)
        ms, launches = ctx.profile_get(api.KERNEL_NAMES.index("sort_keys"))
        keys, values = sk.readPairs()
        order = np.lexsort((values, keys))
        lod, frame = sk.readState()
        leg = {"split_state": form, "createSortKeys_us": round(1e3 * ms / max(launches, 1), 2), "pairs": int(len(keys)),
               "pairs_sha": _sha(np.stack([keys[order], values[order]])), "state_sha": _sha(np.concatenate([lod.view(np.uint32), frame]))}
        name = f"split_state_{form}" + ("_again" if f"split_state_{form}" in out else "")
        out[name] = leg
        sk.setOption(api.KEYS_OPT_SPLIT_STATE, 0)
        del sk
    out["visible"] = int(cs.cull(fr).counts()[0].sum())
    ctx.close()
    return out


def measure_pose(small: bool) -> dict:
    from lumixengine_amd import api, scenes

    ctx = api.Context(0)
    n_inst = 64 if small else 20_000
    n_verts = 2_000 if small else 10_000
    steps = 2 if small else 10
    s = scenes.skeleton(64, seed=4)
    sk = api.Skinning(ctx)
    model = sk.addModel(s["parents"], s["bind"], s["first_nonroot"])
    mesh = sk.addMesh(*scenes.skinned_mesh(n_verts, 64, seed=6))
    sk.setInstances(np.full(n_inst, model, np.uint32), np.full(n_inst, mesh, np.uint32))
    pos, rot = scenes.relative_poses(n_inst, 64, seed=8)
    for _ in range(2):
        sk.uploadPoses(pos, rot)  # (every run turns the library's poses into absolute ones: Pose::is_absolute)
        sk.run()
    ctx.synchronize()
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(steps):
        sk.uploadPoses(pos, rot)
        sk.run()
    ctx.synchronize()
    ctx.profile_enable(False)
    out = {"instances": n_inst, "bones": 64, "verts_per_instance": n_verts}
    for k, name in ((api.K_POSE_PALETTE, "pose_palette_us"), (api.K_SKIN_VERTICES, "skin_vertices_us")):
        ms, launches = ctx.profile_get(k)
        out[name] = round(1e3 * ms / max(launches, 1), 2)
    sample = np.arange(0, n_inst, max(1, n_inst // 8))[:8]
    out["palette_sha"] = _sha(np.concatenate([sk.readPalette(int(i)).view(np.float32).reshape(-1) for i in sample]))
    out["positions_sha"] = _sha(np.concatenate([sk.readVertices(int(i)).reshape(-1) for i in sample]))
    ctx.close()
    return out


MEASURE = {"cull": measure_cull, "keys": measure_keys, "pose": measure_pose}
# what a variant must reproduce bit for bit (compared with the base library's child)
IDENTITY = {"cull": (("sparse", "ids_sha"), ("all_test", "ids_sha")), "keys": (("split_state_0", "pairs_sha"), ("split_state_0", "state_sha"), ("split_state_2", "pairs_sha")),
            "pose": (("palette_sha",), ("positions_sha",))}


def _dig(d, path):
    for k in path:
        d = d[k]
    return d


def _child(group: str, lib, timeout_s: float, small: bool, extra_env=None) -> dict:
    env = dict(os.environ)
    env.update(extra_env or {})
    if lib:
        env["LMX_LIB_PATH"] = lib
    cmd = [sys.executable, os.path.abspath(__file__), "--measure", group] + (["--small"] if small else [])
    t0 = time.time()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT, stdin=subprocess.DEVNULL)
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout_s:.0f} s"}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
    out = json.loads(lines[-1])
    out["child_s"] = round(time.time() - t0, 1)
    return out


def run_all(log=print, budget_s: float = 150.0, small: bool = False, base_lib=None, libs=None) -> dict:
    """{group: {"base": ..., "<variant>": ..., "base_again": ...}}; a group / variant that does not fit the time budget is listed as skipped.
    base_lib / libs: other builds of the same sources (the tests pass the simulated device's)."""
    t_start = time.time()
    table = {"what": "kernel experiments that are bit-exact on the simulated device (tests/hostsim) but not timed before this run; each leg is a child process, "
                     "kernel time = the dispatches' own timestamps (lmx_profile_*); tools/ab_variants.py",
             "variants": {n: v[3] for n, v in VARIANTS.items()}}
    per_child = 75.0
    for gi, group in enumerate(GROUPS):
        legs = [("base", base_lib)] + [(n, (libs or {}).get(n) or variant_lib(n)) for n, v in VARIANTS.items() if v[0] == group] + [("base_again", base_lib)]
        rows = {}
        # a group may use what is left minus a reserve for the groups still to come (18 s per leg of theirs: enough for their base and
        # variant legs) - a slow group cannot turn every later row into "skipped", and time a fast group leaves is handed on
        reserve = 18.0 * sum(2 + sum(1 for v in VARIANTS.values() if v[0] == g) for g in GROUPS[gi + 1:])
        group_end = t_start + max(budget_s - reserve, 0.0)
        for name, lib in legs:
            left = min(budget_s - (time.time() - t_start), group_end - time.time())
            if lib and not os.path.exists(lib):
                rows[name] = {"skipped": "library not built (python tools/ab_variants.py --build)"}
                continue
            if left < 20.0:
                rows[name] = {"skipped": "time budget of the A/B spent"}
                continue
            # (the base library's first cull child also carries the rounds probe: tile-granularity of the 10 M launch, measure_cull)
            probe = {"LMX_AB_ROUNDS_PROBE": "1"} if (group == "cull" and name == "base") else None
            if group == "keys" and name != "base":
                probe = {"LMX_AB_KEYS_FORMS": "0,2"}
            rows[name] = _child(group, lib, min(per_child + (45.0 if group == "cull" and probe else 0.0), left), small, probe)
            log(f"[ab_variants] {group} / {name}: {json.dumps(rows[name])[:400]}")
        base = rows.get("base", {})
        for name, row in rows.items():
            if name.startswith("base") or "error" in row or "skipped" in row or "error" in base or "skipped" in base:
                continue
            try:
                row["results_equal_base"] = all(_dig(row, p) == _dig(base, p) for p in IDENTITY[group])
            except KeyError:
                row["results_equal_base"] = None
        table[group] = rows
    table["seconds"] = round(time.time() - t_start, 1)
    return table


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--measure", choices=GROUPS)
    ap.add_argument("--small", action="store_true", help="sizes for the simulated device")
    ap.add_argument("--budget", type=float, default=150.0)
    args = ap.parse_args()
    if args.build:
        for name, lib in build_all().items():
            print(name, lib)
    if args.measure:
        print(json.dumps(MEASURE[args.measure](args.small)), flush=True)
    if args.run:
        print(json.dumps(run_all(lambda *a: print(*a, file=sys.stderr, flush=True), args.budget, args.small), indent=1))


if __name__ == "__main__":
    main()
