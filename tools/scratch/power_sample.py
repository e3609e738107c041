"""clocks / package power (rocm-smi) while one workload runs back to back: cull_all_test (k_cull_tile<F = 1>, every sphere fetched and tested), cull8
(k_cull_tile<F = 0>, 8 frusta in one pass), keys (cull + createSortKeys), xform (propagate 1 M nodes)"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from lumixengine_amd import api, scenes


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    keep = [l.split(":", 1)[1].strip() for l in out.splitlines() if "sclk" in l or "Package Power" in l]
    return " | ".join(keep)


ctx = api.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
N = 10_000_000
sc = scenes.cull_scene(N, 15000.0, seed=2)
sc["radius"] = scenes.all_test_radii(N)
cs = api.CullingSystem(ctx)
cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
fr1 = api.viewport_frustum()
fr8 = np.concatenate([api.viewport_frustum(**kw) for kw in scenes.config5_cascade_kwargs()])
cs.setPassWidth(8)
print("idle:", smi(), flush=True)
for name, fr, n in (("cull_all_test, 1 frustum", fr1, 60000), ("cull_all_test, 8 frusta in one pass", fr8, 16000)):
    stop = False
    t0 = time.perf_counter()
    k = 0
    samples = []

    def sampler():
        time.sleep(1.0)
        for _ in range(3):
            samples.append(smi())
            time.sleep(0.3)

    th = threading.Thread(target=sampler)
    th.start()
    while th.is_alive():
        for _ in range(200):
            cs.cull(fr)
        k += 200
        ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {1e6 * dt / k:.1f} us per cull over {dt:.1f} s", flush=True)
    for s in samples:
        print("   ", s, flush=True)
