"""Where do the slow frames of an update stream with LMX_CULL_OPT_ASYNC_COMPACTION come from? 10 M sorted + 2 M overflow entities, option on,
100 adds + a cull per frame at 1 kHz until the sets have traded places; prints the slowest frames with the worker's state around them."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lumixengine_amd import api, scenes

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ctx = api.Context(0)
sc = scenes.cull_scene(N, 15000.0, seed=2)
cs = api.CullingSystem(ctx)
over = N // 5
cs.setOption(api.CULL_OPT_AUTO_COMPACTION, 0)
cs.setOption(api.CULL_OPT_OVERFLOW_RESERVE, over + 400_000)
cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
rng = np.random.default_rng(12)
ids = np.arange(N, N + over, dtype=np.int32)
cs.addMany(ids, np.zeros(over, np.uint8), rng.uniform(-15000, 15000, size=(over, 3)), np.exp(rng.uniform(np.log(0.5), np.log(50.0), size=over)).astype(np.float32))
fr = api.viewport_frustum()
for _ in range(20):
    cs.cull(fr)
ctx.synchronize()
t0 = time.perf_counter()
cs.setOption(api.CULL_OPT_ASYNC_COMPACTION, 1)
print(f"enable: {1e3 * (time.perf_counter() - t0):.1f} ms")
cs.setOption(api.CULL_OPT_AUTO_COMPACTION, 1)
rows = []
next_id = N + over
t_start = time.perf_counter()
after = 0
while time.perf_counter() - t_start < 15.0:
    k = 100
    ids = np.arange(next_id, next_id + k, dtype=np.int32)
    next_id += k
    p = rng.uniform(-15000, 15000, size=(k, 3))
    r = np.ones(k, np.float32)
    t0 = time.perf_counter()
    cs.addMany(ids, np.zeros(k, np.uint8), p, r)
    t1 = time.perf_counter()
    cs.cull(fr)
    t2 = time.perf_counter()
    ctx.synchronize()
    t3 = time.perf_counter()
    st = cs.asyncStats()
    rows.append((len(rows), 1e3 * (t3 - t0), 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), st["state"], st["swaps"]))
    if st["swaps"] >= 1:
        after += 1
        if after > 200:
            break
    pause = 1e-3 - (time.perf_counter() - t0)
    if pause > 0:
        time.sleep(pause)
print(f"{len(rows)} frames, stats {cs.asyncStats()}, state {cs.updateStats()}")
first_swap = next((r[0] for r in rows if r[6] >= 1), None)
print("first frame with swaps >= 1:", first_swap)
print("slowest frames: (index, total ms, addMany, cull call, sync, worker state after, swaps)")
for r in sorted(rows[2:], key=lambda r: -r[1])[:12]:
    print("  %5d  %8.3f  %7.3f %8.3f %7.3f  state %d swaps %d" % r)
t = np.array([r[1] for r in rows[2:]])
print(f"median {np.median(t):.3f} ms, p99 {np.percentile(t, 99):.3f} ms, max {t.max():.3f} ms")
cs.setOption(api.CULL_OPT_ASYNC_COMPACTION, 0)
ctx.close()
