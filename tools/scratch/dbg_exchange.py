import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lumixengine_amd import api, scenes
ctx = api.Context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sc = scenes.cull_scene(N, 15000.0 * (N / 1e7) ** (1 / 3), seed=2)
cs = api.CullingSystem(ctx)
cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
fr = api.viewport_frustum()
res = cs.cull(fr)
visible = int(res.counts()[0].sum())
cap = (visible * 5 // 4 + 1023) // 1024 * 1024
x = api.VisibleExchange(ctx, 0, 1, api.exchange_unique_id(), cap)
for i in range(6):
    slot = x.cull(fr)
x.wait(slot)
counts, ids = x.read(slot, 0)
local = np.sort(cs.cull(fr, view=2).all_ids(0)[0])
print("visible", visible, "record counts", counts.tolist(), "ids", len(ids), "local", len(local))
g = np.sort(ids)
print("equal", np.array_equal(g, local), "n_unique", len(np.unique(g)), "setxor", len(np.setxor1d(g, local)), "missing sample", np.setdiff1d(local, g)[:5], "extra sample", np.setdiff1d(g, local)[:5])
# shards view
print("min/max id", g.min() if len(g) else None, g.max() if len(g) else None)
x.close()
