"""Development probe: host enqueue time per cull over a 3000-call loop, unpaced and paced (an event every 16 calls, wait on the one
recorded 32-48 calls earlier). Shows the HIP runtime's one-time ~50 ms stall around the 850th kernel launch of a process that
bench.py steps over with its warm-up launches:

    python tools/hip_queue_stall_probe.py
"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from lumixengine_amd import api, scenes
ctx = api.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
N = 10_000_000
sc = scenes.cull_scene(N, 15000.0, seed=2)
cs = api.CullingSystem(ctx); cs.build(sc["entity"], sc["type"], sc["pos"], sc["radius"])
fr = api.viewport_frustum()
for _ in range(20): cs.cull(fr)
def run(total, pace):
    torch.cuda.synchronize(); T0 = time.perf_counter(); marks = []; events = []
    for i in range(total):
        cs.cull(fr)
        if pace and (i & 15) == 15:
            e = torch.cuda.Event(); e.record(); events.append(e)
            if len(events) > 2: events.pop(0).synchronize()
        if i % 250 == 249: marks.append(time.perf_counter())
    torch.cuda.synchronize(); T1 = time.perf_counter()
    prev = T0; blocks = []
    for m in marks: blocks.append(round((m - prev) * 1e6 / 250, 1)); prev = m
    print("pace", pace, "us/step overall %.2f" % ((T1 - T0) * 1e6 / total), "host us/call per 250-block:", blocks)
run(3000, False)
run(3000, True)
run(3000, False)
