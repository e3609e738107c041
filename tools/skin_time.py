"""k_skin_multi per 1e9 vertices on the north-star frame's skinning load (100 k instances x 10 k vertices, 64 bones, one shared mesh): the worst-case
mesh (4 random bones per vertex) and the character-like mesh.
Launch begin / end timestamps of the kernels themselves, 12 frames after 12 of warm-up (the first launches first-touch the 12 GB output)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lumixengine_amd import api, scenes  # noqa: E402

ctx = api.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n_inst, n_verts = int(os.environ.get("LMX_SKIN_INSTANCES", 100_000)), 10_000
s = scenes.skeleton(64, seed=4)
sk = api.Skinning(ctx)
model = sk.addModel(s["parents"], s["bind"], s["first_nonroot"])
worst = sk.addMesh(*scenes.skinned_mesh(n_verts, 64, seed=6))
char = sk.addMesh(*scenes.skinned_mesh_character(n_verts, 52, seed=6))
pos, rot = scenes.relative_poses(n_inst, 64, seed=8)
d_pos, d_rot = torch.from_numpy(pos).cuda(), torch.from_numpy(rot).cuda()
for name, mesh in (("worst-case mesh", worst), ("character mesh", char), ("worst-case mesh again", worst)):
    sk.setInstances(np.full(n_inst, model, np.uint32), np.full(n_inst, mesh, np.uint32))
    sk.setPoseSourceDevice(d_pos.data_ptr(), d_rot.data_ptr(), n_inst * 64)
    for _ in range(12):
        sk.run()
    ctx.synchronize()
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(12):
        sk.run()
    ctx.synchronize(); ctx.profile_enable(False)
    t_skin, n_skin = ctx.profile_get(api.K_SKIN_VERTICES)
    t_pose, n_pose = ctx.profile_get(api.K_POSE_PALETTE)
    print(f"{name:28s} skin {t_skin / n_skin:7.3f} ms per launch = {t_skin / n_skin * 1e9 / (n_inst * n_verts):6.3f} ms per 1e9 vertices   pose palette {1e3 * t_pose / max(n_pose, 1):7.1f} us", flush=True)
