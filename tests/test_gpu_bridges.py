"""WorldSync and PoseBridge (lumixengine_amd/host/), the C++ hand-off code mi355_plugin.cpp's module uses, driven by a C++ program
against the in-memory mocks of lumix_compat.h and compared with the CPU oracle: World mirror -> staged writes -> GPU propagation ->
transforms written back in World::getTransforms() order; lockPose -> relative poses -> pose / palette / skin on the GPU ->
absolute poses stored back through unlockPose."""
import os
import subprocess

import numpy as np
import pytest

from lumixengine_amd import api, scenes
from tests import helpers as H
from tests.test_gpu_world_skin import _depths, oracle_world

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_bridges.cpp")
EXE = os.path.join(ROOT, "tests", "_build", "test_bridges")
HOST = os.path.join(ROOT, "lumixengine_amd", "host")


def build_exe():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    deps = [SRC] + [os.path.join(HOST, h) for h in ("world_sync.h", "pose_bridge.h")] + [os.path.join(ROOT, "tests", "cpp", "lumix_compat.h")]
    if os.path.exists(EXE) and all(os.path.getmtime(d) <= os.path.getmtime(EXE) for d in deps):
        return
    lib_dir = os.path.join(ROOT, "lumixengine_amd")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + HOST, "-I" + os.path.join(ROOT, "tests", "cpp"), SRC, "-o", EXE, "-L" + lib_dir, "-llumix_mi355",
                    "-Wl,-rpath," + lib_dir, "-pthread"], check=True)


def test_bridges_compile_and_link():
    from lumixengine_amd import build

    if not os.path.exists(api.LIB_PATH):
        build.build()
    build_exe()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_bridges_match_oracle(tmp_path, oracle_port):
    build_exe()
    h = scenes.hierarchy_fans(15, 5, 4, seed=41)
    ow, roots, kids = oracle_world(oracle_port, h)
    parent = h["parent"]
    n = len(parent)
    depth = _depths(parent)
    rng = np.random.default_rng(8)
    picked = rng.permutation(n)[: n // 2].astype(np.int32)
    how = rng.integers(0, 2, size=len(picked))
    tr = scenes.random_transforms(rng, len(picked), 40.0)
    loc = (how == 0) | (parent[picked] < 0)
    world0, local0 = ow.get_transforms(), ow.get_local_transforms()
    for d in range(int(depth.max()) + 1):
        for i in np.flatnonzero(depth[picked] == d):
            e = picked[i : i + 1]
            (ow.set_local_transforms if loc[i] and parent[e[0]] >= 0 else ow.set_transforms)(e, tr[i : i + 1])
    n_bones, n_verts, n_inst = 48, 900, 6
    s = scenes.skeleton(n_bones, seed=14)
    verts, skin = scenes.skinned_mesh(n_verts, n_bones, seed=16)
    pos, rot = scenes.relative_poses(n_inst, n_bones, seed=15)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(np.array([n, int(loc.sum()), int((~loc).sum()), n_bones, n_verts, n_inst, 0, 0], np.uint32).tobytes())
        for a in (parent.astype(np.int32), world0, local0, picked[loc], tr[loc], picked[~loc], tr[~loc], s["parents"].astype(np.int16), s["bind"], verts.astype(np.float32), skin,
                  pos.astype(np.float32), rot.astype(np.float32)):
            f.write(np.ascontiguousarray(a).tobytes())
    r = subprocess.run([EXE, str(inp), str(outp)], capture_output=True, text=True)
    assert r.returncode == 0, f"rc {r.returncode}: {r.stdout}{r.stderr}"
    raw = open(outp, "rb").read()
    tsz = api.TRANSFORM.itemsize
    got_world = np.frombuffer(raw[: n * tsz], api.TRANSFORM)
    got_local = np.frombuffer(raw[n * tsz : 2 * n * tsz], api.TRANSFORM)
    assert H.transforms_bits_equal(got_world, ow.get_transforms())
    assert H.transforms_bits_equal(got_local[kids], ow.get_local_transforms()[kids])
    cur = 2 * n * tsz
    abs_pos = np.frombuffer(raw[cur : cur + n_inst * n_bones * 12], np.float32).reshape(n_inst, n_bones, 3)
    cur += n_inst * n_bones * 12
    abs_rot = np.frombuffer(raw[cur : cur + n_inst * n_bones * 16], np.float32).reshape(n_inst, n_bones, 4)
    cur += n_inst * n_bones * 16
    skinned = np.frombuffer(raw[cur:], np.float32).reshape(n_inst, n_verts, 3)
    want_pos, want_rot = oracle_port.pose_compute_absolute(pos, rot, s["parents"], s["first_nonroot"])
    assert H.bits_equal(abs_pos, want_pos) and H.bits_equal(abs_rot, want_rot)
    want = oracle_port.evaluate_skin(verts, skin, oracle_port.skin_matrices(want_pos, want_rot, oracle_port.invert_bind(s["bind"])))
    for i in range(n_inst):
        assert np.allclose(skinned[i], want[i], rtol=1e-5, atol=1e-5 * float(np.abs(want[i]).max()))
