"""Animation sampling (AnimationModuleImpl::updateAnimable, animation_module.cpp:439-472; AnimationSampler, animation.cpp:29-204;
simd_nlerp, core/simd_math.h:107-123): oracle vs an independent pure-Python restatement (CPU), HIP path vs the oracle (GPU),
bit for bit. Pinned: tests/test_oracle_vs_ref.py::test_animation_sampling_bit_exact compares the oracle with the reference's own
sampler (cut out of animation.cpp / simd.h / simd_math.h into oracle/_ref at build time), and tests/golden/animation.npz is that
sampler's output."""
import numpy as np
import pytest

from lumixengine_amd import api, scenes
from tests import helpers as H

f32 = np.float32
ONE_SECOND = 1 << 15


def py_nlerp(q1, q2, t):
    q1, q2, t = [f32(x) for x in q1], [f32(x) for x in q2], f32(t)
    inv = f32(f32(1.0) - t)
    p = [f32(a * b) for a, b in zip(q1, q2)]
    d = f32(f32(p[0] + p[1]) + f32(p[2] + p[3]))
    if d < 0:
        t = f32(-t)
    q = [f32(f32(a * inv) + f32(b * t)) for a, b in zip(q1, q2)]
    s = [f32(x * x) for x in q]
    l = f32(f32(1.0) / np.sqrt(f32(f32(s[0] + s[1]) + f32(s[2] + s[3]))))
    return np.array([f32(x * l) for x in q], f32)


def py_bits(stream, byte_offset):
    return int.from_bytes(bytes(stream[byte_offset : byte_offset + 8]), "little")


def py_update_animable(a, time, time_delta, weight, rel):
    pos = np.array(rel["pos"], f32)
    rot = np.array(rel["rot"], f32)
    if a is None:
        return pos, rot, time
    bones = [int(x) for k in ("const_translations", "translations", "const_rotations", "rotations") for x in a[k]["bone_index"]]
    if max(bones) < len(pos):
        use_w = f32(weight) < f32(0.9999)
        w, invw = f32(weight), f32(f32(1.0) - f32(weight))
        sample = f32(time / float(ONE_SECOND) * float(a["fps"]))
        hi = f32(f32(a["frame_count"]) - f32(0.00001))
        sample = min(max(sample, f32(0)), hi)
        idx = int(sample)
        t = f32(sample - f32(idx))

        def blend_pos(b, v):
            pos[b] = [f32(f32(p * invw) + f32(x * w)) for p, x in zip(pos[b], v)] if use_w else v

        for c in a["const_translations"]:
            blend_pos(int(c["bone_index"]), c["value"])
        for i, tr in enumerate(a["translations"]):
            vals = []
            for frame in (idx, idx + 1):
                if i == a["root_translation_track"]:
                    vals.append(a["root_pose_translations"][frame])
                    continue
                off = a["translations_frame_size_bits"] * frame + int(tr["offset_bits"])
                tmp = py_bits(a["translation_stream"], off // 8) >> (off & 7)
                v = []
                for c in range(3):
                    nb = int(tr["bitsizes"][c])
                    v.append(f32(float(tr["min"][c]) + float(tr["to_range"][c]) * float(tmp & ((1 << nb) - 1))))
                    tmp >>= nb
                vals.append(v)
            invt = f32(f32(1.0) - t)
            blend_pos(int(tr["bone_index"]), [f32(f32(x * invt) + f32(y * t)) for x, y in zip(vals[0], vals[1])])

        def blend_rot(b, v):
            rot[b] = py_nlerp(rot[b], v, w) if use_w else v

        for c in a["const_rotations"]:
            blend_rot(int(c["bone_index"]), c["value"])
        for i, tr in enumerate(a["rotations"]):
            if i == a["root_rotation_track"]:
                v = py_nlerp(a["root_pose_rotations"][idx], a["root_pose_rotations"][idx + 1], t)
            else:
                qs = []
                for frame in (idx, idx + 1):
                    off = a["rotations_frame_size_bits"] * frame + int(tr["offset_bits"])
                    packed = (py_bits(a["rotation_stream"], off // 8) >> (off & 7)) & ((1 << 64) - 1)
                    neg = packed & 1
                    packed >>= 1
                    v3 = []
                    for c in range(3):
                        nb = int(tr["bitsizes"][c])
                        v3.append(f32(tr["min"][c] + f32(tr["to_range"][c] * f32(packed & ((1 << nb) - 1)))))
                        packed >>= nb
                    dot = f32(f32(f32(v3[0] * v3[0]) + f32(v3[1] * v3[1])) + f32(v3[2] * v3[2]))
                    rest = f32(f32(1) - dot)
                    skipped = f32(np.sqrt(max(rest, f32(0))) * f32(-1 if neg else 1))
                    q = list(v3)
                    q.insert(int(tr["skipped_channel"]), skipped)
                    qs.append(q)
                v = py_nlerp(qs[0], qs[1], t)
            blend_rot(int(tr["bone_index"]), v)
    l = int(a["length"])
    if time_delta > 0:
        nt = (time + int(f32(f32(time_delta) * f32(ONE_SECOND)))) % l
    else:
        dt = int(f32(f32(-time_delta) * f32(ONE_SECOND))) % l
        nt = (time + l - dt) % l
    return pos, rot, nt


def test_nlerp_matches_second_restatement_and_normalises(oracle_port):
    rng = np.random.default_rng(2)
    q1, q2 = scenes.random_unit_quats(rng, 500), scenes.random_unit_quats(rng, 500)
    t = rng.random(500).astype(f32)
    got = oracle_port.nlerp(q1, q2, t)
    want = np.array([py_nlerp(a, b, x) for a, b, x in zip(q1, q2, t)])
    assert H.bits_equal(got, want)
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-6)
    assert H.bits_equal(oracle_port.nlerp(q1[:1], q1[:1], [0.3]), oracle_port.nlerp(q1[:1], q1[:1], [0.7]))  # same endpoints: t is irrelevant
    assert ((q1 * q2).sum(axis=1) < 0).any()  # the short-way-round branch is exercised


@pytest.mark.parametrize("weight,dt", [(1.0, 1 / 60), (0.4, 0.25), (1.0, -0.4)])
def test_oracle_update_animable_matches_second_restatement(oracle_port, weight, dt):
    sk = scenes.skeleton(24, seed=4)
    anims = [scenes.animation(24, 12, 30.0, seed=31), scenes.animation(24, 7, 24.0, seed=32, root_motion=False), scenes.animation(40, 5, 30.0, seed=33, bone_limit=40)]
    pick = [0, 1, 2, -1, 0, 1]
    times = [0, 5000, 100, 7, 13000, 1 << 20]
    pos, rot, nt = oracle_port.update_animables(anims, pick, times, dt, weight, sk["bind"])
    for i, k in enumerate(pick):
        wp, wr, wt = py_update_animable(anims[k] if k >= 0 else None, times[i], dt, weight, sk["bind"])
        assert H.bits_equal(pos[i], wp) and H.bits_equal(rot[i], wr), f"instance {i}"
        assert int(nt[i]) == wt
    assert H.bits_equal(pos[2], np.array(sk["bind"]["pos"])) and H.bits_equal(pos[3], np.array(sk["bind"]["pos"]))  # mismatched skeleton / no animation
    assert not H.bits_equal(pos[0], np.array(sk["bind"]["pos"]))


def load_fixture():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "animation.npz"))
    return g, scenes.animation(24, 12, 30.0, seed=int(g["anim_seed"][0])), scenes.skeleton(24, seed=int(g["skeleton_seed"][0]))


def test_oracle_reproduces_committed_fixture(oracle_port):
    """tests/golden/animation.npz is output of the reference's sampler compiled into oracle/_ref (tests/golden/make_golden.py)."""
    g, anim, sk = load_fixture()
    for tag, w, dt in (("w04", 0.4, 0.25), ("w1", 1.0, 1 / 60)):
        pos, rot, nt = oracle_port.update_animables([anim], [0] * 4, g["times"], dt, w, sk["bind"])
        assert H.bits_equal(pos, g["pos_" + tag]) and H.bits_equal(rot, g["rot_" + tag]) and np.array_equal(nt, g["times_" + tag])


@pytest.mark.gpu
def test_gpu_animation_matches_committed_fixture(gpu_ctx):
    g, anim, s = load_fixture()
    sk = api.Skinning(gpu_ctx)
    model = sk.addModel(s["parents"], s["bind"], s["first_nonroot"])
    mesh = sk.addMesh(*scenes.skinned_mesh(30, 24, seed=3))
    sk.setInstances([model] * 4, [mesh] * 4)
    sk.setModelPose(model, s["bind"])
    aid = sk.addAnimation(anim)
    for tag, w, dt in (("w04", 0.4, 0.25), ("w1", 1.0, 1 / 60)):
        sk.setAnimables([aid] * 4, g["times"])
        sk.setAnimWeight(w)
        sk.updateAnimables(dt)
        for i in range(4):
            pos, rot = sk.readRelativePose(i)
            assert H.bits_equal(pos, g["pos_" + tag][i]) and H.bits_equal(rot, g["rot_" + tag][i])
        assert np.array_equal(sk.readTimes(), g["times_" + tag])
    sk.setAnimWeight(1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("weight,dt", [(1.0, 1 / 60), (0.4, 0.25), (1.0, -0.4)])
def test_gpu_animation_matches_oracle(gpu_ctx, live_oracle, weight, dt):
    """Three consecutive frames of animation -> absolute pose -> palette on the device, every stage bit-exact."""
    oracle_port = live_oracle
    skel = [scenes.skeleton(64, seed=4), scenes.skeleton(100, seed=24)]
    anims = [scenes.animation(64, 30, 30.0, seed=41), scenes.animation(64, 9, 24.0, seed=42, root_motion=False), scenes.animation(100, 20, 60.0, seed=43),
             scenes.animation(100, 6, 30.0, seed=44, bone_limit=90)]
    sk = api.Skinning(gpu_ctx)
    models = [sk.addModel(s["parents"], s["bind"], s["first_nonroot"]) for s in skel]
    meshes = [sk.addMesh(*scenes.skinned_mesh(64, 64, seed=6)), sk.addMesh(*scenes.skinned_mesh(64, 100, seed=7))]
    rng = np.random.default_rng(9)
    inst_model = np.array([0] * 40 + [1] * 25 + [0] * 3)
    # instance -> animation: model 0 may play 0, 1, 3 (3 reaches bone 89: skeleton mismatch, pose untouched) or nothing; model 1 plays 2, 3 or 1
    choice = {0: [0, 1, 3, api_none()], 1: [2, 3, 1, api_none()]}
    anim_of = np.array([choice[m][rng.integers(0, 4)] for m in inst_model], np.uint32)
    times = rng.integers(0, 3 * ONE_SECOND, size=len(inst_model)).astype(np.uint32)
    sk.setInstances([models[m] for m in inst_model], [meshes[m] for m in inst_model])
    rel = [s["bind"] for s in skel]  # any rigid transforms serve as Model::Bone::relative_transform
    for m in range(2):
        sk.setModelPose(models[m], rel[m])
    anim_ids = [sk.addAnimation(a) for a in anims]
    # the context is shared between tests: ids returned by the library are used, not assumed
    sk.setAnimables(np.array([anim_ids[k] if k != api_none() else api_none() for k in anim_of], np.uint32), times)
    sk.setAnimWeight(weight)
    sk.setMode(True)
    for frame in range(3):
        sk.updateAnimables(dt)
        want_t = np.zeros_like(times)
        want_pose = []
        for i, m in enumerate(inst_model):
            k = int(anim_of[i]) if anim_of[i] != api_none() else -1
            wp, wr, nt = oracle_port.update_animables(anims, [k], [times[i]], dt, weight, rel[m])
            gp, gr = sk.readRelativePose(i)
            assert H.bits_equal(gp, wp[0]) and H.bits_equal(gr, wr[0]), f"frame {frame} instance {i} (animation {k})"
            want_t[i] = nt[0]
            want_pose.append((wp, wr))
        if frame == 2:  # the chain continues bit-exactly into the palette
            sk.run()
            for i in range(0, len(inst_model), 7):
                m = inst_model[i]
                apos, arot = oracle_port.pose_compute_absolute(want_pose[i][0], want_pose[i][1], skel[m]["parents"], skel[m]["first_nonroot"])
                pal = oracle_port.skin_matrices(apos, arot, oracle_port.invert_bind(skel[m]["bind"]))
                assert H.bits_equal(sk.readPalette(i), pal[0])
        assert np.array_equal(sk.readTimes(), want_t)
        times = want_t
    sk.setMode(False)
    sk.setAnimWeight(1.0)


@pytest.mark.gpu
def test_gpu_blend_stacks_match_oracle(gpu_ctx, live_oracle):
    """The Animator path on the device: lmx_anim_eval_blend_stacks (Model::getRelativePose + evalBlendStack's SAMPLE instructions,
    animation_module.cpp:602-636, controller.cpp:267-293) -> lmx_skin_run (Pose::computeAbsolute + palette), against the oracle doing
    the same sequence layer by layer (the reference oracle runs the reference's own getRelativePose / computeAbsolute code). Two
    skeletons (100 bones: two bone rounds per wave), clips that do not fit a skeleton, empty stacks, looped and clamped times."""
    from tests.test_oracle_vs_ref import random_blend_stacks
    skel = [scenes.skeleton(64, seed=5), scenes.skeleton(100, seed=25)]
    anims = [scenes.animation(64, 30, 30.0, seed=51), scenes.animation(64, 9, 24.0, seed=52, root_motion=False), scenes.animation(100, 20, 60.0, seed=53),
             scenes.animation(100, 6, 30.0, seed=54, bone_limit=90)]
    sk = api.Skinning(gpu_ctx)
    models = [sk.addModel(s["parents"], s["bind"], s["first_nonroot"]) for s in skel]
    meshes = [sk.addMesh(*scenes.skinned_mesh(64, 64, seed=6)), sk.addMesh(*scenes.skinned_mesh(64, 100, seed=7))]
    rng = np.random.default_rng(19)
    inst_model = np.array([0] * 30 + [1] * 20 + [0] * 3)
    sk.setInstances([models[m] for m in inst_model], [meshes[m] for m in inst_model])
    for m in range(2):
        sk.setModelPose(models[m], skel[m]["bind"])
    ids = [sk.addAnimation(a) for a in anims]
    fits = {0: [0, 1, 3], 1: [0, 1, 2, 3]}  # clip 3 reaches bone 89: skipped on the 64-bone skeleton; clips 0 and 1 animate bones 0..63 of the 100
    stacks = [random_blend_stacks(rng, 1, fits[m])[0] for m in inst_model]
    stacks[2] = []
    stacks[3] = [(0, 1.0, anims[0]["length"], False), (1, 0.25, 7 * anims[1]["length"] + 3, True), (3, 0.5, 100, False)]
    for frame in range(2):
        sk.evalBlendStacks([[(ids[k], w, t, lp) for (k, w, t, lp) in st] for st in stacks])
        for i, m in enumerate(inst_model):
            wp, wr = live_oracle.update_animators(anims, [stacks[i]], skel[m]["bind"])
            gp, gr = sk.readRelativePose(i)
            assert H.bits_equal(gp, wp[0]) and H.bits_equal(gr, wr[0]), f"frame {frame} instance {i}: {stacks[i]}"
        sk.run()
        for i in range(0, len(inst_model), 3):
            m = inst_model[i]
            ap, ar = live_oracle.update_animators(anims, [stacks[i]], skel[m]["bind"], skel[m]["parents"], skel[m]["first_nonroot"])
            pal = live_oracle.skin_matrices(ap, ar, live_oracle.invert_bind(skel[m]["bind"]))
            assert H.bits_equal(sk.readPalette(i), pal[0])
        # the next frame's instructions: every clock moved on by a 60 Hz tick, as the controller's nodes would emit them
        stacks = [[(k, w, t + ONE_SECOND // 60, lp) for (k, w, t, lp) in st] for st in stacks]


@pytest.mark.gpu
def test_gpu_blend_stacks_reject_bad_arguments(gpu_ctx):
    s = scenes.skeleton(16, seed=2)
    sk = api.Skinning(gpu_ctx)
    model = sk.addModel(s["parents"], s["bind"], s["first_nonroot"])
    mesh = sk.addMesh(*scenes.skinned_mesh(16, 16, seed=3))
    sk.setInstances([model] * 2, [mesh] * 2)
    sk.setModelPose(model, s["bind"])
    aid = sk.addAnimation(scenes.animation(16, 5, 30.0, seed=4))
    with pytest.raises(api.LumixError):
        sk.evalBlendStacks([[(aid, 1.0, 0, True)]])  # one stack for two instances
    with pytest.raises(api.LumixError):
        sk.evalBlendStacks([[(aid + 1000, 1.0, 0, True)], []])
    with pytest.raises(api.LumixError):
        sk.evalBlendStacks([[(aid, 1.5, 0, True)], []])
    sk.evalBlendStacks([[], []])
    gp, gr = sk.readRelativePose(1)
    assert H.bits_equal(gp, s["bind"]["pos"]) and H.bits_equal(gr, s["bind"]["rot"])


def api_none():
    return 0xFFFFFFFF
