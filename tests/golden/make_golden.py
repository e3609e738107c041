"""Generates tests/golden/*.npz from the REFERENCE's own object code (oracle/_ref/liblmx_ref.so).

Run in the development container, where /root/reference exists:

    python tests/golden/make_golden.py

The fixtures are committed; `/root/reference` does not exist on the GPU box, so tests only ever read the .npz files.
Each file stores the inputs next to the outputs, so a fixture never depends on a random generator's stream.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from lumixengine_amd import scenes  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tests import helpers as H  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def cull_fixture(ref, sc, frusta):
    cs = ref.culling_system()
    cs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    out = {"entity": sc["entity"], "type": sc["type"], "pos": sc["pos"], "radius": sc["radius"], "frusta": frusta, "n_cells": np.array([cs.cell_count()])}
    for f in range(len(frusta)):
        ids, types, _ = cs.cull(frusta[f : f + 1])
        order = np.lexsort((ids, types))
        out[f"vis_ids_{f}"] = ids[order]
        out[f"vis_types_{f}"] = types[order]
    return out


def main():
    po.build()
    ref = po.Oracle("reference")
    print(ref.describe())
    frusta = H.frusta(ref)
    np.savez_compressed(os.path.join(OUT, "frusta.npz"), names=np.array([c[0] for c in H.CAMERAS]), frusta=frusta, cascades=H.cascade_frusta(ref))

    np.savez_compressed(os.path.join(OUT, "cull_edge.npz"), **cull_fixture(ref, H.edge_case_scene(), frusta))
    np.savez_compressed(os.path.join(OUT, "cull_mixed.npz"), **cull_fixture(ref, H.mixed_scene(), np.concatenate([frusta, H.cascade_frusta(ref)])))
    # config 1 (100 k static MESH entities, 1 perspective frustum): inputs are regenerated from the seed by the test,
    # guarded by a checksum; all 7 cameras are stored
    sc = scenes.cull_scene(100_000, 3000.0, seed=1)
    fx = cull_fixture(ref, sc, frusta)
    for k in ("entity", "type", "pos", "radius"):
        del fx[k]
    fx["pos_sum"] = np.array([sc["pos"].sum(), np.abs(sc["pos"]).sum()])
    fx["radius_sum"] = np.array([float(sc["radius"].astype(np.float64).sum())])
    np.savez_compressed(os.path.join(OUT, "cull_config1.npz"), **fx)

    # transforms
    rng = np.random.default_rng(9)
    a = scenes.random_transforms(rng, 256, 1.0e6)
    b = scenes.random_transforms(rng, 256, 50.0)
    h = scenes.hierarchy_fans(6, 3, 4, seed=3)
    n = len(h["parent"])
    w = ref.world(n)
    roots = np.flatnonzero(h["parent"] < 0).astype(np.int32)
    kids = np.flatnonzero(h["parent"] >= 0).astype(np.int32)
    w.init_transforms(roots, h["local"][roots])
    w.set_parents(h["parent"][kids], kids)
    w.set_local_transforms(kids, h["local"][kids])
    locals_ = w.get_local_transforms()
    locals_[roots] = w.get_transforms()[roots]  # a root's hierarchy record holds indeterminate memory in the reference (world.cpp:681-687): not part of the fixture
    world0 = w.get_transforms()
    new_root = scenes.random_transforms(rng, len(roots), 4000.0)
    w.set_transforms(roots, new_root)
    np.savez_compressed(
        os.path.join(OUT, "transforms.npz"), a=a, b=b, compose=ref.compose(a, b), compute_local=ref.compute_local(a, b), parent=h["parent"],
        locals=locals_, world0=world0, new_root=new_root, world1=w.get_transforms(),
    )

    # pose / palette / skin
    sk = scenes.skeleton(64, seed=4)
    pos, rot = scenes.relative_poses(4, 64, seed=5)
    verts, skin = scenes.skinned_mesh(500, 64, seed=6)
    inv = ref.invert_bind(sk["bind"])
    apos, arot = ref.pose_compute_absolute(pos, rot, sk["parents"], sk["first_nonroot"])
    pal = ref.skin_matrices(apos, arot, inv)
    out = ref.evaluate_skin(verts, skin, pal)
    dq = ref.dual_quats(apos, arot, inv)
    np.savez_compressed(
        os.path.join(OUT, "skin.npz"), parents=sk["parents"], bind=sk["bind"], first_nonroot=np.array([sk["first_nonroot"]]), rel_pos=pos, rel_rot=rot,
        verts=verts, skin=skin, inv_bind=inv, abs_pos=apos, abs_rot=arot, palette=pal, skinned=out, dual_quats=dq,
    )
    # bone attachments: updateBoneAttachment on the reference's own compose / LocalRigidTransform::operator*
    n = 64
    att_parent = scenes.random_transforms(rng, n, 5000.0)
    att_rel = np.zeros(n, po.LOCAL_RIGID)
    att_rel["pos"], att_rel["rot"] = rng.uniform(-1, 1, size=(n, 3)), scenes.random_unit_quats(rng, n)
    att_scale = rng.uniform(0.5, 2.0, size=(n, 3)).astype(np.float32)
    att_bone = rng.integers(0, 64, size=n).astype(np.uint32)
    att_inst = rng.integers(0, 4, size=n).astype(np.uint32)
    np.savez_compressed(
        os.path.join(OUT, "attach.npz"), parent=att_parent, relative=att_rel, scale=att_scale, bone=att_bone, instance=att_inst,
        result=ref.bone_attachment(att_parent, apos[att_inst, att_bone], arot[att_inst, att_bone], att_rel, att_scale),
    )
    # Pose::blend on the reference's own nlerp: the skin golden's relative poses blended with a second set
    b_pos, b_rot = scenes.relative_poses(4, 64, seed=55)
    blended = ref.pose_blend(pos, rot, b_pos, b_rot, 0.35)
    np.savez_compressed(os.path.join(OUT, "blend.npz"), rhs_pos=b_pos, rhs_rot=b_rot, weight=np.array([0.35], np.float32), pos=blended[0], rot=blended[1])
    # serialized World compressed by the LZ4 the reference vendors (external/lz4/lz4.c in oracle/_ref); the record layout is the
    # restated writer of tests/helpers.py
    from tests.test_world_blob import make_world, ref_lz4
    wn, wents, wworld, whier, wh = make_world(seed=5)
    wdata, wblob = H.write_world_blob(wents, wworld[wents], whier, names=[(wents[3], "crate"), (wents[5], "lamp")], compress=ref_lz4(ref))
    open(os.path.join(OUT, "world_blob.bin"), "wb").write(wdata)
    # expected arrays from the INPUTS of the writer (not from the parser under test)
    wn_slots = max(wents) + 1
    wparent = np.full(wn_slots, -1, np.int32)
    wtr = np.zeros(wn_slots, po.TRANSFORM)
    wtr["rot"][:, 3] = 1.0
    wtr["scale"] = 1.0
    wwtr = wtr.copy()
    wvalid = np.zeros(wn_slots, np.uint8)
    for e in wents:
        wvalid[e] = 1
        wwtr[e] = wworld[e]
        if wh["parent"][e] >= 0:
            wparent[e] = wh["parent"][e]
            wtr[e] = wh["local"][e]
        else:
            wtr[e] = wworld[e]
    # the reference's own World::serialize (engine/world.cpp compiled in place into oracle/_ref) on a real World
    from tests.test_world_blob import make_reference_world
    rw, rh, rgone = make_reference_world(ref)
    open(os.path.join(OUT, "world_blob_ref.bin"), "wb").write(rw.serialize(1))
    ralive = np.ones(len(rh["parent"]), np.uint8)
    ralive[rgone] = 0
    np.savez_compressed(os.path.join(OUT, "world_blob_ref.npz"), parent=np.where(ralive.astype(bool), rh["parent"], -1).astype(np.int32), alive=ralive,
                        world=rw.get_transforms(), local=np.where((rh["parent"] >= 0)[:, None], rw.get_local_transforms().view(np.uint8).reshape(len(ralive), -1),
                                                                rw.get_transforms().view(np.uint8).reshape(len(ralive), -1)).view(po.TRANSFORM).reshape(-1))
    import struct
    unc, comp = struct.unpack_from("<II", wdata, len(wdata) - len(ref_lz4(ref)(wblob)) - 8)
    np.savez_compressed(os.path.join(OUT, "world_blob.npz"), parent=wparent, transforms=wtr, world=wwtr, valid=wvalid, sizes=np.array([unc, comp]))
    # ---- createSortKeys: outputs of the reference's own function (oracle/ref/slice_sort_keys.py + keys_shim.cpp in oracle/_ref), one worker
    from lumixengine_amd import api as lapi  # dtypes only (no GPU needed)

    n = 4000
    r2 = np.random.default_rng(77)
    types = np.where(r2.random(n) < 0.85, 0, np.where(r2.random(n) < 0.5, 1, 3)).astype(np.uint8)
    ks = scenes.keys_scene(n, types, seed=78)
    kpos = r2.uniform(-3000, 3000, size=(n, 3))
    vis = r2.random(n) < 0.5
    ids = {t: r2.permutation(np.flatnonzero(vis & (types == t))).astype(np.int32) for t in (0, 1, 3)}
    kv = lapi.keys_view(camera_pos=(120.5, -30.25, 900.0), lod_ref_point=(100.0, 0.0, 800.0), time_delta=0.5, frame_number=8, lod_multiplier=2.5,
                        layer_to_bucket=ks["layer_to_bucket"], bucket_depth_sorted=ks["bucket_depth_sorted"])
    got = ref.create_sort_keys(kv, ks["max_sort_key"], ids[0], ids[1], ids[3], ks, kpos)
    order = np.lexsort((got["values"], got["keys"]))
    np.savez_compressed(
        os.path.join(OUT, "sort_keys.npz"), types=types, pos=kpos, mesh_ids=ids[0], decal_ids=ids[1], curve_ids=ids[3], kv=kv,
        keys=got["keys"][order], values=got["values"][order], group_offsets=got["group_offsets"], poses=np.sort(got["poses"]), dirty=np.sort(got["dirty"]),
        lod=got["lod"], pose_frame=got["pose_frame"], scene_seed=np.array([78]),
    )
    # animation sampling: outputs of the reference's own sampler (oracle/ref/slice_animation.py + anim_shim.cpp in oracle/_ref)
    sk24 = scenes.skeleton(24, seed=4)
    anim = scenes.animation(24, 12, 30.0, seed=31)
    times = np.array([0, 5000, 13000, 1 << 20], np.uint32)
    apos2, arot2, nt = ref.update_animables([anim], [0, 0, 0, 0], times, 0.25, 0.4, sk24["bind"])
    bpos, brot, nt1 = ref.update_animables([anim], [0, 0, 0, 0], times, 1 / 60, 1.0, sk24["bind"])
    np.savez_compressed(os.path.join(OUT, "animation.npz"), times=times, pos_w04=apos2, rot_w04=arot2, times_w04=nt, pos_w1=bpos, rot_w1=brot, times_w1=nt1,
                        anim_seed=np.array([31]), skeleton_seed=np.array([4]))
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
