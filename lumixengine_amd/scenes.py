"""Seeded synthetic scenes for the BASELINE.json configs. Pure numpy, no oracle, no GPU.

All generators are deterministic in ``seed`` (numpy PCG64) so the CPU oracle and the HIP path see identical
inputs. Frusta are *not* built here: they come from the reference's own frustum construction (oracle) in tests
or from ``lumixengine_amd.frustum`` (the host mirror of core/geometry.cpp) in the product.
"""
from __future__ import annotations

import numpy as np

TRANSFORM = np.dtype([("pos", "<f8", 3), ("rot", "<f4", 4), ("scale", "<f4", 3), ("_pad", "<f4")], align=True)
LOCAL_RIGID = np.dtype([("pos", "<f4", 3), ("rot", "<f4", 4)], align=True)
SKIN = np.dtype([("weights", "<f4", 4), ("indices", "<i2", 4)], align=True)


def cull_scene(n: int, half_extent: float, seed: int = 1, big_fraction: float = 0.001, mixed_types: bool = False):
    """n bounding spheres in the cube [-half_extent, half_extent]^3 (fp64 positions).

    radii: log-uniform in [0.5, 50]; ``big_fraction`` of them in (300, 900] ("big" cells, culling_system.cpp:140).
    types: all MESH, or 90 % MESH / 5 % LOCAL_LIGHT / 5 % DECAL when ``mixed_types`` (config 5).
    Returns dict(entity int32[n], type uint8[n], pos float64[n,3], radius float32[n]).
    """
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-half_extent, half_extent, size=(n, 3))
    radius = np.exp(rng.uniform(np.log(0.5), np.log(50.0), size=n)).astype(np.float32)
    n_big = int(n * big_fraction)
    if n_big:
        idx = rng.choice(n, size=n_big, replace=False)
        radius[idx] = rng.uniform(300.0, 900.0, size=n_big).astype(np.float32) + np.float32(1e-3)
    if mixed_types:
        r = rng.random(n)
        type_ = np.where(r < 0.90, 0, np.where(r < 0.95, 2, 1)).astype(np.uint8)
    else:
        type_ = np.zeros(n, np.uint8)
    entity = np.arange(n, dtype=np.int32)
    return {"entity": entity, "type": type_, "pos": np.ascontiguousarray(pos), "radius": radius}


def all_test_radii(n: int) -> np.ndarray:
    """Radii of the roofline leg's scene: every sphere in (300, 330], so every cell is a "big" cell (culling_system.cpp:140,342-344),
    every cell is CELL_TEST and every sphere is fetched and tested (bench.py `all_test`, tests/golden/cull_bench_scenes.json)."""
    return np.random.default_rng(5).uniform(300.5, 330.0, size=n).astype(np.float32)


def slab_half_extent(n: int) -> float:
    """half side of slab_scene(n): a square of cells with ~10.2 spheres each"""
    return float(int(np.ceil(np.sqrt(n / 10.2))) * 150.0)


def slab_scene(n: int, seed: int = 2):
    """n spheres of NORMAL radii (0.5 .. 50) in one layer of culling cells (y in (10, 290)), ~10 per 300-unit cell like BASELINE config 2:
    under slab_frustum_kwargs() every occupied cell straddles the frustum's near and far plane, so every cell comes out CELL_TEST through
    the reference's AABB pre-tests (culling_system.cpp:342-363) - the config-2-faithful "every sphere is fetched and tested" case, where the
    all_test_radii() scene reaches it through the big-sphere shortcut (which skips the pre-tests)."""
    rng = np.random.default_rng(seed)
    half = slab_half_extent(n)
    pos = np.empty((n, 3))
    pos[:, 0] = rng.uniform(-half, half, n)
    pos[:, 1] = rng.uniform(10.0, 290.0, n)
    pos[:, 2] = rng.uniform(-half, half, n)
    radius = np.exp(rng.uniform(np.log(0.5), np.log(50.0), size=n)).astype(np.float32)
    return {"entity": np.arange(n, dtype=np.int32), "type": np.zeros(n, np.uint8), "pos": pos, "radius": radius, "half": half}


def slab_frustum_kwargs(half: float):
    """viewport_frustum(**kw): an orthographic camera above slab_scene looking straight down, as wide as the slab, its depth range the
    100 units between y = 200 and y = 100: no cell of the layer is inside it, every one intersects it."""
    return dict(is_ortho=True, ortho_size=half + 900.0, w=1024, h=1024, near=0.0, far=100.0, pos=(0.0, 200.0, 0.0), rot=(-0.70710678, 0.0, 0.0, 0.70710678))


def scaled_half_extent(n: int) -> float:
    """Half extent of the cube that keeps BASELINE config 2's density (10 M in +-15000: ~10 spheres per 300-unit cell) at n entities."""
    return 15000.0 * (n / 1e7) ** (1.0 / 3.0)


def config5_cascade_kwargs(n: int = 8):
    """viewport_frustum(**kw) arguments of the 8 ortho shadow-cascade frusta (2 light directions x 4 cascades, growing extents; the
    reference builds its cascades from split distances {0.1, 3, 10, 60, 150}, pipeline.cpp:734-827) that bench.py's config-5 legs and
    the 100 M parity test cull in one call. One definition for the bench, the tests and the golden generator."""
    return [dict(is_ortho=True, ortho_size=[30.0, 90.0, 400.0, 1500.0][k % 4] * 4.0, w=1024, h=1024, near=0.0, far=20000.0,
                 pos=(5.0 * k, 9000.0, -3.0 * k), rot=(-0.6, 0.25 * (k // 4), 0.0, 0.76)) for k in range(n)]


def random_unit_quats(rng, n: int) -> np.ndarray:
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32)


def random_transforms(rng, n: int, pos_extent: float, scale_lo=0.5, scale_hi=2.0) -> np.ndarray:
    t = np.zeros(n, TRANSFORM)
    t["pos"] = rng.uniform(-pos_extent, pos_extent, size=(n, 3))
    t["rot"] = random_unit_quats(rng, n)
    t["scale"] = rng.uniform(scale_lo, scale_hi, size=(n, 3)).astype(np.float32)
    return t


def hierarchy_chains(n_roots: int, depth: int, seed: int = 2, root_extent: float = 5000.0):
    """n_roots chains of ``depth`` nodes (root + depth-1 descendants). Entity ids: level-major (roots first).

    Returns dict(parent int32[n] (-1 for roots), local TRANSFORM[n] (for roots: the world transform)).
    """
    rng = np.random.default_rng(seed)
    n = n_roots * depth
    parent = np.full(n, -1, np.int32)
    for lvl in range(1, depth):
        parent[lvl * n_roots : (lvl + 1) * n_roots] = np.arange((lvl - 1) * n_roots, lvl * n_roots, dtype=np.int32)
    local = random_transforms(rng, n, 10.0)
    local["pos"][:n_roots] = rng.uniform(-root_extent, root_extent, size=(n_roots, 3))
    return {"parent": parent, "local": local, "n_roots": n_roots, "depth": depth}


def hierarchy_fans(n_roots: int, fanout: int, depth: int, seed: int = 3, root_extent: float = 5000.0):
    """n_roots trees where every node of level < depth-1 has ``fanout`` children (1k roots x 10 x 10 x 10 in config 3)."""
    rng = np.random.default_rng(seed)
    counts = [n_roots * fanout**lvl for lvl in range(depth)]
    n = sum(counts)
    parent = np.full(n, -1, np.int32)
    start = 0
    for lvl in range(1, depth):
        prev_start, prev_n = start, counts[lvl - 1]
        start += prev_n
        parent[start : start + counts[lvl]] = prev_start + np.arange(counts[lvl], dtype=np.int32) // fanout
    local = random_transforms(rng, n, 10.0)
    local["pos"][:n_roots] = rng.uniform(-root_extent, root_extent, size=(n_roots, 3))
    return {"parent": parent, "local": local, "n_roots": n_roots, "depth": depth}


def skeleton(n_bones: int = 64, seed: int = 4):
    """Random bone tree with parent < child (renderer/model.cpp:381-384), random rigid bind pose (model space)."""
    rng = np.random.default_rng(seed)
    parents = np.full(n_bones, -1, np.int16)
    for i in range(1, n_bones):
        parents[i] = rng.integers(max(0, i - 8), i)
    bind = np.zeros(n_bones, LOCAL_RIGID)
    bind["pos"] = rng.uniform(-1.0, 1.0, size=(n_bones, 3)).astype(np.float32)
    bind["rot"] = random_unit_quats(rng, n_bones)
    return {"parents": parents, "bind": bind, "first_nonroot": 1}


def relative_poses(n_instances: int, n_bones: int, seed: int = 5):
    """Per-instance *relative* poses: random small offsets + random unit rotations (fp32)."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-0.5, 0.5, size=(n_instances, n_bones, 3)).astype(np.float32)
    rot = random_unit_quats(rng, n_instances * n_bones).reshape(n_instances, n_bones, 4)
    return pos, rot


DISTINCT_MESH_POSES = 10_000  # relative_poses() draws positions before rotations: an instance's pose depends on how many instances are drawn with it


def distinct_mesh(base_verts, base_skin, i: int):
    """Mesh i of BASELINE config 3's distinct-mesh variant ("10 k skinned meshes, 64 bones, 10 k verts each": every instance its OWN 10 k-vertex
    mesh): the base mesh rotated by i vertices - different records at every vertex index, the same statistics. One definition for bench.py's leg,
    the -m gpu test and tests/golden/make_golden_skin_distinct.py."""
    return np.roll(base_verts, i, axis=0), np.roll(base_skin, i, axis=0)


DISTINCT_MESH_SAMPLE = (0, 1, 2, 777, 1499, 4999, 9999)  # instances whose skinned positions the reference's digests cover (tests/golden/skin_distinct.json)


def skinned_mesh(n_verts: int, n_bones: int = 64, seed: int = 6):
    """Vertex positions + Mesh::Skin{weights decoded from u16/65535 (model.cpp:542-550), 4 bone indices}."""
    rng = np.random.default_rng(seed)
    verts = rng.uniform(-1.0, 1.0, size=(n_verts, 3)).astype(np.float32)
    w = rng.random((n_verts, 4))
    w /= w.sum(axis=1, keepdims=True)
    w16 = np.round(w * 65535.0).astype(np.uint16)
    skin = np.zeros(n_verts, SKIN)
    skin["weights"] = (w16.astype(np.float32) / np.float32(65535.0)).astype(np.float32)
    skin["indices"] = rng.integers(0, n_bones, size=(n_verts, 4)).astype(np.int16)
    return verts, skin


def skinned_mesh_character(n_verts: int, n_bones: int = 52, seed: int = 6, run: int = 190, second_every: int = 6):
    """A mesh with the skinning statistics of a real character instead of skinned_mesh()'s worst case: consecutive vertices follow one
    bone (`run` vertices per bone: a limb at a time), every `second_every`-th vertex is also influenced by the neighbouring bone,
    influences are sorted by weight and zero-padded. The reference's demo character (demo/models/ybot/ybot.fbx, read with the OpenFBX
    the reference vendors: tools/fbx_skin_stats.cpp) has 52 bones, 1.0-1.2 influences per control point, and a tile of 5120 consecutive
    control points touches 13.5-15.7 bones on average, 24-27 at most - this generator sits at the upper end (27-28 per tile)."""
    rng = np.random.default_rng(seed)
    verts = rng.uniform(-1.0, 1.0, size=(n_verts, 3)).astype(np.float32)
    v = np.arange(n_verts)
    b0 = (v // run) % n_bones
    b1 = (b0 + 1) % n_bones
    two = (v % second_every) == 0
    w1 = np.where(two, rng.uniform(0.1, 0.5, size=n_verts), 0.0)
    w16 = np.stack([np.round((1.0 - w1) * 65535.0), np.round(w1 * 65535.0), np.zeros(n_verts), np.zeros(n_verts)], axis=1)
    skin = np.zeros(n_verts, SKIN)
    skin["weights"] = (w16.astype(np.float32) / np.float32(65535.0)).astype(np.float32)
    skin["indices"][:, 0] = b0
    skin["indices"][:, 1] = np.where(two, b1, 0)
    return verts, skin


def keys_scene(n_entities: int, types: np.ndarray, seed: int = 11, n_models: int = 6, max_sort_key: int = 63, meshes_per_lod=(1, 4), moved_fraction: float = 0.2):
    """Model-instance / material tables for createSortKeys over `n_entities` entities whose renderable types are `types`:
    models with 1-4 LODs of 1-3 meshes (some skinned), per-entity material spans, LOD state in [0, 4], MOVED / dirty flags.
    A mesh sort key identifies (mesh, material), so every key maps to one layer (pipeline.cpp:3958-3968 relies on it)."""
    from .api import KEYS_MODEL, MESH_MATERIAL
    rng = np.random.default_rng(seed)
    models = np.zeros(n_models, KEYS_MODEL)
    mesh_types = []
    for m in range(n_models):
        n_lods = int(rng.integers(1, 5))
        dist = np.sort(rng.uniform(50.0, 4000.0, size=n_lods).astype(np.float32)) ** 2
        models["lod_distances"][m] = np.finfo(np.float32).max  # Model::Model, model.cpp:99
        models["lod_indices"][m]["from"], models["lod_indices"][m]["to"] = 0, -1
        first = len(mesh_types)
        k = 0
        for lod in range(n_lods):
            c = int(rng.integers(meshes_per_lod[0], meshes_per_lod[1]))
            models["lod_indices"][m][lod] = (k, k + c - 1)
            models["lod_distances"][m][lod] = dist[lod]
            k += c
        skinned = rng.random() < 0.3
        mesh_types += [1 if skinned and rng.random() < 0.8 else 0 for _ in range(k)]
        models["first_mesh"][m], models["mesh_count"][m] = first, k
    n_layers = 6
    key_layer = rng.integers(0, n_layers, size=max_sort_key + 1).astype(np.uint8)  # sort key -> layer
    model = np.where(types == 0, rng.integers(0, n_models, size=n_entities), -1).astype(np.int32)
    counts = np.where(model >= 0, models["mesh_count"][np.maximum(model, 0)], 0).astype(np.uint32)
    material_offset = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint32)
    mm = np.zeros(int(counts.sum()), MESH_MATERIAL)
    mm["sort_key"] = rng.integers(0, max_sort_key + 1, size=len(mm))
    mm["layer"] = key_layer[mm["sort_key"]]
    lod = rng.integers(0, 5, size=n_entities).astype(np.float32)
    frac = rng.random(n_entities) < 0.3
    lod = np.where(frac, np.minimum(lod + rng.random(n_entities).astype(np.float32), np.float32(4.0)), lod).astype(np.float32)
    flags = (rng.random(n_entities) < moved_fraction).astype(np.uint8) * 8 | 6  # MOVED | VALID | ENABLED
    dirty = (rng.random(n_entities) < 0.05).astype(np.uint8)
    pose_frame = np.where(rng.random(n_entities) < 0.5, 7, 6).astype(np.uint32)  # half already processed in frame 7
    decal_key = rng.integers(0, 1 << 24, size=n_entities).astype(np.uint32)
    decal_layer = rng.integers(0, n_layers, size=n_entities).astype(np.uint8)
    # layers 0..5 -> buckets: 0, 1 plain; 2 depth-sorted; layer 3 not rendered; 4 -> bucket 1 again; 5 depth-sorted bucket 3
    layer_to_bucket = np.full(255, 0xFF, np.uint8)
    layer_to_bucket[:6] = [0, 1, 2, 0xFF, 1, 3]
    bucket_depth_sorted = np.array([0, 0, 1, 1], np.uint8)
    return {"models": models, "mesh_types": np.array(mesh_types, np.uint8), "model": model, "material_offset": material_offset, "mesh_materials": mm,
            "lod": lod, "flags": flags, "dirty": dirty, "pose_frame": pose_frame, "decal_key": decal_key, "decal_layer": decal_layer,
            "curve_key": decal_key[::-1].copy(), "curve_layer": decal_layer[::-1].copy(), "layer_to_bucket": layer_to_bucket,
            "bucket_depth_sorted": bucket_depth_sorted, "max_sort_key": max_sort_key}


def animation(n_bones: int = 64, frame_count: int = 30, fps: float = 30.0, seed: int = 21, root_motion: bool = True, bone_limit: int | None = None):
    """A compressed animation in the layout AnimationSampler reads (animation/animation.h:86-115): per bone a constant or a
    bit-packed translation track (or none) and a constant or bit-packed rotation track (3 channels + sign bit, the largest
    component skipped), streams of frame_count + 1 frames (animation.cpp:464), optional root-motion tracks served from
    uncompressed per-frame arrays (animation.cpp:33-37, :320)."""
    from .api import ANIM_CONST_TRANSLATION, ANIM_TRANSLATION_TRACK, ANIM_CONST_ROTATION, ANIM_ROTATION_TRACK
    rng = np.random.default_rng(seed)
    nb = n_bones if bone_limit is None else bone_limit
    ct, tt, cr, rt = [], [], [], []
    for b in range(nb):
        k = rng.integers(0, 4)
        if k == 0:
            ct.append(b)
        elif k <= 2:
            tt.append(b)
        k = rng.integers(0, 5)
        if k == 0:
            cr.append(b)
        elif k <= 3:
            rt.append(b)
    const_t = np.zeros(len(ct), ANIM_CONST_TRANSLATION)
    const_t["bone_index"], const_t["value"] = ct, rng.uniform(-1, 1, size=(len(ct), 3))
    const_r = np.zeros(len(cr), ANIM_CONST_ROTATION)
    const_r["bone_index"], const_r["value"] = cr, random_unit_quats(rng, len(cr))
    tracks_t = np.zeros(len(tt), ANIM_TRANSLATION_TRACK)
    tracks_r = np.zeros(len(rt), ANIM_ROTATION_TRACK)
    n_frames = frame_count + 1

    def pack(tracks, with_sign):
        off = 0
        for i in range(len(tracks)):
            bits = rng.integers(5, 17, size=3)
            tracks["bitsizes"][i] = bits
            tracks["offset_bits"][i] = off
            off += int(bits.sum()) + (1 if with_sign else 0)
        frame_bits = off
        stream = np.zeros((frame_bits * n_frames + 7) // 8 + 8, np.uint8)
        big = 0
        for f in range(n_frames):
            for i in range(len(tracks)):
                pos = frame_bits * f + int(tracks["offset_bits"][i])
                val, sh = 0, 0
                if with_sign:
                    val |= int(rng.integers(0, 2))
                    sh = 1
                for c in range(3):
                    nbits = int(tracks["bitsizes"][i][c])
                    val |= int(rng.integers(0, 1 << nbits)) << sh
                    sh += nbits
                big |= val << pos
        raw = big.to_bytes(len(stream), "little")
        stream[:] = np.frombuffer(raw, np.uint8)
        return frame_bits, stream

    tracks_t["bone_index"] = tt
    tracks_t["min"] = rng.uniform(-1.0, 0.0, size=(len(tt), 3))
    tfs, tstream = pack(tracks_t, False)
    tracks_t["to_range"] = rng.uniform(0.5, 2.0, size=(len(tt), 3)) / ((1 << tracks_t["bitsizes"].astype(np.int64)) - 1)
    tracks_r["bone_index"] = rt
    tracks_r["skipped_channel"] = rng.integers(0, 4, size=len(rt))
    tracks_r["min"] = np.float32(-0.70710678)
    rfs, rstream = pack(tracks_r, True)
    tracks_r["to_range"] = np.float32(1.41421356) / ((1 << tracks_r["bitsizes"].astype(np.int64)) - 1)
    root_t = int(rng.integers(0, len(tt))) if root_motion and len(tt) else -1
    root_r = int(rng.integers(0, len(rt))) if root_motion and len(rt) else -1
    return {"fps": np.float32(fps), "frame_count": frame_count, "length": int(frame_count / fps * 32768), "translations_frame_size_bits": tfs,
            "rotations_frame_size_bits": rfs, "const_translations": const_t, "translations": tracks_t, "const_rotations": const_r, "rotations": tracks_r,
            "translation_stream": tstream, "rotation_stream": rstream, "root_translation_track": root_t, "root_rotation_track": root_r,
            "root_pose_translations": rng.uniform(-2, 2, size=(n_frames, 3)).astype(np.float32), "root_pose_rotations": random_unit_quats(rng, n_frames)}
