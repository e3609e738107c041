"""Shared test inputs: cameras, edge-case culling scenes, comparison helpers."""
from __future__ import annotations

import os

import numpy as np

from lumixengine_amd import scenes

# pytest --hostsim (tests/conftest.py): the kernels run lane by lane on the CPU - results are checked as on the GPU, wall-clock bounds are not
TIMELESS = os.environ.get("LMX_HOSTSIM") == "1"


def quat_from_yaw_pitch(yaw: float, pitch: float) -> np.ndarray:
    cy, sy, cp, sp = np.cos(yaw / 2), np.sin(yaw / 2), np.cos(pitch / 2), np.sin(pitch / 2)
    qy = np.array([0, sy, 0, cy])
    qx = np.array([sp, 0, 0, cp])
    x1, y1, z1, w1 = qy
    x2, y2, z2, w2 = qx
    q = np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])
    return (q / np.linalg.norm(q)).astype(np.float32)


# (name, kwargs for viewport_frustum): the cameras SURVEY.md §8d lists plus shadow-cascade-like ortho views
CAMERAS = [
    ("origin_identity", dict(pos=(0.0, 0.0, 0.0), rot=(0, 0, 0, 1))),
    ("origin_yaw_pitch", dict(pos=(12.5, -3.25, 40.0), rot=quat_from_yaw_pitch(0.7, -0.3))),
    ("far_camera", dict(pos=(1.0e6, 50.0, -1.0e6), rot=quat_from_yaw_pitch(2.1, 0.2))),
    ("narrow_fov", dict(pos=(-800.0, 120.0, 300.0), rot=quat_from_yaw_pitch(-1.3, 0.1), fov=float(np.deg2rad(20.0)), far=4000.0)),
    ("ortho_cascade_small", dict(is_ortho=True, ortho_size=37.5, w=1024, h=1024, near=0.0, far=1200.0, pos=(10.0, 400.0, -20.0), rot=quat_from_yaw_pitch(0.4, -1.0))),
    ("ortho_cascade_large", dict(is_ortho=True, ortho_size=600.0, w=1024, h=1024, near=0.0, far=3000.0, pos=(-300.0, 900.0, 250.0), rot=quat_from_yaw_pitch(-2.0, -0.8))),
    ("ortho_axis_aligned", dict(is_ortho=True, ortho_size=128.0, w=512, h=512, near=0.0, far=512.0, pos=(0.0, 0.0, 0.0), rot=(0, 0, 0, 1))),
]


def frusta(builder, names=None) -> np.ndarray:
    """Stack of ShiftedFrustum records built by `builder.viewport_frustum` (oracle or api module)."""
    cams = [c for c in CAMERAS if names is None or c[0] in names]
    return np.concatenate([builder.viewport_frustum(**kw) for _, kw in cams])


def cascade_frusta(builder, n: int = 8) -> np.ndarray:
    """n ortho frusta of growing extent around two light directions (config 5: 2 x 4 shadow cascades)."""
    out = []
    for k in range(n):
        size = [30.0, 90.0, 400.0, 1500.0][k % 4]
        rot = quat_from_yaw_pitch(0.3 + 1.1 * (k // 4), -0.9)
        out.append(builder.viewport_frustum(is_ortho=True, ortho_size=size, w=1024, h=1024, near=0.0, far=5000.0 + 2 * size, pos=(5.0 * k, 700.0, -3.0 * k), rot=rot))
    return np.concatenate(out)


def edge_case_scene():
    """Hand-built spheres around the quirks SURVEY.md §7/§8c lists. Returns the same dict as scenes.cull_scene."""
    pos, radius, type_ = [], [], []

    def add(p, r, t=0):
        pos.append(p)
        radius.append(r)
        type_.append(t)

    # cells straddling zero: int() truncates toward zero, so (-299.9 .. 299.9) is cell 0
    for x in (-299.9, -0.0, 0.0, 299.9, -300.0, 300.0, -300.1, 300.1, -450.0, 299.99999999):
        add((x, 10.0, -50.0), 5.0)
        add((10.0, x, -50.0), 5.0)
        add((10.0, 10.0, x), 5.0)
    # radius exactly at / just above the big-cell threshold (radius > 300.0f)
    add((100.0, 0.0, -500.0), 300.0)
    add((100.0, 0.0, -500.0), float(np.nextafter(np.float32(300.0), np.float32(400.0))))
    add((100.0, 0.0, -500.0), 900.0)
    add((5000.0, 5000.0, 5000.0), 8000.0)  # big sphere far outside, reaches the frustum
    # tangent to the axis-aligned ortho frustum planes (exact arithmetic: t == 0 -> visible)
    for d in (0.0, 1.0, -1.0):
        add((128.0 + 16.0 + d, 0.0, -100.0), 16.0)
        add((-128.0 - 16.0 + d, 0.0, -100.0), 16.0)
        add((0.0, 128.0 + 8.0 + d, -100.0), 8.0)
        add((0.0, 0.0, -512.0 - 4.0 + d), 4.0)
        add((0.0, 0.0, 4.0 + d), 4.0)
    # zero and negative radius, infinite and NaN radius (NaN is never culled by `t < 0`)
    add((1.0, 2.0, -30.0), 0.0)
    add((1.0, 2.0, -30.0), -5.0)
    add((50.0, 20.0, -300.0), float("inf"))
    add((50.0, 20.0, -300.0), float("nan"))
    add((7000.0, 20.0, 300.0), float("nan"))
    # one crowded cell: > 200 spheres -> several reference pages in one cell
    rng = np.random.default_rng(7)
    for p in rng.uniform(0.0, 299.0, size=(450, 3)):
        add((float(p[0]), float(p[1]), float(-p[2] - 300.0)), float(rng.uniform(0.5, 30.0)))
    # far from the origin (1e6) and very far (1e9): exercises the fp64 shift
    for p in rng.uniform(-2000.0, 2000.0, size=(300, 3)):
        add((1.0e6 + float(p[0]), 50.0 + float(p[1]), -1.0e6 + float(p[2])), float(rng.uniform(0.5, 50.0)))
    add((1.0e9, 1.0e9, -1.0e9), 10.0)
    # other renderable types
    for p in rng.uniform(-400.0, 400.0, size=(200, 3)):
        add((float(p[0]), float(p[1]), float(p[2] - 400.0)), float(rng.uniform(0.5, 80.0)), int(rng.integers(1, 5)))
    n = len(pos)
    return {
        "entity": (np.arange(n, dtype=np.int32) * 3 + 1),  # sparse, non-contiguous entity indices
        "type": np.array(type_, np.uint8),
        "pos": np.array(pos, np.float64),
        "radius": np.array(radius, np.float32),
    }


def mixed_scene(n=20000, half_extent=2500.0, seed=11):
    sc = scenes.cull_scene(n, half_extent, seed=seed, big_fraction=0.002, mixed_types=True)
    return sc


def sorted_by_type(ids: np.ndarray, types: np.ndarray):
    """{type: sorted ids} — the parity form of a cull result (order inside a type is unspecified in the reference)."""
    return {int(t): np.sort(ids[types == t]) for t in np.unique(types)}


def assert_same_visible(a, b, what=""):
    ka, kb = sorted(a.keys()), sorted(b.keys())
    assert ka == kb, f"{what}: type sets differ {ka} vs {kb}"
    for t in ka:
        assert np.array_equal(a[t], b[t]), f"{what}: type {t}: {len(a[t])} vs {len(b[t])} ids, symmetric difference {np.setxor1d(a[t], b[t])[:10]}"


def bits_equal(a: np.ndarray, b: np.ndarray) -> bool:
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def transforms_bits_equal(a: np.ndarray, b: np.ndarray) -> bool:
    return all(bits_equal(np.ascontiguousarray(a[k]), np.ascontiguousarray(b[k])) for k in ("pos", "rot", "scale"))


def write_world_blob(entities, world, hierarchy, module_names=("renderer", "animation"), names=None, partitions=False, compress=None, n_slots=None):
    """World::serialize (engine/world.cpp:837-898) for a world without module payloads: `entities` = valid entity indices, `world` =
    Transform per listed entity, `hierarchy` = list of (entity, parent, first_child, next_sibling, local Transform). `compress` =
    bytes -> bytes LZ4 block compressor (the reference's own, oracle/_ref). `n_slots` = m_entities.size(), the blob's first field
    (entity SLOTS, valid or not; default: highest listed index + 1). Test helper: restated writer, byte-identical to the reference's
    own World::serialize where both can express the world (tests/test_world_blob.py::test_restated_writer_matches_reference_serializer)."""
    import struct

    def tr_bytes(t):
        return struct.pack("<3d4f3f", *[float(x) for x in t["pos"]], *[float(x) for x in t["rot"]], *[float(x) for x in t["scale"]])

    if n_slots is None:
        n_slots = (max(int(e) for e in entities) + 1) if len(entities) else 0
    blob = bytearray(struct.pack("<I", n_slots))
    for e, t in zip(entities, world):
        blob += struct.pack("<i", int(e)) + tr_bytes(t)
        if partitions:
            blob += struct.pack("<H", 0)
    blob += struct.pack("<i", -1)
    names = names or []
    blob += struct.pack("<I", len(names))
    for e, name in names:
        blob += struct.pack("<i", int(e)) + name.encode() + b"\0"
    blob += struct.pack("<I", len(hierarchy))
    for (e, p, fc, ns, local) in hierarchy:
        blob += struct.pack("<4i", int(e), int(p), int(fc), int(ns)) + tr_bytes(local)
    blob += struct.pack("<i", 0)  # module count inside the blob: no payloads
    packed = compress(bytes(blob))
    out = bytearray(struct.pack("<II", 0x4C57524C, 6))  # WorldHeader {'LWRL', WorldVersion::LATEST}
    out += struct.pack("<i", len(module_names))
    for m in module_names:
        out += m.encode() + b"\0"
    out += struct.pack("<I", 1 if partitions else 0)
    out += struct.pack("<II", len(blob), len(packed)) + packed
    return bytes(out), bytes(blob)


# ---- BASELINE config 2 at full size: scenes, cameras and a deterministic update stream ------------------------------------
CONFIG2_SCENES = {
    # name: (half extent, mixed renderable types) - 10 M entities each, seed 2 (the scene bench.py times is "sparse" without types)
    "sparse_mixed": (15000.0, True),
    "dense": (5000.0, False),
}


def config2_cameras(builder):
    """[(name, frustum)] for the 10 M parity test: the default player camera (bench.py's), a yaw/pitch camera, SURVEY.md 8d's
    off-origin camera at (1e6, 50, -1e6) (sees nothing of a scene around the origin: exercises the fp64 shift), a narrow camera
    (selects the 2048-sphere tile variant) and a camera outside that sees the whole cube (every tile TILE_ACCEPT)."""
    return [
        ("default", builder.viewport_frustum()),
        ("yaw_pitch", builder.viewport_frustum(pos=(120.5, -30.25, 400.0), rot=quat_from_yaw_pitch(0.7, -0.3))),
        ("far_origin", builder.viewport_frustum(pos=(1.0e6, 50.0, -1.0e6), rot=quat_from_yaw_pitch(2.1, 0.2))),
        ("narrow", builder.viewport_frustum(pos=(-800.0, 120.0, 300.0), rot=quat_from_yaw_pitch(-1.3, 0.1), fov=float(np.deg2rad(20.0)), far=4000.0)),
        ("all_visible", builder.viewport_frustum(pos=(0.0, 0.0, 60000.0), far=300000.0)),
    ]


def churn_stream(pos: np.ndarray, half_extent: float, frames: int, per_frame: int, seed: int = 77):
    """Deterministic update stream for a scene whose entities 0..n-1 sit at `pos`: per frame `per_frame` removals of distinct
    alive entities, `per_frame` adds of new entity ids (n, n+1, ...) and per_frame // 2 `set` calls on entities the stream never
    removes - every other one stays within a unit of its old position (in-cell move: a 16-byte patch), the rest jump anywhere
    (cell change: tombstone + overflow). Yields dicts of numpy arrays."""
    n = len(pos)
    rng = np.random.default_rng(seed)
    victims = rng.permutation(n)[: frames * per_frame * 2].astype(np.int32)
    next_id = n
    for f in range(frames):
        rem = victims[f * per_frame : (f + 1) * per_frame]
        add_ids = np.arange(next_id, next_id + per_frame, dtype=np.int32)
        next_id += per_frame
        add_pos = rng.uniform(-half_extent, half_extent, size=(per_frame, 3))
        add_r = np.exp(rng.uniform(np.log(0.5), np.log(50.0), size=per_frame)).astype(np.float32)
        add_r[:: 97] = np.float32(420.0)  # a few big ones
        add_t = (rng.random(per_frame) < 0.1).astype(np.uint8) * 2
        movers = victims[(frames + f) * per_frame : (frames + f) * per_frame + per_frame // 2]  # never removed by this stream
        set_pos = rng.uniform(-half_extent, half_extent, size=(len(movers), 3))
        near = np.arange(len(movers)) % 2 == 0
        set_pos[near] = pos[movers[near]] + rng.uniform(-1.0, 1.0, size=(int(near.sum()), 3))
        set_r = np.exp(rng.uniform(np.log(0.5), np.log(50.0), size=len(movers))).astype(np.float32)
        yield dict(remove=rem, add_ids=add_ids, add_type=add_t, add_pos=add_pos, add_radius=add_r, set_ids=movers, set_pos=set_pos, set_radius=set_r)


def visible_digest(ids: np.ndarray, types: np.ndarray):
    """Parity form of a large result: per-type counts + sha256 over the per-type sorted id lists (types ascending)."""
    import hashlib

    h = hashlib.sha256()
    counts = [0] * 8
    for t in range(8):
        a = np.sort(ids[types == t]).astype(np.int32)
        counts[t] = int(len(a))
        h.update(a.tobytes())
    return counts, h.hexdigest()


def ids_digest(ids: np.ndarray) -> str:
    """sha256 of the sorted visible ids regardless of renderable type (what bench.py asserts for the scenes it times)."""
    import hashlib

    return hashlib.sha256(np.sort(np.asarray(ids, np.int32)).tobytes()).hexdigest()


def array_digest(*arrays):
    import hashlib

    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()
