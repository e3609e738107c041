"""Shared test inputs: cameras, edge-case culling scenes, comparison helpers."""
from __future__ import annotations

import numpy as np

from lumixengine_amd import scenes


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


def write_world_blob(entities, world, hierarchy, module_names=("renderer", "animation"), names=None, partitions=False, compress=None):
    """World::serialize (engine/world.cpp:837-898) for a world without module payloads: `entities` = valid entity indices, `world` =
    Transform per listed entity, `hierarchy` = list of (entity, parent, first_child, next_sibling, local Transform). `compress` =
    bytes -> bytes LZ4 block compressor (the reference's own, oracle/_ref). Test helper: restated writer, not reference code."""
    import struct

    def tr_bytes(t):
        return struct.pack("<3d4f3f", *[float(x) for x in t["pos"]], *[float(x) for x in t["rot"]], *[float(x) for x in t["scale"]])

    blob = bytearray(struct.pack("<I", len(entities)))
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
