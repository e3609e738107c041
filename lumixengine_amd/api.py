"""ctypes binding of liblumix_mi355.so (include/lumix_mi355.h) + thin host classes that mirror the reference's
interfaces for the hot path (names and argument meaning follow the C++ originals):

* :class:`CullingSystem`  — src/renderer/culling_system.h:58-77 (`add`, `remove`, `set`, `setPosition`, `setRadius`,
  `getRadius`, `isAdded`, `cull`)
* :class:`World`          — the transform/hierarchy subset of src/engine/world.h:49-209 in its batch form
* :class:`Skinning`       — Pose::computeAbsolute + computeSkinMatrices + evaluateSkin (src/renderer/pose.cpp,
  src/renderer/model.cpp:103-137) over many instances

This module is plumbing for tests and bench.py; a LumixEngine build binds the same C ABI from C++.
It never computes anything on the CPU: if the library or a gfx950 device is missing, it raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LMX_LIB_PATH") or os.path.join(PKG, "liblumix_mi355.so")  # LMX_LIB_PATH: tools/ sweeps over kernel build variants

MAX_FRUSTA, MAX_TYPES, MAX_VIEWS = 8, 8, 8
TYPE_ALL = 0xFF
CULL_OPT_TILE_VARIANT, CULL_OPT_MAX_SHARDS, CULL_OPT_COUNTER_PAD, CULL_OPT_AUTO_COMPACTION, CULL_OPT_DEVICE_OWNS_BOUND, CULL_OPT_OVERFLOW_RESERVE, CULL_OPT_ASYNC_COMPACTION, CULL_OPT_COMPACTION_MIN, CULL_OPT_MAP_ZERO_COPY = 0, 2, 3, 4, 5, 6, 7, 8, 9  # (1: retired)
KEYS_OPT_SLOT_ORDER, KEYS_OPT_SPLIT_STATE, KEYS_OPT_WALK_SHARDS, KEYS_OPT_BLOCK_RANKS = 0, 1, 2, 3
WORLD_OPT_FUSED_LEVELS = 0
SKIN_OPT_INSTANCES_PER_BLOCK = 0
SKIN_INSTANCES_PER_BLOCK_DEFAULT = 2  # what a fresh context uses (lmx_context.h: SkinState::multi)
(K_CULL_CLASSIFY, K_CULL_SPHERES, K_XFORM_LEVEL, K_SPHERE_REFRESH, K_POSE_PALETTE, K_SKIN_VERTICES, K_CULL_DYNAMIC) = range(7)
KERNEL_NAMES = ["cull_classify", "cull_spheres", "xform_level", "sphere_refresh", "pose_palette", "skin_vertices", "cull_dynamic", "sort_keys", "anim_update", "cull_patch"]
K_CULL_PATCH = 9

SHIFTED_FRUSTUM = np.dtype(
    [("xs", "<f4", 8), ("ys", "<f4", 8), ("zs", "<f4", 8), ("ds", "<f4", 8), ("points", "<f4", (8, 3)), ("origin", "<f8", 3), ("_pad", "<f8")],
    align=True,
)
TRANSFORM = np.dtype([("pos", "<f8", 3), ("rot", "<f4", 4), ("scale", "<f4", 3), ("_pad", "<f4")], align=True)
LOCAL_RIGID = np.dtype([("pos", "<f4", 3), ("rot", "<f4", 4)], align=True)
MATRIX = np.dtype([("columns", "<f4", (4, 4))], align=True)
SKIN = np.dtype([("weights", "<f4", 4), ("indices", "<i2", 4)], align=True)
ANIM_CONST_TRANSLATION = np.dtype([("value", "<f4", 3), ("bone_index", "<u2"), ("_pad", "<u2")], align=True)
ANIM_TRANSLATION_TRACK = np.dtype([("min", "<f4", 3), ("to_range", "<f4", 3), ("offset_bits", "<u2"), ("bone_index", "<u2"), ("bitsizes", "u1", 3), ("_pad", "u1")], align=True)
ANIM_CONST_ROTATION = np.dtype([("value", "<f4", 4), ("bone_index", "<u2"), ("_pad", "<u2")], align=True)
BLEND_SAMPLE = np.dtype([("animation", "<u4"), ("weight", "<f4"), ("time", "<u4"), ("looped", "<u4")], align=True)  # LmxBlendSample
ANIM_ROTATION_TRACK = np.dtype([("min", "<f4", 3), ("to_range", "<f4", 3), ("offset_bits", "<u2"), ("bone_index", "<u2"), ("bitsizes", "u1", 3), ("skipped_channel", "u1")],
                               align=True)


class LmxAnimation(C.Structure):
    """LmxAnimation of include/lmx_types.h."""
    _fields_ = [("fps", C.c_float), ("frame_count", C.c_uint32), ("length", C.c_uint32), ("translations_frame_size_bits", C.c_uint32),
                ("rotations_frame_size_bits", C.c_uint32), ("n_const_translations", C.c_uint32), ("n_translations", C.c_uint32),
                ("n_const_rotations", C.c_uint32), ("n_rotations", C.c_uint32), ("const_translations", C.c_void_p), ("translations", C.c_void_p),
                ("const_rotations", C.c_void_p), ("rotations", C.c_void_p), ("translation_stream", C.c_void_p), ("translation_stream_size", C.c_uint64),
                ("rotation_stream", C.c_void_p), ("rotation_stream_size", C.c_uint64), ("root_translation_track", C.c_int32),
                ("root_rotation_track", C.c_int32), ("root_pose_translations", C.c_void_p), ("root_pose_rotations", C.c_void_p)]


def animation_struct(a: dict):
    """(LmxAnimation, keep-alive list) from a lumixengine_amd.scenes.animation dict."""
    keep = [np.ascontiguousarray(a["const_translations"], ANIM_CONST_TRANSLATION), np.ascontiguousarray(a["translations"], ANIM_TRANSLATION_TRACK),
            np.ascontiguousarray(a["const_rotations"], ANIM_CONST_ROTATION), np.ascontiguousarray(a["rotations"], ANIM_ROTATION_TRACK),
            np.ascontiguousarray(a["translation_stream"], np.uint8), np.ascontiguousarray(a["rotation_stream"], np.uint8),
            np.ascontiguousarray(a["root_pose_translations"], np.float32), np.ascontiguousarray(a["root_pose_rotations"], np.float32)]
    st = LmxAnimation(float(a["fps"]), int(a["frame_count"]), int(a["length"]), int(a["translations_frame_size_bits"]), int(a["rotations_frame_size_bits"]),
                      len(keep[0]), len(keep[1]), len(keep[2]), len(keep[3]), _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]), _ptr(keep[3]), _ptr(keep[4]),
                      len(keep[4]), _ptr(keep[5]), len(keep[5]), int(a["root_translation_track"]), int(a["root_rotation_track"]), _ptr(keep[6]), _ptr(keep[7]))
    return st, keep


LOD_INDICES = np.dtype([("from", "<i4"), ("to", "<i4")])
KEYS_MODEL = np.dtype([("lod_distances", "<f4", 4), ("lod_indices", LOD_INDICES, 5), ("first_mesh", "<u4"), ("mesh_count", "<u4")], align=True)
MESH_MATERIAL = np.dtype([("sort_key", "<u4"), ("layer", "u1"), ("_pad", "u1", 3)], align=True)
KEYS_VIEW = np.dtype(
    [("camera_pos", "<f8", 3), ("lod_ref_point", "<f8", 3), ("lod_multiplier", "<f4"), ("time_delta", "<f4"), ("frame_number", "<u4"), ("is_shadow", "u1"),
     ("layer_to_bucket", "u1", 255), ("bucket_depth_sorted", "u1", 256)],
    align=True,
)
KEYS_COUNTS = np.dtype([("pairs", "<u4"), ("instanced", "<u4"), ("groups", "<u4"), ("poses", "<u4"), ("dirty", "<u4"), ("overflow", "<u4")])
VIEWPORT = np.dtype(
    [("is_ortho", "<i4"), ("fov", "<f4"), ("ortho_size", "<f4"), ("w", "<i4"), ("h", "<i4"), ("pos", "<f8", 3), ("rot", "<f4", 4), ("near_plane", "<f4"), ("far_plane", "<f4")],
    align=True,
)

# every symbol include/lumix_mi355.h declares: (name, restype, argtypes)
_vp, _u32, _i32, _u8, _f32, _ci, _sz = C.c_void_p, C.c_uint32, C.c_int32, C.c_uint8, C.c_float, C.c_int, C.c_size_t
SYMBOLS = {
    "lmx_ctx_create": (_ci, [_ci, C.POINTER(_vp)]),
    "lmx_ctx_destroy": (None, [_vp]),
    "lmx_ctx_acquire_shared": (_ci, [_vp, _ci, C.POINTER(_vp)]),
    "lmx_ctx_release_shared": (None, [_vp]),
    "lmx_ctx_lock": (None, [_vp]),
    "lmx_ctx_unlock": (None, [_vp]),
    "lmx_last_error": (C.c_char_p, [_vp]),
    "lmx_ctx_set_stream": (_ci, [_vp, _vp]),
    "lmx_ctx_synchronize": (_ci, [_vp]),
    "lmx_profile_enable": (_ci, [_vp, _ci]),
    "lmx_profile_reset": (_ci, [_vp]),
    "lmx_profile_get": (_ci, [_vp, _ci, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "lmx_cull_build": (_ci, [_vp, _u32, _vp, _vp, _vp, _vp]),
    "lmx_cull_add": (_ci, [_vp, _i32, _u8, _vp, _f32]),
    "lmx_cull_remove": (_ci, [_vp, _i32]),
    "lmx_cull_set": (_ci, [_vp, _i32, _vp, _f32]),
    "lmx_cull_set_position": (_ci, [_vp, _i32, _vp]),
    "lmx_cull_set_radius": (_ci, [_vp, _i32, _f32]),
    "lmx_cull_get_radius": (_ci, [_vp, _i32, C.POINTER(_f32)]),
    "lmx_cull_is_added": (_ci, [_vp, _i32]),
    "lmx_cull_flush": (_ci, [_vp]),
    "lmx_cull_add_many": (_ci, [_vp, _u32, _vp, _vp, _vp, _vp]),
    "lmx_cull_set_many": (_ci, [_vp, _u32, _vp, _vp, _vp]),
    "lmx_cull_remove_many": (_ci, [_vp, _u32, _vp]),
    "lmx_cull_compact": (_ci, [_vp]),
    "lmx_cull_update_stats": (_ci, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "lmx_cull_async_stats": (_ci, [_vp, C.POINTER(_ci), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "lmx_cull_set_option": (_ci, [_vp, _ci, _ci]),
    "lmx_cull_read_all": (_ci, [_vp, _u32, _u32, _vp, _u32, _vp]),
    "lmx_cull_map_all": (_ci, [_vp, _u32, _u32, _vp, _vp]),
    "lmx_cull_map_many": (_ci, [_vp, _u32, _u32, _vp, _vp]),
    "lmx_cull_map_begin": (_ci, [_vp, _u32, _u32]),
    "lmx_cull_map_end": (_ci, [_vp, _u32, _u32, _vp, _vp]),
    "lmx_cull_view_acquire": (_ci, [_vp, _vp, _u32]),
    "lmx_cull_view_release": (_ci, [_vp, _u32]),
    "lmx_cull_pack_device": (_ci, [_vp, _u32, _u32, _vp, _vp]),
    "lmx_cull_device_shards": (_ci, [_vp, _u32, _u32, _vp]),
    "lmx_cull_stats": (_ci, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "lmx_cull_layout_info": (_ci, [_vp, C.POINTER(_u32), C.POINTER(C.c_uint64)]),
    "lmx_cull": (_ci, [_vp, _u32, _vp, _u32, _u8]),
    "lmx_cull_set_pass_width": (_ci, [_vp, _u32]),
    "lmx_cull_counts": (_ci, [_vp, _u32, _vp]),
    "lmx_cull_read": (_ci, [_vp, _u32, _u32, _u8, _vp, _u32, C.POINTER(_u32)]),
    "lmx_cull_bind_output": (_ci, [_vp, _u32, _vp, _sz, _vp]),
    "lmx_cull_device_result": (_ci, [_vp, _u32, _u32, C.POINTER(_vp), C.POINTER(_vp), _vp, C.POINTER(_u32)]),
    "lmx_exchange_unique_id": (_ci, [_vp]),
    "lmx_exchange_create": (_ci, [_vp, _ci, _ci, _vp, _u32, C.POINTER(_vp)]),
    "lmx_exchange_destroy": (None, [_vp]),
    "lmx_exchange_cull": (_ci, [_vp, _vp, _u8, C.POINTER(_u32)]),
    "lmx_exchange_cull_many": (_ci, [_vp, _vp, _u32, _u8, C.POINTER(_u32)]),
    "lmx_exchange_read_many": (_ci, [_vp, _u32, _ci, _u32, _vp, _vp, _u32]),
    "lmx_exchange_wait": (_ci, [_vp, _u32]),
    "lmx_exchange_result": (_ci, [_vp, _u32, C.POINTER(_vp), C.POINTER(_u32), C.POINTER(_vp)]),
    "lmx_exchange_read": (_ci, [_vp, _u32, _ci, _vp, _vp, _u32]),
    "lmx_exchange_info": (_ci, [_vp, _vp, _vp, _vp]),
    "lmx_exchange_set_caps": (_ci, [_vp, _u32, _vp, _ci]),
    "lmx_exchange_time_gather": (_ci, [_vp, _u32, _vp, _vp]),
    "lmx_exchange_layout": (_ci, [_vp, _u32, _vp, _vp, _vp, _vp]),
    "lmx_exchange_stats": (_ci, [_vp, _u32, _vp]),
    "lmx_world_build": (_ci, [_vp, _u32, _vp, _vp]),
    "lmx_world_build_with_world": (_ci, [_vp, _u32, _vp, _vp, _vp]),
    "lmx_world_set_parent": (_ci, [_vp, _i32, _i32]),
    "lmx_world_read_local_transforms": (_ci, [_vp, _vp, _u32]),
    "lmx_transform_compose": (_ci, [_vp, _vp, _vp]),
    "lmx_transform_compute_local": (_ci, [_vp, _vp, _vp]),
    "lmx_world_set_transforms": (_ci, [_vp, _u32, _vp, _vp]),
    "lmx_world_set_world_transforms": (_ci, [_vp, _u32, _vp, _vp]),
    "lmx_world_set_transforms_device": (_ci, [_vp, _u32, _vp, _vp]),
    "lmx_world_bind_culling": (_ci, [_vp, _u32, _vp, _vp]),
    "lmx_world_propagate": (_ci, [_vp]),
    "lmx_world_read_transforms": (_ci, [_vp, _vp, _u32]),
    "lmx_world_set_option": (_ci, [_vp, _ci, _ci]),
    "lmx_world_track_moved": (_ci, [_vp, _ci]),
    "lmx_world_read_moved": (_ci, [_vp, _vp, _vp, _u32, C.POINTER(_u32)]),
    "lmx_world_set_bone_attachments": (_ci, [_vp, _u32, _vp, _vp, _vp, _vp, _vp]),
    "lmx_world_update_bone_attachments": (_ci, [_vp]),
    "lmx_skin_add_model": (_ci, [_vp, _u32, _vp, _vp, _i32, C.POINTER(_u32)]),
    "lmx_skin_add_mesh": (_ci, [_vp, _u32, _vp, _vp, C.POINTER(_u32)]),
    "lmx_skin_set_instances": (_ci, [_vp, _u32, _vp, _vp]),
    "lmx_skin_upload_poses": (_ci, [_vp, _vp, _vp, _sz]),
    "lmx_skin_upload_poses_device": (_ci, [_vp, _vp, _vp, _sz]),
    "lmx_skin_set_pose_source_device": (_ci, [_vp, _vp, _vp, _sz]),
    "lmx_skin_blend_poses": (_ci, [_vp, _vp, _vp, _sz, _f32]),
    "lmx_skin_blend_poses_device": (_ci, [_vp, _vp, _vp, _sz, _f32]),
    "lmx_skin_set_mode": (_ci, [_vp, _ci]),
    "lmx_skin_set_option": (_ci, [_vp, _ci, _ci]),
    "lmx_skin_run": (_ci, [_vp]),
    "lmx_skin_read_vertices": (_ci, [_vp, _u32, _vp, _u32]),
    "lmx_skin_read_vertices_range": (_ci, [_vp, _u32, _u32, _vp, C.c_size_t]),
    "lmx_skin_device_output": (_ci, [_vp, _vp, _vp]),
    "lmx_skin_read_palette": (_ci, [_vp, _u32, _vp, _u32]),
    "lmx_skin_set_pose_writeback": (_ci, [_vp, _ci]),
    "lmx_skin_enable_dual_quats": (_ci, [_vp, _ci]),
    "lmx_skin_read_dual_quats": (_ci, [_vp, _u32, _vp, _u32]),
    "lmx_skin_read_pose": (_ci, [_vp, _u32, _vp, _vp, _u32]),
    "lmx_anim_add": (_ci, [_vp, _vp, C.POINTER(_u32)]),
    "lmx_anim_set_model_pose": (_ci, [_vp, _u32, _vp, _u32]),
    "lmx_anim_set_animables": (_ci, [_vp, _u32, _vp, _vp]),
    "lmx_anim_set_weight": (_ci, [_vp, _f32]),
    "lmx_anim_update": (_ci, [_vp, _f32]),
    "lmx_anim_eval_blend_stacks": (_ci, [_vp, _u32, _vp, _vp]),
    "lmx_anim_read_times": (_ci, [_vp, _vp, _u32]),
    "lmx_anim_read_pose": (_ci, [_vp, _u32, _vp, _vp, _u32]),
    "lmx_keys_set_models": (_ci, [_vp, _vp, _u32, _vp, _u32]),
    "lmx_keys_set_instances": (_ci, [_vp, _u32, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    "lmx_keys_set_decals": (_ci, [_vp, _u32, _vp, _vp, _vp, _vp]),
    "lmx_keys_set_positions": (_ci, [_vp, _vp, _u32]),
    "lmx_keys_bind_world": (_ci, [_vp, _ci]),
    "lmx_keys_set_option": (_ci, [_vp, _ci, _ci]),
    "lmx_keys_run": (_ci, [_vp, _u32, _u32, _vp, _u32]),
    "lmx_keys_sort": (_ci, [_vp]),
    "lmx_keys_counts": (_ci, [_vp, _vp]),
    "lmx_keys_read_pairs": (_ci, [_vp, _vp, _vp, _u32]),
    "lmx_keys_read_instancer": (_ci, [_vp, _vp, _vp, _u32]),
    "lmx_keys_read_poses": (_ci, [_vp, _vp, _u32]),
    "lmx_keys_read_dirty": (_ci, [_vp, _vp, _u32]),
    "lmx_keys_read_state": (_ci, [_vp, _vp, _vp, _u32]),
    "lmx_keys_device_pairs": (_ci, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "lmx_viewport_frustum": (_ci, [_vp, _vp]),
    "lmx_frustum_perspective": (_ci, [_vp, _vp, _vp, _f32, _f32, _f32, _f32, _vp]),
    "lmx_frustum_ortho": (_ci, [_vp, _vp, _vp, _f32, _f32, _f32, _f32, _vp]),
    "lmx_world_blob_info": (_ci, [_vp, _sz, _vp]),
    "lmx_world_blob_read": (_ci, [_vp, _sz, _u32, _vp, _vp, _vp, _vp]),
    "lmx_world_blob_find_module": (_ci, [_vp, _sz, C.c_char_p, C.POINTER(_u32), C.POINTER(_i32)]),
    "lmx_render_blob_info": (_ci, [_vp, _sz, _vp]),
    "lmx_render_blob_read_bone_attachments": (_ci, [_vp, _sz, _u32, _vp]),
    "lmx_render_blob_read_model_instances": (_ci, [_vp, _sz, _u32, _vp, _vp, _vp, _u32]),
    "lmx_version": (C.c_char_p, []),
}

ERROR_NAMES = {1: "INVALID_ARGUMENT", 2: "NO_DEVICE", 3: "HIP", 4: "OUT_OF_MEMORY", 5: "CAPACITY", 6: "NOT_BUILT", 7: "BUSY"}
ERR_BUSY = 7


class LumixError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"LMX_ERR_{ERROR_NAMES.get(code, code)}: {message}")
        self.code = code


_lib = None


def load_library() -> C.CDLL:
    """Loads liblumix_mi355.so and declares every exported entry point. Raises if the HIP extension is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -m lumixengine_amd.build` (there is no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64x3(v) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(3))


def _f32x3(v) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(3))


class Context:
    """One GPU + one HIP stream (lmx_ctx_*)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.lmx_ctx_create(device, C.byref(h))
        if rc != 0:
            raise LumixError(rc, self.lib.lmx_last_error(None).decode())
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.lmx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int):
        if rc != 0:
            raise LumixError(rc, self.lib.lmx_last_error(self.h).decode())

    def set_stream(self, hip_stream: Optional[int]):
        self.check(self.lib.lmx_ctx_set_stream(self.h, hip_stream))

    def synchronize(self):
        self.check(self.lib.lmx_ctx_synchronize(self.h))

    def profile_enable(self, on: bool = True):
        self.check(self.lib.lmx_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self.check(self.lib.lmx_profile_reset(self.h))

    def profile_get(self, kernel: int):
        ms, n = C.c_double(0), C.c_uint64(0)
        self.check(self.lib.lmx_profile_get(self.h, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value


# ---- frusta (host mirror of core/geometry.cpp) ----------------------------------------------------------------
def viewport_frustum(is_ortho=False, fov=float(np.deg2rad(60.0)), ortho_size=100.0, w=1920, h=1080, pos=(0, 0, 0), rot=(0, 0, 0, 1), near=0.1,
                     far=10000.0) -> np.ndarray:
    """Viewport::getFrustum() (core/geometry.cpp:793-818)."""
    lib = load_library()
    vp = np.zeros(1, VIEWPORT)
    vp["is_ortho"], vp["fov"], vp["ortho_size"], vp["w"], vp["h"] = int(is_ortho), fov, ortho_size, w, h
    vp["pos"], vp["rot"], vp["near_plane"], vp["far_plane"] = pos, rot, near, far
    out = np.zeros(1, SHIFTED_FRUSTUM)
    rc = lib.lmx_viewport_frustum(_ptr(vp), _ptr(out))
    if rc:
        raise LumixError(rc, "lmx_viewport_frustum")
    return out


def frustum_perspective(pos, direction, up, fov, ratio, near, far) -> np.ndarray:
    out = np.zeros(1, SHIFTED_FRUSTUM)
    rc = load_library().lmx_frustum_perspective(_ptr(_f64x3(pos)), _ptr(_f32x3(direction)), _ptr(_f32x3(up)), fov, ratio, near, far, _ptr(out))
    if rc:
        raise LumixError(rc, "lmx_frustum_perspective")
    return out


def frustum_ortho(pos, direction, up, width, height, near, far) -> np.ndarray:
    out = np.zeros(1, SHIFTED_FRUSTUM)
    rc = load_library().lmx_frustum_ortho(_ptr(_f64x3(pos)), _ptr(_f32x3(direction)), _ptr(_f32x3(up)), width, height, near, far, _ptr(out))
    if rc:
        raise LumixError(rc, "lmx_frustum_ortho")
    return out


def transform_compose(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Transform::compose (core/math.cpp:801-807), element-wise over two TRANSFORM arrays."""
    lib = load_library()
    a, b = np.ascontiguousarray(a, TRANSFORM), np.ascontiguousarray(b, TRANSFORM)
    out = np.zeros(len(a), TRANSFORM)
    for i in range(len(a)):
        lib.lmx_transform_compose(_ptr(a[i : i + 1]), _ptr(b[i : i + 1]), _ptr(out[i : i + 1]))
    return out


def transform_compute_local(parent: np.ndarray, child: np.ndarray) -> np.ndarray:
    """Transform::computeLocal (core/math.cpp:809-816)."""
    lib = load_library()
    parent, child = np.ascontiguousarray(parent, TRANSFORM), np.ascontiguousarray(child, TRANSFORM)
    out = np.zeros(len(parent), TRANSFORM)
    for i in range(len(parent)):
        lib.lmx_transform_compute_local(_ptr(parent[i : i + 1]), _ptr(child[i : i + 1]), _ptr(out[i : i + 1]))
    return out


class CullResult:
    """Result of one cull call: the visible ids per (frustum, type), resident in HBM (view slot) until the slot is reused.

    The reference returns a linked list of 4 KiB pages, each tagged with a renderable type (culling_system.h:17-56);
    `ids(frustum, type)` is the concatenation of the pages of that type, `pages(...)` re-creates the page split.
    """

    def __init__(self, cs: "CullingSystem", view: int, n_frusta: int):
        self.cs, self.view, self.n_frusta = cs, view, n_frusta
        self._counts = None

    def counts(self) -> np.ndarray:
        if self._counts is None:
            out = np.zeros((self.n_frusta, MAX_TYPES), np.uint32)
            self.cs.ctx.check(self.cs.lib.lmx_cull_counts(self.cs.ctx.h, self.view, _ptr(out)))
            self._counts = out
        return self._counts

    def count(self, frustum: int = 0) -> int:
        return int(self.counts()[frustum].sum())

    def ids(self, frustum: int = 0, type_: int = 0) -> np.ndarray:
        n = int(self.counts()[frustum, type_])
        out = np.zeros(n, np.int32)
        got = C.c_uint32(0)
        self.cs.ctx.check(self.cs.lib.lmx_cull_read(self.cs.ctx.h, self.view, frustum, type_, _ptr(out), n, C.byref(got)))
        return out[: got.value]

    def all_ids(self, frustum: int = 0):
        """(ids, types) over all types, like walking the whole CullResult list (lmx_cull_read_all: two host waits)."""
        n = int(self.counts()[frustum].sum())
        out = np.zeros(max(n, 1), np.int32)
        cnt = np.zeros(MAX_TYPES, np.uint32)
        self.cs.ctx.check(self.cs.lib.lmx_cull_read_all(self.cs.ctx.h, self.view, frustum, _ptr(out), len(out), _ptr(cnt)))
        return out[: int(cnt.sum())], np.repeat(np.arange(MAX_TYPES, dtype=np.uint8), cnt)

    def map_all(self, frustum: int = 0):
        """(ids, types) like all_ids, through lmx_cull_map_all (normally one host wait; the ids are copied out of the library's pinned
        buffer here - a C caller reads them in place)."""
        p = C.POINTER(C.c_int32)()
        cnt = np.zeros(MAX_TYPES, np.uint32)
        self.cs.ctx.check(self.cs.lib.lmx_cull_map_all(self.cs.ctx.h, self.view, frustum, C.byref(p), _ptr(cnt)))
        n = int(cnt.sum())
        ids = np.frombuffer(C.string_at(p, 4 * n), np.int32) if n else np.zeros(0, np.int32)
        return ids, np.repeat(np.arange(MAX_TYPES, dtype=np.uint8), cnt)

    def pages(self, frustum: int = 0, type_: int = 0, page_ids: int = 1020):
        a = self.ids(frustum, type_)
        return [a[i : i + page_ids] for i in range(0, len(a), page_ids)]

    def device_result(self, frustum: int = 0):
        """(d_ids ptr, d_counts ptr, type_offsets[MAX_TYPES], capacity) for GPU-side consumers."""
        d_ids, d_counts, cap = C.c_void_p(), C.c_void_p(), C.c_uint32(0)
        offs = np.zeros(MAX_TYPES, np.uint32)
        self.cs.ctx.check(self.cs.lib.lmx_cull_device_result(self.cs.ctx.h, self.view, frustum, C.byref(d_ids), C.byref(d_counts), _ptr(offs), C.byref(cap)))
        return d_ids.value, d_counts.value, offs, cap.value


class CullingSystem:
    """GPU-backed CullingSystem (src/renderer/culling_system.h:58-77)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.lib = ctx.lib

    # bulk form of `add` for scene load
    def build(self, entity, type_, pos, radius):
        entity = np.ascontiguousarray(entity, np.int32)
        type_ = np.ascontiguousarray(type_, np.uint8)
        pos = np.ascontiguousarray(pos, np.float64).reshape(-1, 3)
        radius = np.ascontiguousarray(radius, np.float32)
        assert len(entity) == len(type_) == len(pos) == len(radius)
        self.ctx.check(self.lib.lmx_cull_build(self.ctx.h, len(entity), _ptr(entity), _ptr(type_), _ptr(pos), _ptr(radius)))

    def add(self, entity: int, type_: int, pos, radius: float):
        self.ctx.check(self.lib.lmx_cull_add(self.ctx.h, int(entity), int(type_), _ptr(_f64x3(pos)), float(radius)))

    def remove(self, entity: int):
        self.ctx.check(self.lib.lmx_cull_remove(self.ctx.h, int(entity)))

    def set(self, entity: int, pos, radius: float):
        self.ctx.check(self.lib.lmx_cull_set(self.ctx.h, int(entity), _ptr(_f64x3(pos)), float(radius)))

    def setPosition(self, entity: int, pos):
        self.ctx.check(self.lib.lmx_cull_set_position(self.ctx.h, int(entity), _ptr(_f64x3(pos))))

    def setRadius(self, entity: int, radius: float):
        self.ctx.check(self.lib.lmx_cull_set_radius(self.ctx.h, int(entity), float(radius)))

    def getRadius(self, entity: int) -> float:
        r = C.c_float(0)
        self.ctx.check(self.lib.lmx_cull_get_radius(self.ctx.h, int(entity), C.byref(r)))
        return r.value

    def isAdded(self, entity: int) -> bool:
        return bool(self.lib.lmx_cull_is_added(self.ctx.h, int(entity)))

    def flush(self):
        self.ctx.check(self.lib.lmx_cull_flush(self.ctx.h))

    def addMany(self, entity, type_, pos, radius):
        entity = np.ascontiguousarray(entity, np.int32)
        type_ = np.ascontiguousarray(type_, np.uint8)
        pos = np.ascontiguousarray(pos, np.float64).reshape(-1, 3)
        radius = np.ascontiguousarray(radius, np.float32)
        assert len(entity) == len(type_) == len(pos) == len(radius)
        self.ctx.check(self.lib.lmx_cull_add_many(self.ctx.h, len(entity), _ptr(entity), _ptr(type_), _ptr(pos), _ptr(radius)))

    def setMany(self, entity, pos, radius):
        entity = np.ascontiguousarray(entity, np.int32)
        pos = np.ascontiguousarray(pos, np.float64).reshape(-1, 3)
        radius = np.ascontiguousarray(radius, np.float32)
        assert len(entity) == len(pos) == len(radius)
        self.ctx.check(self.lib.lmx_cull_set_many(self.ctx.h, len(entity), _ptr(entity), _ptr(pos), _ptr(radius)))

    def removeMany(self, entity):
        entity = np.ascontiguousarray(entity, np.int32)
        self.ctx.check(self.lib.lmx_cull_remove_many(self.ctx.h, len(entity), _ptr(entity)))

    def compact(self):
        self.ctx.check(self.lib.lmx_cull_compact(self.ctx.h))

    def updateStats(self):
        v = [C.c_uint32(0) for _ in range(4)]
        self.ctx.check(self.lib.lmx_cull_update_stats(self.ctx.h, *[C.byref(x) for x in v]))
        return dict(zip(("static", "bound", "overflow", "tombstones"), (x.value for x in v)))

    def asyncStats(self):
        """LMX_CULL_OPT_ASYNC_COMPACTION: {"state": -1 off / 0 idle / 1 requested / 2 running / 3 ready / 4 failed, "jobs", "swaps", "ops_replayed_at_swaps", "log_drains"}"""
        st, jobs, swaps, ops, drains = C.c_int(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self.ctx.check(self.lib.lmx_cull_async_stats(self.ctx.h, C.byref(st), C.byref(jobs), C.byref(swaps), C.byref(ops), C.byref(drains)))
        return {"state": st.value, "jobs": jobs.value, "swaps": swaps.value, "ops_replayed_at_swaps": ops.value, "log_drains": drains.value}

    def setOption(self, option: int, value: int):
        self.ctx.check(self.lib.lmx_cull_set_option(self.ctx.h, int(option), int(value)))

    def stats(self):
        a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        self.ctx.check(self.lib.lmx_cull_stats(self.ctx.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"entities": a.value, "cells": b.value, "chunks": c.value}

    def layoutInfo(self):
        """{"cell_key_bytes": 8 (keys relative to the tile's box) | 16, "table_bytes": per-tile tables + chunk headers a full cull reads}"""
        kb, tb = C.c_uint32(0), C.c_uint64(0)
        self.ctx.check(self.lib.lmx_cull_layout_info(self.ctx.h, C.byref(kb), C.byref(tb)))
        return {"cell_key_bytes": kb.value, "table_bytes": tb.value}

    def bindOutput(self, view: int, d_ids: Optional[int], ids_capacity: int, d_counts: Optional[int]):
        """Result slot `view` writes into caller-owned device memory (raw pointers, e.g. torch tensors' data_ptr())."""
        self.ctx.check(self.lib.lmx_cull_bind_output(self.ctx.h, view, d_ids, ids_capacity, d_counts))

    def setPassWidth(self, frusta_per_pass: int):
        self.ctx.check(self.lib.lmx_cull_set_pass_width(self.ctx.h, frusta_per_pass))

    def packDevice(self, view: int = 0, frustum: int = 0):
        """(device pointer, words) of the packed record [8 counts | ids, types back to back] of the view's last cull - no host wait."""
        p, n = C.c_void_p(), _u32(0)
        self.ctx.check(self.lib.lmx_cull_pack_device(self.ctx.h, view, frustum, C.byref(p), C.byref(n)))
        return p.value, n.value

    def cull(self, frusta: np.ndarray, type_: int = TYPE_ALL, view: int = 0) -> CullResult:
        """cull(frustum[, type]); `frusta` may hold up to 8 ShiftedFrustum records tested in one pass."""
        frusta = np.ascontiguousarray(frusta, SHIFTED_FRUSTUM).reshape(-1)
        self.ctx.check(self.lib.lmx_cull(self.ctx.h, view, _ptr(frusta), len(frusta), type_))
        return CullResult(self, view, len(frusta))


def exchange_unique_id() -> bytes:
    """ncclGetUniqueId (rank 0): 128 opaque bytes every rank passes to VisibleExchange."""
    buf = np.zeros(128, np.uint8)
    rc = load_library().lmx_exchange_unique_id(_ptr(buf))
    if rc:
        raise LumixError(rc, "lmx_exchange_unique_id (is RCCL installed?)")
    return buf.tobytes()


class ExchangeStats(C.Structure):
    """LmxExchangeStats (include/lumix_mi355.h)"""
    _fields_ = [("n_frusta", C.c_uint32), ("record_words", C.c_uint32), ("caps", C.c_uint32 * 8), ("max_visible", C.c_uint32 * 8), ("overflow_mask", C.c_uint32),
                ("used_words_own", C.c_uint32), ("used_words_max", C.c_uint32), ("mode", C.c_int32), ("bytes_shipped_per_peer", C.c_uint64), ("bytes_used", C.c_uint64),
                ("gather_us", C.c_double), ("gather_us_record_words", C.c_uint32), ("reserved", C.c_uint32)]


class VisibleExchange:
    """lmx_exchange_*: per frame one cull of this rank's entities + one RCCL all-gather of [8 counts | cap ids] per rank."""

    def __init__(self, ctx: Context, rank: int, world: int, unique_id: bytes, ids_per_rank: int):
        self.ctx, self.lib, self.rank, self.world, self.cap = ctx, ctx.lib, rank, world, int(ids_per_rank)
        uid = np.frombuffer(unique_id, np.uint8).copy()
        h = C.c_void_p()
        ctx.check(self.lib.lmx_exchange_create(ctx.h, rank, world, _ptr(uid), self.cap, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.lmx_exchange_destroy(self.h)
            self.h = None

    def cull(self, frustum: np.ndarray, type_: int = TYPE_ALL) -> int:
        frustum = np.ascontiguousarray(frustum, SHIFTED_FRUSTUM).reshape(-1)
        slot = C.c_uint32(0)
        self.ctx.check(self.lib.lmx_exchange_cull(self.h, _ptr(frustum), type_, C.byref(slot)))
        self.last_slot = slot.value
        return slot.value

    def cullMany(self, frusta: np.ndarray, type_: int = TYPE_ALL) -> int:
        """The frame's views (<= 8 frusta, the same number on every rank) in one pass + ONE all-gather; returns the slot."""
        frusta = np.ascontiguousarray(frusta, SHIFTED_FRUSTUM).reshape(-1)
        slot = C.c_uint32(0)
        self.ctx.check(self.lib.lmx_exchange_cull_many(self.h, _ptr(frusta), len(frusta), type_, C.byref(slot)))
        self.last_slot = slot.value
        return slot.value

    def layout(self, slot: int) -> dict:
        """lmx_exchange_layout: {"n_frusta", "caps"[n], "offsets"[n] (words from a rank's record start), "record_words"} of the slot's last frame"""
        n, rec = C.c_uint32(0), C.c_uint32(0)
        caps, offs = np.zeros(8, np.uint32), np.zeros(8, np.uint32)
        self.ctx.check(self.lib.lmx_exchange_layout(self.h, slot, C.byref(n), _ptr(caps), _ptr(offs), C.byref(rec)))
        return {"n_frusta": n.value, "caps": [int(c) for c in caps[: n.value]], "offsets": [int(o) for o in offs[: n.value]], "record_words": rec.value}

    def readMany(self, slot: int, rank: int, frustum: int):
        """(counts[8], ids) of one (rank, frustum) sub-record; ids clipped to the sub-record's capacity (layout(slot)["caps"][frustum])."""
        cap_f = self.layout(slot)["caps"][frustum]
        counts = np.zeros(MAX_TYPES, np.uint32)
        ids = np.zeros(max(cap_f, 1), np.int32)
        self.ctx.check(self.lib.lmx_exchange_read_many(self.h, slot, rank, frustum, _ptr(counts), _ptr(ids), cap_f))
        return counts, ids[: min(int(counts.sum()), cap_f)]

    def setCaps(self, caps, keep_fixed: bool = False):
        """capacities of the sub-records of frames of len(caps) frusta from now on (every rank: the same call at the same point)"""
        caps = np.ascontiguousarray(caps, np.uint32)
        self.ctx.check(self.lib.lmx_exchange_set_caps(self.h, len(caps), _ptr(caps), 1 if keep_fixed else 0))

    def timeGather(self, n_frusta: int = 1):
        """COLLECTIVE: (us per all-gather of the record a frame of n_frusta frusta ships now, that record's words per rank)"""
        us, words = C.c_double(-1.0), C.c_uint32(0)
        self.ctx.check(self.lib.lmx_exchange_time_gather(self.h, n_frusta, C.byref(us), C.byref(words)))
        return us.value, words.value

    def stats(self, slot: int) -> dict:
        """lmx_exchange_stats of the slot's last frame (waits for its gather)"""
        st = ExchangeStats()
        self.ctx.check(self.lib.lmx_exchange_stats(self.h, slot, C.byref(st)))
        n = st.n_frusta
        return {"n_frusta": n, "record_words": st.record_words, "caps": list(st.caps[:n]), "max_visible": list(st.max_visible[:n]), "overflow_mask": st.overflow_mask,
                "used_words_own": st.used_words_own, "used_words_max": st.used_words_max, "mode": ("inline", "side", "p2p")[st.mode],
                "bytes_shipped_per_peer": int(st.bytes_shipped_per_peer), "bytes_used": int(st.bytes_used),
                "gather_us": st.gather_us if st.gather_us >= 0 else None, "gather_us_record_words": st.gather_us_record_words}

    def wait(self, slot: int):
        self.ctx.check(self.lib.lmx_exchange_wait(self.h, slot))

    def info(self) -> dict:
        """how frames of the shape that ran last run: {"mode": "inline" | "side" | "p2p", "gather_us": one all-gather of that shape's record as last timed (None: never), "why"}"""
        mode, us, why = C.c_int(0), C.c_double(-1.0), C.c_char_p()
        self.ctx.check(self.lib.lmx_exchange_info(self.h, C.byref(mode), C.byref(us), C.byref(why)))
        return {"mode": ("inline", "side", "p2p")[mode.value], "gather_us": us.value if us.value >= 0 else None, "why": (why.value or b"").decode()}

    def read(self, slot: int, rank: int):
        """(counts[8], ids) of one rank's gathered record of a one-frustum frame (ids of type 0 first; clipped to the record's capacity)."""
        return self.readMany(slot, rank, 0)


class World:
    """Batch form of World's transform hierarchy (src/engine/world.cpp:255-282)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.lib = ctx.lib
        self.n = 0

    def build(self, parent, transforms):
        parent = np.ascontiguousarray(parent, np.int32)
        transforms = np.ascontiguousarray(transforms, TRANSFORM)
        assert len(parent) == len(transforms)
        self.n = len(parent)
        self.ctx.check(self.lib.lmx_world_build(self.ctx.h, self.n, _ptr(parent), _ptr(transforms)))

    def buildWithWorld(self, parent, local_transforms, world_transforms):
        """Mirror of a live World: the stored world transform of every entity + Hierarchy::local_transform of the parented ones."""
        parent = np.ascontiguousarray(parent, np.int32)
        local_transforms = np.ascontiguousarray(local_transforms, TRANSFORM)
        world_transforms = np.ascontiguousarray(world_transforms, TRANSFORM)
        assert len(parent) == len(local_transforms) == len(world_transforms)
        self.n = len(parent)
        self.ctx.check(self.lib.lmx_world_build_with_world(self.ctx.h, self.n, _ptr(parent), _ptr(local_transforms), _ptr(world_transforms)))

    def setTransforms(self, entity, transforms):
        """World::setTransform for roots / World::setLocalTransform for children, staged until propagate()."""
        entity = np.ascontiguousarray(entity, np.int32)
        transforms = np.ascontiguousarray(transforms, TRANSFORM)
        assert len(entity) == len(transforms)
        self.ctx.check(self.lib.lmx_world_set_transforms(self.ctx.h, len(entity), _ptr(entity), _ptr(transforms)))

    def setWorldTransforms(self, entity, transforms):
        """World::setTransform (world-space) on any entity, staged until propagate()."""
        entity = np.ascontiguousarray(entity, np.int32)
        transforms = np.ascontiguousarray(transforms, TRANSFORM)
        assert len(entity) == len(transforms)
        self.ctx.check(self.lib.lmx_world_set_world_transforms(self.ctx.h, len(entity), _ptr(entity), _ptr(transforms)))

    def setTransformsDevice(self, n: int, d_entity: int, d_transforms: int):
        """Same as setTransforms with both arrays already resident in HBM (raw device pointers)."""
        self.ctx.check(self.lib.lmx_world_set_transforms_device(self.ctx.h, n, d_entity, d_transforms))

    def bindCulling(self, entity, model_radius):
        entity = np.ascontiguousarray(entity, np.int32)
        model_radius = np.ascontiguousarray(model_radius, np.float32)
        self.ctx.check(self.lib.lmx_world_bind_culling(self.ctx.h, len(entity), _ptr(entity), _ptr(model_radius)))

    def setParent(self, new_parent: int, child: int):
        """World::setParent (world.cpp:619-701); new_parent < 0 detaches."""
        self.ctx.check(self.lib.lmx_world_set_parent(self.ctx.h, int(new_parent), int(child)))

    def setBoneAttachments(self, entity, parent_entity, skin_instance, bone_index, relative):
        """RenderModuleImpl::m_bone_attachments: entity follows bone `bone_index` of `skin_instance`, posed on `parent_entity`."""
        a = [np.ascontiguousarray(entity, np.int32), np.ascontiguousarray(parent_entity, np.int32), np.ascontiguousarray(skin_instance, np.uint32),
             np.ascontiguousarray(bone_index, np.uint32), np.ascontiguousarray(relative, LOCAL_RIGID)]
        self.ctx.check(self.lib.lmx_world_set_bone_attachments(self.ctx.h, len(a[0]), *[_ptr(x) for x in a]))

    def updateBoneAttachments(self):
        self.ctx.check(self.lib.lmx_world_update_bone_attachments(self.ctx.h))

    def getLocalTransforms(self) -> np.ndarray:
        out = np.zeros(self.n, TRANSFORM)
        self.ctx.check(self.lib.lmx_world_read_local_transforms(self.ctx.h, _ptr(out), self.n))
        return out

    def propagate(self):
        self.ctx.check(self.lib.lmx_world_propagate(self.ctx.h))

    def getTransforms(self) -> np.ndarray:
        out = np.zeros(self.n, TRANSFORM)
        self.ctx.check(self.lib.lmx_world_read_transforms(self.ctx.h, _ptr(out), self.n))
        return out

    def setOption(self, option: int, value: int):
        """WORLD_OPT_FUSED_LEVELS: 1 = hierarchies of <= 8 levels propagate in one launch, 0 (default) = one launch per level."""
        self.ctx.check(self.lib.lmx_world_set_option(self.ctx.h, option, value))

    def trackMoved(self, on: bool = True):
        """propagate() collects the entities whose world transform changed (what World::transformEntity would have visited)."""
        self.ctx.check(self.lib.lmx_world_track_moved(self.ctx.h, int(on)))

    def readMoved(self):
        """(entity int32[k], TRANSFORM[k]) moved by the propagate() calls since the last read; an entity moved by two of them is listed twice, newest last."""
        cap = max(2 * self.n, 1)
        ent, tr, n = np.zeros(cap, np.int32), np.zeros(cap, TRANSFORM), _u32(0)
        self.ctx.check(self.lib.lmx_world_read_moved(self.ctx.h, _ptr(ent), _ptr(tr), cap, C.byref(n)))
        return ent[: n.value].copy(), tr[: n.value].copy()


WORLD_BLOB_INFO = np.dtype([(k, "<u4") for k in ("version", "flags", "n_modules", "uncompressed_size", "compressed_size", "n_entities", "max_entity_index", "n_names",
                                                  "n_hierarchy")])


def world_blob_read(data: bytes):
    """Serialized World (engine/world.cpp:837-1043) -> (info dict, parent, transforms for World.build, world transforms, valid). Host only."""
    lib = load_library()
    buf = np.frombuffer(data, np.uint8)
    info = np.zeros(1, WORLD_BLOB_INFO)
    rc = lib.lmx_world_blob_info(_ptr(buf), len(buf), _ptr(info))
    if rc != 0:
        raise LumixError(rc, "not a current, LZ4-compressed World blob")
    n = int(info["max_entity_index"][0]) + 1 if info["n_entities"][0] else 0
    parent, tr, world, valid = np.zeros(n, np.int32), np.zeros(n, TRANSFORM), np.zeros(n, TRANSFORM), np.zeros(n, np.uint8)
    rc = lib.lmx_world_blob_read(_ptr(buf), len(buf), n, _ptr(parent), _ptr(tr), _ptr(world), _ptr(valid))
    if rc != 0:
        raise LumixError(rc, "World blob truncated or inconsistent")
    return {k: int(info[k][0]) for k in WORLD_BLOB_INFO.names}, parent, tr, world, valid


RENDER_BLOB_INFO = np.dtype([(k, np.int32 if k == "version" else np.uint32) for k in (
    "version", "payload_offset", "payload_size", "n_cameras", "n_model_instance_slots", "n_model_instances", "n_point_lights", "n_environments", "n_terrains",
    "n_particle_systems", "n_bone_attachments", "n_environment_probes", "n_reflection_probes", "n_decals", "n_curve_decals", "n_instanced_models",
    "n_procedural_geometries", "model_paths_size")])
BLOB_BONE_ATTACHMENT = np.dtype([("bone_name_hash", np.uint64), ("entity", np.int32), ("parent_entity", np.int32), ("pos", np.float32, 3), ("rot", np.float32, 4), ("_pad", np.uint32)])


def world_blob_find_module(data: bytes, name: str):
    """(payload offset in the decompressed blob, serialized version) of module `name`, or None. Host only."""
    lib = load_library()
    buf = np.frombuffer(data, np.uint8)
    off, ver = _u32(0), _i32(0)
    rc = lib.lmx_world_blob_find_module(_ptr(buf), len(buf), name.encode(), C.byref(off), C.byref(ver))
    return None if rc != 0 else (int(off.value), int(ver.value))


def render_blob_read(data: bytes):
    """The "renderer" module's payload of a serialized World (render_module.cpp:962-976): (info dict, bone attachments, {entity: model path}).
    Host only."""
    lib = load_library()
    buf = np.frombuffer(data, np.uint8)
    info = np.zeros(1, RENDER_BLOB_INFO)
    rc = lib.lmx_render_blob_info(_ptr(buf), len(buf), _ptr(info))
    if rc != 0:
        raise LumixError(rc, "no readable renderer payload (RenderModuleVersion 16..18) in this World blob")
    att = np.zeros(int(info["n_bone_attachments"][0]), BLOB_BONE_ATTACHMENT)
    rc = lib.lmx_render_blob_read_bone_attachments(_ptr(buf), len(buf), len(att), _ptr(att) if len(att) else None)
    if rc != 0:
        raise LumixError(rc, "bone attachment records truncated")
    n = int(info["n_model_instance_slots"][0])
    flags, off = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint32)
    paths = np.zeros(max(int(info["model_paths_size"][0]), 1), np.uint8)
    rc = lib.lmx_render_blob_read_model_instances(_ptr(buf), len(buf), n, _ptr(flags), _ptr(off), _ptr(paths), len(paths))
    if rc != 0:
        raise LumixError(rc, "model instance records truncated")
    table = paths.tobytes()
    models = {}
    for e in range(n):
        if flags[e] & 4:
            models[e] = None if off[e] == 0xFFFFFFFF else table[off[e] : table.index(b"\0", off[e])].decode()
    return {k: int(info[k][0]) for k in RENDER_BLOB_INFO.names}, att, models


def keys_view(camera_pos=(0, 0, 0), lod_ref_point=None, lod_multiplier=1.0, time_delta=1 / 60, frame_number=1, is_shadow=False,
              layer_to_bucket=None, bucket_depth_sorted=None) -> np.ndarray:
    """LmxKeysView: the per-view state PipelineImpl::createSortKeys reads (pipeline.cpp:3797-3832)."""
    kv = np.zeros(1, KEYS_VIEW)
    kv["camera_pos"] = camera_pos
    kv["lod_ref_point"] = camera_pos if lod_ref_point is None else lod_ref_point
    kv["lod_multiplier"], kv["time_delta"], kv["frame_number"], kv["is_shadow"] = lod_multiplier, time_delta, frame_number, int(is_shadow)
    kv["layer_to_bucket"] = 0xFF if layer_to_bucket is None else np.asarray(layer_to_bucket, np.uint8)
    if bucket_depth_sorted is not None:
        kv["bucket_depth_sorted"][0, : len(bucket_depth_sorted)] = np.asarray(bucket_depth_sorted, np.uint8)
    return kv


class SortKeys:
    """PipelineImpl::createSortKeys (renderer/pipeline.cpp:3789-3968) on the visible list a cull left on the device."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.lib = ctx.lib
        self.n_entities = 0
        self.max_sort_key = 0

    def setModels(self, models, mesh_types):
        models = np.ascontiguousarray(models, KEYS_MODEL)
        mesh_types = np.ascontiguousarray(mesh_types, np.uint8)
        self.ctx.check(self.lib.lmx_keys_set_models(self.ctx.h, _ptr(models), len(models), _ptr(mesh_types), len(mesh_types)))

    def setInstances(self, model, material_offset, mesh_materials, lod, flags, dirty, pose_frame):
        model = np.ascontiguousarray(model, np.int32)
        arrs = [np.ascontiguousarray(material_offset, np.uint32), np.ascontiguousarray(mesh_materials, MESH_MATERIAL), np.ascontiguousarray(lod, np.float32),
                np.ascontiguousarray(flags, np.uint8), np.ascontiguousarray(dirty, np.uint8), np.ascontiguousarray(pose_frame, np.uint32)]
        self.n_entities = len(model)
        self.ctx.check(self.lib.lmx_keys_set_instances(self.ctx.h, len(model), _ptr(model), _ptr(arrs[0]), _ptr(arrs[1]), len(arrs[1]), _ptr(arrs[2]),
                                                       _ptr(arrs[3]), _ptr(arrs[4]), _ptr(arrs[5])))

    def setDecals(self, n_entities, decal_sort_key=None, decal_layer=None, curve_sort_key=None, curve_layer=None):
        a = [None if x is None else np.ascontiguousarray(x, t) for x, t in ((decal_sort_key, np.uint32), (decal_layer, np.uint8), (curve_sort_key, np.uint32),
                                                                             (curve_layer, np.uint8))]
        self.ctx.check(self.lib.lmx_keys_set_decals(self.ctx.h, n_entities, *[None if x is None else _ptr(x) for x in a]))
        self.n_entities = max(self.n_entities, n_entities)

    def setPositions(self, xyz):
        xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
        self.ctx.check(self.lib.lmx_keys_set_positions(self.ctx.h, _ptr(xyz), len(xyz)))

    def bindWorld(self, on: bool = True):
        self.ctx.check(self.lib.lmx_keys_bind_world(self.ctx.h, int(on)))

    def setOption(self, option: int, value: int):
        self.ctx.check(self.lib.lmx_keys_set_option(self.ctx.h, int(option), int(value)))

    def run(self, kv, max_sort_key: int, view: int = 0, frustum: int = 0):
        kv = np.ascontiguousarray(kv, KEYS_VIEW)
        self.max_sort_key = int(max_sort_key)
        self.ctx.check(self.lib.lmx_keys_run(self.ctx.h, view, frustum, _ptr(kv), self.max_sort_key))

    def sort(self):
        self.ctx.check(self.lib.lmx_keys_sort(self.ctx.h))

    def counts(self) -> dict:
        c = np.zeros(1, KEYS_COUNTS)
        self.ctx.check(self.lib.lmx_keys_counts(self.ctx.h, _ptr(c)))
        return {k: int(c[k][0]) for k in KEYS_COUNTS.names}

    def readPairs(self):
        n = self.counts()["pairs"]
        keys, values = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        self.ctx.check(self.lib.lmx_keys_read_pairs(self.ctx.h, _ptr(keys), _ptr(values), n))
        return keys, values

    def readInstancer(self):
        n = self.counts()["instanced"]
        offsets, values = np.zeros(self.max_sort_key + 2, np.uint32), np.zeros(n, np.uint64)
        self.ctx.check(self.lib.lmx_keys_read_instancer(self.ctx.h, _ptr(offsets), _ptr(values), n))
        return offsets, values

    def readPoses(self) -> np.ndarray:
        out = np.zeros(self.counts()["poses"], np.int32)
        self.ctx.check(self.lib.lmx_keys_read_poses(self.ctx.h, _ptr(out), len(out)))
        return out

    def readDirty(self) -> np.ndarray:
        out = np.zeros(self.counts()["dirty"], np.int32)
        self.ctx.check(self.lib.lmx_keys_read_dirty(self.ctx.h, _ptr(out), len(out)))
        return out

    def readState(self):
        lod, frame = np.zeros(self.n_entities, np.float32), np.zeros(self.n_entities, np.uint32)
        self.ctx.check(self.lib.lmx_keys_read_state(self.ctx.h, _ptr(lod), _ptr(frame), self.n_entities))
        return lod, frame


SKIN_FUSED, SKIN_EXACT, SKIN_DQS = 0, 1, 2


class Skinning:
    """Pose -> palette -> skinned vertices for many model instances."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.lib = ctx.lib
        self._inst = []
        self._models = {}
        self._meshes = {}

    def addModel(self, parents, bind, first_nonroot: int) -> int:
        parents = np.ascontiguousarray(parents, np.int16)
        bind = np.ascontiguousarray(bind, LOCAL_RIGID)
        out = C.c_uint32(0)
        self.ctx.check(self.lib.lmx_skin_add_model(self.ctx.h, len(parents), _ptr(parents), _ptr(bind), int(first_nonroot), C.byref(out)))
        self._models[out.value] = len(parents)
        return out.value

    def addMesh(self, positions, skin) -> int:
        positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        skin = np.ascontiguousarray(skin, SKIN)
        out = C.c_uint32(0)
        self.ctx.check(self.lib.lmx_skin_add_mesh(self.ctx.h, len(positions), _ptr(positions), _ptr(skin), C.byref(out)))
        self._meshes[out.value] = len(positions)
        return out.value

    def setInstances(self, model, mesh):
        model = np.ascontiguousarray(model, np.uint32)
        mesh = np.ascontiguousarray(mesh, np.uint32)
        self._inst = [(int(a), int(b)) for a, b in zip(model, mesh)] if len(model) <= 4096 else None
        self._inst_model, self._inst_mesh = model, mesh
        self.ctx.check(self.lib.lmx_skin_set_instances(self.ctx.h, len(model), _ptr(model), _ptr(mesh)))

    def uploadPoses(self, positions, rotations):
        positions = np.ascontiguousarray(positions, np.float32)
        rotations = np.ascontiguousarray(rotations, np.float32)
        n_bones_total = positions.size // 3
        assert rotations.size == n_bones_total * 4
        self.ctx.check(self.lib.lmx_skin_upload_poses(self.ctx.h, _ptr(positions), _ptr(rotations), n_bones_total))

    def uploadPosesDevice(self, d_positions: int, d_rotations: int, n_bones_total: int):
        self.ctx.check(self.lib.lmx_skin_upload_poses_device(self.ctx.h, d_positions, d_rotations, n_bones_total))

    def setPoseSourceDevice(self, d_positions: Optional[int], d_rotations: Optional[int], n_bones_total: int):
        """run() reads the relative poses straight from this device memory (no copy) every time."""
        self.ctx.check(self.lib.lmx_skin_set_pose_source_device(self.ctx.h, d_positions, d_rotations, n_bones_total))

    def setMode(self, exact):
        """True / 1: FMA-free linear blend, bit-identical to the reference; False / 0 (default): fused multiply-adds, within 1e-5;
        2 (SKIN_DQS): the dual-quaternion blend of the reference's vertex shader."""
        self.ctx.check(self.lib.lmx_skin_set_mode(self.ctx.h, int(exact)))

    def setOption(self, option: int, value: int):
        """SKIN_OPT_INSTANCES_PER_BLOCK: 0 = k_skin_shared, 1 / 2 / 4 / 8 / 16 = k_skin_multi with that many instances per block."""
        self.ctx.check(self.lib.lmx_skin_set_option(self.ctx.h, int(option), int(value)))

    def run(self):
        self.ctx.check(self.lib.lmx_skin_run(self.ctx.h))

    def readVertices(self, instance: int) -> np.ndarray:
        n = self._meshes[int(self._inst_mesh[instance])]
        out = np.zeros((n, 3), np.float32)
        self.ctx.check(self.lib.lmx_skin_read_vertices(self.ctx.h, instance, _ptr(out), n))
        return out

    def readVerticesRange(self, first: int, count: int) -> np.ndarray:
        """Skinned positions of instances [first, first + count) in one copy: float32 [total vertices, 3], instances back to back."""
        n = int(sum(self._meshes[int(m)] for m in self._inst_mesh[first : first + count]))
        out = np.empty((n, 3), np.float32)
        self.ctx.check(self.lib.lmx_skin_read_vertices_range(self.ctx.h, first, count, _ptr(out), n))
        return out

    def deviceOutput(self):
        """(device pointer, total vertices) of the skinned positions in HBM (3 floats per vertex, instances back to back)."""
        p, n = C.c_void_p(), C.c_size_t()
        self.ctx.check(self.lib.lmx_skin_device_output(self.ctx.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def readPalette(self, instance: int) -> np.ndarray:
        n = self._models[int(self._inst_model[instance])]
        out = np.zeros(n, MATRIX)
        self.ctx.check(self.lib.lmx_skin_read_palette(self.ctx.h, instance, _ptr(out), n))
        return out

    # ---- animation sampling (AnimationModuleImpl::updateAnimable for every instance) ----
    def addAnimation(self, anim: dict) -> int:
        st, keep = animation_struct(anim)
        out = C.c_uint32(0)
        self.ctx.check(self.lib.lmx_anim_add(self.ctx.h, C.addressof(st), C.byref(out)))
        return out.value

    def setModelPose(self, model: int, relative):
        relative = np.ascontiguousarray(relative, LOCAL_RIGID)
        self.ctx.check(self.lib.lmx_anim_set_model_pose(self.ctx.h, model, _ptr(relative), len(relative)))

    def setAnimables(self, animation, time):
        animation, time = np.ascontiguousarray(animation, np.uint32), np.ascontiguousarray(time, np.uint32)
        self._n_animables = len(animation)
        self.ctx.check(self.lib.lmx_anim_set_animables(self.ctx.h, len(animation), _ptr(animation), _ptr(time)))

    def setAnimWeight(self, weight: float):
        self.ctx.check(self.lib.lmx_anim_set_weight(self.ctx.h, float(weight)))

    def updateAnimables(self, time_delta: float):
        self.ctx.check(self.lib.lmx_anim_update(self.ctx.h, float(time_delta)))

    def evalBlendStacks(self, stacks):
        """updateAnimator's pose work (animation_module.cpp:602-636): `stacks[i]` = instance i's SAMPLE instructions in order, each
        (animation id, weight, time, looped)."""
        first = np.zeros(len(stacks) + 1, np.uint32)
        first[1:] = np.cumsum([len(s) for s in stacks])
        samples = np.zeros(max(int(first[-1]), 1), BLEND_SAMPLE)
        k = 0
        for s in stacks:
            for (a, w, t, looped) in s:
                samples[k] = (a, w, t, 1 if looped else 0)
                k += 1
        self._n_animables = len(stacks)
        self.ctx.check(self.lib.lmx_anim_eval_blend_stacks(self.ctx.h, len(stacks), _ptr(first), _ptr(samples)))

    def readTimes(self) -> np.ndarray:
        out = np.zeros(self._n_animables, np.uint32)
        self.ctx.check(self.lib.lmx_anim_read_times(self.ctx.h, _ptr(out), len(out)))
        return out

    def readRelativePose(self, instance: int):
        n = self._models[int(self._inst_model[instance])]
        pos, rot = np.zeros((n, 3), np.float32), np.zeros((n, 4), np.float32)
        self.ctx.check(self.lib.lmx_anim_read_pose(self.ctx.h, instance, _ptr(pos), _ptr(rot), n))
        return pos, rot

    def blendPoses(self, positions, rotations, weight: float):
        """Pose::blend(rhs, weight) of the library's relative poses with `positions` / `rotations` (all instances back to back)."""
        positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        rotations = np.ascontiguousarray(rotations, np.float32).reshape(-1, 4)
        self.ctx.check(self.lib.lmx_skin_blend_poses(self.ctx.h, _ptr(positions), _ptr(rotations), len(positions), float(weight)))

    def setPoseWriteback(self, on: bool = True):
        """Store the absolute pose next to the palette (default) or not (readPose then fails)."""
        self.ctx.check(self.lib.lmx_skin_set_pose_writeback(self.ctx.h, int(on)))

    def enableDualQuats(self, on: bool = True):
        self.ctx.check(self.lib.lmx_skin_enable_dual_quats(self.ctx.h, int(on)))

    def readDualQuats(self, instance: int) -> np.ndarray:
        n = self._models[int(self._inst_model[instance])]
        out = np.zeros((n, 8), np.float32)
        self.ctx.check(self.lib.lmx_skin_read_dual_quats(self.ctx.h, instance, _ptr(out), n))
        return out

    def readPose(self, instance: int):
        n = self._models[int(self._inst_model[instance])]
        pos = np.zeros((n, 3), np.float32)
        rot = np.zeros((n, 4), np.float32)
        self.ctx.check(self.lib.lmx_skin_read_pose(self.ctx.h, instance, _ptr(pos), _ptr(rot), n))
        return pos, rot
