"""pyoracle — TEST INFRASTRUCTURE: ctypes bindings for the CPU oracle libraries.

Two libraries export the same functions under different prefixes:

* ``orc_*``  oracle/liblmx_oracle.so      plain-C restatement (oracle/lmx_oracle.c), always buildable
* ``ref_*``  oracle/_ref/liblmx_ref.so    the reference's OWN code: math.cpp, geometry.cpp, culling_system.cpp, page_allocator.cpp,
                                           world.cpp compiled in place; pose / palette / skin code, the animation sampler and
                                           createSortKeys sliced out of their files at build time (oracle/ref/*_shim.cpp,
                                           slice_*.py); only buildable where /root/reference exists, but the built .so travels
                                           with the repo snapshot

Only tests/, bench.py's ``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import this module. The product
package (lumixengine_amd) must never import it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liblmx_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "liblmx_ref.so")

# numpy mirrors of include/lmx_types.h
SHIFTED_FRUSTUM = np.dtype(
    [("xs", "<f4", 8), ("ys", "<f4", 8), ("zs", "<f4", 8), ("ds", "<f4", 8), ("points", "<f4", (8, 3)), ("origin", "<f8", 3), ("_pad", "<f8")],
    align=True,
)
FRUSTUM = np.dtype([("xs", "<f4", 8), ("ys", "<f4", 8), ("zs", "<f4", 8), ("ds", "<f4", 8), ("points", "<f4", (8, 3))], align=True)
TRANSFORM = np.dtype([("pos", "<f8", 3), ("rot", "<f4", 4), ("scale", "<f4", 3), ("_pad", "<f4")], align=True)
LOCAL_RIGID = np.dtype([("pos", "<f4", 3), ("rot", "<f4", 4)], align=True)
MATRIX = np.dtype([("columns", "<f4", (4, 4))], align=True)
SKIN = np.dtype([("weights", "<f4", 4), ("indices", "<i2", 4)], align=True)
VIEWPORT = np.dtype(
    [("is_ortho", "<i4"), ("fov", "<f4"), ("ortho_size", "<f4"), ("w", "<i4"), ("h", "<i4"), ("pos", "<f8", 3), ("rot", "<f4", 4), ("near_plane", "<f4"), ("far_plane", "<f4")],
    align=True,
)
assert SHIFTED_FRUSTUM.itemsize == 256 and FRUSTUM.itemsize == 224 and TRANSFORM.itemsize == 56
assert LOCAL_RIGID.itemsize == 28 and MATRIX.itemsize == 64 and SKIN.itemsize == 24


def build(force: bool = False) -> None:
    """Build the C oracle (always) and oracle/_ref (only where the reference tree exists)."""
    args = ["make", "-s", "-f", os.path.join(HERE, "Makefile")]
    if force:
        args.append("-B")
    subprocess.run(args + ["oracle"], check=True)
    subprocess.run(args + ["ref"], check=True)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64x3(v) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(3))


def _f32x3(v) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(3))


class Oracle:
    """Uniform wrapper over either oracle library. ``kind`` is 'port' (orc_*) or 'reference' (ref_*)."""

    def __init__(self, kind: str = "port"):
        if kind == "port":
            path, self.prefix = ORACLE_SO, "orc_"
        elif kind == "port_o3":  # -O3 -march=x86-64-v3 build of the port: a labelled CPU baseline (bench.py), never a checker - FMA changes the rounding
            path, self.prefix = ORACLE_SO.replace("liblmx_oracle.so", "liblmx_oracle_o3.so"), "orc_"
        elif kind == "reference":
            path, self.prefix = REF_SO, "ref_"
        else:
            raise ValueError(kind)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not built (run oracle.pyoracle.build())")
        self.kind = kind
        self.lib = C.CDLL(path)
        self._declare()

    def _fn(self, name, restype, argtypes):
        f = getattr(self.lib, self.prefix + name)
        f.restype = restype
        f.argtypes = argtypes
        return f

    def _declare(self):
        vp, u32, i32, u8, f32, ci = C.c_void_p, C.c_uint32, C.c_int32, C.c_uint8, C.c_float, C.c_int
        self.f_cs_create = self._fn("cs_create", vp, [])
        self.f_cs_destroy = self._fn("cs_destroy", None, [vp])
        self.f_cs_add = self._fn("cs_add", None, [vp, i32, u8, vp, f32])
        self.f_cs_add_bulk = self._fn("cs_add_bulk", None, [vp, u32, vp, vp, vp, vp])
        self.f_cs_remove = self._fn("cs_remove", None, [vp, i32])
        self.f_cs_set = self._fn("cs_set", None, [vp, i32, vp, f32])
        self.f_cs_set_position = self._fn("cs_set_position", None, [vp, i32, vp])
        self.f_cs_set_radius = self._fn("cs_set_radius", None, [vp, i32, f32])
        self.f_cs_get_radius = self._fn("cs_get_radius", f32, [vp, i32])
        self.f_cs_is_added = self._fn("cs_is_added", ci, [vp, i32])
        self.f_cs_cell_count = self._fn("cs_cell_count", u32, [vp])
        self.f_cs_cull = self._fn("cs_cull", u32, [vp, vp, u8, ci, vp, vp, u32, vp])
        self.f_viewport_frustum = self._fn("viewport_frustum", None, [vp, vp])
        self.f_frustum_perspective = self._fn("frustum_perspective", None, [vp, vp, vp, f32, f32, f32, f32, vp])
        self.f_frustum_ortho = self._fn("frustum_ortho", None, [vp, vp, vp, f32, f32, f32, f32, vp])
        self.f_contains_aabb = self._fn("contains_aabb", ci, [vp, vp, vp])
        self.f_intersects_aabb = self._fn("intersects_aabb", ci, [vp, vp, vp])
        self.f_get_relative = self._fn("get_relative", None, [vp, vp, vp])
        self.f_compose = self._fn("compose", None, [vp, vp, vp])
        self.f_bone_attachment = self._fn("bone_attachment", None, [vp, vp, vp, vp, vp, vp])
        self.f_compute_local = self._fn("compute_local", None, [vp, vp, vp])
        self.f_world_create = self._fn("world_create", vp, [u32])
        self.f_world_destroy = self._fn("world_destroy", None, [vp])
        self.f_world_init_transforms = self._fn("world_init_transforms", None, [vp, u32, vp, vp])
        self.f_world_set_parents = self._fn("world_set_parents", None, [vp, u32, vp, vp])
        self.f_world_set_transforms = self._fn("world_set_transforms", None, [vp, u32, vp, vp])
        self.f_world_set_local_transforms = self._fn("world_set_local_transforms", None, [vp, u32, vp, vp])
        self.f_world_get_transforms = self._fn("world_get_transforms", None, [vp, u32, vp])
        self.f_world_get_local_transforms = self._fn("world_get_local_transforms", None, [vp, u32, vp])
        self.f_world_bind_culling = self._fn("world_bind_culling", None, [vp, vp, u32, vp, vp])
        self.f_pose_compute_absolute = self._fn("pose_compute_absolute", None, [vp, vp, vp, i32, u32, u32, ci])
        self.f_invert_bind = self._fn("invert_bind", None, [vp, vp, u32])
        self.f_pose_blend = self._fn("pose_blend", None, [vp, vp, vp, vp, u32, C.c_float])
        self.f_skin_matrices = self._fn("skin_matrices", None, [vp, vp, vp, vp, u32, u32, ci])
        self.f_dual_quats = self._fn("dual_quats", None, [vp, vp, vp, vp, u32, u32])
        self.f_evaluate_skin = self._fn("evaluate_skin", None, [vp, vp, vp, vp, u32, u32, u32, ci])
        self.f_rand_fill = self._fn("rand_fill", None, [u32, u32, u32, vp])
        self.f_describe = self._fn("describe", C.c_char_p, [])
        # createSortKeys exists only as a restatement (pipeline.cpp cannot be compiled on its own: no ref_ twin)
        self.f_create_sort_keys = getattr(self.lib, self.prefix + "create_sort_keys", None)
        if self.f_create_sort_keys is not None:
            self.f_create_sort_keys.restype = C.c_int
            self.f_create_sort_keys.argtypes = [vp, u32, vp, u32, vp, u32, vp, u32] + [vp] * 15

    def describe(self) -> str:
        return self.f_describe().decode()

    # ---- frusta -----------------------------------------------------------------------------------------
    def viewport_frustum(self, is_ortho=False, fov=np.deg2rad(60.0), ortho_size=100.0, w=1920, h=1080, pos=(0, 0, 0), rot=(0, 0, 0, 1),
                         near=0.1, far=10000.0) -> np.ndarray:
        vp = np.zeros(1, VIEWPORT)
        vp["is_ortho"], vp["fov"], vp["ortho_size"], vp["w"], vp["h"] = int(is_ortho), fov, ortho_size, w, h
        vp["pos"], vp["rot"], vp["near_plane"], vp["far_plane"] = pos, rot, near, far
        out = np.zeros(1, SHIFTED_FRUSTUM)
        self.f_viewport_frustum(_ptr(vp), _ptr(out))
        return out

    def frustum_perspective(self, pos, direction, up, fov, ratio, near, far) -> np.ndarray:
        out = np.zeros(1, SHIFTED_FRUSTUM)
        self.f_frustum_perspective(_ptr(_f64x3(pos)), _ptr(_f32x3(direction)), _ptr(_f32x3(up)), fov, ratio, near, far, _ptr(out))
        return out

    def frustum_ortho(self, pos, direction, up, width, height, near, far) -> np.ndarray:
        out = np.zeros(1, SHIFTED_FRUSTUM)
        self.f_frustum_ortho(_ptr(_f64x3(pos)), _ptr(_f32x3(direction)), _ptr(_f32x3(up)), width, height, near, far, _ptr(out))
        return out

    def contains_aabb(self, frustum, pos, size) -> bool:
        return bool(self.f_contains_aabb(_ptr(frustum), _ptr(_f64x3(pos)), _ptr(_f32x3(size))))

    def intersects_aabb(self, frustum, pos, size) -> bool:
        return bool(self.f_intersects_aabb(_ptr(frustum), _ptr(_f64x3(pos)), _ptr(_f32x3(size))))

    def get_relative(self, frustum, origin) -> np.ndarray:
        out = np.zeros(1, FRUSTUM)
        self.f_get_relative(_ptr(frustum), _ptr(_f64x3(origin)), _ptr(out))
        return out

    # ---- transforms -------------------------------------------------------------------------------------
    def bone_attachment(self, parent, bone_pos, bone_rot, relative, original_scale) -> np.ndarray:
        """updateBoneAttachment (render_module.cpp:396-402) per row: parent Transform[n], bone pose [n,3]/[n,4], relative LocalRigidTransform[n], scale [n,3]."""
        parent = np.ascontiguousarray(parent, TRANSFORM)
        bp, br = np.ascontiguousarray(bone_pos, np.float32).reshape(-1, 3), np.ascontiguousarray(bone_rot, np.float32).reshape(-1, 4)
        rel = np.ascontiguousarray(relative, LOCAL_RIGID)
        sc = np.ascontiguousarray(original_scale, np.float32).reshape(-1, 3)
        out = np.zeros(len(parent), TRANSFORM)
        for i in range(len(parent)):
            self.f_bone_attachment(_ptr(parent[i : i + 1]), _ptr(bp[i : i + 1]), _ptr(br[i : i + 1]), _ptr(rel[i : i + 1]), _ptr(sc[i : i + 1]), _ptr(out[i : i + 1]))
        return out

    def compose(self, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        out = np.zeros(len(a), TRANSFORM)
        for i in range(len(a)):
            self.f_compose(_ptr(a[i : i + 1]), _ptr(b[i : i + 1]), _ptr(out[i : i + 1]))
        return out

    def compute_local(self, parent: np.ndarray, child: np.ndarray) -> np.ndarray:
        out = np.zeros(len(parent), TRANSFORM)
        for i in range(len(parent)):
            self.f_compute_local(_ptr(parent[i : i + 1]), _ptr(child[i : i + 1]), _ptr(out[i : i + 1]))
        return out

    # ---- pose / skin ------------------------------------------------------------------------------------
    def pose_compute_absolute(self, positions, rotations, parents, first_nonroot, n_threads=1):
        """positions [I, B, 3] f32, rotations [I, B, 4] f32 (copied); returns absolute (positions, rotations)."""
        pos = np.array(positions, dtype=np.float32, order="C", copy=True)
        rot = np.array(rotations, dtype=np.float32, order="C", copy=True)
        par = np.ascontiguousarray(parents, dtype=np.int16)
        n_inst, count = pos.shape[0], pos.shape[1]
        self.f_pose_compute_absolute(_ptr(pos), _ptr(rot), _ptr(par), int(first_nonroot), count, n_inst, n_threads)
        return pos, rot

    def pose_blend(self, positions, rotations, rhs_positions, rhs_rotations, weight):
        """Pose::blend (pose.cpp:30-41): returns blended copies of (positions [..., 3], rotations [..., 4])."""
        pos = np.array(positions, dtype=np.float32, order="C", copy=True)
        rot = np.array(rotations, dtype=np.float32, order="C", copy=True)
        rp, rr = np.ascontiguousarray(rhs_positions, np.float32), np.ascontiguousarray(rhs_rotations, np.float32)
        self.f_pose_blend(_ptr(pos), _ptr(rot), _ptr(rp), _ptr(rr), pos.size // 3, float(weight))
        return pos, rot

    def invert_bind(self, bind: np.ndarray) -> np.ndarray:
        bind = np.ascontiguousarray(bind, dtype=LOCAL_RIGID)
        out = np.zeros(len(bind), LOCAL_RIGID)
        self.f_invert_bind(_ptr(bind), _ptr(out), len(bind))
        return out

    def skin_matrices(self, pose_pos, pose_rot, inv_bind, n_threads=1) -> np.ndarray:
        pos = np.ascontiguousarray(pose_pos, dtype=np.float32)
        rot = np.ascontiguousarray(pose_rot, dtype=np.float32)
        inv = np.ascontiguousarray(inv_bind, dtype=LOCAL_RIGID)
        n_inst, count = pos.shape[0], pos.shape[1]
        out = np.zeros((n_inst, count), MATRIX)
        self.f_skin_matrices(_ptr(pos), _ptr(rot), _ptr(inv), _ptr(out), count, n_inst, n_threads)
        return out

    def dual_quats(self, pose_pos, pose_rot, inv_bind) -> np.ndarray:
        """[I, B, 8] float32: DualQuat {r.xyzw, d.xyzw} palette of computeSkeletonDualQuats (pipeline.cpp:2680-2745)."""
        pos = np.ascontiguousarray(pose_pos, dtype=np.float32)
        rot = np.ascontiguousarray(pose_rot, dtype=np.float32)
        inv = np.ascontiguousarray(inv_bind, dtype=LOCAL_RIGID)
        n_inst, count = pos.shape[0], pos.shape[1]
        out = np.zeros((n_inst, count, 8), np.float32)
        self.f_dual_quats(_ptr(pos), _ptr(rot), _ptr(inv), _ptr(out), count, n_inst)
        return out

    def evaluate_skin(self, verts, skin, palettes, n_threads=1) -> np.ndarray:
        verts = np.ascontiguousarray(verts, dtype=np.float32)
        skin = np.ascontiguousarray(skin, dtype=SKIN)
        palettes = np.ascontiguousarray(palettes, dtype=MATRIX)
        n_inst, n_bones = palettes.shape[0], palettes.shape[1]
        out = np.zeros((n_inst, len(verts), 3), np.float32)
        self.f_evaluate_skin(_ptr(verts), _ptr(skin), _ptr(palettes), _ptr(out), len(verts), n_bones, n_inst, n_threads)
        return out

    def create_sort_keys(self, kv, max_sort_key, mesh_ids, decal_ids, curve_ids, sc, pos_xyz, lod=None, pose_frame=None):
        """PipelineImpl::createSortKeys, single worker (pipeline.cpp:3789-3968). `sc` = lumixengine_amd.scenes.keys_scene tables.
        Returns a dict: keys, values (insertion order, AUTOINSTANCED pairs last), group_offsets, group_values, poses, dirty, lod, pose_frame."""
        if self.f_create_sort_keys is None:
            raise RuntimeError("this oracle library has no create_sort_keys (rebuild oracle/_ref)")
        from lumixengine_amd.api import KEYS_VIEW, KEYS_MODEL, MESH_MATERIAL
        kv = np.ascontiguousarray(kv, KEYS_VIEW)
        ids = [np.ascontiguousarray(x, np.int32) for x in (mesh_ids, decal_ids, curve_ids)]
        models = np.ascontiguousarray(sc["models"], KEYS_MODEL)
        mm = np.ascontiguousarray(sc["mesh_materials"], MESH_MATERIAL)
        lod = np.array(sc["lod"] if lod is None else lod, np.float32)
        pose_frame = np.array(sc["pose_frame"] if pose_frame is None else pose_frame, np.uint32)
        span = int(max(1, (models["lod_indices"]["to"] - models["lod_indices"]["from"] + 1).max()))
        cap_inst = len(ids[0]) * span * 2 + 1
        cap_pairs = cap_inst + len(ids[1]) + len(ids[2]) + max_sort_key + 2
        keys, values = np.zeros(cap_pairs, np.uint64), np.zeros(cap_pairs, np.uint64)
        offsets, gvalues = np.zeros(max_sort_key + 2, np.uint32), np.zeros(cap_inst, np.uint64)
        poses, dirty = np.zeros(len(ids[0]) + 1, np.int32), np.zeros(len(ids[0]) + 1, np.int32)

        class Out(C.Structure):
            _fields_ = [("keys", C.c_void_p), ("values", C.c_void_p), ("cap_pairs", C.c_uint32), ("n_pairs", C.c_uint32), ("group_offsets", C.c_void_p),
                        ("group_values", C.c_void_p), ("cap_instanced", C.c_uint32), ("n_instanced", C.c_uint32), ("poses", C.c_void_p), ("n_poses", C.c_uint32),
                        ("dirty", C.c_void_p), ("n_dirty", C.c_uint32), ("n_groups", C.c_uint32)]

        out = Out(_ptr(keys), _ptr(values), cap_pairs, 0, _ptr(offsets), _ptr(gvalues), cap_inst, 0, _ptr(poses), 0, _ptr(dirty), 0, 0)
        arrs = [np.ascontiguousarray(sc["mesh_types"], np.uint8), np.ascontiguousarray(sc["model"], np.int32), np.ascontiguousarray(sc["material_offset"], np.uint32),
                np.ascontiguousarray(sc["flags"], np.uint8), np.ascontiguousarray(sc["dirty"], np.uint8), np.ascontiguousarray(sc["decal_key"], np.uint32),
                np.ascontiguousarray(sc["decal_layer"], np.uint8), np.ascontiguousarray(sc["curve_key"], np.uint32), np.ascontiguousarray(sc["curve_layer"], np.uint8),
                np.ascontiguousarray(pos_xyz, np.float64)]
        rc = self.f_create_sort_keys(_ptr(kv), max_sort_key, _ptr(ids[0]), len(ids[0]), _ptr(ids[1]), len(ids[1]), _ptr(ids[2]), len(ids[2]), _ptr(models),
                                     _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), _ptr(mm), _ptr(lod), _ptr(arrs[3]), _ptr(arrs[4]), _ptr(pose_frame), _ptr(arrs[5]),
                                     _ptr(arrs[6]), _ptr(arrs[7]), _ptr(arrs[8]), _ptr(arrs[9]), C.addressof(out))
        if rc != 0:
            raise RuntimeError(f"{self.prefix}create_sort_keys failed with {rc}")
        return {"keys": keys[: out.n_pairs], "values": values[: out.n_pairs], "group_offsets": offsets, "group_values": gvalues[: out.n_instanced],
                "poses": poses[: out.n_poses], "dirty": dirty[: out.n_dirty], "lod": lod, "pose_frame": pose_frame, "groups": out.n_groups}

    def evaluate_dq_skin(self, verts, skin, dual_quats) -> np.ndarray:
        """Dual-quaternion vertex blend of the reference's shader (surface_base.hlsli:196-217); dual_quats [n_inst, n_bones, 8]. Port oracle only."""
        verts = np.ascontiguousarray(verts, np.float32)
        skin = np.ascontiguousarray(skin, SKIN)
        dq = np.ascontiguousarray(dual_quats, np.float32)
        n_inst, n_bones = dq.shape[0], dq.shape[1]
        out = np.zeros((n_inst, len(verts), 3), np.float32)
        f = self.lib.orc_evaluate_dq_skin
        f.restype, f.argtypes = None, [C.c_void_p] * 4 + [C.c_uint32] * 3
        f(_ptr(verts), _ptr(skin), _ptr(dq), _ptr(out), len(verts), n_bones, n_inst)
        return out

    def evaluate_dq_skin_hlsl(self, verts, skin, dual_quats) -> np.ndarray:
        """The same blend from the reference's OWN shader text (surface_base.hlsli:197-205 + transformByDualQuat, common.hlsli:632-636),
        sliced at build time and compiled as C++ (oracle/ref/slice_hlsl.py, hlsl_shim.cpp). Reference oracle only."""
        if self.kind != "reference":
            raise RuntimeError("the sliced shader code lives in oracle/_ref/liblmx_ref.so: Oracle('reference')")
        verts = np.ascontiguousarray(verts, np.float32)
        skin = np.ascontiguousarray(skin, SKIN)
        dq = np.ascontiguousarray(dual_quats, np.float32)
        n_inst, n_bones = dq.shape[0], dq.shape[1]
        out = np.zeros((n_inst, len(verts), 3), np.float32)
        f = self.lib.ref_hlsl_dq_skin
        f.restype, f.argtypes = None, [C.c_void_p] * 4 + [C.c_uint32] * 3
        f(_ptr(verts), _ptr(skin), _ptr(dq), _ptr(out), len(verts), n_bones, n_inst)
        return out

    def nlerp(self, q1, q2, t) -> np.ndarray:
        """simd_nlerp, core/simd_math.h:107-123 (port oracle only)."""
        q1, q2, t = np.ascontiguousarray(q1, np.float32).reshape(-1, 4), np.ascontiguousarray(q2, np.float32).reshape(-1, 4), np.ascontiguousarray(t, np.float32)
        out = np.zeros_like(q1)
        f = self.lib.orc_nlerp
        f.restype, f.argtypes = None, [C.c_void_p] * 4 + [C.c_uint32]
        f(_ptr(q1), _ptr(q2), _ptr(t), _ptr(out), len(q1))
        return out

    def update_animables(self, anims, anim_of_instance, times, time_delta, weight, model_relative):
        """AnimationModuleImpl::updateAnimable per instance (animation_module.cpp:439-472): returns (pos [n, bones, 3], rot [n, bones, 4],
        new times). `anims` = list of lumixengine_amd.scenes.animation dicts, anim_of_instance < 0 or 0xffffffff = no animation."""
        from lumixengine_amd.api import animation_struct, LOCAL_RIGID
        # 'port': the plain-C restatement; 'reference': the reference's own AnimationSampler / simd_nlerp code, sliced out of
        # animation.cpp / simd*.h at build time and compiled into oracle/_ref (oracle/ref/anim_shim.cpp)
        f = getattr(self.lib, self.prefix + "update_animable")
        f.restype = C.c_uint32
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        structs = [animation_struct(a) for a in anims]
        rel = np.ascontiguousarray(model_relative, LOCAL_RIGID)
        nb, n = len(rel), len(times)
        pos, rot, out_t = np.zeros((n, nb, 3), np.float32), np.zeros((n, nb, 4), np.float32), np.zeros(n, np.uint32)
        for i in range(n):
            k = int(anim_of_instance[i])
            a = C.addressof(structs[k][0]) if 0 <= k < len(structs) else None
            out_t[i] = f(a, int(times[i]), float(time_delta), float(weight), _ptr(rel), nb, pos[i].ctypes.data, rot[i].ctypes.data)
        return pos, rot, out_t

    def update_animators(self, anims, stacks, model_relative, parents=None, first_nonroot=0):
        """AnimationModuleImpl::updateAnimator's pose work per Animator (animation_module.cpp:602-636): Model::getRelativePose, then
        evalBlendStack's SAMPLE instructions in order (controller.cpp:267-293) - `stacks[i]` = [(animation index, weight, time, looped), ...] -
        then, when `parents` is given, Pose::computeAbsolute. Returns (pos [n, bones, 3], rot [n, bones, 4])."""
        from lumixengine_amd.api import animation_struct, LOCAL_RIGID
        f = getattr(self.lib, self.prefix + "blend_stack_sample")
        f.restype, f.argtypes = None, [C.c_void_p, C.c_uint32, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        structs = [animation_struct(a) for a in anims]
        rel = np.ascontiguousarray(model_relative, LOCAL_RIGID)
        nb, n = len(rel), len(stacks)
        pos, rot = np.zeros((n, nb, 3), np.float32), np.zeros((n, nb, 4), np.float32)
        for i, stack in enumerate(stacks):
            pos[i], rot[i] = rel["pos"], rel["rot"]  # Model::getRelativePose, model.cpp:226-237
            for (k, w, t, looped) in stack:
                f(C.addressof(structs[k][0]), int(t), float(w), 1 if looped else 0, nb, pos[i].ctypes.data, rot[i].ctypes.data)
            if parents is not None:
                ap, ar = self.pose_compute_absolute(pos[i : i + 1], rot[i : i + 1], parents, first_nonroot)
                pos[i], rot[i] = ap[0], ar[0]
        return pos, rot

    def rand_fill(self, u: int, v: int, n: int) -> np.ndarray:
        out = np.zeros(n, np.uint32)
        self.f_rand_fill(u, v, n, _ptr(out))
        return out

    def culling_system(self) -> "OracleCullingSystem":
        return OracleCullingSystem(self)

    def world(self, n_entities: int) -> "OracleWorld":
        return OracleWorld(self, n_entities)


class OracleCullingSystem:
    """CullingSystem (renderer/culling_system.h:58-77) on the CPU oracle."""

    def __init__(self, oracle: Oracle):
        self.o = oracle
        self.h = oracle.f_cs_create()

    def __del__(self):
        if getattr(self, "h", None):
            self.o.f_cs_destroy(self.h)
            self.h = None

    def add(self, entity, type_, pos, radius):
        self.o.f_cs_add(self.h, int(entity), int(type_), _ptr(_f64x3(pos)), float(radius))

    def add_bulk(self, entity, type_, pos, radius):
        entity = np.ascontiguousarray(entity, np.int32)
        type_ = np.ascontiguousarray(type_, np.uint8)
        pos = np.ascontiguousarray(pos, np.float64)
        radius = np.ascontiguousarray(radius, np.float32)
        self.o.f_cs_add_bulk(self.h, len(entity), _ptr(entity), _ptr(type_), _ptr(pos), _ptr(radius))

    def remove(self, entity):
        self.o.f_cs_remove(self.h, int(entity))

    def set(self, entity, pos, radius):
        self.o.f_cs_set(self.h, int(entity), _ptr(_f64x3(pos)), float(radius))

    def set_position(self, entity, pos):
        self.o.f_cs_set_position(self.h, int(entity), _ptr(_f64x3(pos)))

    def set_radius(self, entity, radius):
        self.o.f_cs_set_radius(self.h, int(entity), float(radius))

    def get_radius(self, entity) -> float:
        return float(self.o.f_cs_get_radius(self.h, int(entity)))

    def is_added(self, entity) -> bool:
        return bool(self.o.f_cs_is_added(self.h, int(entity)))

    def cell_count(self) -> int:
        return int(self.o.f_cs_cell_count(self.h))

    def cull(self, frustum: np.ndarray, type_: int = 0xFF, n_threads: int = 1, cap: Optional[int] = None, want_ids=True):
        """Returns (ids[int32], types[uint8], n_result_pages). Order is unspecified, like the reference's."""
        if cap is None:
            cap = int(self.o.f_cs_cull(self.h, _ptr(frustum), type_, n_threads, None, None, 0, None))
        ids = np.zeros(cap, np.int32) if want_ids else None
        types = np.zeros(cap, np.uint8) if want_ids else None
        pages = C.c_uint32(0)
        n = int(self.o.f_cs_cull(self.h, _ptr(frustum), type_, n_threads, _ptr(ids), _ptr(types), cap, C.byref(pages)))
        if want_ids:
            return ids[: min(n, cap)], types[: min(n, cap)], pages.value
        return n, pages.value


class OracleWorld:
    """World transform/hierarchy subset (engine/world.cpp) on the CPU oracle."""

    def __init__(self, oracle: Oracle, n_entities: int):
        self.o = oracle
        self.n = n_entities
        self.h = oracle.f_world_create(n_entities)
        self._cs = None

    def __del__(self):
        if getattr(self, "h", None):
            self.o.f_world_destroy(self.h)
            self.h = None

    def init_transforms(self, entity, tr):
        entity = np.ascontiguousarray(entity, np.int32)
        tr = np.ascontiguousarray(tr, TRANSFORM)
        self.o.f_world_init_transforms(self.h, len(entity), _ptr(entity), _ptr(tr))

    def set_parents(self, parent, child):
        parent = np.ascontiguousarray(parent, np.int32)
        child = np.ascontiguousarray(child, np.int32)
        self.o.f_world_set_parents(self.h, len(child), _ptr(parent), _ptr(child))

    def set_transforms(self, entity, tr):
        entity = np.ascontiguousarray(entity, np.int32)
        tr = np.ascontiguousarray(tr, TRANSFORM)
        self.o.f_world_set_transforms(self.h, len(entity), _ptr(entity), _ptr(tr))

    def set_local_transforms(self, entity, tr):
        entity = np.ascontiguousarray(entity, np.int32)
        tr = np.ascontiguousarray(tr, TRANSFORM)
        self.o.f_world_set_local_transforms(self.h, len(entity), _ptr(entity), _ptr(tr))

    def get_transforms(self) -> np.ndarray:
        out = np.zeros(self.n, TRANSFORM)
        self.o.f_world_get_transforms(self.h, self.n, _ptr(out))
        return out

    def get_local_transforms(self) -> np.ndarray:
        out = np.zeros(self.n, TRANSFORM)
        self.o.f_world_get_local_transforms(self.h, self.n, _ptr(out))
        return out

    def serialize(self, flags: int = 0) -> bytes:
        """World::serialize (engine/world.cpp:837-897) - reference oracle only (the real World)."""
        f = self.o.lib.ref_world_serialize
        f.restype, f.argtypes = C.c_uint32, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        n = f(self.h, flags, None, 0)
        buf = C.create_string_buffer(n)
        assert f(self.h, flags, buf, n) == n
        return buf.raw

    def set_name(self, entity: int, name: str):
        f = self.o.lib.ref_world_set_name
        f.restype, f.argtypes = None, [C.c_void_p, C.c_int32, C.c_char_p]
        f(self.h, int(entity), name.encode())

    def destroy_entity(self, entity: int):
        f = self.o.lib.ref_world_destroy_entity
        f.restype, f.argtypes = None, [C.c_void_p, C.c_int32]
        f(self.h, int(entity))

    def bind_culling(self, cs: OracleCullingSystem, entity, model_radius):
        entity = np.ascontiguousarray(entity, np.int32)
        model_radius = np.ascontiguousarray(model_radius, np.float32)
        self._cs = cs
        self.o.f_world_bind_culling(self.h, cs.h, len(entity), _ptr(entity), _ptr(model_radius))


def have_reference() -> bool:
    return os.path.exists(REF_SO)
