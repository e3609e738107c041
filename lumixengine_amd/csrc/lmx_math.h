// lmx_math.h — arithmetic contract of the cull / transform / skin kernels, usable from HIP device code and host C++.
//
// Every function reproduces the operation ORDER of the LumixEngine routine it cites (paths relative to the
// reference tree, src/...), because the parity bar for visibility and world transforms is bit-exactness against
// the reference CPU path. All translation units that include this header are compiled with -ffp-contract=off
// (no FMA contraction); nothing here may be rewritten with fmaf()/__fmaf_rn or reassociated.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LMX_HD __host__ __device__ __forceinline__
#else
#define LMX_HD inline
#endif

namespace lmx {

struct V3 { float x, y, z; };
struct DV3 { double x, y, z; };
struct Q4 { float x, y, z, w; };
struct IV3 { int32_t x, y, z; };

// ---- core/math.cpp vector primitives --------------------------------------------------------------------
LMX_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
LMX_HD V3 add(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }                 // math.cpp:444-446
LMX_HD V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }                 // math.cpp:452-454
LMX_HD V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }                                       // math.cpp:448-450
LMX_HD V3 mul(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }                     // math.cpp:456-458
LMX_HD V3 cross(V3 a, V3 b) {                                                               // math.cpp:1274-1276
	return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
LMX_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                 // math.cpp:1266-1268

LMX_HD DV3 dv3(double x, double y, double z) { return DV3{x, y, z}; }
LMX_HD DV3 add(DV3 a, DV3 b) { return DV3{a.x + b.x, a.y + b.y, a.z + b.z}; }             // math.cpp:510
LMX_HD DV3 sub(DV3 a, DV3 b) { return DV3{a.x - b.x, a.y - b.y, a.z - b.z}; }             // math.cpp:508
LMX_HD DV3 add(DV3 a, V3 b) { return DV3{a.x + b.x, a.y + b.y, a.z + b.z}; }              // math.cpp:514
LMX_HD DV3 sub(DV3 a, V3 b) { return DV3{a.x - b.x, a.y - b.y, a.z - b.z}; }              // math.cpp:512
LMX_HD DV3 mul(DV3 a, V3 s) { return DV3{a.x * s.x, a.y * s.y, a.z * s.z}; }              // math.cpp:498
LMX_HD DV3 mul(DV3 a, double s) { return DV3{a.x * s, a.y * s, a.z * s}; }                // math.cpp:516
LMX_HD DV3 cross(DV3 a, DV3 b) {                                                            // math.cpp:1278-1280
	return DV3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
LMX_HD V3 to_v3(DV3 a) { return V3{(float)a.x, (float)a.y, (float)a.z}; }                  // math.cpp:526-530

LMX_HD Q4 qmul(Q4 a, Q4 r) {                                                                // math.cpp:694-700
	return Q4{a.w * r.x + r.w * a.x + a.y * r.z - r.y * a.z,
		a.w * r.y + r.w * a.y + a.z * r.x - r.z * a.x,
		a.w * r.z + r.w * a.z + a.x * r.y - r.x * a.y,
		a.w * r.w - a.x * r.x - a.y * r.y - a.z * r.z};
}
LMX_HD Q4 conjugated(Q4 q) { return Q4{q.x, q.y, q.z, -q.w}; }                             // math.cpp:664-667 (negates w)
LMX_HD V3 rotate(Q4 q, V3 v) {                                                              // math.cpp:164-175
	const V3 qvec = V3{q.x, q.y, q.z};
	V3 uv = cross(qvec, v);
	V3 uuv = cross(qvec, uv);
	uv = mul(uv, 2.0f * q.w);
	uuv = mul(uuv, 2.0f);
	return add(add(v, uv), uuv);
}
LMX_HD DV3 rotate(Q4 q, DV3 v) {                                                            // math.cpp:177-188
	const DV3 qvec = DV3{(double)q.x, (double)q.y, (double)q.z};
	DV3 uv = cross(qvec, v);
	DV3 uuv = cross(qvec, uv);
	uv = mul(uv, 2.0 * (double)q.w);
	uuv = mul(uuv, 2.0);
	return add(add(v, uv), uuv);
}

// ---- Transform::compose, math.cpp:801-807 ---------------------------------------------------------------
struct Xform { DV3 pos; Q4 rot; V3 scale; };
LMX_HD Xform compose(const Xform& a, const Xform& rhs) {
	Xform r;
	r.pos = add(rotate(a.rot, mul(rhs.pos, a.scale)), a.pos);
	r.rot = qmul(a.rot, rhs.rot);
	r.scale = V3{a.scale.x * rhs.scale.x, a.scale.y * rhs.scale.y, a.scale.z * rhs.scale.z}; // math.cpp:459-461
	return r;
}
// RenderModuleImpl::updateBoneAttachment (render_module.cpp:377-404): parent_entity_transform.compose(bone * relative) with
// the attached entity's own scale kept. LocalRigidTransform::operator* math.cpp:859-861; Transform::compose(LocalRigidTransform)
// math.cpp:763: {pos + rot.rotate(rhs.pos * scale), rot * rhs.rot, scale}; Vec3 * Vec3 is element-wise (math.cpp:459-461).
LMX_HD Xform bone_attachment(const Xform& parent, V3 bone_pos, Q4 bone_rot, V3 rel_pos, Q4 rel_rot, V3 original_scale) {
	const V3 bt_pos = add(rotate(bone_rot, rel_pos), bone_pos);
	const Q4 bt_rot = qmul(bone_rot, rel_rot);
	Xform r;
	r.pos = add(parent.pos, rotate(parent.rot, V3{bt_pos.x * parent.scale.x, bt_pos.y * parent.scale.y, bt_pos.z * parent.scale.z}));
	r.rot = qmul(parent.rot, bt_rot);
	r.scale = original_scale;
	return r;
}

// Transform::computeLocal, math.cpp:809-816 (setParent / re-parenting on the host; k_xform_level for entities written through
// World::setLocalTransform / setTransform, which re-derive the stored local)
LMX_HD Xform compute_local(const Xform& parent, const Xform& child) {
	const Q4 conj = conjugated(parent.rot);
	const DV3 rp = rotate(conj, DV3{-parent.pos.x, -parent.pos.y, -parent.pos.z});
	const DV3 inv_parent_pos = DV3{rp.x / parent.scale.x, rp.y / parent.scale.y, rp.z / parent.scale.z};
	const DV3 rc = rotate(conj, child.pos);
	Xform r;
	r.pos = add(DV3{rc.x / parent.scale.x, rc.y / parent.scale.y, rc.z / parent.scale.z}, inv_parent_pos);
	r.rot = qmul(conj, child.rot);
	r.scale = V3{child.scale.x / parent.scale.x, child.scale.y / parent.scale.y, child.scale.z / parent.scale.z};
	return r;
}
LMX_HD float maximum3(float a, float b, float c) {                                          // core/math.h:468-475
	const float mb = b > c ? b : c;
	return a > mb ? a : mb;
}

// ---- culling cells, renderer/culling_system.cpp:23-40,131-157 --------------------------------------------
constexpr float CELL_SIZE = 300.0f;                                                          // culling_system.cpp:75
// CellIndices ctor: IVec3(pos * (1 / cell_size)) — double * float -> double, C cast truncates toward zero. The cast is
// x86's cvttsd2si on the reference's platforms: NaN and values outside int32 give INT32_MIN ("integer indefinite"). The GPU's
// v_cvt_i32_f64 saturates instead (NaN -> 0), so the conversion is spelled out to keep host mirror, oracle and kernels identical.
LMX_HD int32_t trunc_i32(double v) {
	return (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : (int32_t)0x80000000u;
}
LMX_HD IV3 cell_of(DV3 pos) {
	const float inv = 1 / CELL_SIZE;
	return IV3{trunc_i32(pos.x * inv), trunc_i32(pos.y * inv), trunc_i32(pos.z * inv)};
}
// header.origin = i.pos * double(m_cell_size) (IVec3::operator*(double), math.cpp:149-152)
LMX_HD DV3 cell_origin(IV3 i) {
	const double cs = (double)CELL_SIZE;
	return DV3{cs * i.x, cs * i.y, cs * i.z};
}
LMX_HD bool is_big_radius(float radius) { return radius > CELL_SIZE; }                      // culling_system.cpp:140

// ---- frustum as the kernels see it -----------------------------------------------------------------------
// Only what ShiftedFrustum::{containsAABB,intersectsAABB,getRelative} read: the 6 unique plane normals, the
// frustum's own ds for the AABB tests, the corner point each plane is re-anchored on by getRelative
// (geometry.cpp:134-142: NEAR->p0, FAR->p4, LEFT->p1, RIGHT->p0, TOP->p0, BOTTOM->p2; EXTRA0/1 duplicate NEAR
// bit for bit and are therefore not evaluated), and the fp64 origin.
struct DevFrustum {
	float nx[6], ny[6], nz[6], d[6];
	float px[6], py[6], pz[6];
	float pad[2];
	double origin[3];
};

enum CellClass : uint32_t { CELL_REJECT = 0, CELL_ACCEPT = 1, CELL_TEST = 2 };

// Per-cell classification, culling_system.cpp:342-363 + geometry.cpp:99-118 / 159-178.
// Returns the class and the getRelative() offset Vec3(frustum.origin - cell_origin) (geometry.cpp:124).
// `skip`: planes (bit i) every cell of the caller's tile is known to pass in BOTH tests below (tile_plane_skip_mask): they are
// left out - the verdict is the same by construction, the tile-level bound carries the rounding margin. The planes are walked
// through the set bits of the remaining mask (a real loop with a wave-uniform trip count on the device, not six predicated bodies).
LMX_HD uint32_t classify_cell(const DevFrustum& f, IV3 idx, bool is_big, V3* out_offset, uint32_t skip = 0u) {
	const DV3 origin = cell_origin(idx);
	const DV3 forigin = DV3{f.origin[0], f.origin[1], f.origin[2]};
	*out_offset = to_v3(sub(forigin, origin));
	if (is_big) return CELL_TEST;
	const V3 cs = V3{CELL_SIZE, CELL_SIZE, CELL_SIZE};
	{ // containsAABB(origin + v3_cell_size, v3_cell_size)
		const V3 rel = to_v3(sub(add(origin, cs), forigin));
		const V3 hi = add(rel, cs);
		bool inside = true;
		for (uint32_t todo = ~skip & 63u; todo != 0; todo &= todo - 1) {
			const int i = __builtin_ctz(todo);
			const float bx = f.nx[i] < 0.0f ? hi.x : rel.x;
			const float by = f.ny[i] < 0.0f ? hi.y : rel.y;
			const float bz = f.nz[i] < 0.0f ? hi.z : rel.z;
			const float dp = (f.nx[i] * bx) + (f.ny[i] * by) + (f.nz[i] * bz);
			if (dp < -f.d[i]) inside = false;
		}
		if (inside) return CELL_ACCEPT;
	}
	{ // intersectsAABB(origin - v3_cell_size, v3_2_cell_size)
		const V3 cs2 = V3{2 * CELL_SIZE, 2 * CELL_SIZE, 2 * CELL_SIZE};
		const V3 rel = to_v3(sub(sub(origin, cs), forigin));
		const V3 hi = add(rel, cs2);
		bool hit = true;
		for (uint32_t todo = ~skip & 63u; todo != 0; todo &= todo - 1) {
			const int i = __builtin_ctz(todo);
			const float bx = f.nx[i] > 0.0f ? hi.x : rel.x;
			const float by = f.ny[i] > 0.0f ? hi.y : rel.y;
			const float bz = f.nz[i] > 0.0f ? hi.z : rel.z;
			const float dp = (f.nx[i] * bx) + (f.ny[i] * by) + (f.nz[i] * bz);
			if (dp < -f.d[i]) hit = false;
		}
		if (hit) return CELL_TEST;
	}
	return CELL_REJECT;
}

// Tile-level tests of k_cull_tile (no reference twin: conservative bounds on classify_cell). A tile of the sorted sphere array
// covers a run of cells; `lo..hi` is the box of their cell indices.
//   * TILE_REJECT: the union of the cells' intersectsAABB boxes ([300 lo - 300, 300 hi + 300] per axis) lies behind one frustum
//     plane by more than `margin`: a cell's positive vertex is never further along the plane normal than the union box's, so
//     every cell of the tile is CELL_REJECT.
//   * TILE_ACCEPT: the union of the cells' containsAABB boxes ([300 lo + 300, 300 hi + 600], the reference's +300 quirk included)
//     lies inside every plane by more than `margin`: every cell's negative vertex passes, so every cell is CELL_ACCEPT.
//   * TILE_MIXED: anything else - the cells are classified one by one.
// Corners are formed in fp64 and rounded once; the plane expression is evaluated in fp32 like the per-cell test. Error budget:
// each evaluation (this one and the per-cell one it stands in for) errs by < 6 * 2^-24 * sum_i |n_i b_i| <= 3.6e-7 * |n|_inf *
// B1, where B1 = sum over axes of the box's largest |coordinate| bounds |b|_1 of EVERY cell vertex inside the union box (not just
// of the vertex picked here), and the comparison's right-hand side -d -+ margin rounds by <= 2^-24 (|d| + margin). The margin
// max(1, |n|_1) * (2 + 4e-6 * B1) + 1e-6 * |d| covers the sum several times over for any plane scale. Tiles holding big-sphere cells
// (always CELL_TEST) are TILE_MIXED; NaN planes / corners compare false and give TILE_MIXED.
struct TileBox { int32_t lo[3], hi[3]; uint32_t flags, pad; };
enum : uint32_t { TILE_EMPTY = 1, TILE_HAS_BIG = 2, TILE_DENSE = 4 }; // TILE_DENSE: every slot of the tile holds a live id (no padding, no tombstone)
enum TileStatus : uint32_t { TILE_REJECT = 0, TILE_ACCEPT = 1, TILE_MIXED = 2 };
LMX_HD float abs_f(float v) { return v < 0 ? -v : v; }
LMX_HD float max_f(float a, float b) { return a > b ? a : b; }
LMX_HD uint32_t tile_status(const DevFrustum& f, const TileBox& b) {
	if (b.flags & TILE_EMPTY) return TILE_REJECT;
	if (b.flags & TILE_HAS_BIG) return TILE_MIXED;
	const double cs = (double)CELL_SIZE;
	// intersectsAABB union box
	const float lx = (float)(cs * b.lo[0] - cs - f.origin[0]), ly = (float)(cs * b.lo[1] - cs - f.origin[1]), lz = (float)(cs * b.lo[2] - cs - f.origin[2]);
	const float hx = (float)(cs * b.hi[0] + cs - f.origin[0]), hy = (float)(cs * b.hi[1] + cs - f.origin[1]), hz = (float)(cs * b.hi[2] + cs - f.origin[2]);
	// containsAABB union box
	const float clx = (float)(cs * b.lo[0] + cs - f.origin[0]), cly = (float)(cs * b.lo[1] + cs - f.origin[1]), clz = (float)(cs * b.lo[2] + cs - f.origin[2]);
	const float chx = (float)(cs * b.hi[0] + 2 * cs - f.origin[0]), chy = (float)(cs * b.hi[1] + 2 * cs - f.origin[1]), chz = (float)(cs * b.hi[2] + 2 * cs - f.origin[2]);
	const float b1 = max_f(abs_f(lx), abs_f(chx)) + max_f(abs_f(ly), abs_f(chy)) + max_f(abs_f(lz), abs_f(chz));
	bool inside = true;
	for (int i = 0; i < 6; ++i) {
		const float n1 = abs_f(f.nx[i]) + abs_f(f.ny[i]) + abs_f(f.nz[i]);
		const float margin = max_f(1.0f, n1) * (2.0f + 4e-6f * b1) + 1e-6f * abs_f(f.d[i]);
		{ // positive vertex of the intersects box
			const float bx = f.nx[i] > 0.0f ? hx : lx, by = f.ny[i] > 0.0f ? hy : ly, bz = f.nz[i] > 0.0f ? hz : lz;
			const float dp = (f.nx[i] * bx) + (f.ny[i] * by) + (f.nz[i] * bz);
			if (dp < -f.d[i] - margin) return TILE_REJECT;
		}
		{ // negative vertex of the contains box
			const float bx = f.nx[i] < 0.0f ? chx : clx, by = f.ny[i] < 0.0f ? chy : cly, bz = f.nz[i] < 0.0f ? chz : clz;
			const float dp = (f.nx[i] * bx) + (f.ny[i] * by) + (f.nz[i] * bz);
			if (!(dp > -f.d[i] + margin)) inside = false;
		}
	}
	return inside ? TILE_ACCEPT : TILE_MIXED;
}
LMX_HD bool tile_rejected(const DevFrustum& f, const TileBox& b) { return tile_status(f, b) == TILE_REJECT; }
// Planes (bit i) that every cell of a MIXED tile passes in both of classify_cell's tests: the NEGATIVE vertex (w.r.t. the plane's
// normal) of the union of all the cells' containsAABB and intersectsAABB boxes ([300 lo - 300, 300 hi + 600] per axis) lies inside
// the plane by more than the margin of tile_status() - every vertex either per-cell test picks for that plane lies further inside.
// Phase A of k_cull_tile then evaluates the remaining planes only (a frustum much larger than a tile: one or two of six).
LMX_HD uint32_t tile_plane_skip_mask(const DevFrustum& f, const TileBox& b) {
	if (b.flags & (TILE_EMPTY | TILE_HAS_BIG)) return 0u;
	const double cs = (double)CELL_SIZE;
	const float lx = (float)(cs * b.lo[0] - cs - f.origin[0]), ly = (float)(cs * b.lo[1] - cs - f.origin[1]), lz = (float)(cs * b.lo[2] - cs - f.origin[2]);
	const float chx = (float)(cs * b.hi[0] + 2 * cs - f.origin[0]), chy = (float)(cs * b.hi[1] + 2 * cs - f.origin[1]), chz = (float)(cs * b.hi[2] + 2 * cs - f.origin[2]);
	const float b1 = max_f(abs_f(lx), abs_f(chx)) + max_f(abs_f(ly), abs_f(chy)) + max_f(abs_f(lz), abs_f(chz));
	uint32_t mask = 0;
	for (int i = 0; i < 6; ++i) {
		const float n1 = abs_f(f.nx[i]) + abs_f(f.ny[i]) + abs_f(f.nz[i]);
		const float margin = max_f(1.0f, n1) * (2.0f + 4e-6f * b1) + 1e-6f * abs_f(f.d[i]);
		const float bx = f.nx[i] < 0.0f ? chx : lx, by = f.ny[i] < 0.0f ? chy : ly, bz = f.nz[i] < 0.0f ? chz : lz;
		const float dp = (f.nx[i] * bx) + (f.ny[i] * by) + (f.nz[i] * bz);
		if (dp > -f.d[i] + margin) mask |= 1u << i;
	}
	return mask;
}

// ShiftedFrustum::getRelative for one plane (geometry.cpp:121-149, setPlane :421-427): the plane is re-anchored on its corner
// point shifted by offset = Vec3(frustum.origin - cell_origin): d_k = -dot(point_k + offset, n_k). Per cell, not per sphere.
LMX_HD float relative_plane_d(const DevFrustum& f, V3 offset, int k) {
	const V3 n = V3{f.nx[k], f.ny[k], f.nz[k]};
	const V3 q = add(V3{f.px[k], f.py[k], f.pz[k]}, offset);
	return -dot(q, n);
}
// doCulling for one sphere, culling_system.cpp:283-306, against the cell-relative planes:
// t = ((cx*nx + cy*ny) + cz*nz) + d; t = t - (-r); culled iff t < 0 (scalar f4MoveMask, simd.h:332-338).
LMX_HD bool sphere_visible_d(const DevFrustum& f, const float d[6], float cx, float cy, float cz, float radius) {
	const float r = -radius;
	bool culled = false;
	for (int k = 0; k < 6; ++k) {
		float t = cx * f.nx[k] + cy * f.ny[k] + cz * f.nz[k] + d[k];
		t = t - r;
		culled = culled || (t < 0);
	}
	return !culled;
}
LMX_HD bool sphere_visible(const DevFrustum& f, V3 offset, float cx, float cy, float cz, float radius) {
	float d[6];
	for (int k = 0; k < 6; ++k) d[k] = relative_plane_d(f, offset, k);
	return sphere_visible_d(f, d, cx, cy, cz, radius);
}

// ---- pose / palette / skin -------------------------------------------------------------------------------
struct Mat4 { float c[4][4]; }; // column-major, c[col][row], core/math.h:329-393

// (LocalRigidTransform{pose_pos, pose_rot} * inv_bind).toMatrix(), model.cpp:132-137;
// operator* math.cpp:859-861; Matrix(pos, rot) math.cpp:887-890; Quat::toMatrix math.cpp:727-756
LMX_HD Mat4 skin_matrix(V3 pose_pos, Q4 pose_rot, V3 inv_pos, Q4 inv_rot) {
	const V3 p = add(rotate(pose_rot, inv_pos), pose_pos);
	const Q4 q = qmul(pose_rot, inv_rot);
	const float fx = q.x + q.x, fy = q.y + q.y, fz = q.z + q.z;
	const float fwx = fx * q.w, fwy = fy * q.w, fwz = fz * q.w;
	const float fxx = fx * q.x, fxy = fy * q.x, fxz = fz * q.x;
	const float fyy = fy * q.y, fyz = fz * q.y, fzz = fz * q.z;
	Mat4 m;
	m.c[0][0] = 1.0f - (fyy + fzz);
	m.c[1][0] = fxy - fwz;
	m.c[2][0] = fxz + fwy;
	m.c[0][1] = fxy + fwz;
	m.c[1][1] = 1.0f - (fxx + fzz);
	m.c[2][1] = fyz - fwx;
	m.c[0][2] = fxz - fwy;
	m.c[1][2] = fyz + fwx;
	m.c[2][2] = 1.0f - (fxx + fyy);
	m.c[0][3] = 0; m.c[1][3] = 0; m.c[2][3] = 0;
	m.c[3][0] = p.x; m.c[3][1] = p.y; m.c[3][2] = p.z; m.c[3][3] = 1;
	return m;
}

// (LocalRigidTransform{pose_pos, pose_rot} * inv_bind).toDualQuat(): the palette entry the reference's GPU skinning path
// uploads (PipelineImpl::computeSkeletonDualQuats, renderer/pipeline.cpp:2680-2745; toDualQuat math.cpp:843-853)
struct DualQ { Q4 r; Q4 d; };
LMX_HD DualQ skin_dual_quat(V3 pose_pos, Q4 pose_rot, V3 inv_pos, Q4 inv_rot) {
	const V3 p = add(rotate(pose_rot, inv_pos), pose_pos);
	const Q4 q = qmul(pose_rot, inv_rot);
	DualQ res;
	res.r = q;
	res.d = Q4{0.5f * (p.x * q.w + p.y * q.z - p.z * q.y),
		0.5f * (-p.x * q.z + p.y * q.w + p.z * q.x),
		0.5f * (p.x * q.y - p.y * q.x + p.z * q.w),
		-0.5f * (p.x * q.x + p.y * q.y + p.z * q.z)};
	return res;
}

// invert(LocalRigidTransform), model.cpp:24-30 (load time)
LMX_HD void invert_rigid(V3 pos, Q4 rot, V3* out_pos, Q4* out_rot) {
	*out_rot = conjugated(rot);
	*out_pos = rotate(*out_rot, neg(pos));
}

} // namespace lmx
