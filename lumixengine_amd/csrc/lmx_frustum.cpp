// lmx_frustum.cpp — host mirror of the reference's frustum construction (per view, not per entity).
//
// An engine-side adapter hands the finished 256-byte ShiftedFrustum straight to lmx_cull(); these entry points exist
// for standalone hosts (bench.py, tools) that have no LumixEngine to build one. Operation order follows
// src/core/geometry.cpp exactly (computePerspective :502-533, computeOrtho :390-409, setPoints :354-382,
// setPlanesFromPoints :339-352, Viewport::getFrustum :793-818) so that the planes are bit-identical to the engine's.
// Compiled with -ffp-contract=off.
#include <cmath>
#include <cstring>

#include "lumix_mi355.h"
#include "lmx_math.h"

using namespace lmx;

namespace {

V3 normalize3(V3 v) { // core/math.cpp:367-376
	float x = v.x, y = v.y, z = v.z;
	const float inv_len = 1 / sqrtf(x * x + y * y + z * z);
	x *= inv_len;
	y *= inv_len;
	z *= inv_len;
	return V3{x, y, z};
}

struct Builder {
	LmxShiftedFrustum* f;
	V3 p[8];

	void plane(int side, V3 normal, V3 point) { // setPlane, geometry.cpp:421-427
		f->xs[side] = normal.x;
		f->ys[side] = normal.y;
		f->zs[side] = normal.z;
		f->ds[side] = -dot(point, normal);
	}

	void planes_from_points() { // geometry.cpp:339-352
		const V3 normal_near = neg(normalize3(cross(sub(p[0], p[1]), sub(p[0], p[2]))));
		const V3 normal_far = normalize3(cross(sub(p[4], p[5]), sub(p[4], p[6])));
		plane(LMX_PLANE_EXTRA0, normal_near, p[0]);
		plane(LMX_PLANE_EXTRA1, normal_near, p[0]);
		plane(LMX_PLANE_NEAR, normal_near, p[0]);
		plane(LMX_PLANE_FAR, normal_far, p[4]);
		plane(LMX_PLANE_LEFT, normalize3(cross(sub(p[1], p[2]), sub(p[1], p[5]))), p[1]);
		plane(LMX_PLANE_RIGHT, neg(normalize3(cross(sub(p[0], p[3]), sub(p[0], p[4])))), p[0]);
		plane(LMX_PLANE_TOP, normalize3(cross(sub(p[0], p[1]), sub(p[0], p[4]))), p[0]);
		plane(LMX_PLANE_BOTTOM, normalize3(cross(sub(p[2], p[3]), sub(p[2], p[6]))), p[2]);
	}

	// setPoints, geometry.cpp:354-382, with the 7-argument overloads' viewport {-1,-1}..{1,1}
	void points(V3 near_center, V3 far_center, V3 right_near, V3 up_near, V3 right_far, V3 up_far) {
		const float lo = -1, hi = 1;
		p[0] = add(add(near_center, mul(right_near, hi)), mul(up_near, hi));
		p[1] = add(add(near_center, mul(right_near, lo)), mul(up_near, hi));
		p[2] = add(add(near_center, mul(right_near, lo)), mul(up_near, lo));
		p[3] = add(add(near_center, mul(right_near, hi)), mul(up_near, lo));
		p[4] = add(add(far_center, mul(right_far, hi)), mul(up_far, hi));
		p[5] = add(add(far_center, mul(right_far, lo)), mul(up_far, hi));
		p[6] = add(add(far_center, mul(right_far, lo)), mul(up_far, lo));
		p[7] = add(add(far_center, mul(right_far, hi)), mul(up_far, lo));
		planes_from_points();
		for (int i = 0; i < 8; ++i) {
			f->points[i][0] = p[i].x;
			f->points[i][1] = p[i].y;
			f->points[i][2] = p[i].z;
		}
	}
};

void perspective(LmxShiftedFrustum* out, const double pos[3], V3 direction, V3 up, float fov, float ratio, float near_d, float far_d) {
	memset(out, 0, sizeof(*out));
	const float scale = tanf(fov * 0.5f);
	const V3 right = cross(direction, up);
	const V3 up_near = mul(mul(up, near_d), scale);
	const V3 right_near = mul(right, near_d * scale * ratio);
	const V3 up_far = mul(mul(up, far_d), scale);
	const V3 right_far = mul(right, far_d * scale * ratio);
	const V3 z = normalize3(direction);
	const V3 near_center = mul(z, near_d);
	const V3 far_center = mul(z, far_d);
	out->origin[0] = pos[0];
	out->origin[1] = pos[1];
	out->origin[2] = pos[2];
	Builder b{out, {}};
	b.points(near_center, far_center, right_near, up_near, right_far, up_far);
}

void ortho(LmxShiftedFrustum* out, const double pos[3], V3 direction, V3 up, float width, float height, float near_d, float far_d) {
	memset(out, 0, sizeof(*out));
	const V3 z = normalize3(direction);
	out->origin[0] = pos[0];
	out->origin[1] = pos[1];
	out->origin[2] = pos[2];
	const V3 near_center = mul(neg(z), near_d);
	const V3 far_center = mul(neg(z), far_d);
	const V3 x = mul(normalize3(cross(up, z)), width);
	const V3 y = mul(normalize3(cross(z, x)), height);
	Builder b{out, {}};
	b.points(near_center, far_center, x, y, x, y);
}

} // namespace

extern "C" {

int lmx_frustum_perspective(const double pos[3], const float dir[3], const float up[3], float fov, float ratio, float near_d, float far_d,
	LmxShiftedFrustum* out) {
	if (!pos || !dir || !up || !out) return LMX_ERR_INVALID_ARGUMENT;
	perspective(out, pos, V3{dir[0], dir[1], dir[2]}, V3{up[0], up[1], up[2]}, fov, ratio, near_d, far_d);
	return LMX_OK;
}

int lmx_frustum_ortho(const double pos[3], const float dir[3], const float up[3], float width, float height, float near_d, float far_d,
	LmxShiftedFrustum* out) {
	if (!pos || !dir || !up || !out) return LMX_ERR_INVALID_ARGUMENT;
	ortho(out, pos, V3{dir[0], dir[1], dir[2]}, V3{up[0], up[1], up[2]}, width, height, near_d, far_d);
	return LMX_OK;
}

int lmx_viewport_frustum(const LmxViewport* vp, LmxShiftedFrustum* out) { // Viewport::getFrustum(), geometry.cpp:793-818
	if (!vp || !out) return LMX_ERR_INVALID_ARGUMENT;
	const Q4 rot = Q4{vp->rot[0], vp->rot[1], vp->rot[2], vp->rot[3]};
	const float ratio = vp->h > 0 ? vp->w / (float)vp->h : 1;
	const double zero[3] = {0, 0, 0};
	if (vp->is_ortho) {
		ortho(out, zero, rotate(rot, V3{0, 0, 1}), rotate(rot, V3{0, 1, 0}), vp->ortho_size * ratio, vp->ortho_size, vp->near_plane, vp->far_plane);
	} else {
		perspective(out, zero, rotate(rot, V3{0, 0, -1}), rotate(rot, V3{0, 1, 0}), vp->fov, ratio, vp->near_plane, vp->far_plane);
	}
	out->origin[0] = vp->pos[0];
	out->origin[1] = vp->pos[1];
	out->origin[2] = vp->pos[2];
	return LMX_OK;
}

} // extern "C"
