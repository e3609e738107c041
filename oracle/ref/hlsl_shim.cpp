// TEST INFRASTRUCTURE. The reference's dual-quaternion vertex blend, compiled from its own shader text: the SKINNED branch of
// data/shaders/surface_base.hlsli (:197-205) and transformByDualQuat (data/shaders/common.hlsli:632-636), cut out at build time by
// oracle/ref/slice_hlsl.py (gen/hlsl_blend.inc, gen/hlsl_transform.inc) and included below as C++. What stands in for HLSL is this
// file's float3 / float4 / float2x4 with exactly the operators those two fragments use; the arithmetic is plain IEEE fp32 in source
// order (-ffp-contract=off: HLSL itself leaves contraction to the shader compiler, so bit-exactness has no target - the product is
// compared within the north star's 1e-5). Links into oracle/_ref/liblmx_ref.so; only tests/ call it.
#include <cmath>
#include <cstdint>

#include "lmx_types.h"

namespace hlsl {

struct float3 { float x, y, z; };
inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator*(float a, float3 b) { return {a * b.x, a * b.y, a * b.z}; }
inline float3 operator*(float3 a, float b) { return {a.x * b, a.y * b, a.z * b}; }
inline float3 cross(float3 a, float3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct float4 {
	union {
		struct { float x, y, z, w; };
		struct { float3 xyz; float w_; }; // the swizzle the fragments read
	};
};
inline float dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline float length(float4 a) { return sqrtf(dot(a, a)); }

struct float2x4 {
	float4 r[2];
	float4& operator[](int i) { return r[i]; }
	const float4& operator[](int i) const { return r[i]; }
};
inline float2x4 mul(float2x4 m, float s) { // HLSL: matrix * scalar, component-wise
	float2x4 o;
	for (int i = 0; i < 2; ++i) { o.r[i].x = m.r[i].x * s; o.r[i].y = m.r[i].y * s; o.r[i].z = m.r[i].z * s; o.r[i].w = m.r[i].w * s; }
	return o;
}
inline float2x4& operator+=(float2x4& a, float2x4 b) {
	for (int i = 0; i < 2; ++i) { a.r[i].x += b.r[i].x; a.r[i].y += b.r[i].y; a.r[i].z += b.r[i].z; a.r[i].w += b.r[i].w; }
	return a;
}
inline float2x4& operator*=(float2x4& a, float s) {
	a = mul(a, s);
	return a;
}

struct uint4 { uint32_t x, y, z, w; };
struct VSInput { uint4 indices; float4 weights; const float* palette; }; // what the fragment reads of the shader's input
// getBones (surface_base.hlsli:122-127): two float4 loads at bone_index * 32 bytes
inline float2x4 getBones(const VSInput& input, uint32_t bone_index) {
	const float* p = input.palette + (size_t)bone_index * 8;
	float2x4 m;
	m.r[0].x = p[0]; m.r[0].y = p[1]; m.r[0].z = p[2]; m.r[0].w = p[3];
	m.r[1].x = p[4]; m.r[1].y = p[5]; m.r[1].z = p[6]; m.r[1].w = p[7];
	return m;
}

#include "gen/hlsl_transform.inc"

inline float3 skinned_position(const VSInput& input, float3 mpos) {
#include "gen/hlsl_blend.inc"
	return transformByDualQuat(dq, mpos); // surface_base.hlsli:213 before the instance's rotation / scale / translation
}

} // namespace hlsl

// out[i][v] = the shader's model-space skinned position of vertex v against instance i's dual-quaternion palette (8 floats per bone)
extern "C" __attribute__((visibility("default"))) void ref_hlsl_dq_skin(const float* verts, const LmxSkin* skin, const float* dual_quats, float* out, uint32_t n_verts,
	uint32_t n_bones, uint32_t n_inst) {
	for (uint32_t i = 0; i < n_inst; ++i)
		for (uint32_t v = 0; v < n_verts; ++v) {
			hlsl::VSInput in;
			in.indices = hlsl::uint4{(uint32_t)skin[v].indices[0], (uint32_t)skin[v].indices[1], (uint32_t)skin[v].indices[2], (uint32_t)skin[v].indices[3]};
			in.weights.x = skin[v].weights[0]; in.weights.y = skin[v].weights[1]; in.weights.z = skin[v].weights[2]; in.weights.w = skin[v].weights[3];
			in.palette = dual_quats + (size_t)i * n_bones * 8;
			const hlsl::float3 p = hlsl::skinned_position(in, hlsl::float3{verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]});
			float* o = out + ((size_t)i * n_verts + v) * 3;
			o[0] = p.x; o[1] = p.y; o[2] = p.z;
		}
}
