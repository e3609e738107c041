// anim_kernels.hip — AnimationModuleImpl::updateAnimable (animation/animation_module.cpp:439-472) for every skinned instance:
// Model::getRelativePose (renderer/model.cpp:226-237) -> Animation::getRelativePose without a bone mask
// (animation/animation.cpp:117-204: constant tracks, bit-packed translation tracks decoded in fp64 and lerped, bit-packed
// rotation tracks with the skipped channel rebuilt from the norm and simd_nlerp, core/simd_math.h:107-123) -> time advance
// (:458-470). One block per instance, one lane per bone: a bone has at most one translation and one rotation source, so the
// reference's four sequential track loops become two table lookups per bone (tables built at lmx_anim_add). The relative pose
// is written where k_pose_palette reads it. FMA-free (-ffp-contract=off), sqrt / divide correctly rounded: bit-exact.
#include "lmx_kernels.h"

namespace lmx {

namespace {

// 8 bytes at an arbitrary byte offset of a stream (memcpy(&tmp, &stream[offset / 8], 8)); streams are padded by 16 bytes
__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t* __restrict__ stream, uint32_t byte_offset) {
	const uint64_t* w = reinterpret_cast<const uint64_t*>(stream + (byte_offset & ~7u));
	const uint32_t sh = (byte_offset & 7u) * 8u;
	const uint64_t lo = w[0];
	if (sh == 0) return lo;
	return (lo >> sh) | (w[1] << (64u - sh));
}

// simd_nlerp, core/simd_math.h:107-123: hadd-ordered dot products, t negated for the short way round, exact sqrt and divide
__device__ __forceinline__ float4 simd_nlerp(float4 q1, float4 q2, float t) {
	const float inv = 1.0f - t;
	const float d = (q1.x * q2.x + q1.y * q2.y) + (q1.z * q2.z + q1.w * q2.w);
	if (d < 0) t = -t;
	const float4 q = make_float4(q1.x * inv + q2.x * t, q1.y * inv + q2.y * t, q1.z * inv + q2.z * t, q1.w * inv + q2.w * t);
	const float l = 1 / sqrtf((q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w));
	return make_float4(q.x * l, q.y * l, q.z * l, q.w * l);
}

__device__ __forceinline__ float unpack_channel(uint64_t val, float mn, float to_range, uint32_t bitsize) { // animation.cpp:313-316
	const uint64_t mask = (1ull << bitsize) - 1ull;
	return (float)((double)mn + (double)to_range * (double)(val & mask));
}

__device__ __forceinline__ V3 anim_translation(const AnimDevice& a, const AnimTables& t, uint32_t frame, uint32_t track_idx) { // :318-334
	if ((int32_t)track_idx == a.root_translation_track) {
		const float* p = t.root_translations + 3 * (size_t)(a.root_off + frame);
		return V3{p[0], p[1], p[2]};
	}
	const LmxAnimTranslationTrack tr = t.translations[a.tt_off + track_idx];
	const uint32_t offset = a.tfs_bits * frame + tr.offset_bits;
	uint64_t tmp = load_u64_unaligned(t.translation_stream + a.tstream_off, offset / 8) >> (offset & 7u);
	V3 r;
	r.x = unpack_channel(tmp, tr.min[0], tr.to_range[0], tr.bitsizes[0]);
	tmp >>= tr.bitsizes[0];
	r.y = unpack_channel(tmp, tr.min[1], tr.to_range[1], tr.bitsizes[1]);
	tmp >>= tr.bitsizes[1];
	r.z = unpack_channel(tmp, tr.min[2], tr.to_range[2], tr.bitsizes[2]);
	return r;
}

__device__ __forceinline__ float4 unpack_rotation(const LmxAnimRotationTrack& tr, uint64_t packed) { // :51-91
	const bool is_negative = packed & 1ull;
	packed >>= 1;
	const uint64_t mask_x = (1ull << tr.bitsizes[0]) - 1ull, mask_y = (1ull << tr.bitsizes[1]) - 1ull, mask_z = (1ull << tr.bitsizes[2]) - 1ull;
	const uint64_t py = packed >> tr.bitsizes[0], pz = py >> tr.bitsizes[1];
	const float vx = tr.min[0] + tr.to_range[0] * (float)(packed & mask_x);
	const float vy = tr.min[1] + tr.to_range[1] * (float)(py & mask_y);
	const float vz = tr.min[2] + tr.to_range[2] * (float)(pz & mask_z);
	const float rest = 1 - (vx * vx + vy * vy + vz * vz);
	const float skipped = sqrtf(rest > 0.f ? rest : 0.f) * (is_negative ? -1 : 1);
	switch (tr.skipped_channel) {
		case 0: return make_float4(skipped, vx, vy, vz);
		case 1: return make_float4(vx, skipped, vy, vz);
		case 2: return make_float4(vx, vy, skipped, vz);
		default: return make_float4(vx, vy, vz, skipped);
	}
}

__device__ __forceinline__ float4 anim_rotation(const AnimDevice& a, const AnimTables& t, uint32_t frame, uint32_t track_idx, float f) { // :30-95
	if ((int32_t)track_idx == a.root_rotation_track) {
		const float4* r = t.root_rotations + (a.root_off + frame);
		return simd_nlerp(r[0], r[1], f);
	}
	const LmxAnimRotationTrack tr = t.rotations[a.rt_off + track_idx];
	const uint32_t offset1 = a.rfs_bits * frame + tr.offset_bits;
	const uint32_t offset2 = offset1 + a.rfs_bits;
	const uint8_t* stream = t.rotation_stream + a.rstream_off;
	const uint64_t packed1 = load_u64_unaligned(stream, offset1 / 8) >> (offset1 & 7u);
	const uint64_t packed2 = load_u64_unaligned(stream, offset2 / 8) >> (offset2 & 7u);
	return simd_nlerp(unpack_rotation(tr, packed1), unpack_rotation(tr, packed2), f);
}

// one bone of Animation::getRelativePose (animation.cpp:118-160, :294-311): the bone's tracks of `a` at (sample_idx, f), written over or
// blended into (p, r) with `weight`
__device__ __forceinline__ void anim_apply_bone(const AnimDevice& a, const AnimTables& t, uint32_t b, uint32_t sample_idx, float f, float weight, V3& p, float4& r) {
	const bool use_weight = weight < 0.9999f; // :304
	const float invw = 1.0f - weight;
	const int32_t ts = t.src[a.src_off + 2 * b], rs = t.src[a.src_off + 2 * b + 1]; // -1 none, 2 * i const track i, 2 * i + 1 packed track i
	if (ts >= 0) {
		V3 v;
		if (ts & 1) {
			const V3 a0 = anim_translation(a, t, sample_idx, (uint32_t)ts >> 1), a1 = anim_translation(a, t, sample_idx + 1, (uint32_t)ts >> 1);
			const float invt = 1.0f - f; // lerp(Vec3), math.cpp:194-201
			v = V3{a0.x * invt + a1.x * f, a0.y * invt + a1.y * f, a0.z * invt + a1.z * f};
		} else {
			const LmxAnimConstTranslation c = t.const_translations[a.ct_off + ((uint32_t)ts >> 1)];
			v = V3{c.value[0], c.value[1], c.value[2]};
		}
		p = use_weight ? V3{p.x * invw + v.x * weight, p.y * invw + v.y * weight, p.z * invw + v.z * weight} : v;
	}
	if (rs >= 0) {
		float4 v;
		if (rs & 1) v = anim_rotation(a, t, sample_idx, (uint32_t)rs >> 1, f);
		else {
			const LmxAnimConstRotation c = t.const_rotations[a.cr_off + ((uint32_t)rs >> 1)];
			v = make_float4(c.value[0], c.value[1], c.value[2], c.value[3]);
		}
		r = use_weight ? simd_nlerp(r, v, weight) : v;
	}
}

// float sample = clamp(time.toFrame(fps), 0.f, frame_count - 0.00001f), animation.cpp:132-134
__device__ __forceinline__ void anim_sample_point(const AnimDevice& a, uint32_t time, uint32_t& sample_idx, float& f) {
	float sample = (float)((double)time / (double)LMX_TIME_ONE_SECOND * (double)a.fps);
	const float hi = (float)a.frame_count - 0.00001f;
	sample = sample < 0.f ? 0.f : (sample > hi ? hi : sample);
	sample_idx = (uint32_t)sample;
	f = sample - (float)sample_idx;
}

__global__ __launch_bounds__(64) void k_anim_update(const SkinInstance* __restrict__ inst, uint32_t n_inst, const AnimDevice* __restrict__ anims, AnimTables t,
	const uint32_t* __restrict__ anim_of_instance, uint32_t* __restrict__ time_of_instance, float time_delta, float weight,
	const float* __restrict__ model_rel_pos, const float4* __restrict__ model_rel_rot, float* __restrict__ pose_pos, float4* __restrict__ pose_rot) {
	const uint32_t ii = blockIdx.x;
	if (ii >= n_inst) return;
	const SkinInstance in = inst[ii];
	const uint32_t anim_id = anim_of_instance[ii];
	const uint32_t time = time_of_instance[ii];
	const bool has_anim = anim_id != LMX_ANIM_NONE;
	AnimDevice a = {};
	if (has_anim) a = anims[anim_id];
	const bool sampled = has_anim && a.max_bone < in.n_bones; // m_max_accessed_bone_index >= pose.count: skeletons do not match (:120)
	uint32_t sample_idx;
	float f;
	anim_sample_point(a, time, sample_idx, f);
	for (uint32_t b = threadIdx.x; b < in.n_bones; b += 64) {
		// Model::getRelativePose, model.cpp:226-237
		V3 p = V3{model_rel_pos[3 * (size_t)(in.model_offset + b)], model_rel_pos[3 * (size_t)(in.model_offset + b) + 1], model_rel_pos[3 * (size_t)(in.model_offset + b) + 2]};
		float4 r = model_rel_rot[in.model_offset + b];
		if (sampled && b <= a.max_bone) anim_apply_bone(a, t, b, sample_idx, f, weight, p, r);
		float* gp = pose_pos + 3 * (size_t)(in.bone_offset + b);
		gp[0] = p.x; gp[1] = p.y; gp[2] = p.z;
		pose_rot[in.bone_offset + b] = r;
	}
	// (one wave per block: every lane has read time_of_instance[ii] before lane 0 replaces it - no instruction, the ordering made explicit)
	__builtin_amdgcn_wave_barrier();
	if (threadIdx.x == 0 && has_anim) { // animation_module.cpp:458-470
		const uint32_t l = a.length;
		uint32_t nt;
		if (time_delta > 0) nt = (time + (uint32_t)(time_delta * (float)LMX_TIME_ONE_SECOND)) % l;
		else {
			const uint32_t dt = (uint32_t)(-time_delta * (float)LMX_TIME_ONE_SECOND) % l;
			nt = (time + l - dt) % l;
		}
		time_of_instance[ii] = nt;
	}
}

// updateAnimator's pose work for one Animator per block (animation_module.cpp:602-636): Model::getRelativePose into the pose, then
// evalBlendStack's SAMPLE instructions in order (controller.cpp:267-293; getPose :142-157 wraps or clamps the time, then
// Animation::getRelativePose with the instruction's weight). A bone's tracks of different layers only meet in that bone, so a lane
// carries its bone through every layer in registers and the pose is written once. IK instructions are not on the path.
__global__ __launch_bounds__(64) void k_anim_blend_stack(const SkinInstance* __restrict__ inst, uint32_t n_inst, const AnimDevice* __restrict__ anims, AnimTables t,
	uint32_t n_anims, const LmxBlendSample* __restrict__ samples, const uint32_t* __restrict__ first_sample, const float* __restrict__ model_rel_pos,
	const float4* __restrict__ model_rel_rot, float* __restrict__ pose_pos, float4* __restrict__ pose_rot) {
	const uint32_t ii = blockIdx.x;
	if (ii >= n_inst) return;
	const SkinInstance in = inst[ii];
	const uint32_t s0 = first_sample[ii], s1 = first_sample[ii + 1];
	for (uint32_t b0 = 0; b0 < in.n_bones; b0 += 64) { // uniform trip count: the layer loop below reads uniform data
		const uint32_t b = b0 + threadIdx.x;
		const bool live = b < in.n_bones;
		V3 p = V3{0, 0, 0};
		float4 r = make_float4(0, 0, 0, 1);
		if (live) { // Model::getRelativePose, model.cpp:226-237
			const float* mp = model_rel_pos + 3 * (size_t)(in.model_offset + b);
			p = V3{mp[0], mp[1], mp[2]};
			r = model_rel_rot[in.model_offset + b];
		}
		for (uint32_t s = s0; s < s1; ++s) {
			const LmxBlendSample ins = samples[s];
			if (ins.animation >= n_anims) continue;
			const AnimDevice a = anims[ins.animation];
			if (a.max_bone >= in.n_bones) continue; // skeletons do not match, animation.cpp:120
			const uint32_t time = ins.looped ? ins.time % a.length : (ins.time < a.length ? ins.time : a.length); // controller.cpp:148
			uint32_t sample_idx;
			float f;
			anim_sample_point(a, time, sample_idx, f);
			if (live && b <= a.max_bone) anim_apply_bone(a, t, b, sample_idx, f, ins.weight, p, r);
		}
		if (live) {
			float* gp = pose_pos + 3 * (size_t)(in.bone_offset + b);
			gp[0] = p.x; gp[1] = p.y; gp[2] = p.z;
			pose_rot[in.bone_offset + b] = r;
		}
	}
}

} // namespace

hipError_t launch_anim_update(hipStream_t s, const SkinInstance* inst, uint32_t n_inst, const AnimDevice* anims, const AnimTables& t,
	const uint32_t* anim_of_instance, uint32_t* time_of_instance, float time_delta, float weight, const float* model_rel_pos,
	const float4* model_rel_rot, float* pose_pos, float4* pose_rot) {
	if (!n_inst) return hipSuccess;
	hipLaunchKernelGGL(k_anim_update, dim3(n_inst), dim3(64), 0, s, inst, n_inst, anims, t, anim_of_instance, time_of_instance, time_delta, weight,
		model_rel_pos, model_rel_rot, pose_pos, pose_rot);
	return hipGetLastError();
}

hipError_t launch_anim_blend_stack(hipStream_t s, const SkinInstance* inst, uint32_t n_inst, const AnimDevice* anims, const AnimTables& t, uint32_t n_anims,
	const LmxBlendSample* samples, const uint32_t* first_sample, const float* model_rel_pos, const float4* model_rel_rot, float* pose_pos, float4* pose_rot) {
	if (!n_inst) return hipSuccess;
	hipLaunchKernelGGL(k_anim_blend_stack, dim3(n_inst), dim3(64), 0, s, inst, n_inst, anims, t, n_anims, samples, first_sample, model_rel_pos, model_rel_rot,
		pose_pos, pose_rot);
	return hipGetLastError();
}

} // namespace lmx
