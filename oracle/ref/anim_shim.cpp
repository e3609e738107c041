// anim_shim.cpp — TEST INFRASTRUCTURE. Compiles the REFERENCE'S OWN animation sampling code into oracle/_ref/liblmx_ref.so:
// AnimationSampler::getRelativePose / getRotation, Animation::getTranslation / unpackChannel (src/animation/animation.cpp:29-203,
// 294-330) and simd_nlerp (src/core/simd_math.h) on the SSE `float4` of src/core/simd.h. The code itself is NOT in this file: it is
// cut out of /root/reference at build time by oracle/ref/slice_animation.py into a temporary gen/ directory (deleted after the compile, see oracle/Makefile) and included
// below. What IS in this file, and is mine: the stand-ins for what the engine would provide around it (an Array with begin / end /
// size, Pose, Model, the Animation object's shell) and the extern "C" entry point that fills an Animation from an LmxAnimation.
// animation.cpp cannot be compiled whole: Animation is a Resource (resource manager, file system, streams).
#include <immintrin.h>
#include <math.h>
#include <string.h>

#include <vector>

#include "core/hash.h"
#include "core/math.h"
#include "lmx_types.h"

namespace Lumix {
// Everything below lives in Lumix::anim_shim: liblmx_ref.so also holds real engine object code (cull_shim.cpp / world_shim.cpp), and
// same-named inline members of the stand-ins (Array, Pose, Model) would be merged with the real ones at link time.
namespace anim_shim {

// ---- src/core/simd.h, SSE branch + src/core/simd_math.h simd_nlerp (sliced) ----
#include "gen/simd_sse.inc"
#include "gen/simd_nlerp.inc"

// ---- stand-ins (mine) ----
template <typename T> struct Array { // the part of core/array.h the sampler uses
	std::vector<T> v;
	const T* begin() const { return v.data(); }
	const T* end() const { return v.data() + v.size(); }
	T* begin() { return v.data(); }
	T* end() { return v.data() + v.size(); }
	u32 size() const { return (u32)v.size(); }
	bool empty() const { return v.empty(); }
	const T& operator[](u32 i) const { return v[i]; }
	T& operator[](u32 i) { return v[i]; }
	const T& back() const { return v.back(); }
};
struct Pose { // renderer/pose.h:15-35
	bool is_absolute = false;
	u32 count = 0;
	Vec3* positions = nullptr;
	Quat* rotations = nullptr;
};
struct Model {
	bool isReady() const { return true; }
};
struct BoneMask;
struct IAllocator;

// ---- src/animation/animation.h (sliced: Time, track records, data members) inside my shell of the class ----
#include "gen/anim_time.inc"

struct Animation {
	friend struct AnimationSampler;
	enum Flags : u32 { NONE = 0, Y_ROOT_TRANSLATION = 1 << 0, XZ_ROOT_TRANSLATION = 1 << 1, ROOT_ROTATION = 1 << 2 };
#include "gen/anim_tracks.inc"
	Animation();
	void getRelativePose(const SampleContext& ctx);
	Vec3 getTranslation(u32 frame, const TranslationTrack& track) const;
	Flags m_flags = Flags::NONE;
#include "gen/anim_members.inc"
};
Animation::RootMotion::RootMotion(IAllocator&) {}
Animation::Animation() : m_root_motion(*(IAllocator*)nullptr) {}

// ---- src/animation/animation.cpp (sliced) ----
#include "gen/anim_sampler.inc"
#include "gen/anim_methods.inc"

} // namespace anim_shim
} // namespace Lumix

using namespace Lumix;
using namespace Lumix::anim_shim;

// Animation::getRelativePose (the reference's own, sliced from animation/animation.cpp) of `a` at `time` with `weight` onto the pose that
// pos / rot already hold.
static void sample_onto(const LmxAnimation* a, uint32_t time, float weight, uint32_t n_bones, float* pos, float* rot) {
	struct Access : Animation { // the members are private in the engine's class; here the shell is mine
		void fill(const LmxAnimation* a) {
			u32 max_bone = 0;
			for (u32 i = 0; i < a->n_const_translations; ++i) {
				ConstTranslationTrack t = {};
				t.bone_index = a->const_translations[i].bone_index;
				t.value = Vec3(a->const_translations[i].value[0], a->const_translations[i].value[1], a->const_translations[i].value[2]);
				m_const_translations.v.push_back(t);
				max_bone = maximum(max_bone, (u32)t.bone_index);
			}
			for (u32 i = 0; i < a->n_translations; ++i) {
				const LmxAnimTranslationTrack& s = a->translations[i];
				TranslationTrack t = {};
				t.bone_index = s.bone_index;
				t.min = Vec3(s.min[0], s.min[1], s.min[2]);
				t.to_range = Vec3(s.to_range[0], s.to_range[1], s.to_range[2]);
				t.offset_bits = s.offset_bits;
				memcpy(t.bitsizes, s.bitsizes, 3);
				m_translations.v.push_back(t);
				max_bone = maximum(max_bone, (u32)t.bone_index);
			}
			for (u32 i = 0; i < a->n_const_rotations; ++i) {
				ConstRotationTrack t = {};
				t.bone_index = a->const_rotations[i].bone_index;
				t.value = Quat(a->const_rotations[i].value[0], a->const_rotations[i].value[1], a->const_rotations[i].value[2], a->const_rotations[i].value[3]);
				m_const_rotations.v.push_back(t);
				max_bone = maximum(max_bone, (u32)t.bone_index);
			}
			for (u32 i = 0; i < a->n_rotations; ++i) {
				const LmxAnimRotationTrack& s = a->rotations[i];
				RotationTrack t = {};
				t.bone_index = s.bone_index;
				t.min = Vec3(s.min[0], s.min[1], s.min[2]);
				t.to_range = Vec3(s.to_range[0], s.to_range[1], s.to_range[2]);
				t.offset_bits = s.offset_bits;
				memcpy(t.bitsizes, s.bitsizes, 3);
				t.skipped_channel = s.skipped_channel;
				m_rotations.v.push_back(t);
				max_bone = maximum(max_bone, (u32)t.bone_index);
			}
			m_max_accessed_bone_index = max_bone; // animation.cpp:369-393
			m_rotation_stream = a->rotation_stream;
			m_translation_stream = a->translation_stream;
			m_rotations_frame_size_bits = a->rotations_frame_size_bits;
			m_translations_frame_size_bits = a->translations_frame_size_bits;
			m_frame_count = a->frame_count;
			m_fps = a->fps;
			m_root_motion.translation_track_idx = a->root_translation_track;
			m_root_motion.rotation_track_idx = a->root_rotation_track;
			if (a->root_pose_translations)
				for (u32 f = 0; f <= a->frame_count; ++f) m_root_motion.pose_translations.v.push_back(Vec3(a->root_pose_translations[3 * f], a->root_pose_translations[3 * f + 1], a->root_pose_translations[3 * f + 2]));
			if (a->root_pose_rotations)
				for (u32 f = 0; f <= a->frame_count; ++f)
					m_root_motion.pose_rotations.v.push_back(Quat(a->root_pose_rotations[4 * f], a->root_pose_rotations[4 * f + 1], a->root_pose_rotations[4 * f + 2], a->root_pose_rotations[4 * f + 3]));
		}
	} anim;
	anim.fill(a);
	Pose pose;
	pose.count = n_bones;
	pose.positions = (Vec3*)pos;
	pose.rotations = (Quat*)rot;
	Model model;
	Animation::SampleContext ctx;
	ctx.pose = &pose;
	ctx.model = &model;
	ctx.time = Time(time);
	ctx.weight = weight;
	anim.getRelativePose(ctx);
}

// One SAMPLE instruction of evalBlendStack (animation/controller.cpp:282-289) onto an existing pose: getPose's time wrap / clamp
// (controller.cpp:148, one line, restated - getPose itself is a static function over the controller's RuntimeContext) and then the
// reference's Animation::getRelativePose with the instruction's weight and no bone mask.
extern "C" __attribute__((visibility("default"))) void ref_blend_stack_sample(const LmxAnimation* a, uint32_t time, float weight, uint32_t looped, uint32_t n_bones,
	float* pos, float* rot) {
	const Time length(a->length);
	const Time anim_time = looped ? Time(time) % length : minimum(Time(time), length);
	sample_onto(a, anim_time.raw(), weight, n_bones, pos, rot);
}

// AnimationModuleImpl::updateAnimable for one Animable (animation/animation_module.cpp:439-472): Model::getRelativePose
// (renderer/model.cpp:226-237) into the pose, the reference's Animation::getRelativePose on it, then the time advance (restated: five
// lines of integer arithmetic on Time). Same signature as orc_update_animable of the plain-C restatement.
extern "C" __attribute__((visibility("default"))) uint32_t ref_update_animable(const LmxAnimation* a, uint32_t time, float time_delta, float weight,
	const LmxLocalRigidTransform* model_relative, uint32_t n_bones, float* pos, float* rot) {
	for (uint32_t i = 0; i < n_bones; ++i) {
		memcpy(pos + 3 * i, model_relative[i].pos, 12);
		memcpy(rot + 4 * i, model_relative[i].rot, 16);
	}
	if (!a) return time;
	sample_onto(a, time, weight, n_bones, pos, rot);
	const uint32_t l = a->length; // animation_module.cpp:458-470
	if (time_delta > 0) return (Time(time) + Time::fromSeconds(time_delta)).raw() % l;
	const uint32_t dt = Time::fromSeconds(-time_delta).raw() % l;
	return (time + l - dt) % l;
}
