// pose_shim.cpp — TEST INFRASTRUCTURE. Compiles the REFERENCE'S OWN pose / palette code into oracle/_ref/liblmx_ref.so:
//   Pose::blend, Pose::computeAbsolute (the 4-wide SOA path for aligned groups of four bones whose parents precede the group + the scalar
//   tail), Pose::computeRelative                                                  src/renderer/pose.cpp:30-41, 62-133, 136-147
//   PipelineImpl::computeSkeletonDualQuats (4-wide batches through toDualQuat(SIMDLocalRigidTransform) + scalar tail)
//                                                                                   src/renderer/pipeline.cpp:2680-2745
//   static invert / evaluateSkin / computeSkinMatrices                             src/renderer/model.cpp:24-30, 103-109, 132-137
// on the SSE `float4` of src/core/simd.h and the SOA helpers of src/core/simd_math.h. The code itself is NOT in this file: it is cut out
// of /root/reference at build time by oracle/ref/slice_pose.py into a temporary gen/ directory (deleted after the compile, see
// oracle/Makefile) and included below. What IS in this file, and is mine: the shells of Pose / Model / ModelInstance (the data members
// that code reads) and the extern "C" entry points. pose.cpp / model.cpp / pipeline.cpp cannot be compiled whole (resource system,
// renderer), and core/simd.h's non-MSVC float4 lacks the helpers pose.cpp uses.
#include <immintrin.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "core/core.h"
#include "core/math.h"
#include "lmx_types.h"
#include "worker_pool.h"

namespace Lumix {
// own namespace: liblmx_ref.so also holds real engine object code; same-named inline members of the shells must not be merged with it
namespace pose_shim {

// ---- src/core/simd.h SSE branch, src/core/simd_math.h SOA helpers (sliced) ----
#include "gen/simd_sse.inc"
#include "gen/simd_soa.inc"
// ---- src/renderer/model.h (sliced): SOATransform, Mesh::Skin ----
#include "gen/model_soa.inc"

// ---- shells (mine) ----
struct Model {
	struct Bone {};
	int getFirstNonrootBoneIndex() const { return first_nonroot; }
	i32 getBoneParent(u32 i) const { return parents[i]; }
	const SOATransform& getInverseBindPose() { return inv_bind_soa; }
	LocalRigidTransform getInverseBindTransform(i32 i) const { return inv_bind[i]; }
	const Bone& getBone(int) const { return bone; }
	int first_nonroot = 0;
	const i16* parents = nullptr;
	const LocalRigidTransform* inv_bind = nullptr;
	SOATransform inv_bind_soa;
	Bone bone;
};
struct TransientSlice { void* ptr = nullptr; };
struct Pose { // renderer/pose.h:15-35
	void blend(Pose& rhs, float weight);
	void computeAbsolute(Model& model);
	void computeRelative(Model& model);
	bool is_absolute = false;
	u32 count = 0;
	Vec3* positions = nullptr;
	Quat* rotations = nullptr;
	TransientSlice slice;
};
struct ModelInstance {
	Model* model = nullptr;
	Pose* pose = nullptr;
};

// ---- src/renderer/pose.cpp, model.cpp (sliced) ----
#include "gen/pose_methods.inc"
#include "gen/model_statics.inc"

struct PipelineShell {
	// ---- src/renderer/pipeline.cpp (sliced) ----
#include "gen/pose_dual_quats.inc"
};

// 16-byte aligned working copies: the reference loads rotations with f4Load and reads one float past the last position
struct PoseBuffers {
	Vec3* pos;
	Quat* rot;
	PoseBuffers(u32 count) {
		if (posix_memalign((void**)&pos, 16, sizeof(Vec3) * (count + 1) + 16) != 0) abort(); // Pose::resize: + 1 padding
		if (posix_memalign((void**)&rot, 16, sizeof(Quat) * (count ? count : 1)) != 0) abort();
		memset((void*)pos, 0, sizeof(Vec3) * (count + 1));
	}
	~PoseBuffers() { free(pos); free(rot); }
};

} // namespace pose_shim
} // namespace Lumix

using namespace Lumix;
using namespace Lumix::pose_shim;

extern "C" {
#define REF_API __attribute__((visibility("default")))

REF_API void ref_pose_blend(float* positions, float* rotations, const float* rhs_positions, const float* rhs_rotations, uint32_t count, float weight) {
	Pose a, b;
	a.count = b.count = count;
	a.positions = (Vec3*)positions;
	a.rotations = (Quat*)rotations;
	b.positions = (Vec3*)rhs_positions;
	b.rotations = (Quat*)rhs_rotations;
	a.blend(b, weight);
}

REF_API void ref_pose_compute_absolute(float* positions, float* rotations, const int16_t* parents, int32_t first_nonroot, uint32_t count,
	uint32_t n_instances, int n_threads) {
	lmx_ref::forEachJob(n_instances, n_threads, [&](unsigned inst) {
		PoseBuffers buf(count);
		float* p = positions + (size_t)inst * count * 3;
		float* r = rotations + (size_t)inst * count * 4;
		memcpy((void*)buf.pos, p, sizeof(Vec3) * count);
		memcpy((void*)buf.rot, r, sizeof(Quat) * count);
		Model model;
		model.first_nonroot = first_nonroot;
		model.parents = parents;
		Pose pose;
		pose.count = count;
		pose.positions = buf.pos;
		pose.rotations = buf.rot;
		pose.computeAbsolute(model);
		memcpy(p, (void*)buf.pos, sizeof(Vec3) * count);
		memcpy(r, (void*)buf.rot, sizeof(Quat) * count);
	});
}

REF_API void ref_pose_compute_relative(float* positions, float* rotations, const int16_t* parents, int32_t first_nonroot, uint32_t count, uint32_t n_instances) {
	for (uint32_t inst = 0; inst < n_instances; ++inst) {
		Model model;
		model.first_nonroot = first_nonroot;
		model.parents = parents;
		Pose pose;
		pose.count = count;
		pose.is_absolute = true;
		pose.positions = (Vec3*)(positions + (size_t)inst * count * 3);
		pose.rotations = (Quat*)(rotations + (size_t)inst * count * 4);
		pose.computeRelative(model);
	}
}

REF_API void ref_invert_bind(const LmxLocalRigidTransform* bind, LmxLocalRigidTransform* out, uint32_t n) {
	static_assert(sizeof(LocalRigidTransform) == sizeof(LmxLocalRigidTransform), "LocalRigidTransform layout");
	for (uint32_t i = 0; i < n; ++i) {
		LocalRigidTransform tr;
		memcpy((void*)&tr, &bind[i], sizeof(tr));
		const LocalRigidTransform result = invert(tr);
		memcpy(&out[i], &result, sizeof(result));
	}
}

REF_API void ref_skin_matrices(const float* pose_pos, const float* pose_rot, const LmxLocalRigidTransform* inv_bind, LmxMatrix* out, uint32_t count,
	uint32_t n_instances, int n_threads) {
	static_assert(sizeof(Matrix) == sizeof(LmxMatrix), "Matrix layout");
	lmx_ref::forEachJob(n_instances, n_threads, [&](unsigned inst) {
		Model model;
		model.inv_bind = (const LocalRigidTransform*)inv_bind;
		Pose pose;
		pose.count = count;
		pose.positions = (Vec3*)(pose_pos + (size_t)inst * count * 3);
		pose.rotations = (Quat*)(pose_rot + (size_t)inst * count * 4);
		computeSkinMatrices(pose, model, (Matrix*)(out + (size_t)inst * count));
	});
}

REF_API void ref_dual_quats(const float* pose_pos, const float* pose_rot, const LmxLocalRigidTransform* inv_bind, float* out, uint32_t count,
	uint32_t n_instances) {
	static_assert(sizeof(DualQuat) == 32, "DualQuat layout");
	// Model::m_inverse_bind: the inverse bind pose as seven SOA arrays (model.cpp fills them next to the per-bone transforms), 16-byte aligned
	const u32 padded = (count + 3) & ~3u;
	float* soa = nullptr;
	if (posix_memalign((void**)&soa, 16, sizeof(float) * 7 * (padded ? padded : 4)) != 0) abort();
	memset(soa, 0, sizeof(float) * 7 * (padded ? padded : 4));
	for (u32 i = 0; i < count; ++i) {
		for (int k = 0; k < 3; ++k) soa[k * padded + i] = inv_bind[i].pos[k];
		for (int k = 0; k < 4; ++k) soa[(3 + k) * padded + i] = inv_bind[i].rot[k];
	}
	Model model;
	model.inv_bind = (const LocalRigidTransform*)inv_bind;
	model.inv_bind_soa = {soa, soa + padded, soa + 2 * padded, soa + 3 * padded, soa + 4 * padded, soa + 5 * padded, soa + 6 * padded};
	PoseBuffers buf(count);
	DualQuat* dq = nullptr;
	if (posix_memalign((void**)&dq, 16, sizeof(DualQuat) * (count ? count : 1)) != 0) abort();
	PipelineShell pipeline;
	for (uint32_t inst = 0; inst < n_instances; ++inst) {
		memcpy((void*)buf.pos, pose_pos + (size_t)inst * count * 3, sizeof(Vec3) * count);
		memcpy((void*)buf.rot, pose_rot + (size_t)inst * count * 4, sizeof(Quat) * count);
		Pose pose;
		pose.count = count;
		pose.positions = buf.pos;
		pose.rotations = buf.rot;
		pose.slice.ptr = dq;
		ModelInstance mi;
		mi.model = &model;
		mi.pose = &pose;
		pipeline.computeSkeletonDualQuats(&mi);
		_mm_sfence(); // f4Stream = non-temporal stores
		memcpy(out + (size_t)inst * count * 8, (void*)dq, sizeof(DualQuat) * count);
	}
	free(dq);
	free(soa);
}

REF_API void ref_evaluate_skin(const float* verts, const LmxSkin* skin, const LmxMatrix* palettes, float* out, uint32_t n_verts, uint32_t n_bones,
	uint32_t n_instances, int n_threads) {
	static_assert(sizeof(Mesh::Skin) == sizeof(LmxSkin), "Mesh::Skin layout");
	lmx_ref::forEachJob(n_instances, n_threads, [&](unsigned inst) {
		const Matrix* matrices = (const Matrix*)(palettes + (size_t)inst * n_bones);
		float* o = out + (size_t)inst * n_verts * 3;
		for (u32 v = 0; v < n_verts; ++v) {
			Mesh::Skin s;
			memcpy((void*)&s, &skin[v], sizeof(s));
			Vec3 p(verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]);
			const Vec3 r = evaluateSkin(p, s, matrices);
			o[3 * v] = r.x;
			o[3 * v + 1] = r.y;
			o[3 * v + 2] = r.z;
		}
	});
}
} // extern "C"
