// pose_bridge.h — Pose / Model hand-off for the skinning path (C++ host side of include/lumix_mi355.h "skinning").
//
// In the reference the animation module writes a relative pose into ModelInstance::pose between RenderModule::lockPose and
// unlockPose (src/renderer/render_module.h:402-403; src/animation/animation_module.cpp:742-758), Pose::computeAbsolute turns it
// into model space (src/renderer/pose.cpp:63-134) and the palette / vertex transform follow (src/renderer/model.cpp:103-137).
// PoseBridge registers the skeletons and skinned meshes of the models in use once, gathers the relative poses of the skinned
// instances through lockPose each frame, runs lmx_skin_run, and stores the absolute poses back (Pose::is_absolute = true,
// unlockPose(entity, true)) for the consumers that stay on the CPU (bone attachments, ray casts).
#pragma once

#include <cstring>
#include <vector>

#include "lumix_mi355.h"

#ifdef LMX_WITH_LUMIX_HEADERS
	#include "core/math.h"
	#include "renderer/model.h"
	#include "renderer/pose.h"
	#include "renderer/render_module.h"
#else
	#include "lumix_compat.h"
#endif

namespace Lumix {

struct PoseBridge {
	explicit PoseBridge(LmxContext* ctx) : m_ctx(ctx) {}

	// Model skeleton: bones' model-space bind transforms + parents (Model::getBones / getParents / getFirstNonrootBoneIndex,
	// renderer/model.h:187-209); the inverse bind pose is derived by the library like model.cpp:404-413. Returns the model id.
	i32 addModel(const Model& model) {
		static_assert(sizeof(LocalRigidTransform) == sizeof(LmxLocalRigidTransform), "LocalRigidTransform is handed to the C ABI as is (28 bytes)");
		const auto bones = model.getBones();
		const auto parents = model.getParents();
		const u32 n = (u32)bones.length();
		m_bind.resize(n);
		m_parents.resize(n);
		for (u32 i = 0; i < n; ++i) {
			memcpy(&m_bind[i], &bones[i].transform, sizeof(LmxLocalRigidTransform));
			m_parents[i] = parents[i];
		}
		uint32_t id = 0;
		if (lmx_skin_add_model(m_ctx, n, m_parents.data(), m_bind.data(), model.getFirstNonrootBoneIndex(), &id) != LMX_OK) return -1;
		return (i32)id;
	}

	// One skinned mesh of a model: Mesh::vertices + Mesh::skin (renderer/model.h:81-131). Returns the mesh id.
	i32 addMesh(const Mesh& mesh) {
		static_assert(sizeof(Mesh::Skin) == sizeof(LmxSkin), "Mesh::Skin is handed to the C ABI as is (24 bytes)");
		uint32_t id = 0;
		if (lmx_skin_add_mesh(m_ctx, (u32)mesh.vertices.size(), reinterpret_cast<const float*>(mesh.vertices.begin()),
				reinterpret_cast<const LmxSkin*>(mesh.skin.begin()), &id) != LMX_OK)
			return -1;
		return (i32)id;
	}

	// The skinned instances of the frame, in the order their poses are gathered: entity + the ids addModel / addMesh returned.
	bool setInstances(const EntityRef* entities, const i32* models, const i32* meshes, const u32* bone_counts, u32 n) {
		m_entities.assign(entities, entities + n);
		m_bone_offset.resize(n + 1);
		m_model_ids.resize(n);
		m_mesh_ids.resize(n);
		u32 total = 0;
		for (u32 i = 0; i < n; ++i) {
			m_bone_offset[i] = total;
			total += bone_counts[i];
			m_model_ids[i] = (uint32_t)models[i];
			m_mesh_ids[i] = (uint32_t)meshes[i];
		}
		m_bone_offset[n] = total;
		m_pos.resize((size_t)total * 3);
		m_rot.resize((size_t)total * 4);
		return lmx_skin_set_instances(m_ctx, n, m_model_ids.data(), m_mesh_ids.data()) == LMX_OK;
	}

	// relative poses: lockPose -> copy -> unlockPose(entity, false)   (what AnimationModule left in ModelInstance::pose)
	bool gather(RenderModule& module) {
		for (u32 i = 0; i < (u32)m_entities.size(); ++i) {
			Pose* pose = module.lockPose(m_entities[i]);
			if (!pose) return false;
			const u32 nb = m_bone_offset[i + 1] - m_bone_offset[i];
			if (pose->count < nb) {
				module.unlockPose(m_entities[i], false);
				return false;
			}
			memcpy(&m_pos[(size_t)m_bone_offset[i] * 3], pose->positions, sizeof(float) * 3 * nb);
			memcpy(&m_rot[(size_t)m_bone_offset[i] * 4], pose->rotations, sizeof(float) * 4 * nb);
			module.unlockPose(m_entities[i], false);
		}
		return lmx_skin_upload_poses(m_ctx, m_pos.data(), m_rot.data(), m_bone_offset.back()) == LMX_OK;
	}

	// Pose::computeAbsolute + computeSkinMatrices + evaluateSkin for every instance; palettes and vertices stay in HBM
	bool run() { return lmx_skin_run(m_ctx) == LMX_OK; }

	// absolute poses back into ModelInstance::pose (Pose::is_absolute, pose.cpp:133), unlockPose(entity, true)
	bool scatter(RenderModule& module) {
		for (u32 i = 0; i < (u32)m_entities.size(); ++i) {
			Pose* pose = module.lockPose(m_entities[i]);
			if (!pose) return false;
			const u32 nb = m_bone_offset[i + 1] - m_bone_offset[i];
			const bool ok = lmx_skin_read_pose(m_ctx, i, reinterpret_cast<float*>(pose->positions), reinterpret_cast<float*>(pose->rotations), nb) == LMX_OK;
			if (ok) pose->is_absolute = true;
			module.unlockPose(m_entities[i], ok);
			if (!ok) return false;
		}
		return true;
	}

	const char* lastError() const { return lmx_last_error(m_ctx); }

private:
	LmxContext* m_ctx;
	std::vector<EntityRef> m_entities;
	std::vector<u32> m_bone_offset;
	std::vector<uint32_t> m_model_ids, m_mesh_ids;
	std::vector<LmxLocalRigidTransform> m_bind;
	std::vector<int16_t> m_parents;
	std::vector<float> m_pos, m_rot;
};

} // namespace Lumix
