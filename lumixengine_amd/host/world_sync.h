// world_sync.h — World <-> MI355X hand-off of the transform hierarchy (C++ host side of include/lumix_mi355.h "world transforms").
//
// The reference's World (src/engine/world.h:49-209) is a concrete class owned by the Engine, not a pluggable interface: its
// per-entity DFS (World::transformEntity, src/engine/world.cpp:255-282) cannot be replaced from outside. What a module can do is
// mirror the hierarchy once (getTransforms / getParent / getLocalTransform), take the frame's transform writes as a batch instead
// of one setTransform call each, let the GPU propagate level by level, and hand the result back in World::getTransforms() order:
//
//     WorldSync sync(ctx);
//     sync.build(world);                                  // scene load / after structural edits (setParent, create / destroy)
//     sync.setTransform(e, tr); sync.setLocalTransform(c, l);   // instead of world.setTransform / world.setLocalTransform
//     sync.propagate();                                   // == every transformEntity DFS of the frame, bit for bit
//     sync.readTransforms(out, n);                        // Transform[n] indexed by EntityRef::index, like World::getTransforms()
//
// Entities bound with bindCulling() also refresh their culling spheres on the device (RenderModuleImpl::onModelInstanceMoved,
// src/renderer/render_module.cpp:1544-1554), so the `transformed` delegate fan-out never runs on the CPU for them.
#pragma once

#include <cstring>
#include <vector>

#include "lumix_mi355.h"

#ifdef LMX_WITH_LUMIX_HEADERS
	#include "core/math.h"
	#include "engine/world.h"
#else
	#include "lumix_compat.h"
#endif

namespace Lumix {

struct WorldSync {
	explicit WorldSync(LmxContext* ctx) : m_ctx(ctx) {}

	// Mirror `world`: parents from World::getParent, the stored world transform of every entity (World::getTransforms) and
	// Hierarchy::local_transform (World::getLocalTransform, world.cpp:756-766) of the parented ones. Both are needed: a local that
	// went through computeLocal does not reproduce the stored world transform bit for bit, and nothing is recomputed before it is
	// written (lmx_world_build_with_world). Entity slots that hold no entity become detached identity placeholders.
	bool build(const World& world) {
		static_assert(sizeof(Transform) == sizeof(LmxTransform), "Transform is handed to the C ABI as is (56 bytes)");
		i32 max_index = -1;
		for (EntityPtr e = world.getFirstEntity(); e.isValid(); e = world.getNextEntity((EntityRef)e)) max_index = e.index > max_index ? e.index : max_index;
		const u32 n = (u32)(max_index + 1);
		m_parent.assign(n, -1);
		m_stage.resize(n);
		m_stage_world.resize(n);
		for (u32 i = 0; i < n; ++i) {
			identity(m_stage[i]);
			identity(m_stage_world[i]);
		}
		const Transform* transforms = world.getTransforms();
		for (EntityPtr e = world.getFirstEntity(); e.isValid(); e = world.getNextEntity((EntityRef)e)) {
			const EntityRef r = (EntityRef)e;
			const EntityPtr p = world.getParent(r);
			m_parent[e.index] = p.isValid() ? p.index : -1;
			const Transform t = world.getLocalTransform(r); // == getTransform(r) for entities without a parent
			memcpy(&m_stage[e.index], &t, sizeof(t));
			memcpy(&m_stage_world[e.index], &transforms[e.index], sizeof(Transform));
		}
		m_n = n;
		m_set_entities.clear();
		m_set_world_entities.clear();
		return check(lmx_world_build_with_world(m_ctx, n, m_parent.data(), m_stage.data(), m_stage_world.data()));
	}

	u32 entityCount() const { return m_n; }

	// World::setTransform for entities without a parent, World::setLocalTransform for the others (world.cpp:337-342, 741-753)
	void setLocalTransform(EntityRef e, const Transform& t) { push(m_set_entities, m_set_values, e, t); }
	// World::setTransform (world-space) on any entity
	void setTransform(EntityRef e, const Transform& t) { push(m_set_world_entities, m_set_world_values, e, t); }

	// RenderModuleImpl::onModelInstanceMoved for `n` entities: culling sphere = (world position, model_radius * max scale);
	// model_radius < 0 binds the position only (decals, lights: onDecalMoved / onPointLightMoved, render_module.cpp:1568-1592)
	bool bindCulling(const EntityRef* entities, const float* model_radius, u32 n) {
		m_tmp_entities.resize(n);
		for (u32 i = 0; i < n; ++i) m_tmp_entities[i] = entities[i].index;
		return check(lmx_world_bind_culling(m_ctx, n, m_tmp_entities.data(), model_radius));
	}

	// Everything staged since the last call reaches the device, then one level-by-level pass (+ the sphere refresh of bound entities).
	bool propagate() {
		bool ok = true;
		if (!m_set_entities.empty()) ok = check(lmx_world_set_transforms(m_ctx, (u32)m_set_entities.size(), m_set_entities.data(), m_set_values.data())) && ok;
		if (!m_set_world_entities.empty())
			ok = check(lmx_world_set_world_transforms(m_ctx, (u32)m_set_world_entities.size(), m_set_world_entities.data(), m_set_world_values.data())) && ok;
		m_set_entities.clear();
		m_set_values.clear();
		m_set_world_entities.clear();
		m_set_world_values.clear();
		return check(lmx_world_propagate(m_ctx)) && ok;
	}

	// World::getTransforms(): Transform[n] by EntityRef::index. `out` may be the engine's own array
	// (const_cast<Transform*>(world.getTransforms())) when the module is the only writer of transforms.
	bool readTransforms(Transform* out, u32 n) { return check(lmx_world_read_transforms(m_ctx, reinterpret_cast<LmxTransform*>(out), n)); }
	// World::getLocalTransform for every entity (children written through setLocalTransform / setTransform carry the re-derived local)
	bool readLocalTransforms(Transform* out, u32 n) { return check(lmx_world_read_local_transforms(m_ctx, reinterpret_cast<LmxTransform*>(out), n)); }

	const char* lastError() const { return lmx_last_error(m_ctx); }

private:
	static void identity(LmxTransform& t) {
		memset(&t, 0, sizeof(t));
		t.rot[3] = 1.f;
		t.scale[0] = t.scale[1] = t.scale[2] = 1.f;
	}
	static void push(std::vector<int32_t>& entities, std::vector<LmxTransform>& values, EntityRef e, const Transform& t) {
		LmxTransform v;
		memset(&v, 0, sizeof(v));
		memcpy(&v, &t, sizeof(t));
		entities.push_back(e.index);
		values.push_back(v);
	}
	bool check(int rc) const { return rc == LMX_OK; }

	LmxContext* m_ctx;
	u32 m_n = 0;
	std::vector<int32_t> m_parent, m_set_entities, m_set_world_entities, m_tmp_entities;
	std::vector<LmxTransform> m_stage, m_stage_world, m_set_values, m_set_world_values;
};

} // namespace Lumix
