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
//     sync.readMoved(entities, transforms);               // what moved (what transformEntity would have visited) -> World::getTransforms()
//
// Entities bound with bindCulling() also refresh their culling spheres on the device (RenderModuleImpl::onModelInstanceMoved,
// src/renderer/render_module.cpp:1544-1554), so the `transformed` delegate fan-out never runs on the CPU for them.
#pragma once

#include <cstring>
#include <string>
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
	// `ctx` is the World's shared context (lmx_ctx_acquire_shared): the same one the World's GpuCullingSystem works on, so that
	// bindCulling() finds the entities RenderModuleImpl added to it. Calls serialise on the context's lock.
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
		// writes staged before a structural change (an entity created or destroyed mid-frame re-mirrors the World) are not in the World
		// yet: they stay staged for the entities that still exist
		std::vector<Staged> keep;
		keep.swap(m_staged);
		m_staged_at.assign(n, -1);
		for (const Staged& st : keep) {
			if ((u32)st.entity >= n || !world.hasEntity(EntityRef{st.entity})) continue;
			m_staged_at[st.entity] = (int32_t)m_staged.size();
			m_staged.push_back(st);
		}
		Lock guard(m_ctx);
		return check(lmx_world_track_moved(m_ctx, 1)) && check(lmx_world_build_with_world(m_ctx, n, m_parent.data(), m_stage.data(), m_stage_world.data()));
	}

	u32 entityCount() const { return m_n; }

	// World::setTransform for entities without a parent, World::setLocalTransform for the others (world.cpp:337-342, 741-753)
	void setLocalTransform(EntityRef e, const Transform& t) { push(e, t, false); }
	// World::setTransform (world-space) on any entity
	void setTransform(EntityRef e, const Transform& t) { push(e, t, true); }

	// RenderModuleImpl::onModelInstanceMoved for `n` entities: culling sphere = (world position, model_radius * max scale);
	// model_radius < 0 binds the position only (decals, lights: onDecalMoved / onPointLightMoved, render_module.cpp:1568-1592).
	// Fails (lastError() says why) when an entity is not in THIS context's culling system - e.g. a culling system that was
	// created on a context of its own instead of the World's shared one.
	bool bindCulling(const EntityRef* entities, const float* model_radius, u32 n) {
		m_tmp_entities.resize(n);
		for (u32 i = 0; i < n; ++i) m_tmp_entities[i] = entities[i].index;
		Lock guard(m_ctx);
		return check(lmx_world_bind_culling(m_ctx, n, m_tmp_entities.data(), model_radius));
	}

	// Everything staged since the last call reaches the device, then one level-by-level pass (+ the sphere refresh of bound
	// entities). The reference applies writes one by one, in call order: for two writes to ONE entity in a frame the last one wins
	// (push() keeps only it, whichever of the two entry points each used).
	bool propagate() {
		m_set_entities.clear();
		m_set_values.clear();
		m_set_world_entities.clear();
		m_set_world_values.clear();
		for (const Staged& st : m_staged) {
			(st.world_space ? m_set_world_entities : m_set_entities).push_back(st.entity);
			(st.world_space ? m_set_world_values : m_set_values).push_back(st.value);
			m_staged_at[st.entity] = -1;
		}
		m_staged.clear();
		Lock guard(m_ctx);
		bool ok = true;
		if (!m_set_entities.empty()) ok = check(lmx_world_set_transforms(m_ctx, (u32)m_set_entities.size(), m_set_entities.data(), m_set_values.data())) && ok;
		if (!m_set_world_entities.empty())
			ok = check(lmx_world_set_world_transforms(m_ctx, (u32)m_set_world_entities.size(), m_set_world_entities.data(), m_set_world_values.data())) && ok;
		return check(lmx_world_propagate(m_ctx)) && ok;
	}

	// The frame's hand-back: the entities World::transformEntity would have visited since the last call (every propagate() in
	// between) and their new world transforms - costs what moved, not the size of the world. An entity may be listed twice
	// (moved by two propagations): apply in order, later entries are newer.
	bool readMoved(std::vector<int32_t>& entities, std::vector<Transform>& transforms) {
		Lock guard(m_ctx);
		uint32_t n = 0;
		entities.resize(m_moved_cap);
		transforms.resize(m_moved_cap);
		int rc = lmx_world_read_moved(m_ctx, entities.data(), reinterpret_cast<LmxTransform*>(transforms.data()), m_moved_cap, &n);
		if (rc == LMX_ERR_CAPACITY && n > m_moved_cap && n <= 2 * m_n) { // grow to what this frame needs and read again
			m_moved_cap = n + n / 4 + 64;
			entities.resize(m_moved_cap);
			transforms.resize(m_moved_cap);
			rc = lmx_world_read_moved(m_ctx, entities.data(), reinterpret_cast<LmxTransform*>(transforms.data()), m_moved_cap, &n);
		}
		if (!check(rc)) {
			entities.clear();
			transforms.clear();
			return false;
		}
		entities.resize(n);
		transforms.resize(n);
		return true;
	}

	// World::getTransforms(): Transform[n] by EntityRef::index - every entity, whatever moved (scene load, tools)
	bool readTransforms(Transform* out, u32 n) {
		Lock guard(m_ctx);
		return check(lmx_world_read_transforms(m_ctx, reinterpret_cast<LmxTransform*>(out), n));
	}
	// World::getLocalTransform for every entity (children written through setLocalTransform / setTransform carry the re-derived local)
	bool readLocalTransforms(Transform* out, u32 n) {
		Lock guard(m_ctx);
		return check(lmx_world_read_local_transforms(m_ctx, reinterpret_cast<LmxTransform*>(out), n));
	}

	const char* lastError() const { return m_error.c_str(); }

private:
	struct Lock {
		explicit Lock(LmxContext* c) : ctx(c) { lmx_ctx_lock(ctx); }
		~Lock() { lmx_ctx_unlock(ctx); }
		Lock(const Lock&) = delete;
		Lock& operator=(const Lock&) = delete;
		LmxContext* ctx;
	};
	struct Staged { int32_t entity; bool world_space; LmxTransform value; };

	static void identity(LmxTransform& t) {
		memset(&t, 0, sizeof(t));
		t.rot[3] = 1.f;
		t.scale[0] = t.scale[1] = t.scale[2] = 1.f;
	}
	void push(EntityRef e, const Transform& t, bool world_space) {
		if (e.index < 0 || (u32)e.index >= m_n) return; // created after build(): the owner rebuilds the mirror (entityCreated)
		LmxTransform v;
		memset(&v, 0, sizeof(v));
		memcpy(&v, &t, sizeof(t));
		int32_t& at = m_staged_at[e.index];
		if (at < 0) {
			at = (int32_t)m_staged.size();
			m_staged.push_back(Staged{e.index, world_space, v});
		} else { // the entity's earlier write of this frame is superseded
			m_staged[at].world_space = world_space;
			m_staged[at].value = v;
		}
	}
	bool check(int rc) {
		if (rc == LMX_OK) return true;
		m_error = lmx_last_error(m_ctx);
		return false;
	}

	LmxContext* m_ctx;
	u32 m_n = 0;
	u32 m_moved_cap = 1024;
	std::string m_error;
	std::vector<Staged> m_staged;
	std::vector<int32_t> m_staged_at; // entity -> index into m_staged, -1 = nothing staged
	std::vector<int32_t> m_parent, m_set_entities, m_set_world_entities, m_tmp_entities;
	std::vector<LmxTransform> m_stage, m_stage_world, m_set_values, m_set_world_values;
};

} // namespace Lumix
