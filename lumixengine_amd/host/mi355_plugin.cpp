// mi355_plugin.cpp — the engine-side module of the MI355X hot path: an ISystem / IModule pair registered through
// LUMIX_PLUGIN_ENTRY (src/engine/plugin.h:37-96; the renderer's own instance: src/renderer/renderer.cpp:1410-1413).
//
// Compiled inside a LumixEngine tree (-DLMX_WITH_LUMIX_HEADERS, include paths of the engine's src/ and this repository's
// include/ + lumixengine_amd/host/). The reference as a whole does not compile on Linux at this snapshot (src/core/sync.h:20-24 is
// `#error "Not implemented"`), so this repository checks the file two ways against a copy of the reference's headers in which that
// one line is patched: tests/test_plugin_compile.py (syntax + object code + exported entry points) and, on the GPU,
// oracle/_ref/real_header_harness (tests/cpp/real_header_harness.cpp: this object code linked with the reference's engine/world.cpp,
// driven through IModule::update on a real World and compared with a CPU-only World frame by frame).
//
// What the module owns and does per frame (Engine::update, src/engine/engine.cpp:289-341):
//   createModules   world.addModule(Mi355Module)                                    (world.cpp:218-235)
//   init            mirror the World's hierarchy (WorldSync::build), find the "renderer" module, register skeletons / meshes
//   update          propagate the frame's staged transform writes level by level on the GPU (+ culling sphere refresh),
//                   gather relative poses through lockPose / unlockPose, run pose -> palette -> skin, move bone-attached entities,
//                   store absolute poses back
//   createGpuCullingSystem   what RenderModuleImpl's constructor calls instead of CullingSystem::create (render_module.cpp:3569)
//
// ONE LmxContext per World. The reference creates the culling system inside RenderModuleImpl (render_module.cpp:3569) and moves its
// spheres from RenderModuleImpl::onModelInstanceMoved (:1544-1554) on the same object. Here the CullingSystem replacement and this
// module are created independently of each other (one by the renderer, one by the plugin manager), so both take the World's context
// from the library's registry under the same key - the World's address (lmx_ctx_acquire_shared): bindCulling() then finds the
// entities the renderer added, and the device-side transform pass refreshes the spheres of the set the renderer culls.
#include "core/allocator.h"
#include "core/log.h"
#include "core/page_allocator.h"
#include "engine/engine.h"
#include "engine/plugin.h"
#include "engine/reflection.h"
#include "engine/world.h"
#include "renderer/culling_system.h"
#include "renderer/model.h"
#include "renderer/pose.h"
#include "renderer/render_module.h"

#include "gpu_culling_system.h"
#include "pose_bridge.h"
#include "world_sync.h"

namespace Lumix {

// Drop-in for `m_culling_system = CullingSystem::create(m_allocator, engine.getPageAllocator())` (render_module.cpp:3569): same
// ownership (UniquePtr destroyed through the same IAllocator), plus the World the RenderModule belongs to (RenderModuleImpl has it
// as m_world) - the key under which the plugin's module finds the same context.
// async_compaction (default on): the re-sort of the sorted set runs on a worker thread against a SECOND host + device copy of the culling
// sets (~40 bytes per entity on the device: 400 MB of HBM at 10 M entities, plus an operation log of up to 40 MB) and is traded with the
// live set inside a later cull - no frame stalls, as the reference's add / remove / set* never do (culling_system.cpp:131-258). Pass false
// where memory matters more than the worst frame (tools, tests, worlds that only change at level boundaries: the re-sort then happens
// inside the flush that crosses the threshold, 0.5 s at 10 M entities, or when the host calls lmx_cull_compact).
UniquePtr<CullingSystem> createGpuCullingSystem(IAllocator& allocator, PageAllocator& page_allocator, World& world, bool async_compaction) {
	UniquePtr<GpuCullingSystem> cs = UniquePtr<GpuCullingSystem>::create(allocator, page_allocator, static_cast<const void*>(&world));
	if (async_compaction && cs.get() && cs->isValid()) cs->setAsyncCompaction(true);
	return cs;
}
UniquePtr<CullingSystem> createGpuCullingSystem(IAllocator& allocator, PageAllocator& page_allocator, World& world) {
	return createGpuCullingSystem(allocator, page_allocator, world, true);
}

struct Mi355Module final : IModule {
	Mi355Module(ISystem& system, Engine& engine, World& world)
		: m_system(system)
		, m_engine(engine)
		, m_world(world) {
		// the World's shared context: the renderer's GpuCullingSystem (createGpuCullingSystem) holds the same one
		if (lmx_ctx_acquire_shared(&world, 0, &m_ctx) != LMX_OK) {
			logError("mi355: ", lmx_last_error(nullptr)); // the module then idles
			m_ctx = nullptr;
		}
		// structural changes invalidate the mirror (entity slots, hierarchy); setParent has no delegate: callers use markDirty()
		m_world.entityCreated().bind<&Mi355Module::onEntityChanged>(this);
		m_world.entityDestroyed().bind<&Mi355Module::onEntityChanged>(this);
	}
	~Mi355Module() override {
		m_world.entityCreated().unbind<&Mi355Module::onEntityChanged>(this);
		m_world.entityDestroyed().unbind<&Mi355Module::onEntityChanged>(this);
		unbindTransformed();
		m_sync.reset();
		m_poses.reset();
		lmx_ctx_release_shared(m_ctx);
	}

	const char* getName() const override { return "mi355_hot_path"; }
	ISystem& getSystem() const override { return m_system; }
	World& getWorld() override { return m_world; }
	// GPU state is derived data: it is rebuilt from the World and the renderer's components, never serialized
	void serialize(OutputMemoryStream&) override {}
	void deserialize(InputMemoryStream&, const EntityMap&, i32) override { m_dirty = true; }
	i32 getVersion() const override { return 0; }

	void init() override {
		m_render_module = static_cast<RenderModule*>(m_world.getModule("renderer"));
		m_dirty = true;
	}
	void startGame() override { m_dirty = true; }

	// staged writes: game code (or a script binding) calls these instead of World::setTransform / setLocalTransform
	void setTransform(EntityRef e, const Transform& t) { if (m_sync) m_sync->setTransform(e, t); }
	void setLocalTransform(EntityRef e, const Transform& t) { if (m_sync) m_sync->setLocalTransform(e, t); }
	// after World::setParent (the World has no delegate for it), or whenever the mirror must be re-read
	void markDirty() { m_dirty = true; }
	LmxContext* context() { return m_ctx; }
	// RenderModuleImpl::onModelInstanceMoved on the device for these entities (culling sphere = world position, model_radius *
	// max scale, render_module.cpp:1553-1554): rebuild() calls it with every valid model instance of the RenderModule; hosts without
	// a RenderModule (tools, the real-header test harness) call it themselves. Fails - loudly - when the entities are not in the
	// culling set of THIS World's context, i.e. when the renderer's culling system was not created by createGpuCullingSystem(…, world).
	bool bindModelInstances(const EntityRef* entities, const float* model_radius, u32 n) {
		if (!m_sync) return false;
		if (n == 0 || m_sync->bindCulling(entities, model_radius, n)) return true;
		fail("binding model instances to the culling system (is RenderModuleImpl's culling system createGpuCullingSystem(allocator, pages, world)?)", m_sync->lastError());
		return false;
	}
	// false after a rebuild whose culling binding / instance registration failed (logged; lastError() says why)
	bool isBound() const { return m_bound_ok; }
	const char* lastError() const { return m_error.c_str(); }

	void update(float) override {
		if (!m_ctx) return;
		if (m_dirty) rebuild();
		if (!m_sync) return;
		// 1. every transformEntity DFS of the frame + onModelInstanceMoved for bound entities, on the device
		if (!m_sync->propagate()) fail("propagate", m_sync->lastError());
		// 2. relative poses -> absolute poses, palettes, skinned vertices
		if (m_render_module && m_poses && !m_skinned.empty()) {
			if (m_poses->gather(*m_render_module) && m_poses->run()) {
				// 3. bone attachments follow the fresh absolute poses on the device (RenderModuleImpl::updateBoneAttachment,
				//    render_module.cpp:377-404, for every attachment of a moved pose, :1964-1981), then their subtrees and culling
				//    spheres in a second (small) propagation - same frame, as the reference's eager setTransform
				if (m_n_attachments) {
					lmx_ctx_lock(m_ctx);
					const bool ok = lmx_world_update_bone_attachments(m_ctx) == LMX_OK;
					lmx_ctx_unlock(m_ctx);
					if (!ok) fail("bone attachments", lmx_last_error(m_ctx));
					else if (!m_sync->propagate()) fail("propagating bone-attached subtrees", m_sync->lastError());
				}
				m_poses->scatter(*m_render_module);
			}
		}
		// 4. hand-back: the entities that moved (and only those) into World::getTransforms(), then the `transformed` delegates
		//    World::transformEntity fires for each of them (world.cpp:257-260) - lights, physics, audio, the renderer's MOVED flags.
		handBack();
		if (++m_frame_stamp == 0) { // (wrapped: forget the stamps)
			m_engine_write_stamp.assign(m_engine_write_stamp.size(), 0);
			m_frame_stamp = 1;
		}
	}

private:
	void onEntityChanged(EntityRef) { m_dirty = true; }

	// A World::setTransform* by other engine code (editor gizmo, physics, scripts): the World has already run its own DFS and fired the
	// delegates. The device mirror takes the WRITTEN entity's new world transform as a staged world-space write; the descendants the
	// World's DFS announces right after it (it fires `transformed` top-down, world.cpp:255-282) are skipped - the mirror re-derives
	// them from its own stored locals in the next propagation, and the hand-back then overwrites what the World's DFS computed from
	// ITS locals, which are stale for children whose local was written through this module (Hierarchy::local_transform is private
	// to World: INTEGRATION.md, "what the module cannot reach"). (Entities without any component are not announced by the World;
	// they matter only as parents, and their children are announced.)
	void onTransformedByEngine(EntityRef e) {
		if (m_in_hand_back || !m_sync || (u32)e.index >= m_sync->entityCount()) return;
		if (m_engine_write_stamp.size() < m_sync->entityCount()) m_engine_write_stamp.assign(m_sync->entityCount(), 0);
		for (EntityPtr a = m_world.getParent(e); a.isValid(); a = m_world.getParent((EntityRef)a)) {
			if ((u32)a.index < m_engine_write_stamp.size() && m_engine_write_stamp[a.index] == m_frame_stamp) return; // inside an announced subtree
		}
		m_engine_write_stamp[e.index] = m_frame_stamp;
		m_sync->setTransform(e, m_world.getTransform(e));
	}

	void bindTransformed() {
		unbindTransformed();
		for (const reflection::RegisteredComponent& rc : reflection::getComponents()) {
			if (!rc.cmp) continue;
			m_world.componentTransformed(rc.cmp->component_type).bind<&Mi355Module::onTransformedByEngine>(this);
			m_transformed_types.push_back(rc.cmp->component_type);
		}
	}
	void unbindTransformed() {
		for (ComponentType t : m_transformed_types) m_world.componentTransformed(t).unbind<&Mi355Module::onTransformedByEngine>(this);
		m_transformed_types.clear();
	}

	void handBack() {
		if (!m_sync->readMoved(m_moved_entities, m_moved_transforms)) {
			fail("read moved", m_sync->lastError());
			return;
		}
		if (m_moved_entities.empty()) return;
		Transform* transforms = const_cast<Transform*>(m_world.getTransforms()); // the module is the writer of staged transforms
		for (size_t i = 0; i < m_moved_entities.size(); ++i) transforms[m_moved_entities[i]] = m_moved_transforms[i];
		// RenderModuleImpl::onModelInstanceMoved would now call CullingSystem::set per entity: the device has refreshed those spheres
		// already (k_sphere_refresh); the culling system drops the repeats while the delegates run, everything else they do happens
		lmx_ctx_lock(m_ctx);
		lmx_cull_set_option(m_ctx, LMX_CULL_OPT_DEVICE_OWNS_BOUND, 1);
		m_in_hand_back = true;
		for (size_t i = 0; i < m_moved_entities.size(); ++i) {
			const EntityRef e{m_moved_entities[i]};
			for (ComponentType type : m_world.getComponents(e)) m_world.componentTransformed(type).invoke(e);
		}
		m_in_hand_back = false;
		lmx_cull_set_option(m_ctx, LMX_CULL_OPT_DEVICE_OWNS_BOUND, 0);
		lmx_ctx_unlock(m_ctx);
	}

	void fail(const char* what, const char* why) {
		m_error = why ? why : "";
		if (m_error != m_last_logged) { // once per distinct failure, not once per frame
			logError("mi355: ", what, ": ", m_error.c_str());
			m_last_logged = m_error;
		}
	}

	// Mirror World + RenderModule state: hierarchy, culling bindings of model instances, skeletons / meshes / skinned instances
	void rebuild() {
		m_dirty = false;
		m_bound_ok = false;
		if (!m_sync) m_sync = UniquePtr<WorldSync>::create(m_engine.getAllocator(), m_ctx);
		if (!m_sync->build(m_world)) {
			fail("mirroring the World", m_sync->lastError());
			return;
		}
		bindTransformed();
		if (!m_render_module) {
			m_bound_ok = true; // nothing to bind
			return;
		}
		Span<ModelInstance> instances = m_render_module->getModelInstances();
		m_bound.clear();
		m_bound_radius.clear();
		m_skinned.clear();
		m_skinned_models.clear();
		m_skinned_meshes.clear();
		m_skinned_bones.clear();
		if (!m_poses) m_poses = UniquePtr<PoseBridge>::create(m_engine.getAllocator(), m_ctx);
		Model* last_model = nullptr;
		i32 last_model_id = -1, last_mesh_id = -1;
		for (u32 i = 0; i < instances.length(); ++i) {
			ModelInstance& mi = instances[i];
			if (!(mi.flags & ModelInstance::VALID) || !mi.model || !mi.model->isReady()) continue;
			const EntityRef e{(i32)i};
			// onModelInstanceMoved: radius = model->getOriginBoundingRadius() * maximum(scale) (render_module.cpp:1553-1554)
			m_bound.push_back(e);
			m_bound_radius.push_back(mi.model->getOriginBoundingRadius());
			if (!mi.pose || mi.pose->count == 0) continue;
			if (mi.model != last_model) { // instances of one model are usually consecutive: register each skeleton / mesh once per run
				last_model = mi.model;
				last_model_id = m_poses->addModel(*mi.model);
				last_mesh_id = -1;
				for (int m = 0; m < mi.model->getMeshCount(); ++m) {
					if (mi.model->getMesh(m).type == Mesh::SKINNED) {
						last_mesh_id = m_poses->addMesh(mi.model->getMesh(m));
						break;
					}
				}
			}
			if (last_model_id < 0 || last_mesh_id < 0) continue;
			m_skinned.push_back(e);
			m_skinned_models.push_back(last_model_id);
			m_skinned_meshes.push_back(last_mesh_id);
			m_skinned_bones.push_back(mi.pose->count);
		}
		// entities handed to bindCulling must already be in the culling system: RenderModuleImpl added them through
		// CullingSystem::add when the model became ready (render_module.cpp:2880-2940) - on the World's shared context, the one
		// this module holds too. A renderer whose culling system lives on another context makes this fail: said loudly, and
		// retried at the next rebuild (the spheres then follow only through onModelInstanceMoved -> CullingSystem::set on the host).
		bool ok = bindModelInstances(m_bound.begin(), m_bound_radius.data(), (u32)m_bound.size());
		if (!m_skinned.empty() &&
			!m_poses->setInstances(m_skinned.begin(), m_skinned_models.data(), m_skinned_meshes.data(), m_skinned_bones.data(), (u32)m_skinned.size())) {
			fail("registering skinned instances", m_poses->lastError());
			ok = false;
		}
		bindBoneAttachments();
		m_bound_ok = ok;
	}

	// RenderModuleImpl::m_bone_attachments through the module's public accessors (render_module.h:404-413): attached entity ->
	// {parent entity, bone index, relative transform}; the parent must be one of the skinned instances registered above.
	// getBoneAttachmentRotation hands the stored quaternion out as Euler angles (render_module.cpp:484-487): Quat::fromEuler of
	// them equals the stored rotation only up to rounding. A tree that wants updateBoneAttachment's results bit for bit adds a
	// `LocalRigidTransform getBoneAttachmentTransform(EntityRef)` accessor to RenderModule (one line) - or takes
	// the attachments from the serialized scene, where the quaternion is stored as is (render_module.cpp:895-902).
	void bindBoneAttachments() {
		m_n_attachments = 0;
		if (!m_render_module || m_skinned.empty()) return;
		const ComponentType attachment_type = reflection::getComponentType("bone_attachment");
		std::vector<i32> entity, parent;
		std::vector<u32> instance, bone;
		std::vector<LmxLocalRigidTransform> relative;
		for (EntityPtr it = m_world.getFirstEntity(); it.isValid(); it = m_world.getNextEntity((EntityRef)it)) {
			const EntityRef e = (EntityRef)it;
			if (!m_world.hasComponent(e, attachment_type)) continue;
			const EntityPtr p = m_render_module->getBoneAttachmentParent(e);
			const int b = m_render_module->getBoneAttachmentBone(e);
			if (!p.isValid() || b < 0) continue; // updateBoneAttachment returns early for these (:378, :393-397)
			u32 inst = 0;
			while (inst < (u32)m_skinned.size() && m_skinned.begin()[inst].index != p.index) ++inst;
			if (inst == (u32)m_skinned.size() || (u32)b >= m_skinned_bones[inst]) continue; // parent without a pose / bone out of range (:386-397)
			const Vec3 pos = m_render_module->getBoneAttachmentPosition(e);
			Quat rot;
			rot.fromEuler(m_render_module->getBoneAttachmentRotation(e));
			LmxLocalRigidTransform rel = {{pos.x, pos.y, pos.z}, {rot.x, rot.y, rot.z, rot.w}};
			entity.push_back(e.index);
			parent.push_back(p.index);
			instance.push_back(inst);
			bone.push_back((u32)b);
			relative.push_back(rel);
		}
		if (entity.empty()) return;
		if (lmx_world_set_bone_attachments(m_ctx, (u32)entity.size(), entity.data(), parent.data(), instance.data(), bone.data(), relative.data()) == LMX_OK)
			m_n_attachments = (u32)entity.size();
	}

	ISystem& m_system;
	Engine& m_engine;
	World& m_world;
	LmxContext* m_ctx = nullptr;
	RenderModule* m_render_module = nullptr;
	UniquePtr<WorldSync> m_sync;
	UniquePtr<PoseBridge> m_poses;
	bool m_dirty = true;
	bool m_bound_ok = false;
	bool m_in_hand_back = false;
	u32 m_n_attachments = 0;
	std::string m_error, m_last_logged;
	std::vector<int32_t> m_moved_entities;
	std::vector<Transform> m_moved_transforms;
	std::vector<ComponentType> m_transformed_types;
	std::vector<u32> m_engine_write_stamp; // per entity: the frame in which the World announced a direct write to it
	u32 m_frame_stamp = 1;
	struct EntityList {
		std::vector<EntityRef> v;
		void clear() { v.clear(); }
		void push_back(EntityRef e) { v.push_back(e); }
		bool empty() const { return v.empty(); }
		size_t size() const { return v.size(); }
		const EntityRef* begin() const { return v.data(); }
	} m_bound, m_skinned;
	std::vector<float> m_bound_radius;
	std::vector<i32> m_skinned_models, m_skinned_meshes;
	std::vector<u32> m_skinned_bones;
};

struct Mi355System final : ISystem {
	explicit Mi355System(Engine& engine) : m_engine(engine) {}
	const char* getName() const override { return "mi355"; }
	void serialize(OutputMemoryStream&) const override {}
	bool deserialize(i32, InputMemoryStream&) override { return true; }
	void createModules(World& world) override {
		world.addModule(UniquePtr<Mi355Module>::create(m_engine.getAllocator(), *this, m_engine, world)); // world.cpp:218-235
	}
	Engine& m_engine;
};

} // namespace Lumix

LUMIX_PLUGIN_ENTRY(mi355) {
	return LUMIX_NEW(engine.getAllocator(), Lumix::Mi355System)(engine);
}
