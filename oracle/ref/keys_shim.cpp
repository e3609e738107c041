// keys_shim.cpp — TEST INFRASTRUCTURE. Compiles the REFERENCE'S OWN PipelineImpl::createSortKeys (src/renderer/pipeline.cpp:3789-3968:
// bucket map, the per-page loops over DECAL / CURVE_DECAL / MESH renderables, LOD selection and transition, create_key, the
// Pose::frame stamping loop, the AUTOINSTANCED pairs) together with the records it works on (Sorter + Inserter, AutoInstancer, View,
// CullResult, PagedListIterator, MeshMaterial, LODMeshIndices, Mesh::Type, Model::getLODMeshIndices, ModelInstance, RenderableTypes,
// BucketDesc::Sort, CameraParams, the SORT_KEY_* / SORT_VALUE_* constants and make*SortKey / make*SortValue) into
// oracle/_ref/liblmx_ref.so. The code itself is NOT in this file: it is cut out of /root/reference at build time by
// oracle/ref/slice_sort_keys.py into a temporary gen/ directory (deleted after the compile, see oracle/Makefile) and included below.
// What IS in this file, and is mine: stand-ins for what the engine provides around that code (allocators, a one-worker jobs
// namespace, Renderer / Engine / RenderModule / World shells that hand out the arrays, Material, Model, Mesh, Pose, PoseProcessor
// as a recorder) and the extern "C" entry point with the signature of orc_create_sort_keys.
// pipeline.cpp cannot be compiled whole: it is the renderer (gpu back end, resource system, draw streams, job system).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "core/allocator.h"
#include "core/array.h"
#include "core/atomic.h"
#include "core/geometry.h"
#include "core/math.h"
#include "core/span.h"
#include "engine/lumix.h"
#include "lmx_types.h"

#define PROFILE_BLOCK(name)

namespace Lumix {

// core/atomic.h declares compareExchangePtr(void* volatile*, ...); core/linux/atomic.cpp (compiled in place into oracle/_ref/atomic.o)
// defines a different overload, so the declared one is provided here with the same builtin.
bool compareExchangePtr(void* volatile* value, void* exchange, void* comperand) { return __sync_bool_compare_and_swap(value, comperand, exchange); }

// Everything below - stand-ins and sliced code alike - lives in Lumix::keys_shim: liblmx_ref.so also holds the real PageAllocator,
// World, jobs:: ... (cull_shim.cpp / world_shim.cpp), and same-named inline members would be merged with them at link time.
// Unqualified names (Vec3, DVec3, Transform, Array, IAllocator, EntityRef, ...) still resolve to the enclosing Lumix.
namespace keys_shim {

// ---- stand-ins (mine) ----

struct MallocAllocator : IAllocator {
	void* allocate(size_t size, size_t align) override {
		void* p = nullptr;
		if (posix_memalign(&p, align < sizeof(void*) ? sizeof(void*) : align, size ? size : 1) != 0) abort();
		return p;
	}
	void deallocate(void* ptr) override { free(ptr); }
	void* reallocate(void* ptr, size_t new_size, size_t old_size, size_t align) override {
		if (new_size == 0) { free(ptr); return nullptr; }
		void* p = allocate(new_size, align);
		if (ptr) { memcpy(p, ptr, old_size < new_size ? old_size : new_size); free(ptr); }
		return p;
	}
};
struct ArenaAllocator : MallocAllocator {};
struct PageAllocator {
	enum { PAGE_SIZE = 4096 };
	void* allocate() { void* p = nullptr; if (posix_memalign(&p, PAGE_SIZE, PAGE_SIZE) != 0) abort(); return p; }
	void deallocate(void* mem) { free(mem); }
};
namespace jobs { // one worker, no contention
struct Mutex {};
struct MutexGuard { MutexGuard(Mutex&) {} };
struct Signal { int state = 0; };
inline u8 getWorkersCount() { return 1; }
} // namespace jobs
namespace profiler { inline void pushInt(const char*, int) {} }
struct TransientSlice { void* ptr = nullptr; u32 offset = 0, size = 0; };
struct TransientPool {};
struct Material {
	u8 getLayer() const { return m_layer; }
	u32 getSortKey() const { return m_sort_key; }
	u8 m_layer = 0;
	u32 m_sort_key = 0;
};
struct Pose { // renderer/pose.h: the two members createSortKeys touches
	u32 count = 0;
	AtomicI32 frame = 0xffFFffFF;
};
struct Mesh {
#include "gen/keys_mesh_type.inc"
	Type type;
};

// ---- src/renderer/model.h (sliced): MaterialIndex, MeshMaterial, LODMeshIndices; getLODMeshIndices inside my shell of Model ----
#include "gen/keys_model_structs.inc"
struct Model {
#include "gen/keys_lod_fn.inc"
	const LODMeshIndices* getLODIndices() const { return m_lod_indices; }
	float m_lod_distances[4];
	LODMeshIndices m_lod_indices[5];
};

// ---- src/renderer/render_module.h (sliced): RenderableTypes, ModelInstance; culling_system.h: CullResult; page_allocator.h: iterator ----
#include "gen/keys_model_instance.inc"
#include "gen/keys_cull_result.inc"
#include "gen/keys_paged_iter.inc"
#include "gen/keys_camera_params.inc"
struct BucketDesc {
#include "gen/keys_bucket_sort.inc"
};

struct DecalShell { Material* material = nullptr; }; // Decal / CurveDecal: createSortKeys reads .material only
struct World {
	const Transform* getTransforms() const { return transforms; }
	const Transform* transforms = nullptr;
};
struct RenderModule {
	RenderModule(IAllocator& a) : model_instances(a) {}
	Array<ModelInstance>& getModelInstances() { return model_instances; }
	World& getWorld() { return world; }
	DecalShell& getDecal(EntityRef e) { return decals[e.index]; }
	DecalShell& getCurveDecal(EntityRef e) { return curve_decals[e.index]; }
	Array<ModelInstance> model_instances;
	World world;
	DecalShell* decals = nullptr;
	DecalShell* curve_decals = nullptr;
};
struct Engine {
	float getLastTimeDelta() const { return time_delta; }
	PageAllocator& getPageAllocator() { return page_allocator; }
	float time_delta = 0;
	PageAllocator page_allocator;
};
struct Renderer {
	ArenaAllocator& getCurrentFrameAllocator() { return arena; }
	Engine& getEngine() { return engine; }
	float getLODMultiplier() const { return lod_multiplier; }
	TransientPool& getTransientPool() { return pool; }
	u32 frameNumber() const { return frame; }
	u32 getMaxSortKey() const { return max_sort_key; }
	ArenaAllocator arena;
	Engine engine;
	TransientPool pool;
	float lod_multiplier = 1;
	u32 frame = 0, max_sort_key = 0;
};

namespace {
// ---- src/renderer/pipeline.cpp (sliced): DrawCommandTypes, floatFlip, shifts, key / value makers ----
#include "gen/keys_consts.inc"
} // namespace

struct PipelineImpl {
	struct Bucket { BucketDesc::Sort sort = BucketDesc::DEFAULT; }; // PipelineImpl::Bucket minus its draw streams
	// ---- src/renderer/pipeline.cpp (sliced): Sorter, AutoInstancer, View ----
#include "gen/keys_sorter.inc"
#include "gen/keys_instancer.inc"
#include "gen/keys_view.inc"

	struct PoseProcessor { // the reference batches the instances and computes dual quaternions on a job; here: record who was pushed
		PoseProcessor(PipelineImpl& p, ArenaAllocator&) : pipeline(p) {}
		void push(ModelInstance* mi) { pipeline.pushed_poses.push_back(mi); }
		PipelineImpl& pipeline;
	};

	PipelineImpl(Renderer& r, RenderModule* m) : m_renderer(r), m_module(m) {}
	void queueMaterialOverrideRefresh(EntityRef e) { refresh_queue.push_back(e.index); }

	void createSortKeys(View& view) {
		// ---- src/renderer/pipeline.cpp (sliced): createSortKeys up to jobs::runOnWorkers ----
#include "gen/keys_head.inc"
		(void)transient_pool;
		{ // jobs::runOnWorkers with one worker: the lambda's body, once
#include "gen/keys_worker.inc"
		}
	}

	Renderer& m_renderer;
	RenderModule* m_module;
	Viewport m_viewport; // core/geometry.h
	std::vector<ModelInstance*> pushed_poses;
	std::vector<i32> refresh_queue;
};

} // namespace keys_shim
} // namespace Lumix

using namespace Lumix;
using namespace Lumix::keys_shim;

namespace {
struct KeysOut { // = OrcKeysOut of oracle/lmx_oracle.c
	uint64_t* keys; uint64_t* values; uint32_t cap_pairs, n_pairs;
	uint32_t* group_offsets;
	uint64_t* group_values; uint32_t cap_instanced, n_instanced;
	int32_t* poses; uint32_t n_poses;
	int32_t* dirty; uint32_t n_dirty;
	uint32_t n_groups;
};

CullResult* append_pages(PageAllocator& pa, CullResult*& first, CullResult* last, const int32_t* ids, uint32_t n, RenderableTypes type) {
	for (uint32_t done = 0; done < n;) { // pages as CullingSystemImpl fills them: one type per page, up to lengthOf(entities) each
		CullResult* page = new (NewPlaceholder(), pa.allocate()) CullResult;
		page->header.type = (u8)type;
		const uint32_t c = n - done < (uint32_t)lengthOf(page->entities) ? n - done : (uint32_t)lengthOf(page->entities);
		for (uint32_t i = 0; i < c; ++i) page->entities[i] = EntityRef{ids[done + i]};
		page->header.count = c;
		done += c;
		if (last) last->header.next = page; else first = page;
		last = page;
	}
	return last;
}
} // namespace

// PipelineImpl::createSortKeys for one view and one worker. Same signature and outputs as orc_create_sort_keys (oracle/lmx_oracle.c).
extern "C" __attribute__((visibility("default"))) int ref_create_sort_keys(const LmxKeysView* kv, uint32_t max_sort_key, const int32_t* mesh_ids, uint32_t n_mesh,
	const int32_t* decal_ids, uint32_t n_decal, const int32_t* curve_ids, uint32_t n_curve, const LmxKeysModel* models, const uint8_t* mesh_types,
	const int32_t* model, const uint32_t* material_offset, const LmxMeshMaterial* mesh_materials, float* lod, const uint8_t* flags, const uint8_t* dirty,
	uint32_t* pose_frame, const uint32_t* decal_key, const uint8_t* decal_layer, const uint32_t* curve_key, const uint8_t* curve_layer,
	const double* pos_xyz, KeysOut* out) {
	uint32_t n_entities = 0; // the arrays are entity-indexed: size them by the largest visible id
	for (uint32_t i = 0; i < n_mesh; ++i) if ((uint32_t)mesh_ids[i] + 1 > n_entities) n_entities = (uint32_t)mesh_ids[i] + 1;
	for (uint32_t i = 0; i < n_decal; ++i) if ((uint32_t)decal_ids[i] + 1 > n_entities) n_entities = (uint32_t)decal_ids[i] + 1;
	for (uint32_t i = 0; i < n_curve; ++i) if ((uint32_t)curve_ids[i] + 1 > n_entities) n_entities = (uint32_t)curve_ids[i] + 1;

	Renderer renderer;
	renderer.lod_multiplier = kv->lod_multiplier;
	renderer.engine.time_delta = kv->time_delta;
	renderer.frame = kv->frame_number;
	renderer.max_sort_key = max_sort_key;
	MallocAllocator heap;
	RenderModule module(heap);

	// engine-side objects from the flat tables
	int32_t n_models = -1;
	for (uint32_t i = 0; i < n_mesh; ++i) if (model[mesh_ids[i]] > n_models) n_models = model[mesh_ids[i]];
	++n_models;
	std::vector<Model> model_objs((size_t)n_models);
	std::vector<std::vector<Mesh>> model_meshes((size_t)n_models);
	for (int32_t m = 0; m < n_models; ++m) {
		memcpy(model_objs[m].m_lod_distances, models[m].lod_distances, sizeof(float) * 4);
		for (int k = 0; k < 5; ++k) model_objs[m].m_lod_indices[k] = {models[m].lod_indices[k].from, models[m].lod_indices[k].to};
		model_meshes[m].resize(models[m].mesh_count);
		for (uint32_t k = 0; k < models[m].mesh_count; ++k) model_meshes[m][k].type = mesh_types[models[m].first_mesh + k] == LMX_MESH_SKINNED ? Mesh::SKINNED : Mesh::RIGID;
	}
	std::vector<Transform> transforms(n_entities, Transform{DVec3(0), Quat(0, 0, 0, 1), Vec3(1)});
	std::vector<Pose> poses(n_entities);
	std::vector<Material> materials;       // one Material object per MeshMaterial entry in use (layer only; sort_key sits in MeshMaterial)
	std::vector<MeshMaterial> mesh_mats;
	std::vector<uint32_t> mm_first(n_entities, 0);
	size_t total_mm = 0;
	for (uint32_t i = 0; i < n_mesh; ++i) { const int32_t e = mesh_ids[i]; if (model[e] >= 0) total_mm += models[model[e]].mesh_count; }
	materials.reserve(total_mm + n_decal + n_curve);
	mesh_mats.reserve(total_mm);
	module.model_instances.resize(n_entities);
	std::vector<uint8_t> is_instance(n_entities, 0);
	for (uint32_t i = 0; i < n_mesh; ++i) {
		const int32_t e = mesh_ids[i];
		if (model[e] < 0 || is_instance[e]) continue;
		is_instance[e] = 1;
		const LmxKeysModel& km = models[model[e]];
		mm_first[e] = (uint32_t)mesh_mats.size();
		for (uint32_t k = 0; k < km.mesh_count; ++k) {
			const LmxMeshMaterial& src = mesh_materials[material_offset[e] + k];
			Material mat;
			mat.m_layer = src.layer;
			materials.push_back(mat);
			MeshMaterial mm;
			mm.material = &materials.back();
			mm.sort_key = src.sort_key;
			mm.material_index = MaterialIndex(0);
			mm.flags = MeshMaterial::NONE;
			mesh_mats.push_back(mm);
		}
	}
	for (uint32_t e = 0; e < n_entities; ++e) {
		transforms[e].pos = DVec3(pos_xyz[3 * (size_t)e], pos_xyz[3 * (size_t)e + 1], pos_xyz[3 * (size_t)e + 2]);
		if (!is_instance[e]) continue;
		ModelInstance& mi = module.model_instances[e];
		mi.model = &model_objs[model[e]];
		mi.meshes = model_meshes[model[e]].data();
		mi.mesh_materials = Span<MeshMaterial>(mesh_mats.data() + mm_first[e], models[model[e]].mesh_count);
		mi.pose = &poses[e];
		mi.pose->frame = (i32)pose_frame[e];
		mi.lod = lod[e];
		mi.flags = ModelInstance::Flags(flags[e]);
		mi.mesh_count = (u16)models[model[e]].mesh_count;
		mi.dirty = dirty[e] != 0;
	}
	module.world.transforms = transforms.data();
	std::vector<DecalShell> decals(n_entities), curves(n_entities);
	for (uint32_t i = 0; i < n_decal; ++i) {
		const int32_t e = decal_ids[i];
		if (decals[e].material) continue;
		Material mat;
		mat.m_layer = decal_layer[e];
		mat.m_sort_key = decal_key[e];
		materials.push_back(mat);
		decals[e].material = &materials.back();
	}
	for (uint32_t i = 0; i < n_curve; ++i) {
		const int32_t e = curve_ids[i];
		if (curves[e].material) continue;
		Material mat;
		mat.m_layer = curve_layer[e];
		mat.m_sort_key = curve_key[e];
		materials.push_back(mat);
		curves[e].material = &materials.back();
	}
	module.decals = decals.data();
	module.curve_decals = curves.data();

	int rc = 0;
	{
		PipelineImpl pipeline(renderer, &module);
		pipeline.m_viewport.pos = DVec3(kv->lod_ref_point[0], kv->lod_ref_point[1], kv->lod_ref_point[2]);
		PipelineImpl::View view(renderer.arena, renderer.engine.page_allocator);
		view.cp.pos = DVec3(kv->camera_pos[0], kv->camera_pos[1], kv->camera_pos[2]);
		view.cp.is_shadow = kv->is_shadow != 0;
		memcpy(view.layer_to_bucket, kv->layer_to_bucket, 255);
		for (int b = 0; b < 256; ++b) view.buckets.emplace().sort = kv->bucket_depth_sorted[b] ? BucketDesc::DEPTH : BucketDesc::DEFAULT;
		// the visible lists as CullResult pages, in the order the plain-C restatement walks them: DECAL, CURVE_DECAL, MESH
		CullResult* first = nullptr;
		CullResult* last = nullptr;
		last = append_pages(renderer.engine.page_allocator, first, last, decal_ids, n_decal, RenderableTypes::DECAL);
		last = append_pages(renderer.engine.page_allocator, first, last, curve_ids, n_curve, RenderableTypes::CURVE_DECAL);
		// entities that are in the MESH list without being model instances cannot exist in the reference: the restatement skips them
		std::vector<int32_t> mesh_only;
		mesh_only.reserve(n_mesh);
		for (uint32_t i = 0; i < n_mesh; ++i) if (model[mesh_ids[i]] >= 0) mesh_only.push_back(mesh_ids[i]);
		last = append_pages(renderer.engine.page_allocator, first, last, mesh_only.data(), (uint32_t)mesh_only.size(), RenderableTypes::MESH);
		view.renderables = first;

		pipeline.createSortKeys(view);

		// ---- outputs ----
		out->n_pairs = out->n_instanced = out->n_poses = out->n_dirty = out->n_groups = 0;
		for (auto* p = view.sorter.first_page; p; p = p->header.next) {
			for (u32 i = 0; i < p->header.count; ++i) {
				if (out->n_pairs >= out->cap_pairs) { rc = 1; break; }
				out->keys[out->n_pairs] = (uint64_t)p->keys[i];
				out->values[out->n_pairs] = (uint64_t)p->values[i];
				++out->n_pairs;
			}
		}
		PipelineImpl::AutoInstancer& inst = view.instancers[0];
		memset(out->group_offsets, 0, sizeof(uint32_t) * (max_sort_key + 2));
		for (u32 k = 0; k <= max_sort_key && rc == 0; ++k) {
			out->group_offsets[k] = out->n_instanced;
			if (inst.instances[k].begin) ++out->n_groups;
			for (auto* g = inst.instances[k].begin; g; g = g->next) {
				for (u32 i = 0; i < g->count; ++i) {
					if (out->n_instanced >= out->cap_instanced) { rc = 2; break; }
					out->group_values[out->n_instanced++] = g->renderables[i];
				}
			}
		}
		out->group_offsets[max_sort_key + 1] = out->n_instanced;
		for (ModelInstance* mi : pipeline.pushed_poses) out->poses[out->n_poses++] = (int32_t)(mi - module.model_instances.begin());
		for (i32 e : pipeline.refresh_queue) out->dirty[out->n_dirty++] = e;
		for (uint32_t e = 0; e < n_entities; ++e) {
			if (!is_instance[e]) continue;
			lod[e] = module.model_instances[e].lod;
			pose_frame[e] = (uint32_t)(i32)poses[e].frame;
		}
		for (CullResult* p = first; p;) { CullResult* n = p->header.next; renderer.engine.page_allocator.deallocate(p); p = n; }
	}
	return rc;
}
