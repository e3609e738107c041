// world_shim.cpp — TEST INFRASTRUCTURE. The reference's own World behind the ref_world_* C entry points.
//
// engine/world.cpp is compiled IN PLACE by oracle/Makefile (with core/string.cpp, stream.cpp, hash.cpp, log.cpp, arena_allocator.cpp,
// default_allocator.cpp from the same temporary, sync.h-patched copy of src/core that cull_shim.cpp uses): createEntity, setParent,
// setTransform, setLocalTransform, transformEntity / updateGlobalTransform, getLocalTransform, the per-component `transformed`
// delegates and World::serialize are reference object code. Nothing of those files is in this repository.
//
// What IS in this file, and is mine: an Engine that owns nothing but an allocator (World's constructor asks it for the allocator and
// the - empty - system list; serialize asks it to compress, restated from engine/engine.cpp:254-269 on the vendored LZ4), the two
// link stubs reflection::getComponents / ResourceType::ResourceType, the RenderModuleImpl::onModelInstanceMoved hook
// (render_module.cpp:1544-1554: culling set(pos, radius * max scale)) bound to the World's `transformed` delegate of a component
// type, and the extern "C" entry points.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "core/allocator.h"
#include "core/array.h"
#include "core/delegate_list.h"
#include "core/hash.h"
#include "core/math.h"
#include "core/page_allocator.h"
#include "core/path.h"
#include "core/stream.h"
#include "core/string.h"
#include "engine/engine.h"
#include "engine/plugin.h"
#include "engine/reflection.h"
#include "engine/resource.h"
#include "engine/world.h"
#include "lmx_types.h"

extern "C" int LZ4_compress_fast(const char* src, char* dst, int srcSize, int dstCapacity, int acceleration);
extern "C" int LZ4_decompress_safe(const char* src, char* dst, int compressedSize, int dstCapacity);
extern "C" int LZ4_compressBound(int inputSize);
extern "C" int ref_cs_is_added(void* cs, int32_t entity);
extern "C" void ref_cs_set(void* cs, int32_t entity, const double* pos, float radius);

namespace Lumix {
// link stubs: no component is registered, and the one static ResourceType of world.cpp only needs its hash
namespace reflection { Span<const RegisteredComponent> getComponents() { return {}; } }
ResourceType::ResourceType(const char* type_name) { type = RuntimeHash(type_name); }
} // namespace Lumix

using namespace Lumix;

namespace {
struct HeapAllocator final : IAllocator {
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

struct NoSystems final : SystemManager {
	NoSystems(IAllocator& a) : systems(a), libraries(a), loaded(a) {}
	void initSystems() override {}
	void unload(ISystem*) override {}
	ISystem* load(const char*) override { return nullptr; }
	void addSystem(ISystem*, void*) override {}
	void update(float) override {}
	ISystem* getSystem(const char*) override { return nullptr; }
	const Array<ISystem*>& getSystems() const override { return systems; }
	const Array<void*>& getLibraries() const override { return libraries; }
	void* getLibrary(ISystem*) const override { return nullptr; }
	DelegateList<void(void*)>& libraryLoaded() override { return loaded; }
	Array<ISystem*> systems;
	Array<void*> libraries;
	DelegateList<void(void*)> loaded;
};

[[noreturn]] void unused() { abort(); }

struct ShellEngine final : Engine {
	ShellEngine() : systems(heap) {}
	void init() override {}
	World& createWorld() override { unused(); }
	void destroyWorld(World&) override {}
	void setMainWindow(os::WindowHandle) override {}
	os::WindowHandle getMainWindow() override { return os::WindowHandle(); }
	FileSystem& getFileSystem() override { unused(); }
	InputSystem& getInputSystem() override { unused(); }
	SystemManager& getSystemManager() override { return systems; }
	ResourceManagerHub& getResourceManager() override { unused(); }
	PageAllocator& getPageAllocator() override { unused(); }
	IAllocator& getAllocator() override { return heap; }
	EntityPtr instantiatePrefab(World&, const PrefabResource&, const DVec3&, const Quat&, const Vec3&, EntityMap&) override { unused(); }
	void startGame(World&) override {}
	void stopGame(World&) override {}
	void update(World&) override {}
	DeserializeProjectResult deserializeProject(InputMemoryStream&, Path&) override { unused(); }
	void serializeProject(OutputMemoryStream&, const Path&) const override {}
	float getLastTimeDelta() const override { return 0; }
	void setTimeMultiplier(float) override {}
	void pause(bool) override {}
	bool isPaused() const override { return false; }
	void nextFrame() override {}
	bool decompress(Span<const u8> src, Span<u8> dst) override { // engine/engine.cpp:254-258
		const int res = LZ4_decompress_safe((const char*)src.begin(), (char*)dst.begin(), (int)src.length(), (int)dst.length());
		return res == (int)dst.length();
	}
	bool compress(Span<const u8> src, OutputMemoryStream& dst) override { // engine/engine.cpp:260-269
		const int cap = LZ4_compressBound((int)src.length());
		const u64 start = dst.size();
		dst.resize(start + cap);
		// LZ4_compress_fast_extState(state, ..., acceleration 1) in the reference; LZ4_compress_fast is the same call on a local state
		const int res = LZ4_compress_fast((const char*)src.begin(), (char*)dst.getMutableData() + start, (int)src.length(), cap, 1);
		if (res == 0) return false;
		dst.resize(start + res);
		return true;
	}
	HeapAllocator heap;
	NoSystems systems;
};

const ComponentType MODEL_INSTANCE_TYPE = {0}; // any index works: the World only uses it to pick the `transformed` delegate list

struct Handle {
	ShellEngine engine;
	World* world = nullptr;
	void* culling = nullptr; // a ref_cs_create handle
	std::vector<float> model_radius;

	void onModelInstanceMoved(EntityRef entity) { // RenderModuleImpl::onModelInstanceMoved, render_module.cpp:1544-1554
		if (!culling || !ref_cs_is_added(culling, entity.index)) return;
		const Transform& tr = world->getTransform(entity);
		const double pos[3] = {tr.pos.x, tr.pos.y, tr.pos.z};
		ref_cs_set(culling, entity.index, pos, model_radius[entity.index] * maximum(tr.scale.x, tr.scale.y, tr.scale.z));
	}
};

Transform toRef(const LmxTransform* t) {
	static_assert(sizeof(Transform) == sizeof(LmxTransform), "Transform layout");
	Transform r;
	memcpy((void*)&r, t, sizeof(r));
	return r;
}
void fromRef(const Transform& t, LmxTransform* out) {
	memset(out, 0, sizeof(*out));
	memcpy(out->pos, &t.pos, sizeof(out->pos));
	memcpy(out->rot, &t.rot, sizeof(out->rot));
	memcpy(out->scale, &t.scale, sizeof(out->scale));
}
} // namespace

extern "C" {
#define REF_API __attribute__((visibility("default")))

REF_API void* ref_world_create(uint32_t n_entities) {
	Handle* h = new Handle;
	h->world = new World(h->engine);
	for (uint32_t i = 0; i < n_entities; ++i) {
		const EntityRef e = h->world->createEntity(DVec3(0), Quat::IDENTITY); // fresh world: indices 0, 1, 2, ...
		if (e.index != (i32)i) abort();
	}
	h->model_radius.assign(n_entities, -1.f);
	return h;
}
REF_API void ref_world_destroy(void* w) {
	Handle* h = (Handle*)w;
	delete h->world;
	delete h;
}
// entity creation with a transform (World::createEntity writes pos / rot, deserialize writes all three) - no propagation
REF_API void ref_world_init_transforms(void* w, uint32_t n, const int32_t* entity, const LmxTransform* tr) {
	Transform* transforms = const_cast<Transform*>(((Handle*)w)->world->getTransforms());
	for (uint32_t i = 0; i < n; ++i) transforms[entity[i]] = toRef(&tr[i]);
}
REF_API void ref_world_set_parents(void* w, uint32_t n, const int32_t* parent, const int32_t* child) {
	for (uint32_t i = 0; i < n; ++i) ((Handle*)w)->world->setParent(EntityPtr{parent[i]}, EntityRef{child[i]});
}
REF_API void ref_world_set_transforms(void* w, uint32_t n, const int32_t* entity, const LmxTransform* tr) {
	for (uint32_t i = 0; i < n; ++i) ((Handle*)w)->world->setTransform(EntityRef{entity[i]}, toRef(&tr[i]));
}
REF_API void ref_world_set_local_transforms(void* w, uint32_t n, const int32_t* entity, const LmxTransform* tr) {
	for (uint32_t i = 0; i < n; ++i) ((Handle*)w)->world->setLocalTransform(EntityRef{entity[i]}, toRef(&tr[i]));
}
REF_API void ref_world_get_transforms(void* w, uint32_t n, LmxTransform* out) {
	const Transform* transforms = ((Handle*)w)->world->getTransforms();
	for (uint32_t i = 0; i < n; ++i) fromRef(transforms[i], &out[i]);
}
REF_API void ref_world_get_local_transforms(void* w, uint32_t n, LmxTransform* out) {
	for (uint32_t i = 0; i < n; ++i) fromRef(((Handle*)w)->world->getLocalTransform(EntityRef{(i32)i}), &out[i]);
}
REF_API void ref_world_bind_culling(void* w, void* cs, uint32_t n, const int32_t* entity, const float* model_radius) {
	Handle* h = (Handle*)w;
	if (!h->culling) h->world->componentTransformed(MODEL_INSTANCE_TYPE).bind<&Handle::onModelInstanceMoved>(h);
	h->culling = cs;
	for (uint32_t i = 0; i < n; ++i) {
		if (h->model_radius[entity[i]] < 0) h->world->onComponentCreated(EntityRef{entity[i]}, MODEL_INSTANCE_TYPE, nullptr);
		h->model_radius[entity[i]] = model_radius[i];
	}
}
REF_API void ref_world_destroy_entity(void* w, int32_t entity) { ((Handle*)w)->world->destroyEntity(EntityRef{entity}); }
REF_API void ref_world_set_name(void* w, int32_t entity, const char* name) { ((Handle*)w)->world->setEntityName(EntityRef{entity}, name); }
// World::serialize (world.cpp:837-897): header, module list, flags, then the LZ4-compressed blob. Returns the size (also when > cap).
REF_API uint32_t ref_world_serialize(void* w, uint32_t flags, uint8_t* out, uint32_t cap) {
	Handle* h = (Handle*)w;
	OutputMemoryStream stream(h->engine.heap);
	h->world->serialize(stream, (WorldSerializeFlags)flags);
	if (stream.size() <= cap) memcpy(out, stream.data(), stream.size());
	return (uint32_t)stream.size();
}
} // extern "C"
