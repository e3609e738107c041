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
#include "shell_engine.h"

extern "C" int ref_cs_is_added(void* cs, int32_t entity);
extern "C" void ref_cs_set(void* cs, int32_t entity, const double* pos, float radius);

namespace Lumix {
// link stubs: no component is registered, and the one static ResourceType of world.cpp only needs its hash
namespace reflection { Span<const RegisteredComponent> getComponents() { return {}; } }
ResourceType::ResourceType(const char* type_name) { type = RuntimeHash(type_name); }
} // namespace Lumix

using namespace Lumix;

namespace {
using lmx_ref::ShellEngine;

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
