// lumix_compat.h — the handful of LumixEngine declarations the hot-path seam touches, for STANDALONE builds of the
// adapters of lumixengine_amd/host/ (TEST INFRASTRUCTURE: it lives in tests/cpp/, add -Itests/cpp). Inside a LumixEngine tree define LMX_WITH_LUMIX_HEADERS and the adapters
// include the engine's own headers instead; the layouts below are byte-compatible with
//   EntityRef / EntityPtr      src/engine/lumix.h:11-47
//   Vec3 / DVec3 / Quat        src/core/math.h
//   ShiftedFrustum             src/core/geometry.h:102-153   (== LmxShiftedFrustum, 256 B)
//   CullResult                 src/renderer/culling_system.h:17-56   (one 4096-byte page, 1020 ids)
//   CullingSystem              src/renderer/culling_system.h:58-77
//   PageAllocator              src/core/page_allocator.h:16-33 (allocate / deallocate of 4096-byte pages)
//   Quat / Transform           src/core/math.h:200-327 (Transform = 56 B == LmxTransform)
//   EntityPtr, World (read side), Pose, Model (skeleton side), Mesh::Skin, ModelInstance, RenderModule (pose hand-off)
//                              the members world_sync.h / pose_bridge.h call, as small in-memory mocks
// These are interface shims for standalone test builds: they restate the public shape of those types (CullResult and the
// CullingSystem vtable necessarily member for member); inside the engine the real headers are used instead.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "lmx_types.h"

namespace Lumix {

using u8 = uint8_t;
using u32 = uint32_t;
using i32 = int32_t;

using i16 = int16_t;
using u16 = uint16_t;

struct EntityRef {
	i32 index = -1;
	bool operator==(const EntityRef& rhs) const { return rhs.index == index; }
};
struct EntityPtr {
	i32 index = -1;
	bool isValid() const { return index >= 0; }
	explicit operator EntityRef() const { return EntityRef{index}; }
};

struct Vec3 { float x, y, z; };
struct Vec4 { float x, y, z, w; };
struct DVec3 { double x, y, z; };
struct Quat { float x, y, z, w; };
struct Transform { DVec3 pos; Quat rot; Vec3 scale; };
struct LocalRigidTransform { Vec3 pos; Quat rot; };
static_assert(sizeof(Transform) == sizeof(LmxTransform) && sizeof(LocalRigidTransform) == sizeof(LmxLocalRigidTransform), "handed to the C ABI as is");

template <typename T> struct Span {
	T* m_begin = nullptr;
	T* m_end = nullptr;
	Span() {}
	Span(T* b, uint64_t len) : m_begin(b), m_end(b + len) {}
	T& operator[](u32 i) const { return m_begin[i]; }
	u32 length() const { return (u32)(m_end - m_begin); }
	T* begin() const { return m_begin; }
	T* end() const { return m_end; }
};

struct alignas(16) ShiftedFrustum {
	float xs[8], ys[8], zs[8], ds[8];
	Vec3 points[8];
	DVec3 origin;
};
static_assert(sizeof(ShiftedFrustum) == sizeof(LmxShiftedFrustum), "ShiftedFrustum is handed to lmx_cull as is");

struct PageAllocator { // 4096-byte pages with a free list, like core/page_allocator.cpp:41-64
	enum { PAGE_SIZE = 4096 };
	~PageAllocator() { for (void* p : m_free) free(p); }
	void* allocate() {
		std::lock_guard<std::mutex> guard(m_mutex);
		if (!m_free.empty()) { void* p = m_free.back(); m_free.pop_back(); return p; }
		return aligned_alloc(PAGE_SIZE, PAGE_SIZE);
	}
	void deallocate(void* mem) {
		std::lock_guard<std::mutex> guard(m_mutex);
		m_free.push_back(mem);
	}
private:
	std::mutex m_mutex;
	std::vector<void*> m_free;
};

struct CullResult {
	void merge(CullResult* other) {
		CullResult** last = &header.next;
		while (*last) last = &(*last)->header.next;
		*last = other;
	}
	u32 count() const {
		u32 res = 0;
		for (const CullResult* j = this; j; j = j->header.next) res += j->header.count;
		return res;
	}
	void free(PageAllocator& allocator) {
		CullResult* i = this;
		while (i) { CullResult* tmp = i; i = i->header.next; allocator.deallocate(tmp); }
	}
	template <typename F> void forEach(F&& f) const {
		for (const CullResult* j = this; j; j = j->header.next)
			for (u32 i = 0, c = j->header.count; i < c; ++i) f(j->entities[i]);
	}
	struct {
		CullResult* next = nullptr;
		u32 count = 0;
		u8 type;
	} header;
	EntityRef entities[(4096 - sizeof(header)) / sizeof(EntityRef)];
};
static_assert(sizeof(CullResult) == PageAllocator::PAGE_SIZE, "CullResult is one page");

struct CullingSystem {
	virtual ~CullingSystem() {}
	virtual CullResult* cull(const ShiftedFrustum& frustum, u8 type) = 0;
	virtual CullResult* cull(const ShiftedFrustum& frustum) = 0;
	virtual bool isAdded(EntityRef entity) = 0;
	virtual void add(EntityRef entity, u8 type, const DVec3& pos, float radius) = 0;
	virtual void remove(EntityRef entity) = 0;
	virtual void setPosition(EntityRef entity, const DVec3& pos) = 0;
	virtual void setRadius(EntityRef entity, float radius) = 0;
	virtual void set(EntityRef entity, const DVec3& pos, float radius) = 0;
	virtual float getRadius(EntityRef entity) = 0;
};

// ---- mocks of the engine objects the world / pose bridges read and write (standalone tests only) ---------------------------
struct World { // read side of engine/world.h:49-209 + the transform array
	std::vector<Transform> transforms; // m_transforms
	std::vector<Transform> locals;     // Hierarchy::local_transform
	std::vector<i32> parents;
	const Transform* getTransforms() const { return transforms.data(); }
	bool hasEntity(EntityRef e) const { return e.index >= 0 && e.index < (i32)transforms.size(); }
	EntityPtr getParent(EntityRef e) const { return EntityPtr{parents[e.index]}; }
	Transform getLocalTransform(EntityRef e) const { return parents[e.index] < 0 ? transforms[e.index] : locals[e.index]; }
	EntityPtr getFirstEntity() const { return EntityPtr{transforms.empty() ? -1 : 0}; }
	EntityPtr getNextEntity(EntityRef e) const { return EntityPtr{e.index + 1 < (i32)transforms.size() ? e.index + 1 : -1}; }
};

struct Pose { // renderer/pose.h:15-35
	bool is_absolute = false;
	u32 count = 0;
	Vec3* positions = nullptr;
	Quat* rotations = nullptr;
};

struct Mesh { // renderer/model.h:81-131 (what evaluateSkin reads)
	struct Skin { Vec4 weights; i16 indices[4]; };
	struct VertexArray { std::vector<Vec3> v; int size() const { return (int)v.size(); } const Vec3* begin() const { return v.data(); } } vertices;
	struct SkinArray { std::vector<Skin> v; int size() const { return (int)v.size(); } const Skin* begin() const { return v.data(); } } skin;
};

struct Model { // renderer/model.h:140-215 (skeleton side)
	struct Bone { LocalRigidTransform transform; LocalRigidTransform relative_transform; };
	std::vector<Bone> bones;
	std::vector<i16> parents;
	int first_nonroot = 0;
	std::vector<Mesh> meshes;
	float origin_bounding_radius = 1.f;
	Span<const Bone> getBones() const { return Span<const Bone>(bones.data(), bones.size()); }
	Span<const i16> getParents() const { return Span<const i16>(parents.data(), parents.size()); }
	int getFirstNonrootBoneIndex() const { return first_nonroot; }
	int getMeshCount() const { return (int)meshes.size(); }
	const Mesh& getMesh(u32 i) const { return meshes[i]; }
	float getOriginBoundingRadius() const { return origin_bounding_radius; }
};

struct ModelInstance { // renderer/render_module.h:206-226 (the members the bridges touch)
	Model* model = nullptr;
	Pose* pose = nullptr;
};

struct RenderModule { // renderer/render_module.h:402-403, 461-462
	std::vector<ModelInstance> instances; // indexed by entity.index
	Pose* lockPose(EntityRef e) { return instances[e.index].pose; }
	void unlockPose(EntityRef, bool) {}
	Span<ModelInstance> getModelInstances() { return Span<ModelInstance>(instances.data(), instances.size()); }
};

} // namespace Lumix
