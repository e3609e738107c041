// lumix_compat.h — the handful of LumixEngine declarations the hot-path seam touches, for STANDALONE builds of the
// adapters in this directory (tests, tools). Inside a LumixEngine tree define LMX_WITH_LUMIX_HEADERS and the adapters
// include the engine's own headers instead; the layouts below are byte-compatible with
//   EntityRef / EntityPtr      src/engine/lumix.h:11-47
//   Vec3 / DVec3 / Quat        src/core/math.h
//   ShiftedFrustum             src/core/geometry.h:102-153   (== LmxShiftedFrustum, 256 B)
//   CullResult                 src/renderer/culling_system.h:17-56   (one 4096-byte page, 1020 ids)
//   CullingSystem              src/renderer/culling_system.h:58-77
//   PageAllocator              src/core/page_allocator.h:16-33 (allocate / deallocate of 4096-byte pages)
// Nothing here is copied from the reference sources; it restates the public shape of those types.
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

struct EntityRef {
	i32 index = -1;
	bool operator==(const EntityRef& rhs) const { return rhs.index == index; }
};

struct Vec3 { float x, y, z; };
struct DVec3 { double x, y, z; };

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

} // namespace Lumix
