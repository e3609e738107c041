// gpu_culling_system.h — C++ host side of the drop-in: a CullingSystem (src/renderer/culling_system.h:58-77) whose
// cull() runs on an MI355X through the C ABI of liblumix_mi355.so.
//
// RenderModuleImpl owns its culling system through `UniquePtr<CullingSystem> m_culling_system`, created by the static
// factory CullingSystem::create (src/renderer/render_module.cpp:3407,3569). GpuCullingSystem is a drop-in for that
// member: same virtuals, same argument meaning, same ownership rules — cull() returns a linked list of 4 KiB
// CullResult pages taken from the engine's PageAllocator, each page tagged with one renderable type, which the caller
// frees with CullResult::free (src/renderer/culling_system.cpp:388-396); it returns nullptr when nothing is resident
// (culling_system.cpp:322). Error convention of the reference: no exceptions; failures are logged and surface as an
// empty result / ignored call (see lastError()).
//
// Threading: add/remove/set* arrive on the update thread; cull() may be called for several views per
// frame from render jobs. The C ABI context is not re-entrant, and it is SHARED with the plugin's module (one LmxContext per
// World: the culling set, the transform hierarchy that refreshes its spheres and the skinning tables are one device-side
// object, as the reference's CullingSystem and RenderModuleImpl::onModelInstanceMoved work on one object): every call takes
// the context's own recursive lock (lmx_ctx_lock), not a lock of this adapter. A cull() holds it only while it ENQUEUES (the cull,
// the pack of its record, the copy into the view's pinned buffer: lmx_cull_map_begin); the wait for its own view's event and the
// page building (lmx_cull_map_end, toPages) run outside, so concurrent views overlap everything but the enqueue. Each cull()
// RESERVES its result slot (lmx_cull_view_acquire) before it enqueues and releases it when the ids are in the engine's pages: the
// reference hands every caller an independent list (culling_system.cpp:321-369), so a slot whose record is still being read is never
// culled into again - with all LMX_MAX_VIEWS slots held, the next caller waits for a release (any number of concurrent callers is
// correct; LMX_MAX_VIEWS of them overlap). cullMany() culls all views of a frame in one pass over the spheres with one host wait
// (INTEGRATION.md shows the Pipeline change that calls it).
#pragma once

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "lumix_mi355.h"

#ifdef LMX_WITH_LUMIX_HEADERS
	#include "core/geometry.h"
	#include "core/log.h"
	#include "core/math.h"
	#include "core/page_allocator.h"
	#include "renderer/culling_system.h"
#else
	#include "lumix_compat.h"
#endif

namespace Lumix {

struct GpuCullingSystem final : CullingSystem {
	// a context of its own (tests, tools, an engine without the plugin module)
	GpuCullingSystem(PageAllocator& page_allocator, int device = 0) : m_page_allocator(page_allocator) {
		if (lmx_ctx_create(device, &m_ctx) != LMX_OK) noContext();
	}
	// THE drop-in form: the context of `world_key` (the address of the World the RenderModule belongs to), shared with every other
	// adapter of that World - the Mi355Module that propagates its transforms finds the same culling set under the same key
	GpuCullingSystem(PageAllocator& page_allocator, const void* world_key, int device = 0) : m_page_allocator(page_allocator), m_shared(true) {
		if (lmx_ctx_acquire_shared(world_key, device, &m_ctx) != LMX_OK) noContext();
	}
	~GpuCullingSystem() override {
		if (m_shared) lmx_ctx_release_shared(m_ctx);
		else lmx_ctx_destroy(m_ctx);
	}

	bool isValid() const { return m_ctx != nullptr; }
	// The reference's add / remove / set* never stall a frame (culling_system.cpp:131-258). Here the sorted set is re-sorted now and then
	// (O(n): 0.5 s at 10 M entities); with this on, that happens on a worker thread against a second copy of the sets, and the copy is
	// traded with the live set inside a later cull (LMX_CULL_OPT_ASYNC_COMPACTION). createGpuCullingSystem() switches it on.
	bool setAsyncCompaction(bool on) {
		CtxLock guard(m_ctx);
		return m_ctx && check(lmx_cull_set_option(m_ctx, LMX_CULL_OPT_ASYNC_COMPACTION, on ? 1 : 0));
	}
	const std::string& lastError() const { return m_error; }
	LmxContext* context() { return m_ctx; }

	CullResult* cull(const ShiftedFrustum& frustum, u8 type) override { return cullInternal(frustum, type); } // culling_system.cpp:310-314
	CullResult* cull(const ShiftedFrustum& frustum) override { return cullInternal(frustum, 0xff); }           // :316-319

	// add / remove / set* arrive on the update thread while render jobs may be inside cull(): every call takes the context's lock
	// (uncontended in the engine's frame structure)
	bool isAdded(EntityRef entity) override {
		CtxLock guard(m_ctx);
		return m_ctx && lmx_cull_is_added(m_ctx, entity.index) != 0;
	}
	void add(EntityRef entity, u8 type, const DVec3& pos, float radius) override {
		const double p[3] = {pos.x, pos.y, pos.z};
		CtxLock guard(m_ctx);
		check(lmx_cull_add(m_ctx, entity.index, type, p, radius));
	}
	void remove(EntityRef entity) override {
		CtxLock guard(m_ctx);
		check(lmx_cull_remove(m_ctx, entity.index));
	}
	void setPosition(EntityRef entity, const DVec3& pos) override {
		const double p[3] = {pos.x, pos.y, pos.z};
		CtxLock guard(m_ctx);
		check(lmx_cull_set_position(m_ctx, entity.index, p));
	}
	void setRadius(EntityRef entity, float radius) override {
		CtxLock guard(m_ctx);
		check(lmx_cull_set_radius(m_ctx, entity.index, radius));
	}
	void set(EntityRef entity, const DVec3& pos, float radius) override {
		const double p[3] = {pos.x, pos.y, pos.z};
		CtxLock guard(m_ctx);
		check(lmx_cull_set(m_ctx, entity.index, p, radius));
	}
	float getRadius(EntityRef entity) override {
		float r = 0;
		CtxLock guard(m_ctx);
		check(lmx_cull_get_radius(m_ctx, entity.index, &r));
		return r;
	}

	// All views of a frame in ONE pass over the spheres and ONE host wait (the reference culls its 4 shadow cascades, the main view
	// and a light query one after the other, each a jobs::forEach over every cell: pipeline.cpp:1036-1045, :1252-1258). out[f] receives
	// what cull(frusta[f], type) would return: a list of CullResult pages (nullptr when nothing is resident), freed by the caller.
	bool cullMany(const ShiftedFrustum* frusta, u32 n_frusta, u8 type, CullResult** out) {
		for (u32 f = 0; f < n_frusta; ++f) out[f] = nullptr;
		if (!m_ctx || n_frusta == 0 || n_frusta > LMX_MAX_FRUSTA) return false;
		ViewSlot slot(m_ctx); // reserved until the ids are in the engine's pages (released by the destructor)
		if (!slot.ok) return busy();
		const uint32_t view = slot.view;
		{ // enqueue under the context's lock: the cull, the pack of its records and their copy into the view's pinned buffer
			CtxLock guard(m_ctx);
			uint32_t n_static = 0, n_bound = 0, n_overflow = 0;
			if (!check(lmx_cull_update_stats(m_ctx, &n_static, &n_bound, &n_overflow, nullptr))) return false;
			if (n_static + n_bound + n_overflow == 0) return true; // no cells: culling_system.cpp:322
			if (!check(lmx_cull(m_ctx, view, reinterpret_cast<const LmxShiftedFrustum*>(frusta), n_frusta, type))) return false;
			if (!check(lmx_cull_map_begin(m_ctx, view, n_frusta))) return false;
		}
		// the host wait and the page building run outside the lock: other render jobs enqueue their views meanwhile
		uint32_t counts[LMX_MAX_FRUSTA * LMX_MAX_TYPES];
		const int32_t* ids[LMX_MAX_FRUSTA];
		if (!checkUnlocked(lmx_cull_map_end(m_ctx, view, n_frusta, ids, counts))) return false;
		for (u32 f = 0; f < n_frusta; ++f) out[f] = toPages(ids[f], counts + f * LMX_MAX_TYPES, type);
		return true;
	}

private:
	struct CtxLock { // the context's own recursive lock: shared with the other adapters of the World
		explicit CtxLock(LmxContext* c) : ctx(c) { lmx_ctx_lock(ctx); }
		~CtxLock() { lmx_ctx_unlock(ctx); }
		CtxLock(const CtxLock&) = delete;
		CtxLock& operator=(const CtxLock&) = delete;
		LmxContext* ctx;
	};

	// A result slot of the context, reserved for the lifetime of this object. Taken BEFORE the context lock (lmx_cull_view_acquire may
	// wait for another caller's release, and that caller needs the lock to get there).
	struct ViewSlot {
		explicit ViewSlot(LmxContext* c) : ctx(c) { ok = c && lmx_cull_view_acquire(c, &view, 5000) == LMX_OK; }
		~ViewSlot() { if (ok) lmx_cull_view_release(ctx, view); }
		ViewSlot(const ViewSlot&) = delete;
		ViewSlot& operator=(const ViewSlot&) = delete;
		LmxContext* ctx;
		uint32_t view = 0;
		bool ok = false;
	};

	bool busy() {
		CtxLock guard(m_ctx);
		m_error = "every result slot stayed reserved for 5 s: a caller never returned from cull()";
		report();
		return false;
	}

	void report() { // the reference's error convention: log and carry on (no exceptions)
#ifdef LMX_WITH_LUMIX_HEADERS
		logError("GpuCullingSystem: ", m_error.c_str());
#else
		fprintf(stderr, "GpuCullingSystem: %s\n", m_error.c_str());
#endif
	}

	void noContext() {
		m_error = lmx_last_error(nullptr);
		report();
		m_ctx = nullptr;
	}

	bool check(int rc) {
		if (rc == LMX_OK) return true;
		m_error = m_ctx ? lmx_last_error(m_ctx) : "no context";
		report();
		return false;
	}

	bool checkUnlocked(int rc) { // (the error string belongs to the context: read it under its lock)
		if (rc == LMX_OK) return true;
		CtxLock guard(m_ctx);
		return check(rc);
	}

	CullResult* newPage(u8 type) {
		CullResult* page = new (m_page_allocator.allocate()) CullResult;
		page->header.type = type;
		return page;
	}

	// [ids of type 0 | type 1 | ...] + per-type counts -> linked 4 KiB CullResult pages of 1020 ids, one type per page
	CullResult* toPages(const int32_t* ids, const uint32_t* counts, u8 type) {
		CullResult* first = nullptr;
		CullResult* last = nullptr;
		constexpr uint32_t PAGE_IDS = sizeof(CullResult::entities) / sizeof(EntityRef);
		size_t at = 0;
		for (uint32_t t = 0; t < LMX_MAX_TYPES; ++t) {
			const uint32_t got = counts[t];
			for (uint32_t i = 0; i < got; i += PAGE_IDS) {
				CullResult* page = newPage((u8)t);
				const uint32_t n = got - i < PAGE_IDS ? got - i : PAGE_IDS;
				static_assert(sizeof(EntityRef) == sizeof(int32_t), "EntityRef is its index");
				memcpy(static_cast<void*>(page->entities), ids + at + i, (size_t)n * sizeof(int32_t));
				page->header.count = n;
				if (last) last->header.next = page; else first = page;
				last = page;
			}
			at += got;
		}
		if (!first) first = newPage(type == 0xff ? 0 : type); // the reference returns >= 1 (possibly empty) page per visited cell
		return first;
	}

	CullResult* cullInternal(const ShiftedFrustum& frustum, u8 type) {
		if (!m_ctx) return nullptr;
		ViewSlot slot(m_ctx); // reserved until toPages() has copied the ids out of the slot's pinned record
		if (!slot.ok) { busy(); return nullptr; }
		const uint32_t view = slot.view;
		{ // enqueue under the context's lock (several views may be in flight concurrently, pipeline.cpp:1036-1041: they only serialise HERE)
			CtxLock guard(m_ctx);
			uint32_t n_static = 0, n_bound = 0, n_overflow = 0;
			if (!check(lmx_cull_update_stats(m_ctx, &n_static, &n_bound, &n_overflow, nullptr))) return nullptr;
			if (n_static + n_bound + n_overflow == 0) return nullptr; // no cells: culling_system.cpp:322
			static_assert(sizeof(ShiftedFrustum) == sizeof(LmxShiftedFrustum), "layout");
			if (!check(lmx_cull(m_ctx, view, reinterpret_cast<const LmxShiftedFrustum*>(&frustum), 1, type))) return nullptr;
			if (!check(lmx_cull_map_begin(m_ctx, view, 1))) return nullptr;
		}
		// one host wait per cull, on this view's own event: totals + ids arrive as one record in the library's pinned host memory and
		// are copied into the engine's pages here, outside the lock (the slot stays reserved until that copy is done)
		uint32_t counts[LMX_MAX_TYPES];
		const int32_t* ids = nullptr;
		if (!checkUnlocked(lmx_cull_map_end(m_ctx, view, 1, &ids, counts))) return nullptr;
		return toPages(ids, counts, type);
	}

	PageAllocator& m_page_allocator;
	LmxContext* m_ctx = nullptr;
	bool m_shared = false;
	std::string m_error;
};

} // namespace Lumix
