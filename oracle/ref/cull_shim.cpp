// cull_shim.cpp — TEST INFRASTRUCTURE. The reference's own CullingSystemImpl behind the ref_cs_* C entry points.
//
// renderer/culling_system.cpp is compiled IN PLACE: it is #included below from /root/reference (one translation unit, so that
// ref_cs_cell_count can read CullingSystemImpl::m_cells), and core/page_allocator.cpp (PageAllocator: lock-free ring of free pages +
// locked fallback) is compiled in place next to it by oracle/Makefile. Nothing of either file is in this repository.
// They do not compile against the tree as it lies: core/sync.h:20-24 is `#error "Not implemented"` for SRWLock on Linux, so the
// Makefile compiles against a temporary copy of src/core whose sync.h has that one line replaced by a pthread_rwlock_t member
// (the copy is deleted after the compile). Everything culling does - cell hashing into the engine's HashMap, CellPage fill / split /
// swap-remove, big-sphere cells, cullInternal's contains / intersects / doCulling per cell through jobs::forEach, the result pages
// through PagedList<CullResult> - is therefore reference object code.
//
// What IS in this file, and is mine: the pieces of the engine underneath that cannot be built here -
//   jobs::getWorkersCount / runN / wait   core/job_system.cpp is fibers + its own scheduler; here: worker_pool.h threads. The
//                                         reference's jobs::forEach template (core/job_system.h:131-180) runs on top unchanged.
//   Mutex                                  core/linux/sync.cpp does not compile (Semaphore mismatch); the four pthread calls
//   os::memReserve / memCommit / memRelease  core/linux/os.cpp needs X11; anonymous mmap / munmap like it (os.cpp:974-986)
//   profiler::beginBlock / endBlock / pushInt  no-ops
// - and the extern "C" entry points.
#include "renderer/culling_system.cpp"

#include <pthread.h>
#include <stdlib.h>
#include <sys/mman.h>

#include <functional>

#include "core/os.h"
#include "lmx_types.h"
#include "worker_pool.h"

namespace Lumix {

Mutex::Mutex() { pthread_mutex_init(&mutex, nullptr); }
Mutex::~Mutex() { pthread_mutex_destroy(&mutex); }
void Mutex::enter() { pthread_mutex_lock(&mutex); }
void Mutex::exit() { pthread_mutex_unlock(&mutex); }

namespace os {
// The reference maps every 4 KiB page on its own (os.cpp:974-978). Pages are carved from 32 MiB anonymous mappings instead: the same
// steady-state cull time here (225 ms either way at 2 M entities), but no million-entry VMA list at 10 M entities.
static pthread_mutex_t g_slab_mutex = PTHREAD_MUTEX_INITIALIZER;
static char* g_slab_cur = nullptr;
static char* g_slab_end = nullptr;
void* memReserve(size_t size) {
	const size_t SLAB = 32u << 20;
	if (size > SLAB / 2) return mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
	size = (size + 4095) & ~(size_t)4095;
	pthread_mutex_lock(&g_slab_mutex);
	if (g_slab_cur == nullptr || (size_t)(g_slab_end - g_slab_cur) < size) {
		g_slab_cur = (char*)mmap(nullptr, SLAB, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		g_slab_end = g_slab_cur + SLAB;
		madvise(g_slab_cur, SLAB, MADV_HUGEPAGE);
	}
	void* p = g_slab_cur;
	g_slab_cur += size;
	pthread_mutex_unlock(&g_slab_mutex);
	return p;
}
void memCommit(void*, size_t) {}
void memRelease(void*, size_t) {} // PageAllocator releases pages only in its destructor, which this shim never runs
} // namespace os

namespace profiler {
void beginBlock(const char*) {}
void endBlock() {}
void pushInt(const char*, int) {}
} // namespace profiler

namespace jobs {
static int g_workers = 1;                  // what getWorkersCount() reports: set per cull from the caller's n_threads
static std::function<void()> g_job;        // one job in flight at a time (culls of this shim are not concurrent)
static bool g_started = false;

u8 getWorkersCount() { return (u8)g_workers; }

void runN(void* data, void (*task)(void*), Counter*, u32 num_jobs) {
	if (num_jobs == 0) return;
	g_job = [data, task]() { task(data); };
	g_started = true;
	lmx_ref::pool().start((int)num_jobs, g_job);
}

void wait(Counter*) {
	if (!g_started) return;
	lmx_ref::pool().finish();
	g_started = false;
}
} // namespace jobs

} // namespace Lumix

using namespace Lumix;

namespace {
struct HeapAllocator final : IAllocator { // the engine's default allocator stand-in (core/default_allocator.cpp needs os.cpp)
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
HeapAllocator& heap() { static HeapAllocator& a = *new HeapAllocator; return a; }
// the engine has a single PageAllocator (engine.h:45): cell pages and result pages share it; never destroyed (its destructor
// asserts that every page came back)
PageAllocator& pages() { static PageAllocator& p = *new PageAllocator(heap()); return p; }

struct Handle { UniquePtr<CullingSystem> cs; };
CullingSystem& sys(void* h) { return *((Handle*)h)->cs.get(); }
ShiftedFrustum toRef(const LmxShiftedFrustum* f) {
	static_assert(sizeof(ShiftedFrustum) == sizeof(LmxShiftedFrustum), "ShiftedFrustum layout");
	ShiftedFrustum r;
	memcpy((void*)&r, f, sizeof(r));
	return r;
}
} // namespace

static_assert(sizeof(CellPage::spheres) / sizeof(Sphere) == LMX_CULL_PAGE_SPHERES, "CellPage slots");

extern "C" {
#define REF_API __attribute__((visibility("default")))

REF_API void* ref_cs_create(void) {
	Handle* h = new Handle;
	h->cs = CullingSystem::create(heap(), pages());
	return h;
}
REF_API void ref_cs_destroy(void* cs) { delete (Handle*)cs; }
REF_API void ref_cs_add(void* cs, int32_t entity, uint8_t type, const double* pos, float radius) {
	sys(cs).add(EntityRef{entity}, type, DVec3(pos[0], pos[1], pos[2]), radius);
}
REF_API void ref_cs_add_bulk(void* cs, uint32_t n, const int32_t* entity, const uint8_t* type, const double* pos, const float* radius) {
	for (uint32_t i = 0; i < n; ++i) ref_cs_add(cs, entity[i], type[i], pos + 3 * (size_t)i, radius[i]);
}
REF_API void ref_cs_remove(void* cs, int32_t entity) { sys(cs).remove(EntityRef{entity}); }
REF_API void ref_cs_set(void* cs, int32_t entity, const double* pos, float radius) {
	sys(cs).set(EntityRef{entity}, DVec3(pos[0], pos[1], pos[2]), radius);
}
REF_API void ref_cs_set_position(void* cs, int32_t entity, const double* pos) { sys(cs).setPosition(EntityRef{entity}, DVec3(pos[0], pos[1], pos[2])); }
REF_API void ref_cs_set_radius(void* cs, int32_t entity, float radius) { sys(cs).setRadius(EntityRef{entity}, radius); }
REF_API float ref_cs_get_radius(void* cs, int32_t entity) { return sys(cs).getRadius(EntityRef{entity}); }
REF_API int ref_cs_is_added(void* cs, int32_t entity) { return entity >= 0 && sys(cs).isAdded(EntityRef{entity}) ? 1 : 0; }
REF_API uint32_t ref_cs_cell_count(void* cs) { return (uint32_t)static_cast<CullingSystemImpl&>(sys(cs)).m_cells.size(); }

// Runs one cull and flattens the CullResult page list into (id, type) arrays. Returns the total count and the number of result
// pages the reference allocated (the page traffic is part of what the CPU path pays for). type 0xff = all types.
REF_API uint32_t ref_cs_cull(void* cs, const LmxShiftedFrustum* frustum, uint8_t type, int n_threads, int32_t* out_ids, uint8_t* out_types,
	uint32_t cap, uint32_t* out_pages) {
	jobs::g_workers = n_threads < 1 ? 1 : (n_threads > 255 ? 255 : n_threads);
	const ShiftedFrustum f = toRef(frustum);
	CullResult* first = type == 0xff ? sys(cs).cull(f) : sys(cs).cull(f, type);
	uint32_t n = 0, pages_n = 0;
	for (CullResult* page = first; page; page = page->header.next, ++pages_n) {
		for (u32 i = 0; i < page->header.count; ++i, ++n) {
			if (n < cap) {
				if (out_ids) out_ids[n] = page->entities[i].index;
				if (out_types) out_types[n] = page->header.type;
			}
		}
	}
	if (first) first->free(pages());
	if (out_pages) *out_pages = pages_n;
	return n;
}
} // extern "C"
