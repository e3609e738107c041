// real_header_harness.cpp — TEST INFRASTRUCTURE. The engine-side host code of this repository (lumixengine_amd/host/:
// GpuCullingSystem, WorldSync, mi355_plugin.cpp's ISystem / IModule pair) compiled against the reference's REAL headers
// (-DLMX_WITH_LUMIX_HEADERS) and RUN on the GPU inside the pieces of the engine that build here: a real Lumix::World
// (engine/world.cpp), a real PageAllocator (core/page_allocator.cpp), the real CullingSystem vtable (renderer/culling_system.h:58-77)
// - all reference object code from oracle/_ref/liblmx_ref.so - compared call by call with the reference's own CullingSystemImpl
// driven by a second real World through the same delegate RenderModuleImpl uses (onModelInstanceMoved, render_module.cpp:1544-1554).
//
// Built by `make -C oracle harness` where /root/reference exists (the binary lands in oracle/_ref/, git-ignored, and travels to
// the GPU box with the snapshot); run by tests/test_gpu_real_headers.py.
//
//   real_header_harness            the wiring a maintainer gets from INTEGRATION.md: createGpuCullingSystem(allocator, pages, world)
//                                  + createPlugin_mi355 -> createModules(world): ONE context per World
//   real_header_harness --two-contexts   round 2's wiring (the culling system on a context of its own): bindModelInstances must
//                                  FAIL and say so; exit code 42. The test asserts that this mode fails - it is what makes the
//                                  default mode's pass mean something.
//
// What plays the renderer: `FakeRenderer` below owns the World's culling system exactly where RenderModuleImpl does and binds the same
// per-entity delegate; RenderModuleImpl itself (DX12 back end, resources, 193 virtuals) cannot be built here.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>
#include <thread>
#include <atomic>
#include <chrono>
#include <unistd.h>

#include "shell_engine.h" // oracle/ref: an Engine that owns an allocator (mine)

#include "core/geometry.h"
#include "engine/reflection.h"
#include "renderer/culling_system.h"

// the product's host code, against the real headers
#include "../../lumixengine_amd/host/mi355_plugin.cpp"

using namespace Lumix;

// ---- link stubs the harness owns (they interpose the ones inside liblmx_ref.so) -----------------------------------------------------
namespace Lumix {
ISystem::~ISystem() = default; // engine/plugin.cpp:21 (the file drags in the dynamic library loader of core/os)
namespace reflection {
ComponentBase::ComponentBase(IAllocator& allocator) : props(allocator), functions(allocator) {}
static RegisteredComponent g_components[2];
static int g_n_components = 0;
Span<const RegisteredComponent> getComponents() { return Span<const RegisteredComponent>(g_components, (u32)g_n_components); }
ComponentType getComponentType(StringView) { return INVALID_COMPONENT_TYPE; }
} // namespace reflection
} // namespace Lumix

namespace {

const ComponentType MODEL_INSTANCE_TYPE = {0};
const ComponentType POINT_LIGHT_TYPE = {1};

struct Rng { // xorshift: the same stream drives both sides
	uint64_t s;
	explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
	uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
	double uni(double a, double b) { return a + (b - a) * (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
	u32 below(u32 n) { return (u32)(next() % n); }
};

Quat randomQuat(Rng& r) {
	float q[4];
	float len = 0;
	for (float& v : q) { v = (float)r.uni(-1, 1); len += v * v; }
	len = sqrtf(len);
	return Quat(q[0] / len, q[1] / len, q[2] / len, q[3] / len);
}

Transform randomTransform(Rng& r, double extent) {
	Transform t;
	t.pos = DVec3(r.uni(-extent, extent), r.uni(-extent, extent), r.uni(-extent, extent));
	t.rot = randomQuat(r);
	t.scale = Vec3((float)r.uni(0.5, 2.0), (float)r.uni(0.5, 2.0), (float)r.uni(0.5, 2.0));
	return t;
}

// What RenderModuleImpl does with its culling system (render_module.cpp:3569, :1544-1554, :2880-2940): owns it, adds model instances
// with radius = model radius * max scale, and refreshes the sphere from the World's `transformed` delegate.
struct FakeRenderer {
	FakeRenderer(World& world, UniquePtr<CullingSystem>&& cs) : m_world(world), m_culling_system(cs.move()) {
		m_world.componentTransformed(MODEL_INSTANCE_TYPE).bind<&FakeRenderer::onModelInstanceMoved>(this);
	}
	~FakeRenderer() { m_world.componentTransformed(MODEL_INSTANCE_TYPE).unbind<&FakeRenderer::onModelInstanceMoved>(this); }
	void addModelInstance(EntityRef e, float model_radius) {
		if ((size_t)e.index >= m_model_radius.size()) m_model_radius.resize(e.index + 1, -1.f);
		m_model_radius[e.index] = model_radius;
		m_world.onComponentCreated(e, MODEL_INSTANCE_TYPE, nullptr);
		const Transform& tr = m_world.getTransform(e);
		m_culling_system->add(e, 0, tr.pos, model_radius * maximum(tr.scale.x, tr.scale.y, tr.scale.z));
	}
	void onModelInstanceMoved(EntityRef entity) {
		if (!m_culling_system->isAdded(entity)) return;
		const Transform& tr = m_world.getTransform(entity);
		++m_moved_calls;
		m_culling_system->set(entity, tr.pos, m_model_radius[entity.index] * maximum(tr.scale.x, tr.scale.y, tr.scale.z));
	}
	World& m_world;
	UniquePtr<CullingSystem> m_culling_system;
	std::vector<float> m_model_radius;
	u64 m_moved_calls = 0;
};

struct Visible { std::vector<u64> v; u32 pages = 0; };

Visible flatten(CullResult* res, PageAllocator& pages) {
	Visible out;
	for (CullResult* p = res; p; p = p->header.next, ++out.pages) {
		if (p->header.count > sizeof(p->entities) / sizeof(p->entities[0])) { fprintf(stderr, "page with %u ids\n", p->header.count); exit(5); }
		for (u32 i = 0; i < p->header.count; ++i) out.v.push_back(((u64)p->header.type << 32) | (u32)p->entities[i].index);
	}
	if (res) res->free(pages);
	std::sort(out.v.begin(), out.v.end());
	return out;
}

int g_failures = 0;
#define CHECK(cond, ...)                                                                                                         \
	do {                                                                                                                         \
		if (!(cond)) {                                                                                                           \
			fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__);                                                                 \
			fprintf(stderr, __VA_ARGS__);                                                                                        \
			fprintf(stderr, "\n");                                                                                               \
			if (++g_failures > 20) exit(1);                                                                                      \
		}                                                                                                                        \
	} while (0)

bool sameVisible(CullingSystem& a, CullingSystem& b, const ShiftedFrustum& f, PageAllocator& pages, const char* what, int type = -1) {
	Visible va = flatten(type < 0 ? a.cull(f) : a.cull(f, (u8)type), pages);
	Visible vb = flatten(type < 0 ? b.cull(f) : b.cull(f, (u8)type), pages);
	const bool ok = va.v == vb.v;
	CHECK(ok, "%s: %zu visible vs reference %zu", what, va.v.size(), vb.v.size());
	return ok;
}

bool sameTransforms(const World& a, const World& b, u32 n, const char* what) {
	const Transform* ta = a.getTransforms();
	const Transform* tb = b.getTransforms();
	for (u32 i = 0; i < n; ++i) {
		const bool ok = memcmp(&ta[i].pos, &tb[i].pos, sizeof(DVec3)) == 0 && memcmp(&ta[i].rot, &tb[i].rot, sizeof(Quat)) == 0 &&
						memcmp(&ta[i].scale, &tb[i].scale, sizeof(Vec3)) == 0;
		if (!ok) {
			CHECK(false, "%s: entity %u world transform differs: pos (%.17g %.17g %.17g) vs (%.17g %.17g %.17g)", what, i, ta[i].pos.x, ta[i].pos.y, ta[i].pos.z,
				tb[i].pos.x, tb[i].pos.y, tb[i].pos.z);
			return false;
		}
	}
	return true;
}

std::vector<ShiftedFrustum> makeFrusta() {
	std::vector<ShiftedFrustum> out;
	const struct { DVec3 pos; Vec3 dir; float fov, far_d; } cams[] = {
		{DVec3(0, 0, 0), Vec3(0, 0, -1), 1.0472f, 10000.f},
		{DVec3(300.5, -40, 900), Vec3(0.6f, -0.1f, -0.79f), 1.3f, 6000.f},
		{DVec3(-2500, 200, -1200), Vec3(1, 0, 0), 0.7f, 4000.f},
		{DVec3(1e6, 50, -1e6), Vec3(0, 0, 1), 1.0f, 5000.f}, // far from everything: the fp64 shift
	};
	for (const auto& c : cams) {
		ShiftedFrustum f;
		f.computePerspective(c.pos, normalize(c.dir), Vec3(0, 1, 0), c.fov, 16.f / 9.f, 0.1f, c.far_d);
		out.push_back(f);
	}
	ShiftedFrustum o;
	o.computeOrtho(DVec3(100, 800, -50), normalize(Vec3(0.2f, -1.f, 0.1f)), Vec3(0, 0, 1), 600.f, 600.f, 0.f, 3000.f);
	out.push_back(o);
	ShiftedFrustum o2;
	o2.computeOrtho(DVec3(-700, 500, 300), normalize(Vec3(-0.3f, -1.f, 0.4f)), Vec3(1, 0, 0), 1500.f, 1500.f, 0.f, 5000.f);
	out.push_back(o2);
	return out;
}

// ---- part 1: the CullingSystem vtable, call by call against CullingSystemImpl --------------------------------------------------------
void testCullingSystem(IAllocator& heap, PageAllocator& pages) {
	UniquePtr<CullingSystem> ref = CullingSystem::create(heap, pages); // reference object code
	int dummy_key = 0;
	UniquePtr<CullingSystem> gpu = UniquePtr<GpuCullingSystem>::create(heap, pages, static_cast<const void*>(&dummy_key));
	CHECK(static_cast<GpuCullingSystem*>(gpu.get())->isValid(), "no device: %s", static_cast<GpuCullingSystem*>(gpu.get())->lastError().c_str());
	if (g_failures) return;
	const std::vector<ShiftedFrustum> frusta = makeFrusta();
	CHECK(gpu->cull(frusta[0]) == nullptr && ref->cull(frusta[0]) == nullptr, "an empty system returns nullptr (culling_system.cpp:322)");
	Rng rng(7);
	const u32 n = 40000;
	std::vector<bool> alive(n + 2000, false);
	auto add = [&](u32 e) {
		const DVec3 p(rng.uni(-3000, 3000), rng.uni(-3000, 3000), rng.uni(-3000, 3000));
		const float radius = rng.below(100) == 0 ? (float)rng.uni(300.0, 900.0) : (float)exp(rng.uni(log(0.5), log(50.0)));
		const u8 type = (u8)(rng.below(10) == 0 ? 2 : (rng.below(10) == 0 ? 1 : 0));
		gpu->add(EntityRef{(i32)e}, type, p, radius);
		ref->add(EntityRef{(i32)e}, type, p, radius);
		alive[e] = true;
	};
	for (u32 e = 0; e < n; ++e) add(e);
	for (size_t f = 0; f < frusta.size(); ++f) sameVisible(*gpu, *ref, frusta[f], pages, "after add");
	u32 next = n;
	for (int step = 0; step < 6000; ++step) {
		const u32 op = rng.below(7);
		u32 e = rng.below(next);
		if (op == 0 && next < alive.size()) {
			add(next++);
		} else if (!alive[e]) {
			CHECK(!gpu->isAdded(EntityRef{(i32)e}) && !ref->isAdded(EntityRef{(i32)e}), "isAdded of a removed entity");
		} else if (op == 1) {
			gpu->remove(EntityRef{(i32)e});
			ref->remove(EntityRef{(i32)e});
			alive[e] = false;
		} else {
			const DVec3 p(rng.uni(-3200, 3200), rng.uni(-3200, 3200), rng.uni(-3200, 3200));
			const float radius = rng.below(4) == 0 ? (float)rng.uni(280.0, 330.0) : (float)rng.uni(0.5, 60.0);
			if (op == 2 || op == 3) { gpu->set(EntityRef{(i32)e}, p, radius); ref->set(EntityRef{(i32)e}, p, radius); }
			else if (op == 4) { gpu->setPosition(EntityRef{(i32)e}, p); ref->setPosition(EntityRef{(i32)e}, p); }
			else { gpu->setRadius(EntityRef{(i32)e}, radius); ref->setRadius(EntityRef{(i32)e}, radius); }
			const float ra = gpu->getRadius(EntityRef{(i32)e}), rb = ref->getRadius(EntityRef{(i32)e});
			CHECK(memcmp(&ra, &rb, 4) == 0, "getRadius(%u) %g vs %g", e, ra, rb);
			CHECK(gpu->isAdded(EntityRef{(i32)e}) && ref->isAdded(EntityRef{(i32)e}), "isAdded(%u)", e);
		}
		if (step % 1000 == 999) {
			for (size_t f = 0; f < frusta.size(); ++f) sameVisible(*gpu, *ref, frusta[f], pages, "update stream");
			sameVisible(*gpu, *ref, frusta[1], pages, "type filter 0", 0);
			sameVisible(*gpu, *ref, frusta[1], pages, "type filter 2", 2);
		}
	}
	// the frame's views in one pass (cullMany) == the reference's one cull per view
	CullResult* many[LMX_MAX_FRUSTA];
	const u32 nf = (u32)frusta.size();
	CHECK(static_cast<GpuCullingSystem*>(gpu.get())->cullMany(frusta.data(), nf, 0xff, many), "cullMany failed");
	for (u32 f = 0; f < nf; ++f) {
		Visible a = flatten(many[f], pages), b = flatten(ref->cull(frusta[f]), pages);
		CHECK(a.v == b.v, "cullMany view %u: %zu vs reference %zu", f, a.v.size(), b.v.size());
	}
	printf("culling system: %u entities, 6000 interleaved calls, %zu views: identical to CullingSystemImpl\n", next, frusta.size());

	// Several views in flight concurrently (the reference's render jobs cull the views of a frame from different threads,
	// pipeline.cpp:1036-1041): six threads, one view each, 50 frames, started together. cull() serialises on the context's lock only
	// while it enqueues; the waits (each on its own view's event) and the page building overlap. Every result is the reference's, and
	// the host time of a frame's six views is reported three ways: one after the other, six threads, and ONE cullMany pass.
	{
		GpuCullingSystem* gc = static_cast<GpuCullingSystem*>(gpu.get());
		std::vector<Visible> want(nf);
		for (u32 f = 0; f < nf; ++f) want[f] = flatten(ref->cull(frusta[f]), pages);
		auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
		for (int warm = 0; warm < 5; ++warm)
			for (u32 f = 0; f < nf; ++f) flatten(gpu->cull(frusta[f]), pages);
		const int frames = 50;
		double t0 = now();
		for (int k = 0; k < frames; ++k)
			for (u32 f = 0; f < nf; ++f) { CullResult* r = gpu->cull(frusta[f]); if (r) r->free(pages); }
		const double ms_serial = (now() - t0) / frames;
		std::atomic<int> bad{0}, arrived{0};
		std::atomic<int> go{0};
		std::vector<std::thread> threads;
		std::vector<double> t_thread(nf, 0.0);
		for (u32 f = 0; f < nf; ++f)
			threads.emplace_back([&, f] {
				for (int k = 0; k < frames; ++k) {
					arrived.fetch_add(1);
					while (go.load(std::memory_order_acquire) <= k) std::this_thread::yield(); // every frame's six culls start together
					CullResult* r = gpu->cull(frusta[f]);
					if (k % 10 == 0) { // (checking every frame would time the check)
						Visible got = flatten(r, pages);
						if (got.v != want[f].v) bad.fetch_add(1);
					} else if (r) {
						r->free(pages);
					}
				}
			});
		double t_conc = 0;
		for (int k = 0; k < frames; ++k) {
			while (arrived.load() < (int)nf * (k + 1)) std::this_thread::yield();
			const double a = now();
			go.store(k + 1, std::memory_order_release);
			while (arrived.load() < (int)nf * (k + 2) && k + 1 < frames) std::this_thread::yield(); // all six are back at the gate = the frame's culls are done
			if (k + 1 == frames) break;
			t_conc += now() - a;
		}
		for (std::thread& t : threads) t.join();
		const double ms_conc = t_conc / (frames - 1);
		CHECK(bad.load() == 0, "%d concurrent culls differ from the reference's", bad.load());
		for (int warm = 0; warm < LMX_MAX_VIEWS; ++warm) { // (every result slot has held an nf-wide batch once: their buffers exist)
			CHECK(gc->cullMany(frusta.data(), nf, 0xff, many), "cullMany failed");
			for (u32 f = 0; f < nf; ++f) if (many[f]) many[f]->free(pages);
		}
		t0 = now();
		for (int k = 0; k < frames; ++k) {
			CHECK(gc->cullMany(frusta.data(), nf, 0xff, many), "cullMany failed");
			for (u32 f = 0; f < nf; ++f) if (many[f]) many[f]->free(pages);
		}
		const double ms_many = (now() - t0) / frames;
		printf("culling system: %u views of a frame, host time incl. CullResult pages: %.3f ms one after the other, %.3f ms from %u concurrent threads, %.3f ms as ONE cullMany pass; "
			   "every concurrent result identical to CullingSystemImpl\n", nf, ms_serial, ms_conc, nf, ms_many);
	}

	// More callers in flight than result slots (culling_system.cpp:321-369: every cull() returns an independent list; callers beyond the
	// frame's six views: pipeline.cpp:3380, editor/scene_view.cpp:144). Twelve threads cull without any gate between them, every one
	// of their lists is compared with CullingSystemImpl's: a slot that were culled into again while its previous holder still copies
	// its ids out of the pinned record (round 4's round-robin slots did exactly that with a ninth caller) shows up as a torn list here.
	// Half of the callers dawdle between the enqueue's result and its release by reading the list twice.
	{
		std::vector<Visible> want(nf);
		for (u32 f = 0; f < nf; ++f) want[f] = flatten(ref->cull(frusta[f]), pages);
		const u32 callers = 12;
		std::atomic<int> bad{0}, failed{0};
		std::vector<std::thread> threads;
		for (u32 c = 0; c < callers; ++c)
			threads.emplace_back([&, c] {
				for (int k = 0; k < 40; ++k) {
					const u32 f = (c + (u32)k) % nf;
					CullResult* r = gpu->cull(frusta[f]);
					if (!r) { failed.fetch_add(1); continue; }
					Visible got = flatten(r, pages);
					if (got.v != want[f].v) bad.fetch_add(1);
				}
			});
		for (std::thread& t : threads) t.join();
		CHECK(failed.load() == 0, "%d of the 12 callers' culls returned nothing", failed.load());
		CHECK(bad.load() == 0, "%d culls of 12 concurrent callers differ from the reference's (a result slot was reused while it was being read)", bad.load());
		// the reservation itself: with every slot held, a further acquire times out with LMX_ERR_BUSY instead of handing out a held slot
		LmxContext* c = static_cast<GpuCullingSystem*>(gpu.get())->context();
		uint32_t held[LMX_MAX_VIEWS], extra = 99;
		uint32_t seen = 0;
		for (int k = 0; k < LMX_MAX_VIEWS; ++k) {
			CHECK(lmx_cull_view_acquire(c, &held[k], 100) == LMX_OK, "acquire %d failed", k);
			seen |= 1u << held[k];
		}
		CHECK(seen == (1u << LMX_MAX_VIEWS) - 1u, "the %d acquired slots are not distinct: mask %x", (int)LMX_MAX_VIEWS, seen);
		CHECK(lmx_cull_view_acquire(c, &extra, 50) == LMX_ERR_BUSY, "a ninth acquire succeeded with all slots held");
		std::thread late([&] { std::this_thread::sleep_for(std::chrono::milliseconds(30)); lmx_cull_view_release(c, held[3]); });
		CHECK(lmx_cull_view_acquire(c, &extra, 2000) == LMX_OK && extra == held[3], "a waiter was not woken by the release (got slot %u)", extra);
		late.join();
		CHECK(lmx_cull_view_release(c, extra) == LMX_OK, "release of the re-acquired slot failed");
		for (int k = 0; k < LMX_MAX_VIEWS; ++k)
			if (k != 3) CHECK(lmx_cull_view_release(c, held[k]) == LMX_OK, "release %d failed", k);
		CHECK(lmx_cull_view_release(c, held[0]) != LMX_OK, "releasing a free slot was accepted");
		printf("culling system: 12 concurrent callers x 40 culls on %d result slots: every list identical to CullingSystemImpl; a ninth reservation waits, times out with LMX_ERR_BUSY, is woken by a release\n", (int)LMX_MAX_VIEWS);
	}

	// what createGpuCullingSystem() switches on for the engine: the re-sort of the sorted set on a worker thread. 80 000 more entities
	// push the overflow past the compaction threshold; updates and culls go on while the worker re-sorts, and after the sets have traded
	// places every view is still the reference's.
	GpuCullingSystem* g = static_cast<GpuCullingSystem*>(gpu.get());
	CHECK(g->setAsyncCompaction(true), "setAsyncCompaction: %s", g->lastError().c_str());
	alive.resize(next + 80000 + 16, false);
	for (u32 k = 0; k < 80000; ++k) add(next++);
	uint64_t swaps = 0;
	int state = 0;
	for (int frame = 0; frame < 4000 && swaps == 0; ++frame) {
		for (int k = 0; k < 20; ++k) { // the update stream does not stop for the worker
			const u32 e = rng.below(next);
			if (!alive[e]) continue;
			const DVec3 p(rng.uni(-3200, 3200), rng.uni(-3200, 3200), rng.uni(-3200, 3200));
			const float radius = (float)rng.uni(0.5, 60.0);
			if (k % 5 == 0) { gpu->remove(EntityRef{(i32)e}); ref->remove(EntityRef{(i32)e}); alive[e] = false; }
			else { gpu->set(EntityRef{(i32)e}, p, radius); ref->set(EntityRef{(i32)e}, p, radius); }
		}
		if (frame % 50 == 0) sameVisible(*gpu, *ref, frusta[frame / 50 % frusta.size()], pages, "while the worker re-sorts");
		else if (CullResult* r = gpu->cull(frusta[0])) r->free(pages);
		CHECK(lmx_cull_async_stats(g->context(), &state, nullptr, &swaps, nullptr, nullptr) == LMX_OK && state != 4, "asynchronous compaction failed");
		usleep(1000);
	}
	CHECK(swaps >= 1, "the sets never traded places (state %d)", state);
	for (size_t f = 0; f < frusta.size(); ++f) sameVisible(*gpu, *ref, frusta[f], pages, "after the asynchronous compaction");
	printf("culling system: asynchronous compaction of %u entities under an update stream: identical to CullingSystemImpl\n", next);
}

// ---- part 2: World + plugin module + culling system of one World ----------------------------------------------------------------------
int testModule(bool two_contexts, IAllocator& heap, PageAllocator& pages) {
	lmx_ref::ShellEngine engine_gpu, engine_ref;
	World world(engine_gpu), ref_world(engine_ref);
	Rng rng(11);
	// roots + chains of depth <= 4; every other entity carries a model instance, a few a "light" (a second component type)
	const u32 n_roots = 3000, n = 9000;
	std::vector<i32> parent(n, -1);
	for (u32 e = 0; e < n; ++e) {
		const EntityRef a = world.createEntity(DVec3(0), Quat::IDENTITY), b = ref_world.createEntity(DVec3(0), Quat::IDENTITY);
		if (a.index != (i32)e || b.index != (i32)e) return 3;
	}
	for (u32 e = 0; e < n; ++e) {
		const Transform t = randomTransform(rng, e < n_roots ? 2500.0 : 20.0);
		if (e >= n_roots) {
			parent[e] = (i32)(e < 2 * n_roots ? rng.below(n_roots) : n_roots + rng.below(e - n_roots)); // parents precede children: depth grows
			i32 depth = 0;
			for (i32 p = parent[e]; p >= 0; p = parent[p]) ++depth;
			if (depth > 3) parent[e] = (i32)rng.below(n_roots);
			world.setParent(EntityPtr{parent[e]}, EntityRef{(i32)e});
			ref_world.setParent(EntityPtr{parent[e]}, EntityRef{(i32)e});
			world.setLocalTransform(EntityRef{(i32)e}, t);
			ref_world.setLocalTransform(EntityRef{(i32)e}, t);
		} else {
			world.setTransform(EntityRef{(i32)e}, t);
			ref_world.setTransform(EntityRef{(i32)e}, t);
		}
	}
	// the renderer of each World: GPU-backed (ours) and CullingSystemImpl (the reference's)
	int other_key = 0;
	FakeRenderer renderer(world, two_contexts ? UniquePtr<CullingSystem>(UniquePtr<GpuCullingSystem>::create(heap, pages, static_cast<const void*>(&other_key)))
											  : createGpuCullingSystem(heap, pages, world));
	FakeRenderer ref_renderer(ref_world, CullingSystem::create(heap, pages));
	std::vector<EntityRef> instances;
	std::vector<float> radii;
	for (u32 e = 0; e < n; e += 2) {
		const float r = (float)rng.uni(0.5, 40.0);
		renderer.addModelInstance(EntityRef{(i32)e}, r);
		ref_renderer.addModelInstance(EntityRef{(i32)e}, r);
		instances.push_back(EntityRef{(i32)e});
		radii.push_back(r);
	}
	for (u32 e = 1; e < n; e += 50) {
		world.onComponentCreated(EntityRef{(i32)e}, POINT_LIGHT_TYPE, nullptr);
		ref_world.onComponentCreated(EntityRef{(i32)e}, POINT_LIGHT_TYPE, nullptr);
	}
	u64 light_moves = 0, ref_light_moves = 0;
	struct Counter { u64* n; void hit(EntityRef) { ++*n; } } c_gpu{&light_moves}, c_ref{&ref_light_moves};
	world.componentTransformed(POINT_LIGHT_TYPE).bind<&Counter::hit>(&c_gpu);
	ref_world.componentTransformed(POINT_LIGHT_TYPE).bind<&Counter::hit>(&c_ref);

	// the plugin, the way SystemManager loads it: createPlugin_<name>(engine) -> ISystem -> createModules(world) -> IModule::init
	ISystem* system = createPlugin_mi355(engine_gpu);
	if (!system) return 4;
	system->createModules(world);
	Mi355Module* module = static_cast<Mi355Module*>(world.getModule("mi355_hot_path"));
	if (!module || !module->context()) { fprintf(stderr, "module missing / no device\n"); return 4; }
	module->init();
	module->update(0.016f); // mirrors the World (no RenderModule in this World: the harness hands the model instances over itself)
	const bool bound = module->bindModelInstances(instances.data(), radii.data(), (u32)instances.size());
	if (two_contexts) {
		printf("two contexts: bindModelInstances %s: %s\n", bound ? "SUCCEEDED (unexpected)" : "failed as it must", module->lastError());
		return bound ? 1 : 42;
	}
	CHECK(bound, "bindModelInstances: %s", module->lastError());
	if (g_failures) return 1;

	const std::vector<ShiftedFrustum> frusta = makeFrusta();
	sameTransforms(world, ref_world, n, "after build");
	for (size_t f = 0; f < frusta.size(); ++f) sameVisible(*renderer.m_culling_system, *ref_renderer.m_culling_system, frusta[f], pages, "after bind");

	std::vector<i32> depth(n, 0);
	for (u32 e = 0; e < n; ++e)
		for (i32 p = parent[e]; p >= 0; p = parent[p]) ++depth[e];
	u64 staged = 0, moved_calls_checked = 0;
	struct Write { u32 e; bool world_space; Transform t; };
	for (int frame = 0; frame < 6; ++frame) {
		// Game code moves entities through the module (one batch per frame) / the reference World (eager DFS per call). A batch has the
		// semantics "the frame's writes applied in hierarchy order, the last write of an entity wins": the reference side applies them so.
		// Frames 0..3 write entities of ONE depth (disjoint subtrees: the delegate counts must then agree exactly); frames 4, 5 mix depths
		// and add direct World writes by engine code that does not know the module.
		std::vector<Write> writes, direct;
		std::vector<i32> at(n, -1);
		std::vector<bool> is_direct(n, false);
		if (frame >= 4) {
			for (int k = 0; k < 40; ++k) {
				const u32 e = rng.below(n_roots) & ~1u; // roots with a model instance
				if (is_direct[e]) continue;
				is_direct[e] = true;
				direct.push_back(Write{e, true, randomTransform(rng, 2500.0)});
			}
		}
		for (int k = 0; k < 700; ++k) {
			u32 e = rng.below(n);
			if (frame < 4) {
				for (int tries = 0; depth[e] != frame % 4 && tries < 64; ++tries) e = rng.below(n);
				if (depth[e] != frame % 4) continue;
			}
			if (is_direct[e]) continue; // written directly this frame
			Write w{e, parent[e] < 0 || rng.below(3) == 0, randomTransform(rng, parent[e] < 0 ? 2500.0 : 25.0)};
			if (at[e] >= 0) writes[at[e]] = w;
			else { at[e] = (i32)writes.size(); writes.push_back(w); }
		}
		std::stable_sort(writes.begin(), writes.end(), [&](const Write& a, const Write& b) { return depth[a.e] < depth[b.e]; });
		// reference: ALL of the frame's writes in hierarchy order - the direct ones are roots, so they come first
		for (const Write& w : direct) ref_world.setTransform(EntityRef{(i32)w.e}, w.t);
		for (const Write& w : writes) {
			if (w.world_space) {
				module->setTransform(EntityRef{(i32)w.e}, w.t);
				ref_world.setTransform(EntityRef{(i32)w.e}, w.t);
			} else {
				module->setLocalTransform(EntityRef{(i32)w.e}, w.t);
				ref_world.setLocalTransform(EntityRef{(i32)w.e}, w.t);
			}
			++staged;
		}
		// engine code that does not know the module writes the World directly, at any point of the frame (here: after the module's
		// writes were staged): the World runs its own DFS now, the mirror takes the written roots at the next propagation
		for (const Write& w : direct) world.setTransform(EntityRef{(i32)w.e}, w.t);
		module->update(0.016f);
		char what[64];
		snprintf(what, sizeof(what), "frame %d", frame);
		if (!sameTransforms(world, ref_world, n, what)) break;
		bool ok = true;
		for (size_t f = 0; f < frusta.size() && ok; ++f) ok = sameVisible(*renderer.m_culling_system, *ref_renderer.m_culling_system, frusta[f], pages, what);
		if (!ok) break;
		if (frame == 3) {
			// every entity the reference's DFS visited got its delegates exactly once on both sides
			CHECK(light_moves == ref_light_moves && light_moves > 0, "`transformed` delegate of the second component type fired %llu times vs reference %llu",
				(unsigned long long)light_moves, (unsigned long long)ref_light_moves);
			CHECK(renderer.m_moved_calls == ref_renderer.m_moved_calls, "onModelInstanceMoved fired %llu times vs reference %llu", (unsigned long long)renderer.m_moved_calls,
				(unsigned long long)ref_renderer.m_moved_calls);
			moved_calls_checked = renderer.m_moved_calls;
		}
	}
	CHECK(module->isBound(), "module lost its binding: %s", module->lastError());
	printf("module: %u entities, %llu staged writes + direct World writes over 6 frames: transforms bit-identical, visible sets identical, %llu onModelInstanceMoved calls each in frames 0-3\n", n,
		(unsigned long long)staged, (unsigned long long)moved_calls_checked);
	world.componentTransformed(POINT_LIGHT_TYPE).unbind<&Counter::hit>(&c_gpu);
	ref_world.componentTransformed(POINT_LIGHT_TYPE).unbind<&Counter::hit>(&c_ref);
	return 0;
}

} // namespace

int main(int argc, char** argv) {
	const bool two_contexts = argc > 1 && strcmp(argv[1], "--two-contexts") == 0;
	static lmx_ref::HeapAllocator heap;
	static PageAllocator& pages = *new PageAllocator(heap); // never destroyed: its destructor asserts that every page came back
	// what the engine's reflection registry would hold for the two component types of this test
	static reflection::ComponentBase cmp_model(heap), cmp_light(heap);
	cmp_model.component_type = MODEL_INSTANCE_TYPE;
	cmp_light.component_type = POINT_LIGHT_TYPE;
	reflection::g_components[0] = reflection::RegisteredComponent{RuntimeHash("model_instance"), RuntimeHash("renderer"), &cmp_model};
	reflection::g_components[1] = reflection::RegisteredComponent{RuntimeHash("point_light"), RuntimeHash("renderer"), &cmp_light};
	reflection::g_n_components = 2;
	if (!two_contexts) testCullingSystem(heap, pages);
	if (g_failures) return 1;
	const int rc = testModule(two_contexts, heap, pages);
	if (rc) return rc;
	if (g_failures) return 1;
	printf("real-header harness OK\n");
	return 0;
}
