// lmx_capi.hip — the C ABI of liblumix_mi355.so (include/lumix_mi355.h): context, HBM residency, host mirrors of the
// engine-side bookkeeping (CullingSystem cell assignment, World hierarchy order, Model bone tables) and kernel launches.
//
// There is no CPU fallback: every compute entry point launches gfx950 kernels and reports LMX_ERR_NO_DEVICE /
// LMX_ERR_HIP when it cannot.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lumix_mi355.h"
#include "lmx_kernels.h"
#include "lmx_cull_layout.h"

using namespace lmx;

namespace {

thread_local std::string g_create_error;

template <typename T> struct DevBuf {
	T* p = nullptr;
	size_t cap = 0; // elements
	~DevBuf() { release(); }
	void release() {
		if (p) (void)hipFree(p);
		p = nullptr;
		cap = 0;
	}
	// grow-only; contents are NOT preserved
	hipError_t reserve(size_t n) {
		if (n <= cap) return hipSuccess;
		release();
		const size_t want = n + n / 8 + 64;
		hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
		if (e != hipSuccess) {
			p = nullptr;
			return e;
		}
		cap = want;
		return hipSuccess;
	}
};

struct CullView {
	DevBuf<float4> cellinfo;
	DevBuf<int32_t> out;
	DevBuf<uint32_t> counts;
	uint32_t n_frusta = 0;
	uint32_t out_stride = 0;
	uint32_t cell_stride = 0;
	uint32_t ent_start[MAX_TYPES] = {};
	uint32_t ent_cap[MAX_TYPES] = {};
	bool valid = false;
	// caller-owned result buffers (lmx_cull_bind_output)
	int32_t* ext_out = nullptr;
	size_t ext_out_cap = 0;
	uint32_t* ext_counts = nullptr;
	// library-owned counters are double-buffered: the fused kernel clears the half the NEXT cull will use
	uint32_t flip = 0;
	bool next_half_is_zero = false;
	int32_t* out_ptr() const { return ext_out ? ext_out : out.p; }
	uint32_t* counts_ptr() const { return ext_counts ? ext_counts : counts.p + flip * (MAX_FRUSTA * MAX_TYPES); }
	uint32_t* counts_other() const { return counts.p + (flip ^ 1u) * (MAX_FRUSTA * MAX_TYPES); }
};

struct CullState {
	std::vector<CullRec> recs;
	std::vector<int32_t> ent_to_rec;
	std::vector<uint32_t> rec_slot; // rec -> device sphere slot, valid while !structure_dirty
	bool structure_dirty = false;
	bool built = false;
	bool mirror_stale = false; // device spheres are newer than recs (in-place refresh by lmx_world_propagate)
	std::vector<uint32_t> patch_slot;
	std::vector<float4> patch_val;
	DevBuf<uint32_t> d_patch_slot;
	DevBuf<float4> d_patch_val;

	DevBuf<float4> spheres;
	DevBuf<int32_t> ids;
	DevBuf<uint32_t> chunk_cell;
	DevBuf<uint64_t> chunk_flags;
	DevBuf<CellKey> cells;
	DevBuf<CellKey> tile_cells[3];
	DevBuf<uint32_t> tile_tab[3];
	uint32_t tile_cap[3] = {16, 16, 16};
	DevBuf<uint32_t> sphere_cell; // per-slot cell index, only filled when a world binding needs it
	std::vector<uint32_t> h_sphere_cell;
	uint32_t n_padded = 0, n_cells = 0;
	uint32_t max_tile_cells[3] = {0, 0, 0};
	uint32_t n_dead_cells = 0;
	TypeTable tt = {};
	uint32_t cell_begin[MAX_TYPES] = {}, cell_end[MAX_TYPES] = {};
	uint64_t generation = 0;
	CullView views[LMX_MAX_VIEWS];
};

struct WorldState {
	uint32_t n = 0;
	bool built = false;
	std::vector<int32_t> slot_of_entity, entity_of_slot, parent_slot;
	std::vector<uint32_t> level_start; // size levels + 1
	DevBuf<double> pos[6];             // lpx lpy lpz wpx wpy wpz
	DevBuf<float4> rot[2];             // lrot wrot
	DevBuf<float> scl[6];              // lsx lsy lsz wsx wsy wsz
	DevBuf<int32_t> d_parent_slot, d_slot_of_entity, d_entity_of_slot;
	DevBuf<int32_t> d_stage_entity;
	DevBuf<LmxTransform> d_stage_tr;
	DevBuf<LmxTransform> d_export;
	// culling binding
	std::vector<int32_t> bound_entity;
	std::vector<float> bound_radius;
	DevBuf<uint32_t> d_bound_slot, d_bound_sphere;
	DevBuf<float> d_bound_radius;
	DevBuf<uint32_t> d_rebin_count;
	DevBuf<RebinItem> d_rebin;
	uint64_t bound_generation = ~0ull; // cull generation the device binding tables were built for
	WorldDevice dev() {
		WorldDevice w;
		w.lpx = pos[0].p; w.lpy = pos[1].p; w.lpz = pos[2].p; w.lrot = rot[0].p; w.lsx = scl[0].p; w.lsy = scl[1].p; w.lsz = scl[2].p;
		w.wpx = pos[3].p; w.wpy = pos[4].p; w.wpz = pos[5].p; w.wrot = rot[1].p; w.wsx = scl[3].p; w.wsy = scl[4].p; w.wsz = scl[5].p;
		w.parent_slot = d_parent_slot.p;
		return w;
	}
};

struct SkinModel { uint32_t bone_offset, n_bones, max_depth; int32_t first_nonroot; };
struct SkinMesh { uint32_t vert_offset, n_verts; };

struct SkinState {
	std::vector<SkinModel> models;
	std::vector<SkinMesh> meshes;
	// concatenated host copies (re-uploaded when models/meshes are added)
	std::vector<int16_t> parents;
	std::vector<uint8_t> depth;
	std::vector<float> inv_pos;
	std::vector<float4> inv_rot;
	std::vector<float> verts;
	std::vector<float4> weights;
	std::vector<int16_t> indices;
	bool models_dirty = false, meshes_dirty = false;
	DevBuf<int16_t> d_parents;
	DevBuf<uint8_t> d_depth;
	DevBuf<float> d_inv_pos;
	DevBuf<float4> d_inv_rot;
	DevBuf<float> d_verts;
	DevBuf<float4> d_weights;
	DevBuf<int16_t> d_indices;
	std::vector<SkinInstance> inst;
	DevBuf<SkinInstance> d_inst;
	DevBuf<float> d_pose_pos;
	DevBuf<float4> d_pose_rot;
	DevBuf<float4> d_palette;
	DevBuf<float> d_out;
	size_t bones_total = 0, verts_total = 0;
	uint32_t max_verts = 0;
	bool poses_uploaded = false;
};

struct ProfSlot { hipEvent_t a, b; int kernel; };

} // namespace

struct LmxContext {
	int device = 0;
	hipStream_t own_stream = nullptr;
	hipStream_t stream = nullptr;
	std::string error;
	bool profiling = false;
	std::vector<ProfSlot> prof_pending;
	std::vector<hipEvent_t> event_pool;
	double prof_ms[LMX_K_COUNT] = {};
	uint64_t prof_launches[LMX_K_COUNT] = {};
	CullState cull;
	WorldState world;
	SkinState skin;
};

namespace {

int fail(LmxContext* ctx, int code, const char* fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	if (ctx) ctx->error = buf;
	else g_create_error = buf;
	return code;
}

#define LMX_HIP(ctx, expr)                                                                                             \
	do {                                                                                                               \
		hipError_t e_ = (expr);                                                                                        \
		if (e_ != hipSuccess)                                                                                          \
			return fail(ctx, e_ == hipErrorOutOfMemory ? LMX_ERR_OUT_OF_MEMORY : LMX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
				hipGetErrorString(e_), __FILE__, __LINE__);                                                           \
	} while (0)

#define LMX_CHECK_CTX(ctx)                                                                                             \
	do {                                                                                                               \
		if (!(ctx)) return LMX_ERR_INVALID_ARGUMENT;                                                                   \
		hipError_t e_ = hipSetDevice((ctx)->device);                                                                   \
		if (e_ != hipSuccess) return fail(ctx, LMX_ERR_NO_DEVICE, "hipSetDevice(%d): %s", (ctx)->device, hipGetErrorString(e_)); \
	} while (0)

struct ProfScope { // records HIP events around one launch on the launch stream when profiling is enabled
	LmxContext* ctx;
	ProfSlot slot;
	bool on;
	ProfScope(LmxContext* c, int kernel) : ctx(c), on(c->profiling) {
		if (!on) return;
		slot.kernel = kernel;
		slot.a = take();
		slot.b = take();
		(void)hipEventRecord(slot.a, ctx->stream);
	}
	~ProfScope() {
		if (!on) return;
		(void)hipEventRecord(slot.b, ctx->stream);
		ctx->prof_pending.push_back(slot);
	}
	hipEvent_t take() {
		if (!ctx->event_pool.empty()) {
			hipEvent_t e = ctx->event_pool.back();
			ctx->event_pool.pop_back();
			return e;
		}
		hipEvent_t e = nullptr;
		(void)hipEventCreate(&e);
		return e;
	}
};

void prof_drain(LmxContext* ctx) {
	for (ProfSlot& s : ctx->prof_pending) {
		float ms = 0.f;
		if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
			ctx->prof_ms[s.kernel] += ms;
			ctx->prof_launches[s.kernel] += 1;
		}
		ctx->event_pool.push_back(s.a);
		ctx->event_pool.push_back(s.b);
	}
	ctx->prof_pending.clear();
}

// ---- culling host mirror ---------------------------------------------------------------------------------

void cull_mark_patch(CullState& cs, uint32_t rec) {
	if (cs.structure_dirty || !cs.built) return;
	const CullRec& r = cs.recs[rec];
	cs.patch_slot.push_back(cs.rec_slot[rec]);
	cs.patch_val.push_back(make_float4(r.rel.x, r.rel.y, r.rel.z, r.radius));
}

int cull_find(LmxContext* ctx, int32_t entity, uint32_t* rec) {
	CullState& cs = ctx->cull;
	if (entity < 0 || (size_t)entity >= cs.ent_to_rec.size() || cs.ent_to_rec[entity] < 0)
		return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d is not in the culling system", entity);
	*rec = (uint32_t)cs.ent_to_rec[entity];
	return LMX_OK;
}

int cull_apply_patches(LmxContext* ctx) {
	CullState& cs = ctx->cull;
	if (cs.patch_slot.empty()) return LMX_OK;
	const size_t n = cs.patch_slot.size();
	LMX_HIP(ctx, cs.d_patch_slot.reserve(n));
	LMX_HIP(ctx, cs.d_patch_val.reserve(n));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	LMX_HIP(ctx, hipMemcpy(cs.d_patch_slot.p, cs.patch_slot.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
	LMX_HIP(ctx, hipMemcpy(cs.d_patch_val.p, cs.patch_val.data(), n * sizeof(float4), hipMemcpyHostToDevice));
	LMX_HIP(ctx, launch_patch_spheres(ctx->stream, cs.spheres.p, cs.d_patch_slot.p, cs.d_patch_val.p, (uint32_t)n));
	cs.patch_slot.clear();
	cs.patch_val.clear();
	return LMX_OK;
}

// recs <- device spheres. Called before the first host-side read or mutation that follows an in-place device refresh;
// at that point the mirror and the device layout still describe the same set of spheres (structure_dirty == false).
int cull_sync_mirror(LmxContext* ctx) {
	CullState& cs = ctx->cull;
	if (!cs.mirror_stale) return LMX_OK;
	cs.mirror_stale = false;
	if (!cs.built || cs.structure_dirty || !cs.n_padded) return LMX_OK;
	if (int rc = cull_apply_patches(ctx)) return rc;
	std::vector<float4> all(cs.n_padded);
	LMX_HIP(ctx, hipMemcpyAsync(all.data(), cs.spheres.p, (size_t)cs.n_padded * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	for (size_t r = 0; r < cs.recs.size(); ++r) {
		const float4 s = all[cs.rec_slot[r]];
		cs.recs[r].rel = V3{s.x, s.y, s.z};
		cs.recs[r].radius = s.w;
	}
	return LMX_OK;
}

static_assert(sizeof(LayoutSphere) == sizeof(float4) && sizeof(LayoutCell) == sizeof(CellKey), "layout PODs mirror the device types");
static_assert(LAYOUT_MAX_TYPES == MAX_TYPES && LAYOUT_CHUNK == CHUNK && LAYOUT_TILE_ALIGN == TILE_ALIGN && LAYOUT_CELL_DEAD == CELL_DEAD, "layout constants");

// Rebuild the device layout from the host mirror (lmx_cull_layout.h) and upload it.
int cull_rebuild(LmxContext* ctx) {
	CullState& cs = ctx->cull;
	CullLayout lay;
	if (!build_cull_layout(cs.recs, lay)) return fail(ctx, LMX_ERR_CAPACITY, "too many spheres (%zu)", cs.recs.size());
	const size_t n_padded = lay.n_padded;
	const size_t n_chunks = n_padded / CHUNK;
	for (int t = 0; t < MAX_TYPES; ++t) {
		cs.tt.ent_start[t] = lay.ent_start[t];
		cs.tt.ent_end[t] = lay.ent_end[t];
		cs.cell_begin[t] = lay.cell_begin[t];
		cs.cell_end[t] = lay.cell_end[t];
	}
	cs.n_padded = (uint32_t)n_padded;
	cs.n_cells = (uint32_t)lay.cells.size();
	for (int k = 0; k < 3; ++k) cs.max_tile_cells[k] = lay.max_tile_cells[k];
	cs.n_dead_cells = lay.n_dead_cells;
	LMX_HIP(ctx, cs.spheres.reserve(std::max<size_t>(n_padded, 1)));
	LMX_HIP(ctx, cs.ids.reserve(std::max<size_t>(n_padded, 1)));
	LMX_HIP(ctx, cs.chunk_cell.reserve(std::max<size_t>(n_chunks, 1)));
	LMX_HIP(ctx, cs.chunk_flags.reserve(std::max<size_t>(n_chunks, 1)));
	LMX_HIP(ctx, cs.cells.reserve(std::max<size_t>(cs.n_cells, 1)));
	if (n_padded) {
		// synchronous copies from pageable memory: the layout dies at the end of this function
		LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		LMX_HIP(ctx, hipMemcpy(cs.spheres.p, lay.spheres.data(), n_padded * sizeof(float4), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(cs.ids.p, lay.ids.data(), n_padded * sizeof(int32_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(cs.chunk_cell.p, lay.chunk_cell.data(), n_chunks * sizeof(uint32_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(cs.chunk_flags.p, lay.chunk_flags.data(), n_chunks * sizeof(uint64_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(cs.cells.p, lay.cells.data(), cs.n_cells * sizeof(CellKey), hipMemcpyHostToDevice));
	}
	for (int k = 0; k < 3; ++k) {
		cs.tile_cap[k] = lay.tile_cap[k];
		LMX_HIP(ctx, cs.tile_cells[k].reserve(std::max<size_t>(lay.tile_cells[k].size(), 1)));
		LMX_HIP(ctx, cs.tile_tab[k].reserve(std::max<size_t>(lay.tile_tab[k].size(), 1)));
		if (!lay.tile_cells[k].empty()) {
			LMX_HIP(ctx, hipMemcpy(cs.tile_cells[k].p, lay.tile_cells[k].data(), lay.tile_cells[k].size() * sizeof(CellKey), hipMemcpyHostToDevice));
			LMX_HIP(ctx, hipMemcpy(cs.tile_tab[k].p, lay.tile_tab[k].data(), lay.tile_tab[k].size() * sizeof(uint32_t), hipMemcpyHostToDevice));
		}
	}
	cs.rec_slot.swap(lay.rec_slot);
	cs.h_sphere_cell.swap(lay.slot_cell);
	cs.structure_dirty = false;
	cs.built = true;
	cs.patch_slot.clear();
	cs.patch_val.clear();
	cs.generation++;
	for (CullView& v : cs.views) v.valid = false;
	return LMX_OK;
}

int cull_flush(LmxContext* ctx) {
	CullState& cs = ctx->cull;
	if (cs.structure_dirty || !cs.built) return cull_rebuild(ctx);
	if (int rc = cull_apply_patches(ctx)) return rc;
	return LMX_OK;
}

void cull_readd(CullState& cs, uint32_t rec, DV3 pos, float radius) { // remove(entity); add(entity, type, pos, radius)
	const CullRec old = cs.recs[rec];
	cs.recs[rec] = make_cull_rec(old.entity, old.type, pos, radius);
	cs.structure_dirty = true;
}

CullDeviceView cull_dev(const CullState& cs) {
	CullDeviceView v;
	v.spheres = cs.spheres.p;
	v.ids = cs.ids.p;
	v.chunk_cell = cs.chunk_cell.p;
	v.chunk_flags = cs.chunk_flags.p;
	v.cells = cs.cells.p;
	v.n_padded = cs.n_padded;
	v.n_cells = cs.n_cells;
	for (int k = 0; k < 3; ++k) {
		v.tile_cells[k] = cs.tile_cells[k].p;
		v.tile_tab[k] = cs.tile_tab[k].p;
		v.tile_cap[k] = cs.tile_cap[k];
	}
	return v;
}

} // namespace

// ==========================================================================================================
// context
// ==========================================================================================================
extern "C" {

int lmx_ctx_create(int device, LmxContext** out) {
	if (!out) return LMX_ERR_INVALID_ARGUMENT;
	*out = nullptr;
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0)
		return fail(nullptr, LMX_ERR_NO_DEVICE, "no HIP device available (%s); liblumix_mi355 has no CPU fallback",
			e == hipSuccess ? "device count 0" : hipGetErrorString(e));
	if (device < 0 || device >= count) return fail(nullptr, LMX_ERR_INVALID_ARGUMENT, "device %d out of range [0,%d)", device, count);
	e = hipSetDevice(device);
	if (e != hipSuccess) return fail(nullptr, LMX_ERR_NO_DEVICE, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
	hipDeviceProp_t prop;
	e = hipGetDeviceProperties(&prop, device);
	if (e != hipSuccess) return fail(nullptr, LMX_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(nullptr, LMX_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code objects only", device, prop.gcnArchName);
	LmxContext* ctx = new LmxContext;
	ctx->device = device;
	e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
	if (e != hipSuccess) {
		delete ctx;
		return fail(nullptr, LMX_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
	}
	ctx->stream = ctx->own_stream;
	*out = ctx;
	return LMX_OK;
}

void lmx_ctx_destroy(LmxContext* ctx) {
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	prof_drain(ctx);
	for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
	if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
	delete ctx;
}

const char* lmx_last_error(const LmxContext* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int lmx_ctx_set_stream(LmxContext* ctx, void* hip_stream) {
	LMX_CHECK_CTX(ctx);
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	ctx->stream = (hipStream_t)hip_stream; // NULL is the legacy default (null) stream, as everywhere in HIP
	return LMX_OK;
}

int lmx_ctx_synchronize(LmxContext* ctx) {
	LMX_CHECK_CTX(ctx);
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

int lmx_profile_enable(LmxContext* ctx, int enable) {
	LMX_CHECK_CTX(ctx);
	ctx->profiling = enable != 0;
	return LMX_OK;
}

int lmx_profile_reset(LmxContext* ctx) {
	LMX_CHECK_CTX(ctx);
	prof_drain(ctx);
	for (int k = 0; k < LMX_K_COUNT; ++k) {
		ctx->prof_ms[k] = 0;
		ctx->prof_launches[k] = 0;
	}
	return LMX_OK;
}

int lmx_profile_get(LmxContext* ctx, int kernel_id, double* total_ms, uint64_t* launches) {
	LMX_CHECK_CTX(ctx);
	if (kernel_id < 0 || kernel_id >= LMX_K_COUNT) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "kernel id %d", kernel_id);
	prof_drain(ctx);
	if (total_ms) *total_ms = ctx->prof_ms[kernel_id];
	if (launches) *launches = ctx->prof_launches[kernel_id];
	return LMX_OK;
}

const char* lmx_version(void) { return "lumix-mi355 0.1 (gfx950)"; }

// ==========================================================================================================
// culling
// ==========================================================================================================
int lmx_cull_build(LmxContext* ctx, uint32_t n, const int32_t* entity, const uint8_t* type, const double* pos_xyz, const float* radius) {
	LMX_CHECK_CTX(ctx);
	if (n && (!entity || !type || !pos_xyz || !radius)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	CullState& cs = ctx->cull;
	int32_t max_entity = -1;
	for (uint32_t i = 0; i < n; ++i) {
		if (entity[i] < 0) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity[%u] = %d is negative", i, entity[i]);
		if (type[i] >= MAX_TYPES) return fail(ctx, LMX_ERR_CAPACITY, "type[%u] = %u >= LMX_MAX_TYPES", i, type[i]);
		max_entity = std::max(max_entity, entity[i]);
	}
	cs.recs.clear();
	cs.recs.reserve(n);
	cs.mirror_stale = false;
	cs.ent_to_rec.assign((size_t)max_entity + 1, -1);
	for (uint32_t i = 0; i < n; ++i) {
		if (cs.ent_to_rec[entity[i]] >= 0) {
			cs.recs.clear();
			cs.ent_to_rec.clear();
			cs.structure_dirty = true;
			return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d added twice", entity[i]);
		}
		cs.ent_to_rec[entity[i]] = (int32_t)i;
		cs.recs.push_back(make_cull_rec(entity[i], type[i], DV3{pos_xyz[3 * (size_t)i], pos_xyz[3 * (size_t)i + 1], pos_xyz[3 * (size_t)i + 2]}, radius[i]));
	}
	cs.structure_dirty = true;
	return cull_rebuild(ctx);
}

int lmx_cull_add(LmxContext* ctx, int32_t entity, uint8_t type, const double pos[3], float radius) {
	LMX_CHECK_CTX(ctx);
	if (entity < 0 || !pos) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad entity/pos");
	if (type >= MAX_TYPES) return fail(ctx, LMX_ERR_CAPACITY, "type %u >= LMX_MAX_TYPES", type);
	if (int rc = cull_sync_mirror(ctx)) return rc;
	CullState& cs = ctx->cull;
	if ((size_t)entity >= cs.ent_to_rec.size()) cs.ent_to_rec.resize((size_t)entity + 1, -1);
	if (cs.ent_to_rec[entity] >= 0) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d already added", entity);
	cs.ent_to_rec[entity] = (int32_t)cs.recs.size();
	cs.recs.push_back(make_cull_rec(entity, type, DV3{pos[0], pos[1], pos[2]}, radius));
	cs.structure_dirty = true;
	return LMX_OK;
}

int lmx_cull_remove(LmxContext* ctx, int32_t entity) { // culling_system.cpp:160-190 (unknown entities are ignored, :162-165)
	LMX_CHECK_CTX(ctx);
	if (int rc = cull_sync_mirror(ctx)) return rc;
	CullState& cs = ctx->cull;
	if (entity < 0 || (size_t)entity >= cs.ent_to_rec.size() || cs.ent_to_rec[entity] < 0) return LMX_OK;
	const uint32_t rec = (uint32_t)cs.ent_to_rec[entity];
	const uint32_t last = (uint32_t)cs.recs.size() - 1;
	if (rec != last) {
		cs.recs[rec] = cs.recs[last];
		cs.ent_to_rec[cs.recs[rec].entity] = (int32_t)rec;
	}
	cs.recs.pop_back();
	cs.ent_to_rec[entity] = -1;
	cs.structure_dirty = true;
	return LMX_OK;
}

int lmx_cull_set(LmxContext* ctx, int32_t entity, const double pos[3], float radius) { // culling_system.cpp:225-242
	LMX_CHECK_CTX(ctx);
	uint32_t rec;
	if (int rc = cull_sync_mirror(ctx)) return rc;
	if (int rc = cull_find(ctx, entity, &rec)) return rc;
	CullState& cs = ctx->cull;
	CullRec& r = cs.recs[rec];
	const DV3 p = DV3{pos[0], pos[1], pos[2]};
	const IV3 idx = cell_of(p);
	if (r.big == is_big_radius(radius) && idx.x == r.cell.x && idx.y == r.cell.y && idx.z == r.cell.z) {
		r.radius = radius;
		r.rel = to_v3(sub(p, cell_origin(r.cell)));
		cull_mark_patch(cs, rec);
		return LMX_OK;
	}
	cull_readd(cs, rec, p, radius);
	return LMX_OK;
}

int lmx_cull_set_position(LmxContext* ctx, int32_t entity, const double pos[3]) { // culling_system.cpp:201-217
	LMX_CHECK_CTX(ctx);
	uint32_t rec;
	if (int rc = cull_sync_mirror(ctx)) return rc;
	if (int rc = cull_find(ctx, entity, &rec)) return rc;
	CullState& cs = ctx->cull;
	CullRec& r = cs.recs[rec];
	const DV3 p = DV3{pos[0], pos[1], pos[2]};
	const IV3 idx = cell_of(p);
	if (idx.x == r.cell.x && idx.y == r.cell.y && idx.z == r.cell.z) {
		r.rel = to_v3(sub(p, cell_origin(r.cell)));
		cull_mark_patch(cs, rec);
		return LMX_OK;
	}
	cull_readd(cs, rec, p, r.radius);
	return LMX_OK;
}

int lmx_cull_set_radius(LmxContext* ctx, int32_t entity, float radius) { // culling_system.cpp:244-260
	LMX_CHECK_CTX(ctx);
	uint32_t rec;
	if (int rc = cull_sync_mirror(ctx)) return rc;
	if (int rc = cull_find(ctx, entity, &rec)) return rc;
	CullState& cs = ctx->cull;
	CullRec& r = cs.recs[rec];
	if (r.big == is_big_radius(radius)) {
		r.radius = radius;
		cull_mark_patch(cs, rec);
		return LMX_OK;
	}
	const DV3 p = add(cell_origin(r.cell), r.rel); // cell.header.origin + sphere->position
	cull_readd(cs, rec, p, radius);
	return LMX_OK;
}

int lmx_cull_get_radius(LmxContext* ctx, int32_t entity, float* out_radius) {
	LMX_CHECK_CTX(ctx);
	uint32_t rec;
	if (int rc = cull_sync_mirror(ctx)) return rc;
	if (int rc = cull_find(ctx, entity, &rec)) return rc;
	if (out_radius) *out_radius = ctx->cull.recs[rec].radius;
	return LMX_OK;
}

int lmx_cull_is_added(LmxContext* ctx, int32_t entity) {
	if (!ctx) return 0;
	const CullState& cs = ctx->cull;
	return entity >= 0 && (size_t)entity < cs.ent_to_rec.size() && cs.ent_to_rec[entity] >= 0 ? 1 : 0;
}

int lmx_cull_flush(LmxContext* ctx) {
	LMX_CHECK_CTX(ctx);
	return cull_flush(ctx);
}

int lmx_cull_stats(LmxContext* ctx, uint32_t* n_entities, uint32_t* n_cells, uint32_t* n_chunks) {
	LMX_CHECK_CTX(ctx);
	if (int rc = cull_flush(ctx)) return rc;
	const CullState& cs = ctx->cull;
	if (n_entities) *n_entities = (uint32_t)cs.recs.size();
	if (n_cells) *n_cells = cs.n_cells - cs.n_dead_cells;
	if (n_chunks) *n_chunks = cs.n_padded / CHUNK;
	return LMX_OK;
}

int lmx_cull(LmxContext* ctx, uint32_t view, const LmxShiftedFrustum* frusta, uint32_t n_frusta, uint8_t type) {
	LMX_CHECK_CTX(ctx);
	if (view >= LMX_MAX_VIEWS) return fail(ctx, LMX_ERR_CAPACITY, "view %u >= LMX_MAX_VIEWS", view);
	if (!frusta || n_frusta == 0 || n_frusta > LMX_MAX_FRUSTA) return fail(ctx, LMX_ERR_CAPACITY, "n_frusta %u not in [1,%d]", n_frusta, LMX_MAX_FRUSTA);
	if (type != LMX_TYPE_ALL && type >= MAX_TYPES) return fail(ctx, LMX_ERR_CAPACITY, "type %u >= LMX_MAX_TYPES", type);
	if (int rc = cull_flush(ctx)) return rc;
	CullState& cs = ctx->cull;
	CullView& v = cs.views[view];
	if (v.ext_out) {
		if (v.ext_out_cap < (size_t)cs.n_padded * n_frusta)
			return fail(ctx, LMX_ERR_CAPACITY, "bound output holds %zu ids, need %zu", v.ext_out_cap, (size_t)cs.n_padded * n_frusta);
	} else {
		if (!v.counts.p) {
			LMX_HIP(ctx, v.counts.reserve(2 * MAX_FRUSTA * MAX_TYPES));
			v.flip = 0;
			v.next_half_is_zero = false;
		}
		LMX_HIP(ctx, v.out.reserve((size_t)std::max(cs.n_padded, 1u) * n_frusta));
	}
	v.n_frusta = n_frusta;
	v.cell_stride = cs.n_cells;
	v.out_stride = cs.n_padded;
	for (int t = 0; t < MAX_TYPES; ++t) {
		v.ent_start[t] = cs.tt.ent_start[t];
		v.ent_cap[t] = cs.tt.ent_end[t] - cs.tt.ent_start[t];
	}
	FrustaArg fr;
	memset(&fr, 0, sizeof(fr));
	for (uint32_t f = 0; f < n_frusta; ++f) fr.f[f] = to_dev_frustum(frusta[f]);

	uint32_t cell_begin = 0, cell_n = cs.n_cells, ent_begin = 0, ent_end = cs.n_padded;
	if (type != LMX_TYPE_ALL) {
		cell_begin = cs.cell_begin[type];
		cell_n = cs.cell_end[type] - cs.cell_begin[type];
		ent_begin = cs.tt.ent_start[type];
		ent_end = cs.tt.ent_end[type];
	}
	const CullDeviceView dv = cull_dev(cs);
	// fused single-launch path when the per-tile cell table fits the dynamic-LDS budget, else classify + spheres
	const uint32_t tile = cull_tile_size((int)n_frusta);
	const uint32_t tile_k = tile == 4096 ? 0 : (tile == 2048 ? 1 : 2);
	const uint32_t cell_cap = cs.tile_cap[tile_k];
	static const bool force_two_kernels = getenv("LMX_CULL_TWO_KERNELS") != nullptr;
	const bool fused = !force_two_kernels && fused_lds_bytes((int)n_frusta, tile, cell_cap) <= 64 * 1024;
	if (fused) {
		uint32_t* counts_next = nullptr;
		if (v.ext_counts) {
			LMX_HIP(ctx, hipMemsetAsync(v.ext_counts, 0, sizeof(uint32_t) * MAX_FRUSTA * MAX_TYPES, ctx->stream));
		} else {
			v.flip ^= 1u;
			if (!v.next_half_is_zero) LMX_HIP(ctx, hipMemsetAsync(v.counts_ptr(), 0, sizeof(uint32_t) * MAX_FRUSTA * MAX_TYPES, ctx->stream));
			counts_next = v.counts_other();
			v.next_half_is_zero = ent_end > ent_begin; // block 0 of the launch below clears it
		}
		ProfScope ps(ctx, LMX_K_CULL_SPHERES);
		LMX_HIP(ctx, launch_cull_fused(ctx->stream, dv, ent_begin, ent_end, cs.tt, fr, (int)n_frusta, v.out_ptr(), v.out_stride,
			v.counts_ptr(), counts_next));
	} else {
		if (!v.ext_counts) {
			v.flip ^= 1u;
			v.next_half_is_zero = false;
		}
		LMX_HIP(ctx, v.cellinfo.reserve((size_t)std::max(cs.n_cells, 1u) * n_frusta));
		{
			ProfScope ps(ctx, LMX_K_CULL_CLASSIFY);
			LMX_HIP(ctx, launch_cull_classify(ctx->stream, dv, cell_begin, cell_n, fr, (int)n_frusta, v.cellinfo.p, v.cell_stride, v.counts_ptr()));
		}
		{
			ProfScope ps(ctx, LMX_K_CULL_SPHERES);
			LMX_HIP(ctx, launch_cull_spheres(ctx->stream, dv, ent_begin, ent_end, cs.tt, fr, (int)n_frusta, v.cellinfo.p, v.cell_stride, v.out_ptr(),
				v.out_stride, v.counts_ptr()));
		}
	}
	v.valid = true;
	return LMX_OK;
}

int lmx_cull_counts(LmxContext* ctx, uint32_t view, uint32_t* counts) {
	LMX_CHECK_CTX(ctx);
	if (view >= LMX_MAX_VIEWS || !counts) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad view/counts");
	CullView& v = ctx->cull.views[view];
	if (!v.valid) return fail(ctx, LMX_ERR_NOT_BUILT, "view %u holds no cull result", view);
	uint32_t all[MAX_FRUSTA * MAX_TYPES];
	LMX_HIP(ctx, hipMemcpyAsync(all, v.counts_ptr(), sizeof(all), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	memcpy(counts, all, sizeof(uint32_t) * v.n_frusta * MAX_TYPES);
	return LMX_OK;
}

int lmx_cull_read(LmxContext* ctx, uint32_t view, uint32_t frustum, uint8_t type, int32_t* out_ids, uint32_t cap, uint32_t* out_count) {
	LMX_CHECK_CTX(ctx);
	if (view >= LMX_MAX_VIEWS) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad view");
	CullView& v = ctx->cull.views[view];
	if (!v.valid) return fail(ctx, LMX_ERR_NOT_BUILT, "view %u holds no cull result", view);
	if (frustum >= v.n_frusta || type >= MAX_TYPES) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "frustum %u / type %u out of range", frustum, type);
	uint32_t c = 0;
	LMX_HIP(ctx, hipMemcpyAsync(&c, v.counts_ptr() + frustum * MAX_TYPES + type, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (out_count) *out_count = c;
	if (c > v.ent_cap[type]) return fail(ctx, LMX_ERR_HIP, "corrupt count %u > %u", c, v.ent_cap[type]);
	if (!out_ids || c == 0) return LMX_OK;
	if (c > cap) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u ids, got %u", c, cap);
	LMX_HIP(ctx, hipMemcpyAsync(out_ids, v.out_ptr() + (size_t)frustum * v.out_stride + v.ent_start[type], (size_t)c * sizeof(int32_t),
		hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

int lmx_cull_bind_output(LmxContext* ctx, uint32_t view, void* d_ids, size_t ids_capacity, void* d_counts) {
	LMX_CHECK_CTX(ctx);
	if (view >= LMX_MAX_VIEWS) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad view");
	if ((d_ids == nullptr) != (d_counts == nullptr)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bind both buffers or neither");
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	CullView& v = ctx->cull.views[view];
	v.ext_out = (int32_t*)d_ids;
	v.ext_out_cap = d_ids ? ids_capacity : 0;
	v.ext_counts = (uint32_t*)d_counts;
	v.valid = false;
	return LMX_OK;
}

int lmx_cull_device_result(LmxContext* ctx, uint32_t view, uint32_t frustum, const int32_t** d_ids, const uint32_t** d_counts,
	uint32_t* type_offsets, uint32_t* capacity) {
	LMX_CHECK_CTX(ctx);
	if (view >= LMX_MAX_VIEWS) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad view");
	CullView& v = ctx->cull.views[view];
	if (!v.valid) return fail(ctx, LMX_ERR_NOT_BUILT, "view %u holds no cull result", view);
	if (frustum >= v.n_frusta) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "frustum %u out of range", frustum);
	if (d_ids) *d_ids = v.out_ptr() + (size_t)frustum * v.out_stride;
	if (d_counts) *d_counts = v.counts_ptr();
	if (type_offsets) memcpy(type_offsets, v.ent_start, sizeof(v.ent_start));
	if (capacity) *capacity = v.out_stride;
	return LMX_OK;
}

// ==========================================================================================================
// world transforms
// ==========================================================================================================
int lmx_world_build(LmxContext* ctx, uint32_t n, const int32_t* parent, const LmxTransform* transforms) {
	LMX_CHECK_CTX(ctx);
	if (n && (!parent || !transforms)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	WorldState& w = ctx->world;
	w.built = false;
	// children lists (CSR by parent), then BFS from the roots: slot order = (level, parent slot)
	std::vector<uint32_t> child_start((size_t)n + 1, 0);
	for (uint32_t e = 0; e < n; ++e) {
		if (parent[e] >= (int32_t)n || parent[e] == (int32_t)e) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "parent[%u] = %d invalid", e, parent[e]);
		if (parent[e] >= 0) child_start[(size_t)parent[e] + 1]++;
	}
	for (uint32_t e = 0; e < n; ++e) child_start[e + 1] += child_start[e];
	std::vector<uint32_t> child_list(child_start[n]);
	{
		std::vector<uint32_t> cursor(child_start.begin(), child_start.end() - 1);
		for (uint32_t e = 0; e < n; ++e)
			if (parent[e] >= 0) child_list[cursor[parent[e]]++] = e;
	}
	w.entity_of_slot.clear();
	w.entity_of_slot.reserve(n);
	w.level_start.clear();
	w.level_start.push_back(0);
	for (uint32_t e = 0; e < n; ++e)
		if (parent[e] < 0) w.entity_of_slot.push_back((int32_t)e);
	size_t level_begin = 0;
	while (level_begin < w.entity_of_slot.size()) {
		const size_t level_end = w.entity_of_slot.size();
		w.level_start.push_back((uint32_t)level_end);
		for (size_t s = level_begin; s < level_end; ++s) {
			const uint32_t e = (uint32_t)w.entity_of_slot[s];
			for (uint32_t k = child_start[e]; k < child_start[e + 1]; ++k) w.entity_of_slot.push_back((int32_t)child_list[k]);
		}
		level_begin = level_end;
	}
	if (w.entity_of_slot.size() != n) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "hierarchy contains a cycle (%zu of %u entities reachable)", w.entity_of_slot.size(), n);
	w.slot_of_entity.assign(n, -1);
	for (uint32_t s = 0; s < n; ++s) w.slot_of_entity[w.entity_of_slot[s]] = (int32_t)s;
	w.parent_slot.assign(n, -1);
	for (uint32_t s = 0; s < n; ++s) {
		const int32_t p = parent[w.entity_of_slot[s]];
		w.parent_slot[s] = p < 0 ? -1 : w.slot_of_entity[p];
	}
	w.n = n;
	const size_t cap = std::max(n, 1u);
	for (auto& b : w.pos) LMX_HIP(ctx, b.reserve(cap));
	for (auto& b : w.rot) LMX_HIP(ctx, b.reserve(cap));
	for (auto& b : w.scl) LMX_HIP(ctx, b.reserve(cap));
	LMX_HIP(ctx, w.d_parent_slot.reserve(cap));
	LMX_HIP(ctx, w.d_slot_of_entity.reserve(cap));
	LMX_HIP(ctx, w.d_entity_of_slot.reserve(cap));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (n) {
		LMX_HIP(ctx, hipMemcpy(w.d_parent_slot.p, w.parent_slot.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(w.d_slot_of_entity.p, w.slot_of_entity.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(w.d_entity_of_slot.p, w.entity_of_slot.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
		// initial values: every entity's transform is staged through the scatter kernel (roots -> world, children -> local)
		std::vector<int32_t> all(n);
		for (uint32_t e = 0; e < n; ++e) all[e] = (int32_t)e;
		LMX_HIP(ctx, w.d_stage_entity.reserve(n));
		LMX_HIP(ctx, w.d_stage_tr.reserve(n));
		LMX_HIP(ctx, hipMemcpy(w.d_stage_entity.p, all.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(w.d_stage_tr.p, transforms, n * sizeof(LmxTransform), hipMemcpyHostToDevice));
		LMX_HIP(ctx, launch_xform_scatter(ctx->stream, w.dev(), w.d_slot_of_entity.p, w.d_stage_entity.p, w.d_stage_tr.p, n));
	}
	w.bound_entity.clear();
	w.bound_radius.clear();
	w.bound_generation = ~0ull;
	w.built = true;
	return LMX_OK;
}

int lmx_world_set_transforms(LmxContext* ctx, uint32_t n, const int32_t* entity, const LmxTransform* transforms) {
	LMX_CHECK_CTX(ctx);
	WorldState& w = ctx->world;
	if (!w.built) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_world_build has not been called");
	if (!n) return LMX_OK;
	if (!entity || !transforms) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	for (uint32_t i = 0; i < n; ++i)
		if (entity[i] < 0 || (uint32_t)entity[i] >= w.n) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity[%u] = %d out of range", i, entity[i]);
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream)); // staging buffers may still be read by a previous scatter
	LMX_HIP(ctx, w.d_stage_entity.reserve(n));
	LMX_HIP(ctx, w.d_stage_tr.reserve(n));
	LMX_HIP(ctx, hipMemcpy(w.d_stage_entity.p, entity, n * sizeof(int32_t), hipMemcpyHostToDevice));
	LMX_HIP(ctx, hipMemcpy(w.d_stage_tr.p, transforms, n * sizeof(LmxTransform), hipMemcpyHostToDevice));
	LMX_HIP(ctx, launch_xform_scatter(ctx->stream, w.dev(), w.d_slot_of_entity.p, w.d_stage_entity.p, w.d_stage_tr.p, n));
	return LMX_OK;
}

int lmx_world_set_transforms_device(LmxContext* ctx, uint32_t n, const void* d_entity, const void* d_transforms) {
	LMX_CHECK_CTX(ctx);
	WorldState& w = ctx->world;
	if (!w.built) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_world_build has not been called");
	if (!n) return LMX_OK;
	if (!d_entity || !d_transforms) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null device pointer");
	LMX_HIP(ctx, launch_xform_scatter(ctx->stream, w.dev(), w.d_slot_of_entity.p, (const int32_t*)d_entity, d_transforms, n));
	return LMX_OK;
}

int lmx_world_bind_culling(LmxContext* ctx, uint32_t n, const int32_t* entity, const float* model_radius) {
	LMX_CHECK_CTX(ctx);
	WorldState& w = ctx->world;
	if (!w.built) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_world_build has not been called");
	if (n && (!entity || !model_radius)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	for (uint32_t i = 0; i < n; ++i) {
		if (entity[i] < 0 || (uint32_t)entity[i] >= w.n) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity[%u] = %d out of range", i, entity[i]);
		if (!lmx_cull_is_added(ctx, entity[i])) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d is not in the culling system", entity[i]);
	}
	w.bound_entity.assign(entity, entity + n);
	w.bound_radius.assign(model_radius, model_radius + n);
	w.bound_generation = ~0ull;
	return LMX_OK;
}

static int world_upload_binding(LmxContext* ctx) {
	WorldState& w = ctx->world;
	CullState& cs = ctx->cull;
	if (int rc = cull_flush(ctx)) return rc;
	if (w.bound_generation == cs.generation) return LMX_OK;
	const size_t n = w.bound_entity.size();
	std::vector<uint32_t> slot(n), sphere(n);
	for (size_t i = 0; i < n; ++i) {
		slot[i] = (uint32_t)w.slot_of_entity[w.bound_entity[i]];
		sphere[i] = cs.rec_slot[cs.ent_to_rec[w.bound_entity[i]]];
	}
	LMX_HIP(ctx, w.d_bound_slot.reserve(n));
	LMX_HIP(ctx, w.d_bound_sphere.reserve(n));
	LMX_HIP(ctx, w.d_bound_radius.reserve(n));
	LMX_HIP(ctx, w.d_rebin.reserve(n));
	LMX_HIP(ctx, w.d_rebin_count.reserve(1));
	LMX_HIP(ctx, cs.sphere_cell.reserve(std::max(cs.n_padded, 1u)));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	LMX_HIP(ctx, hipMemcpy(w.d_bound_slot.p, slot.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
	LMX_HIP(ctx, hipMemcpy(w.d_bound_sphere.p, sphere.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
	LMX_HIP(ctx, hipMemcpy(w.d_bound_radius.p, w.bound_radius.data(), n * sizeof(float), hipMemcpyHostToDevice));
	if (cs.n_padded) LMX_HIP(ctx, hipMemcpy(cs.sphere_cell.p, cs.h_sphere_cell.data(), (size_t)cs.n_padded * sizeof(uint32_t), hipMemcpyHostToDevice));
	w.bound_generation = cs.generation;
	return LMX_OK;
}

int lmx_world_propagate(LmxContext* ctx) {
	LMX_CHECK_CTX(ctx);
	WorldState& w = ctx->world;
	if (!w.built) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_world_build has not been called");
	const WorldDevice dev = w.dev();
	for (size_t l = 1; l + 1 < w.level_start.size(); ++l) {
		ProfScope ps(ctx, LMX_K_XFORM_LEVEL);
		LMX_HIP(ctx, launch_xform_level(ctx->stream, dev, w.level_start[l], w.level_start[l + 1] - w.level_start[l]));
	}
	if (!w.bound_entity.empty()) {
		if (int rc = world_upload_binding(ctx)) return rc;
		CullState& cs = ctx->cull;
		const uint32_t n = (uint32_t)w.bound_entity.size();
		LMX_HIP(ctx, hipMemsetAsync(w.d_rebin_count.p, 0, sizeof(uint32_t), ctx->stream));
		{
			ProfScope ps(ctx, LMX_K_SPHERE_REFRESH);
			LMX_HIP(ctx, launch_sphere_refresh(ctx->stream, dev, w.d_bound_slot.p, w.d_bound_sphere.p, w.d_bound_radius.p, cs.sphere_cell.p,
				cs.cells.p, cs.spheres.p, n, w.d_rebin_count.p, w.d_rebin.p));
		}
		// Entities that left their cell (or crossed the is_big threshold) go through the host mirror, exactly like
		// CullingSystem::set -> remove + add; everything else was refreshed in place on the device, which makes the
		// device copy authoritative until the mirror is synchronised (cull_sync_mirror).
		uint32_t rebin = 0;
		LMX_HIP(ctx, hipMemcpyAsync(&rebin, w.d_rebin_count.p, sizeof(rebin), hipMemcpyDeviceToHost, ctx->stream));
		LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		cs.mirror_stale = true;
		if (rebin) {
			std::vector<RebinItem> items(rebin);
			LMX_HIP(ctx, hipMemcpy(items.data(), w.d_rebin.p, (size_t)rebin * sizeof(RebinItem), hipMemcpyDeviceToHost));
			if (int rc = cull_sync_mirror(ctx)) return rc;
			for (const RebinItem& it : items) {
				const uint32_t rec = (uint32_t)cs.ent_to_rec[w.bound_entity[it.bound_index]];
				cull_readd(cs, rec, DV3{it.pos[0], it.pos[1], it.pos[2]}, it.radius);
			}
		}
	}
	return LMX_OK;
}

int lmx_world_read_transforms(LmxContext* ctx, LmxTransform* out, uint32_t n) {
	LMX_CHECK_CTX(ctx);
	WorldState& w = ctx->world;
	if (!w.built) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_world_build has not been called");
	if (n < w.n || !out) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u transforms", w.n);
	if (!w.n) return LMX_OK;
	LMX_HIP(ctx, w.d_export.reserve(w.n));
	LMX_HIP(ctx, launch_xform_export(ctx->stream, w.dev(), w.d_entity_of_slot.p, w.n, w.d_export.p));
	LMX_HIP(ctx, hipMemcpyAsync(out, w.d_export.p, (size_t)w.n * sizeof(LmxTransform), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}


// ==========================================================================================================
// skinning
// ==========================================================================================================
int lmx_skin_add_model(LmxContext* ctx, uint32_t n_bones, const int16_t* parents, const LmxLocalRigidTransform* bind, int32_t first_nonroot,
	uint32_t* out_model) {
	LMX_CHECK_CTX(ctx);
	if (!n_bones || n_bones > LMX_MAX_BONES) return fail(ctx, LMX_ERR_CAPACITY, "n_bones %u not in [1,%d] (Model::Bone::MAX_COUNT)", n_bones, LMX_MAX_BONES);
	if (!parents || !bind) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	if (first_nonroot < 0) first_nonroot = (int32_t)n_bones;
	SkinState& sk = ctx->skin;
	SkinModel m;
	m.bone_offset = (uint32_t)sk.parents.size();
	m.n_bones = n_bones;
	m.first_nonroot = first_nonroot;
	m.max_depth = 0;
	std::vector<uint8_t> depth(n_bones, 0);
	for (uint32_t i = 0; i < n_bones; ++i) {
		const int32_t p = parents[i];
		if (p >= (int32_t)i) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "parents[%u] = %d must precede the bone (model.cpp:381-384)", i, p);
		if ((int32_t)i >= first_nonroot && p < 0) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bone %u >= first_nonroot has no parent", i);
		// bones below first_nonroot are never touched by Pose::computeAbsolute (pose.cpp:66): depth 0
		depth[i] = ((int32_t)i >= first_nonroot) ? (uint8_t)(depth[p] + 1) : 0;
		m.max_depth = std::max<uint32_t>(m.max_depth, depth[i]);
	}
	for (uint32_t i = 0; i < n_bones; ++i) {
		V3 ip;
		Q4 ir;
		invert_rigid(V3{bind[i].pos[0], bind[i].pos[1], bind[i].pos[2]}, Q4{bind[i].rot[0], bind[i].rot[1], bind[i].rot[2], bind[i].rot[3]}, &ip, &ir);
		sk.parents.push_back(parents[i]);
		sk.depth.push_back(depth[i]);
		sk.inv_pos.push_back(ip.x);
		sk.inv_pos.push_back(ip.y);
		sk.inv_pos.push_back(ip.z);
		sk.inv_rot.push_back(make_float4(ir.x, ir.y, ir.z, ir.w));
	}
	sk.models.push_back(m);
	sk.models_dirty = true;
	if (out_model) *out_model = (uint32_t)sk.models.size() - 1;
	return LMX_OK;
}

int lmx_skin_add_mesh(LmxContext* ctx, uint32_t n_verts, const float* positions_xyz, const LmxSkin* skin, uint32_t* out_mesh) {
	LMX_CHECK_CTX(ctx);
	if (!n_verts || !positions_xyz || !skin) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "empty mesh / null input array");
	SkinState& sk = ctx->skin;
	SkinMesh m;
	m.vert_offset = (uint32_t)(sk.verts.size() / 3);
	m.n_verts = n_verts;
	sk.verts.insert(sk.verts.end(), positions_xyz, positions_xyz + (size_t)n_verts * 3);
	for (uint32_t v = 0; v < n_verts; ++v) {
		sk.weights.push_back(make_float4(skin[v].weights[0], skin[v].weights[1], skin[v].weights[2], skin[v].weights[3]));
		for (int k = 0; k < 4; ++k) {
			if (skin[v].indices[k] < 0 || skin[v].indices[k] >= LMX_MAX_BONES) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "skin[%u].indices[%d] = %d out of range", v, k, skin[v].indices[k]);
			sk.indices.push_back(skin[v].indices[k]);
		}
	}
	sk.meshes.push_back(m);
	sk.meshes_dirty = true;
	if (out_mesh) *out_mesh = (uint32_t)sk.meshes.size() - 1;
	return LMX_OK;
}

static int skin_upload_static(LmxContext* ctx) {
	SkinState& sk = ctx->skin;
	if (sk.models_dirty) {
		const size_t nb = sk.parents.size();
		LMX_HIP(ctx, sk.d_parents.reserve(nb));
		LMX_HIP(ctx, sk.d_depth.reserve(nb));
		LMX_HIP(ctx, sk.d_inv_pos.reserve(nb * 3));
		LMX_HIP(ctx, sk.d_inv_rot.reserve(nb));
		LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		LMX_HIP(ctx, hipMemcpy(sk.d_parents.p, sk.parents.data(), nb * sizeof(int16_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(sk.d_depth.p, sk.depth.data(), nb * sizeof(uint8_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(sk.d_inv_pos.p, sk.inv_pos.data(), nb * 3 * sizeof(float), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(sk.d_inv_rot.p, sk.inv_rot.data(), nb * sizeof(float4), hipMemcpyHostToDevice));
		sk.models_dirty = false;
	}
	if (sk.meshes_dirty) {
		const size_t nv = sk.weights.size();
		LMX_HIP(ctx, sk.d_verts.reserve(nv * 3));
		LMX_HIP(ctx, sk.d_weights.reserve(nv));
		LMX_HIP(ctx, sk.d_indices.reserve(nv * 4));
		LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		LMX_HIP(ctx, hipMemcpy(sk.d_verts.p, sk.verts.data(), nv * 3 * sizeof(float), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(sk.d_weights.p, sk.weights.data(), nv * sizeof(float4), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(sk.d_indices.p, sk.indices.data(), nv * 4 * sizeof(int16_t), hipMemcpyHostToDevice));
		sk.meshes_dirty = false;
	}
	return LMX_OK;
}

int lmx_skin_set_instances(LmxContext* ctx, uint32_t n, const uint32_t* model, const uint32_t* mesh) {
	LMX_CHECK_CTX(ctx);
	if (n && (!model || !mesh)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	SkinState& sk = ctx->skin;
	std::vector<SkinInstance> inst(n);
	size_t bones = 0, verts = 0;
	uint32_t max_verts = 0;
	for (uint32_t i = 0; i < n; ++i) {
		if (model[i] >= sk.models.size() || mesh[i] >= sk.meshes.size()) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "instance %u: unknown model/mesh id", i);
		const SkinModel& mo = sk.models[model[i]];
		const SkinMesh& me = sk.meshes[mesh[i]];
		SkinInstance& in = inst[i];
		if (bones + mo.n_bones > 0xffffffffull || verts + me.n_verts > 0xffffffffull) return fail(ctx, LMX_ERR_CAPACITY, "instance table exceeds 2^32 bones or vertices");
		in.bone_offset = (uint32_t)bones;
		in.n_bones = mo.n_bones;
		in.model_offset = mo.bone_offset;
		in.first_nonroot = mo.first_nonroot;
		in.vert_offset = me.vert_offset;
		in.n_verts = me.n_verts;
		in.out_offset = (uint32_t)verts;
		in.max_depth = mo.max_depth;
		bones += mo.n_bones;
		verts += me.n_verts;
		max_verts = std::max(max_verts, me.n_verts);
	}
	sk.inst.swap(inst);
	sk.bones_total = bones;
	sk.verts_total = verts;
	sk.max_verts = max_verts;
	sk.poses_uploaded = false;
	LMX_HIP(ctx, sk.d_inst.reserve(std::max<size_t>(n, 1)));
	LMX_HIP(ctx, sk.d_pose_pos.reserve(std::max<size_t>(bones * 3, 1)));
	LMX_HIP(ctx, sk.d_pose_rot.reserve(std::max<size_t>(bones, 1)));
	LMX_HIP(ctx, sk.d_palette.reserve(std::max<size_t>(bones * 4, 1)));
	LMX_HIP(ctx, sk.d_out.reserve(std::max<size_t>(verts * 3, 1)));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (n) LMX_HIP(ctx, hipMemcpy(sk.d_inst.p, sk.inst.data(), (size_t)n * sizeof(SkinInstance), hipMemcpyHostToDevice));
	return LMX_OK;
}

int lmx_skin_upload_poses(LmxContext* ctx, const float* positions, const float* rotations, size_t n_bones_total) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (n_bones_total != sk.bones_total) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "expected %zu bones over all instances, got %zu", sk.bones_total, n_bones_total);
	if (!n_bones_total) return LMX_OK;
	if (!positions || !rotations) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	LMX_HIP(ctx, hipMemcpy(sk.d_pose_pos.p, positions, n_bones_total * 3 * sizeof(float), hipMemcpyHostToDevice));
	LMX_HIP(ctx, hipMemcpy(sk.d_pose_rot.p, rotations, n_bones_total * sizeof(float4), hipMemcpyHostToDevice));
	sk.poses_uploaded = true;
	return LMX_OK;
}

int lmx_skin_upload_poses_device(LmxContext* ctx, const void* d_positions, const void* d_rotations, size_t n_bones_total) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (n_bones_total != sk.bones_total) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "expected %zu bones over all instances, got %zu", sk.bones_total, n_bones_total);
	if (!n_bones_total) return LMX_OK;
	if (!d_positions || !d_rotations) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null device pointer");
	LMX_HIP(ctx, hipMemcpyAsync(sk.d_pose_pos.p, d_positions, n_bones_total * 3 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
	LMX_HIP(ctx, hipMemcpyAsync(sk.d_pose_rot.p, d_rotations, n_bones_total * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
	sk.poses_uploaded = true;
	return LMX_OK;
}

int lmx_skin_run(LmxContext* ctx) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (sk.inst.empty()) return LMX_OK;
	if (!sk.poses_uploaded) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_skin_upload_poses has not been called for this instance table");
	if (int rc = skin_upload_static(ctx)) return rc;
	const uint32_t n = (uint32_t)sk.inst.size();
	{
		ProfScope ps(ctx, LMX_K_POSE_PALETTE);
		LMX_HIP(ctx, launch_pose_palette(ctx->stream, sk.d_inst.p, n, sk.d_pose_pos.p, sk.d_pose_rot.p, sk.d_parents.p, sk.d_depth.p, sk.d_inv_pos.p,
			sk.d_inv_rot.p, sk.d_palette.p));
	}
	{
		ProfScope ps(ctx, LMX_K_SKIN_VERTICES);
		LMX_HIP(ctx, launch_skin_vertices(ctx->stream, sk.d_inst.p, n, sk.max_verts, sk.d_verts.p, sk.d_weights.p, sk.d_indices.p, sk.d_palette.p,
			sk.d_out.p));
	}
	// the poses are absolute now; running again needs fresh relative poses (Pose::is_absolute, pose.cpp:64)
	sk.poses_uploaded = false;
	return LMX_OK;
}

int lmx_skin_read_vertices(LmxContext* ctx, uint32_t instance, float* out_xyz, uint32_t cap_verts) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (instance >= sk.inst.size() || !out_xyz) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad instance/out");
	const SkinInstance& in = sk.inst[instance];
	if (cap_verts < in.n_verts) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u vertices", in.n_verts);
	LMX_HIP(ctx, hipMemcpyAsync(out_xyz, sk.d_out.p + (size_t)in.out_offset * 3, (size_t)in.n_verts * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

int lmx_skin_read_palette(LmxContext* ctx, uint32_t instance, LmxMatrix* out, uint32_t cap_bones) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (instance >= sk.inst.size() || !out) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad instance/out");
	const SkinInstance& in = sk.inst[instance];
	if (cap_bones < in.n_bones) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u bones", in.n_bones);
	LMX_HIP(ctx, hipMemcpyAsync(out, sk.d_palette.p + (size_t)in.bone_offset * 4, (size_t)in.n_bones * sizeof(LmxMatrix), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

int lmx_skin_read_pose(LmxContext* ctx, uint32_t instance, float* out_pos, float* out_rot, uint32_t cap_bones) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (instance >= sk.inst.size()) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad instance");
	const SkinInstance& in = sk.inst[instance];
	if (cap_bones < in.n_bones) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u bones", in.n_bones);
	if (out_pos) LMX_HIP(ctx, hipMemcpyAsync(out_pos, sk.d_pose_pos.p + (size_t)in.bone_offset * 3, (size_t)in.n_bones * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	if (out_rot) LMX_HIP(ctx, hipMemcpyAsync(out_rot, sk.d_pose_rot.p + in.bone_offset, (size_t)in.n_bones * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

} // extern "C"
