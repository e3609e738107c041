// lmx_capi_cull.hip — CullingSystem behind the C ABI (include/lumix_mi355.h, "culling" section).
//
// Two resident sets, one visibility function:
//   * static set   host mirror of (cell, cell-relative sphere) per entity + a device layout sorted by (type, is_big, cell)
//                  with chunk headers and tile-major cell keys (lmx_cull_layout.h), culled by k_cull_tile. Between two
//                  compactions the layout only takes O(1) patches: an in-cell move rewrites 16 B, a removal turns the id
//                  into a tombstone (-1), and an entity that is added or leaves its cell goes to the dynamic set.
//   * dynamic set  unsorted world position (fp64) + radius per entity: entities bound to the world hierarchy
//                  (lmx_world_bind_culling, refreshed on the device by lmx_world_propagate) and the overflow of the static
//                  set. k_cull_dynamic re-derives cell, cell-relative position and per-cell class per entity — what
//                  CullingSystem::set + cullInternal would compute (culling_system.cpp:225-242, 321-369). Slots are
//                  stable: add takes a free slot, remove frees one, both cost one 40-byte patch.
// A compaction (structure rebuild) folds the unbound part of the dynamic set back into the sorted layout once it outgrows a
// threshold; it is the only operation that costs O(n).
// Both kernels write to per-shard output windows (see CullOut); k_cull_finalize / k_cull_consolidate turn those into per-type
// totals and contiguous lists on demand.
#include "lmx_context.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <thread>

using namespace lmx;

namespace {

static_assert(sizeof(LayoutSphere) == sizeof(float4) && sizeof(LayoutCell) == sizeof(CellKey) && sizeof(LayoutChunkHdr) == sizeof(ChunkHdr), "layout PODs mirror the device types");
static_assert(LAYOUT_MAX_TYPES == MAX_TYPES && LAYOUT_CHUNK == CHUNK && LAYOUT_TILE_ALIGN == TILE_ALIGN && LAYOUT_CELL_DEAD == CELL_DEAD, "layout constants");

constexpr uint32_t DYN_ALIGN = 2048;       // largest k_cull_dynamic tile: a tile never straddles two types
constexpr uint32_t DYN_MAX_SHARDS = 8;     // output shards per type of the dynamic set

enum class Where { NONE, STATIC, DYNAMIC };

// ---- asynchronous compaction: types (the machinery is further down) -------------------------------------------------------------
enum : uint8_t { OP_ADD, OP_REMOVE, OP_SET, OP_SET_POS, OP_SET_RADIUS, OP_BIND, OP_UNBIND };
struct PinnedUploader; // further down
struct CullOp { // one EFFECTIVE mutation of the live set, replayed onto the shadow set
	double pos[3];
	float radius;
	int32_t entity;
	uint8_t op, type;
};
} // namespace

namespace lmx {
struct CullAsync {
	enum State : int { IDLE, REQUESTED, RUNNING, READY, FAILED, QUIT };
	CullSet shadow;                 // owned by the worker while RUNNING, by the update thread otherwise
	std::vector<CullOp> log_local;  // update thread only: operations since the last hand-over
	std::vector<CullOp> log_shared; // under `mu`: operations the shadow set has not seen yet
	std::mutex mu;
	std::condition_variable cv;
	std::condition_variable cv_idle; // signalled by the worker when a job ends (async_wait_idle sleeps on it)
	State state = IDLE;             // under `mu`
	std::thread worker;
	hipStream_t stream = nullptr;   // the worker's own (non-blocking) stream
	PinnedUploader* uploader = nullptr; // its host -> device copies go through pinned staging buffers
	hipEvent_t swapped = nullptr;   // recorded on the context's stream when the sets trade places: the worker's uploads into what WAS the live set wait for it
	bool swapped_pending = false;
	uint32_t overflow_reserve = 0;  // copy of the tuning value for the job in flight
	bool drain_only = false;        // the job in flight only brings the shadow's mirror up to date (the log had grown long with no re-sort due)
	uint64_t drains = 0;
	std::string error;              // the worker's failure (state FAILED)
	DevBuf<int32_t> d_new_slot;     // entity -> dynamic slot of the shadow set (bound spheres are copied device to device at the swap)
	uint32_t n_new_slot = 0;
	uint64_t jobs_done = 0, swaps = 0, ops_replayed_at_swap = 0;
};
} // namespace lmx

namespace {

inline void async_log(CullState& cs, uint8_t op, int32_t entity, uint8_t type, const double* pos, float radius) {
	if (!cs.async) return;
	CullOp o;
	o.pos[0] = pos ? pos[0] : 0.0;
	o.pos[1] = pos ? pos[1] : 0.0;
	o.pos[2] = pos ? pos[2] : 0.0;
	o.radius = radius;
	o.entity = entity;
	o.op = op;
	o.type = type;
	cs.async->log_local.push_back(o);
}

Where locate(const CullSet& cs, int32_t entity, uint32_t* index) {
	if (entity < 0) return Where::NONE;
	if ((size_t)entity < cs.ent_to_rec.size() && cs.ent_to_rec[entity] >= 0) {
		*index = (uint32_t)cs.ent_to_rec[entity];
		return Where::STATIC;
	}
	if ((size_t)entity < cs.ent_to_dyn.size() && cs.ent_to_dyn[entity] >= 0) {
		*index = (uint32_t)cs.ent_to_dyn[entity];
		return Where::DYNAMIC;
	}
	return Where::NONE;
}

bool layout_live(const CullSet& cs) { return cs.built && !cs.structure_dirty; }

// ---- dynamic set: slots and patches -----------------------------------------------------------------------------
void queue_dyn_patch(CullSet& cs, const DynRec& r, bool alive) {
	if (r.slot == DYN_NO_SLOT || cs.dyn_layout_dirty) return; // the pending rebuild uploads the whole mirror
	const PatchDyn p{r.slot, alive ? r.entity : -1, r.radius, 0u, r.pos[0], r.pos[1], r.pos[2]};
	if (cs.q_dyn_at.size() < cs.dyn_padded) cs.q_dyn_at.resize(cs.dyn_padded, ~0u);
	uint32_t& at = cs.q_dyn_at[r.slot];
	if (at != ~0u) { // a freed slot taken again / an entity set twice before the next flush: the last write wins
		cs.q_dyn[at] = p;
		return;
	}
	at = (uint32_t)cs.q_dyn.size();
	cs.q_dyn.push_back(p);
}

uint32_t take_dyn_slot(CullSet& cs, uint8_t type) {
	if (cs.dyn_layout_dirty) return DYN_NO_SLOT;
	if (!cs.dyn_free[type].empty()) {
		const uint32_t s = cs.dyn_free[type].back();
		cs.dyn_free[type].pop_back();
		return s;
	}
	if (cs.dyn_next[type] < cs.dyn_tt.ent_end[type]) return cs.dyn_next[type]++;
	cs.dyn_layout_dirty = true; // region full: the next flush reassigns every slot with more room
	return DYN_NO_SLOT;
}

void dyn_append(CullSet& cs, int32_t entity, uint8_t type, DV3 pos, float radius, bool bound) {
	if ((size_t)entity >= cs.ent_to_dyn.size()) cs.ent_to_dyn.resize((size_t)entity + 1, -1);
	cs.ent_to_dyn[entity] = (int32_t)cs.dyn.size();
	DynRec r{{pos.x, pos.y, pos.z}, radius, entity, take_dyn_slot(cs, type), type, bound};
	cs.dyn.push_back(r);
	if (!bound) cs.n_unbound++;
	queue_dyn_patch(cs, r, true);
}

void remove_dynamic(CullSet& cs, uint32_t idx) {
	const DynRec r = cs.dyn[idx];
	queue_dyn_patch(cs, r, false);
	if (r.slot != DYN_NO_SLOT && !cs.dyn_layout_dirty) cs.dyn_free[r.type].push_back(r.slot);
	if (!r.bound) cs.n_unbound--;
	const uint32_t last = (uint32_t)cs.dyn.size() - 1;
	if (idx != last) {
		cs.dyn[idx] = cs.dyn[last];
		cs.ent_to_dyn[cs.dyn[idx].entity] = (int32_t)idx;
	}
	cs.dyn.pop_back();
	cs.ent_to_dyn[r.entity] = -1;
}

// ---- static set: host mirror ops --------------------------------------------------------------------------------
void remove_static(CullSet& cs, uint32_t rec) { // culling_system.cpp:160-190: the device slot becomes a tombstone
	const int32_t entity = cs.recs[rec].entity;
	if (layout_live(cs)) {
		cs.q_id.push_back(PatchId{cs.rec_slot[rec], -1});
		cs.n_tombstones++;
	}
	const uint32_t last = (uint32_t)cs.recs.size() - 1;
	if (rec != last) {
		cs.recs[rec] = cs.recs[last];
		cs.ent_to_rec[cs.recs[rec].entity] = (int32_t)rec;
		if (layout_live(cs)) cs.rec_slot[rec] = cs.rec_slot[last];
	}
	cs.recs.pop_back();
	if (layout_live(cs)) cs.rec_slot.pop_back();
	cs.ent_to_rec[entity] = -1;
}

// remove(entity); add(entity, type, pos, radius) of culling_system.cpp:201-258 when the cell or the big flag changes
void readd_static(CullSet& cs, uint32_t rec, DV3 pos, float radius) {
	const CullRec old = cs.recs[rec];
	if (!layout_live(cs)) {
		cs.recs[rec] = make_cull_rec(old.entity, old.type, pos, radius);
		return;
	}
	remove_static(cs, rec);
	dyn_append(cs, old.entity, old.type, pos, radius, false);
}

void mark_patch(CullSet& cs, uint32_t rec) {
	if (!layout_live(cs)) return;
	const CullRec& r = cs.recs[rec];
	const PatchSphere p{cs.rec_slot[rec], r.rel.x, r.rel.y, r.rel.z, r.radius};
	if (cs.q_sphere_at.size() < cs.n_padded) cs.q_sphere_at.resize(cs.n_padded, ~0u);
	uint32_t& at = cs.q_sphere_at[p.slot];
	if (at != ~0u) { // set twice before the next flush: the last write wins
		cs.q_sphere[at] = p;
		return;
	}
	at = (uint32_t)cs.q_sphere.size();
	cs.q_sphere.push_back(p);
}

// What the reference's stored state (cell, cell-relative fp32 position) means as a world position:
// cell.header.origin + sphere->position (culling_system.cpp:255)
DV3 stored_position(DV3 pos) {
	const IV3 idx = cell_of(pos);
	const DV3 origin = cell_origin(idx);
	return add(origin, to_v3(sub(pos, origin)));
}

void clear_static_queues(CullSet& cs) {
	for (const PatchSphere& p : cs.q_sphere) if (p.slot < cs.q_sphere_at.size()) cs.q_sphere_at[p.slot] = ~0u;
	cs.q_sphere.clear();
	cs.q_id.clear();
}
void clear_dyn_queue(CullSet& cs) {
	for (const PatchDyn& p : cs.q_dyn) cs.q_dyn_at[p.slot] = ~0u;
	cs.q_dyn.clear();
}

DynDeviceView dyn_view(const CullSet& cs) {
	DynDeviceView dd;
	dd.px = cs.dyn_px.p;
	dd.py = cs.dyn_py.p;
	dd.pz = cs.dyn_pz.p;
	dd.radius = cs.dyn_radius.p;
	dd.ids = cs.dyn_ids.p;
	dd.n_padded = cs.dyn_padded;
	return dd;
}

// Ship the queued patch records: one copy into pinned, device-visible host memory and ONE kernel that reads the records from there
// (a frame's records are tens of KB; the H2D copy call alone cost more host time than the 2000 mirror updates it carried) - no host
// wait: the two staging halves alternate, a half is rewritten two flushes after the kernel that read it was enqueued.
// Host -> device copies of the asynchronous compaction's worker go through two pinned staging buffers, chunk by chunk: a
// hipMemcpyAsync from PAGEABLE memory is staged by the runtime in a way that held up the context's own stream for the length of the
// whole upload (measured: one 32 ms frame while 400 MB of a re-sorted 12 M-entity set went up; tools/scratch/async_stream_probe.py).
struct PinnedUploader {
	static constexpr size_t CHUNK = 4u << 20;
	void* buf[2] = {nullptr, nullptr};
	hipEvent_t ev[2] = {nullptr, nullptr};
	bool used[2] = {false, false};
	int k = 0;
	hipStream_t stream = nullptr;
	hipError_t init(hipStream_t s) {
		stream = s;
		for (int i = 0; i < 2; ++i) {
			hipError_t e = hipHostMalloc(&buf[i], CHUNK, hipHostMallocDefault);
			if (e != hipSuccess) return e;
			e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
			if (e != hipSuccess) return e;
		}
		return hipSuccess;
	}
	void destroy() {
		for (int i = 0; i < 2; ++i) {
			if (buf[i]) (void)hipHostFree(buf[i]);
			if (ev[i]) (void)hipEventDestroy(ev[i]);
			buf[i] = nullptr;
			ev[i] = nullptr;
		}
	}
	hipError_t copy(void* dst, const void* src, size_t bytes) {
		for (size_t off = 0; off < bytes; off += CHUNK) {
			const size_t n = std::min(CHUNK, bytes - off);
			if (used[k]) {
				hipError_t e = hipEventSynchronize(ev[k]);
				if (e != hipSuccess) return e;
			}
			memcpy(buf[k], (const char*)src + off, n);
			hipError_t e = hipMemcpyAsync((char*)dst + off, buf[k], n, hipMemcpyHostToDevice, stream);
			if (e != hipSuccess) return e;
			e = hipEventRecord(ev[k], stream);
			if (e != hipSuccess) return e;
			used[k] = true;
			k ^= 1;
		}
		return hipSuccess;
	}
};
inline hipError_t upload(PinnedUploader* up, void* dst, const void* src, size_t bytes, hipStream_t stream) {
	return up ? up->copy(dst, src, bytes) : hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
}

int apply_patches_on(LmxContext* ctx, CullSet& cs, hipStream_t stream, bool profile) {
	const size_t n_ps = cs.q_sphere.size(), n_pi = cs.q_id.size(), n_pd = cs.q_dyn.size();
	if (!(n_ps + n_pi + n_pd)) return LMX_OK;
	const size_t b_ps = n_ps * sizeof(PatchSphere), b_pi = n_pi * sizeof(PatchId), b_pd = n_pd * sizeof(PatchDyn);
	const size_t o_pd = 0, o_ps = (o_pd + b_pd + 15) & ~(size_t)15, o_pi = (o_ps + b_ps + 15) & ~(size_t)15; // PatchDyn needs 8-byte alignment
	const size_t total = o_pi + b_pi;
	PatchStaging& st = cs.staging;
	const uint32_t k = st.next;
	st.next ^= 1u;
	if (!st.done[k]) LMX_HIP(ctx, hipEventCreateWithFlags(&st.done[k], hipEventDisableTiming));
	else LMX_HIP(ctx, hipEventSynchronize(st.done[k])); // the kernel that last read this half (two flushes ago) has long finished
	if (st.cap[k] < total) {
		if (st.host[k]) LMX_HIP(ctx, hipHostFree(st.host[k]));
		st.host[k] = nullptr;
		st.dev[k] = nullptr;
		st.cap[k] = 0;
		const size_t want = std::max<size_t>(total * 2, 1u << 16);
		LMX_HIP(ctx, hipHostMalloc(&st.host[k], want, hipHostMallocMapped));
		LMX_HIP(ctx, hipHostGetDevicePointer(&st.dev[k], st.host[k], 0));
		st.cap[k] = want;
	}
	char* h = (char*)st.host[k];
	if (b_pd) memcpy(h + o_pd, cs.q_dyn.data(), b_pd);
	if (b_ps) memcpy(h + o_ps, cs.q_sphere.data(), b_ps);
	if (b_pi) memcpy(h + o_pi, cs.q_id.data(), b_pi);
	const char* d = (const char*)st.dev[k];
	TileBox* const boxes[3] = {cs.tile_box[0].p, cs.tile_box[1].p, cs.tile_box[2].p};
	if (&cs == static_cast<CullSet*>(&ctx->cull) && n_pi) { // a slot about to become a tombstone: its per-slot sort-key state follows the entity
		if (int rc = keys_before_tombstones(ctx, (const PatchId*)(d + o_pi), (uint32_t)n_pi)) return rc;
	}
	if (profile) { // (the profiler's event pool belongs to the update thread: the worker of the asynchronous compaction passes false)
		ProfScope ps(ctx, LMX_K_CULL_PATCH);
		LMX_HIP(ctx, launch_apply_patches(stream, cs.spheres.p, cs.ids.p, boxes, dyn_view(cs), (const PatchSphere*)(d + o_ps), (uint32_t)n_ps,
			(const PatchId*)(d + o_pi), (uint32_t)n_pi, (const PatchDyn*)(d + o_pd), (uint32_t)n_pd));
	} else {
		LMX_HIP(ctx, launch_apply_patches(stream, cs.spheres.p, cs.ids.p, boxes, dyn_view(cs), (const PatchSphere*)(d + o_ps), (uint32_t)n_ps,
			(const PatchId*)(d + o_pi), (uint32_t)n_pi, (const PatchDyn*)(d + o_pd), (uint32_t)n_pd));
	}
	LMX_HIP(ctx, hipEventRecord(st.done[k], stream));
	clear_static_queues(cs);
	clear_dyn_queue(cs);
	return LMX_OK;
}
int apply_patches(LmxContext* ctx) { return apply_patches_on(ctx, ctx->cull, ctx->stream, true); }

// Rebuild the static device layout from the host mirror (lmx_cull_layout.h) and upload it.
int rebuild_static_on(LmxContext* ctx, CullSet& cs, hipStream_t stream, uint32_t overflow_reserve, PinnedUploader* up = nullptr) {
	CullLayout lay;
	if (!build_cull_layout(cs.recs, lay)) return fail(ctx, LMX_ERR_CAPACITY, "too many spheres (%zu)", cs.recs.size());
	const size_t n_padded = lay.n_padded;
	const size_t n_chunks = n_padded / CHUNK;
	for (int t = 0; t < MAX_TYPES; ++t) {
		cs.tt.ent_start[t] = lay.ent_start[t];
		cs.tt.ent_end[t] = lay.ent_end[t];
	}
	cs.n_padded = (uint32_t)n_padded;
	cs.n_cells = (uint32_t)lay.cells.size();
	cs.n_dead_cells = lay.n_dead_cells;
	for (int k = 0; k < 3; ++k) cs.max_tile_cells[k] = lay.max_tile_cells[k];
	for (int a = 0; a < 3; ++a) { cs.scene_lo[a] = INFINITY; cs.scene_hi[a] = -INFINITY; }
	size_t big_tiles = 0, live_tiles = 0;
	for (const TileBox& b : lay.tile_box[0]) { // world-space box of the occupied cells (static set)
		if (b.flags & TILE_EMPTY) continue;
		++live_tiles;
		if (b.flags & TILE_HAS_BIG) ++big_tiles;
		for (int a = 0; a < 3; ++a) {
			cs.scene_lo[a] = std::min(cs.scene_lo[a], (double)CELL_SIZE * b.lo[a]);
			cs.scene_hi[a] = std::max(cs.scene_hi[a], (double)CELL_SIZE * b.hi[a] + (double)CELL_SIZE);
		}
	}
	cs.big_tile_fraction = live_tiles ? (double)big_tiles / (double)live_tiles : 0.0;
	// the buffers below may be reallocated; the copies are ordered on `stream` - the context's for the synchronous path, the worker's own
	// for the asynchronous compaction (a plain hipMemcpy would go through the null stream and order itself against every blocking stream)
	LMX_HIP(ctx, hipStreamSynchronize(stream));
	LMX_HIP(ctx, cs.spheres.reserve(std::max<size_t>(n_padded, 1)));
	LMX_HIP(ctx, cs.ids.reserve(std::max<size_t>(n_padded, 1)));
	LMX_HIP(ctx, cs.hdr.reserve(std::max<size_t>(n_chunks, 1)));
	if (n_padded) {
		LMX_HIP(ctx, upload(up, cs.spheres.p, lay.spheres.data(), n_padded * sizeof(float4), stream));
		LMX_HIP(ctx, upload(up, cs.ids.p, lay.ids.data(), n_padded * sizeof(int32_t), stream));
		LMX_HIP(ctx, upload(up, cs.hdr.p, lay.hdr.data(), n_chunks * sizeof(ChunkHdr), stream));
	}
	// Cell keys travel in 8 bytes where every tile's cells lie within 65535 cell indices of its box's low corner (any scene that is not a handful
	// of entities millions of units apart): offsets against TileBox::lo + the two flags the kernel reads (PackedCellKey). 4096-sphere tiles (k = 0)
	// are walked by no kernel since round 6: their keys are not uploaded at all (their boxes are: k_apply_patches clears TILE_DENSE in all three).
	bool packable = getenv("LMX_CULL_WIDE_KEYS") == nullptr;
	for (int k = 1; k < 3 && packable; ++k) {
		const size_t cap = lay.tile_cap[k];
		for (size_t ti = 0; ti < lay.tile_box[k].size() && packable; ++ti) {
			const TileBox& b = lay.tile_box[k][ti];
			if (b.flags & TILE_EMPTY) continue;
			for (int a = 0; a < 3; ++a) packable = packable && (int64_t)b.hi[a] - (int64_t)b.lo[a] <= 65535;
		}
		(void)cap;
	}
	cs.keys_packed = packable;
	std::vector<PackedCellKey> packed_k[3]; // (alive until the synchronize below: the copies are asynchronous)
	for (int k = 0; k < 3; ++k) {
		cs.tile_cap[k] = lay.tile_cap[k];
		const bool keys_used = k != 0;
		const size_t key_bytes = !keys_used ? 0 : lay.tile_cells[k].size() * (packable ? sizeof(PackedCellKey) : sizeof(CellKey));
		LMX_HIP(ctx, cs.tile_cells[k].reserve(std::max<size_t>((key_bytes + sizeof(CellKey) - 1) / sizeof(CellKey), 1)));
		LMX_HIP(ctx, cs.tile_tab[k].reserve(std::max<size_t>(lay.tile_tab[k].size(), 1)));
		LMX_HIP(ctx, cs.tile_box[k].reserve(std::max<size_t>(lay.tile_box[k].size(), 1)));
		if (!lay.tile_cells[k].empty()) {
			if (keys_used && packable) {
				const size_t cap = lay.tile_cap[k];
				std::vector<PackedCellKey>& packed = packed_k[k];
				packed.resize(lay.tile_cells[k].size());
				parallel_ranges(lay.tile_box[k].size(), [&](size_t tb, size_t te) {
					for (size_t ti = tb; ti < te; ++ti) {
						const TileBox& b = lay.tile_box[k][ti];
						for (size_t j = 0; j < cap; ++j) {
							const LayoutCell& c = lay.tile_cells[k][ti * cap + j];
							PackedCellKey pk{0u, PACKED_CELL_DEAD};
							if (!(c.meta & LAYOUT_CELL_DEAD))
								pk = PackedCellKey{(uint32_t)(c.ix - b.lo[0]) | ((uint32_t)(c.iy - b.lo[1]) << 16), (uint32_t)(c.iz - b.lo[2]) | ((c.meta & 0x100u) ? PACKED_CELL_BIG : 0u)};
							packed[ti * cap + j] = pk;
						}
					}
				});
				LMX_HIP(ctx, upload(up, cs.tile_cells[k].p, packed.data(), key_bytes, stream));
			} else if (keys_used) {
				LMX_HIP(ctx, upload(up, cs.tile_cells[k].p, lay.tile_cells[k].data(), key_bytes, stream));
			}
			LMX_HIP(ctx, upload(up, cs.tile_tab[k].p, lay.tile_tab[k].data(), lay.tile_tab[k].size() * sizeof(uint32_t), stream));
			LMX_HIP(ctx, upload(up, cs.tile_box[k].p, lay.tile_box[k].data(), lay.tile_box[k].size() * sizeof(TileBox), stream));
		}
	}
	LMX_HIP(ctx, hipStreamSynchronize(stream)); // `lay` is about to go
	cs.rec_slot.swap(lay.rec_slot);
	cs.block_live.swap(lay.block_live);
	cs.structure_dirty = false;
	cs.built = true;
	if (overflow_reserve) cs.dyn_layout_dirty = true; // the reserve follows the new static set's type shares
	cs.n_tombstones = 0;
	clear_static_queues(cs);
	return LMX_OK;
}
std::atomic<uint64_t> g_layout_generation{1};
int rebuild_static(LmxContext* ctx) {
	if (int rc = keys_before_layout_change(ctx)) return rc; // per-slot state of the sort-key tables goes back to its entity-indexed home first
	if (int rc = rebuild_static_on(ctx, ctx->cull, ctx->stream, ctx->cull.overflow_reserve)) return rc;
	ctx->cull.layout_generation = g_layout_generation++;
	return LMX_OK;
}

// (Re)assign the device slots of the dynamic set: one region per type, padded to DYN_ALIGN, with room to grow
// (region = 1.5 x live + one tile), and upload everything.
int rebuild_dynamic_on(LmxContext* ctx, CullSet& cs, hipStream_t stream, uint32_t overflow_reserve, PinnedUploader* up = nullptr) {
	const size_t n = cs.dyn.size();
	size_t count_by_type[MAX_TYPES] = {};
	for (const DynRec& r : cs.dyn) count_by_type[r.type]++;
	// LMX_CULL_OPT_OVERFLOW_RESERVE: room for that many more entities, shared out over the renderable types by their share of the
	// static set, so that adds / re-celling sets between compactions take free slots and never trigger this function again
	size_t static_by_type[MAX_TYPES] = {}, static_total = 0;
	for (int t = 0; t < MAX_TYPES; ++t) {
		static_by_type[t] = cs.tt.ent_end[t] - cs.tt.ent_start[t];
		static_total += static_by_type[t];
	}
	size_t padded = 0;
	for (int t = 0; t < MAX_TYPES; ++t) {
		cs.dyn_tt.ent_start[t] = (uint32_t)padded;
		size_t want = count_by_type[t] ? count_by_type[t] + count_by_type[t] / 2 + DYN_ALIGN : 0;
		if (overflow_reserve && static_by_type[t]) want = std::max<size_t>(want, count_by_type[t] + (size_t)((double)overflow_reserve * static_by_type[t] / static_total) + DYN_ALIGN);
		if (want) padded += want / DYN_ALIGN * DYN_ALIGN;
		cs.dyn_tt.ent_end[t] = (uint32_t)padded;
		cs.dyn_free[t].clear();
	}
	if (padded > 0x7fffffffull) return fail(ctx, LMX_ERR_CAPACITY, "too many dynamic spheres (%zu)", n);
	std::vector<double> px(padded, 0.0), py(padded, 0.0), pz(padded, 0.0);
	std::vector<float> radius(padded, 0.f);
	std::vector<int32_t> ids(padded, -1);
	size_t cursor[MAX_TYPES];
	for (int t = 0; t < MAX_TYPES; ++t) cursor[t] = cs.dyn_tt.ent_start[t];
	for (size_t i = 0; i < n; ++i) {
		DynRec& r = cs.dyn[i];
		const size_t s = cursor[r.type]++;
		r.slot = (uint32_t)s;
		px[s] = r.pos[0];
		py[s] = r.pos[1];
		pz[s] = r.pos[2];
		radius[s] = r.radius;
		ids[s] = r.entity;
	}
	for (int t = 0; t < MAX_TYPES; ++t) cs.dyn_next[t] = (uint32_t)cursor[t];
	cs.dyn_padded = (uint32_t)padded;
	const size_t cap = std::max<size_t>(padded, 1);
	LMX_HIP(ctx, hipStreamSynchronize(stream));
	LMX_HIP(ctx, cs.dyn_px.reserve(cap));
	LMX_HIP(ctx, cs.dyn_py.reserve(cap));
	LMX_HIP(ctx, cs.dyn_pz.reserve(cap));
	LMX_HIP(ctx, cs.dyn_radius.reserve(cap));
	LMX_HIP(ctx, cs.dyn_ids.reserve(cap));
	if (padded) {
		LMX_HIP(ctx, upload(up, cs.dyn_px.p, px.data(), padded * sizeof(double), stream));
		LMX_HIP(ctx, upload(up, cs.dyn_py.p, py.data(), padded * sizeof(double), stream));
		LMX_HIP(ctx, upload(up, cs.dyn_pz.p, pz.data(), padded * sizeof(double), stream));
		LMX_HIP(ctx, upload(up, cs.dyn_radius.p, radius.data(), padded * sizeof(float), stream));
		LMX_HIP(ctx, upload(up, cs.dyn_ids.p, ids.data(), padded * sizeof(int32_t), stream));
		LMX_HIP(ctx, hipStreamSynchronize(stream)); // the staging vectors are about to go
	}
	cs.dyn_layout_dirty = false;
	cs.q_dyn.clear();
	cs.q_dyn_at.assign(padded, ~0u);
	cs.dyn_generation++;
	return LMX_OK;
}
int rebuild_dynamic(LmxContext* ctx) { return rebuild_dynamic_on(ctx, ctx->cull, ctx->stream, ctx->cull.overflow_reserve); }

// Output shards: per type, the windows of the static set's shards, then those of the dynamic set's. A static window holds
// exactly the live ids of its blocks; a dynamic window the slots of its tiles.
int recompute_out_layout(LmxContext* ctx) {
	CullState& cs = ctx->cull;
	cs.shard_type.clear();
	cs.win_base.clear();
	uint32_t off = 0, max_cap = 0;
	for (int t = 0; t < MAX_TYPES; ++t) {
		cs.type_start[t] = off;
		const uint32_t s_blocks = (cs.tt.ent_end[t] - cs.tt.ent_start[t]) / TILE_ALIGN;
		const uint32_t s_n = std::min(s_blocks, std::max(1u, cs.max_shards));
		cs.tt.shard_first[t] = (uint32_t)cs.win_base.size();
		cs.tt.shard_n[t] = s_n;
		if (s_n) {
			std::vector<uint32_t> cap(s_n, 0u);
			const uint32_t b0 = cs.tt.ent_start[t] / TILE_ALIGN;
			for (uint32_t b = 0; b < s_blocks; ++b) cap[b % s_n] += cs.block_live[b0 + b];
			for (uint32_t k = 0; k < s_n; ++k) {
				cs.win_base.push_back(off);
				cs.shard_type.push_back((uint8_t)t);
				off += cap[k];
				max_cap = std::max(max_cap, cap[k]);
			}
		}
		const uint32_t d_blocks = (cs.dyn_tt.ent_end[t] - cs.dyn_tt.ent_start[t]) / DYN_ALIGN;
		const uint32_t d_n = std::min(d_blocks, DYN_MAX_SHARDS);
		cs.dyn_tt.shard_first[t] = (uint32_t)cs.win_base.size();
		cs.dyn_tt.shard_n[t] = d_n;
		for (uint32_t k = 0; k < d_n; ++k) {
			const uint32_t cap = ((d_blocks - k + d_n - 1) / d_n) * DYN_ALIGN; // blocks k, k + d_n, ...
			cs.win_base.push_back(off);
			cs.shard_type.push_back((uint8_t)t);
			off += cap;
			max_cap = std::max(max_cap, cap);
		}
		cs.type_cap[t] = off - cs.type_start[t];
	}
	cs.out_total = off;
	cs.n_shards = (uint32_t)cs.win_base.size();
	cs.max_shard_cap = max_cap;
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	LMX_HIP(ctx, cs.d_win_base.reserve(std::max<size_t>(cs.n_shards, 1)));
	LMX_HIP(ctx, cs.d_shard_type.reserve(std::max<size_t>(cs.n_shards, 1)));
	LMX_HIP(ctx, cs.d_type_start.reserve(MAX_TYPES));
	if (cs.n_shards) {
		LMX_HIP(ctx, hipMemcpy(cs.d_win_base.p, cs.win_base.data(), cs.n_shards * sizeof(uint32_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(cs.d_shard_type.p, cs.shard_type.data(), cs.n_shards, hipMemcpyHostToDevice));
	}
	LMX_HIP(ctx, hipMemcpy(cs.d_type_start.p, cs.type_start, sizeof(cs.type_start), hipMemcpyHostToDevice));
	for (int k = 0; k < 3; ++k) { // every tile's shard and window start, per tile size (k_cull_tile reads one 8-byte entry instead of deriving them)
		const uint32_t tile = TILE_ALIGN >> k;
		std::vector<uint2> tab(cs.n_padded / tile);
		for (int t = 0; t < MAX_TYPES; ++t) {
			for (uint32_t e = cs.tt.ent_start[t]; e < cs.tt.ent_end[t]; e += tile) {
				const uint32_t shard = cs.tt.shard_first[t] + ((e - cs.tt.ent_start[t]) / TILE_ALIGN) % cs.tt.shard_n[t];
				tab[e / tile] = make_uint2(shard, cs.win_base[shard]);
			}
		}
		LMX_HIP(ctx, cs.d_tile_out[k].reserve(std::max<size_t>(tab.size(), 1)));
		if (!tab.empty()) LMX_HIP(ctx, hipMemcpy(cs.d_tile_out[k].p, tab.data(), tab.size() * sizeof(uint2), hipMemcpyHostToDevice));
	}
	for (CullView& v : cs.views) {
		v.valid = v.finalized = v.consolidated = false;
		v.cnt_words = 0; // counters are re-sized (and zeroed) by the next cull on the view
	}
	return LMX_OK;
}

// Move the unbound part of the dynamic set back into the static mirror (the next rebuild sorts it in).
void fold_overflow(CullSet& cs) {
	for (uint32_t i = (uint32_t)cs.dyn.size(); i-- > 0;) {
		if (cs.dyn[i].bound) continue;
		const DynRec r = cs.dyn[i];
		remove_dynamic(cs, i); // swaps the last record into i: already visited
		if ((size_t)r.entity >= cs.ent_to_rec.size()) cs.ent_to_rec.resize((size_t)r.entity + 1, -1);
		cs.ent_to_rec[r.entity] = (int32_t)cs.recs.size();
		cs.recs.push_back(make_cull_rec(r.entity, r.type, DV3{r.pos[0], r.pos[1], r.pos[2]}, r.radius));
	}
}

// ---- the mutating operations, on a given set ------------------------------------------------------------------------------------
// `cs` is the context's live set for the public entry points and the SHADOW set when the asynchronous compaction replays the
// operation log (replay = true: no device round trips, and LMX_CULL_OPT_DEVICE_OWNS_BOUND - a statement about the live device set at
// the time of the call - is not consulted: only operations that took effect are logged). `*effective` = the set changed.
static int cull_add_impl(LmxContext* ctx, CullSet& cs, int32_t entity, uint8_t type, const double pos[3], float radius) { // culling_system.cpp:131-157
	if (entity < 0 || !pos) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad entity/pos");
	if (type >= MAX_TYPES) return fail(ctx, LMX_ERR_CAPACITY, "type %u >= LMX_MAX_TYPES", type);
	uint32_t idx;
	if (locate(cs, entity, &idx) != Where::NONE) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d already added", entity);
	if (layout_live(cs)) {
		dyn_append(cs, entity, type, DV3{pos[0], pos[1], pos[2]}, radius, false); // sorted in by the next compaction
		return LMX_OK;
	}
	if ((size_t)entity >= cs.ent_to_rec.size()) cs.ent_to_rec.resize((size_t)entity + 1, -1);
	cs.ent_to_rec[entity] = (int32_t)cs.recs.size();
	cs.recs.push_back(make_cull_rec(entity, type, DV3{pos[0], pos[1], pos[2]}, radius));
	cs.structure_dirty = true;
	return LMX_OK;
}

static int cull_remove_impl(CullSet& cs, int32_t entity, bool* effective) { // culling_system.cpp:160-190 (unknown entities are ignored, :162-165)
	uint32_t idx;
	*effective = true;
	switch (locate(cs, entity, &idx)) {
		case Where::STATIC: remove_static(cs, idx); break;
		case Where::DYNAMIC: remove_dynamic(cs, idx); break;
		case Where::NONE: *effective = false; break;
	}
	return LMX_OK;
}

static int cull_set_impl(LmxContext* ctx, CullSet& cs, bool device_owns_bound, int32_t entity, const double pos[3], float radius, bool* effective) { // culling_system.cpp:225-242
	*effective = false;
	if (!pos) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null pos");
	uint32_t idx;
	const DV3 p = DV3{pos[0], pos[1], pos[2]};
	switch (locate(cs, entity, &idx)) {
		case Where::STATIC: {
			CullRec& r = cs.recs[idx];
			const IV3 c = cell_of(p);
			if (r.big == is_big_radius(radius) && c.x == r.cell.x && c.y == r.cell.y && c.z == r.cell.z) {
				r.radius = radius;
				r.rel = to_v3(sub(p, cell_origin(r.cell)));
				mark_patch(cs, idx);
			} else {
				readd_static(cs, idx, p, radius);
			}
			*effective = true;
			return LMX_OK;
		}
		case Where::DYNAMIC: {
			DynRec& r = cs.dyn[idx];
			if (r.bound && device_owns_bound) return LMX_OK; // lmx_world_propagate already refreshed this sphere on the device
			r.pos[0] = p.x; r.pos[1] = p.y; r.pos[2] = p.z;
			r.radius = radius;
			queue_dyn_patch(cs, r, true);
			*effective = true;
			return LMX_OK;
		}
		case Where::NONE: break;
	}
	return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d is not in the culling system", entity);
}

static int cull_set_position_impl(LmxContext* ctx, CullSet& cs, bool device_owns_bound, bool replay, int32_t entity, const double pos[3], bool* effective) { // culling_system.cpp:201-217
	*effective = false;
	if (!pos) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null pos");
	uint32_t idx;
	const DV3 p = DV3{pos[0], pos[1], pos[2]};
	switch (locate(cs, entity, &idx)) {
		case Where::STATIC: {
			CullRec& r = cs.recs[idx];
			const IV3 c = cell_of(p);
			if (c.x == r.cell.x && c.y == r.cell.y && c.z == r.cell.z) {
				r.rel = to_v3(sub(p, cell_origin(r.cell)));
				mark_patch(cs, idx);
			} else {
				readd_static(cs, idx, p, r.radius);
			}
			*effective = true;
			return LMX_OK;
		}
		case Where::DYNAMIC: {
			if (cs.dyn[idx].bound) { // the radius the patch carries must be the one the device last computed
				if (device_owns_bound) return LMX_OK;
				if (!replay) { // (the shadow set's copy of a bound sphere is overwritten from the live device set when the sets trade places)
					if (int rc = cull_dyn_sync_mirror(ctx)) return rc;
				}
			}
			DynRec& r = cs.dyn[idx];
			r.pos[0] = p.x; r.pos[1] = p.y; r.pos[2] = p.z;
			queue_dyn_patch(cs, r, true);
			*effective = true;
			return LMX_OK;
		}
		case Where::NONE: break;
	}
	return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d is not in the culling system", entity);
}

static int cull_set_radius_impl(LmxContext* ctx, CullSet& cs, bool device_owns_bound, bool replay, int32_t entity, float radius, bool* effective) { // culling_system.cpp:244-260
	*effective = false;
	uint32_t idx;
	switch (locate(cs, entity, &idx)) {
		case Where::STATIC: {
			CullRec& r = cs.recs[idx];
			if (r.big == is_big_radius(radius)) {
				r.radius = radius;
				mark_patch(cs, idx);
			} else {
				readd_static(cs, idx, add(cell_origin(r.cell), r.rel), radius); // pos = cell.header.origin + sphere->position
			}
			*effective = true;
			return LMX_OK;
		}
		case Where::DYNAMIC: {
			if (cs.dyn[idx].bound) {
				if (device_owns_bound) return LMX_OK;
				if (!replay) {
					if (int rc = cull_dyn_sync_mirror(ctx)) return rc;
				}
			}
			DynRec& r = cs.dyn[idx];
			if (is_big_radius(r.radius) != is_big_radius(radius)) {
				// the reference re-adds at origin + fp32 relative position, which loses the low bits of the position
				const DV3 p = stored_position(DV3{r.pos[0], r.pos[1], r.pos[2]});
				r.pos[0] = p.x; r.pos[1] = p.y; r.pos[2] = p.z;
			}
			r.radius = radius;
			queue_dyn_patch(cs, r, true);
			*effective = true;
			return LMX_OK;
		}
		case Where::NONE: break;
	}
	return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d is not in the culling system", entity);
}


// lmx_world_bind_culling / unbind on a given set (see cull_make_dynamic)
static bool make_dynamic_impl(CullSet& cs, int32_t entity) {
	uint32_t idx;
	const Where w = locate(cs, entity, &idx);
	if (w == Where::DYNAMIC) {
		if (!cs.dyn[idx].bound) {
			cs.dyn[idx].bound = true;
			cs.n_unbound--;
		}
		return true;
	}
	if (w != Where::STATIC) return false;
	const CullRec r = cs.recs[idx];
	const DV3 pos = add(cell_origin(r.cell), r.rel);
	remove_static(cs, idx);
	dyn_append(cs, entity, r.type, pos, r.radius, true);
	return true;
}
static void unbind_impl(CullSet& cs, int32_t entity) {
	uint32_t idx;
	if (locate(cs, entity, &idx) == Where::DYNAMIC && cs.dyn[idx].bound) {
		cs.dyn[idx].bound = false;
		cs.n_unbound++;
	}
}


// ---- asynchronous compaction (LMX_CULL_OPT_ASYNC_COMPACTION) ----------------------------------------------------------------------
// The re-sort of the static set is the one O(n) step of the culling system (0.4-0.5 s at 10 M entities). With this option it runs on a
// worker thread, on a SECOND complete copy of the sets (host mirror + device arrays): the shadow set.
//   * Every effective add / remove / set* / bind of the live set is also appended to an operation log (40 bytes, no lock: the log is
//     handed to the worker once per flush).
//   * A job (requested by lmx_cull_flush when the live set's overflow / tombstones pass the usual thresholds): the worker replays the
//     log onto the shadow set's mirror, folds its overflow into its static mirror, builds and uploads a fresh layout on its own
//     stream, then keeps replaying newer log segments - now as O(1) patches on the shadow's device arrays - until a segment is short.
//   * The swap, on the update thread inside a flush: the last few operations are replayed, the two sets trade places (O(1): vectors and
//     device buffers swap storage), the spheres of hierarchy-bound entities - refreshed on the device, not by the host - are copied
//     device to device from the old set, and the output shards are re-derived. The old live set, which has seen every operation, is
//     the next job's shadow.
// What a frame pays: the log appends, and one swap of a few hundred replayed operations per compaction.
int async_replay(LmxContext* ctx, CullSet& cs, const CullOp* ops, size_t n) {
	for (size_t i = 0; i < n; ++i) {
		const CullOp& o = ops[i];
		bool eff;
		int rc = LMX_OK;
		switch (o.op) {
			case OP_ADD: rc = cull_add_impl(ctx, cs, o.entity, o.type, o.pos, o.radius); break;
			case OP_REMOVE: rc = cull_remove_impl(cs, o.entity, &eff); break;
			case OP_SET: rc = cull_set_impl(ctx, cs, false, o.entity, o.pos, o.radius, &eff); break;
			case OP_SET_POS: rc = cull_set_position_impl(ctx, cs, false, true, o.entity, o.pos, &eff); break;
			case OP_SET_RADIUS: rc = cull_set_radius_impl(ctx, cs, false, true, o.entity, o.radius, &eff); break;
			case OP_BIND: rc = make_dynamic_impl(cs, o.entity) ? LMX_OK : LMX_ERR_INVALID_ARGUMENT; break;
			case OP_UNBIND: unbind_impl(cs, o.entity); break;
			default: rc = LMX_ERR_INVALID_ARGUMENT;
		}
		if (rc != LMX_OK) return rc; // the shadow set has drifted from the live one: the job fails, the caller falls back to the synchronous path
	}
	return LMX_OK;
}


constexpr size_t ASYNC_LOG_LIMIT = 1u << 20;     // operations (40 MB) the log may hold with no job due before a drain job brings the shadow up to date
constexpr size_t ASYNC_SHORT_SEGMENT = 4096; // a log segment this short ends the catch-up: the swap replays what arrived meanwhile

int async_job(LmxContext* ctx, CullAsync& a) {
	CullSet& sh = a.shadow;
	std::vector<CullOp> seg;
	{
		std::lock_guard<std::mutex> g(a.mu);
		seg.swap(a.log_shared);
	}
	// 1. mirror-only replay (no patches: the layout is about to be rebuilt), fold, rebuild both sets of the shadow
	sh.structure_dirty = true;
	sh.dyn_layout_dirty = true;
	clear_static_queues(sh);
	sh.q_dyn.clear();
	if (int rc = async_replay(ctx, sh, seg.data(), seg.size())) return rc;
	if (a.drain_only) return LMX_OK; // the shadow's mirror is current again; no re-sort was due
	fold_overflow(sh);
	if (a.swapped_pending) { // kernels enqueued on the context's stream before the last swap may still read what is now the shadow set
		LMX_HIP(ctx, hipStreamWaitEvent(a.stream, a.swapped, 0));
		a.swapped_pending = false;
	}
	if (int rc = rebuild_static_on(ctx, sh, a.stream, a.overflow_reserve, a.uploader)) return rc;
	if (int rc = rebuild_dynamic_on(ctx, sh, a.stream, a.overflow_reserve, a.uploader)) return rc;
	// 2. catch up: newer segments as O(1) patches on the shadow's own device arrays
	for (int round = 0; round < 64; ++round) {
		seg.clear();
		{
			std::lock_guard<std::mutex> g(a.mu);
			seg.swap(a.log_shared);
		}
		if (int rc = async_replay(ctx, sh, seg.data(), seg.size())) return rc;
		if (sh.dyn_layout_dirty) { // a type's region of the shadow's dynamic set ran full during the replay
			if (int rc = rebuild_dynamic_on(ctx, sh, a.stream, a.overflow_reserve, a.uploader)) return rc;
		}
		if (int rc = apply_patches_on(ctx, sh, a.stream, false)) return rc;
		if (seg.size() < ASYNC_SHORT_SEGMENT) break;
	}
	// 3. entity -> dynamic slot of the shadow set, for the device-to-device copy of bound spheres at the swap
	a.n_new_slot = 0;
	bool any_bound = false;
	for (const DynRec& r : sh.dyn) any_bound = any_bound || r.bound;
	if (any_bound) {
		std::vector<int32_t> slot(sh.ent_to_dyn.size(), -1);
		for (const DynRec& r : sh.dyn)
			if (r.slot != DYN_NO_SLOT) slot[r.entity] = (int32_t)r.slot;
		LMX_HIP(ctx, a.d_new_slot.reserve(std::max<size_t>(slot.size(), 1)));
		if (!slot.empty()) LMX_HIP(ctx, upload(a.uploader, a.d_new_slot.p, slot.data(), slot.size() * sizeof(int32_t), a.stream));
		LMX_HIP(ctx, hipStreamSynchronize(a.stream));
		a.n_new_slot = (uint32_t)slot.size();
	}
	LMX_HIP(ctx, hipStreamSynchronize(a.stream));
	return LMX_OK;
}

void async_worker(LmxContext* ctx, CullAsync* a) {
	(void)hipSetDevice(ctx->device);
	t_layout_thread_cap = 8; // a background re-sort: a quarter of what a synchronous build takes
	t_fail_sink = &a->error; // the worker's errors must not land in LmxContext::error (the update thread may be writing it): fail() honours this
	for (;;) {
		{
			std::unique_lock<std::mutex> g(a->mu);
			a->cv.wait(g, [&] { return a->state == CullAsync::REQUESTED || a->state == CullAsync::QUIT; });
			if (a->state == CullAsync::QUIT) return;
			a->state = CullAsync::RUNNING;
		}
		a->error.clear();
		const int rc = async_job(ctx, *a);
		std::lock_guard<std::mutex> g(a->mu);
		if (a->state == CullAsync::QUIT) return;
		if (rc != LMX_OK) a->state = CullAsync::FAILED;
		else if (a->drain_only) {
			a->state = CullAsync::IDLE;
			a->drains++;
			a->cv_idle.notify_all();
			continue;
		} else a->state = CullAsync::READY;
		a->jobs_done++;
		a->cv_idle.notify_all(); // (async_wait_idle)
	}
}

// update thread: hand the operations of this flush to the log the worker reads
void async_publish_log(CullAsync& a) {
	if (a.log_local.empty()) return;
	std::lock_guard<std::mutex> g(a.mu);
	a.log_shared.insert(a.log_shared.end(), a.log_local.begin(), a.log_local.end());
	a.log_local.clear();
}

CullAsync::State async_state(CullAsync& a) {
	std::lock_guard<std::mutex> g(a.mu);
	return a.state;
}

// The shadow set := a copy of the live set's host mirror (O(n), once: when the option is switched on, after lmx_cull_build and after
// a synchronous compaction); its device arrays are rebuilt by the first job anyway.
void async_reseed(CullState& cs) {
	CullAsync& a = *cs.async;
	CullSet& sh = a.shadow;
	sh.recs = cs.recs;
	sh.ent_to_rec = cs.ent_to_rec;
	sh.rec_slot.clear();
	sh.dyn = cs.dyn;
	sh.ent_to_dyn = cs.ent_to_dyn;
	sh.n_unbound = cs.n_unbound;
	sh.built = cs.built;
	sh.structure_dirty = true;
	sh.dyn_layout_dirty = true;
	sh.n_tombstones = 0;
	clear_static_queues(sh);
	sh.q_dyn.clear();
	a.log_local.clear();
	std::lock_guard<std::mutex> g(a.mu);
	a.log_shared.clear();
}

void async_wait_idle(CullAsync& a) { // update thread: let a running job finish (its result is discarded by the caller)
	// (a sleep on the worker's own condition variable, not a yield spin: the caller holds the context's lock for as long as the job
	// runs - 0.5 s at 10 M entities - and should not burn a core next to the worker meanwhile)
	std::unique_lock<std::mutex> g(a.mu);
	a.cv_idle.wait(g, [&] { return a.state != CullAsync::REQUESTED && a.state != CullAsync::RUNNING; });
}

int recompute_out_layout(LmxContext* ctx);

// update thread, inside a flush, the worker's job is READY: the sets trade places
int async_swap(LmxContext* ctx) {
	CullState& cs = ctx->cull;
	CullAsync& a = *cs.async;
	CullSet& sh = a.shadow;
	// what happened since the worker's last segment (normally a frame or two of operations)
	std::vector<CullOp> tail;
	{
		std::lock_guard<std::mutex> g(a.mu);
		tail.swap(a.log_shared);
	}
	tail.insert(tail.end(), a.log_local.begin(), a.log_local.end());
	a.log_local.clear();
	a.ops_replayed_at_swap += tail.size();
	if (int rc = async_replay(ctx, sh, tail.data(), tail.size())) return rc;
	if (sh.dyn_layout_dirty) {
		if (int rc = rebuild_dynamic_on(ctx, sh, ctx->stream, cs.overflow_reserve)) return rc;
		a.n_new_slot = 0; // slots moved: fall back to the host copy of the bound spheres below
	}
	if (int rc = apply_patches(ctx)) return rc;                                // the live set's pending patches (its device ids / positions are read below)
	if (int rc = apply_patches_on(ctx, sh, ctx->stream, true)) return rc;      // ordered behind the worker's uploads: its stream was synchronised before READY
	// spheres of hierarchy-bound entities live on the device (k_sphere_refresh): old set -> new set, slot by slot through the entity id
	bool any_bound = false;
	for (const DynRec& r : sh.dyn) {
		if (r.bound) {
			any_bound = true;
			break;
		}
	}
	if (any_bound) {
		if (a.n_new_slot && cs.dyn_padded) {
			LMX_HIP(ctx, launch_dyn_carry_over(ctx->stream, dyn_view(cs), dyn_view(sh), a.d_new_slot.p, a.n_new_slot));
		} else {
			if (int rc = cull_dyn_sync_mirror(ctx)) return rc; // (rare path: O(bound entities) on the host)
			for (DynRec& r : sh.dyn) {
				uint32_t idx;
				if (!r.bound || locate(cs, r.entity, &idx) != Where::DYNAMIC) continue;
				const DynRec& o = cs.dyn[idx];
				r.pos[0] = o.pos[0]; r.pos[1] = o.pos[1]; r.pos[2] = o.pos[2];
				r.radius = o.radius;
				queue_dyn_patch(sh, r, true);
			}
			if (int rc = apply_patches_on(ctx, sh, ctx->stream, true)) return rc;
		}
	}
	if (int rc = keys_before_layout_change(ctx)) return rc; // (reads the OLD set's slot -> id array)
	const uint64_t generation = std::max(cs.dyn_generation, sh.dyn_generation) + 1;
	static_cast<CullSet&>(cs).swap_with(sh);
	cs.layout_generation = g_layout_generation++;
	cs.dyn_generation = generation; // the world's binding tables (slots of bound entities) are re-derived at the next propagation
	sh.dyn_generation = generation;
	if (any_bound) cs.dyn_mirror_stale = true; // the host copies of bound spheres are older than the device's
	// the old live set is the next shadow: it has seen every operation; its device arrays are dead weight until the next job rebuilds them
	sh.structure_dirty = true;
	sh.dyn_layout_dirty = true;
	clear_static_queues(sh);
	sh.q_dyn.clear();
	sh.q_sphere_at.clear();
	if (!a.swapped) LMX_HIP(ctx, hipEventCreateWithFlags(&a.swapped, hipEventDisableTiming));
	LMX_HIP(ctx, hipEventRecord(a.swapped, ctx->stream));
	a.swapped_pending = true;
	a.swaps++;
	{
		std::lock_guard<std::mutex> g(a.mu);
		a.state = CullAsync::IDLE;
	}
	return recompute_out_layout(ctx);
}

bool wants_compaction(const CullState& cs);

// update thread, every flush of a live layout while the option is on. Returns LMX_OK; *handled = the sets were swapped.
int async_poll(LmxContext* ctx, bool* swapped) {
	CullState& cs = ctx->cull;
	CullAsync& a = *cs.async;
	*swapped = false;
	async_publish_log(a);
	const CullAsync::State st = async_state(a);
	if (st == CullAsync::READY) {
		if (int rc = async_swap(ctx)) { // could not adopt the shadow set: start over from a copy of the live one
			async_reseed(cs);
			std::lock_guard<std::mutex> g(a.mu);
			a.state = CullAsync::IDLE;
			return rc;
		}
		*swapped = true;
		return LMX_OK;
	}
	if (st == CullAsync::FAILED) {
		fail(ctx, LMX_ERR_HIP, "asynchronous compaction failed: %s", a.error.c_str());
		async_reseed(cs);
		std::lock_guard<std::mutex> g(a.mu);
		a.state = CullAsync::IDLE;
		return LMX_OK; // the live set is intact; the next request starts from a fresh copy
	}
	if (st == CullAsync::IDLE) {
		const bool resort = cs.auto_compaction && wants_compaction(cs);
		// in-cell moves patch the sorted set in place and never make a re-sort due: the log must not grow without bound meanwhile
		size_t backlog;
		{
			std::lock_guard<std::mutex> g(a.mu);
			backlog = a.log_shared.size();
		}
		const bool drain = !resort && backlog > std::max<size_t>(ASYNC_LOG_LIMIT, cs.recs.size() / 4);
		if (resort || drain) {
			a.overflow_reserve = cs.overflow_reserve;
			a.drain_only = drain;
			std::lock_guard<std::mutex> g(a.mu);
			a.state = CullAsync::REQUESTED;
			a.cv.notify_one();
		}
	}
	return LMX_OK;
}

int async_enable(LmxContext* ctx) {
	CullState& cs = ctx->cull;
	if (cs.async) return LMX_OK;
	CullAsync* a = new CullAsync;
	hipError_t e = hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking);
	if (e != hipSuccess) {
		delete a;
		return fail(ctx, LMX_ERR_HIP, "hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
	}
	a->uploader = new PinnedUploader;
	e = a->uploader->init(a->stream);
	if (e != hipSuccess) {
		a->uploader->destroy();
		delete a->uploader;
		(void)hipStreamDestroy(a->stream);
		delete a;
		return fail(ctx, LMX_ERR_HIP, "pinned staging for the asynchronous compaction: %s", hipGetErrorString(e));
	}
	cs.async = a;
	async_reseed(cs);
	a->worker = std::thread(async_worker, ctx, a);
	return LMX_OK;
}

void async_disable(CullState& cs) {
	CullAsync* a = cs.async;
	if (!a) return;
	async_wait_idle(*a);
	{
		std::lock_guard<std::mutex> g(a->mu);
		a->state = CullAsync::QUIT;
		a->cv.notify_one();
	}
	if (a->worker.joinable()) a->worker.join();
	if (a->uploader) {
		a->uploader->destroy();
		delete a->uploader;
	}
	if (a->stream) (void)hipStreamDestroy(a->stream);
	if (a->swapped) (void)hipEventDestroy(a->swapped);
	cs.async = nullptr;
	delete a;
}

bool wants_compaction(const CullState& cs) {
	if (!layout_live(cs)) return true;
	if (!cs.auto_compaction) return false; // the host schedules lmx_cull_compact itself (loading screen, level streaming boundary)
	const size_t n_static = cs.recs.size();
	return cs.n_unbound > std::max<size_t>(cs.compaction_min, n_static / 8) || cs.n_tombstones > std::max<size_t>(cs.compaction_min, n_static / 4);
}

int flush_impl(LmxContext* ctx, bool force_compaction) {
	CullState& cs = ctx->cull;
	bool layout_changed = false;
	bool compact = wants_compaction(cs) || (force_compaction && (cs.n_unbound || cs.n_tombstones));
	if (cs.async && layout_live(cs) && !force_compaction) {
		// the re-sort belongs to the worker: hand it this flush's operations, adopt its result if one is ready, ask for a job when due
		bool swapped = false;
		if (int rc = async_poll(ctx, &swapped)) return rc;
		compact = false;
		(void)swapped; // (async_swap re-derived the output layout itself)
	}
	if (compact && cs.async) async_wait_idle(*cs.async); // a synchronous rebuild (first build, lmx_cull_compact): the shadow set is re-seeded below
	if (compact) {
		if (cs.built && cs.n_unbound) {
			cs.structure_dirty = true; // from here on the mirror ops below must not queue patches against the old layout
			fold_overflow(cs);
		}
		cs.structure_dirty = true;
		if (int rc = rebuild_static(ctx)) return rc;
		layout_changed = true;
		if (cs.async) { // the live set changed outside the operation log
			async_reseed(cs);
			std::lock_guard<std::mutex> g(cs.async->mu);
			if (cs.async->state != CullAsync::QUIT) cs.async->state = CullAsync::IDLE;
		}
	}
	if (cs.dyn_layout_dirty) {
		if (int rc = cull_dyn_sync_mirror(ctx)) return rc; // keep what the device refreshed before slots move
		if (int rc = rebuild_dynamic(ctx)) return rc;
		layout_changed = true;
	}
	if (layout_changed) {
		if (int rc = recompute_out_layout(ctx)) return rc;
	}
	return apply_patches(ctx);
}

CullDeviceView static_view(const CullSet& cs) {
	CullDeviceView v;
	v.spheres = cs.spheres.p;
	v.ids = cs.ids.p;
	v.hdr = cs.hdr.p;
	v.n_padded = cs.n_padded;
	v.keys_packed = cs.keys_packed;
	for (int k = 0; k < 3; ++k) {
		v.tile_cells[k] = cs.tile_cells[k].p;
		v.tile_tab[k] = cs.tile_tab[k].p;
		v.tile_box[k] = cs.tile_box[k].p;
		v.tile_cap[k] = cs.tile_cap[k];
		v.tile_out[k] = nullptr; // (belongs to the output layout: the caller fills it)
	}
	return v;
}

} // namespace

namespace lmx {

// dyn[] <- device for the entities lmx_world_propagate refreshes (the host is the only writer of everything else)
int cull_dyn_sync_mirror(LmxContext* ctx) {
	CullState& cs = ctx->cull;
	if (!cs.dyn_mirror_stale) return LMX_OK;
	cs.dyn_mirror_stale = false;
	if (!cs.dyn_padded) return LMX_OK;
	LMX_CHECK_CTX(ctx); // reached from host-only entry points too
	if (int rc = apply_patches(ctx)) return rc; // host-side sets queued since the refresh are newer than what the device holds
	const size_t padded = cs.dyn_padded;
	std::vector<double> px(padded), py(padded), pz(padded);
	std::vector<float> radius(padded);
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	LMX_HIP(ctx, hipMemcpy(px.data(), cs.dyn_px.p, padded * sizeof(double), hipMemcpyDeviceToHost));
	LMX_HIP(ctx, hipMemcpy(py.data(), cs.dyn_py.p, padded * sizeof(double), hipMemcpyDeviceToHost));
	LMX_HIP(ctx, hipMemcpy(pz.data(), cs.dyn_pz.p, padded * sizeof(double), hipMemcpyDeviceToHost));
	LMX_HIP(ctx, hipMemcpy(radius.data(), cs.dyn_radius.p, padded * sizeof(float), hipMemcpyDeviceToHost));
	for (DynRec& r : cs.dyn) {
		if (!r.bound || r.slot == DYN_NO_SLOT || r.slot >= padded) continue;
		r.pos[0] = px[r.slot];
		r.pos[1] = py[r.slot];
		r.pos[2] = pz[r.slot];
		r.radius = radius[r.slot];
	}
	return LMX_OK;
}

bool cull_make_dynamic(LmxContext* ctx, int32_t entity) {
	if (!make_dynamic_impl(ctx->cull, entity)) return false;
	async_log(ctx->cull, OP_BIND, entity, 0, nullptr, 0.f);
	return true;
}

void cull_unbind(LmxContext* ctx, int32_t entity) {
	unbind_impl(ctx->cull, entity);
	async_log(ctx->cull, OP_UNBIND, entity, 0, nullptr, 0.f);
}

int cull_flush(LmxContext* ctx) { return flush_impl(ctx, false); }

void cull_async_shutdown(LmxContext* ctx) { async_disable(ctx->cull); }

int cull_view_finalize(LmxContext* ctx, CullView& v) {
	CullState& cs = ctx->cull;
	if (v.finalized) return LMX_OK;
	LMX_HIP(ctx, v.totals.reserve(MAX_FRUSTA * MAX_TYPES));
	LMX_HIP(ctx, v.pref.reserve(std::max<size_t>((size_t)MAX_FRUSTA * cs.n_shards, 1)));
	uint32_t* totals = v.ext_counts ? v.ext_counts : v.totals.p;
	LMX_HIP(ctx, launch_cull_finalize(ctx->stream, v.counts_ptr(), cs.cnt_pad, cs.n_shards * cs.cnt_pad, cs.d_shard_type.p, cs.n_shards, v.n_frusta, totals, v.pref.p, nullptr));
	v.finalized = true;
	return LMX_OK;
}

int cull_view_consolidate(LmxContext* ctx, CullView& v) {
	CullState& cs = ctx->cull;
	if (v.consolidated) return LMX_OK;
	if (int rc = cull_view_finalize(ctx, v)) return rc;
	int32_t* dst = v.ext_out;
	if (!dst) {
		LMX_HIP(ctx, v.cons.reserve(std::max<size_t>((size_t)v.out_stride * v.n_frusta, 1)));
		dst = v.cons.p;
	}
	if (v.has_slots) LMX_HIP(ctx, v.cons_slots.reserve(std::max<size_t>((size_t)v.out_stride * v.n_frusta, 1))); // the same gather for the slots, in the same launch
	LMX_HIP(ctx, launch_cull_consolidate(ctx->stream, v.out.p, v.out_stride, cs.d_win_base.p, v.counts_ptr(), cs.cnt_pad, cs.n_shards * cs.cnt_pad, cs.d_shard_type.p,
		cs.d_type_start.p, 0, v.pref.p, cs.n_shards, v.n_frusta, cs.max_shard_cap, dst, v.out_stride, 0xffffffffu, v.has_slots ? v.out_slots.p : nullptr,
		v.has_slots ? v.cons_slots.p : nullptr));
	v.consolidated = true;
	return LMX_OK;
}

} // namespace lmx

extern "C" {

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
	cs.dyn.clear();
	cs.ent_to_dyn.clear();
	cs.n_unbound = 0;
	cs.dyn_layout_dirty = true;
	cs.dyn_mirror_stale = false;
	clear_static_queues(cs);
	cs.q_dyn.clear();
	cs.q_dyn_at.clear();
	cs.ent_to_rec.assign((size_t)max_entity + 1, -1);
	cs.structure_dirty = true;
	for (uint32_t i = 0; i < n; ++i) {
		if (cs.ent_to_rec[entity[i]] >= 0) {
			cs.recs.clear();
			cs.ent_to_rec.clear();
			return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d added twice", entity[i]);
		}
		cs.ent_to_rec[entity[i]] = (int32_t)i;
	}
	cs.recs.resize(n);
	parallel_ranges(n, [&](size_t b, size_t e) {
		for (size_t i = b; i < e; ++i) cs.recs[i] = make_cull_rec(entity[i], type[i], DV3{pos_xyz[3 * i], pos_xyz[3 * i + 1], pos_xyz[3 * i + 2]}, radius[i]);
	});
	return cull_flush(ctx);
}

// add / remove / set only touch the host mirror and the patch queues: no HIP call, no hipSetDevice per entity
int lmx_cull_add(LmxContext* ctx, int32_t entity, uint8_t type, const double pos[3], float radius) {
	if (!ctx) return LMX_ERR_INVALID_ARGUMENT;
	const int rc = cull_add_impl(ctx, ctx->cull, entity, type, pos, radius);
	if (rc == LMX_OK) async_log(ctx->cull, OP_ADD, entity, type, pos, radius);
	return rc;
}
int lmx_cull_remove(LmxContext* ctx, int32_t entity) {
	if (!ctx) return LMX_ERR_INVALID_ARGUMENT;
	bool effective;
	const int rc = cull_remove_impl(ctx->cull, entity, &effective);
	if (rc == LMX_OK && effective) async_log(ctx->cull, OP_REMOVE, entity, 0, nullptr, 0.f);
	return rc;
}
int lmx_cull_set(LmxContext* ctx, int32_t entity, const double pos[3], float radius) {
	if (!ctx) return LMX_ERR_INVALID_ARGUMENT;
	bool effective;
	const int rc = cull_set_impl(ctx, ctx->cull, ctx->cull.device_owns_bound, entity, pos, radius, &effective);
	if (rc == LMX_OK && effective) async_log(ctx->cull, OP_SET, entity, 0, pos, radius);
	return rc;
}

int lmx_cull_set_position(LmxContext* ctx, int32_t entity, const double pos[3]) { // culling_system.cpp:201-217
	LMX_CHECK_CTX(ctx);
	bool effective;
	const int rc = cull_set_position_impl(ctx, ctx->cull, ctx->cull.device_owns_bound, false, entity, pos, &effective);
	if (rc == LMX_OK && effective) async_log(ctx->cull, OP_SET_POS, entity, 0, pos, 0.f);
	return rc;
}

int lmx_cull_set_radius(LmxContext* ctx, int32_t entity, float radius) { // culling_system.cpp:244-260
	LMX_CHECK_CTX(ctx);
	bool effective;
	const int rc = cull_set_radius_impl(ctx, ctx->cull, ctx->cull.device_owns_bound, false, entity, radius, &effective);
	if (rc == LMX_OK && effective) async_log(ctx->cull, OP_SET_RADIUS, entity, 0, nullptr, radius);
	return rc;
}

int lmx_cull_get_radius(LmxContext* ctx, int32_t entity, float* out_radius) {
	LMX_CHECK_CTX(ctx);
	CullState& cs = ctx->cull;
	uint32_t idx;
	switch (locate(cs, entity, &idx)) {
		case Where::STATIC:
			if (out_radius) *out_radius = cs.recs[idx].radius;
			return LMX_OK;
		case Where::DYNAMIC:
			if (cs.dyn[idx].bound) {
				if (int rc = cull_dyn_sync_mirror(ctx)) return rc;
			}
			if (out_radius) *out_radius = cs.dyn[idx].radius;
			return LMX_OK;
		case Where::NONE: break;
	}
	return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d is not in the culling system", entity);
}

int lmx_cull_is_added(LmxContext* ctx, int32_t entity) {
	if (!ctx) return 0;
	uint32_t idx;
	return locate(ctx->cull, entity, &idx) != Where::NONE ? 1 : 0;
}

// Batched forms of add / remove / set for hosts that pay per call (ctypes, scripting): same semantics, one ABI crossing. An update
// touches 3-5 random entries of tables that hold one element per entity (entity -> record, record, record -> device slot): with 10 M
// entities every one of them is a DRAM miss, and the misses of ONE update depend on each other. The batch forms run a two-stage
// software prefetch ahead of the update loop (entity -> record index PF_FAR updates ahead, the record and its slot PF_NEAR ahead), so
// the misses of neighbouring updates overlap.
constexpr uint32_t PF_FAR = 24, PF_NEAR = 12;
static inline void prefetch_update(const CullSet& cs, const int32_t* entity, uint32_t n, uint32_t i) {
	if (i + PF_FAR < n) {
		const int32_t e = entity[i + PF_FAR];
		if (e >= 0) {
			if ((size_t)e < cs.ent_to_rec.size()) __builtin_prefetch(&cs.ent_to_rec[e]);
			if ((size_t)e < cs.ent_to_dyn.size()) __builtin_prefetch(&cs.ent_to_dyn[e]);
		}
	}
	if (i + PF_NEAR < n) {
		const int32_t e = entity[i + PF_NEAR];
		if (e >= 0 && (size_t)e < cs.ent_to_rec.size()) {
			const int32_t r = cs.ent_to_rec[e]; // prefetched PF_FAR - PF_NEAR updates ago; may be stale by the time it is used: a hint only
			if (r >= 0 && (size_t)r < cs.recs.size()) {
				__builtin_prefetch(&cs.recs[r]);
				if ((size_t)r < cs.rec_slot.size()) __builtin_prefetch(&cs.rec_slot[r]);
			}
		}
		if (e >= 0 && (size_t)e < cs.ent_to_dyn.size()) {
			const int32_t d = cs.ent_to_dyn[e];
			if (d >= 0 && (size_t)d < cs.dyn.size()) __builtin_prefetch(&cs.dyn[d]);
		}
	}
}

int lmx_cull_add_many(LmxContext* ctx, uint32_t n, const int32_t* entity, const uint8_t* type, const double* pos_xyz, const float* radius) {
	if (!ctx) return LMX_ERR_INVALID_ARGUMENT;
	if (n && (!entity || !type || !pos_xyz || !radius)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	for (uint32_t i = 0; i < n; ++i) {
		prefetch_update(ctx->cull, entity, n, i);
		if (int rc = cull_add_impl(ctx, ctx->cull, entity[i], type[i], pos_xyz + 3 * (size_t)i, radius[i])) return rc;
		async_log(ctx->cull, OP_ADD, entity[i], type[i], pos_xyz + 3 * (size_t)i, radius[i]);
	}
	return LMX_OK;
}

int lmx_cull_set_many(LmxContext* ctx, uint32_t n, const int32_t* entity, const double* pos_xyz, const float* radius) {
	if (!ctx) return LMX_ERR_INVALID_ARGUMENT;
	if (n && (!entity || !pos_xyz || !radius)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	for (uint32_t i = 0; i < n; ++i) {
		prefetch_update(ctx->cull, entity, n, i);
		bool effective;
		if (int rc = cull_set_impl(ctx, ctx->cull, ctx->cull.device_owns_bound, entity[i], pos_xyz + 3 * (size_t)i, radius[i], &effective)) return rc;
		if (effective) async_log(ctx->cull, OP_SET, entity[i], 0, pos_xyz + 3 * (size_t)i, radius[i]);
	}
	return LMX_OK;
}

int lmx_cull_remove_many(LmxContext* ctx, uint32_t n, const int32_t* entity) {
	if (!ctx) return LMX_ERR_INVALID_ARGUMENT;
	if (n && !entity) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	for (uint32_t i = 0; i < n; ++i) {
		prefetch_update(ctx->cull, entity, n, i);
		bool effective;
		if (int rc = cull_remove_impl(ctx->cull, entity[i], &effective)) return rc;
		if (effective) async_log(ctx->cull, OP_REMOVE, entity[i], 0, nullptr, 0.f);
	}
	return LMX_OK;
}

int lmx_cull_flush(LmxContext* ctx) {
	LMX_CHECK_CTX(ctx);
	return cull_flush(ctx);
}

int lmx_cull_compact(LmxContext* ctx) {
	LMX_CHECK_CTX(ctx);
	return flush_impl(ctx, true);
}

int lmx_cull_stats(LmxContext* ctx, uint32_t* n_entities, uint32_t* n_cells, uint32_t* n_chunks) {
	LMX_CHECK_CTX(ctx);
	if (int rc = flush_impl(ctx, true)) return rc; // the cell count below is that of the sorted layout: fold the overflow in first
	const CullState& cs = ctx->cull;
	if (n_entities) *n_entities = (uint32_t)(cs.recs.size() + cs.dyn.size());
	if (n_cells) *n_cells = cs.n_cells - cs.n_dead_cells;
	if (n_chunks) *n_chunks = (cs.out_total + CHUNK - 1) / CHUNK;
	return LMX_OK;
}

// What the device layout's per-tile tables look like: *cell_key_bytes = 8 (keys relative to the tile's box: every tile spans <= 65535 cell indices per
// axis) or 16; *table_bytes = the cell keys + tile tables + chunk headers a cull of the whole static set reads besides spheres and ids.
int lmx_cull_layout_info(LmxContext* ctx, uint32_t* cell_key_bytes, uint64_t* table_bytes) {
	LMX_CHECK_CTX(ctx);
	if (int rc = flush_impl(ctx, true)) return rc;
	const CullState& cs = ctx->cull;
	const uint32_t kb = cs.keys_packed ? (uint32_t)sizeof(PackedCellKey) : (uint32_t)sizeof(CellKey);
	if (cell_key_bytes) *cell_key_bytes = kb;
	if (table_bytes) { // (the 2048-sphere tiles' tables: what the 1-frustum kernels walk)
		const uint64_t tiles = cs.n_padded / 2048u;
		*table_bytes = tiles * ((uint64_t)cs.tile_cap[1] * kb + 2 * sizeof(uint32_t) + sizeof(TileBox) + sizeof(uint2)) + (uint64_t)(cs.n_padded / CHUNK) * sizeof(ChunkHdr);
	}
	return LMX_OK;
}

int lmx_cull_update_stats(LmxContext* ctx, uint32_t* n_static, uint32_t* n_dynamic_bound, uint32_t* n_overflow, uint32_t* n_tombstones) {
	LMX_CHECK_CTX(ctx);
	const CullState& cs = ctx->cull;
	if (n_static) *n_static = (uint32_t)cs.recs.size();
	if (n_dynamic_bound) *n_dynamic_bound = (uint32_t)(cs.dyn.size() - cs.n_unbound);
	if (n_overflow) *n_overflow = cs.n_unbound;
	if (n_tombstones) *n_tombstones = cs.n_tombstones;
	return LMX_OK;
}

int lmx_cull_async_stats(LmxContext* ctx, int* state, uint64_t* jobs, uint64_t* swaps, uint64_t* ops_replayed_at_swaps, uint64_t* log_drains) {
	if (!ctx) return LMX_ERR_INVALID_ARGUMENT;
	CullAsync* a = ctx->cull.async;
	if (state) *state = a ? (int)async_state(*a) : -1;
	if (a) {
		std::lock_guard<std::mutex> g(a->mu);
		if (jobs) *jobs = a->jobs_done;
		if (log_drains) *log_drains = a->drains;
	} else {
		if (jobs) *jobs = 0;
		if (log_drains) *log_drains = 0;
	}
	if (swaps) *swaps = a ? a->swaps : 0;
	if (ops_replayed_at_swaps) *ops_replayed_at_swaps = a ? a->ops_replayed_at_swap : 0;
	return LMX_OK;
}

// Fraction of the static set's bounding box that the frustum's own bounding box (its 8 corner points) overlaps: a cheap, stateless
// predictor of how many tiles survive the tile-level test. Only used to pick between kernel variants that return identical results.
static double frustum_box_overlap(const CullState& cs, const LmxShiftedFrustum& f) {
	if (cs.big_tile_fraction > 0.5) return 1.0; // tiles that hold big spheres are never rejected as a whole
	double vol_scene = 1, vol_overlap = 1;
	for (int a = 0; a < 3; ++a) {
		double lo = INFINITY, hi = -INFINITY;
		for (int k = 0; k < 8; ++k) {
			const double p = f.origin[a] + (double)f.points[k][a];
			lo = std::min(lo, p);
			hi = std::max(hi, p);
		}
		const double extent = cs.scene_hi[a] - cs.scene_lo[a];
		if (!(extent > 0) || !(hi >= lo)) return 1.0; // empty set / non-finite corners: no prediction
		vol_scene *= extent;
		vol_overlap *= std::max(0.0, std::min(hi, cs.scene_hi[a]) - std::max(lo, cs.scene_lo[a]));
	}
	return vol_overlap / vol_scene;
}

int lmx_cull(LmxContext* ctx, uint32_t view, const LmxShiftedFrustum* frusta, uint32_t n_frusta, uint8_t type) {
	LMX_CHECK_CTX(ctx);
	if (view >= LMX_MAX_VIEWS) return fail(ctx, LMX_ERR_CAPACITY, "view %u >= LMX_MAX_VIEWS", view);
	if (!frusta || n_frusta == 0 || n_frusta > LMX_MAX_FRUSTA) return fail(ctx, LMX_ERR_CAPACITY, "n_frusta %u not in [1,%d]", n_frusta, LMX_MAX_FRUSTA);
	if (type != LMX_TYPE_ALL && type >= MAX_TYPES) return fail(ctx, LMX_ERR_CAPACITY, "type %u >= LMX_MAX_TYPES", type);
	if (int rc = cull_flush(ctx)) return rc;
	CullState& cs = ctx->cull;
	CullView& v = cs.views[view];
	if (v.ext_out && v.ext_out_cap < (size_t)cs.out_total * n_frusta)
		return fail(ctx, LMX_ERR_CAPACITY, "bound output holds %zu ids, need %zu", v.ext_out_cap, (size_t)cs.out_total * n_frusta);
	const uint32_t cnt_frustum_stride = cs.n_shards * cs.cnt_pad;
	const uint32_t cnt_words = std::max(1u, MAX_FRUSTA * cnt_frustum_stride);
	if (v.cnt_words != cnt_words) { // first cull on this view / the shard layout changed: both halves start from zero
		LMX_HIP(ctx, v.counts.reserve(2 * (size_t)cnt_words));
		v.cnt_words = cnt_words;
		v.flip = 0;
		LMX_HIP(ctx, hipMemsetAsync(v.counts.p, 0, 2 * (size_t)cnt_words * sizeof(uint32_t), ctx->stream));
		v.next_half_is_zero = true;
	}
	LMX_HIP(ctx, v.out.reserve(std::max<size_t>((size_t)cs.out_total * n_frusta, 1)));
	if (cs.emit_slots) LMX_HIP(ctx, v.out_slots.reserve(std::max<size_t>((size_t)cs.out_total * n_frusta, 1)));
	v.has_slots = cs.emit_slots;
	v.n_frusta = n_frusta;
	v.out_stride = cs.out_total;
	v.valid = v.finalized = v.consolidated = false;
	for (int t = 0; t < MAX_TYPES; ++t) {
		v.out_start[t] = cs.type_start[t];
		v.out_cap[t] = cs.type_cap[t];
	}
	FrustaArg fr;
	memset(&fr, 0, sizeof(fr));
	for (uint32_t f = 0; f < n_frusta; ++f) fr.f[f] = to_dev_frustum(frusta[f]);

	uint32_t ent_begin = 0, ent_end = cs.n_padded, dyn_begin = 0, dyn_end = cs.dyn_padded;
	if (type != LMX_TYPE_ALL) {
		ent_begin = cs.tt.ent_start[type];
		ent_end = cs.tt.ent_end[type];
		dyn_begin = cs.dyn_tt.ent_start[type];
		dyn_end = cs.dyn_tt.ent_end[type];
	}
	// counters: this cull uses the half the previous one cleared; its first static launch clears the other half
	v.flip ^= 1u;
	if (!v.next_half_is_zero) LMX_HIP(ctx, hipMemsetAsync(v.counts_ptr(), 0, (size_t)cnt_words * sizeof(uint32_t), ctx->stream));
	v.next_half_is_zero = ent_end > ent_begin;
	CullOut out;
	out.ids = v.out.p;
	out.stride = v.out_stride;
	out.win_base = cs.d_win_base.p;
	out.counts = v.counts_ptr();
	out.cnt_pad = cs.cnt_pad;
	out.cnt_frustum_stride = cnt_frustum_stride;
	out.counts_next = v.counts_other();
	out.n_zero = cnt_words;
	out.slots = cs.emit_slots ? v.out_slots.p : nullptr;
	CullDeviceView dv = static_view(cs);
	for (int k = 0; k < 3; ++k) dv.tile_out[k] = cs.d_tile_out[k].p;
	// The kernel is latency-bound for small frusta: wide variants (many frusta per pass) hold more state per wave and run at lower
	// occupancy, so a batch is split into passes of at most `pass_width` frusta.
	// pass_width 0 = automatic: ONE launch for all frusta of the call up to 32 M spheres, frustum by frustum above. Every launch of this
	// latency-bound kernel costs its floor (launch + one round of block start-ups), and since round 4's rework of the several-frusta kernel
	// one pass over the spheres beats eight at 10 M in every regime measured (profiles/r04/width_rule_call39.txt: a frame's 6 views
	// 66 -> 39 us, 8 small cascades 82 -> 49; every sphere tested against 8 frusta 298 -> 141, cull8_pass_widths_call23_*.txt). At 100 M
	// it is the other way round (6 views 185 vs 207 us, config 5's cascades 281 vs 373): the several-frusta kernel walks 1024-sphere tiles
	// with 28 KiB of LDS each - 98 k blocks, five resident per CU - and the blocks that only reject their tile are what the launch is
	// made of there; the 1-frustum kernels walk 2048- / 4096-sphere tiles, eight blocks per CU.
	const uint32_t pass_width = cs.pass_width ? cs.pass_width : (ent_end - ent_begin <= (1u << 25) ? n_frusta : 1u);
	for (uint32_t f0 = 0; f0 < n_frusta; f0 += pass_width) {
		const uint32_t fw = std::min(pass_width, n_frusta - f0);
		FrustaArg sub;
		memset(&sub, 0, sizeof(sub));
		for (uint32_t k = 0; k < fw; ++k) sub.f[k] = fr.f[f0 + k];
		CullOut po = out;
		po.ids = out.ids + (size_t)f0 * out.stride;
		if (out.slots) po.slots = out.slots + (size_t)f0 * out.stride;
		po.counts = out.counts + (size_t)f0 * cnt_frustum_stride;
		if (f0 != 0) { // the first pass of the cull has cleared the next cull's counters already
			po.counts_next = nullptr;
			po.n_zero = 0;
		}
		// 2048-sphere tiles of 4 waves x 8 chunks measured best in every regime (default camera, all-accept, all-test; 10 M and 100 M).
		// With all 8 chunks' loads in flight (variant 4, 66 VGPRs) a launch in which few tiles survive the tile-level test is 7 % shorter
		// (its duration is the latency of the surviving tiles), a launch that streams the whole set 2 % longer: picked by how much of the
		// set's bounding box the frustum's bounding box overlaps.
		const int variant = cs.tile_variant >= 0 ? cs.tile_variant : (fw == 1 && frustum_box_overlap(cs, frusta[f0]) < 0.25 ? 4 : 1);
		ProfScope ps(ctx, LMX_K_CULL_SPHERES, true);
		po.ev_start = ps.slot.a;
		po.ev_stop = ps.slot.b;
		const hipError_t launched = launch_cull_tile(ctx->stream, dv, ent_begin, ent_end, cs.tt, sub, (int)fw, po, variant);
		if (launched != hipSuccess) ps.cancel(); // (the launch fills the scope's events itself: none were recorded)
		LMX_HIP(ctx, launched);
	}
	// dynamic set: its own shards of the same rows / counters
	if (dyn_end > dyn_begin) {
		CullOut po = out;
		po.counts_next = nullptr;
		po.n_zero = 0;
		ProfScope ps(ctx, LMX_K_CULL_DYNAMIC);
		LMX_HIP(ctx, launch_cull_dynamic(ctx->stream, dyn_view(cs), dyn_begin, dyn_end, cs.dyn_tt, fr, (int)n_frusta, po));
	}
	v.valid = true;
	if (v.ext_out) return cull_view_consolidate(ctx, v); // a bound output receives the contiguous form right away
	return LMX_OK;
}

int lmx_cull_set_pass_width(LmxContext* ctx, uint32_t frusta_per_pass) {
	LMX_CHECK_CTX(ctx);
	if (frusta_per_pass > LMX_MAX_FRUSTA) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "pass width %u not in [0,%d]", frusta_per_pass, LMX_MAX_FRUSTA);
	ctx->cull.pass_width = frusta_per_pass;
	return LMX_OK;
}

int lmx_cull_set_option(LmxContext* ctx, int option, int value) {
	LMX_CHECK_CTX(ctx);
	CullState& cs = ctx->cull;
	switch (option) {
		case LMX_CULL_OPT_TILE_VARIANT:
			if (value != -1 && value != 1 && value != 4) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "tile variant %d: -1 (auto), 1 (streaming) or 4 (all loads in flight)", value);
			cs.tile_variant = value;
			return LMX_OK;
		case LMX_CULL_OPT_AUTO_COMPACTION: cs.auto_compaction = value != 0; return LMX_OK;
		case LMX_CULL_OPT_COMPACTION_MIN:
			if (value < 1) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "compaction minimum %d < 1", value);
			cs.compaction_min = (uint32_t)value;
			return LMX_OK;
		case LMX_CULL_OPT_DEVICE_OWNS_BOUND: cs.device_owns_bound = value != 0; return LMX_OK;
		case LMX_CULL_OPT_MAP_ZERO_COPY: cs.map_zero_copy = value != 0; cs.map_zero_copy_max = value > 1 ? (uint32_t)value : (1u << 20); return LMX_OK; // (value > 1: the threshold in ids)
		case LMX_CULL_OPT_ASYNC_COMPACTION:
			if (value) return async_enable(ctx);
			async_disable(cs);
			return LMX_OK;
		case LMX_CULL_OPT_OVERFLOW_RESERVE:
			if (value < 0) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "overflow reserve %d < 0", value);
			cs.overflow_reserve = (uint32_t)value;
			if (value) {
				cs.dyn_layout_dirty = true; // the next flush lays the dynamic set out with the reserve
				// the host mirror of the overflow gets its room now as well: a std::vector that doubles under an add copies tens of MB
				// at 10 M entities (measured: one 8.6 ms frame in a stream of 2 M adds, the entity -> overflow table crossing 10 M ids)
				const size_t ids = std::max(cs.ent_to_dyn.size(), cs.ent_to_rec.size());
				cs.ent_to_dyn.reserve(ids + (size_t)value);
				if (cs.ent_to_dyn.size() < ids) cs.ent_to_dyn.resize(ids, -1);
				cs.dyn.reserve(cs.dyn.size() + (size_t)value);
			}
			return LMX_OK;
		case LMX_CULL_OPT_MAX_SHARDS:
			if (value < 1 || value > (int)LAYOUT_MAX_SHARDS) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "max shards %d not in [1,%u]", value, LAYOUT_MAX_SHARDS);
			cs.max_shards = (uint32_t)value;
			if (cs.built) return recompute_out_layout(ctx);
			return LMX_OK;
		case LMX_CULL_OPT_COUNTER_PAD:
			if (value < 1 || value > 64) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "counter pad %d not in [1,64]", value);
			cs.cnt_pad = (uint32_t)value;
			for (CullView& v : cs.views) {
				v.valid = v.finalized = v.consolidated = false;
				v.cnt_words = 0;
			}
			return LMX_OK;
		default: return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "unknown cull option %d", option);
	}
}

int lmx_cull_counts(LmxContext* ctx, uint32_t view, uint32_t* counts) {
	LMX_CHECK_CTX(ctx);
	if (view >= LMX_MAX_VIEWS || !counts) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad view/counts");
	CullView& v = ctx->cull.views[view];
	if (!v.valid) return fail(ctx, LMX_ERR_NOT_BUILT, "view %u holds no cull result", view);
	if (int rc = cull_view_finalize(ctx, v)) return rc;
	uint32_t all[MAX_FRUSTA * MAX_TYPES];
	LMX_HIP(ctx, hipMemcpyAsync(all, v.totals_ptr(), sizeof(uint32_t) * v.n_frusta * MAX_TYPES, hipMemcpyDeviceToHost, ctx->stream));
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
	if (int rc = cull_view_consolidate(ctx, v)) return rc;
	uint32_t c = 0;
	LMX_HIP(ctx, hipMemcpyAsync(&c, v.totals_ptr() + frustum * MAX_TYPES + type, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (out_count) *out_count = c;
	if (c > v.out_cap[type]) return fail(ctx, LMX_ERR_HIP, "corrupt count %u > %u", c, v.out_cap[type]);
	if (!out_ids || c == 0) return LMX_OK;
	if (c > cap) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u ids, got %u", c, cap);
	LMX_HIP(ctx, hipMemcpyAsync(out_ids, v.cons_ptr() + (size_t)frustum * v.out_stride + v.out_start[type], (size_t)c * sizeof(int32_t),
		hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

// All types of one frustum with two host waits (totals, then every non-empty type's ids): what the CullResult adapter needs.
int lmx_cull_read_all(LmxContext* ctx, uint32_t view, uint32_t frustum, int32_t* out_ids, uint32_t cap, uint32_t* out_counts) {
	LMX_CHECK_CTX(ctx);
	if (view >= LMX_MAX_VIEWS || !out_counts) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad view / counts");
	CullView& v = ctx->cull.views[view];
	if (!v.valid) return fail(ctx, LMX_ERR_NOT_BUILT, "view %u holds no cull result", view);
	if (frustum >= v.n_frusta) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "frustum %u out of range", frustum);
	if (int rc = cull_view_consolidate(ctx, v)) return rc;
	LMX_HIP(ctx, hipMemcpyAsync(out_counts, v.totals_ptr() + frustum * MAX_TYPES, sizeof(uint32_t) * MAX_TYPES, hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	size_t total = 0;
	for (int t = 0; t < MAX_TYPES; ++t) {
		if (out_counts[t] > v.out_cap[t]) return fail(ctx, LMX_ERR_HIP, "corrupt count %u > %u", out_counts[t], v.out_cap[t]);
		total += out_counts[t];
	}
	if (!total) return LMX_OK;
	if (!out_ids || total > cap) return fail(ctx, LMX_ERR_CAPACITY, "need room for %zu ids, got %u", total, cap);
	size_t at = 0;
	for (int t = 0; t < MAX_TYPES; ++t) {
		if (!out_counts[t]) continue;
		LMX_HIP(ctx, hipMemcpyAsync(out_ids + at, v.cons_ptr() + (size_t)frustum * v.out_stride + v.out_start[t], (size_t)out_counts[t] * sizeof(int32_t),
			hipMemcpyDeviceToHost, ctx->stream));
		at += out_counts[t];
	}
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

// All types of one frustum with (normally) ONE host wait: the gather kernels pack [counts | ids, type 0 first] into one device
// record, and the copy into pinned host memory is enqueued right behind them for as many ids as the previous call on this view
// returned (+25 %): frames are coherent, so the guess almost always covers the list; when it does not, the rest follows with a second
// wait. The caller reads the ids in place: *out_ids stays valid until the next lmx_cull_map_all on this view.
// (Letting the gather kernel store straight into mapped host memory was measured too: 4-byte stores over PCIe, 0.49 ms for 334 k ids.)
// Records [MAX_TYPES counts | ids, types back to back] of frusta [first, first + n) of a view, each packed by one k_cull_pack launch into
// its own area of map_rec and copied into pinned host memory - counts + the first map_guess ids before the count is known - with ONE
// host wait for all of them (a second one only for a frustum whose list outgrew its guess, this frame only).
// The host read of a view's result in two halves, so that render jobs culling different views only serialise on the ENQUEUE:
//   cull_map_begin  (context lock held) packs the shard windows into one record per frustum, enqueues its copy into the view's pinned
//                   buffer - as many ids as the last frame on that view needed + 25 % - and records the view's event behind it;
//   cull_map_end    (no lock needed: touches this view's buffers only) waits for THAT event, reads the counts, and - only if the list
//                   outgrew the guess - takes the lock for a second copy.
static int cull_map_begin(LmxContext* ctx, CullView& v, uint32_t first, uint32_t n) {
	CullState& cs = ctx->cull;
	const size_t need = (size_t)MAX_TYPES + v.out_stride; // words per record area
	if (v.map_words < need || v.map_frusta < v.n_frusta) {
		LMX_HIP(ctx, hipStreamSynchronize(ctx->stream)); // nothing may still write the old buffer
		if (v.map_host) LMX_HIP(ctx, hipHostFree(v.map_host));
		v.map_host = nullptr;
		v.map_words = 0;
		const size_t want = need + need / 4 + 1024;
		const size_t areas = std::max<size_t>(v.n_frusta, v.map_frusta);
		LMX_HIP(ctx, hipHostMalloc(&v.map_host, want * areas * sizeof(int32_t), hipHostMallocDefault));
		LMX_HIP(ctx, v.map_rec.reserve(want * areas));
		v.map_words = want;
		v.map_frusta = areas;
	}
	if (!v.map_event) LMX_HIP(ctx, hipEventCreateWithFlags(&v.map_event, hipEventDisableTiming));
	const uint32_t cnt_frustum_stride = cs.n_shards * cs.cnt_pad;
	// Lists of up to 1 M ids last frame: k_cull_pack writes the record STRAIGHT into the pinned host buffer (the buffer's device mapping:
	// posted writes over PCIe) - no copy command behind the kernel, whose fixed cost (~10 us of a ~45 us cull of the harness's 40 k-entity
	// scene) is what a host read of a small list consists of; at the headline camera's 334 k ids (1.3 MB) the host read is still 20 us
	// shorter this way (104 against 124 us per cull + read through the Python wrapper, profiles/r04/readback_zero_copy_call43.txt; round 4's
	// first cut stopped at 64 k ids). Larger lists keep the device record + one DMA copy of the ids the last frame needed: a kernel that
	// streams many megabytes over PCIe holds its CUs for the duration.
	int32_t* host_dev = nullptr;
	// (only once a count has been read back on this view: the initial guess says nothing about the list, and a zero-copy record streams
	// ALL its ids over PCIe with the CUs held - a first map of a 10 M-id list would be tens of megabytes of posted writes)
	bool zero_copy = cs.map_zero_copy && v.map_seen;
	for (uint32_t k = 0; k < n && zero_copy; ++k) zero_copy = v.map_guess[first + k].load(std::memory_order_relaxed) <= cs.map_zero_copy_max;
	if (zero_copy && hipHostGetDevicePointer(reinterpret_cast<void**>(&host_dev), v.map_host, 0) != hipSuccess) zero_copy = false;
	CullView::MapTicket& tk = v.ticket;
	tk.n = 0;
	tk.zero_copy = zero_copy;
	tk.host = reinterpret_cast<int32_t*>(v.map_host);
	tk.rec = v.map_rec.p;
	tk.words = v.map_words;
	memcpy(tk.out_cap, v.out_cap, sizeof(tk.out_cap));
	{ // the records of all n frusta: ONE launch (a frame's six views cost six launch gaps otherwise)
		int32_t* rec = (zero_copy ? host_dev : v.map_rec.p) + (size_t)first * v.map_words;
		if (v.map_words > 0xffffffffull) return fail(ctx, LMX_ERR_CAPACITY, "record stride exceeds 32 bits");
		LMX_HIP(ctx, launch_cull_pack(ctx->stream, v.out.p + (size_t)first * v.out_stride, cs.d_win_base.p, v.counts_ptr() + (size_t)first * cnt_frustum_stride, cs.cnt_pad,
			cs.d_shard_type.p, cs.n_shards, cs.max_shard_cap, reinterpret_cast<uint32_t*>(rec), rec + MAX_TYPES, v.out_stride, n, (uint32_t)v.out_stride, cnt_frustum_stride,
			(uint32_t)v.map_words));
	}
	for (uint32_t k = 0; k < n; ++k) {
		const uint32_t f = first + k;
		const int32_t* rec = v.map_rec.p + (size_t)f * v.map_words;
		tk.guess[k] = zero_copy ? (size_t)v.out_stride : std::min<size_t>(v.out_stride, v.map_guess[f].load(std::memory_order_relaxed));
		if (!zero_copy)
			LMX_HIP(ctx, hipMemcpyAsync(tk.host + (size_t)f * v.map_words, rec, (MAX_TYPES + tk.guess[k]) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
	}
	LMX_HIP(ctx, hipEventRecord(v.map_event, ctx->stream));
	tk.first = first;
	tk.n = n;
	v.map_seen = true; // (the matching map_end reads the counts before the next map_begin on this view can run)
	return LMX_OK;
}

static int cull_map_end(LmxContext* ctx, CullView& v, uint32_t first, uint32_t n, const int32_t** out_ids, uint32_t* out_counts) {
	struct Locked { // (error strings and stream operations belong to the context: taken only on the rare paths)
		LmxContext* c;
		explicit Locked(LmxContext* c_) : c(c_) { c->lock.lock(); }
		~Locked() { c->lock.unlock(); }
	};
	// everything read here is the view's ticket (filled by map_begin under the lock, untouched until the next map_begin on this view), its
	// event and the pinned buffer the ticket names
	const CullView::MapTicket& tk = v.ticket;
	if (!v.map_event || tk.n != n || tk.first != first) {
		Locked l(ctx);
		return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_cull_map_end without a matching lmx_cull_map_begin on this view");
	}
	if (hipEventSynchronize(v.map_event) != hipSuccess) {
		Locked l(ctx);
		return fail(ctx, LMX_ERR_HIP, "waiting for the view's record failed");
	}
	bool more = false;
	for (uint32_t k = 0; k < n; ++k) {
		const uint32_t f = first + k;
		int32_t* host = tk.host + (size_t)f * tk.words;
		const uint32_t* h = reinterpret_cast<const uint32_t*>(host);
		size_t total = 0;
		for (int t = 0; t < MAX_TYPES; ++t) {
			if (h[t] > tk.out_cap[t]) {
				Locked l(ctx);
				return fail(ctx, LMX_ERR_HIP, "corrupt count %u > %u", h[t], tk.out_cap[t]);
			}
			out_counts[k * MAX_TYPES + t] = h[t];
			total += h[t];
		}
		if (total > tk.guess[k] && !tk.zero_copy) { // the list outgrew the guess: fetch the rest (second wait, this frame only)
			Locked l(ctx);
			LMX_HIP(ctx, hipSetDevice(ctx->device)); // (this thread may never have selected the context's device)
			LMX_HIP(ctx, hipMemcpyAsync(host + MAX_TYPES + tk.guess[k], tk.rec + (size_t)f * tk.words + MAX_TYPES + tk.guess[k],
				(total - tk.guess[k]) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
			more = true;
		}
		v.map_guess[f].store((uint32_t)std::min<size_t>(total + total / 4 + 1024, 0xffffffffu), std::memory_order_relaxed);
		out_ids[k] = host + MAX_TYPES;
	}
	if (more) {
		Locked l(ctx);
		LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	}
	v.ticket.n = 0;
	return LMX_OK;
}

static int cull_map_range(LmxContext* ctx, CullView& v, uint32_t first, uint32_t n, const int32_t** out_ids, uint32_t* out_counts) {
	if (int rc = cull_map_begin(ctx, v, first, n)) return rc;
	return cull_map_end(ctx, v, first, n, out_ids, out_counts);
}

// The packed record of one frustum left in HBM, no host wait: what a device-side consumer of "one cull incl. compaction" reads
// (bench.py's timed step; the exchange packs into its own send buffer the same way).
int lmx_cull_pack_device(LmxContext* ctx, uint32_t view, uint32_t frustum, const int32_t** d_record, uint32_t* record_words) {
	LMX_CHECK_CTX(ctx);
	if (view >= LMX_MAX_VIEWS || !d_record) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad view / null output");
	CullState& cs = ctx->cull;
	CullView& v = cs.views[view];
	if (!v.valid) return fail(ctx, LMX_ERR_NOT_BUILT, "view %u holds no cull result", view);
	if (frustum >= v.n_frusta) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "frustum %u out of range", frustum);
	const size_t need = (size_t)MAX_TYPES + v.out_stride;
	if (v.pack_words < need) {
		LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		LMX_HIP(ctx, v.pack_rec.reserve(need + need / 4 + 1024));
		v.pack_words = need + need / 4 + 1024;
	}
	const uint32_t* counts = v.counts_ptr() + (size_t)frustum * cs.n_shards * cs.cnt_pad;
	LMX_HIP(ctx, launch_cull_pack(ctx->stream, v.out.p + (size_t)frustum * v.out_stride, cs.d_win_base.p, counts, cs.cnt_pad, cs.d_shard_type.p, cs.n_shards, cs.max_shard_cap,
		reinterpret_cast<uint32_t*>(v.pack_rec.p), v.pack_rec.p + MAX_TYPES, v.out_stride));
	*d_record = v.pack_rec.p;
	if (record_words) *record_words = (uint32_t)need;
	return LMX_OK;
}

int lmx_cull_map_all(LmxContext* ctx, uint32_t view, uint32_t frustum, const int32_t** out_ids, uint32_t* out_counts) {
	LMX_CHECK_CTX(ctx);
	if (view >= LMX_MAX_VIEWS || !out_counts || !out_ids) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad view / null output");
	CullView& v = ctx->cull.views[view];
	if (!v.valid) return fail(ctx, LMX_ERR_NOT_BUILT, "view %u holds no cull result", view);
	if (frustum >= v.n_frusta) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "frustum %u out of range", frustum);
	return cull_map_range(ctx, v, frustum, 1, out_ids, out_counts);
}

int lmx_cull_map_many(LmxContext* ctx, uint32_t view, uint32_t n_frusta, const int32_t** out_ids, uint32_t* out_counts) {
	LMX_CHECK_CTX(ctx);
	if (view >= LMX_MAX_VIEWS || !out_counts || !out_ids) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad view / null output");
	CullView& v = ctx->cull.views[view];
	if (!v.valid) return fail(ctx, LMX_ERR_NOT_BUILT, "view %u holds no cull result", view);
	if (n_frusta != v.n_frusta) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "the view holds %u frusta, not %u", v.n_frusta, n_frusta);
	return cull_map_range(ctx, v, 0, n_frusta, out_ids, out_counts);
}

int lmx_cull_map_begin(LmxContext* ctx, uint32_t view, uint32_t n_frusta) {
	LMX_CHECK_CTX(ctx);
	if (view >= LMX_MAX_VIEWS) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad view");
	CullView& v = ctx->cull.views[view];
	if (!v.valid) return fail(ctx, LMX_ERR_NOT_BUILT, "view %u holds no cull result", view);
	if (n_frusta != v.n_frusta) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "the view holds %u frusta, not %u", v.n_frusta, n_frusta);
	return cull_map_begin(ctx, v, 0, n_frusta);
}

int lmx_cull_map_end(LmxContext* ctx, uint32_t view, uint32_t n_frusta, const int32_t** out_ids, uint32_t* out_counts) {
	if (!ctx) return LMX_ERR_INVALID_ARGUMENT; // (LMX_CHECK_CTX selects the context's device: not needed to wait for an event and read host memory)
	if (view >= LMX_MAX_VIEWS || !out_counts || !out_ids) return LMX_ERR_INVALID_ARGUMENT;
	return cull_map_end(ctx, ctx->cull.views[view], 0, n_frusta, out_ids, out_counts);
}

// Result slots that cannot alias (CullingSystem::cull returns an independent list per call, culling_system.cpp:321-369; callers
// pipeline.cpp:1036-1045, :3380, editor/scene_view.cpp:144): a slot handed out here is not handed out again before its holder has
// released it, i.e. before it has copied the ids out of the slot's pinned record. With every slot taken the caller waits for the
// next release - bounded: a holder that never releases turns into LMX_ERR_BUSY, not into a hang.
int lmx_cull_view_acquire(LmxContext* ctx, uint32_t* view, uint32_t timeout_ms) {
	if (!ctx || !view) return LMX_ERR_INVALID_ARGUMENT;
	CullState& cs = ctx->cull;
	std::unique_lock<std::mutex> l(cs.views_mutex);
	constexpr uint32_t ALL = (1u << LMX_MAX_VIEWS) - 1u;
	if ((cs.views_busy & ALL) == ALL) {
		const bool got = cs.views_cv.wait_for(l, std::chrono::milliseconds(timeout_ms), [&] { return (cs.views_busy & ALL) != ALL; });
		if (!got) return LMX_ERR_BUSY; // (no fail(): the error string belongs to the context's lock, which this path never takes)
	}
	for (uint32_t k = 0; k < (uint32_t)LMX_MAX_VIEWS; ++k) { // round robin: consecutive culls of a frame land on different slots (their buffers stay sized for their view)
		const uint32_t s = (cs.views_next + k) % (uint32_t)LMX_MAX_VIEWS;
		if (!((cs.views_busy >> s) & 1u)) {
			cs.views_busy |= 1u << s;
			cs.views_next = (s + 1u) % (uint32_t)LMX_MAX_VIEWS;
			*view = s;
			return LMX_OK;
		}
	}
	return LMX_ERR_BUSY; // (unreachable)
}

int lmx_cull_view_release(LmxContext* ctx, uint32_t view) {
	if (!ctx || view >= (uint32_t)LMX_MAX_VIEWS) return LMX_ERR_INVALID_ARGUMENT;
	CullState& cs = ctx->cull;
	{
		std::lock_guard<std::mutex> l(cs.views_mutex);
		if (!((cs.views_busy >> view) & 1u)) return LMX_ERR_INVALID_ARGUMENT; // released twice / never acquired
		cs.views_busy &= ~(1u << view);
	}
	cs.views_cv.notify_one();
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
	v.valid = v.finalized = v.consolidated = false;
	return LMX_OK;
}

int lmx_cull_device_result(LmxContext* ctx, uint32_t view, uint32_t frustum, const int32_t** d_ids, const uint32_t** d_counts,
	uint32_t* type_offsets, uint32_t* capacity) {
	LMX_CHECK_CTX(ctx);
	if (view >= LMX_MAX_VIEWS) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad view");
	CullView& v = ctx->cull.views[view];
	if (!v.valid) return fail(ctx, LMX_ERR_NOT_BUILT, "view %u holds no cull result", view);
	if (frustum >= v.n_frusta) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "frustum %u out of range", frustum);
	if (int rc = cull_view_consolidate(ctx, v)) return rc;
	if (d_ids) *d_ids = v.cons_ptr() + (size_t)frustum * v.out_stride;
	if (d_counts) *d_counts = v.totals_ptr();
	if (type_offsets) memcpy(type_offsets, v.out_start, sizeof(v.out_start));
	if (capacity) *capacity = v.out_stride;
	return LMX_OK;
}

int lmx_cull_device_shards(LmxContext* ctx, uint32_t view, uint32_t frustum, LmxCullShards* out) {
	LMX_CHECK_CTX(ctx);
	if (view >= LMX_MAX_VIEWS || !out) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad view / out");
	CullState& cs = ctx->cull;
	CullView& v = cs.views[view];
	if (!v.valid) return fail(ctx, LMX_ERR_NOT_BUILT, "view %u holds no cull result", view);
	if (frustum >= v.n_frusta) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "frustum %u out of range", frustum);
	out->d_ids = v.out.p + (size_t)frustum * v.out_stride;
	out->d_counts = v.counts_ptr() + (size_t)frustum * cs.n_shards * cs.cnt_pad;
	out->count_stride = cs.cnt_pad;
	out->d_window_start = cs.d_win_base.p;
	out->d_shard_type = cs.d_shard_type.p;
	out->n_shards = cs.n_shards;
	return LMX_OK;
}

} // extern "C"
