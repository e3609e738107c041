// lmx_context.h — internal state behind the opaque LmxContext of include/lumix_mi355.h, shared by the lmx_capi_*.hip
// translation units (context / culling / world transforms / skinning). Nothing here crosses the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "lumix_mi355.h"
#include "lmx_kernels.h"
#include "lmx_cull_layout.h"

namespace lmx {

template <typename T> struct DevBuf {
	T* p = nullptr;
	size_t cap = 0; // elements
	~DevBuf() { release(); }
	void release() {
		if (p) (void)hipFree(p);
		p = nullptr;
		cap = 0;
	}
	DevBuf() = default;
	DevBuf(const DevBuf&) = delete;
	DevBuf& operator=(const DevBuf&) = delete;
	void swap(DevBuf& o) {
		std::swap(p, o.p);
		std::swap(cap, o.cap);
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
	DevBuf<int32_t> out;      // raw result: [n_frusta][out_total], one window per output shard
	DevBuf<uint32_t> counts;  // [2][cnt_words] shard counters, double-buffered: the cull kernel clears the half the NEXT cull uses
	DevBuf<uint32_t> totals;  // [MAX_FRUSTA][MAX_TYPES] visible ids per (frustum, type)            (k_cull_finalize)
	DevBuf<uint32_t> pref;    // [MAX_FRUSTA][n_shards] offset of a shard inside its type's contiguous list (k_cull_finalize)
	DevBuf<int32_t> cons;     // [n_frusta][out_total] one contiguous list per (frustum, type)       (k_cull_consolidate)
	DevBuf<int32_t> out_slots, cons_slots; // CullState::emit_slots: the static-set slot of every id of `out` / `cons` (-1: dynamic set)
	bool has_slots = false;   // this view's result was culled with emit_slots on
	uint32_t n_frusta = 0;
	uint32_t out_stride = 0;
	uint32_t cnt_words = 0;   // words per half of `counts`
	uint32_t out_start[MAX_TYPES] = {};
	uint32_t out_cap[MAX_TYPES] = {};
	bool valid = false;       // holds a cull result
	bool finalized = false;   // totals / pref are current for that result
	bool consolidated = false;
	// caller-owned buffers for the CONSOLIDATED result (lmx_cull_bind_output)
	int32_t* ext_out = nullptr;
	size_t ext_out_cap = 0;
	uint32_t* ext_counts = nullptr;
	uint32_t flip = 0;
	bool next_half_is_zero = false;
	// lmx_cull_map_all: [MAX_TYPES counts | ids, type 0 first] gathered into map_rec, copied into pinned host memory
	void* map_host = nullptr;
	size_t map_words = 0;
	// ids (per frustum) the next call copies before it knows the count. Written by lmx_cull_map_end WITHOUT the context lock (a hint; relaxed atomics)
	std::atomic<uint32_t> map_guess[LMX_MAX_FRUSTA] = {{4096}, {4096}, {4096}, {4096}, {4096}, {4096}, {4096}, {4096}};
	bool map_seen = false;   // a count has been read back on this view: until then nothing is known about the list's size (no zero-copy record)
	size_t map_frusta = 0;   // record areas the buffers were sized for
	DevBuf<int32_t> map_rec;
	hipEvent_t map_event = nullptr; // recorded behind the record's copies (lmx_cull_map_begin): lmx_cull_map_end waits for THIS view only
	// What lmx_cull_map_begin enqueued, as lmx_cull_map_end needs it: map_end runs WITHOUT the context lock and reads nothing else of the
	// view (out_cap, map_words, map_host are touched by locked code paths of other threads: recompute_out_layout, the counter padding)
	struct MapTicket {
		uint32_t first = 0, n = 0;      // n == 0: no map_begin outstanding
		bool zero_copy = false;         // the pack kernel wrote the whole record into the pinned buffer itself
		int32_t* host = nullptr;        // the pinned buffer
		const int32_t* rec = nullptr;   // the device records
		size_t words = 0;               // words per record area
		size_t guess[LMX_MAX_FRUSTA] = {}; // ids the copies were asked for
		uint32_t out_cap[MAX_TYPES] = {};
	} ticket;
	DevBuf<uint32_t> map_pref, map_start;
	DevBuf<int32_t> pack_rec; // lmx_cull_pack_device: the packed record of one frustum, left on the device
	size_t pack_words = 0;
	~CullView() {
		if (map_host) (void)hipHostFree(map_host);
		if (map_event) (void)hipEventDestroy(map_event);
	}
	CullView() = default;
	CullView(const CullView&) = delete;
	CullView& operator=(const CullView&) = delete;
	uint32_t* counts_ptr() const { return counts.p + flip * cnt_words; }
	uint32_t* counts_other() const { return counts.p + (flip ^ 1u) * cnt_words; }
	const int32_t* cons_ptr() const { return ext_out ? ext_out : cons.p; }
	const uint32_t* totals_ptr() const { return ext_counts ? ext_counts : totals.p; }
};

// One entity of the dynamic set (see DynDeviceView): what CullingSystem::set(entity, pos, radius) was last called with.
struct DynRec {
	double pos[3];
	float radius;
	int32_t entity;
	uint32_t slot;  // device slot, DYN_NO_SLOT until the next rebuild_dynamic when the type's region was full
	uint8_t type;
	bool bound;     // refreshed on the device by lmx_world_propagate (never folded back into the static set)
};
constexpr uint32_t DYN_NO_SLOT = 0xffffffffu;

// Pinned host staging buffer for the patch records of one flush, double-buffered (the host refills one half while the copy of
// the other may still be in flight).
struct PatchStaging {
	void* host[2] = {nullptr, nullptr}; // pinned, mapped
	void* dev[2] = {nullptr, nullptr};  // the same memory as the device addresses it
	size_t cap[2] = {0, 0};
	hipEvent_t done[2] = {nullptr, nullptr};
	uint32_t next = 0;
	PatchStaging() = default;
	PatchStaging(const PatchStaging&) = delete;
	PatchStaging& operator=(const PatchStaging&) = delete;
	void swap(PatchStaging& o) {
		for (int i = 0; i < 2; ++i) {
			std::swap(host[i], o.host[i]);
			std::swap(dev[i], o.dev[i]);
			std::swap(cap[i], o.cap[i]);
			std::swap(done[i], o.done[i]);
		}
		std::swap(next, o.next);
	}
	~PatchStaging() {
		for (int i = 0; i < 2; ++i) {
			if (host[i]) (void)hipHostFree(host[i]);
			if (done[i]) (void)hipEventDestroy(done[i]);
		}
	}
};

// One complete copy of the culling sets: host mirror + device layout of the static set, the dynamic set, the pending patch queues.
// A context has ONE live set (CullState derives from it); with LMX_CULL_OPT_ASYNC_COMPACTION a second, shadow set exists that a worker
// thread re-sorts in the background (lmx_capi_cull.hip, "asynchronous compaction") and that trades places with the live one in O(1).
struct CullSet {
	// ---- static set: host mirror (one CullRec per entity) + sorted device layout -------------------------------
	std::vector<CullRec> recs;
	std::vector<int32_t> ent_to_rec; // entity -> index into recs, or -1
	std::vector<uint32_t> rec_slot;  // rec -> device sphere slot, valid while !structure_dirty
	bool structure_dirty = false;    // the device layout must be rebuilt from recs (first build / compaction)
	bool built = false;
	uint32_t n_tombstones = 0;       // slots of the device layout whose entity was removed or moved to the dynamic set
	DevBuf<float4> spheres;
	DevBuf<int32_t> ids;
	DevBuf<ChunkHdr> hdr;
	DevBuf<CellKey> tile_cells[3];
	DevBuf<uint32_t> tile_tab[3];
	DevBuf<TileBox> tile_box[3];
	uint32_t tile_cap[3] = {16, 16, 16};
	bool keys_packed = false; // tile_cells hold PackedCellKey (8 bytes) instead of CellKey
	uint32_t n_padded = 0, n_cells = 0, n_dead_cells = 0;
	uint32_t max_tile_cells[3] = {0, 0, 0};
	std::vector<uint32_t> block_live; // live ids per TILE_ALIGN-slot block at build time (capacity of the output shards)
	double scene_lo[3] = {0, 0, 0}, scene_hi[3] = {0, 0, 0}; // world-space box of the static set's occupied cells
	double big_tile_fraction = 0;                             // share of the non-empty tiles that hold a big sphere (never rejected as a whole)
	TypeTable tt = {};
	// ---- dynamic set: entities bound to the world hierarchy + entities added / re-celled since the last compaction ----
	std::vector<DynRec> dyn;
	std::vector<int32_t> ent_to_dyn;  // entity -> index into dyn, or -1
	uint32_t n_unbound = 0;           // dyn records that are not bound to the hierarchy (folded into the static set by a compaction)
	bool dyn_layout_dirty = false;    // a type's region is full (or the set was reset): regions / slots must be reassigned
	bool dyn_mirror_stale = false;    // the device refreshed pos / radius of bound entities (lmx_world_propagate): dyn[] is older
	DevBuf<double> dyn_px, dyn_py, dyn_pz;
	DevBuf<float> dyn_radius;
	DevBuf<int32_t> dyn_ids;
	uint32_t dyn_padded = 0;
	TypeTable dyn_tt = {};
	uint32_t dyn_next[MAX_TYPES] = {};           // first never-used slot of the type's region
	std::vector<uint32_t> dyn_free[MAX_TYPES];   // slots freed by removals
	uint64_t dyn_generation = 0; // bumped whenever slots are reassigned (world binding tables depend on it)
	// ---- O(1) updates between culls ---------------------------------------------------------------------------
	// at most ONE pending record per slot (the patch kernel applies a batch in parallel): a second change of a slot
	// overwrites its pending record
	std::vector<PatchSphere> q_sphere;
	std::vector<PatchId> q_id;
	std::vector<PatchDyn> q_dyn;
	std::vector<uint32_t> q_sphere_at; // static slot -> index into q_sphere, ~0u = no pending record (a hash map here cost 0.7 us per set)
	std::vector<uint32_t> q_dyn_at;                     // dynamic slot -> index into q_dyn, or ~0u
	PatchStaging staging;
	// everything above trades places with `o` (O(1): vectors and device buffers swap their storage)
	void swap_with(CullSet& o) {
		recs.swap(o.recs);
		ent_to_rec.swap(o.ent_to_rec);
		rec_slot.swap(o.rec_slot);
		std::swap(structure_dirty, o.structure_dirty);
		std::swap(built, o.built);
		std::swap(n_tombstones, o.n_tombstones);
		spheres.swap(o.spheres);
		ids.swap(o.ids);
		hdr.swap(o.hdr);
		for (int k = 0; k < 3; ++k) {
			tile_cells[k].swap(o.tile_cells[k]);
			tile_tab[k].swap(o.tile_tab[k]);
			tile_box[k].swap(o.tile_box[k]);
			std::swap(tile_cap[k], o.tile_cap[k]);
			std::swap(max_tile_cells[k], o.max_tile_cells[k]);
			std::swap(scene_lo[k], o.scene_lo[k]);
			std::swap(scene_hi[k], o.scene_hi[k]);
		}
		std::swap(keys_packed, o.keys_packed);
		std::swap(n_padded, o.n_padded);
		std::swap(n_cells, o.n_cells);
		std::swap(n_dead_cells, o.n_dead_cells);
		block_live.swap(o.block_live);
		std::swap(big_tile_fraction, o.big_tile_fraction);
		std::swap(tt, o.tt);
		dyn.swap(o.dyn);
		ent_to_dyn.swap(o.ent_to_dyn);
		std::swap(n_unbound, o.n_unbound);
		std::swap(dyn_layout_dirty, o.dyn_layout_dirty);
		std::swap(dyn_mirror_stale, o.dyn_mirror_stale);
		dyn_px.swap(o.dyn_px);
		dyn_py.swap(o.dyn_py);
		dyn_pz.swap(o.dyn_pz);
		dyn_radius.swap(o.dyn_radius);
		dyn_ids.swap(o.dyn_ids);
		std::swap(dyn_padded, o.dyn_padded);
		std::swap(dyn_tt, o.dyn_tt);
		for (int t = 0; t < MAX_TYPES; ++t) {
			std::swap(dyn_next[t], o.dyn_next[t]);
			dyn_free[t].swap(o.dyn_free[t]);
		}
		std::swap(dyn_generation, o.dyn_generation);
		q_sphere.swap(o.q_sphere);
		q_id.swap(o.q_id);
		q_dyn.swap(o.q_dyn);
		q_sphere_at.swap(o.q_sphere_at);
		q_dyn_at.swap(o.q_dyn_at);
		staging.swap(o.staging);
	}
};

struct CullAsync; // lmx_capi_cull.hip

struct CullState : CullSet {
	// ---- output shards ----------------------------------------------------------------------------------------
	uint32_t n_shards = 0, max_shard_cap = 0;
	std::vector<uint8_t> shard_type;
	std::vector<uint32_t> win_base;
	DevBuf<uint32_t> d_win_base, d_type_start;
	DevBuf<uint8_t> d_shard_type;
	uint32_t type_start[MAX_TYPES] = {}, type_cap[MAX_TYPES] = {};
	// ---- tuning (lmx_cull_set_option) -------------------------------------------------------------------------
	uint32_t pass_width = 0;   // frusta tested per pass over the static set; 0 = automatic (all of them for a set of <= 32 M spheres, else 1: lmx_cull)
	int tile_variant = -1;     // -1: chosen per cull from the frustum's coverage of the scene
	uint32_t overflow_reserve = 0; // LMX_CULL_OPT_OVERFLOW_RESERVE: slots kept free in the dynamic set for entities added / re-celled between compactions
	bool device_owns_bound = false; // LMX_CULL_OPT_DEVICE_OWNS_BOUND: set* calls on hierarchy-bound entities are dropped
	bool auto_compaction = true; // false: overflow / tombstones accumulate until the host calls lmx_cull_compact
	uint32_t compaction_min = 1u << 16; // overflow entities / tombstones tolerated before a compaction is considered at all (LMX_CULL_OPT_COMPACTION_MIN)
	bool map_zero_copy = true;   // LMX_CULL_OPT_MAP_ZERO_COPY: small host records are written by the pack kernel straight into pinned host memory
	uint32_t map_zero_copy_max = 1u << 20; // ... for views whose lists held at most this many ids last frame
	CullAsync* async = nullptr;  // LMX_CULL_OPT_ASYNC_COMPACTION: shadow set + worker thread (owned; lmx_capi_cull.hip)
	bool emit_slots = false;     // culls also write the static-set slot of every visible id (switched on by the sort-key tables' slot-ordered mirror)
	uint64_t layout_generation = 0; // a process-wide unique number per build of the static layout (consumers that mirror it by slot compare)
	uint32_t max_shards = LAYOUT_MAX_SHARDS; // output shards per type of the static set
	uint32_t cnt_pad = 32;     // words between shard counters (32 = one 128-byte line each)
	uint32_t out_total = 0;    // ids per frustum row = sum of the shard capacities
	DevBuf<uint2> d_tile_out[3]; // CullDeviceView::tile_out of the LIVE set under the current output layout (recompute_out_layout)
	CullView views[LMX_MAX_VIEWS];
	// lmx_cull_view_acquire / _release: result slots whose record a caller is still reading are not handed out again. Own mutex (not the
	// context's recursive lock: a waiter must be able to sleep while other threads enqueue and release)
	std::mutex views_mutex;
	std::condition_variable views_cv;
	uint32_t views_busy = 0, views_next = 0;
};

struct WorldState {
	uint32_t n = 0;
	bool built = false;
	std::vector<int32_t> parent; // by entity, -1 for roots
	std::vector<int32_t> slot_of_entity, entity_of_slot, parent_slot;
	std::vector<uint32_t> level_start; // size levels + 1
	DevBuf<double> pos[6];             // lpx lpy lpz wpx wpy wpz
	DevBuf<float4> rot[2];             // lrot wrot
	DevBuf<float> scl[6];              // lsx lsy lsz wsx wsy wsz
	DevBuf<int32_t> d_parent_slot, d_slot_of_entity, d_entity_of_slot;
	DevBuf<uint8_t> d_dirty;
	DevBuf<uint32_t> d_sub_table; // k_xform_subtree: (n_sub_runs + 1) x n_levels first slots (runs of consecutive roots)
	uint32_t n_sub_runs = 0;      // 0: the hierarchy does not fit the table (too deep, or one root too heavy): per-level launches
	bool fused_levels = true;     // lmx_world_set_option(LMX_WORLD_OPT_FUSED_LEVELS): 1 = one launch per propagation where the table fits; 0 = one launch per level
	DevBuf<uint32_t> d_bound_dyn_of_slot; // per slot: the bound entity's index in the culling system's dynamic set, or 0xffffffff
	DevBuf<float> d_bound_radius_of_slot;
	DevBuf<int32_t> d_stage_entity;
	DevBuf<LmxTransform> d_stage_tr;
	DevBuf<LmxTransform> d_export;
	std::vector<uint32_t> stage_mark; // per entity: the staging call that last wrote it (duplicate detection, last write wins)
	uint32_t stage_stamp = 0;
	// the moved list of the last propagation(s) (lmx_world_track_moved / lmx_world_read_moved)
	bool track_moved = false;
	DevBuf<int32_t> d_moved_entity;
	DevBuf<LmxTransform> d_moved_tr;
	DevBuf<uint32_t> d_moved_count;
	uint32_t moved_guess = 1024; // records lmx_world_read_moved copies before it knows the count (last frame's count x 1.25)
	// culling binding (RenderModuleImpl::onModelInstanceMoved)
	std::vector<int32_t> bound_entity;
	std::vector<float> bound_radius;
	DevBuf<uint32_t> d_bound_slot, d_bound_dyn;
	DevBuf<float> d_bound_radius;
	uint64_t bound_generation = ~0ull; // CullState::dyn_generation the device binding tables were built for
	DevBuf<BoneAttachDevice> d_attach;  // RenderModuleImpl::m_bone_attachments
	uint32_t n_attach = 0;
	bool attach_invalidated = false;    // a hierarchy rebuild renumbered the slots the attachment table refers to
	size_t attach_skin_instances = 0;
	WorldDevice dev() {
		WorldDevice w;
		w.lpx = pos[0].p; w.lpy = pos[1].p; w.lpz = pos[2].p; w.lrot = rot[0].p; w.lsx = scl[0].p; w.lsy = scl[1].p; w.lsz = scl[2].p;
		w.wpx = pos[3].p; w.wpy = pos[4].p; w.wpz = pos[5].p; w.wrot = rot[1].p; w.wsx = scl[3].p; w.wsy = scl[4].p; w.wsz = scl[5].p;
		w.parent_slot = d_parent_slot.p;
		w.dirty = d_dirty.p;
		return w;
	}
};

struct SkinModel { uint32_t bone_offset, n_bones, max_depth; int32_t first_nonroot; uint32_t lv_items_offset, lv_off_offset; };
// max_bone: the largest bone index its vertices reference. k_skin_shared's view of the mesh: tiles of tile_verts vertices
// (tiles_at .. + n_tiles in SkinState::tiles), each with the list of bones it references and records that index into that list.
struct SkinMesh { uint32_t vert_offset, n_verts, max_bone, tile_verts, n_tiles, tiles_at; };
struct SkinTile { uint32_t bones_at, n_bones; }; // into SkinState::tile_bones

struct SkinState {
	std::vector<SkinModel> models;
	std::vector<SkinMesh> meshes;
	// concatenated host copies (re-uploaded when models/meshes are added)
	std::vector<int16_t> parents;
	std::vector<uint8_t> depth;
	std::vector<uint32_t> level_items; // per model: bone | parent << 16 of the bones >= first_nonroot, sorted by depth
	std::vector<uint16_t> level_off;   // per model: max_depth + 1 offsets into its level_items (k_pose_palette)
	DevBuf<uint32_t> d_level_items;
	DevBuf<uint16_t> d_level_off;
	std::vector<PoseGroup> groups;     // sorted by capacity class (4, 2, 1 instances per group)
	uint32_t n_groups[3] = {0, 0, 0};
	std::vector<SkinChunk> chunks;     // k_skin_shared work items (runs of instances sharing a mesh)
	struct Run { uint32_t first, count, mesh; };
	std::vector<Run> runs;             // the runs themselves (lmx_skin_set_instances): k_skin_multi's work items are cut from them at run time
	std::vector<SkinMultiChunk> multi_chunks;
	DevBuf<SkinMultiChunk> d_multi_chunks;
	uint32_t multi = 2;                // LMX_SKIN_OPT_INSTANCES_PER_BLOCK (0: k_skin_shared). 2: a store instruction writes two runs of 32 vertices = 384 bytes = three whole 128-byte lines each; 4 (runs of 192 bytes) measured 3.3-3.5 against 2.4-2.7 ms per 1e9 vertices, 8 / 16 7 / 13 ms (profiles/r04/skin_ab_*.txt)
	uint32_t multi_built = 0;          // the value multi_chunks were cut for (0: stale)
	uint32_t multi_max_stage = 0;      // the largest staging of a chunk in float4 slots (skin_multi_lds_slots): sizes the launch's LDS
	std::vector<uint32_t> solo;        // instances skinned by k_skin_vertices (empty + no chunks = all of them)
	DevBuf<SkinChunk> d_chunks;
	DevBuf<uint32_t> d_solo;
	uint32_t solo_max_verts = 0;
	DevBuf<PoseGroup> d_groups;
	std::vector<float> inv_pos;
	std::vector<float4> inv_rot;
	std::vector<float4> mesh_local;    // the same records with TILE-LOCAL bone indices (k_skin_shared), same vertex offsets as `mesh`
	std::vector<SkinTile> tiles;
	std::vector<uint8_t> tile_bones;   // per tile: the model bones it references, in local-index order
	DevBuf<float4> d_mesh_local;
	DevBuf<uint8_t> d_tile_bones;
	std::vector<float4> mesh; // 2 per vertex: {w0, w1, w2, w3}, {x, y, z, bone indices as 4 x u8} (skin_kernels.hip: RawVertex)
	bool models_dirty = false, meshes_dirty = false;
	DevBuf<int16_t> d_parents;
	DevBuf<float> d_inv_pos;
	DevBuf<float4> d_inv_rot;
	DevBuf<float4> d_mesh;
	std::vector<SkinInstance> inst;
	DevBuf<SkinInstance> d_inst;
	DevBuf<float> d_pose_pos;
	DevBuf<float4> d_pose_rot;
	DevBuf<float4> d_palette;
	DevBuf<float> d_out;
	size_t bones_total = 0, verts_total = 0;
	uint32_t max_verts = 0;
	bool poses_uploaded = false;
	int mode = 0; // LMX_SKIN_FUSED / LMX_SKIN_EXACT / LMX_SKIN_DQS
	bool want_dual_quats = false;
	bool pose_writeback = true;    // store the absolute pose (Pose::is_absolute) next to the palette
	bool pose_is_absolute = false; // d_pose_* hold the absolute pose of the last run
	DevBuf<float4> d_palette_expanded;
	DevBuf<float> d_blend_pos;     // staging of lmx_skin_blend_poses (host variant)
	DevBuf<float4> d_blend_rot;
	const float* borrowed_pos = nullptr;  // lmx_skin_set_pose_source_device: relative poses read in place from caller memory
	const float4* borrowed_rot = nullptr;
	DevBuf<float4> d_dual_quats;
};


// createSortKeys (lmx_capi_keys.hip): entity-indexed model-instance tables, decal material tables, outputs of the last run
struct KeysState {
	std::vector<LmxKeysModel> models;
	std::vector<uint8_t> mesh_types;
	uint32_t n_meshes = 0, max_lod_span = 1;
	uint32_t n_entities = 0, n_positions = 0, max_sort_key = 0;
	size_t offsets_at = 0; // where the CSR offsets of the last run start inside d_groups
	size_t counters_at = 0; // ... and the list counters of the last run (KEYS_COUNTERS words)
	// the instancer's counter tables and the list counters take turns from run to run (lmx_keys_run): which of the two this run uses, and
	// the layout the pair was last zeroed for
	uint32_t run_parity = 0, table_parity = 0;
	bool groups_clean = false;
	const uint32_t* groups_at = nullptr;
	uint32_t groups_copies = 0;
	size_t groups_keys = 0;
	bool walk_shards = true; // lmx_keys_set_option(LMX_KEYS_OPT_WALK_SHARDS)
	bool block_ranks = true; // lmx_keys_set_option(LMX_KEYS_OPT_BLOCK_RANKS)
	DevBuf<uint32_t> d_block_rows, d_rec_rank, d_total_pad;
	size_t pad_keys = 0;
	uint32_t pad_parity = 0;
	bool have_instances = false, have_decals = false, have_curves = false, use_world = false, ran = false, sorted = false;
	DevBuf<LmxKeysModel> d_models;
	DevBuf<uint8_t> d_mesh_types;
	std::vector<KeysInstance> inst; // host mirror of the per-entity records (tables + positions), uploaded when it changes
	bool inst_dirty = false;
	size_t inst_uploaded = 0; // records on the device
	DevBuf<KeysInstance> d_inst;
	DevBuf<LmxMeshMaterial> d_mesh_materials;
	DevBuf<uint32_t> d_decal_key, d_curve_key;
	DevBuf<uint8_t> d_decal_layer, d_curve_layer;
	DevBuf<uint64_t> d_keys, d_values, d_keys_alt, d_values_alt, d_rec_value, d_group_values;
	DevBuf<uint32_t> d_rec_key, d_groups;
	DevBuf<int32_t> d_poses, d_dirty_list;
	DevBuf<char> d_sort_temp;
	// slot-ordered mirror of d_inst / d_mesh_materials for the entities of the culling system's sorted set (keys_kernels.hip)
	bool slot_order = true;          // lmx_keys_set_option(LMX_KEYS_OPT_SLOT_ORDER)
	bool mirror_valid = false;
	uint64_t mirror_generation = 0;  // CullState::layout_generation the mirror was built for
	uint32_t mirror_slots = 0;       // n_padded of that layout
	size_t n_mesh_materials = 0;
	DevBuf<KeysInstance> d_inst_s;
	DevBuf<LmxMeshMaterial> d_mm_s;
	DevBuf<KeysSlotState> d_state_s; // LMX_KEYS_OPT_SPLIT_STATE: lod / Pose::frame of the sorted set's entities, 8 bytes per slot
	DevBuf<double> d_soa_pos;        // LMX_KEYS_OPT_SPLIT_STATE = 2: px | py | pz, n_slots each
	DevBuf<int32_t> d_soa_model;
	DevBuf<uint32_t> d_soa_mat;
	DevBuf<uint16_t> d_soa_flags;
	int split_state = LMX_KEYS_SPLIT_STATE_DEFAULT; // 0: AoS mirror, 1: + lod / Pose::frame in d_state_s, 2: structure-of-arrays mirror
	int mirror_split = 0;            // the form the current mirror was built in
	KeysSoA soa() const { // the current mirror's arrays (all null unless it was built as a structure of arrays)
		KeysSoA a{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
		if (mirror_split == 2) a = KeysSoA{d_soa_pos.p, d_soa_pos.p + mirror_slots, d_soa_pos.p + 2 * (size_t)mirror_slots, d_soa_model.p, d_soa_mat.p, d_soa_flags.p};
		return a;
	}
	DevBuf<uint32_t> d_mm_count, d_mm_off;
	DevBuf<char> d_scan_temp;
};

// animation sampling (lmx_capi_anim.hip): Animation resources flattened into concatenated tables, one Animable per skin instance
struct AnimState {
	std::vector<AnimDevice> anims;
	std::vector<int32_t> src;
	std::vector<LmxAnimConstTranslation> ct;
	std::vector<LmxAnimTranslationTrack> tt;
	std::vector<LmxAnimConstRotation> cr;
	std::vector<LmxAnimRotationTrack> rt;
	std::vector<uint8_t> tstream, rstream;
	std::vector<float> root_t;
	std::vector<float4> root_r;
	std::vector<float> rel_pos;    // Model::Bone::relative_transform of every skin model, by model bone offset
	std::vector<float4> rel_rot;
	bool tables_dirty = false;
	uint32_t n_animables = 0;
	float weight = 1.f;
	DevBuf<AnimDevice> d_anims;
	DevBuf<int32_t> d_src;
	DevBuf<LmxAnimConstTranslation> d_ct;
	DevBuf<LmxAnimTranslationTrack> d_tt;
	DevBuf<LmxAnimConstRotation> d_cr;
	DevBuf<LmxAnimRotationTrack> d_rt;
	DevBuf<uint8_t> d_tstream, d_rstream;
	DevBuf<float> d_root_t, d_rel_pos;
	DevBuf<float4> d_root_r, d_rel_rot;
	DevBuf<uint32_t> d_anim_of, d_time_of;
	DevBuf<LmxBlendSample> d_samples; // lmx_anim_eval_blend_stacks: the frame's SAMPLE instructions and their per-instance ranges
	DevBuf<uint32_t> d_first_sample;
};

struct ProfSlot { hipEvent_t a, b; int kernel; };


} // namespace lmx

struct LmxContext {
	std::recursive_mutex lock; // lmx_ctx_lock / lmx_ctx_unlock: adapters sharing the context serialise here
	int device = 0;
	hipStream_t own_stream = nullptr;
	hipStream_t stream = nullptr;
	std::string error;
	bool profiling = false;
	std::vector<lmx::ProfSlot> prof_pending;
	std::vector<hipEvent_t> event_pool;
	double prof_ms[LMX_K_COUNT] = {};
	uint64_t prof_launches[LMX_K_COUNT] = {};
	lmx::CullState cull;
	lmx::WorldState world;
	lmx::SkinState skin;
	lmx::KeysState keys;
	lmx::AnimState anim;
};

namespace lmx {

int fail(LmxContext* ctx, int code, const char* fmt, ...); // records the message, returns `code`
extern thread_local std::string* t_fail_sink; // non-null on a library-owned thread: fail() writes there instead of LmxContext::error
void cull_async_shutdown(LmxContext* ctx);   // lmx_capi_cull.hip: stop the asynchronous compaction's worker (context teardown)
int keys_before_layout_change(LmxContext* ctx); // lmx_capi_keys.hip: the slot-ordered mirror of the sort-key tables hands its state back (the static layout is about to change)
int keys_before_tombstones(LmxContext* ctx, const PatchId* d_patches, uint32_t n); // ... for the slots of these id patches (device-visible records; enqueued BEFORE the patch kernel)
void prof_drain(LmxContext* ctx);
int cull_flush(LmxContext* ctx);          // lmx_capi_cull.hip: make the device copy of the culling sets current
int cull_dyn_sync_mirror(LmxContext* ctx); // dyn[] <- device when lmx_world_propagate refreshed it
bool cull_make_dynamic(LmxContext* ctx, int32_t entity); // move an entity to the dynamic set and mark it as bound to the hierarchy
void cull_unbind(LmxContext* ctx, int32_t entity);       // the entity is no longer refreshed by lmx_world_propagate
int cull_view_finalize(LmxContext* ctx, CullView& v);    // per-type totals of the view's result
int cull_view_consolidate(LmxContext* ctx, CullView& v); // + one contiguous id list per (frustum, type)

#define LMX_HIP(ctx, expr)                                                                                             \
	do {                                                                                                               \
		hipError_t e_ = (expr);                                                                                        \
		if (e_ != hipSuccess)                                                                                          \
			return lmx::fail(ctx, e_ == hipErrorOutOfMemory ? LMX_ERR_OUT_OF_MEMORY : LMX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
				hipGetErrorString(e_), __FILE__, __LINE__);                                                           \
	} while (0)

#define LMX_CHECK_CTX(ctx)                                                                                             \
	do {                                                                                                               \
		if (!(ctx)) return LMX_ERR_INVALID_ARGUMENT;                                                                   \
		hipError_t e_ = hipSetDevice((ctx)->device);                                                                   \
		if (e_ != hipSuccess) return lmx::fail(ctx, LMX_ERR_NO_DEVICE, "hipSetDevice(%d): %s", (ctx)->device, hipGetErrorString(e_)); \
	} while (0)

struct ProfScope { // records HIP events around one launch on the launch stream when profiling is enabled
	LmxContext* ctx;
	ProfSlot slot;
	bool on;
	bool ext; // the launch itself fills the events (hipExtLaunchKernelGGL: the dispatch's own timestamps)
	ProfScope(LmxContext* c, int kernel, bool launch_records = false) : ctx(c), on(c->profiling), ext(launch_records) {
		slot.a = slot.b = nullptr;
		if (!on) return;
		slot.kernel = kernel;
		slot.a = take();
		slot.b = take();
		if (!ext) (void)hipEventRecord(slot.a, ctx->stream);
	}
	~ProfScope() {
		if (!on) return;
		if (!ext) (void)hipEventRecord(slot.b, ctx->stream);
		ctx->prof_pending.push_back(slot);
	}
	// the launch that should have filled the events failed before it was issued: the pair goes back to the pool unrecorded (queued, the
	// drain would call hipEventElapsedTime on events nobody recorded and hide the launch's own error behind its own)
	void cancel() {
		if (!on) return;
		ctx->event_pool.push_back(slot.a);
		ctx->event_pool.push_back(slot.b);
		on = false;
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

} // namespace lmx
