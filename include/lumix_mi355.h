/* lumix_mi355.h — C ABI of liblumix_mi355.so: the MI355X (gfx950) implementation of LumixEngine's per-frame
 * cull / transform / skin hot path.
 *
 * The reference exposes exactly one C symbol (`createPlugin`, src/engine/plugin.h:92-96); everything behind it is
 * C++ vtables. This header is the seam a LumixEngine-side adapter binds instead (INTEGRATION.md shows the
 * adapter): plain pointers and sizes, caller-owned host buffers, `int` error codes, no C++ or torch types.
 * Each entry point cites the reference interface it replaces (paths relative to the reference tree).
 *
 * Threading: a context is bound to one GPU and one HIP stream. Mutating calls (build/add/remove/set*) come from
 * one thread (the engine's update thread, like World::transformEntity delegates). lmx_cull() may be issued for
 * several views per frame; each view has its own result slot (`view` argument), so results of different views
 * never alias — the analogue of the reference returning an independent CullResult list per call
 * (src/renderer/culling_system.cpp:321-369).
 */
#ifndef LUMIX_MI355_H
#define LUMIX_MI355_H

#include <stddef.h>
#include <stdint.h>

#include "lmx_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LMX_API __attribute__((visibility("default")))

typedef struct LmxContext LmxContext;

enum {
	LMX_OK = 0,
	LMX_ERR_INVALID_ARGUMENT = 1,
	LMX_ERR_NO_DEVICE = 2,     /* no HIP device / kernels cannot run: the product never falls back to the CPU */
	LMX_ERR_HIP = 3,           /* a HIP runtime call failed; see lmx_last_error() */
	LMX_ERR_OUT_OF_MEMORY = 4,
	LMX_ERR_CAPACITY = 5,      /* caller buffer too small / too many views, frusta or types */
	LMX_ERR_NOT_BUILT = 6,     /* operation needs data that has not been uploaded yet */
	LMX_ERR_BUSY = 7           /* every result slot is held by a caller that has not released it (lmx_cull_view_acquire timed out) */
};

enum {
	LMX_MAX_FRUSTA = 8,  /* frusta tested in one pass over the spheres (config 5: 2 x 4 shadow cascades) */
	LMX_MAX_TYPES = 8,   /* renderable types 0..7 (RenderableTypes::COUNT == 5, render_module.h:293-301) */
	LMX_MAX_VIEWS = 8    /* concurrent result slots (main view + 4 cascades + light query, pipeline.cpp:1252-1285,3380) */
};

/* ---- context -------------------------------------------------------------------------------------------- */
LMX_API int lmx_ctx_create(int device, LmxContext** out);
LMX_API void lmx_ctx_destroy(LmxContext* ctx);
/* ONE context per World, shared by every adapter of that World. The reference creates one CullingSystem per RenderModuleImpl
 * (src/renderer/render_module.cpp:3569) and moves its spheres from onModelInstanceMoved (:1544-1554) on the SAME object; here the
 * culling set, the transform hierarchy and the skinning tables of one World live in one LmxContext, and the engine-side pieces that
 * are created independently of each other - the CullingSystem replacement inside RenderModuleImpl, the plugin's IModule - find it
 * through a process-wide registry keyed by the World's address. acquire creates the context on first use and counts references;
 * release destroys it with the last one. `key` is only compared, never dereferenced. */
LMX_API int lmx_ctx_acquire_shared(const void* key, int device, LmxContext** out);
LMX_API void lmx_ctx_release_shared(LmxContext* ctx);
/* A context is not re-entrant. Adapters that share one (update thread: add / set / propagate; render jobs: cull) serialise on the
 * context's own recursive lock. */
LMX_API void lmx_ctx_lock(LmxContext* ctx);
LMX_API void lmx_ctx_unlock(LmxContext* ctx);
/* Last error text of this context (or of the failed lmx_ctx_create when ctx == NULL). Never NULL. */
LMX_API const char* lmx_last_error(const LmxContext* ctx);
/* Use an external HIP stream (hipStream_t as void*) for all launches/copies instead of the context's own non-blocking
 * stream. NULL means the legacy default (null) stream, as everywhere in HIP. */
LMX_API int lmx_ctx_set_stream(LmxContext* ctx, void* hip_stream);
LMX_API int lmx_ctx_synchronize(LmxContext* ctx);
/* Per-kernel timing with HIP events on the launch stream (bench.py's roofline leg). kernel ids: LMX_K_*. */
enum {
	LMX_K_CULL_CLASSIFY = 0,
	LMX_K_CULL_SPHERES = 1,
	LMX_K_XFORM_LEVEL = 2,
	LMX_K_SPHERE_REFRESH = 3,
	LMX_K_POSE_PALETTE = 4,
	LMX_K_SKIN_VERTICES = 5,
	LMX_K_CULL_DYNAMIC = 6,
	LMX_K_SORT_KEYS = 7,
	LMX_K_ANIM_UPDATE = 8,
	LMX_K_CULL_PATCH = 9, /* the copy + k_apply_patches launch that ships queued add / remove / set records */
	LMX_K_COUNT = 10
};
LMX_API int lmx_profile_enable(LmxContext* ctx, int enable);
LMX_API int lmx_profile_reset(LmxContext* ctx);
/* Synchronizes, then returns accumulated device time (ms) and launch count of one kernel since the last reset. */
LMX_API int lmx_profile_get(LmxContext* ctx, int kernel_id, double* total_ms, uint64_t* launches);

/* ---- culling: CullingSystem, src/renderer/culling_system.h:58-77 -------------------------------------------
 * Device layout: a static set of spheres sorted by (type, is_big, cell) in SoA chunks of 64, plus an unsorted dynamic set for
 * entities that move every frame or were added since the static set was last compacted; see DESIGN.md. */

/* Bulk CullingSystem::add (culling_system.cpp:131-157) of n entities; replaces any previous content.
 * pos_xyz: n x 3 doubles (world position), type < LMX_MAX_TYPES, entity >= 0 and unique. */
LMX_API int lmx_cull_build(LmxContext* ctx, uint32_t n, const int32_t* entity, const uint8_t* type, const double* pos_xyz,
	const float* radius);
/* Incremental interface, same semantics and the same O(1) cost as the virtuals of CullingSystem (culling_system.cpp:131-258).
 * Changes are staged on the host mirror and reach the GPU as small patch records at the next lmx_cull()/lmx_cull_flush():
 * an in-cell move rewrites one sphere, a removal leaves a tombstone, an added or re-celled entity lives in the unsorted
 * dynamic set until the static set is compacted (automatically once the overflow exceeds 1/8 of it, or lmx_cull_compact). */
LMX_API int lmx_cull_add(LmxContext* ctx, int32_t entity, uint8_t type, const double pos[3], float radius);
LMX_API int lmx_cull_remove(LmxContext* ctx, int32_t entity);
LMX_API int lmx_cull_set(LmxContext* ctx, int32_t entity, const double pos[3], float radius);
LMX_API int lmx_cull_set_position(LmxContext* ctx, int32_t entity, const double pos[3]);
LMX_API int lmx_cull_set_radius(LmxContext* ctx, int32_t entity, float radius);
LMX_API int lmx_cull_get_radius(LmxContext* ctx, int32_t entity, float* out_radius);
LMX_API int lmx_cull_is_added(LmxContext* ctx, int32_t entity); /* 1 / 0 */
LMX_API int lmx_cull_flush(LmxContext* ctx);
/* The same add / set / remove for n entities in one ABI crossing (hosts that pay per call). */
LMX_API int lmx_cull_add_many(LmxContext* ctx, uint32_t n, const int32_t* entity, const uint8_t* type, const double* pos_xyz, const float* radius);
LMX_API int lmx_cull_set_many(LmxContext* ctx, uint32_t n, const int32_t* entity, const double* pos_xyz, const float* radius);
LMX_API int lmx_cull_remove_many(LmxContext* ctx, uint32_t n, const int32_t* entity);
/* Re-sort the static set now, folding in every entity added / re-celled since the last compaction and dropping tombstones
 * (O(n): a loading-screen operation; never needed for correctness). */
LMX_API int lmx_cull_compact(LmxContext* ctx);
/* Bookkeeping of the incremental path: entities in the sorted set, bound to the world hierarchy, waiting in the overflow, and
 * tombstones left in the sorted layout. Host-only, no synchronisation. */
LMX_API int lmx_cull_update_stats(LmxContext* ctx, uint32_t* n_static, uint32_t* n_dynamic_bound, uint32_t* n_overflow, uint32_t* n_tombstones);
/* LMX_CULL_OPT_ASYNC_COMPACTION: what the worker has done so far (host-only, no sync; any pointer may be null). *state: 0 idle,
   1 requested, 2 running, 3 a re-sorted set is ready for the next flush, 4 the last job failed (lmx_last_error), -1 the option is off.
   jobs = re-sorts finished, swaps = sets traded, ops_replayed_at_swaps = operations the update thread replayed inside those flushes,
   log_drains = jobs that only brought the second copy's mirror up to date (the operation log had passed 2^20 entries with no re-sort due:
   in-cell moves patch the sorted set in place). */
LMX_API int lmx_cull_async_stats(LmxContext* ctx, int* state, uint64_t* jobs, uint64_t* swaps, uint64_t* ops_replayed_at_swaps, uint64_t* log_drains);
/* Number of resident spheres / occupied (cell,type,is_big) groups of the static set / capacity of one frustum's output
 * row in units of 64 ids: a bound output buffer needs 64 * n_chunks ids per frustum. Compacts the static set first (the cell
 * count is that of the sorted layout). */
LMX_API int lmx_cull_stats(LmxContext* ctx, uint32_t* n_entities, uint32_t* n_cells, uint32_t* n_chunks);
/* The per-tile tables of the device layout: *cell_key_bytes = 8 (cell keys relative to their tile's box; any scene whose tiles span <= 65535 cell
 * indices per axis) or 16, *table_bytes = cell keys + tile tables + chunk headers a cull of the whole static set reads besides spheres and ids
 * (2048-sphere tiles). Either pointer may be NULL. (Environment LMX_CULL_WIDE_KEYS, read when a layout is built, forces 16-byte keys: tests.) */
LMX_API int lmx_cull_layout_info(LmxContext* ctx, uint32_t* cell_key_bytes, uint64_t* table_bytes);

/* CullingSystem::cull(frustum[, type]) (culling_system.cpp:310-369) for n_frusta <= LMX_MAX_FRUSTA frusta in one
 * call (e.g. the 4 shadow cascades + main view of a frame). type == LMX_TYPE_ALL (0xff) culls every type. Asynchronous
 * on the context stream; the result stays in HBM in slot `view` until the next lmx_cull() on the same slot. */
LMX_API int lmx_cull(LmxContext* ctx, uint32_t view, const LmxShiftedFrustum* frusta, uint32_t n_frusta, uint8_t type);
/* How many frusta of a call are tested per pass over the static set (1..LMX_MAX_FRUSTA; 0 = automatic, the default: all of them
 * in ONE launch when the set holds <= 32 M spheres, frustum by frustum above). Measured on one MI355X (profiles/r04): at 10 M spheres one
 * pass beats eight in every regime - a frame's 6 views 39 against 66 us, 8 cascades over a sparse scene 84 against 119 us, every sphere
 * tested against 8 frusta 141 against 298 us; at 100 M the 1-frustum launches win (a frame's 6 views 185 against 207 us): the
 * several-frusta kernel walks smaller tiles with more LDS per block, and the blocks that only reject their tile are what a launch over
 * 100 M spheres is made of. */
LMX_API int lmx_cull_set_pass_width(LmxContext* ctx, uint32_t frusta_per_pass);
/* Kernel tuning knobs (no reference twin; results never depend on them). */
enum {
	LMX_CULL_OPT_TILE_VARIANT = 0,            /* form of the 1-frustum kernel (4 waves x 8 chunks, 2048-sphere tiles): -1 auto (default: 4 when the frustum's box overlaps < 25 % of the set's box, else 1), 1 = streaming (4 chunks' loads in flight per wave, non-temporal loads, ids staged in LDS), 4 = all 8 chunks' loads in flight (few surviving tiles: the launch is their latency) */
	/* (1 was the tile-level-test mode of rounds 2-5: only its default form - one plane per lane in wave 0, verdict through LDS - is left) */
	LMX_CULL_OPT_MAX_SHARDS = 2,              /* output shards (reservation counters) per renderable type, 1..64 (default 64) */
	LMX_CULL_OPT_COUNTER_PAD = 3,             /* 32-bit words between two shard counters, 1..64 (default 32 = one 128-byte line each) */
	LMX_CULL_OPT_AUTO_COMPACTION = 4,         /* 1 (default): the sorted set is re-built (O(n log n) on the host, ~0.5 s at 10 M) when the overflow set exceeds max(COMPACTION_MIN, n/8) or the tombstones max(COMPACTION_MIN, n/4); 0: never on its own - the host calls lmx_cull_compact when a hitch is acceptable */
	LMX_CULL_OPT_DEVICE_OWNS_BOUND = 5,       /* 1: lmx_cull_set / set_position / set_radius on an entity bound with lmx_world_bind_culling are accepted and dropped - lmx_world_propagate has already refreshed its sphere on the device (what the adapter sets while it replays the engine's `transformed` delegates, whose RenderModuleImpl::onModelInstanceMoved would repeat the refresh per entity on the host); 0 (default): they apply */
	LMX_CULL_OPT_OVERFLOW_RESERVE = 6,        /* n >= 0 (default 0): free slots kept in the unsorted overflow set for entities added (or moved to another cell) after the sorted set was built. The reference's add / remove never stall (culling_system.cpp:131-190); here an add takes a free overflow slot in O(1), and only when a type's overflow region is FULL is the whole overflow set laid out again (a host pass + re-upload: milliseconds at 10^6 entities). With AUTO_COMPACTION = 0 and a reserve that covers the churn between two lmx_cull_compact calls (a loading screen, a streaming boundary) no frame ever pays for either; the overflow entities cost k_cull_dynamic's ~350 instead of ~27 instructions per cull until then */
	LMX_CULL_OPT_ASYNC_COMPACTION = 7,        /* 1: the re-sort of the sorted set runs on a worker thread, on a second complete copy of the sets (host mirror + device arrays: twice the memory), and the copy trades places with the live set inside a flush in O(1) + a replay of the operations of the last frame or two: no frame pays the O(n) step (the reference's add / remove / set never stall, culling_system.cpp:131-258). Every effective add / remove / set* / bind is also appended to a 40-byte operation log. Switching it on copies the host mirror once (O(n)); lmx_cull_build and lmx_cull_compact stay synchronous and re-seed the copy. Results are the same id sets as without it; 0 (default): the re-sort happens inside the flush that finds the thresholds exceeded */
	LMX_CULL_OPT_COMPACTION_MIN = 8,          /* n >= 1 (default 65536): the floor of AUTO_COMPACTION's thresholds - a re-sort is considered once the overflow set holds more than max(n, static / 8) entities or the sorted set more than max(n, static / 4) tombstones. The default keeps small scenes from re-sorting at all (their overflow set costs microseconds per cull); a host whose scene is small but churns for hours lowers it, and the tests do, to reach the re-sort - synchronous or on the worker - with a few thousand entities */
	LMX_CULL_OPT_MAP_ZERO_COPY = 9            /* 1 (default): lmx_cull_map_* of a view whose lists held <= 1 M ids last frame has the pack kernel write the record straight into the pinned host buffer (posted PCIe writes, no copy command behind the kernel); 0: always pack on the device + one DMA copy; n > 1: the same with a threshold of n ids. Results do not depend on it */
};
LMX_API int lmx_cull_set_option(LmxContext* ctx, int option, int value);
/* counts[f * LMX_MAX_TYPES + t] = visible entities of type t for frustum f (synchronizes the stream). */
LMX_API int lmx_cull_counts(LmxContext* ctx, uint32_t view, uint32_t* counts /* [n_frusta][LMX_MAX_TYPES] */);
/* Copies the visible ids of (frustum, type) to the host: the content of the CullResult pages of that type
 * (culling_system.h:17-56). Order is unspecified, as in the reference (job scheduling + mutexed page list). */
LMX_API int lmx_cull_read(LmxContext* ctx, uint32_t view, uint32_t frustum, uint8_t type, int32_t* out_ids, uint32_t cap,
	uint32_t* out_count);
/* All types of one frustum at once: out_counts[LMX_MAX_TYPES], ids of type 0 first, then type 1, ... (two host waits in
 * total; what an adapter that rebuilds CullResult pages needs). */
LMX_API int lmx_cull_read_all(LmxContext* ctx, uint32_t view, uint32_t frustum, int32_t* out_ids, uint32_t cap, uint32_t* out_counts);
/* The same list with (normally) one host wait: *out_ids points into pinned host memory owned by the library (type 0's ids first,
 * out_counts[LMX_MAX_TYPES] per type), valid until the next lmx_cull_map_all on this view. The copy is enqueued before the count is
 * known, sized from the previous call on this view; a list that outgrew it costs a second wait. */
LMX_API int lmx_cull_map_all(LmxContext* ctx, uint32_t view, uint32_t frustum, const int32_t** out_ids, uint32_t* out_counts);
/* The same packed record [LMX_MAX_TYPES counts | ids, types back to back] left in HBM, stream-ordered, NO host wait: "one cull incl.
 * compaction" for a consumer that stays on the device. Valid until the next lmx_cull_pack_device on this view. */
LMX_API int lmx_cull_pack_device(LmxContext* ctx, uint32_t view, uint32_t frustum, const int32_t** d_record, uint32_t* record_words);
/* The same for ALL frusta of the view's last lmx_cull in one go - the frame's views (the reference culls 4 shadow cascades + the main
 * view + a light query per frame, pipeline.cpp:1036-1045, :1252-1258) as one lmx_cull with n_frusta frusta and ONE host wait:
 * out_ids[f] / out_counts[f * LMX_MAX_TYPES + t] per frustum, valid until the next cull or map on this view. */
LMX_API int lmx_cull_map_many(LmxContext* ctx, uint32_t view, uint32_t n_frusta, const int32_t** out_ids, uint32_t* out_counts);
/* lmx_cull_map_many in two halves, for render jobs that cull DIFFERENT views at the same time (the reference culls the views of a frame
 * from concurrent jobs: pipeline.cpp:1036-1041). lmx_cull_map_begin enqueues the pack + copy of the view's records behind its cull and
 * records an event; like every entry point that enqueues, it needs the context lock (lmx_ctx_lock). lmx_cull_map_end waits for THAT
 * view's event and hands out the pointers into the view's pinned buffer: it may be called WITHOUT the lock, concurrently with any
 * other call that does not use the same view - the host wait and the consumer's copy out of the buffer overlap other threads'
 * enqueues. host/gpu_culling_system.h is the caller; at most LMX_MAX_VIEWS results are in flight at a time, which the slot
 * reservation below enforces. */
LMX_API int lmx_cull_map_begin(LmxContext* ctx, uint32_t view, uint32_t n_frusta);
LMX_API int lmx_cull_map_end(LmxContext* ctx, uint32_t view, uint32_t n_frusta, const int32_t** out_ids, uint32_t* out_counts);
/* Result slots that cannot alias. CullingSystem::cull hands every caller an independent list (culling_system.cpp:321-369; callers
 * pipeline.cpp:1036-1045, :3380, editor/scene_view.cpp:144), here a caller reads its ids out of the slot's pinned record AFTER the
 * enqueue, outside the context lock - so a slot must not be culled into again before that reader is done. lmx_cull_view_acquire
 * reserves a free slot (round robin); with all LMX_MAX_VIEWS reserved it waits up to timeout_ms for a release and then returns
 * LMX_ERR_BUSY. lmx_cull_view_release frees the slot once the ids have been copied out. Both are thread safe, take no context
 * lock, and must NOT be called with the context lock held (a waiter would keep the enqueuing threads out). Callers that name
 * their view slots themselves (one thread, or one fixed slot per thread) do not need them. */
LMX_API int lmx_cull_view_acquire(LmxContext* ctx, uint32_t* view, uint32_t timeout_ms);
LMX_API int lmx_cull_view_release(LmxContext* ctx, uint32_t view);
/* Device-side view of a result for GPU consumers (sort keys, RCCL all-gather): ids of (frustum, type) start at
 * d_ids + type_offsets[type] and number d_counts[frustum * LMX_MAX_TYPES + type]. All pointers are device memory
 * except type_offsets (host, LMX_MAX_TYPES entries, in ids). The cull kernels leave the visible ids in up to a few hundred
 * per-shard windows (lmx_cull_device_shards); this call gathers them into one contiguous list per type first (two small
 * launches, once per cull result). */
LMX_API int lmx_cull_device_result(LmxContext* ctx, uint32_t view, uint32_t frustum, const int32_t** d_ids, const uint32_t** d_counts,
	uint32_t* type_offsets, uint32_t* capacity);

/* The raw form of a result, as the cull kernels leave it: shard s of the frustum holds d_counts[s * count_stride] ids of
 * renderable type d_shard_type[s] at d_ids + d_window_start[s]. The analogue of the reference's unordered list of CullResult
 * pages; consumers that can walk several windows skip the gather of lmx_cull_device_result. All pointers are device memory. */
typedef struct LmxCullShards {
	const int32_t* d_ids;
	const uint32_t* d_counts;
	uint32_t count_stride;
	const uint32_t* d_window_start;
	const uint8_t* d_shard_type;
	uint32_t n_shards;
} LmxCullShards;
LMX_API int lmx_cull_device_shards(LmxContext* ctx, uint32_t view, uint32_t frustum, LmxCullShards* out);

/* Lets slot `view` write its (contiguous, per-type) result into caller-owned DEVICE memory (e.g. a buffer that is then handed to an RCCL
 * all-gather) instead of library-owned buffers: d_ids holds ids_capacity int32 (>= n_frusta * 64 * n_chunks of
 * lmx_cull_stats at cull time), d_counts LMX_MAX_FRUSTA * LMX_MAX_TYPES uint32. Frustum f's ids start at
 * d_ids + f * (64 * n_chunks). Passing NULL pointers restores the library-owned buffers. */
LMX_API int lmx_cull_bind_output(LmxContext* ctx, uint32_t view, void* d_ids, size_t ids_capacity, void* d_counts);

/* ---- exchange: multi-GPU all-gather of visible-entity lists (no reference twin: the reference is single-process) -------------
 * One process per GPU, every rank's context holds a disjoint share of the entities (partition by the reference's cell hash, or by
 * index). Culling needs no communication; the one exchange step per frame is a single ncclAllGather (RCCL over xGMI) of a fixed-size
 * record per rank: per frustum of the frame LMX_MAX_TYPES counts followed by that frustum's capacity of ids (types packed back to back).
 * The record is written by the cull's gather kernels IN PLACE (where the collective would put it), frames are double-buffered. RCCL is
 * loaded on first use.
 * How the step runs (environment LMX_EXCHANGE_MODE, read once at lmx_exchange_create):
 *   auto (default)  PER FRAME SHAPE (number of frusta): at the first frame of a shape - and again when the shape's record has grown or shrunk
 *                   2x since - 32 all-gathers of the record that shape ships are timed (a collective every rank runs in the same call); the
 *                   collective goes on a SIDE stream (the next cull overlaps it, ~16 us more host work per step) when one gather takes longer
 *                   than LMX_EXCHANGE_OVERLAP_US (default 16), else INLINE behind the pack kernel
 *   inline / side   force one of the two (LMX_EXCHANGE_INLINE=1 / 0 of earlier rounds still works)
 *   p2p             no collective in the step: every rank stores the USED part of its record (counts + the ids it has) into each peer's
 *                   receive buffer through hipIpc mappings and raises a sequence flag; consumers wait on the device with a bounded spin
 *                   (LMX_EXCHANGE_P2P_TIMEOUT_MS, default 2000). A peer that never shows up costs the frame: lmx_exchange_wait returns
 *                   LMX_ERR_BUSY and the exchange refuses further steps (a rank cannot fall back to a collective on its own - the others
 *                   would not join it; destroy and re-create). One node (<= 8 ranks). Needs fine-grained device memory for the receive
 *                   slots: lmx_exchange_create FAILS (LMX_ERR_NO_DEVICE) where the runtime has none. Opt-in: unmeasured over xGMI.
 * lmx_exchange_info says which mode the last frame's shape takes, the gather time it was taken on, and why.
 * Capacities: a frame of n frusta ships n sub-records [counts | cap[f] ids]. They start as ids_per_rank / n each (or lmx_exchange_set_caps)
 * and - frames of >= 2 frusta by default, LMX_EXCHANGE_AUTO_CAPS=1 / 0: all / no frames - follow the lists: cap[f] of frame j is derived from
 * the largest list ANY rank gathered for frustum f in frame j - 2 (+20 %, 256-id grain; the same numbers on every rank, so the ranks agree
 * without another collective). A sub-record that overflowed is flagged in lmx_exchange_stats and regrown when its slot comes round again. */
typedef struct LmxExchange LmxExchange;
/* ncclGetUniqueId: rank 0 calls this and ships the 128 bytes to the other ranks over any side channel (the engine's network layer,
 * a file, torch.distributed in bench.py). */
LMX_API int lmx_exchange_unique_id(void* out_id_128_bytes);
LMX_API int lmx_exchange_create(LmxContext* ctx, int rank, int world, const void* unique_id_128_bytes, uint32_t ids_per_rank, LmxExchange** out);
LMX_API void lmx_exchange_destroy(LmxExchange* x);
/* One frame of this rank: lmx_cull(frustum, type) into result slot 0 / 1 (alternating; the views of the same index are used), then
 * the all-gather of its record, asynchronously. *out_slot identifies the frame for the calls below. */
LMX_API int lmx_exchange_cull(LmxExchange* x, const LmxShiftedFrustum* frustum, uint8_t type, uint32_t* out_slot);
/* The frame's views in ONE collective: cull n_frusta frusta (<= LMX_MAX_FRUSTA; every rank the same number) in one pass over the rank's
 * spheres and all-gather n_frusta x [LMX_MAX_TYPES counts | cap[f] ids] per rank (capacities: above) - config 5's 8 cascades are one
 * exchange step, not eight. lmx_exchange_cull is the n_frusta = 1 case. lmx_exchange_read_many reads one (rank, frustum) sub-record. */
LMX_API int lmx_exchange_cull_many(LmxExchange* x, const LmxShiftedFrustum* frusta, uint32_t n_frusta, uint8_t type, uint32_t* out_slot);
LMX_API int lmx_exchange_read_many(LmxExchange* x, uint32_t slot, int rank, uint32_t frustum, uint32_t* out_counts, int32_t* out_ids, uint32_t cap);
LMX_API int lmx_exchange_wait(LmxExchange* x, uint32_t slot);
/* Device view: rank r's record = d_records + r * record_words (sub-records as lmx_exchange_layout says: counts, then ids). sum(counts) >
 * the sub-record's capacity means that list was clipped. gathered_event: hipEvent_t recorded after the collective (P2P mode: a device-side
 * consumer must also see lmx_exchange_wait succeed - a frame whose bounded wait gave up holds stale records). */
LMX_API int lmx_exchange_result(LmxExchange* x, uint32_t slot, const int32_t** d_records, uint32_t* record_words, void** gathered_event);
LMX_API int lmx_exchange_read(LmxExchange* x, uint32_t slot, int rank, uint32_t* out_counts, int32_t* out_ids, uint32_t cap);
/* mode: 0 = all-gather on the cull stream, 1 = on a side stream, 2 = P2P stores - of the frame shape that ran last; gather_us: one all-gather
 * of that shape's record as last timed (< 0: never - the mode was forced, or no frame has run); why: a static string. Any pointer may be NULL. */
LMX_API int lmx_exchange_info(LmxExchange* x, int* mode, double* gather_us, const char** why);
/* Capacities (ids per frustum, sum <= ids_per_rank) of the sub-records of frames of n_frusta frusta from now on. EVERY rank makes the same call
 * between the same two frames. keep_fixed != 0: no regrowth from the gathered counts for this shape. */
LMX_API int lmx_exchange_set_caps(LmxExchange* x, uint32_t n_frusta, const uint32_t* caps, int keep_fixed);
/* One all-gather of the record frames of n_frusta frusta ship now, timed over 32 gathers (us; the record's words per rank). A COLLECTIVE: every
 * rank calls it at the same point; waits for everything the exchange has in flight. Not in P2P mode. */
LMX_API int lmx_exchange_time_gather(LmxExchange* x, uint32_t n_frusta, double* out_us, uint32_t* out_record_words);
/* Layout of the gathered records of `slot`: rank r's record at r * record_words, its sub-record f at offsets[f] from there (LMX_MAX_TYPES
 * counts, then caps[f] ids). caps / offsets: LMX_MAX_FRUSTA entries. Any pointer may be NULL. */
LMX_API int lmx_exchange_layout(LmxExchange* x, uint32_t slot, uint32_t* n_frusta, uint32_t* caps, uint32_t* offsets, uint32_t* record_words);
/* What the frame in `slot` shipped and what of it was used (waits for the frame's gather). */
typedef struct LmxExchangeStats {
	uint32_t n_frusta;
	uint32_t record_words;               /* words per rank in the receive buffer = what a collective form ships per rank */
	uint32_t caps[LMX_MAX_FRUSTA];        /* ids per sub-record */
	uint32_t max_visible[LMX_MAX_FRUSTA]; /* the largest list any rank saw, per frustum */
	uint32_t overflow_mask;              /* bit f: some rank's list of frustum f was clipped in this frame */
	uint32_t used_words_own;             /* counts + ids this rank's record actually holds */
	uint32_t used_words_max;             /* the same, largest over the ranks */
	int32_t mode;                        /* 0 inline, 1 side, 2 p2p: how this frame was shipped */
	uint64_t bytes_shipped_per_peer;     /* record_words * 4 (collective forms), used_words_own * 4 (P2P) */
	uint64_t bytes_used;                 /* used_words_own * 4 */
	double gather_us;                    /* one all-gather of this shape's record as last timed (< 0: never) ... */
	uint32_t gather_us_record_words;     /* ... and the record size it was timed on */
	uint32_t reserved;
} LmxExchangeStats;
LMX_API int lmx_exchange_stats(LmxExchange* x, uint32_t slot, LmxExchangeStats* out);

/* ---- world transforms: World hierarchy, src/engine/world.cpp:255-282 ---------------------------------------
 * The reference propagates eagerly (one recursive DFS per setTransform). The batch form: stage new root/local
 * transforms, then lmx_world_propagate() recomputes child.world = parent.world.compose(child.local)
 * (core/math.cpp:801-807) level by level for exactly the nodes those DFS walks would visit (written nodes and their
 * subtrees); results equal the DFS bit for bit. */

/* n entities, entity index = array index (EntityRef::index). parent[i] = -1 for roots.
 * transforms[i] = world transform for roots, local transform (Hierarchy::local_transform) for children. */
LMX_API int lmx_world_build(LmxContext* ctx, uint32_t n, const int32_t* parent, const LmxTransform* transforms);
/* The same from a live World: world_transforms[i] = World::getTransforms()[i] for EVERY entity, local_transforms[i] =
 * Hierarchy::local_transform for entities with a parent (ignored for the others). Nothing is recomputed until something is
 * written: a stored local that went through Transform::computeLocal does not reproduce the stored world transform bit for
 * bit, and the reference only recomposes the subtrees it visits (world.cpp:255-282) - so does lmx_world_propagate. */
LMX_API int lmx_world_build_with_world(LmxContext* ctx, uint32_t n, const int32_t* parent, const LmxTransform* local_transforms,
	const LmxTransform* world_transforms);
/* World::setParent (world.cpp:619-701): the child keeps its world transform and its local becomes
 * Transform::computeLocal(parent world, child world); new_parent < 0 detaches the child. Cycles are rejected
 * (LMX_ERR_INVALID_ARGUMENT, the reference logs "Hierarchy can not contain a cycle."). Editing operation: rebuilds the
 * slot order on the host; culling bindings survive. */
LMX_API int lmx_world_set_parent(LmxContext* ctx, int32_t new_parent, int32_t child);
/* World::getLocalTransform (world.cpp:756-766) for every entity: Hierarchy::local_transform, or the world transform of
 * entities without a parent. */
LMX_API int lmx_world_read_local_transforms(LmxContext* ctx, LmxTransform* out, uint32_t n);
/* Host-side Transform::compose / Transform::computeLocal (core/math.cpp:801-816), bit-identical to the engine's. */
LMX_API int lmx_transform_compose(const LmxTransform* a, const LmxTransform* b, LmxTransform* out);
LMX_API int lmx_transform_compute_local(const LmxTransform* parent, const LmxTransform* child, LmxTransform* out);
/* World::setTransform for roots / World::setLocalTransform for children (world.cpp:337-342, 741-753), staged until
 * lmx_world_propagate. Exactly like the reference, a child written this way gets world = parent.compose(local) and its STORED
 * local is then re-derived as Transform::computeLocal(parent, world) (world.cpp:266-269 reached through :704-712) - which is
 * what later frames compose with. The writes of one batch are applied as if issued ancestors-first (level order). */
LMX_API int lmx_world_set_transforms(LmxContext* ctx, uint32_t n, const int32_t* entity, const LmxTransform* transforms);
/* World::setTransform (world-space write, world.cpp:337-342) on any entity: an entity with a parent keeps the given world
 * transform and its local becomes computeLocal(parent world, world); its subtree follows at the next lmx_world_propagate. */
LMX_API int lmx_world_set_world_transforms(LmxContext* ctx, uint32_t n, const int32_t* entity, const LmxTransform* transforms);
/* Same, with both arrays already in device memory (entity indices must be valid; they are not checked). */
LMX_API int lmx_world_set_transforms_device(LmxContext* ctx, uint32_t n, const void* d_entity, const void* d_transforms);
/* RenderModuleImpl::onModelInstanceMoved binding (render_module.cpp:1544-1554): after propagation the culling
 * sphere of entity[i] becomes (world.pos, model_radius[i] * maximum(scale.x, scale.y, scale.z)); model_radius[i] < 0
 * binds the position only and keeps the radius (onDecalMoved / onPointLightMoved -> setPosition, :1568-1592). Bound entities are
 * moved to the culling system's dynamic set (unsorted, re-binned implicitly by the cull kernel every frame). */
LMX_API int lmx_world_bind_culling(LmxContext* ctx, uint32_t n, const int32_t* entity, const float* model_radius);
LMX_API int lmx_world_propagate(LmxContext* ctx);
/* World::getTransforms() (world.h:65): AoS Transform[n] indexed by entity. */
/* Bone attachments (RenderModuleImpl::m_bone_attachments, updateBoneAttachment, renderer/render_module.cpp:377-404; called for
 * every attachment of a parent whose pose changed, :1964-1981): attachment i moves `entity[i]` to
 * transform(parent_entity[i]) . (absolute pose of bone bone_index[i] of skin instance skin_instance[i] . relative[i]), keeping the
 * entity's own scale. Attached entities must be hierarchy roots (the reference moves them with World::setTransform) and may not
 * hang off another attachment. lmx_world_update_bone_attachments runs after lmx_skin_run (absolute poses, pose write-back on)
 * and before lmx_world_propagate, which carries the attached entities' subtrees and culling spheres along. lmx_world_build and
 * lmx_world_set_parent drop the attachment table (slots are renumbered). */
LMX_API int lmx_world_set_bone_attachments(LmxContext* ctx, uint32_t n, const int32_t* entity, const int32_t* parent_entity,
	const uint32_t* skin_instance, const uint32_t* bone_index, const LmxLocalRigidTransform* relative);
LMX_API int lmx_world_update_bone_attachments(LmxContext* ctx);
LMX_API int lmx_world_read_transforms(LmxContext* ctx, LmxTransform* out, uint32_t n);
/* The entities whose world transform changed in the LAST lmx_world_propagate (staged writes, their subtrees, bone attachments): what
 * World::transformEntity would have visited (world.cpp:255-282). lmx_world_track_moved(1) makes propagate collect them on the device
 * (one compaction pass instead of the plain clear of the marks); lmx_world_read_moved copies the list - entity indices and their new
 * world transforms, in no particular order - to the host. out_n is the number moved, also when it exceeds `cap` (LMX_ERR_CAPACITY:
 * nothing is copied; read everything with lmx_world_read_transforms instead). The hand-back of a frame costs what moved, not n. */
/* Tuning knob, results never depend on it. LMX_WORLD_OPT_FUSED_LEVELS (default 1): hierarchies of up to 16 levels are propagated by ONE
 * launch - a block owns the subtrees of a run of consecutive roots, walks them level by level (a parent comes out of the block's own
 * cache, not HBM) and also clears the marks, collects the moved list and refreshes the bound culling spheres; 0 = one launch per level
 * + one each for the marks / the moved list and the spheres (rounds 1-3; also what deeper or very lop-sided hierarchies use). */
enum { LMX_WORLD_OPT_FUSED_LEVELS = 0 };
LMX_API int lmx_world_set_option(LmxContext* ctx, int option, int value);
LMX_API int lmx_world_track_moved(LmxContext* ctx, int enable);
LMX_API int lmx_world_read_moved(LmxContext* ctx, int32_t* entity, LmxTransform* transforms, uint32_t cap, uint32_t* out_n);

/* ---- skinning: Pose / Model, src/renderer/pose.cpp:63-134, src/renderer/model.cpp:103-137 -------------------- */

/* Model bones: parents[i] < i (model.cpp:381-384), parents[root] = -1; bind = Model::Bone::transform (model
 * space). The inverse bind pose is derived at load like model.cpp:404-413. Returns a model id. */
LMX_API int lmx_skin_add_model(LmxContext* ctx, uint32_t n_bones, const int16_t* parents, const LmxLocalRigidTransform* bind,
	int32_t first_nonroot, uint32_t* out_model);
/* Mesh vertices + Mesh::Skin (model.h:81-84). Returns a mesh id. */
LMX_API int lmx_skin_add_mesh(LmxContext* ctx, uint32_t n_verts, const float* positions_xyz, const LmxSkin* skin, uint32_t* out_mesh);
/* Instances: instance i uses model[i] and mesh[i]. Replaces the instance table. */
LMX_API int lmx_skin_set_instances(LmxContext* ctx, uint32_t n, const uint32_t* model, const uint32_t* mesh);
/* Relative poses (what AnimationModule writes before Pose::computeAbsolute): instances back to back,
 * positions n_bones x 3 floats, rotations n_bones x 4 floats per instance. */
LMX_API int lmx_skin_upload_poses(LmxContext* ctx, const float* positions, const float* rotations, size_t n_bones_total);
/* Same, from device memory (device-to-device copy on the context stream). */
LMX_API int lmx_skin_upload_poses_device(LmxContext* ctx, const void* d_positions, const void* d_rotations, size_t n_bones_total);
/* Pose::blend(rhs, weight) (renderer/pose.cpp:30-41, nlerp core/math.cpp:677-691) for every instance: the library's relative
 * poses (uploaded or sampled by lmx_anim_update) are blended with a second pose set of the same layout - what the Animator's
 * blend stack does with each layer's pose (animation_module.cpp:602-636). weight <= 0.001 is a no-op, weight is clamped to 1. */
LMX_API int lmx_skin_blend_poses(LmxContext* ctx, const float* positions, const float* rotations, size_t n_bones_total, float weight);
LMX_API int lmx_skin_blend_poses_device(LmxContext* ctx, const void* d_positions, const void* d_rotations, size_t n_bones_total, float weight);
/* Zero-copy variant: lmx_skin_run reads the relative poses from this device memory every time it runs (the animation
 * system's output buffer) and writes the absolute poses to the library's own arrays. NULL pointers end the borrowing. */
LMX_API int lmx_skin_set_pose_source_device(LmxContext* ctx, const void* d_positions, const void* d_rotations, size_t n_bones_total);
/* Arithmetic of the vertex blend/transform (model.cpp:103-109). LMX_SKIN_FUSED (default): products fused into the adds
 * (v_fma_f32), results within 1e-5 relative of the reference CPU path - the tolerance BASELINE's north star sets for
 * skinned positions. LMX_SKIN_EXACT: separate multiplies and adds in the reference's order, bit-identical to it.
 * Poses and palettes are always bit-identical. LMX_SKIN_DQS: the dual-quaternion blend of the reference's own GPU skinning
 * path instead (the SKINNED branch of data/shaders/surface_base.hlsli:196-217 + transformByDualQuat, common.hlsli:632-636):
 * model-space positions from the dual-quaternion palette; a different deformation than linear blending by design. */
enum { LMX_SKIN_FUSED = 0, LMX_SKIN_EXACT = 1, LMX_SKIN_DQS = 2 };
LMX_API int lmx_skin_set_mode(LmxContext* ctx, int mode);
/* How runs of consecutive instances that share a mesh are skinned (evaluateSkin, model.cpp:103-109, is per vertex: the grouping is
 * ours). LMX_SKIN_OPT_INSTANCES_PER_BLOCK = I in {1, 2, 4, 8, 16}: k_skin_multi - one block stages the palettes of I instances ONCE
 * (the 16 bank columns of the LDS palette hold I instances x 16 / I copies) and streams the run's vertex records past them; 0:
 * k_skin_shared - one instance at a time against a register-resident vertex tile (rounds 2 / 3). Default 2. Same results in every form
 * (bit-identical positions in LMX_SKIN_EXACT). Takes effect at the next lmx_skin_run. */
enum { LMX_SKIN_OPT_INSTANCES_PER_BLOCK = 0 };
LMX_API int lmx_skin_set_option(LmxContext* ctx, int option, int value);
/* Pose::computeAbsolute -> computeSkinMatrices -> evaluateSkin for every instance; outputs stay in HBM. */
LMX_API int lmx_skin_run(LmxContext* ctx);
LMX_API int lmx_skin_read_vertices(LmxContext* ctx, uint32_t instance, float* out_xyz, uint32_t cap_verts);
/* Instances [first, first + n) in ONE copy: their outputs are consecutive in HBM in instance order (out = sum of the meshes' vertex
 * counts, in order). What a renderer-side consumer (or a full-size parity check) reads instead of n single-instance calls. */
LMX_API int lmx_skin_read_vertices_range(LmxContext* ctx, uint32_t first_instance, uint32_t n_instances, float* out_xyz, size_t cap_verts);
/* The skinned positions where they lie: device pointer to 3 floats per vertex, instances back to back (valid until the next
 * lmx_skin_set_instances), and the total vertex count. Stream-ordered after lmx_skin_run on the context stream. */
LMX_API int lmx_skin_device_output(LmxContext* ctx, const float** d_xyz, size_t* n_verts_total);
LMX_API int lmx_skin_read_palette(LmxContext* ctx, uint32_t instance, LmxMatrix* out, uint32_t cap_bones);
LMX_API int lmx_skin_read_pose(LmxContext* ctx, uint32_t instance, float* out_pos, float* out_rot, uint32_t cap_bones);
/* The reference keeps the absolute pose in the instance's Pose (Pose::is_absolute, pose.cpp:133) for bone attachments and
 * the next consumer; lmx_skin_run stores it next to the palette by default (28 B per bone). A renderer that only consumes
 * palettes / vertices can switch the store off: lmx_skin_read_pose then fails with LMX_ERR_NOT_BUILT, and uploaded (not
 * borrowed) relative poses stay valid for the next run. */
LMX_API int lmx_skin_set_pose_writeback(LmxContext* ctx, int enable);
/* Also emit the dual-quaternion palette of the reference's own GPU skinning path (PipelineImpl::computeSkeletonDualQuats,
 * renderer/pipeline.cpp:2680-2745): per bone a DualQuat {r.xyzw, d.xyzw} (core/math.h:257-260, 32 B) of pose * inverse bind. */
LMX_API int lmx_skin_enable_dual_quats(LmxContext* ctx, int enable);
LMX_API int lmx_skin_read_dual_quats(LmxContext* ctx, uint32_t instance, float* out /* 8 floats per bone */, uint32_t cap_bones);

/* ---- host mirror of core/geometry.cpp frustum construction (per view, not per entity) ----------------------- */
/* Viewport::getFrustum() (geometry.cpp:793-818). */
LMX_API int lmx_viewport_frustum(const LmxViewport* viewport, LmxShiftedFrustum* out);
/* ShiftedFrustum::computePerspective / computeOrtho, 7-argument overloads (geometry.cpp:502-533, 390-409). */
LMX_API int lmx_frustum_perspective(const double pos[3], const float dir[3], const float up[3], float fov, float ratio, float near_d,
	float far_d, LmxShiftedFrustum* out);
LMX_API int lmx_frustum_ortho(const double pos[3], const float dir[3], const float up[3], float width, float height, float near_d,
	float far_d, LmxShiftedFrustum* out);

/* ------------------------------------------------------------------------------------------------------------------
 * Animation sampling: AnimationModuleImpl::updateAnimable (animation/animation_module.cpp:439-472) for every instance of
 * the skin instance table: pose = Model::getRelativePose (renderer/model.cpp:226-237), overwritten by
 * Animation::getRelativePose without a bone mask (animation/animation.cpp:117-204, :294-311: constant tracks, bit-packed
 * translation / rotation tracks of two neighbouring frames, lerp / simd_nlerp, blended with `weight` when it is < 0.9999),
 * then the Animable's time advances by time_delta modulo the animation length (:458-470). The relative poses land in the
 * pose buffers lmx_skin_run consumes, so animation -> absolute pose -> palette -> vertices never leaves HBM.
 * ------------------------------------------------------------------------------------------------------------------ */
LMX_API int lmx_anim_add(LmxContext* ctx, const LmxAnimation* animation, uint32_t* out_animation);
/* Model::Bone::relative_transform of a model registered with lmx_skin_add_model (the pose an un-animated bone keeps). */
LMX_API int lmx_anim_set_model_pose(LmxContext* ctx, uint32_t model, const LmxLocalRigidTransform* relative, uint32_t n_bones);
/* One Animable per skin instance: animation id (LMX_ANIM_NONE = the instance keeps its model's relative pose) and
 * Animable::time in Time units (1 / 32768 s, animation/animation.h:17-41). Times then live and advance on the device. */
LMX_API int lmx_anim_set_animables(LmxContext* ctx, uint32_t n_instances, const uint32_t* animation, const uint32_t* time);
LMX_API int lmx_anim_set_weight(LmxContext* ctx, float weight); /* SampleContext::weight, default 1 */
LMX_API int lmx_anim_update(LmxContext* ctx, float time_delta);
/* AnimationModuleImpl::updateAnimator's pose work for every instance in one launch (animation_module.cpp:602-636): Model::getRelativePose
 * into the pose, then evalBlendStack's SAMPLE instructions in order (controller.cpp:267-293, getPose :142-157) - instance i owns
 * samples[first_sample[i] .. first_sample[i + 1]), first_sample has n_instances + 1 entries. An instruction whose animation does not
 * fit the instance's skeleton is skipped, as Animation::getRelativePose does (animation.cpp:120). The controller's node graph that
 * emits the instructions, bone masks and IK instructions stay on the CPU / out of scope. lmx_skin_run then does
 * Pose::computeAbsolute and the palette. Host arrays; copied before the call returns. */
LMX_API int lmx_anim_eval_blend_stacks(LmxContext* ctx, uint32_t n_instances, const uint32_t* first_sample, const LmxBlendSample* samples);
LMX_API int lmx_anim_read_times(LmxContext* ctx, uint32_t* time, uint32_t n_instances);
/* The relative pose of an instance after lmx_anim_update (before lmx_skin_run turns it into the absolute one). */
LMX_API int lmx_anim_read_pose(LmxContext* ctx, uint32_t instance, float* out_pos, float* out_rot, uint32_t cap_bones);

/* ------------------------------------------------------------------------------------------------------------------
 * Sort keys: PipelineImpl::createSortKeys (renderer/pipeline.cpp:3789-3968), the consumer of the visible list (SURVEY.md
 * §8f rank 1). Runs on the device straight from a cull result (no read-back of the list): LOD selection per visible MESH
 * entity (fp64 squared distance :3876, Model::getLODMeshIndices model.h:173-179, the ModelInstance::lod transition
 * :3937-3957), one (SortKey, SortValue) pair per skinned / moved / depth-sorted mesh and per decal (:83-142), auto-instancer
 * groups by mesh sort key (AutoInstancer::add :505-523) and one AUTOINSTANCED pair per non-empty group (:3958-3968), the
 * list of model instances whose pose must be processed this frame (Pose::frame stamp :3889-3898) and the instances whose
 * material overrides are dirty (:3879-3882). The reference runs one AutoInstancer per worker thread; the device run is the
 * single-worker case (instancer index 0). Pair order is unspecified, as in the reference (per-worker page lists).
 * ------------------------------------------------------------------------------------------------------------------ */
LMX_API int lmx_keys_set_models(LmxContext* ctx, const LmxKeysModel* models, uint32_t n_models, const uint8_t* mesh_types, uint32_t n_meshes);
/* Model instances by entity index (RenderModule::getModelInstances): model < 0 = entity has no model instance. Entity e's
 * MeshMaterial span starts at mesh_materials[material_offset[e]] (ModelInstance::mesh_materials, indexed by mesh index).
 * A call with a different n_entities drops decal tables uploaded for the previous entity range. */
LMX_API int lmx_keys_set_instances(LmxContext* ctx, uint32_t n_entities, const int32_t* model, const uint32_t* material_offset,
	const LmxMeshMaterial* mesh_materials, uint32_t n_mesh_materials, const float* lod, const uint8_t* flags, const uint8_t* dirty,
	const uint32_t* pose_frame);
/* Decal / curve-decal materials by entity index: Material::getSortKey() and getLayer() (pipeline.cpp:83-89). */
LMX_API int lmx_keys_set_decals(LmxContext* ctx, uint32_t n_entities, const uint32_t* decal_sort_key, const uint8_t* decal_layer,
	const uint32_t* curve_sort_key, const uint8_t* curve_layer);
/* World::getTransforms()[e].pos by entity index, from the host (may be called every frame: ModelInstance::lod and Pose::frame keep
 * the state the device advanced; lmx_keys_set_instances resets them to the uploaded values) ... */
LMX_API int lmx_keys_set_positions(LmxContext* ctx, const double* xyz, uint32_t n_entities);
/* ... or read in place from the world hierarchy of this context (after lmx_world_propagate); 0 = back to the uploaded array. */
LMX_API int lmx_keys_bind_world(LmxContext* ctx, int enable);
/* LMX_KEYS_OPT_SLOT_ORDER (default 1): the per-entity tables are mirrored in the order of the culling system's sorted set, the culls
   also emit the slot of every visible id, and the key kernels read entities of the sorted set through that mirror (sequential instead
   of one random cache line per table and entity). ModelInstance::lod / Pose::frame of those entities then live in the mirror and are
   handed back to the entity-indexed records whenever a slot dies (removal, move to the overflow set, re-sort) and before
   lmx_keys_read_state. 0: entity-indexed tables only. Results do not depend on it.
   LMX_KEYS_OPT_SPLIT_STATE (default 2; with SLOT_ORDER): 1 = ModelInstance::lod and Pose::frame of the sorted set's entities - the two fields
   the key kernel WRITES - live in a dense 8-byte-per-slot array instead of inside the 64-byte mirror records: an update then dirties 8
   bytes of a line its neighbours update too, not one sector per visible entity. 2: the whole mirror is a structure of arrays (one dense array
   per field the key kernel reads, 42 bytes per slot instead of the 64-byte records). Results do not depend on it.
   LMX_KEYS_OPT_WALK_SHARDS (default 1): the key kernels read the visible ids out of the per-shard windows the cull kernels wrote
   (lmx_cull_device_shards); 0: the windows are gathered into one list per type first (two more launches). Results do not depend on it
   (the order of the unsorted pairs / of the renderables inside an instancer group is unspecified either way).
   LMX_KEYS_OPT_BLOCK_RANKS (default 1): for max_sort_key < 4096 the auto-instancer's groups are built without global atomics - every
   block of the key kernel owns a row of per-key counts and every record carries its rank inside its block; 0 (and larger key
   ranges): privatised global counters + one cursor atomic per record. Results do not depend on it. */
enum { LMX_KEYS_OPT_SLOT_ORDER = 0, LMX_KEYS_OPT_SPLIT_STATE = 1, LMX_KEYS_OPT_WALK_SHARDS = 2, LMX_KEYS_OPT_BLOCK_RANKS = 3 };
LMX_API int lmx_keys_set_option(LmxContext* ctx, int option, int value);
/* createSortKeys for (view, frustum) of the last lmx_cull on that slot. max_sort_key = Renderer::getMaxSortKey(). Async. */
LMX_API int lmx_keys_run(LmxContext* ctx, uint32_t view, uint32_t frustum, const LmxKeysView* kv, uint32_t max_sort_key);
/* Sorter::pack + radix sort by key (pipeline.cpp:411-440, radixSort): pairs in ascending key order, stable. */
LMX_API int lmx_keys_sort(LmxContext* ctx);
typedef struct LmxKeysCounts {
	uint32_t pairs;            /* (key, value) pairs pushed to the sorter, AUTOINSTANCED ones included */
	uint32_t instanced;        /* renderables added to auto-instancer groups */
	uint32_t groups;           /* non-empty auto-instancer groups */
	uint32_t poses;            /* model instances handed to the pose processor */
	uint32_t dirty;            /* model instances queued for a material-override refresh */
	uint32_t overflow;         /* != 0: an output buffer was too small (never with library-sized buffers) */
} LmxKeysCounts;
LMX_API int lmx_keys_counts(LmxContext* ctx, LmxKeysCounts* out); /* synchronizes the stream */
LMX_API int lmx_keys_read_pairs(LmxContext* ctx, uint64_t* keys, uint64_t* values, uint32_t cap);
/* Auto-instancer groups as CSR: group k (= mesh sort key k) holds values[offsets[k] .. offsets[k + 1]); offsets has
 * max_sort_key + 2 entries. Order inside a group is unspecified. */
LMX_API int lmx_keys_read_instancer(LmxContext* ctx, uint32_t* offsets, uint64_t* values, uint32_t cap_values);
LMX_API int lmx_keys_read_poses(LmxContext* ctx, int32_t* entities, uint32_t cap);
LMX_API int lmx_keys_read_dirty(LmxContext* ctx, int32_t* entities, uint32_t cap);
/* ModelInstance::lod and Pose::frame after the run (both are updated in place on the device). */
LMX_API int lmx_keys_read_state(LmxContext* ctx, float* lod, uint32_t* pose_frame, uint32_t n_entities);
/* Device pointers of the last run for GPU consumers: pairs (keys, values, count on the device). Valid until the next lmx_keys_run /
 * lmx_keys_set_*: fetch them again after every run - d_count alternates between two addresses from run to run (the next run's counters
 * are zeroed by this run's kernels, not by a fill), and the buffers are re-reserved when the key range or the copy count grows. */
LMX_API int lmx_keys_device_pairs(LmxContext* ctx, const uint64_t** d_keys, const uint64_t** d_values, const uint32_t** d_count);

/* ------------------------------------------------------------------------------------------------------------------
 * Scene ingest: the part of a serialized World (World::serialize / deserialize, engine/world.cpp:837-1043, current
 * WorldVersion, LZ4-compressed) that feeds lmx_world_build - entity transforms and Hierarchy records.
 * Host-only, needs no context or device. Entities keep the indices of the file (EntityMap = identity: loading into an empty
 * world). Module payloads (the blob's tail) are not read; legacy headers and uncompressed versions are rejected.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct LmxWorldBlobInfo {
	uint32_t version, flags;                  /* WorldVersion, WorldSerializeFlags */
	uint32_t n_modules;
	uint32_t uncompressed_size, compressed_size;
	uint32_t n_entities, max_entity_index;    /* valid entities in the file and the largest EntityRef::index among them */
	uint32_t n_names, n_hierarchy;
} LmxWorldBlobInfo;
LMX_API int lmx_world_blob_info(const void* data, size_t size, LmxWorldBlobInfo* out);
/* parent[e] / transforms[e] for e < n_slots (n_slots > max_entity_index) exactly as lmx_world_build takes them: roots carry their
 * world transform, entities with a parent their Hierarchy::local_transform. Optional: world[e] = the serialized m_transforms[e]
 * of every entity, valid[e] = 1 for entities present in the file (the others are detached identity placeholders). */
LMX_API int lmx_world_blob_read(const void* data, size_t size, uint32_t n_slots, int32_t* parent, LmxTransform* transforms, LmxTransform* world,
	uint8_t* valid);

/* The "renderer" module's payload inside a serialized World (RenderModuleImpl::serialize / deserialize,
 * renderer/render_module.cpp:962-976, :1225-1250): what the skinning / attachment path needs from a scene file - which entity
 * carries which model (serializeModelInstances :650-679) and the bone attachments (serializeBoneAttachments :581-590, records of
 * bone name hash, entity, parent entity, relative LocalRigidTransform). Module payloads are not size-prefixed (world.cpp:877-882):
 * the reader finds the module by its header (NUL-terminated name + version) after the hierarchy records and walks the renderer's
 * sections in order - cameras, model instances, lights, terrains, particle systems, bone attachments, probes, decals, instanced
 * models, procedural geometries. RenderModuleVersion 16..18 (what the reference's shipped demo/maps carry; LATEST = 18). Checked on every
 * .unv under the reference's demo/maps: each payload is consumed to the byte where the next module's header starts
 * (tests/test_world_blob.py). Host-only. */
/* Where module `name`'s payload starts in the decompressed blob (right behind its header) and its serialized version;
 * LMX_ERR_INVALID_ARGUMENT when the file has no such module. */
LMX_API int lmx_world_blob_find_module(const void* data, size_t size, const char* name, uint32_t* payload_offset, int32_t* version);
typedef struct LmxRenderBlobInfo {
	int32_t version;                       /* RenderModuleVersion of the payload */
	uint32_t payload_offset, payload_size; /* in the decompressed blob; payload_size = 0 when procedural geometries are present (not walked) */
	uint32_t n_cameras, n_model_instance_slots, n_model_instances, n_point_lights, n_environments, n_terrains, n_particle_systems,
		n_bone_attachments, n_environment_probes, n_reflection_probes, n_decals, n_curve_decals, n_instanced_models, n_procedural_geometries;
	uint32_t model_paths_size;             /* bytes of the NUL-separated model path table */
} LmxRenderBlobInfo;
typedef struct LmxBlobBoneAttachment {
	uint64_t bone_name_hash;               /* BoneNameHash = StableHash of the bone's name (core/hash.h:44-76): Model::getBoneIndex resolves it;
	                                          version <= 17 stored a bone index instead (low 32 bits here) */
	int32_t entity, parent_entity;
	LmxLocalRigidTransform relative;
} LmxBlobBoneAttachment;
LMX_API int lmx_render_blob_info(const void* data, size_t size, LmxRenderBlobInfo* out);
LMX_API int lmx_render_blob_read_bone_attachments(const void* data, size_t size, uint32_t cap, LmxBlobBoneAttachment* out);
/* flags[e] = ModelInstance::Flags of entity e (0: no model instance; VALID = 4), path_offset[e] = offset of its model's path in
 * `paths` (0xffffffff: none), for e < n_slots (>= n_model_instance_slots); paths: the file's path table, model_paths_size bytes. */
LMX_API int lmx_render_blob_read_model_instances(const void* data, size_t size, uint32_t n_slots, uint8_t* flags, uint32_t* path_offset, char* paths,
	uint32_t paths_cap);

LMX_API const char* lmx_version(void);

#ifdef __cplusplus
}
#endif

#endif
