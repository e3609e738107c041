// lmx_kernels.h — host-callable launchers of the gfx950 kernels (defined in *_kernels.hip) + shared device layouts.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lmx_math.h"
#include "lmx_types.h"

// Dynamic LDS of a launch. (tests/hostsim compiles these sources for the CPU and supplies its own definition: there the block is
// a buffer of the simulated device, not an `extern __shared__` array.)
#ifndef LMX_DYNAMIC_LDS
#define LMX_DYNAMIC_LDS(T, name) extern __shared__ T name[]
#endif

namespace lmx {

constexpr int MAX_FRUSTA = 8;
constexpr int MAX_TYPES = 8;
constexpr uint32_t CHUNK = 64;          // spheres per chunk = one wavefront
constexpr uint32_t TILE_ALIGN = 4096;   // type ranges are padded to this many spheres (largest tile of any variant)
constexpr uint32_t CELL_DEAD = 0x80000000u;

// One (cell index, type, is_big) group. meta = type | is_big << 8 | CELL_DEAD.
struct CellKey { int32_t ix, iy, iz; uint32_t meta; };
// The same group in 8 bytes, relative to its TILE's box (TileBox::lo): xy = dx | dy << 16, zf = dz | flags << 16 with flags bit 0 = is_big,
// bit 15 = dead. The per-tile cell tables are the largest part of what k_cull_tile reads besides the spheres and ids themselves (16 B x ~1 M
// cells per 10 M entities, padded per tile: 18.7 of the launch's 224.5 MB); the type is not needed on the device (the tile's shard says it).
struct PackedCellKey { uint32_t xy, zf; };
constexpr uint32_t PACKED_CELL_BIG = 1u << 16, PACKED_CELL_DEAD = 1u << 31;
// Header of a 64-sphere chunk: cell slot of its first sphere, bit l of flags = "sphere l starts the next cell".
struct ChunkHdr { uint32_t cell, pad; uint64_t flags; };

// Per-type ranges in the padded slot space of one set (static or dynamic) + the type's output shards.
struct TypeTable {
	uint32_t ent_start[MAX_TYPES];   // first slot of the type (static set: multiple of TILE_ALIGN, dynamic set: of 2048)
	uint32_t ent_end[MAX_TYPES];     // end of the padded range
	uint32_t shard_first[MAX_TYPES]; // global index of the first output shard the type's tiles of this set reserve from
	uint32_t shard_n[MAX_TYPES];     // number of shards: block b (TILE_ALIGN slots) of the type uses shard_first + b % shard_n
};

struct FrustaArg { DevFrustum f[MAX_FRUSTA]; };

struct CullDeviceView {
	const float4* spheres;       // [n_padded] {rel.x, rel.y, rel.z, radius}, cell-relative fp32 (culling_system.cpp:100)
	const int32_t* ids;          // [n_padded] entity index, -1 for padding and removed entities
	const ChunkHdr* hdr;         // [n_padded / 64]
	uint32_t n_padded;
	// per tile-size variant k (tile = 4096 >> k): tile-major cell keys + {first cell, n cells} + cell-index box per tile
	const CellKey* tile_cells[3];   // 16-byte keys - or, keys_packed, 8-byte ones (PackedCellKey) behind the same pointers
	bool keys_packed;               // every tile's cells lie within 65535 cell indices of its box's low corner on every axis (any real scene)
	const uint32_t* tile_tab[3];
	const TileBox* tile_box[3];
	uint32_t tile_cap[3];
	// per tile: {output shard, start of that shard's window in an ids row} - what a surviving tile needs to reserve and write (a function of the
	// tile's type range and of the output layout: looked up, it was ~90 scalar instructions incl. an integer modulo, and two dependent loads)
	const uint2* tile_out[3];
};

// Where a cull writes. The visible ids of (frustum f, shard s) go to ids[f * stride + win_base[s] + k], k < the shard's counter
// counts[f * cnt_frustum_stride + s * cnt_pad]; counters sit cnt_pad words apart (own cache line: same-line atomics serialise).
// Every launch also clears counts_next[0 .. n_zero) (the counters of the following cull on this view, ping-pong).
struct CullOut {
	int32_t* ids;
	uint32_t stride;
	const uint32_t* win_base;
	uint32_t* counts;
	uint32_t cnt_pad, cnt_frustum_stride;
	uint32_t* counts_next;
	uint32_t n_zero;
	// profiling only (lmx_profile_enable): the dispatch's OWN begin / end timestamps go into these events (hipExtLaunchKernelGGL) -
	// events recorded around the launch also time ~3 us of command processing per pair, 7 % of a 40 us kernel
	hipEvent_t ev_start = nullptr, ev_stop = nullptr;
	// optional, parallel to `ids` (same stride, same windows): the STATIC-SET SLOT every visible id came from, -1 for ids of the dynamic
	// set. Consumers that keep per-entity tables in slot order (the sort-key kernels) read them with the locality of the sorted set.
	int32_t* slots = nullptr;
};

// k_cull_tile over the static set's slots [ent_begin, ent_end) (multiples of TILE_ALIGN). `variant` picks the form of the 1-frustum kernel
// (both: 4 waves x 8 chunks, 2048-sphere tiles): 1 = streaming (4 chunks' loads in flight per wave), 4 = all 8 in flight; ignored for n_frusta > 1.
size_t cull_tile_lds_bytes(int n_frusta, uint32_t cell_cap);
uint32_t cull_tile_size(int n_frusta, int variant);
hipError_t launch_cull_tile(hipStream_t s, const CullDeviceView& v, uint32_t ent_begin, uint32_t ent_end, const TypeTable& tt,
	const FrustaArg& fr, int n_frusta, const CullOut& out, int variant);

// Dynamic set: entities whose transform changes every frame (bound to the world hierarchy) and entities added / re-celled since
// the last compaction of the static set are kept UNSORTED as world position (fp64) + radius + id. Their cell, cell-relative
// position and per-cell class are recomputed per entity per cull with exactly the arithmetic CullingSystem::set + cullInternal
// would apply (cell_of, classify_cell, sphere_visible), so no re-binning is ever needed.
struct DynDeviceView {
	double* px; double* py; double* pz;
	float* radius;
	int32_t* ids; // -1 = free slot
	uint32_t n_padded;
};
hipError_t launch_dyn_carry_over(hipStream_t s, const DynDeviceView& from, const DynDeviceView& to, const int32_t* new_slot_of_entity, uint32_t n_entities);
hipError_t launch_cull_dynamic(hipStream_t s, const DynDeviceView& d, uint32_t slot_begin, uint32_t slot_end, const TypeTable& dyn_tt,
	const FrustaArg& fr, int n_frusta, const CullOut& out);
uint32_t cull_dynamic_tile(int n_frusta, uint32_t n_slots);

// Patch records staged by the host between two culls (CullingSystem::add / remove / set* are O(1): culling_system.cpp:131-258)
struct PatchSphere { uint32_t slot; float x, y, z, radius; };                 // static set: in-cell move / radius change
struct PatchId { uint32_t slot; int32_t id; };                                 // static set: removal (tombstone, id = -1)
struct PatchDyn { uint32_t slot; int32_t id; float radius; uint32_t pad; double px, py, pz; }; // dynamic set: add / remove / set
// (a tombstone also clears TILE_DENSE of the tiles that hold its slot, in all three tile-size variants)
hipError_t launch_apply_patches(hipStream_t s, float4* spheres, int32_t* ids, TileBox* const tile_box[3], const DynDeviceView& d, const PatchSphere* ps,
	uint32_t n_ps, const PatchId* pi, uint32_t n_pi, const PatchDyn* pd, uint32_t n_pd);

// Per-(frustum, type) totals and per-(frustum, shard) offsets of the consolidated lists: totals[f * MAX_TYPES + t], pref[f * n_shards + s]
// (offset of shard s inside its type's consolidated list). shard_type[s] = renderable type of shard s. packed_start (optional):
// [f * MAX_TYPES + t] = start of type t when the types of a frustum are packed back to back.
hipError_t launch_cull_finalize(hipStream_t s, const uint32_t* counts, uint32_t cnt_pad, uint32_t cnt_frustum_stride, const uint8_t* shard_type,
	uint32_t n_shards, uint32_t n_frusta, uint32_t* totals, uint32_t* pref, uint32_t* packed_start);
// One contiguous list per (frustum, type): dst[f * dst_stride + type_start[f * type_start_stride + type] + pref + k] = src[f * src_stride + win_base[s] + k],
// clipped to dst_cap ids per frustum row
// finalize + consolidate in one launch, into the packed record [MAX_TYPES counts | ids, types back to back] of each of n_frusta frusta
// (frustum f: ids row src + f * src_stride, counters counts + f * cnt_stride, record header / dst + f * rec_stride; strides in words)
hipError_t launch_cull_pack(hipStream_t s, const int32_t* src, const uint32_t* win_base, const uint32_t* counts, uint32_t cnt_pad, const uint8_t* shard_type,
	uint32_t n_shards, uint32_t max_shard_cap, uint32_t* header, int32_t* dst /* behind the header */, uint32_t dst_cap, uint32_t n_frusta = 1,
	uint32_t src_stride = 0, uint32_t cnt_stride = 0, uint32_t rec_stride = 0);
// the same with a place and a capacity PER record: record f = [MAX_TYPES counts | lay.cap[f] ids] at rec_base + lay.off[f] (words)
struct PackLayout {
	uint64_t off[MAX_FRUSTA];
	uint32_t cap[MAX_FRUSTA];
};
hipError_t launch_cull_pack_layout(hipStream_t s, const int32_t* src, const uint32_t* win_base, const uint32_t* counts, uint32_t cnt_pad, const uint8_t* shard_type,
	uint32_t n_shards, uint32_t max_shard_cap, int32_t* rec_base, const PackLayout& lay, uint32_t n_frusta, uint32_t src_stride, uint32_t cnt_stride);
hipError_t launch_cull_consolidate(hipStream_t s, const int32_t* src, uint32_t src_stride, const uint32_t* win_base, const uint32_t* counts,
	uint32_t cnt_pad, uint32_t cnt_frustum_stride, const uint8_t* shard_type, const uint32_t* type_start /* device */, uint32_t type_start_stride,
	const uint32_t* pref, uint32_t n_shards, uint32_t n_frusta, uint32_t max_shard_cap, int32_t* dst, uint32_t dst_stride, uint32_t dst_cap, const int32_t* src2 = nullptr /* a second array with the same windows, gathered by the same launch */, int32_t* dst2 = nullptr);

// ---- world transforms ------------------------------------------------------------------------------------
struct WorldDevice {
	// slot order = (level, parent slot); all arrays [n]
	double* lpx; double* lpy; double* lpz; float4* lrot; float* lsx; float* lsy; float* lsz; // local (roots: unused)
	double* wpx; double* wpy; double* wpz; float4* wrot; float* wsx; float* wsy; float* wsz; // world
	const int32_t* parent_slot; // -1 for roots
	uint8_t* dirty;             // XF_* mark per slot: what was staged since the last propagation
};
enum : uint8_t { XF_CLEAN = 0, XF_SET_LOCAL = 1, XF_SET_WORLD = 2, XF_MOVED = 4 };
enum { XF_STAGE_RAW = 0, XF_STAGE_RAW_WORLD = 1, XF_STAGE_SET_LOCAL = 2, XF_STAGE_SET_WORLD = 3 };
// world[s] = compose(world[parent_slot[s]], local[s]) for s in [first, first + n)
hipError_t launch_xform_level(hipStream_t s, const WorldDevice& w, uint32_t first, uint32_t n);
// out[entity_of_slot[s]] = AoS Transform (56 B) for s in [0, n)
hipError_t launch_xform_export(hipStream_t s, const WorldDevice& w, const int32_t* entity_of_slot, uint32_t n, void* out_transforms);
// every level in ONE launch (k_xform_subtree): run r of consecutive roots owns slots [table[r * n_levels + l], table[(r + 1) * n_levels + l]) of
// level l. The block also clears its marks, appends its moved nodes to the hand-back lists (count != nullptr) and refreshes the culling
// spheres of its bound entities (bound_dyn_of_slot != nullptr: dynamic-set index per slot or 0xffffffff, model radius per slot).
constexpr uint32_t XF_SUBTREE_NODES = 1024;      // nodes a run aims at (4 per thread of its block)
constexpr uint32_t XF_SUBTREE_MAX_RUN = 16384;   // ... and the most one root's subtree may hold before the per-level launches take over
constexpr uint32_t XF_SUBTREE_MAX_LEVELS = 16;
struct XformSubtree {
	const uint32_t* table;
	uint32_t n_levels;
	const int32_t* entity_of_slot;
	uint32_t cap;
	int32_t* out_entity;
	void* out_tr; // Transform AoS, 56 B
	uint32_t* count;
	const uint32_t* bound_dyn_of_slot;
	const float* bound_radius_of_slot;
	double *dyn_px, *dyn_py, *dyn_pz;
	float* dyn_radius;
};
hipError_t launch_xform_subtree(hipStream_t s, const WorldDevice& w, const XformSubtree& a, uint32_t n_runs);
// append {entity, world transform} of every slot marked XF_MOVED to the lists (ballot-compacted, one atomic per wave) and clear all marks
hipError_t launch_xform_collect_moved(hipStream_t s, const WorldDevice& w, const int32_t* entity_of_slot, uint32_t n, uint32_t cap, int32_t* out_entity,
	void* out_transforms, uint32_t* count);
// stage transforms (AoS LmxTransform, device memory) into the SoA arrays; mode = XF_STAGE_* (see k_xform_scatter)
hipError_t launch_xform_scatter(hipStream_t s, const WorldDevice& w, const int32_t* slot_of_entity, const int32_t* entity,
	const void* transforms, uint32_t n, int mode);
// culling refresh for bound entities (RenderModuleImpl::onModelInstanceMoved, render_module.cpp:1544-1554): the dynamic
// set's position / radius of bound entity i become (world.pos, model_radius[i] * maximum(scale.x, scale.y, scale.z))
hipError_t launch_sphere_refresh(hipStream_t s, const WorldDevice& w, const uint32_t* bound_slot, const uint32_t* bound_dyn,
	const float* model_radius, double* dyn_px, double* dyn_py, double* dyn_pz, float* dyn_radius, uint32_t n);

// ---- skinning --------------------------------------------------------------------------------------------
struct SkinInstance {
	uint32_t bone_offset;  // into pose arrays / palette (in bones)
	uint32_t n_bones;
	uint32_t model_offset; // into parents / inverse bind arrays (in bones)
	int32_t first_nonroot;
	uint32_t vert_offset;  // into mesh vertex arrays
	uint32_t n_verts;
	uint32_t out_offset;   // into output vertex array (in vertices)
	uint32_t max_depth;    // deepest bone level of the model (root = 0)
	uint32_t lv_items_offset; // into level_items: the model's bones >= first_nonroot sorted by depth, as bone | parent << 16
	uint32_t lv_off_offset;   // into level_off: max_depth + 1 offsets (bones of depth d = [off[d-1], off[d]))
};
// k_skin_shared work item: instances [first_inst, first_inst + count) share mesh and bone count; vertices [v_begin, v_end) of it
// (the tile's records live in the mesh table with TILE-LOCAL bone indices: rec_offset = the mesh's first record there, in vertices;
// tile_bones[bones_at .. + n_tile_bones) = the model bones the tile references, in local-index order)
struct SkinChunk { uint32_t first_inst, count, v_begin, v_end, rec_offset, bones_at, n_tile_bones, pad; };
struct PoseGroup { uint32_t first_inst; uint32_t count; }; // consecutive instances of one model, count <= 4 / 2 / 1 by bone count (<= 64 / 128 / 196)
#ifndef LMX_POSE_GROUP_SHIFT
#define LMX_POSE_GROUP_SHIFT 2
#endif
constexpr int POSE_GROUP_SHIFT = LMX_POSE_GROUP_SHIFT; // log2 of the instances per group of the smallest bone-count class; halves per class (k_pose_palette, skin_kernels.hip)
constexpr uint32_t POSE_GROUP_CAP = 1u << POSE_GROUP_SHIFT;
// ---- bone attachments (xform_kernels.hip) ----
struct BoneAttachDevice { uint32_t slot, parent_slot, skin_instance, bone; float rel_pos[3]; float rel_rot[4]; };
hipError_t launch_bone_attach(hipStream_t s, const WorldDevice& w, const BoneAttachDevice* att, uint32_t n, const SkinInstance* inst,
	const float* pose_pos, const float4* pose_rot);

// ---- animation sampling (anim_kernels.hip) ----
struct AnimDevice { // one Animation resource: offsets into the concatenated tables of AnimTables
	float fps;
	uint32_t frame_count, length, tfs_bits, rfs_bits;
	uint32_t max_bone;                 // m_max_accessed_bone_index
	uint32_t src_off;                  // 2 * (max_bone + 1) int32: translation / rotation source per bone
	uint32_t ct_off, tt_off, cr_off, rt_off;
	uint32_t tstream_off, rstream_off; // bytes, 8-byte aligned
	int32_t root_translation_track, root_rotation_track;
	uint32_t root_off;                 // frames, into root_translations / root_rotations
};
struct AnimTables {
	const int32_t* src;
	const LmxAnimConstTranslation* const_translations;
	const LmxAnimTranslationTrack* translations;
	const LmxAnimConstRotation* const_rotations;
	const LmxAnimRotationTrack* rotations;
	const uint8_t *translation_stream, *rotation_stream;
	const float* root_translations;
	const float4* root_rotations;
};
hipError_t launch_anim_update(hipStream_t s, const SkinInstance* inst, uint32_t n_inst, const AnimDevice* anims, const AnimTables& t,
	const uint32_t* anim_of_instance, uint32_t* time_of_instance, float time_delta, float weight, const float* model_rel_pos,
	const float4* model_rel_rot, float* pose_pos, float4* pose_rot);
hipError_t launch_anim_blend_stack(hipStream_t s, const SkinInstance* inst, uint32_t n_inst, const AnimDevice* anims, const AnimTables& t, uint32_t n_anims,
	const LmxBlendSample* samples, const uint32_t* first_sample, const float* model_rel_pos, const float4* model_rel_rot, float* pose_pos, float4* pose_rot);

// ---- createSortKeys (keys_kernels.hip) ----
// Counter words. {pairs, recs} and {poses, dirty} are two 64-bit cells on their own 128-byte lines: k_keys_mesh reserves a tile's four
// output ranges with two returning atomics on two lines instead of four on one (measured: no change of the kernel's 124 us per
// million visible entities - it is bound by its random 64-byte gathers and line write-backs, see DESIGN.md - but half the atomics).
enum { KEYS_N_PAIRS = 0, KEYS_N_RECS = 1, KEYS_N_POSES = 32, KEYS_N_DIRTY = 33, KEYS_OVERFLOW = 64, KEYS_N_GROUPS = 96, KEYS_COUNTERS = 128 };
struct KeysViewDevice { // what the kernels read of a LmxKeysView, bucket_map as built at pipeline.cpp:3802-3812
	uint32_t bucket_map[255];
	uint8_t layer_to_bucket[255];
	uint8_t is_shadow;
	double cam[3], ref[3];
	float lod_multiplier_rcp, time_delta;
	uint32_t frame_number;
};
// ModelInstance {model, mesh_materials, lod, flags, dirty, pose->frame} + World::getTransforms()[e].pos, one cache-line-friendly record
struct alignas(64) KeysInstance {
	double pos[3];
	int32_t model;            // -1: the entity has no model instance
	uint32_t material_offset;
	float lod;
	uint32_t pose_frame;
	uint8_t flags, dirty;
	uint8_t pad[22];
};
static_assert(sizeof(KeysInstance) == 64, "one record per 64-byte sector");
struct KeysSlotState { float lod; uint32_t pose_frame; };
// (LMX_KEYS_OPT_SPLIT_STATE = 2, the default since round 4: createSortKeys 138 -> 129 us on the round-3 driver box) the slot-ordered mirror as a structure of arrays: the 42 bytes of a record the key kernel reads,
// one dense array per field, so that a wave's load of a field is one contiguous run instead of 64 pieces at a 64-byte stride
#ifndef LMX_KEYS_MAX_COPIES
#define LMX_KEYS_MAX_COPIES 256 // copies of the instancer's per-key counters (power of two, <= 256: 8 bits of rec_key)
#endif
struct KeysSoA { double *px, *py, *pz; int32_t* model; uint32_t* material_offset; uint16_t* flags_dirty; /* flags | dirty << 8 */ };
#ifndef LMX_KEYS_SPLIT_STATE_DEFAULT
#define LMX_KEYS_SPLIT_STATE_DEFAULT 2 // initial value of lmx_keys_set_option(LMX_KEYS_OPT_SPLIT_STATE)
#endif
struct KeysDevice {
	// model instances by entity index: ONE 64-byte record per entity (the visible ids are in cell order, entity indices are not:
	// seven separate per-entity arrays meant seven random 128-byte lines per visible entity and made the kernel traffic-bound)
	uint32_t n_entities;
	KeysInstance* inst;       // lod and pose_frame are updated in place
	const LmxMeshMaterial* mesh_materials;
	// the same records in STATIC-SLOT order (the order the cull emits visible ids in), material_offset pointing into mm_s: entities of
	// the sorted set are read with the locality of the set; ids of the dynamic set (slot -1) take the entity-indexed tables above
	KeysInstance* inst_s;
	const LmxMeshMaterial* mm_s;
	// (LMX_KEYS_SPLIT_STATE, experiment) the two fields of a record the kernel WRITES - ModelInstance::lod and Pose::frame - of the sorted
	// set's entities in a dense array of their own, 8 bytes per slot: a 4-byte update of a 64-byte record dirties a whole sector per visible
	// entity (the traffic model of tests/hostsim and the PMC counters agree on ~230 B of traffic per visible entity against 87 algorithmic)
	KeysSlotState* state_s; // nullptr: lod / pose_frame live in inst_s
	KeysSoA soa;            // soa.model != nullptr: the mirror's records are these arrays (inst_s is not allocated), state_s holds lod / pose_frame
	const LmxKeysModel* models;
	const uint32_t *decal_sort_key, *curve_sort_key;
	const uint8_t *decal_layer, *curve_layer;
	// positions: KeysInstance::pos, or the world hierarchy's SoA through slot_of_entity when bound
	const double *wpx, *wpy, *wpz;
	const int32_t* slot_of_entity;
	// outputs
	uint64_t *keys, *values;
	uint32_t cap_pairs;
	uint32_t* rec_key;
	uint64_t* rec_value;
	uint32_t cap_recs;
	uint32_t max_sort_key;
	// auto-instancer groups. The per-key counters are PRIVATISED: copy c (= block index mod n_copies) has its own count / cursor
	// row, so the atomics of a scene with few distinct mesh sort keys (all of them on a handful of cache lines) spread over
	// n_copies times as many lines. A record remembers its copy in bits 24..31 of rec_key.
	uint32_t n_copies;         // power of two, <= LMX_KEYS_MAX_COPIES
	// Two counter tables take turns (round 5: the chain had a fill of its own): a run's key kernels build the histogram in group_count,
	// which arrives ZERO; k_keys_reduce_copies turns it into per-copy bases (group_base), zeroes it - it is the scatter's cursor table
	// from then on - and zeroes the OTHER table (group_count_next: the previous run's cursors), which is the next run's histogram.
	uint32_t *group_count;      // [n_copies][max_sort_key + 1]: per-copy sizes, then the scatter's cursors
	uint32_t *group_count_next; // the same table of the next run
	uint32_t *group_base;       // [n_copies][max_sort_key + 1]: where copy c's records of group k start inside the group
	uint32_t *group_total;      // [max_sort_key + 1]
	uint32_t *group_offset;     // [max_sort_key + 2]
	uint64_t* group_values;
	// block ranks (key ranges that fit k_keys_mesh's LDS histogram; nullptr: the private copies above): every block of k_keys_mesh owns
	// a row [max_sort_key + 1]: per key the records of the blocks that finished before it; a record carries its rank among its block's
	// records of its key (rec_rank) and its row in bits 12..31 of rec_key
	uint32_t* block_rows;
	// ... (key ranges up to 1024; nullptr beyond: k_keys_reduce_rows scans the columns) one counter per key on a cache line of its own: a
	// block adds its row's counts at its end, what the add returns is the row's entry (the records of the key in the blocks before),
	// the counters end up as the groups' sizes. total_pad_next: the next run's, zeroed by this run's last kernel.
	uint32_t *total_pad, *total_pad_next;
	uint32_t cap_rows, n_rows; // rows allocated; rows of this run (= k_keys_mesh's grid: set by launch_keys)
	uint32_t* rec_rank;
	int32_t *poses, *dirty_list;
	uint32_t cap_list;
	uint32_t* counters;       // KEYS_*: arrive zero
	uint32_t* counters_next;  // the next run's (the previous run's results: valid until this run starts), zeroed by k_keys_reduce_copies
};
// The visible ids of ONE renderable type as the cull kernels leave them (lmx_cull_device_shards): `n` shards, shard s holding
// counts[s * cnt_pad] ids at ids + win_base[s] (and their static-set slots at slots + win_base[s]). The key kernels walk the windows
// themselves: gathering them into one list first (k_cull_finalize + k_cull_consolidate) was 14 of the chain's 94 us. win_base == nullptr:
// one contiguous list (n == 1, counts -> its length) - what a type with more than KEYS_MAX_SHARDS shards falls back to.
constexpr int KEYS_PAD_WORDS = 32; // a padded per-key counter: one 128-byte line
constexpr int KEYS_SCATTER_OFFSETS = 1024; // key ranges up to here: k_keys_scatter forms the group offsets itself and (block ranks) the rows' entries come out of returning adds
constexpr int KEYS_MAX_SHARDS = 128; // the layout builder gives a type <= 64 shards for the sorted set + <= 8 for the dynamic one
struct KeysShardList {
	const int32_t* ids;
	const int32_t* slots; // optional
	const uint32_t* counts;
	const uint32_t* win_base;
	uint32_t cnt_pad, n, cap; // cap: an upper bound of the ids (grid size)
};
hipError_t launch_keys_mirror_count(hipStream_t s, const int32_t* slot_ids, uint32_t n_slots, const KeysInstance* inst, uint32_t n_entities, const LmxKeysModel* models, uint32_t* count);
hipError_t launch_keys_mirror_fill(hipStream_t s, const int32_t* slot_ids, uint32_t n_slots, const KeysInstance* inst, uint32_t n_entities, const LmxKeysModel* models,
	const LmxMeshMaterial* mesh_materials, const uint32_t* offset, KeysInstance* inst_s /* or */, const KeysSoA& soa, LmxMeshMaterial* mm_s, KeysSlotState* state_s /* optional */);
hipError_t launch_keys_mirror_positions(hipStream_t s, const int32_t* slot_ids, uint32_t n_slots, const KeysInstance* inst, uint32_t n_entities, KeysInstance* inst_s, const KeysSoA& soa);
hipError_t launch_keys_mirror_sync(hipStream_t s, const int32_t* slot_ids, uint32_t n_slots, const KeysInstance* inst_s, const int32_t* model_s, const KeysSlotState* state_s,
	KeysInstance* inst, uint32_t n_entities);
hipError_t launch_keys_mirror_carry(hipStream_t s, const PatchId* patches, uint32_t n, const int32_t* slot_ids, uint32_t n_slots, const KeysInstance* inst_s, const int32_t* model_s,
	const KeysSlotState* state_s, KeysInstance* inst, uint32_t n_entities);
uint32_t keys_mesh_grid_cap(); // the largest grid of k_keys_mesh = the rows a block-ranks table needs
hipError_t launch_keys(hipStream_t s, const KeysDevice& d, const KeysViewDevice& view, const KeysShardList& meshes, const KeysShardList& decals, const KeysShardList& curves);

// Pose::computeAbsolute + computeSkinMatrices (+ optional dual-quaternion palette), one wave per PoseGroup
hipError_t launch_pose_palette(hipStream_t s, const SkinInstance* inst, const PoseGroup* groups, const uint32_t n_groups[3] /* by capacity 4, 2, 1 */,
	const float* rel_pos, const float4* rel_rot, float* pose_pos, float4* pose_rot, const uint32_t* level_items, const uint16_t* level_off,
	const float* inv_pos, const float4* inv_rot, float4* palette, float4* dual_quats /* optional */);
// Pose::blend over all bones of all instances
hipError_t launch_pose_blend(hipStream_t s, float* pos, float4* rot, const float* rhs_pos, const float4* rhs_rot, size_t n_bones, float weight);
// palette rows (3 x float4 per bone) -> column-major 4 x 4 matrices
hipError_t launch_palette_expand(hipStream_t s, const float4* rows, uint32_t n_bones, float4* out);
// evaluateSkin over every vertex of every instance
hipError_t launch_skin_vertices(hipStream_t s, const SkinInstance* inst, const uint32_t* inst_index /* optional subset */, uint32_t n_inst,
	uint32_t max_verts, const float4* mesh /* 2 x float4 per vertex: weights | (position, 4 x u8 bone indices) */,
	const float4* palette /* 3 rows per bone, or the dual quaternions in LMX_SKIN_DQS */, float* out, int mode);
// the same for runs of consecutive instances that share a mesh and a bone count (vertex records held in registers)
hipError_t launch_skin_shared(hipStream_t s, const SkinInstance* inst, const SkinChunk* chunks, uint32_t n_chunks, const float4* mesh_local,
	const uint8_t* tile_bones, const float4* palette, float* out, int mode);
// k_skin_multi: `per_block` instances per block (1, 2, 4, 8, 16; capped by skin_multi_instances for models above 64 bones). A work item
// carries everything the block needs (ONE dependent load at block start, not chunk -> instance -> palette): `count` consecutive
// instances of one model and one mesh - their palettes start at bone `bone_offset` (n_bones each), their outputs at vertex `out_offset`
// (n_verts each) - and the vertex range [v_begin, v_end) of the mesh whose first record in `mesh` (global bone indices) is rec_offset.
struct SkinMultiChunk { uint32_t bone_offset, n_bones, count, v_begin, v_end, rec_offset, n_verts, out_offset, n_stage /* bones staged in LDS: the mesh's largest bone index + 1 */; };
uint32_t skin_multi_instances(uint32_t per_block, uint32_t n_bones);
uint32_t skin_multi_lds_slots(uint32_t n_stage);
hipError_t launch_skin_multi(hipStream_t s, uint32_t per_block, const SkinMultiChunk* chunks, uint32_t n_chunks, uint32_t lds_slots /* largest skin_multi_lds_slots(n_stage) of the chunks */, const float4* mesh, const float4* palette, float* out, int mode);
constexpr uint32_t SKIN_SHARED_TILE_VERTS = 5120; // k_skin_shared: 1024 lanes x 5 vertex records

} // namespace lmx
