// lmx_kernels.h — host-callable launchers of the gfx950 kernels (defined in *_kernels.hip) + shared device layouts.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lmx_math.h"
#include "lmx_types.h"

namespace lmx {

constexpr int MAX_FRUSTA = 8;
constexpr int MAX_TYPES = 8;
constexpr uint32_t CHUNK = 64;          // spheres per chunk = one wavefront
constexpr uint32_t TILE_ALIGN = 4096;   // type ranges are padded to this many spheres (largest tile of any variant)
constexpr uint32_t CELL_DEAD = 0x80000000u;

// One (cell index, type, is_big) group. meta = type | is_big << 8 | CELL_DEAD.
struct CellKey { int32_t ix, iy, iz; uint32_t meta; };

// Per-type ranges in the padded sphere index space.
struct TypeTable {
	uint32_t ent_start[MAX_TYPES]; // first slot of the type (static set: multiple of TILE_ALIGN, dynamic set: of 2048)
	uint32_t ent_end[MAX_TYPES];   // end of the padded range
	uint32_t out_start[MAX_TYPES]; // where the type's visible ids start in a frustum's output row (static + dynamic share it)
};

struct FrustaArg { DevFrustum f[MAX_FRUSTA]; };

struct CullDeviceView {
	const float4* spheres;       // [n_padded] {rel.x, rel.y, rel.z, radius}, cell-relative fp32 (culling_system.cpp:100)
	const int32_t* ids;          // [n_padded] entity index, -1 for padding
	const uint32_t* chunk_cell;  // [n_padded / 64] cell slot of the first sphere of the chunk
	const uint64_t* chunk_flags; // [n_padded / 64] bit l: sphere l of the chunk starts a new cell
	const CellKey* cells;        // [n_cells]
	uint32_t n_padded;
	uint32_t n_cells;
	// fused kernel: per tile-size variant k (tile = 4096 >> k), tile-major cell keys + {first cell, n cells} per tile
	const CellKey* tile_cells[3];
	const uint32_t* tile_tab[3];
	const TileBox* tile_box[3];
	uint32_t tile_cap[3];
};

// classify cells [cell_begin, cell_begin + n) for n_frusta frusta -> cellinfo[f * cell_stride + c] =
// {offset.x, offset.y, offset.z, bits(class)}; also zeroes counts[0 .. MAX_FRUSTA * MAX_TYPES).
hipError_t launch_cull_classify(hipStream_t s, const CullDeviceView& v, uint32_t cell_begin, uint32_t n, const FrustaArg& fr,
	int n_frusta, float4* cellinfo, uint32_t cell_stride, uint32_t* counts);

// test spheres [ent_begin, ent_end) (multiples of TILE_ALIGN, one type per tile) and compact visible ids into
// out_ids[f * out_stride + tt.ent_start[type] + ...], counts[f * MAX_TYPES + type].
hipError_t launch_cull_spheres(hipStream_t s, const CullDeviceView& v, uint32_t ent_begin, uint32_t ent_end, const TypeTable& tt,
	const FrustaArg& fr, int n_frusta, const float4* cellinfo, uint32_t cell_stride, int32_t* out_ids, uint32_t out_stride,
	uint32_t* counts);

// Fused single-launch variant (per-tile classification in LDS) for the tile size cull_tile_size(n_frusta); the layout
// bounds the cells per tile, so fused_lds_bytes(...) always fits the 64 KiB dynamic-LDS limit (the caller still checks). counts must be zero on entry; counts_next (may be null) is
// cleared for the following cull.
uint32_t cull_tile_size(int n_frusta);
size_t fused_lds_bytes(int n_frusta, uint32_t tile, uint32_t cell_cap);
hipError_t launch_cull_fused(hipStream_t s, const CullDeviceView& v, uint32_t ent_begin, uint32_t ent_end, const TypeTable& tt,
	const FrustaArg& fr, int n_frusta, int32_t* out_ids, uint32_t out_stride, uint32_t* counts, uint32_t* counts_next, bool small_tiles);

// Dynamic set: entities whose transform changes every frame (bound to the world hierarchy) are kept UNSORTED as world
// position (fp64) + radius + id. Their cell, cell-relative position and per-cell class are recomputed per entity per cull
// with exactly the arithmetic CullingSystem::set + cullInternal would apply (cell_of, classify_cell, sphere_visible), so
// no re-binning is ever needed. Visible ids are appended to the same per-type output segments / counters as the static set.
struct DynDeviceView {
	const double* px; const double* py; const double* pz;
	const float* radius;
	const int32_t* ids; // -1 = padding
	uint32_t n_padded;
};
hipError_t launch_cull_dynamic(hipStream_t s, const DynDeviceView& d, uint32_t slot_begin, uint32_t slot_end, const TypeTable& dyn_tt,
	const FrustaArg& fr, int n_frusta, int32_t* out_ids, uint32_t out_stride, uint32_t* counts);

// spheres[slot[i]] = value[i]
hipError_t launch_patch_spheres(hipStream_t s, float4* spheres, const uint32_t* slot, const float4* value, uint32_t n);

// ---- world transforms ------------------------------------------------------------------------------------
struct WorldDevice {
	// slot order = (level, parent slot); all arrays [n]
	double* lpx; double* lpy; double* lpz; float4* lrot; float* lsx; float* lsy; float* lsz; // local (roots: unused)
	double* wpx; double* wpy; double* wpz; float4* wrot; float* wsx; float* wsy; float* wsz; // world
	const int32_t* parent_slot; // -1 for roots
};
// world[s] = compose(world[parent_slot[s]], local[s]) for s in [first, first + n)
hipError_t launch_xform_level(hipStream_t s, const WorldDevice& w, uint32_t first, uint32_t n);
// out[entity_of_slot[s]] = AoS Transform (56 B) for s in [0, n)
hipError_t launch_xform_export(hipStream_t s, const WorldDevice& w, const int32_t* entity_of_slot, uint32_t n, void* out_transforms);
// stage transforms (AoS LmxTransform, device memory) into the SoA arrays: roots -> world, children -> local
// (force_world: every entity's value goes to the world arrays)
hipError_t launch_xform_scatter(hipStream_t s, const WorldDevice& w, const int32_t* slot_of_entity, const int32_t* entity,
	const void* transforms, uint32_t n, bool force_world = false);
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
struct SkinChunk { uint32_t first_inst, count, v_begin, v_end; };
struct PoseGroup { uint32_t first_inst; uint32_t count; }; // consecutive instances of one model, count <= 16 / 8 / 4 by bone count
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

// ---- createSortKeys (keys_kernels.hip) ----
enum { KEYS_N_PAIRS = 0, KEYS_N_RECS = 1, KEYS_N_POSES = 2, KEYS_N_DIRTY = 3, KEYS_OVERFLOW = 4, KEYS_N_GROUPS = 5, KEYS_COUNTERS = 8 };
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
struct KeysDevice {
	// model instances by entity index: ONE 64-byte record per entity (the visible ids are in cell order, entity indices are not:
	// seven separate per-entity arrays meant seven random 128-byte lines per visible entity and made the kernel traffic-bound)
	uint32_t n_entities;
	KeysInstance* inst;       // lod and pose_frame are updated in place
	const LmxMeshMaterial* mesh_materials;
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
	// n_copies times as many lines. A record remembers its copy in bits 24..29 of rec_key.
	uint32_t n_copies;         // power of two, <= 64
	uint32_t *group_count;     // [n_copies][max_sort_key + 1]: per-copy sizes, turned into per-copy bases by k_keys_offsets
	uint32_t *group_cursor;    // [n_copies][max_sort_key + 1]
	uint32_t *group_total;     // [max_sort_key + 1]
	uint32_t *group_offset;    // [max_sort_key + 2]
	uint64_t* group_values;
	int32_t *poses, *dirty_list;
	uint32_t cap_list;
	uint32_t* counters;       // KEYS_*
};
hipError_t launch_keys(hipStream_t s, const KeysDevice& d, const KeysViewDevice& view, const int32_t* mesh_ids, const uint32_t* mesh_count,
	uint32_t mesh_cap, const int32_t* decal_ids, const uint32_t* decal_count, uint32_t decal_cap, const int32_t* curve_ids,
	const uint32_t* curve_count, uint32_t curve_cap);

// Pose::computeAbsolute + computeSkinMatrices (+ optional dual-quaternion palette), one wave per PoseGroup
hipError_t launch_pose_palette(hipStream_t s, const SkinInstance* inst, const PoseGroup* groups, const uint32_t n_groups[3] /* by capacity 16, 8, 4 */,
	const float* rel_pos, const float4* rel_rot, float* pose_pos, float4* pose_rot, const uint32_t* level_items, const uint16_t* level_off,
	const float* inv_pos, const float4* inv_rot, float4* palette, float4* dual_quats /* optional */);
// Pose::blend over all bones of all instances
hipError_t launch_pose_blend(hipStream_t s, float* pos, float4* rot, const float* rhs_pos, const float4* rhs_rot, size_t n_bones, float weight);
// palette rows (3 x float4 per bone) -> column-major 4 x 4 matrices
hipError_t launch_palette_expand(hipStream_t s, const float4* rows, uint32_t n_bones, float4* out);
// evaluateSkin over every vertex of every instance
hipError_t launch_skin_vertices(hipStream_t s, const SkinInstance* inst, const uint32_t* inst_index /* optional subset */, uint32_t n_inst,
	uint32_t max_verts, const float* verts, const float4* weights, const int16_t* indices, const float4* palette /* 3 rows per bone, or the dual quaternions in LMX_SKIN_DQS */, float* out, int mode);
// the same for runs of consecutive instances that share a mesh and a bone count (vertex records held in registers)
hipError_t launch_skin_shared(hipStream_t s, const SkinInstance* inst, const SkinChunk* chunks, uint32_t n_chunks, const float* verts,
	const float4* weights, const int16_t* indices, const float4* palette /* 3 rows per bone, or the dual quaternions in LMX_SKIN_DQS */, float* out, int mode);
constexpr uint32_t SKIN_SHARED_TILE_VERTS = 5120; // k_skin_shared: 1024 lanes x 5 vertex records

} // namespace lmx
