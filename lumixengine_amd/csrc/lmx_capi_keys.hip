// lmx_capi_keys.hip — createSortKeys entry points (include/lumix_mi355.h, "sort keys" section): model / instance / material
// tables, the per-view state, the launch chain of keys_kernels.hip and the read-backs.
#include "lmx_context.h"

#include <cstddef>

#include <hipcub/hipcub.hpp>

using namespace lmx;

namespace {

template <typename T> int upload(LmxContext* ctx, DevBuf<T>& buf, const T* src, size_t n) {
	LMX_HIP(ctx, buf.reserve(std::max<size_t>(n, 1)));
	if (n) LMX_HIP(ctx, hipMemcpyAsync(buf.p, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
	return LMX_OK;
}

} // namespace

namespace lmx {

// (Re)build the slot-ordered mirror for the culling system's current static layout, if it is not current already.
static int keys_build_mirror(LmxContext* ctx) {
	KeysState& ks = ctx->keys;
	CullState& cs = ctx->cull;
	if (ks.mirror_valid && ks.mirror_generation == cs.layout_generation && ks.mirror_slots == cs.n_padded) return LMX_OK;
	if (int rc = keys_before_layout_change(ctx)) return rc; // (a mirror of an older layout cannot exist here - the layout change dropped it - but be safe)
	const uint32_t n_slots = cs.n_padded;
	if (!n_slots || !ks.d_inst.p) return LMX_OK; // nothing sorted / no tables: the entity-indexed path
	if (ks.split_state == 2) { // structure of arrays: 42 bytes per slot instead of the 64-byte records
		LMX_HIP(ctx, ks.d_soa_pos.reserve((size_t)n_slots * 3));
		LMX_HIP(ctx, ks.d_soa_model.reserve(n_slots));
		LMX_HIP(ctx, ks.d_soa_mat.reserve(n_slots));
		LMX_HIP(ctx, ks.d_soa_flags.reserve(n_slots));
	} else {
		LMX_HIP(ctx, ks.d_inst_s.reserve(n_slots));
	}
	LMX_HIP(ctx, ks.d_mm_s.reserve(std::max<size_t>(ks.n_mesh_materials, 1)));
	if (ks.split_state) LMX_HIP(ctx, ks.d_state_s.reserve(n_slots));
	LMX_HIP(ctx, ks.d_mm_count.reserve((size_t)n_slots + 1));
	LMX_HIP(ctx, ks.d_mm_off.reserve((size_t)n_slots + 1));
	LMX_HIP(ctx, launch_keys_mirror_count(ctx->stream, cs.ids.p, n_slots, ks.d_inst.p, ks.n_entities, ks.d_models.p, ks.d_mm_count.p));
	size_t temp = 0;
	LMX_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, temp, ks.d_mm_count.p, ks.d_mm_off.p, (int)(n_slots + 1), ctx->stream));
	LMX_HIP(ctx, ks.d_scan_temp.reserve(temp));
	LMX_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(ks.d_scan_temp.p, temp, ks.d_mm_count.p, ks.d_mm_off.p, (int)(n_slots + 1), ctx->stream));
	LMX_HIP(ctx, launch_keys_mirror_fill(ctx->stream, cs.ids.p, n_slots, ks.d_inst.p, ks.n_entities, ks.d_models.p, ks.d_mesh_materials.p, ks.d_mm_off.p,
		ks.split_state == 2 ? nullptr : ks.d_inst_s.p,
		ks.split_state == 2 ? KeysSoA{ks.d_soa_pos.p, ks.d_soa_pos.p + n_slots, ks.d_soa_pos.p + 2 * (size_t)n_slots, ks.d_soa_model.p, ks.d_soa_mat.p, ks.d_soa_flags.p}
		                    : KeysSoA{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr},
		ks.d_mm_s.p, ks.split_state ? ks.d_state_s.p : nullptr));
	ks.mirror_split = ks.split_state;
	ks.mirror_valid = true;
	ks.mirror_generation = cs.layout_generation;
	ks.mirror_slots = n_slots;
	return LMX_OK;
}

// The static layout is about to change (re-sort, the asynchronous compaction's swap) or the tables are about to be replaced in part:
// lod / Pose::frame of the sorted set's entities go back to the entity-indexed records, the mirror is dropped (rebuilt by the next run).
int keys_before_layout_change(LmxContext* ctx) {
	KeysState& ks = ctx->keys;
	if (!ks.mirror_valid) return LMX_OK;
	ks.mirror_valid = false;
	LMX_HIP(ctx, launch_keys_mirror_sync(ctx->stream, ctx->cull.ids.p, std::min(ks.mirror_slots, ctx->cull.n_padded), ks.d_inst_s.p, ks.soa().model,
		ks.mirror_split ? ks.d_state_s.p : nullptr, ks.d_inst.p, ks.n_entities));
	return LMX_OK;
}

// These id patches are about to turn slots into tombstones (removal, move to the overflow set): the state follows the entity.
int keys_before_tombstones(LmxContext* ctx, const PatchId* d_patches, uint32_t n) {
	KeysState& ks = ctx->keys;
	if (!ks.mirror_valid || !n) return LMX_OK;
	LMX_HIP(ctx, launch_keys_mirror_carry(ctx->stream, d_patches, n, ctx->cull.ids.p, std::min(ks.mirror_slots, ctx->cull.n_padded), ks.d_inst_s.p, ks.soa().model,
		ks.mirror_split ? ks.d_state_s.p : nullptr, ks.d_inst.p, ks.n_entities));
	return LMX_OK;
}

} // namespace lmx

extern "C" {

int lmx_keys_set_models(LmxContext* ctx, const LmxKeysModel* models, uint32_t n_models, const uint8_t* mesh_types, uint32_t n_meshes) {
	LMX_CHECK_CTX(ctx);
	if ((n_models && !models) || (n_meshes && !mesh_types)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null model / mesh-type table");
	KeysState& ks = ctx->keys;
	uint32_t max_span = 1;
	for (uint32_t i = 0; i < n_models; ++i) {
		const LmxKeysModel& m = models[i];
		if ((uint64_t)m.first_mesh + m.mesh_count > n_meshes) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "model %u: meshes [%u, +%u) outside the mesh-type table", i, m.first_mesh, m.mesh_count);
		for (int l = 0; l < 5; ++l) {
			const LmxLodIndices& li = m.lod_indices[l];
			if (li.to < li.from) continue;
			if (li.from < 0 || (uint32_t)li.to >= m.mesh_count) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "model %u: LOD %d = [%d, %d] outside its %u meshes", i, l, li.from, li.to, m.mesh_count);
			max_span = std::max<uint32_t>(max_span, (uint32_t)(li.to - li.from + 1));
		}
	}
	// k_keys_mesh scans a lane's (pairs, records) counts as two 16-bit fields of one word: 64 lanes x 2 x span must stay below 2^16.
	// Checked BEFORE anything is uploaded or assigned: a refused table leaves the previous one in place.
	if (max_span > 511) return fail(ctx, LMX_ERR_CAPACITY, "a LOD range of %u meshes exceeds the 511 the key kernel's packed scan holds", max_span);
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (int rc = upload(ctx, ks.d_models, models, n_models)) return rc;
	if (int rc = upload(ctx, ks.d_mesh_types, mesh_types, n_meshes)) return rc;
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	ks.mirror_valid = false; // (mesh counts may have changed; set_instances follows)
	ks.models.assign(models, models + n_models);
	ks.mesh_types.assign(mesh_types, mesh_types + n_meshes);
	ks.n_meshes = n_meshes;
	ks.max_lod_span = max_span;
	return LMX_OK;
}

int lmx_keys_set_instances(LmxContext* ctx, uint32_t n_entities, const int32_t* model, const uint32_t* material_offset,
	const LmxMeshMaterial* mesh_materials, uint32_t n_mesh_materials, const float* lod, const uint8_t* flags, const uint8_t* dirty,
	const uint32_t* pose_frame) {
	LMX_CHECK_CTX(ctx);
	if (n_entities && (!model || !material_offset || !lod || !flags || !dirty || !pose_frame)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null instance table");
	if (n_mesh_materials && !mesh_materials) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null mesh-material table");
	KeysState& ks = ctx->keys;
	for (uint32_t e = 0; e < n_entities; ++e) {
		if (model[e] < 0) continue;
		if ((size_t)model[e] >= ks.models.size()) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %u: unknown model %d (lmx_keys_set_models first)", e, model[e]);
		if ((uint64_t)material_offset[e] + ks.models[model[e]].mesh_count > n_mesh_materials)
			return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %u: mesh materials [%u, +%u) outside the table", e, material_offset[e], ks.models[model[e]].mesh_count);
		if (!(lod[e] >= 0.f && lod[e] <= 4.f)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %u: ModelInstance::lod %g outside [0, 4]", e, (double)lod[e]);
	}
	for (uint32_t i = 0; i < n_mesh_materials; ++i) // bucket_map / layer_to_bucket have 255 entries (pipeline.cpp:573, :3802)
		if (mesh_materials[i].layer >= 255) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "mesh material %u: layer %u (layers are 0..254)", i, mesh_materials[i].layer);
	// device copy of the material spans with Mesh::type of the model's mesh folded into the padding byte: one gather less
	// on the device's dependent-load chain (entity -> model -> mesh type)
	std::vector<LmxMeshMaterial> dev_mm(mesh_materials, mesh_materials + n_mesh_materials);
	for (LmxMeshMaterial& m : dev_mm) m._pad[0] = m._pad[1] = m._pad[2] = 0;
	for (uint32_t e = 0; e < n_entities; ++e) {
		if (model[e] < 0) continue;
		const LmxKeysModel& m = ks.models[model[e]];
		for (uint32_t k = 0; k < m.mesh_count; ++k) dev_mm[material_offset[e] + k]._pad[0] = ks.mesh_types[m.first_mesh + k];
	}
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (int rc = upload(ctx, ks.d_mesh_materials, dev_mm.data(), n_mesh_materials)) return rc;
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	// one 64-byte record per entity (positions of an earlier lmx_keys_set_positions are kept)
	ks.inst.resize(n_entities);
	for (uint32_t e = 0; e < n_entities; ++e) {
		KeysInstance& r = ks.inst[e];
		r.model = model[e]; r.material_offset = material_offset[e]; r.lod = lod[e]; r.pose_frame = pose_frame[e]; r.flags = flags[e]; r.dirty = dirty[e];
		memset(r.pad, 0, sizeof(r.pad));
	}
	ks.inst_dirty = true;
	ks.mirror_valid = false; // lod / Pose::frame restart from the uploaded values: nothing to hand back
	ks.n_mesh_materials = n_mesh_materials;
	if (n_entities != ks.n_entities) ks.have_decals = ks.have_curves = false; // decal tables of another entity range are dropped
	ks.n_entities = n_entities;
	ks.have_instances = true;
	if (ks.slot_order) ctx->cull.emit_slots = true; // the culls from now on also emit the static-set slot of every visible id
	return LMX_OK;
}

int lmx_keys_set_decals(LmxContext* ctx, uint32_t n_entities, const uint32_t* decal_sort_key, const uint8_t* decal_layer,
	const uint32_t* curve_sort_key, const uint8_t* curve_layer) {
	LMX_CHECK_CTX(ctx);
	if ((decal_sort_key == nullptr) != (decal_layer == nullptr) || (curve_sort_key == nullptr) != (curve_layer == nullptr))
		return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "sort-key and layer tables come in pairs");
	KeysState& ks = ctx->keys;
	if (ks.have_instances && n_entities != ks.n_entities) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "decal tables cover %u entities, instance tables %u", n_entities, ks.n_entities);
	for (uint32_t e = 0; e < n_entities; ++e)
		if ((decal_layer && decal_layer[e] >= 255) || (curve_layer && curve_layer[e] >= 255)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %u: decal layer 255 (layers are 0..254)", e);
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	ks.have_decals = decal_sort_key != nullptr;
	ks.have_curves = curve_sort_key != nullptr;
	if (ks.have_decals) {
		if (int rc = upload(ctx, ks.d_decal_key, decal_sort_key, n_entities)) return rc;
		if (int rc = upload(ctx, ks.d_decal_layer, decal_layer, n_entities)) return rc;
	}
	if (ks.have_curves) {
		if (int rc = upload(ctx, ks.d_curve_key, curve_sort_key, n_entities)) return rc;
		if (int rc = upload(ctx, ks.d_curve_layer, curve_layer, n_entities)) return rc;
	}
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (!ks.have_instances) ks.n_entities = n_entities;
	return LMX_OK;
}

int lmx_keys_set_positions(LmxContext* ctx, const double* xyz, uint32_t n_entities) {
	LMX_CHECK_CTX(ctx);
	if (n_entities && !xyz) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null positions");
	KeysState& ks = ctx->keys;
	if (ks.inst.size() < n_entities) {
		const size_t old_n = ks.inst.size();
		ks.inst.resize(n_entities);
		for (size_t e = old_n; e < n_entities; ++e) { memset(&ks.inst[e], 0, sizeof(KeysInstance)); ks.inst[e].model = -1; }
	}
	for (uint32_t e = 0; e < n_entities; ++e) memcpy(ks.inst[e].pos, xyz + 3 * (size_t)e, sizeof(double) * 3);
	if (!ks.inst_dirty && ks.d_inst.p && ks.d_inst.cap >= ks.inst.size() && ks.inst.size() == ks.inst_uploaded) {
		// the records are on the device already: only the positions are replaced (24 of every 64 bytes), ModelInstance::lod and
		// Pose::frame keep the state the kernels advanced
		LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		LMX_HIP(ctx, hipMemcpy2D(reinterpret_cast<char*>(ks.d_inst.p) + offsetof(KeysInstance, pos), sizeof(KeysInstance), xyz, sizeof(double) * 3, sizeof(double) * 3, n_entities,
			hipMemcpyHostToDevice));
		// ... and the slot-ordered mirror takes them from there (one gather over the slots) instead of being dropped and rebuilt - count,
		// scan, fill, a copy of the material table - by the next run: a host that uploads positions every frame (no lmx_keys_bind_world)
		// paid an O(sorted set) rebuild per frame for a mirror that exists to save time
		if (ks.mirror_valid)
			LMX_HIP(ctx, launch_keys_mirror_positions(ctx->stream, ctx->cull.ids.p, std::min(ks.mirror_slots, ctx->cull.n_padded), ks.d_inst.p, ks.n_entities, ks.d_inst_s.p, ks.soa()));
	} else {
		if (int rc = keys_before_layout_change(ctx)) return rc; // the whole table is uploaded again by the next run: the mirror is rebuilt from it
		ks.inst_dirty = true;
	}
	ks.n_positions = n_entities;
	return LMX_OK;
}

int lmx_keys_bind_world(LmxContext* ctx, int enable) {
	LMX_CHECK_CTX(ctx);
	ctx->keys.use_world = enable != 0;
	return LMX_OK;
}

int lmx_keys_set_option(LmxContext* ctx, int option, int value) {
	LMX_CHECK_CTX(ctx);
	KeysState& ks = ctx->keys;
	if (option == LMX_KEYS_OPT_SPLIT_STATE) {
		if (value < 0 || value > 2) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "mirror form %d not in [0,2]", value);
		if (value != ks.split_state) {
			if (int rc = keys_before_layout_change(ctx)) return rc; // hand the state back, drop the mirror: the next run builds it in the other form
			ks.split_state = value;
		}
		return LMX_OK;
	}
	if (option == LMX_KEYS_OPT_WALK_SHARDS) {
		ks.walk_shards = value != 0;
		return LMX_OK;
	}
	if (option == LMX_KEYS_OPT_BLOCK_RANKS) {
		ks.block_ranks = value != 0;
		return LMX_OK;
	}
	if (option != LMX_KEYS_OPT_SLOT_ORDER) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "unknown sort-key option %d", option);
	if (!value) {
		if (int rc = keys_before_layout_change(ctx)) return rc; // hand the state back, drop the mirror
		ctx->cull.emit_slots = false;
	}
	ks.slot_order = value != 0;
	if (ks.slot_order && ks.have_instances) ctx->cull.emit_slots = true;
	return LMX_OK;
}

int lmx_keys_run(LmxContext* ctx, uint32_t view, uint32_t frustum, const LmxKeysView* kv, uint32_t max_sort_key) {
	LMX_CHECK_CTX(ctx);
	if (!kv) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null view state");
	if (view >= LMX_MAX_VIEWS) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad view");
	if (max_sort_key >= (1u << 24)) return fail(ctx, LMX_ERR_CAPACITY, "max_sort_key %u: mesh sort keys occupy 24 bits (pipeline.cpp:66)", max_sort_key);
	KeysState& ks = ctx->keys;
	CullView& v = ctx->cull.views[view];
	if (!v.valid) return fail(ctx, LMX_ERR_NOT_BUILT, "view %u holds no cull result", view);
	if (frustum >= v.n_frusta) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "frustum %u out of range", frustum);
	if (!ks.have_instances && !ks.have_decals && !ks.have_curves) return fail(ctx, LMX_ERR_NOT_BUILT, "no instance / decal tables uploaded");
	const WorldState& w = ctx->world;
	if (ks.use_world) {
		if (w.slot_of_entity.size() < ks.n_entities) return fail(ctx, LMX_ERR_NOT_BUILT, "world hierarchy covers %zu entities, instance tables %u", w.slot_of_entity.size(), ks.n_entities);
	} else if (ks.have_instances && ks.n_positions < ks.n_entities) {
		return fail(ctx, LMX_ERR_NOT_BUILT, "positions cover %u entities, instance tables %u", ks.n_positions, ks.n_entities);
	}
	// bucket_map, pipeline.cpp:3802-3812
	KeysViewDevice hv;
	memset(&hv, 0, sizeof(hv));
	for (uint32_t i = 0; i < 255; ++i) {
		uint32_t b = kv->layer_to_bucket[i];
		if (b == 0xff) b = 0xffFFffFFu;
		else if (kv->bucket_depth_sorted[b]) b |= 0x100;
		hv.bucket_map[i] = b;
		hv.layer_to_bucket[i] = kv->layer_to_bucket[i];
	}
	hv.is_shadow = kv->is_shadow != 0;
	for (int k = 0; k < 3; ++k) { hv.cam[k] = kv->camera_pos[k]; hv.ref[k] = kv->lod_ref_point[k]; }
	hv.lod_multiplier_rcp = 1 / kv->lod_multiplier; // const float global_lod_multiplier_rcp = 1 / global_lod_multiplier, :3799
	hv.time_delta = kv->time_delta;
	hv.frame_number = kv->frame_number;

	auto seg_cap = [&](int t) { return (t + 1 < MAX_TYPES ? v.out_start[t + 1] : v.out_stride) - v.out_start[t]; };
	const uint32_t mesh_cap = ks.have_instances ? seg_cap(LMX_TYPE_MESH) : 0, decal_cap = ks.have_decals ? seg_cap(LMX_TYPE_DECAL) : 0,
		curve_cap = ks.have_curves ? seg_cap(LMX_TYPE_CURVE_DECAL) : 0;
	// worst case: every visible mesh entity is between two LODs and pushes every mesh of both
	const size_t cap_recs = (size_t)mesh_cap * ks.max_lod_span * 2;
	const size_t cap_pairs = cap_recs + decal_cap + curve_cap + max_sort_key + 1;
	if (cap_pairs > 0xffffffffull) return fail(ctx, LMX_ERR_CAPACITY, "sort-key capacity %zu exceeds 32 bits", cap_pairs);
	LMX_HIP(ctx, ks.d_keys.reserve(std::max<size_t>(cap_pairs, 1)));
	LMX_HIP(ctx, ks.d_values.reserve(std::max<size_t>(cap_pairs, 1)));
	LMX_HIP(ctx, ks.d_rec_key.reserve(std::max<size_t>(cap_recs, 1)));
	LMX_HIP(ctx, ks.d_rec_value.reserve(std::max<size_t>(cap_recs, 1)));
	LMX_HIP(ctx, ks.d_group_values.reserve(std::max<size_t>(cap_recs, 1)));
	// privatised group counters: as many copies as keep the table under 256 k entries (256 for <= 1024 keys, 64 for <= 4096, 1 for > 128 k);
	// a record remembers its copy in the top 8 bits of rec_key
	uint32_t n_copies = LMX_KEYS_MAX_COPIES;
	while (n_copies > 1 && (uint64_t)n_copies * (max_sort_key + 1) > 262144) n_copies >>= 1;
	const size_t g = (size_t)max_sort_key + 1;
	// [table 0 | table 1 | counters 0 | counters 1 | group_base n_copies * g | group_total g | group_offset g + 1]. The two counter tables
	// and the two blocks of list counters take turns from run to run: what a run needs zeroed is zeroed by the run before it
	// (k_keys_reduce_copies), so the chain has no fill of its own - it was one launch of ~5 us in a chain of eight. Only a new layout
	// (key range, copy count, a new allocation) or a failed run starts from a fill.
	const size_t gc = (n_copies * g + 31) & ~(size_t)31; // a table, padded so that the list counters start on a 128-byte line (64-bit atomics)
	LMX_HIP(ctx, ks.d_groups.reserve(2 * gc + 2 * KEYS_COUNTERS + n_copies * g + g + g + 1));
	if (!ks.groups_clean || ks.groups_at != ks.d_groups.p || ks.groups_copies != n_copies || ks.groups_keys != g) {
		LMX_HIP(ctx, hipMemsetAsync(ks.d_groups.p, 0, (2 * gc + 2 * KEYS_COUNTERS) * sizeof(uint32_t), ctx->stream));
		ks.groups_at = ks.d_groups.p; ks.groups_copies = n_copies; ks.groups_keys = g;
		ks.run_parity = 0;
		ks.table_parity = 0;
	}
	const bool counters_were_clean = ks.groups_clean;
	ks.groups_clean = false; // until this run's launches are enqueued
	// block ranks: a row of per-key counts for every block of k_keys_mesh (a fixed-size grid), for key ranges that fit the key kernel's
	// LDS histogram; otherwise the private copies above do the counting
	const size_t rows_cap = keys_mesh_grid_cap();
	const bool block_ranks = ks.block_ranks && ks.have_instances && mesh_cap != 0 && max_sort_key < 4096;
	const bool row_adds = block_ranks && max_sort_key < (uint32_t)KEYS_SCATTER_OFFSETS;
	if (row_adds) { // one counter per key, a cache line each, two sets taking turns (a run's last kernel zeroes the next run's)
		const uint32_t* before = ks.d_total_pad.p;
		LMX_HIP(ctx, ks.d_total_pad.reserve(2 * g * KEYS_PAD_WORDS));
		// (a failed or aborted run may have left its adds in the per-key counters without flipping the parity: they start over with the group tables)
		if (before != ks.d_total_pad.p || ks.pad_keys != g || !counters_were_clean) {
			LMX_HIP(ctx, hipMemsetAsync(ks.d_total_pad.p, 0, 2 * g * KEYS_PAD_WORDS * sizeof(uint32_t), ctx->stream));
			ks.pad_keys = g;
			ks.pad_parity = 0;
		}
	}
	if (block_ranks) {
		LMX_HIP(ctx, ks.d_block_rows.reserve(rows_cap * g));
		LMX_HIP(ctx, ks.d_rec_rank.reserve(std::max<size_t>(cap_recs, 1)));
	}
	LMX_HIP(ctx, ks.d_poses.reserve(std::max<size_t>(mesh_cap, 1)));
	LMX_HIP(ctx, ks.d_dirty_list.reserve(std::max<size_t>(mesh_cap, 1)));

	if (ks.inst_dirty) { // the host mirror changed (tables or positions): lod / Pose::frame restart from the uploaded values
		LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		if (int rc = upload(ctx, ks.d_inst, ks.inst.data(), ks.inst.size())) return rc;
		LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		ks.inst_dirty = false;
		ks.inst_uploaded = ks.inst.size();
	}
	// slot order: from now on the culls also emit the static-set slot of every visible id; a view culled before that (or with the
	// option off) is walked through the entity-indexed tables
	CullState& cs = ctx->cull;
	if (ks.slot_order && ks.have_instances) {
		cs.emit_slots = true;
		if (v.has_slots) {
			if (int rc = keys_build_mirror(ctx)) return rc;
		} else if (ks.mirror_valid) {
			// this view was culled before the culls emitted slots (or kept across a flush): it is walked through the entity-indexed tables,
			// but ModelInstance::lod / Pose::frame of the sorted set's entities live in the mirror - hand them back first, or this run
			// would read stale lod values and stamp a second copy of Pose::frame (a pose pushed twice in one frame)
			if (int rc = keys_before_layout_change(ctx)) return rc;
		}
	} else if (ks.mirror_valid) {
		if (int rc = keys_before_layout_change(ctx)) return rc;
	}
	ProfScope ps(ctx, LMX_K_SORT_KEYS);
	const uint32_t parity = ks.run_parity;
	ks.counters_at = 2 * gc + parity * KEYS_COUNTERS;
	KeysDevice d;
	memset(&d, 0, sizeof(d));
	d.n_entities = ks.n_entities;
	if (ks.have_instances) {
		d.inst = ks.d_inst.p;
		d.mesh_materials = ks.d_mesh_materials.p;
		d.models = ks.d_models.p;
		d.inst_s = ks.mirror_valid ? ks.d_inst_s.p : nullptr;
		d.mm_s = ks.mirror_valid ? ks.d_mm_s.p : nullptr;
		d.state_s = ks.mirror_valid && ks.mirror_split ? ks.d_state_s.p : nullptr;
		if (ks.mirror_valid) d.soa = ks.soa();
	}
	if (ks.have_decals) { d.decal_sort_key = ks.d_decal_key.p; d.decal_layer = ks.d_decal_layer.p; }
	if (ks.have_curves) { d.curve_sort_key = ks.d_curve_key.p; d.curve_layer = ks.d_curve_layer.p; }
	if (ks.use_world) {
		d.wpx = ctx->world.pos[3].p; d.wpy = ctx->world.pos[4].p; d.wpz = ctx->world.pos[5].p;
		d.slot_of_entity = ctx->world.d_slot_of_entity.p;
	}
	d.keys = ks.d_keys.p; d.values = ks.d_values.p; d.cap_pairs = (uint32_t)cap_pairs;
	d.rec_key = ks.d_rec_key.p; d.rec_value = ks.d_rec_value.p; d.cap_recs = (uint32_t)cap_recs;
	d.max_sort_key = max_sort_key;
	d.n_copies = n_copies;
	// (the counter TABLES only change hands on runs that use them: a block-ranks run in between leaves both as they are - with one
	// parity for tables and list counters such a run handed the next one the table that still held the previous scatter's cursors)
	const uint32_t table = ks.table_parity;
	d.group_count = ks.d_groups.p + table * gc; d.group_count_next = ks.d_groups.p + (table ^ 1u) * gc;
	d.group_base = ks.d_groups.p + 2 * gc + 2 * KEYS_COUNTERS; d.group_total = d.group_base + n_copies * g; d.group_offset = d.group_total + g;
	ks.offsets_at = 2 * gc + 2 * KEYS_COUNTERS + n_copies * g + g;
	d.group_values = ks.d_group_values.p;
	d.block_rows = block_ranks ? ks.d_block_rows.p : nullptr;
	d.cap_rows = (uint32_t)rows_cap;
	d.rec_rank = block_ranks ? ks.d_rec_rank.p : nullptr;
	d.total_pad = row_adds ? ks.d_total_pad.p + (size_t)ks.pad_parity * g * KEYS_PAD_WORDS : nullptr;
	d.total_pad_next = row_adds ? ks.d_total_pad.p + (size_t)(ks.pad_parity ^ 1u) * g * KEYS_PAD_WORDS : nullptr;
	d.poses = ks.d_poses.p; d.dirty_list = ks.d_dirty_list.p; d.cap_list = mesh_cap;
	d.counters = ks.d_groups.p + ks.counters_at;
	d.counters_next = ks.d_groups.p + 2 * gc + (parity ^ 1u) * KEYS_COUNTERS;
	// The visible ids, per type, where the cull left them: one window per output shard (shards of a type are contiguous in shard order).
	// Gathering them into one list per type first (k_cull_finalize + k_cull_consolidate) was 14 of the chain's 94 us.
	const bool use_slots = ks.mirror_valid && v.has_slots && ks.slot_order;
	uint32_t first[MAX_TYPES], count[MAX_TYPES];
	bool walk = ks.walk_shards;
	for (int t = 0; t < MAX_TYPES; ++t) { first[t] = 0; count[t] = 0; }
	for (uint32_t sidx = 0; sidx < cs.n_shards; ++sidx) {
		const uint8_t t = cs.shard_type[sidx];
		if (count[t]++ == 0) first[t] = sidx;
		else if (first[t] + count[t] - 1 != sidx) walk = false; // (never: the layout builder keeps a type's shards together)
	}
	for (int t : {LMX_TYPE_MESH, LMX_TYPE_DECAL, LMX_TYPE_CURVE_DECAL}) if (count[t] > (uint32_t)KEYS_MAX_SHARDS) walk = false;
	KeysShardList lists[3];
	const int list_type[3] = {LMX_TYPE_MESH, LMX_TYPE_DECAL, LMX_TYPE_CURVE_DECAL};
	const uint32_t list_cap[3] = {mesh_cap, decal_cap, curve_cap};
	if (walk) {
		const uint32_t* counts = v.counts_ptr() + (size_t)frustum * cs.n_shards * cs.cnt_pad;
		for (int k = 0; k < 3; ++k) {
			const int t = list_type[k];
			lists[k].ids = v.out.p + (size_t)frustum * v.out_stride;
			lists[k].slots = k == 0 && use_slots ? v.out_slots.p + (size_t)frustum * v.out_stride : nullptr;
			lists[k].counts = counts + (size_t)first[t] * cs.cnt_pad;
			lists[k].win_base = cs.d_win_base.p + first[t];
			lists[k].cnt_pad = cs.cnt_pad;
			lists[k].n = count[t];
			lists[k].cap = list_cap[k];
		}
	} else { // LMX_KEYS_OPT_WALK_SHARDS 0: one contiguous list per type, gathered first
		if (int rc = cull_view_consolidate(ctx, v)) return rc;
		const int32_t* row = v.cons_ptr() + (size_t)frustum * v.out_stride;
		const uint32_t* counts = v.totals_ptr() + (size_t)frustum * MAX_TYPES;
		for (int k = 0; k < 3; ++k) {
			const int t = list_type[k];
			lists[k].ids = row + v.out_start[t];
			lists[k].slots = k == 0 && use_slots ? v.cons_slots.p + (size_t)frustum * v.out_stride + v.out_start[t] : nullptr;
			lists[k].counts = counts + t;
			lists[k].win_base = nullptr;
			lists[k].cnt_pad = 0;
			lists[k].n = 1;
			lists[k].cap = list_cap[k];
		}
	}
	LMX_HIP(ctx, launch_keys(ctx->stream, d, hv, lists[0], lists[1], lists[2]));
	ks.groups_clean = true;
	ks.run_parity = parity ^ 1u;
	if (!block_ranks) ks.table_parity = table ^ 1u;
	if (row_adds) ks.pad_parity ^= 1u;
	ks.max_sort_key = max_sort_key;
	ks.ran = true;
	ks.sorted = false;
	return LMX_OK;
}

static int keys_host_counters(LmxContext* ctx, uint32_t* c) {
	KeysState& ks = ctx->keys;
	if (!ks.ran) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_keys_run has not run");
	LMX_HIP(ctx, hipMemcpyAsync(c, ks.d_groups.p + ks.counters_at, KEYS_COUNTERS * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

int lmx_keys_counts(LmxContext* ctx, LmxKeysCounts* out) {
	LMX_CHECK_CTX(ctx);
	if (!out) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null out");
	uint32_t c[KEYS_COUNTERS];
	if (int rc = keys_host_counters(ctx, c)) return rc;
	out->pairs = c[KEYS_N_PAIRS]; out->instanced = c[KEYS_N_RECS]; out->groups = c[KEYS_N_GROUPS]; out->poses = c[KEYS_N_POSES];
	out->dirty = c[KEYS_N_DIRTY]; out->overflow = c[KEYS_OVERFLOW];
	return LMX_OK;
}

int lmx_keys_sort(LmxContext* ctx) {
	LMX_CHECK_CTX(ctx);
	KeysState& ks = ctx->keys;
	uint32_t c[KEYS_COUNTERS];
	if (int rc = keys_host_counters(ctx, c)) return rc;
	if (c[KEYS_OVERFLOW]) return fail(ctx, LMX_ERR_CAPACITY, "sort-key output overflowed (code %u)", c[KEYS_OVERFLOW]);
	const uint32_t n = c[KEYS_N_PAIRS];
	if (n > 1) {
		LMX_HIP(ctx, ks.d_keys_alt.reserve(n));
		LMX_HIP(ctx, ks.d_values_alt.reserve(n));
		size_t temp = 0;
		LMX_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, temp, ks.d_keys.p, ks.d_keys_alt.p, ks.d_values.p, ks.d_values_alt.p, (int)n, 0, 64, ctx->stream));
		LMX_HIP(ctx, ks.d_sort_temp.reserve(temp));
		ProfScope ps(ctx, LMX_K_SORT_KEYS);
		LMX_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(ks.d_sort_temp.p, temp, ks.d_keys.p, ks.d_keys_alt.p, ks.d_values.p, ks.d_values_alt.p, (int)n, 0, 64, ctx->stream));
		LMX_HIP(ctx, hipMemcpyAsync(ks.d_keys.p, ks.d_keys_alt.p, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
		LMX_HIP(ctx, hipMemcpyAsync(ks.d_values.p, ks.d_values_alt.p, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
	}
	ks.sorted = true;
	return LMX_OK;
}

int lmx_keys_read_pairs(LmxContext* ctx, uint64_t* keys, uint64_t* values, uint32_t cap) {
	LMX_CHECK_CTX(ctx);
	KeysState& ks = ctx->keys;
	uint32_t c[KEYS_COUNTERS];
	if (int rc = keys_host_counters(ctx, c)) return rc;
	if (c[KEYS_OVERFLOW]) return fail(ctx, LMX_ERR_CAPACITY, "sort-key output overflowed (code %u)", c[KEYS_OVERFLOW]);
	const uint32_t n = c[KEYS_N_PAIRS];
	if (cap < n) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u pairs", n);
	if (n && keys) LMX_HIP(ctx, hipMemcpyAsync(keys, ks.d_keys.p, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
	if (n && values) LMX_HIP(ctx, hipMemcpyAsync(values, ks.d_values.p, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

int lmx_keys_read_instancer(LmxContext* ctx, uint32_t* offsets, uint64_t* values, uint32_t cap_values) {
	LMX_CHECK_CTX(ctx);
	KeysState& ks = ctx->keys;
	uint32_t c[KEYS_COUNTERS];
	if (int rc = keys_host_counters(ctx, c)) return rc;
	if (c[KEYS_OVERFLOW]) return fail(ctx, LMX_ERR_CAPACITY, "sort-key output overflowed (code %u)", c[KEYS_OVERFLOW]);
	const uint32_t n = c[KEYS_N_RECS];
	if (values && cap_values < n) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u instanced renderables", n);
	if (offsets) LMX_HIP(ctx, hipMemcpyAsync(offsets, ks.d_groups.p + ks.offsets_at, (size_t)(ks.max_sort_key + 2) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
	if (values && n) LMX_HIP(ctx, hipMemcpyAsync(values, ks.d_group_values.p, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

static int keys_read_list(LmxContext* ctx, int which, const int32_t* src, int32_t* entities, uint32_t cap) {
	uint32_t c[KEYS_COUNTERS];
	if (int rc = keys_host_counters(ctx, c)) return rc;
	if (c[KEYS_OVERFLOW]) return fail(ctx, LMX_ERR_CAPACITY, "sort-key output overflowed (code %u)", c[KEYS_OVERFLOW]);
	const uint32_t n = c[which];
	if (cap < n) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u entities", n);
	if (n && entities) LMX_HIP(ctx, hipMemcpyAsync(entities, src, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

int lmx_keys_read_poses(LmxContext* ctx, int32_t* entities, uint32_t cap) {
	LMX_CHECK_CTX(ctx);
	return keys_read_list(ctx, KEYS_N_POSES, ctx->keys.d_poses.p, entities, cap);
}

int lmx_keys_read_dirty(LmxContext* ctx, int32_t* entities, uint32_t cap) {
	LMX_CHECK_CTX(ctx);
	return keys_read_list(ctx, KEYS_N_DIRTY, ctx->keys.d_dirty_list.p, entities, cap);
}

int lmx_keys_read_state(LmxContext* ctx, float* lod, uint32_t* pose_frame, uint32_t n_entities) {
	LMX_CHECK_CTX(ctx);
	KeysState& ks = ctx->keys;
	if (!ks.have_instances) return fail(ctx, LMX_ERR_NOT_BUILT, "no instance tables uploaded");
	if (n_entities != ks.n_entities) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "expected %u entities", ks.n_entities);
	if (ks.inst_dirty) return fail(ctx, LMX_ERR_NOT_BUILT, "tables changed since the last lmx_keys_run");
	if (ks.mirror_valid) // entities of the sorted set keep lod / Pose::frame in their slot records
		LMX_HIP(ctx, launch_keys_mirror_sync(ctx->stream, ctx->cull.ids.p, ks.mirror_slots, ks.d_inst_s.p, ks.soa().model, ks.mirror_split ? ks.d_state_s.p : nullptr, ks.d_inst.p,
			ks.n_entities));
	const char* base = reinterpret_cast<const char*>(ks.d_inst.p);
	if (lod && n_entities)
		LMX_HIP(ctx, hipMemcpy2DAsync(lod, sizeof(float), base + offsetof(KeysInstance, lod), sizeof(KeysInstance), sizeof(float), n_entities, hipMemcpyDeviceToHost, ctx->stream));
	if (pose_frame && n_entities)
		LMX_HIP(ctx, hipMemcpy2DAsync(pose_frame, sizeof(uint32_t), base + offsetof(KeysInstance, pose_frame), sizeof(KeysInstance), sizeof(uint32_t), n_entities, hipMemcpyDeviceToHost,
			ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

int lmx_keys_device_pairs(LmxContext* ctx, const uint64_t** d_keys, const uint64_t** d_values, const uint32_t** d_count) {
	LMX_CHECK_CTX(ctx);
	KeysState& ks = ctx->keys;
	if (!ks.ran) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_keys_run has not run");
	if (d_keys) *d_keys = ks.d_keys.p;
	if (d_values) *d_values = ks.d_values.p;
	if (d_count) *d_count = ks.d_groups.p + ks.counters_at + KEYS_N_PAIRS;
	return LMX_OK;
}

} // extern "C"
