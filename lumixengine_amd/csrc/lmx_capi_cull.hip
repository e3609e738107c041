// lmx_capi_cull.hip — CullingSystem behind the C ABI (include/lumix_mi355.h, "culling" section).
//
// Two resident sets, one visibility function:
//   * static set   entities nobody moves every frame: host mirror of (cell, cell-relative sphere) per entity, device
//                  layout sorted by (type, is_big, cell) with chunk headers and tile-major cell keys (lmx_cull_layout.h),
//                  culled by the fused single-launch kernel (cull_kernels.hip). Structural changes (add / remove / a
//                  set* that leaves the cell) rebuild the layout at the next cull; in-cell changes patch 16 B.
//   * dynamic set  entities bound to the world hierarchy (lmx_world_bind_culling): unsorted world position (fp64) +
//                  radius, refreshed on the device by lmx_world_propagate and culled by k_cull_dynamic, which re-derives
//                  cell, cell-relative position and per-cell class per entity — what CullingSystem::set + cullInternal
//                  would compute (culling_system.cpp:225-242, 321-369) — so moving entities are never re-binned.
// Both kernels append to the same per-type output segments and counters.
#include "lmx_context.h"

using namespace lmx;

namespace {

static_assert(sizeof(LayoutSphere) == sizeof(float4) && sizeof(LayoutCell) == sizeof(CellKey), "layout PODs mirror the device types");
static_assert(LAYOUT_MAX_TYPES == MAX_TYPES && LAYOUT_CHUNK == CHUNK && LAYOUT_TILE_ALIGN == TILE_ALIGN && LAYOUT_CELL_DEAD == CELL_DEAD, "layout constants");

constexpr uint32_t DYN_ALIGN = 2048; // largest k_cull_dynamic tile: a tile never straddles two types

enum class Where { NONE, STATIC, DYNAMIC };

Where locate(const CullState& cs, int32_t entity, uint32_t* index) {
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

void mark_patch(CullState& cs, uint32_t rec) {
	if (cs.structure_dirty || !cs.built) return;
	const CullRec& r = cs.recs[rec];
	cs.patch_slot.push_back(cs.rec_slot[rec]);
	cs.patch_val.push_back(make_float4(r.rel.x, r.rel.y, r.rel.z, r.radius));
}

int apply_patches(LmxContext* ctx) {
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

// Rebuild the static device layout from the host mirror (lmx_cull_layout.h) and upload it.
int rebuild_static(LmxContext* ctx) {
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
	cs.n_dead_cells = lay.n_dead_cells;
	for (int k = 0; k < 3; ++k) cs.max_tile_cells[k] = lay.max_tile_cells[k];
	for (int a = 0; a < 3; ++a) { cs.scene_lo[a] = INFINITY; cs.scene_hi[a] = -INFINITY; }
	for (const TileBox& b : lay.tile_box[0]) { // world-space box of the occupied cells (static set)
		if (b.flags & TILE_EMPTY) continue;
		for (int a = 0; a < 3; ++a) {
			cs.scene_lo[a] = std::min(cs.scene_lo[a], (double)CELL_SIZE * b.lo[a]);
			cs.scene_hi[a] = std::max(cs.scene_hi[a], (double)CELL_SIZE * b.hi[a] + (double)CELL_SIZE);
		}
	}
	LMX_HIP(ctx, cs.spheres.reserve(std::max<size_t>(n_padded, 1)));
	LMX_HIP(ctx, cs.ids.reserve(std::max<size_t>(n_padded, 1)));
	LMX_HIP(ctx, cs.chunk_cell.reserve(std::max<size_t>(n_chunks, 1)));
	LMX_HIP(ctx, cs.chunk_flags.reserve(std::max<size_t>(n_chunks, 1)));
	LMX_HIP(ctx, cs.cells.reserve(std::max<size_t>(cs.n_cells, 1)));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream)); // synchronous copies from pageable memory: the layout dies with this function
	if (n_padded) {
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
		LMX_HIP(ctx, cs.tile_box[k].reserve(std::max<size_t>(lay.tile_box[k].size(), 1)));
		if (!lay.tile_cells[k].empty()) {
			LMX_HIP(ctx, hipMemcpy(cs.tile_cells[k].p, lay.tile_cells[k].data(), lay.tile_cells[k].size() * sizeof(CellKey), hipMemcpyHostToDevice));
			LMX_HIP(ctx, hipMemcpy(cs.tile_tab[k].p, lay.tile_tab[k].data(), lay.tile_tab[k].size() * sizeof(uint32_t), hipMemcpyHostToDevice));
			LMX_HIP(ctx, hipMemcpy(cs.tile_box[k].p, lay.tile_box[k].data(), lay.tile_box[k].size() * sizeof(TileBox), hipMemcpyHostToDevice));
		}
	}
	cs.rec_slot.swap(lay.rec_slot);
	cs.structure_dirty = false;
	cs.built = true;
	cs.patch_slot.clear();
	cs.patch_val.clear();
	return LMX_OK;
}

// (Re)assign device slots of the dynamic set (grouped by type, each type padded to DYN_ALIGN) and upload everything.
int rebuild_dynamic(LmxContext* ctx) {
	CullState& cs = ctx->cull;
	const size_t n = cs.dyn.size();
	size_t count_by_type[MAX_TYPES] = {};
	for (const DynRec& r : cs.dyn) count_by_type[r.type]++;
	size_t padded = 0;
	for (int t = 0; t < MAX_TYPES; ++t) {
		cs.dyn_tt.ent_start[t] = (uint32_t)padded;
		padded += (count_by_type[t] + DYN_ALIGN - 1) / DYN_ALIGN * DYN_ALIGN;
		cs.dyn_tt.ent_end[t] = (uint32_t)padded;
	}
	if (padded > 0x7fffffffull) return fail(ctx, LMX_ERR_CAPACITY, "too many dynamic spheres (%zu)", n);
	std::vector<double> px(padded, 0.0), py(padded, 0.0), pz(padded, 0.0);
	std::vector<float> radius(padded, 0.f);
	std::vector<int32_t> ids(padded, -1);
	cs.dyn_slot.assign(n, 0);
	size_t cursor[MAX_TYPES];
	for (int t = 0; t < MAX_TYPES; ++t) cursor[t] = cs.dyn_tt.ent_start[t];
	for (size_t i = 0; i < n; ++i) {
		const DynRec& r = cs.dyn[i];
		const size_t s = cursor[r.type]++;
		cs.dyn_slot[i] = (uint32_t)s;
		px[s] = r.pos[0];
		py[s] = r.pos[1];
		pz[s] = r.pos[2];
		radius[s] = r.radius;
		ids[s] = r.entity;
	}
	cs.dyn_padded = (uint32_t)padded;
	const size_t cap = std::max<size_t>(padded, 1);
	LMX_HIP(ctx, cs.dyn_px.reserve(cap));
	LMX_HIP(ctx, cs.dyn_py.reserve(cap));
	LMX_HIP(ctx, cs.dyn_pz.reserve(cap));
	LMX_HIP(ctx, cs.dyn_radius.reserve(cap));
	LMX_HIP(ctx, cs.dyn_ids.reserve(cap));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (padded) {
		LMX_HIP(ctx, hipMemcpy(cs.dyn_px.p, px.data(), padded * sizeof(double), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(cs.dyn_py.p, py.data(), padded * sizeof(double), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(cs.dyn_pz.p, pz.data(), padded * sizeof(double), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(cs.dyn_radius.p, radius.data(), padded * sizeof(float), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(cs.dyn_ids.p, ids.data(), padded * sizeof(int32_t), hipMemcpyHostToDevice));
	}
	cs.dyn_layout_dirty = false;
	cs.dyn_values_dirty = false;
	cs.dyn_generation++;
	return LMX_OK;
}

// host-side changes of pos / radius of dynamic entities: re-upload the four value arrays (slots are unchanged)
int upload_dynamic_values(LmxContext* ctx) {
	CullState& cs = ctx->cull;
	const size_t padded = cs.dyn_padded;
	if (padded) {
		std::vector<double> px(padded, 0.0), py(padded, 0.0), pz(padded, 0.0);
		std::vector<float> radius(padded, 0.f);
		for (size_t i = 0; i < cs.dyn.size(); ++i) {
			const size_t s = cs.dyn_slot[i];
			px[s] = cs.dyn[i].pos[0];
			py[s] = cs.dyn[i].pos[1];
			pz[s] = cs.dyn[i].pos[2];
			radius[s] = cs.dyn[i].radius;
		}
		LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		LMX_HIP(ctx, hipMemcpy(cs.dyn_px.p, px.data(), padded * sizeof(double), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(cs.dyn_py.p, py.data(), padded * sizeof(double), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(cs.dyn_pz.p, pz.data(), padded * sizeof(double), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(cs.dyn_radius.p, radius.data(), padded * sizeof(float), hipMemcpyHostToDevice));
	}
	cs.dyn_values_dirty = false;
	return LMX_OK;
}

void recompute_out_layout(CullState& cs) {
	uint32_t off = 0;
	for (int t = 0; t < MAX_TYPES; ++t) {
		cs.tt.out_start[t] = off;
		cs.dyn_tt.out_start[t] = off;
		off += (cs.tt.ent_end[t] - cs.tt.ent_start[t]) + (cs.dyn_tt.ent_end[t] - cs.dyn_tt.ent_start[t]);
	}
	cs.out_total = off;
}

void readd_static(CullState& cs, uint32_t rec, DV3 pos, float radius) { // remove(entity); add(entity, type, pos, radius)
	const CullRec old = cs.recs[rec];
	cs.recs[rec] = make_cull_rec(old.entity, old.type, pos, radius);
	cs.structure_dirty = true;
}

void remove_static(CullState& cs, uint32_t rec) {
	const int32_t entity = cs.recs[rec].entity;
	const uint32_t last = (uint32_t)cs.recs.size() - 1;
	if (rec != last) {
		cs.recs[rec] = cs.recs[last];
		cs.ent_to_rec[cs.recs[rec].entity] = (int32_t)rec;
	}
	cs.recs.pop_back();
	cs.ent_to_rec[entity] = -1;
	cs.structure_dirty = true;
}

void remove_dynamic(CullState& cs, uint32_t idx) {
	const int32_t entity = cs.dyn[idx].entity;
	const uint32_t last = (uint32_t)cs.dyn.size() - 1;
	if (idx != last) {
		cs.dyn[idx] = cs.dyn[last];
		cs.ent_to_dyn[cs.dyn[idx].entity] = (int32_t)idx;
	}
	cs.dyn.pop_back();
	cs.ent_to_dyn[entity] = -1;
	cs.dyn_layout_dirty = true;
}

// What the reference's stored state (cell, cell-relative fp32 position) means as a world position:
// cell.header.origin + sphere->position (culling_system.cpp:255)
DV3 stored_position(DV3 pos) {
	const IV3 idx = cell_of(pos);
	const DV3 origin = cell_origin(idx);
	return add(origin, to_v3(sub(pos, origin)));
}

CullDeviceView static_view(const CullState& cs) {
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
		v.tile_box[k] = cs.tile_box[k].p;
		v.tile_cap[k] = cs.tile_cap[k];
	}
	return v;
}

} // namespace

namespace lmx {

int cull_dyn_sync_mirror(LmxContext* ctx) {
	CullState& cs = ctx->cull;
	if (!cs.dyn_mirror_stale) return LMX_OK;
	cs.dyn_mirror_stale = false;
	if (cs.dyn_layout_dirty || !cs.dyn_padded) return LMX_OK; // (propagate always flushes first, so slots are current)
	const size_t padded = cs.dyn_padded;
	std::vector<double> px(padded), py(padded), pz(padded);
	std::vector<float> radius(padded);
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	LMX_HIP(ctx, hipMemcpy(px.data(), cs.dyn_px.p, padded * sizeof(double), hipMemcpyDeviceToHost));
	LMX_HIP(ctx, hipMemcpy(py.data(), cs.dyn_py.p, padded * sizeof(double), hipMemcpyDeviceToHost));
	LMX_HIP(ctx, hipMemcpy(pz.data(), cs.dyn_pz.p, padded * sizeof(double), hipMemcpyDeviceToHost));
	LMX_HIP(ctx, hipMemcpy(radius.data(), cs.dyn_radius.p, padded * sizeof(float), hipMemcpyDeviceToHost));
	for (size_t i = 0; i < cs.dyn.size(); ++i) {
		const size_t s = cs.dyn_slot[i];
		cs.dyn[i].pos[0] = px[s];
		cs.dyn[i].pos[1] = py[s];
		cs.dyn[i].pos[2] = pz[s];
		cs.dyn[i].radius = radius[s];
	}
	return LMX_OK;
}

bool cull_make_dynamic(LmxContext* ctx, int32_t entity) {
	CullState& cs = ctx->cull;
	uint32_t idx;
	const Where w = locate(cs, entity, &idx);
	if (w == Where::DYNAMIC) return true;
	if (w != Where::STATIC) return false;
	const CullRec r = cs.recs[idx];
	const DV3 pos = add(cell_origin(r.cell), r.rel);
	remove_static(cs, idx);
	if ((size_t)entity >= cs.ent_to_dyn.size()) cs.ent_to_dyn.resize((size_t)entity + 1, -1);
	cs.ent_to_dyn[entity] = (int32_t)cs.dyn.size();
	cs.dyn.push_back(DynRec{{pos.x, pos.y, pos.z}, r.radius, entity, r.type});
	cs.dyn_layout_dirty = true;
	return true;
}

int cull_flush(LmxContext* ctx) {
	CullState& cs = ctx->cull;
	bool layout_changed = false;
	if (cs.structure_dirty || !cs.built) {
		if (int rc = rebuild_static(ctx)) return rc;
		layout_changed = true;
	} else if (int rc = apply_patches(ctx)) {
		return rc;
	}
	if (cs.dyn_layout_dirty) {
		if (int rc = cull_dyn_sync_mirror(ctx)) return rc; // keep what the device refreshed before slots move
		if (int rc = rebuild_dynamic(ctx)) return rc;
		layout_changed = true;
	} else if (cs.dyn_values_dirty) {
		if (int rc = upload_dynamic_values(ctx)) return rc;
	}
	if (layout_changed) {
		recompute_out_layout(cs);
		for (CullView& v : cs.views) v.valid = false;
	}
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
	cs.recs.reserve(n);
	cs.dyn.clear();
	cs.ent_to_dyn.clear();
	cs.dyn_layout_dirty = true;
	cs.dyn_mirror_stale = false;
	cs.ent_to_rec.assign((size_t)max_entity + 1, -1);
	cs.structure_dirty = true;
	for (uint32_t i = 0; i < n; ++i) {
		if (cs.ent_to_rec[entity[i]] >= 0) {
			cs.recs.clear();
			cs.ent_to_rec.clear();
			return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d added twice", entity[i]);
		}
		cs.ent_to_rec[entity[i]] = (int32_t)i;
		cs.recs.push_back(make_cull_rec(entity[i], type[i], DV3{pos_xyz[3 * (size_t)i], pos_xyz[3 * (size_t)i + 1], pos_xyz[3 * (size_t)i + 2]}, radius[i]));
	}
	return cull_flush(ctx);
}

int lmx_cull_add(LmxContext* ctx, int32_t entity, uint8_t type, const double pos[3], float radius) { // culling_system.cpp:131-157
	LMX_CHECK_CTX(ctx);
	if (entity < 0 || !pos) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad entity/pos");
	if (type >= MAX_TYPES) return fail(ctx, LMX_ERR_CAPACITY, "type %u >= LMX_MAX_TYPES", type);
	CullState& cs = ctx->cull;
	uint32_t idx;
	if (locate(cs, entity, &idx) != Where::NONE) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d already added", entity);
	if ((size_t)entity >= cs.ent_to_rec.size()) cs.ent_to_rec.resize((size_t)entity + 1, -1);
	cs.ent_to_rec[entity] = (int32_t)cs.recs.size();
	cs.recs.push_back(make_cull_rec(entity, type, DV3{pos[0], pos[1], pos[2]}, radius));
	cs.structure_dirty = true;
	return LMX_OK;
}

int lmx_cull_remove(LmxContext* ctx, int32_t entity) { // culling_system.cpp:160-190 (unknown entities are ignored, :162-165)
	LMX_CHECK_CTX(ctx);
	CullState& cs = ctx->cull;
	uint32_t idx;
	switch (locate(cs, entity, &idx)) {
		case Where::STATIC: remove_static(cs, idx); break;
		case Where::DYNAMIC:
			if (int rc = cull_dyn_sync_mirror(ctx)) return rc;
			remove_dynamic(cs, idx);
			break;
		case Where::NONE: break;
	}
	return LMX_OK;
}

int lmx_cull_set(LmxContext* ctx, int32_t entity, const double pos[3], float radius) { // culling_system.cpp:225-242
	LMX_CHECK_CTX(ctx);
	if (!pos) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null pos");
	CullState& cs = ctx->cull;
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
			return LMX_OK;
		}
		case Where::DYNAMIC: {
			if (int rc = cull_dyn_sync_mirror(ctx)) return rc;
			DynRec& r = cs.dyn[idx];
			r.pos[0] = p.x; r.pos[1] = p.y; r.pos[2] = p.z;
			r.radius = radius;
			cs.dyn_values_dirty = true;
			return LMX_OK;
		}
		case Where::NONE: break;
	}
	return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d is not in the culling system", entity);
}

int lmx_cull_set_position(LmxContext* ctx, int32_t entity, const double pos[3]) { // culling_system.cpp:201-217
	LMX_CHECK_CTX(ctx);
	if (!pos) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null pos");
	CullState& cs = ctx->cull;
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
			return LMX_OK;
		}
		case Where::DYNAMIC: {
			if (int rc = cull_dyn_sync_mirror(ctx)) return rc;
			DynRec& r = cs.dyn[idx];
			r.pos[0] = p.x; r.pos[1] = p.y; r.pos[2] = p.z;
			cs.dyn_values_dirty = true;
			return LMX_OK;
		}
		case Where::NONE: break;
	}
	return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d is not in the culling system", entity);
}

int lmx_cull_set_radius(LmxContext* ctx, int32_t entity, float radius) { // culling_system.cpp:244-260
	LMX_CHECK_CTX(ctx);
	CullState& cs = ctx->cull;
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
			return LMX_OK;
		}
		case Where::DYNAMIC: {
			if (int rc = cull_dyn_sync_mirror(ctx)) return rc;
			DynRec& r = cs.dyn[idx];
			if (is_big_radius(r.radius) != is_big_radius(radius)) {
				// the reference re-adds at origin + fp32 relative position, which loses the low bits of the position
				const DV3 p = stored_position(DV3{r.pos[0], r.pos[1], r.pos[2]});
				r.pos[0] = p.x; r.pos[1] = p.y; r.pos[2] = p.z;
			}
			r.radius = radius;
			cs.dyn_values_dirty = true;
			return LMX_OK;
		}
		case Where::NONE: break;
	}
	return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d is not in the culling system", entity);
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
			if (int rc = cull_dyn_sync_mirror(ctx)) return rc;
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

int lmx_cull_flush(LmxContext* ctx) {
	LMX_CHECK_CTX(ctx);
	return cull_flush(ctx);
}

int lmx_cull_stats(LmxContext* ctx, uint32_t* n_entities, uint32_t* n_cells, uint32_t* n_chunks) {
	LMX_CHECK_CTX(ctx);
	if (int rc = cull_flush(ctx)) return rc;
	const CullState& cs = ctx->cull;
	if (n_entities) *n_entities = (uint32_t)(cs.recs.size() + cs.dyn.size());
	if (n_cells) *n_cells = cs.n_cells - cs.n_dead_cells;
	if (n_chunks) *n_chunks = cs.out_total / CHUNK;
	return LMX_OK;
}

// Share of the scene's bounding box that the frustum's bounding box covers (8 corner points + fp64 origin, geometry.h:102-153):
// a launch-time hint for the tile size of the 1-frustum kernel, nothing the results depend on.
static double frustum_scene_fraction(const CullState& cs, const LmxShiftedFrustum& f) {
	double frac = 1.0;
	for (int a = 0; a < 3; ++a) {
		double lo = INFINITY, hi = -INFINITY;
		for (int p = 0; p < 8; ++p) {
			const double v = (double)f.points[p][a] + f.origin[a];
			lo = std::min(lo, v);
			hi = std::max(hi, v);
		}
		const double s_lo = cs.scene_lo[a], s_hi = cs.scene_hi[a];
		if (!(s_hi > s_lo) || !(hi >= lo)) return 1.0; // empty scene / NaN frustum: no hint
		const double overlap = std::min(hi, s_hi) - std::max(lo, s_lo);
		frac *= overlap <= 0 ? 0.0 : overlap / (s_hi - s_lo);
	}
	return frac;
}

int lmx_cull(LmxContext* ctx, uint32_t view, const LmxShiftedFrustum* frusta, uint32_t n_frusta, uint8_t type) {
	LMX_CHECK_CTX(ctx);
	if (view >= LMX_MAX_VIEWS) return fail(ctx, LMX_ERR_CAPACITY, "view %u >= LMX_MAX_VIEWS", view);
	if (!frusta || n_frusta == 0 || n_frusta > LMX_MAX_FRUSTA) return fail(ctx, LMX_ERR_CAPACITY, "n_frusta %u not in [1,%d]", n_frusta, LMX_MAX_FRUSTA);
	if (type != LMX_TYPE_ALL && type >= MAX_TYPES) return fail(ctx, LMX_ERR_CAPACITY, "type %u >= LMX_MAX_TYPES", type);
	if (int rc = cull_flush(ctx)) return rc;
	CullState& cs = ctx->cull;
	CullView& v = cs.views[view];
	const size_t row = std::max(cs.out_total, 1u);
	if (v.ext_out) {
		if (v.ext_out_cap < (size_t)cs.out_total * n_frusta)
			return fail(ctx, LMX_ERR_CAPACITY, "bound output holds %zu ids, need %zu", v.ext_out_cap, (size_t)cs.out_total * n_frusta);
	} else {
		if (!v.counts.p) {
			LMX_HIP(ctx, v.counts.reserve(2 * MAX_FRUSTA * MAX_TYPES));
			v.flip = 0;
			v.next_half_is_zero = false;
		}
		LMX_HIP(ctx, v.out.reserve(row * n_frusta));
	}
	v.n_frusta = n_frusta;
	v.cell_stride = cs.n_cells;
	v.out_stride = cs.out_total;
	for (int t = 0; t < MAX_TYPES; ++t) {
		v.out_start[t] = cs.tt.out_start[t];
		v.out_cap[t] = (cs.tt.ent_end[t] - cs.tt.ent_start[t]) + (cs.dyn_tt.ent_end[t] - cs.dyn_tt.ent_start[t]);
	}
	FrustaArg fr;
	memset(&fr, 0, sizeof(fr));
	for (uint32_t f = 0; f < n_frusta; ++f) fr.f[f] = to_dev_frustum(frusta[f]);

	uint32_t cell_begin = 0, cell_n = cs.n_cells, ent_begin = 0, ent_end = cs.n_padded, dyn_begin = 0, dyn_end = cs.dyn_padded;
	if (type != LMX_TYPE_ALL) {
		cell_begin = cs.cell_begin[type];
		cell_n = cs.cell_end[type] - cs.cell_begin[type];
		ent_begin = cs.tt.ent_start[type];
		ent_end = cs.tt.ent_end[type];
		dyn_begin = cs.dyn_tt.ent_start[type];
		dyn_end = cs.dyn_tt.ent_end[type];
	}
	const CullDeviceView dv = static_view(cs);
	// static set: fused single-launch kernel; the layout bounds the cells per tile so its LDS table always fits. The
	// classify + spheres pair stays available as an ablation / fallback (LMX_CULL_TWO_KERNELS=1).
	static const bool force_two_kernels = getenv("LMX_CULL_TWO_KERNELS") != nullptr;
	bool fused = !force_two_kernels;
	for (int k = 0; k < 3 && fused; ++k) fused = fused_lds_bytes(k == 0 ? 1 : (k == 1 ? 4 : 8), TILE_ALIGN >> k, cs.tile_cap[k]) <= 64 * 1024;
	if (fused) {
		uint32_t* counts_next = nullptr;
		if (v.ext_counts) {
			LMX_HIP(ctx, hipMemsetAsync(v.ext_counts, 0, sizeof(uint32_t) * MAX_FRUSTA * MAX_TYPES, ctx->stream));
		} else {
			v.flip ^= 1u;
			if (!v.next_half_is_zero) LMX_HIP(ctx, hipMemsetAsync(v.counts_ptr(), 0, sizeof(uint32_t) * MAX_FRUSTA * MAX_TYPES, ctx->stream));
			counts_next = v.counts_other();
			v.next_half_is_zero = ent_end > ent_begin; // block 0 of the launch(es) below clears it
		}
		// The kernel is latency-bound, not bandwidth-bound: wide variants (many frusta per pass) hold more state per wave
		// and run at lower occupancy, so a batch is split into passes of at most `pass_width` frusta.
		const uint32_t pass_width = cs.pass_width;
		for (uint32_t f0 = 0; f0 < n_frusta; f0 += pass_width) {
			const uint32_t fw = std::min(pass_width, n_frusta - f0);
			FrustaArg sub;
			memset(&sub, 0, sizeof(sub));
			for (uint32_t k = 0; k < fw; ++k) sub.f[k] = fr.f[f0 + k];
			ProfScope ps(ctx, LMX_K_CULL_SPHERES);
			LMX_HIP(ctx, launch_cull_fused(ctx->stream, dv, ent_begin, ent_end, cs.tt, sub, (int)fw, v.out_ptr() + (size_t)f0 * v.out_stride, v.out_stride,
				v.counts_ptr() + f0 * MAX_TYPES, counts_next, fw == 1 && frustum_scene_fraction(cs, frusta[f0]) < 0.25));
		}
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
	// dynamic set: appended to the same segments / counters
	if (dyn_end > dyn_begin) {
		DynDeviceView dd;
		dd.px = cs.dyn_px.p;
		dd.py = cs.dyn_py.p;
		dd.pz = cs.dyn_pz.p;
		dd.radius = cs.dyn_radius.p;
		dd.ids = cs.dyn_ids.p;
		dd.n_padded = cs.dyn_padded;
		ProfScope ps(ctx, LMX_K_CULL_DYNAMIC);
		LMX_HIP(ctx, launch_cull_dynamic(ctx->stream, dd, dyn_begin, dyn_end, cs.dyn_tt, fr, (int)n_frusta, v.out_ptr(), v.out_stride, v.counts_ptr()));
	}
	v.valid = true;
	return LMX_OK;
}

int lmx_cull_set_pass_width(LmxContext* ctx, uint32_t frusta_per_pass) {
	LMX_CHECK_CTX(ctx);
	if (frusta_per_pass < 1 || frusta_per_pass > LMX_MAX_FRUSTA) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "pass width %u not in [1,%d]", frusta_per_pass, LMX_MAX_FRUSTA);
	ctx->cull.pass_width = frusta_per_pass;
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
	if (c > v.out_cap[type]) return fail(ctx, LMX_ERR_HIP, "corrupt count %u > %u", c, v.out_cap[type]);
	if (!out_ids || c == 0) return LMX_OK;
	if (c > cap) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u ids, got %u", c, cap);
	LMX_HIP(ctx, hipMemcpyAsync(out_ids, v.out_ptr() + (size_t)frustum * v.out_stride + v.out_start[type], (size_t)c * sizeof(int32_t),
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
	if (type_offsets) memcpy(type_offsets, v.out_start, sizeof(v.out_start));
	if (capacity) *capacity = v.out_stride;
	return LMX_OK;
}

} // extern "C"
