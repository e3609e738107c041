// emul_kernels.cpp — TEST INFRASTRUCTURE: sequential host emulation of the gfx950 kernels' LOGIC.
//
// There is no GPU in the development container, so the parts of the device path that can be checked on a CPU are
// checked on a CPU before a GPU run is spent: the shared arithmetic header (lmx_math.h, compiled here for x86 with
// -ffp-contract=off), the device layout builder (lmx_cull_layout.h: sort, padding, dead cells, chunk headers) and the
// per-chunk cell resolution (`base_cell + popcount(flags & bits 1..lane)`). The emulation mirrors cull_kernels.hip /
// xform_kernels.hip / skin_kernels.hip statement by statement but runs lanes in a loop. It is NOT a fallback: nothing
// in lumixengine_amd/ links it.
#include <cstdint>
#include <cstring>
#include <vector>

#include "lmx_cull_layout.h"
#include "lmx_math.h"
#include "lmx_types.h"

using namespace lmx;

// tiles visited, tiles ended by TILE_REJECT, tiles whose cells are all CELL_REJECT, tiles taken by TILE_ACCEPT, tiles whose live cells are all CELL_ACCEPT
static uint32_t g_tile_stats[5];
static uint64_t g_skip_stats[2]; // planes of MIXED tiles: {possible, left out by tile_plane_skip_mask}

extern "C" {

void emul_tile_stats(uint32_t* out) { memcpy(out, g_tile_stats, sizeof(g_tile_stats)); }
void emul_skip_stats(uint64_t* out) { memcpy(out, g_skip_stats, sizeof(g_skip_stats)); }

// tile_status for a hand-built box (tests of the margin with scaled / adversarial planes)
uint32_t emul_tile_status(const LmxShiftedFrustum* f, const int32_t* lo, const int32_t* hi, uint32_t flags) {
	TileBox b = {{lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}, flags, 0};
	return tile_status(to_dev_frustum(*f), b);
}
// classify_cell for one cell
uint32_t emul_classify_cell(const LmxShiftedFrustum* f, const int32_t* idx, int big) {
	V3 off;
	return classify_cell(to_dev_frustum(*f), IV3{idx[0], idx[1], idx[2]}, big != 0, &off);
}

// emulates lmx_cull_build + lmx_cull over n_frusta frusta; out_ids / out_types are [n_frusta][n] (first
// sum(out_counts[f]) entries used), out_counts [n_frusta][8]. `tile_variant`: tile size of the 1-frustum kernel
// (0: 4096, 1 / 2: 2048, 3: 1024), as lmx_cull_set_option(LMX_CULL_OPT_TILE_VARIANT).
int emul_cull_variant(uint32_t n, const int32_t* entity, const uint8_t* type, const double* pos, const float* radius,
	const LmxShiftedFrustum* frusta, uint32_t n_frusta, uint8_t type_filter, int tile_variant, int32_t* out_ids, uint8_t* out_types, uint32_t* out_counts) {
	std::vector<CullRec> recs(n);
	for (uint32_t i = 0; i < n; ++i) recs[i] = make_cull_rec(entity[i], type[i], DV3{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]}, radius[i]);
	CullLayout lay;
	if (!build_cull_layout(recs, lay)) return 1;
	const size_t n_cells = lay.cells.size();
	memset(out_counts, 0, sizeof(uint32_t) * n_frusta * LAYOUT_MAX_TYPES);
	memset(g_tile_stats, 0, sizeof(g_tile_stats));
	memset(g_skip_stats, 0, sizeof(g_skip_stats));
	// the live ids of every TILE_ALIGN block must add up to what the shard windows are sized for
	if (lay.block_live.size() != lay.n_padded / LAYOUT_TILE_ALIGN) return 9;
	{
		size_t live = 0;
		for (uint32_t b : lay.block_live) live += b;
		if (live != n) return 9;
	}
	struct Info { float d[6]; uint32_t cls; };
	for (uint32_t f = 0; f < n_frusta; ++f) {
		uint32_t total = 0;
		const DevFrustum fr = to_dev_frustum(frusta[f]);
		// phase A of k_cull_tile for every cell (the kernel does it tile by tile into LDS): class + cell-relative plane distances
		std::vector<Info> info(n_cells);
		for (size_t c = 0; c < n_cells; ++c) {
			const LayoutCell key = lay.cells[c];
			Info ci = {{0, 0, 0, 0, 0, 0}, CELL_REJECT};
			if (!(key.meta & LAYOUT_CELL_DEAD)) {
				V3 off;
				ci.cls = classify_cell(fr, IV3{key.ix, key.iy, key.iz}, (key.meta & 0x100u) != 0, &off);
				if (ci.cls == CELL_TEST) {
					for (int k = 0; k < 6; ++k) ci.d[k] = relative_plane_d(fr, off, k);
				}
			}
			info[c] = ci;
		}
		uint32_t ent_begin = 0, ent_end = lay.n_padded;
		if (type_filter != 0xff) {
			ent_begin = lay.ent_start[type_filter];
			ent_end = lay.ent_end[type_filter];
		}
		// tile bookkeeping: the cells a tile touches are [first_cell, last_cell], bounded by the layout's max
		const uint32_t tile = n_frusta <= 1 ? ((tile_variant == 0 || tile_variant == 5) ? 4096u : (tile_variant == 3 ? 1024u : 2048u)) : (n_frusta <= 4 ? 2048u : 1024u);
		const uint32_t tile_k = tile == 4096 ? 0 : (tile == 2048 ? 1 : 2);
		const uint32_t nch = tile / 64;
		for (uint32_t chunk = ent_begin / 64; chunk < ent_end / 64; ++chunk) {
			const uint32_t tile_chunk = (chunk / nch) * nch;
			const uint32_t tile_index = chunk / nch;
			const uint32_t cap = lay.tile_cap[tile_k];
			const uint32_t first_cell = lay.tile_tab[tile_k][2 * tile_index];
			const uint32_t tile_n_cells = lay.tile_tab[tile_k][2 * tile_index + 1];
			const LayoutChunkHdr last_hdr = lay.hdr[tile_chunk + nch - 1];
			const uint32_t last_cell = last_hdr.cell + (uint32_t)__builtin_popcountll(last_hdr.flags & ~1ull);
			if (first_cell != lay.hdr[tile_chunk].cell || last_cell != first_cell + tile_n_cells - 1) return 3;
			if (tile_n_cells > cap || tile_n_cells > lay.max_tile_cells[tile_k] || last_cell >= n_cells) return 4;
			if (n_frusta <= 8 && (size_t)n_frusta * cap * 32 > 65536) return 6; // LDS budget of the per-tile cell table
			const uint32_t st = tile_status(fr, lay.tile_box[tile_k][tile_index]);
			if (chunk == tile_chunk) { // TILE_DENSE (accepted tile = straight copy of its ids) <=> no slot of the tile is padding
				bool all_live = true;
				for (uint32_t e = tile_chunk * 64; e < (tile_chunk + nch) * 64; ++e) all_live = all_live && lay.ids[e] >= 0;
				if (all_live != ((lay.tile_box[tile_k][tile_index].flags & TILE_DENSE) != 0)) return 11;
			}
			if (chunk == tile_chunk) { // once per tile
				bool none = true, all_in = true, any_cell = false;
				for (uint32_t c = first_cell; c <= last_cell; ++c) {
					none = none && info[c].cls == CELL_REJECT;
					if (lay.cells[c].meta & LAYOUT_CELL_DEAD) continue;
					any_cell = true;
					all_in = all_in && info[c].cls == CELL_ACCEPT;
				}
				g_tile_stats[0]++;
				g_tile_stats[1] += st == TILE_REJECT ? 1u : 0u;
				g_tile_stats[2] += none ? 1u : 0u;
				g_tile_stats[3] += st == TILE_ACCEPT ? 1u : 0u;
				g_tile_stats[4] += (any_cell && all_in) ? 1u : 0u;
			}
			// the tile-level verdicts may only fire when the cell-by-cell classification agrees for every cell of the tile
			if (st == TILE_REJECT) {
				for (uint32_t c = first_cell; c <= last_cell; ++c)
					if (info[c].cls != CELL_REJECT) return 8;
				continue;
			}
			if (st == TILE_ACCEPT) {
				for (uint32_t c = first_cell; c <= last_cell; ++c)
					if (!(lay.cells[c].meta & LAYOUT_CELL_DEAD) && info[c].cls != CELL_ACCEPT) return 10;
			}
			if (st == TILE_MIXED && chunk == tile_chunk) { // phase A leaves out the planes the whole tile is known to pass: same class for every cell
				const uint32_t skip = tile_plane_skip_mask(fr, lay.tile_box[tile_k][tile_index]);
				g_skip_stats[0] += 6;
				g_skip_stats[1] += (uint64_t)__builtin_popcount(skip);
				for (uint32_t c = first_cell; c <= last_cell && skip; ++c) {
					const LayoutCell key = lay.cells[c];
					if (key.meta & LAYOUT_CELL_DEAD) continue;
					V3 off;
					if (classify_cell(fr, IV3{key.ix, key.iy, key.iz}, (key.meta & 0x100u) != 0, &off, skip) != info[c].cls) return 13;
				}
			}
			uint32_t t = 0;
			for (int k = 0; k < LAYOUT_MAX_TYPES; ++k)
				if (chunk * 64 >= lay.ent_start[k] && chunk * 64 < lay.ent_end[k]) t = (uint32_t)k;
			const LayoutChunkHdr h = lay.hdr[chunk];
			for (uint32_t lane = 0; lane < 64; ++lane) {
				const uint64_t le_mask = (~0ull >> (63u - lane)) & ~1ull;
				const uint32_t cell = h.cell + (uint32_t)__builtin_popcountll(h.flags & le_mask);
				if (cell != lay.slot_cell[chunk * 64 + lane]) return 2; // chunk header does not reproduce the slot->cell map
				if (cell < first_cell || cell > last_cell) return 5;    // tile-local LDS index would be out of range
				// phase A: the class comes from the tile-major copy of the cell key
				const LayoutCell tkey = lay.tile_cells[tile_k][(size_t)tile_index * cap + (cell - first_cell)];
				const LayoutCell gkey = lay.cells[cell];
				if (tkey.ix != gkey.ix || tkey.iy != gkey.iy || tkey.iz != gkey.iz || tkey.meta != gkey.meta) return 7;
				const uint32_t e = chunk * 64 + lane;
				const int32_t id = lay.ids[e];
				bool vis;
				if (st == TILE_MIXED) {
					const Info& ci = info[cell];
					vis = ci.cls == CELL_ACCEPT;
					if (ci.cls == CELL_TEST) {
						const LayoutSphere s = lay.spheres[e];
						vis = sphere_visible_d(fr, ci.d, s.x, s.y, s.z, s.radius);
					}
				} else {
					vis = true; // TILE_ACCEPT: ids are copied without looking at cells or spheres
				}
				vis = vis && id >= 0;
				if (vis) {
					out_counts[f * LAYOUT_MAX_TYPES + t]++;
					out_ids[(size_t)f * n + total] = id;
					out_types[(size_t)f * n + total] = (uint8_t)t;
					++total;
				}
			}
		}
	}
	return 0;
}

int emul_cull(uint32_t n, const int32_t* entity, const uint8_t* type, const double* pos, const float* radius,
	const LmxShiftedFrustum* frusta, uint32_t n_frusta, uint8_t type_filter, int32_t* out_ids, uint8_t* out_types, uint32_t* out_counts) {
	return emul_cull_variant(n, entity, type, pos, radius, frusta, n_frusta, type_filter, 0, out_ids, out_types, out_counts);
}

// emulates k_cull_dynamic: every entity derives cell / is_big / cell-relative position from its fp64 position and
// classifies its own cell; out_ids / out_types as in emul_cull
int emul_cull_dynamic(uint32_t n, const int32_t* entity, const uint8_t* type, const double* pos, const float* radius,
	const LmxShiftedFrustum* frusta, uint32_t n_frusta, int32_t* out_ids, uint8_t* out_types, uint32_t* out_counts) {
	memset(out_counts, 0, sizeof(uint32_t) * n_frusta * LAYOUT_MAX_TYPES);
	for (uint32_t f = 0; f < n_frusta; ++f) {
		const DevFrustum fr = to_dev_frustum(frusta[f]);
		uint32_t total = 0;
		for (uint32_t i = 0; i < n; ++i) {
			const DV3 p = DV3{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
			const IV3 idx = cell_of(p);
			const bool big = is_big_radius(radius[i]);
			const V3 rel = to_v3(sub(p, cell_origin(idx)));
			V3 off;
			const uint32_t cls = classify_cell(fr, idx, big, &off);
			bool vis = cls == CELL_ACCEPT;
			if (cls == CELL_TEST) vis = sphere_visible(fr, off, rel.x, rel.y, rel.z, radius[i]);
			if (vis) {
				out_counts[f * LAYOUT_MAX_TYPES + type[i]]++;
				out_ids[(size_t)f * n + total] = entity[i];
				out_types[(size_t)f * n + total] = type[i];
				++total;
			}
		}
	}
	return 0;
}

// emulates lmx_world_build + lmx_world_propagate (level order, compose) -> world transforms by entity
int emul_world(uint32_t n, const int32_t* parent, const LmxTransform* tr, LmxTransform* out) {
	std::vector<int> depth(n, -1);
	std::vector<uint32_t> order;
	order.reserve(n);
	// same BFS as lmx_world_build
	std::vector<uint32_t> child_start((size_t)n + 1, 0);
	for (uint32_t e = 0; e < n; ++e)
		if (parent[e] >= 0) child_start[(size_t)parent[e] + 1]++;
	for (uint32_t e = 0; e < n; ++e) child_start[e + 1] += child_start[e];
	std::vector<uint32_t> child_list(child_start[n]);
	{
		std::vector<uint32_t> cursor(child_start.begin(), child_start.end() - 1);
		for (uint32_t e = 0; e < n; ++e)
			if (parent[e] >= 0) child_list[cursor[parent[e]]++] = e;
	}
	for (uint32_t e = 0; e < n; ++e)
		if (parent[e] < 0) order.push_back(e);
	for (size_t s = 0; s < order.size(); ++s) {
		const uint32_t e = order[s];
		for (uint32_t k = child_start[e]; k < child_start[e + 1]; ++k) order.push_back(child_list[k]);
	}
	if (order.size() != n) return 1;
	auto load = [](const LmxTransform& t) {
		Xform x;
		x.pos = DV3{t.pos[0], t.pos[1], t.pos[2]};
		x.rot = Q4{t.rot[0], t.rot[1], t.rot[2], t.rot[3]};
		x.scale = V3{t.scale[0], t.scale[1], t.scale[2]};
		return x;
	};
	std::vector<Xform> world(n);
	for (uint32_t s = 0; s < n; ++s) {
		const uint32_t e = order[s];
		world[e] = parent[e] < 0 ? load(tr[e]) : compose(world[parent[e]], load(tr[e]));
	}
	for (uint32_t e = 0; e < n; ++e) {
		memset(&out[e], 0, sizeof(LmxTransform));
		out[e].pos[0] = world[e].pos.x; out[e].pos[1] = world[e].pos.y; out[e].pos[2] = world[e].pos.z;
		out[e].rot[0] = world[e].rot.x; out[e].rot[1] = world[e].rot.y; out[e].rot[2] = world[e].rot.z; out[e].rot[3] = world[e].rot.w;
		out[e].scale[0] = world[e].scale.x; out[e].scale[1] = world[e].scale.y; out[e].scale[2] = world[e].scale.z;
	}
	return 0;
}

// emulates k_pose_palette (level-by-level walk) + k_skin_vertices for one instance
int emul_skin(uint32_t n_bones, const int16_t* parents, int32_t first_nonroot, const LmxLocalRigidTransform* bind, float* pose_pos,
	float* pose_rot, LmxMatrix* palette, uint32_t n_verts, const float* verts, const LmxSkin* skin, float* out) {
	std::vector<uint8_t> depth(n_bones, 0);
	uint32_t max_depth = 0;
	for (uint32_t i = 0; i < n_bones; ++i) {
		depth[i] = ((int32_t)i >= first_nonroot) ? (uint8_t)(depth[parents[i]] + 1) : 0;
		if (depth[i] > max_depth) max_depth = depth[i];
	}
	V3* pos = (V3*)pose_pos;
	Q4* rot = (Q4*)pose_rot;
	for (uint32_t d = 1; d <= max_depth; ++d) {
		for (uint32_t b = 0; b < n_bones; ++b) {
			if (depth[b] == d && (int32_t)b >= first_nonroot) {
				const int32_t p = parents[b];
				const V3 np = add(rotate(rot[p], pos[b]), pos[p]);
				const Q4 nr = qmul(rot[p], rot[b]);
				pos[b] = np;
				rot[b] = nr;
			}
		}
	}
	for (uint32_t b = 0; b < n_bones; ++b) {
		V3 ip;
		Q4 ir;
		invert_rigid(V3{bind[b].pos[0], bind[b].pos[1], bind[b].pos[2]}, Q4{bind[b].rot[0], bind[b].rot[1], bind[b].rot[2], bind[b].rot[3]}, &ip, &ir);
		const Mat4 m = skin_matrix(pos[b], rot[b], ip, ir);
		memcpy(&palette[b], &m, sizeof(m));
	}
	for (uint32_t v = 0; v < n_verts; ++v) {
		const float px = verts[3 * v], py = verts[3 * v + 1], pz = verts[3 * v + 2];
		const float* w = skin[v].weights;
		float o[3];
		for (int r = 0; r < 3; ++r) {
			float a[4], b[4], c[4], d[4];
			for (int col = 0; col < 4; ++col) {
				a[col] = palette[skin[v].indices[0]].columns[col][r];
				b[col] = palette[skin[v].indices[1]].columns[col][r];
				c[col] = palette[skin[v].indices[2]].columns[col][r];
				d[col] = palette[skin[v].indices[3]].columns[col][r];
			}
			const float m0 = a[0] * w[0] + b[0] * w[1] + c[0] * w[2] + d[0] * w[3];
			const float m1 = a[1] * w[0] + b[1] * w[1] + c[1] * w[2] + d[1] * w[3];
			const float m2 = a[2] * w[0] + b[2] * w[1] + c[2] * w[2] + d[2] * w[3];
			const float m3 = a[3] * w[0] + b[3] * w[1] + c[3] * w[2] + d[3] * w[3];
			o[r] = m0 * px + m1 * py + m2 * pz + m3;
		}
		out[3 * v] = o[0];
		out[3 * v + 1] = o[1];
		out[3 * v + 2] = o[2];
	}
	return 0;
}

} // extern "C"
