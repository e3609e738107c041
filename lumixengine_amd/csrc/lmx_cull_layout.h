// lmx_cull_layout.h — pure-host construction of the device layout of the culling set (no HIP types), shared by
// lmx_capi.hip (which uploads it) and the CPU-side layout tests.
//
// Spheres are sorted by (type, is_big, cell.x, cell.y, cell.z): every occupied CellIndices group of the reference
// (src/renderer/culling_system.cpp:23-40) becomes one contiguous run ("cell"), whatever number of 4 KiB CellPages
// the reference would chain for it. Each type range is padded to TILE_ALIGN slots; padding slots carry id -1 and
// belong to a per-type dead cell that the classify kernel always rejects. Cell slots are consecutive along the
// sphere order, so a 64-sphere chunk is described by the cell slot of its first sphere plus a bit mask of
// "this sphere starts the next cell".
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

#include "lmx_math.h"
#include "lmx_types.h"

namespace lmx {

constexpr int LAYOUT_MAX_TYPES = 8;
constexpr uint32_t LAYOUT_CHUNK = 64;
constexpr uint32_t LAYOUT_TILE_ALIGN = 4096;
constexpr uint32_t LAYOUT_CELL_DEAD = 0x80000000u;

struct CullRec { // host mirror of one Sphere + its CellIndices (culling_system.cpp:23-40, 98-128)
	IV3 cell;
	V3 rel;
	float radius;
	int32_t entity;
	uint8_t type;
	bool big;
};

struct LayoutSphere { float x, y, z, radius; };
struct LayoutCell { int32_t ix, iy, iz; uint32_t meta; }; // meta = type | is_big << 8 | LAYOUT_CELL_DEAD

struct CullLayout {
	std::vector<LayoutSphere> spheres; // [n_padded]
	std::vector<int32_t> ids;          // [n_padded]
	std::vector<uint32_t> slot_cell;   // [n_padded]
	std::vector<LayoutCell> cells;     // [n_cells] (including one dead cell per present type)
	std::vector<uint32_t> chunk_cell;  // [n_padded / 64]
	std::vector<uint64_t> chunk_flags; // [n_padded / 64]
	std::vector<uint32_t> rec_slot;    // [recs] -> sphere slot
	uint32_t ent_start[LAYOUT_MAX_TYPES], ent_end[LAYOUT_MAX_TYPES];
	uint32_t cell_begin[LAYOUT_MAX_TYPES], cell_end[LAYOUT_MAX_TYPES];
	uint32_t n_padded = 0;
};

// CullingSystemImpl::add, culling_system.cpp:131-157 + addToCell :100
inline CullRec make_cull_rec(int32_t entity, uint8_t type, DV3 pos, float radius) {
	CullRec r;
	r.cell = cell_of(pos);
	r.big = is_big_radius(radius);
	r.type = type;
	r.rel = to_v3(sub(pos, cell_origin(r.cell)));
	r.radius = radius;
	r.entity = entity;
	return r;
}

// The kernels' view of a ShiftedFrustum: getRelative re-anchors each plane on a fixed corner point
// (core/geometry.cpp:134-142: NEAR->p0, FAR->p4, LEFT->p1, RIGHT->p0, TOP->p0, BOTTOM->p2).
inline DevFrustum to_dev_frustum(const LmxShiftedFrustum& f) {
	static const int plane_point[6] = {0, 4, 1, 0, 0, 2};
	DevFrustum d = {};
	for (int k = 0; k < 6; ++k) {
		d.nx[k] = f.xs[k];
		d.ny[k] = f.ys[k];
		d.nz[k] = f.zs[k];
		d.d[k] = f.ds[k];
		d.px[k] = f.points[plane_point[k]][0];
		d.py[k] = f.points[plane_point[k]][1];
		d.pz[k] = f.points[plane_point[k]][2];
	}
	d.origin[0] = f.origin[0];
	d.origin[1] = f.origin[1];
	d.origin[2] = f.origin[2];
	return d;
}

// returns false when the set does not fit the 31-bit slot space
inline bool build_cull_layout(const std::vector<CullRec>& recs, CullLayout& out) {
	struct SortItem { uint64_t hi, lo; uint32_t rec; };
	const size_t n = recs.size();
	std::vector<SortItem> items(n);
	for (size_t i = 0; i < n; ++i) {
		const CullRec& r = recs[i];
		items[i].hi = ((uint64_t)r.type << 33) | ((uint64_t)(r.big ? 1 : 0) << 32) | (uint32_t)((uint32_t)r.cell.x ^ 0x80000000u);
		items[i].lo = ((uint64_t)((uint32_t)r.cell.y ^ 0x80000000u) << 32) | (uint32_t)((uint32_t)r.cell.z ^ 0x80000000u);
		items[i].rec = (uint32_t)i;
	}
	std::sort(items.begin(), items.end(), [](const SortItem& a, const SortItem& b) {
		if (a.hi != b.hi) return a.hi < b.hi;
		if (a.lo != b.lo) return a.lo < b.lo;
		return a.rec < b.rec;
	});

	size_t count_by_type[LAYOUT_MAX_TYPES] = {};
	for (size_t i = 0; i < n; ++i) count_by_type[recs[i].type]++;
	size_t n_padded = 0;
	for (int t = 0; t < LAYOUT_MAX_TYPES; ++t) {
		out.ent_start[t] = (uint32_t)n_padded;
		// at least one dead slot per present type, so the dead cell is reachable through the flag chain
		if (count_by_type[t]) n_padded += ((count_by_type[t] + 1 + LAYOUT_TILE_ALIGN - 1) / LAYOUT_TILE_ALIGN) * LAYOUT_TILE_ALIGN;
		out.ent_end[t] = (uint32_t)n_padded;
	}
	if (n_padded > 0x7fffffffull) return false;

	out.spheres.assign(n_padded, LayoutSphere{0.f, 0.f, 0.f, 0.f});
	out.ids.assign(n_padded, -1);
	out.slot_cell.assign(n_padded, 0);
	out.cells.clear();
	out.cells.reserve(n / 8 + 64);
	out.rec_slot.assign(n, 0);

	size_t it = 0;
	for (int t = 0; t < LAYOUT_MAX_TYPES; ++t) {
		out.cell_begin[t] = (uint32_t)out.cells.size();
		if (!count_by_type[t]) {
			out.cell_end[t] = out.cell_begin[t];
			continue;
		}
		size_t slot = out.ent_start[t];
		bool have_prev = false;
		uint64_t prev_hi = 0, prev_lo = 0;
		for (size_t k = 0; k < count_by_type[t]; ++k, ++it, ++slot) {
			const SortItem& si = items[it];
			const CullRec& r = recs[si.rec];
			if (!have_prev || si.hi != prev_hi || si.lo != prev_lo) {
				out.cells.push_back(LayoutCell{r.cell.x, r.cell.y, r.cell.z, (uint32_t)r.type | (r.big ? 0x100u : 0u)});
				prev_hi = si.hi;
				prev_lo = si.lo;
				have_prev = true;
			}
			out.slot_cell[slot] = (uint32_t)out.cells.size() - 1;
			out.spheres[slot] = LayoutSphere{r.rel.x, r.rel.y, r.rel.z, r.radius};
			out.ids[slot] = r.entity;
			out.rec_slot[si.rec] = (uint32_t)slot;
		}
		out.cells.push_back(LayoutCell{0, 0, 0, (uint32_t)t | LAYOUT_CELL_DEAD});
		const uint32_t dead = (uint32_t)out.cells.size() - 1;
		for (; slot < out.ent_end[t]; ++slot) out.slot_cell[slot] = dead;
		out.cell_end[t] = (uint32_t)out.cells.size();
	}

	const size_t n_chunks = n_padded / LAYOUT_CHUNK;
	out.chunk_cell.resize(n_chunks);
	out.chunk_flags.resize(n_chunks);
	for (size_t c = 0; c < n_chunks; ++c) {
		const size_t base = c * LAYOUT_CHUNK;
		out.chunk_cell[c] = out.slot_cell[base];
		uint64_t flags = 0;
		for (uint32_t l = 1; l < LAYOUT_CHUNK; ++l) {
			if (out.slot_cell[base + l] != out.slot_cell[base + l - 1]) flags |= 1ull << l;
		}
		out.chunk_flags[c] = flags;
	}
	out.n_padded = (uint32_t)n_padded;
	return true;
}

} // namespace lmx
