// lmx_cull_layout.h — pure-host construction of the device layout of the culling set (no HIP types), shared by
// lmx_capi.hip (which uploads it) and the CPU-side layout tests.
//
// Spheres are sorted by (type, is_big, Morton code of the cell index): every occupied CellIndices group of the reference
// (src/renderer/culling_system.cpp:23-40) becomes one contiguous run ("cell"), whatever number of 4 KiB CellPages
// the reference would chain for it. Each type range is padded to TILE_ALIGN slots; padding slots carry id -1 and
// belong to a per-type dead cell that the classify kernel always rejects. Cell slots are consecutive along the
// sphere order, so a 64-sphere chunk is described by the cell slot of its first sphere plus a bit mask of
// "this sphere starts the next cell".
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "lmx_math.h"
#include "lmx_types.h"

namespace lmx {

constexpr int LAYOUT_MAX_TYPES = 8;
constexpr uint32_t LAYOUT_CHUNK = 64;
constexpr uint32_t LAYOUT_TILE_ALIGN = 4096;
constexpr uint32_t LAYOUT_CELL_DEAD = 0x80000000u;
// No aligned block of LAYOUT_CELL_BLOCK slots touches more than LAYOUT_MAX_CELLS_PER_BLOCK cell slots (dead cells
// included): the fused cull kernel keeps one tile's cell table in LDS (8 frusta x 240 cells x 16 B = 30 KiB for the
// 1024-slot tile, 4 x 240 cells for the 4096-slot tile). Runs of tiny cells (e.g. "big" spheres, one per cell) are
// spread over more blocks with dead padding, which costs slots but no sphere traffic (dead chunks are never fetched).
constexpr uint32_t LAYOUT_CELL_BLOCK = 1024;
constexpr uint32_t LAYOUT_MAX_CELLS_PER_BLOCK = 240;

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
// Header of one 64-sphere chunk: cell slot of its first sphere + bit l = "sphere l starts the next cell" (one 16-byte scalar load)
struct LayoutChunkHdr { uint32_t cell, pad; uint64_t flags; };
// The visible ids of a type are written to up to LAYOUT_MAX_SHARDS windows ("shards"): the 4096-slot block b of a type reserves
// its output in shard b % n_shards of that type, so the returning atomics of one cull spread over that many counters instead of
// serialising on one address (~88 per microsecond chip-wide on gfx950). A window's capacity is the number of live ids its blocks hold.
constexpr uint32_t LAYOUT_MAX_SHARDS = 64;

struct CullLayout {
	std::vector<LayoutSphere> spheres; // [n_padded]
	std::vector<int32_t> ids;          // [n_padded]
	std::vector<uint32_t> slot_cell;   // [n_padded]
	std::vector<LayoutCell> cells;     // [n_cells] (including one dead cell per present type)
	std::vector<LayoutChunkHdr> hdr;   // [n_padded / 64]
	std::vector<uint32_t> rec_slot;    // [recs] -> sphere slot
	uint32_t ent_start[LAYOUT_MAX_TYPES], ent_end[LAYOUT_MAX_TYPES];
	uint32_t cell_begin[LAYOUT_MAX_TYPES], cell_end[LAYOUT_MAX_TYPES];
	uint32_t n_padded = 0;
	uint32_t n_dead_cells = 0; // dead entries inside `cells`
	std::vector<uint32_t> block_live;  // [n_padded / LAYOUT_TILE_ALIGN] live ids per block (capacities of the output shards)
	// max number of distinct cell slots touched by one tile, for tile sizes 4096 / 2048 / 1024 spheres (fused kernel LDS)
	uint32_t max_tile_cells[3] = {0, 0, 0};
	// Per tile-size variant k (tile = 4096 >> k): the cell keys each tile touches, stored tile-major with a fixed stride
	// tile_cap[k] (= max_tile_cells[k] rounded up to 16), so that a block finds its cells at an address that depends on
	// blockIdx only (no dependent load), plus {first cell slot, number of cells} per tile.
	uint32_t tile_cap[3] = {16, 16, 16};
	std::vector<LayoutCell> tile_cells[3]; // [n_tiles_k * tile_cap[k]], unused tail entries are dead
	std::vector<uint32_t> tile_tab[3];     // [n_tiles_k * 2] = {first_cell, n_cells}
	std::vector<TileBox> tile_box[3];      // [n_tiles_k]: cell-index box of the tile's live cells (k_cull_fused's tile-level early out)
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

// bits of a 21-bit value spread to every third bit (bit i -> bit 3 i)
inline uint64_t spread3(uint64_t x) {
	x &= 0x1fffffull;
	x = (x | x << 32) & 0x1f00000000ffffull;
	x = (x | x << 16) & 0x1f0000ff0000ffull;
	x = (x | x << 8) & 0x100f00f00f00f00full;
	x = (x | x << 4) & 0x10c30c30c30c30c3ull;
	x = (x | x << 2) & 0x1249249249249249ull;
	return x;
}

// ---- small host-side parallel helpers (scene load / layout rebuild of 10^7..10^8 spheres) ------------------------------
inline thread_local unsigned t_layout_thread_cap = 32; // the asynchronous compaction's worker lowers it: a background re-sort should not take the machine
inline unsigned layout_threads(size_t n) {
	if (n < (1u << 18)) return 1;
	const unsigned hw = std::thread::hardware_concurrency();
	return std::max(1u, std::min(hw ? hw : 1u, t_layout_thread_cap));
}
template <typename Fn> inline void parallel_ranges(size_t n, Fn fn) { // fn(begin, end) over a partition of [0, n)
	const unsigned t = layout_threads(n);
	if (t <= 1) {
		fn((size_t)0, n);
		return;
	}
	std::vector<std::thread> th;
	th.reserve(t);
	for (unsigned k = 0; k < t; ++k) th.emplace_back([=] { fn(n * k / t, n * (k + 1) / t); });
	for (std::thread& x : th) x.join();
}
// Sample sort: splitters from a sorted sample cut the keys into 4 x threads buckets, items are scattered to their buckets and the
// buckets sorted independently - every pass runs on all threads. (The merge sort this replaces ended in log2(threads) rounds of
// ever fewer, ever longer merges - the last one a single thread over the whole array - and was the long pole of the layout build.)
// `less` must be a strict total order (the layout's keys end in the record index), so the result does not depend on the algorithm.
template <typename T, typename Less> inline void parallel_sort(std::vector<T>& v, Less less) {
	const size_t n = v.size();
	const unsigned t = layout_threads(n);
	if (t <= 1) {
		std::sort(v.begin(), v.end(), less);
		return;
	}
	const unsigned n_buckets = t * 4, oversample = 32;
	std::vector<T> sample(n_buckets * oversample);
	for (size_t i = 0; i < sample.size(); ++i) sample[i] = v[(n / sample.size()) * i + (n / sample.size()) / 2];
	std::sort(sample.begin(), sample.end(), less);
	std::vector<T> split(n_buckets - 1);
	for (unsigned k = 0; k + 1 < n_buckets; ++k) split[k] = sample[(k + 1) * oversample];
	auto run = [&](auto fn) { // fn(thread index) on t threads
		std::vector<std::thread> th;
		th.reserve(t);
		for (unsigned k = 0; k < t; ++k) th.emplace_back([=] { fn(k); });
		for (std::thread& x : th) x.join();
	};
	std::vector<uint16_t> bucket(n);
	std::vector<size_t> count((size_t)t * n_buckets, 0); // [thread][bucket]
	run([&](unsigned k) {
		size_t* c = &count[(size_t)k * n_buckets];
		for (size_t i = n * k / t; i < n * (k + 1) / t; ++i) {
			const unsigned bk = (unsigned)(std::upper_bound(split.begin(), split.end(), v[i], less) - split.begin());
			bucket[i] = (uint16_t)bk;
			++c[bk];
		}
	});
	std::vector<size_t> bucket_begin(n_buckets + 1, 0);
	{
		size_t at = 0;
		for (unsigned bk = 0; bk < n_buckets; ++bk) { // bucket-major, thread-minor: every (thread, bucket) pair gets its own range
			bucket_begin[bk] = at;
			for (unsigned k = 0; k < t; ++k) {
				const size_t c = count[(size_t)k * n_buckets + bk];
				count[(size_t)k * n_buckets + bk] = at;
				at += c;
			}
		}
		bucket_begin[n_buckets] = at;
	}
	std::vector<T> tmp(n);
	run([&](unsigned k) {
		size_t* at = &count[(size_t)k * n_buckets];
		for (size_t i = n * k / t; i < n * (k + 1) / t; ++i) tmp[at[bucket[i]]++] = v[i];
	});
	{ std::vector<uint16_t>().swap(bucket); }
	std::atomic<unsigned> next{0};
	run([&](unsigned) {
		for (unsigned bk = next.fetch_add(1); bk < n_buckets; bk = next.fetch_add(1)) std::sort(tmp.begin() + bucket_begin[bk], tmp.begin() + bucket_begin[bk + 1], less);
	});
	v.swap(tmp);
}

// returns false when the set does not fit the 31-bit slot space
inline bool build_cull_layout(const std::vector<CullRec>& recs, CullLayout& out) {
	struct SortItem { uint64_t hi, lo; uint32_t rec; };
	const size_t n = recs.size();
	std::vector<SortItem> items(n);
	parallel_ranges(n, [&](size_t b, size_t e) {
		for (size_t i = b; i < e; ++i) {
			const CullRec& r = recs[i];
			// cells in Morton (Z-curve) order of their sign-biased indices: a tile of consecutive spheres then covers a compact
			// block of cells, which is what makes the tile-level tests of k_cull_tile (tile_status) effective. 96 bits of
			// Morton code = the interleaved high 11 bits of x, y, z (33 bits) above their interleaved low 21 bits (63 bits).
			const uint32_t bx = (uint32_t)r.cell.x ^ 0x80000000u, by = (uint32_t)r.cell.y ^ 0x80000000u, bz = (uint32_t)r.cell.z ^ 0x80000000u;
			const uint64_t m_hi = (spread3(bx >> 21) << 2) | (spread3(by >> 21) << 1) | spread3(bz >> 21);
			const uint64_t m_lo = (spread3(bx & 0x1fffffu) << 2) | (spread3(by & 0x1fffffu) << 1) | spread3(bz & 0x1fffffu);
			items[i].hi = ((uint64_t)r.type << 35) | ((uint64_t)(r.big ? 1 : 0) << 34) | m_hi;
			items[i].lo = m_lo;
			items[i].rec = (uint32_t)i;
		}
	});
	parallel_sort(items, [](const SortItem& a, const SortItem& b) {
		if (a.hi != b.hi) return a.hi < b.hi;
		if (a.lo != b.lo) return a.lo < b.lo;
		return a.rec < b.rec;
	});

	size_t count_by_type[LAYOUT_MAX_TYPES] = {};
	for (size_t i = 0; i < n; ++i) count_by_type[recs[i].type]++;
	// Placement. The padding rules are a serial state machine, but its state only changes at CELL boundaries: the spheres are cut
	// into cells in parallel, the state machine walks the ~n / 10 cells (not the n spheres), and the spheres / ids / cell slots are
	// written in parallel from the cells' start slots - the one random pass over the 10^7..10^8 records (read in sorted order, slot
	// written back in record order) runs on all threads.
	std::vector<uint8_t> starts_cell(n); // the i-th sphere in sorted order is the first of a (type, is_big, cell) group
	parallel_ranges(n, [&](size_t b, size_t e) {
		for (size_t i = b; i < e; ++i) starts_cell[i] = i == 0 || items[i].hi != items[i - 1].hi || items[i].lo != items[i - 1].lo; // (type is part of `hi`)
	});
	std::vector<size_t> group_first; // sorted index of every group's first sphere, + n
	{
		const unsigned T = layout_threads(n);
		std::vector<size_t> cnt(T + 1, 0);
		{
			std::vector<std::thread> th;
			for (unsigned k = 0; k < T; ++k) th.emplace_back([&, k] {
				size_t c = 0;
				for (size_t i = n * k / T; i < n * (k + 1) / T; ++i) c += starts_cell[i];
				cnt[k + 1] = c;
			});
			for (std::thread& x : th) x.join();
		}
		for (unsigned k = 0; k < T; ++k) cnt[k + 1] += cnt[k];
		group_first.resize(cnt[T] + 1);
		{
			std::vector<std::thread> th;
			for (unsigned k = 0; k < T; ++k) th.emplace_back([&, k] {
				size_t at = cnt[k];
				for (size_t i = n * k / T; i < n * (k + 1) / T; ++i) if (starts_cell[i]) group_first[at++] = i;
			});
			for (std::thread& x : th) x.join();
		}
		group_first.back() = n;
	}
	{ std::vector<uint8_t>().swap(starts_cell); }
	const size_t n_groups = group_first.size() - 1;
	std::vector<uint32_t> group_slot(n_groups), group_cell(n_groups); // first sphere slot / index into out.cells of every group
	struct Pad { uint32_t begin, end, dead_cell; };
	std::vector<Pad> pads;
	out.cells.clear();
	out.cells.reserve(n_groups + n_groups / 64 + 2 * LAYOUT_MAX_TYPES + 64);
	uint64_t pos = 0;
	auto round_up = [](uint64_t v, uint64_t a) { return (v + a - 1) / a * a; };
	size_t g = 0;
	for (int t = 0; t < LAYOUT_MAX_TYPES; ++t) {
		out.ent_start[t] = (uint32_t)pos;
		out.cell_begin[t] = (uint32_t)out.cells.size();
		if (!count_by_type[t]) {
			out.ent_end[t] = out.ent_start[t];
			out.cell_end[t] = out.cell_begin[t];
			continue;
		}
		uint32_t block_cells = 0; // distinct cell slots overlapping the current LAYOUT_CELL_BLOCK-slot block
		for (; g < n_groups && recs[items[group_first[g]].rec].type == (uint8_t)t; ++g) {
			const CullRec& r = recs[items[group_first[g]].rec];
			const uint64_t size = group_first[g + 1] - group_first[g];
			if (pos % LAYOUT_CELL_BLOCK == 0) block_cells = 0;
			if (block_cells + 2 > LAYOUT_MAX_CELLS_PER_BLOCK) {
				// this block already touches its quota of cells (one is kept for the dead cell): close it with dead slots
				out.cells.push_back(LayoutCell{0, 0, 0, (uint32_t)t | LAYOUT_CELL_DEAD});
				const uint64_t to = round_up(pos, LAYOUT_CELL_BLOCK);
				pads.push_back(Pad{(uint32_t)pos, (uint32_t)to, (uint32_t)out.cells.size() - 1});
				pos = to;
				block_cells = 0;
			}
			out.cells.push_back(LayoutCell{r.cell.x, r.cell.y, r.cell.z, (uint32_t)r.type | (r.big ? 0x100u : 0u)});
			++block_cells;
			group_slot[g] = (uint32_t)pos;
			group_cell[g] = (uint32_t)out.cells.size() - 1;
			// a cell that continues into the next block is the one cell that block has seen so far
			if (round_up(pos + 1, LAYOUT_CELL_BLOCK) <= pos + size - 1) block_cells = 1;
			pos += size;
			if (pos > 0x7fffffffull) return false;
		}
		// at least one dead slot per type, then pad the type range to the largest tile
		out.cells.push_back(LayoutCell{0, 0, 0, (uint32_t)t | LAYOUT_CELL_DEAD});
		const uint64_t to = round_up(pos + 1, LAYOUT_TILE_ALIGN);
		pads.push_back(Pad{(uint32_t)pos, (uint32_t)to, (uint32_t)out.cells.size() - 1});
		pos = to;
		out.ent_end[t] = (uint32_t)pos;
		out.cell_end[t] = (uint32_t)out.cells.size();
		if (pos > 0x7fffffffull) return false;
	}
	const size_t n_padded = (size_t)pos;
	out.spheres.resize(n_padded);
	out.ids.resize(n_padded);
	out.slot_cell.resize(n_padded);
	out.rec_slot.assign(n, 0);
	for (const Pad& p : pads) {
		for (uint32_t s = p.begin; s < p.end; ++s) {
			out.spheres[s] = LayoutSphere{0.f, 0.f, 0.f, 0.f};
			out.ids[s] = -1;
			out.slot_cell[s] = p.dead_cell;
		}
	}
	parallel_ranges(n_groups, [&](size_t gb, size_t ge) {
		for (size_t k = gb; k < ge; ++k) {
			uint32_t slot = group_slot[k];
			const uint32_t cell = group_cell[k];
			for (size_t i = group_first[k]; i < group_first[k + 1]; ++i, ++slot) {
				const CullRec& r = recs[items[i].rec];
				out.spheres[slot] = LayoutSphere{r.rel.x, r.rel.y, r.rel.z, r.radius};
				out.ids[slot] = r.entity;
				out.slot_cell[slot] = cell;
				out.rec_slot[items[i].rec] = slot;
			}
		}
	});
	{ std::vector<SortItem>().swap(items); std::vector<size_t>().swap(group_first); }

	const size_t n_chunks = n_padded / LAYOUT_CHUNK;
	out.hdr.resize(n_chunks);
	parallel_ranges(n_chunks, [&](size_t cb, size_t ce) {
		for (size_t c = cb; c < ce; ++c) {
			const size_t base = c * LAYOUT_CHUNK;
			uint64_t flags = 0;
			for (uint32_t l = 1; l < LAYOUT_CHUNK; ++l) {
				if (out.slot_cell[base + l] != out.slot_cell[base + l - 1]) flags |= 1ull << l;
			}
			out.hdr[c] = LayoutChunkHdr{out.slot_cell[base], 0u, flags};
		}
	});
	out.n_padded = (uint32_t)n_padded;
	out.n_dead_cells = 0;
	for (const LayoutCell& c : out.cells) out.n_dead_cells += (c.meta & LAYOUT_CELL_DEAD) ? 1u : 0u;
	out.block_live.assign(n_padded / LAYOUT_TILE_ALIGN, 0u);
	parallel_ranges(out.block_live.size(), [&](size_t bb, size_t be) {
		for (size_t b = bb; b < be; ++b) {
			uint32_t live = 0;
			for (uint32_t l = 0; l < LAYOUT_TILE_ALIGN; ++l) live += out.ids[b * LAYOUT_TILE_ALIGN + l] >= 0 ? 1u : 0u;
			out.block_live[b] = live;
		}
	});
	for (int k = 0; k < 3; ++k) {
		const size_t tile = (size_t)LAYOUT_TILE_ALIGN >> k;
		const size_t n_tiles = n_padded / tile;
		uint32_t m = 0;
		for (size_t b = 0; b + tile <= n_padded; b += tile) {
			const uint32_t c = out.slot_cell[b + tile - 1] - out.slot_cell[b] + 1; // cell slots are consecutive along the sphere order
			if (c > m) m = c;
		}
		out.max_tile_cells[k] = m;
		const uint32_t cap = ((m > 0 ? m : 1u) + 15u) / 16u * 16u;
		out.tile_cap[k] = cap;
		out.tile_cells[k].assign(n_tiles * cap, LayoutCell{0, 0, 0, LAYOUT_CELL_DEAD});
		out.tile_tab[k].assign(n_tiles * 2, 0u);
		out.tile_box[k].assign(n_tiles, TileBox{{0, 0, 0}, {0, 0, 0}, TILE_EMPTY, 0});
		parallel_ranges(n_tiles, [&, k, tile, cap](size_t tb, size_t te) {
			for (size_t ti = tb; ti < te; ++ti) {
				const uint32_t first = out.slot_cell[ti * tile];
				const uint32_t cnt = out.slot_cell[ti * tile + tile - 1] - first + 1;
				out.tile_tab[k][2 * ti] = first;
				out.tile_tab[k][2 * ti + 1] = cnt;
				TileBox box = {{INT32_MAX, INT32_MAX, INT32_MAX}, {INT32_MIN, INT32_MIN, INT32_MIN}, TILE_EMPTY, 0};
				bool dense = true; // a tile without padding slots: an accepted tile is then a straight copy of its ids
				for (size_t e = ti * tile; e < (ti + 1) * tile && dense; ++e) dense = out.ids[e] >= 0;
				if (dense) box.flags |= TILE_DENSE;
				for (uint32_t j = 0; j < cnt; ++j) {
					const LayoutCell& c = out.cells[first + j];
					out.tile_cells[k][ti * cap + j] = c;
					if (c.meta & LAYOUT_CELL_DEAD) continue;
					box.flags &= ~(uint32_t)TILE_EMPTY;
					if (c.meta & 0x100u) box.flags |= TILE_HAS_BIG;
					const int32_t idx[3] = {c.ix, c.iy, c.iz};
					for (int a = 0; a < 3; ++a) {
						box.lo[a] = std::min(box.lo[a], idx[a]);
						box.hi[a] = std::max(box.hi[a], idx[a]);
					}
				}
				out.tile_box[k][ti] = box;
			}
		});
	}
	return true;
}

} // namespace lmx
