// cull_kernels.hip — frustum culling of CullingSystem spheres on gfx950 (wave64).
//
// Replaces CullingSystemImpl::cullInternal / doCulling (src/renderer/culling_system.cpp:262-369) and the per-cell
// ShiftedFrustum tests (src/core/geometry.cpp:99-178). Two kernels per cull:
//
//   k_cull_classify   1 thread per occupied (cell,type,is_big) group x frustum: fp64 origin shift, the
//                     contains/intersects AABB tests, and the getRelative() offset -> 16 B per cell per frustum.
//   k_cull_spheres    1 wave per 64-sphere chunk: chunk header (scalar load) -> per-lane cell slot via the
//                     "new cell" bit mask + mbcnt -> 16-B cell-info gather (L2-resident) -> whole-wave early out
//                     for rejected cells -> 8-plane test -> wave64 ballot compaction into an LDS staging list ->
//                     ONE global atomic per (tile, frustum) and a coalesced flush of the visible ids.
//
// Built with -ffp-contract=off: the plane arithmetic must round exactly like the reference's scalar float4.
#include <cstdlib>

#include "lmx_kernels.h"

namespace lmx {

namespace {

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t mbcnt64(uint64_t mask) {
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

template <int F>
__global__ __launch_bounds__(256) void k_cull_classify(const CellKey* __restrict__ cells, uint32_t cell_begin, uint32_t n,
	FrustaArg fr, float4* __restrict__ cellinfo, uint32_t cell_stride, uint32_t* __restrict__ counts) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (blockIdx.x == 0 && threadIdx.x < MAX_FRUSTA * MAX_TYPES) counts[threadIdx.x] = 0;
	if (i >= n) return;
	const uint32_t c = cell_begin + i;
	const CellKey key = cells[c];
	const bool dead = (key.meta & CELL_DEAD) != 0;
	const bool big = (key.meta & 0x100u) != 0;
#pragma unroll
	for (int f = 0; f < F; ++f) {
		V3 off = V3{0.f, 0.f, 0.f};
		uint32_t cls = CELL_REJECT;
		if (!dead) cls = classify_cell(fr.f[f], IV3{key.ix, key.iy, key.iz}, big, &off);
		cellinfo[(size_t)f * cell_stride + c] = make_float4(off.x, off.y, off.z, __uint_as_float(cls));
	}
}

// WAVES waves per block, CHW chunks per wave -> TILE = WAVES * CHW * 64 spheres per block.
template <int F, int WAVES, int CHW>
__global__ __launch_bounds__(WAVES * 64) void k_cull_spheres(const float4* __restrict__ spheres, const int32_t* __restrict__ ids,
	const uint32_t* __restrict__ chunk_cell, const uint64_t* __restrict__ chunk_flags, const float4* __restrict__ cellinfo,
	uint32_t cell_stride, FrustaArg fr, TypeTable tt, uint32_t ent_begin, int32_t* __restrict__ out_ids, uint32_t out_stride,
	uint32_t* __restrict__ counts) {
	constexpr int TILE = WAVES * CHW * 64;
	__shared__ int32_t s_buf[F * TILE];
	__shared__ uint32_t s_cnt[F];
	__shared__ uint32_t s_base[F];

	const uint32_t lane = lane_id();
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t tile_ent = ent_begin + blockIdx.x * (uint32_t)TILE; // first sphere slot of this tile

	if (threadIdx.x < F) s_cnt[threadIdx.x] = 0;
	__syncthreads();

	// type of this tile (type ranges are TILE_ALIGN-aligned, so a tile never straddles two types)
	uint32_t type = 0;
#pragma unroll
	for (int t = 0; t < MAX_TYPES; ++t) {
		if (tile_ent >= tt.ent_start[t] && tile_ent < tt.ent_end[t]) type = t;
	}

	const uint32_t chunk0 = (tile_ent >> 6) + wave * CHW; // this wave's CHW consecutive chunks
	const uint64_t le_mask = (~0ull >> (63u - lane)) & ~1ull; // bits 1..lane

	// phase 1: chunk headers (wave-uniform -> scalar loads), per-lane cell slot, cell-info gather
	uint32_t cell[CHW];
#pragma unroll
	for (int i = 0; i < CHW; ++i) {
		const uint32_t base_cell = chunk_cell[chunk0 + i];
		const uint64_t flags = chunk_flags[chunk0 + i];
		cell[i] = base_cell + (uint32_t)__popcll(flags & le_mask);
	}
	float4 info[F][CHW];
#pragma unroll
	for (int f = 0; f < F; ++f) {
#pragma unroll
		for (int i = 0; i < CHW; ++i) info[f][i] = cellinfo[(size_t)f * cell_stride + cell[i]];
	}

	// phase 2: which chunks need their spheres / ids at all (whole-cell reject costs no sphere traffic)
	bool need_id[CHW], need_sphere[CHW];
#pragma unroll
	for (int i = 0; i < CHW; ++i) {
		bool any_live = false, any_test = false;
#pragma unroll
		for (int f = 0; f < F; ++f) {
			const uint32_t cls = __float_as_uint(info[f][i].w);
			any_live |= cls != CELL_REJECT;
			any_test |= cls == CELL_TEST;
		}
		need_id[i] = __ballot(any_live) != 0;     // wave-uniform
		need_sphere[i] = __ballot(any_test) != 0; // wave-uniform
	}
	int32_t id[CHW];
	float4 sp[CHW];
#pragma unroll
	for (int i = 0; i < CHW; ++i) {
		const uint32_t e = ((chunk0 + i) << 6) + lane;
		id[i] = -1;
		sp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
		if (need_id[i]) id[i] = ids[e];
		if (need_sphere[i]) sp[i] = spheres[e];
	}

	// phase 3: plane tests + wave ballot compaction into the LDS staging lists
#pragma unroll
	for (int i = 0; i < CHW; ++i) {
		if (!need_id[i]) continue;
#pragma unroll
		for (int f = 0; f < F; ++f) {
			const uint32_t cls = __float_as_uint(info[f][i].w);
			bool vis = cls == CELL_ACCEPT;
			if (cls == CELL_TEST) {
				vis = sphere_visible(fr.f[f], V3{info[f][i].x, info[f][i].y, info[f][i].z}, sp[i].x, sp[i].y, sp[i].z, sp[i].w);
			}
			vis = vis && id[i] >= 0;
			const uint64_t mask = __ballot(vis);
			if (mask != 0) {
				uint32_t base = 0;
				if (lane == 0) base = atomicAdd(&s_cnt[f], (uint32_t)__popcll(mask));
				base = __builtin_amdgcn_readfirstlane(base);
				if (vis) s_buf[f * TILE + base + mbcnt64(mask)] = id[i];
			}
		}
	}
	__syncthreads();

	// one global atomic per (tile, frustum) with anything visible, then a coalesced flush
	if (threadIdx.x < F) {
		const uint32_t c = s_cnt[threadIdx.x];
		s_base[threadIdx.x] = c ? atomicAdd(&counts[threadIdx.x * MAX_TYPES + type], c) : 0u;
	}
	__syncthreads();
#pragma unroll
	for (int f = 0; f < F; ++f) {
		const uint32_t c = s_cnt[f];
		int32_t* dst = out_ids + (size_t)f * out_stride + tt.out_start[type] + s_base[f];
		for (uint32_t k = threadIdx.x; k < c; k += WAVES * 64) dst[k] = s_buf[f * TILE + k];
	}
}

// ---- fused variant: per-tile classification in LDS -----------------------------------------------------------
// One launch per cull. Cell slots are consecutive along the sphere order, so the cells a tile touches are the range
// [chunk_cell[first chunk], cell of the tile's last sphere]; the block classifies exactly those cells (one thread per
// cell x frustum, same arithmetic as k_cull_classify) into LDS, votes whether anything in the tile survives, and only
// then touches spheres/ids. Compared with classify + spheres this removes a kernel boundary, the 16 B/cell/frustum
// round trip through global memory and one dependent gather per chunk. Cells straddling two tiles are classified by
// both (harmless). The host guarantees cell_cap >= cells per tile (layout max) and falls back to the two-kernel path
// when the LDS budget would not fit.
template <int F, int WAVES, int CHW>
__global__ __launch_bounds__(WAVES * 64) void k_cull_fused(const float4* __restrict__ spheres, const int32_t* __restrict__ ids,
	const uint32_t* __restrict__ chunk_cell, const uint64_t* __restrict__ chunk_flags, const CellKey* __restrict__ tile_cells,
	const uint32_t* __restrict__ tile_tab, const TileBox* __restrict__ tile_box, FrustaArg fr, TypeTable tt, uint32_t ent_begin, uint32_t cell_cap,
	int32_t* __restrict__ out_ids, uint32_t out_stride, uint32_t* __restrict__ counts, uint32_t* __restrict__ counts_next) {
	constexpr int TILE = WAVES * CHW * 64;
	constexpr int NCH = WAVES * CHW;
	extern __shared__ float4 s_dyn[]; // [F * cell_cap] cell info | [F * TILE] staged ids | [F] counts | [F] bases
	float4* s_info = s_dyn;
	int32_t* s_buf = reinterpret_cast<int32_t*>(s_dyn + (size_t)F * cell_cap);
	uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_buf + F * TILE);
	uint32_t* s_base = s_cnt + F;

	const uint32_t lane = lane_id();
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t tile_ent = ent_begin + blockIdx.x * (uint32_t)TILE;
	const uint32_t tile_chunk = tile_ent >> 6;

	// the counters of the NEXT cull on this view are cleared here (ping-pong), so no cull needs a separate memset
	if (counts_next != nullptr && blockIdx.x == 0 && threadIdx.x < MAX_FRUSTA * MAX_TYPES) counts_next[threadIdx.x] = 0;
	if (threadIdx.x < F) s_cnt[threadIdx.x] = 0;

	// Everything the block needs first sits at addresses that depend on blockIdx only: {first cell, n cells} of the tile
	// (scalar load) and the tile's cell keys (tile-major, stride cell_cap) - no dependent round trip before phase A.
	const uint32_t chunk0 = tile_chunk + wave * CHW;
	const uint32_t tile_index = tile_ent / (uint32_t)TILE;
	// tile-level early out: the box of the tile's cell indices against every frustum of the pass (block-uniform; conservative,
	// see tile_rejected). Most tiles of a large scene end here without touching their ~400 cell keys.
	{
		// Low-coverage variant (2048-sphere tiles, chosen when the frustum covers little of the scene): ONE wave evaluates the
		// test - it is block-uniform, and four copies of ~100 VALU instructions per tile were a third of the kernel's VALU time
		// when 95 % of the tiles end here - the others wait at the barrier (17.0 -> 14.6 us). Elsewhere few tiles are
		// rejected and the barrier would only add latency (+3 us with everything visible): every wave evaluates it.
		if constexpr (F == 1 && WAVES == 4) {
			__shared__ uint32_t s_tile_rejected;
			if (wave == 0) {
				const bool rejected = tile_rejected(fr.f[0], tile_box[tile_index]);
				if (lane == 0) s_tile_rejected = rejected ? 1u : 0u;
			}
			__syncthreads();
			if (s_tile_rejected) return;
		} else {
			const TileBox box = tile_box[tile_index];
			bool all_rejected = true;
#pragma unroll
			for (int f = 0; f < F; ++f) all_rejected = all_rejected && tile_rejected(fr.f[f], box);
			if (all_rejected) return;
		}
	}
	const uint32_t first_cell = tile_tab[2 * tile_index];
	const uint32_t n_cells = tile_tab[2 * tile_index + 1];
	const CellKey* keys = tile_cells + (size_t)tile_index * cell_cap;

	// phase A: classify the tile's cells into LDS
	bool live = false;
	for (uint32_t t = threadIdx.x; t < cell_cap; t += WAVES * 64) {
		const CellKey key = keys[t]; // issued before n_cells is known; the tail of the slice holds dead keys
		if (t < n_cells) {
			const bool dead = (key.meta & CELL_DEAD) != 0;
			const bool big = (key.meta & 0x100u) != 0;
#pragma unroll
			for (int f = 0; f < F; ++f) {
				V3 off = V3{0.f, 0.f, 0.f};
				uint32_t cls = CELL_REJECT;
				if (!dead) cls = classify_cell(fr.f[f], IV3{key.ix, key.iy, key.iz}, big, &off);
				s_info[f * cell_cap + t] = make_float4(off.x, off.y, off.z, __uint_as_float(cls));
				live |= cls != CELL_REJECT;
			}
		}
	}
	if (!__syncthreads_or(live ? 1 : 0)) return; // nothing in this tile survives the per-cell tests

	uint32_t type = 0;
#pragma unroll
	for (int t = 0; t < MAX_TYPES; ++t) {
		if (tile_ent >= tt.ent_start[t] && tile_ent < tt.ent_end[t]) type = t;
	}

	// phase B: per-lane cell, class from LDS, spheres / ids only for chunks that need them. The wave's CHW chunks are
	// processed in groups of GRP so that at most GRP chunks' worth of cell info / spheres / ids are live in registers
	// (VGPR count decides how many tiles a CU keeps in flight, and this kernel is latency-bound).
	const uint64_t le_mask = (~0ull >> (63u - lane)) & ~1ull; // bits 1..lane
	constexpr int GRP = CHW > 4 ? 4 : CHW;
#pragma unroll
	for (int g = 0; g < CHW; g += GRP) {
		__builtin_amdgcn_sched_barrier(0); // keep the groups' loads from being hoisted over each other (register peak)
		float4 info[F][GRP];
		bool need_id[GRP], need_sphere[GRP];
		uint32_t base_cell[GRP];
		uint64_t flags[GRP];
#pragma unroll
		for (int i = 0; i < GRP; ++i) {
			base_cell[i] = chunk_cell[chunk0 + g + i];
			flags[i] = chunk_flags[chunk0 + g + i];
		}
#pragma unroll
		for (int i = 0; i < GRP; ++i) {
			const uint32_t local = base_cell[i] + (uint32_t)__popcll(flags[i] & le_mask) - first_cell;
			bool any_live = false, any_test = false;
#pragma unroll
			for (int f = 0; f < F; ++f) {
				info[f][i] = s_info[f * cell_cap + local];
				const uint32_t cls = __float_as_uint(info[f][i].w);
				any_live |= cls != CELL_REJECT;
				any_test |= cls == CELL_TEST;
			}
			need_id[i] = __ballot(any_live) != 0;
			need_sphere[i] = __ballot(any_test) != 0;
		}
		int32_t id[GRP];
		float4 sp[GRP];
#pragma unroll
		for (int i = 0; i < GRP; ++i) {
			const uint32_t e = ((chunk0 + g + i) << 6) + lane;
			id[i] = -1;
			sp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
			if (need_id[i]) id[i] = ids[e];
			if (need_sphere[i]) sp[i] = spheres[e];
		}
#pragma unroll
		for (int i = 0; i < GRP; ++i) {
			if (!need_id[i]) continue;
#pragma unroll
			for (int f = 0; f < F; ++f) {
				const uint32_t cls = __float_as_uint(info[f][i].w);
				bool vis = cls == CELL_ACCEPT;
				if (cls == CELL_TEST) {
					vis = sphere_visible(fr.f[f], V3{info[f][i].x, info[f][i].y, info[f][i].z}, sp[i].x, sp[i].y, sp[i].z, sp[i].w);
				}
				vis = vis && id[i] >= 0;
				const uint64_t mask = __ballot(vis);
				if (mask != 0) {
					uint32_t base = 0;
					if (lane == 0) base = atomicAdd(&s_cnt[f], (uint32_t)__popcll(mask));
					base = __builtin_amdgcn_readfirstlane(base);
					if (vis) s_buf[f * TILE + base + mbcnt64(mask)] = id[i];
				}
			}
		}
	}
	__syncthreads();
	if (threadIdx.x < F) {
		const uint32_t c = s_cnt[threadIdx.x];
		s_base[threadIdx.x] = c ? atomicAdd(&counts[threadIdx.x * MAX_TYPES + type], c) : 0u;
	}
	__syncthreads();
#pragma unroll
	for (int f = 0; f < F; ++f) {
		const uint32_t c = s_cnt[f];
		int32_t* dst = out_ids + (size_t)f * out_stride + tt.out_start[type] + s_base[f];
		for (uint32_t k = threadIdx.x; k < c; k += WAVES * 64) dst[k] = s_buf[f * TILE + k];
	}
}

// ---- dynamic set ------------------------------------------------------------------------------------------------
// One thread per unsorted entity: cell index, is_big and the cell-relative fp32 position are derived from the fp64 world
// position exactly like CullingSystemImpl::add / set would (culling_system.cpp:26-30,100,140), the cell is classified per
// lane (no sharing between lanes: the set is not sorted) and the sphere is tested. ~350 VALU ops per entity and frustum,
// 36 B per entity: the VALU-heavier, bandwidth-lighter sibling of the sorted path, used only for entities that move.
constexpr int DYN_THREADS = 256;

// A block handles TILE entities in TILE / 256 batches and stages the visible ids of the whole tile in LDS, so that the
// global counter sees one atomic per (tile, frustum): the dynamic set is unsorted, visible entities are spread over every
// block, and a 256-entity granule would put ~4 k returning atomics on one address per million entities.
template <int TILE>
__global__ __launch_bounds__(DYN_THREADS) void k_cull_dynamic(const double* __restrict__ px, const double* __restrict__ py,
	const double* __restrict__ pz, const float* __restrict__ radius, const int32_t* __restrict__ ids, FrustaArg fr, int n_frusta,
	TypeTable dyn_tt, uint32_t slot_begin, int32_t* __restrict__ out_ids, uint32_t out_stride, uint32_t* __restrict__ counts) {
	extern __shared__ int32_t s_stage[]; // [n_frusta][TILE] staged ids | [MAX_FRUSTA] counts | [MAX_FRUSTA] bases
	uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_stage + n_frusta * TILE);
	uint32_t* s_base = s_cnt + MAX_FRUSTA;
	const uint32_t lane = lane_id();
	const uint32_t block_slot = slot_begin + blockIdx.x * (uint32_t)TILE;
	uint32_t type = 0;
#pragma unroll
	for (int t = 0; t < MAX_TYPES; ++t) {
		if (block_slot >= dyn_tt.ent_start[t] && block_slot < dyn_tt.ent_end[t]) type = t;
	}
	if (threadIdx.x < MAX_FRUSTA) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	for (uint32_t b = 0; b < TILE / DYN_THREADS; ++b) {
		const uint32_t slot = block_slot + b * DYN_THREADS + threadIdx.x;
		const int32_t id = ids[slot];
		const DV3 pos = DV3{px[slot], py[slot], pz[slot]};
		const float r = radius[slot];
		const IV3 idx = cell_of(pos);
		const bool big = is_big_radius(r);
		const V3 rel = to_v3(sub(pos, cell_origin(idx))); // addToCell, culling_system.cpp:100
		for (int f = 0; f < n_frusta; ++f) {
			V3 off;
			const uint32_t cls = classify_cell(fr.f[f], idx, big, &off);
			bool vis = cls == CELL_ACCEPT;
			if (cls == CELL_TEST) vis = sphere_visible(fr.f[f], off, rel.x, rel.y, rel.z, r);
			vis = vis && id >= 0;
			const uint64_t mask = __ballot(vis);
			if (mask != 0) {
				uint32_t base = 0;
				if (lane == 0) base = atomicAdd(&s_cnt[f], (uint32_t)__popcll(mask));
				base = __builtin_amdgcn_readfirstlane(base);
				if (vis) s_stage[f * TILE + base + mbcnt64(mask)] = id;
			}
		}
	}
	__syncthreads();
	if ((int)threadIdx.x < n_frusta) {
		const uint32_t c = s_cnt[threadIdx.x];
		s_base[threadIdx.x] = c ? atomicAdd(&counts[threadIdx.x * MAX_TYPES + type], c) : 0u;
	}
	__syncthreads();
	for (int f = 0; f < n_frusta; ++f) {
		const uint32_t c = s_cnt[f];
		int32_t* dst = out_ids + (size_t)f * out_stride + dyn_tt.out_start[type] + s_base[f];
		for (uint32_t k = threadIdx.x; k < c; k += DYN_THREADS) dst[k] = s_stage[f * TILE + k];
	}
}

__global__ __launch_bounds__(256) void k_patch_spheres(float4* __restrict__ spheres, const uint32_t* __restrict__ slot,
	const float4* __restrict__ value, uint32_t n) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) spheres[slot[i]] = value[i];
}

template <int F>
hipError_t classify_f(hipStream_t s, const CullDeviceView& v, uint32_t cell_begin, uint32_t n, const FrustaArg& fr, float4* cellinfo,
	uint32_t cell_stride, uint32_t* counts) {
	const uint32_t blocks = n ? (n + 255u) / 256u : 1u; // always >= 1 block: block 0 zeroes the counters
	hipLaunchKernelGGL((k_cull_classify<F>), dim3(blocks), dim3(256), 0, s, v.cells, cell_begin, n, fr, cellinfo, cell_stride, counts);
	return hipGetLastError();
}

template <int F, int WAVES, int CHW>
hipError_t spheres_f(hipStream_t s, const CullDeviceView& v, uint32_t ent_begin, uint32_t ent_end, const TypeTable& tt,
	const FrustaArg& fr, const float4* cellinfo, uint32_t cell_stride, int32_t* out_ids, uint32_t out_stride, uint32_t* counts) {
	constexpr uint32_t TILE = WAVES * CHW * 64;
	static_assert(TILE_ALIGN % TILE == 0, "tiles must not straddle type ranges");
	const uint32_t tiles = (ent_end - ent_begin) / TILE;
	if (!tiles) return hipSuccess;
	hipLaunchKernelGGL((k_cull_spheres<F, WAVES, CHW>), dim3(tiles), dim3(WAVES * 64), 0, s, v.spheres, v.ids, v.chunk_cell,
		v.chunk_flags, cellinfo, cell_stride, fr, tt, ent_begin, out_ids, out_stride, counts);
	return hipGetLastError();
}

template <int F, int WAVES, int CHW>
hipError_t fused_f(hipStream_t s, const CullDeviceView& v, uint32_t ent_begin, uint32_t ent_end, const TypeTable& tt, const FrustaArg& fr,
	int32_t* out_ids, uint32_t out_stride, uint32_t* counts, uint32_t* counts_next) {
	constexpr uint32_t TILE = WAVES * CHW * 64;
	static_assert(TILE_ALIGN % TILE == 0, "tiles must not straddle type ranges");
	constexpr int K = TILE == 4096 ? 0 : (TILE == 2048 ? 1 : 2);
	const uint32_t tiles = (ent_end - ent_begin) / TILE;
	if (!tiles) return hipSuccess;
	const uint32_t cell_cap = v.tile_cap[K];
	const size_t lds = fused_lds_bytes(F, TILE, cell_cap);
	hipLaunchKernelGGL((k_cull_fused<F, WAVES, CHW>), dim3(tiles), dim3(WAVES * 64), lds, s, v.spheres, v.ids, v.chunk_cell, v.chunk_flags,
		v.tile_cells[K], v.tile_tab[K], v.tile_box[K], fr, tt, ent_begin, cell_cap, out_ids, out_stride, counts, counts_next);
	return hipGetLastError();
}

} // namespace

uint32_t cull_tile_size(int n_frusta) {
	static const uint32_t f1_tile = [] { // experiment knob: LMX_CULL_TILE=2048 runs the 1-frustum kernel with 4 waves per block
		const char* e = getenv("LMX_CULL_TILE"); // 2048 / 4096 force a tile size (8192 is the internal code for "always 4096")
		return (e && atoi(e) == 2048) ? 2048u : ((e && atoi(e) == 4096) ? 8192u : 4096u);
	}();
	return n_frusta <= 1 ? f1_tile : (n_frusta <= 4 ? 2048u : 1024u);
}

size_t fused_lds_bytes(int n_frusta, uint32_t tile, uint32_t cell_cap) {
	return (size_t)n_frusta * cell_cap * sizeof(float4) + (size_t)n_frusta * tile * sizeof(int32_t) + 2 * MAX_FRUSTA * sizeof(uint32_t);
}

hipError_t launch_cull_fused(hipStream_t s, const CullDeviceView& v, uint32_t ent_begin, uint32_t ent_end, const TypeTable& tt,
	const FrustaArg& fr, int n_frusta, int32_t* out_ids, uint32_t out_stride, uint32_t* counts, uint32_t* counts_next, bool small_tiles) {
#define LMX_FUSED(F, W, C) return fused_f<F, W, C>(s, v, ent_begin, ent_end, tt, fr, out_ids, out_stride, counts, counts_next)
	switch (n_frusta) {
		case 1:
			// 2048-sphere tiles when the frustum covers a small part of the scene (few surviving tiles: shorter per-block chain,
			// 17.9 vs 19.6 us on the 10 M scene), 4096 when much of it is visible (the one returning atomic per block and
			// list is then the limit: 42 vs 69 us with everything visible)
			if (cull_tile_size(1) == 2048u || (small_tiles && cull_tile_size(1) != 8192u)) LMX_FUSED(1, 4, 8);
			LMX_FUSED(1, 8, 8);
		case 2: LMX_FUSED(2, 8, 4);
		case 3: LMX_FUSED(3, 8, 4);
		case 4: LMX_FUSED(4, 8, 4);
		case 5: LMX_FUSED(5, 4, 4);
		case 6: LMX_FUSED(6, 4, 4);
		case 7: LMX_FUSED(7, 4, 4);
		case 8: LMX_FUSED(8, 4, 4);
		default: return hipErrorInvalidValue;
	}
#undef LMX_FUSED
}

hipError_t launch_cull_classify(hipStream_t s, const CullDeviceView& v, uint32_t cell_begin, uint32_t n, const FrustaArg& fr,
	int n_frusta, float4* cellinfo, uint32_t cell_stride, uint32_t* counts) {
	switch (n_frusta) {
		case 1: return classify_f<1>(s, v, cell_begin, n, fr, cellinfo, cell_stride, counts);
		case 2: return classify_f<2>(s, v, cell_begin, n, fr, cellinfo, cell_stride, counts);
		case 3: return classify_f<3>(s, v, cell_begin, n, fr, cellinfo, cell_stride, counts);
		case 4: return classify_f<4>(s, v, cell_begin, n, fr, cellinfo, cell_stride, counts);
		case 5: return classify_f<5>(s, v, cell_begin, n, fr, cellinfo, cell_stride, counts);
		case 6: return classify_f<6>(s, v, cell_begin, n, fr, cellinfo, cell_stride, counts);
		case 7: return classify_f<7>(s, v, cell_begin, n, fr, cellinfo, cell_stride, counts);
		case 8: return classify_f<8>(s, v, cell_begin, n, fr, cellinfo, cell_stride, counts);
		default: return hipErrorInvalidValue;
	}
}

hipError_t launch_cull_spheres(hipStream_t s, const CullDeviceView& v, uint32_t ent_begin, uint32_t ent_end, const TypeTable& tt,
	const FrustaArg& fr, int n_frusta, const float4* cellinfo, uint32_t cell_stride, int32_t* out_ids, uint32_t out_stride,
	uint32_t* counts) {
#define LMX_SPH(F, W, C) return spheres_f<F, W, C>(s, v, ent_begin, ent_end, tt, fr, cellinfo, cell_stride, out_ids, out_stride, counts)
	switch (n_frusta) {
		case 1:
			if (cull_tile_size(1) == 2048u) LMX_SPH(1, 4, 8);
			LMX_SPH(1, 8, 8); // TILE 4096, 16 KiB LDS
		case 2: LMX_SPH(2, 8, 4); // TILE 2048, 16 KiB
		case 3: LMX_SPH(3, 8, 4); // 24 KiB
		case 4: LMX_SPH(4, 8, 4); // 32 KiB
		case 5: LMX_SPH(5, 4, 4); // TILE 1024, 20 KiB
		case 6: LMX_SPH(6, 4, 4);
		case 7: LMX_SPH(7, 4, 4);
		case 8: LMX_SPH(8, 4, 4); // 32 KiB
		default: return hipErrorInvalidValue;
	}
#undef LMX_SPH
}

uint32_t cull_dynamic_tile(int n_frusta) { return n_frusta <= 2 ? 2048u : (n_frusta <= 4 ? 1024u : 512u); } // <= 16 KiB of staging

hipError_t launch_cull_dynamic(hipStream_t s, const DynDeviceView& d, uint32_t slot_begin, uint32_t slot_end, const TypeTable& dyn_tt,
	const FrustaArg& fr, int n_frusta, int32_t* out_ids, uint32_t out_stride, uint32_t* counts) {
	const uint32_t tile = cull_dynamic_tile(n_frusta);
	const uint32_t blocks = (slot_end - slot_begin) / tile;
	if (!blocks) return hipSuccess;
	const size_t lds = (size_t)n_frusta * tile * sizeof(int32_t) + 2 * MAX_FRUSTA * sizeof(uint32_t);
#define LMX_DYN(T) hipLaunchKernelGGL(k_cull_dynamic<T>, dim3(blocks), dim3(DYN_THREADS), lds, s, d.px, d.py, d.pz, d.radius, d.ids, fr, n_frusta, dyn_tt, slot_begin, out_ids, out_stride, counts)
	if (tile == 2048) LMX_DYN(2048);
	else if (tile == 1024) LMX_DYN(1024);
	else LMX_DYN(512);
#undef LMX_DYN
	return hipGetLastError();
}

hipError_t launch_patch_spheres(hipStream_t s, float4* spheres, const uint32_t* slot, const float4* value, uint32_t n) {
	if (!n) return hipSuccess;
	hipLaunchKernelGGL(k_patch_spheres, dim3((n + 255u) / 256u), dim3(256), 0, s, spheres, slot, value, n);
	return hipGetLastError();
}

} // namespace lmx
