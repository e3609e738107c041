// cull_kernels.hip — frustum culling of CullingSystem spheres on gfx950 (wave64).
//
// Replaces CullingSystemImpl::cullInternal / doCulling (src/renderer/culling_system.cpp:262-369) and the per-cell
// ShiftedFrustum tests (src/core/geometry.cpp:99-178). One launch per cull pass:
//
//   k_cull_tile      block = one tile of WAVES x CHW x 64 consecutive spheres of the sorted static set.
//                      0. tile-level box test (tile_status): REJECT ends the block, ACCEPT skips the per-cell work
//                      A. MIXED tiles: one thread per cell x frustum classifies the tile's cells into LDS (fp64 origin shift,
//                         the contains / intersects AABB tests, the getRelative() offset)
//                      B. per wave, autonomous: chunk headers -> per-lane cell -> class from LDS -> ids / spheres fetched only
//                         when a chunk needs them -> 6-plane test -> per-lane visibility bits
//                      C. per wave: ONE returning atomic per frustum on the tile's output shard, then the visible ids go
//                         straight from registers to the reserved range (wave64 ballot + mbcnt ranks)
//   k_cull_dynamic   one thread per unsorted entity (moving / recently added entities), same arithmetic per entity
//   k_cull_finalize / k_cull_consolidate   per-type totals and one contiguous list per (frustum, type) for consumers that want it
//   k_apply_patches  O(1) add / remove / set between culls
//
// Built with -ffp-contract=off: the plane arithmetic must round exactly like the reference's scalar float4.
#include <algorithm>
#include <cstddef>
#include <cstdlib>

#include "lmx_kernels.h"
#include <type_traits>

#include <hip/hip_ext.h>

namespace lmx {

namespace {

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t mbcnt64(uint64_t mask) {
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4))); // a 16-byte access the compiler may only assume 4-byte aligned
__device__ __forceinline__ float readlane_f(float v, int lane) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane)); }

// Everything k_cull_tile needs besides the frusta. The frusta are the FIRST kernel argument (offset 0 of the kernarg segment):
// the lane-parallel tile test reads plane k's coefficients with a per-lane load from there.
// (pointers are separate __restrict__ kernel parameters: only then are the wave-uniform loads selected as scalar loads)
struct TileScalars {
	uint32_t ent_begin, cell_cap, n_frusta;
	uint32_t keys_packed; // the cell tables hold PackedCellKey (8 bytes, offsets against the tile's box) instead of CellKey
	uint32_t out_stride, cnt_pad, cnt_frustum_stride, n_zero;
	int32_t* out_slots; // SLOTS kernels only
	float pretest_n1s;  // (several frusta) 2^27 x the largest |n|_1 of the call's planes, rounded up; +inf switches the matrix-pipe pre-test off
};

// float index of DevFrustum members inside the kernarg segment (frustum 0)
constexpr int KA_NX = 0, KA_NY = 6, KA_NZ = 12, KA_D = 18, KA_ORIGIN = 44;
static_assert(sizeof(DevFrustum) == 200 && offsetof(DevFrustum, origin) == KA_ORIGIN * 4 && offsetof(DevFrustum, d) == KA_D * 4, "kernarg offsets of DevFrustum");

// tile_status() of lmx_math.h for frustum 0, evaluated across the lanes of a wave: lanes 0..11 form the 12 box-corner
// coordinates (fp64 shift, one rounding), lanes 0..5 then evaluate one plane each. Same expressions, same rounding, ~50 VALU
// instructions per wave instead of ~300 uniform ones; on the default camera 95 % of the blocks do nothing else.
__device__ __forceinline__ uint32_t tile_status_lanes(const TileBox* box, uint32_t lane) {
	const float* ka = (const float*)__builtin_amdgcn_kernarg_segment_ptr();
	const int32_t* bi = reinterpret_cast<const int32_t*>(box);
	const uint32_t flags = (uint32_t)bi[6];
	const uint32_t a = lane % 3u, kind = (lane / 3u) & 3u;
	const int32_t idx = bi[(kind & 1u) * 3u + a];
	const double org = reinterpret_cast<const double*>(ka + KA_ORIGIN)[a];
	const uint32_t p = (lane & 7u) < 6u ? (lane & 7u) : 5u;
	const float nx = ka[KA_NX + p], ny = ka[KA_NY + p], nz = ka[KA_NZ + p], d = ka[KA_D + p];
	const uint32_t hi = flags << 2; // the caller also wants the tile's flags
	if (flags & TILE_EMPTY) return TILE_REJECT | hi;
	if (flags & TILE_HAS_BIG) return TILE_MIXED | hi;
	const double cs = (double)CELL_SIZE;
	const double off = kind == 0u ? -cs : (kind == 3u ? 2 * cs : cs);
	const float val = (float)(cs * idx + off - org);
	const float lx = readlane_f(val, 0), ly = readlane_f(val, 1), lz = readlane_f(val, 2);
	const float hx = readlane_f(val, 3), hy = readlane_f(val, 4), hz = readlane_f(val, 5);
	const float clx = readlane_f(val, 6), cly = readlane_f(val, 7), clz = readlane_f(val, 8);
	const float chx = readlane_f(val, 9), chy = readlane_f(val, 10), chz = readlane_f(val, 11);
	const float b1 = max_f(abs_f(lx), abs_f(chx)) + max_f(abs_f(ly), abs_f(chy)) + max_f(abs_f(lz), abs_f(chz));
	const float n1 = abs_f(nx) + abs_f(ny) + abs_f(nz);
	const float margin = max_f(1.0f, n1) * (2.0f + 4e-6f * b1) + 1e-6f * abs_f(d);
	bool rej, in;
	{
		const float bx = nx > 0.0f ? hx : lx, by = ny > 0.0f ? hy : ly, bz = nz > 0.0f ? hz : lz;
		const float dp = (nx * bx) + (ny * by) + (nz * bz);
		rej = dp < -d - margin;
	}
	{
		const float bx = nx < 0.0f ? chx : clx, by = ny < 0.0f ? chy : cly, bz = nz < 0.0f ? chz : clz;
		const float dp = (nx * bx) + (ny * by) + (nz * bz);
		in = dp > -d + margin;
	}
	const bool plane_lane = lane < 6u;
	if (__ballot(plane_lane && rej) != 0) return TILE_REJECT | hi;
	if (__ballot(plane_lane && !in) == 0) return TILE_ACCEPT | hi;
	// MIXED: the planes every cell of the tile passes in both per-cell tests (lmx_math.h: tile_plane_skip_mask) go into bits 8..13
	bool skip;
	{
		const float bx = nx < 0.0f ? chx : lx, by = ny < 0.0f ? chy : ly, bz = nz < 0.0f ? chz : lz;
		const float dp = (nx * bx) + (ny * by) + (nz * bz);
		skip = dp > -d + margin;
	}
	return TILE_MIXED | hi | (((uint32_t)__ballot(plane_lane && skip) & 63u) << 8);
}

// tile_status() for up to 8 frusta at once: lane 6 f + k evaluates plane k of frustum f (<= 48 lanes), every lane forms the box
// corners of ITS frustum itself (fp64 shift, one rounding: the expressions of tile_status(), no cross-lane traffic), the per-frustum
// verdicts come out of two ballots. ~110 instructions for one wave instead of ~300 uniform ones per frustum for every wave: with the
// per-frustum loop a launch of the 8-cascade pass spent most of its instructions here. Returns 2 status bits per frustum; *flags = the
// tile's flags.
__device__ __forceinline__ uint32_t tile_status_lanes_multi(const TileBox* box, uint32_t lane, int nf, uint32_t* flags_out) {
	const float* ka = (const float*)__builtin_amdgcn_kernarg_segment_ptr();
	const int32_t* bi = reinterpret_cast<const int32_t*>(box);
	const uint32_t flags = (uint32_t)bi[6];
	*flags_out = flags;
	if (flags & TILE_EMPTY) return 0u; // TILE_REJECT == 0 for every frustum
	uint32_t all_mixed = 0;
	for (int f = 0; f < nf; ++f) all_mixed |= (uint32_t)TILE_MIXED << (2 * f);
	if (flags & TILE_HAS_BIG) return all_mixed;
	const uint32_t f = lane / 6u < (uint32_t)nf ? lane / 6u : (uint32_t)nf - 1u, k = lane % 6u;
	const bool plane_lane = lane < (uint32_t)nf * 6u;
	const float* fr = ka + f * (uint32_t)(sizeof(DevFrustum) / sizeof(float));
	const float nx = fr[KA_NX + k], ny = fr[KA_NY + k], nz = fr[KA_NZ + k], d = fr[KA_D + k];
	const double* org = reinterpret_cast<const double*>(fr + KA_ORIGIN);
	const double cs = (double)CELL_SIZE;
	const double ox = org[0], oy = org[1], oz = org[2];
	const float lx = (float)(cs * bi[0] - cs - ox), ly = (float)(cs * bi[1] - cs - oy), lz = (float)(cs * bi[2] - cs - oz);
	const float hx = (float)(cs * bi[3] + cs - ox), hy = (float)(cs * bi[4] + cs - oy), hz = (float)(cs * bi[5] + cs - oz);
	const float clx = (float)(cs * bi[0] + cs - ox), cly = (float)(cs * bi[1] + cs - oy), clz = (float)(cs * bi[2] + cs - oz);
	const float chx = (float)(cs * bi[3] + 2 * cs - ox), chy = (float)(cs * bi[4] + 2 * cs - oy), chz = (float)(cs * bi[5] + 2 * cs - oz);
	const float b1 = max_f(abs_f(lx), abs_f(chx)) + max_f(abs_f(ly), abs_f(chy)) + max_f(abs_f(lz), abs_f(chz));
	const float n1 = abs_f(nx) + abs_f(ny) + abs_f(nz);
	const float margin = max_f(1.0f, n1) * (2.0f + 4e-6f * b1) + 1e-6f * abs_f(d);
	bool rej, in;
	{
		const float bx = nx > 0.0f ? hx : lx, by = ny > 0.0f ? hy : ly, bz = nz > 0.0f ? hz : lz;
		const float dp = (nx * bx) + (ny * by) + (nz * bz);
		rej = dp < -d - margin;
	}
	{
		const float bx = nx < 0.0f ? chx : clx, by = ny < 0.0f ? chy : cly, bz = nz < 0.0f ? chz : clz;
		const float dp = (nx * bx) + (ny * by) + (nz * bz);
		in = dp > -d + margin;
	}
	const uint64_t rej_mask = __ballot(plane_lane && rej), out_mask = __ballot(plane_lane && !in);
	uint32_t st_bits = 0;
	for (int g = 0; g < nf; ++g) {
		const uint32_t r = (uint32_t)(rej_mask >> (6 * g)) & 63u, o = (uint32_t)(out_mask >> (6 * g)) & 63u;
		st_bits |= (r != 0 ? (uint32_t)TILE_REJECT : (o == 0 ? (uint32_t)TILE_ACCEPT : (uint32_t)TILE_MIXED)) << (2 * g);
	}
	return st_bits;
}

// sphere_visible_d() of lmx_math.h on packed fp32: two planes per v_pk_mul_f32 / v_pk_add_f32, every product and sum rounded on its own
// exactly like the scalar expression ((cx*nx + cy*ny) + cz*nz) + d, then t - (-r) == t + r. The all-test launch was issue-bound (69 % of
// all SIMD cycles were VALU, profiles/r02/cull_all_test_counters_before_packed_fp32.json); the plane arithmetic is half of its VALU instructions.
// Choices of the 1-frustum kernels that were build knobs while they were measured (rounds 3-5; each had one value ever since, the knobs went in round 6):
//   * non-temporal loads of the streamed spheres / ids in the streaming form (every sphere is read once per cull): cache-cold all-test launch 43.8-44.3 -> 42.8 us,
//     back to back 37.0-37.4 -> 37.9 us (profiles/r03/cull_ab_variants.txt)
//   * the streaming form compacts a wave's visible ids in LDS and writes them with full-width stores instead of one partial-width store per chunk: the launch with
//     43 % visible 47.6 -> 46.0 us. In the latency form (all 8 chunks in flight: the headline camera) it cost +0.3 us per step and +0.9 us cache-cold: not there
//   * phase A leaves out the planes the whole tile is known to pass (lmx_math.h: tile_plane_skip_mask; the emulation re-classifies every cell with and without):
//     launch with every cell CELL_TEST through the AABB pre-tests 49.9 -> 47.6 us; nothing else moves
//   * k_cull_pack: blocks without a slice of their shard's window leave at once, the others fetch their first ids under the prefix of the counters
// Measured and NOT kept (profiles/r03/cull_ab_variants.txt): touching the NEXT tile's box / cell keys / chunk headers at block start
// (global_load_lds into a scratch corner, nothing waits): +2-3 us in every regime, cold included; capping the kernel at 80 SGPRs so
// that 8 instead of 7 blocks are resident per CU (MI355X_MICROARCH.md "Residency"): within noise; a persistent grid for the streaming
// launches (1628 resident blocks x 3 tiles each: no ragged last round, one block start-up per 3 tiles): 67 instead of 42 VGPRs for
// the loop around the tile body, all-test launch 37.3 -> 43.5 us back to back, 43.0 -> 47.9 us cache-cold.

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// {bf16(a), bf16(b) << 16}, round to nearest even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) { const v2f v = {a, b}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t)); }
typedef float v8f_t __attribute__((ext_vector_type(8)));
// min of three, NaN operands dropped (all NaN: NaN): the compiler folds the nested fminf into ONE v_min3_f32 without canonicalising
// its operands (a two-operand fminf gets a v_max_f32 x, x in front of every operand it cannot prove quiet: always spell three)
__device__ __forceinline__ float min3_drop_nan(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ bool sphere_visible_d_pk(const DevFrustum& f, const float d[6], float cx, float cy, float cz, float radius) {
	const v2f x2 = {cx, cx}, y2 = {cy, cy}, z2 = {cz, cz}, r2 = {radius, radius};
	bool culled = false;
#pragma unroll
	for (int k = 0; k < 6; k += 2) {
		const v2f nx = {f.nx[k], f.nx[k + 1]}, ny = {f.ny[k], f.ny[k + 1]}, nz = {f.nz[k], f.nz[k + 1]}, dd = {d[k], d[k + 1]};
		v2f t = x2 * nx;
		t = t + y2 * ny;
		t = t + z2 * nz;
		t = t + dd;
		t = t + r2;
		culled = culled || (t.x < 0) || (t.y < 0);
	}
	return !culled;
}

// ---- sphere x plane pre-test on the matrix pipe (several frusta) ---------------------------------------------------------------
// The 8-frusta pass is bound by instruction issue: doCulling's unfused arithmetic is 21 packed operations + 4 min / compare per (64 spheres,
// frustum), 62.6 M VALU wave-instructions per launch at 10 M all-test (profiles/r04/final/counters_summary.json), and sharing dot
// products between frusta does not apply to real cascades: of the 48 planes of config 5's frusta 34 have bitwise distinct directions
// even up to sign (the reference builds every plane from cross products of corner differences, geometry.cpp:324-337: the cascades of
// one light differ in the last bits, near / far are not exact negations) - profiles/r05/README.md.
// What the test needs bit for bit is only the SIGN of u_k = fl(fl(fl(fl(fl(x nx) + fl(y ny)) + fl(z nz)) + d_k) + r) for six planes. ONE
// v_mfma_f32_32x32x16_bf16 evaluates, for 32 spheres x 32 plane rows (four frusta), u'_k = d_k + sum over the K slots
//     x1 nx1 + y1 ny1 + z1 nz1 + r1 + x2 nx1 + y2 ny1 + z2 nz1 + r2 + x1 nx2 + y1 ny2 + z1 nz2        (v1 = bf16(v), v2 = bf16(v - v1), RNE)
// with d_k in the fp32 accumulator: the leading terms of the two-term splits of every factor. Both values approximate
// T_k = x nx + y ny + z nz + d_k + r. With u = 2^-24, G >= |x nx| + |y ny| + |z nz| + |r| and M_k = G + |d_k| <= 2 G + |T_k|:
//   * |u_k - T_k| <= ((1 + u)^5 - 1) M_k                                                 (five roundings on the longest path)
//   * the splits: |v - v1 - v2| <= 2^-16 |v|, |v2| <= 2^-8 (1 + 2^-8) |v|, so the terms left out (x2 nx2, the remainders times the other
//     factor) are <= 3.01 x 2^-16 |x nx| per axis and 2^-16 |r|: <= 3.01 x 2^-16 G in all
//   * the instruction's own accumulation: <= 16 u x (|d_k| + sum |a b|) assumed, 4.5 u measured over wide exponent ranges, cancellation
//     and |C| >> products (tools/mfma_contract_probe.hip, which also pins the operand maps and the conversion)
// so |u_k - u'_k| <= 21.01 u (2 G + |u'_k|) + 3.01 x 2^-16 G (1.01), and |u'_k| > 4.9e-5 G  =>  sign(u_k) == sign(u'_k), u_k != 0.
// The kernel uses eps = 1.5 x 2^-14 G' + tau = 9.2e-5 G' + tau, G' = (|x| + |y| + |z|) n1 + |r| evaluated in fp32 (n1 >= every plane's
// |n|_1, rounded up on the host), and per (sphere, frustum) with m = min_k u'_k (NaNs dropped):
//     m < -eps                    some plane has u_k < 0: culled, as the reference
//     m >  eps                    every plane has u_k > 0 (a NaN u'_k with finite G' means d_k is NaN, so u_k is NaN too and culls
//                                 nothing): visible, as the reference
//     otherwise (also eps = NaN)  undecided: the chunk is evaluated again by the exact loop below, all frusta
// tau = 2^-120 covers products that underflow (absolute instead of relative error). G' is formed 2^27 times too large and scaled back by
// the last multiplication: any G >= 2^101 makes eps infinite, so sums that could overflow in one evaluation order and not in the other
// are never decided here; infinite / NaN coordinates and radii end in eps = inf / NaN (an infinite v has v - v1 = NaN besides), and
// d_k = +-inf gives u'_k = u_k = +-inf. Undecided (sphere, plane) pairs on a scene of extent 3e4: 2 eps / 6e4 ~ 3e-6 at G ~ 1e3, i.e.
// about 1 % of the chunks take the exact loop.
// Layout: rows = planes. Row i of a group of four frusta 4 g .. 4 g + 3 is plane p = (i & 3) + 4 ((i >> 3) & 1) of frustum
// 4 g + 2 h + q with h = (i >> 2) & 1, q = i >> 4, so that accumulator registers 8 q + p of a lane in half h (= lane >> 5) are the
// six values of ONE sphere (column lane & 31) and ONE frustum: the minimum is lane-local, and the six d_k are three ds_read_b64 straight
// into the accumulator. Columns = 32 spheres: a chunk's 64 spheres are two column groups; v_permlane32_swap hands each half-wave the
// partner sphere's K slots, and brings the verdict bits of the other four frusta back.
// Measured (profiles/r05/cull8_*): 8 frusta x 10 M all-test 141 us (exact loop) -> 137 (the same pre-test on v_mfma_f32_32x32x2_f32, two
// per set: the f32-input MFMA runs at the vector rate and, as tools/mfma_overlap_probe.hip shows, VALU work does not hide behind it in
// compiler-scheduled code - deleted) -> 117 us cold / 110 warm (this form); VALU wave-instructions 62.6 M -> 38.8 M.

// LDS record of one (cell, frustum): the six cell-relative plane distances of ShiftedFrustum::getRelative and the cell's class
struct alignas(16) CellInfo { float d[6]; uint32_t cls, pad; };
static_assert(sizeof(CellInfo) == 32, "two ds_read_b128 per (lane, chunk, frustum)");

// F == 1: the single-frustum kernel (the common case: one launch per view). F == 0: n_frusta (2..8) is a runtime value and every
// per-frustum loop is a real loop, so registers do not scale with the number of frusta (8 unrolled copies needed 172 VGPRs and
// 600 spilled SGPRs); FS is the stride of a chunk's visibility bits.
#ifndef LMX_ASM_SGPR
#define LMX_ASM_SGPR(x) "+s"(x) // an empty asm's operand that pins a wave-uniform value in a scalar register (tests/hostsim's runtime header maps it to a general register)
#endif
// (several frusta) at least 5 waves per SIMD, i.e. at most 96 VGPRs: the residency the LDS of the cell records allows anyway (5 blocks of 4
// waves per CU) - and with a register budget of <= 256 the compiler selects the VGPR form of the MFMA (accumulators in plain VGPRs, where the
// ds_reads put the plane distances and the v_min3 read the results) instead of AGPR accumulators + a v_accvgpr_write / _read per value
#define LMX_CULL_WAVES_ATTR(F) __attribute__((amdgpu_waves_per_eu((F) == 1 ? 1 : 5)))

template <int F, int WAVES, int CHW, int GRP, int FORM, int SLOTS_I>
__global__ __launch_bounds__(WAVES * 64) LMX_CULL_WAVES_ATTR(F) void k_cull_tile(const FrustaArg fr_arg, const float4* __restrict__ g_spheres, const int32_t* __restrict__ g_ids,
	const ChunkHdr* __restrict__ g_hdr, const CellKey* __restrict__ g_tile_cells, const uint32_t* __restrict__ g_tile_tab, const TileBox* __restrict__ g_tile_box,
	const uint2* __restrict__ g_tile_out, int32_t* __restrict__ g_out_ids, uint32_t* __restrict__ g_counts, uint32_t* __restrict__ g_counts_next, const TileScalars a) {
	constexpr bool SLOTS = SLOTS_I != 0; // also write the slot of every visible id (CullOut::slots)
	constexpr uint32_t TILE = WAVES * CHW * 64;
	constexpr uint32_t THREADS = WAVES * 64;
	constexpr int FS = F == 1 ? 1 : MAX_FRUSTA;
	static_assert(CHW * FS <= 32, "visibility bits of a wave's chunks x frusta live in one register");
	LMX_DYNAMIC_LDS(CellInfo, s_info); // [n_frusta * cell_cap] (MIXED tiles only)
	__shared__ __attribute__((aligned(16))) float s_nrm[F != 1 ? MAX_FRUSTA : 1][F != 1 ? 20 : 4]; // (several frusta) plane normals nx[6] ny[6] nz[6] of every frustum, for phase B
	// the frusta are read through the kernarg segment pointer: uniform scalar loads placed where they are used
	const DevFrustum* __restrict__ frp = (const DevFrustum*)__builtin_amdgcn_kernarg_segment_ptr();
	const int nf = F == 1 ? 1 : (int)a.n_frusta;

	const uint32_t lane = lane_id();
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t tile_ent = a.ent_begin + blockIdx.x * TILE;
	const uint32_t tile_index = tile_ent / TILE;

	// the counters of the NEXT cull on this view are cleared here (ping-pong), so no cull needs a separate memset
	for (uint32_t i = blockIdx.x * THREADS + threadIdx.x; i < a.n_zero; i += gridDim.x * THREADS) g_counts_next[i] = 0;

	// 0. tile-level test per frustum (2 bits each). Everything read here sits at addresses that depend on blockIdx only.
	uint32_t st_bits = 0, tile_flags = 0, plane_skip = 0;
	bool any_mixed = false, any_live = false;
	if constexpr (FORM != 0) {
		static_assert(F == 1, "the lane-parallel tile test handles one frustum");
		// wave 0 alone evaluates the verdict and hands it to the others through LDS: on a launch where most tiles are rejected
		// here, the other waves' ~80 VALU instructions each were most of what the chip executed (all-rejected launch of 4883
		// tiles: 8.9 -> 7.3 us, the headline camera 13.1 -> 11.9 us; a launch of the same shape that does nothing takes 1.8 us,
		// with one dependent load per block 2.8 us - tools/launch_floor_probe.hip)
		__shared__ uint32_t s_verdict;
		if (wave == 0) {
			const uint32_t v = tile_status_lanes(g_tile_box + tile_index, lane);
			if (lane == 0) s_verdict = v;
		}
		__syncthreads();
		const uint32_t r = s_verdict;
		st_bits = r & 3u;
		tile_flags = (r >> 2) & 63u;
		plane_skip = __builtin_amdgcn_readfirstlane((r >> 8) & 63u);
		any_mixed = st_bits == TILE_MIXED;
		any_live = st_bits != TILE_REJECT;
	} else {
		// several frusta: wave 0 evaluates every (frustum, plane) pair on its own lane and hands the verdicts to the other waves
		__shared__ uint32_t s_verdict_multi[2];
		if (wave == 0) {
			uint32_t fl;
			const uint32_t v = tile_status_lanes_multi(g_tile_box + tile_index, lane, nf, &fl);
			if (lane == 0) {
				s_verdict_multi[0] = v;
				s_verdict_multi[1] = fl;
			}
		}
		__syncthreads();
		st_bits = __builtin_amdgcn_readfirstlane(s_verdict_multi[0]);
		tile_flags = __builtin_amdgcn_readfirstlane(s_verdict_multi[1]);
		static_assert(TILE_REJECT == 0 && TILE_ACCEPT == 1 && TILE_MIXED == 2, "the verdicts of all frusta are read off the 2-bit fields by bit logic");
		any_mixed = ((st_bits >> 1) & ~st_bits & 0x5555u) != 0; // (frusta >= nf hold TILE_REJECT)
		any_live = st_bits != 0;
	}
	if (!any_live) return;
	// (several frusta) the class word of a cell: CellClass of frustum f in bits 2f, 2f + 1. tile_word = what the tile-level test settled.
	uint32_t tile_word = 0;
	// (several frusta) a cell's record under a frustum is its six plane distances, 24 bytes - the class lives in the word: 25 % less LDS per
	// block than the 32-byte CellInfo, and LDS is what bounds the resident blocks of this kernel (8 frusta x ~140 cells per tile)
	v2f* s_d2 = reinterpret_cast<v2f*>(s_info);                                                  // [n_frusta * cell_cap][3]
	uint32_t* s_word = reinterpret_cast<uint32_t*>(s_d2 + (F != 1 ? (size_t)nf * a.cell_cap * 3 : 0)); // [cell_cap], behind the records
	// (the 5..8-frusta shape) classification is split over the WAVES by frustum: wave w classifies frusta w, w + 4 for ALL the tile's cells, 64 cells per iteration.
	// With a lane per cell and all frusta in every wave, a tile of ~100 cells (the 10 M all-test scene) ran 8 frustum bodies per wave with 100 of 256 lanes busy;
	// now 2 iterations x 2 frusta. A cell's class word is the OR of the four waves' 16-bit parts: s_w16[4 t + w], read back as one 8-byte LDS load.
	constexpr bool SPLIT = F != 1 && WAVES == 4;
	uint16_t* s_w16 = reinterpret_cast<uint16_t*>(s_word); // [cell_cap][4] (SPLIT)
	if constexpr (F != 1) {
		static_assert(TILE_ACCEPT == 1 && CELL_ACCEPT == 1, "a settled frustum contributes CELL_ACCEPT or nothing: its verdict field IS its class field");
		tile_word = st_bits & ~(st_bits >> 1) & 0x5555u;
	}

	// (streaming forms) the wave's first group of spheres and ids, requested UNDER phase A: a MIXED tile's blocks otherwise have nothing in flight from the
	// verdict to the barrier behind the classification (keys -> classes -> barrier -> headers), and a launch's first round of blocks all sit there at once.
	// Requested behind the keys: loads return in order, the classification waits for the keys only. What a chunk turns out not to need is ignored.
	constexpr bool STREAMING = F == 1 && (FORM == 2 || FORM == 4); // (template slot FORM: 0 = several frusta; one frustum: 1 = latency form, 2 = streaming form, 3 / 4 = the same two with 16-byte cell keys)
	constexpr bool PRELOAD = STREAMING || F != 1; // (the several-frusta kernels: their one group)
	typedef float v4f_pre __attribute__((ext_vector_type(4)));
	v4f_pre pre_sp[PRELOAD ? GRP : 1];
	int32_t pre_id[PRELOAD ? GRP : 1];
	auto preload_group0 = [&](auto with_spheres) {
		if constexpr (PRELOAD) {
			const size_t e0 = ((size_t)((tile_ent >> 6) + wave * CHW) << 6) + lane;
#pragma unroll
			for (int i = 0; i < GRP; ++i) {
				pre_id[i] = __builtin_nontemporal_load(g_ids + e0 + i * 64);
				if constexpr (decltype(with_spheres)::value) pre_sp[i] = __builtin_nontemporal_load(reinterpret_cast<const v4f_pre*>(g_spheres + e0 + i * 64));
			}
		}
	};
	uint32_t first_cell = 0;
	if (any_mixed) {
		// A. classify the tile's cells into LDS: class + the cell-relative plane distances (per cell, as the reference's getRelative)
		first_cell = g_tile_tab[2 * tile_index];
		const uint32_t n_cells = g_tile_tab[2 * tile_index + 1];
		// a tile's cell table: 16-byte keys or (the common case) 8-byte ones relative to the tile's box, whose low corner is a uniform 12-byte load.
		// ONE base pointer for both forms (the streaming kernel sits at its limit of 80 scalar registers)
		// (one frustum: the key format is part of the instantiation - FORM 1 / 2 packed, 3 / 4 wide - because both decoders in one body cost the streaming
		// kernel the 80-SGPR limit of 8 resident blocks per CU; the several-frusta kernels, at 5 blocks per CU by their LDS, branch on a launch-uniform flag)
		const bool packed_keys = F == 1 ? FORM <= 2 : a.keys_packed != 0;
		const char* keys = reinterpret_cast<const char*>(g_tile_cells) + ((size_t)tile_index * a.cell_cap << (packed_keys ? 3 : 4));
		struct FetchedKey { uint4 raw; int32_t lo[3]; }; // a key as it comes out of memory (+ the tile box's low corner, for a packed one): requested in one place, decoded in another
		auto fetch_key = [&](uint32_t t) -> FetchedKey {
			FetchedKey k;
			if (packed_keys) { // launch-uniform
				const uint2 raw = reinterpret_cast<const uint2*>(keys)[t];
				k.raw = uint4{raw.x, raw.y, 0u, 0u};
				// (the box's low corner through the VECTOR memory path - an opaque zero in the address - so that it lands in vector registers, of which
				// phase A has plenty: as scalars it puts the streaming form at 81 SGPRs, one over what 8 resident blocks per CU allow)
				uint32_t zero = 0;
				asm volatile("" : "+v"(zero));
				const int32_t* box_lo = reinterpret_cast<const int32_t*>(g_tile_box + tile_index) + zero;
				k.lo[0] = box_lo[0]; k.lo[1] = box_lo[1]; k.lo[2] = box_lo[2];
			} else {
				k.raw = reinterpret_cast<const uint4*>(keys)[t]; // ONE 16-byte load (ix, iy, iz, meta), not meta -> branch -> the rest
				k.lo[0] = k.lo[1] = k.lo[2] = 0;
			}
			return k;
		};
		auto decode_key = [&](const FetchedKey& k) -> CellKey {
			CellKey key;
			if (packed_keys) {
				key.ix = k.lo[0] + (int32_t)(k.raw.x & 0xffffu);
				key.iy = k.lo[1] + (int32_t)(k.raw.x >> 16);
				key.iz = k.lo[2] + (int32_t)(k.raw.y & 0xffffu);
				key.meta = ((k.raw.y & PACKED_CELL_BIG) ? 0x100u : 0u) | ((k.raw.y & PACKED_CELL_DEAD) ? (uint32_t)CELL_DEAD : 0u);
			} else {
				key.ix = (int32_t)k.raw.x; key.iy = (int32_t)k.raw.y; key.iz = (int32_t)k.raw.z; key.meta = k.raw.w;
			}
			return key;
		};
		auto load_key = [&](uint32_t t) -> CellKey { return decode_key(fetch_key(t)); };
		auto classify = [&](uint32_t t, int f, const CellKey key) -> uint32_t {
			const bool dead = (key.meta & CELL_DEAD) != 0;
			const bool big = (key.meta & 0x100u) != 0;
			CellInfo ci;
			ci.cls = CELL_REJECT;
			ci.pad = 0u;
#pragma unroll
			for (int k = 0; k < 6; ++k) ci.d[k] = 0.f;
			if (!dead) {
				V3 off;
				ci.cls = classify_cell(frp[f], IV3{key.ix, key.iy, key.iz}, big, &off, F == 1 ? plane_skip : 0u);
				if (ci.cls == CELL_TEST) {
					if constexpr (F != 1) {
						// relative_plane_d for two planes at a time (the frustum's arrays hold planes k, k + 1 side by side: packed operands out of the
						// kernarg segment): q = point + offset, d = -((q.x n.x + q.y n.y) + q.z n.z), every operation rounded on its own as the scalar form
						const v2f ox = {off.x, off.x}, oy = {off.y, off.y}, oz = {off.z, off.z};
#pragma unroll
						for (int k = 0; k < 6; k += 2) {
							const v2f qx = v2f{frp[f].px[k], frp[f].px[k + 1]} + ox, qy = v2f{frp[f].py[k], frp[f].py[k + 1]} + oy, qz = v2f{frp[f].pz[k], frp[f].pz[k + 1]} + oz;
							v2f t = qx * v2f{frp[f].nx[k], frp[f].nx[k + 1]};
							t = t + qy * v2f{frp[f].ny[k], frp[f].ny[k + 1]};
							t = t + qz * v2f{frp[f].nz[k], frp[f].nz[k + 1]};
							ci.d[k] = -t.x; ci.d[k + 1] = -t.y;
						}
					} else {
#pragma unroll
						for (int k = 0; k < 6; ++k) ci.d[k] = relative_plane_d(frp[f], off, k);
					}
				}
			}
			if constexpr (F != 1) {
				v2f* rec = s_d2 + (size_t)(f * a.cell_cap + t) * 3;
				rec[0] = v2f{ci.d[0], ci.d[1]}; rec[1] = v2f{ci.d[2], ci.d[3]}; rec[2] = v2f{ci.d[4], ci.d[5]};
			} else {
				s_info[f * a.cell_cap + t] = ci;
			}
			return ci.cls;
		};
		if constexpr (F == 1) {
			// The lane's first key is requested NOW, next to the tile table's scalar load, and pinned: written as `key = load(t); if (t < n_cells) ...` the compiler
			// sank the load behind the wait for n_cells (seen in the ISA: table -> wait -> key -> wait -> classify; the tail of a tile's slice holds dead keys
			// precisely so that the load needs no bound). Tiles of more than 256 cells (<= 480) take a second, ordinary iteration.
			const FetchedKey fetched0 = fetch_key(threadIdx.x < a.cell_cap ? threadIdx.x : 0u);
			__builtin_amdgcn_sched_barrier(0);
			preload_group0(std::true_type{});
			__builtin_amdgcn_sched_barrier(0);
			CellKey key0 = decode_key(fetched0);
			asm volatile("" : "+v"(key0.ix), "+v"(key0.iy), "+v"(key0.iz), "+v"(key0.meta));
			for (uint32_t t = threadIdx.x; t < n_cells; t += THREADS) classify(t, 0, t == threadIdx.x ? key0 : load_key(t));
		} else {
			// Several frusta: a lane per cell, the MIXED frusta in an inner, wave-uniform loop (scalar loads of the frustum); the cell's key is loaded once
			// and its fp64 origin formed once for the frusta of the loop. The 8-wave shape (<= 4 frusta) runs all frusta in every wave, a cell per thread;
			// the 4-wave shape (5..8 frusta) gives wave w the frusta w and w + 4 and ALL the tile's cells, 64 per iteration (SPLIT above: round 6 - full lanes
			// weigh more than the keys and origins formed four times over). Rounds 2 / 3 spread (cell, frustum) PAIRS over the waves with a load -> wait ->
			// branch -> load chain in each of 16 half-empty wave iterations per tile: on the 10 M all-test scene (one cell per ~10 spheres) the classification
			// issued more instructions than the 8 x 10 M sphere tests (72.6 M VALU wave-instructions a launch, 35 M of them the tests: profiles/r04/cull8_counters).
			// Frusta whose tile verdict is not MIXED need no per-cell work (phase B reads their verdict from st_bits).
			// The classes of a cell under ALL frusta are also packed into one word (2 bits a frustum; a frustum the tile-level test settled
			// contributes its verdict): phase B learns what a chunk needs from ONE LDS read per chunk instead of one per (chunk, frustum) - a
			// chain of 32 dependent LDS round trips per wave in front of its loads (profiles/r04/cull8_probes.txt: 96 of the launch's 201 us
			// were neither classification nor sphere tests).
			const uint32_t t_first = SPLIT ? lane : threadIdx.x, t_step = SPLIT ? 64u : THREADS;
			const FetchedKey fetched0 = fetch_key(t_first < a.cell_cap ? t_first : 0u); // (the tail of a tile's slice holds dead keys: no bound needed)
			// (split form: the lane's second cell as well - a tile of the 10 M scenes holds ~100 cells, two iterations of 64 lanes - instead of a load and its wait inside the loop)
			FetchedKey fetched1 = fetched0;
			if constexpr (SPLIT) fetched1 = fetch_key(t_first + t_step < a.cell_cap ? t_first + t_step : 0u);
			__builtin_amdgcn_sched_barrier(0);
			preload_group0(std::true_type{});
			__builtin_amdgcn_sched_barrier(0);
			const uint32_t my_fields = SPLIT ? 0x0303u << (2u * wave) : 0xffffu; // the class fields of this wave's frusta (w, w + 4)
			for (uint32_t t = t_first; t < n_cells; t += t_step) {
				const CellKey key = t == t_first ? decode_key(fetched0) : (SPLIT && t == t_first + t_step ? decode_key(fetched1) : load_key(t));
				uint32_t word = tile_word & my_fields;
#pragma unroll 1
				for (int f = SPLIT ? (int)wave : 0; f < nf; f += SPLIT ? WAVES : 1) {
					if (((st_bits >> (2 * f)) & 3u) != TILE_MIXED) continue; // wave-uniform
					word |= classify(t, f, key) << (2 * f);
				}
				if constexpr (SPLIT) s_w16[4u * t + wave] = (uint16_t)word;
				else s_word[t] = word;
			}
		}
		if constexpr (F != 1) {
			for (uint32_t t = threadIdx.x; t < (uint32_t)nf * 18u; t += THREADS) {
				const uint32_t f = t / 18u, k = t % 18u;
				s_nrm[f][k] = k < 6u ? frp[f].nx[k] : (k < 12u ? frp[f].ny[k - 6u] : frp[f].nz[k - 12u]);
			}
		}
		// one barrier per MIXED tile. (A block-wide vote "does any cell survive" would let such a tile end here, but costs two more
		// barriers on every MIXED tile; waves whose chunks all sit in rejected cells fall through below without loading anything.)
		__syncthreads();
	}

	// the tile's output shard and where that shard's window starts: one entry of a table the host derives from the type ranges and the
	// output layout (CullDeviceView::tile_out)
	const uint2 tile_out = g_tile_out[tile_index];
	const uint32_t shard = tile_out.x, win = tile_out.y;

	// Accepted tile without padding or tombstones (1-frustum kernels): its ids are a straight copy. The wave's count is known
	// (CHW x 64), so the reservation does not have to wait for the loads' results, and ids move 16 bytes per lane.
	if constexpr (F == 1) {
		if (!any_mixed && (tile_flags & TILE_DENSE)) {
			constexpr uint32_t PER_WAVE = CHW * 64;
			const uint4* src = reinterpret_cast<const uint4*>(g_ids + ((size_t)((tile_ent >> 6) + wave * CHW) << 6));
			uint4 v[CHW / 4];
#pragma unroll
			for (int k = 0; k < CHW / 4; ++k) v[k] = src[k * 64 + lane];
			// the reservation goes out BEHIND the loads, in the same breath: the compiler waits for a returning atomic at the end of
			// its `lane == 0` branch (vmcnt(0)), so issued first it stood, one full round trip, in front of the loads
			__builtin_amdgcn_sched_barrier(0);
			uint32_t base = 0;
			if (lane == 0) base = atomicAdd(&g_counts[shard * a.cnt_pad], PER_WAVE);
			base = __builtin_amdgcn_readfirstlane(base) + win;
			// the reserved range starts at an arbitrary id: 16-byte stores need 4-byte alignment only on global memory
			int32_t* dst = g_out_ids + base;
#pragma unroll
			for (int k = 0; k < CHW / 4; ++k) {
				u32x4_a4 t = {v[k].x, v[k].y, v[k].z, v[k].w};
				*reinterpret_cast<u32x4_a4*>(dst + (k * 64 + lane) * 4) = t; // global_store_dwordx4, lanes contiguous: 1 KiB per instruction
			}
			if constexpr (SLOTS) { // the slots of a dense tile are consecutive
				int32_t* sdst = a.out_slots + base;
				const uint32_t s0 = ((tile_ent >> 6) + wave * CHW) << 6;
#pragma unroll
				for (int k = 0; k < CHW / 4; ++k) {
					const uint32_t s = s0 + (k * 64 + lane) * 4;
					u32x4_a4 t = {s, s + 1, s + 2, s + 3};
					*reinterpret_cast<u32x4_a4*>(sdst + (k * 64 + lane) * 4) = t;
				}
			}
			return;
		}
	}

	if constexpr (PRELOAD) {
		if (!any_mixed) preload_group0(std::false_type{}); // an accepted tile with padding or tombstones: its first group's ids (no sphere of it is ever tested)
	}

	// (several frusta) per-tile operands of the matrix-pipe pre-test: the plane rows of each group of four frusta (the A operands of the
	// two instructions: {nx | ny} and {nz | 1} by half-wave), where a lane's two frustum slots keep their cell records, and which of
	// them the tile-level test left MIXED (only those can leave a sphere undecided: the records of the others were never written)
	constexpr int NG = F != 1 ? (WAVES == 4 ? 2 : 1) : 1; // groups of four frusta: the 5..8-frusta shape has 4 waves, the 2..4-frusta shape 8
	constexpr bool PRETEST = F != 1;
	u32x4_t pre_ab[NG]; // the plane rows of a group: eight K slots per half-wave
	uint32_t pre_off[NG][2];
	uint64_t pre_care[NG][2];
	// The pre-test evaluates ALL frusta of a group for a chunk at once; the exact loop only the (chunk, frustum) pairs with a lane in a
	// CELL_TEST cell. A tile that few frusta left MIXED (a sparse scene: most cells are settled by the cell tests) is cheaper there.
	constexpr uint32_t PRETEST_MIN_MIXED = 3;
	bool pretest_tile = false;
	if constexpr (PRETEST) {
		const uint32_t mixed_bits = (st_bits >> 1) & ~st_bits & 0x5555u; // bit 2 f: frustum f is MIXED
		pretest_tile = (uint32_t)__popc(mixed_bits) >= PRETEST_MIN_MIXED;
		if (pretest_tile) {
			const uint32_t row = lane & 31u, k = lane >> 5;
			const uint32_t rp = (row & 3u) + 4u * ((row >> 3) & 1u), rh = (row >> 2) & 1u, rq = row >> 4;
#pragma unroll
			for (int gq = 0; gq < NG; ++gq) {
				const uint32_t rf = 4u * gq + 2u * rh + rq;
				const bool row_on = rp < 6u && rf < (uint32_t)nf;
				const uint32_t fs = row_on ? rf : 0u, ps = row_on ? rp : 0u;
				// K slots of a plane row: lanes < 32 hold {nx1 ny1 nz1 1 nx1 ny1 nz1 1} (against {x1 y1 z1 r1 x2 y2 z2 r2} of a sphere), lanes >= 32
				// {nx2 ny2 nz2 0 0 0 0 0} (against {x1 y1 z1 r1 0 0 0 0}): n1 = bf16(n), n2 = bf16(n - n1), both rounded to nearest
				const float nx = row_on ? s_nrm[fs][ps] : 0.f, ny = row_on ? s_nrm[fs][6u + ps] : 0.f, nz = row_on ? s_nrm[fs][12u + ps] : 0.f;
				const uint32_t hi_xy = pk_bf16(nx, ny), hi_z1 = pk_bf16(nz, row_on ? 1.0f : 0.f);
				const uint32_t lo_xy = pk_bf16(nx - __uint_as_float(hi_xy << 16), ny - __uint_as_float(hi_xy & 0xffff0000u)), lo_z0 = pk_bf16(nz - __uint_as_float(hi_z1 << 16), 0.f);
				pre_ab[gq] = k ? u32x4_t{lo_xy, lo_z0, 0u, 0u} : u32x4_t{hi_xy, hi_z1, hi_xy, hi_z1};
#pragma unroll
				for (int q = 0; q < 2; ++q) {
					const uint32_t fl = 4u * gq + 2u * k + q; // the frustum whose six values this lane finds in accumulators 8 q .. 8 q + 5
					pre_off[gq][q] = (fl < (uint32_t)nf ? fl : (uint32_t)nf - 1u) * a.cell_cap * 3u;
					const uint32_t f_lo = 4u * gq + q, f_hi = f_lo + 2u;
					pre_care[gq][q] = ((mixed_bits >> (2 * f_lo)) & 1u ? 0x00000000ffffffffull : 0ull) | ((mixed_bits >> (2 * f_hi)) & 1u ? 0xffffffff00000000ull : 0ull);
				}
			}
		}
	}

	// B. this wave's CHW chunks, in groups of GRP so that at most GRP chunks' worth of spheres are live in registers
	// (VGPR count decides how many tiles a CU keeps in flight). Fetching the headers at kernel start through lanes (one vector
	// load + v_readlane) instead of scalar loads here measured no gain, HBM-cold included: other waves cover the load.
	const uint32_t chunk0 = (tile_ent >> 6) + wave * CHW;
	static_assert(CHW % GRP == 0, "groups tile the wave's chunks");
	int32_t id[CHW];
	uint32_t vis_bits = 0; // bit i * FS + f: sphere `lane` of chunk i is visible in frustum f
	uint32_t mine = 0;     // lane f: this wave's visible count for frustum f
	uint32_t vis2[F != 1 ? CHW : 1]; // (F != 1) bit 2f: sphere `lane` of chunk i is visible in frustum f (the class words' spacing)
#pragma unroll
	for (int i = 0; i < (F != 1 ? CHW : 1); ++i) vis2[i] = 0;
	static_assert(CHW * 64 <= 1023, "a wave's count per frustum fits 10 bits");
	// 1-frustum kernels: the wave's visible ids (and slots) are compacted in LDS as they are found - the write-out below is then a
	// handful of full-width stores instead of one partial-width store per chunk (156 k of them on a launch with 43 % visible)
	constexpr bool STAGE = STREAMING; // streaming variants only (as the non-temporal loads): the latency variant pays for the extra LDS and the wait at the wave's end
	__shared__ int32_t s_stage_ids[STAGE ? WAVES : 1][STAGE ? CHW * 64 : 1];
	__shared__ int32_t s_stage_slots[STAGE && SLOTS ? WAVES : 1][STAGE && SLOTS ? CHW * 64 : 1];
	uint32_t staged = 0; // wave-uniform
#pragma unroll
	for (int g = 0; g < CHW; g += GRP) {
		__builtin_amdgcn_sched_barrier(0); // keep the groups' loads from being hoisted over each other (register peak)
		uint32_t local[GRP];
		uint32_t cls_word[GRP]; // the class of the lane's cell: one frustum - the CellClass itself; several - 2 bits per frustum
		// bit i: chunk i of the group has a lane in a live cell (its ids are needed) / in a CELL_TEST cell (its spheres are). Wave-uniform, ONE scalar
		// register each (GRP bools of this kind live as GRP 64-bit lane masks: 16 scalar registers of a kernel that sits at its limit of 80)
		uint32_t need_id_bits = 0, need_sphere_bits = 0;
		// The group's chunk headers, then the classes of the lanes' cells, each as ONE batch: all headers requested (they are adjacent: a
		// 64-byte scalar load for four chunks) before the first is used, all class reads issued before the first is looked at. Round 5 had
		// header -> wait -> LDS read -> wait per chunk inside one `if (any_mixed)` per chunk: 2 x GRP serialised round trips in front of the
		// group's first sphere load (seen in the ISA, profiles/r06/cull1_isa_before_after.txt), and the class read again in front of every test.
		if (any_mixed) {
			ChunkHdr hdr_g[GRP];
#pragma unroll
			for (int i = 0; i < GRP; ++i) hdr_g[i] = g_hdr[chunk0 + g + i]; // wave-uniform
#pragma unroll
			for (int i = 0; i < GRP; ++i) local[i] = hdr_g[i].cell + mbcnt64(hdr_g[i].flags >> 1) - first_cell; // cell boundaries at positions 1..lane: two v_mbcnt on a wave-uniform mask
#pragma unroll
			for (int i = 0; i < GRP; ++i) {
				if constexpr (SPLIT) {
					const uint2 parts = *reinterpret_cast<const uint2*>(s_w16 + 4u * local[i]); // the four waves' parts of the cell's class word
					const uint32_t both = parts.x | parts.y;
					cls_word[i] = (both | (both >> 16)) & 0xffffu;
				} else {
					cls_word[i] = F != 1 ? s_word[local[i]] : s_info[local[i]].cls;
				}
			}
			if constexpr (F == 1) { // one frustum: from here on a lane's cell is the byte offset of its 32-byte record, with the class in the free low bits (ONE register per chunk)
#pragma unroll
				for (int i = 0; i < GRP; ++i) cls_word[i] |= local[i] << 5;
			}
		} else {
#pragma unroll
			for (int i = 0; i < GRP; ++i) {
				local[i] = 0;
				cls_word[i] = F != 1 ? tile_word : (uint32_t)CELL_ACCEPT; // no frustum is MIXED and at least one is ACCEPT
			}
		}
#pragma unroll
		for (int i = 0; i < GRP; ++i) {
			const bool lane_live = F != 1 ? cls_word[i] != 0 : (cls_word[i] & 3u) != CELL_REJECT;
			const bool lane_test = F != 1 ? (cls_word[i] & 0xaaaau) != 0 /* CELL_TEST = 2: the odd bits */ : (cls_word[i] & 3u) == CELL_TEST;
			// (population counts of the ballots: scalar instructions all the way - spelled as `ballot != 0 ? 1 : 0` the bits were assembled in vector registers)
			need_id_bits |= ((uint32_t)__popcll(__ballot(lane_live)) != 0u ? 1u : 0u) << i;
			need_sphere_bits |= ((uint32_t)__popcll(__ballot(lane_test)) != 0u ? 1u : 0u) << i;
		}
#define need_id(i) (((need_id_bits >> (i)) & 1u) != 0)
#define need_sphere(i) (((need_sphere_bits >> (i)) & 1u) != 0)
		float4 sp[GRP];
		typedef float v4f __attribute__((ext_vector_type(4)));
		v4f spv[F != 1 ? GRP : 1]; // (several frusta) the spheres as register tuples
#pragma unroll
		for (int i = 0; i < GRP; ++i) {
			const uint32_t e = ((chunk0 + g + i) << 6) + lane;
			if constexpr (F != 1) {
				// Several frusta: the loads are UNCONDITIONAL. Under `if (need_sphere(i))` each load's destination merges with the zeros of the
				// other branch, the compiler resolved that with copies right behind the load - and an s_waitcnt vmcnt(0) in front of them, i.e.
				// in front of the next chunk's loads: four serialised round trips to memory per wave (seen in the ISA; the 1-frustum kernels
				// issue all their loads first). A chunk that needs nothing reads one 16-byte sphere / one id at a wave-uniform address instead
				// (the wave's first entity: one cache line, usually the one its neighbour chunk fetches anyway) and ignores the value.
				// (round 6: requested under phase A, all of them - what a chunk turns out not to need is ignored)
				static_assert(F == 1 || GRP == CHW, "one group");
				(void)e;
				id[g + i] = pre_id[i];
				sp[i] = make_float4(pre_sp[i].x, pre_sp[i].y, pre_sp[i].z, pre_sp[i].w);
				continue;
			}
			// One frustum: the loads are UNCONDITIONAL too (round 6). Under `if (need_sphere(i))` every load cost a branch, five moves for the zeros of the
			// other arm and a 64-bit address of its own, and the first test waited for ALL of the group's loads (the compiler cannot count loads behind
			// branches): ~12 of a chunk's ~63 vector instructions on a launch bound by their issue (profiles/r06/cull1_probes.txt). Now: the group's
			// uniform base + ONE lane offset + an immediate per chunk; a chunk that needs nothing reads one line at lane offset 0 and ignores it (its
			// lanes' classes are CELL_REJECT: whatever the registers hold, nothing of it is visible).
			if constexpr (PRELOAD) {
				if (g == 0) { // requested under phase A (an accepted tile's ids: below the dense copy)
					id[i] = pre_id[i];
					sp[i] = make_float4(pre_sp[i].x, pre_sp[i].y, pre_sp[i].z, pre_sp[i].w);
					continue;
				}
			}
			const uint32_t off_i = (lane * 4u) & (0u - ((need_id_bits >> i) & 1u)), off_s = (lane * 16u) & (0u - ((need_sphere_bits >> i) & 1u)); // (bytes: one select each)
			const int32_t* id_at = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(g_ids + ((size_t)(chunk0 + g) << 6) + i * 64) + off_i);
			const v4f* sp_at = reinterpret_cast<const v4f*>(reinterpret_cast<const char*>(g_spheres + ((size_t)(chunk0 + g) << 6) + i * 64) + off_s);
			v4f t;
			if constexpr (F == 1 && (FORM == 2 || FORM == 4)) { // the streaming form: every sphere is read once per cull and nothing of it is reused
				id[g + i] = __builtin_nontemporal_load(id_at);
				t = __builtin_nontemporal_load(sp_at);
			} else {
				id[g + i] = *id_at;
				t = *sp_at;
			}
			sp[i] = make_float4(t.x, t.y, t.z, t.w);
		}
		if constexpr (F != 1) {
			// (the values pass through here together: all eight loads are in flight before the first wait)
#pragma unroll
			for (int i = 0; i < GRP; ++i) {
				v4f t = {sp[i].x, sp[i].y, sp[i].z, sp[i].w}; // (as ONE register tuple: the tests below use its halves as packed operands)
				asm volatile("" : "+v"(t), "+v"(id[g + i]));
				spv[i] = t;
			}
#pragma unroll
			for (int i = 0; i < GRP; ++i) {
				if (!need_id(i)) id[g + i] = -1; // (wave-uniform select behind the loads)
			}
			uint32_t culled2[GRP]; // bit 2f: frustum f culls the lane's sphere of chunk i (read where the cell's class is CELL_TEST)
#pragma unroll
			for (int i = 0; i < GRP; ++i) culled2[i] = 0;
			uint32_t word_or = 0; // the lane's classes over the group's chunks
#pragma unroll
			for (int i = 0; i < GRP; ++i) word_or |= cls_word[i];
			// The pre-test on the matrix pipe (see "sphere x plane pre-test" above): per chunk with a lane in a CELL_TEST cell, the verdict
			// bits of all frusta; a chunk with an undecided (sphere, frustum) pair of a MIXED frustum is left to the exact loop below.
			uint32_t exact_chunks = (1u << GRP) - 1u; // chunks the exact loop evaluates
			if constexpr (PRETEST) {
				if (pretest_tile) {
					exact_chunks = 0;
#pragma unroll
					for (int i = 0; i < GRP; ++i) {
						if (!need_sphere(i)) continue; // wave-uniform
						const float x = spv[i].x, y = spv[i].y, z = spv[i].z, r = spv[i].w;
						// eps of the lane's own sphere (the bound above; formed 2^27 too large so that huge operands end in inf)
						const float gs = ((__builtin_fabsf(x) + __builtin_fabsf(y)) + __builtin_fabsf(z)) * a.pretest_n1s + __builtin_fabsf(r) * 134217728.0f;
						const float eps_own = gs * 6.821210263296962e-13f /* 1.5 x 2^-14 x 2^-27 */ + 7.52316384526264e-37f /* tau = 2^-120 */;
						// what each half-wave needs of the partner sphere: [own | partner] per column group
						const auto sw_eps = __builtin_amdgcn_permlane32_swap(__float_as_uint(eps_own), __float_as_uint(eps_own), false, false);
						const auto sw_loc = __builtin_amdgcn_permlane32_swap(local[i] * 3u, local[i] * 3u, false, false);
						// the sphere as K slots: {x1 y1 z1 r1 x2 y2 z2 r2} for the lanes < 32, {x1 y1 z1 r1 0 0 0 0} for the others (v1 = bf16(v), v2 = bf16(v - v1),
						// round to nearest: v_cvt_pk_bf16_f32; v - v1 is exact)
						const uint32_t hi_xy = pk_bf16(x, y), hi_zr = pk_bf16(z, r);
						const uint32_t lo_xy = pk_bf16(x - __uint_as_float(hi_xy << 16), y - __uint_as_float(hi_xy & 0xffff0000u)), lo_zr = pk_bf16(z - __uint_as_float(hi_zr << 16), r - __uint_as_float(hi_zr & 0xffff0000u));
						const auto sw_xy = __builtin_amdgcn_permlane32_swap(hi_xy, hi_xy, false, false), sw_zr = __builtin_amdgcn_permlane32_swap(hi_zr, hi_zr, false, false);
						const auto sw_lo0 = __builtin_amdgcn_permlane32_swap(lo_xy, 0u, false, false), sw_lo1 = __builtin_amdgcn_permlane32_swap(lo_zr, 0u, false, false);
						uint64_t unsure = 0; // lanes with an undecided pair (wave-uniform accumulation of compare masks)
						uint32_t w_cg[2];
#pragma unroll
						for (int cg = 0; cg < 2; ++cg) { // column group: spheres 32 cg .. 32 cg + 31 of the chunk
							const float eps = __uint_as_float(sw_eps[cg]);
							f32x16 acc[NG];
#pragma unroll
							for (int gq = 0; gq < NG; ++gq) {
								// the (cell, frustum) records of the lane's two frustum slots: three ds_read_b64 each, straight into accumulators 8 q .. 8 q + 5
								// (6, 7 stay undefined - rows of zero planes, never read: spelled as shuffles with undefined lanes, a zero there costs a v_mov each)
								v8f_t half[2];
#pragma unroll
								for (int q = 0; q < 2; ++q) {
									const v2f* rec = s_d2 + pre_off[gq][q] + sw_loc[cg];
									const v2f d01 = rec[0], d23 = rec[1], d45 = rec[2];
									const v4f_t lo = __builtin_shufflevector(d01, d23, 0, 1, 2, 3), hi = __builtin_shufflevector(d45, d45, 0, 1, -1, -1);
									half[q] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
								}
								acc[gq] = __builtin_shufflevector(half[0], half[1], 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
							}
							const u32x4_t bb = {sw_xy[cg], sw_zr[cg], sw_lo0[cg], sw_lo1[cg]};
#pragma unroll
							for (int gq = 0; gq < NG; ++gq) acc[gq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, pre_ab[gq]), __builtin_bit_cast(bf16x8_t, bb), acc[gq], 0, 0, 0);
							uint32_t w = 0;
#pragma unroll
							for (int gq = 0; gq < NG; ++gq) {
#pragma unroll
								for (int q = 0; q < 2; ++q) {
									const float m = min3_drop_nan(min3_drop_nan(acc[gq][8 * q], acc[gq][8 * q + 1], acc[gq][8 * q + 2]), acc[gq][8 * q + 3], min3_drop_nan(acc[gq][8 * q + 4], acc[gq][8 * q + 5], acc[gq][8 * q])); // (the first value twice: three v_min3)
									w |= m < -eps ? 1u << (8 * gq + 2 * q) : 0u;
									unsure |= __ballot(!(__builtin_fabsf(m) > eps)) & pre_care[gq][q];
								}
							}
							w_cg[cg] = w << (4u * (lane >> 5)); // frustum 4 g + 2 h + q: bit 8 g + 4 h + 2 q
							if constexpr (SLOTS) __builtin_amdgcn_sched_barrier(0); // (the variant that also carries the slots: one column group's accumulators at a time, or one register spills)
						}
						// [own column group | the other one] back: lanes < 32 own group 0's bits, lanes >= 32 group 1's; the partner lane holds the other four frusta
						const auto sw_w = __builtin_amdgcn_permlane32_swap(w_cg[0], w_cg[1], false, false);
						culled2[i] = sw_w[0] | sw_w[1];
						if (unsure != 0) exact_chunks |= 1u << i;
					}
				}
			}
			// Several frusta over the same spheres (the frame's shadow cascades, BASELINE config 5). Rounds 2 / 3 ran the single-frustum body
			// per (chunk, frustum): class from LDS -> wait -> branch -> frustum normals from the kernarg segment + distances from LDS ->
			// wait -> test, i.e. two serialized waits and an 18-dword scalar load per 64 spheres and frustum - 224 us for 10 M spheres
			// x 8 frusta, 0.13 of the VALU rate (profiles/r03). Now, per frustum: the normals are read ONCE into SGPRs, the (cell, frustum)
			// records of all the group's chunks are fetched together (one LDS round trip), the chunks are tested back to back without a
			// branch on a lane's class (the class selects the verdict), and the per-frustum counts are kept packed per lane
			// (10 bits a frustum) and summed over the wave once, behind the loop.
			// (Measured and NOT kept, round 4, profiles/r04/cull8_call27_28_shared_dots_batched_lds.txt: frusta with bitwise identical plane
			// normals - the cascades of one light - sharing a sphere's rounded dot products, 7 instead of 21 packed operations per plane set:
			// 154 us against 142 on the all-test scene - 24 more live VGPRs, a fifth block per CU only with spills (205 us), four blocks 166 us;
			// all of a frustum's cell records read in one batch instead of two chunks at a time: 149 against 142.)
			if constexpr (PRETEST) {
#pragma unroll
				for (int i = 0; i < GRP; ++i) culled2[i] = (exact_chunks >> i) & 1u ? 0u : culled2[i]; // (wave-uniform: an undecided chunk starts over)
			}
#pragma unroll 1
			for (int f = 0; f < (exact_chunks != 0 ? nf : 0); ++f) {
				const uint32_t st = (st_bits >> (2 * f)) & 3u;
				if (st == TILE_REJECT) continue; // nothing of this tile is visible in frustum f
				const bool mixed = any_mixed && st == TILE_MIXED; // wave-uniform
				// what this wave's chunks hold for frustum f, from the class words (wave-uniform): nothing at all -> next frustum; the chunks with
				// a lane in a CELL_TEST cell (bit i) -> only those are tested, only for them the cell records are read
				if (__ballot(((word_or >> (2 * f)) & 3u) != 0) == 0) continue;
				uint32_t test_chunks = 0;
				if (mixed) {
#pragma unroll
					for (int i = 0; i < GRP; ++i) {
						if (need_sphere(i) && ((exact_chunks >> i) & 1u) && __ballot(((cls_word[i] >> (2 * f)) & 3u) == CELL_TEST) != 0) test_chunks |= 1u << i;
					}
				}
				// the frustum's plane normals out of LDS (phase 0 put them there), every lane the same address: a broadcast read. (From the
				// kernarg segment they are scalar loads that share lgkmcnt with the LDS reads below and return out of order: the compiler
				// waited for the one before it issued the other.)
				float4 nq[5]; // nx[0..5] ny[0..5] nz[0..5] + 2 pad
				if (test_chunks != 0) {
#pragma unroll
					for (int k = 0; k < 5; ++k) nq[k] = reinterpret_cast<const float4*>(s_nrm[f])[k];
				}
				const float* nrm = reinterpret_cast<const float*>(nq);
#pragma unroll
				for (int h = 0; h < GRP; h += 2) { // two chunks' cell records in flight at a time
					float dq[2][6]; // d[0..5] of the lane's cell
					if ((test_chunks >> h) & 3u) {
#pragma unroll
						for (int j = 0; j < 2; ++j) {
							const v2f* rec = s_d2 + (size_t)(f * a.cell_cap + local[h + j]) * 3; // three ds_read_b64
							const v2f d01 = rec[0], d23 = rec[1], d45 = rec[2];
							dq[j][0] = d01.x; dq[j][1] = d01.y; dq[j][2] = d23.x; dq[j][3] = d23.y; dq[j][4] = d45.x; dq[j][5] = d45.y;
						}
					}
#pragma unroll
					for (int j = 0; j < 2; ++j) {
						const int i = h + j;
						uint32_t culled = 0;
						if ((test_chunks >> i) & 1u) { // wave-uniform: some lane of the chunk is in a CELL_TEST cell of THIS frustum
							// doCulling (culling_system.cpp:283-306), the operations of sphere_visible_d_pk: t = ((x*nx + y*ny) + z*nz) + d, t + r < 0 culls
							// (the {c, c} operands as shuffles of the loaded register pairs: the broadcast folds into op_sel of v_pk_*_f32 - as explicit
							// pairs they were 8 instead of 4 VGPRs per chunk and six v_mov each; the empty asm keeps the shuffles inside the frustum
							// loop: hoisted, they become the explicit pairs again)
							asm volatile("" : "+v"(spv[i]));
							const v2f xy = __builtin_shufflevector(spv[i], spv[i], 0, 1), zw = __builtin_shufflevector(spv[i], spv[i], 2, 3);
							const v2f x2 = __builtin_shufflevector(xy, xy, 0, 0), y2 = __builtin_shufflevector(xy, xy, 1, 1);
							const v2f z2 = __builtin_shufflevector(zw, zw, 0, 0), r2 = __builtin_shufflevector(zw, zw, 1, 1);
#pragma unroll
							for (int k = 0; k < 6; k += 2) {
								const v2f n_x = {nrm[k], nrm[k + 1]}, n_y = {nrm[6 + k], nrm[7 + k]}, n_z = {nrm[12 + k], nrm[13 + k]}, dd = {dq[j][k], dq[j][k + 1]};
								v2f t = x2 * n_x;
								t = t + y2 * n_y;
								t = t + z2 * n_z;
								t = t + dd;
								t = t + r2;
								culled |= (t.x < 0 ? 1u : 0u) | (t.y < 0 ? 1u : 0u); // (no short circuit: straight-line code)
							}
							culled2[i] |= culled << (2 * f);
						}
					}
				}
			}
			// the verdicts of all frusta at once, per chunk, from the class words: visible = CELL_TEST and not culled, or CELL_ACCEPT (a frustum
			// the tile test rejected, a cell classified CELL_REJECT: 0). Per (chunk, frustum) this was a class extract, two selects, a shift-or and
			// a count - a quarter of the loop's instructions on a launch that is bound by their issue (profiles/r04/cull8_probe_counters.txt).
			static_assert(CELL_REJECT == 0 && CELL_ACCEPT == 1 && CELL_TEST == 2, "bit 0 of a class: accepted, bit 1: to be tested");
#pragma unroll
			for (int i = 0; i < GRP; ++i) {
				const uint32_t w = cls_word[i];
				const uint32_t v = (((w >> 1) & ~culled2[i]) | w) & 0x5555u;
				vis2[g + i] = id[g + i] >= 0 ? v : 0u;
			}
		} else {
			// One frustum. The lanes' classes are in registers since the header batch; the cell records (six plane distances each) of the chunks that
			// test are read DB chunks at a time, ahead of the arithmetic, so that no test waits for LDS behind the wait for its sphere (round 5: class
			// read -> wait -> branch -> record read + 16-dword scalar load of the planes -> wait, per chunk: with every wave of a SIMD in that
			// pattern the vector unit idled half the time on a launch whose loads were free, profiles/r06/cull1_probes.txt). The test runs on every
			// lane of a chunk that has a CELL_TEST lane (the records of the other cells hold zeros) and the class selects the verdict: no divergent branch.
			constexpr int DB = GRP < CHW ? 2 : 1; // records in flight (six registers per chunk; the latency form has all eight chunks' spheres live)
			// the frustum's plane normals, read ONCE per group and pinned in scalar registers (the compiler otherwise re-reads all 18 from the kernarg
			// segment in front of every chunk's arithmetic - cheap to issue, but a scalar-cache round trip that nothing hides once the spheres are in)
			DevFrustum fn; // (only nx / ny / nz are used)
			if (need_sphere_bits != 0) {
#pragma unroll
				for (int k = 0; k < 6; ++k) {
					fn.nx[k] = frp[0].nx[k]; fn.ny[k] = frp[0].ny[k]; fn.nz[k] = frp[0].nz[k];
					asm volatile("" : LMX_ASM_SGPR(fn.nx[k]), LMX_ASM_SGPR(fn.ny[k]), LMX_ASM_SGPR(fn.nz[k]));
				}
			}
			static_assert(CELL_REJECT == 0 && CELL_ACCEPT == 1 && CELL_TEST == 2, "class encoding");
#pragma unroll
			for (int h = 0; h < GRP; h += DB) {
				v4f d03[DB];
				v2f d45[DB];
#pragma unroll
				for (int j = 0; j < DB; ++j) {
					// (unconditional: a lane's offset is always that of a record of this tile - of record 0 on a tile without classified cells - and what a
					// chunk without a CELL_TEST lane reads is never looked at; behind `if (need_sphere)` the other arm's zeros were six moves per chunk)
					const char* ci = reinterpret_cast<const char*>(s_info) + (cls_word[h + j] & ~31u);
					d03[j] = *reinterpret_cast<const v4f*>(ci); // d[0..3]: ds_read_b128, d[4..5]: ds_read_b64
					d45[j] = *reinterpret_cast<const v2f*>(ci + 16);
				}
#pragma unroll
				for (int j = 0; j < DB; ++j) {
					const int i = h + j;
					if (!need_id(i)) continue; // wave-uniform
					bool vis = (cls_word[i] & 3u) == CELL_ACCEPT;
					if (need_sphere(i)) {
						const float d[6] = {d03[j].x, d03[j].y, d03[j].z, d03[j].w, d45[j].x, d45[j].y};
						const bool pass = sphere_visible_d_pk(fn, d, sp[i].x, sp[i].y, sp[i].z, sp[i].w);
						vis = vis || ((cls_word[i] & 3u) == CELL_TEST && pass);
					}
					vis = vis && id[g + i] >= 0;
					if constexpr (STAGE) {
						const uint64_t mask = __ballot(vis);
						if (vis) {
							s_stage_ids[wave][staged + mbcnt64(mask)] = id[g + i];
							if constexpr (SLOTS) s_stage_slots[wave][staged + mbcnt64(mask)] = (int32_t)(((chunk0 + g + i) << 6) + lane);
						}
						staged += (uint32_t)__popcll(mask);
					} else {
						const uint32_t c = (uint32_t)__popcll(__ballot(vis));
						mine += lane == 0u ? c : 0u;
						vis_bits |= (vis ? 1u : 0u) << ((g + i) * FS);
					}
				}
			}
		}
	}
#undef need_id
#undef need_sphere
	if constexpr (F != 1) {
		// Per-frustum counts of the wave, lane f keeps frustum f's. A lane's four chunks are summed in packed fields first (a verdict bit per
		// 2-bit field: two words of sums <= 2, then nibbles <= 4 for the even and the odd frusta, spread to 16-bit fields), the four words are
		// summed over the wave by four DPP steps inside every row of 16 lanes + one v_readlane per row. (Per frustum and chunk a ballot and a
		// scalar population count - 32 of each per wave, with their extract / compare - were ~440 of a wave's ~2400 instructions.)
		static_assert(CHW == 4, "the packed sums hold four chunks");
		const uint32_t s01 = vis2[0] + vis2[1], s23 = vis2[2] + vis2[3];
		const uint32_t ev = (s01 & 0x3333u) + (s23 & 0x3333u), od = ((s01 >> 2) & 0x3333u) + ((s23 >> 2) & 0x3333u); // nibble j: frustum 2 j / 2 j + 1
		uint32_t word[4] = {(ev & 0xfu) | ((ev & 0xf0u) << 12), ((ev >> 8) & 0xfu) | ((ev & 0xf000u) << 4),   // {f0 | f2 << 16}, {f4 | f6 << 16}
			(od & 0xfu) | ((od & 0xf0u) << 12), ((od >> 8) & 0xfu) | ((od & 0xf000u) << 4)};                  // {f1 | f3 << 16}, {f5 | f7 << 16}
		uint32_t tot[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			uint32_t v = word[k];
			v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);  // quad_perm [1, 0, 3, 2]: + lane ^ 1
			v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);  // quad_perm [2, 3, 0, 1]: + lane ^ 2 -> the quad's sum
			v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true); // row_half_mirror: + the other quad of the 8 lanes
			v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true); // row_mirror: + the other half of the row of 16
			tot[k] = (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) + (uint32_t)__builtin_amdgcn_readlane((int)v, 32) +
				(uint32_t)__builtin_amdgcn_readlane((int)v, 48);
		}
		// frustum f: word (f & 1) * 2 + (f >> 2), field (f >> 1) & 1
#pragma unroll
		for (int f = 0; f < MAX_FRUSTA; ++f) {
			const uint32_t c = (tot[(f & 1) * 2 + (f >> 2)] >> (16 * ((f >> 1) & 1))) & 0xffffu;
			mine = lane == (uint32_t)f ? c : mine;
		}
	}

	if constexpr (STAGE) {
		// C (1 frustum). one reservation for the wave, then the compacted ids leave LDS 64 lanes at a time
		if (staged == 0) return;
		uint32_t base = 0;
		if (lane == 0) base = atomicAdd(&g_counts[shard * a.cnt_pad], staged);
		base = __builtin_amdgcn_readfirstlane(base) + win;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the wave's own LDS writes above: issue order is execution order, only the compiler must not move the reads up)
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		for (uint32_t k = lane; k < staged; k += 64u) {
			g_out_ids[base + k] = s_stage_ids[wave][k];
			if constexpr (SLOTS) a.out_slots[base + k] = s_stage_slots[wave][k];
		}
		return;
	}
	// C. reserve: lane f adds this wave's count for frustum f to the shard's counter (one atomic instruction for all frusta),
	// then the ids go from registers to the reserved ranges in chunk order
	uint32_t base_v = 0;
	if (lane < (uint32_t)nf && mine != 0) base_v = atomicAdd(&g_counts[lane * a.cnt_frustum_stride + shard * a.cnt_pad], mine);
#pragma unroll 1
	for (int f = 0; f < nf; ++f) {
		if (__builtin_amdgcn_readlane(mine, f) == 0) continue;
		uint32_t run = __builtin_amdgcn_readlane(base_v, f) + win;
		int32_t* dst = g_out_ids + (size_t)f * a.out_stride;
#pragma unroll
		for (int i = 0; i < CHW; ++i) {
			const bool v = F != 1 ? ((vis2[F != 1 ? i : 0] >> (2 * f)) & 1u) != 0 : ((vis_bits >> (i * FS + f)) & 1u) != 0;
			const uint64_t mask = __ballot(v);
			if (v) dst[run + mbcnt64(mask)] = id[i];
			if constexpr (SLOTS) {
				if (v) (a.out_slots + (size_t)f * a.out_stride)[run + mbcnt64(mask)] = (int32_t)(((chunk0 + i) << 6) + lane);
			}
			run += (uint32_t)__popcll(mask);
		}
	}
}

// ---- dynamic set ------------------------------------------------------------------------------------------------
// One thread per unsorted entity: cell index, is_big and the cell-relative fp32 position are derived from the fp64 world
// position exactly like CullingSystemImpl::add / set would (culling_system.cpp:26-30,100,140), the cell is classified per
// lane (no sharing between lanes: the set is not sorted) and the sphere is tested. ~350 VALU ops per entity and frustum,
// 36 B per entity: the VALU-heavier, bandwidth-lighter sibling of the sorted path, used for entities that move every frame
// and for entities added since the static set was last compacted.
constexpr int DYN_THREADS = 256;

// A block handles TILE entities in TILE / 256 batches and stages the visible ids of the whole tile in LDS, so that a shard
// counter sees one atomic per (tile, frustum): the set is unsorted, visible entities are spread thinly over every block.
template <int TILE>
__global__ __launch_bounds__(DYN_THREADS) void k_cull_dynamic(const double* __restrict__ px, const double* __restrict__ py,
	const double* __restrict__ pz, const float* __restrict__ radius, const int32_t* __restrict__ ids, FrustaArg fr, int n_frusta,
	TypeTable dyn_tt, uint32_t slot_begin, CullOut out) {
	LMX_DYNAMIC_LDS(int32_t, s_stage); // [n_frusta][TILE] staged ids | [MAX_FRUSTA] counts | [MAX_FRUSTA] bases
	uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_stage + n_frusta * TILE);
	uint32_t* s_base = s_cnt + MAX_FRUSTA;
	const uint32_t lane = lane_id();
	const uint32_t block_slot = slot_begin + blockIdx.x * (uint32_t)TILE;
	uint32_t type = 0;
#pragma unroll
	for (int t = 0; t < MAX_TYPES; ++t) {
		if (block_slot >= dyn_tt.ent_start[t] && block_slot < dyn_tt.ent_end[t]) type = t;
	}
	const uint32_t shard = dyn_tt.shard_first[type] + ((block_slot - dyn_tt.ent_start[type]) / 2048u) % dyn_tt.shard_n[type];
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
		s_base[threadIdx.x] = c ? atomicAdd(&out.counts[threadIdx.x * out.cnt_frustum_stride + shard * out.cnt_pad], c) : 0u;
	}
	__syncthreads();
	const uint32_t win = out.win_base[shard];
	for (int f = 0; f < n_frusta; ++f) {
		const uint32_t c = s_cnt[f];
		int32_t* dst = out.ids + (size_t)f * out.stride + win + s_base[f];
		for (uint32_t k = threadIdx.x; k < c; k += DYN_THREADS) dst[k] = s_stage[f * TILE + k];
		if (out.slots != nullptr) { // ids of the dynamic set carry no static slot
			int32_t* sdst = out.slots + (size_t)f * out.stride + win + s_base[f];
			for (uint32_t k = threadIdx.x; k < c; k += DYN_THREADS) sdst[k] = -1;
		}
	}
}

// ---- patches ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_apply_patches(float4* __restrict__ spheres, int32_t* __restrict__ ids, TileBox* __restrict__ box0,
	TileBox* __restrict__ box1, TileBox* __restrict__ box2, DynDeviceView d, const PatchSphere* __restrict__ ps, uint32_t n_ps,
	const PatchId* __restrict__ pi, uint32_t n_pi, const PatchDyn* __restrict__ pd, uint32_t n_pd) {
	uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n_ps) {
		const PatchSphere p = ps[i];
		spheres[p.slot] = make_float4(p.x, p.y, p.z, p.radius);
		return;
	}
	i -= n_ps;
	if (i < n_pi) {
		const PatchId p = pi[i];
		ids[p.slot] = p.id;
		if (p.id < 0) { // a tombstone: the tiles holding the slot are no longer a straight copy when accepted
			atomicAnd(&box0[p.slot >> 12].flags, ~(uint32_t)TILE_DENSE);
			atomicAnd(&box1[p.slot >> 11].flags, ~(uint32_t)TILE_DENSE);
			atomicAnd(&box2[p.slot >> 10].flags, ~(uint32_t)TILE_DENSE);
		}
		return;
	}
	i -= n_pi;
	if (i < n_pd) {
		const PatchDyn p = pd[i];
		d.px[p.slot] = p.px;
		d.py[p.slot] = p.py;
		d.pz[p.slot] = p.pz;
		d.radius[p.slot] = p.radius;
		d.ids[p.slot] = p.id;
	}
}

// ---- finalize / consolidate ---------------------------------------------------------------------------------------
// One block per frustum: totals per type and the exclusive prefix of every shard inside its type. The counters sit on separate
// cache lines, so they are loaded in parallel (one per thread) and scanned in LDS; a lane per type walking them serially took
// 21 us for 65 shards. Shards of one type are contiguous in shard order. packed_start (optional): where type t starts when the
// types of a frustum are packed back to back (type 0 first) - the layout of the exchange's send buffer.
constexpr int FIN_THREADS = 256;
constexpr int FIN_MAX_SHARDS = 1024; // 8 types x (64 static + 8 dynamic) = 576 at most
__global__ __launch_bounds__(FIN_THREADS) void k_cull_finalize(const uint32_t* __restrict__ counts, uint32_t cnt_pad, uint32_t cnt_frustum_stride,
	const uint8_t* __restrict__ shard_type, uint32_t n_shards, uint32_t* __restrict__ totals, uint32_t* __restrict__ pref, uint32_t* __restrict__ packed_start) {
	__shared__ uint32_t s_cnt[FIN_MAX_SHARDS];
	__shared__ uint8_t s_type[FIN_MAX_SHARDS];
	__shared__ uint32_t s_tot[MAX_TYPES];
	const uint32_t f = blockIdx.x;
	const uint32_t t = threadIdx.x;
	if (t < MAX_TYPES) s_tot[t] = 0;
	for (uint32_t s = t; s < n_shards; s += FIN_THREADS) {
		s_cnt[s] = counts[f * cnt_frustum_stride + s * cnt_pad];
		s_type[s] = shard_type[s];
	}
	__syncthreads();
	// per shard: sum of the counts of the earlier shards of its type (<= 72 of them, contiguous, in LDS)
	for (uint32_t s = t; s < n_shards; s += FIN_THREADS) {
		const uint8_t ty = s_type[s];
		uint32_t run = 0;
		for (uint32_t k = s; k-- > 0 && s_type[k] == ty;) run += s_cnt[k];
		pref[f * n_shards + s] = run;
		if (s + 1 == n_shards || s_type[s + 1] != ty) s_tot[ty] = run + s_cnt[s]; // the last shard of a type knows the total
	}
	__syncthreads();
	if (t < MAX_TYPES) {
		totals[f * MAX_TYPES + t] = s_tot[t];
		if (packed_start != nullptr) {
			uint32_t at = 0;
			for (uint32_t k = 0; k < t; ++k) at += s_tot[k];
			packed_start[f * MAX_TYPES + t] = at;
		}
	}
}

// grid (n_shards, n_frusta, splits): copies shard s of frustum f to its place in the type's contiguous list. type_start is
// indexed [f * type_start_stride + type] (stride 0: the same capacity-based starts for every frustum; MAX_TYPES: packed starts).
// Ids that would land at or beyond dst_cap (per frustum row) are dropped: a fixed-size send buffer reports the overflow through
// its totals instead.
__global__ __launch_bounds__(256) void k_cull_consolidate(const int32_t* __restrict__ src, uint32_t src_stride, const uint32_t* __restrict__ win_base,
	const uint32_t* __restrict__ counts, uint32_t cnt_pad, uint32_t cnt_frustum_stride, const uint8_t* __restrict__ shard_type,
	const uint32_t* __restrict__ type_start, uint32_t type_start_stride, const uint32_t* __restrict__ pref, uint32_t n_shards, int32_t* __restrict__ dst,
	uint32_t dst_stride, uint32_t dst_cap, const int32_t* __restrict__ src2 /* optional: a second array with the same windows (the visible ids' slots) */,
	int32_t* __restrict__ dst2) {
	const uint32_t s = blockIdx.x, f = blockIdx.y;
	const uint32_t c = counts[f * cnt_frustum_stride + s * cnt_pad];
	const size_t from_at = (size_t)f * src_stride + win_base[s];
	const uint32_t at = type_start[f * type_start_stride + shard_type[s]] + pref[f * n_shards + s];
	const size_t to_at = (size_t)f * dst_stride + at;
	const uint32_t room = at < dst_cap ? dst_cap - at : 0u;
	const uint32_t n = c < room ? c : room;
	for (uint32_t k = blockIdx.z * 256u + threadIdx.x; k < n; k += gridDim.z * 256u) {
		dst[to_at + k] = src[from_at + k];
		if (src2 != nullptr) dst2[to_at + k] = src2[from_at + k];
	}
}

// One launch instead of finalize + consolidate for the PACKED record [MAX_TYPES counts | ids, types back to back] of ONE frustum
// (the exchange's send buffer, lmx_cull_map_all's host record). Shards are ordered by type (recompute_out_layout), so a shard's
// place in the packed list is the plain exclusive prefix of the counts of ALL shards before it: every block sums those <= 575
// counters itself (one per thread, separate cache lines, L2 hits) instead of waiting for a one-block scan kernel - 7 us of GPU time
// and one launch less per frame. Block (0, 0) also writes the per-type totals. grid (n_shards, splits, frusta): the frusta of one cull
// (a frame's views, the exchange's sub-records) are packed by ONE launch, frustum f reading row f of the ids / counters and writing
// record f (strides in words).
// Record f = [MAX_TYPES counts | lay.cap[f] ids] at rec_base + lay.off[f]: the records of a launch need neither the same capacity nor a
// common stride (the exchange's sub-records carry per-frustum capacities, lmx_capi_exchange.hip).
__global__ __launch_bounds__(256) void k_cull_pack(const int32_t* __restrict__ src, const uint32_t* __restrict__ win_base, const uint32_t* __restrict__ counts,
	uint32_t cnt_pad, const uint8_t* __restrict__ shard_type, uint32_t n_shards, int32_t* __restrict__ rec_base, PackLayout lay,
	uint32_t src_stride, uint32_t cnt_stride) {
	__shared__ uint32_t s_part[4];
	__shared__ uint32_t s_tot[MAX_TYPES];
	const uint32_t s = blockIdx.x, t = threadIdx.x;
	src += (size_t)blockIdx.z * src_stride;
	counts += (size_t)blockIdx.z * cnt_stride;
	uint32_t* const header = reinterpret_cast<uint32_t*>(rec_base + lay.off[blockIdx.z]);
	int32_t* const dst = rec_base + lay.off[blockIdx.z] + MAX_TYPES;
	const uint32_t dst_cap = lay.cap[blockIdx.z];
	const bool totals_block = s == 0 && blockIdx.y == 0;
	const uint32_t c = counts[s * cnt_pad];
	// a block whose slice of the shard's window is empty has nothing to place (most of them when little is visible: the grid is sized
	// for a full window), and the others fetch their first ids now, under the prefix of the counters instead of behind it
	if (blockIdx.y * 256u >= c && !totals_block) return;
	const int32_t* from = src + win_base[s];
	const uint32_t k0 = blockIdx.y * 256u + t;
	const int32_t first = k0 < c ? from[k0] : 0;
	uint32_t before = 0;
	for (uint32_t k = t; k < s; k += 256u) before += counts[k * cnt_pad]; // (<= 3 iterations)
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) before += (uint32_t)__shfl_xor((int)before, o);
	if ((t & 63u) == 0) s_part[t >> 6] = before;
	if (totals_block && t < MAX_TYPES) s_tot[t] = 0;
	__syncthreads();
	const uint32_t at = s_part[0] + s_part[1] + s_part[2] + s_part[3];
	if (totals_block) { // totals per type: what the rank saw, also beyond dst_cap
		for (uint32_t k = t; k < n_shards; k += 256u) atomicAdd(&s_tot[shard_type[k]], counts[k * cnt_pad]);
		__syncthreads();
		if (t < MAX_TYPES) header[t] = s_tot[t];
	}
	int32_t* to = dst + at;
	const uint32_t room = at < dst_cap ? dst_cap - at : 0u;
	const uint32_t n = c < room ? c : room;
	if (k0 < n) to[k0] = first;
	for (uint32_t k = k0 + gridDim.y * 256u; k < n; k += gridDim.y * 256u) to[k] = from[k];
}

template <int F, int WAVES, int CHW, int GRP, int FORM, int SLOTS_I>
hipError_t tile_f(hipStream_t s, const CullDeviceView& v, uint32_t ent_begin, uint32_t ent_end, const TypeTable& tt, const FrustaArg& fr, int n_frusta,
	const CullOut& out) {
	constexpr uint32_t TILE = WAVES * CHW * 64;
	static_assert(TILE_ALIGN % TILE == 0, "tiles must not straddle type ranges");
	constexpr int K = TILE == 4096 ? 0 : (TILE == 2048 ? 1 : 2);
	const uint32_t tiles = (ent_end - ent_begin) / TILE;
	if (!tiles) {
		if (out.ev_start != nullptr) { // keep the profiling pair valid
			(void)hipEventRecord(out.ev_start, s);
			(void)hipEventRecord(out.ev_stop, s);
		}
		return hipSuccess;
	}
	TileScalars a;
	(void)tt; // (the tiles' shards come out of CullDeviceView::tile_out)
	a.ent_begin = ent_begin;
	a.cell_cap = v.tile_cap[K];
	a.keys_packed = v.keys_packed ? 1u : 0u;
	a.n_frusta = (uint32_t)n_frusta;
	a.out_stride = out.stride;
	a.cnt_pad = out.cnt_pad;
	a.cnt_frustum_stride = out.cnt_frustum_stride;
	a.n_zero = out.n_zero;
	a.out_slots = out.slots;
	{ // the pre-test's bound scale: 2^27 x the largest |n|_1 of the call's planes, rounded up; anything not finite switches the pre-test off
		float n1 = 0.f;
		bool finite = true;
		for (int f = 0; f < n_frusta; ++f) {
			for (int k = 0; k < 6; ++k) {
				const float v = (__builtin_fabsf(fr.f[f].nx[k]) + __builtin_fabsf(fr.f[f].ny[k])) + __builtin_fabsf(fr.f[f].nz[k]);
				finite = finite && v <= 3.0e38f; // (false for NaN)
				n1 = v > n1 ? v : n1;
			}
		}
		a.pretest_n1s = finite ? n1 * 1.000001f * 134217728.0f : __builtin_inff();
	}
	if (out.ev_start != nullptr) // profiling: the events receive the dispatch's own begin / end timestamps
		hipExtLaunchKernelGGL((k_cull_tile<F, WAVES, CHW, GRP, FORM, SLOTS_I>), dim3(tiles), dim3(WAVES * 64), cull_tile_lds_bytes(n_frusta, a.cell_cap), s, out.ev_start, out.ev_stop, 0, fr,
			v.spheres, v.ids, v.hdr, v.tile_cells[K], v.tile_tab[K], v.tile_box[K], v.tile_out[K], out.ids, out.counts, out.counts_next, a);
	else
		hipLaunchKernelGGL((k_cull_tile<F, WAVES, CHW, GRP, FORM, SLOTS_I>), dim3(tiles), dim3(WAVES * 64), cull_tile_lds_bytes(n_frusta, a.cell_cap), s, fr, v.spheres, v.ids, v.hdr,
			v.tile_cells[K], v.tile_tab[K], v.tile_box[K], v.tile_out[K], out.ids, out.counts, out.counts_next, a);
	return hipGetLastError();
}

} // namespace

size_t cull_tile_lds_bytes(int n_frusta, uint32_t cell_cap) { // several frusta: 24-byte records + the cells' class words (5..8 frusta: in four 16-bit parts)
	return n_frusta > 1 ? (size_t)n_frusta * cell_cap * 24 + (size_t)cell_cap * (n_frusta > 4 ? 8 : 4) : (size_t)cell_cap * sizeof(CellInfo);
}

uint32_t cull_tile_size(int n_frusta, int variant) {
	(void)variant; // (both forms of the 1-frustum kernel walk 2048-sphere tiles)
	return n_frusta <= 4 ? 2048u : 1024u;
}

// The 1-frustum kernel exists in two forms, both 4 waves x 8 chunks (2048-sphere tiles) with the tile-level test on the lanes of wave 0:
// variant 1 streams (two groups of four chunks, non-temporal loads, LDS-staged ids), variant 4 keeps all eight chunks' loads in flight (the
// latency form: launches in which few tiles survive the tile-level test). Rounds 2-5 carried four more tile shapes and two more forms of
// the tile-level test that no selection rule ever picked: removed with their test parametrisations.
hipError_t launch_cull_tile(hipStream_t s, const CullDeviceView& v, uint32_t ent_begin, uint32_t ent_end, const TypeTable& tt, const FrustaArg& fr,
	int n_frusta, const CullOut& out, int variant) {
#define LMX_TILE(F, W, C, G, L) return out.slots ? tile_f<F, W, C, G, L, 1>(s, v, ent_begin, ent_end, tt, fr, n_frusta, out) : tile_f<F, W, C, G, L, 0>(s, v, ent_begin, ent_end, tt, fr, n_frusta, out)
	if (n_frusta < 1 || n_frusta > MAX_FRUSTA) return hipErrorInvalidValue;
	if (n_frusta == 1) {
		const int wide = v.keys_packed ? 0 : 2; // (the instantiations with 16-byte cell keys: scenes whose tiles span more than 65535 cells on an axis)
		if (variant == 4) { if (wide) LMX_TILE(1, 4, 8, 8, 3); LMX_TILE(1, 4, 8, 8, 1); }
		if (wide) LMX_TILE(1, 4, 8, 4, 4);
		LMX_TILE(1, 4, 8, 4, 2); // (all eight chunks in flight in the streaming form as well: 41.0 vs 37.9 us on the all-test launch, profiles/r06/cull1_ab_grp8.txt)
	}
	if (n_frusta <= 4) LMX_TILE(0, 8, 4, 4, 0); // 2048-sphere tiles, <= 4 x 32 B of LDS per cell
	LMX_TILE(0, 4, 4, 4, 0);                    // 1024-sphere tiles
#undef LMX_TILE
}

// Slots per block: at most 16 KiB of staging, and small enough that the launch has ~2000 blocks: a block walks its tile in batches
// of 256 one after the other (load -> ~350 VALU -> LDS append), so a small set on few large tiles is a handful of CUs each waiting
// on its own chain of loads (160 k overflow entities on 80 blocks: 14 us; on 640 blocks of 256: see DESIGN.md).
uint32_t cull_dynamic_tile(int n_frusta, uint32_t n_slots) {
	uint32_t tile = n_frusta <= 2 ? 2048u : (n_frusta <= 4 ? 1024u : 512u);
	while (tile > 256u && n_slots / tile < 2048u) tile >>= 1;
	return tile;
}

// Asynchronous compaction, at the swap: the spheres of the OLD dynamic set that still exist in the NEW one keep what the device last
// computed for them (bound entities are refreshed by k_sphere_refresh, not by the host). One thread per old slot; new_slot_of_entity is
// the new set's entity -> slot table as of the end of the worker's catch-up - an entry that went stale since then (the entity was
// removed, its slot given to another one) is caught by comparing the id the new slot holds NOW.
__global__ __launch_bounds__(256) void k_dyn_carry_over(const double* __restrict__ opx, const double* __restrict__ opy, const double* __restrict__ opz,
	const float* __restrict__ oradius, const int32_t* __restrict__ oids, uint32_t n_old, double* __restrict__ npx, double* __restrict__ npy, double* __restrict__ npz,
	float* __restrict__ nradius, const int32_t* __restrict__ nids, uint32_t n_new, const int32_t* __restrict__ new_slot_of_entity, uint32_t n_entities) {
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s >= n_old) return;
	const int32_t e = oids[s];
	if (e < 0 || (uint32_t)e >= n_entities) return;
	const int32_t t = new_slot_of_entity[e];
	if (t < 0 || (uint32_t)t >= n_new || nids[t] != e) return;
	npx[t] = opx[s];
	npy[t] = opy[s];
	npz[t] = opz[s];
	nradius[t] = oradius[s];
}

hipError_t launch_dyn_carry_over(hipStream_t s, const DynDeviceView& from, const DynDeviceView& to, const int32_t* new_slot_of_entity, uint32_t n_entities) {
	if (!from.n_padded || !to.n_padded || !n_entities) return hipSuccess;
	hipLaunchKernelGGL(k_dyn_carry_over, dim3((from.n_padded + 255u) / 256u), dim3(256), 0, s, from.px, from.py, from.pz, from.radius, from.ids, from.n_padded, to.px, to.py,
		to.pz, to.radius, to.ids, to.n_padded, new_slot_of_entity, n_entities);
	return hipGetLastError();
}

hipError_t launch_cull_dynamic(hipStream_t s, const DynDeviceView& d, uint32_t slot_begin, uint32_t slot_end, const TypeTable& dyn_tt,
	const FrustaArg& fr, int n_frusta, const CullOut& out) {
	const uint32_t tile = cull_dynamic_tile(n_frusta, slot_end - slot_begin);
	const uint32_t blocks = (slot_end - slot_begin) / tile;
	if (!blocks) return hipSuccess;
	const size_t lds = (size_t)n_frusta * tile * sizeof(int32_t) + 2 * MAX_FRUSTA * sizeof(uint32_t);
#define LMX_DYN(T) hipLaunchKernelGGL(k_cull_dynamic<T>, dim3(blocks), dim3(DYN_THREADS), lds, s, d.px, d.py, d.pz, d.radius, d.ids, fr, n_frusta, dyn_tt, slot_begin, out)
	if (tile == 2048) LMX_DYN(2048);
	else if (tile == 1024) LMX_DYN(1024);
	else if (tile == 512) LMX_DYN(512);
	else LMX_DYN(256);
#undef LMX_DYN
	return hipGetLastError();
}

hipError_t launch_apply_patches(hipStream_t s, float4* spheres, int32_t* ids, TileBox* const tile_box[3], const DynDeviceView& d, const PatchSphere* ps,
	uint32_t n_ps, const PatchId* pi, uint32_t n_pi, const PatchDyn* pd, uint32_t n_pd) {
	const uint32_t n = n_ps + n_pi + n_pd;
	if (!n) return hipSuccess;
	hipLaunchKernelGGL(k_apply_patches, dim3((n + 255u) / 256u), dim3(256), 0, s, spheres, ids, tile_box[0], tile_box[1], tile_box[2], d, ps, n_ps, pi, n_pi, pd, n_pd);
	return hipGetLastError();
}

hipError_t launch_cull_finalize(hipStream_t s, const uint32_t* counts, uint32_t cnt_pad, uint32_t cnt_frustum_stride, const uint8_t* shard_type,
	uint32_t n_shards, uint32_t n_frusta, uint32_t* totals, uint32_t* pref, uint32_t* packed_start) {
	if (!n_frusta) return hipSuccess;
	if (n_shards > (uint32_t)FIN_MAX_SHARDS) return hipErrorInvalidValue;
	hipLaunchKernelGGL(k_cull_finalize, dim3(n_frusta), dim3(FIN_THREADS), 0, s, counts, cnt_pad, cnt_frustum_stride, shard_type, n_shards, totals, pref, packed_start);
	return hipGetLastError();
}

hipError_t launch_cull_consolidate(hipStream_t s, const int32_t* src, uint32_t src_stride, const uint32_t* win_base, const uint32_t* counts,
	uint32_t cnt_pad, uint32_t cnt_frustum_stride, const uint8_t* shard_type, const uint32_t* type_start, uint32_t type_start_stride, const uint32_t* pref,
	uint32_t n_shards, uint32_t n_frusta, uint32_t max_shard_cap, int32_t* dst, uint32_t dst_stride, uint32_t dst_cap, const int32_t* src2, int32_t* dst2) {
	if (!n_frusta || !n_shards) return hipSuccess;
	const uint32_t splits = std::max(1u, std::min(64u, max_shard_cap / 4096u));
	hipLaunchKernelGGL(k_cull_consolidate, dim3(n_shards, n_frusta, splits), dim3(256), 0, s, src, src_stride, win_base, counts, cnt_pad, cnt_frustum_stride,
		shard_type, type_start, type_start_stride, pref, n_shards, dst, dst_stride, dst_cap, src2, dst2);
	return hipGetLastError();
}

hipError_t launch_cull_pack_layout(hipStream_t s, const int32_t* src, const uint32_t* win_base, const uint32_t* counts, uint32_t cnt_pad, const uint8_t* shard_type,
	uint32_t n_shards, uint32_t max_shard_cap, int32_t* rec_base, const PackLayout& lay, uint32_t n_frusta, uint32_t src_stride, uint32_t cnt_stride) {
	if (!n_frusta) return hipSuccess;
	if (n_frusta > (uint32_t)MAX_FRUSTA) return hipErrorInvalidValue;
	if (!n_shards) { // an empty set still reports its (zero) counts
		for (uint32_t f = 0; f < n_frusta; ++f)
			if (hipError_t e = hipMemsetAsync(rec_base + lay.off[f], 0, MAX_TYPES * sizeof(uint32_t), s)) return e;
		return hipSuccess;
	}
	if (n_shards > (uint32_t)FIN_MAX_SHARDS) return hipErrorInvalidValue;
	const uint32_t splits = std::max(1u, std::min(64u, max_shard_cap / 4096u));
	hipLaunchKernelGGL(k_cull_pack, dim3(n_shards, splits, n_frusta), dim3(256), 0, s, src, win_base, counts, cnt_pad, shard_type, n_shards, rec_base, lay, src_stride,
		cnt_stride);
	return hipGetLastError();
}

hipError_t launch_cull_pack(hipStream_t s, const int32_t* src, const uint32_t* win_base, const uint32_t* counts, uint32_t cnt_pad, const uint8_t* shard_type,
	uint32_t n_shards, uint32_t max_shard_cap, uint32_t* header, int32_t* dst, uint32_t dst_cap, uint32_t n_frusta, uint32_t src_stride, uint32_t cnt_stride,
	uint32_t rec_stride) {
	if (n_frusta > (uint32_t)MAX_FRUSTA || dst != reinterpret_cast<int32_t*>(header) + MAX_TYPES) return hipErrorInvalidValue; // (a record's ids sit behind its header)
	PackLayout lay = {};
	for (uint32_t f = 0; f < n_frusta; ++f) {
		lay.off[f] = (uint64_t)f * rec_stride;
		lay.cap[f] = dst_cap;
	}
	return launch_cull_pack_layout(s, src, win_base, counts, cnt_pad, shard_type, n_shards, max_shard_cap, reinterpret_cast<int32_t*>(header), lay, n_frusta, src_stride, cnt_stride);
}

} // namespace lmx
