// Development probe: what a PURE READ of k_cull_tile's streaming footprint costs on one MI355X - 16-byte spheres + 4-byte ids of N entities (+ optionally 4 bytes
// written per visible-fraction entity), cache-cold behind a 1 GiB scrub, in the access pattern of the 1-frustum kernel (block = 2048 consecutive entities, 4 waves x
// 8 chunks in two groups of four, non-temporal loads) and with nothing else in the kernel: the launch-sized ceiling `roofline.frac` of bench.py can be held against
// besides the 8 TB/s peak and the 1 GiB copy. Also: the same bytes by a grid-stride kernel (resident blocks only), to tell the pattern's share from the chip's.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/read_probe.hip -o tools/_build/read_probe ; tools/_build/read_probe [entities]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int NT, int GRP> __global__ __launch_bounds__(256) void read_tiles(const v4f* __restrict__ sp, const int* __restrict__ ids, int* __restrict__ out, float thresh) {
	const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const size_t chunk0 = (size_t)blockIdx.x * 32 + wave * 8;
	float acc = 0.f; int iacc = 0;
#pragma unroll
	for (int g = 0; g < 8; g += GRP) {
		v4f s[GRP]; int id[GRP];
#pragma unroll
		for (int i = 0; i < GRP; ++i) {
			const size_t e = ((chunk0 + g + i) << 6) + lane;
			if (NT) { s[i] = __builtin_nontemporal_load(sp + e); id[i] = __builtin_nontemporal_load(ids + e); } else { s[i] = sp[e]; id[i] = ids[e]; }
		}
#pragma unroll
		for (int i = 0; i < GRP; ++i) { acc += s[i].x + s[i].y + s[i].z + s[i].w; iacc ^= id[i]; }
		__builtin_amdgcn_sched_barrier(0);
	}
	if (acc + (float)iacc == thresh) out[blockIdx.x * 256 + threadIdx.x] = iacc; // never - and BOTH sums feed the condition (with `acc == thresh` alone the compiler moves the id loads into the branch: 16 bytes per entity read, not 20)
}
// the same read with pieces of k_cull_tile's skeleton around it (FEAT bits): 1 = prologue (wave 0 loads the tile's 32-byte box, ~40 dependent VALU operations, verdict through
// LDS + barrier, then an 8-byte per-tile table entry), 2 = a 16-byte scalar header load + an LDS read per chunk IN FRONT of the chunk's loads, 4 = epilogue (per wave: count 1/32
// of the lanes, ONE returning atomic on one of 64 padded counters, the counted ids stored), 8 = 24 bytes of LDS reads per chunk + 21 packed multiply / add per chunk behind the loads
template <int FEAT> __global__ __launch_bounds__(256) void read_feat(const v4f* __restrict__ sp, const int* __restrict__ ids, const int4* __restrict__ box, const uint2* __restrict__ tab,
	const int4* __restrict__ hdr, unsigned* __restrict__ counters, int* __restrict__ out, float thresh) {
	__shared__ float s_rec[2048];
	__shared__ unsigned s_verdict;
	const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	unsigned shard = blockIdx.x & 63u, win = 0;
	for (unsigned k = threadIdx.x; k < 2048; k += 256) s_rec[k] = (float)k;
	__shared__ unsigned s_arrived0;
	if (threadIdx.x == 0) s_arrived0 = 0;
	if (FEAT & 1) {
		if (wave == 0) {
			const int4 b0 = box[blockIdx.x * 2], b1 = box[blockIdx.x * 2 + 1];
			float v = (float)(b0.x + b1.y) * 0.5f + (float)lane;
#pragma unroll
			for (int k = 0; k < 40; ++k) v = v * 1.0001f + (float)b0.z;
			if (lane == 0) s_verdict = v == thresh ? 0u : 2u;
		}
		__syncthreads();
		if (s_verdict == 0u) return;
		const uint2 t = tab[blockIdx.x];
		shard = t.x & 63u; win = t.y;
	} else __syncthreads();
	const size_t chunk0 = (size_t)blockIdx.x * 32 + wave * 8;
	float acc = 0.f; unsigned staged = 0; int keep[8];
#pragma unroll
	for (int g = 0; g < 8; g += 4) {
		v4f s[4]; int id[4]; unsigned rec[4];
#pragma unroll
		for (int i = 0; i < 4; ++i) rec[i] = (lane * 8u) & 2047u;
		if (FEAT & 2) {
			int4 h[4];
#pragma unroll
			for (int i = 0; i < 4; ++i) h[i] = hdr[chunk0 + g + i];
#pragma unroll
			for (int i = 0; i < 4; ++i) rec[i] = (unsigned)(__builtin_amdgcn_readfirstlane(h[i].x) + __builtin_amdgcn_mbcnt_hi((unsigned)h[i].z, __builtin_amdgcn_mbcnt_lo((unsigned)h[i].y, 0u))) * 8u & 2040u;
#pragma unroll
			for (int i = 0; i < 4; ++i) rec[i] = (rec[i] & ~7u) | ((unsigned)s_rec[rec[i] + 6] & 0u); // (the class read the loads wait for)
		}
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const size_t e = ((chunk0 + g + i) << 6) + lane;
			s[i] = __builtin_nontemporal_load(sp + e); id[i] = __builtin_nontemporal_load(ids + e);
		}
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			bool vis;
			if (FEAT & 8) {
				const v4f d03 = *reinterpret_cast<const v4f*>(&s_rec[rec[i] & 2040u]);
				const v2f d45 = *reinterpret_cast<const v2f*>(&s_rec[(rec[i] & 2040u) + 4]);
				const v2f x2 = {s[i].x, s[i].x}, y2 = {s[i].y, s[i].y}, z2 = {s[i].z, s[i].z}, r2 = {s[i].w, s[i].w};
				const v2f dd[3] = {{d03.x, d03.y}, {d03.z, d03.w}, d45};
				float m = 1e30f;
#pragma unroll
				for (int k = 0; k < 3; ++k) {
					v2f t = x2 * v2f{thresh + k, thresh - k};
					t = t + y2 * v2f{thresh * 2 + k, thresh * 3 - k};
					t = t + z2 * v2f{thresh * 5 + k, thresh * 7 - k};
					t = t + dd[k];
					t = t + r2;
					m = fminf(fminf(m, t.x), t.y);
				}
				vis = m == thresh * 11.f || (id[i] & 31) == 1;
			} else {
				acc += s[i].x + s[i].y + s[i].z + s[i].w;
				vis = (id[i] & 31) == 1 || acc == thresh;
			}
			keep[g + i] = vis ? id[i] : -1;
			staged += (unsigned)__builtin_popcountll(__ballot(vis));
		}
		__builtin_amdgcn_sched_barrier(0);
	}
	if (FEAT & 16) { // the epilogue by ONE wave per block: every wave compacts its ids in LDS and checks in; the last one in reserves for the block and writes all four lists, the others are gone
		__shared__ int s_ids[4][512];
		__shared__ unsigned s_cnt[4];
		unsigned& s_arrived = s_arrived0; // (zeroed in front of the kernel's first barrier)
		unsigned n = 0;
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const unsigned long long mask = __ballot(keep[i] >= 0);
			if (keep[i] >= 0) s_ids[wave][n + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u))] = keep[i];
			n += (unsigned)__builtin_popcountll(mask);
		}
		unsigned prev = 0;
		if (lane == 0) { s_cnt[wave] = n; prev = atomicAdd(&s_arrived, 1u); }
		prev = __builtin_amdgcn_readfirstlane(prev);
		if (prev != 3u) return;
		const unsigned c0 = s_cnt[0], c1 = s_cnt[1], c2 = s_cnt[2], c3 = s_cnt[3], total = c0 + c1 + c2 + c3;
		if (total == 0) return;
		unsigned base = 0;
		if (lane == 0) base = atomicAdd(&counters[shard * 32], total);
		base = __builtin_amdgcn_readfirstlane(base) + win;
		const unsigned cs[4] = {c0, c1, c2, c3};
#pragma unroll
		for (int w = 0; w < 4; ++w) {
			for (unsigned k = lane; k < cs[w]; k += 64u) out[(base + k) % (1u << 22)] = s_ids[w][k];
			base += cs[w];
		}
		return;
	}
	if (FEAT & (4 | 32 | 64 | 128 | 256 | 512 | 1024 | 2048 | 4096 | 8192 | 16384 | 32768)) { // 4: atomic + stores, 32: the atomic alone (stores predicated off by its result), 64: the stores alone (at a place that needs no atomic)
		if (staged == 0) return;
		unsigned base = (blockIdx.x * 4u + wave) * 16u;
		if (FEAT & 128) { // a NON-returning atomic: nothing waits for it
			if (lane == 0) __hip_atomic_fetch_add(&counters[shard * 32], staged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return;
		}
		if (FEAT & 4096) { // per wave: one NON-TEMPORAL 4-byte store
			if (lane == 0) __builtin_nontemporal_store((int)staged, out + base);
			return;
		}
		if (FEAT & 8192) { // per wave: one write-through (sc0 sc1) 4-byte store
			if (lane == 0) { int* p = out + base; const int v = (int)staged; asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory"); }
			return;
		}
		if (FEAT & 16384) { // per wave: one returning SYSTEM-scope atomic, then one write-through store at the place it names
			if (lane == 0) base = __hip_atomic_fetch_add(&counters[shard * 32], staged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			base = __builtin_amdgcn_readfirstlane(base);
			if (lane == 0) { int* p = out + (base % (1u << 22)); const int v = (int)staged; asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory"); }
			return;
		}
		if (FEAT & 32768) { // per wave: one store with sc1 only / nt sc1
			if (lane == 0) { int* p = out + base; const int v = (int)staged; asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
			return;
		}
		if (FEAT & 1024) { // ONE store of 4 bytes by ONE wave of the whole grid
			if (lane == 0 && blockIdx.x == 77 && wave == 1) out[base] = (int)staged;
			return;
		}
		if (FEAT & 2048) { // one 4-byte store by one wave of every 16th block
			if (lane == 0 && (blockIdx.x & 15) == 7 && wave == 1) out[base] = (int)staged;
			return;
		}
		if (FEAT & 256) { // ONE store of 4 bytes per wave, no atomic
			if (lane == 0) out[base] = (int)staged;
			return;
		}
		if (FEAT & 512) { // ONE store instruction of 64 bytes per wave (16 lanes), no atomic
			if (lane < 16) out[base + lane] = (int)staged;
			return;
		}
		if (!(FEAT & 64)) {
			if (lane == 0) base = atomicAdd(&counters[shard * 32], staged);
			base = __builtin_amdgcn_readfirstlane(base) + win;
			if ((FEAT & 32) && base != 0xfffffff0u) return;
		}
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const unsigned long long mask = __ballot(keep[i] >= 0);
			if (keep[i] >= 0) out[(base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u))) % (1u << 22)] = keep[i];
			base += (unsigned)__builtin_popcountll(mask);
		}
	} else if (acc + (float)staged == thresh) out[blockIdx.x * 256 + threadIdx.x] = keep[0] ^ keep[7]; // (never; `staged` needs every id)
}
// the epilogue form (one atomic + the stores per wave and tile) in a grid of RESIDENT blocks that walk the tiles: a wave's stores and atomic of tile k are in flight
// while it loads tile k + 1 - nothing is waited for at a wave's end except once
template <int GRP> __global__ __launch_bounds__(256) void read_persistent(const v4f* __restrict__ sp, const int* __restrict__ ids, unsigned* __restrict__ counters, int* __restrict__ out, unsigned tiles, float thresh) {
	const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	for (unsigned tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
		const size_t chunk0 = (size_t)tile * 32 + wave * 8;
		int keep[8]; unsigned staged = 0;
#pragma unroll
		for (int g = 0; g < 8; g += GRP) {
			v4f s[GRP]; int id[GRP];
#pragma unroll
			for (int i = 0; i < GRP; ++i) {
				const size_t e = ((chunk0 + g + i) << 6) + lane;
				s[i] = __builtin_nontemporal_load(sp + e); id[i] = __builtin_nontemporal_load(ids + e);
			}
#pragma unroll
			for (int i = 0; i < GRP; ++i) {
				const bool vis = (id[i] & 31) == 1 || s[i].x + s[i].y + s[i].z + s[i].w == thresh;
				keep[g + i] = vis ? id[i] : -1;
				staged += (unsigned)__builtin_popcountll(__ballot(vis));
			}
			__builtin_amdgcn_sched_barrier(0);
		}
		if (staged == 0) continue;
		unsigned base = 0;
		if (lane == 0) base = atomicAdd(&counters[(tile & 63u) * 32], staged);
		base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const unsigned long long mask = __ballot(keep[i] >= 0);
			if (keep[i] >= 0) out[(base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u))) % (1u << 22)] = keep[i];
			base += (unsigned)__builtin_popcountll(mask);
		}
	}
}
__global__ __launch_bounds__(256) void read_stride(const v4f* __restrict__ sp, const int* __restrict__ ids, int* __restrict__ out, size_t n, float thresh) {
	float acc = 0.f; int iacc = 0;
	for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
		const v4f s = __builtin_nontemporal_load(sp + e);
		acc += s.x + s.y + s.z + s.w; iacc ^= __builtin_nontemporal_load(ids + e);
	}
	if (acc + (float)iacc == thresh) out[blockIdx.x * 256 + threadIdx.x] = iacc;
}
__global__ __launch_bounds__(256) void scrub(const int4* __restrict__ p, size_t n, int* out) {
	int a = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a ^= p[i].x;
	if (a == 0x12345678) out[0] = a;
}
int main(int argc, char** argv) {
	const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : 10000384; // (a multiple of 2048)
	const size_t tiles = n / 2048;
	v4f* sp; int *ids, *out; int4* big;
	CK(hipMalloc(&sp, n * 16)); CK(hipMalloc(&ids, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&big, (size_t)1 << 30));
	CK(hipMemset(sp, 0, n * 16)); CK(hipMemset(ids, 0, n * 4)); CK(hipMemset(big, 1, (size_t)1 << 30));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto run = [&](const char* name, auto launch) {
		std::vector<float> t;
		for (int it = 0; it < 24; ++it) {
			hipLaunchKernelGGL(scrub, dim3(4096), dim3(256), 0, 0, big, ((size_t)1 << 30) / 16, out);
			launch(e0, e1);
			CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			if (it >= 4) t.push_back(ms * 1e3f);
		}
		std::sort(t.begin(), t.end());
		double avg = 0; for (float x : t) avg += x; avg /= t.size();
		printf("%-44s avg %6.2f us  median %6.2f  min %6.2f   %6.0f GB/s (avg)  = %.3f of 8 TB/s\n", name, avg, t[t.size() / 2], t[0], 20.0 * n / avg * 1e-3, 20.0 * n / avg * 1e-3 / 8000.0);
	};
	printf("%zu entities (%zu tiles of 2048): 20 B per entity = %.1f MB, cache-cold behind a 1 GiB read scrub, the dispatch's own begin/end timestamps\n", n, tiles, 20.0 * n * 1e-6);
	run("tiles, groups of 4, nt loads", [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((read_tiles<1, 4>), dim3(tiles), dim3(256), 0, 0, a, b, 0, sp, ids, out, 12345.f); });
	run("tiles, groups of 4, plain loads", [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((read_tiles<0, 4>), dim3(tiles), dim3(256), 0, 0, a, b, 0, sp, ids, out, 12345.f); });
	run("tiles, all 8 chunks in flight, nt loads", [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((read_tiles<1, 8>), dim3(tiles), dim3(256), 0, 0, a, b, 0, sp, ids, out, 12345.f); });
	int4 *box, *hdr; uint2* tab; unsigned* counters;
	CK(hipMalloc(&box, tiles * 32)); CK(hipMalloc(&tab, tiles * 8)); CK(hipMalloc(&hdr, n / 64 * 16)); CK(hipMalloc(&counters, 64 * 32 * 4));
	CK(hipMemset(box, 0, tiles * 32)); CK(hipMemset(tab, 0, tiles * 8)); CK(hipMemset(hdr, 0, n / 64 * 16)); CK(hipMemset(counters, 0, 64 * 32 * 4));
#define FEAT_RUN(F, label) run(label, [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((read_feat<F>), dim3(tiles), dim3(256), 0, 0, a, b, 0, sp, ids, box, tab, hdr, counters, out, 12345.f); })
	FEAT_RUN(0, "skeleton 0: LDS fill + barrier only");
	FEAT_RUN(1, "+ prologue (box -> verdict -> barrier)");
	FEAT_RUN(2, "+ header + class read before loads");
	FEAT_RUN(4, "+ epilogue (atomic + 3 % stored)");
	FEAT_RUN(8, "+ record reads + plane arithmetic");
	FEAT_RUN(3, "prologue + headers");
	FEAT_RUN(5, "prologue + epilogue");
	FEAT_RUN(7, "prologue + headers + epilogue");
	FEAT_RUN(12, "epilogue + arithmetic");
	FEAT_RUN(15, "all four");
	FEAT_RUN(32, "+ the atomic alone");
	FEAT_RUN(64, "+ the stores alone");
	FEAT_RUN(1024, "+ ONE 4-byte store in the whole grid");
	FEAT_RUN(2048, "+ one 4-byte store per 64 waves");
	FEAT_RUN(4096, "+ one NON-TEMPORAL store per wave");
	FEAT_RUN(8192, "+ one sc0 sc1 (write-through) store per wave");
	FEAT_RUN(32768, "+ one sc1 store per wave");
	FEAT_RUN(16384, "+ system-scope atomic + sc0 sc1 store per wave");
	FEAT_RUN(128, "+ a non-returning atomic alone");
	FEAT_RUN(256, "+ one 4-byte store per wave alone");
	FEAT_RUN(512, "+ one 64-byte store per wave alone");
	FEAT_RUN(16, "+ epilogue by the block's last wave (LDS)");
	for (unsigned g : {1024u, 1536u, 2048u, 2560u}) {
		char label[96]; snprintf(label, sizeof label, "persistent, %u blocks, epilogue per tile", g);
		run(label, [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((read_persistent<4>), dim3(g), dim3(256), 0, 0, a, b, 0, sp, ids, counters, out, (unsigned)tiles, 12345.f); });
	}
	run("persistent, 2048 blocks, 8 chunks in flight", [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((read_persistent<8>), dim3(2048), dim3(256), 0, 0, a, b, 0, sp, ids, counters, out, (unsigned)tiles, 12345.f); });
	FEAT_RUN(16 + 11, "all four, epilogue by the last wave");
	// is the cost of "the kernel stores anything" a property of the launch's end (its release) or of the measurement? WARM, back to back, by the host's clock and by the events
	auto back_to_back = [&](const char* name, auto launch) {
		for (int it = 0; it < 5; ++it) launch(nullptr, nullptr);
		CK(hipDeviceSynchronize());
		hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
		const int reps = 200;
		CK(hipEventRecord(a, 0));
		for (int it = 0; it < reps; ++it) launch(nullptr, nullptr);
		CK(hipEventRecord(b, 0));
		CK(hipEventSynchronize(b));
		float ms; CK(hipEventElapsedTime(&ms, a, b));
		printf("%-44s %6.2f us per launch, %d launches back to back, no scrub (stream time between two events)\n", name, ms * 1e3 / reps, reps);
	};
	back_to_back("warm: no store at all", [&](hipEvent_t, hipEvent_t) { hipLaunchKernelGGL((read_feat<0>), dim3(tiles), dim3(256), 0, 0, sp, ids, box, tab, hdr, counters, out, 12345.f); });
	back_to_back("warm: ONE 4-byte store in the grid", [&](hipEvent_t, hipEvent_t) { hipLaunchKernelGGL((read_feat<1024>), dim3(tiles), dim3(256), 0, 0, sp, ids, box, tab, hdr, counters, out, 12345.f); });
	back_to_back("warm: atomic + 3 % stored per wave", [&](hipEvent_t, hipEvent_t) { hipLaunchKernelGGL((read_feat<4>), dim3(tiles), dim3(256), 0, 0, sp, ids, box, tab, hdr, counters, out, 12345.f); });
	run("grid-stride, 2048 blocks, nt loads", [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(read_stride, dim3(2048), dim3(256), 0, 0, a, b, 0, sp, ids, out, n, 12345.f); });
	run("grid-stride, 4096 blocks, nt loads", [&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(read_stride, dim3(4096), dim3(256), 0, 0, a, b, 0, sp, ids, out, n, 12345.f); });
	return 0;
}
