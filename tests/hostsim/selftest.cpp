// selftest.cpp — TEST INFRASTRUCTURE: pins the simulated device's semantics (tests/hostsim) on kernels whose results are known by
// construction: wave64 ballots and ranks, readlane / readfirstlane under divergence, shuffles, DPP quad permutes, LDS exchange behind
// the block barrier and the wave barrier, early-exit loops (lanes that left a loop wait for the others at the reconvergence point),
// dynamic LDS, the kernarg segment, atomics. Exit code 0 = all good.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

namespace {

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

// out[tid]: [0] ballot of (lane % 3 == 0) low word, [1] rank among them, [2] ballot inside a divergent branch (odd lanes), [3] readfirstlane in that branch,
// [4] full ballot after the branch
__global__ void k_ballots(uint32_t* out) {
	const uint32_t lane = lane_id(), t = threadIdx.x;
	const uint64_t m = __ballot(lane % 3u == 0u);
	out[5 * t + 0] = (uint32_t)m ^ (uint32_t)(m >> 32);
	out[5 * t + 1] = mbcnt64(m);
	uint32_t inner = 0, first = 0;
	if (lane & 1u) {
		inner = (uint32_t)__popcll(__ballot(lane < 40u)); // only odd lanes are active: 20 of them are below 40
		first = __builtin_amdgcn_readfirstlane(lane * 7u); // lowest active lane is lane 1
	}
	out[5 * t + 2] = inner;
	out[5 * t + 3] = first;
	out[5 * t + 4] = (uint32_t)__popcll(__ballot(true));
}

// lanes leave the loop after (lane % 5) + 1 iterations; each iteration holds a ballot of the lanes still looping
__global__ void k_loop(uint32_t* out) {
	const uint32_t lane = lane_id();
	uint32_t acc = 0;
	for (uint32_t it = 0; it <= lane % 5u; ++it) acc += (uint32_t)__popcll(__ballot(true)); // lanes with lane % 5 >= it
	const uint32_t after = (uint32_t)__popcll(__ballot(true));                                 // all 64 again
	out[2 * threadIdx.x] = acc;
	out[2 * threadIdx.x + 1] = after;
}

__global__ void k_shuffles(int* out) {
	const int lane = (int)lane_id();
	const int v = 100 + lane;
	int* o = out + 6 * threadIdx.x;
	o[0] = __shfl(v, 5);
	o[1] = __shfl_up(v, 3);
	o[2] = __shfl_down(v, 4);
	o[3] = __shfl_xor(v, 32);
	o[4] = __builtin_amdgcn_readlane(v, 63);
	o[5] = __builtin_amdgcn_mov_dpp(v, 1 | 2 << 2 | 3 << 4 | 3 << 6, 0xf, 0xf, true); // quad_perm [1, 2, 3, 3]
}

// block of 256: reverse through static LDS behind the block barrier, rotate inside the wave through LDS behind the wave barrier,
// dynamic LDS carries a per-wave sum, the kernarg segment is read directly
struct Args { uint32_t magic[4]; };
__global__ void k_lds(const Args a, const uint32_t* in, uint32_t* out, uint32_t* counter) {
	__shared__ uint32_t s_a[256];
	__shared__ uint32_t s_b[256];
	LMX_DYNAMIC_LDS(uint32_t, s_dyn);
	const uint32_t t = threadIdx.x, lane = lane_id(), wave = t >> 6;
	s_a[t] = in[blockIdx.x * 256u + t];
	if (t < 4) s_dyn[t] = 0;
	__syncthreads();
	const uint32_t rev = s_a[255u - t];
	s_b[t] = rev;
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	const uint32_t rot = s_b[wave * 64u + ((lane + 1u) & 63u)];
	atomicAdd(&s_dyn[wave], rot);
	__syncthreads();
	const uint32_t* ka = (const uint32_t*)__builtin_amdgcn_kernarg_segment_ptr();
	out[blockIdx.x * 256u + t] = rot + ka[lane & 3u] + s_dyn[wave] * 0u + (a.magic[0] - ka[0]);
	if (lane == 0) atomicAdd(counter, s_dyn[wave]);
}

int fails = 0;
void check(bool ok, const char* what) {
	if (!ok) {
		printf("FAIL: %s\n", what);
		++fails;
	}
}

} // namespace

int main() {
	{
		uint32_t* d;
		(void)hipMalloc(&d, 128 * 5 * 4);
		hipLaunchKernelGGL(k_ballots, dim3(1), dim3(128), 0, 0, d);
		std::vector<uint32_t> h(128 * 5);
		(void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
		uint64_t m = 0;
		for (int l = 0; l < 64; ++l) if (l % 3 == 0) m |= 1ull << l;
		bool ok = true;
		for (int t = 0; t < 128; ++t) {
			const int l = t & 63;
			ok &= h[5 * t] == ((uint32_t)m ^ (uint32_t)(m >> 32));
			ok &= h[5 * t + 1] == (uint32_t)__builtin_popcountll(m & ((1ull << l) - 1));
			ok &= h[5 * t + 2] == ((l & 1) ? 20u : 0u);
			ok &= h[5 * t + 3] == ((l & 1) ? 7u : 0u);
			ok &= h[5 * t + 4] == 64u;
		}
		check(ok, "ballot / mbcnt / divergent ballot / readfirstlane");
		(void)hipFree(d);
	}
	{
		uint32_t* d;
		(void)hipMalloc(&d, 64 * 2 * 4);
		hipLaunchKernelGGL(k_loop, dim3(1), dim3(64), 0, 0, d);
		std::vector<uint32_t> h(128);
		(void)hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
		bool ok = true;
		uint32_t still[5];
		for (int it = 0; it < 5; ++it) {
			still[it] = 0;
			for (int l = 0; l < 64; ++l) still[it] += (l % 5 >= it);
		}
		for (int l = 0; l < 64; ++l) {
			uint32_t want = 0;
			for (int it = 0; it <= l % 5; ++it) want += still[it];
			ok &= h[2 * l] == want && h[2 * l + 1] == 64u;
		}
		check(ok, "early-exit loop: per-iteration EXEC masks and reconvergence behind the loop");
		(void)hipFree(d);
	}
	{
		int* d;
		(void)hipMalloc(&d, 64 * 6 * 4);
		hipLaunchKernelGGL(k_shuffles, dim3(1), dim3(64), 0, 0, d);
		std::vector<int> h(64 * 6);
		(void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
		bool ok = true;
		const int perm[4] = {1, 2, 3, 3};
		for (int l = 0; l < 64; ++l) {
			const int* o = &h[6 * l];
			ok &= o[0] == 105;
			ok &= o[1] == (l >= 3 ? 100 + l - 3 : 100 + l);
			ok &= o[2] == (l + 4 < 64 ? 100 + l + 4 : 100 + l);
			ok &= o[3] == 100 + (l ^ 32);
			ok &= o[4] == 163;
			ok &= o[5] == 100 + (l & ~3) + perm[l & 3];
		}
		check(ok, "shuffles / readlane / DPP quad_perm");
		(void)hipFree(d);
	}
	{
		const uint32_t blocks = 7;
		std::vector<uint32_t> in(blocks * 256), out(blocks * 256);
		for (size_t i = 0; i < in.size(); ++i) in[i] = (uint32_t)(i * 2654435761u);
		uint32_t *din, *dout, *dcnt;
		(void)hipMalloc(&din, in.size() * 4);
		(void)hipMalloc(&dout, in.size() * 4);
		(void)hipMalloc(&dcnt, 4);
		(void)hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
		(void)hipMemset(dcnt, 0, 4);
		Args a = {{11, 22, 33, 44}};
		hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 16, 0, a, din, dout, dcnt);
		check(hipGetLastError() == hipSuccess, "launch");
		(void)hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
		uint32_t cnt = 0, want_cnt = 0;
		(void)hipMemcpy(&cnt, dcnt, 4, hipMemcpyDeviceToHost);
		bool ok = true;
		for (uint32_t b = 0; b < blocks; ++b)
			for (uint32_t t = 0; t < 256; ++t) {
				const uint32_t wave = t >> 6, lane = t & 63u;
				const uint32_t src = wave * 64u + ((lane + 1u) & 63u); // index into the reversed array
				const uint32_t rot = in[b * 256u + 255u - src];
				ok &= out[b * 256u + t] == rot + a.magic[lane & 3u];
				want_cnt += rot;
			}
		check(ok, "static LDS / wave barrier / kernarg segment");
		check(cnt == want_cnt, "dynamic LDS + atomics");
		(void)hipFree(din);
		(void)hipFree(dout);
		(void)hipFree(dcnt);
	}
	printf(fails ? "hostsim selftest: %d failure(s)\n" : "hostsim selftest: ok\n", fails);
	return fails ? 1 : 0;
}
