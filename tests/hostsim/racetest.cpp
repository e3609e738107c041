// racetest.cpp — TEST INFRASTRUCTURE: checks the race detector of tests/hostsim (--sanitize thread + HOSTSIM_RACE=1) on three kernels:
//   racy_lds     wave 1 reads LDS words wave 0 wrote, no barrier in between         -> must be reported
//   racy_global  two blocks add to one global word with plain loads and stores       -> must be reported
//   clean        the same exchange behind __syncthreads(), the global word by atomicAdd, lanes of ONE wave exchanging through LDS behind
//                the wave barrier                                                      -> must be quiet
// argv[1] selects the kernel; the caller greps ThreadSanitizer's report.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

namespace {

__global__ void racy_lds(uint32_t* out) {
	__shared__ uint32_t s[64];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	if (wave == 0) s[lane] = lane * 3u;
	if (wave == 1) out[lane] = s[63u - lane];
}

__global__ void racy_global(uint32_t* out) {
	if (threadIdx.x == 0) out[0] = out[0] + blockIdx.x + 1u;
}

__global__ void clean(uint32_t* out) {
	__shared__ uint32_t s[128];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	s[threadIdx.x] = threadIdx.x * 3u;
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	const uint32_t own_wave = s[wave * 64u + (63u - lane)]; // lanes of one wave: ordered by the wave barrier
	__syncthreads();
	const uint32_t other_wave = s[(1u - wave) * 64u + lane]; // the other wave's words: ordered by the block barrier
	out[1 + blockIdx.x * 128u + threadIdx.x] = own_wave + other_wave;
	if (lane == 0) atomicAdd(out, 1u);
}

} // namespace

int main(int argc, char** argv) {
	uint32_t* d;
	(void)hipMalloc(&d, (1 + 4 * 128) * 4);
	(void)hipMemset(d, 0, (1 + 4 * 128) * 4);
	const char* which = argc > 1 ? argv[1] : "clean";
	if (!strcmp(which, "racy_lds")) hipLaunchKernelGGL(racy_lds, dim3(1), dim3(128), 0, 0, d);
	else if (!strcmp(which, "racy_global")) hipLaunchKernelGGL(racy_global, dim3(2), dim3(64), 0, 0, d);
	else hipLaunchKernelGGL(clean, dim3(4), dim3(128), 0, 0, d);
	uint32_t h = 0;
	(void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
	printf("%s done (%u)\n", which, h);
	(void)hipFree(d);
	return 0;
}
