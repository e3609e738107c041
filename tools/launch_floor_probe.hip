// launch_floor_probe.hip — what a kernel of the cull kernel's SHAPE costs when it does nothing: back-to-back launches of
//   (a) an empty kernel, (b) a kernel whose first lane reads one 16-byte header per block and exits,
// for the grids the cull uses (2442 x 256 threads for 10 M entities) and for smaller persistent grids. Prints JSON.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_floor_probe.hip -o tools/_build/launch_floor_probe && tools/_build/launch_floor_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_empty() {}
__global__ void k_hdr(const uint4* __restrict__ hdr, unsigned* __restrict__ out) {
	if (threadIdx.x == 0) {
		const uint4 h = hdr[blockIdx.x];
		if (h.x == 0xdeadbeefu) out[0] = h.y; // never
	}
}
// one wave tests 64 "tiles" at once, then the block exits: the strided persistent shape
__global__ void k_hdr_strided(const uint4* __restrict__ hdr, unsigned n_tiles, unsigned* __restrict__ out) {
	if (threadIdx.x < 64) {
		const unsigned t = blockIdx.x + gridDim.x * threadIdx.x;
		if (t < n_tiles) {
			const uint4 h = hdr[t];
			if (h.x == 0xdeadbeefu) out[0] = h.y;
		}
	}
}

// the tile test's shape: wave 0 reads the block's 32-byte box, a few dozen dependent operations, verdict through LDS + barrier, everybody exits
__global__ void k_box_verdict(const uint4* __restrict__ hdr, unsigned* __restrict__ out) {
	__shared__ unsigned s_v;
	if (threadIdx.x < 64) {
		const uint4 h = hdr[blockIdx.x * 2 + (threadIdx.x & 1)];
		float v = (float)h.x + (float)threadIdx.x;
#pragma unroll
		for (int k = 0; k < 40; ++k) v = v * 1.0001f + (float)h.y;
		if (threadIdx.x == 0) s_v = v == 12345.f ? 1u : 0u;
	}
	__syncthreads();
	if (s_v) out[threadIdx.x] = s_v; // never
}
// the same for TWO tiles per block of twice the threads (waves of the upper half would take the second tile)
__global__ void k_box_verdict2(const uint4* __restrict__ hdr, unsigned* __restrict__ out) {
	__shared__ unsigned s_v;
	if (threadIdx.x < 64) {
		const uint4 h = hdr[blockIdx.x * 4 + (threadIdx.x & 3)];
		float v = (float)h.x + (float)threadIdx.x;
#pragma unroll
		for (int k = 0; k < 40; ++k) v = v * 1.0001f + (float)h.y;
		if (threadIdx.x == 0) s_v = v == 12345.f ? 1u : 0u;
	}
	__syncthreads();
	if (s_v) out[threadIdx.x] = s_v;
}

template <typename F> static void time_it(const char* name, int grid, int block, F launch, bool last = false) {
	hipStream_t s;
	hipStreamCreate(&s);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int i = 0; i < 50; ++i) launch(s);
	hipStreamSynchronize(s);
	const int N = 400;
	hipEventRecord(e0, s);
	const auto t0 = std::chrono::steady_clock::now();
	for (int i = 0; i < N; ++i) launch(s);
	hipEventRecord(e1, s);
	hipStreamSynchronize(s);
	const auto t1 = std::chrono::steady_clock::now();
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	printf("  {\"kernel\": \"%s\", \"grid\": %d, \"block\": %d, \"event_us_per_launch\": %.3f, \"wall_us_per_launch\": %.3f}%s\n", name, grid, block, ms * 1e3 / N,
		std::chrono::duration<double, std::micro>(t1 - t0).count() / N, last ? "" : ",");
	hipStreamDestroy(s);
}

int main() {
	const unsigned n_tiles = 2442;
	uint4* hdr;
	unsigned* out;
	hipMalloc(&hdr, sizeof(uint4) * 65536);
	hipMemset(hdr, 0, sizeof(uint4) * 65536);
	hipMalloc(&out, 64);
	printf("[\n");
	for (int grid : {1, 256, 512, 1024, 2442, 4884, 24420}) {
		time_it("empty", grid, 256, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, s); });
		time_it("hdr", grid, 256, [&](hipStream_t s) { hipLaunchKernelGGL(k_hdr, dim3(grid), dim3(256), 0, s, hdr, out); });
	}
	time_it("empty", 2442, 64, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(2442), dim3(64), 0, s); });
	time_it("empty", 2442, 1024, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(2442), dim3(1024), 0, s); });
	// round 6: what the headline camera's launch is made of (95 % of its 4883 tiles end at the box test) - and whether fewer, larger blocks would make it cheaper
	time_it("box_verdict", 4884, 256, [&](hipStream_t s) { hipLaunchKernelGGL(k_box_verdict, dim3(4884), dim3(256), 0, s, hdr, out); });
	time_it("box_verdict", 4884, 64, [&](hipStream_t s) { hipLaunchKernelGGL(k_box_verdict, dim3(4884), dim3(64), 0, s, hdr, out); });
	time_it("box_verdict 2 tiles per block", 2442, 512, [&](hipStream_t s) { hipLaunchKernelGGL(k_box_verdict2, dim3(2442), dim3(512), 0, s, hdr, out); });
	time_it("box_verdict 2 tiles per block", 2442, 256, [&](hipStream_t s) { hipLaunchKernelGGL(k_box_verdict2, dim3(2442), dim3(256), 0, s, hdr, out); });
	time_it("box_verdict 2 tiles per block", 2442, 64, [&](hipStream_t s) { hipLaunchKernelGGL(k_box_verdict2, dim3(2442), dim3(64), 0, s, hdr, out); });
	time_it("hdr_strided", 39, 256, [&](hipStream_t s) { hipLaunchKernelGGL(k_hdr_strided, dim3(39), dim3(256), 0, s, hdr, n_tiles, out); });
	time_it("hdr_strided", 256, 256, [&](hipStream_t s) { hipLaunchKernelGGL(k_hdr_strided, dim3(256), dim3(256), 0, s, hdr, n_tiles, out); }, true);
	printf("]\n");
	return 0;
}
