// Development probe: pure-store bandwidth of one MI355X for the store shapes k_skin_vertices / k_pose_palette can use.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/write_probe.hip -o tools/_build/write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct F3 { float x, y, z; };
template <int NT> __global__ __launch_bounds__(256) void fill4(float4* p, size_t n, int per) {
	size_t i = (size_t)blockIdx.x * 256 * per + threadIdx.x;
	const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
	for (int k = 0; k < per; ++k, i += 256) if (i < n) { if (NT) { typedef float v4 __attribute__((ext_vector_type(4))); __builtin_nontemporal_store(v4{v.x, v.y, v.z, v.w}, (v4*)(p + i)); } else p[i] = v; }
}
template <int NT> __global__ __launch_bounds__(256) void fill3(F3* p, size_t n, int per) {
	size_t i = (size_t)blockIdx.x * 256 * per + threadIdx.x;
	for (int k = 0; k < per; ++k, i += 256) if (i < n) {
		if (NT) { float* f = (float*)(p + i); __builtin_nontemporal_store(1.f, f); __builtin_nontemporal_store(2.f, f + 1); __builtin_nontemporal_store((float)k, f + 2); }
		else p[i] = F3{1.f, 2.f, (float)k};
	}
}
// a lane writes FOUR consecutive 12-byte vertices = 48 contiguous bytes as three 16-byte stores (lane stride 48 B): the store form VERDICT r5 item 5a asks about
// ("each lane skinning 4 consecutive vertices so its output is three aligned global_store_dwordx4, no LDS transposition")
template <int NT> __global__ __launch_bounds__(256) void fill48(float4* p, size_t n4, int per) {
	size_t lane0 = ((size_t)blockIdx.x * 256 * per + threadIdx.x) * 3;
	const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
	typedef float v4 __attribute__((ext_vector_type(4)));
	for (int k = 0; k < per; ++k, lane0 += 256 * 3) {
#pragma unroll
		for (int j = 0; j < 3; ++j)
			if (lane0 + j < n4) { if (NT) __builtin_nontemporal_store(v4{v.x, v.y, v.z, v.w}, (v4*)(p + lane0 + j)); else p[lane0 + j] = v; }
	}
}
// k_skin_multi<2>'s own pattern: lane l writes the 12 bytes of vertex l / 2 of instance l % 2 (two runs of 32 x 12 = 384 contiguous bytes per store instruction)
template <int NT> __global__ __launch_bounds__(256) void fill3x2(F3* p, size_t n_half, int per) {
	const unsigned inst = threadIdx.x & 1u;
	size_t v = (size_t)blockIdx.x * 128 * per + (threadIdx.x >> 1);
	for (int k = 0; k < per; ++k, v += 128) if (v < n_half) {
		F3* dst = p + inst * n_half + v;
		if (NT) { float* f = (float*)dst; __builtin_nontemporal_store(1.f, f); __builtin_nontemporal_store(2.f, f + 1); __builtin_nontemporal_store((float)k, f + 2); }
		else *dst = F3{1.f, 2.f, (float)k};
	}
}
__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n, int per) {
	size_t i = (size_t)blockIdx.x * 256 * per + threadIdx.x;
	for (int k = 0; k < per; ++k, i += 256) if (i < n) b[i] = a[i];
}
int main() {
	const size_t bytes = (size_t)6 << 30;
	char *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto time = [&](const char* name, auto launch, double moved) {
		float best = 1e9f;
		for (int it = 0; it < 4; ++it) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < best) best = ms; }
		printf("%-28s %8.3f ms  %7.1f GB/s\n", name, best, moved / best * 1e-6);
	};
	for (int per : {1, 4, 16}) {
		const size_t n4 = bytes / 16, n3 = bytes / 12;
		const unsigned g4 = (unsigned)((n4 + 256 * per - 1) / (256 * per)), g3 = (unsigned)((n3 + 256 * per - 1) / (256 * per));
		printf("-- %d stores per thread\n", per);
		time("fill float4", [&] { hipLaunchKernelGGL(fill4<0>, dim3(g4), dim3(256), 0, 0, (float4*)a, n4, per); }, (double)bytes);
		time("fill float4 nt", [&] { hipLaunchKernelGGL(fill4<1>, dim3(g4), dim3(256), 0, 0, (float4*)a, n4, per); }, (double)bytes);
		time("fill 12B", [&] { hipLaunchKernelGGL(fill3<0>, dim3(g3), dim3(256), 0, 0, (F3*)a, n3, per); }, (double)bytes);
		time("fill 12B nt", [&] { hipLaunchKernelGGL(fill3<1>, dim3(g3), dim3(256), 0, 0, (F3*)a, n3, per); }, (double)bytes);
		{ const unsigned g48 = (unsigned)((n4 / 3 + 256 * per - 1) / (256 * per)); const unsigned g32 = (unsigned)((n3 / 2 + 128 * per - 1) / (128 * per));
		time("fill 48 B per lane (3 x 16 B)", [&] { hipLaunchKernelGGL(fill48<0>, dim3(g48), dim3(256), 0, 0, (float4*)a, n4, per); }, (double)bytes);
		time("fill 48 B per lane nt", [&] { hipLaunchKernelGGL(fill48<1>, dim3(g48), dim3(256), 0, 0, (float4*)a, n4, per); }, (double)bytes);
		time("fill 12B, 2 instances interleaved", [&] { hipLaunchKernelGGL(fill3x2<0>, dim3(g32), dim3(256), 0, 0, (F3*)a, n3 / 2, per); }, (double)bytes);
		time("fill 12B, 2 instances interleaved nt", [&] { hipLaunchKernelGGL(fill3x2<1>, dim3(g32), dim3(256), 0, 0, (F3*)a, n3 / 2, per); }, (double)bytes); }
		time("copy float4 (r+w bytes)", [&] { hipLaunchKernelGGL(copy4, dim3(g4), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4, per); }, 2.0 * bytes);
	}
	time("hipMemsetAsync", [&] { CK(hipMemsetAsync(a, 1, bytes, 0)); }, (double)bytes);
	return 0;
}
