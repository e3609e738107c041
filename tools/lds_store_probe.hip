// Development probe: do LDS reads (ds_read_b128 streams) and global stores (global_store_dwordx3) of one CU overlap?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lds_store_probe.hip -o tools/_build/lds_store_probe
// k_skin_shared runs at (LDS-read time + store time), not at their maximum (tools/skin_probe.hip: 1.36 ms with the stores compiled
// out, 2.2 ms with the LDS reads compiled out, 3.2 ms with both, per 1e9 vertices). This probe reproduces the two streams without
// the skinning arithmetic, per block of 1024 lanes = one CU, in four arrangements:
//   A  every wave: 12 ds_read_b128 per "vertex", results folded into a register (no stores)
//   B  every wave: one 12-byte store per "vertex" (no LDS reads)
//   C  every wave: both, alternating (the skinning kernel's shape)
//   D  specialised: even waves do two vertices' LDS reads, odd waves do two vertices' stores (same totals per block as C)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct F3 { float x, y, z; };
constexpr int THREADS = 1024, SLOTS = 3072;

__device__ __forceinline__ float lds_vertex(const float4* rows, uint32_t& h, uint32_t col) {
	// 4 "bones" x 3 rows, 16 copies: slot = (bone * 3 + row) * 16 + col  (conflict-free, as the skinning kernel)
	h = h * 1664525u + 1013904223u;
	const uint32_t b0 = (h >> 8) & 63u, b1 = (h >> 14) & 63u, b2 = (h >> 20) & 63u, b3 = (h >> 26) & 63u;
	float acc = 0.f;
#pragma unroll
	for (int r = 0; r < 3; ++r) {
		const float4 A = rows[(b0 * 3 + r) * 16 + col], B = rows[(b1 * 3 + r) * 16 + col], C = rows[(b2 * 3 + r) * 16 + col], D = rows[(b3 * 3 + r) * 16 + col];
		acc += ((A.x + A.y) + (A.z + A.w)) + ((B.x + B.y) + (B.z + B.w)) + ((C.x + C.y) + (C.z + C.w)) + ((D.x + D.y) + (D.z + D.w)); // all 16 floats: ds_read_b128
	}
	return acc;
}

template <int MODE>
__global__ __launch_bounds__(THREADS) void k_probe(F3* __restrict__ out, uint32_t iters, uint32_t verts_per_block_iter) {
	__shared__ float4 s_rows[2][SLOTS];
	const uint32_t tid = threadIdx.x, col = tid & 15u, wave = tid >> 6;
	for (uint32_t i = tid; i < 2 * SLOTS; i += THREADS) (&s_rows[0][0])[i] = make_float4((float)i, 1.f, 2.f, 3.f);
	__syncthreads();
	uint32_t h = tid * 2654435761u + blockIdx.x;
	F3* o = out + (size_t)blockIdx.x * iters * verts_per_block_iter;
	float keep = 0.f;
	for (uint32_t it = 0; it < iters; ++it) {
		const float4* rows = s_rows[it & 1];
		F3* oi = o + (size_t)it * verts_per_block_iter;
#pragma unroll
		for (int k = 0; k < 5; ++k) {
			if (MODE == 0) { // A
				keep += lds_vertex(rows, h, col);
			} else if (MODE == 1) { // B
				oi[k * THREADS + tid] = F3{(float)it, (float)k, keep};
			} else if (MODE == 2) { // C
				const float a = lds_vertex(rows, h, col);
				oi[k * THREADS + tid] = F3{a, (float)k, keep};
			} else { // D: even waves read for two vertices, odd waves store two vertices
				if ((wave & 1u) == 0) {
					keep += lds_vertex(rows, h, col);
					keep += lds_vertex(rows, h, col);
				} else {
					oi[k * THREADS + tid] = F3{(float)it, (float)k, keep};
					oi[k * THREADS + (tid - 64)] = F3{(float)it, (float)k, keep + 1.f};
				}
			}
			__builtin_amdgcn_sched_barrier(0);
		}
		if (MODE != 3) __syncthreads(); // the skinning kernel's per-instance barrier (D's halves never meet)
	}
	if (keep == 123.456f) out[0].x = keep;
}

int main(int argc, char** argv) {
	const uint32_t blocks = argc > 1 ? atoi(argv[1]) : 626, iters = 64, vpi = 5 * THREADS;
	F3* out;
	const size_t n = (size_t)blocks * iters * vpi;
	CK(hipMalloc(&out, n * sizeof(F3)));
	CK(hipMemset(out, 0, n * sizeof(F3)));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto run = [&](const char* name, auto kern) {
		float best = 1e9f;
		for (int r = 0; r < 5; ++r) {
			CK(hipEventRecord(e0));
			hipLaunchKernelGGL(kern, dim3(blocks), dim3(THREADS), 0, 0, out, iters, vpi);
			CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			if (r && ms < best) best = ms;
		}
		printf("%-44s %.4f ms = %.3f ms per 1e9 vertices\n", name, best, best * 1e9 / (double)n);
	};
	run("A  LDS reads only", k_probe<0>);
	run("B  stores only", k_probe<1>);
	run("C  both, every wave alternating", k_probe<2>);
	run("D  specialised waves (even: reads, odd: stores)", k_probe<3>);
	return 0;
}
