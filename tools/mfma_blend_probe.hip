// mfma_blend_probe.hip — evidence for / against MFMA in the linear-blend-skinning inner loop (BASELINE north star: "MFMA only for the
// batched 4x4 bone-matrix x vertex contractions - choices evidenced by rocprof ... MFMA utilisation vs peak").
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_blend_probe.hip -o tools/_build/mfma_blend_probe && tools/_build/mfma_blend_probe
//   rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -- tools/_build/mfma_blend_probe
//
// evaluateSkin (src/renderer/model.cpp:103-109): out = (sum_j w_j * P[idx_j]) . (p, 1), P = 3 x 4 palette rows gathered PER VERTEX
// (4 bone indices per vertex, unrelated between neighbouring vertices). Both kernels skin the same vertices against the same
// palette staged in LDS, repeated REPS times on register-resident vertex records (no HBM traffic in the timed loop):
//   k_blend_valu   one vertex per lane, 12 ds_read_b128 + 30 v_pk_fma_f32 / v_fma_f32 per vertex (the product's FUSED arithmetic)
//   k_blend_mfma   v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products per instruction, block = vertex (4 lanes), one
//                  instruction per (bone j, column c): D[row i][0] += P[idx_j][i][c] * (w_j * p4[c]). A matrix core needs an operand
//                  SHARED across the tile; here every vertex brings its own 4 matrices, so only column 0 of each 4x4 block (and 3 of
//                  its 4 rows) carries a result: 12 useful MACs of the 256 an instruction performs per... block.
// Output: ns per vertex of both, max relative difference, and the instruction counts come from the --pmc run.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int N_BONES = 64;
constexpr int THREADS = 256;
typedef float v4f __attribute__((ext_vector_type(4)));

struct Vtx { float px, py, pz; float w[4]; uint32_t idx[4]; };

__global__ __launch_bounds__(THREADS) void k_blend_valu(const float4* __restrict__ palette, const Vtx* __restrict__ vtx, float* __restrict__ out, int reps) {
	__shared__ float4 s_rows[N_BONES * 3];
	for (int i = threadIdx.x; i < N_BONES * 3; i += THREADS) s_rows[i] = palette[i];
	__syncthreads();
	const uint32_t v = blockIdx.x * THREADS + threadIdx.x;
	Vtx c = vtx[v];
	float ax = 0, ay = 0, az = 0;
	for (int rep = 0; rep < reps; ++rep) {
		asm volatile("" : "+v"(c.px), "+v"(c.py), "+v"(c.pz), "+v"(c.w[0]), "+v"(c.w[1]), "+v"(c.w[2]), "+v"(c.w[3]));
		float o[3];
#pragma unroll
		for (int r = 0; r < 3; ++r) {
			const float4 A = s_rows[c.idx[0] * 3 + r], B = s_rows[c.idx[1] * 3 + r], C = s_rows[c.idx[2] * 3 + r], D = s_rows[c.idx[3] * 3 + r];
			const float m0 = fmaf(D.x, c.w[3], fmaf(C.x, c.w[2], fmaf(B.x, c.w[1], A.x * c.w[0])));
			const float m1 = fmaf(D.y, c.w[3], fmaf(C.y, c.w[2], fmaf(B.y, c.w[1], A.y * c.w[0])));
			const float m2 = fmaf(D.z, c.w[3], fmaf(C.z, c.w[2], fmaf(B.z, c.w[1], A.z * c.w[0])));
			const float m3 = fmaf(D.w, c.w[3], fmaf(C.w, c.w[2], fmaf(B.w, c.w[1], A.w * c.w[0])));
			o[r] = fmaf(m2, c.pz, fmaf(m1, c.py, m0 * c.px)) + m3;
		}
		ax += o[0]; ay += o[1]; az += o[2];
		c.px += 1e-7f; // the next repetition is not loop-invariant
	}
	out[3 * v] = ax; out[3 * v + 1] = ay; out[3 * v + 2] = az;
}

// 4 lanes per vertex: lane 4 b + i is row i of block b. A operand: P[idx_j][i][c] (i = 3: 0). B operand: w_j * p4[c] in column 0 only.
__global__ __launch_bounds__(THREADS) void k_blend_mfma(const float4* __restrict__ palette, const Vtx* __restrict__ vtx, float* __restrict__ out, int reps) {
	__shared__ float s_pal[N_BONES * 12];
	for (int i = threadIdx.x; i < N_BONES * 12; i += THREADS) s_pal[i] = reinterpret_cast<const float*>(palette)[i];
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u, row = lane & 3u;
	// a wave covers 64 vertices in 4 passes of 16 blocks
	const uint32_t wave_v0 = (blockIdx.x * THREADS + (threadIdx.x & ~63u));
	float acc[4] = {0, 0, 0, 0};
	for (int pass = 0; pass < 4; ++pass) {
		const uint32_t v = wave_v0 + pass * 16 + (lane >> 2);
		Vtx c = vtx[v];
		for (int rep = 0; rep < reps; ++rep) {
			asm volatile("" : "+v"(c.px), "+v"(c.py), "+v"(c.pz), "+v"(c.w[0]), "+v"(c.w[1]), "+v"(c.w[2]), "+v"(c.w[3]));
			v4f d = {0, 0, 0, 0};
			const float p4[4] = {c.px, c.py, c.pz, 1.0f};
#pragma unroll
			for (int j = 0; j < 4; ++j) {
#pragma unroll
				for (int col = 0; col < 4; ++col) {
					const float a = row < 3u ? s_pal[c.idx[j] * 12 + row * 4 + col] : 0.0f; // P[idx_j][row][col]
					const float b = row == 0u ? c.w[j] * p4[col] : 0.0f;                      // only column 0 of the block is wanted
					d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 0, 0, 0);
				}
			}
			// D[i][0]: lane 4 b + 0 holds column 0 of block b, register i = row i
			acc[0] += d[0]; acc[1] += d[1]; acc[2] += d[2];
			c.px += 1e-7f;
		}
		if (row == 0u) {
			out[3 * v] = acc[0]; out[3 * v + 1] = acc[1]; out[3 * v + 2] = acc[2];
		}
		acc[0] = acc[1] = acc[2] = 0;
	}
}

int main(int argc, char** argv) {
	const int n = 256 * 1024, reps = argc > 1 ? atoi(argv[1]) : 64;
	std::vector<float> pal(N_BONES * 12);
	std::vector<Vtx> vt(n);
	srand(7);
	for (float& x : pal) x = (float)(rand() / (double)RAND_MAX) * 2 - 1;
	for (Vtx& v : vt) {
		v.px = (float)(rand() / (double)RAND_MAX); v.py = (float)(rand() / (double)RAND_MAX); v.pz = (float)(rand() / (double)RAND_MAX);
		float s = 0;
		for (int j = 0; j < 4; ++j) { v.w[j] = (float)(rand() / (double)RAND_MAX) + 0.01f; s += v.w[j]; v.idx[j] = rand() % N_BONES; }
		for (int j = 0; j < 4; ++j) v.w[j] /= s;
	}
	float4* d_pal; Vtx* d_v; float *d_o1, *d_o2;
	CK(hipMalloc(&d_pal, pal.size() * 4)); CK(hipMalloc(&d_v, vt.size() * sizeof(Vtx))); CK(hipMalloc(&d_o1, n * 12)); CK(hipMalloc(&d_o2, n * 12));
	CK(hipMemcpy(d_pal, pal.data(), pal.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_v, vt.data(), vt.size() * sizeof(Vtx), hipMemcpyHostToDevice));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	float ms_valu = 0, ms_mfma = 0;
	for (int it = 0; it < 3; ++it) {
		CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_blend_valu, dim3(n / THREADS), dim3(THREADS), 0, 0, d_pal, d_v, d_o1, reps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		CK(hipEventElapsedTime(&ms_valu, e0, e1));
		CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_blend_mfma, dim3(n / THREADS), dim3(THREADS), 0, 0, d_pal, d_v, d_o2, reps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		CK(hipEventElapsedTime(&ms_mfma, e0, e1));
	}
	std::vector<float> o1(n * 3), o2(n * 3);
	CK(hipMemcpy(o1.data(), d_o1, n * 12, hipMemcpyDeviceToHost)); CK(hipMemcpy(o2.data(), d_o2, n * 12, hipMemcpyDeviceToHost));
	double max_rel = 0, scale = 0; // largest difference relative to the largest output (outputs are sums over `reps` skinned positions)
	for (int i = 0; i < n * 3; ++i) scale = fmax(scale, fabs(o1[i]));
	for (int i = 0; i < n * 3; ++i) max_rel = fmax(max_rel, fabs(o1[i] - o2[i]) / scale);
	const double per = (double)n * reps;
	printf("{\"vertices\": %d, \"reps\": %d, \"valu_ms\": %.4f, \"mfma_ms\": %.4f, \"valu_ns_per_vertex\": %.4f, \"mfma_ns_per_vertex\": %.4f, \"mfma_over_valu\": %.2f, \"max_diff_over_max_output\": %.3g}\n",
		n, reps, ms_valu, ms_mfma, ms_valu * 1e6 / per, ms_mfma * 1e6 / per, ms_mfma / ms_valu, max_rel);
	return 0;
}
