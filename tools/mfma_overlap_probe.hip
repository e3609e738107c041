// mfma_overlap_probe.hip — do MFMA instructions run beside the vector ALU on gfx950, per data type?
//
// k_cull_tile<F = 0>'s pre-test (cull_kernels.hip) moved the sphere x plane arithmetic onto v_mfma_f32_32x32x2_f32; the launch's VALU
// instruction count fell by 38 % and its duration by 5 %. This probe asks why: every SIMD of a CU gets two waves, one that issues only
// MFMAs (f32-input 32x32x2, or bf16 32x32x16) and one that issues only independent v_fma_f32. Timed three ways - MFMA waves alone, VALU
// waves alone, both together: "together ~ max" means separate pipes, "together ~ sum" means the instructions share the execution unit.
//
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_overlap_probe.hip -o tools/_build/mfma_overlap_probe && tools/_build/mfma_overlap_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 512 threads = 8 waves = 2 per SIMD. Waves 0..3 (one per SIMD) are the MFMA waves, 4..7 the VALU waves.
template <int KIND> // 0: f32-input MFMA, 1: bf16 MFMA
__global__ __launch_bounds__(512) void k_overlap(float* out, int n_mfma, int n_valu, int mode /* 1 = MFMA waves work, 2 = VALU waves work, 3 = both */) {
	const int wave = threadIdx.x >> 6;
	float sink = 0.f;
	if (wave < 4) {
		if (mode & 1) {
			f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
			const float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)blockIdx.x * 1e-6f;
			bf16x8 ab, bb;
			for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(a + i); bb[i] = (__bf16)(b - i); }
			for (int i = 0; i < n_mfma; i += 4) { // four independent accumulators: no dependent-accumulator stalls
				if constexpr (KIND == 0) {
					acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
					acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
					acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
					acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
				} else {
					acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc0, 0, 0, 0);
					acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc1, 0, 0, 0);
					acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc2, 0, 0, 0);
					acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc3, 0, 0, 0);
				}
			}
			sink = acc0[0] + acc1[1] + acc2[2] + acc3[3];
		}
	} else if (mode & 2) {
		float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
		const float m = 1.0000001f, c = 1e-9f;
		for (int i = 0; i < n_valu; i += 8) { // eight independent chains
			v0 = __builtin_fmaf(v0, m, c); v1 = __builtin_fmaf(v1, m, c); v2 = __builtin_fmaf(v2, m, c); v3 = __builtin_fmaf(v3, m, c);
			v4 = __builtin_fmaf(v4, m, c); v5 = __builtin_fmaf(v5, m, c); v6 = __builtin_fmaf(v6, m, c); v7 = __builtin_fmaf(v7, m, c);
		}
		sink = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
	}
	if (sink == 123.456f) out[threadIdx.x] = sink;
}

// One stream per wave: an MFMA followed by FILL independent v_fma_f32 (8 chains), four accumulators in rotation. How many fillers ride in
// the shadow of an MFMA? (time per iteration against FILL: flat while they are hidden, then 1 issue slot per filler)
template <int KIND, int FILL>
__global__ __launch_bounds__(512) void k_interleave(float* out, int n_iter) {
	f32x16 acc[4] = {};
	const float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)blockIdx.x * 1e-6f;
	bf16x8 ab, bb;
	for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(a + i); bb[i] = (__bf16)(b - i); }
	float v[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, (float)threadIdx.x};
	const float m = 1.0000001f, c = 1e-9f;
	for (int i = 0; i < n_iter; i += 4) {
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			if constexpr (KIND == 0) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
			else acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[k], 0, 0, 0);
#pragma unroll
			for (int f = 0; f < FILL; ++f) v[f & 7] = __builtin_fmaf(v[f & 7], m, c);
			__builtin_amdgcn_sched_barrier(0);
		}
	}
	float sink = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
	for (int f = 0; f < 8; ++f) sink += v[f];
	if (sink == 123.456f) out[threadIdx.x] = sink;
}

template <int KIND, int FILL> static int run_fill(const char* name, int threads, int n_iter, float* d_out) {
	hipEvent_t a, b;
	CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	float ms = 0;
	for (int rep = 0; rep < 3; ++rep) {
		CK(hipEventRecord(a, 0));
		hipLaunchKernelGGL((k_interleave<KIND, FILL>), dim3(256), dim3(threads), 0, 0, d_out, n_iter);
		CK(hipEventRecord(b, 0));
		CK(hipEventSynchronize(b));
		CK(hipEventElapsedTime(&ms, a, b));
	}
	printf("%-26s %d wave(s) per SIMD, MFMA + %2d v_fma_f32: %7.1f ns per iteration = %6.1f cycles at 2.4 GHz\n", name, threads / 256, FILL, 1e6 * ms / n_iter, 2.4e6 * ms / n_iter);
	return 0;
}

template <int KIND> static int run(const char* name, int n_mfma, int n_valu, float* d_out) {
	hipEvent_t a, b;
	CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	float ms[4] = {};
	for (int mode = 1; mode <= 3; ++mode) {
		for (int rep = 0; rep < 3; ++rep) {
			CK(hipEventRecord(a, 0));
			hipLaunchKernelGGL(k_overlap<KIND>, dim3(256), dim3(512), 0, 0, d_out, n_mfma, n_valu, mode);
			CK(hipEventRecord(b, 0));
			CK(hipEventSynchronize(b));
			CK(hipEventElapsedTime(&ms[mode], a, b));
		}
	}
	printf("%-28s %d MFMA per wave alone %.3f ms | %d v_fma_f32 per wave alone %.3f ms | together %.3f ms  (max %.3f, sum %.3f)\n", name, n_mfma, ms[1], n_valu, ms[2], ms[3],
		ms[1] > ms[2] ? ms[1] : ms[2], ms[1] + ms[2]);
	return 0;
}

int main() {
	float* d_out;
	CK(hipMalloc(&d_out, 4096));
	// sized so that both kinds of wave are busy for about the same time alone
	if (run<0>("v_mfma_f32_32x32x2_f32", 40000, 640000, d_out)) return 2;
	if (run<1>("v_mfma_f32_32x32x16_bf16", 80000, 640000, d_out)) return 2;
	if (run<0>("v_mfma_f32_32x32x2_f32", 40000, 1280000, d_out)) return 2;
	if (run<1>("v_mfma_f32_32x32x16_bf16", 80000, 1280000, d_out)) return 2;
#define FILLS(KIND, NAME, T) \
	if (run_fill<KIND, 0>(NAME, T, 40000, d_out) || run_fill<KIND, 2>(NAME, T, 40000, d_out) || run_fill<KIND, 4>(NAME, T, 40000, d_out) || run_fill<KIND, 6>(NAME, T, 40000, d_out) || \
		run_fill<KIND, 8>(NAME, T, 40000, d_out) || run_fill<KIND, 12>(NAME, T, 40000, d_out) || run_fill<KIND, 16>(NAME, T, 40000, d_out) || run_fill<KIND, 24>(NAME, T, 40000, d_out)) return 2;
	FILLS(0, "v_mfma_f32_32x32x2_f32", 256)
	FILLS(1, "v_mfma_f32_32x32x16_bf16", 256)
	FILLS(0, "v_mfma_f32_32x32x2_f32", 512)
	FILLS(1, "v_mfma_f32_32x32x16_bf16", 512)
	return 0;
}
