// mfma_contract_probe.hip — the two hardware contracts the several-frusta cull kernel's MFMA pre-test builds on (k_cull_tile<F = 0>,
// cull_kernels.hip "sphere x plane pre-test on the matrix pipe"), checked on the device against a host restatement:
//
//   1. v_mfma_f32_32x32x2_f32 is, bit for bit, the k-ordered chain D = fmaf(a_k1, b_k1, fmaf(a_k0, b_k0, C)) with the operand maps
//      A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31], D[row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)][col = lane & 31]
//      (MI355X guide, "FP32-input MFMA"): two of them in a row are the 4-term chain fmaf(r, 1, fmaf(z, nz, fmaf(y, ny, fmaf(x, nx, d)))),
//      including subnormal, infinite and NaN operands (tests/tests/hostsim emulates the instruction as exactly that chain; the product kernel uses the bf16 form of check 3, the f32 form is kept here as the measured alternative).
//   2. v_permlane32_swap_b32 vdst, src: lanes 32..63 of vdst trade places with lanes 0..31 of src.
//
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/mfma_contract_probe.hip -o tools/_build/mfma_contract_probe && tools/_build/mfma_contract_probe
//
// Prints one line per check and "mfma_contract_probe: OK" (exit 0) or the first mismatches (exit 1).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 3. v_mfma_f32_32x32x16_bf16: A[i = lane & 31][k = 8 (lane >> 5) + e], B[k = 8 (lane >> 5) + e][j = lane & 31], element e of the 8-vector in
//    bits 16 (e & 1) of register e >> 1; same C / D map. Products of bf16 pairs are exact in fp32; what the accumulation of the 16 products
//    and C does internally is not documented - the probe reports the worst |device - exact| / (|C| + sum |a b|) in units of u = 2^-24.
//    The operands are made the way the kernel makes them: __builtin_convertvector(float2 -> bf16x2) (v_cvt_pk_bf16_f32, round to nearest even,
//    first element in the low half).
__global__ __launch_bounds__(64) void k_mfma_bf16(const float* __restrict__ a /* [case][64][8] */, const float* __restrict__ b, const float* __restrict__ c, float* __restrict__ d,
	uint32_t* __restrict__ a_bits /* [case][64][4]: the packed operand as the device formed it */, uint32_t* __restrict__ b_bits) {
	const uint32_t lane = threadIdx.x, w = blockIdx.x;
	f32x16 acc;
#pragma unroll
	for (int r = 0; r < 16; ++r) acc[r] = c[(w * 64 + lane) * 16 + r];
	u32x4 ap, bp;
#pragma unroll
	for (int e = 0; e < 8; e += 2) {
		const f32x2 av = {a[(w * 64 + lane) * 8 + e], a[(w * 64 + lane) * 8 + e + 1]}, bv = {b[(w * 64 + lane) * 8 + e], b[(w * 64 + lane) * 8 + e + 1]};
		ap[e >> 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(av, bf16x2));
		bp[e >> 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(bv, bf16x2));
	}
	acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ap), __builtin_bit_cast(bf16x8, bp), acc, 0, 0, 0);
#pragma unroll
	for (int r = 0; r < 16; ++r) d[(w * 64 + lane) * 16 + r] = acc[r];
#pragma unroll
	for (int e = 0; e < 4; ++e) { a_bits[(w * 64 + lane) * 4 + e] = ap[e]; b_bits[(w * 64 + lane) * 4 + e] = bp[e]; }
}

// one wave per case: a0/b0 = operands of the first instruction, a1/b1 of the second, c = the accumulator input (16 per lane)
__global__ __launch_bounds__(64) void k_mfma_chain(const float* __restrict__ a0, const float* __restrict__ b0, const float* __restrict__ a1, const float* __restrict__ b1,
	const float* __restrict__ c, float* __restrict__ d) {
	const uint32_t lane = threadIdx.x, w = blockIdx.x;
	f32x16 acc;
#pragma unroll
	for (int r = 0; r < 16; ++r) acc[r] = c[(w * 64 + lane) * 16 + r];
	acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[w * 64 + lane], b0[w * 64 + lane], acc, 0, 0, 0);
	acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[w * 64 + lane], b1[w * 64 + lane], acc, 0, 0, 0);
#pragma unroll
	for (int r = 0; r < 16; ++r) d[(w * 64 + lane) * 16 + r] = acc[r];
}

__global__ __launch_bounds__(64) void k_swap(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t* __restrict__ out) {
	const uint32_t lane = threadIdx.x;
	const auto r = __builtin_amdgcn_permlane32_swap(a[lane], b[lane], false, false);
	out[lane] = r[0];
	out[64 + lane] = r[1];
}

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static uint16_t bf16_rne(float f) { // round to nearest even, as v_cvt_pk_bf16_f32 (NaN stays NaN)
	uint32_t u; memcpy(&u, &f, 4);
	if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
	return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float bf16_to_float(uint16_t h) { const uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static float from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int main() {
	const int CASES = 512;
	std::mt19937 rng(12345);
	std::uniform_real_distribution<float> uni(-1.f, 1.f);
	std::vector<float> a0(CASES * 64), b0(CASES * 64), a1(CASES * 64), b1(CASES * 64), c(CASES * 64 * 16), d(CASES * 64 * 16);
	const float specials[] = {0.f, -0.f, 1e-39f, -3e-41f, 1e-45f, INFINITY, -INFINITY, NAN, 3e38f, -3e38f, 1.17549435e-38f, 16777216.f, 1e-20f, -1e20f};
	const int n_special = (int)(sizeof(specials) / sizeof(specials[0]));
	for (int w = 0; w < CASES; ++w) {
		// case classes: 0 plain magnitudes (cell-relative spheres x unit normals + a plane distance), 1 wide exponent range, 2 specials sprinkled in
		const int cls = w % 3;
		auto draw = [&](float scale) {
			float v = uni(rng) * scale;
			if (cls == 1) v = ldexpf(uni(rng), (int)(uni(rng) * 60.f));
			if (cls == 2 && (rng() % 5u) == 0u) v = specials[rng() % (uint32_t)n_special];
			return v;
		};
		for (int l = 0; l < 64; ++l) {
			a0[w * 64 + l] = draw(1.f); a1[w * 64 + l] = draw(1.f);
			b0[w * 64 + l] = draw(300.f); b1[w * 64 + l] = draw(300.f);
			for (int r = 0; r < 16; ++r) c[(w * 64 + l) * 16 + r] = draw(20000.f);
		}
	}
	float *da0, *db0, *da1, *db1, *dc, *dd;
	CK(hipMalloc(&da0, a0.size() * 4)); CK(hipMalloc(&db0, b0.size() * 4)); CK(hipMalloc(&da1, a1.size() * 4)); CK(hipMalloc(&db1, b1.size() * 4));
	CK(hipMalloc(&dc, c.size() * 4)); CK(hipMalloc(&dd, d.size() * 4));
	CK(hipMemcpy(da0, a0.data(), a0.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db0, b0.data(), b0.size() * 4, hipMemcpyHostToDevice));
	CK(hipMemcpy(da1, a1.data(), a1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db1, b1.data(), b1.size() * 4, hipMemcpyHostToDevice));
	CK(hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(k_mfma_chain, dim3(CASES), dim3(64), 0, 0, da0, db0, da1, db1, dc, dd);
	CK(hipDeviceSynchronize());
	CK(hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost));
	long bad = 0, nan_cases = 0, checked = 0;
	for (int w = 0; w < CASES; ++w) {
		for (int l = 0; l < 64; ++l) {
			for (int r = 0; r < 16; ++r) {
				const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
				float acc = c[(w * 64 + l) * 16 + r];
				// first instruction: k = 0 then k = 1; A[row][k] sits in lane row + 32 k, B[k][col] in lane col + 32 k
				acc = fmaf(a0[w * 64 + row], b0[w * 64 + col], acc);
				acc = fmaf(a0[w * 64 + row + 32], b0[w * 64 + col + 32], acc);
				acc = fmaf(a1[w * 64 + row], b1[w * 64 + col], acc);
				acc = fmaf(a1[w * 64 + row + 32], b1[w * 64 + col + 32], acc);
				const float got = d[(w * 64 + l) * 16 + r];
				++checked;
				if (std::isnan(acc) && std::isnan(got)) { ++nan_cases; continue; } // NaN payloads are not part of the contract
				if (bits(acc) != bits(got)) {
					if (bad < 10) fprintf(stderr, "mfma chain mismatch: case %d (class %d) lane %d reg %d: device %a (%08x), fmaf chain %a (%08x)\n", w, w % 3, l, r, got, bits(got), acc, bits(acc));
					++bad;
				}
			}
		}
	}
	printf("v_mfma_f32_32x32x2_f32 x 2 == 4-term fmaf chain: %ld results, %ld NaN on both sides, %ld mismatches\n", checked, nan_cases, bad);

	std::vector<uint32_t> sa(64), sb(64), so(128);
	for (int l = 0; l < 64; ++l) { sa[l] = 0x1000u + l; sb[l] = 0x2000u + l; }
	uint32_t *dsa, *dsb, *dso;
	CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dso, 512));
	CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(k_swap, dim3(1), dim3(64), 0, 0, dsa, dsb, dso);
	CK(hipDeviceSynchronize());
	CK(hipMemcpy(so.data(), dso, 512, hipMemcpyDeviceToHost));
	long bad_swap = 0;
	for (int l = 0; l < 64; ++l) {
		const uint32_t want0 = l < 32 ? sa[l] : sb[l - 32], want1 = l < 32 ? sa[l + 32] : sb[l];
		if (so[l] != want0 || so[64 + l] != want1) {
			if (bad_swap < 6) fprintf(stderr, "permlane32_swap mismatch: lane %d got {%x, %x}, expected {%x, %x}\n", l, so[l], so[64 + l], want0, want1);
			++bad_swap;
		}
	}
	printf("v_permlane32_swap_b32: r[0] = {a.lo, b.lo}, r[1] = {a.hi, b.hi}: %ld mismatches\n", bad_swap);
	// ---- bf16 MFMA: operand maps, the conversion, and the size of the accumulation error
	long bad_cvt = 0, bad_bf = 0;
	double worst = 0.0;
	{
		const int CB = 384;
		std::vector<float> ba(CB * 64 * 8), bbv(CB * 64 * 8), bc(CB * 64 * 16), bd(CB * 64 * 16);
		std::vector<uint32_t> abits(CB * 64 * 4), bbits(CB * 64 * 4);
		for (int w = 0; w < CB; ++w) {
			const int cls = w % 3; // 0: the kernel's magnitudes (unit normals x cell-relative coordinates, C = a plane distance), 1: wide exponents with cancellation, 2: C far larger than the products
			for (int l = 0; l < 64; ++l) {
				for (int e = 0; e < 8; ++e) {
					float av = uni(rng), bv = uni(rng) * 300.f;
					if (cls == 1) { av = ldexpf(uni(rng), (int)(uni(rng) * 20.f)); bv = ldexpf(uni(rng), (int)(uni(rng) * 20.f)); }
					ba[(w * 64 + l) * 8 + e] = av; bbv[(w * 64 + l) * 8 + e] = bv;
				}
				for (int r = 0; r < 16; ++r) bc[(w * 64 + l) * 16 + r] = cls == 2 ? uni(rng) * 1e6f : (cls == 1 ? ldexpf(uni(rng), (int)(uni(rng) * 24.f)) : uni(rng) * 20000.f);
			}
		}
		float *dba, *dbb, *dbc, *dbd; uint32_t *dab, *dbbits;
		CK(hipMalloc(&dba, ba.size() * 4)); CK(hipMalloc(&dbb, bbv.size() * 4)); CK(hipMalloc(&dbc, bc.size() * 4)); CK(hipMalloc(&dbd, bd.size() * 4));
		CK(hipMalloc(&dab, abits.size() * 4)); CK(hipMalloc(&dbbits, bbits.size() * 4));
		CK(hipMemcpy(dba, ba.data(), ba.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbb, bbv.data(), bbv.size() * 4, hipMemcpyHostToDevice));
		CK(hipMemcpy(dbc, bc.data(), bc.size() * 4, hipMemcpyHostToDevice));
		hipLaunchKernelGGL(k_mfma_bf16, dim3(CB), dim3(64), 0, 0, dba, dbb, dbc, dbd, dab, dbbits);
		CK(hipDeviceSynchronize());
		CK(hipMemcpy(bd.data(), dbd, bd.size() * 4, hipMemcpyDeviceToHost));
		CK(hipMemcpy(abits.data(), dab, abits.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(bbits.data(), dbbits, bbits.size() * 4, hipMemcpyDeviceToHost));
		for (size_t i = 0; i < ba.size(); ++i) { // the conversion: element e in bits 16 (e & 1) of word e >> 1, round to nearest even
			const uint16_t got_a = (uint16_t)(abits[i / 2] >> (16 * (i & 1))), got_b = (uint16_t)(bbits[i / 2] >> (16 * (i & 1)));
			if (got_a != bf16_rne(ba[i]) || got_b != bf16_rne(bbv[i])) { if (bad_cvt < 5) fprintf(stderr, "bf16 conversion mismatch at %zu: %04x vs %04x (%a)\n", i, got_a, bf16_rne(ba[i]), ba[i]); ++bad_cvt; }
		}
		for (int w = 0; w < CB; ++w) {
			for (int l = 0; l < 64; ++l) {
				for (int r = 0; r < 16; ++r) {
					const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
					double exact = bc[(w * 64 + l) * 16 + r], mag = fabs(exact);
					for (int k = 0; k < 16; ++k) {
						const double av = bf16_to_float(bf16_rne(ba[(w * 64 + row + 32 * (k >> 3)) * 8 + (k & 7)])), bv = bf16_to_float(bf16_rne(bbv[(w * 64 + col + 32 * (k >> 3)) * 8 + (k & 7)]));
						exact += av * bv; mag += fabs(av * bv);
					}
					const double err = fabs((double)bd[(w * 64 + l) * 16 + r] - exact) / (mag * 5.9604644775390625e-08);
					if (err > worst) worst = err;
					if (err > 16.0) { if (bad_bf < 5) fprintf(stderr, "bf16 mfma: case %d (class %d) lane %d reg %d: device %a, exact %a, error %.2f u of the magnitude sum\n", w, w % 3, l, r, bd[(w * 64 + l) * 16 + r], exact, err); ++bad_bf; }
				}
			}
		}
		printf("v_mfma_f32_32x32x16_bf16: operand / result maps as assumed; worst |device - exact| = %.3f u x (|C| + sum |a b|) over %d results (bound used by the kernel: 16 u); conversion mismatches %ld\n",
			worst, CB * 64 * 16, bad_cvt);
	}
	(void)from_bits;
	if (bad || bad_swap || bad_cvt || bad_bf) { printf("mfma_contract_probe: FAILED\n"); return 1; }
	printf("mfma_contract_probe: OK\n");
	return 0;
}
