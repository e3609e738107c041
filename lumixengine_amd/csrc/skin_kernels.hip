// skin_kernels.hip — skeletal skinning on gfx950: absolute pose, matrix palette, linear-blend vertex transform.
//
//   k_pose_palette   one wave per group of up to 16 instances of one model. Pose::computeAbsolute (src/renderer/
//                    pose.cpp:63-134, scalar recurrence :129-130) is a chain over the bone tree: a bone only depends on
//                    its parent's final value, so the wave walks the tree level by level with the group's poses in
//                    LDS, lanes spread over (instance, bone of that level) pairs; every bone is computed by exactly the
//                    reference's operations, so the result is bit-identical to the index-order loop. Then
//                    computeSkinMatrices (src/renderer/model.cpp:132-137): palette[i] = (pose[i] * inverse_bind[i])
//                    .toMatrix(), written as 4 x float4 per bone (+ optionally the dual-quaternion palette).
//   k_skin_vertices  evaluateSkin (model.cpp:103-109): the instance's palette is staged in LDS as 3 rows x float4
//                    per bone (row w of the blended matrix never reaches transformPoint, core/math.cpp:1231-1235),
//                    replicated per bank column so the random bone-matrix reads are conflict-free; each lane blends
//                    its 4 bone matrices element-wise in the reference's left-to-right order and transforms its
//                    vertices. FMA-free VALU, bit-exact with the reference.
#include "lmx_kernels.h"

namespace lmx {

namespace {

constexpr int SKIN_MAX_BONES = 196; // Model::Bone::MAX_COUNT, renderer/model.h:155

// Lanes of ONE wave exchange data through LDS: the LDS executes a wave's instructions in issue order, so only the
// compiler has to be kept from reordering the accesses (wavefront-scope fences emit no instructions).
__device__ __forceinline__ void wave_lds_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One wave per GROUP of up to K consecutive instances of one model (K = 16 / 8 / 4 for <= 64 / 128 / 196 bones). A bone
// only depends on its parent's final value, so the tree is walked level by level; the work items of a level are the
// (instance, bone at that depth) pairs of the whole group, which fills the lanes even though a level of one skeleton holds
// only a handful of bones (one instance per wave kept ~4 of 64 lanes busy and made this kernel VALU-bound). Poses live in
// LDS bone-major ([bone][instance]) so that neighbouring lanes (instances) touch neighbouring banks. Every bone is computed by
// exactly the reference's operations (pose.cpp:129-130), so the result is bit-identical to the index-order loop.
constexpr int POSE_LDS_BONES = 1024; // K * n_bones <= 1024: 16 x 64, 8 x 128, 4 x 196

__device__ __forceinline__ uint32_t pose_group_capacity(uint32_t n_bones) { return n_bones <= 64 ? 16u : (n_bones <= 128 ? 8u : 4u); }

__global__ __launch_bounds__(64) void k_pose_palette(const SkinInstance* __restrict__ inst, const PoseGroup* __restrict__ groups,
	const float* rel_pos, const float4* rel_rot, float* pose_pos, float4* pose_rot /* rel_* may alias pose_*: no __restrict__ */,
	const int16_t* __restrict__ parents, const uint16_t* __restrict__ level_bones, const uint16_t* __restrict__ level_off,
	const float* __restrict__ inv_pos, const float4* __restrict__ inv_rot, float4* __restrict__ palette, float4* __restrict__ dual_quats) {
	__shared__ float4 s_rot[POSE_LDS_BONES];
	__shared__ float s_pos[POSE_LDS_BONES * 3];
	__shared__ int32_t s_parent[SKIN_MAX_BONES];
	const uint32_t lane = threadIdx.x;
	const PoseGroup g = groups[blockIdx.x];
	const SkinInstance in = inst[g.first_inst]; // all instances of the group share the model; their bones are consecutive in memory
	const uint32_t nb = in.n_bones;
	const uint32_t K = pose_group_capacity(nb);
	const uint32_t kshift = K == 16 ? 4u : (K == 8 ? 3u : 2u);
	const size_t bone0 = in.bone_offset;
	// stage relative poses: coalesced per instance, transposed to [bone][instance] in LDS
	for (uint32_t k = 0; k < g.count; ++k) {
		const size_t base = bone0 + (size_t)k * nb;
		for (uint32_t b = lane; b < nb; b += 64) {
			s_rot[b * K + k] = rel_rot[base + b];
			const float* p = rel_pos + (base + b) * 3;
			s_pos[(b * K + k) * 3] = p[0];
			s_pos[(b * K + k) * 3 + 1] = p[1];
			s_pos[(b * K + k) * 3 + 2] = p[2];
		}
	}
	for (uint32_t b = lane; b < nb; b += 64) s_parent[b] = parents[in.model_offset + b];
	wave_lds_sync();
	const uint16_t* lv_off = level_off + in.lv_off_offset;     // lv_off[d - 1] .. lv_off[d]: bones of depth d
	const uint16_t* lv_bones = level_bones + in.lv_bones_offset; // bones >= first_nonroot, sorted by depth
	for (uint32_t d = 1; d <= in.max_depth; ++d) {
		const uint32_t start = lv_off[d - 1];
		const uint32_t items = ((uint32_t)lv_off[d] - start) << kshift;
		for (uint32_t j = lane; j < items; j += 64) {
			const uint32_t k = j & (K - 1);
			if (k < g.count) {
				const uint32_t b = lv_bones[start + (j >> kshift)];
				const uint32_t ib = b * K + k, ip = (uint32_t)s_parent[b] * K + k;
				const float4 pr4 = s_rot[ip];
				const float4 r4 = s_rot[ib];
				const Q4 pr = Q4{pr4.x, pr4.y, pr4.z, pr4.w};
				const V3 np = add(rotate(pr, V3{s_pos[3 * ib], s_pos[3 * ib + 1], s_pos[3 * ib + 2]}), V3{s_pos[3 * ip], s_pos[3 * ip + 1], s_pos[3 * ip + 2]});
				const Q4 nr = qmul(pr, Q4{r4.x, r4.y, r4.z, r4.w});
				s_pos[3 * ib] = np.x; s_pos[3 * ib + 1] = np.y; s_pos[3 * ib + 2] = np.z;
				s_rot[ib] = make_float4(nr.x, nr.y, nr.z, nr.w);
			}
		}
		wave_lds_sync();
	}
	// palette (computeSkinMatrices), optional dual quaternions, absolute pose write-back: coalesced per instance
	const float* ipos = inv_pos + (size_t)in.model_offset * 3;
	const float4* irot = inv_rot + in.model_offset;
	for (uint32_t k = 0; k < g.count; ++k) {
		const size_t base = bone0 + (size_t)k * nb;
		for (uint32_t b = lane; b < nb; b += 64) {
			const uint32_t ib = b * K + k;
			const float4 r4 = s_rot[ib];
			const float4 ir = irot[b];
			const V3 p = V3{s_pos[3 * ib], s_pos[3 * ib + 1], s_pos[3 * ib + 2]};
			const V3 ip = V3{ipos[3 * b], ipos[3 * b + 1], ipos[3 * b + 2]};
			const Mat4 m = skin_matrix(p, Q4{r4.x, r4.y, r4.z, r4.w}, ip, Q4{ir.x, ir.y, ir.z, ir.w});
			if (dual_quats != nullptr) { // the palette format of the reference's own GPU skinning path (32 B per bone)
				const DualQ dq = skin_dual_quat(p, Q4{r4.x, r4.y, r4.z, r4.w}, ip, Q4{ir.x, ir.y, ir.z, ir.w});
				float4* o = dual_quats + (base + b) * 2;
				o[0] = make_float4(dq.r.x, dq.r.y, dq.r.z, dq.r.w);
				o[1] = make_float4(dq.d.x, dq.d.y, dq.d.z, dq.d.w);
			}
			float4* out = palette + (base + b) * 4;
			out[0] = make_float4(m.c[0][0], m.c[0][1], m.c[0][2], m.c[0][3]);
			out[1] = make_float4(m.c[1][0], m.c[1][1], m.c[1][2], m.c[1][3]);
			out[2] = make_float4(m.c[2][0], m.c[2][1], m.c[2][2], m.c[2][3]);
			out[3] = make_float4(m.c[3][0], m.c[3][1], m.c[3][2], m.c[3][3]);
			// the pose becomes absolute (Pose::is_absolute = true, pose.cpp:133)
			float* gp = pose_pos + (base + b) * 3;
			gp[0] = p.x; gp[1] = p.y; gp[2] = p.z;
			pose_rot[base + b] = r4;
		}
	}
}

// ---- linear-blend skinning ----------------------------------------------------------------------------------
// One block skins a tile of one instance's vertices with that instance's palette staged in LDS as 3 rows x float4 per
// bone. The bone indices of neighbouring vertices are unrelated in the worst case, so the 12 ds_read_b128 per vertex
// would collide on LDS banks (16 random 16-B slots per service group -> ~3x serialisation). The palette is therefore
// REPLICATED: copy c of every row lives in 16-B bank column c (slot = (bone*3 + row) * COPIES + c) and lane l reads copy
// l % COPIES. With 16 copies every lane of a ds_read_b128 service group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...:
// 16 lanes with distinct l % 16, MI355X_MICROARCH.md LDS table) owns its own bank column -> conflict-free whatever the
// indices. 48 KiB per block (64 bones x 16 copies, 128 x 8, 196 x 4), staged once per tile of thousands of vertices.
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int SKIN_THREADS = 512;
constexpr int SKIN_LDS_SLOTS = 3072; // float4 slots = 48 KiB

template <int COPIES, bool EXACT>
__device__ __forceinline__ void skin_tile(const SkinInstance& in, uint32_t v_begin, uint32_t v_end, float4* s_rows,
	const float* __restrict__ verts, const float4* __restrict__ weights, const int16_t* __restrict__ indices,
	const float4* __restrict__ palette, float* __restrict__ out) {
	// stage: global column-major 4 x float4 per bone -> LDS rows {c0[r], c1[r], c2[r], c3[r]}, COPIES times
	const float4* pal = palette + (size_t)in.bone_offset * 4;
	for (uint32_t e = threadIdx.x; e < in.n_bones * COPIES; e += SKIN_THREADS) {
		const uint32_t b = e / COPIES, c = e % COPIES;
		const float4 c0 = pal[4 * b], c1 = pal[4 * b + 1], c2 = pal[4 * b + 2], c3 = pal[4 * b + 3];
		s_rows[(3 * b) * COPIES + c] = make_float4(c0.x, c1.x, c2.x, c3.x);
		s_rows[(3 * b + 1) * COPIES + c] = make_float4(c0.y, c1.y, c2.y, c3.y);
		s_rows[(3 * b + 2) * COPIES + c] = make_float4(c0.z, c1.z, c2.z, c3.z);
	}
	__syncthreads();
	const uint32_t col = threadIdx.x & (COPIES - 1);
	const float4* rows = s_rows + col;
	// per-instance base pointers are wave-uniform (SGPRs); per-lane offsets stay 32-bit
	struct F3 { float x, y, z; }; // 12-byte records: loaded / stored as one dwordx3 per lane
	const F3* vbase = reinterpret_cast<const F3*>(verts) + in.vert_offset;
	const float4* wbase = weights + in.vert_offset;
	const int2* ibase = reinterpret_cast<const int2*>(indices) + in.vert_offset; // 4 x i16 per vertex, little endian
	F3* obase = reinterpret_cast<F3*>(out) + in.out_offset;
	struct VertexIn { float px, py, pz; float4 w; int2 iw; };
	auto load = [&](uint32_t v) {
		VertexIn r;
		const F3 p = vbase[v];
		r.px = p.x;
		r.py = p.y;
		r.pz = p.z;
		r.w = wbase[v];
		r.iw = ibase[v];
		return r;
	};
	auto skin_one = [&](const VertexIn& c, uint32_t v) {
		// bone indices are non-negative i16 (validated at lmx_skin_add_mesh): plain 16-bit fields, no sign extension
		const float4* r0 = rows + (uint32_t)(c.iw.x & 0xffff) * (3 * COPIES);
		const float4* r1 = rows + ((uint32_t)c.iw.x >> 16) * (3 * COPIES);
		const float4* r2 = rows + (uint32_t)(c.iw.y & 0xffff) * (3 * COPIES);
		const float4* r3 = rows + ((uint32_t)c.iw.y >> 16) * (3 * COPIES);
		const float4 w = c.w;
		float o[3];
#pragma unroll
		for (int r = 0; r < 3; ++r) {
			const float4 A = r0[r * COPIES], B = r1[r * COPIES], C = r2[r * COPIES], D = r3[r * COPIES];
			if constexpr (EXACT) {
				// Matrix::operator*(float) and operator+ (math.cpp:1022-1071), left to right: ((A*w.x + B*w.y) + C*w.z) + D*w.w
				const float m0 = A.x * w.x + B.x * w.y + C.x * w.z + D.x * w.w;
				const float m1 = A.y * w.x + B.y * w.y + C.y * w.z + D.y * w.w;
				const float m2 = A.z * w.x + B.z * w.y + C.z * w.z + D.z * w.w;
				const float m3 = A.w * w.x + B.w * w.y + C.w * w.z + D.w * w.w;
				// Matrix::transformPoint (math.cpp:1231-1235): c0.r*p.x + c1.r*p.y + c2.r*p.z + c3.r
				o[r] = m0 * c.px + m1 * c.py + m2 * c.pz + m3;
			} else {
				// same association with the products fused into the adds, on register pairs (v_pk_fma_f32): the four
				// floats of an LDS row land in consecutive VGPRs, so {x,y} and {z,w} are packed operands as they are.
				const v2f a01 = {A.x, A.y}, a23 = {A.z, A.w}, b01 = {B.x, B.y}, b23 = {B.z, B.w};
				const v2f c01 = {C.x, C.y}, c23 = {C.z, C.w}, d01 = {D.x, D.y}, d23 = {D.z, D.w};
				// weight splats as shuffles of the loaded pairs, so that the broadcast folds into op_sel of v_pk_*_f32
				const v2f w01 = {w.x, w.y}, w23 = {w.z, w.w};
				const v2f wx = __builtin_shufflevector(w01, w01, 0, 0), wy = __builtin_shufflevector(w01, w01, 1, 1);
				const v2f wz = __builtin_shufflevector(w23, w23, 0, 0), ww = __builtin_shufflevector(w23, w23, 1, 1);
				v2f m01 = a01 * wx, m23 = a23 * wx;
				m01 = __builtin_elementwise_fma(b01, wy, m01);
				m23 = __builtin_elementwise_fma(b23, wy, m23);
				m01 = __builtin_elementwise_fma(c01, wz, m01);
				m23 = __builtin_elementwise_fma(c23, wz, m23);
				m01 = __builtin_elementwise_fma(d01, ww, m01);
				m23 = __builtin_elementwise_fma(d23, ww, m23);
				o[r] = fmaf(m23.x, c.pz, fmaf(m01.y, c.py, m01.x * c.px)) + m23.y;
			}
		}
		obase[v] = F3{o[0], o[1], o[2]};
	};
	// software-pipelined and unrolled by two (A / B ping-pong): the next vertex's loads are in flight while the current one
	// is blended, and no register block is copied between iterations
	uint32_t v = v_begin + threadIdx.x;
	if (v >= v_end) return;
	VertexIn a = load(v);
	for (;;) {
		const uint32_t vb = v + SKIN_THREADS;
		const bool has_b = vb < v_end;
		VertexIn b = a;
		if (has_b) b = load(vb);
		skin_one(a, v);
		if (!has_b) break;
		const uint32_t va = vb + SKIN_THREADS;
		const bool has_a = va < v_end;
		if (has_a) a = load(va);
		skin_one(b, vb);
		if (!has_a) break;
		v = va;
	}
}

template <bool EXACT>
__global__ __launch_bounds__(SKIN_THREADS, 6) void k_skin_vertices(const SkinInstance* __restrict__ inst, uint32_t tiles_per_inst,
	uint32_t tile_verts, const float* __restrict__ verts, const float4* __restrict__ weights, const int16_t* __restrict__ indices,
	const float4* __restrict__ palette, float* __restrict__ out) {
	__shared__ float4 s_rows[SKIN_LDS_SLOTS];
	const uint32_t ii = blockIdx.x / tiles_per_inst;
	const uint32_t tile = blockIdx.x - ii * tiles_per_inst;
	const SkinInstance in = inst[ii];
	const uint32_t v_begin = tile * tile_verts;
	if (v_begin >= in.n_verts) return; // block-uniform
	const uint32_t v_end = min(v_begin + tile_verts, in.n_verts);
	if (in.n_bones <= 64) skin_tile<16, EXACT>(in, v_begin, v_end, s_rows, verts, weights, indices, palette, out);
	else if (in.n_bones <= 128) skin_tile<8, EXACT>(in, v_begin, v_end, s_rows, verts, weights, indices, palette, out);
	else skin_tile<4, EXACT>(in, v_begin, v_end, s_rows, verts, weights, indices, palette, out);
}

} // namespace

hipError_t launch_pose_palette(hipStream_t s, const SkinInstance* inst, const PoseGroup* groups, uint32_t n_groups, const float* rel_pos,
	const float4* rel_rot, float* pose_pos, float4* pose_rot, const int16_t* parents, const uint16_t* level_bones, const uint16_t* level_off,
	const float* inv_pos, const float4* inv_rot, float4* palette, float4* dual_quats) {
	if (!n_groups) return hipSuccess;
	hipLaunchKernelGGL(k_pose_palette, dim3(n_groups), dim3(64), 0, s, inst, groups, rel_pos, rel_rot, pose_pos, pose_rot, parents, level_bones,
		level_off, inv_pos, inv_rot, palette, dual_quats);
	return hipGetLastError();
}

hipError_t launch_skin_vertices(hipStream_t s, const SkinInstance* inst, uint32_t n_inst, uint32_t max_verts, const float* verts,
	const float4* weights, const int16_t* indices, const float4* palette, float* out, bool exact) {
	if (!n_inst || !max_verts) return hipSuccess;
	// tiles: as large as possible (the 48 KiB palette staging is paid per tile) while still giving the chip >= ~3000 blocks
	const uint32_t max_tiles = (max_verts + 1023u) / 1024u;
	uint32_t tiles = (3072u + n_inst - 1) / n_inst;
	if (tiles > max_tiles) tiles = max_tiles;
	if (tiles < 1) tiles = 1;
	const uint32_t tile_verts = (max_verts + tiles - 1) / tiles;
	const uint64_t blocks = (uint64_t)tiles * n_inst;
	if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
	if (exact) {
		hipLaunchKernelGGL(k_skin_vertices<true>, dim3((uint32_t)blocks), dim3(SKIN_THREADS), 0, s, inst, tiles, tile_verts, verts, weights,
			indices, palette, out);
	} else {
		hipLaunchKernelGGL(k_skin_vertices<false>, dim3((uint32_t)blocks), dim3(SKIN_THREADS), 0, s, inst, tiles, tile_verts, verts, weights,
			indices, palette, out);
	}
	return hipGetLastError();
}

} // namespace lmx
