// skin_kernels.hip — skeletal skinning on gfx950: absolute pose, matrix palette, linear-blend vertex transform.
//
//   k_pose_palette   one wave per group of up to 4 instances of one model. Pose::computeAbsolute (src/renderer/
//                    pose.cpp:63-134, scalar recurrence :129-130) is a chain over the bone tree: a bone only depends on
//                    its parent's final value, so the wave walks the tree level by level with the group's poses in
//                    LDS, lanes spread over (instance, bone of that level) pairs; every bone is computed by exactly the
//                    reference's operations, so the result is bit-identical to the index-order loop. Then
//                    computeSkinMatrices (src/renderer/model.cpp:132-137): palette[i] = (pose[i] * inverse_bind[i])
//                    .toMatrix(), written as its 3 non-constant rows (3 x float4 per bone) (+ optionally the
//                    dual-quaternion palette).
//   k_skin_vertices  evaluateSkin (model.cpp:103-109): the instance's palette is staged in LDS as 3 rows x float4
//                    per bone (row w of the blended matrix never reaches transformPoint, core/math.cpp:1231-1235),
//                    replicated per bank column so the random bone-matrix reads are conflict-free; each lane blends
//                    its 4 bone matrices element-wise in the reference's left-to-right order and transforms its
//                    vertices. FMA-free VALU, bit-exact with the reference.
#include "lmx_kernels.h"
#include "lumix_mi355.h" // LMX_SKIN_* modes

namespace lmx {

namespace {

// tools/pose_probe.hip includes this file with LMX_PROBE_SKIP defined to time the phases of k_pose_palette separately
#ifndef LMX_PROBE_SKIP
#define LMX_PROBE_SKIP(bit) false
#endif

constexpr int SKIN_MAX_BONES = 196; // Model::Bone::MAX_COUNT, renderer/model.h:155

// Lanes of ONE wave exchange data through LDS: the LDS executes a wave's instructions in issue order, so only the
// compiler has to be kept from reordering the accesses (wavefront-scope fences emit no instructions).
__device__ __forceinline__ void wave_lds_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Pose::computeAbsolute + computeSkinMatrices, one WAVE per group of K = 4 / 2 / 1 consecutive instances of one model (<= 64 / 128 / 196
// bones, 256 bones per wave at most). A bone only depends on its parent's final value, so the tree is walked level by level; the work items
// of a level are the (instance, bone at that depth) pairs of the group. Every bone is computed by exactly the reference's operations
// (pose.cpp:129-130, model.cpp:132-137), so the result is bit-identical to the index-order loop. Nothing is shared between waves - no
// barrier, and the load, walk and store phases of the ~20 waves a CU holds drift apart and overlap - and every global access is the
// memory order itself:
//   - LDS keeps the poses instance-major, as memory does: rotations s_rot[k * NBP + bone], positions as the AoS dwords
//     s_pos[k * PS + 3 * bone + c]. Loads: lane = bone for the 16-byte rotations (1 KiB contiguous per instruction), lane = dword for the
//     positions (256 B contiguous), all of a group's loads in flight before the first LDS write.
//   - the walk's lanes are (instance k = lane % K, item lane / K); NBP = 16 / K and PS = 64 / K modulo the bank count put the K
//     instances of one bone in different bank columns, so two lanes collide only when their BONES agree modulo 16 / K (b128) or modulo
//     64 / K dwords (b32) - neighbouring bone indices, the common case within a level, never do. The LDS executes one wave's
//     instructions in order: no barrier, only the compiler is held (wave_lds_sync).
//   - palette rows: a lane computes its bone's 3 x float4, parks them in the (already consumed) rotation area at float4 3 * lane + r
//     (3 is odd: a b128 service group's 16 lanes land in 16 different columns) and the wave reads them back as float4 64 * j + lane: three
//     1 KiB stores per 64 bones. The absolute rotations leave from registers (1 KiB per instruction), the absolute positions as the
//     dwords the LDS already holds in memory order. All stores are non-temporal.
// Palette layout in HBM: 3 x float4 per bone (48 B) holding rows 0..2 of the matrix - what evaluateSkin reads; row 3 of
// (pose * inverse_bind).toMatrix() is the constant (0, 0, 0, 1) (math.cpp:887-890) and is re-attached on read-back. With row r =
// (c0[r], c1[r], c2[r], c3[r]):  P0 = (r0.x, r1.x, r0.y, r1.y)   P1 = (r0.z, r1.z, r0.w, r1.w)   P2 = r2.
// Rows 0 and 1 are interleaved by column so that the blended {row 0, row 1} components are register PAIRS as they come out of the
// packed blend, and the transform of a vertex by rows 0 and 1 is four packed operations on them (round 4: with plain rows the compiler
// packed the transform itself and paid eight v_mov per vertex to build the pairs - 61 VALU wave-instructions per 64 vertices on a
// kernel that is VALU- and power-bound, profiles/r04/final/target_counters_summary.json).
//
// Round 3's kernel (a 4-wave block per 16 instances, (instance, bone) lanes in every phase, a barrier per level) moved 16-byte pieces -
// a global instruction touched 64 different 128-byte lines - and took 172 us for 100 000 x 64 bones; this one 133 us, 105 us with
// non-temporal stores (6.3 TB/s on the 104 B / bone; 8 instances per wave: 112 us; profiles/r04/pose_probe.txt, tools/pose_probe.hip).
#ifndef LMX_POSE_NT
#define LMX_POSE_NT 1 // non-temporal stores for the palette and the absolute pose (100 k x 64 bones: 133 -> 105 us, profiles/r04/pose_probe.txt)
#endif
__device__ __forceinline__ void pose_store(float* p, float v) {
#if LMX_POSE_NT
	__builtin_nontemporal_store(v, p);
#else
	*p = v;
#endif
}
__device__ __forceinline__ void pose_store(float4* p, const float4& v) {
#if LMX_POSE_NT
	typedef float pose_v4f __attribute__((ext_vector_type(4)));
	const pose_v4f t = {v.x, v.y, v.z, v.w};
	__builtin_nontemporal_store(t, reinterpret_cast<pose_v4f*>(p));
#else
	*p = v;
#endif
}

template <int NBMAX, int KSHIFT> struct PoseWave {
	static constexpr uint32_t K = 1u << KSHIFT;
	static constexpr uint32_t CHUNKS = (NBMAX + 63) / 64;                            // 64-bone chunks per instance
	static constexpr uint32_t Q = K * CHUNKS;                                        // (instance, chunk) pairs of a group = bones per lane
	static constexpr uint32_t NBP = K == 1 ? NBMAX : ((NBMAX + 15) / 16) * 16 + 16 / K;       // float4 per instance: = 16 / K modulo 16
	static constexpr uint32_t PS = K == 1 ? 3 * NBMAX : ((3 * NBMAX + 63) / 64) * 64 + 64 / K; // dwords per instance: = 64 / K modulo 64
	static constexpr uint32_t PCH = (3 * NBMAX + 63) / 64;                           // dword loads per lane and instance
	static constexpr uint32_t ICH = (NBMAX + 63) / 64;                               // item words per lane
};

template <int NBMAX, int KSHIFT>
__global__ __launch_bounds__(64) void k_pose_palette(const SkinInstance* __restrict__ inst, const PoseGroup* __restrict__ groups,
	const float* rel_pos, const float4* rel_rot, float* pose_pos, float4* pose_rot /* rel_* may alias pose_*: no __restrict__; null = no write-back */,
	const uint32_t* __restrict__ level_items, const uint16_t* __restrict__ level_off, const float* __restrict__ inv_pos,
	const float4* __restrict__ inv_rot, float4* __restrict__ palette, float4* __restrict__ dual_quats) {
	using C = PoseWave<NBMAX, KSHIFT>;
	constexpr uint32_t K = C::K, NBP = C::NBP, PS = C::PS, PCH = C::PCH, Q = C::Q, CHUNKS = C::CHUNKS, ICH = C::ICH;
	static_assert(K * NBP >= 192, "the rotation area doubles as the staging rows of 64 bones' palette");
	__shared__ float4 s_rot[K * NBP];
	__shared__ float s_pos[K * PS];
	__shared__ uint32_t s_item[NBMAX];     // bone | parent << 16, sorted by depth (bones >= first_nonroot only)
	__shared__ uint16_t s_off[NBMAX + 1];  // s_off[d - 1] .. s_off[d]: items of depth d
	const uint32_t lane = threadIdx.x;
	const PoseGroup g = groups[blockIdx.x];
	const SkinInstance in = inst[g.first_inst]; // all instances of the group share the model; their bones are consecutive in memory
	const uint32_t nb = in.n_bones;
	const size_t bone0 = in.bone_offset;
	// ---- loads: the model's level tables first (loads return in order: the walk's tables are not queued behind the poses), then
	// everything the group reads, all issued before anything is used
	uint32_t item_w[ICH], off_w[ICH + 1];
#pragma unroll
	for (uint32_t c = 0; c < ICH; ++c) { // (n_items <= n_bones - 1 < 64 * ICH; an index past the end re-reads the model's neighbours, never out of the table: + 0)
		const uint32_t i = c * 64 + lane;
		item_w[c] = level_items[in.lv_items_offset + (i < nb ? i : 0u)];
	}
#pragma unroll
	for (uint32_t c = 0; c <= ICH; ++c) {
		const uint32_t i = c * 64 + lane;
		off_w[c] = level_off[in.lv_off_offset + (i <= in.max_depth ? i : 0u)];
	}
	float4 rv[Q];
	float pv[K][PCH];
#pragma unroll
	for (uint32_t q = 0; q < Q; ++q) {
		const uint32_t k = q / CHUNKS, b = (q % CHUNKS) * 64 + lane;
		const uint32_t kk = k < g.count && !LMX_PROBE_SKIP(4) ? k : 0u; // instances past the group's end re-read its first one (no branch around the loads)
		rv[q] = rel_rot[bone0 + (size_t)kk * nb + (b < nb ? b : nb - 1)];
	}
#pragma unroll
	for (uint32_t k = 0; k < K; ++k) {
		const uint32_t kk = k < g.count && !LMX_PROBE_SKIP(4) ? k : 0u;
		const float* src = rel_pos + (bone0 + (size_t)kk * nb) * 3;
#pragma unroll
		for (uint32_t c = 0; c < PCH; ++c) {
			const uint32_t j = c * 64 + lane;
			pv[k][c] = src[j < 3 * nb ? j : 3 * nb - 1];
		}
	}
#pragma unroll
	for (uint32_t c = 0; c < ICH; ++c)
		if (c * 64 + lane < (uint32_t)NBMAX) s_item[c * 64 + lane] = item_w[c];
#pragma unroll
	for (uint32_t c = 0; c <= ICH; ++c)
		if (c * 64 + lane <= (uint32_t)NBMAX) s_off[c * 64 + lane] = (uint16_t)off_w[c];
#pragma unroll
	for (uint32_t q = 0; q < Q; ++q) {
		const uint32_t k = q / CHUNKS, b = (q % CHUNKS) * 64 + lane;
		if (b < nb) s_rot[k * NBP + b] = rv[q];
	}
#pragma unroll
	for (uint32_t k = 0; k < K; ++k) {
#pragma unroll
		for (uint32_t c = 0; c < PCH; ++c) {
			const uint32_t j = c * 64 + lane;
			if (j < 3 * nb) s_pos[k * PS + j] = pv[k][c];
		}
	}
	wave_lds_sync();
	// ---- Pose::computeAbsolute, level by level; lanes = (instance, bone of the level). The next level's bounds and this lane's first
	// item of it are read while the current level computes (three dependent LDS round trips per level otherwise)
	{
		const uint32_t k = lane & (K - 1), j0 = lane >> KSHIFT;
		const bool k_live = k < g.count;
		const uint32_t md = LMX_PROBE_SKIP(1) ? 0u : in.max_depth;
		uint32_t start = s_off[0], end = md ? s_off[1] : start;
		uint32_t it0 = s_item[start + j0 < (uint32_t)NBMAX ? start + j0 : 0u];
		for (uint32_t d = 1; d <= md; ++d) {
			const uint32_t n_start = end, n_end = s_off[d < md ? d + 1 : d];
			const uint32_t n_it0 = s_item[n_start + j0 < (uint32_t)NBMAX ? n_start + j0 : 0u];
			for (uint32_t j = j0; start + j < end; j += 64 / K) {
				if (k_live) {
					const uint32_t it = j == j0 ? it0 : s_item[start + j];
					const uint32_t bi = it & 0xffffu, pi = it >> 16;
					const float4 pr4 = s_rot[k * NBP + pi];
					const float4 r4 = s_rot[k * NBP + bi];
					const float* pp = s_pos + k * PS + 3 * pi;
					float* bp = s_pos + k * PS + 3 * bi;
					const Q4 pr = Q4{pr4.x, pr4.y, pr4.z, pr4.w};
					const V3 np = add(rotate(pr, V3{bp[0], bp[1], bp[2]}), V3{pp[0], pp[1], pp[2]});
					const Q4 nr = qmul(pr, Q4{r4.x, r4.y, r4.z, r4.w});
					bp[0] = np.x; bp[1] = np.y; bp[2] = np.z;
					s_rot[k * NBP + bi] = make_float4(nr.x, nr.y, nr.z, nr.w);
				}
			}
			wave_lds_sync();
			start = n_start; end = n_end; it0 = n_it0;
		}
	}
	// ---- absolute poses back into registers (lane = bone), absolute positions out as the dwords the LDS holds in memory order
	V3 ap[Q];
#pragma unroll
	for (uint32_t q = 0; q < Q; ++q) {
		const uint32_t k = q / CHUNKS, b = (q % CHUNKS) * 64 + lane;
		const uint32_t bb = b < nb ? b : nb - 1;
		rv[q] = s_rot[k * NBP + bb];
		const float* sp = s_pos + k * PS + 3 * bb;
		ap[q] = V3{sp[0], sp[1], sp[2]};
	}
	if (pose_pos != nullptr && !LMX_PROBE_SKIP(2)) { // the pose becomes absolute (Pose::is_absolute = true, pose.cpp:133)
#pragma unroll
		for (uint32_t k = 0; k < K; ++k) {
			float* dst = pose_pos + (bone0 + (size_t)k * nb) * 3;
#pragma unroll
			for (uint32_t c = 0; c < PCH; ++c) {
				const uint32_t j = c * 64 + lane;
				if (k < g.count && j < 3 * nb) pose_store(dst + j, s_pos[k * PS + j]);
			}
		}
#pragma unroll
		for (uint32_t q = 0; q < Q; ++q) {
			const uint32_t k = q / CHUNKS, b = (q % CHUNKS) * 64 + lane;
			if (k < g.count && b < nb) pose_store(pose_rot + bone0 + (size_t)k * nb + b, rv[q]);
		}
	}
	wave_lds_sync(); // every lane holds its bones: the rotation area is free for the staging rows
	// ---- palette (computeSkinMatrices), optional dual quaternions
	const float* ipos = inv_pos + (size_t)in.model_offset * 3;
	const float4* irot = inv_rot + in.model_offset;
#pragma unroll
	for (uint32_t q = 0; q < Q; ++q) {
		const uint32_t k = q / CHUNKS, cb = (q % CHUNKS) * 64, b = cb + lane;
		if (k >= g.count || cb >= nb) continue; // wave-uniform
		const uint32_t bb = b < nb ? b : nb - 1;
		const float4 ir4 = irot[bb];
		const Q4 ir = Q4{ir4.x, ir4.y, ir4.z, ir4.w};
		const V3 ip = V3{ipos[3 * bb], ipos[3 * bb + 1], ipos[3 * bb + 2]};
		const Q4 r = Q4{rv[q].x, rv[q].y, rv[q].z, rv[q].w};
		const Mat4 m = skin_matrix(ap[q], r, ip, ir);
		if (LMX_PROBE_SKIP(2) && m.c[0][0] != 123.f) continue;
		const size_t i0 = bone0 + (size_t)k * nb + cb; // first bone of the chunk
		if (dual_quats != nullptr && b < nb) { // the palette format of the reference's own GPU skinning path (32 B per bone)
			const DualQ dq = skin_dual_quat(ap[q], r, ip, ir);
			float4* o = dual_quats + (i0 + lane) * 2;
			o[0] = make_float4(dq.r.x, dq.r.y, dq.r.z, dq.r.w);
			o[1] = make_float4(dq.d.x, dq.d.y, dq.d.z, dq.d.w);
		}
		// the three float4 of a bone (see "Palette layout" above): rows 0 and 1 interleaved by column, then row 2
		s_rot[3 * lane] = make_float4(m.c[0][0], m.c[0][1], m.c[1][0], m.c[1][1]);
		s_rot[3 * lane + 1] = make_float4(m.c[2][0], m.c[2][1], m.c[3][0], m.c[3][1]);
		s_rot[3 * lane + 2] = make_float4(m.c[0][2], m.c[1][2], m.c[2][2], m.c[3][2]);
		wave_lds_sync();
		const uint32_t n_rows = 3 * (nb - cb < 64 ? nb - cb : 64);
		float4* out = palette + i0 * 3;
		const float4 o0 = s_rot[lane], o1 = s_rot[64 + lane], o2 = s_rot[128 + lane];
		if (lane < n_rows) pose_store(out + lane, o0);
		if (64 + lane < n_rows) pose_store(out + 64 + lane, o1);
		if (128 + lane < n_rows) pose_store(out + 128 + lane, o2);
		wave_lds_sync(); // the rows are in registers before the next chunk overwrites them
	}
}

// Pose::blend (renderer/pose.cpp:30-41) for every bone of every instance: positions = positions * inv + rhs * weight, rotations =
// nlerp(rotations, rhs, weight) (core/math.cpp:677-691: left-to-right dot and length, t negated for the short way round)
__global__ __launch_bounds__(256) void k_pose_blend(float* __restrict__ pos, float4* __restrict__ rot, const float* __restrict__ rhs_pos,
	const float4* __restrict__ rhs_rot, size_t n_bones, float weight) {
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n_bones) return;
	const float inv = 1.0f - weight;
	pos[3 * i] = pos[3 * i] * inv + rhs_pos[3 * i] * weight;
	pos[3 * i + 1] = pos[3 * i + 1] * inv + rhs_pos[3 * i + 1] * weight;
	pos[3 * i + 2] = pos[3 * i + 2] * inv + rhs_pos[3 * i + 2] * weight;
	const float4 q1 = rot[i], q2 = rhs_rot[i];
	float t = weight;
	if (q1.x * q2.x + q1.y * q2.y + q1.z * q2.z + q1.w * q2.w < 0) t = -t;
	float4 r = make_float4(q1.x * inv + q2.x * t, q1.y * inv + q2.y * t, q1.z * inv + q2.z * t, q1.w * inv + q2.w * t);
	const float l = 1 / sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
	rot[i] = make_float4(r.x * l, r.y * l, r.z * l, r.w * l);
}

// lmx_skin_read_palette: 3 x float4 rows -> the reference's column-major Matrix (row 3 = (0, 0, 0, 1), math.cpp:887-890)
__global__ __launch_bounds__(256) void k_palette_expand(const float4* __restrict__ rows, uint32_t n_bones, float4* __restrict__ out) {
	const uint32_t b = blockIdx.x * 256 + threadIdx.x;
	if (b >= n_bones) return;
	const float4 p0 = rows[3 * b], p1 = rows[3 * b + 1], r2 = rows[3 * b + 2]; // P0 = (r0.x, r1.x, r0.y, r1.y), P1 = (r0.z, r1.z, r0.w, r1.w), P2 = r2
	out[4 * b] = make_float4(p0.x, p0.y, r2.x, 0.f);
	out[4 * b + 1] = make_float4(p0.z, p0.w, r2.y, 0.f);
	out[4 * b + 2] = make_float4(p1.x, p1.y, r2.z, 0.f);
	out[4 * b + 3] = make_float4(p1.z, p1.w, r2.w, 1.f);
}

// ---- linear-blend skinning ----------------------------------------------------------------------------------
// A block skins vertices with one instance's palette staged in LDS as 3 rows x float4 per bone. The bone indices of
// neighbouring vertices are unrelated in the worst case, so the 12 ds_read_b128 per vertex would collide on LDS banks (16
// random 16-B slots per service group -> ~3x serialisation). The palette is therefore REPLICATED: copy c of every row lives
// in 16-B bank column c (slot = (bone*3 + row) * COPIES + c) and lane l reads copy l % COPIES. With 16 copies every lane of a
// ds_read_b128 service group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: 16 lanes with distinct l % 16,
// MI355X_MICROARCH.md LDS table) owns its own bank column -> conflict-free whatever the indices. 48 KiB per palette
// (64 bones x 16 copies, 128 x 8, 196 x 4).
//
// Two kernels share the blend:
//   k_skin_vertices  one block = one tile of ONE instance's vertices; vertex records stream from memory. General path.
//   k_skin_shared    one block = one tile of a mesh x a run of instances that share it: the vertex records are loaded ONCE
//                    into registers and re-used for every instance of the run, palettes are double-buffered in LDS. A CU's
//                    vector-memory path is in order, so in k_skin_vertices the (L2-resident) mesh loads queue behind the
//                    stores waiting on HBM and the kernel runs at blend time + store time (tools/skin_probe.hip); here
//                    the steady state issues stores only.
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int SKIN_THREADS = 512;
constexpr int SKIN_WAVES_PER_SIMD = 6; // 3 blocks of 8 waves per CU: <= 80 VGPRs, 48 KiB LDS each
constexpr int SKIN_LDS_SLOTS = 3072;   // float4 slots = 48 KiB
#ifndef LMX_SKIN_PIPE
#define LMX_SKIN_PIPE 2
#endif
constexpr int SKIN_PIPE = LMX_SKIN_PIPE; // k_skin_vertices: vertex records in flight per lane (see skin_tile)

constexpr int skin_rows(int mode) { return mode == LMX_SKIN_DQS ? 2 : 3; } // LDS rows per bone: {real, dual} or the 3 matrix rows

struct F3 { float x, y, z; }; // 12-byte records: loaded / stored as one dwordx3 per lane
struct VertexIn { float px, py, pz; float4 w; int2 iw; };
typedef float v4f __attribute__((ext_vector_type(4)));
// A vertex record as the mesh table holds it and as the loads define it: two 16-byte register tuples.
//   a = the 4 bone weights;  b = (position.xyz, the 4 bone indices as 4 x u8 - Model::Bone::MAX_COUNT is 196, model.h:155)
// Built by lmx_skin_add_mesh from the engine's arrays: 32 bytes per vertex in two dwordx4 loads, where positions (12 B) + weights
// (16 B) + i16 indices (8 B) in three arrays cost 36 bytes and three loads.
struct RawVertex { v4f a, b; };
__device__ __forceinline__ RawVertex load_vertex(const float4* __restrict__ mesh, uint32_t v) {
	RawVertex r;
	__builtin_memcpy(&r.a, mesh + 2 * (size_t)v, sizeof(float4));
	__builtin_memcpy(&r.b, mesh + 2 * (size_t)v + 1, sizeof(float4));
	return r;
}

// evaluateSkin of one vertex (model.cpp:103-109) against the palette rows of its 4 bones in LDS (r0..r3 point at row 0 of
// the lane's copy; rows are COPIES slots apart)
template <int COPIES, int MODE>
__device__ __forceinline__ F3 skin_blend_rows(const float4* r0, const float4* r1, const float4* r2, const float4* r3, const VertexIn& c) {
	const float4 w = c.w;
	if constexpr (MODE == LMX_SKIN_DQS) {
		// the SKINNED branch of the reference's vertex shader (data/shaders/surface_base.hlsli:196-217): dual quaternions of the 4
		// bones blended with the weights' signs flipped towards bone 0's hemisphere, normalised by the real part's length, then
		// transformByDualQuat (data/shaders/common.hlsli:632-636). Row 0 = real part, row 1 = dual part (DualQuat, math.h:257-260).
		// bone by bone (two rows in flight at a time): mul(getBones(i), w) accumulated left to right as the shader does
		const float4 ra = r0[0];
		float4 qr, qd;
		{
			const float4 da = r0[COPIES];
			qr = make_float4(ra.x * w.x, ra.y * w.x, ra.z * w.x, ra.w * w.x);
			qd = make_float4(da.x * w.x, da.y * w.x, da.z * w.x, da.w * w.x);
		}
		auto accumulate = [&](const float4* rp, float weight) {
			const float4 rb = rp[0], db = rp[COPIES];
			const float ws = (rb.x * ra.x + rb.y * ra.y + rb.z * ra.z + rb.w * ra.w) < 0 ? -weight : weight;
			qr = make_float4(qr.x + rb.x * ws, qr.y + rb.y * ws, qr.z + rb.z * ws, qr.w + rb.w * ws);
			qd = make_float4(qd.x + db.x * ws, qd.y + db.y * ws, qd.z + db.z * ws, qd.w + db.w * ws);
			__builtin_amdgcn_sched_barrier(0);
		};
		accumulate(r1, w.y);
		accumulate(r2, w.z);
		accumulate(r3, w.w);
		const float inv_len = 1 / sqrtf(qr.x * qr.x + qr.y * qr.y + qr.z * qr.z + qr.w * qr.w); // dq *= 1 / length(dq[0])
		qr = make_float4(qr.x * inv_len, qr.y * inv_len, qr.z * inv_len, qr.w * inv_len);
		qd = make_float4(qd.x * inv_len, qd.y * inv_len, qd.z * inv_len, qd.w * inv_len);
		// pos + 2 * cross(r.xyz, cross(r.xyz, pos) + r.w * pos) + 2 * (r.w * d.xyz - d.w * r.xyz + cross(r.xyz, d.xyz))
		const float ix = (qr.y * c.pz - qr.z * c.py) + qr.w * c.px, iy = (qr.z * c.px - qr.x * c.pz) + qr.w * c.py, iz = (qr.x * c.py - qr.y * c.px) + qr.w * c.pz;
		const float ox = qr.y * iz - qr.z * iy, oy = qr.z * ix - qr.x * iz, oz = qr.x * iy - qr.y * ix;
		const float tx = (qr.w * qd.x - qd.w * qr.x) + (qr.y * qd.z - qr.z * qd.y), ty = (qr.w * qd.y - qd.w * qr.y) + (qr.z * qd.x - qr.x * qd.z),
			tz = (qr.w * qd.z - qd.w * qr.z) + (qr.x * qd.y - qr.y * qd.x);
		return F3{(c.px + 2 * ox) + 2 * tx, (c.py + 2 * oy) + 2 * ty, (c.pz + 2 * oz) + 2 * tz};
	}
	// the three float4 of the four bones (P0, P1, P2 of "Palette layout"), each consumed as soon as it is blended (12 blended floats alive at
	// once cost the kernel its register budget): P2 -> the z row; P0 -> {row 0, row 1} columns x, y -> the packed transform's first two
	// terms; P1 -> columns z, w -> its last two
	const v2f w01 = {w.x, w.y}, w23 = {w.z, w.w};
	// (the weight splats are shuffles of the loaded pairs, so that the broadcast folds into op_sel of v_pk_*_f32)
	const v2f wx = __builtin_shufflevector(w01, w01, 0, 0), wy = __builtin_shufflevector(w01, w01, 1, 1);
	const v2f wz = __builtin_shufflevector(w23, w23, 0, 0), ww = __builtin_shufflevector(w23, w23, 1, 1);
	auto blend = [&](int slot, v2f& lo, v2f& hi) {
		float4 A, B, C, D;
		if (LMX_PROBE_SKIP(16)) { A = w; B = c.w; C = make_float4(c.px, c.py, c.pz, w.x); D = make_float4(w.y, c.px, w.z, c.py); }
		else { A = r0[slot * COPIES]; B = r1[slot * COPIES]; C = r2[slot * COPIES]; D = r3[slot * COPIES]; }
		// (the four floats of an LDS slot land in consecutive VGPRs: {x, y} and {z, w} are packed operands as they are)
		const v2f a01 = {A.x, A.y}, a23 = {A.z, A.w}, b01 = {B.x, B.y}, b23 = {B.z, B.w};
		const v2f c01 = {C.x, C.y}, c23 = {C.z, C.w}, d01 = {D.x, D.y}, d23 = {D.z, D.w};
		if constexpr (MODE == LMX_SKIN_EXACT) {
			// Matrix::operator*(float) and operator+ (math.cpp:1022-1071), left to right per component: ((A*w.x + B*w.y) + C*w.z) + D*w.w,
			// every product and sum rounded on its own (-ffp-contract=off: v_pk_mul_f32 / v_pk_add_f32)
			lo = a01 * wx; lo = lo + b01 * wy; lo = lo + c01 * wz; lo = lo + d01 * ww;
			hi = a23 * wx; hi = hi + b23 * wy; hi = hi + c23 * wz; hi = hi + d23 * ww;
		} else {
			// the same association with the products fused into the adds (v_pk_fma_f32)
			lo = a01 * wx; hi = a23 * wx;
			lo = __builtin_elementwise_fma(b01, wy, lo); hi = __builtin_elementwise_fma(b23, wy, hi);
			lo = __builtin_elementwise_fma(c01, wz, lo); hi = __builtin_elementwise_fma(c23, wz, hi);
			lo = __builtin_elementwise_fma(d01, ww, lo); hi = __builtin_elementwise_fma(d23, ww, hi);
		}
	};
	const v2f px2 = {c.px, c.px}, py2 = {c.py, c.py}, pz2 = {c.pz, c.pz};
	v2f lo, hi, o01;
	float o2;
	blend(2, lo, hi); // row 2: (x, y), (z, w)
	if constexpr (MODE == LMX_SKIN_EXACT) o2 = lo.x * c.px + lo.y * c.py + hi.x * c.pz + hi.y; // Matrix::transformPoint (math.cpp:1231-1235), left to right
	else o2 = fmaf(hi.x, c.pz, fmaf(lo.y, c.py, lo.x * c.px)) + hi.y;
	asm volatile("" : "+v"(o2)); // (row 2 is done before the next slot's reads are issued: their 16 registers are the ones just freed)
	blend(0, lo, hi); // {row 0, row 1}: column x, column y
	if constexpr (MODE == LMX_SKIN_EXACT) { o01 = lo * px2; o01 = o01 + hi * py2; }
	else o01 = __builtin_elementwise_fma(hi, py2, lo * px2);
#ifndef LMX_HOSTSIM
	asm volatile("" : "+v"(o01)); // (a scheduling fence only; the simulated device's compiler has no register class for an 8-byte vector)
#endif
	blend(1, lo, hi); // {row 0, row 1}: column z, column w
	if constexpr (MODE == LMX_SKIN_EXACT) { o01 = o01 + lo * pz2; o01 = o01 + hi; }
	else o01 = __builtin_elementwise_fma(lo, pz2, o01) + hi;
	return F3{o01.x, o01.y, o2};
}

// LMX_SKIN_FUSED over the first N of the vertex's four bone slots (the others carry weight 0 for EVERY lane of the wave): the same
// operations as skin_blend_rows minus the fused multiply-adds whose product is an exact zero - bit-identical for finite palettes -
// and minus their 3 x ds_read_b128 per slot. A rigged character binds most vertices to one or two bones (the reference's demo
// character: 1.17 influences per control point), and consecutive vertices follow the same limb: whole waves take N = 1.
template <int COPIES, int N>
__device__ __forceinline__ F3 skin_blend_fused_n(const float4* rows, const RawVertex& rec) {
	const uint32_t idx = __float_as_uint(rec.b.w);
	constexpr uint32_t STRIDE = 3 * COPIES;
	const float4* r0 = rows + (idx & 0xffu) * STRIDE;
	const float4* r1 = rows + ((idx >> 8) & 0xffu) * STRIDE;
	const float4* r2 = rows + ((idx >> 16) & 0xffu) * STRIDE;
	const float4* r3 = rows + (idx >> 24) * STRIDE;
	const v2f w01 = {rec.a.x, rec.a.y}, w23 = {rec.a.z, rec.a.w};
	const v2f wx = __builtin_shufflevector(w01, w01, 0, 0), wy = __builtin_shufflevector(w01, w01, 1, 1);
	const v2f wz = __builtin_shufflevector(w23, w23, 0, 0), ww = __builtin_shufflevector(w23, w23, 1, 1);
	v2f q[4], m2[2];
#pragma unroll
	for (int slot = 0; slot < 3; ++slot) {
		const float4 A = r0[slot * COPIES];
		v2f lo = v2f{A.x, A.y} * wx, hi = v2f{A.z, A.w} * wx;
		if constexpr (N >= 2) {
			const float4 B = r1[slot * COPIES];
			lo = __builtin_elementwise_fma(v2f{B.x, B.y}, wy, lo);
			hi = __builtin_elementwise_fma(v2f{B.z, B.w}, wy, hi);
		}
		if constexpr (N >= 3) {
			const float4 C = r2[slot * COPIES];
			lo = __builtin_elementwise_fma(v2f{C.x, C.y}, wz, lo);
			hi = __builtin_elementwise_fma(v2f{C.z, C.w}, wz, hi);
		}
		if constexpr (N >= 4) {
			const float4 D = r3[slot * COPIES];
			lo = __builtin_elementwise_fma(v2f{D.x, D.y}, ww, lo);
			hi = __builtin_elementwise_fma(v2f{D.z, D.w}, ww, hi);
		}
		if (slot == 0) { q[0] = lo; q[1] = hi; } else if (slot == 1) { q[2] = lo; q[3] = hi; } else { m2[0] = lo; m2[1] = hi; }
	}
	const v2f px2 = {rec.b.x, rec.b.x}, py2 = {rec.b.y, rec.b.y}, pz2 = {rec.b.z, rec.b.z};
	const v2f o01 = __builtin_elementwise_fma(q[2], pz2, __builtin_elementwise_fma(q[1], py2, q[0] * px2)) + q[3];
	const float o2 = fmaf(m2[1].x, rec.b.z, fmaf(m2[0].y, rec.b.y, m2[0].x * rec.b.x)) + m2[1].y;
	return F3{o01.x, o01.y, o2};
}

// `rows` already points at the lane's copy
template <int COPIES, int MODE>
__device__ __forceinline__ F3 skin_blend(const float4* rows, const RawVertex& r) {
	VertexIn c;
	c.px = r.b.x; c.py = r.b.y; c.pz = r.b.z;
	c.w = make_float4(r.a.x, r.a.y, r.a.z, r.a.w);
	const uint32_t idx = __float_as_uint(r.b.w);
	constexpr uint32_t STRIDE = skin_rows(MODE) * COPIES;
	return skin_blend_rows<COPIES, MODE>(rows + (idx & 0xffu) * STRIDE, rows + ((idx >> 8) & 0xffu) * STRIDE, rows + ((idx >> 16) & 0xffu) * STRIDE,
		rows + (idx >> 24) * STRIDE, c);
}

// Palette staging: COPIES / 4 lanes share a row; each fetches the row's float4 (one 16-byte load, the lanes of a row coalesce)
// and writes FOUR of its COPIES slots. Work item `w` = (row f = w / LPR, quarter q = w % LPR), n_rows * LPR items per palette:
// 768 for 64 bones x 16 copies or 128 bones x 8, 588 for 196 bones x 4 - every wave of the block takes part. (One lane per row
// writing all 16 copies put the whole 48 KiB - ~620 cycles of the ds_write_b128 path - on three waves at the end of every
// instance of k_skin_shared while the other thirteen sat at the barrier: 17 % of that kernel, tools/skin_probe.hip mask 32.)
// The copy written in step i is rotated by the row so that the 8 lanes of a ds_write_b128 service group (consecutive lanes =
// 8 / LPR rows x LPR quarters) land on 8 different 16-byte bank columns.
template <int COPIES>
__device__ __forceinline__ float4 palette_fetch(const float4* __restrict__ pal, uint32_t n_rows, uint32_t w) {
	constexpr uint32_t LPR = COPIES / 4;
	const uint32_t f = w / LPR;
	return pal[f < n_rows ? f : 0u];
}
template <int COPIES>
__device__ __forceinline__ void palette_spread(float4* s_rows, uint32_t n_rows, uint32_t w, float4 t) {
	constexpr uint32_t LPR = COPIES / 4;
	const uint32_t f = w / LPR, q = w % LPR;
	if (f < n_rows) {
		const uint32_t rot = COPIES == 4 ? (f >> 1) : LPR * f;
#pragma unroll
		for (uint32_t i = 0; i < 4; ++i) s_rows[f * COPIES + ((q + LPR * i + rot) & (COPIES - 1))] = t;
	}
}
constexpr uint32_t palette_items(uint32_t max_rows, uint32_t copies) { return max_rows * (copies / 4); }

// experiment knobs (tools/skin_probe.hip builds the variants; the defaults are what measured best)
#ifndef LMX_SHARED_NT
#define LMX_SHARED_NT 1    // non-temporal output stores (the 12 GB of positions are never read back by this kernel)
#endif

typedef float v3f_a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));

// one vertex's result to out[v] (12 bytes, lanes contiguous: 768 bytes per wave-instruction)
__device__ __forceinline__ void store_position(F3* dst, const F3& r) {
#if LMX_SHARED_NT
	v3f_a4 t = {r.x, r.y, r.z};
	__builtin_nontemporal_store(t, reinterpret_cast<v3f_a4*>(dst));
#else
	*dst = r;
#endif
}

template <int COPIES, int MODE>
__device__ __forceinline__ void skin_tile(const SkinInstance& in, uint32_t v_begin, uint32_t v_end, float4* s_rows,
	const float4* __restrict__ mesh, const float4* __restrict__ palette, float* __restrict__ out) {
	const uint32_t col = threadIdx.x & (COPIES - 1);
	const float4* rows = s_rows + col;
	// per-instance base pointers are wave-uniform (SGPRs); per-lane offsets stay 32-bit
	const float4* mbase = mesh + 2 * (size_t)in.vert_offset;
	F3* obase = reinterpret_cast<F3*>(out) + in.out_offset;
	// records travel through the pipeline as the register TUPLES the loads define: carried as scalars, the compiler copied every
	// freshly loaded tuple into the loop-carried scalars right behind the load - and waited for it there
	auto load = [&](uint32_t v) { return load_vertex(mbase, v); };
	// Software pipeline, SKIN_PIPE vertex records deep, WITHOUT a branch in the steady state: loads and stores retire through one
	// in-order counter (vmcnt), so waiting for the record loaded SKIN_PIPE steps ago also waits for every store older than it,
	// and the compiler only emits the exact `s_waitcnt vmcnt(N)` when it can count the operations in between. With the loads /
	// stores under `if (v < v_end)` it emitted vmcnt(0) at the loop header: each wave drained its previous store to HBM before
	// blending the next vertex, and the kernel ran at blend time + store time. Now lanes and steps past the tile's end are CLAMPED
	// to its last vertex (they recompute and rewrite the same 12 bytes), every load / store is unconditional, a wave keeps
	// SKIN_PIPE - 1 stores in flight, and the wait covers only stores at least that many steps old.
	const uint32_t v_last = v_end - 1; // v_begin < v_end (checked by the kernel)
	const uint32_t v0 = v_begin + threadIdx.x;
	RawVertex rec[SKIN_PIPE];
#pragma unroll
	for (int d = 0; d < SKIN_PIPE; ++d) rec[d] = load(min(v0 + d * SKIN_THREADS, v_last)); // in flight during the palette staging
	{
		const float4* pal = palette + (size_t)in.bone_offset * skin_rows(MODE);
		const uint32_t n_rows = in.n_bones * skin_rows(MODE);
		// <= 768 work items (64 x 3 rows x 4 lanes, 128 x 3 x 2, 196 x 3 x 1) on 512 lanes: two passes, both loads in flight
		const float4 t0 = palette_fetch<COPIES>(pal, n_rows, threadIdx.x);
		const float4 t1 = palette_fetch<COPIES>(pal, n_rows, threadIdx.x + SKIN_THREADS);
		palette_spread<COPIES>(s_rows, n_rows, threadIdx.x, t0);
		palette_spread<COPIES>(s_rows, n_rows, threadIdx.x + SKIN_THREADS, t1);
	}
	__syncthreads();
	const uint32_t n_steps = (v_end - v_begin + SKIN_THREADS - 1) / SKIN_THREADS; // block-uniform
	for (uint32_t it = 0; it < n_steps; it += SKIN_PIPE) {
#pragma unroll
		for (int d = 0; d < SKIN_PIPE; ++d) {
			const uint32_t v = min(v0 + (it + d) * SKIN_THREADS, v_last);
			// opaque to the optimiser: whatever it derives from the record (packed operand pairs, LDS addresses) is formed HERE, at the
			// use, not right behind the loads two steps earlier, where it would have to wait for them
			asm volatile("" : "+v"(rec[d].a), "+v"(rec[d].b));
			F3 o = skin_blend<COPIES, MODE>(rows, rec[d]);
			// the blend is COMPLETE here (pure arithmetic is otherwise free to sink below the refill, towards the store, which keeps the
			// old record alive and forces the refill into fresh registers + copies at the loop latch, waited for with vmcnt)
			asm volatile("" : "+v"(o.x), "+v"(o.y), "+v"(o.z));
			__builtin_amdgcn_sched_barrier(0); // the refill goes into the registers the blend has just finished with: no copies
			rec[d] = load(min(v0 + (it + d + SKIN_PIPE) * SKIN_THREADS, v_last));
			if (!LMX_PROBE_SKIP(8) || o.x == 123.25f) store_position(obase + v, o);
			__builtin_amdgcn_sched_barrier(0);
		}
	}
}

template <int MODE>
__global__ __launch_bounds__(SKIN_THREADS, SKIN_WAVES_PER_SIMD) void k_skin_vertices(const SkinInstance* __restrict__ inst,
	const uint32_t* __restrict__ inst_index /* optional: the instances this launch covers */, uint32_t tiles_per_inst, uint32_t tile_verts,
	const float4* __restrict__ mesh, const float4* __restrict__ palette, float* __restrict__ out) {
	__shared__ float4 s_rows[SKIN_LDS_SLOTS];
	uint32_t ii = blockIdx.x / tiles_per_inst;
	const uint32_t tile = blockIdx.x - ii * tiles_per_inst;
	if (inst_index != nullptr) ii = inst_index[ii];
	const SkinInstance in = inst[ii];
	const uint32_t v_begin = tile * tile_verts;
	if (v_begin >= in.n_verts) return; // block-uniform
	const uint32_t v_end = min(v_begin + tile_verts, in.n_verts);
	if (in.n_bones <= 64) skin_tile<16, MODE>(in, v_begin, v_end, s_rows, mesh, palette, out);
	else if (in.n_bones <= 128) skin_tile<8, MODE>(in, v_begin, v_end, s_rows, mesh, palette, out);
	else skin_tile<4, MODE>(in, v_begin, v_end, s_rows, mesh, palette, out);
}

// ---- shared-mesh runs: vertex records in registers, palettes double-buffered in LDS ---------------------------------------------
constexpr int SHARED_THREADS = 1024; // 16 waves = 4 per SIMD at <= 128 VGPRs; one block per CU (2 x 48 KiB LDS)
constexpr int SHARED_VPT = 5;        // vertex records per lane (8 VGPRs each) -> tiles of up to 5120 vertices
#ifndef LMX_SHARED_SPREAD_AFTER
#define LMX_SHARED_SPREAD_AFTER 3
#endif
constexpr int SHARED_SPREAD_AFTER = LMX_SHARED_SPREAD_AFTER; // the next palette is spread into LDS after this vertex of the lane's five
static_assert(SHARED_THREADS * SHARED_VPT == SKIN_SHARED_TILE_VERTS, "host tiling and kernel disagree");

// One palette staging instruction: 64 lanes x 16 bytes from per-lane global addresses (wave-uniform base + 32-bit lane offset)
// straight into 1 KiB of LDS at a wave-uniform address (LDS-DMA: no VGPR round trip). Inline asm because (a) M0 carries the LDS
// address and is compiler-reserved, (b) the compiler must NOT count this operation: it would drain vmcnt to 0 at the next barrier
// (cdna_hip_programming.md, "Pipelining across barriers"), i.e. every store of the wave once per instance; the issuing wave waits
// for it itself with an exact count.
__device__ __forceinline__ void lds_dma_16(uint32_t lane_byte_offset, const void* uniform_base, uint32_t lds_byte_address) {
#ifdef LMX_HOSTSIM // tests/hostsim executes this source on the CPU: the instruction's effect, lane by lane
	memcpy(static_cast<char*>(hostsim::lds_pointer(lds_byte_address)) + 16u * hostsim::lane(), static_cast<const char*>(uniform_base) + lane_byte_offset, 16);
#else
	uint32_t keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
				 : "=&s"(keep)
				 : "v"(lane_byte_offset), "s"(uniform_base), "s"(lds_byte_address)
				 : "memory");
#endif
}

#ifndef LMX_SHARED_DMA_NUM
#define LMX_SHARED_DMA_NUM 1 // share of the palette rows staged by LDS-DMA = NUM / DEN, the rest goes VGPR -> ds_write
#endif
#ifndef LMX_SHARED_DMA_DEN
#define LMX_SHARED_DMA_DEN 2
#endif

// A chunk = one tile of <= 5120 vertices of a mesh x a run of consecutive instances that share the mesh. The tile's vertex records
// carry TILE-LOCAL bone indices: only the palette rows of the bones the tile references (SkinChunk::n_tile_bones of them, listed in
// tile_bones) are staged, in the order of that list. On the reference's demo character a tile touches 13-16 of 52 bones
// (tools/fbx_skin_stats.cpp); the synthetic worst-case mesh (4 random bones of 64 per vertex) touches all of them.
//
// Staging of the replicated palette (48 KiB per instance for 64 bones) is SPLIT between the two paths a CU has into LDS, because
// each alone became the bound (profiles/r03/skin_ab_*.txt; ms per 1e9 vertices, worst-case mesh, round 2's kernel 3.14-3.25):
//   * LDS-DMA (global_load_lds_dwordx4, replication on the source side: the COPIES lanes of a row fetch the same 16 bytes, no VGPR
//     round trip, no ds_write): alone 2.83 WITH OR WITHOUT the stores - the DMA path moves ~12 B per cycle and CU whatever the source;
//   * VGPR -> 4 x ds_write_b128 per staging lane (round 2's way): alone 3.02 - the ds_write wave-instructions cost 13 cycles each
//     on the VGPR -> LDS path, and the row's global load puts a vmcnt wait behind four of every wave's stores.
// Rows [0, r_dma) go by DMA, issued at the top of the instance BEFORE the row load of the VGPR path: the compiler's own counted
// wait for that load (`vmcnt(4)`, the loop's only one) then covers the older DMAs too - in-order counter - without knowing them.
// Measured and NOT kept: the unique rows by DMA into a ring + LDS -> LDS replication (3.05: the ds_writes stay); three 16-byte
// stores per four lanes instead of four 12-byte ones (+0.3); skipping the LDS reads of zero-weight slots (LMX_SKIN_FUSED; -6 % on a
// character-like mesh, +6 % on the worst case, and a second copy of the loop spilled). Non-temporal stores: -0.15 ... -0.75.
template <int COPIES, int MODE>
__device__ __forceinline__ void skin_shared_tile(const SkinInstance& in0, const SkinChunk& ch, float4 (*s_rows)[SKIN_LDS_SLOTS],
	const float4* __restrict__ mesh_local, const uint8_t* __restrict__ tile_bones, const float4* __restrict__ palette, float* __restrict__ out) {
	constexpr uint32_t LPR = COPIES / 4;   // VGPR path: lanes per palette row
	constexpr uint32_t RPI = 64 / COPIES;  // DMA path: palette rows one instruction covers (1 KiB of LDS)
	constexpr uint32_t SHARED_WAVES = SHARED_THREADS / 64;
	constexpr uint32_t MAX_DMA = (SKIN_LDS_SLOTS / 64 + SHARED_WAVES - 1) / SHARED_WAVES; // DMA instructions per wave: <= 3
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t col = tid & (COPIES - 1);
	const float4* mbase = mesh_local + 2 * (size_t)ch.rec_offset;
	// the lane's vertex records: loaded once, used for every instance of the chunk, kept as loaded (8 VGPRs each). Lanes past the
	// tile's end take its LAST vertex (they recompute and rewrite the same 12 bytes): every store is unconditional.
	const uint32_t v_last = ch.v_end - 1; // chunks are never empty
	RawVertex vin[SHARED_VPT];
#pragma unroll
	for (int k = 0; k < SHARED_VPT; ++k) vin[k] = load_vertex(mbase, min(ch.v_begin + tid + k * SHARED_THREADS, v_last));
	// staging plan, the same for every instance: row f of the tile-local palette = row (f % 3) of bone tile_bones[f / 3]
	const uint32_t n_rows = ch.n_tile_bones * 3u;
	const uint32_t n_dma = (n_rows * LMX_SHARED_DMA_NUM / LMX_SHARED_DMA_DEN) / RPI; // DMA instructions per palette (whole instructions only)
	const uint32_t r_dma = n_dma * RPI;                                              // rows [0, r_dma) by DMA, [r_dma, n_rows) through VGPRs
	auto global_row = [&](uint32_t f) { // float4 index of tile-local row f inside an instance's palette
		const uint32_t lb = f / 3u;
		return (uint32_t)tile_bones[ch.bones_at + lb] * 3u + (f - lb * 3u);
	};
	uint32_t dma_off[MAX_DMA];
#pragma unroll
	for (uint32_t k = 0; k < MAX_DMA; ++k) dma_off[k] = global_row(min((wave + k * SHARED_WAVES) * RPI + lane / COPIES, n_rows - 1)) * (uint32_t)sizeof(float4);
	const uint32_t f = r_dma + tid / LPR;                // VGPR path: the row this lane fetches and writes four copies of
	const bool stages = f < n_rows;
	const uint32_t src_row = global_row(stages ? f : 0u); // (lanes without an item re-fetch row 0: no branch around the load)
	const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&s_rows[0][0]) + wave * 1024u;
	const size_t pal_stride = (size_t)in0.n_bones * 3;   // float4 per instance (consecutive instances: consecutive palettes)
	const float4* pal = palette + (size_t)in0.bone_offset * 3;
	auto stage_dma = [&](uint32_t instance, uint32_t buffer) {
		const float4* base = pal + instance * pal_stride;
		const uint32_t dst = lds0 + buffer * (uint32_t)(SKIN_LDS_SLOTS * sizeof(float4));
#pragma unroll
		for (uint32_t k = 0; k < MAX_DMA; ++k) {
			if (wave + k * SHARED_WAVES < n_dma) lds_dma_16(dma_off[k], base, dst + k * SHARED_WAVES * 1024u); // wave-uniform branch
		}
	};
	auto spread = [&](uint32_t buffer, float4 t) { // palette_spread for row f (rows below r_dma belong to the DMA)
		if (stages) {
			const uint32_t q = tid % LPR;
			const uint32_t rot = COPIES == 4 ? (f >> 1) : LPR * f;
#pragma unroll
			for (uint32_t i = 0; i < 4; ++i) s_rows[buffer][f * COPIES + ((q + LPR * i + rot) & (COPIES - 1))] = t;
		}
	};
	F3* obase = reinterpret_cast<F3*>(out) + in0.out_offset;
	const uint32_t last = ch.count - 1;
	stage_dma(0, 0);
	spread(0, pal[src_row]);
	// every load so far (vertex records, first palette incl. its DMA part, which the compiler does not know) is complete before the
	// loop; the compiler's s_waitcnt placement merges the loop-entry state into the steady state, and with loads possibly pending
	// at the entry it tightens the waits INSIDE the loop. vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding).
	__builtin_amdgcn_s_waitcnt(0x0F70);
	__syncthreads();
	for (uint32_t j = 0; j <= last; ++j) {
		// Next palette: DMA part first, then the row load of the VGPR part, both BEFORE this instance's stores. Loads, DMA and stores
		// retire through ONE in-order counter (vmcnt): the wait for the row load, placed after k of the lane's stores, must be
		// `s_waitcnt vmcnt(k)` - it then covers the row load, the older DMAs and the stores of the previous instance, never this
		// instance's own. The compiler emits that count only when it can count: no branch around a compiler-visible load or a store
		// in this loop (lanes past the end of the tile or the palette are clamped, the last instance re-stages its own palette).
		const uint32_t next = min(j + 1, last);
		stage_dma(next, (j + 1) & 1);
		float4 t = pal[next * pal_stride + src_row];
		__builtin_amdgcn_sched_barrier(0); // the load stays here, ahead of the stores
		const float4* rows = s_rows[j & 1] + col;
		F3* o = obase + (size_t)j * in0.n_verts;
#pragma unroll
		for (int k = 0; k < SHARED_VPT; ++k) {
			const uint32_t v_wave = ch.v_begin + wave * 64u + k * SHARED_THREADS; // first vertex of this wave's step
			const uint32_t v = min(v_wave + lane, v_last);
			// opaque to the optimiser: nothing derived from the record (LDS addresses, operand pairs) is hoisted out of the instance
			// loop into registers that do not exist
			asm volatile("" : "+v"(vin[k].a), "+v"(vin[k].b));
			const F3 r = skin_blend<COPIES, MODE>(rows, vin[k]);
			if (!LMX_PROBE_SKIP(8) || r.x == 123.25f) {
				store_position(o + v, r);
			}
			__builtin_amdgcn_sched_barrier(0); // one vertex's 12 palette rows (48 VGPRs) in flight at a time
			if (k == SHARED_SPREAD_AFTER) {
				// the VGPR part of the next palette goes into the other buffer late in the instance (sooner, the wait for its load
				// stalls). `t` is used by every lane here: without that its load is sunk into the branch, behind the stores, and
				// waited for with vmcnt(0). (After the last instance: into the idle buffer, never read.)
				asm volatile("" : "+v"(t.x), "+v"(t.y), "+v"(t.z), "+v"(t.w));
				spread((j + 1) & 1, t);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		__syncthreads(); // buffer (j + 1) & 1 is complete for every wave; buffer j & 1 is free for instance j + 2
	}
}

template <int MODE>
__global__ __launch_bounds__(SHARED_THREADS) void k_skin_shared(const SkinInstance* __restrict__ inst, const SkinChunk* __restrict__ chunks,
	const float4* __restrict__ mesh_local, const uint8_t* __restrict__ tile_bones, const float4* __restrict__ palette, float* __restrict__ out) {
	__shared__ float4 s_rows[2][SKIN_LDS_SLOTS];
	const SkinChunk ch = chunks[blockIdx.x];
	const SkinInstance in0 = inst[ch.first_inst];
	if (ch.n_tile_bones <= 64) skin_shared_tile<16, MODE>(in0, ch, s_rows, mesh_local, tile_bones, palette, out);
	else if (ch.n_tile_bones <= 128) skin_shared_tile<8, MODE>(in0, ch, s_rows, mesh_local, tile_bones, palette, out);
	else skin_shared_tile<4, MODE>(in0, ch, s_rows, mesh_local, tile_bones, palette, out);
}

// ---- shared-mesh runs, several instances per block: the palette is staged ONCE per block, the vertex records stream ------------------
// k_skin_shared stages 48 KiB of replicated palette per 5120 outputs (one instance x one register-resident tile): 9.6 bytes of LDS
// writes per skinned vertex, a barrier and an in-order vmcnt wait per instance - the 0.5 ms per 1e9 vertices round 2's term-by-term
// probe put on "palette staging". Here the 16 bank columns of the replicated palette hold I DIFFERENT instances (16 / I copies each):
// column c = (copy, instance c % I), and lane l - bank column l % 16 as before, so every lane of a ds_read_b128 service group still owns
// its column whatever the bone indices - skins vertex l / I of the wave's step for instance l % I. One staging per block then serves
// I instances x the block's whole vertex range (48 KiB per I x ~10 k outputs: under 1.3 B per vertex at I = 4), the steady state has no
// barrier and no palette traffic, and the block is 8 waves with 48 KiB of LDS: three resident per CU instead of one.
// The price: vertex records are no longer register-resident; they stream through a PIPE-deep software pipeline as in k_skin_vertices
// (64 / I distinct records per wave-step, every one shared by I lanes: 32 / I bytes of L2 traffic per output), and a store instruction
// writes I runs of 64 / I consecutive vertices (I = 4: four runs of 192 bytes) instead of one run of 768 bytes.
#ifndef LMX_MULTI_SKIP_ZERO
#define LMX_MULTI_SKIP_ZERO 1 // LMX_SKIN_FUSED: bone slots whose weight is zero in every lane of the wave are neither read nor multiplied
#endif
template <int COLS, int I, int MODE, int PIPE, int THREADS>
__device__ __forceinline__ void skin_multi_tile(const SkinMultiChunk& ch, float4* s_rows, const float4* __restrict__ mesh,
	const float4* __restrict__ palette, float* __restrict__ out, uint32_t& sink0, uint32_t& sink1 /* destinations of the caller's uncounted touches */) {
	static_assert(I >= 1 && I <= COLS && (COLS % I) == 0 && (64 % I) == 0, "instances per block divide the bank columns");
	constexpr uint32_t VPW = 64 / I;                   // vertices per wave-step
	constexpr uint32_t VPB = VPW * (THREADS / 64); // vertices per block-step: consecutive across the block's waves
	constexpr uint32_t ROWS = skin_rows(MODE);
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint32_t last_inst = ch.count - 1;            // 1 <= count <= I; lanes of missing instances redo the last one (same bytes, same place)
	const uint32_t inst_l = min(lane % I, last_inst);
	const size_t pal_stride = (size_t)ch.n_bones * ROWS; // float4 per instance: consecutive instances, consecutive palettes
	const float4* rows = s_rows + (lane & (COLS - 1));
	const float4* mbase = mesh + 2 * (size_t)ch.rec_offset;
	F3* obase = reinterpret_cast<F3*>(out) + ch.out_offset + (size_t)inst_l * ch.n_verts; // per-lane base: the lane's instance
	const uint32_t v_last = ch.v_end - 1; // chunks are never empty
	const uint32_t v0 = ch.v_begin + wave * VPW + lane / I;
	auto load = [&](uint32_t v) { return load_vertex(mbase, v); };
	// the first PIPE vertex records and the palette rows: ONE round trip - all loads go out before anything waits
	RawVertex rec[PIPE];
#pragma unroll
	for (int d = 0; d < PIPE; ++d) rec[d] = load(min(v0 + d * VPB, v_last));
	__builtin_amdgcn_sched_barrier(0); // (issued here, ahead of the palette's loads and waits; an asm use of the records would WAIT for them here)
	{
		const float4* pal = palette + (size_t)ch.bone_offset * ROWS;
		const uint32_t items = ch.n_stage * ROWS * COLS; // only the bones the mesh references are staged (n_stage = its largest bone index + 1)
		// consecutive lanes fill consecutive columns of one row: conflict-free ds_write_b128; the COLS / I lanes of one instance fetch the same
		// 16 bytes. All of a lane's loads go out before the first is written (items past the palette's end re-read its first row).
		constexpr uint32_t PER_LANE = SKIN_LDS_SLOTS / THREADS;
		float4 t[PER_LANE];
#pragma unroll
		for (uint32_t k = 0; k < PER_LANE; ++k) {
			const uint32_t w = tid + k * THREADS;
			t[k] = pal[w < items ? min((w % COLS) % I, last_inst) * pal_stride + w / COLS : 0u];
		}
#pragma unroll
		for (uint32_t k = 0; k < PER_LANE; ++k) asm volatile("" : "+v"(t[k].x), "+v"(t[k].y), "+v"(t[k].z), "+v"(t[k].w)); // used HERE by every lane: the loads are not sunk into the branches below
#pragma unroll
		for (uint32_t k = 0; k < PER_LANE; ++k) {
			const uint32_t w = tid + k * THREADS;
			if (w < items) s_rows[w] = t[k];
		}
	}
	// every load so far is complete before the loop: with loads possibly pending at the loop's entry the compiler merges that state into
	// the steady state and tightens the wait at the loop head to cover the previous step's STORE (seen in the ISA: vmcnt(2) instead of
	// vmcnt(2 * PIPE)). vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding).
	__builtin_amdgcn_s_waitcnt(0x0F70);
	asm volatile("" : : "v"(sink0), "v"(sink1)); // (the touches have landed: their registers are free from here on)
	__syncthreads();
	const uint32_t n_steps = (ch.v_end - ch.v_begin + VPB - 1) / VPB; // block-uniform
	// the pipeline of skin_tile: no branch around a load or a store (steps and lanes past the range's end are clamped to its last vertex),
	// so that the compiler emits exact vmcnt counts: a wave waits for the record loaded PIPE steps ago, never for its newest stores
	for (uint32_t it = 0; it < n_steps; it += PIPE) {
#pragma unroll
		for (int d = 0; d < PIPE; ++d) {
			const uint32_t v = min(v0 + (it + d) * VPB, v_last);
			asm volatile("" : "+v"(rec[d].a), "+v"(rec[d].b));
			F3 o;
			if constexpr (MODE == LMX_SKIN_FUSED && LMX_MULTI_SKIP_ZERO) {
				// wave-uniform: how many of the four bone slots carry a weight in ANY lane (slots are used front to back by every rigging tool;
				// a zero in the middle just counts as used)
				const bool u2 = __ballot(rec[d].a.y != 0.f) != 0, u3 = __ballot(rec[d].a.z != 0.f) != 0, u4 = __ballot(rec[d].a.w != 0.f) != 0;
				if (u3 || u4) o = skin_blend<COLS, MODE>(rows, rec[d]);
				else if (u2) o = skin_blend_fused_n<COLS, 2>(rows, rec[d]);
				else o = skin_blend_fused_n<COLS, 1>(rows, rec[d]);
			} else {
				o = skin_blend<COLS, MODE>(rows, rec[d]);
			}
			asm volatile("" : "+v"(o.x), "+v"(o.y), "+v"(o.z));
			__builtin_amdgcn_sched_barrier(0);
			rec[d] = load(min(v0 + (it + d + PIPE) * VPB, v_last));
			if (!LMX_PROBE_SKIP(8) || o.x == 123.25f) store_position(obase + v, o);
			__builtin_amdgcn_sched_barrier(0);
		}
	}
}

#ifndef LMX_MULTI_PIPE
#define LMX_MULTI_PIPE 2
#endif
#ifndef LMX_MULTI_THREADS
#define LMX_MULTI_THREADS 512 // 512: three 8-wave blocks per CU (48 KiB of LDS each); 1024: two 16-wave blocks (32 waves per CU, the VGPR budget of 64 holds)
#endif
constexpr int MULTI_THREADS = LMX_MULTI_THREADS;
#ifndef LMX_MULTI_PREFETCH
#define LMX_MULTI_PREFETCH 768 // blocks ahead (a multiple of 8: block b and block b + 768 run on the same XCD, i.e. behind the same L2)
#endif
template <int I, int MODE>
__global__ __launch_bounds__(MULTI_THREADS, MULTI_THREADS == 1024 ? 8 : SKIN_WAVES_PER_SIMD) void k_skin_multi(const SkinMultiChunk* __restrict__ chunks, uint32_t n_chunks,
	const float4* __restrict__ mesh, const float4* __restrict__ palette, float* __restrict__ out) {
	// sized by the launch for the largest staging of its chunks (<= SKIN_LDS_SLOTS): a 52-bone character stages 39 KiB, not 48, and a CU
	// holds four blocks of it instead of three
	LMX_DYNAMIC_LDS(float4, s_rows);
	const SkinMultiChunk ch = chunks[blockIdx.x];
	uint32_t sink0 = 0, sink1 = 0;
	if (LMX_MULTI_PREFETCH != 0) {
		// A block cannot blend a vertex before its palettes are in LDS, and with the memory system saturated by 12 bytes of stores per
		// vertex a palette row that has to come from HBM takes several microseconds of a ~40 us block (100 k instances: 307 MB of
		// palettes, more than the Infinity Cache; measured 3.56 ms per 1e9 vertices against 2.45 with 20 k instances, whose palettes stay
		// cached). So every block also TOUCHES the palettes of the block that will run where it runs now - block b + 768: three resident
		// blocks on each of 256 CUs, and the same XCD (b % 8) - with loads whose results nobody waits for: by the time that block
		// starts, its rows sit in its XCD's L2. The touches are issued before anything else, so the in-order vmcnt waits of the
		// staging below cover them without counting them.
		const uint32_t ahead = blockIdx.x + LMX_MULTI_PREFETCH;
		if (ahead < n_chunks) { // block-uniform
			const SkinMultiChunk nx = chunks[ahead];
			const uint32_t pieces = (nx.n_bones * skin_rows(MODE) * nx.count + 3u) / 4u; // 64-byte pieces of the next block's palettes (contiguous): <= 588
			const char* src = reinterpret_cast<const char*>(palette + (size_t)nx.bone_offset * skin_rows(MODE));
#ifndef LMX_HOSTSIM
			// 4 bytes of a piece pull all of it in. The destination registers stay allocated until the staging's vmcnt(0) (skin_multi_tile
			// consumes them there): a load the compiler does not know about must not land in a register it has handed to something else.
			if (threadIdx.x < pieces) asm volatile("global_load_dword %0, %1, off" : "=v"(sink0) : "v"(src + 64u * threadIdx.x) : "memory");
			if (threadIdx.x + MULTI_THREADS < pieces) asm volatile("global_load_dword %0, %1, off" : "=v"(sink1) : "v"(src + 64u * (threadIdx.x + MULTI_THREADS)) : "memory");
#else
			(void)pieces; (void)src;
#endif
		}
	}
	if (ch.n_stage <= 64) skin_multi_tile<16, I, MODE, LMX_MULTI_PIPE, MULTI_THREADS>(ch, s_rows, mesh, palette, out, sink0, sink1);
	else if (ch.n_stage <= 128) skin_multi_tile<8, (I < 8 ? I : 8), MODE, LMX_MULTI_PIPE, MULTI_THREADS>(ch, s_rows, mesh, palette, out, sink0, sink1);
	else skin_multi_tile<4, (I < 4 ? I : 4), MODE, LMX_MULTI_PIPE, MULTI_THREADS>(ch, s_rows, mesh, palette, out, sink0, sink1);
}

} // namespace

// instances one k_skin_multi block serves for a model of n_bones bones when the launch is asked for `per_block`
uint32_t skin_multi_instances(uint32_t per_block, uint32_t n_bones) {
	const uint32_t cols = n_bones <= 64 ? 16u : (n_bones <= 128 ? 8u : 4u);
	return per_block < cols ? per_block : cols;
}

template <int I>
static hipError_t launch_skin_multi_i(hipStream_t s, const SkinMultiChunk* chunks, uint32_t n_chunks, const float4* mesh, const float4* palette, float* out, int mode, size_t lds) {
	const dim3 grid(n_chunks), block(MULTI_THREADS);
	if (mode == LMX_SKIN_EXACT) hipLaunchKernelGGL((k_skin_multi<I, LMX_SKIN_EXACT>), grid, block, lds, s, chunks, n_chunks, mesh, palette, out);
	else if (mode == LMX_SKIN_DQS) hipLaunchKernelGGL((k_skin_multi<I, LMX_SKIN_DQS>), grid, block, lds, s, chunks, n_chunks, mesh, palette, out);
	else hipLaunchKernelGGL((k_skin_multi<I, LMX_SKIN_FUSED>), grid, block, lds, s, chunks, n_chunks, mesh, palette, out);
	return hipGetLastError();
}

// float4 slots of LDS a k_skin_multi block stages for a mesh that references bones [0, n_stage) (3 rows a bone; the dual-quaternion mode's 2 fit).
// NOT monotonic in n_stage - the bank columns halve at 65 and 129 bones - so a launch is sized by the largest value over its chunks.
uint32_t skin_multi_lds_slots(uint32_t n_stage) {
	const uint32_t cols = n_stage <= 64 ? 16u : (n_stage <= 128 ? 8u : 4u);
	return n_stage * 3u * cols;
}

hipError_t launch_skin_multi(hipStream_t s, uint32_t per_block, const SkinMultiChunk* chunks, uint32_t n_chunks, uint32_t slots, const float4* mesh, const float4* palette, float* out, int mode) {
	if (!n_chunks) return hipSuccess;
	if (slots == 0 || slots > (uint32_t)SKIN_LDS_SLOTS) return hipErrorInvalidValue;
	const size_t lds = (size_t)slots * sizeof(float4);
	switch (per_block) {
	case 1: return launch_skin_multi_i<1>(s, chunks, n_chunks, mesh, palette, out, mode, lds);
	case 2: return launch_skin_multi_i<2>(s, chunks, n_chunks, mesh, palette, out, mode, lds);
	case 4: return launch_skin_multi_i<4>(s, chunks, n_chunks, mesh, palette, out, mode, lds);
	case 8: return launch_skin_multi_i<8>(s, chunks, n_chunks, mesh, palette, out, mode, lds);
	case 16: return launch_skin_multi_i<16>(s, chunks, n_chunks, mesh, palette, out, mode, lds);
	default: return hipErrorInvalidValue;
	}
}

// groups are sorted by capacity class on the host: [0, n0) hold <= 4 instances of <= 64 bones, then 2 x <= 128, then 1 x <= 196
// (POSE_GROUP_CAP_SHIFT, lmx_kernels.h)
hipError_t launch_pose_palette(hipStream_t s, const SkinInstance* inst, const PoseGroup* groups, const uint32_t n_groups[3], const float* rel_pos,
	const float4* rel_rot, float* pose_pos, float4* pose_rot, const uint32_t* level_items, const uint16_t* level_off, const float* inv_pos,
	const float4* inv_rot, float4* palette, float4* dual_quats) {
	if (n_groups[0])
		hipLaunchKernelGGL((k_pose_palette<64, POSE_GROUP_SHIFT>), dim3(n_groups[0]), dim3(64), 0, s, inst, groups, rel_pos, rel_rot, pose_pos, pose_rot, level_items, level_off, inv_pos,
			inv_rot, palette, dual_quats);
	if (n_groups[1])
		hipLaunchKernelGGL((k_pose_palette<128, (POSE_GROUP_SHIFT > 1 ? POSE_GROUP_SHIFT - 1 : 0)>), dim3(n_groups[1]), dim3(64), 0, s, inst, groups + n_groups[0], rel_pos, rel_rot, pose_pos, pose_rot, level_items,
			level_off, inv_pos, inv_rot, palette, dual_quats);
	if (n_groups[2])
		hipLaunchKernelGGL((k_pose_palette<196, (POSE_GROUP_SHIFT > 2 ? POSE_GROUP_SHIFT - 2 : 0)>), dim3(n_groups[2]), dim3(64), 0, s, inst, groups + n_groups[0] + n_groups[1], rel_pos, rel_rot, pose_pos, pose_rot,
			level_items, level_off, inv_pos, inv_rot, palette, dual_quats);
	return hipGetLastError();
}

hipError_t launch_pose_blend(hipStream_t s, float* pos, float4* rot, const float* rhs_pos, const float4* rhs_rot, size_t n_bones, float weight) {
	if (!n_bones) return hipSuccess;
	hipLaunchKernelGGL(k_pose_blend, dim3((uint32_t)((n_bones + 255) / 256)), dim3(256), 0, s, pos, rot, rhs_pos, rhs_rot, n_bones, weight);
	return hipGetLastError();
}

hipError_t launch_palette_expand(hipStream_t s, const float4* rows, uint32_t n_bones, float4* out) {
	if (!n_bones) return hipSuccess;
	hipLaunchKernelGGL(k_palette_expand, dim3((n_bones + 255) / 256), dim3(256), 0, s, rows, n_bones, out);
	return hipGetLastError();
}

hipError_t launch_skin_vertices(hipStream_t s, const SkinInstance* inst, const uint32_t* inst_index, uint32_t n_inst, uint32_t max_verts,
	const float4* mesh, const float4* palette, float* out, int mode) {
	if (!n_inst || !max_verts) return hipSuccess;
	// tiles: as large as possible (the 48 KiB palette staging is paid per tile) while still giving the chip >= ~3000 blocks
	const uint32_t max_tiles = (max_verts + 1023u) / 1024u;
	uint32_t tiles = (3072u + n_inst - 1) / n_inst;
	if (tiles > max_tiles) tiles = max_tiles;
	if (tiles < 1) tiles = 1;
	const uint32_t tile_verts = (max_verts + tiles - 1) / tiles;
	const uint64_t blocks = (uint64_t)tiles * n_inst;
	if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
	const dim3 grid((uint32_t)blocks), block(SKIN_THREADS);
	if (mode == LMX_SKIN_EXACT) hipLaunchKernelGGL(k_skin_vertices<LMX_SKIN_EXACT>, grid, block, 0, s, inst, inst_index, tiles, tile_verts, mesh, palette, out);
	else if (mode == LMX_SKIN_DQS) hipLaunchKernelGGL(k_skin_vertices<LMX_SKIN_DQS>, grid, block, 0, s, inst, inst_index, tiles, tile_verts, mesh, palette, out);
	else hipLaunchKernelGGL(k_skin_vertices<LMX_SKIN_FUSED>, grid, block, 0, s, inst, inst_index, tiles, tile_verts, mesh, palette, out);
	return hipGetLastError();
}

hipError_t launch_skin_shared(hipStream_t s, const SkinInstance* inst, const SkinChunk* chunks, uint32_t n_chunks, const float4* mesh_local,
	const uint8_t* tile_bones, const float4* palette, float* out, int mode) {
	if (!n_chunks) return hipSuccess;
	const dim3 grid(n_chunks), block(SHARED_THREADS);
	if (mode == LMX_SKIN_EXACT) hipLaunchKernelGGL(k_skin_shared<LMX_SKIN_EXACT>, grid, block, 0, s, inst, chunks, mesh_local, tile_bones, palette, out);
	else if (mode == LMX_SKIN_DQS) return hipErrorInvalidValue; // the dual-quaternion blend needs ~75 VGPRs of its own: with five resident records it spilled (284 B of scratch per lane); lmx_skin_run sends LMX_SKIN_DQS through k_skin_vertices (61 VGPRs)
	else hipLaunchKernelGGL(k_skin_shared<LMX_SKIN_FUSED>, grid, block, 0, s, inst, chunks, mesh_local, tile_bones, palette, out);
	return hipGetLastError();
}

} // namespace lmx
