// keys_kernels.hip — PipelineImpl::createSortKeys (renderer/pipeline.cpp:3789-3968) on gfx950, straight from the visible list
// a cull left in HBM.
//
//   k_keys_mesh      one lane per visible MESH entity: LOD selection (fp64 squared distance to the LOD reference point
//                    :3876, Model::getLODMeshIndices model.h:173-179, the ModelInstance::lod transition :3937-3957) and
//                    create_key (:3884-3935) for every mesh of the selected LOD range(s): the materials are read ONCE (loads back to
//                    back) and classified, a tile reserves its output ranges with two atomics, the pairs / records are built from
//                    registers into LDS and leave as contiguous stores.
//   k_keys_decal     DECAL / CURVE_DECAL pages (:3841-3868).
//   k_keys_reduce_copies + k_keys_offsets   per-key sum over the private counter copies, then the exclusive scan of the group sizes (AutoInstancer::instances, :452-523) -> CSR offsets
//   k_keys_scatter   instancer records -> CSR values; the lane that places a group's FIRST value also pushes the group's
//                    AUTOINSTANCED pair (:3958-3968)
// The visible ids are read where the cull kernels left them (KeysShardList: one window per output shard); no fill, no gather and
// no single-purpose launch in the chain: TWO launches for key ranges up to 1024 (k_keys_mesh, k_keys_scatter), four beyond (round 4: 8).
// Integer work is bit-exact by construction; the two fp64 -> fp32 distances use the reference's operation order.
#include "lmx_kernels.h"

#include <algorithm>

namespace lmx {

namespace {

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t rank_in(uint64_t mask) {
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// wave-aggregated append: every lane with `want` gets a distinct index of a list whose length lives at *counter
__device__ __forceinline__ uint32_t wave_append(bool want, uint32_t* counter) {
	const uint64_t mask = __ballot(want);
	if (mask == 0) return 0;
	uint32_t base = 0;
	const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
	if (lane_id() == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
	base = __shfl(base, (int)leader);
	return base + rank_in(mask);
}

// histogram increment with one atomic per distinct key of the wave (a scene of one model puts every lane on one counter)
__device__ __forceinline__ void wave_histogram(bool want, uint32_t key, uint32_t* hist) {
	uint64_t todo = __ballot(want);
	while (todo) {
		const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1u;
		const uint32_t k = (uint32_t)__shfl((int)key, (int)leader);
		const uint64_t same = __ballot(want && key == k) & todo;
		if (lane_id() == leader) atomicAdd(hist + k, (uint32_t)__popcll(same));
		todo &= ~same;
	}
}

// floatFlip, pipeline.cpp:57-60
__device__ __forceinline__ uint32_t float_flip(uint32_t bits) {
	const uint32_t mask = (uint32_t)(-(int32_t)(bits >> 31)) | 0x80000000u;
	return bits ^ mask;
}

// ---- walking a KeysShardList -------------------------------------------------------------------------------------------------------
// A block first copies the list's shard counts and window starts into LDS (s_cnt / s_win, KEYS_MAX_SHARDS words each) and forms
// s_first[s] = the sum of unit(count) over the shards before s (s_first[n] = the total), all from threads < KEYS_MAX_SHARDS.
template <typename Unit> __device__ __forceinline__ void shard_list_to_lds(const KeysShardList& L, uint32_t* s_cnt, uint32_t* s_win, uint32_t* s_first, Unit unit) {
	const uint32_t t = threadIdx.x;
	if (t < (uint32_t)KEYS_MAX_SHARDS) {
		const bool in = t < L.n;
		s_cnt[t] = in ? L.counts[(size_t)t * L.cnt_pad] : 0u;
		s_win[t] = in && L.win_base != nullptr ? L.win_base[t] : 0u;
	}
	__syncthreads();
	if (t <= (uint32_t)KEYS_MAX_SHARDS) {
		uint32_t run = 0;
		for (uint32_t k = 0; k < t && k < L.n; ++k) run += unit(s_cnt[k]);
		s_first[t] = run;
	}
	__syncthreads();
}

// The same walk out of registers, per wave and without LDS or barriers (k_keys_mesh: a block has about one tile, so what stands in front
// of the tile's first load is paid per tile - through LDS and two barriers the kernel was 53.5 us instead of 48): lane l holds shards l
// and l + 64, first[] = units in the shards before (wave scan), a unit's shard = the last one whose first[] is <= it (empty shards share
// their first[] with the next non-empty one and are never that).
struct ShardWalk {
	uint32_t cnt[2], win[2], first[2], total;
	template <typename Unit> __device__ __forceinline__ void load(const KeysShardList& L, uint32_t lane, Unit unit) {
		uint32_t u[2];
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const uint32_t sh = lane + 64u * h;
			const bool in = sh < L.n;
			cnt[h] = in ? L.counts[(size_t)sh * L.cnt_pad] : 0u;
			win[h] = in && L.win_base != nullptr ? L.win_base[sh] : 0u;
		}
		uint32_t before = 0;
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			u[h] = unit(cnt[h]);
			uint32_t incl = u[h];
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
				if (lane >= (uint32_t)o) incl += up;
			}
			first[h] = before + incl - u[h];
			before += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		}
		total = before;
	}
	// unit t < total: its shard's count, window start and first unit (wave-uniform)
	__device__ __forceinline__ void locate(uint32_t t, uint32_t& sh_cnt, uint32_t& sh_win, uint32_t& sh_first) const {
		const uint32_t below = (uint32_t)__popcll(__ballot(first[0] <= t)) + (uint32_t)__popcll(__ballot(first[1] <= t)); // >= 1: first[0] of lane 0 is 0
		const uint32_t sh = (uint32_t)__builtin_amdgcn_readfirstlane((int)(below - 1u));
		const bool hi = sh >= 64u;
		sh_cnt = (uint32_t)__builtin_amdgcn_readlane((int)(hi ? cnt[1] : cnt[0]), (int)(sh & 63u));
		sh_win = (uint32_t)__builtin_amdgcn_readlane((int)(hi ? win[1] : win[0]), (int)(sh & 63u));
		sh_first = (uint32_t)__builtin_amdgcn_readlane((int)(hi ? first[1] : first[0]), (int)(sh & 63u));
	}
};

constexpr int KEYS_BLOCK = 512; // entities per tile = threads per block. 8 waves: 3 blocks per CU (79 VGPRs: 6 waves per SIMD). Tiles of 256 / 1024 entities measured slower (161.8 / 145.0 against 138.1 us for the whole chain, round 3's driver box)
#ifndef LMX_KEYS_MM_REGS
#define LMX_KEYS_MM_REGS 6 // every mesh of two LODs of three
#endif
constexpr int KEYS_MM_REGS = LMX_KEYS_MM_REGS; // meshes of an entity's LOD range(s) whose materials stay in registers between the count and the emit
constexpr int KEYS_STAGE_PAIRS = 3 * KEYS_BLOCK; // pairs (24 KiB) and records (18 KiB) of one 512-entity tile held in LDS

// The visible list is walked in tiles of 512 entities by a fixed-size grid. Per tile every lane first COUNTS what it will
// emit, the block reserves its four output ranges with two 64-bit atomics on two cache lines (returning atomics on one line retire at
// ~90 per microsecond chip-wide: one per wave and mesh was 15x slower than this kernel's memory work), and a second walk writes at
// lane-private positions.
#ifndef LMX_KEYS_PROBE
#define LMX_KEYS_PROBE 0 // timing probes (tools/build_variant.py): 1 = no group histogram, 4 = no tile reservations, 8 = k_keys_scatter without its cursor atomics, 16 = without its stores; results are wrong
#endif
#ifndef LMX_KEYS_LDS_HIST
#define LMX_KEYS_LDS_HIST 1 // the instancer's group histogram per tile in LDS, one global atomic per (tile, key) instead of one per record
#endif
constexpr int KEYS_HIST_LDS = 4096; // keys (16 KiB): larger ranges keep the global atomics
constexpr int KEYS_ROW_SHIFT = 12;   // block ranks: rec_key = key (< KEYS_HIST_LDS) | k_keys_mesh block << 12
constexpr uint32_t KEYS_RANK_DEAD = 0xffffffffu; // block ranks: the rank of a record whose mesh sort key lies above max_sort_key (dropped by the scatter, as the privatised-counter path drops it)
static_assert((1 << KEYS_ROW_SHIFT) == KEYS_HIST_LDS, "a record's key and row share 32 bits");
#ifndef LMX_KEYS_MIN_WAVES
#define LMX_KEYS_MIN_WAVES 4 // waves per SIMD the register allocation aims at: 106 VGPRs, no scratch, two 8-wave blocks per CU. Round 4 (profiles/r04/keys_ab.txt, k_keys_mesh per 1.05 M visible): 6 waves (80 VGPRs, 32-44 B of scratch, three blocks) 65.5-70.5 us, 5 waves 59.1, 4 waves 60.2 - the kernel is not short of waves, spills cost it more
#endif
__global__ __launch_bounds__(KEYS_BLOCK, LMX_KEYS_MIN_WAVES) void k_keys_mesh(KeysDevice d, const KeysViewDevice kv /* by value: captured at launch */,
	const KeysShardList L /* the visible MESH entities; L.slots optional: static-set slot per id, -1 = dynamic set */) {
	const int32_t* __restrict__ ids = L.ids;
	const int32_t* __restrict__ slots = L.slots;
	// tiles are cut per shard window (a window's last tile is partial)
	__shared__ uint32_t s_wave[KEYS_BLOCK / 64][3]; // per wave: pairs | recs << 16, poses, dirty
	__shared__ uint32_t s_base[6]; // bases of the four lists; [4], [5]: this tile's pairs / records (LMX_KEYS_STAGE_PAIRS)
	__shared__ uint32_t s_bucket[256]; // bucket_map: an LDS read instead of one more dependent global load per mesh
	if (threadIdx.x < 255) s_bucket[threadIdx.x] = kv.bucket_map[threadIdx.x];
	// The instancer's group histogram of a TILE is collected in LDS (key ranges up to KEYS_HIST_LDS) and leaves as one atomic per key the
	// tile saw, from consecutive lanes - one global atomic per instancer record, from inside the emit loop, was 10 of the kernel's 56 us
	// (profiles/r04/keys_probes2.txt): they are executed memory-side, a few tens of thousands per microsecond over the whole chip.
	__shared__ uint32_t s_hist[KEYS_HIST_LDS];
	const bool lds_hist = LMX_KEYS_LDS_HIST != 0 && d.max_sort_key < (uint32_t)KEYS_HIST_LDS; // launch-uniform
	// BLOCK RANKS (round 5; the host offers d.block_rows for key ranges that fit the LDS histogram): the histogram stays in LDS over ALL
	// tiles of the block and leaves once, as the block's own row of a table, and every instancer record carries its rank among the
	// block's records of its key (what the LDS increment returns). The row's entries are the records of the key in the blocks that
	// finished before (below) and a record's place in its group is offset + row[block][key] + rank: neither this kernel's ~300 k global
	// histogram adds nor the scatter's 560 k returning cursor atomics (7 of its 18 us, profiles/r05/keys_scatter_probes.txt) exist any
	// more - 75 k adds at the blocks' ends instead. (A row per TILE, 2114 of them for the headline view, and a kernel scanning the
	// columns took 12-25 us: profiles/r05/keys_rows_per_tile.txt; per block 5.4 us + a launch gap.)
	const bool block_ranks = lds_hist && d.block_rows != nullptr; // launch-uniform
	if (lds_hist) {
		for (uint32_t k = threadIdx.x; k <= d.max_sort_key; k += KEYS_BLOCK) s_hist[k] = 0;
	}
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	ShardWalk walk;
	walk.load(L, lane, [](uint32_t c) { return (c + (uint32_t)KEYS_BLOCK - 1u) / (uint32_t)KEYS_BLOCK; });
	const uint32_t n_tiles = walk.total;
	__syncthreads(); // s_bucket / s_hist are in
	const uint32_t copy = blockIdx.x & (d.n_copies - 1); // this block's private row of the group counters
	// where tile t's ids start and how many it holds
	auto locate = [&](uint32_t t, uint32_t& at, uint32_t& cnt) {
		uint32_t sh_cnt, sh_win, sh_first;
		walk.locate(t, sh_cnt, sh_win, sh_first);
		const uint32_t j0 = (t - sh_first) * (uint32_t)KEYS_BLOCK;
		at = sh_win + j0;
		cnt = min(sh_cnt - j0, (uint32_t)KEYS_BLOCK);
	};
	// A block walks its tiles one after the other and a tile is a chain of dependent loads (id -> record -> model -> materials) in front of
	// three barriers: the next tile's id and slot - the chain's first link - are fetched at the top of the current tile.
	// (Measured and NOT kept, round 5, profiles/r05/keys_record_prefetch_and_lds_models.txt: the next tile's whole record fetched under the
	// current tile's emit + the models' LOD tables staged in LDS - two links fewer - 43.8 against 44.3 us, 123 VGPRs: with the atomics
	// compiled out the kernel still takes 37 us for 150 MB of scattered 32-64 B accesses; it is the memory system, not the chain.)
	uint32_t e_next = 0, cnt_next = 0;
	int32_t slot_next = -1;
	if (blockIdx.x < n_tiles) {
		uint32_t at;
		locate(blockIdx.x, at, cnt_next);
		if (threadIdx.x < cnt_next) { e_next = (uint32_t)ids[at + threadIdx.x]; slot_next = slots != nullptr ? slots[at + threadIdx.x] : -1; }
	}
	for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const bool has = threadIdx.x < cnt_next; // this lane holds an entity of the tile
		const uint32_t e_now = e_next;
		const int32_t slot_now = slot_next;
		if (tile + gridDim.x < n_tiles) {
			uint32_t at;
			locate(tile + gridDim.x, at, cnt_next);
			if (threadIdx.x < cnt_next) { e_next = (uint32_t)ids[at + threadIdx.x]; slot_next = slots != nullptr ? slots[at + threadIdx.x] : -1; }
		}
		// ranges of mesh indices this lane emits keys for: [from0, to0] then [from1, to1]
		int32_t from0 = 0, to0 = -1, from1 = 0, to1 = -1;
		uint32_t e = 0, pose_stamp = 0;
		int32_t slot = -1; // the entity's slot in the sorted set's mirror, or -1
		bool moved = false, queue_dirty = false;
		uint32_t depth_key = 0; // makeDepthSortKey's floatFlip(squared distance to the camera), used by depth-sorted buckets only
		KeysInstance* rec = nullptr;                   // the entity's record: by slot for the sorted set, by entity otherwise
		float* lod_at = nullptr;                       // where ModelInstance::lod and Pose::frame of the entity live: in the record, or
		                                               // (sorted set, LMX_KEYS_SPLIT_STATE) in the dense per-slot array
		const LmxMeshMaterial* mmb = d.mesh_materials; // ... and the table its material_offset points into; from the LOD ranges on: the entity's first material
		if (has) {
			e = e_now;
			KeysInstance in;
			in.model = -1;
			const int32_t sl = slot_now;
			slot = sl;
			// (world binding) the entity's slot in the hierarchy: depends on `e` alone, so it travels with the record's loads - read where the
			// position is needed it was one more link in the chain of dependent loads, behind the model table's
			int32_t world_slot = -1;
			if (d.slot_of_entity != nullptr && e < d.n_entities) world_slot = d.slot_of_entity[e];
			if (sl >= 0 && d.soa.model != nullptr) { // the mirror as a structure of arrays: every field a contiguous load across the wave
				mmb = d.mm_s;
				in.model = d.soa.model[sl];
				in.material_offset = d.soa.material_offset[sl];
				const uint32_t fd = d.soa.flags_dirty[sl];
				in.flags = (uint8_t)fd;
				in.dirty = (uint8_t)(fd >> 8);
				in.pos[0] = d.soa.px[sl]; in.pos[1] = d.soa.py[sl]; in.pos[2] = d.soa.pz[sl];
				lod_at = &d.state_s[sl].lod;
				in.lod = *lod_at;
				in.pose_frame = d.state_s[sl].pose_frame;
			} else {
				if (sl >= 0) {
					rec = d.inst_s + sl;
					mmb = d.mm_s;
				} else if (e < d.n_entities) {
					rec = d.inst + e;
				}
				if (rec != nullptr) {
					in = *rec; // one 64-byte record
					lod_at = &rec->lod;
					if (sl >= 0 && d.state_s != nullptr) {
						lod_at = &d.state_s[sl].lod;
						in.lod = *lod_at;
						in.pose_frame = d.state_s[sl].pose_frame;
					}
				}
			}
			const int32_t mdl = in.model;
			if (mdl >= 0) {
				// The model's LOD table in ONE round trip: the four distances as one 16-byte load and the five index pairs as 8-byte loads, all
				// unconditional and back to back (a record is 64 bytes, hipMalloc'ed table: 16-byte aligned). Read where they are used - the
				// `else if` chain over lod_distances[k], then lod_indices[lod_idx] - they were up to five DEPENDENT loads, each behind its own
				// s_waitcnt vmcnt(0) (seen in the ISA), in a kernel whose time is the length of its chain of dependent loads.
				static_assert(sizeof(LmxKeysModel) == 64 && offsetof(LmxKeysModel, lod_indices) == 16, "the model record is read as 16 + 5 x 8 bytes");
				const LmxKeysModel* mp = d.models + mdl;
				const float4 lod_d = *reinterpret_cast<const float4*>(mp->lod_distances);
				int2 lod_i[5];
#pragma unroll
				for (int k = 0; k < 5; ++k) lod_i[k] = *reinterpret_cast<const int2*>(&mp->lod_indices[k]);
				auto lod_range = [&](uint32_t k) { // lod_indices[k] out of registers (a dynamic index would put the array into scratch)
					int2 r = lod_i[0];
#pragma unroll
					for (uint32_t j = 1; j < 5; ++j) r = k == j ? lod_i[j] : r;
					return r;
				};
				double px = in.pos[0], py = in.pos[1], pz = in.pos[2];
				if (world_slot >= 0) { // World::getTransforms()[e].pos out of the hierarchy's SoA
					px = d.wpx[world_slot]; py = d.wpy[world_slot]; pz = d.wpz[world_slot];
				}
				{
					const double cx = px - kv.cam[0], cy = py - kv.cam[1], cz = pz - kv.cam[2];
					depth_key = float_flip(__float_as_uint((float)(cx * cx + cy * cy + cz * cz))); // :3925-3931
				}
				const double rx = px - kv.ref[0], ry = py - kv.ref[1], rz = pz - kv.ref[2];
				const float squared_length = (float)(rx * rx + ry * ry + rz * rz); // float(squaredLength(pos - lod_ref_point)), math.cpp:397
				const float sd = squared_length * kv.lod_multiplier_rcp;
				uint32_t lod_idx = 4; // Model::getLODMeshIndices, model.h:173-179
				if (sd < lod_d.x) lod_idx = 0;
				else if (sd < lod_d.y) lod_idx = 1;
				else if (sd < lod_d.z) lod_idx = 2;
				else if (sd < lod_d.w) lod_idx = 3;
				if (in.dirty) {
					queue_dirty = true; // queueMaterialOverrideRefresh(e); continue;  (:3879-3882)
				} else {
					mmb += in.material_offset;
					moved = (in.flags & LMX_MODEL_INSTANCE_MOVED) != 0;
					pose_stamp = in.pose_frame;
					float lod = in.lod;
					if (lod != (float)lod_idx) { // :3937-3952
						const float dl = (float)lod_idx - lod;
						const float ad = fabsf(dl);
						if (ad <= kv.time_delta) {
							*lod_at = (float)lod_idx;
							const int2 r = lod_range(lod_idx);
							from0 = r.x; to0 = r.y;
						} else {
							if (!kv.is_shadow) { lod = lod + dl / ad * kv.time_delta; *lod_at = lod; }
							const uint32_t cur = (uint32_t)lod;
							const int2 r = lod_range(cur);
							from0 = r.x; to0 = r.y;
							if (cur < 3) { const int2 r1 = lod_range(cur + 1); from1 = r1.x; to1 = r1.y; }
						}
					} else {
						const int2 r = lod_range(lod_idx);
						from0 = r.x; to0 = r.y;
					}
				}
			}
		}
		const int32_t len0 = to0 >= from0 ? to0 - from0 + 1 : 0, len1 = to1 >= from1 ? to1 - from1 + 1 : 0, total = len0 + len1;
		auto mesh_index = [&](int32_t it) { return it < len0 ? from0 + it : from1 + (it - len0); };
		// what create_key does with one mesh of the range (:3884-3935): 0 nothing, 1 skinned pair, 2 moved pair, 3 instancer record, 4 depth-sorted pair
		auto classify = [&](const LmxMeshMaterial& mm, uint32_t bucket) -> uint32_t {
			if (mm._pad[0] == LMX_MESH_SKINNED) return 1u; // device copy: _pad[0] = Mesh::type of the model's mesh
			if (moved && !kv.is_shadow) return 2u;
			if (bucket < 0xffu) return 3u;
			if (bucket < 0xffffu) return 4u;
			return 0u;
		};
		// ---- ONE walk over the materials: the first KEYS_MM_REGS meshes of the range(s) - all of them unless a model's LODs hold more than
		// three meshes - are fetched with their loads back to back (one memory round trip, not one per mesh) and kept as (sort key | bucket,
		// kind) in registers for the emit below. Rounds 1-3 walked the table twice (count, then emit) and every step waited for its own load.
		uint32_t item[KEYS_MM_REGS]; // sort_key | (u8)bucket << 24
		uint32_t kinds = 0;          // 3 bits per cached mesh
		uint32_t n_pairs = 0, n_recs = 0;
		bool any_skinned = false;
		{
			LmxMeshMaterial raw[KEYS_MM_REGS];
#pragma unroll
			for (int32_t k = 0; k < KEYS_MM_REGS; ++k) raw[k] = mmb[k < total ? mesh_index(k) : 0];
#pragma unroll
			for (int32_t k = 0; k < KEYS_MM_REGS; ++k) {
				const uint32_t bucket = s_bucket[raw[k].layer];
				const uint32_t kind = k < total ? classify(raw[k], bucket) : 0u;
				item[k] = raw[k].sort_key | (bucket << 24);
				kinds |= kind << (3 * k);
				any_skinned |= kind == 1u;
				n_pairs += (kind == 1u || kind == 2u || kind == 4u) ? 1u : 0u;
				n_recs += kind == 3u ? 1u : 0u;
			}
			for (int32_t it = KEYS_MM_REGS; it < total; ++it) { // longer ranges: counted here, read again by the emit
				const LmxMeshMaterial mm = mmb[mesh_index(it)];
				const uint32_t kind = classify(mm, s_bucket[mm.layer]);
				any_skinned |= kind == 1u;
				n_pairs += (kind == 1u || kind == 2u || kind == 4u) ? 1u : 0u;
				n_recs += kind == 3u ? 1u : 0u;
			}
		}
		// Pose::frame stamp (:3889-3898): exactly one visit per frame hands the instance to the pose processor
		bool push_pose = false;
		if (any_skinned && pose_stamp != kv.frame_number) { // (the address is rebuilt here instead of living in two registers since the record was read)
			uint32_t* frame_at = slot >= 0 ? (d.state_s != nullptr ? &d.state_s[slot].pose_frame : &d.inst_s[slot].pose_frame) : &d.inst[e].pose_frame;
			// A plain store: the stamp this lane read at the top of the tile is still the truth - an entity is in the visible list once, and
			// the launches of a context run one after the other on its stream (the reference needs its CAS loop because its views are
			// concurrent jobs). As a returning atomic per skinned instance it was 6.5 of the kernel's 62 us (profiles/r04/keys_probes.txt).
			*frame_at = kv.frame_number;
			push_pose = true;
		}
		// ---- block-wide exclusive prefix of (pairs, recs) and ranks of the two flags; one atomic per pair of lists
		uint32_t incl = n_pairs | (n_recs << 16); // <= 2 * span per lane, <= 64 * that per wave: 16 bits each
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
			if (lane >= (uint32_t)o) incl += up;
		}
		const uint64_t pose_mask = __ballot(push_pose), dirty_mask = __ballot(queue_dirty);
		if (lane == 63) { s_wave[wave][0] = incl; s_wave[wave][1] = (uint32_t)__popcll(pose_mask); s_wave[wave][2] = (uint32_t)__popcll(dirty_mask); }
		__syncthreads();
		// Every thread sums the waves' counts itself (8 LDS words): the tile's totals and this wave's offsets inside the tile need no second
		// barrier, and they are all the LDS-staged emit below needs - it writes at positions RELATIVE to the tile's ranges.
		uint32_t tile_pairs = 0, tile_recs = 0, tile_poses = 0, tile_dirty = 0;
		uint32_t pair_at = (incl & 0xffffu) - n_pairs, rec_at = (incl >> 16) - n_recs; // relative to the tile's ranges (direct stores: absolute from `publish_bases` on)
		uint32_t pose_at = rank_in(pose_mask), dirty_at = rank_in(dirty_mask);
#pragma unroll
		for (uint32_t w = 0; w < KEYS_BLOCK / 64; ++w) {
			const uint32_t c0 = s_wave[w][0], c1 = s_wave[w][1], c2 = s_wave[w][2];
			tile_pairs += c0 & 0xffffu; tile_recs += c0 >> 16; tile_poses += c1; tile_dirty += c2;
			if (w < wave) { pair_at += c0 & 0xffffu; rec_at += c0 >> 16; pose_at += c1; dirty_at += c2; }
		}
		// a tile's (key, value) pairs and instancer records are collected in LDS at their positions inside the tile's output ranges and leave
		// as contiguous stores (8-byte stores at every lane's own run of positions used 26 % of the sectors they touched); tiles with more
		// output than the buffers hold keep the direct stores
		__shared__ uint64_t s_pair_key[KEYS_STAGE_PAIRS], s_pair_value[KEYS_STAGE_PAIRS], s_rec_value[KEYS_STAGE_PAIRS];
		__shared__ uint32_t s_rec_key[KEYS_STAGE_PAIRS], s_rec_rank[KEYS_STAGE_PAIRS];
		const bool stage = tile_pairs <= (uint32_t)KEYS_STAGE_PAIRS && tile_recs <= (uint32_t)KEYS_STAGE_PAIRS; // block-uniform
		// The two reservations - thread 0: {pairs, recs}, thread 1: {poses, dirty}; one 64-bit returning atomic each, on two cache lines - are
		// ISSUED here and their results consumed behind the emit: the ranges' bases are needed when the staged outputs leave LDS, not while
		// they are collected, so the atomics' round trip (same-address atomics retire at ~90 per microsecond chip-wide: a block's turn in that
		// queue is microseconds away) runs under the emit instead of in front of it, and the tile has three barriers instead of five.
		// (Measured and NOT kept, round 4: a ticket per tile + decoupled look-back over per-tile descriptor words, agent-scope loads / stores -
		// 81-88 us against 60-70 us with the atomics, profiles/r04/keys_ab_look_back.txt: the serial chain of cross-XCD hand-offs costs more.)
		unsigned long long reservation = 0;
		if (threadIdx.x < 2) {
			const uint32_t lo = threadIdx.x == 0 ? tile_pairs : tile_poses, hi = threadIdx.x == 0 ? tile_recs : tile_dirty;
			if ((lo | hi) && !(LMX_KEYS_PROBE & 4)) reservation = atomicAdd(reinterpret_cast<unsigned long long*>(d.counters + (threadIdx.x == 0 ? KEYS_N_PAIRS : KEYS_N_POSES)), (unsigned long long)lo | ((unsigned long long)hi << 32));
		}
		auto publish_bases = [&]() { // threads 0 / 1 hand the ranges' bases to the block (the first use of the atomics' results)
			if (threadIdx.x < 2) {
				s_base[2 * threadIdx.x] = (uint32_t)reservation;
				s_base[2 * threadIdx.x + 1] = (uint32_t)(reservation >> 32);
			}
		};
		uint32_t tile_pair0 = 0, tile_rec0 = 0; // staged: the emit works on relative positions
		if (!stage) { // direct stores need the bases now
			publish_bases();
			__syncthreads();
			tile_pair0 = s_base[0]; tile_rec0 = s_base[1];
			pair_at += tile_pair0; rec_at += tile_rec0;
		}
		// ---- emit, in lockstep over the wave so that the group histogram costs one atomic per distinct key and step
		auto emit = [&](int32_t it, uint32_t word, uint32_t kind) {
			bool add_inst = false;
			const uint32_t mesh_sort_key = word & 0xffffffu;
			if (kind != 0u) {
				const int32_t mesh_idx = mesh_index(it);
				const uint64_t bucket_bits = (uint64_t)(word >> 24) << LMX_SORT_KEY_BUCKET_SHIFT; // makeMeshSortKey / makeDepthSortKey take the bucket as u8
				uint64_t key = 0, value = (uint64_t)e | ((uint64_t)(uint32_t)mesh_idx << LMX_SORT_VALUE_MESH_IDX_SHIFT);
				if (kind == 1u) {
					value |= (uint64_t)LMX_DRAW_SKINNED << LMX_SORT_VALUE_TYPE_SHIFT;
					key = (uint64_t)mesh_sort_key | bucket_bits;
				} else if (kind == 2u) {
					value |= (uint64_t)LMX_DRAW_MESH << LMX_SORT_VALUE_TYPE_SHIFT;
					key = (uint64_t)mesh_sort_key | bucket_bits;
				} else if (kind == 4u) { // depth sorted
					value |= (uint64_t)LMX_DRAW_MESH << LMX_SORT_VALUE_TYPE_SHIFT;
					key = (uint64_t)depth_key | bucket_bits;
				} else {
					add_inst = true; // instancer.add(mesh_sort_key, value)
				}
				// the record's first word: its key and where its counter lives - the tile's row (tile ranks) or the block's private copy
				uint32_t rec_word = mesh_sort_key | (copy << 24), rank = 0;
				if (block_ranks && add_inst) {
					rec_word = (mesh_sort_key & (uint32_t)(KEYS_HIST_LDS - 1)) | (blockIdx.x << KEYS_ROW_SHIFT);
					if (mesh_sort_key <= d.max_sort_key) rank = atomicAdd(&s_hist[mesh_sort_key], 1u); // ds_add_rtn_u32: the record's rank among the block's records of its key
					else rank = KEYS_RANK_DEAD; // a key above the range: its 12-bit field would ALIAS a valid key - the scatter drops the record by its rank (the run is flagged KEYS_OVERFLOW = 2 below)
				}
				if (stage) {
					if (!add_inst) { s_pair_key[pair_at] = key; s_pair_value[pair_at] = value; }
					else { s_rec_key[rec_at] = rec_word; s_rec_value[rec_at] = value; if (block_ranks) s_rec_rank[rec_at] = rank; }
				} else if (!add_inst) {
					if (pair_at < d.cap_pairs) { d.keys[pair_at] = key; d.values[pair_at] = value; } else atomicMax(&d.counters[KEYS_OVERFLOW], 1u);
				} else {
					if (rec_at < d.cap_recs) { d.rec_key[rec_at] = rec_word; d.rec_value[rec_at] = value; if (block_ranks) d.rec_rank[rec_at] = rank; } else atomicMax(&d.counters[KEYS_OVERFLOW], 1u);
				}
				// (plain arithmetic: with `++pair_at` / `++rec_at` in the branches the compiler indexed the two cursors in scratch memory)
				pair_at += add_inst ? 0u : 1u;
				rec_at += add_inst ? 1u : 0u;
			}
			const bool in_range = add_inst && mesh_sort_key <= d.max_sort_key;
			if (add_inst && !in_range) atomicMax(&d.counters[KEYS_OVERFLOW], 2u); // a mesh sort key above Renderer::getMaxSortKey(): the reference indexes out of bounds
			// with many private copies the counters are spread thinly enough for one atomic per lane; the per-wave de-duplication
			// (one atomic per distinct key, ~15 scalar + vector instructions per key: 60 % of this kernel's time at 256 live keys)
			// is kept for key ranges too large to privatise
			if (LMX_KEYS_PROBE & 1) { // (timing probe only: no group histogram - the instancer's groups come out wrong)
			} else if (block_ranks) { // (counted above, where the rank was taken)
			} else if (lds_hist) {
				if (in_range) atomicAdd(&s_hist[mesh_sort_key], 1u); // ds_add_u32, nothing returned
			} else if (d.n_copies >= 8) {
				if (in_range) atomicAdd(d.group_count + (size_t)copy * (d.max_sort_key + 1) + mesh_sort_key, 1u);
			} else {
				wave_histogram(in_range, mesh_sort_key, d.group_count + (size_t)copy * (d.max_sort_key + 1));
			}
		};
		// ONE rolled loop: the cached words rotate through item[0] (five register moves per step instead of six copies of the emit code)
		for (int32_t it = 0; __ballot(it < total) != 0; ++it) {
			uint32_t word = item[0], kind = kinds & 7u; // (kind 0 past the lane's own range)
#pragma unroll
			for (int k = 0; k + 1 < KEYS_MM_REGS; ++k) item[k] = item[k + 1];
			kinds >>= 3;
			if (it >= KEYS_MM_REGS) { // a range longer than the cache: read again
				word = 0;
				kind = 0;
				if (it < total) {
					const LmxMeshMaterial mm = mmb[mesh_index(it)];
					const uint32_t bucket = s_bucket[mm.layer];
					kind = classify(mm, bucket);
					word = mm.sort_key | (bucket << 24);
				}
			}
			emit(it, word, kind);
		}
		if (stage) publish_bases();
		__syncthreads(); // the staged outputs and the bases are in LDS
		{
			const uint32_t pose0 = s_base[2], dirty0 = s_base[3];
			if (queue_dirty) { if (dirty0 + dirty_at < d.cap_list) d.dirty_list[dirty0 + dirty_at] = (int32_t)e; else atomicMax(&d.counters[KEYS_OVERFLOW], 1u); }
			if (push_pose) { if (pose0 + pose_at < d.cap_list) d.poses[pose0 + pose_at] = (int32_t)e; else atomicMax(&d.counters[KEYS_OVERFLOW], 1u); }
		}
		if (block_ranks) { // (the histogram runs on: it leaves behind the block's last tile)
		} else if (lds_hist && tile_recs != 0) { // (behind the barrier: every lane's LDS increments are in)
			for (uint32_t k = threadIdx.x; k <= d.max_sort_key; k += KEYS_BLOCK) {
				const uint32_t c = s_hist[k];
				if (c != 0) {
					atomicAdd(d.group_count + (size_t)copy * (d.max_sort_key + 1) + k, c);
					s_hist[k] = 0; // ready for the next tile (the tile's last barrier is ahead)
				}
			}
		}
		if (stage) { // the tile's outputs leave in position order: consecutive lanes, consecutive 8-byte (4-byte) elements
			tile_pair0 = s_base[0]; tile_rec0 = s_base[1];
			for (uint32_t j = threadIdx.x; j < tile_pairs; j += KEYS_BLOCK) {
				const uint32_t at = tile_pair0 + j;
				if (at < d.cap_pairs) { d.keys[at] = s_pair_key[j]; d.values[at] = s_pair_value[j]; } else atomicMax(&d.counters[KEYS_OVERFLOW], 1u);
			}
			for (uint32_t j = threadIdx.x; j < tile_recs; j += KEYS_BLOCK) {
				const uint32_t at = tile_rec0 + j;
				if (at < d.cap_recs) { d.rec_key[at] = s_rec_key[j]; d.rec_value[at] = s_rec_value[j]; if (block_ranks) d.rec_rank[at] = s_rec_rank[j]; } else atomicMax(&d.counters[KEYS_OVERFLOW], 1u);
			}
		}
		__syncthreads(); // s_wave, s_base and the staging buffers are rewritten by the next tile
	}
	if (block_ranks) { // the block's histogram becomes its ROW of the table ...
		// ... and (key ranges up to KEYS_SCATTER_OFFSETS) its entries are the records of key k in the blocks that got there before: what a returning add on the key's counter hands
		// back (one counter per 128-byte line: a block's adds go out together, 512 blocks x the keys they saw, at the very end of the
		// kernel - k_keys_mesh 44.2 -> 44.3 us; the column scan this replaces, k_keys_reduce_rows, was a launch of 5.4 us between two gaps:
		// chain 65.3 -> 61.5 us, profiles/r05/keys_row_prefix_by_atomics.txt). The counters end up as the groups' sizes.
		uint32_t* row = d.block_rows + (size_t)blockIdx.x * (d.max_sort_key + 1);
		if (d.total_pad != nullptr) {
			for (uint32_t k = threadIdx.x; k <= d.max_sort_key; k += KEYS_BLOCK) {
				const uint32_t c = s_hist[k];
				row[k] = c != 0 ? atomicAdd(d.total_pad + (size_t)k * KEYS_PAD_WORDS, c) : 0u;
			}
		} else { // larger key ranges: the counts themselves (zeros included), k_keys_reduce_rows scans the columns
			for (uint32_t k = threadIdx.x; k <= d.max_sort_key; k += KEYS_BLOCK) row[k] = s_hist[k];
		}
	}
}

__global__ __launch_bounds__(256) void k_keys_decal(KeysDevice d, const KeysViewDevice kv, const KeysShardList L, const uint32_t* __restrict__ sort_key,
	const uint8_t* __restrict__ layer, uint32_t draw_type) {
	__shared__ uint32_t s_sh_cnt[KEYS_MAX_SHARDS], s_sh_win[KEYS_MAX_SHARDS], s_sh_first[KEYS_MAX_SHARDS + 1]; // s_sh_first[s]: ids in the shards before s
	shard_list_to_lds(L, s_sh_cnt, s_sh_win, s_sh_first, [](uint32_t c) { return c; });
	const uint32_t ns = L.n < (uint32_t)KEYS_MAX_SHARDS ? L.n : (uint32_t)KEYS_MAX_SHARDS;
	const uint32_t n = s_sh_first[ns];
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i - (i & 63u) < n; i += gridDim.x * 256) {
		bool push = false;
		uint64_t key = 0, value = 0;
		if (i < n) {
			uint32_t lo = 0, hi = ns; // the shard that holds list position i: the last s with s_sh_first[s] <= i (empty shards share their start with the next one)
			while (hi - lo > 1) {
				const uint32_t mid = (lo + hi) >> 1;
				if (s_sh_first[mid] <= i) lo = mid; else hi = mid;
			}
			const uint32_t e = (uint32_t)L.ids[s_sh_win[lo] + (i - s_sh_first[lo])];
			if (e < d.n_entities) {
				const uint8_t bucket = (uint8_t)kv.bucket_map[layer[e]]; // const u8 bucket = bucket_map[layer], :3845
				if (bucket < 0xff) {
					key = (uint64_t)sort_key[e] | ((uint64_t)bucket << LMX_SORT_KEY_BUCKET_SHIFT); // makeDecalSortKey, :83-89
					value = (uint64_t)e | ((uint64_t)draw_type << LMX_SORT_VALUE_TYPE_SHIFT);      // make(Curve)DecalSortValue, :125-131
					push = true;
				}
			}
		}
		const uint32_t idx = wave_append(push, d.counters + KEYS_N_PAIRS);
		if (push) { if (idx < d.cap_pairs) { d.keys[idx] = key; d.values[idx] = value; } else atomicMax(&d.counters[KEYS_OVERFLOW], 1u); }
	}
}

// one wave per key: lane c holds the counts of copies c, c + 64, ... in turn; total[k] = their sum, group_base[c][k] = copy c's base
// inside group k (exclusive prefix over the copies). The histogram itself is zeroed - it is the scatter's cursor table from here on -
// and so are the other table (the previous run's cursors: the next run's histogram) and the next run's list counters.
__global__ __launch_bounds__(256) void k_keys_reduce_copies(KeysDevice d) {
	const uint32_t n = d.max_sort_key + 1;
	const uint32_t k = blockIdx.x * 4 + (threadIdx.x >> 6);
	const uint32_t lane = threadIdx.x & 63u;
	if (blockIdx.x == 0 && threadIdx.x < (uint32_t)KEYS_COUNTERS) d.counters_next[threadIdx.x] = 0;
	if (k >= n) return;
	uint32_t carry = 0; // wave-uniform: the copies before this round's
	for (uint32_t c0 = 0; c0 < d.n_copies; c0 += 64u) {
		const uint32_t c = c0 + lane;
		const size_t at = (size_t)c * n + k;
		const uint32_t v = c < d.n_copies ? d.group_count[at] : 0;
		uint32_t incl = v;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
			if (lane >= (uint32_t)o) incl += up;
		}
		if (c < d.n_copies) {
			d.group_base[at] = carry + incl - v;
			d.group_count[at] = 0;
			d.group_count_next[at] = 0;
		}
		carry += (uint32_t)__shfl((int)incl, 63);
	}
	if (lane == 63) d.group_total[k] = carry;
}

// Block ranks, key ranges beyond KEYS_SCATTER_OFFSETS (up to there the blocks of k_keys_mesh fetch their rows' entries with returning adds;
// with thousands of live keys per block that is a million atomics at the kernel's end - 4096 random keys: chain 84 us with this kernel,
// 94 us with the adds): the columns of the rows' table become exclusive prefixes (block_rows[b][k] = the records of key k in the rows
// before b) and total[k] their sums. A block owns 8 adjacent keys; a wave-wide access covers 8 rows x those 8 keys (lane = 8 * stripe + key: eight
// 32-byte row segments; adjacent lanes on adjacent keys - with adjacent lanes on adjacent ROWS the same kernel was a third slower, with a
// lane per row of ONE key twice); the 16 waves split the rows into contiguous ranges, sum theirs (loads independent, 8 in flight), meet
// once in LDS, and a second walk over the (cached) range writes the prefixes.
constexpr int KEYS_RR_WAVES = 16;
__global__ __launch_bounds__(KEYS_RR_WAVES * 64) void k_keys_reduce_rows(KeysDevice d) {
	__shared__ uint32_t s_tot[KEYS_RR_WAVES][8];
	const uint32_t n = d.max_sort_key + 1;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t ks = lane & 7u, stripe = lane >> 3;
	const uint32_t k = blockIdx.x * 8u + ks;
	if (blockIdx.x == 0 && threadIdx.x < (uint32_t)KEYS_COUNTERS) d.counters_next[threadIdx.x] = 0;
	const uint32_t n_rows = d.n_rows;
	const uint32_t per_wave = ((n_rows + KEYS_RR_WAVES * 8u - 1u) / (KEYS_RR_WAVES * 8u)) * 8u; // a multiple of the 8 stripes
	const uint32_t r0 = min(wave * per_wave, n_rows), r1 = min(r0 + per_wave, n_rows);
	const bool live = k < n;
	uint32_t* col = d.block_rows + (live ? k : 0u);
	uint32_t sum = 0; // of this lane's stripe of the wave's range
	for (uint32_t r = r0 + stripe; r < r1; r += 64u) {
		uint32_t v[8];
#pragma unroll
		for (uint32_t j = 0; j < 8; ++j) v[j] = live && r + 8u * j < r1 ? col[(size_t)(r + 8u * j) * n] : 0u;
#pragma unroll
		for (uint32_t j = 0; j < 8; ++j) sum += v[j];
	}
	// the wave's total of key ks: over the 8 stripes (lanes ks, ks + 8, ...)
	uint32_t wave_total = sum;
	wave_total += (uint32_t)__shfl_xor((int)wave_total, 8);
	wave_total += (uint32_t)__shfl_xor((int)wave_total, 16);
	wave_total += (uint32_t)__shfl_xor((int)wave_total, 32);
	if (stripe == 0) s_tot[wave][ks] = wave_total;
	__syncthreads();
	uint32_t run = 0, total = 0; // the rows before this wave's range; all rows
#pragma unroll
	for (uint32_t w = 0; w < (uint32_t)KEYS_RR_WAVES; ++w) {
		const uint32_t c = s_tot[w][ks];
		total += c;
		if (w < wave) run += c;
	}
	if (wave == 0 && stripe == 0 && live) d.group_total[k] = total;
	for (uint32_t r = r0; r < r1; r += 64u) { // (wave-uniform bounds: the shuffles below see every lane)
		uint32_t v[8];
#pragma unroll
		for (uint32_t j = 0; j < 8; ++j) v[j] = live && r + 8u * j + stripe < r1 ? col[(size_t)(r + 8u * j + stripe) * n] : 0u;
#pragma unroll
		for (uint32_t j = 0; j < 8; ++j) { // step j: the 8 rows r + 8 j .. r + 8 j + 7, one per stripe
			uint32_t incl = v[j]; // inclusive over the stripes (lanes 8 apart hold the same key)
			uint32_t up = (uint32_t)__shfl_up((int)incl, 8);
			if (stripe >= 1u) incl += up;
			up = (uint32_t)__shfl_up((int)incl, 16);
			if (stripe >= 2u) incl += up;
			up = (uint32_t)__shfl_up((int)incl, 32);
			if (stripe >= 4u) incl += up;
			if (live && r + 8u * j + stripe < r1) col[(size_t)(r + 8u * j + stripe) * n] = run + incl - v[j];
			run += (uint32_t)__shfl((int)incl, (int)(56u + ks));
		}
	}
}

// one block: offsets[k] = sum of total[0..k), offsets[n] = grand total; non-empty groups counted
__global__ __launch_bounds__(1024) void k_keys_offsets(KeysDevice d) {
	__shared__ uint32_t s_wave[16];
	__shared__ uint32_t s_carry;
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const uint32_t n = d.max_sort_key + 1;
	if (tid == 0) s_carry = 0;
	__syncthreads();
	uint32_t non_empty = 0;
	for (uint32_t base = 0; base < n; base += 1024) {
		const uint32_t k = base + tid;
		const uint32_t c = k < n ? d.group_total[k] : 0;
		non_empty += c != 0;
		uint32_t incl = c; // inclusive scan inside the wave
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
			if (lane >= (uint32_t)o) incl += up;
		}
		if (lane == 63) s_wave[wave] = incl;
		__syncthreads();
		uint32_t before = s_carry;
		for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
		if (k < n) d.group_offset[k] = before + incl - c;
		__syncthreads();
		if (tid == 1023) s_carry = before + incl;
		__syncthreads();
	}
	if (tid == 0) d.group_offset[n] = s_carry;
	uint32_t total = non_empty;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) total += (uint32_t)__shfl_down((int)total, o);
	if (lane == 0 && total) atomicAdd(d.counters + KEYS_N_GROUPS, total);
}

// Instancer records -> CSR values. The lane whose record lands on its group's FIRST position also pushes the group's AUTOINSTANCED
// pair (:3958-3968, instancer index 0: `instances[i].begin->renderables[0]` is any member - they share the material): every
// non-empty group has exactly one such record, so the pairs need no launch of their own (k_keys_groups, rounds 1-4).
// OWN_OFFSETS (key ranges up to KEYS_SCATTER_OFFSETS): every block forms the exclusive scan of the group sizes itself, in LDS - 1 to 4
// loads per thread and one block-wide scan, issued next to the tile's first loads - and block 0 also writes it out (group_offset, the
// number of non-empty groups): k_keys_offsets, a single-block launch of ~5 us between two launch gaps, leaves the chain.
template <bool OWN_OFFSETS> __global__ __launch_bounds__(256) void k_keys_scatter(KeysDevice d, const KeysViewDevice kv) {
	__shared__ uint32_t s_off[OWN_OFFSETS ? KEYS_SCATTER_OFFSETS : 1];
	__shared__ uint32_t s_wave_sum[4];
	// A block has about one tile and a tile is a chain of dependent round trips (the ISA waited for every load where it was issued:
	// record key -> cursor atomic -> base -> record value -> store, behind the number of records): the tile's two record loads are issued
	// FIRST, bounded by the capacity instead of the count, next to the count's and the offsets' loads; base and cursor go out together.
	const bool block_ranks = d.block_rows != nullptr; // launch-uniform
	uint32_t packed_next = 0, rank_next = 0;
	uint64_t value_next = 0;
	{
		const uint32_t i0 = blockIdx.x * 256u + threadIdx.x;
		if (i0 < d.cap_recs) { packed_next = d.rec_key[i0]; value_next = d.rec_value[i0]; if (block_ranks) rank_next = d.rec_rank[i0]; }
	}
	const uint32_t n = min(d.counters[KEYS_N_RECS], d.cap_recs);
	const uint32_t stride = d.max_sort_key + 1;
	if (OWN_OFFSETS) {
		if (blockIdx.x != 0 && blockIdx.x * 256u >= n) return; // (block 0 stays: it writes the offsets out)
		const uint32_t k0 = threadIdx.x * 4u, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
		uint32_t c[4], sum = 0, non_empty = 0;
#pragma unroll
		for (uint32_t j = 0; j < 4; ++j) {
			c[j] = k0 + j < stride ? (d.total_pad != nullptr ? d.total_pad[(size_t)(k0 + j) * KEYS_PAD_WORDS] : d.group_total[k0 + j]) : 0u;
			sum += c[j]; non_empty += c[j] != 0u;
		}
		if (d.total_pad != nullptr && blockIdx.x == 0) { // block ranks: no k_keys_reduce_copies ran - the next run's counters are zeroed here
			if (threadIdx.x < (uint32_t)KEYS_COUNTERS) d.counters_next[threadIdx.x] = 0;
			for (uint32_t k = threadIdx.x; k < stride; k += 256u) d.total_pad_next[(size_t)k * KEYS_PAD_WORDS] = 0;
		}
		uint32_t incl = sum;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
			if (lane >= (uint32_t)o) incl += up;
		}
		if (lane == 63) s_wave_sum[wave] = incl;
		__syncthreads();
		uint32_t run = incl - sum;
		for (uint32_t w = 0; w < wave; ++w) run += s_wave_sum[w];
#pragma unroll
		for (uint32_t j = 0; j < 4; ++j) {
			if (k0 + j < stride) { s_off[k0 + j] = run; if (blockIdx.x == 0) d.group_offset[k0 + j] = run; }
			run += c[j];
		}
		if (blockIdx.x == 0) {
			if (threadIdx.x == 255) d.group_offset[stride] = run; // (thread 255 holds the grand total: keys past the range count 0)
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) non_empty += (uint32_t)__shfl_down((int)non_empty, o);
			if (lane == 0 && non_empty) atomicAdd(d.counters + KEYS_N_GROUPS, non_empty);
		}
		__syncthreads();
	}
	auto offset_of = [&](uint32_t key) { return OWN_OFFSETS ? s_off[key] : d.group_offset[key]; };
	for (uint32_t tile = blockIdx.x * 256; tile < n; tile += gridDim.x * 256) {
		const uint32_t i = tile + threadIdx.x;
		const uint32_t packed = i < n ? packed_next : 0; // mesh sort key | copy << 24, or (block ranks) | row << 12
		const uint64_t renderable = value_next;
		const uint32_t rank_next_now = rank_next;
		{
			const uint32_t i1 = i + gridDim.x * 256u;
			if (i1 < n) { packed_next = d.rec_key[i1]; value_next = d.rec_value[i1]; if (block_ranks) rank_next = d.rec_rank[i1]; }
		}
		const uint32_t key = block_ranks ? packed & (uint32_t)(KEYS_HIST_LDS - 1) : packed & 0xffffffu;
		const bool has = i < n && key <= d.max_sort_key && !(block_ranks && rank_next_now == KEYS_RANK_DEAD);
		const size_t at = block_ranks ? (size_t)(packed >> KEYS_ROW_SHIFT) * stride + key : (size_t)(packed >> 24) * stride + key;
		uint32_t in_group = 0; // the record's position inside its group
		if (block_ranks) { // no atomics: the records of key k in the rows before this record's + its rank inside its row
			const uint32_t rank = rank_next_now;
			if (has) in_group = d.block_rows[at] + rank;
		} else if (d.n_copies >= 8) { // privatised cursors: one returning atomic per lane, all in flight together
			if (has) {
				const uint32_t base = d.group_base[at];
#if LMX_KEYS_PROBE & 8 // (timing probe: no cursor atomics - the groups come out wrong)
				in_group = base + (i & 3u);
#else
				in_group = base + atomicAdd(d.group_count + at, 1u);
#endif
			}
		} else {
			// per distinct (copy, key) of the wave: its first lane (leader), the number of lanes holding it and every lane's rank among
			// them - ALU only; then ALL leaders reserve their cursor ranges at once (one memory round trip per wave, not one per key)
			uint64_t todo = __ballot(has);
			uint32_t leader_of = 0, rank = 0, count = 0;
			while (todo) {
				const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1u;
				const uint32_t k = (uint32_t)__shfl((int)packed, (int)leader);
				const uint64_t same = __ballot(has && packed == k) & todo;
				if ((same >> lane_id()) & 1ull) { leader_of = leader; rank = rank_in(same); count = (uint32_t)__popcll(same); }
				todo &= ~same;
			}
			uint32_t base = 0;
			if (has && leader_of == lane_id()) base = atomicAdd(d.group_count + at, count);
			base = (uint32_t)__shfl((int)base, (int)leader_of);
			if (has) in_group = d.group_base[at] + base + rank;
		}
		bool push = false;
		uint64_t pair_key = 0, pair_value = 0;
		if (has) {
			if (!(LMX_KEYS_PROBE & 16)) d.group_values[offset_of(key) + in_group] = renderable; // (16: timing probe, no scatter stores)
			if (in_group == 0) {
				const uint32_t entity_index = (uint32_t)(renderable & 0xffFFffull);
				const uint32_t mesh_idx = (uint32_t)(renderable >> LMX_SORT_VALUE_MESH_IDX_SHIFT);
				const uint8_t layer = d.mesh_materials[d.inst[entity_index].material_offset + mesh_idx].layer;
				const uint8_t bucket = kv.layer_to_bucket[layer];
				pair_value = (uint64_t)key | ((uint64_t)LMX_DRAW_AUTOINSTANCED << LMX_SORT_VALUE_TYPE_SHIFT);               // makeAutoInstancedSortValue(i, 0)
				pair_key = (uint64_t)key | LMX_SORT_KEY_INSTANCED_FLAG | ((uint64_t)bucket << LMX_SORT_KEY_BUCKET_SHIFT);  // makeAutoInstancedSortKey(i, bucket)
				push = true;
			}
		}
		const uint32_t idx = wave_append(push, d.counters + KEYS_N_PAIRS);
		if (push) { if (idx < d.cap_pairs) { d.keys[idx] = pair_key; d.values[idx] = pair_value; } else atomicMax(&d.counters[KEYS_OVERFLOW], 1u); }
	}
}

// ---- slot-ordered mirror of the instance tables ------------------------------------------------------------------------------------
// The cull emits visible ids in the order of the sorted set; entity indices are unrelated to it, so the entity-indexed 64-byte records
// (and the per-instance material spans) were fetched one random 128-byte line each: 285 B fetched + 126 B written per visible entity.
// The mirror holds the same records by static slot, and the material spans packed in slot order; lod / Pose::frame of an entity of the
// sorted set live in its slot record and go back to the entity-indexed record whenever the slot dies (tombstone, re-sort).
__global__ __launch_bounds__(256) void k_keys_mirror_count(const int32_t* __restrict__ slot_ids, uint32_t n_slots, const KeysInstance* __restrict__ inst,
	uint32_t n_entities, const LmxKeysModel* __restrict__ models, uint32_t* __restrict__ count /* [n_slots + 1] */) {
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s > n_slots) return;
	uint32_t c = 0;
	if (s < n_slots) {
		const int32_t e = slot_ids[s];
		if (e >= 0 && (uint32_t)e < n_entities) {
			const int32_t m = inst[e].model;
			if (m >= 0) c = models[m].mesh_count;
		}
	}
	count[s] = c;
}

__global__ __launch_bounds__(256) void k_keys_mirror_fill(const int32_t* __restrict__ slot_ids, uint32_t n_slots, const KeysInstance* __restrict__ inst,
	uint32_t n_entities, const LmxKeysModel* __restrict__ models, const LmxMeshMaterial* __restrict__ mesh_materials, const uint32_t* __restrict__ offset,
	KeysInstance* __restrict__ inst_s, KeysSoA soa, LmxMeshMaterial* __restrict__ mm_s, KeysSlotState* __restrict__ state_s) {
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s >= n_slots) return;
	KeysInstance r;
	memset(&r, 0, sizeof(r));
	r.model = -1;
	const int32_t e = slot_ids[s];
	if (e >= 0 && (uint32_t)e < n_entities) {
		r = inst[e];
		if (r.model >= 0) {
			const uint32_t n = models[r.model].mesh_count, from = r.material_offset, to = offset[s];
			for (uint32_t k = 0; k < n; ++k) mm_s[to + k] = mesh_materials[from + k];
			r.material_offset = to;
		}
	}
	if (soa.model != nullptr) {
		soa.model[s] = r.model;
		soa.material_offset[s] = r.material_offset;
		soa.flags_dirty[s] = (uint16_t)(r.flags | (r.dirty << 8));
		soa.px[s] = r.pos[0]; soa.py[s] = r.pos[1]; soa.pz[s] = r.pos[2];
	} else {
		inst_s[s] = r;
	}
	if (state_s != nullptr) state_s[s] = KeysSlotState{r.lod, r.pose_frame};
}

// new positions of the entity-indexed records into the mirror (lmx_keys_set_positions)
__global__ __launch_bounds__(256) void k_keys_mirror_positions(const int32_t* __restrict__ slot_ids, uint32_t n_slots, const KeysInstance* __restrict__ inst, uint32_t n_entities,
	KeysInstance* __restrict__ inst_s, KeysSoA soa) {
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s >= n_slots) return;
	const int32_t e = slot_ids[s];
	if (e < 0 || (uint32_t)e >= n_entities) return;
	const double x = inst[e].pos[0], y = inst[e].pos[1], z = inst[e].pos[2];
	if (soa.model != nullptr) { soa.px[s] = x; soa.py[s] = y; soa.pz[s] = z; }
	else { inst_s[s].pos[0] = x; inst_s[s].pos[1] = y; inst_s[s].pos[2] = z; }
}

// lod / Pose::frame of the entities of slots [0, n_slots) (or of the slots the id patches are about to turn into tombstones) back
// into the entity-indexed records
__device__ __forceinline__ void mirror_hand_back(uint32_t s, const int32_t* slot_ids, const KeysInstance* inst_s, const int32_t* model_s, const KeysSlotState* state_s,
	KeysInstance* inst, uint32_t n_entities) {
	const int32_t e = slot_ids[s];
	if (e < 0 || (uint32_t)e >= n_entities) return;
	if ((model_s != nullptr ? model_s[s] : inst_s[s].model) < 0) return;
	inst[e].lod = state_s != nullptr ? state_s[s].lod : inst_s[s].lod;
	inst[e].pose_frame = state_s != nullptr ? state_s[s].pose_frame : inst_s[s].pose_frame;
}
__global__ __launch_bounds__(256) void k_keys_mirror_sync(const int32_t* __restrict__ slot_ids, uint32_t n_slots, const KeysInstance* __restrict__ inst_s,
	const int32_t* __restrict__ model_s, const KeysSlotState* __restrict__ state_s, KeysInstance* __restrict__ inst, uint32_t n_entities) {
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s < n_slots) mirror_hand_back(s, slot_ids, inst_s, model_s, state_s, inst, n_entities);
}
__global__ __launch_bounds__(256) void k_keys_mirror_carry(const PatchId* __restrict__ patches, uint32_t n, const int32_t* __restrict__ slot_ids, uint32_t n_slots,
	const KeysInstance* __restrict__ inst_s, const int32_t* __restrict__ model_s, const KeysSlotState* __restrict__ state_s, KeysInstance* __restrict__ inst, uint32_t n_entities) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const PatchId p = patches[i];
	if (p.id < 0 && p.slot < n_slots) mirror_hand_back(p.slot, slot_ids, inst_s, model_s, state_s, inst, n_entities);
}

} // namespace

hipError_t launch_keys_mirror_count(hipStream_t s, const int32_t* slot_ids, uint32_t n_slots, const KeysInstance* inst, uint32_t n_entities, const LmxKeysModel* models, uint32_t* count) {
	hipLaunchKernelGGL(k_keys_mirror_count, dim3((n_slots + 1 + 255u) / 256u), dim3(256), 0, s, slot_ids, n_slots, inst, n_entities, models, count);
	return hipGetLastError();
}
hipError_t launch_keys_mirror_fill(hipStream_t s, const int32_t* slot_ids, uint32_t n_slots, const KeysInstance* inst, uint32_t n_entities, const LmxKeysModel* models,
	const LmxMeshMaterial* mesh_materials, const uint32_t* offset, KeysInstance* inst_s, const KeysSoA& soa, LmxMeshMaterial* mm_s, KeysSlotState* state_s) {
	if (!n_slots) return hipSuccess;
	hipLaunchKernelGGL(k_keys_mirror_fill, dim3((n_slots + 255u) / 256u), dim3(256), 0, s, slot_ids, n_slots, inst, n_entities, models, mesh_materials, offset, inst_s, soa, mm_s,
		state_s);
	return hipGetLastError();
}
hipError_t launch_keys_mirror_positions(hipStream_t s, const int32_t* slot_ids, uint32_t n_slots, const KeysInstance* inst, uint32_t n_entities, KeysInstance* inst_s, const KeysSoA& soa) {
	if (!n_slots) return hipSuccess;
	hipLaunchKernelGGL(k_keys_mirror_positions, dim3((n_slots + 255u) / 256u), dim3(256), 0, s, slot_ids, n_slots, inst, n_entities, inst_s, soa);
	return hipGetLastError();
}
hipError_t launch_keys_mirror_sync(hipStream_t s, const int32_t* slot_ids, uint32_t n_slots, const KeysInstance* inst_s, const int32_t* model_s, const KeysSlotState* state_s,
	KeysInstance* inst, uint32_t n_entities) {
	if (!n_slots) return hipSuccess;
	hipLaunchKernelGGL(k_keys_mirror_sync, dim3((n_slots + 255u) / 256u), dim3(256), 0, s, slot_ids, n_slots, inst_s, model_s, state_s, inst, n_entities);
	return hipGetLastError();
}
hipError_t launch_keys_mirror_carry(hipStream_t s, const PatchId* patches, uint32_t n, const int32_t* slot_ids, uint32_t n_slots, const KeysInstance* inst_s, const int32_t* model_s,
	const KeysSlotState* state_s, KeysInstance* inst, uint32_t n_entities) {
	if (!n) return hipSuccess;
	hipLaunchKernelGGL(k_keys_mirror_carry, dim3((n + 255u) / 256u), dim3(256), 0, s, patches, n, slot_ids, n_slots, inst_s, model_s, state_s, inst, n_entities);
	return hipGetLastError();
}

#ifndef LMX_KEYS_MESH_GRID
#define LMX_KEYS_MESH_GRID 512 // k_keys_mesh: as many blocks as are resident (2 per CU: LDS), each walking ~4 tiles of the headline view - the next tile's first loads run under the current tile and a block's set-up is paid once. k_keys_mesh per 1.05 M visible (profiles/r05/keys_ab_grid.txt): 512 blocks 44.1 us, 768 49.4, 1024 45.2, 1536 46.1, 2048 48.5
#endif
uint32_t keys_mesh_grid_cap() { return (uint32_t)LMX_KEYS_MESH_GRID; }

hipError_t launch_keys(hipStream_t s, const KeysDevice& d_in, const KeysViewDevice& view, const KeysShardList& meshes, const KeysShardList& decals, const KeysShardList& curves) {
	KeysDevice d = d_in;
	d.n_rows = 0;
	const uint32_t grid_cap = 256 * 8; // fixed-size grids walk the lists in tiles: the counts live on the device
	if (meshes.n > (uint32_t)KEYS_MAX_SHARDS || decals.n > (uint32_t)KEYS_MAX_SHARDS || curves.n > (uint32_t)KEYS_MAX_SHARDS) return hipErrorInvalidValue;
	if (meshes.cap && d.inst != nullptr) {
		const uint32_t grid = std::min((meshes.cap + KEYS_BLOCK - 1) / KEYS_BLOCK + meshes.n, (uint32_t)LMX_KEYS_MESH_GRID);
		if (d.block_rows != nullptr && grid > d.cap_rows) return hipErrorInvalidValue;
		d.n_rows = grid; // block ranks: one row of the table per block of this launch
		hipLaunchKernelGGL(k_keys_mesh, dim3(grid), dim3(KEYS_BLOCK), 0, s, d, view, meshes);
	}
	if (decals.cap && d.decal_sort_key != nullptr)
		hipLaunchKernelGGL(k_keys_decal, dim3(std::min((decals.cap + 255) / 256, grid_cap)), dim3(256), 0, s, d, view, decals, d.decal_sort_key, d.decal_layer, (uint32_t)LMX_DRAW_DECAL);
	if (curves.cap && d.curve_sort_key != nullptr)
		hipLaunchKernelGGL(k_keys_decal, dim3(std::min((curves.cap + 255) / 256, grid_cap)), dim3(256), 0, s, d, view, curves, d.curve_sort_key, d.curve_layer, (uint32_t)LMX_DRAW_CURVE_DECAL);
	if (d.block_rows == nullptr) hipLaunchKernelGGL(k_keys_reduce_copies, dim3((d.max_sort_key + 4) / 4), dim3(256), 0, s, d);
	else if (d.total_pad == nullptr) hipLaunchKernelGGL(k_keys_reduce_rows, dim3((d.max_sort_key + 8) / 8), dim3(KEYS_RR_WAVES * 64), 0, s, d); // (with the padded counters the rows' entries came out of the key kernel's own adds)
#ifndef LMX_KEYS_SCATTER_GRID
#define LMX_KEYS_SCATTER_GRID 2048 // the capacity of the record list is a multiple of the list (every mesh of two LODs of every entity of the type): with 8192 blocks most of them only fetched the first tile's records - issued before the count is known - to find nothing to do. Span of the chain (profiles/r05/keys_scatter_grid.txt): 8192 blocks 68.5 us, 4096 65.8, 2048 65.3, 1024 65.8
#endif
	const dim3 scatter_grid(std::max(1u, std::min((d.cap_recs + 255) / 256, (uint32_t)LMX_KEYS_SCATTER_GRID)));
	if (d.max_sort_key < (uint32_t)KEYS_SCATTER_OFFSETS) {
		hipLaunchKernelGGL(k_keys_scatter<true>, scatter_grid, dim3(256), 0, s, d, view);
	} else {
		hipLaunchKernelGGL(k_keys_offsets, dim3(1), dim3(1024), 0, s, d);
		if (d.cap_recs) hipLaunchKernelGGL(k_keys_scatter<false>, scatter_grid, dim3(256), 0, s, d, view);
	}
	return hipGetLastError();
}

} // namespace lmx
