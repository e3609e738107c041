// lmx_capi_exchange.hip — multi-GPU exchange of visible-entity lists behind the C ABI (include/lumix_mi355.h, "exchange").
//
// One process per GPU; every rank owns a disjoint set of entities (visibility depends only on the frustum and the
// entity's own cell, so the cull itself needs no communication). The only exchange step of a frame is ONE ncclAllGather (RCCL over
// xGMI) of a fixed-size record per rank: [8 per-type counts | cap visible ids, types packed back to back]. The cull's raw result is
// gathered straight into the send buffer by the finalize / consolidate kernels (no staging copy), the collective runs on a side
// stream ordered by events, and frames are double-buffered so that the next cull overlaps the previous frame's gather. No torch,
// no host wait in the steady state. RCCL is loaded with dlopen at the first lmx_exchange_* call: single-GPU users of the library
// do not need it.
//
// Three ways to run the step (LmxExchange::mode; same records, same results - tests/test_gpu_exchange.py runs all of them):
//   INLINE   the all-gather on the cull stream behind the pack kernel: four API calls per step, no overlap with the next cull
//   SIDE     the all-gather on a side stream between two events: ~16 us more host work per step, the next cull overlaps the gather
//   P2P      no collective: every rank STORES the used part of its record (counts + the ids it has, not the fixed-size slot) into each
//            peer's receive buffer through hipIpc mappings and raises a sequence flag there; consumers wait for the flags on the
//            device with a BOUNDED spin. Opt-in (LMX_EXCHANGE_MODE=p2p): no multi-GPU box has run it, RCCL stays the default.
// LMX_EXCHANGE_MODE=auto (the default when nothing is forced) times K gathers OF THE RECORD A FRAME SHAPE ACTUALLY SHIPS - at the first
// lmx_exchange_cull_many with that number of frusta, and again when the shape's record has grown or shrunk by 2x since - and takes SIDE
// when one gather is longer than the host work SIDE adds (LMX_EXCHANGE_OVERLAP_US, default 16), INLINE otherwise: with one rank the gather
// is a local copy of a few microseconds, over xGMI with eight it is not, and an 8-cascade record is not the 1-frustum record.
//
// Record of a frame of n frusta: n sub-records [8 counts | cap[f] ids] back to back, cap[f] PER FRUSTUM (round 6; one capacity split
// equally made config 5's cascades - 535 ... 1.15 M visible ids - ship 46 MB per rank for 9.9 MB of ids). Every rank must use the same
// capacities in the same frame (an all-gather has one count): they are a pure function of what every rank has gathered -
//   caps of frame j = policy(caps of frame j - 1, per-frustum maxima over ALL ranks of frame j - 2's gathered counts)
// - so no extra collective is needed: a one-wave kernel behind frame j - 2's gather leaves the maxima in pinned memory, frame j (the next
// user of that slot) waits for that gather on the host (long finished in a pipelined loop) and reads them. A sub-record that overflowed
// is flagged (lmx_exchange_stats.overflow_mask; readers see sum(counts) > cap as before) and regrown the next time its slot comes round.
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "lmx_context.h"

using namespace lmx;

namespace {

// the handful of RCCL entry points used (rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220, ncclCommDestroy :260, ncclAllGather :678)
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
struct Rccl {
	void* lib = nullptr;
	int (*GetUniqueId)(NcclUniqueId*) = nullptr;
	int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
	int (*CommDestroy)(NcclComm) = nullptr;
	int (*AllGather)(const void*, void*, size_t, int /* ncclDataType_t */, NcclComm, hipStream_t) = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
	const char* error = nullptr;
};
constexpr int NCCL_INT32 = 2; // ncclInt32 (rccl.h ncclDataType_t)

Rccl& rccl() {
	static Rccl r = [] {
		Rccl x;
		// LMX_RCCL_LIBRARY: an explicit library instead of the system's RCCL (a site's own build; the several-ranks-on-one-GPU test double
		// of tests/cpp/loopback_rccl.cpp). Loaded RTLD_LOCAL: its nccl* symbols must not shadow a real RCCL in the process.
		if (const char* path = getenv("LMX_RCCL_LIBRARY")) {
			x.lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
			if (!x.lib) {
				x.error = "LMX_RCCL_LIBRARY could not be loaded";
				return x;
			}
		}
		// A process must not end up with two RCCL runtimes (e.g. the copy PyTorch bundles and the system one): first take whatever
		// is already loaded, only then load one.
		for (const char* name : {"librccl.so.1", "librccl.so"}) {
			if (x.lib) break;
			x.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
		}
		for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			if (x.lib) break;
			x.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
		}
		if (!x.lib) {
			x.error = "librccl.so.1 not found";
			return x;
		}
		x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(x.lib, "ncclGetUniqueId");
		x.CommInitRank = (decltype(x.CommInitRank))dlsym(x.lib, "ncclCommInitRank");
		x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.lib, "ncclCommDestroy");
		x.AllGather = (decltype(x.AllGather))dlsym(x.lib, "ncclAllGather");
		x.GetErrorString = (decltype(x.GetErrorString))dlsym(x.lib, "ncclGetErrorString");
		if (!x.GetUniqueId || !x.CommInitRank || !x.CommDestroy || !x.AllGather) x.error = "librccl.so.1 lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
		return x;
	}();
	return r;
}

} // namespace

namespace {

constexpr int P2P_MAX_RANKS = 8;   // one node
constexpr uint32_t P2P_FLAG_PAD = 32; // words between flags (a 128-byte line each: peers write neighbouring flags concurrently)
enum { P2P_READY = 0, P2P_DATA = 1 }; // flag kinds: "my receive slot may be overwritten for sequence s" / "my record of sequence s is in your slot"
struct P2PTargets { int32_t* recv[P2P_MAX_RANKS]; uint32_t* flags[P2P_MAX_RANKS]; };
__host__ __device__ inline uint32_t p2p_flag_index(uint32_t kind, uint32_t slot, uint32_t src) { return ((kind * 2u + slot) * (uint32_t)P2P_MAX_RANKS + src) * P2P_FLAG_PAD; }

// thread p raises flag (kind, slot, me) at peer p. One tiny launch BEHIND the stores it announces: a kernel boundary orders them (a fence per
// storing block would be an L2 write-back each on this chip, profiles/r05/keys_last_block_fences.txt); the release below makes them
// visible beyond this device before the flag.
// `error` (may be null): a step whose slot-free wait timed out has stored NOTHING (k_p2p_scatter returns at once) - its "record stored" flag must
// not rise either, or a peer that did not time out would consume last sequence's record from its slot without an error on its side.
__global__ __launch_bounds__(64) void k_p2p_signal(P2PTargets t, uint32_t world, uint32_t index, uint32_t value, const uint32_t* error) {
	if (error != nullptr && *error != 0) return;
	if (threadIdx.x < world) {
		__threadfence_system();
		__hip_atomic_store(t.flags[threadIdx.x] + index, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}
// thread r waits until the flag (kind, slot, r) in THIS rank's memory has reached `value`. BOUNDED: after timeout_ticks of the 100 MHz
// wall clock the kernel gives up, says so in *error (rank and kind of the missing flag) and ends - a peer that died or fell out of step
// costs the frame and an error code, never the device.
__global__ __launch_bounds__(64) void k_p2p_wait(const uint32_t* flags, uint32_t world, uint32_t base_index, uint32_t value, uint64_t timeout_ticks, uint32_t* error, uint32_t code) {
	if (threadIdx.x >= world) return;
	const uint32_t* f = flags + base_index + threadIdx.x * P2P_FLAG_PAD;
	const uint64_t t0 = wall_clock64();
	while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) { // (sequence numbers wrap)
		if (wall_clock64() - t0 > timeout_ticks) {
			atomicExch(error, code | (threadIdx.x << 8) | 0x80000000u);
			return;
		}
		__builtin_amdgcn_s_sleep(16);
	}
}
// block (peer, frustum, piece): the used part of this rank's sub-record f - its 8 counts and min(sum, cap[f]) ids - into peer's receive slot
__global__ __launch_bounds__(256) void k_p2p_scatter(const int32_t* __restrict__ send, P2PTargets t, uint64_t slot_words_offset, PackLayout lay, const uint32_t* error) {
	if (*error != 0) return; // a peer did not free its slot in time: nothing of it is overwritten
	const int32_t* rec = send + lay.off[blockIdx.y];
	int32_t* dst = t.recv[blockIdx.x] + slot_words_offset + lay.off[blockIdx.y];
	const uint32_t cap_f = lay.cap[blockIdx.y];
	uint32_t total = 0;
#pragma unroll
	for (int k = 0; k < MAX_TYPES; ++k) total += (uint32_t)rec[k];
	const uint32_t words = MAX_TYPES + (total < cap_f ? total : cap_f);
	for (uint32_t w = blockIdx.z * 256u + threadIdx.x; w < words; w += gridDim.z * 256u) dst[w] = rec[w];
}

// What a frame's gathered records say, for the host (pinned, mapped): per frustum the LARGEST list any rank saw (what the next capacities
// are derived from: the same numbers on every rank), the words of its record this rank actually used, the largest used record of any rank.
// One wave: lane = (rank r mod 8, frustum f); ranks beyond 8 take further rounds.
struct FrameStats {
	uint32_t max_visible[MAX_FRUSTA];
	uint32_t used_words_own, used_words_max;
};
__global__ __launch_bounds__(64) void k_exchange_stats(const int32_t* __restrict__ records, uint32_t world, uint32_t rank, uint32_t record, PackLayout lay, uint32_t n_frusta,
	FrameStats* __restrict__ out) {
	const uint32_t f = threadIdx.x & 7u, r0 = threadIdx.x >> 3;
	uint32_t most = 0, used_own = 0, used_max = 0;
	for (uint32_t base = 0; base < world; base += 8u) {
		const uint32_t r = base + r0;
		uint32_t total = 0, used = 0;
		if (r < world && f < n_frusta) {
			const int32_t* rec = records + (size_t)r * record + lay.off[f];
#pragma unroll
			for (int k = 0; k < MAX_TYPES; ++k) total += (uint32_t)rec[k];
			used = MAX_TYPES + (total < lay.cap[f] ? total : lay.cap[f]);
		}
		most = total > most ? total : most;
#pragma unroll
		for (int o = 1; o < 8; o <<= 1) used += (uint32_t)__shfl_xor((int)used, o); // over the frusta of rank r
		if (r == rank) used_own = used;
		used_max = used > used_max ? used : used_max;
	}
#pragma unroll
	for (int o = 8; o < 64; o <<= 1) { // over the ranks
		const uint32_t m = (uint32_t)__shfl_xor((int)most, o), u = (uint32_t)__shfl_xor((int)used_max, o), w = (uint32_t)__shfl_xor((int)used_own, o);
		most = m > most ? m : most;
		used_max = u > used_max ? u : used_max;
		used_own = w > used_own ? w : used_own; // (non-zero on the lanes of one rank only)
	}
	if (threadIdx.x < MAX_FRUSTA) out->max_visible[threadIdx.x] = threadIdx.x < n_frusta ? most : 0u;
	if (threadIdx.x == 0) {
		out->used_words_own = used_own;
		out->used_words_max = used_max;
	}
}

} // namespace

struct LmxExchange {
	enum Mode { INLINE = 0, SIDE = 1, P2P = 2 };
	// P2P: one allocation per rank [recv slot 0 | recv slot 1 | flags], shared with the peers through hipIpc handles
	struct {
		void* block = nullptr;
		size_t slot_words = 0;            // words of one receive slot (max_record * world)
		P2PTargets targets = {};
		void* opened[P2P_MAX_RANKS] = {};  // what hipIpcCloseMemHandle takes
		uint32_t* my_flags = nullptr;
		uint32_t seq[2] = {0, 0};
		uint32_t* h_error = nullptr;      // pinned, mapped: the wait kernels report a timeout here
		uint32_t* d_error = nullptr;
		uint64_t timeout_ticks = 0;
		bool failed = false;
	} p2p;
	// LMX_EXCHANGE_TRACE=1: host time of every API call of a step, summed and printed by lmx_exchange_destroy (a measurement aid)
	bool trace = false;
	double t_host[8] = {};
	uint64_t t_steps = 0;
	LmxContext* ctx = nullptr;
	NcclComm comm = nullptr;
	int rank = 0, world = 1;
	uint32_t cap = 0;         // ids per rank record (all frusta of a frame together): what the buffers are sized for
	size_t max_record = 0;    // words: MAX_FRUSTA headers + cap ids - any layout fits
	// How the step runs. forced_mode >= 0: LMX_EXCHANGE_MODE / LMX_EXCHANGE_INLINE said so; -1 ("auto"): decided PER FRAME SHAPE (number of
	// frusta) from the gather time of the record that shape ships (Shape::mode).
	int forced_mode = -1;
	double overlap_us = 16.0;
	int auto_caps_env = -1;   // LMX_EXCHANGE_AUTO_CAPS: -1 unset (on for frames of >= 2 frusta), 0 never, 1 always
	struct Shape { // per number of frusta of a frame
		bool have_caps = false, caps_fixed = false;
		uint32_t cap[MAX_FRUSTA] = {};
		int mode = -1;             // undecided
		double gather_us = -1.0;   // one all-gather of this shape's record (timed_words words per rank); < 0: never timed
		uint32_t timed_words = 0;
		const char* why = "not decided yet: no frame of this shape has run";
	} shape[MAX_FRUSTA + 1];
	uint32_t last_shape = 1;
	struct Frame { // what slot k's last frame looked like
		uint32_t n_frusta = 1, record = 0;
		PackLayout lay = {};
		int mode = INLINE;
		bool on_side = false, stats_launched = false, stats_read = false;
	} frame[2];
	FrameStats* h_stats = nullptr; // [2], pinned + mapped: written by k_exchange_stats behind a frame's gather
	FrameStats* d_stats = nullptr;
	hipStream_t side = nullptr;
	DevBuf<int32_t> send[2], recv[2];
	hipEvent_t culled[2] = {nullptr, nullptr}, gathered[2] = {nullptr, nullptr};
	bool in_flight[2] = {false, false};
	uint32_t next = 0;
};

extern "C" void lmx_exchange_destroy(LmxExchange* x);

namespace {

int32_t* recv_base(const LmxExchange* x, uint32_t slot) {
	return x->forced_mode == LmxExchange::P2P ? static_cast<int32_t*>(x->p2p.block) + (size_t)slot * x->p2p.slot_words : x->recv[slot].p;
}

constexpr uint32_t CAP_GRAIN = 256; // ids: a sub-record starts on a 1 KiB boundary + 32 B of counts

uint32_t round_cap(uint64_t ids) { return (uint32_t)std::min<uint64_t>((ids + CAP_GRAIN - 1) / CAP_GRAIN * CAP_GRAIN, 0x7fffff00u); }

// what a list of m ids (the largest any rank saw) should be given: 20 % head room, at least one grain
uint32_t cap_for(uint32_t m) { return round_cap((uint64_t)m + std::max<uint64_t>(m / 5u, 64u)); }

// capacities -> offsets of the sub-records; false when they do not fit the buffers (the caller clips them first)
bool make_layout(const LmxExchange* x, uint32_t n_frusta, const uint32_t* cap, LmxExchange::Frame& fr) {
	uint64_t at = 0;
	fr.n_frusta = n_frusta;
	fr.lay = PackLayout{};
	for (uint32_t f = 0; f < n_frusta; ++f) {
		fr.lay.off[f] = at;
		fr.lay.cap[f] = cap[f];
		at += (uint64_t)MAX_TYPES + cap[f];
	}
	fr.record = (uint32_t)at;
	return at <= x->max_record;
}

// The policy (pure: the same inputs on every rank give the same capacities). A list that came within its capacity's head room grows, one
// that shrank to under 5/8 of what it was given shrinks; in between the capacity stays (no re-timing, no layout churn from frame to frame).
// Whatever the lists ask for, the sum stays within the buffers: the capacities are scaled down together (and the frame is flagged as
// overflowed by its readers) - the remedy is an exchange created with more ids_per_rank.
void regrow(const LmxExchange* x, uint32_t n_frusta, const uint32_t* max_visible, uint32_t* cap) {
	uint64_t sum = 0;
	for (uint32_t f = 0; f < n_frusta; ++f) {
		const uint32_t want = cap_for(max_visible[f]);
		if (max_visible[f] > cap[f] || (uint64_t)max_visible[f] * 16u > (uint64_t)cap[f] * 15u || (uint64_t)want * 8u < (uint64_t)cap[f] * 5u) cap[f] = want;
		sum += cap[f];
	}
	if (sum > x->cap) {
		uint64_t left = x->cap;
		for (uint32_t f = 0; f < n_frusta; ++f) {
			const uint64_t share = (uint64_t)cap[f] * x->cap / sum / CAP_GRAIN * CAP_GRAIN;
			cap[f] = (uint32_t)std::min<uint64_t>(share, left);
			left -= cap[f];
		}
	}
}

// P2P mode: this rank's receive slots + flags in ONE allocation, its hipIpc handle all-gathered through the communicator (the only use of
// RCCL in this mode), every peer's allocation mapped here. Collective: every rank runs it inside lmx_exchange_create.
int p2p_setup(LmxExchange* x) {
	LmxContext* ctx = x->ctx;
	Rccl& r = rccl();
	if (x->world > P2P_MAX_RANKS) return fail(ctx, LMX_ERR_CAPACITY, "LMX_EXCHANGE_MODE=p2p serves up to %d ranks (one node), not %d", P2P_MAX_RANKS, x->world);
	const size_t flag_words = (size_t)2 * 2 * P2P_MAX_RANKS * P2P_FLAG_PAD; // kinds x slots x sources
	x->p2p.slot_words = x->max_record * (size_t)x->world;
	const size_t bytes = (2 * x->p2p.slot_words + flag_words) * sizeof(int32_t);
	// FINE-GRAINED memory or nothing: what a peer GPU stores here must not be shadowed by a stale line of this GPU's L2. The acquire on the
	// flag does not make a peer's plain stores into a coarse-grained allocation coherent, so there is no quiet fall-back to hipMalloc (round
	// 5 had one; on one device it cannot be told apart, across devices it is the stale-record case): the caller takes another mode.
	if (hipExtMallocWithFlags(&x->p2p.block, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
		(void)hipGetLastError();
		x->p2p.block = nullptr;
		return fail(ctx, LMX_ERR_NO_DEVICE, "LMX_EXCHANGE_MODE=p2p needs fine-grained device memory for the receive slots (hipExtMallocWithFlags(hipDeviceMallocFinegrained) of %zu bytes failed): use auto, inline or side",
			bytes);
	}
	LMX_HIP(ctx, hipMemset(x->p2p.block, 0, bytes));
	x->p2p.my_flags = static_cast<uint32_t*>(x->p2p.block) + 2 * x->p2p.slot_words;
	hipIpcMemHandle_t mine;
	static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 16 int32");
	if (hipIpcGetMemHandle(&mine, x->p2p.block) != hipSuccess) return fail(ctx, LMX_ERR_HIP, "hipIpcGetMemHandle failed: the receive buffer cannot be shared with the other ranks");
	DevBuf<int32_t> h_send, h_recv;
	LMX_HIP(ctx, h_send.reserve(16));
	LMX_HIP(ctx, h_recv.reserve((size_t)16 * x->world));
	LMX_HIP(ctx, hipMemcpy(h_send.p, &mine, 64, hipMemcpyHostToDevice));
	if (r.AllGather(h_send.p, h_recv.p, 16, NCCL_INT32, x->comm, x->side) != 0) return fail(ctx, LMX_ERR_HIP, "all-gather of the hipIpc handles failed");
	LMX_HIP(ctx, hipStreamSynchronize(x->side));
	hipIpcMemHandle_t all[P2P_MAX_RANKS];
	LMX_HIP(ctx, hipMemcpy(all, h_recv.p, (size_t)64 * x->world, hipMemcpyDeviceToHost));
	for (int p = 0; p < x->world; ++p) {
		void* base = x->p2p.block;
		if (p != x->rank) {
			if (hipIpcOpenMemHandle(&base, all[p], hipIpcMemLazyEnablePeerAccess) != hipSuccess) return fail(ctx, LMX_ERR_HIP, "hipIpcOpenMemHandle of rank %d's receive buffer failed", p);
			x->p2p.opened[p] = base;
		}
		x->p2p.targets.recv[p] = static_cast<int32_t*>(base);
		x->p2p.targets.flags[p] = static_cast<uint32_t*>(base) + 2 * x->p2p.slot_words;
	}
	LMX_HIP(ctx, hipHostMalloc(&x->p2p.h_error, sizeof(uint32_t), hipHostMallocMapped));
	*x->p2p.h_error = 0;
	LMX_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&x->p2p.d_error), x->p2p.h_error, 0));
	const char* to = getenv("LMX_EXCHANGE_P2P_TIMEOUT_MS");
	const double ms = to ? atof(to) : 2000.0; // fractions of a millisecond count ("0.5" is 50 000 ticks, not zero)
	x->p2p.timeout_ticks = ms > 0.0 ? (uint64_t)(ms * 100000.0 + 0.5) : 1ull; // the wall clock of the wait kernels ticks at 100 MHz
	// nobody stores into a peer before every peer has zeroed its flags and mapped everybody: one more collective as the barrier
	if (r.AllGather(h_send.p, h_recv.p, 16, NCCL_INT32, x->comm, x->side) != 0) return fail(ctx, LMX_ERR_HIP, "the barrier behind the hipIpc mappings failed");
	LMX_HIP(ctx, hipStreamSynchronize(x->side));
	return LMX_OK;
}

// K gathers of `words` words per rank back to back on the side stream, by events: what ONE gather of that record costs here. A COLLECTIVE:
// every rank gets here in the same call (the frame shapes and their capacities are the same everywhere). Both streams are idle afterwards.
int time_gathers(LmxExchange* x, uint32_t slot, uint32_t words, double* us) {
	LmxContext* ctx = x->ctx;
	Rccl& r = rccl();
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream)); // (nothing of this exchange is in flight while the slot's buffer carries the probe)
	LMX_HIP(ctx, hipStreamSynchronize(x->side));
	const int K = 32;
	hipEvent_t a = nullptr, b = nullptr;
	LMX_HIP(ctx, hipEventCreate(&a));
	LMX_HIP(ctx, hipEventCreate(&b));
	int32_t* const buf = x->recv[slot].p; // in place, like the step: this rank's record sits in its segment
	int rc = 0;
	for (int k = 0; k < 4 && rc == 0; ++k) rc = r.AllGather(buf + (size_t)x->rank * words, buf, words, NCCL_INT32, x->comm, x->side); // (connection set-up is not the gather)
	if (rc == 0 && hipEventRecord(a, x->side) != hipSuccess) rc = -1;
	for (int k = 0; k < K && rc == 0; ++k) rc = r.AllGather(buf + (size_t)x->rank * words, buf, words, NCCL_INT32, x->comm, x->side);
	float ms = 0;
	if (rc == 0 && (hipEventRecord(b, x->side) != hipSuccess || hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess)) rc = -1;
	(void)hipEventDestroy(a);
	(void)hipEventDestroy(b);
	if (rc != 0) return fail(ctx, LMX_ERR_HIP, "timing the all-gather failed");
	*us = 1e3 * ms / K;
	return LMX_OK;
}

// the mode of this frame: forced, or - "auto" - decided for the frame's shape on the record it ships now
int mode_for(LmxExchange* x, uint32_t slot, uint32_t n_frusta, uint32_t record, int* mode) {
	if (x->forced_mode >= 0) {
		*mode = x->forced_mode;
		return LMX_OK;
	}
	LmxExchange::Shape& sh = x->shape[n_frusta];
	if (sh.mode < 0 || record > 2u * sh.timed_words || 2u * record < sh.timed_words) {
		if (int rc = time_gathers(x, slot, record, &sh.gather_us)) return rc;
		sh.timed_words = record;
		sh.mode = sh.gather_us > x->overlap_us ? LmxExchange::SIDE : LmxExchange::INLINE;
		sh.why = sh.mode == LmxExchange::SIDE ? "auto: one gather of this frame shape's record takes longer than the host work the side stream adds"
		                                      : "auto: one gather of this frame shape's record is shorter than the host work the side stream adds";
	}
	*mode = sh.mode;
	return LMX_OK;
}

bool auto_caps(const LmxExchange* x, uint32_t n_frusta) { return x->auto_caps_env < 0 ? n_frusta >= 2 : x->auto_caps_env == 1; }

hipError_t launch_stats(LmxExchange* x, uint32_t k, hipStream_t s) {
	const LmxExchange::Frame& fr = x->frame[k];
	hipLaunchKernelGGL(k_exchange_stats, dim3(1), dim3(64), 0, s, recv_base(x, k), (uint32_t)x->world, (uint32_t)x->rank, fr.record, fr.lay, fr.n_frusta, x->d_stats + k);
	return hipGetLastError();
}

} // namespace

extern "C" {

int lmx_exchange_unique_id(void* out_id_128_bytes) {
	if (!out_id_128_bytes) return LMX_ERR_INVALID_ARGUMENT;
	Rccl& r = rccl();
	if (r.error) return LMX_ERR_NO_DEVICE;
	NcclUniqueId id;
	if (r.GetUniqueId(&id) != 0) return LMX_ERR_HIP;
	memcpy(out_id_128_bytes, &id, sizeof(id));
	return LMX_OK;
}

int lmx_exchange_create(LmxContext* ctx, int rank, int world, const void* unique_id_128_bytes, uint32_t ids_per_rank, LmxExchange** out) {
	LMX_CHECK_CTX(ctx);
	if (!out || !unique_id_128_bytes || world < 1 || rank < 0 || rank >= world || ids_per_rank == 0) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad exchange arguments");
	if (ids_per_rank > 0x7fffffffu - LMX_MAX_FRUSTA * LMX_MAX_TYPES) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "ids_per_rank %u does not fit a 32-bit record", ids_per_rank);
	Rccl& r = rccl();
	if (r.error) return fail(ctx, LMX_ERR_NO_DEVICE, "RCCL is not available: %s", r.error);
	// ---- how the step runs (validated before any rank joins the communicator)
	const char* inl = getenv("LMX_EXCHANGE_INLINE");
	const char* mode_env = getenv("LMX_EXCHANGE_MODE");
	const std::string mode = mode_env ? mode_env : (inl ? (inl[0] == '0' ? "side" : "inline") : "auto");
	int forced = -1;
	if (mode == "side") forced = LmxExchange::SIDE;
	else if (mode == "inline") forced = LmxExchange::INLINE;
	else if (mode == "p2p") forced = LmxExchange::P2P;
	else if (mode != "auto") return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "LMX_EXCHANGE_MODE=%s: auto, inline, side or p2p", mode.c_str());
	LmxExchange* x = new LmxExchange;
	x->ctx = ctx;
	x->trace = getenv("LMX_EXCHANGE_TRACE") != nullptr;
	x->forced_mode = forced;
	if (const char* thr = getenv("LMX_EXCHANGE_OVERLAP_US")) x->overlap_us = atof(thr);
	if (const char* ac = getenv("LMX_EXCHANGE_AUTO_CAPS")) x->auto_caps_env = ac[0] == '0' ? 0 : 1;
	x->rank = rank;
	x->world = world;
	x->cap = ids_per_rank;
	x->max_record = (size_t)MAX_FRUSTA * MAX_TYPES + ids_per_rank; // any split of the capacity over <= 8 frusta fits
	for (int k = 0; k < 2; ++k) {
		const uint32_t one[1] = {ids_per_rank};
		make_layout(x, 1, one, x->frame[k]);
	}
	for (LmxExchange::Shape& sh : x->shape)
		if (forced >= 0) {
			sh.mode = forced;
			sh.why = "forced";
		}
	NcclUniqueId id;
	memcpy(&id, unique_id_128_bytes, sizeof(id));
	const int rc = r.CommInitRank(&x->comm, world, id, rank);
	if (rc != 0) {
		const char* msg = r.GetErrorString ? r.GetErrorString(rc) : "?";
		delete x;
		return fail(ctx, LMX_ERR_HIP, "ncclCommInitRank failed: %s", msg);
	}
	// from here on a failure must not leak the communicator (the other ranks would be left with a half-alive one), the side stream,
	// the events or the buffers: everything goes through lmx_exchange_destroy
	auto setup = [&]() -> int {
		LMX_HIP(ctx, hipStreamCreateWithFlags(&x->side, hipStreamNonBlocking));
		for (int i = 0; i < 2; ++i) {
			if (forced == LmxExchange::P2P) LMX_HIP(ctx, x->send[i].reserve(x->max_record)); // (the collective forms are in place: no send buffer)
			else LMX_HIP(ctx, x->recv[i].reserve(x->max_record * world));
			LMX_HIP(ctx, hipEventCreateWithFlags(&x->culled[i], hipEventDisableTiming));
			LMX_HIP(ctx, hipEventCreateWithFlags(&x->gathered[i], hipEventDisableTiming));
		}
		LMX_HIP(ctx, hipHostMalloc(&x->h_stats, 2 * sizeof(FrameStats), hipHostMallocMapped));
		memset(x->h_stats, 0, 2 * sizeof(FrameStats));
		LMX_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&x->d_stats), x->h_stats, 0));
		if (forced == LmxExchange::P2P) return p2p_setup(x);
		return LMX_OK;
	};
	if (int rc2 = setup()) {
		lmx_exchange_destroy(x);
		return rc2;
	}
	*out = x;
	return LMX_OK;
}

void lmx_exchange_destroy(LmxExchange* x) {
	if (!x) return;
	if (x->trace && x->t_steps) {
		const double n = (double)x->t_steps;
		fprintf(stderr, "lmx_exchange trace (%llu steps, host us per step): wait-for-slot %.2f | lmx_cull %.2f | pack launches %.2f | event record %.2f | side stream wait %.2f | ncclAllGather %.2f | event record (side) %.2f\n",
			(unsigned long long)x->t_steps, x->t_host[0] / n, x->t_host[1] / n, x->t_host[2] / n, x->t_host[3] / n, x->t_host[4] / n, x->t_host[5] / n, x->t_host[6] / n);
	}
	if (x->side) (void)hipStreamSynchronize(x->side);
	if (x->ctx) (void)hipStreamSynchronize(x->ctx->stream); // (the gathers of the inline form, the stores and waits of the P2P form run there)
	for (int p = 0; p < P2P_MAX_RANKS; ++p)
		if (x->p2p.opened[p]) (void)hipIpcCloseMemHandle(x->p2p.opened[p]);
	if (x->p2p.block) (void)hipFree(x->p2p.block);
	if (x->p2p.h_error) (void)hipHostFree(x->p2p.h_error);
	if (x->h_stats) (void)hipHostFree(x->h_stats);
	if (x->comm) (void)rccl().CommDestroy(x->comm);
	for (int i = 0; i < 2; ++i) {
		if (x->culled[i]) (void)hipEventDestroy(x->culled[i]);
		if (x->gathered[i]) (void)hipEventDestroy(x->gathered[i]);
	}
	if (x->side) (void)hipStreamDestroy(x->side);
	delete x;
}

// Capacities of the sub-records of frames of n_frusta frusta from now on (ids per frustum; their sum <= ids_per_rank of the exchange). EVERY RANK
// must make the same call between the same two frames. keep_fixed != 0: the capacities stay as given (no regrowth from the gathered counts).
int lmx_exchange_set_caps(LmxExchange* x, uint32_t n_frusta, const uint32_t* caps, int keep_fixed) {
	if (!x) return LMX_ERR_INVALID_ARGUMENT;
	LmxContext* ctx = x->ctx;
	LMX_CHECK_CTX(ctx);
	if (!caps || n_frusta < 1 || n_frusta > LMX_MAX_FRUSTA) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "lmx_exchange_set_caps: n_frusta %u not in [1,%d] or no capacities", n_frusta, LMX_MAX_FRUSTA);
	uint64_t sum = 0;
	for (uint32_t f = 0; f < n_frusta; ++f) sum += caps[f];
	if (sum > x->cap) return fail(ctx, LMX_ERR_CAPACITY, "lmx_exchange_set_caps: %llu ids in all, the exchange was created for %u per rank", (unsigned long long)sum, x->cap);
	LmxExchange::Shape& sh = x->shape[n_frusta];
	memcpy(sh.cap, caps, n_frusta * sizeof(uint32_t));
	sh.have_caps = true;
	sh.caps_fixed = keep_fixed != 0;
	return LMX_OK;
}

// One frame of this rank: cull the frame's n_frusta views (the reference culls its 4 shadow cascades + main view + light query per
// frame, pipeline.cpp:1036-1045, :1252-1258) over the entities this context holds, in one lmx_cull (result slot = view `slot`), pack
// every frustum's visible ids behind its 8 counts, and enqueue ONE all-gather of the whole record:
//     rank record = n_frusta x [LMX_MAX_TYPES counts | cap[f] ids, types packed back to back]
// Every rank must pass the same n_frusta. Returns the slot (0 / 1) to pass to lmx_exchange_wait / lmx_exchange_result; the slot's
// previous gather must have been waited for or is waited for here.
int lmx_exchange_cull_many(LmxExchange* x, const LmxShiftedFrustum* frusta, uint32_t n_frusta, uint8_t type, uint32_t* out_slot) {
	if (!x) return LMX_ERR_INVALID_ARGUMENT;
	LmxContext* ctx = x->ctx;
	LMX_CHECK_CTX(ctx);
	if (n_frusta < 1 || n_frusta > LMX_MAX_FRUSTA) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "n_frusta %u not in [1,%d]", n_frusta, LMX_MAX_FRUSTA);
	if (x->cap / n_frusta == 0) return fail(ctx, LMX_ERR_CAPACITY, "%u ids per rank cannot be split over %u frusta", x->cap, n_frusta);
	const uint32_t k = x->next;
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto lap = [&](int i, std::chrono::steady_clock::time_point& t) {
		if (!x->trace) return;
		const auto t1 = now();
		x->t_host[i] += std::chrono::duration<double, std::micro>(t1 - t).count();
		t = t1;
	};
	auto t = now();
	const bool p2p = x->forced_mode == LmxExchange::P2P;
	if (p2p && x->p2p.failed) return fail(ctx, LMX_ERR_BUSY, "this exchange's P2P mode has failed before (a peer fell out of step): destroy it and create one in another mode");
	// ---- this frame's capacities: the shape's current ones, regrown from what the frame that used this slot last (two frames ago) gathered
	LmxExchange::Shape& sh = x->shape[n_frusta];
	LmxExchange::Frame& fr = x->frame[k];
	if (!sh.have_caps) {
		for (uint32_t f = 0; f < n_frusta; ++f) sh.cap[f] = x->cap / n_frusta;
		sh.have_caps = true;
	}
	if (!sh.caps_fixed && fr.stats_launched && !fr.stats_read && fr.n_frusta == n_frusta && x->in_flight[k]) {
		LMX_HIP(ctx, hipEventSynchronize(x->gathered[k])); // (that frame's gather + its statistics: two frames back, normally long done)
		regrow(x, n_frusta, x->h_stats[k].max_visible, sh.cap);
		fr.stats_read = true;
	}
	LmxExchange::Frame nf;
	if (!make_layout(x, n_frusta, sh.cap, nf)) return fail(ctx, LMX_ERR_CAPACITY, "the capacities of a %u-frusta frame do not fit the exchange's %u ids per rank", n_frusta, x->cap);
	int mode = LmxExchange::INLINE;
	if (int rc = mode_for(x, k, n_frusta, nf.record, &mode)) return rc;
	nf.mode = mode;
	nf.on_side = mode == LmxExchange::SIDE;
	x->next ^= 1u;
	x->last_shape = n_frusta;
	uint32_t seq = 0;
	if (p2p) {
		// this rank's slot k may be overwritten for sequence `seq` (the caller is done with its previous contents: the API contract) - said
		// first, so that the peers' stores never wait for this rank's cull
		seq = ++x->p2p.seq[k];
		hipLaunchKernelGGL(k_p2p_signal, dim3(1), dim3(64), 0, ctx->stream, x->p2p.targets, (uint32_t)x->world, p2p_flag_index(P2P_READY, k, (uint32_t)x->rank), seq, (const uint32_t*)nullptr);
	}
	// the receive buffer of this slot is free once its previous gather has finished: when that one ran on the side stream, the cull stream
	// waits for it (device-side); on the cull stream itself, stream order covers it
	if (x->in_flight[k] && fr.on_side) LMX_HIP(ctx, hipStreamWaitEvent(ctx->stream, x->gathered[k], 0));
	lap(0, t);
	if (int rc = lmx_cull(ctx, k, frusta, n_frusta, type)) return rc;
	lap(1, t);
	CullState& cs = ctx->cull;
	CullView& v = cs.views[k];
	const uint32_t cnt_frustum_stride = cs.n_shards * cs.cnt_pad;
	// The collective is IN PLACE: this rank's record is packed where the gather would put it (recv + rank * record), so RCCL moves the
	// peers' records only - no local copy at any world size, nothing at all in a world of one. The P2P form keeps a send buffer: its scatter
	// writes every peer's slot, this rank's included.
	int32_t* const own = p2p ? x->send[k].p : x->recv[k].p + (size_t)x->rank * nf.record;
	// per frustum: per-type totals straight into its sub-record's header and the ids behind it (clipped to cap[f]) - one launch for all sub-records
	LMX_HIP(ctx, launch_cull_pack_layout(ctx->stream, v.out.p, cs.d_win_base.p, v.counts_ptr(), cs.cnt_pad, cs.d_shard_type.p, cs.n_shards, cs.max_shard_cap, own, nf.lay, n_frusta,
		(uint32_t)v.out_stride, cnt_frustum_stride));
	const bool stats = auto_caps(x, n_frusta) && !sh.caps_fixed;
	nf.stats_launched = stats;
	nf.stats_read = false;
	fr = nf;
	lap(2, t);
	if (p2p) {
		// every peer's slot k is free -> the used part of this rank's record into all of them -> "my record of sequence seq is there" ->
		// everybody's record of this sequence is here. Two bounded waits; a timeout is reported by lmx_exchange_wait.
		uint32_t most = 0;
		for (uint32_t f = 0; f < n_frusta; ++f) most = std::max(most, fr.lay.cap[f]);
		hipLaunchKernelGGL(k_p2p_wait, dim3(1), dim3(64), 0, ctx->stream, x->p2p.my_flags, (uint32_t)x->world, p2p_flag_index(P2P_READY, k, 0), seq, x->p2p.timeout_ticks, x->p2p.d_error, 1u);
		hipLaunchKernelGGL(k_p2p_scatter, dim3((uint32_t)x->world, n_frusta, std::max(1u, std::min(64u, most / 16384u))), dim3(256), 0, ctx->stream, x->send[k].p, x->p2p.targets,
			(uint64_t)k * x->p2p.slot_words + (uint64_t)x->rank * fr.record, fr.lay, x->p2p.d_error);
		hipLaunchKernelGGL(k_p2p_signal, dim3(1), dim3(64), 0, ctx->stream, x->p2p.targets, (uint32_t)x->world, p2p_flag_index(P2P_DATA, k, (uint32_t)x->rank), seq, (const uint32_t*)x->p2p.d_error);
		hipLaunchKernelGGL(k_p2p_wait, dim3(1), dim3(64), 0, ctx->stream, x->p2p.my_flags, (uint32_t)x->world, p2p_flag_index(P2P_DATA, k, 0), seq, x->p2p.timeout_ticks, x->p2p.d_error, 2u);
		LMX_HIP(ctx, hipGetLastError());
		if (stats) LMX_HIP(ctx, launch_stats(x, k, ctx->stream));
		LMX_HIP(ctx, hipEventRecord(x->gathered[k], ctx->stream));
		x->in_flight[k] = true;
		if (out_slot) *out_slot = k;
		return LMX_OK;
	}
	hipStream_t gather_stream = x->side;
	if (!fr.on_side) {
		gather_stream = ctx->stream; // behind the pack kernel in stream order
	} else {
		LMX_HIP(ctx, hipEventRecord(x->culled[k], ctx->stream));
		lap(3, t);
		LMX_HIP(ctx, hipStreamWaitEvent(x->side, x->culled[k], 0));
		lap(4, t);
	}
	const int rc = rccl().AllGather(own, x->recv[k].p, (size_t)fr.record, NCCL_INT32, x->comm, gather_stream);
	if (rc != 0) return fail(ctx, LMX_ERR_HIP, "ncclAllGather failed: %s", rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
	lap(5, t);
	if (stats) LMX_HIP(ctx, launch_stats(x, k, gather_stream));
	LMX_HIP(ctx, hipEventRecord(x->gathered[k], gather_stream));
	lap(6, t);
	x->t_steps += x->trace ? 1 : 0;
	x->in_flight[k] = true;
	if (out_slot) *out_slot = k;
	return LMX_OK;
}

int lmx_exchange_cull(LmxExchange* x, const LmxShiftedFrustum* frustum, uint8_t type, uint32_t* out_slot) { return lmx_exchange_cull_many(x, frustum, 1, type, out_slot); }

// Host wait for the gather of `slot` (a consumer on another stream can instead make that stream wait: lmx_exchange_result's event; in P2P
// mode such a consumer must ALSO look at the error this call reports - a frame whose bounded wait gave up holds stale records).
int lmx_exchange_wait(LmxExchange* x, uint32_t slot) {
	if (!x || slot > 1) return LMX_ERR_INVALID_ARGUMENT;
	LmxContext* ctx = x->ctx;
	LMX_CHECK_CTX(ctx);
	if (x->in_flight[slot]) LMX_HIP(ctx, hipEventSynchronize(x->gathered[slot]));
	if (x->forced_mode == LmxExchange::P2P && *x->p2p.h_error != 0) {
		const uint32_t e = *x->p2p.h_error;
		x->p2p.failed = true;
		return fail(ctx, LMX_ERR_BUSY, "P2P exchange: rank %u's %s flag did not arrive within the bounded wait (a peer died or fell out of step); the frame is lost, the exchange unusable",
			(e >> 8) & 0xffu, (e & 0xffu) == 1u ? "slot-free" : "record-stored");
	}
	return LMX_OK;
}

// How this exchange runs the frames of the shape it ran last and why: mode 0 = all-gather on the cull stream, 1 = on a side stream (overlaps
// the next cull), 2 = P2P stores; gather_us = one all-gather of that shape's record as last timed (< 0: never timed - the mode was forced, or
// no frame has run yet).
int lmx_exchange_info(LmxExchange* x, int* mode, double* gather_us, const char** why) {
	if (!x) return LMX_ERR_INVALID_ARGUMENT;
	const LmxExchange::Shape& sh = x->shape[x->last_shape];
	if (mode) *mode = x->forced_mode >= 0 ? x->forced_mode : (sh.mode >= 0 ? sh.mode : (int)LmxExchange::INLINE);
	if (gather_us) *gather_us = sh.gather_us;
	if (why) *why = sh.why;
	return LMX_OK;
}

// One all-gather of the record frames of n_frusta frusta ship NOW (their current capacities), timed over 32 gathers on the side stream. A
// COLLECTIVE: every rank calls it at the same point. Waits for everything this exchange has in flight. Not in P2P mode (no collective to time).
int lmx_exchange_time_gather(LmxExchange* x, uint32_t n_frusta, double* out_us, uint32_t* out_record_words) {
	if (!x) return LMX_ERR_INVALID_ARGUMENT;
	LmxContext* ctx = x->ctx;
	LMX_CHECK_CTX(ctx);
	if (n_frusta < 1 || n_frusta > LMX_MAX_FRUSTA) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "n_frusta %u not in [1,%d]", n_frusta, LMX_MAX_FRUSTA);
	if (x->forced_mode == LmxExchange::P2P) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "lmx_exchange_time_gather: the P2P form has no collective");
	LmxExchange::Shape& sh = x->shape[n_frusta];
	if (!sh.have_caps) {
		for (uint32_t f = 0; f < n_frusta; ++f) sh.cap[f] = x->cap / n_frusta;
		sh.have_caps = true;
	}
	LmxExchange::Frame probe;
	if (!make_layout(x, n_frusta, sh.cap, probe)) return fail(ctx, LMX_ERR_CAPACITY, "the capacities of a %u-frusta frame do not fit", n_frusta);
	// the probe runs in the slot the NEXT frame takes: that slot's previous contents are dead by the API contract; its bookkeeping says so
	const uint32_t k = x->next;
	double us = -1.0;
	if (int rc = time_gathers(x, k, probe.record, &us)) return rc;
	x->in_flight[k] = false;
	x->frame[k].stats_launched = false;
	sh.gather_us = us;
	sh.timed_words = probe.record;
	if (out_us) *out_us = us;
	if (out_record_words) *out_record_words = probe.record;
	return LMX_OK;
}

// Layout of the gathered records of `slot`: rank r's record starts at r * record_words; sub-record f of a rank at offsets[f] from there:
// LMX_MAX_TYPES counts, then caps[f] ids. Arrays of LMX_MAX_FRUSTA entries; any pointer may be NULL.
int lmx_exchange_layout(LmxExchange* x, uint32_t slot, uint32_t* n_frusta, uint32_t* caps, uint32_t* offsets, uint32_t* record_words) {
	if (!x || slot > 1) return LMX_ERR_INVALID_ARGUMENT;
	const LmxExchange::Frame& fr = x->frame[slot];
	if (n_frusta) *n_frusta = fr.n_frusta;
	for (uint32_t f = 0; f < (uint32_t)MAX_FRUSTA; ++f) {
		if (caps) caps[f] = f < fr.n_frusta ? fr.lay.cap[f] : 0u;
		if (offsets) offsets[f] = f < fr.n_frusta ? (uint32_t)fr.lay.off[f] : 0u;
	}
	if (record_words) *record_words = fr.record;
	return LMX_OK;
}

// What the frame in `slot` shipped and what of it was used (waits for its gather; when the frame did not compute its statistics itself -
// frames of one frustum by default - a one-wave kernel does it here): see LmxExchangeStats.
int lmx_exchange_stats(LmxExchange* x, uint32_t slot, LmxExchangeStats* out) {
	if (!x || slot > 1 || !out) return LMX_ERR_INVALID_ARGUMENT;
	LmxContext* ctx = x->ctx;
	LMX_CHECK_CTX(ctx);
	if (!x->in_flight[slot]) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_exchange_stats: no frame has run in slot %u", slot);
	if (int rc = lmx_exchange_wait(x, slot)) return rc;
	LmxExchange::Frame& fr = x->frame[slot];
	if (!fr.stats_launched) {
		LMX_HIP(ctx, launch_stats(x, slot, ctx->stream));
		LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	}
	const FrameStats& st = x->h_stats[slot];
	memset(out, 0, sizeof(*out));
	out->n_frusta = fr.n_frusta;
	out->record_words = fr.record;
	out->mode = fr.mode;
	out->gather_us = x->shape[fr.n_frusta].gather_us;
	out->gather_us_record_words = x->shape[fr.n_frusta].timed_words;
	for (uint32_t f = 0; f < fr.n_frusta; ++f) {
		out->caps[f] = fr.lay.cap[f];
		out->max_visible[f] = st.max_visible[f];
		if (st.max_visible[f] > fr.lay.cap[f]) out->overflow_mask |= 1u << f;
	}
	out->used_words_own = st.used_words_own;
	out->used_words_max = st.used_words_max;
	// what leaves this rank towards EACH peer: the fixed-size record in the collective forms, the used part of it in the P2P form
	out->bytes_shipped_per_peer = sizeof(int32_t) * (uint64_t)(fr.mode == LmxExchange::P2P ? st.used_words_own : fr.record);
	out->bytes_used = sizeof(int32_t) * (uint64_t)st.used_words_own;
	return LMX_OK;
}

// Device view of the gathered records of `slot`: rank r's record starts at d_records + r * record_words; its sub-records as lmx_exchange_layout
// says (one frustum: LMX_MAX_TYPES counts - what the rank saw, also when it exceeds the capacity: an overflow is visible as sum(counts) >
// capacity - then the ids, type 0 first). `gathered_event` (hipEvent_t as void*) is recorded when the collective has finished.
int lmx_exchange_result(LmxExchange* x, uint32_t slot, const int32_t** d_records, uint32_t* record_words, void** gathered_event) {
	if (!x || slot > 1) return LMX_ERR_INVALID_ARGUMENT;
	if (d_records) *d_records = recv_base(x, slot);
	if (record_words) *record_words = x->frame[slot].record;
	if (gathered_event) *gathered_event = x->gathered[slot];
	return LMX_OK;
}

// Host copy of one (rank, frustum) sub-record of `slot`: counts[LMX_MAX_TYPES] and min(sum(counts), the sub-record's capacity, cap) ids. Waits for
// the gather. (A clipped list is visible as sum(counts) > the sub-record's capacity, lmx_exchange_layout.)
int lmx_exchange_read_many(LmxExchange* x, uint32_t slot, int rank, uint32_t frustum, uint32_t* out_counts, int32_t* out_ids, uint32_t cap) {
	if (!x || slot > 1 || rank < 0 || rank >= x->world || !out_counts || frustum >= x->frame[slot].n_frusta) return LMX_ERR_INVALID_ARGUMENT;
	LmxContext* ctx = x->ctx;
	LMX_CHECK_CTX(ctx);
	if (int rc = lmx_exchange_wait(x, slot)) return rc;
	const LmxExchange::Frame& fr = x->frame[slot];
	const int32_t* rec = recv_base(x, slot) + (size_t)rank * fr.record + fr.lay.off[frustum];
	LMX_HIP(ctx, hipMemcpy(out_counts, rec, sizeof(uint32_t) * MAX_TYPES, hipMemcpyDeviceToHost));
	uint64_t total = 0;
	for (int t = 0; t < MAX_TYPES; ++t) total += out_counts[t];
	const uint32_t n = (uint32_t)std::min<uint64_t>(total, std::min(fr.lay.cap[frustum], cap));
	if (n && out_ids) LMX_HIP(ctx, hipMemcpy(out_ids, rec + MAX_TYPES, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
	return LMX_OK;
}

int lmx_exchange_read(LmxExchange* x, uint32_t slot, int rank, uint32_t* out_counts, int32_t* out_ids, uint32_t cap) {
	return lmx_exchange_read_many(x, slot, rank, 0, out_counts, out_ids, cap);
}

} // extern "C"
