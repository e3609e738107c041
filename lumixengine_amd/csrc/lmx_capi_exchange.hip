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
// LMX_EXCHANGE_MODE=auto (the default when nothing is forced) times K gathers at creation and takes SIDE when one gather is longer
// than the host work SIDE adds (LMX_EXCHANGE_OVERLAP_US, default 16), INLINE otherwise: with one rank the gather is a local copy of a few
// microseconds, over xGMI with eight it is not - the choice is made where the exchange runs, not where it was written.
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
__global__ __launch_bounds__(64) void k_p2p_signal(P2PTargets t, uint32_t world, uint32_t index, uint32_t value) {
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
// block (peer, frustum, piece): the used part of this rank's sub-record f - its 8 counts and min(sum, cap_f) ids - into peer's receive slot
__global__ __launch_bounds__(256) void k_p2p_scatter(const int32_t* __restrict__ send, P2PTargets t, uint32_t slot_words_offset, uint32_t sub, uint32_t cap_f, const uint32_t* error) {
	if (*error != 0) return; // a peer did not free its slot in time: nothing of it is overwritten
	const int32_t* rec = send + (size_t)blockIdx.y * sub;
	int32_t* dst = t.recv[blockIdx.x] + slot_words_offset + (size_t)blockIdx.y * sub;
	uint32_t total = 0;
#pragma unroll
	for (int k = 0; k < MAX_TYPES; ++k) total += (uint32_t)rec[k];
	const uint32_t words = MAX_TYPES + (total < cap_f ? total : cap_f);
	for (uint32_t w = blockIdx.z * 256u + threadIdx.x; w < words; w += gridDim.z * 256u) dst[w] = rec[w];
}

} // namespace

struct LmxExchange {
	enum Mode { INLINE = 0, SIDE = 1, P2P = 2 };
	int mode = INLINE;
	double gather_us = -1.0;   // one all-gather of this exchange's record, timed at creation (LMX_EXCHANGE_MODE=auto); < 0: not measured
	const char* mode_why = "default";
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
	// Default (round 4): the all-gather is enqueued on the CULL stream, behind the pack kernel - no side stream, no event pair between
	// the two, no wait for the slot's previous gather (stream order covers it): four API calls instead of seven per step (20.5 against
	// 36.6 us per step with one rank on the round-3 driver box), at the price of the cull of frame k + 1 not overlapping the gather of
	// frame k. LMX_EXCHANGE_INLINE=0 (read at creation) selects the side-stream, double-buffered form again: which side wins with 8
	// ranks depends on how long the gather takes over xGMI, and no multi-GPU number exists yet. Same results either way
	// (tests/test_gpu_exchange.py runs both).
	bool inline_gather = true;
	double t_host[8] = {};
	uint64_t t_steps = 0;
	LmxContext* ctx = nullptr;
	NcclComm comm = nullptr;
	int rank = 0, world = 1;
	uint32_t cap = 0;      // ids per rank record (all frusta of a frame together)
	uint32_t record = 0;   // words per rank record of the slot's last frame = n_frusta * (LMX_MAX_TYPES + cap / n_frusta)
	uint32_t n_frusta[2] = {1, 1}, cap_f[2] = {0, 0}; // layout of each slot's last frame
	hipStream_t side = nullptr;
	DevBuf<int32_t> send[2], recv[2];
	DevBuf<uint32_t> packed_start[2];
	hipEvent_t culled[2] = {nullptr, nullptr}, gathered[2] = {nullptr, nullptr};
	bool in_flight[2] = {false, false};
	uint32_t next = 0;
};

extern "C" void lmx_exchange_destroy(LmxExchange* x);

namespace {

const int32_t* recv_base(const LmxExchange* x, uint32_t slot) {
	return x->mode == LmxExchange::P2P ? static_cast<const int32_t*>(x->p2p.block) + (size_t)slot * x->p2p.slot_words : x->recv[slot].p;
}

// P2P mode: this rank's receive slots + flags in ONE allocation, its hipIpc handle all-gathered through the communicator (the only use of
// RCCL in this mode), every peer's allocation mapped here. Collective: every rank runs it inside lmx_exchange_create.
int p2p_setup(LmxExchange* x, size_t max_record) {
	LmxContext* ctx = x->ctx;
	Rccl& r = rccl();
	if (x->world > P2P_MAX_RANKS) return fail(ctx, LMX_ERR_CAPACITY, "LMX_EXCHANGE_MODE=p2p serves up to %d ranks (one node), not %d", P2P_MAX_RANKS, x->world);
	const size_t flag_words = (size_t)2 * 2 * P2P_MAX_RANKS * P2P_FLAG_PAD; // kinds x slots x sources
	x->p2p.slot_words = max_record * (size_t)x->world;
	const size_t bytes = (2 * x->p2p.slot_words + flag_words) * sizeof(int32_t);
	// fine-grained memory where the runtime has it: what a peer GPU stores here must not be shadowed by a stale line of this GPU's L2
	if (hipExtMallocWithFlags(&x->p2p.block, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
		(void)hipGetLastError();
		x->p2p.block = nullptr;
		LMX_HIP(ctx, hipMalloc(&x->p2p.block, bytes));
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
	x->p2p.timeout_ticks = (uint64_t)(to ? atof(to) : 2000.0) * 100000ull; // the wall clock of the wait kernels ticks at 100 MHz
	// nobody stores into a peer before every peer has zeroed its flags and mapped everybody: one more collective as the barrier
	if (r.AllGather(h_send.p, h_recv.p, 16, NCCL_INT32, x->comm, x->side) != 0) return fail(ctx, LMX_ERR_HIP, "the barrier behind the hipIpc mappings failed");
	LMX_HIP(ctx, hipStreamSynchronize(x->side));
	return LMX_OK;
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
	LmxExchange* x = new LmxExchange;
	x->ctx = ctx;
	x->trace = getenv("LMX_EXCHANGE_TRACE") != nullptr;
	const char* inl = getenv("LMX_EXCHANGE_INLINE");
	x->inline_gather = !(inl != nullptr && inl[0] == '0');
	x->rank = rank;
	x->world = world;
	x->cap = ids_per_rank;
	x->record = LMX_MAX_TYPES + ids_per_rank;
	x->cap_f[0] = x->cap_f[1] = ids_per_rank;
	const size_t max_record = (size_t)MAX_FRUSTA * MAX_TYPES + ids_per_rank; // any split of the capacity over <= 8 frusta fits
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
			LMX_HIP(ctx, x->send[i].reserve(max_record));
			LMX_HIP(ctx, x->recv[i].reserve(max_record * world));
			LMX_HIP(ctx, x->packed_start[i].reserve(MAX_FRUSTA * MAX_TYPES));
			LMX_HIP(ctx, hipEventCreateWithFlags(&x->culled[i], hipEventDisableTiming));
			LMX_HIP(ctx, hipEventCreateWithFlags(&x->gathered[i], hipEventDisableTiming));
		}
		return LMX_OK;
	};
	if (int rc2 = setup()) {
		lmx_exchange_destroy(x);
		return rc2;
	}
	// ---- how the step runs
	const char* mode_env = getenv("LMX_EXCHANGE_MODE");
	std::string mode = mode_env ? mode_env : (inl ? (x->inline_gather ? "inline" : "side") : "auto");
	auto time_gathers = [&]() -> int { // K gathers of the 1-frustum record back to back on the side stream, by events: what ONE gather costs there
		const int K = 32;
		hipEvent_t a = nullptr, b = nullptr;
		LMX_HIP(ctx, hipEventCreate(&a));
		LMX_HIP(ctx, hipEventCreate(&b));
		int rc = 0;
		for (int k = 0; k < 4 && rc == 0; ++k) rc = r.AllGather(x->send[0].p, x->recv[0].p, x->record, NCCL_INT32, x->comm, x->side); // (connection set-up is not the gather)
		if (rc == 0 && hipEventRecord(a, x->side) != hipSuccess) rc = -1;
		for (int k = 0; k < K && rc == 0; ++k) rc = r.AllGather(x->send[0].p, x->recv[0].p, x->record, NCCL_INT32, x->comm, x->side);
		float ms = 0;
		if (rc == 0 && (hipEventRecord(b, x->side) != hipSuccess || hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess)) rc = -1;
		(void)hipEventDestroy(a);
		(void)hipEventDestroy(b);
		if (rc != 0) return fail(ctx, LMX_ERR_HIP, "timing the all-gather failed");
		x->gather_us = 1e3 * ms / K;
		return LMX_OK;
	};
	int rc3 = LMX_OK;
	if (mode == "auto") {
		LMX_HIP(ctx, hipMemsetAsync(x->send[0].p, 0, max_record * sizeof(int32_t), x->side));
		rc3 = time_gathers();
		const char* thr = getenv("LMX_EXCHANGE_OVERLAP_US");
		const double threshold = thr ? atof(thr) : 16.0;
		x->mode = x->gather_us > threshold ? LmxExchange::SIDE : LmxExchange::INLINE;
		x->mode_why = x->mode == LmxExchange::SIDE ? "auto: one gather takes longer than the host work the side stream adds" : "auto: one gather is shorter than the host work the side stream adds";
	} else if (mode == "side") {
		x->mode = LmxExchange::SIDE; x->mode_why = "forced";
	} else if (mode == "inline") {
		x->mode = LmxExchange::INLINE; x->mode_why = "forced";
	} else if (mode == "p2p") {
		x->mode = LmxExchange::P2P; x->mode_why = "forced";
		rc3 = p2p_setup(x, max_record);
	} else {
		rc3 = fail(ctx, LMX_ERR_INVALID_ARGUMENT, "LMX_EXCHANGE_MODE=%s: auto, inline, side or p2p", mode.c_str());
	}
	if (rc3 != LMX_OK) {
		lmx_exchange_destroy(x);
		return rc3;
	}
	x->inline_gather = x->mode != LmxExchange::SIDE;
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
	(void)hipStreamSynchronize(x->side);
	if (x->inline_gather && x->ctx) (void)hipStreamSynchronize(x->ctx->stream); // (the gathers of this mode run there)
	for (int p = 0; p < P2P_MAX_RANKS; ++p)
		if (x->p2p.opened[p]) (void)hipIpcCloseMemHandle(x->p2p.opened[p]);
	if (x->p2p.block) (void)hipFree(x->p2p.block);
	if (x->p2p.h_error) (void)hipHostFree(x->p2p.h_error);
	if (x->comm) (void)rccl().CommDestroy(x->comm);
	for (int i = 0; i < 2; ++i) {
		if (x->culled[i]) (void)hipEventDestroy(x->culled[i]);
		if (x->gathered[i]) (void)hipEventDestroy(x->gathered[i]);
	}
	if (x->side) (void)hipStreamDestroy(x->side);
	delete x;
}

// One frame of this rank: cull the frame's n_frusta views (the reference culls its 4 shadow cascades + main view + light query per
// frame, pipeline.cpp:1036-1045, :1252-1258) over the entities this context holds, in one lmx_cull (result slot = view `slot`), pack
// every frustum's visible ids behind its 8 counts, and enqueue ONE all-gather of the whole record on the side stream:
//     rank record = n_frusta x [LMX_MAX_TYPES counts | cap / n_frusta ids, types packed back to back]
// Every rank must pass the same n_frusta. Returns the slot (0 / 1) to pass to lmx_exchange_wait / lmx_exchange_result; the slot's
// previous gather must have been waited for or is waited for here.
int lmx_exchange_cull_many(LmxExchange* x, const LmxShiftedFrustum* frusta, uint32_t n_frusta, uint8_t type, uint32_t* out_slot) {
	if (!x) return LMX_ERR_INVALID_ARGUMENT;
	LmxContext* ctx = x->ctx;
	LMX_CHECK_CTX(ctx);
	if (n_frusta < 1 || n_frusta > LMX_MAX_FRUSTA) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "n_frusta %u not in [1,%d]", n_frusta, LMX_MAX_FRUSTA);
	const uint32_t cap_f = x->cap / n_frusta;
	if (cap_f == 0) return fail(ctx, LMX_ERR_CAPACITY, "%u ids per rank cannot be split over %u frusta", x->cap, n_frusta);
	const uint32_t k = x->next;
	x->next ^= 1u;
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto lap = [&](int i, std::chrono::steady_clock::time_point& t) {
		if (!x->trace) return;
		const auto t1 = now();
		x->t_host[i] += std::chrono::duration<double, std::micro>(t1 - t).count();
		t = t1;
	};
	auto t = now();
	const bool p2p = x->mode == LmxExchange::P2P;
	uint32_t seq = 0;
	if (p2p) {
		if (x->p2p.failed) return fail(ctx, LMX_ERR_BUSY, "this exchange's P2P mode has failed before (a peer fell out of step): destroy it and create one in another mode");
		// this rank's slot k may be overwritten for sequence `seq` (the caller is done with its previous contents: the API contract) - said
		// first, so that the peers' stores never wait for this rank's cull
		seq = ++x->p2p.seq[k];
		hipLaunchKernelGGL(k_p2p_signal, dim3(1), dim3(64), 0, ctx->stream, x->p2p.targets, (uint32_t)x->world, p2p_flag_index(P2P_READY, k, (uint32_t)x->rank), seq);
	}
	// the send / recv buffers of this slot are free once its previous gather has finished: the cull stream waits for it (device-side)
	if (x->in_flight[k] && !x->inline_gather) LMX_HIP(ctx, hipStreamWaitEvent(ctx->stream, x->gathered[k], 0));
	lap(0, t);
	if (int rc = lmx_cull(ctx, k, frusta, n_frusta, type)) return rc;
	lap(1, t);
	CullState& cs = ctx->cull;
	CullView& v = cs.views[k];
	// per frustum: per-type totals straight into its sub-record's header and the ids behind it (clipped to cap_f), one launch each (k_cull_pack)
	const uint32_t sub = MAX_TYPES + cap_f;
	const uint32_t cnt_frustum_stride = cs.n_shards * cs.cnt_pad;
	// The collective is IN PLACE: this rank's record is packed where the gather would put it (recv + rank * record), so RCCL moves the
	// peers' records only - no local copy at any world size, nothing at all in a world of one (the one-rank step 19.3 -> the plain
	// step's 14.4 us + the call). The P2P form keeps its send buffer: its scatter writes every peer's slot, this rank's included.
	int32_t* const own = p2p ? x->send[k].p : x->recv[k].p + (size_t)x->rank * n_frusta * sub;
	{ // (one launch for all sub-records)
		int32_t* rec = own;
		LMX_HIP(ctx, launch_cull_pack(ctx->stream, v.out.p, cs.d_win_base.p, v.counts_ptr(), cs.cnt_pad, cs.d_shard_type.p, cs.n_shards, cs.max_shard_cap,
			reinterpret_cast<uint32_t*>(rec), rec + MAX_TYPES, cap_f, n_frusta, (uint32_t)v.out_stride, cnt_frustum_stride, sub));
	}
	x->n_frusta[k] = n_frusta;
	x->cap_f[k] = cap_f;
	x->record = n_frusta * sub;
	lap(2, t);
	if (p2p) {
		// every peer's slot k is free -> the used part of this rank's record into all of them -> "my record of sequence seq is there" ->
		// everybody's record of this sequence is here. Two bounded waits; a timeout is reported by lmx_exchange_wait.
		const uint32_t record = n_frusta * sub;
		hipLaunchKernelGGL(k_p2p_wait, dim3(1), dim3(64), 0, ctx->stream, x->p2p.my_flags, (uint32_t)x->world, p2p_flag_index(P2P_READY, k, 0), seq, x->p2p.timeout_ticks, x->p2p.d_error, 1u);
		hipLaunchKernelGGL(k_p2p_scatter, dim3((uint32_t)x->world, n_frusta, std::max(1u, std::min(64u, cap_f / 16384u))), dim3(256), 0, ctx->stream, x->send[k].p, x->p2p.targets,
			(uint32_t)((size_t)k * x->p2p.slot_words + (size_t)x->rank * record), sub, cap_f, x->p2p.d_error);
		hipLaunchKernelGGL(k_p2p_signal, dim3(1), dim3(64), 0, ctx->stream, x->p2p.targets, (uint32_t)x->world, p2p_flag_index(P2P_DATA, k, (uint32_t)x->rank), seq);
		hipLaunchKernelGGL(k_p2p_wait, dim3(1), dim3(64), 0, ctx->stream, x->p2p.my_flags, (uint32_t)x->world, p2p_flag_index(P2P_DATA, k, 0), seq, x->p2p.timeout_ticks, x->p2p.d_error, 2u);
		LMX_HIP(ctx, hipGetLastError());
		LMX_HIP(ctx, hipEventRecord(x->gathered[k], ctx->stream));
		x->in_flight[k] = true;
		if (out_slot) *out_slot = k;
		return LMX_OK;
	}
	hipStream_t gather_stream = x->side;
	if (x->inline_gather) {
		gather_stream = ctx->stream; // behind the pack kernel in stream order
	} else {
		LMX_HIP(ctx, hipEventRecord(x->culled[k], ctx->stream));
		lap(3, t);
		LMX_HIP(ctx, hipStreamWaitEvent(x->side, x->culled[k], 0));
		lap(4, t);
	}
	const int rc = rccl().AllGather(own, x->recv[k].p, (size_t)n_frusta * sub, NCCL_INT32, x->comm, gather_stream);
	if (rc != 0) return fail(ctx, LMX_ERR_HIP, "ncclAllGather failed: %s", rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
	lap(5, t);
	LMX_HIP(ctx, hipEventRecord(x->gathered[k], gather_stream));
	lap(6, t);
	x->t_steps += x->trace ? 1 : 0;
	x->in_flight[k] = true;
	if (out_slot) *out_slot = k;
	return LMX_OK;
}

int lmx_exchange_cull(LmxExchange* x, const LmxShiftedFrustum* frustum, uint8_t type, uint32_t* out_slot) { return lmx_exchange_cull_many(x, frustum, 1, type, out_slot); }

// Host wait for the gather of `slot` (a consumer on another stream can instead make that stream wait: lmx_exchange_result's event).
int lmx_exchange_wait(LmxExchange* x, uint32_t slot) {
	if (!x || slot > 1) return LMX_ERR_INVALID_ARGUMENT;
	LmxContext* ctx = x->ctx;
	LMX_CHECK_CTX(ctx);
	if (x->in_flight[slot]) LMX_HIP(ctx, hipEventSynchronize(x->gathered[slot]));
	if (x->mode == LmxExchange::P2P && *x->p2p.h_error != 0) {
		const uint32_t e = *x->p2p.h_error;
		x->p2p.failed = true;
		return fail(ctx, LMX_ERR_BUSY, "P2P exchange: rank %u's %s flag did not arrive within the bounded wait (a peer died or fell out of step); the frame is lost, the exchange unusable",
			(e >> 8) & 0xffu, (e & 0xffu) == 1u ? "slot-free" : "record-stored");
	}
	return LMX_OK;
}

// How this exchange runs its step and why: mode 0 = all-gather on the cull stream, 1 = on a side stream (overlaps the next cull), 2 = P2P stores;
// gather_us = one all-gather as timed at creation (< 0: not timed, the mode was forced).
int lmx_exchange_info(LmxExchange* x, int* mode, double* gather_us, const char** why) {
	if (!x) return LMX_ERR_INVALID_ARGUMENT;
	if (mode) *mode = x->mode;
	if (gather_us) *gather_us = x->gather_us;
	if (why) *why = x->mode_why;
	return LMX_OK;
}

// Device view of the gathered records of `slot`: rank r's record starts at d_records + r * record_words: LMX_MAX_TYPES counts
// (what the rank saw, also when it exceeds ids_per_rank: an overflow is visible as sum(counts) > ids_per_rank), then the ids,
// type 0 first. `gathered_event` (hipEvent_t as void*) is recorded when the collective has finished.
int lmx_exchange_result(LmxExchange* x, uint32_t slot, const int32_t** d_records, uint32_t* record_words, void** gathered_event) {
	if (!x || slot > 1) return LMX_ERR_INVALID_ARGUMENT;
	if (d_records) *d_records = recv_base(x, slot);
	if (record_words) *record_words = x->n_frusta[slot] * (MAX_TYPES + x->cap_f[slot]);
	if (gathered_event) *gathered_event = x->gathered[slot];
	return LMX_OK;
}

// Host copy of one (rank, frustum) sub-record of `slot`: counts[LMX_MAX_TYPES] and min(sum(counts), ids per frustum, cap) ids. Waits for
// the gather. (A clipped list is visible as sum(counts) > the slot's ids per frustum = ids_per_rank / n_frusta.)
int lmx_exchange_read_many(LmxExchange* x, uint32_t slot, int rank, uint32_t frustum, uint32_t* out_counts, int32_t* out_ids, uint32_t cap) {
	if (!x || slot > 1 || rank < 0 || rank >= x->world || !out_counts || frustum >= x->n_frusta[slot]) return LMX_ERR_INVALID_ARGUMENT;
	LmxContext* ctx = x->ctx;
	LMX_CHECK_CTX(ctx);
	if (int rc = lmx_exchange_wait(x, slot)) return rc;
	const uint32_t sub = MAX_TYPES + x->cap_f[slot];
	const int32_t* rec = recv_base(x, slot) + (size_t)rank * x->n_frusta[slot] * sub + (size_t)frustum * sub;
	LMX_HIP(ctx, hipMemcpy(out_counts, rec, sizeof(uint32_t) * MAX_TYPES, hipMemcpyDeviceToHost));
	uint64_t total = 0;
	for (int t = 0; t < MAX_TYPES; ++t) total += out_counts[t];
	const uint32_t n = (uint32_t)std::min<uint64_t>(total, std::min(x->cap_f[slot], cap));
	if (n && out_ids) LMX_HIP(ctx, hipMemcpy(out_ids, rec + MAX_TYPES, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
	return LMX_OK;
}

int lmx_exchange_read(LmxExchange* x, uint32_t slot, int rank, uint32_t* out_counts, int32_t* out_ids, uint32_t cap) {
	return lmx_exchange_read_many(x, slot, rank, 0, out_counts, out_ids, cap);
}

} // extern "C"
