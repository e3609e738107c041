// lmx_capi_exchange.hip — multi-GPU exchange of visible-entity lists behind the C ABI (include/lumix_mi355.h, "exchange").
//
// One process per GPU; every rank owns a disjoint set of entities (visibility depends only on the frustum and the
// entity's own cell, so the cull itself needs no communication). The only exchange step of a frame is ONE ncclAllGather (RCCL over
// xGMI) of a fixed-size record per rank: [8 per-type counts | cap visible ids, types packed back to back]. The cull's raw result is
// gathered straight into the send buffer by the finalize / consolidate kernels (no staging copy), the collective runs on a side
// stream ordered by events, and frames are double-buffered so that the next cull overlaps the previous frame's gather. No torch,
// no host wait in the steady state. RCCL is loaded with dlopen at the first lmx_exchange_* call: single-GPU users of the library
// do not need it.
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

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

struct LmxExchange {
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

extern "C" {

void lmx_exchange_destroy(LmxExchange* x);

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
	{ // (one launch for all sub-records)
		int32_t* rec = x->send[k].p;
		LMX_HIP(ctx, launch_cull_pack(ctx->stream, v.out.p, cs.d_win_base.p, v.counts_ptr(), cs.cnt_pad, cs.d_shard_type.p, cs.n_shards, cs.max_shard_cap,
			reinterpret_cast<uint32_t*>(rec), rec + MAX_TYPES, cap_f, n_frusta, (uint32_t)v.out_stride, cnt_frustum_stride, sub));
	}
	x->n_frusta[k] = n_frusta;
	x->cap_f[k] = cap_f;
	x->record = n_frusta * sub;
	lap(2, t);
	hipStream_t gather_stream = x->side;
	if (x->inline_gather) {
		gather_stream = ctx->stream; // behind the pack kernel in stream order
	} else {
		LMX_HIP(ctx, hipEventRecord(x->culled[k], ctx->stream));
		lap(3, t);
		LMX_HIP(ctx, hipStreamWaitEvent(x->side, x->culled[k], 0));
		lap(4, t);
	}
	const int rc = rccl().AllGather(x->send[k].p, x->recv[k].p, (size_t)n_frusta * sub, NCCL_INT32, x->comm, gather_stream);
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
	return LMX_OK;
}

// Device view of the gathered records of `slot`: rank r's record starts at d_records + r * record_words: LMX_MAX_TYPES counts
// (what the rank saw, also when it exceeds ids_per_rank: an overflow is visible as sum(counts) > ids_per_rank), then the ids,
// type 0 first. `gathered_event` (hipEvent_t as void*) is recorded when the collective has finished.
int lmx_exchange_result(LmxExchange* x, uint32_t slot, const int32_t** d_records, uint32_t* record_words, void** gathered_event) {
	if (!x || slot > 1) return LMX_ERR_INVALID_ARGUMENT;
	if (d_records) *d_records = x->recv[slot].p;
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
	const int32_t* rec = x->recv[slot].p + (size_t)rank * x->n_frusta[slot] * sub + (size_t)frustum * sub;
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
