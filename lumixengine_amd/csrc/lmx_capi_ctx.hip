// lmx_capi_ctx.hip — context lifetime, streams, per-kernel event profiling (include/lumix_mi355.h, "context" section).
// There is no CPU fallback: lmx_ctx_create refuses to run without a gfx950 device.
#include "lmx_context.h"

using namespace lmx;

namespace {
thread_local std::string g_create_error;
}

namespace lmx {

thread_local std::string* t_fail_sink = nullptr;

int fail(LmxContext* ctx, int code, const char* fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	if (t_fail_sink) *t_fail_sink = buf;
	else if (ctx) ctx->error = buf;
	else g_create_error = buf;
	return code;
}

void prof_drain(LmxContext* ctx) {
	for (ProfSlot& s : ctx->prof_pending) {
		float ms = 0.f;
		if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
			ctx->prof_ms[s.kernel] += ms;
			ctx->prof_launches[s.kernel] += 1;
		}
		ctx->event_pool.push_back(s.a);
		ctx->event_pool.push_back(s.b);
	}
	ctx->prof_pending.clear();
	// keep a small pool only: hundreds of live events slow the HIP runtime's queue management down (seen as ~0.7 ms per launch
	// once the GPU, not the host, paces a launch loop)
	while (ctx->event_pool.size() > 64) {
		(void)hipEventDestroy(ctx->event_pool.back());
		ctx->event_pool.pop_back();
	}
}

} // namespace lmx

extern "C" {

int lmx_ctx_create(int device, LmxContext** out) {
	if (!out) return LMX_ERR_INVALID_ARGUMENT;
	*out = nullptr;
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0)
		return fail(nullptr, LMX_ERR_NO_DEVICE, "no HIP device available (%s); liblumix_mi355 has no CPU fallback",
			e == hipSuccess ? "device count 0" : hipGetErrorString(e));
	if (device < 0 || device >= count) return fail(nullptr, LMX_ERR_INVALID_ARGUMENT, "device %d out of range [0,%d)", device, count);
	e = hipSetDevice(device);
	if (e != hipSuccess) return fail(nullptr, LMX_ERR_NO_DEVICE, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
	hipDeviceProp_t prop;
	e = hipGetDeviceProperties(&prop, device);
	if (e != hipSuccess) return fail(nullptr, LMX_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(nullptr, LMX_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code objects only", device, prop.gcnArchName);
	LmxContext* ctx = new LmxContext;
	ctx->device = device;
	e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
	if (e != hipSuccess) {
		delete ctx;
		return fail(nullptr, LMX_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
	}
	ctx->stream = ctx->own_stream;
	*out = ctx;
	return LMX_OK;
}

void lmx_ctx_destroy(LmxContext* ctx) {
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	cull_async_shutdown(ctx); // joins the asynchronous compaction's worker, if one runs
	(void)hipStreamSynchronize(ctx->stream);
	prof_drain(ctx);
	for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
	if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
	delete ctx;
}

// ---- one context per World (registry keyed by the World's address) ------------------------------------------------------------
namespace {
struct SharedEntry { const void* key; int device; LmxContext* ctx; int refs; };
std::mutex g_shared_mutex;
std::vector<SharedEntry> g_shared;
} // namespace

int lmx_ctx_acquire_shared(const void* key, int device, LmxContext** out) {
	if (!out || !key) return LMX_ERR_INVALID_ARGUMENT;
	*out = nullptr;
	std::lock_guard<std::mutex> guard(g_shared_mutex);
	for (SharedEntry& e : g_shared) {
		if (e.key == key && e.device == device) {
			++e.refs;
			*out = e.ctx;
			return LMX_OK;
		}
	}
	LmxContext* ctx = nullptr;
	if (int rc = lmx_ctx_create(device, &ctx)) return rc;
	g_shared.push_back(SharedEntry{key, device, ctx, 1});
	*out = ctx;
	return LMX_OK;
}

void lmx_ctx_release_shared(LmxContext* ctx) {
	if (!ctx) return;
	{
		std::lock_guard<std::mutex> guard(g_shared_mutex);
		for (size_t i = 0; i < g_shared.size(); ++i) {
			if (g_shared[i].ctx != ctx) continue;
			if (--g_shared[i].refs > 0) return;
			g_shared.erase(g_shared.begin() + i);
			break;
		}
	}
	lmx_ctx_destroy(ctx); // last reference (or a context that never was in the registry)
}

void lmx_ctx_lock(LmxContext* ctx) {
	if (ctx) ctx->lock.lock();
}
void lmx_ctx_unlock(LmxContext* ctx) {
	if (ctx) ctx->lock.unlock();
}

const char* lmx_last_error(const LmxContext* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int lmx_ctx_set_stream(LmxContext* ctx, void* hip_stream) {
	LMX_CHECK_CTX(ctx);
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	ctx->stream = (hipStream_t)hip_stream; // NULL is the legacy default (null) stream, as everywhere in HIP
	return LMX_OK;
}

int lmx_ctx_synchronize(LmxContext* ctx) {
	LMX_CHECK_CTX(ctx);
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

int lmx_profile_enable(LmxContext* ctx, int enable) {
	LMX_CHECK_CTX(ctx);
	ctx->profiling = enable != 0;
	return LMX_OK;
}

int lmx_profile_reset(LmxContext* ctx) {
	LMX_CHECK_CTX(ctx);
	prof_drain(ctx);
	for (int k = 0; k < LMX_K_COUNT; ++k) {
		ctx->prof_ms[k] = 0;
		ctx->prof_launches[k] = 0;
	}
	return LMX_OK;
}

int lmx_profile_get(LmxContext* ctx, int kernel_id, double* total_ms, uint64_t* launches) {
	LMX_CHECK_CTX(ctx);
	if (kernel_id < 0 || kernel_id >= LMX_K_COUNT) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "kernel id %d", kernel_id);
	prof_drain(ctx);
	if (total_ms) *total_ms = ctx->prof_ms[kernel_id];
	if (launches) *launches = ctx->prof_launches[kernel_id];
	return LMX_OK;
}

const char* lmx_version(void) { return "lumix-mi355 0.1 (gfx950)"; }

} // extern "C"
