// hip/hip_runtime.h of tests/hostsim — TEST INFRASTRUCTURE, not a product path and not a fallback.
//
// The development container has no GPU. This header and hostsim_runtime.cpp let the UNCHANGED kernel and C-ABI sources of
// lumixengine_amd/csrc be compiled for x86 (amdclang++ as a plain host compiler, -ffp-contract=off) into
// tests/_build/liblumix_hostsim.so, where a kernel launch executes the kernel's own source once per lane: every lane is a
// fiber, a block is run by one scheduler, and the wave64 cross-lane operations (ballot, readlane, readfirstlane, DPP, shuffles,
// wave barriers) and __syncthreads() are rendezvous points of the fibers (hostsim_runtime.cpp). The simulated device calls
// itself "gfx950:hostsim". What it is for:
//   * the parity tests of tests/ (-m gpu) can be run against the kernel SOURCES on a CPU, under AddressSanitizer /
//     UndefinedBehaviorSanitizer / ThreadSanitizer (device memory is plain heap memory: out-of-bounds accesses of a kernel are
//     reported instead of silently reading a neighbour's bytes);
//   * a kernel change can be checked for bit-exactness against the oracle before a GPU call is spent on it.
// What it is NOT: it says nothing about speed, and it does not model the memory system, LDS bank conflicts or the scheduler.
// Nothing under lumixengine_amd/ references this directory; lumixengine_amd/build.py never builds it; the library the product
// loads (liblumix_mi355.so) contains gfx950 code objects only.
#pragma once

#define LMX_HOSTSIM 1
#ifndef __HIPCC__
#define __HIPCC__ 1
#endif

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <tuple>
#include <type_traits>
#include <utility>

// ---- language ---------------------------------------------------------------------------------------------------------------
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define amdgpu_waves_per_eu(...) /* __attribute__((amdgpu_waves_per_eu(n))) of a kernel: an empty attribute here */
#define LMX_ASM_SGPR(x) "+r"(x) /* the "+s" (scalar register) operand of an empty asm: any general register here */
#define __shared__ static thread_local
// `extern __shared__ T name[];` (dynamic LDS) is spelled LMX_DYNAMIC_LDS(T, name) in the kernels; here it is a pointer to the
// launch's dynamic LDS block
#define LMX_DYNAMIC_LDS(T, name) T* const name = reinterpret_cast<T*>(::hostsim::tls.dyn_lds)

struct dim3 {
	uint32_t x, y, z;
	constexpr dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(16) double2 { double x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

// ---- runtime types ----------------------------------------------------------------------------------------------------------
typedef enum hipError_t {
	hipSuccess = 0,
	hipErrorInvalidValue = 1,
	hipErrorOutOfMemory = 2,
	hipErrorNoDevice = 100,
	hipErrorInvalidDevice = 101,
	hipErrorNotReady = 600,
	hipErrorUnknown = 999,
} hipError_t;
typedef struct ihipStream_t* hipStream_t;
typedef struct ihipEvent_t* hipEvent_t;
typedef enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum { hipEventDefault = 0, hipEventBlockingSync = 1, hipEventDisableTiming = 2 };
enum { hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipHostMallocMapped = 2, hipHostMallocWriteCombined = 4 };
struct hipDeviceProp_t {
	char name[256];
	char gcnArchName[256];
	size_t totalGlobalMem;
	int multiProcessorCount, warpSize, maxThreadsPerBlock;
	size_t sharedMemPerBlock;
	int clockRate, major, minor;
};

#pragma GCC visibility push(default) // exported: test doubles built against this header (tests/cpp/loopback_rccl.cpp) call them
extern "C" {
hipError_t hipGetDeviceCount(int* count);
hipError_t hipSetDevice(int device);
hipError_t hipGetDevice(int* device);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int device);
hipError_t hipGetLastError(void);
const char* hipGetErrorString(hipError_t e);
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned flags);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind);
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemset(void* dst, int value, size_t bytes);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipDeviceSynchronize(void);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
}
// Inter-process mappings do not exist on the simulated device: the calls fail, and with them LMX_EXCHANGE_MODE=p2p - loudly, at creation.
struct hipIpcMemHandle_t { char reserved[64]; };
enum { hipIpcMemLazyEnablePeerAccess = 1, hipDeviceMallocFinegrained = 1 };
static inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t*, void*) { return hipErrorUnknown; }
static inline hipError_t hipIpcOpenMemHandle(void**, hipIpcMemHandle_t, unsigned) { return hipErrorUnknown; }
static inline hipError_t hipIpcCloseMemHandle(void*) { return hipErrorUnknown; }
static inline hipError_t hipExtMallocWithFlags(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
#pragma GCC visibility pop
// the C++ overloads of the HIP headers
template <typename T> static inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc(reinterpret_cast<void**>(p), bytes); }
template <typename T> static inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned flags = hipHostMallocDefault) { return hipHostMalloc(reinterpret_cast<void**>(p), bytes, flags); }
template <typename T> static inline hipError_t hipHostGetDevicePointer(T** dev, void* host, unsigned flags) { return hipHostGetDevicePointer(reinterpret_cast<void**>(dev), host, flags); }
static inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e) { return hipStreamWaitEvent(s, e, 0); }

// ---- the simulated device -----------------------------------------------------------------------------------------------------
namespace hostsim {

struct Idx3 { uint32_t x, y, z; };
struct LaneIds { Idx3 tid; uint32_t lane, wave; };
// what a lane sees after a wave rendezvous: the values every lane of its convergence group posted
struct WaveSnapshot {
	uint64_t exec;    // lanes of the group (the EXEC mask of the operation)
	uint64_t nonzero; // lanes of the group that posted a non-zero value
	uint32_t first;   // lowest lane of the group
	uint64_t val[64]; // posted values (stale for lanes outside `exec`, as a VGPR of an inactive lane would be)
};
struct Tls {
	const LaneIds* ids;
	Idx3 bid, bdim, gdim;
	const void* kernarg;
	void* dyn_lds;
};
extern thread_local Tls tls;

// rendezvous of the calling lane with the other lanes of its wave that reach the same code location; returns the group's values
__attribute__((noinline, convergent)) const WaveSnapshot* wave_exchange(uint64_t value);
__attribute__((noinline, convergent)) void block_barrier();
// the LDS word a 32-bit "LDS address" (a truncated pointer to a __shared__ object) refers to
void* lds_pointer(uint32_t lds_byte_address);

typedef void (*LaneEntry)(void* closure);
void run_grid(const char* label, dim3 grid, dim3 block, size_t dyn_lds_bytes, const void* kernarg, LaneEntry entry, void* closure);
void note_launch_error(hipError_t e);

template <typename T> static inline void pack_kernarg(unsigned char* buf, size_t& off, size_t cap, const T& v) {
	const size_t a = alignof(T) > 16 ? 16 : alignof(T);
	off = (off + a - 1) / a * a;
	if (off + sizeof(T) <= cap) memcpy(buf + off, &v, sizeof(T));
	off += sizeof(T);
}

template <typename... KArgs, typename... Args>
static inline void launch(const char* label, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t dyn_lds_bytes, hipStream_t, Args&&... args) {
	static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel launched with a different number of arguments than it declares");
	typedef std::tuple<std::decay_t<KArgs>...> Tuple;
	struct Closure { void (*kernel)(KArgs...); Tuple args; };
	Closure c{kernel, Tuple{static_cast<std::decay_t<KArgs>>(std::forward<Args>(args))...}};
	alignas(16) unsigned char kernarg[4096]; // the kernarg segment as the AMDGPU ABI lays it out: arguments in order, naturally aligned
	size_t off = 0;
	std::apply([&](const auto&... a) { (pack_kernarg(kernarg, off, sizeof(kernarg), a), ...); }, c.args);
	if (off > sizeof(kernarg)) { note_launch_error(hipErrorInvalidValue); return; }
	run_grid(label, grid, block, dyn_lds_bytes, kernarg, [](void* p) {
		Closure* cl = static_cast<Closure*>(p);
		std::apply(cl->kernel, cl->args);
	}, &c);
}

} // namespace hostsim

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) ::hostsim::launch(#kernel, kernel, grid, block, lds, stream, __VA_ARGS__)

#define threadIdx (::hostsim::tls.ids->tid)
#define blockIdx (::hostsim::tls.bid)
#define blockDim (::hostsim::tls.bdim)
#define gridDim (::hostsim::tls.gdim)
static constexpr int warpSize = 64;

// ---- device intrinsics ------------------------------------------------------------------------------------------------------
static inline void __syncthreads() { ::hostsim::block_barrier(); }
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline float __uint_as_float(unsigned v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
static inline unsigned __float_as_uint(float f) { unsigned v; memcpy(&v, &f, 4); return v; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }
static inline float __fdividef(float a, float b) { return a / b; }

// overloads the HIP headers put into the global namespace
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
static inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
static inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
static inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }

// atomics (blocks may run on several host threads: real atomics)
template <typename T> static inline T atomicAdd(T* p, T v) {
	if constexpr (std::is_floating_point<T>::value) {
		typedef typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type U;
		U* up = reinterpret_cast<U*>(p);
		U cur = __atomic_load_n(up, __ATOMIC_RELAXED);
		for (;;) {
			T f; memcpy(&f, &cur, sizeof(T));
			const T n = f + v;
			U nu; memcpy(&nu, &n, sizeof(T));
			if (__atomic_compare_exchange_n(up, &cur, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return f;
		}
	} else {
		return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
	}
}
template <typename T> static inline T atomicSub(T* p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicAnd(T* p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicXor(T* p, T v) { return __atomic_fetch_xor(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicCAS(T* p, T expect, T v) { __atomic_compare_exchange_n(p, &expect, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return expect; }
template <typename T> static inline T atomicMax(T* p, T v) {
	T cur = __atomic_load_n(p, __ATOMIC_RELAXED);
	while (cur < v && !__atomic_compare_exchange_n(p, &cur, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
	return cur;
}
template <typename T> static inline T atomicMin(T* p, T v) {
	T cur = __atomic_load_n(p, __ATOMIC_RELAXED);
	while (cur > v && !__atomic_compare_exchange_n(p, &cur, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
	return cur;
}
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
static inline unsigned long long wall_clock64() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (unsigned long long)ts.tv_sec * 100000000ull + (unsigned long long)ts.tv_nsec / 10ull; } // 100 MHz, as the device's
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ---- wave64 cross-lane operations ---------------------------------------------------------------------------------------------
namespace hostsim {
static __forceinline__ uint32_t lane() { return tls.ids->lane; }
template <typename T> static __forceinline__ uint64_t to_bits(T v) {
	static_assert(sizeof(T) <= 8, "cross-lane value wider than 64 bits");
	uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b;
}
template <typename T> static __forceinline__ T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }
template <typename T> static __forceinline__ T read_lane(T v, uint32_t src) { return from_bits<T>(wave_exchange(to_bits(v))->val[src & 63u]); }
template <typename T> static __forceinline__ T shfl_from(T v, int src, bool valid) {
	const WaveSnapshot* s = wave_exchange(to_bits(v));
	// a source lane outside the group: the HIP shuffles return the lane's own value for out-of-range sources; an inactive source lane
	// yields whatever its register holds (here: what it posted last)
	return valid ? from_bits<T>(s->val[(uint32_t)src & 63u]) : v;
}
static __forceinline__ uint32_t mbcnt(uint32_t mask, uint32_t add, bool hi) {
	const uint32_t l = lane();
	uint32_t below;
	if (!hi) below = l >= 32 ? 0xffffffffu : ((1u << l) - 1u);
	else below = l <= 32 ? 0u : ((1u << (l - 32)) - 1u);
	return add + (uint32_t)__builtin_popcount(mask & below);
}
// DPP quad_perm / row shifts as used by the kernels (bound_ctrl = true: lanes without a source read 0)
__attribute__((noinline, convergent)) int mov_dpp(int v, int ctrl, int row_mask, int bank_mask, bool bound_ctrl);
} // namespace hostsim

static __forceinline__ unsigned long long __ballot(int pred) { return ::hostsim::wave_exchange(pred ? 1u : 0u)->nonzero; }
static __forceinline__ unsigned long long __activemask() { return ::hostsim::wave_exchange(1u)->exec; }
static __forceinline__ int __any(int pred) { return ::hostsim::wave_exchange(pred ? 1u : 0u)->nonzero != 0; }
static __forceinline__ int __all(int pred) { const ::hostsim::WaveSnapshot* s = ::hostsim::wave_exchange(pred ? 1u : 0u); return s->nonzero == s->exec; }
template <typename T> static __forceinline__ T __shfl(T v, int src, int width = 64) {
	const int l = (int)::hostsim::lane();
	const int s = (l & ~(width - 1)) + (src & (width - 1));
	return ::hostsim::shfl_from(v, s, true);
}
template <typename T> static __forceinline__ T __shfl_up(T v, unsigned delta, int width = 64) {
	const int l = (int)::hostsim::lane();
	const int s = l - (int)delta;
	return ::hostsim::shfl_from(v, s, s >= (l & ~(width - 1)));
}
template <typename T> static __forceinline__ T __shfl_down(T v, unsigned delta, int width = 64) {
	const int l = (int)::hostsim::lane();
	const int s = l + (int)delta;
	return ::hostsim::shfl_from(v, s, s < (l & ~(width - 1)) + width);
}
template <typename T> static __forceinline__ T __shfl_xor(T v, int mask, int width = 64) {
	const int l = (int)::hostsim::lane();
	const int s = l ^ mask;
	return ::hostsim::shfl_from(v, s, s < (l & ~(width - 1)) + width);
}

// v_mfma_f32_32x32x2_f32: bit for bit the k-ordered chain D = fmaf(A[i][1], B[1][j], fmaf(A[i][0], B[0][j], C)) with the operand maps
// A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31], D[row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)][col = lane & 31] -
// checked on the MI355X by tools/mfma_contract_probe.hip (subnormal, infinite and NaN operands included). All 64 lanes take part.
// v_permlane32_swap_b32 vdst, src: lanes 32..63 of vdst trade places with lanes 0..31 of src; the builtin returns {new vdst, new src}.
namespace hostsim {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
static __forceinline__ f32x16 mfma_f32_32x32x2f32(float a, float b, f32x16 c) {
	float av[64], bv[64];
	{
		const WaveSnapshot* s = wave_exchange(to_bits(a));
		for (int l = 0; l < 64; ++l) av[l] = from_bits<float>(s->val[l]);
	}
	{
		const WaveSnapshot* s = wave_exchange(to_bits(b));
		for (int l = 0; l < 64; ++l) bv[l] = from_bits<float>(s->val[l]);
	}
	const uint32_t l = lane(), col = l & 31u;
	f32x16 d;
	for (int r = 0; r < 16; ++r) {
		const uint32_t row = (uint32_t)(r & 3) + 8u * (uint32_t)(r >> 2) + 4u * (l >> 5);
		d[r] = fmaf(av[row + 32], bv[col + 32], fmaf(av[row], bv[col], c[r]));
	}
	return d;
}
// v_mfma_f32_32x32x16_bf16: A[i = lane & 31][k = 8 (lane >> 5) + e], B[k = 8 (lane >> 5) + e][j = lane & 31] (element e of the 8-vector), the C / D
// map of the f32 form. Modelled as the exact sum C + sum a b rounded once; the hardware's accumulation differs from that by up to ~4.5 u of
// the magnitude sum (tools/mfma_contract_probe.hip) - code built on it must not depend on the last bits, and the cull pre-test does not.
template <typename V8> static __forceinline__ f32x16 mfma_f32_32x32x16_bf16(V8 a, V8 b, f32x16 c) {
	static_assert(sizeof(V8) == 16, "eight bf16 per lane");
	uint64_t w[4];
	memcpy(w, &a, 16);
	memcpy(w + 2, &b, 16);
	uint16_t av[64][8], bv[64][8];
	for (int part = 0; part < 4; ++part) {
		const WaveSnapshot* s = wave_exchange(w[part]);
		for (int l = 0; l < 64; ++l) memcpy((part < 2 ? av[l] : bv[l]) + 4 * (part & 1), &s->val[l], 8);
	}
	const uint32_t l = lane(), col = l & 31u;
	f32x16 d;
	for (int r = 0; r < 16; ++r) {
		const uint32_t row = (uint32_t)(r & 3) + 8u * (uint32_t)(r >> 2) + 4u * (l >> 5);
		double acc = c[r];
		for (int k = 0; k < 16; ++k) {
			const uint32_t ab = (uint32_t)av[row + 32 * (k >> 3)][k & 7] << 16, bb = (uint32_t)bv[col + 32 * (k >> 3)][k & 7] << 16;
			acc += (double)from_bits<float>(ab) * (double)from_bits<float>(bb);
		}
		d[r] = (float)acc;
	}
	return d;
}
static __forceinline__ u32x2 permlane32_swap(unsigned a, unsigned b) {
	unsigned av[64], bv[64];
	{
		const WaveSnapshot* s = wave_exchange((uint64_t)a);
		for (int l = 0; l < 64; ++l) av[l] = (unsigned)s->val[l];
	}
	{
		const WaveSnapshot* s = wave_exchange((uint64_t)b);
		for (int l = 0; l < 64; ++l) bv[l] = (unsigned)s->val[l];
	}
	const uint32_t l = lane();
	u32x2 r;
	r[0] = l < 32 ? av[l] : bv[l - 32];
	r[1] = l < 32 ? av[l + 32] : bv[l];
	return r;
}
} // namespace hostsim
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, cbsz, abid, blgp) ::hostsim::mfma_f32_32x32x2f32((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, cbsz, abid, blgp) ::hostsim::mfma_f32_32x32x16_bf16((a), (b), (c))
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) ::hostsim::permlane32_swap((unsigned)(a), (unsigned)(b))
#define __builtin_amdgcn_mbcnt_lo(mask, add) ::hostsim::mbcnt((mask), (add), false)
#define __builtin_amdgcn_mbcnt_hi(mask, add) ::hostsim::mbcnt((mask), (add), true)
#define __builtin_amdgcn_readfirstlane(v) ([&](auto hostsim_v) __attribute__((always_inline)) { const ::hostsim::WaveSnapshot* hostsim_s = ::hostsim::wave_exchange(::hostsim::to_bits(hostsim_v)); return ::hostsim::from_bits<decltype(hostsim_v)>(hostsim_s->val[hostsim_s->first]); }(v))
#define __builtin_amdgcn_readlane(v, l) ::hostsim::read_lane((v), (uint32_t)(l))
#define __builtin_amdgcn_mov_dpp(v, ctrl, row_mask, bank_mask, bound_ctrl) ::hostsim::mov_dpp((v), (ctrl), (row_mask), (bank_mask), (bound_ctrl))
#define __builtin_amdgcn_wave_barrier() ((void)::hostsim::wave_exchange(0))
#define __builtin_amdgcn_fence(order, scope) __atomic_signal_fence(__ATOMIC_SEQ_CST)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#define __builtin_amdgcn_s_waitcnt(imm) ((void)0)
#define __builtin_amdgcn_s_sleep(imm) ((void)0)
#define __builtin_amdgcn_kernarg_segment_ptr() (::hostsim::tls.kernarg)
// Cache hints mean nothing here. A 3-element vector is stored element by element: clang widens vec3 accesses to vec4 on x86
// (16 bytes written), while the AMDGPU target keeps `store <3 x float>` (global_store_dwordx3, 12 bytes).
namespace hostsim {
template <typename T> struct NtAccess {
	static __forceinline__ void store(const T& v, void* p) { memcpy(p, &v, sizeof(T)); }
	static __forceinline__ T load(const void* p) { T v; memcpy(&v, p, sizeof(T)); return v; }
};
template <typename E, int N> struct NtAccess<E __attribute__((ext_vector_type(N)))> { // element by element: exactly N * sizeof(E) bytes
	typedef E V __attribute__((ext_vector_type(N)));
	typedef E Unaligned __attribute__((aligned(1), may_alias));
	static __forceinline__ void store(const V& v, void* p) {
		Unaligned* q = static_cast<Unaligned*>(p);
		for (int i = 0; i < N; ++i) q[i] = v[i];
	}
	static __forceinline__ V load(const void* p) {
		const Unaligned* q = static_cast<const Unaligned*>(p);
		V v;
		for (int i = 0; i < N; ++i) v[i] = q[i];
		return v;
	}
};
template <typename V> static __forceinline__ void nontemporal_store(const V& v, void* p) { NtAccess<V>::store(v, p); }
template <typename P> static __forceinline__ P nontemporal_load(const void* p) { return NtAccess<P>::load(p); }
} // namespace hostsim
#define __builtin_nontemporal_store(value, pointer) ::hostsim::nontemporal_store((value), static_cast<void*>(pointer))
#define __builtin_nontemporal_load(pointer) ::hostsim::nontemporal_load<std::remove_cv_t<std::remove_pointer_t<decltype(pointer)>>>(static_cast<const void*>(pointer))
