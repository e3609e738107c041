// hostsim_runtime.cpp — TEST INFRASTRUCTURE (see include/hip/hip_runtime.h): the simulated device behind the HIP entry points
// the C ABI uses, and the lane scheduler that executes a kernel's source once per lane.
//
// Execution model
//   * one block at a time per host thread; each lane of the block is a fiber with its own stack;
//   * a lane runs until it finishes or reaches a rendezvous: a wave-level one (wave_exchange: every cross-lane operation and the
//     wave barrier) or the block barrier;
//   * when every live lane of the block waits, the scheduler resolves, per wave, ONE convergence group: the lanes waiting at the
//     lowest code address (the return address of wave_exchange inside the kernel: the operation's own location, because all
//     wrappers are always_inline). Lanes of a wave that wait at different locations are on divergent paths; with the code layout
//     compilers produce for structured control flow the lower address is the path that has not reconverged yet, so it goes first
//     and the lanes further down keep waiting, as a wave's reconvergence stack would make them. The group's lanes form the
//     operation's EXEC mask. wave_exchange is declared `convergent`, so the optimiser may not duplicate or sink the calls;
//   * the block barrier opens when no wave group is pending and every live lane waits at it.
// Streams are synchronous: an operation is complete when its call returns (a legal, maximally ordered schedule).
// Device and pinned memory are heap memory.
//
// Environment
//   HOSTSIM_THREADS=n   blocks of a launch spread over n host threads (default 1; every thread has its own lane stacks and its own copy
//                       of the kernels' __shared__ objects, which are thread-local storage here)
//   HOSTSIM_ORDER=...   forward (default) | reverse | shuffle[:seed]: order in which runnable lanes are resumed and blocks are started
//   HOSTSIM_RACE=1      ThreadSanitizer build only: kernel-level race detection (below)
//   HOSTSIM_STACK_KB=n  stack of one lane (default 128, 512 under AddressSanitizer)
//   HOSTSIM_DEBUG=1     ThreadSanitizer build only: one line per block
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HOSTSIM_ASAN 1
#include <sanitizer/common_interface_defs.h>
#endif
#endif
#if defined(HOSTSIM_WITH_TSAN)
// (this file itself is compiled WITHOUT -fsanitize=thread, with -DHOSTSIM_WITH_TSAN: the scheduler's own bookkeeping is touched from
// every fiber and is not the subject)
// ThreadSanitizer: every lane is announced as a fiber of the host thread; switches synchronise (lanes of a block run one after the
// other on one thread), so what is reported are races between HOST threads - the asynchronous compaction's worker against the frame
#define HOSTSIM_TSAN 1
extern "C" {
void* __tsan_get_current_fiber(void);
void* __tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void* fiber);
void __tsan_switch_to_fiber(void* fiber, unsigned flags);
void __tsan_acquire(void* addr);
void __tsan_release(void* addr);
}
// HOSTSIM_RACE=1 (with --sanitize thread): kernel-level race detection. Every WAVE of a block is one ThreadSanitizer fiber (its lanes run
// in lockstep on the hardware: they are one thread of execution), switches between waves do NOT synchronise, and the only
// happens-before edges are the ones the hardware gives: launch -> every wave, __syncthreads() between the waves of a block, every wave ->
// the host after the launch. Waves of different blocks are never ordered. A conflicting pair of plain accesses to LDS or global memory
// that the kernel does not order by a barrier or make atomic is then reported as a data race - on the GPU it would be one.
#endif

asm(R"(
.text
.globl hostsim_switch
.hidden hostsim_switch
.type hostsim_switch,@function
hostsim_switch:
	pushq %rbp
	pushq %rbx
	pushq %r12
	pushq %r13
	pushq %r14
	pushq %r15
	movq %rsp, (%rdi)
	movq %rsi, %rsp
	popq %r15
	popq %r14
	popq %r13
	popq %r12
	popq %rbx
	popq %rbp
	ret
.size hostsim_switch,.-hostsim_switch
)");
extern "C" void hostsim_switch(void** save_sp, void* load_sp);

namespace hostsim {

thread_local Tls tls;

#ifdef HOSTSIM_TRAFFIC // build mode "traffic": traffic_runtime.cpp records the kernels' accesses to device memory
struct TrafficBlock;
extern thread_local TrafficBlock* t_traffic;
TrafficBlock* traffic_block_new();
void traffic_alloc(void* p, size_t n);
void traffic_free(void* p);
void traffic_flush_block(TrafficBlock* b, void* launch_acc, uint32_t block_threads_x);
void* traffic_launch_begin();
void traffic_launch_end(const char* label, void* launch_acc, uint64_t lanes);
#endif

namespace {

constexpr uint32_t MAX_LANES = 1024;
enum LaneState : uint8_t { RUNNABLE, WAIT_WAVE, WAIT_BLOCK, DONE };

struct Fiber {
	void* sp;
	LaneIds ids;
	LaneState state;
	const void* site;
	uint64_t post;
#ifdef HOSTSIM_ASAN
	void* fake_stack;
#endif
#ifdef HOSTSIM_TSAN
	void* tsan_fiber;
#endif
};

struct Worker {
	void* sched_sp = nullptr;
	char* stacks = nullptr;
	size_t stack_bytes = 0;
	Fiber* current = nullptr;
	LaneEntry entry = nullptr;
	void* closure = nullptr;
	std::vector<unsigned char> dyn_lds;
#ifdef HOSTSIM_TRAFFIC
	TrafficBlock* traffic = nullptr;
#endif
	Fiber fibers[MAX_LANES];
	WaveSnapshot snap[MAX_LANES / 64];
#ifdef HOSTSIM_ASAN
	void* sched_fake = nullptr;
	const void* sched_bottom = nullptr;
	size_t sched_size = 0;
#endif
#ifdef HOSTSIM_TSAN
	void* sched_tsan = nullptr;
	void* wave_tsan[MAX_LANES / 64] = {}; // this host thread's wave fibers (live as long as the worker)
	bool race_mode = false;
	char launch_token = 0, done_token = 0, barrier_token = 0; // addresses for __tsan_release / __tsan_acquire
#endif
	~Worker() {
		if (stacks) munmap(stacks, stack_bytes * MAX_LANES);
	}
};

thread_local Worker* t_worker = nullptr;
thread_local hipError_t t_last_error = hipSuccess;
thread_local char t_lds_anchor;

size_t env_size(const char* name, size_t dflt) {
	const char* v = getenv(name);
	if (!v || !*v) return dflt;
	const long long n = atoll(v);
	return n > 0 ? (size_t)n : dflt;
}

Worker* worker() {
	if (t_worker) return t_worker;
	static thread_local struct Holder { Worker* w = nullptr; ~Holder() { delete w; } } holder;
	Worker* w = new Worker;
#ifdef HOSTSIM_ASAN
	w->stack_bytes = env_size("HOSTSIM_STACK_KB", 512) * 1024;
#else
	w->stack_bytes = env_size("HOSTSIM_STACK_KB", 128) * 1024;
#endif
	void* p = mmap(nullptr, w->stack_bytes * MAX_LANES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
	if (p == MAP_FAILED) {
		fprintf(stderr, "hostsim: cannot map %zu bytes of lane stacks\n", w->stack_bytes * MAX_LANES);
		abort();
	}
	w->stacks = static_cast<char*>(p);
#ifdef HOSTSIM_TSAN
	w->race_mode = env_size("HOSTSIM_RACE", 0) != 0;
#endif
	holder.w = w;
	t_worker = w;
	return w;
}

[[noreturn]] void fiber_main() {
	Worker* w = t_worker;
#ifdef HOSTSIM_ASAN
	__sanitizer_finish_switch_fiber(nullptr, &w->sched_bottom, &w->sched_size);
#endif
#ifdef HOSTSIM_TSAN
	if (w->race_mode) __tsan_acquire(&w->launch_token);
#endif
	w->entry(w->closure);
	Fiber* f = w->current;
	f->state = DONE;
#ifdef HOSTSIM_TSAN
	if (w->race_mode) __tsan_release(&w->done_token);
#endif
#ifdef HOSTSIM_ASAN
	__sanitizer_start_switch_fiber(nullptr, w->sched_bottom, w->sched_size); // this fiber's fake stack is released
#endif
#ifdef HOSTSIM_TSAN
	__tsan_switch_to_fiber(w->sched_tsan, w->race_mode ? 1u : 0u);
#endif
	hostsim_switch(&f->sp, w->sched_sp);
	__builtin_unreachable();
}

void yield_to_scheduler() {
	Worker* w = t_worker;
	Fiber* f = w->current;
#ifdef HOSTSIM_ASAN
	__sanitizer_start_switch_fiber(&f->fake_stack, w->sched_bottom, w->sched_size);
#endif
#ifdef HOSTSIM_TSAN
	__tsan_switch_to_fiber(w->sched_tsan, w->race_mode ? 1u : 0u);
#endif
	hostsim_switch(&f->sp, w->sched_sp);
#ifdef HOSTSIM_ASAN
	__sanitizer_finish_switch_fiber(f->fake_stack, &w->sched_bottom, &w->sched_size);
#endif
}

void resume(Worker* w, Fiber* f) {
	w->current = f;
	tls.ids = &f->ids;
#ifdef HOSTSIM_ASAN
	const size_t index = (size_t)(f - w->fibers);
	__sanitizer_start_switch_fiber(&w->sched_fake, w->stacks + index * w->stack_bytes, w->stack_bytes);
#endif
#ifdef HOSTSIM_TSAN
	if (w->race_mode) __tsan_switch_to_fiber(w->wave_tsan[f->ids.wave], 1u);
	else __tsan_switch_to_fiber(f->tsan_fiber, 0);
#endif
#ifdef HOSTSIM_TRAFFIC
	t_traffic = w->traffic;
#endif
	hostsim_switch(&w->sched_sp, f->sp);
#ifdef HOSTSIM_TRAFFIC
	t_traffic = nullptr;
#endif
#ifdef HOSTSIM_ASAN
	__sanitizer_finish_switch_fiber(w->sched_fake, nullptr, nullptr);
#endif
}

// HOSTSIM_ORDER = forward (default) | reverse | shuffle[:seed]: the order in which the runnable lanes of a block are resumed and the
// blocks of a grid are started. The hardware promises neither; a result that changes with it depends on a schedule.
struct Order {
	int mode = 0; // 0 forward, 1 reverse, 2 shuffle
	uint64_t seed = 1;
	Order() {
		const char* v = getenv("HOSTSIM_ORDER");
		if (!v) return;
		if (!strncmp(v, "reverse", 7)) mode = 1;
		else if (!strncmp(v, "shuffle", 7)) {
			mode = 2;
			if (v[7] == ':') seed = strtoull(v + 8, nullptr, 10) * 2 + 1;
		}
	}
	uint64_t block(uint64_t b, uint64_t n) const {
		if (mode == 1) return n - 1 - b;
		if (mode == 2) return (b * 2654435761ull + seed) % n; // a bijection: the multiplier is a prime larger than any grid
		return b;
	}
};
const Order& order() {
	static const Order o;
	return o;
}

struct LaunchDesc {
	dim3 grid, block;
	size_t dyn_lds;
	const void* kernarg;
	LaneEntry entry;
	void* closure;
	void* traffic_acc;
};

void run_block(Worker* w, const LaunchDesc& L, uint64_t linear_block) {
	const uint32_t n_lanes = L.block.x * L.block.y * L.block.z;
	const uint32_t n_waves = (n_lanes + 63) / 64;
	tls.bid = Idx3{(uint32_t)(linear_block % L.grid.x), (uint32_t)(linear_block / L.grid.x % L.grid.y), (uint32_t)(linear_block / ((uint64_t)L.grid.x * L.grid.y))};
	tls.bdim = Idx3{L.block.x, L.block.y, L.block.z};
	tls.gdim = Idx3{L.grid.x, L.grid.y, L.grid.z};
	tls.kernarg = L.kernarg;
	tls.dyn_lds = w->dyn_lds.data();
	w->entry = L.entry;
	w->closure = L.closure;
#ifdef HOSTSIM_TRAFFIC
	if (!w->traffic) w->traffic = traffic_block_new();
#endif
	for (uint32_t i = 0; i < n_lanes; ++i) {
		Fiber& f = w->fibers[i];
		f.ids.tid = Idx3{i % L.block.x, i / L.block.x % L.block.y, i / (L.block.x * L.block.y)};
		f.ids.lane = i & 63u;
		f.ids.wave = i >> 6;
		f.state = RUNNABLE;
		f.site = nullptr;
		f.post = 0;
		// initial frame: six callee-saved registers, the entry point as return address, one slot that keeps the ABI's alignment
		void** top = reinterpret_cast<void**>(w->stacks + (size_t)(i + 1) * w->stack_bytes);
		top[-1] = nullptr;
		top[-2] = reinterpret_cast<void*>(&fiber_main);
		for (int k = 3; k <= 8; ++k) top[-k] = nullptr;
		f.sp = top - 8;
#ifdef HOSTSIM_ASAN
		f.fake_stack = nullptr;
#endif
#ifdef HOSTSIM_TSAN
		f.tsan_fiber = w->race_mode ? nullptr : __tsan_create_fiber(0);
#endif
	}
#ifdef HOSTSIM_TSAN
	if (getenv("HOSTSIM_DEBUG")) fprintf(stderr, "hostsim: block %llu race_mode %d waves %u\n", (unsigned long long)linear_block, (int)w->race_mode, n_waves);
	w->sched_tsan = __tsan_get_current_fiber();
	if (w->race_mode) {
		// The blocks of a launch are spread over (at least) two host threads (run_grid): blocks on different threads are concurrent, as
		// all blocks are on the GPU; the blocks one thread runs re-use its wave fibers - and its copy of every __shared__ object, which
		// is thread-local storage - and are ordered one after the other. The host takes the waves' clocks after the whole grid.
		for (uint32_t wv = 0; wv < n_waves; ++wv)
			if (!w->wave_tsan[wv]) w->wave_tsan[wv] = __tsan_create_fiber(0);
		__tsan_acquire(&w->done_token); // the blocks this thread ran before (their LDS is this block's LDS) ...
		__tsan_release(&w->launch_token); // ... and the launch happen before every wave of this block
	}
#endif
	uint32_t live = n_lanes;
	const Order& ord = order();
	uint16_t lane_order[MAX_LANES];
	for (uint32_t i = 0; i < n_lanes; ++i) lane_order[i] = (uint16_t)(ord.mode == 1 ? n_lanes - 1 - i : i);
	if (ord.mode == 2) {
		uint64_t x = ord.seed * 0x9E3779B97F4A7C15ull + linear_block;
		for (uint32_t i = n_lanes; i > 1; --i) { // Fisher-Yates with a 64-bit LCG
			x = x * 6364136223846793005ull + 1442695040888963407ull;
			const uint32_t j = (uint32_t)((x >> 33) % i);
			const uint16_t t = lane_order[i - 1];
			lane_order[i - 1] = lane_order[j];
			lane_order[j] = t;
		}
	}
	while (live) {
		for (uint32_t n = 0; n < n_lanes; ++n) {
			const uint32_t i = lane_order[n];
			Fiber& f = w->fibers[i];
			if (f.state != RUNNABLE) continue;
			resume(w, &f);
			if (f.state == DONE) --live;
		}
		if (!live) break;
		bool any_group = false;
		uint32_t at_barrier = 0;
		for (uint32_t wv = 0; wv < n_waves; ++wv) {
			const uint32_t lo = wv * 64, hi = lo + 64 < n_lanes ? lo + 64 : n_lanes;
			const void* site = nullptr;
			for (uint32_t i = lo; i < hi; ++i) {
				const Fiber& f = w->fibers[i];
				if (f.state == WAIT_BLOCK) ++at_barrier;
				if (f.state == WAIT_WAVE && (!site || f.site < site)) site = f.site;
			}
			if (!site) continue;
			WaveSnapshot& s = w->snap[wv];
			s.exec = 0;
			s.nonzero = 0;
			s.first = 64;
			for (uint32_t i = lo; i < hi; ++i) {
				Fiber& f = w->fibers[i];
				s.val[i - lo] = f.post; // lanes outside the group: what they posted last (a stale register)
				if (f.state != WAIT_WAVE || f.site != site) continue;
				s.exec |= 1ull << (i - lo);
				if (f.post) s.nonzero |= 1ull << (i - lo);
				if (s.first == 64) s.first = i - lo;
				f.state = RUNNABLE;
			}
			any_group = true;
		}
		if (any_group) continue;
		if (at_barrier != live) {
			fprintf(stderr, "hostsim: block (%u,%u,%u) is stuck: %u live lanes, %u at the block barrier, none at a wave operation\n", tls.bid.x, tls.bid.y, tls.bid.z, live, at_barrier);
			abort();
		}
		for (uint32_t i = 0; i < n_lanes; ++i)
			if (w->fibers[i].state == WAIT_BLOCK) w->fibers[i].state = RUNNABLE;
	}
#ifdef HOSTSIM_TSAN
	if (!w->race_mode) {
		for (uint32_t i = 0; i < n_lanes; ++i) __tsan_destroy_fiber(w->fibers[i].tsan_fiber);
	}
#endif
}

void block_done(Worker* w, const LaunchDesc& L) {
#ifdef HOSTSIM_TRAFFIC
	traffic_flush_block(w->traffic, L.traffic_acc, L.block.x);
#else
	(void)w;
	(void)L;
#endif
}

void grid_done(Worker* w) {
#ifdef HOSTSIM_TSAN
	if (w->race_mode) __tsan_acquire(&w->done_token); // launch complete: the host is ordered after every wave of the grid
#else
	(void)w;
#endif
}

// ---- optional pool: blocks of one launch spread over host threads (HOSTSIM_THREADS > 1) -------------------------------------------
struct Pool {
	std::mutex launch_lock; // one launch at a time uses the pool
	std::mutex m;
	std::condition_variable cv_work, cv_done;
	std::vector<std::thread> threads;
	const LaunchDesc* desc = nullptr;
	std::atomic<uint64_t> next{0};
	uint64_t n_blocks = 0;
	uint64_t generation = 0;
	uint32_t busy = 0;
	bool stop = false;

	// `index`: 0 = the launching thread, k = pool thread k. Blocks are handed out on demand - except under the race detector, where
	// thread k takes blocks k, k + T, ...: with a shared counter the launching thread can be through a small grid before a pool thread
	// has woken up, every block then ran on ONE host thread and an unordered exchange between two blocks goes unreported (seen as a
	// flaky tests/test_hostsim.py::test_race_detector_reports_races_and_only_races on a loaded machine).
	// ... and the launching thread starts its own blocks only once every pool thread is past the handshake: a pool thread that wakes up
	// AFTER the launching thread has finished its blocks takes `m` behind the launching thread's cv_done.wait - a happens-before edge from
	// all of block 0's accesses to all of block 1's that the hardware does not have. (Relaxed counter: no ordering of its own.)
	bool static_split = false;
	std::atomic<uint32_t> started{0};
	void work(const LaunchDesc& L, uint32_t index) {
		Worker* w = worker();
		if (w->dyn_lds.size() < L.dyn_lds) w->dyn_lds.resize(L.dyn_lds);
		const uint64_t stride = (uint64_t)threads.size() + 1;
		for (uint64_t k = 0;; ++k) {
			const uint64_t b = static_split ? index + k * stride : next.fetch_add(1, std::memory_order_relaxed);
			if (b >= n_blocks) break;
			run_block(w, L, order().block(b, n_blocks));
			block_done(w, L);
		}
		grid_done(w);
	}
	void thread_main(uint32_t index) {
		uint64_t seen = 0;
		std::unique_lock<std::mutex> lk(m);
		for (;;) {
			cv_work.wait(lk, [&] { return stop || generation != seen; });
			if (stop) return;
			seen = generation;
			const LaunchDesc* d = desc;
			lk.unlock();
			started.fetch_add(1, std::memory_order_relaxed);
			work(*d, index);
			lk.lock();
			if (--busy == 0) cv_done.notify_all();
		}
	}
	void run(const LaunchDesc& L, uint64_t blocks, uint32_t n_threads, bool split) {
		std::lock_guard<std::mutex> guard(launch_lock);
		while (threads.size() + 1 < n_threads) {
			const uint32_t index = (uint32_t)threads.size() + 1;
			threads.emplace_back([this, index] { thread_main(index); });
		}
		{
			std::lock_guard<std::mutex> lk(m);
			desc = &L;
			n_blocks = blocks;
			static_split = split;
			started.store(0, std::memory_order_relaxed);
			next.store(0);
			busy = (uint32_t)threads.size();
			++generation;
		}
		cv_work.notify_all();
		if (split) {
			while (started.load(std::memory_order_relaxed) < threads.size()) std::this_thread::yield();
		}
		work(L, 0);
		std::unique_lock<std::mutex> lk(m);
		cv_done.wait(lk, [&] { return busy == 0; });
	}
	~Pool() {
		{
			std::lock_guard<std::mutex> lk(m);
			stop = true;
		}
		cv_work.notify_all();
		for (std::thread& t : threads) t.join();
	}
};

Pool& pool() {
	static Pool* p = new Pool; // never destroyed: worker threads must not be joined from a static destructor of a dlopen'ed library
	return *p;
}

struct Event { double ms; bool recorded; };
double now_ms() {
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

void note_launch_error(hipError_t e) { t_last_error = e; }

__attribute__((noinline, convergent)) const WaveSnapshot* wave_exchange(uint64_t value) {
	Worker* w = t_worker;
	Fiber* f = w->current;
	f->post = value;
	f->site = __builtin_return_address(0);
	f->state = WAIT_WAVE;
	yield_to_scheduler();
	return &t_worker->snap[f->ids.wave];
}

__attribute__((noinline, convergent)) void block_barrier() {
	Worker* w = t_worker;
	Fiber* f = w->current;
	f->state = WAIT_BLOCK;
#ifdef HOSTSIM_TSAN
	if (w->race_mode) __tsan_release(&w->barrier_token);
#endif
	yield_to_scheduler();
#ifdef HOSTSIM_TSAN
	if (w->race_mode) __tsan_acquire(&w->barrier_token);
#endif
}

void* lds_pointer(uint32_t lds_byte_address) {
	// a __shared__ object is thread-local storage of this library: its address shares the upper half with the anchor's unless the
	// thread's block happens to straddle a 4 GiB boundary
	const uint64_t anchor = reinterpret_cast<uint64_t>(&t_lds_anchor);
	uint64_t best = 0, best_dist = ~0ull;
	for (int k = -1; k <= 1; ++k) {
		const uint64_t cand = (((anchor >> 32) + (uint64_t)(int64_t)k) << 32) | lds_byte_address;
		const uint64_t dist = cand > anchor ? cand - anchor : anchor - cand;
		if (dist < best_dist) {
			best_dist = dist;
			best = cand;
		}
	}
	return reinterpret_cast<void*>(best);
}

int mov_dpp(int v, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
	const WaveSnapshot* s = wave_exchange((uint64_t)(uint32_t)v);
	const uint32_t l = tls.ids->lane, row = l & ~15u, in_row = l & 15u;
	int src = -1;
	if (ctrl >= 0 && ctrl <= 0xff) src = (int)((l & ~3u) + (((uint32_t)ctrl >> (2 * (l & 3u))) & 3u)); // quad_perm
	else if (ctrl >= 0x101 && ctrl <= 0x10f) src = in_row + (uint32_t)(ctrl - 0x100) < 16 ? (int)(l + (uint32_t)(ctrl - 0x100)) : -1; // row_shl
	else if (ctrl >= 0x111 && ctrl <= 0x11f) src = in_row >= (uint32_t)(ctrl - 0x110) ? (int)(l - (uint32_t)(ctrl - 0x110)) : -1; // row_shr
	else if (ctrl >= 0x121 && ctrl <= 0x12f) src = (int)(row + ((in_row + 16 - (uint32_t)(ctrl - 0x120)) & 15u)); // row_ror
	else if (ctrl == 0x140) src = (int)(row + 15 - in_row); // row_mirror
	else if (ctrl == 0x141) src = (int)((l & ~7u) + 7 - (l & 7u)); // row_half_mirror
	else {
		fprintf(stderr, "hostsim: DPP control 0x%x is not modelled\n", ctrl);
		abort();
	}
	const bool row_on = (row_mask >> (l >> 4)) & 1, bank_on = (bank_mask >> ((l >> 2) & 3)) & 1;
	if (!row_on || !bank_on) return v;
	if (src < 0 || !((s->exec >> src) & 1)) return bound_ctrl ? 0 : v;
	return (int)(uint32_t)s->val[src];
}

void run_grid(const char* label, dim3 grid, dim3 block, size_t dyn_lds_bytes, const void* kernarg, LaneEntry entry, void* closure) {
	const uint64_t n_lanes = (uint64_t)block.x * block.y * block.z;
	const uint64_t n_blocks = (uint64_t)grid.x * grid.y * grid.z;
	if (!n_lanes || n_lanes > MAX_LANES || !n_blocks || dyn_lds_bytes > 160 * 1024) {
		t_last_error = hipErrorInvalidValue;
		return;
	}
	LaunchDesc L{grid, block, dyn_lds_bytes, kernarg, entry, closure, nullptr};
#ifdef HOSTSIM_TRAFFIC
	struct TrafficScope { // one accumulator per launch; the blocks run on this thread only
		const char* label; void* acc; uint64_t lanes;
		~TrafficScope() { traffic_launch_end(label, acc, lanes); }
	} traffic_scope{label, traffic_launch_begin(), n_lanes * n_blocks};
	L.traffic_acc = traffic_scope.acc;
#else
	(void)label;
#endif
	static const bool race_mode = env_size("HOSTSIM_RACE", 0) != 0; // (only acted on in a ThreadSanitizer build)
	static const uint32_t n_threads = std::max<uint32_t>((uint32_t)env_size("HOSTSIM_THREADS", 1), race_mode ? 2u : 1u);
#ifndef HOSTSIM_TRAFFIC
	if (n_threads > 1 && n_blocks >= (race_mode ? 2u : 4u)) {
		pool().run(L, n_blocks, n_threads, race_mode);
		return;
	}
#else
	(void)race_mode;
	(void)n_threads;
#endif
	Worker* w = worker();
	if (w->dyn_lds.size() < dyn_lds_bytes) w->dyn_lds.resize(dyn_lds_bytes);
	for (uint64_t b = 0; b < n_blocks; ++b) {
		run_block(w, L, order().block(b, n_blocks));
		block_done(w, L);
	}
	grid_done(w);
}

} // namespace hostsim

// ---- the HIP entry points the C ABI uses ------------------------------------------------------------------------------------------
using hostsim::t_last_error;

extern "C" {

hipError_t hipGetDeviceCount(int* count) {
	*count = 1;
	return hipSuccess;
}
hipError_t hipSetDevice(int device) { return device == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDevice(int* device) {
	*device = 0;
	return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int device) {
	if (device != 0) return hipErrorInvalidDevice;
	memset(prop, 0, sizeof(*prop));
	snprintf(prop->name, sizeof(prop->name), "hostsim (kernel sources executed on the CPU)");
	snprintf(prop->gcnArchName, sizeof(prop->gcnArchName), "gfx950:hostsim");
	prop->totalGlobalMem = 288ull << 30;
	prop->multiProcessorCount = 256;
	prop->warpSize = 64;
	prop->maxThreadsPerBlock = 1024;
	prop->sharedMemPerBlock = 160 * 1024;
	prop->major = 9;
	prop->minor = 5;
	return hipSuccess;
}
hipError_t hipGetLastError(void) {
	const hipError_t e = t_last_error;
	t_last_error = hipSuccess;
	return e;
}
const char* hipGetErrorString(hipError_t e) {
	switch (e) {
		case hipSuccess: return "no error";
		case hipErrorInvalidValue: return "invalid argument";
		case hipErrorOutOfMemory: return "out of memory";
		case hipErrorNoDevice: return "no device";
		case hipErrorInvalidDevice: return "invalid device ordinal";
		case hipErrorNotReady: return "not ready";
		default: return "unknown error";
	}
}
hipError_t hipMalloc(void** p, size_t bytes) {
	*p = nullptr;
	if (!bytes) return hipSuccess;
	if (bytes > (64ull << 30)) return hipErrorOutOfMemory;
	void* q = aligned_alloc(256, (bytes + 255) / 256 * 256);
	if (!q) return hipErrorOutOfMemory;
#ifdef HOSTSIM_TRAFFIC
	hostsim::traffic_alloc(q, (bytes + 255) / 256 * 256);
#endif
	*p = q;
	return hipSuccess;
}
hipError_t hipFree(void* p) {
#ifdef HOSTSIM_TRAFFIC
	if (p) hostsim::traffic_free(p);
#endif
	free(p);
	return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) {
	*dev = host;
	return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
	if (bytes) memmove(dst, src, bytes);
	return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t) { return hipMemcpy(dst, src, bytes, kind); }
hipError_t hipMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
	if (width > dpitch || width > spitch) return hipErrorInvalidValue;
	for (size_t r = 0; r < height; ++r) memmove(static_cast<char*>(dst) + r * dpitch, static_cast<const char*>(src) + r * spitch, width);
	return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t) {
	return hipMemcpy2D(dst, dpitch, src, spitch, width, height, kind);
}
hipError_t hipMemset(void* dst, int value, size_t bytes) {
	if (bytes) memset(dst, value, bytes);
	return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t) { return hipMemset(dst, value, bytes); }
hipError_t hipStreamCreate(hipStream_t* s) {
	*s = reinterpret_cast<hipStream_t>(new int(0));
	return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) {
	delete reinterpret_cast<int*>(s);
	return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) {
	*e = reinterpret_cast<hipEvent_t>(new hostsim::Event{0.0, false});
	return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
	delete reinterpret_cast<hostsim::Event*>(e);
	return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
	hostsim::Event* ev = reinterpret_cast<hostsim::Event*>(e);
	ev->ms = hostsim::now_ms();
	ev->recorded = true;
	return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
	const hostsim::Event* ea = reinterpret_cast<const hostsim::Event*>(a);
	const hostsim::Event* eb = reinterpret_cast<const hostsim::Event*>(b);
	if (!ea->recorded || !eb->recorded) return hipErrorInvalidValue;
	*ms = (float)(eb->ms - ea->ms);
	return hipSuccess;
}

} // extern "C"
