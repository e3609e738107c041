// traffic_runtime.cpp — TEST INFRASTRUCTURE (tests/hostsim, build mode "traffic"): a memory-traffic model of the kernels, taken from the
// kernels' own accesses. The product's sources are compiled with clang's ThreadSanitizer INSTRUMENTATION (a call in front of every load
// and store) but linked against THIS file instead of the sanitizer's runtime: every access a lane makes to device memory (hipMalloc'ed
// ranges; LDS, stacks and host data are ignored) is recorded, and per launch the records of a block are grouped into wave-instructions
// (same wave, same code address, same occurrence). Per kernel label the model reports
//     requested bytes            what the lanes asked for (reads / writes)          - the ALGORITHMIC bytes of DESIGN.md section 4
//                                (a wave-instruction whose lanes all read ONE address is a scalar load on the GPU: counted once, apart)
//     64-byte sector requests    distinct sectors per wave-instruction, summed       - what the L1 / TA path has to process
//     128-byte line requests     the same per cache line
//     footprint                  distinct 128-byte lines touched by the whole launch - the least HBM can move if every line moved once
// It knows nothing about time, caches or the order in which the hardware issues anything.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <unordered_set>
#include <vector>

namespace hostsim {

struct TrafficAccess { uint64_t pc, addr; uint32_t seq; uint16_t lane; uint8_t size, write; };
struct TrafficBlock { std::vector<TrafficAccess> log; };
thread_local TrafficBlock* t_traffic = nullptr; // non-null while a lane of a launch runs on this thread
TrafficBlock* traffic_block_new() { return new TrafficBlock; }

namespace {

struct Range { uint64_t lo, hi; };
std::mutex g_mutex;
std::vector<Range> g_ranges; // device allocations, sorted
struct KernelStats {
	uint64_t launches = 0, lanes = 0, read_bytes = 0, write_bytes = 0, uniform_read_bytes = 0, wave_instr = 0, read_sectors64 = 0, write_sectors64 = 0, lines128 = 0, footprint_read_lines = 0,
		footprint_write_lines = 0;
};
std::map<std::string, KernelStats> g_stats;
struct PcStats { uint64_t read_bytes = 0, write_bytes = 0, sectors64 = 0, wave_instr = 0; };
std::map<std::string, std::map<uint64_t, PcStats>> g_by_pc; // HOSTSIM_TRAFFIC_BY_PC=1: per kernel label and code address
thread_local Range t_last{1, 0};

bool in_device_memory(uint64_t a) {
	if (a >= t_last.lo && a < t_last.hi) return true;
	std::lock_guard<std::mutex> guard(g_mutex);
	auto it = std::upper_bound(g_ranges.begin(), g_ranges.end(), a, [](uint64_t v, const Range& r) { return v < r.lo; });
	if (it == g_ranges.begin()) return false;
	--it;
	if (a >= it->hi) return false;
	t_last = *it;
	return true;
}

} // namespace

void traffic_alloc(void* p, size_t n) {
	std::lock_guard<std::mutex> guard(g_mutex);
	const Range r{reinterpret_cast<uint64_t>(p), reinterpret_cast<uint64_t>(p) + n};
	g_ranges.insert(std::upper_bound(g_ranges.begin(), g_ranges.end(), r.lo, [](uint64_t v, const Range& x) { return v < x.lo; }), r);
}
void traffic_free(void* p) {
	std::lock_guard<std::mutex> guard(g_mutex);
	const uint64_t a = reinterpret_cast<uint64_t>(p);
	for (size_t i = 0; i < g_ranges.size(); ++i)
		if (g_ranges[i].lo == a) {
			g_ranges.erase(g_ranges.begin() + (long)i);
			break;
		}
	t_last = Range{1, 0};
}

static inline void record(const void* addr, uint32_t size, bool write, const void* pc) {
	TrafficBlock* b = t_traffic;
	if (!b) return;
	uint64_t a = reinterpret_cast<uint64_t>(addr);
	if (!in_device_memory(a)) return;
	const uint16_t lane = (uint16_t)(tls.ids->wave * 64u + tls.ids->lane);
	// the x86 compilation splits a 16-byte vector access into its components (four 4-byte accesses at consecutive addresses, back to back):
	// pieces of one lane that continue each other are one access again, up to the 16 bytes a lane moves per instruction
	if (!b->log.empty()) {
		TrafficAccess& prev = b->log.back();
		if (prev.lane == lane && prev.write == (uint8_t)write && prev.addr + prev.size == a && prev.size + size <= 16) {
			prev.size = (uint8_t)(prev.size + size);
			return;
		}
	}
	while (size) { // (memcpy-sized accesses are cut into 16-byte pieces: what one lane moves per instruction at most)
		const uint32_t n = size > 16 ? 16 : size;
		b->log.push_back(TrafficAccess{reinterpret_cast<uint64_t>(pc), a, (uint32_t)b->log.size(), lane, (uint8_t)n, (uint8_t)write});
		size -= n;
		a += n;
	}
}

struct LaunchAccumulator {
	KernelStats s;
	std::unordered_set<uint64_t> read_lines, write_lines;
	std::map<uint64_t, PcStats> by_pc;
	bool want_pc = getenv("HOSTSIM_TRAFFIC_BY_PC") != nullptr;
};

// groups one block's records into wave-instructions and adds them to the launch
void traffic_flush_block(TrafficBlock* b, void* launch_acc, uint32_t block_threads_x) {
	LaunchAccumulator* acc = static_cast<LaunchAccumulator*>(launch_acc);
	std::vector<TrafficAccess>& log = b->log;
	(void)block_threads_x;
	// occurrence index of (lane, pc): the k-th time this lane executed this access
	std::sort(log.begin(), log.end(), [](const TrafficAccess& x, const TrafficAccess& y) {
		if (x.pc != y.pc) return x.pc < y.pc;
		if (x.lane != y.lane) return x.lane < y.lane;
		return x.seq < y.seq;
	});
	std::vector<uint32_t> occ(log.size());
	for (size_t i = 0; i < log.size(); ++i) occ[i] = (i && log[i - 1].pc == log[i].pc && log[i - 1].lane == log[i].lane) ? occ[i - 1] + 1 : 0;
	std::vector<uint32_t> order(log.size());
	for (size_t i = 0; i < order.size(); ++i) order[i] = (uint32_t)i;
	std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
		const TrafficAccess &a = log[x], &c = log[y];
		if (a.pc != c.pc) return a.pc < c.pc;
		if ((a.lane >> 6) != (c.lane >> 6)) return (a.lane >> 6) < (c.lane >> 6);
		if (occ[x] != occ[y]) return occ[x] < occ[y];
		return a.lane < c.lane;
	});
	std::vector<uint64_t> sec, lin;
	size_t i = 0;
	while (i < order.size()) {
		size_t j = i;
		const TrafficAccess& first = log[order[i]];
		sec.clear();
		lin.clear();
		uint64_t bytes = 0;
		bool uniform = true;
		while (j < order.size()) {
			const TrafficAccess& a = log[order[j]];
			if (a.pc != first.pc || (a.lane >> 6) != (first.lane >> 6) || occ[order[j]] != occ[order[i]]) break;
			uniform = uniform && a.addr == first.addr && a.size == first.size;
			for (uint64_t s = a.addr >> 6; s <= (a.addr + a.size - 1) >> 6; ++s) sec.push_back(s);
			for (uint64_t l = a.addr >> 7; l <= (a.addr + a.size - 1) >> 7; ++l) {
				lin.push_back(l);
				(a.write ? acc->write_lines : acc->read_lines).insert(l);
			}
			bytes += a.size;
			++j;
		}
		std::sort(sec.begin(), sec.end());
		std::sort(lin.begin(), lin.end());
		const uint64_t n_sec = (uint64_t)(std::unique(sec.begin(), sec.end()) - sec.begin());
		acc->s.lines128 += (uint64_t)(std::unique(lin.begin(), lin.end()) - lin.begin());
		if (acc->want_pc) {
			PcStats& ps = acc->by_pc[first.pc];
			(first.write ? ps.write_bytes : ps.read_bytes) += (!first.write && uniform && j - i > 1) ? first.size : bytes;
			ps.sectors64 += n_sec;
			ps.wave_instr += 1;
		}
		if (first.write) {
			acc->s.write_bytes += bytes;
			acc->s.write_sectors64 += n_sec;
		} else if (uniform && j - i > 1) {
			// every lane of the wave reads the same address: a wave-uniform (scalar) load on the GPU - one request, not 64
			acc->s.uniform_read_bytes += first.size;
		} else {
			acc->s.read_bytes += bytes;
			acc->s.read_sectors64 += n_sec;
		}
		acc->s.wave_instr += 1;
		i = j;
	}
	log.clear();
}

void* traffic_launch_begin() { return new LaunchAccumulator; }
void traffic_launch_end(const char* label, void* launch_acc, uint64_t lanes) {
	LaunchAccumulator* acc = static_cast<LaunchAccumulator*>(launch_acc);
	std::lock_guard<std::mutex> guard(g_mutex);
	KernelStats& k = g_stats[label];
	k.launches += 1;
	k.lanes += lanes;
	k.read_bytes += acc->s.read_bytes;
	k.write_bytes += acc->s.write_bytes;
	k.uniform_read_bytes += acc->s.uniform_read_bytes;
	k.wave_instr += acc->s.wave_instr;
	k.read_sectors64 += acc->s.read_sectors64;
	k.write_sectors64 += acc->s.write_sectors64;
	k.lines128 += acc->s.lines128;
	k.footprint_read_lines += acc->read_lines.size();
	k.footprint_write_lines += acc->write_lines.size();
	for (const auto& kv : acc->by_pc) {
		PcStats& ps = g_by_pc[label][kv.first];
		ps.read_bytes += kv.second.read_bytes;
		ps.write_bytes += kv.second.write_bytes;
		ps.sectors64 += kv.second.sectors64;
		ps.wave_instr += kv.second.wave_instr;
	}
	delete acc;
}

} // namespace hostsim

extern "C" {

#pragma GCC visibility push(default)
void hostsim_traffic_reset(void) {
	std::lock_guard<std::mutex> guard(hostsim::g_mutex);
	hostsim::g_stats.clear();
	hostsim::g_by_pc.clear();
}
// HOSTSIM_TRAFFIC_BY_PC=1: one line per (kernel label, code address): "label<TAB>library offset (hex)<TAB>read bytes<TAB>written bytes<TAB>sector requests<TAB>wave-instructions"
int hostsim_traffic_dump_by_pc(const char* path) {
	std::lock_guard<std::mutex> guard(hostsim::g_mutex);
	FILE* f = fopen(path, "w");
	if (!f) return 1;
	for (const auto& kv : hostsim::g_by_pc)
		for (const auto& pc : kv.second) {
			Dl_info info;
			const uint64_t base = dladdr(reinterpret_cast<void*>(pc.first), &info) ? reinterpret_cast<uint64_t>(info.dli_fbase) : 0;
			fprintf(f, "%s\t%llx\t%llu\t%llu\t%llu\t%llu\n", kv.first.c_str(), (unsigned long long)(pc.first - base - 1), (unsigned long long)pc.second.read_bytes,
				(unsigned long long)pc.second.write_bytes, (unsigned long long)pc.second.sectors64, (unsigned long long)pc.second.wave_instr);
		}
	fclose(f);
	return 0;
}
// one JSON object: kernel label -> totals since the last reset
int hostsim_traffic_dump(const char* path) {
	std::lock_guard<std::mutex> guard(hostsim::g_mutex);
	FILE* f = fopen(path, "w");
	if (!f) return 1;
	fprintf(f, "{\n");
	bool first = true;
	for (const auto& kv : hostsim::g_stats) {
		const hostsim::KernelStats& k = kv.second;
		fprintf(f, "%s  \"%s\": {\"launches\": %llu, \"lanes\": %llu, \"read_bytes\": %llu, \"write_bytes\": %llu, \"uniform_read_bytes\": %llu, \"wave_instructions\": %llu, \"read_sector64_requests\": %llu, \"write_sector64_requests\": %llu, \"line128_requests\": %llu, \"footprint_read_bytes\": %llu, \"footprint_write_bytes\": %llu}",
			first ? "" : ",\n", kv.first.c_str(), (unsigned long long)k.launches, (unsigned long long)k.lanes, (unsigned long long)k.read_bytes, (unsigned long long)k.write_bytes,
			(unsigned long long)k.uniform_read_bytes, (unsigned long long)k.wave_instr, (unsigned long long)k.read_sectors64, (unsigned long long)k.write_sectors64, (unsigned long long)k.lines128, (unsigned long long)k.footprint_read_lines * 128ull, (unsigned long long)k.footprint_write_lines * 128ull);
		first = false;
	}
	fprintf(f, "\n}\n");
	fclose(f);
	return 0;
}
#pragma GCC visibility pop

// ---- the instrumentation's entry points (the names clang's ThreadSanitizer pass emits) ----------------------------------------------------
#define HOSTSIM_RW(n)                                                                                                                \
	__attribute__((visibility("default"))) void __tsan_read##n(void* a) { hostsim::record(a, n, false, __builtin_return_address(0)); }            \
	__attribute__((visibility("default"))) void __tsan_write##n(void* a) { hostsim::record(a, n, true, __builtin_return_address(0)); }            \
	__attribute__((visibility("default"))) void __tsan_unaligned_read##n(void* a) { hostsim::record(a, n, false, __builtin_return_address(0)); }  \
	__attribute__((visibility("default"))) void __tsan_unaligned_write##n(void* a) { hostsim::record(a, n, true, __builtin_return_address(0)); }  \
	__attribute__((visibility("default"))) void __tsan_volatile_read##n(void* a) { hostsim::record(a, n, false, __builtin_return_address(0)); }   \
	__attribute__((visibility("default"))) void __tsan_volatile_write##n(void* a) { hostsim::record(a, n, true, __builtin_return_address(0)); }   \
	__attribute__((visibility("default"))) void __tsan_read_write##n(void* a) {                                                                  \
		hostsim::record(a, n, false, __builtin_return_address(0));                                                                              \
		hostsim::record(a, n, true, static_cast<const char*>(__builtin_return_address(0)) + 1);                                                 \
	}                                                                                                                                            \
	__attribute__((visibility("default"))) void __tsan_unaligned_read_write##n(void* a) {                                                        \
		hostsim::record(a, n, false, __builtin_return_address(0));                                                                              \
		hostsim::record(a, n, true, static_cast<const char*>(__builtin_return_address(0)) + 1);                                                 \
	}
HOSTSIM_RW(1)
HOSTSIM_RW(2)
HOSTSIM_RW(4)
HOSTSIM_RW(8)
HOSTSIM_RW(16)
__attribute__((visibility("default"))) void __tsan_init(void) {}
__attribute__((visibility("default"))) void __tsan_vptr_update(void**, void*) {}
__attribute__((visibility("default"))) void __tsan_vptr_read(void**) {}
__attribute__((visibility("default"))) void* __tsan_memcpy(void* d, const void* s, size_t n) {
	hostsim::record(s, (uint32_t)n, false, __builtin_return_address(0));
	hostsim::record(d, (uint32_t)n, true, static_cast<const char*>(__builtin_return_address(0)) + 1);
	return memcpy(d, s, n);
}
__attribute__((visibility("default"))) void* __tsan_memmove(void* d, const void* s, size_t n) {
	hostsim::record(s, (uint32_t)n, false, __builtin_return_address(0));
	hostsim::record(d, (uint32_t)n, true, static_cast<const char*>(__builtin_return_address(0)) + 1);
	return memmove(d, s, n);
}
__attribute__((visibility("default"))) void* __tsan_memset(void* d, int v, size_t n) {
	hostsim::record(d, (uint32_t)n, true, __builtin_return_address(0));
	return memset(d, v, n);
}

} // extern "C"
