// loopback_rccl.cpp — TEST DOUBLE for the five RCCL entry points lmx_capi_exchange.hip binds (ncclGetUniqueId, ncclCommInitRank,
// ncclCommDestroy, ncclAllGather, ncclGetErrorString), for ONE purpose: running the exchange with a world of SEVERAL ranks on a box that
// has one GPU. RCCL refuses two ranks on one device; this transport does not care: every rank is a process with its own HIP context
// (possibly on the same device), records travel through a POSIX shared-memory segment:
//     ncclAllGather(send, recv, count, type, comm, stream):  wait for `stream`, copy send -> segment[rank], barrier over the ranks,
//                                                              copy segment[0 .. world) -> recv, barrier
// It is host-synchronous where RCCL is stream-ordered: what it exercises is everything AROUND the collective - record layout per
// rank and frustum, offsets of the peers' records in the receive buffer, slot alternation, clipping, the readers - not the wire.
// Loaded through LMX_RCCL_LIBRARY (csrc/lmx_capi_exchange.hip); never part of the product.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr size_t SLOT_BYTES = 8u << 20; // per rank and collective
struct Header {
	std::atomic<uint32_t> arrived;
	std::atomic<uint32_t> generation;
	std::atomic<uint32_t> attached;
};
struct Comm {
	int rank, world;
	char name[64];
	size_t bytes;
	Header* hdr;
	char* data;
};
const char* g_error = "ok";

bool barrier(Comm* c) { // sense-reversing, 30 s timeout
	const uint32_t gen = c->hdr->generation.load();
	if (c->hdr->arrived.fetch_add(1) + 1 == (uint32_t)c->world) {
		c->hdr->arrived.store(0);
		c->hdr->generation.fetch_add(1);
		return true;
	}
	const auto t0 = std::chrono::steady_clock::now();
	while (c->hdr->generation.load() == gen) {
		sched_yield();
		if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) {
			g_error = "loopback barrier timed out (a rank is missing)";
			return false;
		}
	}
	return true;
}

size_t type_bytes(int t) { // ncclDataType_t: int8 0, uint8 1, int32 2, uint32 3, int64 4, uint64 5, half 6, float 7, double 8
	switch (t) {
		case 0: case 1: return 1;
		case 6: return 2;
		case 2: case 3: case 7: return 4;
		default: return 8;
	}
}

} // namespace

extern "C" {

struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId* id) {
	memset(id, 0, sizeof(*id));
	const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
	static std::atomic<uint32_t> serial{0};
	snprintf(id->internal, sizeof(id->internal), "/lmx_loopback_%d_%lld_%u", (int)getpid(), (long long)now, serial.fetch_add(1));
	return 0;
}

int ncclCommInitRank(void** out, int world, ncclUniqueId id, int rank) {
	if (!out || world < 1 || rank < 0 || rank >= world) { g_error = "bad arguments"; return 4; }
	Comm* c = new Comm;
	c->rank = rank;
	c->world = world;
	memcpy(c->name, id.internal, sizeof(c->name));
	c->name[sizeof(c->name) - 1] = 0;
	c->bytes = 4096 + SLOT_BYTES * (size_t)world;
	const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
	if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { g_error = "shm_open / ftruncate failed"; delete c; return 2; }
	void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (p == MAP_FAILED) { g_error = "mmap failed"; delete c; return 2; }
	c->hdr = (Header*)p; // a fresh segment is zero-filled: arrived = generation = attached = 0
	c->data = (char*)p + 4096;
	c->hdr->attached.fetch_add(1);
	const auto t0 = std::chrono::steady_clock::now();
	while (c->hdr->attached.load() < (uint32_t)world) { // like ncclCommInitRank: returns when every rank has joined
		sched_yield();
		if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { g_error = "loopback init timed out"; return 2; }
	}
	*out = c;
	return 0;
}

int ncclCommDestroy(void* comm) {
	Comm* c = (Comm*)comm;
	if (!c) return 0;
	munmap((void*)c->hdr, c->bytes);
	if (c->rank == 0) shm_unlink(c->name);
	delete c;
	return 0;
}

int ncclAllGather(const void* send, void* recv, size_t count, int type, void* comm, hipStream_t stream) {
	Comm* c = (Comm*)comm;
	const size_t bytes = count * type_bytes(type);
	if (!c || bytes > SLOT_BYTES) { g_error = "record larger than the loopback slot"; return 4; }
	if (hipStreamSynchronize(stream) != hipSuccess) { g_error = "hipStreamSynchronize failed"; return 1; }
	// LOOPBACK_RCCL_DELAY_US: a gather that takes this long (what a collective over several links costs, where this one is a memcpy): the
	// exchange's choice of mode by measured gather time is tested with it (tests/test_gpu_exchange.py, "auto_slow_gather")
	static const long delay_us = getenv("LOOPBACK_RCCL_DELAY_US") ? atol(getenv("LOOPBACK_RCCL_DELAY_US")) : 0;
	if (delay_us > 0) usleep((useconds_t)delay_us);
	if (hipMemcpy(c->data + SLOT_BYTES * (size_t)c->rank, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) { g_error = "D2H failed"; return 1; }
	std::atomic_thread_fence(std::memory_order_seq_cst);
	if (!barrier(c)) return 2;
	for (int r = 0; r < c->world; ++r)
		if (hipMemcpy((char*)recv + bytes * (size_t)r, c->data + SLOT_BYTES * (size_t)r, bytes, hipMemcpyHostToDevice) != hipSuccess) { g_error = "H2D failed"; return 1; }
	if (!barrier(c)) return 2; // nobody overwrites its slot while a peer still reads it
	return 0;
}

const char* ncclGetErrorString(int) { return g_error; }

} // extern "C"
