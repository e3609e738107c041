// worker_pool.h — TEST INFRASTRUCTURE (mine, not reference code). The thread pool under the reference-side shims: persistent workers
// asleep on a condition variable between jobs, like the engine's job-system threads (core/job_system.cpp). ref_shim.cpp runs its
// per-instance loops on it; cull_shim.cpp implements jobs::runN / jobs::wait (which the reference's jobs::forEach template calls) on it.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace lmx_ref {

struct WorkerPool {
	std::mutex mutex;
	std::condition_variable work, done;
	std::vector<std::thread> threads;
	const std::function<void()>* job = nullptr;
	unsigned generation = 0;
	int wanted = 0, claimed = 0, running = 0;

	void workerLoop() {
		unsigned seen = 0;
		std::unique_lock<std::mutex> lock(mutex);
		for (;;) {
			while (generation == seen || claimed >= wanted) {
				if (generation != seen) seen = generation; // job fully staffed: skip it
				work.wait(lock);
			}
			seen = generation;
			++claimed;
			const std::function<void()>* j = job;
			lock.unlock();
			(*j)();
			lock.lock();
			if (--running == 0) done.notify_one();
		}
	}

	// wakes `helpers` workers on `body` and returns at once; `body` must stay alive until finish()
	void start(int helpers, const std::function<void()>& body) {
		std::unique_lock<std::mutex> lock(mutex);
		while ((int)threads.size() < helpers && threads.size() < 255) {
			threads.emplace_back([this] { workerLoop(); });
			threads.back().detach();
		}
		const int staffed = helpers < (int)threads.size() ? helpers : (int)threads.size();
		job = &body;
		wanted = staffed;
		claimed = 0;
		running = staffed;
		++generation;
		work.notify_all();
	}

	void finish() {
		std::unique_lock<std::mutex> lock(mutex);
		done.wait(lock, [this] { return running == 0; });
		job = nullptr;
		wanted = 0;
	}

	void run(int helpers, const std::function<void()>& body) {
		start(helpers, body);
		body(); // the caller works too
		finish();
	}
};

// never destroyed: its detached workers wait on the condition variable until the process ends
inline WorkerPool& pool() {
	static WorkerPool& p = *new WorkerPool;
	return p;
}

// min(n_threads, count) workers pulling indices from one atomic cursor
template <typename F> void forEachJob(unsigned count, int n_threads, const F& f) {
	if (n_threads <= 1 || count <= 1) {
		for (unsigned i = 0; i < count; ++i) f(i);
		return;
	}
	std::atomic<unsigned> cursor{0};
	const std::function<void()> worker = [&]() {
		for (;;) {
			const unsigned i = cursor.fetch_add(1, std::memory_order_relaxed);
			if (i >= count) return;
			f(i);
		}
	};
	const int n = n_threads < (int)count ? n_threads : (int)count;
	pool().run(n - 1, worker);
}

} // namespace lmx_ref
