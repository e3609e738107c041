// Development probe: host time of build_cull_layout (the sorted device layout of the culling set; initial build and every compaction).
//   g++ -O2 -std=c++17 -pthread -I lumixengine_amd/csrc -I include tools/layout_bench.cpp -o tools/_build/layout_bench && tools/_build/layout_bench 10000000
#include <chrono>
#include <cstdio>
#include <random>
#include "lmx_cull_layout.h"
using namespace lmx;
int main(int argc, char** argv) {
	const size_t n = argc > 1 ? atol(argv[1]) : 10000000;
	std::vector<CullRec> recs(n);
	std::mt19937_64 rng(2);
	std::uniform_real_distribution<double> U(-15000.0, 15000.0);
	for (size_t i = 0; i < n; ++i) recs[i] = make_cull_rec((int32_t)i, 0, DV3{U(rng), U(rng), U(rng)}, 1.0f + (float)(i % 50));
	for (int r = 0; r < 3; ++r) {
		CullLayout out;
		auto t0 = std::chrono::steady_clock::now();
		build_cull_layout(recs, out);
		auto t1 = std::chrono::steady_clock::now();
		printf("build %.3f s (%zu cells, %u padded)\n", std::chrono::duration<double>(t1 - t0).count(), out.cells.size(), out.n_padded);
	}
}
