// test_adapter.cpp — exercises the C++ host side (GpuCullingSystem : CullingSystem) the way RenderModuleImpl would:
// add entities one by one, move some, cull through the virtual interface, walk the CullResult pages, free them.
// Reads a scene written by tests/test_gpu_adapter.py, writes the visible (type, id) pairs back for comparison with
// the CPU oracle. Needs an MI355X.
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "gpu_culling_system.h"

using namespace Lumix;

int main(int argc, char** argv) {
	if (argc < 3) return 2;
	FILE* in = fopen(argv[1], "rb");
	FILE* out = fopen(argv[2], "wb");
	if (!in || !out) return 2;
	uint32_t n = 0, n_moves = 0, n_frusta = 0;
	if (fread(&n, 4, 1, in) != 1 || fread(&n_moves, 4, 1, in) != 1 || fread(&n_frusta, 4, 1, in) != 1) return 2;
	std::vector<int32_t> entity(n), move_entity(n_moves);
	std::vector<uint8_t> type(n);
	std::vector<double> pos(3 * (size_t)n), move_pos(3 * (size_t)n_moves);
	std::vector<float> radius(n), move_radius(n_moves);
	std::vector<ShiftedFrustum> frusta(n_frusta);
	bool ok = fread(entity.data(), 4, n, in) == n && fread(type.data(), 1, n, in) == n && fread(pos.data(), 8, 3 * (size_t)n, in) == 3 * (size_t)n &&
			  fread(radius.data(), 4, n, in) == n && fread(move_entity.data(), 4, n_moves, in) == n_moves &&
			  fread(move_pos.data(), 8, 3 * (size_t)n_moves, in) == 3 * (size_t)n_moves && fread(move_radius.data(), 4, n_moves, in) == n_moves &&
			  fread(frusta.data(), sizeof(ShiftedFrustum), n_frusta, in) == n_frusta;
	if (!ok) return 2;

	PageAllocator pages;
	GpuCullingSystem gpu(pages);
	if (!gpu.isValid()) { fprintf(stderr, "no device: %s\n", gpu.lastError().c_str()); return 3; }
	CullingSystem& cs = gpu; // everything below goes through the reference's virtual interface

	if (cs.cull(frusta[0]) != nullptr) return 4; // empty system -> nullptr, like culling_system.cpp:322
	for (uint32_t i = 0; i < n; ++i) cs.add(EntityRef{entity[i]}, type[i], DVec3{pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]}, radius[i]);
	for (uint32_t i = 0; i < n_moves; ++i) {
		const EntityRef e{move_entity[i]};
		if (!cs.isAdded(e)) return 5;
		if (i % 3 == 0) cs.set(e, DVec3{move_pos[3 * i], move_pos[3 * i + 1], move_pos[3 * i + 2]}, move_radius[i]);
		else if (i % 3 == 1) cs.setPosition(e, DVec3{move_pos[3 * i], move_pos[3 * i + 1], move_pos[3 * i + 2]});
		else cs.setRadius(e, move_radius[i]);
	}
	cs.remove(EntityRef{entity[0]});

	for (uint32_t f = 0; f < n_frusta; ++f) {
		CullResult* res = (f % 2) ? cs.cull(frusta[f], 0) : cs.cull(frusta[f]);
		uint32_t total = res ? res->count() : 0;
		fwrite(&total, 4, 1, out);
		uint32_t pages_seen = 0;
		for (const CullResult* p = res; p; p = p->header.next, ++pages_seen) {
			if (p->header.count > 1020) return 6;
			for (uint32_t i = 0; i < p->header.count; ++i) {
				const int32_t rec[2] = {(int32_t)p->header.type, p->entities[i].index};
				fwrite(rec, 4, 2, out);
			}
		}
		if (res) res->free(pages);
	}
	// the frame's views in one pass over the spheres and one host wait: the same ids as one cull per view
	{
		const uint32_t nf = n_frusta < LMX_MAX_FRUSTA ? n_frusta : (uint32_t)LMX_MAX_FRUSTA;
		CullResult* many[LMX_MAX_FRUSTA];
		if (!gpu.cullMany(frusta.data(), nf, 0xff, many)) return 7;
		for (uint32_t f = 0; f < nf; ++f) {
			CullResult* one = cs.cull(frusta[f]);
			std::vector<int64_t> a, b;
			for (const CullResult* p = many[f]; p; p = p->header.next)
				for (uint32_t i = 0; i < p->header.count; ++i) a.push_back(((int64_t)p->header.type << 32) | (uint32_t)p->entities[i].index);
			for (const CullResult* p = one; p; p = p->header.next)
				for (uint32_t i = 0; i < p->header.count; ++i) b.push_back(((int64_t)p->header.type << 32) | (uint32_t)p->entities[i].index);
			std::sort(a.begin(), a.end());
			std::sort(b.begin(), b.end());
			if (many[f]) many[f]->free(pages);
			if (one) one->free(pages);
			if (a != b) { fprintf(stderr, "cullMany view %u: %zu ids vs %zu\n", f, a.size(), b.size()); return 8; }
		}
	}
	fclose(out);
	fclose(in);
	printf("adapter ok\n");
	return 0;
}
