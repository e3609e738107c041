// Development probe: times k_skin_vertices / k_skin_shared with parts compiled out. LMX_PROBE_MASK bits (k_skin_vertices and the round-2
// k_skin_shared; the LDS-DMA k_skin_shared only honours 8 and 16):
//     8 vertex stores      16 LDS palette reads      32 palette staging (fetch + spread; every instance reads buffer 0)
//    64 per-instance barrier (k_skin_shared)      128 per-vertex scheduling barrier      256 staging without its global load
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DLMX_PROBE_MASK=<bits>] [-DLMX_SHARED_NT=0|1] [-DLMX_SHARED_ST16=0|1]
//         [-DLMX_SHARED_ZSKIP=0|1] -I lumixengine_amd/csrc -I include tools/skin_probe.hip -o tools/_build/skin_probe_<name>
//   skin_probe_<name> [instances] [instances per block] [1 = k_skin_shared, 0 = k_skin_vertices] [mesh: 0 = 4 random bones of 64 per
//                     vertex (worst case), 1 = character-like: consecutive vertices follow one bone, 1-2 influences, zero-padded]
// Round-2 readings, 20 000 instances x 10 000 vertices x 64 bones, ms per 1e9 vertices (k_skin_shared): all 3.1-3.2 | no stores 1.36 |
// no LDS reads 2.2 | neither 0.79 | no staging 2.59 | no staging, no stores 0.93 | staging without its load 2.85 |
// no staging, no barrier 3.16 (the barrier keeps a block's waves on ONE instance's 61 KB of output: without it the store stream
// loses locality). A bare 12-byte fill of the same shape: 2.0-2.4 (tools/write_probe.hip, tools/lds_store_probe.hip).
#include <hip/hip_runtime.h>
#ifndef LMX_PROBE_MASK
#define LMX_PROBE_MASK 0
#endif
#define LMX_PROBE_SKIP(bit) ((LMX_PROBE_MASK & (bit)) != 0) // compile-time: the probed kernel carries no extra code
#include "skin_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace lmx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
	const uint32_t n_inst = argc > 1 ? atoi(argv[1]) : 20000, nb = 64, nv = 10000;
	std::vector<SkinInstance> inst(n_inst);
	for (uint32_t i = 0; i < n_inst; ++i) { SkinInstance in{}; in.bone_offset = i * nb; in.n_bones = nb; in.vert_offset = 0; in.n_verts = nv; in.out_offset = i * nv; inst[i] = in; }
	std::vector<float4> mesh(nv * 2);
	std::vector<float> pal((size_t)n_inst * nb * 12);
	srand(3);
	const int mesh_kind = argc > 4 ? atoi(argv[4]) : 0;
	std::vector<uint32_t> gidx(nv);
	for (uint32_t v = 0; v < nv; ++v) {
		uint32_t idx;
		float w[4] = {0.25f, 0.25f, 0.25f, 0.25f};
		if (mesh_kind == 0) {
			idx = (uint32_t)(rand() % nb) | ((uint32_t)(rand() % nb) << 8) | ((uint32_t)(rand() % nb) << 16) | ((uint32_t)(rand() % nb) << 24);
		} else { // a limb at a time: ~190 consecutive vertices per bone; every 6th vertex also follows the neighbouring bone
			const uint32_t b = (v / 190u) % nb, b2 = (b + 1) % nb;
			const bool two = v % 6u == 0;
			idx = b | (two ? b2 : 0u) << 8;
			w[0] = two ? 0.7f : 1.f; w[1] = two ? 0.3f : 0.f; w[2] = 0.f; w[3] = 0.f;
		}
		gidx[v] = idx;
		float bits; memcpy(&bits, &idx, 4);
		mesh[2 * v] = make_float4(w[0], w[1], w[2], w[3]);
		mesh[2 * v + 1] = make_float4(rand() / (float)RAND_MAX, rand() / (float)RAND_MAX, rand() / (float)RAND_MAX, bits);
	}
	for (auto& v : pal) v = rand() / (float)RAND_MAX;
	SkinInstance* d_inst; float* d_out; float4 *d_mesh, *d_pal;
	CK(hipMalloc(&d_inst, inst.size() * sizeof(SkinInstance))); CK(hipMemcpy(d_inst, inst.data(), inst.size() * sizeof(SkinInstance), hipMemcpyHostToDevice));
	CK(hipMalloc(&d_mesh, mesh.size() * 16)); CK(hipMemcpy(d_mesh, mesh.data(), mesh.size() * 16, hipMemcpyHostToDevice));
	CK(hipMalloc(&d_pal, pal.size() * 4)); CK(hipMemcpy(d_pal, pal.data(), pal.size() * 4, hipMemcpyHostToDevice));
	CK(hipMalloc(&d_out, (size_t)n_inst * nv * 12));
	const uint32_t per_block = argc > 2 ? atoi(argv[2]) : 64, tile = 5056;
	std::vector<SkinChunk> chunks;
	// per tile: the bones it references + records with tile-local indices (what lmx_skin_add_mesh builds)
	std::vector<uint8_t> tile_bones;
	std::vector<float4> mesh_local(mesh);
	struct Tile { uint32_t at, n; };
	std::vector<Tile> tiles;
	for (uint32_t t = 0; t * tile < nv; ++t) {
		int local_of[256];
		for (int& x : local_of) x = -1;
		Tile tl{(uint32_t)tile_bones.size(), 0};
		for (uint32_t v = t * tile; v < nv && v < (t + 1) * tile; ++v) {
			uint32_t li = 0;
			for (int k = 0; k < 4; ++k) {
				const uint32_t b = (gidx[v] >> (8 * k)) & 0xffu;
				if (local_of[b] < 0) { local_of[b] = (int)tl.n++; tile_bones.push_back((uint8_t)b); }
				li |= (uint32_t)local_of[b] << (8 * k);
			}
			float bits; memcpy(&bits, &li, 4);
			mesh_local[2 * v + 1].w = bits;
		}
		tiles.push_back(tl);
		printf("tile %u references %u bones\n", t, tl.n);
	}
	for (uint32_t f = 0; f < n_inst; f += per_block)
		for (uint32_t t = 0; t * tile < nv; ++t)
			chunks.push_back(SkinChunk{f, n_inst - f < per_block ? n_inst - f : per_block, t * tile, (t + 1) * tile < nv ? (t + 1) * tile : nv, 0u, tiles[t].at, tiles[t].n, 0u});
	float4* d_mesh_local; uint8_t* d_tile_bones;
	CK(hipMalloc(&d_mesh_local, mesh_local.size() * 16)); CK(hipMemcpy(d_mesh_local, mesh_local.data(), mesh_local.size() * 16, hipMemcpyHostToDevice));
	CK(hipMalloc(&d_tile_bones, tile_bones.size())); CK(hipMemcpy(d_tile_bones, tile_bones.data(), tile_bones.size(), hipMemcpyHostToDevice));
	SkinChunk* d_chunks; CK(hipMalloc(&d_chunks, chunks.size() * sizeof(SkinChunk))); CK(hipMemcpy(d_chunks, chunks.data(), chunks.size() * sizeof(SkinChunk), hipMemcpyHostToDevice));
	const bool hot_palettes = argc > 5 && atoi(argv[5]) != 0; // k_skin_multi: every block reads the first instances' palettes (cache-resident): what the palette loads' latency costs
	const int kind = argc > 3 ? atoi(argv[3]) : 1; // 0: k_skin_vertices, 1: k_skin_shared, 2: k_skin_multi for I = 1, 2, 4, 8, 16 (x vertex-range splits 1, 2, 4)
	const bool shared = kind == 1;
	printf("%s kernel, %zu chunks of %u instances\n", kind == 2 ? "multi" : shared ? "shared" : "streaming", chunks.size(), per_block);
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	if (kind == 2) {
		for (uint32_t I : {1u, 2u, 4u})
			for (uint32_t splits : {1u, 2u}) {
				std::vector<SkinMultiChunk> mc;
				const uint32_t stage_bones = argc > 6 ? (uint32_t)atoi(argv[6]) : nb; // bones staged per block (the mesh's largest bone index + 1)
				const uint32_t range = ((nv + splits - 1) / splits + 63u) & ~63u;
				for (uint32_t f = 0; f < n_inst; f += I)
					for (uint32_t v = 0; v < nv; v += range) mc.push_back(SkinMultiChunk{hot_palettes ? 0u : f * nb, nb, n_inst - f < I ? n_inst - f : I, v, v + range < nv ? v + range : nv, 0u, nv, f * nv, stage_bones});
				SkinMultiChunk* d_mc; CK(hipMalloc(&d_mc, mc.size() * sizeof(SkinMultiChunk))); CK(hipMemcpy(d_mc, mc.data(), mc.size() * sizeof(SkinMultiChunk), hipMemcpyHostToDevice));
				float best = 1e9f;
				const int repeat = getenv("PROBE_REPEAT") ? atoi(getenv("PROBE_REPEAT")) : 1; // launches back to back per timed region (sustained clocks: is a long launch slower per vertex than a short one?)
				for (int it = 0; it < 5; ++it) {
					CK(hipEventRecord(e0));
					for (int r = 0; r < repeat; ++r) CK(launch_skin_multi(0, I, d_mc, (uint32_t)mc.size(), skin_multi_lds_slots(stage_bones), d_mesh, d_pal, d_out, LMX_SKIN_FUSED));
					CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
					float ms; CK(hipEventElapsedTime(&ms, e0, e1));
					ms /= (float)repeat;
					if (it && ms < best) best = ms;
				}
				printf("multi I=%2u splits=%u pipe=%d prefetch=%d (%zu blocks): %.4f ms  = %.3f ms per 1e9 verts\n", I, splits, LMX_MULTI_PIPE, LMX_MULTI_PREFETCH, mc.size(), best, best * 1e9 / ((double)n_inst * nv));
				CK(hipFree(d_mc));
			}
		return 0;
	}
	for (int mask : {LMX_PROBE_MASK}) {
		float best = 1e9f;
		for (int it = 0; it < 5; ++it) {
			CK(hipEventRecord(e0));
			if (shared) CK(launch_skin_shared(0, d_inst, d_chunks, (uint32_t)chunks.size(), d_mesh_local, d_tile_bones, d_pal, d_out, false));
			else CK(launch_skin_vertices(0, d_inst, nullptr, n_inst, nv, d_mesh, d_pal, d_out, false));
			CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			if (it && ms < best) best = ms;
		}
		printf("mask %2d (skip%s%s): %.4f ms  = %.3f ms per 1e9 verts\n", mask, mask & 8 ? " stores" : "", mask & 16 ? " lds" : "", best, best * 1e9 / ((double)n_inst * nv));
	}
	return 0;
}
