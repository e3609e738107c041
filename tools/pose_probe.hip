// Development probe: times k_pose_palette<64, 2> / <64, 3> (a wave per 4 / 8 instances)
// with phases switched off (bit 1: level walk, 2: global stores, 4: pose loads).   usage: pose_probe [instances]
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I lumixengine_amd/csrc tools/pose_probe.hip -o gpurun_out/pose_probe
#include <hip/hip_runtime.h>
__device__ int g_probe_mask;
#define LMX_PROBE_SKIP(bit) ((g_probe_mask & (bit)) != 0)
#include "skin_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace lmx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
	const uint32_t n_inst = argc > 1 ? atoi(argv[1]) : 100000, nb = 64;
	std::vector<int16_t> parents(nb, -1);
	std::vector<uint8_t> depth(nb, 0);
	srand(4);
	uint32_t maxd = 0;
	for (uint32_t i = 1; i < nb; ++i) { int lo = i > 8 ? i - 8 : 0; parents[i] = lo + rand() % (i - lo); depth[i] = depth[parents[i]] + 1; if (depth[i] > maxd) maxd = depth[i]; }
	std::vector<uint32_t> items; std::vector<uint16_t> off{0};
	for (uint32_t d = 1; d <= maxd; ++d) { for (uint32_t i = 0; i < nb; ++i) if (depth[i] == d) items.push_back(i | ((uint32_t)parents[i] << 16)); off.push_back((uint16_t)items.size()); }
	printf("max_depth %u\n", maxd);
	std::vector<SkinInstance> inst(n_inst); std::vector<PoseGroup> groups;
	for (uint32_t i = 0; i < n_inst; ++i) { SkinInstance in{}; in.bone_offset = i * nb; in.n_bones = nb; in.model_offset = 0; in.max_depth = maxd; in.lv_items_offset = 0; in.lv_off_offset = 0; inst[i] = in; }
	for (uint32_t i = 0; i < n_inst; i += 16) groups.push_back(PoseGroup{i, n_inst - i < 16 ? n_inst - i : 16});
	const uint32_t n_groups16 = (uint32_t)groups.size();
	for (uint32_t i = 0; i < n_inst; i += 4) groups.push_back(PoseGroup{i, n_inst - i < 4 ? n_inst - i : 4});
	const uint32_t n_groups4 = (uint32_t)groups.size() - n_groups16;
	for (uint32_t i = 0; i < n_inst; i += 8) groups.push_back(PoseGroup{i, n_inst - i < 8 ? n_inst - i : 8});
	const uint32_t n_groups8 = (uint32_t)groups.size() - n_groups16 - n_groups4;
	const size_t bones = (size_t)n_inst * nb;
	std::vector<float> pos(bones * 3), rot(bones * 4);
	for (auto& v : pos) v = rand() / (float)RAND_MAX - 0.5f;
	for (size_t i = 0; i < bones; ++i) { rot[4 * i] = 0.5f; rot[4 * i + 1] = 0.5f; rot[4 * i + 2] = 0.5f; rot[4 * i + 3] = 0.5f; }
	SkinInstance* d_inst; PoseGroup* d_groups; float *d_rp, *d_pp, *d_ip; float4 *d_rr, *d_pr, *d_ir, *d_pal; uint32_t* d_items; uint16_t* d_off;
	CK(hipMalloc(&d_inst, inst.size() * sizeof(SkinInstance))); CK(hipMemcpy(d_inst, inst.data(), inst.size() * sizeof(SkinInstance), hipMemcpyHostToDevice));
	CK(hipMalloc(&d_groups, groups.size() * sizeof(PoseGroup))); CK(hipMemcpy(d_groups, groups.data(), groups.size() * sizeof(PoseGroup), hipMemcpyHostToDevice));
	CK(hipMalloc(&d_rp, bones * 12)); CK(hipMemcpy(d_rp, pos.data(), bones * 12, hipMemcpyHostToDevice));
	CK(hipMalloc(&d_rr, bones * 16)); CK(hipMemcpy(d_rr, rot.data(), bones * 16, hipMemcpyHostToDevice));
	CK(hipMalloc(&d_pp, bones * 12)); CK(hipMalloc(&d_pr, bones * 16)); CK(hipMalloc(&d_pal, bones * 48));
	CK(hipMalloc(&d_ip, nb * 12)); CK(hipMemcpy(d_ip, pos.data(), nb * 12, hipMemcpyHostToDevice));
	CK(hipMalloc(&d_ir, nb * 16)); CK(hipMemcpy(d_ir, rot.data(), nb * 16, hipMemcpyHostToDevice));
	CK(hipMalloc(&d_items, (items.size() + 256) * 4)); CK(hipMemcpy(d_items, items.data(), items.size() * 4, hipMemcpyHostToDevice));
	CK(hipMalloc(&d_off, off.size() * 2)); CK(hipMemcpy(d_off, off.data(), off.size() * 2, hipMemcpyHostToDevice));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int kind = 1; kind < 3; ++kind)
	for (int mask = 0; mask < 8; ++mask) {
		CK(hipMemcpyToSymbol(HIP_SYMBOL(g_probe_mask), &mask, sizeof(int)));
		float best = 1e9f;
		for (int it = 0; it < 6; ++it) {
			CK(hipEventRecord(e0));
			if (kind == 1) hipLaunchKernelGGL((k_pose_palette<64, 2>), dim3(n_groups4), dim3(64), 0, 0, d_inst, d_groups + n_groups16, d_rp, d_rr, d_pp, d_pr, d_items, d_off, d_ip, d_ir, d_pal, (float4*)nullptr);
			else hipLaunchKernelGGL((k_pose_palette<64, 3>), dim3(n_groups8), dim3(64), 0, 0, d_inst, d_groups + n_groups16 + n_groups4, d_rp, d_rr, d_pp, d_pr, d_items, d_off, d_ip, d_ir, d_pal, (float4*)nullptr);
			CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			if (it && ms < best) best = ms;
		}
		printf("%s mask %d (skip%s%s%s): %.4f ms\n", kind == 0 ? "k_pose_palette<4> " : (kind == 1 ? "k_pose_palette<64,2> " : "k_pose_palette<64,3> "), mask, mask & 1 ? " walk" : "", mask & 2 ? " stores" : "", mask & 4 ? " loads" : "", best);
	}
	return 0;
}
