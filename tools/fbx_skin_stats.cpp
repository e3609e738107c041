// Development probe (needs /root/reference): skinning statistics of a REAL mesh - the reference's demo character - read with the OpenFBX the
// reference vendors (external/openfbx, compiled in place, nothing copied): influences per control point and the number of distinct bones
// a tile of consecutive control points touches. Evidence for k_skin_shared's next step: stage only the bones a tile uses, skip
// bone slots whose weight is zero for the whole wave.
//   R=/root/reference/external/openfbx; g++ -O2 -std=c++17 -c -I $R $R/ofbx.cpp -o /tmp/ofbx.o; gcc -O2 -c $R/libdeflate.c -o /tmp/libdeflate.o
//   g++ -O2 -std=c++17 -I $R tools/fbx_skin_stats.cpp /tmp/ofbx.o /tmp/libdeflate.o -o tools/_build/fbx_skin_stats
//   tools/_build/fbx_skin_stats /root/reference/demo/models/ybot/ybot.fbx
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>
#include "ofbx.h"
int main(int argc, char** argv) {
	FILE* f = fopen(argv[1], "rb");
	if (!f) return 1;
	fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
	std::vector<unsigned char> buf(n);
	if (fread(buf.data(), 1, n, f) != (size_t)n) return 1;
	fclose(f);
	ofbx::IScene* scene = ofbx::load(buf.data(), buf.size(), (ofbx::u16)ofbx::LoadFlags::NONE);
	if (!scene) { printf("load failed: %s\n", ofbx::getError()); return 2; }
	printf("meshes %d\n", scene->getMeshCount());
	for (int m = 0; m < scene->getMeshCount(); ++m) {
		const ofbx::Mesh* mesh = scene->getMesh(m);
		const ofbx::GeometryData& gd = mesh->getGeometryData();
		ofbx::Vec3Attributes pos = gd.getPositions();
		const ofbx::Skin* skin = mesh->getSkin();
		printf("mesh %d '%s': position values %d (indexed count %d), skin %p\n", m, mesh->name, pos.values_count, pos.count, (const void*)skin);
		if (!skin) continue;
		const int nv = pos.values_count; // control points
		std::vector<std::vector<std::pair<double,int>>> w(nv);
		printf("  clusters %d\n", skin->getClusterCount());
		for (int c = 0; c < skin->getClusterCount(); ++c) {
			const ofbx::Cluster* cl = skin->getCluster(c);
			for (int k = 0; k < cl->getIndicesCount(); ++k) {
				const int v = cl->getIndices()[k];
				if (v >= 0 && v < nv) w[v].push_back({cl->getWeights()[k], c});
			}
		}
		// per control point: the 4 largest weights (what the importer keeps), in control-point order
		std::vector<int> bones(nv * 4, -1);
		size_t influences = 0, over4 = 0;
		for (int v = 0; v < nv; ++v) {
			auto& a = w[v];
			influences += a.size();
			if (a.size() > 4) ++over4;
			std::sort(a.begin(), a.end(), [](auto& x, auto& y) { return x.first > y.first; });
			for (size_t k = 0; k < a.size() && k < 4; ++k) bones[v * 4 + k] = a[k].second;
		}
		printf("  control points %d, avg influences %.2f, >4 influences %zu\n", nv, (double)influences / nv, over4);
		for (int tile : {1024, 2048, 5120}) {
			size_t tiles = 0, sum = 0, mx = 0;
			for (int b = 0; b < nv; b += tile) {
				std::set<int> s;
				for (int v = b; v < nv && v < b + tile; ++v) for (int k = 0; k < 4; ++k) if (bones[v * 4 + k] >= 0) s.insert(bones[v * 4 + k]);
				++tiles; sum += s.size(); mx = std::max(mx, s.size());
			}
			printf("  tiles of %d control points: %zu tiles, distinct bones per tile avg %.1f max %zu (of %d)\n", tile, tiles, (double)sum / tiles, mx, skin->getClusterCount());
		}
	}
	return 0;
}
