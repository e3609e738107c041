// test_bridges.cpp — drives WorldSync and PoseBridge (lumixengine_amd/host/) the way mi355_plugin.cpp's module does, against the
// in-memory mocks of lumix_compat.h: mirror a World, stage a frame of transform writes, propagate on the GPU, read the transforms
// back; register a skeleton + mesh, gather relative poses through lockPose / unlockPose, run the skin pass, scatter the absolute
// poses back. Reads a scene written by tests/test_gpu_bridges.py, writes the results for comparison with the CPU oracle.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pose_bridge.h"
#include "world_sync.h"

using namespace Lumix;

template <typename T> static bool rd(FILE* f, std::vector<T>& v, size_t n) {
	v.resize(n);
	return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

int main(int argc, char** argv) {
	if (argc < 3) return 2;
	FILE* in = fopen(argv[1], "rb");
	FILE* out = fopen(argv[2], "wb");
	if (!in || !out) return 2;
	uint32_t hdr[8];
	if (fread(hdr, 4, 8, in) != 8) return 2;
	const uint32_t n = hdr[0], n_local = hdr[1], n_world = hdr[2], n_bones = hdr[3], n_verts = hdr[4], n_inst = hdr[5];
	World world;
	std::vector<int32_t> local_e, world_e;
	std::vector<Transform> local_t, world_t;
	if (!rd(in, world.parents, n) || !rd(in, world.transforms, n) || !rd(in, world.locals, n) || !rd(in, local_e, n_local) || !rd(in, local_t, n_local) ||
		!rd(in, world_e, n_world) || !rd(in, world_t, n_world))
		return 2;
	Model model;
	model.bones.resize(n_bones);
	std::vector<LocalRigidTransform> bind;
	if (!rd(in, model.parents, n_bones) || !rd(in, bind, n_bones)) return 2;
	for (uint32_t i = 0; i < n_bones; ++i) model.bones[i].transform = bind[i];
	model.first_nonroot = 1;
	model.meshes.resize(1);
	if (!rd(in, model.meshes[0].vertices.v, n_verts) || !rd(in, model.meshes[0].skin.v, n_verts)) return 2;
	std::vector<Vec3> pose_pos;
	std::vector<Quat> pose_rot;
	if (!rd(in, pose_pos, (size_t)n_inst * n_bones) || !rd(in, pose_rot, (size_t)n_inst * n_bones)) return 2;

	LmxContext* ctx = nullptr;
	if (lmx_ctx_create(0, &ctx) != LMX_OK) { fprintf(stderr, "no device: %s\n", lmx_last_error(nullptr)); return 3; }

	// ---- world ----
	WorldSync sync(ctx);
	if (!sync.build(world)) { fprintf(stderr, "build: %s\n", sync.lastError()); return 4; }
	if (sync.entityCount() != n) return 4;
	for (uint32_t i = 0; i < n_local; ++i) sync.setLocalTransform(EntityRef{local_e[i]}, local_t[i]);
	for (uint32_t i = 0; i < n_world; ++i) sync.setTransform(EntityRef{world_e[i]}, world_t[i]);
	// an entity created or destroyed mid-frame makes the module re-mirror the World: writes staged before that are not in the World
	// yet and must survive the rebuild (the expected output below assumes every staged write is applied)
	if (!sync.build(world)) { fprintf(stderr, "rebuild: %s\n", sync.lastError()); return 4; }
	if (!sync.propagate()) { fprintf(stderr, "propagate: %s\n", sync.lastError()); return 5; }
	std::vector<Transform> result(n), locals(n);
	// the module writes into the engine's own array: const_cast<Transform*>(world.getTransforms())
	if (!sync.readTransforms(const_cast<Transform*>(world.getTransforms()), n) || !sync.readLocalTransforms(locals.data(), n)) return 6;
	fwrite(world.getTransforms(), sizeof(Transform), n, out);
	fwrite(locals.data(), sizeof(Transform), n, out);

	// ---- poses ----
	RenderModule module;
	std::vector<Pose> poses(n_inst);
	module.instances.resize(n_inst);
	std::vector<EntityRef> entities(n_inst);
	std::vector<int32_t> model_ids(n_inst), mesh_ids(n_inst);
	std::vector<uint32_t> bone_counts(n_inst, n_bones);
	PoseBridge bridge(ctx);
	const int32_t model_id = bridge.addModel(model);
	const int32_t mesh_id = bridge.addMesh(model.getMesh(0));
	if (model_id < 0 || mesh_id < 0) { fprintf(stderr, "skin setup: %s\n", bridge.lastError()); return 7; }
	for (uint32_t i = 0; i < n_inst; ++i) {
		poses[i].count = n_bones;
		poses[i].positions = &pose_pos[(size_t)i * n_bones];
		poses[i].rotations = &pose_rot[(size_t)i * n_bones];
		module.instances[i].model = &model;
		module.instances[i].pose = &poses[i];
		entities[i] = EntityRef{(int32_t)i};
		model_ids[i] = model_id;
		mesh_ids[i] = mesh_id;
	}
	if (!bridge.setInstances(entities.data(), model_ids.data(), mesh_ids.data(), bone_counts.data(), n_inst)) return 8;
	if (!bridge.gather(module) || !bridge.run() || !bridge.scatter(module)) { fprintf(stderr, "skin: %s\n", bridge.lastError()); return 9; }
	for (uint32_t i = 0; i < n_inst; ++i)
		if (!poses[i].is_absolute) return 10;
	fwrite(pose_pos.data(), sizeof(Vec3), pose_pos.size(), out);
	fwrite(pose_rot.data(), sizeof(Quat), pose_rot.size(), out);
	std::vector<float> verts((size_t)n_verts * 3);
	for (uint32_t i = 0; i < n_inst; ++i) {
		if (lmx_skin_read_vertices(ctx, i, verts.data(), n_verts) != LMX_OK) return 11;
		fwrite(verts.data(), sizeof(float), verts.size(), out);
	}
	lmx_ctx_destroy(ctx);
	fclose(in);
	fclose(out);
	return 0;
}
