// lmx_capi_skin.hip — skinning entry points (include/lumix_mi355.h, "skinning" section): model / mesh registration, the
// instance table, pose upload, and the pose -> palette -> vertex launches.
#include "lmx_context.h"

using namespace lmx;

extern "C" {

int lmx_skin_add_model(LmxContext* ctx, uint32_t n_bones, const int16_t* parents, const LmxLocalRigidTransform* bind, int32_t first_nonroot,
	uint32_t* out_model) {
	LMX_CHECK_CTX(ctx);
	if (!n_bones || n_bones > LMX_MAX_BONES) return fail(ctx, LMX_ERR_CAPACITY, "n_bones %u not in [1,%d] (Model::Bone::MAX_COUNT)", n_bones, LMX_MAX_BONES);
	if (!parents || !bind) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	if (first_nonroot < 0) first_nonroot = (int32_t)n_bones;
	SkinState& sk = ctx->skin;
	SkinModel m;
	m.bone_offset = (uint32_t)sk.parents.size();
	m.n_bones = n_bones;
	m.first_nonroot = first_nonroot;
	m.max_depth = 0;
	std::vector<uint8_t> depth(n_bones, 0);
	for (uint32_t i = 0; i < n_bones; ++i) {
		const int32_t p = parents[i];
		if (p >= (int32_t)i) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "parents[%u] = %d must precede the bone (model.cpp:381-384)", i, p);
		if ((int32_t)i >= first_nonroot && p < 0) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bone %u >= first_nonroot has no parent", i);
		// bones below first_nonroot are never touched by Pose::computeAbsolute (pose.cpp:66): depth 0
		depth[i] = ((int32_t)i >= first_nonroot) ? (uint8_t)(depth[p] + 1) : 0;
		m.max_depth = std::max<uint32_t>(m.max_depth, depth[i]);
	}
	// bones that Pose::computeAbsolute touches, sorted by depth, + per-depth offsets: the level walk of k_pose_palette
	m.lv_items_offset = (uint32_t)sk.level_items.size();
	m.lv_off_offset = (uint32_t)sk.level_off.size();
	sk.level_off.push_back(0);
	for (uint32_t d = 1; d <= m.max_depth; ++d) {
		for (uint32_t i = 0; i < n_bones; ++i)
			if (depth[i] == d) sk.level_items.push_back(i | ((uint32_t)parents[i] << 16));
		sk.level_off.push_back((uint16_t)(sk.level_items.size() - m.lv_items_offset));
	}
	for (uint32_t i = 0; i < n_bones; ++i) {
		V3 ip;
		Q4 ir;
		invert_rigid(V3{bind[i].pos[0], bind[i].pos[1], bind[i].pos[2]}, Q4{bind[i].rot[0], bind[i].rot[1], bind[i].rot[2], bind[i].rot[3]}, &ip, &ir);
		sk.parents.push_back(parents[i]);
		sk.depth.push_back(depth[i]);
		sk.inv_pos.push_back(ip.x);
		sk.inv_pos.push_back(ip.y);
		sk.inv_pos.push_back(ip.z);
		sk.inv_rot.push_back(make_float4(ir.x, ir.y, ir.z, ir.w));
	}
	sk.models.push_back(m);
	sk.models_dirty = true;
	if (out_model) *out_model = (uint32_t)sk.models.size() - 1;
	return LMX_OK;
}

int lmx_skin_add_mesh(LmxContext* ctx, uint32_t n_verts, const float* positions_xyz, const LmxSkin* skin, uint32_t* out_mesh) {
	LMX_CHECK_CTX(ctx);
	if (!n_verts || !positions_xyz || !skin) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "empty mesh / null input array");
	SkinState& sk = ctx->skin;
	for (uint32_t v = 0; v < n_verts; ++v) // validate before anything is appended: a refused mesh leaves the tables untouched
		for (int k = 0; k < 4; ++k)
			if (skin[v].indices[k] < 0 || skin[v].indices[k] >= LMX_MAX_BONES) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "skin[%u].indices[%d] = %d out of range", v, k, skin[v].indices[k]);
	SkinMesh m;
	m.vert_offset = (uint32_t)(sk.mesh.size() / 2);
	m.n_verts = n_verts;
	m.max_bone = 0;
	for (uint32_t v = 0; v < n_verts; ++v)
		for (int k = 0; k < 4; ++k) m.max_bone = std::max<uint32_t>(m.max_bone, (uint32_t)skin[v].indices[k]);
	static_assert(LMX_MAX_BONES <= 256, "bone indices are packed as u8");
	// (no exact-size reserve here: it defeats the vector's geometric growth - the 10 000th mesh of BASELINE config 3's distinct-mesh variant would copy
	// 3.2 GB of records to append 320 KB, 16 TB over the whole scene)
	if (sk.mesh.capacity() < sk.mesh.size() + 2 * (size_t)n_verts) sk.mesh.reserve(std::max(2 * sk.mesh.capacity(), sk.mesh.size() + 2 * (size_t)n_verts));
	for (uint32_t v = 0; v < n_verts; ++v) {
		const uint32_t idx = (uint32_t)skin[v].indices[0] | ((uint32_t)skin[v].indices[1] << 8) | ((uint32_t)skin[v].indices[2] << 16) | ((uint32_t)skin[v].indices[3] << 24);
		float idx_bits;
		std::memcpy(&idx_bits, &idx, 4);
		sk.mesh.push_back(make_float4(skin[v].weights[0], skin[v].weights[1], skin[v].weights[2], skin[v].weights[3]));
		sk.mesh.push_back(make_float4(positions_xyz[3 * (size_t)v], positions_xyz[3 * (size_t)v + 1], positions_xyz[3 * (size_t)v + 2], idx_bits));
	}
	// k_skin_shared's tiling of the mesh (<= SKIN_SHARED_TILE_VERTS vertices per tile, equal shares rounded up to whole waves) and,
	// per tile, the bones it references + a copy of its records whose index bytes point into that list: the kernel stages only those
	// bones' palette rows (a tile of a real character touches a fraction of the skeleton; tools/fbx_skin_stats.cpp)
	m.n_tiles = (n_verts + SKIN_SHARED_TILE_VERTS - 1) / SKIN_SHARED_TILE_VERTS;
	m.tile_verts = ((n_verts + m.n_tiles - 1) / m.n_tiles + 63u) & ~63u;
	m.tiles_at = (uint32_t)sk.tiles.size();
	sk.mesh_local.resize(sk.mesh.size());
	for (uint32_t t = 0; t < m.n_tiles; ++t) {
		const uint32_t v0 = t * m.tile_verts, v1 = std::min(n_verts, (t + 1) * m.tile_verts);
		int16_t local_of[LMX_MAX_BONES];
		std::fill(local_of, local_of + LMX_MAX_BONES, (int16_t)-1);
		SkinTile tile{(uint32_t)sk.tile_bones.size(), 0};
		for (uint32_t v = v0; v < v1; ++v) {
			uint32_t idx = 0;
			for (int k = 0; k < 4; ++k) {
				const int b = skin[v].indices[k];
				if (local_of[b] < 0) {
					local_of[b] = (int16_t)tile.n_bones++;
					sk.tile_bones.push_back((uint8_t)b);
				}
				idx |= (uint32_t)local_of[b] << (8 * k);
			}
			float idx_bits;
			std::memcpy(&idx_bits, &idx, 4);
			const size_t at = 2 * ((size_t)m.vert_offset + v);
			sk.mesh_local[at] = sk.mesh[at];
			sk.mesh_local[at + 1] = make_float4(sk.mesh[at + 1].x, sk.mesh[at + 1].y, sk.mesh[at + 1].z, idx_bits);
		}
		sk.tiles.push_back(tile);
	}
	sk.meshes.push_back(m);
	sk.meshes_dirty = true;
	if (out_mesh) *out_mesh = (uint32_t)sk.meshes.size() - 1;
	return LMX_OK;
}

static int skin_upload_static(LmxContext* ctx) {
	SkinState& sk = ctx->skin;
	if (sk.models_dirty) {
		const size_t nb = sk.parents.size();
		LMX_HIP(ctx, sk.d_parents.reserve(nb));
		LMX_HIP(ctx, sk.d_inv_pos.reserve(nb * 3));
		LMX_HIP(ctx, sk.d_inv_rot.reserve(nb));
		LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		LMX_HIP(ctx, hipMemcpy(sk.d_parents.p, sk.parents.data(), nb * sizeof(int16_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(sk.d_inv_pos.p, sk.inv_pos.data(), nb * 3 * sizeof(float), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(sk.d_inv_rot.p, sk.inv_rot.data(), nb * sizeof(float4), hipMemcpyHostToDevice));
		// (+ 256: k_pose_palette's lanes read a model's item list 64 words at a time, up to one bone count past its end - never used, never out of the buffer)
		LMX_HIP(ctx, sk.d_level_items.reserve(sk.level_items.size() + 256));
		LMX_HIP(ctx, sk.d_level_off.reserve(std::max<size_t>(sk.level_off.size(), 1)));
		if (!sk.level_items.empty()) LMX_HIP(ctx, hipMemcpy(sk.d_level_items.p, sk.level_items.data(), sk.level_items.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(sk.d_level_off.p, sk.level_off.data(), sk.level_off.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
		sk.models_dirty = false;
	}
	if (sk.meshes_dirty) {
		LMX_HIP(ctx, sk.d_mesh.reserve(std::max<size_t>(sk.mesh.size(), 1)));
		LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		LMX_HIP(ctx, hipMemcpy(sk.d_mesh.p, sk.mesh.data(), sk.mesh.size() * sizeof(float4), hipMemcpyHostToDevice));
		LMX_HIP(ctx, sk.d_mesh_local.reserve(std::max<size_t>(sk.mesh_local.size(), 1)));
		LMX_HIP(ctx, sk.d_tile_bones.reserve(std::max<size_t>(sk.tile_bones.size(), 1)));
		LMX_HIP(ctx, hipMemcpy(sk.d_mesh_local.p, sk.mesh_local.data(), sk.mesh_local.size() * sizeof(float4), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(sk.d_tile_bones.p, sk.tile_bones.data(), sk.tile_bones.size(), hipMemcpyHostToDevice));
		sk.meshes_dirty = false;
	}
	return LMX_OK;
}

int lmx_skin_set_instances(LmxContext* ctx, uint32_t n, const uint32_t* model, const uint32_t* mesh) {
	LMX_CHECK_CTX(ctx);
	if (n && (!model || !mesh)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	SkinState& sk = ctx->skin;
	std::vector<SkinInstance> inst(n);
	size_t bones = 0, verts = 0;
	uint32_t max_verts = 0;
	for (uint32_t i = 0; i < n; ++i) {
		if (model[i] >= sk.models.size() || mesh[i] >= sk.meshes.size()) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "instance %u: unknown model/mesh id", i);
		const SkinModel& mo = sk.models[model[i]];
		const SkinMesh& me = sk.meshes[mesh[i]];
		SkinInstance& in = inst[i];
		// the vertex kernels index the instance's palette in LDS with the mesh's bone indices, unchecked
		if (me.max_bone >= mo.n_bones) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "instance %u: its mesh references bone %u, its model has %u bones", i, me.max_bone, mo.n_bones);
		if (bones + mo.n_bones > 0xffffffffull || verts + me.n_verts > 0xffffffffull) return fail(ctx, LMX_ERR_CAPACITY, "instance table exceeds 2^32 bones or vertices");
		in.bone_offset = (uint32_t)bones;
		in.n_bones = mo.n_bones;
		in.model_offset = mo.bone_offset;
		in.first_nonroot = mo.first_nonroot;
		in.vert_offset = me.vert_offset;
		in.n_verts = me.n_verts;
		in.out_offset = (uint32_t)verts;
		in.max_depth = mo.max_depth;
		in.lv_items_offset = mo.lv_items_offset;
		in.lv_off_offset = mo.lv_off_offset;
		bones += mo.n_bones;
		verts += me.n_verts;
		max_verts = std::max(max_verts, me.n_verts);
	}
	// pose groups: runs of consecutive instances of one model, at most 4 / 2 / 1 (<= 64 / 128 / 196 bones) per group, stored by
	// capacity class (one launch of k_pose_palette<NBMAX> per class)
	std::vector<PoseGroup> groups[3];
	for (uint32_t i = 0; i < n;) {
		const uint32_t nbm = sk.models[model[i]].n_bones;
		const uint32_t cls = nbm <= 64 ? 0u : (nbm <= 128 ? 1u : 2u);
		const uint32_t cap = std::max(1u, POSE_GROUP_CAP >> cls);
		uint32_t c = 1;
		while (i + c < n && c < cap && model[i + c] == model[i]) ++c;
		groups[cls].push_back(PoseGroup{i, c});
		i += c;
	}
	sk.groups.clear();
	for (int c = 0; c < 3; ++c) {
		sk.n_groups[c] = (uint32_t)groups[c].size();
		sk.groups.insert(sk.groups.end(), groups[c].begin(), groups[c].end());
	}
	// skinning work: runs of consecutive instances that share a mesh (and a bone count) go to k_skin_shared, which keeps the
	// vertex records in registers across the run; everything else goes to k_skin_vertices, one instance at a time
	sk.chunks.clear();
	sk.solo.clear();
	sk.solo_max_verts = 0;
	{
		struct Run { uint32_t first, count, tiles, mesh; };
		std::vector<Run> runs;
		sk.runs.clear();
		sk.multi_built = 0;
		uint64_t run_tiles = 0;
		for (uint32_t i = 0; i < n;) {
			uint32_t c = 1;
			while (i + c < n && mesh[i + c] == mesh[i] && inst[i + c].n_bones == inst[i].n_bones) ++c;
			if (c >= 2 && inst[i].n_verts >= 2048) {
				const uint32_t tiles = sk.meshes[mesh[i]].n_tiles;
				runs.push_back(Run{i, c, tiles, mesh[i]});
				sk.runs.push_back(SkinState::Run{i, c, mesh[i]});
				run_tiles += (uint64_t)c * tiles;
			} else {
				for (uint32_t k = 0; k < c; ++k) {
					sk.solo.push_back(i + k);
					sk.solo_max_verts = std::max(sk.solo_max_verts, inst[i + k].n_verts);
				}
			}
			i += c;
		}
		// instances per block: as many as still leave ~6 blocks per CU (the mesh tile is loaded once per block)
		const uint32_t per_block = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, run_tiles / 1536));
		for (const Run& r : runs) {
			const uint32_t nv = inst[r.first].n_verts;
			const SkinMesh& me = sk.meshes[r.mesh];
			const uint32_t tile_verts = me.tile_verts;
			for (uint32_t f = 0; f < r.count; f += per_block)
				for (uint32_t t = 0; t < r.tiles; ++t) {
					if (t * tile_verts >= nv) continue;
					const SkinTile& tile = sk.tiles[me.tiles_at + t];
					sk.chunks.push_back(SkinChunk{r.first + f, std::min(per_block, r.count - f), t * tile_verts, std::min(nv, (t + 1) * tile_verts), me.vert_offset,
						tile.bones_at, tile.n_bones, 0u});
				}
		}
		if (sk.chunks.empty()) sk.solo.clear(); // every instance: identity index
	}
	sk.inst.swap(inst);
	sk.bones_total = bones;
	sk.verts_total = verts;
	sk.max_verts = max_verts;
	sk.poses_uploaded = false;
	sk.pose_is_absolute = false;
	sk.borrowed_pos = nullptr;
	sk.borrowed_rot = nullptr;
	LMX_HIP(ctx, sk.d_inst.reserve(std::max<size_t>(n, 1)));
	LMX_HIP(ctx, sk.d_pose_pos.reserve(std::max<size_t>(bones * 3, 1)));
	LMX_HIP(ctx, sk.d_pose_rot.reserve(std::max<size_t>(bones, 1)));
	LMX_HIP(ctx, sk.d_palette.reserve(std::max<size_t>(bones * 3, 1)));
	LMX_HIP(ctx, sk.d_out.reserve(std::max<size_t>(verts * 3, 1)));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (n) LMX_HIP(ctx, hipMemcpy(sk.d_inst.p, sk.inst.data(), (size_t)n * sizeof(SkinInstance), hipMemcpyHostToDevice));
	LMX_HIP(ctx, sk.d_chunks.reserve(std::max<size_t>(sk.chunks.size(), 1)));
	LMX_HIP(ctx, sk.d_solo.reserve(std::max<size_t>(sk.solo.size(), 1)));
	if (!sk.chunks.empty()) LMX_HIP(ctx, hipMemcpy(sk.d_chunks.p, sk.chunks.data(), sk.chunks.size() * sizeof(SkinChunk), hipMemcpyHostToDevice));
	if (!sk.solo.empty()) LMX_HIP(ctx, hipMemcpy(sk.d_solo.p, sk.solo.data(), sk.solo.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	LMX_HIP(ctx, sk.d_groups.reserve(std::max<size_t>(sk.groups.size(), 1)));
	if (n) LMX_HIP(ctx, hipMemcpy(sk.d_groups.p, sk.groups.data(), sk.groups.size() * sizeof(PoseGroup), hipMemcpyHostToDevice));
	return LMX_OK;
}

int lmx_skin_upload_poses(LmxContext* ctx, const float* positions, const float* rotations, size_t n_bones_total) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (n_bones_total != sk.bones_total) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "expected %zu bones over all instances, got %zu", sk.bones_total, n_bones_total);
	if (!n_bones_total) return LMX_OK;
	if (!positions || !rotations) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	LMX_HIP(ctx, hipMemcpy(sk.d_pose_pos.p, positions, n_bones_total * 3 * sizeof(float), hipMemcpyHostToDevice));
	LMX_HIP(ctx, hipMemcpy(sk.d_pose_rot.p, rotations, n_bones_total * sizeof(float4), hipMemcpyHostToDevice));
	sk.poses_uploaded = true;
	sk.pose_is_absolute = false;
	return LMX_OK;
}

int lmx_skin_upload_poses_device(LmxContext* ctx, const void* d_positions, const void* d_rotations, size_t n_bones_total) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (n_bones_total != sk.bones_total) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "expected %zu bones over all instances, got %zu", sk.bones_total, n_bones_total);
	if (!n_bones_total) return LMX_OK;
	if (!d_positions || !d_rotations) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null device pointer");
	LMX_HIP(ctx, hipMemcpyAsync(sk.d_pose_pos.p, d_positions, n_bones_total * 3 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
	LMX_HIP(ctx, hipMemcpyAsync(sk.d_pose_rot.p, d_rotations, n_bones_total * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
	sk.poses_uploaded = true;
	sk.pose_is_absolute = false;
	return LMX_OK;
}

// Pose::blend(rhs, weight) (renderer/pose.cpp:30-41) of the library's relative poses with a second set of the same layout
static int skin_blend(LmxContext* ctx, const float* d_pos, const float4* d_rot, size_t n_bones_total, float weight) {
	SkinState& sk = ctx->skin;
	if (!sk.poses_uploaded || sk.pose_is_absolute || sk.borrowed_pos) return fail(ctx, LMX_ERR_NOT_BUILT, "the library holds no relative poses to blend into (upload / lmx_anim_update first)");
	if (weight <= 0.001f) return LMX_OK; // pose.cpp:33
	weight = weight < 0.0f ? 0.0f : (weight > 1.0f ? 1.0f : weight); // clamp, pose.cpp:34
	LMX_HIP(ctx, launch_pose_blend(ctx->stream, sk.d_pose_pos.p, sk.d_pose_rot.p, d_pos, d_rot, n_bones_total, weight));
	return LMX_OK;
}

int lmx_skin_blend_poses_device(LmxContext* ctx, const void* d_positions, const void* d_rotations, size_t n_bones_total, float weight) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (n_bones_total != sk.bones_total) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "expected %zu bones over all instances, got %zu", sk.bones_total, n_bones_total);
	if (!n_bones_total) return LMX_OK;
	if (!d_positions || !d_rotations) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null device pointer");
	if (!(weight == weight)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "weight is NaN");
	return skin_blend(ctx, (const float*)d_positions, (const float4*)d_rotations, n_bones_total, weight);
}

int lmx_skin_blend_poses(LmxContext* ctx, const float* positions, const float* rotations, size_t n_bones_total, float weight) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (n_bones_total != sk.bones_total) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "expected %zu bones over all instances, got %zu", sk.bones_total, n_bones_total);
	if (!n_bones_total) return LMX_OK;
	if (!positions || !rotations) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	if (!(weight == weight)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "weight is NaN");
	LMX_HIP(ctx, sk.d_blend_pos.reserve(n_bones_total * 3));
	LMX_HIP(ctx, sk.d_blend_rot.reserve(n_bones_total));
	LMX_HIP(ctx, hipMemcpyAsync(sk.d_blend_pos.p, positions, n_bones_total * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
	LMX_HIP(ctx, hipMemcpyAsync(sk.d_blend_rot.p, rotations, n_bones_total * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
	const int rc = skin_blend(ctx, sk.d_blend_pos.p, sk.d_blend_rot.p, n_bones_total, weight);
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream)); // the caller's arrays may be pageable: do not return before they are read
	return rc;
}

int lmx_skin_set_pose_source_device(LmxContext* ctx, const void* d_positions, const void* d_rotations, size_t n_bones_total) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if ((d_positions == nullptr) != (d_rotations == nullptr)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "set both pointers or neither");
	if (d_positions && n_bones_total != sk.bones_total) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "expected %zu bones over all instances, got %zu", sk.bones_total, n_bones_total);
	sk.borrowed_pos = (const float*)d_positions;
	sk.borrowed_rot = (const float4*)d_rotations;
	sk.poses_uploaded = d_positions != nullptr;
	return LMX_OK;
}

int lmx_skin_set_mode(LmxContext* ctx, int mode) {
	LMX_CHECK_CTX(ctx);
	if (mode != LMX_SKIN_FUSED && mode != LMX_SKIN_EXACT && mode != LMX_SKIN_DQS) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "unknown skin mode %d", mode);
	ctx->skin.mode = mode;
	return LMX_OK;
}

int lmx_skin_set_option(LmxContext* ctx, int option, int value) {
	LMX_CHECK_CTX(ctx);
	if (option != LMX_SKIN_OPT_INSTANCES_PER_BLOCK) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "unknown skin option %d", option);
	if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "instances per block %d not in {0, 1, 2, 4, 8, 16}", value);
	ctx->skin.multi = (uint32_t)value;
	return LMX_OK;
}

// k_skin_multi's work items: every run in groups of I instances, each group's vertices in as many ranges as give the chip ~3000 blocks
static int skin_build_multi_chunks(LmxContext* ctx) {
	SkinState& sk = ctx->skin;
	if (sk.multi_built == sk.multi) return LMX_OK;
	sk.multi_chunks.clear();
	uint64_t groups = 0;
	auto staged = [&](const SkinState::Run& r) { return std::min(sk.inst[r.first].n_bones, sk.meshes[r.mesh].max_bone + 1); }; // bones a block stages
	sk.multi_max_stage = 0;
	for (const SkinState::Run& r : sk.runs) {
		const uint32_t per = skin_multi_instances(sk.multi, staged(r));
		groups += (r.count + per - 1) / per;
		sk.multi_max_stage = std::max(sk.multi_max_stage, skin_multi_lds_slots(staged(r)));
	}
	for (const SkinState::Run& r : sk.runs) {
		const SkinInstance& in = sk.inst[r.first];
		const uint32_t per = skin_multi_instances(sk.multi, staged(r));
		uint32_t splits = (uint32_t)std::max<uint64_t>(1, (3072 + groups - 1) / std::max<uint64_t>(groups, 1));
		splits = std::min(splits, std::max(1u, in.n_verts / 1024u)); // a range is worth its 48 KiB palette staging from ~1000 vertices x I instances on
		const uint32_t range = ((in.n_verts + splits - 1) / splits + 63u) & ~63u;
		for (uint32_t f = 0; f < r.count; f += per)
			for (uint32_t v = 0; v < in.n_verts; v += range)
				sk.multi_chunks.push_back(SkinMultiChunk{sk.inst[r.first + f].bone_offset, in.n_bones, std::min(per, r.count - f), v, std::min(in.n_verts, v + range), in.vert_offset,
					in.n_verts, sk.inst[r.first + f].out_offset, staged(r)});
	}
	LMX_HIP(ctx, sk.d_multi_chunks.reserve(std::max<size_t>(sk.multi_chunks.size(), 1)));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream)); // (a launch of the previous frame may still read the old list)
	if (!sk.multi_chunks.empty()) LMX_HIP(ctx, hipMemcpy(sk.d_multi_chunks.p, sk.multi_chunks.data(), sk.multi_chunks.size() * sizeof(SkinMultiChunk), hipMemcpyHostToDevice));
	sk.multi_built = sk.multi;
	return LMX_OK;
}

int lmx_skin_run(LmxContext* ctx) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (sk.inst.empty()) return LMX_OK;
	if (!sk.poses_uploaded) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_skin_upload_poses has not been called for this instance table");
	if (int rc = skin_upload_static(ctx)) return rc;
	const uint32_t n = (uint32_t)sk.inst.size();
	{
		ProfScope ps(ctx, LMX_K_POSE_PALETTE);
		const bool dual_quats = sk.want_dual_quats || sk.mode == LMX_SKIN_DQS; // the DQS vertex blend reads the dual-quaternion palette
		if (dual_quats) LMX_HIP(ctx, sk.d_dual_quats.reserve(std::max<size_t>(sk.bones_total * 2, 1)));
		const float* rel_pos = sk.borrowed_pos ? sk.borrowed_pos : sk.d_pose_pos.p;
		const float4* rel_rot = sk.borrowed_rot ? sk.borrowed_rot : sk.d_pose_rot.p;
		LMX_HIP(ctx, launch_pose_palette(ctx->stream, sk.d_inst.p, sk.d_groups.p, sk.n_groups, rel_pos, rel_rot, sk.pose_writeback ? sk.d_pose_pos.p : nullptr,
			sk.pose_writeback ? sk.d_pose_rot.p : nullptr, sk.d_level_items.p, sk.d_level_off.p, sk.d_inv_pos.p, sk.d_inv_rot.p, sk.d_palette.p, dual_quats ? sk.d_dual_quats.p : nullptr));
	}
	{
		ProfScope ps(ctx, LMX_K_SKIN_VERTICES);
		const float4* vertex_palette = sk.mode == LMX_SKIN_DQS ? sk.d_dual_quats.p : sk.d_palette.p;
		if (!sk.chunks.empty() && sk.multi) { // runs of instances that share a mesh, several instances per block (every mode)
			if (int rc = skin_build_multi_chunks(ctx)) return rc;
			LMX_HIP(ctx, launch_skin_multi(ctx->stream, sk.multi, sk.d_multi_chunks.p, (uint32_t)sk.multi_chunks.size(), sk.multi_max_stage, sk.d_mesh.p, vertex_palette, sk.d_out.p, sk.mode));
			LMX_HIP(ctx, launch_skin_vertices(ctx->stream, sk.d_inst.p, sk.d_solo.p, (uint32_t)sk.solo.size(), sk.solo_max_verts, sk.d_mesh.p, vertex_palette, sk.d_out.p, sk.mode));
		} else if (sk.chunks.empty() || sk.mode == LMX_SKIN_DQS) { // (DQS: k_skin_shared's resident records leave too few registers for the dual-quaternion blend)
			LMX_HIP(ctx, launch_skin_vertices(ctx->stream, sk.d_inst.p, nullptr, n, sk.max_verts, sk.d_mesh.p, vertex_palette,
				sk.d_out.p, sk.mode));
		} else {
			LMX_HIP(ctx, launch_skin_shared(ctx->stream, sk.d_inst.p, sk.d_chunks.p, (uint32_t)sk.chunks.size(), sk.d_mesh_local.p, sk.d_tile_bones.p,
				vertex_palette, sk.d_out.p, sk.mode));
			LMX_HIP(ctx, launch_skin_vertices(ctx->stream, sk.d_inst.p, sk.d_solo.p, (uint32_t)sk.solo.size(), sk.solo_max_verts, sk.d_mesh.p, vertex_palette, sk.d_out.p, sk.mode));
		}
	}
	// the library's poses are absolute now; running again needs fresh relative poses (Pose::is_absolute, pose.cpp:64) unless
	// a borrowed source provides them every frame
	sk.poses_uploaded = sk.borrowed_pos != nullptr || !sk.pose_writeback;
	sk.pose_is_absolute = sk.pose_writeback;
	return LMX_OK;
}

int lmx_skin_read_vertices(LmxContext* ctx, uint32_t instance, float* out_xyz, uint32_t cap_verts) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (instance >= sk.inst.size() || !out_xyz) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad instance/out");
	const SkinInstance& in = sk.inst[instance];
	if (cap_verts < in.n_verts) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u vertices", in.n_verts);
	LMX_HIP(ctx, hipMemcpyAsync(out_xyz, sk.d_out.p + (size_t)in.out_offset * 3, (size_t)in.n_verts * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

int lmx_skin_read_vertices_range(LmxContext* ctx, uint32_t first_instance, uint32_t n_instances, float* out_xyz, size_t cap_verts) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (!out_xyz || (size_t)first_instance + n_instances > sk.inst.size()) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad instance range/out");
	if (!n_instances) return LMX_OK;
	const SkinInstance& a = sk.inst[first_instance];
	const SkinInstance& b = sk.inst[first_instance + n_instances - 1];
	const size_t n = (size_t)b.out_offset + b.n_verts - a.out_offset; // outputs are laid out in instance order
	if (cap_verts < n) return fail(ctx, LMX_ERR_CAPACITY, "need room for %zu vertices", n);
	LMX_HIP(ctx, hipMemcpyAsync(out_xyz, sk.d_out.p + (size_t)a.out_offset * 3, n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

int lmx_skin_device_output(LmxContext* ctx, const float** d_xyz, size_t* n_verts_total) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (!d_xyz || !n_verts_total) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null out pointer");
	if (sk.inst.empty()) return fail(ctx, LMX_ERR_NOT_BUILT, "no skinned instances");
	*d_xyz = sk.d_out.p;
	*n_verts_total = (size_t)sk.inst.back().out_offset + sk.inst.back().n_verts;
	return LMX_OK;
}

int lmx_skin_read_palette(LmxContext* ctx, uint32_t instance, LmxMatrix* out, uint32_t cap_bones) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (instance >= sk.inst.size() || !out) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad instance/out");
	const SkinInstance& in = sk.inst[instance];
	if (cap_bones < in.n_bones) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u bones", in.n_bones);
	// the palette is kept as the 3 rows evaluateSkin reads (48 B per bone); the constant 4th row is re-attached here
	LMX_HIP(ctx, sk.d_palette_expanded.reserve(in.n_bones * 4));
	LMX_HIP(ctx, launch_palette_expand(ctx->stream, sk.d_palette.p + (size_t)in.bone_offset * 3, in.n_bones, sk.d_palette_expanded.p));
	LMX_HIP(ctx, hipMemcpyAsync(out, sk.d_palette_expanded.p, (size_t)in.n_bones * sizeof(LmxMatrix), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

int lmx_skin_set_pose_writeback(LmxContext* ctx, int enable) {
	LMX_CHECK_CTX(ctx);
	ctx->skin.pose_writeback = enable != 0;
	return LMX_OK;
}

int lmx_skin_enable_dual_quats(LmxContext* ctx, int enable) {
	LMX_CHECK_CTX(ctx);
	ctx->skin.want_dual_quats = enable != 0;
	return LMX_OK;
}

int lmx_skin_read_dual_quats(LmxContext* ctx, uint32_t instance, float* out, uint32_t cap_bones) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (instance >= sk.inst.size() || !out) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad instance/out");
	if (!(sk.want_dual_quats || sk.mode == LMX_SKIN_DQS) || !sk.d_dual_quats.p) return fail(ctx, LMX_ERR_NOT_BUILT, "dual-quaternion palette not enabled before lmx_skin_run");
	const SkinInstance& in = sk.inst[instance];
	if (cap_bones < in.n_bones) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u bones", in.n_bones);
	LMX_HIP(ctx, hipMemcpyAsync(out, sk.d_dual_quats.p + (size_t)in.bone_offset * 2, (size_t)in.n_bones * 8 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

int lmx_skin_read_pose(LmxContext* ctx, uint32_t instance, float* out_pos, float* out_rot, uint32_t cap_bones) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (instance >= sk.inst.size()) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad instance");
	const SkinInstance& in = sk.inst[instance];
	if (cap_bones < in.n_bones) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u bones", in.n_bones);
	if (!sk.pose_is_absolute) return fail(ctx, LMX_ERR_NOT_BUILT, "no absolute pose: lmx_skin_run has not run, or pose write-back is disabled");
	if (out_pos) LMX_HIP(ctx, hipMemcpyAsync(out_pos, sk.d_pose_pos.p + (size_t)in.bone_offset * 3, (size_t)in.n_bones * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	if (out_rot) LMX_HIP(ctx, hipMemcpyAsync(out_rot, sk.d_pose_rot.p + in.bone_offset, (size_t)in.n_bones * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}


} // extern "C"
