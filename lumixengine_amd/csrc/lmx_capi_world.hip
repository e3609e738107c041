// lmx_capi_world.hip — batch form of World's transform hierarchy behind the C ABI (include/lumix_mi355.h, "world transforms").
// Host side: BFS order of the hierarchy ((level, parent slot) slots), staging of new transforms, one k_xform_level launch per
// level, and the RenderModule "moved" binding that refreshes the culling system's dynamic set on the device.
#include "lmx_context.h"

using namespace lmx;

extern "C" {

} // extern "C"

namespace {

// (Re)build the slot order for `parent` and upload: transforms[e] = world transform for roots, Hierarchy::local_transform for
// children; world_all (optional) additionally seeds the world values of every entity (re-parenting keeps them).
int world_rebuild(LmxContext* ctx, uint32_t n, const int32_t* parent, const LmxTransform* transforms, const LmxTransform* world_all) {
	WorldState& w = ctx->world;
	w.built = false;
	w.n_sub_runs = 0;
	// children lists (CSR by parent), then BFS from the roots: slot order = (level, parent slot)
	std::vector<uint32_t> child_start((size_t)n + 1, 0);
	for (uint32_t e = 0; e < n; ++e) {
		if (parent[e] >= (int32_t)n || parent[e] == (int32_t)e) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "parent[%u] = %d invalid", e, parent[e]);
		if (parent[e] >= 0) child_start[(size_t)parent[e] + 1]++;
	}
	for (uint32_t e = 0; e < n; ++e) child_start[e + 1] += child_start[e];
	std::vector<uint32_t> child_list(child_start[n]);
	{
		std::vector<uint32_t> cursor(child_start.begin(), child_start.end() - 1);
		for (uint32_t e = 0; e < n; ++e)
			if (parent[e] >= 0) child_list[cursor[parent[e]]++] = e;
	}
	std::vector<int32_t> entity_of_slot;
	std::vector<uint32_t> level_start;
	entity_of_slot.reserve(n);
	level_start.push_back(0);
	for (uint32_t e = 0; e < n; ++e)
		if (parent[e] < 0) entity_of_slot.push_back((int32_t)e);
	size_t level_begin = 0;
	while (level_begin < entity_of_slot.size()) {
		const size_t level_end = entity_of_slot.size();
		level_start.push_back((uint32_t)level_end);
		for (size_t s = level_begin; s < level_end; ++s) {
			const uint32_t e = (uint32_t)entity_of_slot[s];
			for (uint32_t k = child_start[e]; k < child_start[e + 1]; ++k) entity_of_slot.push_back((int32_t)child_list[k]);
		}
		level_begin = level_end;
	}
	if (entity_of_slot.size() != n) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "hierarchy contains a cycle (%zu of %u entities reachable)", entity_of_slot.size(), n);
	w.entity_of_slot.swap(entity_of_slot);
	w.level_start.swap(level_start);
	w.parent.assign(parent, parent + n);
	w.slot_of_entity.assign(n, -1);
	for (uint32_t s = 0; s < n; ++s) w.slot_of_entity[w.entity_of_slot[s]] = (int32_t)s;
	w.parent_slot.assign(n, -1);
	for (uint32_t s = 0; s < n; ++s) {
		const int32_t p = parent[w.entity_of_slot[s]];
		w.parent_slot[s] = p < 0 ? -1 : w.slot_of_entity[p];
	}
	w.n = n;
	const size_t cap = std::max(n, 1u);
	for (auto& b : w.pos) LMX_HIP(ctx, b.reserve(cap));
	for (auto& b : w.rot) LMX_HIP(ctx, b.reserve(cap));
	for (auto& b : w.scl) LMX_HIP(ctx, b.reserve(cap));
	LMX_HIP(ctx, w.d_parent_slot.reserve(cap));
	LMX_HIP(ctx, w.d_slot_of_entity.reserve(cap));
	LMX_HIP(ctx, w.d_entity_of_slot.reserve(cap));
	LMX_HIP(ctx, w.d_dirty.reserve(cap));
	if (w.track_moved) { // two propagations per frame (staged writes, bone-attached subtrees) fit between two reads
		LMX_HIP(ctx, w.d_moved_entity.reserve(cap * 2));
		LMX_HIP(ctx, w.d_moved_tr.reserve(cap * 2));
	}
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	// scene load from locals (no world values given): every node counts as moved, so the first propagation derives all world
	// transforms; with world values (re-parenting, lmx_world_build_with_world) nothing is recomputed until something is written
	LMX_HIP(ctx, hipMemset(w.d_dirty.p, world_all ? 0 : XF_MOVED, cap));
	if (n) {
		LMX_HIP(ctx, hipMemcpy(w.d_parent_slot.p, w.parent_slot.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(w.d_slot_of_entity.p, w.slot_of_entity.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(w.d_entity_of_slot.p, w.entity_of_slot.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
		// k_xform_subtree's table: the roots cut into runs of ~XF_SUBTREE_NODES nodes (subtrees included); per run and level the first slot.
		// The subtrees of consecutive roots are contiguous in every level because a level is ordered by parent slot.
		{
			const size_t n_levels = w.level_start.size() - 1;
			const uint32_t n_roots = n_levels ? w.level_start[1] : 0;
			w.n_sub_runs = 0;
			if (n_levels >= 1 && n_levels <= XF_SUBTREE_MAX_LEVELS && n_roots) {
				// bound[l][r]: first slot of level l that belongs to root r or a later root (r = n_roots: the level's end)
				std::vector<std::vector<uint32_t>> bound(n_levels, std::vector<uint32_t>((size_t)n_roots + 1));
				for (uint32_t r = 0; r <= n_roots; ++r) bound[0][r] = r;
				for (size_t l = 1; l < n_levels; ++l) {
					uint32_t c = w.level_start[l];
					for (uint32_t r = 0; r <= n_roots; ++r) { // children of slots below bound[l - 1][r] come first
						const uint32_t pb = bound[l - 1][r];
						while (c < w.level_start[l + 1] && (uint32_t)w.parent_slot[c] < pb) ++c;
						bound[l][r] = c;
					}
				}
				std::vector<uint32_t> table;
				auto row = [&](uint32_t r) { for (size_t l = 0; l < n_levels; ++l) table.push_back(bound[l][r]); };
				uint64_t heaviest = 0;
				uint32_t run_first = 0;
				uint64_t in_run = 0;
				row(0);
				for (uint32_t r = 0; r < n_roots; ++r) {
					uint64_t size = 0;
					for (size_t l = 0; l < n_levels; ++l) size += bound[l][r + 1] - bound[l][r];
					if (in_run && in_run + size > XF_SUBTREE_NODES) { // close the run before this root
						row(r);
						heaviest = std::max(heaviest, in_run);
						run_first = r;
						in_run = 0;
					}
					in_run += size;
				}
				(void)run_first;
				heaviest = std::max(heaviest, in_run);
				row(n_roots);
				if (heaviest <= XF_SUBTREE_MAX_RUN) {
					w.n_sub_runs = (uint32_t)(table.size() / n_levels) - 1;
					LMX_HIP(ctx, w.d_sub_table.reserve(table.size()));
					LMX_HIP(ctx, hipMemcpy(w.d_sub_table.p, table.data(), table.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
				}
			}
		}
		// every entity's transform is staged through the scatter kernel (roots -> world, children -> local)
		std::vector<int32_t> all(n);
		for (uint32_t e = 0; e < n; ++e) all[e] = (int32_t)e;
		LMX_HIP(ctx, w.d_stage_entity.reserve(n));
		LMX_HIP(ctx, w.d_stage_tr.reserve(n));
		LMX_HIP(ctx, hipMemcpy(w.d_stage_entity.p, all.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
		if (world_all) {
			LMX_HIP(ctx, hipMemcpy(w.d_stage_tr.p, world_all, n * sizeof(LmxTransform), hipMemcpyHostToDevice));
			LMX_HIP(ctx, launch_xform_scatter(ctx->stream, w.dev(), w.d_slot_of_entity.p, w.d_stage_entity.p, w.d_stage_tr.p, n, XF_STAGE_RAW_WORLD));
			LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
		}
		LMX_HIP(ctx, hipMemcpy(w.d_stage_tr.p, transforms, n * sizeof(LmxTransform), hipMemcpyHostToDevice));
		LMX_HIP(ctx, launch_xform_scatter(ctx->stream, w.dev(), w.d_slot_of_entity.p, w.d_stage_entity.p, w.d_stage_tr.p, n, XF_STAGE_RAW));
	}
	w.bound_generation = ~0ull;
	if (w.n_attach) w.attach_invalidated = true; // attachment slots refer to the previous slot order: lmx_world_set_bone_attachments again
	w.n_attach = 0;
	w.built = true;
	return LMX_OK;
}

// AoS Transform[n] by entity of either the world values or the stored locals
int world_download(LmxContext* ctx, bool locals, LmxTransform* out) {
	WorldState& w = ctx->world;
	if (!w.n) return LMX_OK;
	WorldDevice dev = w.dev();
	if (locals) {
		dev.wpx = dev.lpx; dev.wpy = dev.lpy; dev.wpz = dev.lpz; dev.wrot = dev.lrot; dev.wsx = dev.lsx; dev.wsy = dev.lsy; dev.wsz = dev.lsz;
	}
	LMX_HIP(ctx, w.d_export.reserve(w.n));
	LMX_HIP(ctx, launch_xform_export(ctx->stream, dev, w.d_entity_of_slot.p, w.n, w.d_export.p));
	LMX_HIP(ctx, hipMemcpyAsync(out, w.d_export.p, (size_t)w.n * sizeof(LmxTransform), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

Xform load_xform(const LmxTransform& t) {
	Xform x;
	x.pos = DV3{t.pos[0], t.pos[1], t.pos[2]};
	x.rot = Q4{t.rot[0], t.rot[1], t.rot[2], t.rot[3]};
	x.scale = V3{t.scale[0], t.scale[1], t.scale[2]};
	return x;
}

void store_xform(const Xform& x, LmxTransform* t) {
	memset(t, 0, sizeof(*t));
	t->pos[0] = x.pos.x; t->pos[1] = x.pos.y; t->pos[2] = x.pos.z;
	t->rot[0] = x.rot.x; t->rot[1] = x.rot.y; t->rot[2] = x.rot.z; t->rot[3] = x.rot.w;
	t->scale[0] = x.scale.x; t->scale[1] = x.scale.y; t->scale[2] = x.scale.z;
}

} // namespace

extern "C" {

int lmx_world_build(LmxContext* ctx, uint32_t n, const int32_t* parent, const LmxTransform* transforms) {
	LMX_CHECK_CTX(ctx);
	if (n && (!parent || !transforms)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	WorldState& w = ctx->world;
	w.bound_entity.clear();
	w.bound_radius.clear();
	return world_rebuild(ctx, n, parent, transforms, nullptr);
}

int lmx_world_build_with_world(LmxContext* ctx, uint32_t n, const int32_t* parent, const LmxTransform* local_transforms, const LmxTransform* world_transforms) {
	LMX_CHECK_CTX(ctx);
	if (n && (!parent || !local_transforms || !world_transforms)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	WorldState& w = ctx->world;
	w.bound_entity.clear();
	w.bound_radius.clear();
	std::vector<LmxTransform> tr(local_transforms, local_transforms + n);
	for (uint32_t e = 0; e < n; ++e)
		if (parent[e] < 0) tr[e] = world_transforms[e];
	return world_rebuild(ctx, n, parent, tr.data(), world_transforms);
}

// World::setParent (world.cpp:619-701): the child keeps its world transform, its local becomes
// Transform::computeLocal(parent world, child world) (math.cpp:809-816); new_parent < 0 detaches it. An editing
// operation, not a per-frame one: the slot order is rebuilt on the host.
int lmx_world_set_parent(LmxContext* ctx, int32_t new_parent, int32_t child) {
	LMX_CHECK_CTX(ctx);
	WorldState& w = ctx->world;
	if (!w.built) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_world_build has not been called");
	if (child < 0 || (uint32_t)child >= w.n || new_parent >= (int32_t)w.n) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity out of range");
	for (int32_t a = new_parent; a >= 0; a = w.parent[a]) { // "Hierarchy can not contain a cycle." (world.cpp:621-626)
		if (a == child) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d is an ancestor of %d: hierarchy can not contain a cycle", child, new_parent);
	}
	std::vector<LmxTransform> world(w.n), tr(w.n);
	if (int rc = world_download(ctx, false, world.data())) return rc;
	if (int rc = world_download(ctx, true, tr.data())) return rc;
	std::vector<int32_t> parent = w.parent;
	parent[child] = new_parent < 0 ? -1 : new_parent;
	for (uint32_t e = 0; e < w.n; ++e)
		if (parent[e] < 0) tr[e] = world[e];
	if (new_parent >= 0) store_xform(compute_local(load_xform(world[new_parent]), load_xform(world[child])), &tr[child]);
	return world_rebuild(ctx, w.n, parent.data(), tr.data(), world.data());
}

int lmx_world_read_local_transforms(LmxContext* ctx, LmxTransform* out, uint32_t n) { // World::getLocalTransform, world.cpp:756-766
	LMX_CHECK_CTX(ctx);
	WorldState& w = ctx->world;
	if (!w.built) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_world_build has not been called");
	if (n < w.n || !out) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u transforms", w.n);
	std::vector<LmxTransform> world(w.n);
	if (int rc = world_download(ctx, true, out)) return rc;
	if (int rc = world_download(ctx, false, world.data())) return rc;
	for (uint32_t e = 0; e < w.n; ++e)
		if (w.parent[e] < 0) out[e] = world[e]; // entities without a parent: their transform (world.cpp:759)
	return LMX_OK;
}

// Transform::compose / Transform::computeLocal (math.cpp:801-816) on the host, for adapters that edit hierarchies
int lmx_transform_compose(const LmxTransform* a, const LmxTransform* b, LmxTransform* out) {
	if (!a || !b || !out) return LMX_ERR_INVALID_ARGUMENT;
	store_xform(compose(load_xform(*a), load_xform(*b)), out);
	return LMX_OK;
}

int lmx_transform_compute_local(const LmxTransform* parent, const LmxTransform* child, LmxTransform* out) {
	if (!parent || !child || !out) return LMX_ERR_INVALID_ARGUMENT;
	store_xform(compute_local(load_xform(*parent), load_xform(*child)), out);
	return LMX_OK;
}

static int world_stage(LmxContext* ctx, uint32_t n, const int32_t* entity, const LmxTransform* transforms, int mode) {
	WorldState& w = ctx->world;
	if (!w.built) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_world_build has not been called");
	if (!n) return LMX_OK;
	if (!entity || !transforms) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	for (uint32_t i = 0; i < n; ++i)
		if (entity[i] < 0 || (uint32_t)entity[i] >= w.n) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity[%u] = %d out of range", i, entity[i]);
	// The scatter kernel runs one thread per record: two records of ONE entity in a batch would race (SoA components of the two mixed).
	// The reference applies writes one by one - the last one wins - so only the last record of an entity is staged.
	std::vector<int32_t> ent_dedup;
	std::vector<LmxTransform> tr_dedup;
	{
		if (w.stage_mark.size() < w.n) w.stage_mark.assign(w.n, 0);
		const uint32_t stamp = ++w.stage_stamp;
		if (stamp == 0) { // wrapped
			std::fill(w.stage_mark.begin(), w.stage_mark.end(), 0u);
			w.stage_stamp = 1;
		}
		bool dup = false;
		for (uint32_t i = 0; i < n && !dup; ++i) {
			dup = w.stage_mark[entity[i]] == w.stage_stamp;
			w.stage_mark[entity[i]] = w.stage_stamp;
		}
		if (dup) {
			const uint32_t stamp2 = ++w.stage_stamp;
			(void)stamp2;
			ent_dedup.reserve(n);
			tr_dedup.reserve(n);
			for (uint32_t i = n; i-- > 0;) { // from the back: the first record seen of an entity is its last write
				if (w.stage_mark[entity[i]] == w.stage_stamp) continue;
				w.stage_mark[entity[i]] = w.stage_stamp;
				ent_dedup.push_back(entity[i]);
				tr_dedup.push_back(transforms[i]);
			}
			entity = ent_dedup.data();
			transforms = tr_dedup.data();
			n = (uint32_t)ent_dedup.size();
		}
	}
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream)); // staging buffers may still be read by a previous scatter
	LMX_HIP(ctx, w.d_stage_entity.reserve(n));
	LMX_HIP(ctx, w.d_stage_tr.reserve(n));
	LMX_HIP(ctx, hipMemcpy(w.d_stage_entity.p, entity, n * sizeof(int32_t), hipMemcpyHostToDevice));
	LMX_HIP(ctx, hipMemcpy(w.d_stage_tr.p, transforms, n * sizeof(LmxTransform), hipMemcpyHostToDevice));
	LMX_HIP(ctx, launch_xform_scatter(ctx->stream, w.dev(), w.d_slot_of_entity.p, w.d_stage_entity.p, w.d_stage_tr.p, n, mode));
	return LMX_OK;
}

int lmx_world_set_transforms(LmxContext* ctx, uint32_t n, const int32_t* entity, const LmxTransform* transforms) {
	LMX_CHECK_CTX(ctx);
	return world_stage(ctx, n, entity, transforms, XF_STAGE_SET_LOCAL);
}

int lmx_world_set_world_transforms(LmxContext* ctx, uint32_t n, const int32_t* entity, const LmxTransform* transforms) {
	LMX_CHECK_CTX(ctx);
	return world_stage(ctx, n, entity, transforms, XF_STAGE_SET_WORLD);
}

int lmx_world_set_transforms_device(LmxContext* ctx, uint32_t n, const void* d_entity, const void* d_transforms) {
	LMX_CHECK_CTX(ctx);
	WorldState& w = ctx->world;
	if (!w.built) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_world_build has not been called");
	if (!n) return LMX_OK;
	if (!d_entity || !d_transforms) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null device pointer");
	LMX_HIP(ctx, launch_xform_scatter(ctx->stream, w.dev(), w.d_slot_of_entity.p, (const int32_t*)d_entity, d_transforms, n, XF_STAGE_SET_LOCAL));
	return LMX_OK;
}

int lmx_world_bind_culling(LmxContext* ctx, uint32_t n, const int32_t* entity, const float* model_radius) {
	LMX_CHECK_CTX(ctx);
	WorldState& w = ctx->world;
	if (!w.built) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_world_build has not been called");
	if (n && (!entity || !model_radius)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	for (uint32_t i = 0; i < n; ++i) {
		if (entity[i] < 0 || (uint32_t)entity[i] >= w.n) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity[%u] = %d out of range", i, entity[i]);
		if (!lmx_cull_is_added(ctx, entity[i])) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d is not in the culling system", entity[i]);
	}
	// bound entities move every frame: they live in the culling system's dynamic (unsorted) set from now on; entities of the
	// previous binding that are not bound again keep their last refreshed sphere and rejoin the sorted set at its next compaction
	if (int rc = cull_dyn_sync_mirror(ctx)) return rc;
	for (int32_t e : w.bound_entity) cull_unbind(ctx, e);
	for (uint32_t i = 0; i < n; ++i) {
		if (!cull_make_dynamic(ctx, entity[i])) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d is not in the culling system", entity[i]);
	}
	w.bound_entity.assign(entity, entity + n);
	w.bound_radius.assign(model_radius, model_radius + n);
	w.bound_generation = ~0ull;
	return LMX_OK;
}

static int world_upload_binding(LmxContext* ctx) {
	WorldState& w = ctx->world;
	CullState& cs = ctx->cull;
	if (int rc = cull_flush(ctx)) return rc; // dynamic slots must be current
	if (w.bound_generation == cs.dyn_generation) return LMX_OK;
	const size_t n = w.bound_entity.size();
	std::vector<uint32_t> slot(n), dyn(n);
	for (size_t i = 0; i < n; ++i) {
		const int32_t e = w.bound_entity[i];
		if ((size_t)e >= cs.ent_to_dyn.size() || cs.ent_to_dyn[e] < 0)
			return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bound entity %d was removed from the culling system; call lmx_world_bind_culling again", e);
		slot[i] = (uint32_t)w.slot_of_entity[e];
		dyn[i] = cs.dyn[cs.ent_to_dyn[e]].slot;
	}
	LMX_HIP(ctx, w.d_bound_slot.reserve(std::max<size_t>(n, 1)));
	LMX_HIP(ctx, w.d_bound_dyn.reserve(std::max<size_t>(n, 1)));
	LMX_HIP(ctx, w.d_bound_radius.reserve(std::max<size_t>(n, 1)));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (n) {
		LMX_HIP(ctx, hipMemcpy(w.d_bound_slot.p, slot.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(w.d_bound_dyn.p, dyn.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(w.d_bound_radius.p, w.bound_radius.data(), n * sizeof(float), hipMemcpyHostToDevice));
	}
	{ // the same binding by slot (k_xform_subtree refreshes the spheres of the slots it walks)
		std::vector<uint32_t> dyn_of_slot(std::max<size_t>(w.n, 1), 0xffffffffu);
		std::vector<float> radius_of_slot(std::max<size_t>(w.n, 1), 0.f);
		for (size_t i = 0; i < n; ++i) {
			dyn_of_slot[slot[i]] = dyn[i];
			radius_of_slot[slot[i]] = w.bound_radius[i];
		}
		LMX_HIP(ctx, w.d_bound_dyn_of_slot.reserve(dyn_of_slot.size()));
		LMX_HIP(ctx, w.d_bound_radius_of_slot.reserve(radius_of_slot.size()));
		LMX_HIP(ctx, hipMemcpy(w.d_bound_dyn_of_slot.p, dyn_of_slot.data(), dyn_of_slot.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(w.d_bound_radius_of_slot.p, radius_of_slot.data(), radius_of_slot.size() * sizeof(float), hipMemcpyHostToDevice));
	}
	w.bound_generation = cs.dyn_generation;
	return LMX_OK;
}

int lmx_world_propagate(LmxContext* ctx) {
	LMX_CHECK_CTX(ctx);
	WorldState& w = ctx->world;
	if (!w.built) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_world_build has not been called");
	const WorldDevice dev = w.dev();
	const size_t n_levels = w.level_start.size() - 1;
	const bool bound = !w.bound_entity.empty();
	if (bound) {
		if (int rc = world_upload_binding(ctx)) return rc;
	}
	CullState& cs = ctx->cull;
	if (w.fused_levels && w.n_sub_runs) {
		// one launch: the levels, the marks, the moved list and the bound spheres (k_xform_subtree)
		XformSubtree a;
		memset(&a, 0, sizeof(a));
		a.table = w.d_sub_table.p;
		a.n_levels = (uint32_t)n_levels;
		a.entity_of_slot = w.d_entity_of_slot.p;
		a.cap = w.n * 2u;
		if (w.track_moved) { a.out_entity = w.d_moved_entity.p; a.out_tr = w.d_moved_tr.p; a.count = w.d_moved_count.p; }
		if (bound) {
			a.bound_dyn_of_slot = w.d_bound_dyn_of_slot.p; a.bound_radius_of_slot = w.d_bound_radius_of_slot.p;
			a.dyn_px = cs.dyn_px.p; a.dyn_py = cs.dyn_py.p; a.dyn_pz = cs.dyn_pz.p; a.dyn_radius = cs.dyn_radius.p;
		}
		ProfScope ps(ctx, LMX_K_XFORM_LEVEL);
		LMX_HIP(ctx, launch_xform_subtree(ctx->stream, dev, a, w.n_sub_runs));
	} else {
		for (size_t l = 1; l + 1 < w.level_start.size(); ++l) {
			ProfScope ps(ctx, LMX_K_XFORM_LEVEL);
			LMX_HIP(ctx, launch_xform_level(ctx->stream, dev, w.level_start[l], w.level_start[l + 1] - w.level_start[l]));
		}
		if (w.n && w.track_moved) { // the frame's "moved" marks end here: collected for the hand-back, then cleared
			LMX_HIP(ctx, launch_xform_collect_moved(ctx->stream, dev, w.d_entity_of_slot.p, w.n, w.n * 2u, w.d_moved_entity.p, w.d_moved_tr.p, w.d_moved_count.p));
		} else if (w.n) {
			LMX_HIP(ctx, hipMemsetAsync(w.d_dirty.p, 0, w.n, ctx->stream));
		}
		if (bound) {
			ProfScope ps(ctx, LMX_K_SPHERE_REFRESH);
			LMX_HIP(ctx, launch_sphere_refresh(ctx->stream, dev, w.d_bound_slot.p, w.d_bound_dyn.p, w.d_bound_radius.p, cs.dyn_px.p, cs.dyn_py.p,
				cs.dyn_pz.p, cs.dyn_radius.p, (uint32_t)w.bound_entity.size()));
		}
	}
	if (bound) cs.dyn_mirror_stale = true; // the device copy of the bound entities is now newer than the host mirror
	return LMX_OK;
}

int lmx_world_set_bone_attachments(LmxContext* ctx, uint32_t n, const int32_t* entity, const int32_t* parent_entity, const uint32_t* skin_instance,
	const uint32_t* bone_index, const LmxLocalRigidTransform* relative) {
	LMX_CHECK_CTX(ctx);
	WorldState& w = ctx->world;
	const SkinState& sk = ctx->skin;
	if (!w.built) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_world_build has not been called");
	if (n && (!entity || !parent_entity || !skin_instance || !bone_index || !relative)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null input array");
	std::vector<BoneAttachDevice> att(n);
	std::vector<uint8_t> attached(w.n, 0);
	for (uint32_t i = 0; i < n; ++i) {
		if (entity[i] < 0 || (uint32_t)entity[i] >= w.n || parent_entity[i] < 0 || (uint32_t)parent_entity[i] >= w.n)
			return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "attachment %u: entity %d / parent %d out of range", i, entity[i], parent_entity[i]);
		if (w.parent[entity[i]] >= 0) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "attachment %u: entity %d has a hierarchy parent; attached entities are moved with World::setTransform as roots", i, entity[i]);
		if (attached[entity[i]]) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "entity %d is attached twice", entity[i]);
		attached[entity[i]] = 1;
		if (skin_instance[i] >= sk.inst.size()) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "attachment %u: unknown skin instance %u", i, skin_instance[i]);
		if (bone_index[i] >= sk.inst[skin_instance[i]].n_bones) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "attachment %u: bone %u >= %u", i, bone_index[i], sk.inst[skin_instance[i]].n_bones);
		BoneAttachDevice& a = att[i];
		a.slot = (uint32_t)w.slot_of_entity[entity[i]];
		a.parent_slot = (uint32_t)w.slot_of_entity[parent_entity[i]];
		a.skin_instance = skin_instance[i];
		a.bone = bone_index[i];
		memcpy(a.rel_pos, relative[i].pos, sizeof(a.rel_pos));
		memcpy(a.rel_rot, relative[i].rot, sizeof(a.rel_rot));
	}
	for (uint32_t i = 0; i < n; ++i) // one parallel pass: an attachment may not hang off another attachment's subtree root
		if (attached[parent_entity[i]]) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "attachment %u: parent %d is itself a bone attachment (chains are not batched)", i, parent_entity[i]);
	LMX_HIP(ctx, w.d_attach.reserve(std::max<size_t>(n, 1)));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (n) LMX_HIP(ctx, hipMemcpy(w.d_attach.p, att.data(), (size_t)n * sizeof(BoneAttachDevice), hipMemcpyHostToDevice));
	w.n_attach = n;
	w.attach_invalidated = false;
	w.attach_skin_instances = sk.inst.size();
	return LMX_OK;
}

int lmx_world_update_bone_attachments(LmxContext* ctx) {
	LMX_CHECK_CTX(ctx);
	WorldState& w = ctx->world;
	SkinState& sk = ctx->skin;
	if (!w.built) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_world_build has not been called");
	if (w.attach_invalidated) return fail(ctx, LMX_ERR_NOT_BUILT, "the hierarchy was rebuilt (lmx_world_build / lmx_world_set_parent) after the attachments were set; call lmx_world_set_bone_attachments again");
	if (!w.n_attach) return LMX_OK;
	if (w.attach_skin_instances != sk.inst.size()) return fail(ctx, LMX_ERR_NOT_BUILT, "the skin instance table changed; call lmx_world_set_bone_attachments again");
	if (!sk.pose_is_absolute) return fail(ctx, LMX_ERR_NOT_BUILT, "bone attachments read the absolute pose (ASSERT(pose->is_absolute), render_module.cpp:424): run lmx_skin_run with pose write-back first");
	LMX_HIP(ctx, launch_bone_attach(ctx->stream, w.dev(), w.d_attach.p, w.n_attach, sk.d_inst.p, sk.d_pose_pos.p, sk.d_pose_rot.p));
	return LMX_OK;
}

int lmx_world_set_option(LmxContext* ctx, int option, int value) {
	LMX_CHECK_CTX(ctx);
	if (option != LMX_WORLD_OPT_FUSED_LEVELS) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "unknown world option %d", option);
	ctx->world.fused_levels = value != 0;
	return LMX_OK;
}

int lmx_world_track_moved(LmxContext* ctx, int enable) {
	LMX_CHECK_CTX(ctx);
	WorldState& w = ctx->world;
	w.track_moved = enable != 0;
	if (w.track_moved) {
		LMX_HIP(ctx, w.d_moved_count.reserve(1));
		LMX_HIP(ctx, hipMemsetAsync(w.d_moved_count.p, 0, sizeof(uint32_t), ctx->stream));
		if (w.built) {
			LMX_HIP(ctx, w.d_moved_entity.reserve(std::max<size_t>((size_t)w.n * 2, 1)));
			LMX_HIP(ctx, w.d_moved_tr.reserve(std::max<size_t>((size_t)w.n * 2, 1)));
		}
	}
	return LMX_OK;
}

int lmx_world_read_moved(LmxContext* ctx, int32_t* entity, LmxTransform* transforms, uint32_t cap, uint32_t* out_n) {
	LMX_CHECK_CTX(ctx);
	WorldState& w = ctx->world;
	if (!w.built) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_world_build has not been called");
	if (!w.track_moved) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_world_track_moved(1) first");
	if (!out_n) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null out_n");
	// the count and - sized from the last frame's count - the records travel together: one host wait per frame in the steady state
	// (the copies below the wait only run when this frame moved more than the guess covers)
	uint32_t n = 0;
	const uint32_t list_cap = w.n * 2u;
	const uint32_t ahead = (entity && transforms) ? std::min(std::min(w.moved_guess, cap), list_cap) : 0u;
	LMX_HIP(ctx, hipMemcpyAsync(&n, w.d_moved_count.p, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
	if (ahead) {
		LMX_HIP(ctx, hipMemcpyAsync(entity, w.d_moved_entity.p, (size_t)ahead * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
		LMX_HIP(ctx, hipMemcpyAsync(transforms, w.d_moved_tr.p, (size_t)ahead * sizeof(LmxTransform), hipMemcpyDeviceToHost, ctx->stream));
	}
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	*out_n = n;
	// (an entity can be listed twice when a frame propagated twice - staged writes, then bone-attached subtrees; later entries are newer)
	const uint32_t stored = std::min(n, w.n * 2u); // what the lists can hold: see the reservation in world_rebuild
	if (n > stored) { // more than two propagations without a read: the overflow was counted, not stored
		LMX_HIP(ctx, hipMemsetAsync(w.d_moved_count.p, 0, sizeof(uint32_t), ctx->stream));
		return fail(ctx, LMX_ERR_CAPACITY, "%u moved records since the last read exceed the list (%u): read every transform with lmx_world_read_transforms", n, stored);
	}
	if (n > cap || (n && (!entity || !transforms))) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u moved entities", n);
	if (n > ahead) { // the rest of a frame that moved more than the guess
		LMX_HIP(ctx, hipMemcpyAsync(entity + ahead, w.d_moved_entity.p + ahead, (size_t)(n - ahead) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
		LMX_HIP(ctx, hipMemcpyAsync(transforms + ahead, w.d_moved_tr.p + ahead, (size_t)(n - ahead) * sizeof(LmxTransform), hipMemcpyDeviceToHost, ctx->stream));
	}
	LMX_HIP(ctx, hipMemsetAsync(w.d_moved_count.p, 0, sizeof(uint32_t), ctx->stream)); // (stream-ordered behind the copies; the next propagation is behind it)
	if (n > ahead) LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	w.moved_guess = n + n / 4 + 64;
	return LMX_OK;
}

int lmx_world_read_transforms(LmxContext* ctx, LmxTransform* out, uint32_t n) {
	LMX_CHECK_CTX(ctx);
	WorldState& w = ctx->world;
	if (!w.built) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_world_build has not been called");
	if (n < w.n || !out) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u transforms", w.n);
	if (!w.n) return LMX_OK;
	LMX_HIP(ctx, w.d_export.reserve(w.n));
	LMX_HIP(ctx, launch_xform_export(ctx->stream, w.dev(), w.d_entity_of_slot.p, w.n, w.d_export.p));
	LMX_HIP(ctx, hipMemcpyAsync(out, w.d_export.p, (size_t)w.n * sizeof(LmxTransform), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}


} // extern "C"
