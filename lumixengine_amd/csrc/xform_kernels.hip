// xform_kernels.hip — hierarchical world-transform update of World on gfx950.
//
// The reference propagates eagerly and serially: World::transformEntity (src/engine/world.cpp:255-282) walks the
// subtree of every written entity, child.world = my.compose(child.local) (src/core/math.cpp:801-807), and fires
// the per-component `transformed` delegates, of which RenderModuleImpl::onModelInstanceMoved
// (src/renderer/render_module.cpp:1544-1554) refreshes the culling sphere.
//
// Here nodes live in (level, parent-slot) order in SoA arrays; one launch per level computes every node of that
// level from its parent's already-final world transform (fp64 position, fp32 rotation/scale, no FMA), so sibling
// lanes read the same or adjacent parent slots. A fused pass refreshes the culling spheres of bound entities.
#include "lmx_kernels.h"

namespace lmx {

namespace {

// dirty[s] & 3: what the host staged for node s since the last propagation; dirty[s] & XF_MOVED: the node's world transform
// changed in this propagation (roots: staged; others: recomputed by this kernel). A node is recomputed only when its parent moved
// or it was staged itself - exactly the nodes the reference's DFS visits. (Recomputing an untouched node from its stored local is
// NOT a no-op: locals re-derived by computeLocal do not reproduce the stored world transform bit for bit.)
//   XF_CLEAN      world = parent.compose(local)                                  (World::transformEntity's descent, world.cpp:271-280)
//   XF_SET_LOCAL  World::setLocalTransform (world.cpp:741-753 -> updateGlobalTransform :704-712 -> setTransform :337-342 ->
//                 transformEntity(update_local = true) :266-269): world = parent.compose(local), then the stored local is RE-DERIVED
//                 as Transform::computeLocal(parent, world) (math.cpp:809-816) - lossy, and what later frames compose with
//   XF_SET_WORLD  World::setTransform on an entity with a parent: the world transform is the staged one, local = computeLocal(parent, world)
__global__ __launch_bounds__(256) void k_xform_level(WorldDevice w, uint32_t first, uint32_t n) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const uint32_t s = first + i;
	const int32_t p = w.parent_slot[s];
	const uint8_t dirty = w.dirty[s] & 3u;
	if (dirty == XF_CLEAN && !(w.dirty[p] & XF_MOVED)) return;
	Xform parent, local;
	const float4 pr = w.wrot[p];
	parent.pos = DV3{w.wpx[p], w.wpy[p], w.wpz[p]};
	parent.rot = Q4{pr.x, pr.y, pr.z, pr.w};
	parent.scale = V3{w.wsx[p], w.wsy[p], w.wsz[p]};
	Xform r;
	if (dirty != XF_SET_WORLD) {
		const float4 lr = w.lrot[s];
		local.pos = DV3{w.lpx[s], w.lpy[s], w.lpz[s]};
		local.rot = Q4{lr.x, lr.y, lr.z, lr.w};
		local.scale = V3{w.lsx[s], w.lsy[s], w.lsz[s]};
		r = compose(parent, local);
		w.wpx[s] = r.pos.x;
		w.wpy[s] = r.pos.y;
		w.wpz[s] = r.pos.z;
		w.wrot[s] = make_float4(r.rot.x, r.rot.y, r.rot.z, r.rot.w);
		w.wsx[s] = r.scale.x;
		w.wsy[s] = r.scale.y;
		w.wsz[s] = r.scale.z;
	} else {
		const float4 wr = w.wrot[s];
		r.pos = DV3{w.wpx[s], w.wpy[s], w.wpz[s]};
		r.rot = Q4{wr.x, wr.y, wr.z, wr.w};
		r.scale = V3{w.wsx[s], w.wsy[s], w.wsz[s]};
	}
	if (dirty != XF_CLEAN) {
		const Xform l = compute_local(parent, r);
		w.lpx[s] = l.pos.x;
		w.lpy[s] = l.pos.y;
		w.lpz[s] = l.pos.z;
		w.lrot[s] = make_float4(l.rot.x, l.rot.y, l.rot.z, l.rot.w);
		w.lsx[s] = l.scale.x;
		w.lsy[s] = l.scale.y;
		w.lsz[s] = l.scale.z;
	}
	w.dirty[s] = XF_MOVED;
}

// ---- shallow hierarchies: every level in ONE launch ------------------------------------------------------------------------------
// One launch per level is bound by launch latency when a level holds a few hundred thousand nodes (3 launches x 9-11 us for 250 k nodes
// each: 0.43 of HBM by algorithmic bytes). Here every non-root node walks UP its ancestor chain (<= XF_FUSED_MAX_DEPTH links), finds the
// topmost ancestor that was written this frame, and re-composes DOWN from there in registers: child.world = parent.compose(child.local)
// with the operations of k_xform_level in the same order, so the result is bit-identical - a node just does not wait for its parent's
// thread. What makes that legal: during this kernel nothing a thread READS is written by another one - marks stay as staged (the
// "moved" output goes to a second byte array), stored locals stay as staged (the re-derivation of a written child's local,
// Transform::computeLocal, moves to k_xform_finalize, when every world value is final), and a world value is only read from a node
// that is not recomputed (the untouched ancestor above the topmost written one, or a node whose world transform was staged).
// The ancestors' work is repeated by their descendants (depth-4 chains: 6 composes AND 9 instead of 6 record loads for 3 nodes). MEASURED
// (profiles/r03/xform_fused.txt; 250 k roots x depth-4 chains, every root moved): the three level launches take 33-35 us together, this
// kernel 27.7 us with a parent_slot walk and 32 us with the ancestor table - the level kernels are bound by their ~20 separate 4-8-byte
// SoA streams per node, not by launch latency, and the repeated loads eat what the saved launches give (whole step 32.7 vs 34.3 us).
// Kept as an option (LMX_WORLD_OPT_FUSED_LEVELS, default off) for small worlds, where the launches dominate; bit-identical either way.
constexpr int XF_FUSED_MAX_DEPTH = 7; // ancestors above a node: hierarchies of up to 8 levels

__device__ __forceinline__ Xform load_world(const WorldDevice& w, int32_t s) {
	Xform x;
	const float4 r = w.wrot[s];
	x.pos = DV3{w.wpx[s], w.wpy[s], w.wpz[s]};
	x.rot = Q4{r.x, r.y, r.z, r.w};
	x.scale = V3{w.wsx[s], w.wsy[s], w.wsz[s]};
	return x;
}
__device__ __forceinline__ Xform load_local(const WorldDevice& w, int32_t s) {
	Xform x;
	const float4 r = w.lrot[s];
	x.pos = DV3{w.lpx[s], w.lpy[s], w.lpz[s]};
	x.rot = Q4{r.x, r.y, r.z, r.w};
	x.scale = V3{w.lsx[s], w.lsy[s], w.lsz[s]};
	return x;
}

__global__ __launch_bounds__(256) void k_xform_fused(WorldDevice w, uint8_t* __restrict__ moved_out, const int32_t* __restrict__ ancestors, uint32_t n_slots,
	uint32_t n_anc, uint32_t first, uint32_t n) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const int32_t s = (int32_t)(first + i);
	// the chain: c[0] = the node, c[k] = its k-th ancestor, -1 beyond the root (static indices: the arrays stay in registers). The
	// ancestors come from a table built with the hierarchy (ancestors[(k - 1) * n_slots + s], k <= n_anc = levels - 1): independent
	// coalesced loads instead of a chain of parent_slot[parent_slot[...]] round trips
	int32_t c[XF_FUSED_MAX_DEPTH + 2];
	uint32_t m[XF_FUSED_MAX_DEPTH + 1];
	c[0] = s;
#pragma unroll
	for (int k = 1; k <= XF_FUSED_MAX_DEPTH + 1; ++k) c[k] = (uint32_t)k <= n_anc ? ancestors[(size_t)(k - 1) * n_slots + s] : -1;
#pragma unroll
	for (int k = 0; k <= XF_FUSED_MAX_DEPTH; ++k) m[k] = c[k] >= 0 ? (uint32_t)w.dirty[c[k]] : 0u;
	// topmost written node of the chain: a root counts when its world transform was staged (XF_MOVED), a child when a local or a
	// world transform was staged for it
	int top = -1;
#pragma unroll
	for (int k = 0; k <= XF_FUSED_MAX_DEPTH; ++k) {
		if (c[k] < 0) continue;
		const bool is_root = c[k + 1] < 0;
		if (is_root ? (m[k] & XF_MOVED) != 0 : (m[k] & 3u) != 0) top = k;
	}
	if (top < 0) return; // nothing above (or at) this node was written: exactly the nodes the reference's DFS does not visit
	Xform cur = {};
#pragma unroll
	for (int k = XF_FUSED_MAX_DEPTH; k >= 0; --k) {
		if (k > top || c[k] < 0) continue;
		const bool is_root = c[k + 1] < 0;
		const uint32_t mark = m[k] & 3u;
		if (is_root || mark == XF_SET_WORLD) cur = load_world(w, c[k]);          // staged world transform: taken as it is
		else if (k == top) cur = compose(load_world(w, c[k + 1]), load_local(w, c[k])); // XF_SET_LOCAL under an untouched parent
		else cur = compose(cur, load_local(w, c[k]));                             // World::transformEntity's descent (world.cpp:271-280)
	}
	if ((m[0] & 3u) != XF_SET_WORLD) {
		w.wpx[s] = cur.pos.x;
		w.wpy[s] = cur.pos.y;
		w.wpz[s] = cur.pos.z;
		w.wrot[s] = make_float4(cur.rot.x, cur.rot.y, cur.rot.z, cur.rot.w);
		w.wsx[s] = cur.scale.x;
		w.wsy[s] = cur.scale.y;
		w.wsz[s] = cur.scale.z;
	}
	moved_out[s] = XF_MOVED;
}

struct TransformAoS { double pos[3]; float rot[4]; float scale[3]; float pad; }; // core/math.h:306-327, 56 B

__global__ __launch_bounds__(256) void k_xform_export(WorldDevice w, const int32_t* __restrict__ entity_of_slot, uint32_t n,
	TransformAoS* __restrict__ out) {
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s >= n) return;
	const float4 r = w.wrot[s];
	TransformAoS t;
	t.pos[0] = w.wpx[s]; t.pos[1] = w.wpy[s]; t.pos[2] = w.wpz[s];
	t.rot[0] = r.x; t.rot[1] = r.y; t.rot[2] = r.z; t.rot[3] = r.w;
	t.scale[0] = w.wsx[s]; t.scale[1] = w.wsy[s]; t.scale[2] = w.wsz[s];
	t.pad = 0.f;
	out[entity_of_slot[s]] = t;
}

// The hand-back of a frame: what World::transformEntity would have visited. One pass over the marks: moved slots append
// {entity, world transform} to the lists (wave64 ballot + mbcnt ranks, ONE returning atomic per wave), every mark is cleared.
// `count` keeps running across launches (a frame may propagate twice: staged writes, then bone-attached subtrees); entries
// beyond `cap` are counted but not stored.
__global__ __launch_bounds__(256) void k_xform_collect_moved(WorldDevice w, const int32_t* __restrict__ entity_of_slot, uint32_t n, uint32_t cap,
	int32_t* __restrict__ out_entity, TransformAoS* __restrict__ out_tr, uint32_t* __restrict__ count) {
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	const bool moved = s < n && (w.dirty[s] & XF_MOVED) != 0;
	const uint64_t mask = __ballot(moved);
	if (s < n && w.dirty[s] != 0) w.dirty[s] = 0;
	if (mask == 0) return;
	const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
	uint32_t base = 0;
	if (lane == 0) base = atomicAdd(count, (uint32_t)__popcll(mask));
	base = __builtin_amdgcn_readfirstlane(base);
	if (!moved) return;
	const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
	if (at >= cap) return;
	const float4 r = w.wrot[s];
	TransformAoS t;
	t.pos[0] = w.wpx[s]; t.pos[1] = w.wpy[s]; t.pos[2] = w.wpz[s];
	t.rot[0] = r.x; t.rot[1] = r.y; t.rot[2] = r.z; t.rot[3] = r.w;
	t.scale[0] = w.wsx[s]; t.scale[1] = w.wsy[s]; t.scale[2] = w.wsz[s];
	t.pad = 0.f;
	out_entity[at] = entity_of_slot[s];
	out_tr[at] = t;
}

// After k_xform_fused, when every world value is final: written children get their stored local RE-DERIVED (World::transformEntity
// with update_local, world.cpp:266-269: Transform::computeLocal(parent, world) - lossy, and what later frames compose with), the
// moved nodes (staged roots + everything k_xform_fused recomputed) are appended to the hand-back lists when tracking is on, and both
// mark arrays are cleared.
__global__ __launch_bounds__(256) void k_xform_finalize(WorldDevice w, uint8_t* __restrict__ moved_out, const int32_t* __restrict__ entity_of_slot, uint32_t n, uint32_t cap,
	int32_t* __restrict__ out_entity, TransformAoS* __restrict__ out_tr, uint32_t* __restrict__ count) {
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	uint32_t mark = 0, mv = 0;
	if (s < n) {
		mark = w.dirty[s];
		mv = moved_out[s];
		if (mark != 0) w.dirty[s] = 0;
		if (mv != 0) moved_out[s] = 0;
		const int32_t p = w.parent_slot[s];
		if ((mark & 3u) != 0 && p >= 0) {
			const Xform l = compute_local(load_world(w, p), load_world(w, (int32_t)s));
			w.lpx[s] = l.pos.x;
			w.lpy[s] = l.pos.y;
			w.lpz[s] = l.pos.z;
			w.lrot[s] = make_float4(l.rot.x, l.rot.y, l.rot.z, l.rot.w);
			w.lsx[s] = l.scale.x;
			w.lsy[s] = l.scale.y;
			w.lsz[s] = l.scale.z;
		}
	}
	if (count == nullptr) return; // no hand-back list wanted
	const bool moved = ((mark & XF_MOVED) | mv) != 0;
	const uint64_t mask = __ballot(moved);
	if (mask == 0) return;
	const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
	uint32_t base = 0;
	if (lane == 0) base = atomicAdd(count, (uint32_t)__popcll(mask));
	base = __builtin_amdgcn_readfirstlane(base);
	if (!moved) return;
	const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
	if (at >= cap) return;
	const float4 r = w.wrot[s];
	TransformAoS t;
	t.pos[0] = w.wpx[s]; t.pos[1] = w.wpy[s]; t.pos[2] = w.wpz[s];
	t.rot[0] = r.x; t.rot[1] = r.y; t.rot[2] = r.z; t.rot[3] = r.w;
	t.scale[0] = w.wsx[s]; t.scale[1] = w.wsy[s]; t.scale[2] = w.wsz[s];
	t.pad = 0.f;
	out_entity[at] = entity_of_slot[s];
	out_tr[at] = t;
}

// Stage new transforms. mode XF_STAGE_SET_LOCAL: roots get their world transform (World::setTransform, world.cpp:337-342), children
// their local transform + the XF_SET_LOCAL mark (World::setLocalTransform, world.cpp:741-753); XF_STAGE_SET_WORLD: World::setTransform
// on any entity (children keep the staged world transform, XF_SET_WORLD mark); XF_STAGE_RAW: roots -> world, children -> stored local
// as is (scene load: Hierarchy::local_transform comes from the file); XF_STAGE_RAW_WORLD: every value to the world arrays.
__global__ __launch_bounds__(256) void k_xform_scatter(WorldDevice w, const int32_t* __restrict__ slot_of_entity,
	const int32_t* __restrict__ entity, const TransformAoS* __restrict__ tr, uint32_t n, int mode) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const int32_t s = slot_of_entity[entity[i]];
	const TransformAoS t = tr[i];
	const bool is_root = w.parent_slot[s] < 0;
	if (mode == XF_STAGE_RAW_WORLD || mode == XF_STAGE_SET_WORLD || is_root) {
		w.wpx[s] = t.pos[0]; w.wpy[s] = t.pos[1]; w.wpz[s] = t.pos[2];
		w.wrot[s] = make_float4(t.rot[0], t.rot[1], t.rot[2], t.rot[3]);
		w.wsx[s] = t.scale[0]; w.wsy[s] = t.scale[1]; w.wsz[s] = t.scale[2];
		if (mode == XF_STAGE_SET_WORLD && !is_root) w.dirty[s] = XF_SET_WORLD;
		else if (mode != XF_STAGE_RAW_WORLD && mode != XF_STAGE_RAW) w.dirty[s] = XF_MOVED; // a root moved: its subtree follows
	} else {
		w.lpx[s] = t.pos[0]; w.lpy[s] = t.pos[1]; w.lpz[s] = t.pos[2];
		w.lrot[s] = make_float4(t.rot[0], t.rot[1], t.rot[2], t.rot[3]);
		w.lsx[s] = t.scale[0]; w.lsy[s] = t.scale[1]; w.lsz[s] = t.scale[2];
		if (mode == XF_STAGE_SET_LOCAL) w.dirty[s] = XF_SET_LOCAL;
	}
}

// onModelInstanceMoved (render_module.cpp:1544-1554): CullingSystem::set(entity, tr.pos, bounding_radius * maximum(scale)).
// Bound entities live in the culling system's dynamic set, which stores exactly these two values; the cell assignment
// of CullingSystem::set is re-derived from them by the cull kernel, so nothing is re-binned here.
__global__ __launch_bounds__(256) void k_sphere_refresh(WorldDevice w, const uint32_t* __restrict__ bound_slot,
	const uint32_t* __restrict__ bound_dyn, const float* __restrict__ model_radius, double* __restrict__ dyn_px,
	double* __restrict__ dyn_py, double* __restrict__ dyn_pz, float* __restrict__ dyn_radius, uint32_t n) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const uint32_t s = bound_slot[i];
	const uint32_t d = bound_dyn[i];
	dyn_px[d] = w.wpx[s];
	dyn_py[d] = w.wpy[s];
	dyn_pz[d] = w.wpz[s];
	// model_radius < 0 marks a position-only binding: onDecalMoved / onPointLightMoved call CullingSystem::setPosition and
	// keep the radius (render_module.cpp:1568-1592)
	const float mr = model_radius[i];
	if (!(mr < 0.f)) dyn_radius[d] = mr * maximum3(w.wsx[s], w.wsy[s], w.wsz[s]); // (a NaN model radius is not a marker)
}

} // namespace

hipError_t launch_xform_level(hipStream_t s, const WorldDevice& w, uint32_t first, uint32_t n) {
	if (!n) return hipSuccess;
	hipLaunchKernelGGL(k_xform_level, dim3((n + 255u) / 256u), dim3(256), 0, s, w, first, n);
	return hipGetLastError();
}

hipError_t launch_xform_collect_moved(hipStream_t s, const WorldDevice& w, const int32_t* entity_of_slot, uint32_t n, uint32_t cap, int32_t* out_entity,
	void* out_transforms, uint32_t* count) {
	if (!n) return hipSuccess;
	hipLaunchKernelGGL(k_xform_collect_moved, dim3((n + 255u) / 256u), dim3(256), 0, s, w, entity_of_slot, n, cap, out_entity, (TransformAoS*)out_transforms, count);
	return hipGetLastError();
}

hipError_t launch_xform_fused(hipStream_t s, const WorldDevice& w, uint8_t* moved_out, const int32_t* ancestors, uint32_t n_slots, uint32_t n_anc, uint32_t first_nonroot,
	uint32_t n_nonroot) {
	if (!n_nonroot) return hipSuccess;
	hipLaunchKernelGGL(k_xform_fused, dim3((n_nonroot + 255u) / 256u), dim3(256), 0, s, w, moved_out, ancestors, n_slots, n_anc, first_nonroot, n_nonroot);
	return hipGetLastError();
}

hipError_t launch_xform_finalize(hipStream_t s, const WorldDevice& w, uint8_t* moved_out, const int32_t* entity_of_slot, uint32_t n, uint32_t cap, int32_t* out_entity,
	void* out_transforms, uint32_t* count) {
	if (!n) return hipSuccess;
	hipLaunchKernelGGL(k_xform_finalize, dim3((n + 255u) / 256u), dim3(256), 0, s, w, moved_out, entity_of_slot, n, cap, out_entity, (TransformAoS*)out_transforms, count);
	return hipGetLastError();
}

hipError_t launch_xform_export(hipStream_t s, const WorldDevice& w, const int32_t* entity_of_slot, uint32_t n, void* out_transforms) {
	if (!n) return hipSuccess;
	hipLaunchKernelGGL(k_xform_export, dim3((n + 255u) / 256u), dim3(256), 0, s, w, entity_of_slot, n, (TransformAoS*)out_transforms);
	return hipGetLastError();
}

// RenderModuleImpl::updateBoneAttachment (render_module.cpp:377-404) for every attachment: world transform of the attached
// entity = parent entity's world transform . (absolute bone pose . relative transform), own scale kept. Attached entities are
// hierarchy roots (World::setTransform writes their world transform; their subtrees follow in lmx_world_propagate).
__global__ __launch_bounds__(256) void k_bone_attach(WorldDevice w, const BoneAttachDevice* __restrict__ att, uint32_t n,
	const SkinInstance* __restrict__ inst, const float* __restrict__ pose_pos, const float4* __restrict__ pose_rot) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const BoneAttachDevice a = att[i];
	const uint32_t ps = a.parent_slot, s = a.slot;
	Xform parent;
	parent.pos = DV3{w.wpx[ps], w.wpy[ps], w.wpz[ps]};
	const float4 pr = w.wrot[ps];
	parent.rot = Q4{pr.x, pr.y, pr.z, pr.w};
	parent.scale = V3{w.wsx[ps], w.wsy[ps], w.wsz[ps]};
	const size_t b = (size_t)inst[a.skin_instance].bone_offset + a.bone;
	const float4 br = pose_rot[b];
	const Xform r = bone_attachment(parent, V3{pose_pos[3 * b], pose_pos[3 * b + 1], pose_pos[3 * b + 2]}, Q4{br.x, br.y, br.z, br.w},
		V3{a.rel_pos[0], a.rel_pos[1], a.rel_pos[2]}, Q4{a.rel_rot[0], a.rel_rot[1], a.rel_rot[2], a.rel_rot[3]}, V3{w.wsx[s], w.wsy[s], w.wsz[s]});
	w.wpx[s] = r.pos.x; w.wpy[s] = r.pos.y; w.wpz[s] = r.pos.z;
	w.wrot[s] = make_float4(r.rot.x, r.rot.y, r.rot.z, r.rot.w);
	w.dirty[s] = XF_MOVED; // World::setTransform on the attached entity: its subtree follows in lmx_world_propagate
}

hipError_t launch_bone_attach(hipStream_t s, const WorldDevice& w, const BoneAttachDevice* att, uint32_t n, const SkinInstance* inst,
	const float* pose_pos, const float4* pose_rot) {
	if (!n) return hipSuccess;
	hipLaunchKernelGGL(k_bone_attach, dim3((n + 255u) / 256u), dim3(256), 0, s, w, att, n, inst, pose_pos, pose_rot);
	return hipGetLastError();
}

hipError_t launch_xform_scatter(hipStream_t s, const WorldDevice& w, const int32_t* slot_of_entity, const int32_t* entity,
	const void* transforms, uint32_t n, int mode) {
	if (!n) return hipSuccess;
	hipLaunchKernelGGL(k_xform_scatter, dim3((n + 255u) / 256u), dim3(256), 0, s, w, slot_of_entity, entity,
		(const TransformAoS*)transforms, n, mode);
	return hipGetLastError();
}

hipError_t launch_sphere_refresh(hipStream_t s, const WorldDevice& w, const uint32_t* bound_slot, const uint32_t* bound_dyn,
	const float* model_radius, double* dyn_px, double* dyn_py, double* dyn_pz, float* dyn_radius, uint32_t n) {
	if (!n) return hipSuccess;
	hipLaunchKernelGGL(k_sphere_refresh, dim3((n + 255u) / 256u), dim3(256), 0, s, w, bound_slot, bound_dyn, model_radius, dyn_px, dyn_py,
		dyn_pz, dyn_radius, n);
	return hipGetLastError();
}

} // namespace lmx
