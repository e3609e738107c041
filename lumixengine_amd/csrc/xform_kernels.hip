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
__device__ __forceinline__ void xform_node(const WorldDevice& w, uint32_t s) {
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

__global__ __launch_bounds__(256) void k_xform_level(WorldDevice w, uint32_t first, uint32_t n) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) xform_node(w, first + i);
}

struct TransformAoS { double pos[3]; float rot[4]; float scale[3]; float pad; }; // core/math.h:306-327, 56 B

// ---- every level in ONE launch: a block owns the subtrees of a run of roots -------------------------------------------------------------
// Slots are in (level, parent slot) order, so the subtrees of CONSECUTIVE roots occupy one contiguous range of slots in every level: the
// children of a contiguous range of parents are a contiguous range (induction over the levels). The host cuts the roots into runs whose
// subtrees hold ~XF_SUBTREE_NODES nodes and records every run's first slot per level (`table[run * n_levels + level]`, one more row at the
// end); a block walks its run level by level - the very statements of k_xform_level per node - with a workgroup barrier in between: the
// parent a node composes with was written by this block one step earlier and comes out of the CU's own L1 / L2 instead of HBM, and the
// 2-7 dependent launches of a frame (one per level, each with its ramp and tail at a few hundred thousand nodes) become one. Behind
// the last level the same block clears its marks, appends its moved nodes to the hand-back lists (k_xform_collect_moved's job) and
// refreshes the culling spheres of its bound entities (k_sphere_refresh's): nothing of a frame's propagation is a launch of its own.
// Bit-identical to the per-level launches: a node still sees exactly its parent's final value. Hierarchies the table does not fit
// (more than XF_SUBTREE_MAX_LEVELS levels, or one root with more than XF_SUBTREE_MAX_RUN nodes under it - a block would walk them
// alone) keep the per-level launches.
// What a node's update reads that does NOT depend on its parent's new value: fetched one level ahead (prefetch), so that a level costs
// one dependent round trip - its parents' values - instead of three (parent slot -> marks -> transforms).
struct XformPre { int32_t p; uint8_t mark; Xform local; };
__device__ __forceinline__ XformPre xform_prefetch(const WorldDevice& w, uint32_t s) {
	XformPre x;
	x.p = w.parent_slot[s];
	x.mark = w.dirty[s] & 3u;
	const float4 lr = w.lrot[s];
	x.local.pos = DV3{w.lpx[s], w.lpy[s], w.lpz[s]};
	x.local.rot = Q4{lr.x, lr.y, lr.z, lr.w};
	x.local.scale = V3{w.lsx[s], w.lsy[s], w.lsz[s]};
	return x;
}
// LDS only has to be visible to the block's own waves: no need to drain the wave's global stores (vmcnt) as __syncthreads() does
__device__ __forceinline__ void lds_barrier() {
#ifdef LMX_HOSTSIM
	__syncthreads();
#else
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

#ifndef LMX_XFORM_MIN_WAVES
#define LMX_XFORM_MIN_WAVES 4 // waves per SIMD the register allocation aims at: 4 = 108 VGPRs, 5 = 96 (five 27-KiB blocks per CU, what LDS admits), 6 spills
#endif
__global__ __launch_bounds__(256, LMX_XFORM_MIN_WAVES) void k_xform_subtree(WorldDevice w, XformSubtree a) {
	__shared__ uint32_t s_count[4];
	__shared__ uint32_t s_base;
	// the previous and the current level's world transforms of a narrow run (<= 256 nodes per level): a node's parent comes out of LDS
	__shared__ double s_px[2][256], s_py[2][256], s_pz[2][256];
	__shared__ float4 s_rot[2][256];
	__shared__ float s_sx[2][256], s_sy[2][256], s_sz[2][256];
	__shared__ uint8_t s_moved[2][256];
	const uint32_t* t0 = a.table + (size_t)blockIdx.x * a.n_levels;
	const uint32_t* t1 = t0 + a.n_levels;
	const uint32_t tid = threadIdx.x;
	bool narrow = true; // block-uniform
	for (uint32_t l = 0; l < a.n_levels; ++l) narrow = narrow && (t1[l] - t0[l] <= 256u);
	if (narrow) {
		// Every load that does not depend on a parent's NEW value goes out ahead: the roots' world transforms now, a node's parent slot /
		// mark / stored local two levels ahead. A level then costs an LDS read, the compose and a barrier that only waits for LDS - the
		// global stores of a level are never waited for (nobody of this block reads them back), so the chain of a run is its first loads
		// plus arithmetic, not a memory round trip (plus a store drain) per level as in the per-level launches.
		XformPre pre0 = {}, pre1 = {};
		bool have0 = false, have1 = false;
		if (a.n_levels > 1 && tid < t1[1] - t0[1]) { pre0 = xform_prefetch(w, t0[1] + tid); have0 = true; }
		if (a.n_levels > 2 && tid < t1[2] - t0[2]) { pre1 = xform_prefetch(w, t0[2] + tid); have1 = true; }
		if (tid < t1[0] - t0[0]) {
			const uint32_t s = t0[0] + tid;
			const float4 r = w.wrot[s];
			s_px[0][tid] = w.wpx[s]; s_py[0][tid] = w.wpy[s]; s_pz[0][tid] = w.wpz[s];
			s_rot[0][tid] = r;
			s_sx[0][tid] = w.wsx[s]; s_sy[0][tid] = w.wsy[s]; s_sz[0][tid] = w.wsz[s];
			s_moved[0][tid] = w.dirty[s] & XF_MOVED;
		}
		lds_barrier();
		for (uint32_t l = 1; l < a.n_levels; ++l) { // block-uniform
			const XformPre cur = pre0;
			const bool mine = have0;
			pre0 = pre1;
			have0 = have1;
			have1 = false;
			if (l + 2 < a.n_levels && tid < t1[l + 2] - t0[l + 2]) { pre1 = xform_prefetch(w, t0[l + 2] + tid); have1 = true; }
			if (mine) {
				const uint32_t s = t0[l] + tid, from = (l - 1) & 1u, to = l & 1u;
				const uint32_t pi = (uint32_t)cur.p - t0[l - 1];
				Xform r;
				bool moved = true;
				if (cur.mark == XF_CLEAN && !s_moved[from][pi]) { // untouched (the reference's DFS does not come here): its children may still need its value
					const float4 wr = w.wrot[s];
					r.pos = DV3{w.wpx[s], w.wpy[s], w.wpz[s]};
					r.rot = Q4{wr.x, wr.y, wr.z, wr.w};
					r.scale = V3{w.wsx[s], w.wsy[s], w.wsz[s]};
					moved = false;
				} else { // xform_node's statements with the parent read from LDS
					Xform parent;
					const float4 pr = s_rot[from][pi];
					parent.pos = DV3{s_px[from][pi], s_py[from][pi], s_pz[from][pi]};
					parent.rot = Q4{pr.x, pr.y, pr.z, pr.w};
					parent.scale = V3{s_sx[from][pi], s_sy[from][pi], s_sz[from][pi]};
					if (cur.mark != XF_SET_WORLD) {
						r = compose(parent, cur.local);
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
					if (cur.mark != XF_CLEAN) {
						const Xform lc = compute_local(parent, r);
						w.lpx[s] = lc.pos.x;
						w.lpy[s] = lc.pos.y;
						w.lpz[s] = lc.pos.z;
						w.lrot[s] = make_float4(lc.rot.x, lc.rot.y, lc.rot.z, lc.rot.w);
						w.lsx[s] = lc.scale.x;
						w.lsy[s] = lc.scale.y;
						w.lsz[s] = lc.scale.z;
					}
					w.dirty[s] = XF_MOVED;
				}
				if (l + 1 < a.n_levels) { // (the last level has no readers)
					s_px[to][tid] = r.pos.x; s_py[to][tid] = r.pos.y; s_pz[to][tid] = r.pos.z;
					s_rot[to][tid] = make_float4(r.rot.x, r.rot.y, r.rot.z, r.rot.w);
					s_sx[to][tid] = r.scale.x; s_sy[to][tid] = r.scale.y; s_sz[to][tid] = r.scale.z;
					s_moved[to][tid] = moved ? XF_MOVED : 0;
				}
			}
			lds_barrier();
		}
	} else {
		for (uint32_t l = 1; l < a.n_levels; ++l) { // block-uniform
			const uint32_t first = t0[l], end = t1[l];
			for (uint32_t s = first + tid; s < end; s += 256u) xform_node(w, s);
			__syncthreads(); // (a workgroup-scope release / acquire: the next level's parents are this level's nodes)
		}
	}
	// marks, moved list, bound spheres. The moved list takes ONE reservation per block: every wave appending for itself is an atomic on one
	// address per 64 nodes - 15.6 k of them for 10^6 moved nodes, ~180 us at the ~90 per microsecond one address retires (measured: 203 us).
	// Inside the block's range a wave's entries of one step are consecutive (ballot + mbcnt ranks): a step writes one run of <= 64
	// 56-byte transforms and one of ids, not 64 scattered ones (a thread numbering its own slots: 65 us for 10^6 moved nodes, 23 untracked).
	const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), wave = tid >> 6;
	uint32_t wave_at = 0; // wave-uniform: the wave's next free entry
	if (a.count != nullptr) {
		uint32_t wave_total = 0;
		for (uint32_t l = 0; l < a.n_levels; ++l)
			for (uint32_t s0 = t0[l] + wave * 64u; s0 < t1[l]; s0 += 256u) { // wave-uniform trip count
				const uint32_t s = s0 + lane;
				wave_total += (uint32_t)__popcll(__ballot(s < t1[l] && (w.dirty[s] & XF_MOVED) != 0));
			}
		if (lane == 0) s_count[wave] = wave_total;
		__syncthreads();
		if (tid == 0) {
			const uint32_t total = s_count[0] + s_count[1] + s_count[2] + s_count[3];
			s_base = total ? atomicAdd(a.count, total) : 0u;
		}
		__syncthreads();
		wave_at = s_base;
		for (uint32_t k = 0; k < wave; ++k) wave_at += s_count[k];
	}
	for (uint32_t l = 0; l < a.n_levels; ++l) {
		for (uint32_t s0 = t0[l] + wave * 64u; s0 < t1[l]; s0 += 256u) { // the order of the count above
			const uint32_t s = s0 + lane;
			const bool live = s < t1[l];
			const uint8_t mark = live ? w.dirty[s] : 0;
			if (mark != 0) w.dirty[s] = 0;
			const bool collect = (mark & XF_MOVED) != 0 && a.count != nullptr;
			const uint64_t mask = __ballot(collect);
			const uint32_t at = wave_at + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
			wave_at += (uint32_t)__popcll(mask);
			const uint32_t dyn = live && a.bound_dyn_of_slot != nullptr ? a.bound_dyn_of_slot[s] : 0xffffffffu;
			if (dyn == 0xffffffffu && !collect) continue;
			const float4 r = w.wrot[s];
			const double px = w.wpx[s], py = w.wpy[s], pz = w.wpz[s];
			const float sx = w.wsx[s], sy = w.wsy[s], sz = w.wsz[s];
			if (dyn != 0xffffffffu) { // onModelInstanceMoved (render_module.cpp:1544-1554), as k_sphere_refresh: every bound entity, every propagation
				a.dyn_px[dyn] = px; a.dyn_py[dyn] = py; a.dyn_pz[dyn] = pz;
				const float mr = a.bound_radius_of_slot[s];
				if (!(mr < 0.f)) a.dyn_radius[dyn] = mr * maximum3(sx, sy, sz);
			}
			if (collect && at < a.cap) {
				TransformAoS t;
				t.pos[0] = px; t.pos[1] = py; t.pos[2] = pz;
				t.rot[0] = r.x; t.rot[1] = r.y; t.rot[2] = r.z; t.rot[3] = r.w;
				t.scale[0] = sx; t.scale[1] = sy; t.scale[2] = sz;
				t.pad = 0.f;
				a.out_entity[at] = a.entity_of_slot[s];
				reinterpret_cast<TransformAoS*>(a.out_tr)[at] = t;
			}
		}
	}
}

__device__ __forceinline__ Xform load_world(const WorldDevice& w, int32_t s) {
	Xform x;
	const float4 r = w.wrot[s];
	x.pos = DV3{w.wpx[s], w.wpy[s], w.wpz[s]};
	x.rot = Q4{r.x, r.y, r.z, r.w};
	x.scale = V3{w.wsx[s], w.wsy[s], w.wsz[s]};
	return x;
}

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

hipError_t launch_xform_subtree(hipStream_t s, const WorldDevice& w, const XformSubtree& a, uint32_t n_runs) {
	if (!n_runs) return hipSuccess;
	hipLaunchKernelGGL(k_xform_subtree, dim3(n_runs), dim3(256), 0, s, w, a);
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
