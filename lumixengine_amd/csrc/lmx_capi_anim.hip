// lmx_capi_anim.hip — animation sampling entry points (include/lumix_mi355.h, "animation" section): Animation resources are
// flattened into concatenated device tables, every skinned instance is an Animable {animation, time}, and lmx_anim_update is
// AnimationModuleImpl::updateAnimable for all of them at once; its output is the relative pose lmx_skin_run consumes.
#include "lmx_context.h"

using namespace lmx;

namespace {

template <typename T> int upload_vec(LmxContext* ctx, DevBuf<T>& buf, const std::vector<T>& v) {
	LMX_HIP(ctx, buf.reserve(std::max<size_t>(v.size(), 1)));
	if (!v.empty()) LMX_HIP(ctx, hipMemcpy(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
	return LMX_OK;
}

int anim_upload_tables(LmxContext* ctx) {
	AnimState& an = ctx->anim;
	if (!an.tables_dirty) return LMX_OK;
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (int rc = upload_vec(ctx, an.d_anims, an.anims)) return rc;
	if (int rc = upload_vec(ctx, an.d_src, an.src)) return rc;
	if (int rc = upload_vec(ctx, an.d_ct, an.ct)) return rc;
	if (int rc = upload_vec(ctx, an.d_tt, an.tt)) return rc;
	if (int rc = upload_vec(ctx, an.d_cr, an.cr)) return rc;
	if (int rc = upload_vec(ctx, an.d_rt, an.rt)) return rc;
	if (int rc = upload_vec(ctx, an.d_tstream, an.tstream)) return rc;
	if (int rc = upload_vec(ctx, an.d_rstream, an.rstream)) return rc;
	if (int rc = upload_vec(ctx, an.d_root_t, an.root_t)) return rc;
	if (int rc = upload_vec(ctx, an.d_root_r, an.root_r)) return rc;
	if (int rc = upload_vec(ctx, an.d_rel_pos, an.rel_pos)) return rc;
	if (int rc = upload_vec(ctx, an.d_rel_rot, an.rel_rot)) return rc;
	an.tables_dirty = false;
	return LMX_OK;
}

} // namespace

extern "C" {

int lmx_anim_add(LmxContext* ctx, const LmxAnimation* a, uint32_t* out_animation) {
	LMX_CHECK_CTX(ctx);
	if (!a || !out_animation) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null animation / out");
	if (!(a->fps > 0.f) || !a->frame_count || !a->length) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "fps, frame_count and length must be positive");
	if ((a->n_const_translations && !a->const_translations) || (a->n_translations && !a->translations) || (a->n_const_rotations && !a->const_rotations) ||
		(a->n_rotations && !a->rotations))
		return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null track array");
	const uint64_t t_need = ((uint64_t)a->translations_frame_size_bits * (a->frame_count + 1) + 7) / 8, r_need = ((uint64_t)a->rotations_frame_size_bits * (a->frame_count + 1) + 7) / 8;
	if ((a->n_translations && (!a->translation_stream || a->translation_stream_size < t_need)) || (a->n_rotations && (!a->rotation_stream || a->rotation_stream_size < r_need)))
		return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "track streams must hold frame_count + 1 frames (animation.cpp:464): need %llu / %llu bytes", (unsigned long long)t_need, (unsigned long long)r_need);
	if ((a->root_translation_track >= 0 && ((uint32_t)a->root_translation_track >= a->n_translations || !a->root_pose_translations)) ||
		(a->root_rotation_track >= 0 && ((uint32_t)a->root_rotation_track >= a->n_rotations || !a->root_pose_rotations)))
		return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "root-motion track index out of range or pose arrays missing");
	AnimState& an = ctx->anim;
	uint32_t max_bone = 0; // m_max_accessed_bone_index, animation.cpp:369-393
	for (uint32_t i = 0; i < a->n_const_translations; ++i) max_bone = std::max<uint32_t>(max_bone, a->const_translations[i].bone_index);
	for (uint32_t i = 0; i < a->n_translations; ++i) max_bone = std::max<uint32_t>(max_bone, a->translations[i].bone_index);
	for (uint32_t i = 0; i < a->n_const_rotations; ++i) max_bone = std::max<uint32_t>(max_bone, a->const_rotations[i].bone_index);
	for (uint32_t i = 0; i < a->n_rotations; ++i) max_bone = std::max<uint32_t>(max_bone, a->rotations[i].bone_index);
	if (max_bone >= LMX_MAX_BONES) return fail(ctx, LMX_ERR_CAPACITY, "track bone index %u >= %d (Model::Bone::MAX_COUNT)", max_bone, LMX_MAX_BONES);
	// per-bone sources: the reference runs the four track lists in order; with at most one translation and one rotation source per
	// bone the order is irrelevant and a lane can look its bone up directly
	std::vector<int32_t> src(2 * (size_t)(max_bone + 1), -1);
	auto claim = [&](uint32_t bone, int which, int32_t code) {
		int32_t& s = src[2 * bone + which];
		if (s != -1) return false;
		s = code;
		return true;
	};
	for (uint32_t i = 0; i < a->n_const_translations; ++i)
		if (!claim(a->const_translations[i].bone_index, 0, (int32_t)(2 * i))) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bone %u has two translation tracks", a->const_translations[i].bone_index);
	for (uint32_t i = 0; i < a->n_translations; ++i) {
		const LmxAnimTranslationTrack& t = a->translations[i];
		if (!claim(t.bone_index, 0, (int32_t)(2 * i + 1))) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bone %u has two translation tracks", t.bone_index);
		if (t.bitsizes[0] + t.bitsizes[1] + t.bitsizes[2] > 57u) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "translation track %u: more than 57 bits per frame cannot be read with one 64-bit load (animation.cpp:323-325)", i);
		if (t.offset_bits + t.bitsizes[0] + t.bitsizes[1] + t.bitsizes[2] > a->translations_frame_size_bits) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "translation track %u exceeds the frame size", i);
	}
	for (uint32_t i = 0; i < a->n_const_rotations; ++i)
		if (!claim(a->const_rotations[i].bone_index, 1, (int32_t)(2 * i))) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bone %u has two rotation tracks", a->const_rotations[i].bone_index);
	for (uint32_t i = 0; i < a->n_rotations; ++i) {
		const LmxAnimRotationTrack& t = a->rotations[i];
		if (!claim(t.bone_index, 1, (int32_t)(2 * i + 1))) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bone %u has two rotation tracks", t.bone_index);
		if (1u + t.bitsizes[0] + t.bitsizes[1] + t.bitsizes[2] > 57u) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "rotation track %u: more than 57 bits per frame", i);
		if (t.offset_bits + 1u + t.bitsizes[0] + t.bitsizes[1] + t.bitsizes[2] > a->rotations_frame_size_bits) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "rotation track %u exceeds the frame size", i);
		if (t.skipped_channel > 3) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "rotation track %u: skipped_channel %u", i, t.skipped_channel);
	}
	AnimDevice d;
	memset(&d, 0, sizeof(d));
	d.fps = a->fps; d.frame_count = a->frame_count; d.length = a->length; d.tfs_bits = a->translations_frame_size_bits; d.rfs_bits = a->rotations_frame_size_bits;
	d.max_bone = max_bone;
	d.src_off = (uint32_t)an.src.size();
	d.ct_off = (uint32_t)an.ct.size(); d.tt_off = (uint32_t)an.tt.size(); d.cr_off = (uint32_t)an.cr.size(); d.rt_off = (uint32_t)an.rt.size();
	d.tstream_off = (uint32_t)an.tstream.size(); d.rstream_off = (uint32_t)an.rstream.size();
	d.root_translation_track = a->root_translation_track; d.root_rotation_track = a->root_rotation_track;
	d.root_off = (uint32_t)(an.root_r.size());
	an.src.insert(an.src.end(), src.begin(), src.end());
	an.ct.insert(an.ct.end(), a->const_translations, a->const_translations + a->n_const_translations);
	an.tt.insert(an.tt.end(), a->translations, a->translations + a->n_translations);
	an.cr.insert(an.cr.end(), a->const_rotations, a->const_rotations + a->n_const_rotations);
	an.rt.insert(an.rt.end(), a->rotations, a->rotations + a->n_rotations);
	auto append_stream = [](std::vector<uint8_t>& dst, const uint8_t* src_bytes, uint64_t need) { // 8-byte aligned start, 16 bytes of slack for the 64-bit reads
		if (need) dst.insert(dst.end(), src_bytes, src_bytes + need);
		dst.resize((dst.size() + 16 + 7) & ~(size_t)7, 0);
	};
	append_stream(an.tstream, a->translation_stream, a->n_translations ? t_need : 0);
	append_stream(an.rstream, a->rotation_stream, a->n_rotations ? r_need : 0);
	const size_t frames = (size_t)a->frame_count + 1;
	for (size_t f = 0; f < frames; ++f) {
		for (int k = 0; k < 3; ++k) an.root_t.push_back(a->root_translation_track >= 0 ? a->root_pose_translations[3 * f + k] : 0.f);
		an.root_r.push_back(a->root_rotation_track >= 0 ? make_float4(a->root_pose_rotations[4 * f], a->root_pose_rotations[4 * f + 1], a->root_pose_rotations[4 * f + 2], a->root_pose_rotations[4 * f + 3])
		                                                : make_float4(0.f, 0.f, 0.f, 1.f));
	}
	*out_animation = (uint32_t)an.anims.size();
	an.anims.push_back(d);
	an.tables_dirty = true;
	return LMX_OK;
}

int lmx_anim_set_model_pose(LmxContext* ctx, uint32_t model, const LmxLocalRigidTransform* relative, uint32_t n_bones) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	AnimState& an = ctx->anim;
	if (model >= sk.models.size()) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "unknown skin model %u", model);
	if (!relative || n_bones != sk.models[model].n_bones) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "model %u has %u bones", model, sk.models[model].n_bones);
	const size_t total = sk.parents.size(); // bones of all models
	an.rel_pos.resize(total * 3, 0.f);
	an.rel_rot.resize(total, make_float4(0.f, 0.f, 0.f, 1.f));
	const uint32_t off = sk.models[model].bone_offset;
	for (uint32_t b = 0; b < n_bones; ++b) {
		for (int k = 0; k < 3; ++k) an.rel_pos[3 * (size_t)(off + b) + k] = relative[b].pos[k];
		an.rel_rot[off + b] = make_float4(relative[b].rot[0], relative[b].rot[1], relative[b].rot[2], relative[b].rot[3]);
	}
	an.tables_dirty = true;
	return LMX_OK;
}

int lmx_anim_set_animables(LmxContext* ctx, uint32_t n_instances, const uint32_t* animation, const uint32_t* time) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	AnimState& an = ctx->anim;
	if (n_instances != sk.inst.size()) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "the skin instance table has %zu instances", sk.inst.size());
	if (n_instances && (!animation || !time)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null animation / time array");
	for (uint32_t i = 0; i < n_instances; ++i)
		if (animation[i] != LMX_ANIM_NONE && animation[i] >= an.anims.size()) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "instance %u: unknown animation %u", i, animation[i]);
	LMX_HIP(ctx, an.d_anim_of.reserve(std::max<size_t>(n_instances, 1)));
	LMX_HIP(ctx, an.d_time_of.reserve(std::max<size_t>(n_instances, 1)));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (n_instances) {
		LMX_HIP(ctx, hipMemcpy(an.d_anim_of.p, animation, (size_t)n_instances * sizeof(uint32_t), hipMemcpyHostToDevice));
		LMX_HIP(ctx, hipMemcpy(an.d_time_of.p, time, (size_t)n_instances * sizeof(uint32_t), hipMemcpyHostToDevice));
	}
	an.n_animables = n_instances;
	return LMX_OK;
}

int lmx_anim_set_weight(LmxContext* ctx, float weight) {
	LMX_CHECK_CTX(ctx);
	if (!(weight >= 0.f && weight <= 1.f)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "weight %g outside [0, 1]", (double)weight);
	ctx->anim.weight = weight;
	return LMX_OK;
}

int lmx_anim_update(LmxContext* ctx, float time_delta) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	AnimState& an = ctx->anim;
	if (sk.inst.empty()) return LMX_OK;
	if (an.n_animables != sk.inst.size()) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_anim_set_animables has not been called for this instance table");
	if (an.rel_rot.size() != sk.parents.size()) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_anim_set_model_pose is missing for a model added later");
	if (int rc = anim_upload_tables(ctx)) return rc;
	AnimTables t;
	t.src = an.d_src.p; t.const_translations = an.d_ct.p; t.translations = an.d_tt.p; t.const_rotations = an.d_cr.p; t.rotations = an.d_rt.p;
	t.translation_stream = an.d_tstream.p; t.rotation_stream = an.d_rstream.p; t.root_translations = an.d_root_t.p; t.root_rotations = an.d_root_r.p;
	ProfScope ps(ctx, LMX_K_ANIM_UPDATE);
	LMX_HIP(ctx, launch_anim_update(ctx->stream, sk.d_inst.p, (uint32_t)sk.inst.size(), an.d_anims.p, t, an.d_anim_of.p, an.d_time_of.p, time_delta, an.weight,
		an.d_rel_pos.p, an.d_rel_rot.p, sk.d_pose_pos.p, sk.d_pose_rot.p));
	// the library's pose buffers hold fresh relative poses: the source of the next lmx_skin_run
	sk.borrowed_pos = nullptr;
	sk.borrowed_rot = nullptr;
	sk.poses_uploaded = true;
	sk.pose_is_absolute = false;
	return LMX_OK;
}

int lmx_anim_eval_blend_stacks(LmxContext* ctx, uint32_t n_instances, const uint32_t* first_sample, const LmxBlendSample* samples) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	AnimState& an = ctx->anim;
	if (n_instances != sk.inst.size()) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "the skin instance table has %zu instances", sk.inst.size());
	if (!n_instances) return LMX_OK;
	if (!first_sample) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "null first_sample");
	if (an.rel_rot.size() != sk.parents.size()) return fail(ctx, LMX_ERR_NOT_BUILT, "lmx_anim_set_model_pose is missing for a model added later");
	for (uint32_t i = 0; i < n_instances; ++i)
		if (first_sample[i + 1] < first_sample[i]) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "first_sample decreases at instance %u", i);
	const uint32_t n_samples = first_sample[n_instances];
	if (first_sample[0] != 0 || (n_samples && !samples)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "first_sample[0] must be 0 and samples non-null");
	for (uint32_t k = 0; k < n_samples; ++k) {
		if (samples[k].animation >= an.anims.size()) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "sample %u: unknown animation %u", k, samples[k].animation);
		if (!(samples[k].weight >= 0.f && samples[k].weight <= 1.f)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "sample %u: weight %g outside [0, 1]", k, (double)samples[k].weight);
	}
	if (int rc = anim_upload_tables(ctx)) return rc;
	LMX_HIP(ctx, an.d_first_sample.reserve((size_t)n_instances + 1));
	LMX_HIP(ctx, an.d_samples.reserve(std::max<size_t>(n_samples, 1)));
	// the caller's arrays are pageable: the copies below complete before this returns, and the stream orders them against the
	// previous frame's kernel that still reads the same device buffers
	LMX_HIP(ctx, hipMemcpyAsync(an.d_first_sample.p, first_sample, ((size_t)n_instances + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
	if (n_samples) LMX_HIP(ctx, hipMemcpyAsync(an.d_samples.p, samples, (size_t)n_samples * sizeof(LmxBlendSample), hipMemcpyHostToDevice, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	AnimTables t;
	t.src = an.d_src.p; t.const_translations = an.d_ct.p; t.translations = an.d_tt.p; t.const_rotations = an.d_cr.p; t.rotations = an.d_rt.p;
	t.translation_stream = an.d_tstream.p; t.rotation_stream = an.d_rstream.p; t.root_translations = an.d_root_t.p; t.root_rotations = an.d_root_r.p;
	ProfScope ps(ctx, LMX_K_ANIM_UPDATE);
	LMX_HIP(ctx, launch_anim_blend_stack(ctx->stream, sk.d_inst.p, (uint32_t)sk.inst.size(), an.d_anims.p, t, (uint32_t)an.anims.size(), an.d_samples.p,
		an.d_first_sample.p, an.d_rel_pos.p, an.d_rel_rot.p, sk.d_pose_pos.p, sk.d_pose_rot.p));
	sk.borrowed_pos = nullptr;
	sk.borrowed_rot = nullptr;
	sk.poses_uploaded = true;
	sk.pose_is_absolute = false;
	return LMX_OK;
}

int lmx_anim_read_times(LmxContext* ctx, uint32_t* time, uint32_t n_instances) {
	LMX_CHECK_CTX(ctx);
	AnimState& an = ctx->anim;
	if (n_instances != an.n_animables || (n_instances && !time)) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "expected %u animables", an.n_animables);
	if (n_instances) LMX_HIP(ctx, hipMemcpyAsync(time, an.d_time_of.p, (size_t)n_instances * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

int lmx_anim_read_pose(LmxContext* ctx, uint32_t instance, float* out_pos, float* out_rot, uint32_t cap_bones) {
	LMX_CHECK_CTX(ctx);
	SkinState& sk = ctx->skin;
	if (instance >= sk.inst.size()) return fail(ctx, LMX_ERR_INVALID_ARGUMENT, "bad instance");
	const SkinInstance& in = sk.inst[instance];
	if (cap_bones < in.n_bones) return fail(ctx, LMX_ERR_CAPACITY, "need room for %u bones", in.n_bones);
	if (sk.pose_is_absolute || !sk.poses_uploaded || sk.borrowed_pos) return fail(ctx, LMX_ERR_NOT_BUILT, "the pose buffers do not hold a relative pose (call after lmx_anim_update, before lmx_skin_run)");
	if (out_pos) LMX_HIP(ctx, hipMemcpyAsync(out_pos, sk.d_pose_pos.p + (size_t)in.bone_offset * 3, (size_t)in.n_bones * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	if (out_rot) LMX_HIP(ctx, hipMemcpyAsync(out_rot, sk.d_pose_rot.p + in.bone_offset, (size_t)in.n_bones * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
	LMX_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return LMX_OK;
}

} // extern "C"
