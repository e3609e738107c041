#!/usr/bin/env python
"""TEST INFRASTRUCTURE. Cuts the pose / palette code out of the reference tree, where it lies, into include fragments under a temporary
directory (oracle/Makefile deletes it after the compile; nothing is committed), so that oracle/ref/pose_shim.cpp can compile the
REFERENCE'S OWN code into oracle/_ref/liblmx_ref.so:

    simd_sse.inc          src/core/simd.h            the `float4 = __m128` branch (see slice_animation.py)
    simd_soa.inc          src/core/simd_math.h       SOAVec3 / SOAQuat / SIMDLocalRigidTransform / SIMDDualQuat, transposeStore, loadTranspose, cross,
                                                     rotate, the SOA operators, toDualQuat(SIMDLocalRigidTransform)
    pose_methods.inc      src/renderer/pose.cpp      Pose::blend, Pose::computeAbsolute (4-wide path + scalar tail), Pose::computeRelative
    pose_dual_quats.inc   src/renderer/pipeline.cpp  PipelineImpl::computeSkeletonDualQuats (4-wide batches + scalar tail)
    model_statics.inc     src/renderer/model.cpp     static invert, evaluateSkin, computeSkinMatrices
    model_soa.inc         src/renderer/model.h       struct SOATransform, Mesh::Skin

pose.cpp / model.cpp / pipeline.cpp cannot be compiled whole here (resource system, renderer), and on Linux core/simd.h selects a scalar
float4 that lacks f4LoadUnaligned / f4Transpose, which pose.cpp needs.

    python oracle/ref/slice_pose.py /root/reference <tmp>/gen
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from slice_animation import block, sse_branch  # noqa: E402


def main():
    ref, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    src = os.path.join(ref, "src")

    def put(name, text):
        open(os.path.join(out, name), "w").write(text + "\n")

    put("simd_sse.inc", sse_branch(src))
    sm = open(os.path.join(src, "core", "simd_math.h")).read()
    soa = sm[sm.index("struct SOAVec3 {"):sm.index("LUMIX_FORCE_INLINE float4 simd_nlerp(float4 q1, float4 q2, float t)")]
    assert "toDualQuat" in soa and "SOAQuat operator *" in soa and "loadTranspose" in soa
    put("simd_soa.inc", soa)
    pc = open(os.path.join(src, "renderer", "pose.cpp")).read()
    put("pose_methods.inc", "\n\n".join(block(pc, a) for a in ("void Pose::blend(Pose& rhs, float weight)", "void Pose::computeAbsolute(Model& model) {",
                                                             "void Pose::computeRelative(Model& model)")))
    pl = open(os.path.join(src, "renderer", "pipeline.cpp")).read()
    dq = block(pl, "void computeSkeletonDualQuats(const ModelInstance* mi) {")
    assert "toDualQuat(tmp * inv_bind_tr)" in dq and "f4Stream" in dq
    put("pose_dual_quats.inc", dq)
    mc = open(os.path.join(src, "renderer", "model.cpp")).read()
    put("model_statics.inc", "\n\n".join(block(mc, a) for a in ("static LocalRigidTransform invert(const LocalRigidTransform& tr)",
                                                              "static Vec3 evaluateSkin(Vec3& p, Mesh::Skin s, const Matrix* matrices)",
                                                              "static void computeSkinMatrices(const Pose& pose, const Model& model, Matrix* matrices) {")))
    mh = open(os.path.join(src, "renderer", "model.h")).read()
    mesh = block(mh, "struct LUMIX_RENDERER_API Mesh {", trailer=";")
    put("model_soa.inc", block(mh, "struct SOATransform {", trailer=";") + "\n\nstruct Mesh {\n\t" + block(mesh, "struct Skin {", trailer=";") + "\n};")
    print("sliced", sorted(f for f in os.listdir(out)))


if __name__ == "__main__":
    main()
