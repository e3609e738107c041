#!/usr/bin/env python
"""TEST INFRASTRUCTURE. Cuts the animation-sampling code out of the reference tree, where it lies, into include fragments under
a temporary directory (oracle/Makefile deletes it after the compile; nothing is committed), so that oracle/ref/anim_shim.cpp can compile the REFERENCE'S OWN sampling code
into oracle/_ref/liblmx_ref.so:

    simd_sse.inc      src/core/simd.h          the `float4 = __m128` branch (the file selects it on MSVC only; the intrinsics are plain SSE)
    simd_nlerp.inc    src/core/simd_math.h     simd_nlerp(float4, float4, float) and simd_nlerp(Quat, Quat, float)
    anim_time.inc     src/animation/animation.h   struct Time
    anim_tracks.inc   src/animation/animation.h   Animation::{ConstTranslationTrack, TranslationTrack, ConstRotationTrack, RotationTrack, SampleContext}
    anim_members.inc  src/animation/animation.h   the data members AnimationSampler reads (tracks, root motion, streams, frame info)
    anim_sampler.inc  src/animation/animation.cpp struct AnimationSampler (getRotation, maskRootMotion, getRelativePose<use_mask, use_weight>)
    anim_methods.inc  src/animation/animation.cpp Animation::getRelativePose, unpackChannel, Animation::getTranslation

animation.cpp cannot be compiled whole here: Animation is a Resource (resource manager, file system, streams) and the file as a whole
needs the engine. Pieces are located by anchor strings + brace matching, and the script fails loudly if an anchor is missing.

    python oracle/ref/slice_animation.py /root/reference <tmp>/gen
"""
import os
import sys


def block(text, anchor, open_ch="{", close_ch="}", trailer=""):
    """text from `anchor` through the brace block that follows it (+ an optional trailer such as ';')"""
    a = text.index(anchor)
    i = text.index(open_ch, a)
    depth = 0
    while True:
        c = text[i]
        if c == open_ch:
            depth += 1
        elif c == close_ch:
            depth -= 1
            if depth == 0:
                break
        i += 1
    end = i + 1
    if trailer:
        end = text.index(trailer, end) + len(trailer)
    return text[a:end]


def sse_branch(src):
    """the `float4 = __m128` branch of src/core/simd.h without its operator overloads"""
    simd = open(os.path.join(src, "core", "simd.h")).read()
    cond = "#if defined _WIN32 && !defined __clang__"
    first = simd.index(cond)
    second = simd.index(cond, first + 1)
    body_start = simd.index("\n", second) + 1
    body_end = simd.index("\n#else", body_start)
    sse = simd[body_start:body_end]
    assert "using float4 = __m128;" in sse and "f4LoadUnaligned" in sse
    # The branch overloads + - * on float4. MSVC's __m128 is a union, so that is legal there; for GCC / clang __m128 is a built-in
    # vector type that cannot be overloaded and ALREADY has element-wise + - * (they lower to the same addps / subps / mulps, a scalar
    # operand is broadcast like _mm_set_ps1): the overloads are dropped. The one that differs in its zero sign, unary minus
    # (_mm_sub_ps(0, a)), is not used by the sampling code.
    for sig in ("float4 operator +(float4 a, float4 b)", "float4 operator -(float4 a, float4 b)", "float4 operator -(float4 a)", "float4 operator *(float4 a, float4 b)",
                "float4 operator *(float4 a, float b)"):
        cut = block(sse, "LUMIX_FORCE_INLINE " + sig)
        body = cut[cut.index("{"):]
        assert body.count("_mm_") in (1, 2), body
        sse = sse.replace(cut, "// (operator overload dropped: built-in vector operator, see slice_animation.py)")
    return sse


def main():
    ref, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    src = os.path.join(ref, "src")
    sse = sse_branch(src)
    open(os.path.join(out, "simd_sse.inc"), "w").write(sse + "\n")

    sm = open(os.path.join(src, "core", "simd_math.h")).read()
    nl = block(sm, "LUMIX_FORCE_INLINE float4 simd_nlerp(float4 q1, float4 q2, float t)") + "\n\n" + block(sm, "LUMIX_FORCE_INLINE Quat simd_nlerp(const Quat& a, const Quat& b, float t)")
    open(os.path.join(out, "simd_nlerp.inc"), "w").write(nl + "\n")

    ah = open(os.path.join(src, "animation", "animation.h")).read()
    open(os.path.join(out, "anim_time.inc"), "w").write(block(ah, "struct Time {", trailer=";") + "\n")
    tracks = "\n\n".join(block(ah, "struct %s {" % n, trailer=";") for n in ("ConstTranslationTrack", "TranslationTrack", "ConstRotationTrack", "RotationTrack", "SampleContext"))
    open(os.path.join(out, "anim_tracks.inc"), "w").write(tracks + "\n")
    m0 = ah.index("Array<TranslationTrack> m_translations;")
    m1 = ah.index("u32 m_max_accessed_bone_index = 0;") + len("u32 m_max_accessed_bone_index = 0;")
    members = "\n".join(l for l in ah[m0:m1].split("\n") if "m_mem" not in l and "m_skeleton" not in l)
    assert "m_root_motion" in members and "m_rotation_stream" in members
    open(os.path.join(out, "anim_members.inc"), "w").write(members + "\n")

    ac = open(os.path.join(src, "animation", "animation.cpp")).read()
    sampler = block(ac, "struct AnimationSampler {", trailer=";")
    assert "getRelativePose" in sampler and "getRotation" in sampler
    open(os.path.join(out, "anim_sampler.inc"), "w").write(sampler + "\n")
    methods = "\n\n".join([block(ac, "void Animation::getRelativePose(const SampleContext& ctx) {"), block(ac, "static float unpackChannel("),
                           block(ac, "Vec3 Animation::getTranslation(u32 frame, const TranslationTrack& track) const {")])
    open(os.path.join(out, "anim_methods.inc"), "w").write(methods + "\n")
    print("sliced", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
