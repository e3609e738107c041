#!/usr/bin/env python
"""TEST INFRASTRUCTURE. Cuts the reference's dual-quaternion skinning out of its SHADER sources, where they lie, into include fragments
under a temporary directory (oracle/Makefile deletes it after the compile; nothing is committed), so that oracle/ref/hlsl_shim.cpp can
compile the REFERENCE'S OWN shader text - as C++, against a 60-line float3 / float4 / float2x4 shim - into oracle/_ref/liblmx_ref.so:

    hlsl_blend.inc       data/shaders/surface_base.hlsli   the SKINNED branch of mainVS: `float2x4 dq = mul(getBones(...` ... `dq *= 1 / length(dq[0]);`
    hlsl_transform.inc   data/shaders/common.hlsli         float3 transformByDualQuat(float2x4 dq, float3 pos)

HLSL cannot be compiled here (no dxc / fxc), and it does not pin the association of floating-point expressions the way C++ with
-ffp-contract=off does - the GPU's shader compiler may contract a * b + c. What this pins is the EXPRESSIONS: which products, which
signs, which hemisphere test, which normalisation - the things a hand restatement can get wrong. Pieces are located by anchor strings;
the script fails loudly if an anchor is missing.

    python oracle/ref/slice_hlsl.py /root/reference <tmp>/gen
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from slice_animation import block  # noqa: E402


def main():
    ref, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    shaders = os.path.join(ref, "data", "shaders")
    sb = open(os.path.join(shaders, "surface_base.hlsli")).read()
    a = sb.index("float2x4 dq = mul(getBones(input, input.indices.x), input.weights.x);")
    end = "dq *= 1 / length(dq[0]);"
    b = sb.index(end, a) + len(end)
    blend = sb[a:b]
    assert blend.count("getBones(") == 10 and blend.count("dq +=") == 3 and "< 0 ? -input.weights.w : input.weights.w" in blend, "the SKINNED branch changed shape"
    open(os.path.join(out, "hlsl_blend.inc"), "w").write(blend + "\n")
    cm = open(os.path.join(shaders, "common.hlsli")).read()
    fn = block(cm, "float3 transformByDualQuat(float2x4 dq, float3 pos) {")
    assert "cross(dq[0].xyz, dq[1].xyz)" in fn
    open(os.path.join(out, "hlsl_transform.inc"), "w").write(fn + "\n")


if __name__ == "__main__":
    main()
