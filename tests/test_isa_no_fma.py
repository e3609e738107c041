"""Bit-exact parity with the reference's scalar fp32 arithmetic (-msse2, no FMA contraction: scripts/genie.lua:301-315) rests on the
device code rounding every product and every sum on its own. hipcc contracts a * b + c into one fused instruction by default; the
kernels are built with -ffp-contract=off (lumixengine_amd/build.py) and written without fmaf(). This test compiles them the way the
build does and looks at the ISA (no GPU needed): the kernels whose results are compared bit for bit - culling, transform propagation,
pose / palette, the EXACT skinning mode, animation sampling - must not contain a single fp32 / fp64 fused multiply-add. (The FUSED
skinning mode and the dual-quaternion blend are allowed to: their bar is 1e-5.)"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lumixengine_amd", "csrc")
FMA = re.compile(r"\b(v_fma_f32|v_fma_f64|v_fmac_f32|v_fmac_f64|v_pk_fma_f32|v_mad_f32|v_mac_f32|v_fmaak_f32|v_fmamk_f32|v_dot2c?_f32)")  # (no trailing boundary: the e32 / e64 / dpp encodings append _e32 ...)


def isa_of(source, tmp):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from lumixengine_amd import build as B

    out = tmp / (os.path.splitext(source)[0] + ".s")
    flags = [f for f in B.FLAGS if f not in ("-c", "-fPIC")]
    assert "-ffp-contract=off" in flags, "the build no longer passes -ffp-contract=off"
    r = subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", "-x", "hip", "-o", str(out), os.path.join(CSRC, source)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return out.read_text().splitlines()


def kernels(lines):
    """{mangled kernel name: instruction lines} for every kernel of the file"""
    out, name = {}, None
    for l in lines:
        m = re.match(r"^(_ZN\S+):", l)
        if m and name is None:
            name = m.group(1)
            out[name] = []
            continue
        if name is not None:
            if l.startswith("\t") and not l.lstrip().startswith((";", ".")):
                out[name].append(l.strip())
            if "s_endpgm" in l:
                name = None
    return out


@pytest.mark.parametrize("source, must_be_exact, may_fuse", [
    ("cull_kernels.hip", ["k_cull_tile", "k_cull_dynamic"], []),
    ("xform_kernels.hip", ["k_xform_level", "k_xform_subtree", "k_sphere_refresh", "k_bone_attach"], []),
    ("skin_kernels.hip", ["k_pose_palette", "k_pose_blend", "k_skin_sharedILi1E", "k_skin_verticesILi1E", "k_skin_multiILi1ELi1E", "k_skin_multiILi2ELi1E", "k_skin_multiILi4ELi1E", "k_skin_multiILi16ELi1E"],
     ["k_skin_sharedILi0E", "k_skin_verticesILi0E", "k_skin_verticesILi2E"]),
    ("anim_kernels.hip", ["k_anim_update"], []),
])
def test_bit_exact_kernels_contain_no_fused_multiply_add(tmp_path, source, must_be_exact, may_fuse):
    if not os.path.exists(os.path.join(CSRC, source)):
        pytest.skip(f"{source} not in this tree")
    ks = kernels(isa_of(source, tmp_path))
    checked = 0
    for name, body in ks.items():
        if not any(tag in name for tag in must_be_exact):
            continue
        if any(tag in name for tag in may_fuse):
            continue
        # Fused multiply-adds the COMPILER's own expansions use are fine - they produce the correctly rounded / exact result of ONE source
        # operation: IEEE division (v_div_scale, v_rcp, Newton steps with v_fma, v_div_fmas, v_div_fixup), correctly rounded sqrt, and
        # small unsigned integer divisions through v_rcp_iflag_f32 (waves per frustum = WAVES / n_frusta). They sit within a couple of
        # dozen instructions behind the expansion's opening instruction; anything else is a contraction of the algorithm's arithmetic.
        opener = re.compile(r"\b(v_div_scale_f(32|64)|v_rcp_(iflag_)?f(32|64)|v_rsq_f(32|64)|v_sqrt_f(32|64))")
        bad = [l for i, l in enumerate(body) if FMA.search(l) and not any(opener.search(p) for p in body[max(0, i - 28) : i])]
        assert not bad, f"{source}: {name} contains fused multiply-adds: {bad[:5]}"
        checked += 1
    assert checked >= len([t for t in must_be_exact if "ILi" not in t]), (source, checked, list(ks)[:5])
    # the FUSED skinning kernels do use them (that is their point): the pattern above would see them
    for tag in may_fuse[:1]:
        fused = [n for n in ks if tag in n]
        assert fused and any(FMA.search(l) for l in ks[fused[0]]), f"{tag}: expected v_pk_fma_f32 / v_fma_f32 in the FUSED mode's ISA"
