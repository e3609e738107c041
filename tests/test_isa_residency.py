"""Register budgets that decide how many blocks of the hot kernels a CU keeps resident (no GPU needed: hipcc -S for gfx950).

MI355X_MICROARCH.md, "Residency": a 256-thread block is admitted 8 times per CU only up to 80 SGPRs (82-96: 7), and a wave's VGPR count
sets the waves per SIMD (<= 64: 8, <= 72: 7, <= 80: 6, <= 96: 5, <= 128: 4). The headline launch of the cull is a launch of mostly
rejected blocks: its duration is block residency x block lifetime, so a register that creeps past one of these steps is a
measurable regression that no parity test sees. Scratch (spills) is never acceptable in these kernels."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lumixengine_amd", "csrc")


def metadata(source, tmp):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from lumixengine_amd import build as B

    out = tmp / (os.path.splitext(source)[0] + ".s")
    flags = [f for f in B.FLAGS if f not in ("-c", "-fPIC")]
    r = subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", "-x", "hip", "-o", str(out), os.path.join(CSRC, source)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    meta, name = {}, None
    for l in out.read_text().splitlines():
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", l)
        if m:
            name = m.group(1)
            meta[name] = {}
            continue
        if name:
            m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|private_segment_fixed_size|group_segment_fixed_size)\s+(\d+)", l)
            if m:
                meta[name][m.group(1)] = int(m.group(2))
            if ".end_amdhsa_kernel" in l:
                name = None
    return meta


def pick(meta, tag):
    hits = [v for k, v in meta.items() if tag in k]
    assert hits, f"no kernel matching {tag}"
    return hits


def test_cull_tile_register_budgets(tmp_path):
    meta = metadata("cull_kernels.hip", tmp_path)
    # k_cull_tile<F = 1, 4 waves, 8 chunks, GRP, lane-parallel verdict by wave 0, no slots>: GRP 8 = the latency variant (headline camera),
    # GRP 4 = the streaming variant (roofline legs)
    # (round 6: the latency form keeps every chunk's cell-record offset + class in a register next to its eight chunks' spheres and ids: 68 VGPRs = 7 waves
    # per SIMD instead of 8. It runs where few tiles survive the tile-level test - the launch is the latency of the survivors, not their number:
    # 10.8 vs 11.0 us on the headline camera against round 5's 60-VGPR form, profiles/r06/cull1_ab_restructured.txt)
    # (the latency form is at 7 blocks per CU by its 68 VGPRs: up to 96 SGPRs cost it nothing; the streaming form - 8 blocks - must stay at <= 80,
    # with 8-byte cell keys (FORM 2) and with 16-byte ones (FORM 4))
    for tag, max_vgpr, max_sgpr in (("k_cull_tileILi1ELi4ELi8ELi8ELi1ELi0E", 72, 96), ("k_cull_tileILi1ELi4ELi8ELi4ELi2ELi0E", 48, 80), ("k_cull_tileILi1ELi4ELi8ELi4ELi4ELi0E", 48, 80)):
        for k in pick(meta, tag):
            assert k["private_segment_fixed_size"] == 0, (tag, k)
            assert k["next_free_sgpr"] <= max_sgpr, f"{tag}: {k['next_free_sgpr']} SGPRs - 8 resident blocks per CU need <= 80, 7 <= 96"
            assert k["next_free_vgpr"] <= max_vgpr, f"{tag}: {k['next_free_vgpr']} VGPRs"
            assert k["group_segment_fixed_size"] <= 8300, (tag, k)  # + <= 7.7 KiB of dynamic cell records: 8 blocks fit 160 KiB
    for k in pick(meta, "k_cull_tile"):  # every instantiation, the multi-frustum ones included: no spills
        assert k["private_segment_fixed_size"] == 0, k
    for k in pick(meta, "k_cull_pack") + pick(meta, "k_cull_dynamic") + pick(meta, "k_apply_patches"):
        assert k["private_segment_fixed_size"] == 0, k
    # the several-frusta kernels (F = 0): <= 96 VGPRs = 5 waves per SIMD, what the LDS of the cell records admits anyway; and the pre-test's
    # MFMA in its VGPR form - with accumulators in AGPRs every plane distance going in and every result coming out costs a v_accvgpr move
    # (504 of them per kernel when the register budget was left at 512: round 5)
    for k in pick(meta, "k_cull_tileILi0E"):
        assert k["next_free_vgpr"] <= 96, k
    text = (tmp_path / "cull_kernels.s").read_text()
    bodies = re.findall(r"^(_ZN3lmx\S*k_cull_tileILi0E\S*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
    assert len(bodies) == 4, [b[0] for b in bodies]
    for name, body in bodies:
        assert "v_mfma_f32_32x32x16_bf16" in body, name
        assert "v_accvgpr" not in body, name


def test_skin_and_pose_register_budgets(tmp_path):
    meta = metadata("skin_kernels.hip", tmp_path)
    for k in pick(meta, "k_skin_shared"):  # 1024-thread blocks: 4 waves per SIMD <=> 128 VGPRs; two palette buffers of 48 KiB
        assert k["private_segment_fixed_size"] == 0 and k["next_free_vgpr"] <= 128 and k["group_segment_fixed_size"] <= 98304, k
    for k in pick(meta, "k_skin_vertices"):  # 512-thread blocks, 3 per CU
        assert k["private_segment_fixed_size"] == 0 and k["next_free_vgpr"] <= 80, k
    for k in pick(meta, "k_pose_palette"):  # one-wave blocks, 19-20 per CU: <= 96 VGPRs (5 waves per SIMD), <= 8.5 KiB of LDS
        assert k["private_segment_fixed_size"] == 0 and k["next_free_vgpr"] <= 96 and k["group_segment_fixed_size"] <= 8704, k


def test_keys_and_xform_do_not_spill(tmp_path):
    for source, tags in (("keys_kernels.hip", ["k_keys_mesh", "k_keys_scatter", "k_keys_reduce_rows"]), ("xform_kernels.hip", ["k_xform_level", "k_sphere_refresh"])):
        meta = metadata(source, tmp_path)
        for tag in tags:
            for k in pick(meta, tag):
                assert k["private_segment_fixed_size"] == 0, (tag, k)


def test_round4_kernels_keep_their_residency(tmp_path):
    """The budgets round 4's measurements rest on (profiles/r04): the several-frusta cull kernel is held by LDS to five 4-wave blocks per CU
    (24-byte cell records + class words) - its registers must not be what cuts that further (<= 96 VGPRs) and 24 more live registers cost it
    12 % when tried (shared dot products); k_keys_mesh runs two 8-wave blocks per CU: <= 128 VGPRs, <= 80 KiB of LDS, and spills cost it more
    than waves; k_skin_multi's LDS is sized by the launch (dynamic), its FUSED / EXACT forms stay within the 6-waves-per-SIMD budget of
    their launch bounds without scratch; k_xform_subtree: <= 128 VGPRs, five 27-KiB blocks per CU."""
    cull = metadata("cull_kernels.hip", tmp_path)
    for k in pick(cull, "k_cull_tileILi0E"):
        assert k["private_segment_fixed_size"] == 0 and k["next_free_vgpr"] <= 96, k
        assert k["group_segment_fixed_size"] <= 1024, k  # static LDS only (normals, verdicts): the cell records are the launch's dynamic LDS
    keys = metadata("keys_kernels.hip", tmp_path)
    for k in pick(keys, "k_keys_mesh"):
        assert k["private_segment_fixed_size"] == 0 and k["next_free_vgpr"] <= 128 and k["group_segment_fixed_size"] <= 80 * 1024, k
    skin = metadata("skin_kernels.hip", tmp_path)
    for tag in ("k_skin_multiILi2ELi0E", "k_skin_multiILi2ELi1E", "k_skin_multiILi1ELi0E"):
        for k in pick(skin, tag):
            assert k["private_segment_fixed_size"] == 0 and k["next_free_vgpr"] <= 80, (tag, k)
            assert k["group_segment_fixed_size"] == 0, (tag, k)  # the palette staging is dynamic LDS, sized by the bones the launch's meshes reference
    xform = metadata("xform_kernels.hip", tmp_path)
    for k in pick(xform, "k_xform_subtree"):
        assert k["private_segment_fixed_size"] == 0 and k["next_free_vgpr"] <= 128 and k["group_segment_fixed_size"] <= 28 * 1024, k
