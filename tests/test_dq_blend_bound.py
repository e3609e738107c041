"""f3, CPU half: the plain-C restatement of the reference's dual-quaternion vertex blend
(oracle/lmx_oracle.c: orc_evaluate_dq_skin, from data/shaders/surface_base.hlsli:196-217 + common.hlsli:632-636) is held to an
extended-precision evaluation of the shader's expressions (tests/dq_exact.py says what is and is not pinnable for HLSL), on random
AND adversarial inputs. The `-m gpu` twin (tests/test_gpu_world_skin.py::test_skin_dqs_bounded_by_exact_evaluation) holds the HIP
kernel to the same bound."""
import numpy as np
import pytest

from lumixengine_amd import scenes
from tests import dq_exact as DQ


def _identity_inv(n):
    inv = np.zeros(n, scenes.LOCAL_RIGID)
    inv["rot"][:, 3] = 1.0
    return inv


def _palette(oracle, pos, rot):
    return oracle.dual_quats(pos[None], rot[None], oracle.invert_bind(_identity_inv(len(pos))))[0]


def test_oracle_dq_blend_within_bound_adversarial(oracle_port):
    pos, rot, verts, skin = DQ.adversarial_case()
    dq = _palette(oracle_port, pos, rot)
    got = oracle_port.evaluate_dq_skin(verts, skin, dq[None])[0]
    cands, bounds = DQ.dq_skin_candidates(verts, skin, dq)
    ok = DQ.within_bound(got, cands, bounds)
    assert ok.all(), f"{int((~ok).sum())} vertices outside the bound, worst ratio {DQ.worst_ratio(got, cands, bounds):.2f}, first {np.flatnonzero(~ok)[:8]}"
    assert len(cands) > 1, "the adversarial case must contain sign-ambiguous vertices"
    # the crafted vertices: antipodal copies of ONE rotation blend to exactly that bone's rigid transform (the sign fix works)
    rigid = DQ.dq_skin_candidates(verts[:2], np.array([((1, 0, 0, 0), (0, 0, 0, 0))] * 2, scenes.SKIN), dq)[0][0]
    assert np.allclose(np.asarray(got[:2], np.float64), np.asarray(rigid, np.float64), rtol=0, atol=3e-5)
    # the bound is not vacuous: 1e-3 off is outside it wherever the blend is well conditioned (L > 0.5)
    tight = np.asarray(bounds.min(axis=0), np.float64) < 1e-4
    assert tight.sum() > len(verts) // 2 and not DQ.within_bound(got + np.float32(1e-3), cands, bounds)[tight].any()


@pytest.mark.parametrize("seed", [6, 7])
def test_oracle_dq_blend_within_bound_random(oracle_port, seed):
    s = scenes.skeleton(64, seed=4)
    verts, skin = scenes.skinned_mesh(5000, 64, seed=seed)
    pos, rot = scenes.relative_poses(1, 64, seed=700 + seed)
    apos, arot = oracle_port.pose_compute_absolute(pos, rot, s["parents"], s["first_nonroot"])
    dq = oracle_port.dual_quats(apos, arot, oracle_port.invert_bind(s["bind"]))[0]
    got = oracle_port.evaluate_dq_skin(verts, skin, dq[None])[0]
    cands, bounds = DQ.dq_skin_candidates(verts, skin, dq)
    ok = DQ.within_bound(got, cands, bounds)
    assert ok.all(), f"worst ratio {DQ.worst_ratio(got, cands, bounds):.2f}"
    # how tight: the restatement sits well inside the bound on ordinary inputs
    assert DQ.worst_ratio(got, cands, bounds) < 0.5
