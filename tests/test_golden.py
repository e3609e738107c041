"""The plain-C oracle against the committed golden fixtures (tests/golden/*.npz, generated from the reference's own
object code by tests/golden/make_golden.py). Runs anywhere: needs neither /root/reference nor a GPU."""
import os

import numpy as np

from lumixengine_amd import scenes
from tests import helpers as H

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_frusta(oracle_port):
    g = load("frusta.npz")
    assert [str(n) for n in g["names"]] == [c[0] for c in H.CAMERAS]
    assert H.bits_equal(H.frusta(oracle_port), g["frusta"])
    assert H.bits_equal(H.cascade_frusta(oracle_port), g["cascades"])


def _check_cull(oracle_port, g, sc):
    cs = oracle_port.culling_system()
    cs.add_bulk(sc["entity"], sc["type"], sc["pos"], sc["radius"])
    assert cs.cell_count() == int(g["n_cells"][0])
    frusta = g["frusta"]
    for f in range(len(frusta)):
        ids, types, _ = cs.cull(np.ascontiguousarray(frusta[f : f + 1]))
        got = H.sorted_by_type(ids, types)
        want = H.sorted_by_type(g[f"vis_ids_{f}"], g[f"vis_types_{f}"])
        H.assert_same_visible(got, want, f"frustum {f}")


def test_cull_edge_cases(oracle_port):
    g = load("cull_edge.npz")
    _check_cull(oracle_port, g, {k: g[k] for k in ("entity", "type", "pos", "radius")})
    # the fixture really contains the interesting outcomes
    assert sum(len(g[f"vis_ids_{f}"]) for f in range(len(g["frusta"]))) > 500


def test_cull_mixed_types_and_cascades(oracle_port):
    g = load("cull_mixed.npz")
    _check_cull(oracle_port, g, {k: g[k] for k in ("entity", "type", "pos", "radius")})


def test_cull_config1(oracle_port):
    """BASELINE config 1: 100 k static entities, CPU reference path, bit-exact visible list."""
    g = load("cull_config1.npz")
    sc = scenes.cull_scene(100_000, 3000.0, seed=1)
    assert np.array_equal(np.array([sc["pos"].sum(), np.abs(sc["pos"]).sum()]), g["pos_sum"]), "scene generator drifted; regenerate goldens"
    assert float(sc["radius"].astype(np.float64).sum()) == float(g["radius_sum"][0])
    _check_cull(oracle_port, g, sc)


def test_transforms(oracle_port):
    g = load("transforms.npz")
    assert H.transforms_bits_equal(oracle_port.compose(g["a"], g["b"]), g["compose"])
    assert H.transforms_bits_equal(oracle_port.compute_local(g["a"], g["b"]), g["compute_local"])
    parent = g["parent"]
    n = len(parent)
    w = oracle_port.world(n)
    roots = np.flatnonzero(parent < 0).astype(np.int32)
    kids = np.flatnonzero(parent >= 0).astype(np.int32)
    h = scenes.hierarchy_fans(6, 3, 4, seed=3)
    w.init_transforms(roots, h["local"][roots])
    w.set_parents(parent[kids], kids)
    w.set_local_transforms(kids, h["local"][kids])
    assert H.transforms_bits_equal(w.get_local_transforms()[kids], g["locals"][kids])  # stored locals exist for parented entities only
    assert H.transforms_bits_equal(w.get_transforms(), g["world0"])
    w.set_transforms(roots, g["new_root"])
    assert H.transforms_bits_equal(w.get_transforms(), g["world1"])


def test_pose_palette_skin(oracle_port):
    g = load("skin.npz")
    inv = oracle_port.invert_bind(g["bind"])
    assert H.bits_equal(inv, g["inv_bind"])
    apos, arot = oracle_port.pose_compute_absolute(g["rel_pos"], g["rel_rot"], g["parents"], int(g["first_nonroot"][0]))
    assert H.bits_equal(apos, g["abs_pos"]) and H.bits_equal(arot, g["abs_rot"])
    pal = oracle_port.skin_matrices(apos, arot, inv)
    assert H.bits_equal(pal, g["palette"])
    assert H.bits_equal(oracle_port.evaluate_skin(g["verts"], g["skin"], pal), g["skinned"])
    assert H.bits_equal(oracle_port.dual_quats(apos, arot, inv), g["dual_quats"])
