"""C-ABI surface checks that need no GPU: the library loads, exports exactly what include/lumix_mi355.h declares,
fails loudly without a device, and its host-side frustum mirror is bit-identical to the reference's construction."""
import os
import re

import numpy as np
import pytest

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    from lumixengine_amd import api as a
    from lumixengine_amd import build

    if not os.path.exists(a.LIB_PATH):
        build.build()
    return a


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lumix_mi355.h")).read()
    return sorted(set(re.findall(r"LMX_API\s+[\w\s\*]+?\b(lmx_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(api):
    lib = api.load_library()
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in lumix_mi355.h but not exported"
        assert n in api.SYMBOLS, f"{n} has no ctypes prototype in lumixengine_amd/api.py"
    assert sorted(api.SYMBOLS) == names


def test_no_silent_cpu_fallback(api):
    """Without a gfx950 device the product refuses to create a context instead of computing on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; covered by the gpu tests")
    with pytest.raises(api.LumixError) as e:
        api.Context(0)
    assert e.value.code == 2  # LMX_ERR_NO_DEVICE


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "lumixengine_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "lmx_oracle" not in text and "liblmx_ref" not in text, f"{f} references the oracle"
                # ... nor the simulated device of tests/hostsim (the kernels carry two compile-time hooks for it, nothing loads it)
                assert "liblumix_hostsim" not in text and "hostsim_runtime" not in text, f"{f} references the simulated device's library"


def test_host_frustum_mirror_matches_oracle(api, oracle_port):
    assert H.bits_equal(H.frusta(api), H.frusta(oracle_port))
    assert H.bits_equal(H.cascade_frusta(api), H.cascade_frusta(oracle_port))
    rng = np.random.default_rng(3)
    for _ in range(25):
        pos = rng.uniform(-1e6, 1e6, 3)
        d = rng.normal(size=3).astype(np.float32)
        up = np.cross(d, rng.normal(size=3)).astype(np.float32)
        args = (pos, d, up, float(rng.uniform(0.2, 2.0)), float(rng.uniform(0.5, 2.5)), 0.1, float(rng.uniform(10, 1e4)))
        assert H.bits_equal(api.frustum_perspective(*args), oracle_port.frustum_perspective(*args))
        args = (pos, d, up, float(rng.uniform(1, 500)), float(rng.uniform(1, 500)), 0.0, float(rng.uniform(10, 1e4)))
        assert H.bits_equal(api.frustum_ortho(*args), oracle_port.frustum_ortho(*args))


def test_struct_layouts(api):
    assert api.SHIFTED_FRUSTUM.itemsize == 256 and api.TRANSFORM.itemsize == 56
    assert api.LOCAL_RIGID.itemsize == 28 and api.MATRIX.itemsize == 64 and api.SKIN.itemsize == 24


def test_struct_layouts_match_the_c_header(api, tmp_path):
    """Every POD the Python binding mirrors has the size and field offsets the C compiler gives include/lmx_types.h / lumix_mi355.h."""
    import subprocess

    structs = {
        "LmxShiftedFrustum": (api.SHIFTED_FRUSTUM, ["xs", "points", "origin"]),
        "LmxTransform": (api.TRANSFORM, ["pos", "rot", "scale"]),
        "LmxLocalRigidTransform": (api.LOCAL_RIGID, ["pos", "rot"]),
        "LmxSkin": (api.SKIN, ["weights", "indices"]),
        "LmxKeysModel": (api.KEYS_MODEL, ["lod_distances", "lod_indices", "first_mesh", "mesh_count"]),
        "LmxMeshMaterial": (api.MESH_MATERIAL, ["sort_key", "layer"]),
        "LmxKeysView": (api.KEYS_VIEW, ["camera_pos", "lod_ref_point", "lod_multiplier", "time_delta", "frame_number", "is_shadow", "layer_to_bucket", "bucket_depth_sorted"]),
        "LmxKeysCounts": (api.KEYS_COUNTS, ["pairs", "instanced", "groups", "poses", "dirty", "overflow"]),
        "LmxAnimConstTranslation": (api.ANIM_CONST_TRANSLATION, ["value", "bone_index"]),
        "LmxAnimTranslationTrack": (api.ANIM_TRANSLATION_TRACK, ["min", "to_range", "offset_bits", "bone_index", "bitsizes"]),
        "LmxAnimConstRotation": (api.ANIM_CONST_ROTATION, ["value", "bone_index"]),
        "LmxAnimRotationTrack": (api.ANIM_ROTATION_TRACK, ["min", "to_range", "offset_bits", "bone_index", "bitsizes", "skipped_channel"]),
        "LmxWorldBlobInfo": (api.WORLD_BLOB_INFO, ["version", "flags", "n_modules", "uncompressed_size", "compressed_size", "n_entities", "max_entity_index", "n_names", "n_hierarchy"]),
        "LmxRenderBlobInfo": (api.RENDER_BLOB_INFO, list(api.RENDER_BLOB_INFO.names)),
        "LmxBlobBoneAttachment": (api.BLOB_BONE_ATTACHMENT, ["bone_name_hash", "entity", "parent_entity"]),
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "lumix_mi355.h"', "int main(void) {"]
    for name, (_, fields) in structs.items():
        lines.append(f'printf("{name} %zu", sizeof({name}));')
        for f in fields:
            lines.append(f'printf(" %zu", offsetof({name}, {f}));')
        lines.append('printf("\\n");')
    lines.append(f'printf("LmxAnimation %zu %zu %zu\\n", sizeof(LmxAnimation), offsetof(LmxAnimation, const_translations), offsetof(LmxAnimation, root_pose_rotations));')
    lines.append("return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run(["gcc", "-std=c11", "-I", inc, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    got = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in out}
    for name, (dtype, fields) in structs.items():
        want = [dtype.itemsize] + [dtype.fields[f][1] for f in fields]
        assert got[name] == want, f"{name}: C {got[name]} vs numpy {want}"
    import ctypes as C

    assert got["LmxAnimation"] == [C.sizeof(api.LmxAnimation), api.LmxAnimation.const_translations.offset, api.LmxAnimation.root_pose_rotations.offset]


def test_integration_doc_covers_every_entry_point():
    """INTEGRATION.md names the reference interface each C-ABI entry point replaces: no exported symbol may be missing from it."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "lumix_mi355.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    names = re.findall(r"LMX_API\s+[\w\s\*]+?\b(lmx_\w+)\s*\(", header)
    assert len(names) > 70
    missing = []
    for n in names:
        if n in doc:
            continue
        parts = n.split("_")
        ok = False
        for k in range(len(parts) - 1, 1, -1):  # `lmx_cull_add/remove/set` and `lmx_keys_read_*` style groupings
            pre, post = "_".join(parts[:k]), "_".join(parts[k:])
            ok = ok or pre + "_*" in doc or re.search(re.escape(pre) + r"[\w/]*?/" + re.escape(post) + r"\b", doc) is not None
            ok = ok or re.search(re.escape(pre) + r"_[\w/]*\b" + re.escape(post), doc) is not None
        if not ok:
            missing.append(n)
    assert not missing, missing


def test_host_transform_utilities_match_oracle(api, oracle_port):
    """lmx_transform_compose / lmx_transform_compute_local (Transform::compose / computeLocal, core/math.cpp:801-816)."""
    from lumixengine_amd import scenes

    rng = np.random.default_rng(9)
    a = scenes.random_transforms(rng, 500, 1.0e6)
    b = scenes.random_transforms(rng, 500, 50.0)
    assert H.transforms_bits_equal(api.transform_compose(a, b), oracle_port.compose(a, b))
    assert H.transforms_bits_equal(api.transform_compute_local(a, b), oracle_port.compute_local(a, b))
