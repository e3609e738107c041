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


def test_host_transform_utilities_match_oracle(api, oracle_port):
    """lmx_transform_compose / lmx_transform_compute_local (Transform::compose / computeLocal, core/math.cpp:801-816)."""
    from lumixengine_amd import scenes

    rng = np.random.default_rng(9)
    a = scenes.random_transforms(rng, 500, 1.0e6)
    b = scenes.random_transforms(rng, 500, 50.0)
    assert H.transforms_bits_equal(api.transform_compose(a, b), oracle_port.compose(a, b))
    assert H.transforms_bits_equal(api.transform_compute_local(a, b), oracle_port.compute_local(a, b))
