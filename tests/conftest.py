"""pytest configuration: `gpu` marker + shared fixtures (CPU oracle libraries, emulation shim)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption(
        "--hostsim", action="store", nargs="?", const="plain", default=None, metavar="SANITIZERS",
        help="run the `-m gpu` tests against tests/hostsim: the kernel and C-ABI SOURCES compiled for the CPU, every lane a fiber "
             "(test infrastructure for the GPU-less container; optional value: -fsanitize list, e.g. address,undefined)")


def hostsim_active() -> bool:
    """True when the tests run against the simulated device (wall-clock assertions make no sense there)."""
    return os.environ.get("LMX_HOSTSIM") == "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with `-m gpu` on the GPU box")
    mode = config.getoption("--hostsim")
    if mode:
        from tests.hostsim import build as hostsim_build

        lib = hostsim_build.build(sanitize="" if mode == "plain" else mode)
        lib_dir = os.path.dirname(lib)
        alias = os.path.join(lib_dir, "liblumix_mi355.so")  # the C++ harnesses link -llumix_mi355: they find this one first
        if not os.path.islink(alias):
            try:
                os.symlink(os.path.basename(lib), alias)
            except FileExistsError:
                pass
        os.environ["LMX_LIB_PATH"] = lib
        os.environ["LMX_HOSTSIM"] = "1"
        os.environ["LD_LIBRARY_PATH"] = lib_dir + os.pathsep + os.environ.get("LD_LIBRARY_PATH", "")
        # the exchange's collective: the shared-memory stand-in for RCCL, built against the simulated device
        os.environ["LMX_RCCL_LIBRARY"] = hostsim_build.build_loopback(sanitize="" if mode == "plain" else mode)


@pytest.fixture(scope="session")
def oracle_port():
    from oracle import pyoracle

    if not os.path.exists(pyoracle.ORACLE_SO):
        pyoracle.build()
    return pyoracle.Oracle("port")


@pytest.fixture(scope="session")
def oracle_ref():
    """The reference's own object code (oracle/_ref). Built here when /root/reference exists; prebuilt on the GPU box."""
    from oracle import pyoracle

    if not pyoracle.have_reference():
        try:
            pyoracle.build()
        except Exception:
            pass
    if not pyoracle.have_reference():
        pytest.skip("oracle/_ref/liblmx_ref.so not available (no reference tree and no prebuilt copy)")
    return pyoracle.Oracle("reference")


@pytest.fixture(params=["port", "reference"])
def live_oracle(request):
    """Both CPU checkers in turn: the plain-C restatement and the reference's own code (skipped where oracle/_ref is unavailable)."""
    return request.getfixturevalue("oracle_port" if request.param == "port" else "oracle_ref")


@pytest.fixture(scope="session")
def emul_lib():
    """Host emulation of the kernels' logic (tests/emul/emul_kernels.cpp), compiled with g++ -ffp-contract=off."""
    import ctypes

    src = os.path.join(ROOT, "tests", "emul", "emul_kernels.cpp")
    out_dir = os.path.join(ROOT, "tests", "_build")
    out = os.path.join(out_dir, "libemul.so")
    deps = [src] + [os.path.join(ROOT, "lumixengine_amd", "csrc", h) for h in ("lmx_math.h", "lmx_cull_layout.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(out_dir, exist_ok=True)
        subprocess.run(
            ["g++", "-std=c++17", "-O2", "-msse2", "-mfpmath=sse", "-ffp-contract=off", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
             "-I" + os.path.join(ROOT, "lumixengine_amd", "csrc"), src, "-o", out],
            check=True,
        )
    return ctypes.CDLL(out)


@pytest.fixture(scope="session")
def gpu_ctx():
    """A lumixengine_amd Context on cuda:0. Fails (does not skip) when the HIP extension or the GPU is missing."""
    from lumixengine_amd import api

    ctx = api.Context(0)
    yield ctx
    ctx.close()
