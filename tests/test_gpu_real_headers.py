"""The drop-in boundary EXECUTED against the reference's real headers.

tests/cpp/real_header_harness.cpp compiles lumixengine_amd/host/ (GpuCullingSystem, WorldSync, mi355_plugin.cpp with its ISystem /
IModule pair) with -DLMX_WITH_LUMIX_HEADERS against /root/reference/src and links it with the reference object code of oracle/_ref
(a real Lumix::World, PageAllocator, CullingSystemImpl). It is built where the reference tree exists (`make -C oracle harness`, also
part of __graft_entry__.build()); the binary travels to the GPU box inside oracle/_ref/. On the GPU it drives

  1. the CullingSystem vtable (renderer/culling_system.h:58-77): 40 k adds + 6000 interleaved add / remove / set / setPosition /
     setRadius / getRadius / isAdded calls + culls with and without a type filter + cullMany, every result compared with the reference's
     CullingSystemImpl running next to it;
  2. the plugin the way the engine loads it (createPlugin_mi355 -> createModules(world) -> init -> update per frame) in a real World
     whose "renderer" owns createGpuCullingSystem(allocator, pages, world): staged writes through the module, direct World writes by
     engine code, `transformed` delegates, moved-entity hand-back - World::getTransforms() bit-identical to, and every visible set equal
     to, a second real World + CullingSystemImpl that take the same writes through World::setTransform / setLocalTransform.

`--two-contexts` re-creates round 2's wiring (the culling system on a context of its own): binding the model instances must then
fail loudly. That run failing is what makes the default run's pass meaningful."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "real_header_harness")
REF = "/root/reference"


def _build():
    if os.path.isdir(os.path.join(REF, "src")):
        from lumixengine_amd import build

        build.build()
        subprocess.run(["make", "-s", "-f", os.path.join(ROOT, "oracle", "Makefile"), "ref", "harness"], check=True)
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/real_header_harness not available (no reference tree here and no prebuilt copy)")


def test_harness_builds_against_real_headers_and_needs_the_gpu():
    """CPU check: the harness compiles against the real headers, links the reference object code + the product library, and refuses to
    run without a device (no CPU fallback anywhere in the product path)."""
    import torch

    _build()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the -m gpu tests run the harness for real")
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "no HIP device" in r.stderr, r.stdout + r.stderr


@pytest.mark.gpu
def test_real_header_harness_one_context_per_world():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "real-header harness OK" in r.stdout
    assert "identical to CullingSystemImpl" in r.stdout and "transforms bit-identical" in r.stdout
    assert "asynchronous compaction of" in r.stdout  # the worker re-sorted and the sets traded places under the update stream
    assert "12 concurrent callers x 40 culls on 8 result slots: every list identical" in r.stdout  # more callers than result slots never alias


def test_real_header_harness_on_the_simulated_device():
    """CPU check of the same binary: the simulated device's library (tests/hostsim: the product's kernel + C-ABI sources compiled for the
    CPU) is put in front of liblumix_mi355.so, so the adapter's slot reservation, the 12 concurrent callers and the module's hand-back run
    here too - against the reference's CullingSystemImpl / World as on the GPU."""
    import torch

    _build()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the -m gpu tests run the harness on it")
    from tests.hostsim import build as hostsim_build

    lib = hostsim_build.build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=900, env=dict(os.environ, LD_PRELOAD=lib))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "real-header harness OK" in r.stdout
    assert "12 concurrent callers x 40 culls on 8 result slots: every list identical" in r.stdout


@pytest.mark.gpu
def test_real_header_harness_detects_round2_wiring():
    """The culling system on a context of its own (round 2's defect): the module's binding fails, says why, and the harness exits 42."""
    _build()
    r = subprocess.run([EXE, "--two-contexts"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 42, r.stdout[-3000:] + r.stderr[-3000:]
    assert "is not in the culling system" in r.stdout + r.stderr
