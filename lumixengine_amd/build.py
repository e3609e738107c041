"""Builds lumixengine_amd/liblumix_mi355.so for gfx950 with hipcc (in-tree, so the .so travels with the repo snapshot).

    python -m lumixengine_amd.build [--force]

Every translation unit is compiled with -ffp-contract=off: visibility and world transforms must round exactly like the
reference's FMA-free CPU path.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "build")
LIB = os.path.join(PKG, "liblumix_mi355.so")
SOURCES = ["cull_kernels.hip", "xform_kernels.hip", "skin_kernels.hip", "lmx_capi_ctx.hip", "lmx_capi_cull.hip", "lmx_capi_world.hip",
           "lmx_capi_skin.hip", "lmx_capi_exchange.hip", "keys_kernels.hip", "lmx_capi_keys.hip", "anim_kernels.hip", "lmx_capi_anim.hip", "lmx_frustum.cpp", "lmx_world_blob.cpp"]
HEADERS = [os.path.join(CSRC, "lmx_math.h"), os.path.join(CSRC, "lmx_kernels.h"), os.path.join(CSRC, "lmx_cull_layout.h"), os.path.join(CSRC, "lmx_context.h"), os.path.join(ROOT, "include", "lumix_mi355.h"),
           os.path.join(ROOT, "include", "lmx_types.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + os.environ.get("LMX_HIPCC_EXTRA", "").split()  # experiments: -DLMX_...=n


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the gfx950 extension cannot be built")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str) -> str:
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    if _stale(obj, [path] + HEADERS):
        cmd = [hipcc()] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
        if os.path.exists(LIB):
            os.remove(LIB)
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if _stale(LIB, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
