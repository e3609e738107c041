"""Builds tests/_build/hostsim/liblumix_hostsim.so: the product's kernel and C-ABI sources compiled for the CPU against the
simulated device of tests/hostsim (TEST INFRASTRUCTURE; see include/hip/hip_runtime.h there).

    python -m tests.hostsim.build [--force] [--sanitize address,undefined | thread]

The library exports the same C ABI as lumixengine_amd/liblumix_mi355.so. It is only ever loaded by tests (conftest's
`--hostsim` option); the product never looks for it.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "lumixengine_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _sources():
    sys.path.insert(0, ROOT)
    from lumixengine_amd import build as product

    return list(product.SOURCES), list(product.HEADERS)


def _extra():
    """LMX_HOSTSIM_EXTRA="-DLMX_CULL_HDR_AHEAD=1 ...": experiment knobs of the kernels (the twin of LMX_HIPCC_EXTRA of the gfx950 build);
    every set of flags gets its own build directory."""
    return os.environ.get("LMX_HOSTSIM_EXTRA", "").split()


def lib_path(sanitize: str = "") -> str:
    tag = "hostsim" + ("_" + sanitize.replace(",", "_") if sanitize else "")
    extra = _extra()
    if extra:
        import hashlib

        tag += "_x" + hashlib.sha1(" ".join(extra).encode()).hexdigest()[:8]
    return os.path.join(ROOT, "tests", "_build", tag, "liblumix_hostsim.so")


def build(force: bool = False, sanitize: str = "", opt: str = "-O2") -> str:
    if not os.path.exists(CLANG):
        raise RuntimeError("amdclang++ (used here as a plain x86 compiler) not found at " + CLANG)
    sources, headers = _sources()
    out = lib_path(sanitize)
    obj_dir = os.path.dirname(out)
    os.makedirs(obj_dir, exist_ok=True)
    inc = os.path.join(HERE, "include")
    headers += [os.path.join(inc, "hip", "hip_runtime.h"), os.path.join(inc, "hip", "hip_ext.h"), os.path.join(inc, "hipcub", "hipcub.hpp"), os.path.abspath(__file__)]
    flags = ["-std=c++17", opt, "-g1", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-pthread", "-Wall", "-Wno-unused-function", "-Wno-unknown-attributes",
             "-Wno-ignored-attributes", "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-I" + inc, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + _extra()
    traffic = sanitize == "traffic"  # not a sanitizer: ThreadSanitizer's instrumentation linked against traffic_runtime.cpp (a traffic model)
    if traffic:
        flags += ["-fsanitize=thread", "-mllvm", "-tsan-instrument-func-entry-exit=0", "-mllvm", "-tsan-instrument-atomics=0", "-mllvm", "-tsan-compound-read-before-write=1"]
    elif sanitize:
        flags += ["-fsanitize=" + sanitize, "-fno-omit-frame-pointer"]
        if "thread" in sanitize:
            flags += ["-mllvm", "-tsan-instrument-func-entry-exit=0"]  # lanes share their wave's fiber: no per-lane shadow call stacks
        if "undefined" in sanitize:
            flags += ["-fno-sanitize=function"]  # the RCCL entry points come out of dlsym: their parameter structs are declared on both sides

    def stale(target, deps):
        return force or not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)

    def compile_one(path):
        obj = os.path.join(obj_dir, os.path.splitext(os.path.basename(path))[0] + ".o")
        if stale(obj, [path] + headers):
            use = list(flags)
            if "thread" in sanitize and os.path.basename(path) == "hostsim_runtime.cpp":  # the scheduler itself is not the subject
                use = [f for f in use if not f.startswith("-fsanitize=")] + ["-DHOSTSIM_WITH_TSAN=1"]
            if traffic and os.path.basename(path) in ("hostsim_runtime.cpp", "traffic_runtime.cpp"):
                use = [f for f in use if not f.startswith("-fsanitize=") and not f.startswith("-tsan-") and f != "-mllvm"] + ["-DHOSTSIM_TRAFFIC=1"]
            r = subprocess.run([CLANG] + use + ["-x", "c++", "-c", path, "-o", obj], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hostsim: compiling {os.path.basename(path)} failed:\n{r.stdout}\n{r.stderr}")
        return obj

    paths = [os.path.join(CSRC, s) for s in sources] + [os.path.join(HERE, "hostsim_runtime.cpp")] + ([os.path.join(HERE, "traffic_runtime.cpp")] if traffic else [])
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(compile_one, paths))
    if stale(out, objs):
        cmd = [CLANG, "-shared", "-fPIC", "-pthread", "-o", out] + objs + ["-ldl"]
        if sanitize and not traffic:
            cmd += ["-fsanitize=" + sanitize, "-shared-libsan"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hostsim: link failed:\n{r.stdout}\n{r.stderr}")
    return out


def build_loopback(sanitize: str = "") -> str:
    """tests/cpp/loopback_rccl.cpp (the shared-memory stand-in for the five RCCL entry points) against the simulated device."""
    lib = build(sanitize=sanitize)
    out = os.path.join(os.path.dirname(lib), "libloopback_rccl.so")
    src = os.path.join(ROOT, "tests", "cpp", "loopback_rccl.cpp")
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in (src, lib)):
        cmd = [CLANG, "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + os.path.join(HERE, "include"), "-x", "c++", src, "-o", out, "-L" + os.path.dirname(lib),
               "-llumix_hostsim", "-Wl,-rpath," + os.path.dirname(lib), "-lrt", "-pthread"]
        if sanitize:
            cmd += ["-fsanitize=" + sanitize, "-shared-libsan"]
        subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    san = ""
    if "--sanitize" in sys.argv:
        san = sys.argv[sys.argv.index("--sanitize") + 1]
    print(build(force="--force" in sys.argv, sanitize=san))
