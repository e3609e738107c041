#!/usr/bin/env python
"""Build variants of liblumix_mi355.so for A/B sweeps on the GPU box: one kernel source re-compiled with extra -D flags, linked with the
other objects of the regular build.

    python tools/build_variant.py <name> <source.hip>[,<source2.hip>...] "<extra hipcc flags>"     ->  tools/_build/variants/<name>/liblumix_mi355.so

Use with LMX_LIB_PATH=<that file> (lumixengine_amd/api.py) - e.g. tools/run_workload.py, bench.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lumixengine_amd import build as B  # noqa: E402


def main():
    name, sources, extra = sys.argv[1], sys.argv[2].split(","), sys.argv[3].split()
    B.build()  # the regular objects
    out_dir = os.path.join(ROOT, "tools", "_build", "variants", name)
    os.makedirs(out_dir, exist_ok=True)
    own = {}
    for source in sources:
        own[source] = os.path.join(out_dir, os.path.splitext(source)[0] + ".o")
        subprocess.run([B.hipcc()] + B.FLAGS + extra + ["-x", "hip", "-c", os.path.join(B.CSRC, source), "-o", own[source]], check=True)
    objs = [own.get(s, os.path.join(B.OBJ, os.path.splitext(s)[0] + ".o")) for s in B.SOURCES]
    lib = os.path.join(out_dir, "liblumix_mi355.so")
    subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"], check=True)
    print(lib)


if __name__ == "__main__":
    main()
