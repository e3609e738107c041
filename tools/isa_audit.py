"""ISA audit (no GPU needed): for every kernel of the library, registers / scratch and - per loop - the vmcnt waits next to the
loop's loads and stores. A loop that stores AND waits with vmcnt(0) drains its stores every iteration (vector loads and stores retire
through one in-order counter on gfx9-family parts; the compiler only emits an exact count when no load / store sits under a branch).
    python tools/isa_audit.py [file.hip ...]        (default: every csrc/*_kernels.hip)"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lumixengine_amd", "csrc")


def demangle(names):
    filt = shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    if not os.path.exists(filt):
        return names
    r = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.split("\n")[: len(names)] if r.returncode == 0 else names


def audit(src):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, src]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-2000:])
            return
        lines = open(out).read().splitlines()
    meta = {}
    name = None
    for l in lines:
        m = re.match(r"\s+\.name:\s+(\S+)", l)
        if m:
            name = m.group(1)
        for key in ("vgpr_count", "sgpr_count", "private_segment_fixed_size"):
            m = re.match(r"\s+\." + key + r":\s+(\d+)", l)
            if m and name:
                meta.setdefault(name, {})[key] = int(m.group(1))
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if l.startswith("_ZN") and ":" in l and "@" in l]
    names = demangle([n for _, n in starts])
    for (i, mangled), pretty in zip(starts, names):
        end = next(j for j in range(i, len(lines)) if "s_endpgm" in lines[j])
        body = lines[i : end + 1]
        loops, current = {}, None
        for l in body:
            m = re.match(r"^(\.LBB\d+_\d+):.*=>This (Inner )?Loop Header", l)
            if m:
                current = m.group(1)[2:]
                loops.setdefault(current, [])
                continue
            if l.startswith(".LBB") or l.startswith("; %bb."):
                m = re.search(r"in Loop: Header=(BB\d+_\d+)", l)
                current = m.group(1) if m else None
                if current is not None:
                    loops.setdefault(current, [])
                continue
            if current is not None and l.startswith("\t") and not l.lstrip().startswith(";"):
                loops[current].append(l.strip())
        md = meta.get(mangled, {})
        print(f"{pretty[:110]}\n    vgpr {md.get('vgpr_count')} sgpr {md.get('sgpr_count')} scratch {md.get('private_segment_fixed_size')} B, {len(body)} lines, {len(loops)} loops")
        for h, ins in loops.items():
            n_st = sum(bool(re.match(r"(global|buffer|flat)_store", x)) for x in ins)
            n_ld = sum(bool(re.match(r"(global|buffer|flat)_load", x)) for x in ins)
            n_at = sum(bool(re.match(r"(global|buffer|flat)_atomic", x)) for x in ins)
            waits = [int(m.group(1)) for x in ins for m in [re.search(r"vmcnt\((\d+)\)", x)] if m]
            if n_st + n_ld + n_at == 0:
                continue
            flag = "  <-- drains its stores" if n_st and waits and min(waits) == 0 else ""
            print(f"      loop {h}: {n_ld} loads, {n_st} stores, {n_at} atomics, vmcnt waits {waits}{flag}")


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*_kernels.hip")))
    for f in files:
        print(f"==== {os.path.relpath(f, ROOT)}")
        audit(f)
