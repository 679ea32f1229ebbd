#!/usr/bin/env python
"""Command-line form of warp_rnnt_amd/_isa_check.py (the static checks `_build.build()` runs on every build and that fail the
build on a violation -- read that module's header first): the lattice kernels' in-place LDS reloads, and the wait-state
hazards the compiler does not pad around inline assembly.

    python tools/check_inplace_reloads.py [file.s]        exit status 1 on a violation

Without an argument it compiles csrc/lattice_wd.hip for gfx950 with -save-temps (hipcc cross-compiles, no GPU needed) and
checks the ISA that comes out -- the same thing the build does with the object it ships."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from warp_rnnt_amd._isa_check import check, check_hazards  # noqa: E402

CSRC = os.path.join(ROOT, "warp_rnnt_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-fno-slp-vectorize"]


def compile_to_asm(src, extra=()):
    tmp = tempfile.mkdtemp(prefix="rnnt_isa_")
    hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc] + FLAGS + list(extra) + ["-save-temps", "-c", src, "-o", os.path.join(tmp, "x.o")],
                          cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in os.listdir(tmp):
        if f.endswith("gfx950.s"):
            return os.path.join(tmp, f)
    raise RuntimeError("no device assembly in " + tmp)


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else compile_to_asm(os.path.join(CSRC, "lattice_wd.hip"))
    kernels, reloads, bad = check(path)
    print(f"{path}: {kernels} lattice kernels, {reloads} in-place reloads checked, {len(bad)} violation(s)")
    for fn, ln, text, regs in bad[:40]:
        print(f"  line {ln}: `{text}` touches v{regs} while its reload is in flight   [{fn[:60]}]")
    nk, ni, found = check_hazards(path)
    fatal = [f for f in found if f[4]]
    print(f"{path}: {nk} kernels, {ni} instructions walked for wait-state hazards, {len(fatal)} around inline assembly, "
          f"{len(found) - len(fatal)} between compiler instructions")
    for fn, ln, text, what, asm_side in found[:40]:
        print(f"  line {ln}: `{text}`: {what}{'' if asm_side else '   (compiler only)'}   [{fn[:60]}]")
    return 1 if bad or fatal or (reloads == 0 and len(sys.argv) < 2) else 0


if __name__ == "__main__":
    sys.exit(main())
