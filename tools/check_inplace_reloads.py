#!/usr/bin/env python
"""Static check of the in-place LDS reloads of the column-block lattice kernels (csrc/lattice_step.h, lattice_wd.hip).

The compute wave refills the registers of a block's (blank, label) pairs and boundary seeds with the NEXT block's values
while it is still working on the current block: `ds_read2st64_b64` / `ds_read_b32` in inline assembly, which the compiler
does not count.  The data lands some hundred cycles later; the only thing that makes the registers valid is the
`s_waitcnt lgkmcnt(0)` in front of the block's barrier.  Nothing may read or write those registers in between -- and
the one who could is the compiler (a register copy at a loop head, a spill, a reuse as a temporary), silently.

This script compiles lattice_wd.hip for gfx950 with -save-temps (hipcc cross-compiles, no GPU needed), walks the
generated ISA of every kernel in program order and reports any instruction that touches a register with a reload in
flight.  Conservative along straight-line code and fall-through edges; an unconditional branch ends a path.

    python tools/check_inplace_reloads.py [file.s]        exit status 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "warp_rnnt_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-fno-slp-vectorize"]

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
RELOAD = re.compile(r"^\s*(ds_read2st64_b64|ds_read_b32|ds_read_b64|ds_read_b128)\s+(v\d+|v\[\d+:\d+\])\s*,\s*(v\d+)")
WAIT = re.compile(r"^\s*s_waitcnt\b.*lgkmcnt\(0\)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def compile_to_asm(src, extra=()):
    tmp = tempfile.mkdtemp(prefix="rnnt_isa_")
    hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc] + FLAGS + list(extra) + ["-save-temps", "-c", src, "-o", os.path.join(tmp, "x.o")],
                          cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in os.listdir(tmp):
        if f.endswith("gfx950.s"):
            return os.path.join(tmp, f)
    raise RuntimeError("no device assembly in " + tmp)


def check(path):
    """Returns (kernels seen, in-place reloads seen, [violations])."""
    kernels, reloads, bad = 0, 0, []
    fn, in_asm, pending = None, False, {}
    for ln, line in enumerate(open(path), 1):
        s = line.split(";")[0].rstrip() if not line.lstrip().startswith(";;#") else line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(_Z\w+):", s)
        if m:
            fn, pending = m.group(1), {}
            kernels += "k_lattice" in fn
            continue
        if fn is None or "k_lattice" not in fn or not s.strip() or s.lstrip().startswith("."):
            continue
        if re.match(r"^\s*s_endpgm", s):
            fn = None
            continue
        if WAIT.match(s):
            pending = {}
            continue
        if re.match(r"^\s*s_branch\b", s):
            pending = {}
            continue
        r = RELOAD.match(s) if in_asm else None
        touched = regs_of(s)
        if r:
            dst, addr = regs_of(r.group(2)), regs_of(r.group(3))
            hit = addr & set(pending)
            if hit:
                bad.append((fn, ln, s.strip(), sorted(hit)))
            reloads += 1
            for v in dst:
                pending[v] = ln
            continue
        hit = touched & set(pending)
        if hit:
            bad.append((fn, ln, s.strip(), sorted(hit)))
            for v in hit:          # report a register once per reload
                pending.pop(v, None)
    return kernels, reloads, bad


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else compile_to_asm(os.path.join(CSRC, "lattice_wd.hip"))
    kernels, reloads, bad = check(path)
    print(f"{path}: {kernels} lattice kernels, {reloads} in-place reloads checked, {len(bad)} violation(s)")
    for fn, ln, text, regs in bad[:40]:
        print(f"  line {ln}: `{text}` touches v{regs} while its reload is in flight   [{fn[:60]}]")
    return 1 if bad or reloads == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
