#!/usr/bin/env python
"""Static check of the in-place LDS reloads of the column-block lattice kernels (csrc/lattice_step.h, lattice_wd.hip).

The compute wave refills the registers of a block's (blank, label) pairs and boundary seeds with the NEXT block's values
while it is still working on the current block: `ds_read2st64_b64` / `ds_read_b32` in inline assembly, which the compiler
does not count.  The data lands some hundred cycles later; the only thing that makes the registers valid is the
`s_waitcnt lgkmcnt(0)` in front of the block's barrier.  Nothing may read or write those registers in between -- and
the one who could is the compiler (a register copy at a loop head, a spill, a reuse as a temporary), silently.

(Round 5 tried leaving the last reloads in flight ACROSS the barrier with counted waits: this script found nothing wrong
with it and neither did any test -- tools/wd_soak.py with three processes on one GPU did: lattice_step.h.  The check is
necessary, not sufficient.)

This script compiles lattice_wd.hip for gfx950 with -save-temps (hipcc cross-compiles, no GPU needed) and walks the
generated ISA of every lattice kernel with the wave's LDS operations modelled as the in-order queue they are
(`lgkmcnt(N)` retires all but the N youngest): any instruction that touches a register whose reload is still in the
queue is reported.  One path per kernel: straight-line code, fall-through edges, unconditional branches followed once --
which takes the walk around every loop body twice, the second time with what the first left in flight.

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
    """Returns (kernels seen, in-place reloads seen, [violations]).

    Walks every lattice kernel along ONE path: straight-line code, the fall-through edge of conditional branches, and
    unconditional branches followed to their label (each branch site once, which takes the walk around every loop body
    a second time -- with whatever the first pass left in flight).  The wave's LDS operations are modelled as the
    in-order queue they are: `s_waitcnt lgkmcnt(N)` retires all but the N youngest, and a register is "in flight" from
    its reload until that reload retires."""
    lines = open(path).read().split("\n")
    # function extents and labels
    funcs, cur = [], None
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = [m.group(1), i, None, {}]
            funcs.append(cur)
        elif cur is not None and cur[2] is None:
            lm = re.match(r"^(\.LBB\w+):", line)
            if lm:
                cur[3][lm.group(1)] = i
            if re.match(r"^\s*s_endpgm", line):
                cur[2] = i
    kernels, reloads, bad = 0, 0, []
    for fn, start, end, labels in funcs:
        if "k_lattice" not in fn or end is None:
            continue
        kernels += 1
        fifo = []            # [(line, regs-or-None)]: outstanding LDS operations, oldest first; regs for in-place reloads
        followed, seen_reload_lines, reported = set(), set(), set()
        i, in_asm, steps = start + 1, False, 0
        while i <= end and steps < 400000:
            steps += 1
            raw = lines[i]
            ln = i + 1
            st = raw.strip()
            if st.startswith(";;#ASMSTART"):
                in_asm = True; i += 1; continue
            if st.startswith(";;#ASMEND"):
                in_asm = False; i += 1; continue
            s = raw.split(";")[0].rstrip()
            if not s.strip() or s.lstrip().startswith(".") or re.match(r"^\.?\w+:", s.strip()):
                i += 1; continue
            if re.match(r"^\s*s_endpgm", s):
                break
            w = re.match(r"^\s*s_waitcnt\b(.*)", s)
            if w:
                m = re.search(r"lgkmcnt\((\d+)\)", w.group(1))
                if m:
                    keep = int(m.group(1))
                    fifo = fifo[len(fifo) - keep:] if keep else []
                elif re.match(r"^\s*s_waitcnt\s+0\s*$", s):
                    fifo = []
                i += 1; continue
            b = re.match(r"^\s*s_branch\s+(\.LBB\w+)", s)
            if b:
                if i in followed or b.group(1) not in labels:
                    fifo = []      # this path has been walked: go on behind the branch as a fresh one
                    i += 1
                else:
                    followed.add(i)
                    i = labels[b.group(1)]
                continue
            pending = set()
            for _, regs in fifo:
                if regs:
                    pending |= regs
            r = RELOAD.match(s) if in_asm else None
            touched = regs_of(s)
            if r:
                dst, addr = regs_of(r.group(2)), regs_of(r.group(3))
                hit = addr & pending
                if hit and ln not in reported:
                    reported.add(ln); bad.append((fn, ln, s.strip(), sorted(hit)))
                if ln not in seen_reload_lines:
                    seen_reload_lines.add(ln); reloads += 1
                fifo.append((ln, dst))
                i += 1; continue
            hit = touched & pending
            if hit and ln not in reported:
                reported.add(ln); bad.append((fn, ln, s.strip(), sorted(hit)))
            if re.match(r"^\s*(ds_|s_load|s_buffer_load)", s):
                fifo.append((ln, None))     # any other operation of the same counter: in the queue, nothing to protect
            i += 1
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
