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
generated ISA of every lattice kernel as a forward data-flow problem over ALL edges of its control-flow graph: the set of
registers that may have a reload in flight, emptied only by a full `s_waitcnt lgkmcnt(0)`; any instruction that touches a
register of the set is reported.

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

    Forward data flow over the control-flow graph of every lattice kernel -- ALL edges, to a fixed point: the state is the
    set of registers an in-place reload may still have in flight; an inline-assembly `ds_read*` adds its destination, a
    full `s_waitcnt lgkmcnt(0)` empties the set (a counted wait retires nothing here: the kernels' own rule is that only
    the zero wait in front of the block barrier makes the registers valid), and any instruction that reads or writes a
    register of the set -- or a reload whose ADDRESS register is in it -- is a violation.  (Until the end of round 5 this
    walked one path per kernel with the LDS queue modelled in order; that missed whatever sits on the other edges.)"""
    lines = open(path).read().split("\n")
    funcs, cur = [], None
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = [m.group(1), i, None]
            funcs.append(cur)
        elif cur is not None and cur[2] is None and re.match(r"^\s*s_endpgm", line):
            cur[2] = i
    kernels, reloads, bad = 0, 0, []
    for fn, start, end in funcs:
        if "k_lattice" not in fn or end is None:
            continue
        kernels += 1
        # instructions of the kernel: (line number, text, in inline assembly?)
        insts, labels, in_asm = [], {}, False
        for i in range(start + 1, end + 1):
            raw = lines[i]
            st = raw.strip()
            if st.startswith(";;#ASMSTART"):
                in_asm = True; continue
            if st.startswith(";;#ASMEND"):
                in_asm = False; continue
            lm = re.match(r"^(\.LBB\w+):", raw)
            if lm:
                labels[lm.group(1)] = len(insts); continue
            t = raw.split(";")[0].rstrip()
            if not t.strip() or t.lstrip().startswith(".") or re.match(r"^\.?\w+:", t.strip()):
                continue
            insts.append((i + 1, t.strip(), in_asm))
        n = len(insts)
        succ = [[] for _ in range(n)]
        for k, (_, t, _) in enumerate(insts):
            b = re.match(r"^s_c?branch\w*\s+(\.LBB\w+)", t)
            if t.startswith("s_endpgm"):
                continue
            if b and b.group(1) in labels and labels[b.group(1)] < n:
                succ[k].append(labels[b.group(1)])
            if not t.startswith("s_branch") and k + 1 < n:
                succ[k].append(k + 1)
        reloads += sum(1 for (_, t, a) in insts if a and RELOAD.match("\t" + t))
        state_in = [None] * n           # set of registers possibly in flight on entry
        state_in[0] = frozenset()
        work = [0]
        reported = set()
        while work:
            k = work.pop()
            ln, t, a = insts[k]
            inset = state_in[k]
            out = inset
            w = re.match(r"^s_waitcnt\b(.*)", t)
            if w:
                if re.search(r"lgkmcnt\(0\)", w.group(1)) or re.match(r"^\s*0\s*$", w.group(1)):
                    out = frozenset()
            else:
                r = RELOAD.match("\t" + t) if a else None
                if r:
                    dst, addr = regs_of(r.group(2)), regs_of(r.group(3))
                    hit = (addr | dst) & inset
                    if hit and ln not in reported:
                        reported.add(ln); bad.append((fn, ln, t, sorted(hit)))
                    out = inset | dst
                else:
                    hit = regs_of(t) & inset
                    if hit and ln not in reported:
                        reported.add(ln); bad.append((fn, ln, t, sorted(hit)))
            for s_ in succ[k]:
                merged = out if state_in[s_] is None else (state_in[s_] | out)
                if merged != state_in[s_]:
                    state_in[s_] = merged
                    work.append(s_)
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
