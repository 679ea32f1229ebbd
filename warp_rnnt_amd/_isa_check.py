"""Static check of the in-place LDS reloads of the column-block lattice kernels (csrc/lattice_step.h, lattice_wd.hip),
run by _build.build() on the ISA of the very object that goes into the library -- a violation FAILS THE BUILD.

The compute wave refills the registers of a block's (blank, label) pairs and boundary seeds with the NEXT block's values
while it is still working on the current block: `ds_read2st64_b64` / `ds_read_b128` in inline assembly, which the compiler
does not count.  The data lands some hundred cycles later; the only thing that makes the registers valid is the
`s_waitcnt lgkmcnt(0)` in front of the block's barrier.  Nothing may read or write those registers in between -- and
the one who could is the compiler (a register copy at a loop head, a live-range split, a spill, a reuse as a temporary),
silently, and differently with every compiler version or flag.  Round 5 shipped two silent wrong-answer bugs of this
family that only a multi-process soak saw (lattice_step.h: wait_lds); round 6 made this check part of the build and saw
it refuse a well-meant change of the wait's operand constraints (96 compiler-inserted copies of in-flight registers).

The check walks the generated ISA of every lattice kernel as a forward data-flow problem over ALL edges of its
control-flow graph: the state is the set of registers that may have a reload in flight, emptied only by a full
`s_waitcnt lgkmcnt(0)`; any instruction that touches a register of the set is reported.  Necessary, not sufficient
(tools/wd_soak.py is the other half).
"""
import re

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
RELOAD = re.compile(r"^\s*(ds_read2st64_b64|ds_read_b32|ds_read_b64|ds_read_b128)\s+(v\d+|v\[\d+:\d+\])\s*,\s*(v\d+)")


class ReloadCheckError(RuntimeError):
    """The generated ISA touches a register whose in-place reload may still be in flight (or holds no reload at all
    where some are expected: the check would be vacuous)."""


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


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


def require_clean(path, min_kernels=1, min_reloads=1):
    """check(path), as a gate: raises ReloadCheckError on any violation, and when fewer kernels / reloads were seen than
    the caller knows the file to hold (a pattern that stopped matching must not pass as "nothing wrong")."""
    kernels, reloads, bad = check(path)
    if kernels < min_kernels or reloads < min_reloads:
        raise ReloadCheckError(f"{path}: {kernels} lattice kernels / {reloads} in-place reloads found, expected at least "
                               f"{min_kernels} / {min_reloads}: the check no longer sees what it is there to check")
    if bad:
        lines = "\n".join(f"  line {ln}: `{text}` touches v{regs} while its reload is in flight   [{fn[:60]}]"
                          for fn, ln, text, regs in bad[:12])
        raise ReloadCheckError(f"{path}: {len(bad)} instruction(s) touch a register whose in-place LDS reload may still be "
                               f"in flight (csrc/lattice_step.h: wait_lds) -- this build would compute wrong lattices under "
                               f"load:\n{lines}")
    return kernels, reloads


# ---------------------------------------------------------------------------------------------------------------------
# Second rule, on the SOURCES: no inline-assembly VMEM store of more than 64 bits.
#
# gfx940 / gfx950: a global / buffer / flat store of more than 64 bits of data reads its data registers up to two
# wait states after it issues, and a VALU write of one of them inside that window lands in the stored value.  The
# compiler pads its own stores (GCNHazardRecognizer); it does not look inside an `asm` statement, so an inline
# `global_store_dwordx4 ... sc1` followed by the compiler's next VALU instruction is a coin toss.  Found in round 6 by a
# micro-benchmark's bit check (tools/ubench/lsm_store_policy.hip: 0.18 % of the float4 of one lane group wrong, every
# cache policy, only the inline-assembly forms); the library's one inline store is a dwordx2 (prologue.hip: the dense
# gather's write-through pairs), which has no such window.  This rule keeps it that way; the one form it lets through is the
# store with an `s_nop 1` of its own behind it in the same assembly string.
WIDE_ASM_STORE = re.compile(r"\b(global|buffer|flat|scratch)_store_(dwordx[34]|b96|b128)\b")


def wide_asm_stores(path):
    """[(line number, text)] of inline-assembly VMEM stores of more than 64 bits in a source file: lines inside an
    `asm` statement (from the `asm` keyword to the `;` that ends it) that name such an instruction in a string."""
    out, in_asm = [], False
    for i, line in enumerate(open(path).read().split("\n"), 1):
        code = line.split("//")[0]
        if re.search(r"\basm\b", code):
            in_asm = True
        if in_asm and WIDE_ASM_STORE.search(code) and not re.search(r"\\n\\ts_nop [1-9]\"", code):
            out.append((i, line.strip()))      # (allowed: the store with its own `s_nop 1` behind it in the same string)
        if in_asm and ";" in code.split('"')[-1]:
            in_asm = False
    return out


def require_no_wide_asm_stores(paths):
    bad = [(p, ln, t) for p in paths for ln, t in wide_asm_stores(p)]
    if bad:
        lines = "\n".join(f"  {p}:{ln}: {t[:120]}" for p, ln, t in bad[:12])
        raise ReloadCheckError("inline-assembly VMEM store(s) of more than 64 bits: on gfx950 the VALU instruction the compiler "
                               "puts behind one may overwrite the data before the store has read it (no hazard padding inside "
                               "or after `asm`); use the compiler's store, or a dwordx2:\n" + lines)


# ---------------------------------------------------------------------------------------------------------------------
# Third rule, on the generated ISA: the manually-inserted-wait-state hazards of gfx940 / gfx950 that the compiler's hazard
# recognizer resolves for its own instructions and cannot resolve for (or against) the inside of an `asm` statement:
#   * a DPP instruction reads, through the DPP path (its first source), a VGPR a VALU instruction wrote fewer than 2 wait
#     states before; a DPP instruction fewer than 5 wait states behind a VALU write of EXEC;
#   * an ordinary VALU instruction reads the result of a transcendental (v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin /
#     v_cos) in the very next slot (1 wait state needed);
#   * a VALU instruction writes a data register of a VMEM store of more than 64 bits fewer than 2 wait states behind it
#     (the second rule above, seen from the ISA side);
#   * a VMEM instruction reads an SGPR a VALU instruction (v_readfirstlane, v_readlane, a compare) wrote fewer than 5 wait
#     states before -- the descriptor, offset and LDS base of the inline LDS-DMA loads come out of v_readfirstlane;
#   * an LDS-DMA load (`... lds`) in the slot behind the SALU write of M0 (1 wait state needed).
# The hand-written lattice steps (csrc/lattice_step.h) keep these "by construction" -- and by one instruction the COMPILER
# places between two statements (the store of the cell's value).  This makes the construction a checked property of every
# build: forward data flow over all edges of each kernel's control-flow graph, the state being the wait states since the
# last such write of each register (capped), merged by minimum.  `s_nop N` counts N + 1 wait states, everything else 1.
# Only findings with an inline-assembly instruction on one side fail the build (a finding between two compiler
# instructions would be an error of this model, not of the compiler: reported, not fatal -- none so far).
TRANS = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)(_legacy|_iflag|_clamp)?_(f32|f16|f64|bf16)")
DPPCTL = re.compile(r"quad_perm:|row_shl:|row_shr:|row_ror:|wave_shl|wave_shr|wave_rol|wave_ror|row_mirror|row_half_mirror|"
                    r"row_bcast|row_newbcast")
WIDE_STORE_ISA = re.compile(r"^(?:global|flat|scratch)_store_(?:dwordx3|dwordx4|b96|b128)\s+\S+,\s*(v\[\d+:\d+\])|"
                            r"^buffer_store_(?:dwordx3|dwordx4|b96|b128)\s+(v\[\d+:\d+\])")
SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")
VMEM = re.compile(r"^(buffer|tbuffer|global|flat|scratch|image)_")
HAZARD_CAP = 6


def sregs_of(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    if re.search(r"\bvcc\b", text):
        out.add("vcc")
    return out


class AsmHazardError(ReloadCheckError):
    """An inline-assembly instruction sits inside a hazard window the compiler does not pad."""


def _operands(text):
    m = re.match(r"^(\S+)\s*(.*)$", text)
    rest = m.group(2)
    return m.group(1), ([o.strip() for o in re.split(r",(?![^\[]*\])", rest)] if rest else [])


def check_hazards(path):
    """Returns (kernels, instructions, [(function, line, text, what, involves inline assembly)])."""
    lines = open(path).read().split("\n")
    funcs, cur = [], None
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = [m.group(1), i, None]
            funcs.append(cur)
        elif cur is not None and cur[2] is None and re.match(r"^\s*s_endpgm", line):
            cur[2] = i
    found, nk, ni = [], 0, 0
    for fn, start, end in funcs:
        if end is None:
            continue
        nk += 1
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
            t = raw.split(";")[0].strip()
            if not t or t.startswith(".") or re.match(r"^\.?\w+:", t):
                continue
            insts.append((i + 1, t, in_asm))
        n = len(insts)
        ni += n
        succ = [[] for _ in range(n)]
        dec = []
        for k, (_, t, _) in enumerate(insts):
            if not t.startswith("s_endpgm"):
                b = re.match(r"^s_c?branch\w*\s+(\.LBB\w+)", t)
                if b and b.group(1) in labels and labels[b.group(1)] < n:
                    succ[k].append(labels[b.group(1)])
                if not t.startswith("s_branch") and k + 1 < n:
                    succ[k].append(k + 1)
            mn, ops = _operands(t)
            valu = mn.startswith("v_")
            d = dict(ws=(int(ops[0], 0) + 1) if mn == "s_nop" else 1, valu=valu, trans=bool(TRANS.match(mn)),
                     dpp=valu and ("_dpp" in mn or bool(DPPCTL.search(t))), dst=set(), src=set(), dppsrc=set(), wide=set(),
                     cmpx=mn.startswith("v_cmpx"))
            if valu and ops:
                has_vdst = bool(re.match(r"^v(\d+|\[)", ops[0]))
                d["dst"] = regs_of(ops[0]) if has_vdst else set()
                if mn.startswith("v_swap"):
                    d["dst"] |= regs_of(ops[1])
                srcs = ops[1:]
                d["src"] = set().union(*[regs_of(o) for o in srcs]) if srcs else set()
                if "mac" in mn:                         # (v_fmac / v_mac accumulate into their destination)
                    d["src"] |= d["dst"]
                if d["dpp"] and srcs:
                    d["dppsrc"] = regs_of(srcs[0])      # the operand that goes through the DPP path
            w = WIDE_STORE_ISA.match(t)
            if w:
                d["wide"] = regs_of(w.group(1) or w.group(2))
            d["sdst"] = set()
            if valu and ops:        # SGPRs a VALU instruction writes: readlane / readfirstlane, e64 compares, carry outs
                if not re.match(r"^v(\d+|\[)", ops[0]):
                    d["sdst"] = sregs_of(ops[0])
                elif len(ops) > 1 and re.match(r"^v_(add|sub|subrev|addc|subb|subbrev)_co|^v_mad_[ui]64|^v_div_scale", mn):
                    d["sdst"] = sregs_of(ops[1])
                if mn.startswith("v_cmp") and mn.endswith("_e32"):
                    d["sdst"] = {"vcc"}
            d["vmem"] = bool(VMEM.match(mn))
            d["ssrc"] = sregs_of(t) if d["vmem"] else set()
            d["m0w"] = mn.startswith("s_") and bool(ops) and ops[0] == "m0"
            # SGPRs a scalar instruction overwrites (a later reader sees the scalar value: the VALU write before it is dead)
            d["skill"] = (sregs_of(ops[0]) if mn.startswith("s_") and ops and re.match(r"^(s\d+|s\[|vcc)", ops[0]) and not
                          re.match(r"^s_(cmp|cmpk|bitcmp|cbranch|branch|waitcnt|nop|set|sleep|barrier|sendmsg|store|buffer_store|"
                                   r"dcache|icache|endpgm|trap|ttrace|code_end)", mn) else set())
            d["ldsdma"] = d["vmem"] and bool(re.search(r"\blds\b", t))
            dec.append(d)
        state_in = [None] * n           # {(kind, register): (wait states since, written inside inline assembly?)}
        state_in[0] = {}
        work, reported = [0], set()

        def report(ln, t, kind, what, asm_side):
            if (ln, kind) not in reported:
                reported.add((ln, kind))
                found.append((fn, ln, t, what, asm_side))

        while work:
            k = work.pop()
            ln, t, a = insts[k]
            d, st = dec[k], state_in[k]
            if d["valu"]:
                for r in d["dppsrc"]:
                    e = st.get(("v", r))
                    if e and e[0] < 2:
                        report(ln, t, "dpp", f"DPP read of v{r} {e[0]} wait state(s) behind its VALU write (2 needed)", a or e[1])
                if d["dpp"]:
                    e = st.get(("x", 0))
                    if e and e[0] < 5:
                        report(ln, t, "exec", f"DPP {e[0]} wait state(s) behind a VALU write of EXEC (5 needed)", a or e[1])
                if not d["trans"]:
                    for r in d["src"]:
                        e = st.get(("t", r))
                        if e and e[0] < 1:
                            report(ln, t, "trans", f"VALU read of v{r} in the slot behind the transcendental that wrote it "
                                                   f"(1 wait state needed)", a or e[1])
                for r in d["dst"]:
                    e = st.get(("w", r))
                    if e and e[0] < 2:
                        report(ln, t, "wide", f"VALU write of v{r} {e[0]} wait state(s) behind a store of more than 64 bits that "
                                              f"reads it (2 needed)", a or e[1])
            if d["vmem"]:
                for r in d["ssrc"]:
                    e = st.get(("s", r))
                    if e and e[0] < 5:
                        report(ln, t, "sgpr", f"VMEM read of s{r} {e[0]} wait state(s) behind its VALU write (5 needed)", a or e[1])
                e = st.get(("m", 0))
                if d["ldsdma"] and e and e[0] < 1:
                    report(ln, t, "m0", "LDS-DMA in the slot behind the write of M0 (1 wait state needed)", a or e[1])
            out = {}
            for key, (age, ia) in st.items():
                if age + d["ws"] < HAZARD_CAP:
                    out[key] = (age + d["ws"], ia)
            if d["valu"]:
                for r in d["dst"]:
                    out[("v", r)] = (0, a)
                    if d["trans"]:
                        out[("t", r)] = (0, a)
                    else:
                        out.pop(("t", r), None)
                if d["cmpx"]:
                    out[("x", 0)] = (0, a)
                for r in d["sdst"]:
                    out[("s", r)] = (0, a)
            for r in d["skill"]:
                out.pop(("s", r), None)
            if d["m0w"]:
                out[("m", 0)] = (0, a)
            for r in d["wide"]:
                out[("w", r)] = (0, a)
            for s_ in succ[k]:
                cur_ = state_in[s_]
                if cur_ is None:
                    state_in[s_] = dict(out)
                    work.append(s_)
                    continue
                changed = False
                for key, v in out.items():
                    c = cur_.get(key)
                    if c is None or v[0] < c[0] or (v[0] == c[0] and v[1] and not c[1]):
                        cur_[key] = (v[0], v[1] or (c is not None and c[0] == v[0] and c[1]))
                        changed = True
                if changed:
                    work.append(s_)
    return nk, ni, found


def require_no_asm_hazards(path, min_kernels=1):
    """check_hazards(path) as a gate: raises AsmHazardError when a finding has an inline-assembly instruction on either side.
    Returns (kernels, instructions, findings between compiler instructions only)."""
    nk, ni, found = check_hazards(path)
    if nk < min_kernels:
        raise AsmHazardError(f"{path}: {nk} kernels found, expected at least {min_kernels}: the check no longer sees its input")
    bad = [f for f in found if f[4]]
    if bad:
        lines = "\n".join(f"  line {ln}: `{text}`: {what}   [{fn[:60]}]" for fn, ln, text, what, _ in bad[:12])
        raise AsmHazardError(f"{path}: {len(bad)} hazard(s) the compiler does not pad around inline assembly:\n{lines}")
    return nk, ni, [f for f in found if not f[4]]
