"""NumPy model of the arithmetic of the probability-domain lattice kernel (csrc/lattice_pd.hip).

TEST INFRASTRUCTURE: an executable statement of WHAT that kernel computes per lane and per diagonal --
the operations in the kernel's order and precision, the per-lane binary exponents, the renormalisation
every K diagonals, the input check that sends an utterance to the log-domain kernel -- so that the scheme
can be checked against the fp64 oracle on the CPU (tests/test_pd_model.py) and a GPU mismatch can be
bisected.  It models no waves, LDS or barriers: the column to the left of a wave boundary hands over
exactly what a lane inside the wave would (same value, same exponent), so one vector over all columns is
the same arithmetic.

Scheme (alpha; beta is the mirrored lattice):
    probabilities      p = exp2(fl32(lp * log2e))  in fp32 (v_mul_f32 + v_exp_f32), widened to fp64
    state per column   Y, X fp64 and an integer exponent E:  true value = Y * 2**E
    per diagonal       val = fma(X[u-1], 2**(E[u-1]-E[u]), Y)          (the factor is a power of two: exact)
                       Y   = val * pB(cell),  X = val * pL(cell)
    every K diagonals  every column pulls the binary exponent out of Y (exact rescale of Y and X), columns
                       that have not started yet adopt the exponent of the last started column, and the
                       factors 2**(E[u-1]-E[u]) are rebuilt.
    fp64 keeps 2**+-1022: K steps of the smallest fp32-representable probability (2**-126) cannot leave it,
    so the only bookkeeping inside an interval is none.  What is checked, off the chain: every finite log-prob is
    above -80; the exponent gap between neighbouring live columns at a renormalisation stays within MAX_GAP (a
    frame at which the label probabilities step up puts (pL(t+1)/pL(t))**u between two columns of one diagonal --
    2**1558 with every log-prob above -10 -- and the factor 2**gap must stay an fp64); and every column's final
    value is a positive finite number (an overflow, a NaN or a total underflow anywhere upstream is sticky).
    A sweep that fails any of them is redone by the log-domain kernel.
    stored alpha       = ln2 * (log2(m) + e) with m = the top 24 significant bits of val as an fp32 in [1,2),
                         e = its binary exponent + E   (v_log_f32, one multiply, one fma: <= 0.5 ulp of the result)
"""
import numpy as np

K = 8                  # diagonals between renormalisations (= one block of the kernel)
LP_MIN = -80.0         # log-probs below this (p would flush to 0 in fp32) send the utterance to the log-domain kernel
MAX_GAP = 400          # largest |E[u-1] - E[u]| between live columns the fp64 state is trusted with (lattice_pd.hip)
GAP_CLAMP = 1000       # the shift is clamped so that the factor stays finite whatever the gap
LOG2E = np.float32(1.44269504088896340736)
LN2 = np.float32(0.693147180559945309417)


def to_prob(lp):
    """exp2(lp * log2e) as the loader wave evaluates it (v_mul_f32 + v_exp_f32), then v_cvt_f64_f32."""
    with np.errstate(under="ignore"):
        p = np.exp2((lp.astype(np.float32) * LOG2E).astype(np.float32)).astype(np.float32)
    p[p < np.float32(2.0 ** -126)] = 0          # v_exp_f32 does not produce denormals
    return p.astype(np.float64)


def out_log(val, E):
    """What the storer wave writes for value val * 2**E (val fp64 > 0): the double's top 24 significant bits
    re-labelled as an fp32 in [1,2) (v_alignbit_b32 + v_bfi_b32: truncation, relative error < 2**-23),
    v_log_f32, and ln2*(log2 + exponent) as one multiply and one fma."""
    val = np.ascontiguousarray(val, dtype=np.float64)
    bits = val.view(np.uint64)
    mant23 = ((bits >> np.uint64(29)) & np.uint64(0x7FFFFF)).astype(np.uint32)
    m = (mant23 | np.uint32(0x3F800000)).view(np.float32)
    e11 = ((bits >> np.uint64(52)) & np.uint64(0x7FF)).astype(np.int64)
    l2 = np.log2(m).astype(np.float32)
    et = (E + e11 - 1023).astype(np.float32)
    small = (l2 * LN2).astype(np.float32)
    out = (et.astype(np.float64) * np.float64(LN2) + small.astype(np.float64)).astype(np.float32)   # one fma
    return np.where(val > 0, out, np.float32(-np.inf))


def sweep(lpB, lpL, beta=False, k_renorm=K):
    """One direction of one utterance.  lpB, lpL: (T,U) log-probs (lpL[:, U-1] unused).
    Returns (log-values (T,U) fp32, log-likelihood fp32 [alpha only, else None], chain_ok): chain_ok is False when
    the kernel would flag the sweep for the log-domain kernel (column gap beyond MAX_GAP, non-finite or zero final
    value)."""
    T, U = lpB.shape
    if beta:
        # mirrored lattice: sweep cell (t',u') is lattice cell (T-1-t', U-1-u'); both weights of the beta
        # recurrence are the RECEIVING cell's own probabilities, so mirroring the two planes is all it takes
        lpB = lpB[::-1, ::-1]
        lpL = lpL[::-1, ::-1]
    pB, pL = to_prob(lpB), to_prob(lpL)
    out = np.full((T, U), np.nan, dtype=np.float32)
    Y = np.zeros(U)
    X = np.zeros(U)
    E = np.zeros(U, np.int64)
    Y[0] = 1.0
    c = np.ones(U)
    ucol = np.arange(U)
    gap_hi = gap_lo = 0
    for d in range(T + U - 1):
        if d % k_renorm == 0:
            started = (ucol <= d - 1) | (ucol == 0)      # column 0 carries the initial 1 from the start
            m, e = np.frexp(Y)                           # v_frexp_mant_f64 / v_frexp_exp_i32_f64 (0 -> 0, 0)
            Y = m
            X = np.ldexp(X, -e)
            E = E + e
            front = max(d - 1, 0)
            E = np.where(started, E, E[min(front, U - 1)])
            dE = np.zeros(U, np.int64)
            dE[1:] = E[:-1] - E[1:]
            done = (d - ucol) >= T                       # a finished column's exponent is frozen: not a lattice gap
            judged = np.where(done, 0, dE)
            gap_hi, gap_lo = max(gap_hi, int(judged.max())), min(gap_lo, int(judged.min()))
            with np.errstate(over="ignore", under="ignore"):
                c = np.ldexp(np.ones(U), np.clip(dE, -GAP_CLAMP, GAP_CLAMP))
            c[0] = 0.0
        t = d - ucol
        live = (t >= 0) & (t < T)
        tc = np.clip(t, 0, T - 1)
        b, l = pB[tc, ucol], pL[tc, ucol]
        Xl = np.zeros(U)
        Xl[1:] = X[:-1]
        with np.errstate(over="ignore", invalid="ignore", under="ignore"):
            if beta:
                val = Xl * (l * c) + Y * b               # tmp = pL*c ; S = Y*pB ; val = fma(X_left, tmp, S)
                Yn, Xn = val, val
            else:
                val = Xl * c + Y                         # fma
                Xn = val * l
                Yn = val * b
            out[tc[live], ucol[live]] = out_log(np.nan_to_num(val[live], nan=0.0, posinf=0.0), E[live])
        Y = np.where(live, Yn, Y)
        X = np.where(live, Xn, X)
    chain_ok = bool(gap_hi <= MAX_GAP and gap_lo >= -MAX_GAP and np.all((Y > 0) & np.isfinite(Y)))
    ll = None
    if not beta:
        ll = out_log(np.nan_to_num(Y[U - 1:U], nan=0.0, posinf=0.0), E[U - 1:U])[0]
    else:
        out = out[::-1, ::-1]
    return out, ll, chain_ok


def in_range(lp2):
    """The loader waves' input check: every log-prob the sweep touches is >= LP_MIN (-inf and NaN fail it
    too: those keep the reference's log-domain semantics, core.cu:26-39)."""
    v = lp2[..., 0], lp2[:, :-1, 1]
    return bool(all(np.all(x >= LP_MIN) for x in v))


def lattice(lp2, k_renorm=K):
    """(T,U,2) gathered log-probs of one utterance -> alphas, betas (log, fp32), ll_alpha, ok.  ok = the kernel
    keeps both sweeps (inputs in range and both chains in range); otherwise the log-domain kernel redoes what was
    flagged and the values returned here are not what the caller sees."""
    lpB, lpL = lp2[..., 0], lp2[..., 1]
    al, ll, ok_a = sweep(lpB, lpL, beta=False, k_renorm=k_renorm)
    be, _, ok_b = sweep(lpB, lpL, beta=True, k_renorm=k_renorm)
    return al, be, ll, in_range(lp2) and ok_a and ok_b
