"""fp64 NumPy restatement of the transducer forward-backward -- TEST INFRASTRUCTURE ONLY.

BASELINE.json names awni/transducer's ``ref_transduce.py`` (pinned by the
reference's README.md:6,11 at commit 6b37e98c21551c7ed2181e2f526053bae8ae94d2)
as the CPU comparator.  That file is NOT vendored in /root/reference and is
not available in this image (no network), and no reference test runs against
it, so the boundary to it is "parity unpinned".  This module restates the
published algorithm (Graves 2012, eqs. 16-20; the same maths as SURVEY.md
Appendix A) in the per-utterance style of that script: a forward pass, a
backward pass, and gradients w.r.t. the log-probabilities, looped over the
batch.  It is anchored instead on the reference's own golden vectors
(tests/golden/reference_vectors.json), see tests/test_oracle.py.

Everything is float64; it is the "truth" leg used to report how far any fp32
implementation (the reference's, the C oracle's, ours) sits from exact
arithmetic.
"""
import numpy as np


def forward_pass(log_probs, labels, blank):
    """alphas (T,U) and log-likelihood for one utterance; log_probs (T,U,V)."""
    T, U, _ = log_probs.shape
    lpb = log_probs[:, :, blank]
    lpl = np.zeros((T, U))
    if U > 1:
        lpl[:, :U - 1] = log_probs[:, np.arange(U - 1), labels[:U - 1]]
    alphas = np.full((T, U), -np.inf)
    alphas[0, 0] = 0.0
    for t in range(1, T):
        alphas[t, 0] = alphas[t - 1, 0] + lpb[t - 1, 0]
    for u in range(1, U):
        alphas[0, u] = alphas[0, u - 1] + lpl[0, u - 1]
    for t in range(1, T):
        for u in range(1, U):
            skip = alphas[t - 1, u] + lpb[t - 1, u]
            emit = alphas[t, u - 1] + lpl[t, u - 1]
            alphas[t, u] = np.logaddexp(skip, emit)
    return alphas, alphas[T - 1, U - 1] + lpb[T - 1, U - 1]


def backward_pass(log_probs, labels, blank):
    """betas (T,U) and log-likelihood (= betas[0,0]) for one utterance."""
    T, U, _ = log_probs.shape
    lpb = log_probs[:, :, blank]
    lpl = np.zeros((T, U))
    if U > 1:
        lpl[:, :U - 1] = log_probs[:, np.arange(U - 1), labels[:U - 1]]
    betas = np.full((T, U), -np.inf)
    betas[T - 1, U - 1] = lpb[T - 1, U - 1]
    for t in range(T - 2, -1, -1):
        betas[t, U - 1] = betas[t + 1, U - 1] + lpb[t, U - 1]
    for u in range(U - 2, -1, -1):
        betas[T - 1, u] = betas[T - 1, u + 1] + lpl[T - 1, u]
    for t in range(T - 2, -1, -1):
        for u in range(U - 2, -1, -1):
            skip = betas[t + 1, u] + lpb[t, u]
            emit = betas[t, u + 1] + lpl[t, u]
            betas[t, u] = np.logaddexp(skip, emit)
    return betas, betas[0, 0]


def _sweeps_fast(lpb, lpl):
    """Same recurrences vectorised along anti-diagonals (for the big cases)."""
    T, U = lpb.shape
    al = np.full((T, U), -np.inf)
    be = np.full((T, U), -np.inf)
    al[0, 0] = 0.0
    be[T - 1, U - 1] = lpb[T - 1, U - 1]
    for d in range(1, T + U - 1):
        u = np.arange(max(0, d - (T - 1)), min(U - 1, d) + 1)
        t = d - u
        skip = np.full(u.shape, -np.inf)
        emit = np.full(u.shape, -np.inf)
        m = t > 0
        skip[m] = al[t[m] - 1, u[m]] + lpb[t[m] - 1, u[m]]
        m = u > 0
        emit[m] = al[t[m], u[m] - 1] + lpl[t[m], u[m] - 1]
        al[t, u] = np.logaddexp(skip, emit)
        # mirrored cell for beta
        tb, ub = T - 1 - t, U - 1 - u
        skip = np.full(u.shape, -np.inf)
        emit = np.full(u.shape, -np.inf)
        m = tb < T - 1
        skip[m] = be[tb[m] + 1, ub[m]] + lpb[tb[m], ub[m]]
        m = ub < U - 1
        emit[m] = be[tb[m], ub[m] + 1] + lpl[tb[m], ub[m]]
        be[tb, ub] = np.logaddexp(skip, emit)
    return al, be


def compute_gradient(log_probs, alphas, betas, labels, blank, fastemit_lambda=0.0):
    """d(-log-likelihood)/d(log_probs) for one utterance, shape (T,U,V)."""
    T, U, _ = log_probs.shape
    grads = np.zeros_like(log_probs, dtype=np.float64)
    ll = betas[0, 0]
    lpb = log_probs[:, :, blank]
    gb = np.zeros((T, U))
    gb[:T - 1, :] = -np.exp(alphas[:T - 1, :] + betas[1:, :] + lpb[:T - 1, :] - ll)
    gb[T - 1, U - 1] = -np.exp(alphas[T - 1, U - 1] + lpb[T - 1, U - 1] - ll)
    grads[:, :, blank] = gb
    for u in range(U - 1):
        lab = labels[u]
        gl = -(1.0 + fastemit_lambda) * np.exp(
            alphas[:, u] + betas[:, u + 1] + log_probs[:, u, lab] - ll)
        grads[:, u, lab] = gl  # overwrite, like the reference's dense kernels
    return grads


def transduce(log_probs, labels, blank=0, fastemit_lambda=0.0, fast=False):
    """(cost, grads, alphas, betas) for one utterance, all float64."""
    log_probs = np.asarray(log_probs, dtype=np.float64)
    labels = np.asarray(labels, dtype=np.int64)
    if fast:
        T, U, _ = log_probs.shape
        lpb = log_probs[:, :, blank]
        lpl = np.zeros((T, U))
        if U > 1:
            lpl[:, :U - 1] = log_probs[:, np.arange(U - 1), labels[:U - 1]]
        alphas, betas = _sweeps_fast(lpb, lpl)
    else:
        alphas, _ = forward_pass(log_probs, labels, blank)
        betas, _ = backward_pass(log_probs, labels, blank)
    grads = compute_gradient(log_probs, alphas, betas, labels, blank, fastemit_lambda)
    return -betas[0, 0], grads, alphas, betas


def transduce_batch(log_probs, labels, flen, glen, blank=0, fastemit_lambda=0.0, fast=False):
    """Batch driver: log_probs (N,T,U,V), labels (N,U-1), flen (N,), glen (N,).

    Returns costs (N,) and grads (N,T,U,V) (zero outside each utterance's
    (flen, glen+1) window).
    """
    log_probs = np.asarray(log_probs, dtype=np.float64)
    N = log_probs.shape[0]
    grads = np.zeros_like(log_probs)
    costs = np.zeros((N,))
    for n in range(N):
        t, u = int(flen[n]), int(glen[n]) + 1
        lab = np.asarray(labels[n], dtype=np.int64)[:u - 1] if u > 1 else np.zeros((0,), np.int64)
        c, g, _, _ = transduce(log_probs[n, :t, :u, :], lab, blank, fastemit_lambda, fast)
        costs[n] = c
        grads[n, :t, :u, :] = g
    return costs, grads


def log_softmax(x):
    x = np.asarray(x, dtype=np.float64)
    m = x.max(axis=-1, keepdims=True)
    return (x - m) - np.log(np.exp(x - m).sum(axis=-1, keepdims=True))
