"""CPU check of the probability-domain lattice scheme (tests/pd_model.py = the arithmetic of
csrc/lattice_pd.hip) against the fp64 oracle and the reference-ordered fp32 oracle.

What this pins without a GPU: the per-column exponent bookkeeping (renormalisation every K diagonals,
columns that start late, ragged ends), that nothing accumulates along the sweep (the error of the stored
log-values stays at the final rounding, where the log-domain chain drifts), and that inputs it cannot
represent are recognised.
"""
import numpy as np
import pytest

import oracle
from oracle import transduce_np
import pd_model as pm
from helpers import reference_cases, np_log_softmax32


def _pairs(T, U, V, seed, scale=1.0):
    rng = np.random.RandomState(seed)
    if V > 1000:    # the two channels of a large-vocabulary log-softmax, without building the other V-2
        return (rng.randn(T, U, 2) * scale - np.log(V) - 0.5 * scale * scale).astype(np.float32)
    lp = transduce_np.log_softmax(rng.randn(T, U, V) * scale)
    lab = rng.randint(1, V, max(U - 1, 0))
    lp2 = np.zeros((T, U, 2))
    lp2[..., 0] = lp[..., 0]
    lp2[:, :U - 1, 1] = lp[:, np.arange(U - 1), lab]
    return lp2.astype(np.float32)


@pytest.mark.parametrize("T,U,V,scale", [(5, 4, 3, 1.0), (40, 12, 7, 1.0), (150, 40, 28, 1.0), (33, 130, 9, 1.0),
                                         (300, 100, 50, 1.0), (200, 70, 5000, 1.0), (60, 90, 6, 3.0), (1, 9, 4, 1.0),
                                         (9, 2, 4, 5.0)])
def test_model_tracks_fp64_closer_than_the_log_domain_oracle(T, U, V, scale):
    lp2 = _pairs(T, U, V, 7, scale)
    al, be, ll, ok = pm.lattice(lp2)
    assert ok
    _, _, a64, b64 = transduce_np.transduce(lp2.astype(np.float64), np.ones(U - 1, int), 0, 0.0, True)
    ref = oracle.rnnt_loss_f32(lp2[None], None, np.array([T], np.int32), np.array([U - 1], np.int32), blank=-1,
                               scan_mode=1)
    # one fp32 rounding of the result (0.5 ulp) + 2^-23 from the 24-bit mantissa + the fp32 exp2 of every factor
    ulp = np.spacing(np.abs(a64).max().astype(np.float32))
    bound = 0.75 * ulp + 4e-7 * np.sqrt(T + U) + 2e-7
    assert np.abs(al - a64).max() <= bound
    assert np.abs(be - b64).max() <= bound
    assert abs(ll - b64[0, 0]) <= bound
    if T + U > 100:   # the log-domain chain has drifted by then
        assert np.abs(al - a64).max() < np.abs(ref["alphas"][0] - a64).max()


def test_model_reproduces_the_reference_golden_vectors():
    """costs and gradient pairs of test.py:34-188 from the model's alphas/betas through k_grads' formula."""
    for case in reference_cases():
        lp = np_log_softmax32(np.array(case["logits"], dtype=np.float32))
        N, T, U, V = lp.shape
        labels = np.array(case["labels"], dtype=np.int32).reshape(N, U - 1)
        lp2 = oracle.gather_f32(lp, labels, case["blank"])
        for n in range(N):
            t, u = case["xn"][n], case["yn"][n] + 1
            x = lp2[n, :t, :u]
            al, be, ll, ok = pm.lattice(x)
            assert ok
            np.testing.assert_allclose(-be[0, 0], case["costs"][n], atol=1.5e-6)
            np.testing.assert_allclose(ll, be[0, 0], atol=2e-6)
            g = np.array(case["grads"])[n]
            gB = -np.exp((al[:t - 1] + be[1:]) + x[:t - 1, :, 0] - be[0, 0]) if t > 1 else np.zeros((0, u))
            want = g[:t - 1, :u, 0] if case["layout"] == "gathered" else g[:t - 1, :u, case["blank"]]
            np.testing.assert_allclose(gB, want, atol=1.5e-6)


def test_columns_may_differ_by_hundreds_of_binary_orders():
    """Sharp logits on a lattice wider than long: neighbouring columns differ by 2^80 and more.  fp64 state with
    per-column exponents carries that (an fp32 state does not, whatever the renormalisation interval)."""
    lp2 = _pairs(60, 90, 6, 1, 3.0)
    al, be, ll, ok = pm.lattice(lp2)
    _, _, a64, b64 = transduce_np.transduce(lp2.astype(np.float64), np.ones(89, int), 0, 0.0, True)
    gap = max(np.abs(np.diff([a64[d - u, u] for u in range(max(0, d - 59), min(89, d) + 1)])).max()
              for d in range(2, 148))
    assert gap / np.log(2) > 60
    assert ok and np.abs(al - a64).max() < 5e-5 and np.abs(be - b64).max() < 5e-5


def test_input_check():
    lp2 = _pairs(20, 6, 5, 3)
    assert pm.in_range(lp2)
    bad = lp2.copy(); bad[3, 2, 1] = -95.0
    assert not pm.in_range(bad)
    bad = lp2.copy(); bad[0, 0, 0] = -np.inf
    assert not pm.in_range(bad)
    ok = lp2.copy(); ok[:, -1, 1] = np.nan          # label channel of the last column: not part of the lattice
    assert pm.in_range(ok)


def _step_case(T, U, t_step, lo, hi):
    """Label log-probs `lo` up to frame t_step and `hi` after it, blank = the rest of the mass: every log-prob is
    well inside the input range, yet neighbouring columns of a diagonal that crosses the step differ by
    exp((hi-lo)*u) (2^1558 at u = 120 for -10 -> -1)."""
    lp2 = np.zeros((T, U, 2), np.float32)
    lab = np.where(np.arange(T) <= t_step, lo, hi).astype(np.float64)
    lp2[..., 1] = lab[:, None]
    lp2[..., 0] = np.log1p(-np.exp(lab))[:, None]
    return lp2


@pytest.mark.parametrize("T,U,t_step,lo,hi", [(300, 120, 150, -10.0, -1.0), (60, 24, 30, -70.0, -1e-3)])
def test_column_gap_beyond_fp64_is_flagged_not_silently_wrong(T, U, t_step, lo, hi):
    """The per-column exponents may drift apart without bound while every input passes the range check; the factor
    2**(E_left - E_own) must then not be trusted.  The model (as the kernel) flags the sweep, and the flagged result
    is indeed wrong -- which is why the log-domain kernel redoes it."""
    lp2 = _step_case(T, U, t_step, lo, hi)
    assert pm.in_range(lp2)
    al, be, ll, ok = pm.lattice(lp2)
    assert not ok
    _, _, a64, b64 = transduce_np.transduce(lp2.astype(np.float64), np.ones(U - 1, int), 0, 0.0, True)
    assert np.isfinite(b64[0, 0])


@pytest.mark.parametrize("hi", [-6.0, -4.0])
def test_moderate_steps_are_carried_and_exact(hi):
    """A step that keeps the column gap inside MAX_GAP is carried: same accuracy as everywhere else."""
    T, U = 200, 60
    lp2 = _step_case(T, U, 100, -8.0, hi)
    al, be, ll, ok = pm.lattice(lp2)
    assert ok
    _, _, a64, b64 = transduce_np.transduce(lp2.astype(np.float64), np.ones(U - 1, int), 0, 0.0, True)
    ulp = np.spacing(np.abs(a64).max().astype(np.float32))
    bound = 0.75 * ulp + 4e-7 * np.sqrt(T + U) + 2e-7
    assert np.abs(al - a64).max() <= bound and np.abs(be - b64).max() <= bound


def test_nan_in_a_live_cell_is_flagged_by_the_chain():
    """v_min3/v_max3 of the loader's full-block path skip a NaN operand; the NaN is sticky in the chain and the
    final-value check catches it."""
    lp2 = _pairs(40, 12, 7, 5)
    assert pm.sweep(lp2[..., 0], lp2[..., 1])[2] and pm.sweep(lp2[..., 0], lp2[..., 1], beta=True)[2]
    for ch in (0, 1):
        bad = lp2.copy(); bad[20, 5, ch] = np.nan
        assert not pm.sweep(bad[..., 0], bad[..., 1])[2]
        assert not pm.sweep(bad[..., 0], bad[..., 1], beta=True)[2]


@pytest.mark.parametrize("seed", range(12))
def test_random_lattices_kept_or_flagged_never_silently_wrong(seed):
    """Random shapes, scales and step inputs: whatever the model KEEPS (ok) is within the accuracy bound of fp64;
    whatever it cannot carry is flagged.  (The property the GPU kernel is tested for in tests/pd_vs_oracle.py, on
    inputs the CPU can sweep in milliseconds.)"""
    rng = np.random.RandomState(1000 + seed)
    T, U = int(rng.randint(1, 70)), int(rng.randint(2, 70))
    kind = seed % 3
    if kind == 0:
        lp2 = _pairs(T, U, int(rng.choice([3, 9, 40])), seed, float(rng.choice([0.5, 2.0, 6.0])))
    elif kind == 1:
        lp2 = _step_case(T, U, int(rng.randint(0, T)), float(rng.uniform(-30, -5)), float(rng.uniform(-3, -0.01)))
    else:
        lp2 = _step_case(T, U, int(rng.randint(0, T)), float(rng.uniform(-3, -0.01)), float(rng.uniform(-30, -5)))
    al, be, ll, ok = pm.lattice(lp2)
    _, _, a64, b64 = transduce_np.transduce(lp2.astype(np.float64), np.ones(U - 1, int), 0, 0.0, True)
    if ok:
        ulp = np.spacing(max(np.abs(a64).max(), np.abs(b64).max()).astype(np.float32))
        bound = 0.75 * ulp + 4e-7 * np.sqrt(T + U) + 2e-7
        assert np.abs(al - a64).max() <= bound and np.abs(be - b64).max() <= bound
        assert abs(ll - b64[0, 0]) <= bound
    else:
        assert np.isfinite(b64[0, 0])          # a valid input: the log-domain kernel handles it
