"""Pins the CPU oracle (oracle/) against the reference's own golden vectors.

CPU-only.  The vectors are the literals of the reference's unit tests
(pytorch_binding/warp_rnnt/test.py), see tests/golden/reference_vectors.json.
Tolerance = the reference's own: decimal=6 (|d| < 1.5e-6).
"""
import numpy as np
import pytest

import oracle
from oracle import transduce_np
from helpers import reference_cases, np_log_softmax32, make_case

TOL = 1.5e-6


def _prep(case):
    logits = np.array(case["logits"], dtype=np.float32)
    lp = np_log_softmax32(logits)
    N, T, U, V = lp.shape
    labels = np.array(case["labels"], dtype=np.int32).reshape(N, U - 1)
    xn = np.array(case["xn"], dtype=np.int32)
    yn = np.array(case["yn"], dtype=np.int32)
    return lp, labels, xn, yn


@pytest.mark.parametrize("scan_mode", [0, 1])
@pytest.mark.parametrize("case", reference_cases(), ids=lambda c: c["name"])
def test_c_oracle_matches_reference_golden(case, scan_mode):
    lp, labels, xn, yn = _prep(case)
    blank = case["blank"]
    if case["layout"] == "gathered":
        lp = oracle.gather_f32(lp, labels, blank)
        blank = -1
    out = oracle.rnnt_loss_f32(lp, labels, xn, yn, blank=blank, scan_mode=scan_mode)
    np.testing.assert_allclose(out["costs"], np.array(case["costs"]), atol=TOL, rtol=0)
    np.testing.assert_allclose(out["grads"], np.array(case["grads"]), atol=TOL, rtol=0)
    assert not out["mismatch"].any()


@pytest.mark.parametrize("case", reference_cases(("dense",)), ids=lambda c: c["name"])
def test_numpy_fp64_matches_reference_golden(case):
    lp, labels, xn, yn = _prep(case)
    costs, grads = transduce_np.transduce_batch(lp, labels, xn, yn, blank=case["blank"])
    np.testing.assert_allclose(costs, np.array(case["costs"]), atol=TOL, rtol=0)
    np.testing.assert_allclose(grads, np.array(case["grads"]), atol=TOL, rtol=0)


def test_compact_golden_is_dense_golden_repacked():
    """The compact golden rows (test.py:259-336) are the dense forward_batch grads packed ragged."""
    from helpers import reference_doc
    doc = reference_doc()
    comp = [c for c in doc["cases"] if c["name"] == "forward_batch_compact"][0]
    lp, labels, xn, yn = _prep(comp)
    out = oracle.rnnt_loss_f32(lp, labels, xn, yn, blank=0)
    rows = np.concatenate([out["grads"][n, :xn[n], :yn[n] + 1].reshape(-1, lp.shape[-1])
                           for n in range(lp.shape[0])])
    np.testing.assert_allclose(rows, np.array(comp["grads_rows"]), atol=TOL, rtol=0)


def test_gather_matches_wrapper_rule():
    logits, labels, xn, yn = make_case(3, 2, 5, 4, 7)
    lp = np_log_softmax32(logits)
    g = oracle.gather_f32(lp, labels, blank=0)
    idx = np.zeros((2, 5, 4, 2), dtype=np.int64)
    idx[:, :, :3, 1] = labels[:, None, :]
    np.testing.assert_array_equal(g, np.take_along_axis(lp, idx, axis=3))


@pytest.mark.parametrize("ragged", [False, True])
def test_fast_numpy_sweeps_equal_loops(ragged):
    logits, labels, xn, yn = make_case(5, 3, 9, 6, 5, ragged=ragged)
    lp = transduce_np.log_softmax(logits)
    c0, g0 = transduce_np.transduce_batch(lp, labels, xn, yn, fast=False, fastemit_lambda=0.01)
    c1, g1 = transduce_np.transduce_batch(lp, labels, xn, yn, fast=True, fastemit_lambda=0.01)
    np.testing.assert_allclose(c0, c1, rtol=1e-13)
    np.testing.assert_allclose(g0, g1, atol=1e-13)


@pytest.mark.parametrize("N,T,U,V,ragged,lam,blank", [
    (1, 150, 40, 28, False, 0.0, 0),      # BASELINE config 1 (CPU plumbing case)
    (4, 60, 25, 11, True, 0.01, 0),
    (3, 33, 17, 6, True, 0.0, 3),
    (2, 70, 1, 4, False, 0.0, 0),         # U == 1 (no labels)
    (2, 1, 9, 4, False, 0.25, 1),         # T == 1
])
def test_c_oracle_fp32_close_to_fp64(N, T, U, V, ragged, lam, blank):
    """fp32 restatement vs exact arithmetic on seeded cases (incl. the unpinned flags)."""
    logits, labels, xn, yn = make_case(11, N, T, U, V, ragged=ragged, blank=blank)
    lp = np_log_softmax32(logits)
    ref_c, ref_g = transduce_np.transduce_batch(lp, labels, xn, yn, blank=blank,
                                                fastemit_lambda=lam, fast=True)
    for scan_mode in (0, 1):
        out = oracle.rnnt_loss_f32(lp, labels, xn, yn, blank=blank, fastemit_lambda=lam,
                                   scan_mode=scan_mode)
        np.testing.assert_allclose(out["costs"], ref_c, rtol=2e-6)
        np.testing.assert_allclose(out["grads"], ref_g, atol=2e-4)
        assert not out["mismatch"].any()
    # gathered layout gives the gathered view of the same numbers
    lp2 = oracle.gather_f32(lp, labels, blank)
    out2 = oracle.rnnt_loss_f32(lp2, labels, xn, yn, blank=-1, fastemit_lambda=lam, scan_mode=1)
    np.testing.assert_allclose(out2["costs"], out["costs"], rtol=1e-6)
    idx = np.full((N, T, U, 2), blank, dtype=np.int64)
    if U > 1:
        idx[:, :, :U - 1, 1] = labels[:, None, :]
    dense_from_g = np.zeros_like(lp)
    # label grads live in ch1 for u<U-1; blank grads in ch0
    np.put_along_axis(dense_from_g, idx[..., :1], out2["grads"][..., :1], axis=3)
    lab_part = np.zeros_like(lp)
    if U > 1:
        np.put_along_axis(lab_part[:, :, :U - 1], idx[:, :, :U - 1, 1:], out2["grads"][:, :, :U - 1, 1:], axis=3)
    np.testing.assert_allclose(dense_from_g + lab_part, out["grads"], atol=1e-6)


def test_path_occupancy_invariants():
    """Every path crosses each frame once by a blank and emits each label once:
    sum_u gB[t,u] = -1 for all t, sum_t gL[t,u] = -(1+lambda) for all u."""
    lam = 0.05
    logits, labels, xn, yn = make_case(2, 2, 40, 12, 9)
    lp = oracle.gather_f32(np_log_softmax32(logits), labels, 0)
    out = oracle.rnnt_loss_f32(lp, labels, xn, yn, blank=-1, fastemit_lambda=lam)
    g = out["grads"]
    np.testing.assert_allclose(g[..., 0].sum(axis=2), -1.0, atol=1e-4)
    np.testing.assert_allclose(g[:, :, :-1, 1].sum(axis=1), -(1 + lam), atol=1e-4)


def test_mismatch_guard_silent_on_consistent_input():
    """The forward/backward consistency guard (core_gather.cu:341-354) must stay
    silent on sane input (alpha corner and beta[0,0] agree to ~1e-6 relative)."""
    logits, labels, xn, yn = make_case(1, 2, 10, 5, 4)
    lp = np_log_softmax32(logits)
    out = oracle.rnnt_loss_f32(lp, labels, xn, yn)
    assert out["mismatch"].sum() == 0
    a = out["alphas"][:, -1, -1] + lp[:, -1, -1, 0]
    np.testing.assert_allclose(a, out["betas"][:, 0, 0], rtol=1e-5)


def test_log_softmax_oracle():
    x = np.random.RandomState(0).randn(37, 50).astype(np.float32) * 3
    np.testing.assert_allclose(oracle.log_softmax_f32(x), transduce_np.log_softmax(x), atol=2e-6)


def test_masked_log_probs_follow_the_reference_rim_and_interior_rules():
    """What the reference does with -inf log-probs, restated: the rim of the lattice is plain sums (core_gather.cu:76-104,
    177-205), so a masked label in row 0 / the last row, or a masked blank in column 0 / the last column, leaves -inf
    there and a FINITE cost (the other paths carry it); the interior is log_sum_exp (core_gather.cu:22-35), where two
    -inf operands give -inf - -inf = NaN, so a frame nobody can cross gives a NaN cost.  The GPU tests hold the HIP
    kernels to this pattern (tests/test_gpu_parity.py)."""
    np.seterr(all="ignore")
    rng = np.random.RandomState(3)
    N, T, U = 3, 7, 5
    lp2 = np_log_softmax32(rng.randn(N, T, U, 2))
    labels = np.ones((N, U - 1), dtype=np.int32)
    xn, yn = np.full((N,), T, np.int32), np.full((N,), U - 1, np.int32)
    clean = oracle.rnnt_loss_f32(lp2, labels, xn, yn, blank=-1, scan_mode=1)
    lp2[0, 0, 1, 1] = -np.inf            # rim: label out of (0,1)
    lp2[0, T - 1, 2, 1] = -np.inf        # rim: label out of (T-1,2)
    lp2[1, 2, 0, 0] = -np.inf            # rim: blank out of (2,0)
    lp2[1, 3, U - 1, 0] = -np.inf        # rim: blank out of (3,U-1)
    lp2[2, 3] = -np.inf                  # a frame without an exit
    for scan_mode in (0, 1):
        r = oracle.rnnt_loss_f32(lp2, labels, xn, yn, blank=-1, scan_mode=scan_mode)
        assert np.isfinite(r["costs"][:2]).all() and np.isnan(r["costs"][2])
        assert (r["costs"][:2] > clean["costs"][:2]).all()            # paths were removed, none added
        assert not np.isnan(r["grads"][:2]).any()
        assert np.isneginf(r["alphas"][0, 0, 2:]).all() and np.isfinite(r["alphas"][0, 1:, 2:]).all()
        assert np.isneginf(r["alphas"][1, 3:, 0]).all() and np.isfinite(r["alphas"][1, 3:, 1:]).all()
        assert r["grads"][0, 0, 1, 1] == 0 and r["grads"][1, 2, 0, 0] == 0   # exp(-inf): no gradient through a masked arc
        assert np.isnan(r["alphas"][2, 4, 1:]).all() and np.isneginf(r["alphas"][2, 4, 0])
