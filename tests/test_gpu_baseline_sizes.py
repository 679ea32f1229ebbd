"""Parity at BASELINE.json's full sizes, through the entry point each configuration names.

  c2  N=16, T=150,  U=40,  V=28     rnnt_loss(log_probs)                      + backward
  c3  N=32, T=150,  U=20,  V=5000   rnnt_loss(ops.log_softmax(x), gather=True) + backward
  c4  N=16, T=1500, U=300, V=50     the same (per-rank slice of configs[3]); full and ragged lengths
  c5  N=1,  T=1500, U=300, V=10000  in-place ops.log_softmax, gather=True, fastemit_lambda=0.01 + backward
      (configs[4] is 8 utterances per rank = 144 GB of logits; one utterance = 18 GB exercises the same
      kernels -- k_lsm_large<512,8> in place, k_to_diagonal<true> at V=10000, FastEmit at T=1500 x U=300,
      k_expand_large -- on the test box without spending its time budget on 144 GB of random numbers)

Every case is compared three ways on the gradient PAIRS (blank, label) of all live cells:
  hip <-> fp32 oracle (oracle/rnnt_oracle.c: the reference's operation order, libm expf/log1pf),
  hip <-> fp64 (oracle/transduce_np.py on an fp64 log-softmax of the same logits),
  oracle <-> fp64,
max and 99.9th percentile, and the dense gradient is checked to hold nothing but those pairs.
The bar (BASELINE.json: "within 1e-4 fp32"; VERDICT r1: no additive slack at the headline size): the HIP
path is at most HIP_VS_ORACLE x as far from exact arithmetic as the reference-ordered fp32 oracle is; where
fp32 itself is good to 1e-4 (c2) the absolute bar applies as well.

One arithmetic serves every call since round 6 -- the reference's: fp32 log-sum-exp per cell in its operation order, on
whichever kernel the shape selects (they produce the same bits, tests/test_gpu_wd.py) -- so HIP and oracle are two fp32
implementations of ONE operation order (they differ in the lse transcendentals and in the association of the first
column's prefix sums), and are asserted against each other directly, three ways (round 6: the bars sit on the
measurements, and the ulp argument that explains the maxima on long lattices is itself asserted):
  * max |hip - oracle| and its 99.9th percentile within LOGDOMAIN_VS_ORACLE: 2e-4 / 5e-5 at T = 150 (measured 6e-5 ...
    1.2e-4 / 4e-5), LONG_ULPS = 3 ulp of the largest |cost| / 5e-5 at T = 1500 (c4: 3 x 4.9e-4 = 1.46e-3, measured
    4.5e-4 ... 9.9e-4 / 1.5e-8; c5, |cost| ~ 1.6e4: 5.9e-3, measured 7.9e-4 ... 2.0e-3);
  * the slots further than 1e-4 apart (BASELINE.json's fp32 bar) are at most ABOVE_1E4_FRAC of the live slots ...
  * ... and every one of them sits where the planes are large: max(|alpha|, |beta|, |alpha + beta|) >= 2^10 there (one ulp
    >= 1.2e-4; >= 2^11 on the long lattices), and
    NO live slot is further than MAX_ULP_OF_PLANE ulp of that magnitude from the oracle -- i.e. the 4.5e-4 ... 2e-3 maxima
    are one to two ulp of plane values of 6e3 ... 1.6e4, which is the claim (oracle.grad_error_report);
plus the fp64 bar above.  `test_results_do_not_depend_on_the_batch` states the contract that goes with it: an utterance
gets the same bits whatever batch it is computed in and whichever kernel runs, like the reference's (blockIdx.z = n,
core.cu:49).

With RNNT_PARITY_TABLE=<file.json> every case appends its numbers there (label RNNT_PARITY_BUILD);
profiles/r02_parity_errors.json is the committed copy for the default, log-domain and libm builds.
"""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import transduce_np

pytestmark = pytest.mark.gpu

HIP_VS_ORACLE = 1.5     # max |hip - fp64| <= HIP_VS_ORACLE * max |oracle - fp64|
# |hip - oracle| on the gradients: (max, 99.9th percentile) bars by lattice length.  Measured
# (profiles/r05_parity_errors.json): c2 6.1e-5 / 2.4e-5, c3 1.2e-4 / 4.0e-5 (short lattices: many cells carry a
# visible gradient), c4 4.5e-4 / 1.5e-8, c5 7.9e-4 / 1.0e-7 (long lattices: sharply peaked posteriors, the maxima
# sit on the best path where |alpha| ~ 6e3 makes one ulp 4.9e-4); the full c5 batch (8 ragged utterances) 2.0e-3 /
# 1.4e-5 at |cost| ~ 1.6e4 (one ulp: 2e-3) -- for scale: both are 2.3e-2 / 8.8e-3 away from fp64 there.
# Long lattices: the max bar is LONG_ULPS ulp of the largest |cost| of the batch (c4: 1.46e-3)
LOGDOMAIN_VS_ORACLE = {"short": (2e-4, 5e-5), "long": (None, 5e-5)}
LONG_ULPS = 3.0
ABOVE_1E4_FRAC = 1e-3              # live slots with |hip - oracle| > 1e-4, as a fraction of all live slots (long lattices;
                                   # measured: 3.4e-4 on bench.py's c4 batch, where the 99.9th percentile is 1.3e-5)
MAX_ULP_OF_PLANE = 4.0             # |hip - oracle| <= this many ulp of max(|alpha|, |beta|, |alpha + beta|), every live slot
COST_RTOL_FP64 = 2e-6
COST_RTOL_ORACLE = 1e-5


def dev():
    return torch.device("cuda:0")


def device_case(seed, N, T, U, V, ragged=False):
    """benchmark.py:9-28 on the device: N(0,1) logits, labels in [1,V), full or ragged lengths."""
    g = torch.Generator(device=dev())
    g.manual_seed(seed)
    xs = torch.randn((N, T, U, V), dtype=torch.float32, device=dev(), generator=g)
    ys = torch.randint(1, V, (N, U - 1), dtype=torch.int32, device=dev(), generator=g)
    if ragged:
        rng = np.random.RandomState(seed)
        xn = rng.randint(T // 2, T + 1, size=(N,))
        yn = rng.randint(U // 2, U, size=(N,))
        xn = (xn + T - xn.max()).astype(np.int32)
        yn = (yn + (U - 1) - yn.max()).astype(np.int32)
    else:
        xn = np.full((N,), T, dtype=np.int32)
        yn = np.full((N,), U - 1, dtype=np.int32)
    return xs, ys, xn, yn


def pair_index(ys, U, blank=0):
    """(N,1,U,2) int64 gather index: channel 0 = blank, channel 1 = label of the column (blank on the last)."""
    N = ys.shape[0]
    idx = torch.full((N, 1, U, 2), blank, dtype=torch.int64, device=ys.device)
    idx[:, 0, :U - 1, 1] = ys.long()
    return idx


def take_pairs(dense, ys, blank=0, chunk_frames=64):
    """(N,T,U,V) -> (N,T,U,2) on the device, a few frames at a time (torch is plumbing here)."""
    N, T, U, V = dense.shape
    idx = pair_index(ys, U, blank)
    out = torch.empty((N, T, U, 2), dtype=dense.dtype, device=dense.device)
    for t0 in range(0, T, chunk_frames):
        t1 = min(T, t0 + chunk_frames)
        out[:, t0:t1] = torch.gather(dense[:, t0:t1], 3, idx.expand(N, t1 - t0, U, 2))
    return out


def pairs_fp64(logits, ys, blank=0, chunk_frames=16):
    """(blank,label) pairs of an fp64 log-softmax of the logits, computed chunk by chunk."""
    N, T, U, V = logits.shape
    idx = pair_index(ys, U, blank)
    out = torch.empty((N, T, U, 2), dtype=torch.float64, device=logits.device)
    for t0 in range(0, T, chunk_frames):
        t1 = min(T, t0 + chunk_frames)
        lp = torch.log_softmax(logits[:, t0:t1].double(), dim=-1)
        out[:, t0:t1] = torch.gather(lp, 3, idx.expand(N, t1 - t0, U, 2))
    return out.cpu().numpy()


def live_mask(N, T, U, xn, yn):
    """(N,T,U,2) bool: slots that carry a gradient (channel 1 only where the column has a label)."""
    t = np.arange(T)[None, :, None]
    u = np.arange(U)[None, None, :]
    cell = (t < xn[:, None, None]) & (u <= yn[:, None, None])
    lab = cell & (u < yn[:, None, None])
    return np.stack([cell, lab], axis=-1)


def dist(a, b, mask):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))[mask]
    return {"max": float(d.max()), "p999": float(np.quantile(d, 0.999))}


def record(row):
    path = os.environ.get("RNNT_PARITY_TABLE")
    if not path:
        return
    row = dict(row, build=os.environ.get("RNNT_PARITY_BUILD", "default"))
    rows = []
    if os.path.exists(path):
        with open(path) as f:
            rows = json.load(f)
    key = lambda r: (r["case"], r["build"])
    rows = [r for r in rows if key(r) != key(row)] + [row]
    with open(path, "w") as f:
        json.dump(rows, f, indent=1)


def three_way(name, costs, gpairs, lp2_f32, lp2_f64, xn, yn, lam, fp64_utts=None, abs_bar=None):
    """costs (N,), gpairs (N,T,U,2) from the HIP path; lp2_* the pairs fed to the two CPU legs."""
    N, T, U, _ = gpairs.shape
    ones = np.ones((N, max(U - 1, 1)), dtype=np.int32)[:, :U - 1]
    ref = oracle.rnnt_loss_f32(lp2_f32, None, xn, yn, blank=-1, fastemit_lambda=lam, scan_mode=1)
    assert not ref["mismatch"].any()
    sel = list(range(N)) if fp64_utts is None else list(fp64_utts)
    c64, g64 = transduce_np.transduce_batch(lp2_f64[sel], ones[sel], xn[sel], yn[sel], blank=0,
                                            fastemit_lambda=lam, fast=True)
    mask = live_mask(N, T, U, xn, yn)
    row = {
        "case": name, "N": N, "T": T, "U": U, "fastemit_lambda": lam, "fp64_utterances": len(sel),
        "max_abs_loglik": float(np.abs(c64).max()),
        "grad_hip_vs_oracle": dist(gpairs, ref["grads"], mask),
        "grad_hip_vs_fp64": dist(gpairs[sel], g64, mask[sel]),
        "grad_oracle_vs_fp64": dist(ref["grads"][sel], g64, mask[sel]),
        "cost_rel_hip_vs_fp64": float(np.abs(costs[sel] / c64 - 1).max()),
        "cost_rel_oracle_vs_fp64": float(np.abs(ref["costs"][sel] / c64 - 1).max()),
        "cost_rel_hip_vs_oracle": float(np.abs(costs / ref["costs"] - 1).max()),
    }
    print(json.dumps(row))
    record(row)
    # dead slots are exactly zero
    assert not gpairs[~mask].any()
    np.testing.assert_allclose(costs, ref["costs"], rtol=COST_RTOL_ORACLE)
    np.testing.assert_allclose(costs[sel], c64, rtol=COST_RTOL_FP64)
    assert row["grad_hip_vs_fp64"]["max"] <= HIP_VS_ORACLE * row["grad_oracle_vs_fp64"]["max"], row
    assert row["grad_hip_vs_fp64"]["p999"] <= HIP_VS_ORACLE * row["grad_oracle_vs_fp64"]["p999"], row
    if abs_bar is not None:
        assert row["grad_hip_vs_oracle"]["max"] <= abs_bar, row
    assert_close_to_the_oracle(row, gpairs, ref, xn, yn, T)
    record(row)          # (again: now with the ulp report in it)
    # path-occupancy invariants (exact in exact arithmetic), to the accuracy just established
    # (gradient errors are relative errors of exp(.), so a row/column sum is off by about as much as its
    # largest entry)
    tol = 4 * max(row["grad_hip_vs_fp64"]["max"], 1e-6)
    for n in sel:
        tn, un = int(xn[n]), int(yn[n]) + 1
        np.testing.assert_allclose(gpairs[n, :tn, :un, 0].sum(axis=1, dtype=np.float64), -1.0, atol=tol)
        if un > 1:
            np.testing.assert_allclose(gpairs[n, :tn, :un - 1, 1].sum(axis=0, dtype=np.float64),
                                       -(1 + lam), atol=tol)
    return row


def assert_close_to_the_oracle(row, gpairs, ref, xn, yn, T):
    """The direct bars of the module docstring; adds the ulp report to `row`."""
    rep = oracle.grad_error_report(gpairs, ref, xn, yn)
    row["vs_oracle"] = rep
    print(json.dumps({"case": row["case"], "vs_oracle": rep}))
    long = T >= 640
    bar_max, bar_p999 = LOGDOMAIN_VS_ORACLE["long" if long else "short"]
    if long:
        bar_max = LONG_ULPS * float(np.spacing(np.float32(np.abs(ref["costs"]).max())))
    assert rep["max_abs"] <= bar_max, (rep, bar_max)
    assert rep["p999"] <= bar_p999, rep
    assert rep["max_ulp_of_plane"] <= MAX_ULP_OF_PLANE, rep
    if long:
        assert rep["frac_above"] <= ABOVE_1E4_FRAC, rep
    if rep["cells_above"]:
        # further than 1e-4 from the oracle ONLY where one ulp of the planes is itself larger than 1e-4 (magnitude >= 2^10:
        # ulp 1.2e-4; c3's |log-likelihood| ~ 1.4e3 is just there, c4's 6e3 and c5's 1.6e4 far beyond)
        assert float(np.spacing(np.float32(rep["min_plane_magnitude_above"]))) >= 1e-4, rep
        if long:
            assert rep["min_plane_magnitude_above"] >= 2.0 ** 11, rep


def run_through_wrapper(name, xs, ys, xn, yn, gather, lam, inplace=False, fp64_utts=None, abs_bar=None):
    import warp_rnnt
    from warp_rnnt_amd import ops
    N, T, U, V = xs.shape
    txn, tyn = torch.tensor(xn, device=dev()), torch.tensor(yn, device=dev())
    lp2_64 = pairs_fp64(xs, ys)                      # before the in-place log-softmax overwrites the logits
    lp = ops.log_softmax(xs, out=xs if inplace else None).requires_grad_(True)
    costs = warp_rnnt.rnnt_loss(lp, ys, txn, tyn, gather=gather, fastemit_lambda=lam)
    # non-unit upstream gradient: backward must scale per utterance (__init__.py:23)
    w = torch.linspace(0.5, 1.5, N, device=dev())
    (costs * w).sum().backward()
    torch.cuda.synchronize()
    dense = lp.grad
    assert dense.shape == lp.shape
    gp = take_pairs(dense, ys)
    gp[:, :, U - 1, 1] = 0                            # the last column has no label slot (index = blank)
    # the dense gradient holds nothing but the pairs (labels never equal the blank here)
    assert int(torch.count_nonzero(dense)) == int(torch.count_nonzero(gp))
    gp = (gp / w.view(-1, 1, 1, 1)).cpu().numpy()
    lp2_32 = take_pairs(lp.detach(), ys).cpu().numpy()
    del dense, lp
    return three_way(name, costs.detach().cpu().numpy(), gp, lp2_32, lp2_64, xn, yn, lam,
                     fp64_utts=fp64_utts, abs_bar=abs_bar)


def test_c2_dense_entry():
    xs, ys, xn, yn = device_case(2, 16, 150, 40, 28)
    run_through_wrapper("c2 N=16 T=150 U=40 V=28 gather=False", xs, ys, xn, yn, gather=False, lam=0.0, abs_bar=1e-4)


def test_c3_gather_entry_and_backward():
    xs, ys, xn, yn = device_case(3, 32, 150, 20, 5000)
    run_through_wrapper("c3 N=32 T=150 U=20 V=5000 gather=True", xs, ys, xn, yn, gather=True, lam=0.0)


@pytest.mark.parametrize("ragged", [False, True], ids=["full", "ragged"])
def test_c4_gather_entry_and_backward(ragged):
    xs, ys, xn, yn = device_case(4, 16, 1500, 300, 50, ragged=ragged)
    run_through_wrapper(f"c4 N=16 T=1500 U=300 V=50 gather=True{' ragged' if ragged else ''}", xs, ys, xn, yn,
                        gather=True, lam=0.0, fp64_utts=(0, 5, 10, 15))


def test_c5_per_rank_shape_one_utterance():
    xs, ys, xn, yn = device_case(5, 1, 1500, 300, 10000)
    run_through_wrapper("c5 N=1 T=1500 U=300 V=10000 gather=True fastemit=0.01 in-place", xs, ys, xn, yn,
                        gather=True, lam=0.01, inplace=True)


def _pairs_grads(lp, ys, xn, yn):
    """costs, (N,T,U,2) gradient pairs of rnnt_amd_loss(dense log-probs)."""
    from warp_rnnt_amd import ops
    c, g = ops.loss(lp, ys, torch.tensor(xn, device=dev()), torch.tensor(yn, device=dev()),
                    ops.IN_LOG_PROBS_DENSE, ops.GRADS_GATHERED)
    torch.cuda.synchronize()
    return c.cpu().numpy(), g.cpu().numpy()


def test_results_do_not_depend_on_the_batch():
    """The reference's per-utterance results never depend on N (blockIdx.z = n, core.cu:49).  Neither do these: the
    library picks its lattice KERNEL from the batch shape (one workgroup per 64-column block at N=16 and N=32, T=1500,
    U=300; one per sweep from 2N*ceil(U/64) > 2 x the compute units on), but the kernels share the step function.
    Bit-identical per utterance between a batch and its first half, and between the kernels pinned
    (tests/test_gpu_wd.py does that at more shapes)."""
    from warp_rnnt_amd import debug, ops
    xs, ys, xn, yn = device_case(6, 32, 1500, 300, 50)
    lp2_64 = pairs_fp64(xs[:2], ys[:2])
    lp = ops.log_softmax(xs, out=xs)
    half = slice(0, 16)
    ca, ga = _pairs_grads(lp, ys, xn, yn)
    c16, g16 = _pairs_grads(lp[half].contiguous(), ys[half].contiguous(), xn[half], yn[half])
    np.testing.assert_array_equal(ca[half], c16)
    np.testing.assert_array_equal(ga[half], g16)
    for kernel in ("ws", "wd"):
        with debug.lattice_kernel(kernel):
            ck, gk = _pairs_grads(lp, ys, xn, yn)
        np.testing.assert_array_equal(ck, ca, err_msg=kernel)
        np.testing.assert_array_equal(gk, ga, err_msg=kernel)
    ones = np.ones((2, 299), dtype=np.int32)
    c64, g64 = transduce_np.transduce_batch(lp2_64, ones, xn[:2], yn[:2], blank=0, fast=True)
    np.testing.assert_allclose(ca[:2], c64, rtol=COST_RTOL_FP64)


def test_c5_full_per_rank_batch_forward():
    """BASELINE.json configs[4] at its per-rank size: 8 utterances of T=1500, U=300, V=10000 = 144 GB of logits
    generated on the device, log-softmax IN PLACE (a second 144 GB tensor would not fit), gather=True,
    fastemit_lambda=0.01, forward only -- no dense (N,T,U,V) gradient exists at this size; the gradient that does
    exist is the (N,T,U,2) gathered one, which the native op returns.  Checked: costs through
    warp_rnnt.rnnt_loss against the fp32 oracle on pairs extracted chunk by chunk from the same log-probs and, for
    one utterance, against fp64 on an fp64 log-softmax of the logits; the gathered gradients against the oracle
    and fp64 with the module's bars, and the path-occupancy invariants on every utterance."""
    import warp_rnnt
    from warp_rnnt_amd import ops
    N, T, U, V, lam = 8, 1500, 300, 10000, 0.01
    torch.cuda.empty_cache()                              # (blocks cached by earlier tests count as used)
    free, _ = torch.cuda.mem_get_info(dev())
    need = N * T * U * V * 4
    if free < need + 40e9:
        pytest.skip(f"{free / 1e9:.0f} GB free on the device, the case needs {need / 1e9:.0f} GB + scratch")
    xs, ys, xn, yn = device_case(55, N, T, U, V, ragged=True)
    txn, tyn = torch.tensor(xn, device=dev()), torch.tensor(yn, device=dev())
    lp2_64 = pairs_fp64(xs[:1], ys[:1])                   # before the logits are overwritten
    lp = ops.log_softmax(xs, out=xs)
    assert lp.data_ptr() == xs.data_ptr()
    costs = warp_rnnt.rnnt_loss(lp, ys, txn, tyn, gather=True, fastemit_lambda=lam).cpu().numpy()
    lp2_32 = take_pairs(lp, ys, chunk_frames=16).cpu().numpy()
    ref = oracle.rnnt_loss_f32(lp2_32, None, xn, yn, blank=-1, fastemit_lambda=lam, scan_mode=1)
    assert not ref["mismatch"].any()
    np.testing.assert_allclose(costs, ref["costs"], rtol=COST_RTOL_ORACLE)
    ones = np.ones((1, U - 1), dtype=np.int32)
    c64, g64 = transduce_np.transduce_batch(lp2_64, ones, xn[:1], yn[:1], blank=0, fastemit_lambda=lam, fast=True)
    np.testing.assert_allclose(costs[:1], c64, rtol=COST_RTOL_FP64)
    mask = live_mask(N, T, U, xn, yn)
    c2, g2 = ops.loss(lp, ys, txn, tyn, ops.IN_LOG_PROBS_DENSE, ops.GRADS_GATHERED, 0, lam)
    torch.cuda.synchronize()
    g2 = g2.cpu().numpy()
    np.testing.assert_allclose(c2.cpu().numpy(), ref["costs"], rtol=COST_RTOL_ORACLE)
    assert not g2[~mask].any()
    row = {"case": "c5 N=8 T=1500 U=300 V=10000 gather=True fastemit=0.01 in-place, forward",
           "N": N, "T": T, "U": U, "fastemit_lambda": lam, "fp64_utterances": 1,
           "grad_hip_vs_oracle": dist(g2, ref["grads"], mask),
           "grad_hip_vs_fp64": dist(g2[:1], g64, mask[:1]),
           "grad_oracle_vs_fp64": dist(ref["grads"][:1], g64, mask[:1])}
    print(json.dumps(row))
    assert row["grad_hip_vs_fp64"]["max"] <= HIP_VS_ORACLE * row["grad_oracle_vs_fp64"]["max"], row
    assert_close_to_the_oracle(row, g2, ref, xn, yn, T)
    record(row)
    tol = 4 * max(row["grad_hip_vs_fp64"]["max"], 1e-6)     # (the accuracy just established on utterance 0)
    for n in range(N):
        tn, un = int(xn[n]), int(yn[n]) + 1
        np.testing.assert_allclose(g2[n, :tn, :un, 0].sum(axis=1, dtype=np.float64), -1.0, atol=tol)
        np.testing.assert_allclose(g2[n, :tn, :un - 1, 1].sum(axis=0, dtype=np.float64), -(1 + lam), atol=tol)
    del xs, lp
    torch.cuda.empty_cache()                              # give the 144 GB back before the next test
