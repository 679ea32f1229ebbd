"""GPU tests of the compact (ragged packed) layout: rnnt_loss(compact=True), _C.rnnt_loss_compact,
_C.rnnt_loss_compact_backward (reference: core_compact.cu, binding.cpp:108-247, __init__.py:26-54)."""
import os

import numpy as np
import pytest
import torch

import oracle
from helpers import GOLDEN, make_case, np_log_softmax32, reference_doc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    a = np.asarray(a)
    return torch.tensor(a if a.ndim == 0 else np.ascontiguousarray(a), device=DEV)


def pack(lp, labels, xn, yn):
    V = lp.shape[-1]
    xs = np.concatenate([lp[n, :xn[n], :yn[n] + 1].reshape(-1, V) for n in range(lp.shape[0])])
    ys = np.concatenate([labels[n, :yn[n]] for n in range(lp.shape[0])]).astype(np.int32)
    return np.ascontiguousarray(xs), ys


def test_reference_golden_compact():
    """test.py:259-336: costs and the (15,5) scattered gradient rows."""
    import warp_rnnt._C as core
    case = [c for c in reference_doc()["cases"] if c["name"] == "forward_batch_compact"][0]
    lp = np_log_softmax32(np.array(case["logits"], dtype=np.float32))
    labels = np.array(case["labels"], dtype=np.int32)
    xn = np.array(case["xn"], dtype=np.int32)
    yn = np.array(case["yn"], dtype=np.int32)
    xs, ys = pack(lp, labels, xn, yn)
    costs, grads, loc = core.rnnt_loss_compact(T(xs), T(ys), T(xn), T(yn))
    np.testing.assert_allclose(costs.cpu().numpy(), np.array(case["costs"]), atol=1.5e-6, rtol=0)
    cumlen = torch.cumsum(T(xn) * (T(yn) + 1), dim=0, dtype=torch.int32)
    dense = core.rnnt_loss_compact_backward(torch.ones_like(costs).contiguous(), grads, cumlen, loc,
                                            lp.shape[-1], 0)
    np.testing.assert_allclose(dense.cpu().numpy(), np.array(case["grads_rows"]), atol=1.5e-6, rtol=0)


@pytest.mark.parametrize("N,Tm,Um,V,lam,blank", [
    (5, 40, 12, 9, 0.0, 0),
    (3, 70, 90, 5, 0.02, 2),      # two waves, ragged
    (4, 9, 1, 4, 0.0, 0),         # no labels at all
    (2, 33, 140, 6, 0.0, 1),      # V % 4 != 0 -> scalar scatter path
])
def test_compact_vs_oracle(N, Tm, Um, V, lam, blank):
    import warp_rnnt._C as core
    logits, labels, xn, yn = make_case(77 + Tm, N, Tm, Um, V, ragged=True, blank=blank)
    lp = np_log_softmax32(logits)
    ref = oracle.rnnt_loss_f32(lp, labels, xn, yn, blank=blank, fastemit_lambda=lam, scan_mode=1)
    xs, ys = pack(lp, labels, xn, yn)
    costs, grads, loc = core.rnnt_loss_compact(T(xs), T(ys), T(xn), T(yn), blank=blank, fastemit_lambda=lam)
    np.testing.assert_allclose(costs.cpu().numpy(), ref["costs"], rtol=1e-5)
    # loc: label index per row, blank on each utterance's last column
    want_loc = []
    for n in range(N):
        l = np.full((xn[n], yn[n] + 1), blank, dtype=np.int64)
        l[:, :yn[n]] = labels[n, :yn[n]][None, :]
        want_loc.append(l.reshape(-1))
    np.testing.assert_array_equal(loc.cpu().numpy(), np.concatenate(want_loc))
    gc = np.random.RandomState(1).rand(N).astype(np.float32) + 0.5
    cumlen = torch.cumsum(T(xn) * (T(yn) + 1), dim=0, dtype=torch.int32)
    dense = core.rnnt_loss_compact_backward(T(gc), grads, cumlen, loc, V, blank).cpu().numpy()
    want = np.concatenate([(ref["grads"][n, :xn[n], :yn[n] + 1] * gc[n]).reshape(-1, V) for n in range(N)])
    np.testing.assert_allclose(dense, want, atol=1e-4)
    # costs-only mode
    c2, g2, _ = core.rnnt_loss_compact(T(xs), T(ys), T(xn), T(yn), blank=blank, fastemit_lambda=lam,
                                       required_grad=False)
    np.testing.assert_array_equal(c2.cpu().numpy(), costs.cpu().numpy())
    assert g2.numel() == 0


def test_compact_wrapper_fixtures_from_reference_wrapper():
    import warp_rnnt
    fx = np.load(os.path.join(GOLDEN, "wrapper_fixtures.npz"))
    xn, yn = T(fx["xn"]), T(fx["yn"])
    for row in fx["compact_cases"]:
        key, blank, reduction, avg, lam = row.split(";")
        lp = T(fx[key + "_xs"]).requires_grad_(True)
        loss = warp_rnnt.rnnt_loss(lp, T(fx[key + "_ys"]), xn, yn, average_frames=bool(int(avg)),
                                   reduction=reduction, blank=int(blank), fastemit_lambda=float(lam),
                                   compact=True)
        np.testing.assert_allclose(loss.detach().cpu().numpy(), fx[key + "_loss"], rtol=1e-5, err_msg=row)
        loss.backward(T(fx[key + "_up"]))
        np.testing.assert_allclose(lp.grad.cpu().numpy(), fx[key + "_grad"], atol=2e-6, err_msg=row)


def test_compact_shape_errors():
    import warp_rnnt._C as core
    xs = torch.zeros((10, 4), device=DEV)
    ys = torch.zeros((3,), dtype=torch.int, device=DEV)
    xn = torch.tensor([2, 2], dtype=torch.int, device=DEV)
    yn = torch.tensor([1, 1], dtype=torch.int, device=DEV)
    with pytest.raises(RuntimeError, match="xs must have 2 dimensions"):
        core.rnnt_loss_compact(xs.view(5, 2, 4), ys, xn, yn)
    with pytest.raises(RuntimeError, match=r"ys shape must be equal to \(sum\(yn\), \)"):
        core.rnnt_loss_compact(xs, ys, xn, yn)
    with pytest.raises(RuntimeError, match="xs shape mismatch"):
        core.rnnt_loss_compact(xs, ys[:2].contiguous(), xn, yn)


def test_compact_empty_utterance_is_contained():
    """An utterance with xn = 0 owns no rows of the packing (the reference would index row -1,
    core_compact.cu): its cost is NaN, every other utterance is unaffected, nothing is written
    outside the other utterances' cells."""
    import warp_rnnt._C as core
    N, Tm, Um, V = 4, 21, 7, 6
    logits, labels, xn, yn = make_case(5, N, Tm, Um, V, ragged=True)
    lp = np_log_softmax32(logits)
    ref = oracle.rnnt_loss_f32(lp, labels, xn, yn, blank=0, scan_mode=1)
    keep = [0, 2, 3]
    xn_bad = xn.copy()
    xn_bad[1] = 0
    V_ = lp.shape[-1]
    xs = np.concatenate([lp[n, :xn[n], :yn[n] + 1].reshape(-1, V_) for n in keep])
    ys = np.concatenate([labels[n, :yn[n]] for n in range(N)]).astype(np.int32)   # labels of n=1 stay packed
    costs, grads, loc = core.rnnt_loss_compact(T(np.ascontiguousarray(xs)), T(ys), T(xn_bad), T(yn))
    c = costs.cpu().numpy()
    assert np.isnan(c[1])
    np.testing.assert_allclose(c[keep], ref["costs"][keep], rtol=1e-5)
    cumlen = torch.cumsum(T(xn_bad) * (T(yn) + 1), dim=0, dtype=torch.int32)
    w = torch.ones(N, device=DEV)
    dense = core.rnnt_loss_compact_backward(w, grads, cumlen, loc, V_, 0).cpu().numpy()
    want = np.concatenate([ref["grads"][n, :xn[n], :yn[n] + 1].reshape(-1, V_) for n in keep])
    np.testing.assert_allclose(dense, want, atol=1e-5)


@pytest.mark.parametrize("N", [1, 7, 1024, 1025, 40000])
def test_compact_offsets_kernel(N):
    """rnnt_amd_compact_offsets: prefix sums + maxima in one launch."""
    import warp_rnnt_amd
    L = warp_rnnt_amd.load()
    rng = np.random.RandomState(N)
    xn = rng.randint(1, 3000, (N,)).astype(np.int32)
    yn = rng.randint(0, 700, (N,)).astype(np.int32)
    offs = torch.empty((N + 1,), dtype=torch.int64, device=DEV)
    loffs = torch.empty((N + 1,), dtype=torch.int32, device=DEV)
    stats = torch.empty((4,), dtype=torch.int64, device=DEV)
    txn, tyn = T(xn), T(yn)      # keep the device copies alive across the call
    st = L.rnnt_amd_compact_offsets(torch.cuda.current_stream().cuda_stream, txn.data_ptr(), tyn.data_ptr(), N,
                                    offs.data_ptr(), loffs.data_ptr(), stats.data_ptr())
    assert st == 0
    cells = xn.astype(np.int64) * (yn.astype(np.int64) + 1)
    np.testing.assert_array_equal(offs.cpu().numpy(), np.concatenate([[0], np.cumsum(cells)]))
    np.testing.assert_array_equal(loffs.cpu().numpy(), np.concatenate([[0], np.cumsum(yn)]).astype(np.int32))
    assert stats.tolist() == [int(cells.sum()), int(yn.sum()), int(xn.max()), int(yn.max())]


def _ref_compact_abi(xs, ys, xn, yn, blank, lam, required_grad=True):
    """The call sequence of the reference's binding (binding.cpp:139-207, 209-247) on the reference-named entry
    points of the library (core.h:41-60), raw device pointers through ctypes, NULL stream."""
    import warp_rnnt_amd
    L = warp_rnnt_amd.load()
    N, V = len(xn), xs.shape[1]
    cells = (xn * (yn + 1)).astype(np.int64)
    STU = int(cells.sum())
    mem_pref = np.concatenate([[0], np.cumsum(cells)[:-1]]).astype(np.int32)       # exclusive, as binding.cpp:147-160
    lab_pref = np.concatenate([[0], np.cumsum(yn)[:-1]]).astype(np.int32)
    txs, tys, txn, tyn = T(xs), T(ys), T(xn), T(yn)
    tmem, tlab = T(mem_pref), T(lab_pref)
    gather_xs = torch.empty((STU, 2), device=DEV)
    loc = torch.zeros((STU,), dtype=torch.int64, device=DEV)
    torch.cuda.synchronize()          # the entry points use the NULL stream, like the reference
    L.run_gather_for_compact(txs.data_ptr(), tys.data_ptr(), txn.data_ptr(), tyn.data_ptr(), gather_xs.data_ptr(),
                             loc.data_ptr(), tmem.data_ptr(), tlab.data_ptr(), N, int(xn.max()), int(yn.max()) + 1, V,
                             blank)
    costs = torch.empty((N,), device=DEV)
    counts = torch.zeros((int(ys.size) * 2 + 2 * N,), dtype=torch.int32, device=DEV)
    betas = torch.empty((STU,), device=DEV)
    alphas = torch.empty_like(betas) if required_grad else betas
    grads = torch.empty_like(gather_xs) if required_grad else betas
    L.run_warp_rnnt_compact(counts.data_ptr(), alphas.data_ptr(), betas.data_ptr(), gather_xs.data_ptr(),
                            grads.data_ptr(), costs.data_ptr(), txn.data_ptr(), tyn.data_ptr(), tmem.data_ptr(),
                            tlab.data_ptr(), N, int(xn.max()), int(yn.max()) + 1, lam, required_grad)
    torch.cuda.synchronize()
    assert L.rnnt_amd_compact_last_status() == 0
    return costs, grads, loc, gather_xs


def test_reference_named_compact_entry_points():
    """run_gather_for_compact / run_warp_rnnt_compact / run_scatter_grad_for_compact (core.h:41-60) on the
    reference's compact golden vector (test.py:259-336) and on a seeded ragged batch against the oracle."""
    import warp_rnnt_amd
    L = warp_rnnt_amd.load()
    case = [c for c in reference_doc()["cases"] if c["name"] == "forward_batch_compact"][0]
    lp = np_log_softmax32(np.array(case["logits"], dtype=np.float32))
    labels = np.array(case["labels"], dtype=np.int32)
    xn = np.array(case["xn"], dtype=np.int32)
    yn = np.array(case["yn"], dtype=np.int32)
    xs, ys = pack(lp, labels, xn, yn)
    costs, grads, loc, _ = _ref_compact_abi(xs, ys, xn, yn, 0, 0.0)
    np.testing.assert_allclose(costs.cpu().numpy(), np.array(case["costs"]), atol=1.5e-6, rtol=0)
    V = lp.shape[-1]
    STU = xs.shape[0]
    cumlen = torch.cumsum(T(xn) * (T(yn) + 1), dim=0, dtype=torch.int32)
    dense = torch.zeros((STU, V), device=DEV)              # binding.cpp:237: zeros, then the scatter
    ones = torch.ones((len(xn),), device=DEV)
    torch.cuda.synchronize()
    L.run_scatter_grad_for_compact(ones.data_ptr(), grads.data_ptr(), loc.data_ptr(), cumlen.data_ptr(),
                                   dense.data_ptr(), STU, len(xn), V, 0)
    torch.cuda.synchronize()
    assert L.rnnt_amd_compact_last_status() == 0
    np.testing.assert_allclose(dense.cpu().numpy(), np.array(case["grads_rows"]), atol=1.5e-6, rtol=0)

    # seeded ragged batch, two column blocks, FastEmit, blank != 0; and the costs-only mode
    N, Tm, Um, V, lam, blank = 3, 70, 90, 5, 0.02, 2
    logits, labels, xn, yn = make_case(77 + Tm, N, Tm, Um, V, ragged=True, blank=blank)
    lp = np_log_softmax32(logits)
    ref = oracle.rnnt_loss_f32(lp, labels, xn, yn, blank=blank, fastemit_lambda=lam, scan_mode=1)
    xs, ys = pack(lp, labels, xn, yn)
    costs, grads, loc, gx = _ref_compact_abi(xs, ys, xn, yn, blank, lam)
    np.testing.assert_allclose(costs.cpu().numpy(), ref["costs"], rtol=1e-5)
    want_pairs = np.concatenate([oracle.gather_f32(lp[n:n + 1, :xn[n], :yn[n] + 1], labels[n:n + 1, :yn[n]], blank)[0]
                                 .reshape(-1, 2) for n in range(N)])
    got_pairs = gx.cpu().numpy()
    lastcol = np.concatenate([np.tile(np.arange(yn[n] + 1) == yn[n], xn[n]) for n in range(N)])
    np.testing.assert_array_equal(got_pairs[:, 0], want_pairs[:, 0])
    np.testing.assert_array_equal(got_pairs[~lastcol, 1], want_pairs[~lastcol, 1])
    g2 = np.concatenate([oracle.gather_f32(ref["grads"][n:n + 1, :xn[n], :yn[n] + 1], labels[n:n + 1, :yn[n]], blank)[0]
                         .reshape(-1, 2) for n in range(N)])
    g2[lastcol, 1] = 0
    np.testing.assert_allclose(grads.cpu().numpy(), g2, atol=1e-4)
    c2, _, _, _ = _ref_compact_abi(xs, ys, xn, yn, blank, lam, required_grad=False)
    np.testing.assert_allclose(c2.cpu().numpy(), ref["costs"], rtol=1e-5)
    # bad arguments are reported, not fatal
    L.run_scatter_grad_for_compact(None, None, None, None, None, 1, 1, 3, 7)
    assert L.rnnt_amd_compact_last_status() == 5 and L.rnnt_amd_compact_last_status() == 0


@pytest.mark.parametrize("N,Tm,Um,V,kernel", [
    (40, 500, 60, 5, "lattice_wd"),        # one column block: the plain launch of the column-block kernel
    (20, 500, 110, 5, "lattice_wl"),       # two: its single-workgroup form
    (6, 420, 430, 4, "lattice_ws"),        # seven: lattice_ws.hip on 32-bit offsets
])
def test_reference_named_compact_entry_points_staged(N, Tm, Um, V, kernel):
    """From 2^20 cells of launch bound on run_warp_rnnt_compact turns the row-major pairs into per-utterance
    diagonal-major planes inside the caller's `grads`, sweeps on the tuned kernels and turns the gradient pairs back
    (csrc/api.hip).  Against the oracle, ragged, FastEmit; bit-equal to the native compact entry, which runs the same
    kernels on 64-bit offsets."""
    import warp_rnnt_amd
    import warp_rnnt._C as core
    assert N * Tm * Um >= 1 << 20
    lam = 0.01
    logits, labels, xn, yn = make_case(300 + Um, N, Tm, Um, V, ragged=True)
    yn[1] = 0                                                   # an utterance without labels
    lp = np_log_softmax32(logits)
    ref = oracle.rnnt_loss_f32(lp, labels, xn, yn, blank=0, fastemit_lambda=lam, scan_mode=1)
    xs, ys = pack(lp, labels, xn, yn)
    costs, grads, loc, _ = _ref_compact_abi(xs, ys, xn, yn, 0, lam)
    from warp_rnnt_amd import debug
    assert debug.last_lattice_kernel() == kernel
    np.testing.assert_allclose(costs.cpu().numpy(), ref["costs"], rtol=1e-5)
    g2 = np.concatenate([oracle.gather_f32(ref["grads"][n:n + 1, :xn[n], :yn[n] + 1], labels[n:n + 1, :yn[n]], 0)[0]
                         .reshape(-1, 2) for n in range(N)])
    lastcol = np.concatenate([np.tile(np.arange(yn[n] + 1) == yn[n], xn[n]) for n in range(N)])
    g2[lastcol, 1] = 0
    np.testing.assert_allclose(grads.cpu().numpy(), g2, atol=5e-4)
    c_nat, g_nat, loc_nat = core.rnnt_loss_compact(T(xs), T(ys), T(xn), T(yn), blank=0, fastemit_lambda=lam)
    assert torch.equal(c_nat, costs) and torch.equal(g_nat, grads) and torch.equal(loc_nat, loc)
    # costs-only mode (alphas and grads alias betas: the direct form)
    c2, _, _, _ = _ref_compact_abi(xs, ys, xn, yn, 0, lam, required_grad=False)
    np.testing.assert_allclose(c2.cpu().numpy(), ref["costs"], rtol=1e-5)


@pytest.mark.parametrize("kernel", ["auto", "ws", "wd", "wl"])
@pytest.mark.parametrize("N,Tm,Um,V,lam", [
    (3, 40, 12, 9, 0.0),          # one column block
    (3, 70, 150, 5, 0.02),        # three column blocks, ragged ends in different blocks
    (2, 700, 200, 6, 0.0),        # long lattice: the distributed kernel by itself (rings sized by the launch bounds Tmax / Umax)
    (5, 9, 1, 4, 0.0),            # no labels at all
])
def test_compact_on_every_lattice_kernel(kernel, N, Tm, Um, V, lam):
    """The compact layout on every lattice kernel (each has a COMPACT instantiation; the hand-over rings of k_lattice_wd
    are sized by the launch bounds Tmax / Umax): gathered (STU,2) gradients and costs against the fp32 oracle, the same
    bits whichever kernel ran, and identical costs-only results."""
    from warp_rnnt_amd import debug, ops
    logits, labels, xn, yn = make_case(500 + Tm + Um, N, Tm, Um, V, ragged=True)
    lp = np_log_softmax32(logits)
    ref = oracle.rnnt_loss_f32(lp, labels, xn, yn, fastemit_lambda=lam, scan_mode=1)
    xs, ys = pack(lp, labels, xn, yn)
    c_auto, g_auto, _ = ops.loss_compact(T(xs), T(ys), T(xn), T(yn), 0, lam)
    with debug.lattice_kernel(kernel):
        costs, grads2, loc = ops.loss_compact(T(xs), T(ys), T(xn), T(yn), 0, lam)
        c_only, g_none, _ = ops.loss_compact(T(xs), T(ys), T(xn), T(yn), 0, lam, required_grad=False)
    torch.cuda.synchronize()
    assert torch.equal(costs, c_auto) and torch.equal(grads2, g_auto)
    np.testing.assert_allclose(costs.cpu().numpy(), ref["costs"], rtol=1e-5)
    np.testing.assert_array_equal(c_only.cpu().numpy(), costs.cpu().numpy())
    assert g_none is None
    # the oracle's dense gradient, reduced to the (blank, label) pair of every live cell in packing order
    want = []
    for n in range(N):
        tn, un = int(xn[n]), int(yn[n]) + 1
        g = ref["grads"][n, :tn, :un]                       # (tn, un, V)
        pair = np.zeros((tn, un, 2), np.float32)
        pair[..., 0] = g[..., 0]
        if un > 1:
            pair[:, :un - 1, 1] = np.take_along_axis(g[:, :un - 1], labels[n, :un - 1][None, :, None].astype(np.int64),
                                                     axis=2)[..., 0]
        want.append(pair.reshape(-1, 2))
    atol = 1e-4 * max(1.0, float(np.abs(ref["costs"]).max()) / 100.0)     # (fp32 noise of the oracle grows with |loglik|)
    np.testing.assert_allclose(grads2.cpu().numpy(), np.concatenate(want), atol=atol)
