"""GPU tests of the drop-in Python API (warp_rnnt.rnnt_loss) against fixtures produced by the
REFERENCE's own wrapper (tests/golden/make_wrapper_fixtures.py) and against the oracle."""
import os

import numpy as np
import pytest
import torch

import oracle
from helpers import GOLDEN, make_case, np_log_softmax32

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    a = np.asarray(a)
    return torch.tensor(a if a.ndim == 0 else np.ascontiguousarray(a), device=DEV)


def test_wrapper_fixtures_from_reference_wrapper():
    """loss values and log_probs.grad for every (blank, gather, reduction, average_frames, lambda)
    combination, with a random upstream gradient."""
    import warp_rnnt
    fx = np.load(os.path.join(GOLDEN, "wrapper_fixtures.npz"))
    lp0 = np_log_softmax32(fx["logits"])
    xn, yn = T(fx["xn"]), T(fx["yn"])
    for row in fx["cases"]:
        key, blank, gather, reduction, avg, lam, _ = row.split(";")
        lp = T(lp0).requires_grad_(True)
        loss = warp_rnnt.rnnt_loss(lp, T(fx[key + "_labels"]), xn, yn, average_frames=bool(int(avg)),
                                   reduction=reduction, blank=int(blank), gather=bool(int(gather)),
                                   fastemit_lambda=float(lam))
        np.testing.assert_allclose(loss.detach().cpu().numpy(), fx[key + "_loss"], rtol=1e-5, err_msg=row)
        loss.backward(T(fx[key + "_up"]))
        np.testing.assert_allclose(lp.grad.cpu().numpy(), fx[key + "_grad"], atol=2e-6, err_msg=row)


def test_gather_true_equals_dense_path():
    """tensorflow_binding/warp_rnnt_tf/test.py:227-252: wrapper-level gather=True must give the
    dense (N,T,U,V) gradients of the gather=False path."""
    import warp_rnnt
    logits, labels, xn, yn = make_case(21, 4, 33, 70, 9, ragged=True)
    lp0 = np_log_softmax32(logits)
    grads = []
    for gather in (False, True):
        lp = T(lp0).requires_grad_(True)
        loss = warp_rnnt.rnnt_loss(lp, T(labels), T(xn), T(yn), gather=gather, reduction="sum",
                                   fastemit_lambda=0.01)
        loss.backward()
        grads.append((loss.item(), lp.grad.cpu().numpy()))
    assert grads[0][0] == grads[1][0]
    np.testing.assert_array_equal(grads[0][1], grads[1][1])
    ref = oracle.rnnt_loss_f32(lp0, labels, xn, yn, fastemit_lambda=0.01, scan_mode=1)
    np.testing.assert_allclose(grads[0][1], ref["grads"], atol=1e-4)


@pytest.mark.parametrize("name", ["forward_single", "forward_batch", "one_to_many"])
def test_reference_golden_through_gather_true_backward(name):
    """The golden vectors of the reference's dense tests (test.py:34-188; the TensorFlow binding's test_forward_single
    holds the same numbers, tensorflow_binding/warp_rnnt_tf/test.py) through the WRAPPER's gather=True path:
    rnnt_loss(log_softmax(logits), gather=True) + backward() must give the golden costs and the golden dense
    (N,T,U,V) gradients, within the reference's own tolerance."""
    import warp_rnnt
    from helpers import reference_cases
    case = [c for c in reference_cases(("dense",)) if c["name"] == name][0]
    lp0 = np_log_softmax32(np.array(case["logits"], dtype=np.float32))
    N, _, U, _ = lp0.shape
    lp = T(lp0).requires_grad_(True)
    costs = warp_rnnt.rnnt_loss(lp, T(np.array(case["labels"], dtype=np.int32).reshape(N, U - 1)),
                                T(np.array(case["xn"], dtype=np.int32)),
                                T(np.array(case["yn"], dtype=np.int32)), gather=True, blank=case["blank"])
    np.testing.assert_allclose(costs.detach().cpu().numpy(), np.array(case["costs"]), atol=1.5e-6, rtol=0)
    costs.sum().backward()
    np.testing.assert_allclose(lp.grad.cpu().numpy(), np.array(case["grads"]), atol=1.5e-6, rtol=0)


def test_forward_computes_grads_without_requires_grad_and_second_backward():
    """Appendix B: grads are produced in forward; backward is a broadcast multiply and can be
    repeated (the reference multiplies in place, which would double-scale)."""
    import warp_rnnt
    logits, labels, xn, yn = make_case(2, 2, 12, 5, 6)
    lp = T(np_log_softmax32(logits))
    c = warp_rnnt.rnnt_loss(lp, T(labels), T(xn), T(yn))          # no grad needed: still fine
    assert c.shape == (2,) and not c.requires_grad
    lp.requires_grad_(True)
    c = warp_rnnt.rnnt_loss(lp, T(labels), T(xn), T(yn))
    g1, = torch.autograd.grad(c.sum() * 3.0, lp, retain_graph=True)
    g2, = torch.autograd.grad(c.sum() * 3.0, lp)
    assert torch.equal(g1, g2)


def test_log_softmax_then_loss_matches_fused_from_logits():
    """The fused logits entry (log-softmax + gather in one kernel) gives the same costs/gathered
    grads as log_softmax followed by the gather path."""
    from warp_rnnt_amd import ops
    logits, labels, xn, yn = make_case(8, 3, 40, 21, 50, ragged=True)
    x = T(logits)
    c1, g1 = ops.loss(ops.log_softmax(x), T(labels), T(xn), T(yn), ops.IN_LOG_PROBS_DENSE,
                      ops.GRADS_GATHERED, 0, 0.0)
    c2, g2 = ops.loss(x, T(labels), T(xn), T(yn), ops.IN_LOGITS_DENSE, ops.GRADS_GATHERED, 0, 0.0)
    np.testing.assert_allclose(c1.cpu().numpy(), c2.cpu().numpy(), rtol=1e-6)
    # the two routes may differ by an ulp in the log-probs, which the lattice turns into ~1e-5
    # relative differences of the gradients (|alpha+beta| ~ 1e2 here)
    np.testing.assert_allclose(g1.cpu().numpy(), g2.cpu().numpy(), atol=1e-4)


def test_gather_true_takes_any_strides_like_the_reference_and_gather_false_still_raises():
    """In the reference `log_probs.gather(dim=3, index)` (__init__.py:126) runs before CHECK_CONTIGUOUS(xs) ever sees the
    tensor (binding.cpp:33): a transposed or sliced joint output works with gather=True and raises "xs must be contiguous"
    with gather=False.  Same here (missing #2): values and gradients equal to the contiguous call's."""
    import warp_rnnt
    logits, labels, xn, yn = make_case(21, 3, 30, 17, 9, ragged=True)
    lp = np_log_softmax32(logits)
    base = T(lp).requires_grad_(True)
    c0 = warp_rnnt.rnnt_loss(base, T(labels), T(xn), T(yn), gather=True, fastemit_lambda=0.01)
    (c0 * T(np.array([1.0, 2.0, 3.0], np.float32))).sum().backward()
    # (N,U,T,V) storage viewed as (N,T,U,V): what a joint network that builds (N,U,T,*) and transposes hands over
    store = T(np.ascontiguousarray(lp.transpose(0, 2, 1, 3))).requires_grad_(True)
    view = store.transpose(1, 2)
    assert not view.is_contiguous() and view.shape == base.shape
    c1 = warp_rnnt.rnnt_loss(view, T(labels), T(xn), T(yn), gather=True, fastemit_lambda=0.01)
    (c1 * T(np.array([1.0, 2.0, 3.0], np.float32))).sum().backward()
    assert torch.equal(c0, c1) and torch.equal(store.grad.transpose(1, 2), base.grad)
    # a slice along V (a wider joint output of which the loss sees a prefix)
    wide = T(np.concatenate([lp, np.zeros_like(lp[..., :3])], axis=-1))
    c2 = warp_rnnt.rnnt_loss(wide[..., :9], T(labels), T(xn), T(yn), gather=True, fastemit_lambda=0.01)
    assert torch.equal(c0, c2)
    with pytest.raises(RuntimeError, match="xs must be contiguous"):
        warp_rnnt.rnnt_loss(view, T(labels), T(xn), T(yn), gather=False)
    with pytest.raises(RuntimeError, match="ys must be contiguous"):         # only log_probs went through torch.gather there
        warp_rnnt.rnnt_loss(base, T(np.concatenate([labels, labels], 1))[:, ::2], T(xn), T(yn), gather=True)


@pytest.mark.parametrize("N,Tm,Um,V,lam", [(3, 40, 21, 50, 0.0), (2, 700, 70, 11, 0.01), (4, 33, 9, 1500, 0.0)])
def test_lazy_log_softmax_keeps_the_reference_call_shape_and_runs_the_fused_path(N, Tm, Um, V, lam):
    """`rnnt_loss(log_softmax(logits), ..., gather=True)` -- benchmark.py:65-70's call -- with
    warp_rnnt_amd.functional.log_softmax: the handle it returns computes nothing; rnnt_loss recognises it and runs the fused
    logits -> pairs -> loss path and, in backward, logits -> d/d logits.  THE SAME BITS as
    rnnt_loss_from_logits, forward and backward, under every reduction; the same numbers as the materialised chain; any
    other consumer of the handle gets the log-probabilities (and their gradient flows through the log-softmax backward)."""
    import warp_rnnt
    from warp_rnnt_amd import functional as F2
    from warp_rnnt_amd.fused import rnnt_loss_from_logits
    logits, labels, xn, yn = make_case(40 + Tm, N, Tm, Um, V, ragged=True)
    tl, txn, tyn = T(labels), T(xn), T(yn)
    up = T(np.linspace(0.5, 1.5, N).astype(np.float32))
    for reduction, avg in (("none", False), ("mean", True), ("sum", False)):
        xa, xb, xc = (T(logits).requires_grad_(True) for _ in range(3))
        handle = F2.log_softmax(xa)
        assert isinstance(handle, F2.LazyLogSoftmax) and not handle.materialised and handle.shape == xa.shape
        assert handle.dtype == torch.float32 and handle.is_cuda and handle.requires_grad and handle.logits is xa
        la = warp_rnnt.rnnt_loss(handle, tl, txn, tyn, gather=True, fastemit_lambda=lam, reduction=reduction,
                                 average_frames=avg)
        assert not handle.materialised                               # the log-probabilities never existed
        lb = rnnt_loss_from_logits(xb, tl, txn, tyn, fastemit_lambda=lam, reduction=reduction, average_frames=avg)
        lc = warp_rnnt.rnnt_loss(F2.log_softmax(xc, lazy=False), tl, txn, tyn, gather=True, fastemit_lambda=lam,
                                 reduction=reduction, average_frames=avg)
        for l_ in (la, lb, lc):
            (l_ * up if reduction == "none" else l_ * 1.25).sum().backward()
        assert torch.equal(la, lb) and torch.equal(xa.grad, xb.grad)                       # bit for bit the fused entry
        np.testing.assert_allclose(la.detach().cpu().numpy(), lc.detach().cpu().numpy(), rtol=2e-6)
        # (two fp32 routes to the log-probs, an ulp apart; the lattice turns that into ~ulp(|log-likelihood|) on the gradients)
        np.testing.assert_allclose(xa.grad.cpu().numpy(), xc.grad.cpu().numpy(), atol=1e-3 if Tm > 500 else 2e-4)
    # another consumer: the handle becomes the log-probabilities (once), with autograd through the log-softmax backward
    xd, xe = T(logits).requires_grad_(True), T(logits).requires_grad_(True)
    h = F2.log_softmax(xd)
    w = T(np.random.RandomState(1).randn(*logits.shape).astype(np.float32))
    (h * w).sum().backward()
    assert h.materialised
    (torch.log_softmax(xe, -1) * w).sum().backward()
    np.testing.assert_allclose(h.detach().cpu().numpy(), torch.log_softmax(xe, -1).detach().cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xe.grad.cpu().numpy(), atol=2e-5)
    # both routes on one handle: the gradients add up
    xf = T(logits).requires_grad_(True)
    hf = F2.log_softmax(xf)
    (warp_rnnt.rnnt_loss(hf, tl, txn, tyn, gather=True, reduction="sum") + (hf * w).sum()).backward()
    xg = T(logits).requires_grad_(True)
    lpg = torch.log_softmax(xg, -1)
    (warp_rnnt.rnnt_loss(lpg, tl, txn, tyn, gather=True, reduction="sum") + (lpg * w).sum()).backward()
    np.testing.assert_allclose(xf.grad.cpu().numpy(), xg.grad.cpu().numpy(), atol=1e-3 if Tm > 500 else 2e-4)
    # gather=False / a leaf handle that requires grad itself: the ordinary path on materialised log-probabilities
    h2 = F2.log_softmax(T(logits))
    if V <= 64:
        c_dense = warp_rnnt.rnnt_loss(h2, tl, txn, tyn, gather=False)
        assert h2.materialised
        c_ref = warp_rnnt.rnnt_loss(torch.log_softmax(T(logits), -1), tl, txn, tyn, gather=False)
        np.testing.assert_allclose(c_dense.cpu().numpy(), c_ref.cpu().numpy(), rtol=2e-6)
    h3 = F2.log_softmax(T(logits)).requires_grad_(True)
    assert h3.is_leaf and not h3.fusable()
    warp_rnnt.rnnt_loss(h3, tl, txn, tyn, gather=True, reduction="sum").backward()
    lp3 = torch.log_softmax(T(logits), -1).requires_grad_(True)
    warp_rnnt.rnnt_loss(lp3, tl, txn, tyn, gather=True, reduction="sum").backward()
    np.testing.assert_allclose(h3.grad.cpu().numpy(), lp3.grad.cpu().numpy(), atol=1e-3 if Tm > 500 else 2e-4)


def test_sharded_loss_single_process():
    from warp_rnnt_amd.distributed import sharded_rnnt_loss
    logits, labels, xn, yn = make_case(4, 5, 20, 7, 8, ragged=True)
    lp0 = np_log_softmax32(logits)
    lp = T(lp0).requires_grad_(True)
    loss, glob = sharded_rnnt_loss(lp, T(labels), T(xn), T(yn), reduction="mean", gather=True)
    ref = oracle.rnnt_loss_f32(lp0, labels, xn, yn, scan_mode=1)
    np.testing.assert_allclose(glob.item(), ref["costs"].mean(), rtol=1e-5)
    loss.backward()
    np.testing.assert_allclose(lp.grad.cpu().numpy(), ref["grads"] / 5.0, atol=1e-5)


def test_non_default_stream_and_device_guard():
    """Launches go to the caller's current stream (binding.cpp:77) and are ordered with it."""
    import warp_rnnt
    logits, labels, xn, yn = make_case(9, 2, 30, 9, 6)
    lp0 = np_log_softmax32(logits)
    ref = oracle.rnnt_loss_f32(lp0, labels, xn, yn, scan_mode=1)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        lp = T(lp0) * 1.0                       # produced on stream s
        c = warp_rnnt.rnnt_loss(lp, T(labels), T(xn), T(yn), gather=True)
        tot = c.sum()
    s.synchronize()
    np.testing.assert_allclose(tot.item(), ref["costs"].sum(), rtol=1e-5)


# V = 32, 64, 128, 256 and V % 4 == 0 from 448 on: the rows-in-registers fused gather (prologue.hip: k_lsm_rows, 8 ... 64
# lanes per row, ragged last float4; k_lsm_rows_diag for V = 32, 64 and its T < 16 fallback); the others stay on the LDS
# tiles (straight-line row pass for 9 ... 16 columns per lane: 34 ... 258 here; run-time loops below: V=7)
@pytest.mark.parametrize("N,Tm,Um,V", [(3, 30, 12, 50), (2, 9, 5, 5000), (2, 11, 70, 7), (2, 6, 4, 1030),
                                       (3, 40, 21, 128), (2, 33, 9, 64), (2, 21, 12, 32), (2, 17, 14, 80),
                                       (2, 17, 14, 96), (2, 19, 8, 160), (2, 19, 8, 192), (2, 19, 8, 256),
                                       (2, 13, 8, 200), (2, 9, 20, 128), (2, 30, 35, 64),
                                       (2, 21, 9, 34), (2, 21, 9, 66), (2, 15, 9, 130), (2, 11, 7, 258), (2, 9, 5, 510),
                                       (2, 9, 5, 514), (2, 9, 5, 1000)])
def test_fused_from_logits_forward_backward(N, Tm, Um, V):
    """rnnt_loss_from_logits == rnnt_loss(torch.log_softmax(logits), gather=True) incl. d/d logits
    (small-V, large-V and generic log-softmax kernels)."""
    import warp_rnnt
    from warp_rnnt_amd.fused import rnnt_loss_from_logits
    logits, labels, xn, yn = make_case(5 + V, N, Tm, Um, V, ragged=True)
    up = np.random.RandomState(0).rand(N).astype(np.float32) + 0.5
    z1 = T(logits).requires_grad_(True)
    l1 = warp_rnnt.rnnt_loss(torch.log_softmax(z1, -1), T(labels), T(xn), T(yn), gather=True,
                             fastemit_lambda=0.01)
    l1.backward(T(up))
    z2 = T(logits).requires_grad_(True)
    l2 = rnnt_loss_from_logits(z2, T(labels), T(xn), T(yn), fastemit_lambda=0.01)
    l2.backward(T(up))
    np.testing.assert_allclose(l2.detach().cpu().numpy(), l1.detach().cpu().numpy(), rtol=1e-5)
    # (two fp32 evaluations of one function: torch's log-softmax and its backward against the fused kernels)
    np.testing.assert_allclose(z2.grad.cpu().numpy(), z1.grad.cpu().numpy(), atol=3e-5, rtol=3e-5)
    # against exact arithmetic: dz = g - softmax * sum(g) with the fp64 oracle's g
    from oracle import transduce_np
    lp64 = transduce_np.log_softmax(logits)
    c64, g64 = transduce_np.transduce_batch(lp64, labels, xn, yn, fastemit_lambda=0.01, fast=True)
    g64 = g64 * up[:, None, None, None]
    dz64 = g64 - np.exp(lp64) * g64.sum(-1, keepdims=True)
    np.testing.assert_allclose(z2.grad.cpu().numpy(), dz64, atol=1e-4)
    np.testing.assert_allclose(l2.detach().cpu().numpy(), c64, rtol=1e-5)


def test_hip_graph_capture_and_replay():
    """The op only enqueues work on the caller's stream (no host sync, no implicit allocation outside
    torch's allocator), so a whole log_softmax + loss + backward step can be captured in a HIP graph
    and replayed -- the launch-bound small configurations (BASELINE config 2) benefit most."""
    import warp_rnnt
    from warp_rnnt_amd import ops
    logits, labels, xn, yn = make_case(31, 16, 150, 40, 28)
    x = T(logits)
    ys, txn, tyn = T(labels), T(xn), T(yn)
    static_lp = torch.empty_like(x).requires_grad_(True)

    def step():
        ops.log_softmax(x, out=static_lp.detach())
        loss = warp_rnnt.rnnt_loss(static_lp, ys, txn, tyn, reduction="sum")
        g, = torch.autograd.grad(loss, static_lp)
        return loss, g

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss, g = step()
    x.copy_(T(logits * 0.5))          # new data in the static input buffer
    graph.replay()
    torch.cuda.synchronize()
    lp = np_log_softmax32(logits * 0.5)
    ref = oracle.rnnt_loss_f32(lp, labels, xn, yn, scan_mode=1)
    np.testing.assert_allclose(loss.item(), ref["costs"].sum(), rtol=1e-5)
    np.testing.assert_allclose(g.cpu().numpy(), ref["grads"], atol=1e-4)


@pytest.mark.parametrize("shape", [(1000, 50), (37, 5000), (64, 1030), (5, 7, 3, 28), (11, 2), (33, 2560), (33, 2564),
                                   (300, 17), (300, 36), (300, 100), (300, 128), (300, 200), (300, 500), (70, 1000),
                                   (21, 5124), (9, 10000), (9, 12292), (5, 16384), (3, 17000)])
def test_log_softmax_autograd(shape):
    from warp_rnnt_amd.functional import log_softmax
    x = (torch.randn(*shape, device=DEV) * 2).requires_grad_(True)
    up = torch.randn(*shape, device=DEV)
    y = log_softmax(x)
    y.backward(up)
    xr = x.detach().double().requires_grad_(True)
    yr = torch.log_softmax(xr, -1)
    yr.backward(up.double())
    assert (y.double() - yr).abs().max().item() < 5e-6
    assert (x.grad.double() - xr.grad).abs().max().item() < 5e-5 * max(1.0, up.abs().sum(-1).max().item() / 50)


# ----------------------------------------------------------------------------
# joint network in front of the loss (examples/joint_benchmark.py; benchmark2.py:93-164)
# ----------------------------------------------------------------------------
def _load_joint_example():
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "joint_benchmark.py")
    spec = importlib.util.spec_from_file_location("joint_benchmark", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.gpu
def test_joint_network_all_loss_paths_agree():
    """dense, gather, compact, fused-from-logits and the lazy log_softmax give the same loss and the same d/df, d/dg."""
    jb = _load_joint_example()
    N, T, U, V, H = 4, 17, 6, 23, 32
    torch.manual_seed(5)
    f, g, ys, f_len, g_len = jb.make_batch(N, T, U, V, H, True, torch.device("cuda"))
    weights = None
    results = {}
    for name in jb.LOSSES:
        joint = jb.JointNetwork(H, V, packed=name.endswith("compact"),
                                log_softmax="lazy" if name.endswith("lazy") else not name.endswith("fused")).cuda()
        if weights is None:
            weights = {k: v.clone() for k, v in joint.state_dict().items()}
        joint.load_state_dict(weights)
        ff, gg = f.clone().requires_grad_(True), g.clone().requires_grad_(True)
        costs = jb.pick_loss(name)(joint(ff, gg, f_len, g_len), ys, f_len, g_len)
        (costs * torch.arange(1, N + 1, device=costs.device)).sum().backward()
        results[name] = (costs.detach().cpu().numpy(), ff.grad.cpu().numpy(), gg.grad.cpu().numpy(),
                         joint.proj.weight.grad.cpu().numpy())
    ref = results["warp-rnnt"]
    for name, got in results.items():
        np.testing.assert_allclose(got[0], ref[0], rtol=2e-5, err_msg=name)
        for a, b in zip(got[1:], ref[1:]):
            np.testing.assert_allclose(a, b, atol=5e-4, rtol=1e-3, err_msg=name)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--ddp"], ["--fwd-only", "--random-length"]])
def test_joint_benchmark_cli_runs(extra):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    for loss in ("warp-rnnt-gather", "warp-rnnt-fused", "warp-rnnt-compact", "warp-rnnt-lazy"):
        out = subprocess.run([sys.executable, os.path.join(root, "examples", "joint_benchmark.py"), "--loss", loss,
                              "--shapes", "20,5,11", "--batches", "3", "--iters", "2", "--warmup", "1",
                              "--hidden", "16"] + extra, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "| 20 | 5 | 11 | 3 |" in out.stdout


def test_both_host_bindings_give_the_same_bits():
    """The compiled binding (warp_rnnt._C_native) and the ctypes fallback drive the same library: the golden
    vectors and a seeded gather=True case through both, bit for bit (the fallback in a subprocess, selected with
    WARP_RNNT_AMD_NO_NATIVE_BINDING)."""
    import os
    import subprocess
    import sys
    import warp_rnnt._C as core
    assert core._native is not None
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import warp_rnnt, warp_rnnt._C as core
from helpers import make_case, np_log_softmax32
assert (core._native is None) == bool(int(sys.argv[1]))
logits, labels, xn, yn = make_case(9, 5, 37, 70, 11, ragged=True)
d = torch.device("cuda:0")
lp = torch.tensor(np_log_softmax32(logits), device=d, requires_grad=True)
c = warp_rnnt.rnnt_loss(lp, torch.tensor(labels, device=d), torch.tensor(xn, device=d), torch.tensor(yn, device=d),
                        gather=bool(int(sys.argv[2])), fastemit_lambda=0.01)
(c * torch.arange(1, 6, device=d)).sum().backward()
sys.stdout.buffer.write(c.detach().cpu().numpy().tobytes() + lp.grad.cpu().numpy().tobytes())
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    for gather in ("0", "1"):
        outs = []
        for no_native in ("0", "1"):
            env = dict(os.environ)
            env.pop("WARP_RNNT_AMD_NO_NATIVE_BINDING", None)
            if no_native == "1":
                env["WARP_RNNT_AMD_NO_NATIVE_BINDING"] = "1"
            r = subprocess.run([sys.executable, "-c", code, no_native, gather], env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=300)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            outs.append(r.stdout)
        assert len(outs[0]) > 1000 and outs[0] == outs[1]


# ---------------------------------------------------------------------------
# the reference's benchmark CLI (tools/benchmark_table.py; pytorch_binding/benchmark.py:53-103): every --loss runs
# on a small grid, prints the reference's line format and writes the table
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("loss", ["warp-rnnt", "warp-rnnt-gather", "warp-rnnt-compact", "warp-rnnt-fused",
                                  "warp-rnnt-lazy", "torch-log-softmax-gather"])
def test_benchmark_table_cli(loss, tmp_path):
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = tmp_path / "table.md"
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "benchmark_table.py"), "--loss", loss,
                          "--grid", "2,30,9,11;2,150,40,28", "--batches", "1,3", "--random_length", "--markdown", str(md)],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-2000:] + out.stderr.decode()[-2000:]
    lines = re.findall(r"^T=(\d+)\tU=(\d+)\tV=(\d+)\tN=(\d+)\ttime=([0-9.]+)$", text, flags=re.M)
    assert [(int(t), int(u), int(v), int(n)) for t, u, v, n, _ in lines] == \
        [(30, 9, 11, 1), (30, 9, 11, 3), (150, 40, 28, 1), (150, 40, 28, 3)], text[-2000:]
    assert all(0.0 < float(ms) < 1000.0 for *_, ms in lines)
    table = md.read_text().splitlines()
    assert len(table) == 2 + 4 and table[0].startswith("| T | U | V | N |")
    # the reference's published cell is quoted where there is one (README.md:38: T=150,U=40,V=28,N=1)
    assert table[4].split("|")[6].strip() in ("0.5", "0.54")


def test_python_dash_m_warp_rnnt_test():
    """The package's self-test entry point (the reference: pytorch_binding/README.md:76-79) runs green on the GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "warp_rnnt.test"], cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "OK" in out.stderr and "skipped" not in out.stderr
