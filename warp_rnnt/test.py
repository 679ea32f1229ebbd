"""Self-test of the installed package: ``python -m warp_rnnt.test``.

What the reference promises its users (pytorch_binding/README.md:76-79: "python -m warp_rnnt.test") for this build:
the known answers of the reference's own unit tests -- shipped as data in ``golden_vectors.json`` next to this file --
through every layout of the native op (dense, gathered, compact), the argument errors of the native op with the
reference's messages, the wrapper's reductions, and a stress run at the reference's "calls" size whose results are checked
through invariants of the loss (the reference only checks that it returns).  Needs a GPU; nothing here touches the
oracle or the repository's own test tree.
"""
import json
import os
import unittest

import numpy as np
import torch

import warp_rnnt
from warp_rnnt import _C as core

_HERE = os.path.dirname(os.path.abspath(__file__))
_DOC = None
TOL = 1.5e-6          # the reference's assert_almost_equal(decimal=6)


def _golden():
    global _DOC
    if _DOC is None:
        with open(os.path.join(_HERE, "golden_vectors.json")) as f:
            _DOC = json.load(f)
    return _DOC


def _case(name):
    return next(c for c in _golden()["cases"] if c["name"] == name)


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _inputs(case):
    """log_softmax(logits), labels, frame and label counts of a golden case, on the GPU."""
    d = _dev()
    lp = torch.log_softmax(torch.tensor(case["logits"], dtype=torch.float32, device=d), dim=-1)
    n, _, u, _ = lp.shape
    ys = torch.tensor(case["labels"], dtype=torch.int32, device=d).reshape(n, u - 1)
    xn = torch.tensor(case["xn"], dtype=torch.int32, device=d)
    yn = torch.tensor(case["yn"], dtype=torch.int32, device=d)
    return lp.contiguous(), ys, xn, yn


def _gathered(lp, ys, blank):
    """The two-channel view the wrapper builds for gather=True: [blank, label], last column [blank, blank]."""
    n, t, u, _ = lp.shape
    idx = torch.full((n, t, u, 2), blank, dtype=torch.int64, device=lp.device)
    idx[:, :, :u - 1, 1] = ys.long()[:, None, :]
    return torch.gather(lp, 3, idx).contiguous()


@unittest.skipUnless(torch.cuda.is_available(), "warp_rnnt needs a GPU: there is no CPU path")
class KnownAnswers(unittest.TestCase):
    def check_dense(self, name):
        case = _case(name)
        lp, ys, xn, yn = _inputs(case)
        costs, grads = core.rnnt_loss(lp, ys, xn, yn, blank=case["blank"])
        np.testing.assert_allclose(costs.cpu().numpy(), np.array(case["costs"]), atol=TOL, rtol=0)
        np.testing.assert_allclose(grads.cpu().numpy(), np.array(case["grads"]), atol=TOL, rtol=0)
        return lp, ys, xn, yn, case

    def test_one_to_many(self):
        self.check_dense("one_to_many")

    def test_one_to_empty(self):
        self.check_dense("one_to_empty")

    def test_forward_single(self):
        self.check_dense("forward_single")

    def test_forward_batch(self):
        self.check_dense("forward_batch")

    def test_forward_single_gather(self):
        case = _case("forward_single_gather")
        lp, ys, xn, yn = _inputs(case)
        costs, grads = core.rnnt_loss(_gathered(lp, ys, case["blank"]), ys, xn, yn, blank=-1)
        np.testing.assert_allclose(costs.cpu().numpy(), np.array(case["costs"]), atol=TOL, rtol=0)
        np.testing.assert_allclose(grads.cpu().numpy(), np.array(case["grads"]), atol=TOL, rtol=0)

    def test_forward_batch_compact(self):
        case = _case("forward_batch_compact")
        lp, ys, xn, yn = _inputs(case)
        v = lp.shape[-1]
        rows = torch.cat([lp[n, :int(xn[n]), :int(yn[n]) + 1].reshape(-1, v) for n in range(lp.shape[0])]).contiguous()
        labs = torch.cat([ys[n, :int(yn[n])] for n in range(lp.shape[0])]).contiguous()
        costs, pairs, loc = core.rnnt_loss_compact(rows, labs, xn, yn, blank=case["blank"])
        np.testing.assert_allclose(costs.cpu().numpy(), np.array(case["costs"]), atol=TOL, rtol=0)
        ends = torch.cumsum(xn * (yn + 1), dim=0, dtype=torch.int32)
        dense = core.rnnt_loss_compact_backward(torch.ones_like(costs), pairs, ends, loc, v, case["blank"])
        np.testing.assert_allclose(dense.cpu().numpy(), np.array(case["grads_rows"]), atol=TOL, rtol=0)

    def test_wrapper_agrees_with_the_native_op_in_every_layout(self):
        """rnnt_loss(...), gather=True and compact=True on the batch case: one set of costs, one gradient."""
        case = _case("forward_batch")
        lp, ys, xn, yn = _inputs(case)
        want_c, want_g = np.array(case["costs"]), np.array(case["grads"])
        for gather in (False, True):
            x = lp.clone().requires_grad_(True)
            costs = warp_rnnt.rnnt_loss(x, ys, xn, yn, gather=gather)
            costs.sum().backward()
            np.testing.assert_allclose(costs.detach().cpu().numpy(), want_c, atol=TOL, rtol=0)
            np.testing.assert_allclose(x.grad.cpu().numpy(), want_g, atol=TOL, rtol=0)
        x = lp.clone().requires_grad_(True)
        mean = warp_rnnt.rnnt_loss(x, ys, xn, yn, reduction="mean", average_frames=True)
        self.assertAlmostEqual(mean.item(), float(np.mean(want_c / np.array(case["xn"]))), places=5)
        with self.assertRaises((AssertionError, ValueError)):      # (the reference asserts on the flag before it gets there)
            warp_rnnt.rnnt_loss(lp, ys, xn, yn, reduction="median")


@unittest.skipUnless(torch.cuda.is_available(), "warp_rnnt needs a GPU: there is no CPU path")
class ArgumentErrors(unittest.TestCase):
    """The four misuse cases the reference's tests pin, with its messages (golden_vectors.json: argument_errors)."""

    def messages(self):
        return {e["what"]: e["message"] for e in _golden()["argument_errors"]}

    def test_messages(self):
        msg = self.messages()
        d = _dev()
        e32 = lambda dt, dev=None: torch.tensor([], dtype=dt, device=dev)      # noqa: E731
        cpu = (e32(torch.float32), e32(torch.int32), e32(torch.int32), e32(torch.int32))
        striped = torch.zeros((4, 3, 2, 1), dtype=torch.float32).transpose(0, 1)
        with self.assertRaisesRegex(RuntimeError, msg["non-contiguous xs"]):
            core.rnnt_loss(striped, *cpu[1:])
        with self.assertRaisesRegex(RuntimeError, msg["CPU tensors"]):
            core.rnnt_loss(*cpu)
        with self.assertRaisesRegex(RuntimeError, msg["1-D empty xs on device"]):
            core.rnnt_loss(*(t.to(d) for t in cpu))
        with self.assertRaisesRegex(RuntimeError, msg["int64 ys"]):
            core.rnnt_loss(cpu[0], e32(torch.int64), cpu[2], cpu[3])


@unittest.skipUnless(torch.cuda.is_available(), "warp_rnnt needs a GPU: there is no CPU path")
class Stress(unittest.TestCase):
    def test_calls(self):
        """The reference's `calls` sizes (N=128, T=100, U=90, V=3, random label counts), two seeds.  There the test is that
        the op returns; here its results also have to be a loss: finite positive costs, and gradients that are minus the
        expected number of times each arc is taken -- every frame leaves through exactly one blank, every label is emitted
        exactly once, nothing flows outside an utterance's own lattice."""
        meta = _golden()["smoke_only"][0]
        n, t, u, v = meta["N"], meta["T"], meta["U"], meta["V"]
        d = _dev()
        for seed in (0, 1):
            g = torch.Generator(device=d)
            g.manual_seed(seed)
            lp = torch.log_softmax(torch.randn((n, t, u, v), device=d, generator=g), dim=-1)
            ys = torch.randint(1, v, (n, u - 1), dtype=torch.int32, device=d, generator=g)
            xn = torch.full((n,), t, dtype=torch.int32, device=d)
            yn = torch.randint(1, u, (n,), dtype=torch.int32, device=d, generator=g)
            costs, grads = core.rnnt_loss(_gathered(lp, ys, 0), ys, xn, yn, blank=-1)
            torch.cuda.synchronize()
            self.assertTrue(bool(torch.isfinite(costs).all()) and bool((costs > 0).all()))
            self.assertTrue(bool((grads <= 0).all()))
            blank_per_frame = grads[..., 0].sum(dim=2)                     # (N, T): -1 for every frame
            np.testing.assert_allclose(blank_per_frame.cpu().numpy(), -1.0, atol=5e-4)
            label_per_column = grads[..., 1].sum(dim=1).cpu().numpy()      # (N, U): -1 for u < yn, 0 beyond
            live = np.arange(u)[None, :] < yn.cpu().numpy()[:, None]
            np.testing.assert_allclose(label_per_column[live], -1.0, atol=5e-4)
            self.assertTrue((label_per_column[~live] == 0).all())


def main():
    unittest.main(module=__name__, verbosity=2)


if __name__ == "__main__":
    main()
