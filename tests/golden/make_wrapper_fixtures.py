#!/usr/bin/env python
"""Generates tests/golden/wrapper_fixtures.npz by running the REFERENCE's own Python wrapper
(/root/reference/pytorch_binding/warp_rnnt/__init__.py, imported in place, never copied) on CPU
tensors with its native module `warp_rnnt._C` replaced by a stub that calls the fp32 CPU oracle.

This pins the wrapper-level semantics the reference's tests never check:
the gathered tensor handed to the native op for gather=True, average_frames / reduction outputs,
and log_probs.grad after backward() with a non-trivial upstream gradient -- including the
gather prologue's scatter-add backward.

Only runs in the build container (needs /root/reference); the resulting .npz is data
(inputs + expected outputs) and is what travels to the GPU box.

    python tests/golden/make_wrapper_fixtures.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

REF = "/root/reference/pytorch_binding/warp_rnnt/__init__.py"
calls = []


def stub_rnnt_loss(xs, ys, xn, yn, blank=0, fastemit_lambda=0.0):
    out = oracle.rnnt_loss_f32(xs.detach().numpy(), ys.numpy(), xn.numpy(), yn.numpy(), blank=blank,
                               fastemit_lambda=fastemit_lambda, scan_mode=1)
    calls.append(dict(xs=xs.detach().numpy().copy(), blank=blank))
    return torch.from_numpy(out["costs"].copy()), torch.from_numpy(out["grads"].copy())


def _unpack(xs, ys, xn, yn):
    """compact (STU,V)/(SU,) -> padded (N,Tm,Um,V)/(N,Um-1) (zeros in the padding)."""
    N = len(xn)
    Tm, Um, V = int(xn.max()), int(yn.max()) + 1, xs.shape[1]
    lp = np.zeros((N, Tm, Um, V), dtype=np.float32)
    lab = np.zeros((N, max(Um - 1, 1)), dtype=np.int32)
    o = lo = 0
    for n in range(N):
        t, u = int(xn[n]), int(yn[n]) + 1
        lp[n, :t, :u] = xs[o:o + t * u].reshape(t, u, V)
        lab[n, :u - 1] = ys[lo:lo + u - 1]
        o += t * u
        lo += u - 1
    return lp, lab[:, :max(Um - 1, 0)] if Um > 1 else lab[:, :0]


def stub_rnnt_loss_compact(xs, ys, xn, yn, blank=0, fastemit_lambda=0.0, required_grad=True):
    """binding.cpp:109-207 semantics on top of the padded oracle: (costs, grads (STU,2), loc (STU,))."""
    xs_, ys_, xn_, yn_ = xs.detach().numpy(), ys.numpy(), xn.numpy(), yn.numpy()
    lp, lab = _unpack(xs_, ys_, xn_, yn_)
    lp2 = oracle.gather_f32(lp, lab, blank) if lab.shape[1] else np.stack([lp[..., blank]] * 2, -1)
    out = oracle.rnnt_loss_f32(lp2, lab, xn_, yn_, blank=-1, fastemit_lambda=fastemit_lambda, scan_mode=1)
    g, loc = [], []
    for n in range(len(xn_)):
        t, u = int(xn_[n]), int(yn_[n]) + 1
        g.append(out["grads"][n, :t, :u].reshape(-1, 2))
        l = np.full((t, u), blank, dtype=np.int64)
        l[:, :u - 1] = lab[n, :u - 1][None, :]
        loc.append(l.reshape(-1))
    return (torch.from_numpy(out["costs"].copy()), torch.from_numpy(np.concatenate(g).copy()),
            torch.from_numpy(np.concatenate(loc)))


def stub_rnnt_loss_compact_backward(grad_cost, grad_xs, cum_lens, loc, V, blank):
    """binding.cpp:209-247 / core_compact.cu:456-484."""
    gc, g, cl, lc = grad_cost.numpy(), grad_xs.numpy(), cum_lens.numpy(), loc.numpy()
    out = np.zeros((g.shape[0], V), dtype=np.float32)
    n_of = np.searchsorted(cl, np.arange(g.shape[0]), side="right")
    out[np.arange(g.shape[0]), blank] = g[:, 0] * gc[n_of]
    m = lc != blank
    out[np.arange(g.shape[0])[m], lc[m]] = (g[:, 1] * gc[n_of])[m]
    return torch.from_numpy(out)


def load_reference_wrapper():
    pkg = types.ModuleType("warp_rnnt")
    pkg.__path__ = []
    core = types.ModuleType("warp_rnnt._C")
    core.rnnt_loss = stub_rnnt_loss
    core.rnnt_loss_compact = stub_rnnt_loss_compact
    core.rnnt_loss_compact_backward = stub_rnnt_loss_compact_backward
    sys.modules["warp_rnnt"] = pkg
    sys.modules["warp_rnnt._C"] = core
    pkg._C = core
    import pkg_resources

    class _Dist:
        version = "0.7.0"
    pkg_resources.get_distribution = lambda name: _Dist()
    spec = importlib.util.spec_from_file_location("warp_rnnt_reference_wrapper", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference_wrapper()
    rng = np.random.RandomState(7)
    N, T, U, V = 3, 6, 4, 5
    logits = rng.randn(N, T, U, V).astype(np.float32)
    xn = np.array([6, 4, 5], dtype=np.int32)
    yn = np.array([3, 1, 2], dtype=np.int32)
    store = dict(logits=logits, xn=xn, yn=yn)
    cases = []
    cid = 0
    for blank in (0, 2):
        choices = np.array([v for v in range(V) if v != blank], dtype=np.int32)
        labels = choices[rng.randint(0, len(choices), size=(N, U - 1))].astype(np.int32)
        for gather in (False, True):
            for reduction in ("none", "sum", "mean"):
                for average_frames in (False, True):
                    for lam in (0.0, 0.05):
                        if lam and (reduction == "none" and average_frames):
                            continue   # keep the file small
                        lp = torch.log_softmax(torch.from_numpy(logits), dim=-1).clone().requires_grad_(True)
                        calls.clear()
                        loss = ref.rnnt_loss(lp, torch.from_numpy(labels), torch.from_numpy(xn),
                                             torch.from_numpy(yn), average_frames=average_frames,
                                             reduction=reduction, blank=blank, gather=gather,
                                             fastemit_lambda=lam)
                        up = torch.tensor(np.asarray(rng.rand(*loss.shape) + 0.5, dtype=np.float32))
                        loss.backward(up)
                        key = f"c{cid:03d}"
                        store[key + "_labels"] = labels
                        store[key + "_loss"] = loss.detach().numpy()
                        store[key + "_up"] = up.numpy()
                        store[key + "_grad"] = lp.grad.numpy()
                        store[key + "_native_in"] = calls[0]["xs"]
                        cases.append((key, blank, int(gather), reduction, int(average_frames), lam,
                                      int(calls[0]["blank"])))
                        cid += 1
    # ---- compact layout (warp_rnnt/__init__.py:26-54,109-116) ----
    ccases = []
    lp_full = torch.log_softmax(torch.from_numpy(logits), dim=-1).numpy()
    for blank in (0, 2):
        choices = np.array([v for v in range(V) if v != blank], dtype=np.int32)
        labels = choices[rng.randint(0, len(choices), size=(N, U - 1))].astype(np.int32)
        xs_c = np.concatenate([lp_full[n, :xn[n], :yn[n] + 1].reshape(-1, V) for n in range(N)])
        ys_c = np.concatenate([labels[n, :yn[n]] for n in range(N)]).astype(np.int32)
        for reduction, average_frames, lam in (("none", False, 0.0), ("mean", True, 0.05), ("sum", False, 0.0)):
            lp = torch.from_numpy(xs_c.copy()).requires_grad_(True)
            loss = ref.rnnt_loss(lp, torch.from_numpy(ys_c), torch.from_numpy(xn), torch.from_numpy(yn),
                                 average_frames=average_frames, reduction=reduction, blank=blank,
                                 fastemit_lambda=lam, compact=True)
            up = torch.tensor(np.asarray(rng.rand(*loss.shape) + 0.5, dtype=np.float32))
            loss.backward(up)
            key = f"k{len(ccases):03d}"
            store[key + "_xs"] = xs_c
            store[key + "_ys"] = ys_c
            store[key + "_loss"] = loss.detach().numpy()
            store[key + "_up"] = up.numpy()
            store[key + "_grad"] = lp.grad.numpy()
            ccases.append((key, blank, reduction, int(average_frames), lam))
    store["compact_cases"] = np.array([";".join(map(str, c)) for c in ccases])
    store["cases"] = np.array([";".join(map(str, c)) for c in cases])
    out = os.path.join(HERE, "wrapper_fixtures.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, len(cases), "cases", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
