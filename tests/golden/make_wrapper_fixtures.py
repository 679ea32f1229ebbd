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


def load_reference_wrapper():
    pkg = types.ModuleType("warp_rnnt")
    pkg.__path__ = []
    core = types.ModuleType("warp_rnnt._C")
    core.rnnt_loss = stub_rnnt_loss
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
    store["cases"] = np.array([";".join(map(str, c)) for c in cases])
    out = os.path.join(HERE, "wrapper_fixtures.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, len(cases), "cases", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
