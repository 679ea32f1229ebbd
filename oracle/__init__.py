"""CPU oracle for the RNN-T loss hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package.  The product packages
(``warp_rnnt``, ``warp_rnnt_amd``) never do, and fail loudly when their HIP
library is missing instead of falling back to anything here.

Two restatements live here:

* ``rnnt_oracle.c`` (via :func:`rnnt_loss_f32`): fp32, follows the reference's
  operation order (core_gather.cu / core.cu, cited per function in the C file).
* ``transduce_np.py``: fp64 NumPy, awni/transducer ``ref_transduce.py`` style
  (that file is not vendored in the reference and not available offline; it is
  restated from the published algorithm, see the module docstring).

Parity pin: both are checked against the golden vectors of the reference's own
tests (``tests/golden/reference_vectors.json``) in ``tests/test_oracle.py``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librnnt_oracle.so")
_lib = None


def build(force=False):
    """Compile rnnt_oracle.c with gcc (seconds)."""
    src = os.path.join(_HERE, "rnnt_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "librnnt_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        L.oracle_rnnt_loss.restype = ctypes.c_int
        L.oracle_rnnt_loss.argtypes = [fp, ip, ip, ip] + [ctypes.c_int] * 5 + \
            [ctypes.c_float, ctypes.c_int, fp, fp, fp, fp, ip]
        L.oracle_log_softmax.restype = None
        L.oracle_log_softmax.argtypes = [fp, fp, ctypes.c_long, ctypes.c_int]
        L.oracle_gather.restype = None
        L.oracle_gather.argtypes = [fp, ip, fp] + [ctypes.c_int] * 5
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_set_threads.restype = None
        L.oracle_set_threads.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def rnnt_loss_f32(log_probs, labels, xn, yn, blank=0, fastemit_lambda=0.0, scan_mode=0):
    """fp32 oracle.  ``blank=-1`` selects the gathered (N,T,U,2) layout.

    Returns dict(costs, grads, alphas, betas, mismatch); grads has the shape
    of ``log_probs``.
    """
    lp = np.ascontiguousarray(log_probs, dtype=np.float32)
    N, T, U, V = lp.shape
    xn = np.ascontiguousarray(xn, dtype=np.int32)
    yn = np.ascontiguousarray(yn, dtype=np.int32)
    if labels is None or np.asarray(labels).size == 0:
        lab = np.zeros((N, max(U - 1, 1)), dtype=np.int32)
    else:
        lab = np.ascontiguousarray(labels, dtype=np.int32).reshape(N, U - 1)
    assert xn.shape == (N,) and yn.shape == (N,)
    assert (xn >= 1).all() and (xn <= T).all() and (yn >= 0).all() and (yn <= U - 1).all()
    alphas = np.zeros((N, T, U), dtype=np.float32)
    betas = np.zeros((N, T, U), dtype=np.float32)
    grads = np.empty_like(lp)
    costs = np.empty((N,), dtype=np.float32)
    mism = np.zeros((N,), dtype=np.int32)
    rc = lib().oracle_rnnt_loss(_f(lp), _i(lab), _i(xn), _i(yn), N, T, U, V, int(blank),
                                float(fastemit_lambda), int(scan_mode),
                                _f(alphas), _f(betas), _f(grads), _f(costs), _i(mism))
    if rc != 0:
        raise ValueError("oracle_rnnt_loss: bad arguments")
    return dict(costs=costs, grads=grads, alphas=alphas, betas=betas, mismatch=mism)


def log_softmax_f32(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    V = x.shape[-1]
    lib().oracle_log_softmax(_f(x), _f(out), x.size // V, V)
    return out


def gather_f32(log_probs, labels, blank=0):
    lp = np.ascontiguousarray(log_probs, dtype=np.float32)
    N, T, U, V = lp.shape
    lab = np.ascontiguousarray(labels, dtype=np.int32).reshape(N, max(U - 1, 0))
    if lab.size == 0:
        lab = np.zeros((N, 1), dtype=np.int32)
    out = np.empty((N, T, U, 2), dtype=np.float32)
    lib().oracle_gather(_f(lp), _i(lab), _f(out), N, T, U, V, int(blank))
    return out


def num_threads():
    return int(lib().oracle_num_threads())


def set_threads(n):
    lib().oracle_set_threads(int(n))


def grad_error_report(hip_pairs, ref, xn, yn, threshold=1e-4):
    """How far gathered gradient pairs (N,T,U,2) of the HIP path are from the oracle's (``ref`` = rnnt_loss_f32's dict on
    the gathered layout), in the terms of the fp32 argument that explains the distance on long lattices.

    A gradient is ``-exp((alpha + beta) + lp - beta00)``: its argument is a difference of numbers of magnitude
    |log-likelihood| (6e3 at T=1500, U=300, V=50), so ONE ulp of the plane values -- 4.9e-4 there -- moves the argument, and
    with it a gradient of magnitude ~1, by that much.  Two fp32 implementations of the same operation order that differ
    in the last bit of a transcendental somewhere along 1800 dependent steps therefore differ by a few ulp OF THE PLANE
    VALUE on the handful of cells that carry the path's mass, and by nothing visible elsewhere.  Returned:
      max_abs, p999                       over all live slots
      cells_above, frac_above             live slots with |delta| > threshold (1e-4: BASELINE.json's fp32 bar)
      max_ulp_of_plane                    max over live slots of |delta| / ulp32(max(|alpha|, |beta|, |alpha + beta|))
      min_plane_magnitude_above           the smallest such plane magnitude among the slots above the threshold
                                          (None if there are none): the claim is that it is >= 2^11
    """
    g = np.asarray(hip_pairs, dtype=np.float64)
    r = np.asarray(ref["grads"], dtype=np.float64)
    N, T, U, _ = g.shape
    t = np.arange(T)[None, :, None]
    u = np.arange(U)[None, None, :]
    xn = np.asarray(xn)
    yn = np.asarray(yn)
    cell = (t < xn[:, None, None]) & (u <= yn[:, None, None])
    live = np.stack([cell, cell & (u < yn[:, None, None])], axis=-1)
    a = np.abs(ref["alphas"].astype(np.float64))
    b = np.abs(ref["betas"].astype(np.float64))
    mag = np.maximum(np.maximum(a, b), np.abs(ref["alphas"].astype(np.float64) + ref["betas"].astype(np.float64)))
    ulp = np.spacing(np.maximum(mag, 1.0).astype(np.float32)).astype(np.float64)[..., None]
    d = np.abs(g - r)
    dl = d[live]
    above = live & (d > threshold)
    n_above = int(above.sum())
    mags_above = np.broadcast_to(mag[..., None], d.shape)[above]
    return {"max_abs": float(dl.max()), "p999": float(np.quantile(dl, 0.999)),
            "cells_above": n_above, "frac_above": n_above / max(int(live.sum()), 1), "threshold": threshold,
            "max_ulp_of_plane": float((d / ulp)[live].max()),
            "min_plane_magnitude_above": float(mags_above.min()) if n_above else None}
