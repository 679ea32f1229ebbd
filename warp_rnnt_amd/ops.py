"""Torch-tensor front ends of the native entry points (device memory + streams only)."""
import os
import warnings

import torch

from . import _lib, _mismatch
from ._lib import (GRADS_DENSE, GRADS_GATHERED, GRADS_GATHERED_DIAGONAL, GRADS_NONE,  # noqa: F401
                   IN_LOG_PROBS_DENSE, IN_LOG_PROBS_GATHERED, IN_LOGITS_DENSE, STATUS_NAMES)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _check(status):
    if status != 0:
        # same text as the reference's TORCH_CHECK (binding.cpp:102-103)
        raise RuntimeError("rnnt_loss status " + str(status) +
                           " (" + STATUS_NAMES.get(status, "?") + ")")


def _mismatch_policy():
    """WARP_RNNT_AMD_CHECK_MISMATCH = warn | raise: read the guard flags back after every loss call
    (one host synchronisation: exact and immediate).  Unset (default): no read-back; the counterpart of the
    reference's device-side WARNING printf (core_gather.cu:345-349) is the sticky per-device word the gradient kernel
    writes and _mismatch.poll() turns into a RuntimeWarning at the next call or backward.  off: neither."""
    return os.environ.get("WARP_RNNT_AMD_CHECK_MISMATCH", "").lower()


def loss(input, labels, xn, yn, input_kind, grads_kind, blank=0, fastemit_lambda=0.0, return_mismatch=False):
    """costs (N,), grads (layout per grads_kind; None for GRADS_NONE) [, mismatch (N,) int32].
    Tensors must be validated by the caller (contiguous, fp32/int32, same GPU).  ``mismatch[n]`` is 1
    where the forward/backward consistency guard zeroed an utterance's gradients (or its lengths were
    out of range); without asking for it the same event surfaces as a RuntimeWarning a little later, with no
    synchronisation (warp_rnnt_amd/_mismatch.py)."""
    L = _lib.load()
    N, T, U, V = input.shape
    dev = input.device
    if input_kind == IN_LOG_PROBS_GATHERED:
        blank = 0          # channel 0 of the 2-channel layout; the caller's blank is -1 by convention
    elif not 0 <= blank < V:
        raise RuntimeError(f"rnnt_loss status 5 (RNNT_STATUS_INVALID_ARGUMENT): blank={blank} is not a "
                           f"vocabulary index of xs (V={V})")
    with torch.cuda.device(dev):
        costs = torch.empty((N,), dtype=torch.float32, device=dev)
        if grads_kind == GRADS_DENSE:
            grads = torch.empty_like(input)
        elif grads_kind == GRADS_NONE:
            grads = None
        else:
            grads = torch.empty((N, T, U, 2), dtype=torch.float32, device=dev)
        if N == 0:
            if return_mismatch:
                return costs, grads, torch.zeros((0,), dtype=torch.int32, device=dev)
            return costs, grads
        ws_bytes = L.rnnt_amd_workspace_size(N, T, U)
        if ws_bytes == 0:
            raise RuntimeError("rnnt_loss status 5 (RNNT_STATUS_INVALID_ARGUMENT): unsupported sizes "
                               f"N={N} T={T} U={U}")
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        _mismatch.poll(dev)          # (what an EARLIER call's kernels reported; sets the device's words up at first use)
        st = L.rnnt_amd_loss(_stream(dev), ws.data_ptr(), input_kind, input.data_ptr(), _ptr(labels),
                             xn.data_ptr(), yn.data_ptr(), costs.data_ptr(), _ptr(grads), grads_kind,
                             N, T, U, V, blank, float(fastemit_lambda))
        _check(st)
        policy = _mismatch_policy()
        if policy not in ("warn", "raise", "1", "on"):
            policy = ""
        if return_mismatch or policy:
            off = L.rnnt_amd_workspace_mismatch_offset(N, T, U)
            mismatch = ws[off:off + 4 * N].view(torch.int32).clone()
            if policy:
                bad = mismatch.nonzero().flatten().tolist()      # host synchronisation (opt-in)
                if bad:
                    msg = (f"rnnt_loss: forward/backward mismatch or invalid lengths for utterance(s) {bad}: "
                           "their gradients are zero (core_gather.cu:341-354)")
                    if policy == "raise":
                        raise RuntimeError(msg)
                    warnings.warn(msg, RuntimeWarning, stacklevel=2)
            if return_mismatch:
                return costs, grads, mismatch
    return costs, grads


def expand_grads(grads_diagonal, labels, xn, yn, grad_costs, V, blank, overwrite=False):
    """Dense (N,T,U,V) d/d log_probs from diagonal-major gathered grads, scaled per utterance."""
    L = _lib.load()
    N, T, U, _ = grads_diagonal.shape
    dev = grads_diagonal.device
    with torch.cuda.device(dev):
        out = torch.empty((N, T, U, V), dtype=torch.float32, device=dev)
        if N == 0:
            return out
        st = L.rnnt_amd_expand_grads(_stream(dev), grads_diagonal.data_ptr(), _ptr(labels), xn.data_ptr(),
                                     yn.data_ptr(), _ptr(grad_costs), out.data_ptr(), N, T, U, V, blank,
                                     1 if overwrite else 0)
        _check(st)
    return out


def _native_binding():
    """warp_rnnt._C_native when it has been built (one compiled call instead of ctypes marshalling)."""
    global _NATIVE
    if _NATIVE is False:
        try:
            from warp_rnnt import _C as _wc
            _NATIVE = _wc._native
        except ImportError:
            _NATIVE = None
    return _NATIVE


_NATIVE = False


def log_softmax(x, out=None):
    """Row-wise log-softmax over the last axis (fp32, contiguous, GPU). ``out`` may be ``x``."""
    nb = _native_binding()
    if nb is not None:
        return nb.log_softmax(x, out)
    L = _lib.load()
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    V = x.shape[-1]
    rows = x.numel() // max(V, 1)
    with torch.cuda.device(x.device):
        _check(L.rnnt_amd_log_softmax(_stream(x.device), x.data_ptr(), out.data_ptr(), rows, V))
    return out


def gather(log_probs, labels, blank=0):
    """(N,T,U,V) -> (N,T,U,2) [blank, label] pairs (row-major), as the reference wrapper builds them."""
    L = _lib.load()
    N, T, U, V = log_probs.shape
    out = torch.empty((N, T, U, 2), dtype=torch.float32, device=log_probs.device)
    with torch.cuda.device(log_probs.device):
        _check(L.rnnt_amd_gather(_stream(log_probs.device), log_probs.data_ptr(), _ptr(labels),
                                 out.data_ptr(), N, T, U, V, blank))
    return out


def loss_compact(xs, ys, xn, yn, blank=0, fastemit_lambda=0.0, required_grad=True, max_frames=None, max_labels=None):
    """Compact (ragged packed) layout: xs (STU,V), ys (sum yn,), xn/yn (N,).
    Returns (costs (N,), grads (STU,2) or None, loc (STU,) int64).

    Without bounds: one host synchronisation (the maxima of the lengths size the launches, and the shape checks of the
    reference's binding need the sums; the reference does four).  With ``max_frames >= max(xn)`` and ``max_labels >=
    max(yn)`` supplied by the caller: none -- offsets, maxima and checks stay on the device (``rnnt_amd_loss_compact_
    bounded``), the call can be captured into a HIP graph; a batch that does not fit the bounds or the tensors' sizes
    comes back with NaN costs and zero gradients instead of an exception."""
    L = _lib.load()
    dev = xs.device
    N = xn.shape[0]
    STU, V = xs.shape
    if (max_frames is None) != (max_labels is None):
        raise ValueError("max_frames and max_labels go together")
    _mismatch.poll(dev)
    with torch.cuda.device(dev):
        costs = torch.empty((N,), dtype=torch.float32, device=dev)
        loc = torch.empty((STU,), dtype=torch.int64, device=dev)
        grads = torch.empty((STU, 2), dtype=torch.float32, device=dev) if required_grad else None
        if N == 0:
            return costs, grads, loc
        if max_frames is not None:
            tmax, umax = int(max_frames), int(max_labels) + 1
            if tmax < 1 or umax < 1:
                raise ValueError("max_frames >= 1 and max_labels >= 0 expected")
            ws_bytes = L.rnnt_amd_workspace_size_compact_bounded(N, STU, tmax, umax)
            if ws_bytes == 0:
                raise RuntimeError("rnnt_loss status 5 (RNNT_STATUS_INVALID_ARGUMENT): unsupported sizes")
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
            _check(L.rnnt_amd_loss_compact_bounded(_stream(dev), ws.data_ptr(), xs.data_ptr(), _ptr(ys), ys.numel(),
                                                   xn.data_ptr(), yn.data_ptr(), costs.data_ptr(), _ptr(grads),
                                                   loc.data_ptr(), N, STU, tmax, umax, V, blank, float(fastemit_lambda)))
            return costs, grads, loc
        offs = torch.empty((N + 1 + 4,), dtype=torch.int64, device=dev)    # offsets + the 4 stats
        loffs = torch.empty((N + 1,), dtype=torch.int32, device=dev)
        _check(L.rnnt_amd_compact_offsets(_stream(dev), xn.data_ptr(), yn.data_ptr(), N, offs.data_ptr(),
                                          loffs.data_ptr(), offs[N + 1:].data_ptr()))
        stats = offs[N + 1:].tolist()                                       # the one host sync
        stu_chk, su, tmax, umax = int(stats[0]), int(stats[1]), int(stats[2]), int(stats[3]) + 1
        if ys.numel() != su:
            raise RuntimeError("ys shape must be equal to (sum(yn), )")
        if STU != stu_chk:
            raise RuntimeError("xs shape mismatch with (\\sum{xn*(yn+1)}, )")
        ws_bytes = L.rnnt_amd_workspace_size_compact(N, STU, tmax, umax)
        if ws_bytes == 0:
            raise RuntimeError("rnnt_loss status 5 (RNNT_STATUS_INVALID_ARGUMENT): unsupported sizes")
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        st = L.rnnt_amd_loss_compact(_stream(dev), ws.data_ptr(), xs.data_ptr(), _ptr(ys), xn.data_ptr(),
                                     yn.data_ptr(), offs.data_ptr(), loffs.data_ptr(), costs.data_ptr(),
                                     _ptr(grads), loc.data_ptr(), N, STU, tmax, umax, V, blank,
                                     float(fastemit_lambda))
        _check(st)
    return costs, grads, loc


def compact_scatter_grads(grad_cost, grad_xs, cum_lens, loc, V, blank):
    """(STU,V) gradient rows from the (STU,2) pairs (reference: rnnt_loss_compact_backward)."""
    L = _lib.load()
    dev = grad_xs.device
    STU = grad_xs.shape[0]
    N = grad_cost.shape[0]
    with torch.cuda.device(dev):
        out = torch.empty((STU, V), dtype=torch.float32, device=dev)
        if STU == 0:
            return out
        _check(L.rnnt_amd_compact_scatter_grads(_stream(dev), grad_cost.data_ptr(), grad_xs.data_ptr(),
                                                loc.data_ptr(), cum_lens.data_ptr(), out.data_ptr(), STU, N,
                                                int(V), int(blank)))
    return out


def logits_backward(logits, labels, grads_diagonal, grad_costs, blank=0, out=None):
    """d(sum_n grad_costs[n]*cost[n]) / d(logits) for the fused RNNT_IN_LOGITS_DENSE path."""
    L = _lib.load()
    N, T, U, V = logits.shape
    dev = logits.device
    with torch.cuda.device(dev):
        if out is None:
            out = torch.empty_like(logits)
        if N == 0:
            return out
        _check(L.rnnt_amd_logits_backward(_stream(dev), logits.data_ptr(), _ptr(labels),
                                          grads_diagonal.data_ptr(), _ptr(grad_costs), out.data_ptr(),
                                          N, T, U, V, blank))
    return out


def log_softmax_backward(grad_out, out, grad_in=None):
    """grad_in = grad_out - exp(out) * rowsum(grad_out) (rows = everything but the last axis)."""
    L = _lib.load()
    assert grad_out.is_cuda and grad_out.dtype == torch.float32 and grad_out.is_contiguous()
    assert out.is_contiguous() and out.shape == grad_out.shape and out.dtype == torch.float32
    if grad_in is None:
        grad_in = torch.empty_like(grad_out)
    V = out.shape[-1]
    rows = out.numel() // max(V, 1)
    with torch.cuda.device(out.device):
        _check(L.rnnt_amd_log_softmax_backward(_stream(out.device), grad_out.data_ptr(), out.data_ptr(),
                                               grad_in.data_ptr(), rows, V))
    return grad_in
