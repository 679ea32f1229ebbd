"""Native-op module of the drop-in package: same functions, keyword names, check order and
error texts as the reference's pybind module ``warp_rnnt._C``
(pytorch_binding/binding.cpp:28-106, 249-254), implemented on the C ABI of
libwarp_rnnt_amd.so through ctypes.  There is no CPU path: tensors must live on a GPU
("CUDA" device type under PyTorch-ROCm) and the HIP library must be built.
"""
import os
import warnings

import torch

from warp_rnnt_amd import ops as _ops
from warp_rnnt_amd import _mismatch

try:                     # the compiled binding (warp_rnnt_amd/csrc/binding.cpp, built by _build.build_binding)
    from . import _C_native as _native
except ImportError:      # not built: the ctypes path below does the same work, a few tens of microseconds slower
    _native = None
if os.environ.get("WARP_RNNT_AMD_NO_NATIVE_BINDING") or os.environ.get("WARP_RNNT_AMD_LIB"):
    _native = None       # (the compiled module is linked against the in-tree library; another build of the C ABI --
                         #  the A/B variants of _build.VARIANTS -- is reached through the ctypes loader)


def _check_contiguous(x, name):
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _check_float(x, name):
    if x.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a Float tensor")


def _check_int(x, name):
    if x.dtype != torch.int32:
        raise RuntimeError(f"{name} must be a Int tensor")


def _check_cuda(x, name):
    if x.device.type != "cuda":
        raise RuntimeError(f"{name} must be located in the CUDA")


def check_inputs(xs, ys, xn, yn):
    """binding.cpp:32-51 -- contiguity, then dtypes, then device, then shapes."""
    for x, name in ((xs, "xs"), (ys, "ys"), (xn, "xn"), (yn, "yn")):
        _check_contiguous(x, name)
    _check_float(xs, "xs")
    for x, name in ((ys, "ys"), (xn, "xn"), (yn, "yn")):
        _check_int(x, name)
    for x, name in ((xs, "xs"), (ys, "ys"), (xn, "xn"), (yn, "yn")):
        _check_cuda(x, name)
    if xs.dim() != 4:
        raise RuntimeError("xs must have 4 dimensions")
    if xn.numel() != xs.size(0):
        raise RuntimeError("xn shape must be equal (N,)")
    if yn.numel() != xs.size(0):
        raise RuntimeError("yn shape must be equal (N,)")
    if ys.dim() != 2 or xs.size(2) != ys.size(1) + 1:
        raise RuntimeError("ys shape (N, U-1) mismatched with xs (N, T, U, V)")
    for x, name in ((ys, "ys"), (xn, "xn"), (yn, "yn")):
        if x.device != xs.device:
            raise RuntimeError(f"{name} must be on the same device as xs")


def _native_call(fn, xs, ys, xn, yn, blank, fastemit_lambda):
    """One call into the compiled binding.  The forward/backward guard (core_gather.cu:341-354): by default a look at the
    device's sticky diagnostics words before the call (no synchronisation; what earlier kernels reported becomes a
    RuntimeWarning -- warp_rnnt_amd/_mismatch.py); WARP_RNNT_AMD_CHECK_MISMATCH = warn | raise reads this call's own
    flags back instead (one host synchronisation), off does neither."""
    policy = os.environ.get("WARP_RNNT_AMD_CHECK_MISMATCH", "").lower()
    exact = policy in ("warn", "raise", "1", "on")
    if xs.is_cuda:
        _mismatch.poll(xs.device)
    costs, grads, mismatch = fn(xs, ys, xn, yn, blank, fastemit_lambda, exact)
    if exact:
        bad = mismatch.nonzero().flatten().tolist()          # host synchronisation (opt-in)
        if bad:
            msg = (f"rnnt_loss: forward/backward mismatch or invalid lengths for utterance(s) {bad}: "
                   "their gradients are zero (core_gather.cu:341-354)")
            if policy == "raise":
                raise RuntimeError(msg)
            warnings.warn(msg, RuntimeWarning, stacklevel=3)
    return costs, grads


def rnnt_loss(xs, ys, xn, yn, blank=0, fastemit_lambda=0.0):
    """(costs (N,), grads like xs).  blank == -1 selects the gathered (N,T,U,2) layout."""
    if _native is not None:
        return _native_call(_native.rnnt_loss, xs, ys, xn, yn, blank, fastemit_lambda)
    check_inputs(xs, ys, xn, yn)
    if blank == -1:
        if xs.size(3) != 2:
            raise RuntimeError("xs must have values only for blank and label")
        return _ops.loss(xs, None, xn, yn, _ops.IN_LOG_PROBS_GATHERED, _ops.GRADS_GATHERED,
                         -1, fastemit_lambda)
    return _ops.loss(xs, ys, xn, yn, _ops.IN_LOG_PROBS_DENSE, _ops.GRADS_DENSE, blank, fastemit_lambda)


def rnnt_loss_gather(xs, ys, xn, yn, blank=0, fastemit_lambda=0.0):
    """Native form of the wrapper's ``gather=True`` branch: dense log-probs in, costs and the
    (opaque, diagonal-major) gathered gradients out; feed those to :func:`rnnt_loss_gather_backward`."""
    if _native is not None:
        return _native_call(_native.rnnt_loss_gather, xs, ys, xn, yn, blank, fastemit_lambda)
    check_inputs(xs, ys, xn, yn)
    return _ops.loss(xs, ys, xn, yn, _ops.IN_LOG_PROBS_DENSE, _ops.GRADS_GATHERED_DIAGONAL,
                     blank, fastemit_lambda)


def rnnt_loss_gather_backward(grad_costs, grads_diagonal, ys, xn, yn, V, blank=0):
    """d loss / d log_probs (N,T,U,V) = scatter-add of the gathered grads times grad_costs[n]."""
    if _native is not None:
        return _native.rnnt_loss_gather_backward(grad_costs, grads_diagonal, ys, xn, yn, V, blank)
    return _ops.expand_grads(grads_diagonal, ys, xn, yn, grad_costs, V, blank, overwrite=False)


def rnnt_loss_compact(xs, ys, xn, yn, blank=0, fastemit_lambda=0.0, required_grad=True, max_frames=None,
                      max_labels=None):
    """binding.cpp:109-207: (costs (N,), grads (STU,2), loc (STU,) int64) for the compact layout
    (xs (STU,V) with STU = sum(xn*(yn+1)), ys (sum(yn),)).  Same check order and messages.
    ``max_frames`` / ``max_labels`` (extension): launch bounds the caller vouches for -- no host synchronisation, the
    op can be captured into a HIP graph; see :func:`warp_rnnt_amd.ops.loss_compact`."""
    if _native is not None:
        return _native.rnnt_loss_compact(xs, ys, xn, yn, blank, fastemit_lambda, required_grad,
                                                 -1 if max_frames is None else int(max_frames),
                                                 -1 if max_labels is None else int(max_labels))
    for x, name in ((xs, "xs"), (ys, "ys"), (xn, "xn"), (yn, "yn")):
        _check_contiguous(x, name)
    _check_float(xs, "xs")
    for x, name in ((ys, "ys"), (xn, "xn"), (yn, "yn")):
        _check_int(x, name)
    for x, name in ((xs, "xs"), (ys, "ys"), (xn, "xn"), (yn, "yn")):
        _check_cuda(x, name)
    if xs.dim() != 2:
        raise RuntimeError("xs must have 2 dimensions")
    if xn.size(0) != yn.size(0):
        raise RuntimeError("xn and yn shape must be equal (N,)")
    costs, grads, loc = _ops.loss_compact(xs, ys, xn, yn, blank, fastemit_lambda, required_grad, max_frames, max_labels)
    if grads is None:
        grads = costs.new_empty((0, 2))    # the reference aliases an unused buffer here
    return costs, grads, loc


def rnnt_loss_compact_backward(grad_cost, grad_xs, cum_lens, loc, V, blank):
    """binding.cpp:209-247: scatter the (STU,2) gradients, scaled per utterance, into (STU,V)."""
    if _native is not None:
        return _native.rnnt_loss_compact_backward(grad_cost, grad_xs, cum_lens, loc, int(V), int(blank))
    _check_contiguous(grad_cost, "grad_cost")
    _check_contiguous(grad_xs, "grad_xs")
    _check_contiguous(loc, "loc")
    _check_float(grad_cost, "grad_cost")
    _check_float(grad_xs, "grad_xs")
    if loc.dtype != torch.int64:
        raise RuntimeError("loc must be a Long tensor")
    for x, name in ((grad_cost, "grad_cost"), (grad_xs, "grad_xs"), (cum_lens, "cum_lens"), (loc, "loc")):
        _check_cuda(x, name)
    if grad_cost.dim() != 1:
        raise RuntimeError("grad_cost must have 1 dimensions")
    if grad_xs.dim() != 2:
        raise RuntimeError("grad must have 2 dimensions")
    if grad_xs.size(0) != loc.size(0):
        raise RuntimeError("grad and loc must be equal in dim=0")
    if cum_lens.dtype != torch.int32 or not cum_lens.is_contiguous():
        raise RuntimeError("cum_lens must be a contiguous Int tensor")
    return _ops.compact_scatter_grads(grad_cost, grad_xs, cum_lens, loc, V, blank)
