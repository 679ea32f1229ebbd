"""`log_softmax` for callers of the reference API -- ``rnnt_loss(log_softmax(logits), ..., gather=True)``
(pytorch_binding/benchmark.py:65-70, README.md:59) -- that lets the loss fuse with it WITHOUT a change of signature.

``log_softmax(x)`` returns a LAZY handle: a tensor subclass that looks like the ``(N,T,U,V)`` log-probabilities (shape,
dtype, device, autograd history) but has not computed them.

  * ``warp_rnnt.rnnt_loss(handle, labels, frames_lengths, labels_lengths, gather=True, ...)`` recognises the handle and
    runs the fused entry on the logits behind it (``RNNT_IN_LOGITS_DENSE``: log-softmax + gather in one read of the
    logits, 4V+8 B/cell; backward ``rnnt_amd_logits_backward``: one more read and the write of d/d logits, 8V+8 B/cell)
    -- the log-probabilities and their dense gradient never exist in HBM.  Same bits as
    ``warp_rnnt_amd.fused.rnnt_loss_from_logits`` (it IS that path), half the time of the materialised chain at
    N=16, T=1500, U=300, V=50 (forward 0.36-0.41 vs 0.80 ms, training step 0.83 vs 1.75).
  * ANY other consumer -- an arithmetic op, indexing, ``.cpu()``, printing, ``gather=False``, ``compact=True``, a leaf
    handle that itself requires grad -- materialises the log-probabilities once, through the library's streaming
    log-softmax kernel (k_lsm_regs / k_lsm_small / k_lsm_large), and carries on with an ordinary tensor; backward through
    that route is the library's log-softmax backward kernel.  Both routes may be taken on the same handle; their
    gradients add up in ``logits.grad`` as autograd's always do.

``log_softmax(x, lazy=False)`` is the eager function of rounds 2-5 (same kernels, an ordinary tensor at once).
"""
import torch
from torch.utils._pytree import tree_map

from . import ops


class _Cell:
    """The logits (detached) and, once somebody needed them, the log-probabilities.  Shared by the handle and by the
    autograd node behind it (neither refers to the other: no reference cycle)."""
    __slots__ = ("x", "y")

    def __init__(self, x):
        self.x, self.y = x, None

    def value(self):
        if self.y is None:
            self.y = ops.log_softmax(self.x.contiguous())
        return self.y


class LazyLogSoftmax(torch.Tensor):
    """What :func:`log_softmax` returns.  ``materialised`` tells whether the log-probabilities exist yet;
    ``logits`` is the tensor it was made from (with its autograd history)."""

    @staticmethod
    def __new__(cls, cell):
        x = cell.x
        r = torch.Tensor._make_wrapper_subclass(cls, x.shape, dtype=x.dtype, device=x.device, requires_grad=False)
        r._cell = cell
        r._src = None
        return r

    @property
    def materialised(self):
        return self._cell.y is not None

    @property
    def logits(self):
        return self._src

    def fusable(self):
        """True when a loss may bypass this handle and differentiate w.r.t. the logits instead: it is 4-D and it is not
        itself a leaf somebody asked gradients for (``handle.requires_grad_()``: then d/d log-probs is what is wanted)."""
        return self.dim() == 4 and self._src is not None and (self.grad_fn is not None or not self.requires_grad)

    def materialise(self):
        """An ordinary tensor with the log-probabilities, connected to this handle in the autograd graph."""
        return self.view_as(self)

    def __repr__(self):
        return (f"LazyLogSoftmax(shape={tuple(self.shape)}, device={self.device}, "
                f"materialised={self.materialised}, grad_fn={self.grad_fn})")

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        def real(t):
            return t._cell.value() if isinstance(t, LazyLogSoftmax) else t
        return func(*tree_map(real, args), **tree_map(real, kwargs or {}))


class _LazyLogSoftmaxFn(torch.autograd.Function):
    """Forward: a handle, no kernel.  Backward (reached only through the materialised route): the log-softmax backward."""

    @staticmethod
    def forward(ctx, x):
        cell = _Cell(x.detach())
        ctx.cell = cell
        return LazyLogSoftmax(cell)

    @staticmethod
    def backward(ctx, grad_out):
        return ops.log_softmax_backward(grad_out.contiguous(), ctx.cell.value())


class _LogSoftmax(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x):
        y = ops.log_softmax(x.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        y, = ctx.saved_tensors
        return ops.log_softmax_backward(grad_out.contiguous(), y)


def log_softmax(x: torch.Tensor, lazy: bool = True) -> torch.Tensor:
    """``torch.log_softmax(x, dim=-1)`` for fp32 GPU tensors.  ``lazy=True`` (default): the handle described in the
    module docstring -- free until somebody other than ``rnnt_loss(..., gather=True)`` looks at it.  ``lazy=False``:
    computed now (forward and backward in HIP, about 2x the speed of torch's own kernels at V=50 on MI355X)."""
    if x.dtype != torch.float32 or not x.is_cuda:
        raise RuntimeError("warp_rnnt_amd.functional.log_softmax needs an fp32 tensor on the GPU")
    if not lazy:
        return _LogSoftmax.apply(x)
    out = _LazyLogSoftmaxFn.apply(x)
    out._src = x
    return out
