"""Drop-in replacement for the ``warp_rnnt`` package of 1ytic/warp-rnnt on AMD MI355X.

Mirrors pytorch_binding/warp_rnnt/__init__.py:9-24,57-143: ``rnnt_loss`` keeps the signature,
the assert/exception behaviour, ``average_frames`` / ``reduction`` semantics and the autograd
contract (gradients w.r.t. ``log_probs`` are computed in the forward pass, ``backward`` scales them by
the incoming per-utterance gradient).  Everything numeric runs in hand-written HIP kernels
(warp_rnnt_amd/csrc); see DESIGN.md.
"""
from typing import Optional

import torch

from . import _C as core

__version__ = "0.7.0+amd.mi355x"


class RNNTLoss(torch.autograd.Function):
    """log_probs in the layout the native op takes (dense, or gathered with blank=-1)."""

    @staticmethod
    def forward(ctx, log_probs, labels, frames_lengths, labels_lengths, blank=0, fastemit_lambda=0.0):
        costs, grads = core.rnnt_loss(
            xs=log_probs, ys=labels,
            xn=frames_lengths, yn=labels_lengths,
            blank=blank,
            fastemit_lambda=fastemit_lambda,
        )
        ctx.grads = grads
        return costs

    @staticmethod
    def backward(ctx, grads_output):
        grads_output = grads_output.view(-1, 1, 1, 1).to(ctx.grads)
        return ctx.grads * grads_output, None, None, None, None, None


class RNNTLossGather(torch.autograd.Function):
    """``gather=True``: dense log_probs in; the gather prologue, the loss and the scatter
    backward are native kernels (no int64 index tensor, no dense zero-fill + scatter_add)."""

    @staticmethod
    def forward(ctx, log_probs, labels, frames_lengths, labels_lengths, blank=0, fastemit_lambda=0.0):
        costs, grads = core.rnnt_loss_gather(
            xs=log_probs, ys=labels,
            xn=frames_lengths, yn=labels_lengths,
            blank=blank,
            fastemit_lambda=fastemit_lambda,
        )
        ctx.grads = grads
        ctx.aux = (labels, frames_lengths, labels_lengths, log_probs.size(3), blank)
        return costs

    @staticmethod
    def backward(ctx, grads_output):
        labels, xn, yn, V, blank = ctx.aux
        go = grads_output.reshape(-1).to(ctx.grads).contiguous()
        dense = core.rnnt_loss_gather_backward(go, ctx.grads, labels, xn, yn, V, blank)
        return dense, None, None, None, None, None


class RNNTLossCompact(torch.autograd.Function):
    """Compact (ragged packed) layout, mirror of __init__.py:26-54."""

    @staticmethod
    def forward(ctx, log_probs, labels, frames_lengths, labels_lengths, blank=0, fastemit_lambda=0.0,
                enable_grad: bool = True):
        costs, grads, loc = core.rnnt_loss_compact(
            xs=log_probs, ys=labels,
            xn=frames_lengths, yn=labels_lengths,
            blank=blank,
            fastemit_lambda=fastemit_lambda,
            required_grad=enable_grad
        )
        if enable_grad:
            cumlen = torch.cumsum(frames_lengths * (labels_lengths + 1), dim=0, dtype=torch.int32)
            ctx.V = log_probs.size(-1)
            ctx.blank = blank
            ctx.save_for_backward(grads, loc, cumlen)
        return costs

    @staticmethod
    def backward(ctx, grads_output):
        grads, loc, cumlen = ctx.saved_tensors
        grads_input = core.rnnt_loss_compact_backward(
            grads_output.contiguous(),
            grads, cumlen,
            loc, ctx.V, ctx.blank
        )
        return grads_input, None, None, None, None, None, None


def rnnt_loss(log_probs: torch.FloatTensor,
              labels: torch.IntTensor,
              frames_lengths: torch.IntTensor,
              labels_lengths: torch.IntTensor,
              average_frames: bool = False,
              reduction: Optional[str] = 'none',
              blank: int = 0,
              gather: bool = False,
              fastemit_lambda: float = 0.0,
              compact: bool = False) -> torch.Tensor:
    """RNN-Transducer loss (same arguments as the reference, __init__.py:57-98).

    Args:
        log_probs: (N, T, U, V) log-probabilities (already log-softmaxed), fp32, contiguous, on GPU.
        labels: (N, U-1) int32 reference labels.
        frames_lengths: (N,) int32 number of frames per utterance.
        labels_lengths: (N,) int32 number of labels per utterance.
        average_frames: divide each utterance's loss by its number of frames.
        reduction: 'none' | 'mean' | 'sum' (None = 'none').
        blank: index of the blank symbol.
        gather: run the lattice on the 2-channel (blank, label) view of ``log_probs``.
        fastemit_lambda: FastEmit regularisation weight (https://arxiv.org/abs/2010.11148).
        compact: ragged packed layout: log_probs (STU, V) with STU = sum(frames_lengths*(labels_lengths+1)),
            labels (sum(labels_lengths),).
    """
    assert average_frames is None or isinstance(average_frames, bool)
    assert reduction is None or reduction in ("none", "mean", "sum")
    assert isinstance(blank, int)
    assert isinstance(gather, bool)

    assert not labels.requires_grad, "labels does not require gradients"
    assert not frames_lengths.requires_grad, "frames_lengths does not require gradients"
    assert not labels_lengths.requires_grad, "labels_lengths does not require gradients"

    if compact:
        costs = RNNTLossCompact.apply(
            log_probs.float(),
            labels, frames_lengths,
            labels_lengths, blank,
            fastemit_lambda,
            (log_probs.requires_grad and torch.is_grad_enabled())
        )
    elif gather:
        costs = RNNTLossGather.apply(log_probs, labels, frames_lengths, labels_lengths, blank, fastemit_lambda)
    else:
        costs = RNNTLoss.apply(log_probs, labels, frames_lengths, labels_lengths, blank, fastemit_lambda)

    if average_frames:
        costs = costs / frames_lengths.to(log_probs)

    if reduction == "none" or reduction is None:
        return costs
    elif reduction == "sum":
        return costs.sum()
    elif reduction == "mean":
        return costs.mean()
    else:
        raise ValueError(
            f"Unknown reduction method: {reduction}, expected to be one of ['mean', 'sum', 'none']")
