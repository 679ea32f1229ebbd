"""Drop-in ``warp_rnnt`` package for AMD MI355X.

Public surface = the reference's (pytorch_binding/warp_rnnt/__init__.py:9-24 ``RNNTLoss``,
:26-54 ``RNNTLossCompact``, :57-143 ``rnnt_loss``): same call signatures, same flag checking
(``AssertionError`` for ill-typed flags or differentiable integer inputs, ``ValueError`` text for
an unknown reduction), same ``average_frames`` / ``reduction`` arithmetic, and the same autograd
contract -- the gradient w.r.t. ``log_probs`` is produced during the forward pass and ``backward``
only scales it by the incoming per-utterance gradient.  All numerics run in the hand-written HIP
kernels of ``warp_rnnt_amd/csrc``; there is no CPU path.
"""
from typing import Optional

import torch

from . import _C as core
from warp_rnnt_amd import _mismatch
from warp_rnnt_amd.functional import LazyLogSoftmax

__version__ = "0.7.0+amd.mi355x"

_REDUCTIONS = ("none", "mean", "sum")


def _per_utterance(t, like):
    """(N,) upstream gradient broadcastable against a per-cell gradient tensor."""
    return t.reshape(-1, *([1] * (like.dim() - 1))).to(like)


class RNNTLoss(torch.autograd.Function):
    """Loss on ``log_probs`` in the layout the native op takes: dense ``(N,T,U,V)``, or the
    2-channel (blank, label) layout when ``blank == -1``.  Six inputs, six gradients
    (the reference returns a seventh, spurious ``None``)."""

    @staticmethod
    def forward(ctx, log_probs, labels, frames_lengths, labels_lengths, blank=0, fastemit_lambda=0.0):
        costs, ctx.grads = core.rnnt_loss(xs=log_probs, ys=labels, xn=frames_lengths, yn=labels_lengths,
                                          blank=blank, fastemit_lambda=fastemit_lambda)
        return costs

    @staticmethod
    def backward(ctx, grads_output):
        _mismatch.poll(ctx.grads.device)      # (the forward's guard, if it fired: a memory read, no synchronisation)
        # out of place: a second backward (retain_graph) sees the same gradient again
        return (ctx.grads * _per_utterance(grads_output, ctx.grads),) + (None,) * 5


class RNNTLossGather(torch.autograd.Function):
    """``gather=True``.  The reference builds an int64 index, calls ``torch.gather`` and lets autograd
    scatter-add the result back into a dense zero tensor (__init__.py:118-128); here the gather, the
    loss and the dense expansion of the gradient are three native kernels and no index exists."""

    @staticmethod
    def forward(ctx, log_probs, labels, frames_lengths, labels_lengths, blank=0, fastemit_lambda=0.0):
        # The reference's `log_probs.gather(dim=3, index)` (__init__.py:126) takes ANY strides and hands the native op a
        # fresh contiguous (N,T,U,2) tensor, so a transposed or sliced joint output works with gather=True there (and
        # raises "xs must be contiguous" with gather=False, binding.cpp:33 -- as it does here).  The native gather reads
        # the dense tensor itself, so the strides are resolved here, where the reference resolved them.
        if log_probs.dim() == 4 and not log_probs.is_contiguous():
            log_probs = log_probs.contiguous()
        costs, pairs_grad = core.rnnt_loss_gather(xs=log_probs, ys=labels, xn=frames_lengths,
                                                  yn=labels_lengths, blank=blank,
                                                  fastemit_lambda=fastemit_lambda)
        ctx.pairs_grad = pairs_grad
        ctx.meta = (labels, frames_lengths, labels_lengths, log_probs.size(3), blank)
        return costs

    @staticmethod
    def backward(ctx, grads_output):
        labels, xn, yn, vocab, blank = ctx.meta
        _mismatch.poll(ctx.pairs_grad.device)
        scale = grads_output.reshape(-1).to(ctx.pairs_grad).contiguous()
        dense = core.rnnt_loss_gather_backward(scale, ctx.pairs_grad, labels, xn, yn, vocab, blank)
        return (dense,) + (None,) * 5


class RNNTLossCompact(torch.autograd.Function):
    """Ragged packed layout: ``log_probs`` is ``(sum_n T_n*(U_n+1), V)``, ``labels`` is ``(sum_n U_n,)``."""

    @staticmethod
    def forward(ctx, log_probs, labels, frames_lengths, labels_lengths, blank=0, fastemit_lambda=0.0,
                enable_grad: bool = True, max_frames=None, max_labels=None):
        costs, pairs_grad, loc = core.rnnt_loss_compact(xs=log_probs, ys=labels, xn=frames_lengths,
                                                        yn=labels_lengths, blank=blank,
                                                        fastemit_lambda=fastemit_lambda,
                                                        required_grad=enable_grad, max_frames=max_frames,
                                                        max_labels=max_labels)
        if enable_grad:
            rows_per_utt = frames_lengths * (labels_lengths + 1)
            ctx.save_for_backward(pairs_grad, loc, torch.cumsum(rows_per_utt, dim=0, dtype=torch.int32))
            ctx.vocab, ctx.blank = log_probs.size(-1), blank
        return costs

    @staticmethod
    def backward(ctx, grads_output):
        pairs_grad, loc, row_ends = ctx.saved_tensors
        dense = core.rnnt_loss_compact_backward(grads_output.contiguous(), pairs_grad, row_ends, loc,
                                                ctx.vocab, ctx.blank)
        return (dense,) + (None,) * 8


def _check_call(average_frames, reduction, blank, gather, labels, frames_lengths, labels_lengths):
    # the reference uses bare asserts (__init__.py:100-107); keep the exception type
    assert isinstance(average_frames, bool) or average_frames is None, "average_frames must be a bool"
    assert reduction in _REDUCTIONS or reduction is None, "reduction must be one of %s" % (_REDUCTIONS,)
    assert isinstance(blank, int), "blank must be an int"
    assert isinstance(gather, bool), "gather must be a bool"
    for t, what in ((labels, "labels"), (frames_lengths, "frames_lengths"), (labels_lengths, "labels_lengths")):
        assert not t.requires_grad, what + " does not require gradients"


def _reduce(costs, reduction):
    if reduction in (None, "none"):
        return costs
    if reduction == "sum":
        return costs.sum()
    if reduction == "mean":
        return costs.mean()
    raise ValueError(
        f"Unknown reduction method: {reduction}, expected to be one of ['mean', 'sum', 'none']")


def rnnt_loss(log_probs: torch.FloatTensor,
              labels: torch.IntTensor,
              frames_lengths: torch.IntTensor,
              labels_lengths: torch.IntTensor,
              average_frames: bool = False,
              reduction: Optional[str] = 'none',
              blank: int = 0,
              gather: bool = False,
              fastemit_lambda: float = 0.0,
              compact: bool = False,
              max_frames: Optional[int] = None,
              max_labels: Optional[int] = None) -> torch.Tensor:
    """RNN-Transducer negative log-likelihood of a minibatch.

    ``log_probs``       fp32, contiguous, on the GPU, already log-softmaxed over the last axis:
                        ``(N, T, U, V)`` (T frames, U = longest label sequence + 1, V symbols incl. blank)
                        or, with ``compact=True``, the ragged ``(sum_n T_n*(U_n+1), V)`` packing.
                        With ``gather=True`` any strides are accepted (as the reference's ``torch.gather`` accepts them),
                        and the lazy result of ``warp_rnnt_amd.functional.log_softmax(logits)`` is fused with: same
                        value and gradients, log-softmax + gather + loss in one read of the logits.
    ``labels``          int32 ``(N, U-1)`` (``(sum_n U_n,)`` when compact).
    ``frames_lengths``  int32 ``(N,)`` -- T_n.
    ``labels_lengths``  int32 ``(N,)`` -- U_n.
    ``average_frames``  divide every utterance's cost by T_n before the reduction.
    ``reduction``       ``'none'`` / ``None`` -> ``(N,)`` costs, ``'sum'``, ``'mean'``.
    ``blank``           vocabulary index of the blank symbol.
    ``gather``          run on the 2-channel (blank, label) view of ``log_probs``; same result, the
                        gradient tensor kept for backward is ``V/2`` times smaller.
    ``fastemit_lambda`` FastEmit weight (arXiv:2010.11148): scales the label gradients by ``1+lambda``.
    ``compact``         the ragged packed layout (the reference's: `__init__.py:109-116`).
    ``max_frames``, ``max_labels``  (not in the reference; compact only) upper bounds of ``frames_lengths`` /
                        ``labels_lengths`` the caller vouches for.  With them the compact path reads nothing back
                        from the device -- no host synchronisation (the reference's has four, this one otherwise one)
                        and the call can be captured into a HIP graph.  In exchange the shape errors the reference raises
                        from the host (a length above the bound, ``labels`` / ``log_probs`` sizes that are not the sums of
                        the lengths) cannot be raised: such a batch is refused on the device and comes back with NaN costs
                        and zero gradients -- which ``reduction='mean'`` turns into a NaN loss.  Passing them without
                        ``compact=True`` is an error (``ValueError``): the padded layouts need no bounds.
    """
    _check_call(average_frames, reduction, blank, gather, labels, frames_lengths, labels_lengths)
    if not compact and (max_frames is not None or max_labels is not None):
        raise ValueError("max_frames / max_labels are launch bounds of the compact layout: pass compact=True with them")

    if isinstance(log_probs, LazyLogSoftmax):
        # `rnnt_loss(warp_rnnt_amd.functional.log_softmax(logits), ..., gather=True)`: the reference's call shape
        # (benchmark.py:65-70), served by the fused logits -> loss -> d/d logits path: the log-probabilities never
        # materialise.  Everything else the handle is used for makes it an ordinary tensor first.
        if gather and not compact and log_probs.fusable():
            from warp_rnnt_amd.fused import RNNTLossFromLogits
            logits = log_probs.logits
            costs = RNNTLossFromLogits.apply(logits if logits.is_contiguous() else logits.contiguous(), labels,
                                             frames_lengths, labels_lengths, blank, fastemit_lambda)
            if average_frames:
                costs = costs / frames_lengths.to(costs)
            return _reduce(costs, reduction)
        log_probs = log_probs.materialise()

    if compact:
        wants_grad = log_probs.requires_grad and torch.is_grad_enabled()
        costs = RNNTLossCompact.apply(log_probs.float(), labels, frames_lengths, labels_lengths, blank,
                                      fastemit_lambda, wants_grad, max_frames, max_labels)
    else:
        fn = RNNTLossGather if gather else RNNTLoss
        costs = fn.apply(log_probs, labels, frames_lengths, labels_lengths, blank, fastemit_lambda)

    if average_frames:
        costs = costs / frames_lengths.to(log_probs)
    return _reduce(costs, reduction)
