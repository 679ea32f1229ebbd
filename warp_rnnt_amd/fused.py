"""Fused logits -> RNN-T loss (additive API).

The reference's callers compute ``rnnt_loss(F.log_softmax(logits, -1), ..., gather=True)``
(pytorch_binding/benchmark.py:65-70): the dense (N,T,U,V) log-probabilities are written and re-read in
forward, and backward materialises a dense gradient w.r.t. them before the log-softmax backward
turns it into a gradient w.r.t. the logits -- about 28V bytes of HBM traffic per lattice cell.
Here forward reads the logits once (log-softmax + gather fused, 4V+8 B/cell) and backward reads them
once more and writes d(logits) (8V B/cell); log-probabilities never exist in HBM.
"""
from typing import Optional

import torch

from . import ops
from warp_rnnt import _C as _core


class RNNTLossFromLogits(torch.autograd.Function):

    @staticmethod
    def forward(ctx, logits, labels, frames_lengths, labels_lengths, blank=0, fastemit_lambda=0.0):
        _core.check_inputs(logits, labels, frames_lengths, labels_lengths)
        costs, grads = ops.loss(logits, labels, frames_lengths, labels_lengths, ops.IN_LOGITS_DENSE,
                                ops.GRADS_GATHERED_DIAGONAL, blank, fastemit_lambda)
        ctx.save_for_backward(logits, labels, grads)
        ctx.blank = blank
        return costs

    @staticmethod
    def backward(ctx, grads_output):
        logits, labels, grads = ctx.saved_tensors
        go = grads_output.reshape(-1).to(torch.float32).contiguous()
        return ops.logits_backward(logits, labels, grads, go, ctx.blank), None, None, None, None, None


def rnnt_loss_from_logits(logits: torch.Tensor, labels: torch.Tensor, frames_lengths: torch.Tensor,
                          labels_lengths: torch.Tensor, average_frames: bool = False,
                          reduction: Optional[str] = "none", blank: int = 0,
                          fastemit_lambda: float = 0.0) -> torch.Tensor:
    """Same value and gradients as ``warp_rnnt.rnnt_loss(F.log_softmax(logits, -1), ..., gather=True)``
    (arguments as there), without materialising the log-probabilities."""
    assert reduction is None or reduction in ("none", "mean", "sum")
    assert isinstance(blank, int)
    costs = RNNTLossFromLogits.apply(logits, labels, frames_lengths, labels_lengths, blank, fastemit_lambda)
    if average_frames:
        costs = costs / frames_lengths.to(logits)
    if reduction == "none" or reduction is None:
        return costs
    if reduction == "sum":
        return costs.sum()
    return costs.mean()
