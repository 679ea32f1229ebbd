"""Batch-sharded RNN-T loss across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference has no distributed code at all: utterances are independent
(`blockIdx.z = n`, core.cu:49), so the minibatch shards with NO data-path collective -- every rank
runs the loss on its own slice of (log_probs, labels, lengths) and the gradients w.r.t. its own
log_probs never leave the rank.  The only exchange is the scalar loss: one all-reduce of
(sum of costs, number of utterances[, number of frames]) per step -- 8 or 12 bytes, latency-bound,
independent of the 7x153 GB/s xGMI link budget.  `reduction='none'` uses one all-gather of N/world
floats instead.

All functions work on any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_global: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of a global batch owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(tensors, rank: Optional[int] = None, world: Optional[int] = None):
    """Slice every tensor of a (global) minibatch along dim 0 for this rank."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(tensors[0].shape[0], rank, world)
    return tuple(t[lo:hi].contiguous() for t in tensors)


def reduce_costs(costs: torch.Tensor, reduction: str = "mean", group=None, n_global: Optional[int] = None):
    """Global reduction of per-utterance costs computed on this rank's shard.

    Returns (loss, global_value):
      loss          differentiable tensor to call .backward() on.  It is this rank's share of the
                    global objective (local sum divided by the GLOBAL utterance count for 'mean'),
                    so that summing parameter gradients over ranks (what DDP's gradient all-reduce
                    does, with its 1/world averaging undone, or a plain SUM all-reduce) gives the
                    gradient of the global loss.  No collective runs in backward.
      global_value  detached scalar, identical on every rank: the reduced loss over the whole
                    minibatch (for logging / early stopping).
    For reduction='none' returns (local costs, all-gathered costs of the global batch in rank order).  When the
    batch was sliced with :func:`shard_bounds` / :func:`shard_batch`, pass its size as ``n_global``: every rank then
    knows every shard's size, and the exchange is ONE all-gather with no host synchronisation (without it the sizes
    are gathered first and read back on the host).
    """
    if reduction not in ("mean", "sum", "none", None):
        raise ValueError(f"Unknown reduction method: {reduction}, expected to be one of ['mean', 'sum', 'none']")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if reduction in ("none", None):
        if world == 1:
            return costs, costs.detach()
        if n_global is not None:
            spans = [shard_bounds(n_global, r, world) for r in range(world)]
            sizes = [hi - lo for lo, hi in spans]
            if sizes[dist.get_rank(group)] != costs.shape[0]:
                raise ValueError(f"rank {dist.get_rank(group)} holds {costs.shape[0]} costs, shard_bounds({n_global}, ...)"
                                 f" gives it {sizes[dist.get_rank(group)]}")
        else:
            n = torch.tensor([costs.shape[0]], device=costs.device, dtype=torch.int64)
            sizes = [torch.zeros_like(n) for _ in range(world)]
            dist.all_gather(sizes, n, group=group)
            sizes = [int(s.item()) for s in sizes]
        mx = max(sizes)
        pad = torch.zeros((mx,), dtype=costs.dtype, device=costs.device)
        pad[:costs.shape[0]] = costs.detach()
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        return costs, torch.cat([b[:s] for b, s in zip(bufs, sizes)])
    local_sum = costs.sum()
    stats = torch.stack([local_sum.detach().to(torch.float32),
                         torch.tensor(float(costs.shape[0]), device=costs.device)])
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)   # the path's only exchange
    if reduction == "sum":
        return local_sum, stats[0]
    n_global = stats[1]
    return local_sum / n_global, stats[0] / n_global


def sharded_rnnt_loss(log_probs, labels, frames_lengths, labels_lengths, average_frames=False,
                      reduction="mean", blank=0, gather=False, fastemit_lambda=0.0, group=None, n_global=None):
    """`warp_rnnt.rnnt_loss` on this rank's shard + the global scalar reduction.

    Arguments are this rank's slice of the minibatch (see :func:`shard_batch`).  Returns
    (loss_for_backward, global_loss) as described in :func:`reduce_costs`.
    """
    import warp_rnnt
    costs = warp_rnnt.rnnt_loss(log_probs, labels, frames_lengths, labels_lengths,
                                average_frames=average_frames, reduction="none", blank=blank,
                                gather=gather, fastemit_lambda=fastemit_lambda)
    return reduce_costs(costs, reduction, group, n_global)
