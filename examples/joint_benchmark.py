#!/usr/bin/env python
"""Joint network + RNN-T loss training step on MI355X.

What the reference's second benchmark measures (pytorch_binding/benchmark2.py:93-164): the loss together
with a typical joint network -- encoder frames f (N,T,H) and predictor states g (N,U+1,H) are added by
broadcast, passed through tanh + Linear(H,V), and the result goes into the loss; timing covers forward and
(unless --fwd-only) backward to f and g.  Same CLI, same shape grid, same length generator; differences:

  * timing is HIP-event based with warm-up iterations (the reference uses the torch profiler table only;
    pass --profile for that table as well);
  * `--loss warp-rnnt-lazy` keeps the reference's call shape -- `rnnt_loss(log_softmax(joint_out), ..., gather=True)` --
    with `log_softmax` imported from `warp_rnnt_amd.functional`: the same fused path, no change of signature.
  * `--loss warp-rnnt-fused` feeds the joint's *logits* to `rnnt_loss_from_logits`, so the (N,T,U,V)
    log-probabilities and their gradient never exist -- the call chain the reference cannot express;
  * `--ddp` wraps the joint in DistributedDataParallel (one process per GPU, RCCL) and shards the batch:

        python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
            examples/joint_benchmark.py --loss warp-rnnt-fused --ddp

    Each rank runs the loss on its own utterances; the only loss-side exchange is the scalar all-reduce in
    `warp_rnnt_amd.distributed.reduce_costs`; DDP all-reduces the joint's parameter gradients.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402

LOSSES = ("warp-rnnt", "warp-rnnt-gather", "warp-rnnt-compact", "warp-rnnt-fused", "warp-rnnt-lazy")
GRID = [(150, 40, 28), (150, 20, 5000), (1500, 300, 50)]      # benchmark2.py:133 (U = label count here)
BATCHES = [1, 16, 32, 64, 128]


class JointNetwork(nn.Module):
    """tanh(f_t + g_u) -> Linear(H, V).  `packed=True` emits the ragged (sum_n T_n*(U_n+1), V) layout of
    rnnt_loss(compact=True); `log_softmax=False` returns logits (for the fused loss); `log_softmax="lazy"` normalises with
    warp_rnnt_amd.functional.log_softmax -- the reference's call shape, which rnnt_loss(gather=True) then fuses with."""

    def __init__(self, hidden: int, vocab: int, packed: bool = False, log_softmax: bool = True):
        super().__init__()
        self.proj = nn.Linear(hidden, vocab)
        self.packed, self.normalise = packed, log_softmax

    def forward(self, f, g, f_len=None, g_len=None):
        if self.packed:
            H = f.size(-1)
            fl, gl = f_len.tolist(), g_len.tolist()          # one read-back each, not one per utterance
            rows = [(f[n, :fl[n]].unsqueeze(1) + g[n, :gl[n] + 1].unsqueeze(0)).reshape(-1, H)
                    for n in range(f.size(0))]
            x = torch.cat(rows, dim=0)
        else:
            x = f.unsqueeze(2) + g.unsqueeze(1)
        out = self.proj(torch.tanh(x))
        if self.normalise == "lazy":
            from warp_rnnt_amd.functional import log_softmax
            return log_softmax(out)
        return out.log_softmax(dim=-1) if self.normalise else out


def make_batch(N, T, U, V, H, random_length, device):
    f = torch.randn(N, T, H, device=device)
    g = torch.randn(N, U + 1, H, device=device)
    ys = torch.randint(1, V, (N, U), dtype=torch.int, device=device)
    if random_length:
        f_len = torch.randint(T // 2, T + 1, (N,), dtype=torch.int, device=device)
        g_len = torch.randint(U // 2, U + 1, (N,), dtype=torch.int, device=device)
        f_len += T - f_len.max()
        g_len += U - g_len.max()
    else:
        f_len = torch.full((N,), T, dtype=torch.int, device=device)
        g_len = torch.full((N,), U, dtype=torch.int, device=device)
    return f, g, ys, f_len, g_len


def pick_loss(name):
    from warp_rnnt import rnnt_loss
    from warp_rnnt_amd.fused import rnnt_loss_from_logits
    if name == "warp-rnnt":
        return lambda xs, ys, xn, yn: rnnt_loss(xs, ys, xn, yn, gather=False)
    if name == "warp-rnnt-gather":
        return lambda xs, ys, xn, yn: rnnt_loss(xs, ys, xn, yn, gather=True)
    if name == "warp-rnnt-compact":
        def compact(xs, ys, xn, yn):
            yl = yn.tolist()
            packed = torch.cat([ys[n, :yl[n]] for n in range(ys.size(0))])
            # the padded tensors' own sizes are launch bounds: the loss itself then needs no read-back
            return rnnt_loss(xs, packed, xn, yn, compact=True, max_frames=int(max(xn.tolist())), max_labels=ys.size(1))
        return compact
    if name == "warp-rnnt-fused":
        return lambda xs, ys, xn, yn: rnnt_loss_from_logits(xs, ys, xn, yn)
    if name == "warp-rnnt-lazy":       # xs is the lazy handle of warp_rnnt_amd.functional.log_softmax: the same call as "gather"
        return lambda xs, ys, xn, yn: rnnt_loss(xs, ys, xn, yn, gather=True)
    raise ValueError(f"Unrecognized type of loss:{name}")


def bytes_needed(N, T, U, V, H, loss, fwd_only):
    """Rough HBM need of one step: the dense (N,T,U+1,·) tensors autograd keeps alive."""
    cells = N * T * (U + 1)
    dense_v = {"warp-rnnt": 4, "warp-rnnt-gather": 3, "warp-rnnt-compact": 3, "warp-rnnt-fused": 2, "warp-rnnt-lazy": 2}[loss]
    if fwd_only:
        dense_v -= 1
    return 4 * cells * (dense_v * V + 2 * H + 16)


def main():
    p = argparse.ArgumentParser(description="Benchmark RNN-T loss together with a joint network")
    p.add_argument("--loss", type=str, required=True, choices=LOSSES, help="The target implementation")
    p.add_argument("--random-length", action="store_true", default=False, help="The random length")
    p.add_argument("--fwd-only", action="store_true", default=False,
                   help="forward pass only; otherwise forward and backward are timed")
    p.add_argument("--hidden", type=int, default=512)
    p.add_argument("--iters", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--batches", type=int, nargs="*", default=BATCHES)
    p.add_argument("--shapes", type=str, nargs="*", default=None, help="T,U,V triples, e.g. 1500,300,50")
    p.add_argument("--profile", action="store_true", help="also print the torch profiler table")
    p.add_argument("--ddp", action="store_true", help="shard each batch over the ranks of torch.distributed.run")
    args = p.parse_args()

    rank, world = 0, 1
    if args.ddp:
        local = int(os.environ.get("LOCAL_RANK", 0))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        rank, world = dist.get_rank(), dist.get_world_size()
    device = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(rank)
    run_loss = pick_loss(args.loss)
    grid = GRID if not args.shapes else [tuple(int(v) for v in s.split(",")) for s in args.shapes]
    from warp_rnnt_amd.distributed import reduce_costs, shard_bounds

    if rank == 0:
        print(f"# loss={args.loss} H={args.hidden} random_length={args.random_length} fwd_only={args.fwd_only} "
              f"world={world}")
        print("| T | U | V | N (global) | ms / step | peak HBM MB |")
        print("|---|---|---|---|---|---|")
    free = torch.cuda.mem_get_info()[0]
    for T, U, V in grid:
        for N_global in args.batches:
            lo, hi = shard_bounds(N_global, rank, world)
            N = hi - lo
            if N == 0 or bytes_needed(N, T, U, V, args.hidden, args.loss, args.fwd_only) > 0.8 * free:
                if rank == 0:
                    print(f"| {T} | {U} | {V} | {N_global} | skipped (shard empty or larger than HBM) | |")
                continue
            joint = JointNetwork(args.hidden, V, packed=args.loss.endswith("compact"),
                                 log_softmax="lazy" if args.loss.endswith("lazy") else not args.loss.endswith("fused")).to(device)
            if args.fwd_only:
                joint.requires_grad_(False)
            model = nn.parallel.DistributedDataParallel(joint, device_ids=[device.index]) \
                if (args.ddp and not args.fwd_only) else joint
            f, g, ys, f_len, g_len = make_batch(N, T, U, V, args.hidden, args.random_length, device)
            if not args.fwd_only:
                f.requires_grad_(True)
                g.requires_grad_(True)

            def step():
                xs = model(f, g, f_len, g_len)
                costs = run_loss(xs, ys, f_len, g_len)
                loss, _ = reduce_costs(costs, "mean") if args.ddp else (costs.mean(), None)
                if not args.fwd_only:
                    loss.backward()
                    joint.zero_grad(set_to_none=True)
                    f.grad = g.grad = None
                return loss

            torch.cuda.reset_peak_memory_stats()
            for _ in range(args.warmup):
                step()
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if args.ddp:
                dist.barrier()
            torch.cuda.synchronize()
            start.record()
            for _ in range(args.iters):
                step()
            stop.record()
            torch.cuda.synchronize()
            ms = torch.tensor([start.elapsed_time(stop) / args.iters], device=device)
            if args.ddp:
                dist.all_reduce(ms, op=dist.ReduceOp.MAX)
            peak = torch.cuda.max_memory_allocated() / 1e6
            if rank == 0:
                print(f"| {T} | {U} | {V} | {N_global} | {ms.item():.3f} | {peak:.0f} |", flush=True)
            if args.profile and rank == 0:
                from torch.profiler import ProfilerActivity, profile
                with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                    for _ in range(3):
                        step()
                    torch.cuda.synchronize()
                print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=10))
            del f, g, ys, f_len, g_len, joint, model
            torch.cuda.empty_cache()
    if args.ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
