#!/usr/bin/env python
"""The reference's benchmark grid on MI355X (what produced the README table, README.md:35-53).

Same protocol and CLI as pytorch_binding/benchmark.py:9-50,53-103: for every (T,U,V) x N, E iterations
of fresh N(0,1) logits (torch.manual_seed(N)), labels in [1,V), full or random lengths; timed region =
loss(xs, ys, xn, yn) between two device synchronisations, wall clock, mean in ms -- with the one
difference the survey asked for: `--warmup` untimed iterations first (default 1; the reference has 0).

    python tools/benchmark_table.py --loss warp-rnnt-gather [--random_length] [--markdown out.md]

--loss: warp-rnnt | warp-rnnt-gather            rnnt_loss(log_softmax(xs), ..., gather=False|True)
        warp-rnnt-compact                       rnnt_loss(log_softmax(xs) packed, ..., compact=True)
        warp-rnnt-fused                         rnnt_loss_from_logits(xs, ...) (no counterpart in the reference)
        warp-rnnt-lazy                          rnnt_loss(warp_rnnt_amd.functional.log_softmax(xs), ..., gather=True): the
                                                reference's call shape, fused
        torch-log-softmax-gather                F.log_softmax from torch + rnnt_loss(gather=True)
"""
import argparse
import os
import sys
from timeit import default_timer as timer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

GRID = [(100, 150, 40, 28), (50, 150, 20, 5000), (10, 1500, 300, 50)]     # benchmark.py:85
BATCHES = [1, 16, 32, 64, 128]                                            # benchmark.py:86
README_MS = {   # README.md:38-53  (gather=False, gather=True) on RTX 2070 Super
    (150, 40, 28): {1: (0.50, 0.54), 16: (1.79, 1.72), 32: (3.09, 2.94), 64: (5.83, 5.54), 128: (11.30, 10.74)},
    (150, 20, 5000): {1: (0.95, 0.80), 16: (8.74, 6.24), 32: (17.26, 12.35), 64: (None, None), 128: (None, None)},
    (1500, 300, 50): {1: (5.89, 4.99), 16: (95.46, 78.88), 32: (None, 157.86), 64: (None, None), 128: (None, None)},
}


def run_benchmark(loss, E, N, T, U, V, random_length=False, warmup=1, device="cuda:0"):
    torch.manual_seed(N)
    elapsed = 0.0
    for i in range(E + warmup):
        xs = torch.randn((N, T, U, V), dtype=torch.float32, device=device, requires_grad=True)
        ys = torch.randint(1, V, (N, U - 1), dtype=torch.int, device=device)
        if random_length:
            xn = torch.randint(T // 2, T + 1, (N,), dtype=torch.int, device=device)
            yn = torch.randint(U // 2, U, (N,), dtype=torch.int, device=device)
            xn = xn + T - xn.max()
            yn = yn + U - 1 - yn.max()
        else:
            xn = torch.ones((N,), dtype=torch.int, device=device) * T
            yn = torch.ones((N,), dtype=torch.int, device=device) * (U - 1)
        if hasattr(loss, "prepare"):      # data layout work that is not part of the loss (untimed)
            xs, ys = loss.prepare(xs, ys, xn, yn)
        torch.cuda.synchronize(device)
        t = timer()
        costs = loss(xs, ys, xn, yn)
        torch.cuda.synchronize(device)
        if i >= warmup:
            elapsed += timer() - t
        del xs, ys, xn, yn, costs
    return elapsed * 1000 / E


def main():
    p = argparse.ArgumentParser(description="Benchmark RNN-T loss implementation")
    p.add_argument("--loss", type=str, required=True)
    p.add_argument("--device", type=int, default=0, help="GPU index (benchmark.py:57)")
    p.add_argument("--random_length", action="store_true")
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--markdown", type=str, default=None)
    p.add_argument("--grid", type=str, default=None,
                   help="E,T,U,V[;E,T,U,V...] instead of the reference's three shapes (tests use a tiny one)")
    p.add_argument("--batches", type=str, default=None, help="comma-separated N instead of 1,16,32,64,128")
    a = p.parse_args()
    grid = [tuple(int(x) for x in g.split(",")) for g in a.grid.split(";")] if a.grid else GRID
    batches = [int(x) for x in a.batches.split(",")] if a.batches else BATCHES
    device = torch.device("cuda", a.device)
    torch.cuda.set_device(device)
    import warp_rnnt
    from warp_rnnt_amd import ops
    from warp_rnnt_amd.fused import rnnt_loss_from_logits
    if a.loss == "warp-rnnt":
        def run_loss(xs, ys, xn, yn):
            return warp_rnnt.rnnt_loss(ops.log_softmax(xs.detach()), ys, xn, yn, gather=False)
    elif a.loss == "warp-rnnt-gather":
        def run_loss(xs, ys, xn, yn):
            return warp_rnnt.rnnt_loss(ops.log_softmax(xs.detach()), ys, xn, yn, gather=True)
    elif a.loss == "torch-log-softmax-gather":
        def run_loss(xs, ys, xn, yn):
            return warp_rnnt.rnnt_loss(torch.log_softmax(xs, -1), ys, xn, yn, gather=True)
    elif a.loss == "warp-rnnt-compact":
        def run_loss(xs, ys, xn, yn):       # xs (sum T_n*(U_n+1), V) packed logits, ys (sum U_n,)
            return warp_rnnt.rnnt_loss(ops.log_softmax(xs), ys, xn, yn, compact=True)

        def pack(xs, ys, xn, yn):           # what a compact-layout joint network emits directly
            xl, yl = xn.tolist(), yn.tolist()
            V = xs.size(-1)
            rows = torch.cat([xs[n, :xl[n], :yl[n] + 1].reshape(-1, V) for n in range(xs.size(0))]).detach()
            labs = torch.cat([ys[n, :yl[n]] for n in range(ys.size(0))]).contiguous()
            return rows.contiguous(), labs
        run_loss.prepare = pack
    elif a.loss == "warp-rnnt-fused":
        def run_loss(xs, ys, xn, yn):
            return rnnt_loss_from_logits(xs, ys, xn, yn)
    elif a.loss == "warp-rnnt-lazy":
        from warp_rnnt_amd.functional import log_softmax as lazy_log_softmax

        def run_loss(xs, ys, xn, yn):
            return warp_rnnt.rnnt_loss(lazy_log_softmax(xs), ys, xn, yn, gather=True)
    else:
        raise ValueError("Unknown RNN-T loss")
    col = 0 if a.loss == "warp-rnnt" else 1
    rows = []
    for E, T, U, V in grid:
        for N in batches:
            print(f"T={T}\tU={U}\tV={V}\tN={N}\t", end="", flush=True)
            try:
                ms = run_benchmark(run_loss, E=E, N=N, T=T, U=U, V=V, random_length=a.random_length,
                                   warmup=a.warmup, device=device)
                print(f"time={ms:.2f}")
                ref = README_MS.get((T, U, V), {}).get(N, ("n/a", "n/a"))[col]     # None = out of memory there
                rows.append((T, U, V, N, ms, ref))
            except RuntimeError as e:
                print(f"error={e}")
                break
        print()
    if a.markdown:
        with open(a.markdown, "w") as f:
            f.write(f"| T | U | V | N | MI355X `{a.loss}` ms | reference RTX 2070 Super ms (README.md:38-53) |\n|---|---|---|---|---|---|\n")
            for T, U, V, N, ms, ref in rows:
                f.write(f"| {T} | {U} | {V} | {N} | {ms:.3f} | {ref if ref is not None else 'out-of-memory'} |\n")


if __name__ == "__main__":
    main()
