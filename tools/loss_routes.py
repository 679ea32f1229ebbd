#!/usr/bin/env python
"""The whole loss entry (dense log-probs in, gathered gradients out: gather + sweeps + gradients, the ring preparation folded
into the gather's launch as in every real call) timed with each lattice kernel pinned -- what tools/lattice_routes.py's
sweeps-alone numbers (which pay the preparation as a launch of its own on the wd route) leave open at the margins.

    python tools/loss_routes.py [--fused] N,T,U[,V] ...     us per call: median over interleaved rounds; auto = launch_lattice's choice

--fused: logits in (log-softmax fused into the gather: the lazy log_softmax's route), whose producer does not carry the ring
preparation -- k_lattice_wd pays it as a launch of its own there, and launch_lattice's thresholds are the later ones.
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from warp_rnnt_amd import debug, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    args = sys.argv[1:]
    fused = "--fused" in args
    if fused:
        args.remove("--fused")
    kind = ops.IN_LOGITS_DENSE if fused else ops.IN_LOG_PROBS_DENSE
    shapes = [tuple(int(v) for v in a.split(",")) for a in args]
    print(f"us per loss call ({'logits' if fused else 'dense log-probs'} in, gathered gradients out): median (min)")
    for sh in shapes:
        N, T, U = sh[:3]
        V = sh[3] if len(sh) > 3 else 64
        g = torch.Generator(device=dev)
        g.manual_seed(0)
        lp = torch.log_softmax(torch.randn(N, T, U, V, device=dev, generator=g), -1)
        ys = torch.randint(1, V, (N, U - 1), device=dev, generator=g, dtype=torch.int32)
        xn = torch.full((N,), T, dtype=torch.int32, device=dev)
        yn = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
        times = {k: [] for k in ("ws", "wd", "wl", "auto")}
        ran, ref = {}, None
        for rnd in range(8):
            for k in times:
                with debug.lattice_kernel(k):
                    c, _ = ops.loss(lp, ys, xn, yn, kind, ops.GRADS_GATHERED)
                    ran[k] = debug.last_lattice_kernel()
                    if ref is None:
                        ref = c.clone()
                    assert torch.equal(c, ref), k
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        ops.loss(lp, ys, xn, yn, kind, ops.GRADS_GATHERED)
                    e1.record()
                    torch.cuda.synchronize()
                    if rnd >= 2:
                        times[k].append(e0.elapsed_time(e1) * 100.0)
        line = f"N={N:4d} T={T:5d} U={U:4d} V={V:4d}  "
        for k in times:
            line += f"{k} {statistics.median(times[k]):7.1f} ({min(times[k]):7.1f}) [{ran[k].replace('lattice_', '')}]  "
        print(line, flush=True)


if __name__ == "__main__":
    main()
