#!/usr/bin/env python
"""Eager launches vs one captured HIP graph per step (torch.cuda.CUDAGraph), ms per step of the bench workload
(native log_softmax + rnnt_loss, gradients included). Usage: graph_probe.py N T U V gather(0|1) [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from warp_rnnt import _C
from warp_rnnt_amd import ops

N, T, U, V, gather = (int(a) for a in sys.argv[1:6])
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 300
dev = torch.device("cuda:0")
torch.manual_seed(0)
logits = torch.randn(N, T, U, V, device=dev)
lp = torch.empty_like(logits)
labels = torch.randint(1, V, (N, U - 1), device=dev, dtype=torch.int32)
xn = torch.full((N,), T, device=dev, dtype=torch.int32)
yn = torch.full((N,), U - 1, device=dev, dtype=torch.int32)


def step():
    ops.log_softmax(logits, lp)
    if gather:
        return _C.rnnt_loss_gather(lp, labels, xn, yn, 0, 0.0)
    return _C.rnnt_loss(lp, labels, xn, yn, 0, 0.0)


def timed(fn):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


eager = timed(step)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
ref = step()
g.replay(); torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(out, ref))
graph = timed(g.replay)
print(f"N={N} T={T} U={U} V={V} gather={gather}: eager {eager:.4f} ms/step, graph replay {graph:.4f} ms/step "
      f"(results equal: {same})")
