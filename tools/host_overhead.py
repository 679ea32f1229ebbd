#!/usr/bin/env python
"""Host-side cost per call of the drop-in API at a tiny shape (GPU work negligible)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, warp_rnnt
from warp_rnnt_amd import ops
import warp_rnnt._C as core
N, T, U, V = 16, 150, 40, 28
x = torch.randn(N, T, U, V, device="cuda")
ys = torch.randint(1, V, (N, U - 1), dtype=torch.int32, device="cuda")
xn = torch.full((N,), T, dtype=torch.int32, device="cuda"); yn = torch.full((N,), U - 1, dtype=torch.int32, device="cuda")
lp = ops.log_softmax(x)
def bench(name, fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:40s} host {1e6*(t1-t0)/n:7.1f} us/call   wall {1e6*(t2-t0)/n:7.1f} us/call")
bench("ops.log_softmax", lambda: ops.log_softmax(x))
bench("_C.rnnt_loss dense", lambda: core.rnnt_loss(lp, ys, xn, yn))
bench("rnnt_loss(gather=False)", lambda: warp_rnnt.rnnt_loss(lp, ys, xn, yn))
bench("rnnt_loss(gather=True)", lambda: warp_rnnt.rnnt_loss(lp, ys, xn, yn, gather=True))
bench("torch.empty_like x3", lambda: (torch.empty_like(lp), torch.empty(16, device='cuda'), torch.empty(100, device='cuda')))
bench("check_inputs", lambda: core.check_inputs(lp, ys, xn, yn))
# (the compact entry, with launch bounds and with its read-back: tools/compact_host_probe.py -- measured there on its own,
#  in both orders; as rows of this script they inherit the allocator state the rows above leave behind)
