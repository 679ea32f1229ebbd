#!/usr/bin/env python
"""Per-step time series of the benchmark step (c4) from a cold process: how long the first steps take.

Every step is bracketed by its own pair of events (which costs a few microseconds per step against the
unbracketed run of bench.py), no warm-up.  Prints the first 12 steps and then means over windows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import warp_rnnt
from warp_rnnt_amd import ops

N, T, U, V = 16, 1500, 300, 50
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")
xs = torch.randn(N, T, U, V, device=dev)
ys = torch.randint(1, V, (N, U - 1), device=dev, dtype=torch.int32)
xn = torch.full((N,), T, device=dev, dtype=torch.int32)
yn = torch.full((N,), U - 1, device=dev, dtype=torch.int32)
torch.cuda.synchronize()
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 0     # this many back-to-back streaming kernels in front, no sync
if pre:
    tmp = torch.empty_like(xs)
    for _ in range(pre):
        torch.mul(xs, 1.0, out=tmp)
ea = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
eb = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
ec = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
for i in range(n):
    ea[i].record()
    lp = ops.log_softmax(xs)
    eb[i].record()
    warp_rnnt.rnnt_loss(lp, ys, xn, yn, gather=True)
    ec[i].record()
torch.cuda.synchronize()
lsm = [ea[i].elapsed_time(eb[i]) for i in range(n)]
loss = [eb[i].elapsed_time(ec[i]) for i in range(n)]
gap = [ec[i].elapsed_time(ea[i + 1]) for i in range(n - 1)]
print("step  log_softmax_ms  loss_ms")
for i in range(12):
    print(f"{i:4d}  {lsm[i]:.4f}  {loss[i]:.4f}")
for lo, hi in ((12, 25), (25, 50), (50, 100), (100, 200), (200, n)):
    if hi <= n:
        k = hi - lo
        print(f"{lo}-{hi}: log_softmax {sum(lsm[lo:hi]) / k:.4f}  loss {sum(loss[lo:hi]) / k:.4f}  "
              f"between steps {sum(gap[lo:hi - 1]) / (k - 1):.4f}")
