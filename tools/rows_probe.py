#!/usr/bin/env python
"""Times the row-wide kernels (log-softmax, its backward, the dense expansion) at (N,T,U,V) with HIP events and
checks them against torch. Usage: rows_probe.py N T U V    (RNNT_LSM_NO_SHIFT=1 selects the unshifted lane map)"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _ab import use_ab_build  # noqa: E402
use_ab_build()      # (the build that reads the A/B knobs from the environment: tools/_ab.py)
import torch
from warp_rnnt_amd import ops

N, T, U, V = (int(a) for a in sys.argv[1:5])
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(N, T, U, V, device=dev)
y = torch.empty_like(x)
dx = torch.empty_like(x)
labels = torch.randint(1, V, (N, U - 1), device=dev, dtype=torch.int32)
xn = torch.full((N,), T, device=dev, dtype=torch.int32)
yn = torch.full((N,), U - 1, device=dev, dtype=torch.int32)
g2 = torch.randn(N, T, U, 2, device=dev)
gc = torch.ones(N, device=dev)


def timed(name, fn, nbytes):
    ts = []
    for r in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            fn()
        e1.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(e0.elapsed_time(e1) / 4 * 1e3)
    print(f"{name:22s} median {statistics.median(ts):8.1f} us  min {min(ts):8.1f} us  "
          f"{nbytes / (statistics.median(ts) * 1e-6) / 1e12:.2f} TB/s (median)", flush=True)


tag = "unshifted" if os.environ.get("RNNT_LSM_NO_SHIFT") else "line-aligned"
print(f"N={N} T={T} U={U} V={V} ({x.numel() * 4 / 1e9:.2f} GB per tensor), {tag}")
timed("log_softmax", lambda: ops.log_softmax(x, y), x.numel() * 8)
ref = torch.log_softmax(x[0, :8], -1)
print("   err", (y[0, :8] - ref).abs().max().item())
timed("log_softmax in place", lambda: ops.log_softmax(dx, dx), x.numel() * 8)
timed("log_softmax_backward", lambda: ops.log_softmax_backward(x, y, dx), x.numel() * 12)
refb = x[0, :8] - torch.exp(y[0, :8]) * x[0, :8].sum(-1, keepdim=True)
print("   err", (dx[0, :8] - refb).abs().max().item())
out = [None]
def ex(): out[0] = ops.expand_grads(g2, labels, xn, yn, gc, V, 0)
del dx
timed("expand_grads", ex, x.numel() * 4)
d = out[0]
print("   nonzero per row (first rows)", (d[0, 0] != 0).sum(-1)[:4].tolist(), "sum check",
      float(d[0].sum()), )
