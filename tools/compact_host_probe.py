#!/usr/bin/env python
"""Host time per call of the compact entry at c2's size, compiled binding and ctypes path, bounds and read-back, in two
orders (is a row's cost its own or the previous row's backlog?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, warp_rnnt
import warp_rnnt._C as core
from warp_rnnt_amd import ops
N, T, U, V = 16, 150, 40, 28
x = torch.randn(N, T, U, V, device="cuda")
ys = torch.randint(1, V, (N, U - 1), dtype=torch.int32, device="cuda")
xn = torch.full((N,), T, dtype=torch.int32, device="cuda"); yn = torch.full((N,), U - 1, dtype=torch.int32, device="cuda")
lp = ops.log_softmax(x)
xs = lp.reshape(-1, V).contiguous(); ysc = ys.reshape(-1).contiguous()


def bench(name, fn, n=1000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:60s} host {1e6*(t1-t0)/n:7.1f} us/call   wall {1e6*(t2-t0)/n:7.1f} us/call", flush=True)


rows = [("_C.rnnt_loss_compact bounds", lambda: core.rnnt_loss_compact(xs, ysc, xn, yn, 0, 0.0, True, T, U - 1)),
        ("_C.rnnt_loss_compact read-back", lambda: core.rnnt_loss_compact(xs, ysc, xn, yn, 0, 0.0, True)),
        ("ops.loss_compact bounds (ctypes)", lambda: ops.loss_compact(xs, ysc, xn, yn, 0, 0.0, True, T, U - 1)),
        ("rnnt_loss(compact=True, bounds)", lambda: warp_rnnt.rnnt_loss(xs, ysc, xn, yn, compact=True, max_frames=T, max_labels=U - 1))]
print("native binding:", core._native is not None)
for r in rows: bench(*r)
for r in reversed(rows): bench(*r)
