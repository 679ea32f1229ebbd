#!/usr/bin/env python
"""Times lsm / gather+loss pieces of the c4 step for a given lib (.so path optional)."""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from warp_rnnt_amd import _lib
path = sys.argv[1] if len(sys.argv) > 1 else _lib.lib_path()
L = ctypes.CDLL(path)
for sym, (res, a_) in _lib.SYMBOLS.items():
    if hasattr(L, sym):
        fn = getattr(L, sym); fn.restype, fn.argtypes = res, a_
N, T, U, V = 16, 1500, 300, 50
dev = "cuda"
x = torch.randn(N, T, U, V, device=dev); lp = torch.empty_like(x)
ys = torch.randint(1, V, (N, U - 1), dtype=torch.int32, device=dev)
xn = torch.full((N,), T, dtype=torch.int32, device=dev); yn = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
costs = torch.empty(N, device=dev); g2 = torch.empty(N, T, U, 2, device=dev)
ws = torch.empty(L.rnnt_amd_workspace_size(N, T, U), dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream().cuda_stream
def lsm(): L.rnnt_amd_log_softmax(s, x.data_ptr(), lp.data_ptr(), N * T * U, V)
def loss(): L.rnnt_amd_loss(s, ws.data_ptr(), 0, lp.data_ptr(), ys.data_ptr(), xn.data_ptr(), yn.data_ptr(), costs.data_ptr(), g2.data_ptr(), 1, N, T, U, V, 0, 0.0)
def both(): lsm(); loss()
for name, fn in (("lsm", lsm), ("loss(gather)", loss), ("lsm+loss", both)):
    ts = []
    for r in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(e0.elapsed_time(e1) / 5 * 1e3)
    print(f"{os.path.basename(os.path.dirname(path))}: {name:14s} median {statistics.median(ts):7.1f} us  min {min(ts):7.1f}")
