#!/usr/bin/env python
"""Times the log-softmax kernel alone (HIP events) at a given (rows, V). Usage: lsm_probe.py rows V [lib.so]"""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from warp_rnnt_amd import _lib
rows, V = int(sys.argv[1]), int(sys.argv[2])
path = sys.argv[3] if len(sys.argv) > 3 else _lib.lib_path()
L = ctypes.CDLL(path)
for sym, (res, a_) in _lib.SYMBOLS.items():
    fn = getattr(L, sym); fn.restype, fn.argtypes = res, a_
x = torch.randn(rows, V, device="cuda")
out = torch.empty_like(x)
s = torch.cuda.current_stream().cuda_stream
ts = []
for r in range(14):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        L.rnnt_amd_log_softmax(s, x.data_ptr(), out.data_ptr(), rows, V)
    e1.record(); torch.cuda.synchronize()
    if r >= 2: ts.append(e0.elapsed_time(e1) / 5 * 1e3)
ref = torch.log_softmax(x[:1000], -1)
err = (out[:1000] - ref).abs().max().item()
gb = rows * V * 8 / 1e9
print(f"{os.path.basename(os.path.dirname(path)) or 'lib'}: rows={rows} V={V}: median {statistics.median(ts):.1f} us  min {min(ts):.1f} us  -> {gb / (min(ts) * 1e-6) / 1e3:.2f} TB/s (max) err {err:.1e}")
