#!/usr/bin/env python
"""Times rnnt_amd_loss(dense log-probs -> diagonal gradient pairs) = gather + lattice + grads for one build
of the library: gather_probe.py path/to/lib.so [N T U V].  Differences between builds that only touch the
gather kernel show up one to one."""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from warp_rnnt_amd import _lib
path = sys.argv[1]
N, T, U, V = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (16, 1500, 300, 50)
KIND = int(os.environ.get("PROBE_INPUT_KIND", "0"))      # 0 dense log-probs, 2 logits (fused log-softmax + gather)
L = ctypes.CDLL(path)
for sym, (res, a_) in _lib.SYMBOLS.items():
    if hasattr(L, sym):
        fn = getattr(L, sym); fn.restype, fn.argtypes = res, a_
dev = torch.device("cuda:0")
lp = torch.log_softmax(torch.randn(N, T, U, V, device=dev), -1)
ys = torch.randint(1, V, (N, U - 1), dtype=torch.int32, device=dev)
xn = torch.full((N,), T, dtype=torch.int32, device=dev); yn = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
costs = torch.empty((N,), device=dev); grads = torch.empty((N, T, U, 2), device=dev)
ws = torch.empty((L.rnnt_amd_workspace_size(N, T, U),), dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream().cuda_stream
ts = []
for r in range(14):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        st = L.rnnt_amd_loss(s, ws.data_ptr(), KIND, lp.data_ptr(), ys.data_ptr(), xn.data_ptr(), yn.data_ptr(),
                             costs.data_ptr(), grads.data_ptr(), 1, N, T, U, V, 0, 0.0)
    e1.record(); torch.cuda.synchronize()
    assert st == 0
    if r >= 2: ts.append(e0.elapsed_time(e1) / 5 * 1e3)
print(f"{os.path.basename(os.path.dirname(path))}: median {statistics.median(ts):.1f} us  min {min(ts):.1f} us  sum(costs) {costs.double().sum().item():.4f}")
