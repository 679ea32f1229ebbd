#!/usr/bin/env python
"""Block-by-block timeline of the probability-domain lattice kernel: when every wave of every column block entered
and left each interval (s_memrealtime, 10 ns ticks) and how long the first loader waited for the neighbour.

The stamps are compiled out of the product build (lattice_pd.hip: RNNT_PD_STATS): build the diagnostics library with
`python tools/lattice_probe.py stats:-DRNNT_PD_STATS` (no GPU needed), then run this on the GPU with RNNT_LATTICE=pd.
Outputs: profiles/r02_pd_trace_c4.txt, profiles/r03_pd_trace_c4.txt.
Usage: pd_trace.py N T U [sweep index]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from warp_rnnt_amd import _lib
N, T, U = (int(a) for a in sys.argv[1:4])
sweep = int(sys.argv[4]) if len(sys.argv) > 4 else 0
L = ctypes.CDLL(os.path.join(ROOT, "tools", "_probe", "stats", "lib.so"))
for sym, (res, a_) in _lib.SYMBOLS.items():
    if hasattr(L, sym):
        fn = getattr(L, sym); fn.restype, fn.argtypes = res, a_
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
lp2 = torch.log_softmax(torch.randn(N, T, U, 2, device=dev, generator=g), -1).contiguous()
xn = torch.full((N,), T, dtype=torch.int32, device=dev)
yn = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
costs = torch.empty((N,), device=dev); grads = torch.empty((N, T, U, 2), device=dev)
ws = torch.zeros((L.rnnt_amd_workspace_size(N, T, U),), dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream().cuda_stream
K, GPITCH = 16, 64
nA = (U + 63) // 64
mb = (T + U - 1 + K - 1) // K + 1
redo_off = L.rnnt_amd_debug_redo_offset(N, T, U)
mail_off = redo_off + ((2 * N + 2) * 4 + 255) // 256 * 256
trace_off = mail_off + 2 * N * max(nA - 1, 0) * mb * GPITCH * 8
nwords = 2 * N * nA * (mb + 8) * 16
for rep in range(3):
    ws[trace_off:trace_off + nwords * 8] = 0
    st = L.rnnt_amd_loss(s, ws.data_ptr(), 1, lp2.data_ptr(), None, xn.data_ptr(), yn.data_ptr(), costs.data_ptr(),
                         grads.data_ptr(), 1, N, T, U, 2, 0, 0.0)
    assert st == 0
    torch.cuda.synchronize()
tr = ws[trace_off:trace_off + nwords * 8].view(torch.int64).cpu().numpy().reshape(2 * N, nA, mb + 8, 16)
t0 = tr[..., 0][tr[..., 0] > 0].min()
print(f"N={N} T={T} U={U}: {nA} column blocks, {mb} blocks of {K} diagonals; times in us from the first stamp of the launch")
allend = 0
for sw in ([sweep] if sweep >= 0 else range(2 * N)):
    for cb in range(nA):
        st_ = tr[sw, cb, :, 0]
        idx = np.nonzero(st_)[0]
        blk = idx - 4
        t = (st_[idx] - t0) / 100.0
        d = np.diff(t)
        waits = tr[sw, cb, :, 1] / 100.0
        slow = [(int(blk[i]), round(float(d[i]), 2)) for i in range(len(d)) if d[i] > 2.0 * np.median(d)]
        print(f"sweep {sw} column block {cb}: blocks {blk[0]}..{blk[-1]}, first at {t[0]:7.2f}, last at {t[-1]:7.2f}; "
              f"per block median {np.median(d):.3f} mean {d.mean():.3f}; first 8: {np.round(d[:8], 2).tolist()}; "
              f"last 6: {np.round(d[-6:], 2).tolist()}")
        print(f"     blocks slower than 2x median: {slow[:24]}")
        w = [(int(i) - 4, round(float(waits[i]), 2)) for i in np.nonzero(waits)[0]]
        print(f"     loader waits (block, us): {w[:24]}  total {waits.sum():.2f}")
        allend = max(allend, t[-1])
        # the first and last intervals in detail: when each wave entered the step with that index (+4)
        rows = list(range(max(int(idx[0]) - 3, 0), int(idx[0]) + 9)) + list(range(int(idx[-1]) - 5, int(idx[-1]) + 4))
        print("     interval: start | busy time of compute(block = interval - 2), loader0, loader1, storer0, storer1 (us)")
        for r in rows:
            # compute block lb runs in interval lb + 2; helper step p in interval p
            starts = [tr[sw, cb, r - 2, 0] if r >= 2 else 0] + [tr[sw, cb, r, c] for c in (2, 3, 4, 5)]
            ends = [tr[sw, cb, r - 2, 6] if r >= 2 else 0] + [tr[sw, cb, r, c] for c in (7, 8, 9, 10)]
            st0 = min([x for x in starts if x] or [0])
            busy = " ".join(f"{(e - b) / 100.0:6.2f}" if (b and e) else "     -" for b, e in zip(starts, ends))
            print(f"     {r - 4:5d}: {(st0 - t0) / 100.0 if st0 else float('nan'):7.2f} | {busy}")
ends = (tr[..., 0].max(axis=(1, 2)) - t0) / 100.0
print("last stamp of every sweep:", np.round(ends, 1).tolist())
