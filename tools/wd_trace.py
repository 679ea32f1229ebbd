#!/usr/bin/env python
"""Interval-by-interval timeline of the distributed log-domain lattice kernel (csrc/lattice_wd.hip): when the compute
wave and the I/O wave of every column block entered and left each interval (s_memrealtime, 10 ns ticks) and how long
the I/O wave waited for the left neighbour's boundary column.

The stamps are compiled out of the product build (RNNT_WD_STATS): build the diagnostics library without a GPU with
`python tools/lattice_probe.py wdstats:-DRNNT_WD_STATS`, then run this on the GPU.
Usage: wd_trace.py N T U [sweep index | -1 for all]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from warp_rnnt_amd import _lib  # noqa: E402

N, T, U = (int(a) for a in sys.argv[1:4])
sweep = int(sys.argv[4]) if len(sys.argv) > 4 else 0
L = ctypes.CDLL(os.path.join(ROOT, "tools", "_probe", "wdstats", "lib.so"))
for sym, (res, a_) in _lib.SYMBOLS.items():
    if hasattr(L, sym):
        fn = getattr(L, sym)
        fn.restype, fn.argtypes = res, a_
L.rnnt_amd_debug_set_lattice_kernel(2)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
lp2 = torch.log_softmax(torch.randn(N, T, U, 2, device=dev, generator=g), -1).contiguous()
xn = torch.full((N,), T, dtype=torch.int32, device=dev)
yn = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
costs = torch.empty((N,), device=dev)
grads = torch.empty((N, T, U, 2), device=dev)
ws = torch.zeros((L.rnnt_amd_workspace_size(N, T, U),), dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream().cuda_stream
K = 16 if T >= int(os.environ.get("RNNT_WD_K16_FROM_T", "1024")) else 8     # (csrc/lattice_wd.hip: wd_block_diagonals)
nA = (U + 63) // 64
pitch = ((T + U - 1 + K - 1) // K + 2) * K
slots = (T + U - 1) // K + 24
redo_off = L.rnnt_amd_debug_redo_offset(N, T, U)
mail_off = redo_off + ((2 * N + 2) * 4 + 255) // 256 * 256
trace_off = mail_off + 2 * N * max(nA - 1, 0) * pitch * 8
nwords = 2 * N * nA * slots * 8
assert trace_off + nwords * 8 <= ws.numel()
for rep in range(3):
    ws[trace_off:trace_off + nwords * 8] = 0
    st = L.rnnt_amd_loss(s, ws.data_ptr(), 1, lp2.data_ptr(), None, xn.data_ptr(), yn.data_ptr(), costs.data_ptr(),
                         grads.data_ptr(), 1, N, T, U, 2, 0, 0.0)
    assert st == 0
    torch.cuda.synchronize()
tr = ws[trace_off:trace_off + nwords * 8].view(torch.int64).cpu().numpy().reshape(2 * N, nA, slots, 8)
io0 = tr[..., 2]
t0 = io0[io0 > 0].min()
print(f"N={N} T={T} U={U}: {nA} column blocks, blocks of {K} diagonals; times in us from the first stamp of the launch; "
      f"sum(costs) {float(costs.double().sum()):.4f}")
us = lambda x: (x - t0) / 100.0
for sw in ([sweep] if sweep >= 0 else range(2 * N)):
    for cb in range(nA):
        c0, c1, i0, i1, wt, s0, s1 = (tr[sw, cb, :, k] for k in range(7))
        idx = np.nonzero(c0)[0]
        idx = idx[idx >= 10]          # (slots 8, 9: the storer's dry run of the compute loop)
        if len(idx) == 0:
            continue
        t = us(c0[idx])
        d = np.diff(t)
        busy = (c1[idx] - c0[idx]) / 100.0
        iob = np.where((i0 > 0) & (i1 > 0), (i1 - i0) / 100.0, 0.0)
        stb = np.where((s0 > 0) & (s1 > 0), (s1 - s0) / 100.0, 0.0)
        dry = [round(float(c1[q] - c0[q]) / 100.0, 2) for q in (8, 9) if c0[q]]
        waits = [(int(i) - 8 - 1, round(float(wt[i]) / 100.0, 2)) for i in np.nonzero(wt)[0]]
        print(f"sweep {sw} cb {cb}: blocks {idx[0] - 10}..{idx[-1] - 10}; first compute at {t[0]:7.2f}, last ends "
              f"{us(c1[idx[-1]]):7.2f}; interval median {np.median(d):.3f} mean {d.mean():.3f}; compute busy median "
              f"{np.median(busy):.3f}; loader busy median {np.median(iob[iob > 0]):.3f}; storer busy median "
              f"{np.median(stb[stb > 0]):.3f}; dry-run blocks {dry} from {us(c0[8]) if c0[8] else us(c0[9]):.2f}")
        print(f"     first 14 intervals: {np.round(d[:14], 2).tolist()}")
        print(f"     first 14 compute busy: {np.round(busy[:14], 2).tolist()}")
        print(f"     first 14 io busy: {np.round(iob[idx[0]:idx[0] + 14], 2).tolist()}")
        print(f"     last 12 intervals: {np.round(d[-12:], 2).tolist()}")
        print(f"     last 12 io busy: {np.round(iob[idx[-1] - 11:idx[-1] + 1], 2).tolist()}")
        slow = [(int(idx[i]) - 10, round(float(d[i]), 2)) for i in range(len(d)) if d[i] > 1.8 * np.median(d)]
        print(f"     intervals slower than 1.8x median (block, us): {slow[:40]}")
        print(f"     I/O wave waits for the neighbour (block, us): {waits[:40]}  total {sum(w for _, w in waits):.2f}")
ends = (tr[..., 1].max(axis=(1, 2)) - t0) / 100.0
print("last compute stamp of every sweep:", np.round(ends, 1).tolist())
