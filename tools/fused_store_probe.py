#!/usr/bin/env python
"""What do the scattered 8-byte pair stores of the fused logits -> pairs kernel cost?  Times the fused entry (forward:
log-softmax + gather fused, sweeps, gradients) at c4 on the shipped library and on two probe builds whose results are
WRONG by construction -- the pairs stored into a 64 KB region that stays in L2 (no DRAM writes), and in row-major order
(coalesced) -- in interleaved child processes.  HIP events around 20 calls, median of 7, per build.

    python tools/fused_store_probe.py [N T U V]"""
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from warp_rnnt_amd import ops
    N, T, U, V = (int(a) for a in sys.argv[2:6])
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    xs = torch.randn((N, T, U, V), device=dev, generator=g)
    ys = torch.randint(1, V, (N, U - 1), dtype=torch.int32, device=dev, generator=g)
    xn = torch.full((N,), T, dtype=torch.int32, device=dev)
    yn = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
    for _ in range(3):
        ops.loss(xs, ys, xn, yn, ops.IN_LOGITS_DENSE, ops.GRADS_GATHERED_DIAGONAL, 0, 0.0)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.loss(xs, ys, xn, yn, ops.IN_LOGITS_DENSE, ops.GRADS_GATHERED_DIAGONAL, 0, 0.0)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{statistics.median(ts):8.1f} us (min {min(ts):.1f})", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
        sys.exit(0)
    shape = sys.argv[1:5] if len(sys.argv) >= 5 else ["16", "1500", "300", "50"]
    from warp_rnnt_amd import _build
    libs = [("shipped", _build.LIB), ("pairs into a hot 64 KB buffer (no DRAM writes; wrong results)", _build.variant_path("probe_hot_pairs")),
            ("pairs in row-major order (coalesced; wrong layout)", _build.variant_path("probe_linear_pairs"))]
    print(f"fused entry, forward, N,T,U,V = {','.join(shape)}")
    for rnd in range(3):
        for name, lib in libs:
            if not os.path.exists(lib):
                continue
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + shape,
                                 env=dict(os.environ, WARP_RNNT_AMD_LIB=lib), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            print(f"  {name:70s} {out.stdout.decode().strip()}", flush=True)
