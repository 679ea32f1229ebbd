#!/usr/bin/env python
"""Rate of the fused entry points (warp_rnnt_amd.fused: logits -> loss, loss -> d/d logits) against V, with the
rows-in-registers kernel (k_lsm_rows) against the LDS-staged one (RNNT_LSM_NO_ROWS=1, read once per process: the tool
runs itself once per setting).

    python tools/fused_rate.py [V ...]        (default: 64 96 128 160 192 256 50)

Per V: N=32, T=500, U=100 (1.6 M cells; FUSED_RATE_SHAPE=N,T,U for another lattice); ms per call of the forward (logits -> costs + gradient pairs) and of the
backward (d/d logits), HIP events around 10 back-to-back calls, median of 5, after 30 ms of load; TB/s on the
algorithmic bytes (forward 4V+8, backward 8V+8 per cell)."""
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _ab import use_ab_build  # noqa: E402
use_ab_build()      # (the build that reads the A/B knobs from the environment: tools/_ab.py)


def timed(torch, fn):
    for _ in range(20):
        fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    return statistics.median(ts)


def child(vs):
    import torch
    from warp_rnnt_amd import ops
    dev = torch.device("cuda:0")
    N, T, U = (int(v) for v in os.environ.get("FUSED_RATE_SHAPE", "32,500,100").split(","))
    for V in vs:
        g = torch.Generator(device=dev).manual_seed(V)
        x = torch.randn((N, T, U, V), device=dev, generator=g)
        ys = torch.randint(1, V, (N, U - 1), dtype=torch.int32, device=dev, generator=g)
        xn = torch.full((N,), T, dtype=torch.int32, device=dev)
        yn = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
        costs, grads = ops.loss(x, ys, xn, yn, ops.IN_LOGITS_DENSE, ops.GRADS_GATHERED_DIAGONAL, 0, 0.0)
        go = torch.ones((N,), device=dev)
        fwd = timed(torch, lambda: ops.loss(x, ys, xn, yn, ops.IN_LOGITS_DENSE, ops.GRADS_GATHERED_DIAGONAL, 0, 0.0))
        bwd = timed(torch, lambda: ops.logits_backward(x, ys, grads, go, 0))
        cells = N * T * U
        print(f"V={V:5d}  forward {fwd * 1e3:7.1f} us  backward {bwd * 1e3:7.1f} us ({(8 * V + 8) * cells / bwd / 1e9:5.2f} TB/s)"
              f"  sum(costs) {float(costs.double().sum()):.4f}", flush=True)
        del x


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child([int(v) for v in sys.argv[2:]])
    else:
        vs = sys.argv[1:] or ["64", "96", "128", "160", "192", "256", "50"]
        runs = [("rows in registers (shipped)", {}), ("RNNT_LSM_NO_DIAG=1 (consecutive rows per wave)", {"RNNT_LSM_NO_DIAG": "1"}),
                ("RNNT_LSM_NO_ROWS=1 (LDS-staged kernel)", {"RNNT_LSM_NO_ROWS": "1"})]
        if os.environ.get("FUSED_RATE_ROWS_ANY"):
            runs.insert(1, ("RNNT_LSM_ROWS_ANY=1 (probe: rows in registers for every even V)", {"RNNT_LSM_ROWS_ANY": "1"}))
        # FUSED_RATE_LIBS="name=/path/lib.so ...": other builds of the library (A/B of compile-time knobs)
        for item in os.environ.get("FUSED_RATE_LIBS", "").split():
            name, path = item.split("=", 1)
            runs.append((name, {"WARP_RNNT_AMD_LIB": os.path.abspath(path), "WARP_RNNT_AMD_NO_NATIVE_BINDING": "1"}))
        for label, env in runs:
            print(f"== {label}", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + vs, env=dict(os.environ, **env), check=True)
