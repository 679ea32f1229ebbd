#!/usr/bin/env python
"""us per call of the dense entry (_C.rnnt_loss(log_probs): costs + dense (N,T,U,V) gradients) over small lattices, once
per value of RNNT_DENSE_ONE_LAUNCH_CELLS given as arguments (the size up to which the gradient kernel and the expansion
run as one launch; read once per process: the tool runs itself per value).

    python tools/dense_rate.py 0 262144 1048576
"""
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _ab import use_ab_build  # noqa: E402
use_ab_build()      # (the build that reads the A/B knobs from the environment: tools/_ab.py)
SHAPES = [(16, 150, 40, 28), (32, 150, 40, 28), (64, 150, 40, 28), (128, 150, 40, 28), (256, 150, 40, 28),
          (4, 500, 100, 50), (8, 500, 100, 50), (16, 500, 100, 50), (32, 500, 100, 50), (2, 150, 20, 5000), (8, 150, 20, 5000)]


def child():
    import torch
    import warp_rnnt._C as core
    dev = torch.device("cuda:0")
    for (N, T, U, V) in SHAPES:
        g = torch.Generator(device=dev).manual_seed(N + T + U + V)
        lp = torch.log_softmax(torch.randn((N, T, U, V), device=dev, generator=g), -1)
        ys = torch.randint(1, V, (N, U - 1), dtype=torch.int32, device=dev, generator=g)
        xn = torch.full((N,), T, dtype=torch.int32, device=dev)
        yn = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
        fn = lambda: core.rnnt_loss(lp, ys, xn, yn)
        for _ in range(20):
            fn()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                c, gr = fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 50)
        print(f"N={N:4d} T={T:4d} U={U:4d} V={V:5d}  cells {N * T * U:8d}  {statistics.median(ts) * 1e3:7.1f} us per call"
              f"   sum(costs) {float(c.double().sum()):.4f}  sum(grads) {float(gr.double().sum()):.4f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        for v in sys.argv[1:] or ["0", "262144"]:
            print(f"== RNNT_DENSE_ONE_LAUNCH_CELLS={v}", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child"],
                           env=dict(os.environ, RNNT_DENSE_ONE_LAUNCH_CELLS=v), check=True)
