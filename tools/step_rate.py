#!/usr/bin/env python
"""ms per forward step `rnnt_loss(ops.log_softmax(x), ..., gather=True)` (the reference's protocol, benchmark.py:62-70) by
vocabulary, once per setting of the environment variables given as NAME=VALUE arguments (read once per process: the tool
runs itself per setting).  STEP_RATE_SHAPE=N,T,U (default 16,1500,300), STEP_RATE_V="28 40 50" (default).

    python tools/step_rate.py RNNT_LSM_NO_REGS=1
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


def child():
    import torch
    import warp_rnnt
    from warp_rnnt_amd import ops
    dev = torch.device("cuda:0")
    N, T, U = (int(v) for v in os.environ.get("STEP_RATE_SHAPE", "16,1500,300").split(","))
    for V in (int(v) for v in os.environ.get("STEP_RATE_V", "28 40 50").split()):
        g = torch.Generator(device=dev).manual_seed(V)
        x = torch.randn((N, T, U, V), device=dev, generator=g)
        ys = torch.randint(1, V, (N, U - 1), dtype=torch.int32, device=dev, generator=g)
        xn = torch.full((N,), T, dtype=torch.int32, device=dev)
        yn = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
        fn = lambda: warp_rnnt.rnnt_loss(ops.log_softmax(x), ys, xn, yn, gather=True)
        for _ in range(30):
            fn()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                c = fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        print(f"N={N} T={T} U={U} V={V:5d}  {statistics.median(ts):8.4f} ms per step   sum(costs) {float(c.double().sum()):.4f}",
              flush=True)
        del x


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        for env in [{}] + [dict([a.split("=", 1)]) for a in sys.argv[1:]]:
            print("== " + (" ".join(f"{k}={v}" for k, v in env.items()) or "shipped"), flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, **env), check=True)
