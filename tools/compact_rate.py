#!/usr/bin/env python
"""us per call of rnnt_loss(compact=True, max_frames=, max_labels=) on ragged batches (lengths 50-100 % of T and U), HIP
events around 20 back-to-back calls, median of 5 -- once per setting of the environment variables given as arguments:

    python tools/compact_rate.py [NAME=VALUE ...]     e.g.  RNNT_COMPACT_PLAIN_TILE_ORDER=1
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
SHAPES = [(32, 250, 100, 128), (32, 500, 50, 128), (32, 500, 100, 128), (32, 500, 200, 128), (32, 1000, 100, 128),
          (64, 500, 100, 128), (64, 1000, 200, 128), (32, 500, 100, 1024), (16, 1500, 300, 50)]


def child():
    import torch
    import warp_rnnt
    from warp_rnnt_amd import ops
    from tools.shape_map import lengths
    dev = torch.device("cuda:0")
    for (N, T, U, V) in SHAPES:
        g = torch.Generator(device=dev).manual_seed(N + T + U + V)
        xs = torch.randn((N, T, U, V), device=dev, generator=g)
        ys = torch.randint(1, V, (N, U - 1), dtype=torch.int32, device=dev, generator=g)
        xn, yn = (t.to(dev) for t in lengths(torch, N, T, U, True, T + U))
        lp = ops.log_softmax(xs)
        rows = torch.cat([lp[n, :int(xn[n]), :int(yn[n]) + 1].reshape(-1, V) for n in range(N)]).contiguous()
        labs = torch.cat([ys[n, :int(yn[n])] for n in range(N)]).contiguous()
        del lp, xs
        fn = lambda: warp_rnnt.rnnt_loss(rows, labs, xn, yn, compact=True, max_frames=T, max_labels=U - 1)
        for _ in range(10):
            fn()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                c = fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        print(f"N={N:3d} T={T:5d} U={U:4d} V={V:5d}  live cells {rows.shape[0] / (N * T * U):.2f} of the padded plane  "
              f"{statistics.median(ts) * 1e3:8.1f} us per call   sum(costs) {float(c.double().sum()):.4f}", flush=True)
        del rows
        torch.cuda.empty_cache()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        # NAME=VALUE: environment variables; lib:<name>=<path>: another build of the library
        settings = [{}]
        for a in sys.argv[1:]:
            if a.startswith("lib:"):
                settings.append({"WARP_RNNT_AMD_LIB": os.path.abspath(a[4:].split("=", 1)[1]), "WARP_RNNT_AMD_NO_NATIVE_BINDING": "1"})
            else:
                settings.append(dict([a.split("=", 1)]))
        for env in settings:
            print("== " + (" ".join(f"{k}={v}" for k, v in env.items()) or "shipped"), flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, **env), check=True)
