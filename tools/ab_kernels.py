#!/usr/bin/env python
"""A/B of two builds of the library on the kernels around the lattice, op by op (ms per call: HIP events around 10 calls,
median of 5, after 300 ms of load), one process per build and round, interleaved:

    python tools/ab_kernels.py name=/path/lib.so name=/path/lib.so [--rounds 3]

Every op is timed on BASELINE's c4 tensor (N=16, T=1500, U=300, V=50) unless it says otherwise."""
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from warp_rnnt_amd import ops
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from cabi_probe import ragged_compact_batch
    dev = torch.device("cuda:0")

    def timed(fn):
        for _ in range(10):
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

    spin = torch.empty(64 << 20, device=dev)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(600):
        spin.mul_(1.0)
    t1.record(); torch.cuda.synchronize()
    res = {}
    N, T, U, V = 16, 1500, 300, 50
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn((N, T, U, V), device=dev, generator=g)
    lp = torch.log_softmax(x, -1)
    ys = torch.randint(1, V, (N, U - 1), dtype=torch.int32, device=dev, generator=g)
    xn = torch.full((N,), T, dtype=torch.int32, device=dev)
    yn = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
    res["loss, dense log-probs in, pairs out"] = timed(lambda: ops.loss(lp, ys, xn, yn, ops.IN_LOG_PROBS_DENSE, ops.GRADS_GATHERED_DIAGONAL))
    res["loss, dense log-probs in, dense out"] = timed(lambda: ops.loss(lp, ys, xn, yn, ops.IN_LOG_PROBS_DENSE, ops.GRADS_DENSE))
    res["fused forward (logits in)"] = timed(lambda: ops.loss(x, ys, xn, yn, ops.IN_LOGITS_DENSE, ops.GRADS_GATHERED_DIAGONAL))
    costs, grads = ops.loss(x, ys, xn, yn, ops.IN_LOGITS_DENSE, ops.GRADS_GATHERED_DIAGONAL)
    go = torch.ones((N,), device=dev)
    res["fused backward (d/d logits)"] = timed(lambda: ops.logits_backward(x, ys, grads, go, 0))
    out = torch.empty_like(x)
    res["log_softmax V=50"] = timed(lambda: ops.log_softmax(x, out=out))
    res["log_softmax_backward V=50"] = timed(lambda: ops.log_softmax_backward(x, lp, grad_in=out))
    del out, lp, grads
    xs, cys, cxn, cyn = ragged_compact_batch(N, T, U, V, dev)
    res["loss_compact (ragged c4)"] = timed(lambda: ops.loss_compact(xs, cys, cxn, cyn))
    res["loss_compact bounded (ragged c4)"] = timed(lambda: ops.loss_compact(xs, cys, cxn, cyn, max_frames=T, max_labels=U - 1))
    del xs, x
    for Vb, rows in ((5000, 96000), (10000, 48000), (1000, 480000), (256, 1875000), (2048, 234000), (3000, 160000), (4096, 117000), (8192, 58600)):
        xb = torch.randn((rows, Vb), device=dev, generator=g)
        ob = torch.empty_like(xb)
        res[f"log_softmax V={Vb}"] = timed(lambda: ops.log_softmax(xb, out=ob))
        yb = ops.log_softmax(xb)
        res[f"log_softmax_backward V={Vb}"] = timed(lambda: ops.log_softmax_backward(xb, yb, grad_in=ob))
        del xb, ob, yb
    # c3: N=1, T=150, U=40... the fused forward at a large vocabulary
    N3, T3, U3, V3 = 16, 150, 40, 5000
    x3 = torch.randn((N3, T3, U3, V3), device=dev, generator=g)
    ys3 = torch.randint(1, V3, (N3, U3 - 1), dtype=torch.int32, device=dev, generator=g)
    xn3 = torch.full((N3,), T3, dtype=torch.int32, device=dev)
    yn3 = torch.full((N3,), U3 - 1, dtype=torch.int32, device=dev)
    res["fused forward V=5000 (c3)"] = timed(lambda: ops.loss(x3, ys3, xn3, yn3, ops.IN_LOGITS_DENSE, ops.GRADS_GATHERED_DIAGONAL))
    c3, g3 = ops.loss(x3, ys3, xn3, yn3, ops.IN_LOGITS_DENSE, ops.GRADS_GATHERED_DIAGONAL)
    go3 = torch.ones((N3,), device=dev)
    res["fused backward V=5000 (c3)"] = timed(lambda: ops.logits_backward(x3, ys3, g3, go3, 0))
    for k, v in res.items():
        print(f"{k}\t{v * 1e3:.1f}", flush=True)


def main():
    args = sys.argv[1:]
    rounds = 3
    if "--rounds" in args:
        i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
    libs = [a.split("=", 1) for a in args]
    table = {}
    for r in range(rounds):
        for name, path in libs:
            env = dict(os.environ, WARP_RNNT_AMD_LIB=os.path.abspath(path), WARP_RNNT_AMD_NO_NATIVE_BINDING="1")
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
            if out.returncode != 0:
                print(out.stderr[-2000:]); sys.exit(1)
            for line in out.stdout.splitlines():
                if "\t" in line:
                    k, v = line.split("\t")
                    table.setdefault(k, {}).setdefault(name, []).append(float(v))
            if os.environ.get("AB_CABI"):        # the reference-named C entry points, through tools/cabi_probe.py
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cabi_probe.py"), "c4"], env=env,
                                     capture_output=True, text=True)
                sect = ""
                for line in out.stdout.splitlines():
                    if line.startswith("c4"):
                        sect = "ragged " if "ragged" in line else ""
                    elif line.strip().endswith(" ms"):
                        k, v = line.strip()[:-3].rsplit(None, 1)
                        table.setdefault("cabi: " + sect + k.strip(), {}).setdefault(name, []).append(float(v) * 1e3)
    names = [n for n, _ in libs]
    print(f"{'op (us per call; every round, then the median)':58s}" + "".join(f"{n:>34s}" for n in names))
    for k, row in table.items():
        print(f"{k:58s}" + "".join(f"{' '.join('%.0f' % v for v in row[n]):>26s} {statistics.median(row[n]):7.1f}" for n in names))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        main()
