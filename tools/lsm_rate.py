#!/usr/bin/env python
"""Rate of the plain log-softmax kernel (ops.log_softmax, out of place) against tensor size and V: GB in + out per
launch over the median launch time (HIP events around 10 back-to-back launches, 5 rounds), after 30 ms of load."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _ab import use_ab_build  # noqa: E402
use_ab_build()      # (the build that reads the A/B knobs from the environment: tools/_ab.py)
import torch
from warp_rnnt_amd import ops

dev = torch.device("cuda:0")
cases = [(50, g) for g in (0.36, 0.72, 1.44, 2.88, 5.76)] + [(5000, g) for g in (0.48, 0.96, 1.92, 3.84)] + \
        [(10000, g) for g in (1.92, 3.84)] + [(1000, 1.92), (2048, 1.92), (4096, 1.92), (8192, 1.92)]
if len(sys.argv) > 1:
    cases = [(int(a.split(":")[0]), float(a.split(":")[1])) for a in sys.argv[1:]]
# LG_VARIANTS="640,2 320,4 ...": with a -DRNNT_LG_PROBE build (WARP_RNNT_AMD_LIB, WARP_RNNT_AMD_NO_NATIVE_BINDING=1) every
# case is run once per "threads,float4-per-thread" cover of the row-per-workgroup kernel (RNNT_LG_VARIANT, read per call)
variants = os.environ.get("LG_VARIANTS", "").split() or [None]
chunks = os.environ.get("XCD_CHUNKS", "").split() or [None]      # RNNT_XCD_CHUNK values (probe build): XCD run lengths
cases = [(V, gb, v, c) for V, gb in cases for v in variants for c in chunks]
backward = bool(os.environ.get("LSM_BACKWARD"))               # time ops.log_softmax_backward (three streams) instead
inplace = bool(os.environ.get("LSM_INPLACE"))                 # out = x (how c5 runs: 144 GB of logits leave no room for a copy)
for V, gb, variant, chunk in cases:
    if chunk is not None:
        os.environ["RNNT_XCD_CHUNK"] = chunk
        print(f"[xcd run {chunk:>5s}] ", end="")
    if variant is not None:
        th, nv = (int(t) for t in variant.split(","))
        if V > th * 4 * nv or V % 4:
            continue
        os.environ["RNNT_LG_VARIANT"] = variant
        print(f"[{variant:>8s}] ", end="")
    rows = int(gb * 1e9 / 4 / V)
    x = torch.randn(rows, V, device=dev)
    out = x if inplace else torch.empty_like(x)
    if backward:
        y = ops.log_softmax(x)
        run = lambda: ops.log_softmax_backward(x, y, grad_in=out)
    else:
        run = lambda: ops.log_softmax(x, out=out)
    for _ in range(80):
        run()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = statistics.median(ts)
    copy_txt = ""
    if os.environ.get("LSM_COPY"):
        # the yardstick: a plain streaming copy of the same bytes on the same stream, same conditioning -- torch's
        # vectorised elementwise kernel, out of place (dst = src * 1) or, for the in-place runs, x *= 1 (reads and writes
        # the same addresses, as the in-place log-softmax does)
        cp = (lambda: torch.mul(x, 1.0, out=x)) if inplace else (lambda: torch.mul(x, 1.0, out=out))
        for _ in range(20):
            cp()
        tc = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                cp()
            e1.record()
            torch.cuda.synchronize()
            tc.append(e0.elapsed_time(e1) / 10)
        cms = statistics.median(tc)
        copy_txt = f"   copy {2 * rows * V * 4 / cms / 1e9:6.2f} TB/s  (log-softmax / copy {cms / ms:5.3f})"
    streams = 3 if backward else 2
    print(f"V={V:6d} rows={rows:9d} in={rows * V * 4 / 1e9:5.2f} GB  {ms * 1e3:8.1f} us  {streams * rows * V * 4 / ms / 1e9:6.2f} TB/s"
          f"{' (backward: dy, y in, dx out)' if backward else ''}{' (in place)' if inplace else ''}{copy_txt}", flush=True)
    del x, out
