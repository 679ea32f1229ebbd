#!/usr/bin/env python
"""A/B timing of the lattice kernel alone for several builds of the library.

    python tools/lattice_probe.py [--shape N,T,U] name1:-DFLAG1,-DFLAG2 name2: ...
    python tools/lattice_probe.py [--shape N,T,U] main:        (the in-tree library, no build)

Each variant is compiled (hipcc, extra flags) into its own shared object, loaded with ctypes and
the alpha/beta sweep is timed with HIP events in interleaved rounds (median / min in us)."""
import ctypes
import os
import shutil
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from warp_rnnt_amd import _build, _lib  # noqa: E402


def build_variant(name, flags):
    objdir = os.path.join(ROOT, "tools", "_probe", name)
    os.makedirs(objdir, exist_ok=True)
    lib = os.path.join(objdir, "lib.so")
    objs = []
    for src in _build.SOURCES:
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"] + flags +
                              ["-c", os.path.join(_build.CSRC, src), "-o", o])
        objs.append(o)
    subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    L = ctypes.CDLL(lib)
    for sym, (res, args) in _lib.SYMBOLS.items():
        fn = getattr(L, sym)
        fn.restype, fn.argtypes = res, args
    return L


def main():
    args = sys.argv[1:]
    shape = (16, 1500, 300)
    if args and args[0] == "--shape":
        shape = tuple(int(x) for x in args[1].split(","))
        args = args[2:]
    if "--prebuilt" in args:
        args.remove("--prebuilt")
    variants = []
    for a in args:
        name, _, fl = a.partition(":")
        variants.append((name, [f for f in fl.split(",") if f]))
    if not torch.cuda.is_available():       # build-only mode (no GPU): compile and exit
        for name, flags in variants:
            build_variant(name, flags)
        print("built", [v[0] for v in variants])
        return
    if len(variants) > 1:
        # one process per variant: template kernels are STB_GNU_UNIQUE symbols, so two builds of the
        # library in one process would silently share the first one's kernels
        for rnd in range(2):
            for name, flags in variants:
                subprocess.call([sys.executable, os.path.abspath(__file__), "--shape",
                                 ",".join(map(str, shape)), name + ":" + ",".join(flags)])
        return
    N, T, U = shape
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    lp2 = torch.log_softmax(torch.randn(N, T, U, 2, device=dev, generator=g), -1).contiguous()
    xn = torch.full((N,), T, dtype=torch.int32, device=dev)
    yn = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
    costs = torch.empty((N,), device=dev)
    grads = torch.empty((N, T, U, 2), device=dev)
    name, flags = variants[0]
    pre = os.path.join(ROOT, "tools", "_probe", name, "lib.so")
    if name == "main":                      # the in-tree library as built by warp_rnnt_amd._build
        pre = _lib.lib_path()
    if os.path.exists(pre):
        L = ctypes.CDLL(pre)
        for sym, (res, a_) in _lib.SYMBOLS.items():
            if hasattr(L, sym):
                fn = getattr(L, sym)
                fn.restype, fn.argtypes = res, a_
    else:
        L = build_variant(name, flags)
    stream = torch.cuda.current_stream().cuda_stream
    ws = torch.empty((L.rnnt_amd_workspace_size(N, T, U),), dtype=torch.uint8, device=dev)
    st = L.rnnt_amd_loss(stream, ws.data_ptr(), 1, lp2.data_ptr(), None, xn.data_ptr(), yn.data_ptr(),
                         costs.data_ptr(), grads.data_ptr(), 1, N, T, U, 2, 0, 0.0)
    assert st == 0
    torch.cuda.synchronize()
    csum = float(costs.double().sum().item())
    times = []
    for rnd in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            L.rnnt_amd_debug_lattice_only(stream, ws.data_ptr(), xn.data_ptr(), yn.data_ptr(), N, T, U)
        e1.record()
        torch.cuda.synchronize()
        if rnd >= 2:
            times.append(e0.elapsed_time(e1) * 1000 / 5)
    print(f"{name:24s} median {statistics.median(times):8.1f} us   min {min(times):8.1f} us   "
          f"sum(costs) {csum:.4f}", flush=True)


if __name__ == "__main__":
    main()
