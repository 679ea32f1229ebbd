#!/usr/bin/env python
"""Times the alpha+beta sweeps alone on every lattice kernel of the in-tree library, in one process, for a list of shapes:

    ws   one workgroup per sweep, compute + I/O wave pairs (csrc/lattice_ws.hip; U <= 512)
    wd   one workgroup per column block ("distributed")    (csrc/lattice_wd.hip; RNNT_WD_K16_FROM_T picks its block size)
    wl   one workgroup per sweep, wd's three wave roles    (csrc/lattice_wd.hip: k_lattice_wl; 64 < U <= 320)
    auto what launch_lattice picks by shape

and checks that ws, wd, wl and auto leave the same bits in the alpha / beta planes (full-length utterances, so every cell
is live).  WARP_RNNT_AMD_LIB=<variant .so> times another build of the library.  HIP events around 5 back-to-back launches, 10 rounds, median / min in us.

    python tools/lattice_routes.py [N,T,U ...]
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from warp_rnnt_amd import _lib, debug  # noqa: E402

DEFAULT = ["16,1500,64", "16,1500,128", "16,1500,300", "16,1500,512", "8,3000,500", "24,1500,300", "32,1500,300",
           "64,1500,300", "128,1500,300", "16,700,100", "16,400,100", "32,250,100", "16,150,40", "32,150,20", "64,500,100",
           "32,1000,200", "32,500,200", "64,300,128", "16,1500,600"]


def run(shape, L, dev):
    N, T, U = shape
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    lp2 = torch.log_softmax(torch.randn(N, T, U, 2, device=dev, generator=g), -1).contiguous()
    xn = torch.full((N,), T, dtype=torch.int32, device=dev)
    yn = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
    costs = torch.empty((N,), device=dev)
    grads = torch.empty((N, T, U, 2), device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ws = torch.zeros((L.rnnt_amd_workspace_size(N, T, U),), dtype=torch.uint8, device=dev)
    cells = N * T * U
    plane = (cells * 4 + 255) // 256 * 256
    row = {}
    planes = {}
    for name in ("ws", "wd", "wl", "auto"):
        if name == "ws" and U > 512:
            continue
        if name == "wl" and not 64 < U <= 320:
            continue
        debug.set_lattice_kernel(name)
        # the pairs are consumed in place when gradients are produced: rebuild them for every route
        st = L.rnnt_amd_loss(stream, ws.data_ptr(), 1, lp2.data_ptr(), None, xn.data_ptr(), yn.data_ptr(),
                             costs.data_ptr(), grads.data_ptr(), 1, N, T, U, 2, 0, 0.0)
        assert st == 0, st
        torch.cuda.synchronize()
        planes[name] = (ws[:cells * 4].view(torch.float32).clone(), ws[plane:plane + cells * 4].view(torch.float32).clone(),
                        float(costs.double().sum().item()), debug.last_lattice_kernel())
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
        row[name] = (statistics.median(times), min(times))
    base = "ws" if "ws" in planes else "wd"
    same = []
    for k in ("wd", "wl", "auto"):
        if k in planes and k != base:
            ok = torch.equal(planes[base][0], planes[k][0]) and torch.equal(planes[base][1], planes[k][1])
            same.append(f"{k} {'=' if ok else 'DIFFERS FROM'} {base}")
    cells_txt = " ".join(f"{k} {row[k][0]:7.1f} ({row[k][1]:6.1f})" if k in row else f"{k}       -         "
                         for k in ("ws", "wd", "wl", "auto"))
    print(f"N={N:4d} T={T:5d} U={U:4d}   {cells_txt}   [{', '.join(same)}; auto -> {planes['auto'][3]}]", flush=True)
    debug.set_lattice_kernel("auto")


def main():
    shapes = [tuple(int(x) for x in a.split(",")) for a in (sys.argv[1:] or DEFAULT)]
    dev = torch.device("cuda:0")
    L = _lib.load()
    print("us per alpha+beta launch: median (min)")
    for shape in shapes:
        run(shape, L, dev)


if __name__ == "__main__":
    main()
