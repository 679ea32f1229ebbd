#!/usr/bin/env python
"""Times the REFERENCE-NAMED C entry points the way the reference's own binding calls them (pytorch_binding/
binding.cpp:58-99): grads = zeros_like(xs), counts = zeros(N, 2U), alphas / betas / costs = empty, then
run_warp_rnnt_gather (xs = (N,T,U,2) pairs) or run_warp_rnnt (xs = dense (N,T,U,V) log-probs) -- through ctypes,
straight into libwarp_rnnt_amd.so -- next to the native workspace entry on the same inputs.

    python tools/cabi_probe.py [c2 c4 ...]           (shapes of bench.py's configs; default c2 c4)

ms per call, HIP events around `reps` back-to-back calls; "alloc+call" includes the binding's allocations and the
zero-fills the C contract asks for, "call" is the entry point alone on buffers zeroed outside the timed region."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from warp_rnnt_amd import _lib, ops  # noqa: E402

SHAPES = {"c2": (16, 150, 40, 28), "c3": (32, 150, 20, 5000), "c4": (16, 1500, 300, 50)}


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return statistics.median(out)


def time_entries(lp, ys, xn, yn, reps=5):
    """For bench.py: ms per call of the two reference-named entry points on the caller's tensors, the binding's
    allocations and zero-fills included (binding.cpp:58-99), dense log-probs `lp` (N,T,U,V)."""
    N, T, U, V = lp.shape
    dev = lp.device
    L = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    lp2 = ops.gather(lp, ys, 0)

    def gather_call():
        grads = torch.zeros_like(lp2)
        counts = torch.zeros((N, 2 * U), dtype=torch.int32, device=dev)
        al, be, costs = torch.empty((N, T, U), device=dev), torch.empty((N, T, U), device=dev), torch.empty((N,), device=dev)
        st = L.run_warp_rnnt_gather(stream, counts.data_ptr(), al.data_ptr(), be.data_ptr(), lp2.data_ptr(),
                                    grads.data_ptr(), costs.data_ptr(), xn.data_ptr(), yn.data_ptr(), N, T, U, 0.0)
        assert st == 0, st

    def dense_call():
        grads = torch.zeros_like(lp)
        counts = torch.zeros((N, 2 * U), dtype=torch.int32, device=dev)
        al, be, costs = torch.empty((N, T, U), device=dev), torch.empty((N, T, U), device=dev), torch.empty((N,), device=dev)
        st = L.run_warp_rnnt(stream, counts.data_ptr(), al.data_ptr(), be.data_ptr(), ys.data_ptr(), lp.data_ptr(),
                             grads.data_ptr(), costs.data_ptr(), xn.data_ptr(), yn.data_ptr(), N, T, U, V, 0, 0.0)
        assert st == 0, st

    return {"cabi_gather_ms": round(timed(gather_call, reps), 4), "cabi_dense_ms": round(timed(dense_call, reps), 4),
            "cabi_note": "run_warp_rnnt_gather / run_warp_rnnt through ctypes with the reference binding's own "
                         "allocations and zero-fills (binding.cpp:58-99), tools/cabi_probe.py"}


def probe(name):
    N, T, U, V = SHAPES[name]
    dev = torch.device("cuda:0")
    L = _lib.load()
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    lp = torch.log_softmax(torch.randn((N, T, U, V), device=dev, generator=g), -1)
    ys = torch.randint(1, V, (N, U - 1), dtype=torch.int32, device=dev, generator=g)
    xn = torch.full((N,), T, dtype=torch.int32, device=dev)
    yn = torch.full((N,), U - 1, dtype=torch.int32, device=dev)
    lp2 = ops.gather(lp, ys, 0) if hasattr(ops, "gather") else None
    if lp2 is None:
        idx = torch.zeros((N, 1, U, 2), dtype=torch.int64, device=dev)
        idx[:, 0, :U - 1, 1] = ys.long()
        lp2 = torch.gather(lp, 3, idx.expand(N, T, U, 2)).contiguous()
    stream = torch.cuda.current_stream().cuda_stream
    reps = 20 if N * T * U * V < 10 ** 8 else 5
    rows = {}

    def gather_call(alloc):
        def fn():
            if alloc or not hasattr(fn, "bufs"):
                fn.bufs = (torch.zeros_like(lp2), torch.zeros((N, 2 * U), dtype=torch.int32, device=dev),
                           torch.empty((N, T, U), device=dev), torch.empty((N, T, U), device=dev),
                           torch.empty((N,), device=dev))
            grads, counts, al, be, costs = fn.bufs
            if not alloc:
                counts.zero_()
            st = L.run_warp_rnnt_gather(stream, counts.data_ptr(), al.data_ptr(), be.data_ptr(), lp2.data_ptr(),
                                        grads.data_ptr(), costs.data_ptr(), xn.data_ptr(), yn.data_ptr(), N, T, U, 0.0)
            assert st == 0, st
        return fn

    def dense_call(alloc):
        def fn():
            if alloc or not hasattr(fn, "bufs"):
                fn.bufs = (torch.zeros_like(lp), torch.zeros((N, 2 * U), dtype=torch.int32, device=dev),
                           torch.empty((N, T, U), device=dev), torch.empty((N, T, U), device=dev),
                           torch.empty((N,), device=dev))
            grads, counts, al, be, costs = fn.bufs
            if not alloc:
                counts.zero_()
                # (the dense contract wants grads zeroed: only the two live slots of a cell are written.  The same
                #  slots are rewritten by every call, so for timing the call alone the buffer is zeroed once)
            st = L.run_warp_rnnt(stream, counts.data_ptr(), al.data_ptr(), be.data_ptr(), ys.data_ptr(), lp.data_ptr(),
                                 grads.data_ptr(), costs.data_ptr(), xn.data_ptr(), yn.data_ptr(), N, T, U, V, 0, 0.0)
            assert st == 0, st
        return fn

    rows["run_warp_rnnt_gather  alloc+call"] = timed(gather_call(True), reps)
    rows["run_warp_rnnt_gather  call"] = timed(gather_call(False), reps)
    rows["run_warp_rnnt (dense) alloc+call"] = timed(dense_call(True), reps)
    rows["run_warp_rnnt (dense) call"] = timed(dense_call(False), reps)
    rows["native rnnt_amd_loss, gathered in / gathered grads out"] = timed(
        lambda: ops.loss(lp2, None, xn, yn, ops.IN_LOG_PROBS_GATHERED, ops.GRADS_GATHERED), reps)
    rows["native rnnt_amd_loss, dense in / gathered grads out"] = timed(
        lambda: ops.loss(lp, ys, xn, yn, ops.IN_LOG_PROBS_DENSE, ops.GRADS_GATHERED), reps)
    rows["native rnnt_amd_loss, dense in / dense grads out"] = timed(
        lambda: ops.loss(lp, ys, xn, yn, ops.IN_LOG_PROBS_DENSE, ops.GRADS_DENSE), reps)
    # same answers
    c_ref, _ = ops.loss(lp2, None, xn, yn, ops.IN_LOG_PROBS_GATHERED, ops.GRADS_GATHERED)
    fn = gather_call(True)
    fn()
    torch.cuda.synchronize()
    rel = float((fn.bufs[4] / c_ref - 1).abs().max())
    print(f"{name}: N={N} T={T} U={U} V={V}   (costs of the C entry vs the native entry: max rel {rel:.1e})")
    for k, v in rows.items():
        print(f"    {k:58s} {v:8.4f} ms")
    return rows


if __name__ == "__main__":
    for nm in (sys.argv[1:] or ["c2", "c4"]):
        probe(nm)
