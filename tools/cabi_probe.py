#!/usr/bin/env python
"""Times the REFERENCE-NAMED C entry points the way the reference's own binding calls them (pytorch_binding/
binding.cpp:58-99): grads = zeros_like(xs), counts = zeros(N, 2U), alphas / betas / costs = empty, then
run_warp_rnnt_gather (xs = (N,T,U,2) pairs) or run_warp_rnnt (xs = dense (N,T,U,V) log-probs) -- through ctypes,
straight into libwarp_rnnt_amd.so -- next to the native workspace entry on the same inputs.

    python tools/cabi_probe.py [c2 c4 ...]           (shapes of bench.py's configs; default c2 c4)

and the three compact entry points (run_gather_for_compact, run_warp_rnnt_compact, run_scatter_grad_for_compact) in the
sequence of binding.cpp:139-204 / 209-247 on a ragged batch of the same shape, next to the native compact entries.

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


def ragged_compact_batch(N, T, U, V, dev, seed=11):
    """A ragged batch in the reference's compact packing (benchmark.py:20-24's length rule: frames in [T/2, T], labels in
    [U/2, U-1], shifted so that the maxima are T and U-1): xs (STU,V) log-probs, ys (sum yn,), xn, yn."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    xn = torch.randint(T // 2, T + 1, (N,), generator=g, dtype=torch.int32)
    yn = torch.randint(U // 2, U, (N,), generator=g, dtype=torch.int32)
    xn = (xn + (T - xn.max())).to(torch.int32)
    yn = (yn + (U - 1 - yn.max())).to(torch.int32)
    STU = int((xn.long() * (yn.long() + 1)).sum())
    gd = torch.Generator(device=dev)
    gd.manual_seed(seed)
    xs = torch.log_softmax(torch.randn((STU, V), device=dev, generator=gd), -1)
    ys = torch.randint(1, V, (int(yn.sum()),), dtype=torch.int32, device=dev, generator=gd)
    return xs, ys, xn.to(dev), yn.to(dev)


def time_compact_entries(xs, ys, xn, yn, reps=5, backward=True):
    """ms per call of the reference binding's compact sequence on the reference-named entry points (binding.cpp:139-204,
    209-247: exclusive prefix sums with torch ops, run_gather_for_compact, run_warp_rnnt_compact, and -- backward --
    run_scatter_grad_for_compact into zeros), its allocations included, NULL stream as there; next to the native entries
    on the same tensors (rnnt_amd_compact_offsets + rnnt_amd_loss_compact [+ rnnt_amd_compact_scatter_grads] through
    warp_rnnt_amd.ops, one read-back).  The reference's four host read-backs (yn.sum, xn.max, yn.max, the last prefix) are
    kept in the timed sequence: they are part of what its binding does per call.  `*_calls_ms` / `*_call_ms`: the entry
    points alone, on buffers and prefix sums prepared outside the timed region."""
    L = _lib.load()
    dev = xs.device
    N, V = xn.shape[0], xs.shape[1]
    out = {}

    def shim_call():
        n_labels = int(yn.sum().item())
        Tm, Um = int(xn.max().item()), int(yn.max().item()) + 1
        mem = (xn * (yn + 1)).cumsum(0, dtype=torch.int32)
        lab = yn.cumsum(0, dtype=torch.int32)
        STU = int(mem[-1].item())
        assert STU == xs.shape[0] and n_labels == ys.numel()
        mem_pref = torch.cat([mem.new_zeros(1), mem[:-1]]).contiguous()
        lab_pref = torch.cat([lab.new_zeros(1), lab[:-1]]).contiguous()
        gather_xs = torch.empty((STU, 2), device=dev)
        loc = torch.zeros((STU,), dtype=torch.int64, device=dev)
        L.run_gather_for_compact(xs.data_ptr(), ys.data_ptr(), xn.data_ptr(), yn.data_ptr(), gather_xs.data_ptr(),
                                 loc.data_ptr(), mem_pref.data_ptr(), lab_pref.data_ptr(), N, Tm, Um, V, 0)
        costs = torch.empty((N,), device=dev)
        counts = torch.zeros((n_labels * 2 + 2 * N,), dtype=torch.int32, device=dev)
        betas = torch.empty((STU,), device=dev)
        alphas, grads = torch.empty_like(betas), torch.empty_like(gather_xs)
        L.run_warp_rnnt_compact(counts.data_ptr(), alphas.data_ptr(), betas.data_ptr(), gather_xs.data_ptr(),
                                grads.data_ptr(), costs.data_ptr(), xn.data_ptr(), yn.data_ptr(), mem_pref.data_ptr(),
                                lab_pref.data_ptr(), N, Tm, Um, 0.0, True)
        shim_call.res = (costs, grads, loc, mem)
        return costs, grads, loc, mem

    ones = torch.ones((N,), device=dev)

    def shim_train():
        costs, grads, loc, cum = shim_call()
        dense = torch.zeros((xs.shape[0], V), device=dev)
        L.run_scatter_grad_for_compact(ones.data_ptr(), grads.data_ptr(), loc.data_ptr(), cum.data_ptr(), dense.data_ptr(),
                                       xs.shape[0], N, V, 0)

    def native_call():
        native_call.res = ops.loss_compact(xs, ys, xn, yn)
        return native_call.res

    def native_train():
        costs, grads, loc = native_call()
        cum = (xn * (yn + 1)).cumsum(0, dtype=torch.int32)
        ops.compact_scatter_grads(ones, grads, cum, loc, V, 0)

    # the entry points alone: everything the binding allocates and reads back is done once, outside the timed region
    n_labels = int(yn.sum().item())
    Tm, Um = int(xn.max().item()), int(yn.max().item()) + 1
    mem = (xn * (yn + 1)).cumsum(0, dtype=torch.int32)
    lab = yn.cumsum(0, dtype=torch.int32)
    STU = int(mem[-1].item())
    mem_pref = torch.cat([mem.new_zeros(1), mem[:-1]]).contiguous()
    lab_pref = torch.cat([lab.new_zeros(1), lab[:-1]]).contiguous()
    pre = dict(gather_xs=torch.empty((STU, 2), device=dev), loc=torch.zeros((STU,), dtype=torch.int64, device=dev),
               costs=torch.empty((N,), device=dev), counts=torch.zeros((n_labels * 2 + 2 * N,), dtype=torch.int32, device=dev),
               betas=torch.empty((STU,), device=dev), alphas=torch.empty((STU,), device=dev),
               grads=torch.empty((STU, 2), device=dev))

    def shim_calls_only():
        L.run_gather_for_compact(xs.data_ptr(), ys.data_ptr(), xn.data_ptr(), yn.data_ptr(), pre["gather_xs"].data_ptr(),
                                 pre["loc"].data_ptr(), mem_pref.data_ptr(), lab_pref.data_ptr(), N, Tm, Um, V, 0)
        L.run_warp_rnnt_compact(pre["counts"].data_ptr(), pre["alphas"].data_ptr(), pre["betas"].data_ptr(),
                                pre["gather_xs"].data_ptr(), pre["grads"].data_ptr(), pre["costs"].data_ptr(), xn.data_ptr(),
                                yn.data_ptr(), mem_pref.data_ptr(), lab_pref.data_ptr(), N, Tm, Um, 0.0, True)

    offs = torch.empty((N + 1 + 4,), dtype=torch.int64, device=dev)
    loffs = torch.empty((N + 1,), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    assert L.rnnt_amd_compact_offsets(stream, xn.data_ptr(), yn.data_ptr(), N, offs.data_ptr(), loffs.data_ptr(),
                                      offs[N + 1:].data_ptr()) == 0
    wsb = torch.empty((L.rnnt_amd_workspace_size_compact(N, STU, Tm, Um),), dtype=torch.uint8, device=dev)

    def native_calls_only():
        st = L.rnnt_amd_loss_compact(stream, wsb.data_ptr(), xs.data_ptr(), ys.data_ptr(), xn.data_ptr(), yn.data_ptr(),
                                     offs.data_ptr(), loffs.data_ptr(), pre["costs"].data_ptr(), pre["grads"].data_ptr(),
                                     pre["loc"].data_ptr(), N, STU, Tm, Um, V, 0, 0.0)
        assert st == 0, st

    out["cabi_compact_calls_ms"] = round(timed(shim_calls_only, reps), 4)
    out["native_compact_call_ms"] = round(timed(native_calls_only, reps), 4)

    # The shims work on the NULL stream, torch on its own (blocking) stream: the two serialise against each other, and the
    # events of timed() are recorded on torch's stream, so they bracket the NULL-stream work as well.
    out["cabi_compact_ms"] = round(timed(shim_call, reps), 4)
    out["native_compact_ms"] = round(timed(native_call, reps), 4)
    if backward:
        out["cabi_compact_train_ms"] = round(timed(shim_train, reps), 4)
        out["native_compact_train_ms"] = round(timed(native_train, reps), 4)
    torch.cuda.synchronize()
    assert L.rnnt_amd_compact_last_status() == 0
    c_shim, c_nat = shim_call.res[0], native_call.res[0]
    out["cabi_compact_vs_native_max_rel_cost"] = float((c_shim / c_nat - 1).abs().max())
    out["cabi_compact_note"] = ("binding.cpp:139-204's sequence on run_gather_for_compact + run_warp_rnnt_compact "
                                "(+ run_scatter_grad_for_compact: *_train_ms) with its allocations and four read-backs, "
                                "next to ops.loss_compact (+ ops.compact_scatter_grads) on the same ragged batch, "
                                "tools/cabi_probe.py")
    return out


def probe_compact(name):
    N, T, U, V = SHAPES[name]
    dev = torch.device("cuda:0")
    xs, ys, xn, yn = ragged_compact_batch(N, T, U, V, dev)
    reps = 20 if N * T * U * V < 10 ** 8 else 5
    r = time_compact_entries(xs, ys, xn, yn, reps)
    print(f"{name} ragged, compact packing: N={N} T<={T} U<={U} V={V}, {xs.shape[0]} of {N * T * U} cells "
          f"(costs shim vs native: max rel {r['cabi_compact_vs_native_max_rel_cost']:.1e})")
    for k in ("cabi_compact_calls_ms", "native_compact_call_ms", "cabi_compact_ms", "native_compact_ms",
              "cabi_compact_train_ms", "native_compact_train_ms"):
        print(f"    {k:58s} {r[k]:8.4f} ms")
    print(f"    shim / native: the entry points alone {r['cabi_compact_calls_ms'] / r['native_compact_call_ms']:.2f}x, "
          f"with each binding's allocations and read-backs {r['cabi_compact_ms'] / r['native_compact_ms']:.2f}x, "
          f"+ scatter {r['cabi_compact_train_ms'] / r['native_compact_train_ms']:.2f}x")
    return r


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
        probe_compact(nm)
