#!/usr/bin/env python
"""Soak of the column-block lattice kernels (k_lattice_wd with its hand-over through L2, k_lattice_wl, the plain launch
of single-column-block lattices) under the load that found round 5's two silent wrong-answer bugs: several processes
launching on the one GPU at the same time (csrc/lattice_step.h: wait_lds).

Every launch is the whole loss entry on gathered log-probs (re-layout, ring preparation, sweeps, gradients), and every
launch's costs, gradients AND alpha / beta planes are compared bit for bit with a REFERENCE COMPUTED ONCE BY THE
OTHER KERNEL -- k_lattice_ws, the compiler-scheduled one-workgroup-per-sweep kernel without in-place reloads and without
hand-over through L2 (a first launch that is already wrong, or a deterministic error, must not pass as
"the same as the first launch").  The comparisons run on the device and are read back once per batch of launches, so
the processes keep the GPU saturated; the redo flags (bit 1 = a hand-over wait timed out and the kernel behind redid
the sweep: allowed, counted) are accumulated the same way.  A case that shows a mismatch is re-run launch by launch with
a description of what differs.

Shapes (N, T, U; r = ragged lengths) -- the six-shape set, and a seventh since the end of round 6:
    16,1500,300     c4: five column blocks, k_lattice_wd with rings, blocks of 16 diagonals (RNNT_WD_K16_FROM_T=1000000: of 8)
    12,700,180,r    three column blocks, k_lattice_wd with rings
    16,400,100      two column blocks, k_lattice_wl
    32,250,100,r    two column blocks, k_lattice_wl, ragged
    32,500,200      four column blocks on a short batch: k_lattice_wl in its large-LDS form
    16,150,40       c2's lattice: one column block, k_lattice_wd as a plain launch
    12,1300,120,r   two column blocks on a long sweep: k_lattice_wd with rings (from T >= 1200; csrc/lattice.hip), ragged
WD_SOAK_SHAPES="N,T,U[,r] ..." overrides.

    python tools/wd_soak.py --seconds 60 [--procs 6]      (--procs: that many copies at once on the one GPU)
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = ((16, 1500, 300, False), (12, 700, 180, True), (16, 400, 100, False), (32, 250, 100, True), (32, 500, 200, False),
          (16, 150, 40, False), (12, 1300, 120, True))
BATCH = 40          # launches between two read-backs of the device-side counters


def make_pairs(seed, N, T, U, ragged, dev):
    """(N,T,U,2) gathered log-probs (blank, label) the way a V=5 log-softmax would give them, lengths per the
    reference's ragged rule (benchmark.py:20-24), all from a seeded torch generator on the device."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lp = torch.log_softmax(torch.randn((N, T, U, 5), device=dev, generator=g), -1)
    lab = torch.randint(1, 5, (N, 1, U, 1), device=dev, generator=g).expand(N, T, U, 1)
    lp2 = torch.cat([lp[..., :1], lp.gather(3, lab)], -1).contiguous()
    if ragged:
        xn = torch.randint(max(T // 2, 1), T + 1, (N,), device=dev, generator=g, dtype=torch.int32)
        yn = torch.randint(U // 2, U, (N,), device=dev, generator=g, dtype=torch.int32)
        xn = xn + (T - xn.max())
        yn = yn + (U - 1 - yn.max())
    else:
        xn = torch.full((N,), T, device=dev, dtype=torch.int32)
        yn = torch.full((N,), U - 1, device=dev, dtype=torch.int32)
    return lp2, xn.contiguous(), yn.contiguous()


def child(seconds, seed):
    import torch
    from warp_rnnt_amd import debug, ops
    dev = torch.device("cuda:0")
    L = ops._lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    shapes = SHAPES
    if os.environ.get("WD_SOAK_SHAPES"):            # "N,T,U[,r] N,T,U ...": r = ragged
        shapes = tuple((int(f[0]), int(f[1]), int(f[2]), len(f) > 3) for f in (x.split(",") for x in os.environ["WD_SOAK_SHAPES"].split()))

    def launch(c, costs, grads):
        st = L.rnnt_amd_loss(stream, c["ws"].data_ptr(), 1, c["lp2"].data_ptr(), None, c["xn"].data_ptr(), c["yn"].data_ptr(),
                             costs.data_ptr(), grads.data_ptr(), 0, c["N"], c["T"], c["U"], 2, 0, 0.0)
        assert st == 0, st

    def planes(c):
        cells = c["N"] * c["T"] * c["U"]
        w = c["ws"]
        return w[:4 * cells].view(torch.float32), w[4 * cells:8 * cells].view(torch.float32)   # alphas, betas (api.hip: carve)

    def same(a, b):      # bit equality that takes NaN == NaN of the same payload (the planes of ragged batches hold padding)
        return torch.equal(a.view(torch.int32), b.view(torch.int32))

    cases = []
    pinned = L.rnnt_amd_debug_get_lattice_kernel()
    for i, (N, T, U, ragged) in enumerate(shapes):
        lp2, xn, yn = make_pairs(seed * 1000 + i, N, T, U, ragged, dev)
        c = dict(N=N, T=T, U=U, lp2=lp2, xn=xn, yn=yn,
                 ws=torch.empty((L.rnnt_amd_workspace_size(N, T, U),), dtype=torch.uint8, device=dev))
        # the reference: the same entry on k_lattice_ws, once, before any load
        L.rnnt_amd_debug_set_lattice_kernel(1)
        rc, rg = torch.empty((N,), device=dev), torch.empty((N, T, U, 2), device=dev)
        launch(c, rc, rg)
        torch.cuda.synchronize()
        assert debug.last_lattice_kernel() == "lattice_ws", debug.last_lattice_kernel()
        ra, rb = (p.clone() for p in planes(c))
        L.rnnt_amd_debug_set_lattice_kernel(pinned)
        c.update(ref=(rc, rg, ra, rb), costs=torch.empty_like(rc), grads=torch.empty_like(rg))
        launch(c, c["costs"], c["grads"])            # which kernel this case runs under the setting being soaked
        torch.cuda.synchronize()
        c["kernel"] = debug.last_lattice_kernel()
        c["rings"] = U > 64 and c["kernel"] == "lattice_wd"      # (only the ring kernel prepares and uses the flags)
        off = L.rnnt_amd_debug_redo_offset(N, T, U)
        c["flags"] = c["ws"][off:off + 8 * N].view(torch.int32)
        c["bad"] = torch.zeros((), dtype=torch.int64, device=dev)       # launches whose results differ from the reference
        c["lost"] = torch.zeros((), dtype=torch.int64, device=dev)      # sweeps redone after a lost hand-over
        c["launches"] = 0
        cases.append(c)

    def describe(c, i):
        """launch by launch, with the details (a case that has shown a mismatch)"""
        rc, rg, ra, rb = c["ref"]
        N, T, U = c["N"], c["T"], c["U"]
        shown = 0
        for rep in range(200):
            launch(c, c["costs"], c["grads"])
            torch.cuda.synchronize()
            a, b = planes(c)
            if same(c["costs"], rc) and same(c["grads"], rg) and same(a, ra) and same(b, rb):
                continue
            shown += 1
            dc = (c["costs"] - rc).abs()
            dg = (c["grads"] - rg).abs().reshape(N, -1).max(dim=1).values
            print(f"  mismatch: case {i} (N={N}, T={T}, U={U}, {c['kernel']}); flags {c['flags'].tolist() if c['rings'] else '-'}; "
                  f"utterances whose costs differ {dc.ne(0).nonzero().flatten().tolist()} (max {float(dc.max()):.3e}); whose gradients "
                  f"differ {dg.ne(0).nonzero().flatten().tolist()} (max {float(dg.max()):.3e}; NaN: {bool(torch.isnan(c['grads']).any())})",
                  flush=True)
            for name, now, ref in (("alphas", a, ra), ("betas", b, rb)):
                badp = now.view(torch.int32) != ref.view(torch.int32)
                if badp.any():
                    idx = badp.nonzero().flatten()
                    nn, rem = idx // (T * U), idx % (T * U)
                    rr, cc = rem // U, rem % U
                    k3 = idx[:4].tolist()
                    print(f"    {name} plane (diagonal-major): {int(badp.sum())} elements differ; utterances {sorted(set(nn.tolist()))}; "
                          f"rows {int(rr.min())}..{int(rr.max())}, columns {int(cc.min())}..{int(cc.max())}; first: "
                          f"{[(int(rr[j]), int(cc[j]), float(now[k3[j]]), float(ref[k3[j]])) for j in range(len(k3))]} "
                          f"(row, col, now, reference)", flush=True)
            if shown >= 3:
                break

    t0 = time.time()
    described = set()
    while time.time() - t0 < seconds:
        for i, c in enumerate(cases):
            rc, rg, ra, rb = c["ref"]
            for _ in range(BATCH):
                launch(c, c["costs"], c["grads"])
                a, b = planes(c)
                ok = ((c["costs"].view(torch.int32) == rc.view(torch.int32)).all() &
                      (c["grads"].view(torch.int32) == rg.view(torch.int32)).all() &
                      (a.view(torch.int32) == ra.view(torch.int32)).all() & (b.view(torch.int32) == rb.view(torch.int32)).all())
                c["bad"] += (~ok).to(torch.int64)
                if c["rings"]:
                    c["lost"] += (c["flags"] & 2).ne(0).sum()
            c["launches"] += BATCH
            if int(c["bad"].item()) and i not in described:        # (the batch's one read-back)
                described.add(i)
                describe(c, i)
    torch.cuda.synchronize()
    launches = sum(c["launches"] for c in cases)
    mism = sum(int(c["bad"].item()) for c in cases)
    lost = sum(int(c["lost"].item()) for c in cases)
    per = ", ".join(f"{c['N']}x{c['T']}x{c['U']} {c['kernel'].replace('lattice_', '')} {c['launches']}" +
                    (f" ({int(c['bad'].item())} BAD)" if int(c["bad"].item()) else "") for c in cases)
    print(f"wd soak: {launches} launches in {time.time() - t0:.0f} s (seed {seed}); results differing from the k_lattice_ws "
          f"reference: {mism}; sweeps redone after a lost hand-over: {lost}; per shape: {per}", flush=True)
    return 1 if mism else 0


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--procs", type=int, default=1)
    ap.add_argument("--child", type=int, default=-1)
    a = ap.parse_args()
    if a.child >= 0:
        sys.exit(child(a.seconds, a.child))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--seconds", str(a.seconds), "--child", str(100 + i)])
             for i in range(a.procs)]
    sys.exit(max(p.wait() for p in procs))
