#!/usr/bin/env python
"""Soak of the column-block lattice kernel's hand-over protocol: the loss entry on diagonal-major pairs, back to back for
--seconds, on shapes that run k_lattice_wd with rings (c4: five column blocks, blocks of 16 diagonals; a ragged batch of
three column blocks, blocks of 8).  Every launch's costs and gradients are compared bit for bit with the first launch's,
and the redo flags of the workspace are read after every launch: bit 1 = a hand-over wait timed out and the sweep was
redone by the kernel behind (allowed, never observed outside the short-spin build; counted here).

    python tools/wd_soak.py --seconds 60 [--procs 3]      (--procs: that many copies at once on the one GPU)"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(seconds, seed):
    import numpy as np
    import torch
    import oracle
    from helpers import make_case, np_log_softmax32
    import warp_rnnt_amd
    from warp_rnnt_amd import ops
    dev = torch.device("cuda:0")
    L = ops._lib.load()
    cases = []
    shapes = ((16, 1500, 300, False), (12, 700, 180, True))
    if os.environ.get("WD_SOAK_SHAPES"):            # "N,T,U[,r] N,T,U ...": r = ragged
        shapes = tuple((int(f[0]), int(f[1]), int(f[2]), len(f) > 3) for f in (x.split(",") for x in os.environ["WD_SOAK_SHAPES"].split()))
    for (N, T, U, ragged) in shapes:
        logits, labels, xn, yn = make_case(seed + T, N, T, U, 5, ragged=ragged)
        lp2 = torch.tensor(oracle.gather_f32(np_log_softmax32(logits), labels, 0), device=dev)
        cases.append((N, T, U, lp2, torch.tensor(xn, device=dev), torch.tensor(yn, device=dev)))
    first, launches, lost, mism, shown = {}, 0, 0, 0, 0
    t0 = time.time()
    while time.time() - t0 < seconds:
        for i, (N, T, U, lp2, txn, tyn) in enumerate(cases):
            ws = torch.empty((L.rnnt_amd_workspace_size(N, T, U),), dtype=torch.uint8, device=dev)
            costs = torch.empty((N,), device=dev)
            grads = torch.empty((N, T, U, 2), device=dev)
            for rep in range(25):
                st = L.rnnt_amd_loss(torch.cuda.current_stream().cuda_stream, ws.data_ptr(), 1, lp2.data_ptr(), None,
                                     txn.data_ptr(), tyn.data_ptr(), costs.data_ptr(), grads.data_ptr(), 0, N, T, U, 2, 0, 0.0)
                assert st == 0, st
                launches += 1
                off = L.rnnt_amd_debug_redo_offset(N, T, U)
                flags = ws[off:off + 8 * N].view(torch.int32)
                if not (U > 64 and warp_rnnt_amd.last_lattice_kernel() == "lattice_wd"):
                    flags = torch.zeros_like(flags)      # (only the ring kernel prepares and uses the flags)
                nl = int((flags & 2).ne(0).sum().item())
                lost += nl
                if flags.ne(0).any() and shown < 8:
                    shown += 1
                    nz = flags.ne(0).nonzero().flatten().tolist()
                    same = first.get(i) is not None and torch.equal(costs, first[i][0]) and torch.equal(grads, first[i][1])
                    print(f"  flags raised: case {i}, sweeps {nz} (2 n + direction), values {[hex(int(flags[j]) & 0xffffffff) for j in nz]}; "
                          f"results equal to the first launch's: {same}", flush=True)
                cells = N * T * U
                planes = (ws[:4 * cells].view(torch.float32), ws[4 * cells:8 * cells].view(torch.float32))   # alphas, betas (api.hip: carve)
                if i not in first:
                    first[i] = (costs.clone(), grads.clone(), planes[0].clone(), planes[1].clone())
                elif not (torch.equal(costs, first[i][0]) and torch.equal(grads, first[i][1])):
                    mism += 1
                    if mism <= 6:
                        dc = (costs - first[i][0]).abs()
                        dg = (grads - first[i][1]).abs().reshape(N, -1).max(dim=1).values
                        print(f"  mismatch: case {i} (N={N}, T={T}, U={U}); flags {flags.tolist()}; utterances whose costs differ "
                              f"{dc.ne(0).nonzero().flatten().tolist()} (max {float(dc.max()):.3e}); whose gradients differ "
                              f"{dg.ne(0).nonzero().flatten().tolist()} (max {float(dg.max()):.3e}; NaN: {bool(torch.isnan(grads).any())})", flush=True)
                        for name, now, ref in (("alphas", planes[0], first[i][2]), ("betas", planes[1], first[i][3])):
                            badp = (now != ref) & ~(torch.isnan(now) & torch.isnan(ref))
                            if badp.any():
                                idx = badp.nonzero().flatten()
                                nn, rem = idx // (T * U), idx % (T * U)
                                rr, cc = rem // U, rem % U
                                k3 = idx[:4].tolist()
                                print(f"    {name} plane (diagonal-major): {int(badp.sum())} elements differ; utterances {sorted(set(nn.tolist()))}; rows "
                                      f"{int(rr.min())}..{int(rr.max())}, columns {int(cc.min())}..{int(cc.max())}; first: "
                                      f"{[(int(rr[j]), int(cc[j]), float(now[k3[j]]), float(ref[k3[j]])) for j in range(len(k3))]} (row, col, now, first launch)", flush=True)
                            else:
                                print(f"    {name} plane: equal to the first launch's", flush=True)
                        for nb in dg.ne(0).nonzero().flatten().tolist()[:2]:
                            bad = (grads[nb] != first[i][1][nb]).any(dim=-1)          # (T, U)
                            tt, uu = bad.nonzero(as_tuple=True)
                            dd = tt + uu
                            print(f"    utterance {nb}: {int(bad.sum())} cells differ; t {int(tt.min())}..{int(tt.max())}, u {int(uu.min())}.."
                                  f"{int(uu.max())}, diagonal {int(dd.min())}..{int(dd.max())}; columns hit: {sorted(set((uu // 64).tolist()))} (blocks of 64); "
                                  f"launch index in this workspace: {rep}", flush=True)
    print(f"wd soak: {launches} launches in {time.time() - t0:.0f} s (seed {seed}); results differing from the first launch: {mism}; "
          f"sweeps redone after a lost hand-over: {lost}; kernel of the last launch: {warp_rnnt_amd.last_lattice_kernel()}", flush=True)
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
