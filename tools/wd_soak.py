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
    from warp_rnnt_amd import ops
    dev = torch.device("cuda:0")
    L = ops._lib.load()
    cases = []
    for (N, T, U, ragged) in ((16, 1500, 300, False), (12, 700, 180, True)):
        logits, labels, xn, yn = make_case(seed + T, N, T, U, 5, ragged=ragged)
        lp2 = torch.tensor(oracle.gather_f32(np_log_softmax32(logits), labels, 0), device=dev)
        cases.append((N, T, U, lp2, torch.tensor(xn, device=dev), torch.tensor(yn, device=dev)))
    first, launches, lost, mism = {}, 0, 0, 0
    t0 = time.time()
    while time.time() - t0 < seconds:
        for i, (N, T, U, lp2, txn, tyn) in enumerate(cases):
            ws = torch.empty((L.rnnt_amd_workspace_size(N, T, U),), dtype=torch.uint8, device=dev)
            costs = torch.empty((N,), device=dev)
            grads = torch.empty((N, T, U, 2), device=dev)
            for _ in range(25):
                st = L.rnnt_amd_loss(torch.cuda.current_stream().cuda_stream, ws.data_ptr(), 1, lp2.data_ptr(), None,
                                     txn.data_ptr(), tyn.data_ptr(), costs.data_ptr(), grads.data_ptr(), 0, N, T, U, 2, 0, 0.0)
                assert st == 0, st
                launches += 1
                off = L.rnnt_amd_debug_redo_offset(N, T, U)
                flags = ws[off:off + 8 * N].view(torch.int32)
                lost += int((flags & 2).ne(0).sum().item())
                if i not in first:
                    first[i] = (costs.clone(), grads.clone())
                elif not (torch.equal(costs, first[i][0]) and torch.equal(grads, first[i][1])):
                    mism += 1
                    if mism <= 6:
                        dc = (costs - first[i][0]).abs()
                        dg = (grads - first[i][1]).abs().reshape(N, -1).max(dim=1).values
                        print(f"  mismatch: case {i} (N={N}, T={T}, U={U}); flags {flags.tolist()}; utterances whose costs differ "
                              f"{dc.ne(0).nonzero().flatten().tolist()} (max {float(dc.max()):.3e}); whose gradients differ "
                              f"{dg.ne(0).nonzero().flatten().tolist()} (max {float(dg.max()):.3e}; NaN: {bool(torch.isnan(grads).any())})", flush=True)
    print(f"wd soak: {launches} launches in {time.time() - t0:.0f} s (seed {seed}); results differing from the first launch: {mism}; "
          f"sweeps redone after a lost hand-over: {lost}; kernel of the last launch: {__import__('warp_rnnt_amd').last_lattice_kernel()}", flush=True)
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
