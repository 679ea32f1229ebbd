#!/usr/bin/env python
"""First-contact check of the probability-domain lattice kernel on a GPU: a few shapes against the
log-domain build of the same library (WARP_RNNT_AMD_LIB), printing alpha/beta/ll agreement.  Dev tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from warp_rnnt_amd import ops

def run(N, T, U, ragged, seed=0, scale=1.0):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    lp = torch.log_softmax(torch.randn((N, T, U, 7), device="cuda", generator=g) * scale, -1)
    ys = torch.randint(1, 7, (N, max(U - 1, 1)), dtype=torch.int32, device="cuda", generator=g)[:, :U - 1].contiguous()
    rng = np.random.RandomState(seed)
    xn = rng.randint(max(T // 2, 1), T + 1, N) if ragged else np.full(N, T)
    yn = rng.randint(U // 2, U, N) if ragged else np.full(N, U - 1)
    xn[0], yn[0] = T, U - 1
    txn = torch.tensor(xn, dtype=torch.int32, device="cuda"); tyn = torch.tensor(yn, dtype=torch.int32, device="cuda")
    c, gr = ops.loss(lp, ys, txn, tyn, ops.IN_LOG_PROBS_DENSE, ops.GRADS_GATHERED, 0, 0.0)
    torch.cuda.synchronize()
    return c.double().cpu().numpy(), gr.double().cpu().numpy()

if __name__ == "__main__":
    shapes = [(2, 5, 4, False), (3, 40, 12, True), (2, 150, 40, False), (2, 33, 130, True), (2, 300, 200, True),
              (1, 20, 320, False), (2, 700, 300, True), (1, 1, 9, False), (2, 9, 1, False), (2, 64, 64, False),
              (2, 65, 65, True), (3, 8, 129, True)]
    out = {}
    for s in shapes:
        t0 = time.time()
        out[s] = run(*s)
        print(s, "cost", out[s][0][:3], "finite", np.isfinite(out[s][1]).all(), f"{time.time()-t0:.2f}s", flush=True)
    np.savez(sys.argv[1], **{str(k): np.concatenate([v[0].ravel(), v[1].ravel()]) for k, v in out.items()})
