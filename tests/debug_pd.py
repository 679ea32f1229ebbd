"""Dev helper (GPU): dump the alpha/beta planes the lattice kernel left in the workspace and compare them with
the fp64 oracle cell by cell.  python tests/debug_pd.py N T U ragged"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
from warp_rnnt_amd import ops, _lib
from oracle import transduce_np

def main(N, T, U, ragged, seed=0):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    lp = torch.log_softmax(torch.randn((N, T, U, 7), device="cuda", generator=g), -1)
    ys = torch.randint(1, 7, (N, max(U - 1, 1)), dtype=torch.int32, device="cuda", generator=g)[:, :U - 1].contiguous()
    rng = np.random.RandomState(seed)
    xn = rng.randint(max(T // 2, 1), T + 1, N) if ragged else np.full(N, T)
    yn = rng.randint(U // 2, U, N) if ragged else np.full(N, U - 1)
    xn[0], yn[0] = T, U - 1
    txn = torch.tensor(xn, dtype=torch.int32, device="cuda"); tyn = torch.tensor(yn, dtype=torch.int32, device="cuda")
    L = _lib.load()
    wsb = L.rnnt_amd_workspace_size(N, T, U)
    ws = torch.zeros((wsb,), dtype=torch.uint8, device="cuda")
    costs = torch.empty((N,), device="cuda"); grads = torch.empty((N, T, U, 2), device="cuda")
    st = L.rnnt_amd_loss(torch.cuda.current_stream().cuda_stream, ws.data_ptr(), 0, lp.data_ptr(), ys.data_ptr(),
                         txn.data_ptr(), tyn.data_ptr(), costs.data_ptr(), grads.data_ptr(), 0, N, T, U, 7, 0, 0.0)
    torch.cuda.synchronize()
    assert st == 0
    cells = N * T * U
    al = ws[:cells * 4].view(torch.float32).cpu().numpy().reshape(N, T, U)
    off = (cells * 4 + 255) // 256 * 256
    be = ws[off:off + cells * 4].view(torch.float32).cpu().numpy().reshape(N, T, U)
    lpn = lp.cpu().numpy().astype(np.float64); ysn = ys.cpu().numpy()
    print("costs", costs.cpu().numpy())
    for n in range(N):
        t_, u_ = int(xn[n]), int(yn[n]) + 1
        c, gg, a64, b64 = transduce_np.transduce(lpn[n, :t_, :u_], ysn[n, :u_ - 1], 0, 0.0, True)
        # un-skew
        A = np.zeros((t_, u_)); B = np.zeros((t_, u_))
        for t in range(t_):
            for u in range(u_):
                r = (t + u) % T
                A[t, u] = al[n].reshape(-1)[r * U + u]; B[t, u] = be[n].reshape(-1)[r * U + u]
        ea, eb = np.abs(A - a64), np.abs(B - b64)
        print(f"n={n} Tn={t_} Un={u_} cost64={c:.4f} alpha maxerr {np.nanmax(ea):.3e} beta maxerr {np.nanmax(eb):.3e}")
        for name, e, M, R in (("alpha", ea, A, a64), ("beta", eb, B, b64)):
            bad = np.argwhere(~(e < 1e-2))
            if len(bad):
                print("  ", name, "bad cells", len(bad), "first", bad[:6].tolist(), "got", [float(M[tuple(b)]) for b in bad[:4]],
                      "want", [float(R[tuple(b)]) for b in bad[:4]])
                print("   by column:", sorted(set(bad[:, 1].tolist()))[:20], " by diagonal:", sorted(set((bad[:, 0] + bad[:, 1]).tolist()))[:20])

if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]), int(a[1]), int(a[2]), a[3] == "1")
