import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from warp_rnnt_amd import ops, _lib
from oracle import transduce_np
def run(N, T, U, ragged, seed=0):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    lp = torch.log_softmax(torch.randn((N, T, U, 7), device="cuda", generator=g), -1)
    ys = torch.randint(1, 7, (N, max(U - 1, 1)), dtype=torch.int32, device="cuda", generator=g)[:, :U - 1].contiguous()
    rng = np.random.RandomState(seed)
    xn = rng.randint(max(T // 2, 1), T + 1, N) if ragged else np.full(N, T)
    yn = rng.randint(U // 2, U, N) if ragged else np.full(N, U - 1)
    xn[0], yn[0] = T, U - 1
    txn = torch.tensor(xn, dtype=torch.int32, device="cuda"); tyn = torch.tensor(yn, dtype=torch.int32, device="cuda")
    L = _lib.load()
    ws = torch.zeros((L.rnnt_amd_workspace_size(N, T, U),), dtype=torch.uint8, device="cuda")
    costs = torch.empty((N,), device="cuda"); grads = torch.empty((N, T, U, 2), device="cuda")
    st = L.rnnt_amd_loss(torch.cuda.current_stream().cuda_stream, ws.data_ptr(), 0, lp.data_ptr(), ys.data_ptr(),
                         txn.data_ptr(), tyn.data_ptr(), costs.data_ptr(), grads.data_ptr(), 0, N, T, U, 7, 0, 0.0)
    torch.cuda.synchronize(); assert st == 0
    cells = N * T * U
    al = ws[:cells * 4].view(torch.float32).cpu().numpy().reshape(N, T * U)
    off = (cells * 4 + 255) // 256 * 256
    be = ws[off:off + cells * 4].view(torch.float32).cpu().numpy().reshape(N, T * U)
    lpn = lp.cpu().numpy().astype(np.float64); ysn = ys.cpu().numpy()
    res = []
    for n in range(N):
        t_, u_ = int(xn[n]), int(yn[n]) + 1
        c, gg, a64, b64 = transduce_np.transduce(lpn[n, :t_, :u_], ysn[n, :u_ - 1], 0, 0.0, True)
        tt, uu = np.meshgrid(np.arange(t_), np.arange(u_), indexing="ij")
        idx = ((tt + uu) % T) * U + uu
        A, B = al[n][idx], be[n][idx]
        ba = np.argwhere(~(np.abs(A - a64) < 1e-2)); bb = np.argwhere(~(np.abs(B - b64) < 1e-2))
        nd = t_ + u_ - 1
        blocks_a = sorted(set(((ba[:, 0] + ba[:, 1]) // 8).tolist())); blocks_b = sorted(set(((nd - 1 - (bb[:, 0] + bb[:, 1])) // 8).tolist())) if len(bb) else []
        cols_a = sorted(set((ba[:, 1] // 64).tolist()))
        res.append(f"n{n}(T{t_},U{u_}) A-bad-blocks{blocks_a} colblk{cols_a} B-bad-sweepblocks{blocks_b}")
    print((N, T, U, ragged), " | ".join(res), flush=True)
for s in [(1,5,4,0),(1,8,4,0),(1,9,4,0),(1,16,4,0),(1,17,12,0),(1,40,12,0),(1,41,12,0),(1,64,12,0),(1,150,12,0),(1,150,40,0),(1,40,40,0),(1,24,40,0),(1,20,320,0),(1,33,130,0),(2,33,130,1),(1,8,129,0),(1,100,129,0),(3,40,12,1),(2,700,300,1),(2,300,257,1),(4,1500,300,0)]:
    run(*s)
