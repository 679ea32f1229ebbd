"""The loss under HIP graph capture (torch.cuda.CUDAGraph): a captured call is enqueue-only, so it must replay,
and every replay must see its own inputs.

The probability-domain lattice hands boundary columns between workgroups through tagged granules in the workspace;
the tag carries a launch epoch. Kernel arguments are frozen at capture time, so the epoch also needs a part that
lives on the device (lattice_wd.hip: k_prepare) -- with a frozen epoch a replay accepts the granules the previous
replay left behind. The shapes below take that kernel (T >= 640, T >= 2U) with two and three column blocks.

No counterpart in the reference (its launches are capturable as well; it has no cross-workgroup hand-over)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs(torch, N, T, U, V, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    logits = torch.randn(N, T, U, V, generator=g) * 2.0
    labels = torch.randint(1, V, (N, U - 1), generator=g, dtype=torch.int32)
    xn = torch.randint(T // 2, T + 1, (N,), generator=g, dtype=torch.int32)
    yn = torch.randint(U // 2, U, (N,), generator=g, dtype=torch.int32)
    xn[0], yn[0] = T, U - 1
    return torch.log_softmax(logits, -1), labels, xn, yn


@pytest.mark.parametrize("shape", [(2, 700, 70, 11), (1, 900, 150, 7)])
def test_native_op_replays_with_fresh_inputs(shape):
    import torch
    from warp_rnnt import _C
    N, T, U, V = shape
    dev = torch.device("cuda:0")
    sets = [tuple(t.to(dev) for t in _inputs(torch, N, T, U, V, seed)) for seed in (1, 2, 3)]
    eager = [_C.rnnt_loss_gather(*s, 0, 0.0) for s in sets]
    torch.cuda.synchronize()

    static = [t.clone() for t in sets[0]]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            _C.rnnt_loss_gather(*static, 0, 0.0)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        costs, grads = _C.rnnt_loss_gather(*static, 0, 0.0)
    for rep in range(12):
        k = rep % len(sets)
        for dst, src in zip(static, sets[k]):
            dst.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(costs.cpu().numpy(), eager[k][0].cpu().numpy(),
                                      err_msg=f"replay {rep} (input set {k})")
        assert torch.equal(grads, eager[k][1]), f"replay {rep} (input set {k}): gradients differ from eager"


def test_training_step_replays():
    """forward + backward of warp_rnnt.rnnt_loss(gather=True) from logits, captured as one graph."""
    import torch
    import warp_rnnt
    N, T, U, V = 2, 700, 70, 11
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    batches = [torch.randn(N, T, U, V, generator=g).to(dev) for _ in range(3)]
    _, labels, xn, yn = (t.to(dev) for t in _inputs(torch, N, T, U, V, 7))

    def step(logits):
        lp = torch.log_softmax(logits, -1)
        loss = warp_rnnt.rnnt_loss(lp, labels, xn, yn, reduction="mean", gather=True)
        loss.backward()
        return loss

    want = []
    for b in batches:
        x = b.clone().requires_grad_(True)
        want.append((step(x).detach().clone(), x.grad.clone()))

    x = batches[0].clone().requires_grad_(True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            x.grad = None
            step(x)
    torch.cuda.current_stream().wait_stream(side)
    x.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss = step(x)
    for rep in range(9):
        k = rep % len(batches)
        with torch.no_grad():
            x.copy_(batches[k])
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(loss, want[k][0]), f"replay {rep}: loss {loss.item()} vs eager {want[k][0].item()}"
        assert torch.equal(x.grad, want[k][1]), f"replay {rep}: d/d logits differ from eager"


def test_replays_do_not_depend_on_what_the_workspace_holds():
    """The workspace is caller scratch with unspecified contents (include/warp_rnnt_amd.h).  Under capture it is a
    tensor freed back to the graph's pool, so between replays anything may land in it: the previous replay's
    hand-over granules (the realistic case), constant bytes, random bytes.  k_prepare clears the rings and takes the
    launch epoch from the library's own device-side counter, so every replay must reproduce the eager result
    (with the counter in the workspace, constant data over that word froze the epoch)."""
    import torch
    from warp_rnnt_amd import _lib, debug, ops
    L = _lib.load()
    N, T, U, V = 2, 700, 150, 11                       # three column blocks per sweep: two hand-over rings each
    dev = torch.device("cuda:0")
    sets = [tuple(t.to(dev) for t in _inputs(torch, N, T, U, V, seed)) for seed in (11, 12, 13)]
    ws = torch.empty((L.rnnt_amd_workspace_size(N, T, U),), dtype=torch.uint8, device=dev)
    costs = torch.empty((N,), device=dev)
    grads = torch.empty((N, T, U, 2), device=dev)
    static = [t.clone() for t in sets[0]]

    def call():
        st = L.rnnt_amd_loss(torch.cuda.current_stream().cuda_stream, ws.data_ptr(), ops.IN_LOG_PROBS_DENSE,
                             static[0].data_ptr(), static[1].data_ptr(), static[2].data_ptr(), static[3].data_ptr(),
                             costs.data_ptr(), grads.data_ptr(), ops.GRADS_GATHERED, N, T, U, V, 0, 0.0)
        assert st == 0

    # (k_lattice_wd at this shape by itself; pinned so that the test keeps testing the rings whatever the routing rule)
    with debug.lattice_kernel("wd"):
        eager = []
        for s in sets:
            for dst, src in zip(static, s):
                dst.copy_(src)
            ws.random_(0, 256)
            call()
            torch.cuda.synchronize()
            eager.append((costs.clone(), grads.clone()))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            call()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            call()
    for rep in range(15):
        k = rep % len(sets)
        for dst, src in zip(static, sets[k]):
            dst.copy_(src)
        if rep % 3 == 1:
            ws.fill_(0x5A)
        elif rep % 3 == 2:
            ws.random_(0, 256)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(costs, eager[k][0]), f"replay {rep}: costs {costs.tolist()} vs eager {eager[k][0].tolist()}"
        assert torch.equal(grads, eager[k][1]), f"replay {rep}: gradients differ from eager"


def test_compact_with_launch_bounds_is_captured_and_replays():
    """rnnt_loss(compact=True, max_frames=, max_labels=): the caller supplies the launch bounds, nothing is read back
    from the device (rnnt_amd_loss_compact_bounded), so forward + backward capture into one graph.  Replayed with other
    data and with the utterances in another order (same totals, other offsets and lengths); a batch that does not fit
    the bounds comes back as NaN costs and zero gradients -- on replay too."""
    import torch
    import warp_rnnt
    dev = torch.device("cuda:0")
    V, Tb, Ub = 9, 700, 130                      # bounds: frames, labels (3 column blocks: the distributed kernel)
    shapes = [(700, 130), (420, 64), (650, 100)]

    def batch(order, seed):
        g = torch.Generator(device="cpu").manual_seed(seed)
        rows, labs, xn, yn = [], [], [], []
        for i in order:
            t, u = shapes[i]
            rows.append(torch.log_softmax(torch.randn(t * (u + 1), V, generator=g) * 2.0, -1))
            labs.append(torch.randint(1, V, (u,), generator=g, dtype=torch.int32))
            xn.append(t); yn.append(u)
        return (torch.cat(rows).to(dev), torch.cat(labs).to(dev), torch.tensor(xn, dtype=torch.int32, device=dev),
                torch.tensor(yn, dtype=torch.int32, device=dev))

    sets = [batch((0, 1, 2), 1), batch((2, 0, 1), 2), batch((1, 2, 0), 3)]
    w = torch.tensor([0.5, 1.0, 1.5], device=dev)

    def run(xs, ys, xn, yn, **kw):
        x = xs.detach().requires_grad_(True)
        costs = warp_rnnt.rnnt_loss(x, ys, xn, yn, compact=True, fastemit_lambda=0.01, **kw)
        (costs * w).sum().backward()
        return costs.detach(), x.grad

    eager = []
    for s_ in sets:
        c_sync, g_sync = run(*s_)                                          # the path with its one host read-back
        c_b, g_b = run(*s_, max_frames=Tb, max_labels=Ub)                  # bounds: none
        torch.cuda.synchronize()
        assert torch.equal(c_sync, c_b) and torch.equal(g_sync, g_b)
        eager.append((c_b.clone(), g_b.clone()))

    static = [t.clone() for t in sets[0]]
    x_static = static[0].requires_grad_(True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            x_static.grad = None
            c = warp_rnnt.rnnt_loss(x_static, static[1], static[2], static[3], compact=True, fastemit_lambda=0.01,
                                    max_frames=Tb, max_labels=Ub)
            (c * w).sum().backward()
    torch.cuda.current_stream().wait_stream(side)
    x_static.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        costs = warp_rnnt.rnnt_loss(x_static, static[1], static[2], static[3], compact=True, fastemit_lambda=0.01,
                                    max_frames=Tb, max_labels=Ub)
        (costs * w).sum().backward()
    for rep in range(9):
        k = rep % len(sets)
        with torch.no_grad():
            for dst, src in zip(static, sets[k]):
                dst.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(costs, eager[k][0]), f"replay {rep}"
        assert torch.equal(x_static.grad, eager[k][1]), f"replay {rep}: gradients differ from eager"
    # a length beyond the bound: the replayed graph refuses the batch on the device
    with torch.no_grad():
        static[2].copy_(torch.tensor([701, 420, 649], dtype=torch.int32))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.isnan(costs).all() and not x_static.grad.any()
    # eager, same story, and totals that do not match the tensors
    xs, ys, xn, yn = sets[0]
    c, g = run(xs, ys, xn, yn, max_frames=600, max_labels=Ub)
    assert torch.isnan(c).all() and not g.any()
    c, g = run(xs, ys[:-1].contiguous(), xn, yn, max_frames=Tb, max_labels=Ub)
    assert torch.isnan(c).all() and not g.any()
    # no rows at all against lengths that ask for some (STU == 0, N > 0): refused like any other mismatch -- NaN costs,
    # not whatever the costs buffer held (api.hip: the bounded entry fills them before it returns)
    from warp_rnnt_amd import ops
    empty = torch.empty((0, V), device=dev)
    c, g, _ = ops.loss_compact(empty, ys, xn, yn, max_frames=Tb, max_labels=Ub)
    torch.cuda.synchronize()
    assert c.shape == (3,) and torch.isnan(c).all()
    assert g is None or g.numel() == 0
