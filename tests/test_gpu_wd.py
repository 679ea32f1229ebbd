"""The distributed log-domain lattice kernel (csrc/lattice_wd.hip: one workgroup per 64-column block, boundary
columns through L2 rings) and its single-workgroup form (k_lattice_wl: boundary columns through LDS) against the older
single-workgroup kernel (csrc/lattice_ws.hip) and the oracle.  The steady-state blocks of the first two are hand-written
assembly since round 5 (csrc/lattice_step.h), the predicated blocks and all of lattice_ws.hip are C++ with the same
operations in the same order: these tests are what holds the two implementations of the step to the same bits.

Both kernels call the same step function (csrc/lattice_step.h), so they must agree BIT FOR BIT on costs and gradients
whatever the shape, the batch size, the layout and the timing of the hand-overs -- that is what makes the kernel choice a
pure speed knob (one arithmetic: the reference's, core_gather.cu:22-35,106-126)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
from helpers import make_case, np_log_softmax32
from warp_rnnt_amd import debug, ops

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DEV = torch.device("cuda:0")


def _pairs(logits, labels, blank=0):
    lp = np_log_softmax32(logits)
    return oracle.gather_f32(lp, labels, blank)


def _run(lp2, xn, yn, kernel, lam=0.0):
    with debug.lattice_kernel(kernel):
        costs, grads = ops.loss(lp2, None, xn, yn, ops.IN_LOG_PROBS_GATHERED, ops.GRADS_GATHERED, fastemit_lambda=lam)
        torch.cuda.synchronize()
        return costs, grads


# (N, T, U, ragged): one, two and many column blocks; widths one short of / one beyond a block; lattices shorter than a
# block is wide; batches that put several workgroups on every CU
SHAPES = [(3, 37, 70, True), (2, 5, 130, False), (4, 200, 65, True), (2, 150, 64, False), (16, 150, 40, False),
          (5, 333, 129, True), (4, 600, 300, True), (2, 64, 512, False), (70, 90, 200, True), (3, 1, 100, True),
          (2, 700, 257, False), (300, 40, 130, True),
          # one column block per sweep: the plain launch (no rings, no queue, no redo kernel behind)
          (300, 40, 33, True), (3, 5, 20, True), (2, 1, 30, False), (130, 200, 64, True), (3, 100, 1, False),
          # long sweeps: lone, two and five column blocks, ragged and not, U - 1 on a block edge (also run on the kernel's
          # second instantiation, blocks of 16 diagonals, by test_blocks_of_sixteen_diagonals_same_bits below)
          (3, 1030, 40, True), (2, 1100, 70, True), (2, 1200, 300, False), (3, 1024, 129, True), (2, 1500, 17, False)]


@pytest.mark.parametrize("N,T,U,ragged", SHAPES)
def test_same_bits_as_the_single_workgroup_kernel(N, T, U, ragged):
    logits, labels, xn, yn = make_case(1000 + N + T + U, N, T, U, 6, ragged=ragged)
    if ragged and N > 2:
        yn[1] = 0                       # an utterance without labels (single-column scan)
        xn[2] = max(1, T // 3)
    lp2 = torch.tensor(_pairs(logits, labels), device=DEV)
    txn, tyn = torch.tensor(xn, device=DEV), torch.tensor(yn, device=DEV)
    c_ws, g_ws = _run(lp2, txn, tyn, "ws", lam=0.01)
    c_wd, g_wd = _run(lp2, txn, tyn, "wd", lam=0.01)
    assert torch.equal(c_ws, c_wd)
    assert torch.equal(g_ws, g_wd)
    # its single-workgroup form (k_lattice_wl: three waves per column block, boundary columns through LDS; "wl" lets it
    # take every lattice its LDS holds: U <= 320, 148 KiB -- beyond that the call falls through to lattice_ws.hip)
    c_wl, g_wl = _run(lp2, txn, tyn, "wl", lam=0.01)
    if 64 < U <= 320:
        assert debug.last_lattice_kernel() == "lattice_wl"
    assert torch.equal(c_ws, c_wl)
    assert torch.equal(g_ws, g_wl)
    if N * T * U <= 400_000:
        o = oracle.rnnt_loss_f32(lp2.cpu().numpy(), None, xn, yn, blank=-1, fastemit_lambda=0.01)
        # (two fp32 implementations of one operation order: 1e-4 up to T+U ~ 200, the rounding of |alpha| beyond)
        np.testing.assert_allclose(c_wd.cpu().numpy(), o["costs"], rtol=1e-5)
        # (... 3e-3 from a thousand frames on: one rounding of |alpha| ~ 4e3 on best-path cells, tests/test_gpu_baseline_sizes.py)
        np.testing.assert_allclose(g_wd.cpu().numpy(), o["grads"], atol=1e-4 if T + U <= 250 else 3e-4 if T + U <= 900 else 3e-3)


def test_default_route_picks_either_kernel_by_batch_and_the_bits_do_not_change():
    """csrc/lattice.hip: launch_lattice takes the distributed kernel while 2N*ceil(U/64) <= 2 x the compute units and the
    single-workgroup one beyond.  One utterance, computed in a batch on either side of that line: the same bits."""
    N, T, U = 200, 700, 130                # 2 * 200 * 3 = 1200 column blocks: one workgroup per sweep
    logits, labels, xn, yn = make_case(31, 4, T, U, 5, ragged=True)
    lp2_small = torch.tensor(_pairs(logits, labels), device=DEV)
    lp2 = lp2_small.repeat(N // 4, 1, 1, 1).contiguous()
    txn = torch.tensor(np.tile(xn, N // 4), device=DEV)
    tyn = torch.tensor(np.tile(yn, N // 4), device=DEV)
    c_big, g_big = ops.loss(lp2, None, txn, tyn, ops.IN_LOG_PROBS_GATHERED, ops.GRADS_GATHERED)
    c_small, g_small = ops.loss(lp2_small, None, txn[:4].contiguous(), tyn[:4].contiguous(), ops.IN_LOG_PROBS_GATHERED,
                                ops.GRADS_GATHERED)                         # 24 column blocks: one workgroup each
    torch.cuda.synchronize()
    assert torch.equal(c_big[:4], c_small) and torch.equal(g_big[:4], g_small)
    assert torch.equal(c_big[-4:], c_small) and torch.equal(g_big[-4:], g_small)


def test_wider_than_one_workgroup_can_sweep():
    """U > 512: the single-workgroup kernel cannot take it (lattice.hip's striped kernel does); the distributed one
    simply has more column blocks.  Against the oracle."""
    N, T, U = 2, 60, 700
    logits, labels, xn, yn = make_case(77, N, T, U, 5, ragged=True)
    lp2 = torch.tensor(_pairs(logits, labels), device=DEV)
    txn, tyn = torch.tensor(xn, device=DEV), torch.tensor(yn, device=DEV)
    c_wd, g_wd = _run(lp2, txn, tyn, "wd")
    c_st, g_st = _run(lp2, txn, tyn, "ws")      # (falls through to the striped kernel at this width)
    o = oracle.rnnt_loss_f32(lp2.cpu().numpy(), None, xn, yn, blank=-1)
    np.testing.assert_allclose(c_wd.cpu().numpy(), o["costs"], rtol=1e-5)
    np.testing.assert_allclose(g_wd.cpu().numpy(), o["grads"], atol=3e-4)
    np.testing.assert_allclose(c_st.cpu().numpy(), o["costs"], rtol=1e-5)
    np.testing.assert_allclose(g_st.cpu().numpy(), o["grads"], atol=3e-4)


def test_results_do_not_depend_on_timing_or_workspace_contents():
    """Replays on a workspace that is scribbled over between calls (stale rings, stale tags) and under a competing
    stream of copies: the same bits every time."""
    N, T, U = 6, 400, 200
    logits, labels, xn, yn = make_case(5, N, T, U, 4, ragged=True)
    lp2 = torch.tensor(_pairs(logits, labels), device=DEV)
    txn, tyn = torch.tensor(xn, device=DEV), torch.tensor(yn, device=DEV)
    c0, g0 = _run(lp2, txn, tyn, "ws")
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    for it in range(6):
        with torch.cuda.stream(side):
            for _ in range(8):
                junk.copy_(junk.flip(0))
        # churn the allocator so that the next workspace reuses memory with other contents
        scrib = torch.randint(0, 255, (ops._lib.load().rnnt_amd_workspace_size(N, T, U),), dtype=torch.uint8, device=DEV)
        del scrib
        c, g = _run(lp2, txn, tyn, "wd")
        assert torch.equal(c, c0) and torch.equal(g, g0), it
    torch.cuda.synchronize()


@pytest.mark.parametrize("N,T,U", [(4, 600, 300), (3, 1100, 150), (5, 333, 129), (6, 200, 64), (3, 1500, 70)])
@pytest.mark.parametrize("poison", [float("nan"), float("inf"), -float("inf"), 3.0e38])
def test_cells_outside_an_utterance_never_reach_a_result(N, T, U, poison):
    """Ragged batch with every pair OUTSIDE an utterance's own (T_n, U_n) lattice poisoned.  The hand-written blocks run
    unpredicated: lanes compute on before their first frame (head blocks), behind their last (since round 6 the blocks
    lanes finish in run the steady-state code too, csrc/lattice_wd_body.h: RNNT_WD_FAST_TAIL) and in columns beyond U_n --
    on whatever the planes hold there.  None of it may reach a cost, a gradient, or the forward/backward check: the same
    bits as the single-workgroup kernel on the clean batch, from every kernel and both block sizes' default route."""
    logits, labels, xn, yn = make_case(77 + N + T + U, N, T, U, 5, ragged=True)
    xn[0], yn[0] = T, U - 1                                     # one full-size utterance keeps the launch bounds
    clean = _pairs(logits, labels)
    dirty = clean.copy()
    for n in range(N):
        dirty[n, xn[n]:, :, :] = poison
        dirty[n, :, yn[n] + 1:, :] = poison
        dirty[n, :, yn[n], 1] = poison                         # (the label slot of the last column: no label there)
    txn, tyn = torch.tensor(xn, device=DEV), torch.tensor(yn, device=DEV)
    c0, g0 = _run(torch.tensor(clean, device=DEV), txn, tyn, "ws", lam=0.01)
    lp2 = torch.tensor(dirty, device=DEV)
    for kernel in ("wd", "wl", "ws"):
        c, g = _run(lp2, txn, tyn, kernel, lam=0.01)
        assert torch.isfinite(c).all(), (kernel, c)
        assert torch.equal(c, c0), kernel
        # gradients: identical on every cell -- the utterances' own lattices, and the zeros outside them
        assert torch.equal(g, g0), kernel


def test_lost_hand_over_is_redone_by_the_single_workgroup_kernel():
    """`short_spin` build: a hand-over wait gives up at the first poll, so every column block that catches up with its
    neighbour runs on stale boundary values, flags its sweep, and the kernel launched behind redoes it.  Same bits."""
    from warp_rnnt_amd import _build
    lib = _build.build(variant="short_spin")
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle
from warp_rnnt_amd import debug, ops
from helpers import make_case, np_log_softmax32
dev = torch.device("cuda:0")
L = ops._lib.load()
bad = 0
for (N, T, U) in [(4, 500, 300), (16, 300, 130), (40, 200, 200)]:
    logits, labels, xn, yn = make_case(N + T, N, T, U, 5, ragged=True)
    lp2 = torch.tensor(oracle.gather_f32(np_log_softmax32(logits), labels, 0), device=dev)
    txn, tyn = torch.tensor(xn, device=dev), torch.tensor(yn, device=dev)
    res = {}
    for k in ("ws", "wd"):
        debug.set_lattice_kernel(k)
        ws = torch.empty((L.rnnt_amd_workspace_size(N, T, U),), dtype=torch.uint8, device=dev)
        costs = torch.empty((N,), device=dev); grads = torch.empty((N, T, U, 2), device=dev)
        st = L.rnnt_amd_loss(torch.cuda.current_stream().cuda_stream, ws.data_ptr(), 1, lp2.data_ptr(), None,
                             txn.data_ptr(), tyn.data_ptr(), costs.data_ptr(), grads.data_ptr(), 0, N, T, U, 2, 0, 0.0)
        assert st == 0
        torch.cuda.synchronize()
        off = L.rnnt_amd_debug_redo_offset(N, T, U)
        flags = ws[off:off + 8 * N].view(torch.int32).clone()
        res[k] = (costs, grads, flags)
    assert torch.equal(res["ws"][0], res["wd"][0]) and torch.equal(res["ws"][1], res["wd"][1]), (N, T, U)
    bad += int((res["wd"][2] & 2).ne(0).sum().item())
assert bad > 0, "the short-spin build never lost a hand-over: the redo path was not exercised"
print("WD_SHORT_SPIN_OK", bad)
''' % (os.path.dirname(HERE), HERE)
    env = dict(os.environ, WARP_RNNT_AMD_LIB=lib)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         timeout=900)
    text = out.stdout.decode()
    assert out.returncode == 0 and "WD_SHORT_SPIN_OK" in text, text[-4000:]


@pytest.mark.parametrize("N,T,U,V", [(5, 220, 150, 7), (6, 120, 50, 7)])      # (the second: one column block, plain launch)
def test_compact_layout_same_bits(N, T, U, V):
    """The native compact entry (64-bit cell offsets): per-utterance planes, rings sized by the launch bounds."""
    import warp_rnnt
    logits, labels, xn, yn = make_case(9, N, T, U, V, ragged=True)
    lp = torch.tensor(np_log_softmax32(logits), device=DEV)
    tl, txn, tyn = (torch.tensor(a, device=DEV) for a in (labels, xn, yn))
    # pack: (sum_n T_n*U_n, V) rows, labels concatenated -- what rnnt_loss(compact=True) takes (__init__.py:109-116)
    rows = torch.cat([lp[n, :xn[n], :yn[n] + 1].reshape(-1, V) for n in range(N)]).contiguous()
    labs = torch.cat([tl[n, :yn[n]] for n in range(N)]).contiguous()
    out = {}
    for k in ("ws", "wd", "wl"):
        with debug.lattice_kernel(k):
            x = rows.clone().requires_grad_(True)
            loss = warp_rnnt.rnnt_loss(x, labs, txn, tyn, compact=True, reduction="sum")
            loss.backward()
            torch.cuda.synchronize()
            out[k] = (loss.detach().clone(), x.grad.clone())
    assert torch.equal(out["ws"][0], out["wd"][0]) and torch.equal(out["ws"][1], out["wd"][1])
    assert torch.equal(out["ws"][0], out["wl"][0]) and torch.equal(out["ws"][1], out["wl"][1])


@pytest.mark.parametrize("from_t", ["1", "1000000"], ids=["sixteen_everywhere", "eight_everywhere"])
def test_both_block_sizes_same_bits(from_t):
    """csrc/lattice_wd_body.h is compiled twice: blocks of 8 diagonals and blocks of 16, the second the choice from launch
    bound T >= 1024 on since round 6 (RNNT_WD_K16_FROM_T overrides; csrc/lattice_wd.hip has the history).  Each has to
    give the bits of the single-workgroup kernel at every length: short and long shapes of SHAPES in a subprocess with
    the knob forcing one size everywhere."""
    code = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import oracle
from helpers import make_case, np_log_softmax32
from warp_rnnt_amd import debug, ops
dev = torch.device("cuda:0")
L = ops._lib.load()
for (N, T, U, ragged) in [(3, 37, 70, True), (4, 200, 65, True), (16, 150, 40, False), (5, 333, 129, True), (4, 600, 300, True), (2, 5, 130, False),
                          (3, 1030, 40, True), (2, 1100, 70, True), (2, 1200, 300, False), (3, 1024, 129, True), (2, 1500, 17, False)]:
    logits, labels, xn, yn = make_case(1000 + N + T + U, N, T, U, 6, ragged=ragged)
    lp2 = torch.tensor(oracle.gather_f32(np_log_softmax32(logits), labels, 0), device=dev)
    txn, tyn = torch.tensor(xn, device=dev), torch.tensor(yn, device=dev)
    res = {}
    for k in ("ws", "wd"):
        debug.set_lattice_kernel(k)
        ws = torch.empty((L.rnnt_amd_workspace_size(N, T, U),), dtype=torch.uint8, device=dev)
        costs = torch.empty((N,), device=dev); grads = torch.empty((N, T, U, 2), device=dev)
        st = L.rnnt_amd_loss(torch.cuda.current_stream().cuda_stream, ws.data_ptr(), 1, lp2.data_ptr(), None,
                             txn.data_ptr(), tyn.data_ptr(), costs.data_ptr(), grads.data_ptr(), 0, N, T, U, 2, 0, 0.01)
        assert st == 0
        torch.cuda.synchronize()
        res[k] = (costs, grads)
    assert torch.equal(res["ws"][0], res["wd"][0]) and torch.equal(res["ws"][1], res["wd"][1]), (N, T, U)
print("BLOCK_SIZE_SAME_BITS_OK")
''' % (os.path.dirname(HERE), HERE)
    env = dict(os.environ, RNNT_WD_K16_FROM_T=from_t)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert out.returncode == 0 and b"BLOCK_SIZE_SAME_BITS_OK" in out.stdout, out.stdout.decode()[-3000:]
