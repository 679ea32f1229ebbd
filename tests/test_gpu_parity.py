"""GPU parity tests: the HIP path (through the C ABI and through the drop-in Python API)
against the CPU oracle, the reference's golden vectors and size-independent invariants.

Tolerances (BASELINE.json north_star: "loss and grads matching the reference within 1e-4 fp32"):
  costs: rtol 1e-5 (fp32 ulp at |cost|~6e3 is 5e-4 absolute, i.e. ~8e-8 relative)
  grads: atol 1e-4 against the fp32 oracle at sizes where fp32 itself is that good
         (T+U <~ 200); at the BASELINE sizes fp32 implementations differ from exact
         arithmetic by ~1e-2, there the invariants + an fp64 comparison
         with a stated looser bound are used.
"""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from oracle import transduce_np
from helpers import make_case, np_log_softmax32, reference_cases, reference_doc

pytestmark = pytest.mark.gpu

GRAD_ATOL = 1e-4
COST_RTOL = 1e-5


def dev():
    return torch.device("cuda:0")


def t32(a):
    return torch.tensor(np.ascontiguousarray(a), device=dev())


def run_native(lp, labels, xn, yn, blank=0, lam=0.0):
    import warp_rnnt._C as core
    ys = t32(labels.astype(np.int32)).reshape(lp.shape[0], lp.shape[2] - 1)
    costs, grads = core.rnnt_loss(t32(lp), ys, t32(xn), t32(yn), blank=blank, fastemit_lambda=lam)
    torch.cuda.synchronize()
    return costs.cpu().numpy(), grads.cpu().numpy()


# ----------------------------------------------------------------------------
# 1. the reference's own golden vectors (test.py:34-188, 214-257), tolerance decimal=6
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("case", reference_cases(), ids=lambda c: c["name"])
def test_reference_golden_native_op(case):
    logits = np.array(case["logits"], dtype=np.float32)
    lp = np_log_softmax32(logits)
    N, T, U, V = lp.shape
    labels = np.array(case["labels"], dtype=np.int32).reshape(N, U - 1)
    xn = np.array(case["xn"], dtype=np.int32)
    yn = np.array(case["yn"], dtype=np.int32)
    blank = case["blank"]
    if case["layout"] == "gathered":
        lp = oracle.gather_f32(lp, labels, blank)
        blank = -1
    costs, grads = run_native(lp, labels, xn, yn, blank=blank)
    np.testing.assert_allclose(costs, np.array(case["costs"]), atol=1.5e-6, rtol=0)
    np.testing.assert_allclose(grads, np.array(case["grads"]), atol=1.5e-6, rtol=0)


# ----------------------------------------------------------------------------
# 2. the reference C ABI (core.h:29-39) called directly with raw device pointers
# ----------------------------------------------------------------------------
def _call_ref_abi(lp, labels, xn, yn, blank, lam):
    import warp_rnnt_amd
    L = warp_rnnt_amd.load()
    N, T, U, V = lp.shape
    d = dev()
    xs = t32(lp)
    ys = t32(labels.astype(np.int32).reshape(N, max(U - 1, 0)))
    txn, tyn = t32(xn), t32(yn)
    grads = torch.zeros_like(xs)                              # contract: caller zeroes grads
    counts = torch.zeros((N, 2 * U), dtype=torch.int32, device=d)
    alphas = torch.empty((N, T, U), dtype=torch.float32, device=d)
    betas = torch.empty((N, T, U), dtype=torch.float32, device=d)
    costs = torch.empty((N,), dtype=torch.float32, device=d)
    stream = torch.cuda.current_stream().cuda_stream
    if blank == -1:
        st = L.run_warp_rnnt_gather(stream, counts.data_ptr(), alphas.data_ptr(), betas.data_ptr(),
                                    xs.data_ptr(), grads.data_ptr(), costs.data_ptr(), txn.data_ptr(),
                                    tyn.data_ptr(), N, T, U, lam)
    else:
        st = L.run_warp_rnnt(stream, counts.data_ptr(), alphas.data_ptr(), betas.data_ptr(),
                             ys.data_ptr(), xs.data_ptr(), grads.data_ptr(), costs.data_ptr(),
                             txn.data_ptr(), tyn.data_ptr(), N, T, U, V, blank, lam)
    torch.cuda.synchronize()
    assert st == 0
    return costs.cpu().numpy(), grads.cpu().numpy()


@pytest.mark.parametrize("case", reference_cases(), ids=lambda c: c["name"])
def test_reference_golden_c_abi(case):
    logits = np.array(case["logits"], dtype=np.float32)
    lp = np_log_softmax32(logits)
    N, T, U, V = lp.shape
    labels = np.array(case["labels"], dtype=np.int32).reshape(N, U - 1)
    xn = np.array(case["xn"], dtype=np.int32)
    yn = np.array(case["yn"], dtype=np.int32)
    blank = case["blank"]
    if case["layout"] == "gathered":
        lp = oracle.gather_f32(lp, labels, blank)
        blank = -1
    costs, grads = _call_ref_abi(lp, labels, xn, yn, blank, 0.0)
    np.testing.assert_allclose(costs, np.array(case["costs"]), atol=1.5e-6, rtol=0)
    np.testing.assert_allclose(grads, np.array(case["grads"]), atol=1.5e-6, rtol=0)


# ----------------------------------------------------------------------------
# 3. seeded cases against the fp32 oracle: every entry point, ragged lengths, FastEmit,
#    blank != 0, U == 1, T == 1, multi-wave (U > 64) and striped (U > 1024) lattices
# ----------------------------------------------------------------------------
SEEDED = [
    # N, T, U, V, ragged, lambda, blank
    (16, 150, 40, 28, False, 0.0, 0),     # BASELINE config 2 shape
    (4, 150, 20, 500, True, 0.0, 0),      # config 3 shape, smaller V
    (3, 61, 25, 11, True, 0.01, 0),
    (3, 33, 17, 6, True, 0.0, 3),
    (2, 70, 1, 4, False, 0.0, 0),
    (2, 1, 9, 4, False, 0.25, 1),
    (2, 3, 70, 5, True, 0.0, 0),          # U > T, two waves
    (2, 90, 200, 4, True, 0.0, 0),        # four waves
    (1, 40, 1100, 3, False, 0.0, 0),      # U > 1024: column stripes
    (5, 7, 5, 3, True, 0.0, 2),           # T < K
]


@pytest.mark.parametrize("N,T,U,V,ragged,lam,blank", SEEDED)
def test_seeded_vs_oracle_all_entry_points(N, T, U, V, ragged, lam, blank):
    logits, labels, xn, yn = make_case(1234 + T + U, N, T, U, V, ragged=ragged, blank=blank)
    lp = np_log_softmax32(logits)
    ref = oracle.rnnt_loss_f32(lp, labels, xn, yn, blank=blank, fastemit_lambda=lam, scan_mode=1)
    assert not ref["mismatch"].any()
    lp2 = oracle.gather_f32(lp, labels, blank)
    ref2 = oracle.rnnt_loss_f32(lp2, labels, xn, yn, blank=-1, fastemit_lambda=lam, scan_mode=1)

    # native op, dense layout and gathered layout
    c, g = run_native(lp, labels, xn, yn, blank=blank, lam=lam)
    np.testing.assert_allclose(c, ref["costs"], rtol=COST_RTOL)
    np.testing.assert_allclose(g, ref["grads"], atol=GRAD_ATOL)
    c2, g2 = run_native(lp2, labels, xn, yn, blank=-1, lam=lam)
    np.testing.assert_allclose(c2, ref2["costs"], rtol=COST_RTOL)
    np.testing.assert_allclose(g2, ref2["grads"], atol=GRAD_ATOL)
    # identical numbers whichever route the pairs took into the lattice kernel
    np.testing.assert_array_equal(c, c2)

    # reference C ABI
    ca, ga = _call_ref_abi(lp, labels, xn, yn, blank, lam)
    np.testing.assert_allclose(ca, ref["costs"], rtol=COST_RTOL)
    np.testing.assert_allclose(ga, ref["grads"], atol=GRAD_ATOL)
    cb, gb = _call_ref_abi(lp2, labels, xn, yn, -1, lam)
    np.testing.assert_allclose(cb, ref2["costs"], rtol=COST_RTOL)
    np.testing.assert_allclose(gb, ref2["grads"], atol=GRAD_ATOL)
    # the reference-named entry points run the log-domain kernels on the caller's layout (lattice.hip), the
    # native op the probability-domain kernel on the diagonal-major workspace: same results to fp32 accuracy,
    # bit-equal within each family
    np.testing.assert_array_equal(ca, cb)
    np.testing.assert_allclose(ca, c, rtol=COST_RTOL)
    np.testing.assert_allclose(ga, g, atol=GRAD_ATOL)
    np.testing.assert_allclose(gb, g2, atol=GRAD_ATOL)


def test_staged_forms_of_the_entry_points_from_a_million_cells_on():
    """From 2^20 lattice cells on (csrc/api.hip: STAGED_FROM_CELLS) the reference-named C entry points stage the pairs
    in the caller's `grads`, sweep on the tuned kernels, park the gradient pairs in alphas / betas and turn the layout
    through LDS tiles (gathered) or expand whole dense rows (dense, which then no longer depends on the caller's
    zero-fill); the native gathered-gradient form takes the tiled turn as well.  All of them against the fp32 oracle,
    with ragged lengths, FastEmit and blank != 0; the dense entry also on a `grads` buffer full of junk."""
    N, T, U, V, lam, blank = 6, 420, 430, 5, 0.01, 1
    assert N * T * U >= 1 << 20
    logits, labels, xn, yn = make_case(99, N, T, U, V, ragged=True, blank=blank)
    yn[1] = 0                                                   # an utterance without labels
    lp = np_log_softmax32(logits)
    ref = oracle.rnnt_loss_f32(lp, labels, xn, yn, blank=blank, fastemit_lambda=lam, scan_mode=1)
    lp2 = oracle.gather_f32(lp, labels, blank)
    ref2 = oracle.rnnt_loss_f32(lp2, labels, xn, yn, blank=-1, fastemit_lambda=lam, scan_mode=1)
    atol = 5e-4                     # (T + U = 850: one rounding of |alpha| ~ 2e3 on best-path cells)
    ca, ga = _call_ref_abi(lp, labels, xn, yn, blank, lam)
    np.testing.assert_allclose(ca, ref["costs"], rtol=COST_RTOL)
    np.testing.assert_allclose(ga, ref["grads"], atol=atol)
    assert not ga[ref["grads"] == 0].any()                     # nothing but the two slots of live cells
    cb, gb = _call_ref_abi(lp2, labels, xn, yn, -1, lam)
    np.testing.assert_allclose(cb, ref2["costs"], rtol=COST_RTOL)
    np.testing.assert_allclose(gb, ref2["grads"], atol=atol)
    np.testing.assert_array_equal(ca, cb)
    c2, g2 = run_native(lp2, labels, xn, yn, blank=-1, lam=lam)
    np.testing.assert_array_equal(c2, cb)
    np.testing.assert_array_equal(g2, gb)                       # the same kernels behind both
    c, g = run_native(lp, labels, xn, yn, blank=blank, lam=lam)
    np.testing.assert_array_equal(g, ga)
    # dense entry on a dirty `grads`: every row is written whole
    import warp_rnnt_amd
    L = warp_rnnt_amd.load()
    d = dev()
    xs, ys, txn, tyn = t32(lp), t32(labels), t32(xn), t32(yn)
    grads = torch.full(xs.shape, 7.0, dtype=torch.float32, device=d)
    counts = torch.zeros((N, 2 * U), dtype=torch.int32, device=d)
    alphas, betas = torch.empty((N, T, U), device=d), torch.empty((N, T, U), device=d)
    costs = torch.empty((N,), device=d)
    st = L.run_warp_rnnt(torch.cuda.current_stream().cuda_stream, counts.data_ptr(), alphas.data_ptr(), betas.data_ptr(),
                         ys.data_ptr(), xs.data_ptr(), grads.data_ptr(), costs.data_ptr(), txn.data_ptr(), tyn.data_ptr(),
                         N, T, U, V, blank, lam)
    torch.cuda.synchronize()
    assert st == 0
    np.testing.assert_array_equal(grads.cpu().numpy(), ga)


def test_calls_stress():
    """test.py:190-212: N=128,T=100,U=90,V=3, random yn, two seeds -- the reference only checks
    that nothing hangs; here the values are checked too."""
    for i in range(2):
        rng = np.random.RandomState(i)
        xs = rng.randn(128, 100, 90, 3).astype(np.float32)
        lp = np_log_softmax32(xs)
        ys = rng.randint(1, 3, (128, 89)).astype(np.int32)
        xn = np.full((128,), 100, dtype=np.int32)
        yn = rng.randint(1, 90, 128).astype(np.int32)
        ref = oracle.rnnt_loss_f32(lp, ys, xn, yn, blank=0, scan_mode=1)
        c, g = run_native(lp, ys, xn, yn)
        np.testing.assert_allclose(c, ref["costs"], rtol=COST_RTOL)
        np.testing.assert_allclose(g, ref["grads"], atol=GRAD_ATOL)


# ----------------------------------------------------------------------------
# 4. prologue kernels
# ----------------------------------------------------------------------------
# (1024 / 2048 / 3072 / 4096 / 5632 / 6144 / 8192 / 12288 / 16384: where the launcher changes kernel or workgroup
#  shape, prologue.hip: dispatch_lsm, each with its neighbour on the other side;
#  32 ... 128: the rows-in-registers kernel with 1, 2, 3 and 4 rows per group -- 50 is c4 -- and its neighbours that
#  fall back to the LDS-staged one (below 32 -- c2's 28 -- always: its straight-line row pass for 9 ... 16 columns per
#  lane, the run-time loops below that); 1003 rows: a tail that is no whole group;
#  132 ... 1024: the row-in-registers kernel with small workgroups where a row fills its cover (252, 256, 484, 500, 512,
#  724, 768, 964, 1000, 1024) and the LDS-staged one where it does not (132, 244, 248, 480, 600, 720, 960);
#  136, 160, 192: LDS-staged, rows on the same banks)
@pytest.mark.parametrize("V", [2, 3, 5, 9, 13, 16, 17, 20, 24, 28, 30, 32, 33, 36, 40, 42, 48, 50, 51, 64, 80, 100, 126, 128, 132, 136, 160, 192,
                               200, 244, 248, 252, 256,
                               257, 480, 484, 500, 512, 600, 720, 724, 768, 960, 964, 1000, 1024, 1028,
                               1030, 2048, 2052, 2560, 2564, 3072, 3076, 4096, 4100, 5000, 5120, 5124, 5632, 5636,
                               6144, 6148, 8192, 8196, 10000, 12288, 12292, 16384, 16388, 20000])
def test_log_softmax_kernel(V):
    from warp_rnnt_amd import ops
    rows = 1003 if V < 2000 else 77
    x = torch.randn(rows, V, device=dev()) * 3.0
    ref = torch.log_softmax(x.double(), dim=-1)
    out = ops.log_softmax(x)
    assert (out.double() - ref).abs().max().item() < 2e-6 * max(1.0, float(np.log(V)))
    tref = torch.log_softmax(x, dim=-1)
    assert (out - tref).abs().max().item() < 4e-6
    # in place
    y = x.clone()
    ops.log_softmax(y, out=y)
    assert torch.equal(y, out)


def test_log_softmax_4d_and_unaligned():
    from warp_rnnt_amd import ops
    x = torch.randn(3, 7, 5, 50, device=dev())
    np.testing.assert_allclose(ops.log_softmax(x).cpu().numpy(),
                               oracle.log_softmax_f32(x.cpu().numpy()), atol=2e-6)
    base = torch.randn(1 + 37 * 50, device=dev())
    xu = base[1:].view(37, 50)           # 4-byte aligned only -> generic kernel
    np.testing.assert_allclose(ops.log_softmax(xu.contiguous()).cpu().numpy(),
                               torch.log_softmax(xu, -1).cpu().numpy(), atol=4e-6)


@pytest.mark.parametrize("blank", [0, 4])
def test_gather_kernel(blank):
    from warp_rnnt_amd import ops
    logits, labels, xn, yn = make_case(5, 3, 13, 9, 7, blank=blank)
    lp = np_log_softmax32(logits)
    out = ops.gather(t32(lp), t32(labels), blank).cpu().numpy()
    np.testing.assert_array_equal(out, oracle.gather_f32(lp, labels, blank))


# ----------------------------------------------------------------------------
# 5. argument validation (test.py:15-32) -- texts and order
# ----------------------------------------------------------------------------
def test_argument_errors():
    import warp_rnnt._C as core
    xs = torch.tensor([], dtype=torch.float32)
    ys = torch.tensor([], dtype=torch.int)
    xn = torch.tensor([], dtype=torch.int)
    yn = torch.tensor([], dtype=torch.int)
    nc = torch.tensor(np.zeros((4, 3, 2, 1)), dtype=torch.float32).transpose(0, 1)
    with pytest.raises(RuntimeError, match="xs must be contiguous"):
        core.rnnt_loss(nc, ys, xn, yn)
    with pytest.raises(RuntimeError, match="xs must be located in the CUDA"):
        core.rnnt_loss(xs, ys, xn, yn)
    with pytest.raises(RuntimeError, match="xs must have 4 dimensions"):
        core.rnnt_loss(xs.cuda(), ys.cuda(), xn.cuda(), yn.cuda())
    with pytest.raises(RuntimeError, match="ys must be a Int tensor"):
        core.rnnt_loss(xs, torch.tensor([], dtype=torch.long), xn, yn)
    good = torch.zeros((2, 3, 4, 5), device=dev())
    with pytest.raises(RuntimeError, match="xn shape must be equal"):
        core.rnnt_loss(good, torch.zeros((2, 3), dtype=torch.int, device=dev()),
                       torch.ones((3,), dtype=torch.int, device=dev()),
                       torch.ones((2,), dtype=torch.int, device=dev()))
    with pytest.raises(RuntimeError, match="ys shape"):
        core.rnnt_loss(good, torch.zeros((2, 2), dtype=torch.int, device=dev()),
                       torch.ones((2,), dtype=torch.int, device=dev()),
                       torch.ones((2,), dtype=torch.int, device=dev()))
    with pytest.raises(RuntimeError, match="only for blank and label"):
        core.rnnt_loss(good, torch.zeros((2, 3), dtype=torch.int, device=dev()),
                       torch.ones((2,), dtype=torch.int, device=dev()),
                       torch.ones((2,), dtype=torch.int, device=dev()), blank=-1)


# ----------------------------------------------------------------------------
# 6. BASELINE-size lattice: invariants and distance to exact arithmetic
# ----------------------------------------------------------------------------
def test_config4_size_invariants_and_fp64():
    """N=2 utterances of the north-star shape T=1500, U=300 (gathered layout)."""
    N, T, U, V, lam = 2, 1500, 300, 50, 0.01
    logits, labels, xn, yn = make_case(16, N, T, U, V)
    xn[1] = 1337
    yn[1] = 250
    lp = np_log_softmax32(logits)
    lp2 = oracle.gather_f32(lp, labels, 0)
    c, g = run_native(lp2, labels, xn, yn, blank=-1, lam=lam)
    # (a) path-occupancy invariants (exact in exact arithmetic)
    for n in range(N):
        tn, un = xn[n], yn[n] + 1
        np.testing.assert_allclose(g[n, :tn, :un, 0].sum(axis=1), -1.0, atol=2e-2)
        np.testing.assert_allclose(g[n, :tn, :un - 1, 1].sum(axis=0), -(1 + lam), atol=2e-2)
        assert np.all(g[n, tn:] == 0) and np.all(g[n, :, un:] == 0)
    assert np.all(g <= 0) and np.all(g >= -(1 + lam) * 1.001)
    # (b) against the fp32 oracle and fp64 truth
    ref = oracle.rnnt_loss_f32(lp2, labels, xn, yn, blank=-1, fastemit_lambda=lam, scan_mode=1)
    np.testing.assert_allclose(c, ref["costs"], rtol=COST_RTOL)
    c64, g64 = transduce_np.transduce_batch(lp, labels, xn, yn, fastemit_lambda=lam, fast=True)
    np.testing.assert_allclose(c, c64, rtol=2e-6)
    idx = np.zeros((N, T, U, 2), dtype=np.int64)
    idx[:, :, :U - 1, 1] = labels[:, None, :]
    g64_2 = np.take_along_axis(g64, idx, axis=3)
    g64_2[:, :, U - 1, 1] = 0
    for n in range(N):
        g64_2[n, :, yn[n]:, 1] = 0
    err_hip = np.abs(g - g64_2).max()
    err_ora = np.abs(ref["grads"] - g64_2).max()
    print(f"max |grad - fp64|: hip {err_hip:.3e}  fp32-oracle {err_ora:.3e}  hip-vs-oracle "
          f"{np.abs(g - ref['grads']).max():.3e}")
    assert err_hip < 3e-2                      # fp32-vs-exact at this size (~1e-2)
    assert err_hip < 3.0 * err_ora + 1e-3      # no worse than the reference-ordered fp32 restatement


# ----------------------------------------------------------------------------
# 7. shape fuzz around the kernels' structural boundaries (column blocks of 64, the 512-column
#    limit of the wave-specialised lattice kernel, blocks of 8 diagonals, 32x32 gather tiles)
# ----------------------------------------------------------------------------
FUZZ = [(3, t, u) for u in (1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129) for t in (1, 2, 7, 8, 9, 33)] + \
       [(2, 20, 511), (2, 20, 512), (2, 20, 513), (1, 70, 600), (2, 257, 300), (2, 5, 1025)]


def test_shape_fuzz_vs_oracle():
    rng = np.random.RandomState(99)
    for i, (N, T, U) in enumerate(FUZZ):
        V = int(rng.choice([2, 3, 5, 9])) if U > 1 else 3
        logits, labels, xn, yn = make_case(1000 + i, N, T, U, V, ragged=bool(i % 2))
        lp = np_log_softmax32(logits)
        lam = 0.0 if i % 3 else 0.03
        ref = oracle.rnnt_loss_f32(lp, labels, xn, yn, blank=0, fastemit_lambda=lam, scan_mode=1)
        c, g = run_native(lp, labels, xn, yn, blank=0, lam=lam)
        np.testing.assert_allclose(c, ref["costs"], rtol=COST_RTOL, err_msg=f"case {N},{T},{U},{V}")
        np.testing.assert_allclose(g, ref["grads"], atol=GRAD_ATOL, err_msg=f"case {N},{T},{U},{V}")


# ----------------------------------------------------------------------------
# 8. BASELINE config-3 / config-2 shapes at full size, invariants + oracle
# ----------------------------------------------------------------------------
def test_config3_full_size_from_logits():
    """N=32,T=150,U=20,V=5000 (large-vocabulary kernels): fused logits entry vs oracle."""
    from warp_rnnt_amd import ops
    N, T, U, V = 32, 150, 20, 5000
    logits, labels, xn, yn = make_case(3, N, T, U, V)
    costs, g2 = ops.loss(t32(logits), t32(labels), t32(xn), t32(yn), ops.IN_LOGITS_DENSE, ops.GRADS_GATHERED,
                         0, 0.0)
    lp = oracle.log_softmax_f32(logits)
    ref = oracle.rnnt_loss_f32(oracle.gather_f32(lp, labels, 0), labels, xn, yn, blank=-1, scan_mode=1)
    np.testing.assert_allclose(costs.cpu().numpy(), ref["costs"], rtol=COST_RTOL)
    g = g2.cpu().numpy()
    # |alpha| reaches ~1.3e3 here (log-probs ~ -8.5): one fp32 ulp there is 1.2e-4, so two fp32
    # implementations whose log-softmax differs in the last bit disagree by a few 1e-4 in the
    # gradients.  Judge both against exact arithmetic instead.
    lp64 = transduce_np.log_softmax(logits)
    c64, g64 = transduce_np.transduce_batch(lp64, labels, xn, yn, fast=True)
    idx = np.zeros((N, T, U, 2), dtype=np.int64)
    idx[:, :, :U - 1, 1] = labels[:, None, :]
    g64_2 = np.take_along_axis(g64, idx, axis=3)
    g64_2[:, :, U - 1, 1] = 0
    err_hip, err_ora = np.abs(g - g64_2).max(), np.abs(ref["grads"] - g64_2).max()
    print(f"config3: max |grad - fp64|: hip {err_hip:.2e}, fp32 oracle {err_ora:.2e}")
    assert err_hip < 2e-3 and err_hip < 3.0 * err_ora + 2e-4
    np.testing.assert_allclose(costs.cpu().numpy(), c64, rtol=2e-6)
    np.testing.assert_allclose(g[..., 0].sum(axis=2), -1.0, atol=2e-3)
    np.testing.assert_allclose(g[:, :, :-1, 1].sum(axis=1), -1.0, atol=2e-3)


def test_status_codes_for_bad_arguments():
    """The C ABI rejects unusable arguments before any launch (status 5), the Python layer raises."""
    import warp_rnnt_amd
    import warp_rnnt._C as core
    L = warp_rnnt_amd.load()
    assert L.rnnt_amd_loss(None, None, 0, None, None, None, None, None, None, 0, 1, 1, 1, 1, 0, 0.0) == 5
    assert L.run_warp_rnnt(None, None, None, None, None, None, None, None, None, None, 1, 0, 1, 1, 0, 0.0) == 5
    assert L.run_warp_rnnt(None, None, None, None, None, None, None, None, None, None, 1, 1, 1, 4, 7, 0.0) == 5
    assert L.rnnt_amd_log_softmax(None, None, None, -1, 3) == 5
    x = torch.zeros((1, 2, 2, 3), device=dev())
    ys = torch.zeros((1, 1), dtype=torch.int, device=dev())
    one = torch.ones((1,), dtype=torch.int, device=dev())
    with pytest.raises(RuntimeError, match="rnnt_loss status 5"):
        core.rnnt_loss(x, ys, one, one, blank=3)
    c, g = core.rnnt_loss(torch.zeros((0, 2, 2, 3), device=dev()), torch.zeros((0, 1), dtype=torch.int, device=dev()),
                          torch.zeros((0,), dtype=torch.int, device=dev()),
                          torch.zeros((0,), dtype=torch.int, device=dev()))
    assert c.shape == (0,) and g.shape == (0, 2, 2, 3)


@pytest.mark.parametrize("gather", [False, True])
@pytest.mark.parametrize("U", [6, 130])
def test_out_of_range_lengths_are_contained(gather, U):
    """The reference does not check 1 <= xn <= T, 0 <= yn <= U-1 (binding.cpp:47-51) and reads out of
    range.  Here such an utterance gets cost NaN and zero gradients; its neighbours are unaffected."""
    import warp_rnnt._C as core
    N, T, V = 5, 11, 7
    logits, labels, xn, yn = make_case(77, N, T, U, V)
    lp = np_log_softmax32(logits)
    ref = oracle.rnnt_loss_f32(lp, labels, xn, yn, blank=0)
    bad_x, bad_y = xn.copy(), yn.copy()
    bad_x[0], bad_x[2], bad_y[3] = 0, T + 3, U          # utterance 4 and 1 stay valid
    bad_y[0] = -1
    if gather:
        c, g = core.rnnt_loss_gather(t32(lp), t32(labels), t32(bad_x), t32(bad_y), blank=0)
        # internal diagonal-major pairs: compare through the dense expansion
        dense = core.rnnt_loss_gather_backward(torch.ones(N, device=dev()), g, t32(labels), t32(bad_x),
                                               t32(bad_y), V, 0).cpu().numpy()
        for n in (1, 4):
            np.testing.assert_allclose(dense[n], ref["grads"][n], atol=1e-5)
        for n in (0, 2, 3):
            assert not dense[n].any()
    else:
        c, g = core.rnnt_loss(t32(lp), t32(labels), t32(bad_x), t32(bad_y), blank=0)
        g = g.cpu().numpy()
        for n in (1, 4):
            np.testing.assert_allclose(g[n], ref["grads"][n], atol=1e-5)
        for n in (0, 2, 3):
            assert not g[n].any()
    c = c.cpu().numpy()
    assert np.isnan(c[[0, 2, 3]]).all()
    np.testing.assert_allclose(c[[1, 4]], ref["costs"][[1, 4]], rtol=1e-5)


@pytest.mark.parametrize("gather", [False, True])
def test_long_utterance_without_labels_accuracy(gather):
    """U_n == 1 (empty transcript), T_n ~ 1000, sharp logits: alpha/beta are pure prefix/suffix sums.  Serial fp32
    accumulation drifts by ~0.3*sqrt(T) ulp per direction (6e-3 on the gradients here); the reference scans its
    boundary chains (core_gather.cu:86-104,187-205) and so does single_column_scan: the HIP path must stay as
    close to exact arithmetic as the fp32 oracle does."""
    from oracle import transduce_np
    N, T, U, V = 3, 1174, 9, 10
    rng = np.random.RandomState(4)
    logits = (rng.randn(N, T, U, V) * 4.0).astype(np.float32)
    labels = rng.randint(1, V, (N, U - 1)).astype(np.int32)
    xn = np.array([970, 1174, 1], dtype=np.int32)
    yn = np.array([0, 0, 0], dtype=np.int32)
    lp = np_log_softmax32(logits)
    ref = oracle.rnnt_loss_f32(lp, labels, xn, yn, blank=0, scan_mode=1)
    c64, g64 = transduce_np.transduce_batch(lp.astype(np.float64), labels, xn, yn, blank=0, fast=True)
    if gather:
        lp2 = oracle.gather_f32(lp, labels, 0)
        c, g = run_native(lp2, labels, xn, yn, blank=-1)
        idx = np.concatenate([labels, np.zeros((N, 1), np.int32)], 1).astype(np.int64)     # (N,U) label per column
        lab64 = np.take_along_axis(g64, idx[:, None, :, None].repeat(T, 1), axis=3)[..., 0]
        lab64[:, :, U - 1] = 0.0                                                            # no label on the last column
        g64 = np.stack([g64[..., 0], lab64], -1)
        gref = oracle.rnnt_loss_f32(lp2, labels, xn, yn, blank=-1, scan_mode=1)["grads"]
    else:
        c, g = run_native(lp, labels, xn, yn, blank=0)
        gref = ref["grads"]
    np.testing.assert_allclose(c, c64, rtol=2e-6)
    err_hip, err_ora = np.abs(g - g64).max(), np.abs(gref - g64).max()
    print(f"U_n=1, T_n={xn.tolist()}: max |grad - fp64|: hip {err_hip:.2e}, fp32 oracle {err_ora:.2e}")
    assert err_hip <= 3.0 * err_ora + 1e-5


# ----------------------------------------------------------------------------
# 9. the alpha/beta consistency guard (core_gather.cu:341-354; neither repository tested it before)
# ----------------------------------------------------------------------------
def _guard_case():
    """T=1, U=4: the lattice is one chain of label emissions, summed left-to-right by the alpha sweep and
    right-to-left by the beta sweep.  +-3e8 on the chain make fp32 absorb the -7 in one direction only
    (and the final blank differently), so the two log-likelihoods come out as -2 and 0: ratio > 1e-3."""
    N, T, U = 3, 1, 4
    lp2 = np.full((N, T, U, 2), -1.0, dtype=np.float32)
    lp2[1, 0, :, 1] = [3e8, -7.0, -3e8, 0.0]
    lp2[1, 0, 3, 0] = -2.0
    xn = np.ones((N,), dtype=np.int32)
    yn = np.full((N,), U - 1, dtype=np.int32)
    return lp2, xn, yn


def test_mismatch_guard_fires_like_the_oracle():
    from warp_rnnt_amd import ops
    lp2, xn, yn = _guard_case()
    ref = oracle.rnnt_loss_f32(lp2, None, xn, yn, blank=-1, scan_mode=1)
    assert ref["mismatch"].tolist() == [0, 1, 0] and ref["costs"][1] == 1.0     # -( -2 + 0 ) / 2
    # native op (diagonal-major kernels) with the flag vector read back
    c, g, mism = ops.loss(t32(lp2), None, t32(xn), t32(yn), ops.IN_LOG_PROBS_GATHERED, ops.GRADS_GATHERED,
                          -1, 0.0, return_mismatch=True)
    assert mism.cpu().tolist() == [0, 1, 0]
    np.testing.assert_array_equal(c.cpu().numpy(), ref["costs"])
    np.testing.assert_allclose(g.cpu().numpy(), ref["grads"], atol=1e-6)
    assert not g[1].any().item()
    # reference C ABI, gathered and dense layouts (row-major loaders)
    cb, gb = _call_ref_abi(lp2, np.zeros((3, 3), np.int32), xn, yn, -1, 0.0)
    np.testing.assert_array_equal(cb, ref["costs"])
    np.testing.assert_allclose(gb, ref["grads"], atol=1e-6)
    V = 5
    labels = np.array([[1, 2, 3]] * 3, dtype=np.int32)
    dense = np.full((3, 1, 4, V), -9.0, dtype=np.float32)
    dense[..., 0] = lp2[..., 0]
    for u in range(3):
        dense[:, 0, u, labels[0, u]] = lp2[:, 0, u, 1]
    refd = oracle.rnnt_loss_f32(dense, labels, xn, yn, blank=0, scan_mode=1)
    assert refd["mismatch"].tolist() == [0, 1, 0]
    cd, gd = _call_ref_abi(dense, labels, xn, yn, 0, 0.0)
    np.testing.assert_array_equal(cd, refd["costs"])
    np.testing.assert_allclose(gd, refd["grads"], atol=1e-6)
    assert not gd[1].any()


def test_mismatch_policy_env(monkeypatch):
    """WARP_RNNT_AMD_CHECK_MISMATCH surfaces the guard on the host (the reference prints from the device)."""
    import warp_rnnt._C as core
    lp2, xn, yn = _guard_case()
    ys = torch.zeros((3, 3), dtype=torch.int32, device=dev())
    monkeypatch.setenv("WARP_RNNT_AMD_CHECK_MISMATCH", "warn")
    with pytest.warns(RuntimeWarning, match=r"utterance\(s\) \[1\]"):
        core.rnnt_loss(t32(lp2), ys, t32(xn), t32(yn), blank=-1)
    monkeypatch.setenv("WARP_RNNT_AMD_CHECK_MISMATCH", "raise")
    with pytest.raises(RuntimeError, match="forward/backward mismatch"):
        core.rnnt_loss(t32(lp2), ys, t32(xn), t32(yn), blank=-1)
    monkeypatch.setenv("WARP_RNNT_AMD_CHECK_MISMATCH", "off")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        core.rnnt_loss(t32(lp2), ys, t32(xn), t32(yn), blank=-1)   # off: nothing is looked at, nothing is said
        torch.cuda.synchronize()
        core.rnnt_loss(t32(lp2), ys, t32(xn), t32(yn), blank=-1)


def test_mismatch_warning_is_on_by_default_and_needs_no_synchronisation(monkeypatch):
    """The reference prints "WARNING: sample %d [%d, %d] has a forward/backward mismatch %f / %f" from the device whenever
    its guard fires (core_gather.cu:345-349).  Here, by default: the gradient kernel writes the same facts to the device's
    sticky words in pinned host memory (rnnt_amd_mismatch_flag), and the host looks at them around the next call and in
    every backward -- a memory read, no synchronisation -- and warns once per firing.  Through the drop-in wrapper (both
    bindings), dense and gather=True, and the accessor."""
    import warnings
    import warp_rnnt
    import warp_rnnt_amd
    from warp_rnnt_amd import _mismatch
    monkeypatch.delenv("WARP_RNNT_AMD_CHECK_MISMATCH", raising=False)
    lp2, xn, yn = _guard_case()
    V, labels = 5, np.array([[1, 2, 3]] * 3, dtype=np.int32)
    dense = np.full((3, 1, 4, V), -9.0, dtype=np.float32)
    dense[..., 0] = lp2[..., 0]
    for u in range(3):
        dense[:, 0, u, labels[0, u]] = lp2[:, 0, u, 1]
    torch.cuda.synchronize()
    warp_rnnt_amd.last_mismatch()                                   # drain whatever earlier tests left behind
    seen0 = (warp_rnnt_amd.last_mismatch() or {"seen": 0})["seen"]
    for gather in (False, True):
        x = t32(dense).requires_grad_(True)
        with warnings.catch_warnings():
            warnings.simplefilter("error")                          # the forward call itself never waits, never warns
            costs = warp_rnnt.rnnt_loss(x, t32(labels), t32(xn), t32(yn), gather=gather)
        torch.cuda.synchronize()                                    # (the test's own: so that the kernel HAS fired)
        with pytest.warns(RuntimeWarning, match=r"sample 1 \[1, 3\] has a forward/backward mismatch -2\.0+ / 0\.0+"):
            costs.sum().backward()                                  # the look in backward finds it
        assert not x.grad[1].any().item() and costs[1].item() == 1.0
        info = warp_rnnt_amd.last_mismatch()
        assert info["kind"] == "mismatch" and info["utterance"] == 1 and info["frames"] == 1 and info["labels"] == 3
        assert info["loglik_alpha"] == -2.0 and info["loglik_beta"] == 0.0 and info["device"] == 0
    assert warp_rnnt_amd.last_mismatch()["seen"] == seen0 + 2
    # a clean batch says nothing, before or after
    lpc = t32(np.full((2, 3, 4, 2), -1.0, np.float32))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        core_costs = warp_rnnt._C.rnnt_loss(lpc, torch.zeros((2, 3), dtype=torch.int32, device=dev()),
                                            t32(np.full((2,), 3, np.int32)), t32(np.full((2,), 3, np.int32)), blank=-1)
        torch.cuda.synchronize()
        _mismatch.poll(torch.device("cuda:0"))
    assert torch.isfinite(core_costs[0]).all()
    # invalid lengths go the same way (the reference reads out of bounds there)
    bad_xn = np.array([1, 7, 1], dtype=np.int32)                     # 7 > T = 1
    warp_rnnt._C.rnnt_loss(t32(lp2), torch.zeros((3, 3), dtype=torch.int32, device=dev()), t32(bad_xn), t32(yn), blank=-1)
    torch.cuda.synchronize()
    with pytest.warns(RuntimeWarning, match="sample 1 has lengths out of range"):
        assert warp_rnnt_amd.last_mismatch()["kind"] == "invalid lengths"


# ----------------------------------------------------------------------------
# 10. label padding beyond yn[n]: the reference's dense path never reads it, so -1 or any
#     sentinel is legal there; every entry point must give the bits it gives with a valid padding value
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("V", [7, 1500])       # LDS-tile kernels / row-per-workgroup kernels
@pytest.mark.parametrize("pad", [-1, "V+7"])
def test_padded_labels_may_hold_any_sentinel(V, pad):
    import warp_rnnt
    from warp_rnnt_amd import ops
    from warp_rnnt_amd.fused import rnnt_loss_from_logits
    N, T, U = 4, 19, 9
    logits, labels, xn, yn = make_case(321, N, T, U, V, ragged=True)
    yn[0] = 0                                      # an utterance whose whole label row is padding
    padv = V + 7 if pad == "V+7" else pad
    bad = labels.copy()
    for n in range(N):
        bad[n, yn[n]:] = padv
        labels[n, yn[n]:] = 1
    lp = np_log_softmax32(logits)

    def run(lab, how):
        x = t32(lp if how != "fused" else logits).requires_grad_(True)
        if how == "fused":
            c = rnnt_loss_from_logits(x, t32(lab), t32(xn), t32(yn), fastemit_lambda=0.01)
        else:
            c = warp_rnnt.rnnt_loss(x, t32(lab), t32(xn), t32(yn), gather=(how == "gather"), fastemit_lambda=0.01)
        (c * torch.arange(1, N + 1, device=dev())).sum().backward()
        torch.cuda.synchronize()
        return c.detach().cpu().numpy(), x.grad.cpu().numpy()

    ref = oracle.rnnt_loss_f32(lp, labels, xn, yn, blank=0, fastemit_lambda=0.01, scan_mode=1)
    for how in ("dense", "gather", "fused"):
        c_ok, g_ok = run(labels, how)
        c_bad, g_bad = run(bad, how)
        np.testing.assert_array_equal(c_bad, c_ok, err_msg=how)
        np.testing.assert_array_equal(g_bad, g_ok, err_msg=how)
        np.testing.assert_allclose(c_ok, ref["costs"], rtol=COST_RTOL, err_msg=how)
    # the row-major gather entry as well
    out = ops.gather(t32(lp), t32(bad), 0).cpu().numpy()
    good = oracle.gather_f32(lp, labels, 0)
    for n in range(N):
        np.testing.assert_array_equal(out[n, :, :yn[n] + 1, 0], good[n, :, :yn[n] + 1, 0])
        np.testing.assert_array_equal(out[n, :, :yn[n], 1], good[n, :, :yn[n], 1])


# ----------------------------------------------------------------------------
# 9. masked log-probs (-inf) -- the rim of the lattice is plain sums in the reference (core_gather.cu:76-104, 177-205:
#    a -inf there stays -inf and the lattice behind it stays finite), the interior is log_sum_exp (core_gather.cu:22-35:
#    lse(-inf, -inf) = NaN).  Same NaN pattern, same zeros, same finite values as the oracle, on every route.
# ----------------------------------------------------------------------------
def _masked_case(N, T, U, V, seed):
    """Ragged batch with -inf planted: utterance 0 a masked label in row 0 (alpha's rim) and one in the last row (beta's),
    utterance 1 a masked blank in column 0 and one in the last column, utterance 2 a whole masked frame (every path
    crosses it: NaN), utterance 3 interior cells only, the rest untouched."""
    logits, labels, xn, yn = make_case(seed, N, T, U, V, ragged=True)
    lp = np_log_softmax32(logits)
    lab = lambda n, u: labels[n, u]                               # noqa: E731
    lp[0, 0, 1, lab(0, 1)] = -np.inf
    lp[0, xn[0] - 1, min(2, yn[0] - 1), lab(0, min(2, yn[0] - 1))] = -np.inf
    lp[1, min(2, xn[1] - 2), 0, 0] = -np.inf
    lp[1, xn[1] // 2, yn[1], 0] = -np.inf
    lp[2, xn[2] // 2, :, :] = -np.inf
    lp[3, xn[3] // 3, yn[3] // 2, lab(3, yn[3] // 2)] = -np.inf
    lp[3, xn[3] // 2, yn[3] // 3, 0] = -np.inf
    return lp, labels, xn, yn


def _same_pattern(got_c, got_g, ref, atol, what):
    rc, rg = ref["costs"], ref["grads"]
    np.testing.assert_array_equal(np.isnan(got_c), np.isnan(rc), err_msg=what + ": NaN costs")
    np.testing.assert_array_equal(np.isnan(got_g), np.isnan(rg), err_msg=what + ": NaN gradients")
    ok = ~np.isnan(rc)
    np.testing.assert_allclose(got_c[ok], rc[ok], rtol=COST_RTOL, err_msg=what)
    okg = ~np.isnan(rg)
    np.testing.assert_allclose(got_g[okg], rg[okg], atol=atol, err_msg=what)
    # exact zeros: where one side has none of a gradient (a masked arc, a dead cell) the other has none either -- up to
    # the denormals exp() flushes on one side and keeps on the other (|g| < 1e-37)
    assert np.abs(got_g[(rg == 0) & okg]).max(initial=0.0) < 1e-37 and np.abs(rg[(got_g == 0) & okg]).max(initial=0.0) < 1e-37, what


@pytest.mark.parametrize("N,T,U,V", [(5, 9, 6, 5), (5, 70, 40, 6), (5, 260, 150, 5), (6, 80, 330, 4)])
def test_masked_log_probs_nan_pattern_equals_the_oracles(N, T, U, V):
    import warp_rnnt
    from warp_rnnt_amd import debug
    np.seterr(all="ignore")
    lp, labels, xn, yn = _masked_case(N, T, U, V, 4242 + T)
    ref = oracle.rnnt_loss_f32(lp, labels, xn, yn, blank=0, scan_mode=1)
    assert np.isnan(ref["costs"][2]) and np.isfinite(ref["costs"][[0, 1, 3]]).all()     # what the reference does
    lp2 = oracle.gather_f32(lp, labels, 0)
    ref2 = oracle.rnnt_loss_f32(lp2, labels, xn, yn, blank=-1, scan_mode=1)
    atol = 1e-4 if T + U <= 250 else 5e-4
    # native op, dense and gathered layouts, under every lattice kernel
    for kern in ("auto", "ws", "wd", "wl"):
        with debug.lattice_kernel(kern):
            c, g = run_native(lp, labels, xn, yn, blank=0)
            _same_pattern(c, g, ref, atol, f"dense, kernel {kern}")
            c2, g2 = run_native(lp2, labels, xn, yn, blank=-1)
            _same_pattern(c2, g2, ref2, atol, f"gathered, kernel {kern}")
    # reference C ABI, both layouts
    ca, ga = _call_ref_abi(lp, labels, xn, yn, 0, 0.0)
    _same_pattern(ca, ga, ref, atol, "run_warp_rnnt")
    cb, gb = _call_ref_abi(lp2, labels, xn, yn, -1, 0.0)
    _same_pattern(cb, gb, ref2, atol, "run_warp_rnnt_gather")
    # the wrapper, gather=True: dense gradients through the gather's backward
    x = t32(lp).requires_grad_(True)
    costs = warp_rnnt.rnnt_loss(x, t32(labels), t32(xn), t32(yn), gather=True)
    costs.sum().backward()
    _same_pattern(costs.detach().cpu().numpy(), x.grad.cpu().numpy(), ref, atol, "rnnt_loss(gather=True)")
    # compact layout
    V_ = lp.shape[-1]
    rows = np.concatenate([lp[n, :xn[n], :yn[n] + 1].reshape(-1, V_) for n in range(N)])
    labs = np.concatenate([labels[n, :yn[n]] for n in range(N)]).astype(np.int32)
    xr = t32(np.ascontiguousarray(rows)).requires_grad_(True)
    cc = warp_rnnt.rnnt_loss(xr, t32(labs), t32(xn), t32(yn), compact=True)
    cc.sum().backward()
    want = np.concatenate([ref["grads"][n, :xn[n], :yn[n] + 1].reshape(-1, V_) for n in range(N)])
    _same_pattern(cc.detach().cpu().numpy(), xr.grad.cpu().numpy(), {"costs": ref["costs"], "grads": want}, atol,
                  "rnnt_loss(compact=True)")


def test_the_reference_named_entry_points_give_one_utterance_the_same_bits_in_any_batch():
    """csrc/api.hip: run_warp_rnnt / run_warp_rnnt_gather take the staged forms from 2^20 cells on and the direct ones
    below -- different kernels (the tuned column-block kernels on a staged plane / the single-role kernel on the
    caller's layout), one arithmetic: an utterance's costs and gradients do not depend on the batch it came in."""
    T, U, V, lam = 300, 200, 5, 0.01
    logits, labels, xn, yn = make_case(77, 2, T, U, V, ragged=True)
    lp = np_log_softmax32(logits)
    reps = 9                                                         # 18 x 300 x 200 = 1.08 M cells: staged
    assert 2 * T * U < (1 << 20) <= 2 * reps * T * U
    big = lambda a: np.concatenate([a] * reps)                       # noqa: E731
    for blank, x in ((0, lp), (-1, oracle.gather_f32(lp, labels, 0))):
        c_small, g_small = _call_ref_abi(x, labels, xn, yn, blank, lam)
        c_big, g_big = _call_ref_abi(big(x), big(labels), big(xn), big(yn), blank, lam)
        np.testing.assert_array_equal(c_big[:2], c_small)
        np.testing.assert_array_equal(c_big[-2:], c_small)
        np.testing.assert_array_equal(g_big[:2], g_small)
        np.testing.assert_array_equal(g_big[-2:], g_small)


def test_one_arithmetic_and_a_debug_only_kernel_pin():
    """Round 6: the library has ONE arithmetic (the probability-domain route and the per-call / process-wide route
    settings are gone: nothing a caller can set changes a bit of the result).  What is left is a debug pin of the KERNEL,
    and every kernel gives the same bits; two threads with different pins cannot disagree."""
    import threading
    import warp_rnnt_amd
    from warp_rnnt_amd import debug, ops
    N, T, U, V = 3, 700, 70, 6
    logits, labels, xn, yn = make_case(5, N, T, U, V, ragged=True)
    lp2 = t32(oracle.gather_f32(np_log_softmax32(logits), labels, 0))
    txn, tyn = t32(xn), t32(yn)
    run = lambda: ops.loss(lp2, None, txn, tyn, ops.IN_LOG_PROBS_GATHERED, ops.GRADS_GATHERED)   # noqa: E731
    for gone in ("set_lattice", "get_lattice", "lattice_route", "set_logdomain_kernel"):
        assert not hasattr(warp_rnnt_amd, gone), gone
    L = warp_rnnt_amd.load()
    for gone in ("rnnt_amd_set_lattice", "rnnt_amd_loss_ex", "rnnt_amd_loss_compact_ex", "rnnt_amd_set_logdomain_kernel"):
        assert not hasattr(L, gone), gone
    with pytest.raises(TypeError):
        ops.loss(lp2, None, txn, tyn, ops.IN_LOG_PROBS_GATHERED, ops.GRADS_GATHERED, lattice="pd")
    assert debug.get_lattice_kernel() == "auto"
    c_def, g_def = run()
    results = {}

    def worker(kern):
        for _ in range(20):
            with debug.lattice_kernel(kern):       # (process-wide: the threads race on the pin -- and it must not matter)
                results[kern] = run()
    threads = [threading.Thread(target=worker, args=(k,)) for k in ("ws", "wl")]
    [t.start() for t in threads]
    [t.join() for t in threads]
    torch.cuda.synchronize()
    for kern, (c, g) in results.items():
        assert torch.equal(c, c_def) and torch.equal(g, g_def), kern
    debug.set_lattice_kernel("auto")
    with pytest.raises(ValueError, match="unknown lattice kernel"):
        debug.set_lattice_kernel("exact")
    assert L.rnnt_amd_debug_set_lattice_kernel(9) == -1 and debug.get_lattice_kernel() == "auto"


@pytest.mark.parametrize("N,T,U", [(3, 100, 9), (3, 100, 33), (2, 200, 65), (2, 300, 129), (3, 90, 41)])
def test_last_column_starting_on_a_block_boundary(N, T, U):
    """U - 1 a multiple of the block size (8 diagonals): the last column's first cell -- a rim cell -- is the first
    diagonal of a block.  It must be computed by a block variant that knows the rim rule (and, in the column-block
    kernels, before the steady-state blocks take over: what a lane holds before its first cell is unspecified there).
    With the label arc into that cell masked: alpha[0, U-1] = -inf exactly, a finite cost, no NaN anywhere."""
    from warp_rnnt_amd import debug
    np.seterr(all="ignore")
    logits, labels, xn, yn = make_case(900 + U, N, T, U, 5)
    lp = np_log_softmax32(logits)
    lp2 = oracle.gather_f32(lp, labels, 0)
    lp2[0, 0, U - 2, 1] = -np.inf                       # the only way into (0, U-1)
    lp2[1, T - 1, 0, 1] = -np.inf                       # beta's mirror image: the only way out of (T-1, 0) ... to the right
    ref = oracle.rnnt_loss_f32(lp2, labels, xn, yn, blank=-1, scan_mode=1)
    assert np.isfinite(ref["costs"]).all() and np.isneginf(ref["alphas"][0, 0, U - 1])
    for kern in ("auto", "ws", "wd", "wl"):
        with debug.lattice_kernel(kern):
            c, g = run_native(lp2, labels, xn, yn, blank=-1)
        _same_pattern(c, g, ref, 1e-4 if T + U <= 250 else 5e-4, f"kernel {kern}")
    ca, ga = _call_ref_abi(lp2, labels, xn, yn, -1, 0.0)
    _same_pattern(ca, ga, ref, 1e-4 if T + U <= 250 else 5e-4, "run_warp_rnnt_gather")
