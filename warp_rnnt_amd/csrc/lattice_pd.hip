// Probability-domain alpha / beta lattice sweep for MI355X (gfx950), diagonal-major layout (padded or compact),
// one workgroup (five waves: one computes, two load, two store) per 64-column block of a sweep.
//
// Same anti-diagonal schedule as lattice_ws.hip (read its header first: lanes are lattice columns, a column
// block of 64 runs behind its left neighbour, blocks of K=8 diagonals, one s_barrier per block).  Two
// things change.
//
// 1. The ARITHMETIC of the serial chain.  The log-domain step lse(a,b) = max + log1p(exp(-|a-b|)) costs the
//    computing wave ~19 instructions per diagonal and a lone wave issues one instruction every ~6.5 cycles
//    (tools/ubench/pd_steps.hip: 65 cycles per step).  Here the recurrence runs on PROBABILITIES with a
//    per-column binary exponent:
//        true alpha of column u on the current diagonal = A[u] * 2^E[u]          A fp64, E int32
//        per diagonal      val = fma(A_left, 2^(E_left - E_own), Y)              the factor is exact
//                          Y   = val * pB(cell)      X = val * pL(cell)           (beta: mirrored lattice, the
//                                                                                  weights of the receiving cell)
//        every K diagonals Y and X are rescaled by the exponent of Y (exact) and E takes it up
//    = two 32-bit DPP moves and three or four fp64 operations per diagonal (27 cycles per step in the same
//    micro-benchmark; fp64 VALU issues at the fp32 rate on this chip).  fp64 state because its exponent range
//    (2^+-1022) cannot be left within K steps of fp32-representable probabilities (>= 2^-126 each): no range
//    bookkeeping inside the chain, and columns may differ by hundreds of binary orders of magnitude (they do:
//    tests/pd_model.py, tests/test_pd_model.py).  The conversions live on two helper waves, off the chain:
//      LOADER  HBM -> registers -> p = exp2(lp*log2e) (v_mul_f32, v_exp_f32, v_cvt_f64_f32) -> LDS, two blocks
//              ahead; also the input check: a probability that leaves [2^-115, 2^100] (log-prob below -80, -inf,
//              or far above 0) cannot be carried as an fp32 -- the (utterance, direction) is then redone by the
//              log-domain kernel (lattice_ws.hip, launched behind this one; it returns at once otherwise);
//      STORER  LDS -> ln2*(log2(mantissa) + exponent) -> HBM, one block behind: the top 24 significant bits of
//              the fp64 value relabelled as an fp32 in [1,2) (v_alignbit_b32, v_and_or_b32), v_log_f32, one
//              multiply + one fma: the stored alpha/beta carry <= 0.5 ulp of rounding plus 2^-23 relative --
//              nothing accumulates along the sweep (the log-domain chain rounds at |alpha| ~ 6e3 every step:
//              1e-2 on the gradients at T=1500,U=300, against 8e-4 here).
//    Off the chain as well, the chain's own range checks (round 3): the extremes of E_left - E_own over the
//    renormalisations (neighbouring columns may drift further apart than an fp64 holds while every input is in range)
//    and a final-value check (overflow, NaN and total underflow are sticky); either flags the sweep for the log-domain
//    kernel like an out-of-range input does.
//    tests/pd_model.py is the executable statement of this arithmetic (checked on the CPU against fp64).
//
// 2. WHERE the column blocks run.  With all column blocks of a sweep in one workgroup (lattice_ws.hip) a
//    U=300 sweep puts ten waves on the four SIMDs of one CU and the CU's issue rate becomes the limit (95 us at
//    U=64, 166 us at U=300 for T=1500) while seven eighths of the chip idle at N=16.  Here every column block
//    is its own workgroup, wherever the dispatcher puts it; the boundary column travels through L2:
//      * the producer's first storer wave publishes, per block of 16 diagonals, 34 self-validating 8-byte granules
//        {32 data bits, 32-bit tag = launch epoch ^ hash(ring) ^ hash(block), never 0} with one agent-scope (sc1) store
//        instruction; the rings are zeroed in front of every launch (k_prepare) and the epoch comes from a launch
//        counter in module-scope device memory, so a granule validates only if THIS launch wrote it;
//      * the consumer's loader wave requests them two blocks ahead with agent-scope loads (its HBM prefetch
//        queue), checks the tags when it needs the block and re-polls only while the producer is not there
//        yet -- so the consumer settles a few microseconds behind the producer and the hand-over costs
//        nothing per block; there is no flag, no fence, no back-pressure (the ring is as long as the sweep);
//      * work items (column block, sweep) are handed out by an atomic counter in column-block-major order:
//        whoever holds item i knows every item < i has been taken by a workgroup that is running or done, so
//        a consumer never waits for a producer that has not been scheduled (no assumption about dispatch order);
//      * every wait is bounded; a timeout flags the sweep for the log-domain kernel instead of hanging.
//
// Reference counterpart: core_gather.cu:37-133 (alphas), :135-234 (betas) -- 32x1 warp tiles ordered by
// global spin locks, lse per cell.  Outputs are the same quantities (log alpha, log beta, fp32).
#include <algorithm>
#include <atomic>
#include <random>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace rnnt {

namespace pd {

#ifndef RNNT_PD_K
#define RNNT_PD_K 16
#endif
constexpr int K = RNNT_PD_K;     // diagonals per block (one s_barrier, one hand-over per block)
constexpr int KR = 8;            // diagonals between renormalisations
constexpr int NH = K / KR;       // renormalisations ("halves") per block
static_assert(K % KR == 0 && (K == 8 || K == 16), "block = one or two renormalisation intervals");
constexpr int KH = K / 2;        // diagonals per block and helper wave
constexpr int PSLOTS = 2;        // LDS ring of probability blocks
constexpr int VSLOTS = 2;        // LDS ring of value blocks
constexpr int MSLOTS = 2;        // LDS rings of boundary-column blocks (incoming and outgoing)
constexpr int DLOAD = 2;         // the loader issues its HBM loads this many blocks before it converts them
constexpr int NBR = DLOAD + 1;   // its register ring
constexpr int PSTRIDE = 4 * K + 4;      // dwords per lane per probability slot: 8 cells x (pB,pL) fp64 = 32, +4: the
                                 // 16-lane groups of ds_read/write_b128 then cover all 64 banks (MI355X_MICROARCH.md)
constexpr int VSTRIDE = 2 * K + 4;      // dwords per lane per value slot: 8 fp64 = 16, +4 (same reason)
constexpr int GRAN = 2 * K + NH;         // granules per block: 16 dwords of lane 63's X on entry to the 8 steps + its exponent
constexpr int GPITCH = 4 * K;       // granules reserved per block in the global ring (256 bytes)
constexpr int RSRC_WORD3 = 0x00020000;
constexpr int OOB = (int)0x80000000;
constexpr float P_MIN = 0x1p-115f, P_MAX = 0x1p100f;   // accepted range of an fp32 probability (see header)
// Largest binary-exponent gap between neighbouring columns the fp64 state is trusted with.  Within one
// renormalisation interval a value can grow by the gap of every column boundary it crosses (KR of them at most) on
// top of its own 2^+-1008 of drift: 400 leaves two such crossings inside 2^1022, anything beyond shows up as a
// non-finite final value, which is checked too (sweep()).  Measured gaps of sharp-model data: 2^60...2^120
// (tests/test_pd_model.py).
constexpr int MAX_GAP = 400;
constexpr int GAP_CLAMP = 1000;
constexpr float LOG2E = 1.44269504088896340736f;
constexpr float LN2 = 0.693147180559945309417f;
#ifndef RNNT_PD_SPIN_LIMIT
#define RNNT_PD_SPIN_LIMIT (1 << 21)
#endif
constexpr int SPIN_LIMIT = RNNT_PD_SPIN_LIMIT;           // polls before a hand-over is declared lost (a few seconds;
                                                         // the `short_spin` build variant sets 0: tests/test_gpu_pd.py)
#ifndef RNNT_PD_LAG
#define RNNT_PD_LAG 1
#endif
constexpr int LAG = RNNT_PD_LAG; // blocks a column block lets its left neighbour get ahead when it has caught up with it

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

struct alignas(16) Smem {
    float probs[PSLOTS][WAVE * PSTRIDE];    // [lane][k] (pB, pL) fp64
    float vals[VSLOTS][WAVE * VSTRIDE];     // [lane][k] fp64 value, scale 2^exps[lane]
    int exps[VSLOTS][NH][WAVE];          // one scale per renormalisation interval
    unsigned mail_in[MSLOTS][GPITCH];       // left neighbour's boundary column, as its granule payloads
    unsigned mail_out[MSLOTS][GPITCH];      // this block's boundary column: [0,16) X, [16] exponent
    double dumpx[WAVE][2];                  // where the other 63 lanes put their copy of the boundary column
    int dumpe[WAVE];
};

__device__ __forceinline__ void block_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Lane i receives src from lane i-1; lane 0 receives 0.  Two v_mov_b32_dpp wave_shr:1 bound_ctrl:1.
__device__ __forceinline__ double wave_shr1_f64(double src) {
    const u64 b = __builtin_bit_cast(u64, src);
    const int lo = __builtin_amdgcn_mov_dpp((int)b, 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((u64)(unsigned)hi << 32) | (unsigned)lo);
}
// Lane i receives src from lane i-1; lane 0 keeps `first` (DPP "old" operand, bound_ctrl off: no extra instruction
// when `first` dies here and the destination takes its registers).
__device__ __forceinline__ double wave_shr1_f64_seed(double first, double src) {
    const u64 b = __builtin_bit_cast(u64, src), f = __builtin_bit_cast(u64, first);
    const int lo = __builtin_amdgcn_update_dpp((int)f, (int)b, 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(f >> 32), (int)(b >> 32), 0x138, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((u64)(unsigned)hi << 32) | (unsigned)lo);
}
// Lane i receives src from lane i-1; lane 0 receives `first`.
__device__ __forceinline__ int wave_shr1_i32(int first, int src) {
    return __builtin_amdgcn_update_dpp(first, src, 0x138, 0xf, 0xf, false);
}

// Tag of block lb of one ring: launch epoch, ring (sweep and boundary) and block folded together, never 0 -- the rings
// are cleared to zero in front of every launch (k_prepare), so a granule validates only if THIS launch wrote it,
// whatever the workspace held before (fresh allocator memory, another shape's rings, an earlier graph replay).
__device__ __forceinline__ unsigned ring_tag(unsigned epoch, unsigned ring) { return epoch ^ (ring * 0x85EBCA6Bu); }
__device__ __forceinline__ unsigned block_tag(unsigned ring_epoch, int lb) {
    const unsigned t = ring_epoch ^ ((unsigned)(lb + 1) * 0x9E3779B1u);
    return t ? t : 1u;
}

struct Cell2 { double b, l; };   // blank / label probability of one lattice cell

// One block of the compute wave: renormalisation + K diagonals.  State: Y, X (fp64), E (binary exponent).
//   SEED:   the column block has a left neighbour: lane 0 receives seed[k] (scale 2^e_mail) instead of nothing
//   MASKED: some lane starts or finishes inside the block (updates are predicated, E of lanes that have not
//           started follows the last started lane)
//   MAIL:   lane 63's X on entry to every step is recorded for the right neighbour
template <bool BETA, bool MASKED, bool SEED, bool MAIL>
__device__ __forceinline__ void compute_block(const Cell2 (&cur)[K], Cell2 (&nxt)[K], const double (&seed)[K],
                                              const int e_mail0, const int e_mail1, double& Y, double& X, int& E,
                                              const int d0, const int ucol_chk, const int Tn, const int wave_c,
                                              const float* next_probs, float* vdst, double* xdst, int* edst,
                                              int* email_dst, int& gap_hi, int& gap_lo) {
    double vprev = 0.0, xprev = 0.0, c = 1.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (k % KR == 0) {
            // ---- renormalisation (exact: powers of two), every KR diagonals ----
            const int h = k / KR;
            const int e = __builtin_amdgcn_frexp_exp(Y);      // 0 for Y == 0
            Y = __builtin_amdgcn_frexp_mant(Y);
            X = __builtin_ldexp(X, -e);
            E += e;
            const int e_mail_h = h ? e_mail1 : e_mail0;
            if constexpr (MASKED) {
                // lanes that have not started adopt the exponent of the last started column (their first value
                // arrives from it), or the mailbox's when the whole column block has not started
                const int front = d0 + k - 1 - wave_c;         // last lane that has started (may be < 0 or > 63)
                const bool lane_started = (ucol_chk <= d0 + k - 1) || (ucol_chk == 0);
                int ef;
                if (front >= 0) ef = __builtin_amdgcn_readlane(E, front > WAVE - 1 ? WAVE - 1 : front);
                else ef = SEED ? e_mail_h : __builtin_amdgcn_readlane(E, 0);
                if (front < WAVE - 1) E = lane_started ? E : ef;
            }
            edst[h * WAVE] = E;                               // the storer's scale for this interval's values
            if constexpr (MAIL) email_dst[h] = E;             // (lane 63's pointer; the others aim at a dump slot)
            const int e_left = wave_shr1_i32(SEED ? e_mail_h : E, E);
            // 2^(E_left - E_own); lane 0 of the first block: 1.  The gap between neighbouring columns is unbounded in
            // principle (a frame at which the label probabilities step up puts (pL(t+1)/pL(t))^u between two columns
            // of one diagonal: 2^1558 with every log-prob above -10) and fp64 holds 2^+-1022: the extremes of the gap
            // are recorded (two instructions per interval) and a sweep that saw more than MAX_GAP is handed to the
            // log-domain kernel; the shift itself is clamped so that the factor stays finite either way.
            int gap = e_left - E;
            if constexpr (MASKED) {
                // a finished lane's exponent is frozen while its left neighbours' move on: not a gap of the lattice
                const bool done = d0 + k - ucol_chk >= Tn;
                gap = done ? 0 : gap;
            }
            gap_hi = max(gap_hi, gap);
            gap_lo = min(gap_lo, gap);
            c = __builtin_ldexp(1.0, min(max(gap, -GAP_CLAMP), GAP_CLAMP));
        }
        const double xin = X;                             // what the right neighbour reads at this step
        // the left neighbour's value: the lane to the left, or -- lane 0 of a column block that has a neighbour --
        // the boundary column that neighbour published (its scale is in lane 0's factor c)
        double xl;
        if constexpr (SEED) xl = wave_shr1_f64_seed(seed[k], X);
        else xl = wave_shr1_f64(X);
        double val;
        if constexpr (BETA) {
            // beta[t,u] = beta[t+1,u]*pB[t,u] + beta[t,u+1]*pL[t,u]: both weights belong to the receiving cell
            val = __builtin_fma(xl, cur[k].l * c, Y * cur[k].b);
        } else {
            // alpha[t,u] = alpha[t-1,u]*pB[t-1,u] + alpha[t,u-1]*pL[t,u-1]: Y and X carry the products
            val = __builtin_fma(xl, c, Y);
        }
        double Yn, Xn;
        if constexpr (BETA) { Yn = val; Xn = val; }
        else { Xn = val * cur[k].l; Yn = val * cur[k].b; }
        if constexpr (MASKED) {
            const bool live = (unsigned)(d0 + k - ucol_chk) < (unsigned)Tn;
            Y = live ? Yn : Y;
            X = live ? Xn : X;
        } else {
            Y = Yn; X = Xn;
        }
        // one cell of the next block per step (written by the loaders one block ago, needed one block from
        // now; behind the last block this reads a stale slot that nobody uses)
        {
            const f64x2 pr = *reinterpret_cast<const f64x2*>(next_probs + 4 * k);
            nxt[k].b = pr.x; nxt[k].l = pr.y;
        }
        // publish two steps at a time (16-byte LDS stores)
        if (k & 1) {
            f64x2 two;
            two.x = vprev; two.y = val;
            *reinterpret_cast<f64x2*>(vdst + 2 * (k - 1)) = two;
            if constexpr (MAIL) {
                two.x = xprev; two.y = xin;
                *reinterpret_cast<f64x2*>(xdst + (k - 1)) = two;
            }
        } else {
            vprev = val; xprev = xin;
        }
    }
}

struct Item { int n, dir, cb; };

// HAS_LEFT / HAS_RIGHT are template parameters so that the helper waves' steady-state loops contain no
// data-dependent branch around their memory instructions: with one, the compiler's s_waitcnt bookkeeping falls
// back to vmcnt(0) at the join and every block pays a full HBM round trip (measured: 1.3 us instead of 0.45 us
// per block).
template <bool BETA, bool COMPACT, bool HAS_LEFT, bool HAS_RIGHT>
__device__ __forceinline__ void sweep(const LatticeArgs& a, const Item it, const UttLens len, const int nA, Smem& sm,
                                      int* wg_bad) {
    const int n = it.n, idx = it.cb;
    const int Tn = len.Tn, Un = len.Un;
    const int T = COMPACT ? Tn : a.T, U = COMPACT ? Un : a.U;
    const int lane = threadIdx.x & (WAVE - 1);
    // five waves: 0 computes; 1,2 load (diagonals 0-3 / 4-7 of every block); 3,4 store (same split).  The conversions
    // cost a helper wave about as many instructions per diagonal as the chain costs the compute wave; two helpers of
    // each kind keep them off the critical path
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wv == 0 ? 0 : (wv <= 2 ? 1 : 2);   // 0 compute, 1 loader, 2 storer
    const int half_rt = (wv - 1) & 1;                   // which four diagonals of a block a helper wave handles
    if (Un == 1) {   // no labels: prefix / suffix sums by one wave of the first column block's workgroup
        if (idx == 0 && role == 0) {
            const size_t nb1 = COMPACT ? (size_t)a.offs[n] : (size_t)n * T * U;
            const float2* lp2 = reinterpret_cast<const float2*>(a.lp) + nb1;
            const float total = single_column_scan<BETA>(Tn, (BETA ? a.betas : a.alphas) + nb1, U, lane,
                                                         [&](int t) { return lp2[(size_t)t * U].x; });
            if (!BETA && lane == 0) a.ll[n] = total;
        }
        return;
    }
    const size_t nbase = COMPACT ? (size_t)a.offs[n] : (size_t)n * T * U;
    float* out = (BETA ? a.betas : a.alphas) + nbase;
    const int ndiag = Tn + Un - 1;

    const int wave_c = WAVE * idx;                    // first sweep column of this column block
    const int ucol = wave_c + lane;                   // column in sweep coordinates
    const bool colvalid = ucol < Un;
    const int u = BETA ? (Un - 1 - ucol) : ucol;
    const int uc = min(max(u, 0), U - 1);
    const int ucol_chk = colvalid ? ucol : 0x40000000;
    const int nwa = (Un + WAVE - 1) / WAVE;           // column blocks with a live column
    const int lo = wave_c / K;
    const int hi = (min(ndiag, Tn + wave_c + WAVE) + K - 1) / K;
    if (idx >= nwa || lo >= hi) return;               // nothing to sweep here (uniform)
    constexpr bool has_left = HAS_LEFT, has_right = HAS_RIGHT;
    // blocks for which the left neighbour publishes its boundary column
    const int hi_left = has_left ? (min(ndiag, Tn + wave_c) + K - 1) / K : 0;
    const int G = (hi - lo) + 3 + DLOAD;              // barriers every wave executes
    // Every lane that owns a lattice column live for the whole block?  (all of them started, none finished.)  Lanes
    // beyond the last column (the last column block of a lattice whose width is not a multiple of 64) then run the
    // unpredicated code too: they read an in-range cell, their values are garbage that only travels to the right
    // -- to other such lanes --, their stores are dropped by the buffer bounds check (voffset = OOB) and the input
    // check ignores them.  Without this the last column block would run the predicated paths for the whole sweep
    // and, being the slowest link of the chain, set the pace (T=1500: 305 us at U=300 against 130 us at U=128).
    // (started BEFORE the block, d0 > last_col: a column that starts in the block takes its exponent from its
    // left neighbour in the predicated renormalisation -- found by tools/fuzz_parity.py on lattices whose last
    // column block holds a single column: its factor 2^(E_left - 0) underflowed fp64 and the sweep went to zero.)
    const int last_col = min(wave_c + WAVE - 1, Un - 1);
    auto full_block = [&](const int lb) {
        const int d0 = lb * K;
        return (d0 > last_col) && (d0 + K <= wave_c + Tn);
    };
    // hand-over rings in global memory: one per (sweep, column-block boundary)
    const size_t sweep_id = (size_t)2 * n + (BETA ? 1 : 0);
    u64* ring_in = has_left ? a.mail + ((sweep_id * (nA - 1) + (idx - 1)) * (size_t)a.mail_blocks) * GPITCH : nullptr;
    u64* ring_out = has_right ? a.mail + ((sweep_id * (nA - 1) + idx) * (size_t)a.mail_blocks) * GPITCH : nullptr;
    const unsigned tag_in = ring_tag(a.epoch, (unsigned)(sweep_id * (nA - 1) + (idx - 1)));
    const unsigned tag_out = ring_tag(a.epoch, (unsigned)(sweep_id * (nA - 1) + idx));
    (void)tag_in; (void)tag_out;
#ifdef RNNT_PD_STATS       // diagnostics build (tools/pd_trace.py): a (sweep, column block, interval) table of
                           // s_memrealtime stamps behind the rings -- 16 words per interval: 0 compute start, 1 time the
                           // first loader polled for the neighbour, 2/3 loader start, 4/5 storer start, 6 compute end,
                           // 7/8 loader end, 9/10 storer end.  Costs a few per cent; never part of the product build.
    u64* const trace = a.mail + (size_t)2 * (gridDim.x / nA / 2) * (nA - 1) * a.mail_blocks * GPITCH +
                       ((sweep_id * nA + idx) * (size_t)(a.mail_blocks + 8)) * 16;
#define RNNT_PD_STAMP(slot, word) do { if (lane == 0 && (slot) + 4 >= 0) trace[16 * ((slot) + 4) + (word)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RNNT_PD_STAMP(slot, word) do { } while (0)
#endif

    const int rowb_lp = U * 8, rowb_out = U * 4;
    // row (forward diagonal mod T) of the first diagonal of block `lo`; rows advance by one per diagonal
    // (alpha upwards, beta downwards) and wrap modulo T
    const int dF0 = BETA ? (ndiag - 1 - lo * K) : lo * K;
    const int row0 = ((dF0 % T) + T) % T;
    // byte offsets of the K rows of the next block, rowb bytes per row; `row` moves on by K
    auto advance = [&](int& row, const int rowb, int (&offs)[K]) {
        const bool nowrap = BETA ? (row >= K - 1) : (row + K <= T);
        if (nowrap) {                                   // the common case: one multiply, K-1 adds
            const int base = row * rowb;
#pragma unroll
            for (int k = 0; k < K; ++k) offs[k] = BETA ? base - k * rowb : base + k * rowb;
            row = BETA ? row - K : row + K;
            if (BETA) { if (row < 0) row += T; } else { if (row >= T) row -= T; }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                offs[k] = row * rowb;
                row = BETA ? (row == 0 ? T - 1 : row - 1) : (row + 1 == T ? 0 : row + 1);
            }
        }
    };

    // the helper waves' bodies are generic over which half of a block they handle (compile time: no data-dependent
    // branch in their steady-state loops)
    auto loader = [&](auto half_c) {
        constexpr int half = decltype(half_c)::value;
        constexpr int k0 = half * KH;
        // ------------------------------ loader ------------------------------
        const __amdgpu_buffer_rsrc_t rs_lp = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.lp) + nbase * 2, 0, T * U * 8, RSRC_WORD3);
        f32x2 regs[NBR][KH];
        u64 mregs[NBR] = {0, 0, 0};
        constexpr bool does_mail = has_left && half == 0;  // the first loader wave also fetches the boundary column
        int row_ld = row0;
        const bool last_column = u == Un - 1;
        float pmin = 1.0f, pmax = 1.0f;               // range of the probabilities of live cells this wave produced
        const u64* gsrc = has_left ? ring_in + (lane < GRAN ? lane : GRAN) : nullptr;   // lanes >= 17: a pad granule
        (void)gsrc; (void)mregs;
        auto load_block = [&](f32x2 (&dst)[KH]) {
            int offs[K];
            advance(row_ld, rowb_lp, offs);
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const int off = half ? offs[k + KH] : offs[k];
#ifdef RNNT_PD_NOLOAD      // timing probe: no HBM reads
                dst[k] = f32x2{-1.5f - 0.001f * (float)(off & 7), -2.5f};
#else
                dst[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_lp, uc * 8, off, 0));
#endif
            }
        };
        // the left neighbour's boundary column of block lb: requested two blocks ahead, checked when needed
        auto mail_request = [&](const int lb) {
            return __hip_atomic_load(gsrc + (size_t)lb * GPITCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto mail_valid = [&](const int lb, const u64 g) {
            const bool ok = lane >= GRAN || (unsigned)(g >> 32) == block_tag(tag_in, lb);
            return __builtin_amdgcn_ballot_w64(!ok) == 0;
        };
        bool lost = false;     // a wait has timed out: the sweep is flagged for the log-domain kernel, the rest of it
                               // runs on whatever the ring holds without waiting again
        auto mail_wait = [&](const int lb) {               // poll until block lb of the neighbour is there
            for (int spins = 0;; ++spins) {
                const u64 g = mail_request(lb);
                if (lost || mail_valid(lb, g)) return g;
                if (spins >= SPIN_LIMIT) { atomicOr(wg_bad, 2); lost = true; return g; }
                __builtin_amdgcn_s_sleep(8);
            }
        };
        auto mail_stage = [&](const int lb, u64 g) {
            if (!mail_valid(lb, g)) {
                // The request was issued two blocks ago and the producer had not got there yet: this column
                // block has caught up with its neighbour.  Let the neighbour get LAG blocks ahead (or finish)
                // before going on, so that the look-ahead requests of the following blocks find their data --
                // otherwise every block would pay a polling round trip (~1.3 us against ~0.4 us of work).
#ifdef RNNT_PD_STATS
                const u64 t_wait = __builtin_amdgcn_s_memrealtime();
#endif
                mail_wait(min(lb + LAG, hi_left - 1));
                g = mail_wait(lb);
#ifdef RNNT_PD_STATS
                if (lane == 0) trace[16 * (lb + 4) + 1] = __builtin_amdgcn_s_memrealtime() - t_wait;
#endif
            }
            if (lane < GRAN) sm.mail_in[lb & (MSLOTS - 1)][lane] = (unsigned)g;
        };
        // local time p: converts pairs(p) (loaded at p-DLOAD) into LDS and stages the neighbour's block p-1,
        // then issues the loads of pairs(p+DLOAD) and the request for the neighbour's block p+1
        auto l_step = [&](const int p, auto ph, auto guarded) {
            constexpr int PH = decltype(ph)::value;           // p mod NBR
            constexpr bool GUARDED = decltype(guarded)::value;
            RNNT_PD_STAMP(p, 2 + half);
            if (!GUARDED || (p >= lo && p < hi)) {
                float* dst = &sm.probs[p & (PSLOTS - 1)][lane * PSTRIDE + 4 * k0];
                const bool full = !GUARDED || full_block(p);
                const int d0 = p * K + k0;
#pragma unroll
                for (int k = 0; k < KH; ++k) {
                    const f32x2 v = regs[PH][k];
#ifdef RNNT_PD_NOCONV      // timing probe: helper waves without their arithmetic
                    const float pb = __builtin_fmaf(v.x, 0.001f, 0.5f), pl = __builtin_fmaf(v.y, 0.001f, 0.25f);
#else
                    const float pb = __builtin_amdgcn_exp2f(v.x * LOG2E);
                    const float pl = __builtin_amdgcn_exp2f(v.y * LOG2E);
#endif
                    if (full) {
                        // (v_min3/v_max3 skip a NaN operand: a NaN in a live cell is caught by the compute wave's
                        //  final check instead -- it is sticky in the chain.  The label channel of the last column is
                        //  not part of the lattice: whatever it holds is not judged.)
                        const float pl_j = last_column ? pb : pl;
                        asm("v_min3_f32 %0, %0, %1, %2" : "+v"(pmin) : "v"(pb), "v"(pl_j));
                        asm("v_max3_f32 %0, %0, %1, %2" : "+v"(pmax) : "v"(pb), "v"(pl_j));
                    } else {
                        // cells outside the utterance may hold anything: they are never used, never judged;
                        // nor is the label channel of the last column
                        const bool live = (unsigned)(d0 + k - ucol_chk) < (unsigned)Tn;
                        const float pl_j = last_column ? pb : pl;
                        if (live) {
                            pmin = __builtin_fminf(pmin, __builtin_fminf(pb, pl_j));
                            pmax = __builtin_fmaxf(pmax, __builtin_fmaxf(pb, pl_j));
                            if (!(pb == pb) || !(pl_j == pl_j)) pmax = __builtin_inff();   // NaN in a live cell
                        }
                    }
                    f64x2 pr;
#ifdef RNNT_PD_NOCONV
                    pr.x = __builtin_bit_cast(double, ((u64)__builtin_bit_cast(unsigned, pb) << 29) | 0x3fd0000000000000ull);
                    pr.y = __builtin_bit_cast(double, ((u64)__builtin_bit_cast(unsigned, pl) << 29) | 0x3fc0000000000000ull);
#else
                    pr.x = (double)pb;
                    pr.y = (double)pl;
#endif
                    *reinterpret_cast<f64x2*>(dst + 4 * k) = pr;
                }
            }
            if constexpr (does_mail) { if (!GUARDED || (p - 1 >= lo && p - 1 < hi_left)) mail_stage(p - 1, mregs[PH]); }
            if (!GUARDED || (p + DLOAD >= lo && p + DLOAD < hi)) load_block(regs[(PH + DLOAD) % NBR]);
            if constexpr (does_mail) { if (!GUARDED || (p + 1 >= lo && p + 1 < hi_left)) mregs[(PH + DLOAD) % NBR] = mail_request(p + 1); }
            RNNT_PD_STAMP(p, 7 + half);
            block_barrier();
        };
        auto l_any = [&](const int p, auto guarded) {
            const int m = ((p % NBR) + NBR) % NBR;
            if (m == 0) l_step(p, std::integral_constant<int, 0>{}, guarded);
            else if (m == 1) l_step(p, std::integral_constant<int, 1>{}, guarded);
            else l_step(p, std::integral_constant<int, 2>{}, guarded);
        };
        int p = lo - DLOAD;                                // interval index = p - (lo - DLOAD)
        const int p_end = hi + 1;                          // last conversion at hi-1, last staging (block hi-1) at hi
        // steady range: both activities live and every lane live (no predicates), p = 0 (mod NBR) at its start
        int ps0 = max(lo + 1, last_col / K + 1);           // (lo + 1: the neighbour's block p-1 >= lo)
        ps0 += (NBR - ((ps0 % NBR) + NBR) % NBR) % NBR;
        int ps1 = min(hi - DLOAD, (wave_c + Tn) / K);      // exclusive: blocks [.., ps1) are full
        if (has_left) ps1 = min(ps1, hi_left - 1);         // and the neighbour's block p+1 < hi_left
        for (; p < p_end && p < ps0; ++p) l_any(p, std::true_type{});
        for (; p + NBR <= ps1; p += NBR) {
            l_step(p, std::integral_constant<int, 0>{}, std::false_type{});
            l_step(p + 1, std::integral_constant<int, 1>{}, std::false_type{});
            l_step(p + 2, std::integral_constant<int, 2>{}, std::false_type{});
        }
        for (; p < p_end; ++p) l_any(p, std::true_type{});
        // input check: one flag per workgroup, read after the last barrier
        if (colvalid && !(pmin >= P_MIN && pmax <= P_MAX)) atomicOr(wg_bad, 1);   // (LDS atomic: the reasons are bits)
        for (int g = p - (lo - DLOAD); g < G; ++g) block_barrier();
        static_assert(NBR == 3 && PSLOTS == 2 && MSLOTS == 2, "l_step phases are written for these ring depths");
    };
    if (role == 1) {
        if (half_rt == 0) loader(std::integral_constant<int, 0>{});
        else loader(std::integral_constant<int, 1>{});
        return;
    }

    auto storer = [&](auto half_c) {
        constexpr int half = decltype(half_c)::value;
        constexpr int k0 = half * KH;
        // ------------------------------ storer ------------------------------
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, T * U * 4, RSRC_WORD3);
        const int voff_out = colvalid ? uc * 4 : OOB;
        int row_st = row0;
        // local time p: stores values(p-3) and publishes the boundary column of block p-3, both written by the
        // compute wave during p-1
        auto s_step = [&](const int p, auto guarded) {
            constexpr bool GUARDED = decltype(guarded)::value;
            const int ps = p - 3;
            RNNT_PD_STAMP(p, 4 + half);
            if (!GUARDED || (ps >= lo && ps < hi)) {
                if (has_right && half == 0 && lane < GRAN) {      // (has_right: compile time)
                    const u64 g = ((u64)block_tag(tag_out, ps) << 32) | sm.mail_out[ps & (MSLOTS - 1)][lane];
                    __hip_atomic_store(ring_out + (size_t)ps * GPITCH + lane, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const float* src = &sm.vals[ps & (VSLOTS - 1)][lane * VSTRIDE + 2 * k0];
                const int eb = sm.exps[ps & (VSLOTS - 1)][k0 / KR][lane] - 1023;
                int offs[K];
                advance(row_st, rowb_out, offs);
                const int d0 = ps * K + k0;
                const bool full = !GUARDED || full_block(ps);
#pragma unroll
                for (int k = 0; k < KH; k += 2) {
                    const f32x4 two = *reinterpret_cast<const f32x4*>(src + 2 * k);   // values k and k+1
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const unsigned vlo = __builtin_bit_cast(unsigned, j ? two.z : two.x);
                        const unsigned vhi = __builtin_bit_cast(unsigned, j ? two.w : two.y);
                        // top 24 significant bits -> fp32 in [1,2): mantissa = (vhi[19:0] << 3) | vlo[31:29]
                        const unsigned mant = __builtin_amdgcn_alignbit(vhi, vlo, 29);
                        const float m = __builtin_bit_cast(float, (mant & 0x007fffffu) | 0x3f800000u);
                        const float l2 = __builtin_amdgcn_logf(m);
                        const int ex = (int)((vhi >> 20) & 0x7ffu) + eb;
#ifdef RNNT_PD_NOCONV
                        const float res = __builtin_bit_cast(float, vhi ^ vlo) + (float)eb;
#else
                        const float res = __builtin_fmaf((float)ex, LN2, l2 * LN2);
#endif
                        int voff = voff_out;
                        if (!full) {
                            const bool live = (unsigned)(d0 + k + j - ucol_chk) < (unsigned)Tn;
                            voff = live ? voff_out : OOB;
                        }
#ifdef RNNT_PD_NOSTORE     // timing probe: no HBM writes (one comparison keeps the conversion alive)
                        if (res == 123.456f)
#endif
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, res), rs_out, voff,
                                                              half ? offs[k + j + KH] : offs[k + j], 0);
                    }
                }
            }
            RNNT_PD_STAMP(p, 9 + half);
            block_barrier();
        };
        int g = 0;
        for (; g < 3 + DLOAD; ++g) block_barrier();         // values of block lo exist after interval 4
        int p = lo + 3;
        const int p_end = hi + 3;
        const int ps0 = max(lo, last_col / K + 1) + 3;
        const int ps1 = min(hi, (wave_c + Tn) / K) + 3;    // exclusive
        for (; p < p_end && p < ps0; ++p) s_step(p, std::true_type{});
        for (; p < ps1; ++p) s_step(p, std::false_type{});
        for (; p < p_end; ++p) s_step(p, std::true_type{});
        // 3+DLOAD + (hi-lo) = G barriers
    };
    if (role == 2) {
        if (half_rt == 0) storer(std::integral_constant<int, 0>{});
        else storer(std::integral_constant<int, 1>{});
        return;
    }

    // ------------------------------ compute wave ------------------------------
    double Y = (ucol == 0) ? 1.0 : 0.0;
    double X = 0.0;
    int E = 0;
    int gap_hi = 0, gap_lo = 0;      // extremes of E_left - E_own over the renormalisations of this column block
    Cell2 bufA[K], bufB[K];
    auto do_block = [&](const int lb, const Cell2 (&cur)[K], Cell2 (&nxt)[K]) {
        const int d0 = lb * K;
        RNNT_PD_STAMP(lb, 0);
        double seed[K];
        int e_mail0 = 0, e_mail1 = 0;
        if constexpr (has_left) {
            // the left neighbour's boundary column, staged by the loader one interval ago (every lane reads it -- one
            // broadcast LDS access per pair -- only lane 0 uses it)
            const f64x2* mp = reinterpret_cast<const f64x2*>(&sm.mail_in[lb & (MSLOTS - 1)][0]);
#pragma unroll
            for (int k = 0; k < K; k += 2) {
                const f64x2 two = mp[k >> 1];
                seed[k] = two.x; seed[k + 1] = two.y;
            }
            e_mail0 = (int)sm.mail_in[lb & (MSLOTS - 1)][2 * K];
            e_mail1 = (int)sm.mail_in[lb & (MSLOTS - 1)][2 * K + NH - 1];
        }
        const float* next_probs = &sm.probs[(lb + 1) & (PSLOTS - 1)][lane * PSTRIDE];
        float* vdst = &sm.vals[lb & (VSLOTS - 1)][lane * VSTRIDE];
        int* edst = &sm.exps[lb & (VSLOTS - 1)][0][lane];
        double* xdst = (lane == WAVE - 1) ? reinterpret_cast<double*>(&sm.mail_out[lb & (MSLOTS - 1)][0]) : &sm.dumpx[lane][0];
        int* email_dst = (lane == WAVE - 1) ? reinterpret_cast<int*>(&sm.mail_out[lb & (MSLOTS - 1)][2 * K]) : &sm.dumpe[lane];
        const bool full = full_block(lb);
#define RNNT_PD_CALL(MASKED, SEED, MAIL)                                                                          \
    compute_block<BETA, MASKED, SEED, MAIL>(cur, nxt, seed, e_mail0, e_mail1, Y, X, E, d0, ucol_chk, Tn, wave_c, \
                                            next_probs, vdst, xdst, edst, email_dst, gap_hi, gap_lo)
        if (full) RNNT_PD_CALL(false, HAS_LEFT, HAS_RIGHT);
        else RNNT_PD_CALL(true, HAS_LEFT, HAS_RIGHT);
#undef RNNT_PD_CALL
        RNNT_PD_STAMP(lb, 6);
        block_barrier();
    };
    int g = 0;
    for (; g < 1 + DLOAD; ++g) block_barrier();            // the loader converts block lo during interval DLOAD
    {
        const float* src = &sm.probs[lo & (PSLOTS - 1)][lane * PSTRIDE];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const f64x2 pr = *reinterpret_cast<const f64x2*>(src + 4 * k);
            bufA[k].b = pr.x; bufA[k].l = pr.y;
        }
    }
    block_barrier(); ++g;
    int lb = lo;
    for (; lb + 2 <= hi; lb += 2) {
        do_block(lb, bufA, bufB);
        do_block(lb + 1, bufB, bufA);
    }
    if (lb < hi) { do_block(lb, bufA, bufB); ++lb; }
    // Range check of the chain, after the fact (nothing in the steady-state loop): a column boundary wider than the
    // fp64 state is trusted with, or a final value that is not a positive finite number -- an overflowed factor,
    // a NaN or a total underflow anywhere upstream is sticky in Y (every weight is a non-negative probability, nothing
    // ever subtracts) -- hands the sweep to the log-domain kernel.  Lanes beyond the last column carry garbage by
    // design and are not judged.
    if (colvalid && (gap_hi > MAX_GAP || gap_lo < -MAX_GAP || !(Y > 0.0 && Y < __builtin_inf()))) atomicOr(wg_bad, 1);
    for (g = (lb - lo) + 2 + DLOAD; g < G; ++g) block_barrier();
    if constexpr (!BETA) {
        // the finished last column holds (alpha * pB)(T-1,U-1) in Y: the alpha-side log-likelihood
        // (core_gather.cu:339); once per utterance, so libm's fp64 log2 is affordable
        if (ucol == Un - 1) a.ll[n] = (float)((::log2(Y) + (double)E) * 0.693147180559945309417);
    }
}

template <bool COMPACT>
__global__ void __launch_bounds__(5 * WAVE) k_lattice_pd(LatticeArgs a, const int nA) {
    __shared__ Smem sm;
    __shared__ int wg_bad, s_item;
    // launch epoch = host counter (constant across the replays of a captured graph) + device counter (queue[1]: the
    // library's own per-device launch counter, bumped by k_prepare in front of every launch, replayed or not)
    a.epoch += (unsigned)a.queue[1];
    // work items in column-block-major order from an atomic counter: the workgroup that holds item i knows that
    // every item < i -- in particular its left neighbour, item i - 2N -- is held by a workgroup that has started
    if (threadIdx.x == 0) { s_item = atomicAdd(a.queue, 1); wg_bad = 0; }
    __syncthreads();
    const int sweeps = gridDim.x / nA;                 // 2N
    Item it;
    it.cb = s_item / sweeps;
    const int s = s_item - it.cb * sweeps;
    it.n = s >> 1;
    it.dir = s & 1;
    const UttLens len = utt_lens<COMPACT>(a.xn, a.yn, it.n, a.T, a.U);
    if (!COMPACT || len.ok) {   // (compact: an utterance with bad lengths has no plane of its own to sweep)
        const bool hl = it.cb > 0, hr = it.cb + 1 < (len.Un + WAVE - 1) / WAVE;
#define RNNT_PD_SWEEP(B)                                                                    \
    do {                                                                                    \
        if (hl) { if (hr) sweep<B, COMPACT, true, true>(a, it, len, nA, sm, &wg_bad);       \
                  else sweep<B, COMPACT, true, false>(a, it, len, nA, sm, &wg_bad); }       \
        else { if (hr) sweep<B, COMPACT, false, true>(a, it, len, nA, sm, &wg_bad);         \
               else sweep<B, COMPACT, false, false>(a, it, len, nA, sm, &wg_bad); }         \
    } while (0)
        if (it.dir) RNNT_PD_SWEEP(true); else RNNT_PD_SWEEP(false);
#undef RNNT_PD_SWEEP
    }
    __syncthreads();
    if (threadIdx.x == 0 && wg_bad) atomicOr(&a.redo[2 * it.n + it.dir], wg_bad);
}

// The library's launch counter of this device: module-scope device memory (zero when the code object is loaded,
// never part of anybody's workspace), so it cannot be recycled, scribbled over or left uninitialised, and every
// REPLAY of a captured graph advances it too -- kernel arguments are frozen at capture time, and with a frozen
// epoch the granules of the previous replay would carry this replay's tags.
__device__ unsigned g_launch_counter;

// In front of every launch: clears the redo flags and the queue head (n words), hands the next value of the launch
// counter to the kernel (p[n]) and zeroes the hand-over rings (mail_vec 16-byte words; tag 0 never validates), so
// that nothing the workspace held before -- it is caller scratch with unspecified contents -- can be taken for a
// granule of this launch.  The clear is 2 % of the bytes the sweeps move and rides on a launch that existed anyway.
__global__ void __launch_bounds__(256) k_prepare(int* p, int n, uint4* mail, size_t mail_vec) {
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < n; i += 256) p[i] = 0;
        if (threadIdx.x == 0) p[n] = (int)(atomicAdd(&g_launch_counter, 1u) + 1u);
    }
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < mail_vec; i += (size_t)gridDim.x * 256) mail[i] = z;
}

}  // namespace pd

size_t pd_mail_blocks(int T, int U) { return (size_t)(T + U - 1 + pd::K - 1) / pd::K + 1; }

size_t pd_mail_bytes(int N, int T, int U) {
    const int nA = (U + WAVE - 1) / WAVE;
    if (nA < 2 || !pd_shape_supported(T, U)) return 0;     // one column block, or a shape the kernel never takes
    size_t bytes = (size_t)2 * N * (nA - 1) * pd_mail_blocks(T, U) * pd::GPITCH * sizeof(pd::u64);
#ifdef RNNT_PD_STATS       // the stamp table of the diagnostics build, behind the rings
    bytes += (size_t)2 * N * nA * (pd_mail_blocks(T, U) + 8) * sizeof(pd::u64) * 16;
#endif
    return bytes;
}

unsigned next_launch_epoch() {
    // granules of earlier launches (same buffer) never validate.  Random start so that a recycled allocation of
    // another process does not either; the device-side counter (k_prepare) is added in the kernel.
    static std::atomic<unsigned> epoch{std::random_device{}()};
    return epoch.fetch_add(1, std::memory_order_relaxed) + 1;
}

hipError_t launch_ring_prepare(hipStream_t stream, int* flags, int n_flags, void* rings, size_t ring_bytes) {
    // the flags (2N ints) and the queue head are contiguous in the workspace (api.hip: carve).  One tiny kernel:
    // hipMemsetAsync of these few bytes becomes two fill kernels of ~5 us each.
    const size_t mail_vec = ring_bytes / 16;
    const unsigned prep_blocks = (unsigned)std::min<size_t>(512, std::max<size_t>(1, mail_vec / (256 * 8)));
    pd::k_prepare<<<prep_blocks, 256, 0, stream>>>(flags, n_flags, reinterpret_cast<uint4*>(rings), mail_vec);
    return hipGetLastError();
}

// Needs a.redo, a.queue = a.redo + 2N with the launch counter behind it (and a.mail when U > 64); zeroes redo and
// the queue head itself.
hipError_t launch_lattice_pd(hipStream_t stream, const LatticeArgs& a0, int N) {
    if (N <= 0) return hipSuccess;
    const int nA = (a0.U + WAVE - 1) / WAVE;
    if (!a0.redo || !a0.queue || (nA > 1 && !a0.mail)) return hipErrorNotSupported;
    if ((long long)2 * N * nA >= (1ll << 31)) return hipErrorNotSupported;
    LatticeArgs a = a0;
    a.epoch = next_launch_epoch();
    a.mail_blocks = (int)pd_mail_blocks(a.T, a.U);
    const size_t ring_bytes = nA > 1 ? (size_t)2 * N * (nA - 1) * a.mail_blocks * pd::GPITCH * sizeof(pd::u64) : 0;
    hipError_t e = launch_ring_prepare(stream, a.redo, 2 * N + 1, a.mail, ring_bytes);
    if (e != hipSuccess) return e;
    const dim3 grid(2 * N * nA), block(5 * WAVE);
    // (Forcing one workgroup per CU -- by claiming more than half of a CU's LDS -- was measured: no difference at
    // 160 workgroups, the dispatcher spreads them already.)
    if (a.offs)
        pd::k_lattice_pd<true><<<grid, block, 0, stream>>>(a, nA);
    else
        pd::k_lattice_pd<false><<<grid, block, 0, stream>>>(a, nA);
    return hipGetLastError();
}

}  // namespace rnnt
