// Probability-domain alpha / beta lattice sweep for MI355X (gfx950), diagonal-major layout.
//
// Same anti-diagonal schedule as lattice_ws.hip (read its header first: lanes are lattice columns, a
// column block of 64 runs one block of K=8 diagonals behind its left neighbour, one s_barrier per block,
// the boundary column is handed over through LDS).  What changes is the ARITHMETIC of the serial chain.
// The log-domain step  lse(a,b) = max + log1p(exp(-|a-b|))  costs the computing wave 16 instructions per
// diagonal (sub, mul, v_exp_f32, add, v_log_f32, three for the rounding correction, max, adds, DPP ...),
// and a lone wave issues one instruction every ~5 cycles whatever it is -- the sweep is bound by that
// count.  Here the recurrence runs on PROBABILITIES with a per-column binary exponent:
//
//     true alpha of column u on the current diagonal = A[u] * 2^E[u]          A fp64, E int32
//     per diagonal      val = fma(A_left, 2^(E_left - E_own), Y)              the factor is exact
//                       Y   = val * pB(cell)      X = val * pL(cell)           (beta: mirrored lattice, weights of
//                                                                               the receiving cell, see step())
//     every K diagonals Y and X are rescaled by the exponent of Y (exact) and E takes it up
//
//   = two 32-bit DPP moves and four fp64 operations per diagonal.  fp64 state because its exponent range
//   (2^+-1022) cannot be left within K steps of fp32-representable probabilities (>= 2^-126 each): no range
//   bookkeeping inside the chain, and columns may differ by hundreds of binary orders of magnitude (they do:
//   tests/pd_model.py).  The conversions live on two helper waves per column block, off the chain:
//     LOADER  HBM -> registers -> p = exp2(lp*log2e) (v_mul_f32, v_exp_f32, v_cvt_f64_f32) -> LDS, two blocks
//             ahead; also the input check: a log-prob outside [-80, 80] (or -inf) cannot be represented as
//             an fp32 probability -- the whole (utterance, direction) is then redone by the log-domain kernel
//             (lattice_ws.hip, launched behind this one; it returns at once when no flag is set);
//     STORER  LDS -> ln2*(log2(mantissa) + exponent) -> HBM, one block behind: the top 24 significant bits of
//             the fp64 value relabelled as an fp32 in [1,2) (v_alignbit_b32, v_bfi_b32), v_log_f32, and one
//             multiply + one fma: the stored alpha/beta carry <= 0.5 ulp of rounding plus 2^-23 relative --
//             no error accumulates along the sweep (the log-domain chain rounds at |alpha| ~ 6e3 every step:
//             1e-2 on the gradients at T=1500,U=300, against 8e-4 here; tests/test_pd_model.py).
//   tests/pd_model.py is the executable statement of this arithmetic (checked on the CPU against fp64).
//
// Reference counterpart: core_gather.cu:37-133 (alphas), :135-234 (betas) -- 32x1 warp tiles ordered by
// global spin locks, lse per cell.  Outputs are the same quantities (log alpha, log beta, fp32).
#include <atomic>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace rnnt {

namespace pd {

constexpr int K = 8;             // diagonals per block (= renormalisation interval)
constexpr int MAXA = 5;          // column blocks per workgroup: 3 waves each -> 15 waves, 320 columns per pass
constexpr int PSLOTS = 2;        // LDS ring of probability blocks
constexpr int VSLOTS = 2;        // LDS ring of value blocks
constexpr int MSLOTS = 2;        // LDS ring of boundary-column blocks
constexpr int DLOAD = 2;         // the loader issues its HBM loads this many blocks before it converts them
constexpr int NBR = DLOAD + 1;   // its register ring
constexpr int SHIFT = DLOAD;     // global block g = local time + idx + SHIFT, so the first load is at g >= 0
constexpr int PSTRIDE = 36;      // dwords per lane per probability slot: 8 cells x (pB,pL) fp64 = 32, +4: the
                                 // 16-lane groups of ds_read/write_b128 then cover all 64 banks (MI355X_MICROARCH.md)
constexpr int VSTRIDE = 20;      // dwords per lane per value slot: 8 fp64 = 16, +4 (same reason)
constexpr int RSRC_WORD3 = 0x00020000;
constexpr int OOB = (int)0x80000000;
constexpr float LP_ABS_MAX = 80.0f;   // |log-prob| beyond this: exp() leaves fp32 -> log-domain kernel
constexpr float LOG2E = 1.44269504088896340736f;
constexpr float LN2 = 0.693147180559945309417f;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

struct alignas(16) Smem {   // per column block
    float probs[PSLOTS][WAVE * PSTRIDE];    // [lane][k] (pB, pL) fp64
    float vals[VSLOTS][WAVE * VSTRIDE];     // [lane][k] fp64 value, scale 2^exps[lane]
    int exps[VSLOTS][WAVE];
    double mailx[MSLOTS][K];                // lane 63's X on entry to step k of the block
    double zeros[K];                        // what every lane but lane 0 reads instead of the mailbox
    int maile[MSLOTS][4];                   // [0]: lane 63's exponent of that block
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
    const unsigned long long b = __builtin_bit_cast(unsigned long long, src);
    const int lo = __builtin_amdgcn_mov_dpp((int)b, 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// Lane i receives src from lane i-1; lane 0 receives `first`.
__device__ __forceinline__ int wave_shr1_i32(int first, int src) {
    return __builtin_amdgcn_update_dpp(first, src, 0x138, 0xf, 0xf, false);
}

struct Cell2 { double b, l; };   // blank / label probability of one lattice cell

// One block of the compute wave: renormalisation + K diagonals.  State: Y, X (fp64), E (binary exponent).
//   SEED:   the column block has a left neighbour: lane 0 receives seed[k] (scale 2^e_mail) instead of nothing
//   MASKED: some lane starts or finishes inside the block (updates are predicated, E of lanes that have not
//           started follows the last started lane)
//   MAIL:   lane 63's X on entry to every step is recorded for the right neighbour
template <bool BETA, bool MASKED, bool SEED, bool MAIL>
__device__ __forceinline__ void compute_block(const Cell2 (&cur)[K], Cell2 (&nxt)[K], const double (&seed)[K], const int e_mail,
                                              double& Y, double& X, int& E, const int d0, const int ucol_chk,
                                              const int Tn, const int front, const bool started_all,
                                              const bool lane_started, const float* next_probs, float* vdst,
                                              double* xdst, int* edst, int* email_dst) {
    // ---- renormalisation (exact: powers of two) ----
    const int e = __builtin_amdgcn_frexp_exp(Y);          // 0 for Y == 0
    Y = __builtin_amdgcn_frexp_mant(Y);
    X = __builtin_ldexp(X, -e);
    E += e;
    if constexpr (MASKED) {
        // lanes that have not started adopt the exponent of the last started column (their first value
        // arrives from it), or the mailbox's when the whole column block has not started
        int ef;
        if (front >= 0) ef = __builtin_amdgcn_readlane(E, front > WAVE - 1 ? WAVE - 1 : front);
        else ef = SEED ? e_mail : __builtin_amdgcn_readlane(E, 0);
        if (!started_all) E = lane_started ? E : ef;
    }
    *edst = E;                                            // the storer's scale for this block's values
    if constexpr (MAIL) *email_dst = E;                   // (lane 63's pointer; the others aim at a dump slot)
    const int e_left = wave_shr1_i32(SEED ? e_mail : E, E);
    const double c = __builtin_ldexp(1.0, e_left - E);    // 2^(E_left - E_own); lane 0 of the first block: 1
    double vprev = 0.0, xprev = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double xin = X;                             // what the right neighbour reads at this step
        const double xl = wave_shr1_f64(X);
        double val;
        if constexpr (BETA) {
            // beta[t,u] = beta[t+1,u]*pB[t,u] + beta[t,u+1]*pL[t,u]: both weights belong to the receiving cell
            const double w = cur[k].l * c;
            val = __builtin_fma(xl, w, Y * cur[k].b);
            if constexpr (SEED) val = __builtin_fma(seed[k], w, val);
        } else {
            // alpha[t,u] = alpha[t-1,u]*pB[t-1,u] + alpha[t,u-1]*pL[t,u-1]: Y and X carry the products
            val = __builtin_fma(xl, c, Y);
            if constexpr (SEED) val = __builtin_fma(seed[k], c, val);
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
        // one cell of the next block per step (written by the loader one block ago, needed one block from
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

template <bool BETA, bool COMPACT>
__device__ __forceinline__ void sweep(const LatticeArgs& a, const int n, Smem* smem, int* wg_bad) {
    const UttLens len = utt_lens<COMPACT>(a.xn, a.yn, n, a.T, a.U);
    if (COMPACT && !len.ok) return;   // no plane of its own to sweep (uniform over the workgroup, before any barrier)
    const int Tn = len.Tn, Un = len.Un;
    const int T = COMPACT ? Tn : a.T, U = COMPACT ? Un : a.U;
    const int lane = threadIdx.x & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (Un == 1) {   // no labels: prefix / suffix sums by one wave (uniform over the workgroup, no barrier yet)
        if (w == 0) {
            const size_t nb1 = COMPACT ? (size_t)a.offs[n] : (size_t)n * T * U;
            const float2* lp2 = reinterpret_cast<const float2*>(a.lp) + nb1;
            const float total = single_column_scan<BETA>(Tn, (BETA ? a.betas : a.alphas) + nb1, U, lane,
                                                         [&](int t) { return lp2[(size_t)t * U].x; });
            if (!BETA && lane == 0) a.ll[n] = total;
        }
        return;
    }
    const int nA = blockDim.x / (3 * WAVE);           // column blocks
    const int role = w / nA;                          // 0 compute, 1 loader, 2 storer
    const int idx = w - role * nA;                    // column block
    const size_t nbase = COMPACT ? (size_t)a.offs[n] : (size_t)n * T * U;
    float* out = (BETA ? a.betas : a.alphas) + nbase;
    const int ndiag = Tn + Un - 1;
    Smem& sm = smem[idx];

    const int ucol = WAVE * idx + lane;               // column in sweep coordinates (one pass: Un <= 64*nA)
    const bool colvalid = ucol < Un;
    const int u = BETA ? (Un - 1 - ucol) : ucol;
    const int uc = min(max(u, 0), U - 1);
    const int ucol_chk = colvalid ? ucol : 0x40000000;
    const int nwa = min(nA, (Un + WAVE - 1) / WAVE);  // column blocks with a live column
    const int wave_c = WAVE * idx;
    const int lo = wave_c / K;
    const int hi = (min(ndiag, Tn + wave_c + WAVE) + K - 1) / K;
    const bool live_blk = (idx < nwa) && (lo < hi);
    const int hi_last = (min(ndiag, Tn + WAVE * (nwa - 1) + WAVE) + K - 1) / K;
    const int G = hi_last + (nwa - 1) + 3 + SHIFT;    // barriers every wave executes
    // every lane live for the whole block?  (all started, none finished, all 64 columns inside the lattice)
    auto full_block = [&](const int lb) {
        const int d0 = lb * K;
        return (d0 >= wave_c + WAVE - 1) && (d0 + K <= wave_c + Tn) && (wave_c + WAVE <= Un);
    };

    if (!live_blk) {
        for (int g = 0; g < G; ++g) block_barrier();
        return;
    }
    const int rowb_lp = U * 8, rowb_out = U * 4;
    // row (forward diagonal mod T) of the first diagonal of block `lo`, and how rows advance
    const int dF0 = BETA ? (ndiag - 1 - lo * K) : lo * K;
    const int row0 = ((dF0 % T) + T) % T;
    auto advance = [&](int& row, int (&rows)[K]) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            rows[k] = row;
            row = BETA ? (row == 0 ? T - 1 : row - 1) : (row + 1 == T ? 0 : row + 1);
        }
    };

    if (role == 1) {
        // ------------------------------ loader ------------------------------
        const __amdgpu_buffer_rsrc_t rs_lp = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.lp) + nbase * 2, 0, T * U * 8, RSRC_WORD3);
        f32x2 regs[NBR][K];
        int row_ld = row0;
        float amax = 0.0f;                            // max |log-prob| over the live cells this wave converted
        auto load_block = [&](f32x2 (&dst)[K]) {
            int rows[K];
            advance(row_ld, rows);
#pragma unroll
            for (int k = 0; k < K; ++k) {
#ifdef RNNT_PD_NOLOAD      // timing probe: no HBM reads
                dst[k] = f32x2{-1.5f - 0.001f * (float)(rows[k] & 7), -2.5f};
#else
                dst[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_lp, uc * 8, rows[k] * rowb_lp, 0));
#endif
            }
        };
        // local time p: converts pairs(p) (loaded at p-DLOAD) into LDS, then issues the loads of pairs(p+DLOAD)
        auto l_step = [&](const int p, auto ph, auto guarded) {
            constexpr int PH = decltype(ph)::value;           // p mod NBR
            constexpr bool GUARDED = decltype(guarded)::value;
            if (!GUARDED || (p >= lo && p < hi)) {
                float* dst = &sm.probs[p & (PSLOTS - 1)][lane * PSTRIDE];
                const bool full = !GUARDED || full_block(p);
                const int d0 = p * K;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const f32x2 v = regs[PH][k];
                    const float m = __builtin_fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y));
                    if (full) {
                        amax = __builtin_fmaxf(amax, m);
                    } else {
                        // cells outside the utterance may hold anything: they are never used, never judged.
                        // the label channel of the last column is not part of the lattice either
                        const bool live = (unsigned)(d0 + k - ucol_chk) < (unsigned)Tn;
                        const float mm = (u == Un - 1) ? __builtin_fabsf(v.x) : m;
                        amax = live ? __builtin_fmaxf(amax, mm) : amax;
                    }
                    f64x2 pr;
                    pr.x = (double)__builtin_amdgcn_exp2f(v.x * LOG2E);
                    pr.y = (double)__builtin_amdgcn_exp2f(v.y * LOG2E);
                    *reinterpret_cast<f64x2*>(dst + 4 * k) = pr;
                }
            }
            if (!GUARDED || (p + DLOAD >= lo && p + DLOAD < hi)) load_block(regs[(PH + DLOAD) % NBR]);
            block_barrier();
        };
        auto l_any = [&](const int p, auto guarded) {
            const int m = ((p % NBR) + NBR) % NBR;
            if (m == 0) l_step(p, std::integral_constant<int, 0>{}, guarded);
            else if (m == 1) l_step(p, std::integral_constant<int, 1>{}, guarded);
            else l_step(p, std::integral_constant<int, 2>{}, guarded);
        };
        const int p_first = lo - DLOAD;
        int g = 0;
        for (; g < p_first + idx + SHIFT; ++g) block_barrier();
        int p = p_first;
        const int p_end = hi;                              // last conversion at p = hi-1
        // steady range: both activities live and every lane live (no predicates), p = 0 (mod NBR) at its start
        int ps0 = max(lo, (wave_c + WAVE - 1 + K - 1) / K);
        ps0 += (NBR - ((ps0 % NBR) + NBR) % NBR) % NBR;
        int ps1 = min(hi - DLOAD, (wave_c + Tn) / K);      // exclusive: blocks [.., ps1) are full
        if (wave_c + WAVE > Un) ps1 = p_first;             // a column block with columns outside the lattice is never "full"
        for (; p < p_end && p < ps0; ++p) l_any(p, std::true_type{});
        for (; p + NBR <= ps1; p += NBR) {
            l_step(p, std::integral_constant<int, 0>{}, std::false_type{});
            l_step(p + 1, std::integral_constant<int, 1>{}, std::false_type{});
            l_step(p + 2, std::integral_constant<int, 2>{}, std::false_type{});
        }
        for (; p < p_end; ++p) l_any(p, std::true_type{});
        // input check: one flag per workgroup, read after the last barrier
        if (!(amax <= LP_ABS_MAX)) *wg_bad = 1;            // benign race: every writer stores 1
        for (g = p + idx + SHIFT; g < G; ++g) block_barrier();
        static_assert(NBR == 3 && PSLOTS == 2, "l_step phases are written for a 3-deep register ring");
        return;
    }

    if (role == 2) {
        // ------------------------------ storer ------------------------------
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, T * U * 4, RSRC_WORD3);
        const int voff_out = colvalid ? uc * 4 : OOB;
        int row_st = row0;
        // local time p: stores values(p-3), which the compute wave wrote during p-1
        auto s_step = [&](const int p, auto guarded) {
            constexpr bool GUARDED = decltype(guarded)::value;
            const int ps = p - 3;
            if (!GUARDED || (ps >= lo && ps < hi)) {
                const float* src = &sm.vals[ps & (VSLOTS - 1)][lane * VSTRIDE];
                const int eb = sm.exps[ps & (VSLOTS - 1)][lane];
                int rows[K];
                advance(row_st, rows);
                const int d0 = ps * K;
                const bool full = !GUARDED || full_block(ps);
#pragma unroll
                for (int k = 0; k < K; k += 2) {
                    const f32x4 two = *reinterpret_cast<const f32x4*>(src + 2 * k);   // values k and k+1
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const unsigned vlo = __builtin_bit_cast(unsigned, j ? two.z : two.x);
                        const unsigned vhi = __builtin_bit_cast(unsigned, j ? two.w : two.y);
                        // top 24 significant bits -> fp32 in [1,2): mantissa = (vhi[19:0] << 3) | vlo[31:29]
                        const unsigned mant = __builtin_amdgcn_alignbit(vhi, vlo, 29);
                        const float m = __builtin_bit_cast(float, (mant & 0x007fffffu) | 0x3f800000u);
                        const float l2 = __builtin_amdgcn_logf(m);
                        const int ex = (int)((vhi >> 20) & 0x7ffu) - 1023 + eb;
                        const float res = __builtin_fmaf((float)ex, LN2, l2 * LN2);
                        int voff = voff_out;
                        if (!full) {
                            const bool live = (unsigned)(d0 + k + j - ucol_chk) < (unsigned)Tn;
                            voff = live ? voff_out : OOB;
                        }
#ifdef RNNT_PD_NOSTORE     // timing probe: no HBM writes (one lane keeps the conversion alive)
                        if (res == 123.456f)
#endif
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, res), rs_out, voff,
                                                              rows[k + j] * rowb_out, 0);
                    }
                }
            }
            block_barrier();
        };
        const int p_first = lo + 3;
        int g = 0;
        for (; g < p_first + idx + SHIFT; ++g) block_barrier();
        int p = p_first;
        const int p_end = hi + 3;
        int ps0 = max(lo, (wave_c + WAVE - 1 + K - 1) / K) + 3;
        int ps1 = min(hi, (wave_c + Tn) / K) + 3;          // exclusive
        if (wave_c + WAVE > Un) ps1 = p_first;             // never "full": no unguarded range
        for (; p < p_end && p < ps0; ++p) s_step(p, std::true_type{});
        for (; p < ps1; ++p) s_step(p, std::false_type{});
        for (; p < p_end; ++p) s_step(p, std::true_type{});
        for (g = p + idx + SHIFT; g < G; ++g) block_barrier();
        return;
    }

    // ------------------------------ compute wave ------------------------------
    double Y = (ucol == 0) ? 1.0 : 0.0;
    double X = 0.0;
    int E = 0;
    Cell2 bufA[K], bufB[K];
    // lane 0 reads the left neighbour's boundary column, every other lane reads zeros: the seed term of the
    // step then needs no predicate
    const Smem& left = smem[idx > 0 ? idx - 1 : 0];
    const bool has_right = idx + 1 < nwa;
    auto do_block = [&](const int lb, const Cell2 (&cur)[K], Cell2 (&nxt)[K]) {
        const int d0 = lb * K;
        double seed[K];
        int e_mail = 0;
        if (idx > 0) {
            const double* mp = (lane == 0) ? &left.mailx[lb & (MSLOTS - 1)][0] : &left.zeros[0];
#pragma unroll
            for (int k = 0; k < K; ++k) seed[k] = mp[k];
            e_mail = left.maile[lb & (MSLOTS - 1)][0];
        }
        const float* next_probs = &sm.probs[(lb + 1) & (PSLOTS - 1)][lane * PSTRIDE];
        float* vdst = &sm.vals[lb & (VSLOTS - 1)][lane * VSTRIDE];
        int* edst = &sm.exps[lb & (VSLOTS - 1)][lane];
        double* xdst = (lane == WAVE - 1) ? &sm.mailx[lb & (MSLOTS - 1)][0] : &sm.dumpx[lane][0];
        int* email_dst = (lane == WAVE - 1) ? &sm.maile[lb & (MSLOTS - 1)][0] : &sm.dumpe[lane];
        const bool full = full_block(lb);
        const int front = d0 - 1 - wave_c;                 // last lane that has started (may be < 0 or > 63)
        const bool lane_started = (ucol <= d0 - 1) || (ucol == 0);
        const bool started_all = front >= WAVE - 1;
#define RNNT_PD_CALL(MASKED, SEED, MAIL)                                                                      \
    compute_block<BETA, MASKED, SEED, MAIL>(cur, nxt, seed, e_mail, Y, X, E, d0, ucol_chk, Tn, front, started_all, \
                                            lane_started, next_probs, vdst, xdst, edst, email_dst)
        if (idx > 0) {
            if (full) { if (has_right) RNNT_PD_CALL(false, true, true); else RNNT_PD_CALL(false, true, false); }
            else { if (has_right) RNNT_PD_CALL(true, true, true); else RNNT_PD_CALL(true, true, false); }
        } else {
            if (full) { if (has_right) RNNT_PD_CALL(false, false, true); else RNNT_PD_CALL(false, false, false); }
            else { if (has_right) RNNT_PD_CALL(true, false, true); else RNNT_PD_CALL(true, false, false); }
        }
#undef RNNT_PD_CALL
        block_barrier();
    };
    if (lane < K) sm.zeros[lane] = 0.0;                    // before the first barrier this wave takes part in
    int g = 0;
    for (; g < lo + idx + 1 + SHIFT; ++g) block_barrier();
    {   // first block: written by the loader during the previous interval
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
    for (g = lb + idx + 2 + SHIFT; g < G; ++g) block_barrier();
    if constexpr (!BETA) {
        // the finished last column holds (alpha * pB)(T-1,U-1) in Y: the alpha-side log-likelihood
        // (core_gather.cu:339); once per utterance, so libm's fp64 log2 is affordable
        if (ucol == Un - 1) a.ll[n] = (float)((::log2(Y) + (double)E) * 0.693147180559945309417);
    }
}

template <bool COMPACT>
__global__ void __launch_bounds__(3 * MAXA * WAVE) k_lattice_pd(const LatticeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem* smem = reinterpret_cast<Smem*>(smem_raw);
    __shared__ int wg_bad;
    // same XCD-aware (utterance, direction) -> workgroup map as k_lattice_ws: the two sweeps of an utterance
    // read the same plane from opposite ends and share an XCD's L2 (speed only)
    const unsigned b = blockIdx.x, pairs_total = gridDim.x >> 1;
    const unsigned grp = b >> 4, in = b & 15;
    unsigned n, dir;
    if ((grp << 3) + 8 <= pairs_total) { n = (grp << 3) + (in & 7); dir = in >> 3; }
    else { const unsigned r = b - (grp << 4); n = (grp << 3) + (r >> 1); dir = r & 1; }   // tail group
    if (threadIdx.x == 0) wg_bad = 0;
    __syncthreads();
    if (dir)
        sweep<true, COMPACT>(a, n, smem, &wg_bad);
    else
        sweep<false, COMPACT>(a, n, smem, &wg_bad);
    // every wave has left its barrier sequence (or never had one): waves arrive here in any order, so the
    // flag is published by whoever saw it set, and cleared by thread 0 only if nobody did
    __syncthreads();
    if (threadIdx.x == 0) a.redo[2 * n + dir] = wg_bad;
}

}  // namespace pd

// hipErrorNotSupported when the lattice is too wide for one pass of this kernel.
hipError_t launch_lattice_pd(hipStream_t stream, const LatticeArgs& a, int N) {
    if (N <= 0) return hipSuccess;
    const int nA = (a.U + WAVE - 1) / WAVE;
    if (nA > pd::MAXA || !a.redo) return hipErrorNotSupported;
    const size_t lds = sizeof(pd::Smem) * nA;
    const dim3 grid(2 * N), block(3 * nA * WAVE);
    static std::atomic<bool> attr_set[2][64];
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    const int ci = a.offs ? 1 : 0;
    const bool tracked = dev >= 0 && dev < 64;
    if (!tracked || !attr_set[ci][dev].load(std::memory_order_acquire)) {
        const void* fn = a.offs ? reinterpret_cast<const void*>(&pd::k_lattice_pd<true>)
                                : reinterpret_cast<const void*>(&pd::k_lattice_pd<false>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(sizeof(pd::Smem) * pd::MAXA));
        if (e != hipSuccess) return e;
        if (tracked) attr_set[ci][dev].store(true, std::memory_order_release);
    }
    if (a.offs)
        pd::k_lattice_pd<true><<<grid, block, lds, stream>>>(a);
    else
        pd::k_lattice_pd<false><<<grid, block, lds, stream>>>(a);
    return hipGetLastError();
}

}  // namespace rnnt
