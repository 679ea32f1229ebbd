// The single-role lattice sweep (one workgroup per sweep, every wave computes and does its own I/O; any of the three
// loaders; lattices wider than the workgroup in column stripes): the body of lattice.hip's k_lattice.  A header since
// round 6, when k_lattice_wd was given a copy of it as an in-kernel redo of sweeps with a lost hand-over (measured
// slower than the idle redo launch it replaced, and removed again: lattice_wd_body.h); lattice.hip is its one user.
// Reference: core_gather.cu:37-133 (alphas), :135-234 (betas); the arithmetic and its order are lattice_step.h's (same
// bits: tests/test_gpu_wd.py).  Read lattice.hip's header first.
#pragma once
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace rnnt {
namespace single {

#ifndef RNNT_K
#define RNNT_K 8
#endif
constexpr int K = RNNT_K;     // diagonals per block (= inter-wave lag)
// Register ring of NB blocks: log-probs are prefetched NB-1 blocks ahead (24 diagonals for the
// diagonal-major loader: an L2-miss/MALL round trip is ~1 us, a block ~0.3 us).  The two
// reference-layout loaders need 64-bit addresses per load and keep a 2-deep ring.
#ifndef RNNT_STORE_AUX
#define RNNT_STORE_AUX 0
#endif
#ifndef RNNT_NB
#define RNNT_NB 4
#endif
template <int LOADER> constexpr int ring_depth() { return LOADER == LOAD_SKEWED ? RNNT_NB : 2; }
constexpr int RING = 4 * K;   // mailbox ring entries per wave boundary
#ifndef RNNT_MAXW
#define RNNT_MAXW 16
#endif
constexpr int MAXW = RNNT_MAXW;   // waves per workgroup
constexpr int MAIL_TRASH = WAVE + K;   // per-wave dump area for the lanes that are not lane 63

struct Cell { float b, l; };  // blank / label log-prob of one lattice cell

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int RSRC_WORD3 = 0x00020000;   // raw buffer, 32-bit data format (gfx90a/gfx94x/gfx950)
constexpr int OOB = (int)0x80000000;     // voffset beyond any num_records: loads return 0, stores drop

// Log-probs of the cell on forward diagonal row (= dF mod T) in lattice column u.
//   SKEWED:    the row is enough: one coalesced 8-byte buffer load, row offset in an SGPR.
//   ROWMAJOR2: needs t = dF - u (clamped for lanes outside the lattice).
//   DENSE:     same, plus the label index of column u (lab < 0: no label, use blank).
template <int LOADER>
__device__ __forceinline__ Cell load_cell(const LatticeArgs& a, __amdgpu_buffer_rsrc_t rs, size_t nbase,
                                          int U, int row, int t, int u, int lab) {
    Cell c;
    if constexpr (LOADER == LOAD_SKEWED) {
        const f32x2 v = __builtin_bit_cast(
            f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, u * 8, row * U * 8, 0));
        c.b = v.x; c.l = v.y;
    } else if constexpr (LOADER == LOAD_ROWMAJOR2) {
        const float2 v = reinterpret_cast<const float2*>(a.lp)[nbase + (size_t)t * U + u];
        c.b = v.x; c.l = v.y;
    } else {
        const float* p = a.lp + (nbase + (size_t)t * U + u) * (size_t)a.V;
        c.b = p[a.blank];
        c.l = p[lab < 0 ? a.blank : lab];
    }
    return c;
}

// K consecutive diagonals of one wave.  MASKED: some lane of the wave starts or finishes inside
// the block, so state updates are predicated per lane; otherwise every lane is live throughout.
template <bool BETA, bool MASKED, bool MAIL>
__device__ __forceinline__ void run_block(const Cell (&cur)[K], const float mvec, float& Y, float& X,
                                          const int d0, const int ucol_chk, const int Tn,
                                          __amdgpu_buffer_rsrc_t rs_out, const int voff_out, int& row_st,
                                          const int T, const int U, float* mail_slot) {
    // Hoisted out of the dependency chain: the K boundary values (SGPRs) and the K store row
    // offsets.  A single wave issues in order and a dependent VALU op costs ~6-11 cycles on
    // gfx950 while an independent one costs ~3, so everything that is not lse() is kept short
    // and early.
    float first[K];
    int soff[K];
#pragma unroll
    for (int k = 0; k < K; ++k) first[k] = readlane(mvec, k);
    const int ucol_first = ucol_chk == 0 ? 0x40000001 : ucol_chk;   // the lane's first live diagonal (column 0: none)
    const bool col0 = ucol_chk == 0;
    const int rowbytes = U * 4;
    const bool nowrap = BETA ? (row_st >= K - 1) : (row_st + K <= T);
    if (nowrap) {
        const int base = row_st * rowbytes;
#pragma unroll
        for (int k = 0; k < K; ++k) soff[k] = BETA ? base - k * rowbytes : base + k * rowbytes;
        row_st = BETA ? row_st - K : row_st + K;
        if (BETA) { if (row_st < 0) row_st += T; } else { if (row_st >= T) row_st -= T; }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            soff[k] = row_st * rowbytes;
            row_st = BETA ? (row_st == 0 ? T - 1 : row_st - 1) : (row_st + 1 == T ? 0 : row_st + 1);
        }
    }
    // lse(a,b) = log(exp(a)+exp(b)) keeps the reference's structure (core.cu:26-39):
    //   max(a,b) + log1p(exp(-|a-b|)).
    // The transcendental part runs on the hardware v_exp_f32 / v_log_f32 units instead of the
    // ~150-instruction ocml expf+log1pf pair (4x slower end to end; -DRNNT_PRECISE_LIBM builds that
    // version for comparison -- this chain is the latency-critical path of the whole op):
    //   e = 2^(-|a-b|*log2 e)                  abs. error <~ 1e-8 (it shrinks as fast as e)
    //   u = fl(1+e);  log1p(e) = ln2*log2(u) + (e-(u-1))    the last term is the exact rounding
    //                                                       error of 1+e; ~2 ulp of a value <= ln 2
    // Both are far below the fp32 rounding of the final `max + ...` whenever |max| >= 1.  Measured
    // against the libm-based fp32 oracle: sum of costs 7418.4797 vs 7418.4796 on the T=1500,U=300
    // benchmark lattice, gradients within 1e-4 at T+U <~ 200 (tests/test_gpu_parity.py).
    //
    // Hand-ordered step: a single wave issues in order, so the store / mailbox write / helper ops
    // are placed in the latency shadow of the dependent chain
    //   dpp -> add -> sub -> mul -> exp2 -> add -> log2 -> fma -> add
    // and pinned there with sched_barrier (the compiler otherwise puts the store of step k
    // between val(k) and the DPP that starts step k+1).  Stores and mailbox writes trail one
    // step behind the values they publish.
#define RNNT_PIN() __builtin_amdgcn_sched_barrier(0)
    float pval = 0.0f, pX = 0.0f;     // value / hand-over of the previous step, still to be stored
    int pvoff = OOB;
#pragma unroll
    for (int k = 0; k <= K; ++k) {
        float left = 0.0f, skip = 0.0f;
        if (k < K) {
            left = wave_shr1(first[k], X);                                   // chain
            RNNT_PIN();
            if constexpr (BETA) skip = Y + cur[k].b;
            RNNT_PIN();
        }
        if (k > 0) {
#ifndef RNNT_PROBE_NOSTORE
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, pval), rs_out, pvoff, soff[k - 1], RNNT_STORE_AUX);
#endif
            RNNT_PIN();
        }
        if (k == K) {
#ifndef RNNT_PROBE_NOMAIL
            if constexpr (MAIL) mail_slot[k - 1] = pX;
#endif
            break;
        }
        float emit;
        if constexpr (BETA) { emit = left + cur[k].l; } else { emit = left; skip = Y; }   // chain
        RNNT_PIN();
        if (k > 0) {
#ifndef RNNT_PROBE_NOMAIL
            // only lane 63's pointer aims at the mailbox, the others at a dump area
            if constexpr (MAIL) mail_slot[k - 1] = pX;
#endif
            RNNT_PIN();
        }
        // ---- lse(skip, emit), see common.h ----
        const float t = skip - emit;                                                   // chain
        RNNT_PIN();
        float mx;
        asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(skip), "v"(emit));                 // shadow
        RNNT_PIN();
        const float m = -__builtin_fabsf(t) * 1.44269504088896340736f;                 // chain
        RNNT_PIN();
        const bool live = (unsigned)(d0 + k - ucol_chk) < (unsigned)Tn;                // shadow
        RNNT_PIN();
        const float e = __builtin_amdgcn_exp2f(m);                                     // chain
        RNNT_PIN();
        const float u = 1.0f + e;                                                      // chain
        RNNT_PIN();
        const float l2 = __builtin_amdgcn_logf(u);                                     // chain
        RNNT_PIN();
#if defined(RNNT_PRECISE_LIBM)
        float val = mx + log1pf(expf(-__builtin_fabsf(t)));
        (void)l2;
#elif !defined(RNNT_LSE_UNCORRECTED)
        const float c = e - (u - 1.0f);                                                // shadow
        RNNT_PIN();
        const float l = __builtin_fmaf(l2, 0.693147180559945309417f, c);               // chain
        RNNT_PIN();
        float val = mx + l;                                                            // chain
#else
        // Probe only: max + ln2*log2(1+e) in one fma, rounding of 1+e left uncorrected.  4 % faster,
        // but pushes gradients past the 1e-4 parity bar at T=150,U=40 -- not used.
        float val = __builtin_fmaf(l2, 0.693147180559945309417f, mx);                  // chain
#endif
        RNNT_PIN();
        // the rim of the lattice is plain sums in the reference, not lse (lattice_step.h: a -inf there must stay -inf,
        // not turn into NaN): a lane's first live diagonal takes `emit`, sweep column 0 takes `skip`
        if constexpr (MASKED) val = (d0 + k == ucol_first) ? emit : val;
        val = col0 ? skip : val;
        RNNT_PIN();
        float Yn, Xn;
        if constexpr (BETA) { Yn = val; Xn = val; }
        else { Xn = val + cur[k].l; RNNT_PIN(); Yn = val + cur[k].b; }
        pval = val;
        pvoff = (MASKED && !live) ? OOB : voff_out;
        if constexpr (MASKED) {
            Y = live ? Yn : Y;
            X = live ? Xn : X;
        } else {
            Y = Yn; X = Xn;
        }
        pX = X;
        RNNT_PIN();
    }
#undef RNNT_PIN
}

template <int LOADER, bool BETA, bool COMPACT>
__device__ __forceinline__ void sweep(const LatticeArgs& a, const int n, float (*mail)[RING],
                                      float (*trash)[MAIL_TRASH]) {
    const UttLens len = utt_lens<COMPACT>(a.xn, a.yn, n, a.T, a.U);
    if (COMPACT && !len.ok) return;   // no plane of its own to sweep (uniform over the workgroup, before any barrier)
    const int Tn = len.Tn, Un = len.Un;
    // padded planes (N,T,U), or -- compact layout -- one (T_n,U_n) plane per utterance at offs[n]
    const int T = COMPACT ? Tn : a.T, U = COMPACT ? Un : a.U;
    const int lane = threadIdx.x & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    const int nw = blockDim.x >> 6;
    const size_t nbase = COMPACT ? compact_base(a, n) : (size_t)n * T * U;
    float* out = (BETA ? a.betas : a.alphas) + nbase;
    if (Un == 1) {   // no labels: prefix / suffix sums by one wave (see common.h); uniform, before any barrier
        if (w == 0) {
            const float total = single_column_scan<BETA>(Tn, out, U, lane, [&](int t) {
                if constexpr (LOADER == LOAD_DENSE)
                    return a.lp[(nbase + (size_t)t * U) * (size_t)a.V + a.blank];
                else   // diagonal-major and row-major pairs keep cell (t,0) at the same index t*U
                    return reinterpret_cast<const float2*>(a.lp)[nbase + (size_t)t * U].x;
            });
            if (!BETA && lane == 0) a.ll[n] = total;
        }
        return;
    }
    const int ndiag = Tn + Un - 1;
    const float NEG_INF = -__builtin_inff();
    constexpr int NB = ring_depth<LOADER>();
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(out, 0, T * U * 4, RSRC_WORD3);
    const __amdgpu_buffer_rsrc_t rs_lp = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.lp) + (LOADER == LOAD_SKEWED ? nbase * 2 : 0), 0,
        LOADER == LOAD_SKEWED ? T * U * 8 : 0, RSRC_WORD3);

    // One column stripe of up to blockDim.x columns.  FIRST (c0 == 0) is the common case and is
    // compiled separately: later stripes fetch their boundary column with global loads, and
    // merely having that path in the block prologue makes the compiler drain the prefetch queue.
    auto stripe = [&](auto first_tag, const int c0) {
        constexpr bool FIRST = decltype(first_tag)::value;
        // ---- per-lane column bookkeeping (sweep coordinates: beta runs mirrored) ----
        const int ucol = c0 + (int)threadIdx.x;          // column in sweep coordinates
        const bool colvalid = ucol < Un;
        const int u = BETA ? (Un - 1 - ucol) : ucol;     // lattice column
        const int uc = min(max(u, 0), U - 1);            // address-safe column
        const int ucol_chk = colvalid ? ucol : 0x40000000;  // makes the live predicate false
        int lab = -1;
        if constexpr (LOADER == LOAD_DENSE) {
            if (uc < U - 1) lab = safe_label(a.labels[(size_t)n * (U - 1) + uc], a.V, a.blank);
        }
        const int nwa = min(nw, (Un - c0 + WAVE - 1) / WAVE);  // waves with a live column
        const int wave_c = c0 + WAVE * w;                      // first sweep column of this wave
        // blocks (of K diagonals) in which this wave has live cells
        const int lo = wave_c / K;
        const int hi = (min(ndiag, Tn + wave_c + WAVE) + K - 1) / K;
        const int nblk = (ndiag + K - 1) / K + (nwa - 1);      // uniform over the workgroup

        float Y = (ucol == 0) ? 0.0f : NEG_INF;   // alpha: alpha+lpB of own previous cell; beta: beta
        float X = NEG_INF;                        // alpha: alpha+lpL handed to the right; beta: beta
        Cell bufs[NB][K];  // register ring, always indexed with compile-time constants
        int row_nxt = 0;   // row (dF mod T) of the first diagonal of the block to prefetch next
        int row_st = 0;    // row of the diagonal being computed (store row)
        const int voff_out = colvalid ? uc * 4 : OOB;

        // One block of this wave: global block b = PH (mod NB); its log-probs sit in bufs[PH] and
        // the block NB-1 ahead is prefetched into bufs[PH-1], the buffer the previous block freed.
        // (A ring + an NB-times unrolled loop instead of copying registers: a copy would force
        // `s_waitcnt vmcnt(0)` at the end of every block, which on gfx950 also drains the
        // alpha/beta stores just issued.)
        auto load_block = [&](Cell (&dst)[K], const int dblk) {   // diagonals dblk*K ..., rows from row_nxt
            int rows[K];
            const bool nowrap = BETA ? (row_nxt >= K - 1) : (row_nxt + K <= T);
            if (nowrap) {
#pragma unroll
                for (int k = 0; k < K; ++k) rows[k] = BETA ? row_nxt - k : row_nxt + k;
                row_nxt = BETA ? row_nxt - K : row_nxt + K;
                if (BETA) { if (row_nxt < 0) row_nxt += T; } else { if (row_nxt >= T) row_nxt -= T; }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    rows[k] = row_nxt;
                    row_nxt = BETA ? (row_nxt == 0 ? T - 1 : row_nxt - 1)
                                   : (row_nxt + 1 == T ? 0 : row_nxt + 1);
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                int t = 0;
                if constexpr (LOADER != LOAD_SKEWED) {
                    const int tt = dblk * K + k - ucol;
                    t = min(max(BETA ? (Tn - 1 - tt) : tt, 0), T - 1);
                }
                dst[k] = load_cell<LOADER>(a, rs_lp, nbase, U, rows[k], t, uc, lab);
            }
        };
        auto do_block = [&](const int lb, auto ph) {
            constexpr int PH = decltype(ph)::value;
            Cell (&cur)[K] = bufs[PH];
            {
                const int d0 = lb * K;
                // -- boundary column of this block: K values for diagonals d0-1 .. d0+K-2.
                //    Fetched BEFORE the prefetch loads are issued so that waiting for it does
                //    not drain them (vmcnt retires in order).
                float mvec = NEG_INF;
                if (w > 0) {
                    if (lane < K) mvec = mail[w - 1][(d0 - 1 + lane) & (RING - 1)];
                } else if (!FIRST) {
                    // stripe boundary: the previous pass of this workgroup left column c0-1 in `out`
                    const int dd = d0 - 1 + lane;               // sweep diagonal of the neighbour cell
                    if (lane < K && dd >= c0 - 1 && dd - (c0 - 1) < Tn) {
                        const int ub = BETA ? (Un - c0) : (c0 - 1);
                        const int dFb = BETA ? (ndiag - 1 - dd) : dd;
                        const int rb = dFb % T;
                        float v = out[(size_t)rb * U + ub];
                        if constexpr (!BETA) {
                            int labb = -1;
                            if constexpr (LOADER == LOAD_DENSE) labb = safe_label(a.labels[(size_t)n * (U - 1) + ub], a.V, a.blank);
                            v += load_cell<LOADER>(a, rs_lp, nbase, U, rb, dd - (c0 - 1), ub, labb).l;
                        }
                        mvec = v;
                    }
                }
                // -- prefetch block lb+NB-1 (always in-bounds addresses, may be unused) --
                load_block(bufs[(PH + NB - 1) % NB], lb + NB - 1);

                // lane 63 of a wave with a right neighbour publishes X; everyone else dumps it
                float* mail_slot = (lane == WAVE - 1 && w + 1 < nwa) ? &mail[w][d0 & (RING - 1)]
                                                                     : &trash[w][lane];
                // every lane live for the whole block?  (started: d0 >= last lane's column;
                // not finished: d0+K-1 - first column < Tn; all 64 columns inside the lattice)
                const bool full = (d0 > wave_c + WAVE - 1) && (d0 + K <= wave_c + Tn) &&
                                  (wave_c + WAVE <= Un);
                const bool has_right = w + 1 < nwa;   // someone consumes this wave's boundary column
                if (full) {
                    if (has_right)
                        run_block<BETA, false, true>(cur, mvec, Y, X, d0, ucol_chk, Tn, rs_out, voff_out, row_st,
                                                     T, U, mail_slot);
                    else
                        run_block<BETA, false, false>(cur, mvec, Y, X, d0, ucol_chk, Tn, rs_out, voff_out, row_st,
                                                      T, U, mail_slot);
                } else {
                    if (has_right)
                        run_block<BETA, true, true>(cur, mvec, Y, X, d0, ucol_chk, Tn, rs_out, voff_out, row_st,
                                                    T, U, mail_slot);
                    else
                        run_block<BETA, true, false>(cur, mvec, Y, X, d0, ucol_chk, Tn, rs_out, voff_out, row_st,
                                                     T, U, mail_slot);
                }
            }
#ifndef RNNT_PROBE_NOBARRIER
            if (nwa > 1) {
                // LDS-only release/acquire around the barrier: global prefetches stay in flight.
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            }
#endif
        };

        auto barrier_only = [&]() {
            if (nwa > 1) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            }
        };
        // Every wave executes exactly nblk barriers: idle ones before its first live block
        // (global block lo+w), one per live block, idle ones after its last.  The live range is a
        // straight-line NB-times unrolled loop so that the compiler's vmcnt bookkeeping is exact
        // and the prefetched loads really stay in flight across blocks.
        const bool wave_live = (w < nwa) && (lo < hi);
        const int first_b = wave_live ? lo + w : nblk;
        for (int b = 0; b < first_b; ++b) barrier_only();
        if (wave_live) {
            {   // fill the ring: blocks lo .. lo+NB-2 (the loop prefetches lo+NB-1 onwards)
                const int dF0 = BETA ? (ndiag - 1 - lo * K) : lo * K;
                row_st = ((dF0 % T) + T) % T;
                row_nxt = row_st;
                load_block(bufs[0], lo);
                if constexpr (NB > 2) load_block(bufs[1], lo + 1);
                if constexpr (NB > 3) load_block(bufs[2], lo + 2);
                static_assert(NB >= 2 && NB <= 4, "ring depth");
            }
            int lb = lo;
            for (; lb + NB <= hi; lb += NB) {
                do_block(lb, std::integral_constant<int, 0>{});
                do_block(lb + 1, std::integral_constant<int, 1 % NB>{});
                if constexpr (NB > 2) do_block(lb + 2, std::integral_constant<int, 2 % NB>{});
                if constexpr (NB > 3) do_block(lb + 3, std::integral_constant<int, 3 % NB>{});
            }
            if (lb < hi) { do_block(lb, std::integral_constant<int, 0>{}); ++lb; }
            if (lb < hi) { do_block(lb, std::integral_constant<int, 1 % NB>{}); ++lb; }
            if constexpr (NB > 3) { if (lb < hi) { do_block(lb, std::integral_constant<int, 2 % NB>{}); ++lb; } }
            for (int b = hi + w; b < nblk; ++b) barrier_only();
        }
        if constexpr (!BETA) {
            // Y of a finished lane is frozen at alpha + lpB of its last live cell: for the last
            // column that is the alpha-side log-likelihood alpha[T-1,U-1] + lpB[T-1,U-1]
            // (core_gather.cu:339)
            if (ucol == Un - 1) a.ll[n] = Y;
        }
        if (c0 + (int)blockDim.x < Un) {
            // next stripe reads column c0+blockDim.x-1 of `out` written by this workgroup
            __threadfence_block();
            __syncthreads();
        }
    };
    stripe(std::true_type{}, 0);
    for (int c0 = blockDim.x; c0 < Un; c0 += blockDim.x) stripe(std::false_type{}, c0);
}


}  // namespace single
}  // namespace rnnt
