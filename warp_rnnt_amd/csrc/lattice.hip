// Alpha / beta lattice sweeps of the RNN-Transducer loss for MI355X (gfx950).
//
// What it computes (per utterance n; maths = SURVEY.md Appendix A; reference
// kernels: core_gather.cu:37-133 (alphas), :135-234 (betas), dense indexing
// core.cu:84-87,116-120,189-192,221-225):
//   alpha[t,u] = lse(alpha[t-1,u] + lpB[t-1,u], alpha[t,u-1] + lpL[t,u-1]),  alpha[0,0] = 0
//   beta [t,u] = lse(beta[t+1,u] + lpB[t,u],    beta[t,u+1] + lpL[t,u]),     beta[T-1,U-1] = lpB[T-1,U-1]
//
// How (nothing like the reference's 32x1 warp tiles ordered by global spin
// locks):
//   * one workgroup per (utterance, direction); lane <-> lattice column,
//     wave w owns columns [64w, 64w+64); up to 16 waves (1024 columns) per
//     pass, wider lattices are swept in column stripes;
//   * true anti-diagonal sweep: at diagonal d every lane computes its cell
//     (d - u, u).  The value a cell needs from its left neighbour moves one
//     lane up with a single DPP `wave_shr:1`; the value it needs from the
//     previous frame is the lane's own register;
//   * wave w runs K diagonals (one "block") behind wave w-1; the boundary
//     column between two waves is handed over through a tiny LDS ring, K
//     values per block, with ONE s_barrier per K diagonals.  No global
//     atomics, no inter-workgroup ordering;
//   * log-probs are read from the diagonal-major workspace (common.h): each
//     diagonal is one contiguous row, K rows are prefetched into registers a
//     block ahead; alphas/betas are written in the same diagonal-major layout
//     with coalesced stores.
//   Critical path: (T_n + U_n - 1) + K*(waves-1) dependent lse steps.
#include "common.h"
#include "kernels.h"

namespace rnnt {

constexpr int K = 8;          // diagonals per block (= inter-wave lag)
constexpr int RING = 4 * K;   // mailbox ring entries per wave boundary
constexpr int MAXW = 16;      // waves per workgroup
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
                                          int row, int t, int u, int lab) {
    Cell c;
    if constexpr (LOADER == LOAD_SKEWED) {
        const f32x2 v = __builtin_bit_cast(
            f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, u * 8, row * a.U * 8, 0));
        c.b = v.x; c.l = v.y;
    } else if constexpr (LOADER == LOAD_ROWMAJOR2) {
        const float2 v = reinterpret_cast<const float2*>(a.lp)[nbase + (size_t)t * a.U + u];
        c.b = v.x; c.l = v.y;
    } else {
        const float* p = a.lp + (nbase + (size_t)t * a.U + u) * (size_t)a.V;
        c.b = p[a.blank];
        c.l = p[lab < 0 ? a.blank : lab];
    }
    return c;
}

// K consecutive diagonals of one wave.  MASKED: some lane of the wave starts or finishes inside
// the block, so state updates are predicated per lane; otherwise every lane is live throughout.
template <bool BETA, bool MASKED>
__device__ __forceinline__ void run_block(const Cell (&cur)[K], const float mvec, float& Y, float& X,
                                          const int d0, const int ucol_chk, const int Tn,
                                          __amdgpu_buffer_rsrc_t rs_out, const int voff_out, int& row_st,
                                          const int T, const int U, float* mail_slot) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float first = readlane(mvec, k);
        const float left = wave_shr1(first, X);
        float val, Yn, Xn;
        if constexpr (BETA) {
            val = lse(Y + cur[k].b, left + cur[k].l);
            Yn = val; Xn = val;
        } else {
            val = lse(Y, left);
            Yn = val + cur[k].b;
            Xn = val + cur[k].l;
        }
        // live cell <=> 0 <= d - ucol < Tn (ucol_chk is huge for columns outside the lattice)
        const bool live = (unsigned)(d0 + k - ucol_chk) < (unsigned)Tn;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, val), rs_out,
                                              (MASKED && !live) ? OOB : voff_out, row_st * U * 4, 0);
        if constexpr (MASKED) {
            Y = live ? Yn : Y;
            X = live ? Xn : X;
        } else {
            Y = Yn; X = Xn;
        }
        mail_slot[k] = X;   // only lane 63's pointer aims at the mailbox, the others at a dump area
        row_st = BETA ? (row_st == 0 ? T - 1 : row_st - 1) : (row_st + 1 == T ? 0 : row_st + 1);
    }
}

template <int LOADER, bool BETA>
__device__ __forceinline__ void sweep(const LatticeArgs& a, const int n, float (*mail)[RING],
                                      float (*trash)[MAIL_TRASH]) {
    const int T = a.T, U = a.U;
    const int Tn = a.xn[n], Un = a.yn[n] + 1;
    const int lane = threadIdx.x & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
    const int nw = blockDim.x >> 6;
    const size_t nbase = (size_t)n * T * U;
    float* out = (BETA ? a.betas : a.alphas) + nbase;
    const int ndiag = Tn + Un - 1;
    const float NEG_INF = -__builtin_inff();
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(out, 0, T * U * 4, RSRC_WORD3);
    const __amdgpu_buffer_rsrc_t rs_lp = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.lp) + (LOADER == LOAD_SKEWED ? nbase * 2 : 0), 0,
        LOADER == LOAD_SKEWED ? T * U * 8 : 0, RSRC_WORD3);

    for (int c0 = 0; c0 < Un; c0 += blockDim.x) {
        // ---- per-lane column bookkeeping (sweep coordinates: beta runs mirrored) ----
        const int ucol = c0 + (int)threadIdx.x;          // column in sweep coordinates
        const bool colvalid = ucol < Un;
        const int u = BETA ? (Un - 1 - ucol) : ucol;     // lattice column
        const int uc = min(max(u, 0), U - 1);            // address-safe column
        const int ucol_chk = colvalid ? ucol : 0x40000000;  // makes the live predicate false
        int lab = -1;
        if constexpr (LOADER == LOAD_DENSE) {
            if (uc < U - 1) lab = a.labels[(size_t)n * (U - 1) + uc];
        }
        const int nwa = min(nw, (Un - c0 + WAVE - 1) / WAVE);  // waves with a live column
        const int wave_c = c0 + WAVE * w;                      // first sweep column of this wave
        // blocks (of K diagonals) in which this wave has live cells
        const int lo = wave_c / K;
        const int hi = (min(ndiag, Tn + wave_c + WAVE) + K - 1) / K;
        const int nblk = (ndiag + K - 1) / K + (nwa - 1);      // uniform over the workgroup

        float Y = (ucol == 0) ? 0.0f : NEG_INF;   // alpha: alpha+lpB of own previous cell; beta: beta
        float X = NEG_INF;                        // alpha: alpha+lpL handed to the right; beta: beta
        Cell bufA[K], bufB[K];   // ping-pong: one holds the current block, the other the prefetch
        int row_nxt = 0;   // row (dF mod T) of the first diagonal of the block to prefetch next
        int row_st = 0;    // row of the diagonal being computed (store row)
        bool primed = false;
        const int voff_out = colvalid ? uc * 4 : OOB;

        // One block of this wave: global block b, log-probs in `cur`, prefetching into `nxt`.
        // (Two buffers + a 2x unrolled loop instead of copying nxt->cur: a register copy would
        // force `s_waitcnt vmcnt(0)` at the end of every block, which on gfx950 also drains the
        // alpha/beta stores just issued.)
        auto do_block = [&](const int b, Cell (&cur)[K], Cell (&nxt)[K]) {
            const int lb = b - w;   // local block of this wave (one block behind wave w-1)
            if (w < nwa && lb >= lo && lb < hi) {
                const int d0 = lb * K;
                // -- boundary column of this block: K values for diagonals d0-1 .. d0+K-2.
                //    Fetched BEFORE the prefetch loads are issued so that waiting for it does
                //    not drain them (vmcnt retires in order).
                float mvec = NEG_INF;
                if (w > 0) {
                    if (lane < K) mvec = mail[w - 1][(d0 - 1 + lane) & (RING - 1)];
                } else if (c0 > 0) {
                    // stripe boundary: the previous pass of this workgroup left column c0-1 in `out`
                    const int dd = d0 - 1 + lane;               // sweep diagonal of the neighbour cell
                    if (lane < K && dd >= c0 - 1 && dd - (c0 - 1) < Tn) {
                        const int ub = BETA ? (Un - c0) : (c0 - 1);
                        const int dFb = BETA ? (ndiag - 1 - dd) : dd;
                        const int rb = dFb % T;
                        float v = out[(size_t)rb * U + ub];
                        if constexpr (!BETA) {
                            int labb = -1;
                            if constexpr (LOADER == LOAD_DENSE) labb = a.labels[(size_t)n * (U - 1) + ub];
                            v += load_cell<LOADER>(a, rs_lp, nbase, rb, dd - (c0 - 1), ub, labb).l;
                        }
                        mvec = v;
                    }
                }
                // -- log-probs: the first live block loads synchronously, later ones were prefetched --
                if (!primed) {
                    const int dF0 = BETA ? (ndiag - 1 - d0) : d0;
                    row_st = ((dF0 % T) + T) % T;
                    row_nxt = row_st;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const int tt = d0 + k - ucol;
                        const int t = min(max(BETA ? (Tn - 1 - tt) : tt, 0), T - 1);
                        cur[k] = load_cell<LOADER>(a, rs_lp, nbase, row_nxt, t, uc, lab);
                        row_nxt = BETA ? (row_nxt == 0 ? T - 1 : row_nxt - 1)
                                       : (row_nxt + 1 == T ? 0 : row_nxt + 1);
                    }
                    primed = true;
                }
                // -- prefetch the next block's K diagonals (always in-bounds, may be unused) --
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int tt = d0 + K + k - ucol;
                    const int t = min(max(BETA ? (Tn - 1 - tt) : tt, 0), T - 1);
                    nxt[k] = load_cell<LOADER>(a, rs_lp, nbase, row_nxt, t, uc, lab);
                    row_nxt = BETA ? (row_nxt == 0 ? T - 1 : row_nxt - 1)
                                   : (row_nxt + 1 == T ? 0 : row_nxt + 1);
                }

                // lane 63 of a wave with a right neighbour publishes X; everyone else dumps it
                float* mail_slot = (lane == WAVE - 1 && w + 1 < nwa) ? &mail[w][d0 & (RING - 1)]
                                                                     : &trash[w][lane];
                // every lane live for the whole block?  (started: d0 >= last lane's column;
                // not finished: d0+K-1 - first column < Tn; all 64 columns inside the lattice)
                const bool full = (d0 >= wave_c + WAVE - 1) && (d0 + K <= wave_c + Tn) &&
                                  (wave_c + WAVE <= Un);
                if (full)
                    run_block<BETA, false>(cur, mvec, Y, X, d0, ucol_chk, Tn, rs_out, voff_out, row_st, T, U,
                                           mail_slot);
                else
                    run_block<BETA, true>(cur, mvec, Y, X, d0, ucol_chk, Tn, rs_out, voff_out, row_st, T, U,
                                          mail_slot);
            }
            if (nwa > 1) {
                // LDS-only release/acquire around the barrier: global prefetches stay in flight.
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            }
        };

        for (int b = 0; b < nblk; b += 2) {
            do_block(b, bufA, bufB);
            if (b + 1 < nblk) do_block(b + 1, bufB, bufA);
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
    }
}

template <int LOADER>
__global__ void __launch_bounds__(MAXW * WAVE) k_lattice(const LatticeArgs a) {
    __shared__ float mail[MAXW][RING];
    __shared__ float trash[MAXW][MAIL_TRASH];
    const int n = blockIdx.x >> 1;
    if (blockIdx.x & 1)
        sweep<LOADER, true>(a, n, mail, trash);
    else
        sweep<LOADER, false>(a, n, mail, trash);
}

hipError_t launch_lattice(hipStream_t stream, const LatticeArgs& a, int N, int loader) {
    if (N <= 0) return hipSuccess;
    int waves = (a.U + WAVE - 1) / WAVE;
    waves = waves < 1 ? 1 : (waves > MAXW ? MAXW : waves);
    const dim3 grid(2 * N), block(waves * WAVE);
    switch (loader) {
        case LOAD_SKEWED:    k_lattice<LOAD_SKEWED><<<grid, block, 0, stream>>>(a); break;
        case LOAD_ROWMAJOR2: k_lattice<LOAD_ROWMAJOR2><<<grid, block, 0, stream>>>(a); break;
        default:             k_lattice<LOAD_DENSE><<<grid, block, 0, stream>>>(a); break;
    }
    return hipGetLastError();
}

}  // namespace rnnt
