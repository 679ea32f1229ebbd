// Wave-specialised alpha / beta lattice sweep for MI355X (gfx950), diagonal-major layout.
//
// Same maths and the same anti-diagonal / DPP scheme as lattice.hip (read its header first);
// what changes is WHO issues the memory instructions.  Measured on MI355X (rocprofv3 SQ
// counters, tools/ubench/issue_cost.hip): a lone wave issues one instruction every ~4.5 cycles
// whatever its type, so the per-diagonal cost of the sweep is the NUMBER of instructions the
// computing wave executes (26 in lattice.hip: 15 VALU + loads, stores, row-offset SALU, branches)
// plus the time it sits in s_waitcnt behind its own loads and stores.  Here every 64-column
// block of the lattice gets two waves:
//   * a COMPUTE wave that touches only registers and LDS: per diagonal one ds_read_b64 of the
//     (blank,label) pair (prefetched a block ahead), the DPP shift, the lse chain, one
//     ds_write_b32 of the result (+ one for the hand-over to the next column block);
//   * an I/O wave on another SIMD of the same CU that streams the pairs HBM -> registers -> LDS
//     two blocks ahead and the results LDS -> HBM one block behind, with all the row-offset
//     arithmetic, liveness predicates and vmcnt waits.
// One s_barrier per block of 8 diagonals orders both hand-overs (pairs: I/O -> compute, values:
// compute -> I/O) and the column-block boundary mailbox.  LDS rings: pairs 3 slots, values 2.
//
// Timeline in global blocks g (every wave executes exactly G barriers), column block `idx`,
// local time p = g - idx - SHIFT:
//   I/O wave   at p: loads pairs(p+2) HBM -> registers ; writes pairs(p) registers -> LDS ;
//                    stores values(p-3) LDS -> HBM
//   compute    at p: computes local block lb = p-2 (its pairs were read from LDS during p-1),
//                    prefetches pairs(lb+1) from LDS, writes values(lb) to LDS
#include <atomic>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "lattice_step.h"

namespace rnnt {

namespace ws {

template <bool BETA, bool COMPACT>
__device__ __forceinline__ void sweep(const LatticeArgs& a, const int n, Smem* smem) {
    const UttLens len = utt_lens<COMPACT>(a.xn, a.yn, n, a.T, a.U);
    if (COMPACT && !len.ok) return;   // no plane of its own to sweep (uniform over the workgroup, before any barrier)
    const int Tn = len.Tn, Un = len.Un;
    const int T = COMPACT ? Tn : a.T, U = COMPACT ? Un : a.U;
    const int lane = threadIdx.x & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (Un == 1) {   // no labels: prefix / suffix sums by one wave (uniform over the workgroup, no barrier yet)
        if (w == 0) {
            const size_t nb1 = COMPACT ? compact_base(a, n) : (size_t)n * T * U;
            const float2* lp2 = reinterpret_cast<const float2*>(a.lp) + nb1;
            const float total = single_column_scan<BETA>(Tn, (BETA ? a.betas : a.alphas) + nb1, U, lane,
                                                         [&](int t) { return lp2[(size_t)t * U].x; });
            if (!BETA && lane == 0) a.ll[n] = total;
        }
        return;
    }
    const int nA = blockDim.x >> 7;                   // compute waves = I/O waves
    const bool is_io = w >= nA;
    const int idx = is_io ? w - nA : w;               // column block
    const size_t nbase = COMPACT ? compact_base(a, n) : (size_t)n * T * U;
    float* out = (BETA ? a.betas : a.alphas) + nbase;
    const int ndiag = Tn + Un - 1;
    const float NEG_INF = -__builtin_inff();
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
    // number of global blocks: the last column block finishes computing at hi+idx+1, its I/O wave
    // stores one block later
    const int hi_last = (min(ndiag, Tn + WAVE * (nwa - 1) + WAVE) + K - 1) / K;
    const int G = hi_last + (nwa - 1) + 3 + SHIFT;

    if (!live_blk) {
        for (int g = 0; g < G; ++g) block_barrier();
        return;
    }

    if (is_io) {
        // ------------------------------ I/O wave ------------------------------
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, T * U * 4, RSRC_WORD3);
        const __amdgpu_buffer_rsrc_t rs_lp = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.lp) + nbase * 2, 0, T * U * 8, RSRC_WORD3);
        const int voff_out = colvalid ? uc * 4 : OOB;
        const int rowb_lp = U * 8, rowb_out = U * 4;
        f32x2 regs[NBR][K];
        // row (forward diagonal mod T) of the next block to LOAD and of the next block to STORE
        const int dF0 = BETA ? (ndiag - 1 - lo * K) : lo * K;
        int row_ld = ((dF0 % T) + T) % T;
        int row_st = row_ld;
        auto advance = [&](int& row, int (&rows)[K]) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                rows[k] = row;
                row = BETA ? (row == 0 ? T - 1 : row - 1) : (row + 1 == T ? 0 : row + 1);
            }
        };
        auto load_block = [&](f32x2 (&dst)[K]) {
            int rows[K];
            advance(row_ld, rows);
#pragma unroll
            for (int k = 0; k < K; ++k)
                dst[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_lp, uc * 8, rows[k] * rowb_lp, 0));
        };
        // local time p: writes pairs(p) to LDS (loaded at p-DLOAD), stores values(p-3), then issues
        // the loads of pairs(p+DLOAD) -- in that order, so that the youngest loads have a whole block
        // to land.  GUARDED = near the ends of the live range, where not all three activities exist;
        // the steady state is straight-line so that the compiler's vmcnt counting is exact.
        auto io_step = [&](const int p, auto ph, auto guarded) {
            constexpr int PH = decltype(ph)::value;           // p mod NBR
            constexpr bool GUARDED = decltype(guarded)::value;
#ifndef RNNT_WS_NOIO
            if (!GUARDED || (p >= lo && p < hi)) {
                f32x2* dst = &sm.pairs[p % PSLOTS][0][lane];
#pragma unroll
                for (int k = 0; k < K; ++k) dst[k * WAVE] = regs[PH][k];
            }
            const int ps = p - 3;
            if (!GUARDED || (ps >= lo && ps < hi)) {
                const float* src = &sm.vals[ps & (VSLOTS - 1)][0][lane];
                int rows[K];
                advance(row_st, rows);
                const int d0 = ps * K;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool live = (unsigned)(d0 + k - ucol_chk) < (unsigned)Tn;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, src[k * WAVE]), rs_out,
                                                          live ? voff_out : OOB, rows[k] * rowb_out, 0);
                }
            }
            if (!GUARDED || (p + DLOAD >= lo && p + DLOAD < hi)) load_block(regs[(PH + DLOAD) % NBR]);
#endif
            block_barrier();
        };
        auto io_any = [&](const int p, auto guarded) {        // dispatch on p mod NBR
            const int m = ((p % NBR) + NBR) % NBR;
            if (m == 0) io_step(p, std::integral_constant<int, 0>{}, guarded);
            else if (m == 1) io_step(p, std::integral_constant<int, 1>{}, guarded);
            else io_step(p, std::integral_constant<int, 2>{}, guarded);
        };
        // barriers before this wave's first action (global block g = p + idx + SHIFT, first p = lo - DLOAD)
        const int p_first = lo - DLOAD;
        int g = 0;
        for (; g < p_first + idx + SHIFT; ++g) block_barrier();
        int p = p_first;
        const int p_end = hi + 3;                          // last store happens at p = hi+2
        // steady range: all three activities live, p = 0 (mod NBR) at its start
        int ps0 = lo + 3;
        ps0 += (NBR - ((ps0 % NBR) + NBR) % NBR) % NBR;
        const int ps1 = hi - DLOAD;                        // exclusive
        for (; p < p_end && p < ps0; ++p) io_any(p, std::true_type{});
        // Empty the memory queue once, where the compiler can see it: the guarded steps above issue their loads and
        // stores under conditions, so on the path into the loop the waitcnt pass can only assume the worst for the
        // registers they filled, and it would make the first step of EVERY iteration wait for (nearly) everything in
        // flight -- a memory round trip every third block (round 4: `vmcnt(3)` at the loop head in the ISA).  With a
        // known-empty queue here only the loop's own back edge decides, and that one it counts exactly.
#ifndef RNNT_WS_NO_DRAIN
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
#endif
        for (; p + NBR <= ps1; p += NBR) {
            io_step(p, std::integral_constant<int, 0>{}, std::false_type{});
            io_step(p + 1, std::integral_constant<int, 1>{}, std::false_type{});
            io_step(p + 2, std::integral_constant<int, 2>{}, std::false_type{});
        }
        for (; p < p_end; ++p) io_any(p, std::true_type{});
        for (g = p + idx + SHIFT; g < G; ++g) block_barrier();
        static_assert(NBR == 3, "io_step phases are written for a 3-deep register ring");
        return;
    }

    // ------------------------------ compute wave ------------------------------
#ifdef RNNT_WS_PRIO
    __builtin_amdgcn_s_setprio(3);   // the dependent chain is the critical path; I/O waves yield
#endif
    float Y = (ucol == 0) ? 0.0f : NEG_INF;
    float X = NEG_INF;
    f32x2 bufA[K], bufB[K];
    auto read_pairs = [&](f32x2 (&dst)[K], const int lb) {
        const f32x2* src = &sm.pairs[lb % PSLOTS][0][lane];
#pragma unroll
        for (int k = 0; k < K; ++k) dst[k] = src[k * WAVE];
    };
    auto do_block = [&](const int lb, f32x2 (&cur)[K], f32x2 (&nxt)[K]) {
        const int d0 = lb * K;
        float mvec = NEG_INF;
        if (idx > 0) {
            if (lane < K) mvec = smem[idx - 1].mail[(d0 - 1 + lane) & (RING - 1)];
        }
        if (lb + 1 < hi) read_pairs(nxt, lb + 1);         // written by the I/O wave two blocks ago
        float* vslot = &sm.vals[lb & (VSLOTS - 1)][0][lane];
        const bool has_right = idx + 1 < nwa;
        float* mail_slot = (lane == WAVE - 1) ? &sm.mail[d0 & (RING - 1)] : &sm.trash[lane];
        // every lane that owns a lattice column live for the whole block?  Lanes beyond the last column (last column
        // block of a lattice whose width is not a multiple of 64) run the unpredicated code as well: their values
        // only travel to the right, to other such lanes, and the I/O wave drops their stores (voffset = OOB).
        // Otherwise that column block -- the slowest link of the chain -- would run the predicated step for the
        // whole sweep.
        // (strictly behind the last column's first diagonal: that cell is on the rim -- it takes `emit`, not the lse)
        const bool full = (d0 > min(wave_c + WAVE - 1, Un - 1)) && (d0 + K <= wave_c + Tn);
        // (idx == 0: sweep column 0 is this wave's lane 0 -- the rim variant of the step, lattice_step.h)
#define RNNT_WS_BLOCK(M_, R_)                                                                                         \
    do {                                                                                                              \
        if (idx == 0) compute_block<BETA, M_, R_, true>(cur, mvec, Y, X, d0, ucol_chk, Tn, vslot, mail_slot);         \
        else compute_block<BETA, M_, R_, false>(cur, mvec, Y, X, d0, ucol_chk, Tn, vslot, mail_slot);                 \
    } while (0)
        if (full) {
            if (has_right) RNNT_WS_BLOCK(false, true); else RNNT_WS_BLOCK(false, false);
        } else {
            if (has_right) RNNT_WS_BLOCK(true, true); else RNNT_WS_BLOCK(true, false);
        }
#undef RNNT_WS_BLOCK
        block_barrier();
    };
    int g = 0;
    for (; g < lo + idx + 2 + SHIFT; ++g) block_barrier();
    read_pairs(bufA, lo);                                  // first block: exposed LDS latency once
    int lb = lo;
    for (; lb + 2 <= hi; lb += 2) {
        do_block(lb, bufA, bufB);
        do_block(lb + 1, bufB, bufA);
    }
    if (lb < hi) { do_block(lb, bufA, bufB); ++lb; }
    for (g = lb + idx + 2 + SHIFT; g < G; ++g) block_barrier();
    if constexpr (!BETA) {
        // Y of a finished lane is frozen at alpha + lpB of its last live cell (core_gather.cu:339)
        if (ucol == Un - 1) a.ll[n] = Y;
    }
}

template <bool COMPACT>
__global__ void __launch_bounds__(2 * MAXA * WAVE) k_lattice_ws(const LatticeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem* smem = reinterpret_cast<Smem*>(smem_raw);
    // XCD-aware placement: workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md), each XCD has its
    // own L2.  The alpha and the beta sweep of one utterance read the same diagonal-major plane
    // (from opposite ends), so they are given ids b and b+8: same XCD, shared L2 lines.
    // (Speed only; nothing depends on the placement.)
    const unsigned b = blockIdx.x, pairs_total = gridDim.x >> 1;
    const unsigned grp = b >> 4, in = b & 15;
    unsigned n, dir;
    if ((grp << 3) + 8 <= pairs_total) { n = (grp << 3) + (in & 7); dir = in >> 3; }
    else { const unsigned r = b - (grp << 4); n = (grp << 3) + (r >> 1); dir = r & 1; }   // tail group
    if (a.beta_only && !dir) return;                  // (the alpha plane is not the caller's to write: compact shim)
    if (a.redo && a.redo[2 * n + dir] == 0) return;   // swept by the probability-domain kernel (uniform)
    if (dir)
        sweep<true, COMPACT>(a, n, smem);
    else
        sweep<false, COMPACT>(a, n, smem);
}

}  // namespace ws

// Returns hipErrorNotSupported when the lattice is too wide for one pass (caller falls back to
// the single-role kernel of lattice.hip).
hipError_t launch_lattice_ws(hipStream_t stream, const LatticeArgs& a, int N) {
    if (N <= 0) return hipSuccess;
    const int nA = (a.U + WAVE - 1) / WAVE;
    if (nA > ws::MAXA) return hipErrorNotSupported;
    const size_t lds = sizeof(ws::Smem) * nA;
    const dim3 grid(2 * N), block(2 * nA * WAVE);
    // > 64 KiB of dynamic LDS needs an opt-in per kernel and per device.  hipFuncSetAttribute is idempotent
    // and thread-safe, so the only state kept is a per-(kernel, device) "already done" bit; devices beyond
    // the table simply repeat the call every launch.
    static std::atomic<bool> attr_set[2][64];
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    const bool compact = is_compact(a);       // 64-bit (native) or 32-bit (core.h shims) offsets: compact_base()
    const int ci = compact ? 1 : 0;
    const bool tracked = dev >= 0 && dev < 64;
    if (!tracked || !attr_set[ci][dev].load(std::memory_order_acquire)) {
        const void* fn = compact ? reinterpret_cast<const void*>(&ws::k_lattice_ws<true>)
                                : reinterpret_cast<const void*>(&ws::k_lattice_ws<false>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(sizeof(ws::Smem) * ws::MAXA));
        if (e != hipSuccess) return e;
        if (tracked) attr_set[ci][dev].store(true, std::memory_order_release);
    }
    if (compact)
        ws::k_lattice_ws<true><<<grid, block, lds, stream>>>(a);
    else
        ws::k_lattice_ws<false><<<grid, block, lds, stream>>>(a);
    return hipGetLastError();
}

}  // namespace rnnt
