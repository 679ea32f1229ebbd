// The log-domain lattice step shared by the two wave-specialised sweep kernels (lattice_ws.hip: all column blocks of
// a sweep in one workgroup; lattice_wd.hip: one workgroup per column block).  One definition, so that both kernels
// produce the same bits: K diagonals of lse(skip, emit) per call, lanes = lattice columns, the left neighbour's value
// through one DPP wave_shr:1.  Reference counterpart: core_gather.cu:22-35 (log_sum_exp), :106-126 / :207-227 (the
// per-cell recurrences).
#pragma once
#include "common.h"

namespace rnnt {

namespace ws {

constexpr int K = 8;             // diagonals per block
constexpr int RING = 4 * K;      // mailbox ring entries per column-block boundary
constexpr int MAXA = 8;          // compute waves per workgroup (=> 512 columns per pass)
constexpr int PSLOTS = 3;        // LDS ring of pair blocks
constexpr int VSLOTS = 2;        // LDS ring of value blocks
constexpr int DLOAD = 2;         // I/O wave loads pairs this many blocks before it writes them to LDS
constexpr int NBR = DLOAD + 1;   // its register ring
constexpr int SHIFT = DLOAD;     // global block g = local time + idx + SHIFT, so the first load is at g >= 0
constexpr int TRASH = WAVE + K;
constexpr int RSRC_WORD3 = 0x00020000;
constexpr int OOB = (int)0x80000000;

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Smem {   // per column block
    f32x2 pairs[PSLOTS][K][WAVE];
    float vals[VSLOTS][K][WAVE];
    float mail[RING];
    float trash[TRASH];
};

__device__ __forceinline__ void block_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// K diagonals of the compute wave.  cur = this block's pairs (registers).  Values go to LDS.
template <bool BETA, bool MASKED, bool MAIL>
__device__ __forceinline__ void compute_block(const f32x2 (&cur)[K], const float mvec, float& Y, float& X,
                                              const int d0, const int ucol_chk, const int Tn,
                                              float* vslot /* [K][WAVE] + lane */, float* mail_slot) {
    float first[K];
#pragma unroll
    for (int k = 0; k < K; ++k) first[k] = readlane(mvec, k);
#define RNNT_PIN() __builtin_amdgcn_sched_barrier(0)
    // A lone wave issues one instruction per ~5.8 cycles whatever it is (tools/ubench/step_order.hip),
    // so the step is ordered to need NO hazard nops: the LDS write and the next skip/Y add sit between
    // the value and the DPP that reads it (2 wait states), v_max sits behind v_exp_f32 (1 wait state),
    // and the v_mov that seeds the next DPP's lane 0 is issued well before it.
    float fk = first[0];
    asm volatile("" : "+v"(fk));   // materialise the DPP's lane-0 seed in a VGPR here, not next to the DPP
    RNNT_PIN();
    float pval = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float skip, emit;
        if constexpr (BETA) {
            // beta: the value of the previous diagonal is published, then extended by this cell's
            // blank log-prob -- both read `Y` and sit between its producer and the DPP below
            if (k > 0) {
#ifndef RNNT_WS_NOVAL
                vslot[(k - 1) * WAVE] = pval;
#endif
                RNNT_PIN();
            }
            skip = Y + cur[k].x;
            RNNT_PIN();
        } else {
            skip = Y;
        }
        const float left = wave_shr1(fk, X);                                           // chain
        RNNT_PIN();
        if constexpr (BETA) {
            emit = left + cur[k].y;                                                    // chain (scalar add: the
                                                                                       // file is built with -fno-slp-vectorize)
        } else {
            emit = left;
        }
        RNNT_PIN();
        // lse(skip, emit) = max + log1p(exp(-|skip-emit|)), see lattice.hip
        const float t = skip - emit;                                                   // chain
        RNNT_PIN();
        const float m = -__builtin_fabsf(t) * 1.44269504088896340736f;                 // chain
        RNNT_PIN();
        const float e = __builtin_amdgcn_exp2f(m);                                     // chain
        RNNT_PIN();
        const float mx = __builtin_fmaxf(skip, emit);                                  // fills the trans wait state
        RNNT_PIN();
        const float u = 1.0f + e;                                                      // chain
        RNNT_PIN();
        const float l2 = __builtin_amdgcn_logf(u);                                     // chain
        RNNT_PIN();
        if (k + 1 < K) { fk = first[k + 1]; asm volatile("" : "+v"(fk)); RNNT_PIN(); }
        const float um1 = u - 1.0f;
        RNNT_PIN();
        const float c = e - um1;
        RNNT_PIN();
        const float l = __builtin_fmaf(l2, 0.693147180559945309417f, c);               // chain
        RNNT_PIN();
        const float val = mx + l;                                                      // chain
        RNNT_PIN();
        float Yn, Xn;
        if constexpr (BETA) {
            Yn = val; Xn = val;
        } else {
            Xn = val + cur[k].y;                                                       // chain (feeds the DPP)
            RNNT_PIN();
#ifndef RNNT_WS_NOVAL
            vslot[k * WAVE] = val;
#endif
            RNNT_PIN();
            Yn = val + cur[k].x;
            RNNT_PIN();
        }
        if constexpr (MASKED) {
            const bool live = (unsigned)(d0 + k - ucol_chk) < (unsigned)Tn;
            Y = live ? Yn : Y;
            X = live ? Xn : X;
        } else {
            Y = Yn; X = Xn;
        }
        pval = val;
        if constexpr (MAIL) { mail_slot[k] = X; RNNT_PIN(); }
    }
    if constexpr (BETA) {
#ifndef RNNT_WS_NOVAL
        vslot[(K - 1) * WAVE] = pval;
#endif
    }
#undef RNNT_PIN
}

// ---------------------------------------------------------------------------------------------------------------
// The same K diagonals -- same instructions on the dependent chain, same bits -- for lattice_wd.hip, with everything a
// block needs from LDS fetched IN PLACE one block ahead:
//   * the pair of diagonal k is dead once step k has used it: step k reloads cur[k] with the next block's pair k;
//   * SEEDED (the column block has a left neighbour): seed[k] holds the neighbour's boundary value for step k in
//     every lane (a broadcast LDS read; only lane 0's copy matters: it is the `old` operand of the DPP shift, which
//     lane 0 keeps).  The DPP consumes it, step k reloads it with the next block's value.  No v_readlane / v_mov pair
//     per step, no exposed LDS round trip at the head of the block.
// One buffer each = one copy of the block per variant in the instruction stream (a register ping-pong needs the loop
// unrolled twice), which matters for instruction fetch at the head of a column block.
// The reloads are inline assembly (as C++ the compiler loads into fresh registers and copies them over at the head of the
// loop: sixteen v_mov and eight waits per block on the wave whose instruction count IS the sweep's critical path), so
// they are not counted by the compiler: the caller must not let a block start before an `s_waitcnt lgkmcnt(0)` it can
// rely on -- the block barrier's (this wave has LDS writes of its own pending in front of every barrier, so the release
// fence always carries one).
template <int OFF>
__device__ __forceinline__ void lds_reload_b64(f32x2& dst, const unsigned lds_byte_addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_byte_addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_reload_b32(float& dst, const unsigned lds_byte_addr) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_byte_addr), "n"(OFF));
}
// (k is a constant once the caller's loop is unrolled; the offset has to be one for the assembler)
__device__ __forceinline__ void reload_pair(f32x2& dst, const unsigned nsrc, const int k) {
    constexpr int ROW = WAVE * 8;
    switch (k) {
        case 0: lds_reload_b64<0 * ROW>(dst, nsrc); break;
        case 1: lds_reload_b64<1 * ROW>(dst, nsrc); break;
        case 2: lds_reload_b64<2 * ROW>(dst, nsrc); break;
        case 3: lds_reload_b64<3 * ROW>(dst, nsrc); break;
        case 4: lds_reload_b64<4 * ROW>(dst, nsrc); break;
        case 5: lds_reload_b64<5 * ROW>(dst, nsrc); break;
        case 6: lds_reload_b64<6 * ROW>(dst, nsrc); break;
        default: lds_reload_b64<7 * ROW>(dst, nsrc); break;
    }
    static_assert(K == 8, "one case per diagonal of a block");
}
__device__ __forceinline__ void reload_seed(float& dst, const unsigned nseed, const int k) {
    switch (k) {
        case 0: lds_reload_b32<0>(dst, nseed); break;
        case 1: lds_reload_b32<4>(dst, nseed); break;
        case 2: lds_reload_b32<8>(dst, nseed); break;
        case 3: lds_reload_b32<12>(dst, nseed); break;
        case 4: lds_reload_b32<16>(dst, nseed); break;
        case 5: lds_reload_b32<20>(dst, nseed); break;
        case 6: lds_reload_b32<24>(dst, nseed); break;
        default: lds_reload_b32<28>(dst, nseed); break;
    }
}

template <bool BETA, bool MASKED, bool MAIL, bool SEEDED>
__device__ __forceinline__ void compute_block_ip(f32x2 (&cur)[K], float (&seed)[K], const unsigned nsrc, const unsigned nseed,
                                                 float& Y, float& X, const int d0, const int ucol_chk, const int Tn,
                                                 float* vslot /* [K][WAVE] + lane */, float* mail_slot) {
#define RNNT_PIN() __builtin_amdgcn_sched_barrier(0)
    // (the order of the step is compute_block's: no hazard nops, see there)
    float fk = -__builtin_inff();
    if constexpr (!SEEDED) { asm volatile("" : "+v"(fk)); RNNT_PIN(); }   // the DPP's lane-0 seed, in a VGPR ahead of the DPP
    float pval = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float skip, emit;
        if constexpr (BETA) {
            if (k > 0) {
                vslot[(k - 1) * WAVE] = pval;
                RNNT_PIN();
            }
            skip = Y + cur[k].x;
            RNNT_PIN();
        } else {
            skip = Y;
        }
        const float left = wave_shr1(SEEDED ? seed[k] : fk, X);                        // chain
        RNNT_PIN();
        if constexpr (BETA) {
            emit = left + cur[k].y;                                                    // chain
            RNNT_PIN();
            reload_pair(cur[k], nsrc, k);                                              // next block's pair k
        } else {
            emit = left;
        }
        RNNT_PIN();
        const float t = skip - emit;                                                   // chain
        RNNT_PIN();
        const float m = -__builtin_fabsf(t) * 1.44269504088896340736f;                 // chain
        RNNT_PIN();
        const float e = __builtin_amdgcn_exp2f(m);                                     // chain
        RNNT_PIN();
        const float mx = __builtin_fmaxf(skip, emit);                                  // fills the trans wait state
        RNNT_PIN();
        const float u = 1.0f + e;                                                      // chain
        RNNT_PIN();
        const float l2 = __builtin_amdgcn_logf(u);                                     // chain
        RNNT_PIN();
        if constexpr (SEEDED) {
            reload_seed(seed[k], nseed, k);                                            // next block's boundary value k
            RNNT_PIN();
        } else if (k + 1 < K) {
            fk = -__builtin_inff(); asm volatile("" : "+v"(fk)); RNNT_PIN();
        }
        const float um1 = u - 1.0f;
        RNNT_PIN();
        const float c = e - um1;
        RNNT_PIN();
        const float l = __builtin_fmaf(l2, 0.693147180559945309417f, c);               // chain
        RNNT_PIN();
        const float val = mx + l;                                                      // chain
        RNNT_PIN();
        float Yn, Xn;
        if constexpr (BETA) {
            Yn = val; Xn = val;
        } else {
            Xn = val + cur[k].y;                                                       // chain (feeds the DPP)
            RNNT_PIN();
            vslot[k * WAVE] = val;
            RNNT_PIN();
            Yn = val + cur[k].x;
            RNNT_PIN();
            reload_pair(cur[k], nsrc, k);                                              // next block's pair k
            RNNT_PIN();
        }
        if constexpr (MASKED) {
            const bool live = (unsigned)(d0 + k - ucol_chk) < (unsigned)Tn;
            Y = live ? Yn : Y;
            X = live ? Xn : X;
        } else {
            Y = Yn; X = Xn;
        }
        pval = val;
        if constexpr (MAIL) { mail_slot[k] = X; RNNT_PIN(); }
    }
    if constexpr (BETA) vslot[(K - 1) * WAVE] = pval;
#undef RNNT_PIN
}

}  // namespace ws

}  // namespace rnnt
