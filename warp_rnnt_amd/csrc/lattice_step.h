// The log-domain lattice step shared by the wave-specialised sweep kernels (lattice_ws.hip: all column blocks of a sweep
// in one workgroup; lattice_wd.hip: one workgroup per column block, and its single-workgroup form k_lattice_wl).  One
// definition, so that all of them produce the same bits: K diagonals of lse(skip, emit) per call, lanes = lattice columns,
// the left neighbour's value through one DPP wave_shr:1.  Reference counterpart: core_gather.cu:22-35 (log_sum_exp),
// :106-126 / :207-227 (the per-cell recurrences), :76-104 / :177-205 (the boundary row and column: plain sums).
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

// ---------------------------------------------------------------------------------------------------------------
// The cells on the rim of the lattice.  The reference does not run them through log_sum_exp: alpha[0,u] and beta[T-1,u]
// are plain sums along the row (core_gather.cu:76-84, 177-185), alpha[t,0] and beta[t,U-1] plain sums along the column
// (:86-104, 187-205).  A sweep that feeds them to lse(x, -inf) gets the same bits for every finite x -- max(x,-inf) +
// log1p(exp(-inf)) = x + 0 -- which is what these kernels did until round 5; it gets NaN where x itself is -inf
// (a masked label or blank on the rim: -inf - -inf), where the reference's sum stays -inf and the lattice behind the
// cell stays finite.  So:
//   * a lane's FIRST live diagonal (row 0 of alpha, row T-1 of beta in sweep coordinates; sweep column 0 excepted) takes
//     `emit`.  Such diagonals only exist in the predicated blocks (MASKED): one compare + one select per step there;
//   * sweep column 0 (lane 0 of the column block without a left neighbour: COL0) takes `skip` on every diagonal: one
//     select per step, paid for by the v_mov that used to seed the DPP shift's lane 0 with -inf (wave_shr1_z needs none).
// The interior keeps the reference's NaN: lse(-inf, -inf) = NaN there as in core_gather.cu:22-35.
// ucol_chk: the lane's sweep column (0x40000000 for lanes without one).
__device__ __forceinline__ int first_diag_of(const int ucol_chk) { return ucol_chk == 0 ? 0x40000001 : ucol_chk; }

// K diagonals of the compute wave.  cur = this block's pairs (registers).  Values go to LDS.
// COL0: this wave's lane 0 is sweep column 0 (no left neighbour; mvec is not read).
template <bool BETA, bool MASKED, bool MAIL, bool COL0>
__device__ __forceinline__ void compute_block(const f32x2 (&cur)[K], const float mvec, float& Y, float& X,
                                              const int d0, const int ucol_chk, const int Tn,
                                              float* vslot /* [K][WAVE] + lane */, float* mail_slot) {
    float first[K];
    if constexpr (!COL0) {
#pragma unroll
        for (int k = 0; k < K; ++k) first[k] = readlane(mvec, k);
    }
    const int ucol_first = first_diag_of(ucol_chk);
    const bool col0 = ucol_chk == 0;
#define RNNT_PIN() __builtin_amdgcn_sched_barrier(0)
    // A lone wave issues one instruction per ~5.8 cycles whatever it is (tools/ubench/step_order.hip),
    // so the step is ordered to need NO hazard nops: the LDS write and the next skip/Y add sit between
    // the value and the DPP that reads it (2 wait states), v_max sits behind v_exp_f32 (1 wait state),
    // and the v_mov that seeds the next DPP's lane 0 is issued well before it.
    float fk = 0.0f;
    if constexpr (!COL0) {
        fk = first[0];
        asm volatile("" : "+v"(fk));   // materialise the DPP's lane-0 seed in a VGPR here, not next to the DPP
        RNNT_PIN();
    }
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
        float left;
        if constexpr (COL0) left = wave_shr1_z(X); else left = wave_shr1(fk, X);      // chain
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
        bool rim = false;                                                              // (compared here, two wait states
        if constexpr (MASKED) { rim = d0 + k == ucol_first; RNNT_PIN(); }              //  ahead of the select that reads it)
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
        if constexpr (!COL0) {
            if (k + 1 < K) { fk = first[k + 1]; asm volatile("" : "+v"(fk)); RNNT_PIN(); }
        }
        const float um1 = u - 1.0f;
        RNNT_PIN();
        const float c = e - um1;
        RNNT_PIN();
        const float l = __builtin_fmaf(l2, 0.693147180559945309417f, c);               // chain
        RNNT_PIN();
        float val = mx + l;                                                            // chain
        RNNT_PIN();
        if constexpr (MASKED) { val = rim ? emit : val; RNNT_PIN(); }                    // rim: first live diagonal
        if constexpr (COL0) { val = col0 ? skip : val; RNNT_PIN(); }                       // rim: sweep column 0
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
// The same diagonals -- same operations in the same order, same bits -- for lattice_wd.hip, KK of them per call, with
// everything a block needs from LDS fetched IN PLACE one block ahead:
//   * the pairs of two consecutive diagonals share a register quad (cur2[j] = {pair 2j, pair 2j+1}); once step 2j+1 has
//     used it, ONE ds_read2st64_b64 refills it with the next block's two pairs (the rows of a block are 512 bytes apart:
//     the instruction's stride);
//   * SEEDED (the column block has a left neighbour): the neighbour's boundary values for four consecutive steps share a
//     register quad, every lane holding all four (broadcast LDS reads; only lane 0's copy matters: it is the `old`
//     operand of the DPP shift, which lane 0 keeps).  The DPP of step k consumes component k mod 4 in place; once the
//     fourth is spent ONE ds_read_b128 refills the quad with the next block's four.  Not SEEDED = sweep column 0 is this
//     wave's lane 0 (COL0 above).
// One buffer each = one copy of the block per variant in the instruction stream.
// The reloads are inline assembly, so they are not counted by the compiler: the caller must not let a block start before
// an `s_waitcnt lgkmcnt(0)` of its own (lattice_wd.hip: in front of every barrier of the compute wave).
//
// Two bodies.  The predicated block (MASKED: lanes start or finish inside it -- a few blocks at either end of a column
// block's life) is C++.  The block every lane is live in -- nine tenths of a long sweep, and the sweep runs at the pace of
// its slowest wave -- is written instruction by instruction (round 5), because a lone wave issues one instruction every
// ~5.8 cycles whatever it is (priced link by link in round 6, tools/ubench/step_chain.hip: 4.7 - 5.0 cycles for a dependent
// vector instruction, 4.0 for an independent one or an s_nop, 8 for v_exp_f32 / v_log_f32, nothing extra for DPP -- there
// are no latency shadows to hide work in), so the sweep's time IS that wave's instruction count, and the compiler's version
// carried three to four instructions per diagonal that do nothing for the result:
//   * fmaxf(a, b) is up to three v_max_f32: the compiler quiets signalling NaNs first (v_max x, x, x) on every operand it
//     cannot prove to be the result of an arithmetic instruction -- the DPP shift's output always -- which made the
//     alpha sweep two instructions per diagonal longer than the beta sweep.  Nothing on the chain produces a signalling
//     NaN, and a quiet one ends in val = NaN whatever the max returns (l is NaN then): one v_max_f32;
//   * a v_mov per diagonal that copied the (wave-uniform) address of the seeds from an SGPR into a VGPR for the reload;
//   * the column block without a left neighbour seeded lane 0 of the shift with -inf through a v_mov per diagonal: here
//     the shift is folded into the instructions that consume it (v_sub_f32_dpp / v_max_f32_dpp / v_add_f32_dpp with
//     bound_ctrl: lane 0 reads 0.0), and lane 0 -- sweep column 0, whose value is `skip` by the rim rule -- is put right
//     by one v_cndmask;
//   * (an asm statement in the middle of compiler code costs an s_nop at its end -- the compiler pads one wait state
//     before the first use of anything an asm statement wrote -- hence whole steps, seed reload included.
//     The one thing left to the compiler between two steps is the store of the value: an instruction it can see, which
//     therefore counts as that wait state.)
// Per diagonal: 14 VALU/LDS instructions + half a reload of pairs + a quarter of a reload of seeds, against 17-18.
// Hazards kept by construction (the compiler does not look inside): >= 2 instructions between the VALU write of the
// value handed right and the DPP read of it (the store + one add), one instruction between v_exp_f32 / v_log_f32 and the
// first ordinary VALU read of their result (v_max; v_add + v_sub).
// Round 6: checked on every build instead of assumed (_isa_check.check_hazards walks the ISA of the object that ships).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_float;   // (a pointer that can only be LDS: ds_write, never flat_store)

template <int OFF0>
__device__ __forceinline__ void lds_reload_2rows(f32x4& dst, const unsigned lds_byte_addr) {
    // rows OFF0 and OFF0 + 1 of a [KK][WAVE] block of pairs, this lane's column: two 8-byte reads 64 x 8 bytes apart
    asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(lds_byte_addr), "n"(OFF0), "n"(OFF0 + 1) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_reload_b32(float& dst, const unsigned lds_byte_addr) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_byte_addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_reload_b128(f32x4& dst, const unsigned lds_byte_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_byte_addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Every in-place reload of a block is retired by ONE s_waitcnt lgkmcnt(0) in front of the block's barrier (wait_lds).
// For a while in round 5 the hand-written blocks left their last reloads -- the next block's last seeds and last two
// pairs -- in flight ACROSS the barrier (lgkmcnt(2)) and retired them half-way through the next block with a counted
// lgkmcnt(N), N = the LDS operations issued since: ~1 us at c4, the same bits in every test, in the fuzz runs and in
// 260 000 back-to-back launches of a lone process -- and WRONG RESULTS in 1-3 % of the launches as soon as two other
// processes kept the GPU busy (tools/wd_soak.py --procs 3: costs off by 0.1 ... 1, no flag raised; the same build with
// zeros in the two waits: none in 360 000).  The mechanism, as far as it was traced: a register that an inline-assembly
// load writes is, to the compiler, free again the moment its C++ variable dies -- it hands it to something else while the
// load is still on its way, and a load that lands late (an LDS kept busy by other processes' workgroups) overwrites the
// new tenant.  Reloads in flight across the barrier multiply the places where that can happen; the storer's dry run
// without its end-of-block wait was one more (lattice_wd_body.h: one_block).  THE RULE: a register written by an
// inline-assembly load stays an operand of the code until an explicit wait has retired the load, and every block --
// the dry ones too -- ends with lgkmcnt(0).  tools/check_inplace_reloads.py checks it over every edge of the control-flow
// graph; tools/wd_soak.py is the regression test (tests/test_gpu_fuzz.py).
// The wait at the end of a block, with the refilled registers as its INPUT OPERANDS (round 6).  wait_lds() alone relies on
// the compiler happening to keep cur2[] / seed4[] where the reloads were aimed until the wait has run -- true while they
// are loop-carried, not true where they die first (the storer's dry run: the bug above).  As "v" inputs of the statement
// that contains the wait they are live, in the registers the reloads wrote, up to the wait, whatever comes after: the
// rule above in a form the register allocator keeps, not only one a checker verifies afterwards.
// Inputs, NOT "+v": with read-write operands the allocator ties them to the loop's phi registers, splits the live range
// and copies the quads into place IN FRONT of the wait -- v_mov_b64 of registers whose reload is still in flight, 96 of
// them over the eight kernels (measured, round 6; the checker below refused the build).  An input may sit anywhere, so
// the reload's own destination is used and whatever copy the loop needs comes behind the wait.
// The checker still runs on every build (_build.py: the ISA of the object that ships).
#ifdef RNNT_PLANT_RELOAD_VIOLATION   // tests/test_host_cpu.py: the build must refuse this (two reloads left in flight)
#define RNNT_BLOCK_END_WAIT "s_waitcnt lgkmcnt(2)"
#else
#define RNNT_BLOCK_END_WAIT "s_waitcnt lgkmcnt(0)"
#endif
template <int KK, bool SEEDED>
__device__ __forceinline__ void wait_lds_keep(const f32x4 (&c)[KK / 2], const f32x4 (&s)[KK / 4]) {
    static_assert(KK == 8 || KK == 16, "blocks of 8 or 16 diagonals");
    if constexpr (KK == 8) {
        if constexpr (SEEDED)
            asm volatile(RNNT_BLOCK_END_WAIT ::"v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(s[0]), "v"(s[1]) : "memory");
        else
            asm volatile(RNNT_BLOCK_END_WAIT ::"v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]) : "memory");
    } else {
        if constexpr (SEEDED)
            asm volatile(RNNT_BLOCK_END_WAIT ::"v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]),
                         "v"(c[7]), "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]) : "memory");
        else
            asm volatile(RNNT_BLOCK_END_WAIT ::"v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]),
                         "v"(c[7]) : "memory");
    }
}
#undef RNNT_BLOCK_END_WAIT

#define RNNT_LSE_TAIL(T, E, MX, U)                                                                                      \
    "v_add_f32 " U ", 1.0, " E "\n\t"                                                                                    \
    "v_log_f32 " T ", " U "\n\t"                                                                                         \
    "v_add_f32 " U ", -1.0, " U "\n\t"                                                                                   \
    "v_sub_f32 " E ", " E ", " U "\n\t"                                                                                  \
    "v_fmac_f32 " E ", 0x3f317218, " T "\n\t"
#define RNNT_DPP " wave_shr:1 row_mask:0xf bank_mask:0xf"
// the seed of step K_ is component K_ mod 4 of a register quad (a vector element can be an asm operand, not a reference)
#define RNNT_WITH_SEED(C, STMT)                                                                                         \
    do {                                                                                                                \
        if constexpr ((C) == 0) { STMT(sd4.x); } else if constexpr ((C) == 1) { STMT(sd4.y); }                          \
        else if constexpr ((C) == 2) { STMT(sd4.z); } else { STMT(sd4.w); }                                             \
    } while (0)

// One diagonal of a block every lane is live in; returns the cell's value (the caller stores it: a store the compiler can
// see, between two statements -- see above).  K_ = the diagonal's index in the block (the seed's LDS offset).
// alpha: Y = alpha + blank log-prob of the own previous cell, X = alpha + label log-prob (handed right).
template <int K_, bool SEEDED>
__device__ __forceinline__ float alpha_step(float& Y, float& X, f32x4& sd4, const float cx, const float cy, const float log2e, const unsigned long long col0_mask) {
    float t, e, val, u;
    if constexpr (SEEDED) {
        #define RNNT_STEP_ASM(SD) asm volatile("v_mov_b32_dpp %2, %1" RNNT_DPP "\n\t"          /* emit: lane i <- X of lane i-1, lane 0 keeps its seed */ \
                     "v_sub_f32 %3, %0, %2\n\t"                      /* t = skip - emit */ \
                     "v_mul_f32_e64 %3, -|%3|, %9\n\t" \
                     "v_exp_f32 %4, %3\n\t" \
                     "v_max_f32 %5, %0, %2\n\t" \
                     RNNT_LSE_TAIL("%3", "%4", "%5", "%6") \
                     "v_add_f32 %5, %5, %4\n\t"                      /* val = max + l */ \
                     "v_add_f32 %1, %5, %8\n\t"                      /* X = val + label log-prob */ \
                     "v_add_f32 %0, %5, %7"                           /* Y = val + blank log-prob */ \
                     : "+v"(Y), "+v"(X), "+v"(SD), "=&v"(t), "=&v"(e), "=&v"(val), "=&v"(u) \
                     : "v"(cx), "v"(cy), "s"(log2e))
        RNNT_WITH_SEED(K_ % 4, RNNT_STEP_ASM);
        #undef RNNT_STEP_ASM
    } else {
        asm volatile("v_sub_f32_dpp %2, %1, %0" RNNT_DPP " bound_ctrl:1\n\t"   // t = emit - skip (lane 0: 0 - skip, unused)
                     "v_mul_f32_e64 %2, -|%2|, %8\n\t"
                     "v_exp_f32 %3, %2\n\t"
                     "v_max_f32_dpp %4, %1, %0" RNNT_DPP " bound_ctrl:1\n\t"
                     RNNT_LSE_TAIL("%2", "%3", "%4", "%5")
                     "v_add_f32 %4, %4, %3\n\t"
                     "v_cndmask_b32_e64 %4, %4, %0, %9\n\t"         // sweep column 0 (lane 0): the rim takes skip
                     "v_add_f32 %1, %4, %7\n\t"
                     "v_add_f32 %0, %4, %6"
                     : "+v"(Y), "+v"(X), "=&v"(t), "=&v"(e), "=&v"(val), "=&v"(u)
                     : "v"(cx), "v"(cy), "s"(log2e), "s"(col0_mask));
        (void)sd4;
    }
    return val;
}
// beta: V = beta of the own previous cell = what is handed right = the value.
template <int K_, bool SEEDED>
__device__ __forceinline__ void beta_step(float& V, f32x4& sd4, const float cx, const float cy, const float log2e, const unsigned long long col0_mask) {
    float sk, t, e, u;
    if constexpr (SEEDED) {
        #define RNNT_STEP_ASM(SD) asm volatile("v_add_f32 %2, %0, %6\n\t"                      /* skip = beta[t+1,u] + blank log-prob */ \
                     "v_mov_b32_dpp %1, %0" RNNT_DPP "\n\t"          /* left */ \
                     "v_add_f32 %1, %1, %7\n\t"                      /* emit = beta[t,u+1] + label log-prob */ \
                     "v_sub_f32 %3, %2, %1\n\t" \
                     "v_mul_f32_e64 %3, -|%3|, %8\n\t" \
                     "v_exp_f32 %4, %3\n\t" \
                     "v_max_f32 %2, %2, %1\n\t" \
                     RNNT_LSE_TAIL("%3", "%4", "%2", "%5") \
                     "v_add_f32 %0, %2, %4" \
                     : "+v"(V), "+v"(SD), "=&v"(sk), "=&v"(t), "=&v"(e), "=&v"(u) \
                     : "v"(cx), "v"(cy), "s"(log2e))
        RNNT_WITH_SEED(K_ % 4, RNNT_STEP_ASM);
        #undef RNNT_STEP_ASM
    } else {
        float em;
        asm volatile("v_add_f32 %1, %0, %6\n\t"                      // skip
                     "v_add_f32_dpp %2, %0, %7" RNNT_DPP " bound_ctrl:1\n\t"   // emit (lane 0: 0 + label log-prob, unused)
                     "v_sub_f32 %3, %1, %2\n\t"
                     "v_mul_f32_e64 %3, -|%3|, %8\n\t"
                     "v_exp_f32 %4, %3\n\t"
                     "v_max_f32 %2, %1, %2\n\t"
                     RNNT_LSE_TAIL("%3", "%4", "%2", "%5")
                     "v_add_f32 %0, %2, %4\n\t"
                     "v_cndmask_b32_e64 %0, %0, %1, %9"              // sweep column 0 (lane 0): the rim takes skip
                     : "+v"(V), "=&v"(sk), "=&v"(em), "=&v"(t), "=&v"(e), "=&v"(u)
                     : "v"(cx), "v"(cy), "s"(log2e), "s"(col0_mask));
        (void)sd4;
    }
}
// The same diagonal in a HEAD block: lanes start inside it (their first live diagonal), none finishes.  What a lane
// computes before its first live diagonal never reaches a result -- its values are not stored (the storer predicates),
// what it hands right is read by lanes that are not live either, and its own state is replaced wholesale at the first
// live diagonal, where the rim rule makes the value `emit` (which comes from the left neighbour's first live cell) and
// Y, X follow from the value -- so the per-lane `live` selects of the general predicated step are not needed here, only
// the rim select: one v_cndmask on a single-bit lane mask (bit = the lane whose first diagonal this is; 0 if none).
template <int K_, bool SEEDED>
__device__ __forceinline__ float alpha_head_step(float& Y, float& X, f32x4& sd4, const float cx, const float cy, const float log2e, const unsigned long long col0_mask,
                                                 const unsigned long long rim_mask) {
    float t, e, val, u;
    if constexpr (SEEDED) {
        #define RNNT_STEP_ASM(SD) asm volatile("v_mov_b32_dpp %2, %1" RNNT_DPP "\n\t"          /* emit: lane 0 keeps its seed */ \
                     "v_sub_f32 %3, %0, %2\n\t" \
                     "v_mul_f32_e64 %3, -|%3|, %9\n\t" \
                     "v_exp_f32 %4, %3\n\t" \
                     "v_max_f32 %5, %0, %2\n\t" \
                     RNNT_LSE_TAIL("%3", "%4", "%5", "%6") \
                     "v_add_f32 %5, %5, %4\n\t" \
                     "v_cndmask_b32_e64 %5, %5, %2, %10\n\t"         /* rim: the lane's first live diagonal takes emit */ \
                     "v_add_f32 %1, %5, %8\n\t" \
                     "v_add_f32 %0, %5, %7" \
                     : "+v"(Y), "+v"(X), "+v"(SD), "=&v"(t), "=&v"(e), "=&v"(val), "=&v"(u) \
                     : "v"(cx), "v"(cy), "s"(log2e), "s"(rim_mask))
        RNNT_WITH_SEED(K_ % 4, RNNT_STEP_ASM);
        #undef RNNT_STEP_ASM
    } else {
        float em;
        asm volatile("v_mov_b32_dpp %2, %1" RNNT_DPP " bound_ctrl:1\n\t"   // emit (lane 0: 0.0, unused)
                     "v_sub_f32 %3, %0, %2\n\t"
                     "v_mul_f32_e64 %3, -|%3|, %9\n\t"
                     "v_exp_f32 %4, %3\n\t"
                     "v_max_f32 %5, %0, %2\n\t"
                     RNNT_LSE_TAIL("%3", "%4", "%5", "%6")
                     "v_add_f32 %5, %5, %4\n\t"
                     "v_cndmask_b32_e64 %5, %5, %2, %11\n\t"         // rim: first live diagonal takes emit
                     "v_cndmask_b32_e64 %5, %5, %0, %10\n\t"         // rim: sweep column 0 (lane 0) takes skip
                     "v_add_f32 %1, %5, %8\n\t"
                     "v_add_f32 %0, %5, %7"
                     : "+v"(Y), "+v"(X), "=&v"(em), "=&v"(t), "=&v"(e), "=&v"(val), "=&v"(u)
                     : "v"(cx), "v"(cy), "s"(log2e), "s"(col0_mask), "s"(rim_mask));
        (void)sd4;
    }
    return val;
}
template <int K_, bool SEEDED>
__device__ __forceinline__ void beta_head_step(float& V, f32x4& sd4, const float cx, const float cy, const float log2e, const unsigned long long col0_mask,
                                               const unsigned long long rim_mask) {
    float sk, t, e, u;
    if constexpr (SEEDED) {
        #define RNNT_STEP_ASM(SD) asm volatile("v_add_f32 %2, %0, %6\n\t"                      /* skip */ \
                     "v_mov_b32_dpp %1, %0" RNNT_DPP "\n\t"          /* left */ \
                     "v_add_f32 %1, %1, %7\n\t"                      /* emit */ \
                     "v_sub_f32 %3, %2, %1\n\t" \
                     "v_mul_f32_e64 %3, -|%3|, %8\n\t" \
                     "v_exp_f32 %4, %3\n\t" \
                     "v_max_f32 %2, %2, %1\n\t" \
                     RNNT_LSE_TAIL("%3", "%4", "%2", "%5") \
                     "v_add_f32 %0, %2, %4\n\t" \
                     "v_cndmask_b32_e64 %0, %0, %1, %9"              /* rim: first live diagonal takes emit */ \
                     : "+v"(V), "+v"(SD), "=&v"(sk), "=&v"(t), "=&v"(e), "=&v"(u) \
                     : "v"(cx), "v"(cy), "s"(log2e), "s"(rim_mask))
        RNNT_WITH_SEED(K_ % 4, RNNT_STEP_ASM);
        #undef RNNT_STEP_ASM
    } else {
        float em, mx;
        asm volatile("v_add_f32 %1, %0, %7\n\t"                      // skip
                     "v_add_f32_dpp %2, %0, %8" RNNT_DPP " bound_ctrl:1\n\t"   // emit (lane 0: 0 + label log-prob, unused)
                     "v_sub_f32 %4, %1, %2\n\t"
                     "v_mul_f32_e64 %4, -|%4|, %9\n\t"
                     "v_exp_f32 %5, %4\n\t"
                     "v_max_f32 %3, %1, %2\n\t"
                     RNNT_LSE_TAIL("%4", "%5", "%3", "%6")
                     "v_add_f32 %0, %3, %5\n\t"
                     "v_cndmask_b32_e64 %0, %0, %2, %11\n\t"         // rim: first live diagonal takes emit
                     "v_cndmask_b32_e64 %0, %0, %1, %10"             // rim: sweep column 0 (lane 0) takes skip
                     : "+v"(V), "=&v"(sk), "=&v"(em), "=&v"(mx), "=&v"(t), "=&v"(e), "=&v"(u)
                     : "v"(cx), "v"(cy), "s"(log2e), "s"(col0_mask), "s"(rim_mask));
        (void)sd4;
    }
}
#undef RNNT_WITH_SEED
#undef RNNT_LSE_TAIL
#undef RNNT_DPP

// steps [K_, KK) of a block, recursively (every LDS offset has to be an immediate)
// HEAD: rim0 = the lane mask of the lane whose first live diagonal is the block's first (bit d0 - first sweep column of the
// column block); step K_'s lane is K_ further up
template <int KK, int K_, bool BETA, bool SEEDED, bool HEAD>
__device__ __forceinline__ void fast_steps(f32x4 (&cur2)[KK / 2], f32x4 (&seed4)[KK / 4], const unsigned nsrc, const unsigned nseed_v,
                                           float& Y, float& X, lds_float* vslot, const float log2e,
                                           const unsigned long long col0_mask, const unsigned long long rim0) {
    if constexpr (K_ < KK) {
        const float cx = (K_ & 1) ? cur2[K_ / 2].z : cur2[K_ / 2].x;
        const float cy = (K_ & 1) ? cur2[K_ / 2].w : cur2[K_ / 2].y;
        // (a head block starts before the column block's last column does: first_off + K_ <= 63 -- lattice_wd.hip; one
        //  scalar shift of the block's first mask per step)
        const unsigned long long rim_mask = HEAD ? rim0 << K_ : 0ull;
        // (the seed of step K_: a component of a register quad -- the DPP shift works on it in place; four steps on, one
        //  ds_read_b128 refills the quad with the next block's four)
        f32x4 unused_seed = {0.0f, 0.0f, 0.0f, 0.0f};
        f32x4& sd4 = SEEDED ? seed4[K_ / 4] : unused_seed;
        if constexpr (BETA) {
            if constexpr (HEAD) beta_head_step<K_, SEEDED>(X, sd4, cx, cy, log2e, col0_mask, rim_mask);
            else beta_step<K_, SEEDED>(X, sd4, cx, cy, log2e, col0_mask);
            vslot[K_ * WAVE] = X;       // (this store and the next step's first add sit between the value and its DPP read)
        } else {
            float val;
            if constexpr (HEAD) val = alpha_head_step<K_, SEEDED>(Y, X, sd4, cx, cy, log2e, col0_mask, rim_mask);
            else val = alpha_step<K_, SEEDED>(Y, X, sd4, cx, cy, log2e, col0_mask);
            vslot[K_ * WAVE] = val;     // (alpha: the add of Y and this store sit between X and its DPP read)
        }
        // (the store above is what separates the step's statement from the reloads: the compiler pads a wait state in
        //  front of a statement that redefines a register the previous statement wrote, unless an instruction of its own
        //  sits between them)
        if constexpr (SEEDED && K_ % 4 == 3) lds_reload_b128<(K_ - 3) * 4>(seed4[K_ / 4], nseed_v);   // next block's seeds
        if constexpr ((K_ & 1) != 0) lds_reload_2rows<K_ - 1>(cur2[K_ / 2], nsrc);       // next block's pairs K_ - 1, K_
        fast_steps<KK, K_ + 1, BETA, SEEDED, HEAD>(cur2, seed4, nsrc, nseed_v, Y, X, vslot, log2e, col0_mask, rim0);
    }
}

// nsrc / nseed: LDS byte addresses of the next block's pairs (this lane's column) and seeds; vslot: this block's slot of
// the value ring, [KK][WAVE] + lane.  BETA keeps one state (X; Y is not used).
// MODE: what the lanes of the wave do inside the block
enum BlockMode : int {
    BLOCK_FULL = 0,      // every lane that owns a column is live throughout (hand-written steps)
    BLOCK_HEAD = 1,      // lanes start inside it, none finishes (hand-written steps + the rim select)
    BLOCK_MASKED = 2     // anything: lanes start and / or finish inside it (C++, per-lane predicates)
};
template <int KK, bool BETA, int MODE, bool SEEDED>
__device__ __forceinline__ void compute_block_ip(f32x4 (&cur2)[KK / 2], f32x4 (&seed4)[KK / 4], const unsigned nsrc, const unsigned nseed,
                                                 float& Y, float& X, const int d0, const int ucol_chk, const int Tn,
                                                 lds_float* vslot, const int first_off) {
    static_assert(KK == 8 || KK == 16, "blocks of 8 or 16 diagonals");
    // the seeds' LDS address in a VGPR the compiler cannot see through: it is wave-uniform, and left to itself the
    // compiler keeps it in an SGPR and copies it into a VGPR in front of EVERY reload (one v_mov per diagonal)
    unsigned nseed_v = nseed;
    if constexpr (SEEDED) asm volatile("" : "+v"(nseed_v));
    if constexpr (MODE != BLOCK_MASKED) {
        if constexpr (BETA && MODE == BLOCK_HEAD) {
            // (one state register in the hand-written beta steps: lane 0 of the first column block starts from Y = 0, the
            //  value it hands right before its first diagonal is read by nobody -- make the two one)
            X = Y;
        }
        fast_steps<KK, 0, BETA, SEEDED, MODE == BLOCK_HEAD>(cur2, seed4, nsrc, nseed_v, Y, X, vslot, 1.44269504088896340736f,
                                                          1ull, 1ull << (first_off & 63));
        if constexpr (BETA) Y = X;
        return;
    }
#define RNNT_PIN() __builtin_amdgcn_sched_barrier(0)
    wait_lds();   // (a hand-written block in front of this one leaves its last reloads in flight)
    // (the order of the step is compute_block's: no hazard nops, see there)
    const int ucol_first = first_diag_of(ucol_chk);
    const bool col0 = ucol_chk == 0;
    float pval = 0.0f;
#pragma unroll
    for (int k = 0; k < KK; ++k) {
        const float cx = (k & 1) ? cur2[k / 2].z : cur2[k / 2].x;
        const float cy = (k & 1) ? cur2[k / 2].w : cur2[k / 2].y;
        float skip, emit;
        if constexpr (BETA) {
            if (k > 0) {
                vslot[(k - 1) * WAVE] = pval;
                RNNT_PIN();
            }
            skip = Y + cx;
            RNNT_PIN();
        } else {
            skip = Y;
        }
        float left;
        if constexpr (SEEDED) left = wave_shr1(seed4[k / 4][k % 4], X); else left = wave_shr1_z(X);   // chain
        RNNT_PIN();
        if constexpr (BETA) {
            emit = left + cy;                                                          // chain
        } else {
            emit = left;
        }
        RNNT_PIN();
        const float t = skip - emit;                                                   // chain
        RNNT_PIN();
        const bool rim = d0 + k == ucol_first;                                         // (compared here, two wait states
        RNNT_PIN();                                                                    //  ahead of the select that reads it)
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
        const float um1 = u - 1.0f;
        RNNT_PIN();
        const float c = e - um1;
        RNNT_PIN();
        const float l = __builtin_fmaf(l2, 0.693147180559945309417f, c);               // chain
        RNNT_PIN();
        float val = mx + l;                                                            // chain
        RNNT_PIN();
        val = rim ? emit : val;                                                        // rim: first live diagonal
        RNNT_PIN();
        if constexpr (SEEDED) {
            if (k % 4 == 3) {   // next block's boundary values k - 3 ... k (their registers are spent)
                switch (k) {
                    case 3: lds_reload_b128<0>(seed4[k / 4], nseed_v); break;   case 7: lds_reload_b128<16>(seed4[k / 4], nseed_v); break;
                    case 11: lds_reload_b128<32>(seed4[k / 4], nseed_v); break; default: lds_reload_b128<48>(seed4[k / 4], nseed_v); break;
                }
                RNNT_PIN();
            }
        }
        if constexpr (!SEEDED) { val = col0 ? skip : val; RNNT_PIN(); }                // rim: sweep column 0
        float Yn, Xn;
        if constexpr (BETA) {
            Yn = val; Xn = val;
        } else {
            Xn = val + cy;                                                             // chain (feeds the DPP)
            RNNT_PIN();
            vslot[k * WAVE] = val;
            RNNT_PIN();
            Yn = val + cx;
            RNNT_PIN();
        }
        if (k & 1) {                                                                   // next block's pairs k - 1, k
            switch (k) {
                case 1: lds_reload_2rows<0>(cur2[k / 2], nsrc); break;   case 3: lds_reload_2rows<2>(cur2[k / 2], nsrc); break;
                case 5: lds_reload_2rows<4>(cur2[k / 2], nsrc); break;   case 7: lds_reload_2rows<6>(cur2[k / 2], nsrc); break;
                case 9: lds_reload_2rows<8>(cur2[k / 2], nsrc); break;   case 11: lds_reload_2rows<10>(cur2[k / 2], nsrc); break;
                case 13: lds_reload_2rows<12>(cur2[k / 2], nsrc); break; default: lds_reload_2rows<14>(cur2[k / 2], nsrc); break;
            }
            RNNT_PIN();
        }
        const bool live = (unsigned)(d0 + k - ucol_chk) < (unsigned)Tn;
        Y = live ? Yn : Y;
        X = live ? Xn : X;
        pval = val;
    }
    if constexpr (BETA) vslot[(KK - 1) * WAVE] = pval;
#undef RNNT_PIN
}

}  // namespace ws

}  // namespace rnnt
