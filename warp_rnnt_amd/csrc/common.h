// Shared device helpers for the MI355X (gfx950) RNN-T loss kernels.
// gfx950 only: wave64, DPP wave_shr, v_exp_f32 / v_log_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

namespace rnnt {

constexpr int WAVE = 64;

// Kernel-selection knobs for A/B timing runs (which log-softmax cover, which tile order, ...) are read
// from the environment ONLY in the build made for that -- -DRNNT_AB_KNOBS, the `ab` variant of warp_rnnt_amd/_build.py,
// which the probes under tools/ load through WARP_RNNT_AMD_LIB.  The shipped library ignores them: what it does is a
// function of the call's arguments (round 6; until then ~20 variables were read once per process by every build).
inline const char* ab_getenv(const char* name) {
#ifdef RNNT_AB_KNOBS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// ---------------------------------------------------------------------------
// Diagonal-major ("skewed") lattice layout.
//
// A per-utterance (T,U) plane is stored so that a whole anti-diagonal
// d = t + u is one contiguous row:   sk(t,u) = ((t + u) mod T) * U + u.
// For fixed u the map t -> (t+u) mod T is a bijection on [0,T), so the plane
// still has exactly T*U cells (it fits the reference's (N,T,U) scratch,
// binding.cpp:69-75).  The alpha/beta sweep at diagonal d touches row d mod T
// only: every global access of the serial part of the algorithm is a
// coalesced row segment, and the address needs no per-lane t.
// ---------------------------------------------------------------------------

// Lane i receives src from lane i-1 of the same wave64; lane 0 receives `first`.
// One v_mov_b32_dpp ... wave_shr:1 (the CDNA form of a 64-lane shuffle-up by one).
__device__ __forceinline__ float wave_shr1(float first, float src) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first),
                                           __builtin_bit_cast(int, src),
                                           0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}

// The same shift with lane 0 receiving 0.0 (bound_ctrl: a lane without a source reads zero): no `old` operand, hence no
// v_mov in front of every shift to seed it.  For the column block that has no left neighbour -- its lane 0 is lattice
// column 0 (sweep coordinates), whose value never comes out of the lse anyway (boundary_fix in lattice_step.h).
__device__ __forceinline__ float wave_shr1_z(float src) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}

__device__ __forceinline__ float readlane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// Per-utterance lengths as the kernels use them.  The reference never checks 1 <= xn <= T and
// 0 <= yn <= U-1 (binding.cpp:47-51; xn = 0 reads index -1, core.cu:343).  Checking on the host would
// cost a device sync, so the check lives here: an utterance with out-of-range lengths is swept as a
// one-cell lattice (no out-of-range access) and k_grads reports cost = NaN with zero gradients.
// Which chunk of a streaming kernel's work an XCD takes.  Workgroups go to the eight XCDs by blockIdx mod 8: with
// chunk = blockIdx every XCD works on every eighth chunk of one moving front; stream_block() renumbers the workgroups so
// that XCD k streams the k-th contiguous eighth (grid = stream_grid(chunks): a multiple of 8; renumbered blocks beyond the
// last chunk return).  Per kernel family, bit of RNNT_XCD_STREAM (measured per family: profiles/r04_lsm_xcd_order_ab.txt).
#ifndef RNNT_XCD_STREAM
#define RNNT_XCD_STREAM 15      // all four: c4, three interleaved pairs against -DRNNT_XCD_STREAM=0: fused forward 0.467 -> 0.464 ms,
#endif                          // fused training step 0.978 -> 0.963, native log-softmax chain 1.86 -> 1.82, run_warp_rnnt 0.893 -> 0.879
enum : int { XCD_LSM_SMALL = 1, XCD_EXPAND_SMALL = 2, XCD_LSMBWD_SMALL = 4, XCD_LSM_ROWS = 8 };
template <int FAMILY> __device__ __forceinline__ unsigned stream_block() {
    if constexpr ((RNNT_XCD_STREAM & FAMILY) != 0) return (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    else return blockIdx.x;
}
template <int FAMILY> inline unsigned stream_grid(unsigned chunks) {
    return (RNNT_XCD_STREAM & FAMILY) != 0 ? (chunks + 7u) & ~7u : chunks;
}

struct UttLens {
    int Tn, Un;
    bool ok;
};
template <bool COMPACT>
__device__ __forceinline__ UttLens utt_lens(const int* xn, const int* yn, int n, int T, int U) {
    const int x = xn[n], y = yn[n];
    UttLens l;
    l.ok = COMPACT ? (x >= 1 && y >= 0) : (x >= 1 && x <= T && y >= 0 && y < U);
    l.Tn = l.ok ? x : 1;
    l.Un = l.ok ? y + 1 : 1;
    return l;
}

// Single-column lattice (U_n == 1, an utterance without labels): alpha[t,0] / beta[t,0] are plain prefix /
// suffix sums of the blank log-probs.  The general sweep would accumulate them serially -- T_n dependent
// fp32 additions, a random walk of ~0.3*sqrt(T_n) ulp that the two directions do not share, which shows up
// one to one in exp(alpha + beta - loglik) (6e-3 at T_n = 970, |loglik| = 2270).  The reference computes these
// boundary chains with 32-wide shuffle scans plus a per-tile carry (core_gather.cu:86-104, 187-205); this does
// the same with one wave: 64-wide inclusive scan per chunk, serial carry across chunks.
// lpB(t) returns the blank log-prob of cell (t,0); out[t * pitch] receives the value.  Returns the total.
template <bool BETA, typename F>
__device__ __forceinline__ float single_column_scan(const int Tn, float* __restrict__ out, const int pitch,
                                                    const int lane, F lpB) {
    float carry = 0.0f;
    for (int c0 = 0; c0 < Tn; c0 += WAVE) {
        const int k = c0 + lane;                          // position in sweep order
        const int t = BETA ? (Tn - 1 - k) : k;
        const float v = k < Tn ? lpB(t) : 0.0f;
        float s = v;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const float y = __shfl_up(s, o, WAVE);
            if (lane >= o) s += y;
        }
        // alpha[t,0] excludes the own cell (sum over frames before t); beta[t,0] includes it
        float ex = __shfl_up(s, 1, WAVE);                 // exclusive prefix (exact, no s - v cancellation)
        if (lane == 0) ex = 0.0f;
        if (k < Tn) out[(size_t)t * pitch] = carry + (BETA ? s : ex);
        carry += readlane(s, WAVE - 1);
    }
    return carry;
}

// Label of a lattice column as a vocabulary index that is always safe to address with.  Entries beyond
// yn[n] are padding: the reference's dense path and C ABI never read them (kernel_grads_label returns
// early, core.cu:305-309), so callers pad with -1 or any sentinel.  Nothing a padded column reads or
// writes reaches a result (its cells are dead), so it is simply pointed at the blank slot.
__device__ __forceinline__ int safe_label(int lab, int V, int blank) {
    return (unsigned)lab < (unsigned)V ? lab : blank;
}

// How a kernel finds the blank / label log-probability of lattice cell (t,u).
enum Loader : int {
    LOAD_SKEWED = 0,   // float2 workspace, diagonal-major (internal layout)
    LOAD_ROWMAJOR2 = 1,// (N,T,U,2) row-major, ch0 blank / ch1 label (reference "gather" layout)
    LOAD_DENSE = 2     // (N,T,U,V) row-major + labels (reference dense layout)
};

// Where the gradient kernel puts d(cost)/d(log_probs).
enum Writer : int {
    WRITE_SKEWED2 = 0,   // float2, diagonal-major (internal, consumed by the expand kernel)
    WRITE_ROWMAJOR2 = 1, // (N,T,U,2) row-major (what _C.rnnt_loss(blank=-1) returns)
    WRITE_DENSE_SLOTS = 2// two slots of a pre-zeroed (N,T,U,V) tensor (reference C ABI contract)
};

}  // namespace rnnt
