// Shared device helpers for the MI355X (gfx950) RNN-T loss kernels.
// gfx950 only: wave64, DPP wave_shr, v_exp_f32 / v_log_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rnnt {

constexpr int WAVE = 64;

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

__device__ __forceinline__ float readlane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// Per-utterance lengths as the kernels use them.  The reference never checks 1 <= xn <= T and
// 0 <= yn <= U-1 (binding.cpp:47-51; xn = 0 reads index -1, core.cu:343).  Checking on the host would
// cost a device sync, so the check lives here: an utterance with out-of-range lengths is swept as a
// one-cell lattice (no out-of-range access) and k_grads reports cost = NaN with zero gradients.
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
