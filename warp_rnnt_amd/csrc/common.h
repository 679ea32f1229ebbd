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
__device__ __forceinline__ int skew_row(int t, int u, int T) {
    int r = t + u;          // t < T, u < U ; callers guarantee r < 2T or reduce first
    r = r >= T ? r % T : r;
    return r;
}

// log(exp(a) + exp(b)) with the reference's structure (core.cu:26-39):
// max + log1p(exp(-|a-b|)).  The transcendental part uses the hardware
// v_exp_f32 / v_log_f32 units with a first-order correction for the rounding
// of 1+e, instead of the ~150-instruction ocml expf+log1pf pair: this is the
// latency-critical dependency chain of the whole op.
//   e = 2^(d*log2(e)), d <= 0          abs. error <~ 1e-8 (shrinks as fast as e)
//   log1p(e) = ln(u) + (e-(u-1)), u = fl(1+e)     ~2 ulp of a value <= ln 2
// Both are far below the fp32 rounding of the final `max + ...` whenever
// |max| >= 1; see tests/test_lattice_gpu.py for the measured deviation from
// the libm-based oracle.  Define RNNT_PRECISE_LIBM to use ocml instead.
__device__ __forceinline__ float lse(float a, float b) {
#if defined(RNNT_PRECISE_LIBM)
    const bool gt = a > b;
    const float mx = gt ? a : b;
    const float d = (gt ? b : a) - mx;
    return mx + log1pf(expf(d));
#elif defined(RNNT_PROBE_NOLSE)
    return fmaxf(a, b);
#else
    // diff = -|a-b| is the same value whichever operand is larger; the abs/neg ride on the
    // multiply as source modifiers, and max(a,b) is off the dependency chain until the last add.
    const float t = a - b;
    const float mx = __builtin_fmaxf(a, b);
    const float e = __builtin_amdgcn_exp2f(-__builtin_fabsf(t) * 1.44269504088896340736f);
    const float u = 1.0f + e;
    const float c = e - (u - 1.0f);                       // exact rounding error of 1+e
    const float l2 = __builtin_amdgcn_logf(u);            // log2(u), u in [1,2]
#ifdef RNNT_LSE_RCP
    const float l = __builtin_fmaf(l2, 0.693147180559945309417f, c * __builtin_amdgcn_rcpf(u));
#else
    // log(1+e) = log(u) + log1p(c/u) ~ log(u) + c/u; c <= 2^-24 and 1/u in [0.5,1], so using c
    // for c/u is off by < 3e-8 absolute -- a quarter ulp of anything >= 1 it gets added to.
    const float l = __builtin_fmaf(l2, 0.693147180559945309417f, c);
#endif
    return mx + l;
#endif
}

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
