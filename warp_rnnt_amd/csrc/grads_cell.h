// The gradient of one lattice cell and the per-utterance guard, shared by k_grads (grads.hip: one thread per cell in
// diagonal-major order) and by the dense-row writer that computes its two slots straight from the planes (expand.hip:
// GradRows, small lattices) -- one definition, so both give the same bits.
// Maths: core_gather.cu:248-284 (blank), :286-319 (label + FastEmit), :321-357 (cost and forward/backward guard).
#pragma once
#include "common.h"

namespace rnnt {

struct UttGuard {
    float b00;    // beta[0,0]
    float ll_a;   // alpha-side log-likelihood (written by the alpha sweep)
    bool bad;     // the two log-likelihoods differ by more than 1e-3 relative, or the lengths are out of range
};
__device__ __forceinline__ UttGuard utt_guard(float b00, float ll_a, bool len_ok) {
    const float ratio = fabsf(ll_a - b00) / fabsf(fmaxf(ll_a, b00));
    return UttGuard{b00, ll_a, ratio > 0.001f || !len_ok};
}
__device__ __forceinline__ float utt_cost(const UttGuard& g, bool len_ok) {
    return !len_ok ? __builtin_nanf("") : g.bad ? -((g.ll_a + g.b00) / 2.0f) : -g.b00;
}

// The reference prints from the device whenever the guard fires ("WARNING: sample %d [%d, %d] has a forward/backward
// mismatch %f / %f", core_gather.cu:345-349).  Here the same facts go to the device's sticky diagnostics words -- eight
// words of pinned host memory (rnnt_amd_mismatch_flag, include/warp_rnnt_amd.h), written ONLY when a guard fires, read
// by the host whenever it likes: no printf in the kernel, no host synchronisation anywhere, nothing at all on the
// common path.  w == nullptr: nobody asked for the diagnostics yet.  Concurrent firings may interleave their details
// (diagnostics: the last writer wins); w[0] is written last.
__device__ __forceinline__ void report_guard(unsigned* w, int n, int xn_raw, int yn_raw, const UttGuard& g, bool len_ok) {
    if (!w) return;
    volatile unsigned* v = w;
    v[1] = len_ok ? 1u : 2u;                       // 1 = forward/backward mismatch, 2 = lengths out of range
    v[2] = (unsigned)n;
    v[3] = (unsigned)xn_raw;
    v[4] = (unsigned)yn_raw;
    v[5] = __builtin_bit_cast(unsigned, g.ll_a);   // alpha-side log-likelihood
    v[6] = __builtin_bit_cast(unsigned, g.b00);    // beta[0,0]
    __threadfence_system();
    v[0] = 1u;                                     // "something fired since you last cleared this"
}

// (gB, gL) of live cell (t,u) of an utterance with Tn frames and Un columns; beta_next(c) = beta of the next diagonal
// at column c, i.e. beta[t+1,u] for c = u and beta[t,u+1] for c = u+1 (read only where the formula needs it).
template <typename BetaNext>
__device__ __forceinline__ float2 cell_grads(float alpha, float lpB, float lpL, float b00, int t, int u, int Tn, int Un,
                                             float fastemit_lambda, BetaNext beta_next) {
    float gB = 0.0f, gL = 0.0f;
    if (t < Tn - 1) {
        const float x = alpha + beta_next(u);
        gB = -expf(x + lpB - b00);
    } else if (u == Un - 1) {
        gB = -expf(alpha + lpB - b00);
    }
    if (u < Un - 1) {
        const float x = alpha + beta_next(u + 1);
        const float e = expf(x + lpL - b00);
        gL = -(float)((1. + fastemit_lambda) * e);
    }
    return make_float2(gB, gL);
}

}  // namespace rnnt
