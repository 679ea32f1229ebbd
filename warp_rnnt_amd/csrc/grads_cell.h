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
