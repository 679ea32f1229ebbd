// Gradients of the RNN-T cost w.r.t. the log-probabilities + per-utterance costs.
//
// Maths (reference: core_gather.cu:248-284 blank, :286-319 label + FastEmit,
// :321-357 cost and forward/backward guard; dense write order core.cu:382-393):
//   gB[t,u] = -exp((alpha[t,u] + beta[t+1,u]) + lpB[t,u] - beta[0,0])     t <  T_n-1
//   gB[T_n-1,U_n-1] = -exp(alpha + lpB - beta[0,0]);  other cells of the last frame: 0
//   gL[t,u] = -(float)((1.0 + lambda) * exp((alpha[t,u] + beta[t,u+1]) + lpL[t,u] - beta[0,0])),  u < U_n-1
//   cost[n] = -beta[0,0]   (guard: if alpha-side and beta-side log-likelihoods differ by
//                           more than 1e-3 relative, grads of the sample are zeroed and the
//                           cost is the mean of the two, like the reference)
//
// One fused element-wise kernel (the reference uses three launches with lanes
// along t).  Threads walk the lattice in diagonal-major order, so alpha,
// beta[t+1,u], beta[t,u+1] (= next diagonal, columns u and u+1) and the
// workspace log-probs are all coalesced row reads.
#include <atomic>
#include <cstring>

#include "common.h"
#include "grads_cell.h"
#include "kernels.h"

namespace rnnt {

template <int LOADER, int WRITER, bool COMPACT>
__global__ void __launch_bounds__(256) k_grads(const GradArgs a) {
    const int n = blockIdx.y;
    const UttLens len = utt_lens<COMPACT>(a.xn, a.yn, n, a.T, a.U);
    if (COMPACT && !len.ok) {          // packed layout: the utterance owns no cells, only its cost is reported
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            a.costs[n] = __builtin_nanf("");
            if (a.mismatch) a.mismatch[n] = 1;
            report_guard(a.sticky, n, a.xn[n], a.yn[n], UttGuard{0.0f, 0.0f, true}, false);
        }
        return;
    }
    const int Tn = len.Tn, Un = len.Un;
    const int T = COMPACT ? Tn : a.T, U = COMPACT ? Un : a.U;     // compact: per-utterance planes
    const size_t nb = COMPACT ? compact_base(a, n) : (size_t)n * T * U;
    const float* __restrict__ al = a.alphas + nb;
    const float* __restrict__ be = a.betas + nb;

    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    const bool in = idx < (unsigned)(T * U);

    // ---- per-utterance scalars: beta[0,0], alpha-side log-likelihood, guard ----
    const UttGuard guard = utt_guard(be[0], a.ll[n], len.ok);   // be[0]: sk(0,0) = 0
    const float b00 = guard.b00;
    const bool bad = guard.bad;
    if (idx == 0) {
        a.costs[n] = utt_cost(guard, len.ok);
        if (a.mismatch) a.mismatch[n] = bad ? 1 : 0;
        if (bad) report_guard(a.sticky, n, a.xn[n], a.yn[n], guard, len.ok);
    }
    if (!in) return;

    const int r = idx / (unsigned)U;
    const int u = idx - r * U;
    int t = (r - u) % T;
    if (t < 0) t += T;
    const bool valid = (t < Tn) && (u < Un);

    float gB = 0.0f, gL = 0.0f;
    if (valid && !bad) {
        float lpB, lpL;
        if constexpr (LOADER == LOAD_SKEWED) {
            const float2 v = reinterpret_cast<const float2*>(a.lp)[nb + idx];
            lpB = v.x; lpL = v.y;
        } else if constexpr (LOADER == LOAD_ROWMAJOR2) {
            const float2 v = reinterpret_cast<const float2*>(a.lp)[nb + (size_t)t * U + u];
            lpB = v.x; lpL = v.y;
        } else {
            const float* p = a.lp + (nb + (size_t)t * U + u) * (size_t)a.V;
            lpB = p[a.blank];
            lpL = (u < Un - 1) ? p[safe_label(a.labels[(size_t)n * (U - 1) + u], a.V, a.blank)] : 0.0f;
        }
        const float alpha = al[idx];
        const int r1 = (r + 1 == T) ? 0 : r + 1;
        const float2 g = cell_grads(alpha, lpB, lpL, b00, t, u, Tn, Un, a.fastemit_lambda,
                                    [&](int c) { return be[(size_t)r1 * U + c]; });
        gB = g.x;
        gL = g.y;
    }

    if constexpr (WRITER == WRITE_SKEWED2) {
        reinterpret_cast<float2*>(a.grads)[nb + idx] = make_float2(gB, gL);
    } else if constexpr (WRITER == WRITE_ROWMAJOR2) {
        reinterpret_cast<float2*>(a.grads)[nb + (size_t)t * U + u] = make_float2(gB, gL);
    } else {
        // reference C-ABI contract: grads pre-zeroed by the caller, only live slots are written;
        // blank first, label second (a label equal to blank overwrites, core.cu:382-393)
        if (valid) {
            float* g = a.grads + (nb + (size_t)t * U + u) * (size_t)a.V;
            if (t < Tn - 1 || u == Un - 1) g[a.blank] = gB;
            if (u < Un - 1) g[safe_label(a.labels[(size_t)n * (U - 1) + u], a.V, a.blank)] = gL;
        }
    }
}

template <int LOADER>
static hipError_t launch_grads_w(hipStream_t stream, const GradArgs& a, dim3 grid, int writer) {
    switch (writer) {
        case WRITE_SKEWED2:   k_grads<LOADER, WRITE_SKEWED2, false><<<grid, 256, 0, stream>>>(a); break;
        case WRITE_ROWMAJOR2: k_grads<LOADER, WRITE_ROWMAJOR2, false><<<grid, 256, 0, stream>>>(a); break;
        default:              k_grads<LOADER, WRITE_DENSE_SLOTS, false><<<grid, 256, 0, stream>>>(a); break;
    }
    return hipGetLastError();
}

// ---- the sticky diagnostics words (kernels.h) ----
namespace {
constexpr int MAX_DEVICES = 64, WORDS = 16;     // (8 used; a 64-byte line per device)
std::atomic<unsigned*> g_words{nullptr};
}
unsigned* mismatch_words(int device, bool allocate) {
    if (device < 0 || device >= MAX_DEVICES) return nullptr;
    unsigned* base = g_words.load(std::memory_order_acquire);
    if (!base && allocate) {
        void* p = nullptr;
        if (hipHostMalloc(&p, sizeof(unsigned) * WORDS * MAX_DEVICES, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        memset(p, 0, sizeof(unsigned) * WORDS * MAX_DEVICES);
        unsigned* expected = nullptr;
        if (g_words.compare_exchange_strong(expected, static_cast<unsigned*>(p), std::memory_order_acq_rel)) base = static_cast<unsigned*>(p);
        else { (void)hipHostFree(p); base = expected; }      // another thread was first
    }
    return base ? base + (size_t)device * WORDS : nullptr;
}
unsigned* mismatch_words_of(hipStream_t stream) {
    if (!g_words.load(std::memory_order_acquire)) return nullptr;      // (the common case until somebody asks: no runtime call)
    int dev = -1;
    if (hipStreamGetDevice(stream, &dev) != hipSuccess && hipGetDevice(&dev) != hipSuccess) return nullptr;
    return mismatch_words(dev, false);
}

hipError_t launch_grads(hipStream_t stream, const GradArgs& a0, int N, int loader, int writer) {
    if (N <= 0) return hipSuccess;
    GradArgs a = a0;
    a.sticky = mismatch_words_of(stream);
    const dim3 grid(((unsigned)(a.T * a.U) + 255u) / 256u, (unsigned)N);
    if (is_compact(a)) {   // compact layout (a.T, a.U = maxima): packed pairs out
        if (loader == LOAD_ROWMAJOR2) {   // row-major packed pairs in (core.h shims)
            k_grads<LOAD_ROWMAJOR2, WRITE_ROWMAJOR2, true><<<grid, 256, 0, stream>>>(a);
            return hipGetLastError();
        }
        if (writer == WRITE_ROWMAJOR2)
            k_grads<LOAD_SKEWED, WRITE_ROWMAJOR2, true><<<grid, 256, 0, stream>>>(a);
        else
            k_grads<LOAD_SKEWED, WRITE_SKEWED2, true><<<grid, 256, 0, stream>>>(a);
        return hipGetLastError();
    }
    switch (loader) {
        case LOAD_SKEWED:    return launch_grads_w<LOAD_SKEWED>(stream, a, grid, writer);
        case LOAD_ROWMAJOR2: return launch_grads_w<LOAD_ROWMAJOR2>(stream, a, grid, writer);
        default:             return launch_grads_w<LOAD_DENSE>(stream, a, grid, writer);
    }
}

// cost[n] = -beta[0,0] for a compact batch whose alpha sweep was skipped (run_warp_rnnt_compact with
// required_grad = false; core_compact.cu:349-357).  Cell (0,0) is the first cell of the diagonal-major plane too.
__global__ void k_costs_from_betas(const float* __restrict__ betas, const unsigned* __restrict__ mem_pref,
                                   const int* __restrict__ xn, const int* __restrict__ yn, float* __restrict__ costs,
                                   int N) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const bool ok = xn[n] >= 1 && yn[n] >= 0;
    costs[n] = ok ? -betas[mem_pref[n]] : __builtin_nanf("");
}

hipError_t launch_costs_from_betas(hipStream_t stream, const float* betas, const unsigned* mem_pref, const int* xn,
                                   const int* yn, float* costs, int N) {
    if (N <= 0) return hipSuccess;
    k_costs_from_betas<<<(N + 255) / 256, 256, 0, stream>>>(betas, mem_pref, xn, yn, costs, N);
    return hipGetLastError();
}

}  // namespace rnnt
