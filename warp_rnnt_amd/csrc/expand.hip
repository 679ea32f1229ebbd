// Expansion of the (blank,label) gradient pairs into dense (N,T,U,V) rows.
//
// Serves two callers:
//   * gather=True backward: d(loss)/d(log_probs) = scatter_add of the gathered
//     gradients scaled by grad_output[n] into a zero tensor -- the autograd
//     backward of torch.gather in warp_rnnt/__init__.py:126 preceded by the
//     in-place scaling of __init__.py:23-24.  (mode 0, "sum")
//   * gather=False forward: the dense gradient tensor the native op returns
//     (binding.cpp:58 zeros_like + core.cu:260-332 slot writes, where the label
//     kernel runs after the blank kernel and overwrites).  (mode 1, "overwrite")
// Either way the reference pays a full zero-fill pass plus scattered 4-byte
// writes; here every output row is produced once, zeros included, with
// 16-byte coalesced stores, and the int64 index tensor never exists.
#include "common.h"
#include "kernels.h"

namespace rnnt {

template <int VEC>
__global__ void __launch_bounds__(256)
k_expand(const float2* __restrict__ g2, const int* __restrict__ labels, const int* __restrict__ xn,
         const int* __restrict__ yn, const float* __restrict__ scale, float* __restrict__ dense,
         int T, int U, int V, int blank, int overwrite) {
    const size_t frame = blockIdx.x;                 // n*T + t
    const unsigned UV = (unsigned)U * (unsigned)V;
    const unsigned e0 = (blockIdx.y * 256u + threadIdx.x) * VEC;
    if (e0 >= UV) return;
    const size_t n = frame / (unsigned)T;
    const int t = (int)(frame - n * (unsigned)T);
    const float sc = scale ? scale[n] : 1.0f;
    const int Tn = xn[n], Un = yn[n] + 1;

    int u = e0 / (unsigned)V;
    int v = e0 - u * V;
    float out[VEC];
    int cu = -1;
    float gB = 0.f, gL = 0.f;
    int lab = -1;
    bool labvalid = false;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        if (u != cu && u < U) {
            cu = u;
            int r = t + u;
            r = r >= T ? r % T : r;
            const float2 g = g2[(n * T + r) * (size_t)U + u];
            gB = g.x * sc; gL = g.y * sc;
            lab = (u < U - 1) ? labels[n * (size_t)(U - 1) + u] : blank;
            labvalid = (t < Tn) && (u < Un - 1);
        }
        float val;
        if (overwrite)
            val = (v == lab && labvalid) ? gL : (v == blank ? gB : 0.0f);
        else
            val = (v == blank ? gB : 0.0f) + (v == lab ? gL : 0.0f);
        out[j] = val;
        if (++v == V) { v = 0; ++u; }
    }
    float* dst = dense + frame * (size_t)UV + e0;
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(dst) = make_float4(out[0], out[1], out[2], out[3]);
    } else {
        dst[0] = out[0];
    }
}

hipError_t launch_expand(hipStream_t stream, const float* g2_skewed, const int* labels, const int* xn,
                         const int* yn, const float* scale, float* dense, int N, int T, int U, int V,
                         int blank, int overwrite_mode) {
    const size_t frames = (size_t)N * T;
    if (frames == 0 || U == 0 || V == 0) return hipSuccess;
    const unsigned UV = (unsigned)U * (unsigned)V;
    const bool vec = (UV % 4 == 0) && (reinterpret_cast<uintptr_t>(dense) % 16 == 0);
    if (vec) {
        const dim3 grid((unsigned)frames, (UV / 4 + 255) / 256);
        k_expand<4><<<grid, 256, 0, stream>>>(reinterpret_cast<const float2*>(g2_skewed), labels, xn, yn,
                                              scale, dense, T, U, V, blank, overwrite_mode);
    } else {
        const dim3 grid((unsigned)frames, (UV + 255) / 256);
        k_expand<1><<<grid, 256, 0, stream>>>(reinterpret_cast<const float2*>(g2_skewed), labels, xn, yn,
                                              scale, dense, T, U, V, blank, overwrite_mode);
    }
    return hipGetLastError();
}

}  // namespace rnnt
