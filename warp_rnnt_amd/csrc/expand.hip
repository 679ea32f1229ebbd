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
//
// The kernels are pure HBM writers (4V+8 bytes per cell), so the work per
// element is kept to a fraction of an instruction: small vocabularies build a
// tile of whole rows in LDS (zero fill with ds_write_b128, two 4-byte patches
// per cell by one lane, ds_read_b128 + 16-byte stores out); large vocabularies
// write zeros straight from registers and compare the column index against the
// cell's two slots.  (The first version -- one thread per float4 with its own
// index divisions -- was VALU-bound at 2.4 TB/s.)
#include "common.h"
#include "grads_cell.h"
#include "kernels.h"

namespace rnnt {

// cache policy of the dense-row writes (a pure writer of 4V bytes per cell): 1 = non-temporal stores (round 3: the
// gather=True training step through the native log-softmax function 1.96 -> 1.90 ms at c4, profiles/r03_bwd_nt_ab.txt)
#ifndef RNNT_EX_NT
#define RNNT_EX_NT 1
#endif
typedef float ex_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ex_store4(float4* p, float4 v) {
    if (RNNT_EX_NT) {
        const ex_f4 w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<ex_f4*>(p));
    } else {
        *p = v;
    }
}

struct ExpandCell {
    float gB, gL;   // scaled gradients of the blank and label slot
    int lab;        // vocabulary index of the label slot, or -1 when nothing goes there
};

// cell = flat (n,t,u) index.  N*T*U < 2^32 is checked by the C ABI.
__device__ __forceinline__ ExpandCell expand_cell(unsigned cell, const float2* __restrict__ g2,
                                                  const int* __restrict__ labels, const int* __restrict__ xn,
                                                  const int* __restrict__ yn, const float* __restrict__ scale,
                                                  int T, int U, int V, int blank, int overwrite) {
    const unsigned frame = cell / (unsigned)U;
    const int u = (int)(cell - frame * (unsigned)U);
    const unsigned n = frame / (unsigned)T;
    const int t = (int)(frame - n * (unsigned)T);
    int r = t + u;
    r = r >= T ? r % T : r;
    const float sc = scale ? scale[n] : 1.0f;
    const float2 g = g2[((size_t)n * T + r) * (size_t)U + u];
    ExpandCell c;
    c.gB = g.x * sc;
    c.gL = g.y * sc;
    c.lab = (u < U - 1) ? labels[(size_t)n * (U - 1) + u] : blank;
    if ((unsigned)c.lab >= (unsigned)V) c.lab = -1;   // padding beyond yn[n] (-1 or any sentinel): its gL is 0
    if (overwrite) {
        // dense-kernel rule (core.cu:382-393): the label slot is written only for live cells with a
        // label, after the blank slot
        const bool labvalid = (t < xn[n]) && (u < yn[n]);
        if (!labvalid) c.lab = -1;
        else if (c.lab == blank) c.gB = c.gL;      // label write lands on the blank slot and wins
        if (labvalid && c.lab == blank) c.lab = -1;
    }
    return c;
}

// Row sources: where row `i` of the dense output gets its two values from.
struct PaddedRows {    // (N,T,U,V) output, diagonal-major pairs (gather backward / dense forward)
    const float2* g2; const int* labels; const int* xn; const int* yn; const float* scale;
    int T, U, V, blank, overwrite;
    __device__ __forceinline__ ExpandCell operator()(unsigned row) const {
        return expand_cell(row, g2, labels, xn, yn, scale, T, U, V, blank, overwrite);
    }
};
struct SplitRows {     // as PaddedRows, the two channels in two float planes (the reference-named dense C entry, api.hip)
    const float* ga; const float* gb; const int* labels; const int* xn; const int* yn;
    int T, U, V, blank;
    __device__ __forceinline__ ExpandCell operator()(unsigned cell) const {
        const unsigned frame = cell / (unsigned)U;
        const int u = (int)(cell - frame * (unsigned)U);
        const unsigned n = frame / (unsigned)T;
        const int t = (int)(frame - n * (unsigned)T);
        int r = t + u;
        r = r >= T ? r % T : r;
        const size_t at = ((size_t)n * T + r) * (size_t)U + u;
        ExpandCell c;
        c.gB = ga[at];
        c.gL = gb[at];
        c.lab = (u < U - 1) ? labels[(size_t)n * (U - 1) + u] : blank;
        if ((unsigned)c.lab >= (unsigned)V) c.lab = -1;
        // dense-kernel rule (core.cu:382-393), as PaddedRows in overwrite mode
        const bool labvalid = (t < xn[n]) && (u < yn[n]);
        if (!labvalid) c.lab = -1;
        else if (c.lab == blank) c.gB = c.gL;
        if (labvalid && c.lab == blank) c.lab = -1;
        return c;
    }
};
struct GradRows {      // the dense forward result computed straight from the planes: k_grads and the expansion in one launch
    // (small lattices, whose planes sit in L2: the cell's seven dwords are scattered reads here, one lane per row)
    const float2* lp2; const float* alphas; const float* betas; const float* ll;
    const int* labels; const int* xn; const int* yn; float* costs; int* mismatch;
    int T, U, V, blank; float fastemit_lambda; unsigned* sticky;
    __device__ __forceinline__ ExpandCell operator()(unsigned cell) const {
        const unsigned frame = cell / (unsigned)U;
        const int u = (int)(cell - frame * (unsigned)U);
        const unsigned n = frame / (unsigned)T;
        const int t = (int)(frame - n * (unsigned)T);
        const UttLens len = utt_lens<false>(xn, yn, (int)n, T, U);
        const size_t nb = (size_t)n * T * U;
        const float* __restrict__ be = betas + nb;
        const UttGuard guard = utt_guard(be[0], ll[n], len.ok);
        if (t == 0 && u == 0) {
            costs[n] = utt_cost(guard, len.ok);
            if (mismatch) mismatch[n] = guard.bad ? 1 : 0;
            if (guard.bad) report_guard(sticky, (int)n, xn[n], yn[n], guard, len.ok);
        }
        int r = t + u;
        r = r >= T ? r % T : r;
        ExpandCell c;
        c.gB = 0.0f;
        c.gL = 0.0f;
        if (t < len.Tn && u < len.Un && !guard.bad) {
            const size_t idx = (size_t)r * U + u;
            const float2 v = lp2[nb + idx];
            const int r1 = (r + 1 == T) ? 0 : r + 1;
            const float2 g = cell_grads(alphas[nb + idx], v.x, v.y, guard.b00, t, u, len.Tn, len.Un, fastemit_lambda,
                                        [&](int col) { return be[(size_t)r1 * U + col]; });
            c.gB = g.x;
            c.gL = g.y;
        }
        c.lab = (u < U - 1) ? labels[(size_t)n * (U - 1) + u] : blank;
        if ((unsigned)c.lab >= (unsigned)V) c.lab = -1;
        // dense-kernel rule (core.cu:382-393), as PaddedRows in overwrite mode
        const bool labvalid = (t < xn[n]) && (u < yn[n]);
        if (!labvalid) c.lab = -1;
        else if (c.lab == blank) c.gB = c.gL;
        if (labvalid && c.lab == blank) c.lab = -1;
        return c;
    }
};
struct CompactRows {   // (STU,V) output, row-major pairs + loc (core_compact.cu:456-484)
    const float2* g2; const int64_t* loc; const int* cum_lens; const float* grad_cost;
    int N, V, blank;
    __device__ __forceinline__ ExpandCell operator()(unsigned row) const {
        // utterance of this row: first n with cum_lens[n] > row (inclusive prefix sums)
        int lo = 0, hi = N - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((unsigned)cum_lens[mid] > row) hi = mid; else lo = mid + 1;
        }
        const float sc = grad_cost[lo];
        const float2 g = g2[row];
        ExpandCell c;
        c.gB = g.x * sc;
        c.gL = g.y * sc;
        c.lab = (int)loc[row];
        if (c.lab == blank || (unsigned)c.lab >= (unsigned)V) c.lab = -1;        // the reference writes the label slot only if loc != blank
        return c;
    }
};

// ---- V <= 1024: R whole rows per workgroup through LDS ----
constexpr int EX_THREADS = 256;
constexpr int EX_FLOATS = 3200;

template <typename Rows>
__global__ void __launch_bounds__(EX_THREADS)
k_expand_small(const Rows rows, float* __restrict__ dense, unsigned cells, int R, int V, int blank) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int tid = threadIdx.x;
    const unsigned blk = stream_block<XCD_EXPAND_SMALL>();
    if ((unsigned long long)blk * (unsigned)R >= cells) return;
    const unsigned cell0 = blk * (unsigned)R;
    const int nrows = (int)min((unsigned)R, cells - cell0);
    const int nel = nrows * V;
    const int nvec = nel >> 2;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < nvec; i += EX_THREADS) reinterpret_cast<float4*>(tile)[i] = z;
    for (int e = (nvec << 2) + tid; e < nel; e += EX_THREADS) tile[e] = 0.f;
    __syncthreads();
    for (int r = tid; r < nrows; r += EX_THREADS) {
        const ExpandCell c = rows(cell0 + r);
        float* row = tile + r * V;
        row[blank] = c.gB;
        if (c.lab >= 0) row[c.lab] += c.gL;     // scatter-add semantics when label == blank (sum mode)
    }
    __syncthreads();
    float* dst = dense + (size_t)cell0 * V;
    for (int i = tid; i < nvec; i += EX_THREADS)
        ex_store4(reinterpret_cast<float4*>(dst) + i, reinterpret_cast<const float4*>(tile)[i]);
    for (int e = (nvec << 2) + tid; e < nel; e += EX_THREADS) dst[e] = tile[e];
}

// ---- larger V: one row per workgroup iteration, zeros from registers ----
template <int VEC, typename Rows>
__global__ void __launch_bounds__(256)
k_expand_large(const Rows rows, float* __restrict__ dense, unsigned cells, int V, int blank) {
    for (unsigned cell = blockIdx.x; cell < cells; cell += gridDim.x) {
        const ExpandCell c = rows(cell);
        float* dst = dense + (size_t)cell * V;
        for (int v0 = threadIdx.x * VEC; v0 < V; v0 += 256 * VEC) {
            float o[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int v = v0 + j;
                o[j] = ((v == blank) ? c.gB : 0.0f) + ((v == c.lab) ? c.gL : 0.0f);
            }
            if constexpr (VEC == 4) ex_store4(reinterpret_cast<float4*>(dst + v0), make_float4(o[0], o[1], o[2], o[3]));
            else dst[v0] = o[0];
        }
    }
}

template <typename Rows>
static hipError_t launch_rows(hipStream_t stream, const Rows& rows, float* dense, unsigned cells, int V, int blank) {
    const bool aligned = reinterpret_cast<uintptr_t>(dense) % 16 == 0;
    if (aligned && V <= 1024) {
        int R = (EX_FLOATS / V) / 4 * 4;          // R % 4 == 0 keeps every tile start 16-byte aligned
        if (R < 4) R = 4;
        const size_t lds = (size_t)R * V * sizeof(float);
        k_expand_small<Rows><<<stream_grid<XCD_EXPAND_SMALL>((cells + R - 1) / R), EX_THREADS, lds, stream>>>(rows, dense, cells, R,
                                                                                                       V, blank);
    } else {
        const unsigned grid = cells < (1u << 22) ? cells : (1u << 22);
        if (aligned && V % 4 == 0)
            k_expand_large<4, Rows><<<grid, 256, 0, stream>>>(rows, dense, cells, V, blank);
        else
            k_expand_large<1, Rows><<<grid, 256, 0, stream>>>(rows, dense, cells, V, blank);
    }
    return hipGetLastError();
}

hipError_t launch_expand(hipStream_t stream, const float* g2_skewed, const int* labels, const int* xn,
                         const int* yn, const float* scale, float* dense, int N, int T, int U, int V,
                         int blank, int overwrite_mode) {
    const size_t cells64 = (size_t)N * T * U;
    if (cells64 == 0 || V == 0) return hipSuccess;
    const PaddedRows rows{reinterpret_cast<const float2*>(g2_skewed), labels, xn, yn, scale, T, U, V, blank,
                          overwrite_mode};
    return launch_rows(stream, rows, dense, (unsigned)cells64, V, blank);
}

hipError_t launch_grads_dense(hipStream_t stream, const GradArgs& a, float* dense, int N) {
    const size_t cells64 = (size_t)N * a.T * a.U;
    if (cells64 == 0 || a.V == 0) return hipSuccess;
    if (is_compact(a)) return hipErrorNotSupported;
    const GradRows rows{reinterpret_cast<const float2*>(a.lp), a.alphas, a.betas, a.ll, a.labels, a.xn, a.yn, a.costs,
                        a.mismatch, a.T, a.U, a.V, a.blank, a.fastemit_lambda, mismatch_words_of(stream)};
    return launch_rows(stream, rows, dense, (unsigned)cells64, a.V, a.blank);
}

hipError_t launch_expand_split(hipStream_t stream, const float* ga_skewed, const float* gb_skewed, const int* labels,
                               const int* xn, const int* yn, float* dense, int N, int T, int U, int V, int blank) {
    const size_t cells64 = (size_t)N * T * U;
    if (cells64 == 0 || V == 0) return hipSuccess;
    const SplitRows rows{ga_skewed, gb_skewed, labels, xn, yn, T, U, V, blank};
    return launch_rows(stream, rows, dense, (unsigned)cells64, V, blank);
}

// (STU,V) gradient rows of the compact layout; replaces the first version in prologue.hip
hipError_t launch_scatter_compact(hipStream_t stream, const float* grad_cost, const float* grads2,
                                  const int64_t* loc, const int* cum_lens, float* out, int64_t STU, int N,
                                  int V, int blank) {
    if (STU <= 0 || V <= 0) return hipSuccess;
    if (STU >= ((int64_t)1 << 32)) return hipErrorInvalidValue;
    const CompactRows rows{reinterpret_cast<const float2*>(grads2), loc, cum_lens, grad_cost, N, V, blank};
    return launch_rows(stream, rows, out, (unsigned)STU, V, blank);
}

}  // namespace rnnt
