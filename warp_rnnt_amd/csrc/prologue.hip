// Streaming prologue kernels: log-softmax over the vocabulary axis, the gather
// of the (blank, label) log-prob pair per lattice cell, and the fusion of the
// two.  These are the HBM-bound part of the op (the dense (N,T,U,V) tensor is
// 25x..5000x larger than everything the lattice kernels touch).
//
// Reference counterparts:
//   log-softmax  : caller side, pytorch_binding/benchmark.py:65,70 (F.log_softmax)
//   gather       : warp_rnnt/__init__.py:118-128 (torch.full int64 index + slice-assign +
//                  torch.gather; 16 B of index per cell) and core_compact.cu:403-436
//   The fused form reads the logits once and never materialises log-probs.
//
// Output of the gather kernels is the diagonal-major float2 workspace
// (common.h) that the lattice sweep reads with coalesced row loads.
#include "common.h"
#include "kernels.h"

namespace rnnt {

// ---------------------------------------------------------------------------
// wave / block reductions
// ---------------------------------------------------------------------------
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// Where the (blank,label) pair of flat cell index `cell` (row-major over N,T,U) lives in the
// diagonal-major workspace, and which label the cell uses.
struct CellMap {
    size_t sk;   // float2 index into the workspace
    int label;   // vocabulary index of the label channel (blank for the last column)
};
__device__ __forceinline__ CellMap map_cell(size_t cell, const int* __restrict__ labels, int T, int U,
                                            int blank) {
    const size_t frame = cell / (unsigned)U;          // n*T + t
    const int u = (int)(cell - frame * (unsigned)U);
    const size_t n = frame / (unsigned)T;
    const int t = (int)(frame - n * (unsigned)T);
    int r = t + u;
    r = r >= T ? r % T : r;
    CellMap m;
    m.sk = (n * T + r) * (size_t)U + u;
    m.label = (u < U - 1) ? labels[n * (size_t)(U - 1) + u] : blank;
    return m;
}

// ---------------------------------------------------------------------------
// Small vocabularies (V <= 1024): a workgroup stages R whole rows in LDS with
// 16-byte coalesced loads, L lanes cooperate on a row, results leave with
// 16-byte coalesced stores (or as one float2 per row for the fused gather).
// LDS rows are padded to a stride S = L*odd so that the L-lane groups of a
// 32-lane LDS access hit distinct banks.
// ---------------------------------------------------------------------------
constexpr int SM_THREADS = 256;
constexpr int SM_FLOATS = 6144;   // LDS tile budget in floats (24 KiB -> 6 workgroups per CU)

template <int L, bool GATHER>
__global__ void __launch_bounds__(SM_THREADS)
k_lsm_small(const float* x, float* out, const int* __restrict__ labels,
            int64_t rows, int V, int S, int R, int T, int U, int blank) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * R;
    const int nrows = (int)min((int64_t)R, rows - row0);
    const int nel = nrows * V;                      // floats in this chunk
    const float* src = x + row0 * V;   // 16-byte aligned: R % 4 == 0 (out may alias x)

    // ---- stage: coalesced float4 loads, scattered into padded LDS rows ----
    const int nvec = nel >> 2;
    for (int i = tid; i < nvec; i += SM_THREADS) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        const int e = i << 2;
        int r = e / V, c = e - r * V;
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tile[r * S + c] = vv[j];
            if (++c == V) { c = 0; ++r; }
        }
    }
    for (int e = (nvec << 2) + tid; e < nel; e += SM_THREADS) {   // tail (only in the last chunk)
        const int r = e / V, c = e - r * V;
        tile[r * S + c] = src[e];
    }
    __syncthreads();

    // ---- per-row max / sum(exp) / normalise: L lanes per row ----
    constexpr int RPP = SM_THREADS / L;             // rows per pass
    const int h = tid % L, rr = tid / L;
    for (int r = rr; r < nrows; r += RPP) {
        float* row = tile + r * S;
        float mx = -__builtin_inff();
        for (int c = h; c < V; c += L) mx = fmaxf(mx, row[c]);
        mx = group_max<L>(mx);
        float s = 0.0f;
        for (int c = h; c < V; c += L) s += expf(row[c] - mx);
        s = group_sum<L>(s);
        const float ls = logf(s);
        if constexpr (GATHER) {
            if (h == 0) {
                const CellMap m = map_cell((size_t)(row0 + r), labels, T, U, blank);
                reinterpret_cast<float2*>(out)[m.sk] =
                    make_float2((row[blank] - mx) - ls, (row[m.label] - mx) - ls);
            }
        } else {
            for (int c = h; c < V; c += L) row[c] = (row[c] - mx) - ls;
        }
    }
    if constexpr (!GATHER) {
        __syncthreads();
        float* dst = out + row0 * V;
        for (int i = tid; i < nvec; i += SM_THREADS) {
            const int e = i << 2;
            int r = e / V, c = e - r * V;
            float vv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                vv[j] = tile[r * S + c];
                if (++c == V) { c = 0; ++r; }
            }
            reinterpret_cast<float4*>(dst)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        }
        for (int e = (nvec << 2) + tid; e < nel; e += SM_THREADS) {
            const int r = e / V, c = e - r * V;
            dst[e] = tile[r * S + c];
        }
    }
}

// ---------------------------------------------------------------------------
// Large vocabularies (1024 < V <= 16384, V % 4 == 0): one workgroup per row,
// the row lives in registers (up to 16 float4 per lane), one HBM read and one
// HBM write per element.
// ---------------------------------------------------------------------------
constexpr int LG_THREADS = 256;
constexpr int LG_MAXVEC = 16;

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
    v = is_max ? group_max<WAVE>(v) : group_sum<WAVE>(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();                      // protect `red` from the previous use
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int i = 1; i < LG_THREADS / WAVE; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

template <bool GATHER>
__global__ void __launch_bounds__(LG_THREADS)
k_lsm_large(const float* x, float* out, const int* __restrict__ labels,
            int64_t rows, int V, int T, int U, int blank) {
    __shared__ float red[LG_THREADS / WAVE];
    for (size_t row = blockIdx.x; row < (size_t)rows; row += gridDim.x) {
    const float4* src = reinterpret_cast<const float4*>(x + row * V);
    const int nvec = V >> 2;
    float4 v[LG_MAXVEC];
    float mx = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < LG_MAXVEC; ++i) {
        const int j = threadIdx.x + i * LG_THREADS;
        if (j < nvec) {
            v[i] = src[j];
            mx = fmaxf(fmaxf(mx, fmaxf(v[i].x, v[i].y)), fmaxf(v[i].z, v[i].w));
        }
    }
    mx = block_reduce(mx, true, red);
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < LG_MAXVEC; ++i) {
        const int j = threadIdx.x + i * LG_THREADS;
        if (j < nvec) s += (expf(v[i].x - mx) + expf(v[i].y - mx)) + (expf(v[i].z - mx) + expf(v[i].w - mx));
    }
    s = block_reduce(s, false, red);
    const float ls = logf(s);
    if constexpr (GATHER) {
        if (threadIdx.x == 0) {
            const CellMap m = map_cell(row, labels, T, U, blank);
            const float* xr = x + row * V;
            reinterpret_cast<float2*>(out)[m.sk] =
                make_float2((xr[blank] - mx) - ls, (xr[m.label] - mx) - ls);
        }
    } else {
        float4* dst = reinterpret_cast<float4*>(out + row * V);
#pragma unroll
        for (int i = 0; i < LG_MAXVEC; ++i) {
            const int j = threadIdx.x + i * LG_THREADS;
            if (j < nvec)
                dst[j] = make_float4((v[i].x - mx) - ls, (v[i].y - mx) - ls, (v[i].z - mx) - ls,
                                     (v[i].w - mx) - ls);
        }
    }
    }
}

// ---------------------------------------------------------------------------
// Generic fallback (any V, any alignment): one wave per row, three passes.
// ---------------------------------------------------------------------------
template <bool GATHER>
__global__ void __launch_bounds__(256)
k_lsm_generic(const float* x, float* out, const int* __restrict__ labels,
              int64_t rows, int V, int T, int U, int blank) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * V;
    float mx = -__builtin_inff();
    for (int c = lane; c < V; c += WAVE) mx = fmaxf(mx, xr[c]);
    mx = group_max<WAVE>(mx);
    float s = 0.0f;
    for (int c = lane; c < V; c += WAVE) s += expf(xr[c] - mx);
    s = group_sum<WAVE>(s);
    const float ls = logf(s);
    if constexpr (GATHER) {
        if (lane == 0) {
            const CellMap m = map_cell((size_t)row, labels, T, U, blank);
            reinterpret_cast<float2*>(out)[m.sk] =
                make_float2((xr[blank] - mx) - ls, (xr[m.label] - mx) - ls);
        }
    } else {
        float* o = out + row * V;
        for (int c = lane; c < V; c += WAVE) o[c] = (xr[c] - mx) - ls;
    }
}

template <bool GATHER>
static hipError_t dispatch_lsm(hipStream_t stream, const float* x, float* out, const int* labels,
                               int64_t rows, int V, int T, int U, int blank) {
    if (rows <= 0) return hipSuccess;
    const bool aligned = (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                         (GATHER || reinterpret_cast<uintptr_t>(out) % 16 == 0);
    if (aligned && V <= 1024) {
        int L = 1;
        while (L < 64 && L * 16 < V) L <<= 1;          // ~<=16 elements per lane
        int q = (V + L - 1) / L;
        if (L < 32 && (q & 1) == 0) ++q;               // stride = L*odd: conflict-free row groups
        const int S = (L < 32) ? L * q : V;
        int R = SM_FLOATS / S;
        R = (R / 4) * 4;
        if (R < 4) R = 4;
        const int rpp = SM_THREADS / L;
        if (R > rpp) R = (R / rpp) * rpp;              // whole passes (rpp is a multiple of 4)
        const size_t lds = (size_t)R * S * sizeof(float);
        const unsigned grid = (unsigned)((rows + R - 1) / R);
#define LSM_SMALL(LL)                                                                           \
    case LL:                                                                                    \
        k_lsm_small<LL, GATHER><<<grid, SM_THREADS, lds, stream>>>(x, out, labels, rows, V, S, R, T, \
                                                                    U, blank);                  \
        break;
        switch (L) {
            LSM_SMALL(1) LSM_SMALL(2) LSM_SMALL(4) LSM_SMALL(8) LSM_SMALL(16) LSM_SMALL(32)
            LSM_SMALL(64)
        }
#undef LSM_SMALL
    } else if (aligned && V % 4 == 0 && V <= LG_THREADS * 4 * LG_MAXVEC) {
        k_lsm_large<GATHER><<<(unsigned)(rows < (1 << 22) ? rows : (1 << 22)), LG_THREADS, 0, stream>>>(
            x, out, labels, rows, V, T, U, blank);
    } else {
        k_lsm_generic<GATHER><<<(unsigned)((rows + 3) / 4), 256, 0, stream>>>(x, out, labels, rows, V,
                                                                               T, U, blank);
    }
    return hipGetLastError();
}

hipError_t launch_log_softmax(hipStream_t stream, const float* x, float* out, int64_t rows, int V) {
    return dispatch_lsm<false>(stream, x, out, nullptr, rows, V, 1, 1, 0);
}

hipError_t launch_log_softmax_gather_skewed(hipStream_t stream, const float* logits, const int* labels,
                                            float* ws2, int N, int T, int U, int V, int blank) {
    return dispatch_lsm<true>(stream, logits, ws2, labels, (int64_t)N * T * U, V, T, U, blank);
}

// ---------------------------------------------------------------------------
// gather: dense log-probs -> diagonal-major (blank,label) pairs. One thread per
// cell, lanes along u: the float2 stores of one t-row land on U different
// diagonals (8-byte scattered stores, merged in L2), the reads are two 4-byte
// accesses per 4V-byte row.
// ---------------------------------------------------------------------------
template <bool SKEWED>
__global__ void __launch_bounds__(256)
k_gather(const float* __restrict__ lp, const int* __restrict__ labels, float2* __restrict__ out2,
         size_t cells, int T, int U, int V, int blank) {
    const size_t cell = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (cell >= cells) return;
    const CellMap m = map_cell(cell, labels, T, U, blank);
    const float* p = lp + cell * (size_t)V;
    out2[SKEWED ? m.sk : cell] = make_float2(p[blank], p[m.label]);
}

hipError_t launch_gather(hipStream_t stream, const float* log_probs, const int* labels, float* out2,
                         int N, int T, int U, int V, int blank, bool skewed) {
    const size_t cells = (size_t)N * T * U;
    if (cells == 0) return hipSuccess;
    const unsigned grid = (unsigned)((cells + 255) / 256);
    if (skewed)
        k_gather<true><<<grid, 256, 0, stream>>>(log_probs, labels, reinterpret_cast<float2*>(out2),
                                                 cells, T, U, V, blank);
    else
        k_gather<false><<<grid, 256, 0, stream>>>(log_probs, labels, reinterpret_cast<float2*>(out2),
                                                  cells, T, U, V, blank);
    return hipGetLastError();
}

// (N,T,U,2) row-major -> diagonal-major. Same cell walk, 8-byte coalesced reads.
__global__ void __launch_bounds__(256)
k_reskew(const float2* __restrict__ in, float2* __restrict__ ws2, size_t cells, int T, int U) {
    const size_t cell = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (cell >= cells) return;
    const size_t frame = cell / (unsigned)U;
    const int u = (int)(cell - frame * (unsigned)U);
    const size_t n = frame / (unsigned)T;
    const int t = (int)(frame - n * (unsigned)T);
    int r = t + u;
    r = r >= T ? r % T : r;
    ws2[(n * T + r) * (size_t)U + u] = in[cell];
}

hipError_t launch_reskew(hipStream_t stream, const float* lp2_rowmajor, float* ws2, int N, int T, int U) {
    const size_t cells = (size_t)N * T * U;
    if (cells == 0) return hipSuccess;
    k_reskew<<<(unsigned)((cells + 255) / 256), 256, 0, stream>>>(
        reinterpret_cast<const float2*>(lp2_rowmajor), reinterpret_cast<float2*>(ws2), cells, T, U);
    return hipGetLastError();
}

}  // namespace rnnt
