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
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#include "common.h"
#include <type_traits>

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
    int n;       // utterance
};
__device__ __forceinline__ CellMap map_cell(size_t cell, const int* __restrict__ labels, int T, int U,
                                            int V, int blank) {
    // N*T*U < 2^32 is checked by the C ABI, so 32-bit divisions are enough
    const unsigned c32 = (unsigned)cell;
    const unsigned frame = c32 / (unsigned)U;         // n*T + t
    const int u = (int)(c32 - frame * (unsigned)U);
    const unsigned n = frame / (unsigned)T;
    const int t = (int)(frame - n * (unsigned)T);
    int r = t + u;
    r = r >= T ? r % T : r;
    CellMap m;
    m.sk = ((size_t)n * T + r) * (size_t)U + u;
    m.label = (u < U - 1) ? safe_label(labels[(size_t)n * (U - 1) + u], V, blank) : blank;
    m.n = (int)n;
    return m;
}

// ---------------------------------------------------------------------------
// Small vocabularies (V <= 1024): a workgroup stages R whole rows in LDS with
// 16-byte coalesced loads (ds_write_b128, rows kept at their natural stride V so
// the tile is a byte copy of the global chunk), L lanes cooperate on a row, and
// results leave with ds_read_b128 + 16-byte coalesced stores (or as one float2
// per row for the fused gather).  The first version of this kernel was
// VALU-bound, not HBM-bound (rocprofv3: 871 VALU instructions per wave, i.e.
// ~70 per element: libm expf, per-element index division for padded LDS rows,
// per-lane loop control); this one spends ~12.
//   exp(x - max) is evaluated as exp2(x*log2e - max*log2e) on the hardware
//   v_exp_f32 unit, log(sum) as v_log_f32 * ln2 (sum in [1,V]); both are within
//   ~2 ulp, the result is within 4e-6 of torch.log_softmax (tests).
// ---------------------------------------------------------------------------
// Cache policy of the LDS-staged kernel's 16-byte global loads and stores: non-temporal in the fused modes (gather:
// a read-only stream of the logits; backward: logits in, d/d logits out -- fused forward 0.472 -> 0.464 ms, fused
// training step 1.00 -> 0.975 ms at c4, profiles/r03_bwd_nt_ab.txt), plain for the log-softmax itself, where the
// hints measured nothing to worse in rounds 1-2 (HISTORY.md).  -DRNNT_LSM_NT forces them everywhere (A/B builds).
// Written-through stores (sc1 / sc1 nt; round 6, after the dense gather gained from them): nothing at c4 in any mode of this
// kernel, c3's row-per-workgroup kernel 0.658 -> 0.73-0.76 ms, the register kernel 480 -> 520-580 us -- they pay only where
// every store instruction covers whole 128-byte lines (profiles/r06_lsm_store_policy.txt).
typedef float rnnt_f4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 lsm_load4(const float4* p) {
    if constexpr (NT) {
        const rnnt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const rnnt_f4*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return *p;
    }
}
template <bool NT> __device__ __forceinline__ void lsm_store4(float4* p, float4 v) {
    if constexpr (NT) {
        const rnnt_f4 w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<rnnt_f4*>(p));
    } else {
        *p = v;
    }
}
#ifdef RNNT_LSM_NT
#define RNNT_LSM_NT_MODE(MODE) true
#else
#define RNNT_LSM_NT_MODE(MODE) ((MODE) != LSM_NORM)
#endif
#define RNNT_LSM_LOAD(p) lsm_load4<RNNT_LSM_NT_MODE(MODE)>(p)
#define RNNT_LSM_STORE(p, v) lsm_store4<RNNT_LSM_NT_MODE(MODE)>(p, v)

// What the log-softmax kernels emit.
enum LsmMode : int {
    LSM_NORM = 0,    // log-softmax rows
    LSM_GATHER = 1,  // diagonal-major (blank,label) log-prob pairs; log-probs never materialise
    LSM_BWD = 2      // d(loss)/d(logits) rows from the gathered gradients:
                     //   dz[v] = s*( [v==blank]gB + [v==label]gL - softmax(z)[v]*(gB+gL) )
};
struct LsmBwd {
    const float2* g2;    // diagonal-major gathered gradients (RNNT_GRADS_GATHERED_DIAGONAL)
    const float* scale;  // (N,) upstream gradient per utterance, or nullptr
    int xcd;             // row-per-workgroup kernel: 1 = every XCD streams a contiguous eighth of the rows
};

#ifndef RNNT_SM_THREADS
#define RNNT_SM_THREADS 256
#endif
constexpr int SM_THREADS = RNNT_SM_THREADS;
#ifndef RNNT_SM_FLOATS
#define RNNT_SM_FLOATS 3200
#endif
constexpr int SM_FLOATS = RNNT_SM_FLOATS;   // LDS tile budget in floats: one pass of the 256 threads over a 12.5 KiB tile.  (512 threads x 25 KiB: 2 % faster in the isolated probe, slower in bench.py and in the fused gather mode; two passes per tile or 50 KiB tiles are clearly worse.)
constexpr float LOG2E = 1.44269504088896340736f;
constexpr float LN2 = 0.693147180559945309417f;

// WP ("wave private", L <= 16 and one pass per tile): every wave stages, normalises and stores its own
// WAVE/L consecutive rows (a multiple of 4, so its chunk is 16-byte aligned) and the workgroup never
// synchronises -- 32 independent streams per CU instead of 8 workgroups that each wait for their slowest wave.
__device__ __forceinline__ void wave_sync_lds() {
    // LDS operations of one wave retire in order; this only stops the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int L, int MODE, bool WP>
__global__ void __launch_bounds__(SM_THREADS)
k_lsm_small(const float* x, float* out, const int* __restrict__ labels,
            int64_t rows, int V, int R, int q, int T, int U, int blank, LsmBwd bw) {
    constexpr bool GATHER = MODE == LSM_GATHER;
    extern __shared__ __attribute__((aligned(16))) float tile[];
    float2* stat = reinterpret_cast<float2*>(tile + (size_t)R * V);   // GATHER: (max, log-sum) per row
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)stream_block<XCD_LSM_SMALL>() * R;
    if (row0 >= rows) return;
    const int nrows = (int)min((int64_t)R, rows - row0);
    const int nel = nrows * V;                      // floats in this chunk
    const float* src = x + row0 * V;                // 16-byte aligned: R % 4 == 0 (out may alias x)
    // wave-private view of the same tile: rows [wr0, wr0 + wn) of the chunk
    constexpr int RW = (WAVE / L) > 0 ? (WAVE / L) : 1;
    const int lane = tid & (WAVE - 1);
    const int wr0 = (tid >> 6) * RW;
    const int wn = min(max(nrows - wr0, 0), RW);
    const int wel = wn * V, wvec = wel >> 2;
    float* wtile = tile + wr0 * V;

    // ---- stage: the tile is a plain copy of the chunk ----
    const int nvec = nel >> 2;
    // fused backward: the gradient pair (and scale) of the row this thread works on first, requested before the tile
    [[maybe_unused]] CellMap pm = {0, 0, 0};
    [[maybe_unused]] float2 pg = make_float2(0.0f, 0.0f);
    [[maybe_unused]] float psc = 1.0f;
    if constexpr (MODE == LSM_BWD) {
        pm = map_cell((size_t)(row0 + min(tid / L, nrows - 1)), labels, T, U, V, blank);
        pg = bw.g2[pm.sk];
        psc = bw.scale ? bw.scale[pm.n] : 1.0f;
    }
    // LOADS FIRST (round 5).  Written as `for (i ...) tile[i] = load(src + i)` the compiler emits load, s_waitcnt vmcnt(0),
    // ds_write per iteration: a wave had ONE 16-byte load per lane in flight at a time and paid the memory latency three
    // to four times per tile -- 1 KB per wave in flight, 32 KB per CU, which at ~1.5 us of loaded latency is the 5.2 TB/s
    // the fused gather ran at (read-only streams reach 7.0, tools/ubench/copy_rate.hip).  A tile is at most four passes
    // of the threads that stage it (SM_FLOATS, and V <= 16 L for the wave-private form): all of a lane's loads are
    // issued before the first of them is written to LDS -- unconditionally, at an index clamped into the tile, and so are
    // the LDS writes (lanes past the end rewrite the tile's last 16 bytes with the bytes that are there).  A predicate per
    // load comes out as a branch per load with a conservative wait at every join; predicates on the writes alone and the
    // compiler sinks the loads into them.
    constexpr int STAGE_UN = WP ? 4 : (SM_FLOATS / 4 + SM_THREADS - 1) / SM_THREADS;   // (3200 floats, 256 threads: 4)
    if constexpr (WP) {
        const float4* wsrc4 = reinterpret_cast<const float4*>(src + (size_t)wr0 * V);
        for (int base = lane; base < wvec; base += STAGE_UN * WAVE) {
            float4 sv[STAGE_UN];
#pragma unroll
            for (int k = 0; k < STAGE_UN; ++k)
                sv[k] = RNNT_LSM_LOAD(wsrc4 + min(base + k * WAVE, wvec - 1));
#pragma unroll
            for (int k = 0; k < STAGE_UN; ++k)
                reinterpret_cast<float4*>(wtile)[min(base + k * WAVE, wvec - 1)] = sv[k];
        }
        const float* wsrc = src + (size_t)wr0 * V;
        for (int e = (wvec << 2) + lane; e < wel; e += WAVE) wtile[e] = wsrc[e];
        wave_sync_lds();
    } else {
        const float4* src4 = reinterpret_cast<const float4*>(src);
        for (int base = tid; base < nvec; base += STAGE_UN * SM_THREADS) {
            float4 sv[STAGE_UN];
#pragma unroll
            for (int k = 0; k < STAGE_UN; ++k)
                sv[k] = RNNT_LSM_LOAD(src4 + min(base + k * SM_THREADS, nvec - 1));
#pragma unroll
            for (int k = 0; k < STAGE_UN; ++k)
                reinterpret_cast<float4*>(tile)[min(base + k * SM_THREADS, nvec - 1)] = sv[k];
        }
        for (int e = (nvec << 2) + tid; e < nel; e += SM_THREADS) tile[e] = src[e];   // last chunk only
        __syncthreads();
    }

    // ---- per-row max / sum(exp) / normalise: L lanes per row, lane h owns columns h, h+L, ... ----
    constexpr int RPP = SM_THREADS / L;             // rows per pass
    const int h = tid % L, rr = tid / L;
    const int ctail = h + (q - 1) * L;              // this lane's last column, may be >= V
    const bool tail_ok = ctail < V;
    // One row: the lane's q values are read ONCE into registers by a straight-line sequence (all LDS reads in flight
    // together), reduced, and -- in the modes that rewrite the row -- written back from the registers.  QC = q as a
    // compile-time constant (9 ... 16: what the launcher's choice of L gives for V > 16); the run-time loops of the first
    // version waited for every LDS read of the max pass on its own (~9 instructions and one LDS latency per element) and
    // read every element a second time for the sum.
    auto one_row = [&](auto QC, const int r) {
        constexpr int Q = decltype(QC)::value;
        float* row = tile + r * V;
        float v[Q];
#pragma unroll
        for (int i = 0; i < Q - 1; ++i) v[i] = row[h + i * L];
        v[Q - 1] = tail_ok ? row[ctail] : -__builtin_inff();
        float mx = v[0];
#pragma unroll
        for (int i = 1; i < Q; ++i) mx = fmaxf(mx, v[i]);
        mx = group_max<L>(mx);
        const float mb = -mx * LOG2E;
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < Q; ++i) s += __builtin_amdgcn_exp2f(__builtin_fmaf(v[i], LOG2E, mb));   // (exp2(-inf) = 0)
        s = group_sum<L>(s);
        const float ls = __builtin_amdgcn_logf(s) * LN2;
        if constexpr (GATHER) {
            if (h == 0) stat[r] = make_float2(mx, ls);
        } else if constexpr (MODE == LSM_BWD) {
            const bool first = r == rr;                // (the row whose pair was requested up front)
            const CellMap m = first ? pm : map_cell((size_t)(row0 + r), labels, T, U, V, blank);
            const float sc = first ? psc : (bw.scale ? bw.scale[m.n] : 1.0f);
            const float2 g = first ? pg : bw.g2[m.sk];
            const float gB = g.x * sc, gL = g.y * sc, gs = gB + gL;
            const float mb2 = -(mx + ls) * LOG2E;
#pragma unroll
            for (int i = 0; i < Q - 1; ++i)
                row[h + i * L] = -__builtin_amdgcn_exp2f(__builtin_fmaf(v[i], LOG2E, mb2)) * gs;
            if (tail_ok) row[ctail] = -__builtin_amdgcn_exp2f(__builtin_fmaf(v[Q - 1], LOG2E, mb2)) * gs;
            // the L lanes of a row sit in one wave and LDS operations of a wave retire in order
            if (h == 0) { row[blank] += gB; row[m.label] += gL; }
        } else {
#pragma unroll
            for (int i = 0; i < Q - 1; ++i) row[h + i * L] = (v[i] - mx) - ls;
            if (tail_ok) row[ctail] = (v[Q - 1] - mx) - ls;
        }
    };
    auto all_rows = [&](auto QC) {
        for (int r = rr; r < nrows; r += RPP) one_row(QC, r);
    };
    switch (q) {
#define LSM_Q(QQ) case QQ: all_rows(std::integral_constant<int, QQ>{}); break;
        LSM_Q(9) LSM_Q(10) LSM_Q(11) LSM_Q(12) LSM_Q(13) LSM_Q(14) LSM_Q(15) LSM_Q(16)
#undef LSM_Q
        default:      // q <= 8 (V <= 16, or rows shorter than the lane cover): run-time loops
    for (int r = rr; r < nrows; r += RPP) {
        float* row = tile + r * V;
        float mx = -__builtin_inff();
        for (int i = 0, c = h; i < q - 1; ++i, c += L) mx = fmaxf(mx, row[c]);
        if (tail_ok) mx = fmaxf(mx, row[ctail]);
        mx = group_max<L>(mx);
        const float mb = -mx * LOG2E;
        float s = 0.0f;
        for (int i = 0, c = h; i < q - 1; ++i, c += L) s += __builtin_amdgcn_exp2f(__builtin_fmaf(row[c], LOG2E, mb));
        if (tail_ok) s += __builtin_amdgcn_exp2f(__builtin_fmaf(row[ctail], LOG2E, mb));
        s = group_sum<L>(s);
        const float ls = __builtin_amdgcn_logf(s) * LN2;
        if constexpr (GATHER) {
            if (h == 0) stat[r] = make_float2(mx, ls);
        } else if constexpr (MODE == LSM_BWD) {
            const CellMap m = map_cell((size_t)(row0 + r), labels, T, U, V, blank);
            const float sc = bw.scale ? bw.scale[m.n] : 1.0f;
            const float2 g = bw.g2[m.sk];
            const float gB = g.x * sc, gL = g.y * sc, gs = gB + gL;
            const float mb2 = -(mx + ls) * LOG2E;
            for (int i = 0, c = h; i < q - 1; ++i, c += L)
                row[c] = -__builtin_amdgcn_exp2f(__builtin_fmaf(row[c], LOG2E, mb2)) * gs;
            if (tail_ok) row[ctail] = -__builtin_amdgcn_exp2f(__builtin_fmaf(row[ctail], LOG2E, mb2)) * gs;
            // the L lanes of a row sit in one wave and LDS operations of a wave retire in order
            if (h == 0) { row[blank] += gB; row[m.label] += gL; }
        } else {
            for (int i = 0, c = h; i < q - 1; ++i, c += L) row[c] = (row[c] - mx) - ls;
            if (tail_ok) row[ctail] = (row[ctail] - mx) - ls;
        }
    }
    }
    if constexpr (GATHER && WP) {
        wave_sync_lds();
        for (int r = wr0 + lane; r < wr0 + wn; r += WAVE) {
            const CellMap m = map_cell((size_t)(row0 + r), labels, T, U, V, blank);
            const float2 st = stat[r];
            const float* row = tile + r * V;
            reinterpret_cast<float2*>(out)[m.sk] =
                make_float2((row[blank] - st.x) - st.y, (row[m.label] - st.x) - st.y);
        }
    } else if constexpr (GATHER) {
        // one lane per row with all lanes busy (the per-row index arithmetic costs ~60 instructions;
        // doing it inside the L-lane row loop ran it with a quarter of the lanes)
        __syncthreads();
        for (int r = tid; r < nrows; r += SM_THREADS) {
            const CellMap m = map_cell((size_t)(row0 + r), labels, T, U, V, blank);
            const float2 st = stat[r];
            const float* row = tile + r * V;
            const float2 pr = make_float2((row[blank] - st.x) - st.y, (row[m.label] - st.x) - st.y);
#ifdef RNNT_PROBE_HOT_PAIRS     // timing probe (wrong results): the pairs into a 64 KB region that stays in L2 -- what the
                                // kernel costs without its DRAM writes (the gather's stores cost 40-55 us in any shape)
            reinterpret_cast<float2*>(out)[m.sk & 8191] = pr;
#elif defined(RNNT_PROBE_LINEAR_PAIRS)   // ... and with the pairs in row-major order (coalesced 512-byte runs; wrong layout)
            reinterpret_cast<float2*>(out)[row0 + r] = pr;
#else
            reinterpret_cast<float2*>(out)[m.sk] = pr;      // (written through, sc1: +38 us at c4 -- scattered 8-byte stores need L2 to merge them)
#endif
        }
    } else if constexpr (WP) {
        wave_sync_lds();
        float* wdst = out + (row0 + wr0) * V;
        for (int i = lane; i < wvec; i += WAVE)
            RNNT_LSM_STORE(reinterpret_cast<float4*>(wdst) + i, reinterpret_cast<const float4*>(wtile)[i]);
        for (int e = (wvec << 2) + lane; e < wel; e += WAVE) wdst[e] = wtile[e];
    } else {
        __syncthreads();
        float* dst = out + row0 * V;
        for (int i = tid; i < nvec; i += SM_THREADS)
            RNNT_LSM_STORE(reinterpret_cast<float4*>(dst) + i, reinterpret_cast<const float4*>(tile)[i]);
        for (int e = (nvec << 2) + tid; e < nel; e += SM_THREADS) dst[e] = tile[e];
    }
}

// ---------------------------------------------------------------------------
// Large vocabularies (1024 < V <= 16384, V % 4 == 0): one workgroup per row,
// the row lives in registers (up to 16 float4 per lane), one HBM read and one
// HBM write per element.
// ---------------------------------------------------------------------------
// Shape of the row-per-workgroup kernel: THREADS x NV float4 must cover a row.  The registers that hold
// the row set the residency (NV=16 x 256 threads: 84 VGPRs, 5 waves/SIMD; NV=8: 8 waves/SIMD).  The launcher
// picks 1.25-2.5 float4 per thread for the plain log-softmax and the smallest cover for the read-mostly fused modes
// (dispatch_lsm).
constexpr int LG_MAXV = 16384;
// cache policy of the plain (LSM_NORM) row-per-workgroup stream: bit 0 = non-temporal loads, bit 1 = non-temporal stores.
// Both (round 3; round 1 had tried them on the LDS-staged small-V kernel only, where they do nothing): c5 (V=10000, in
// place, 288 GB of traffic) 57.4 -> 51.3 ms per step, c3 (V=5000) 0.708 -> 0.695 ms; loads alone are WORSE (c3 0.733),
// stores alone neutral (profiles/r03_lg_nt_ab.txt)
#ifndef RNNT_LG_NT
#define RNNT_LG_NT 3
#endif
#ifndef RNNT_LG_NT_FUSED      // the same policy in the fused gather / backward modes: c3 fused forward 0.336 -> 0.325 ms,
#define RNNT_LG_NT_FUSED 1    // fused training step 1.051 -> 1.018 ms (profiles/r03_lg_fused_nt_ab.txt)
#endif
typedef float lg_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 rnnt_nt_load4(const float4* p) {
    const lg_f4 v = __builtin_nontemporal_load(reinterpret_cast<const lg_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void rnnt_nt_store4(float4* p, float4 v) {
    const lg_f4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<lg_f4*>(p));
}

template <int THREADS>
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
    v = is_max ? group_max<WAVE>(v) : group_sum<WAVE>(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();                      // protect `red` from the previous use
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int i = 1; i < THREADS / WAVE; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

template <int MODE, int LG_THREADS, int LG_MAXVEC>
__global__ void __launch_bounds__(LG_THREADS)
k_lsm_large(const float* x, float* out, const int* __restrict__ labels,
            int64_t rows, int V, int T, int U, int blank, LsmBwd bw) {
    constexpr bool GATHER = MODE == LSM_GATHER;
    __shared__ float red[LG_THREADS / WAVE];
    // bw.xcd: every XCD (workgroups go to them by blockIdx mod 8; the grid is a multiple of 8) streams a contiguous
    // eighth of the rows instead of every eighth row -- see dispatch_lsm
    const size_t per_xcd = ((size_t)rows + 7) / 8;
    const size_t items = bw.xcd ? per_xcd * 8 : (size_t)rows;
    for (size_t it = blockIdx.x; it < items; it += gridDim.x) {
    const size_t row = bw.xcd ? (it & 7) * per_xcd + (it >> 3) : it;
    if (row >= (size_t)rows) continue;
    const float4* src = reinterpret_cast<const float4*>(x + row * V);
    const int nvec = V >> 2;
    // How a lane's LG_MAXVEC loads are issued (round 5).  As first written -- load and running maximum together under
    // `if (j < nvec)` -- every load sits in a branch of its own with an s_waitcnt vmcnt(0) behind it: LG_MAXVEC memory
    // round trips per row, one after the other.  Measured against two loads-first forms (tools/ab_kernels.py, three
    // interleaved rounds, profiles/r05_loads_first_ab.txt):
    //   * the read-mostly FUSED modes gain 5-7 % from all loads issued unconditionally at an index clamped into the row,
    //     what lies beyond the row replaced by -inf afterwards (c3: fused forward 317 -> 300 us, fused backward 739 -> 689);
    //   * the plain log-softmax -- a read and a write stream at the rate of a copy -- does not: V = 5000 629 -> 647 us,
    //     4096 596 -> 606, 2048 590 -> 594, nothing at 1000, 3000, 8192; only the three-pass covers of 768 threads and more
    //     gain (c5's V = 10000: 693 -> 674 clamped, -> 665 with the loads alone under their predicates and the maxima
    //     behind them), so those take the predicated form and everything else stays as it was.
    constexpr bool CLAMPED = MODE != LSM_NORM;
    constexpr bool PREDICATED = MODE == LSM_NORM && LG_THREADS >= 768;
    constexpr bool NT_LOADS = (((MODE == LSM_NORM ? 1 : RNNT_LG_NT_FUSED) * RNNT_LG_NT) & 1) != 0;
    float4 v[LG_MAXVEC];
    float mx = -__builtin_inff();
    if constexpr (CLAMPED || PREDICATED) {
#pragma unroll
        for (int i = 0; i < LG_MAXVEC; ++i) {
            const int j = (int)threadIdx.x + i * LG_THREADS;
            if constexpr (CLAMPED) {
                v[i] = NT_LOADS ? rnnt_nt_load4(src + min(j, nvec - 1)) : src[min(j, nvec - 1)];
            } else {
                if (j < nvec) v[i] = NT_LOADS ? rnnt_nt_load4(src + j) : src[j];
            }
        }
    }
    // what the fused modes need besides the row, requested behind it instead of after the reductions
    [[maybe_unused]] CellMap m = {0, 0, 0};
    [[maybe_unused]] float2 side = make_float2(0.0f, 0.0f);      // GATHER: the row's (blank, label) logits; BWD: its gradient pair
    [[maybe_unused]] float sc = 1.0f;
    if constexpr (MODE != LSM_NORM) {
        m = map_cell(row, labels, T, U, V, blank);
        if constexpr (GATHER) {
            const float* xr = x + row * V;
            side = make_float2(xr[blank], xr[m.label]);
        } else {
            side = bw.g2[m.sk];
            sc = bw.scale ? bw.scale[m.n] : 1.0f;
        }
    }
#pragma unroll
    for (int i = 0; i < LG_MAXVEC; ++i) {
        const int j = (int)threadIdx.x + i * LG_THREADS;
        if constexpr (CLAMPED || PREDICATED) {
            if (j >= nvec) {
                const float ninf = -__builtin_inff();
                v[i] = make_float4(ninf, ninf, ninf, ninf);
            }
            mx = fmaxf(fmaxf(mx, fmaxf(v[i].x, v[i].y)), fmaxf(v[i].z, v[i].w));
        } else {
            if (j < nvec) {
                v[i] = NT_LOADS ? rnnt_nt_load4(src + j) : src[j];
                mx = fmaxf(fmaxf(mx, fmaxf(v[i].x, v[i].y)), fmaxf(v[i].z, v[i].w));
            }
        }
    }
    mx = block_reduce<LG_THREADS>(mx, true, red);
    const float mb = -mx * LOG2E;
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < LG_MAXVEC; ++i) {
        const int j = threadIdx.x + i * LG_THREADS;
        if (j < nvec)
            s += (__builtin_amdgcn_exp2f(__builtin_fmaf(v[i].x, LOG2E, mb)) + __builtin_amdgcn_exp2f(__builtin_fmaf(v[i].y, LOG2E, mb))) +
                 (__builtin_amdgcn_exp2f(__builtin_fmaf(v[i].z, LOG2E, mb)) + __builtin_amdgcn_exp2f(__builtin_fmaf(v[i].w, LOG2E, mb)));
    }
    s = block_reduce<LG_THREADS>(s, false, red);
    const float ls = logf(s);
    if constexpr (GATHER) {
        if (threadIdx.x == 0)
            reinterpret_cast<float2*>(out)[m.sk] = make_float2((side.x - mx) - ls, (side.y - mx) - ls);
    } else if constexpr (MODE == LSM_BWD) {
        const float2 g = side;
        const float gB = g.x * sc, gL = g.y * sc, gs = gB + gL;
        const float mb2 = -(mx + ls) * LOG2E;
        float4* dst = reinterpret_cast<float4*>(out + row * V);
#pragma unroll
        for (int i = 0; i < LG_MAXVEC; ++i) {
            const int j = threadIdx.x + i * LG_THREADS;
            if (j < nvec) {
                float o[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int e = 4 * j + cc;
                    float d = -__builtin_amdgcn_exp2f(__builtin_fmaf(o[cc], LOG2E, mb2)) * gs;
                    d += (e == blank) ? gB : 0.0f;
                    d += (e == m.label) ? gL : 0.0f;
                    o[cc] = d;
                }
                if ((RNNT_LG_NT_FUSED * RNNT_LG_NT) & 2) rnnt_nt_store4(dst + j, make_float4(o[0], o[1], o[2], o[3]));
                else dst[j] = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    } else {
        float4* dst = reinterpret_cast<float4*>(out + row * V);
#pragma unroll
        for (int i = 0; i < LG_MAXVEC; ++i) {
            const int j = threadIdx.x + i * LG_THREADS;
            if (j < nvec) {
                const float4 r = make_float4((v[i].x - mx) - ls, (v[i].y - mx) - ls, (v[i].z - mx) - ls,
                                             (v[i].w - mx) - ls);
                if (RNNT_LG_NT & 2) rnnt_nt_store4(dst + j, r); else dst[j] = r;
            }
        }
    }
    }
}

// ---------------------------------------------------------------------------
// Generic fallback (any V, any alignment): one wave per row, three passes.
// ---------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256)
k_lsm_generic(const float* x, float* out, const int* __restrict__ labels,
              int64_t rows, int V, int T, int U, int blank, LsmBwd bw) {
    constexpr bool GATHER = MODE == LSM_GATHER;
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
            const CellMap m = map_cell((size_t)row, labels, T, U, V, blank);
            reinterpret_cast<float2*>(out)[m.sk] =
                make_float2((xr[blank] - mx) - ls, (xr[m.label] - mx) - ls);
        }
    } else if constexpr (MODE == LSM_BWD) {
        const CellMap m = map_cell((size_t)row, labels, T, U, V, blank);
        const float sc = bw.scale ? bw.scale[m.n] : 1.0f;
        const float2 g = bw.g2[m.sk];
        const float gB = g.x * sc, gL = g.y * sc, gs = gB + gL;
        float* o = out + row * V;
        for (int c = lane; c < V; c += WAVE) {
            float d = -expf((xr[c] - mx) - ls) * gs;
            d += (c == blank) ? gB : 0.0f;
            d += (c == m.label) ? gL : 0.0f;
            o[c] = d;
        }
    } else {
        float* o = out + row * V;
        for (int c = lane; c < V; c += WAVE) o[c] = (xr[c] - mx) - ls;
    }
}

// ---------------------------------------------------------------------------
// Small vocabularies, plain log-softmax, rows in REGISTERS (round 3).  KR whole rows (KR <= 4, KR*V a multiple of 4,
// KR*V/4 <= 32 float4) form a 16-byte aligned group; a wave holds one group in lanes 0.. of each 32-lane half, one
// float4 per lane (V = 50: two rows = 25 lanes of 32 busy, the 800 bytes of a wave's two groups contiguous), and
// reduces the row maxima and sums with DPP butterflies inside the half (quad_perm xor 1 / xor 2, row_ror 4 / 8) plus
// one ds_swizzle across its two DPP rows -- no LDS memory, no barrier, one float4 load and one store per lane and
// group, non-temporal both ways.  This is the shape of the fastest plain copy on the part, and it runs at that
// copy's rate: 462 us for the c4 tensor (6.24 TB/s read + write) where the LDS-staged kernel below takes 498
// (tools/ubench/lsm_regs.hip, profiles/r03_ubench_lsm_regs.txt; without the non-temporal hint 484).
// ---------------------------------------------------------------------------
typedef float lsm_f4 __attribute__((ext_vector_type(4)));
template <int CTRL> __device__ __forceinline__ float lsm_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lsm_swz16(float v) {   // lane ^ 16 inside each 32-lane half
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}
__device__ __forceinline__ float half_max32(float v) {
    v = fmaxf(v, lsm_dpp<0xB1>(v)); v = fmaxf(v, lsm_dpp<0x4E>(v)); v = fmaxf(v, lsm_dpp<0x124>(v));
    v = fmaxf(v, lsm_dpp<0x128>(v));
    return fmaxf(v, lsm_swz16(v));
}
__device__ __forceinline__ float half_sum32(float v) {
    v += lsm_dpp<0xB1>(v); v += lsm_dpp<0x4E>(v); v += lsm_dpp<0x124>(v); v += lsm_dpp<0x128>(v);
    return v + lsm_swz16(v);
}
// The KR row maxima of a group in ONE hand-written statement (round 6).  fmaxf() on a DPP result compiles to three
// instructions per butterfly step -- v_mov_b32_dpp, a v_max x,x that quiets a possible signalling NaN, the v_max -- where
// one v_max_f32_dpp does the work; with two rows per group that is 48 of the kernel's 311 vector instructions per wave, and
// the kernel sits AT the vector-issue bound (311 x 900 k waves / (1024 SIMDs x 0.6 G instructions/s) = 456 us of its 462).
// The KR chains are interleaved, so a step's result is two wait states old when the next step reads it through DPP
// (KR = 1, 2: topped up with s_nop); the leading s_nop 1 covers the compiler's instruction that produced the inputs
// (_isa_check.py walks the generated ISA for exactly these).  v_max_f32 returns the other operand for a quiet NaN as
// fmaxf does; a signalling NaN makes the row's maximum NaN and with it the row, which it would be anyway.
// The results leave through a per-wave LDS strip in ADDRESS order (round 6).  A group's segment (V = 50: 400 bytes) starts
// and ends inside 64-byte granules, and what that costs is the STORES: a copy whose stores sit 16 or 32 bytes off the
// 64-byte grid loses 5-19 %, one whose loads do loses nothing (tools/ubench/copy_shape.hip).  Two ds_write_b128 + two
// ds_read_b128 per lane turn the wave's four segments into one store instruction of 1024 contiguous bytes and one of the
// rest, all whole granules: the V=50 micro-benchmark 465 -> 457-460 us, same bits (tools/ubench/lsm_store_policy.hip).
// In the step (bench.py, c4, interleaved processes, profiles/r06_lsm_regs_ab.txt): 0.7997 -> 0.7924 ms on one box, 0.8418 ->
// 0.8323 on another; and with the stores then also written through AND streaming (sc1 nt: whole granules that nothing
// else will add to -- the case in which write-through pays) 0.8323 -> 0.8267.  Blocks of four groups per
// half (RNNT_LSM_REGS_UN=4) lose 20 us.
#ifndef RNNT_LSM_REGS_LINEAR
#define RNNT_LSM_REGS_LINEAR 1
#endif
#ifndef RNNT_LSM_REGS_STORE_WT
#define RNNT_LSM_REGS_STORE_WT RNNT_LSM_REGS_LINEAR
#endif
#ifndef RNNT_LSM_REGS_ASM_MAX
#define RNNT_LSM_REGS_ASM_MAX 1
#endif
#define RNNT_DPPMAX(R, CTRL) "v_max_f32_dpp " R ", " R ", " R " " CTRL " row_mask:0xf bank_mask:0xf\n\t"
#define RNNT_DPPMAX_STEPS(BODY, GAP)                                                                     \
    "s_nop 1\n\t" BODY("quad_perm:[1,0,3,2]") GAP BODY("quad_perm:[2,3,0,1]") GAP BODY("row_ror:4") GAP BODY("row_ror:8")
template <int KR> __device__ __forceinline__ void half_max32_rows(float (&M)[KR]) {
    static_assert(KR >= 1 && KR <= 4, "one to four rows per group");
#if RNNT_LSM_REGS_ASM_MAX
    float t0, t1, t2, t3;
    if constexpr (KR == 1) {
#define RNNT_B1(C) RNNT_DPPMAX("%0", C)
        asm volatile(RNNT_DPPMAX_STEPS(RNNT_B1, "s_nop 1\n\t")
                     "ds_swizzle_b32 %1, %0 offset:swizzle(SWAP,16)\n\ts_waitcnt lgkmcnt(0)\n\tv_max_f32 %0, %0, %1"
                     : "+v"(M[0]), "=&v"(t0));
#undef RNNT_B1
    } else if constexpr (KR == 2) {
#define RNNT_B2(C) RNNT_DPPMAX("%0", C) RNNT_DPPMAX("%1", C)
        asm volatile(RNNT_DPPMAX_STEPS(RNNT_B2, "s_nop 0\n\t")
                     "ds_swizzle_b32 %2, %0 offset:swizzle(SWAP,16)\n\tds_swizzle_b32 %3, %1 offset:swizzle(SWAP,16)\n\t"
                     "s_waitcnt lgkmcnt(0)\n\tv_max_f32 %0, %0, %2\n\tv_max_f32 %1, %1, %3"
                     : "+v"(M[0]), "+v"(M[1]), "=&v"(t0), "=&v"(t1));
#undef RNNT_B2
    } else if constexpr (KR == 3) {
#define RNNT_B3(C) RNNT_DPPMAX("%0", C) RNNT_DPPMAX("%1", C) RNNT_DPPMAX("%2", C)
        asm volatile(RNNT_DPPMAX_STEPS(RNNT_B3, "")
                     "ds_swizzle_b32 %3, %0 offset:swizzle(SWAP,16)\n\tds_swizzle_b32 %4, %1 offset:swizzle(SWAP,16)\n\t"
                     "ds_swizzle_b32 %5, %2 offset:swizzle(SWAP,16)\n\t"
                     "s_waitcnt lgkmcnt(0)\n\tv_max_f32 %0, %0, %3\n\tv_max_f32 %1, %1, %4\n\tv_max_f32 %2, %2, %5"
                     : "+v"(M[0]), "+v"(M[1]), "+v"(M[2]), "=&v"(t0), "=&v"(t1), "=&v"(t2));
#undef RNNT_B3
    } else {
#define RNNT_B4(C) RNNT_DPPMAX("%0", C) RNNT_DPPMAX("%1", C) RNNT_DPPMAX("%2", C) RNNT_DPPMAX("%3", C)
        asm volatile(RNNT_DPPMAX_STEPS(RNNT_B4, "")
                     "ds_swizzle_b32 %4, %0 offset:swizzle(SWAP,16)\n\tds_swizzle_b32 %5, %1 offset:swizzle(SWAP,16)\n\t"
                     "ds_swizzle_b32 %6, %2 offset:swizzle(SWAP,16)\n\tds_swizzle_b32 %7, %3 offset:swizzle(SWAP,16)\n\t"
                     "s_waitcnt lgkmcnt(0)\n\tv_max_f32 %0, %0, %4\n\tv_max_f32 %1, %1, %5\n\tv_max_f32 %2, %2, %6\n\t"
                     "v_max_f32 %3, %3, %7"
                     : "+v"(M[0]), "+v"(M[1]), "+v"(M[2]), "+v"(M[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3));
#undef RNNT_B4
    }
#else
#pragma unroll
    for (int r = 0; r < KR; ++r) M[r] = half_max32(M[r]);
#endif
}
#undef RNNT_DPPMAX_STEPS
#undef RNNT_DPPMAX
#ifndef RNNT_LSM_REGS_UN
#define RNNT_LSM_REGS_UN 2
#endif
constexpr int RG_UN = RNNT_LSM_REGS_UN;   // groups per half and wave, loads first (1: 500 us, 2: 462-484, 4: 482-498)

// NT: bit 0 = non-temporal loads, bit 1 = non-temporal stores (both: 462 us for the c4 tensor, neither: 484)
#ifndef RNNT_LSM_REGS_NT
#define RNNT_LSM_REGS_NT 3
#endif
template <int KR, int NT>
__global__ void __launch_bounds__(256) k_lsm_regs(const float* __restrict__ x, float* __restrict__ out,
                                                  const int64_t ngroups, const int V, const int xcd) {
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const int g4 = (KR * V) >> 2;                  // float4 per group
    const bool act = j < g4;
#if RNNT_LSM_REGS_LINEAR
    __shared__ lsm_f4 strip[4][64 * RG_UN];        // per wave: its 2 * RG_UN groups of <= 32 float4 in address order
    const int wv = threadIdx.x >> 6;
#endif
    // xcd: the eight XCDs (blockIdx mod 8; the grid is a multiple of 8) each stream a contiguous eighth of the groups
    const unsigned wg = xcd ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int64_t w = (int64_t)wg * 4 + (threadIdx.x >> 6);
    const lsm_f4* __restrict__ xin = reinterpret_cast<const lsm_f4*>(x);
    lsm_f4* __restrict__ xout = reinterpret_cast<lsm_f4*>(out);
    // the lane's four elements: the first `ns` of them belong to row r0 of the group, the rest to row r0 + 1
    const int e0 = 4 * j;
    const int r0 = e0 / V;
    const int ns = min(4, (r0 + 1) * V - e0);
    lsm_f4 v[RG_UN];
#pragma unroll
    for (int i = 0; i < RG_UN; ++i) {
        const int64_t g = (w * RG_UN + i) * 2 + half;
        const float ninf = -__builtin_inff();
        v[i] = lsm_f4{ninf, ninf, ninf, ninf};
        if (act && g < ngroups) v[i] = (NT & 1) ? __builtin_nontemporal_load(xin + g * g4 + j) : xin[g * g4 + j];
    }
#pragma unroll
    for (int i = 0; i < RG_UN; ++i) {
        [[maybe_unused]] const int64_t g = (w * RG_UN + i) * 2 + half;
        const lsm_f4 t = v[i];
        const float ninf = -__builtin_inff();
        // maxima of the lane's two parts, then of every row of the group over the half
        const float a0 = t.x, a1 = ns > 1 ? t.y : ninf, a2 = ns > 2 ? t.z : ninf, a3 = ns > 3 ? t.w : ninf;
        const float b1 = ns > 1 ? ninf : t.y, b2 = ns > 2 ? ninf : t.z, b3 = ns > 3 ? ninf : t.w;
        const float mf = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)), ms = fmaxf(b1, fmaxf(b2, b3));
        float M[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r) M[r] = act ? (r0 == r ? mf : (r0 + 1 == r ? ms : ninf)) : ninf;
        half_max32_rows<KR>(M);
        float m_first = M[0], m_second = M[KR - 1];
#pragma unroll
        for (int r = 1; r < KR; ++r) m_first = r0 == r ? M[r] : m_first;
#pragma unroll
        for (int r = KR - 2; r >= 0; --r) m_second = r0 + 1 == r ? M[r] : m_second;
        const float kf = m_first * LOG2E, ks = m_second * LOG2E;
        const float e0x = __builtin_amdgcn_exp2f(__builtin_fmaf(t.x, LOG2E, -kf));
        const float e1x = __builtin_amdgcn_exp2f(__builtin_fmaf(t.y, LOG2E, ns > 1 ? -kf : -ks));
        const float e2x = __builtin_amdgcn_exp2f(__builtin_fmaf(t.z, LOG2E, ns > 2 ? -kf : -ks));
        const float e3x = __builtin_amdgcn_exp2f(__builtin_fmaf(t.w, LOG2E, ns > 3 ? -kf : -ks));
        const float sf = e0x + (ns > 1 ? e1x : 0.f) + (ns > 2 ? e2x : 0.f) + (ns > 3 ? e3x : 0.f);
        const float ss = (ns > 1 ? 0.f : e1x) + (ns > 2 ? 0.f : e2x) + (ns > 3 ? 0.f : e3x);
        // log-sum of every row; the result is (x - max) - log-sum, the association of the LDS-staged kernel (and of
        // torch): subtracting a rounded max + log-sum instead loses an ulp of |max| per element, which the lattice
        // amplifies to 1e-4 on the gradients at c2's size
        float Lg[KR];
#pragma unroll
        for (int r = 0; r < KR; ++r)
            Lg[r] = __builtin_amdgcn_logf(half_sum32(act ? (r0 == r ? sf : (r0 + 1 == r ? ss : 0.f)) : 0.f)) * LN2;
        float l_first = Lg[0], l_second = Lg[KR - 1];
#pragma unroll
        for (int r = 1; r < KR; ++r) l_first = r0 == r ? Lg[r] : l_first;
#pragma unroll
        for (int r = KR - 2; r >= 0; --r) l_second = r0 + 1 == r ? Lg[r] : l_second;
        const lsm_f4 res = lsm_f4{(t.x - m_first) - l_first,
                                  ns > 1 ? (t.y - m_first) - l_first : (t.y - m_second) - l_second,
                                  ns > 2 ? (t.z - m_first) - l_first : (t.z - m_second) - l_second,
                                  ns > 3 ? (t.w - m_first) - l_first : (t.w - m_second) - l_second};
#if RNNT_LSM_REGS_LINEAR
        if (act) strip[wv][(2 * i + half) * g4 + j] = res;
#else
        if (act && g < ngroups) { if (NT & 2) __builtin_nontemporal_store(res, xout + g * g4 + j); else xout[g * g4 + j] = res; }
#endif
    }
#if RNNT_LSM_REGS_LINEAR
    // the wave's 2 * RG_UN groups are 64 * g4 contiguous bytes: out of the strip in address order, one store instruction of
    // 1024 bytes and one of the rest -- every instruction whole 64-byte granules (the strip is the wave's own: no barrier)
    wave_sync_lds();
    const int64_t f0 = w * (2 * RG_UN) * g4, nf = ngroups * g4;
    const int nw = 2 * RG_UN * g4;                  // float4 of this wave (<= 64 * RG_UN)
#pragma unroll
    for (int k = 0; k < RG_UN; ++k) {
        const int f = k * 64 + lane;
        if (f < nw && f0 + f < nf) {
            const lsm_f4 r = strip[wv][f];
#if RNNT_LSM_REGS_STORE_WT      // written through and streaming (the s_nop: _isa_check.py, second rule)
            asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(xout + f0 + f), "v"(r) : "memory");
#else
            if (NT & 2) __builtin_nontemporal_store(r, xout + f0 + f); else xout[f0 + f] = r;
#endif
        }
    }
#endif
}

// ---------------------------------------------------------------------------
// Rows in registers, L lanes per row (round 4; fused gather, V a multiple of 4).  Every row is 16-byte aligned, so
// its L lanes load Q4 float4 each straight from HBM (a row instruction reads L*16 contiguous bytes: whole 128-byte lines
// from L = 8 on), reduce with L-wide butterflies and never touch LDS; a wave carries UN passes of 64/L rows, all loads
// issued before the first use.  The LDS-staged kernel spends 333 VALU instructions per wave on the same work at V = 128
// (run-time column loops, two LDS reads per element; SQ counters: the vector ALUs 67 % busy at 4.7 TB/s,
// profiles/r04_lsm_rows_ab.txt), this one about 150.  Fused gather only: as the plain log-softmax it runs at the rate of
// the kernels that serve it now (V = 160 ... 600: 5.5-5.7 TB/s either way), so that mode is not instantiated.
//   One lane per row (all rows of the wave in one go: the index arithmetic of map_cell is paid once per
//   wave) fetches the row's blank and label logits again -- the wave has just read those lines -- and stores the pair.
// ---------------------------------------------------------------------------
// all-reduce over aligned groups of L lanes on DPP (quad permutes, then the mirrors: once every lane of a quad holds the
// quad's value, reversing 8 / 16 lanes swaps whole quads / halves), lane ^ 16 on ds_swizzle, lane ^ 32 on a permute
template <int L, bool MAX> __device__ __forceinline__ float lsm_group_reduce(float v) {
#define LSM_STEP(w) v = MAX ? fmaxf(v, (w)) : v + (w)
    if constexpr (L >= 2) LSM_STEP(lsm_dpp<0xB1>(v));
    if constexpr (L >= 4) LSM_STEP(lsm_dpp<0x4E>(v));
    if constexpr (L >= 8) LSM_STEP(lsm_dpp<0x141>(v));
    if constexpr (L >= 16) LSM_STEP(lsm_dpp<0x140>(v));
    if constexpr (L >= 32) LSM_STEP(lsm_swz16(v));
    if constexpr (L >= 64) LSM_STEP(__shfl_xor(v, 32, WAVE));
#undef LSM_STEP
    return v;
}

// passes per wave: two for the 8-lane rows (V <= 128: 16 rows = 8 KB per wave at V = 128), one above (V = 256 ... 1024:
// 299 / 998 us against 306 / 1057 with two, profiles/r04_lsm_rows_ab.txt)
template <int L> struct RowsShape {
    static constexpr int UN = L <= 8 ? 2 : 1;
    static constexpr int RW = WAVE / L;            // rows per pass
    static constexpr int RPW = RW * UN;            // rows per wave
};

// loads (all passes first), row maxima and log-sums of the rows at src[p] (one pointer per lane and pass: the lane's
// first float4 of its row)
template <int L, int Q, int MODE>
__device__ __forceinline__ void lsm_rows_stats(const float* const (&src)[RowsShape<L>::UN], bool last_ok,
                                               float (&mx)[RowsShape<L>::UN], float (&ls)[RowsShape<L>::UN]) {
    constexpr int UN = RowsShape<L>::UN, VEC = 4;
    const float ninf = -__builtin_inff();
    const unsigned last_off = last_ok ? (Q - 1) * L : 0;   // a last float4 past the row: re-read the first, made -inf
    float v[UN][Q][VEC];
#pragma unroll
    for (int p = 0; p < UN; ++p) {
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const unsigned off = i < Q - 1 ? i * L : last_off;
            const float4 t = RNNT_LSM_LOAD(reinterpret_cast<const float4*>(src[p]) + off);
            v[p][i][0] = t.x; v[p][i][1] = t.y; v[p][i][2] = t.z; v[p][i][3] = t.w;
        }
    }
#pragma unroll
    for (int p = 0; p < UN; ++p)
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            if (!last_ok) v[p][Q - 1][e] = ninf;
#pragma unroll
    for (int p = 0; p < UN; ++p) {
        float m = ninf;
#pragma unroll
        for (int i = 0; i < Q; ++i)
#pragma unroll
            for (int e = 0; e < VEC; ++e) m = fmaxf(m, v[p][i][e]);
        m = lsm_group_reduce<L, true>(m);
        const float mb = -m * LOG2E;
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < Q; ++i)
#pragma unroll
            for (int e = 0; e < VEC; ++e) s += __builtin_amdgcn_exp2f(__builtin_fmaf(v[p][i][e], LOG2E, mb));
        s = lsm_group_reduce<L, false>(s);
        mx[p] = m;
        ls[p] = __builtin_amdgcn_logf(s) * LN2;
    }
}

// lane l < RPW picks up the statistics of row l of the wave: first lane of group l % RW, pass l / RW
template <int L>
__device__ __forceinline__ void lsm_rows_stats_of_lane(int lane, const float (&mx)[RowsShape<L>::UN],
                                                       const float (&ls)[RowsShape<L>::UN], float& m, float& lg) {
    constexpr int UN = RowsShape<L>::UN, RW = RowsShape<L>::RW;
    const int srcl = (lane % RW) * L;
    m = 0.0f;
    lg = 0.0f;
#pragma unroll
    for (int p = 0; p < UN; ++p) {
        const float mp = __shfl(mx[p], srcl, WAVE), lp = __shfl(ls[p], srcl, WAVE);
        if (lane / RW == p) { m = mp; lg = lp; }
    }
}

// consecutive rows per wave
template <int L, int Q>
__global__ void __launch_bounds__(256)
k_lsm_rows(const float* x, float* out, const int* __restrict__ labels, int64_t rows, int V, int T, int U, int blank) {
    constexpr int MODE = LSM_GATHER, VEC = 4;
    constexpr int UN = RowsShape<L>::UN, RW = RowsShape<L>::RW, RPW = RowsShape<L>::RPW;
    const int lane = threadIdx.x & 63, h = lane % L, rr = lane / L;
    // wave-uniform values kept in scalar registers (the 64-bit row arithmetic runs on the scalar unit)
    const int64_t row0 = ((int64_t)stream_block<XCD_LSM_ROWS>() * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const bool last_ok = (h + (Q - 1) * L) * VEC < V;   // the lane's last float4 is part of the row
    const bool whole = row0 + RPW <= rows;              // (uniform) every row of this wave exists
    const float* const wave_src = x + row0 * V;
    const unsigned lane_off = (unsigned)rr * (unsigned)V + (unsigned)h * VEC;      // floats inside a pass
    // lane l < RPW owns the pair of row row0 + l.  Its two logits are requested FIRST, next to the row loads that bring
    // the same lines (asked for after the rows have streamed through, they are fetched a second time: forward 252 vs 216
    // us at V = 128, profiles/r04_lsm_rows_ab.txt)
    CellMap cm = {0, 0, 0};
    float xb = 0.0f, xl = 0.0f;
    const bool own = lane < RPW && row0 + lane < rows;
    if (own) {
        cm = map_cell((size_t)(row0 + lane), labels, T, U, V, blank);
        const float* xr = x + (row0 + lane) * V;
        xb = xr[blank];
        xl = xr[cm.label];
    }
    const float* src[UN];
#pragma unroll
    for (int p = 0; p < UN; ++p) {
        src[p] = wave_src + (size_t)(p * RW) * V + lane_off;
        // (rows past the end of the tensor -- last wave only -- re-read the last row and are dropped at the stores)
        if (!whole && row0 + p * RW + rr >= rows) src[p] = x + (rows - 1) * V + h * VEC;
    }
    float mx[UN], ls[UN];
    lsm_rows_stats<L, Q, MODE>(src, last_ok, mx, ls);
    float m, lg;
    lsm_rows_stats_of_lane<L>(lane, mx, ls, m, lg);
#ifdef RNNT_LSM_ROWS_PROBE_LINEAR_STORE      // timing probe only (wrong layout): what the scattered 8-byte stores cost
    if (own) reinterpret_cast<float2*>(out)[row0 + lane] = make_float2((xb - m) - lg, (xl - m) - lg);
#else
    if (own) reinterpret_cast<float2*>(out)[cm.sk] = make_float2((xb - m) - lg, (xl - m) - lg);
#endif
}

// Along the diagonals (rows that are one or two whole 128-byte lines: V = 32, 64; T >= 16): a wave takes the 16 cells
// (t' - k mod T, u0 + k), k = 0 ... 15 -- one run of 16 consecutive pairs of the diagonal-major plane, stored as one
// 128-byte piece -- instead of 16 consecutive rows, whose pairs land 8 bytes each in 16 different lines (counters: 32
// bytes written per pair; with the pairs stored linearly the kernel is 12-17 us of 145 faster at V = 128, N*T*U = 1.6 M).
// V = 32 / 64: forward 96.5 / 134 us against 102 / 144; from V = 96 on the scattered rows cost what the stores save (174
// vs 177, 199 vs 193: consecutive rows kept there).  No index division: grid = (T / 4 rounded up, column blocks of 16, N).
template <int Q>
__global__ void __launch_bounds__(256)
k_lsm_rows_diag(const float* x, float* out, const int* __restrict__ labels, int V, int T, int U, int blank) {
    constexpr int L = 8, VEC = 4, MODE = LSM_GATHER;
    constexpr int UN = RowsShape<L>::UN, RW = RowsShape<L>::RW, RPW = RowsShape<L>::RPW;
    static_assert(RPW == 16, "one run of 16 pairs per wave");
    const int lane = threadIdx.x & 63, h = lane % L, rr = lane / L;
    const int tp = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (tp >= T) return;
    const int u0 = (int)blockIdx.y * RPW, n = (int)blockIdx.z;
    const bool last_ok = (h + (Q - 1) * L) * VEC < V;
    const size_t plane = (size_t)n * T;            // frames in front of this utterance
    // the pair of cell k = lane (lane < 16), requested first
    float xb = 0.0f, xl = 0.0f;
    const bool own = lane < RPW && u0 + lane < U;
    if (own) {
        const int u = u0 + lane;
        int t = tp - lane;
        t += t < 0 ? T : 0;
        const int lab = (u < U - 1) ? safe_label(labels[(size_t)n * (U - 1) + u], V, blank) : blank;
        const float* xr = x + ((plane + t) * U + u) * V;
        xb = xr[blank];
        xl = xr[lab];
    }
    const float* src[UN];
#pragma unroll
    for (int p = 0; p < UN; ++p) {
        const int k = p * RW + rr;
        const int u = min(u0 + k, U - 1);          // (columns past the plane re-read the last one and are dropped)
        int t = tp - k;
        t += t < 0 ? T : 0;
        src[p] = x + ((plane + t) * U + u) * V + h * VEC;
    }
    float mx[UN], ls[UN];
    lsm_rows_stats<L, Q, MODE>(src, last_ok, mx, ls);
    float m, lg;
    lsm_rows_stats_of_lane<L>(lane, mx, ls, m, lg);
    int r = tp + u0;
    r = r >= T ? r % T : r;
    if (own) reinterpret_cast<float2*>(out)[(plane + r) * U + u0 + lane] = make_float2((xb - m) - lg, (xl - m) - lg);
}

// rows per group for k_lsm_regs, or 0 when the kernel does not fit V: the largest KR <= 4 with KR*V a multiple of 4 and
// KR*V/4 <= 32 lanes, if it keeps at least 20 of the 32 lanes of a half busy
static int lsm_regs_rows_per_group(int V) {
    if (V < 4) return 0;
    int best = 0;
    for (int k = 1; k <= 4; ++k)
        if ((k * V) % 4 == 0 && (k * V) / 4 <= 32) best = k;
    return (best && (best * V) / 4 >= 20) ? best : 0;
}

template <int MODE>
static hipError_t dispatch_lsm(hipStream_t stream, const float* x, float* out, const int* labels,
                               int64_t rows, int V, int T, int U, int blank, LsmBwd bw) {
    constexpr bool GATHER = MODE == LSM_GATHER;
    if (rows <= 0) return hipSuccess;
    const bool aligned = (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                         (GATHER || reinterpret_cast<uintptr_t>(out) % 16 == 0);
    if constexpr (MODE == LSM_NORM) {
        // rows in registers where the vocabulary allows it (RNNT_LSM_NO_REGS=1: the LDS-staged kernel, for A/B runs)
        static const bool no_regs = ab_getenv("RNNT_LSM_NO_REGS") != nullptr;
        // (below V = 32 -- four rows per group -- the LDS-staged kernel with its straight-line row pass is the faster one
        //  since round 4: the c4 lattice with V=24 0.584 -> 0.536 ms per step, V=28 0.589-0.605 -> 0.584, c2 0.0343 -> 0.0336;
        //  from V = 32 on this kernel wins inside the step: V=40 0.71 vs 0.73, V=50 0.870 vs 0.893; tools/step_rate.py)
        const int kr = (aligned && !no_regs && V >= 32) ? lsm_regs_rows_per_group(V) : 0;
        if (kr && rows >= kr) {
            const int64_t ngroups = rows / kr;
            const int64_t per_wg = 4 * RG_UN * 2;               // 4 waves x RG_UN groups x 2 halves
            int64_t grid = (ngroups + per_wg - 1) / per_wg;
            // every XCD streams a contiguous eighth of the tensor (as the row-per-workgroup kernel below; here it costs two
            // scalar instructions): V=50 1.44 GB equal, 5.76 GB 5.66 -> 5.86 TB/s, V=64 6.25 -> 6.40, 100 5.92 -> 6.19, 128
            // 6.15 -> 6.44; the c4 step in bench.py 0.8759 / 0.8781 / 0.8779 -> 0.8726 / 0.8702 / 0.8709 ms, three
            // interleaved pairs (profiles/r04_lsm_xcd_order_ab.txt).  RNNT_LSM_REGS_XCD=0: the plain order (A/B runs)
            static const int regs_xcd = ab_getenv("RNNT_LSM_REGS_XCD") ? atoi(ab_getenv("RNNT_LSM_REGS_XCD")) : 1;
            if (regs_xcd) grid = (grid + 7) / 8 * 8;
            if (grid < ((int64_t)1 << 31)) {
#define LSM_REGS(KR) \
    case KR: k_lsm_regs<KR, RNNT_LSM_REGS_NT><<<(unsigned)grid, 256, 0, stream>>>(x, out, ngroups, V, regs_xcd); break;
                switch (kr) { LSM_REGS(1) LSM_REGS(2) LSM_REGS(3) LSM_REGS(4) }
#undef LSM_REGS
                const hipError_t e = hipGetLastError();
                const int64_t done = ngroups * kr;              // (a group boundary: 16-byte aligned)
                if (e != hipSuccess || done == rows) return e;
                return dispatch_lsm<MODE>(stream, x + done * V, out + done * V, labels, rows - done, V, T, U, blank, bw);
            }
        }
    }
    if constexpr (MODE == LSM_NORM) {
        // 128 < V <= 1024 whose rows fill a cover of 64 ... 256 threads x one float4 (or an exact 64x2 / 64x3 / 128x2):
        // the row-in-registers kernel below, one row per small workgroup, instead of the LDS-staged tiles.  Measured
        // round 3 (tools/lsm_rate.py, 1.44 GB in, TB/s in + out, LDS-staged -> registers; profiles/r03_lsm_midv_probe.txt):
        // V=256 5.82 -> 6.44, 496 5.33 -> 6.12, 500 5.14 -> 5.90, 512 5.77 -> 6.47, 768 5.58 -> 6.15, 980 5.26 -> 6.00,
        // 1000 5.14 -> 6.10, 1024 5.76 -> 6.59; with 94 % of the lanes busy still +4 ... +10 % (484, 724, 964), below
        // that -- and below 98 % for a single wave (V=244: 5.53 -> 5.23) -- the tiles win (V=200, 400, 600: 78 / 59 %).
        static const bool no_lgr = ab_getenv("RNNT_LSM_NO_LGR") != nullptr;    // A/B runs: the LDS-staged kernel instead
        if (aligned && !no_lgr && V % 4 == 0 && V > 128 && V <= 1024) {
            const int nvec = V >> 2, th = (nvec + 63) / 64 * 64;
            const unsigned grid = (unsigned)(rows < (1 << 22) ? rows : (1 << 22));
#define LGR(TH, NV) { k_lsm_large<MODE, TH, NV><<<grid, TH, 0, stream>>>(x, out, labels, rows, V, T, U, blank, bw); return hipGetLastError(); }
            if (nvec == 64) LGR(64, 1)
            if (nvec == 128) LGR(64, 2)
            if (nvec == 192) LGR(64, 3)
            if (nvec == 256) LGR(128, 2)
            if (th == 64 && nvec >= 63) LGR(64, 1)
            if (th == 128 && nvec * 100 >= th * 94) LGR(128, 1)
            if (th == 192 && nvec * 100 >= th * 94) LGR(192, 1)
            if (th == 256 && nvec * 100 >= th * 94) LGR(256, 1)
#undef LGR
        }
    }
    if constexpr (MODE == LSM_GATHER) {
        // V % 4 != 0 (c4's own V = 50: rows that pack into 16-byte groups only in twos) stays on the LDS-staged kernel below.
        // Round 5 tried the rows-in-registers loads of k_lsm_regs for it once more, with what round 4 had learnt on
        // k_lsm_rows -- the lane that stores a row's pair asks for its two logits itself, ahead of the group loads, or picks
        // them out of an LDS copy of the groups: whole fused forward at c4 503-512 us (two and four groups per half-wave:
        // 558 / 503; LDS copy 512) against 419-433 for the LDS-staged kernel then, and ~400 since its staging loop issues
        // its loads first (k_lsm_small; the kernel alone 290 -> 225-255 us, 5.6-6.4 TB/s read against 7.0 for a bare
        // read-only stream, tools/ubench/copy_rate.hip, profiles/r05_loads_first_ab.txt).
        // Rows in registers, L lanes per row (k_lsm_rows), against the LDS-staged kernel below -- re-measured after that
        // kernel got its straight-line row pass (forward of the fused entry, N=32, T=500, U=100, us, k_lsm_rows / LDS tiles;
        // tools/fused_rate.py, profiles/r04_lsm_rows_ab.txt section 9): V=32 97 / 127, 64 140 / 146, 128 187 / 197, 256 320 /
        // 341, 320 393 / 404; 448 512 / 522, 480 532 / 554, 500 561 / 583, 512 505 / 587, 544 609 / 753, 640 668 / 749, 768
        // 791 / 812, 896 850 / 930, 1000 986 / 1089, 1024 1016 / 1128; but 96 180 / 163, 160 256 / 237, 192 285 / 262, 224 302 /
        // 290, 352 426 / 418, 384 456 / 426, 400 510 / 485, and everything whose 8- or 16-lane row instructions straddle
        // lines (V=100: 236 / 187, 132: 281 / 257) or needs float2 rows (V=50: 169 / 132).  Rule: the powers of two from 32
        // to 256, and every V % 4 == 0 from 448 on.
        // (RNNT_LSM_NO_ROWS=1: the LDS-staged kernel, for A/B runs; RNNT_LSM_NO_DIAG=1: consecutive rows per wave for
        //  every V; RNNT_LSM_ROWS_ANY=1: this kernel for every V % 4 == 0)
        static const bool no_rows = ab_getenv("RNNT_LSM_NO_ROWS") != nullptr;
        static const bool no_diag = ab_getenv("RNNT_LSM_NO_DIAG") != nullptr;
        static const bool rows_any = ab_getenv("RNNT_LSM_ROWS_ANY") != nullptr;
        const bool rows_rule = V == 32 || V == 64 || V == 128 || V == 256 || V >= 448;
        if (aligned && !no_rows && V % 4 == 0 && V >= 32 && V <= 1024 && (rows_rule || rows_any)) {
            int L = 8;
            while (L < 64 && L * 16 < V) L <<= 1;
            const int q = (V / 4 + L - 1) / L;         // 1 ... 4
            const int64_t N = rows / ((int64_t)T * U), nub = (U + 15) / 16;
            if (V <= 64 && V % 32 == 0 && T >= 16 && !no_diag && N <= 65535 && nub <= 65535) {
                const dim3 grid((unsigned)((T + 3) / 4), (unsigned)nub, (unsigned)N);
                if (q == 1) k_lsm_rows_diag<1><<<grid, 256, 0, stream>>>(x, out, labels, V, T, U, blank);
                else k_lsm_rows_diag<2><<<grid, 256, 0, stream>>>(x, out, labels, V, T, U, blank);
                return hipGetLastError();
            }
            const int64_t rpw = L <= 8 ? 2 * (WAVE / L) : WAVE / L;       // RowsShape<L>::RPW
            const int64_t grid = stream_grid<XCD_LSM_ROWS>((unsigned)((rows + 4 * rpw - 1) / (4 * rpw)));
            if ((rows + 4 * rpw - 1) / (4 * rpw) < ((int64_t)1 << 31) - 8) {
#define LSM_ROWS(LL, QQ) \
    if (L == LL && q == QQ) k_lsm_rows<LL, QQ><<<(unsigned)grid, 256, 0, stream>>>(x, out, labels, rows, V, T, U, blank);
#define LSM_ROWS_L(LL) LSM_ROWS(LL, 1) LSM_ROWS(LL, 2) LSM_ROWS(LL, 3) LSM_ROWS(LL, 4)
                LSM_ROWS_L(8) LSM_ROWS_L(16) LSM_ROWS_L(32) LSM_ROWS_L(64)
#undef LSM_ROWS_L
#undef LSM_ROWS
                return hipGetLastError();
            }
        }
    }
    if (aligned && V <= 1024) {
        int L = 1;
        while (L < 64 && L * 16 < V) L <<= 1;          // <= 16 columns per lane
#ifdef RNNT_LG_PROBE
        if (const char* e = ab_getenv("RNNT_LSM_L")) { const int l = atoi(e); if (l >= 1 && l <= 64 && (l & (l - 1)) == 0) L = l; }
#endif
        const int q = (V + L - 1) / L;
        const int rpp = SM_THREADS / L;                // rows per pass, a multiple of 4
        int R = (SM_FLOATS / V) / rpp * rpp;           // whole passes
        if (R < rpp) R = rpp;
        // wave-private tiles: each wave owns WAVE/L rows (a multiple of 4 for L <= 16), one pass
        static const bool no_wp = ab_getenv("RNNT_LSM_NO_WP") != nullptr;
        // Plain log-softmax only: measured 2-3 % faster there (0.506 -> 0.493 ms at c4), slower for the fused
        // gather (its one-lane-per-row mapping phase wants all rows of the tile in ONE wave: 0.52 -> 0.556 ms)
        // and for the fused backward (+15 us).
        static const bool wp_fused = ab_getenv("RNNT_LSM_WP_FUSED") != nullptr;      // (A/B: the wave-private form in the fused modes)
        const bool wp = (L <= 16) && !no_wp && (MODE == LSM_NORM || wp_fused);
        if (wp) R = rpp;
        size_t lds = (size_t)R * V * sizeof(float) + (GATHER ? (size_t)R * sizeof(float2) : 0);
#ifdef RNNT_LG_PROBE      // probe: fewer resident workgroups per CU (LDS the kernel does not use)
        if (const char* e = ab_getenv("RNNT_LSM_LDS")) { const size_t want = (size_t)atoi(e); if (want > lds && want <= 65536) lds = want; }
#endif
        const unsigned grid = stream_grid<XCD_LSM_SMALL>((unsigned)((rows + R - 1) / R));
#define LSM_SMALL(LL)                                                                           \
    case LL:                                                                                    \
        if (wp && LL <= 16)                                                                     \
            k_lsm_small<LL, MODE, (LL <= 16)><<<grid, SM_THREADS, lds, stream>>>(x, out, labels, rows, V, R, q, \
                                                                                 T, U, blank, bw);             \
        else                                                                                    \
            k_lsm_small<LL, MODE, false><<<grid, SM_THREADS, lds, stream>>>(x, out, labels, rows, V, R, q, T,   \
                                                                            U, blank, bw);                     \
        break;
        switch (L) {
            LSM_SMALL(1) LSM_SMALL(2) LSM_SMALL(4) LSM_SMALL(8) LSM_SMALL(16) LSM_SMALL(32)
            LSM_SMALL(64)
        }
#undef LSM_SMALL
    } else if (aligned && V % 4 == 0 && V <= LG_MAXV) {
        // Which rows an XCD streams (plain log-softmax only).  Workgroups go to the eight XCDs by blockIdx mod 8, so with
        // row = work item every XCD reads every eighth row of one moving front; with bw.xcd each streams a contiguous
        // eighth of the tensor.  Measured (tools/lsm_rate.py, TB/s in + out, every-eighth / contiguous, 1.92 GB in; 8 GB
        // in brackets; profiles/r04_lsm_xcd_order_ab.txt): V=1500 5.9 / 6.2, 3000 6.0 / 6.2 [5.95 / 6.6], 5000 5.8-5.9 /
        // 6.0-6.5 [5.7 / 6.1], 7168 6.2 / 6.4, 8192 6.0-6.2 / 6.3-6.4 [5.8 / 6.2], 16384 5.2-5.4 / 6.0 [5.3 / 6.2];
        // nothing at 2048, 4096, 5120 ... 6144, 12288; WORSE for the three-pass covers of 2048 < V/4 <= 3072 (V=10000:
        // 6.0 / 5.6 [5.9 / 5.6]), which keep the plain order.  RNNT_LG_XCD=0 / 1 forces one or the other (A/B runs).
        static const int xcd_force = ab_getenv("RNNT_LG_XCD") ? atoi(ab_getenv("RNNT_LG_XCD")) : -1;
        // (fused gather / backward modes: no difference at c3 -- fused forward 0.3196 / 0.3184 / 0.3181 vs 0.3186 / 0.3178 /
        //  0.3190 ms -- so they keep the plain order; RNNT_LG_XCD_FUSED=1 to try)
        static const int xcd_fused = ab_getenv("RNNT_LG_XCD_FUSED") ? atoi(ab_getenv("RNNT_LG_XCD_FUSED")) : 0;
        if constexpr (MODE == LSM_NORM) {
            const int nv4 = V >> 2;
            bw.xcd = xcd_force >= 0 ? (xcd_force != 0) : !(nv4 > 2048 && nv4 <= 3072);
        } else {
            bw.xcd = xcd_fused;
        }
        unsigned grid = (unsigned)(rows < (1 << 22) ? rows : (1 << 22));
        if (bw.xcd) grid = (grid + 7u) & ~7u;
#ifdef RNNT_LG_PROBE
        if (const char* e = ab_getenv("RNNT_LG_VARIANT")) {
            int th = 0, nv = 0;
            sscanf(e, "%d,%d", &th, &nv);
#define LGV(TH, NV) if (th == TH && nv == NV && V <= TH * 4 * NV) { k_lsm_large<MODE, TH, NV><<<grid, TH, 0, stream>>>(x, out, labels, rows, V, T, U, blank, bw); return hipGetLastError(); }
            LGV(64, 20) LGV(64, 40) LGV(128, 10) LGV(128, 20) LGV(256, 2) LGV(256, 4) LGV(256, 8) LGV(256, 16) LGV(512, 2) LGV(512, 4) LGV(512, 8) LGV(1024, 2) LGV(1024, 4)
#undef LGV
        }
#endif
        if constexpr (MODE == LSM_NORM) {
            // The read + write stream wants about two float4 per thread and (nearly) every thread busy in every pass;
            // workgroups of 512 or 1024 threads (which tile a CU's 2048 exactly) beat the sizes in between.  Round 2
            // (profiles/r02_lsm_large_variants.txt, threads x passes, us for ~1.9 GB in + out): V=3000 256x3 734 /
            // 384x2 663; V=8192 256x8 687 / 1024x2 666; V=10000 512x5 870 / 1024x3 828-834 / 896x3 811; V=16384 512x8
            // 707 / 1024x4 723.  Re-swept in round 3 with the non-temporal policies in place
            // (profiles/r03_xcd_run_order_probe.txt part 3, profiles/r03_lg_cover_ab.txt; TB/s in + out): V=5000 640x2
            // 5.71 / 512x3 5.79-5.82 (c3 in bench.py: 0.696 -> 0.680 ms); V=5120 640x2 5.90 / 512x3 6.10; V=5600 768x2
            // 5.96 / 512x3 6.17; V=6144 768x2 6.23 / 512x3 5.99 / 1024x2 6.11; V=7168 896x2 5.90 / 1024x2 6.16.
            // The thread count is a template parameter on purpose (the same kernel with blockDim.x read at run
            // time: 780 us at V=5000).
            const int nvec = V >> 2;
#define LGN(TH, NV) case TH: k_lsm_large<MODE, TH, NV><<<grid, TH, 0, stream>>>(x, out, labels, rows, V, T, U, blank, bw); break;
            if (nvec > 3072) {
                k_lsm_large<MODE, 512, 8><<<grid, 512, 0, stream>>>(x, out, labels, rows, V, T, U, blank, bw);
            } else if (nvec > 2048) {
                const int th = (nvec + 383) / 384 * 128;
                switch (th) { LGN(768, 3) LGN(896, 3) LGN(1024, 3) }
            } else if (nvec > 1536) {
                k_lsm_large<MODE, 1024, 2><<<grid, 1024, 0, stream>>>(x, out, labels, rows, V, T, U, blank, bw);
            } else if (nvec > 1408) {
                k_lsm_large<MODE, 768, 2><<<grid, 768, 0, stream>>>(x, out, labels, rows, V, T, U, blank, bw);
            } else if (nvec > 1024) {
                k_lsm_large<MODE, 512, 3><<<grid, 512, 0, stream>>>(x, out, labels, rows, V, T, U, blank, bw);
            } else {
                int th = (nvec + 255) / 256 * 128;
                th = th < 256 ? 256 : th;
                switch (th) { LGN(256, 2) LGN(384, 2) LGN(512, 2) }
            }
#undef LGN
        } else {
            // read-mostly modes (fused gather, fused backward): the smallest cover, for the residency
            if (V <= 4096)
                k_lsm_large<MODE, 256, 4><<<grid, 256, 0, stream>>>(x, out, labels, rows, V, T, U, blank, bw);
            else if (V <= 8192)
                k_lsm_large<MODE, 256, 8><<<grid, 256, 0, stream>>>(x, out, labels, rows, V, T, U, blank, bw);
            else
                k_lsm_large<MODE, 512, 8><<<grid, 512, 0, stream>>>(x, out, labels, rows, V, T, U, blank, bw);
        }
    } else {
        k_lsm_generic<MODE><<<(unsigned)((rows + 3) / 4), 256, 0, stream>>>(x, out, labels, rows, V, T,
                                                                             U, blank, bw);
    }
    return hipGetLastError();
}

hipError_t launch_log_softmax(hipStream_t stream, const float* x, float* out, int64_t rows, int V) {
    return dispatch_lsm<LSM_NORM>(stream, x, out, nullptr, rows, V, 1, 1, 0, LsmBwd{nullptr, nullptr});
}

hipError_t launch_log_softmax_gather_skewed(hipStream_t stream, const float* logits, const int* labels,
                                            float* ws2, int N, int T, int U, int V, int blank) {
    return dispatch_lsm<LSM_GATHER>(stream, logits, ws2, labels, (int64_t)N * T * U, V, T, U, blank,
                                    LsmBwd{nullptr, nullptr});
}

hipError_t launch_logits_backward(hipStream_t stream, const float* logits, const int* labels,
                                  const float* g2_diagonal, const float* scale, float* dlogits, int N, int T,
                                  int U, int V, int blank) {
    return dispatch_lsm<LSM_BWD>(stream, logits, dlogits, labels, (int64_t)N * T * U, V, T, U, blank,
                                 LsmBwd{reinterpret_cast<const float2*>(g2_diagonal), scale});
}

// ---------------------------------------------------------------------------
// cache policy of the log-softmax backward streams: bit 0 = non-temporal loads (dy, y), bit 1 = non-temporal stores (dx)
// Both (round 3): the reference's call chain with the native log-softmax autograd function 1.96 -> 1.89 ms per training
// step at c4 (with the expand kernel's non-temporal stores on top: 1.84), profiles/r03_bwd_nt_ab.txt.
#ifndef RNNT_LSMB_NT
#define RNNT_LSMB_NT 3
#endif
#define LSMB_LD(p) ((RNNT_LSMB_NT & 1) ? rnnt_nt_load4(p) : *(p))
#define LSMB_ST(p, v) do { if (RNNT_LSMB_NT & 2) rnnt_nt_store4((p), (v)); else *(p) = (v); } while (0)

// Backward of log-softmax: dx = dy - exp(y) * sum_v(dy), y = the log-probabilities.
// Same three shapes as the forward kernels (LDS row tiles / one row per workgroup in
// registers / wave per row); 12V bytes per row element (read dy, read y, write dx).
// ---------------------------------------------------------------------------
constexpr int SMB_THREADS = 256;   // backward: two tiles per workgroup, keep the 256-thread shape
constexpr int SMB_FLOATS = 3200;
template <int L>
__global__ void __launch_bounds__(SMB_THREADS)
k_lsmbwd_small(const float* dy, const float* y, float* dx, int64_t rows, int V, int R, int q) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    float* tdy = tile;
    float* ty = tile + (size_t)R * V;
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)stream_block<XCD_LSMBWD_SMALL>() * R;
    if (row0 >= rows) return;
    const int nrows = (int)min((int64_t)R, rows - row0);
    const int nel = nrows * V, nvec = nel >> 2;
    const float* sdy = dy + row0 * V;
    const float* sy = y + row0 * V;
    for (int i = tid; i < nvec; i += SMB_THREADS) {
        reinterpret_cast<float4*>(tdy)[i] = LSMB_LD(reinterpret_cast<const float4*>(sdy) + i);
        reinterpret_cast<float4*>(ty)[i] = LSMB_LD(reinterpret_cast<const float4*>(sy) + i);
    }
    for (int e = (nvec << 2) + tid; e < nel; e += SMB_THREADS) { tdy[e] = sdy[e]; ty[e] = sy[e]; }
    __syncthreads();
    constexpr int RPP = SMB_THREADS / L;
    const int h = tid % L, rr = tid / L;
    const int ctail = h + (q - 1) * L;
    const bool tail_ok = ctail < V;
    // (q = 9 ... 16 as a compile-time constant: both rows read once into registers by straight-line code, as k_lsm_small)
    auto all_rows = [&](auto QC) {
        constexpr int Q = decltype(QC)::value;
        for (int r = rr; r < nrows; r += RPP) {
            float* rdy = tdy + r * V;
            const float* ry = ty + r * V;
            float g[Q], yv[Q];
#pragma unroll
            for (int i = 0; i < Q - 1; ++i) { g[i] = rdy[h + i * L]; yv[i] = ry[h + i * L]; }
            g[Q - 1] = tail_ok ? rdy[ctail] : 0.0f;
            yv[Q - 1] = tail_ok ? ry[ctail] : 0.0f;
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < Q; ++i) s += g[i];
            s = group_sum<L>(s);
#pragma unroll
            for (int i = 0; i < Q - 1; ++i)
                rdy[h + i * L] = __builtin_fmaf(-__builtin_amdgcn_exp2f(yv[i] * LOG2E), s, g[i]);
            if (tail_ok) rdy[ctail] = __builtin_fmaf(-__builtin_amdgcn_exp2f(yv[Q - 1] * LOG2E), s, g[Q - 1]);
        }
    };
    switch (q) {
#define LSMB_Q(QQ) case QQ: all_rows(std::integral_constant<int, QQ>{}); break;
        LSMB_Q(9) LSMB_Q(10) LSMB_Q(11) LSMB_Q(12) LSMB_Q(13) LSMB_Q(14) LSMB_Q(15) LSMB_Q(16)
#undef LSMB_Q
        default:
    for (int r = rr; r < nrows; r += RPP) {
        float* rdy = tdy + r * V;
        const float* ry = ty + r * V;
        float s = 0.0f;
        for (int i = 0, c = h; i < q - 1; ++i, c += L) s += rdy[c];
        if (tail_ok) s += rdy[ctail];
        s = group_sum<L>(s);
        for (int i = 0, c = h; i < q - 1; ++i, c += L)
            rdy[c] = __builtin_fmaf(-__builtin_amdgcn_exp2f(ry[c] * LOG2E), s, rdy[c]);
        if (tail_ok) rdy[ctail] = __builtin_fmaf(-__builtin_amdgcn_exp2f(ry[ctail] * LOG2E), s, rdy[ctail]);
    }
    }
    __syncthreads();
    float* dst = dx + row0 * V;
    for (int i = tid; i < nvec; i += SMB_THREADS)
        LSMB_ST(reinterpret_cast<float4*>(dst) + i, reinterpret_cast<const float4*>(tdy)[i]);
    for (int e = (nvec << 2) + tid; e < nel; e += SMB_THREADS) dst[e] = tdy[e];
}

template <int LG_THREADS, int LG_MAXVEC>
__global__ void __launch_bounds__(LG_THREADS)
k_lsmbwd_large(const float* dy, const float* y, float* dx, int64_t rows, int V, int xcd) {
    __shared__ float red[LG_THREADS / WAVE];
    const int nvec = V >> 2;
    const size_t per_xcd = ((size_t)rows + 7) / 8;       // (xcd: as k_lsm_large)
    const size_t items = xcd ? per_xcd * 8 : (size_t)rows;
    for (size_t it = blockIdx.x; it < items; it += gridDim.x) {
        const size_t row = xcd ? (it & 7) * per_xcd + (it >> 3) : it;
        if (row >= (size_t)rows) continue;
        const float4* sdy = reinterpret_cast<const float4*>(dy + row * V);
        const float4* sy = reinterpret_cast<const float4*>(y + row * V);
        float4 g[LG_MAXVEC];
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < LG_MAXVEC; ++i) {
            const int j = threadIdx.x + i * LG_THREADS;
            if (j < nvec) { g[i] = LSMB_LD(sdy + j); s += (g[i].x + g[i].y) + (g[i].z + g[i].w); }
        }
        s = block_reduce<LG_THREADS>(s, false, red);
        float4* dst = reinterpret_cast<float4*>(dx + row * V);
#pragma unroll
        for (int i = 0; i < LG_MAXVEC; ++i) {
            const int j = threadIdx.x + i * LG_THREADS;
            if (j < nvec) {
                const float4 p = LSMB_LD(sy + j);
                LSMB_ST(dst + j, make_float4(__builtin_fmaf(-__builtin_amdgcn_exp2f(p.x * LOG2E), s, g[i].x),
                                             __builtin_fmaf(-__builtin_amdgcn_exp2f(p.y * LOG2E), s, g[i].y),
                                             __builtin_fmaf(-__builtin_amdgcn_exp2f(p.z * LOG2E), s, g[i].z),
                                             __builtin_fmaf(-__builtin_amdgcn_exp2f(p.w * LOG2E), s, g[i].w)));
            }
        }
    }
}

__global__ void __launch_bounds__(256)
k_lsmbwd_generic(const float* dy, const float* y, float* dx, int64_t rows, int V) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* g = dy + row * V;
    const float* p = y + row * V;
    float s = 0.0f;
    for (int c = lane; c < V; c += WAVE) s += g[c];
    s = group_sum<WAVE>(s);
    float* o = dx + row * V;
    for (int c = lane; c < V; c += WAVE) o[c] = g[c] - expf(p[c]) * s;
}

hipError_t launch_log_softmax_backward(hipStream_t stream, const float* dy, const float* y, float* dx,
                                       int64_t rows, int V) {
    if (rows <= 0) return hipSuccess;
    const bool aligned = reinterpret_cast<uintptr_t>(dy) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0 &&
                         reinterpret_cast<uintptr_t>(dx) % 16 == 0;
    if (aligned && V <= 1024) {
        int L = 1;
        while (L < 64 && L * 16 < V) L <<= 1;
        const int q = (V + L - 1) / L;
        int R = (SMB_FLOATS / V) / 4 * 4;
        if (R < 4) R = 4;
        const size_t lds = (size_t)R * V * sizeof(float) * 2;
        const unsigned grid = stream_grid<XCD_LSMBWD_SMALL>((unsigned)((rows + R - 1) / R));
#define LSMB_SMALL(LL) case LL: k_lsmbwd_small<LL><<<grid, SMB_THREADS, lds, stream>>>(dy, y, dx, rows, V, R, q); break;
        switch (L) { LSMB_SMALL(1) LSMB_SMALL(2) LSMB_SMALL(4) LSMB_SMALL(8) LSMB_SMALL(16) LSMB_SMALL(32) LSMB_SMALL(64) }
#undef LSMB_SMALL
    } else if (aligned && V % 4 == 0 && V <= LG_MAXV) {
        // every XCD streams a contiguous eighth of the rows (as the forward kernel, dispatch_lsm): the reference's call
        // chain with the native log-softmax function at c3 2.08 / 2.07 / 2.03 -> 2.03 / 2.03 / 1.99 ms per training step;
        // RNNT_LSMBWD_XCD=0: the plain order (A/B runs)
        static const int bxcd = ab_getenv("RNNT_LSMBWD_XCD") ? atoi(ab_getenv("RNNT_LSMBWD_XCD")) : 1;
        unsigned grid = (unsigned)(rows < (1 << 22) ? rows : (1 << 22));
        if (bxcd) grid = (grid + 7u) & ~7u;
        static const bool old_rule = ab_getenv("RNNT_LSMBWD_SMALLEST_COVER") != nullptr;    // (A/B knob)
        if (old_rule) {
            if (V <= 4096) k_lsmbwd_large<256, 4><<<grid, 256, 0, stream>>>(dy, y, dx, rows, V, bxcd);
            else if (V <= 8192) k_lsmbwd_large<256, 8><<<grid, 256, 0, stream>>>(dy, y, dx, rows, V, bxcd);
            else k_lsmbwd_large<512, 8><<<grid, 512, 0, stream>>>(dy, y, dx, rows, V, bxcd);
        } else {      // as the forward kernel (dispatch_lsm): two or three passes, (nearly) every thread busy
            const int nvec = V >> 2;
            const int passes = nvec <= 2048 ? 2 : 3;
            int th = (nvec + 128 * passes - 1) / (128 * passes) * 128;
            th = th < 256 ? 256 : th;
#define LGB(TH, NV) case TH: k_lsmbwd_large<TH, NV><<<grid, TH, 0, stream>>>(dy, y, dx, rows, V, bxcd); break;
            if (nvec > 3072) {
                k_lsmbwd_large<512, 8><<<grid, 512, 0, stream>>>(dy, y, dx, rows, V, bxcd);
            } else if (passes == 2) {
                switch (th) { LGB(256, 2) LGB(384, 2) LGB(512, 2) LGB(640, 2) LGB(768, 2) LGB(896, 2) LGB(1024, 2) }
            } else {
                switch (th) { LGB(768, 3) LGB(896, 3) LGB(1024, 3) }
            }
#undef LGB
        }
    } else {
        k_lsmbwd_generic<<<(unsigned)((rows + 3) / 4), 256, 0, stream>>>(dy, y, dx, rows, V);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// To-diagonal kernels: dense log-probs (gather) or row-major pairs (re-layout) ->
// diagonal-major (blank,label) pairs.  A workgroup owns a 32x32 (t,u) tile of one
// utterance: it reads the tile with lanes along u (the contiguous axis of the
// source), parks it in LDS, and writes it back diagonal by diagonal, so that
// every store instruction covers contiguous runs of up to 32 pairs (256 B) of a
// diagonal-major row.  (One thread per cell writing its pair directly costs 4.6x
// write amplification: rocprofv3 WRITE_SIZE 268 MB for a 57.6 MB output.)
// ---------------------------------------------------------------------------
constexpr int TD = 32;   // tile edge
// Tiles are walked from the END of the tensor to its beginning: in the caller's step the kernel in front of this one
// is the log-softmax that has just written these log-probs front to back, and the last ~130-190 MB of what a
// streaming kernel wrote are still in the 256 MB Infinity Cache (tools/ubench/mall_probe.hip: 128 MB read back from
// the end of a freshly written 1.44 GB tensor in 22 us, from its beginning in 33-45 us).  Worth 8 us of the c4 step
// (0.4237 -> 0.4157 ms for the loss entry inside bench.py, three interleaved runs each); nothing on a tensor that
// was not just written (252 vs 249 us alone) -- which is how round 1 measured it and found no change.
#ifndef RNNT_GATHER_REVERSE
#define RNNT_GATHER_REVERSE 1
#endif

#ifndef RNNT_GATHER_TT
#define RNNT_GATHER_TT 32
#endif
// The dense gather's pair stores are written THROUGH (sc1) -- round 6: what its 58 MB of stores cost is dirty lines on their
// way out of L2 holding back the fills of the read stream; written through, nothing is left dirty: the kernel
// alone 252.0 -> 247.6 us (tools/ubench/gather_r06.hip), inside bench.py's c4 step 0.2215 -> 0.2182 ms (four interleaved
// pairs, profiles/r06_gather_sc1_ab.txt).  sc0 sc1 the same, nt / sc0 sc1 nt worse.  -DRNNT_GATHER_STORE_SC1=0: plain stores.
#ifndef RNNT_GATHER_STORE_SC1
#define RNNT_GATHER_STORE_SC1 1
#endif
constexpr int TT = RNNT_GATHER_TT;   // frames per tile of k_to_diagonal (columns: TD)

// The preparation of the ring kernel that follows in the same call (kernels.h: RingPrep; lattice_wd.hip: k_prepare is the
// stand-alone form), carried out at the tail of a producer's workgroups: every workgroup zeroes its slice of the rings,
// the first one clears the flags and the queue head and takes the next value of the device's launch counter.  Everything
// is complete when the producer's launch is, i.e. before the ring kernel starts.
__device__ __forceinline__ void fold_ring_prepare(const RingPrep& p) {
    const size_t per = (p.ring_vec + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per;
    const size_t hi = lo + per < p.ring_vec ? lo + per : p.ring_vec;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) p.rings[i] = z;
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < p.n_flags; i += blockDim.x) p.flags[i] = 0;
        if (threadIdx.x == 0) p.flags[p.n_flags] = (int)(atomicAdd(p.counter, 1u) + 1u);
    }
}

template <bool DENSE, int TTK>
__global__ void __launch_bounds__(256)
k_to_diagonal(const float* __restrict__ src, const int* __restrict__ labels, float2* __restrict__ ws2,
              int T, int U, int V, int blank, int tiles_t, int tiles_u, const RingPrep prep) {
    __shared__ float2 tile[TTK][TD];
    unsigned b = (DENSE && RNNT_GATHER_REVERSE) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int tu = b % tiles_u; b /= tiles_u;
    const int tt = b % tiles_t;
    const int n = b / tiles_t;
    const int t0 = tt * TTK, u0 = tu * TD;
    const int ul = threadIdx.x & (TD - 1), tl0 = threadIdx.x >> 5;   // 8 rows of 32 lanes
    const int u = u0 + ul;
    const size_t nbase = (size_t)n * T * U;
    int lab = blank;
    if (DENSE && u < U - 1) lab = safe_label(labels[(size_t)n * (U - 1) + u], V, blank);
#pragma unroll
    for (int k = 0; k < TTK / 8; ++k) {
        const int tl = tl0 + 8 * k, t = t0 + tl;
        if (t < T && u < U) {
            const size_t cell = nbase + (size_t)t * U + u;
            if constexpr (DENSE) {
                // Two dwords of a 4V-byte row: the memory system fetches whole 128-byte lines (a one-dword-per-line
                // probe over the same tensor takes as long as reading all of it, tools/ubench/gather_variants.hip),
                // so this kernel streams ~1.4 lines per cell and none of them is touched again: non-temporal loads
                // (206 vs 227 us for the probe, 205-229 vs 227-255 us here, box to box).
                const float* p = src + cell * (size_t)V;
                tile[tl][ul] = make_float2(__builtin_nontemporal_load(p + blank), __builtin_nontemporal_load(p + lab));
            } else {
                tile[tl][ul] = reinterpret_cast<const float2*>(src)[cell];
            }
        }
    }
    __syncthreads();
    // diagonal d of the tile holds cells (tl = d - ul, ul): consecutive ul = consecutive pairs of
    // row (t0+u0+d) mod T of the diagonal-major plane
#pragma unroll
    for (int k = 0; k < (TTK + TD + 7) / 8; ++k) {
        const int d = tl0 + 8 * k;
        const int tl = d - ul;
        if (d < TTK + TD - 1 && tl >= 0 && tl < TTK) {
            const int t = t0 + tl;
            if (t < T && u < U) {
                int r = t + u;
                r = r >= T ? r % T : r;
#if RNNT_GATHER_STORE_SC1
                // written THROUGH (agent scope): nothing is left dirty in L2 for the read stream's fills to wait behind
                // (tools/ubench/gather_r06.hip: 247.6 vs 252.0 us alone)
                if constexpr (DENSE) {
                    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(ws2 + nbase + (size_t)r * U + u), "v"(tile[tl][ul]) : "memory");
                } else
#endif
                ws2[nbase + (size_t)r * U + u] = tile[tl][ul];
            }
        }
    }
    if (prep.flags) fold_ring_prepare(prep);
}

// Row-major (N,T,U,2) gather (what the reference's wrapper builds): one thread per cell.
__global__ void __launch_bounds__(256)
k_gather_rowmajor(const float* __restrict__ lp, const int* __restrict__ labels, float2* __restrict__ out2,
                  size_t cells, int T, int U, int V, int blank) {
    const size_t cell = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (cell >= cells) return;
    const CellMap m = map_cell(cell, labels, T, U, V, blank);
    const float* p = lp + cell * (size_t)V;
    out2[cell] = make_float2(p[blank], p[m.label]);
}

template <int TTK>
static hipError_t launch_to_diagonal_tt(hipStream_t stream, const float* src, const int* labels, float* ws2,
                                        int N, int T, int U, int V, int blank, bool dense, const RingPrep& prep) {
    const int tiles_t = (T + TTK - 1) / TTK, tiles_u = (U + TD - 1) / TD;
    const size_t nblk = (size_t)N * tiles_t * tiles_u;
    if (nblk >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    if (dense)
        k_to_diagonal<true, TTK><<<(unsigned)nblk, 256, 0, stream>>>(src, labels, reinterpret_cast<float2*>(ws2), T, U,
                                                                     V, blank, tiles_t, tiles_u, prep);
    else
        k_to_diagonal<false, TTK><<<(unsigned)nblk, 256, 0, stream>>>(src, labels, reinterpret_cast<float2*>(ws2), T, U,
                                                                      2, 0, tiles_t, tiles_u, prep);
    return hipGetLastError();
}

static hipError_t launch_to_diagonal(hipStream_t stream, const float* src, const int* labels, float* ws2,
                                     int N, int T, int U, int V, int blank, bool dense, const RingPrep* prep_in) {
    const RingPrep prep = prep_in ? *prep_in : RingPrep{nullptr, 0, nullptr, nullptr, 0};
    if ((size_t)N * T * U == 0) return hipSuccess;
    // Small problems: 32-frame tiles do not even give every CU one workgroup (c2: 94 tiles for 256 CUs); 8-frame tiles --
    // one cell per thread -- quadruple the workgroups (RNNT_GATHER_SMALL_TILES=0 / 1 forces one or the other, for A/B runs)
    static const int force = ab_getenv("RNNT_GATHER_SMALL_TILES") ? atoi(ab_getenv("RNNT_GATHER_SMALL_TILES")) : -1;
    const size_t tiles32 = (size_t)N * ((T + TT - 1) / TT) * ((U + TD - 1) / TD);
    // (dense entry, us per call, 32- / 8-frame tiles: c2 28.1 / 27.4, N=32 35.1 / 33.2, N=64 46.2 / 45.2, N=128 67.1 / 68.1)
    const bool small_tiles = force >= 0 ? force != 0 : tiles32 < 512;
    if (small_tiles) return launch_to_diagonal_tt<8>(stream, src, labels, ws2, N, T, U, V, blank, dense, prep);
    return launch_to_diagonal_tt<TT>(stream, src, labels, ws2, N, T, U, V, blank, dense, prep);
}

// (Round 5 tried the dense gather as a coalesced STREAM for V <= 64, where the two dwords per row touch nearly every
//  128-byte line anyway: the LDS-staged log-softmax kernel without its arithmetic, pairs picked out of the staged tile.
//  Bit-identical, and slower: c4 274 us against 250 for k_to_diagonal in the same runs, loss path 0.403 vs 0.384 ms
//  -- a stream pays for all 1.44 GB, the scattered requests for the ~0.9 of the lines they touch.)
hipError_t launch_gather(hipStream_t stream, const float* log_probs, const int* labels, float* out2,
                         int N, int T, int U, int V, int blank, bool skewed, const RingPrep* prep) {
    const size_t cells = (size_t)N * T * U;
    if (cells == 0) return hipSuccess;
    if (skewed) return launch_to_diagonal(stream, log_probs, labels, out2, N, T, U, V, blank, true, prep);
    k_gather_rowmajor<<<(unsigned)((cells + 255) / 256), 256, 0, stream>>>(
        log_probs, labels, reinterpret_cast<float2*>(out2), cells, T, U, V, blank);
    return hipGetLastError();
}

hipError_t launch_reskew(hipStream_t stream, const float* lp2_rowmajor, float* ws2, int N, int T, int U,
                         const RingPrep* prep) {
    return launch_to_diagonal(stream, lp2_rowmajor, nullptr, ws2, N, T, U, 2, 0, false, prep);
}

// The way back: diagonal-major pairs -> row-major (N,T,U,2), the same 32x32 tiles walked the other way round (read by
// diagonals: consecutive lanes = consecutive pairs of a diagonal-major row; written by frames).  SPLIT: the two
// channels come from two float planes (the reference-named C entry points park the gradient pairs in the caller's
// alphas / betas buffers while the (N,T,U,2) output is still their staging area, api.hip).
template <bool SPLIT>
__global__ void __launch_bounds__(256)
k_from_diagonal(const float* __restrict__ a, const float* __restrict__ b, float2* __restrict__ out2, int T, int U,
                int tiles_t, int tiles_u) {
    __shared__ float2 tile[TT][TD + 1];
    unsigned blk = blockIdx.x;
    const int tu = blk % tiles_u; blk /= tiles_u;
    const int tt = blk % tiles_t;
    const int n = blk / tiles_t;
    const int t0 = tt * TT, u0 = tu * TD;
    const int ul = threadIdx.x & (TD - 1), tl0 = threadIdx.x >> 5;
    const int u = u0 + ul;
    const size_t nbase = (size_t)n * T * U;
    // LOADS FIRST (round 5: under their conditions the eight diagonals of a thread were eight memory round trips, one
    // after the other).  Every diagonal is loaded, at tile and lattice coordinates clamped into range -- a clamped slot
    // receives the value of the cell it stands for, so the LDS writes need no condition either.
    constexpr int ND = (TT + TD + 7) / 8;
    float2 pr[ND];
    int tls[ND];
    const int uc = min(u, U - 1);
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        tls[k] = min(max(tl0 + 8 * k - ul, 0), TT - 1);
        const int t = min(t0 + tls[k], T - 1);
        int r = t + uc;
        r = r >= T ? r % T : r;
        const size_t at = nbase + (size_t)r * U + uc;
        pr[k] = SPLIT ? make_float2(a[at], b[at]) : reinterpret_cast<const float2*>(a)[at];
    }
#pragma unroll
    for (int k = 0; k < ND; ++k) tile[tls[k]][ul] = pr[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TT / 8; ++k) {
        const int tl = tl0 + 8 * k, t = t0 + tl;
        if (t < T && u < U) out2[nbase + (size_t)t * U + u] = tile[tl][ul];
    }
}

hipError_t launch_unskew(hipStream_t stream, const float* a, const float* b, float* out2_rowmajor, int N, int T, int U) {
    if ((size_t)N * T * U == 0) return hipSuccess;
    const int tiles_t = (T + TT - 1) / TT, tiles_u = (U + TD - 1) / TD;
    const size_t nblk = (size_t)N * tiles_t * tiles_u;
    if (nblk >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    if (b)
        k_from_diagonal<true><<<(unsigned)nblk, 256, 0, stream>>>(a, b, reinterpret_cast<float2*>(out2_rowmajor), T, U,
                                                                  tiles_t, tiles_u);
    else
        k_from_diagonal<false><<<(unsigned)nblk, 256, 0, stream>>>(a, nullptr, reinterpret_cast<float2*>(out2_rowmajor),
                                                                   T, U, tiles_t, tiles_u);
    return hipGetLastError();
}

// (blank, label) pairs -> two planes, same cell order (a plain stream: 8 bytes in, 2 x 4 out per cell)
__global__ void __launch_bounds__(256)
k_split_pairs(const float4* __restrict__ src, float2* __restrict__ a, float2* __restrict__ b, size_t n2, const float2* tail,
              float* ta, float* tb) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n2) {
        const float4 v = src[i];                  // two cells
        a[i] = make_float2(v.x, v.z);
        b[i] = make_float2(v.y, v.w);
    } else if (i == n2 && tail) {                 // an odd last cell
        const float2 v = *tail;
        *ta = v.x; *tb = v.y;
    }
}

hipError_t launch_split_pairs(hipStream_t stream, const float* pairs, float* a, float* b, size_t cells) {
    if (cells == 0) return hipSuccess;
    const size_t n2 = cells / 2;
    const bool odd = cells & 1;
    const bool al = (reinterpret_cast<uintptr_t>(pairs) % 16 == 0) && (reinterpret_cast<uintptr_t>(a) % 8 == 0) &&
                    (reinterpret_cast<uintptr_t>(b) % 8 == 0);
    if (!al) return hipErrorInvalidValue;
    k_split_pairs<<<(unsigned)((n2 + 1 + 255) / 256), 256, 0, stream>>>(
        reinterpret_cast<const float4*>(pairs), reinterpret_cast<float2*>(a), reinterpret_cast<float2*>(b), n2,
        odd ? reinterpret_cast<const float2*>(pairs) + (cells - 1) : nullptr, a + (cells - 1), b + (cells - 1));
    return hipGetLastError();
}


// ---------------------------------------------------------------------------
// Compact (ragged packed) layout, reference: core_compact.cu:403-436 (gather) and
// :456-484 (scatter backward).  log-probs are (STU, V) rows, utterance n owning
// rows [offs[n], offs[n+1]) as a (T_n, U_n) row-major block; labels are packed (sum yn,).
// The gather produces the diagonal-major pairs of each utterance's own (T_n,U_n) plane
// (same 32x32 tile scheme as k_to_diagonal) plus `loc`, the vocabulary index the label
// channel was taken from (blank on the last column), which the backward scatter needs.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_gather_compact(const float* __restrict__ xs, const int* __restrict__ ys, const int* __restrict__ xn,
                 const int* __restrict__ yn, const int64_t* __restrict__ offs,
                 const int* __restrict__ label_offs, float2* __restrict__ ws2, int64_t* __restrict__ loc,
                 int V, int blank, int tiles_t, int tiles_u, int N, int plain_order) {
    __shared__ float2 tile[TD][TD];
    // The grid covers the batch maxima, so a ragged batch has tiles that lie outside their utterance and return at once.
    // Workgroups go to the eight XCDs by blockIdx mod 8: with the tile column as the fastest index (tiles_u = 4 at U = 100)
    // two XCDs would get nothing but last-column tiles, nearly all of them dead.  The utterance is the fastest index
    // instead, skewed by the tile so that no XCD keeps the same utterances (N=32, T=500, U=100, V=128, lengths 50-100 %:
    // whole calls: N=64, T=500, U=100 173 -> 150 us, N=32, T=1000, U=100 194 -> 180, nothing at N=32, T=500, U=100; profiles/r04_compact_gather_order_ab.txt).
    unsigned b = blockIdx.x;
    int n, tt, tu;
    if (plain_order) {
        tu = b % tiles_u; b /= tiles_u;
        tt = b % tiles_t;
        n = b / tiles_t;
    } else {
        const unsigned rest = b / (unsigned)N;
        n = (int)((b % (unsigned)N + rest) % (unsigned)N);
        tu = rest % tiles_u;
        tt = rest / tiles_u;
    }
    const int T = xn[n], U = yn[n] + 1;
    const int t0 = tt * TD, u0 = tu * TD;
    if (t0 >= T || u0 >= U) return;                    // whole tile outside this utterance (uniform)
    const int ul = threadIdx.x & (TD - 1), tl0 = threadIdx.x >> 5;
    const int u = u0 + ul;
    const size_t nbase = (size_t)offs[n];
    int lab = blank;
    if (u < U - 1) lab = safe_label(ys[label_offs[n] + u], V, blank);
#pragma unroll
    for (int k = 0; k < TD / 8; ++k) {
        const int tl = tl0 + 8 * k, t = t0 + tl;
        if (t < T && u < U) {
            const size_t cell = nbase + (size_t)t * U + u;
            const float* p = xs + cell * (size_t)V;      // (non-temporal: see k_to_diagonal)
            tile[tl][ul] = make_float2(__builtin_nontemporal_load(p + blank), __builtin_nontemporal_load(p + lab));
            if (loc) loc[cell] = lab;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (2 * TD) / 8; ++k) {
        const int d = tl0 + 8 * k;
        const int tl = d - ul;
        if (d < 2 * TD - 1 && tl >= 0 && tl < TD) {
            const int t = t0 + tl;
            if (t < T && u < U) {
                int r = t + u;
                r = r >= T ? r % T : r;
                ws2[nbase + (size_t)r * U + u] = tile[tl][ul];
            }
        }
    }
}

// The same gather over the LIVE cells in their packed order: workgroup b takes cells [1024 b, 1024 b + 1024) of the
// (STU, V) tensor whatever utterances they belong to -- no tile outside its utterance, no partly filled tile, every
// workgroup the same amount of work (the tiled kernel above reaches 0.44-0.53 of the line rate on ragged batches of
// 0.1-0.4 GB because its grid covers the batch maxima: profiles/r04_shape_map.md).  Every lane reads its two dwords from
// a row of its own either way, so nothing is lost on the read side; the pairs leave as scattered 8-byte stores (the
// tiles write runs of up to 256 bytes), loc as one coalesced stream.
#ifndef RNNT_GCL_CELLS
#define RNNT_GCL_CELLS 4
#endif
constexpr int GCL_CELLS = RNNT_GCL_CELLS;       // cells per thread
__global__ void __launch_bounds__(256)
k_gather_compact_linear(const float* __restrict__ xs, const int* __restrict__ ys, const int* __restrict__ xn,
                        const int* __restrict__ yn, const int64_t* __restrict__ offs,
                        const int* __restrict__ label_offs, float2* __restrict__ ws2, int64_t* __restrict__ loc,
                        int V, int blank, int N, int64_t STU) {
    __shared__ int s_n0, s_n1;
    const int tid = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * (256 * GCL_CELLS);
    // The utterances the chunk's first and last cell belong to: owner(c) = the first n with offs[n + 1] > c, by binary
    // search (two lanes, <= 17 steps over an array that sits in L2; the first version had every workgroup scan all N + 1
    // offsets).  Utterances without cells are never an owner; a refused batch -- every checked length 0 -- is dropped
    // cell by cell below.  offs must be non-decreasing for the result to mean anything; for a malformed array the search
    // still ends, on one well-defined utterance, and a cell outside that utterance's range is nobody's.
    if (tid < 2) {
        const int64_t c = tid == 0 ? c0 : min(c0 + 256 * GCL_CELLS, STU) - 1;
        int lo = 0, hi = N;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (offs[mid + 1] > c) hi = mid; else lo = mid + 1;
        }
        if (tid == 0) s_n0 = lo; else s_n1 = lo;
    }
    __syncthreads();
    const int n0 = s_n0, n1 = min(s_n1, N - 1);
    if (n0 >= N) return;
    float2 pair[GCL_CELLS];
    size_t dst[GCL_CELLS];
    int labs[GCL_CELLS];
    bool live[GCL_CELLS];
#pragma unroll
    for (int k = 0; k < GCL_CELLS; ++k) {
        const int64_t c = c0 + tid + 256 * k;
        live[k] = false;
        pair[k] = make_float2(0.0f, 0.0f);
        dst[k] = 0;
        labs[k] = blank;
        if (c >= STU) continue;
        int n = n0, hi = n1;                           // owner(c) within [n0, n1]: usually no step, or one
        while (n < hi) {
            const int mid = (n + hi) >> 1;
            if (offs[mid + 1] > c) hi = mid; else n = mid + 1;
        }
        if (c >= offs[n + 1] || c < offs[n]) continue;  // (malformed offsets: nobody's cell)
        const int T = xn[n], U = yn[n] + 1;
        if (T < 1 || U < 1) continue;
        const unsigned local = (unsigned)(c - offs[n]);
        const unsigned t = local / (unsigned)U;
        const int u = (int)(local - t * (unsigned)U);
        if ((int)t >= T) continue;
        if (u < U - 1) labs[k] = safe_label(ys[label_offs[n] + u], V, blank);
        const float* p = xs + (size_t)c * (size_t)V;
        pair[k] = make_float2(__builtin_nontemporal_load(p + blank), __builtin_nontemporal_load(p + labs[k]));
        int r = (int)t + u;
        r = r >= T ? r % T : r;
        dst[k] = (size_t)offs[n] + (size_t)r * U + u;
        live[k] = true;
    }
#pragma unroll
    for (int k = 0; k < GCL_CELLS; ++k) {
        if (!live[k]) continue;
        ws2[dst[k]] = pair[k];
        if (loc) loc[c0 + tid + 256 * k] = labs[k];
    }
}

hipError_t launch_gather_compact(hipStream_t stream, const float* xs, const int* ys, const int* xn,
                                 const int* yn, const int64_t* offs, const int* label_offs, float* ws2,
                                 int64_t* loc, int N, int Tmax, int Umax, int V, int blank, int64_t STU) {
    if (N <= 0 || Tmax <= 0 || Umax <= 0) return hipSuccess;
    // RNNT_COMPACT_GATHER=tiles|linear pins one of the two kernels (A/B runs)
    static const char* pin = ab_getenv("RNNT_COMPACT_GATHER");
    const bool want_linear = pin ? pin[0] == 'l' : true;
    if (want_linear && STU > 0 && N <= 4096) {
        const int64_t nblk = (STU + 256 * GCL_CELLS - 1) / (256 * GCL_CELLS);
        if (nblk < ((int64_t)1 << 31)) {
            k_gather_compact_linear<<<(unsigned)nblk, 256, 0, stream>>>(xs, ys, xn, yn, offs, label_offs,
                                                                        reinterpret_cast<float2*>(ws2), loc, V, blank,
                                                                        N, STU);
            return hipGetLastError();
        }
    }
    const int tiles_t = (Tmax + TD - 1) / TD, tiles_u = (Umax + TD - 1) / TD;
    const size_t nblk = (size_t)N * tiles_t * tiles_u;
    if (nblk >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    static const bool plain_order = ab_getenv("RNNT_COMPACT_PLAIN_TILE_ORDER") != nullptr;      // A/B runs
    k_gather_compact<<<(unsigned)nblk, 256, 0, stream>>>(xs, ys, xn, yn, offs, label_offs,
                                                         reinterpret_cast<float2*>(ws2), loc, V, blank,
                                                         tiles_t, tiles_u, N, plain_order ? 1 : 0);
    return hipGetLastError();
}

// The reference's own compact gather (core_compact.cu:403-450, run_gather_for_compact): ROW-MAJOR packed pairs
// (STU,2) and loc (STU,), 32-bit exclusive offsets.  One thread per cell; serves the core.h shims of api.hip.
__global__ void __launch_bounds__(256)
k_gather_compact_rowmajor(const float* __restrict__ xs, const int* __restrict__ ys, const unsigned* __restrict__ xn,
                          const unsigned* __restrict__ yn, float2* __restrict__ out2, int64_t* __restrict__ loc,
                          const unsigned* __restrict__ mem_pref, const unsigned* __restrict__ label_pref, unsigned V,
                          unsigned blank) {
    const unsigned n = blockIdx.y;
    const unsigned Tn = xn[n], Un = yn[n] + 1;
    if ((int)Tn < 1 || (int)Un < 1) return;
    const unsigned c = blockIdx.x * 256u + threadIdx.x;
    if (c >= Tn * Un) return;
    const unsigned u = c % Un;
    const size_t index = (size_t)mem_pref[n] + c;
    const int l = (u == Un - 1) ? (int)blank : safe_label(ys[label_pref[n] + u], (int)V, (int)blank);
    const float* p = xs + index * (size_t)V;
    out2[index] = make_float2(p[blank], p[l]);
    loc[index] = l;
}

hipError_t launch_gather_compact_rowmajor(hipStream_t stream, const float* xs, const int* ys, const unsigned* xn,
                                          const unsigned* yn, float* gather_xs, int64_t* loc, const unsigned* mem_pref,
                                          const unsigned* label_pref, unsigned N, unsigned T, unsigned U, unsigned V,
                                          unsigned blank) {
    if (N == 0 || T == 0 || U == 0) return hipSuccess;
    const unsigned long long tiles = ((unsigned long long)T * U + 255ull) / 256ull;
    if (tiles >= (1ull << 31) || N > 65535u) return hipErrorInvalidValue;
    k_gather_compact_rowmajor<<<dim3((unsigned)tiles, N), 256, 0, stream>>>(
        xs, ys, xn, yn, reinterpret_cast<float2*>(gather_xs), loc, mem_pref, label_pref, V, blank);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Layout turns of the staged form of run_warp_rnnt_compact (api.hip): the reference's compact tensors are ROW-MAJOR
// packed pairs with 32-bit exclusive prefixes (no total: the last utterance's size closes the batch), the tuned kernels
// want each utterance's own diagonal-major plane.  Same 32x32 LDS tiles as k_to_diagonal / k_from_diagonal, one grid over
// (utterance, tile) with the utterance as the fastest index (tiles outside their utterance return at once).
//   TO_DIAGONAL: src = row-major pairs -> dst = diagonal-major pairs
//   otherwise:   (sa, sb) = the two channels as diagonal-major float planes -> dst = row-major pairs
// ---------------------------------------------------------------------------------------------------------
template <bool TO_DIAGONAL>
__global__ void __launch_bounds__(256)
k_turn_compact32(const float2* __restrict__ src, const float* __restrict__ sa, const float* __restrict__ sb,
                 float2* __restrict__ dst, const unsigned* __restrict__ xn, const unsigned* __restrict__ yn,
                 const unsigned* __restrict__ mem_pref, int tiles_u, unsigned N) {
    __shared__ float2 tile[TD][TD + 1];
    const unsigned rest = blockIdx.x / N;
    const unsigned n = (blockIdx.x % N + rest) % N;
    const int tu = rest % tiles_u, tt = rest / tiles_u;
    const int T = (int)xn[n], U = (int)yn[n] + 1;
    const int t0 = tt * TD, u0 = tu * TD;
    if (T < 1 || U < 1 || t0 >= T || u0 >= U) return;          // (uniform)
    const int ul = threadIdx.x & (TD - 1), tl0 = threadIdx.x >> 5;
    const int u = u0 + ul;
    const size_t nbase = (size_t)mem_pref[n];
    // one pass by frames (lanes along u, row-major side), one by diagonals (diagonal-major side)
    auto by_frames = [&](auto&& f) {
#pragma unroll
        for (int k = 0; k < TD / 8; ++k) {
            const int tl = tl0 + 8 * k, t = t0 + tl;
            if (t < T && u < U) f(tl, nbase + (size_t)t * U + u);
        }
    };
    auto by_diagonals = [&](auto&& f) {
#pragma unroll
        for (int k = 0; k < (2 * TD) / 8; ++k) {
            const int d = tl0 + 8 * k, tl = d - ul;
            if (d < 2 * TD - 1 && tl >= 0 && tl < TD) {
                const int t = t0 + tl;
                if (t < T && u < U) {
                    int r = t + u;
                    r = r >= T ? r % T : r;
                    f(tl, nbase + (size_t)r * U + u);
                }
            }
        }
    };
    // The LOAD side of either direction is written out loads-first (round 5; see k_to_diagonal / k_from_diagonal: all of
    // a thread's loads issued before the first LDS write, at coordinates clamped into the tile and the utterance -- a
    // clamped slot receives the value of the cell it stands for); the store side keeps its conditions.
    const int uc = min(u, U - 1);
    if constexpr (TO_DIAGONAL) {
        constexpr int NK = TD / 8;
        float2 pr[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) pr[k] = src[nbase + (size_t)min(t0 + tl0 + 8 * k, T - 1) * U + uc];
#pragma unroll
        for (int k = 0; k < NK; ++k) tile[tl0 + 8 * k][ul] = pr[k];
        __syncthreads();
        by_diagonals([&](int tl, size_t at) { dst[at] = tile[tl][ul]; });
    } else {
        constexpr int ND = (2 * TD) / 8;
        float2 pr[ND];
        int tls[ND];
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            tls[k] = min(max(tl0 + 8 * k - ul, 0), TD - 1);
            const int t = min(t0 + tls[k], T - 1);
            int r = t + uc;
            r = r >= T ? r % T : r;
            const size_t at = nbase + (size_t)r * U + uc;
            pr[k] = make_float2(sa[at], sb[at]);
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) tile[tls[k]][ul] = pr[k];
        __syncthreads();
        by_frames([&](int tl, size_t at) { dst[at] = tile[tl][ul]; });
    }
}

static hipError_t launch_turn_compact32(hipStream_t stream, bool to_diagonal, const float* src, const float* sa,
                                        const float* sb, float* dst, const unsigned* xn, const unsigned* yn,
                                        const unsigned* mem_pref, unsigned N, unsigned Tmax, unsigned Umax) {
    if (N == 0 || Tmax == 0 || Umax == 0) return hipSuccess;
    const unsigned tiles_t = (Tmax + TD - 1) / TD, tiles_u = (Umax + TD - 1) / TD;
    const size_t nblk = (size_t)N * tiles_t * tiles_u;
    if (nblk >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    if (to_diagonal)
        k_turn_compact32<true><<<(unsigned)nblk, 256, 0, stream>>>(reinterpret_cast<const float2*>(src), nullptr, nullptr,
                                                                   reinterpret_cast<float2*>(dst), xn, yn, mem_pref,
                                                                   (int)tiles_u, N);
    else
        k_turn_compact32<false><<<(unsigned)nblk, 256, 0, stream>>>(nullptr, sa, sb, reinterpret_cast<float2*>(dst), xn,
                                                                    yn, mem_pref, (int)tiles_u, N);
    return hipGetLastError();
}

hipError_t launch_reskew_compact32(hipStream_t stream, const float* pairs_rowmajor, float* pairs_diagonal,
                                   const unsigned* xn, const unsigned* yn, const unsigned* mem_pref, unsigned N,
                                   unsigned Tmax, unsigned Umax) {
    return launch_turn_compact32(stream, true, pairs_rowmajor, nullptr, nullptr, pairs_diagonal, xn, yn, mem_pref, N, Tmax,
                                 Umax);
}

hipError_t launch_unskew_compact32(hipStream_t stream, const float* a_diagonal, const float* b_diagonal,
                                   float* pairs_rowmajor, const unsigned* xn, const unsigned* yn, const unsigned* mem_pref,
                                   unsigned N, unsigned Tmax, unsigned Umax) {
    return launch_turn_compact32(stream, false, nullptr, a_diagonal, b_diagonal, pairs_rowmajor, xn, yn, mem_pref, N, Tmax,
                                 Umax);
}

// (blank, label) pairs -> two planes over the cells of a compact batch whose total only the device knows (the last
// prefix + the last utterance's size): the grid covers the bound, threads beyond the total return.
__global__ void __launch_bounds__(256)
k_split_pairs_compact32(const float2* __restrict__ src, float* __restrict__ a, float* __restrict__ b,
                        const unsigned* __restrict__ xn, const unsigned* __restrict__ yn,
                        const unsigned* __restrict__ mem_pref, unsigned N) {
    const int tl = (int)xn[N - 1], ul = (int)yn[N - 1] + 1;
    const size_t total = (size_t)mem_pref[N - 1] + ((tl >= 1 && ul >= 1) ? (size_t)tl * ul : 0);
    // two cells per thread where the pointers allow 16-byte loads (the caller checks), the odd last cell alone
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < total) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        *reinterpret_cast<float2*>(a + i) = make_float2(v.x, v.z);
        *reinterpret_cast<float2*>(b + i) = make_float2(v.y, v.w);
    } else if (i < total) {
        const float2 v = src[i];
        a[i] = v.x; b[i] = v.y;
    }
}

hipError_t launch_split_pairs_compact32(hipStream_t stream, const float* pairs, float* a, float* b, const unsigned* xn,
                                        const unsigned* yn, const unsigned* mem_pref, unsigned N, size_t cells_bound) {
    if (N == 0 || cells_bound == 0) return hipSuccess;
    const bool al = (reinterpret_cast<uintptr_t>(pairs) % 16 == 0) && (reinterpret_cast<uintptr_t>(a) % 8 == 0) &&
                    (reinterpret_cast<uintptr_t>(b) % 8 == 0);
    if (!al) return hipErrorInvalidValue;
    const size_t nblk = (cells_bound / 2 + 1 + 255) / 256;
    if (nblk >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    k_split_pairs_compact32<<<(unsigned)nblk, 256, 0, stream>>>(reinterpret_cast<const float2*>(pairs), a, b, xn, yn,
                                                                mem_pref, N);
    return hipGetLastError();
}

// Prefix sums and launch bounds of a compact batch in ONE launch (the reference's binding does this
// with a chain of torch ops and four host synchronisations, binding.cpp:139-170):
//   cell_offsets[0..N] = exclusive sums of xn*(yn+1) (int64), label_offsets[0..N] = exclusive sums of yn,
//   stats = {sum cells, sum labels, max xn, max yn}.  One workgroup; N <= 65535.
constexpr int CP_THREADS = 1024;
__global__ void __launch_bounds__(CP_THREADS)
k_compact_offsets(const int* __restrict__ xn, const int* __restrict__ yn, int N, int64_t* __restrict__ cell_offs,
                  int* __restrict__ label_offs, int64_t* __restrict__ stats, const CompactBounds bounds) {
    __shared__ int64_t s_cells[CP_THREADS];
    __shared__ int s_labs[CP_THREADS];
    __shared__ int s_tmax[CP_THREADS / WAVE], s_umax[CP_THREADS / WAVE];
    const int tid = threadIdx.x;
    const int per = (N + CP_THREADS - 1) / CP_THREADS;
    const int lo = min(tid * per, N), hi = min(lo + per, N);
    int64_t c = 0;
    int l = 0, tmax = INT_MIN, umax = INT_MIN;
    for (int n = lo; n < hi; ++n) {
        const int x = xn[n], y = yn[n];
        c += (int64_t)x * (y + 1);
        l += y;
        tmax = max(tmax, x);
        umax = max(umax, y);
    }
    s_cells[tid] = c;
    s_labs[tid] = l;
    for (int o = 32; o > 0; o >>= 1) {
        tmax = max(tmax, __shfl_xor(tmax, o));
        umax = max(umax, __shfl_xor(umax, o));
    }
    if ((tid & (WAVE - 1)) == 0) { s_tmax[tid >> 6] = tmax; s_umax[tid >> 6] = umax; }
    __syncthreads();
    // inclusive scan over the 1024 per-thread totals: inside every wave on shuffles, then the sixteen wave totals by the
    // first wave -- two barriers (the first version's Hillis-Steele over shared memory took twenty)
    __shared__ int64_t s_wc[CP_THREADS / WAVE];
    __shared__ int s_wl[CP_THREADS / WAVE];
    const int lane = tid & (WAVE - 1), wv = tid >> 6;
    int64_t ic = c;
    int il = l;
    for (int o = 1; o < WAVE; o <<= 1) {
        const int64_t vc = __shfl_up(ic, o);
        const int vl = __shfl_up(il, o);
        if (lane >= o) { ic += vc; il += vl; }
    }
    if (lane == WAVE - 1) { s_wc[wv] = ic; s_wl[wv] = il; }
    __syncthreads();
    if (wv == 0) {
        int64_t wc = lane < CP_THREADS / WAVE ? s_wc[lane] : 0;
        int wl = lane < CP_THREADS / WAVE ? s_wl[lane] : 0;
        for (int o = 1; o < CP_THREADS / WAVE; o <<= 1) {
            const int64_t vc = __shfl_up(wc, o);
            const int vl = __shfl_up(wl, o);
            if (lane >= o) { wc += vc; wl += vl; }
        }
        if (lane < CP_THREADS / WAVE) { s_wc[lane] = wc; s_wl[lane] = wl; }   // inclusive totals of waves 0 ... lane
    }
    __syncthreads();
    if (wv > 0) { ic += s_wc[wv - 1]; il += s_wl[wv - 1]; }
    s_cells[tid] = ic;
    s_labs[tid] = il;
    __syncthreads();
    int64_t cbase = ic - c;     // exclusive
    int lbase = il - l;
    for (int n = lo; n < hi; ++n) {
        cell_offs[n] = cbase;
        label_offs[n] = lbase;
        cbase += (int64_t)xn[n] * (yn[n] + 1);
        lbase += yn[n];
    }
    if (tid == CP_THREADS - 1) {
        cell_offs[N] = s_cells[tid];
        label_offs[N] = s_labs[tid];
        int tm = s_tmax[0], um = s_umax[0];
        for (int i = 1; i < CP_THREADS / WAVE; ++i) { tm = max(tm, s_tmax[i]); um = max(um, s_umax[i]); }
        stats[0] = s_cells[tid];
        stats[1] = s_labs[tid];
        stats[2] = tm;
        stats[3] = um;
    }
    if (bounds.xn_checked) {
        // Caller-supplied launch bounds (no read-back of the maxima): what the host would have checked after its
        // synchronisation is checked here.  One length out of range, or totals that are not the tensors' sizes, make
        // every offset meaningless, so the whole batch is refused: lengths of 0 frames go to the kernels, which report
        // cost = NaN and touch nothing.
        __shared__ int s_bad;
        if (tid == 0) s_bad = 0;
        __syncthreads();
        bool bad = false;
        for (int n = lo; n < hi; ++n) {
            const int x = xn[n], y = yn[n];
            bad |= x < 1 || y < 0 || x > bounds.Tmax || y + 1 > bounds.Umax;
        }
        if (tid == CP_THREADS - 1) bad |= s_cells[tid] != bounds.STU || (int64_t)s_labs[tid] != bounds.n_labels;
        if (bad) atomicOr(&s_bad, 1);
        __syncthreads();
        const bool refuse = s_bad != 0;
        for (int n = lo; n < hi; ++n) bounds.xn_checked[n] = refuse ? 0 : xn[n];
        if (tid == 0) stats[4] = refuse ? 1 : 0;
    }
}

// behind a bounded compact call: a refused batch owns no cells the kernels could have written, so its (STU,2)
// gradients are zeroed here; returns at once otherwise (one flag read per thread)
__global__ void __launch_bounds__(256)
k_zero_if_refused(const int64_t* __restrict__ refused, float2* __restrict__ g2, size_t cells) {
    if (*refused == 0) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (size_t)gridDim.x * 256) g2[i] = make_float2(0.f, 0.f);
}

hipError_t launch_zero_if_refused(hipStream_t stream, const int64_t* refused, float* grads2, size_t cells) {
    if (!grads2 || cells == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>(1024, (cells + 255) / 256);
    k_zero_if_refused<<<blocks, 256, 0, stream>>>(refused, reinterpret_cast<float2*>(grads2), cells);
    return hipGetLastError();
}

hipError_t launch_compact_offsets(hipStream_t stream, const int* xn, const int* yn, int N, int64_t* cell_offs,
                                  int* label_offs, int64_t* stats, const CompactBounds* bounds) {
    const CompactBounds none{nullptr, 0, 0, 0, 0};
    k_compact_offsets<<<1, CP_THREADS, 0, stream>>>(xn, yn, N, cell_offs, label_offs, stats, bounds ? *bounds : none);
    return hipGetLastError();
}

}  // namespace rnnt
