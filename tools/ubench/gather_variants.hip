// Micro-benchmark for the gather prologue (dense log-probs (N,T,U,V) -> diagonal-major (blank,label) pairs,
// reference: pytorch_binding/warp_rnnt/__init__.py:118-128) at the c4 shape, round 3.
//
// Part 1 (probes): what granularity does a sparse dword read fetch from HBM, and how many lines per second can
//   the chip touch?  One dword per STRIDE bytes over the whole tensor, 1/2/4/8 independent loads per thread.
// Part 2 (variants): the shipped tile kernel (two sparse dword loads per cell), the same with non-temporal loads,
//   and a row-streaming form (a wave reads the contiguous 64*V*4-byte segment of 64 lattice columns of one frame
//   -- through registers or straight into LDS with global_load_lds_dwordx4 -- and picks the pairs from LDS).
//   Every variant's output is compared with the shipped kernel's.
// hipcc --offload-arch=gfx950 -O3 gather_variants.hip -o gather_variants && ./gather_variants [N T U V]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- probes
template <int UN, bool NT>
__global__ void __launch_bounds__(256) k_probe(const float* __restrict__ a, float* __restrict__ sink, size_t nline,
                                               int stride_dw) {
    const size_t base = ((size_t)blockIdx.x * 256 * UN) + threadIdx.x;
    float v[UN];
#pragma unroll
    for (int j = 0; j < UN; ++j) {
        const size_t i = base + (size_t)j * 256;
        v[j] = 0.f;
        if (i < nline) v[j] = NT ? __builtin_nontemporal_load(a + i * stride_dw) : a[i * stride_dw];
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < UN; ++j) s += v[j];
    if (s == 123.456f) sink[0] = s;
}

// ---------------------------------------------------------------- shipped kernel (prologue.hip k_to_diagonal<true>)
template <int TT, bool NT>
__global__ void __launch_bounds__(256)
k_tile(const float* __restrict__ src, const int* __restrict__ labels, float2* __restrict__ ws2, int T, int U, int V,
       int blank, int tiles_t, int tiles_u) {
    constexpr int TD = 32;
    __shared__ float2 tile[TT][TD];
    unsigned b = blockIdx.x;
    const int tu = b % tiles_u; b /= tiles_u;
    const int tt = b % tiles_t;
    const int n = b / tiles_t;
    const int t0 = tt * TT, u0 = tu * TD;
    const int ul = threadIdx.x & (TD - 1), tl0 = threadIdx.x >> 5;
    const int u = u0 + ul;
    const size_t nbase = (size_t)n * T * U;
    int lab = blank;
    if (u < U - 1) lab = labels[(size_t)n * (U - 1) + u];
#pragma unroll
    for (int k = 0; k < TT / 8; ++k) {
        const int tl = tl0 + 8 * k, t = t0 + tl;
        if (t < T && u < U) {
            const float* p = src + (nbase + (size_t)t * U + u) * (size_t)V;
            if (NT) tile[tl][ul] = make_float2(__builtin_nontemporal_load(p + blank), __builtin_nontemporal_load(p + lab));
            else tile[tl][ul] = make_float2(p[blank], p[lab]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (TT + TD + 7) / 8; ++k) {
        const int d = tl0 + 8 * k;
        const int tl = d - ul;
        if (d < TT + TD - 1 && tl >= 0 && tl < TT) {
            const int t = t0 + tl;
            if (t < T && u < U) {
                int r = t + u;
                r = r >= T ? r % T : r;
                ws2[nbase + (size_t)r * U + u] = tile[tl][ul];
            }
        }
    }
}


// ---------------------------------------------------------------- generalised tile kernel
// THREADS per workgroup (TD = 32 columns, THREADS/32 frame rows per pass), TT frames per tile, AUX = cache policy
// bits of the loads (gfx940+: 1 = sc0, 2 = nt, 16 = sc1), NTS = non-temporal stores of the pairs.
template <int THREADS, int TT, int AUX, bool NTS, int TD = 32>
__global__ void __launch_bounds__(THREADS)
k_tile2(const float* __restrict__ src, const int* __restrict__ labels, float2* __restrict__ ws2, int T, int U, int V,
        int blank, int tiles_t, int tiles_u, unsigned total_bytes) {
    constexpr int RP = THREADS / TD;
    __shared__ float2 tile[TT][TD];
    unsigned b = blockIdx.x;
    const int tu = b % tiles_u; b /= tiles_u;
    const int tt = b % tiles_t;
    const int n = b / tiles_t;
    const int t0 = tt * TT, u0 = tu * TD;
    const int ul = threadIdx.x % TD, tl0 = threadIdx.x / TD;
    const int u = u0 + ul;
    const size_t nbase = (size_t)n * T * U;
    int lab = blank;
    if (u < U - 1) lab = labels[(size_t)n * (U - 1) + u];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)total_bytes, 0x00020000);
    float2 v[TT / RP];
#pragma unroll
    for (int k = 0; k < TT / RP; ++k) {
        const int tl = tl0 + RP * k, t = t0 + tl;
        const bool ok = t < T && u < U;
        const unsigned row = (unsigned)((nbase + (size_t)t * U + u) * (size_t)V * 4);
        const int o0 = ok ? (int)(row + blank * 4) : (int)0x80000000, o1 = ok ? (int)(row + lab * 4) : (int)0x80000000;
        v[k].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o0, 0, AUX));
        v[k].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o1, 0, AUX));
    }
#pragma unroll
    for (int k = 0; k < TT / RP; ++k) tile[tl0 + RP * k][ul] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (TT + TD + RP - 1) / RP; ++k) {
        const int d = tl0 + RP * k;
        const int tl = d - ul;
        if (d < TT + TD - 1 && tl >= 0 && tl < TT) {
            const int t = t0 + tl;
            if (t < T && u < U) {
                int r = t + u;
                r = r >= T ? r % T : r;
                float2* dst = ws2 + nbase + (size_t)r * U + u;
                typedef float f2v __attribute__((ext_vector_type(2)));
                const float2 pr = tile[tl][ul];
                if (NTS) { f2v q; q.x = pr.x; q.y = pr.y; __builtin_nontemporal_store(q, reinterpret_cast<f2v*>(dst)); } else *dst = pr;
            }
        }
    }
}


// ---------------------------------------------------------------- what holds the sparse form back? (timing probes)
// MODE 1: blank loads only, 2: label loads only, 3: lanes 0-31 load the blanks and lanes 32-63 the labels of the
// SAME 32 rows in one instruction (a line that serves two requests is then asked for by one instruction),
// 4: all blank loads of the tile first, the label loads behind a vmcnt(0), 5: as 0 without the stores.
template <int TT, int MODE, int AUX>
__global__ void __launch_bounds__(256)
k_tile3(const float* __restrict__ src, const int* __restrict__ labels, float2* __restrict__ ws2, int T, int U, int V,
        int blank, int tiles_t, int tiles_u, unsigned total_bytes) {
    constexpr int TD = 32;
    __shared__ float tile[TT][TD][2];
    unsigned b = blockIdx.x;
    const int tu = b % tiles_u; b /= tiles_u;
    const int tt = b % tiles_t;
    const int n = b / tiles_t;
    const int t0 = tt * TT, u0 = tu * TD;
    const int ul = threadIdx.x & (TD - 1);
    const int u = u0 + ul;
    const size_t nbase = (size_t)n * T * U;
    int lab = blank;
    if (u < U - 1) lab = labels[(size_t)n * (U - 1) + u];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)total_bytes, 0x00020000);
    if constexpr (MODE == 3) {
        const int which = (threadIdx.x >> 5) & 1, tl0 = threadIdx.x >> 6;     // 4 frame rows per pass
        const int col = which ? lab : blank;
        float v[TT / 4];
#pragma unroll
        for (int k = 0; k < TT / 4; ++k) {
            const int t = t0 + tl0 + 4 * k;
            const bool ok = t < T && u < U;
            const unsigned row = (unsigned)((nbase + (size_t)t * U + u) * (size_t)V * 4);
            v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (int)(row + col * 4) : (int)0x80000000, 0, AUX));
        }
#pragma unroll
        for (int k = 0; k < TT / 4; ++k) tile[tl0 + 4 * k][ul][which] = v[k];
    } else {
        const int tl0 = threadIdx.x >> 5;
        float2 v[TT / 8];
        int o0[TT / 8], o1[TT / 8];
#pragma unroll
        for (int k = 0; k < TT / 8; ++k) {
            const int t = t0 + tl0 + 8 * k;
            const bool ok = t < T && u < U;
            const unsigned row = (unsigned)((nbase + (size_t)t * U + u) * (size_t)V * 4);
            o0[k] = ok ? (int)(row + blank * 4) : (int)0x80000000;
            o1[k] = ok ? (int)(row + lab * 4) : (int)0x80000000;
            v[k] = make_float2(0.f, 0.f);
        }
        if constexpr (MODE == 4) {
#pragma unroll
            for (int k = 0; k < TT / 8; ++k) v[k].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o0[k], 0, AUX));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < TT / 8; ++k) v[k].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o1[k], 0, AUX));
        } else {
#pragma unroll
            for (int k = 0; k < TT / 8; ++k) {
                if (MODE != 2) v[k].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o0[k], 0, AUX));
                if (MODE != 1) v[k].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o1[k], 0, AUX));
            }
        }
#pragma unroll
        for (int k = 0; k < TT / 8; ++k) { tile[tl0 + 8 * k][ul][0] = v[k].x; tile[tl0 + 8 * k][ul][1] = v[k].y; }
    }
    __syncthreads();
    const int tl0 = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < (TT + TD + 7) / 8; ++k) {
        const int d = tl0 + 8 * k;
        const int tl = d - ul;
        if (d < TT + TD - 1 && tl >= 0 && tl < TT) {
            const int t = t0 + tl;
            if (t < T && u < U) {
                int r = t + u;
                r = r >= T ? r % T : r;
                const float2 pr = make_float2(tile[tl][ul][0], tile[tl][ul][1]);
                if (MODE != 5 || pr.x == 123.456f) ws2[nbase + (size_t)r * U + u] = pr;
            }
        }
    }
}

// ---------------------------------------------------------------- row streaming
// Workgroup = 4 waves, tile = TT frames x 64 columns of one utterance.  Wave w takes frames w, w+4, ...: the 64
// rows (t, u0..u0+63) are one contiguous segment of 64*V*4 bytes; the wave copies it into its LDS staging area
// (16-byte aligned start: the few bytes in front of the segment come along) and lane ul picks blank and label of
// row ul from it.  MODE 0: through registers; 1: global_load_lds_dwordx4; 2: the same, non-temporal (aux = 2).
template <int TT, int MODE>
__global__ void __launch_bounds__(256)
k_stream(const float* __restrict__ src, const int* __restrict__ labels, float2* __restrict__ ws2, int T, int U, int V,
         int blank, int tiles_t, int tiles_u, int stage_bytes, size_t total_bytes) {
    constexpr int TD = 64;
    __shared__ float2 tile[TT][TD];
    extern __shared__ __attribute__((aligned(16))) char stage_all[];
    unsigned b = blockIdx.x;
    const int tu = b % tiles_u; b /= tiles_u;
    const int tt = b % tiles_t;
    const int n = b / tiles_t;
    const int t0 = tt * TT, u0 = tu * TD;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int u = u0 + lane;
    const size_t nbase = (size_t)n * T * U;
    int lab = blank;
    if (u < U - 1) lab = labels[(size_t)n * (U - 1) + u];
    char* stage = stage_all + (size_t)w * stage_bytes;
    const int ncol = min(TD, U - u0);
    const int seg_bytes = ncol * V * 4;
    for (int tl = w; tl < TT; tl += 4) {
        const int t = t0 + tl;
        if (t >= T) break;
        const size_t seg0 = (nbase + (size_t)t * U + u0) * (size_t)V * 4;      // byte offset of the segment
        const size_t al0 = seg0 & ~(size_t)15;
        const int delta = (int)(seg0 - al0);
        const int nvec = (delta + seg_bytes + 15) >> 4;
        const char* g = reinterpret_cast<const char*>(src) + al0;
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < nvec; i += 64) {
            if (al0 + (size_t)i * 16 + 16 <= total_bytes + 15) {     // (the allocation is padded by 16 bytes)
                if constexpr (MODE == 0) {
                    *reinterpret_cast<f4*>(stage + i * 16) = *reinterpret_cast<const f4*>(g + (size_t)i * 16);
                } else {
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(g + (size_t)i * 16),
                        (__attribute__((address_space(3))) void*)(stage + (i - lane) * 16), 16, 0,
                        MODE == 2 ? 2 : 0);
                }
            }
        }
        if constexpr (MODE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane < ncol) {
            const float* row = reinterpret_cast<const float*>(stage + delta) + lane * V;
            tile[tl][lane] = make_float2(row[blank], row[lab]);
        }
    }
    __syncthreads();
    // diagonal d of the tile: cells (tl = d - ul, ul)
    for (int d = w; d < TT + TD - 1; d += 4) {
        const int tl = d - lane;
        if (tl >= 0 && tl < TT) {
            const int t = t0 + tl;
            if (t < T && u < U) {
                int r = t + u;
                r = r >= T ? r % T : r;
                ws2[nbase + (size_t)r * U + u] = tile[tl][lane];
            }
        }
    }
}

template <typename F>
static float run(const char* name, F launch, double gb_useful) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int r = 0; r < 12; ++r) {
        hipEventRecord(e0);
        for (int i = 0; i < 4; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) ts.push_back(ms / 4);
    }
    CHECK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    const float med = ts[ts.size() / 2];
    printf("%-58s median %7.1f us  min %7.1f us   %.2f TB/s of %.3f GB\n", name, med * 1e3, ts[0] * 1e3,
           gb_useful / (med * 1e-3) / 1e3, gb_useful);
    fflush(stdout);
    return med;
}

int main(int argc, char** argv) {
    const int N = argc > 4 ? atoi(argv[1]) : 16, T = argc > 4 ? atoi(argv[2]) : 1500, U = argc > 4 ? atoi(argv[3]) : 300,
              V = argc > 4 ? atoi(argv[4]) : 50;
    const size_t cells = (size_t)N * T * U, bytes = cells * V * 4;
    float *src, *sink;
    float2 *ref, *out;
    int* labels;
    CHECK(hipMalloc(&src, bytes + 16)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&ref, cells * 8)); CHECK(hipMalloc(&out, cells * 8));
    CHECK(hipMalloc(&labels, (size_t)N * (U - 1) * 4 + 4));
    {
        std::vector<float> h(cells * V);
        unsigned s = 12345;
        for (size_t i = 0; i < h.size(); ++i) { s = s * 1664525u + 1013904223u; h[i] = -(float)(s >> 8) * (1.0f / 16777216.0f) * 8.f; }
        CHECK(hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice));
        std::vector<int> l((size_t)N * (U - 1));
        for (size_t i = 0; i < l.size(); ++i) { s = s * 1664525u + 1013904223u; l[i] = 1 + (int)((s >> 10) % (unsigned)(V - 1)); }
        CHECK(hipMemcpy(labels, l.data(), l.size() * 4, hipMemcpyHostToDevice));
    }
    printf("N=%d T=%d U=%d V=%d: dense %.3f GB, pairs %.4f GB\n", N, T, U, V, bytes / 1e9, cells * 8 / 1e9);

    // ---- probes (PROBES_ONLY=1: nothing else -- the rocprofv3 --pmc FETCH_SIZE calibration pass of
    //      tools/collect_profiles.sh: how many bytes does the counter tally per touched line?)
    const bool probes_only = getenv("PROBES_ONLY") != nullptr;
    for (int stride : {64, 128, 256, 512}) {
        const size_t nline = bytes / stride;
        char name[128];
#define PROBE(UN, NT)                                                                                          \
    snprintf(name, sizeof name, "probe: 1 dword per %3d B, %d loads/thread%s (%.2f M lines)", stride, UN,       \
             NT ? ", nt" : "", nline / 1e6);                                                                    \
    run(name, [&] { k_probe<UN, NT><<<(unsigned)((nline + 256 * UN - 1) / (256 * UN)), 256>>>(src, sink, nline, stride / 4); }, \
        nline * (double)stride / 1e9);
        PROBE(1, false) PROBE(4, false) PROBE(8, false) PROBE(4, true)
#undef PROBE
    }

    if (probes_only) return 0;
    // ---- variants
    const double useful = cells * 16 / 1e9;
    auto check = [&](const char* name) {
        std::vector<float2> a(cells), b(cells);
        CHECK(hipMemcpy(a.data(), ref, cells * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(b.data(), out, cells * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < cells; ++i) bad += (a[i].x != b[i].x) || (a[i].y != b[i].y);
        printf("    %s: %zu mismatching cells of %zu\n", name, bad, cells);
        CHECK(hipMemset(out, 0xff, cells * 8));
    };
    {
        const int tiles_u = (U + 31) / 32;
#define TILE(TT, NT, dst, label)                                                                                  \
    {                                                                                                             \
        const int tiles_t = (T + TT - 1) / TT;                                                                    \
        run(label, [&] { k_tile<TT, NT><<<(unsigned)((size_t)N * tiles_t * tiles_u), 256>>>(src, labels, dst, T, U, V, 0, tiles_t, tiles_u); }, useful); \
    }
        CHECK(hipMemset(ref, 0xff, cells * 8));
        CHECK(hipMemset(out, 0xff, cells * 8));
        TILE(32, false, ref, "tile 32x32, sparse dword loads (shipped)")
        TILE(32, true, out, "tile 32x32, sparse, non-temporal loads") check("nt");
        TILE(16, false, out, "tile 16x32, sparse") check("tt16");
        TILE(64, false, out, "tile 64x32, sparse") check("tt64");
#undef TILE
    }
    {
        const int tiles_u = (U + 63) / 64;
        const int stage_bytes = ((64 * V * 4 + 16 + 15) / 16) * 16;
#define STREAM(TT, MODE, label)                                                                                   \
    {                                                                                                             \
        const int tiles_t = (T + TT - 1) / TT;                                                                    \
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stream<TT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * stage_bytes)); \
        run(label, [&] { k_stream<TT, MODE><<<(unsigned)((size_t)N * tiles_t * tiles_u), 256, 4 * stage_bytes>>>(src, labels, out, T, U, V, 0, tiles_t, tiles_u, stage_bytes, bytes); }, useful); \
        check(label);                                                                                             \
    }
        if (4 * stage_bytes + 64 * 64 * 8 <= 160 * 1024) {
            STREAM(32, 0, "stream 32x64, through registers")
            STREAM(32, 1, "stream 32x64, global_load_lds")
            STREAM(32, 2, "stream 32x64, global_load_lds nt")
            STREAM(16, 1, "stream 16x64, global_load_lds")
            STREAM(16, 2, "stream 16x64, global_load_lds nt")
            STREAM(64, 2, "stream 64x64, global_load_lds nt")
        }
#undef STREAM
    }
    {
        const int tiles_u = (U + 31) / 32;
#define TILE2(TH, TT, AUX, NTS, label)                                                                            \
    {                                                                                                             \
        const int tiles_t = (T + TT - 1) / TT;                                                                    \
        run(label, [&] { k_tile2<TH, TT, AUX, NTS><<<(unsigned)((size_t)N * tiles_t * tiles_u), TH>>>(src, labels, out, T, U, V, 0, tiles_t, tiles_u, (unsigned)bytes); }, useful); \
        check(label);                                                                                             \
    }
        if (bytes < ((size_t)1 << 32)) {
            TILE2(256, 32, 0, false, "tile2 256thr 32x32 aux=0")
            TILE2(256, 32, 2, false, "tile2 256thr 32x32 nt")
            TILE2(256, 32, 16, false, "tile2 256thr 32x32 sc1")
            TILE2(256, 32, 17, false, "tile2 256thr 32x32 sc0 sc1")
            TILE2(256, 32, 18, false, "tile2 256thr 32x32 sc1 nt")
            TILE2(256, 32, 2, true, "tile2 256thr 32x32 nt, nt stores")
            TILE2(256, 16, 2, false, "tile2 256thr 16x32 nt")
            TILE2(256, 8, 2, false, "tile2 256thr 8x32 nt")
            TILE2(128, 32, 2, false, "tile2 128thr 32x32 nt")
            TILE2(128, 16, 2, false, "tile2 128thr 16x32 nt")
            TILE2(64, 32, 2, false, "tile2 64thr 32x32 nt")
            TILE2(64, 16, 2, false, "tile2 64thr 16x32 nt")
            TILE2(512, 32, 2, false, "tile2 512thr 32x32 nt")
            TILE2(512, 64, 2, false, "tile2 512thr 64x32 nt")
        }
#undef TILE2
#define TILE2W(TH, TT, TD, label)                                                                                 \
    {                                                                                                             \
        const int tiles_t = (T + TT - 1) / TT, tiles_uw = (U + TD - 1) / TD;                                      \
        run(label, [&] { k_tile2<TH, TT, 2, false, TD><<<(unsigned)((size_t)N * tiles_t * tiles_uw), TH>>>(src, labels, out, T, U, V, 0, tiles_t, tiles_uw, (unsigned)bytes); }, useful); \
        check(label);                                                                                             \
    }
        if (bytes < ((size_t)1 << 32)) {
            TILE2W(256, 32, 32, "tile2 nt 32 frames x 32 columns (shipped shape)")
            TILE2W(256, 32, 64, "tile2 nt 32 frames x 64 columns")
            TILE2W(256, 16, 64, "tile2 nt 16 frames x 64 columns")
            TILE2W(256, 16, 128, "tile2 nt 16 frames x 128 columns")
            TILE2W(256, 8, 128, "tile2 nt 8 frames x 128 columns")
            TILE2W(256, 8, 256, "tile2 nt 8 frames x 256 columns")
            TILE2W(512, 16, 128, "tile2 nt 16 frames x 128 columns, 512 threads")
            TILE2W(256, 32, 32, "tile2 nt 32 frames x 32 columns (again)")
        }
#undef TILE2W
    }
    {
        const int tiles_u = (U + 31) / 32;
#define TILE3(TT, MODE, AUX, chk, label)                                                                          \
    {                                                                                                             \
        const int tiles_t = (T + TT - 1) / TT;                                                                    \
        run(label, [&] { k_tile3<TT, MODE, AUX><<<(unsigned)((size_t)N * tiles_t * tiles_u), 256>>>(src, labels, out, T, U, V, 0, tiles_t, tiles_u, (unsigned)bytes); }, useful); \
        if (chk) check(label);                                                                                    \
    }
        if (bytes < ((size_t)1 << 32)) {
            TILE3(32, 0, 0, true, "tile3 normal")
            TILE3(32, 1, 0, false, "tile3 blank loads only")
            TILE3(32, 2, 0, false, "tile3 label loads only")
            TILE3(32, 3, 0, true, "tile3 blank|label lanes of one instruction")
            TILE3(32, 3, 2, true, "tile3 blank|label lanes of one instruction, nt")
            TILE3(16, 3, 0, true, "tile3 16x32 blank|label lanes of one instruction")
            TILE3(32, 4, 0, true, "tile3 blanks first, labels behind vmcnt(0)")
            TILE3(32, 5, 0, false, "tile3 normal, no stores")
            TILE3(32, 0, 0, true, "tile3 normal (again)")
        }
#undef TILE3
    }
    return 0;
}
