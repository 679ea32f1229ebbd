// Round 6 micro-benchmark of the gather prologue (dense log-probs (N,T,U,V) -> diagonal-major (blank,label) pairs;
// reference: pytorch_binding/warp_rnnt/__init__.py:118-128) at the c4 shape.  Round 3's file (gather_variants.hip) settled
// the read side: HBM delivers whole 128-byte lines, the two dwords per row touch 10.0 M of the tensor's 11.25 M lines, and
// the kernel WITHOUT its stores runs at the rate of a one-dword-per-line probe (204 vs 207 us).  What is left is the
// 25-45 us the 57.6 MB of stores cost on top.  This file asks where they go:
//   hot      the shipped structure (LDS tile, barrier, diagonal runs) storing into a small buffer that stays in L2:
//            the structure without the DRAM writes
//   strip    tiles enumerated along the anti-diagonals of the tile grid, one strip per XCD at a time: the two tiles that
//            share the partial 128-byte lines of a diagonal-major row -- (tt, tu) and (tt-1, tu+1) -- run next to each
//            other on ONE L2, where their partial writes can merge before they leave for HBM
//   wide     64-column tiles (runs of 512 bytes: half the partial lines), 32 or 64 frames
//   x4       16-byte stores (two pairs per lane)
//   pipe     persistent workgroups, the next tile's loads issued before the current tile's stores
// Every variant's output is compared with the shipped kernel's.
// hipcc --offload-arch=gfx950 -O3 gather_r06.hip -o gather_r06 && ./gather_r06 [N T U V]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

struct TileId { int n, tt, tu; bool ok; };

// ORDER 0: linear (tu fastest), 1: linear reversed (the shipped walk), 2: strips per XCD, 3: strips per XCD, reversed
template <int ORDER>
__device__ __forceinline__ TileId tile_of(unsigned b, int N, int tiles_t, int tiles_u) {
    TileId id;
    if constexpr (ORDER < 2) {
        if (ORDER == 1) b = gridDim.x - 1 - b;
        id.tu = b % tiles_u; b /= tiles_u;
        id.tt = b % tiles_t;
        id.n = b / tiles_t;
        id.ok = id.n < N;
    } else {
        const unsigned xcd = b & 7u, j = b >> 3;
        const unsigned nstrips = tiles_t + tiles_u - 1, S = (unsigned)N * nstrips;
        unsigned g = (j / tiles_u) * 8u + xcd;
        id.tu = j % tiles_u;
        id.ok = g < S;
        if (ORDER == 3 && id.ok) g = S - 1 - g;
        id.n = g / nstrips;
        id.tt = (int)(g % nstrips) - id.tu;
        id.ok = id.ok && id.tt >= 0 && id.tt < tiles_t;
    }
    return id;
}
static unsigned grid_of(int order, int N, int tiles_t, int tiles_u) {
    if (order < 2) return (unsigned)((size_t)N * tiles_t * tiles_u);
    const unsigned S = (unsigned)N * (tiles_t + tiles_u - 1);
    return ((S + 7) / 8) * tiles_u * 8;
}

// The shipped kernel's structure, generalised: TT frames x TD columns per tile, 256 threads, non-temporal loads,
// HOT: stores go to a 64 KB region (same addresses for every tile), X4: 16-byte stores.
// PITCH: 0 = the plane's rows are U pairs apart (shipped); > 0: rows padded to a multiple of PITCH pairs (16 pairs = one
// 128-byte line: every diagonal run of a tile then starts on a line), planes T * Upad apart;
// -1: the tile is written as ONE contiguous chunk (tile-major: what a perfectly sequential full-line write costs)
template <int TT, int TD, int ORDER, bool HOT, bool X4, int PITCH = 0>
__global__ void __launch_bounds__(256)
k_tile(const float* __restrict__ src, const int* __restrict__ labels, float2* __restrict__ ws2, int N, int T, int U, int V,
       int blank, int tiles_t, int tiles_u) {
    constexpr int RP = 256 / TD;                      // frame rows per pass
    __shared__ float2 tile[TT][TD + (X4 ? 0 : 0)];
    const TileId id = tile_of<ORDER>(blockIdx.x, N, tiles_t, tiles_u);
    if (!id.ok) return;
    const int t0 = id.tt * TT, u0 = id.tu * TD;
    const int ul = threadIdx.x % TD, tl0 = threadIdx.x / TD;
    const int u = u0 + ul;
    const size_t nbase = (size_t)id.n * T * U;
    int lab = blank;
    if (u < U - 1) lab = labels[(size_t)id.n * (U - 1) + u];
    float2 v[TT / RP];
#pragma unroll
    for (int k = 0; k < TT / RP; ++k) {
        const int tl = tl0 + RP * k, t = t0 + tl;
        const bool ok = t < T && u < U;
        const float* p = src + (nbase + (size_t)(ok ? t : 0) * U + (ok ? u : 0)) * (size_t)V;
        v[k] = make_float2(__builtin_nontemporal_load(p + blank), __builtin_nontemporal_load(p + lab));
    }
#pragma unroll
    for (int k = 0; k < TT / RP; ++k) tile[tl0 + RP * k][ul] = v[k];
    __syncthreads();
    if constexpr (!X4) {
#pragma unroll
        for (int k = 0; k < (TT + TD + RP - 1) / RP; ++k) {
            const int d = tl0 + RP * k;
            const int tl = d - ul;
            if (d < TT + TD - 1 && tl >= 0 && tl < TT) {
                const int t = t0 + tl;
                if (t < T && u < U) {
                    int r = t + u;
                    r = r >= T ? r % T : r;
                    size_t at = nbase + (size_t)r * U + u;
                    if constexpr (PITCH > 0) {
                        const int Up = (U + PITCH - 1) / PITCH * PITCH;
                        at = (size_t)id.n * T * Up + (size_t)r * Up + u;
                    } else if constexpr (PITCH < 0) {
                        at = ((size_t)(id.n * tiles_t + id.tt) * tiles_u + id.tu) * (TT * TD) + (size_t)tl * TD + ul;
                    }
                    if (HOT) at &= 8191;
                    ws2[at] = tile[tl][ul];
                }
            }
        }
    } else {
        // two consecutive columns per lane: lane pair index q = ul / 2 ... a diagonal's run of TD pairs = TD / 2 lanes
        constexpr int LPD = TD / 2;                    // lanes per diagonal
        constexpr int DP = 256 / LPD;                  // diagonals per pass
        const int q = threadIdx.x % LPD, dl0 = threadIdx.x / LPD;
#pragma unroll
        for (int k = 0; k < (TT + TD + DP - 1) / DP; ++k) {
            const int d = dl0 + DP * k;
            if (d >= TT + TD - 1) continue;
            const int c0 = 2 * q, c1 = c0 + 1;
            const int tla = d - c0, tlb = d - c1;
            const bool oka = tla >= 0 && tla < TT && t0 + tla < T && u0 + c0 < U;
            const bool okb = tlb >= 0 && tlb < TT && t0 + tlb < T && u0 + c1 < U;
            if (!oka && !okb) continue;
            // row of the diagonal-major plane: (t + u) mod T is the same for both cells unless it wraps between them
            int ra = t0 + tla + u0 + c0; ra = ra >= T ? ra % T : ra;
            size_t at = nbase + (size_t)ra * U + u0 + c0;
            if (HOT) at &= 8190;
            const bool aligned16 = ((at & 1) == 0);
            if (oka && okb && aligned16) {
                const float2 a = tile[tla][c0], b2 = tile[tlb][c1];
                f4v o; o.x = a.x; o.y = a.y; o.z = b2.x; o.w = b2.y;
                *reinterpret_cast<f4v*>(ws2 + at) = o;
            } else {
                if (oka) ws2[at] = tile[tla][c0];
                if (okb) ws2[at + 1] = tile[tlb][c1];
            }
        }
    }
}

// The shipped structure with the pair stores as buffer stores carrying cache-policy bits (gfx940+: 1 = sc0, 2 = nt, 16 = sc1;
// sc0 sc1 = system scope: written through, nothing left dirty in L2) -- do dirty lines on their way out of L2 hold back the
// fills of the read stream (profiles/r06_gather_store_pmc.csv), and does writing THROUGH avoid it?
template <int TT, int TD, int ORDER, int AUX, int PITCH = 0>
__global__ void __launch_bounds__(256)
k_tile_aux(const float* __restrict__ src, const int* __restrict__ labels, float2* __restrict__ ws2, int N, int T, int U, int V,
           int blank, int tiles_t, int tiles_u, unsigned out_bytes) {
    constexpr int RP = 256 / TD;
    __shared__ float2 tile[TT][TD];
    const TileId id = tile_of<ORDER>(blockIdx.x, N, tiles_t, tiles_u);
    if (!id.ok) return;
    const int t0 = id.tt * TT, u0 = id.tu * TD;
    const int ul = threadIdx.x % TD, tl0 = threadIdx.x / TD;
    const int u = u0 + ul;
    const size_t nbase = (size_t)id.n * T * U;
    int lab = blank;
    if (u < U - 1) lab = labels[(size_t)id.n * (U - 1) + u];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ws2, 0, (int)out_bytes, 0x00020000);
    float2 v[TT / RP];
#pragma unroll
    for (int k = 0; k < TT / RP; ++k) {
        const int tl = tl0 + RP * k, t = t0 + tl;
        const bool ok = t < T && u < U;
        const float* p = src + (nbase + (size_t)(ok ? t : 0) * U + (ok ? u : 0)) * (size_t)V;
        v[k] = make_float2(__builtin_nontemporal_load(p + blank), __builtin_nontemporal_load(p + lab));
    }
#pragma unroll
    for (int k = 0; k < TT / RP; ++k) tile[tl0 + RP * k][ul] = v[k];
    __syncthreads();
    typedef int i2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k < (TT + TD + RP - 1) / RP; ++k) {
        const int d = tl0 + RP * k;
        const int tl = d - ul;
        if (d < TT + TD - 1 && tl >= 0 && tl < TT) {
            const int t = t0 + tl;
            if (t < T && u < U) {
                int r = t + u;
                r = r >= T ? r % T : r;
                const int Up = PITCH > 0 ? (U + PITCH - 1) / PITCH * PITCH : U;       // PITCH > 0: rows padded to PITCH pairs
                const size_t at = ((size_t)id.n * T + r) * Up + u;
                const float2 pr = tile[tl][ul];
                i2v q; q.x = __builtin_bit_cast(int, pr.x); q.y = __builtin_bit_cast(int, pr.y);
                __builtin_amdgcn_raw_buffer_store_b64(q, rs, (int)(at * 8), 0, AUX);
            }
        }
    }
}

// Persistent, pipelined: workgroup w walks tiles w, w + G, w + 2G, ... (ORDER as above over the virtual index); the loads
// of the next tile are in flight while the current tile is written out of LDS.
template <int TT, int TD, int ORDER>
__global__ void __launch_bounds__(256)
k_pipe(const float* __restrict__ src, const int* __restrict__ labels, float2* __restrict__ ws2, int N, int T, int U, int V,
       int blank, int tiles_t, int tiles_u, unsigned total) {
    constexpr int RP = 256 / TD;
    __shared__ float2 tile[TT][TD];
    const int ul = threadIdx.x % TD, tl0 = threadIdx.x / TD;
    float2 v[TT / RP];
    auto id_of = [&](unsigned b) {
        TileId id;
        if (ORDER == 1) b = total - 1 - b;
        id.tu = b % tiles_u; b /= tiles_u;
        id.tt = b % tiles_t;
        id.n = b / tiles_t;
        id.ok = true;
        return id;
    };
    auto load = [&](const TileId id) {
        const int t0 = id.tt * TT, u = id.tu * TD + ul;
        const size_t nbase = (size_t)id.n * T * U;
        int lab = blank;
        if (u < U - 1) lab = labels[(size_t)id.n * (U - 1) + u];
#pragma unroll
        for (int k = 0; k < TT / RP; ++k) {
            const int t = t0 + tl0 + RP * k;
            const bool ok = t < T && u < U;
            const float* p = src + (nbase + (size_t)(ok ? t : 0) * U + (ok ? u : 0)) * (size_t)V;
            v[k] = make_float2(__builtin_nontemporal_load(p + blank), __builtin_nontemporal_load(p + lab));
        }
    };
    unsigned b = blockIdx.x;
    if (b >= total) return;
    TileId cur = id_of(b);
    load(cur);
    for (;;) {
#pragma unroll
        for (int k = 0; k < TT / RP; ++k) tile[tl0 + RP * k][ul] = v[k];
        __syncthreads();
        const unsigned nb = b + gridDim.x;
        const bool more = nb < total;
        TileId nxt = cur;
        if (more) { nxt = id_of(nb); load(nxt); }          // in flight during the store phase below
        {
            const int t0 = cur.tt * TT, u0 = cur.tu * TD, u = u0 + ul;
            const size_t nbase = (size_t)cur.n * T * U;
#pragma unroll
            for (int k = 0; k < (TT + TD + RP - 1) / RP; ++k) {
                const int d = tl0 + RP * k;
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
        if (!more) break;
        __syncthreads();
        b = nb; cur = nxt;
    }
}

template <typename F>
static float run(const char* name, F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int r = 0; r < 14; ++r) {
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
    printf("%-66s median %7.1f us  min %7.1f us\n", name, med * 1e3, ts[0] * 1e3);
    fflush(stdout);
    return med;
}

int main(int argc, char** argv) {
    const int N = argc > 4 ? atoi(argv[1]) : 16, T = argc > 4 ? atoi(argv[2]) : 1500, U = argc > 4 ? atoi(argv[3]) : 300,
              V = argc > 4 ? atoi(argv[4]) : 50;
    const size_t cells = (size_t)N * T * U, bytes = cells * V * 4;
    float* src;
    float2 *ref, *out;
    int* labels;
    CHECK(hipMalloc(&src, bytes + 16));
    CHECK(hipMalloc(&ref, cells * 8)); CHECK(hipMalloc(&out, cells * 8 + cells * 8 / 4));
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
    auto check = [&](const char* name) {
        std::vector<float2> a(cells), b(cells);
        CHECK(hipMemcpy(a.data(), ref, cells * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(b.data(), out, cells * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < cells; ++i) bad += (a[i].x != b[i].x) || (a[i].y != b[i].y);
        if (bad) printf("    %s: %zu MISMATCHING cells of %zu\n", name, bad, cells);
        CHECK(hipMemset(out, 0xff, cells * 8));
    };
    CHECK(hipMemset(ref, 0xff, cells * 8));
    CHECK(hipMemset(out, 0xff, cells * 8));
#define TILE(TT, TD, ORDER, HOT, X4, dst, chk, label)                                                                  \
    {                                                                                                                 \
        const int tiles_t = (T + TT - 1) / TT, tiles_u = (U + TD - 1) / TD;                                           \
        const unsigned grid = grid_of(ORDER, N, tiles_t, tiles_u);                                                    \
        run(label, [&] { k_tile<TT, TD, ORDER, HOT, X4><<<grid, 256>>>(src, labels, dst, N, T, U, V, 0, tiles_t, tiles_u); }); \
        if (chk) check(label);                                                                                        \
    }
    if (getenv("GATHER_R06_PMC")) {
        // counter passes (rocprofv3 --pmc around this binary): three kernels, 12 launches each, nothing else --
        // the shipped form, the same with its stores kept in L2, and the tile-major form (sequential full-line writes)
        TILE(32, 32, 1, false, false, out, false, "pmc: 32x32 linear reversed (shipped)")
        TILE(32, 32, 0, true, false, out, false, "pmc: 32x32 linear, stores into a hot 64 KB buffer")
        {
            const int tiles_t = (T + 31) / 32, tiles_u = (U + 31) / 32;
            const unsigned grid = grid_of(1, N, tiles_t, tiles_u);
            run("pmc: 32x32 reversed, tile-major contiguous output",
                [&] { k_tile<32, 32, 1, false, false, -1><<<grid, 256>>>(src, labels, out, N, T, U, V, 0, tiles_t, tiles_u); });
        }
        return 0;
    }
    for (int round = 0; round < 2; ++round) {      // (twice: run-to-run drift on a box is a few us)
        TILE(32, 32, 0, false, false, ref, false, "32x32 linear (shipped shape, forward walk)")
        TILE(32, 32, 1, false, false, out, true, "32x32 linear reversed (shipped)")
        TILE(32, 32, 0, false, false, out, true, "32x32 linear forward, into the other buffer")
        TILE(32, 32, 1, false, false, ref, false, "32x32 linear reversed, into the first buffer")
        TILE(32, 32, 0, false, false, ref, false, "32x32 linear forward, into the first buffer again")
        TILE(32, 32, 0, true, false, out, false, "32x32 linear, stores into a hot 64 KB buffer (no DRAM writes)")
        TILE(32, 32, 2, false, false, out, true, "32x32 strips per XCD")
        TILE(32, 32, 3, false, false, out, true, "32x32 strips per XCD, reversed")
        TILE(32, 32, 0, false, true, out, true, "32x32 linear, 16-byte stores")
        TILE(32, 32, 2, false, true, out, true, "32x32 strips per XCD, 16-byte stores")
        TILE(32, 64, 0, false, false, out, true, "32 frames x 64 columns linear")
        TILE(32, 64, 2, false, false, out, true, "32 frames x 64 columns strips per XCD")
        TILE(64, 64, 0, false, false, out, true, "64 frames x 64 columns linear")
        TILE(64, 64, 2, false, false, out, true, "64 frames x 64 columns strips per XCD")
        TILE(64, 64, 2, false, true, out, true, "64 frames x 64 columns strips per XCD, 16-byte stores")
        TILE(16, 64, 0, false, false, out, true, "16 frames x 64 columns linear")
#define TILEA(AUX, label)                                                                                              \
    {                                                                                                                 \
        const int tiles_t = (T + 31) / 32, tiles_u = (U + 31) / 32;                                                   \
        const unsigned grid = grid_of(1, N, tiles_t, tiles_u);                                                        \
        run(label, [&] { k_tile_aux<32, 32, 1, AUX><<<grid, 256>>>(src, labels, out, N, T, U, V, 0, tiles_t, tiles_u, (unsigned)(cells * 8)); }); \
        check(label);                                                                                                 \
    }
        TILEA(0, "32x32 reversed, buffer stores, default policy")
        TILEA(17, "32x32 reversed, stores sc0 sc1 (system scope: written through)")
        TILEA(16, "32x32 reversed, stores sc1")
        TILEA(1, "32x32 reversed, stores sc0")
        TILEA(19, "32x32 reversed, stores sc0 sc1 nt")
        TILEA(2, "32x32 reversed, stores nt")
#define TILEAP(AUX, PITCH, label)                                                                                      \
    {                                                                                                                 \
        const int tiles_t = (T + 31) / 32, tiles_u = (U + 31) / 32;                                                   \
        const unsigned grid = grid_of(1, N, tiles_t, tiles_u);                                                        \
        run(label, [&] { k_tile_aux<32, 32, 1, AUX, PITCH><<<grid, 256>>>(src, labels, out, N, T, U, V, 0, tiles_t, tiles_u, (unsigned)(cells * 10)); }); \
        CHECK(hipMemset(out, 0xff, cells * 8));                                                                       \
    }
        // written through AND every run on the 64-byte grid (rows padded to 8 / 16 pairs): half of the shipped plane's rows
        // start 32 bytes off it (U = 300: 2400-byte rows)
        TILEAP(16, 8, "32x32 reversed, stores sc1, rows padded to 64 bytes (U 300 -> 304)")
        TILEAP(16, 16, "32x32 reversed, stores sc1, rows padded to 128 bytes (U 300 -> 304)")
        TILEAP(18, 8, "32x32 reversed, stores sc1 nt, rows padded to 64 bytes")
        TILEAP(0, 8, "32x32 reversed, default policy, rows padded to 64 bytes")
        TILEA(18, "32x32 reversed, stores sc1 nt")
#define TILEP(TT, TD, ORDER, PITCH, label)                                                                             \
    {                                                                                                                 \
        const int tiles_t = (T + TT - 1) / TT, tiles_u = (U + TD - 1) / TD;                                           \
        const unsigned grid = grid_of(ORDER, N, tiles_t, tiles_u);                                                    \
        run(label, [&] { k_tile<TT, TD, ORDER, false, false, PITCH><<<grid, 256>>>(src, labels, out, N, T, U, V, 0, tiles_t, tiles_u); }); \
        CHECK(hipMemset(out, 0xff, cells * 8));                                                                       \
    }
        TILEP(32, 32, 1, 16, "32x32 reversed, rows padded to whole 128-byte lines (U 300 -> 304)")
        TILEP(32, 32, 1, 32, "32x32 reversed, rows padded to 256 bytes (U 300 -> 320)")
        TILEP(32, 64, 1, 16, "32 frames x 64 columns reversed, rows padded to whole lines")
        TILEP(32, 32, 1, -1, "32x32 reversed, tile written as one contiguous 8 KB chunk (tile-major)")
        TILEP(32, 32, 0, 16, "32x32 forward, rows padded to whole 128-byte lines")
        TILEP(16, 32, 1, 16, "16x32 reversed, rows padded to whole lines")
        TILE(16, 128, 0, false, false, out, true, "16 frames x 128 columns linear")
        {
            const int tiles_t = (T + 31) / 32, tiles_u = (U + 31) / 32;
            const unsigned total = (unsigned)((size_t)N * tiles_t * tiles_u);
            for (int per_cu : {4, 6, 8}) {
                char name[96];
                snprintf(name, sizeof name, "32x32 persistent pipelined, %d workgroups per CU", per_cu);
                run(name, [&] { k_pipe<32, 32, 0><<<256 * per_cu, 256>>>(src, labels, out, N, T, U, V, 0, tiles_t, tiles_u, total); });
                check(name);
            }
        }
    }
    return 0;
}
