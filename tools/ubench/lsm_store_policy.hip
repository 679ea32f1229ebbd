// Micro-benchmark (round 6): the cache policy of the log-softmax's 16-byte STORES.  The dense gather gained 2 % from
// writing its pairs through (sc1: DESIGN.md 3.5 -- dirty lines on their way out of L2 hold back the read stream's fills);
// the log-softmax writes as much as it reads.  The register-only V=50 kernel of lsm_regs.hip (two row pairs per 32-lane
// half, the shipped shape), loads non-temporal or plain, stores under each of the policies the ISA has:
//   plain / nt / sc1 / sc0 sc1 / sc1 nt / sc0 sc1 nt / sc0
// (the inline-assembly stores carry an `s_nop 1`: a store of more than 64 bits reads its data registers up to two wait states
//  after it issues and the compiler pads only its own -- without it 0.18 % of one lane group's float4 were wrong: _isa_check.py)
// and the same for a bare float4-per-thread copy.
// hipcc --offload-arch=gfx950 -O3 lsm_store_policy.hip -o lsm_store_policy && ./lsm_store_policy
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.693147180559945309417f;

template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float swz16(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}
__device__ __forceinline__ float half_max(float v) {
    v = fmaxf(v, dpp<0xB1>(v)); v = fmaxf(v, dpp<0x4E>(v)); v = fmaxf(v, dpp<0x124>(v)); v = fmaxf(v, dpp<0x128>(v));
    return fmaxf(v, swz16(v));
}
__device__ __forceinline__ float half_sum(float v) {
    v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x124>(v); v += dpp<0x128>(v);
    return v + swz16(v);
}
enum { ST_PLAIN, ST_NT, ST_SC1, ST_SC0SC1, ST_SC1NT, ST_SC0SC1NT, ST_SC0 };
template <int P> __device__ __forceinline__ void store4(f4* p, f4 v) {
    if constexpr (P == ST_PLAIN) *p = v;
    else if constexpr (P == ST_NT) __builtin_nontemporal_store(v, p);
    else if constexpr (P == ST_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == ST_SC0SC1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == ST_SC1NT) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == ST_SC0SC1NT) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <bool NT> __device__ __forceinline__ f4 load4(const f4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

template <int UN, bool LNT, int SP>
__global__ void __launch_bounds__(256) k_lsm_rp(const f4* __restrict__ x, f4* __restrict__ out, size_t npairs) {
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool act = j < 25;
    f4 v[UN];
#pragma unroll
    for (int i = 0; i < UN; ++i) {
        const size_t p = (w * UN + i) * 2 + half;
        v[i] = f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (act && p < npairs) v[i] = load4<LNT>(x + p * 25 + j);
    }
    const bool a01 = j <= 12, a23 = j < 12;
#pragma unroll
    for (int i = 0; i < UN; ++i) {
        const size_t p = (w * UN + i) * 2 + half;
        const f4 t = v[i];
        const float m01 = fmaxf(t.x, t.y), m23 = fmaxf(t.z, t.w);
        const float ma = half_max(a23 ? fmaxf(m01, m23) : (a01 ? m01 : -INFINITY));
        const float mb = half_max(a23 ? -INFINITY : (a01 ? m23 : fmaxf(m01, m23)));
        const float k01 = (a01 ? ma : mb) * LOG2E, k23 = (a23 ? ma : mb) * LOG2E;
        const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(t.x, LOG2E, -k01)), e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(t.y, LOG2E, -k01));
        const float e2 = __builtin_amdgcn_exp2f(__builtin_fmaf(t.z, LOG2E, -k23)), e3 = __builtin_amdgcn_exp2f(__builtin_fmaf(t.w, LOG2E, -k23));
        const float s01 = act ? e0 + e1 : 0.f, s23 = act ? e2 + e3 : 0.f;
        const float sa = half_sum((a01 ? s01 : 0.f) + (a23 ? s23 : 0.f));
        const float sb = half_sum((a01 ? 0.f : s01) + (a23 ? 0.f : s23));
        const float la = ma + __builtin_amdgcn_logf(sa) * LN2, lb = mb + __builtin_amdgcn_logf(sb) * LN2;
        const float l01 = a01 ? la : lb, l23 = a23 ? la : lb;
        const f4 r = f4{t.x - l01, t.y - l01, t.z - l23, t.w - l23};
        if (act && p < npairs) store4<SP>(out + p * 25 + j, r);
    }
}
// The same kernel with its results leaving through LDS in LINEAR order: the wave's four row pairs are 1600 contiguous bytes
// (25 granules of 64 bytes); written to a per-wave LDS strip at their offsets and read back 16 bytes per lane in address
// order, they leave as one store instruction of 1024 bytes and one of 576 -- every store instruction covers whole 64-byte
// granules, where the direct form's 400-byte segments split them (tools/ubench/copy_shape.hip: stores off the 64-byte
// grid cost a copy 5-19 %, loads off it nothing).  No workgroup barrier: the strip is the wave's own.
template <bool LNT, int SP>
__global__ void __launch_bounds__(256) k_lsm_rp_lin(const f4* __restrict__ x, f4* __restrict__ out, size_t npairs) {
    __shared__ f4 strip[4][100];
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5, wv = threadIdx.x >> 6;
    const size_t w = (size_t)blockIdx.x * 4 + wv;
    const bool act = j < 25;
    f4 v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const size_t p = (w * 2 + i) * 2 + half;
        v[i] = f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (act && p < npairs) v[i] = load4<LNT>(x + p * 25 + j);
    }
    const bool a01 = j <= 12, a23 = j < 12;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f4 t = v[i];
        const float m01 = fmaxf(t.x, t.y), m23 = fmaxf(t.z, t.w);
        const float ma = half_max(a23 ? fmaxf(m01, m23) : (a01 ? m01 : -INFINITY));
        const float mb = half_max(a23 ? -INFINITY : (a01 ? m23 : fmaxf(m01, m23)));
        const float k01 = (a01 ? ma : mb) * LOG2E, k23 = (a23 ? ma : mb) * LOG2E;
        const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(t.x, LOG2E, -k01)), e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(t.y, LOG2E, -k01));
        const float e2 = __builtin_amdgcn_exp2f(__builtin_fmaf(t.z, LOG2E, -k23)), e3 = __builtin_amdgcn_exp2f(__builtin_fmaf(t.w, LOG2E, -k23));
        const float s01 = act ? e0 + e1 : 0.f, s23 = act ? e2 + e3 : 0.f;
        const float sa = half_sum((a01 ? s01 : 0.f) + (a23 ? s23 : 0.f));
        const float sb = half_sum((a01 ? 0.f : s01) + (a23 ? 0.f : s23));
        const float la = ma + __builtin_amdgcn_logf(sa) * LN2, lb = mb + __builtin_amdgcn_logf(sb) * LN2;
        const float l01 = a01 ? la : lb, l23 = a23 ? la : lb;
        if (act) strip[wv][(2 * i + half) * 25 + j] = f4{t.x - l01, t.y - l01, t.z - l23, t.w - l23};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const size_t f0 = w * 100, nf = npairs * 25;
    if (f0 + lane < nf) store4<SP>(out + f0 + lane, strip[wv][lane]);
    if (lane < 36 && f0 + 64 + lane < nf) store4<SP>(out + f0 + 64 + lane, strip[wv][64 + lane]);
}
template <bool LNT, int SP>
__global__ void __launch_bounds__(256) k_copy1(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) store4<SP>(b + i, load4<LNT>(a + i));
}
// one workgroup per row of V = 10000 (c5's shape, in place): 2500 float4, 1024 threads, up to three per thread
template <bool LNT, int SP>
__global__ void __launch_bounds__(1024) k_row_copy(f4* __restrict__ a, int q) {
    f4* row = a + (size_t)blockIdx.x * q;
    f4 v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int c = threadIdx.x + i * 1024; if (c < q) v[i] = load4<LNT>(row + c); }
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int c = threadIdx.x + i * 1024; if (c < q) store4<SP>(row + c, v[i] + 1.0f); }
}
template <typename F> static float run(const char* name, F launch, size_t bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int r = 0; r < 12; ++r) {
        hipEventRecord(e0); for (int i = 0; i < 4; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 2) ts.push_back(ms / 4);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-58s median %8.1f us  min %8.1f us   %.2f TB/s (read + write)\n", name, ts[ts.size() / 2] * 1e3, ts[0] * 1e3,
           2.0 * bytes / (ts[ts.size() / 2] * 1e-3) / 1e12);
    fflush(stdout);
    return ts[ts.size() / 2];
}
int main() {
    const size_t rows = (size_t)16 * 1500 * 300, V = 50, bytes = rows * V * 4, n = bytes / 16, npairs = rows / 2;
    f4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    std::vector<float> h(rows * V);
    unsigned s = 777;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 16777216.0f - 0.5f) * 8.f; }
    hipMemcpy(a, h.data(), bytes, hipMemcpyHostToDevice);
    printf("V=50, %zu rows, %.2f GB in + the same out; two passes over the list (order effects)\n", rows, bytes / 1e9);
#define LSM(LNT, SP, label) run(label, [&] { k_lsm_rp<2, LNT, SP><<<(unsigned)((npairs / 2 + 7) / 8), 256>>>(a, b, npairs); }, bytes);
#define CPY(LNT, SP, label) run(label, [&] { k_copy1<LNT, SP><<<(unsigned)((n + 255) / 256), 256>>>(a, b, n); }, bytes);
    for (int pass = 0; pass < 2; ++pass) {
        LSM(true, ST_NT, "log-softmax regs, nt loads, nt stores (shipped)")
        LSM(true, ST_PLAIN, "log-softmax regs, nt loads, plain stores")
        LSM(true, ST_SC1, "log-softmax regs, nt loads, sc1 stores")
        LSM(true, ST_SC0SC1, "log-softmax regs, nt loads, sc0 sc1 stores")
        LSM(true, ST_SC1NT, "log-softmax regs, nt loads, sc1 nt stores")
        LSM(true, ST_SC0SC1NT, "log-softmax regs, nt loads, sc0 sc1 nt stores")
        LSM(true, ST_SC0, "log-softmax regs, nt loads, sc0 stores")
        LSM(false, ST_NT, "log-softmax regs, plain loads, nt stores")
        LSM(false, ST_SC1, "log-softmax regs, plain loads, sc1 stores")
        LSM(false, ST_PLAIN, "log-softmax regs, plain loads, plain stores")
#define LIN(LNT, SP, label) run(label, [&] { k_lsm_rp_lin<LNT, SP><<<(unsigned)((npairs / 2 + 7) / 8), 256>>>(a, b, npairs); }, bytes);
        LIN(true, ST_NT, "log-softmax regs, stores LINEAR through LDS, nt / nt")
        LIN(true, ST_PLAIN, "log-softmax regs, stores LINEAR through LDS, nt / plain")
        LIN(true, ST_SC1, "log-softmax regs, stores LINEAR through LDS, nt / sc1")
        LIN(true, ST_SC1NT, "log-softmax regs, stores LINEAR through LDS, nt / sc1 nt")
        LSM(true, ST_NT, "log-softmax regs, nt loads, nt stores (shipped) again")
        CPY(false, ST_PLAIN, "copy 1 float4 / thread, plain / plain")
        CPY(true, ST_NT, "copy 1 float4 / thread, nt / nt")
        CPY(true, ST_SC1, "copy 1 float4 / thread, nt / sc1")
        CPY(true, ST_SC0SC1, "copy 1 float4 / thread, nt / sc0 sc1")
        CPY(true, ST_SC1NT, "copy 1 float4 / thread, nt / sc1 nt")
        CPY(false, ST_SC1, "copy 1 float4 / thread, plain / sc1")
    }
    // every policy must land the same bits: b poisoned, one launch, read back, compared with the plain-store result
    {
        std::vector<float> ref(rows * V), got(rows * V);
        auto one = [&](auto launch, std::vector<float>& dst) {
            hipMemset(b, 0xFF, bytes); hipDeviceSynchronize(); launch(); hipDeviceSynchronize();
            hipMemcpy(dst.data(), b, bytes, hipMemcpyDeviceToHost);
        };
        const unsigned grid = (unsigned)((npairs / 2 + 7) / 8);
        one([&] { k_lsm_rp<2, true, ST_PLAIN><<<grid, 256>>>(a, b, npairs); }, ref);
        auto cmp = [&](const char* name) {
            size_t bad = 0, first = 0; unsigned lanes[32] = {0};
            for (size_t i = 0; i < rows * V; ++i)
                if (memcmp(&ref[i], &got[i], 4)) { if (!bad) first = i; ++bad; lanes[(i % 100) / 4]++; }
            printf("  %-14s elements differing from plain stores: %zu", name, bad);
            if (bad) { printf(" (first at %zu = pair %zu float %zu: %g vs %g; by float4 of the pair:", first, first / 100, first % 100, got[first], ref[first]);
                       for (int j = 0; j < 25; ++j) printf(" %u", lanes[j]); printf(")"); }
            printf("\n");
        };
#define CHK(SP, label) one([&] { k_lsm_rp<2, true, SP><<<grid, 256>>>(a, b, npairs); }, got); cmp(label);
        one([&] { k_lsm_rp_lin<true, ST_NT><<<grid, 256>>>(a, b, npairs); }, got); cmp("linear nt");
        one([&] { k_lsm_rp_lin<true, ST_SC1><<<grid, 256>>>(a, b, npairs); }, got); cmp("linear sc1");
        CHK(ST_NT, "nt") CHK(ST_SC1, "sc1") CHK(ST_SC0SC1, "sc0 sc1") CHK(ST_SC1NT, "sc1 nt") CHK(ST_SC0SC1NT, "sc0 sc1 nt") CHK(ST_SC0, "sc0")
        one([&] { k_copy1<true, ST_SC1><<<(unsigned)((n + 255) / 256), 256>>>(a, b, n); }, got);
        printf("  copy nt / sc1 against its input: %s\n", memcmp(got.data(), h.data(), bytes) ? "DIFFERENT" : "same");
    }
    // check the last log-softmax against fp64 on sampled rows
    k_lsm_rp<2, true, ST_SC1><<<(unsigned)((npairs / 2 + 7) / 8), 256>>>(a, b, npairs);
    std::vector<float> o(rows * V);
    hipMemcpy(o.data(), b, bytes, hipMemcpyDeviceToHost);
    double worst = 0;
    for (size_t r = 0; r < rows; r += 9973) {
        double m = -1e30, sum = 0;
        for (size_t c = 0; c < V; ++c) m = std::max(m, (double)h[r * V + c]);
        for (size_t c = 0; c < V; ++c) sum += std::exp((double)h[r * V + c] - m);
        for (size_t c = 0; c < V; ++c) worst = std::max(worst, std::fabs((double)o[r * V + c] - ((double)h[r * V + c] - m - std::log(sum))));
    }
    printf("max |out - fp64 log-softmax| over sampled rows (sc1 stores): %.2e\n", worst);
    hipFree(b);
    // c5's shape: rows of V = 10000 rewritten in place, 8 GB of them (the library's row-per-workgroup kernel moves 144 GB a step)
    {
        const int q = 2500; const size_t nrows = 200000, rb = nrows * q * 16;
        f4* c; hipMalloc(&c, rb); hipMemset(c, 0, rb);
        printf("rows of V=10000 in place, %zu rows, %.1f GB read + the same written\n", nrows, rb / 1e9);
#define ROW(LNT, SP, label) run(label, [&] { k_row_copy<LNT, SP><<<(unsigned)nrows, 1024>>>(c, q); }, rb);
        for (int pass = 0; pass < 2; ++pass) {
            ROW(true, ST_NT, "row copy in place, nt / nt (shipped policy)")
            ROW(true, ST_PLAIN, "row copy in place, nt / plain")
            ROW(true, ST_SC1, "row copy in place, nt / sc1")
            ROW(true, ST_SC0SC1, "row copy in place, nt / sc0 sc1")
            ROW(true, ST_SC1NT, "row copy in place, nt / sc1 nt")
            ROW(false, ST_PLAIN, "row copy in place, plain / plain")
            ROW(false, ST_SC1, "row copy in place, plain / sc1")
        }
    }
    return 0;
}
