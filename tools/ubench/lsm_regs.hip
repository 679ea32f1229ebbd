// Micro-benchmark: a register-only log-softmax for V = 50 (the c4 vocabulary) against the shapes a plain copy of the
// same 1.44 GB takes.  Two rows = 100 floats = 25 float4 are 16-byte aligned; a wave holds one such row pair in
// lanes 0-24 and one in lanes 32-56 (50 of 64 lanes busy, the 800 bytes of the two pairs contiguous), reduces the
// row maxima and sums with DPP butterflies inside each 32-lane half (+ one ds_swizzle across its two DPP rows) and
// never touches LDS memory.  Question: does it reach the "one float4 per thread" copy rate (6.27 TB/s) where the
// shipped LDS-staged kernel runs at the "3.2 KB per wave through LDS" rate (5.7-5.9 TB/s)?
// hipcc --offload-arch=gfx950 -O3 lsm_regs.hip -o lsm_regs && ./lsm_regs
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.693147180559945309417f;

template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float swz16(float v) {   // lane ^ 16 inside each 32-lane half
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

// UN row pairs per 32-lane half, loads first.  NT: non-temporal loads and stores.
template <int UN, bool NT>
__global__ void __launch_bounds__(256) k_lsm_rp(const f4* __restrict__ x, f4* __restrict__ out, size_t npairs) {
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool act = j < 25;
    f4 v[UN];
#pragma unroll
    for (int i = 0; i < UN; ++i) {
        const size_t p = (w * UN + i) * 2 + half;
        v[i] = f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (act && p < npairs) v[i] = NT ? __builtin_nontemporal_load(x + p * 25 + j) : x[p * 25 + j];
    }
    const bool a01 = j <= 12, a23 = j < 12;     // elements .x,.y / .z,.w belong to the first row of the pair
    const bool any_a = j <= 12, any_b = j >= 12 && act;
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
        (void)any_a; (void)any_b;
        const f4 r = f4{t.x - l01, t.y - l01, t.z - l23, t.w - l23};
        if (act && p < npairs) { if (NT) __builtin_nontemporal_store(r, out + p * 25 + j); else out[p * 25 + j] = r; }
    }
}
// the same data movement without the arithmetic (50 of 64 lanes)
template <int UN, bool NT>
__global__ void __launch_bounds__(256) k_copy_rp(const f4* __restrict__ x, f4* __restrict__ out, size_t npairs) {
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    f4 v[UN];
#pragma unroll
    for (int i = 0; i < UN; ++i) {
        const size_t p = (w * UN + i) * 2 + half;
        if (j < 25 && p < npairs) v[i] = NT ? __builtin_nontemporal_load(x + p * 25 + j) : x[p * 25 + j];
    }
#pragma unroll
    for (int i = 0; i < UN; ++i) {
        const size_t p = (w * UN + i) * 2 + half;
        if (j < 25 && p < npairs) { if (NT) __builtin_nontemporal_store(v[i], out + p * 25 + j); else out[p * 25 + j] = v[i]; }
    }
}
__global__ void __launch_bounds__(256) k_copy1(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}
template <typename F> static void run(const char* name, F launch, size_t bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int r = 0; r < 12; ++r) {
        hipEventRecord(e0); for (int i = 0; i < 4; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 2) ts.push_back(ms / 4);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-52s median %7.1f us  min %7.1f us   %.2f TB/s (read + write)\n", name, ts[ts.size() / 2] * 1e3, ts[0] * 1e3,
           2.0 * bytes / (ts[ts.size() / 2] * 1e-3) / 1e12);
    fflush(stdout);
}
int main() {
    const size_t rows = (size_t)16 * 1500 * 300, V = 50, bytes = rows * V * 4, n = bytes / 16, npairs = rows / 2;
    f4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    std::vector<float> h(rows * V);
    unsigned s = 777;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 16777216.0f - 0.5f) * 8.f; }
    hipMemcpy(a, h.data(), bytes, hipMemcpyHostToDevice);
    printf("V=50, %zu rows, %.2f GB in + the same out\n", rows, bytes / 1e9);
    run("copy, 1 float4 / thread", [&] { k_copy1<<<(unsigned)((n + 255) / 256), 256>>>(a, b, n); }, bytes);
#define RP(K, UN, NT, label) run(label, [&] { K<UN, NT><<<(unsigned)((npairs / 2 + 4 * UN - 1) / (4 * UN)), 256>>>(a, b, npairs); }, bytes);
    RP(k_copy_rp, 1, false, "copy, row-pair lanes (50/64), 1 pair / half")
    RP(k_copy_rp, 2, false, "copy, row-pair lanes, 2 pairs / half")
    RP(k_copy_rp, 4, false, "copy, row-pair lanes, 4 pairs / half")
    RP(k_copy_rp, 4, true, "copy, row-pair lanes, 4 pairs / half, nt")
    RP(k_lsm_rp, 1, false, "log-softmax in registers, 1 pair / half")
    RP(k_lsm_rp, 2, false, "log-softmax in registers, 2 pairs / half")
    RP(k_lsm_rp, 4, false, "log-softmax in registers, 4 pairs / half")
    RP(k_lsm_rp, 2, true, "log-softmax in registers, 2 pairs / half, nt")
    RP(k_lsm_rp, 4, true, "log-softmax in registers, 4 pairs / half, nt")
    RP(k_lsm_rp, 2, false, "log-softmax in registers, 2 pairs / half (check)")
    // check a sample of rows against a double-precision log-softmax
    std::vector<float> o(rows * V);
    hipMemcpy(o.data(), b, bytes, hipMemcpyDeviceToHost);
    double worst = 0;
    for (size_t r = 0; r < rows; r += 9973) {
        double m = -1e30, sum = 0;
        for (size_t c = 0; c < V; ++c) m = std::max(m, (double)h[r * V + c]);
        for (size_t c = 0; c < V; ++c) sum += std::exp((double)h[r * V + c] - m);
        for (size_t c = 0; c < V; ++c) worst = std::max(worst, std::fabs((double)o[r * V + c] - ((double)h[r * V + c] - m - std::log(sum))));
    }
    printf("max |out - fp64 log-softmax| over sampled rows: %.2e\n", worst);
    return 0;
}
