// Micro-benchmark: copy of a (rows, V) fp32 tensor in the SHAPES a row-wise kernel can take (a workgroup or a wave owns
// a whole row: all loads, a reduction, all stores) -- what k_lsm_large could reach at V = 5000 / 10000.
// hipcc --offload-arch=gfx950 -O3 row_copy.hip -o row_copy && ./row_copy [rows V]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// one workgroup per row, NV float4 per thread in registers, one block reduction between loads and stores
template <int TH, int NV>
__global__ void __launch_bounds__(TH) k_row_wg(const f4* __restrict__ a, f4* __restrict__ b, int nvec) {
    __shared__ float red[TH / 64];
    const f4* src = a + (size_t)blockIdx.x * nvec;
    f4* dst = b + (size_t)blockIdx.x * nvec;
    f4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { const int j = threadIdx.x + i * TH; if (j < nvec) { v[i] = src[j]; s += v[i].x; } }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < TH / 64; ++i) t += red[i];
#pragma unroll
    for (int i = 0; i < NV; ++i) { const int j = threadIdx.x + i * TH; if (j < nvec) { f4 o = v[i]; o.x += t * 1e-30f; dst[j] = o; } }
}
// one wave per row (4 rows per 256-thread workgroup), no barrier
template <int NV>
__global__ void __launch_bounds__(256) k_row_wave(const f4* __restrict__ a, f4* __restrict__ b, int nvec, size_t rows) {
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const f4* src = a + row * nvec;
    f4* dst = b + row * nvec;
    f4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { const int j = lane + i * 64; if (j < nvec) { v[i] = src[j]; s += v[i].x; } }
    s = wave_sum(s);
#pragma unroll
    for (int i = 0; i < NV; ++i) { const int j = lane + i * 64; if (j < nvec) { f4 o = v[i]; o.x += s * 1e-30f; dst[j] = o; } }
}
// row-agnostic: one float4 per thread (the best plain copy)
__global__ void __launch_bounds__(256) k_one(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}
// a workgroup per row, chunked: each wave walks its quarter of the row 1 KiB at a time, partial sums first (pass 1:
// loads only), then a second pass re-reads the row (L2) and stores -- the "two passes, nothing held" shape
template <int TH>
__global__ void __launch_bounds__(TH) k_row_two_pass(const f4* __restrict__ a, f4* __restrict__ b, int nvec) {
    __shared__ float red[TH / 64];
    const f4* src = a + (size_t)blockIdx.x * nvec;
    f4* dst = b + (size_t)blockIdx.x * nvec;
    float s = 0.f;
    for (int j = threadIdx.x; j < nvec; j += TH) s += src[j].x;
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < TH / 64; ++i) t += red[i];
    for (int j = threadIdx.x; j < nvec; j += TH) { f4 o = src[j]; o.x += t * 1e-30f; dst[j] = o; }
}

template <typename F>
static void run(const char* name, F launch, double bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int r = 0; r < 10; ++r) {
        hipEventRecord(e0);
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) ts.push_back(ms / 3);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-52s median %8.1f us   %.2f TB/s\n", name, ts[ts.size() / 2] * 1e3, 2.0 * bytes / (ts[ts.size() / 2] * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const size_t rows = argc > 1 ? atoll(argv[1]) : 96000;
    const int V = argc > 2 ? atoi(argv[2]) : 5000;
    const int nvec = V / 4;
    const double bytes = (double)rows * V * 4;
    f4 *a, *b;
    (void)hipMalloc(&a, (size_t)bytes); (void)hipMalloc(&b, (size_t)bytes);
    (void)hipMemset(a, 1, (size_t)bytes);
    printf("rows=%zu V=%d: %.2f GB read + the same written\n", rows, V, bytes / 1e9);
    const size_t n = rows * (size_t)nvec;
    run("row-agnostic, 1 float4 / thread", [&] { k_one<<<(unsigned)((n + 255) / 256), 256>>>(a, b, n); }, bytes);
    run("in place, 1 float4 / thread", [&] { k_one<<<(unsigned)((n + 255) / 256), 256>>>(a, a, n); }, bytes);
    if (nvec <= 256 * 8) run("workgroup per row, 256 x 8 (k_lsm_large at V<=8192)", [&] { k_row_wg<256, 8><<<(unsigned)rows, 256>>>(a, b, nvec); }, bytes);
    if (nvec <= 256 * 5) run("workgroup per row, 256 x 5", [&] { k_row_wg<256, 5><<<(unsigned)rows, 256>>>(a, b, nvec); }, bytes);
    if (nvec <= 512 * 3) run("workgroup per row, 512 x 3", [&] { k_row_wg<512, 3><<<(unsigned)rows, 512>>>(a, b, nvec); }, bytes);
    if (nvec <= 512 * 8) run("workgroup per row, 512 x 8 (k_lsm_large at V>8192)", [&] { k_row_wg<512, 8><<<(unsigned)rows, 512>>>(a, b, nvec); }, bytes);
    if (nvec <= 512 * 5) run("workgroup per row, 512 x 5", [&] { k_row_wg<512, 5><<<(unsigned)rows, 512>>>(a, b, nvec); }, bytes);
    if (nvec <= 1024 * 3) run("workgroup per row, 1024 x 3", [&] { k_row_wg<1024, 3><<<(unsigned)rows, 1024>>>(a, b, nvec); }, bytes);
    if (nvec <= 1024 * 2) run("workgroup per row, 1024 x 2", [&] { k_row_wg<1024, 2><<<(unsigned)rows, 1024>>>(a, b, nvec); }, bytes);
    if (nvec <= 64 * 20) run("wave per row, 20 float4 / lane", [&] { k_row_wave<20><<<(unsigned)((rows + 3) / 4), 256>>>(a, b, nvec, rows); }, bytes);
    if (nvec <= 64 * 40) run("wave per row, 40 float4 / lane", [&] { k_row_wave<40><<<(unsigned)((rows + 3) / 4), 256>>>(a, b, nvec, rows); }, bytes);
    run("workgroup per row, two passes (256)", [&] { k_row_two_pass<256><<<(unsigned)rows, 256>>>(a, b, nvec); }, bytes);
    run("workgroup per row, two passes (512)", [&] { k_row_two_pass<512><<<(unsigned)rows, 512>>>(a, b, nvec); }, bytes);
    run("workgroup per row, two passes (1024)", [&] { k_row_two_pass<1024><<<(unsigned)rows, 1024>>>(a, b, nvec); }, bytes);
    if (nvec <= 256 * 8) run("in place: workgroup per row, 256 x 8", [&] { k_row_wg<256, 8><<<(unsigned)rows, 256>>>(a, a, nvec); }, bytes);
    return 0;
}
