// Micro-benchmark: what a plain float4 copy of the c4 log-prob tensor (1.44 GB in, 1.44 GB out) reaches on this
// part, by kernel shape -- the yardstick for the streaming kernels (log-softmax runs at 5.8 TB/s).
// hipcc --offload-arch=gfx950 -O3 copy_rate.hip -o copy_rate && ./copy_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));

// one float4 per thread
__global__ void __launch_bounds__(256) k_one(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}
// UN float4 per thread, loads first (block covers 256*UN consecutive float4)
template <int UN, bool NT>
__global__ void __launch_bounds__(256) k_unroll(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
    const size_t base = (size_t)blockIdx.x * 256 * UN + threadIdx.x;
    f4 v[UN];
#pragma unroll
    for (int j = 0; j < UN; ++j) {
        const size_t i = base + (size_t)j * 256;
        if (i < n) v[j] = NT ? __builtin_nontemporal_load(a + i) : a[i];
    }
#pragma unroll
    for (int j = 0; j < UN; ++j) {
        const size_t i = base + (size_t)j * 256;
        if (i < n) { if (NT) __builtin_nontemporal_store(v[j], b + i); else b[i] = v[j]; }
    }
}
// persistent grid-stride, UN in flight
template <int UN>
__global__ void __launch_bounds__(256) k_stride(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
    const size_t step = (size_t)gridDim.x * 256 * UN;
    for (size_t base = (size_t)blockIdx.x * 256 * UN + threadIdx.x; base < n; base += step) {
        f4 v[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) { const size_t i = base + (size_t)j * 256; if (i < n) v[j] = a[i]; }
#pragma unroll
        for (int j = 0; j < UN; ++j) { const size_t i = base + (size_t)j * 256; if (i < n) b[i] = v[j]; }
    }
}
// through LDS like the log-softmax kernel: every wave stages 3.2 KB, then writes it out
__global__ void __launch_bounds__(256) k_lds(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
    __shared__ f4 tile[4][200];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t base = ((size_t)blockIdx.x * 4 + w) * 200;
    for (int i = lane; i < 200; i += 64) if (base + i < n) tile[w][i] = a[base + i];
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < 200; i += 64) if (base + i < n) b[base + i] = tile[w][i];
}

// READ-ONLY streams: what a kernel that reads the tensor and writes (next to) nothing can reach -- the yardstick of the
// fused logits -> pairs kernel (4V + 8 bytes per cell, of which 8 written).  UN float4 per thread, loads first; one lane per
// wave keeps the compiler honest with a 4-byte store of a sum.
template <int UN, bool NT>
__global__ void __launch_bounds__(256) k_read(const f4* __restrict__ a, float* __restrict__ sink, size_t n) {
    const size_t base = (size_t)blockIdx.x * 256 * UN + threadIdx.x;
    f4 v[UN];
#pragma unroll
    for (int j = 0; j < UN; ++j) {
        const size_t i = base + (size_t)j * 256;
        v[j] = f4{0.f, 0.f, 0.f, 0.f};
        if (i < n) v[j] = NT ? __builtin_nontemporal_load(a + i) : a[i];
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < UN; ++j) s += v[j].x + v[j].y + v[j].z + v[j].w;
    if (s == 123456.789f) sink[blockIdx.x] = s;          // (never true for this data: the loads cannot be dropped)
}

template <typename F>
static void run(const char* name, F launch, size_t bytes, double streams = 2.0) {
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
    std::sort(ts.begin(), ts.end());
    printf("%-44s median %7.1f us  min %7.1f us   %.2f TB/s (median)\n", name, ts[ts.size() / 2] * 1e3, ts[0] * 1e3,
           streams * bytes / (ts[ts.size() / 2] * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)16 * 1500 * 300 * 50 * 4;
    const size_t n = bytes / 16;
    f4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    printf("copy of %.2f GB (read) + the same written\n", bytes / 1e9);
    run("hipMemcpyAsync DtoD", [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, bytes);
    run("1 float4 / thread", [&] { k_one<<<(unsigned)((n + 255) / 256), 256>>>(a, b, n); }, bytes);
    run("2 float4 / thread", [&] { k_unroll<2, false><<<(unsigned)((n + 511) / 512), 256>>>(a, b, n); }, bytes);
    run("4 float4 / thread", [&] { k_unroll<4, false><<<(unsigned)((n + 1023) / 1024), 256>>>(a, b, n); }, bytes);
    run("8 float4 / thread", [&] { k_unroll<8, false><<<(unsigned)((n + 2047) / 2048), 256>>>(a, b, n); }, bytes);
    run("4 float4 / thread, nontemporal", [&] { k_unroll<4, true><<<(unsigned)((n + 1023) / 1024), 256>>>(a, b, n); }, bytes);
    run("8 float4 / thread, nontemporal", [&] { k_unroll<8, true><<<(unsigned)((n + 2047) / 2048), 256>>>(a, b, n); }, bytes);
    run("grid-stride 2048 blocks x 4", [&] { k_stride<4><<<2048, 256>>>(a, b, n); }, bytes);
    run("grid-stride 4096 blocks x 2", [&] { k_stride<2><<<4096, 256>>>(a, b, n); }, bytes);
    run("LDS-staged, 3.2 KB per wave", [&] { k_lds<<<(unsigned)((n + 799) / 800), 256>>>(a, b, n); }, bytes);
    printf("read-only stream of %.2f GB (TB/s of bytes READ)\n", bytes / 1e9);
    float* sink = reinterpret_cast<float*>(b);
    run("read only, 1 float4 / thread", [&] { k_read<1, false><<<(unsigned)((n + 255) / 256), 256>>>(a, sink, n); }, bytes, 1.0);
    run("read only, 2 float4 / thread", [&] { k_read<2, false><<<(unsigned)((n + 511) / 512), 256>>>(a, sink, n); }, bytes, 1.0);
    run("read only, 4 float4 / thread", [&] { k_read<4, false><<<(unsigned)((n + 1023) / 1024), 256>>>(a, sink, n); }, bytes, 1.0);
    run("read only, 1 float4 / thread, nontemporal", [&] { k_read<1, true><<<(unsigned)((n + 255) / 256), 256>>>(a, sink, n); }, bytes, 1.0);
    run("read only, 2 float4 / thread, nontemporal", [&] { k_read<2, true><<<(unsigned)((n + 511) / 512), 256>>>(a, sink, n); }, bytes, 1.0);
    run("read only, 4 float4 / thread, nontemporal", [&] { k_read<4, true><<<(unsigned)((n + 1023) / 1024), 256>>>(a, sink, n); }, bytes, 1.0);
    run("read only, 8 float4 / thread, nontemporal", [&] { k_read<8, true><<<(unsigned)((n + 2047) / 2048), 256>>>(a, sink, n); }, bytes, 1.0);
    return 0;
}
