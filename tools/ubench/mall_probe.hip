// Does the 256 MB Infinity Cache keep what a streaming kernel has just WRITTEN?  A kernel writes (or copies) the c4
// log-prob tensor (1.44 GB) front to back; a second kernel then reads S bytes from the END of it (what was written
// last) or from the BEGINNING (long evicted), one float4 per thread.  If writes allocate in the cache, the end reads
// faster than the beginning.  Decides whether the gather could profit from the log-softmax kernel's tail.
// hipcc --offload-arch=gfx950 -O3 mall_probe.hip -o mall_probe && ./mall_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_write(f4* __restrict__ b, size_t n, float v) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = f4{v, v, v, v};
}
__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}
template <bool NT>
__global__ void __launch_bounds__(256) k_read(const f4* __restrict__ a, float* sink, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const f4 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (v.x + v.y + v.z + v.w == 123.456f) sink[0] = v.x;
    }
}
int main() {
    const size_t bytes = (size_t)16 * 1500 * 300 * 50 * 4, n = bytes / 16;
    f4 *a, *b; float* sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 64);
    hipMemset(a, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int producer = 0; producer < 2; ++producer)
        for (size_t mb : {16, 32, 64, 128, 192, 256, 512}) {
            const size_t m = mb * 1024 * 1024 / 16;
            for (int where = 0; where < 2; ++where) {
                std::vector<float> ts;
                for (int r = 0; r < 9; ++r) {
                    if (producer) k_copy<<<(unsigned)((n + 255) / 256), 256>>>(a, b, n);
                    else k_write<<<(unsigned)((n + 255) / 256), 256>>>(b, n, (float)r);
                    const f4* src = where ? b + (n - m) : b;
                    hipEventRecord(e0);
                    k_read<false><<<(unsigned)((m + 255) / 256), 256>>>(src, sink, m);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (r) ts.push_back(ms);
                }
                std::sort(ts.begin(), ts.end());
                printf("%s 1.44 GB, then read %4zu MB from the %s: median %7.1f us = %5.2f TB/s\n", producer ? "copy " : "write",
                       mb, where ? "END      " : "BEGINNING", ts[ts.size() / 2] * 1e3, mb * 1.048576e6 / (ts[ts.size() / 2] * 1e-3) / 1e12);
            }
        }
    return 0;
}
