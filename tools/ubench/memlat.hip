// Dependent-load latency on gfx950: one wave chases pointers through a buffer of a given size
// (stride 4 KiB + random) so every access misses L1/TLB-friendly prefetch; reports ns per load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void chase(const unsigned* p, unsigned* out, int n, long long* cyc) {
    unsigned idx = threadIdx.x == 0 ? 0 : 0;
    long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) idx = p[idx];
    long long t1 = wall_clock64();
    out[0] = idx; cyc[0] = t1 - t0;
}
int main() {
    for (size_t mb : {1, 8, 64, 512, 2048}) {
        size_t n = mb * 1024 * 1024 / 4, stride = 1024;  // one element per 4 KiB
        size_t cnt = n / stride;
        std::vector<unsigned> perm(cnt); std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 g(1); std::shuffle(perm.begin() + 1, perm.end(), g);
        std::vector<unsigned> h(n, 0);
        for (size_t i = 0; i < cnt; ++i) h[perm[i] * stride] = perm[(i + 1) % cnt] * stride;
        unsigned *d, *o; long long* c;
        hipMalloc(&d, n * 4); hipMalloc(&o, 64); hipMalloc(&c, 64);
        hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        int iters = (int)std::min<size_t>(cnt, 20000);
        chase<<<1, 64>>>(d, o, iters, c); hipDeviceSynchronize();
        chase<<<1, 64>>>(d, o, iters, c); hipDeviceSynchronize();
        long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        printf("%5zu MiB footprint: %.1f ns per dependent load (wall_clock64 @100MHz)\n", mb, cy * 10.0 / iters);
        hipFree(d); hipFree(o); hipFree(c);
    }
}
