// Micro-benchmark (round 6): WHY does a copy in the rows-in-registers shape of k_lsm_regs at V=50 -- 25 of every 32 lanes
// busy, 400-byte segments that start off the 128-byte grid -- run 5-8 % below the plain float4-per-thread copy of the same
// bytes?  Lane utilisation or alignment?  All variants move 1.44 GB in and 1.44 GB out (non-temporal both ways), two
// segments per 32-lane half as the kernel has them; they differ in how many lanes of a half are busy (G float4 per segment)
// and in where the segments start:
//   G=25            the kernel's shape: 400-byte segments back to back (every second one starts mid-line)
//   G=24            384-byte segments = three whole lines each, 24 of 32 lanes
//   G=32            512-byte segments, every lane busy (the plain copy in this indexing)
//   G=25 in 512     400 bytes used of every 512-byte slot: 25 of 32 lanes, every segment line-aligned (a padded tensor)
//   G=32 shifted    512-byte segments that all start 16 bytes off the grid: every lane busy, every access straddles lines
// hipcc --offload-arch=gfx950 -O3 copy_shape.hip -o copy_shape && ./copy_shape
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

// segment s (of G float4, SLOT float4 apart, the whole array shifted by SHIFT float4) -> lanes 0..G-1 of a half
template <int G, int SLOT, int SHIFT, int UN, int SHIFT_ST = SHIFT, int GL = G, int OFFL = 0>
__global__ void __launch_bounds__(256) k_copy_seg(const f4* __restrict__ x, f4* __restrict__ out, size_t nseg) {
    const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5;
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    f4 v[UN];
#pragma unroll
    for (int i = 0; i < UN; ++i) {
        const size_t s = (w * UN + i) * 2 + half;
        // GL / OFFL: the LOAD may cover more lanes than the segment (an aligned superset: lanes OFFL.. hold the segment)
        if (j < GL && s < nseg) v[i] = __builtin_nontemporal_load(x + s * SLOT + SHIFT + j - OFFL);
    }
#pragma unroll
    for (int i = 0; i < UN; ++i) {
        const size_t s = (w * UN + i) * 2 + half;
        if (j >= OFFL && j < OFFL + G && s < nseg) __builtin_nontemporal_store(v[i], out + s * SLOT + SHIFT_ST + j - OFFL);
    }
}
__global__ void __launch_bounds__(256) k_copy1(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}
template <typename F> static void run(const char* name, F launch, double bytes) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<float> ts;
    for (int r = 0; r < 12; ++r) {
        (void)hipEventRecord(e0); for (int i = 0; i < 4; ++i) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (r >= 2) ts.push_back(ms / 4);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-64s median %7.1f us  min %7.1f us   %.2f TB/s (read + write)\n", name, ts[ts.size() / 2] * 1e3, ts[0] * 1e3,
           2.0 * bytes / (ts[ts.size() / 2] * 1e-3) / 1e12);
    fflush(stdout);
}
int main() {
    const size_t bytes = (size_t)16 * 1500 * 300 * 50 * 4, n = bytes / 16;      // 1.44 GB = 90 M float4
    const size_t alloc = bytes * 512 / 400 + (1 << 20);                          // room for the padded variant
    f4 *a, *b; (void)hipMalloc(&a, alloc); (void)hipMalloc(&b, alloc); (void)hipMemset(a, 0, alloc);
    printf("1.44 GB in + the same out, non-temporal; two passes\n");
#define SEG(G, SLOT, SHIFT, UN, label) { const size_t nseg = n / G; \
        run(label, [&] { k_copy_seg<G, SLOT, SHIFT, UN><<<(unsigned)((nseg / 2 + 4 * UN - 1) / (4 * UN)), 256>>>(a, b, nseg); }, (double)nseg * G * 16); }
    for (int pass = 0; pass < 2; ++pass) {
        run("plain copy, one float4 per thread", [&] { k_copy1<<<(unsigned)((n + 255) / 256), 256>>>(a, b, n); }, (double)bytes);
        SEG(25, 25, 0, 2, "G=25: 400-byte segments back to back (the kernel's shape)")
        SEG(24, 24, 0, 2, "G=24: 384-byte segments, whole lines, 24 of 32 lanes")
        SEG(32, 32, 0, 2, "G=32: 512-byte segments, every lane busy")
        SEG(25, 32, 0, 2, "G=25 in 512-byte slots: 25 of 32 lanes, line-aligned")
        SEG(32, 32, 1, 2, "G=32 shifted by 16 bytes: every lane busy, off the grid")
        SEG(24, 24, 1, 2, "G=24 shifted by 16 bytes")
        SEG(25, 25, 0, 1, "G=25, one segment per half")
        SEG(25, 25, 0, 4, "G=25, four segments per half")
        SEG(32, 32, 0, 1, "G=32, one segment per half")
#define SEGX(G, SLOT, SL, SS, UN, GL, OFFL, label) { const size_t nseg = n / G - 4; \
        run(label, [&] { k_copy_seg<G, SLOT, SL, UN, SS, GL, OFFL><<<(unsigned)((nseg / 2 + 4 * UN - 1) / (4 * UN)), 256>>>(a + 8, b + 8, nseg); }, (double)nseg * G * 16); }
        SEGX(32, 32, 1, 0, 2, 32, 0, "G=32: loads off the grid, stores aligned")
        SEGX(32, 32, 0, 1, 2, 32, 0, "G=32: loads aligned, stores off the grid")
        SEGX(32, 32, 1, 0, 1, 32, 0, "G=32, one segment: loads off the grid, stores aligned")
        SEGX(32, 32, 0, 1, 1, 32, 0, "G=32, one segment: loads aligned, stores off the grid")
        SEGX(32, 32, 0, 2, 2, 32, 0, "G=32: stores 32 bytes off the grid (whole sectors, split lines)")
        SEGX(32, 32, 0, 4, 2, 32, 0, "G=32: stores 64 bytes off the grid (half lines)")
        SEGX(32, 32, 0, 2, 1, 32, 0, "G=32, one segment: stores 32 bytes off the grid")
        SEGX(32, 32, 0, 4, 1, 32, 0, "G=32, one segment: stores 64 bytes off the grid")
        SEGX(24, 24, 0, 0, 1, 24, 0, "G=24, one segment per half")
        SEGX(24, 24, 0, 0, 4, 24, 0, "G=24, four segments per half")
        SEGX(26, 26, 0, 0, 2, 26, 0, "G=26: 416-byte segments (whole sectors, 26 of 32 lanes)")
        SEGX(28, 28, 0, 0, 2, 28, 0, "G=28: 448-byte segments (half lines)")
        // the kernel's 400-byte segments with the LOAD widened to the 512 aligned bytes around them (emulated: every segment
        // shifted by 7 lanes inside a 32-lane load that starts 7 float4 earlier -- the same bytes stored, 28 % more loaded)
        SEGX(25, 25, 0, 0, 2, 32, 7, "G=25 stored, 32 lanes loaded around it (two per half)")
        SEGX(25, 25, 0, 0, 1, 32, 7, "G=25 stored, 32 lanes loaded around it (one per half)")
        SEGX(25, 25, 0, 0, 4, 32, 7, "G=25 stored, 32 lanes loaded around it (four per half)")
    }
    return 0;
}
