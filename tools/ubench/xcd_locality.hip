// Does it matter WHICH addresses an XCD streams?  Workgroups go to the 8 XCDs round-robin (checked below against
// XCC_ID); memory is interleaved over the HBM stacks at some granularity.  A copy of 1.44 GB -> 1.44 GB in which the
// workgroups of XCD x take, out of every 8 consecutive chunks of S bytes, chunk (x + rot) mod 8: if the interleave is
// visible in virtual addresses and an XCD is closer to some stacks than to others, some (S, rot) run faster than the
// plain order (rot = 0 at S = the workgroup's 16 KB is the plain order).
// hipcc --offload-arch=gfx950 -O3 xcd_locality.hip -o xcd_locality && ./xcd_locality
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr size_t W = 16384;   // bytes per workgroup

// mode 0: copy; 1: read only; 2: write only
template <int MODE>
__global__ void __launch_bounds__(256) k_copy(const char* __restrict__ a, char* __restrict__ b, size_t S, int rot,
                                              unsigned* xcd_mismatch, float* sink) {
    const unsigned w = blockIdx.x, xcd = w & 7, q = w >> 3;
    if (xcd_mismatch && threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        if ((id & 7) != xcd) atomicAdd(xcd_mismatch, 1u);
    }
    const unsigned slot = (xcd + rot) & 7;
    f4 acc = {0, 0, 0, 0};
    if (S <= W) {
        const size_t M = W / S;
        for (size_t j = 0; j < M; ++j) {
            const size_t c = (q * M + j) * 8 + slot;
            const size_t base = c * S;
            for (size_t o = threadIdx.x * 16; o < S; o += 256 * 16) {
                if (MODE != 2) {
                    const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a + base + o));
                    if (MODE == 0) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(b + base + o));
                    else acc += v;
                } else {
                    __builtin_nontemporal_store(f4{1, 2, 3, 4}, reinterpret_cast<f4*>(b + base + o));
                }
            }
        }
    } else {
        const size_t P = S / W, g = q / P, p = q % P;
        const size_t base = (g * 8 + slot) * S + p * W;
        for (size_t o = threadIdx.x * 16; o < W; o += 256 * 16) {
            if (MODE != 2) {
                const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a + base + o));
                if (MODE == 0) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(b + base + o));
                else acc += v;
            } else {
                __builtin_nontemporal_store(f4{1, 2, 3, 4}, reinterpret_cast<f4*>(b + base + o));
            }
        }
    }
    if (MODE == 1 && acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)1408 * 1024 * 1024;            // 1.375 GiB: a multiple of 8 x 4 MiB
    char *a, *b; unsigned* mm; float* sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&mm, 4); hipMalloc(&sink, 64);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(mm, 0, 4);
    printf("a = %p, b = %p\n", (void*)a, (void*)b);
    const unsigned grid = (unsigned)(bytes / W);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_copy<0><<<grid, 256>>>(a, b, W, 0, mm, sink);
    unsigned h = 0; hipMemcpy(&h, mm, 4, hipMemcpyDeviceToHost);
    printf("workgroups whose XCC_ID is not blockIdx %% 8: %u of %u\n", h, grid);
    for (int i = 0; i < 60; ++i) k_copy<0><<<grid, 256>>>(a, b, W, 0, nullptr, sink);   // load state
    const char* names[3] = {"copy", "read", "write"};
    for (int mode = 0; mode < 3; ++mode)
        for (size_t S : {(size_t)256, (size_t)1024, (size_t)4096, (size_t)16384, (size_t)65536, (size_t)262144,
                         (size_t)1048576, (size_t)4194304}) {
            printf("%-5s S = %8zu:", names[mode], S);
            for (int rot = 0; rot < 8; ++rot) {
                std::vector<float> ts;
                for (int r = 0; r < 7; ++r) {
                    hipEventRecord(e0);
                    if (mode == 0) k_copy<0><<<grid, 256>>>(a, b, S, rot, nullptr, sink);
                    else if (mode == 1) k_copy<1><<<grid, 256>>>(a, b, S, rot, nullptr, sink);
                    else k_copy<2><<<grid, 256>>>(a, b, S, rot, nullptr, sink);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (r) ts.push_back(ms);
                }
                std::sort(ts.begin(), ts.end());
                const double gb = (mode == 0 ? 2.0 : 1.0) * bytes / 1e9;
                printf(" %5.2f", gb / ts[ts.size() / 2]);   // GB per ms = TB/s
            }
            printf("  TB/s for rot 0..7\n");
        }
    return 0;
}
