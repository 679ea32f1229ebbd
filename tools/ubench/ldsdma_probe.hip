// What `buffer_load_dwordx4 ... offen lds` (LDS-DMA) does on gfx950 with the operands lattice_wd.hip's loader wave
// gives it: per-lane source offsets that are only 8-byte aligned, lanes whose offset is out of range, lanes masked
// off by EXEC, and a read by the issuing wave right behind its own vmcnt wait (no barrier).
//   hipcc --offload-arch=gfx950 -O3 -o ldsdma_probe ldsdma_probe.hip && ./ldsdma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = __builtin_amdgcn_readfirstlane(0x00020000);
    return r;
}
__device__ __forceinline__ void dma16(int voff, i32x4 rs, int soff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rs), "s"(soff), "s"(lds) : "memory");
}
// mode 0: aligned offsets; 1: offsets shifted by 8 bytes; 2: odd lanes out of range; 3: lanes >= 4 masked off
__global__ void k(const float* src, float* out, int n, int mode) {
    __shared__ __attribute__((aligned(16))) float sm[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) sm[i] = -7.0f;
    __syncthreads();
    const i32x4 rs = make_rsrc(src, n * 4);
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)sm) + 256;
    int voff = lane * 16 + (mode == 1 ? 8 : 0);
    if (mode == 2 && (lane & 1)) voff = (int)0x80000000;
    if (mode != 3 || lane < 4) dma16(voff, rs, 64, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the issuing wave reads its own lane's 16 bytes and a neighbour's, no barrier
    for (int i = lane; i < 1024; i += 64) out[i] = sm[i];
}
int main() {
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 1024 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    const char* names[] = {"aligned", "shifted by 8 bytes", "odd lanes out of range", "lanes >= 4 masked by EXEC"};
    for (int mode = 0; mode < 4; ++mode) {
        k<<<1, 64>>>(d, o, n, mode);
        std::vector<float> r(1024);
        hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
        printf("mode %d (%s): sm[60..64) =", mode, names[mode]);
        for (int i = 60; i < 64; ++i) printf(" %g", r[i]);
        printf(" | landing zone [64..84) =");
        for (int i = 64; i < 84; ++i) printf(" %g", r[i]);
        printf(" | [316..324) =");
        for (int i = 316; i < 324; ++i) printf(" %g", r[i]);
        printf("\n");
    }
    // expected (soffset 64 bytes = 16 floats): aligned: landing zone = 16,17,18,...; shifted: 18,19,20,...
    return 0;
}
