// Micro-benchmark of lattice-step instruction orderings for one wave64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 4000
__device__ __forceinline__ float dpp_shr1(float first, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first), __builtin_bit_cast(int, src), 0x138, 0xf, 0xf, false));
}
#define PIN() __builtin_amdgcn_sched_barrier(0)
template <int MODE>
__global__ void k(float* out, long long* cyc, const float* in, float fs) {
    __shared__ float lds[1024];
    float val = in[threadIdx.x], b = in[64 + threadIdx.x] * 0.01f, l = in[128 + threadIdx.x] * 0.01f;
    float skip = val + b;
    float ln2s, first_s = fs;
    asm volatile("s_mov_b32 %0, 0x3f317218" : "=s"(ln2s));
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int i = 0; i < N_IT; ++i) {
        if constexpr (MODE == 0) {   // order as the compiler-pinned version in lattice_ws.hip
            float fk = first_s; float left = dpp_shr1(fk, val); PIN();
            lds[(i & 7) * 64 + threadIdx.x] = val; PIN();
            float sk = val + b, em = left + l; PIN();
            float t = sk - em; PIN();
            float mx; asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(sk), "v"(em)); PIN();
            float m = -__builtin_fabsf(t) * 1.44269504088896340736f; PIN();
            float e = __builtin_amdgcn_exp2f(m); PIN();
            float u = 1.0f + e; PIN();
            float l2 = __builtin_amdgcn_logf(u); PIN();
            float c = e - (u - 1.0f); PIN();
            float ll = __builtin_fmaf(l2, 0.693147180559945309417f, c); PIN();
            val = mx + ll; PIN();
        } else {                     // reordered: fillers in the shadows, SGPR ln2, scalar adds, early v_mov
            float fk; asm volatile("v_mov_b32 %0, %1" : "=v"(fk) : "s"(first_s));
            PIN();
            float left = dpp_shr1(fk, val); PIN();
            lds[(i & 7) * 64 + threadIdx.x] = val; PIN();
            float em = left + l; asm volatile("" : "+v"(em)); PIN();
            float t = skip - em; PIN();
            float m = -__builtin_fabsf(t) * 1.44269504088896340736f; PIN();
            float e = __builtin_amdgcn_exp2f(m); PIN();
            float mx; asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(skip), "v"(em)); PIN();
            float u = 1.0f + e; PIN();
            float l2 = __builtin_amdgcn_logf(u); PIN();
            float um1 = u - 1.0f; PIN();
            float c = e - um1; PIN();
            float ll; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(ll) : "v"(l2), "s"(ln2s), "v"(c)); PIN();
            val = mx + ll; PIN();
            skip = val + b; PIN();
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = val + skip + lds[threadIdx.x];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, float* out, long long* cyc, float* in) {
    k<MODE><<<1, 64>>>(out, cyc, in, -1e30f); (void)hipDeviceSynchronize();
    k<MODE><<<1, 64>>>(out, cyc, in, -1e30f); (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    float o; (void)hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost);
    printf("%-40s %7.1f cycles/step   (val %g)\n", name, (double)c / N_IT, o);
}
int main() {
    float *out, *in; long long* cyc;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&in, 4096); (void)hipMalloc(&cyc, 64);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = -(float)(i % 17);
    (void)hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    run<0>("compiler-pinned order", out, cyc, in);
    run<1>("reordered + sgpr ln2 + scalar adds", out, cyc, in);
}
