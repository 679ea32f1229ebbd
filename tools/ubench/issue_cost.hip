// Micro-benchmark: cycles per instruction for a single wave64 on gfx950 (dependent chains).
// hipcc --offload-arch=gfx950 -O3 issue_cost.hip -o issue_cost && ./issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_IT 2000

template <int MODE>
__global__ void k(float* out, long long* cyc, const float* in, float seed) {
    __shared__ float lds[512];
    float a = in[threadIdx.x] + seed, b = in[threadIdx.x + 64] + seed, c = a * 0.5f, d = b * 0.25f;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 1 << 20, 0x00020000);
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int i = 0; i < N_IT; ++i) {
        if constexpr (MODE == 0) { a = __builtin_fmaf(a, 1.0001f, 0.5f); }                         // 1 dep fma
        if constexpr (MODE == 1) { a = __builtin_fmaf(a, 1.0001f, 0.5f); b = __builtin_fmaf(b, 1.0001f, 0.5f); }  // 2 indep
        if constexpr (MODE == 2) { a = __builtin_fmaf(a, 1.0001f, 0.5f); b = __builtin_fmaf(b, 1.0001f, 0.5f);
                                   c = __builtin_fmaf(c, 1.0001f, 0.5f); d = __builtin_fmaf(d, 1.0001f, 0.5f); }  // 4 indep
        if constexpr (MODE == 3) { a = __builtin_amdgcn_exp2f(a * -0.001f); }                        // mul + exp dep
        if constexpr (MODE == 4) { a = __builtin_amdgcn_logf(a + 1.5f); }                            // add + log dep
        if constexpr (MODE == 5) { a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x138, 0xf, 0xf, false)) + 1.0f; } // dpp + add
        if constexpr (MODE == 6) { float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63)); a = a + s * 1e-9f; } // readlane + fma
        if constexpr (MODE == 7) { a = __builtin_fmaf(a, 1.0001f, 0.5f);
                                   __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, a), rs, threadIdx.x * 4, (i & 1023) * 256, 0); } // fma + store
        if constexpr (MODE == 8) { a = __builtin_fmaf(a, 1.0001f, 0.5f);
                                   __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, a), rs, threadIdx.x * 4, 0, 0); } // fma + store same addr
        if constexpr (MODE == 9) { a = __builtin_fmaf(a, 1.0001f, 0.5f);
                                   asm volatile("s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0" ::: ); }         // fma + 4 s_nop
        if constexpr (MODE == 10) { a = __builtin_fmaxf(a, b) + 0.5f; }                             // max + add dep
        if constexpr (MODE >= 12 && MODE <= 15) {
            // one beta-like lattice step: dpp -> add -> lse -> (store) (ds_write)
            float left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, c), __builtin_bit_cast(int, a), 0x138, 0xf, 0xf, false));
            float skip = a + b;
            float emit = left + d;
            float t = skip - emit;
            float mx = __builtin_fmaxf(skip, emit);
            float e = __builtin_amdgcn_exp2f(-__builtin_fabsf(t) * 1.44269504088896340736f);
            float u = 1.0f + e;
            float cc = e - (u - 1.0f);
            float l2 = __builtin_amdgcn_logf(u);
            a = mx + __builtin_fmaf(l2, 0.693147180559945309417f, cc);
            if constexpr (MODE == 13 || MODE == 15)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, a), rs, threadIdx.x * 4, (i & 1023) * 256, 0);
            if constexpr (MODE == 14 || MODE == 15) lds[(i & 7) * 64 + threadIdx.x] = a;
        }
        if constexpr (MODE == 11) { typedef float f2 __attribute__((ext_vector_type(2)));
                                    f2 v = {a, b}; f2 w = {c, c}; v = v + w; a = v.x; b = v.y; }      // pk_add
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a + b + c + d + lds[threadIdx.x];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int ninstr, float* out, long long* cyc, float* in) {
    k<MODE><<<1, 64>>>(out, cyc, in, 0.f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<1, 64>>>(out, cyc, in, 1.f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %8.1f cycles/iter  %6.2f cyc/instr   (kernel %.1f us -> %.2f GHz shader clock if cycle counter = shader clock)\n",
           name, (double)c / N_IT, (double)c / N_IT / ninstr, ms * 1e3, c / (ms * 1e6));
}

int main() {
    float *out, *in; long long* cyc;
    hipMalloc(&out, 1 << 21); hipMalloc(&in, 4096); hipMalloc(&cyc, 64);
    std::vector<float> h(1024, 1.0f);
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    run<0>("1 dependent fma", 1, out, cyc, in);
    run<1>("2 independent fma", 2, out, cyc, in);
    run<2>("4 independent fma", 4, out, cyc, in);
    run<3>("mul+exp (dep)", 2, out, cyc, in);
    run<4>("add+log (dep)", 2, out, cyc, in);
    run<5>("dpp wave_shr + add (dep)", 2, out, cyc, in);
    run<6>("readlane + fma (dep)", 2, out, cyc, in);
    run<7>("fma + buffer_store", 2, out, cyc, in);
    run<8>("fma + buffer_store same addr", 2, out, cyc, in);
    run<9>("fma + 4 s_nop", 5, out, cyc, in);
    run<10>("max + add (dep)", 2, out, cyc, in);
    run<11>("pk_add", 1, out, cyc, in);
    run<12>("lattice step (chain only)", 12, out, cyc, in);
    run<13>("lattice step + store", 13, out, cyc, in);
    run<14>("lattice step + ds_write", 13, out, cyc, in);
    run<15>("lattice step + store + ds_write", 14, out, cyc, in);
    return 0;
}
