// Micro-benchmark: cycles per lattice step of a lone wave64 on gfx950 for candidate arithmetic schemes.
// hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize pd_steps.hip -o pd_steps && ./pd_steps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_IT 4000

__device__ __forceinline__ float shr1f(float s) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ int shr1i(int s) { return __builtin_amdgcn_mov_dpp(s, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ double shr1d(double src) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, src);
    const int lo = __builtin_amdgcn_mov_dpp((int)b, 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

template <int MODE>
__global__ void k(float* out, long long* cyc, const float* in, float seed) {
    const float pb = in[threadIdx.x] * 0.97f + seed * 1e-6f, pl = in[threadIdx.x + 64] * 0.31f;
    float Y = 0.7f, X = 0.2f, c = 1.0f;
    double Yd = 0.7, Xd = 0.2, cd = 1.0;
    const double pbd = pb, pld = pl;
    int E = 0;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int i = 0; i < N_IT; ++i) {
        if constexpr (MODE == 0) {          // fp32: mov_dpp + fma + 2 mul
            const float val = __builtin_fmaf(shr1f(X), c, Y);
            X = val * pl; Y = val * pb;
        }
        if constexpr (MODE == 1) {          // fp32: v_fmac_f32_dpp + 2 mul
            float val = Y;
            asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(val) : "v"(X), "v"(c));
            X = val * pl; Y = val * pb;
        }
        if constexpr (MODE == 2) {          // fp64: 2 mov_dpp + fma + 2 mul
            const double val = __builtin_fma(shr1d(Xd), cd, Yd);
            Xd = val * pld; Yd = val * pbd;
        }
        if constexpr (MODE == 3) {          // fp32 + exact renormalisation every step (exponent path on the chain)
            const float val = __builtin_fmaf(shr1f(X), c, Y);
            const float m = __builtin_amdgcn_frexp_mantf(val);
            const int e = __builtin_amdgcn_frexp_expf(val);
            E += e;
            X = m * pl; Y = m * pb;
            c = __builtin_ldexpf(1.0f, shr1i(E) - E);
        }
        if constexpr (MODE == 4) {          // fp32 + renormalisation every second step
            float val = __builtin_fmaf(shr1f(X), c, Y);
            X = val * pl; Y = val * pb;
            if (i & 1) {
                const int e = __builtin_amdgcn_frexp_expf(Y);
                Y = __builtin_amdgcn_frexp_mantf(Y);
                X = __builtin_ldexpf(X, -e);
                E += e;
                c = __builtin_ldexpf(1.0f, shr1i(E) - E);
            }
        }
        if constexpr (MODE == 5) { Yd = __builtin_fma(Yd, pbd, 1e-3); }                       // dependent v_fma_f64
        if constexpr (MODE == 6) { Yd = __builtin_fma(Yd, pbd, 1e-3); Xd = __builtin_fma(Xd, pld, 1e-3); }   // 2 independent
        if constexpr (MODE == 7) { Y = __builtin_fmaf(Y, pb, 1e-3f); }                         // dependent v_fma_f32
        if constexpr (MODE == 8) { Y = __builtin_fmaf(Y, pb, 1e-3f); X = __builtin_fmaf(X, pl, 1e-3f); }
        if constexpr (MODE == 9) { Y = shr1f(Y) + 0.25f; }                                     // mov_dpp + add
        if constexpr (MODE == 10) { Yd = Yd * pbd; }                                           // dependent v_mul_f64
        if constexpr (MODE == 11) { Y = __builtin_amdgcn_frexp_mantf(Y + 0.3f); }              // add + frexp_mant
        if constexpr (MODE == 12) {         // the log-domain step (beta form) for reference
            const float left = shr1f(X);
            const float skip = Y + pb, emit = left + pl;
            const float t = skip - emit;
            const float mx = __builtin_fmaxf(skip, emit);
            const float e = __builtin_amdgcn_exp2f(-__builtin_fabsf(t) * 1.44269504088896340736f);
            const float u = 1.0f + e;
            const float cc = e - (u - 1.0f);
            const float l2 = __builtin_amdgcn_logf(u);
            Y = mx + __builtin_fmaf(l2, 0.693147180559945309417f, cc);
            X = Y;
        }
        if constexpr (MODE == 13) {         // fp64 beta form: tmp = pl*c ; S = Y*pb ; val = fma(xl, tmp, S)
            const double w = pld * cd;
            const double val = __builtin_fma(shr1d(Xd), w, Yd * pbd);
            Xd = val; Yd = val;
        }
        if constexpr (MODE == 14) {         // fp32 beta form
            const float w = pl * c;
            const float val = __builtin_fmaf(shr1f(X), w, Y * pb);
            X = val; Y = val;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = Y + X + c + (float)(Yd + Xd + cd) + (float)E;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, long long* cyc, float* in) {
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
    printf("%-52s %8.1f cycles/step   %7.1f ns/step (kernel %.1f us)\n", name, (double)c / N_IT, ms * 1e6 / N_IT, ms * 1e3);
}

int main() {
    float *out, *in; long long* cyc;
    hipMalloc(&out, 1 << 21); hipMalloc(&in, 4096); hipMalloc(&cyc, 64);
    std::vector<float> h(1024, 1.0f);
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    run<7>("dependent v_fma_f32", out, cyc, in);
    run<8>("2 independent v_fma_f32", out, cyc, in);
    run<5>("dependent v_fma_f64", out, cyc, in);
    run<6>("2 independent v_fma_f64", out, cyc, in);
    run<10>("dependent v_mul_f64", out, cyc, in);
    run<9>("mov_dpp + add (dep)", out, cyc, in);
    run<11>("add + frexp_mant (dep)", out, cyc, in);
    run<0>("fp32 step: mov_dpp, fma, 2 mul", out, cyc, in);
    run<1>("fp32 step: v_fmac_dpp, 2 mul", out, cyc, in);
    run<14>("fp32 beta step: mul, mul, mov_dpp, fma", out, cyc, in);
    run<2>("fp64 step: 2 mov_dpp, fma, 2 mul", out, cyc, in);
    run<13>("fp64 beta step", out, cyc, in);
    run<3>("fp32 step + renorm every step", out, cyc, in);
    run<4>("fp32 step + renorm every 2nd step", out, cyc, in);
    run<12>("log-domain step (lse)", out, cyc, in);
    return 0;
}
