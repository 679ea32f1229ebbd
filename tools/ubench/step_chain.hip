// Micro-benchmark: what one diagonal of the hand-written lattice block (csrc/lattice_step.h) costs a lone wave64 on gfx950,
// and what the links of its dependent chain cost one by one.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-D<variant macro of lattice_step.h>] step_chain.hip -o step_chain
// Part 1: dependent chains of single instructions in the encodings the step uses (4-byte e32, 8-byte VOP3 / DPP / literal).
// Part 2: the real blocks -- ws::compute_block_ip<16, BETA, BLOCK_FULL, SEEDED> in a loop over a two-slot LDS ring with the
// in-place reloads, the value stores and the end-of-block wait, no barrier -- cycles per diagonal and a checksum of the
// state (so that two builds of the step can be compared for bits with the same run).
#include "../../warp_rnnt_amd/csrc/lattice_step.h"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace rnnt;
using namespace rnnt::ws;

#define N_IT 4000
#define REP8(S) S S S S S S S S

template <int MODE>
__global__ void k_chain(float* out, long long* cyc, const float* in) {
    float a = in[threadIdx.x], b = in[64 + threadIdx.x], c = in[128 + threadIdx.x], d = in[192 + threadIdx.x];
    float ln2s, l2es;
    asm volatile("s_mov_b32 %0, 0x3f317218\n\ts_mov_b32 %1, 0x3fb8aa3b" : "=s"(ln2s), "=s"(l2es));
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N_IT / 8; ++i) {
        if constexpr (MODE == 0) asm volatile(REP8("v_add_f32_e32 %0, %1, %0\n\t") : "+v"(a) : "v"(b));
        if constexpr (MODE == 1) asm volatile(REP8("v_add_f32_e64 %0, %1, %0\n\t") : "+v"(a) : "v"(b));
        if constexpr (MODE == 2) asm volatile(REP8("v_fmac_f32 %0, 0x3f317218, %1\n\t") : "+v"(a) : "v"(b));          // a += ln2 * b, but chain on a
        if constexpr (MODE == 3) asm volatile(REP8("v_fmac_f32_e32 %0, %2, %1\n\t") : "+v"(a) : "v"(b), "s"(ln2s));
        if constexpr (MODE == 4) asm volatile(REP8("v_mul_f32_e64 %0, -|%0|, %1\n\t") : "+v"(a) : "s"(l2es));
        if constexpr (MODE == 5) asm volatile(REP8("v_mul_f32_e32 %0, %1, %0\n\t") : "+v"(a) : "s"(l2es));
        if constexpr (MODE == 6) asm volatile(REP8("v_exp_f32 %0, %0\n\tv_add_f32 %0, %1, %0\n\t") : "+v"(a) : "v"(b));   // exp + add
        if constexpr (MODE == 7) asm volatile(REP8("v_log_f32 %0, %0\n\tv_add_f32 %0, %1, %0\n\t") : "+v"(a) : "v"(b));   // log + add
        if constexpr (MODE == 8) asm volatile(REP8("v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_sub_f32 %0, %2, %1\n\ts_nop 0\n\t") : "+v"(a), "+v"(c) : "v"(b));
        if constexpr (MODE == 9) asm volatile(REP8("v_sub_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t") : "+v"(a) : "v"(b));
        if constexpr (MODE == 10) asm volatile(REP8("v_add_f32_e32 %0, %1, %0\n\tv_add_f32_e32 %2, %1, %2\n\t") : "+v"(a), "+v"(c) : "v"(b));   // + 1 filler (independent chain)
        if constexpr (MODE == 11) asm volatile(REP8("v_add_f32_e32 %0, %1, %0\n\tv_add_f32_e32 %2, %1, %2\n\tv_add_f32_e32 %3, %1, %3\n\t") : "+v"(a), "+v"(c), "+v"(d) : "v"(b));
        if constexpr (MODE == 12) asm volatile(REP8("v_add_f32_e32 %0, %1, %0\n\ts_nop 0\n\t") : "+v"(a) : "v"(b));
        if constexpr (MODE == 13) asm volatile(REP8("v_add_f32_e32 %0, %1, %0\n\ts_nop 1\n\t") : "+v"(a) : "v"(b));
        if constexpr (MODE == 14) asm volatile(REP8("v_cndmask_b32_e64 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(b), "s"(1ull));
        if constexpr (MODE == 15) asm volatile(REP8("v_max_f32_e32 %0, %1, %0\n\t") : "+v"(a) : "v"(b));
        if constexpr (MODE == 16) asm volatile(REP8("v_add_f32_e32 %0, %1, %0\n\tv_add_f32_e32 %0, %1, %0\n\ts_nop 0\n\tv_sub_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t") : "+v"(a) : "v"(b));
        if constexpr (MODE == 17) asm volatile(REP8("v_add_f32_e32 %0, %1, %0\n\tv_add_f32_e32 %0, %1, %0\n\ts_nop 0\n\tv_mov_b32_dpp %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_sub_f32 %0, %1, %2\n\t") : "+v"(a), "+v"(c) : "v"(b));
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run_chain(const char* name, int links, float* out, long long* cyc, float* in) {
    k_chain<MODE><<<1, 64>>>(out, cyc, in);
    (void)hipDeviceSynchronize();
    k_chain<MODE><<<1, 64>>>(out, cyc, in);
    (void)hipDeviceSynchronize();
    long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %7.2f cycles/iter  %6.2f per chain link\n", name, (double)c / N_IT, (double)c / N_IT / links);
}

constexpr int KK = 16;
struct Ring {
    f32x2 pairs[2][KK][WAVE];
    float seeds[2][KK];
    float vals[2][KK][WAVE];
};

// SYNC: 0 = the lone wave; 1 = the workgroup has two more waves that do nothing but meet this one at an s_barrier per block
// (the barrier's own cost when the others are already there); 2 = the lone wave WITHOUT the end-of-block wait (timing only:
// what the exposed LDS round trip of the last reloads costs)
template <bool BETA, bool SEEDED, int SYNC>
__global__ void k_block(float* out, long long* cyc, const float* in, int nblocks) {
    __shared__ __attribute__((aligned(16))) Ring sm;
    const int lane = threadIdx.x & 63;
    if (SYNC == 1 && threadIdx.x >= 64) {
        __syncthreads();
        __syncthreads();
#pragma nounroll
        for (int b = 0; b < nblocks; ++b) __builtin_amdgcn_s_barrier();
        return;
    }
    for (int s = 0; s < 2; ++s)
        for (int k = 0; k < KK; ++k) {
            const int j = (s * KK + k) * WAVE + lane;
            sm.pairs[s][k][lane] = f32x2{-0.3f - 0.01f * (float)(in[j & 1023] + (j % 7)), -2.0f - 0.02f * (float)((j * 5) % 11)};
            if (lane == 0) sm.seeds[s][k] = -1.5f * (float)(k + 1 + s * KK);
            sm.vals[s][k][lane] = 0.0f;
        }
    __syncthreads();
    f32x4 cur2[KK / 2], seed4[KK / 4];
    for (int j = 0; j < KK / 2; ++j) {
        const f32x2 a0 = sm.pairs[0][2 * j][lane], a1 = sm.pairs[0][2 * j + 1][lane];
        cur2[j] = f32x4{a0.x, a0.y, a1.x, a1.y};
    }
    for (int j = 0; j < KK / 4; ++j) seed4[j] = f32x4{sm.seeds[0][4 * j], sm.seeds[0][4 * j + 1], sm.seeds[0][4 * j + 2], sm.seeds[0][4 * j + 3]};
    float Y = -(float)lane, X = -(float)lane - 1.0f;
    const unsigned pairs0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)&sm.pairs[0][0][lane];
    const unsigned seeds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)&sm.seeds[0][0];
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
#pragma nounroll
    for (int b = 0; b < nblocks; ++b) {
        const int nslot = (b + 1) & 1;
        unsigned nsrc = pairs0 + (unsigned)nslot * (unsigned)(KK * WAVE * 8);
        unsigned nseed = __builtin_amdgcn_readfirstlane(seeds0 + (unsigned)nslot * (KK * 4));
        lds_float* vslot = (lds_float*)&sm.vals[b & 1][0][lane];
        asm volatile("" : "+v"(nsrc), "+s"(nseed), "+v"(vslot));
        // (seeds drift with the block so that the boundary stays near the interior's magnitude)
        compute_block_ip<KK, BETA, BLOCK_FULL, SEEDED>(cur2, seed4, nsrc, nseed, Y, X, b * KK, SEEDED ? 64 + lane : lane, 1 << 30, vslot, 0);
        if constexpr (SYNC != 2) wait_lds_keep<KK, SEEDED>(cur2, seed4);
        if constexpr (SYNC == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    }
    if constexpr (SYNC == 2) wait_lds();
    long long t1 = __builtin_readcyclecounter();
    out[lane] = Y;
    out[64 + lane] = X;
    out[128 + lane] = sm.vals[(nblocks - 1) & 1][KK - 1][lane];
    if (lane == 0) cyc[0] = t1 - t0;
}

template <bool BETA, bool SEEDED, int SYNC = 0>
void run_block(const char* name, float* out, long long* cyc, float* in) {
    const int nblocks = 500;
    k_block<BETA, SEEDED, SYNC><<<1, SYNC == 1 ? 192 : 64>>>(out, cyc, in, nblocks);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k_block<BETA, SEEDED, SYNC><<<1, SYNC == 1 ? 192 : 64>>>(out, cyc, in, nblocks);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    std::vector<unsigned> h(192);
    (void)hipMemcpy(h.data(), out, 192 * 4, hipMemcpyDeviceToHost);
    unsigned long long sum = 1469598103934665603ull;
    for (unsigned w : h) sum = (sum ^ w) * 1099511628211ull;
    float y0, y63;
    memcpy(&y0, &h[0], 4);
    memcpy(&y63, &h[63], 4);
    printf("%-34s %7.2f cycles/diagonal  %6.2f ns/diagonal (events, launch included)  state %016llx  Y[0] %.6g Y[63] %.6g\n", name,
           (double)c / (nblocks * KK), ms * 1e6 / (nblocks * KK), sum, y0, y63);
}

int main() {
    float *out, *in;
    long long* cyc;
    (void)hipMalloc(&out, 4096);
    (void)hipMalloc(&in, 4096);
    (void)hipMalloc(&cyc, 64);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 1.0f + (float)(i % 13) * 0.125f;
    (void)hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    run_chain<0>("v_add_f32_e32 (4 bytes)", 1, out, cyc, in);
    run_chain<1>("v_add_f32_e64 (8 bytes)", 1, out, cyc, in);
    run_chain<15>("v_max_f32_e32", 1, out, cyc, in);
    run_chain<2>("v_fmac_f32 literal (8 bytes)", 1, out, cyc, in);
    run_chain<3>("v_fmac_f32_e32 sgpr (4 bytes)", 1, out, cyc, in);
    run_chain<4>("v_mul_f32_e64 -|x|, sgpr (8 bytes)", 1, out, cyc, in);
    run_chain<5>("v_mul_f32_e32 sgpr (4 bytes)", 1, out, cyc, in);
    run_chain<14>("v_cndmask_b32_e64 sgpr pair (8 bytes)", 1, out, cyc, in);
    run_chain<6>("v_exp_f32 + v_add", 2, out, cyc, in);
    run_chain<7>("v_log_f32 + v_add", 2, out, cyc, in);
    run_chain<8>("v_mov_b32_dpp + v_sub + s_nop 0", 2, out, cyc, in);
    run_chain<9>("v_sub_f32_dpp + s_nop 1", 1, out, cyc, in);
    run_chain<17>("add, add, s_nop 0, v_mov_b32_dpp, v_sub", 4, out, cyc, in);
    run_chain<16>("add, add, s_nop 0, v_sub_f32_dpp", 3, out, cyc, in);
    run_chain<10>("v_add + 1 independent v_add", 1, out, cyc, in);
    run_chain<11>("v_add + 2 independent v_add", 1, out, cyc, in);
    run_chain<12>("v_add + s_nop 0", 1, out, cyc, in);
    run_chain<13>("v_add + s_nop 1", 1, out, cyc, in);
    run_block<false, true>("alpha block, left neighbour", out, cyc, in);
    run_block<false, false>("alpha block, sweep column 0", out, cyc, in);
    run_block<true, true>("beta block, left neighbour", out, cyc, in);
    run_block<true, false>("beta block, sweep column 0", out, cyc, in);
    run_block<false, true, 1>("alpha, left nb, + s_barrier/block", out, cyc, in);
    run_block<true, true, 1>("beta, left nb, + s_barrier/block", out, cyc, in);
    run_block<false, true, 2>("alpha, left nb, no end wait (!)", out, cyc, in);
    return 0;
}
