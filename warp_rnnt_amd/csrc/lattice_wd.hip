// Distributed log-domain alpha / beta lattice sweep for MI355X (gfx950), diagonal-major layout (padded or compact):
// the reference's arithmetic -- one fp32 lse per cell, core_gather.cu:22-35,106-126,207-227 -- with ONE WORKGROUP PER
// 64-COLUMN BLOCK of a sweep.
//
// The step is lattice_ws.hip's, literally (lattice_step.h: same instructions on the chain, same bits): lanes are
// lattice columns, blocks of K = 8 diagonals, one s_barrier per block, a COMPUTE wave that touches only registers and
// LDS.  What changes is everything around it.
//
// 1. WHERE the column blocks run.  In lattice_ws.hip they share a workgroup, hence a CU, and all meet at every
//    barrier.  Here a column block is a workgroup of its own (three waves), wherever the dispatcher puts it, and the
//    boundary column travels through L2:
//      * ring: one 8-byte granule {fp32 value, 32-bit tag} per diagonal and (sweep, boundary); the producer's storer
//        wave publishes the 8 granules of a block with one agent-scope (sc1) store instruction the interval after the
//        compute wave produced them; tag = launch epoch ^ hash(ring) ^ hash(diagonal), never 0, and the rings are
//        zeroed in front of every launch (launch_ring_prepare, below): a granule validates only if THIS launch wrote it;
//      * the consumer's loader wave fetches the granules of a block DLOAD intervals ahead (LDS-DMA, like the pairs),
//        checks the tags when the block is due, and only if the producer is not there yet polls (after letting it get
//        LAG blocks further ahead, so that the following look-ahead fetches hit).  No flag, no fence, no back-pressure:
//        the ring is as long as the sweep;
//      * work items (column block, sweep) come from an atomic counter in column-block-major order: whoever holds item
//        i knows that its left neighbour, item i - 2N, is held by a workgroup that is running or done -- no assumption
//        about dispatch order or residency, any batch size;
//      * every wait is bounded; a timeout flags the sweep for the single-workgroup kernel launched behind (which
//        returns at once otherwise) instead of hanging.
// 2. The memory side has NO data-dependent control flow around memory instructions and no compiler-counted loads.
//    Measured (profiles/r04_wd_trace_*.txt, ISA): with "guarded" steps at the ends of a column block's life (memory
//    instructions under conditions) the waitcnt pass assumes the worst at every join and each such interval waits for
//    everything in flight -- ~1.5 us instead of 0.4 -- and because the heads of a sweep's column blocks run one
//    after the other that was ~20 us per boundary, in lattice_ws.hip too.  Here:
//      * LOADER wave: the (blank,label) pairs of a block arrive by LDS-DMA (four `buffer_load_dwordx4 ... lds`: two
//        diagonals of 64 pairs each), the neighbour's granules by a fifth; every interval issues all five (an
//        out-of-range offset where there is nothing to fetch: zeros land), waits with ONE hand-counted
//        `s_waitcnt vmcnt` and never holds a byte of it in a register; polls are inline assembly that drains the
//        queue itself;
//      * STORER wave: values LDS -> HBM one block behind (per-lane predicate, dropped by the buffer range check) and the
//        publication of the boundary column; stores only, it never waits for memory.
// 3. Instruction fetch.  A wave runs a 200-instruction block of straight-line code cold in ~2.4 us (the trace: first
//    block of each variant), and a column block's first blocks are on the critical path of the whole sweep.  So the
//    block exists ONCE per variant (compute_block_ip: in-place prefetch instead of two register buffers and a loop
//    unrolled twice) and the storer wave -- idle for the first intervals anyway -- runs the compute loop "dry" for
//    one block per variant before it takes up its own work: same instructions, same addresses, a warm instruction
//    cache when the compute wave gets there.
// Values handed over are the compute wave's own fp32 X registers, so the results are bit-identical to
// lattice_ws.hip's whatever the placement and timing (tests/test_gpu_wd.py).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <random>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "lattice_step.h"

namespace rnnt {

#define RNNT_WD_NS wd8
#define RNNT_WD_KK 8
#include "lattice_wd_body.h"
#undef RNNT_WD_NS
#undef RNNT_WD_KK
#define RNNT_WD_NS wd16
#define RNNT_WD_KK 16
#include "lattice_wd_body.h"
#undef RNNT_WD_NS
#undef RNNT_WD_KK

// Blocks of 16 diagonals from launch bound T >= 1024 on, blocks of 8 below (tools/lattice_routes.py, us per alpha+beta launch,
// 8 / 16: N=16, T=1500: U=64 68.7 / 64.5, U=300 102.4 / 100.2, U=512 123.8 / 122.6; T=700, U=100 48.2 / 48.8; T=150, U=40
// 13.2 / 13.7; round 6, whole c4 step in bench.py: 0.8306 / 0.8219 ms).  History: the choice for most of round 5, then
// switched off at its end -- with several processes sharing the GPU (tools/wd_soak.py) this instantiation was where the
// storer's dry run, reloads left in flight into registers the compiler reused (lattice_wd_body.h: one_block), showed as
// lost hand-overs and rare wrong plane values -- and opt-in while the fix was two commits old.  Re-qualified in round 6:
// the reload check is part of the build, the end-of-block wait holds the refilled registers as operands, and the soak
// record (profiles/r06_wd_soak.txt: millions of launches under three and six processes, every launch compared with
// k_lattice_ws's bits) has no mismatch and no lost hand-over on either block size.
// ONE column block per sweep (U <= 64) hands nothing over, so the longer blocks cost no hand-over distance: 16 from T >= 320
// while every workgroup has a CU of its own (end of round 6, profiles/r06_k16_threshold.txt, 8 / 16: N=16, U=64: T=350
// 22.2 / 21.5, T=500 28.5 / 27.5, T=700 36.8 / 34.6, T=1100 53.2 / 49.4; T=150, U=40 12.6 / 12.9 and N=256, T=500 36.4 / 37.3
// stay with 8); several column blocks: T=640, U=300 63.3 / 64.7, T=900 76.3 / 75.3, T=1024 81.9 / 80.6 -- 1024 stays.
// RNNT_WD_K16_FROM_T=<T> replaces both thresholds (1: blocks of 16 everywhere; a huge T: blocks of 8 everywhere; same bits).
static int wd_block_diagonals(hipStream_t stream, int T, int U, int N) {
    static const int from_t = getenv("RNNT_WD_K16_FROM_T") ? atoi(getenv("RNNT_WD_K16_FROM_T")) : -1;
    if (from_t >= 0) return T >= from_t ? 16 : 8;
    if (U <= WAVE && T >= 320 && (long long)2 * N <= (long long)device_cus(stream)) return 16;
    return T >= 1024 ? 16 : 8;
}

// ---------------------------------------------------------------------------------------------------------------
// Ring preparation (until round 5 in lattice_pd.hip, whose probability-domain kernel introduced the protocol).
// The library's launch counter of this device: module-scope device memory (zero when the code object is loaded,
// never part of anybody's workspace), so it cannot be recycled, scribbled over or left uninitialised, and every
// REPLAY of a captured graph advances it too -- kernel arguments are frozen at capture time, and with a frozen
// epoch the granules of the previous replay would carry this replay's tags.
// ---------------------------------------------------------------------------------------------------------------
__device__ unsigned g_launch_counter;

// In front of every launch: clears the redo flags and the queue head (n words), hands the next value of the launch
// counter to the kernel (p[n]) and zeroes the hand-over rings (mail_vec 16-byte words; tag 0 never validates), so
// that nothing the workspace held before -- it is caller scratch with unspecified contents -- can be taken for a
// granule of this launch.  The clear is 1 % of the bytes the sweeps move.
__global__ void __launch_bounds__(256) k_prepare(int* p, int n, uint4* mail, size_t mail_vec) {
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < n; i += 256) p[i] = 0;
        if (threadIdx.x == 0) p[n] = (int)(atomicAdd(&g_launch_counter, 1u) + 1u);
    }
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < mail_vec; i += (size_t)gridDim.x * 256) mail[i] = z;
}

unsigned next_launch_epoch() {
    // granules of earlier launches (same buffer) never validate.  Random start so that a recycled allocation of
    // another process does not either; the device-side counter (k_prepare) is added in the kernel.
    static std::atomic<unsigned> epoch{std::random_device{}()};
    return epoch.fetch_add(1, std::memory_order_relaxed) + 1;
}

hipError_t launch_ring_prepare(hipStream_t stream, int* flags, int n_flags, void* rings, size_t ring_bytes) {
    // the flags (2N ints) and the queue head are contiguous in the workspace (api.hip: carve).  One tiny kernel:
    // hipMemsetAsync of these few bytes becomes two fill kernels of ~5 us each.
    const size_t mail_vec = ring_bytes / 16;
    const unsigned prep_blocks = (unsigned)std::min<size_t>(512, std::max<size_t>(1, mail_vec / (256 * 8)));
    k_prepare<<<prep_blocks, 256, 0, stream>>>(flags, n_flags, reinterpret_cast<uint4*>(rings), mail_vec);
    return hipGetLastError();
}

// Device address of g_launch_counter on the stream's device, for the producers that carry the preparation out themselves
// (they live in another translation unit: the address travels as a kernel argument).  One query per device and process.
static unsigned* launch_counter_address(hipStream_t stream) {
    static std::atomic<unsigned*> cached[64];
    int dev = -1, cur = -2;
    if (hipStreamGetDevice(stream, &dev) != hipSuccess || hipGetDevice(&cur) != hipSuccess) return nullptr;
    if (dev < 0 || dev >= 64 || dev != cur) return nullptr;     // (hipGetSymbolAddress answers for the CURRENT device)
    unsigned* p = cached[dev].load(std::memory_order_acquire);
    if (!p) {
        void* q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_launch_counter)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        p = static_cast<unsigned*>(q);
        cached[dev].store(p, std::memory_order_release);
    }
    return p;
}

size_t wd_mail_bytes(int N, int T, int U) {
    const int nA = (U + WAVE - 1) / WAVE;
    // (by SHAPE only -- the larger of the two block sizes' rings: a workspace's size must not depend on a setting)
    const size_t pitch = std::max(wd8::ring_pitch(T, U), wd16::ring_pitch(T, U));
    size_t bytes = nA < 2 ? 0 : (size_t)2 * N * (nA - 1) * pitch * sizeof(wd8::u64);
#ifdef RNNT_WD_STATS       // the stamp table of the diagnostics build, behind the rings
    bytes += (size_t)2 * N * nA * std::max(wd8::trace_slots(T, U), wd16::trace_slots(T, U)) * 8 * sizeof(wd8::u64);
#endif
    return bytes;
}

// bytes of rings a launch on `a` uses (by the block size it will run with)
static size_t wd_ring_bytes(hipStream_t stream, const LatticeArgs& a, int N) {
    const int nA = (a.U + WAVE - 1) / WAVE;
    const size_t pitch = wd_block_diagonals(stream, a.T, a.U, N) == 16 ? wd16::ring_pitch(a.T, a.U) : wd8::ring_pitch(a.T, a.U);
    return nA < 2 ? 0 : (size_t)2 * N * (nA - 1) * pitch * sizeof(wd8::u64);
}

bool wd_ring_prep(hipStream_t stream, const LatticeArgs& a, int N, RingPrep* prep) {
    if (N <= 0 || !a.redo || !a.queue || a.queue != a.redo + 2 * N) return false;
    const int nA = (a.U + WAVE - 1) / WAVE;
    if (nA > 1 && !a.mail) return false;
    unsigned* counter = launch_counter_address(stream);
    if (!counter) return false;
    const size_t ring_bytes = wd_ring_bytes(stream, a, N);
    *prep = RingPrep{a.redo, 2 * N + 1, counter, reinterpret_cast<uint4*>(a.mail), ring_bytes / 16};
    return true;
}

// Needs a.redo, a.queue = a.redo + 2N with the launch counter's value behind it (and a.mail of wd_mail_bytes when U > 64);
// zeroes flags, queue head and rings itself unless a.prepared.  With a.redo == nullptr and U <= 64 it is a plain launch.  Sweeps it
// flags in a.redo (a lost hand-over: never observed outside the short-spin build) are for the caller to redo with the
// single-workgroup kernel.
hipError_t launch_lattice_wd(hipStream_t stream, const LatticeArgs& a0, int N) {
    if (N <= 0) return hipSuccess;
    const int nA = (a0.U + WAVE - 1) / WAVE;
    if (a0.offs32 && (nA > 1 || a0.redo)) return hipErrorNotSupported;   // (32-bit offsets: the plain launch only)
#ifdef RNNT_WD_STATS
    if (!a0.mail || !a0.redo || !a0.queue) return hipErrorNotSupported;
    const bool lone = false;
#else
    // one column block per sweep (U <= 64) and no flags asked for: nothing to prepare, nothing to redo behind
    const bool lone = nA == 1 && !a0.redo;
#endif
    if (!lone && (!a0.redo || !a0.queue || (nA > 1 && !a0.mail))) return hipErrorNotSupported;
    if ((long long)2 * N * nA >= (1ll << 31)) return hipErrorNotSupported;
    const bool k16 = wd_block_diagonals(stream, a0.T, a0.U, N) == 16;
    LatticeArgs a = a0;
    if (lone) {
        a.queue = nullptr;
        a.mail = nullptr;
    } else {
        a.epoch = next_launch_epoch();
        if (!a.prepared) {     // (prepared: the producer of this call's pair plane did it at the tail of its own launch)
            const hipError_t e = launch_ring_prepare(stream, a.redo, 2 * N + 1, a.mail, wd_ring_bytes(stream, a, N));
            if (e != hipSuccess) return e;
        }
    }
    const dim3 grid(2 * N * nA), block(3 * WAVE);
    if (k16) {
        if (is_compact(a)) wd16::k_lattice_wd<true><<<grid, block, 0, stream>>>(a, nA);
        else wd16::k_lattice_wd<false><<<grid, block, 0, stream>>>(a, nA);
    } else {
        if (is_compact(a)) wd8::k_lattice_wd<true><<<grid, block, 0, stream>>>(a, nA);
        else wd8::k_lattice_wd<false><<<grid, block, 0, stream>>>(a, nA);
    }
    return hipGetLastError();
}

// The single-workgroup form: one workgroup of 3 * ceil(U / 64) waves per sweep, nothing but the planes in global memory.
// Padded or compact (either offset width); honours a.redo (only the flagged sweeps) and a.beta_only.
// hipErrorNotSupported when the lattice is wider than the workgroup's LDS holds (wl_max_blocks() column blocks).
int wl_max_blocks() {
    // 2 column blocks (U <= 128) fit the 64 KiB every kernel gets; 5 (U <= 320) take the large-LDS opt-in, 148 KiB of
    // the CU's 160 (LDS-DMA lands above 64 KiB as well: M0 carries the full address on gfx950 -- tests/test_gpu_wd.py).
    // RNNT_WL_MAX_BLOCKS = 0 ... 5 overrides (0: the kernel is never chosen), for A/B runs.
    static const int v = [] {
        const char* e = ab_getenv("RNNT_WL_MAX_BLOCKS");
        const int d = e ? atoi(e) : RNNT_WL_DEFAULT_MAX_BLOCKS;
        return d < 0 ? 0 : (d > 5 ? 5 : d);
    }();
    return v;
}

hipError_t launch_lattice_wl(hipStream_t stream, const LatticeArgs& a, int N, int max_blocks) {
    if (N <= 0) return hipSuccess;
    const int nA = (a.U + WAVE - 1) / WAVE;
    const size_t lds = sizeof(wd8::Smem) * nA;
    if (nA > max_blocks || nA > 5 || lds > 160 * 1024) return hipErrorNotSupported;
    const dim3 grid(2 * N), block((RNNT_WL_PAD && nA == 2 ? 8 : 3 * nA) * WAVE);
    const bool compact = is_compact(a);
    const bool wide = nA > 2;                           // which instantiation (launch bounds: 512 / 960 threads)
    const void* fn = wide ? (compact ? reinterpret_cast<const void*>(&wd8::k_lattice_wl<true, 5>)
                                     : reinterpret_cast<const void*>(&wd8::k_lattice_wl<false, 5>))
                          : (compact ? reinterpret_cast<const void*>(&wd8::k_lattice_wl<true, 2>)
                                     : reinterpret_cast<const void*>(&wd8::k_lattice_wl<false, 2>));
    if (lds > 64 * 1024) {
        // > 64 KiB of dynamic LDS: an opt-in per kernel and device (idempotent and thread-safe; remembered per device)
        static std::atomic<bool> attr_set[4][64];
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) dev = -1;
        const int ci = (wide ? 2 : 0) + (compact ? 1 : 0);
        const bool tracked = dev >= 0 && dev < 64;
        if (!tracked || !attr_set[ci][dev].load(std::memory_order_acquire)) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     (int)(sizeof(wd8::Smem) * (wide ? 5 : 2)));
            if (e != hipSuccess) return e;
            if (tracked) attr_set[ci][dev].store(true, std::memory_order_release);
        }
    }
    if (wide) {
        if (compact) wd8::k_lattice_wl<true, 5><<<grid, block, lds, stream>>>(a);
        else wd8::k_lattice_wl<false, 5><<<grid, block, lds, stream>>>(a);
    } else {
        if (compact) wd8::k_lattice_wl<true, 2><<<grid, block, lds, stream>>>(a);
        else wd8::k_lattice_wl<false, 2><<<grid, block, lds, stream>>>(a);
    }
    return hipGetLastError();
}

}  // namespace rnnt
