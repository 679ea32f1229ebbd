// Alpha / beta lattice sweeps of the RNN-Transducer loss for MI355X (gfx950).
//
// What it computes (per utterance n; maths = SURVEY.md Appendix A; reference
// kernels: core_gather.cu:37-133 (alphas), :135-234 (betas), dense indexing
// core.cu:84-87,116-120,189-192,221-225):
//   alpha[t,u] = lse(alpha[t-1,u] + lpB[t-1,u], alpha[t,u-1] + lpL[t,u-1]),  alpha[0,0] = 0
//   beta [t,u] = lse(beta[t+1,u] + lpB[t,u],    beta[t,u+1] + lpL[t,u]),     beta[T-1,U-1] = lpB[T-1,U-1]
//
// How (nothing like the reference's 32x1 warp tiles ordered by global spin
// locks):
//   * one workgroup per (utterance, direction); lane <-> lattice column,
//     wave w owns columns [64w, 64w+64); up to 16 waves (1024 columns) per
//     pass, wider lattices are swept in column stripes;
//   * true anti-diagonal sweep: at diagonal d every lane computes its cell
//     (d - u, u).  The value a cell needs from its left neighbour moves one
//     lane up with a single DPP `wave_shr:1`; the value it needs from the
//     previous frame is the lane's own register;
//   * wave w runs K diagonals (one "block") behind wave w-1; the boundary
//     column between two waves is handed over through a tiny LDS ring, K
//     values per block, with ONE s_barrier per K diagonals.  No global
//     atomics, no inter-workgroup ordering;
//   * log-probs are read from the diagonal-major workspace (common.h): each
//     diagonal is one contiguous row, K rows are prefetched into registers a
//     block ahead; alphas/betas are written in the same diagonal-major layout
//     with coalesced stores.
//   Critical path: (T_n + U_n - 1) + K*(waves-1) dependent lse steps.
#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "lattice_single.h"

namespace rnnt {

namespace ws { constexpr int MAXA_HOST = 8; }   // column blocks one lattice_ws.hip workgroup sweeps (its MAXA)
using namespace single;

template <int LOADER, bool COMPACT>
__global__ void __launch_bounds__(MAXW * WAVE) k_lattice(const LatticeArgs a) {
    __shared__ float mail[MAXW][RING];
    __shared__ float trash[MAXW][MAIL_TRASH];
    // XCD-aware placement: workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md), each XCD has its
    // own L2.  The alpha and the beta sweep of one utterance read the same diagonal-major plane
    // (from opposite ends), so they are given ids b and b+8: same XCD, shared L2 lines.
    // (Speed only; nothing depends on the placement.)
    const unsigned b = blockIdx.x, pairs_total = gridDim.x >> 1;
    const unsigned grp = b >> 4, in = b & 15;
    unsigned n, dir;
    if ((grp << 3) + 8 <= pairs_total) { n = (grp << 3) + (in & 7); dir = in >> 3; }
    else { const unsigned r = b - (grp << 4); n = (grp << 3) + (r >> 1); dir = r & 1; }   // tail group
    if (a.beta_only && !dir) return;
    if (a.redo && a.redo[2 * n + dir] == 0) return;   // launched behind a ring kernel: only the sweeps it flagged
    if (dir)
        sweep<LOADER, true, COMPACT>(a, n, mail, trash);
    else
        sweep<LOADER, false, COMPACT>(a, n, mail, trash);
}

// compute units of the stream's device (one query per process and device; 256 on MI355X): the kernels with one
// workgroup per column block want CUs of their own for them
int device_cus(hipStream_t stream) {
    static std::atomic<int> cached[64];
    int dev = 0;
    // the device that owns the stream the kernels go to (not necessarily the current one)
    if (hipStreamGetDevice(stream, &dev) != hipSuccess && hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev < 0 || dev >= 64) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

namespace {
int kernel_override_from_env() {
    const char* v = getenv("RNNT_DEBUG_LATTICE_KERNEL");
    if (v && v[0] == 'w' && v[1] == 's') return 1;
    if (v && v[0] == 'w' && v[1] == 'd') return 2;
    if (v && v[0] == 'w' && v[1] == 'l') return 3;
    return 0;
}
std::atomic<int>& kernel_override_setting() {
    static std::atomic<int> r{kernel_override_from_env()};
    return r;
}

// the single-role kernel: one workgroup per sweep, 1024-column stripes, every loader; honours a.redo
hipError_t launch_single(hipStream_t stream, const LatticeArgs& a, int N, int loader) {
    int waves = (a.U + WAVE - 1) / WAVE;
    waves = waves < 1 ? 1 : (waves > MAXW ? MAXW : waves);
    const dim3 grid(2 * N), block(waves * WAVE);
    if (is_compact(a)) {   // compact layout: diagonal-major pairs (native path) or row-major pairs (core.h shims)
        if (loader == LOAD_ROWMAJOR2) k_lattice<LOAD_ROWMAJOR2, true><<<grid, block, 0, stream>>>(a);
        else k_lattice<LOAD_SKEWED, true><<<grid, block, 0, stream>>>(a);
        return hipGetLastError();
    }
    switch (loader) {
        case LOAD_SKEWED:    k_lattice<LOAD_SKEWED, false><<<grid, block, 0, stream>>>(a); break;
        case LOAD_ROWMAJOR2: k_lattice<LOAD_ROWMAJOR2, false><<<grid, block, 0, stream>>>(a); break;
        default:             k_lattice<LOAD_DENSE, false><<<grid, block, 0, stream>>>(a); break;
    }
    return hipGetLastError();
}
}  // namespace

static thread_local int g_last_kernel = 0;
int last_lattice_kernel() { return g_last_kernel; }

int lattice_kernel_override() { return kernel_override_setting().load(std::memory_order_relaxed); }

int set_lattice_kernel_override(int k) {
    if (k < 0 || k > 3) return -1;
    return kernel_override_setting().exchange(k, std::memory_order_relaxed);
}

// Will launch_lattice hand this call to the ring kernel (k_lattice_wd with flags, queue and -- beyond one column block --
// rings)?  The one predicate both launch_lattice and lattice_ring_prep use.
// folded: the ring preparation rides in the launch of the kernel that produces this call's pair plane (the dense and the
// gathered input routes); otherwise it is a launch of its own in front of the sweeps (fused logits, compact layout), ~4 us
// that move the break-even points.
static bool takes_ring_kernel(hipStream_t stream, const LatticeArgs& a, int N, int loader, bool folded) {
#ifdef RNNT_LATTICE_LEGACY
    return false;
#endif
    if (loader != LOAD_SKEWED || N <= 0) return false;
    const int nA = (a.U + WAVE - 1) / WAVE;
    const bool ring_ok = a.redo && a.queue && !a.offs32 && (nA == 1 || a.mail);
    const int kern = lattice_kernel_override();
    // From which sweep length on the column-block kernel wins, by the number of column blocks, measured on the WHOLE loss
    // entry with each kernel pinned (tools/loss_routes.py, profiles/r06_loss_routes.txt: the ring preparation rides in the
    // gather's launch there, which the sweeps-alone probe charges to wd as a launch of its own).  With a CU for every
    // workgroup: two blocks from T >= 900 (wl's LDS hand-over wins below), three from 640, four from 400, five from 320,
    // six and more always (the alternative there is lattice_ws.hip: -9 ... -17 %); with two workgroups per CU the margins
    // are 1 - 2 % of the call either way: four blocks from T >= 1400, five from 800, six and more always, two and three never
    // (N=64, T=1000, U=200 691 / 678 us wd / wl, N=80, T=1000, U=192 735 / 723; N=64, T=1500, U=256 1114 / 1131).
    const long long wgs = (long long)2 * N * nA, cus = device_cus(stream);
    // (not folded: the break-even points of the fused route, `tools/loss_routes.py --fused` in the same file -- two blocks tie
    //  at T=1000 and wd leads by 2.4 % at 1300, three tie up to 900 and wd leads by 1.9 % at 1200, four tie at 500 and wd
    //  leads by 1.1 % at 700, five: wd by 1.8 % at 400)
    const int from_t = folded ? (nA == 2 ? 900 : nA == 3 ? 640 : nA == 4 ? 400 : nA == 5 ? 320 : 128)
                              : (nA == 2 ? 1200 : nA == 3 ? 1100 : nA == 4 ? 640 : nA == 5 ? 400 : 128);
    const int from_t2 = nA <= 3 ? INT_MAX : nA == 4 ? 1400 : nA == 5 ? 800 : 128;
    bool use_wd = ring_ok && ((wgs <= cus && a.T >= from_t) || (wgs <= 2 * cus && a.T >= from_t2));
    if (nA > ws::MAXA_HOST) use_wd = ring_ok;             // wider than one workgroup sweeps: column blocks or stripes
    if (kern == 1 || kern == 3) use_wd = false;
    if (kern == 2) use_wd = ring_ok;
    if (nA == 1 && kern != 1) return false;               // one column block: the plain launch, nothing to prepare
    return use_wd && (long long)2 * N * nA < (1ll << 31);
}

bool lattice_ring_prep(hipStream_t stream, const LatticeArgs& a, int N, int loader, RingPrep* prep) {
    static const bool off = ab_getenv("RNNT_NO_PREP_FOLD") != nullptr;     // A/B knob: k_prepare as a launch of its own
    if (off || !takes_ring_kernel(stream, a, N, loader, true)) return false;
    return wd_ring_prep(stream, a, N, prep);
}

hipError_t launch_lattice(hipStream_t stream, const LatticeArgs& a, int N, int loader) {
    if (N <= 0) return hipSuccess;
    LatticeArgs plain = a;          // for the kernels that sweep everything: no redo flags to look at
    plain.redo = nullptr;
#ifndef RNNT_LATTICE_LEGACY
    if (loader == LOAD_SKEWED) {
        const int nA = (a.U + WAVE - 1) / WAVE;
        // (the kernel that hands boundary columns over through L2 rings needs the flags, the work queue and the rings --
        //  compact layout: the native entry's 64-bit cell offsets; a.T / a.U are then the launch bounds Tmax / Umax:
        //  takes_ring_kernel above)
        // ... and behind it the single-workgroup kernel for the sweeps it flagged (normally none: its workgroups read one
        // flag and return: 5 us per call; redoing inside k_lattice_wd instead was tried in round 6 and costs more,
        // lattice_wd_body.h)
        auto redo_behind = [&]() {
            const hipError_t e = launch_lattice_ws(stream, a, N);
            return e != hipErrorNotSupported ? e : launch_single(stream, a, N, loader);
        };
        // One arithmetic (the reference's: one fp32 lse per cell, core_gather.cu:22-35,106-126), three kernels with the same
        // instructions on the chain and the same bits: lattice_ws.hip (all column blocks of a sweep in one workgroup of
        // compute + I/O wave pairs; one pass covers U <= 512), lattice_wd.hip (one three-wave workgroup per column block,
        // boundary columns through L2) and its single-workgroup form k_lattice_wl (boundary columns through LDS).
        // Which one, measured on MI355X (tools/lattice_routes.py, profiles/r05_lattice_routes.txt; us per alpha+beta launch,
        // ws / wd / wl): two column blocks -- wl: T=700, U=100 61 / 48 / 45, T=400 42 / 35 / 31, N=32, T=250 32 / 30 / 25, N=64,
        // T=300, U=128 39 / 36 / 30 -- except long sweeps on a chip with a CU per workgroup, where wd's blocks of 16 diagonals
        // win since the end of round 6 (profiles/r06_two_block_routes.txt: U=128, N=16: T=1024 85 / 61 / 62, T=1500 116 / 80 / 85,
        // T=2000 148 / 98 / 109, N=8, T=3000 210 / 134 / 157; N=64, T=1500 116 / 82 / 86; N=128 128 / 121 / 110: wl again; the
        // thresholds themselves come from the whole entry: takes_ring_kernel above); three and more: wd while the chip has
        // CUs for its workgroups and the sweep is long enough to recover two extra launches (ring preparation in front, the
        // idle redo kernel behind): N=16, T=1500, U=300 161 / 102 / 128, N=32, T=1000, U=200 100 / 72 / 75; N=32, T=500, U=200
        // 64 / 52 / 51; full chips: N=64, T=1500, U=300 199 / 192 / 189, N=128 307 / 414 / 335.
        // (Until round 5 a probability-domain kernel, lattice_pd.hip, could be chosen here -- another arithmetic, closer to
        // fp64 on long lattices; retired in round 6: slower than wd at c4 since the hand-written blocks, and not the
        // reference's numbers.  profiles/HISTORY.md keeps its measurements.)
        const int kern = lattice_kernel_override();      // debug / A-B only: 0 by shape, 1 ws, 2 wd, 3 wl
        const bool use_wd = takes_ring_kernel(stream, a, N, loader, a.prepared != 0);
        // One column block per sweep (U <= 64): nothing is handed over, so the distributed kernel needs neither the ring
        // preparation in front nor the redo kernel behind -- a plain launch of its three-wave workgroups (LDS-DMA loader,
        // store-only storer, warm instruction cache), faster than lattice_ws.hip's compute + I/O wave pair at every size
        // (us, ws / wd: N=16, T=150, U=40 16.1 / 15.0; N=32, T=150, U=20 15.1 / 13.7; N=256, T=150, U=40 19.4 / 16.6;
        // N=32, U=50: T=250 22.5 / 21.0, T=500 36.8 / 33.6, T=1000 64.8 / 58.6; N=256, T=500 42.7 / 34.5; T=1500, U=64
        // 93 / 84; profiles/r04_lattice_routes_single_block.txt).
        if (nA == 1 && kern != 1) {
            const hipError_t e = launch_lattice_wd(stream, plain, N);
            if (e != hipErrorNotSupported) { g_last_kernel = 2; return e; }
        }
        if (use_wd) {
            const hipError_t e = launch_lattice_wd(stream, a, N);
            if (e == hipSuccess) { g_last_kernel = 2; return redo_behind(); }
            if (e != hipErrorNotSupported) return e;
        }
        // Everything else that fits: the single-workgroup form of the same kernel (k_lattice_wl: three waves per column
        // block, LDS-DMA loader, hand-written compute blocks, boundary columns through LDS), up to wl_max_blocks() column
        // blocks.  Needs nothing but the planes, so it also serves the callers without flags and rings (the
        // reference-named C entry points, 32-bit compact offsets).  Not when wd is pinned (kern == 2): a pinned A/B run
        // must measure the kernel it names or fall through to ws.
        if (kern != 1 && kern != 2 && nA >= 2) {
            // (by itself: two column blocks always, up to five while one workgroup per sweep leaves CUs idle -- beyond
            //  ~100 utterances lattice_ws.hip's ten waves per workgroup pack the chip better than fifteen; pinned (3):
            //  all it can take, still under RNNT_WL_MAX_BLOCKS -- 0 there means "never chosen", pinned or not)
            const int by_shape = (nA <= 2 || N <= 96) ? wl_max_blocks() : std::min(2, wl_max_blocks());
            const hipError_t e = launch_lattice_wl(stream, plain, N, kern == 3 ? wl_max_blocks() : by_shape);
            if (e != hipErrorNotSupported) { g_last_kernel = 5; return e; }
        }
        const hipError_t e = launch_lattice_ws(stream, plain, N);
        if (e != hipErrorNotSupported) { g_last_kernel = 1; return e; }
    }
#endif
    g_last_kernel = 4;
    return launch_single(stream, plain, N, loader);
}

}  // namespace rnnt
