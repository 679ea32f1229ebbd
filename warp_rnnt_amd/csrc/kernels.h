// Internal launch interface between the C ABI (api.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rnnt {

struct LatticeArgs {
    const float* lp;      // log-probs in the layout named by the loader
    const int* labels;    // (N,U-1), LOAD_DENSE only
    const int* xn;        // (N,) frames per utterance
    const int* yn;        // (N,) labels per utterance
    float* alphas;        // (N,T,U) diagonal-major scratch (out)
    float* betas;         // (N,T,U) diagonal-major scratch (out)
    float* ll;            // (N,) alpha-side log-likelihood alpha[T-1,U-1]+lpB[T-1,U-1] (out)
    int T, U, V, blank;   // V/blank: LOAD_DENSE only
    const int64_t* offs;  // compact layout: (N+1,) cell offset of each utterance's (T_n, U_n) plane;
                          // nullptr = padded (N,T,U) planes
    int* redo;            // (2N,) [2n+dir]: written by the probability-domain kernel (non-zero = inputs outside the
                          // range it can represent, or a lost hand-over), read by the log-domain kernel launched
                          // behind it (0 = nothing to do); nullptr = log-domain kernel only
    int* queue;           // work-item counter of the probability-domain kernel; MUST be redo + 2N (zeroed together);
                          // queue[1] is its launch counter (never zeroed: any start value will do)
    unsigned long long* mail;  // its hand-over rings between column blocks (pd_mail_bytes), needed when U > 64
    int mail_blocks;      // filled in by launch_lattice_pd
    unsigned epoch;       // filled in by launch_lattice_pd
    const unsigned* offs32;  // compact layout with the reference's 32-bit offsets (run_warp_rnnt_compact); used when
                             // offs is null
    int beta_only;        // compact shim, required_grad = false: the alpha sweep is skipped (its buffer aliases betas)
    int route;            // LatticeRoute of this call (diagonal-major loader only); 0 = ROUTE_AUTO
};

// Which arithmetic sweeps the lattice of a diagonal-major call.  Per call: api.hip copies the process-wide setting
// (rnnt_amd_set_lattice; initial value from the environment variable RNNT_LATTICE) into LatticeArgs::route.
enum LatticeRoute : int {
    ROUTE_AUTO = 0,       // probability domain where it is the faster kernel (long lattices), log domain elsewhere
    ROUTE_LOGDOMAIN = 1,  // the reference's arithmetic: lse per cell in fp32 (lattice_ws.hip / lattice.hip)
    ROUTE_PD = 2          // probability domain (lattice_pd.hip) wherever it is supported (padded layout, U <= 512)
};
int lattice_route();              // current setting
int set_lattice_route(int route); // returns the previous setting, or -1 for an unknown value (nothing changes)
// true when launch_lattice may hand this shape to the probability-domain kernel under SOME route: the workspace
// then reserves its hand-over rings (a function of the shape only, so that the size never depends on the setting)
bool pd_shape_supported(int T, int U);
__host__ __device__ inline bool is_compact(const LatticeArgs& a) { return a.offs || a.offs32; }
__device__ inline size_t compact_base(const LatticeArgs& a, int n) { return a.offs ? (size_t)a.offs[n] : (size_t)a.offs32[n]; }

struct GradArgs {
    const float* lp;      // as LatticeArgs
    const int* labels;
    const int* xn;
    const int* yn;
    const float* alphas;  // diagonal-major
    const float* betas;   // diagonal-major
    const float* ll;      // (N,) from the lattice kernel
    float* grads;         // layout named by the writer
    float* costs;         // (N,)
    int* mismatch;        // (N,) optional: 1 where the alpha/beta guard fired
    int T, U, V, blank;
    float fastemit_lambda;
    const int64_t* offs;  // as LatticeArgs
    const unsigned* offs32;
};
__host__ __device__ inline bool is_compact(const GradArgs& a) { return a.offs || a.offs32; }
__device__ inline size_t compact_base(const GradArgs& a, int n) { return a.offs ? (size_t)a.offs[n] : (size_t)a.offs32[n]; }

// gradients + costs + guard AND the dense (N,T,U,V) rows in one launch (a.lp = diagonal-major pairs, a.labels, a.V and
// a.blank of the dense tensor; a.grads unused): for lattices whose planes sit in L2
hipError_t launch_grads_dense(hipStream_t stream, const GradArgs& a, float* dense, int N);
hipError_t launch_lattice(hipStream_t stream, const LatticeArgs& a, int N, int loader);
// wave-specialised log-domain variant (diagonal-major loader only); hipErrorNotSupported when U > 512.
// With a.redo set only the (utterance, direction) pairs flagged there are swept.
hipError_t launch_lattice_ws(hipStream_t stream, const LatticeArgs& a, int N);
// probability-domain variant (lattice_pd.hip; diagonal-major loader; needs a.redo, a.queue and -- for U > 64 --
// a.mail of pd_mail_bytes(N,T,U) bytes); hipErrorNotSupported when they are missing
hipError_t launch_lattice_pd(hipStream_t stream, const LatticeArgs& a, int N);
size_t pd_mail_bytes(int N, int T, int U);
// distributed log-domain variant (lattice_wd.hip; diagonal-major loader, padded or 64-bit compact; one workgroup per
// 64-column block, any U); needs a.redo, a.queue and -- for U > 64 -- a.mail of wd_mail_bytes(N,T,U) bytes;
// hipErrorNotSupported when they are missing.  Bit-identical to launch_lattice_ws.
hipError_t launch_lattice_wd(hipStream_t stream, const LatticeArgs& a, int N);
size_t wd_mail_bytes(int N, int T, int U);
// ... and its single-workgroup form (lattice_wd.hip: k_lattice_wl): all column blocks of a sweep as waves of one
// workgroup, boundary columns through LDS; needs nothing but the planes (no flags, no rings), padded or compact with
// either offset width, honours a.redo and a.beta_only.  hipErrorNotSupported beyond wl_max_blocks() column blocks.
// Bit-identical to the other two.
hipError_t launch_lattice_wl(hipStream_t stream, const LatticeArgs& a, int N, int max_blocks);
int wl_max_blocks();   // what launch_lattice lets it take by itself; set_logdomain_kernel(3) lets it take all it can (5)
// what a workspace reserves for the hand-over rings of either kernel (a function of the shape only)
inline size_t lattice_mail_bytes(int N, int T, int U) {
    const size_t p = pd_mail_bytes(N, T, U), w = wd_mail_bytes(N, T, U);
    return p > w ? p : w;
}
// In front of every launch of a kernel that hands boundary columns over through L2 rings (lattice_pd.hip owns the
// per-device launch counter): clears n_flags words at `flags` (redo flags + queue head), stores the next value of the
// launch counter at flags[n_flags] and zeroes ring_bytes (a multiple of 16) at `rings`.
hipError_t launch_ring_prepare(hipStream_t stream, int* flags, int n_flags, void* rings, size_t ring_bytes);
unsigned next_launch_epoch();     // host part of the launch epoch: random start, +1 per call
// Which kernel serves the log-domain route where several can (same bits either way): 0 = by shape, 1 = single workgroup
// per sweep (lattice_ws.hip), 2 = one workgroup per column block (lattice_wd.hip), 3 = its single-workgroup form
// (k_lattice_wl) wherever it fits.  Initial value from the environment variable RNNT_LOGDOMAIN_KERNEL=ws|wd|wl.
int last_lattice_kernel();        // what the calling thread's last launch_lattice ran: 1 ws, 2 wd, 3 pd, 4 single-role, 5 wl (0 none yet)
int logdomain_kernel();
int set_logdomain_kernel(int k);  // 0 by shape, 1 ws, 2 wd, 3 wl (single-workgroup form of wd); returns the previous setting, or -1
hipError_t launch_grads(hipStream_t stream, const GradArgs& a, int N, int loader, int writer);

// prologue / epilogue streaming kernels
hipError_t launch_log_softmax(hipStream_t stream, const float* x, float* out, int64_t rows, int V);
hipError_t launch_log_softmax_backward(hipStream_t stream, const float* dy, const float* y, float* dx,
                                       int64_t rows, int V);
hipError_t launch_gather(hipStream_t stream, const float* log_probs, const int* labels, float* out2,
                         int N, int T, int U, int V, int blank, bool skewed);
hipError_t launch_reskew(hipStream_t stream, const float* lp2_rowmajor, float* ws2, int N, int T, int U);
// diagonal-major pairs (b == nullptr: float2 plane at a; else two float planes) -> row-major (N,T,U,2)
hipError_t launch_unskew(hipStream_t stream, const float* a, const float* b, float* out2_rowmajor, int N, int T, int U);
hipError_t launch_split_pairs(hipStream_t stream, const float* pairs, float* a, float* b, size_t cells);
hipError_t launch_expand_split(hipStream_t stream, const float* ga_skewed, const float* gb_skewed, const int* labels,
                               const int* xn, const int* yn, float* dense, int N, int T, int U, int V, int blank);
// compact (ragged packed) layout, core_compact.cu:403-436,456-484
// reference-layout compact helpers (core.h:41-60 shims): row-major packed pairs + loc; costs from betas alone
hipError_t launch_gather_compact_rowmajor(hipStream_t stream, const float* xs, const int* ys, const unsigned* xn,
                                          const unsigned* yn, float* gather_xs, int64_t* loc, const unsigned* mem_pref,
                                          const unsigned* label_pref, unsigned N, unsigned T, unsigned U, unsigned V,
                                          unsigned blank);
// the staged form of run_warp_rnnt_compact: row-major packed pairs <-> per-utterance diagonal-major planes, 32-bit prefixes
hipError_t launch_reskew_compact32(hipStream_t stream, const float* pairs_rowmajor, float* pairs_diagonal,
                                   const unsigned* xn, const unsigned* yn, const unsigned* mem_pref, unsigned N,
                                   unsigned Tmax, unsigned Umax);
hipError_t launch_unskew_compact32(hipStream_t stream, const float* a_diagonal, const float* b_diagonal,
                                   float* pairs_rowmajor, const unsigned* xn, const unsigned* yn, const unsigned* mem_pref,
                                   unsigned N, unsigned Tmax, unsigned Umax);
hipError_t launch_split_pairs_compact32(hipStream_t stream, const float* pairs, float* a, float* b, const unsigned* xn,
                                        const unsigned* yn, const unsigned* mem_pref, unsigned N, size_t cells_bound);
hipError_t launch_costs_from_betas(hipStream_t stream, const float* betas, const unsigned* mem_pref, const int* xn,
                                   const int* yn, float* costs, int N);
// launch bounds and tensor sizes the caller vouches for, checked on the device (rnnt_amd_loss_compact_bounded):
// xn_checked (N,) receives xn, or zeros when any length is outside [1, Tmax] x [0, Umax-1] or the totals differ
struct CompactBounds {
    int* xn_checked;
    int64_t STU, n_labels;
    int Tmax, Umax;
};
hipError_t launch_compact_offsets(hipStream_t stream, const int* xn, const int* yn, int N, int64_t* cell_offs,
                                  int* label_offs, int64_t* stats, const CompactBounds* bounds = nullptr);
// (stats[4], written only with bounds: 1 = the batch was refused)
hipError_t launch_zero_if_refused(hipStream_t stream, const int64_t* refused, float* grads2, size_t cells);
hipError_t launch_gather_compact(hipStream_t stream, const float* xs, const int* ys, const int* xn,
                                 const int* yn, const int64_t* offs, const int* label_offs, float* ws2,
                                 int64_t* loc, int N, int Tmax, int Umax, int V, int blank, int64_t STU);
hipError_t launch_scatter_compact(hipStream_t stream, const float* grad_cost, const float* grads2,
                                  const int64_t* loc, const int* cum_lens, float* out, int64_t STU, int N,
                                  int V, int blank);
hipError_t launch_log_softmax_gather_skewed(hipStream_t stream, const float* logits, const int* labels,
                                            float* ws2, int N, int T, int U, int V, int blank);
hipError_t launch_logits_backward(hipStream_t stream, const float* logits, const int* labels,
                                  const float* g2_diagonal, const float* scale, float* dlogits, int N, int T,
                                  int U, int V, int blank);
hipError_t launch_expand(hipStream_t stream, const float* g2_skewed, const int* labels,
                         const int* xn, const int* yn, const float* scale, float* dense, int N,
                         int T, int U, int V, int blank, int overwrite_mode);

}  // namespace rnnt
