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
    int* redo;            // (2N,) [2n+dir]: set by k_lattice_wd when a hand-over between column blocks timed out (bit 1),
                          // read by the single-workgroup kernel launched behind it (0 = nothing to do); nullptr = the
                          // kernels that need no flags only
    int* queue;           // work-item counter of k_lattice_wd; MUST be redo + 2N (zeroed together);
                          // queue[1] is its launch counter (never zeroed: any start value will do)
    unsigned long long* mail;  // its hand-over rings between column blocks (wd_mail_bytes), needed when U > 64
    unsigned epoch;       // filled in by launch_lattice_wd
    const unsigned* offs32;  // compact layout with the reference's 32-bit offsets (run_warp_rnnt_compact); used when
                             // offs is null
    int beta_only;        // compact shim, required_grad = false: the alpha sweep is skipped (its buffer aliases betas)
    int prepared;         // the rings and flags of this call have been prepared by the kernel in front (RingPrep below):
                          // launch_lattice_wd does not launch k_prepare
};

// What k_prepare does in front of every launch of the ring kernel, as a parcel that the PRODUCER of the call's pair plane
// -- the gather / re-layout kernel that runs in front of the sweeps anyway -- can carry out at the tail of its own
// workgroups (round 6: one 5 us launch less on the dense and gathered routes; the routes whose producer does not take
// a parcel keep the launch).  flags == nullptr: nothing to do.
struct RingPrep {
    int* flags;           // n_flags words cleared (redo flags, queue head); flags[n_flags] receives the next value of ...
    int n_flags;
    unsigned* counter;    // ... the device's launch counter (device address of lattice_wd.hip's g_launch_counter)
    uint4* rings;         // ring_vec 16-byte words zeroed
    size_t ring_vec;
};
// Fills `prep` and returns true when launch_lattice(stream, a, N, loader) is going to run the ring kernel on a.redo /
// a.queue / a.mail -- the caller hands the parcel to its producer and sets a.prepared; false (prep untouched, all zero):
// no ring kernel, or the counter's address is not available for this stream's device -- launch_lattice prepares itself.
bool lattice_ring_prep(hipStream_t stream, const LatticeArgs& a, int N, int loader, RingPrep* prep);

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
    unsigned* sticky;     // the device's sticky diagnostics words (filled in by launch_grads / launch_grads_dense)
};
// Eight words of pinned host memory per device that the gradient kernels write when a forward/backward guard fires
// (grads_cell.h: report_guard).  mismatch_words(device, true) allocates the table on first use (one hipHostMalloc per
// process, never freed; not during stream capture); mismatch_words_of(stream) = the words of the stream's device, or
// nullptr while nobody has asked for them.
unsigned* mismatch_words(int device, bool allocate);
unsigned* mismatch_words_of(hipStream_t stream);
__host__ __device__ inline bool is_compact(const GradArgs& a) { return a.offs || a.offs32; }
__device__ inline size_t compact_base(const GradArgs& a, int n) { return a.offs ? (size_t)a.offs[n] : (size_t)a.offs32[n]; }

// gradients + costs + guard AND the dense (N,T,U,V) rows in one launch (a.lp = diagonal-major pairs, a.labels, a.V and
// a.blank of the dense tensor; a.grads unused): for lattices whose planes sit in L2
hipError_t launch_grads_dense(hipStream_t stream, const GradArgs& a, float* dense, int N);
hipError_t launch_lattice(hipStream_t stream, const LatticeArgs& a, int N, int loader);
// ONE arithmetic -- the reference's: one fp32 lse per cell in its operation order -- and several kernels that put the same
// instructions on the chain and give the same bits (tests/test_gpu_wd.py); launch_lattice picks by shape.
// lattice_ws.hip: compute + I/O wave pairs, all column blocks of a sweep in one workgroup (diagonal-major loader only);
// hipErrorNotSupported when U > 512.  With a.redo set only the (utterance, direction) pairs flagged there are swept.
hipError_t launch_lattice_ws(hipStream_t stream, const LatticeArgs& a, int N);
// lattice_wd.hip: one three-wave workgroup per 64-column block, boundary columns through L2 rings (diagonal-major loader,
// padded or 64-bit compact; any U); needs a.redo, a.queue and -- for U > 64 -- a.mail of wd_mail_bytes(N,T,U) bytes;
// hipErrorNotSupported when they are missing.
hipError_t launch_lattice_wd(hipStream_t stream, const LatticeArgs& a, int N);
int device_cus(hipStream_t stream);   // compute units of the stream's device (csrc/lattice.hip)
size_t wd_mail_bytes(int N, int T, int U);
// ... and its single-workgroup form (lattice_wd.hip: k_lattice_wl): all column blocks of a sweep as waves of one
// workgroup, boundary columns through LDS; needs nothing but the planes (no flags, no rings), padded or compact with
// either offset width, honours a.redo and a.beta_only.  hipErrorNotSupported beyond max_blocks (<= 5) column blocks.
hipError_t launch_lattice_wl(hipStream_t stream, const LatticeArgs& a, int N, int max_blocks);
int wl_max_blocks();   // what launch_lattice lets it take by itself (RNNT_WL_MAX_BLOCKS, default 5)
// what a workspace reserves for the hand-over rings (a function of the shape only)
inline size_t lattice_mail_bytes(int N, int T, int U) { return wd_mail_bytes(N, T, U); }
// In front of every launch of the kernel that hands boundary columns over through L2 rings (lattice_wd.hip owns the
// per-device launch counter): clears n_flags words at `flags` (redo flags + queue head), stores the next value of the
// launch counter at flags[n_flags] and zeroes ring_bytes (a multiple of 16) at `rings`.
hipError_t launch_ring_prepare(hipStream_t stream, int* flags, int n_flags, void* rings, size_t ring_bytes);
// the parcel for a launch of the ring kernel on `a` (flags, queue and rings as launch_lattice_wd would prepare them), or
// false when there is nothing to prepare / the launch counter's address cannot be had for the stream's device
bool wd_ring_prep(hipStream_t stream, const LatticeArgs& a, int N, RingPrep* prep);
unsigned next_launch_epoch();     // host part of the launch epoch: random start, +1 per call
// DEBUG / A-B ONLY: pin the kernel where several can serve (same bits whichever runs): 0 = by shape, 1 = lattice_ws.hip,
// 2 = lattice_wd.hip, 3 = k_lattice_wl wherever it fits.  Initial value from the environment variable
// RNNT_DEBUG_LATTICE_KERNEL=ws|wd|wl.  One process-wide atomic; nothing in the product sets it.
int last_lattice_kernel();        // what the calling thread's last launch_lattice ran: 1 ws, 2 wd, 4 single-role, 5 wl (0 none yet)
int lattice_kernel_override();
int set_lattice_kernel_override(int k);  // returns the previous setting, or -1 for an unknown value (nothing changes)
hipError_t launch_grads(hipStream_t stream, const GradArgs& a, int N, int loader, int writer);

// prologue / epilogue streaming kernels
hipError_t launch_log_softmax(hipStream_t stream, const float* x, float* out, int64_t rows, int V);
hipError_t launch_log_softmax_backward(hipStream_t stream, const float* dy, const float* y, float* dx,
                                       int64_t rows, int V);
hipError_t launch_gather(hipStream_t stream, const float* log_probs, const int* labels, float* out2,
                         int N, int T, int U, int V, int blank, bool skewed, const RingPrep* prep = nullptr);
hipError_t launch_reskew(hipStream_t stream, const float* lp2_rowmajor, float* ws2, int N, int T, int U,
                         const RingPrep* prep = nullptr);
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
