/*
 * warp_rnnt_amd.h -- C ABI of the MI355X-native RNN-Transducer loss
 * (libwarp_rnnt_amd.so).  Plain pointers and sizes only; every pointer is a
 * DEVICE pointer unless stated otherwise; every call only ENQUEUES work on
 * `stream` (no allocation, no host synchronisation), so the library is
 * re-entrant across streams and threads.  What the library keeps between calls:
 * a launch counter per process and one per device (they only make the hand-over
 * tags of the column-block lattice kernel unique per launch and per graph
 * replay), a sticky per-device word that the gradient kernel sets when the
 * forward/backward guard fires (rnnt_amd_mismatch_flag: diagnostics only), a
 * per-thread status for the void-returning compact entry points
 * (rnnt_amd_compact_last_status), a once-per-process kernel attribute, and a
 * handful of kernel-selection knobs for A/B runs that are read once from the
 * environment -- none of them changes a bit of a result
 * except the log-softmax knobs, and those only its fp32 rounding.  One arithmetic
 * (the reference's) serves every call; nothing a caller can set changes it.
 * The only setter, rnnt_amd_debug_set_lattice_kernel, pins which of several
 * bit-identical kernels runs and exists for tests and A/B timing.  Nothing
 * depends on what an earlier call left in a workspace: scratch contents are
 * unspecified on entry and exit.
 *
 * Sizes every entry point takes (anything else: RNNT_STATUS_INVALID_ARGUMENT
 * before any launch -- the void compact entries report it their way):
 *     1 <= T, 1 <= U, 0 <= N <= 65535        (N: gridDim.y of the gradient kernel)
 *     T * U   <  2^29                         (one utterance's plane of pairs < 4 GiB: 32-bit buffer offsets)
 *     N * T * U < 2^32                        (flat cell index)
 *     dense (N,T,U,V) entries also: U * V < 2^31
 * The reference's own limit is lower: its `int` index arithmetic overflows at
 * N*T*U*V >= 2^31 elements (core.cu:14-24).
 *
 * Part 1 mirrors the reference's own C interface (1ytic/warp-rnnt core.h: all
 * five entry points, padded and compact) so that the reference's bindings can
 * link against this library unchanged.  Part 2 is the native interface
 * the bundled Python host (warp_rnnt/_C.py) uses: it adds a caller-provided
 * workspace so the lattice kernels can run on the diagonal-major layout.
 */
#ifndef WARP_RNNT_AMD_H
#define WARP_RNNT_AMD_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* HIP's stream handle (hip_runtime_api.h: typedef struct ihipStream_t* hipStream_t). */
typedef struct ihipStream_t *rnntStream_t;

/* Replaces core.h:16-22.  Codes 0-4 keep the reference's numbering; because the two
 * gradient kernels and the cost kernel are one fused launch here, a failure of that
 * launch is reported as 2 and codes 3/4 are never produced.  5-7 are new. */
typedef enum {
    RNNT_STATUS_SUCCESS = 0,
    RNNT_STATUS_WARP_FAILED = 1,         /* alpha/beta lattice launch failed */
    RNNT_STATUS_GRADS_BLANK_FAILED = 2,  /* fused gradient+cost launch failed */
    RNNT_STATUS_GRADS_LABEL_FAILED = 3,
    RNNT_STATUS_COSTS_FAILED = 4,
    RNNT_STATUS_INVALID_ARGUMENT = 5,    /* rejected before any launch */
    RNNT_STATUS_PROLOGUE_FAILED = 6,     /* log-softmax / gather / re-layout launch failed */
    RNNT_STATUS_EXPAND_FAILED = 7        /* dense gradient expansion launch failed */
} rnntStatus_t;

/* ------------------------------------------------------------------------
 * Part 1 -- drop-in for the reference C interface
 * ------------------------------------------------------------------------ */

/*
 * Replaces run_warp_rnnt (core.h:29-33, core.cu:372-401): dense layout.
 *   log_probs (N,T,U,V) fp32, labels (N,U-1) int32, xn/yn (N,) int32.
 *   grads (N,T,U,V): MUST be zeroed by the caller (as binding.cpp:58 does); only
 *     the blank/label slots of live cells are written.
 *   counts (N,2U) uint32, alphas/betas (N,T,U) fp32: caller scratch, contents on
 *     return unspecified (alphas/betas hold the lattices in diagonal-major order,
 *     the first N words of counts hold the alpha-side log-likelihoods as fp32 bits).  counts need
 *     not be zeroed.
 *   costs (N,): out.
 * Sizes: see the head of this file (N <= 65535, T*U < 2^29, N*T*U < 2^32; here also U*V < 2^31 for
 *   the fast path -- beyond it the call still runs, on the single-role kernel): status 5 otherwise.
 * Lengths: the reference does not check 1 <= xn[n] <= T, 0 <= yn[n] <= U-1 (binding.cpp:47-51)
 *   and reads out of range when they are violated.  Every entry point of this library checks
 *   them on the device (no host sync): an offending utterance gets costs[n] = NaN and no
 *   gradient (zeros), the rest of the batch is unaffected.
 */
rnntStatus_t run_warp_rnnt(rnntStream_t stream, unsigned int *counts, float *alphas, float *betas,
                           const int *labels, const float *log_probs, float *grads, float *costs,
                           const int *xn, const int *yn, int N, int T, int U, int V, int blank,
                           float fastemit_lambda);

/*
 * Replaces run_warp_rnnt_gather (core.h:35-39, core_gather.cu:359-388): gathered layout,
 *   log_probs and grads (N,T,U,2) with channel 0 = blank, channel 1 = label.
 *   grads is fully written (zeros included); pre-zeroing is allowed but not needed.
 *   Sizes: N <= 65535, T*U < 2^29, N*T*U < 2^32 (status 5 otherwise).  counts / alphas / betas: scratch
 *   as above (from 2^20 cells on alphas / betas are also used to park the gradient channels: they hold no
 *   defined values on return, as in the reference, whose binding never reads them back).
 */
rnntStatus_t run_warp_rnnt_gather(rnntStream_t stream, unsigned int *counts, float *alphas,
                                  float *betas, const float *log_probs, float *grads, float *costs,
                                  const int *xn, const int *yn, int N, int T, int U,
                                  float fastemit_lambda);

/*
 * The reference's compact (ragged packed) entry points -- core.h:41-60, core_compact.cu:360-500 -- under their
 * own names and argument lists, so that pytorch_binding/binding.cpp (:170, :197, :241) links against this
 * library whole.  As there: void return, work is enqueued on the NULL stream, every pointer is a device pointer,
 * memPref / labelPref are the (N,) EXCLUSIVE prefix sums of xn*(yn+1) and yn (binding.cpp:147-160), counts holds
 * at least 2N words.  Not as there: a failure (a launch error, or sizes this library does not take: N > 65535,
 * N*T*U >= 2^32) is not exit(-1) (core.h:7-14) but is reported three ways, because the reference's binding checks
 * none: a line on stderr, costs filled with NaN (so that an unchecked caller sees a NaN loss, not uninitialised
 * memory), and a per-host-thread status returned (and cleared) by rnnt_amd_compact_last_status() -- poll it after
 * run_warp_rnnt_compact if you can.  Lengths with xn < 1 give cost NaN and
 * touch nothing; required_grad == 0 computes betas and costs only and never writes alphas / grads (the reference
 * aliases both to betas then, binding.cpp:192-195).  `alphas` / `betas` are SCRATCH for run_warp_rnnt_compact, not
 * outputs: with required_grad and a launch bound of N*T*U >= 2^20 cells the call parks the two gradient channels in
 * them on its way to the row-major result, below that they hold the lattices in diagonal-major order -- what they hold
 * on return depends on the batch size and is unspecified (binding.cpp:186-199 allocates them with torch::empty and
 * drops them).  A caller that wants alphas / betas uses the padded entries with a batch below 2^20 cells or reads them
 * from the native workspace (rnnt_amd_workspace_size: alphas at offset 0, betas behind, diagonal-major).  Gradients honour the alpha/beta consistency guard of the
 * padded kernels (core_gather.cu:341-354), which the reference's compact kernels lack.
 * The native compact interface further down (rnnt_amd_loss_compact ...) does the same work on the caller's
 * stream with status codes and is what the bundled host code uses.
 */
void run_gather_for_compact(const float *xs, const int *ys, const unsigned int *xn, const unsigned int *yn,
                            float *gather_xs, long *loc, const unsigned int *memPref,
                            const unsigned int *labelPref, unsigned int N, unsigned int T, unsigned int U,
                            unsigned int V, unsigned int blank);
void run_warp_rnnt_compact(unsigned int *counts, float *alphas, float *betas, const float *log_probs, float *grads,
                           float *costs, const unsigned int *xn, const unsigned int *yn,
                           const unsigned int *memPref, const unsigned int *labelPref, unsigned int N,
                           unsigned int T, unsigned int U, float fastemit_lambda, bool required_grad);
void run_scatter_grad_for_compact(const float *grad_cost, const float *gather_grad, const long *loc,
                                  const int *cum_lens, float *scatter_grad, unsigned int STU, unsigned int N,
                                  unsigned int V, unsigned int blank);
rnntStatus_t rnnt_amd_compact_last_status(void);

/* ------------------------------------------------------------------------
 * Part 2 -- native interface (workspace-based, diagonal-major lattice layout)
 * ------------------------------------------------------------------------ */

/* what `input` holds */
enum {
    RNNT_IN_LOG_PROBS_DENSE = 0,    /* (N,T,U,V) log-probabilities + labels       */
    RNNT_IN_LOG_PROBS_GATHERED = 1, /* (N,T,U,2) blank/label log-probabilities    */
    RNNT_IN_LOGITS_DENSE = 2        /* (N,T,U,V) unnormalised logits + labels: log-softmax
                                       and gather are fused, log-probs never materialise */
};
/* what `grads` receives (always d cost[n] / d log_probs, FastEmit included) */
enum {
    RNNT_GRADS_GATHERED = 0,        /* (N,T,U,2) row-major, fully written          */
    RNNT_GRADS_GATHERED_DIAGONAL = 1, /* (N,T,U,2) diagonal-major, fully written; opaque, feed to
                                       rnnt_amd_expand_grads                        */
    RNNT_GRADS_DENSE = 2,           /* (N,T,U,V) row-major, fully written (no pre-zeroing)  */
    RNNT_GRADS_NONE = 3             /* costs only                                   */
};

/* Bytes of device scratch rnnt_amd_loss needs for a problem of this size. */
size_t rnnt_amd_workspace_size(int N, int T, int U);

/* Byte offset, inside that workspace, of the (N,) int32 vector that rnnt_amd_loss fills with 1 where the
 * forward/backward consistency guard fired (core_gather.cu:341-354: the alpha-side and beta-side
 * log-likelihoods differ by more than 1e-3 relative; the reference prints a WARNING from the device,
 * zeroes that utterance's gradients and returns the mean of the two as its cost -- the last two are
 * reproduced, the message is replaced by this flag) or where the lengths were out of range. */
size_t rnnt_amd_workspace_mismatch_offset(int N, int T, int U);

/*
 * The same guard, process-wide and without a read-back of anybody's workspace -- the counterpart of the reference's
 * device-side printf ("WARNING: sample %d [%d, %d] has a forward/backward mismatch %f / %f", core_gather.cu:345-349),
 * which a caller of this library would otherwise lose.  Returns a HOST pointer to eight words of pinned, device-mapped
 * memory belonging to `device` (NULL: device index out of range, or the allocation failed); from this call on, every
 * entry point of the library whose gradient kernel zeroes an utterance's gradients on `device` -- the guard fired, or
 * the lengths were out of range -- also writes there, from the kernel, with no host synchronisation and at no cost when
 * nothing fires:
 *   [0] != 0  something fired since the reader last stored 0 here (the reader clears it; written last)
 *   [1] kind: 1 forward/backward mismatch, 2 lengths out of range     [2] utterance index n in its batch
 *   [3] xn[n]   [4] yn[n]   [5] alpha-side log-likelihood (fp32 bits)   [6] beta[0,0] (fp32 bits)   [7] unused
 * The words are sticky (nothing but the reader resets them) and describe the LAST firing.  Read them whenever convenient:
 * after a synchronisation they are exact; without one they lag by the kernels still in flight.  The first call
 * allocates the table (one hipHostMalloc per process, never freed): do not make it inside a stream capture.  Until the
 * first call the kernels write nothing.  The bundled Python host calls it once per device and turns [0] into one
 * RuntimeWarning per firing (warp_rnnt_amd.last_mismatch()).
 */
volatile unsigned *rnnt_amd_mismatch_flag(int device);

/*
 * The loss: costs (N,) and gradients in one call.
 *   workspace: >= rnnt_amd_workspace_size(N,T,U) bytes, 256-byte aligned, contents unspecified.
 *   labels may be NULL for RNNT_IN_LOG_PROBS_GATHERED or when U == 1.
 *   V/blank are ignored for RNNT_IN_LOG_PROBS_GATHERED.
 *   RNNT_GRADS_DENSE is available for RNNT_IN_LOG_PROBS_DENSE only.
 */
rnntStatus_t rnnt_amd_loss(rnntStream_t stream, void *workspace, int input_kind, const float *input,
                           const int *labels, const int *xn, const int *yn, float *costs, float *grads,
                           int grads_kind, int N, int T, int U, int V, int blank,
                           float fastemit_lambda);

/*
 * Backward of the gather prologue (warp_rnnt/__init__.py:21-24,126): expands
 * RNNT_GRADS_GATHERED_DIAGONAL gradients, scaled by grad_costs[n] (NULL = 1), into a dense
 * (N,T,U,V) tensor that is fully written.  overwrite != 0 selects the dense kernels'
 * "label slot overwrites blank slot" rule instead of scatter-add.
 */
rnntStatus_t rnnt_amd_expand_grads(rnntStream_t stream, const float *grads_diagonal, const int *labels,
                                   const int *xn, const int *yn, const float *grad_costs,
                                   float *dense_grads, int N, int T, int U, int V, int blank,
                                   int overwrite);

/*
 * Compact (ragged packed) layout -- replaces run_gather_for_compact + run_warp_rnnt_compact
 * (core.h:41-54; core_compact.cu:360-436) with this ABI's conventions (status codes, caller's
 * stream, no exit(), no host synchronisation).
 *   xs (STU,V) log-probs, utterance n owning rows [cell_offsets[n], cell_offsets[n+1]) as a
 *   (xn[n], yn[n]+1) row-major block; ys (sum yn,) packed labels; cell_offsets (N+1,) int64 and
 *   label_offsets (N+1,) int32 exclusive prefix sums; Tmax/Umax = max xn / max yn+1 (launch bounds).
 *   grads2 (STU,2) [blank,label] gradients, fully written (NULL = costs only);
 *   loc (STU,) int64 vocabulary index of the label channel per row (NULL = not wanted).
 */
/* Scratch for rnnt_amd_loss_compact with the same N, STU, Tmax, Umax (0 = these sizes are not supported).
 * (The launch bounds are arguments because the hand-over rings of k_lattice_wd live here too, and their
 * size depends on Tmax and Umax.) */
size_t rnnt_amd_workspace_size_compact(int N, int64_t STU, int Tmax, int Umax);
/* One launch for what binding.cpp:139-170 does with a chain of tensor ops: cell_offsets (N+1,) int64 and
 * label_offsets (N+1,) int32 exclusive prefix sums of xn*(yn+1) and yn, and
 * stats[4] (device, int64) = {sum of cells (= STU), sum of yn, max xn, max yn} -- the caller reads
 * stats back once to size the launches.  N <= 65535. */
rnntStatus_t rnnt_amd_compact_offsets(rnntStream_t stream, const int *xn, const int *yn, int N,
                                      int64_t *cell_offsets, int *label_offsets, int64_t *stats);
rnntStatus_t rnnt_amd_loss_compact(rnntStream_t stream, void *workspace, const float *xs, const int *ys,
                                   const int *xn, const int *yn, const int64_t *cell_offsets,
                                   const int *label_offsets, float *costs, float *grads2, int64_t *loc,
                                   int N, int64_t STU, int Tmax, int Umax, int V, int blank,
                                   float fastemit_lambda);

/* The same in ONE call with launch bounds the caller supplies (Tmax >= every xn, Umax >= every yn + 1; n_labels = the
 * number of elements of ys): offsets, maxima and the shape checks stay on the device, nothing is read back, so the call
 * can be captured into a HIP graph and costs no host synchronisation (rnnt_amd_compact_offsets + a read-back +
 * rnnt_amd_loss_compact cost one).  A batch with a length outside the bounds (or < 1 frame), or whose sums are not
 * STU / n_labels, is refused as a whole: costs = NaN, gradients zero, nothing outside the tensors is touched.
 * Workspace: rnnt_amd_workspace_size_compact_bounded(N, STU, Tmax, Umax) bytes, 256-byte aligned. */
size_t rnnt_amd_workspace_size_compact_bounded(int N, int64_t STU, int Tmax, int Umax);
rnntStatus_t rnnt_amd_loss_compact_bounded(rnntStream_t stream, void *workspace, const float *xs, const int *ys,
                                           int64_t n_labels, const int *xn, const int *yn, float *costs,
                                           float *grads2, int64_t *loc, int N, int64_t STU, int Tmax, int Umax,
                                           int V, int blank, float fastemit_lambda);

/* Replaces run_scatter_grad_for_compact (core.h:56-60): (STU,V) d/d log_probs, fully written.
 * cum_lens (N,) int32 inclusive prefix sums of xn*(yn+1) (warp_rnnt/__init__.py:38). */
rnntStatus_t rnnt_amd_compact_scatter_grads(rnntStream_t stream, const float *grad_costs,
                                            const float *grads2, const int64_t *loc, const int *cum_lens,
                                            float *dense_grads, int64_t STU, int N, int V, int blank);

/*
 * Backward of rnnt_amd_loss(RNNT_IN_LOGITS_DENSE -> RNNT_GRADS_GATHERED_DIAGONAL): the gradient
 * w.r.t. the LOGITS, dz[v] = s_n * ( [v==blank] gB + [v==label] gL - softmax(z)[v] (gB+gL) ), one
 * read of the logits and one write of dlogits (dlogits may alias logits).  The log-probabilities
 * and their dense gradient never exist in HBM (the caller-side chain
 * F.log_softmax -> gather -> loss of benchmark.py:65-70 costs 28V bytes per cell, this 12V+8).
 */
rnntStatus_t rnnt_amd_logits_backward(rnntStream_t stream, const float *logits, const int *labels,
                                      const float *grads_diagonal, const float *grad_costs, float *dlogits,
                                      int N, int T, int U, int V, int blank);

/* Row-wise log-softmax over the last axis; out may alias x. */
rnntStatus_t rnnt_amd_log_softmax(rnntStream_t stream, const float *x, float *out, int64_t rows, int V);

/* Backward of the row-wise log-softmax: grad_in = grad_out - exp(out) * rowsum(grad_out), where
 * `out` holds the log-probabilities the forward produced.  grad_in may alias grad_out. */
rnntStatus_t rnnt_amd_log_softmax_backward(rnntStream_t stream, const float *grad_out, const float *out,
                                           float *grad_in, int64_t rows, int V);

/* (N,T,U,V) log-probs -> (N,T,U,2) row-major gathered log-probs (the tensor the reference's
 * wrapper hands to the native op, __init__.py:122-126).  `workspace` as for rnnt_amd_loss. */
rnntStatus_t rnnt_amd_gather(rnntStream_t stream, const float *log_probs, const int *labels,
                             float *gathered, int N, int T, int U, int V, int blank);

/* Diagnostics (used by tools/lattice_probe.py): re-run only the alpha/beta sweep on the
 * (blank,label) pairs a previous rnnt_amd_loss call left in `workspace` -- they survive a call with
 * RNNT_GRADS_GATHERED_DIAGONAL only (every other kind produces its gradients in their place). */
rnntStatus_t rnnt_amd_debug_lattice_only(rnntStream_t stream, void *workspace, const int *xn,
                                         const int *yn, int N, int T, int U);

/* Diagnostics (bench.py's roofline entry of the gather kernel): only the first step of
 * rnnt_amd_loss(RNNT_IN_LOG_PROBS_DENSE) -- dense log-probs -> diagonal-major (blank,label) pairs in `workspace`. */
rnntStatus_t rnnt_amd_debug_gather_only(rnntStream_t stream, void *workspace, const float *log_probs,
                                        const int *labels, int N, int T, int U, int V, int blank);

/* Diagnostics (tools/wd_soak.py, tests/test_gpu_wd.py): byte offset, inside the workspace, of the (2N,) int32 flags
 * k_lattice_wd leaves for the single-workgroup kernel launched behind it: flags[2n+dir] & 2 means a hand-over
 * between column blocks of sweep `dir` (0 alpha, 1 beta) of utterance n timed out and the sweep was redone. */
size_t rnnt_amd_debug_redo_offset(int N, int T, int U);

/*
 * DEBUG / A-B ONLY.  Which KERNEL runs the sweeps where several can (speed only: they put the same instructions on the
 * chain and produce the same bits, tests/test_gpu_wd.py; the arithmetic is always the reference's fp32 log-sum-exp per
 * cell, core_gather.cu:22-35,106-126 -- until version 105 a second arithmetic could be selected with
 * rnnt_amd_set_lattice; it is gone, and so are rnnt_amd_loss_ex / rnnt_amd_loss_compact_ex that chose it per call):
 *   0 by shape   (default)
 *   1 ws         all column blocks of a sweep in one workgroup, one compute + one I/O wave each (csrc/lattice_ws.hip; U <= 512)
 *   2 wd         one workgroup per 64-column block, boundary columns through L2 rings (csrc/lattice_wd.hip; any U)
 *   3 wl         the single-workgroup form of wd (k_lattice_wl: three waves per column block, boundary columns through
 *                LDS), wherever the workgroup's LDS holds the lattice's column blocks
 * One process-wide atomic, read once per call; initial value from the environment variable
 * RNNT_DEBUG_LATTICE_KERNEL (ws | wd | wl).  Returns the previous setting, or -1 for an unknown value.  Nothing in the
 * product sets it; not part of any contract.
 */
int rnnt_amd_debug_set_lattice_kernel(int kernel);
int rnnt_amd_debug_get_lattice_kernel(void);

/* Diagnostics (bench.py's `lattice_kernel`, tests): the lattice kernel the calling thread's last loss call launched --
 * 1 lattice_ws (one workgroup per sweep), 2 lattice_wd (one per column block), 4 the single-role kernel of lattice.hip
 * (reference layouts, stripes), 5 lattice_wl (one workgroup per sweep, the wave roles of lattice_wd); 0 before the
 * first call (3 was the retired probability-domain kernel). */
int rnnt_amd_debug_last_lattice_kernel(void);

/* Library version, for the host-side loader. */
int rnnt_amd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* WARP_RNNT_AMD_H */
