// C ABI of libwarp_rnnt_amd.so (declared in include/warp_rnnt_amd.h).
// Host-side orchestration only: argument checks, workspace carving, launches.
#include "../../include/warp_rnnt_amd.h"

#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

using namespace rnnt;

namespace {

constexpr size_t ALIGN = 256;
// Lattices from this many cells on take the multi-pass forms (stage, sweep on the tuned kernels, turn the layout
// through LDS tiles); smaller ones are launch-bound and keep the single-kernel forms (tools/cabi_probe.py: N=16, T=150,
// U=40: 24 us in two launches against 35 in five; N=16, T=1500, U=300: 500 against 274)
constexpr size_t STAGED_FROM_CELLS = (size_t)1 << 20;
inline size_t align_up(size_t x) { return (x + ALIGN - 1) / ALIGN * ALIGN; }
// Dense gradients: up to this many lattice cells the gradient kernel and the expansion run as ONE launch whose row writer
// reads alpha / beta / pairs itself (scattered reads of planes that sit in L2 / the Infinity Cache); above, two streaming
// launches.  One launch / two (tools/dense_rate.py, profiles/r04_grads_dense_ab.txt; us per call of the dense entry):
// T=150, U=40, V=28: N=16 28.2 / 30.7, N=32 35.0 / 37.6, N=64 46.2 / 48.2, N=128 67.0 / 68.7, N=256 114 / 113;
// T=500, U=100, V=50: N=4 ... 32 (0.2 M ... 1.6 M cells) equal within 0.5 us.  RNNT_DENSE_ONE_LAUNCH_CELLS overrides
// (0: never), for A/B runs.
inline size_t dense_in_one_launch_cells() {
    static const size_t v = ab_getenv("RNNT_DENSE_ONE_LAUNCH_CELLS") ? (size_t)atoll(ab_getenv("RNNT_DENSE_ONE_LAUNCH_CELLS"))
                                                                  : ((size_t)1 << 20);
    return v;
}

struct Workspace {
    float* alphas;
    float* betas;
    float* ws2;      // diagonal-major (blank,label) pairs; later the gathered grads
    float* ll;
    int* mismatch;
    int* redo;       // (2N + 2,) redo flags of k_lattice_wd, its work-item counter, the launch counter's value
    unsigned long long* mail;   // boundary-column rings of k_lattice_wd (kernels.h)
};

size_t carve(void* base, int N, int T, int U, Workspace* w) {
    const size_t cells = (size_t)N * T * U;
    size_t off = 0;
    char* p = static_cast<char*>(base);
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return p ? p + o : nullptr; };
    float* alphas = reinterpret_cast<float*>(take(cells * sizeof(float)));
    float* betas = reinterpret_cast<float*>(take(cells * sizeof(float)));
    float* ws2 = reinterpret_cast<float*>(take(cells * 2 * sizeof(float)));
    float* ll = reinterpret_cast<float*>(take((size_t)N * sizeof(float)));
    int* mismatch = reinterpret_cast<int*>(take((size_t)N * sizeof(int)));
    int* redo = reinterpret_cast<int*>(take(((size_t)N * 2 + 2) * sizeof(int)));   // flags, queue head, launch counter
    // (reserved by SHAPE, never by the current route: the size of a workspace must not depend on a setting)
    unsigned long long* mail = reinterpret_cast<unsigned long long*>(take(lattice_mail_bytes(N, T, U)));
    if (w) *w = Workspace{alphas, betas, ws2, ll, mismatch, redo, mail};
    return off;
}

bool dims_ok(int N, int T, int U) {
    if (N < 0 || T < 1 || U < 1) return false;
    if (N > 65535) return false;                               // gridDim.y of the gradient kernel
    if ((int64_t)T * U >= (int64_t)1 << 29) return false;      // per-utterance plane < 4 GiB of float2 (buffer descriptors)
    if ((int64_t)N * T * U >= (int64_t)1 << 32) return false;  // flat cell index is 32-bit
    return true;
}

}  // namespace

extern "C" {

int rnnt_amd_version(void) { return 106; }

int rnnt_amd_debug_set_lattice_kernel(int kernel) { return set_lattice_kernel_override(kernel); }

int rnnt_amd_debug_get_lattice_kernel(void) { return lattice_kernel_override(); }

int rnnt_amd_debug_last_lattice_kernel(void) { return last_lattice_kernel(); }

volatile unsigned* rnnt_amd_mismatch_flag(int device) { return mismatch_words(device, true); }

size_t rnnt_amd_workspace_size(int N, int T, int U) {
    if (!dims_ok(N, T, U)) return 0;
    return carve(nullptr, N, T, U, nullptr);
}

size_t rnnt_amd_workspace_mismatch_offset(int N, int T, int U) {
    if (!dims_ok(N, T, U)) return 0;
    Workspace w;
    carve(reinterpret_cast<void*>(ALIGN), N, T, U, &w);   // any non-null base: only the offset is wanted
    return reinterpret_cast<uintptr_t>(w.mismatch) - ALIGN;
}

size_t rnnt_amd_debug_redo_offset(int N, int T, int U) {
    if (!dims_ok(N, T, U)) return 0;
    Workspace w;
    carve(reinterpret_cast<void*>(ALIGN), N, T, U, &w);
    return reinterpret_cast<uintptr_t>(w.redo) - ALIGN;
}

// The reference-named entry points (core.h:29-39) own no workspace: everything they need has to come out of the buffers
// the caller's binding allocates (binding.cpp:58-75) -- and does: the tuned kernels want the (blank,label) pairs in the
// diagonal-major layout, which is exactly one (N,T,U,2) plane, and `grads` is at least that large and not yet written.
// So the pairs are staged in `grads`, the sweeps run on them (alphas / betas are the caller's scratch already, in the
// diagonal-major layout since round 1), and the gradient kernel -- which reads the log-probs from the caller's input
// again -- overwrites the staging area with the result.  Until round 4 these entries ran the single-role kernel with
// per-lane row-major / dense loads: 0.50 / 2.67 ms at N=16, T=1500, U=300 (V=50), now 0.27 / 0.64
// (tools/cabi_probe.py, profiles/r04_cabi_probe.txt).
rnntStatus_t run_warp_rnnt(rnntStream_t stream, unsigned int* counts, float* alphas, float* betas,
                           const int* labels, const float* log_probs, float* grads, float* costs,
                           const int* xn, const int* yn, int N, int T, int U, int V, int blank,
                           float fastemit_lambda) {
    if (!dims_ok(N, T, U) || V < 1 || blank < 0 || blank >= V) return RNNT_STATUS_INVALID_ARGUMENT;
    if (N == 0) return RNNT_STATUS_SUCCESS;
    float* ll = reinterpret_cast<float*>(counts);   // (N,2U) uint32 scratch: first N words reused
    const size_t cells = (size_t)N * T * U;
    const bool stage = V >= 2 && (U == 1 || labels) && (int64_t)U * V < ((int64_t)1 << 31) && cells >= STAGED_FROM_CELLS;
    if (stage) {
        // dense (N,T,U,V) grads, zeroed by the caller: its head is the staging area -- pairs, then (room permitting:
        // V >= 4 or so) the flags and hand-over rings of the distributed lattice kernel -- and is zeroed again before the
        // gradient kernel writes its two slots per cell
        char* base = reinterpret_cast<char*>(grads);
        const size_t avail = cells * (size_t)V * sizeof(float);
        size_t off = align_up(cells * 2 * sizeof(float));
        int* redo = nullptr;
        unsigned long long* mail = nullptr;
        const size_t flag_bytes = align_up(((size_t)N * 2 + 2) * sizeof(int)), ring_bytes = lattice_mail_bytes(N, T, U);
        if (reinterpret_cast<uintptr_t>(base) % ALIGN == 0 && off + flag_bytes + align_up(ring_bytes) <= avail) {
            redo = reinterpret_cast<int*>(base + off);
            off += flag_bytes;
            mail = reinterpret_cast<unsigned long long*>(base + off);
            off += align_up(ring_bytes);
        }
        LatticeArgs la{grads, nullptr, xn, yn, alphas, betas, ll, T, U, 2, 0, nullptr, redo, redo ? redo + 2 * N : nullptr, mail};
        RingPrep prep{nullptr, 0, nullptr, nullptr, 0};
        la.prepared = lattice_ring_prep(stream, la, N, LOAD_SKEWED, &prep) ? 1 : 0;
        if (launch_gather(stream, log_probs, labels, grads, N, T, U, V, blank, true, &prep) != hipSuccess)
            return RNNT_STATUS_WARP_FAILED;
        if (launch_lattice(stream, la, N, LOAD_SKEWED) != hipSuccess) return RNNT_STATUS_WARP_FAILED;
        // gradient pairs in place of the log-prob pairs (coalesced both ways), parked in the caller's alphas / betas --
        // dead by then -- and expanded from there into whole dense rows, zeros included: the staging area is
        // overwritten with the rest, and nothing depends on the caller's zero-fill any more (the slot writes of the
        // direct form below cost 0.8 ms at N=16, T=1500, U=300, V=50, this 0.37)
        const bool split_ok = reinterpret_cast<uintptr_t>(grads) % 16 == 0 && reinterpret_cast<uintptr_t>(alphas) % 8 == 0 &&
                              reinterpret_cast<uintptr_t>(betas) % 8 == 0;
        if (split_ok) {
            GradArgs gs{grads, nullptr, xn, yn, alphas, betas, ll, grads, costs, nullptr, T, U, 2, 0, fastemit_lambda};
            if (launch_grads(stream, gs, N, LOAD_SKEWED, WRITE_SKEWED2) != hipSuccess) return RNNT_STATUS_GRADS_BLANK_FAILED;
            if (launch_split_pairs(stream, grads, alphas, betas, cells) != hipSuccess) return RNNT_STATUS_GRADS_LABEL_FAILED;
            if (launch_expand_split(stream, alphas, betas, labels, xn, yn, grads, N, T, U, V, blank) != hipSuccess)
                return RNNT_STATUS_GRADS_LABEL_FAILED;
            return RNNT_STATUS_SUCCESS;
        }
        if (hipMemsetAsync(grads, 0, off < avail ? off : avail, stream) != hipSuccess) return RNNT_STATUS_WARP_FAILED;
    } else {
        LatticeArgs la{log_probs, labels, xn, yn, alphas, betas, ll, T, U, V, blank};
        if (launch_lattice(stream, la, N, LOAD_DENSE) != hipSuccess) return RNNT_STATUS_WARP_FAILED;
    }
    GradArgs ga{log_probs, labels, xn, yn, alphas, betas, ll, grads, costs, nullptr,
                T, U, V, blank, fastemit_lambda};
    if (launch_grads(stream, ga, N, LOAD_DENSE, WRITE_DENSE_SLOTS) != hipSuccess)
        return RNNT_STATUS_GRADS_BLANK_FAILED;
    return RNNT_STATUS_SUCCESS;
}

rnntStatus_t run_warp_rnnt_gather(rnntStream_t stream, unsigned int* counts, float* alphas,
                                  float* betas, const float* log_probs, float* grads, float* costs,
                                  const int* xn, const int* yn, int N, int T, int U,
                                  float fastemit_lambda) {
    if (!dims_ok(N, T, U)) return RNNT_STATUS_INVALID_ARGUMENT;
    if (N == 0) return RNNT_STATUS_SUCCESS;
    float* ll = reinterpret_cast<float*>(counts);
    const size_t cells = (size_t)N * T * U;
    const bool staged = cells >= STAGED_FROM_CELLS && reinterpret_cast<uintptr_t>(grads) % 16 == 0 &&
                        reinterpret_cast<uintptr_t>(alphas) % 8 == 0 && reinterpret_cast<uintptr_t>(betas) % 8 == 0;
    if (staged) {
        // (N,T,U,2) grads = one pair plane: staging area first, result afterwards.  No room for hand-over rings here:
        // the sweeps run on the single-workgroup log-domain kernel.
        if (launch_reskew(stream, log_probs, grads, N, T, U) != hipSuccess) return RNNT_STATUS_WARP_FAILED;
        LatticeArgs la{grads, nullptr, xn, yn, alphas, betas, ll, T, U, 2, 0};
        if (launch_lattice(stream, la, N, LOAD_SKEWED) != hipSuccess) return RNNT_STATUS_WARP_FAILED;
        // gradient pairs in place of the log-prob pairs, parked in alphas / betas (dead by then), turned back into the
        // row-major layout through LDS tiles: three coalesced passes (33 + 35 + 30 us at N=16, T=1500, U=300) instead
        // of one whose row-major reads and writes are scattered over the diagonal-major thread order (158 us)
        GradArgs gs{grads, nullptr, xn, yn, alphas, betas, ll, grads, costs, nullptr, T, U, 2, 0, fastemit_lambda};
        if (launch_grads(stream, gs, N, LOAD_SKEWED, WRITE_SKEWED2) != hipSuccess) return RNNT_STATUS_GRADS_BLANK_FAILED;
        if (launch_split_pairs(stream, grads, alphas, betas, cells) != hipSuccess) return RNNT_STATUS_GRADS_LABEL_FAILED;
        if (launch_unskew(stream, alphas, betas, grads, N, T, U) != hipSuccess) return RNNT_STATUS_GRADS_LABEL_FAILED;
        return RNNT_STATUS_SUCCESS;
    }
    LatticeArgs la{log_probs, nullptr, xn, yn, alphas, betas, ll, T, U, 2, 0};
    if (launch_lattice(stream, la, N, LOAD_ROWMAJOR2) != hipSuccess) return RNNT_STATUS_WARP_FAILED;
    GradArgs ga{log_probs, nullptr, xn, yn, alphas, betas, ll, grads, costs, nullptr,
                T, U, 2, 0, fastemit_lambda};
    if (launch_grads(stream, ga, N, LOAD_ROWMAJOR2, WRITE_ROWMAJOR2) != hipSuccess)
        return RNNT_STATUS_GRADS_BLANK_FAILED;
    return RNNT_STATUS_SUCCESS;
}

rnntStatus_t rnnt_amd_loss(rnntStream_t stream, void* workspace, int input_kind, const float* input,
                           const int* labels, const int* xn, const int* yn, float* costs, float* grads,
                           int grads_kind, int N, int T, int U, int V, int blank,
                           float fastemit_lambda) {
    if (!dims_ok(N, T, U) || !workspace) return RNNT_STATUS_INVALID_ARGUMENT;
    if (reinterpret_cast<uintptr_t>(workspace) % ALIGN) return RNNT_STATUS_INVALID_ARGUMENT;
    const bool gathered_in = input_kind == RNNT_IN_LOG_PROBS_GATHERED;
    if (!gathered_in) {
        if (V < 1 || blank < 0 || blank >= V) return RNNT_STATUS_INVALID_ARGUMENT;
        if (U > 1 && !labels) return RNNT_STATUS_INVALID_ARGUMENT;
    }
    if (grads_kind == RNNT_GRADS_DENSE && input_kind != RNNT_IN_LOG_PROBS_DENSE)
        return RNNT_STATUS_INVALID_ARGUMENT;
    if (grads_kind < RNNT_GRADS_GATHERED || grads_kind > RNNT_GRADS_NONE)
        return RNNT_STATUS_INVALID_ARGUMENT;
    if (grads_kind != RNNT_GRADS_NONE && !grads) return RNNT_STATUS_INVALID_ARGUMENT;
    if (N == 0) return RNNT_STATUS_SUCCESS;

    Workspace w;
    carve(workspace, N, T, U, &w);

    // 1. bring the (blank,label) log-prob pairs into the diagonal-major workspace -- and, where the sweeps are going to run on
    //    the ring kernel, let that launch prepare its flags and rings on the way (kernels.h: RingPrep)
    LatticeArgs la{w.ws2, nullptr, xn, yn, w.alphas, w.betas, w.ll, T, U, 2, 0, nullptr, w.redo, w.redo + 2 * N, w.mail};
    RingPrep prep{nullptr, 0, nullptr, nullptr, 0};
    hipError_t e;
    switch (input_kind) {
        case RNNT_IN_LOG_PROBS_DENSE:
            la.prepared = lattice_ring_prep(stream, la, N, LOAD_SKEWED, &prep) ? 1 : 0;
            e = launch_gather(stream, input, labels, w.ws2, N, T, U, V, blank, true, &prep); break;
        case RNNT_IN_LOG_PROBS_GATHERED:
            la.prepared = lattice_ring_prep(stream, la, N, LOAD_SKEWED, &prep) ? 1 : 0;
            e = launch_reskew(stream, input, w.ws2, N, T, U, &prep); break;
        case RNNT_IN_LOGITS_DENSE:
            e = launch_log_softmax_gather_skewed(stream, input, labels, w.ws2, N, T, U, V, blank); break;
        default:
            return RNNT_STATUS_INVALID_ARGUMENT;
    }
    if (e != hipSuccess) return RNNT_STATUS_PROLOGUE_FAILED;

    // 2. alpha / beta sweeps
    if (launch_lattice(stream, la, N, LOAD_SKEWED) != hipSuccess) return RNNT_STATUS_WARP_FAILED;

    // 3. gradients + costs (+ guard).  For a dense result the pairs are produced in place in the
    //    workspace and expanded to full rows (zeros included) in one coalesced pass.
    float* gout = grads;
    int writer = WRITE_SKEWED2;
    // (row-major pairs for the caller: produced in place in the workspace like the others, then turned through LDS
    //  tiles -- 33 + 30 us at N=16, T=1500, U=300 against 92 for row-major writes from the diagonal-major thread order)
    const bool unskew_gathered = grads_kind == RNNT_GRADS_GATHERED && (size_t)N * T * U >= STAGED_FROM_CELLS;
    if (grads_kind == RNNT_GRADS_GATHERED && !unskew_gathered) writer = WRITE_ROWMAJOR2;   // (launch-bound sizes: direct)
    else if (grads_kind != RNNT_GRADS_GATHERED_DIAGONAL) gout = w.ws2;
    if (grads_kind == RNNT_GRADS_DENSE && (size_t)N * T * U <= dense_in_one_launch_cells()) {
        // launch-bound sizes: the gradient kernel and the expansion as one launch (c2 in bench.py: 0.0371 -> 0.0348 ms per
        // step); the row writer reads its cells' values from the planes itself
        GradArgs gd{w.ws2, labels, xn, yn, w.alphas, w.betas, w.ll, nullptr, costs, w.mismatch, T, U, V, blank,
                    fastemit_lambda};
        if (launch_grads_dense(stream, gd, grads, N) != hipSuccess) return RNNT_STATUS_GRADS_BLANK_FAILED;
        return RNNT_STATUS_SUCCESS;
    }
    GradArgs ga{w.ws2, nullptr, xn, yn, w.alphas, w.betas, w.ll, gout, costs, w.mismatch,
                T, U, 2, 0, fastemit_lambda};
    if (launch_grads(stream, ga, N, LOAD_SKEWED, writer) != hipSuccess)
        return RNNT_STATUS_GRADS_BLANK_FAILED;
    if (grads_kind == RNNT_GRADS_DENSE) {
        if (launch_expand(stream, w.ws2, labels, xn, yn, nullptr, grads, N, T, U, V, blank, 1) != hipSuccess)
            return RNNT_STATUS_EXPAND_FAILED;
    }
    if (unskew_gathered) {
        if (launch_unskew(stream, w.ws2, nullptr, grads, N, T, U) != hipSuccess) return RNNT_STATUS_EXPAND_FAILED;
    }
    return RNNT_STATUS_SUCCESS;
}

// Diagnostics: run only the alpha/beta sweep on whatever the workspace holds (after a call to
// rnnt_amd_loss the (blank,label) pairs are still there unless grads were produced in place).
rnntStatus_t rnnt_amd_debug_lattice_only(rnntStream_t stream, void* workspace, const int* xn,
                                         const int* yn, int N, int T, int U) {
    if (!dims_ok(N, T, U) || !workspace) return RNNT_STATUS_INVALID_ARGUMENT;
    Workspace w;
    carve(workspace, N, T, U, &w);
    LatticeArgs la{w.ws2, nullptr, xn, yn, w.alphas, w.betas, w.ll, T, U, 2, 0, nullptr, w.redo, w.redo + 2 * N, w.mail};
    if (launch_lattice(stream, la, N, LOAD_SKEWED) != hipSuccess) return RNNT_STATUS_WARP_FAILED;
    return RNNT_STATUS_SUCCESS;
}

namespace {
struct CompactWorkspace {
    float* alphas;
    float* betas;
    float* ws2;
    float* ll;
    int* mismatch;
    int* redo;                  // as Workspace::redo
    unsigned long long* mail;   // as Workspace::mail, sized by the launch bounds (Tmax, Umax)
};
size_t carve_compact(void* base, int N, int64_t STU, int Tmax, int Umax, CompactWorkspace* w) {
    size_t off = 0;
    char* p = static_cast<char*>(base);
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return p ? p + o : nullptr; };
    float* alphas = reinterpret_cast<float*>(take((size_t)STU * 4));
    float* betas = reinterpret_cast<float*>(take((size_t)STU * 4));
    float* ws2 = reinterpret_cast<float*>(take((size_t)STU * 8));
    float* ll = reinterpret_cast<float*>(take((size_t)N * 4));
    int* mismatch = reinterpret_cast<int*>(take((size_t)N * 4));
    int* redo = reinterpret_cast<int*>(take(((size_t)N * 2 + 2) * sizeof(int)));
    unsigned long long* mail = reinterpret_cast<unsigned long long*>(take(lattice_mail_bytes(N, Tmax, Umax)));
    if (w) *w = CompactWorkspace{alphas, betas, ws2, ll, mismatch, redo, mail};
    return off + ALIGN;
}
bool compact_dims_ok(int N, int64_t STU, int Tmax, int Umax) {
    return dims_ok(N, Tmax > 0 ? Tmax : 1, Umax > 0 ? Umax : 1) && STU >= 0 && STU < ((int64_t)1 << 32);
}
}  // namespace

// Diagnostics (bench.py: the gather kernel's own roofline entry): only step 1 of rnnt_amd_loss for dense log-probs,
// i.e. k_to_diagonal<true> into the workspace's pair plane.
rnntStatus_t rnnt_amd_debug_gather_only(rnntStream_t stream, void* workspace, const float* log_probs,
                                        const int* labels, int N, int T, int U, int V, int blank) {
    if (!dims_ok(N, T, U) || !workspace || V < 1 || blank < 0 || blank >= V) return RNNT_STATUS_INVALID_ARGUMENT;
    if (U > 1 && !labels) return RNNT_STATUS_INVALID_ARGUMENT;
    Workspace w;
    carve(workspace, N, T, U, &w);
    if (launch_gather(stream, log_probs, labels, w.ws2, N, T, U, V, blank, true) != hipSuccess)
        return RNNT_STATUS_PROLOGUE_FAILED;
    return RNNT_STATUS_SUCCESS;
}

size_t rnnt_amd_workspace_size_compact(int N, int64_t STU, int Tmax, int Umax) {
    if (!compact_dims_ok(N, STU, Tmax, Umax)) return 0;
    return carve_compact(nullptr, N, STU, Tmax > 0 ? Tmax : 1, Umax > 0 ? Umax : 1, nullptr);
}

// Device-side preparation of a compact batch (offsets + launch bounds), one launch.
rnntStatus_t rnnt_amd_compact_offsets(rnntStream_t stream, const int* xn, const int* yn, int N,
                                      int64_t* cell_offsets, int* label_offsets, int64_t* stats) {
    if (N < 0 || N > 65535 || !cell_offsets || !label_offsets || !stats) return RNNT_STATUS_INVALID_ARGUMENT;
    if (N > 0 && (!xn || !yn)) return RNNT_STATUS_INVALID_ARGUMENT;
    if (launch_compact_offsets(stream, xn, yn, N, cell_offsets, label_offsets, stats) != hipSuccess)
        return RNNT_STATUS_PROLOGUE_FAILED;
    return RNNT_STATUS_SUCCESS;
}

// Compact (ragged packed) layout: replaces run_gather_for_compact + run_warp_rnnt_compact
// (core.h:41-54, core_compact.cu:360-436) with the conventions of the rest of this ABI
// (status codes, caller's stream, no exit(), no host synchronisation).
rnntStatus_t rnnt_amd_loss_compact(rnntStream_t stream, void* workspace, const float* xs, const int* ys,
                                   const int* xn, const int* yn, const int64_t* cell_offsets,
                                   const int* label_offsets, float* costs, float* grads2, int64_t* loc,
                                   int N, int64_t STU, int Tmax, int Umax, int V, int blank,
                                   float fastemit_lambda) {
    if (!compact_dims_ok(N, STU, Tmax, Umax) || !workspace || V < 1 || blank < 0 || blank >= V)
        return RNNT_STATUS_INVALID_ARGUMENT;
    if (reinterpret_cast<uintptr_t>(workspace) % ALIGN) return RNNT_STATUS_INVALID_ARGUMENT;
    if (N == 0 || STU == 0) return RNNT_STATUS_SUCCESS;
    CompactWorkspace w;
    carve_compact(workspace, N, STU, Tmax > 0 ? Tmax : 1, Umax > 0 ? Umax : 1, &w);
    float *alphas = w.alphas, *betas = w.betas, *ws2 = w.ws2, *ll = w.ll;
    int* mismatch = w.mismatch;
    if (launch_gather_compact(stream, xs, ys, xn, yn, cell_offsets, label_offsets, ws2, loc, N, Tmax, Umax, V,
                              blank, STU) != hipSuccess)
        return RNNT_STATUS_PROLOGUE_FAILED;
    LatticeArgs la{ws2, nullptr, xn, yn, alphas, betas, ll, Tmax, Umax, 2, 0, cell_offsets, w.redo, w.redo + 2 * N, w.mail};
    if (launch_lattice(stream, la, N, LOAD_SKEWED) != hipSuccess) return RNNT_STATUS_WARP_FAILED;
    GradArgs ga{ws2, nullptr, xn, yn, alphas, betas, ll, grads2 ? grads2 : ws2, costs, mismatch,
                Tmax, Umax, 2, 0, fastemit_lambda, cell_offsets};
    if (launch_grads(stream, ga, N, LOAD_SKEWED, grads2 ? WRITE_ROWMAJOR2 : WRITE_SKEWED2) != hipSuccess)
        return RNNT_STATUS_GRADS_BLANK_FAILED;
    return RNNT_STATUS_SUCCESS;
}

// The same with the launch bounds supplied by the caller: the offsets are computed here, on the device, and what the
// host would have checked after reading the maxima back is checked there too (prologue.hip: k_compact_offsets) -- no
// host synchronisation anywhere, so the call can be captured into a HIP graph.  A batch whose lengths do not fit the
// bounds, or whose totals are not STU / n_labels, gets NaN costs and zero gradients.
namespace {
struct BoundedExtra {
    int64_t* cell_offsets;   // (N+1,)
    int64_t* stats;          // (5,): the four of rnnt_amd_compact_offsets + "refused"
    int* label_offsets;      // (N+1,)
    int* xn_checked;         // (N,)
};
size_t carve_bounded(char* base, size_t off, int N, BoundedExtra* e) {
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes); return base ? base + o : nullptr; };
    int64_t* co = reinterpret_cast<int64_t*>(take(((size_t)N + 1) * 8));
    int64_t* st = reinterpret_cast<int64_t*>(take(5 * 8));
    int* lo = reinterpret_cast<int*>(take(((size_t)N + 1) * 4));
    int* xc = reinterpret_cast<int*>(take((size_t)N * 4));
    if (e) *e = BoundedExtra{co, st, lo, xc};
    return off;
}
}  // namespace

size_t rnnt_amd_workspace_size_compact_bounded(int N, int64_t STU, int Tmax, int Umax) {
    const size_t base = rnnt_amd_workspace_size_compact(N, STU, Tmax, Umax);
    if (base == 0) return 0;
    return carve_bounded(nullptr, base, N, nullptr);
}

rnntStatus_t rnnt_amd_loss_compact_bounded(rnntStream_t stream, void* workspace, const float* xs, const int* ys,
                                           int64_t n_labels, const int* xn, const int* yn, float* costs, float* grads2,
                                           int64_t* loc, int N, int64_t STU, int Tmax, int Umax, int V, int blank,
                                           float fastemit_lambda) {
    if (!compact_dims_ok(N, STU, Tmax, Umax) || !workspace || Tmax < 1 || Umax < 1 || n_labels < 0)
        return RNNT_STATUS_INVALID_ARGUMENT;
    if (reinterpret_cast<uintptr_t>(workspace) % ALIGN) return RNNT_STATUS_INVALID_ARGUMENT;
    if (N == 0) return RNNT_STATUS_SUCCESS;
    if (STU == 0) {
        // utterances but no cells: no length >= 1 frame can add up to that, so the batch is one this entry always
        // refuses -- and the kernels behind it return before they write anything.  Say so the way a refusal does.
        if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(costs), 0x7fc00000, (size_t)N, stream) != hipSuccess)
            return RNNT_STATUS_COSTS_FAILED;
        return RNNT_STATUS_SUCCESS;
    }
    BoundedExtra e;
    carve_bounded(static_cast<char*>(workspace), rnnt_amd_workspace_size_compact(N, STU, Tmax, Umax), N, &e);
    const CompactBounds b{e.xn_checked, STU, n_labels, Tmax, Umax};
    if (launch_compact_offsets(stream, xn, yn, N, e.cell_offsets, e.label_offsets, e.stats, &b) != hipSuccess)
        return RNNT_STATUS_PROLOGUE_FAILED;
    const rnntStatus_t st = rnnt_amd_loss_compact(stream, workspace, xs, ys, e.xn_checked, yn, e.cell_offsets,
                                                  e.label_offsets, costs, grads2, loc, N, STU, Tmax, Umax, V, blank,
                                                  fastemit_lambda);
    if (st != RNNT_STATUS_SUCCESS) return st;
    if (launch_zero_if_refused(stream, e.stats + 4, grads2, (size_t)STU) != hipSuccess) return RNNT_STATUS_EXPAND_FAILED;
    return RNNT_STATUS_SUCCESS;
}

// Replaces run_scatter_grad_for_compact (core.h:56-60, core_compact.cu:456-500): dense (STU,V)
// gradient rows, fully written.  cum_lens = inclusive prefix sums of xn*(yn+1) (int32, as the
// reference's RNNTLossCompact.forward builds them, __init__.py:38).
rnntStatus_t rnnt_amd_compact_scatter_grads(rnntStream_t stream, const float* grad_costs,
                                            const float* grads2, const int64_t* loc, const int* cum_lens,
                                            float* dense_grads, int64_t STU, int N, int V, int blank) {
    if (N < 0 || STU < 0 || V < 1 || blank < 0 || blank >= V) return RNNT_STATUS_INVALID_ARGUMENT;
    if (launch_scatter_compact(stream, grad_costs, grads2, loc, cum_lens, dense_grads, STU, N, V, blank) !=
        hipSuccess)
        return RNNT_STATUS_EXPAND_FAILED;
    return RNNT_STATUS_SUCCESS;
}

// ---------------------------------------------------------------------------------------------------------
// The reference's own compact entry points (core.h:41-60), same names and argument lists, so that its
// binding.cpp links against this library whole.  Kept conventions: void return, the NULL stream (its binding
// relies on stream 0: it launches nothing on a stream of its own, binding.cpp:170,197,241).  Changed: a failed
// launch does not exit(-1) the process (core.h:7-14) but is remembered per thread -- rnnt_amd_compact_last_status().
// ---------------------------------------------------------------------------------------------------------
static thread_local rnntStatus_t g_compact_status = RNNT_STATUS_SUCCESS;

// A void entry point has failed: remember the status for rnnt_amd_compact_last_status(), say so on stderr (the
// reference's CHECK_KERNEL_STAT prints and exits, core.h:7-14; its binding never looks at a status) and make the
// failure impossible to miss downstream: costs become NaN (best effort, NULL stream like everything here) instead
// of whatever torch.empty held, so a training loop that ignores the status stops on a NaN loss rather than
// learning from uninitialised memory.
// `err` is what the failed launcher returned (it has already taken the error off the runtime's per-thread slot, so
// asking hipGetLastError() here would report "no error").
static void compact_fail(rnntStatus_t st, const char* what, float* costs, unsigned int N, hipError_t err = hipSuccess) {
    g_compact_status = st;
    fprintf(stderr, "%s failed: rnnt status %d (%s)\n", what, (int)st,
            st == RNNT_STATUS_INVALID_ARGUMENT ? "invalid argument or unsupported size" : hipGetErrorString(err));
    if (costs && N) {
        const hipError_t m = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(costs), 0x7fc00000, N, nullptr);
        if (m != hipSuccess)
            fprintf(stderr, "%s: costs could not be set to NaN either (%s)\n", what, hipGetErrorString(m));
    }
}

rnntStatus_t rnnt_amd_compact_last_status(void) {
    const rnntStatus_t s = g_compact_status;
    g_compact_status = RNNT_STATUS_SUCCESS;
    return s;
}

void run_gather_for_compact(const float* xs, const int* ys, const unsigned int* xn, const unsigned int* yn,
                            float* gather_xs, long* loc, const unsigned int* memPref,
                            const unsigned int* labelPref, unsigned int N, unsigned int T, unsigned int U,
                            unsigned int V, unsigned int blank) {
    static_assert(sizeof(long) == sizeof(int64_t), "loc is the reference's `long` (at::kLong)");
    if (V < 1 || blank >= V) { compact_fail(RNNT_STATUS_INVALID_ARGUMENT, "run_gather_for_compact", nullptr, 0); return; }
    const hipError_t e = launch_gather_compact_rowmajor(nullptr, xs, ys, xn, yn, gather_xs,
                                                        reinterpret_cast<int64_t*>(loc), memPref, labelPref, N, T, U, V,
                                                        blank);
    if (e != hipSuccess) compact_fail(RNNT_STATUS_PROLOGUE_FAILED, "run_gather_for_compact", nullptr, 0, e);
}

void run_warp_rnnt_compact(unsigned int* counts, float* alphas, float* betas, const float* log_probs, float* grads,
                           float* costs, const unsigned int* xn, const unsigned int* yn,
                           const unsigned int* memPref, const unsigned int* labelPref, unsigned int N,
                           unsigned int T, unsigned int U, float fastemit_lambda, bool required_grad) {
    (void)labelPref;
    if (N == 0) return;
    if (!dims_ok((int)N, T > 0 ? (int)T : 1, U > 0 ? (int)U : 1)) {
        compact_fail(RNNT_STATUS_INVALID_ARGUMENT, "run_warp_rnnt_compact", costs, N);
        return;
    }
    const int* ixn = reinterpret_cast<const int*>(xn);
    const int* iyn = reinterpret_cast<const int*>(yn);
    // counts (2*sum(yn) + 2N words, binding.cpp:187): the first N hold the alpha-side log-likelihoods
    float* ll = reinterpret_cast<float*>(counts);
    // Staged like the padded entry points (round 5; until then: the single-role kernel with per-lane row-major loads,
    // tools/cabi_probe.py): `grads` (STU,2) is the caller's and not yet written -- exactly one plane of pairs.  The
    // row-major pairs are turned into each utterance's diagonal-major plane there (LDS tiles), the sweeps run on the
    // tuned kernels that need nothing but planes (k_lattice_wd as a plain launch for U <= 64, k_lattice_wl, lattice_ws.hip),
    // the gradient pairs replace the log-prob pairs in place, are parked in alphas / betas (dead by then) and turned
    // back.  Five coalesced passes instead of two whose loads and stores are scattered over the diagonal-major thread
    // order.  Not for required_grad = false (alphas and grads alias betas there, binding.cpp:192-195) and not below
    // 2^20 cells of launch bound (launch-bound sizes keep the two-launch form).
    const bool staged = required_grad && (size_t)N * T * U >= STAGED_FROM_CELLS && reinterpret_cast<uintptr_t>(grads) % 16 == 0 &&
                        reinterpret_cast<uintptr_t>(alphas) % 8 == 0 && reinterpret_cast<uintptr_t>(betas) % 8 == 0 &&
                        grads != betas && alphas != betas;
    // (the staged form's first launch -- nothing of the caller's has been overwritten yet -- may be refused where the direct
    //  form is not, e.g. on its 2^31-workgroup grid limit: then the direct form runs)
    hipError_t e_stage = staged ? launch_reskew_compact32(nullptr, log_probs, grads, xn, yn, memPref, N, T, U) : hipErrorNotSupported;
    if (staged && e_stage != hipSuccess) (void)hipGetLastError();
    if (staged && e_stage == hipSuccess) {
        hipError_t e;
        LatticeArgs ls{grads, nullptr, ixn, iyn, alphas, betas, ll, (int)T, (int)U, 2, 0};
        ls.offs32 = memPref;
        e = launch_lattice(nullptr, ls, (int)N, LOAD_SKEWED);
        if (e != hipSuccess) { compact_fail(RNNT_STATUS_WARP_FAILED, "run_warp_rnnt_compact", costs, N, e); return; }
        GradArgs gs{grads, nullptr, ixn, iyn, alphas, betas, ll, grads, costs, nullptr, (int)T, (int)U, 2, 0, fastemit_lambda};
        gs.offs32 = memPref;
        e = launch_grads(nullptr, gs, (int)N, LOAD_SKEWED, WRITE_SKEWED2);
        if (e != hipSuccess) { compact_fail(RNNT_STATUS_GRADS_BLANK_FAILED, "run_warp_rnnt_compact", costs, N, e); return; }
        e = launch_split_pairs_compact32(nullptr, grads, alphas, betas, xn, yn, memPref, N, (size_t)N * T * U);
        if (e == hipSuccess) e = launch_unskew_compact32(nullptr, alphas, betas, grads, xn, yn, memPref, N, T, U);
        if (e != hipSuccess) compact_fail(RNNT_STATUS_GRADS_LABEL_FAILED, "run_warp_rnnt_compact", costs, N, e);
        return;
    }
    LatticeArgs la{log_probs, nullptr, ixn, iyn, alphas, betas, ll, (int)T, (int)U, 2, 0};
    la.offs32 = memPref;
    la.beta_only = required_grad ? 0 : 1;
    hipError_t e = launch_lattice(nullptr, la, (int)N, LOAD_ROWMAJOR2);
    if (e != hipSuccess) {
        compact_fail(RNNT_STATUS_WARP_FAILED, "run_warp_rnnt_compact", costs, N, e);
        return;
    }
    if (!required_grad) {   // the reference's "beta only" inference mode: costs from beta[0,0], nothing else is touched
        e = launch_costs_from_betas(nullptr, betas, memPref, ixn, iyn, costs, (int)N);
        if (e != hipSuccess) compact_fail(RNNT_STATUS_COSTS_FAILED, "run_warp_rnnt_compact", costs, N, e);
        return;
    }
    GradArgs ga{log_probs, nullptr, ixn, iyn, alphas, betas, ll, grads, costs, nullptr, (int)T, (int)U, 2, 0,
                fastemit_lambda};
    ga.offs32 = memPref;
    e = launch_grads(nullptr, ga, (int)N, LOAD_ROWMAJOR2, WRITE_ROWMAJOR2);
    if (e != hipSuccess) compact_fail(RNNT_STATUS_GRADS_BLANK_FAILED, "run_warp_rnnt_compact", costs, N, e);
}

void run_scatter_grad_for_compact(const float* grad_cost, const float* gather_grad, const long* loc,
                                  const int* cum_lens, float* scatter_grad, unsigned int STU, unsigned int N,
                                  unsigned int V, unsigned int blank) {
    if (V < 1 || blank >= V) { compact_fail(RNNT_STATUS_INVALID_ARGUMENT, "run_scatter_grad_for_compact", nullptr, 0); return; }
    const hipError_t e = launch_scatter_compact(nullptr, grad_cost, gather_grad, reinterpret_cast<const int64_t*>(loc),
                                                cum_lens, scatter_grad, (int64_t)STU, (int)N, (int)V, (int)blank);
    if (e != hipSuccess) compact_fail(RNNT_STATUS_EXPAND_FAILED, "run_scatter_grad_for_compact", nullptr, 0, e);
}

rnntStatus_t rnnt_amd_expand_grads(rnntStream_t stream, const float* grads_diagonal, const int* labels,
                                   const int* xn, const int* yn, const float* grad_costs,
                                   float* dense_grads, int N, int T, int U, int V, int blank,
                                   int overwrite) {
    if (!dims_ok(N, T, U) || V < 1 || blank < 0 || blank >= V) return RNNT_STATUS_INVALID_ARGUMENT;
    if ((int64_t)U * V >= (int64_t)1 << 31) return RNNT_STATUS_INVALID_ARGUMENT;
    if ((int64_t)N * T >= (int64_t)1 << 31) return RNNT_STATUS_INVALID_ARGUMENT;
    if (launch_expand(stream, grads_diagonal, labels, xn, yn, grad_costs, dense_grads, N, T, U, V, blank,
                      overwrite) != hipSuccess)
        return RNNT_STATUS_EXPAND_FAILED;
    return RNNT_STATUS_SUCCESS;
}

// Backward of the fused RNNT_IN_LOGITS_DENSE path: d(sum_n grad_costs[n]*cost[n]) / d(logits).
rnntStatus_t rnnt_amd_logits_backward(rnntStream_t stream, const float* logits, const int* labels,
                                      const float* grads_diagonal, const float* grad_costs, float* dlogits,
                                      int N, int T, int U, int V, int blank) {
    if (!dims_ok(N, T, U) || V < 1 || blank < 0 || blank >= V) return RNNT_STATUS_INVALID_ARGUMENT;
    if (U > 1 && !labels) return RNNT_STATUS_INVALID_ARGUMENT;
    if (launch_logits_backward(stream, logits, labels, grads_diagonal, grad_costs, dlogits, N, T, U, V, blank) !=
        hipSuccess)
        return RNNT_STATUS_EXPAND_FAILED;
    return RNNT_STATUS_SUCCESS;
}

rnntStatus_t rnnt_amd_log_softmax(rnntStream_t stream, const float* x, float* out, int64_t rows, int V) {
    if (rows < 0 || V < 1) return RNNT_STATUS_INVALID_ARGUMENT;
    if (launch_log_softmax(stream, x, out, rows, V) != hipSuccess) return RNNT_STATUS_PROLOGUE_FAILED;
    return RNNT_STATUS_SUCCESS;
}

rnntStatus_t rnnt_amd_log_softmax_backward(rnntStream_t stream, const float* grad_out, const float* out,
                                           float* grad_in, int64_t rows, int V) {
    if (rows < 0 || V < 1) return RNNT_STATUS_INVALID_ARGUMENT;
    if (launch_log_softmax_backward(stream, grad_out, out, grad_in, rows, V) != hipSuccess)
        return RNNT_STATUS_PROLOGUE_FAILED;
    return RNNT_STATUS_SUCCESS;
}

rnntStatus_t rnnt_amd_gather(rnntStream_t stream, const float* log_probs, const int* labels,
                             float* gathered, int N, int T, int U, int V, int blank) {
    if (!dims_ok(N, T, U) || V < 1 || blank < 0 || blank >= V) return RNNT_STATUS_INVALID_ARGUMENT;
    if (launch_gather(stream, log_probs, labels, gathered, N, T, U, V, blank, false) != hipSuccess)
        return RNNT_STATUS_PROLOGUE_FAILED;
    return RNNT_STATUS_SUCCESS;
}

}  // extern "C"
