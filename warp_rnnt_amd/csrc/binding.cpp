// Compiled native-op module `warp_rnnt._C_native`: what the reference's pybind module is to its CUDA kernels
// (pytorch_binding/binding.cpp:28-106, 249-254), written against the C ABI of libwarp_rnnt_amd.so
// (include/warp_rnnt_amd.h).  Host code only: argument checks in the reference's order with the reference's
// messages, output/workspace allocation from PyTorch's caching allocator, the current HIP stream, one C call.
// The ctypes twin (warp_rnnt/_C.py + warp_rnnt_amd/ops.py) stays as the fallback when this module has not been
// built; both sit on the same library and there is no CPU path in either.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>

#include <tuple>

#include "../../include/warp_rnnt_amd.h"

namespace {

#define RNNT_CHECK_CONTIGUOUS(x) TORCH_CHECK((x).is_contiguous(), #x " must be contiguous")
#define RNNT_CHECK_FLOAT(x) TORCH_CHECK((x).scalar_type() == at::ScalarType::Float, #x " must be a Float tensor")
#define RNNT_CHECK_INT(x) TORCH_CHECK((x).scalar_type() == at::ScalarType::Int, #x " must be a Int tensor")
#define RNNT_CHECK_CUDA(x) TORCH_CHECK((x).device().is_cuda(), #x " must be located in the CUDA")

const char* status_name(int st) {
    switch (st) {
        case 1: return "RNNT_STATUS_WARP_FAILED";
        case 2: return "RNNT_STATUS_GRADS_BLANK_FAILED";
        case 3: return "RNNT_STATUS_GRADS_LABEL_FAILED";
        case 4: return "RNNT_STATUS_COSTS_FAILED";
        case 5: return "RNNT_STATUS_INVALID_ARGUMENT";
        case 6: return "RNNT_STATUS_PROLOGUE_FAILED";
        case 7: return "RNNT_STATUS_EXPAND_FAILED";
        default: return "?";
    }
}

void check_status(int st) {
    // same text as the reference's TORCH_CHECK (binding.cpp:102-103)
    TORCH_CHECK(st == 0, "rnnt_loss status ", st, " (", status_name(st), ")");
}

// binding.cpp:32-51 -- contiguity, then dtypes, then device, then shapes
void check_inputs(const at::Tensor& xs, const at::Tensor& ys, const at::Tensor& xn, const at::Tensor& yn) {
    RNNT_CHECK_CONTIGUOUS(xs); RNNT_CHECK_CONTIGUOUS(ys); RNNT_CHECK_CONTIGUOUS(xn); RNNT_CHECK_CONTIGUOUS(yn);
    RNNT_CHECK_FLOAT(xs); RNNT_CHECK_INT(ys); RNNT_CHECK_INT(xn); RNNT_CHECK_INT(yn);
    RNNT_CHECK_CUDA(xs); RNNT_CHECK_CUDA(ys); RNNT_CHECK_CUDA(xn); RNNT_CHECK_CUDA(yn);
    TORCH_CHECK(xs.dim() == 4, "xs must have 4 dimensions");
    TORCH_CHECK(xn.numel() == xs.size(0), "xn shape must be equal (N,)");
    TORCH_CHECK(yn.numel() == xs.size(0), "yn shape must be equal (N,)");
    TORCH_CHECK(ys.dim() == 2 && xs.size(2) == ys.size(1) + 1, "ys shape (N, U-1) mismatched with xs (N, T, U, V)");
    TORCH_CHECK(ys.device() == xs.device(), "ys must be on the same device as xs");
    TORCH_CHECK(xn.device() == xs.device(), "xn must be on the same device as xs");
    TORCH_CHECK(yn.device() == xs.device(), "yn must be on the same device as xs");
}

rnntStream_t current_stream(const at::Tensor& t) {
    return reinterpret_cast<rnntStream_t>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream());
}

// costs, grads (layout per grads_kind), and -- when asked for -- the (N,) int32 guard flags
std::tuple<at::Tensor, at::Tensor, at::Tensor> loss(const at::Tensor& input, const at::Tensor& labels,
                                                    const at::Tensor& xn, const at::Tensor& yn, int input_kind,
                                                    int grads_kind, int64_t blank, double fastemit_lambda,
                                                    bool want_mismatch) {
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(input.device());
    const int64_t N = input.size(0), T = input.size(1), U = input.size(2), V = input.size(3);
    TORCH_CHECK(N < (1ll << 31) && T < (1ll << 31) && U < (1ll << 31) && V < (1ll << 31),
                "rnnt_loss status 5 (RNNT_STATUS_INVALID_ARGUMENT): unsupported sizes");
    if (input_kind == RNNT_IN_LOG_PROBS_GATHERED) blank = 0;   // channel 0 of the 2-channel layout
    else TORCH_CHECK(blank >= 0 && blank < V, "rnnt_loss status 5 (RNNT_STATUS_INVALID_ARGUMENT): blank=", blank,
                     " is not a vocabulary index of xs (V=", V, ")");
    const auto fopt = input.options();
    at::Tensor costs = at::empty({N}, fopt);
    at::Tensor grads = grads_kind == RNNT_GRADS_DENSE ? at::empty_like(input)
                                                       : at::empty({N, T, U, 2}, fopt);
    at::Tensor mismatch;
    if (N == 0) {
        if (want_mismatch) mismatch = at::zeros({0}, fopt.dtype(at::kInt));
        return {costs, grads, mismatch};
    }
    const size_t ws_bytes = rnnt_amd_workspace_size((int)N, (int)T, (int)U);
    TORCH_CHECK(ws_bytes != 0, "rnnt_loss status 5 (RNNT_STATUS_INVALID_ARGUMENT): unsupported sizes N=", N, " T=", T,
                " U=", U);
    at::Tensor ws = at::empty({(int64_t)ws_bytes}, fopt.dtype(at::kByte));
    const int st = rnnt_amd_loss(current_stream(input), ws.data_ptr(), input_kind, input.data_ptr<float>(),
                                 labels.defined() && labels.numel() ? labels.data_ptr<int>() : nullptr,
                                 xn.data_ptr<int>(), yn.data_ptr<int>(), costs.data_ptr<float>(),
                                 grads.data_ptr<float>(), grads_kind, (int)N, (int)T, (int)U, (int)V, (int)blank,
                                 (float)fastemit_lambda);
    check_status(st);
    if (want_mismatch) {
        const int64_t off = (int64_t)rnnt_amd_workspace_mismatch_offset((int)N, (int)T, (int)U);
        mismatch = ws.narrow(0, off, 4 * N).view(at::kInt).clone();
    }
    return {costs, grads, mismatch};
}

// ---- the reference's native op (binding.cpp:28-106): blank == -1 selects the gathered (N,T,U,2) layout ----
std::tuple<at::Tensor, at::Tensor, at::Tensor> rnnt_loss(const at::Tensor& xs, const at::Tensor& ys,
                                                         const at::Tensor& xn, const at::Tensor& yn, int64_t blank,
                                                         double fastemit_lambda, bool want_mismatch) {
    check_inputs(xs, ys, xn, yn);
    if (blank == -1) {
        TORCH_CHECK(xs.size(3) == 2, "xs must have values only for blank and label");
        return loss(xs, at::Tensor(), xn, yn, RNNT_IN_LOG_PROBS_GATHERED, RNNT_GRADS_GATHERED, -1, fastemit_lambda,
                    want_mismatch);
    }
    return loss(xs, ys, xn, yn, RNNT_IN_LOG_PROBS_DENSE, RNNT_GRADS_DENSE, blank, fastemit_lambda, want_mismatch);
}

// native form of the wrapper's gather=True branch (__init__.py:118-128): dense log-probs in, costs and the
// (opaque, diagonal-major) gathered gradients out
std::tuple<at::Tensor, at::Tensor, at::Tensor> rnnt_loss_gather(const at::Tensor& xs, const at::Tensor& ys,
                                                                const at::Tensor& xn, const at::Tensor& yn,
                                                                int64_t blank, double fastemit_lambda,
                                                                bool want_mismatch) {
    check_inputs(xs, ys, xn, yn);
    return loss(xs, ys, xn, yn, RNNT_IN_LOG_PROBS_DENSE, RNNT_GRADS_GATHERED_DIAGONAL, blank, fastemit_lambda,
                want_mismatch);
}

// d loss / d log_probs (N,T,U,V) = scatter-add of the gathered grads times grad_costs[n]
at::Tensor rnnt_loss_gather_backward(const at::Tensor& grad_costs, const at::Tensor& grads_diagonal,
                                     const at::Tensor& ys, const at::Tensor& xn, const at::Tensor& yn, int64_t V,
                                     int64_t blank) {
    RNNT_CHECK_CONTIGUOUS(grad_costs); RNNT_CHECK_CONTIGUOUS(grads_diagonal);
    RNNT_CHECK_FLOAT(grad_costs); RNNT_CHECK_FLOAT(grads_diagonal);
    RNNT_CHECK_CONTIGUOUS(ys); RNNT_CHECK_CONTIGUOUS(xn); RNNT_CHECK_CONTIGUOUS(yn);
    RNNT_CHECK_INT(ys); RNNT_CHECK_INT(xn); RNNT_CHECK_INT(yn);
    RNNT_CHECK_CUDA(grads_diagonal);
    TORCH_CHECK(grads_diagonal.dim() == 4 && grads_diagonal.size(3) == 2, "grads_diagonal must be (N,T,U,2)");
    for (const at::Tensor* t : {&grad_costs, &ys, &xn, &yn})
        TORCH_CHECK(t->device() == grads_diagonal.device(), "all tensors must be on the device of grads_diagonal");
    TORCH_CHECK(grad_costs.numel() == grads_diagonal.size(0) && xn.numel() == grads_diagonal.size(0) &&
                yn.numel() == grads_diagonal.size(0), "grad_costs, xn, yn shape must be equal (N,)");
    TORCH_CHECK(ys.numel() == grads_diagonal.size(0) * (grads_diagonal.size(2) - 1),
                "ys shape (N, U-1) mismatched with grads_diagonal (N, T, U, 2)");
    TORCH_CHECK(V >= 1 && V < (1ll << 31) && blank >= 0 && blank < V, "rnnt_loss status 5 "
                "(RNNT_STATUS_INVALID_ARGUMENT): blank=", blank, " is not a vocabulary index (V=", V, ")");
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(grads_diagonal.device());
    const int64_t N = grads_diagonal.size(0), T = grads_diagonal.size(1), U = grads_diagonal.size(2);
    at::Tensor out = at::empty({N, T, U, V}, grads_diagonal.options());
    if (N == 0) return out;
    check_status(rnnt_amd_expand_grads(current_stream(out), grads_diagonal.data_ptr<float>(),
                                       ys.numel() ? ys.data_ptr<int>() : nullptr, xn.data_ptr<int>(),
                                       yn.data_ptr<int>(), grad_costs.data_ptr<float>(), out.data_ptr<float>(), (int)N,
                                       (int)T, (int)U, (int)V, (int)blank, 0));
    return out;
}

// ---- the reference's compact native ops (binding.cpp:109-247; pybind names and keywords :255-268) ----
// (costs (N,), grads (STU,2) -- an empty (0,2) tensor when required_grad is false --, loc (STU,) int64).
// max_frames / max_labels (an extension: both >= 0, or both -1) are launch bounds the caller vouches for: with them
// nothing is read back from the device (the reference reads yn.sum(), xn.max(), yn.max() and the last prefix sum: four
// synchronisations; this op without bounds: one), so the op can sit inside a captured HIP graph; the shape checks that
// need the sums then happen on the device and a batch that fails them comes back with NaN costs and zero gradients.
std::tuple<at::Tensor, at::Tensor, at::Tensor> rnnt_loss_compact_forward(const at::Tensor& xs, const at::Tensor& ys,
                                                                         const at::Tensor& xn, const at::Tensor& yn,
                                                                         int64_t blank, double fastemit_lambda,
                                                                         bool required_grad, int64_t max_frames,
                                                                         int64_t max_labels) {
    RNNT_CHECK_CONTIGUOUS(xs); RNNT_CHECK_CONTIGUOUS(ys); RNNT_CHECK_CONTIGUOUS(xn); RNNT_CHECK_CONTIGUOUS(yn);
    RNNT_CHECK_FLOAT(xs); RNNT_CHECK_INT(ys); RNNT_CHECK_INT(xn); RNNT_CHECK_INT(yn);
    RNNT_CHECK_CUDA(xs); RNNT_CHECK_CUDA(ys); RNNT_CHECK_CUDA(xn); RNNT_CHECK_CUDA(yn);
    TORCH_CHECK(xs.dim() == 2, "xs must have 2 dimensions");
    TORCH_CHECK(xn.size(0) == yn.size(0), "xn and yn shape must be equal (N,)");
    TORCH_CHECK((max_frames < 0) == (max_labels < 0), "max_frames and max_labels go together");
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(xs.device());
    const int64_t N = xn.size(0), STU = xs.size(0), V = xs.size(1);
    TORCH_CHECK(N < (1ll << 31) && V >= 1 && V < (1ll << 31) && blank >= 0 && blank < V,
                "rnnt_loss status 5 (RNNT_STATUS_INVALID_ARGUMENT): unsupported sizes or blank");
    const auto fopt = xs.options();
    at::Tensor costs = at::empty({N}, fopt);
    at::Tensor loc = at::empty({STU}, fopt.dtype(at::kLong));
    at::Tensor grads = required_grad ? at::empty({STU, 2}, fopt) : at::empty({0, 2}, fopt);
    if (N == 0) return {costs, grads, loc};
    const rnntStream_t stream = current_stream(xs);
    float* gptr = required_grad ? grads.data_ptr<float>() : nullptr;
    const int* ysp = ys.numel() ? ys.data_ptr<int>() : nullptr;
    if (max_frames >= 0) {
        const int tmax = (int)max_frames, umax = (int)max_labels + 1;
        TORCH_CHECK(max_frames >= 1 && max_frames < (1ll << 31) && max_labels < (1ll << 31) - 1,
                    "max_frames >= 1 and max_labels >= 0 expected");
        const size_t ws_bytes = rnnt_amd_workspace_size_compact_bounded((int)N, STU, tmax, umax);
        TORCH_CHECK(ws_bytes != 0, "rnnt_loss status 5 (RNNT_STATUS_INVALID_ARGUMENT): unsupported sizes");
        at::Tensor ws = at::empty({(int64_t)ws_bytes}, fopt.dtype(at::kByte));
        check_status(rnnt_amd_loss_compact_bounded(stream, ws.data_ptr(), xs.data_ptr<float>(), ysp, ys.numel(),
                                                   xn.data_ptr<int>(), yn.data_ptr<int>(), costs.data_ptr<float>(), gptr,
                                                   loc.data_ptr<int64_t>(), (int)N, STU, tmax, umax, (int)V, (int)blank,
                                                   (float)fastemit_lambda));
        return {costs, grads, loc};
    }
    at::Tensor offs = at::empty({N + 1 + 4}, fopt.dtype(at::kLong));
    at::Tensor loffs = at::empty({N + 1}, fopt.dtype(at::kInt));
    check_status(rnnt_amd_compact_offsets(stream, xn.data_ptr<int>(), yn.data_ptr<int>(), (int)N, offs.data_ptr<int64_t>(),
                                          loffs.data_ptr<int>(), offs.data_ptr<int64_t>() + N + 1));
    const at::Tensor stats = offs.narrow(0, N + 1, 4).cpu();              // the one host synchronisation
    const int64_t* st = stats.data_ptr<int64_t>();
    TORCH_CHECK(ys.numel() == st[1], "ys shape must be equal to (sum(yn), )");
    TORCH_CHECK(STU == st[0], "xs shape mismatch with (\\sum{xn*(yn+1)}, )");
    const int tmax = (int)st[2], umax = (int)st[3] + 1;
    const size_t ws_bytes = rnnt_amd_workspace_size_compact((int)N, STU, tmax, umax);
    TORCH_CHECK(ws_bytes != 0, "rnnt_loss status 5 (RNNT_STATUS_INVALID_ARGUMENT): unsupported sizes");
    at::Tensor ws = at::empty({(int64_t)ws_bytes}, fopt.dtype(at::kByte));
    check_status(rnnt_amd_loss_compact(stream, ws.data_ptr(), xs.data_ptr<float>(), ysp, xn.data_ptr<int>(),
                                       yn.data_ptr<int>(), offs.data_ptr<int64_t>(), loffs.data_ptr<int>(),
                                       costs.data_ptr<float>(), gptr, loc.data_ptr<int64_t>(), (int)N, STU, tmax, umax,
                                       (int)V, (int)blank, (float)fastemit_lambda));
    return {costs, grads, loc};
}

// binding.cpp:209-247: the (STU,2) gradients, scaled per utterance, scattered into whole (STU,V) rows
at::Tensor rnnt_loss_compact_backward(const at::Tensor& grad_cost, const at::Tensor& grad_xs, const at::Tensor& cum_lens,
                                      const at::Tensor& loc, int64_t V, int64_t blank) {
    RNNT_CHECK_CONTIGUOUS(grad_cost); RNNT_CHECK_CONTIGUOUS(grad_xs); RNNT_CHECK_CONTIGUOUS(loc);
    RNNT_CHECK_FLOAT(grad_cost); RNNT_CHECK_FLOAT(grad_xs);
    TORCH_CHECK(loc.scalar_type() == at::ScalarType::Long, "loc must be a Long tensor");
    RNNT_CHECK_CUDA(grad_cost); RNNT_CHECK_CUDA(grad_xs); RNNT_CHECK_CUDA(cum_lens); RNNT_CHECK_CUDA(loc);
    TORCH_CHECK(grad_cost.dim() == 1, "grad_cost must have 1 dimensions");
    TORCH_CHECK(grad_xs.dim() == 2, "grad must have 2 dimensions");
    TORCH_CHECK(grad_xs.size(0) == loc.size(0), "grad and loc must be equal in dim=0");
    TORCH_CHECK(cum_lens.scalar_type() == at::ScalarType::Int && cum_lens.is_contiguous(),
                "cum_lens must be a contiguous Int tensor");
    TORCH_CHECK(V >= 1 && V < (1ll << 31) && blank >= 0 && blank < V,
                "rnnt_loss status 5 (RNNT_STATUS_INVALID_ARGUMENT): blank is not a vocabulary index");
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(grad_cost.device());
    const int64_t N = grad_cost.size(0), STU = grad_xs.size(0);
    at::Tensor out = at::empty({STU, V}, grad_cost.options());
    if (STU == 0) return out;
    check_status(rnnt_amd_compact_scatter_grads(current_stream(out), grad_cost.data_ptr<float>(), grad_xs.data_ptr<float>(),
                                                loc.data_ptr<int64_t>(), cum_lens.data_ptr<int>(), out.data_ptr<float>(),
                                                STU, (int)N, (int)V, (int)blank));
    return out;
}

// the prologue the reference benchmarks next to the loss (benchmark.py:65,70): row-wise log-softmax; out may be x
at::Tensor log_softmax(const at::Tensor& x, const c10::optional<at::Tensor>& out_opt) {
    RNNT_CHECK_CONTIGUOUS(x); RNNT_CHECK_FLOAT(x); RNNT_CHECK_CUDA(x);
    at::Tensor out = out_opt.has_value() ? *out_opt : at::empty_like(x);
    TORCH_CHECK(out.is_contiguous() && out.scalar_type() == at::ScalarType::Float && out.sizes() == x.sizes() &&
                out.device() == x.device(), "out must be a contiguous Float tensor like x");
    const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x.device());
    const int64_t V = x.dim() ? x.size(-1) : 1;
    const int64_t rows = V ? x.numel() / V : 0;
    TORCH_CHECK(V < (1ll << 31), "vocabulary too large");
    check_status(rnnt_amd_log_softmax(current_stream(x), x.data_ptr<float>(), out.data_ptr<float>(), rows, (int)V));
    return out;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "MI355X-native RNN-T loss: compiled host binding over libwarp_rnnt_amd.so";
    m.def("rnnt_loss", &rnnt_loss, py::arg("xs"), py::arg("ys"), py::arg("xn"), py::arg("yn"), py::arg("blank") = 0,
          py::arg("fastemit_lambda") = 0.0, py::arg("want_mismatch") = false);
    m.def("rnnt_loss_gather", &rnnt_loss_gather, py::arg("xs"), py::arg("ys"), py::arg("xn"), py::arg("yn"),
          py::arg("blank") = 0, py::arg("fastemit_lambda") = 0.0, py::arg("want_mismatch") = false);
    m.def("rnnt_loss_gather_backward", &rnnt_loss_gather_backward, py::arg("grad_costs"), py::arg("grads_diagonal"),
          py::arg("ys"), py::arg("xn"), py::arg("yn"), py::arg("V"), py::arg("blank") = 0);
    // (names and keywords of the reference's module, binding.cpp:255-268)
    m.def("rnnt_loss_compact", &rnnt_loss_compact_forward, py::arg("xs"), py::arg("ys"), py::arg("xn"),
          py::arg("yn"), py::arg("blank") = 0, py::arg("fastemit_lambda") = 0.0, py::arg("required_grad") = true,
          py::arg("max_frames") = -1, py::arg("max_labels") = -1);
    m.def("rnnt_loss_compact_backward", &rnnt_loss_compact_backward, py::arg("grad_costs"), py::arg("grad_xs"),
          py::arg("cumSum"), py::arg("loc"), py::arg("V"), py::arg("blank") = 0);
    m.def("log_softmax", &log_softmax, py::arg("x"), py::arg("out") = py::none());
    m.def("library_version", []() { return rnnt_amd_version(); });
}
