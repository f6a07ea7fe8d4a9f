// torch::Library binding of the CSR family: pyg::segment_{sum,mean,min,max}_csr, pyg::gather_csr,
// pyg::softmax_csr(+_backward).  Schemas byte-identical to pyg_lib/csrc/ops/segment_csr.cpp:153-172
// and ops/softmax.cpp:46-53; argument checks follow the operator fronts (segment_csr.cpp:9-151,
// softmax.cpp:9-44) and the CPU kernels (ops/cpu/segment_csr_kernel.cpp); autograd formulas follow
// ops/autograd/segment_csr_kernel.cpp and ops/autograd/softmax_kernel.cpp.  Kernels: csrc/hip/csr.hip
// through the C-ABI of include/pyg_hip.h for HIP tensors, cpu_reduce.h for CPU tensors (key CPU, same fronts).
#include <torch/autograd.h>
#include <torch/library.h>

#include "binding_common.h"
#include "cpu_reduce.h"

namespace pyg_amd {
namespace {

using torch::autograd::variable_list;

enum { CSR_SUM = 0, CSR_MEAN = 1, CSR_MIN = 2, CSR_MAX = 3 };

struct CsrView {
  Tensor indptr;      // int64, contiguous: [rows + 1] (shared) or [leading, rows + 1]
  int64_t stride;     // elements between slices (0 = shared)
  int64_t leading, rows, dim;
};

// indptr is broadcast up to src.shape[:indptr.dim()-1] (segment_csr_kernel.cpp:44-52) -- read in place
// when every leading dimension is a broadcast one.
CsrView csr_view(const char* name, const Tensor& src, const Tensor& indptr) {
  TORCH_CHECK(src.device() == indptr.device(), name, ": src and indptr must be on the same device (got src=",
              src.device(), ", indptr=", indptr.device(), ")");
  TORCH_CHECK(src.is_cuda() || src.is_cpu(), name, ": tensors must live on the CPU or on a HIP device");
  TORCH_CHECK(src.dim() >= indptr.dim(), name, ": src.dim() must be >= indptr.dim() (got src.dim()=", src.dim(),
              ", indptr.dim()=", indptr.dim(), ")");
  CsrView v;
  v.dim = indptr.dim() - 1;
  TORCH_CHECK(v.dim >= 0, name, ": indptr must have at least 1 dimension");
  TORCH_CHECK(indptr.scalar_type() == at::kLong, name, ": indptr must be int64");
  v.rows = indptr.size(-1) - 1;
  v.leading = 1;
  bool shared = true;
  for (int64_t i = 0; i < v.dim; ++i) {
    v.leading *= src.size(i);
    shared = shared && (indptr.size(i) == 1 || indptr.stride(i) == 0);
  }
  if (shared) {
    auto row = indptr;
    for (int64_t i = 0; i < v.dim; ++i) row = row.select(0, 0);
    v.indptr = row.contiguous();
    v.stride = 0;
  } else {
    auto sizes = indptr.sizes().vec();
    for (int64_t i = 0; i < v.dim; ++i) sizes[i] = src.size(i);
    v.indptr = indptr.expand(sizes).contiguous();
    v.stride = v.rows + 1;
  }
  return v;
}

std::tuple<Tensor, Tensor> segment_any(int op, const char* name, const Tensor& src, const Tensor& indptr,
                                       const std::optional<Tensor>& optional_out) {
  PYG_TRACE("pyg::segment_csr");
  if (optional_out.has_value())
    TORCH_CHECK(src.device() == optional_out.value().device(), name, ": src and out must be on the same device (got src=",
                src.device(), ", out=", optional_out.value().device(), ")");
  if (op == CSR_MEAN)
    TORCH_CHECK(at::isFloatingType(src.scalar_type()), "\"", name, "_cpu\" not implemented for '", src.scalar_type(), "'");
  const auto v = csr_view(name, src, indptr);
  const bool on_cpu = src.is_cpu();
  std::optional<DeviceGuard> guard;
  if (!on_cpu) guard.emplace(src.device());
  auto src_c = src.contiguous();
  const int64_t dim = v.dim;
  const bool fresh = !optional_out.has_value();
  Tensor out;
  if (!fresh) {
    out = optional_out.value().contiguous();
    TORCH_CHECK(out.scalar_type() == src_c.scalar_type(), name, ": out must have the dtype of src");
    TORCH_CHECK(out.dim() == src_c.dim(), name, ": out must have as many dimensions as src");
    for (int64_t i = 0; i < out.dim(); ++i)
      if (i != dim) TORCH_CHECK(src_c.size(i) == out.size(i), name, ": out.size(", i, ") must match src.size(", i, ")");
    TORCH_CHECK(src_c.numel() == 0 || out.size(dim) == v.rows, name, ": out.size(dim) must equal indptr.size(-1) - 1");
    if (op == CSR_MEAN) out.zero_();  // mean overwrites (segment_csr_kernel.cpp:218-221)
  } else {
    auto sizes = src_c.sizes().vec();
    sizes[dim] = std::max<int64_t>(v.rows, 0);
    // a fresh sum / mean on the device is neither cleared nor read: the kernel writes every slot (fresh = 1 below)
    const bool needs_zero = (op == CSR_SUM || op == CSR_MEAN) && (on_cpu || src_c.numel() == 0);
    out = needs_zero ? at::zeros(sizes, src_c.options()) : at::empty(sizes, src_c.options());
  }
  Tensor arg;
  const int64_t E = src_c.size(dim);
  // (the device kernels write every arg slot -- rows without entries get the sentinel E -- so only the CPU key and the
  // empty source pre-fill it: K x 8 bytes per row that were written twice)
  if (op == CSR_MIN || op == CSR_MAX)
    arg = (on_cpu || src_c.numel() == 0) ? at::full(out.sizes(), E, v.indptr.options()) : at::empty(out.sizes(), v.indptr.options());
  if (src_c.numel() == 0) {
    if (fresh && (op == CSR_MIN || op == CSR_MAX)) out.fill_(0);
    return std::make_tuple(out, arg);
  }
  const int64_t N = out.size(dim) * v.leading;
  const int64_t K = N > 0 ? out.numel() / N : 0;
  if (on_cpu) {
    if (fresh && (op == CSR_MIN || op == CSR_MAX)) cpu::fill_identity(op == CSR_MIN ? PYG_REDUCE_MIN : PYG_REDUCE_MAX, out);
    cpu::segment_csr(op, src_c, v.indptr.data_ptr<int64_t>(), v.stride, out, arg.defined() ? arg.data_ptr<int64_t>() : nullptr,
                     v.leading, out.size(dim), E, K);
    // rows without entries keep the sentinel; a fresh output reads 0 there (segment_csr_kernel.cpp:400-404)
    if (fresh && (op == CSR_MIN || op == CSR_MAX)) out.masked_fill_(arg == E, 0);
    return std::make_tuple(out, arg);
  }
  const int code = dtype_code(src_c.scalar_type());
  void* stream = current_stream(src_c);
  // (fresh != 0: a min / max output starts from the identity inside the kernels without being read -- no pre-fill)
  // scratch for hub rows (0 bytes when no row can be one): their chunks are dealt to all workgroups
  const size_t hub_bytes = pyg_hip_csr_hub_workspace_size(op, code, v.leading, E, K);
  Tensor hub_ws;
  if (hub_bytes) hub_ws = at::empty({(int64_t)hub_bytes}, src_c.options().dtype(at::kByte));
  check_status(pyg_hip_segment_csr_ws(op, code, src_c.data_ptr(), v.indptr.data_ptr<int64_t>(), v.stride, out.data_ptr(),
                                      arg.defined() ? arg.data_ptr<int64_t>() : nullptr, fresh ? 1 : 0, v.leading,
                                      out.size(dim), E, K, hub_bytes ? hub_ws.data_ptr() : nullptr, hub_bytes, stream));
  return std::make_tuple(out, arg);
}

Tensor segment_sum_csr_kernel(const Tensor& src, const Tensor& indptr, const std::optional<Tensor>& out) {
  return std::get<0>(segment_any(CSR_SUM, "segment_sum_csr", src, indptr, out));
}
Tensor segment_mean_csr_kernel(const Tensor& src, const Tensor& indptr, const std::optional<Tensor>& out) {
  return std::get<0>(segment_any(CSR_MEAN, "segment_mean_csr", src, indptr, out));
}
std::tuple<Tensor, Tensor> segment_min_csr_kernel(const Tensor& src, const Tensor& indptr,
                                                  const std::optional<Tensor>& out) {
  return segment_any(CSR_MIN, "segment_min_csr", src, indptr, out);
}
std::tuple<Tensor, Tensor> segment_max_csr_kernel(const Tensor& src, const Tensor& indptr,
                                                  const std::optional<Tensor>& out) {
  return segment_any(CSR_MAX, "segment_max_csr", src, indptr, out);
}

Tensor gather_csr_kernel(const Tensor& src, const Tensor& indptr, const std::optional<Tensor>& optional_out) {
  PYG_TRACE("pyg::gather_csr");
  const char* name = "gather_csr";
  if (optional_out.has_value())
    TORCH_CHECK(src.device() == optional_out.value().device(), name, ": src and out must be on the same device (got src=",
                src.device(), ", out=", optional_out.value().device(), ")");
  const auto v = csr_view(name, src, indptr);
  const int64_t dim = v.dim;
  TORCH_CHECK(src.size(dim) == 0 || src.size(dim) == v.rows, name, ": src.size(dim) must equal indptr.size(-1) - 1");
  const bool on_cpu = src.is_cpu();
  std::optional<DeviceGuard> guard;
  if (!on_cpu) guard.emplace(src.device());
  auto src_c = src.contiguous();
  Tensor out;
  if (optional_out.has_value()) {
    out = optional_out.value().contiguous();
    TORCH_CHECK(out.scalar_type() == src_c.scalar_type(), name, ": out must have the dtype of src");
    TORCH_CHECK(out.dim() == src_c.dim(), name, ": out must have as many dimensions as src");
    for (int64_t i = 0; i < src_c.dim(); ++i)
      if (i != dim) TORCH_CHECK(src_c.size(i) == out.size(i), name, ": out.size(", i, ") must match src.size(", i, ")");
  } else {
    auto sizes = src_c.sizes().vec();
    // the output length is the last offset (one device -> host read, as in the reference)
    sizes[dim] = src_c.numel() > 0 ? v.indptr.flatten()[-1].item<int64_t>() : 0;
    out = at::empty(sizes, src_c.options());
  }
  if (src_c.numel() == 0) {
    if (!optional_out.has_value()) out.fill_(0);
    return out;
  }
  const int64_t N = v.rows * v.leading;
  const int64_t K = src_c.numel() / N;
  if (on_cpu) {
    cpu::gather_csr(src_c, v.indptr.data_ptr<int64_t>(), v.stride, out, v.leading, v.rows, out.size(dim), K);
    return out;
  }
  const int code = dtype_code(src_c.scalar_type());
  const size_t hub_bytes = pyg_hip_csr_hub_workspace_size(4, code, v.leading, out.size(dim), K);   // the chunk list of hub rows
  Tensor hub_ws;
  if (hub_bytes) hub_ws = at::empty({(int64_t)hub_bytes}, src_c.options().dtype(at::kByte));
  check_status(pyg_hip_gather_csr_ws(code, src_c.data_ptr(), v.indptr.data_ptr<int64_t>(), v.stride, out.data_ptr(), v.leading,
                                     v.rows, out.size(dim), K, hub_bytes ? hub_ws.data_ptr() : nullptr, hub_bytes,
                                     current_stream(src_c)));
  return out;
}

// ---- softmax_csr ----------------------------------------------------------------------------------
struct SoftmaxShape {
  int64_t outer, D, inner, dim;
};

SoftmaxShape softmax_shape(const char* name, const Tensor& src, const Tensor& ptr, int64_t dim) {
  TORCH_CHECK(src.is_contiguous(), name, ": Expected contiguous tensor, but got non-contiguous tensor for argument #0 'src'");
  TORCH_CHECK(ptr.is_contiguous(), name, ": Expected contiguous tensor, but got non-contiguous tensor for argument 'ptr'");
  TORCH_CHECK(src.device() == ptr.device() && (src.is_cuda() || src.is_cpu()), name,
              ": src and ptr must live on the same device (CPU or HIP)");
  TORCH_CHECK(ptr.scalar_type() == at::kLong && ptr.dim() == 1, name, ": ptr must be a 1-dimensional int64 tensor");
  TORCH_CHECK(src.scalar_type() == at::kFloat || src.scalar_type() == at::kDouble, "\"", name,
              "_kernel_impl\" not implemented for '", src.scalar_type(), "'");
  SoftmaxShape s;
  s.dim = dim < 0 ? dim + src.dim() : dim;
  TORCH_CHECK(s.dim >= 0 && s.dim < src.dim(), name, ": dim out of range");
  s.outer = 1;
  for (int64_t i = 0; i < s.dim; ++i) s.outer *= src.size(i);
  s.D = src.size(s.dim);
  s.inner = 1;
  for (int64_t i = s.dim + 1; i < src.dim(); ++i) s.inner *= src.size(i);
  return s;
}

Tensor softmax_csr_kernel(const Tensor& src, const Tensor& ptr, int64_t dim) {
  PYG_TRACE("pyg::softmax_csr");
  const auto s = softmax_shape("softmax_csr_forward", src, ptr, dim);
  auto out = at::zeros_like(src);
  if (src.is_cpu()) {
    cpu::softmax_csr(src, ptr.data_ptr<int64_t>(), out, s.outer, s.D, s.inner, ptr.numel() - 1);
    return out;
  }
  DeviceGuard guard(src.device());
  check_status(pyg_hip_softmax_csr(dtype_code(src.scalar_type()), src.data_ptr(), ptr.data_ptr<int64_t>(), out.data_ptr(),
                                   s.outer, s.D, s.inner, ptr.numel() - 1, current_stream(src)));
  return out;
}

Tensor softmax_csr_backward_kernel(const Tensor& out, const Tensor& out_grad, const Tensor& ptr, int64_t dim) {
  PYG_TRACE("pyg::softmax_csr_backward");
  const auto s = softmax_shape("softmax_csr_backward", out, ptr, dim);
  TORCH_CHECK(out_grad.is_contiguous() && out_grad.sizes() == out.sizes() && out_grad.scalar_type() == out.scalar_type() &&
                  out_grad.device() == out.device(),
              "softmax_csr_backward: out_grad must be a contiguous tensor shaped and typed like out");
  auto in_grad = at::zeros_like(out);
  if (out.is_cpu()) {
    cpu::softmax_csr_backward(out, out_grad, ptr.data_ptr<int64_t>(), in_grad, s.outer, s.D, s.inner, ptr.numel() - 1);
    return in_grad;
  }
  DeviceGuard guard(out.device());
  check_status(pyg_hip_softmax_csr_backward(dtype_code(out.scalar_type()), out.data_ptr(), out_grad.data_ptr(),
                                            ptr.data_ptr<int64_t>(), in_grad.data_ptr(), s.outer, s.D, s.inner,
                                            ptr.numel() - 1, current_stream(out)));
  return in_grad;
}

// ---- re-entry through the dispatcher (below Autograd) ---------------------------------------------
using UnarySig = Tensor(const Tensor&, const Tensor&, const std::optional<Tensor>&);
using PairSig = std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const std::optional<Tensor>&);
#define PYG_TYPED_OP(NAME, SIG) \
  static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("pyg::" NAME, "").typed<SIG>()

Tensor call_segment_sum_csr(const Tensor& src, const Tensor& indptr, const std::optional<Tensor>& out) {
  PYG_TYPED_OP("segment_sum_csr", UnarySig);
  return op.call(src, indptr, out);
}
Tensor call_segment_mean_csr(const Tensor& src, const Tensor& indptr, const std::optional<Tensor>& out) {
  PYG_TYPED_OP("segment_mean_csr", UnarySig);
  return op.call(src, indptr, out);
}
std::tuple<Tensor, Tensor> call_segment_min_csr(const Tensor& src, const Tensor& indptr, const std::optional<Tensor>& out) {
  PYG_TYPED_OP("segment_min_csr", PairSig);
  return op.call(src, indptr, out);
}
std::tuple<Tensor, Tensor> call_segment_max_csr(const Tensor& src, const Tensor& indptr, const std::optional<Tensor>& out) {
  PYG_TYPED_OP("segment_max_csr", PairSig);
  return op.call(src, indptr, out);
}
Tensor call_gather_csr(const Tensor& src, const Tensor& indptr, const std::optional<Tensor>& out) {
  PYG_TYPED_OP("gather_csr", UnarySig);
  return op.call(src, indptr, out);
}
Tensor call_softmax_csr(const Tensor& src, const Tensor& ptr, int64_t dim) {
  using SmSig = Tensor(const Tensor&, const Tensor&, int64_t);
  PYG_TYPED_OP("softmax_csr", SmSig);
  return op.call(src, ptr, dim);
}
Tensor call_softmax_csr_backward(const Tensor& out, const Tensor& out_grad, const Tensor& ptr, int64_t dim) {
  using SmbSig = Tensor(const Tensor&, const Tensor&, const Tensor&, int64_t);
  PYG_TYPED_OP("softmax_csr_backward", SmbSig);
  return op.call(out, out_grad, ptr, dim);
}

// ---- autograd (ops/autograd/segment_csr_kernel.cpp) -----------------------------------------------
class SegmentSumCSR : public torch::autograd::Function<SegmentSumCSR> {
 public:
  static variable_list forward(torch::autograd::AutogradContext* ctx, const Tensor& src, const Tensor& indptr,
                               const std::optional<Tensor>& optional_out) {
    at::AutoDispatchBelowADInplaceOrView g;
    auto out = call_segment_sum_csr(src, indptr, optional_out);
    ctx->save_for_backward({indptr});
    ctx->saved_data["src_shape"] = src.sizes();
    if (optional_out.has_value()) ctx->mark_dirty({optional_out.value()});
    return {out};
  }
  static variable_list backward(torch::autograd::AutogradContext* ctx, variable_list grad_outs) {
    const auto grad_out = grad_outs[0];
    const auto indptr = ctx->get_saved_variables()[0];
    auto src_shape = ctx->saved_data["src_shape"].toIntList().vec();
    auto grad_in = at::empty(src_shape, grad_out.options());
    // positions covered by no row must read 0 (the reference leaves them uninitialised)
    grad_in.zero_();
    grad_in = call_gather_csr(grad_out, indptr, grad_in);
    return {grad_in, Tensor(), Tensor()};
  }
};

class SegmentMeanCSR : public torch::autograd::Function<SegmentMeanCSR> {
 public:
  static variable_list forward(torch::autograd::AutogradContext* ctx, const Tensor& src, const Tensor& indptr,
                               const std::optional<Tensor>& optional_out) {
    at::AutoDispatchBelowADInplaceOrView g;
    auto out = call_segment_mean_csr(src, indptr, optional_out);
    ctx->save_for_backward({indptr});
    ctx->saved_data["src_shape"] = src.sizes();
    if (optional_out.has_value()) ctx->mark_dirty({optional_out.value()});
    return {out};
  }
  static variable_list backward(torch::autograd::AutogradContext* ctx, variable_list grad_outs) {
    const auto grad_out = grad_outs[0];
    const auto indptr = ctx->get_saved_variables()[0];
    auto src_shape = ctx->saved_data["src_shape"].toIntList().vec();
    auto grad_in = at::zeros(src_shape, grad_out.options());
    if (grad_in.numel() > 0) {
      grad_in = call_gather_csr(grad_out, indptr, grad_in);
      auto indptr1 = indptr.narrow(-1, 0, indptr.size(-1) - 1);
      auto indptr2 = indptr.narrow(-1, 1, indptr.size(-1) - 1);
      auto count = (indptr2 - indptr1).to(grad_in.options());
      // broadcast counts back to the source positions; uncovered positions divide by 1
      auto count_shape = std::vector<int64_t>(src_shape.begin(), src_shape.begin() + indptr.dim());
      auto rows_shape = count_shape;
      rows_shape.back() = count.size(-1);
      auto count_e = at::ones(count_shape, grad_in.options());
      count_e = call_gather_csr(count.expand(rows_shape), indptr, count_e);
      for (int64_t i = 0; i < grad_out.dim() - indptr.dim(); ++i) count_e = count_e.unsqueeze(-1);
      grad_in.true_divide_(count_e);
    }
    return {grad_in, Tensor(), Tensor()};
  }
};

template <bool IS_MIN>
class SegmentMinMaxCSR : public torch::autograd::Function<SegmentMinMaxCSR<IS_MIN>> {
 public:
  static variable_list forward(torch::autograd::AutogradContext* ctx, const Tensor& src, const Tensor& indptr,
                               const std::optional<Tensor>& optional_out) {
    at::AutoDispatchBelowADInplaceOrView g;
    const int64_t dim = indptr.dim() - 1;
    TORCH_CHECK(dim >= 0, IS_MIN ? "segment_min_csr" : "segment_max_csr", ": indptr must have at least 1 dimension");
    auto result = IS_MIN ? call_segment_min_csr(src, indptr, optional_out) : call_segment_max_csr(src, indptr, optional_out);
    auto out = std::get<0>(result);
    auto arg_out = std::get<1>(result);
    ctx->save_for_backward({indptr, arg_out});
    ctx->saved_data["dim"] = dim;
    ctx->saved_data["src_shape"] = src.sizes();
    ctx->mark_non_differentiable({arg_out});
    if (optional_out.has_value()) ctx->mark_dirty({optional_out.value()});
    return {out, arg_out};
  }
  static variable_list backward(torch::autograd::AutogradContext* ctx, variable_list grad_outs) {
    const auto grad_out = grad_outs[0];
    const auto arg_out = ctx->get_saved_variables()[1];
    const auto dim = ctx->saved_data["dim"].toInt();
    auto src_shape = ctx->saved_data["src_shape"].toIntList().vec();
    // one extra slot along `dim` swallows the sentinel of empty rows (segment_csr_kernel.cpp:146-150)
    src_shape[dim] += 1;
    auto grad_in = at::zeros(src_shape, grad_out.options());
    grad_in.scatter_(dim, arg_out, grad_out);
    grad_in = grad_in.narrow(dim, 0, src_shape[dim] - 1);
    return {grad_in, Tensor(), Tensor()};
  }
};

class GatherCSR : public torch::autograd::Function<GatherCSR> {
 public:
  static variable_list forward(torch::autograd::AutogradContext* ctx, const Tensor& src, const Tensor& indptr,
                               const std::optional<Tensor>& optional_out) {
    at::AutoDispatchBelowADInplaceOrView g;
    auto out = call_gather_csr(src, indptr, optional_out);
    ctx->save_for_backward({indptr});
    ctx->saved_data["src_shape"] = src.sizes();
    if (optional_out.has_value()) ctx->mark_dirty({optional_out.value()});
    return {out};
  }
  static variable_list backward(torch::autograd::AutogradContext* ctx, variable_list grad_outs) {
    const auto grad_out = grad_outs[0];
    const auto indptr = ctx->get_saved_variables()[0];
    auto src_shape = ctx->saved_data["src_shape"].toIntList().vec();
    auto grad_in = at::zeros(src_shape, grad_out.options());
    grad_in = call_segment_sum_csr(grad_out, indptr, /*out=*/grad_in);
    return {grad_in, Tensor(), Tensor()};
  }
};

class SoftmaxCSR : public torch::autograd::Function<SoftmaxCSR> {
 public:
  static variable_list forward(torch::autograd::AutogradContext* ctx, const Tensor& src, const Tensor& ptr, int64_t dim) {
    at::AutoDispatchBelowADInplaceOrView g;
    Tensor out = call_softmax_csr(src, ptr, dim);
    ctx->saved_data["dim"] = dim;
    ctx->save_for_backward({src, out, ptr});
    return {out};
  }
  static variable_list backward(torch::autograd::AutogradContext* ctx, variable_list out_grads) {
    const auto out_grad = out_grads[0];
    const auto saved = ctx->get_saved_variables();
    const auto src = saved[0], out = saved[1], ptr = saved[2];
    const auto dim = ctx->saved_data["dim"].toInt();
    Tensor src_grad;
    if (torch::autograd::any_variable_requires_grad({src}))
      src_grad = call_softmax_csr_backward(out, out_grad.contiguous(), ptr, dim);
    return {src_grad, Tensor(), Tensor()};
  }
};

Tensor segment_sum_csr_autograd(const Tensor& src, const Tensor& indptr, const std::optional<Tensor>& out) {
  return SegmentSumCSR::apply(src, indptr, out)[0];
}
Tensor segment_mean_csr_autograd(const Tensor& src, const Tensor& indptr, const std::optional<Tensor>& out) {
  return SegmentMeanCSR::apply(src, indptr, out)[0];
}
std::tuple<Tensor, Tensor> segment_min_csr_autograd(const Tensor& src, const Tensor& indptr,
                                                    const std::optional<Tensor>& out) {
  auto r = SegmentMinMaxCSR<true>::apply(src, indptr, out);
  return std::make_tuple(r[0], r[1]);
}
std::tuple<Tensor, Tensor> segment_max_csr_autograd(const Tensor& src, const Tensor& indptr,
                                                    const std::optional<Tensor>& out) {
  auto r = SegmentMinMaxCSR<false>::apply(src, indptr, out);
  return std::make_tuple(r[0], r[1]);
}
Tensor gather_csr_autograd(const Tensor& src, const Tensor& indptr, const std::optional<Tensor>& out) {
  return GatherCSR::apply(src, indptr, out)[0];
}
Tensor softmax_csr_autograd(const Tensor& src, const Tensor& ptr, int64_t dim) {
  return SoftmaxCSR::apply(src, ptr, dim)[0];
}

}  // namespace

// ops/segment_csr.cpp:153-172, ops/softmax.cpp:46-53
TORCH_LIBRARY_FRAGMENT(pyg, m) {
  m.def(TORCH_SELECTIVE_SCHEMA("pyg::segment_sum_csr(Tensor src, Tensor indptr, Tensor? out=None) -> Tensor"));
  m.def(TORCH_SELECTIVE_SCHEMA("pyg::segment_mean_csr(Tensor src, Tensor indptr, Tensor? out=None) -> Tensor"));
  m.def(TORCH_SELECTIVE_SCHEMA("pyg::segment_min_csr(Tensor src, Tensor indptr, Tensor? out=None) -> (Tensor, Tensor)"));
  m.def(TORCH_SELECTIVE_SCHEMA("pyg::segment_max_csr(Tensor src, Tensor indptr, Tensor? out=None) -> (Tensor, Tensor)"));
  m.def(TORCH_SELECTIVE_SCHEMA("pyg::gather_csr(Tensor src, Tensor indptr, Tensor? out=None) -> Tensor"));
  m.def(TORCH_SELECTIVE_SCHEMA("pyg::softmax_csr(Tensor src, Tensor ptr, int dim=0) -> Tensor"));
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::softmax_csr_backward(Tensor out, Tensor out_grad, Tensor ptr, int dim=0) -> Tensor"));
}

TORCH_LIBRARY_IMPL(pyg, CUDA, m) {
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_sum_csr"), TORCH_FN(segment_sum_csr_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_mean_csr"), TORCH_FN(segment_mean_csr_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_min_csr"), TORCH_FN(segment_min_csr_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_max_csr"), TORCH_FN(segment_max_csr_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::gather_csr"), TORCH_FN(gather_csr_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::softmax_csr"), TORCH_FN(softmax_csr_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::softmax_csr_backward"), TORCH_FN(softmax_csr_backward_kernel));
}

// key CPU (pyg_lib/csrc/ops/cpu/segment_csr_kernel.cpp:652-661, softmax_kernel.cpp:250-255)
TORCH_LIBRARY_IMPL(pyg, CPU, m) {
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_sum_csr"), TORCH_FN(segment_sum_csr_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_mean_csr"), TORCH_FN(segment_mean_csr_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_min_csr"), TORCH_FN(segment_min_csr_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_max_csr"), TORCH_FN(segment_max_csr_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::gather_csr"), TORCH_FN(gather_csr_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::softmax_csr"), TORCH_FN(softmax_csr_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::softmax_csr_backward"), TORCH_FN(softmax_csr_backward_kernel));
}

TORCH_LIBRARY_IMPL(pyg, Autograd, m) {
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_sum_csr"), TORCH_FN(segment_sum_csr_autograd));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_mean_csr"), TORCH_FN(segment_mean_csr_autograd));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_min_csr"), TORCH_FN(segment_min_csr_autograd));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_max_csr"), TORCH_FN(segment_max_csr_autograd));
  m.impl(TORCH_SELECTIVE_NAME("pyg::gather_csr"), TORCH_FN(gather_csr_autograd));
  m.impl(TORCH_SELECTIVE_NAME("pyg::softmax_csr"), TORCH_FN(softmax_csr_autograd));
}

}  // namespace pyg_amd
