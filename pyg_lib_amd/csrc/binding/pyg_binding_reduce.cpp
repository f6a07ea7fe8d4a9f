// libpyg.so, part 2: index_sort, scatter_*, segment_*_coo and gather_coo.
//
// Schemas are the reference's (cited per op); every kernel is a call into the C-ABI of
// include/pyg_hip.h.  The autograd wrappers restate the reference's formulas
// (pyg_lib/csrc/ops/autograd/scatter_kernel.cpp, segment_coo_kernel.cpp) on top of these ops.
#include <ATen/core/dispatch/Dispatcher.h>
#include <torch/autograd.h>
#include <ATen/Context.h>
#include <torch/library.h>

#include <optional>
#include <tuple>
#include <vector>

#include "binding_common.h"
#include "cpu_reduce.h"

namespace pyg_amd {

using torch::autograd::variable_list;

// ---------------------------------------------------------------------------------------------
// index_sort  (pyg_lib/csrc/ops/index_sort.cpp:9-28, ops/cpu/index_sort_kernel.cpp:14-59)
// ---------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor> index_sort_kernel(const Tensor& input, const at::optional<int64_t> max) {
  PYG_TRACE("pyg::index_sort");
  TORCH_CHECK(input.is_contiguous(), "Input should be contiguous.");
  TORCH_CHECK(input.dim() == 1, "Input should be 1-dimensional.");
  TORCH_CHECK(at::isIntegralType(input.scalar_type(), /*includeBool=*/false),
              "Input should contain integral values.");
  DeviceGuard guard(input.device());
  const int64_t n = input.numel();
  auto vals = at::empty_like(input);
  auto idx = at::empty({n}, input.options().dtype(at::kLong));
  if (n == 0) return std::make_tuple(vals, idx);
  const int dt = dtype_code(input.scalar_type());
  auto ws = at::empty({(int64_t)pyg_hip_index_sort_workspace_size(dt, n)}, input.options().dtype(at::kByte));
  check_status(pyg_hip_index_sort(dt, input.data_ptr(), n, max.value_or(0), max.has_value() ? 1 : 0,
                                  vals.data_ptr(), idx.data_ptr<int64_t>(), ws.data_ptr(), (size_t)ws.numel(),
                                  current_stream(input)));
  return std::make_tuple(vals, idx);
}

// ---------------------------------------------------------------------------------------------
// shared (B, E, K) plumbing
// ---------------------------------------------------------------------------------------------
// pyg_lib/csrc/ops/utils.h:22-34
static Tensor broadcast(const Tensor& src, const Tensor& other, int64_t dim) {
  auto out = src;
  if (out.dim() == 1)
    for (int64_t i = 0; i < dim; ++i) out = out.unsqueeze(0);
  for (int64_t i = out.dim(); i < other.dim(); ++i) out = out.unsqueeze(-1);
  return out.expand(other.sizes());
}

// Collapse dims [b, e) of a strided view into one stride over their flattened index; false if the
// view is not expressible that way (then the index is materialised).
static bool collapse(const Tensor& t, int64_t b, int64_t e, int64_t* stride) {
  int64_t st = 0, expect = 0;
  bool have = false;
  for (int64_t i = e - 1; i >= b; --i) {
    const int64_t sz = t.size(i), s = t.stride(i);
    if (sz == 1) continue;
    if (!have) {
      st = s;
      have = true;
    } else if (s != expect) {
      return false;
    }
    expect = s * sz;
  }
  *stride = st;
  return true;
}

struct Layout {
  int64_t B, E, K, N;
  int64_t isb, ise, isk;
  Tensor index;  // keeps the (possibly re-materialised) index alive
};

// index has src.dim() dims (already broadcast, usually an expanded view)
static Layout scatter_layout(const Tensor& src, const Tensor& index, int64_t dim) {
  Layout l;
  l.B = 1;
  for (int64_t i = 0; i < dim; ++i) l.B *= src.size(i);
  l.E = src.size(dim);
  l.K = 1;
  for (int64_t i = dim + 1; i < src.dim(); ++i) l.K *= src.size(i);
  l.index = index;
  int64_t sb = 0, sk = 0;
  const bool ok = index.sizes() == src.sizes() && collapse(index, 0, dim, &sb) &&
                  collapse(index, dim + 1, index.dim(), &sk);
  if (ok) {
    l.isb = sb;
    l.ise = index.size(dim) == 1 ? 0 : index.stride(dim);
    l.isk = sk;
  } else {
    l.index = index.expand(src.sizes()).contiguous();
    l.isb = l.E * l.K;
    l.ise = l.K;
    l.isk = 1;
  }
  return l;
}

enum { OP_SUM = PYG_REDUCE_SUM, OP_MUL = PYG_REDUCE_MUL, OP_MIN = PYG_REDUCE_MIN, OP_MAX = PYG_REDUCE_MAX };

static const char* op_name(int op, bool coo) {
  static const char* s[] = {"scatter_sum", "scatter_mul", "scatter_min", "scatter_max"};
  static const char* c[] = {"segment_sum_coo", "segment_mul_coo", "segment_min_coo", "segment_max_coo"};
  return coo ? c[op] : s[op];
}

// Core of scatter_* and segment_*_coo: `index_b` has src.dim() dims.
// Set around scatter_mean's count: one unsorted index vector takes the stable sort + CSR-row path for ANY row width.
static thread_local bool tl_prefer_sorted_sum = false;

static std::tuple<Tensor, Tensor> reduce_core(int op, bool coo, const Tensor& src, const Tensor& index_b, int64_t dim,
                                              const std::optional<Tensor>& optional_out,
                                              std::optional<int64_t> dim_size, int64_t inferred_size) {
  PYG_TRACE("pyg::scatter_or_segment_coo");
  const char* name = op_name(op, coo);
  TORCH_CHECK(src.device() == index_b.device(), name, ": src and index must be on the same device (got src=", src.device(),
              ", index=", index_b.device(), ")");
  TORCH_CHECK(index_b.scalar_type() == at::kLong, name, ": index must be int64");
  const bool on_cpu = src.is_cpu();  // dispatch key CPU: same front, kernels of cpu_reduce.h instead of the C-ABI
  TORCH_CHECK(on_cpu || src.is_cuda(), name, ": tensors must live on the CPU or on a HIP device");
  std::optional<DeviceGuard> guard;
  if (!on_cpu) guard.emplace(src.device());
  auto src_c = src.contiguous();
  Tensor out;
  const bool fresh = !optional_out.has_value();
  if (!fresh) {
    out = optional_out.value();
    TORCH_CHECK(out.is_contiguous(), name, ": 'out' must be contiguous");
    TORCH_CHECK(out.device() == src.device(), name, ": src and out must be on the same device");
    TORCH_CHECK(out.scalar_type() == src.scalar_type(), name, ": 'out' must have the dtype of 'src'");
    TORCH_CHECK(out.dim() == src.dim(), name, ": out.dim() must match src.dim()");
    for (int64_t i = 0; i < out.dim(); ++i)
      if (i != dim) TORCH_CHECK(src_c.size(i) == out.size(i), name, ": out.size(", i, ") must match src.size(", i, ")");
  } else {
    auto sizes = src_c.sizes().vec();
    sizes[dim] = dim_size.has_value() ? dim_size.value() : inferred_size;
    // a fresh sum on the device is handed over uninitialised (PYG_HIP_SCATTER_FRESH_SUM): the sorted path writes every
    // slot without reading it, the other paths clear it themselves
    if (op == OP_SUM) out = (on_cpu || src_c.numel() == 0) ? at::zeros(sizes, src_c.options()) : at::empty(sizes, src_c.options());
    else if (op == OP_MUL) out = at::ones(sizes, src_c.options());
    else out = at::empty(sizes, src_c.options());
  }
  const int dt = dtype_code(src_c.scalar_type());
  void* stream = on_cpu ? nullptr : current_stream(src_c);
  Tensor arg, init;
  const bool minmax = op == OP_MIN || op == OP_MAX;
  if (minmax && on_cpu) {
    arg = at::full(out.sizes(), src_c.size(dim), index_b.options().dtype(at::kLong));  // sentinel: no source position
    if (fresh) cpu::fill_identity(op, out);
  } else if (minmax) {
    arg = at::empty(out.sizes(), index_b.options().dtype(at::kLong));
    if (fresh) check_status(pyg_hip_fill_reduce_identity(op, dt, out.data_ptr(), out.numel(), stream));
    else init = out.clone();
  }
  Layout l = scatter_layout(src_c, index_b, dim);
  l.N = out.size(dim);
  if (src_c.numel() == 0) {
    if (minmax) {
      arg.fill_(src_c.size(dim));
      if (fresh) out.fill_(0);
    }
    return std::make_tuple(out, arg);
  }
  if (on_cpu) {
    cpu::scatter(op, src_c, l.index.data_ptr<int64_t>(), l.isb, l.ise, l.isk, out, minmax ? arg.data_ptr<int64_t>() : nullptr,
                 l.B, l.E, l.K, l.N, coo);
    // buckets nobody wrote keep the sentinel; a fresh output reads 0 there (scatter_kernel.cpp:358-366)
    if (minmax && fresh) out.masked_fill_(arg == src_c.size(dim), 0);
    return std::make_tuple(out, arg);
  }
  // scratch for the sort-based sum (large unsorted float scatters only; see pyg_hip_scatter)
  Tensor ws;
  const bool sort_sum = op == OP_SUM && !coo && l.B == 1 && l.isk == 0 && l.ise == 1 && l.E >= (1 << 15) &&
                        at::isFloatingType(src_c.scalar_type()) && l.K * (int64_t)src_c.element_size() >= 64;
  // min / max: atomic-free CSR walk for a sorted (COO) index or one large unsorted index vector
  const bool csr_minmax = minmax && l.isk == 0 && (coo || (l.B == 1 && l.ise == 1 && l.E >= (1 << 15)));
  const bool csr_sum = op == OP_SUM && coo && l.isk == 0;  // sorted index: atomic-free CSR row sums
  // torch.use_deterministic_algorithms(True): floating sums / products must not go through atomics (their result would
  // depend on the order the adds land in).  Sums have an atomic-free kernel for a sorted index and for one unsorted index
  // vector of ANY size (stable sort + CSR rows, source order); where there is none, torch's own convention applies:
  // alertNotDeterministic raises, or warns under warn_only and the atomic kernel runs.
  const bool floating = at::isFloatingType(src_c.scalar_type());
  // (tl_prefer_sorted_sum: scatter_mean's bucket sizes -- see ScatterMean -- ask for the same path without the mode)
  const bool prefer_sorted = tl_prefer_sorted_sum && op == OP_SUM && floating && !coo && l.B == 1 && l.isk == 0 && l.ise == 1 &&
                             !at::globalContext().deterministicAlgorithms();
  bool det = (at::globalContext().deterministicAlgorithms() || prefer_sorted) && floating && (op == OP_SUM || op == OP_MUL);
  const bool det_sort = det && op == OP_SUM && !coo && l.B == 1 && l.isk == 0 && l.ise == 1;
  if (sort_sum || csr_minmax || csr_sum || det_sort)
    ws = at::empty({(int64_t)pyg_hip_scatter_workspace_size(l.B, l.E, l.N)}, src_c.options().dtype(at::kByte));
  const int base_flags = (coo ? PYG_HIP_SCATTER_SORTED : 0) | (fresh && op == OP_SUM ? PYG_HIP_SCATTER_FRESH_SUM : 0);
  int rc = pyg_hip_scatter(op, dt, src_c.data_ptr(), l.index.data_ptr<int64_t>(), l.isb, l.ise, l.isk, out.data_ptr(),
                           minmax ? arg.data_ptr<int64_t>() : nullptr, init.defined() ? init.data_ptr() : nullptr, l.B, l.E,
                           l.K, l.N, base_flags | (det ? PYG_HIP_SCATTER_DETERMINISTIC : 0),
                           ws.defined() ? ws.data_ptr() : nullptr, ws.defined() ? (size_t)ws.numel() : 0, stream);
  if (rc == PYG_HIP_ERR_UNSUPPORTED && det) {
    if (!prefer_sorted) at::globalContext().alertNotDeterministic(
        op == OP_MUL ? "pyg::scatter_mul on floating-point HIP tensors"
                     : "pyg::scatter_sum / scatter_mean on HIP tensors with an element-wise (or batched unsorted) index");
    rc = pyg_hip_scatter(op, dt, src_c.data_ptr(), l.index.data_ptr<int64_t>(), l.isb, l.ise, l.isk, out.data_ptr(),
                         minmax ? arg.data_ptr<int64_t>() : nullptr, init.defined() ? init.data_ptr() : nullptr, l.B, l.E,
                         l.K, l.N, base_flags, ws.defined() ? ws.data_ptr() : nullptr,
                         ws.defined() ? (size_t)ws.numel() : 0, stream);
  }
  check_status(rc);
  return std::make_tuple(out, arg);
}

static int64_t normalize_dim(const char* name, int64_t dim, const Tensor& src) {
  dim = dim < 0 ? src.dim() + dim : dim;
  TORCH_CHECK(dim >= 0 && dim < src.dim(), name, ": dim out of range");
  return dim;
}

static int64_t infer_scatter_size(const Tensor& index) {
  // scatter_kernel.cpp:55-61
  return index.numel() == 0 ? 0 : 1 + index.max().item<int64_t>();
}

// ---------------------------------------------------------------------------------------------
// scatter_{sum,mul,min,max}  (pyg_lib/csrc/ops/scatter.cpp:156-172)
// ---------------------------------------------------------------------------------------------
static std::tuple<Tensor, Tensor> scatter_any(int op, const Tensor& src, const Tensor& index, int64_t dim,
                                              const std::optional<Tensor>& out, std::optional<int64_t> dim_size) {
  const char* name = op_name(op, false);
  TORCH_CHECK(src.dim() == index.dim(), name, ": src.dim() must equal index.dim() after broadcasting (got src.dim()=",
              src.dim(), ", index.dim()=", index.dim(), ")");
  dim = normalize_dim(name, dim, src);
  const int64_t inferred = (out.has_value() || dim_size.has_value()) ? 0 : infer_scatter_size(index);
  return reduce_core(op, false, src, index, dim, out, dim_size, inferred);
}

Tensor scatter_sum_kernel(const Tensor& src, const Tensor& index, int64_t dim, const std::optional<Tensor>& out,
                          std::optional<int64_t> dim_size) {
  return std::get<0>(scatter_any(OP_SUM, src, index, dim, out, dim_size));
}
Tensor scatter_mul_kernel(const Tensor& src, const Tensor& index, int64_t dim, const std::optional<Tensor>& out,
                          std::optional<int64_t> dim_size) {
  return std::get<0>(scatter_any(OP_MUL, src, index, dim, out, dim_size));
}
std::tuple<Tensor, Tensor> scatter_min_kernel(const Tensor& src, const Tensor& index, int64_t dim,
                                              const std::optional<Tensor>& out, std::optional<int64_t> dim_size) {
  return scatter_any(OP_MIN, src, index, dim, out, dim_size);
}
std::tuple<Tensor, Tensor> scatter_max_kernel(const Tensor& src, const Tensor& index, int64_t dim,
                                              const std::optional<Tensor>& out, std::optional<int64_t> dim_size) {
  return scatter_any(OP_MAX, src, index, dim, out, dim_size);
}

// ---------------------------------------------------------------------------------------------
// segment_{sum,mean,min,max}_coo, gather_coo  (pyg_lib/csrc/ops/segment_coo.cpp:150-165)
// ---------------------------------------------------------------------------------------------
static Tensor coo_index_view(const char* name, const Tensor& src, const Tensor& index) {
  TORCH_CHECK(src.dim() >= index.dim(), name, ": src.dim() must be >= index.dim() (got src.dim()=", src.dim(),
              ", index.dim()=", index.dim(), ")");
  TORCH_CHECK(index.dim() >= 1, name, ": index must have at least 1 dimension");
  // broadcast index up to src.shape[:index.dim()] (segment_coo_kernel.cpp:47-52), then give it
  // src.dim() dims by trailing broadcast -- all as views, nothing is materialised.
  auto sizes = index.sizes().vec();
  for (int64_t i = 0; i < index.dim(); ++i) sizes[i] = src.size(i);
  auto ib = index.expand(sizes);
  for (int64_t i = ib.dim(); i < src.dim(); ++i) ib = ib.unsqueeze(-1);
  return ib.expand(src.sizes());
}

static int64_t infer_coo_size(const Tensor& index, const Tensor& src) {
  // segment_coo_kernel.cpp:66-73: last index of the sorted last row(s)
  if (index.numel() == 0 || src.numel() == 0) return 0;
  const int64_t dim = index.dim() - 1;
  auto sizes = index.sizes().vec();
  for (int64_t i = 0; i < index.dim(); ++i) sizes[i] = src.size(i);
  auto ib = index.expand(sizes);
  auto tmp = ib.select(dim, ib.size(dim) - 1);
  tmp = tmp.numel() > 1 ? tmp.max() : tmp;
  return 1 + tmp.reshape({-1})[0].item<int64_t>();
}

static std::tuple<Tensor, Tensor> segment_any(int op, const Tensor& src, const Tensor& index,
                                              const std::optional<Tensor>& out, std::optional<int64_t> dim_size) {
  const char* name = op_name(op, true);
  auto ib = coo_index_view(name, src, index);
  const int64_t dim = index.dim() - 1;
  const int64_t inferred = (out.has_value() || dim_size.has_value()) ? 0 : infer_coo_size(index, src);
  return reduce_core(op, true, src, ib, dim, out, dim_size, inferred);
}

Tensor segment_sum_coo_kernel(const Tensor& src, const Tensor& index, const std::optional<Tensor>& out,
                              std::optional<int64_t> dim_size) {
  return std::get<0>(segment_any(OP_SUM, src, index, out, dim_size));
}
std::tuple<Tensor, Tensor> segment_min_coo_kernel(const Tensor& src, const Tensor& index,
                                                  const std::optional<Tensor>& out, std::optional<int64_t> dim_size) {
  return segment_any(OP_MIN, src, index, out, dim_size);
}
std::tuple<Tensor, Tensor> segment_max_coo_kernel(const Tensor& src, const Tensor& index,
                                                  const std::optional<Tensor>& out, std::optional<int64_t> dim_size) {
  return segment_any(OP_MAX, src, index, out, dim_size);
}

// segment_mean_coo (segment_coo_kernel.cpp:187-331): per-bucket sum and count, divide; buckets touched
// by `index` are OVERWRITTEN, untouched ones keep the caller's value (or 0 for a fresh output).
Tensor segment_mean_coo_kernel(const Tensor& src, const Tensor& index, const std::optional<Tensor>& optional_out,
                               std::optional<int64_t> dim_size) {
  const char* name = "segment_mean_coo";
  TORCH_CHECK(at::isFloatingType(src.scalar_type()), name, ": floating point 'src' expected");
  auto ib = coo_index_view(name, src, index);
  const int64_t dim = index.dim() - 1;
  std::optional<int64_t> n = dim_size;
  if (optional_out.has_value()) n = optional_out.value().size(dim);
  else if (!n.has_value()) n = infer_coo_size(index, src);
  auto sum = std::get<0>(reduce_core(OP_SUM, true, src, ib, dim, std::nullopt, n, 0));
  // count per (b, bucket): scatter ones over the [B, E] index
  auto sizes = index.sizes().vec();
  for (int64_t i = 0; i < index.dim(); ++i) sizes[i] = src.size(i);
  auto idx_be = index.expand(sizes);
  auto ones = at::ones(sizes, src.options());
  auto count = std::get<0>(reduce_core(OP_SUM, true, ones, idx_be, dim, std::nullopt, n, 0));
  auto touched = count > 0;
  count.masked_fill_(count < 1, 1);
  auto count_b = count;
  auto touched_b = touched;
  for (int64_t i = 0; i < src.dim() - index.dim(); ++i) {
    count_b = count_b.unsqueeze(-1);
    touched_b = touched_b.unsqueeze(-1);
  }
  sum.div_(count_b);
  if (!optional_out.has_value()) return sum;
  auto out = optional_out.value();
  out.copy_(at::where(touched_b, sum, out));
  return out;
}

Tensor gather_coo_kernel(const Tensor& src, const Tensor& index, const std::optional<Tensor>& optional_out) {
  PYG_TRACE("pyg::gather_coo");
  const char* name = "gather_coo";
  TORCH_CHECK(src.dim() >= index.dim(), name, ": src.dim() must be >= index.dim() (got src.dim()=", src.dim(),
              ", index.dim()=", index.dim(), ")");
  const int64_t dim = index.dim() - 1;
  TORCH_CHECK(dim >= 0, name, ": index must have at least 1 dimension");
  for (int64_t i = 0; i < dim; ++i)
    TORCH_CHECK(src.size(i) == index.size(i), name, ": src.size(", i, ") must match index.size(", i, ")");
  TORCH_CHECK(src.device() == index.device(), name, ": src and index must be on the same device");
  TORCH_CHECK(index.scalar_type() == at::kLong, name, ": index must be int64");
  const bool on_cpu = src.is_cpu();
  TORCH_CHECK(on_cpu || src.is_cuda(), name, ": tensors must live on the CPU or on a HIP device");
  std::optional<DeviceGuard> guard;
  if (!on_cpu) guard.emplace(src.device());
  auto src_c = src.contiguous();
  auto index_c = index.contiguous();
  Tensor out;
  if (optional_out.has_value()) {
    out = optional_out.value();
    TORCH_CHECK(out.is_contiguous(), name, ": 'out' must be contiguous");
    TORCH_CHECK(out.device() == src.device() && out.scalar_type() == src.scalar_type(), name,
                ": 'out' must live on the device of 'src' and have its dtype");
    for (int64_t i = 0; i < src_c.dim(); ++i)
      if (i != dim) TORCH_CHECK(src_c.size(i) == out.size(i), name, ": out.size(", i, ") must match src.size(", i, ")");
  } else {
    auto sizes = src_c.sizes().vec();
    sizes[dim] = index_c.size(dim);
    out = at::empty(sizes, src_c.options());
  }
  if (src_c.numel() == 0 || index_c.numel() == 0) {
    if (!optional_out.has_value()) out.fill_(0);
    return out;
  }
  const int64_t E = index_c.size(dim);
  int64_t B = 1;
  for (int64_t i = 0; i < dim; ++i) B *= index_c.size(i);
  const int64_t K = out.numel() / index_c.numel();
  const int64_t N = src_c.size(dim);
  if (on_cpu) {
    cpu::gather_coo(src_c, index_c.data_ptr<int64_t>(), out, B, E, K, N);
    return out;
  }
  check_status(pyg_hip_gather_coo(dtype_code(src_c.scalar_type()), src_c.data_ptr(), index_c.data_ptr<int64_t>(),
                                  out.data_ptr(), B, E, K, N, current_stream(src_c)));
  return out;
}

// ---------------------------------------------------------------------------------------------
// dispatcher re-entry helpers
// ---------------------------------------------------------------------------------------------
using ScatterFn = Tensor(const Tensor&, const Tensor&, int64_t, const std::optional<Tensor>&, std::optional<int64_t>);
using ScatterArgFn = std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, int64_t, const std::optional<Tensor>&,
                                                std::optional<int64_t>);
using CooFn = Tensor(const Tensor&, const Tensor&, const std::optional<Tensor>&, std::optional<int64_t>);
using CooArgFn = std::tuple<Tensor, Tensor>(const Tensor&, const Tensor&, const std::optional<Tensor>&,
                                            std::optional<int64_t>);
using GatherFn = Tensor(const Tensor&, const Tensor&, const std::optional<Tensor>&);

template <typename Fn>
static auto find_op(const char* name) {
  return c10::Dispatcher::singleton().findSchemaOrThrow(name, "").typed<Fn>();
}

// ---------------------------------------------------------------------------------------------
// autograd (pyg_lib/csrc/ops/autograd/scatter_kernel.cpp, segment_coo_kernel.cpp)
// ---------------------------------------------------------------------------------------------
template <int OP>
class ScatterFwdBwd : public torch::autograd::Function<ScatterFwdBwd<OP>> {
 public:
  static variable_list forward(torch::autograd::AutogradContext* ctx, const Tensor& src, const Tensor& index,
                               int64_t dim, const std::optional<Tensor>& optional_out,
                               std::optional<int64_t> dim_size) {
    at::AutoDispatchBelowADInplaceOrView g;
    const int64_t dim_norm = dim < 0 ? src.dim() + dim : dim;
    auto index_b = broadcast(index, src, dim_norm);
    ctx->saved_data["dim"] = dim_norm;
    if (optional_out.has_value()) ctx->mark_dirty({optional_out.value()});
    if (OP == OP_SUM) {
      static auto op = find_op<ScatterFn>("pyg::scatter_sum");
      auto out = op.call(src, index_b, dim_norm, optional_out, dim_size);
      ctx->save_for_backward({index_b});
      return {out};
    } else if (OP == OP_MUL) {
      static auto op = find_op<ScatterFn>("pyg::scatter_mul");
      auto out = op.call(src, index_b, dim_norm, optional_out, dim_size);
      ctx->save_for_backward({src, index_b, out});
      return {out};
    } else {
      static auto opmin = find_op<ScatterArgFn>("pyg::scatter_min");
      static auto opmax = find_op<ScatterArgFn>("pyg::scatter_max");
      auto res = (OP == OP_MIN ? opmin : opmax).call(src, index_b, dim_norm, optional_out, dim_size);
      auto out = std::get<0>(res);
      auto arg = std::get<1>(res);
      ctx->save_for_backward({arg});
      ctx->saved_data["src_shape"] = src.sizes();
      ctx->mark_non_differentiable({arg});
      return {out, arg};
    }
  }

  static variable_list backward(torch::autograd::AutogradContext* ctx, variable_list grad_outs) {
    const auto grad_out = grad_outs[0];
    const auto saved = ctx->get_saved_variables();
    const auto dim = ctx->saved_data["dim"].toInt();
    Tensor grad_src;
    if (OP == OP_SUM) {
      grad_src = grad_out.gather(dim, saved[0]);
    } else if (OP == OP_MUL) {
      const auto& src = saved[0];
      auto gathered = (grad_out * saved[2]).gather(dim, saved[1]);
      grad_src = at::where(src != 0, gathered / src, at::zeros_like(src));
    } else {
      auto shape = ctx->saved_data["src_shape"].toIntList().vec();
      shape[dim] += 1;
      auto grad_in = at::zeros(shape, grad_out.options());
      grad_in.scatter_(dim, saved[0], grad_out);
      grad_src = grad_in.narrow(dim, 0, shape[dim] - 1);
    }
    return {grad_src, Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

Tensor scatter_sum_autograd(const Tensor& src, const Tensor& index, int64_t dim, const std::optional<Tensor>& out,
                            std::optional<int64_t> dim_size) {
  return ScatterFwdBwd<OP_SUM>::apply(src, index, dim, out, dim_size)[0];
}
Tensor scatter_mul_autograd(const Tensor& src, const Tensor& index, int64_t dim, const std::optional<Tensor>& out,
                            std::optional<int64_t> dim_size) {
  return ScatterFwdBwd<OP_MUL>::apply(src, index, dim, out, dim_size)[0];
}
std::tuple<Tensor, Tensor> scatter_min_autograd(const Tensor& src, const Tensor& index, int64_t dim,
                                                const std::optional<Tensor>& out, std::optional<int64_t> dim_size) {
  auto r = ScatterFwdBwd<OP_MIN>::apply(src, index, dim, out, dim_size);
  return std::make_tuple(r[0], r[1]);
}
std::tuple<Tensor, Tensor> scatter_max_autograd(const Tensor& src, const Tensor& index, int64_t dim,
                                                const std::optional<Tensor>& out, std::optional<int64_t> dim_size) {
  auto r = ScatterFwdBwd<OP_MAX>::apply(src, index, dim, out, dim_size);
  return std::make_tuple(r[0], r[1]);
}

// scatter_mean: composite of two scatter_sum calls and a division
// (ops/autograd/scatter_kernel.cpp:161-233; floor division for integer dtypes).
class ScatterMean : public torch::autograd::Function<ScatterMean> {
 public:
  static variable_list forward(torch::autograd::AutogradContext* ctx, const Tensor& src, const Tensor& index,
                               int64_t dim, const std::optional<Tensor>& optional_out,
                               std::optional<int64_t> dim_size) {
    at::AutoDispatchBelowADInplaceOrView g;
    static auto op = find_op<ScatterFn>("pyg::scatter_sum");
    const int64_t dim_norm = dim < 0 ? src.dim() + dim : dim;
    auto index_b = broadcast(index, src, dim_norm);
    auto out = op.call(src, index_b, dim_norm, optional_out, dim_size);
    const int64_t count_dim = index.dim() <= dim_norm ? index.dim() - 1 : dim_norm;
    // The bucket sizes are a K = 1 scatter of ones: through atomics, ONE popular destination serialises them (8 M edges, 2.5 %
    // of them into one bucket: 5 ms in float32, 43 ms through the 16-bit compare-and-swap loop, of a 1.5 ms call).  On the
    // device a 16-bit source is counted in float32 (native atomics; a count above 256 is the true count rounded to the storage
    // type afterwards, where `+= 1` in bf16 -- the reference's and the 16-bit atomic path's -- stops at 256), and an index
    // vector of >= 4 M entries through its stable sort (row sums of ones, no atomics: 20 % slower than the atomics without
    // a popular bucket from that size on, 2 x slower at 1 M entries; `tools/narrow_scatter_paths.py`).
    const bool half = src.is_cuda() && (src.scalar_type() == at::kBFloat16 || src.scalar_type() == at::kHalf);
    auto ones = at::ones(index.sizes(), half ? src.options().dtype(at::kFloat) : src.options());
    struct Prefer {
      bool old;
      explicit Prefer(bool on) : old(tl_prefer_sorted_sum) { tl_prefer_sorted_sum = on; }
      ~Prefer() { tl_prefer_sorted_sum = old; }
    } prefer(src.is_cuda() && index.dim() == 1 && index.numel() >= (1 << 22) && at::isFloatingType(src.scalar_type()));
    auto count = op.call(ones, index, count_dim, std::nullopt, out.size(dim_norm));
    if (half) count = count.to(src.scalar_type());
    count.masked_fill_(count < 1, 1);
    auto count_b = broadcast(count, out, dim_norm);
    if (out.is_floating_point()) out.true_divide_(count_b);
    else out.div_(count_b, "floor");
    ctx->save_for_backward({index_b, count_b});
    ctx->saved_data["dim"] = dim_norm;
    if (optional_out.has_value()) ctx->mark_dirty({optional_out.value()});
    return {out};
  }
  static variable_list backward(torch::autograd::AutogradContext* ctx, variable_list grad_outs) {
    const auto saved = ctx->get_saved_variables();
    const auto dim = ctx->saved_data["dim"].toInt();
    auto count = saved[1].gather(dim, saved[0]);
    auto grad_src = grad_outs[0].gather(dim, saved[0]).true_divide_(count);
    return {grad_src, Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

Tensor scatter_mean_autograd(const Tensor& src, const Tensor& index, int64_t dim, const std::optional<Tensor>& out,
                             std::optional<int64_t> dim_size) {
  return ScatterMean::apply(src, index, dim, out, dim_size)[0];
}

// ---- COO family ----
static Tensor expand_coo_index(const Tensor& src, const Tensor& index) {
  auto sizes = index.sizes().vec();
  for (int64_t i = 0; i < index.dim(); ++i) sizes[i] = src.size(i);
  return index.expand(sizes).contiguous();
}

class SegmentSumCOO : public torch::autograd::Function<SegmentSumCOO> {
 public:
  static variable_list forward(torch::autograd::AutogradContext* ctx, const Tensor& src, const Tensor& index,
                               const std::optional<Tensor>& optional_out, std::optional<int64_t> dim_size) {
    at::AutoDispatchBelowADInplaceOrView g;
    static auto op = find_op<CooFn>("pyg::segment_sum_coo");
    auto out = op.call(src, index, optional_out, dim_size);
    ctx->save_for_backward({index});
    if (optional_out.has_value()) ctx->mark_dirty({optional_out.value()});
    return {out};
  }
  static variable_list backward(torch::autograd::AutogradContext* ctx, variable_list grad_outs) {
    static auto gather = find_op<GatherFn>("pyg::gather_coo");
    const auto saved = ctx->get_saved_variables();
    return {gather.call(grad_outs[0], saved[0], std::nullopt), Tensor(), Tensor(), Tensor()};
  }
};

class SegmentMeanCOO : public torch::autograd::Function<SegmentMeanCOO> {
 public:
  static variable_list forward(torch::autograd::AutogradContext* ctx, const Tensor& src, const Tensor& index,
                               const std::optional<Tensor>& optional_out, std::optional<int64_t> dim_size) {
    at::AutoDispatchBelowADInplaceOrView g;
    static auto mean = find_op<CooFn>("pyg::segment_mean_coo");
    static auto sum = find_op<CooFn>("pyg::segment_sum_coo");
    const int64_t dim = index.dim() - 1;
    TORCH_CHECK(dim >= 0, "segment_mean_coo: index must have at least 1 dimension");
    auto index_b = expand_coo_index(src, index);
    auto out = mean.call(src, index, optional_out, dim_size);
    auto ones = at::ones(index_b.sizes(), out.options());
    auto count = sum.call(ones, index_b, std::nullopt, out.size(dim));
    ctx->save_for_backward({index_b, count});
    ctx->saved_data["src_shape"] = src.sizes();
    if (optional_out.has_value()) ctx->mark_dirty({optional_out.value()});
    return {out};
  }
  static variable_list backward(torch::autograd::AutogradContext* ctx, variable_list grad_outs) {
    static auto gather = find_op<GatherFn>("pyg::gather_coo");
    const auto grad_out = grad_outs[0];
    const auto saved = ctx->get_saved_variables();
    const auto index_b = saved[0];
    auto count = saved[1];
    auto shape = ctx->saved_data["src_shape"].toIntList().vec();
    auto grad_in = at::empty(shape, grad_out.options());
    if (grad_in.numel() > 0) {
      gather.call(grad_out, index_b, grad_in);
      count = gather.call(count, index_b, std::nullopt);
      for (int64_t i = 0; i < grad_out.dim() - index_b.dim(); ++i) count = count.unsqueeze(-1);
      grad_in.true_divide_(count);
    }
    return {grad_in, Tensor(), Tensor(), Tensor()};
  }
};

template <bool IS_MIN>
class SegmentMinMaxCOO : public torch::autograd::Function<SegmentMinMaxCOO<IS_MIN>> {
 public:
  static variable_list forward(torch::autograd::AutogradContext* ctx, const Tensor& src, const Tensor& index,
                               const std::optional<Tensor>& optional_out, std::optional<int64_t> dim_size) {
    at::AutoDispatchBelowADInplaceOrView g;
    static auto opmin = find_op<CooArgFn>("pyg::segment_min_coo");
    static auto opmax = find_op<CooArgFn>("pyg::segment_max_coo");
    const int64_t dim = index.dim() - 1;
    auto res = (IS_MIN ? opmin : opmax).call(src, index, optional_out, dim_size);
    auto out = std::get<0>(res);
    auto arg = std::get<1>(res);
    ctx->save_for_backward({arg});
    ctx->saved_data["dim"] = dim;
    ctx->saved_data["src_shape"] = src.sizes();
    ctx->mark_non_differentiable({arg});
    if (optional_out.has_value()) ctx->mark_dirty({optional_out.value()});
    return {out, arg};
  }
  static variable_list backward(torch::autograd::AutogradContext* ctx, variable_list grad_outs) {
    const auto grad_out = grad_outs[0];
    const auto saved = ctx->get_saved_variables();
    const auto dim = ctx->saved_data["dim"].toInt();
    auto shape = ctx->saved_data["src_shape"].toIntList().vec();
    shape[dim] += 1;
    auto grad_in = at::zeros(shape, grad_out.options());
    grad_in.scatter_(dim, saved[0], grad_out);
    return {grad_in.narrow(dim, 0, shape[dim] - 1), Tensor(), Tensor(), Tensor()};
  }
};

class GatherCOO : public torch::autograd::Function<GatherCOO> {
 public:
  static variable_list forward(torch::autograd::AutogradContext* ctx, const Tensor& src, const Tensor& index,
                               const std::optional<Tensor>& optional_out) {
    at::AutoDispatchBelowADInplaceOrView g;
    static auto gather = find_op<GatherFn>("pyg::gather_coo");
    auto out = gather.call(src, index, optional_out);
    ctx->save_for_backward({index});
    ctx->saved_data["src_shape"] = src.sizes();
    if (optional_out.has_value()) ctx->mark_dirty({optional_out.value()});
    return {out};
  }
  static variable_list backward(torch::autograd::AutogradContext* ctx, variable_list grad_outs) {
    // deposit the gradient with segment_sum_coo(out=zeros(src_shape))
    static auto sum = find_op<CooFn>("pyg::segment_sum_coo");
    const auto saved = ctx->get_saved_variables();
    auto shape = ctx->saved_data["src_shape"].toIntList().vec();
    auto grad_in = at::zeros(shape, grad_outs[0].options());
    sum.call(grad_outs[0].contiguous(), saved[0], grad_in, std::nullopt);
    return {grad_in, Tensor(), Tensor()};
  }
};

Tensor segment_sum_coo_autograd(const Tensor& src, const Tensor& index, const std::optional<Tensor>& out,
                                std::optional<int64_t> dim_size) {
  return SegmentSumCOO::apply(src, index, out, dim_size)[0];
}
Tensor segment_mean_coo_autograd(const Tensor& src, const Tensor& index, const std::optional<Tensor>& out,
                                 std::optional<int64_t> dim_size) {
  return SegmentMeanCOO::apply(src, index, out, dim_size)[0];
}
std::tuple<Tensor, Tensor> segment_min_coo_autograd(const Tensor& src, const Tensor& index,
                                                    const std::optional<Tensor>& out,
                                                    std::optional<int64_t> dim_size) {
  auto r = SegmentMinMaxCOO<true>::apply(src, index, out, dim_size);
  return std::make_tuple(r[0], r[1]);
}
std::tuple<Tensor, Tensor> segment_max_coo_autograd(const Tensor& src, const Tensor& index,
                                                    const std::optional<Tensor>& out,
                                                    std::optional<int64_t> dim_size) {
  auto r = SegmentMinMaxCOO<false>::apply(src, index, out, dim_size);
  return std::make_tuple(r[0], r[1]);
}
Tensor gather_coo_autograd(const Tensor& src, const Tensor& index, const std::optional<Tensor>& out) {
  return GatherCOO::apply(src, index, out)[0];
}

// ---------------------------------------------------------------------------------------------
// registration
// ---------------------------------------------------------------------------------------------
TORCH_LIBRARY_FRAGMENT(pyg, m) {
  // pyg_lib/csrc/ops/index_sort.cpp:25-28
  m.def(TORCH_SELECTIVE_SCHEMA("pyg::index_sort(Tensor indices, int? max = None) -> (Tensor, Tensor)"));
  // pyg_lib/csrc/ops/scatter.cpp:156-172
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::scatter_sum(Tensor src, Tensor index, int dim=-1, "
      "Tensor? out=None, int? dim_size=None) -> Tensor"));
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::scatter_mul(Tensor src, Tensor index, int dim=-1, "
      "Tensor? out=None, int? dim_size=None) -> Tensor"));
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::scatter_mean(Tensor src, Tensor index, int dim=-1, "
      "Tensor? out=None, int? dim_size=None) -> Tensor"));
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::scatter_min(Tensor src, Tensor index, int dim=-1, "
      "Tensor? out=None, int? dim_size=None) -> (Tensor, Tensor)"));
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::scatter_max(Tensor src, Tensor index, int dim=-1, "
      "Tensor? out=None, int? dim_size=None) -> (Tensor, Tensor)"));
  // pyg_lib/csrc/ops/segment_coo.cpp:150-165
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::segment_sum_coo(Tensor src, Tensor index, "
      "Tensor? out=None, int? dim_size=None) -> Tensor"));
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::segment_mean_coo(Tensor src, Tensor index, "
      "Tensor? out=None, int? dim_size=None) -> Tensor"));
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::segment_min_coo(Tensor src, Tensor index, "
      "Tensor? out=None, int? dim_size=None) -> (Tensor, Tensor)"));
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::segment_max_coo(Tensor src, Tensor index, "
      "Tensor? out=None, int? dim_size=None) -> (Tensor, Tensor)"));
  m.def(TORCH_SELECTIVE_SCHEMA("pyg::gather_coo(Tensor src, Tensor index, Tensor? out=None) -> Tensor"));
}

TORCH_LIBRARY_IMPL(pyg, CUDA, m) {
  m.impl(TORCH_SELECTIVE_NAME("pyg::index_sort"), TORCH_FN(index_sort_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::scatter_sum"), TORCH_FN(scatter_sum_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::scatter_mul"), TORCH_FN(scatter_mul_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::scatter_min"), TORCH_FN(scatter_min_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::scatter_max"), TORCH_FN(scatter_max_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_sum_coo"), TORCH_FN(segment_sum_coo_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_mean_coo"), TORCH_FN(segment_mean_coo_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_min_coo"), TORCH_FN(segment_min_coo_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_max_coo"), TORCH_FN(segment_max_coo_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::gather_coo"), TORCH_FN(gather_coo_kernel));
}

// key CPU (pyg_lib/csrc/ops/cpu/scatter_kernel.cpp:513-518, segment_coo_kernel.cpp:748-757): the same fronts; they branch
// to the kernels of cpu_reduce.h on CPU tensors
TORCH_LIBRARY_IMPL(pyg, CPU, m) {
  m.impl(TORCH_SELECTIVE_NAME("pyg::scatter_sum"), TORCH_FN(scatter_sum_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::scatter_mul"), TORCH_FN(scatter_mul_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::scatter_min"), TORCH_FN(scatter_min_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::scatter_max"), TORCH_FN(scatter_max_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_sum_coo"), TORCH_FN(segment_sum_coo_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_mean_coo"), TORCH_FN(segment_mean_coo_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_min_coo"), TORCH_FN(segment_min_coo_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_max_coo"), TORCH_FN(segment_max_coo_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::gather_coo"), TORCH_FN(gather_coo_kernel));
}

TORCH_LIBRARY_IMPL(pyg, Autograd, m) {
  m.impl(TORCH_SELECTIVE_NAME("pyg::scatter_sum"), TORCH_FN(scatter_sum_autograd));
  m.impl(TORCH_SELECTIVE_NAME("pyg::scatter_mul"), TORCH_FN(scatter_mul_autograd));
  m.impl(TORCH_SELECTIVE_NAME("pyg::scatter_mean"), TORCH_FN(scatter_mean_autograd));
  m.impl(TORCH_SELECTIVE_NAME("pyg::scatter_min"), TORCH_FN(scatter_min_autograd));
  m.impl(TORCH_SELECTIVE_NAME("pyg::scatter_max"), TORCH_FN(scatter_max_autograd));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_sum_coo"), TORCH_FN(segment_sum_coo_autograd));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_mean_coo"), TORCH_FN(segment_mean_coo_autograd));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_min_coo"), TORCH_FN(segment_min_coo_autograd));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_max_coo"), TORCH_FN(segment_max_coo_autograd));
  m.impl(TORCH_SELECTIVE_NAME("pyg::gather_coo"), TORCH_FN(gather_coo_autograd));
}

// ops/autograd/scatter_kernel.cpp:454-457
TORCH_LIBRARY_IMPL(pyg, CompositeExplicitAutograd, m) {
  m.impl(TORCH_SELECTIVE_NAME("pyg::scatter_mean"), TORCH_FN(scatter_mean_autograd));
}

}  // namespace pyg_amd
