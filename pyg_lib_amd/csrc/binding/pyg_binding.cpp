// libpyg.so -- the drop-in operator library.
//
// Registers the reference's operator schemas in namespace `pyg` (byte-identical strings, cited
// per op) and implements them for PyTorch-ROCm tensors by calling the torch-free C-ABI of
// include/pyg_hip.h and nothing else.  PyTorch is plumbing here: argument checks in the
// reference's wording, output allocation through the caching allocator, the current HIP stream,
// the global CPU generator for the sampler's random words, and autograd glue.
//
// CPU tensors: the samplers, segment/grouped_matmul and index_sort have their own CPU kernels
// (pyg_binding_cpu.cpp, dispatch key `CPU`); every other op of this library is device-only and a CPU tensor
// reaches the dispatcher's "could not run ... with arguments from the 'CPU' backend" error.  A device tensor
// never takes a CPU path: there is no fallback of any kind.
#include <chrono>
#include <ATen/ATen.h>
#include <ATen/CPUGeneratorImpl.h>
#include <ATen/core/dispatch/Dispatcher.h>
#include <ATen/hip/impl/HIPCachingAllocatorMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/autograd.h>
#include <ATen/Context.h>
#include <torch/library.h>

#include <cstdlib>

#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "binding_common.h"

namespace pyg_amd {

typedef std::string node_type;
typedef std::string rel_type;
typedef std::tuple<std::string, std::string, std::string> edge_type;

// pyg_lib/csrc/utils/types.h:10-12
inline rel_type to_rel_type(const edge_type& key) {
  return std::get<0>(key) + "__" + std::get<1>(key) + "__" + std::get<2>(key);
}

// ---------------------------------------------------------------------------------------------
// library.cpp:19-29
// ---------------------------------------------------------------------------------------------
int64_t cuda_version() { return pyg_hip_version(); }

int& matmul_schedule_tls() {
  static thread_local int sched = PYG_HIP_MM_SCHED_AUTO;
  return sched;
}

int matmul_flags(at::ScalarType t) {
  int flags = matmul_schedule_tls() & PYG_HIP_MM_SCHED_MASK;
  if (t == at::kFloat && at::globalContext().float32MatmulPrecision() != at::Float32MatmulPrecision::HIGHEST)
    flags |= PYG_HIP_MM_F32_SPLIT;
  return flags;
}

struct ScheduleGuard {
  int prev;
  explicit ScheduleGuard(int s) : prev(matmul_schedule_tls()) { matmul_schedule_tls() = s; }
  ~ScheduleGuard() { matmul_schedule_tls() = prev; }
};

// ---------------------------------------------------------------------------------------------
// matmul  (front: pyg_lib/csrc/ops/matmul.cpp:12-61, kernels: ops/cuda/matmul_kernel.cu:289-319)
// ---------------------------------------------------------------------------------------------
static Tensor segment_matmul_impl(const Tensor& input, const Tensor& ptr, const Tensor& other,
                                  const c10::optional<Tensor>& bias) {
  PYG_TRACE("pyg::segment_matmul");
  at::TensorArg input_arg{input, "input", 0};
  at::TensorArg ptr_arg{ptr, "ptr", 1};
  at::TensorArg other_arg{other, "other", 2};
  at::CheckedFrom c{"segment_matmul"};
  at::checkAllDefined(c, {input_arg, ptr_arg, other_arg});
  at::checkSameType(c, input_arg, other_arg);
  at::checkDim(c, input_arg, 2);
  at::checkDim(c, ptr_arg, 1);
  at::checkDim(c, other_arg, 3);
  at::checkSize(c, other_arg, 1, input_arg->size(-1));
  at::checkNumel(c, ptr_arg, other_arg->size(0) + 1);
  TORCH_CHECK(ptr.scalar_type() == at::kLong, "expected scalar type Long but found ",
              ptr.scalar_type());
  TORCH_CHECK(input.is_cuda() && other.is_cuda() && input.device() == other.device(),
              "segment_matmul: 'input' and 'other' must live on the same HIP device");

  DeviceGuard guard(input.device());
  const auto x = input.contiguous();
  const auto w = other.contiguous();
  const auto p = ptr.contiguous();
  const int64_t N = x.size(0), K = x.size(1), B = w.size(0), M = w.size(2);
  auto out = x.new_empty({N, M});
  Tensor b;
  if (bias.has_value()) {
    b = bias.value().to(x.options()).contiguous();
    TORCH_CHECK(b.dim() == 2 && b.size(0) == B && b.size(1) == M, "segment_matmul: expected 'bias' of shape [",
                B, ", ", M, "]");
  }
  auto ws = at::empty({(int64_t)pyg_hip_matmul_workspace_size(B)}, x.options().dtype(at::kByte));
  check_status(pyg_hip_segment_matmul(dtype_code(x.scalar_type()), x.data_ptr(), p.data_ptr<int64_t>(),
                                      p.is_cuda() ? 1 : 0, w.data_ptr(),
                                      b.defined() ? b.data_ptr() : nullptr, out.data_ptr(), N, K, M, B,
                                      ws.data_ptr(), (size_t)ws.numel(), matmul_flags(x.scalar_type()),
                                      current_stream(x)));
  return out;
}

Tensor segment_matmul_kernel(const Tensor& input, const Tensor& ptr, const Tensor& other) {
  return segment_matmul_impl(input, ptr, other, c10::nullopt);
}

// Extra op of this build: bias fused as GEMM epilogue (the reference adds it in a Python loop of
// B slice updates, pyg_lib/ops/__init__.py:169-171).
Tensor segment_matmul_bias_kernel(const Tensor& input, const Tensor& ptr, const Tensor& other,
                                  const Tensor& bias) {
  return segment_matmul_impl(input, ptr, other, bias);
}

// `caller_pool` (this build's pyg::grouped_matmul_pool): a contiguous [sum of rows, M] tensor that receives the
// outputs as consecutive row ranges -- the sharded driver passes its slot of the all-gather buffer
// (pyg_lib_amd/sharding.py), so the results are produced where the collective reads them.
static std::vector<Tensor> grouped_matmul_impl(const at::TensorList input, const at::TensorList other,
                                               const c10::optional<Tensor>& caller_pool) {
  PYG_TRACE("pyg::grouped_matmul");
  TORCH_CHECK(input.size() == other.size(),
              "Number of 'input' tensors must match number of 'other' tensors");
  const size_t G = input.size();
  std::vector<Tensor> outs;
  if (G == 0) return outs;
  // The reference's argument checks (TensorArg messages naming the offending list entry).  With 512 groups, building
  // 1024 names and TensorArgs up front cost more host time than anything else in the call: look first, and go through
  // the naming checks only when something is wrong (they then raise the reference's message).
  bool plain = input[0].defined();
  for (size_t i = 0; plain && i < G; ++i) {
    const Tensor& a = input[i];
    const Tensor& o = other[i];
    plain = a.defined() && o.defined() && a.scalar_type() == input[0].scalar_type() &&
            o.scalar_type() == input[0].scalar_type() && a.dim() == 2 && o.dim() == 2 && o.size(0) == a.size(1) &&
            a.is_cuda() && o.is_cuda();
  }
  if (!plain) {
    at::CheckedFrom c{"grouped_matmul"};
    for (size_t i = 0; i < G; ++i) {
      const std::string na = "input[" + std::to_string(i) + "]", no = "other[" + std::to_string(i) + "]";
      at::TensorArg a{input[i], na.c_str(), 0};
      at::TensorArg o{other[i], no.c_str(), 1};
      at::checkDefined(c, a);
      at::checkDefined(c, o);
      at::checkScalarType(c, a, input[0].scalar_type());
      at::checkScalarType(c, o, input[0].scalar_type());
      at::checkDim(c, a, 2);
      at::checkDim(c, o, 2);
      at::checkSize(c, o, 0, a->size(-1));
      TORCH_CHECK(input[i].is_cuda() && other[i].is_cuda(), "grouped_matmul: tensors must live on a HIP device");
    }
  }
  DeviceGuard guard(input[0].device());
  std::vector<pyg_hip_group> groups(G);
  std::vector<Tensor> keep;
  keep.reserve(2 * G);
  outs.reserve(G);
  // Weight-gradient pattern of GroupedMatmul.backward (pyg_lib/ops/__init__.py:88-94): every input is
  // the transposed view X_i^T of a row-major X_i [rows_i, K] and the contraction runs over rows_i.
  // One persistent launch of the dW kernel (csrc/hip/matmul_dw.hip) instead of a transposing copy of
  // every X_i and G skinny GEMMs.
  {
    const auto st = input[0].scalar_type();
    bool dw = (st == at::kBFloat16 || st == at::kHalf || st == at::kFloat) && !caller_pool.has_value();
    const int64_t esz = (int64_t)input[0].element_size();
    // every input is a transposed view (tensors that are both -- one row, one column, no elements -- count), at
    // least one of them a real one: a plain forward call never comes here
    bool any_view = false;
    for (size_t i = 0; dw && i < G; ++i) {
      dw = input[i].t().is_contiguous() && other[i].is_contiguous() && input[i].size(0) > 0 && other[i].size(1) > 0 &&
           (uintptr_t)input[i].data_ptr() % esz == 0 && (uintptr_t)other[i].data_ptr() % esz == 0;
      any_view = any_view || !input[i].is_contiguous();
    }
    if (dw && any_view) {
      // per-group shapes: K_i = input[i].size(0), M_i = other[i].size(1); the results lie back to back in one pool
      std::vector<int64_t> poff(G + 1, 0);
      for (size_t i = 0; i < G; ++i) {
        groups[i].input = input[i].data_ptr();  // X_i, row-major [rows_i, K_i]
        groups[i].other = other[i].data_ptr();  // dY_i [rows_i, M_i]
        groups[i].out = nullptr;
        groups[i].rows = input[i].size(1);
        groups[i].k = (int32_t)input[i].size(0);
        groups[i].m = (int32_t)other[i].size(1);
        groups[i].other_trans = 0;
        groups[i].reserved = 0;
        poff[i + 1] = poff[i] + input[i].size(0) * other[i].size(1);
      }
      auto pool = at::empty({poff[G]}, input[0].options());
      auto ws = at::empty({(int64_t)pyg_hip_grouped_matmul_dw_workspace_size(groups.data(), (int64_t)G)},
                          input[0].options().dtype(at::kByte));
      const int rc = pyg_hip_grouped_matmul_dw(dtype_code(st), groups.data(), (int64_t)G, pool.data_ptr(), ws.data_ptr(),
                                               (size_t)ws.numel(), current_stream(input[0]));
      if (rc == PYG_HIP_OK) {
        // plain aliases of the pool, not tracked views: the reference returns G independent tensors, and
        // outputs that are "views of a multi-output function" could not be modified in place by the caller
        at::AutoDispatchBelowADInplaceOrView untracked;
        for (size_t i = 0; i < G; ++i)
          outs.push_back(pool.as_strided({input[i].size(0), other[i].size(1)}, {other[i].size(1), 1}, poff[i]));
        return outs;
      }
      TORCH_CHECK(rc == PYG_HIP_ERR_UNSUPPORTED, pyg_hip_last_error());
    }
  }
  // One allocation for all outputs (the reference makes G of them, matmul_kernel.cpp:296-298); each
  // output is a 16-byte aligned view into it.
  const int64_t elt = (int64_t)input[0].element_size();
  const int64_t align = 16 / elt > 0 ? 16 / elt : 1;
  std::vector<int64_t> offs(G);
  int64_t total = 0;
  for (size_t i = 0; i < G; ++i) {
    offs[i] = total;
    const int64_t n = input[i].size(0) * other[i].size(-1);
    total += (n + align - 1) / align * align;
  }
  Tensor pool;
  if (caller_pool.has_value()) {
    pool = caller_pool.value();
    const int64_t M0 = other[0].size(-1);
    int64_t rows = 0;
    for (size_t i = 0; i < G; ++i) {
      TORCH_CHECK(other[i].size(-1) == M0, "grouped_matmul_pool: every 'other' must have the same number of columns");
      rows += input[i].size(0);
    }
    TORCH_CHECK(pool.dim() == 2 && pool.size(0) == rows && pool.size(1) == M0 && pool.is_contiguous() &&
                    pool.scalar_type() == input[0].scalar_type() && pool.device() == input[0].device(),
                "grouped_matmul_pool: expected a contiguous 'pool' of shape [", rows, ", ", M0, "] like 'input'");
    TORCH_CHECK((M0 * elt) % 16 == 0 || G == 1, "grouped_matmul_pool: rows of 'pool' must be 16-byte multiples");
    total = 0;
    for (size_t i = 0; i < G; ++i) {
      offs[i] = total;
      total += input[i].size(0) * M0;
    }
    pool = pool.view({-1});
  } else {
    pool = at::empty({std::max<int64_t>(total, 1)}, input[0].options());
  }
  char* const pool_base = static_cast<char*>(pool.data_ptr());
  for (size_t i = 0; i < G; ++i) {
    Tensor a = input[i];
    if (!a.is_contiguous()) {
      a = a.contiguous();
      keep.push_back(a);
    }
    Tensor o = other[i];
    int trans = 0;
    if (!o.is_contiguous()) {
      // a transposed view (backward pass, pyg_lib/ops/__init__.py:84,91) is read in place
      if (o.t().is_contiguous()) {
        trans = 1;
      } else {
        o = o.contiguous();
        keep.push_back(o);
      }
    }
    groups[i].input = a.data_ptr();
    groups[i].other = o.data_ptr();
    groups[i].out = pool_base + offs[i] * elt;
    groups[i].rows = a.size(0);
    groups[i].k = (int32_t)a.size(1);
    groups[i].m = (int32_t)other[i].size(-1);
    groups[i].other_trans = trans;
    groups[i].reserved = 0;
  }
  auto ws = at::empty({(int64_t)pyg_hip_matmul_workspace_size((int64_t)G)},
                      input[0].options().dtype(at::kByte));
  check_status(pyg_hip_grouped_matmul(dtype_code(input[0].scalar_type()), groups.data(), (int64_t)G,
                                      ws.data_ptr(), (size_t)ws.numel(), matmul_flags(input[0].scalar_type()),
                                      current_stream(input[0])));
  // the outputs: one dispatcher call each, made while the kernel runs (aliases of the pool that are NOT tracked as
  // views, see above)
  at::AutoDispatchBelowADInplaceOrView untracked;
  for (size_t i = 0; i < G; ++i)
    outs.push_back(pool.as_strided({input[i].size(0), other[i].size(-1)}, {other[i].size(-1), 1}, offs[i]));
  return outs;
}

std::vector<Tensor> grouped_matmul_kernel(const at::TensorList input, const at::TensorList other) {
  return grouped_matmul_impl(input, other, c10::nullopt);
}

std::vector<Tensor> grouped_matmul_pool_kernel(const at::TensorList input, const at::TensorList other, Tensor pool) {
  return grouped_matmul_impl(input, other, pool);
}

// This build only: gather -> per-relation matmul -> scatter-add in one launch (csrc/hip/rgcn.hip).  `out` is
// accumulated into and returned.  Indices are validated on the device (rgcn_check_flags).
static int rgcn_check_flags() {
  // default: validated on the device without a synchronisation (PYG_HIP_RGCN_DEFERRED: a stale node id is redirected to
  // row 0 instead of reading out of bounds, and reported by the next call); 1: validated, synchronising, fails in the
  // call that has the bad index; 0: no validation (the round-4 behaviour)
  static const int flags = [] {
    const char* e = getenv("PYG_HIP_RGCN_CHECK");
    if (e == nullptr || e[0] == '\0') return PYG_HIP_RGCN_DEFERRED;
    return e[0] == '0' ? 0 : PYG_HIP_RGCN_CHECKED;
  }();
  return flags;
}

static void rgcn_index_checks(const at::TensorList gather_index, const at::TensorList scatter_index, const Tensor& like,
                              size_t r) {
  TORCH_CHECK(gather_index[r].scalar_type() == at::kLong && scatter_index[r].scalar_type() == at::kLong &&
                  gather_index[r].dim() == 1 && gather_index[r].sizes() == scatter_index[r].sizes(),
              "rgcn_fused: index vectors must be 1-D int64 tensors of equal length");
  TORCH_CHECK(gather_index[r].device() == like.device() && scatter_index[r].device() == like.device(),
              "rgcn_fused: index vectors must live on the device of the features (got ", gather_index[r].device(), " / ",
              scatter_index[r].device(), " vs ", like.device(), ")");
}

// `grouped`: every scatter_index is nondecreasing (the samplers' `row`): the atomic-free owner-computes kernel
// (PYG_HIP_RGCN_GROUPED) WRITES `out` -- deterministic, no zero fill needed; verified on the device (rgcn_check_flags)
static at::Tensor rgcn_workspace(const std::vector<pyg_hip_rgcn_relation>& rels, int64_t E, int64_t out_rows, bool grouped,
                                 const at::TensorOptions& like) {
  const size_t bytes = grouped ? pyg_hip_rgcn_grouped_workspace_size(rels.data(), (int64_t)rels.size(), out_rows)
                               : pyg_hip_rgcn_fused_workspace_size((int64_t)rels.size(), E);
  return at::empty({(int64_t)bytes}, like.dtype(at::kByte));
}

Tensor rgcn_fused_kernel(const Tensor& x, const at::TensorList gather_index, const at::TensorList scatter_index,
                         at::IntArrayRef gather_offset, at::IntArrayRef scatter_offset, const Tensor& weight, Tensor out,
                         bool grouped, at::OptionalIntArrayRef scatter_rows) {
  PYG_TRACE("pyg::rgcn_fused");
  // packed 16-bit atomic adds: the result depends on the order they land in (pyg_lib_amd.rgcn takes the atomic-free
  // three-op chain under torch.use_deterministic_algorithms(True) instead of calling this operator); grouped: no atomics
  if (!grouped) at::globalContext().alertNotDeterministic("pyg::rgcn_fused");
  const size_t R = gather_index.size();
  TORCH_CHECK(scatter_index.size() == R && gather_offset.size() == R && scatter_offset.size() == R &&
                  (!scatter_rows.has_value() || scatter_rows->size() == R),
              "rgcn_fused: one gather / scatter index vector and offset (and scatter_rows entry) per relation expected");
  TORCH_CHECK(x.is_cuda() && weight.is_cuda() && out.is_cuda() && x.device() == weight.device() && x.device() == out.device(),
              "rgcn_fused: tensors must live on the same HIP device");
  TORCH_CHECK(x.dim() == 2 && out.dim() == 2 && weight.dim() == 3 && (size_t)weight.size(0) == R &&
                  weight.size(1) == x.size(1) && weight.size(2) == out.size(1),
              "rgcn_fused: expected x [N, K], weight [R, K, M], out [N_out, M]");
  TORCH_CHECK(x.scalar_type() == weight.scalar_type() && x.scalar_type() == out.scalar_type(), "rgcn_fused: dtype mismatch");
  TORCH_CHECK(out.is_contiguous(), "rgcn_fused: 'out' must be contiguous");
  DeviceGuard guard(x.device());
  const auto xc = x.contiguous();
  const auto wc = weight.contiguous();
  std::vector<pyg_hip_rgcn_relation> rels(R);
  std::vector<Tensor> keep;
  int64_t E = 0;
  for (size_t r = 0; r < R; ++r) {
    rgcn_index_checks(gather_index, scatter_index, x, r);
    auto g = gather_index[r].contiguous();
    auto s = scatter_index[r].contiguous();
    rels[r] = pyg_hip_rgcn_relation{};
    rels[r].gather_index = g.data_ptr<int64_t>();
    rels[r].scatter_index = s.data_ptr<int64_t>();
    rels[r].num_edges = g.numel();
    rels[r].gather_offset = gather_offset[r];
    rels[r].scatter_offset = scatter_offset[r];
    rels[r].scatter_rows = scatter_rows.has_value() ? (*scatter_rows)[r] : 0;
    rels[r].weight = static_cast<const char*>(wc.data_ptr()) + (int64_t)r * wc.size(1) * wc.size(2) * wc.element_size();
    E += g.numel();
    keep.push_back(g);
    keep.push_back(s);
  }
  auto ws = rgcn_workspace(rels, E, out.size(0), grouped, x.options());
  check_status(pyg_hip_rgcn_fused(dtype_code(x.scalar_type()), xc.data_ptr(), xc.size(0), rels.data(), (int64_t)R,
                                  out.data_ptr(), out.size(0), xc.size(1), out.size(1),
                                  rgcn_check_flags() | (grouped ? PYG_HIP_RGCN_GROUPED : 0), ws.data_ptr(), (size_t)ws.numel(),
                                  current_stream(x)));
  return out;
}

// The same without the per-batch feature matrix: relation r gathers row node_id[gather_type[r]][gather_index[r][e]] of
// the GLOBAL feature table feat[gather_type[r]] -- what `x = cat([feat[t][node_id[t]] ...])` followed by rgcn_fused
// computes, minus the ATen gathers, the cat and the [sum n_t, K] intermediate.
Tensor rgcn_fused_tables_kernel(const at::TensorList feat, const at::TensorList node_id, at::IntArrayRef gather_type,
                                const at::TensorList gather_index, const at::TensorList scatter_index,
                                at::IntArrayRef scatter_offset, const Tensor& weight, Tensor out, bool grouped,
                                at::OptionalIntArrayRef scatter_rows) {
  PYG_TRACE("pyg::rgcn_fused_tables");
  if (!grouped) at::globalContext().alertNotDeterministic("pyg::rgcn_fused_tables");  // (see rgcn_fused_kernel)
  const size_t R = gather_index.size(), T = feat.size();
  TORCH_CHECK(T > 0 && node_id.size() == T, "rgcn_fused_tables: one node-id vector per feature table expected");
  TORCH_CHECK(scatter_index.size() == R && gather_type.size() == R && scatter_offset.size() == R &&
                  (!scatter_rows.has_value() || scatter_rows->size() == R),
              "rgcn_fused_tables: one gather type, gather / scatter index vector and offset (and scatter_rows entry) per relation expected");
  const Tensor& f0 = feat[0];
  TORCH_CHECK(f0.is_cuda() && weight.is_cuda() && out.is_cuda() && f0.device() == weight.device() && f0.device() == out.device(),
              "rgcn_fused_tables: tensors must live on the same HIP device");
  TORCH_CHECK(out.dim() == 2 && weight.dim() == 3 && (size_t)weight.size(0) == R && weight.size(2) == out.size(1),
              "rgcn_fused_tables: expected weight [R, K, M], out [N_out, M]");
  TORCH_CHECK(out.is_contiguous() && out.scalar_type() == weight.scalar_type(), "rgcn_fused_tables: 'out' must be contiguous and typed like 'weight'");
  DeviceGuard guard(f0.device());
  std::vector<Tensor> keep;
  std::vector<Tensor> fc(T), nc(T);
  for (size_t t = 0; t < T; ++t) {
    TORCH_CHECK(feat[t].dim() == 2 && feat[t].size(1) == weight.size(1) && feat[t].scalar_type() == weight.scalar_type() &&
                    feat[t].device() == f0.device(),
                "rgcn_fused_tables: feat[", t, "] must be [N_t, K] on the common device and typed like 'weight'");
    TORCH_CHECK(node_id[t].dim() == 1 && node_id[t].scalar_type() == at::kLong && node_id[t].device() == f0.device(),
                "rgcn_fused_tables: node_id[", t, "] must be a 1-D int64 tensor on the common device");
    fc[t] = feat[t].contiguous();
    nc[t] = node_id[t].contiguous();
  }
  const auto wc = weight.contiguous();
  std::vector<pyg_hip_rgcn_relation> rels(R);
  int64_t E = 0;
  for (size_t r = 0; r < R; ++r) {
    rgcn_index_checks(gather_index, scatter_index, f0, r);
    TORCH_CHECK(gather_type[r] >= 0 && (size_t)gather_type[r] < T, "rgcn_fused_tables: gather_type out of range");
    auto g = gather_index[r].contiguous();
    auto s = scatter_index[r].contiguous();
    const size_t t = (size_t)gather_type[r];
    rels[r] = pyg_hip_rgcn_relation{};
    rels[r].gather_index = g.data_ptr<int64_t>();
    rels[r].scatter_index = s.data_ptr<int64_t>();
    rels[r].num_edges = g.numel();
    rels[r].gather_offset = 0;
    rels[r].scatter_offset = scatter_offset[r];
    rels[r].scatter_rows = scatter_rows.has_value() ? (*scatter_rows)[r] : 0;
    rels[r].weight = static_cast<const char*>(wc.data_ptr()) + (int64_t)r * wc.size(1) * wc.size(2) * wc.element_size();
    rels[r].x = fc[t].data_ptr();
    rels[r].gather_map = nc[t].data_ptr<int64_t>();
    rels[r].x_rows = fc[t].size(0);
    rels[r].gather_map_len = nc[t].numel();
    E += g.numel();
    keep.push_back(g);
    keep.push_back(s);
  }
  auto ws = rgcn_workspace(rels, E, out.size(0), grouped, f0.options());
  check_status(pyg_hip_rgcn_fused(dtype_code(weight.scalar_type()), nullptr, 0, rels.data(), (int64_t)R, out.data_ptr(),
                                  out.size(0), wc.size(1), out.size(1), rgcn_check_flags() | (grouped ? PYG_HIP_RGCN_GROUPED : 0),
                                  ws.data_ptr(), (size_t)ws.numel(), current_stream(f0)));
  return out;
}

// Autograd, mirroring SegmentMatmul (pyg_lib/csrc/ops/autograd/matmul_kernel.cpp:68-111).
static Tensor segment_matmul_below_autograd(const Tensor& input, const Tensor& ptr, const Tensor& other) {
  static auto op = c10::Dispatcher::singleton()
                       .findSchemaOrThrow("pyg::segment_matmul", "")
                       .typed<Tensor(const Tensor&, const Tensor&, const Tensor&)>();
  return op.call(input, ptr, other);
}

// dW through the C-ABI; returns an undefined tensor when the device kernel does not cover the case.
static Tensor segment_matmul_dw(const Tensor& input, const Tensor& ptr, const Tensor& grad_out, int64_t B) {
  PYG_TRACE("pyg::segment_matmul_backward_dw");
  const auto st = input.scalar_type();
  if (!input.is_cuda()) return Tensor();  // CPU tensors: the reference formula below
  if (st != at::kBFloat16 && st != at::kHalf && st != at::kFloat) return Tensor();
  const int64_t K = input.size(1), M = grad_out.size(1);
  if (B == 0 || K == 0 || M == 0) return at::empty({B, K, M}, input.options());
  DeviceGuard guard(input.device());
  auto x = input.contiguous();
  auto gy = grad_out.contiguous();
  auto p = ptr.contiguous();
  auto out = at::empty({B, K, M}, input.options());
  auto ws = at::empty({(int64_t)pyg_hip_segment_matmul_dw_workspace_size(B, K, M)}, x.options().dtype(at::kByte));
  const int rc = pyg_hip_segment_matmul_dw(dtype_code(st), x.data_ptr(), p.data_ptr<int64_t>(), p.is_cuda() ? 1 : 0,
                                           gy.data_ptr(), out.data_ptr(), x.size(0), K, M, B, ws.data_ptr(),
                                           (size_t)ws.numel(), current_stream(x));
  if (rc == PYG_HIP_ERR_UNSUPPORTED) return Tensor();
  check_status(rc);
  return out;
}

class SegmentMatmul : public torch::autograd::Function<SegmentMatmul> {
 public:
  static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx,
                                                const Tensor& input, const Tensor& ptr,
                                                const Tensor& other) {
    at::AutoDispatchBelowADInplaceOrView g;
    Tensor out = segment_matmul_below_autograd(input, ptr, other);
    ctx->save_for_backward({input, ptr, other});
    ctx->saved_data["sched"] = (int64_t)matmul_schedule_tls();
    return {out};
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::variable_list grad_outs) {
    const auto grad_out = grad_outs[0];
    const auto saved = ctx->get_saved_variables();
    const auto input = saved[0], ptr = saved[1], other = saved[2];
    ScheduleGuard sched((int)ctx->saved_data["sched"].toInt());
    Tensor input_grad, other_grad;
    if (torch::autograd::any_variable_requires_grad({input})) {
      // dX = segment_matmul(dY, ptr, W^T)
      input_grad = segment_matmul_below_autograd(grad_out, ptr, other.transpose(-2, -1));
    }
    if (torch::autograd::any_variable_requires_grad({other})) {
      // dW[b] = X_b^T @ dY_b: one persistent MFMA launch for bf16 / f16 / fp32 and any (K, M) (csrc/hip/matmul_dw.hip,
      // matmul_dw_gen.hip) ...
      other_grad = segment_matmul_dw(input, ptr, grad_out, other.size(0));
    }
    if (torch::autograd::any_variable_requires_grad({other}) && !other_grad.defined()) {
      // ... the reference's per-relation formula elsewhere (fp64, integer types, CPU tensors)
      const auto size = (ptr.narrow(0, 1, ptr.numel() - 1) - ptr.narrow(0, 0, ptr.numel() - 1)).cpu();
      const auto sizes = at::IntArrayRef(size.data_ptr<int64_t>(), (size_t)size.numel());
      const auto xs = input.split_with_sizes(sizes, 0);
      const auto gs = grad_out.split_with_sizes(sizes, 0);
      std::vector<Tensor> parts;
      parts.reserve(xs.size());
      for (size_t i = 0; i < xs.size(); ++i) parts.push_back(at::matmul(xs[i].t(), gs[i]));
      other_grad = at::stack(parts);
    }
    return {input_grad, Tensor(), other_grad};
  }
};

Tensor segment_matmul_autograd(const Tensor& input, const Tensor& ptr, const Tensor& other) {
  return SegmentMatmul::apply(input, ptr, other)[0];
}

// This build only: the weight gradient as an operator of its own,
//   grad_other[b] = input[ptr[b]:ptr[b+1]]^T @ grad_out[ptr[b]:ptr[b+1]]     ([B, K, M], B = ptr.numel() - 1),
// for callers that hold X and dY but never ran the forward through autograd (the backward of the fused R-GCN layer,
// pyg_lib_amd/rgcn.py).  Same kernels as SegmentMatmul::backward; bf16 / f16 / fp32 on the device.
Tensor segment_matmul_grad_other_kernel(const Tensor& input, const Tensor& ptr, const Tensor& grad_out) {
  TORCH_CHECK(input.dim() == 2 && grad_out.dim() == 2 && ptr.dim() == 1 && ptr.numel() >= 1 &&
                  input.size(0) == grad_out.size(0),
              "segment_matmul_grad_other: expected input [N, K], ptr [B + 1], grad_out [N, M]");
  TORCH_CHECK(ptr.scalar_type() == at::kLong, "expected scalar type Long but found ", ptr.scalar_type());
  TORCH_CHECK(input.is_cuda() && grad_out.is_cuda() && input.device() == grad_out.device() &&
                  input.scalar_type() == grad_out.scalar_type(),
              "segment_matmul_grad_other: 'input' and 'grad_out' must share device and dtype");
  Tensor out = segment_matmul_dw(input, ptr, grad_out, ptr.numel() - 1);
  TORCH_CHECK(out.defined(), "segment_matmul_grad_other: float32 / bfloat16 / float16 only (got ", input.scalar_type(), ")");
  return out;
}

// ---------------------------------------------------------------------------------------------
// sampler (schemas: pyg_lib/csrc/sampler/neighbor.cpp:129-147)
// ---------------------------------------------------------------------------------------------
struct SamplerHost {
  hipStream_t stream;
  std::string error;
};

static void* host_alloc(void* user, size_t bytes) {
  auto* h = static_cast<SamplerHost*>(user);
  try {
    return alloc::raw_alloc_with_stream(bytes ? bytes : 16, h->stream);
  } catch (const std::exception& e) {
    h->error = e.what();
    return nullptr;
  }
}

static void host_free(void*, void* ptr) {
  if (ptr) alloc::raw_delete(ptr);
}

// rand_engine.h:79-91.  at::randint(lo, hi, {n}) is empty({n}).random_(lo, hi), and random_ walks
// the CPU generator's mt19937 serially, so filling num_blocks*128 words in one call yields exactly
// the words of that many consecutive 128-word prefetches.
static void host_rng_blocks(void* user, int64_t* words, int64_t num_blocks, int /*first*/) {
  auto* h = static_cast<SamplerHost*>(user);
  try {
    auto buf = at::from_blob(words, {num_blocks * 128}, at::TensorOptions().dtype(at::kLong));
    buf.random_(std::numeric_limits<int64_t>::min(), std::numeric_limits<int64_t>::max());
  } catch (const std::exception& e) {
    h->error = e.what();
  }
}

// Lends the default CPU generator's mt19937 engine to the library for the duration of one call and
// installs the advanced state afterwards (see pyg_hip_sampler_host::mt19937).
struct EngineLoan {
  at::CPUGeneratorImpl* gen;
  at::mt19937 engine;
  at::mt19937_data_pod pod;
  pyg_hip_mt19937 mt;
  bool ok;
  // The generator's mutex is held only while the engine state is copied out and while it is installed again --
  // not across the call: the library may fall back to the host callback (host_rng_blocks -> Tensor.random_(), which
  // takes the same non-recursive mutex), and other threads' CPU RNG use must not wait for device synchronisations.
  // Like the reference (random/cpu/rand_engine.h draws without any lock), concurrent sampler calls are the
  // caller's to serialise.
  EngineLoan()
      : gen(at::get_generator_or_default<at::CPUGeneratorImpl>(c10::nullopt, at::detail::getDefaultCPUGenerator())) {
    {
      std::lock_guard<std::mutex> lock(gen->mutex_);
      engine = gen->engine();
    }
    pod = engine.data();
    ok = engine.is_valid();
    static_assert(sizeof(mt.state) == sizeof(uint32_t) * at::MERSENNE_STATE_N, "mt19937 state size");
    std::memcpy(mt.state, pod.state_.data(), sizeof(mt.state));
    mt.left = pod.left_;
    mt.next = pod.next_;
  }
  pyg_hip_mt19937* ptr() { return ok ? &mt : nullptr; }
  void commit() {
    if (!ok) return;
    std::memcpy(pod.state_.data(), mt.state, sizeof(mt.state));
    pod.left_ = mt.left;
    pod.next_ = mt.next;
    engine.set_data(pod);
    std::lock_guard<std::mutex> lock(gen->mutex_);
    gen->set_engine(engine);
  }
};

static Tensor adopt(int64_t* ptr, at::IntArrayRef sizes, const at::TensorOptions& opts) {
  return at::from_blob(
      ptr, sizes, [](void* p) { alloc::raw_delete(p); }, opts);
}

static void check_index(const Tensor& t, const char* what) {
  TORCH_CHECK(t.is_contiguous(), "Non-contiguous '", what, "'");
  TORCH_CHECK(t.is_cuda(), "pyg (HIP): '", what, "' must live on a HIP device");
  TORCH_CHECK(t.scalar_type() == at::kLong, "pyg (HIP): '", what, "' must be int64 on the device path");
}

// The reference dispatches the sampler on the seeds' integral type (neighbor_kernel.cpp:893,930) and returns
// that type.  The kernels read an int32 CSR (rowptr / col, the large arrays) IN PLACE (`graph` below +
// pyg_hip_relation::index_is32); only the seeds are widened for the call (batch-sized) and the results are
// narrowed back -- same values, same generator stream (dist_neighbor_sample included).
struct IndexArgs {
  at::ScalarType dtype = at::kLong;
  std::vector<Tensor> keep;  // widened copies stay alive until the call returns
  const int64_t* ptr(const Tensor& t, const char* what) {
    TORCH_CHECK(t.is_contiguous(), "Non-contiguous '", what, "'");
    TORCH_CHECK(t.is_cuda(), "pyg (HIP): '", what, "' must live on a HIP device");
    TORCH_CHECK(t.scalar_type() == dtype, "pyg (HIP): '", what, "' must have the seeds' dtype (", dtype, ")");
    if (dtype == at::kLong) return t.data_ptr<int64_t>();
    keep.push_back(t.to(at::kLong));
    return keep.back().data_ptr<int64_t>();
  }
  // rowptr / col of the sampled graph: raw pointer of either width, no copy
  const int64_t* graph(const Tensor& t, const char* what) const {
    TORCH_CHECK(t.is_contiguous(), "Non-contiguous '", what, "'");
    TORCH_CHECK(t.is_cuda(), "pyg (HIP): '", what, "' must live on a HIP device");
    TORCH_CHECK(t.scalar_type() == dtype, "pyg (HIP): '", what, "' must have the seeds' dtype (", dtype, ")");
    return static_cast<const int64_t*>(t.data_ptr());
  }
  int32_t is32() const { return dtype == at::kInt ? 1 : 0; }
  Tensor narrow(const Tensor& t) const { return dtype == at::kLong ? t : t.to(dtype); }
};

static at::ScalarType index_dtype(const Tensor& seed) {
  TORCH_CHECK(seed.scalar_type() == at::kLong || seed.scalar_type() == at::kInt,
              "pyg (HIP): indices must be int64 or int32");
  return seed.scalar_type();
}

struct SampleOutput {
  std::vector<Tensor> node_id, row, col, edge_id;
  std::vector<std::vector<int64_t>> nodes_per_hop, edges_per_hop;
};

// PYG_HIP_OP_TIMING=1: host time of the sampler operators' parts on stderr (arguments -> library call -> results adopted ->
// Dict results built), microseconds since the operator was entered
struct OpTiming {
  static bool on() {
    static const bool v = [] { const char* e = getenv("PYG_HIP_OP_TIMING"); return e && atoi(e) != 0; }();
    return v;
  }
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  std::string line;
  void mark(const char* what) {
    if (!on()) return;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    char buf[64];
    snprintf(buf, sizeof(buf), " %s=%.1f", what, us);
    line += buf;
  }
  ~OpTiming() {
    if (on() && !line.empty()) fprintf(stderr, "[pyg op timing] us:%s\n", line.c_str());
  }
};
static thread_local OpTiming* g_op_timing = nullptr;

static SampleOutput run_sampler(const std::vector<pyg_hip_relation>& rels,
                                const std::vector<pyg_hip_seed_set>& seeds,
                                const std::vector<const int64_t*>& node_time, bool temporal_last,
                                int num_node_types, int L, bool csc, bool replace, bool disjoint,
                                bool return_edge_id, const at::Device& device) {
  DeviceGuard guard(device);
  const auto opts = at::TensorOptions().dtype(at::kLong).device(device);
  SamplerHost host;
  host.stream = current_hip_stream(device.index());
  // Fast path for the random words: the CPU generator's mt19937 engine is continued on the device
  // (the generator ends up exactly where the reference's at::randint / random_ calls would leave it).
  EngineLoan loan;
  pyg_hip_sampler_host cb{&host, &host_alloc, &host_free, &host_rng_blocks, loan.ptr()};
  const int T = num_node_types, E = (int)rels.size();
  std::vector<int64_t*> node_id((size_t)T, nullptr), row((size_t)std::max(E, 1), nullptr),
      col((size_t)std::max(E, 1), nullptr), eid((size_t)std::max(E, 1), nullptr);
  std::vector<int64_t> num_nodes((size_t)T, 0), num_edges((size_t)std::max(E, 1), 0);
  std::vector<int64_t> nph((size_t)T * (L + 1), 0), eph((size_t)std::max(E * L, 1), 0);
  pyg_hip_sample_result res;
  res.node_id = node_id.data();
  res.num_nodes = num_nodes.data();
  res.nodes_per_hop_host = nph.data();
  res.row = row.data();
  res.col = col.data();
  res.edge_id = eid.data();
  res.num_edges = num_edges.data();
  res.edges_per_hop_host = eph.data();
  res.rng_blocks = 0;
  const int rc = pyg_hip_hetero_neighbor_sample(T, E, rels.data(), (int)seeds.size(), seeds.data(),
                                                node_time.empty() ? nullptr : node_time.data(),
                                                temporal_last, L, csc, replace, disjoint, return_edge_id,
                                                &cb, &res, host.stream);
  if (g_op_timing) g_op_timing->mark("library_done");
  if (rc == PYG_HIP_OK) loan.commit();
  TORCH_CHECK(host.error.empty(), host.error);
  check_status(rc);
  SampleOutput out;
  for (int t = 0; t < T; ++t) {
    const int64_t n = num_nodes[(size_t)t];
    out.node_id.push_back(disjoint ? adopt(node_id[(size_t)t], {n, 2}, opts) : adopt(node_id[(size_t)t], {n}, opts));
    out.nodes_per_hop.emplace_back(nph.begin() + (size_t)t * (L + 1), nph.begin() + (size_t)(t + 1) * (L + 1));
  }
  for (int e = 0; e < E; ++e) {
    const int64_t n = num_edges[(size_t)e];
    out.row.push_back(adopt(row[(size_t)e], {n}, opts));
    out.col.push_back(adopt(col[(size_t)e], {n}, opts));
    if (return_edge_id) out.edge_id.push_back(adopt(eid[(size_t)e], {n}, opts));
    out.edges_per_hop.emplace_back(eph.begin() + (size_t)e * L, eph.begin() + (size_t)(e + 1) * L);
  }
  if (g_op_timing) g_op_timing->mark("adopted");
  return out;
}

// ---- K independent batches at once (pyg_hip_hetero_neighbor_sample_batched) ----------------------------------------
// Private streams of the batched sampler, per device: batch b runs on lane b % lanes.  Blocks the lanes allocate go back to
// the caching allocator's pools of THESE streams when the caller drops the outputs; every batched call first orders its
// lanes behind the caller's current stream (the C-ABI call does), so a block is never reused under a consumer the caller
// queued before the call.
static std::vector<hipStream_t> batch_lanes(int device, int want) {
  static std::mutex mu;
  static std::vector<std::vector<hipStream_t>> per_device(64);
  std::lock_guard<std::mutex> lock(mu);
  auto& v = per_device[(size_t)(device < 0 || device >= 64 ? 0 : device)];
  while ((int)v.size() < want) {
    hipStream_t st = nullptr;
    TORCH_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess, "pyg (HIP): hipStreamCreate failed");
    v.push_back(st);
  }
  return std::vector<hipStream_t>(v.begin(), v.begin() + want);
}

// Lanes = batches in flight.  PYG_HIP_SAMPLER_LANES, default 8 -- 2 for heterogeneous graphs: a hetero batch's launches carry
// several relations each and already fill most of the chip, more than two of them in flight only slow one another down
// (C5 graph, K = 8: 2 lanes 1.13 x the single-batch loop, 8 lanes 0.86 x; the C3 graph gives 1.07 - 1.09 x from 2 lanes up:
// profiles/NOTES_r6.md section 2g)
static int batch_lane_count(size_t K, bool hetero) {
  static const int cap = [] {
    const char* e = getenv("PYG_HIP_SAMPLER_LANES");
    const int v = e ? atoi(e) : 0;
    return v < 1 ? 0 : (v > 16 ? 16 : v);
  }();
  const int c = cap ? cap : (hetero ? 2 : 8);
  return (int)std::min<size_t>(K, (size_t)c);
}

// Batch b continues the mt19937 stream torch.manual_seed(generator_seeds[b]) would start (CPUGeneratorImpl::set_current_seed
// installs at::mt19937(seed)); the process's default generator is not touched.
static std::vector<SampleOutput> run_sampler_batched(const std::vector<pyg_hip_relation>& rels,
                                                     const std::vector<std::vector<pyg_hip_seed_set>>& seeds,
                                                     const std::vector<int64_t>& generator_seeds,
                                                     const std::vector<const int64_t*>& node_time, bool temporal_last,
                                                     int num_node_types, int L, bool csc, bool replace, bool disjoint,
                                                     bool return_edge_id, const at::Device& device) {
  const size_t K = seeds.size();
  TORCH_CHECK(generator_seeds.size() == K, "neighbor_sample_batched: one generator seed per batch expected (", K, " batches, ",
              generator_seeds.size(), " seeds)");
  DeviceGuard guard(device);
  const auto opts = at::TensorOptions().dtype(at::kLong).device(device);
  const int T = num_node_types, E = (int)rels.size();
  const auto lanes = batch_lanes(device.index(), batch_lane_count(K, rels.size() > 1));
  struct PerBatch {
    SamplerHost host;
    pyg_hip_mt19937 mt;
    pyg_hip_sampler_host cb;
    std::vector<int64_t*> node_id, row, col, eid;
    std::vector<int64_t> num_nodes, num_edges, nph, eph;
    pyg_hip_sample_result res;
  };
  std::vector<PerBatch> pb(K);
  std::vector<pyg_hip_sample_batch> batches(K);
  for (size_t b = 0; b < K; ++b) {
    PerBatch& p = pb[b];
    p.host.stream = lanes[b % lanes.size()];
    at::mt19937 engine((uint64_t)generator_seeds[b]);
    const at::mt19937_data_pod pod = engine.data();
    std::memcpy(p.mt.state, pod.state_.data(), sizeof(p.mt.state));
    p.mt.left = pod.left_;
    p.mt.next = pod.next_;
    p.cb = pyg_hip_sampler_host{&p.host, &host_alloc, &host_free, &host_rng_blocks, &p.mt};
    p.node_id.assign((size_t)T, nullptr);
    p.row.assign((size_t)std::max(E, 1), nullptr);
    p.col.assign((size_t)std::max(E, 1), nullptr);
    p.eid.assign((size_t)std::max(E, 1), nullptr);
    p.num_nodes.assign((size_t)T, 0);
    p.num_edges.assign((size_t)std::max(E, 1), 0);
    p.nph.assign((size_t)T * (L + 1), 0);
    p.eph.assign((size_t)std::max(E * L, 1), 0);
    p.res.node_id = p.node_id.data();
    p.res.num_nodes = p.num_nodes.data();
    p.res.nodes_per_hop_host = p.nph.data();
    p.res.row = p.row.data();
    p.res.col = p.col.data();
    p.res.edge_id = p.eid.data();
    p.res.num_edges = p.num_edges.data();
    p.res.edges_per_hop_host = p.eph.data();
    p.res.rng_blocks = 0;
    batches[b].num_seed_sets = (int)seeds[b].size();
    batches[b].seeds_host = seeds[b].data();
    batches[b].host = &p.cb;
    batches[b].result = &p.res;
    batches[b].stream = p.host.stream;
  }
  const int rc = pyg_hip_hetero_neighbor_sample_batched(T, E, rels.data(), node_time.empty() ? nullptr : node_time.data(),
                                                        temporal_last, L, csc, replace, disjoint, return_edge_id, (int)K,
                                                        batches.data(), current_hip_stream(device.index()));
  // adopt whatever was handed out (also on failure: the blocks must go back to the allocator)
  std::vector<SampleOutput> outs(K);
  for (size_t b = 0; b < K; ++b) {
    PerBatch& p = pb[b];
    const bool ok = batches[b].status == PYG_HIP_OK;
    for (int t = 0; t < T; ++t) {
      if (!ok || !p.node_id[(size_t)t]) continue;
      const int64_t n = p.num_nodes[(size_t)t];
      outs[b].node_id.push_back(disjoint ? adopt(p.node_id[(size_t)t], {n, 2}, opts) : adopt(p.node_id[(size_t)t], {n}, opts));
      outs[b].nodes_per_hop.emplace_back(p.nph.begin() + (size_t)t * (L + 1), p.nph.begin() + (size_t)(t + 1) * (L + 1));
    }
    for (int e = 0; e < E; ++e) {
      if (!ok) continue;
      const int64_t n = p.num_edges[(size_t)e];
      outs[b].row.push_back(adopt(p.row[(size_t)e], {n}, opts));
      outs[b].col.push_back(adopt(p.col[(size_t)e], {n}, opts));
      if (return_edge_id) outs[b].edge_id.push_back(adopt(p.eid[(size_t)e], {n}, opts));
      outs[b].edges_per_hop.emplace_back(p.eph.begin() + (size_t)e * L, p.eph.begin() + (size_t)(e + 1) * L);
    }
  }
  // (errors only after every block that was handed out has an owner)
  for (size_t b = 0; b < K; ++b) TORCH_CHECK(pb[b].host.error.empty(), pb[b].host.error);
  check_status(rc);
  return outs;
}

static void check_modes(bool has_node_time, bool has_edge_time, bool has_seed_time, bool has_weight,
                        bool directed, bool disjoint, const std::string& temporal_strategy) {
  // precondition checks of the reference kernel, sampler/cpu/neighbor_kernel.cpp:34-36,354-380,501
  TORCH_CHECK(temporal_strategy == "uniform" || temporal_strategy == "last", "No valid temporal strategy found");
  TORCH_CHECK(!has_node_time || disjoint, "Temporal sampling needs to create disjoint subgraphs");
  TORCH_CHECK(!has_edge_time || disjoint, "Temporal sampling needs to create disjoint subgraphs");
  TORCH_CHECK(!(has_node_time && has_edge_time), "Only one of node-level or edge-level sampling is supported ");
  TORCH_CHECK(!has_edge_time || has_seed_time, "Seed time needs to be specified");
  TORCH_CHECK(!(has_node_time && has_weight), "Biased node temporal sampling not yet supported");
  TORCH_CHECK(!(has_edge_time && has_weight), "Biased edge temporal sampling not yet supported");
  TORCH_CHECK(directed, "Undirected subgraphs not yet supported");
}

// biased sampling: per-edge weights of one relation (neighbor_kernel.cpp:52 narrows them per row)
static void set_weight(pyg_hip_relation& rel, const Tensor& w, int64_t num_cols) {
  TORCH_CHECK(w.is_cuda(), "pyg (HIP): 'edge_weight' must live on a HIP device");
  TORCH_CHECK(w.is_contiguous(), "Non-contiguous 'edge_weight'");
  TORCH_CHECK(w.dim() == 1 && w.numel() == num_cols, "pyg (HIP): 'edge_weight' needs one entry per edge");
  TORCH_CHECK(w.scalar_type() == at::kFloat || w.scalar_type() == at::kDouble,
              "pyg (HIP): 'edge_weight' must be float32 or float64");
  rel.edge_weight = w.data_ptr();
  rel.edge_weight_dtype = w.scalar_type() == at::kDouble ? PYG_F64 : PYG_F32;
}

static const int64_t* time_ptr(const Tensor& t, const char* what) {
  TORCH_CHECK(t.is_contiguous(), "Non-contiguous '", what, "'");
  TORCH_CHECK(t.is_cuda(), "pyg (HIP): '", what, "' must live on a HIP device");
  TORCH_CHECK(t.scalar_type() == at::kLong, "pyg (HIP): '", what, "' must be int64");  // temporal_t, :393-394
  return t.data_ptr<int64_t>();
}

std::tuple<Tensor, Tensor, Tensor, c10::optional<Tensor>, std::vector<int64_t>, std::vector<int64_t>>
neighbor_sample_kernel(const Tensor& rowptr, const Tensor& col, const Tensor& seed,
                       const std::vector<int64_t>& num_neighbors, const c10::optional<Tensor>& node_time,
                       const c10::optional<Tensor>& edge_time, const c10::optional<Tensor>& seed_time,
                       const c10::optional<Tensor>& edge_weight, bool csc, bool replace, bool directed,
                       bool disjoint, std::string temporal_strategy, bool return_edge_id) {
  PYG_TRACE("pyg::neighbor_sample");
  check_modes(node_time.has_value(), edge_time.has_value(), seed_time.has_value(), edge_weight.has_value(),
              directed, disjoint, temporal_strategy);
  IndexArgs ix;
  ix.dtype = index_dtype(seed);
  std::vector<pyg_hip_relation> rels(1);
  rels[0].rowptr = ix.graph(rowptr, "rowptr");
  rels[0].num_rows = rowptr.numel() - 1;
  rels[0].col = ix.graph(col, "col");
  rels[0].num_cols = col.numel();
  rels[0].src_type = 0;
  rels[0].dst_type = 0;
  rels[0].num_neighbors_host = num_neighbors.data();
  rels[0].edge_time = edge_time.has_value() ? time_ptr(edge_time.value(), "edge_time") : nullptr;
  rels[0].edge_weight = nullptr;
  rels[0].edge_weight_dtype = 0;
  rels[0].index_is32 = ix.is32();
  if (edge_weight.has_value()) set_weight(rels[0], edge_weight.value(), col.numel());
  std::vector<pyg_hip_seed_set> seeds(1);
  seeds[0].node_type = 0;
  seeds[0].reserved = 0;
  seeds[0].seed = ix.ptr(seed, "seed");
  seeds[0].num_seed = seed.numel();
  seeds[0].seed_time = seed_time.has_value() ? time_ptr(seed_time.value(), "seed_time") : nullptr;
  std::vector<const int64_t*> ntime;
  if (node_time.has_value()) ntime.push_back(time_ptr(node_time.value(), "node_time"));
  auto out = run_sampler(rels, seeds, ntime, temporal_strategy == "last", 1, (int)num_neighbors.size(), csc,
                         replace, disjoint, return_edge_id, rowptr.device());
  c10::optional<Tensor> eid = c10::nullopt;
  if (return_edge_id) eid = ix.narrow(out.edge_id[0]);
  return std::make_tuple(ix.narrow(out.row[0]), ix.narrow(out.col[0]), ix.narrow(out.node_id[0]), eid,
                         out.nodes_per_hop[0], out.edges_per_hop[0]);
}

// This build only: K mini-batches of pyg::neighbor_sample in one call.  Batch b = neighbor_sample(rowptr, col, seeds[b], ...)
// under torch.manual_seed(generator_seeds[b]) -- bit for bit -- but the batches overlap on the device.  Returns the
// per-batch row / col / node_id / edge_id lists and the per-hop counts as [K, L + 1] / [K, L] CPU tensors.
std::tuple<std::vector<Tensor>, std::vector<Tensor>, std::vector<Tensor>, std::vector<Tensor>, Tensor, Tensor>
neighbor_sample_batched_kernel(const Tensor& rowptr, const Tensor& col, const std::vector<Tensor>& seeds,
                               const std::vector<int64_t>& num_neighbors, const std::vector<int64_t>& generator_seeds,
                               const c10::optional<Tensor>& node_time, const c10::optional<Tensor>& edge_time,
                               const c10::optional<std::vector<Tensor>>& seed_times, const c10::optional<Tensor>& edge_weight,
                               bool csc, bool replace, bool directed, bool disjoint, std::string temporal_strategy,
                               bool return_edge_id) {
  PYG_TRACE("pyg::neighbor_sample_batched");
  check_modes(node_time.has_value(), edge_time.has_value(), seed_times.has_value(), edge_weight.has_value(), directed,
              disjoint, temporal_strategy);
  const size_t K = seeds.size();
  TORCH_CHECK(!seed_times.has_value() || seed_times.value().size() == K, "neighbor_sample_batched: one seed_time per batch");
  const int64_t L = (int64_t)num_neighbors.size();
  const auto cpu_long = at::TensorOptions().dtype(at::kLong);
  if (K == 0)
    return std::make_tuple(std::vector<Tensor>(), std::vector<Tensor>(), std::vector<Tensor>(), std::vector<Tensor>(),
                           at::zeros({0, L + 1}, cpu_long), at::zeros({0, L}, cpu_long));
  IndexArgs ix;
  ix.dtype = index_dtype(seeds[0]);
  std::vector<pyg_hip_relation> rels(1);
  rels[0].rowptr = ix.graph(rowptr, "rowptr");
  rels[0].num_rows = rowptr.numel() - 1;
  rels[0].col = ix.graph(col, "col");
  rels[0].num_cols = col.numel();
  rels[0].src_type = 0;
  rels[0].dst_type = 0;
  rels[0].num_neighbors_host = num_neighbors.data();
  rels[0].edge_time = edge_time.has_value() ? time_ptr(edge_time.value(), "edge_time") : nullptr;
  rels[0].edge_weight = nullptr;
  rels[0].edge_weight_dtype = 0;
  rels[0].index_is32 = ix.is32();
  if (edge_weight.has_value()) set_weight(rels[0], edge_weight.value(), col.numel());
  std::vector<std::vector<pyg_hip_seed_set>> sets(K, std::vector<pyg_hip_seed_set>(1));
  for (size_t b = 0; b < K; ++b) {
    sets[b][0].node_type = 0;
    sets[b][0].reserved = 0;
    sets[b][0].seed = ix.ptr(seeds[b], "seed");
    sets[b][0].num_seed = seeds[b].numel();
    sets[b][0].seed_time = seed_times.has_value() ? time_ptr(seed_times.value()[b], "seed_time") : nullptr;
  }
  std::vector<const int64_t*> ntime;
  if (node_time.has_value()) ntime.push_back(time_ptr(node_time.value(), "node_time"));
  auto outs = run_sampler_batched(rels, sets, generator_seeds, ntime, temporal_strategy == "last", 1, (int)L, csc, replace,
                                  disjoint, return_edge_id, rowptr.device());
  std::vector<Tensor> row, colv, node, eid;
  Tensor nph = at::zeros({(int64_t)K, L + 1}, cpu_long), eph = at::zeros({(int64_t)K, L}, cpu_long);
  for (size_t b = 0; b < K; ++b) {
    row.push_back(ix.narrow(outs[b].row[0]));
    colv.push_back(ix.narrow(outs[b].col[0]));
    node.push_back(ix.narrow(outs[b].node_id[0]));
    if (return_edge_id) eid.push_back(ix.narrow(outs[b].edge_id[0]));
    std::memcpy(nph.data_ptr<int64_t>() + b * (size_t)(L + 1), outs[b].nodes_per_hop[0].data(), sizeof(int64_t) * (size_t)(L + 1));
    if (L > 0) std::memcpy(eph.data_ptr<int64_t>() + b * (size_t)L, outs[b].edges_per_hop[0].data(), sizeof(int64_t) * (size_t)L);
  }
  return std::make_tuple(row, colv, node, eid, nph, eph);
}

// This build only: K mini-batches of pyg::hetero_neighbor_sample (uniform sampling; no temporal / biased options here).
std::tuple<std::vector<c10::Dict<rel_type, Tensor>>, std::vector<c10::Dict<rel_type, Tensor>>,
           std::vector<c10::Dict<node_type, Tensor>>, std::vector<c10::Dict<rel_type, Tensor>>,
           std::vector<c10::Dict<node_type, std::vector<int64_t>>>, std::vector<c10::Dict<rel_type, std::vector<int64_t>>>>
hetero_neighbor_sample_batched_kernel(const std::vector<node_type>& node_types, const std::vector<edge_type>& edge_types,
                                      const c10::Dict<rel_type, Tensor>& rowptr_dict, const c10::Dict<rel_type, Tensor>& col_dict,
                                      const std::vector<c10::Dict<node_type, Tensor>>& seed_dicts,
                                      const c10::Dict<rel_type, std::vector<int64_t>>& num_neighbors_dict,
                                      const std::vector<int64_t>& generator_seeds,
                                      const c10::optional<c10::Dict<node_type, Tensor>>& node_time_dict,
                                      const c10::optional<c10::Dict<rel_type, Tensor>>& edge_time_dict,
                                      const c10::optional<std::vector<c10::Dict<node_type, Tensor>>>& seed_time_dicts,
                                      const c10::optional<c10::Dict<rel_type, Tensor>>& edge_weight_dict, bool csc, bool replace,
                                      bool directed, bool disjoint, std::string temporal_strategy, bool return_edge_id) {
  PYG_TRACE("pyg::hetero_neighbor_sample_batched");
  // the modes of pyg::hetero_neighbor_sample (sampler/neighbor.cpp:137-147 is one entry for all of them)
  check_modes(node_time_dict.has_value(), edge_time_dict.has_value(), seed_time_dicts.has_value(), edge_weight_dict.has_value(),
              directed, disjoint, temporal_strategy);
  TORCH_CHECK(!seed_time_dicts.has_value() || seed_time_dicts.value().size() == seed_dicts.size(),
              "hetero_neighbor_sample_batched: one seed_time dict per batch");
  std::unordered_map<std::string, int> nt_index;
  for (size_t i = 0; i < node_types.size(); ++i) nt_index[node_types[i]] = (int)i;
  const size_t K = seed_dicts.size();
  TORCH_CHECK(K > 0 && seed_dicts[0].size() > 0, "hetero_neighbor_sample_batched: no seeds given");
  IndexArgs ix;
  ix.dtype = index_dtype(seed_dicts[0].begin()->value());
  size_t L = 0;
  std::vector<pyg_hip_relation> rels(edge_types.size());
  std::vector<std::vector<int64_t>> fanouts(edge_types.size());
  c10::optional<at::Device> device;
  for (size_t e = 0; e < edge_types.size(); ++e) {
    const auto& k = edge_types[e];
    const auto rel = to_rel_type(k);
    const Tensor& rowptr = rowptr_dict.at(rel);
    const Tensor& col = col_dict.at(rel);
    if (!device.has_value()) device = rowptr.device();
    fanouts[e] = num_neighbors_dict.at(rel);
    L = std::max(L, fanouts[e].size());
    TORCH_CHECK(nt_index.count(std::get<0>(k)) && nt_index.count(std::get<2>(k)),
                "hetero_neighbor_sample_batched: edge type names an unknown node type");
    rels[e] = pyg_hip_relation{};
    rels[e].rowptr = ix.graph(rowptr, "rowptr");
    rels[e].num_rows = rowptr.numel() - 1;
    rels[e].col = ix.graph(col, "col");
    rels[e].num_cols = col.numel();
    rels[e].src_type = nt_index[std::get<0>(k)];
    rels[e].dst_type = nt_index[std::get<2>(k)];
    rels[e].index_is32 = ix.is32();
    if (edge_time_dict.has_value() && edge_time_dict.value().contains(rel))
      rels[e].edge_time = time_ptr(edge_time_dict.value().at(rel), "edge_time");
    if (edge_weight_dict.has_value() && edge_weight_dict.value().contains(rel))
      set_weight(rels[e], edge_weight_dict.value().at(rel), col.numel());
  }
  for (size_t e = 0; e < edge_types.size(); ++e) {
    TORCH_CHECK(fanouts[e].size() == L, "hetero_neighbor_sample_batched: all relations must list ", L, " hops");
    rels[e].num_neighbors_host = fanouts[e].data();
  }
  TORCH_CHECK(device.has_value(), "hetero_neighbor_sample_batched: no tensors given");
  std::vector<std::vector<pyg_hip_seed_set>> sets(K);
  for (size_t b = 0; b < K; ++b)
    for (const auto& kv : seed_dicts[b]) {
      TORCH_CHECK(nt_index.count(kv.key()), "hetero_neighbor_sample_batched: seed type '", kv.key(), "' is not a node type");
      pyg_hip_seed_set st;
      st.node_type = nt_index[kv.key()];
      st.reserved = 0;
      st.seed = ix.ptr(kv.value(), "seed");
      st.num_seed = kv.value().numel();
      st.seed_time = nullptr;
      if (seed_time_dicts.has_value()) st.seed_time = time_ptr(seed_time_dicts.value()[b].at(kv.key()), "seed_time");
      sets[b].push_back(st);
    }
  std::vector<const int64_t*> ntime;
  if (node_time_dict.has_value()) {
    ntime.assign(node_types.size(), nullptr);
    for (const auto& kv : node_time_dict.value()) {
      TORCH_CHECK(nt_index.count(kv.key()), "hetero_neighbor_sample_batched: time given for unknown node type '", kv.key(), "'");
      ntime[(size_t)nt_index[kv.key()]] = time_ptr(kv.value(), "node_time");
    }
  }
  auto outs = run_sampler_batched(rels, sets, generator_seeds, ntime, temporal_strategy == "last", (int)node_types.size(), (int)L,
                                  csc, replace, disjoint, return_edge_id, device.value());
  std::vector<c10::Dict<rel_type, Tensor>> o_row, o_col, o_eid;
  std::vector<c10::Dict<node_type, Tensor>> o_node;
  std::vector<c10::Dict<node_type, std::vector<int64_t>>> o_nph;
  std::vector<c10::Dict<rel_type, std::vector<int64_t>>> o_eph;
  for (size_t b = 0; b < K; ++b) {
    c10::Dict<rel_type, Tensor> r, c, ei;
    c10::Dict<node_type, Tensor> n;
    c10::Dict<node_type, std::vector<int64_t>> np;
    c10::Dict<rel_type, std::vector<int64_t>> ep;
    for (size_t t = 0; t < node_types.size(); ++t) {
      n.insert(node_types[t], ix.narrow(outs[b].node_id[t]));
      np.insert(node_types[t], outs[b].nodes_per_hop[t]);
    }
    for (size_t e = 0; e < edge_types.size(); ++e) {
      const auto rel = to_rel_type(edge_types[e]);
      r.insert(rel, ix.narrow(outs[b].row[e]));
      c.insert(rel, ix.narrow(outs[b].col[e]));
      ep.insert(rel, outs[b].edges_per_hop[e]);
      if (return_edge_id) ei.insert(rel, ix.narrow(outs[b].edge_id[e]));
    }
    o_row.push_back(r), o_col.push_back(c), o_node.push_back(n), o_eid.push_back(ei), o_nph.push_back(np), o_eph.push_back(ep);
  }
  return std::make_tuple(o_row, o_col, o_node, o_eid, o_nph, o_eph);
}

// pyg_binding_cpu.cpp
std::tuple<c10::Dict<std::string, Tensor>, c10::Dict<std::string, Tensor>, c10::Dict<std::string, Tensor>,
           c10::optional<c10::Dict<std::string, Tensor>>, c10::Dict<std::string, std::vector<int64_t>>,
           c10::Dict<std::string, std::vector<int64_t>>>
hetero_neighbor_sample_on_cpu(const std::vector<std::string>& node_types,
                              const std::vector<std::tuple<std::string, std::string, std::string>>& edge_types,
                              const c10::Dict<std::string, Tensor>& rowptr_dict, const c10::Dict<std::string, Tensor>& col_dict,
                              const c10::Dict<std::string, Tensor>& seed_dict,
                              const c10::Dict<std::string, std::vector<int64_t>>& num_neighbors_dict,
                              const c10::optional<c10::Dict<std::string, Tensor>>& node_time_dict,
                              const c10::optional<c10::Dict<std::string, Tensor>>& edge_time_dict,
                              const c10::optional<c10::Dict<std::string, Tensor>>& seed_time_dict,
                              const c10::optional<c10::Dict<std::string, Tensor>>& edge_weight_dict, bool csc, bool replace,
                              bool directed, bool disjoint, std::string temporal_strategy, bool return_edge_id);

std::tuple<c10::Dict<rel_type, Tensor>, c10::Dict<rel_type, Tensor>, c10::Dict<node_type, Tensor>,
           c10::optional<c10::Dict<rel_type, Tensor>>, c10::Dict<node_type, std::vector<int64_t>>,
           c10::Dict<rel_type, std::vector<int64_t>>>
hetero_neighbor_sample_kernel(const std::vector<node_type>& node_types, const std::vector<edge_type>& edge_types,
                              const c10::Dict<rel_type, Tensor>& rowptr_dict,
                              const c10::Dict<rel_type, Tensor>& col_dict,
                              const c10::Dict<node_type, Tensor>& seed_dict,
                              const c10::Dict<rel_type, std::vector<int64_t>>& num_neighbors_dict,
                              const c10::optional<c10::Dict<node_type, Tensor>>& node_time_dict,
                              const c10::optional<c10::Dict<rel_type, Tensor>>& edge_time_dict,
                              const c10::optional<c10::Dict<node_type, Tensor>>& seed_time_dict,
                              const c10::optional<c10::Dict<rel_type, Tensor>>& edge_weight_dict, bool csc,
                              bool replace, bool directed, bool disjoint, std::string temporal_strategy,
                              bool return_edge_id) {
  // BackendSelect: tensors inside Dicts cannot drive dispatch (sampler/cpu/neighbor_kernel.cpp:985-991) -- a graph
  // held in CPU tensors goes to the CPU kernel, a device graph to the HIP sampler
  {
    bool on_device = false;
    for (const auto& kv : rowptr_dict) on_device = on_device || kv.value().is_cuda();
    for (const auto& kv : seed_dict) on_device = on_device || kv.value().is_cuda();
    if (!on_device)
      return hetero_neighbor_sample_on_cpu(node_types, edge_types, rowptr_dict, col_dict, seed_dict, num_neighbors_dict,
                                           node_time_dict, edge_time_dict, seed_time_dict, edge_weight_dict, csc, replace,
                                           directed, disjoint, temporal_strategy, return_edge_id);
  }
  PYG_TRACE("pyg::hetero_neighbor_sample");
  OpTiming timing;
  g_op_timing = OpTiming::on() ? &timing : nullptr;
  check_modes(node_time_dict.has_value(), edge_time_dict.has_value(), seed_time_dict.has_value(),
              edge_weight_dict.has_value(), directed, disjoint, temporal_strategy);
  std::unordered_map<std::string, int> nt_index;
  for (size_t i = 0; i < node_types.size(); ++i) nt_index[node_types[i]] = (int)i;
  size_t L = 0;
  IndexArgs ix;
  TORCH_CHECK(seed_dict.size() > 0, "hetero_neighbor_sample: no seeds given");
  ix.dtype = index_dtype(seed_dict.begin()->value());  // neighbor_kernel.cpp:930
  std::vector<pyg_hip_relation> rels(edge_types.size());
  std::vector<std::vector<int64_t>> fanouts(edge_types.size());
  c10::optional<at::Device> device;
  for (size_t e = 0; e < edge_types.size(); ++e) {
    const auto& k = edge_types[e];
    const auto rel = to_rel_type(k);
    const Tensor& rowptr = rowptr_dict.at(rel);
    const Tensor& col = col_dict.at(rel);
    if (!device.has_value()) device = rowptr.device();
    fanouts[e] = num_neighbors_dict.at(rel);
    L = std::max(L, fanouts[e].size());
    TORCH_CHECK(nt_index.count(std::get<0>(k)) && nt_index.count(std::get<2>(k)),
                "hetero_neighbor_sample: edge type names an unknown node type");
    rels[e].rowptr = ix.graph(rowptr, "rowptr");
    rels[e].num_rows = rowptr.numel() - 1;
    rels[e].col = ix.graph(col, "col");
    rels[e].num_cols = col.numel();
    rels[e].src_type = nt_index[std::get<0>(k)];
    rels[e].dst_type = nt_index[std::get<2>(k)];
    rels[e].edge_time = nullptr;
    if (edge_time_dict.has_value() && edge_time_dict.value().contains(rel))
      rels[e].edge_time = time_ptr(edge_time_dict.value().at(rel), "edge_time");
    rels[e].edge_weight = nullptr;
    rels[e].edge_weight_dtype = 0;
    rels[e].index_is32 = ix.is32();
    if (edge_weight_dict.has_value() && edge_weight_dict.value().contains(rel))
      set_weight(rels[e], edge_weight_dict.value().at(rel), col.numel());
  }
  for (size_t e = 0; e < edge_types.size(); ++e) {
    TORCH_CHECK(fanouts[e].size() == L, "hetero_neighbor_sample: all relations must list ", L, " hops");
    rels[e].num_neighbors_host = fanouts[e].data();
  }
  std::vector<pyg_hip_seed_set> seeds;
  for (const auto& kv : seed_dict) {  // c10::Dict iterates in insertion order, as the reference relies on
    const Tensor& seed = kv.value();
    if (!device.has_value()) device = seed.device();
    TORCH_CHECK(nt_index.count(kv.key()), "hetero_neighbor_sample: seed type '", kv.key(), "' is not a node type");
    pyg_hip_seed_set s;
    s.node_type = nt_index[kv.key()];
    s.reserved = 0;
    s.seed = ix.ptr(seed, "seed");
    s.num_seed = seed.numel();
    s.seed_time = nullptr;
    if (seed_time_dict.has_value()) s.seed_time = time_ptr(seed_time_dict.value().at(kv.key()), "seed_time");
    seeds.push_back(s);
  }
  std::vector<const int64_t*> ntime;
  if (node_time_dict.has_value()) {
    ntime.assign(node_types.size(), nullptr);
    for (const auto& kv : node_time_dict.value()) {
      TORCH_CHECK(nt_index.count(kv.key()), "hetero_neighbor_sample: time given for unknown node type '", kv.key(), "'");
      ntime[(size_t)nt_index[kv.key()]] = time_ptr(kv.value(), "node_time");
    }
  }
  TORCH_CHECK(device.has_value(), "hetero_neighbor_sample: no tensors given");
  timing.mark("args_ready");
  auto out = run_sampler(rels, seeds, ntime, temporal_strategy == "last", (int)node_types.size(), (int)L, csc,
                         replace, disjoint, return_edge_id, device.value());
  c10::Dict<rel_type, Tensor> out_row, out_col;
  c10::Dict<node_type, Tensor> out_node;
  c10::optional<c10::Dict<rel_type, Tensor>> out_eid;
  if (return_edge_id) out_eid = c10::Dict<rel_type, Tensor>();
  c10::Dict<node_type, std::vector<int64_t>> out_nph;
  c10::Dict<rel_type, std::vector<int64_t>> out_eph;
  for (size_t t = 0; t < node_types.size(); ++t) {
    out_node.insert(node_types[t], ix.narrow(out.node_id[t]));
    out_nph.insert(node_types[t], out.nodes_per_hop[t]);
  }
  for (size_t e = 0; e < edge_types.size(); ++e) {
    const auto rel = to_rel_type(edge_types[e]);
    out_row.insert(rel, ix.narrow(out.row[e]));
    out_col.insert(rel, ix.narrow(out.col[e]));
    out_eph.insert(rel, out.edges_per_hop[e]);
    if (return_edge_id) out_eid.value().insert(rel, ix.narrow(out.edge_id[e]));
  }
  timing.mark("dicts_built");
  g_op_timing = nullptr;
  return std::make_tuple(out_row, out_col, out_node, out_eid, out_nph, out_eph);
}

// This build only: frees the idle node tables the library keeps between sampler calls (current device) through the caching
// allocator they came from; returns the number still lent to running calls.
int64_t sampler_release_table_cache_kernel() {
  SamplerHost host;
  host.stream = nullptr;
  pyg_hip_sampler_host cb{&host, &host_alloc, &host_free, &host_rng_blocks, nullptr};
  const int rc = pyg_hip_sampler_table_cache_release(&cb);
  TORCH_CHECK(rc >= 0, pyg_hip_last_error());
  return rc;
}

// pyg::dist_neighbor_sample (sampler/cpu/neighbor_kernel.cpp:957-978)
std::tuple<Tensor, Tensor, std::vector<int64_t>> dist_neighbor_sample_kernel(
    const Tensor& rowptr, const Tensor& col, const Tensor& seed, const int64_t num_neighbors,
    const c10::optional<Tensor>& node_time, const c10::optional<Tensor>& edge_time,
    const c10::optional<Tensor>& seed_time, const c10::optional<Tensor>& edge_weight, bool csc, bool replace,
    bool directed, bool disjoint, std::string temporal_strategy) {
  PYG_TRACE("pyg::dist_neighbor_sample");
  check_modes(node_time.has_value(), edge_time.has_value(), seed_time.has_value(), edge_weight.has_value(), directed,
              disjoint, temporal_strategy);
  // dispatched on the seeds' integral type like the other samplers (neighbor_kernel.cpp:893,930): an int32 CSR is read in
  // place, only the seeds are widened for the call and the results narrowed back
  IndexArgs ix;
  ix.dtype = index_dtype(seed);
  const int64_t* rowptr_p = ix.graph(rowptr, "rowptr");
  const int64_t* col_p = ix.graph(col, "col");
  const int64_t* seed_p = ix.ptr(seed, "seed");
  DeviceGuard guard(rowptr.device());
  const auto opts = at::TensorOptions().dtype(at::kLong).device(rowptr.device());
  SamplerHost host;
  host.stream = current_hip_stream(rowptr.device().index());
  EngineLoan loan;
  pyg_hip_sampler_host cb{&host, &host_alloc, &host_free, &host_rng_blocks, loan.ptr()};
  const int64_t S = seed.numel();
  std::vector<int64_t> cumsum((size_t)S + 1, 0);
  int64_t* node_ptr = nullptr;
  int64_t* edge_ptr = nullptr;
  int64_t E = 0;
  pyg_hip_relation wrel{};  // only carries the weights (set_weight checks them)
  if (edge_weight.has_value()) set_weight(wrel, edge_weight.value(), col.numel());
  const int rc = pyg_hip_dist_neighbor_sample(
      rowptr_p, col_p, seed_p, S, num_neighbors,
      node_time.has_value() ? time_ptr(node_time.value(), "node_time") : nullptr,
      edge_time.has_value() ? time_ptr(edge_time.value(), "edge_time") : nullptr,
      seed_time.has_value() ? time_ptr(seed_time.value(), "seed_time") : nullptr, wrel.edge_weight,
      wrel.edge_weight_dtype, temporal_strategy == "last", replace, disjoint, ix.is32(), &cb, &node_ptr, &edge_ptr, &E,
      cumsum.data(), host.stream);
  if (rc == PYG_HIP_OK) loan.commit();
  TORCH_CHECK(host.error.empty(), host.error);
  check_status(rc);
  auto nodes = disjoint ? adopt(node_ptr, {S + E, 2}, opts) : adopt(node_ptr, {S + E}, opts);
  auto edges = adopt(edge_ptr, {E}, opts);
  return std::make_tuple(ix.narrow(nodes), ix.narrow(edges), cumsum);
}

// ---------------------------------------------------------------------------------------------
// registration
// ---------------------------------------------------------------------------------------------
TORCH_LIBRARY_FRAGMENT(pyg, m) {
  // pyg_lib/csrc/library.cpp:27-29
  m.def("cuda_version", &cuda_version);
  // pyg_lib/csrc/ops/matmul.cpp:63-68
  m.def(TORCH_SELECTIVE_SCHEMA("pyg::grouped_matmul(Tensor[] input, Tensor[] other) -> Tensor[]"));
  m.def(TORCH_SELECTIVE_SCHEMA("pyg::segment_matmul(Tensor input, Tensor ptr, Tensor other) -> Tensor"));
  // this build only: bias as a fused epilogue
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::segment_matmul_bias(Tensor input, Tensor ptr, Tensor other, Tensor bias) -> Tensor"));
  // this build only: the weight gradient of segment_matmul as its own operator
  m.def(TORCH_SELECTIVE_SCHEMA("pyg::segment_matmul_grad_other(Tensor input, Tensor ptr, Tensor grad_out) -> Tensor"));
  // this build only: fused R-GCN aggregation (gather -> per-relation matmul -> scatter-add), csrc/hip/rgcn.hip
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::rgcn_fused(Tensor x, Tensor[] gather_index, Tensor[] scatter_index, int[] gather_offset, "
      "int[] scatter_offset, Tensor weight, Tensor(a!) out, bool grouped = False, int[]? scatter_rows = None) -> Tensor(a!)"));
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::rgcn_fused_tables(Tensor[] feat, Tensor[] node_id, int[] gather_type, Tensor[] gather_index, "
      "Tensor[] scatter_index, int[] scatter_offset, Tensor weight, Tensor(a!) out, bool grouped = False, "
      "int[]? scatter_rows = None) -> Tensor(a!)"));
  // this build only: grouped_matmul writing into a caller-provided [sum rows, M] pool (sharded driver)
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::grouped_matmul_pool(Tensor[] input, Tensor[] other, Tensor(a!) pool) -> Tensor[]"));
  // pyg_lib/csrc/sampler/neighbor.cpp:129-147
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::neighbor_sample(Tensor rowptr, Tensor col, Tensor seed, int[] "
      "num_neighbors, Tensor? node_time = None, Tensor? edge_time = None, "
      "Tensor? seed_time = None, Tensor? edge_weight = None, bool csc = False, "
      "bool replace = False, bool directed = True, bool disjoint = False, "
      "str temporal_strategy = 'uniform', bool return_edge_id = True) -> "
      "(Tensor, Tensor, Tensor, Tensor?, int[], int[])"));
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::hetero_neighbor_sample(str[] node_types, (str, str, str)[] "
      "edge_types, Dict(str, Tensor) rowptr_dict, Dict(str, Tensor) col_dict, "
      "Dict(str, Tensor) seed_dict, Dict(str, int[]) num_neighbors_dict, "
      "Dict(str, Tensor)? node_time_dict = None, Dict(str, Tensor)? "
      "edge_time_dict = None, Dict(str, Tensor)? seed_time_dict = None, "
      "Dict(str, Tensor)? edge_weight_dict = None, bool csc = False, "
      "bool replace = False, bool directed = True, bool disjoint = False, "
      "str temporal_strategy = 'uniform', bool return_edge_id = True) -> "
      "(Dict(str, Tensor), Dict(str, Tensor), Dict(str, Tensor), "
      "Dict(str, Tensor)?, Dict(str, int[]), Dict(str, int[]))"));
  m.def("sampler_release_table_cache() -> int", &sampler_release_table_cache_kernel);
  // this build only: K independent mini-batches in one call, overlapped on the device (pyg_hip_hetero_neighbor_sample_batched)
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::neighbor_sample_batched(Tensor rowptr, Tensor col, Tensor[] seeds, int[] num_neighbors, int[] generator_seeds, "
      "Tensor? node_time = None, Tensor? edge_time = None, Tensor[]? seed_times = None, Tensor? edge_weight = None, "
      "bool csc = False, bool replace = False, bool directed = True, bool disjoint = False, "
      "str temporal_strategy = 'uniform', bool return_edge_id = True) -> "
      "(Tensor[], Tensor[], Tensor[], Tensor[], Tensor, Tensor)"));
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::hetero_neighbor_sample_batched(str[] node_types, (str, str, str)[] edge_types, Dict(str, Tensor) rowptr_dict, "
      "Dict(str, Tensor) col_dict, Dict(str, Tensor)[] seed_dicts, Dict(str, int[]) num_neighbors_dict, "
      "int[] generator_seeds, Dict(str, Tensor)? node_time_dict = None, Dict(str, Tensor)? edge_time_dict = None, "
      "Dict(str, Tensor)[]? seed_time_dicts = None, Dict(str, Tensor)? edge_weight_dict = None, bool csc = False, "
      "bool replace = False, bool directed = True, bool disjoint = False, str temporal_strategy = 'uniform', "
      "bool return_edge_id = True) -> "
      "(Dict(str, Tensor)[], Dict(str, Tensor)[], Dict(str, Tensor)[], Dict(str, Tensor)[], Dict(str, int[])[], "
      "Dict(str, int[])[])"));
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::dist_neighbor_sample(Tensor rowptr, Tensor col, Tensor seed, int "
      "num_neighbors, Tensor? node_time = None, Tensor? edge_time = None, "
      "Tensor? seed_time = None, Tensor? edge_weight = None, bool csc = False, "
      "bool replace = False, bool directed = True, bool disjoint = False, "
      "str temporal_strategy = 'uniform') -> (Tensor, Tensor, int[])"));
}

// HIP tensors dispatch under the CUDA key on PyTorch-ROCm.
TORCH_LIBRARY_IMPL(pyg, CUDA, m) {
  m.impl(TORCH_SELECTIVE_NAME("pyg::grouped_matmul"), TORCH_FN(grouped_matmul_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_matmul"), TORCH_FN(segment_matmul_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_matmul_bias"), TORCH_FN(segment_matmul_bias_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_matmul_grad_other"), TORCH_FN(segment_matmul_grad_other_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::grouped_matmul_pool"), TORCH_FN(grouped_matmul_pool_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::rgcn_fused_tables"), TORCH_FN(rgcn_fused_tables_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::rgcn_fused"), TORCH_FN(rgcn_fused_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::neighbor_sample"), TORCH_FN(neighbor_sample_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::neighbor_sample_batched"), TORCH_FN(neighbor_sample_batched_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::dist_neighbor_sample"), TORCH_FN(dist_neighbor_sample_kernel));
}

// pyg_lib/csrc/ops/autograd/matmul_kernel.cpp:121-124
TORCH_LIBRARY_IMPL(pyg, Autograd, m) {
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_matmul"), TORCH_FN(segment_matmul_autograd));
}

// Tensors inside Dicts cannot drive dispatch (sampler/cpu/neighbor_kernel.cpp:985-991): the
// kernel checks the device itself and refuses CPU graphs.
TORCH_LIBRARY_IMPL(pyg, BackendSelect, m) {
  m.impl(TORCH_SELECTIVE_NAME("pyg::hetero_neighbor_sample"), TORCH_FN(hetero_neighbor_sample_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::hetero_neighbor_sample_batched"), TORCH_FN(hetero_neighbor_sample_batched_kernel));
}

}  // namespace pyg_amd

// Test / measurement hook of libpyg.so (not an operator): tile schedule of the matmul calls made from the calling
// thread (PYG_HIP_MM_SCHED_* of include/pyg_hip.h), returned to automatic by 0.  Thread-local by design.
extern "C" __attribute__((visibility("default"))) void pyg_binding_set_matmul_schedule(int mode) {
  pyg_amd::matmul_schedule_tls() = (mode >= 0 && mode <= PYG_HIP_MM_SCHED_RING) ? mode : PYG_HIP_MM_SCHED_AUTO;
}
extern "C" __attribute__((visibility("default"))) int pyg_binding_get_matmul_schedule(void) {
  return pyg_amd::matmul_schedule_tls();
}
