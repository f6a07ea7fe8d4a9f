// torch::Library binding of the distributed-sampling helpers pyg::relabel_neighborhood and
// pyg::merge_sampler_outputs (homogeneous forms).  Schemas byte-identical to
// pyg_lib/csrc/sampler/dist_relabel.cpp:71-76 and sampler/dist_merge_outputs.cpp:51-55; argument checks follow
// sampler/cpu/dist_relabel_kernel.cpp:37-48.  Kernels: csrc/hip/sampler.hip through the C-ABI.
#include <torch/library.h>

#include "binding_common.h"

namespace pyg_amd {
namespace {

std::tuple<Tensor, Tensor> relabel_neighborhood_kernel(const Tensor& seed, const Tensor& sampled_nodes_with_duplicates,
                                                       const std::vector<int64_t>& num_sampled_neighbors_per_node,
                                                       const int64_t num_nodes, const std::optional<Tensor>& batch,
                                                       bool csc, bool disjoint) {
  (void)num_nodes;  // only a capacity hint in the reference (Mapper)
  TORCH_CHECK(seed.is_cuda() && sampled_nodes_with_duplicates.is_cuda(), "relabel_neighborhood: tensors must live on a HIP device");
  TORCH_CHECK(seed.scalar_type() == at::kLong && sampled_nodes_with_duplicates.scalar_type() == at::kLong,
              "relabel_neighborhood: int64 node ids expected on the device path");
  if (disjoint) {
    TORCH_CHECK(batch.has_value(), "Batch needs to be specified to create disjoint subgraphs");
    TORCH_CHECK(batch.value().is_contiguous(), "Non-contiguous 'batch'");
    TORCH_CHECK(batch.value().numel() == sampled_nodes_with_duplicates.numel(), "Each node must belong to a subgraph");
    TORCH_CHECK(batch.value().is_cuda() && batch.value().scalar_type() == at::kLong, "relabel_neighborhood: int64 device 'batch' expected");
  }
  TORCH_CHECK(seed.is_contiguous(), "Non-contiguous 'seed'");
  TORCH_CHECK(sampled_nodes_with_duplicates.is_contiguous(), "Non-contiguous 'sampled_nodes_with_duplicates'");
  DeviceGuard guard(seed.device());
  const int64_t n = (int64_t)num_sampled_neighbors_per_node.size();
  auto prefix_cpu = at::empty({n + 1}, at::TensorOptions().dtype(at::kLong));
  int64_t* pp = prefix_cpu.data_ptr<int64_t>();
  pp[0] = 0;
  for (int64_t i = 0; i < n; ++i) pp[i + 1] = pp[i] + num_sampled_neighbors_per_node[(size_t)i];
  const int64_t E = pp[n];
  TORCH_CHECK(E <= sampled_nodes_with_duplicates.numel(), "relabel_neighborhood: more sampled neighbours announced than nodes given");
  auto prefix = prefix_cpu.to(seed.device());
  auto row = at::empty({E}, seed.options());
  auto col = at::empty({E}, seed.options());
  auto ws = at::empty({(int64_t)pyg_hip_relabel_workspace_size(seed.numel(), E)}, seed.options().dtype(at::kByte));
  check_status(pyg_hip_relabel_neighborhood(seed.data_ptr<int64_t>(), seed.numel(),
                                            sampled_nodes_with_duplicates.data_ptr<int64_t>(), E, prefix.data_ptr<int64_t>(), n,
                                            disjoint ? batch.value().data_ptr<int64_t>() : nullptr, disjoint ? 1 : 0,
                                            row.data_ptr<int64_t>(), col.data_ptr<int64_t>(), ws.data_ptr(), (size_t)ws.numel(),
                                            current_stream(seed)));
  if (!csc) return std::make_tuple(row, col);
  return std::make_tuple(col, row);
}

std::tuple<Tensor, Tensor, std::optional<Tensor>, std::vector<int64_t>> merge_sampler_outputs_kernel(
    const std::vector<Tensor>& node_ids, const std::vector<Tensor>& edge_ids,
    const std::vector<std::vector<int64_t>>& cumsum_neighbors_per_node, const std::vector<int64_t>& partition_ids,
    const std::vector<int64_t>& partition_orders, const int64_t num_partitions, const int64_t num_neighbors,
    const std::optional<Tensor>& batch, bool disjoint) {
  (void)num_neighbors;  // the reference only uses it to size a padded staging buffer
  TORCH_CHECK(num_partitions > 0 && (int64_t)node_ids.size() >= num_partitions && (int64_t)edge_ids.size() >= num_partitions &&
                  (int64_t)cumsum_neighbors_per_node.size() >= num_partitions,
              "merge_sampler_outputs: one entry per partition expected");
  TORCH_CHECK(partition_ids.size() == partition_orders.size(), "merge_sampler_outputs: partition_ids / partition_orders differ in length");
  if (disjoint) TORCH_CHECK(batch.has_value(), "merge_sampler_outputs: 'batch' needed for disjoint sampling");
  std::vector<Tensor> nodes_c, edges_c;
  for (int64_t p = 0; p < num_partitions; ++p) {
    TORCH_CHECK(node_ids[(size_t)p].is_cuda() && edge_ids[(size_t)p].is_cuda(), "merge_sampler_outputs: tensors must live on a HIP device");
    TORCH_CHECK(node_ids[(size_t)p].scalar_type() == at::kLong && edge_ids[(size_t)p].scalar_type() == at::kLong,
                "merge_sampler_outputs: int64 ids expected on the device path");
    nodes_c.push_back(node_ids[(size_t)p].contiguous());
    edges_c.push_back(edge_ids[(size_t)p].contiguous());
  }
  const auto dev = nodes_c[0].device();
  DeviceGuard guard(dev);
  const int64_t P = num_partitions, n = (int64_t)partition_ids.size();
  // one host block: node bases [P], edge bases [P], part [n], begin_node [n], begin_edge [n], dst_off [n + 1]
  auto meta_cpu = at::empty({2 * P + 3 * n + (n + 1)}, at::TensorOptions().dtype(at::kLong));
  int64_t* m = meta_cpu.data_ptr<int64_t>();
  for (int64_t p = 0; p < P; ++p) {
    m[p] = (int64_t)(uintptr_t)nodes_c[(size_t)p].data_ptr<int64_t>();
    m[P + p] = (int64_t)(uintptr_t)edges_c[(size_t)p].data_ptr<int64_t>();
  }
  int64_t* part = m + 2 * P;
  int64_t* begin_n = part + n;
  int64_t* begin_e = begin_n + n;
  int64_t* dst_off = begin_e + n;
  std::vector<int64_t> counts((size_t)n);
  dst_off[0] = 0;
  for (int64_t j = 0; j < n; ++j) {
    const int64_t p = partition_ids[(size_t)j], o = partition_orders[(size_t)j];
    TORCH_CHECK(p >= 0 && p < P, "merge_sampler_outputs: partition id out of range");
    const auto& cs = cumsum_neighbors_per_node[(size_t)p];
    TORCH_CHECK(o >= 0 && o + 1 < (int64_t)cs.size(), "merge_sampler_outputs: partition order out of range");
    const int64_t bn = cs[(size_t)o], en = cs[(size_t)o + 1];
    TORCH_CHECK(bn >= 0 && en >= bn && en <= nodes_c[(size_t)p].numel() && en - cs[0] <= edges_c[(size_t)p].numel(),
                "merge_sampler_outputs: cumulative sums do not fit the partition's tensors");
    part[j] = p;
    begin_n[j] = bn;
    begin_e[j] = bn - cs[0];
    counts[(size_t)j] = en - bn;
    dst_off[j + 1] = dst_off[j] + (en - bn);
  }
  const int64_t total = dst_off[n];
  auto meta = meta_cpu.to(dev);
  const int64_t* md = meta.data_ptr<int64_t>();
  auto out_node = at::empty({total}, nodes_c[0].options());
  auto out_edge = at::empty({total}, nodes_c[0].options());
  void* stream = current_stream(nodes_c[0]);
  check_status(pyg_hip_segment_concat(reinterpret_cast<const int64_t* const*>(md), md + 2 * P, md + 2 * P + n, md + 2 * P + 3 * n, n,
                                      nullptr, out_node.data_ptr<int64_t>(), total, stream));
  check_status(pyg_hip_segment_concat(reinterpret_cast<const int64_t* const*>(md + P), md + 2 * P, md + 2 * P + 2 * n,
                                      md + 2 * P + 3 * n, n, nullptr, out_edge.data_ptr<int64_t>(), total, stream));
  std::optional<Tensor> out_batch;
  if (disjoint) {
    auto b = batch.value().to(dev).to(at::kLong).contiguous();
    TORCH_CHECK(b.numel() >= n, "merge_sampler_outputs: one batch id per sampled-from node expected");
    auto ob = at::empty({total}, nodes_c[0].options());
    check_status(pyg_hip_segment_concat(nullptr, nullptr, nullptr, md + 2 * P + 3 * n, n, b.data_ptr<int64_t>(),
                                        ob.data_ptr<int64_t>(), total, stream));
    out_batch = ob;
  }
  return std::make_tuple(out_node, out_edge, out_batch, counts);
}

}  // namespace

// sampler/dist_relabel.cpp:71-76, sampler/dist_merge_outputs.cpp:51-55
TORCH_LIBRARY_FRAGMENT(pyg, m) {
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::relabel_neighborhood(Tensor seed, Tensor sampled_nodes_with_duplicates, int[] "
      "num_sampled_neighbors_per_node, int num_nodes, Tensor? batch = None, bool csc = False, bool disjoint = "
      "False) -> (Tensor, Tensor)"));
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::merge_sampler_outputs(Tensor[] node_ids, Tensor[] edge_ids, int[][] cumsum_neighbors_per_node, int[] "
      "partition_ids, int[] partition_orders, int num_partitions, int num_neighbors, Tensor? batch, bool disjoint) -> "
      "(Tensor, Tensor, Tensor?, int[])"));
}

TORCH_LIBRARY_IMPL(pyg, CUDA, m) {
  m.impl(TORCH_SELECTIVE_NAME("pyg::relabel_neighborhood"), TORCH_FN(relabel_neighborhood_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::merge_sampler_outputs"), TORCH_FN(merge_sampler_outputs_kernel));
}

}  // namespace pyg_amd
