// torch::Library binding of the distributed-sampling helpers pyg::relabel_neighborhood,
// pyg::hetero_relabel_neighborhood and pyg::merge_sampler_outputs.  Schemas byte-identical to
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
  PYG_TRACE("pyg::relabel_neighborhood");
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
  PYG_TRACE("pyg::merge_sampler_outputs");
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
typedef std::string node_type;
typedef std::string rel_type;
typedef std::tuple<std::string, std::string, std::string> edge_type;
inline rel_type rel_of(const edge_type& k) { return std::get<0>(k) + "__" + std::get<1>(k) + "__" + std::get<2>(k); }

// pyg::hetero_relabel_neighborhood (sampler/cpu/dist_relabel_kernel.cpp:96-262).  Every node type has one
// Mapper and its sampled list is consumed strictly in order, so the local ids are computed per node type
// (pyg_hip_relabel_nodes) and every (edge type, layer) is a segment: rows = the layer's source range expanded
// by the per-source counts, cols = the next slice of the destination type's local ids.
std::tuple<c10::Dict<rel_type, Tensor>, c10::Dict<rel_type, Tensor>> hetero_relabel_neighborhood_kernel(
    const std::vector<node_type>& node_types, const std::vector<edge_type>& edge_types,
    const c10::Dict<node_type, Tensor>& seed_dict, const c10::Dict<node_type, Tensor>& sampled_nodes_with_duplicates_dict,
    const c10::Dict<rel_type, std::vector<std::vector<int64_t>>>& num_sampled_neighbors_per_node_dict,
    const c10::Dict<node_type, int64_t>& num_nodes_dict, const std::optional<c10::Dict<node_type, Tensor>>& batch_dict,
    bool csc, bool disjoint) {
  PYG_TRACE("pyg::hetero_relabel_neighborhood");
  (void)num_nodes_dict;  // capacity hints of the reference's Mappers
  TORCH_CHECK(!edge_types.empty(), "hetero_relabel_neighborhood: no edge types");
  if (disjoint) TORCH_CHECK(batch_dict.has_value(), "Batch needs to be specified to create disjoint subgraphs");
  std::optional<at::Device> device;
  int64_t total_seeds = 0;
  for (const auto& kv : seed_dict) total_seeds += kv.value().numel();
  // per node type: local ids of the whole sampled list
  std::unordered_map<std::string, Tensor> local;
  std::unordered_map<std::string, int64_t> seed_batch0;
  {
    int64_t b = 0;
    for (const auto& kv : seed_dict) {  // batch ids run on across the seed types (:181-192)
      seed_batch0[kv.key()] = b;
      b += kv.value().numel();
    }
  }
  for (const auto& t : node_types) {
    TORCH_CHECK(sampled_nodes_with_duplicates_dict.contains(t), "hetero_relabel_neighborhood: no sampled nodes for type '", t, "'");
    const Tensor& nodes = sampled_nodes_with_duplicates_dict.at(t);
    TORCH_CHECK(nodes.is_cuda(), "hetero_relabel_neighborhood: tensors must live on a HIP device");
    TORCH_CHECK(nodes.scalar_type() == at::kLong, "hetero_relabel_neighborhood: int64 node ids expected on the device path");
    TORCH_CHECK(nodes.is_contiguous(), "Non-contiguous 'sampled_nodes_with_duplicates'");
    if (!device.has_value()) device = nodes.device();
    DeviceGuard guard(nodes.device());
    const int64_t E = nodes.numel();
    Tensor seed;
    if (seed_dict.contains(t)) {
      seed = seed_dict.at(t);
      TORCH_CHECK(seed.is_cuda() && seed.scalar_type() == at::kLong && seed.is_contiguous(),
                  "hetero_relabel_neighborhood: contiguous int64 device 'seed' expected");
    }
    const int64_t S = seed.defined() ? seed.numel() : 0;
    const int64_t* batch = nullptr;
    if (disjoint) {
      TORCH_CHECK(batch_dict.value().contains(t), "hetero_relabel_neighborhood: no batch vector for type '", t, "'");
      const Tensor& bt = batch_dict.value().at(t);
      TORCH_CHECK(bt.is_cuda() && bt.scalar_type() == at::kLong && bt.is_contiguous() && bt.numel() == E,
                  "hetero_relabel_neighborhood: 'batch' must be a contiguous int64 device tensor, one entry per node");
      batch = bt.data_ptr<int64_t>();
    }
    auto out = at::empty({E}, nodes.options());
    if (E > 0) {
      auto ws = at::empty({(int64_t)pyg_hip_relabel_workspace_size(S, E)}, nodes.options().dtype(at::kByte));
      check_status(pyg_hip_relabel_nodes(S ? seed.data_ptr<int64_t>() : nullptr, S, S ? seed_batch0[t] : 0,
                                         std::max<int64_t>(total_seeds, 1), nodes.data_ptr<int64_t>(), E, batch,
                                         disjoint ? 1 : 0, out.data_ptr<int64_t>(), ws.data_ptr(), (size_t)ws.numel(),
                                         current_stream(nodes)));
    }
    local[t] = out;
  }
  TORCH_CHECK(device.has_value(), "hetero_relabel_neighborhood: no node types");
  DeviceGuard guard(device.value());
  const auto opts = at::TensorOptions().dtype(at::kLong).device(device.value());
  const size_t num_layers = num_sampled_neighbors_per_node_dict.at(rel_of(edge_types[0])).size();
  // host walk of the (layer, edge type) segments (:194-256)
  std::unordered_map<std::string, int64_t> dst_pos, src_off;
  for (const auto& t : node_types) {
    dst_pos[t] = 0;
    src_off[t] = 0;
  }
  struct Seg {
    int64_t src_begin, dst_begin, edges;
    const std::vector<int64_t>* counts;
  };
  std::vector<std::vector<Seg>> segs(edge_types.size());
  std::vector<std::pair<int64_t, int64_t>> slice(edge_types.size());
  for (size_t e = 0; e < edge_types.size(); ++e)
    slice[e] = {0, (int64_t)num_sampled_neighbors_per_node_dict.at(rel_of(edge_types[e]))[0].size()};
  // the Dict hands out copies: keep them alive while Seg points into them
  std::vector<std::vector<std::vector<int64_t>>> counts_keep(edge_types.size());
  for (size_t e = 0; e < edge_types.size(); ++e) counts_keep[e] = num_sampled_neighbors_per_node_dict.at(rel_of(edge_types[e]));
  for (size_t ell = 0; ell < num_layers; ++ell) {
    for (size_t e = 0; e < edge_types.size(); ++e) {
      const auto& k = edge_types[e];
      const std::string& dst = !csc ? std::get<2>(k) : std::get<0>(k);
      TORCH_CHECK(counts_keep[e].size() > ell, "hetero_relabel_neighborhood: edge types list different numbers of layers");
      const auto& cnt = counts_keep[e][ell];
      TORCH_CHECK((int64_t)cnt.size() == slice[e].second - slice[e].first, "hetero_relabel_neighborhood: layer size mismatch");
      int64_t n = 0;
      for (int64_t c : cnt) n += c;
      TORCH_CHECK(dst_pos.count(dst), "hetero_relabel_neighborhood: edge type names an unknown node type");
      TORCH_CHECK(dst_pos[dst] + n <= local[dst].numel(), "hetero_relabel_neighborhood: more sampled neighbours announced than nodes given");
      segs[e].push_back({slice[e].first, dst_pos[dst], n, &cnt});
      dst_pos[dst] += n;
    }
    if (ell + 1 < num_layers) {
      for (size_t e = 0; e < edge_types.size(); ++e) {
        const std::string& src = !csc ? std::get<0>(edge_types[e]) : std::get<2>(edge_types[e]);
        src_off[src] = std::max(src_off[src], slice[e].second);
      }
      for (size_t e = 0; e < edge_types.size(); ++e) {
        const std::string& src = !csc ? std::get<0>(edge_types[e]) : std::get<2>(edge_types[e]);
        slice[e] = {src_off[src], src_off[src] + (int64_t)counts_keep[e][ell + 1].size()};
      }
    }
  }
  c10::Dict<rel_type, Tensor> out_row, out_col;
  for (size_t e = 0; e < edge_types.size(); ++e) {
    const auto& k = edge_types[e];
    const std::string& dst = !csc ? std::get<2>(k) : std::get<0>(k);
    int64_t E = 0, nsrc = 0;
    for (const Seg& sg : segs[e]) {
      E += sg.edges;
      nsrc += (int64_t)sg.counts->size();
    }
    // one prefix over all layers' sources: the expanded index is the position in that list; its source id is
    // src_begin(layer) + position inside the layer -> a second small table maps list position to source id
    auto meta_cpu = at::empty({2 * nsrc + 1}, at::TensorOptions().dtype(at::kLong));
    int64_t* prefix = meta_cpu.data_ptr<int64_t>();
    int64_t* src_id = prefix + nsrc + 1;
    prefix[0] = 0;
    int64_t q = 0;
    for (const Seg& sg : segs[e])
      for (size_t i = 0; i < sg.counts->size(); ++i, ++q) {
        prefix[q + 1] = prefix[q] + (*sg.counts)[i];
        src_id[q] = sg.src_begin + (int64_t)i;
      }
    auto row = at::empty({E}, opts);
    std::vector<Tensor> parts;
    for (const Seg& sg : segs[e]) parts.push_back(local[dst].narrow(0, sg.dst_begin, sg.edges));
    Tensor col = parts.empty() ? at::empty({0}, opts) : at::cat(parts);
    if (E > 0) {
      auto meta = meta_cpu.to(device.value());
      check_status(pyg_hip_expand_rows(meta.data_ptr<int64_t>(), nsrc, E, row.data_ptr<int64_t>(), current_stream(row)));
      row = meta.narrow(0, nsrc + 1, nsrc).index_select(0, row);  // list position -> source id
    }
    if (!csc) {
      out_row.insert(rel_of(k), row);
      out_col.insert(rel_of(k), col);
    } else {
      out_row.insert(rel_of(k), col);
      out_col.insert(rel_of(k), row);
    }
  }
  return std::make_tuple(out_row, out_col);
}

TORCH_LIBRARY_FRAGMENT(pyg, m) {
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::relabel_neighborhood(Tensor seed, Tensor sampled_nodes_with_duplicates, int[] "
      "num_sampled_neighbors_per_node, int num_nodes, Tensor? batch = None, bool csc = False, bool disjoint = "
      "False) -> (Tensor, Tensor)"));
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::merge_sampler_outputs(Tensor[] node_ids, Tensor[] edge_ids, int[][] cumsum_neighbors_per_node, int[] "
      "partition_ids, int[] partition_orders, int num_partitions, int num_neighbors, Tensor? batch, bool disjoint) -> "
      "(Tensor, Tensor, Tensor?, int[])"));
  m.def(TORCH_SELECTIVE_SCHEMA(
      "pyg::hetero_relabel_neighborhood(str[] node_types, (str, str, str)[] edge_types, Dict(str, Tensor) seed_dict, "
      "Dict(str, Tensor) sampled_nodes_with_duplicates_dict, Dict(str, int[][]) num_sampled_neighbors_per_node_dict, "
      "Dict(str, int) num_nodes_dict, Dict(str, Tensor)? batch_dict = None, bool csc = False, bool disjoint = False) -> "
      "(Dict(str, Tensor), Dict(str, Tensor))"));
}

TORCH_LIBRARY_IMPL(pyg, CUDA, m) {
  m.impl(TORCH_SELECTIVE_NAME("pyg::relabel_neighborhood"), TORCH_FN(relabel_neighborhood_kernel));
  m.impl(TORCH_SELECTIVE_NAME("pyg::merge_sampler_outputs"), TORCH_FN(merge_sampler_outputs_kernel));
}

// Tensors inside Dicts cannot drive dispatch (the reference registers this op under BackendSelect too,
// sampler/cpu/dist_relabel_kernel.cpp:315-318): the kernel checks the device itself.
TORCH_LIBRARY_IMPL(pyg, BackendSelect, m) {
  m.impl(TORCH_SELECTIVE_NAME("pyg::hetero_relabel_neighborhood"), TORCH_FN(hetero_relabel_neighborhood_kernel));
}

}  // namespace pyg_amd
