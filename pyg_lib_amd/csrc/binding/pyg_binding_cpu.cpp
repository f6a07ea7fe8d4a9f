// CPU dispatch key of the hot-path operators (a replacement registers `CPU` and `CUDA`).
//
// PyG's loaders hand the sampler CPU tensors, and `segment_matmul` / `grouped_matmul` / `index_sort` are called on
// CPU tensors by CPU-only users: with this translation unit the reference's Python package runs unchanged on either
// device.  These are real CPU implementations written for this build -- not a fallback of the HIP path (device
// tensors never come here; the dispatcher picks by device) and not the test oracle (nothing under oracle/ is
// linked, included or called).  They follow the reference's CPU kernels so that results agree with them bit for
// bit where the reference is deterministic:
//   * neighbor_sample / hetero_neighbor_sample: sampler/cpu/neighbor_kernel.cpp:332-841 -- one prefetching
//     random-integer engine per call on the global CPU generator (random/cpu/rand_engine.h:41-92), Floyd-style
//     sampling without replacement (:231-240), insertion-ordered relabelling (sampler/cpu/mapper.h:12-78),
//     temporal narrowing (:74-144), biased sampling through the same ATen calls (:245-285);
//   * segment_matmul / grouped_matmul: one at::matmul per segment / group (ops/cpu/matmul_kernel.cpp:195-201,
//     281-312, 410-439);
//   * index_sort: stable ascending sort + int64 permutation (ops/cpu/index_sort_kernel.cpp:14-59; its radix path
//     and at::sort agree on every valid -- non-negative -- input).
#include <ATen/ATen.h>
#include <torch/library.h>

#include <cstdint>
#include <limits>
#include <string>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "binding_common.h"

namespace pyg_amd {
namespace cpu {

typedef std::string node_type;
typedef std::string rel_type;
typedef std::tuple<std::string, std::string, std::string> edge_type;

inline rel_type rel_name(const edge_type& k) { return std::get<0>(k) + "__" + std::get<1>(k) + "__" + std::get<2>(k); }

// ---------------------------------------------------------------------------------------------------------------
// random integers: 128 prefetched 64-bit words, consumed 16 / 32 / 64 bits at a time from the last word down
// ---------------------------------------------------------------------------------------------------------------
class WordEngine {
 public:
  WordEngine() {
    buf_ = at::randint(std::numeric_limits<int64_t>::min(), std::numeric_limits<int64_t>::max(), {kWords}, at::kLong);
    words_ = buf_.data_ptr<int64_t>();
  }
  // uniform in [0, range)
  uint64_t below(uint64_t range) {
    const int need = range < (1ull << 16) ? 16 : (range < (1ull << 32) ? 32 : 64);
    if (bits_ < need) {
      if (pos_ > 0) {
        --pos_;
      } else {
        buf_.random_(std::numeric_limits<int64_t>::min(), std::numeric_limits<int64_t>::max());
        pos_ = kWords - 1;
      }
      bits_ = 64;  // whatever was left of the previous word is dropped
    }
    uint64_t w = static_cast<uint64_t>(words_[pos_]);
    const uint64_t mask = need == 64 ? ~0ull : ((1ull << need) - 1);
    const uint64_t r = (w & mask) % range;
    w = need == 64 ? 0 : (w >> need);
    words_[pos_] = static_cast<int64_t>(w);
    bits_ -= need;
    return r;
  }

 private:
  static constexpr int kWords = 128;
  Tensor buf_;
  int64_t* words_;
  int pos_ = kWords - 1;
  int bits_ = 64;
};

// ---------------------------------------------------------------------------------------------------------------
// global -> local ids in order of first appearance
// ---------------------------------------------------------------------------------------------------------------
class IdMap {
 public:
  void init(int64_t num_nodes, int64_t expected, bool pairs, int64_t num_batches) {
    num_batches_ = pairs ? num_batches : 1;
    dense_ = !pairs && num_nodes > 0 && (num_nodes < 1000000 || expected > num_nodes / 10);
    if (dense_) table_.assign(static_cast<size_t>(num_nodes), -1);
  }
  // (local id, first time?)
  std::pair<int64_t, bool> insert(int64_t node, int64_t batch) {
    if (dense_) {
      TORCH_CHECK(node >= 0 && node < static_cast<int64_t>(table_.size()), "neighbor_sample: node id ", node,
                  " out of range");
      int64_t& slot = table_[static_cast<size_t>(node)];
      if (slot >= 0) return {slot, false};
      slot = next_;
      return {next_++, true};
    }
    const uint64_t key = static_cast<uint64_t>(node) * static_cast<uint64_t>(num_batches_) + static_cast<uint64_t>(batch);
    auto it = sparse_.find(key);
    if (it != sparse_.end()) return {it->second, false};
    sparse_.emplace(key, next_);
    return {next_++, true};
  }

 private:
  bool dense_ = false;
  int64_t num_batches_ = 1;
  int64_t next_ = 0;
  std::vector<int64_t> table_;
  std::unordered_map<uint64_t, int64_t> sparse_;
};

// ---------------------------------------------------------------------------------------------------------------
// the sampler
// ---------------------------------------------------------------------------------------------------------------
struct Relation {
  const int64_t* rowptr = nullptr;
  int64_t num_rows = 0;
  const int64_t* col = nullptr;
  int64_t num_cols = 0;
  int src = 0, dst = 0;  // node types as named by the edge type
  std::vector<int64_t> fanout;
  const int64_t* edge_time = nullptr;
  Tensor weight;  // undefined: uniform sampling
};

struct SeedSet {
  int type = 0;
  const int64_t* seed = nullptr;
  int64_t count = 0;
  const int64_t* seed_time = nullptr;
};

struct NodeList {
  std::vector<int64_t> nodes, batch, per_hop;
  IdMap ids;
  int64_t frontier_begin = 0, frontier_end = 0;
};

struct EdgeList {
  std::vector<int64_t> rows, cols, eids, per_hop;
};

struct Options {
  bool csc = false, replace = false, disjoint = false, last = false;
  int hops = 0;
};

class Sampler {
 public:
  Sampler(std::vector<Relation> rels, int num_types, std::vector<const int64_t*> node_time, Options opt)
      : rels_(std::move(rels)), node_time_(std::move(node_time)), opt_(opt), types_(static_cast<size_t>(num_types)),
        out_(rels_.size()) {}

  void run(const std::vector<SeedSet>& seeds) {
    int64_t num_batches = 1;
    if (opt_.disjoint) {
      num_batches = 0;
      for (const auto& s : seeds) num_batches += s.count;
      num_batches = std::max<int64_t>(num_batches, 1);
    }
    bool temporal = false;
    for (const auto* t : node_time_) temporal = temporal || t != nullptr;
    for (const auto& r : rels_) temporal = temporal || r.edge_time != nullptr;
    if (temporal) seed_times_.assign(static_cast<size_t>(num_batches), 0);
    // number of nodes per type where a relation tells (rows of the CSR it expands), for the dense id table
    std::vector<int64_t> num_nodes(types_.size(), 0);
    for (const auto& r : rels_) {
      const int from = opt_.csc ? r.dst : r.src;
      num_nodes[static_cast<size_t>(from)] = std::max(num_nodes[static_cast<size_t>(from)], r.num_rows);
    }
    int64_t expected = 0;
    for (const auto& s : seeds) expected += s.count;
    for (size_t t = 0; t < types_.size(); ++t) types_[t].ids.init(num_nodes[t], expected * 16, opt_.disjoint, num_batches);

    int64_t batch0 = 0;
    for (const auto& s : seeds) {
      NodeList& n = types_[static_cast<size_t>(s.type)];
      for (int64_t i = 0; i < s.count; ++i) {
        const int64_t b = opt_.disjoint ? batch0 + i : 0;
        n.ids.insert(s.seed[i], b);
        n.nodes.push_back(s.seed[i]);
        if (opt_.disjoint) n.batch.push_back(b);
        if (temporal) {
          const int64_t* nt = node_time_.empty() ? nullptr : node_time_[static_cast<size_t>(s.type)];
          TORCH_CHECK(s.seed_time || nt, "Seed time needs to be specified");
          if (opt_.disjoint) seed_times_[static_cast<size_t>(b)] = s.seed_time ? s.seed_time[i] : nt[s.seed[i]];
        }
      }
      batch0 += s.count;
    }
    for (auto& n : types_) {
      n.per_hop.push_back(static_cast<int64_t>(n.nodes.size()));
      n.frontier_begin = 0;
      n.frontier_end = static_cast<int64_t>(n.nodes.size());
    }
    WordEngine engine;  // one per call, shared by all hops and relations
    for (int hop = 0; hop < opt_.hops; ++hop) {
      std::vector<int64_t> before(types_.size());
      for (size_t t = 0; t < types_.size(); ++t) before[t] = static_cast<int64_t>(types_[t].nodes.size());
      for (size_t e = 0; e < rels_.size(); ++e) expand(e, hop, engine);
      for (size_t t = 0; t < types_.size(); ++t) {
        NodeList& n = types_[t];
        n.per_hop.push_back(static_cast<int64_t>(n.nodes.size()) - before[t]);
        n.frontier_begin = n.frontier_end;
        n.frontier_end = static_cast<int64_t>(n.nodes.size());
      }
    }
  }

  const NodeList& nodes(int t) const { return types_[static_cast<size_t>(t)]; }
  const EdgeList& edges(size_t e) const { return out_[e]; }

 private:
  // neighbourhood [rs, re) of `v` that the temporal constraint leaves
  void range_of(const Relation& r, const int64_t* time, bool edge_level, int64_t v, int64_t batch, int64_t count,
                int64_t* rs_out, int64_t* re_out) const {
    TORCH_CHECK(v >= 0 && v < r.num_rows, "neighbor_sample: node id ", v, " has no row in 'rowptr'");
    int64_t rs = r.rowptr[v], re = r.rowptr[v + 1];
    if (time && re > rs && count != 0) {
      const int64_t limit = seed_times_[static_cast<size_t>(batch)];
      auto at_pos = [&](int64_t p) { return edge_level ? time[p] : time[r.col[p]]; };
      int64_t lo = rs, hi = re;  // first position whose time exceeds the seed's
      while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (limit < at_pos(mid)) hi = mid; else lo = mid + 1;
      }
      re = lo;
      if (opt_.last && count >= 0 && re - count > rs) rs = re - count;
      if (re - rs > 1) TORCH_CHECK(at_pos(rs) <= at_pos(re - 1), "Found invalid non-sorted temporal neighborhood");
    }
    *rs_out = rs;
    *re_out = re;
  }

  void expand(size_t e, int hop, WordEngine& engine) {
    const Relation& r = rels_[e];
    const int from = opt_.csc ? r.dst : r.src;
    const int to = opt_.csc ? r.src : r.dst;
    NodeList& src = types_[static_cast<size_t>(from)];
    NodeList& dst = types_[static_cast<size_t>(to)];
    EdgeList& out = out_[e];
    const int64_t count = r.fanout[static_cast<size_t>(hop)];
    const int64_t* time = r.edge_time ? r.edge_time : (node_time_.empty() ? nullptr : node_time_[static_cast<size_t>(to)]);
    const bool edge_level = r.edge_time != nullptr;
    const size_t edges_before = out.rows.size();
    std::vector<int64_t> picks;
    for (int64_t i = src.frontier_begin; i < src.frontier_end; ++i) {
      const int64_t v = src.nodes[static_cast<size_t>(i)];
      const int64_t b = opt_.disjoint ? src.batch[static_cast<size_t>(i)] : 0;
      int64_t rs, re;
      range_of(r, time, edge_level, v, b, count, &rs, &re);
      const int64_t deg = re - rs;
      if (deg <= 0 || count == 0) continue;
      picks.clear();
      if (count < 0 || (!opt_.replace && count >= deg)) {
        for (int64_t k = 0; k < deg; ++k) picks.push_back(rs + k);
      } else if (r.weight.defined()) {
        const Tensor w = r.weight.narrow(0, rs, deg);
        Tensor idx;
        if (opt_.replace) {
          idx = at::multinomial(w, count, /*replacement=*/true);
        } else {
          const Tensor u = at::empty_like(w).uniform_();
          idx = std::get<1>((u.log() / w).topk(count));
        }
        const int64_t* p = idx.data_ptr<int64_t>();
        for (int64_t k = 0; k < count; ++k) picks.push_back(rs + p[k]);
      } else if (opt_.replace) {
        for (int64_t k = 0; k < count; ++k) picks.push_back(rs + static_cast<int64_t>(engine.below(static_cast<uint64_t>(deg))));
      } else {
        // draw j is uniform over [0, j] for j = deg - count .. deg - 1; a value seen before is replaced by j itself
        chosen_.clear();
        for (int64_t j = deg - count; j < deg; ++j) {
          int64_t pick = static_cast<int64_t>(engine.below(static_cast<uint64_t>(j) + 1));
          if (!chosen_.insert(pick).second) {
            pick = j;
            chosen_.insert(pick);
          }
          picks.push_back(rs + pick);
        }
      }
      for (const int64_t ed : picks) {
        const int64_t w = r.col[ed];
        const auto res = dst.ids.insert(w, b);
        if (res.second) {
          dst.nodes.push_back(w);
          if (opt_.disjoint) dst.batch.push_back(b);
        }
        out.rows.push_back(i);
        out.cols.push_back(res.first);
        out.eids.push_back(ed);
      }
    }
    out.per_hop.push_back(static_cast<int64_t>(out.rows.size() - edges_before));
  }

  std::vector<Relation> rels_;
  std::vector<const int64_t*> node_time_;
  Options opt_;
  std::vector<NodeList> types_;
  std::vector<EdgeList> out_;
  std::vector<int64_t> seed_times_;
  std::unordered_set<int64_t> chosen_;
};

// ---- tensor plumbing --------------------------------------------------------------------------------------------
struct Held {
  std::vector<Tensor> keep;
  at::ScalarType dtype = at::kLong;
  const int64_t* index(const Tensor& t, const char* what) {
    TORCH_CHECK(t.device().is_cpu(), "pyg (CPU): '", what, "' must be a CPU tensor like the seeds");
    TORCH_CHECK(t.is_contiguous(), "Non-contiguous '", what, "'");
    TORCH_CHECK(t.scalar_type() == dtype, "pyg (CPU): '", what, "' must have the seeds' dtype (", dtype, ")");
    keep.push_back(dtype == at::kLong ? t : t.to(at::kLong));
    return keep.back().data_ptr<int64_t>();
  }
  const int64_t* time(const Tensor& t, const char* what) {
    TORCH_CHECK(t.device().is_cpu() && t.is_contiguous() && t.scalar_type() == at::kLong, "pyg (CPU): '", what,
                "' must be a contiguous int64 CPU tensor");
    keep.push_back(t);
    return t.data_ptr<int64_t>();
  }
  Tensor weight(const Tensor& w, int64_t num_cols) {
    TORCH_CHECK(w.device().is_cpu() && w.is_contiguous() && w.dim() == 1 && w.numel() == num_cols,
                "pyg (CPU): 'edge_weight' needs one entry per edge");
    TORCH_CHECK(w.scalar_type() == at::kFloat || w.scalar_type() == at::kDouble, "pyg (CPU): 'edge_weight' must be float32 or float64");
    return w;
  }
  Tensor out(const std::vector<int64_t>& v) const {
    Tensor t = at::empty({static_cast<int64_t>(v.size())}, at::kLong);
    if (!v.empty()) std::memcpy(t.data_ptr<int64_t>(), v.data(), v.size() * sizeof(int64_t));
    return dtype == at::kLong ? t : t.to(dtype);
  }
  Tensor out_nodes(const NodeList& n, bool disjoint) const {
    if (!disjoint) return out(n.nodes);
    Tensor t = at::empty({static_cast<int64_t>(n.nodes.size()), 2}, at::kLong);
    int64_t* p = t.data_ptr<int64_t>();
    for (size_t i = 0; i < n.nodes.size(); ++i) {
      p[2 * i] = n.batch[i];
      p[2 * i + 1] = n.nodes[i];
    }
    return dtype == at::kLong ? t : t.to(dtype);
  }
};

static void check_modes(bool has_node_time, bool has_edge_time, bool has_seed_time, bool has_weight, bool directed,
                        bool disjoint, const std::string& temporal_strategy) {
  // sampler/cpu/neighbor_kernel.cpp:34-36,354-380,501
  TORCH_CHECK(temporal_strategy == "uniform" || temporal_strategy == "last", "No valid temporal strategy found");
  TORCH_CHECK(!has_node_time || disjoint, "Temporal sampling needs to create disjoint subgraphs");
  TORCH_CHECK(!has_edge_time || disjoint, "Temporal sampling needs to create disjoint subgraphs");
  TORCH_CHECK(!(has_node_time && has_edge_time), "Only one of node-level or edge-level sampling is supported ");
  TORCH_CHECK(!has_edge_time || has_seed_time, "Seed time needs to be specified");
  TORCH_CHECK(!(has_node_time && has_weight), "Biased node temporal sampling not yet supported");
  TORCH_CHECK(!(has_edge_time && has_weight), "Biased edge temporal sampling not yet supported");
  TORCH_CHECK(directed, "Undirected subgraphs not yet supported");
}

std::tuple<Tensor, Tensor, Tensor, c10::optional<Tensor>, std::vector<int64_t>, std::vector<int64_t>>
neighbor_sample_cpu(const Tensor& rowptr, const Tensor& col, const Tensor& seed, const std::vector<int64_t>& num_neighbors,
                    const c10::optional<Tensor>& node_time, const c10::optional<Tensor>& edge_time,
                    const c10::optional<Tensor>& seed_time, const c10::optional<Tensor>& edge_weight, bool csc, bool replace,
                    bool directed, bool disjoint, std::string temporal_strategy, bool return_edge_id) {
  PYG_TRACE("pyg::neighbor_sample[cpu]");
  check_modes(node_time.has_value(), edge_time.has_value(), seed_time.has_value(), edge_weight.has_value(), directed,
              disjoint, temporal_strategy);
  TORCH_CHECK(seed.scalar_type() == at::kLong || seed.scalar_type() == at::kInt, "pyg (CPU): indices must be int64 or int32");
  Held h;
  h.dtype = seed.scalar_type();
  Relation r;
  r.rowptr = h.index(rowptr, "rowptr");
  r.num_rows = rowptr.numel() - 1;
  r.col = h.index(col, "col");
  r.num_cols = col.numel();
  r.fanout = num_neighbors;
  if (edge_time.has_value()) r.edge_time = h.time(edge_time.value(), "edge_time");
  if (edge_weight.has_value()) r.weight = h.weight(edge_weight.value(), col.numel());
  SeedSet s;
  s.seed = h.index(seed, "seed");
  s.count = seed.numel();
  if (seed_time.has_value()) s.seed_time = h.time(seed_time.value(), "seed_time");
  std::vector<const int64_t*> ntime;
  if (node_time.has_value()) ntime.push_back(h.time(node_time.value(), "node_time"));
  Options opt;
  opt.csc = csc;
  opt.replace = replace;
  opt.disjoint = disjoint;
  opt.last = temporal_strategy == "last";
  opt.hops = static_cast<int>(num_neighbors.size());
  Sampler sampler({r}, 1, ntime, opt);
  sampler.run({s});
  const EdgeList& e = sampler.edges(0);
  c10::optional<Tensor> eid = c10::nullopt;
  if (return_edge_id) eid = h.out(e.eids);
  const Tensor rows = h.out(e.rows), cols = h.out(e.cols);
  return std::make_tuple(csc ? cols : rows, csc ? rows : cols, h.out_nodes(sampler.nodes(0), disjoint), eid,
                         sampler.nodes(0).per_hop, e.per_hop);
}

std::tuple<c10::Dict<rel_type, Tensor>, c10::Dict<rel_type, Tensor>, c10::Dict<node_type, Tensor>,
           c10::optional<c10::Dict<rel_type, Tensor>>, c10::Dict<node_type, std::vector<int64_t>>,
           c10::Dict<rel_type, std::vector<int64_t>>>
hetero_neighbor_sample_cpu(const std::vector<node_type>& node_types, const std::vector<edge_type>& edge_types,
                           const c10::Dict<rel_type, Tensor>& rowptr_dict, const c10::Dict<rel_type, Tensor>& col_dict,
                           const c10::Dict<node_type, Tensor>& seed_dict,
                           const c10::Dict<rel_type, std::vector<int64_t>>& num_neighbors_dict,
                           const c10::optional<c10::Dict<node_type, Tensor>>& node_time_dict,
                           const c10::optional<c10::Dict<rel_type, Tensor>>& edge_time_dict,
                           const c10::optional<c10::Dict<node_type, Tensor>>& seed_time_dict,
                           const c10::optional<c10::Dict<rel_type, Tensor>>& edge_weight_dict, bool csc, bool replace,
                           bool directed, bool disjoint, std::string temporal_strategy, bool return_edge_id) {
  PYG_TRACE("pyg::hetero_neighbor_sample[cpu]");
  check_modes(node_time_dict.has_value(), edge_time_dict.has_value(), seed_time_dict.has_value(),
              edge_weight_dict.has_value(), directed, disjoint, temporal_strategy);
  std::unordered_map<std::string, int> type_index;
  for (size_t i = 0; i < node_types.size(); ++i) type_index[node_types[i]] = static_cast<int>(i);
  TORCH_CHECK(seed_dict.size() > 0, "hetero_neighbor_sample: no seeds given");
  Held h;
  h.dtype = seed_dict.begin()->value().scalar_type();
  TORCH_CHECK(h.dtype == at::kLong || h.dtype == at::kInt, "pyg (CPU): indices must be int64 or int32");
  std::vector<Relation> rels(edge_types.size());
  size_t hops = 0;
  for (size_t e = 0; e < edge_types.size(); ++e) {
    const auto& k = edge_types[e];
    const auto name = rel_name(k);
    TORCH_CHECK(type_index.count(std::get<0>(k)) && type_index.count(std::get<2>(k)),
                "hetero_neighbor_sample: edge type names an unknown node type");
    const Tensor& rowptr = rowptr_dict.at(name);
    const Tensor& col = col_dict.at(name);
    Relation& r = rels[e];
    r.rowptr = h.index(rowptr, "rowptr");
    r.num_rows = rowptr.numel() - 1;
    r.col = h.index(col, "col");
    r.num_cols = col.numel();
    r.src = type_index[std::get<0>(k)];
    r.dst = type_index[std::get<2>(k)];
    r.fanout = num_neighbors_dict.at(name);
    hops = std::max(hops, r.fanout.size());
    if (edge_time_dict.has_value() && edge_time_dict.value().contains(name))
      r.edge_time = h.time(edge_time_dict.value().at(name), "edge_time");
    if (edge_weight_dict.has_value() && edge_weight_dict.value().contains(name))
      r.weight = h.weight(edge_weight_dict.value().at(name), col.numel());
  }
  for (const auto& r : rels) TORCH_CHECK(r.fanout.size() == hops, "hetero_neighbor_sample: all relations must list ", hops, " hops");
  std::vector<SeedSet> seeds;
  for (const auto& kv : seed_dict) {  // insertion order, as the reference relies on
    TORCH_CHECK(type_index.count(kv.key()), "hetero_neighbor_sample: seed type '", kv.key(), "' is not a node type");
    SeedSet s;
    s.type = type_index[kv.key()];
    s.seed = h.index(kv.value(), "seed");
    s.count = kv.value().numel();
    if (seed_time_dict.has_value()) s.seed_time = h.time(seed_time_dict.value().at(kv.key()), "seed_time");
    seeds.push_back(s);
  }
  std::vector<const int64_t*> ntime;
  if (node_time_dict.has_value()) {
    ntime.assign(node_types.size(), nullptr);
    for (const auto& kv : node_time_dict.value()) {
      TORCH_CHECK(type_index.count(kv.key()), "hetero_neighbor_sample: time given for unknown node type '", kv.key(), "'");
      ntime[static_cast<size_t>(type_index[kv.key()])] = h.time(kv.value(), "node_time");
    }
  }
  Options opt;
  opt.csc = csc;
  opt.replace = replace;
  opt.disjoint = disjoint;
  opt.last = temporal_strategy == "last";
  opt.hops = static_cast<int>(hops);
  Sampler sampler(rels, static_cast<int>(node_types.size()), ntime, opt);
  sampler.run(seeds);
  c10::Dict<rel_type, Tensor> out_row, out_col;
  c10::Dict<node_type, Tensor> out_node;
  c10::optional<c10::Dict<rel_type, Tensor>> out_eid;
  if (return_edge_id) out_eid = c10::Dict<rel_type, Tensor>();
  c10::Dict<node_type, std::vector<int64_t>> out_nph;
  c10::Dict<rel_type, std::vector<int64_t>> out_eph;
  for (size_t t = 0; t < node_types.size(); ++t) {
    out_node.insert(node_types[t], h.out_nodes(sampler.nodes(static_cast<int>(t)), disjoint));
    out_nph.insert(node_types[t], sampler.nodes(static_cast<int>(t)).per_hop);
  }
  for (size_t e = 0; e < edge_types.size(); ++e) {
    const auto name = rel_name(edge_types[e]);
    const EdgeList& el = sampler.edges(e);
    const Tensor rows = h.out(el.rows), cols = h.out(el.cols);
    out_row.insert(name, csc ? cols : rows);
    out_col.insert(name, csc ? rows : cols);
    out_eph.insert(name, el.per_hop);
    if (return_edge_id) out_eid.value().insert(name, h.out(el.eids));
  }
  return std::make_tuple(out_row, out_col, out_node, out_eid, out_nph, out_eph);
}

// ---------------------------------------------------------------------------------------------------------------
// matmul / index_sort
// ---------------------------------------------------------------------------------------------------------------
Tensor segment_matmul_cpu(const Tensor& input, const Tensor& ptr, const Tensor& other) {
  PYG_TRACE("pyg::segment_matmul[cpu]");
  at::TensorArg input_arg{input, "input", 0}, ptr_arg{ptr, "ptr", 1}, other_arg{other, "other", 2};
  at::CheckedFrom c{"segment_matmul"};
  at::checkAllDefined(c, {input_arg, ptr_arg, other_arg});
  at::checkSameType(c, input_arg, other_arg);
  at::checkDim(c, input_arg, 2);
  at::checkDim(c, ptr_arg, 1);
  at::checkDim(c, other_arg, 3);
  at::checkSize(c, other_arg, 1, input_arg->size(-1));
  at::checkNumel(c, ptr_arg, other_arg->size(0) + 1);
  TORCH_CHECK(ptr.scalar_type() == at::kLong, "expected scalar type Long but found ", ptr.scalar_type());
  const auto x = input.contiguous();
  const auto w = other.contiguous();
  const auto p = ptr.cpu().contiguous();
  const int64_t* pp = p.data_ptr<int64_t>();
  const int64_t B = w.size(0);
  auto out = x.new_empty({x.size(0), w.size(2)});
  for (int64_t b = 0; b < B; ++b) {
    TORCH_CHECK(pp[b] >= 0 && pp[b] <= pp[b + 1] && pp[b + 1] <= x.size(0), "segment_matmul: 'ptr' must be non-decreasing within [0, ",
                x.size(0), "]");
    if (pp[b + 1] == pp[b]) continue;
    auto dst = out.narrow(0, pp[b], pp[b + 1] - pp[b]);
    at::matmul_out(dst, x.narrow(0, pp[b], pp[b + 1] - pp[b]), w.select(0, b));
  }
  return out;
}

// this build's fused-bias variant: the reference adds bias[b] to every segment in Python (pyg_lib/ops/__init__.py:169-171)
Tensor segment_matmul_bias_cpu(const Tensor& input, const Tensor& ptr, const Tensor& other, const Tensor& bias) {
  Tensor out = segment_matmul_cpu(input, ptr, other);
  TORCH_CHECK(bias.dim() == 2 && bias.size(0) == other.size(0) && bias.size(1) == other.size(2),
              "segment_matmul: expected 'bias' of shape [", other.size(0), ", ", other.size(2), "]");
  const auto p = ptr.cpu().contiguous();
  const int64_t* pp = p.data_ptr<int64_t>();
  for (int64_t b = 0; b < other.size(0); ++b)
    if (pp[b + 1] > pp[b]) out.narrow(0, pp[b], pp[b + 1] - pp[b]).add_(bias.select(0, b));
  return out;
}

std::vector<Tensor> grouped_matmul_cpu(const at::TensorList input, const at::TensorList other) {
  PYG_TRACE("pyg::grouped_matmul[cpu]");
  TORCH_CHECK(input.size() == other.size(), "Number of 'input' tensors must match number of 'other' tensors");
  std::vector<Tensor> outs;
  outs.reserve(input.size());
  for (size_t i = 0; i < input.size(); ++i) {
    TORCH_CHECK(input[i].dim() == 2 && other[i].dim() == 2 && input[i].size(-1) == other[i].size(0) &&
                    input[i].scalar_type() == other[i].scalar_type(),
                "grouped_matmul: operands of group ", i, " do not multiply");
    outs.push_back(at::matmul(input[i].contiguous(), other[i].contiguous()));
  }
  return outs;
}

std::tuple<Tensor, Tensor> index_sort_cpu(const Tensor& input, const at::optional<int64_t> max) {
  PYG_TRACE("pyg::index_sort[cpu]");
  (void)max;  // only bounds the reference's radix passes
  TORCH_CHECK(input.is_contiguous(), "Input should be contiguous.");
  TORCH_CHECK(input.dim() == 1, "Input should be 1-dimensional.");
  TORCH_CHECK(at::isIntegralType(input.scalar_type(), /*includeBool=*/false), "Input should contain integral values.");
  return at::sort(input, /*stable=*/true, /*dim=*/0, /*descending=*/false);
}

}  // namespace cpu

// used by the BackendSelect kernel of hetero_neighbor_sample (pyg_binding.cpp): Dict values cannot drive dispatch
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
                              bool directed, bool disjoint, std::string temporal_strategy, bool return_edge_id) {
  return cpu::hetero_neighbor_sample_cpu(node_types, edge_types, rowptr_dict, col_dict, seed_dict, num_neighbors_dict,
                                         node_time_dict, edge_time_dict, seed_time_dict, edge_weight_dict, csc, replace,
                                         directed, disjoint, temporal_strategy, return_edge_id);
}

TORCH_LIBRARY_IMPL(pyg, CPU, m) {
  m.impl(TORCH_SELECTIVE_NAME("pyg::neighbor_sample"), TORCH_FN(cpu::neighbor_sample_cpu));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_matmul"), TORCH_FN(cpu::segment_matmul_cpu));
  m.impl(TORCH_SELECTIVE_NAME("pyg::segment_matmul_bias"), TORCH_FN(cpu::segment_matmul_bias_cpu));
  m.impl(TORCH_SELECTIVE_NAME("pyg::grouped_matmul"), TORCH_FN(cpu::grouped_matmul_cpu));
  m.impl(TORCH_SELECTIVE_NAME("pyg::index_sort"), TORCH_FN(cpu::index_sort_cpu));
}

}  // namespace pyg_amd
