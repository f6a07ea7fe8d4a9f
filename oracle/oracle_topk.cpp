/*
 * oracle_topk.cpp -- `Tensor.topk(k)` (largest, sorted) of a 1-D CPU tensor, as libtorch 2.10 computes it.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_sampler.c).  Used by the biased-sampling restatement:
 * pyg_lib/csrc/sampler/cpu/neighbor_kernel.cpp:275-277 takes `key.topk(count)` indices.
 *
 * Third-party arithmetic: libtorch's CPU top-k (ATen/native/TopKImpl.h:30-96, torch 2.10.0): (value, index)
 * pairs, std::partial_sort when k * 64 <= n, otherwise std::nth_element + std::sort of the first k - 1, with
 * the comparator "NaN first, then greater".  The ORDER OF TIES is whatever libstdc++'s algorithms produce,
 * so this file calls the very same std:: algorithms with the same comparator (the device path restates
 * them; tests/test_biased_sampler_gpu.py compares the two on tie-heavy rows).
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace {
template <typename T>
void topk_desc(const T* keys, int64_t n, int64_t k, int64_t* idx) {
  if (k == 0) return;
  using elem_t = std::pair<T, int64_t>;
  std::vector<elem_t> queue((size_t)n);
  for (int64_t j = 0; j < n; ++j) queue[(size_t)j] = {keys[j], j};
  auto cmp = [](const elem_t& x, const elem_t& y) -> bool {
    return ((std::isnan(x.first) && !std::isnan(y.first)) || (x.first > y.first));
  };
  if (k * 64 <= n) {
    std::partial_sort(queue.begin(), queue.begin() + k, queue.end(), cmp);
  } else {
    std::nth_element(queue.begin(), queue.begin() + k - 1, queue.end(), cmp);
    std::sort(queue.begin(), queue.begin() + k - 1, cmp);
  }
  for (int64_t j = 0; j < k; ++j) idx[j] = queue[(size_t)j].second;
}
}  // namespace

extern "C" void oracle_topk_desc_f32(const float* keys, int64_t n, int64_t k, int64_t* idx) { topk_desc(keys, n, k, idx); }
extern "C" void oracle_topk_desc_f64(const double* keys, int64_t n, int64_t k, int64_t* idx) { topk_desc(keys, n, k, idx); }
