// Biased (edge_weight) neighbour sampling kernels.  Included by sampler.hip INSIDE its anonymous namespace, after
// the per-hop types (HopArgs, HopInfo, CountAgg, IdxArr, mt_output_at, ...): a reading unit, not a translation unit.
#pragma once

// ---- biased sampling (edge_weight) -------------------------------------------------------------------
// _biased_sample (neighbor_kernel.cpp:245-285), replace == false: a row with more neighbours than the
// fan-out draws `rand = empty_like(weight).uniform_()` straight from the generator, forms
// key = rand.log() / weight and takes `key.topk(count)` -- the sampled edges in descending key order.
//   * uniform_ (serial CPU kernel): float32 -> one engine output, 24 bits kept; float64 -> random64() (two
//     outputs, first = high half), 53 bits kept (ATen/core/DistributionsHelper.h, TransformationHelper.h).
//     Row i of the frontier reads the outputs [raw_off[i], raw_off[i] + deg * outputs_per_draw): an
//     exclusive scan over the frontier (the prefetched RandintEngine is not touched).
//   * log: libtorch evaluates it with MKL (<1 ulp, closed source); here the correctly rounded logarithm
//     (float32: the f64 log rounded once; float64: the f64 log itself) -- see include/pyg_hip.h.
//   * topk (ATen/native/TopKImpl.h:30-96): comparator "NaN first, then greater" on (key, index) pairs,
//     std::partial_sort if count * 64 <= deg, else std::nth_element + std::sort of the first count - 1.
//     Keys are mapped to unsigned integers whose order is that comparator's.  Without equal keys among the
//     selected ones and at the selection boundary the result is simply the `count` largest in descending
//     order: one wave per row finds the count-th largest key bit by bit and ranks the selection by
//     counting.  Rows WITH such ties (zero weights -> -inf keys; equal weights with equal 24-bit draws) are
//     re-done by biased_exact_kernel, which performs libstdc++'s algorithms step for step.

// float32 logarithm of the biased path: the f64 log rounded once (pinned on all 2^24 arguments uniform_ can
// produce through pyg_hip_biased_log_f32, tests/test_biased_sampler_gpu.py)
__device__ __forceinline__ float biased_log_f32(float u) { return (float)log((double)u); }
__global__ void biased_log_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = biased_log_f32(in[i]);
}

template <bool F64> struct BiasedKey;
template <> struct BiasedKey<false> {
  typedef uint32_t K;
  typedef float W;
  static constexpr int kOutputs = 1;
  static constexpr int kBits = 32;
  __device__ static K make(const uint32_t* __restrict__ out32, int64_t o, float w) {
    const float u = (float)(mt_output_at(out32, o) & 0xffffffu) * 0x1p-24f;
    const float key = __fdiv_rn(biased_log_f32(u), w);
    uint32_t b = __float_as_uint(key);
    if (key != key) return ~0u;
    if (key == 0.f) b = 0u;  // -0 and +0 compare equal
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  }
};
template <> struct BiasedKey<true> {
  typedef uint64_t K;
  typedef double W;
  static constexpr int kOutputs = 2;
  static constexpr int kBits = 64;
  __device__ static K make(const uint32_t* __restrict__ out32, int64_t o, double w) {
    const uint64_t v = ((uint64_t)mt_output_at(out32, o) << 32) | mt_output_at(out32, o + 1);
    const double u = (double)(v & ((1ull << 53) - 1)) * 0x1p-53;
    const double key = log(u) / w;
    uint64_t b = (uint64_t)__double_as_longlong(key);
    if (key != key) return ~0ull;
    if (key == 0.0) b = 0ull;
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
  }
};

struct BiasedCountLoad {
  const int64_t* nodes;
  int64_t begin;
  IdxArr rowptr;
  int64_t count;
  int outputs;  // engine outputs per draw
  __device__ CountAgg operator()(int64_t i) const {
    CountAgg r;
    r.tab = rng_identity_packed();
    r.edges = 0;
    const int64_t v = nodes[begin + i];
    const int64_t deg = rowptr[v + 1] - rowptr[v];
    if (deg <= 0 || count == 0) return r;
    if (count < 0 || count >= deg) {
      r.edges = deg;
      return r;
    }
    r.edges = count;
    r.tab |= (u64)(deg * outputs) << 20;  // identity transitions compose additively in the word field
    return r;
  }
};

template <typename K>
struct BiasedArgs {
  HopArgs h;              // nodes, batch, begin, frontier, range.rowptr, col, count, edge_off, rng_word (= the
                          // row's first engine output), emission buffers, table
  HopInfo* info;          // tot.tab is reset to the identity (the engine does not move), overflow cleared
  const void* weight;
  const uint32_t* out32;  // generated engine outputs
  int64_t out_base;       // engine output behind key slot 0
  K* skey;                // [draws] keys of every drawing row, in draw order
  int32_t* sidx;          // [draws] rank scratch / index half of the exact path's pairs
  K* selkey;              // [edges] keys of the selected neighbours, in index order
  int32_t* selidx;        // [edges]
  int32_t* flag;          // [frontier] 1 = the row has ties and is left to biased_exact_kernel
};

__device__ __forceinline__ void wave_mem_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// A row of up to 64 R neighbours: keys stay in registers (R per lane).
template <bool F64, int R>
__device__ __forceinline__ void biased_row_in_registers(const BiasedArgs<typename BiasedKey<F64>::K>& a, int64_t i,
                                                        int lane, int64_t n, int64_t k, int64_t rs, int64_t eo,
                                                        int64_t o0, typename BiasedKey<F64>::K* sk,
                                                        const typename BiasedKey<F64>::W* w, int64_t src_pos,
                                                        int64_t batch) {
  typedef BiasedKey<F64> BK;
  typedef typename BK::K K;
  K x[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t j = lane + 64 * r;
    x[r] = 0;  // absent slot: below every real key (the smallest, -inf, maps to 0x007f...f)
    if (j < n) {
      x[r] = BK::make(a.out32, o0 + j * BK::kOutputs, w[j]);
      sk[j] = x[r];  // the exact path reads the keys from memory
    }
  }
  // T = the k-th largest key, bit by bit; absent slots hold 0 and candidates are > 0
  K T = 0;
  for (int b = BK::kBits - 1; b >= 0; --b) {
    const K cand = T | ((K)1 << b);
    int c = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) c += __popcll(__ballot(x[r] >= cand && lane + 64 * r < n));
    if (c >= k) T = cand;
  }
  int sel = 0, gt = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    sel += __popcll(__ballot(x[r] >= T && lane + 64 * r < n));
    gt += __popcll(__ballot(x[r] > T && lane + 64 * r < n));
  }
  // exactly k keys >= T, and the boundary key T is unique: otherwise a tie
  bool tie = sel != k || sel - gt != 1;
  // rank of every selected key = number of selected keys above it; equal selected keys = tie
  int rank[R];
#pragma unroll
  for (int r = 0; r < R; ++r) rank[r] = 0;
  if (!tie) {
    bool dup = false;
#pragma unroll
    for (int r2 = 0; r2 < R; ++r2) {
      u64 m = __ballot(x[r2] >= T && lane + 64 * r2 < n);
      while (m) {
        const int q = __ffsll((long long)m) - 1;
        m &= m - 1;
        const K y = __shfl(x[r2], q, 64);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          rank[r] += y > x[r] ? 1 : 0;
          dup = dup || (y == x[r] && (q != lane || r2 != r) && x[r] >= T && lane + 64 * r < n);
        }
      }
    }
    tie = __ballot(dup) != 0;
  }
  if (tie) {
    if (lane == 0) a.flag[i] = 1;
    return;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t j = lane + 64 * r;
    if (j < n && x[r] >= T) emit(a.h, eo + rank[r], rs + j, src_pos, batch);
  }
}

template <bool F64>
__global__ __launch_bounds__(256) void biased_sample_kernel(BiasedArgs<typename BiasedKey<F64>::K> a) {
  typedef BiasedKey<F64> BK;
  typedef typename BK::K K;
  const int lane = threadIdx.x & 63;
  const int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.info->tot.tab = rng_identity();
    a.info->overflow = 0;
  }
  if (i >= a.h.frontier) return;
  const int64_t src_pos = a.h.begin + i;
  const int64_t v = a.h.nodes[src_pos];
  const int64_t batch = a.h.batch ? a.h.batch[src_pos] : 0;
  const int64_t rs = a.h.range.rowptr[v];
  const int64_t n = a.h.range.rowptr[v + 1] - rs;
  const int64_t k = a.h.count;
  if (lane == 0) a.flag[i] = 0;
  if (n <= 0 || k == 0) return;
  const int64_t eo = a.h.edge_off[i];
  if (k < 0 || k >= n) {  // the full neighbourhood, no draws (:257-262)
    for (int64_t j = lane; j < n; j += 64) emit(a.h, eo + j, rs + j, src_pos, batch);
    return;
  }
  const int64_t o0 = a.h.rng_word[i];
  const int64_t ko = (o0 - a.out_base) / BK::kOutputs;
  K* sk = a.skey + ko;
  const typename BK::W* w = static_cast<const typename BK::W*>(a.weight) + rs;
  if (n <= 64) {
    biased_row_in_registers<F64, 1>(a, i, lane, n, k, rs, eo, o0, sk, w, src_pos, batch);
    return;
  }
  if (n <= 256) {
    biased_row_in_registers<F64, 4>(a, i, lane, n, k, rs, eo, o0, sk, w, src_pos, batch);
    return;
  }
  for (int64_t j = lane; j < n; j += 64) sk[j] = BK::make(a.out32, o0 + j * BK::kOutputs, w[j]);
  wave_mem_sync();
  // T = the k-th largest key: radix select, one pass over the row per 8-bit digit (per-wave LDS histogram)
  __shared__ uint32_t hist_all[4][256];
  uint32_t* hist = hist_all[threadIdx.x >> 6];
  K T = 0;
  int64_t krem = k;  // the krem-th largest of the keys that share the digits fixed so far
  uint32_t c_eq = 0;
  for (int shift = BK::kBits - 8; shift >= 0; shift -= 8) {
    for (int b = lane; b < 256; b += 64) hist[b] = 0;
    wave_mem_sync();
    const K hi_mask = shift + 8 >= BK::kBits ? (K)0 : (~(K)0) << (shift + 8);
    for (int64_t j = lane; j < n; j += 64) {
      const K x = sk[j];
      if ((x & hi_mask) == T) atomicAdd(&hist[(uint32_t)(x >> shift) & 255u], 1u);
    }
    wave_mem_sync();
    uint32_t h[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) h[q] = hist[4 * lane + q];
    const uint32_t s4 = h[0] + h[1] + h[2] + h[3];
    uint32_t t = s4;  // -> sum over lanes >= lane (higher lanes hold higher digits)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t v = __shfl_down(t, off, 64);
      if (lane + off < 64) t += v;
    }
    uint32_t above = t - s4;  // keys with a digit above this lane's four
    int digit = -1;
    uint32_t need = 0, cnt = 0;
#pragma unroll
    for (int q = 3; q >= 0; --q) {
      if (digit < 0 && (int64_t)above < krem && krem <= (int64_t)above + h[q]) {
        digit = 4 * lane + q;
        need = (uint32_t)(krem - above);
        cnt = h[q];
      }
      above += h[q];
    }
    const u64 m = __ballot(digit >= 0);
    const int src = __ffsll((long long)m) - 1;  // exactly one lane finds it
    digit = __shfl(digit, src, 64);
    krem = __shfl(need, src, 64);
    c_eq = __shfl(cnt, src, 64);
    T |= (K)(uint32_t)digit << shift;
  }
  // exactly k keys are >= T iff the boundary key is unique (c_eq == krem == 1)
  // selection = keys >= T, compacted in index order; more than k of them = a tie at the boundary
  int64_t sel = 0;
  for (int64_t j0 = 0; j0 < n; j0 += 64) {
    const int64_t j = j0 + lane;
    const K x = j < n ? sk[j] : (K)0;
    const bool in = j < n && x >= T;
    const u64 m = __ballot(in);
    const int64_t p = sel + __popcll(m & ((1ull << lane) - 1));
    if (in && p < k) {
      a.selkey[eo + p] = x;
      a.selidx[eo + p] = (int32_t)j;
    }
    sel += __popcll(m);
  }
  bool tie = sel > k || c_eq != 1 || krem != 1;
  wave_mem_sync();
  if (!tie) {
    // rank by counting; equal keys inside the selection are ties as well
    bool dup = false;
    for (int64_t p = lane; p < k; p += 64) {
      const K x = a.selkey[eo + p];
      int32_t rank = 0, same = 0;
      for (int64_t q = 0; q < k; ++q) {
        const K y = a.selkey[eo + q];
        rank += y > x ? 1 : 0;
        same += y == x ? 1 : 0;
      }
      dup = dup || same > 1;
      a.sidx[ko + p] = rank;
    }
    tie = __ballot(dup) != 0;
  }
  if (tie) {
    if (lane == 0) a.flag[i] = 1;
    return;
  }
  for (int64_t p = lane; p < k; p += 64) emit(a.h, eo + a.sidx[ko + p], rs + a.selidx[eo + p], src_pos, batch);
}

// libstdc++'s heap / introselect / introsort on (key, index) pairs held in two arrays, with the comparator
// "x before y  <=>  key(x) > key(y)" (bits/stl_heap.h, bits/stl_algo.h of GCC 11; the algorithms have not
// changed in a decade).  Sequential by nature: one thread per row, for the rare rows with tied keys.
template <typename K>
struct PairSeq {
  K* k;
  int32_t* v;
  struct V {
    K k;
    int32_t v;
  };
  __device__ V at(int64_t i) const { return V{k[i], v[i]}; }
  __device__ void put(int64_t i, V x) const {
    k[i] = x.k;
    v[i] = x.v;
  }
  __device__ void swp(int64_t i, int64_t j) const {
    const V t = at(i);
    put(i, at(j));
    put(j, t);
  }
  __device__ static bool lt(const V& x, const V& y) { return x.k > y.k; }
  __device__ static int lg(int64_t n) { return 63 - __clzll((unsigned long long)n); }

  __device__ void push_heap(int64_t first, int64_t hole, int64_t top, V value) const {
    int64_t parent = (hole - 1) / 2;
    while (hole > top && lt(at(first + parent), value)) {
      put(first + hole, at(first + parent));
      hole = parent;
      parent = (hole - 1) / 2;
    }
    put(first + hole, value);
  }
  __device__ void adjust_heap(int64_t first, int64_t hole, int64_t len, V value) const {
    const int64_t top = hole;
    int64_t child = hole;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      if (lt(at(first + child), at(first + (child - 1)))) child--;
      put(first + hole, at(first + child));
      hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
      child = 2 * (child + 1);
      put(first + hole, at(first + (child - 1)));
      hole = child - 1;
    }
    push_heap(first, hole, top, value);
  }
  __device__ void make_heap(int64_t first, int64_t last) const {
    const int64_t len = last - first;
    if (len < 2) return;
    int64_t parent = (len - 2) / 2;
    for (;;) {
      adjust_heap(first, parent, len, at(first + parent));
      if (parent == 0) return;
      parent--;
    }
  }
  __device__ void pop_heap(int64_t first, int64_t last, int64_t result) const {
    const V value = at(result);
    put(result, at(first));
    adjust_heap(first, 0, last - first, value);
  }
  __device__ void heap_select(int64_t first, int64_t middle, int64_t last) const {
    make_heap(first, middle);
    for (int64_t i = middle; i < last; ++i)
      if (lt(at(i), at(first))) pop_heap(first, middle, i);
  }
  __device__ void sort_heap(int64_t first, int64_t last) const {
    while (last - first > 1) {
      --last;
      pop_heap(first, last, last);
    }
  }
  __device__ void partial_sort(int64_t first, int64_t middle, int64_t last) const {
    heap_select(first, middle, last);
    sort_heap(first, middle);
  }
  __device__ void move_median_to_first(int64_t result, int64_t a, int64_t b, int64_t c) const {
    if (lt(at(a), at(b))) {
      if (lt(at(b), at(c))) swp(result, b);
      else if (lt(at(a), at(c))) swp(result, c);
      else swp(result, a);
    } else if (lt(at(a), at(c))) swp(result, a);
    else if (lt(at(b), at(c))) swp(result, c);
    else swp(result, b);
  }
  __device__ int64_t unguarded_partition(int64_t first, int64_t last, int64_t pivot) const {
    for (;;) {
      while (lt(at(first), at(pivot))) ++first;
      --last;
      while (lt(at(pivot), at(last))) --last;
      if (!(first < last)) return first;
      swp(first, last);
      ++first;
    }
  }
  __device__ int64_t unguarded_partition_pivot(int64_t first, int64_t last) const {
    const int64_t mid = first + (last - first) / 2;
    move_median_to_first(first, first + 1, mid, last - 1);
    return unguarded_partition(first + 1, last, first);
  }
  __device__ void unguarded_linear_insert(int64_t last) const {
    const V val = at(last);
    int64_t next = last - 1;
    while (lt(val, at(next))) {
      put(last, at(next));
      last = next;
      --next;
    }
    put(last, val);
  }
  __device__ void insertion_sort(int64_t first, int64_t last) const {
    if (first == last) return;
    for (int64_t i = first + 1; i != last; ++i) {
      if (lt(at(i), at(first))) {
        const V val = at(i);
        for (int64_t j = i; j > first; --j) put(j, at(j - 1));
        put(first, val);
      } else {
        unguarded_linear_insert(i);
      }
    }
  }
  __device__ void nth_element(int64_t first, int64_t nth, int64_t last) const {
    if (first == last || nth == last) return;
    int depth = lg(last - first) * 2;
    while (last - first > 3) {
      if (depth == 0) {
        heap_select(first, nth + 1, last);
        swp(first, nth);
        return;
      }
      --depth;
      const int64_t cut = unguarded_partition_pivot(first, last);
      if (cut <= nth) first = cut;
      else last = cut;
    }
    insertion_sort(first, last);
  }
  __device__ void sort(int64_t first, int64_t last) const {
    if (first == last) return;
    // __introsort_loop recurses into the right part and loops on the left one; the parts are disjoint, so
    // an explicit stack of pending right parts yields the same arrangement
    int64_t sf[130], sl[130];
    int sd[130];
    int top = 0;
    sf[0] = first;
    sl[0] = last;
    sd[0] = lg(last - first) * 2;
    top = 1;
    while (top > 0) {
      --top;
      int64_t f = sf[top], l = sl[top];
      int d = sd[top];
      while (l - f > 16) {
        if (d == 0) {
          partial_sort(f, l, l);
          break;
        }
        --d;
        const int64_t cut = unguarded_partition_pivot(f, l);
        sf[top] = cut;
        sl[top] = l;
        sd[top] = d;
        ++top;
        l = cut;
      }
    }
    if (last - first > 16) {
      insertion_sort(first, first + 16);
      for (int64_t i = first + 16; i != last; ++i) unguarded_linear_insert(i);
    } else {
      insertion_sort(first, last);
    }
  }
};

template <bool F64>
__global__ __launch_bounds__(64) void biased_exact_kernel(BiasedArgs<typename BiasedKey<F64>::K> a) {
  typedef BiasedKey<F64> BK;
  typedef typename BK::K K;
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= a.h.frontier || !a.flag[i]) return;
  const int64_t src_pos = a.h.begin + i;
  const int64_t v = a.h.nodes[src_pos];
  const int64_t batch = a.h.batch ? a.h.batch[src_pos] : 0;
  const int64_t rs = a.h.range.rowptr[v];
  const int64_t n = a.h.range.rowptr[v + 1] - rs;
  const int64_t k = a.h.count;
  const int64_t eo = a.h.edge_off[i];
  const int64_t ko = (a.h.rng_word[i] - a.out_base) / BK::kOutputs;
  PairSeq<K> s{a.skey + ko, a.sidx + ko};
  for (int64_t j = 0; j < n; ++j) s.v[j] = (int32_t)j;
  if (k * 64 <= n) {
    s.partial_sort(0, k, n);
  } else {
    s.nth_element(0, k - 1, n);
    s.sort(0, k - 1);
  }
  for (int64_t p = 0; p < k; ++p) emit(a.h, eo + p, rs + s.v[p], src_pos, batch);
}

// With replacement (neighbor_kernel.cpp:267-270): index = at::multinomial(weight, count, true).  For count > 1
// libtorch's CPU kernel (ATen/native/cpu/MultinomialKernel.cpp) sums the cumulative distribution SEQUENTIALLY in the
// weights' type, divides every entry by the sum, sets the last one to 1, and for each sample draws one double
// (random64, 53 bits) and binary-searches the first entry that is not below it.  One wave per row: lane 0 runs the
// sequential sum (bit-exactness leaves no choice), the division and the `count` searches are spread over the lanes.
// Every emitting row draws 2 count outputs and emits count edges, so its first output is out_base + 2 edge_off.
// A distribution at::multinomial rejects raises info->overflow = 3.  (count == 1 goes through exponential_ with
// MKL's own generator inside at::multinomial: refused by the host.)
struct BiasedReplaceCountLoad {
  const int64_t* nodes;
  int64_t begin;
  IdxArr rowptr;
  int64_t count;
  __device__ CountAgg operator()(int64_t i) const {
    CountAgg r;
    r.tab = rng_identity_packed();
    r.edges = 0;
    const int64_t v = nodes[begin + i];
    const int64_t deg = rowptr[v + 1] - rowptr[v];
    if (deg <= 0 || count == 0) return r;
    if (count < 0) {
      r.edges = deg;
      return r;
    }
    r.edges = count;
    r.tab |= (u64)deg << 20;  // scratch entries of the cumulative distribution
    return r;
  }
};

template <typename W>
__global__ __launch_bounds__(256) void biased_replace_kernel(HopArgs a, HopInfo* info, const W* __restrict__ weight,
                                                             const uint32_t* __restrict__ out32, int64_t out_base,
                                                             W* __restrict__ cum_all) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) info->tot.tab = rng_identity();
  if (i >= a.frontier) return;
  const int64_t src_pos = a.begin + i;
  const int64_t v = a.nodes[src_pos];
  const int64_t batch = a.batch ? a.batch[src_pos] : 0;
  const int64_t rs = a.range.rowptr[v];
  const int64_t n = a.range.rowptr[v + 1] - rs;
  const int64_t k = a.count;
  if (n <= 0 || k == 0) return;
  const int64_t eo = a.edge_off[i];
  if (k < 0) {
    for (int64_t j = lane; j < n; j += 64) emit(a, eo + j, rs + j, src_pos, batch);
    return;
  }
  W* cum = cum_all + a.rng_word[i];
  const W* w = weight + rs;
  W sum = 0;
  int bad = 0;
  if (lane == 0) {
    for (int64_t j = 0; j < n; ++j) {
      const W x = w[j];
      if (!(x >= (W)0) || isinf(x)) bad = 1;
      sum += x;
      cum[j] = sum;
    }
    if (!(sum > (W)0)) bad = 1;
  }
  bad = __shfl(bad, 0, 64);
  if (bad) {
    if (lane == 0) info->overflow = 3;
    return;
  }
  sum = __shfl(sum, 0, 64);
  wave_mem_sync();
  for (int64_t j = lane; j < n; j += 64) cum[j] = cum[j] / sum;
  wave_mem_sync();
  if (lane == 0) cum[n - 1] = (W)1;
  wave_mem_sync();
  for (int64_t s = lane; s < k; s += 64) {
    const int64_t o = out_base + 2 * (eo + s);
    const uint64_t r64 = ((uint64_t)mt_output_at(out32, o) << 32) | mt_output_at(out32, o + 1);
    const double u = (double)(r64 & ((1ull << 53) - 1)) * 0x1p-53;
    int64_t lo = 0, hi = n;
    while (hi - lo > 0) {
      const int64_t mid = lo + (hi - lo) / 2;
      if ((double)cum[mid] < u) lo = mid + 1;
      else hi = mid;
    }
    emit(a, eo + s, rs + lo, src_pos, batch);
  }
}

// at::multinomial(weight, 1, true): the single-draw route (ATen/native/Distributions.cpp) -- q =
// empty_like(weight).exponential_(1), index = argmax(weight / q).  libtorch 2.10.0 evaluates exponential_ on the
// CPU as -log1p(-u) with ONE 53-bit double per element (random64, also for float32 tensors; the value is then
// rounded to the tensor's type), argmax returns the first of equal maxima and treats NaN as the maximum.  One wave
// per row: every row draws 2 deg outputs (offset = the scan's word field), lanes keep (best key, index).
struct BiasedSingleCountLoad {
  const int64_t* nodes;
  int64_t begin;
  IdxArr rowptr;
  __device__ CountAgg operator()(int64_t i) const {
    CountAgg r;
    r.tab = rng_identity_packed();
    r.edges = 0;
    const int64_t v = nodes[begin + i];
    const int64_t deg = rowptr[v + 1] - rowptr[v];
    if (deg <= 0) return r;
    r.edges = 1;
    r.tab |= (u64)(2 * deg) << 20;
    return r;
  }
};

template <typename W>
__global__ __launch_bounds__(256) void biased_single_kernel(HopArgs a, HopInfo* info, const W* __restrict__ weight,
                                                            const uint32_t* __restrict__ out32) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) info->tot.tab = rng_identity();
  if (i >= a.frontier) return;
  const int64_t src_pos = a.begin + i;
  const int64_t v = a.nodes[src_pos];
  const int64_t batch = a.batch ? a.batch[src_pos] : 0;
  const int64_t rs = a.range.rowptr[v];
  const int64_t n = a.range.rowptr[v + 1] - rs;
  if (n <= 0) return;
  const int64_t o0 = a.rng_word[i];
  const W* w = weight + rs;
  // order: NaN above everything, then by value; ties keep the smaller index
  int64_t best = -1;
  int best_nan = 0;
  double best_key = 0.0;
  double sum = 0.0;
  int bad = 0;
  for (int64_t j = lane; j < n; j += 64) {
    const uint64_t r64 = ((uint64_t)mt_output_at(out32, o0 + 2 * j) << 32) | mt_output_at(out32, o0 + 2 * j + 1);
    const double u = (double)(r64 & ((1ull << 53) - 1)) * 0x1p-53;
    const double q64 = -log1p(-u);
    const W wj = w[j];
    double key;
    if (sizeof(W) == 4) key = (double)__fdiv_rn((float)wj, (float)q64);
    else key = (double)wj / q64;
    if (!(wj >= (W)0) || isinf(wj)) bad = 1;
    sum += (double)wj;
    const int is_nan = key != key;
    if (best < 0 || (!best_nan && (is_nan || key > best_key))) {
      best = j;
      best_key = key;
      best_nan = is_nan;
    }
  }
  // wave reduction of (nan, key, index) with the same order; lanes without elements hold best = -1
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const int64_t ob = __shfl_xor(best, d, 64);
    const int on = __shfl_xor(best_nan, d, 64);
    const double ok = __shfl_xor(best_key, d, 64);
    sum += __shfl_xor(sum, d, 64);
    bad |= __shfl_xor(bad, d, 64);
    bool take = false;
    if (ob >= 0) {
      if (best < 0) take = true;
      else if (on != best_nan) take = on > best_nan;
      else if (!on && ok != best_key) take = ok > best_key;
      else take = ob < best;
    }
    if (take) {
      best = ob;
      best_nan = on;
      best_key = ok;
    }
  }
  if (bad || !(sum > 0.0)) {
    if (lane == 0) info->overflow = 3;
    return;
  }
  if (lane == 0) emit(a, a.edge_off[i], rs + best, src_pos, batch);
}
