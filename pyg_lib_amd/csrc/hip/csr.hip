// segment_{sum,mean,min,max}_csr, gather_csr and softmax_csr (+ backward) for gfx950 (MI355X).
//
// Replaces pyg_lib/csrc/ops/cuda/segment_csr_kernel.cu (warp-32 shuffles) and gives
// pyg::softmax_csr -- CPU only in the reference (ops/cpu/softmax_kernel.cpp) -- a device path.
// Contracts follow pyg_lib/csrc/ops/cpu/segment_csr_kernel.cpp: src viewed as [leading, E, K],
// indptr [leading, rows + 1] (slice stride 0 = one indptr broadcast over the slices, read in place
// instead of `expand().contiguous()`), out / arg [leading, rows, K].
//
// CSR rows write disjoint outputs, so nothing is atomic.  HBM-bound byte work:
//   * L = 1 (default): one thread per (row, 16-byte column slice) walks its row in source order and
//     accumulates in the reference's opmath -- the CPU kernel's exact operation order, so sums and
//     means are BIT-exact with the reference for every dtype (min/max/arg trivially).
//   * L = 8 / 64 lanes per (row, slice) when rows are long and few (average length >= 64 / 1024 and
//     too few rows to fill the chip): lane j takes e = a + j, a + j + L, ...; partial results are
//     combined by xor shuffles (floating sums then differ from the CPU order by rounding; min/max and
//     their first-match arg stay exact through a lexicographic (value, position) combine).
#include "common.h"
#include "elem.h"

#include <algorithm>
#include <cstdint>
#include <type_traits>

namespace pyg_hip {
namespace {

enum { CSR_SUM = 0, CSR_MEAN = 1, CSR_MIN = 2, CSR_MAX = 3 };

template <typename T, int V>
struct alignas(sizeof(T) * V) Pack {
  T v[V];
};

// what a FRESH min / max output starts from (pyg_hip_fill_reduce_identity's value): with fresh != 0 the kernels do not read `out`
template <typename T, int OP>
__device__ __forceinline__ typename Math<T>::acc_t minmax_identity() {
  return Math<T>::up(OP == CSR_MIN ? type_max<T>() : type_lowest<T>());
}

// pin_all(a): every element of `a` is needed HERE, all of them at once.  The row kernels load U positions' values "in flight
// together" and then use them under `if (position < end)`; the compiler sinks each load to its conditional use, and the
// kernel runs with ONE load in flight per thread (load, s_waitcnt vmcnt(0), branch, load, ...: the ISA of every such loop
// before this existed).  One empty asm statement that takes all U values as register operands cannot be split.
template <int BYTES> struct PinReg { using type = uint32_t; };
template <> struct PinReg<8> { using type = uint64_t; };
template <> struct PinReg<16> { typedef unsigned type __attribute__((ext_vector_type(4))); };
template <typename X, int U>
__device__ __forceinline__ void pin_all(X (&a)[U]) {
  static_assert(U == 4 || U == 8 || U == 16, "pin_all: 4, 8 or 16 values");
  if constexpr (sizeof(X) == 32) {   // two 16-byte halves each
    using Q = typename PinReg<16>::type;
    struct H { Q lo, hi; };
    Q h[2 * U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const H t = __builtin_bit_cast(H, a[u]);
      h[2 * u] = t.lo, h[2 * u + 1] = t.hi;
    }
    pin_all(h);
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = __builtin_bit_cast(X, H{h[2 * u], h[2 * u + 1]});
    return;
  } else {
  static_assert(sizeof(X) == 1 || sizeof(X) == 2 || sizeof(X) == 4 || sizeof(X) == 8 || sizeof(X) == 16, "pin_all: value size");
  using R = typename PinReg<sizeof(X)>::type;
  R r[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if constexpr (sizeof(X) == 1) r[u] = __builtin_bit_cast(uint8_t, a[u]);
    else if constexpr (sizeof(X) == 2) r[u] = __builtin_bit_cast(uint16_t, a[u]);
    else r[u] = __builtin_bit_cast(R, a[u]);
  }
  if constexpr (U == 4) {
    asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
  } else if constexpr (U == 8) {
    asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
  } else {
    asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                 "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]));
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if constexpr (sizeof(X) == 1) a[u] = __builtin_bit_cast(X, (uint8_t)r[u]);
    else if constexpr (sizeof(X) == 2) a[u] = __builtin_bit_cast(X, (uint16_t)r[u]);
    else a[u] = __builtin_bit_cast(X, r[u]);
  }
  }
}

template <typename A>
__device__ __forceinline__ A shfl_xor_any(A v, int mask) {
  static_assert(sizeof(A) % 4 == 0 || sizeof(A) < 4, "unsupported accumulator");
  if constexpr (sizeof(A) < 4) {
    int t = (int)v;
    t = __shfl_xor(t, mask);
    return (A)t;
  } else {
    A out;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
    uint32_t* d = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(A) / 4); ++i) d[i] = (uint32_t)__shfl_xor((int)s[i], mask);
    return out;
  }
}

struct CsrShape {
  int64_t leading, rows, E, K;
  int64_t indptr_stride;  // elements between the indptr of consecutive slices (0 = broadcast)
  int64_t long_cut = INT64_MAX;  // segment_csr_kernel: rows of more positions are left to the hub kernels below
  void* hub_ws = nullptr;        // (host side) the caller's scratch for them: pyg_hip_csr_hub_workspace_size()
  size_t hub_ws_bytes = 0;
};

// HUB rows (more than `long_cut` positions: a destination that collects 0.25 % of 8 M positions costs 8 ms in the row
// kernels where the whole call takes 0.5 without it; power-law graphs have such nodes).  The row kernels skip them.  With
// the caller's scratch the thread that skips one REGISTERS it here, cut into chunks of `CH` positions, and a second launch
// deals the chunks to workgroups (hub_chunk kernels); without scratch a second launch finds the hubs again and gives each
// to one workgroup (the *_long kernels: ~9 GB/s per hub row).
struct HubRec {
  int64_t n;            // flat row
  int chunk_base, nch;  // its chunks: chunks[chunk_base ... chunk_base + nch)
  int slot_base, pad;   // nch > 1: its partial results' slots
};
struct HubWs {
  int* counters = nullptr;  // [0] hubs, [1] chunks, [2] partial slots -- zeroed in front of the row kernel
  HubRec* hubs = nullptr;
  int2* chunks = nullptr;   // (hub, chunk of the hub)
  char* partial = nullptr;  // slots of K accumulators ...
  int64_t* partial_best = nullptr;  // ... and, for min / max, of K positions
  int64_t CH = 0;
  int max_hubs = 0, max_chunks = 0, max_slots = 0;   // capacities: valid offsets never reach them; garbage offsets set counters[3]
};

__device__ __forceinline__ void hub_register(const HubWs& hw, int64_t n, int64_t len) {
  const int64_t nch64 = (len + hw.CH - 1) / hw.CH;
  if (nch64 > hw.max_chunks) {   // offsets that are no offsets (the reference does not check them either): nothing is written
    hw.counters[3] = 1;          // behind the scratch's ends, the hub kernels of this call stand down
    return;
  }
  const int nch = (int)nch64;
  const int h = atomicAdd(&hw.counters[0], 1);
  const int cb = atomicAdd(&hw.counters[1], nch);
  const bool slots = nch > 1 && hw.max_slots > 0;   // (gather_csr keeps no partial results)
  const int sb = slots ? atomicAdd(&hw.counters[2], nch) : 0;
  if (h >= hw.max_hubs || cb + nch > hw.max_chunks || (slots && sb + nch > hw.max_slots)) {
    hw.counters[3] = 1;
    return;
  }
  hw.hubs[h] = HubRec{n, cb, nch, sb, 0};
  for (int j = 0; j < nch; ++j) hw.chunks[cb + j] = make_int2(h, j);
}

// OP: CSR_SUM / CSR_MEAN / CSR_MIN / CSR_MAX.  V elements (16 bytes, or 1) per thread, L lanes per item.
// PERM: row positions are read through perm[e] (scatter_min/max after an index sort; perm is ascending
// inside a row, so "first match" is still the smallest source position).
template <typename T, int OP, int V, int L, bool PERM>
__global__ __launch_bounds__(256) void segment_csr_kernel(const T* __restrict__ src, const int64_t* __restrict__ indptr,
                                                          const int64_t* __restrict__ perm, T* __restrict__ out,
                                                          int64_t* __restrict__ arg, int fresh, CsrShape s, HubWs hw) {
  using acc_t = typename Math<T>::acc_t;
  using P = Pack<T, V>;
  const int64_t kv = s.K / V;
  const int64_t t = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / L;
  const int lane = threadIdx.x & (L - 1);
  const int64_t items = s.leading * s.rows * kv;
  const bool live = t < items;
  const int64_t tt = live ? t : 0;
  const int64_t n = tt / kv;        // flat row
  const int64_t c = (tt % kv) * V;  // first column of this thread's slice
  const int64_t slice = n / s.rows;
  const int64_t row = n % s.rows;
  const int64_t a = indptr[slice * s.indptr_stride + row];
  const int64_t b = indptr[slice * s.indptr_stride + row + 1];
  if (b - a > s.long_cut) {   // (all L lanes of the item alike) a hub row
    if (hw.counters && live && lane == 0 && c == 0) hub_register(hw, n, b - a);
    return;
  }
  const T* sp = src + slice * s.E * s.K + c;
  T* op = out + n * s.K + c;

  acc_t acc[V];
  int64_t best[V];
  if constexpr (OP == CSR_SUM) {
    // the accumulator is seeded from the caller's `out` slot; a fresh output starts at +0 without being read (it
    // need not be cleared either: every slot is written below -- 2 x N x K bytes less traffic)
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = acc_t(0);
    if (!fresh) {
      P cur = *reinterpret_cast<const P*>(op);
#pragma unroll
      for (int i = 0; i < V; ++i) acc[i] = lane == 0 ? Math<T>::up(cur.v[i]) : acc_t(0);
    }
  } else if constexpr (OP == CSR_MEAN) {
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = acc_t(0);
  } else {
    // (a fresh output starts from the identity without being read -- and need not be pre-filled: 2 x N x K bytes less traffic)
    P cur;
    if (!fresh) cur = *reinterpret_cast<const P*>(op);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      acc[i] = fresh ? minmax_identity<T, OP>() : Math<T>::up(cur.v[i]);
      best[i] = s.E;
    }
  }
  if (live) {
    // four positions per trip: their loads are in flight together (a row is ~10 positions long in the sampler's
    // graphs and the loop is a chain of dependent memory round trips otherwise); accumulated in position order
#ifndef PYG_CSR_U
#define PYG_CSR_U 4
#endif
    constexpr int U = PYG_CSR_U;
    for (int64_t e0 = a + lane; e0 < b; e0 += (int64_t)U * L) {
      int64_t pp[U];
      P xx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t e = e0 + (int64_t)u * L;
        const int64_t ec = e < b ? e : e0;  // clamped: the load is issued unconditionally, the value is ignored below
        pp[u] = PERM ? perm[ec] : ec;
      }
      if constexpr (PERM) pin_all(pp);
#pragma unroll
      for (int u = 0; u < U; ++u) xx[u] = *reinterpret_cast<const P*>(sp + pp[u] * s.K);
      pin_all(xx);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (e0 + (int64_t)u * L >= b) break;
        const int64_t p = pp[u];
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const acc_t v = Math<T>::up(xx[u].v[i]);
          if constexpr (OP == CSR_SUM || OP == CSR_MEAN) {
            acc[i] += v;
          } else if constexpr (OP == CSR_MIN) {
            if (v < acc[i]) { acc[i] = v; best[i] = p; }
          } else {
            if (v > acc[i]) { acc[i] = v; best[i] = p; }
          }
        }
      }
    }
  }
  if constexpr (L > 1) {
#pragma unroll
    for (int m = L >> 1; m >= 1; m >>= 1) {
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const acc_t ov = shfl_xor_any(acc[i], m);
        if constexpr (OP == CSR_SUM || OP == CSR_MEAN) {
          acc[i] += ov;
        } else {
          const int64_t ob = shfl_xor_any(best[i], m);
          const bool better = OP == CSR_MIN ? ov < acc[i] : ov > acc[i];
          const bool worse = OP == CSR_MIN ? acc[i] < ov : acc[i] > ov;
          if (better || (!worse && ob < best[i])) { acc[i] = ov; best[i] = ob; }
        }
      }
    }
  }
  if (!live || lane != 0) return;
  P res;
  if constexpr (OP == CSR_MEAN) {
    const int64_t len = b - a;
    const acc_t denom = (acc_t)(len > 0 ? len : 1);
#pragma unroll
    for (int i = 0; i < V; ++i) res.v[i] = Math<T>::down(acc[i] / denom);
  } else if constexpr (OP == CSR_SUM) {
#pragma unroll
    for (int i = 0; i < V; ++i) res.v[i] = Math<T>::down(acc[i]);
  } else {
    // rows without a contribution of a fresh output read 0 (segment_csr_kernel.cpp:416-418)
#pragma unroll
    for (int i = 0; i < V; ++i) {
      res.v[i] = Math<T>::down(acc[i]);
      if (fresh && best[i] == s.E) res.v[i] = Math<T>::down(acc_t(0));
      arg[n * s.K + c + i] = best[i];
    }
  }
  *reinterpret_cast<P*>(op) = res;
}

// The hub kernels.  256 threads take positions [pa, pb) of one row as S 16-byte slices x 256 / S position lanes (lane j
// takes pa + j, pa + j + 256 / S, ..., eight loads in flight); the lanes' partial results are combined through LDS by a
// fixed pairwise tree: the same bits on every run; floating sums differ from the sequential order by rounding like the L > 1 variants
// above; min / max and their first-match arg stay exact.
template <typename T, int V>
struct HubGeom {
  int S, logS, EL, sl, lane;
  __device__ explicit HubGeom(int64_t kv) {
    S = 1, logS = 0;
    while (S < kv && S < 256) S <<= 1, ++logS;
    EL = 256 >> logS;                       // position lanes per slice
    sl = (int)threadIdx.x & (S - 1), lane = (int)threadIdx.x >> logS;
  }
};

// tot / tb of the threads with g.lane == 0 (and `on`): start (+) the span's positions; every thread of the workgroup calls it
template <typename T, int OP, int V, bool PERM>
__device__ __forceinline__ void hub_span(const T* __restrict__ sp, const int64_t* __restrict__ perm, int64_t pa, int64_t pb,
                                         int64_t rowK, bool on, const HubGeom<T, V>& g,
                                         const typename Math<T>::acc_t (&start)[V], typename Math<T>::acc_t* part,
                                         int64_t* part_best, typename Math<T>::acc_t (&tot)[V], int64_t (&tb)[V], int64_t E) {
  using acc_t = typename Math<T>::acc_t;
  using P = Pack<T, V>;
  constexpr bool MINMAX = OP == CSR_MIN || OP == CSR_MAX;
  acc_t acc[V];
  int64_t best[V];
#pragma unroll
  for (int i = 0; i < V; ++i) acc[i] = MINMAX ? start[i] : acc_t(0), best[i] = E;
  if (on) {
    constexpr int U = 8;
    for (int64_t e0 = pa + g.lane; e0 < pb; e0 += (int64_t)U * g.EL) {
      int64_t pp[U];
      P xx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t e = e0 + (int64_t)u * g.EL;
        const int64_t ec = e < pb ? e : e0;
        pp[u] = PERM ? perm[ec] : ec;
      }
      if constexpr (PERM) pin_all(pp);
#pragma unroll
      for (int u = 0; u < U; ++u) xx[u] = *reinterpret_cast<const P*>(sp + pp[u] * rowK);
      pin_all(xx);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (e0 + (int64_t)u * g.EL >= pb) break;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const acc_t v = Math<T>::up(xx[u].v[i]);
          if constexpr (!MINMAX) {
            acc[i] += v;
          } else if constexpr (OP == CSR_MIN) {
            if (v < acc[i]) { acc[i] = v; best[i] = pp[u]; }
          } else {
            if (v > acc[i]) { acc[i] = v; best[i] = pp[u]; }
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < V; ++i) {
    part[threadIdx.x * V + i] = acc[i];
    if constexpr (MINMAX) part_best[threadIdx.x * V + i] = best[i];
  }
  __syncthreads();
  // the lanes of a slice combine pairwise, lane l with lane l + stride (a fixed tree: the same bits on every run; one thread
  // walking 256 lanes x 8 values out of LDS took 60 us per chunk)
  for (int st = g.EL >> 1; st >= 1; st >>= 1) {
    if (g.lane < st) {
      const int ia = threadIdx.x * V, ib = (((g.lane + st) << g.logS) + g.sl) * V;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const acc_t ov = part[ib + i];
        if constexpr (!MINMAX) {
          part[ia + i] += ov;
        } else {
          const acc_t cv = part[ia + i];
          const int64_t ob = part_best[ib + i], cb = part_best[ia + i];
          const bool better = OP == CSR_MIN ? ov < cv : ov > cv;
          const bool worse = OP == CSR_MIN ? cv < ov : cv > ov;
          if (better || (!worse && ob < cb)) part[ia + i] = ov, part_best[ia + i] = ob;
        }
      }
    }
    __syncthreads();
  }
  if (on && g.lane == 0) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
      // (min / max: every lane started from `start`, it is inside the combined value already)
      tot[i] = MINMAX ? part[threadIdx.x * V + i] : start[i] + part[threadIdx.x * V + i];
      tb[i] = MINMAX ? part_best[threadIdx.x * V + i] : E;
    }
  }
  __syncthreads();
}

// the row's final value from (tot, tb) -> out / arg
template <typename T, int OP, int V>
__device__ __forceinline__ void hub_store(T* __restrict__ op, int64_t* __restrict__ ap, typename Math<T>::acc_t (&tot)[V],
                                          int64_t (&tb)[V], int64_t len, int fresh, int64_t E) {
  using acc_t = typename Math<T>::acc_t;
  constexpr bool MINMAX = OP == CSR_MIN || OP == CSR_MAX;
  Pack<T, V> res;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    if constexpr (OP == CSR_MEAN) tot[i] = tot[i] / (acc_t)len;
    res.v[i] = Math<T>::down(tot[i]);
    if constexpr (MINMAX) {
      if (fresh && tb[i] == E) res.v[i] = Math<T>::down(acc_t(0));
      ap[i] = tb[i];
    }
  }
  *reinterpret_cast<Pack<T, V>*>(op) = res;
}

// what a row's accumulation starts from: the caller's `out` slot (sum into an existing output; min / max always)
template <typename T, int OP, int V>
__device__ __forceinline__ void hub_seed(const T* __restrict__ op, bool on, int fresh, typename Math<T>::acc_t (&seed)[V]) {
  using acc_t = typename Math<T>::acc_t;
  constexpr bool MINMAX = OP == CSR_MIN || OP == CSR_MAX;
#pragma unroll
  for (int i = 0; i < V; ++i) seed[i] = MINMAX ? minmax_identity<T, MINMAX ? OP : CSR_MIN>() : acc_t(0);
  if (on && !fresh && (MINMAX || OP == CSR_SUM)) {
    const Pack<T, V> cur = *reinterpret_cast<const Pack<T, V>*>(op);
#pragma unroll
    for (int i = 0; i < V; ++i) seed[i] = Math<T>::up(cur.v[i]);
  }
}

// Without scratch: every workgroup looks at 256 rows at a time and takes the long ones among them, one after the other.
template <typename T, int OP, int V, bool PERM>
__global__ __launch_bounds__(256) void segment_csr_long_kernel(const T* __restrict__ src, const int64_t* __restrict__ indptr,
                                                               const int64_t* __restrict__ perm, T* __restrict__ out,
                                                               int64_t* __restrict__ arg, int fresh, CsrShape s) {
  using acc_t = typename Math<T>::acc_t;
  constexpr bool MINMAX = OP == CSR_MIN || OP == CSR_MAX;
  __shared__ int64_t long_rows[256];
  __shared__ int n_long;
  __shared__ acc_t part[256 * V];
  __shared__ int64_t part_best[MINMAX ? 256 * V : 1];
  const int64_t kv = s.K / V;
  const int64_t nrows = s.leading * s.rows;
  const HubGeom<T, V> g(kv);
  for (int64_t base = (int64_t)blockIdx.x * 256; base < nrows; base += (int64_t)gridDim.x * 256) {
    if (threadIdx.x == 0) n_long = 0;
    __syncthreads();
    {
      const int64_t n = base + threadIdx.x;
      if (n < nrows) {
        const int64_t slice = n / s.rows, row = n % s.rows;
        const int64_t a = indptr[slice * s.indptr_stride + row], b = indptr[slice * s.indptr_stride + row + 1];
        if (b - a > s.long_cut) long_rows[atomicAdd(&n_long, 1)] = n;   // (their order does not matter: rows are independent)
      }
    }
    __syncthreads();
    const int cnt = n_long;
    for (int j = 0; j < cnt; ++j) {
      const int64_t n = long_rows[j];
      const int64_t slice = n / s.rows, row = n % s.rows;
      const int64_t a = indptr[slice * s.indptr_stride + row], b = indptr[slice * s.indptr_stride + row + 1];
      for (int64_t c0 = 0; c0 < kv; c0 += g.S) {
        const int64_t ci = c0 + g.sl;
        const bool on = ci < kv;
        const int64_t c = (on ? ci : 0) * V;
        T* op = out + n * s.K + c;
        acc_t seed[V], tot[V];
        int64_t tb[V];
        hub_seed<T, OP, V>(op, on, fresh, seed);
        hub_span<T, OP, V, PERM>(src + slice * s.E * s.K + c, perm, a, b, s.K, on, g, seed, part, part_best, tot, tb, s.E);
        if (on && g.lane == 0) hub_store<T, OP, V>(op, MINMAX ? arg + n * s.K + c : nullptr, tot, tb, b - a, fresh, s.E);
      }
    }
    __syncthreads();
  }
}

// With scratch: one registered chunk per workgroup and trip.  A hub of one chunk is finished on the spot; the chunks of a
// longer one leave their results in the hub's slots ...
template <typename T, int OP, int V, bool PERM>
__global__ __launch_bounds__(256) void segment_csr_hub_chunk_kernel(const T* __restrict__ src, const int64_t* __restrict__ indptr,
                                                                    const int64_t* __restrict__ perm, T* __restrict__ out,
                                                                    int64_t* __restrict__ arg, int fresh, CsrShape s, HubWs hw) {
  using acc_t = typename Math<T>::acc_t;
  constexpr bool MINMAX = OP == CSR_MIN || OP == CSR_MAX;
  __shared__ acc_t part[256 * V];
  __shared__ int64_t part_best[MINMAX ? 256 * V : 1];
  const int64_t kv = s.K / V;
  const HubGeom<T, V> g(kv);
  const int nchunks = hw.counters[3] ? 0 : hw.counters[1];
  for (int q = blockIdx.x; q < nchunks; q += gridDim.x) {
    const int2 cr = hw.chunks[q];
    const HubRec hub = hw.hubs[cr.x];
    const int64_t n = hub.n;
    const int64_t slice = n / s.rows, row = n % s.rows;
    const int64_t a = indptr[slice * s.indptr_stride + row], b = indptr[slice * s.indptr_stride + row + 1];
    const int64_t pa = a + (int64_t)cr.y * hw.CH, pb = pa + hw.CH < b ? pa + hw.CH : b;
    const bool whole = hub.nch == 1;
    const int64_t slot = (int64_t)(hub.slot_base + cr.y) * s.K;
    for (int64_t c0 = 0; c0 < kv; c0 += g.S) {
      const int64_t ci = c0 + g.sl;
      const bool on = ci < kv;
      const int64_t c = (on ? ci : 0) * V;
      T* op = out + n * s.K + c;
      acc_t seed[V], tot[V];
      int64_t tb[V];
      hub_seed<T, OP, V>(op, on && (whole || MINMAX), fresh, seed);   // (the sum of a chunk starts from 0)
      hub_span<T, OP, V, PERM>(src + slice * s.E * s.K + c, perm, pa, pb, s.K, on, g, seed, part, part_best, tot, tb, s.E);
      if (on && g.lane == 0) {
        if (whole) {
          hub_store<T, OP, V>(op, MINMAX ? arg + n * s.K + c : nullptr, tot, tb, b - a, fresh, s.E);
        } else {
#pragma unroll
          for (int i = 0; i < V; ++i) {
            reinterpret_cast<acc_t*>(hw.partial)[slot + c + i] = tot[i];
            if constexpr (MINMAX) hw.partial_best[slot + c + i] = tb[i];
          }
        }
      }
    }
  }
}

// ... and a third launch adds a longer hub's slots up in chunk order: one thread per (hub, 16-byte slice).  (One launch for
// both -- the workgroup that writes a hub's last slot combines them -- needs device-scope fences around the count: each
// writes back and invalidates the XCD's whole L2, 4000 chunks took 650 us that way instead of 60.)
template <typename T, int OP, int V>
__global__ __launch_bounds__(256) void segment_csr_hub_combine_kernel(const int64_t* __restrict__ indptr, T* __restrict__ out,
                                                                      int64_t* __restrict__ arg, int fresh, CsrShape s, HubWs hw) {
  using acc_t = typename Math<T>::acc_t;
  constexpr bool MINMAX = OP == CSR_MIN || OP == CSR_MAX;
  const int64_t kv = s.K / V;
  const int64_t items = hw.counters[3] ? 0 : (int64_t)hw.counters[0] * kv;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < items; t += (int64_t)gridDim.x * blockDim.x) {
    const HubRec hub = hw.hubs[t / kv];
    if (hub.nch == 1) continue;
    const int64_t c = (t % kv) * V;
    const int64_t n = hub.n;
    const int64_t slice = n / s.rows, row = n % s.rows;
    const int64_t len = indptr[slice * s.indptr_stride + row + 1] - indptr[slice * s.indptr_stride + row];
    T* op = out + n * s.K + c;
    acc_t tot[V];
    int64_t tb[V];
    hub_seed<T, OP, V>(op, true, fresh, tot);
#pragma unroll
    for (int i = 0; i < V; ++i) tb[i] = s.E;
    // chunk order, eight slots' values in flight (a 500 000-position hub has 245 slots: 245 dependent trips took 90 us)
    constexpr int U = 8;
    for (int j0 = 0; j0 < hub.nch; j0 += U) {
      Pack<acc_t, V> ov[U];
      Pack<int64_t, V> ob[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = j0 + u < hub.nch ? j0 + u : j0;
        const int64_t sj = (int64_t)(hub.slot_base + j) * s.K + c;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          ov[u].v[i] = reinterpret_cast<const acc_t*>(hw.partial)[sj + i];
          if constexpr (MINMAX) ob[u].v[i] = hw.partial_best[sj + i];
        }
      }
      pin_all(ov);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (j0 + u >= hub.nch) break;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          if constexpr (!MINMAX) {
            tot[i] += ov[u].v[i];
          } else {
            const bool better = OP == CSR_MIN ? ov[u].v[i] < tot[i] : ov[u].v[i] > tot[i];
            const bool worse = OP == CSR_MIN ? tot[i] < ov[u].v[i] : tot[i] > ov[u].v[i];
            if (better || (!worse && ob[u].v[i] < tb[i])) tot[i] = ov[u].v[i], tb[i] = ob[u].v[i];
          }
        }
      }
    }
    hub_store<T, OP, V>(op, MINMAX ? arg + n * s.K + c : nullptr, tot, tb, len, fresh, s.E);
  }
}

// out[slice, e, :] = src[slice, r, :] for every position e of row r: the mirror image of the sum kernel
// (one thread per (row, 16-byte slice), L lanes per item for long rows).  Positions covered by no row
// are never written.
template <typename T, int V, int L>
__global__ __launch_bounds__(256) void gather_csr_kernel(const T* __restrict__ src, const int64_t* __restrict__ indptr,
                                                         T* __restrict__ out, CsrShape s, HubWs hw) {
  using P = Pack<T, V>;
  const int64_t kv = s.K / V;
  const int64_t t = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / L;
  const int lane = threadIdx.x & (L - 1);
  if (t >= s.leading * s.rows * kv) return;
  const int64_t n = t / kv;
  const int64_t c = (t % kv) * V;
  const int64_t slice = n / s.rows, row = n % s.rows;
  const int64_t a = indptr[slice * s.indptr_stride + row];
  const int64_t b = indptr[slice * s.indptr_stride + row + 1];
  if (b - a > s.long_cut) {   // a hub row: gather_csr_hub_chunk_kernel / gather_csr_long_kernel
    if (hw.counters && lane == 0 && c == 0) hub_register(hw, n, b - a);
    return;
  }
  const P v = *reinterpret_cast<const P*>(src + n * s.K + c);
  T* op = out + slice * s.E * s.K + c;
  for (int64_t e = a + lane; e < b; e += L) *reinterpret_cast<P*>(op + e * s.K) = v;
}

// `len` positions of one row written by 256 threads: 16-byte (or one-element) pieces in memory order
template <typename T, int V>
__device__ __forceinline__ void gather_span(const T* __restrict__ sp, T* __restrict__ op, int64_t len, int64_t kv) {
  using P = Pack<T, V>;
  const int64_t total = len * kv;
  if (kv <= 256 && 256 % kv == 0) {   // a thread's piece column never changes
    const P v = *reinterpret_cast<const P*>(sp + ((int64_t)threadIdx.x % kv) * V);
    for (int64_t i = threadIdx.x; i < total; i += 256) *reinterpret_cast<P*>(op + i * V) = v;
  } else {
    for (int64_t i = threadIdx.x; i < total; i += 256)
      *reinterpret_cast<P*>(op + i * V) = *reinterpret_cast<const P*>(sp + (i % kv) * V);
  }
}

// The hub rows of the kernel above (more than `long_cut` positions), a whole workgroup per row: the row's K values are
// written to consecutive positions by consecutive threads (full lines), the workgroups scan the row lengths 256 at a time
// like segment_csr_long_kernel.
template <typename T, int V>
__global__ __launch_bounds__(256) void gather_csr_long_kernel(const T* __restrict__ src, const int64_t* __restrict__ indptr,
                                                              T* __restrict__ out, CsrShape s) {
  __shared__ int64_t long_rows[256];
  __shared__ int n_long;
  const int64_t kv = s.K / V;
  const int64_t nrows = s.leading * s.rows;
  for (int64_t base = (int64_t)blockIdx.x * 256; base < nrows; base += (int64_t)gridDim.x * 256) {
    if (threadIdx.x == 0) n_long = 0;
    __syncthreads();
    {
      const int64_t n = base + threadIdx.x;
      if (n < nrows) {
        const int64_t slice = n / s.rows, row = n % s.rows;
        const int64_t a = indptr[slice * s.indptr_stride + row], b = indptr[slice * s.indptr_stride + row + 1];
        if (b - a > s.long_cut) long_rows[atomicAdd(&n_long, 1)] = n;
      }
    }
    __syncthreads();
    const int cnt = n_long;
    for (int j = 0; j < cnt; ++j) {
      const int64_t n = long_rows[j];
      const int64_t slice = n / s.rows, row = n % s.rows;
      const int64_t a = indptr[slice * s.indptr_stride + row], b = indptr[slice * s.indptr_stride + row + 1];
      gather_span<T, V>(src + n * s.K, out + (slice * s.E + a) * s.K, b - a, kv);
    }
    __syncthreads();
  }
}

// with scratch: one registered chunk of a hub row per workgroup and trip
template <typename T, int V>
__global__ __launch_bounds__(256) void gather_csr_hub_chunk_kernel(const T* __restrict__ src, const int64_t* __restrict__ indptr,
                                                                   T* __restrict__ out, CsrShape s, HubWs hw) {
  const int64_t kv = s.K / V;
  const int nchunks = hw.counters[3] ? 0 : hw.counters[1];
  for (int q = blockIdx.x; q < nchunks; q += gridDim.x) {
    const int2 cr = hw.chunks[q];
    const int64_t n = hw.hubs[cr.x].n;
    const int64_t slice = n / s.rows, row = n % s.rows;
    const int64_t a = indptr[slice * s.indptr_stride + row], b = indptr[slice * s.indptr_stride + row + 1];
    const int64_t pa = a + (int64_t)cr.y * hw.CH, pb = pa + hw.CH < b ? pa + hw.CH : b;
    gather_span<T, V>(src + n * s.K, out + (slice * s.E + pa) * s.K, pb - pa, kv);
  }
}

// ---- softmax over CSR groups (float / double) ------------------------------------------------------
// src [outer, D, inner]; one item per (group, outer, inner) "head".  L lanes per item.
template <typename T, int L, bool BACKWARD>
__global__ __launch_bounds__(256) void softmax_csr_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          const int64_t* __restrict__ ptr, T* __restrict__ y,
                                                          int64_t outer, int64_t D, int64_t inner, int64_t groups,
                                                          int64_t long_cut) {
  const int64_t t = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / L;
  const int lane = threadIdx.x & (L - 1);
  const bool live = t < groups * outer * inner;
  const int64_t tt = live ? t : 0;
  const int64_t sidx = tt % inner;
  const int64_t i = (tt / inner) % outer;
  const int64_t g = tt / (inner * outer);
  const int64_t a = ptr[g], b = ptr[g + 1];
  if (b - a > long_cut) return;   // (all L lanes of the item alike) a hub group: softmax_csr_long_kernel
  const int64_t base = i * D * inner + sidx;
  auto reduce = [&](T v, bool is_max) {
#pragma unroll
    for (int m = L >> 1; m >= 1; m >>= 1) {
      const T o = shfl_xor_any(v, m);
      v = is_max ? (v < o ? o : v) : v + o;
    }
    return v;
  };
  if constexpr (L == 1) {
    // groups of up to 32 positions: the head's values stay in registers between the passes -- one read and one write of the
    // data instead of two or three reads and two writes (same operations in the same order: the same bits).  Three sizes, so
    // that a group of two does not issue 32 loads.
    const int64_t len = b - a;
    auto in_registers = [&](auto rc) __attribute__((always_inline)) {
      constexpr int R = decltype(rc)::value;
      T xv[R], dv[R];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int64_t p = a + (u < len ? u : 0);
        xv[u] = x[base + p * inner];
        dv[u] = BACKWARD ? dy[base + p * inner] : T(0);
      }
      if constexpr (R <= 16) {
        pin_all(xv);
        if constexpr (BACKWARD) pin_all(dv);
      } else {
        T(&x0)[16] = reinterpret_cast<T(&)[16]>(xv[0]);
        T(&x1)[16] = reinterpret_cast<T(&)[16]>(xv[16]);
        pin_all(x0), pin_all(x1);
        if constexpr (BACKWARD) {
          T(&d0)[16] = reinterpret_cast<T(&)[16]>(dv[0]);
          T(&d1)[16] = reinterpret_cast<T(&)[16]>(dv[16]);
          pin_all(d0), pin_all(d1);
        }
      }
      if constexpr (BACKWARD) {
        T sum = 0;
#pragma unroll
        for (int u = 0; u < R; ++u)
          if (u < len) sum += xv[u] * dv[u];
#pragma unroll
        for (int u = 0; u < R; ++u)
          if (u < len) y[base + (a + u) * inner] = xv[u] * (dv[u] - sum);
      } else {
        T mx = type_lowest<T>();
#pragma unroll
        for (int u = 0; u < R; ++u)
          if (u < len) mx = mx < xv[u] ? xv[u] : mx;
        T sum = 0;
#pragma unroll
        for (int u = 0; u < R; ++u)
          if (u < len) {
            xv[u] = exp(xv[u] - mx);
            sum += xv[u];
          }
#pragma unroll
        for (int u = 0; u < R; ++u)
          if (u < len) y[base + (a + u) * inner] = xv[u] / sum;
      }
    };
    // (backward: two values per position -- 32 positions cost the kernel its occupancy: 2 per group 0.39 -> 0.55 ms)
    constexpr int RMAX = BACKWARD ? 16 : 32;
    if (live && len > 1 && len <= RMAX) {
      if (len <= 4) in_registers(std::integral_constant<int, 4>{});
      else if (len <= 16) in_registers(std::integral_constant<int, 16>{});
      else if constexpr (RMAX > 16) in_registers(std::integral_constant<int, 32>{});
      return;
    }
  }
  if constexpr (BACKWARD) {
    // x = out, dy = out_grad, y = in_grad (softmax_kernel.cpp:148-222)
    T sum = 0;
    if (live)
      for (int64_t p = a + lane; p < b; p += L) sum += x[base + p * inner] * dy[base + p * inner];
    if (L > 1) sum = reduce(sum, false);
    if (live)
      for (int64_t p = a + lane; p < b; p += L)
        y[base + p * inner] = x[base + p * inner] * (dy[base + p * inner] - sum);
  } else {
    if (live && b - a == 1) {  // single-element groups are exactly 1 (softmax_kernel.cpp:102-109)
      if (lane == 0) y[base + a * inner] = T(1);
      // (falls through the reductions below with an empty range)
    }
    const bool work = live && b - a != 1;
    // four positions per trip, their loads pinned together (a head's positions are a row apart: one load in flight per thread
    // left long groups of wide heads at 0.17 of HBM)
    auto walk = [&](auto&& body) __attribute__((always_inline)) {
      if (!work) return;
      if constexpr (L > 1) {   // (lanes share a group: a lane has few positions, a batch would be mostly clamped loads)
        for (int64_t p = a + lane; p < b; p += L) body(p, x[base + p * inner]);
        return;
      }
      constexpr int U = 4;
      for (int64_t p0 = a + lane; p0 < b; p0 += (int64_t)U * L) {
        T xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t p = p0 + (int64_t)u * L;
          xv[u] = x[base + (p < b ? p : p0) * inner];
        }
        pin_all(xv);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t p = p0 + (int64_t)u * L;
          if (p >= b) break;
          body(p, xv[u]);
        }
      }
    };
    T mx = type_lowest<T>();
    walk([&](int64_t, T v) { mx = mx < v ? v : mx; });  // std::max(max, v)
    if (L > 1) mx = reduce(mx, true);
    // (the exponentials are computed again in the last pass instead of being parked in `y`: x is read three times and y
    // written once, instead of two reads of x and two writes + one read of y)
    T sum = 0;
    walk([&](int64_t, T v) { sum += exp(v - mx); });
    if (L > 1) sum = reduce(sum, false);
    walk([&](int64_t p, T v) { y[base + p * inner] = exp(v - mx) / sum; });
  }
}

// Hub groups (more than `long_cut` positions) of both softmax kernels, one workgroup per group: the workgroups look at 256
// groups at a time; a hub's [positions, inner] block is contiguous per outer index, its columns are taken 256 at a time by
// 256 / width position lanes each, whose partial maxima / sums are combined through LDS in lane order (sums differ from the
// sequential order by rounding, like the L > 1 variants).
template <typename T, bool BACKWARD>
__global__ __launch_bounds__(256) void softmax_csr_long_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                               const int64_t* __restrict__ ptr, T* __restrict__ y,
                                                               int64_t outer, int64_t D, int64_t inner, int64_t groups,
                                                               int64_t long_cut) {
  __shared__ int64_t long_groups[256];
  __shared__ int n_long;
  __shared__ T part[256];
  __shared__ T colv[256];
  for (int64_t g0 = (int64_t)blockIdx.x * 256; g0 < groups; g0 += (int64_t)gridDim.x * 256) {
    if (threadIdx.x == 0) n_long = 0;
    __syncthreads();
    {
      const int64_t g = g0 + threadIdx.x;
      if (g < groups && ptr[g + 1] - ptr[g] > long_cut) long_groups[atomicAdd(&n_long, 1)] = g;
    }
    __syncthreads();
    const int cnt = n_long;
    for (int j = 0; j < cnt; ++j) {
      const int64_t g = long_groups[j];
      const int64_t a = ptr[g], b = ptr[g + 1];
      for (int64_t i = 0; i < outer; ++i) {
        for (int64_t s0 = 0; s0 < inner; s0 += 256) {
          const int width = (int)(inner - s0 < 256 ? inner - s0 : 256);
          const int EL = 256 / width;
          const int lane = (int)threadIdx.x / width, col = (int)threadIdx.x % width;
          const bool on = lane < EL;
          const int64_t base = i * D * inner + s0 + col;
          // column-wise reduction of the lanes' values: the result is in colv[col] for everybody
          auto combine = [&](T v, bool is_max) {
            part[threadIdx.x] = v;
            __syncthreads();
            if (on && lane == 0) {
              T r = v;
              for (int l = 1; l < EL; ++l) {
                const T o = part[l * width + col];
                r = is_max ? (r < o ? o : r) : r + o;
              }
              colv[col] = r;
            }
            __syncthreads();
            const T r = colv[on ? col : 0];
            __syncthreads();
            return r;
          };
          // 16 positions per trip, their loads in flight together (pin_all): one workgroup has the whole group to itself
          constexpr int U = 16;
          auto walk = [&](auto&& body) {
            if (!on) return;
            for (int64_t p0 = a + lane; p0 < b; p0 += (int64_t)U * EL) {
              T xv[U], dv[U];
#pragma unroll
              for (int u = 0; u < U; ++u) {
                const int64_t p = p0 + (int64_t)u * EL;
                const int64_t pc = p < b ? p : p0;
                xv[u] = x[base + pc * inner];
                dv[u] = BACKWARD ? dy[base + pc * inner] : T(0);
              }
              pin_all(xv);
              if constexpr (BACKWARD) pin_all(dv);
#pragma unroll
              for (int u = 0; u < U; ++u) {
                const int64_t p = p0 + (int64_t)u * EL;
                if (p >= b) break;
                body(p, xv[u], dv[u]);
              }
            }
          };
          if constexpr (BACKWARD) {
            T sum = 0;
            walk([&](int64_t, T xv, T dv) { sum += xv * dv; });
            sum = combine(sum, false);
            walk([&](int64_t p, T xv, T dv) { y[base + p * inner] = xv * (dv - sum); });
          } else {
            T mx = type_lowest<T>();
            walk([&](int64_t, T xv, T) { mx = mx < xv ? xv : mx; });
            mx = combine(mx, true);
            T sum = 0;
            walk([&](int64_t, T xv, T) { sum += exp(xv - mx); });
            sum = combine(sum, false);
            walk([&](int64_t p, T xv, T) { y[base + p * inner] = exp(xv - mx) / sum; });
          }
        }
      }
    }
    __syncthreads();
  }
}

// ---- small K: rows streamed through LDS ----------------------------------------------------------
// With K * sizeof(T) below a cache line, one thread per (row, column) reads 2-4 byte elements that lie
// a whole row apart from its neighbour's (0.35 TB/s at K = 1).  Here a workgroup owns 256 / K consecutive
// rows: their source range is one contiguous span, which is copied to LDS in chunks with fully coalesced
// loads; every thread then walks ITS row's part of the chunk in source order -- same operation order
// as the CPU kernel (bit-exact), HBM traffic = the span once.
// Hub rows (more than `long_cut` positions) are not streamed: the chunk loop jumps over their spans, their threads stand
// by, and the hub kernels take them (a thread walking 200 000 positions out of LDS: 7 ms of a 0.02 ms call).
struct StreamHubs {
  int64_t a[16], b[16];     // spans of this workgroup's (first 16) hub rows, in row order
  unsigned char flag[256];  // per row of the workgroup: longer than the cut
  int n;
};                          // (516 bytes: with kStreamValues below, five workgroups' LDS still fit a CU)
// every thread of the workgroup; `hub`: this thread's row is one (threads of the same row alike), `first`: its k == 0 thread.
// Returns the number of spans listed.  (The span of a 17th hub among one workgroup's rows is streamed like the short rows'
// and ignored.)  Only called when the workgroup's whole span is longer than the cut: most workgroups never pay its barriers.
__device__ __forceinline__ int stream_hubs_collect(StreamHubs& h, bool hub, bool first, int rl, int rpb, const int64_t* ip,
                                                   int64_t r0) {
  if (!__syncthreads_or(hub)) return 0;
  h.flag[threadIdx.x] = 0;
  __syncthreads();
  if (hub && first) h.flag[rl] = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    int n = 0;
    for (int r = 0; r < rpb && n < 16; ++r)
      if (h.flag[r]) h.a[n] = ip[r0 + r], h.b[n] = ip[r0 + r + 1], ++n;
    h.n = n;
  }
  __syncthreads();
  return h.n;
}
// the next chunk [base, base + ce) of [.., b0) that touches none of the nh listed hub spans; false at the end (uniform over
// the workgroup)
__device__ __forceinline__ bool stream_next(const StreamHubs& h, int nh, int& hi, int64_t& base, int64_t& ce, int64_t b0,
                                            int64_t CE) {
  while (hi < nh && h.a[hi] <= base) {
    if (h.b[hi] > base) base = h.b[hi];
    ++hi;
  }
  if (base >= b0) return false;
  const int64_t lim = hi < nh ? h.a[hi] : b0;
  ce = lim - base < CE ? lim - base : CE;
  return true;
}

constexpr int kStreamValues = 8192 - 192;   // LDS values per chunk (32 KB of fp32 less what StreamHubs takes)
constexpr int kSoftmaxValues = 16384;  // softmax: bigger chunks so that typical spans need ONE pass over HBM
// (float64 backward: two images of 16384 doubles are 256 KB -- more LDS than a CU has; half the values)
template <typename T, bool BACKWARD>
constexpr int softmax_values() { return sizeof(T) == 8 && BACKWARD ? kSoftmaxValues / 2 : kSoftmaxValues; }

template <typename T, int OP>
__global__ __launch_bounds__(256) void segment_csr_stream_kernel(const T* __restrict__ src,
                                                                 const int64_t* __restrict__ indptr,
                                                                 T* __restrict__ out, int64_t* __restrict__ arg,
                                                                 int fresh, CsrShape s, int rpb, HubWs hw) {
  using acc_t = typename Math<T>::acc_t;
  __shared__ T buf[kStreamValues];
  __shared__ StreamHubs hubs;
  const int K = (int)s.K;
  const int rl = threadIdx.x / K, k = threadIdx.x % K;
  const int64_t bps = (s.rows + rpb - 1) / rpb;
  const int64_t slice = blockIdx.x / bps;
  const int64_t r0 = (blockIdx.x % bps) * rpb;
  const int64_t rN = r0 + rpb < s.rows ? r0 + rpb : s.rows;
  const int64_t* ip = indptr + slice * s.indptr_stride;
  const int64_t a0 = ip[r0], b0 = ip[rN];
  const int64_t row = r0 + rl;
  bool valid = rl < rpb && row < rN;
  int64_t a = valid ? ip[row] : 0, b = valid ? ip[row + 1] : 0;
  const int64_t n = slice * s.rows + row;
  int nh = 0;
  {
    const bool hub = valid && b - a > s.long_cut;
    if (b0 - a0 > s.long_cut) nh = stream_hubs_collect(hubs, hub, k == 0, rl, rpb, ip, r0);
    if (hub) {
      if (hw.counters && k == 0) hub_register(hw, n, b - a);
      valid = false, a = b = 0;
    }
  }
  acc_t acc = acc_t(0);
  int64_t best = s.E;
  if constexpr (OP == CSR_MIN || OP == CSR_MAX) acc = minmax_identity<T, OP>();
  if (valid && OP != CSR_MEAN && !fresh) acc = Math<T>::up(out[n * K + k]);
  const int64_t CE = kStreamValues / K;
  const T* sp = src + slice * s.E * K;
  int hi_ = 0;
  int64_t ce;
  for (int64_t base = a0; stream_next(hubs, nh, hi_, base, ce, b0, CE); base += ce) {
    const int64_t nv = ce * K;
    for (int64_t i = threadIdx.x; i < nv; i += 256) buf[i] = sp[base * K + i];
    __syncthreads();
    const int64_t lo = a > base ? a : base, hi = b < base + ce ? b : base + ce;
    for (int64_t e = lo; e < hi; ++e) {
      const acc_t v = Math<T>::up(buf[(e - base) * K + k]);
      if constexpr (OP == CSR_SUM || OP == CSR_MEAN) acc += v;
      else if constexpr (OP == CSR_MIN) { if (v < acc) { acc = v; best = e; } }
      else { if (v > acc) { acc = v; best = e; } }
    }
    __syncthreads();
  }
  if (!valid) return;
  if constexpr (OP == CSR_MEAN) {
    const int64_t len = b - a;
    out[n * K + k] = Math<T>::down(acc / (acc_t)(len > 0 ? len : 1));
  } else if constexpr (OP == CSR_SUM) {
    out[n * K + k] = Math<T>::down(acc);
  } else {
    out[n * K + k] = (fresh && best == s.E) ? Math<T>::down(acc_t(0)) : Math<T>::down(acc);
    arg[n * K + k] = best;
  }
}

// softmax over groups with small inner size, streamed the same way: when a workgroup's span fits one
// chunk the three passes (max, exp-sum, normalise) run from LDS -- one HBM read, one write; longer
// spans stream three times.  Per-head operation order = the CPU kernel's.
template <typename T, bool BACKWARD>
__global__ __launch_bounds__(256) void softmax_csr_stream_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                 const int64_t* __restrict__ ptr, T* __restrict__ y,
                                                                 CsrShape s, int rpb) {
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];
  __shared__ StreamHubs hubs;
  T* buf = reinterpret_cast<T*>(sm_raw);
  T* buf2 = buf + softmax_values<T, BACKWARD>();  // backward only
  const int K = (int)s.K;
  const int rl = threadIdx.x / K, k = threadIdx.x % K;
  const int64_t bps = (s.rows + rpb - 1) / rpb;
  const int64_t slice = blockIdx.x / bps;
  const int64_t r0 = (blockIdx.x % bps) * rpb;
  const int64_t rN = r0 + rpb < s.rows ? r0 + rpb : s.rows;
  const int64_t a0 = ptr[r0], b0 = ptr[rN];
  const int64_t row = r0 + rl;
  const bool valid = rl < rpb && row < rN;
  int64_t a = valid ? ptr[row] : 0, b = valid ? ptr[row + 1] : 0;
  int nh = 0;
  {
    const bool hub = valid && b - a > s.long_cut;   // softmax_csr_long_kernel's
    if (b0 - a0 > s.long_cut) nh = stream_hubs_collect(hubs, hub, k == 0, rl, rpb, ptr, r0);
    if (hub) a = b = 0;
  }
  const int64_t CE = softmax_values<T, BACKWARD>() / K;
  const T* xp = x + slice * s.E * K;
  const T* dp = BACKWARD ? dy + slice * s.E * K : nullptr;
  T* yp = y + slice * s.E * K;
  const bool single = b0 - a0 <= CE && nh == 0;
  const bool one = !BACKWARD && b - a == 1;  // single-element groups are exactly 1
  T mx = type_lowest<T>(), sum = T(0);
  // pass p: 0 = max (forward only), 1 = sum, 2 = write
  for (int pass = BACKWARD ? 1 : 0; pass < 3; ++pass) {
    int hi_ = 0;
    int64_t ce;
    for (int64_t base = a0; stream_next(hubs, nh, hi_, base, ce, b0, CE); base += ce) {
      const int64_t nv = ce * K;
      if (!single || pass == (BACKWARD ? 1 : 0)) {
        __syncthreads();
        for (int64_t i = threadIdx.x; i < nv; i += 256) {
          buf[i] = xp[base * K + i];
          if (BACKWARD) buf2[i] = dp[base * K + i];
        }
        __syncthreads();
      }
      const int64_t lo = a > base ? a : base, hi = b < base + ce ? b : base + ce;
      for (int64_t e = lo; e < hi; ++e) {
        const int64_t j = (e - base) * K + k;
        if constexpr (BACKWARD) {
          if (pass == 1) sum += buf[j] * buf2[j];
          else if (pass == 2) yp[e * K + k] = buf[j] * (buf2[j] - sum);
        } else {
          if (pass == 0) { const T v = buf[j]; mx = mx < v ? v : mx; }
          else if (pass == 1) { if (!one) sum += exp(buf[j] - mx); }
          else yp[e * K + k] = one ? T(1) : exp(buf[j] - mx) / sum;
        }
      }
    }
  }
}

// ---- hub scratch ------------------------------------------------------------------------------------
constexpr int64_t kHubCut = 512;      // positions per lane of the row kernels above which a row is a hub
constexpr int64_t kHubCutStream = 4096;  // ... of the LDS-streamed kernels (a thread walks its row out of LDS: ~20 us at the cut)
constexpr int64_t kHubChunk = 2048;   // positions per chunk (doubled until the scratch holds the partial results)

inline size_t hub_align(size_t b) { return (b + 255) & ~(size_t)255; }

// Lays the hub scratch out for `total` positions in rows of K values.  More than total / kHubCut hubs cannot exist; a hub
// of one chunk needs no slot, a longer one at most 2 * len / CH of them.  Returns the bytes used (0: disabled).
inline size_t hub_plan(void* ws, size_t ws_bytes, int64_t total, int64_t K, size_t acc_bytes, bool minmax, HubWs* hw,
                       int64_t* max_chunks_out) {
  if (total <= kHubCut) return 0;
  const int64_t max_hubs = total / kHubCut + 1;
  for (int64_t CH = kHubChunk; CH <= (1ll << 22); CH <<= 1) {
    const int64_t max_chunks = total / CH + max_hubs + 1;
    const int64_t max_slots = acc_bytes ? 2 * (total / CH) + 2 : 0;
    const size_t o_hubs = 256;
    const size_t o_chunks = o_hubs + hub_align(sizeof(HubRec) * (size_t)max_hubs);
    const size_t o_part = o_chunks + hub_align(sizeof(int2) * (size_t)max_chunks);
    const size_t o_best = o_part + hub_align(acc_bytes * (size_t)max_slots * (size_t)K);
    const size_t end = o_best + (minmax ? hub_align(sizeof(int64_t) * (size_t)max_slots * (size_t)K) : 0);
    if (!hw) return end;   // sizing: the smallest chunk
    const size_t skew = (256 - (reinterpret_cast<uintptr_t>(ws) & 255)) & 255;
    if (ws && ws_bytes >= end + skew) {
      char* w = static_cast<char*>(ws) + skew;
      hw->counters = reinterpret_cast<int*>(w);
      hw->hubs = reinterpret_cast<HubRec*>(w + o_hubs);
      hw->chunks = reinterpret_cast<int2*>(w + o_chunks);
      hw->partial = w + o_part;
      hw->partial_best = reinterpret_cast<int64_t*>(w + o_best);
      hw->CH = CH;
      const int64_t cap = 0x7fffffff;
      hw->max_hubs = (int)std::min(max_hubs, cap), hw->max_chunks = (int)std::min(max_chunks, cap);
      hw->max_slots = (int)std::min(max_slots, cap);
      *max_chunks_out = max_chunks;
      return end + skew;
    }
    if (!acc_bytes) break;   // (gather: nothing shrinks with longer chunks but the chunk list)
  }
  return 0;
}

// (narrow rows of whole 16-byte slices: 8 lanes per item from 16 positions per row on -- a load instruction then covers 8
// consecutive positions of a row: fp32 K = 4, 48 per row: 0.096 (streamed) -> 0.044 ms; bf16 K = 8, 16 per row: 0.103 -> 0.062;
// from 8 per row on it loses: fp32 K = 12: 0.19 -> 0.27.  `tools/lease/ab_narrow_vec_lanes.sh`)
#ifndef PYG_CSR_NARROW_VEC_LANES_AVG
#define PYG_CSR_NARROW_VEC_LANES_AVG 16
#endif
constexpr int64_t kNarrowVecLanesAvg = PYG_CSR_NARROW_VEC_LANES_AVG;
// lanes per item: long rows + too few items to fill the chip
int pick_lanes(int64_t items, int64_t total_len, int64_t units, int64_t row_bytes = 64) {
  if (units <= 0 || items <= 0) return 1;
  const int64_t avg = total_len / units;
  // rows narrower than a cache line that are too long for the LDS-streamed kernels: the lanes of an item read neighbouring
  // positions -- one contiguous piece per trip, whatever the number of items (K = 1, 300 positions per row: 0.168 -> ms)
  if (row_bytes < 64 && avg >= 64) return avg >= 256 ? 64 : 8;
  if (row_bytes < 64 && row_bytes % 16 == 0 && avg >= kNarrowVecLanesAvg) return 8;
  const int64_t chip = (int64_t)device_info().num_cus * 2048;  // resident threads
  if (avg >= 1024 && items * 8 < chip) return 64;
  if (avg >= 64 && items < chip) return 8;
  return 1;
}

// The launches around a row kernel that skips hub rows: hub_begin() in front of it (cut, scratch layout, counters), then the
// row kernel with st.sc / st.hw, then hub_end().
struct HubStage {
  CsrShape sc;
  HubWs hw;
  int64_t max_chunks = 0, max_hubs = 0;
  bool hubs = false;
};
template <typename T, int OP>
int hub_begin(const CsrShape& s, int64_t cut, HubStage& st, hipStream_t stream) {
  using acc_t = typename Math<T>::acc_t;
  st.sc = s;
  st.sc.hub_ws = nullptr, st.sc.hub_ws_bytes = 0;
  st.hubs = s.E > cut && s.leading * s.rows > 1;   // (else no row can be that long: nothing is launched for them)
  if (!st.hubs) return PYG_HIP_OK;
  st.sc.long_cut = cut;
  if (hub_plan(s.hub_ws, s.hub_ws_bytes, s.leading * s.E, s.K, sizeof(acc_t), OP == CSR_MIN || OP == CSR_MAX, &st.hw, &st.max_chunks))
    PYG_HIP_CHECK(hipMemsetAsync(st.hw.counters, 0, 16, stream));
  st.max_hubs = s.leading * s.E / cut;   // every hub is longer than the cut
  return PYG_HIP_OK;
}
template <typename T, int OP, int V, bool PERM>
int hub_end(const T* sp, const int64_t* indptr, const int64_t* perm, T* op, int64_t* arg, int fresh, const HubStage& st,
            hipStream_t stream) {
  if (!st.hubs) return PYG_HIP_OK;
  if (st.hw.counters) {
    const int64_t grid = std::min<int64_t>(st.max_chunks, (int64_t)device_info().num_cus * 8);
    hipLaunchKernelGGL((segment_csr_hub_chunk_kernel<T, OP, V, PERM>), dim3((unsigned)grid), dim3(256), 0, stream, sp, indptr, perm,
                       op, arg, fresh, st.sc, st.hw);
    PYG_HIP_CHECK(hipGetLastError());
    if (st.sc.E > st.hw.CH) {   // (else every hub is one chunk long)
      const int64_t cgrid = std::min<int64_t>((st.max_hubs * (st.sc.K / V) + 255) / 256, (int64_t)device_info().num_cus * 4);
      hipLaunchKernelGGL((segment_csr_hub_combine_kernel<T, OP, V>), dim3((unsigned)cgrid), dim3(256), 0, stream, indptr, op, arg,
                         fresh, st.sc, st.hw);
    }
  } else {
    const int64_t batches = (st.sc.leading * st.sc.rows + 255) / 256;
    const int64_t grid = std::min<int64_t>(batches, (int64_t)device_info().num_cus * 8);
    hipLaunchKernelGGL((segment_csr_long_kernel<T, OP, V, PERM>), dim3((unsigned)grid), dim3(256), 0, stream, sp, indptr, perm, op,
                       arg, fresh, st.sc);
  }
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

template <typename T, int OP, int V, bool PERM = false>
int launch_segment(const void* src, const int64_t* indptr, const int64_t* perm, void* out, int64_t* arg, int fresh,
                   const CsrShape& s, hipStream_t stream) {
  const int64_t items = s.leading * s.rows * (s.K / V);
  const int L = pick_lanes(items, s.leading * s.E, s.leading * s.rows, s.K * (int64_t)sizeof(T));
  const T* sp = static_cast<const T*>(src);
  T* op = static_cast<T*>(out);
  HubStage st;
  if (int rc_ = hub_begin<T, OP>(s, kHubCut * (int64_t)L, st, stream)) return rc_;
#define PYG_CSR_LAUNCH(LL)                                                                                   \
  hipLaunchKernelGGL((segment_csr_kernel<T, OP, V, LL, PERM>), dim3((unsigned)((items * LL + 255) / 256)), dim3(256), \
                     0, stream, sp, indptr, perm, op, arg, fresh, st.sc, st.hw)
  if (L == 64) PYG_CSR_LAUNCH(64);
  else if (L == 8) PYG_CSR_LAUNCH(8);
  else PYG_CSR_LAUNCH(1);
#undef PYG_CSR_LAUNCH
  PYG_HIP_CHECK(hipGetLastError());
  return hub_end<T, OP, V, PERM>(sp, indptr, perm, op, arg, fresh, st, stream);
}

template <typename T, int OP>
int launch_stream(const void* src, const int64_t* indptr, void* out, int64_t* arg, int fresh, const CsrShape& s,
                  hipStream_t stream) {
  const int rpb = 256 / (int)s.K;
  const int64_t blocks = s.leading * ((s.rows + rpb - 1) / rpb);
  HubStage st;
  if (int rc_ = hub_begin<T, OP>(s, kHubCutStream, st, stream)) return rc_;
  hipLaunchKernelGGL((segment_csr_stream_kernel<T, OP>), dim3((unsigned)blocks), dim3(256), 0, stream,
                     static_cast<const T*>(src), indptr, static_cast<T*>(out), arg, fresh, st.sc, rpb, st.hw);
  PYG_HIP_CHECK(hipGetLastError());
  return hub_end<T, OP, 1, false>(static_cast<const T*>(src), indptr, nullptr, static_cast<T*>(out), arg, fresh, st, stream);
}

// rows of less than 64 bytes that are not very long on average take the LDS-streamed kernel (a thread walks its row out of LDS:
// from ~64 positions per row on, lanes over the positions are faster)
#ifndef PYG_CSR_STREAM_MAX_AVG
#define PYG_CSR_STREAM_MAX_AVG 64
#endif
constexpr int64_t kStreamMaxAvg = PYG_CSR_STREAM_MAX_AVG;
#ifndef PYG_CSR_STREAM_MAX_ROW_BYTES
#define PYG_CSR_STREAM_MAX_ROW_BYTES 64
#endif
constexpr int64_t kStreamMaxRowBytes = PYG_CSR_STREAM_MAX_ROW_BYTES;
// (rows of fewer than 12 positions on average: one thread per (row, element) on global memory is 1.3 - 2 x faster than the
// workgroup's trip through LDS -- fp32 K = 1, 2 per row: 0.088 -> 0.049 ms; K = 5, 8 per row: 0.194 -> 0.104)
#ifndef PYG_CSR_STREAM_MIN_AVG
#define PYG_CSR_STREAM_MIN_AVG 12
#endif
constexpr int64_t kStreamMinAvg = PYG_CSR_STREAM_MIN_AVG;
// (`slices16`: the rows are whole 16-byte slices the row kernel can load as such.  Then the row kernel always wins -- fp32
// K = 8, 2 positions per row, 16 M positions: 0.71 ms streamed, 0.15 direct; 48 per row: 0.22 streamed, 0.14 with one lane per
// item, 0.095 with eight; `tools/narrow_row_kernels.py`)
template <typename T>
bool use_stream(const CsrShape& s, bool slices16 = false) {
  if (s.K < 1 || s.K > 16 || s.K * (int64_t)sizeof(T) >= kStreamMaxRowBytes) return false;
  const int64_t units = s.leading * s.rows;
  if (units <= 0) return false;
  const int64_t avg = (s.leading * s.E) / units;
  if (slices16 && (s.K * (int64_t)sizeof(T) >= 32 || avg < 32 || kNarrowVecLanesAvg < 64)) return false;
  if (avg < kStreamMinAvg) return false;
  return avg < kStreamMaxAvg && s.leading * ((s.rows + 255) / 256) < (1ll << 31);
}

template <typename T>
int run_segment(int op, const void* src, const int64_t* indptr, void* out, int64_t* arg, int fresh, const CsrShape& s,
                hipStream_t stream) {
  constexpr int VMAX = 16 / (int)sizeof(T);
  const bool vec = VMAX > 1 && s.K % VMAX == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0);
  if (use_stream<T>(s, vec)) {
    switch (op) {
      case CSR_SUM: return launch_stream<T, CSR_SUM>(src, indptr, out, arg, fresh, s, stream);
      case CSR_MEAN:
        if constexpr (std::is_integral<T>::value) return fail(PYG_HIP_ERR_INVALID, "segment_mean_csr: floating dtypes only");
        else return launch_stream<T, CSR_MEAN>(src, indptr, out, arg, fresh, s, stream);
      case CSR_MIN: return launch_stream<T, CSR_MIN>(src, indptr, out, arg, fresh, s, stream);
      case CSR_MAX: return launch_stream<T, CSR_MAX>(src, indptr, out, arg, fresh, s, stream);
      default: return fail(PYG_HIP_ERR_INVALID, "segment_csr: unknown reduction %d", op);
    }
  }
#define PYG_CSR_OP(OPC)                                                                              \
  return vec ? launch_segment<T, OPC, VMAX>(src, indptr, nullptr, out, arg, fresh, s, stream)        \
             : launch_segment<T, OPC, 1>(src, indptr, nullptr, out, arg, fresh, s, stream)
  switch (op) {
    case CSR_SUM: PYG_CSR_OP(CSR_SUM);
    case CSR_MEAN:
      if constexpr (std::is_integral<T>::value) return fail(PYG_HIP_ERR_INVALID, "segment_mean_csr: floating dtypes only");
      else PYG_CSR_OP(CSR_MEAN);
    case CSR_MIN: PYG_CSR_OP(CSR_MIN);
    case CSR_MAX: PYG_CSR_OP(CSR_MAX);
    default: return fail(PYG_HIP_ERR_INVALID, "segment_csr: unknown reduction %d", op);
  }
#undef PYG_CSR_OP
}

template <typename T, int V>
int launch_gather(const void* src, const int64_t* indptr, void* out, const CsrShape& s, hipStream_t stream) {
  const int64_t items = s.leading * s.rows * (s.K / V);
  int L = pick_lanes(items, s.leading * s.E, s.leading * s.rows, s.K * (int64_t)sizeof(T));
  // narrow rows: a thread writing its slice of position after position leaves 16 bytes per row and store instruction; with
  // 8 lanes per item a store covers 8 consecutive positions.  Rows of whole 16-byte slices from 8 positions per row on (fp32
  // K = 4, 16 per row, 16 M positions: 0.213 -> 0.092 ms; K = 12, 48 per row: 0.46 -> 0.22), element-wise rows from 32 (K = 1, 48
  // per row: 0.062 -> 0.039; at 16 per row K = 5 loses: 0.154 -> 0.212): `tools/lease/ab_gather_lanes.sh`
  if (L == 1 && s.K * (int64_t)sizeof(T) < 64 && s.leading * s.rows > 0 &&
      (s.leading * s.E) / (s.leading * s.rows) >= (V > 1 ? 8 : 32))
    L = 8;
  CsrShape sc = s;
  sc.hub_ws = nullptr, sc.hub_ws_bytes = 0;
  const int64_t cut = kHubCut * (int64_t)L;
  const bool hubs = s.E > cut && s.leading * s.rows > 1;
  if (hubs) sc.long_cut = cut;
  HubWs hw;
  int64_t max_chunks = 0;
  if (hubs && hub_plan(s.hub_ws, s.hub_ws_bytes, s.leading * s.E, s.K, 0, false, &hw, &max_chunks))
    PYG_HIP_CHECK(hipMemsetAsync(hw.counters, 0, 16, stream));
#define PYG_CSR_LAUNCH(LL)                                                                                  \
  hipLaunchKernelGGL((gather_csr_kernel<T, V, LL>), dim3((unsigned)((items * LL + 255) / 256)), dim3(256), 0, \
                     stream, static_cast<const T*>(src), indptr, static_cast<T*>(out), sc, hw)
  if (L == 64) PYG_CSR_LAUNCH(64);
  else if (L == 8) PYG_CSR_LAUNCH(8);
  else PYG_CSR_LAUNCH(1);
#undef PYG_CSR_LAUNCH
  PYG_HIP_CHECK(hipGetLastError());
  if (hubs && hw.counters) {
    const int64_t grid = std::min<int64_t>(max_chunks, (int64_t)device_info().num_cus * 8);
    hipLaunchKernelGGL((gather_csr_hub_chunk_kernel<T, V>), dim3((unsigned)grid), dim3(256), 0, stream, static_cast<const T*>(src),
                       indptr, static_cast<T*>(out), sc, hw);
    PYG_HIP_CHECK(hipGetLastError());
  } else if (hubs) {
    const int64_t batches = (s.leading * s.rows + 255) / 256;
    const int64_t grid = std::min<int64_t>(batches, (int64_t)device_info().num_cus * 8);
    hipLaunchKernelGGL((gather_csr_long_kernel<T, V>), dim3((unsigned)grid), dim3(256), 0, stream, static_cast<const T*>(src),
                       indptr, static_cast<T*>(out), sc);
    PYG_HIP_CHECK(hipGetLastError());
  }
  return PYG_HIP_OK;
}

template <typename T>
int run_gather_csr(const void* src, const int64_t* indptr, void* out, const CsrShape& s, hipStream_t stream) {
  constexpr int VMAX = 16 / (int)sizeof(T);
  const bool vec = VMAX > 1 && s.K % VMAX == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0);
  return vec ? launch_gather<T, VMAX>(src, indptr, out, s, stream) : launch_gather<T, 1>(src, indptr, out, s, stream);
}

template <typename T, bool BACKWARD>
int launch_softmax_long(const void* x, const void* dy, const int64_t* ptr, void* y, int64_t outer, int64_t D, int64_t inner,
                        int64_t groups, int64_t cut, hipStream_t stream) {
  const int64_t grid = std::min<int64_t>((groups + 255) / 256, (int64_t)device_info().num_cus * 8);
  hipLaunchKernelGGL((softmax_csr_long_kernel<T, BACKWARD>), dim3((unsigned)grid), dim3(256), 0, stream, static_cast<const T*>(x),
                     static_cast<const T*>(dy), ptr, static_cast<T*>(y), outer, D, inner, groups, cut);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

template <typename T, bool BACKWARD>
int run_softmax(const void* x, const void* dy, const int64_t* ptr, void* y, int64_t outer, int64_t D, int64_t inner,
                int64_t groups, hipStream_t stream) {
  {
    const CsrShape s{outer, groups, D, inner, 0};
    // the LDS-streamed kernel: inner sizes below 32 bytes in groups of 33 ... 63 positions on average (shorter groups stay in
    // the registers of the one-thread-per-head kernel: inner = 1, 16 per group 0.161 -> 0.114 ms).  Shorter groups and
    // wider heads are faster with one thread per head reading global memory (fp32 inner = 8, 2 per group, 8 M positions: 1.32 ->
    // 0.55 ms forward, 1.52 -> 0.39 backward; inner = 1, 2 per group: 0.37 -> 0.14; but inner = 1, 16 per group: 0.16 streamed,
    // 0.35 direct; `tools/narrow_softmax_kernels.py`)
#ifndef PYG_SOFTMAX_STREAM_MIN_AVG
#define PYG_SOFTMAX_STREAM_MIN_AVG 33
#endif
    // (backward keeps two values per position: 16 in registers, so the streamed kernel takes over earlier)
    constexpr int64_t kMinAvg = BACKWARD ? 12 : PYG_SOFTMAX_STREAM_MIN_AVG;
    if (use_stream<T>(s) && inner * (int64_t)sizeof(T) < 32 && D * outer >= kMinAvg * groups * outer) {
      const int rpb = 256 / (int)inner;
      const int64_t blocks = outer * ((groups + rpb - 1) / rpb);
      const int lds = (int)(sizeof(T) * softmax_values<T, BACKWARD>() * (BACKWARD ? 2 : 1));
      if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&softmax_csr_stream_kernel<T, BACKWARD>), lds)) return rc_;
      CsrShape sc = s;
      const bool hubs = D > kHubCutStream && groups > 1;
      if (hubs) sc.long_cut = kHubCutStream;
      hipLaunchKernelGGL((softmax_csr_stream_kernel<T, BACKWARD>), dim3((unsigned)blocks), dim3(256), lds, stream,
                         static_cast<const T*>(x), static_cast<const T*>(dy), ptr, static_cast<T*>(y), sc, rpb);
      PYG_HIP_CHECK(hipGetLastError());
      return hubs ? launch_softmax_long<T, BACKWARD>(x, dy, ptr, y, outer, D, inner, groups, kHubCutStream, stream) : PYG_HIP_OK;
    }
  }
  const int64_t items = groups * outer * inner;
  const int L = pick_lanes(items, D * outer * inner, items, inner * (int64_t)sizeof(T));
  const int64_t cut = kHubCut * (int64_t)L;
  const bool hubs = D > cut && groups > 1;
  const int64_t long_cut = hubs ? cut : INT64_MAX;
#define PYG_SM_LAUNCH(LL)                                                                                         \
  hipLaunchKernelGGL((softmax_csr_kernel<T, LL, BACKWARD>), dim3((unsigned)((items * LL + 255) / 256)), dim3(256), \
                     0, stream, static_cast<const T*>(x), static_cast<const T*>(dy), ptr, static_cast<T*>(y), outer, \
                     D, inner, groups, long_cut)
  if (L == 64) PYG_SM_LAUNCH(64);
  else if (L == 8) PYG_SM_LAUNCH(8);
  else PYG_SM_LAUNCH(1);
#undef PYG_SM_LAUNCH
  PYG_HIP_CHECK(hipGetLastError());
  return hubs ? launch_softmax_long<T, BACKWARD>(x, dy, ptr, y, outer, D, inner, groups, cut, stream) : PYG_HIP_OK;
}

template <typename T>
int run_minmax_perm(int is_min, const void* src, const int64_t* indptr, const int64_t* perm, void* out, int64_t* arg,
                    int fresh, const CsrShape& s, hipStream_t stream) {
  constexpr int VMAX = 16 / (int)sizeof(T);
  const bool vec = VMAX > 1 && s.K % VMAX == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0);
#define PYG_MM(OPC, VV)                                                                                  \
  (perm ? launch_segment<T, OPC, VV, true>(src, indptr, perm, out, arg, fresh, s, stream)                \
        : launch_segment<T, OPC, VV, false>(src, indptr, nullptr, out, arg, fresh, s, stream))
  if (is_min) return vec ? PYG_MM(CSR_MIN, VMAX) : PYG_MM(CSR_MIN, 1);
  return vec ? PYG_MM(CSR_MAX, VMAX) : PYG_MM(CSR_MAX, 1);
#undef PYG_MM
}

}  // namespace

template <typename T>
int run_sum_perm(const void* src, const int64_t* indptr, const int64_t* perm, void* out, int fresh, const CsrShape& s,
                 hipStream_t stream) {
  constexpr int VMAX = 16 / (int)sizeof(T);
  const bool vec = VMAX > 1 && s.K % VMAX == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0);
  return vec ? launch_segment<T, CSR_SUM, VMAX, true>(src, indptr, perm, out, nullptr, fresh, s, stream)
             : launch_segment<T, CSR_SUM, 1, true>(src, indptr, perm, out, nullptr, fresh, s, stream);
}

int segment_csr_sum(int dtype, const void* src, const int64_t* indptr, int64_t indptr_stride, const int64_t* perm, void* out,
                    int64_t leading, int64_t rows, int64_t E, int64_t K, int fresh, hipStream_t stream, void* hub_ws,
                    size_t hub_ws_bytes) {
  if (leading * rows * K == 0) return PYG_HIP_OK;
  CsrShape s{leading, rows, E, K, indptr_stride};
  s.hub_ws = hub_ws, s.hub_ws_bytes = hub_ws_bytes;
  if (perm) PYG_DISPATCH_ALL(dtype, (run_sum_perm<scalar_t>(src, indptr, perm, out, fresh, s, stream)));
  PYG_DISPATCH_ALL(dtype, (run_segment<scalar_t>(CSR_SUM, src, indptr, out, nullptr, fresh, s, stream)));
}

int segment_csr_minmax(int is_min, int dtype, const void* src, const int64_t* indptr, int64_t indptr_stride,
                       const int64_t* perm, void* out, int64_t* arg, int fresh, int64_t leading, int64_t rows,
                       int64_t E, int64_t K, hipStream_t stream, void* hub_ws, size_t hub_ws_bytes) {
  if (leading * rows * K == 0) return PYG_HIP_OK;
  CsrShape s{leading, rows, E, K, indptr_stride};
  s.hub_ws = hub_ws, s.hub_ws_bytes = hub_ws_bytes;
  if (!perm) {  // plain rows: small K takes the LDS-streamed kernel
    PYG_DISPATCH_ALL(dtype, (run_segment<scalar_t>(is_min ? CSR_MIN : CSR_MAX, src, indptr, out, arg, fresh, s, stream)));
  }
  PYG_DISPATCH_ALL(dtype, (run_minmax_perm<scalar_t>(is_min, src, indptr, perm, out, arg, fresh, s, stream)));
}

}  // namespace pyg_hip

using namespace pyg_hip;

extern "C" {

size_t pyg_hip_csr_hub_workspace_size(int op, int dtype, int64_t leading, int64_t E, int64_t K) {
  if (leading < 0 || E < 0 || K <= 0 || op < 0 || op > 4) return 0;
  size_t acc = 0;
  switch (dtype) {
    case PYG_F16: case PYG_BF16: case PYG_F32: case PYG_I32: acc = 4; break;
    case PYG_F64: case PYG_I64: acc = 8; break;
    case PYG_I16: acc = 2; break;
    default: acc = 1; break;
  }
  const bool gather = op == 4;
  return hub_plan(nullptr, 0, leading * E, K, gather ? 0 : acc, op == CSR_MIN || op == CSR_MAX, nullptr, nullptr) +
         (leading * E > kHubCut ? 256 : 0);   // + alignment slack
}

int pyg_hip_segment_csr_ws(int op, int dtype, const void* src, const int64_t* indptr, int64_t indptr_slice_stride,
                           void* out, int64_t* arg_out, int fresh, int64_t leading, int64_t rows, int64_t E, int64_t K,
                           void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(leading >= 0 && rows >= 0 && E >= 0 && K >= 0, "segment_csr: negative size");
  if (leading * rows * K == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(indptr && out && (src || E == 0), "segment_csr: NULL tensor");
  PYG_HIP_REQUIRE((op != CSR_MIN && op != CSR_MAX) || arg_out, "segment_csr: min/max need arg_out");
  PYG_HIP_REQUIRE(indptr_slice_stride == 0 || indptr_slice_stride >= rows + 1, "segment_csr: bad indptr stride");
  CsrShape s{leading, rows, E, K, indptr_slice_stride};
  s.hub_ws = workspace, s.hub_ws_bytes = workspace ? workspace_bytes : 0;
  PYG_DISPATCH_ALL(dtype, (run_segment<scalar_t>(op, src, indptr, out, arg_out, fresh, s, stream)));
}

int pyg_hip_segment_csr(int op, int dtype, const void* src, const int64_t* indptr, int64_t indptr_slice_stride,
                        void* out, int64_t* arg_out, int fresh, int64_t leading, int64_t rows, int64_t E, int64_t K,
                        void* stream_) {
  return pyg_hip_segment_csr_ws(op, dtype, src, indptr, indptr_slice_stride, out, arg_out, fresh, leading, rows, E, K, nullptr, 0,
                                stream_);
}

int pyg_hip_gather_csr_ws(int dtype, const void* src, const int64_t* indptr, int64_t indptr_slice_stride, void* out,
                          int64_t leading, int64_t rows, int64_t E, int64_t K, void* workspace, size_t workspace_bytes,
                          void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(leading >= 0 && rows >= 0 && E >= 0 && K >= 0, "gather_csr: negative size");
  if (leading * E * K == 0 || rows == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(src && indptr && out, "gather_csr: NULL tensor");
  CsrShape s{leading, rows, E, K, indptr_slice_stride};
  s.hub_ws = workspace, s.hub_ws_bytes = workspace ? workspace_bytes : 0;
  PYG_DISPATCH_ALL(dtype, (run_gather_csr<scalar_t>(src, indptr, out, s, stream)));
}

int pyg_hip_gather_csr(int dtype, const void* src, const int64_t* indptr, int64_t indptr_slice_stride, void* out,
                       int64_t leading, int64_t rows, int64_t E, int64_t K, void* stream_) {
  return pyg_hip_gather_csr_ws(dtype, src, indptr, indptr_slice_stride, out, leading, rows, E, K, nullptr, 0, stream_);
}

int pyg_hip_softmax_csr(int dtype, const void* src, const int64_t* ptr, void* out, int64_t outer, int64_t D,
                        int64_t inner, int64_t groups, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(outer >= 0 && D >= 0 && inner >= 0 && groups >= 0, "softmax_csr: negative size");
  if (outer * D * inner == 0 || groups == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(src && ptr && out, "softmax_csr: NULL tensor");
  if (dtype == PYG_F32) return run_softmax<float, false>(src, nullptr, ptr, out, outer, D, inner, groups, stream);
  if (dtype == PYG_F64) return run_softmax<double, false>(src, nullptr, ptr, out, outer, D, inner, groups, stream);
  return fail(PYG_HIP_ERR_INVALID, "softmax_csr: float32 / float64 only (dtype %d)", dtype);
}

int pyg_hip_softmax_csr_backward(int dtype, const void* out, const void* out_grad, const int64_t* ptr, void* in_grad,
                                 int64_t outer, int64_t D, int64_t inner, int64_t groups, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(outer >= 0 && D >= 0 && inner >= 0 && groups >= 0, "softmax_csr_backward: negative size");
  if (outer * D * inner == 0 || groups == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(out && out_grad && ptr && in_grad, "softmax_csr_backward: NULL tensor");
  if (dtype == PYG_F32) return run_softmax<float, true>(out, out_grad, ptr, in_grad, outer, D, inner, groups, stream);
  if (dtype == PYG_F64) return run_softmax<double, true>(out, out_grad, ptr, in_grad, outer, D, inner, groups, stream);
  return fail(PYG_HIP_ERR_INVALID, "softmax_csr_backward: float32 / float64 only (dtype %d)", dtype);
}

}  // extern "C"
