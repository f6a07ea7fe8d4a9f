// Order-preserving device-wide exclusive scan with a generic (possibly non-commutative) operator.
// One launch for a single tile, two (tile aggregates; apply with a per-block spine) up to 2048 tiles,
// three (tile aggregates, spine, apply) beyond.  Load / Store
// are functors so that producers and consumers fuse into the scan's passes.
#pragma once

#include "common.h"

namespace pyg_hip {

struct SumOp {
  __host__ __device__ int64_t operator()(int64_t a, int64_t b) const { return a + b; }
  __host__ __device__ static int64_t identity() { return 0; }
};

// ---- device-wide exclusive scan (order preserving, generic operator) -------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 4;
constexpr int kScanTile = kScanThreads * kScanItems;

// Shuffle any trivially copyable T up by `delta` lanes, 4 bytes at a time.
template <typename T>
__device__ __forceinline__ T shfl_up_any(const T& v, int delta) {
  static_assert(sizeof(T) % 4 == 0, "4-byte multiples only");
  T out;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&v);
  uint32_t* dst = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 4); ++i) dst[i] = (uint32_t)__shfl_up((int)src[i], delta);
  return out;
}

// Ordered exclusive scan across the 256 threads of a block: wave64 shuffle scan, then the four wave
// totals through LDS.  `lds` needs 8 entries.  Valid for any associative operator (not necessarily
// commutative): lane order is thread order.
template <typename T, typename Op>
__device__ T block_exclusive(T agg, T* lds, Op op, T* total) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  T incl = agg;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const T up = shfl_up_any(incl, d);
    if (lane >= d) incl = op(up, incl);
  }
  if (lane == 63) lds[wave] = incl;
  __syncthreads();
  T before = Op::identity();
  T all = Op::identity();
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; ++w) {
    const T t = lds[w];
    if (w < wave) before = op(before, t);
    all = op(all, t);
  }
  T excl = shfl_up_any(incl, 1);
  if (lane == 0) excl = Op::identity();
  excl = op(before, excl);
  *total = all;
  __syncthreads();
  return excl;
}

// Phase A: per-tile aggregates.  `cache` (optional, n entries): the loaded values are kept for phase C, whose
// Load is then not evaluated a second time (the sampler's loads are dependent random gathers).
template <typename T, typename Op, typename Load>
__global__ __launch_bounds__(kScanThreads) void scan_reduce_kernel(Load load, int64_t n,
                                                                   T* __restrict__ tile_agg, T* __restrict__ cache) {
  __shared__ T lds[8];
  Op op;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  // loads are unconditional (index clamped) so that the gathers of all items issue back to back
  T v[kScanItems];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) v[k] = load(base + k < n ? base + k : n - 1);
  T agg = Op::identity();
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (base + k < n) {
      agg = op(agg, v[k]);
      if (cache) cache[base + k] = v[k];
    }
  T total;
  (void)block_exclusive<T, Op>(agg, lds, op, &total);
  if (threadIdx.x == 0) tile_agg[blockIdx.x] = total;
}

// Phase B: exclusive scan of the tile aggregates by one block; writes the grand total.
template <typename T, typename Op>
__global__ __launch_bounds__(kScanThreads) void scan_spine_kernel(T* __restrict__ tile_agg,
                                                                  int64_t ntiles,
                                                                  T* __restrict__ total_out) {
  __shared__ T lds[8];
  Op op;
  T carry = Op::identity();
  for (int64_t c0 = 0; c0 < ntiles; c0 += kScanThreads) {
    const int64_t i = c0 + threadIdx.x;
    const T v = i < ntiles ? tile_agg[i] : Op::identity();
    T total;
    const T excl = block_exclusive<T, Op>(v, lds, op, &total);
    if (i < ntiles) tile_agg[i] = op(carry, excl);
    carry = op(carry, total);
  }
  if (threadIdx.x == 0) *total_out = carry;
}

// Phase C: per-element exclusive prefixes -> Store.  `tile_prefix` holds exclusive tile prefixes (after
// the spine kernel), or -- SELF_SPINE -- the raw tile aggregates, which every block reduces for itself
// (tiles [0, blockIdx.x), in order): one dependent launch less for the few hundred tiles of a hop.
template <typename T, typename Op, typename Load, typename Store, bool SELF_SPINE>
__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(Load load, Store store, int64_t n,
                                                                  const T* __restrict__ tile_prefix,
                                                                  T* __restrict__ total_out,
                                                                  const T* __restrict__ cache = nullptr) {
  __shared__ T lds[8];
  Op op;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  T v[kScanItems];
  if (cache) {
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) v[k] = cache[base + k < n ? base + k : n - 1];
  } else {
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) v[k] = load(base + k < n ? base + k : n - 1);
  }
  T agg = Op::identity();
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k >= n) v[k] = Op::identity();
    agg = op(agg, v[k]);
  }
  T total;
  T run = block_exclusive<T, Op>(agg, lds, op, &total);
  if (SELF_SPINE) {
    T before = Op::identity();
    for (int64_t c0 = 0; c0 < (int64_t)blockIdx.x; c0 += kScanThreads) {
      const int64_t i = c0 + threadIdx.x;
      const T t = i < (int64_t)blockIdx.x ? tile_prefix[i] : Op::identity();
      T chunk;
      (void)block_exclusive<T, Op>(t, lds, op, &chunk);
      before = op(before, chunk);
    }
    run = op(before, run);
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1 && total_out) *total_out = op(before, total);
  } else if (tile_prefix) {
    run = op(tile_prefix[blockIdx.x], run);
  } else if (threadIdx.x == 0 && blockIdx.x == 0 && total_out) {
    *total_out = total;  // single tile
  }
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < n) store(base + k, run, v[k]);
    run = op(run, v[k]);
  }
}

// scratch: ntiles * sizeof(T) + sizeof(T) (total).  Returns device pointer to the total.
// `cache`: optional scratch of n values of T (see scan_reduce_kernel); ignored for a single tile.
template <typename T, typename Op, typename Load, typename Store>
int device_scan(Load load, Store store, int64_t n, T* tile_buf, T* total_dev, hipStream_t stream, T* cache = nullptr) {
  if (n <= 0) {
    T id = Op::identity();
    PYG_HIP_CHECK(hipMemcpyAsync(total_dev, &id, sizeof(T), hipMemcpyHostToDevice, stream));
    return PYG_HIP_OK;
  }
  const int64_t ntiles = (n + kScanTile - 1) / kScanTile;
  if (ntiles == 1) {
    hipLaunchKernelGGL((scan_apply_kernel<T, Op, Load, Store, false>), dim3(1), dim3(kScanThreads), 0,
                       stream, load, store, n, (const T*)nullptr, total_dev);
  } else if (ntiles <= 2048) {
    hipLaunchKernelGGL((scan_reduce_kernel<T, Op, Load>), dim3((unsigned)ntiles),
                       dim3(kScanThreads), 0, stream, load, n, tile_buf, cache);
    hipLaunchKernelGGL((scan_apply_kernel<T, Op, Load, Store, true>), dim3((unsigned)ntiles),
                       dim3(kScanThreads), 0, stream, load, store, n, (const T*)tile_buf, total_dev, (const T*)cache);
  } else {
    hipLaunchKernelGGL((scan_reduce_kernel<T, Op, Load>), dim3((unsigned)ntiles),
                       dim3(kScanThreads), 0, stream, load, n, tile_buf, cache);
    hipLaunchKernelGGL((scan_spine_kernel<T, Op>), dim3(1), dim3(kScanThreads), 0, stream, tile_buf,
                       ntiles, total_dev);
    hipLaunchKernelGGL((scan_apply_kernel<T, Op, Load, Store, false>), dim3((unsigned)ntiles),
                       dim3(kScanThreads), 0, stream, load, store, n, (const T*)tile_buf,
                       (T*)nullptr, (const T*)cache);
  }
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}


}  // namespace pyg_hip
