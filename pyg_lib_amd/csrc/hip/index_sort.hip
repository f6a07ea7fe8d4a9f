// index_sort for gfx950 (MI355X): stable LSD radix sort of an integer key vector, returning the
// sorted keys and the int64 permutation.
//
// Replaces pyg_lib/csrc/ops/cpu/index_sort_kernel.cpp:14-59 + ops/cpu/radix_sort.h:58-198 (OpenMP,
// FBGEMM-derived).  The reference has no device kernel: its Python wrapper sends non-CPU tensors to
// torch.sort (pyg_lib/ops/__init__.py:319-320).
//
// Same algorithm family (8-bit digits, least significant first, number of passes from the largest
// key, radix_sort.h:170-176), re-shaped for the chip: every workgroup owns one contiguous slice of
// the input for the whole pass, so there is one 256-bin histogram per workgroup (not per tile) and
// the digit-major scan over them is tiny.  Inside a tile the 64-lane wavefronts rank their keys
// with 8 ballots per digit (a wave64 "match-any"), LDS holds the per-wave digit counters, and
// the running per-digit output cursors of the workgroup live in LDS across tiles.  Stability follows
// from processing slices, tiles, rounds, waves and lanes in input order, which makes the result
// bit-identical to torch.sort(stable=True) -- for negative keys too (sign bit flipped), which the
// reference's radix path mis-sorts.
// HBM-bound: per pass one key read for the histogram, one key+index read and one key+index write.
#include "common.h"
#include "scan.h"

#include <algorithm>

namespace pyg_hip {
namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kItems = 8;                       // keys per thread per tile
constexpr int kTile = kThreads * kItems;        // 2048 keys per tile

template <typename K>
struct KeyTraits;
#define PYG_KEY(K, U, FLIP)                                             \
  template <>                                                           \
  struct KeyTraits<K> {                                                 \
    using bits_t = U;                                                   \
    __host__ __device__ static U encode(K k) { return (U)((U)k ^ (U)(FLIP)); } \
  };
PYG_KEY(uint8_t, uint8_t, 0)
PYG_KEY(int8_t, uint8_t, 0x80u)
PYG_KEY(int16_t, uint16_t, 0x8000u)
PYG_KEY(int32_t, uint32_t, 0x80000000u)
PYG_KEY(int64_t, uint64_t, 0x8000000000000000ull)
#undef PYG_KEY

template <typename K>
__device__ __forceinline__ unsigned digit_of(K key, int shift) {
  return (unsigned)((KeyTraits<K>::encode(key) >> shift) & 0xff);
}

// ---- min / max of the keys (only when the caller gave no `max`) --------------------------------------
template <typename K>
__global__ __launch_bounds__(kThreads) void minmax_kernel(const K* __restrict__ keys, int64_t n,
                                                          int64_t* __restrict__ out /* [min, max] */) {
  __shared__ int64_t smin[kThreads];
  __shared__ int64_t smax[kThreads];
  int64_t lo = INT64_MAX, hi = INT64_MIN;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = (int64_t)keys[i];
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
  smin[threadIdx.x] = lo;
  smax[threadIdx.x] = hi;
  __syncthreads();
  for (int d = kThreads / 2; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) {
      smin[threadIdx.x] = min(smin[threadIdx.x], smin[threadIdx.x + d]);
      smax[threadIdx.x] = max(smax[threadIdx.x], smax[threadIdx.x + d]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicMin(reinterpret_cast<long long*>(out), (long long)smin[0]);
    atomicMax(reinterpret_cast<long long*>(out + 1), (long long)smax[0]);
  }
}

// ---- pass kernels ---------------------------------------------------------------------------------------
// hist[d * G + g]: number of keys with digit d in the slice of workgroup g
template <typename K>
__global__ __launch_bounds__(kThreads) void hist_kernel(const K* __restrict__ keys, int64_t n, int64_t slice,
                                                        int shift, int64_t* __restrict__ hist) {
  __shared__ unsigned int bins[256];
  bins[threadIdx.x] = 0;
  __syncthreads();
  const int64_t beg = blockIdx.x * slice;
  const int64_t end = min(beg + slice, n);
  for (int64_t i = beg + threadIdx.x; i < end; i += kThreads) atomicAdd(&bins[digit_of(keys[i], shift)], 1u);
  __syncthreads();
  hist[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = bins[threadIdx.x];
}

struct HistLoad {
  const int64_t* h;
  __device__ int64_t operator()(int64_t i) const { return h[i]; }
};
struct HistStore {
  int64_t* o;
  __device__ void operator()(int64_t i, const int64_t& prefix, const int64_t&) const { o[i] = prefix; }
};

// Stable scatter of one pass.  FIRST: the index payload is the identity (arange), not read.
template <typename K, bool FIRST>
__global__ __launch_bounds__(kThreads) void scatter_kernel(const K* __restrict__ keys_in,
                                                           const int64_t* __restrict__ idx_in, K* __restrict__ keys_out,
                                                           int64_t* __restrict__ idx_out, int64_t n, int64_t slice,
                                                           int shift, const int64_t* __restrict__ offsets) {
  __shared__ int64_t cursor[256];                // next output position per digit (this workgroup)
  __shared__ unsigned int wave_cnt[kWaves][256]; // keys per (wave, digit) in the current round
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  cursor[tid] = offsets[(int64_t)tid * gridDim.x + blockIdx.x];
  const int64_t beg = blockIdx.x * slice;
  const int64_t end = min(beg + slice, n);
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));

  for (int64_t tile = beg; tile < end; tile += kTile) {
    // rounds keep input order: element = tile + r * 256 + wave * 64 + lane
#pragma unroll 1
    for (int r = 0; r < kItems; ++r) {
      const int64_t i = tile + (int64_t)r * kThreads + tid;
      const bool valid = i < end;
      for (int w = 0; w < kWaves; ++w) wave_cnt[w][tid] = 0;
      __syncthreads();
      K key = 0;
      unsigned d = 0;
      if (valid) {
        key = keys_in[i];
        d = digit_of(key, shift);
      }
      // wave64 match-any on the 8-bit digit: 8 ballots
      unsigned long long peers = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const unsigned long long m = __ballot((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? m : ~m;
      }
      const unsigned rank_in_wave = (unsigned)__popcll(peers & lt_mask);
      if (valid && rank_in_wave == 0) wave_cnt[wave][d] = (unsigned)__popcll(peers);
      __syncthreads();
      if (valid) {
        unsigned before = 0;
        for (int w = 0; w < wave; ++w) before += wave_cnt[w][d];
        const int64_t pos = cursor[d] + before + rank_in_wave;
        keys_out[pos] = key;
        idx_out[pos] = FIRST ? i : idx_in[i];
      }
      __syncthreads();
      {
        unsigned tot = 0;
        for (int w = 0; w < kWaves; ++w) tot += wave_cnt[w][tid];
        cursor[tid] += tot;
      }
      __syncthreads();
    }
  }
}

template <typename K>
__global__ void copy_identity_kernel(const K* __restrict__ in, K* __restrict__ out, int64_t* __restrict__ idx,
                                     int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) {
    out[i] = in[i];
    idx[i] = i;
  }
}

struct Plan {
  int64_t groups;  // workgroups per pass
  int64_t slice;   // keys per workgroup (multiple of kTile)
};

Plan make_plan(int64_t n) {
  const int64_t tiles = (n + kTile - 1) / kTile;
  int64_t groups = std::min<int64_t>(tiles, (int64_t)device_info().num_cus * 8);
  if (groups < 1) groups = 1;
  const int64_t tiles_per = (tiles + groups - 1) / groups;
  Plan p;
  p.slice = tiles_per * kTile;
  p.groups = (n + p.slice - 1) / p.slice;
  if (p.groups < 1) p.groups = 1;
  return p;
}

size_t ws_bytes(int64_t n, size_t key_size) {
  const Plan p = make_plan(n);
  size_t b = 0;
  b += align_up((size_t)n * key_size, 256);                       // key ping-pong buffer
  b += align_up((size_t)n * sizeof(int64_t), 256);                // index ping-pong buffer
  b += 2 * align_up((size_t)p.groups * 256 * sizeof(int64_t), 256);  // histograms, offsets
  b += align_up(((size_t)(p.groups * 256 + kScanTile - 1) / kScanTile + 2) * sizeof(int64_t), 256);
  b += 256;                                                       // min/max
  return b;
}

template <typename K>
int run_sort(const void* keys_, int64_t n, int64_t max_value, int has_max, void* keys_out_, int64_t* idx_out,
             void* ws, size_t ws_size, hipStream_t stream) {
  const K* keys = static_cast<const K*>(keys_);
  K* keys_out = static_cast<K*>(keys_out_);
  if (n == 0) return PYG_HIP_OK;
  if (ws_size < ws_bytes(n, sizeof(K)) || !ws)
    return fail(PYG_HIP_ERR_WORKSPACE, "index_sort: workspace of %zu bytes needed, got %zu", ws_bytes(n, sizeof(K)),
                ws_size);
  const Plan p = make_plan(n);
  char* w = static_cast<char*>(ws);
  K* kbuf = reinterpret_cast<K*>(w);
  w += align_up((size_t)n * sizeof(K), 256);
  int64_t* ibuf = reinterpret_cast<int64_t*>(w);
  w += align_up((size_t)n * sizeof(int64_t), 256);
  int64_t* hist = reinterpret_cast<int64_t*>(w);
  w += align_up((size_t)p.groups * 256 * sizeof(int64_t), 256);
  int64_t* offs = reinterpret_cast<int64_t*>(w);
  w += align_up((size_t)p.groups * 256 * sizeof(int64_t), 256);
  int64_t* scan_tmp = reinterpret_cast<int64_t*>(w);
  w += align_up(((size_t)(p.groups * 256 + kScanTile - 1) / kScanTile + 2) * sizeof(int64_t), 256);
  int64_t* mm = reinterpret_cast<int64_t*>(w);

  // number of 8-bit passes (radix_sort.h:170-176: from the largest key)
  int passes;
  constexpr int full = (int)sizeof(K);
  if (has_max) {
    // keys are promised to lie in [0, max_value]
    uint64_t m = max_value < 0 ? 0 : (uint64_t)max_value;
    passes = 0;
    while (m) {
      ++passes;
      m >>= 8;
    }
    passes = std::max(1, std::min(passes, full));
  } else {
    static const int64_t init[2] = {INT64_MAX, INT64_MIN};
    PYG_HIP_CHECK(hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL((minmax_kernel<K>), dim3((unsigned)std::min<int64_t>((n + kThreads - 1) / kThreads, 2048)),
                       dim3(kThreads), 0, stream, keys, n, mm);
    PYG_HIP_CHECK(hipGetLastError());
    int64_t host_mm[2];
    PYG_HIP_CHECK(hipMemcpyAsync(host_mm, mm, sizeof(host_mm), hipMemcpyDeviceToHost, stream));
    PYG_HIP_CHECK(hipStreamSynchronize(stream));  // the reference syncs here too: input.max().item()
    if (host_mm[0] < 0) {
      passes = full;  // negative keys: all digits matter (sign bit flipped)
    } else {
      uint64_t m = (uint64_t)host_mm[1];
      passes = 0;
      while (m) {
        ++passes;
        m >>= 8;
      }
      passes = std::max(1, std::min(passes, full));
    }
  }
  // With fewer than `full` passes the flipped sign bit is never looked at, which is fine: all keys are
  // non-negative.  Ping-pong so that the last pass lands in the caller's buffers.
  K* kin = nullptr;
  int64_t* iin = nullptr;
  K* kdst[2] = {keys_out, kbuf};
  int64_t* idst[2] = {idx_out, ibuf};
  int cur = (passes & 1) ? 0 : 1;  // destination of pass 0
  for (int ps = 0; ps < passes; ++ps) {
    const int shift = 8 * ps;
    const K* src_k = ps == 0 ? keys : kin;
    hipLaunchKernelGGL((hist_kernel<K>), dim3((unsigned)p.groups), dim3(kThreads), 0, stream, src_k, n, p.slice,
                       shift, hist);
    PYG_HIP_CHECK(hipGetLastError());
    const int64_t ntiles = (p.groups * 256 + kScanTile - 1) / kScanTile;
    int rc = device_scan<int64_t, SumOp>(HistLoad{hist}, HistStore{offs}, p.groups * 256, scan_tmp,
                                         scan_tmp + ntiles, stream);
    if (rc != PYG_HIP_OK) return rc;
    if (ps == 0)
      hipLaunchKernelGGL((scatter_kernel<K, true>), dim3((unsigned)p.groups), dim3(kThreads), 0, stream, src_k,
                         (const int64_t*)nullptr, kdst[cur], idst[cur], n, p.slice, shift, offs);
    else
      hipLaunchKernelGGL((scatter_kernel<K, false>), dim3((unsigned)p.groups), dim3(kThreads), 0, stream, src_k,
                         (const int64_t*)iin, kdst[cur], idst[cur], n, p.slice, shift, offs);
    PYG_HIP_CHECK(hipGetLastError());
    kin = kdst[cur];
    iin = idst[cur];
    cur ^= 1;
  }
  return PYG_HIP_OK;
}

}  // namespace

size_t index_sort_ws_bytes_i64(int64_t n) { return ws_bytes(n < 0 ? 0 : n, sizeof(int64_t)); }

int index_sort_i64(const int64_t* keys, int64_t n, int64_t max_value, int64_t* keys_out, int64_t* idx_out,
                   void* ws, size_t ws_size, hipStream_t stream) {
  return run_sort<int64_t>(keys, n, max_value, 1, keys_out, idx_out, ws, ws_size, stream);
}

}  // namespace pyg_hip

using namespace pyg_hip;

extern "C" {

size_t pyg_hip_index_sort_workspace_size(int dtype, int64_t n) {
  const size_t ks = dtype_size(dtype);
  return ws_bytes(n < 0 ? 0 : n, ks ? ks : 8);
}

int pyg_hip_index_sort(int dtype, const void* keys, int64_t n, int64_t max_value, int has_max, void* keys_out,
                       int64_t* index_out, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(n >= 0, "index_sort: negative size");
  if (n == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(keys && keys_out && index_out, "index_sort: NULL tensor");
  switch (dtype) {
    case PYG_U8: return run_sort<uint8_t>(keys, n, max_value, has_max, keys_out, index_out, workspace, workspace_bytes, stream);
    case PYG_I8: return run_sort<int8_t>(keys, n, max_value, has_max, keys_out, index_out, workspace, workspace_bytes, stream);
    case PYG_I16: return run_sort<int16_t>(keys, n, max_value, has_max, keys_out, index_out, workspace, workspace_bytes, stream);
    case PYG_I32: return run_sort<int32_t>(keys, n, max_value, has_max, keys_out, index_out, workspace, workspace_bytes, stream);
    case PYG_I64: return run_sort<int64_t>(keys, n, max_value, has_max, keys_out, index_out, workspace, workspace_bytes, stream);
    default:
      // index_sort_kernel.cpp:55-56
      return fail(PYG_HIP_ERR_INVALID, "Input should contain integral values.");
  }
}

}  // extern "C"
