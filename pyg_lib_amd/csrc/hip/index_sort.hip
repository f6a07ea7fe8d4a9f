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
// the digit-major scan over them is tiny.  A tile is 8192 keys: MI355X's 160 KB LDS holds the whole
// tile's (key, index) pairs, so the tile is first sorted by digit INSIDE LDS and then written out --
// a digit's keys leave as one run of ~32 consecutive elements (256-byte bursts) instead of one
// scattered 8-byte write per key.  Ranking: every wave owns 1024 consecutive keys and ranks them in 16
// rounds of 64 with 8 ballots per round (a wave64 "match-any") against its own running digit counters
// in LDS -- no workgroup barrier inside the ranking; three barriers per tile in all.  The running
// per-digit output cursors of the workgroup live in LDS across tiles.  Stability follows from ranking
// in (slice, tile, wave, round, lane) = input order, which makes the result bit-identical to
// torch.sort(stable=True) -- for negative keys too (sign bit flipped), which the reference's radix path
// mis-sorts.
// HBM-bound: per pass one key read for the histogram, one key+index read and one key+index write.
//
// PACKED mode (non-negative keys whose significant bits plus the bits of n - 1 fit 64 -- every index vector PyG
// sorts): key and position travel as ONE 64-bit word (key << index_bits | position) through the passes; the first
// pass packs while it reads the keys, the last pass splits into the two output vectors.  8 instead of 16 bytes per
// element per pass in both directions (12 -> 8 GB for 1e8 int64 keys in 3 passes), 64 KB instead of 128 KB of LDS per
// 8192-key tile (two workgroups per CU).  Same stable order, bit-identical result.
#include "common.h"

#include <stdlib.h>

#include <mutex>
#include "scan.h"

#include <algorithm>

namespace pyg_hip {
namespace {

constexpr int kThreads = 256;                   // histogram / min-max kernels
constexpr int kSThreads = 512;                  // scatter kernel: 8 waves
constexpr int kWaves = kSThreads / 64;
constexpr int kItems = 16;                      // keys per thread per tile
constexpr int kTile = kSThreads * kItems;       // 8192 keys per tile: (key, index) pairs fill 128 KB of LDS
constexpr int kWaveKeys = kTile / kWaves;       // 1024 consecutive keys per wave

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
PYG_KEY(uint64_t, uint64_t, 0)  // packed words
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
// hist[d * G + g]: number of keys with digit d in the slice of workgroup g.  A wave whose keys all share the digit
// (a constant high byte) adds once; otherwise one LDS atomic per key.
// RAW (packed mode, non-negative keys): digits of the value itself, no sign-bit flip.
template <typename K, bool RAW = false>
__global__ __launch_bounds__(kThreads) void hist_kernel(const K* __restrict__ keys, int64_t n, int64_t slice,
                                                        int shift, int64_t* __restrict__ hist) {
  __shared__ unsigned int bins[256];
  bins[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t beg = blockIdx.x * slice;
  const int64_t end = min(beg + slice, n);
  // all lanes on one digit (a constant high byte): one add for the wave; otherwise one LDS atomic per key -- a handful
  // of bank cycles per wave for spread digits, where a 9-ballot match-any cost ~100 cycles per trip
  auto count = [&](K key, bool valid) {
    const unsigned d = !valid ? 0u : RAW ? (unsigned)(((uint64_t)key >> shift) & 0xff) : digit_of(key, shift);
    const unsigned d0 = (unsigned)__builtin_amdgcn_readfirstlane((int)d);
    const unsigned long long same = __ballot(valid && d == d0);
    const unsigned long long live = __ballot(valid);
    if (same == live) {
      if (lane == 0 && live) atomicAdd(&bins[d0], (unsigned)__popcll(live));
    } else if (valid) {
      atomicAdd(&bins[d], 1u);
    }
  };
  constexpr int U = 4;
  bool wide = false;
  if constexpr (sizeof(K) == 8) wide = (reinterpret_cast<uintptr_t>(keys + beg) & 15) == 0;  // a sliced tensor may not be
  if (wide) {
    // 16-byte loads (two keys), four per thread and trip in flight: one 8-byte load per trip left this kernel at
    // 2.2 TB/s.  Slices start on even positions (multiples of the scatter tile).
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const int64_t npair = (end - beg) / 2;
    const u64x2* kp = reinterpret_cast<const u64x2*>(keys + beg);
    for (int64_t j0 = 0; j0 < npair; j0 += (int64_t)U * kThreads) {
      u64x2 kv[U];
      bool valid[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t j = j0 + u * kThreads + threadIdx.x;
        valid[u] = j < npair;
        kv[u] = kp[valid[u] ? j : 0];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        count((K)kv[u][0], valid[u]);
        count((K)kv[u][1], valid[u]);
      }
    }
    if ((end - beg) & 1) count(keys[end - 1], threadIdx.x == 0);
  } else {
    for (int64_t i0 = beg; i0 < end; i0 += (int64_t)U * kThreads) {
      K kv[U];
      bool valid[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + u * kThreads + threadIdx.x;
        valid[u] = i < end;
        kv[u] = keys[valid[u] ? i : beg];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) count(kv[u], valid[u]);
    }
  }
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
template <typename K, bool FIRST, bool ATOMIC_RANK = false>
__global__ __launch_bounds__(kSThreads) void scatter_kernel(const K* __restrict__ keys_in,
                                                            const int64_t* __restrict__ idx_in, K* __restrict__ keys_out,
                                                            int64_t* __restrict__ idx_out, int64_t n, int64_t slice,
                                                            int shift, const int64_t* __restrict__ offsets) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t* sidx = reinterpret_cast<int64_t*>(smem);                      // [kTile] tile sorted by digit: payload
  K* skeys = reinterpret_cast<K*>(smem + sizeof(int64_t) * kTile);        // [kTile]                       keys
  char* rest = smem + (sizeof(int64_t) + sizeof(K)) * kTile;
  rest = reinterpret_cast<char*>(((uintptr_t)rest + 15) & ~(uintptr_t)15);
  int64_t* cursor = reinterpret_cast<int64_t*>(rest);                     // [256] next output position per digit
  unsigned* wcnt = reinterpret_cast<unsigned*>(rest + 256 * sizeof(int64_t));  // [kWaves][256]
  unsigned* tile_pref = wcnt + kWaves * 256;                              // [256] first tile slot of a digit
  unsigned* tile_cnt = tile_pref + 256;                                   // [256]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  if (tid < 256) cursor[tid] = offsets[(int64_t)tid * gridDim.x + blockIdx.x];
  const int64_t beg = blockIdx.x * slice;
  const int64_t end = min(beg + slice, n);
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  unsigned* my_cnt = wcnt + wave * 256;

  for (int64_t tile = beg; tile < end; tile += kTile) {
    const int64_t tile_n = min((int64_t)kTile, end - tile);
    // ---- 1. rank: wave w owns keys [w * 1024, (w + 1) * 1024) of the tile, 16 rounds of 64 -----------
    for (int j = lane; j < 256; j += 64) my_cnt[j] = 0;
    K key[kItems];
    int64_t pay[kItems];
    unsigned short rank[kItems];
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
      const int64_t i = tile + wave * kWaveKeys + r * 64 + lane;
      key[r] = i < end ? keys_in[i] : K(0);
      if (!FIRST) pay[r] = i < end ? idx_in[i] : 0;
    }
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
      const int64_t i = tile + wave * kWaveKeys + r * 64 + lane;
      const bool valid = i < end;
      const unsigned d = valid ? digit_of(key[r], shift) : 0u;
      if (ATOMIC_RANK) {
        // the LDS hands a wave's returning atomics on one address their values in ascending lane order (checked on
        // the device before this variant is used: lds_rank_order_ok): the stable rank without the 9 ballots
        rank[r] = valid ? (unsigned short)atomicAdd(&my_cnt[d], 1u) : (unsigned short)0;
        continue;
      }
      unsigned long long peers = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const unsigned long long m = __ballot((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? m : ~m;
      }
      const unsigned before = my_cnt[d];  // keys of this digit in earlier rounds of this wave
      const unsigned in_round = (unsigned)__popcll(peers & lt_mask);
      rank[r] = (unsigned short)(before + in_round);
      __builtin_amdgcn_wave_barrier();    // every lane has read the counter before the leader bumps it
      if (valid && in_round == 0) my_cnt[d] = before + (unsigned)__popcll(peers);
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- 2. digit totals of the tile, wave offsets, exclusive scan over the 256 digits ---------------
    if (tid < 256) {
      unsigned run = 0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        const unsigned c = wcnt[w * 256 + tid];
        wcnt[w * 256 + tid] = run;
        run += c;
      }
      tile_cnt[tid] = run;
      // scan of 256 totals by the first 4 waves: wave shuffle scan + 4 wave totals
      unsigned incl = run;
#pragma unroll
      for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const unsigned up = (unsigned)__shfl_up((int)incl, dlt);
        if (lane >= dlt) incl += up;
      }
      tile_pref[tid] = incl - run;  // exclusive inside the wave; wave bases added below
      if (lane == 63) sidx[wave] = (int64_t)incl;  // borrow 4 slots of the (not yet written) payload array
    }
    __syncthreads();
    if (tid < 256) {
      unsigned base = 0;
      for (int w = 0; w < wave; ++w) base += (unsigned)sidx[w];
      tile_pref[tid] += base;
    }
    __syncthreads();
    // ---- 3. place every (key, payload) at its digit-sorted slot of the tile ---------------------------
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
      const int64_t i = tile + wave * kWaveKeys + r * 64 + lane;
      if (i < end) {
        const unsigned d = digit_of(key[r], shift);
        const unsigned pos = tile_pref[d] + my_cnt[d] + rank[r];
        skeys[pos] = key[r];
        sidx[pos] = FIRST ? i : pay[r];
      }
    }
    __syncthreads();
    // ---- 4. write out: slot j of digit d goes to cursor[d] + (j - tile_pref[d]); runs are contiguous ---
    for (int j = tid; j < tile_n; j += kSThreads) {
      const K k = skeys[j];
      const unsigned d = digit_of(k, shift);
      const int64_t dest = cursor[d] + (int64_t)(j - tile_pref[d]);
      keys_out[dest] = k;
      idx_out[dest] = sidx[j];
    }
    __syncthreads();
    if (tid < 256) cursor[tid] += tile_cnt[tid];
    // (the next tile's first barrier orders this update before its use in step 4)
  }
}


// Packed mode: stable scatter of one pass over 64-bit words (key << ib | position).  IN_KEYS: the input is the key
// vector (first pass: words are formed on the fly), else words.  OUT_SPLIT: the output is (keys, indices) (last pass),
// else words.  `shift` addresses the digit inside the WORD (ib + 8 * pass).
template <typename K, bool IN_KEYS, bool OUT_SPLIT, bool ATOMIC_RANK = false>
__global__ __launch_bounds__(kSThreads) void scatter_packed_kernel(const K* __restrict__ keys_in, const uint64_t* __restrict__ words_in,
                                                                   uint64_t* __restrict__ words_out, K* __restrict__ keys_out,
                                                                   int64_t* __restrict__ idx_out, int64_t n, int64_t slice, int shift,
                                                                   int ib, const int64_t* __restrict__ offsets) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* swords = reinterpret_cast<uint64_t*>(smem);                   // [kTile] tile sorted by digit
  char* rest = smem + sizeof(uint64_t) * kTile;
  int64_t* cursor = reinterpret_cast<int64_t*>(rest);                     // [256] next output position per digit
  unsigned* wcnt = reinterpret_cast<unsigned*>(rest + 256 * sizeof(int64_t));  // [kWaves][256]
  unsigned* tile_pref = wcnt + kWaves * 256;                              // [256] first tile slot of a digit
  unsigned* tile_cnt = tile_pref + 256;                                   // [256]
  unsigned* wave_tot = tile_cnt + 256;                                    // [4]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  if (tid < 256) cursor[tid] = offsets[(int64_t)tid * gridDim.x + blockIdx.x];
  const int64_t beg = blockIdx.x * slice;
  const int64_t end = min(beg + slice, n);
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const uint64_t imask = ib >= 64 ? ~0ull : ((1ull << ib) - 1);
  unsigned* my_cnt = wcnt + wave * 256;

  for (int64_t tile = beg; tile < end; tile += kTile) {
    const int64_t tile_n = min((int64_t)kTile, end - tile);
    for (int j = lane; j < 256; j += 64) my_cnt[j] = 0;
    uint64_t word[kItems];
    unsigned short rank[kItems];
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
      const int64_t i = tile + wave * kWaveKeys + r * 64 + lane;
      if (IN_KEYS) word[r] = i < end ? (((uint64_t)keys_in[i] << ib) | (uint64_t)i) : 0ull;
      else word[r] = i < end ? words_in[i] : 0ull;
    }
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
      const int64_t i = tile + wave * kWaveKeys + r * 64 + lane;
      const bool valid = i < end;
      const unsigned d = valid ? (unsigned)((word[r] >> shift) & 0xff) : 0u;
      if (ATOMIC_RANK) {
        rank[r] = valid ? (unsigned short)atomicAdd(&my_cnt[d], 1u) : (unsigned short)0;  // see scatter_kernel
        continue;
      }
      unsigned long long peers = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const unsigned long long m = __ballot((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? m : ~m;
      }
      const unsigned before = my_cnt[d];
      const unsigned in_round = (unsigned)__popcll(peers & lt_mask);
      rank[r] = (unsigned short)(before + in_round);
      __builtin_amdgcn_wave_barrier();
      if (valid && in_round == 0) my_cnt[d] = before + (unsigned)__popcll(peers);
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (tid < 256) {
      unsigned run = 0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        const unsigned c = wcnt[w * 256 + tid];
        wcnt[w * 256 + tid] = run;
        run += c;
      }
      tile_cnt[tid] = run;
      unsigned incl = run;
#pragma unroll
      for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const unsigned up = (unsigned)__shfl_up((int)incl, dlt);
        if (lane >= dlt) incl += up;
      }
      tile_pref[tid] = incl - run;
      if (lane == 63) wave_tot[wave] = incl;
    }
    __syncthreads();
    if (tid < 256) {
      unsigned base = 0;
      for (int w = 0; w < wave; ++w) base += wave_tot[w];
      tile_pref[tid] += base;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kItems; ++r) {
      const int64_t i = tile + wave * kWaveKeys + r * 64 + lane;
      if (i < end) {
        const unsigned d = (unsigned)((word[r] >> shift) & 0xff);
        swords[tile_pref[d] + my_cnt[d] + rank[r]] = word[r];
      }
    }
    __syncthreads();
    for (int j = tid; j < tile_n; j += kSThreads) {
      const uint64_t wv = swords[j];
      const unsigned d = (unsigned)((wv >> shift) & 0xff);
      const int64_t dest = cursor[d] + (int64_t)(j - tile_pref[d]);
      if (OUT_SPLIT) {
        keys_out[dest] = (K)(wv >> ib);
        idx_out[dest] = (int64_t)(wv & imask);
      } else {
        words_out[dest] = wv;
      }
    }
    __syncthreads();
    if (tid < 256) cursor[tid] += tile_cnt[tid];
  }
}

constexpr size_t scatter_packed_lds_bytes() {
  return sizeof(uint64_t) * kTile + 256 * sizeof(int64_t) + (kWaves * 256 + 512 + 4) * sizeof(unsigned);
}

// histogram of the packed FIRST pass: digits of the keys themselves (the word is key << ib | position)
constexpr size_t scatter_lds_bytes(size_t key_size) {
  return (sizeof(int64_t) + key_size) * kTile + 16 + 256 * sizeof(int64_t) + (kWaves * 256 + 512) * sizeof(unsigned);
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

// ---- does the LDS rank a wave's conflicting returning atomics in lane order? -------------------------------------
// The scatter kernels' stable rank of a key among the equal digits of its wave costs 9 ballots per round of 64 keys and
// makes those kernels VALU-bound (2.6 - 3.3 TB/s of their traffic).  ds_add_rtn_u32 gives the same number in one
// instruction IF lanes that hit one address receive their return values in ascending lane order.  That is how the LDS
// of this chip resolves the conflict (tools/probe/lds_atomic_order.hip: 2.7e8 ranks, spread / few / constant /
// same-bank digits, no deviation), but it is not an architectural promise, so it is checked on the device the first
// time a sort runs there; a device that fails keeps the ballot variant.
__global__ __launch_bounds__(512) void lds_rank_probe_kernel(unsigned* __restrict__ bad, int rounds, unsigned seed) {
  __shared__ unsigned cnt[8][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = lane; j < 256; j += 64) cnt[wave][j] = 0;
  __syncthreads();
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  unsigned x = seed ^ (blockIdx.x * 9781u + threadIdx.x * 6271u + 1u);
  unsigned errors = 0;
  for (int r = 0; r < rounds; ++r) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const int mode = (blockIdx.x + r) & 3;  // spread, four values, constant, same bank
    const unsigned d = mode == 0 ? (x & 255u) : mode == 1 ? (x & 3u) : mode == 2 ? 7u : ((x & 15u) * 16u);
    unsigned long long peers = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const unsigned want = cnt[wave][d] + (unsigned)__popcll(peers & lt);
    __builtin_amdgcn_wave_barrier();
    const unsigned got = atomicAdd(&cnt[wave][d], 1u);
    __builtin_amdgcn_wave_barrier();
    errors += got != want;
  }
  if (errors) atomicAdd(bad, errors);
}

// The atomic in-wave rank is OPT-IN (environment PYG_HIP_SORT_ATOMIC_RANK=1): it is ~5 % faster, but it rests on an
// LDS property the ISA does not promise (conflicting ds_add_rtn_u32 of one wave return in lane order), and a stable
// LSD radix sort that loses stability in one pass is silently wrong.  The default is the ballot rank, which depends on
// nothing but the ISA.  With the variable set, the probe below still has to pass once per device and process (it
// allocates and synchronises a private stream: do not opt in under stream capture).
bool atomic_rank_requested() {
  static const bool on = [] {
    const char* e = getenv("PYG_HIP_SORT_ATOMIC_RANK");
    return e != nullptr && e[0] != '\0' && e[0] != '0';
  }();
  return on;
}

// per device, once per process; synchronises a private stream (never the caller's) on that first call
bool lds_rank_order_ok() {
  static std::mutex mu;
  static int state[64] = {0};  // 0 unknown, 1 yes, 2 no
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> lock(mu);
  if (state[dev] == 0) {
    state[dev] = 2;
    unsigned* bad = nullptr;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&bad), 4) == hipSuccess) {
      unsigned host = 1;
      if (hipMemsetAsync(bad, 0, 4, st) == hipSuccess) {
        hipLaunchKernelGGL(lds_rank_probe_kernel, dim3(device_info().num_cus * 2), dim3(512), 0, st, bad, 64, 0x9e3779b9u);
        if (hipGetLastError() == hipSuccess && hipMemcpyAsync(&host, bad, 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
            hipStreamSynchronize(st) == hipSuccess && host == 0)
          state[dev] = 1;
      }
    }
    if (bad) (void)hipFree(bad);
    if (st) (void)hipStreamDestroy(st);
  }
  return state[dev] == 1;
}

struct Plan {
  int64_t groups;  // workgroups per pass
  int64_t slice;   // keys per workgroup (multiple of kTile)
};

Plan make_plan(int64_t n) {
  const int64_t tiles = (n + kTile - 1) / kTile;
  int64_t groups = std::min<int64_t>(tiles, (int64_t)device_info().num_cus * 4);
  if (groups < 1) groups = 1;
  const int64_t tiles_per = std::max<int64_t>((tiles + groups - 1) / groups, 1);  // n == 0: one empty slice, no division by zero
  Plan p;
  p.slice = tiles_per * kTile;
  p.groups = (n + p.slice - 1) / p.slice;
  if (p.groups < 1) p.groups = 1;
  return p;
}

size_t ws_bytes(int64_t n, size_t key_size) {
  const Plan p = make_plan(n);
  size_t b = 0;
  (void)key_size;
  b += align_up((size_t)n * sizeof(int64_t), 256);                // key ping-pong buffer / packed words A
  b += align_up((size_t)n * sizeof(int64_t), 256);                // index ping-pong buffer / packed words B
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
  const bool atomic_rank = atomic_rank_requested() && lds_rank_order_ok();
  char* w = static_cast<char*>(ws);
  K* kbuf = reinterpret_cast<K*>(w);
  w += align_up((size_t)n * sizeof(int64_t), 256);
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
  bool nonneg = true;  // a caller-given max promises keys in [0, max] (the documented precondition)
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
      nonneg = false;
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
  // ---- packed mode -----------------------------------------------------------------------------------------
  // non-negative keys < 2^(8 * passes)
  {
    int ib = 1;
    while (ib < 63 && (1ull << ib) < (uint64_t)n) ++ib;  // bits of the largest position n - 1
    if (nonneg && 8 * passes + ib <= 64) {
      uint64_t* wbuf[2] = {reinterpret_cast<uint64_t*>(kbuf), reinterpret_cast<uint64_t*>(ibuf)};
      constexpr int plds = (int)scatter_packed_lds_bytes();
      const uint64_t* win = nullptr;
      for (int ps = 0; ps < passes; ++ps) {
        const bool first = ps == 0, last = ps == passes - 1;
        if (first)
          hipLaunchKernelGGL((hist_kernel<K, true>), dim3((unsigned)p.groups), dim3(kThreads), 0, stream, keys, n, p.slice, 0, hist);
        else
          hipLaunchKernelGGL((hist_kernel<uint64_t, true>), dim3((unsigned)p.groups), dim3(kThreads), 0, stream, win, n, p.slice,
                             ib + 8 * ps, hist);
        PYG_HIP_CHECK(hipGetLastError());
        const int64_t ntiles = (p.groups * 256 + kScanTile - 1) / kScanTile;
        int rc = device_scan<int64_t, SumOp>(HistLoad{hist}, HistStore{offs}, p.groups * 256, scan_tmp, scan_tmp + ntiles, stream);
        if (rc != PYG_HIP_OK) return rc;
        uint64_t* wout = wbuf[ps & 1];
        const int shift = ib + 8 * ps;
#define PYG_PACKED2(INK, OUTS, AR)                                                                                              \
  {                                                                                                                            \
    if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&scatter_packed_kernel<K, INK, OUTS, AR>), plds)) return rc_; \
    hipLaunchKernelGGL((scatter_packed_kernel<K, INK, OUTS, AR>), dim3((unsigned)p.groups), dim3(kSThreads), plds, stream, keys, \
                       win, wout, keys_out, idx_out, n, p.slice, shift, ib, offs);                                             \
  }
#define PYG_PACKED(INK, OUTS)                     \
  {                                               \
    if (atomic_rank) PYG_PACKED2(INK, OUTS, true) \
    else PYG_PACKED2(INK, OUTS, false)            \
  }
        if (first && last) PYG_PACKED(true, true)
        else if (first) PYG_PACKED(true, false)
        else if (last) PYG_PACKED(false, true)
        else PYG_PACKED(false, false)
#undef PYG_PACKED
#undef PYG_PACKED2
        PYG_HIP_CHECK(hipGetLastError());
        win = wout;
      }
      return PYG_HIP_OK;
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
    constexpr int lds = (int)scatter_lds_bytes(sizeof(K));
#define PYG_SCATTER(FIRSTP, AR, IDX)                                                                                        \
  {                                                                                                                     \
    if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&scatter_kernel<K, FIRSTP, AR>), lds)) return rc_;    \
    hipLaunchKernelGGL((scatter_kernel<K, FIRSTP, AR>), dim3((unsigned)p.groups), dim3(kSThreads), lds, stream, src_k, \
                       (const int64_t*)(IDX), kdst[cur], idst[cur], n, p.slice, shift, offs);                            \
  }
    if (ps == 0) {
      if (atomic_rank) PYG_SCATTER(true, true, nullptr)
      else PYG_SCATTER(true, false, nullptr)
    } else {
      if (atomic_rank) PYG_SCATTER(false, true, iin)
      else PYG_SCATTER(false, false, iin)
    }
#undef PYG_SCATTER
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
