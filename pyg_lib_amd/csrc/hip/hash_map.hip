// Device hash map key -> position (first occurrence) for gfx950: the kernels behind
// torch.classes.pyg.CUDAHashMap (pyg_lib/csrc/classes/cuda/hash_map.cu, which wraps
// cuco::static_map).  Open addressing with linear probing over a power-of-two table of 64-bit keys
// (int16 / int32 / int64 keys are widened), values claimed with atomicMin so that duplicate keys
// deterministically map to their first position.  The table lives in caller-owned memory.
#include "common.h"

namespace pyg_hip {
namespace {

typedef unsigned long long u64;
constexpr u64 kEmptyKey = 0x8000000000000000ull;  // numeric_limits<int64>::min(), the reference's sentinel

__device__ __forceinline__ u64 mix64(u64 x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

template <typename K>
__device__ __forceinline__ u64 widen(K k) {
  return (u64)(int64_t)k;
}

template <typename K>
__global__ void hash_insert_kernel(const K* __restrict__ keys, int64_t n, u64* __restrict__ tkeys,
                                   int64_t* __restrict__ tvals, u64 mask, int64_t* __restrict__ distinct) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 k = widen(keys[i]);
  u64 s = mix64(k) & mask;
  while (true) {
    const u64 cur = __hip_atomic_load(&tkeys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == k) break;
    if (cur == kEmptyKey) {
      u64 expected = kEmptyKey;
      if (__hip_atomic_compare_exchange_strong(&tkeys[s], &expected, k, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT)) {
        atomicAdd(reinterpret_cast<u64*>(distinct), 1ull);
        break;
      }
      if (expected == k) break;
    }
    s = (s + 1) & mask;
  }
  atomicMin(reinterpret_cast<long long*>(&tvals[s]), (long long)i);
}

template <typename K>
__global__ void hash_find_kernel(const K* __restrict__ query, int64_t m, const u64* __restrict__ tkeys,
                                 const int64_t* __restrict__ tvals, u64 mask, int64_t* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= m) return;
  const u64 k = widen(query[i]);
  u64 s = mix64(k) & mask;
  int64_t r = -1;
  while (true) {
    const u64 cur = tkeys[s];
    if (cur == k) {
      r = tvals[s];
      break;
    }
    if (cur == kEmptyKey) break;
    s = (s + 1) & mask;
  }
  out[i] = r;
}

__global__ void hash_clear_kernel(u64* tkeys, int64_t* tvals, int64_t slots, int64_t* distinct) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i == 0) *distinct = 0;
  if (i < slots) {
    tkeys[i] = kEmptyKey;
    tvals[i] = 0x7fffffffffffffffll;
  }
}

}  // namespace
}  // namespace pyg_hip

using namespace pyg_hip;

extern "C" {

int64_t pyg_hip_hash_map_slots(int64_t n, double load_factor) {
  if (!(load_factor > 0.0) || load_factor > 1.0) load_factor = 0.5;
  double want = (double)(n > 0 ? n : 1) / load_factor;
  int64_t slots = 16;
  while ((double)slots < want + 1.0) slots <<= 1;
  return slots;
}

int pyg_hip_hash_map_build(int key_dtype, const void* keys, int64_t n, uint64_t* table_keys, int64_t* table_vals,
                           int64_t slots, int64_t* distinct_dev, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(n >= 0 && slots > n && (slots & (slots - 1)) == 0, "hash_map: table must be a power of two > n");
  PYG_HIP_REQUIRE(table_keys && table_vals && distinct_dev && (n == 0 || keys), "hash_map: NULL argument");
  hipLaunchKernelGGL(hash_clear_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream,
                     reinterpret_cast<u64*>(table_keys), table_vals, slots, distinct_dev);
  if (n > 0) {
    const dim3 grid((unsigned)((n + 255) / 256));
    const u64 mask = (u64)slots - 1;
    u64* tk = reinterpret_cast<u64*>(table_keys);
    switch (key_dtype) {
      case PYG_I16: hipLaunchKernelGGL(hash_insert_kernel<int16_t>, grid, dim3(256), 0, stream, static_cast<const int16_t*>(keys), n, tk, table_vals, mask, distinct_dev); break;
      case PYG_I32: hipLaunchKernelGGL(hash_insert_kernel<int32_t>, grid, dim3(256), 0, stream, static_cast<const int32_t*>(keys), n, tk, table_vals, mask, distinct_dev); break;
      case PYG_I64: hipLaunchKernelGGL(hash_insert_kernel<int64_t>, grid, dim3(256), 0, stream, static_cast<const int64_t*>(keys), n, tk, table_vals, mask, distinct_dev); break;
      default: return fail(PYG_HIP_ERR_INVALID, "hash_map: keys must be int16, int32 or int64 (dtype %d)", key_dtype);
    }
  }
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

int pyg_hip_hash_map_get(int key_dtype, const void* query, int64_t m, const uint64_t* table_keys,
                         const int64_t* table_vals, int64_t slots, int64_t* out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(m >= 0 && slots > 0 && (slots & (slots - 1)) == 0, "hash_map: bad sizes");
  if (m == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(query && table_keys && table_vals && out, "hash_map: NULL argument");
  const dim3 grid((unsigned)((m + 255) / 256));
  const u64 mask = (u64)slots - 1;
  const u64* tk = reinterpret_cast<const u64*>(table_keys);
  switch (key_dtype) {
    case PYG_I16: hipLaunchKernelGGL(hash_find_kernel<int16_t>, grid, dim3(256), 0, stream, static_cast<const int16_t*>(query), m, tk, table_vals, mask, out); break;
    case PYG_I32: hipLaunchKernelGGL(hash_find_kernel<int32_t>, grid, dim3(256), 0, stream, static_cast<const int32_t*>(query), m, tk, table_vals, mask, out); break;
    case PYG_I64: hipLaunchKernelGGL(hash_find_kernel<int64_t>, grid, dim3(256), 0, stream, static_cast<const int64_t*>(query), m, tk, table_vals, mask, out); break;
    default: return fail(PYG_HIP_ERR_INVALID, "hash_map: keys must be int16, int32 or int64 (dtype %d)", key_dtype);
  }
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

}  // extern "C"
