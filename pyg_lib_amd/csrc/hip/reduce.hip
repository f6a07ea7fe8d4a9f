// scatter_{sum,mul,min,max}, segment_*_coo and gather_coo for gfx950 (MI355X).
//
// Replaces pyg_lib/csrc/ops/cuda/scatter_kernel.cu and segment_coo_kernel.cu (warp-32 shuffles
// and CAS-emulated atomics, "ROCm wave64 is not a target") and follows the CPU contracts of
// pyg_lib/csrc/ops/cpu/scatter_kernel.cpp and segment_coo_kernel.cpp.
//
// One (B, E, K) layout serves every op (scatter_kernel.cpp:16-24): src[b, e, k] is reduced into
// out[b, index(b, e, k), k].  The index is addressed through three element strides so that a 1-D
// index broadcast along B and K (the common PyG case) or a COO index of shape [B, E] is read in
// place: algorithmic traffic stays 8*E + s*E*K + s*N*K instead of materialising an int64 per
// element as the reference front does (ops/autograd/scatter_kernel.cpp:33-39).
//
// HBM-bound byte work; the levers are coalesced 16-byte rows and few, native atomics:
//   * sum (fp32 / bf16 / fp16 / fp64 / int32 / int64): each thread owns a 16-byte column slice and a
//     short run of consecutive rows, accumulates in fp32 while the index repeats (sorted COO input
//     collapses whole runs, matching segment_coo_kernel.cpp's run accumulation) and flushes with
//     native global atomics (global_atomic_add_f32 / _pk_add_bf16 / _pk_add_f16 / _add_f64 / _add_x2).
//   * min / max: value pass with native integer atomics (CAS loop on the reference's `<` / `>` for
//     floating types), then an arg pass atomicMin(arg, e) over the elements that equal the final
//     value and strictly improved the initial one -- exactly the CPU kernel's first-match rule
//     (scatter_kernel.cpp:249-369), so values AND arg indices are bit-exact.
//   * mul and the 8/16-bit integer types: CAS loop on the containing 32-bit word.
#include "common.h"
#include "elem.h"

#include <limits>
#include <type_traits>

namespace pyg_hip {
namespace {

// ---- atomic read-modify-write on any 1/2/4/8-byte element -----------------------------------------
// f(old) -> {changed, new}.  Loops on the containing 32-bit word for sub-word types.
template <typename T, typename F>
__device__ void atomic_rmw(T* addr, F f) {
  if constexpr (sizeof(T) == 8) {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(addr);
    unsigned long long old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (true) {
      T cur = __builtin_bit_cast(T, old);
      T nv;
      if (!f(cur, &nv)) return;
      unsigned long long want = __builtin_bit_cast(unsigned long long, nv);
      if (__hip_atomic_compare_exchange_strong(p, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT))
        return;
    }
  } else if constexpr (sizeof(T) == 4) {
    unsigned int* p = reinterpret_cast<unsigned int*>(addr);
    unsigned int old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (true) {
      T cur = __builtin_bit_cast(T, old);
      T nv;
      if (!f(cur, &nv)) return;
      unsigned int want = __builtin_bit_cast(unsigned int, nv);
      if (__hip_atomic_compare_exchange_strong(p, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT))
        return;
    }
  } else {
    const uintptr_t a = reinterpret_cast<uintptr_t>(addr);
    unsigned int* p = reinterpret_cast<unsigned int*>(a & ~(uintptr_t)3);
    const int shift = (int)(a & 3) * 8;
    constexpr unsigned int mask = sizeof(T) == 2 ? 0xffffu : 0xffu;
    unsigned int old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (true) {
      using U = typename std::conditional<sizeof(T) == 2, uint16_t, uint8_t>::type;
      const U curbits = (U)((old >> shift) & mask);
      T cur = __builtin_bit_cast(T, curbits);
      T nv;
      if (!f(cur, &nv)) return;
      const unsigned int want =
          (old & ~(mask << shift)) | ((unsigned int)__builtin_bit_cast(U, nv) << shift);
      if (__hip_atomic_compare_exchange_strong(p, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT))
        return;
    }
  }
}

// CAS = true (PYG_HIP_SCATTER_CAS / PYG_HIP_FLOAT_ATOMICS=cas): float and double adds through a compare-and-swap loop
// instead of the hardware's floating-point atomic unit (same sum, same one rounding per add).
template <typename T, bool CAS = false>
__device__ void atomic_add(T* addr, typename Math<T>::acc_t v) {
  if constexpr (CAS && (std::is_same<T, float>::value || std::is_same<T, double>::value)) {
    atomic_rmw(addr, [v](T cur, T* nv) {
      *nv = cur + v;
      return true;
    });
  } else if constexpr (std::is_same<T, float>::value) {
    unsafeAtomicAdd(addr, v);  // global_atomic_add_f32
  } else if constexpr (std::is_same<T, double>::value) {
    unsafeAtomicAdd(addr, v);  // global_atomic_add_f64
  } else if constexpr (std::is_same<T, int32_t>::value) {
    atomicAdd(addr, v);
  } else if constexpr (std::is_same<T, int64_t>::value) {
    atomicAdd(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)v);
  } else {
    atomic_rmw(addr, [v](T cur, T* nv) {
      *nv = Math<T>::down((typename Math<T>::acc_t)(Math<T>::up(cur) + v));
      return true;
    });
  }
}

template <typename T>
__device__ void atomic_mul(T* addr, T v) {
  atomic_rmw(addr, [v](T cur, T* nv) {
    *nv = Math<T>::down((typename Math<T>::acc_t)(Math<T>::up(cur) * Math<T>::up(v)));
    return true;
  });
}

template <typename T, bool IS_MIN>
__device__ void atomic_minmax(T* addr, T v) {
  if constexpr (std::is_same<T, int32_t>::value) {
    if (IS_MIN) atomicMin(addr, v); else atomicMax(addr, v);
  } else if constexpr (std::is_same<T, int64_t>::value) {
    if (IS_MIN) atomicMin(reinterpret_cast<long long*>(addr), (long long)v);
    else atomicMax(reinterpret_cast<long long*>(addr), (long long)v);
  } else {
    // the reference's strict comparison (v < *slot / v > *slot), NaNs never win
    atomic_rmw(addr, [v](T cur, T* nv) {
      const bool better = IS_MIN ? (Math<T>::up(v) < Math<T>::up(cur)) : (Math<T>::up(v) > Math<T>::up(cur));
      *nv = v;
      return better;
    });
  }
}

struct Shape {
  int64_t B, E, K, N;
  int64_t isb, ise, isk;  // index strides (elements) along b, e, k
};

__device__ __forceinline__ int64_t index_at(const int64_t* index, const Shape& s, int64_t b, int64_t e,
                                            int64_t k) {
  return index[b * s.isb + e * s.ise + k * s.isk];
}

// ---- generic element-per-thread kernels -------------------------------------------------------------
template <typename T, int OP, bool CAS = false>
__global__ void scatter_elem_kernel(const T* __restrict__ src, const int64_t* __restrict__ index, T* out,
                                    Shape s) {
  const int64_t total = s.B * s.E * s.K;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = i % s.K;
    const int64_t e = (i / s.K) % s.E;
    const int64_t b = i / (s.K * s.E);
    const int64_t idx = index_at(index, s, b, e, k);
    T* dst = out + (b * s.N + idx) * s.K + k;
    const T v = src[i];
    if (OP == OP_SUM) atomic_add<T, CAS>(dst, Math<T>::up(v));
    else if (OP == OP_MUL) atomic_mul<T>(dst, v);
    else if (OP == OP_MIN) atomic_minmax<T, true>(dst, v);
    else atomic_minmax<T, false>(dst, v);
  }
}

// arg pass of min/max: first source position whose value equals the final bucket value, provided the
// bucket strictly improved on its initial state (init == nullptr: the type's max()/lowest()).
template <typename T, bool IS_MIN>
__global__ void scatter_arg_kernel(const T* __restrict__ src, const int64_t* __restrict__ index,
                                   const T* __restrict__ out, const T* __restrict__ init,
                                   int64_t* arg, Shape s) {
  const int64_t total = s.B * s.E * s.K;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = i % s.K;
    const int64_t e = (i / s.K) % s.E;
    const int64_t b = i / (s.K * s.E);
    const int64_t idx = index_at(index, s, b, e, k);
    const int64_t o = (b * s.N + idx) * s.K + k;
    const T fin = out[o];
    const T v = src[i];
    if (!bits_equal(v, fin)) continue;
    const T start = init ? init[o] : (IS_MIN ? type_max<T>() : type_lowest<T>());
    const bool improved = IS_MIN ? (Math<T>::up(fin) < Math<T>::up(start)) : (Math<T>::up(fin) > Math<T>::up(start));
    if (improved) atomicMin(reinterpret_cast<long long*>(arg + o), (long long)e);
  }
}

// empty buckets (arg still the sentinel E) of a freshly allocated min/max output are reset to 0
template <typename T>
__global__ void reset_empty_kernel(T* out, const int64_t* __restrict__ arg, int64_t n, int64_t sentinel) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n && arg[i] == sentinel) out[i] = Math<T>::down((typename Math<T>::acc_t)0);
}

template <typename T>
__global__ void fill_kernel(T* out, int64_t n, T v) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = v;
}

__global__ void fill_i64_kernel(int64_t* out, int64_t n, int64_t v) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = v;
}

// ---- vectorised sum: 16-byte column slices, run accumulation over consecutive rows ---------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <typename T>
struct Vec;  // VEC elements per 16 bytes, accumulate in float/double/int
template <>
struct Vec<float> {
  static constexpr int N = 4;
  using acc_t = float;
  __device__ static void unpack(u32x4 v, float* a) {
    // (bit_cast straight from a vector-element lvalue miscompiles to element 0: copy out first)
    for (int i = 0; i < 4; ++i) {
      const uint32_t w = v[i];
      a[i] += __builtin_bit_cast(float, w);
    }
  }
  __device__ static void flush(float* dst, const float* a) {
    for (int i = 0; i < 4; ++i) unsafeAtomicAdd(dst + i, a[i]);
  }
  __device__ static void flush_cas(float* dst, const float* a) {
    for (int i = 0; i < 4; ++i) atomic_add<float, true>(dst + i, a[i]);
  }
  __device__ static void add_plain(float* dst, const float* a) {  // rows owned by this thread only
    float4 v = *reinterpret_cast<float4*>(dst);
    v.x += a[0]; v.y += a[1]; v.z += a[2]; v.w += a[3];
    *reinterpret_cast<float4*>(dst) = v;
  }
};
template <>
struct Vec<bf16_t> {
  static constexpr int N = 8;
  using acc_t = float;
  __device__ static void unpack(u32x4 v, float* a) {
    for (int i = 0; i < 4; ++i) {
      a[2 * i] += __builtin_bit_cast(float, v[i] << 16);
      a[2 * i + 1] += __builtin_bit_cast(float, v[i] & 0xffff0000u);
    }
  }
  __device__ static void flush(bf16_t* dst, const float* a) {
    for (int i = 0; i < 4; ++i) {
      bf16x2 p = {(__bf16)a[2 * i], (__bf16)a[2 * i + 1]};
      // global_atomic_pk_add_bf16
      (void)__builtin_amdgcn_global_atomic_fadd_v2bf16(
          (__attribute__((address_space(1))) bf16x2*)(dst + 2 * i), p);
    }
  }
  // the same packed add as a CAS loop on the pair's 32-bit word: each half = round(bf16(half) + bf16(a))
  __device__ static void flush_cas(bf16_t* dst, const float* a) {
    for (int i = 0; i < 4; ++i) {
      const float p0 = (float)(__bf16)a[2 * i], p1 = (float)(__bf16)a[2 * i + 1];
      atomic_rmw(reinterpret_cast<uint32_t*>(dst + 2 * i), [p0, p1](uint32_t cur, uint32_t* nv) {
        const uint16_t lo = __builtin_bit_cast(uint16_t, (__bf16)(__builtin_bit_cast(float, cur << 16) + p0));
        const uint16_t hi = __builtin_bit_cast(uint16_t, (__bf16)(__builtin_bit_cast(float, cur & 0xffff0000u) + p1));
        *nv = (uint32_t)lo | ((uint32_t)hi << 16);
        return true;
      });
    }
  }
  __device__ static void add_plain(bf16_t* dst, const float* a) {
    float cur[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unpack(*reinterpret_cast<const u32x4*>(dst), cur);
    u32x4 o;
    for (int i = 0; i < 4; ++i) {
      const uint16_t lo = __builtin_bit_cast(uint16_t, (__bf16)(cur[2 * i] + a[2 * i]));
      const uint16_t hi = __builtin_bit_cast(uint16_t, (__bf16)(cur[2 * i + 1] + a[2 * i + 1]));
      o[i] = (uint32_t)lo | ((uint32_t)hi << 16);
    }
    *reinterpret_cast<u32x4*>(dst) = o;
  }
};
template <>
struct Vec<f16_t> {
  static constexpr int N = 8;
  using acc_t = float;
  __device__ static void unpack(u32x4 v, float* a) {
    for (int i = 0; i < 4; ++i) {
      a[2 * i] += (float)__builtin_bit_cast(_Float16, (uint16_t)(v[i] & 0xffffu));
      a[2 * i + 1] += (float)__builtin_bit_cast(_Float16, (uint16_t)(v[i] >> 16));
    }
  }
  __device__ static void flush(f16_t* dst, const float* a) {
    for (int i = 0; i < 4; ++i) {
      f16x2 p = {(_Float16)a[2 * i], (_Float16)a[2 * i + 1]};
      (void)__builtin_amdgcn_global_atomic_fadd_v2f16(
          (__attribute__((address_space(1))) f16x2*)(dst + 2 * i), p);
    }
  }
  __device__ static void flush_cas(f16_t* dst, const float* a) {
    for (int i = 0; i < 4; ++i) {
      const float p0 = (float)(_Float16)a[2 * i], p1 = (float)(_Float16)a[2 * i + 1];
      atomic_rmw(reinterpret_cast<uint32_t*>(dst + 2 * i), [p0, p1](uint32_t cur, uint32_t* nv) {
        const uint16_t lo = __builtin_bit_cast(uint16_t, (_Float16)((float)__builtin_bit_cast(_Float16, (uint16_t)(cur & 0xffffu)) + p0));
        const uint16_t hi = __builtin_bit_cast(uint16_t, (_Float16)((float)__builtin_bit_cast(_Float16, (uint16_t)(cur >> 16)) + p1));
        *nv = (uint32_t)lo | ((uint32_t)hi << 16);
        return true;
      });
    }
  }
  __device__ static void add_plain(f16_t* dst, const float* a) {
    float cur[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unpack(*reinterpret_cast<const u32x4*>(dst), cur);
    u32x4 o;
    for (int i = 0; i < 4; ++i) {
      const uint16_t lo = __builtin_bit_cast(uint16_t, (_Float16)(cur[2 * i] + a[2 * i]));
      const uint16_t hi = __builtin_bit_cast(uint16_t, (_Float16)(cur[2 * i + 1] + a[2 * i + 1]));
      o[i] = (uint32_t)lo | ((uint32_t)hi << 16);
    }
    *reinterpret_cast<u32x4*>(dst) = o;
  }
};

// Requires K % Vec<T>::N == 0, 16-byte aligned src/out, index constant along k (isk == 0).
// SORTED: the index is ascending along e (segment_*_coo, or scatter through a sort permutation).  Then
// a thread owns 32 consecutive positions, every run strictly inside its chunk belongs to it alone and is
// added with a plain 16-byte read-modify-write; only the first and last run of a chunk (which may
// continue in the neighbouring chunks) use atomics.  Unsorted input: 8 positions, every flush atomic.
// perm (optional): source row of sorted position e (scatter via index_sort); index is then the SORTED key.
template <typename T, bool SORTED, bool CAS = false>
__global__ __launch_bounds__(256) void scatter_sum_vec_kernel(const T* __restrict__ src,
                                                              const int64_t* __restrict__ index,
                                                              const int64_t* __restrict__ perm, T* out,
                                                              Shape s) {
  constexpr int VN = Vec<T>::N;
  constexpr int R = SORTED ? 32 : 8;
  const int64_t kv = s.K / VN;  // 16-byte slices per row
  const int64_t chunks = (s.E + R - 1) / R;
  const int64_t total = s.B * chunks * kv;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = t % kv;
    const int64_t ch = (t / kv) % chunks;
    const int64_t b = t / (kv * chunks);
    const int64_t e0 = ch * R;
    const int64_t e1 = min(e0 + R, s.E);
    float acc[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) acc[i] = 0.f;
    int64_t cur = index[b * s.isb + e0 * s.ise];
    bool first = true;
    // batches of 8 positions: their indices and 16-byte source slices are all requested before the first is consumed
    // (one dependent load per trip left the sorted sums at 4.3 TB/s: a wave had 1 KiB in flight)
    constexpr int U = 8;
    for (int64_t eb = e0; eb < e1; eb += U) {
      int64_t idxv[U];
      u32x4 val[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t e = eb + u < e1 ? eb + u : e1 - 1;
        idxv[u] = index[b * s.isb + e * s.ise];
        const int64_t srow = perm ? perm[e] : (b * s.E + e);
        val[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + srow * s.K + c * VN));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (eb + u >= e1) break;
        const int64_t idx = idxv[u];
        if (idx != cur) {
          T* dst = out + (b * s.N + cur) * s.K + c * VN;
          if (SORTED && !first) Vec<T>::add_plain(dst, acc);
          else if (CAS) Vec<T>::flush_cas(dst, acc);
          else Vec<T>::flush(dst, acc);
          first = false;
#pragma unroll
          for (int i = 0; i < VN; ++i) acc[i] = 0.f;
          cur = idx;
        }
        Vec<T>::unpack(val[u], acc);
      }
    }
    if (CAS) Vec<T>::flush_cas(out + (b * s.N + cur) * s.K + c * VN, acc);
    else Vec<T>::flush(out + (b * s.N + cur) * s.K + c * VN, acc);
  }
}

// Unsorted 16-bit rows of an EVEN number of elements that the 16-byte-slice kernel above does not take (K = 2, 4, 6, 10, ...,
// and the narrow rows K <= 32): one (edge, pair) per thread, neighbouring lanes on neighbouring words, one packed atomic per
// pair instead of two 16-bit CAS loops on the same word (bf16 K = 4, 20 M edges: 2.5 ms -> 1.0).
template <typename T, bool CAS = false>
__global__ __launch_bounds__(256) void scatter_sum_pair_kernel(const T* __restrict__ src, const int64_t* __restrict__ index,
                                                               T* out, Shape s) {
  const int64_t kp = s.K / 2;
  const int64_t total = s.B * s.E * kp;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = t % kp;
    const int64_t e = (t / kp) % s.E;
    const int64_t b = t / (kp * s.E);
    const int64_t idx = index[b * s.isb + e * s.ise];
    const uint32_t w = *reinterpret_cast<const uint32_t*>(src + (b * s.E + e) * s.K + 2 * c);
    T* dst = out + (b * s.N + idx) * s.K + 2 * c;
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (CAS) {
        const float p0 = __builtin_bit_cast(float, w << 16), p1 = __builtin_bit_cast(float, w & 0xffff0000u);
        atomic_rmw(reinterpret_cast<uint32_t*>(dst), [p0, p1](uint32_t cur, uint32_t* nv) {
          const uint16_t lo = __builtin_bit_cast(uint16_t, (__bf16)(__builtin_bit_cast(float, cur << 16) + p0));
          const uint16_t hi = __builtin_bit_cast(uint16_t, (__bf16)(__builtin_bit_cast(float, cur & 0xffff0000u) + p1));
          *nv = (uint32_t)lo | ((uint32_t)hi << 16);
          return true;
        });
      } else {
        (void)__builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) bf16x2*)dst, __builtin_bit_cast(bf16x2, w));
      }
    } else {
      if (CAS) {
        const float p0 = (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)), p1 = (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
        atomic_rmw(reinterpret_cast<uint32_t*>(dst), [p0, p1](uint32_t cur, uint32_t* nv) {
          const uint16_t lo = __builtin_bit_cast(uint16_t, (_Float16)((float)__builtin_bit_cast(_Float16, (uint16_t)(cur & 0xffffu)) + p0));
          const uint16_t hi = __builtin_bit_cast(uint16_t, (_Float16)((float)__builtin_bit_cast(_Float16, (uint16_t)(cur >> 16)) + p1));
          *nv = (uint32_t)lo | ((uint32_t)hi << 16);
          return true;
        });
      } else {
        (void)__builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) f16x2*)dst, __builtin_bit_cast(f16x2, w));
      }
    }
  }
}

// ---- gather_coo -----------------------------------------------------------------------------------------
template <typename T>
__global__ void gather_elem_kernel(const T* __restrict__ src, const int64_t* __restrict__ index, T* out,
                                   int64_t B, int64_t E, int64_t K, int64_t N) {
  const int64_t total = B * E * K;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = i % K;
    const int64_t e = (i / K) % E;
    const int64_t b = i / (K * E);
    out[i] = src[(b * N + index[b * E + e]) * K + k];
  }
}

// 16-byte slices: `kv` slices per row
__global__ void gather_vec_kernel(const u32x4* __restrict__ src, const int64_t* __restrict__ index,
                                  u32x4* out, int64_t B, int64_t E, int64_t kv, int64_t N) {
  const int64_t total = B * E * kv;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i % kv;
    const int64_t e = (i / kv) % E;
    const int64_t b = i / (kv * E);
    __builtin_nontemporal_store(src[(b * N + index[b * E + e]) * kv + c], out + i);
  }
}

// ---- host dispatch ----------------------------------------------------------------------------------------
inline unsigned grid_for(int64_t n, int threads = 256) {
  int64_t blocks = (n + threads - 1) / threads;
  const int64_t cap = (int64_t)device_info().num_cus * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// workspace of the sort-based scatter: sorted keys + permutation + index_sort's own workspace
inline size_t scatter_sort_ws_bytes(int64_t E) {
  return 2 * align_up(sizeof(int64_t) * (size_t)(E > 0 ? E : 1), 256) + index_sort_ws_bytes_i64(E);
}
// ... followed by the CSR pointer of the (sorted) index: B * (N + 1) offsets
inline size_t scatter_indptr_bytes(int64_t B, int64_t N) {
  return align_up(sizeof(int64_t) * (size_t)(B > 0 ? B : 1) * (size_t)(N + 1), 256);
}

// indptr[b, r] = first position e of row b with index[b, e] >= r (index ascending along e)
__global__ void coo_indptr_kernel(const int64_t* __restrict__ index, int64_t isb, int64_t ise, int64_t B, int64_t E,
                                  int64_t N, int64_t* __restrict__ indptr) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= B * (N + 1)) return;
  const int64_t b = t / (N + 1), r = t % (N + 1);
  const int64_t* ip = index + b * isb;
  int64_t lo = 0, hi = E;
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (ip[mid * ise] < r) lo = mid + 1; else hi = mid;
  }
  indptr[t] = lo;
}

template <typename T>
constexpr int dtype_of();
template <> constexpr int dtype_of<float>() { return PYG_F32; }
template <> constexpr int dtype_of<double>() { return PYG_F64; }
template <> constexpr int dtype_of<f16_t>() { return PYG_F16; }
template <> constexpr int dtype_of<bf16_t>() { return PYG_BF16; }
template <> constexpr int dtype_of<int8_t>() { return PYG_I8; }
template <> constexpr int dtype_of<uint8_t>() { return PYG_U8; }
template <> constexpr int dtype_of<int16_t>() { return PYG_I16; }
template <> constexpr int dtype_of<int32_t>() { return PYG_I32; }
template <> constexpr int dtype_of<int64_t>() { return PYG_I64; }

template <typename T>
int run_scatter(int op, const void* src_, const int64_t* index, void* out_, int64_t* arg, const void* init_,
                const Shape& s, int sorted, void* ws, size_t ws_bytes, hipStream_t stream) {
  const T* src = static_cast<const T*>(src_);
  T* out = static_cast<T*>(out_);
  const T* init = static_cast<const T*>(init_);
  const int64_t total = s.B * s.E * s.K;
  const int64_t outn = s.B * s.N * s.K;
  // PYG_HIP_SCATTER_FRESH_SUM: `out` of a sum is uninitialised.  The sorted (CSR-row) path writes every slot and never
  // reads it; every other path accumulates into zeros, cleared here.
  const bool fresh_sum = op == OP_SUM && (sorted & PYG_HIP_SCATTER_FRESH_SUM) != 0;
  const bool cas = (sorted & PYG_HIP_SCATTER_CAS) != 0 || float_atomic_mode() == 1;
  const bool det = (sorted & PYG_HIP_SCATTER_DETERMINISTIC) != 0;
  sorted &= PYG_HIP_SCATTER_SORTED;
  const bool csr_rows = op == OP_SUM && sorted && s.isk == 0 && ws && ws_bytes >= scatter_indptr_bytes(s.B, s.N) && total > 0;
  // one large unsorted index vector, rows of >= 64 bytes: sort the E indices once (3-4 radix passes over 16 E bytes),
  // buckets become CSR rows summed through the permutation in SOURCE order (the stable sort keeps it): no atomics,
  // deterministic, every output row written once
  const bool float_t = std::is_same<T, float>::value || std::is_same<T, bf16_t>::value || std::is_same<T, f16_t>::value;
  // PYG_HIP_SCATTER_DETERMINISTIC: the same path for ANY size, row width and floating type (float64 included)
  const bool floating = float_t || std::is_same<T, double>::value;
  const bool sort_rows = op == OP_SUM && !sorted && s.isk == 0 && s.B == 1 && s.ise == 1 && ws &&
                         ws_bytes >= scatter_sort_ws_bytes(s.E) + scatter_indptr_bytes(1, s.N) &&
                         ((float_t && s.E >= (1 << 15) && s.K * (int64_t)sizeof(T) >= 64) || (det && floating && s.E > 0));
  if (det && floating && total > 0 && ((op == OP_SUM && !csr_rows && !sort_rows) || op == OP_MUL))
    return fail(PYG_HIP_ERR_UNSUPPORTED,
                "scatter: no atomic-free kernel for this reduction / index layout (PYG_HIP_SCATTER_DETERMINISTIC: floating sums "
                "need an index broadcast along k -- sorted, or one unsorted vector -- and the caller's workspace)");
  if (fresh_sum && !csr_rows && !sort_rows && outn > 0) PYG_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(T) * (size_t)outn, stream));
  if (total == 0) return PYG_HIP_OK;
  if (op == OP_SUM && !csr_rows && !sort_rows && (std::is_floating_point<T>::value || float_t))
    note_accumulate("pyg_hip_scatter (sum, atomic kernels)", out, sizeof(T) * (size_t)outn,
                    fresh_sum ? "this call (hipMemsetAsync on the call's stream)" : "the caller (`out=` accumulation)", stream, cas ? 1 : 0);
  const unsigned grid = grid_for(total);
  if (csr_rows) {
    // COO contract (index ascending along e): buckets are CSR rows -- summed in source order in opmath,
    // seeded from `out`, no atomics (the run accumulation of segment_coo_kernel.cpp:104-166, bit for bit)
    int64_t* indptr = reinterpret_cast<int64_t*>(ws);
    const int64_t n = s.B * (s.N + 1);
    hipLaunchKernelGGL(coo_indptr_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, index, s.isb, s.ise,
                       s.B, s.E, s.N, indptr);
    PYG_HIP_CHECK(hipGetLastError());
    // (what the workspace holds behind the offsets is scratch for hub rows)
    const size_t ip_bytes = scatter_indptr_bytes(s.B, s.N);
    return segment_csr_sum(dtype_of<T>(), src, indptr, s.N + 1, nullptr, out, s.B, s.N, s.E, s.K, fresh_sum ? 1 : 0, stream,
                           static_cast<char*>(ws) + ip_bytes, ws_bytes - ip_bytes);
  }
  if (sort_rows) {
    char* w = static_cast<char*>(ws);
    const size_t sort_bytes = scatter_sort_ws_bytes(s.E);
    int64_t* keys = reinterpret_cast<int64_t*>(w);
    int64_t* perm = reinterpret_cast<int64_t*>(w + align_up(sizeof(int64_t) * (size_t)s.E, 256));
    void* sws = w + 2 * align_up(sizeof(int64_t) * (size_t)s.E, 256);
    int64_t* indptr = reinterpret_cast<int64_t*>(w + sort_bytes);
    int rc = index_sort_i64(index, s.E, s.N > 0 ? s.N - 1 : 0, keys, perm, sws,
                            sort_bytes - 2 * align_up(sizeof(int64_t) * (size_t)s.E, 256), stream);
    if (rc != PYG_HIP_OK) return rc;
    hipLaunchKernelGGL(coo_indptr_kernel, dim3((unsigned)((s.N + 1 + 255) / 256)), dim3(256), 0, stream,
                       (const int64_t*)keys, (int64_t)0, (int64_t)1, (int64_t)1, s.E, s.N, indptr);
    PYG_HIP_CHECK(hipGetLastError());
    // (the sorted keys are dead once the offsets exist: scratch for hub rows)
    return segment_csr_sum(dtype_of<T>(), src, indptr, s.N + 1, perm, out, 1, s.N, s.E, s.K, fresh_sum ? 1 : 0, stream, keys,
                           align_up(sizeof(int64_t) * (size_t)s.E, 256));
  }
  if (op == OP_SUM) {
    if constexpr (std::is_same<T, float>::value || std::is_same<T, bf16_t>::value ||
                  std::is_same<T, f16_t>::value) {
      // (unsorted NARROW rows -- up to four 16-byte slices -- do not come here: a thread of the kernel below owns 8 consecutive
      // edges of one slice, so with few slices per row its lanes are 128+ bytes apart on every load, and every lane's four
      // atomics hit a line of their own.  One element / one packed pair per thread -- the kernels further down -- puts
      // neighbouring lanes on neighbouring words of the same row: K = 4 floats, 20 M edges: 3.9 ms here, 0.96 there (=
      // torch.index_add_); bf16 K = 16: 3.9 -> 1.0.)
      const bool narrow = !sorted && s.K / Vec<T>::N <= 4;
      if (s.isk == 0 && s.K % Vec<T>::N == 0 && aligned16(src) && aligned16(out) && !narrow) {
        if (sorted) {
          const int64_t threads = s.B * ((s.E + 31) / 32) * (s.K / Vec<T>::N);
          if (cas)
            hipLaunchKernelGGL((scatter_sum_vec_kernel<T, true, true>), dim3(grid_for(threads)), dim3(256), 0, stream, src,
                               index, (const int64_t*)nullptr, out, s);
          else
            hipLaunchKernelGGL((scatter_sum_vec_kernel<T, true>), dim3(grid_for(threads)), dim3(256), 0, stream,
                               src, index, (const int64_t*)nullptr, out, s);
          PYG_HIP_CHECK(hipGetLastError());
          return PYG_HIP_OK;
        }
        const int64_t threads = s.B * ((s.E + 7) / 8) * (s.K / Vec<T>::N);
        if (cas)
          hipLaunchKernelGGL((scatter_sum_vec_kernel<T, false, true>), dim3(grid_for(threads)), dim3(256), 0, stream, src,
                             index, (const int64_t*)nullptr, out, s);
        else
          hipLaunchKernelGGL((scatter_sum_vec_kernel<T, false>), dim3(grid_for(threads)), dim3(256), 0, stream,
                             src, index, (const int64_t*)nullptr, out, s);
        PYG_HIP_CHECK(hipGetLastError());
        return PYG_HIP_OK;
      }
    }
    if constexpr (std::is_same<T, bf16_t>::value || std::is_same<T, f16_t>::value) {
      // 16-bit rows of an even number of elements: packed pairs (4-byte aligned: K even, bases 4-byte aligned)
      if (s.isk == 0 && s.K % 2 == 0 && !sorted && (reinterpret_cast<uintptr_t>(src) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0) {
        const int64_t threads = s.B * s.E * (s.K / 2);
        if (cas)
          hipLaunchKernelGGL((scatter_sum_pair_kernel<T, true>), dim3(grid_for(threads)), dim3(256), 0, stream, src, index, out, s);
        else
          hipLaunchKernelGGL((scatter_sum_pair_kernel<T, false>), dim3(grid_for(threads)), dim3(256), 0, stream, src, index, out, s);
        PYG_HIP_CHECK(hipGetLastError());
        return PYG_HIP_OK;
      }
    }
    if constexpr (std::is_same<T, float>::value || std::is_same<T, double>::value) {
      if (cas) {
        hipLaunchKernelGGL((scatter_elem_kernel<T, OP_SUM, true>), dim3(grid), dim3(256), 0, stream, src, index, out, s);
        PYG_HIP_CHECK(hipGetLastError());
        return PYG_HIP_OK;
      }
    }
    hipLaunchKernelGGL((scatter_elem_kernel<T, OP_SUM>), dim3(grid), dim3(256), 0, stream, src, index, out, s);
  } else if (op == OP_MUL) {
    hipLaunchKernelGGL((scatter_elem_kernel<T, OP_MUL>), dim3(grid), dim3(256), 0, stream, src, index, out, s);
  } else if (op == OP_MIN || op == OP_MAX) {
    PYG_HIP_REQUIRE(arg != nullptr, "scatter_min/max: 'arg_out' is NULL");
    // Atomic-free path: with the index ascending along e (the COO contract), or after sorting one
    // index vector, buckets are CSR rows -- one thread per (bucket, 16-byte slice) walks its row in
    // source order with the reference's strict compare (values and first-match arg exact, no CAS
    // loops, no second pass).  Needs an index broadcast along k and the caller's workspace.
    if (s.isk == 0 && ws) {
      char* w = static_cast<char*>(ws);
      const size_t ip_bytes = scatter_indptr_bytes(s.B, s.N);
      if (sorted && ws_bytes >= ip_bytes) {
        int64_t* indptr = reinterpret_cast<int64_t*>(w);
        const int64_t n = s.B * (s.N + 1);
        hipLaunchKernelGGL(coo_indptr_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, index, s.isb,
                           s.ise, s.B, s.E, s.N, indptr);
        PYG_HIP_CHECK(hipGetLastError());
        return segment_csr_minmax(op == OP_MIN, dtype_of<T>(), src, indptr, s.N + 1, nullptr, out, arg, init ? 0 : 1,
                                  s.B, s.N, s.E, s.K, stream, w + ip_bytes, ws_bytes - ip_bytes);
      }
      const size_t sort_bytes = scatter_sort_ws_bytes(s.E);
      if (!sorted && s.B == 1 && s.ise == 1 && s.E >= (1 << 15) && ws_bytes >= sort_bytes + scatter_indptr_bytes(1, s.N)) {
        int64_t* keys = reinterpret_cast<int64_t*>(w);
        int64_t* perm = reinterpret_cast<int64_t*>(w + align_up(sizeof(int64_t) * (size_t)s.E, 256));
        void* sws = w + 2 * align_up(sizeof(int64_t) * (size_t)s.E, 256);
        int64_t* indptr = reinterpret_cast<int64_t*>(w + sort_bytes);
        int rc = index_sort_i64(index, s.E, s.N > 0 ? s.N - 1 : 0, keys, perm, sws,
                                sort_bytes - 2 * align_up(sizeof(int64_t) * (size_t)s.E, 256), stream);
        if (rc != PYG_HIP_OK) return rc;
        hipLaunchKernelGGL(coo_indptr_kernel, dim3((unsigned)((s.N + 1 + 255) / 256)), dim3(256), 0, stream,
                           (const int64_t*)keys, (int64_t)0, (int64_t)1, (int64_t)1, s.E, s.N, indptr);
        PYG_HIP_CHECK(hipGetLastError());
        return segment_csr_minmax(op == OP_MIN, dtype_of<T>(), src, indptr, s.N + 1, perm, out, arg, init ? 0 : 1, 1,
                                  s.N, s.E, s.K, stream, keys, align_up(sizeof(int64_t) * (size_t)s.E, 256));
      }
    }
    hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)((outn + 255) / 256)), dim3(256), 0, stream, arg, outn,
                       s.E);
    if (op == OP_MIN) {
      hipLaunchKernelGGL((scatter_elem_kernel<T, OP_MIN>), dim3(grid), dim3(256), 0, stream, src, index, out, s);
      hipLaunchKernelGGL((scatter_arg_kernel<T, true>), dim3(grid), dim3(256), 0, stream, src, index, out, init,
                         arg, s);
    } else {
      hipLaunchKernelGGL((scatter_elem_kernel<T, OP_MAX>), dim3(grid), dim3(256), 0, stream, src, index, out, s);
      hipLaunchKernelGGL((scatter_arg_kernel<T, false>), dim3(grid), dim3(256), 0, stream, src, index, out, init,
                         arg, s);
    }
    if (!init)
      hipLaunchKernelGGL((reset_empty_kernel<T>), dim3((unsigned)((outn + 255) / 256)), dim3(256), 0, stream, out,
                         arg, outn, s.E);
  } else {
    return fail(PYG_HIP_ERR_INVALID, "scatter: unknown reduce op %d", op);
  }
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

template <typename T>
int run_fill_extreme(int op, void* out_, int64_t n, hipStream_t stream) {
  if (n == 0) return PYG_HIP_OK;
  T* out = static_cast<T*>(out_);
  T v;
  // host-side constants (the device helpers are not callable here)
  if constexpr (std::is_same<T, bf16_t>::value) v = bf16_t{(uint16_t)(op == OP_MIN ? 0x7f7f : 0xff7f)};
  else if constexpr (std::is_same<T, f16_t>::value) v = f16_t{(uint16_t)(op == OP_MIN ? 0x7bff : 0xfbff)};
  else v = op == OP_MIN ? std::numeric_limits<T>::max() : std::numeric_limits<T>::lowest();
  hipLaunchKernelGGL((fill_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out, n, v);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

template <typename T>
int run_gather(const void* src, const int64_t* index, void* out, int64_t B, int64_t E, int64_t K, int64_t N,
               hipStream_t stream) {
  const int64_t total = B * E * K;
  if (total == 0) return PYG_HIP_OK;
  const int64_t row_bytes = K * (int64_t)sizeof(T);
  if (row_bytes % 16 == 0 && aligned16(src) && aligned16(out)) {
    const int64_t kv = row_bytes / 16;
    hipLaunchKernelGGL(gather_vec_kernel, dim3(grid_for(B * E * kv)), dim3(256), 0, stream,
                       static_cast<const u32x4*>(src), index, static_cast<u32x4*>(out), B, E, kv, N);
  } else {
    hipLaunchKernelGGL((gather_elem_kernel<T>), dim3(grid_for(total)), dim3(256), 0, stream,
                       static_cast<const T*>(src), index, static_cast<T*>(out), B, E, K, N);
  }
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

}  // namespace
}  // namespace pyg_hip

using namespace pyg_hip;

extern "C" {

size_t pyg_hip_scatter_workspace_size(int64_t B, int64_t E, int64_t N) {
  return scatter_sort_ws_bytes(E < 0 ? 0 : E) + scatter_indptr_bytes(B < 1 ? 1 : B, N < 0 ? 0 : N);
}

int pyg_hip_scatter(int op, int dtype, const void* src, const int64_t* index, int64_t index_stride_b,
                    int64_t index_stride_e, int64_t index_stride_k, void* out, int64_t* arg_out,
                    const void* out_init, int64_t B, int64_t E, int64_t K, int64_t N, int index_sorted,
                    void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(B >= 0 && E >= 0 && K >= 0 && N >= 0, "scatter: negative size");
  if (B * E * K == 0) {
    if (op == OP_SUM && (index_sorted & PYG_HIP_SCATTER_FRESH_SUM) && out && B * N * K > 0)
      PYG_HIP_CHECK(hipMemsetAsync(out, 0, dtype_size(dtype) * (size_t)(B * N * K), stream));
    if ((op == OP_MIN || op == OP_MAX) && arg_out && B * N * K > 0) {
      hipLaunchKernelGGL(fill_i64_kernel, dim3((unsigned)((B * N * K + 255) / 256)), dim3(256), 0, stream,
                         arg_out, B * N * K, E);
      PYG_HIP_CHECK(hipGetLastError());
    }
    return PYG_HIP_OK;
  }
  PYG_HIP_REQUIRE(src && index && out, "scatter: NULL tensor");
  Shape s{B, E, K, N, index_stride_b, index_stride_e, index_stride_k};
  PYG_DISPATCH_ALL(dtype, (run_scatter<scalar_t>(op, src, index, out, arg_out, out_init, s, index_sorted, workspace,
                                                 workspace_bytes, stream)));
}

int pyg_hip_fill_reduce_identity(int op, int dtype, void* out, int64_t n, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(op == OP_MIN || op == OP_MAX, "fill_reduce_identity: only min/max need a non-trivial fill");
  PYG_DISPATCH_ALL(dtype, (run_fill_extreme<scalar_t>(op, out, n, stream)));
}

int pyg_hip_gather_coo(int dtype, const void* src, const int64_t* index, void* out, int64_t B, int64_t E,
                       int64_t K, int64_t N, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(B >= 0 && E >= 0 && K >= 0 && N >= 0, "gather_coo: negative size");
  if (B * E * K == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(src && index && out, "gather_coo: NULL tensor");
  PYG_DISPATCH_ALL(dtype, (run_gather<scalar_t>(src, index, out, B, E, K, N, stream)));
}

}  // extern "C"
