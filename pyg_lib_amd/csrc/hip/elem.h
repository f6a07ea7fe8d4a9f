// Element helpers shared by the reduction kernels (reduce.hip, csr.hip): storage types, the
// reference's opmath (float for Half/BFloat16), numeric limits and the dtype dispatch.
#pragma once

#include "common.h"

namespace pyg_hip {
namespace {

typedef __bf16 bf16_raw;
struct bf16_t {
  uint16_t v;
};
struct f16_t {
  uint16_t v;
};

enum { OP_SUM = 0, OP_MUL = 1, OP_MIN = 2, OP_MAX = 3 };

// ---- scalar element helpers ----------------------------------------------------------------------
template <typename T>
struct Math {
  using acc_t = T;
  __device__ static acc_t up(T v) { return v; }
  __device__ static T down(acc_t v) { return v; }
};
template <>
struct Math<bf16_t> {
  using acc_t = float;
  __device__ static float up(bf16_t v) { return __builtin_bit_cast(float, (uint32_t)v.v << 16); }
  __device__ static bf16_t down(float f) {
    bf16_t r;
    r.v = __builtin_bit_cast(uint16_t, (__bf16)f);
    return r;
  }
};
template <>
struct Math<f16_t> {
  using acc_t = float;
  __device__ static float up(f16_t v) { return (float)__builtin_bit_cast(_Float16, v.v); }
  __device__ static f16_t down(float f) {
    f16_t r;
    r.v = __builtin_bit_cast(uint16_t, (_Float16)f);
    return r;
  }
};

template <typename T>
__device__ T type_max();
template <typename T>
__device__ T type_lowest();
#define PYG_LIMITS(T, MAXV, LOWV)                     \
  template <>                                         \
  inline __device__ T type_max<T>() { return MAXV; }         \
  template <>                                         \
  inline __device__ T type_lowest<T>() { return LOWV; }
PYG_LIMITS(float, 3.402823466e+38f, -3.402823466e+38f)
PYG_LIMITS(double, 1.7976931348623157e+308, -1.7976931348623157e+308)
PYG_LIMITS(int8_t, 127, -128)
PYG_LIMITS(uint8_t, 255, 0)
PYG_LIMITS(int16_t, 32767, -32768)
PYG_LIMITS(int32_t, 2147483647, (-2147483647 - 1))
PYG_LIMITS(int64_t, 9223372036854775807ll, (-9223372036854775807ll - 1))
#undef PYG_LIMITS
template <>
inline __device__ bf16_t type_max<bf16_t>() { return bf16_t{0x7f7f}; }
template <>
inline __device__ bf16_t type_lowest<bf16_t>() { return bf16_t{0xff7f}; }
template <>
inline __device__ f16_t type_max<f16_t>() { return f16_t{0x7bff}; }
template <>
inline __device__ f16_t type_lowest<f16_t>() { return f16_t{0xfbff}; }

template <typename T>
__device__ bool bits_equal(T a, T b) {
  return Math<T>::up(a) == Math<T>::up(b);
}


#define PYG_DISPATCH_ALL(dtype, CALL)                                         \
  switch (dtype) {                                                            \
    case PYG_F32: { using scalar_t = float; return CALL; }                    \
    case PYG_F64: { using scalar_t = double; return CALL; }                   \
    case PYG_F16: { using scalar_t = f16_t; return CALL; }                    \
    case PYG_BF16: { using scalar_t = bf16_t; return CALL; }                  \
    case PYG_I8: { using scalar_t = int8_t; return CALL; }                    \
    case PYG_U8: { using scalar_t = uint8_t; return CALL; }                   \
    case PYG_I16: { using scalar_t = int16_t; return CALL; }                  \
    case PYG_I32: { using scalar_t = int32_t; return CALL; }                  \
    case PYG_I64: { using scalar_t = int64_t; return CALL; }                  \
    default: return fail(PYG_HIP_ERR_INVALID, "unknown dtype %d", dtype);     \
  }


}  // namespace
}  // namespace pyg_hip
