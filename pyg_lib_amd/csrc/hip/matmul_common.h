// Element types, the device-side group descriptor and the MFMA / rounding helpers shared by the translation units of
// segment_matmul / grouped_matmul (matmul.hip: shape-specialised kernels + dispatch, matmul_gen.hip: general shapes).
#pragma once

#include "common.h"

namespace pyg_hip {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct bf16_t {
  uint16_t v;
};
struct f16_t {
  uint16_t v;
};

// Device-side group descriptor (48 bytes).
struct DevGroup {
  const char* a;
  const char* w;
  char* c;
  const char* bias;
  int64_t rows;
  int32_t k;
  int32_t m;
  int32_t trans;
  int32_t pad;
};

template <typename T>
struct Elem;
template <>
struct Elem<bf16_t> {
  static constexpr int kSize = 2;
  static constexpr int kPerChunk = 8;   // elements per 16-byte chunk
  static constexpr int kStepsPerChunk = 1;  // MFMA k-steps fed by one chunk
};
template <>
struct Elem<f16_t> {
  static constexpr int kSize = 2;
  static constexpr int kPerChunk = 8;
  static constexpr int kStepsPerChunk = 1;
};
template <>
struct Elem<float> {
  static constexpr int kSize = 4;
  static constexpr int kPerChunk = 4;
  static constexpr int kStepsPerChunk = 4;
};

__device__ __forceinline__ f32x16 mfma_chunk(bf16_t, u32x4 a, u32x4 b, f32x16 acc) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_chunk(f16_t, u32x4 a, u32x4 b, f32x16 acc) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_chunk(float, u32x4 a, u32x4 b, f32x16 acc) {
  f32x4 af = __builtin_bit_cast(f32x4, a);
  f32x4 bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
  for (int e = 0; e < 4; ++e)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc, 0, 0, 0);
  return acc;
}

__device__ __forceinline__ float load_bias(const bf16_t* p) {
  return __builtin_bit_cast(float, (uint32_t)p->v << 16);
}
__device__ __forceinline__ float load_bias(const f16_t* p) {
  return (float)__builtin_bit_cast(_Float16, p->v);
}
__device__ __forceinline__ float load_bias(const float* p) { return *p; }
__device__ __forceinline__ float round_to(bf16_t, float v) { return (float)(__bf16)v; }
__device__ __forceinline__ float round_to(f16_t, float v) { return (float)(_Float16)v; }
__device__ __forceinline__ float round_to(float, float v) { return v; }

// Split-bf16 helpers (fp32 K = M-chunk = 128 kernel, FLAGS bit 2).  split2(a, b): round-to-nearest-even bf16 pair of
// (a, b) packed {a low, b high} (v_cvt_pk_bf16_f32), and the exact fp32 residuals a - bf16(a), b - bf16(b).
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
typedef float f32x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t split2(float& a, float& b) {
  const f32x2_hw v = {a, b};
  const uint32_t p = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw));
  a -= __builtin_bit_cast(float, p << 16);
  b -= __builtin_bit_cast(float, p & 0xffff0000u);
  return p;
}

// ---- alignment classes of the general-shape kernel (matmul_gen.h) ---------------------------------
__host__ __device__ inline int gen_log2_align(uint64_t v) {
  // log2 of the largest power of two <= 16 dividing v (v == 0: 16)
  if ((v & 15) == 0) return 4;
  if ((v & 7) == 0) return 3;
  if ((v & 3) == 0) return 2;
  if ((v & 1) == 0) return 1;
  return 0;
}

// Alignment classes of one group, packed into DevGroup::pad: log2 of the vector bytes of the X, W and out accesses.
__host__ __device__ inline int gen_class(const void* a, const void* w, const void* c, int64_t K, int64_t M, int elt,
                                         int trans) {
  auto mn = [](int x, int y) { return x < y ? x : y; };
  const int lx = mn(gen_log2_align((uint64_t)a), gen_log2_align((uint64_t)(K * elt)));
  const int lw = mn(gen_log2_align((uint64_t)w), gen_log2_align((uint64_t)((trans ? K : M) * elt)));
  const int lc = mn(gen_log2_align((uint64_t)c), gen_log2_align((uint64_t)(M * elt)));
  return lx | (lw << 3) | (lc << 6);
}

}  // namespace

// matmul_gen.hip: the general-shape MFMA kernel (per-group K, M and alignment class).  `dtype` is PYG_F32 / PYG_BF16 /
// PYG_F16; `tile_start` the prefix of 128-row tiles per group, `mean_k` the row-weighted mean contraction length.
int launch_matmul_gen(int dtype, const void* descs, const int32_t* tile_start, int B, int64_t tiles_upper, int64_t mean_k,
                      hipStream_t stream);
// matmul_ring.hip: the item-ring kernels.  `tile_start3` is the prefix of 64-row tiles per group, `tiles3_upper` an upper
// bound of their number; `dtype` PYG_BF16 / PYG_F16.
int launch_ring_k256(int dtype, const void* descs, const int32_t* tile_start3, int B, int64_t tiles3_upper, hipStream_t stream);
int launch_ring_k128(int dtype, const void* descs, const int32_t* tile_start3, int B, int64_t tiles3_upper, hipStream_t stream);
int launch_ring_f32x3(const void* descs, const int32_t* tile_start3, int B, int64_t tiles3_upper, hipStream_t stream);
// matmul_dw_gen.hip: the general-shape weight gradient (per-group K, M, alignment class; bf16 / f16 / f32).  The
// workspace is carved as [B + 1 ptr copy][descriptors][tile prefix][two fp32 partial slabs per workgroup].
size_t dw_gen_workspace_bytes(int64_t B);
int dw_gen_segment(int dtype, const void* input, const int64_t* ptr, int ptr_on_device, const void* grad_out,
                   void* grad_other, int64_t N, int64_t K, int64_t M, int64_t B, void* workspace, hipStream_t stream);
int dw_gen_grouped(int dtype, const pyg_hip_group* host_groups, int64_t G, void* out_pool, void* workspace,
                   hipStream_t stream);

}  // namespace pyg_hip
