// Weight gradient of segment_matmul for gfx950 (MI355X):  dW[b] = X_b^T @ dY_b  for every relation b.
//
// Replaces the reference's B-iteration backward (pyg_lib/csrc/ops/autograd/matmul_kernel.cpp:92-107:
// one at::matmul per relation + at::stack) with one persistent launch.  HBM-bound
// like the forward: X and dY are read exactly once (2 N F s bytes), the B small K x M results are
// accumulated in fp32.
//
//   * Same tiling as the forward: 128-row tiles that never cross a relation, every workgroup walks a
//     contiguous tile range, so it meets ~1.3 relations and flushes its accumulators only then.
//   * Each of the 4 waves owns 32 rows of a tile and a FULL K x MC fp32 accumulator block in
//     registers ((K/32) x (MC/32) MFMA blocks x 16 = 256 accumulators; 1 wave / SIMD): the waves never
//     exchange data and there is no workgroup barrier in the loop.
//   * The contraction runs over ROWS, so both MFMA operands are "8 rows of one column" -- transposed
//     with respect to the row-major tensors.  Rows are loaded with fully coalesced 1 KiB wave
//     accesses, parked in a wave-private row-major LDS image, and read back through gfx950's LDS
//     transpose read (ds_read_b64_tr_b16: every 16-lane group turns a 4 x 16 block into per-lane
//     4-row columns): 2 reads per MFMA operand instead of 8 two-byte reads.  The image pitch is
//     16 (mod 64) dwords, which keeps the two 16-lane groups of a 32-lane service group and the 4 rows
//     of a block on disjoint banks.
//   * Accumulators are flushed with native global_atomic_add_f32 into an fp32 [B, K, M] scratch
//     (43 M atomics for C2, L2-resident) and rounded once to the storage type by a tiny epilogue.
#include "matmul_common.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>

namespace pyg_hip {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef short v8i16 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kTile = 128;  // rows per workgroup tile (4 waves x 32)

struct bf16_tag {};
struct f16_tag {};

__device__ __forceinline__ f32x16 mfma16(bf16_tag, v8i16 a, v8i16 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(f16_tag, v8i16 a, v8i16 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// pitch (bytes) of a row-major [32][cols] 16-bit LDS image: >= the row, == 16 (mod 64) dwords
constexpr int pitch_bytes(int cols) {
  const int dw = cols / 2;
  const int extra = dw > 16 ? (dw - 16 + 63) / 64 : 0;
  return (16 + 64 * extra) * 4;
}

// global_* loads (the descriptors hold generic pointers: through them the compiler emits flat_* accesses, which also count
// on lgkmcnt -- every LDS wait then waits for the rows in flight as well)
typedef const __attribute__((address_space(1))) u32x4 GU32x4;

// One relation / group: `rows` rows of X [rows, K] and dY [rows, M] (row-major, M = row pitch of dY).
struct DwGroup {
  const uint16_t* x;
  const uint16_t* dy;
  int64_t rows;
};

// segment form: descriptors + tile prefix from ptr; tile_start[b] = sum_{b' < b} ceil(rows_b' / 128)
__global__ void dw_plan_kernel(const int64_t* __restrict__ ptr, int64_t B, const uint16_t* X, const uint16_t* dY,
                               int64_t K, int64_t M, DwGroup* __restrict__ groups,
                               int32_t* __restrict__ tile_start) {
  __shared__ int64_t part[256];
  const int tid = threadIdx.x;
  const int64_t per = (B + 255) / 256;
  const int64_t beg = min((int64_t)tid * per, B), end = min(beg + per, B);
  int64_t t = 0;
  for (int64_t b = beg; b < end; ++b) {
    const int64_t r = ptr[b + 1] - ptr[b];
    t += r > 0 ? (r + kTile - 1) / kTile : 0;
  }
  part[tid] = t;
  __syncthreads();
  if (tid == 0) {
    int64_t acc = 0;
    for (int i = 0; i < 256; ++i) {
      const int64_t v = part[i];
      part[i] = acc;
      acc += v;
    }
    tile_start[B] = (int32_t)acc;
  }
  __syncthreads();
  t = part[tid];
  for (int64_t b = beg; b < end; ++b) {
    tile_start[b] = (int32_t)t;
    const int64_t p0 = ptr[b];
    const int64_t r = ptr[b + 1] - p0;
    DwGroup d;
    d.x = X + p0 * K;
    d.dy = dY + p0 * M;
    d.rows = r > 0 ? r : 0;
    groups[b] = d;
    t += r > 0 ? (r + kTile - 1) / kTile : 0;
  }
}

template <typename Tag, int K, int MC>
__global__ __launch_bounds__(256, 1) void seg_dw_kernel(const DwGroup* __restrict__ groups,
                                                         const int32_t* __restrict__ tile_start, int B, int M,
                                                         float* __restrict__ acc_out) {
  constexpr int IB = K / 32, JB = MC / 32;
  constexpr int PX = pitch_bytes(K), PY = pitch_bytes(MC);
  constexpr int CX = K / 8, CY = MC / 8;          // 16-byte chunks per row
  constexpr int NX = 32 * CX / 64, NY = 32 * CY / 64;  // chunk loads per lane and tile
  static_assert(IB * JB <= 16, "accumulators must fit the register file");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* xs = smem + wave * 32 * (PX + PY);
  char* ys = xs + 32 * PX;
  // XCD-aware decode of the 1-D grid: the M / MC column-chunk workgroups of one tile range get ids 8 apart
  // (same XCD, same L2), so the X rows they all read come from HBM once
  const int ncol = M / MC;
  const int bx = ncol > 1 ? ((int)blockIdx.x / (8 * ncol)) * 8 + ((int)blockIdx.x & 7) : (int)blockIdx.x;
  const int by = ncol > 1 ? ((int)blockIdx.x / 8) % ncol : 0;
  const int col0 = by * MC;

  const int total = tile_start[B];
  const int G = (int)gridDim.x / ncol;
  const int t_beg = (int)((int64_t)bx * total / G);
  const int t_end = (int)((int64_t)(bx + 1) * total / G);
  if (t_beg >= t_end) return;
  int g = 0;
  {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tile_start[mid] <= t_beg) lo = mid; else hi = mid;
    }
    g = lo;
  }

  f32x16 acc[IB][JB];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  int acc_g = -1;  // relation the accumulators belong to

  auto flush = [&]() {
    if (acc_g < 0) return;
    float* base = acc_out + ((int64_t)acc_g * K) * M + col0;
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int col = j * 32 + (lane & 31);
          __hip_atomic_fetch_add(base + (int64_t)row * M + col, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          acc[i][j][r] = 0.0f;
        }
  };

  // software pipeline: rows of tile t+1 travel to registers while tile t is multiplied
  u32x4 xr[NX], yr[NY];
  int n_g = g;
  DwGroup gd = groups[g];
  auto prefetch = [&](int t) {
    while (t >= tile_start[n_g + 1]) {
      ++n_g;
      gd = groups[n_g];
    }
    const int64_t row0 = (int64_t)(t - tile_start[n_g]) * kTile + wave * 32;
#pragma unroll
    for (int it = 0; it < NX; ++it) {
      const int p = it * 64 + lane;
      const int64_t row = row0 + p / CX;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row < gd.rows) v = *(GU32x4*)(gd.x + row * K + (p % CX) * 8);
      xr[it] = v;
    }
#pragma unroll
    for (int it = 0; it < NY; ++it) {
      const int p = it * 64 + lane;
      const int64_t row = row0 + p / CY;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row < gd.rows) v = *(GU32x4*)(gd.dy + row * M + col0 + (p % CY) * 8);
      yr[it] = v;
    }
  };

  // lane constants of the transpose reads: lane q of a 16-lane group supplies row (q >> 2), columns
  // (q & 3) * 4 of a 4 x 16 block and receives column q; group = (column half, row half kb)
  const int q = lane & 15, half = (lane >> 4) & 1, kb = lane >> 5;
  const int a_off = (kb * 8 + (q >> 2)) * PX + (half * 16 + (q & 3) * 4) * 2;
  const int b_off = (kb * 8 + (q >> 2)) * PY + (half * 16 + (q & 3) * 4) * 2;
  typedef __attribute__((address_space(3))) v4i16* lds_v4;

  prefetch(t_beg);
  for (int t = t_beg; t < t_end; ++t) {
    const int cur_g = n_g;  // relation of the tile held in xr / yr
    if (cur_g != acc_g) {
      flush();
      acc_g = cur_g;
    }
    // park the tile in the wave-private LDS image (row-major)
#pragma unroll
    for (int it = 0; it < NX; ++it) {
      const int p = it * 64 + lane;
      *reinterpret_cast<u32x4*>(xs + (p / CX) * PX + (p % CX) * 16) = xr[it];
    }
#pragma unroll
    for (int it = 0; it < NY; ++it) {
      const int p = it * 64 + lane;
      *reinterpret_cast<u32x4*>(ys + (p / CY) * PY + (p % CY) * 16) = yr[it];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (t + 1 < t_end) prefetch(t + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      v8i16 af[IB], bf[JB];
#pragma unroll
      for (int i = 0; i < IB; ++i) {
        const char* p = xs + a_off + ks * 16 * PX + i * 64;
        const v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p));
        const v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p + 4 * PX));
        af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        const char* p = ys + b_off + ks * 16 * PY + j * 64;
        const v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p));
        const v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p + 4 * PY));
        bf[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) acc[i][j] = mfma16(Tag{}, af[i], bf[j], acc[i][j]);
    }
    __builtin_amdgcn_wave_barrier();
  }
  flush();
}

// K = 256, 256 output columns per workgroup: the four waves SHARE every 32-row slab of X and dY and split the
// COLUMNS of dW instead of the rows (wave w owns dW[:, 64 w .. 64 w + 63]: 8 x 2 accumulator blocks, the same
// 256 registers as above).  With row-split waves a 256-wide dW needs four column-chunk workgroups that each
// stream the whole X again (C4: 4.6 ms); here X and dY pass through one CU once.  The slab lives in a
// workgroup-shared, double-buffered row-major LDS image (same pitch rule, same transpose reads); every wave
// loads a quarter of the next slab while the current one is multiplied, one barrier per slab:
//   iteration s:  registers (slab s+1) -> image[(s+1) & 1];  issue loads of slab s+2;  MFMAs on image[s & 1];  barrier.
template <typename Tag>
__global__ __launch_bounds__(256, 1) void seg_dw_wide256_kernel(const DwGroup* __restrict__ groups,
                                                                const int32_t* __restrict__ tile_start, int B, int M,
                                                                float* __restrict__ acc_out) {
  constexpr int K = 256, MC = 256, IB = K / 32, JB = 2;
  constexpr int PX = pitch_bytes(K), PY = pitch_bytes(MC);
  constexpr int CX = K / 8, CY = MC / 8;                 // 16-byte chunks per row
  constexpr int NX = 32 * CX / 256, NY = 32 * CY / 256;  // chunk loads per THREAD and slab
  constexpr int IMG = 32 * (PX + PY);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncol = M / MC;
  const int bx = ncol > 1 ? ((int)blockIdx.x / (8 * ncol)) * 8 + ((int)blockIdx.x & 7) : (int)blockIdx.x;
  const int by = ncol > 1 ? ((int)blockIdx.x / 8) % ncol : 0;
  const int col0 = by * MC;

  const int total = tile_start[B];
  const int G = (int)gridDim.x / ncol;
  const int t_beg = (int)((int64_t)bx * total / G);
  const int t_end = (int)((int64_t)(bx + 1) * total / G);
  if (t_beg >= t_end) return;
  int g = 0;
  {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tile_start[mid] <= t_beg) lo = mid; else hi = mid;
    }
    g = lo;
  }

  f32x16 acc[IB][JB];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  int acc_g = -1;
  auto flush = [&]() {
    if (acc_g < 0) return;
    float* base = acc_out + ((int64_t)acc_g * K) * M + col0 + 64 * wave;
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int col = j * 32 + (lane & 31);
          __hip_atomic_fetch_add(base + (int64_t)row * M + col, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          acc[i][j][r] = 0.0f;
        }
  };

  // slab index space: 4 slabs of 32 rows per 128-row tile
  const int s_beg = t_beg * 4, s_end = t_end * 4;
  u32x4 xr[NX], yr[NY];
  int n_g = g;
  DwGroup gd = groups[g];
  auto prefetch = [&](int sl) {  // this thread's share of slab sl (group n_g after the call)
    const int t = sl >> 2;
    while (t >= tile_start[n_g + 1]) {
      ++n_g;
      gd = groups[n_g];
    }
    const int64_t row0 = (int64_t)(t - tile_start[n_g]) * kTile + (sl & 3) * 32;
#pragma unroll
    for (int it = 0; it < NX; ++it) {
      const int p = it * 256 + tid;
      const int64_t row = row0 + p / CX;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row < gd.rows) v = *(GU32x4*)(gd.x + row * K + (p % CX) * 8);
      xr[it] = v;
    }
#pragma unroll
    for (int it = 0; it < NY; ++it) {
      const int p = it * 256 + tid;
      const int64_t row = row0 + p / CY;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row < gd.rows) v = *(GU32x4*)(gd.dy + row * M + col0 + (p % CY) * 8);
      yr[it] = v;
    }
  };
  auto park = [&](int buf) {
    char* xs = smem + buf * IMG;
    char* ys = xs + 32 * PX;
#pragma unroll
    for (int it = 0; it < NX; ++it) {
      const int p = it * 256 + tid;
      *reinterpret_cast<u32x4*>(xs + (p / CX) * PX + (p % CX) * 16) = xr[it];
    }
#pragma unroll
    for (int it = 0; it < NY; ++it) {
      const int p = it * 256 + tid;
      *reinterpret_cast<u32x4*>(ys + (p / CY) * PY + (p % CY) * 16) = yr[it];
    }
  };

  const int q = lane & 15, half = (lane >> 4) & 1, kb = lane >> 5;
  const int a_off = (kb * 8 + (q >> 2)) * PX + (half * 16 + (q & 3) * 4) * 2;
  const int b_off = (kb * 8 + (q >> 2)) * PY + (half * 16 + (q & 3) * 4) * 2 + wave * 128;  // this wave's 64 columns
  typedef __attribute__((address_space(3))) v4i16* lds_v4;

  // group of the slab held in the registers / of each image
  prefetch(s_beg);
  int g_img[2];
  g_img[0] = n_g;
  park(0);
  g_img[1] = n_g;
  if (s_beg + 1 < s_end) prefetch(s_beg + 1);
  int g_regs = n_g;
  __syncthreads();

  for (int sl = s_beg; sl < s_end; ++sl) {
    const int cur = (sl - s_beg) & 1;
    if (sl + 1 < s_end) {
      park(cur ^ 1);  // slab sl + 1 (in registers since the previous iteration)
      g_img[cur ^ 1] = g_regs;
      if (sl + 2 < s_end) {
        prefetch(sl + 2);
        g_regs = n_g;
      }
    }
    const int cur_g = g_img[cur];
    if (cur_g != acc_g) {
      flush();
      acc_g = cur_g;
    }
    const char* xs = smem + cur * IMG;
    const char* ys = xs + 32 * PX;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      v8i16 af[IB], bf[JB];
#pragma unroll
      for (int i = 0; i < IB; ++i) {
        const char* p = xs + a_off + ks * 16 * PX + i * 64;
        const v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p));
        const v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p + 4 * PX));
        af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        const char* p = ys + b_off + ks * 16 * PY + j * 64;
        const v4i16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p));
        const v4i16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p + 4 * PY));
        bf[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) acc[i][j] = mfma16(Tag{}, af[i], bf[j], acc[i][j]);
    }
    __syncthreads();
  }
  flush();
}

// fp32: v_mfma_f32_32x32x2_f32 takes ONE value per lane (lane (i, kk) = A[i][kk]), so the "column of
// rows" operand is simply a row-major load -- lane (i, kk) reads X[row 2s + kk][col i]: two coalesced
// 128-byte segments per wave instruction, no LDS and no transposition.  Exact fp32 FMA chains (gfx950 has
// no TF32); bound by the fp32 MFMA rate (157 TFLOP/s: C2 >= 4.4 ms), not by HBM.
template <int K, int MC>
__global__ __launch_bounds__(256, 1) void seg_dw_f32_kernel(const DwGroup* __restrict__ groups,
                                                             const int32_t* __restrict__ tile_start, int B, int M,
                                                             float* __restrict__ acc_out) {
  constexpr int IB = K / 32, JB = MC / 32;
  static_assert(IB * JB <= 16, "accumulators must fit the register file");
  // The MFMA wants ONE value per lane and block, but nothing ties block i to the columns 32 i .. 32 i + 31:
  // lane li loads VA (up to 4) CONSECUTIVE floats of its row in one instruction and feeds them to VA
  // different blocks (block VA h + e = columns 32 VA h + VA li + e) -- 16-byte instead of 4-byte accesses,
  // whole 512-byte rows per wave instruction; the permutation is undone when the accumulators are flushed.
  constexpr int VA = IB >= 4 ? 4 : IB, VB = JB >= 4 ? 4 : JB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kk = lane >> 5;
  const int ncol = M / MC;
  const int bx = ncol > 1 ? ((int)blockIdx.x / (8 * ncol)) * 8 + ((int)blockIdx.x & 7) : (int)blockIdx.x;
  const int by = ncol > 1 ? ((int)blockIdx.x / 8) % ncol : 0;
  const int col0 = by * MC;
  const int total = tile_start[B];
  const int G = (int)gridDim.x / ncol;
  const int t_beg = (int)((int64_t)bx * total / G);
  const int t_end = (int)((int64_t)(bx + 1) * total / G);
  if (t_beg >= t_end) return;
  int g = 0;
  {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tile_start[mid] <= t_beg) lo = mid; else hi = mid;
    }
    g = lo;
  }
  f32x16 acc[IB][JB];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  int acc_g = -1;
  auto flush = [&]() {
    if (acc_g < 0) return;
    float* base = acc_out + ((int64_t)acc_g * K) * M + col0;
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // block i = VA h + e holds the X columns 32 VA h + VA * (MFMA row) + e (see the loads below)
          const int mrow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int row = 32 * VA * (i / VA) + VA * mrow + (i % VA);
          const int col = 32 * VB * (j / VB) + VB * (lane & 31) + (j % VB);
          __hip_atomic_fetch_add(base + (int64_t)row * M + col, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          acc[i][j][r] = 0.0f;
        }
  };
  // Software pipeline over the flattened (tile, row pair) sequence: the operands of step it + D are requested
  // while step it is multiplied (one wave per SIMD and 16 MFMAs = 1 k cycles per step: without the ring an HBM
  // round trip of ~5 k cycles is exposed every few steps).  D slots of (a, b, group) with static indices.
  constexpr int D = 8;
  const int n_it = (t_end - t_beg) * 16;
  float pa[D][IB], pb[D][JB];
  int pg[D];
  int lg = g;  // group walk of the load side
  DwGroup ld = groups[lg];
  auto request = [&](int it, float (&a)[IB], float (&b)[JB], int& grp) {
    const int t = t_beg + (it >> 4);
    while (t >= tile_start[lg + 1]) {
      ++lg;
      ld = groups[lg];
    }
    grp = lg;
    const float* xp = reinterpret_cast<const float*>(ld.x);
    const float* yp = reinterpret_cast<const float*>(ld.dy);
    const int64_t row = (int64_t)(t - tile_start[lg]) * kTile + wave * 32 + 2 * (it & 15) + kk;
    const bool ok = row < ld.rows;
#pragma unroll
    for (int h = 0; h < IB / VA; ++h) {
      typedef float vecA __attribute__((ext_vector_type(VA)));
      vecA v;
#pragma unroll
      for (int e = 0; e < VA; ++e) v[e] = 0.0f;
      if (ok) v = *(const __attribute__((address_space(1))) vecA*)(xp + row * K + 32 * VA * h + VA * li);
#pragma unroll
      for (int e = 0; e < VA; ++e) a[VA * h + e] = v[e];
    }
#pragma unroll
    for (int h = 0; h < JB / VB; ++h) {
      typedef float vecB __attribute__((ext_vector_type(VB)));
      vecB v;
#pragma unroll
      for (int e = 0; e < VB; ++e) v[e] = 0.0f;
      if (ok) v = *(const __attribute__((address_space(1))) vecB*)(yp + row * M + col0 + 32 * VB * h + VB * li);
#pragma unroll
      for (int e = 0; e < VB; ++e) b[VB * h + e] = v[e];
    }
  };
#pragma unroll
  for (int k = 0; k < D; ++k) request(k, pa[k], pb[k], pg[k]);  // n_it is a multiple of 16
  // D divides the 16 steps of a tile and tiles never cross a relation: a chunk of D steps has ONE relation
  static_assert(16 % D == 0, "a chunk of D steps must stay inside one tile");
  for (int it0 = 0; it0 < n_it; it0 += D) {
    if (pg[0] != acc_g) {
      flush();
      acc_g = pg[0];
    }
#pragma unroll
    for (int k = 0; k < D; ++k) {
#pragma unroll
      for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k][i], pb[k][j], acc[i][j], 0, 0, 0);
      if (it0 + k + D < n_it) request(it0 + k + D, pa[k], pb[k], pg[k]);
    }
  }
  flush();
}

// dW[b, k, m] = round(acc[b, k, m])
template <typename Tag>
__global__ void dw_round_kernel(const float* __restrict__ acc, uint16_t* __restrict__ out, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  if constexpr (sizeof(Tag) && __is_same(Tag, bf16_tag)) out[i] = __builtin_bit_cast(uint16_t, (__bf16)acc[i]);
  else out[i] = __builtin_bit_cast(uint16_t, (_Float16)acc[i]);
}

inline size_t dw_groups_bytes(int64_t B) { return align_up(sizeof(DwGroup) * (size_t)(B > 0 ? B : 1), 256); }
inline size_t dw_tiles_bytes(int64_t B) { return align_up(sizeof(int32_t) * (size_t)(B + 1), 256); }
inline size_t dw_ws_bytes(int64_t B, int64_t K, int64_t M) {
  return align_up(sizeof(int64_t) * (size_t)(B + 1), 256) + dw_groups_bytes(B) + dw_tiles_bytes(B) +
         align_up(sizeof(float) * (size_t)B * (size_t)K * (size_t)M, 256);
}

template <typename Tag, int K, int MC>
int launch_dw(const DwGroup* groups, const int32_t* tile_start, int64_t B, int64_t M, int64_t tiles_upper, float* acc,
              hipStream_t stream) {
  constexpr int lds = 4 * 32 * (pitch_bytes(K) + pitch_bytes(MC));
  const void* kern = reinterpret_cast<const void*>(&seg_dw_kernel<Tag, K, MC>);
  if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
  const int64_t cus = device_info().num_cus;
  const int64_t ncol = M / MC;
  int64_t gx = std::max<int64_t>(1, std::min<int64_t>(tiles_upper, cus / ncol));
  if (ncol > 1) gx = (gx + 7) / 8 * 8;  // the kernel's XCD-aware decode works on groups of 8 ids
  hipLaunchKernelGGL((seg_dw_kernel<Tag, K, MC>), dim3((unsigned)(gx * ncol)), dim3(256), lds, stream, groups,
                     tile_start, (int)B, (int)M, acc);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

template <typename Tag>
int launch_dw_wide256(const DwGroup* groups, const int32_t* tile_start, int64_t B, int64_t M, int64_t tiles_upper,
                      float* acc, hipStream_t stream) {
  constexpr int lds = 2 * 32 * (pitch_bytes(256) + pitch_bytes(256));
  const void* kern = reinterpret_cast<const void*>(&seg_dw_wide256_kernel<Tag>);
  if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
  const int64_t cus = device_info().num_cus;
  const int64_t ncol = M / 256;
  int64_t gx = std::max<int64_t>(1, std::min<int64_t>(tiles_upper, cus / ncol));
  if (ncol > 1) gx = (gx + 7) / 8 * 8;
  hipLaunchKernelGGL((seg_dw_wide256_kernel<Tag>), dim3((unsigned)(gx * ncol)), dim3(256), lds, stream, groups, tile_start,
                     (int)B, (int)M, acc);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

template <typename Tag>
int run_dw(const DwGroup* groups, const int32_t* tile_start, int64_t B, int64_t K, int64_t M, int64_t tiles_upper,
           float* acc, hipStream_t stream) {
#ifdef PYG_HIP_MM_EXPERIMENTS
  static const bool nowide = getenv("PYG_HIP_MM_NOWIDE") != nullptr;
#else
  constexpr bool nowide = false;
#endif
  if (K == 256 && M % 256 == 0 && !nowide) return launch_dw_wide256<Tag>(groups, tile_start, B, M, tiles_upper, acc, stream);
  if (K == 128 && M % 128 == 0) return launch_dw<Tag, 128, 128>(groups, tile_start, B, M, tiles_upper, acc, stream);
  if (K == 128 && M % 64 == 0) return launch_dw<Tag, 128, 64>(groups, tile_start, B, M, tiles_upper, acc, stream);
  if (K == 64 && M % 128 == 0) return launch_dw<Tag, 64, 128>(groups, tile_start, B, M, tiles_upper, acc, stream);
  if (K == 64 && M % 64 == 0) return launch_dw<Tag, 64, 64>(groups, tile_start, B, M, tiles_upper, acc, stream);
  if (K == 256 && M % 64 == 0) return launch_dw<Tag, 256, 64>(groups, tile_start, B, M, tiles_upper, acc, stream);
  return fail(PYG_HIP_ERR_UNSUPPORTED, "segment_matmul_dw: K=%lld, M=%lld has no MFMA kernel (K in {64,128,256}, M %% 64 == 0)",
              (long long)K, (long long)M);
}

template <int K, int MC>
int launch_dw_f32(const DwGroup* groups, const int32_t* tile_start, int64_t B, int64_t M, int64_t tiles_upper, float* acc,
                  hipStream_t stream) {
  const int64_t cus = device_info().num_cus;
  const int64_t ncol = M / MC;
  int64_t gx = std::max<int64_t>(1, std::min<int64_t>(tiles_upper, cus / ncol));
  if (ncol > 1) gx = (gx + 7) / 8 * 8;
  hipLaunchKernelGGL((seg_dw_f32_kernel<K, MC>), dim3((unsigned)(gx * ncol)), dim3(256), 0, stream, groups, tile_start,
                     (int)B, (int)M, acc);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

int run_dw_f32(const DwGroup* groups, const int32_t* tile_start, int64_t B, int64_t K, int64_t M, int64_t tiles_upper,
               float* acc, hipStream_t stream) {
  if (K == 128 && M % 128 == 0) return launch_dw_f32<128, 128>(groups, tile_start, B, M, tiles_upper, acc, stream);
  if (K == 128 && M % 64 == 0) return launch_dw_f32<128, 64>(groups, tile_start, B, M, tiles_upper, acc, stream);
  if (K == 64 && M % 128 == 0) return launch_dw_f32<64, 128>(groups, tile_start, B, M, tiles_upper, acc, stream);
  if (K == 64 && M % 64 == 0) return launch_dw_f32<64, 64>(groups, tile_start, B, M, tiles_upper, acc, stream);
  if (K == 256 && M % 64 == 0) return launch_dw_f32<256, 64>(groups, tile_start, B, M, tiles_upper, acc, stream);
  return fail(PYG_HIP_ERR_UNSUPPORTED, "segment_matmul_dw: K=%lld, M=%lld has no MFMA kernel (K in {64,128,256}, M %% 64 == 0)",
              (long long)K, (long long)M);
}

// shapes of the specialised kernels above; everything else (and operands that are not 16-byte aligned) runs
// matmul_dw_gen.hip
inline bool dw_fast_shape(int64_t K, int64_t M) { return (K == 64 || K == 128 || K == 256) && M > 0 && M % 64 == 0; }

int round_out(int dtype, const float* acc, void* out, int64_t n, hipStream_t stream) {
  if (dtype == PYG_F32) {  // the accumulators ARE the result
    PYG_HIP_CHECK(hipMemcpyAsync(out, acc, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, stream));
    return PYG_HIP_OK;
  }
  if (dtype == PYG_BF16)
    hipLaunchKernelGGL(dw_round_kernel<bf16_tag>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, acc,
                       static_cast<uint16_t*>(out), n);
  else
    hipLaunchKernelGGL(dw_round_kernel<f16_tag>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, acc,
                       static_cast<uint16_t*>(out), n);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

}  // namespace
}  // namespace pyg_hip

using namespace pyg_hip;

namespace {
// launches of the shape-specialised / the general-shape weight-gradient kernels since the library was loaded (tests
// assert that a backward pass ran a device kernel and not the caller's fallback formula; the backward runs on an
// autograd thread, so a thread-local "last variant" would not be visible to the test)
std::atomic<int64_t> g_dw_fast{0}, g_dw_gen{0};
}  // namespace

extern "C" {

void pyg_hip_matmul_dw_counters(int64_t* specialised, int64_t* general) {
  if (specialised) *specialised = g_dw_fast.load();
  if (general) *general = g_dw_gen.load();
}

size_t pyg_hip_segment_matmul_dw_workspace_size(int64_t B, int64_t K, int64_t M) {
  B = B < 0 ? 0 : B, K = K < 0 ? 0 : K, M = M < 0 ? 0 : M;
  return std::max(dw_ws_bytes(B, K, M), dw_gen_workspace_bytes(B, B * K * M));
}

size_t pyg_hip_grouped_matmul_dw_workspace_size(const pyg_hip_group* groups, int64_t G) {
  if (G <= 0 || groups == nullptr) return dw_gen_workspace_bytes(0, 0);
  int64_t elems = 0;
  bool uniform = true;
  for (int64_t i = 0; i < G; ++i) {
    elems += (int64_t)std::max(groups[i].k, 0) * std::max(groups[i].m, 0);
    uniform = uniform && groups[i].k == groups[0].k && groups[i].m == groups[0].m;
  }
  const size_t gen = dw_gen_workspace_bytes(G, elems);
  return uniform ? std::max(gen, dw_ws_bytes(G, std::max(groups[0].k, 0), std::max(groups[0].m, 0))) : gen;
}

int pyg_hip_segment_matmul_dw(int dtype, const void* input, const int64_t* ptr, int ptr_on_device, const void* grad_out,
                              void* grad_other, int64_t N, int64_t K, int64_t M, int64_t B, void* workspace,
                              size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(N >= 0 && K >= 0 && M >= 0 && B >= 0, "segment_matmul_dw: negative size");
  if (B * K * M == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(ptr && grad_other && (N == 0 || (input && grad_out)), "segment_matmul_dw: NULL tensor");
  if (dtype != PYG_BF16 && dtype != PYG_F16 && dtype != PYG_F32)
    return fail(PYG_HIP_ERR_UNSUPPORTED, "segment_matmul_dw: float32 / bfloat16 / float16 only (dtype %d)", dtype);
  const size_t need = pyg_hip_segment_matmul_dw_workspace_size(B, K, M);
  if (workspace == nullptr || workspace_bytes < need)
    return fail(PYG_HIP_ERR_WORKSPACE, "segment_matmul_dw: workspace of %zu bytes needed, got %zu", need, workspace_bytes);
  if (!dw_fast_shape(K, M) || ((uintptr_t)input | (uintptr_t)grad_out) % 16 != 0) {
    const int rc = dw_gen_segment(dtype, input, ptr, ptr_on_device, grad_out, grad_other, N, K, M, B, workspace, stream);
    if (rc == PYG_HIP_OK) ++g_dw_gen;
    return rc;
  }
  ++g_dw_fast;
  char* w = static_cast<char*>(workspace);
  int64_t* ptr_dev = reinterpret_cast<int64_t*>(w);
  w += align_up(sizeof(int64_t) * (size_t)(B + 1), 256);
  DwGroup* groups = reinterpret_cast<DwGroup*>(w);
  w += dw_groups_bytes(B);
  int32_t* tile_start = reinterpret_cast<int32_t*>(w);
  w += dw_tiles_bytes(B);
  float* acc = reinterpret_cast<float*>(w);
  const int64_t* dptr = ptr;
  if (!ptr_on_device) {
    void* staged = nullptr;
    int rc = pinned_stage().acquire(sizeof(int64_t) * (size_t)(B + 1), &staged);
    if (rc != PYG_HIP_OK) return rc;
    ::memcpy(staged, ptr, sizeof(int64_t) * (size_t)(B + 1));
    PYG_HIP_CHECK(hipMemcpyAsync(ptr_dev, staged, sizeof(int64_t) * (size_t)(B + 1), hipMemcpyHostToDevice, stream));
    rc = pinned_stage().commit(stream);
    if (rc != PYG_HIP_OK) return rc;
    dptr = ptr_dev;
  }
  PYG_HIP_CHECK(hipMemsetAsync(acc, 0, sizeof(float) * (size_t)B * (size_t)K * (size_t)M, stream));
  // descriptors address the tensors in 2-byte units: an fp32 row is 2 K of them
  const int64_t u = dtype == PYG_F32 ? 2 : 1;
  hipLaunchKernelGGL(dw_plan_kernel, dim3(1), dim3(256), 0, stream, dptr, B, static_cast<const uint16_t*>(input),
                     static_cast<const uint16_t*>(grad_out), K * u, M * u, groups, tile_start);
  PYG_HIP_CHECK(hipGetLastError());
  const int64_t tiles_upper = (N + kTile - 1) / kTile + B;
  int rc = dtype == PYG_F32    ? run_dw_f32(groups, tile_start, B, K, M, tiles_upper, acc, stream)
           : dtype == PYG_BF16 ? run_dw<bf16_tag>(groups, tile_start, B, K, M, tiles_upper, acc, stream)
                               : run_dw<f16_tag>(groups, tile_start, B, K, M, tiles_upper, acc, stream);
  if (rc != PYG_HIP_OK) return rc;
  return round_out(dtype, acc, grad_other, B * K * M, stream);
}

int pyg_hip_grouped_matmul_dw(int dtype, const pyg_hip_group* host_groups, int64_t G, void* out_pool, void* workspace,
                              size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(G >= 0 && G < (1LL << 31), "grouped_matmul_dw: bad group count");
  if (G == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(host_groups && out_pool, "grouped_matmul_dw: NULL argument");
  if (dtype != PYG_BF16 && dtype != PYG_F16 && dtype != PYG_F32)
    return fail(PYG_HIP_ERR_UNSUPPORTED, "grouped_matmul_dw: float32 / bfloat16 / float16 only (dtype %d)", dtype);
  const int64_t K = host_groups[0].k, M = host_groups[0].m;
  int64_t tiles = 0;
  bool fast = dw_fast_shape(K, M);
  for (int64_t i = 0; i < G; ++i) {
    const pyg_hip_group& g = host_groups[i];
    PYG_HIP_REQUIRE(g.rows >= 0 && g.k >= 0 && g.m >= 0, "grouped_matmul_dw: negative size in group %lld", (long long)i);
    PYG_HIP_REQUIRE(g.rows == 0 || g.k == 0 || g.m == 0 || (g.input && g.other), "grouped_matmul_dw: NULL tensor in group %lld",
                    (long long)i);
    if (g.k != K || g.m != M || ((uintptr_t)g.input | (uintptr_t)g.other) % 16 != 0) fast = false;
    tiles += (g.rows + kTile - 1) / kTile;
  }
  const size_t need = pyg_hip_grouped_matmul_dw_workspace_size(host_groups, G);
  if (workspace == nullptr || workspace_bytes < need)
    return fail(PYG_HIP_ERR_WORKSPACE, "grouped_matmul_dw: workspace of %zu bytes needed, got %zu", need, workspace_bytes);
  // per-group shapes, shapes without a specialised kernel, element-aligned views: the general-shape kernel
  if (!fast) {
    const int rc = dw_gen_grouped(dtype, host_groups, G, out_pool, workspace, stream);
    if (rc == PYG_HIP_OK) ++g_dw_gen;
    return rc;
  }
  ++g_dw_fast;
  char* w = static_cast<char*>(workspace) + align_up(sizeof(int64_t) * (size_t)(G + 1), 256);
  DwGroup* groups = reinterpret_cast<DwGroup*>(w);
  w += dw_groups_bytes(G);
  int32_t* tile_start = reinterpret_cast<int32_t*>(w);
  w += dw_tiles_bytes(G);
  float* acc = reinterpret_cast<float*>(w);
  // host-side plan (G is small): descriptors + tile prefix in one pinned H2D copy
  void* staged = nullptr;
  int rc = pinned_stage().acquire(dw_groups_bytes(G) + dw_tiles_bytes(G), &staged);
  if (rc != PYG_HIP_OK) return rc;
  DwGroup* hg = static_cast<DwGroup*>(staged);
  int32_t* ht = reinterpret_cast<int32_t*>(static_cast<char*>(staged) + dw_groups_bytes(G));
  int64_t t = 0;
  for (int64_t i = 0; i < G; ++i) {
    hg[i].x = static_cast<const uint16_t*>(host_groups[i].input);
    hg[i].dy = static_cast<const uint16_t*>(host_groups[i].other);
    hg[i].rows = host_groups[i].rows;
    ht[i] = (int32_t)t;
    t += (host_groups[i].rows + kTile - 1) / kTile;
  }
  ht[G] = (int32_t)t;
  PYG_HIP_CHECK(hipMemcpyAsync(groups, staged, dw_groups_bytes(G) + dw_tiles_bytes(G), hipMemcpyHostToDevice, stream));
  rc = pinned_stage().commit(stream);
  if (rc != PYG_HIP_OK) return rc;
  PYG_HIP_CHECK(hipMemsetAsync(acc, 0, sizeof(float) * (size_t)G * (size_t)K * (size_t)M, stream));
  rc = dtype == PYG_F32    ? run_dw_f32(groups, tile_start, G, K, M, tiles + 1, acc, stream)
       : dtype == PYG_BF16 ? run_dw<bf16_tag>(groups, tile_start, G, K, M, tiles + 1, acc, stream)
                           : run_dw<f16_tag>(groups, tile_start, G, K, M, tiles + 1, acc, stream);
  if (rc != PYG_HIP_OK) return rc;
  return round_out(dtype, acc, out_pool, G * K * M, stream);
}

}  // extern "C"
