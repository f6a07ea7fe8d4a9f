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
//   * No atomics (matmul_dw_out.h): when a relation ends the four waves add their accumulator sets through LDS in a
//     fixed order; a relation that lies inside the workgroup's range is rounded and stored at once, the (at most two)
//     partial ones per workgroup go to fp32 slabs that a small fix-up launch adds in workgroup order.  dW is the same
//     bits in every run, like the reference's per-relation at::matmul.
#include "matmul_dw_out.h"

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
template <typename Tag>
struct OutOf;
template <>
struct OutOf<bf16_tag> {
  using type = bf16_t;
};
template <>
struct OutOf<f16_tag> {
  using type = f16_t;
};

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
                                                         float* __restrict__ slabs,
                                                         typename OutOf<Tag>::type* __restrict__ out) {
  constexpr int IB = K / 32, JB = MC / 32;
  constexpr int PX = pitch_bytes(K), PY = pitch_bytes(MC);
  constexpr int CX = K / 8, CY = MC / 8;          // 16-byte chunks per row
  constexpr int NX = 32 * CX / 64, NY = 32 * CY / 64;  // chunk loads per lane and tile
  static_assert(IB * JB <= 16, "accumulators must fit the register file");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int swave = __builtin_amdgcn_readfirstlane(wave);
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

  // a relation ends: waves combined through LDS (the images are dead: every wave is past the MFMAs of the previous tile),
  // then stored (whole relation inside [t_beg, t_end)) or parked in the head / tail slab (matmul_dw_out.h)
  auto flush = [&]() {
    if (acc_g < 0) return;
    using Pos = DwPosRows<IB, JB>;
    const int u0 = tile_start[acc_g], u1 = tile_start[acc_g + 1];
    const bool head = u0 < t_beg, tail = !head && u1 > t_end;
    float* slab = dw_slab(slabs, (int64_t)IB * JB * kDwBlockFloats, ncol, bx, by, head ? 0 : 1);
    typename OutOf<Tag>::type* base = out + ((int64_t)acc_g * K) * M + col0;
    dw_combine_waves<IB, JB>(acc, smem, swave, lane, [&](int bt, int q, dwf4 s) __attribute__((always_inline)) {
      if (head || tail) dw_quarter_to_slab(slab, bt, q, s, lane);
      else dw_quarter_to_out<Pos>(base, M, K, MC, bt, q, s, lane);
    });
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
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
                                                                float* __restrict__ slabs,
                                                                typename OutOf<Tag>::type* __restrict__ out) {
  constexpr int K = 256, MC = 256, IB = K / 32, JB = 2;
  constexpr int PX = pitch_bytes(K), PY = pitch_bytes(MC);
  constexpr int CX = K / 8, CY = MC / 8;                 // 16-byte chunks per row
  constexpr int NX = 32 * CX / 256, NY = 32 * CY / 256;  // chunk loads per THREAD and slab
  constexpr int IMG = 32 * (PX + PY);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int swave = __builtin_amdgcn_readfirstlane(wave);
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
  // the waves own disjoint COLUMNS of dW: nothing to combine; whole relation -> dW, partial -> this wave's part of the slab
  auto flush = [&]() {
    if (acc_g < 0) return;
    const int u0 = tile_start[acc_g], u1 = tile_start[acc_g + 1];
    const bool head = u0 < t_beg, tail = !head && u1 > t_end;
    float* slab = dw_slab(slabs, (int64_t)DwPosWide::kBlocks * kDwBlockFloats, ncol, bx, by, head ? 0 : 1);
    typename OutOf<Tag>::type* base = out + ((int64_t)acc_g * K) * M + col0;
    // Through a wave-private 16 KiB staging area behind the two images (both hold live slabs here), four blocks at a
    // time: the registers are written with constant indices, the stores walk the area in a ROLLED loop -- the direct
    // form (64 blocks x 4 quarters of address arithmetic next to 256 live accumulators) spilled 2 KB per lane.
    char* stage = smem + 2 * IMG + wave * 16384 + lane * 16;
#pragma unroll
    for (int Q = 0; Q < 4; ++Q) {
#pragma unroll
      for (int L = 0; L < 4; ++L) {
        const int b = Q * 4 + L;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const dwf4 v = {acc[b / JB][b % JB][4 * q], acc[b / JB][b % JB][4 * q + 1], acc[b / JB][b % JB][4 * q + 2],
                          acc[b / JB][b % JB][4 * q + 3]};
          *reinterpret_cast<dwf4*>(stage + (L * 4 + q) * 1024) = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll 1
      for (int it = 0; it < 16; ++it) {
        const dwf4 v = *reinterpret_cast<const dwf4*>(stage + it * 1024);
        const int bt = swave * 16 + Q * 4 + (it >> 2), q = it & 3;
        if (head || tail) dw_quarter_to_slab(slab, bt, q, v, lane);
        else dw_quarter_to_out<DwPosWide>(base, M, K, MC, bt, q, v, lane);
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
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
                                                             float* __restrict__ slabs, float* __restrict__ out) {
  constexpr int IB = K / 32, JB = MC / 32;
  static_assert(IB * JB <= 16, "accumulators must fit the register file");
  // The MFMA wants ONE value per lane and block, but nothing ties block i to the columns 32 i .. 32 i + 31:
  // lane li loads VA (up to 4) CONSECUTIVE floats of its row in one instruction and feeds them to VA
  // different blocks (block VA h + e = columns 32 VA h + VA li + e) -- 16-byte instead of 4-byte accesses,
  // whole 512-byte rows per wave instruction; the permutation is undone when the accumulators are flushed.
  constexpr int VA = IB >= 4 ? 4 : IB, VB = JB >= 4 ? 4 : JB;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // IB * JB * 4 KiB: the waves' combine area (flush only)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int swave = __builtin_amdgcn_readfirstlane(wave);
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
  // block i = VA h + e holds the X columns 32 VA h + VA * (MFMA row) + e (see the loads below): DwPosF32 undoes that
  auto flush = [&]() {
    if (acc_g < 0) return;
    using Pos = DwPosF32<IB, JB>;
    const int u0 = tile_start[acc_g], u1 = tile_start[acc_g + 1];
    const bool head = u0 < t_beg, tail = !head && u1 > t_end;
    float* slab = dw_slab(slabs, (int64_t)IB * JB * kDwBlockFloats, ncol, bx, by, head ? 0 : 1);
    float* base = out + ((int64_t)acc_g * K) * M + col0;
    dw_combine_waves<IB, JB>(acc, smem, swave, lane, [&](int bt, int q, dwf4 s) __attribute__((always_inline)) {
      if (head || tail) dw_quarter_to_slab(slab, bt, q, s, lane);
      else dw_quarter_to_out<Pos>(base, M, K, MC, bt, q, s, lane);
    });
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
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

// The launch behind the main kernel.  Workgroups [0, fix_blocks): wave v = 4 blockIdx + wave is (logical main workgroup
// m = v / (4 NBT), slab block bt, quarter q); if a split relation STARTS in m's tile range, the wave adds that relation's
// slabs in workgroup order (dw_chain_sum) and stores 4 x 64 rounded values.  Workgroups behind them: relation
// blockIdx - fix_blocks; one without rows gets its zeros here (nothing else ever writes it).
template <typename Pos, typename OutT>
__global__ __launch_bounds__(256) void seg_dw_fixup_kernel(const int32_t* __restrict__ tile_start, int B, int K, int M, int MC,
                                                          int G, int fix_blocks, const float* __restrict__ slabs,
                                                          OutT* __restrict__ out) {
  constexpr int NBT = Pos::kBlocks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x >= fix_blocks) {
    const int g = (int)blockIdx.x - fix_blocks;
    if (tile_start[g + 1] != tile_start[g]) return;
    OutT* base = out + (int64_t)g * K * M;
    for (int i = threadIdx.x; i < K * M; i += 256) dw_put(base + i, 0.0f);
    return;
  }
  const int ncol = M / MC;
  const int v = (int)blockIdx.x * 4 + wave;
  const int m = v / (4 * NBT), piece = v - m * (4 * NBT);
  const int bt = piece >> 2, q = piece & 3;
  const int bx = m / ncol, by = m - bx * ncol;
  const int total = tile_start[B];
  const int t_beg = (int)((int64_t)bx * total / G), t_end = (int)((int64_t)(bx + 1) * total / G);
  if (t_beg >= t_end) return;
  int lo = 0, hi = B;  // relation of the range's last tile
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_start[mid] <= t_end - 1) lo = mid; else hi = mid;
  }
  const int u0 = tile_start[lo], u1 = tile_start[lo + 1];
  if (!(u1 > t_end && u0 >= t_beg)) return;  // no split relation starts here
  const dwf4 s = dw_chain_sum(slabs, (int64_t)NBT * kDwBlockFloats, ncol, bx, by, G, total, u1, bt, q, lane);
  dw_quarter_to_out<Pos>(out + (int64_t)lo * K * M + by * MC, M, K, MC, bt, q, s, lane);
}

inline size_t dw_groups_bytes(int64_t B) { return align_up(sizeof(DwGroup) * (size_t)(B > 0 ? B : 1), 256); }
inline size_t dw_tiles_bytes(int64_t B) { return align_up(sizeof(int32_t) * (size_t)(B + 1), 256); }

// column chunk of the kernel a (dtype, K, M) runs (run_dw / run_dw_f32 below) and the grid it gets
inline int dw_mc(bool f32, int64_t K, int64_t M) {
  if (!f32 && K == 256 && M % 256 == 0) return 256;
  if (K == 256) return 64;
  return M % 128 == 0 ? 128 : 64;
}
inline int64_t dw_grid_x(int64_t tiles_upper, int64_t ncol) {
  int64_t gx = std::max<int64_t>(1, std::min<int64_t>(tiles_upper, device_info().num_cus / ncol));
  if (ncol > 1) gx = (gx + 7) / 8 * 8;  // the kernels' XCD-aware decode works on groups of 8 ids
  return gx;
}
// two fp32 slabs of K x MC per main workgroup; the bound holds for every tile count and for both the 16-bit and the
// fp32 kernel of the shape
inline size_t dw_slab_bytes(int64_t K, int64_t M) {
  if (!((K == 64 || K == 128 || K == 256) && M > 0 && M % 64 == 0)) return 0;  // (dw_fast_shape: the other shapes run matmul_dw_gen.hip)
  size_t worst = 0;
  for (int f32 = 0; f32 < 2; ++f32) {
    const int64_t mc = dw_mc(f32 != 0, K, M), ncol = M / mc;
    const int64_t gx = dw_grid_x(INT64_MAX, ncol);
    worst = std::max(worst, (size_t)(gx * ncol) * 2 * (size_t)K * (size_t)mc * sizeof(float));
  }
  return worst;
}
inline size_t dw_ws_bytes(int64_t B, int64_t K, int64_t M) {
  return align_up(sizeof(int64_t) * (size_t)(B + 1), 256) + dw_groups_bytes(B) + dw_tiles_bytes(B) + dw_slab_bytes(K, M);
}

template <typename Pos, typename OutT>
int launch_fixup(const int32_t* tile_start, int64_t B, int64_t K, int64_t M, int MC, int64_t gx, const float* slabs,
                 OutT* out, hipStream_t stream) {
  const int64_t ncol = M / MC;
  const int64_t fix_blocks = gx * ncol * Pos::kBlocks;  // 4 NBT waves per main workgroup, 4 waves per block
  hipLaunchKernelGGL((seg_dw_fixup_kernel<Pos, OutT>), dim3((unsigned)(fix_blocks + B)), dim3(256), 0, stream, tile_start,
                     (int)B, (int)K, (int)M, MC, (int)gx, (int)fix_blocks, slabs, out);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

template <typename Tag, int K, int MC>
int launch_dw(const DwGroup* groups, const int32_t* tile_start, int64_t B, int64_t M, int64_t tiles_upper, float* slabs,
              void* out_, hipStream_t stream) {
  using OutT = typename OutOf<Tag>::type;
  OutT* out = static_cast<OutT*>(out_);
  constexpr int lds = 4 * 32 * (pitch_bytes(K) + pitch_bytes(MC));
  static_assert(lds >= (K / 32) * (MC / 32) * 4096, "the combine area must fit the row images");
  const void* kern = reinterpret_cast<const void*>(&seg_dw_kernel<Tag, K, MC>);
  if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
  const int64_t ncol = M / MC;
  const int64_t gx = dw_grid_x(tiles_upper, ncol);
  hipLaunchKernelGGL((seg_dw_kernel<Tag, K, MC>), dim3((unsigned)(gx * ncol)), dim3(256), lds, stream, groups,
                     tile_start, (int)B, (int)M, slabs, out);
  PYG_HIP_CHECK(hipGetLastError());
  return launch_fixup<DwPosRows<K / 32, MC / 32>, OutT>(tile_start, B, K, M, MC, gx, slabs, out, stream);
}

template <typename Tag>
int launch_dw_wide256(const DwGroup* groups, const int32_t* tile_start, int64_t B, int64_t M, int64_t tiles_upper,
                      float* slabs, void* out_, hipStream_t stream) {
  using OutT = typename OutOf<Tag>::type;
  OutT* out = static_cast<OutT*>(out_);
  constexpr int lds = 2 * 32 * (pitch_bytes(256) + pitch_bytes(256)) + 4 * 16384;  // two images + the flush staging area
  const void* kern = reinterpret_cast<const void*>(&seg_dw_wide256_kernel<Tag>);
  if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
  const int64_t ncol = M / 256;
  const int64_t gx = dw_grid_x(tiles_upper, ncol);
  hipLaunchKernelGGL((seg_dw_wide256_kernel<Tag>), dim3((unsigned)(gx * ncol)), dim3(256), lds, stream, groups, tile_start,
                     (int)B, (int)M, slabs, out);
  PYG_HIP_CHECK(hipGetLastError());
  return launch_fixup<DwPosWide, OutT>(tile_start, B, 256, M, 256, gx, slabs, out, stream);
}

template <typename Tag>
int run_dw(const DwGroup* groups, const int32_t* tile_start, int64_t B, int64_t K, int64_t M, int64_t tiles_upper,
           float* slabs, void* out, hipStream_t stream) {
  if (K == 256 && M % 256 == 0) return launch_dw_wide256<Tag>(groups, tile_start, B, M, tiles_upper, slabs, out, stream);
  if (K == 128 && M % 128 == 0) return launch_dw<Tag, 128, 128>(groups, tile_start, B, M, tiles_upper, slabs, out, stream);
  if (K == 128 && M % 64 == 0) return launch_dw<Tag, 128, 64>(groups, tile_start, B, M, tiles_upper, slabs, out, stream);
  if (K == 64 && M % 128 == 0) return launch_dw<Tag, 64, 128>(groups, tile_start, B, M, tiles_upper, slabs, out, stream);
  if (K == 64 && M % 64 == 0) return launch_dw<Tag, 64, 64>(groups, tile_start, B, M, tiles_upper, slabs, out, stream);
  if (K == 256 && M % 64 == 0) return launch_dw<Tag, 256, 64>(groups, tile_start, B, M, tiles_upper, slabs, out, stream);
  return fail(PYG_HIP_ERR_UNSUPPORTED, "segment_matmul_dw: K=%lld, M=%lld has no MFMA kernel (K in {64,128,256}, M %% 64 == 0)",
              (long long)K, (long long)M);
}

template <int K, int MC>
int launch_dw_f32(const DwGroup* groups, const int32_t* tile_start, int64_t B, int64_t M, int64_t tiles_upper, float* slabs,
                  void* out_, hipStream_t stream) {
  float* out = static_cast<float*>(out_);
  constexpr int lds = (K / 32) * (MC / 32) * 4096;  // the waves' combine area
  const void* kern = reinterpret_cast<const void*>(&seg_dw_f32_kernel<K, MC>);
  if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
  const int64_t ncol = M / MC;
  const int64_t gx = dw_grid_x(tiles_upper, ncol);
  hipLaunchKernelGGL((seg_dw_f32_kernel<K, MC>), dim3((unsigned)(gx * ncol)), dim3(256), lds, stream, groups, tile_start,
                     (int)B, (int)M, slabs, out);
  PYG_HIP_CHECK(hipGetLastError());
  return launch_fixup<DwPosF32<K / 32, MC / 32>, float>(tile_start, B, K, M, MC, gx, slabs, out, stream);
}

int run_dw_f32(const DwGroup* groups, const int32_t* tile_start, int64_t B, int64_t K, int64_t M, int64_t tiles_upper,
               float* slabs, void* out, hipStream_t stream) {
  if (K == 128 && M % 128 == 0) return launch_dw_f32<128, 128>(groups, tile_start, B, M, tiles_upper, slabs, out, stream);
  if (K == 128 && M % 64 == 0) return launch_dw_f32<128, 64>(groups, tile_start, B, M, tiles_upper, slabs, out, stream);
  if (K == 64 && M % 128 == 0) return launch_dw_f32<64, 128>(groups, tile_start, B, M, tiles_upper, slabs, out, stream);
  if (K == 64 && M % 64 == 0) return launch_dw_f32<64, 64>(groups, tile_start, B, M, tiles_upper, slabs, out, stream);
  if (K == 256 && M % 64 == 0) return launch_dw_f32<256, 64>(groups, tile_start, B, M, tiles_upper, slabs, out, stream);
  return fail(PYG_HIP_ERR_UNSUPPORTED, "segment_matmul_dw: K=%lld, M=%lld has no MFMA kernel (K in {64,128,256}, M %% 64 == 0)",
              (long long)K, (long long)M);
}

// shapes of the specialised kernels above; everything else (and operands that are not 16-byte aligned) runs
// matmul_dw_gen.hip
inline bool dw_fast_shape(int64_t K, int64_t M) { return (K == 64 || K == 128 || K == 256) && M > 0 && M % 64 == 0; }

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
  return std::max(dw_ws_bytes(B, K, M), dw_gen_workspace_bytes(B));
}

size_t pyg_hip_grouped_matmul_dw_workspace_size(const pyg_hip_group* groups, int64_t G) {
  if (G <= 0 || groups == nullptr) return dw_gen_workspace_bytes(0);
  bool uniform = true;
  for (int64_t i = 0; i < G; ++i) uniform = uniform && groups[i].k == groups[0].k && groups[i].m == groups[0].m;
  const size_t gen = dw_gen_workspace_bytes(G);
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
  float* slabs = reinterpret_cast<float*>(w);
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
  // descriptors address the tensors in 2-byte units: an fp32 row is 2 K of them
  const int64_t u = dtype == PYG_F32 ? 2 : 1;
  hipLaunchKernelGGL(dw_plan_kernel, dim3(1), dim3(256), 0, stream, dptr, B, static_cast<const uint16_t*>(input),
                     static_cast<const uint16_t*>(grad_out), K * u, M * u, groups, tile_start);
  PYG_HIP_CHECK(hipGetLastError());
  const int64_t tiles_upper = (N + kTile - 1) / kTile + B;
  return dtype == PYG_F32    ? run_dw_f32(groups, tile_start, B, K, M, tiles_upper, slabs, grad_other, stream)
         : dtype == PYG_BF16 ? run_dw<bf16_tag>(groups, tile_start, B, K, M, tiles_upper, slabs, grad_other, stream)
                             : run_dw<f16_tag>(groups, tile_start, B, K, M, tiles_upper, slabs, grad_other, stream);
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
  float* slabs = reinterpret_cast<float*>(w);
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
  return dtype == PYG_F32    ? run_dw_f32(groups, tile_start, G, K, M, tiles + 1, slabs, out_pool, stream)
         : dtype == PYG_BF16 ? run_dw<bf16_tag>(groups, tile_start, G, K, M, tiles + 1, slabs, out_pool, stream)
                             : run_dw<f16_tag>(groups, tile_start, G, K, M, tiles + 1, slabs, out_pool, stream);
}

}  // extern "C"
