// General-shape weight gradient for gfx950 (MI355X):  dW[g] = X_g^T @ dY_g  with per-group (K_g, M_g) of ANY size and
// any element-aligned operands -- the counterpart of matmul_gen.hip for the backward pass.
//
// Replaces the per-relation at::matmul + at::stack loop of SegmentMatmul::backward
// (pyg_lib/csrc/ops/autograd/matmul_kernel.cpp:92-107) and the Python loop behind GroupedMatmul.backward
// (pyg_lib/ops/__init__.py:88-94) for every shape the shape-specialised kernels of matmul_dw.hip do not cover:
// K = 100 (ogbn-products), mixed-K HeteroDictLinear lists, odd M.
//
//   * The output of a group is cut into blocks of KB x MB = 128 x 128 (fp32: 128 x 64) entries; a work tile is
//     (group, block, 128 consecutive rows).  Tiles are numbered group-major, block-major, row-tile-minor, every
//     workgroup walks a contiguous range of them: it keeps a block's fp32 accumulators in registers across its row
//     tiles and hands them over only when the block changes -- without atomics (matmul_dw_out.h): the waves are added
//     through LDS in a fixed order, a block that lies inside the workgroup's range is rounded and stored at once, the
//     (at most two) partial ones per workgroup go to fp32 slabs that the fix-up launch adds in workgroup order.
//     K <= 128 and M <= 128 read X and dY exactly once; wider outputs re-read X per column block and dY per row block.
//   * As in seg_dw_kernel each of the 4 waves owns 32 rows of the tile and a full block of accumulators (1 wave per
//     SIMD), parks its rows in a wave-private row-major LDS image -- columns beyond K / M zero-filled, so the
//     MFMAs never see a tail -- and reads the MFMA operands back transposed (ds_read_b64_tr_b16).  The fp32 kernel
//     skips 32-column sub-blocks that lie entirely beyond K or M (its MFMAs are 8x slower per element).
//   * Rows are fetched with the widest vector every operand of the LAUNCH allows (16 / 8 / 4 / 2 bytes: the smallest
//     alignment class over its groups' base addresses and row pitches selects one of four instantiations -- a class
//     switch inside the loop made the compiler copy the loaded registers at the merge, i.e. wait for the loads where
//     they were issued: 2.1 instead of 5.6 TB/s); a chunk is 16 >> LG individually predicated loads, so row tails need
//     no separate path.
//   * fp32 runs v_mfma_f32_32x32x2_f32 (IEEE fp32 products and sums, like the exact forward kernel): its operands are
//     "2 rows x 32 columns", read from the same row-major image with plain 4-byte LDS reads (no transpose).
#include "matmul_dw_out.h"

#include <stdint.h>
#include <string.h>

#include <algorithm>

namespace pyg_hip {
namespace {

typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef short v8i16 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kTile = 128;  // rows per workgroup tile (4 waves x 32)

struct DwGenGroup {  // 48 bytes
  const char* x;     // [rows, k] row-major
  const char* dy;    // [rows, m] row-major
  int64_t rows;
  int64_t acc_off;   // first element of this group's [k, m] block in the fp32 image (and in the output pool)
  int32_t k, m;
  int16_t lx, ly;    // log2 of the vector bytes the X / dY rows may be fetched with (1 ... 4)
  int16_t nkb, nmb;  // blocks along k and m
};
static_assert(sizeof(DwGenGroup) == 48, "DwGenGroup layout");

template <typename T>
struct GenCfg;
template <>
struct GenCfg<bf16_t> {
  static constexpr int IB = 4, JB = 4;
};
template <>
struct GenCfg<f16_t> {
  static constexpr int IB = 4, JB = 4;
};
template <>
struct GenCfg<float> {
  static constexpr int IB = 4, JB = 2;
};

// pitch (bytes) of a row-major [32][cols] LDS image.  16-bit: == 16 (mod 64) dwords (the transpose reads of two 16-lane
// groups and the 4 rows of a block fall on disjoint banks, as in matmul_dw.hip); fp32: == 32 (mod 64) dwords (lanes
// 0-31 read 32 consecutive floats of one row, lanes 32-63 of the next).
template <int ELT>
constexpr int gen_pitch(int cols) {
  if (ELT == 2) {
    const int dw = cols / 2;
    const int extra = dw > 16 ? (dw - 16 + 63) / 64 : 0;
    return (16 + 64 * extra) * 4;
  }
  const int dw = cols;
  const int extra = dw > 32 ? (dw - 32 + 63) / 64 : 0;
  return (32 + 64 * extra) * 4;
}

// 16 bytes of row `rowp` starting at element `c0`, zero beyond element `n` of the row and for rows behind the group
// (`row_ok`).  LG = log2 of the vector bytes every piece may be fetched with (the launch's alignment class): the chunk is
// 16 >> LG independent, individually predicated loads -- the row pitch is a multiple of 2^LG bytes, so a row's tail is a
// whole number of pieces.  No branch arm holds more than a load: the compiler needs no copy (and hence no wait) between
// the loads and their use a tile later.  (LG = 1 assembles dwords from halves and is only used un-pipelined.)
template <int ELT, int LG>
// (global_* loads: see matmul_dw.hip)
__device__ __forceinline__ u32x4 load_chunk(const char* rowp, int c0, int n, bool row_ok) {
  u32x4 v = {0u, 0u, 0u, 0u};
  const int vb = row_ok ? (n - c0) * ELT : 0;  // valid bytes from the chunk start (<= 0: nothing, >= 16: all)
  const char* p = rowp + (int64_t)c0 * ELT;
  if constexpr (LG >= 4) {
    if (vb >= 16) v = *(const __attribute__((address_space(1))) u32x4*)(p);
  } else if constexpr (LG == 3) {
    u32x2 a = {0u, 0u}, b = {0u, 0u};
    if (vb >= 8) a = *(const __attribute__((address_space(1))) u32x2*)(p);
    if (vb >= 16) b = *(const __attribute__((address_space(1))) u32x2*)(p + 8);
    v[0] = a[0], v[1] = a[1], v[2] = b[0], v[3] = b[1];
  } else if constexpr (LG == 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (vb >= 4 * (e + 1)) v[e] = *(const __attribute__((address_space(1))) uint32_t*)(p + 4 * e);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (vb >= 2 * (e + 1)) v[e >> 1] |= (uint32_t) * (const __attribute__((address_space(1))) uint16_t*)(p + 2 * e) << (16 * (e & 1));
  }
  return v;
}

// The N chunk loads of one operand of a tile: chunk p = it * 64 + lane of the wave's 32 x (C * E)-element slab.
template <int ELT, int LG, int C, int N>
__device__ __forceinline__ void load_rows(u32x4 (&r)[N], const char* base, int64_t row0, int64_t rows, int width, int c0,
                                          int lane) {
  constexpr int E = 16 / ELT;
#pragma unroll
  for (int it = 0; it < N; ++it) {
    const int p = it * 64 + lane;
    int64_t row = row0 + p / C;
    const bool ok = row < rows;
    if (!ok) row = 0;  // (address of a predicated-off load: anything inside the tensor)
    r[it] = load_chunk<ELT, LG>(base + row * width * ELT, c0 + (p % C) * E, width, ok);
  }
}

__device__ __forceinline__ f32x16 gen_mfma16(bf16_t, v8i16 a, v8i16 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 gen_mfma16(f16_t, v8i16 a, v8i16 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// What the loop needs to know about the tile it holds / the block its accumulators belong to.
struct TileKey {
  int g;       // group
  int blk;     // block of the group (kb * nmb + mb)
  int kb, mb;  // block coordinates
  int k, m;    // the group's shape
  int u0, u1;  // tiles of this (group, block): the unit the accumulators belong to
  int64_t acc_off;
};

template <typename T, int LG>
__global__ __launch_bounds__(256, 1) void dw_gen_kernel(const DwGenGroup* __restrict__ groups,
                                                         const int32_t* __restrict__ tile_start, int B,
                                                         float* __restrict__ slabs, T* __restrict__ out) {
  constexpr int ELT = Elem<T>::kSize, E = 16 / ELT;
  constexpr int IB = GenCfg<T>::IB, JB = GenCfg<T>::JB;
  constexpr int KB = 32 * IB, MB = 32 * JB;
  constexpr int PX = gen_pitch<ELT>(KB), PY = gen_pitch<ELT>(MB);
  constexpr int CX = KB / E, CY = MB / E;              // 16-byte chunks per image row
  constexpr int NX = 32 * CX / 64, NY = 32 * CY / 64;  // chunk loads per lane and tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int swave = __builtin_amdgcn_readfirstlane(wave);
  char* xs = smem + wave * 32 * (PX + PY);
  char* ys = xs + 32 * PX;

  const int total = tile_start[B];
  const int G = (int)gridDim.x;
  const int t_beg = (int)((int64_t)blockIdx.x * total / G);
  const int t_end = (int)((int64_t)(blockIdx.x + 1) * total / G);
  if (t_beg >= t_end) return;
  int n_g = 0;
  {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tile_start[mid] <= t_beg) lo = mid; else hi = mid;
    }
    n_g = lo;
  }

  f32x16 acc[IB][JB];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  TileKey ak;  // block the accumulators belong to
  ak.g = -1;
  ak.blk = 0, ak.kb = 0, ak.mb = 0, ak.k = 0, ak.m = 0, ak.u0 = 0, ak.u1 = 0, ak.acc_off = 0;

  // A block ends: the waves are combined through LDS (the images are dead: every wave is past the MFMAs of the
  // previous tile), then the block is stored (it lies inside [t_beg, t_end)) or parked in the head / tail slab.
  auto flush = [&]() {
    if (ak.g < 0) return;
    using Pos = DwPosRows<IB, JB>;
    const bool head = ak.u0 < t_beg, tail = !head && ak.u1 > t_end;
    float* slab = dw_slab(slabs, (int64_t)IB * JB * kDwBlockFloats, 1, (int)blockIdx.x, 0, head ? 0 : 1);
    T* base = out + ak.acc_off + (int64_t)ak.kb * KB * ak.m + ak.mb * MB;
    const int pitch = ak.m, k_lim = ak.k - ak.kb * KB, m_lim = ak.m - ak.mb * MB;
    dw_combine_waves<IB, JB>(acc, smem, swave, lane, [&](int bt, int q, dwf4 s) __attribute__((always_inline)) {
      if (head || tail) dw_quarter_to_slab(slab, bt, q, s, lane);
      else dw_quarter_to_out<Pos>(base, pitch, k_lim, m_lim, bt, q, s, lane);
    });
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  };

  // software pipeline: rows of tile t + 1 travel to registers while tile t is multiplied
  u32x4 xr[NX], yr[NY];
  DwGenGroup gd = groups[n_g];
  TileKey nk;  // the tile held in xr / yr
  auto prefetch = [&](int t) {
    while (t >= tile_start[n_g + 1]) {
      ++n_g;
      gd = groups[n_g];
    }
    const int local = t - tile_start[n_g];
    const int rt_n = (int)((gd.rows + kTile - 1) / kTile);
    const int blk = local / rt_n;
    const int rt = local - blk * rt_n;
    nk.g = n_g;
    nk.blk = blk;
    nk.kb = blk / gd.nmb;
    nk.mb = blk - nk.kb * gd.nmb;
    nk.k = gd.k;
    nk.m = gd.m;
    nk.acc_off = gd.acc_off;
    nk.u0 = tile_start[n_g] + blk * rt_n;
    nk.u1 = nk.u0 + rt_n;
    const int64_t row0 = (int64_t)rt * kTile + wave * 32;
    const int kc0 = nk.kb * KB, mc0 = nk.mb * MB;
    load_rows<ELT, LG, CX, NX>(xr, gd.x, row0, gd.rows, gd.k, kc0, lane);
    load_rows<ELT, LG, CY, NY>(yr, gd.dy, row0, gd.rows, gd.m, mc0, lane);
  };

  // lane constants of the 16-bit transpose reads (see seg_dw_kernel): lane q of a 16-lane group supplies row (q >> 2),
  // columns (q & 3) * 4 of a 4 x 16 block and receives column q; group = (column half, row half kb)
  const int q = lane & 15, half = (lane >> 4) & 1, khalf = lane >> 5;
  const int a_off = (khalf * 8 + (q >> 2)) * PX + (half * 16 + (q & 3) * 4) * 2;
  const int b_off = (khalf * 8 + (q >> 2)) * PY + (half * 16 + (q & 3) * 4) * 2;
  typedef __attribute__((address_space(3))) v4i16* lds_v4;

  // LG >= 2: the rows of tile t + 1 travel while tile t is multiplied; LG = 1 (operands aligned to the element only)
  // assembles its dwords from halves, which needs the data at once: fetched at the top of its own iteration
  constexpr bool PIPE = LG >= 2;
  if constexpr (PIPE) prefetch(t_beg);
  for (int t = t_beg; t < t_end; ++t) {
    if constexpr (!PIPE) prefetch(t);
    if (nk.g != ak.g || nk.blk != ak.blk) {
      flush();
      ak = nk;
    }
    // 32-column sub-blocks of this block that hold data
    [[maybe_unused]] const int ib_n = min(IB, (ak.k - ak.kb * KB + 31) >> 5), jb_n = min(JB, (ak.m - ak.mb * MB + 31) >> 5);
    // park the tile in the wave-private LDS image (row-major, zero-filled tails)
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
    if constexpr (PIPE)
      if (t + 1 < t_end) prefetch(t + 1);
    if constexpr (ELT == 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        // every sub-block is multiplied: the image's zero-filled tails make that exact, the 16-bit MFMAs are far from
        // limiting (a guard per sub-block costs the register allocator its AGPR accumulators)
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
          for (int j = 0; j < JB; ++j) acc[i][j] = gen_mfma16(T{}, af[i], bf[j], acc[i][j]);
      }
    } else {
      // fp32: one MFMA contracts 2 rows; lane l supplies X[2 s + (l >> 5)][32 i + (l & 31)] and the dY counterpart
      const char* xl = xs + (lane >> 5) * PX + (lane & 31) * 4;
      const char* yl = ys + (lane >> 5) * PY + (lane & 31) * 4;
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        float a[IB], b[JB];
#pragma unroll
        for (int i = 0; i < IB; ++i)
          if (i < ib_n) a[i] = *reinterpret_cast<const float*>(xl + 2 * s * PX + 128 * i);
#pragma unroll
        for (int j = 0; j < JB; ++j)
          if (j < jb_n) b[j] = *reinterpret_cast<const float*>(yl + 2 * s * PY + 128 * j);
#pragma unroll
        for (int i = 0; i < IB; ++i)
#pragma unroll
          for (int j = 0; j < JB; ++j)
            if (i < ib_n && j < jb_n) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  flush();
}

// segment form: descriptors + tile prefix from `ptr` (uniform K, M; one thread per run of relations)
__global__ void dw_gen_plan_kernel(const int64_t* __restrict__ ptr, int64_t B, const char* X, const char* dY, int64_t K,
                                   int64_t M, int elt, int kb_size, int mb_size, DwGenGroup* __restrict__ groups,
                                   int32_t* __restrict__ tile_start) {
  __shared__ int64_t part[256];
  const int tid = threadIdx.x;
  const int64_t per = (B + 255) / 256;
  const int64_t beg = min((int64_t)tid * per, B), end = min(beg + per, B);
  const int nkb = (int)((K + kb_size - 1) / kb_size), nmb = (int)((M + mb_size - 1) / mb_size);
  int64_t t = 0;
  for (int64_t b = beg; b < end; ++b) {
    const int64_t r = ptr[b + 1] - ptr[b];
    t += r > 0 ? (r + kTile - 1) / kTile * nkb * nmb : 0;
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
    DwGenGroup d;
    d.x = X + p0 * K * elt;
    d.dy = dY + p0 * M * elt;
    d.rows = r > 0 ? r : 0;
    d.acc_off = b * K * M;
    d.k = (int32_t)K;
    d.m = (int32_t)M;
    d.lx = (int16_t)min(gen_log2_align((uint64_t)d.x), gen_log2_align((uint64_t)(K * elt)));
    d.ly = (int16_t)min(gen_log2_align((uint64_t)d.dy), gen_log2_align((uint64_t)(M * elt)));
    d.nkb = (int16_t)nkb;
    d.nmb = (int16_t)nmb;
    groups[b] = d;
    t += r > 0 ? (r + kTile - 1) / kTile * nkb * nmb : 0;
  }
}

// The launch behind dw_gen_kernel (see seg_dw_fixup_kernel in matmul_dw.hip): workgroups [0, fix_blocks) add the slabs
// of the (group, block) units that are split over main workgroups, in workgroup order; the ones behind them write the
// zeros of groups without rows.
template <typename T>
__global__ __launch_bounds__(256) void dw_gen_fixup_kernel(const DwGenGroup* __restrict__ groups,
                                                          const int32_t* __restrict__ tile_start, int B, int G,
                                                          int fix_blocks, const float* __restrict__ slabs,
                                                          T* __restrict__ out) {
  constexpr int IB = GenCfg<T>::IB, JB = GenCfg<T>::JB, KB = 32 * IB, MB = 32 * JB, NBT = IB * JB;
  using Pos = DwPosRows<IB, JB>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x >= fix_blocks) {
    const DwGenGroup gd = groups[(int)blockIdx.x - fix_blocks];
    if (gd.rows > 0) return;
    T* base = out + gd.acc_off;
    const int64_t n = (int64_t)gd.k * gd.m;
    for (int64_t i = threadIdx.x; i < n; i += 256) dw_put(base + i, 0.0f);
    return;
  }
  const int v = (int)blockIdx.x * 4 + wave;
  const int bx = v / (4 * NBT), piece = v - bx * (4 * NBT);
  const int bt = piece >> 2, q = piece & 3;
  const int total = tile_start[B];
  const int t_beg = (int)((int64_t)bx * total / G), t_end = (int)((int64_t)(bx + 1) * total / G);
  if (t_beg >= t_end) return;
  int lo = 0, hi = B;  // group of the range's last tile
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_start[mid] <= t_end - 1) lo = mid; else hi = mid;
  }
  const DwGenGroup gd = groups[lo];
  const int rt_n = (int)((gd.rows + kTile - 1) / kTile);
  const int blk = (t_end - 1 - tile_start[lo]) / rt_n;
  const int u0 = tile_start[lo] + blk * rt_n, u1 = u0 + rt_n;
  if (!(u1 > t_end && u0 >= t_beg)) return;  // no split unit starts here
  const int kb = blk / gd.nmb, mb = blk - kb * gd.nmb;
  const dwf4 s = dw_chain_sum(slabs, (int64_t)NBT * kDwBlockFloats, 1, bx, 0, G, total, u1, bt, q, lane);
  dw_quarter_to_out<Pos>(out + gd.acc_off + (int64_t)kb * KB * gd.m + mb * MB, gd.m, gd.k - kb * KB, gd.m - mb * MB, bt, q, s,
                         lane);
}

template <typename T, int LG>
int launch_gen_lg(const DwGenGroup* groups, const int32_t* tile_start, int B, int64_t tiles_upper, float* slabs, void* out_,
                  hipStream_t stream) {
  constexpr int ELT = Elem<T>::kSize;
  constexpr int lds = 4 * 32 * (gen_pitch<ELT>(32 * GenCfg<T>::IB) + gen_pitch<ELT>(32 * GenCfg<T>::JB));
  static_assert(lds <= 160 * 1024, "dw_gen_kernel: LDS");
  static_assert(lds >= GenCfg<T>::IB * GenCfg<T>::JB * 4096, "the combine area must fit the row images");
  T* out = static_cast<T*>(out_);
  const void* kern = reinterpret_cast<const void*>(&dw_gen_kernel<T, LG>);
  if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
  const int64_t gx = std::max<int64_t>(1, std::min<int64_t>(tiles_upper, device_info().num_cus));
  hipLaunchKernelGGL((dw_gen_kernel<T, LG>), dim3((unsigned)gx), dim3(256), lds, stream, groups, tile_start, B, slabs, out);
  PYG_HIP_CHECK(hipGetLastError());
  const int64_t fix_blocks = gx * GenCfg<T>::IB * GenCfg<T>::JB;
  hipLaunchKernelGGL((dw_gen_fixup_kernel<T>), dim3((unsigned)(fix_blocks + B)), dim3(256), 0, stream, groups, tile_start, B,
                     (int)gx, (int)fix_blocks, slabs, out);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

// `lg`: alignment class of the launch = the smallest over its groups' operands (log2 of the vector bytes)
template <typename T>
int launch_gen(const DwGenGroup* groups, const int32_t* tile_start, int B, int64_t tiles_upper, float* slabs, void* out, int lg,
               hipStream_t stream) {
  if (lg >= 4) return launch_gen_lg<T, 4>(groups, tile_start, B, tiles_upper, slabs, out, stream);
  if (lg == 3) return launch_gen_lg<T, 3>(groups, tile_start, B, tiles_upper, slabs, out, stream);
  if constexpr (Elem<T>::kSize == 2) {
    if (lg <= 1) return launch_gen_lg<T, 1>(groups, tile_start, B, tiles_upper, slabs, out, stream);
  }
  return launch_gen_lg<T, 2>(groups, tile_start, B, tiles_upper, slabs, out, stream);
}

inline int gen_kb(int dtype) { return dtype == PYG_F32 ? 32 * GenCfg<float>::IB : 32 * GenCfg<bf16_t>::IB; }
inline int gen_mb(int dtype) { return dtype == PYG_F32 ? 32 * GenCfg<float>::JB : 32 * GenCfg<bf16_t>::JB; }

int run_gen(int dtype, const DwGenGroup* groups, const int32_t* tile_start, int64_t B, int64_t tiles_upper, float* slabs,
            void* out, int lg, hipStream_t stream) {
  return dtype == PYG_F32    ? launch_gen<float>(groups, tile_start, (int)B, tiles_upper, slabs, out, lg, stream)
         : dtype == PYG_BF16 ? launch_gen<bf16_t>(groups, tile_start, (int)B, tiles_upper, slabs, out, lg, stream)
                             : launch_gen<f16_t>(groups, tile_start, (int)B, tiles_upper, slabs, out, lg, stream);
}

inline size_t gen_groups_bytes(int64_t B) { return align_up(sizeof(DwGenGroup) * (size_t)(B > 0 ? B : 1), 256); }
inline size_t gen_tiles_bytes(int64_t B) { return align_up(sizeof(int32_t) * (size_t)(B + 1), 256); }

}  // namespace

size_t dw_gen_workspace_bytes(int64_t B) {
  // two fp32 slabs of one 128 x 128 block per main workgroup (the 16-bit configuration is the larger one)
  const size_t slabs = (size_t)device_info().num_cus * 2 * (size_t)(GenCfg<bf16_t>::IB * GenCfg<bf16_t>::JB) * kDwBlockFloats *
                       sizeof(float);
  return align_up(sizeof(int64_t) * (size_t)(B + 1), 256) + gen_groups_bytes(B) + gen_tiles_bytes(B) + slabs;
}

int dw_gen_segment(int dtype, const void* input, const int64_t* ptr, int ptr_on_device, const void* grad_out,
                   void* grad_other, int64_t N, int64_t K, int64_t M, int64_t B, void* workspace, hipStream_t stream) {
  const int elt = dtype == PYG_F32 ? 4 : 2;
  PYG_HIP_REQUIRE(((uintptr_t)input | (uintptr_t)grad_out | (uintptr_t)grad_other) % elt == 0,
                  "segment_matmul_dw: tensors must be element-aligned");
  if (K >= (1LL << 21) || M >= (1LL << 21) || K * M >= (1LL << 28))
    return fail(PYG_HIP_ERR_UNSUPPORTED, "segment_matmul_dw: K x M = %lld x %lld is beyond the kernel's 32-bit offsets", (long long)K,
                (long long)M);
  const int kb = gen_kb(dtype), mb = gen_mb(dtype);
  const int64_t blocks = ((K + kb - 1) / kb) * ((M + mb - 1) / mb);
  const int64_t tiles_upper = ((N + kTile - 1) / kTile + B) * blocks;
  PYG_HIP_REQUIRE(tiles_upper < (1LL << 31), "segment_matmul_dw: too many tiles");
  char* w = static_cast<char*>(workspace);
  int64_t* ptr_dev = reinterpret_cast<int64_t*>(w);
  w += align_up(sizeof(int64_t) * (size_t)(B + 1), 256);
  DwGenGroup* groups = reinterpret_cast<DwGenGroup*>(w);
  w += gen_groups_bytes(B);
  int32_t* tile_start = reinterpret_cast<int32_t*>(w);
  w += gen_tiles_bytes(B);
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
  hipLaunchKernelGGL(dw_gen_plan_kernel, dim3(1), dim3(256), 0, stream, dptr, B, static_cast<const char*>(input),
                     static_cast<const char*>(grad_out), K, M, elt, kb, mb, groups, tile_start);
  PYG_HIP_CHECK(hipGetLastError());
  // a relation starts ptr[b] rows into the tensors: its alignment is at least that of the base and the row pitch
  const int lg = std::min(std::min(gen_log2_align((uint64_t)input), gen_log2_align((uint64_t)(K * elt))),
                          std::min(gen_log2_align((uint64_t)grad_out), gen_log2_align((uint64_t)(M * elt))));
  return run_gen(dtype, groups, tile_start, B, tiles_upper, slabs, grad_other, lg, stream);
}

int dw_gen_grouped(int dtype, const pyg_hip_group* host_groups, int64_t G, void* out_pool, void* workspace,
                   hipStream_t stream) {
  const int elt = dtype == PYG_F32 ? 4 : 2;
  const int kb = gen_kb(dtype), mb = gen_mb(dtype);
  char* w = static_cast<char*>(workspace) + align_up(sizeof(int64_t) * (size_t)(G + 1), 256);
  DwGenGroup* groups = reinterpret_cast<DwGenGroup*>(w);
  w += gen_groups_bytes(G);
  int32_t* tile_start = reinterpret_cast<int32_t*>(w);
  w += gen_tiles_bytes(G);
  float* slabs = reinterpret_cast<float*>(w);
  void* staged = nullptr;
  int rc = pinned_stage().acquire(gen_groups_bytes(G) + gen_tiles_bytes(G), &staged);
  if (rc != PYG_HIP_OK) return rc;
  DwGenGroup* hg = static_cast<DwGenGroup*>(staged);
  int32_t* ht = reinterpret_cast<int32_t*>(static_cast<char*>(staged) + gen_groups_bytes(G));
  int64_t t = 0, off = 0;
  int lg = 4;
  for (int64_t i = 0; i < G; ++i) {
    const pyg_hip_group& g = host_groups[i];
    if (g.k >= (1 << 21) || g.m >= (1 << 21) || (int64_t)g.k * g.m >= (1LL << 28))
      return fail(PYG_HIP_ERR_UNSUPPORTED, "grouped_matmul_dw: K x M = %d x %d is beyond the kernel's 32-bit offsets", g.k, g.m);
    PYG_HIP_REQUIRE(((uintptr_t)g.input | (uintptr_t)g.other) % elt == 0, "grouped_matmul_dw: operands must be element-aligned");
    DwGenGroup d;
    d.x = static_cast<const char*>(g.input);
    d.dy = static_cast<const char*>(g.other);
    d.rows = g.rows;
    d.acc_off = off;
    d.k = g.k;
    d.m = g.m;
    d.lx = (int16_t)std::min(gen_log2_align((uint64_t)d.x), gen_log2_align((uint64_t)((int64_t)g.k * elt)));
    d.ly = (int16_t)std::min(gen_log2_align((uint64_t)d.dy), gen_log2_align((uint64_t)((int64_t)g.m * elt)));
    d.nkb = (int16_t)((g.k + kb - 1) / kb);
    d.nmb = (int16_t)((g.m + mb - 1) / mb);
    PYG_HIP_REQUIRE(d.nkb >= 0 && d.nmb >= 0 && (g.k + kb - 1) / kb < 32768 && (g.m + mb - 1) / mb < 32768,
                    "grouped_matmul_dw: K / M too large");
    if (g.rows > 0 && g.k > 0 && g.m > 0) lg = std::min(lg, (int)std::min(d.lx, d.ly));
    hg[i] = d;
    ht[i] = (int32_t)t;
    t += (g.rows + kTile - 1) / kTile * (int64_t)d.nkb * d.nmb;
    PYG_HIP_REQUIRE(t < (1LL << 31), "grouped_matmul_dw: too many tiles");
    off += (int64_t)g.k * g.m;
  }
  ht[G] = (int32_t)t;
  PYG_HIP_CHECK(hipMemcpyAsync(groups, staged, gen_groups_bytes(G) + gen_tiles_bytes(G), hipMemcpyHostToDevice, stream));
  rc = pinned_stage().commit(stream);
  if (rc != PYG_HIP_OK) return rc;
  if (off == 0) return PYG_HIP_OK;
  if (t == 0) {  // no rows anywhere: the result is all zeros
    PYG_HIP_CHECK(hipMemsetAsync(out_pool, 0, (size_t)elt * (size_t)off, stream));
    return PYG_HIP_OK;
  }
  return run_gen(dtype, groups, tile_start, G, t, slabs, out_pool, lg, stream);
}

}  // namespace pyg_hip
