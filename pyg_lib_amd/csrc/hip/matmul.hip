// segment_matmul / grouped_matmul for gfx950 (MI355X).
//
// Replaces pyg_lib/csrc/ops/cuda/matmul_kernel.cu (CUTLASS GemmGrouped, fp32 only) and the
// CPU path pyg_lib/csrc/ops/cpu/matmul_kernel.cpp:281-312,410-439.
//
// Design (DESIGN.md "segment_matmul"): the op is a stream of rows through a small per-relation
// weight, i.e. HBM-bound for bf16 (64 flop/B at F=128) and f32-MFMA bound for fp32.  One
// persistent launch walks a list of (group, 128-row tile) work items in contiguous ranges per
// workgroup, so a workgroup re-stages the relation's weight into LDS only when it crosses a
// segment boundary.  Every wave owns 32 rows x MC output columns:
//   * X rows go HBM -> VGPR directly as 16-byte loads; lane (x, h) owns the contiguous half-row
//     X[row x][h*K/2 .. (h+1)*K/2) -- the contraction index is permuted between MFMA k-slots so
//     that each lane's fragments are contiguous in memory (k = h*K/2 + 8*s + e for step s).
//   * W^T lives in LDS as [MC][K] (+16 B row pad => conflict-free ds_read_b128) and is the MFMA
//     "A" operand, X is the "B" operand, so D = W^T X^T: lane (x, h) ends up with
//     out[row x][h*MC/2 .. (h+1)*MC/2) -- again contiguous, stored as 16-byte writes.  The
//     output-column permutation that makes this true is folded into the LDS row a lane reads.
//   * v_mfma_f32_32x32x16_{bf16,f16} with fp32 accumulation and one rounding at the store for
//     16-bit types; v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain) for fp32 -- gfx950 has no TF32.
// Shapes outside the specialised (K, M) set, and the remaining dtypes of
// AT_DISPATCH_ALL_TYPES_AND2, run a plain one-thread-per-output kernel ("naive").
#include "matmul_common.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

namespace pyg_hip {
namespace {

constexpr int kTileRows = 128;  // rows per workgroup tile in the MFMA kernels (4 waves x 32)
constexpr int kPairRows = 64;   // rows per tile of the ticket kernel (2 waves x 32)
constexpr int kTicketWords = 256;  // 8 per-XCD tile counters, 128 bytes apart

// ---- plan kernel: ptr on device -> descriptors + tile/row prefix sums --------------------------
__global__ void plan_segments_kernel(const int64_t* __restrict__ ptr, int64_t B, const char* a,
                                     const char* w, char* c, const char* bias, int64_t K,
                                     int64_t M, int elt, DevGroup* __restrict__ descs,
                                     int32_t* __restrict__ tile_start,
                                     int64_t* __restrict__ row_start, int32_t* __restrict__ tile_start2,
                                     int32_t* __restrict__ tile_start3, unsigned int* __restrict__ tickets) {
  // Single block; B is the number of relations (hundreds): a serial-per-chunk scan is plenty.
  __shared__ int64_t s_tiles[256];
  __shared__ int64_t s_tiles2[256];
  __shared__ int64_t s_tiles3[256];
  __shared__ int64_t s_rows[256];
  const int tid = threadIdx.x;
  tickets[tid] = 0;  // kTicketWords == blockDim.x: the per-XCD tile counters of the ticket kernel start at 0
  const int nthr = blockDim.x;
  const int64_t per = (B + nthr - 1) / nthr;
  const int64_t beg = min((int64_t)tid * per, B), end = min(beg + per, B);
  int64_t tiles = 0, tiles2 = 0, tiles3 = 0, rows = 0;
  for (int64_t b = beg; b < end; ++b) {
    int64_t r = ptr[b + 1] - ptr[b];
    if (r < 0) r = 0;
    rows += r;
    tiles += (r + kTileRows - 1) / kTileRows;
    tiles2 += (r + 2 * kTileRows - 1) / (2 * kTileRows);
    tiles3 += (r + kPairRows - 1) / kPairRows;
  }
  // exclusive scan of the four per-thread sums over the block: wave shuffles + the wave totals through LDS (a serial
  // walk of thread 0 over 256 LDS entries made this kernel 15 us in front of every segment_matmul call)
  {
    const int lane = tid & 63, wv = tid >> 6, nw = nthr >> 6;
    int64_t v[4] = {tiles, tiles2, tiles3, rows};
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int lo = __shfl_up((int)(uint32_t)v[q], d), hi = __shfl_up((int)(uint32_t)((uint64_t)v[q] >> 32), d);
        if (lane >= d) v[q] += (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
      }
    }
    if (lane == 63) {  // wave totals (inclusive value of the last lane)
      s_tiles[wv] = v[0];
      s_tiles2[wv] = v[1];
      s_tiles3[wv] = v[2];
      s_rows[wv] = v[3];
    }
    __syncthreads();
    int64_t base[4] = {0, 0, 0, 0}, all[4] = {0, 0, 0, 0};
    for (int i = 0; i < nw; ++i) {
      const int64_t w4[4] = {s_tiles[i], s_tiles2[i], s_tiles3[i], s_rows[i]};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (i < wv) base[q] += w4[q];
        all[q] += w4[q];
      }
    }
    __syncthreads();
    // exclusive prefix of this thread = waves in front + inclusive value - own sum
    s_tiles[tid] = base[0] + v[0] - tiles;
    s_tiles2[tid] = base[1] + v[1] - tiles2;
    s_tiles3[tid] = base[2] + v[2] - tiles3;
    s_rows[tid] = base[3] + v[3] - rows;
    if (tid == 0) {
      tile_start[B] = (int32_t)all[0];
      tile_start2[B] = (int32_t)all[1];
      tile_start3[B] = (int32_t)all[2];
      row_start[B] = all[3];
    }
  }
  int64_t t = s_tiles[tid], t2 = s_tiles2[tid], t3 = s_tiles3[tid], rs = s_rows[tid];
  for (int64_t b = beg; b < end; ++b) {
    const int64_t p0 = ptr[b];
    int64_t r = ptr[b + 1] - p0;
    if (r < 0) r = 0;
    DevGroup d;
    d.a = a + p0 * K * elt;
    d.w = w + b * K * M * elt;
    d.c = c + p0 * M * elt;
    d.bias = bias ? bias + b * M * elt : nullptr;
    d.rows = r;
    d.k = (int32_t)K;
    d.m = (int32_t)M;
    d.trans = 0;
    d.pad = gen_class(d.a, d.w, d.c, K, M, elt, 0);
    descs[b] = d;
    tile_start[b] = (int32_t)t;
    tile_start2[b] = (int32_t)t2;
    tile_start3[b] = (int32_t)t3;
    row_start[b] = rs;
    t += (r + kTileRows - 1) / kTileRows;
    t2 += (r + 2 * kTileRows - 1) / (2 * kTileRows);
    t3 += (r + kPairRows - 1) / kPairRows;
    rs += r;
  }
}

// ---- MFMA kernels ------------------------------------------------------------------------------
// Store 16 consecutive output elements (fp32 accumulators -> T) at `dst` (16-byte aligned).
__device__ __forceinline__ void store16(bf16_t*, char* dst, const float (&v)[16]) {
  u32x4 lo, hi;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint16_t a = __builtin_bit_cast(uint16_t, (__bf16)v[2 * i]);
    uint16_t b = __builtin_bit_cast(uint16_t, (__bf16)v[2 * i + 1]);
    lo[i] = (uint32_t)a | ((uint32_t)b << 16);
    uint16_t c = __builtin_bit_cast(uint16_t, (__bf16)v[8 + 2 * i]);
    uint16_t d = __builtin_bit_cast(uint16_t, (__bf16)v[8 + 2 * i + 1]);
    hi[i] = (uint32_t)c | ((uint32_t)d << 16);
  }
  reinterpret_cast<u32x4*>(dst)[0] = lo;
  reinterpret_cast<u32x4*>(dst)[1] = hi;
}
__device__ __forceinline__ void store16(f16_t*, char* dst, const float (&v)[16]) {
  u32x4 lo, hi;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint16_t a = __builtin_bit_cast(uint16_t, (_Float16)v[2 * i]);
    uint16_t b = __builtin_bit_cast(uint16_t, (_Float16)v[2 * i + 1]);
    lo[i] = (uint32_t)a | ((uint32_t)b << 16);
    uint16_t c = __builtin_bit_cast(uint16_t, (_Float16)v[8 + 2 * i]);
    uint16_t d = __builtin_bit_cast(uint16_t, (_Float16)v[8 + 2 * i + 1]);
    hi[i] = (uint32_t)c | ((uint32_t)d << 16);
  }
  reinterpret_cast<u32x4*>(dst)[0] = lo;
  reinterpret_cast<u32x4*>(dst)[1] = hi;
}
__device__ __forceinline__ void store16(float*, char* dst, const float (&v)[16]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4 q = {v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
    reinterpret_cast<f32x4*>(dst)[i] = q;
  }
}

// K: contraction length (compile time), MC: output columns per workgroup pass (grid.y walks
// M / MC column chunks), NW: waves per workgroup (tile = NW * 32 rows).
template <typename T, int K, int MC, int NW>
__global__ __launch_bounds__(NW * 64) void mfma_rows_kernel(const DevGroup* __restrict__ descs,
                                                            const int32_t* __restrict__ tile_start,
                                                            int B, int ncol) {
  constexpr int SZ = Elem<T>::kSize;
  constexpr int EPC = Elem<T>::kPerChunk;
  constexpr int NCH = (K / 2) / EPC;           // 16-byte chunks per lane (half row)
  constexpr int NT = MC / 32;                  // 32-column MFMA tiles per wave
  constexpr int LDW = K * SZ + 16;             // LDS row stride (bytes) of the W^T image
  constexpr int BM = NW * 32;
  static_assert(BM == kTileRows, "tile table is built for 128-row tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int x = lane & 31;
  const int h = lane >> 5;
  const int bx = ncol > 1 ? ((int)blockIdx.x / (8 * ncol)) * 8 + ((int)blockIdx.x & 7) : (int)blockIdx.x;
  const int by = ncol > 1 ? ((int)blockIdx.x / 8) % ncol : 0;
  const int col0 = by * MC;

  const int total = tile_start[B];
  const int G = (int)gridDim.x / ncol;
  const int t0 = (int)((int64_t)bx * total / G);
  const int t1 = (int)((int64_t)(bx + 1) * total / G);
  if (t0 >= t1) return;

  // group of the first tile: largest g with tile_start[g] <= t0
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_start[mid] <= t0) lo = mid; else hi = mid;
  }
  int g = lo;
  int staged = -1;

  // LDS row (output column within the chunk) whose fragment this lane reads for tile t:
  // c(t, x) = (MC/2)*bit2(x) + 16 t + 4*(x>>3) + (x&3)   (see header comment)
  const int crow0 = (MC / 2) * ((x >> 2) & 1) + 4 * (x >> 3) + (x & 3);
  const char* wfrag = smem + crow0 * LDW + (K / 2) * h * SZ;

  DevGroup d = descs[g];
  for (int t = t0; t < t1; ++t) {
    while (t >= tile_start[g + 1]) {
      ++g;
      d = descs[g];
    }
    if (g != staged) {
      __syncthreads();  // every wave is done reading the previous relation's weight
      const char* w = d.w;
      const int M = d.m;
      if (!d.trans) {
        // W is [K][M]: read 16-byte pieces along M, scatter transposed into the [MC][K] image.
        constexpr int CPR = MC / EPC;  // chunks per W row (within the column chunk)
        for (int idx = tid; idx < K * CPR; idx += NW * 64) {
          const int k = idx / CPR;
          const int cc = (idx - k * CPR) * EPC;
          const u32x4 v = *reinterpret_cast<const u32x4*>(w + ((int64_t)k * M + col0 + cc) * SZ);
          if constexpr (SZ == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const uint16_t s = (uint16_t)(v[e >> 1] >> ((e & 1) * 16));
              *reinterpret_cast<uint16_t*>(smem + (cc + e) * LDW + k * 2) = s;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              *reinterpret_cast<uint32_t*>(smem + (cc + e) * LDW + k * 4) = v[e];
          }
        }
      } else {
        // W is stored [M][K] (transposed view): straight 16-byte copies.
        constexpr int CPR = K / EPC;
        for (int idx = tid; idx < MC * CPR; idx += NW * 64) {
          const int c = idx / CPR;
          const int kk = (idx - c * CPR) * EPC;
          const u32x4 v =
              *reinterpret_cast<const u32x4*>(w + ((int64_t)(col0 + c) * K + kk) * SZ);
          *reinterpret_cast<u32x4*>(smem + c * LDW + kk * SZ) = v;
        }
      }
      __syncthreads();
      staged = g;
    }

    const int64_t rows = d.rows;
    const int64_t row_base = (int64_t)(t - tile_start[g]) * BM + wave * 32;
    if (row_base >= rows) continue;  // wave-uniform: ragged last tile of a segment
    const int64_t row = row_base + x;
    const bool valid = row < rows;
    const int64_t lrow = valid ? row : rows - 1;

    // ---- X: lane's contiguous half row, HBM -> VGPR ----
    u32x4 xv[NCH];
    const u32x4* xp =
        reinterpret_cast<const u32x4*>(d.a + (lrow * K + (K / 2) * h) * SZ);
#pragma unroll
    for (int i = 0; i < NCH; ++i) xv[i] = __builtin_nontemporal_load(xp + i);

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

#pragma unroll
    for (int s = 0; s < NCH; ++s) {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const u32x4 wv = *reinterpret_cast<const u32x4*>(wfrag + tt * 16 * LDW + s * 16);
        acc[tt] = mfma_chunk(T{}, wv, xv[s], acc[tt]);
      }
    }

    // ---- epilogue: lane (x, h) owns out[row][col0 + h*MC/2 + 16 tt + r] ----
    if (valid) {
      const int M = d.m;
      char* op = d.c + (row * M + col0 + (MC / 2) * h) * SZ;
      const T* bp = d.bias ? reinterpret_cast<const T*>(d.bias) + col0 + (MC / 2) * h : nullptr;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[tt][r];
        if (bp) {
          // reference semantics (pyg_lib/ops/__init__.py:169-171): `out` is materialised in T
          // first, then `out += bias` -- so round the product before adding for 16-bit types.
#pragma unroll
          for (int r = 0; r < 16; ++r)
            v[r] = round_to(T{}, v[r]) + load_bias(bp + 16 * tt + r);
        }
        store16((T*)nullptr, op + tt * 16 * SZ, v);
      }
    }
  }
}

// ---- v2: LDS-staged streaming kernel for 16-bit types (the HBM-bound configs) ---------------------
// Same tile walk and MFMA mapping as mfma_rows_kernel, but X and the output move between HBM and
// registers as fully coalesced 1 KiB wave accesses (every instruction covers whole 256-byte rows)
// and are re-shaped into / out of MFMA fragment order through a per-wave LDS stage with a 16-byte
// XOR swizzle (conflict-free ds_read_b128 / ds_write_b128).  The next tile's rows are prefetched
// into registers while the current tile is multiplied (issue-early / write-late).
__device__ __forceinline__ u32x4 pack8(bf16_t, const float* v) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint16_t a = __builtin_bit_cast(uint16_t, (__bf16)v[2 * i]);
    uint16_t b = __builtin_bit_cast(uint16_t, (__bf16)v[2 * i + 1]);
    o[i] = (uint32_t)a | ((uint32_t)b << 16);
  }
  return o;
}
__device__ __forceinline__ u32x4 pack8(f16_t, const float* v) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint16_t a = __builtin_bit_cast(uint16_t, (_Float16)v[2 * i]);
    uint16_t b = __builtin_bit_cast(uint16_t, (_Float16)v[2 * i + 1]);
    o[i] = (uint32_t)a | ((uint32_t)b << 16);
  }
  return o;
}

__device__ __forceinline__ u32x4 pack_chunk(bf16_t t, const float* v) { return pack8(t, v); }
__device__ __forceinline__ u32x4 pack_chunk(f16_t t, const float* v) { return pack8(t, v); }
__device__ __forceinline__ u32x4 pack_chunk(float, const float* v) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = __builtin_bit_cast(uint32_t, v[i]);
  return o;
}

template <typename T, int K, int MC, int NW, int FLAGS = 3>
__global__ __launch_bounds__(NW * 64) void mfma_rows_lds_kernel(
    const DevGroup* __restrict__ descs, const int32_t* __restrict__ tile_start, int B, int chunk, int ncol) {
  constexpr bool NT_LOAD = (FLAGS & 1) != 0;
  constexpr bool NT_STORE = (FLAGS & 2) != 0;
  // FLAGS bit 2 (fp32 only): split-bf16 arithmetic -- x = hi + mid + lo with 8 significant bits each (24 in all), W alike;
  // (round to nearest at every split, so |mid| <= 2^-9 |x|, |lo| <= 2^-18 |x| and the residual left is <= 2^-27 |x|);
  // the six products of weight 2^0, 2^-9, 2^-9, 2^-18, 2^-18, 2^-18 go through v_mfma_f32_32x32x16_bf16 with fp32
  // accumulation (the two of weight 2^-27 and the one of 2^-36 are dropped: 1.5e-8 of |x||w| per product, unbiased --
  // a quarter of the fp32 rounding unit, so the result is as close to the exact product as the fp32 MFMA's).
  // Six 32-cycle MFMAs per 16 k instead of eight 64-cycle v_mfma_f32_32x32x2_f32: 2.7x less matrix time, which makes
  // fp32 F = 128 (AI = 32 flop/B) HBM-bound instead of bound by the fp32 matrix rate.  W^T lives in LDS as three
  // bf16 planes [MC][K] (16-byte chunks XOR-swizzled with the row, no pad: 3 x 32 KB + 4 x 16 KB of stages = 160 KB).
  constexpr bool X3 = (FLAGS & 4) != 0;
  static_assert(!X3 || (std::is_same<T, float>::value && K == 128 && MC == 128), "split-bf16: fp32, K = MC = 128");
  constexpr int SZ = Elem<T>::kSize;
  constexpr int EPC = Elem<T>::kPerChunk;  // elements per 16-byte chunk
  constexpr int NT = MC / 32;
  constexpr int LDW = K * SZ + 16;
  constexpr int BM = NW * 32;
  static_assert(BM == kTileRows, "tile table is built for 128-row tiles");
  constexpr int CPR = K * SZ / 16;              // 16-byte chunks per X row
  constexpr int NI = CPR / 2;                   // coalesced wave loads per 32-row tile
  constexpr int XM = (CPR < 16 ? CPR : 16) - 1; // swizzle mask
  constexpr int CPO = MC * SZ / 16;             // chunks per output row (this column chunk)
  constexpr int NO = CPO / 2;
  constexpr int OM = (CPO < 16 ? CPO : 16) - 1;
  constexpr int STAGE = 32 * 16 * (CPR > CPO ? CPR : CPO);  // bytes per wave
  constexpr int PLANE = MC * K * 2;  // X3: one bf16 plane of W^T
  constexpr int WBYTES = X3 ? 3 * PLANE : MC * LDW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int x = lane & 31;
  const int h = lane >> 5;
  // XCD-aware workgroup decode (1-D grid of G * ncol workgroups): consecutive workgroup ids go to
  // consecutive XCDs, so the `ncol` column-chunk workgroups of one tile range get ids 8 apart -- same
  // XCD, same L2 -- and the X tiles they both read come from HBM once.
  const int bx = ncol > 1 ? ((int)blockIdx.x / (8 * ncol)) * 8 + ((int)blockIdx.x & 7) : (int)blockIdx.x;
  const int by = ncol > 1 ? ((int)blockIdx.x / 8) % ncol : 0;
  const int col0 = by * MC;
  char* stage = smem + WBYTES + wave * STAGE;

  // Tile schedule: workgroup b owns the tile chunks b, b + G, b + 2G, ... of `chunk` consecutive
  // tiles each (chunk <= 0: one contiguous range per workgroup).  Local tile i of this workgroup is
  // global tile tile_of(i).
  const int total = tile_start[B];
  const int G = (int)gridDim.x / ncol;
  int nloc, cbase = 0;
  if (chunk <= 0) {
    cbase = (int)((int64_t)bx * total / G);
    nloc = (int)((int64_t)(bx + 1) * total / G) - cbase;
  } else {
    const int nchunks = (total + chunk - 1) / chunk;
    const int mine = nchunks > bx ? (nchunks - 1 - bx) / G + 1 : 0;
    nloc = mine * chunk;
    if (mine > 0) {
      const int last_chunk = (mine - 1) * G + bx;  // may be the ragged final chunk
      const int over = (last_chunk + 1) * chunk - total;
      if (over > 0) nloc -= over;
    }
  }
  if (nloc <= 0) return;
  auto tile_of = [&](int i) -> int {
    if (chunk <= 0) return cbase + i;
    const int j = i / chunk;
    return (j * G + bx) * chunk + (i - j * chunk);
  };
  const int t0 = 0, t1 = nloc;  // local tile indices

  int lo = 0, hi = B;
  {
    const int first = tile_of(0);
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tile_start[mid] <= first) lo = mid; else hi = mid;
    }
  }
  int g = lo;       // group of the tile being prefetched
  int staged = -1;  // group whose weight is in LDS

  const int crow0 = (MC / 2) * ((x >> 2) & 1) + 4 * (x >> 3) + (x & 3);
  const char* wfrag = X3 ? smem + crow0 * (K * 2) : smem + crow0 * LDW + (K / 2) * h * SZ;

  // per-lane constants of the coalesced <-> fragment re-shaping
  // load/store side: position p = i*64 + lane -> row r = p / CPR, slot c' = p % CPR
  // fragment side:   lane (x, h) reads row x, chunk c at slot c ^ (x & XM)
  u32x4 xr[NI];
  uint32_t xoff[8];  // X3: byte offset of this lane's 16 bytes of load i (< 8) inside a whole 32-row tile
  if constexpr (X3) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = i * 64 + lane;
      const int r = p / CPR;
      xoff[i] = (uint32_t)(r * (K * SZ) + (((p % CPR) ^ (r & XM)) * 16));
    }
  }
  DevGroup dn = descs[g];
  int64_t n_row0 = 0, n_rows = 0;
  bool n_valid = false;

  auto prefetch = [&](int ti) {
    const int t = tile_of(ti);
    while (t >= tile_start[g + 1]) {
      ++g;
      dn = descs[g];
    }
    n_rows = dn.rows;
    n_row0 = (int64_t)(t - tile_start[g]) * BM + wave * 32;
    n_valid = n_row0 < n_rows;
    if (X3 && n_valid && n_row0 + 32 <= n_rows) {
      // whole tile: tile base + per-lane offsets computed once (loads i and i + 8 lie 16 rows = 8 KiB apart)
      const char* base = dn.a + n_row0 * (K * SZ);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        typedef __attribute__((address_space(1))) u32x4 GU32x4;
        const GU32x4* src = (const GU32x4*)(base + xoff[i & 7] + (i >> 3) * 8192);
        if (ncol == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=a"(xr[i]) : "v"(src) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(xr[i]) : "v"(src) : "memory");
      }
    } else if (n_valid) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int p = i * 64 + lane;
        const int r = p / CPR;
        const int cs = p % CPR;
        const int c = cs ^ (r & XM);
        int64_t row = n_row0 + r;
        if (row >= n_rows) row = n_rows - 1;
        // global_* (a flat access also counts on lgkmcnt and makes every LDS wait conservative)
        typedef __attribute__((address_space(1))) u32x4 GU32x4;
        const GU32x4* src = (const GU32x4*)(dn.a + row * (K * SZ) + c * 16);
        // with several column-chunk readers the tile must STAY in L2 for the others: no streaming hint (with it
        // C4's X came from HBM 1.86 times, PMC FETCH_SIZE; without it 1.01 times)
        if constexpr (X3) {
          // through inline asm: the wait is placed by hand (x3_wait) -- the compiler cannot count the stores that were
          // issued after these loads across the loop's branches and would wait for them too (vmcnt retires in order)
          // (into AGPRs: the one wave per SIMD has 192 of them idle, and a value the compiler believes defined must
          // not be moved before its load has landed -- under VGPR pressure it would be)
          if (ncol == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=a"(xr[i]) : "v"(src) : "memory");
          else asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(xr[i]) : "v"(src) : "memory");
        } else {
          xr[i] = (NT_LOAD && ncol == 1) ? __builtin_nontemporal_load(src) : *src;
        }
      }
    }
  };

  // Software pipeline (per wave; the stage buffer is private to the wave):
  //   loop top: X_t is in the LDS stage, L_{t+1} (next tile's rows) is in flight into xr.
  //   1. multiply X_t by W (MFMA, fragments double-buffered in registers)
  //   2. pack the result, swizzle it through the stage, read it back in row order (ov)
  //   3. wait for L_{t+1} (issued a whole tile ago, as were the stores S_{t-1} ahead of it in the
  //      in-order memory queue), write it to the stage
  //   4. issue L_{t+2}, then the global stores S_t
  // so no wave ever waits on a store it has just issued.
  prefetch(t0);
  DevGroup d = dn;
  int cg = g;
  int64_t row0 = n_row0, rows = n_rows;
  bool valid = n_valid;
  // X3: `stores_younger` = exactly the NO unpredicated stores of a whole tile were issued after the loads now awaited
  bool stores_younger = false;
  auto x3_wait = [&]() {
    if constexpr (X3) {
      static_assert(!X3 || NO == 16, "the hand-placed wait counts the 16 stores of a 32 x 128 fp32 tile");
      if (stores_younger && !(FLAGS & 16)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  x3_wait();
  if (valid) {
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(stage + (i * 64 + lane) * 16) = xr[i];
  }
  if (t0 + 1 < t1) prefetch(t0 + 1);

  constexpr bool TIMED = (FLAGS & 32) != 0;  // experiment builds: per-phase cycle sums of workgroup 0 / wave 0
  uint64_t ph[6] = {0, 0, 0, 0, 0, 0}, stamp = 0;
  auto tick = [&](int i) {
    if constexpr (TIMED) {
      const uint64_t now = __builtin_amdgcn_s_memtime();
      ph[i] += now - stamp;
      stamp = now;
    }
  };
  if constexpr (TIMED) stamp = __builtin_amdgcn_s_memtime();
  for (int t = t0; t < t1; ++t) {
    if (cg != staged) {
      __syncthreads();
      const char* w = d.w;
      const int M = d.m;
      if constexpr (X3) {
        // fp32 W[k][m] (or W^T[m][k]) -> three bf16 planes [m][k]: element (m, k) at plane + m * 256 + (((k >> 3) ^ (m & 15)) * 16)
        // + (k & 7) * 2
        constexpr int CW = 32;  // 16-byte chunks per source row (K = MC = 128 floats)
        for (int idx = tid; idx < 128 * CW; idx += NW * 64) {
          const int r = idx / CW;
          const int c4 = (idx - r * CW) * 4;
          const int64_t src = !d.trans ? ((int64_t)r * M + col0 + c4) : ((int64_t)(col0 + r) * K + c4);
          // (whole-vector bit_cast: __builtin_bit_cast(float, v[e]) on a vector element reads element 0, clang 19)
          const f32x4 v = __builtin_bit_cast(f32x4, *reinterpret_cast<const u32x4*>(w + src * 4));
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            float f0 = v[e], f1 = v[e + 1];
            const uint32_t ph = split2(f0, f1);
            const uint32_t pm = split2(f0, f1);
            const uint32_t pl = split2(f0, f1);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int k = !d.trans ? r : c4 + e + u;
              const int mm = !d.trans ? c4 + e + u : r;
              char* dst = smem + mm * (K * 2) + (((k >> 3) ^ (mm & 15)) * 16) + (k & 7) * 2;
              *reinterpret_cast<uint16_t*>(dst) = (uint16_t)(ph >> (16 * u));
              *reinterpret_cast<uint16_t*>(dst + PLANE) = (uint16_t)(pm >> (16 * u));
              *reinterpret_cast<uint16_t*>(dst + 2 * PLANE) = (uint16_t)(pl >> (16 * u));
            }
          }
        }
      } else if (!d.trans) {
        constexpr int CW = MC / EPC;
        for (int idx = tid; idx < K * CW; idx += NW * 64) {
          const int k = idx / CW;
          const int cc = (idx - k * CW) * EPC;
          const u32x4 v = *reinterpret_cast<const u32x4*>(w + ((int64_t)k * M + col0 + cc) * SZ);
          if constexpr (SZ == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const uint16_t sv = (uint16_t)(v[e >> 1] >> ((e & 1) * 16));
              *reinterpret_cast<uint16_t*>(smem + (cc + e) * LDW + k * 2) = sv;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t sv = v[e];
              *reinterpret_cast<uint32_t*>(smem + (cc + e) * LDW + k * 4) = sv;
            }
          }
        }
      } else {
        constexpr int CW = K / EPC;
        for (int idx = tid; idx < MC * CW; idx += NW * 64) {
          const int c = idx / CW;
          const int kk = (idx - c * CW) * EPC;
          const u32x4 v =
              *reinterpret_cast<const u32x4*>(w + ((int64_t)(col0 + c) * K + kk) * SZ);
          *reinterpret_cast<u32x4*>(smem + c * LDW + kk * SZ) = v;
        }
      }
      __syncthreads();
      staged = cg;
    }

    tick(0);  // bookkeeping + W staging
    u32x4 ov[NO];
    if (valid) {
      f32x16 acc[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

      if constexpr (X3) {
        // 16 units of 12 MFMAs: unit u = (K-step s = u >> 1, column blocks 2 (u & 1), +1).  The W fragments of unit u + 1
        // and (even units) the X chunks of step s + 1 are read from LDS while unit u's MFMAs run; odd units also split
        // those X chunks.  One wave per SIMD: nothing else hides an LDS round trip.
        const int wsw = crow0 & 15;  // rows crow0 + 16 tt share it
        const char* xrow = stage + x * (CPR * 16);
        const int xs = x & XM;
        u32x4 wq[2][6], qx[2], xf[2][3];
        auto read_w = [&](int u, u32x4 (&wv)[6]) {
          const int s8 = u >> 1;
          const char* wr = wfrag + (2 * (u & 1)) * 16 * (K * 2) + (((8 * h + s8) ^ wsw) * 16);
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wv[3 * j + pl] = *reinterpret_cast<const u32x4*>(wr + j * 16 * (K * 2) + pl * PLANE);
        };
        auto read_x = [&](int s8) {
          // lane (x, h): k = 64 h + 8 s8 + e, e = 0 ... 7: two 16-byte fp32 chunks of its half row
          const int c = NI * h + 2 * s8;
          qx[0] = *reinterpret_cast<const u32x4*>(xrow + ((c ^ xs) * 16));
          qx[1] = *reinterpret_cast<const u32x4*>(xrow + (((c + 1) ^ xs) * 16));
        };
        auto split_x = [&](u32x4 (&o)[3]) {
          const f32x4 f0 = __builtin_bit_cast(f32x4, qx[0]), f1 = __builtin_bit_cast(f32x4, qx[1]);
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            float a = p < 2 ? f0[2 * p] : f1[2 * p - 4];
            float b2 = p < 2 ? f0[2 * p + 1] : f1[2 * p - 3];
            o[0][p] = split2(a, b2);
            o[1][p] = split2(a, b2);
            o[2][p] = split2(a, b2);
          }
        };
        read_x(0);
        read_w(0, wq[0]);
        split_x(xf[0]);
#pragma unroll
        for (int u = 0; u < ((FLAGS & 8) ? 1 : 16); ++u) {
          const int s8 = u >> 1;
          // this unit's fragments were issued a unit ago: wait for them here, not (with the reads below) at the MFMAs
          asm volatile("" : "+v"(wq[u & 1][5]));
          __builtin_amdgcn_sched_barrier(0);
          if (u + 1 < 16) read_w(u + 1, wq[(u + 1) & 1]);
          if ((u & 1) == 0 && s8 + 1 < 8) read_x(s8 + 1);
          __builtin_amdgcn_sched_barrier(0);
          const u32x4(&xv)[3] = xf[s8 & 1];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int tt = 2 * (u & 1) + j;
            const u32x4 wh = wq[u & 1][3 * j], wm = wq[u & 1][3 * j + 1], wl = wq[u & 1][3 * j + 2];
            // smallest terms first
            acc[tt] = mfma_chunk(bf16_t{}, wl, xv[0], acc[tt]);
            acc[tt] = mfma_chunk(bf16_t{}, wh, xv[2], acc[tt]);
            acc[tt] = mfma_chunk(bf16_t{}, wm, xv[1], acc[tt]);
            acc[tt] = mfma_chunk(bf16_t{}, wm, xv[0], acc[tt]);
            acc[tt] = mfma_chunk(bf16_t{}, wh, xv[1], acc[tt]);
            acc[tt] = mfma_chunk(bf16_t{}, wh, xv[0], acc[tt]);
          }
          if ((u & 1) == 1 && s8 + 1 < 8) split_x(xf[(s8 + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
      // fragments of step s+1 are read from LDS while the MFMAs of step s run
      u32x4 xa = *reinterpret_cast<const u32x4*>(stage + (x * CPR + ((NI * h) ^ (x & XM))) * 16);
      u32x4 wa[NT];
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) wa[tt] = *reinterpret_cast<const u32x4*>(wfrag + tt * 16 * LDW);
#pragma unroll
      for (int s = 0; s < NI; ++s) {
        // wait for this step's fragments (issued a whole step ago) before the next step's reads go out:
        // otherwise the compiler's wait in front of the MFMAs is an lgkmcnt(0) that covers those too
        asm volatile("" : "+v"(wa[NT - 1]));
        __builtin_amdgcn_sched_barrier(0);
        u32x4 xb = xa;
        u32x4 wb[NT];
        if (s + 1 < NI) {
          const int c = NI * h + s + 1;
          xb = *reinterpret_cast<const u32x4*>(stage + (x * CPR + (c ^ (x & XM))) * 16);
#pragma unroll
          for (int tt = 0; tt < NT; ++tt)
            wb[tt] = *reinterpret_cast<const u32x4*>(wfrag + tt * 16 * LDW + (s + 1) * 16);
        }
        // keep the order "issue the next step's LDS reads, then this step's MFMAs": left alone, the
        // scheduler sinks every ds_read to just before its MFMA (lgkmcnt(0) x32 per tile)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SZ == 4) {
          // fp32: a 16-byte chunk feeds four 32x32x2 MFMAs per column block; interleave the column blocks
          // so that back-to-back MFMAs never wait on each other's accumulator
          const f32x4 xf = __builtin_bit_cast(f32x4, xa);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
              acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(f32x4, wa[tt])[e], xf[e], acc[tt], 0, 0, 0);
        } else {
#pragma unroll
          for (int tt = 0; tt < NT; ++tt) acc[tt] = mfma_chunk(T{}, wa[tt], xa, acc[tt]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < NI) {
          xa = xb;
#pragma unroll
          for (int tt = 0; tt < NT; ++tt) wa[tt] = wb[tt];
        }
      }

      }  // !X3
      tick(1);  // K loop
      // epilogue: fragment order -> swizzled stage -> row order (ov)
      const T* bp = d.bias ? reinterpret_cast<const T*>(d.bias) + col0 + (MC / 2) * h : nullptr;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[tt][r];
        if (bp) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = round_to(T{}, v[r]) + load_bias(bp + 16 * tt + r);
        }
#pragma unroll
        for (int j = 0; j < SZ; ++j) {  // 16 values = SZ chunks of EPC elements
          const int c = NO * h + SZ * tt + j;
          *reinterpret_cast<u32x4*>(stage + (x * CPO + (c ^ (x & OM))) * 16) = pack_chunk(T{}, v + EPC * j);
        }
      }
#pragma unroll
      for (int i = 0; i < NO; ++i) ov[i] = *reinterpret_cast<const u32x4*>(stage + (i * 64 + lane) * 16);
    }

    tick(2);  // epilogue through LDS
    // stage the next tile (its loads were issued one tile ago) and issue the loads after it
    const DevGroup d_out = d;
    const int64_t row0_out = row0, rows_out = rows;
    const bool valid_out = valid;
    if (t + 1 < t1) {
      d = dn;
      cg = g;
      row0 = n_row0;
      rows = n_rows;
      valid = n_valid;
      x3_wait();
      tick(3);  // waiting for the X loads
      if (valid) {
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(stage + (i * 64 + lane) * 16) = xr[i];
      }
      if (t + 2 < t1) prefetch(t + 2);
    }
    tick(4);  // staging X + issuing the next loads
    stores_younger = false;

    if (X3 && valid_out && row0_out + 32 <= rows_out) {
      // whole tile: NO unpredicated stores, which the next x3_wait leaves in flight
      const int M = d_out.m;
      char* obase = d_out.c + (row0_out * M + col0) * SZ;
#pragma unroll
      for (int i = 0; i < NO; ++i) {
        const int p = i * 64 + lane;
        const int r = p / CPO;
        const int c = (p % CPO) ^ (r & OM);
        typedef __attribute__((address_space(1))) u32x4 GU32x4;
        GU32x4* dst = (GU32x4*)(obase + (int64_t)r * M * SZ + c * 16);
        if ((FLAGS & 16) && ov[i][0] != 0x12345u) continue;
        if (NT_STORE) __builtin_nontemporal_store(ov[i], dst); else *dst = ov[i];
      }
      stores_younger = true;
    } else if (valid_out) {
      const int M = d_out.m;
      char* obase = d_out.c + (row0_out * M + col0) * SZ;
#pragma unroll
      for (int i = 0; i < NO; ++i) {
        const int p = i * 64 + lane;
        const int r = p / CPO;
        const int cs = p % CPO;
        const int c = cs ^ (r & OM);
        if (row0_out + r < rows_out && (!(FLAGS & 16) || ov[i][0] == 0x12345u)) {
          typedef __attribute__((address_space(1))) u32x4 GU32x4;
          GU32x4* dst = (GU32x4*)(obase + (int64_t)r * M * SZ + c * 16);
          if (NT_STORE) __builtin_nontemporal_store(ov[i], dst); else *dst = ov[i];
        }
      }
    }
    tick(5);  // stores
  }
  if constexpr (TIMED) {
    if (blockIdx.x == 0 && tid == 0)
      printf("x3 phases (cycles, %d tiles): book %llu kloop %llu epilogue %llu loadwait %llu stage+issue %llu stores %llu\n", nloc,
             (unsigned long long)ph[0], (unsigned long long)ph[1], (unsigned long long)ph[2], (unsigned long long)ph[3],
             (unsigned long long)ph[4], (unsigned long long)ph[5]);
  }
}


// ---- 16-bit, K = 128, 128 output columns: cyclic schedule, W by LDS-DMA in its native layout ---------------------
// The contiguous-range kernel above keeps ~1000 independent read / write streams alive (one per wave); how fast the
// HBM side serves that depends on where the caching allocator happened to place `input` and `out` (measured on one
// box, same launch, six candidate output buffers: 5.0 ... 6.1 TB/s, ~3 of 4 allocations at the low end).  Here every
// workgroup (8 waves, 256-row tile) takes every G/8-th tile of its XCD's band, so the chip sweeps eight narrow windows
// of `input` / `out` front to back (5.4 - 6.3 TB/s on the same buffers).  A workgroup then changes relation every other
// tile, so the weight switch must be free:
//   * W[g] is copied [K][M] as it lies in memory by LDS-DMA (global_load_lds_dwordx4, 8 waves x 4 KiB); the 16-byte
//     chunks of every 1 KiB block (4 k-rows) are permuted on the SOURCE side so that
//   * the MFMA "A" fragments (8 consecutive k of one output column) come out of gfx950's transposing LDS read
//     (ds_read_b64_tr_b16, two per fragment) without bank conflicts: a 32-lane service group touches
//     4 k-rows x {chunks 2tt, 2tt+1, 8+2tt, 9+2tt}, which the permutation places in one 256-byte line;
//   * two W buffers: the next relation of this workgroup's tile sequence is in flight while the current one is
//     multiplied; ONE workgroup barrier per relation change (everybody is done with the buffer that is refilled next,
//     and everybody's part of the new W has landed -- each wave has waited for its own DMAs because they are older
//     than the X tile it has just staged).
// X staging, fragment order, epilogue and store order are those of mfma_rows_lds_kernel.
template <typename T, int FLAGS = 3>
__global__ __launch_bounds__(512) void mfma_rows_cyc_kernel(const DevGroup* __restrict__ descs,
                                                            const int32_t* __restrict__ tile_start, int B) {
  constexpr bool NT_LOAD = (FLAGS & 1) != 0;
  constexpr bool NT_STORE = (FLAGS & 2) != 0;
  constexpr int K = 128, MC = 128, SZ = 2, NWV = 8;
  constexpr int NT = 4, NI = 8, NO = 8, CPR = 16;
  constexpr int BM = NWV * 32;
  static_assert(BM == 2 * kTileRows, "tile_start2 is built for 256-row tiles");
  constexpr int WB = K * MC * SZ;  // 32 KB per W buffer
  constexpr int BLK_PER_WAVE = (K / 4) / NWV;
  typedef __attribute__((address_space(3))) void LDSV;
  typedef short v4i16 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) u32x4 GU32x4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 31, h = lane >> 5;
  const int bx = blockIdx.x, G = gridDim.x;
  char* stage = smem + 2 * WB + wave * 8192;
  // Banded cyclic schedule: the tiles are cut into 8 contiguous bands, the workgroups of XCD k (ids k, k + 8, ...:
  // consecutive ids go to consecutive XCDs) sweep band k cyclically -- 8 narrow windows instead of one, every page and
  // L2 line is touched by ONE XCD.  Same rate as the single sweep on unfavourably placed buffers (5.4 TB/s), 6.3
  // instead of 5.8 TB/s on favourable ones (tools/lab: v3:sched=2).
  const int total = tile_start[B];
  const int nb = (G & 7) == 0 ? 8 : 1;
  const int band = bx % nb, per = G / nb;
  const int band0 = (int)((int64_t)band * total / nb), band1 = (int)((int64_t)(band + 1) * total / nb);
  const int cbase = band0 + bx / nb;  // first tile of this workgroup; then every `per`-th tile of the band
  if (cbase >= band1) return;
  const int nloc = (band1 - 1 - cbase) / per + 1;

  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_start[mid] <= cbase) lo = mid; else hi = mid;
  }
  int g = lo;  // group of the tile being prefetched

  // DMA side: lane i of a block's instruction fills LDS position i (16 bytes) of the 1 KiB block
  const int dma_r = (lane & 15) >> 2, dma_ii = lane & 3, dma_u = lane >> 4;
  const int dma_c = 2 * dma_u + (dma_ii & 1) + 8 * (dma_ii >> 1);
  const int dma_src_off = dma_r * (MC * SZ) + dma_c * 16;
  auto issue_w = [&](int grp_id, int buf) {
    const char* w = descs[grp_id].w;
#pragma unroll
    for (int j = 0; j < BLK_PER_WAVE; ++j) {
      const int kb = wave * BLK_PER_WAVE + j;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + kb * 1024 + dma_src_off),
                                       (LDSV*)(smem + buf * WB + kb * 1024), 16, 0, 0);
    }
  };
  // group of the first tile of this workgroup's sequence behind group `gc` (-1: none)
  auto next_group = [&](int gc) -> int {
    const int ts = tile_start[gc + 1];
    if (ts >= total) return -1;
    const int j = ts > cbase ? (ts - cbase + per - 1) / per : 0;
    if (j >= nloc) return -1;
    const int t = cbase + j * per;
    int gg = gc + 1;
    while (tile_start[gg + 1] <= t) ++gg;
    return gg;
  };
  // reader side (transposing read): lane q of a 16-lane group supplies k-row (q >> 2), piece (q & 3)
  const int q = lane & 15, grp16 = lane >> 4;
  const int a_lane_off = 16384 * h + (4 * (q >> 2) + (grp16 & 1) + 2 * (q & 1)) * 16 + ((q & 3) >> 1) * 8;

  u32x4 xr[NI];
  DevGroup dn = descs[g];
  int64_t n_row0 = 0, n_rows = 0;
  bool n_valid = false;
  auto prefetch = [&](int ti) {
    const int t = cbase + ti * per;
    while (t >= tile_start[g + 1]) {
      ++g;
      dn = descs[g];
    }
    n_rows = dn.rows;
    n_row0 = (int64_t)(t - tile_start[g]) * BM + wave * 32;
    n_valid = n_row0 < n_rows;
    if (n_valid) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int p = i * 64 + lane;
        const int r = p / CPR;
        const int cs = p % CPR;
        const int c = cs ^ (r & 15);
        int64_t row = n_row0 + r;
        if (row >= n_rows) row = n_rows - 1;
        const GU32x4* src = (const GU32x4*)(dn.a + row * (K * SZ) + c * 16);
        xr[i] = NT_LOAD ? __builtin_nontemporal_load(src) : *src;
      }
    }
  };

  int wcur = g, wbuf = 0;
  issue_w(wcur, 0);
  int wnext = next_group(wcur);
  if (wnext >= 0) issue_w(wnext, 1);
  prefetch(0);
  DevGroup d = dn;
  int cg = g;
  int64_t row0 = n_row0, rows = n_rows;
  bool valid = n_valid;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (valid) {
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(stage + (i * 64 + lane) * 16) = xr[i];
  }
  if (1 < nloc) prefetch(1);

  for (int t = 0; t < nloc; ++t) {
    u32x4 ov[NO];
    if (valid) {
      f32x16 acc[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      const char* wb = smem + wbuf * WB + a_lane_off;
#pragma unroll
      for (int s = 0; s < NI; ++s) {
        const u32x4 xa = *reinterpret_cast<const u32x4*>(stage + (x * CPR + ((NI * h + s) ^ (x & 15))) * 16);
        u32x4 wa[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const v4i16 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) v4i16*)(wb + (2 * s) * 1024 + tt * 256));
          const v4i16 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) v4i16*)(wb + (2 * s + 1) * 1024 + tt * 256));
          wa[tt] = __builtin_bit_cast(u32x4, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) acc[tt] = mfma_chunk(T{}, wa[tt], xa, acc[tt]);
      }
      const T* bp = d.bias ? reinterpret_cast<const T*>(d.bias) + (MC / 2) * h : nullptr;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[tt][r];
        if (bp) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = round_to(T{}, v[r]) + load_bias(bp + 16 * tt + r);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = NO * h + 2 * tt + j;
          *reinterpret_cast<u32x4*>(stage + (x * 16 + (c ^ (x & 15))) * 16) = pack8(T{}, v + 8 * j);
        }
      }
#pragma unroll
      for (int i = 0; i < NO; ++i) ov[i] = *reinterpret_cast<const u32x4*>(stage + (i * 64 + lane) * 16);
    }
    const DevGroup d_out = d;
    const int64_t row0_out = row0, rows_out = rows;
    const bool valid_out = valid;
    if (t + 1 < nloc) {
      d = dn;
      cg = g;
      row0 = n_row0;
      rows = n_rows;
      valid = n_valid;
      if (valid) {
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(stage + (i * 64 + lane) * 16) = xr[i];
      }
      if (cg != wcur) {
        // relation change (same tile index in every wave of the workgroup)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        wcur = cg;
        wbuf ^= 1;
        wnext = next_group(wcur);
        if (wnext >= 0) issue_w(wnext, wbuf ^ 1);
      }
      if (t + 2 < nloc) prefetch(t + 2);
    }
    if (valid_out) {
      char* obase = d_out.c + (row0_out * MC) * SZ;
#pragma unroll
      for (int i = 0; i < NO; ++i) {
        const int p = i * 64 + lane;
        const int r = p / 16;
        const int cs = p % 16;
        const int c = cs ^ (r & 15);
        if (row0_out + r < rows_out) {
          GU32x4* dst = (GU32x4*)(obase + (int64_t)r * MC * SZ + c * 16);
          if (NT_STORE) __builtin_nontemporal_store(ov[i], dst); else *dst = ov[i];
        }
      }
    }
  }
}

// ---- 16-bit, K = 128, 128 output columns: ticket schedule, W in registers ---------------------------------------
// What bounds the two kernels above is the WRITE side of HBM, and how well it is served depends on the order in which
// the chip touches `out` (tools/lab: write-only sweeps of the same buffer run at 5.3 - 6.9 TB/s depending on nothing
// but that order).  Measured on buffers the allocator placed unfavourably, three things matter, and they add up:
//   1. tiles are handed out IN ADDRESS ORDER, as a non-persistent grid would be dispatched, not pre-assigned: a
//      workgroup draws its next tile from a counter (one per XCD) when it gets there, so the window of rows in flight
//      stays narrow however unevenly the waves progress;
//   2. the XCDs are dealt chunks of 256 KiB (kChunkTiles tiles) of that order -- 32 and 64 KiB chunks are a resonance of
//      the memory side (5.4 TB/s where 16 KiB or >= 128 KiB chunks give 6.2 - 6.6 for the same copy), and 256 KiB is the
//      measured optimum for this kernel (192 / 320 / 512 KiB: 5.5 / 5.7 / 6.0 TB/s);
//   3. few bytes in flight per CU: six waves with one tile ahead each (96 KiB) beat eight or twelve.
// Six waves per CU cannot share a 32 KiB W through LDS three ways (3 x (32 + 2 x 16) KiB), so each wave keeps the
// relation's W in REGISTERS: its 32 MFMA A fragments are 128 VGPRs, refilled through a 16 KiB staging area (two
// halves, by LDS-DMA + ds_read_b64_tr_b16 exactly as in the cyclic kernel) when the relation changes -- at most once
// per relation and workgroup, because a workgroup's tickets ascend.  A workgroup is a PAIR of waves (64-row tile):
//   * X: LDS-DMA into two 8 KiB stages per wave, the next tile in flight while this one is multiplied.  The DMA is
//     issued from inline asm so that the compiler does not know LDS is written behind its back -- it cannot tell the
//     stage buffers or the ticket ring apart and would otherwise put s_waitcnt vmcnt(0) before every LDS access;
//   * tickets: wave (s & 1) requests ticket s TWO tiles ahead with an asynchronous global atomic (inline asm, the
//     return value is collected one iteration later) and hands it to its partner through a 4-slot LDS ring -- the two
//     waves can never be more than two tickets apart, so no slot is overwritten before it was read;
//   * every wait names exactly how many YOUNGER vector-memory operations may stay in flight (vmcnt retires in order):
//     per iteration a wave issues [atomic] [8 DMA of tile i+1] ... [8 stores of tile i], always 8 stores (rows behind
//     the segment end rewrite its last row with that row's own data), so neither the previous tile's stores nor the
//     next tile's DMA are ever waited for;
//   * the loop nest is (runs of one relation) x (tiles): W is loop-invariant in the inner loop, otherwise the register
//     allocator copies all 128 registers around every iteration.
// Lane-derived values are re-derived where used (v_mbcnt, two VALU ops) instead of living in VGPRs across the kernel.
constexpr int kChunkTiles = 16;  // 16 x 64 rows x 256 B = 256 KiB of X (and of out) per XCD turn

__device__ __forceinline__ void wait_vmcnt_16_17(int n) {  // steady state: 16 or 17 younger operations; else drain
  if (n == 17) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
  else if (n == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename T>
__global__ __launch_bounds__(128, 2) void mfma_rows_ticket_kernel(const DevGroup* __restrict__ descs,
                                                                  const int32_t* __restrict__ tile_start, int B,
                                                                  unsigned int* __restrict__ tickets) {
  constexpr int NT = 4, NI = 8, NO = 8;
  typedef __attribute__((address_space(3))) void LDSV;
  typedef short v4i16 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) u32x4 GU32x4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  auto lane_now = [&]() -> int {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  char* wst = smem;                            // 16 KiB: W staging on a relation change, else 2 x 8 KiB epilogue scratch
  char* xs0 = smem + 16384 + wave * 16384;     // this wave's two X stages
  int* ring_val = (int*)(smem + 49152);        // [4] ticket values, [4] generations
  int* ring_gen = ring_val + 4;
  if (threadIdx.x < 8) ring_val[threadIdx.x] = 0;
  __syncthreads();
  const int k8 = blockIdx.x & 7;
  const int total = tile_start[B];
  unsigned int* my_ctr = tickets + k8 * 32;
  auto tile_of = [&](int v) -> int { return ((v / kChunkTiles) * 8 + k8) * kChunkTiles + v % kChunkTiles; };

  unsigned int raw = 0;  // lane 0: return value of the last request, valid once the matching wait has passed
  auto request = [&]() {
    unsigned long long sv;
    asm volatile(
        "s_nop 4\n\t"  // (as in issue_x)
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "s_nop 0\n\t"
        "global_atomic_add %[ret], %[off], %[one], %[base] sc0\n\t"
        "s_mov_b64 exec, %[sv]"
        : [ret] "+v"(raw), [sv] "=&s"(sv)
        : [off] "v"(0), [one] "v"(1u), [base] "s"(my_ctr)
        : "memory");
  };
  auto publish = [&](int s) -> int {  // after the wait for the request
    asm volatile("" : "+v"(raw));
    const int v = __builtin_amdgcn_readfirstlane((int)raw);
    if (lane_now() == 0) {
      __hip_atomic_store(&ring_val[s & 3], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(&ring_gen[s & 3], (s >> 2) + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return v;
  };
  auto consume = [&](int s) -> int {
    int v = 0;
    if (lane_now() == 0) {
      while (__hip_atomic_load(&ring_gen[s & 3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (s >> 2) + 1)
        __builtin_amdgcn_s_sleep(1);
      v = __hip_atomic_load(&ring_val[s & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return __builtin_amdgcn_readfirstlane(v);
  };

  u32x4 wreg[NI][NT];  // A fragment of k-step s, column block tt (layout: see mfma_rows_cyc_kernel)
  T* bias_lds = reinterpret_cast<T*>(smem + 49152 + 64);  // the relation's 128 bias values (read through LDS: a global
                                                         // load in the epilogue would drag an s_waitcnt vmcnt(0) along)
  auto load_w = [&](const char* w, const char* bias, int trans) {
    const int lane = lane_now(), h = lane >> 5;
    if (trans) {
      // `other` stored [M][K] (the dX pass hands W itself and asks for X W^T): a fragment -- 8 consecutive k of one
      // output column -- is then 16 contiguous bytes; lane x is A-row x, i.e. output column
      // 64 ((x >> 2) & 1) + 16 tt + 4 (x >> 3) + (x & 3) (the column mapping of the cyclic kernel's LDS image)
      __syncthreads();  // the partner is done with the previous relation's bias
      if (bias) bias_lds[threadIdx.x] = reinterpret_cast<const T*>(bias)[threadIdx.x];
      const int xx = lane & 31;
      const char* wl = w + (64 * ((xx >> 2) & 1) + 4 * (xx >> 3) + (xx & 3)) * 256 + 128 * h;
#pragma unroll
      for (int s = 0; s < NI; ++s)
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) wreg[s][tt] = *reinterpret_cast<const u32x4*>(wl + tt * 16 * 256 + s * 16);
      __syncthreads();  // the bias is in place
      return;
    }
    const int dma_r = (lane & 15) >> 2, dma_ii = lane & 3, dma_u = lane >> 4;
    const int dma_c = 2 * dma_u + (dma_ii & 1) + 8 * (dma_ii >> 1);
    const int dma_src_off = dma_r * 256 + dma_c * 16;
    const int q = lane & 15, grp16 = lane >> 4;
    const int a_lane_off = 8192 * h + (4 * (q >> 2) + (grp16 & 1) + 2 * (q & 1)) * 16 + ((q & 3) >> 1) * 8;
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {  // k-steps 4 r2 ... 4 r2 + 3: 1 KiB blocks {8 r2 ... 8 r2 + 7} of both k halves
      __syncthreads();
      if (r2 == 0 && bias) bias_lds[threadIdx.x] = reinterpret_cast<const T*>(bias)[threadIdx.x];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int kb = wave * 16 + r2 * 8 + jj;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + kb * 1024 + dma_src_off),
                                         (LDSV*)(wst + (wave * 8 + jj) * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const v4i16 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) v4i16*)(wst + a_lane_off + (2 * s4) * 1024 + tt * 256));
          const v4i16 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) v4i16*)(wst + a_lane_off + (2 * s4 + 1) * 1024 + tt * 256));
          wreg[r2 * 4 + s4][tt] = __builtin_bit_cast(u32x4, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
      }
    }
  };
  // 16-byte chunk cs of row r of a stage holds chunk cs ^ (r & 15) of the X row (permuted on the source side)
  struct Rel {  // the fields of a DevGroup this kernel uses (copying the whole struct sends its tail through scratch)
    const char* a;
    const char* w;
    char* c;
    const char* bias;
    int64_t rows;
    int trans;
  };
  auto rel_of = [&](int gi) -> Rel {
    const DevGroup* p = descs + gi;
    return Rel{p->a, p->w, p->c, p->bias, p->rows, p->trans};
  };
  auto issue_x = [&](const Rel& dg, int64_t row0, int buf) {
    const uint32_t lds = (uint32_t)(size_t)(xs0 + buf * 8192);
    const char* base = dg.a + row0 * 256;
    const int64_t left = dg.rows - row0;
    const int last = left < 32 ? (int)left - 1 : 31;
    const int l = lane_now();
    const int l4 = l >> 4, c0 = (l & 15) ^ l4;
    uint32_t off[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int r = 4 * i + l4;
      const int c = c0 ^ (4 * (i & 3));
      r = r > last ? last : r;
      off[i] = (uint32_t)(r * 256 + c * 16);
    }
    uint32_t sv;
    asm volatile(
        "s_nop 4\n\t"  // base / lds may have been written by v_readfirstlane: VALU-written SGPR -> VMEM address / M0
        "s_mov_b32 %[sv], m0\n\t"
        "s_mov_b32 m0, %[lds]\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[o0], %[base] nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %[o1], %[base] nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %[o2], %[base] nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %[o3], %[base] nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %[o4], %[base] nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %[o5], %[base] nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %[o6], %[base] nt\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %[o7], %[base] nt\n\t"
        "s_mov_b32 m0, %[sv]"
        : [sv] "=&s"(sv)
        : [lds] "s"(lds), [base] "s"(base), [o0] "v"(off[0]), [o1] "v"(off[1]), [o2] "v"(off[2]), [o3] "v"(off[3]),
          [o4] "v"(off[4]), [o5] "v"(off[5]), [o6] "v"(off[6]), [o7] "v"(off[7])
        : "memory", "scc");
  };

  // ticket 0 synchronously (wave 0); wave 1 requests ticket 1 only after that: a workgroup's tickets must ascend
  int t_cur;
  if (wave == 0) {
    request();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t_cur = tile_of(publish(0));
  } else {
    t_cur = tile_of(consume(0));
    request();
  }
  if (t_cur >= total) return;
  int g;
  {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tile_start[mid] <= t_cur) lo = mid; else hi = mid;
    }
    g = lo;
  }
  Rel d = rel_of(g);
  int64_t row0 = (int64_t)(t_cur - tile_start[g]) * kPairRows + wave * 32;
  bool valid = row0 < d.rows;
  int d_cur = 0;  // DMA instructions of the current tile (issued one iteration ago)
  if (valid) {
    issue_x(d, row0, 0);
    d_cur = 8;
  }
  int s_prev = 0;  // store instructions of the previous tile
  int buf = 0, i = 0;
  bool done = false;
  while (!done) {  // one pass per run of tiles of the same relation
    load_w(d.w, d.bias, d.trans);
    const int wcur = g;
    for (;; ++i) {
      // ticket i + 1 (requested one iteration ago by wave (i + 1) & 1; younger: this tile's DMA, the previous stores)
      int v_next;
      if (((i + 1) & 1) == wave) {
        wait_vmcnt_16_17(d_cur + s_prev);
        v_next = publish(i + 1);
      } else {
        v_next = consume(i + 1);
      }
      const int t_next = tile_of(v_next);
      const bool more = t_next < total;
      int a_now = 0;
      if (more && ((i + 2) & 1) == wave) {
        request();
        a_now = 1;
      }
      int gn = g;
      Rel dn = d;
      int64_t n_row0 = 0;
      bool n_valid = false;
      int d_next = 0;
      if (more) {
        if (t_next >= tile_start[gn + 1]) {
          do ++gn; while (t_next >= tile_start[gn + 1]);
          dn = rel_of(gn);
        }
        n_row0 = (int64_t)(t_next - tile_start[gn]) * kPairRows + wave * 32;
        n_valid = n_row0 < dn.rows;
        if (n_valid) {
          issue_x(dn, n_row0, buf ^ 1);
          d_next = 8;
        }
      }
      // this tile's X has landed (younger: the previous stores, the request, the next tile's DMA)
      wait_vmcnt_16_17(s_prev + a_now + d_next);
      int s_now = 0;
      if (valid) {
        const char* stage = xs0 + buf * 8192;
        char* scratch = wst + wave * 8192;
        const int lc = lane_now();
        const int xo = lc & 31, h = lc >> 5;
        const int cb = (NI * h) ^ (xo & 15);
        const bool has_bias = d.bias != nullptr;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {  // one 32-column block at a time: 16 accumulators next to the 128 of W
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          const char* xrow = stage + xo * 256;
          u32x4 xa = *reinterpret_cast<const u32x4*>(xrow + cb * 16);
#pragma unroll
          for (int s = 0; s < NI; ++s) {
            u32x4 xn = xa;
            if (s + 1 < NI) xn = *reinterpret_cast<const u32x4*>(xrow + (cb ^ (s + 1)) * 16);
            acc = mfma_chunk(T{}, wreg[s][tt], xa, acc);
            __builtin_amdgcn_sched_barrier(0);
            xa = xn;
          }
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[r];
          if (has_bias) {
            const T* bp = bias_lds + 64 * h + 16 * tt;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = round_to(T{}, v[r]) + load_bias(bp + r);
          }
#pragma unroll
          for (int j = 0; j < 2; ++j)
            *reinterpret_cast<u32x4*>(scratch + xo * 256 + (cb ^ (2 * tt + j)) * 16) = pack8(T{}, v + 8 * j);
        }
        // always 8 stores (the waits count them): lanes whose row lies behind the segment end rewrite its last row
        char* obase = d.c + row0 * 256;
        const int64_t left = d.rows - row0;
        const int last = left < 32 ? (int)left - 1 : 31;
        const int l = lane_now();
        const int l4 = l >> 4, cs = l & 15;
#pragma unroll
        for (int ii = 0; ii < NO; ++ii) {
          int r = 4 * ii + l4;
          r = r > last ? last : r;
          const u32x4 ov = *reinterpret_cast<const u32x4*>(scratch + r * 256 + cs * 16);
          GU32x4* dst = (GU32x4*)(obase + (uint32_t)(r * 256 + (cs ^ (r & 15)) * 16));
          __builtin_nontemporal_store(ov, dst);
          if ((ii & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        s_now = 8;
      }
      if (!more) {
        done = true;
        break;
      }
      t_cur = t_next;
      g = gn;
      d = dn;
      row0 = n_row0;
      valid = n_valid;
      d_cur = d_next;
      s_prev = s_now;
      buf ^= 1;
      if (g != wcur) {
        ++i;
        break;
      }
    }
  }
}

// ---- 16-bit, K = 256, 256 output columns per workgroup -------------------------------------------------
// With 128-column chunks an F = 256 layer needs two workgroups per tile range, i.e. every X tile travels
// through two CUs' load paths (C4: the kernel then moves 1.5x the algorithmic bytes at the same per-CU
// streaming rate as C2 and lands at 3.3 TB/s of useful traffic).  Here ONE workgroup owns all 256 columns:
// the whole weight matrix (128 KB) stays in LDS as two K-halves [256 columns][128 k] WITHOUT padding -- the
// 16-byte chunks of a row are XOR-swizzled with (row & 15) instead, which keeps the fragment reads
// conflict-free -- and the remaining 32 KB are four 8 KB stages.  A tile is two K-half passes into the same
// 8 x (32x32) accumulators (each pass = the K = 128 kernel's inner loop), X is read from HBM once, and the
// output leaves in two rounds of 4 column blocks through the stage (full 128-byte lines per row).
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void mfma_rows_wide256_kernel(
    const DevGroup* __restrict__ descs, const int32_t* __restrict__ tile_start, int B, int chunk, int ncol) {
  typedef __attribute__((address_space(1))) u32x4 GU32x4;
  constexpr int SZ = 2;
  constexpr int K = 256, KH = 128, MC = 256;
  constexpr int NT = MC / 32;          // 8 accumulator blocks
  constexpr int BM = NW * 32;
  static_assert(BM == kTileRows, "tile table is built for 128-row tiles");
  constexpr int CPR = KH * SZ / 16;    // 16 chunks per half row
  constexpr int NI = CPR / 2;          // 8 loads / K steps per half
  constexpr int WROW = KH * SZ;        // 256 bytes per image row
  constexpr int WIMG = MC * WROW;      // 64 KB per K-half
  constexpr int STAGE = 32 * 256;      // 8 KB per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int x = lane & 31;
  const int h = lane >> 5;
  const int bx = ncol > 1 ? ((int)blockIdx.x / (8 * ncol)) * 8 + ((int)blockIdx.x & 7) : (int)blockIdx.x;
  const int by = ncol > 1 ? ((int)blockIdx.x / 8) % ncol : 0;
  const int col0 = by * MC;
  char* stage = smem + 2 * WIMG + wave * STAGE;

  const int total = tile_start[B];
  const int G = (int)gridDim.x / ncol;
  int nloc, cbase = 0;
  if (chunk <= 0) {
    cbase = (int)((int64_t)bx * total / G);
    nloc = (int)((int64_t)(bx + 1) * total / G) - cbase;
  } else {
    const int nchunks = (total + chunk - 1) / chunk;
    const int mine = nchunks > bx ? (nchunks - 1 - bx) / G + 1 : 0;
    nloc = mine * chunk;
    if (mine > 0) {
      const int last_chunk = (mine - 1) * G + bx;
      const int over = (last_chunk + 1) * chunk - total;
      if (over > 0) nloc -= over;
    }
  }
  if (nloc <= 0) return;
  auto tile_of = [&](int i) -> int {
    if (chunk <= 0) return cbase + i;
    const int j = i / chunk;
    return (j * G + bx) * chunk + (i - j * chunk);
  };
  const int t1 = nloc;
  int lo = 0, hi = B;
  {
    const int first = tile_of(0);
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tile_start[mid] <= first) lo = mid; else hi = mid;
    }
  }
  int g = lo;
  int staged = -1;

  // image row (= output column) of lane x for accumulator block tt: crow0 + 16 tt; its swizzle is crow0 & 15
  const int crow0 = (MC / 2) * ((x >> 2) & 1) + 4 * (x >> 3) + (x & 3);
  const int wsw = crow0 & 15;
  const char* wrow = smem + crow0 * WROW;

  u32x4 xr[2][NI];
  DevGroup dn = descs[g];
  int64_t n_row0 = 0, n_rows = 0;
  bool n_valid = false;
  // plan(ti): which group / rows local tile ti covers for this wave; load_half(kh): its K-half into xr[kh]
  auto plan = [&](int ti) {
    const int t = tile_of(ti);
    while (t >= tile_start[g + 1]) {
      ++g;
      dn = descs[g];
    }
    n_rows = dn.rows;
    n_row0 = (int64_t)(t - tile_start[g]) * BM + wave * 32;
    n_valid = n_row0 < n_rows;
  };
  // The X loads are issued through inline asm and their vmcnt wait is placed by hand (wait_half): stores count
  // on vmcnt as well, and the wait the compiler would insert in front of the stage writes is a vmcnt(0) that
  // also waits for the output stores issued a moment earlier (microseconds per tile).  since[kh] = a LOWER bound
  // of the vector-memory instructions issued after the loads into xr[kh] (memory operations retire in order).
  int since[2] = {0, 0};
  auto load_half = [&](int kh) {
    if (!n_valid) return;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = i * 64 + lane;
      const int r = p / CPR;
      const int c = (p % CPR) ^ (r & 15);
      int64_t row = n_row0 + r;
      if (row >= n_rows) row = n_rows - 1;
      asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(xr[kh][i])
                   : "v"(dn.a + row * (K * SZ) + kh * (KH * SZ) + c * 16) : "memory");
    }
    since[kh] = 0;
    since[kh ^ 1] += NI;
  };
  auto wait_half = [&](int kh) {
    const int n = since[kh];
    if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  plan(0);
  load_half(0);
  load_half(1);
  DevGroup d = dn;
  int cg = g;
  int64_t row0 = n_row0, rows = n_rows;
  bool valid = n_valid;

  for (int t = 0; t < t1; ++t) {
    if (cg != staged) {
      __syncthreads();
      const char* w = d.w;
      const int M = d.m;
      if (!d.trans) {
        // W[k][m] row-major: 8 columns per 16-byte load, scattered as 2-byte stores into the swizzled image
        constexpr int CW = MC / 8;
        for (int idx = tid; idx < K * CW; idx += NW * 64) {
          const int k = idx / CW;
          const int cc = (idx - k * CW) * 8;
          const u32x4 v = *reinterpret_cast<const u32x4*>(w + ((int64_t)k * M + col0 + cc) * SZ);
          char* img = smem + (k >= KH ? WIMG : 0);
          const int kk = k & (KH - 1);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int m = cc + e;
            const uint16_t sv = (uint16_t)(v[e >> 1] >> ((e & 1) * 16));
            *reinterpret_cast<uint16_t*>(img + m * WROW + (((kk >> 3) ^ (m & 15)) * 16) + (kk & 7) * 2) = sv;
          }
        }
      } else {
        // W^T[m][k] row-major: whole chunks
        constexpr int CW = K / 8;
        for (int idx = tid; idx < MC * CW; idx += NW * 64) {
          const int m = idx / CW;
          const int kc = idx - m * CW;  // chunk of 8 k
          const u32x4 v = *reinterpret_cast<const u32x4*>(w + ((int64_t)(col0 + m) * K + kc * 8) * SZ);
          char* img = smem + (kc >= CPR ? WIMG : 0);
          *reinterpret_cast<u32x4*>(img + m * WROW + (((kc & (CPR - 1)) ^ (m & 15)) * 16)) = v;
        }
      }
      __syncthreads();
      staged = cg;
    }

    // the next tile's loads go out as soon as the registers of a half are free: right after that half has been
    // written to the stage, a whole tile before they are needed
    const bool have_next = t + 1 < t1;
    if (have_next) plan(t + 1);
    f32x16 acc[NT];
    if (!valid && have_next) {
      load_half(0);
      load_half(1);
    }
    if (valid) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        // this half of the tile -> stage (row order -> swizzled rows), then the K steps
        wait_half(kh);
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(stage + (i * 64 + lane) * 16) = xr[kh][i];
        if (have_next) load_half(kh);
        const char* img = wrow + kh * WIMG;
        u32x4 xa = *reinterpret_cast<const u32x4*>(stage + (x * CPR + ((NI * h) ^ (x & 15))) * 16);
        u32x4 wa[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
          wa[tt] = *reinterpret_cast<const u32x4*>(img + tt * 16 * WROW + (((NI * h) ^ wsw) * 16));
#pragma unroll
        for (int s = 0; s < NI; ++s) {
          asm volatile("" : "+v"(wa[NT - 1]));  // wait for this step's fragments before the next reads go out
          __builtin_amdgcn_sched_barrier(0);
          u32x4 xb = xa;
          u32x4 wb[NT];
          if (s + 1 < NI) {
            const int c = NI * h + s + 1;
            xb = *reinterpret_cast<const u32x4*>(stage + (x * CPR + (c ^ (x & 15))) * 16);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
              wb[tt] = *reinterpret_cast<const u32x4*>(img + tt * 16 * WROW + ((c ^ wsw) * 16));
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int tt = 0; tt < NT; ++tt) acc[tt] = mfma_chunk(T{}, wa[tt], xa, acc[tt]);
          __builtin_amdgcn_sched_barrier(0);
          if (s + 1 < NI) {
            xa = xb;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) wa[tt] = wb[tt];
          }
        }
      }
    }

    const DevGroup d_out = d;
    const int64_t row0_out = row0, rows_out = rows;
    const bool valid_out = valid;
    if (have_next) {
      d = dn;
      cg = g;
      row0 = n_row0;
      rows = n_rows;
      valid = n_valid;
    }

    if (valid_out) {
      const int M = d_out.m;
      char* obase = d_out.c + (row0_out * M + col0) * SZ;
      const bool full_out = row0_out + 32 <= rows_out;  // all 16 stores below are issued
      if (full_out) {
        since[0] += 16;
        since[1] += 16;
      }
      const T* bp = d_out.bias ? reinterpret_cast<const T*>(d_out.bias) + col0 + (MC / 2) * h : nullptr;
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {  // column blocks 4 rd .. 4 rd + 3 of both lane halves
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int tt = 4 * rd + q;
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[tt][r];
          if (bp) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = round_to(T{}, v[r]) + load_bias(bp + 16 * tt + r);
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int c = 8 * h + 2 * q + j;  // 16 chunks per stage row: [half 0: 8 chunks | half 1: 8 chunks]
            *reinterpret_cast<u32x4*>(stage + (x * 16 + (c ^ (x & 15))) * 16) = pack_chunk(T{}, v + 8 * j);
          }
        }
        u32x4 ov[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ov[i] = *reinterpret_cast<const u32x4*>(stage + (i * 64 + lane) * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int p = i * 64 + lane;
          const int r = p >> 4;
          const int c = (p & 15) ^ (r & 15);
          if (full_out || row0_out + r < rows_out) {
            // columns (MC/2) * half + 64 rd + 8 * (c & 7)
            GU32x4* dst = (GU32x4*)(obase + (int64_t)r * M * SZ + ((MC / 2) * (c >> 3) + 64 * rd + 8 * (c & 7)) * SZ);
            __builtin_nontemporal_store(ov[i], dst);
          }
        }
      }
    }
  }
}

// ---- 16-bit, K = 256, 256 output columns, 64 rows per wave ----------------------------------------------------------
// PMC on the kernel above (C4, profiles/r2_pmc_c4_wide256.json): MFMA pipes busy 24 %, half of the wave cycles issuing
// and 30 % waiting -- its inner loop is bound by LDS reads, not by MFMA: every step a wave reads 8 KiB of W fragments
// + 1 KiB of X for 8 MFMAs (256 cycles), and the four waves of the CU share one 128 B/clk LDS (>= 288 cycles).  Here a
// wave owns TWO 32-row blocks (256-row workgroup tiles) and every W fragment feeds two MFMAs: 10 KiB of LDS reads per
// 16 MFMAs.  The 16 accumulator blocks (256 registers) live in AGPRs (one wave per SIMD: 512 registers), X arrives a
// K-QUARTER at a time -- 64 rows x 64 k = the wave's 8 KiB stage -- through two register buffers that are refilled
// two quarters ahead, and lane half h multiplies the 16-byte k-chunk 2 u + h in step u (a quarter is then one
// contiguous 128-byte line per row).  W image, column mapping and epilogue follow the kernel above.
//   There is no "this wave has no rows in this tile" path: such a wave (and every row behind a segment's end) works on
// the segment's LAST row instead -- loads clamp to it, its result is stored to it again (same bytes as its owner
// writes).  Every tile is then the same straight line of 32 loads and 32 stores, the loads are ordinary (compiler
// visible) loads, and with the first tile peeled the compiler's own s_waitcnt vmcnt counts are exact: the previous
// tile's stores stay in flight while this tile multiplies.  (A version with the loads in inline asm and hand-kept
// counts, as in the kernel above, broke on the invalid -> valid transition: the compiler may copy an asm output
// register at a control-flow merge before the data has arrived.)
template <typename T>
__global__ __launch_bounds__(256) void mfma_rows_wide256r2_kernel(const DevGroup* __restrict__ descs,
                                                                  const int32_t* __restrict__ tile_start, int B) {
  typedef __attribute__((address_space(1))) u32x4 GU32x4;
  constexpr int SZ = 2, NW = 4;
  constexpr int K = 256, KH = 128, MC = 256;
  constexpr int NT = MC / 32;      // 8 column blocks
  constexpr int BM = NW * 64;      // 256-row tiles
  static_assert(BM == 2 * kTileRows, "tile_start2 is built for 256-row tiles");
  constexpr int WROW = KH * SZ;    // 256 bytes per image row
  constexpr int WIMG = MC * WROW;  // 64 KB per K-half
  constexpr int STAGE = 8192;      // per wave: 64 rows x 128 B (a K quarter) / 32 rows x 256 B (an output round)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x = lane & 31;
  const int h = lane >> 5;
  const int bx = blockIdx.x, G = gridDim.x;
  char* stage = smem + 2 * WIMG + wave * STAGE;

  const int total = tile_start[B];
  const int cbase = (int)((int64_t)bx * total / G);
  const int t1 = (int)((int64_t)(bx + 1) * total / G) - cbase;
  if (t1 <= 0) return;
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_start[mid] <= cbase) lo = mid; else hi = mid;
  }
  int g = lo;
  int staged = -1;

  // image row (= output column) of lane x for column block tt: crow0 + 16 tt; its swizzle is crow0 & 15
  const int crow0 = (MC / 2) * ((x >> 2) & 1) + 4 * (x >> 3) + (x & 3);
  const int wsw = crow0 & 15;
  const char* wrow = smem + crow0 * WROW;

  struct Rel {  // what a tile needs of its relation; `first` = the wave's first row, clamped into the segment
    const char* a;
    const char* w;
    char* c;
    const char* bias;
    int64_t first;
    int last;  // rows first .. first + last exist (0 <= last <= 63); later rows of the wave stand for first + last
    int trans;
    int group;
  };
  Rel nx;
  auto plan = [&](int ti) {
    const int t = cbase + ti;
    while (t >= tile_start[g + 1]) ++g;
    const DevGroup* p = descs + g;
    const int64_t rows = p->rows;
    int64_t r0 = (int64_t)(t - tile_start[g]) * BM + wave * 64;
    if (r0 > rows - 1) r0 = rows - 1;
    const int64_t left = rows - r0;
    nx = Rel{p->a, p->w, p->c, p->bias, r0, left < 64 ? (int)left - 1 : 63, p->trans, g};
  };
  // quarter q of the wave's 64 rows: instruction i covers rows 8 i .. 8 i + 7, lane l reads chunk
  // (l & 7) ^ ((row >> 1) & 7) of its row's 128-byte quarter, so that the linear stage write leaves chunk c at slot
  // c ^ ((row >> 1) & 7): ds_read_b128 serves the lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32) in one
  // cycle each over 64 banks, and with 128-byte rows the bank is (row & 1, slot) -- (row >> 1) & 7 is a permutation of
  // 0..7 over the even rows of either group and over the odd ones (row & 7 gave 2-way conflicts).
  u32x4 xr[2][8];
  const int l3 = lane >> 3;
  auto load_q = [&](int b, const Rel& rl, int q) {
    const char* base = rl.a + rl.first * (K * SZ) + q * 128;  // wave-uniform
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 8 * i + l3;
      const int coff = ((lane & 7) ^ ((row >> 1) & 7)) * 16;
      const int r = row > rl.last ? rl.last : row;
      xr[b][i] = __builtin_nontemporal_load((const GU32x4*)(base + (uint32_t)(r * (K * SZ) + coff)));
    }
  };

  Rel cur;
  auto tile_body = [&](int t) {
    if (cur.group != staged) {
      __syncthreads();
      const char* w = cur.w;
      if (!cur.trans) {
        // W[k][m] row-major: 8 columns per 16-byte load, scattered as 2-byte stores into the swizzled image
        constexpr int CW = MC / 8;
        for (int idx = tid; idx < K * CW; idx += NW * 64) {
          const int k = idx / CW;
          const int cc = (idx - k * CW) * 8;
          const u32x4 v = *reinterpret_cast<const u32x4*>(w + ((int64_t)k * MC + cc) * SZ);
          char* img = smem + (k >= KH ? WIMG : 0);
          const int kk = k & (KH - 1);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int m = cc + e;
            const uint16_t sv = (uint16_t)(v[e >> 1] >> ((e & 1) * 16));
            *reinterpret_cast<uint16_t*>(img + m * WROW + (((kk >> 3) ^ (m & 15)) * 16) + (kk & 7) * 2) = sv;
          }
        }
      } else {
        // W^T[m][k] row-major: whole chunks
        constexpr int CW = K / 8;
        for (int idx = tid; idx < MC * CW; idx += NW * 64) {
          const int m = idx / CW;
          const int kc = idx - m * CW;
          const u32x4 v = *reinterpret_cast<const u32x4*>(w + ((int64_t)m * K + kc * 8) * SZ);
          char* img = smem + (kc >= 16 ? WIMG : 0);
          *reinterpret_cast<u32x4*>(img + m * WROW + (((kc & 15) ^ (m & 15)) * 16)) = v;
        }
      }
      __syncthreads();
      staged = cur.group;
    }
    const bool have_next = t + 1 < t1;
    f32x16 acc[2][NT];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][i][r] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int b = q & 1;
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(stage + (i * 64 + lane) * 16) = xr[b][i];
      // refill this buffer two quarters ahead (the last tile of the workgroup reloads its own quarters: same count
      // of loads on every path, and nothing reads them)
      if (q < 2) {
        load_q(b, cur, q + 2);
      } else {
        if (q == 2) {
          if (have_next) plan(t + 1); else nx = cur;
        }
        load_q(b, nx, q - 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      const char* img = wrow + (q >> 1) * WIMG;
      const char* xrow0 = stage + x * 128;
      const char* xrow1 = stage + (32 + x) * 128;
      const int xs = (x >> 1) & 7;  // rows x and 32 + x share it
      const int c0 = ((8 * q) & 15) + h;  // chunk within the K half of step j: c0 + 2 j
      u32x4 xa0 = *reinterpret_cast<const u32x4*>(xrow0 + ((h ^ xs) * 16));
      u32x4 xa1 = *reinterpret_cast<const u32x4*>(xrow1 + ((h ^ xs) * 16));
      u32x4 wa[NT];
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
        wa[tt] = *reinterpret_cast<const u32x4*>(img + tt * 16 * WROW + ((c0 ^ wsw) * 16));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asm volatile("" : "+v"(wa[NT - 1]));  // wait for this step's fragments before the next reads go out
        __builtin_amdgcn_sched_barrier(0);
        u32x4 xb0 = xa0, xb1 = xa1;
        u32x4 wb[NT];
        if (j + 1 < 4) {
          const int cx = 2 * (j + 1) + h;
          xb0 = *reinterpret_cast<const u32x4*>(xrow0 + ((cx ^ xs) * 16));
          xb1 = *reinterpret_cast<const u32x4*>(xrow1 + ((cx ^ xs) * 16));
          const int c = c0 + 2 * (j + 1);
#pragma unroll
          for (int tt = 0; tt < NT; ++tt)
            wb[tt] = *reinterpret_cast<const u32x4*>(img + tt * 16 * WROW + ((c ^ wsw) * 16));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          acc[0][tt] = mfma_chunk(T{}, wa[tt], xa0, acc[0][tt]);
          acc[1][tt] = mfma_chunk(T{}, wa[tt], xa1, acc[1][tt]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < 4) {
          xa0 = xb0;
          xa1 = xb1;
#pragma unroll
          for (int tt = 0; tt < NT; ++tt) wa[tt] = wb[tt];
        }
      }
    }
    // 32 stores, always: row r of the wave goes to row min(r, last) (rows behind the end hold the last row's result)
    const T* bp = cur.bias ? reinterpret_cast<const T*>(cur.bias) + (MC / 2) * h : nullptr;
    char* obase = cur.c + cur.first * MC * SZ;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {  // column blocks 4 rd .. 4 rd + 3 of both lane halves
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int tt = 4 * rd + q4;
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[rb][tt][r];
          if (bp) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = round_to(T{}, v[r]) + load_bias(bp + 16 * tt + r);
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int c = 8 * h + 2 * q4 + j;  // 16 chunks per stage row: [half 0: 8 chunks | half 1: 8 chunks]
            *reinterpret_cast<u32x4*>(stage + (x * 16 + (c ^ (x & 15))) * 16) = pack_chunk(T{}, v + 8 * j);
          }
        }
        u32x4 ov[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ov[i] = *reinterpret_cast<const u32x4*>(stage + (i * 64 + lane) * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int p = i * 64 + lane;
          const int r = p >> 4;
          const int c = (p & 15) ^ (r & 15);
          int ro = 32 * rb + r;
          ro = ro > cur.last ? cur.last : ro;
          GU32x4* dst = (GU32x4*)(obase + (uint32_t)(ro * MC * SZ + ((MC / 2) * (c >> 3) + 64 * rd + 8 * (c & 7)) * SZ));
          __builtin_nontemporal_store(ov[i], dst);
        }
      }
    }
    cur = nx;
  };

  plan(0);
  cur = nx;
  load_q(0, cur, 0);
  load_q(1, cur, 1);
  tile_body(0);  // peeled: inside the loop the memory operations in flight are the same on entry and on the back edge
  for (int t = 1; t < t1; ++t) tile_body(t);
}


// ---- fp32 variant with a pipelined epilogue ---------------------------------------------------------
// fp32 at K = 128 is bound by the MFMA rate (AI = 32 flop/B), and the weight image + X stages leave room for
// one 4-wave workgroup per CU: with one wave per SIMD nothing hides a tile's epilogue (accumulators ->
// row order -> HBM), which mfma_rows_lds_kernel runs after the tile's last MFMA.  Here the accumulators are
// double-buffered and the epilogue of tile t-1 is cut into single instructions that are issued BETWEEN the
// MFMA groups of tile t (the wave issues in order: anything placed behind a block of MFMAs waits for all
// of them to issue).  The output goes through its own 4 KB per-wave stage, one 32-column block at a time:
//   step s of the K loop (NI steps, 4 MFMA groups each)  ->  block tt = s / (NI/NT):
//     first step  : 4 x ds_write_b128 (accumulator fragments, + bias)
//     second step : 4 x ds_read_b128  (row order)
//     third step  : 4 x global_store_dwordx4 (two 64-byte runs per row and instruction)
// X staging and the loads of tile t+2 stay between the tiles, as in mfma_rows_lds_kernel.
template <int K, int MC, int NW, int DBG = 0>  // DBG (timing experiments only): 1 = no HBM loads, 2 = no stores, 4 = no MFMAs
__global__ __launch_bounds__(NW * 64) void mfma_rows_f32_pipe_kernel(
    const DevGroup* __restrict__ descs, const int32_t* __restrict__ tile_start, int B, int chunk, int ncol) {
  // global_* instructions (a flat access would also count on lgkmcnt and make every LDS wait conservative)
  typedef __attribute__((address_space(1))) u32x4 GU32x4;
  typedef __attribute__((address_space(1))) float GF32;
  constexpr int SZ = 4;
  constexpr int NT = MC / 32;
  constexpr int LDW = K * SZ + 16;
  constexpr int BM = NW * 32;
  static_assert(BM == kTileRows, "tile table is built for 128-row tiles");
  constexpr int CPR = K * SZ / 16;
  constexpr int NI = CPR / 2;
  constexpr int XM = (CPR < 16 ? CPR : 16) - 1;
  constexpr int STAGE = 32 * 16 * CPR;
  constexpr int OSTAGE = 32 * 128;
  constexpr int WBYTES = MC * LDW;
  constexpr int SPT = NI / NT;  // K steps per output block
  static_assert(NI % NT == 0 && SPT >= 3, "epilogue pieces need three K steps per output block");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int x = lane & 31;
  const int h = lane >> 5;
  const int bx = ncol > 1 ? ((int)blockIdx.x / (8 * ncol)) * 8 + ((int)blockIdx.x & 7) : (int)blockIdx.x;
  const int by = ncol > 1 ? ((int)blockIdx.x / 8) % ncol : 0;
  const int col0 = by * MC;
  char* stage = smem + WBYTES + wave * STAGE;
  char* ostage = smem + WBYTES + NW * STAGE + wave * OSTAGE;

  const int total = tile_start[B];
  const int G = (int)gridDim.x / ncol;
  int nloc, cbase = 0;
  if (chunk <= 0) {
    cbase = (int)((int64_t)bx * total / G);
    nloc = (int)((int64_t)(bx + 1) * total / G) - cbase;
  } else {
    const int nchunks = (total + chunk - 1) / chunk;
    const int mine = nchunks > bx ? (nchunks - 1 - bx) / G + 1 : 0;
    nloc = mine * chunk;
    if (mine > 0) {
      const int last_chunk = (mine - 1) * G + bx;
      const int over = (last_chunk + 1) * chunk - total;
      if (over > 0) nloc -= over;
    }
  }
  if (nloc <= 0) return;
  auto tile_of = [&](int i) -> int {
    if (chunk <= 0) return cbase + i;
    const int j = i / chunk;
    return (j * G + bx) * chunk + (i - j * chunk);
  };
  const int t1 = nloc;

  int lo = 0, hi = B;
  {
    const int first = tile_of(0);
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tile_start[mid] <= first) lo = mid; else hi = mid;
    }
  }
  int g = lo;
  int staged = -1;

  const int crow0 = (MC / 2) * ((x >> 2) & 1) + 4 * (x >> 3) + (x & 3);
  const char* wfrag = smem + crow0 * LDW + (K / 2) * h * SZ;

  u32x4 xr[NI];
  uint32_t xoff[NI];  // byte offset of this lane's 16 bytes of load i inside a whole 32-row tile
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int p = i * 64 + lane;
    const int r = p / CPR;
    xoff[i] = (uint32_t)(r * (K * SZ) + (((p % CPR) ^ (r & XM)) * 16));
  }
  DevGroup dn = descs[g];
  int64_t n_row0 = 0, n_rows = 0;
  bool n_valid = false;
  // plan(ti): group / rows of local tile ti for this wave (scalar work); the loads themselves are issued either
  // between the first MFMA groups of the tile computed meanwhile (issue_in_loop: whole tiles, tile base + the
  // per-lane offsets computed once) or all at once (issue_all).  They go through inline asm so that their WAIT
  // is placed by hand (stage_x): stores count on vmcnt too, and the compiler -- which cannot see across this
  // loop's branches that exactly the 16 stores of the overlapped epilogue are younger -- would wait for those
  // stores as well (vmcnt(0): several microseconds of store latency per tile).
  const char* n_base = nullptr;  // first byte of a whole tile (+ the per-lane offsets computed once)
  bool n_whole = false;
  auto plan = [&](int ti) {
    const int t = tile_of(ti);
    while (t >= tile_start[g + 1]) {
      ++g;
      dn = descs[g];
    }
    n_rows = dn.rows;
    n_row0 = (int64_t)(t - tile_start[g]) * BM + wave * 32;
    n_valid = n_row0 < n_rows;
    n_whole = n_valid && n_row0 + 32 <= n_rows;
    n_base = dn.a + n_row0 * (K * SZ);
  };
  auto issue_in_loop = [&](int i) {
    if constexpr ((DBG & 1) == 0)
      asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(xr[i]) : "v"(n_base + xoff[i]) : "memory");
    else
      asm volatile("v_mov_b32 %0, %1" : "=v"(xr[i][0]) : "v"(xoff[i]));
  };
  auto issue_all = [&]() {
    if (!n_valid) return;
    if (n_whole) {
#pragma unroll
      for (int i = 0; i < NI; ++i) issue_in_loop(i);
    } else {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int p = i * 64 + lane;
        const int r = p / CPR;
        const int cs = p % CPR;
        const int c = cs ^ (r & XM);
        int64_t row = n_row0 + r;
        if (row >= n_rows) row = n_rows - 1;
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(xr[i]) : "v"(dn.a + row * (K * SZ) + c * 16) : "memory");
      }
    }
  };
  // `younger`: exactly the 4 * NT unpredicated stores of an overlapped epilogue (four per 32-column block) were issued
  // after the loads.  (The count was a fixed 16 until round 2: right for MC = 128 only -- with MC = 64 / 32 the wait
  // let 8 / 12 LOADS stay in flight, which surfaced as one garbage tile in one of ~15 runs of the parity suite.)
  auto stage_x = [&](bool younger) {
    if (!younger) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (NT == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (NT == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (NT == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(stage + (i * 64 + lane) * 16) = xr[i];
  };

  // the tile whose accumulators wait for their epilogue
  bool p_valid = false;
  char* p_obase = nullptr;
  int64_t p_left = 0;    // rows of the tile that exist
  int64_t p_pitch = 0;   // output row pitch in bytes
  const GF32* p_bias = nullptr;
  bool p_full = false;   // all 32 rows exist: stores need no predicate

  // one epilogue instruction: part e (0..3) of K step s, for the pending tile's accumulators
  u32x4 ov[4];
  auto piece = [&](const f32x16 (&acc)[NT], int s, int e, auto full_tile) {
    constexpr bool FULL = decltype(full_tile)::value;
    const int tt = s / SPT;
    const int sub = s - tt * SPT;
    if (sub == 0) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[tt][4 * e + j];
      if (!FULL && p_bias) {  // (tiles with a bias take the predicated path)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += p_bias[(MC / 2) * h + 16 * tt + 4 * e + j];
      }
      *reinterpret_cast<f32x4*>(ostage + (x * 8 + ((h * 4 + e) ^ ((x >> 1) & 7))) * 16) = v;
    } else if (sub == 1) {
      ov[e] = *reinterpret_cast<const u32x4*>(ostage + (e * 64 + lane) * 16);
    } else if (sub == 2) {
      const int p = e * 64 + lane;
      const int r = p >> 3;
      const int c = (p & 7) ^ ((r >> 1) & 7);
      if (FULL || r < p_left) {
        GU32x4* dst = (GU32x4*)(p_obase + (int64_t)r * p_pitch + ((MC / 2) * (c >> 2) + 16 * tt + 4 * (c & 3)) * SZ);
        if constexpr ((DBG & 2) == 0) __builtin_nontemporal_store(ov[e], dst);
        else if (ov[e][0] == 0x12345678u && ov[e][1] == 0x9abcdef0u) *dst = ov[e];
      }
    }
  };
  auto flush = [&](const f32x16 (&acc)[NT]) {
#pragma unroll
    for (int s = 0; s < NI; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) piece(acc, s, e, std::false_type{});
  };

  constexpr int LPG = (NI + 7) / 8;  // loads per MFMA group when the next tile's loads ride in K steps 0 and 1
  auto multiply = [&](f32x16 (&accC)[NT], const f32x16 (&accP)[NT], auto has_prev, bool loads) {
    constexpr bool HAS_PREV = decltype(has_prev)::value;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) accC[i][r] = 0.f;
    u32x4 xa = *reinterpret_cast<const u32x4*>(stage + (x * CPR + ((NI * h) ^ (x & XM))) * 16);
    u32x4 wa[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) wa[tt] = *reinterpret_cast<const u32x4*>(wfrag + tt * 16 * LDW);
#pragma unroll
    for (int s = 0; s < NI; ++s) {
      // Wait for this step's fragments BEFORE the next step's reads are issued: they were issued a whole
      // step (16 MFMAs) ago, so this costs nothing -- whereas the wait the compiler would place in front
      // of the first MFMA is an lgkmcnt(0) that also covers the reads issued just before it.
      asm volatile("" : "+v"(wa[NT - 1]));
      __builtin_amdgcn_sched_barrier(0);
      u32x4 xb = xa;
      u32x4 wb[NT];
      if (s + 1 < NI) {
        const int c = NI * h + s + 1;
        xb = *reinterpret_cast<const u32x4*>(stage + (x * CPR + (c ^ (x & XM))) * 16);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
          wb[tt] = *reinterpret_cast<const u32x4*>(wfrag + tt * 16 * LDW + (s + 1) * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 xf = __builtin_bit_cast(f32x4, xa);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          if constexpr ((DBG & 4) == 0)
            accC[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(f32x4, wa[tt])[e], xf[e], accC[tt], 0, 0, 0);
          else
            accC[tt][e] += __builtin_bit_cast(f32x4, wa[tt])[e] * xf[e];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s < 2 && loads) {  // all of them before the first stores (K step 2) of the overlapped epilogue
#pragma unroll
          for (int q = 0; q < LPG; ++q)
            if ((s * 4 + e) * LPG + q < NI) issue_in_loop((s * 4 + e) * LPG + q);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (HAS_PREV) {
          piece(accP, s, e, std::true_type{});
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (s + 1 < NI) {
        xa = xb;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) wa[tt] = wb[tt];
      }
    }
  };

  plan(0);
  issue_all();
  DevGroup d = dn;
  int cg = g;
  int64_t row0 = n_row0, rows = n_rows;
  bool valid = n_valid;
  if (valid) stage_x(false);
  bool pending = false;  // a planned tile whose loads have not been issued yet
  if (1 < t1) {
    plan(1);
    pending = true;
  }

  int t = 0;
  uint64_t dbg_t[4] = {0, 0, 0, 0};
  uint64_t dbg_start = 0;
  if constexpr ((DBG & 8) != 0) dbg_start = __builtin_readcyclecounter();
  auto one = [&](f32x16 (&accC)[NT], f32x16 (&accP)[NT]) {
    if (cg != staged) {
      __syncthreads();
      const char* w = d.w;
      const int M = d.m;
      if (!d.trans) {
        constexpr int CW = MC / 4;
        for (int idx = tid; idx < K * CW; idx += NW * 64) {
          const int k = idx / CW;
          const int cc = (idx - k * CW) * 4;
          const u32x4 v = *reinterpret_cast<const u32x4*>(w + ((int64_t)k * M + col0 + cc) * SZ);
#pragma unroll
          for (int e = 0; e < 4; ++e) *reinterpret_cast<uint32_t*>(smem + (cc + e) * LDW + k * 4) = v[e];
        }
      } else {
        constexpr int CW = K / 4;
        for (int idx = tid; idx < MC * CW; idx += NW * 64) {
          const int c = idx / CW;
          const int kk = (idx - c * CW) * 4;
          *reinterpret_cast<u32x4*>(smem + c * LDW + kk * SZ) =
              *reinterpret_cast<const u32x4*>(w + ((int64_t)(col0 + c) * K + kk) * SZ);
        }
      }
      __syncthreads();
      staged = cg;
    }
    // the overlapped epilogue is the unpredicated one (whole 32-row tiles without bias); the last tile of a
    // group / biased outputs are flushed on their own
    bool flushed = false;
    if (p_valid && !(valid && p_full)) {
      flush(accP);
      p_valid = false;
      flushed = true;
    }
    bool overlapped = false;
    uint64_t c0 = 0;
    if constexpr ((DBG & 8) != 0) c0 = __builtin_readcyclecounter();
    // the planned tile's loads: inside this tile's K loop when both are whole tiles, else right here
    const bool loads_in_loop = pending && valid && n_whole;
    if (pending && !loads_in_loop) issue_all();
    pending = false;
    if (valid) {
      if (p_valid) {
        multiply(accC, accP, std::true_type{}, loads_in_loop);
        overlapped = true;
      } else {
        multiply(accC, accP, std::false_type{}, loads_in_loop);
      }
    }
    if constexpr ((DBG & 8) != 0) {
      const uint64_t c1 = __builtin_readcyclecounter();
      dbg_t[0] += c1 - c0;
      c0 = c1;
    }
    // this tile's accumulators are the pending ones now
    p_valid = valid;
    if (valid) {
      p_pitch = (int64_t)d.m * SZ;
      p_obase = d.c + row0 * p_pitch + (int64_t)col0 * SZ;
      p_left = rows - row0;
      p_bias = d.bias ? (const GF32*)(reinterpret_cast<const float*>(d.bias) + col0) : nullptr;
      p_full = p_left >= 32 && !d.bias;
    }
    if (t + 1 < t1) {
      d = dn;
      cg = g;
      row0 = n_row0;
      rows = n_rows;
      valid = n_valid;
      if constexpr ((DBG & 8) != 0) {
        const uint64_t c1 = __builtin_readcyclecounter();
        dbg_t[1] += c1 - c0;
        c0 = c1;
      }
      // younger than the staged tile's loads: exactly the 4 * NT stores of the overlapped epilogue -- unless those
      // loads were issued before this iteration's flush / restage traffic (then every older access has to land)
      if (valid) stage_x(overlapped && !flushed && loads_in_loop);
      if constexpr ((DBG & 8) != 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint64_t c1 = __builtin_readcyclecounter();
        dbg_t[2] += c1 - c0;
        c0 = c1;
      }
      if (t + 2 < t1) {
        plan(t + 2);
        pending = true;
      }
      if constexpr ((DBG & 8) != 0) {
        const uint64_t c1 = __builtin_readcyclecounter();
        dbg_t[3] += c1 - c0;
        c0 = c1;
      }
    }
  };

  f32x16 acc0[NT], acc1[NT];
  for (;;) {
    one(acc0, acc1);
    if (++t >= t1) {
      if (p_valid) flush(acc0);
      break;
    }
    one(acc1, acc0);
    if (++t >= t1) {
      if (p_valid) flush(acc1);
      break;
    }
  }
  if constexpr ((DBG & 8) != 0) {
    if (blockIdx.x == 17 && tid == 64) {
      const uint64_t tot = __builtin_readcyclecounter() - dbg_start;
      printf("[pipe dbg] tiles %d total %llu | multiply %llu bookkeeping %llu stage_x %llu prefetch %llu (cycles)\n", t1,
             (unsigned long long)tot, (unsigned long long)dbg_t[0], (unsigned long long)dbg_t[1],
             (unsigned long long)dbg_t[2], (unsigned long long)dbg_t[3]);
    }
  }
}

// ---- generic kernel: one thread per output element, any dtype / shape ----------------------------
template <typename T, typename Acc>
struct NaiveCvt {
  __device__ static Acc load(const T* p) { return (Acc)*p; }
  __device__ static void store(T* p, Acc v) { *p = (T)v; }
  __device__ static Acc round(Acc v) { return (Acc)(T)v; }
};
template <>
struct NaiveCvt<bf16_t, float> {
  __device__ static float load(const bf16_t* p) { return load_bias(p); }
  __device__ static void store(bf16_t* p, float v) {
    p->v = __builtin_bit_cast(uint16_t, (__bf16)v);
  }
  __device__ static float round(float v) { return (float)(__bf16)v; }
};
template <>
struct NaiveCvt<f16_t, float> {
  __device__ static float load(const f16_t* p) { return load_bias(p); }
  __device__ static void store(f16_t* p, float v) {
    p->v = __builtin_bit_cast(uint16_t, (_Float16)v);
  }
  __device__ static float round(float v) { return (float)(_Float16)v; }
};

template <typename T, typename Acc>
__global__ void naive_kernel(const DevGroup* __restrict__ descs,
                             const int64_t* __restrict__ out_start, int B, int64_t total) {
  // out_start[g] = number of output elements in groups < g
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    if (idx >= out_start[B]) break;
    int lo = 0, hi = B;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (out_start[mid] <= idx) lo = mid; else hi = mid;
    }
    const DevGroup d = descs[lo];
    const int64_t local = idx - out_start[lo];
    const int64_t r = local / d.m;
    const int c = (int)(local - r * d.m);
    const T* a = reinterpret_cast<const T*>(d.a) + r * d.k;
    const T* w = reinterpret_cast<const T*>(d.w);
    Acc acc = 0;
    if (!d.trans) {
      for (int k = 0; k < d.k; ++k)
        acc += NaiveCvt<T, Acc>::load(a + k) * NaiveCvt<T, Acc>::load(w + (int64_t)k * d.m + c);
    } else {
      for (int k = 0; k < d.k; ++k)
        acc += NaiveCvt<T, Acc>::load(a + k) * NaiveCvt<T, Acc>::load(w + (int64_t)c * d.k + k);
    }
    if (d.bias)
      acc = NaiveCvt<T, Acc>::round(acc) +
            NaiveCvt<T, Acc>::load(reinterpret_cast<const T*>(d.bias) + c);
    NaiveCvt<T, Acc>::store(reinterpret_cast<T*>(d.c) + r * d.m + c, acc);
  }
}

__global__ void out_start_kernel(const DevGroup* __restrict__ descs, int B,
                                 int64_t* __restrict__ out_start) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int64_t s = 0;
    for (int b = 0; b < B; ++b) {
      out_start[b] = s;
      s += descs[b].rows * descs[b].m;
    }
    out_start[B] = s;
  }
}

// ---- host side -----------------------------------------------------------------------------------
thread_local const char* g_last_variant = "";

// Optional per-launch timing of the dominant kernel (bench.py roofline leg): when enabled, a pair of
// HIP events brackets the main kernel on the stream it is launched on.
struct ProfPair { hipEvent_t a, b; };
// The mode of the call in flight on this thread, decoded from the entry point's `flags` argument (pyg_hip.h): nothing
// here outlives a call, so two threads with different modes never see each other's choice.
// tile schedule (PYG_HIP_MM_SCHED_*): 0 = automatic, 1 = contiguous ranges, 2 = cyclic, ...
thread_local int g_schedule = 0;
// fp32 K = 128, M % 128 == 0: 1 = split-bf16 arithmetic (PYG_HIP_MM_F32_SPLIT), 0 = v_mfma_f32_32x32x2_f32
thread_local int g_f32_split = 0;

inline int decode_mode(int flags) {
  const int sched = flags & PYG_HIP_MM_SCHED_MASK;
  if (sched > PYG_HIP_MM_SCHED_RING || (flags & ~(PYG_HIP_MM_SCHED_MASK | PYG_HIP_MM_F32_SPLIT)) != 0) return -1;
  g_schedule = sched;
  g_f32_split = (flags & PYG_HIP_MM_F32_SPLIT) ? 1 : 0;
  return 0;
}
// 16-bit K = M = 128: relations shorter than this on average take the item-ring kernel (automatic schedule)
constexpr int64_t kRingMeanRows = 4096;
thread_local bool g_prof_on = false;
thread_local std::vector<ProfPair> g_prof;

struct ProfScope {
  hipStream_t s;
  bool on;
  ProfPair p;
  explicit ProfScope(hipStream_t st) : s(st), on(g_prof_on) {
    if (on) {
      on = hipEventCreate(&p.a) == hipSuccess && hipEventCreate(&p.b) == hipSuccess;
      if (on) (void)hipEventRecord(p.a, s);
    }
  }
  ~ProfScope() {
    if (on) {
      (void)hipEventRecord(p.b, s);
      g_prof.push_back(p);
    }
  }
};

struct Workspace {
  DevGroup* descs;
  int32_t* tile_start;
  int64_t* row_start;   // also reused as out_start by the naive path
  int64_t* ptr_copy;
  int32_t* tile_start2;  // prefix of 256-row workgroup tiles (cyclic-schedule kernel)
  int32_t* tile_start3;  // prefix of 64-row tiles (ticket kernel)
  unsigned int* tickets; // kTicketWords counters of the ticket kernel, zero before its launch
  bool any_trans = false;  // host-side note: some group reads a transposed `other`
  int64_t rows_upper = 0;  // host-side note: upper bound of the rows of the call
  bool gen_ok = false;     // host-side note: every group can run the general-shape MFMA kernel (element-aligned pointers)
  int64_t mean_k = 0;      // host-side note: row-weighted mean contraction length (tile run length of that kernel)
};

size_t workspace_bytes(int64_t B) {
  size_t n = 0;
  n += align_up(sizeof(DevGroup) * (size_t)std::max<int64_t>(B, 1), 256);
  n += align_up(sizeof(int32_t) * (size_t)(B + 1), 256);
  n += align_up(sizeof(int64_t) * (size_t)(B + 1), 256);
  n += align_up(sizeof(int64_t) * (size_t)(B + 1), 256);
  n += align_up(sizeof(int32_t) * (size_t)(B + 1), 256);
  n += align_up(sizeof(int32_t) * (size_t)(B + 1), 256);
  n += sizeof(unsigned int) * kTicketWords;
  return n;
}

Workspace carve(void* ws, int64_t B) {
  char* p = static_cast<char*>(ws);
  Workspace w;
  w.descs = reinterpret_cast<DevGroup*>(p);
  p += align_up(sizeof(DevGroup) * (size_t)std::max<int64_t>(B, 1), 256);
  w.tile_start = reinterpret_cast<int32_t*>(p);
  p += align_up(sizeof(int32_t) * (size_t)(B + 1), 256);
  w.row_start = reinterpret_cast<int64_t*>(p);
  p += align_up(sizeof(int64_t) * (size_t)(B + 1), 256);
  w.ptr_copy = reinterpret_cast<int64_t*>(p);
  p += align_up(sizeof(int64_t) * (size_t)(B + 1), 256);
  w.tile_start2 = reinterpret_cast<int32_t*>(p);
  p += align_up(sizeof(int32_t) * (size_t)(B + 1), 256);
  w.tile_start3 = reinterpret_cast<int32_t*>(p);
  p += align_up(sizeof(int32_t) * (size_t)(B + 1), 256);
  w.tickets = reinterpret_cast<unsigned int*>(p);
  return w;
}

template <typename T, int K, int MC>
int launch_mfma(const Workspace& w, int B, int M, int64_t tiles_upper, hipStream_t stream) {
  constexpr int NW = 4;
  constexpr int SZ = Elem<T>::kSize;
  constexpr int wbytes = MC * (K * SZ + 16);
  constexpr int stage = 32 * (K > MC ? K : MC) * SZ;
  constexpr int lds_v2 = wbytes + NW * stage;
  // Everything that fits streams through the LDS-staged kernel (fully coalesced HBM access); the
  // direct-fragment kernel remains for fp32 K=256, whose weight image + stages exceed 160 KB of LDS.
  constexpr bool use_v2 = lds_v2 <= 160 * 1024;
  constexpr int lds = use_v2 ? lds_v2 : wbytes;
  const void* kern;
  if constexpr (use_v2) kern = reinterpret_cast<const void*>(&mfma_rows_lds_kernel<T, K, MC, NW>);
  else kern = reinterpret_cast<const void*>(&mfma_rows_kernel<T, K, MC, NW>);
  if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
  const DeviceInfo& di = device_info();
  int per_cu = std::max(1, std::min(use_v2 ? 2 : 4, (160 * 1024) / lds));
  int flags = 3;  // nt loads + nt stores
  int chunk = 0;  // contiguous tile range per workgroup
#ifdef PYG_HIP_MM_EXPERIMENTS
  // experiment knobs (tools/mm_variants.py), experiment builds only: PYG_HIP_MM_FLAGS (bit0 nt loads, bit1 nt
  // stores), PYG_HIP_MM_CHUNK (blocked-cyclic tile schedule), PYG_HIP_MM_WGS (workgroups per CU)
  if (const char* e = getenv("PYG_HIP_MM_FLAGS")) flags = atoi(e) & 3;
  if (const char* e = getenv("PYG_HIP_MM_CHUNK")) chunk = atoi(e);
  if (const char* e = getenv("PYG_HIP_MM_WGS")) per_cu = std::max(1, atoi(e));
#endif
  int64_t gx = std::min<int64_t>(std::max<int64_t>(tiles_upper, 1), (int64_t)di.num_cus * per_cu);
  const int ncol = M / MC;
  if (ncol > 1) {
    // the column-chunk workgroups of a tile range share the X tiles: 1-D grid, XCD-aware decode in the
    // kernel (workgroup ids 8 apart = same XCD); keep the chip's resident workgroup count
    gx = std::max<int64_t>(8, (std::min<int64_t>(gx, (int64_t)di.num_cus * per_cu / ncol) + 7) / 8 * 8);
  }
#ifdef PYG_HIP_MM_EXPERIMENTS
  if constexpr (SZ == 4 && use_v2) {
    static const bool direct = getenv("PYG_HIP_MM_DIRECT") != nullptr;
    if (direct) {
      const void* dk = reinterpret_cast<const void*>(&mfma_rows_kernel<T, K, MC, NW>);
      if (int rc_ = ensure_dynamic_lds(dk, wbytes)) return rc_;
      int pc = std::max(1, std::min(4, (160 * 1024) / wbytes));
      if (const char* e = getenv("PYG_HIP_MM_WGS")) pc = std::max(1, atoi(e));
      int64_t g2 = std::min<int64_t>(std::max<int64_t>(tiles_upper, 1), (int64_t)di.num_cus * pc);
      if (ncol > 1) g2 = std::max<int64_t>(8, (std::min<int64_t>(g2, (int64_t)di.num_cus * pc / ncol) + 7) / 8 * 8);
      ProfScope prof(stream);
      hipLaunchKernelGGL((mfma_rows_kernel<T, K, MC, NW>), dim3((unsigned)(g2 * ncol)), dim3(NW * 64), wbytes, stream,
                         w.descs, w.tile_start, B, ncol);
      PYG_HIP_CHECK(hipGetLastError());
      return PYG_HIP_OK;
    }
  }
#endif
  if constexpr (SZ == 4 && K == 128 && (MC == 128 || MC == 64 || MC == 32)) {
#ifdef PYG_HIP_MM_EXPERIMENTS
    static const bool nopipe = getenv("PYG_HIP_MM_NOPIPE") != nullptr;
#else
    constexpr bool nopipe = false;
#endif
    if (!nopipe) {
      constexpr int plds = wbytes + NW * (32 * K * SZ) + NW * 4096;
      static_assert(plds <= 160 * 1024, "pipelined fp32 kernel: LDS");
      const void* pk = reinterpret_cast<const void*>(&mfma_rows_f32_pipe_kernel<K, MC, NW>);
      if (int rc_ = ensure_dynamic_lds(pk, plds)) return rc_;
      int pc = std::max(1, std::min(2, (160 * 1024) / plds));
#ifdef PYG_HIP_MM_EXPERIMENTS
      if (const char* e = getenv("PYG_HIP_MM_WGS")) pc = std::max(1, atoi(e));
#endif
      int64_t g2 = std::min<int64_t>(std::max<int64_t>(tiles_upper, 1), (int64_t)di.num_cus * pc);
      if (ncol > 1) g2 = std::max<int64_t>(8, (std::min<int64_t>(g2, (int64_t)di.num_cus * pc / ncol) + 7) / 8 * 8);
      ProfScope prof(stream);
#ifdef PYG_HIP_MM_EXPERIMENTS  // ablation / phase-counter variants: never part of the shipped library
      if constexpr (MC == 128) {
        static const int dbg = getenv("PYG_HIP_MM_DBG") ? atoi(getenv("PYG_HIP_MM_DBG")) : 0;
        if (dbg) {  // timing experiments (wrong results by construction)
          const void* dk = dbg == 1 ? (const void*)&mfma_rows_f32_pipe_kernel<K, MC, NW, 1>
                         : dbg == 2 ? (const void*)&mfma_rows_f32_pipe_kernel<K, MC, NW, 2>
                         : dbg == 3 ? (const void*)&mfma_rows_f32_pipe_kernel<K, MC, NW, 3>
                         : dbg == 4 ? (const void*)&mfma_rows_f32_pipe_kernel<K, MC, NW, 4>
                         : dbg == 8 ? (const void*)&mfma_rows_f32_pipe_kernel<K, MC, NW, 8>
                                    : (const void*)&mfma_rows_f32_pipe_kernel<K, MC, NW, 7>;
          PYG_HIP_CHECK(hipFuncSetAttribute(dk, hipFuncAttributeMaxDynamicSharedMemorySize, plds));
          void* args[] = {(void*)&w.descs, (void*)&w.tile_start, (void*)&B, (void*)&chunk, (void*)&ncol};
          PYG_HIP_CHECK(hipLaunchKernel(dk, dim3((unsigned)(g2 * ncol)), dim3(NW * 64), args, plds, stream));
          return PYG_HIP_OK;
        }
      }
#endif
      hipLaunchKernelGGL((mfma_rows_f32_pipe_kernel<K, MC, NW>), dim3((unsigned)(g2 * ncol)), dim3(NW * 64), plds, stream,
                         w.descs, w.tile_start, B, chunk, ncol);
      PYG_HIP_CHECK(hipGetLastError());
      return PYG_HIP_OK;
    }
  }
  dim3 grid((unsigned)(gx * ncol), 1, 1);
  {
    ProfScope prof(stream);
    if constexpr (use_v2) {
#ifdef PYG_HIP_MM_EXPERIMENTS
      if constexpr (K == 128 && MC == 128) {
        // the headline shape carries the experiment variants
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&mfma_rows_lds_kernel<T, K, MC, NW, 0>), lds)) return rc_;
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&mfma_rows_lds_kernel<T, K, MC, NW, 1>), lds)) return rc_;
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void*>(&mfma_rows_lds_kernel<T, K, MC, NW, 2>), lds)) return rc_;
        if (flags == 0) hipLaunchKernelGGL((mfma_rows_lds_kernel<T, K, MC, NW, 0>), grid, dim3(NW * 64), lds, stream, w.descs, w.tile_start, B, chunk, ncol);
        else if (flags == 1) hipLaunchKernelGGL((mfma_rows_lds_kernel<T, K, MC, NW, 1>), grid, dim3(NW * 64), lds, stream, w.descs, w.tile_start, B, chunk, ncol);
        else if (flags == 2) hipLaunchKernelGGL((mfma_rows_lds_kernel<T, K, MC, NW, 2>), grid, dim3(NW * 64), lds, stream, w.descs, w.tile_start, B, chunk, ncol);
        else hipLaunchKernelGGL((mfma_rows_lds_kernel<T, K, MC, NW, 3>), grid, dim3(NW * 64), lds, stream, w.descs, w.tile_start, B, chunk, ncol);
      } else
#endif
      {
        hipLaunchKernelGGL((mfma_rows_lds_kernel<T, K, MC, NW>), grid, dim3(NW * 64), lds, stream,
                           w.descs, w.tile_start, B, chunk, ncol);
      }
    } else
      hipLaunchKernelGGL((mfma_rows_kernel<T, K, MC, NW>), grid, dim3(NW * 64), lds, stream,
                         w.descs, w.tile_start, B, ncol);
  }
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

template <typename T>
int dispatch_mfma(const char* tname, const Workspace& w, int B, int K, int M, int64_t tiles_upper,
                  hipStream_t stream, bool* handled) {
  static thread_local char name[64];
  *handled = true;
  if constexpr (Elem<T>::kSize == 2) {
#ifdef PYG_HIP_MM_EXPERIMENTS
    static const bool nowide = getenv("PYG_HIP_MM_NOWIDE") != nullptr;
#else
    constexpr bool nowide = false;
#endif
    if (K == 256 && M % 256 == 0 && !nowide) {
      snprintf(name, sizeof(name), "mfma_%s_k256_wide256", tname);
      g_last_variant = name;
      constexpr int NW = 4;
      constexpr int lds = 2 * 256 * 256 + NW * 32 * 256;  // 128 KB weights + 4 x 8 KB stages = 160 KB
      const void* kern = reinterpret_cast<const void*>(&mfma_rows_wide256_kernel<T, NW>);
      if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
      const DeviceInfo& di = device_info();
      const int ncol = M / 256;
      if (ncol == 1 && (g_schedule == 0 || g_schedule == 6)) {
        // W in registers, X tiles by LDS-DMA, two four-wave workgroups per CU
        snprintf(name, sizeof(name), "mfma_%s_k256_regw", tname);
        g_last_variant = name;
        const int64_t tiles3_upper = (w.rows_upper + kPairRows - 1) / kPairRows + B;
        ProfScope prof(stream);
        return launch_ring_k256(std::is_same<T, bf16_t>::value ? PYG_BF16 : PYG_F16, w.descs, w.tile_start3, B, tiles3_upper, stream);
      }
      if (ncol == 1 && g_schedule != 1) {
        // 64 rows per wave (256-row tiles): every W fragment read from LDS feeds two MFMAs
        snprintf(name, sizeof(name), "mfma_%s_k256_wide256r2", tname);
        g_last_variant = name;
        const void* kern2 = reinterpret_cast<const void*>(&mfma_rows_wide256r2_kernel<T>);
        if (int rc_ = ensure_dynamic_lds(kern2, lds)) return rc_;
        const int64_t tiles2_upper = (w.rows_upper + 255) / 256 + B;
        const int64_t gx2 = std::min<int64_t>(std::max<int64_t>(tiles2_upper, 1), (int64_t)di.num_cus);
        ProfScope prof(stream);
        hipLaunchKernelGGL((mfma_rows_wide256r2_kernel<T>), dim3((unsigned)gx2), dim3(256), lds, stream, w.descs, w.tile_start2, B);
        PYG_HIP_CHECK(hipGetLastError());
        return PYG_HIP_OK;
      }
      int64_t gx = std::min<int64_t>(std::max<int64_t>(tiles_upper, 1), (int64_t)di.num_cus);
      if (ncol > 1) gx = std::max<int64_t>(8, (std::min<int64_t>(gx, (int64_t)di.num_cus / ncol) + 7) / 8 * 8);
      ProfScope prof(stream);
      hipLaunchKernelGGL((mfma_rows_wide256_kernel<T, NW>), dim3((unsigned)(gx * ncol)), dim3(NW * 64), lds, stream,
                         w.descs, w.tile_start, B, 0, ncol);
      PYG_HIP_CHECK(hipGetLastError());
      return PYG_HIP_OK;
    }
  }
  if constexpr (Elem<T>::kSize == 2) {
    // the headline shape (K = M = 128, 16-bit): cyclic-schedule kernel when there is enough work for every CU to
    // sweep several 256-row tiles and no group reads a transposed weight (the dX pass keeps the kernel below)
    const DeviceInfo& di = device_info();
    const int sched = g_schedule;
    const bool big = w.rows_upper >= (int64_t)di.num_cus * 256 * 4;
    if (K == 128 && M == 128 && (sched == 6 || (sched == 0 && w.rows_upper < kRingMeanRows * (int64_t)B))) {
      // many short relations: the item ring (a relation change = two ring items)
      snprintf(name, sizeof(name), "mfma_%s_k128_mc128_ring", tname);
      g_last_variant = name;
      const int64_t tiles3_upper = (w.rows_upper + kPairRows - 1) / kPairRows + B;
      ProfScope prof(stream);
      return launch_ring_k128(std::is_same<T, bf16_t>::value ? PYG_BF16 : PYG_F16, w.descs, w.tile_start3, B, tiles3_upper, stream);
    }
    if (K == 128 && M == 128 && di.num_cus >= 8 && (sched == 3 || (sched == 0 && big))) {
      snprintf(name, sizeof(name), "mfma_%s_k128_mc128_ticket", tname);
      g_last_variant = name;
      constexpr int lds = 16384 + 2 * 16384 + 64 + 256;  // W staging / epilogue scratch, 2 x 2 X stages, ticket ring, bias
      const void* kern = reinterpret_cast<const void*>(&mfma_rows_ticket_kernel<T>);
      if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
      const int64_t tiles3_upper = (w.rows_upper + kPairRows - 1) / kPairRows + B;
      int64_t gx = std::min<int64_t>(std::max<int64_t>(tiles3_upper, 8), 3 * (int64_t)di.num_cus);
      gx -= gx % 8;  // whole octets: every one of the 8 counters is served
      ProfScope prof(stream);
      hipLaunchKernelGGL((mfma_rows_ticket_kernel<T>), dim3((unsigned)gx), dim3(128), lds, stream, w.descs, w.tile_start3, B,
                         w.tickets);
      PYG_HIP_CHECK(hipGetLastError());
      return PYG_HIP_OK;
    }
    if (K == 128 && M == 128 && !w.any_trans && sched == 2) {
      snprintf(name, sizeof(name), "mfma_%s_k128_mc128_cyc", tname);
      g_last_variant = name;
      constexpr int lds = 2 * 128 * 128 * 2 + 8 * 8192;  // two W buffers + 8 stages = 128 KB
      const void* kern = reinterpret_cast<const void*>(&mfma_rows_cyc_kernel<T>);
      if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
      const int64_t tiles2_upper = (w.rows_upper + 255) / 256 + B;
      int64_t gx = std::min<int64_t>(std::max<int64_t>(tiles2_upper, 1), (int64_t)di.num_cus);
      if (gx >= 8) gx -= gx % 8;  // whole octets of workgroups: one band of tiles per XCD
      ProfScope prof(stream);
      hipLaunchKernelGGL((mfma_rows_cyc_kernel<T>), dim3((unsigned)gx), dim3(512), lds, stream, w.descs, w.tile_start2, B);
      PYG_HIP_CHECK(hipGetLastError());
      return PYG_HIP_OK;
    }
  }
  if constexpr (Elem<T>::kSize == 4) {
    if (K == 128 && M == 128 && g_f32_split && (g_schedule == 6 || (g_schedule == 0 && w.rows_upper < 512 * (int64_t)B))) {
      // split-bf16 with the W planes in registers and an LDS-DMA item ring, two four-wave workgroups per CU: a
      // relation change costs four ring items instead of a 96 KiB image built with 2-byte LDS writes -- the choice for
      // many short relations (4 Mi rows: 128 rows per relation 1.25 vs 2.52 ms, 1024 rows 1.15 vs 1.13, 16 Ki rows
      // 0.98 vs 0.94); `'ring'` forces it
      g_last_variant = "mfma_f32_k128_regw_x3";
      const int64_t tiles3_upper = (w.rows_upper + kPairRows - 1) / kPairRows + B;
      ProfScope prof(stream);
      return launch_ring_f32x3(w.descs, w.tile_start3, B, tiles3_upper, stream);
    }
    if (K == 128 && M % 128 == 0 && g_f32_split) {
      // fp32 through three bf16 planes per operand (mfma_rows_lds_kernel FLAGS bit 2): HBM-bound instead of bound by
      // the fp32 matrix rate
      g_last_variant = "mfma_f32_k128_mc128_x3";
      constexpr int NW = 4;
      constexpr int lds = 3 * 128 * 128 * 2 + NW * 32 * 128 * 4;  // 96 KB of W planes + 4 x 16 KB stages = 160 KB
      const void* kern = reinterpret_cast<const void*>(&mfma_rows_lds_kernel<T, 128, 128, NW, 7>);
      if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
      const DeviceInfo& di = device_info();
      const int ncol = M / 128;
      int64_t gx = std::min<int64_t>(std::max<int64_t>(tiles_upper, 1), (int64_t)di.num_cus);
      if (ncol > 1) gx = std::max<int64_t>(8, (std::min<int64_t>(gx, (int64_t)di.num_cus / ncol) + 7) / 8 * 8);
      ProfScope prof(stream);
#ifdef PYG_HIP_MM_EXPERIMENTS
      if (const char* e = getenv("PYG_HIP_MM_X3DBG")) {
        const int dbg = atoi(e);
        const void* dk = dbg == 8 ? (const void*)&mfma_rows_lds_kernel<T, 128, 128, NW, 15>
                       : dbg == 16 ? (const void*)&mfma_rows_lds_kernel<T, 128, 128, NW, 23>
                       : dbg == 32 ? (const void*)&mfma_rows_lds_kernel<T, 128, 128, NW, 39>
                                   : (const void*)&mfma_rows_lds_kernel<T, 128, 128, NW, 31>;
        PYG_HIP_CHECK(hipFuncSetAttribute(dk, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        int chunk0 = 0, nc = ncol;
        void* args[] = {(void*)&w.descs, (void*)&w.tile_start, (void*)&B, (void*)&chunk0, (void*)&nc};
        PYG_HIP_CHECK(hipLaunchKernel(dk, dim3((unsigned)(gx * ncol)), dim3(NW * 64), args, lds, stream));
        return PYG_HIP_OK;
      }
#endif
      hipLaunchKernelGGL((mfma_rows_lds_kernel<T, 128, 128, NW, 7>), dim3((unsigned)(gx * ncol)), dim3(NW * 64), lds, stream,
                         w.descs, w.tile_start, B, 0, ncol);
      PYG_HIP_CHECK(hipGetLastError());
      return PYG_HIP_OK;
    }
  }
  int MC = (M % 128 == 0 && K <= 256) ? 128 : (M % 64 == 0 ? 64 : 32);
  if constexpr (Elem<T>::kSize == 2) {
    // 16-bit, K <= 128: one workgroup can own 256 columns (weights + stages fit), X is read by one CU only
#ifdef PYG_HIP_MM_EXPERIMENTS
    static const bool nowide = getenv("PYG_HIP_MM_NOWIDE") != nullptr;
#else
    constexpr bool nowide = false;
#endif
    if (K == 128 && M % 256 == 0 && !nowide) MC = 256;
  }
  snprintf(name, sizeof(name), "mfma_%s_k%d_mc%d", tname, K, MC);
  g_last_variant = name;
#define PYG_CASE(KK, MM)            \
  if (K == KK && MC == MM) return launch_mfma<T, KK, MM>(w, B, M, tiles_upper, stream);
  if constexpr (Elem<T>::kSize == 2) {
    PYG_CASE(128, 256)
  }
  PYG_CASE(32, 32)
  PYG_CASE(32, 64)
  PYG_CASE(32, 128)
  PYG_CASE(64, 32)
  PYG_CASE(64, 64)
  PYG_CASE(64, 128)
  PYG_CASE(128, 32)
  PYG_CASE(128, 64)
  PYG_CASE(128, 128)
  PYG_CASE(256, 32)
  PYG_CASE(256, 64)
  PYG_CASE(256, 128)
  PYG_CASE(512, 32)
  PYG_CASE(512, 64)
#undef PYG_CASE
  *handled = false;
  return PYG_HIP_OK;
}

bool mfma_shape_ok(int dtype, int64_t K, int64_t M) {
  if (!(dtype == PYG_F32 || dtype == PYG_BF16 || dtype == PYG_F16)) return false;
  if (!(K == 32 || K == 64 || K == 128 || K == 256 || K == 512)) return false;
  if (M < 32 || M % 32 != 0 || M > (1 << 20)) return false;
  if (dtype == PYG_F32 && K == 256 && (M % 128 == 0)) {
    // fp32 K=256 x 128 columns needs 133 KB of LDS; fine on gfx950 (160 KB), 1 block/CU.
  }
  return true;
}

template <typename T, typename Acc>
int launch_naive(const Workspace& w, int B, int64_t total_upper, hipStream_t stream) {
  hipLaunchKernelGGL(out_start_kernel, dim3(1), dim3(64), 0, stream, w.descs, B, w.row_start);
  PYG_HIP_CHECK(hipGetLastError());
  int64_t blocks = std::min<int64_t>((total_upper + 255) / 256, 256 * 16);
  if (blocks < 1) blocks = 1;
  // `total` is read on device from out_start[B]; pass the host upper bound for the loop limit
  {
    ProfScope prof(stream);
    hipLaunchKernelGGL((naive_kernel<T, Acc>), dim3((unsigned)blocks), dim3(256), 0, stream,
                       w.descs, w.row_start, B, total_upper);
  }
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

int dispatch_naive(int dtype, const Workspace& w, int B, int64_t total, hipStream_t stream) {
  g_last_variant = "naive";
  switch (dtype) {
    case PYG_F32: return launch_naive<float, float>(w, B, total, stream);
    case PYG_F64: return launch_naive<double, double>(w, B, total, stream);
    case PYG_F16: return launch_naive<f16_t, float>(w, B, total, stream);
    case PYG_BF16: return launch_naive<bf16_t, float>(w, B, total, stream);
    case PYG_I8: return launch_naive<int8_t, int8_t>(w, B, total, stream);
    case PYG_U8: return launch_naive<uint8_t, uint8_t>(w, B, total, stream);
    case PYG_I16: return launch_naive<int16_t, int16_t>(w, B, total, stream);
    case PYG_I32: return launch_naive<int32_t, int32_t>(w, B, total, stream);
    case PYG_I64: return launch_naive<int64_t, int64_t>(w, B, total, stream);
    default: return fail(PYG_HIP_ERR_INVALID, "matmul: unknown dtype %d", dtype);
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int run_planned(int dtype, const Workspace& w, int B, int64_t K, int64_t M, bool uniform,
                int64_t tiles_upper, int64_t out_elems_upper, hipStream_t stream) {
  // schedules 4 / 5 (measurement only): every float shape through the general-shape kernel / the one-thread-per-output one
  if (g_schedule == 5) return dispatch_naive(dtype, w, B, out_elems_upper, stream);
  // Wide contractions go to the general-shape kernel (128-row tiles, K in 64-value chunks through a double-buffered LDS image):
  // the LDS-weight kernel keeps ALL of W's K rows in LDS -- at K = 512, and in fp32 from K = 256, that leaves one small
  // column chunk per pass and every pass re-reads X.  Measured (tools/mm_shape_sweep2.py, long / short / few segments): bf16
  // K = 512 4 - 6 x faster through the general kernel for every M, K = 256 with M not a multiple of 256 1.15 - 3 x, fp32 K >= 256
  // 1.5 - 2.9 x.  (K = 256 with M % 256 == 0 keeps the register-W / 256-column kernels; explicit schedules keep their kernels.)
  const bool wide_k = g_schedule == 0 && w.gen_ok && (K == 512 || (K == 256 && (dtype == PYG_F32 || M % 256 != 0)));
  if (uniform && mfma_shape_ok(dtype, K, M) && g_schedule != 4 && !wide_k) {
    bool handled = false;
    int rc = PYG_HIP_OK;
    if (dtype == PYG_BF16)
      rc = dispatch_mfma<bf16_t>("bf16", w, B, (int)K, (int)M, tiles_upper, stream, &handled);
    else if (dtype == PYG_F16)
      rc = dispatch_mfma<f16_t>("f16", w, B, (int)K, (int)M, tiles_upper, stream, &handled);
    else
      rc = dispatch_mfma<float>("f32", w, B, (int)K, (int)M, tiles_upper, stream, &handled);
    if (rc != PYG_HIP_OK) return rc;
    if (handled) return PYG_HIP_OK;
  }
  if (w.gen_ok && (dtype == PYG_BF16 || dtype == PYG_F16 || dtype == PYG_F32)) {
    // general shapes (per-group K / M / alignment): matmul_gen.hip
    g_last_variant = dtype == PYG_BF16 ? "mfma_bf16_gen" : dtype == PYG_F16 ? "mfma_f16_gen" : "mfma_f32_gen";
    ProfScope prof(stream);
    return launch_matmul_gen(dtype, w.descs, w.tile_start, B, tiles_upper, w.mean_k, stream);
  }
  return dispatch_naive(dtype, w, B, out_elems_upper, stream);
}

}  // namespace
}  // namespace pyg_hip

using namespace pyg_hip;

extern "C" {

size_t pyg_hip_matmul_workspace_size(int64_t num_groups) {
  return workspace_bytes(num_groups < 0 ? 0 : num_groups);
}

const char* pyg_hip_matmul_last_variant(void) { return g_last_variant; }

void pyg_hip_profile_enable(int on) {
  g_prof_on = on != 0;
  if (!g_prof_on) {
    for (auto& p : g_prof) {
      (void)hipEventDestroy(p.a);
      (void)hipEventDestroy(p.b);
    }
    g_prof.clear();
  }
}

int pyg_hip_profile_collect(float* ms_out, int capacity) {
  int n = 0;
  for (auto& p : g_prof) {
    float ms = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      if (n < capacity && ms_out) ms_out[n] = ms;
      ++n;
    }
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  g_prof.clear();
  return n;
}

int pyg_hip_segment_matmul(int dtype, const void* input, const int64_t* ptr, int ptr_on_device,
                           const void* other, const void* bias, void* out, int64_t N, int64_t K,
                           int64_t M, int64_t B, void* workspace, size_t workspace_bytes_,
                           int flags, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const size_t elt = dtype_size(dtype);
  PYG_HIP_REQUIRE(elt != 0, "segment_matmul: unknown dtype %d", dtype);
  PYG_HIP_REQUIRE(decode_mode(flags) == 0, "segment_matmul: unknown bits in 'flags' (0x%x)", flags);
  PYG_HIP_REQUIRE(N >= 0 && K >= 0 && M >= 0 && B >= 0, "segment_matmul: negative size");
  PYG_HIP_REQUIRE(ptr != nullptr, "segment_matmul: 'ptr' is NULL");
  PYG_HIP_REQUIRE(B < (1LL << 31), "segment_matmul: too many segments");
  g_last_variant = "none";
  if (B == 0 || N == 0 || M == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(out && (K == 0 || (input && other)), "segment_matmul: NULL tensor");
  PYG_HIP_REQUIRE((N + kPairRows - 1) / kPairRows + B < (1LL << 31),
                  "segment_matmul: too many row tiles");
  if (workspace_bytes_ < workspace_bytes(B) || workspace == nullptr)
    return fail(PYG_HIP_ERR_WORKSPACE, "segment_matmul: workspace of %zu bytes needed, got %zu",
                workspace_bytes(B), workspace_bytes_);
  Workspace w = carve(workspace, B);

  const int64_t* dptr = ptr;
  if (!ptr_on_device) {
    // ptr lives on the host (the reference's preferred placement): validate, then ship it.
    for (int64_t b = 0; b < B; ++b)
      PYG_HIP_REQUIRE(ptr[b + 1] >= ptr[b] && ptr[b] >= 0 && ptr[b + 1] <= N,
                      "segment_matmul: 'ptr' must be non-decreasing within [0, %lld]",
                      (long long)N);
    void* staged = nullptr;
    int rc = pinned_stage().acquire(sizeof(int64_t) * (size_t)(B + 1), &staged);
    if (rc != PYG_HIP_OK) return rc;
    ::memcpy(staged, ptr, sizeof(int64_t) * (size_t)(B + 1));
    PYG_HIP_CHECK(hipMemcpyAsync(w.ptr_copy, staged, sizeof(int64_t) * (size_t)(B + 1),
                                 hipMemcpyHostToDevice, stream));
    rc = pinned_stage().commit(stream);
    if (rc != PYG_HIP_OK) return rc;
    dptr = w.ptr_copy;
  }
  hipLaunchKernelGGL(plan_segments_kernel, dim3(1), dim3(256), 0, stream, dptr, B,
                     static_cast<const char*>(input), static_cast<const char*>(other),
                     static_cast<char*>(out), static_cast<const char*>(bias), K, M, (int)elt,
                     w.descs, w.tile_start, w.row_start, w.tile_start2, w.tile_start3, w.tickets);
  PYG_HIP_CHECK(hipGetLastError());
  w.rows_upper = N;
  if (K == 0) {
    // empty contraction: out = 0 (+ bias), handled by the generic kernel
    return dispatch_naive(dtype, w, (int)B, N * M, stream);
  }
  const int64_t tiles_upper = (N + kTileRows - 1) / kTileRows + B;
  const bool fast = aligned16(input) && aligned16(other) && aligned16(out);
  const uintptr_t all_ptrs = (uintptr_t)input | (uintptr_t)other | (uintptr_t)out | (uintptr_t)bias;
  w.gen_ok = (dtype == PYG_F32 || dtype == PYG_BF16 || dtype == PYG_F16) && all_ptrs % elt == 0 && K < (1LL << 21) &&
             M < (1LL << 21);
  w.mean_k = K;
  return run_planned(dtype, w, (int)B, K, M, fast, tiles_upper, N * M, stream);
}

int pyg_hip_grouped_matmul(int dtype, const pyg_hip_group* groups, int64_t G, void* workspace,
                           size_t workspace_bytes_, int flags, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const size_t elt = dtype_size(dtype);
  PYG_HIP_REQUIRE(elt != 0, "grouped_matmul: unknown dtype %d", dtype);
  PYG_HIP_REQUIRE(decode_mode(flags) == 0, "grouped_matmul: unknown bits in 'flags' (0x%x)", flags);
  PYG_HIP_REQUIRE(G >= 0 && G < (1LL << 31), "grouped_matmul: bad group count");
  g_last_variant = "none";
  if (G == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(groups != nullptr, "grouped_matmul: 'groups' is NULL");
  if (workspace_bytes_ < workspace_bytes(G) || workspace == nullptr)
    return fail(PYG_HIP_ERR_WORKSPACE, "grouped_matmul: workspace of %zu bytes needed, got %zu",
                workspace_bytes(G), workspace_bytes_);
  Workspace w = carve(workspace, G);

  // Host-side plan (G is small): descriptors + tile prefix, one pinned H2D copy.
  const size_t descs_b = align_up(sizeof(DevGroup) * (size_t)G, 256);
  const size_t tiles_b = align_up(sizeof(int32_t) * (size_t)(G + 1), 256);
  void* staged = nullptr;
  int rc = pinned_stage().acquire(descs_b + 3 * tiles_b, &staged);
  if (rc != PYG_HIP_OK) return rc;
  DevGroup* hd = static_cast<DevGroup*>(staged);
  int32_t* ht = reinterpret_cast<int32_t*>(static_cast<char*>(staged) + descs_b);
  int32_t* ht2 = reinterpret_cast<int32_t*>(static_cast<char*>(staged) + descs_b + tiles_b);
  int32_t* ht3 = reinterpret_cast<int32_t*>(static_cast<char*>(staged) + descs_b + 2 * tiles_b);
  int64_t tiles2 = 0, tiles3 = 0, rows_total = 0;
  bool uniform = true, any_trans = false;
  bool gen_ok = dtype == PYG_F32 || dtype == PYG_BF16 || dtype == PYG_F16;
  int64_t tiles = 0, out_elems = 0, k_rows = 0;
  for (int64_t i = 0; i < G; ++i) {
    const pyg_hip_group& gr = groups[i];
    PYG_HIP_REQUIRE(gr.rows >= 0 && gr.k >= 0 && gr.m >= 0, "grouped_matmul: negative size");
    PYG_HIP_REQUIRE(gr.rows == 0 || gr.m == 0 || (gr.out && (gr.k == 0 || (gr.input && gr.other))),
                    "grouped_matmul: NULL tensor in group %lld", (long long)i);
    hd[i].a = static_cast<const char*>(gr.input);
    hd[i].w = static_cast<const char*>(gr.other);
    hd[i].c = static_cast<char*>(gr.out);
    hd[i].bias = nullptr;
    hd[i].rows = gr.m == 0 ? 0 : gr.rows;
    hd[i].k = gr.k;
    hd[i].m = gr.m;
    hd[i].trans = gr.other_trans ? 1 : 0;
    hd[i].pad = gen_class(gr.input, gr.other, gr.out, gr.k, gr.m, (int)elt, hd[i].trans);
    if ((((uintptr_t)gr.input | (uintptr_t)gr.other | (uintptr_t)gr.out) % elt) != 0 || gr.k >= (1 << 21) ||
        gr.m >= (1 << 21))
      gen_ok = false;
    k_rows += (int64_t)gr.k * hd[i].rows;
    if (gr.k != groups[0].k || gr.m != groups[0].m) uniform = false;
    if (!aligned16(gr.input) || !aligned16(gr.other) || !aligned16(gr.out)) uniform = false;
    if (gr.other_trans) any_trans = true;
    ht[i] = (int32_t)tiles;
    ht2[i] = (int32_t)tiles2;
    ht3[i] = (int32_t)tiles3;
    tiles2 += (hd[i].rows + 2 * kTileRows - 1) / (2 * kTileRows);
    tiles3 += (hd[i].rows + kPairRows - 1) / kPairRows;
    PYG_HIP_REQUIRE(tiles3 < (1LL << 31), "grouped_matmul: too many row tiles");
    rows_total += hd[i].rows;
    tiles += (hd[i].rows + kTileRows - 1) / kTileRows;
    out_elems += hd[i].rows * gr.m;
    PYG_HIP_REQUIRE(tiles < (1LL << 31), "grouped_matmul: too many row tiles");
  }
  w.any_trans = any_trans;
  w.rows_upper = rows_total;
  w.gen_ok = gen_ok;
  w.mean_k = rows_total > 0 ? k_rows / rows_total : 0;
  ht[G] = (int32_t)tiles;
  ht2[G] = (int32_t)tiles2;
  ht3[G] = (int32_t)tiles3;
  PYG_HIP_CHECK(hipMemcpyAsync(w.descs, hd, sizeof(DevGroup) * (size_t)G, hipMemcpyHostToDevice,
                               stream));
  PYG_HIP_CHECK(hipMemcpyAsync(w.tile_start, ht, sizeof(int32_t) * (size_t)(G + 1),
                               hipMemcpyHostToDevice, stream));
  PYG_HIP_CHECK(hipMemcpyAsync(w.tile_start2, ht2, sizeof(int32_t) * (size_t)(G + 1),
                               hipMemcpyHostToDevice, stream));
  PYG_HIP_CHECK(hipMemcpyAsync(w.tile_start3, ht3, sizeof(int32_t) * (size_t)(G + 1),
                               hipMemcpyHostToDevice, stream));
  PYG_HIP_CHECK(hipMemsetAsync(w.tickets, 0, sizeof(unsigned int) * kTicketWords, stream));
  rc = pinned_stage().commit(stream);
  if (rc != PYG_HIP_OK) return rc;
  if (out_elems == 0) return PYG_HIP_OK;
  if (groups[0].k == 0) uniform = false;
  if (uniform && !mfma_shape_ok(dtype, groups[0].k, groups[0].m)) uniform = false;
  return run_planned(dtype, w, (int)G, groups[0].k, groups[0].m, uniform, tiles, out_elems,
                     stream);
}

}  // extern "C"
