// How the weight-gradient kernels (matmul_dw.hip, matmul_dw_gen.hip) hand their fp32 accumulators to memory WITHOUT
// atomics: bit-reproducible like the reference's per-relation at::matmul (ops/autograd/matmul_kernel.cpp:92-107), no
// zero fill of an accumulator image, no order-dependent float adds.
//
//   * A "unit" is what one set of accumulators belongs to: a relation (x column chunk) for the shape-specialised
//     kernels, a (group, 128 x 128 block) for the general-shape kernel.  Every workgroup walks a contiguous tile range
//     [t_beg, t_end); a unit's tiles [u0, u1) are contiguous too, so a workgroup meets every unit once, and at most its
//     FIRST unit (u0 < t_beg: "head") and its LAST one (u1 > t_end: "tail") are shared with other workgroups.
//   * When a unit ends, the four waves -- each holds the sums over ITS 32 rows of every tile -- add their accumulator
//     sets through LDS in a fixed order (dw_combine_waves: wave w produces the totals of a quarter of the registers).
//   * A unit that lies inside the range is rounded and stored straight into dW.  A head / tail partial goes to one of
//     the workgroup's two fp32 slabs in the workspace (raw register layout, 16-byte stores); dw_chain_sum, run by the
//     fix-up launch behind the main kernel, adds the slabs of a split unit in workgroup order -- tail of the workgroup
//     it starts in, then the heads of the following ones -- rounds once and stores.  The fix-up launch also writes
//     the zeros of relations without rows.
//   Result: every dW element is a fixed tree of fp32 adds for a given (input, grid); two runs give the same bits.
#pragma once

#include "matmul_common.h"

namespace pyg_hip {
namespace {

typedef float dwf4 __attribute__((ext_vector_type(4)));
constexpr int kDwBlockFloats = 1024;  // one 32 x 32 accumulator block: 16 registers x 64 lanes

__device__ __forceinline__ void dw_put(float* p, float v) { *p = v; }
__device__ __forceinline__ void dw_put(bf16_t* p, float v) { p->v = __builtin_bit_cast(uint16_t, (__bf16)v); }
__device__ __forceinline__ void dw_put(f16_t* p, float v) { p->v = __builtin_bit_cast(uint16_t, (_Float16)v); }

// Position of accumulator register r of lane `lane` of slab block bt inside the unit's output block, split into a part
// that depends on the lane only and one that depends on (bt, r) only: a store is then uniform base + scalar offset + ONE
// per-lane offset register for the whole flush (64-bit per-store addresses next to 256 live accumulators spill).
// Row-split kernels (seg_dw_kernel, dw_gen_kernel): block bt = i * JB + j is rows 32 i .., columns 32 j ..
template <int IB, int JB>
struct DwPosRows {
  static constexpr int kBlocks = IB * JB;
  __device__ static __forceinline__ void lane_part(int lane, int& row, int& col) {
    row = 4 * (lane >> 5);
    col = lane & 31;
  }
  __device__ static __forceinline__ void reg_part(int bt, int r, int& row, int& col) {
    const int i = bt / JB, j = bt - i * JB;
    row = i * 32 + (r & 3) + 8 * (r >> 2);
    col = j * 32;
  }
};
// seg_dw_wide256_kernel: wave w owns columns 64 w .. 64 w + 63 (8 x 2 blocks); slab block bt = w * 16 + i * 2 + j
struct DwPosWide {
  static constexpr int kBlocks = 64;
  __device__ static __forceinline__ void lane_part(int lane, int& row, int& col) {
    row = 4 * (lane >> 5);
    col = lane & 31;
  }
  __device__ static __forceinline__ void reg_part(int bt, int r, int& row, int& col) {
    const int w = bt >> 4, i = (bt >> 1) & 7, j = bt & 1;
    row = i * 32 + (r & 3) + 8 * (r >> 2);
    col = 64 * w + j * 32;
  }
};
// seg_dw_f32_kernel: a lane feeds VA (VB) consecutive floats of its row to VA (VB) different blocks (see the kernel)
template <int IB, int JB>
struct DwPosF32 {
  static constexpr int kBlocks = IB * JB;
  static constexpr int VA = IB >= 4 ? 4 : IB, VB = JB >= 4 ? 4 : JB;
  __device__ static __forceinline__ void lane_part(int lane, int& row, int& col) {
    row = VA * 4 * (lane >> 5);
    col = VB * (lane & 31);
  }
  __device__ static __forceinline__ void reg_part(int bt, int r, int& row, int& col) {
    const int i = bt / JB, j = bt - i * JB;
    row = 32 * VA * (i / VA) + VA * ((r & 3) + 8 * (r >> 2)) + (i % VA);
    col = 32 * VB * (j / VB) + (j % VB);
  }
};

// Sum of the four waves' accumulator sets through LDS (>= IB * JB * 4 KiB at `smem`, free of live data in EVERY wave
// once the leading barrier has been passed), handed on as it is produced.  Round Q: every wave parks its copy of
// blocks Q NB ... Q NB + NB - 1 (NB = IB JB / 4) in its LDS region; wave w then adds, for each of these blocks, the four
// copies of register quarter w (registers 4 w ... 4 w + 3) in wave order 0 + 1 + 2 + 3 and calls emit(block, w, sum).
// Every register index is a compile-time constant -- the wave only enters addresses (a wave-dependent choice among
// the 256 accumulators would send them to scratch) -- and every element is the same fixed tree of adds.
template <int IB, int JB, typename F>
__device__ __forceinline__ void dw_combine_waves(f32x16 (&acc)[IB][JB], char* smem, int wave, int lane, F&& emit) {
  constexpr int NBLK = IB * JB, NB = NBLK / 4;
  static_assert(NBLK % 4 == 0, "blocks must split over four rounds");
  __syncthreads();
#pragma unroll
  for (int Q = 0; Q < 4; ++Q) {
    char* mine = smem + wave * (NB * 4096) + lane * 16;
#pragma unroll
    for (int L = 0; L < NB; ++L) {
      const int b = Q * NB + L;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const dwf4 v = {acc[b / JB][b % JB][4 * q], acc[b / JB][b % JB][4 * q + 1], acc[b / JB][b % JB][4 * q + 2],
                        acc[b / JB][b % JB][4 * q + 3]};
        *reinterpret_cast<dwf4*>(mine + (L * 4 + q) * 1024) = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int L = 0; L < NB; ++L) {
      const char* src = smem + (L * 4 + wave) * 1024 + lane * 16;
      dwf4 s = *reinterpret_cast<const dwf4*>(src);
#pragma unroll
      for (int w = 1; w < 4; ++w) s += *reinterpret_cast<const dwf4*>(src + w * (NB * 4096));
      emit(Q * NB + L, wave, s);
    }
    __syncthreads();
  }
}

// 4 values of (block bt, register quarter q) -> the slab (raw layout: [block][quarter][lane][4]).
__device__ __forceinline__ void dw_quarter_to_slab(float* slab, int bt, int q, dwf4 s, int lane) {
  *reinterpret_cast<dwf4*>(slab + (bt * 4 + q) * 256 + lane * 4) = s;
}

// Slab of logical workgroup (bx, by): slot 0 = head partial, slot 1 = tail partial.
__device__ __forceinline__ float* dw_slab(float* slabs, int64_t slab_floats, int ncol, int bx, int by, int slot) {
  return slabs + ((int64_t)(bx * ncol + by) * 2 + slot) * slab_floats;
}

// This lane's 4 values of (block bt, quarter q) of the unit that STARTS in workgroup bx (its tail slab) and ends at tile
// u1: tail(bx) + head(bx + 1) + head(bx + 2) + ... in workgroup order; workgroups with an empty range have no slab.
__device__ __forceinline__ dwf4 dw_chain_sum(const float* slabs, int64_t slab_floats, int ncol, int bx, int by, int G,
                                             int total, int u1, int bt, int q, int lane) {
  const int64_t off = (int64_t)(bt * 4 + q) * 256 + lane * 4;
  dwf4 s = *reinterpret_cast<const dwf4*>(slabs + ((int64_t)(bx * ncol + by) * 2 + 1) * slab_floats + off);
  for (int base = bx + 1; base < G; base += 8) {
    if ((int)((int64_t)base * total / G) >= u1) break;
    dwf4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int b2 = base + k;
      v[k] = dwf4{0.0f, 0.0f, 0.0f, 0.0f};
      if (b2 < G) {
        const int tb = (int)((int64_t)b2 * total / G), te = (int)((int64_t)(b2 + 1) * total / G);
        if (tb < u1 && te > tb) v[k] = *reinterpret_cast<const dwf4*>(slabs + ((int64_t)(b2 * ncol + by) * 2) * slab_floats + off);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
  }
  return s;
}

// These 4 values -> dW, rounded once.  `base` = first element of the unit's output block (wave-uniform), `pitch` its row
// pitch in elements, (k_lim, m_lim) = rows / columns of the block that exist.  pitch * rows * sizeof(OutT) < 2^32.
template <typename Pos, typename OutT>
__device__ __forceinline__ void dw_quarter_to_out(OutT* base, int pitch, int k_lim, int m_lim, int bt, int q, dwf4 s, int lane) {
  int row_l, col_l;
  Pos::lane_part(lane, row_l, col_l);
  const uint32_t lane_off = ((uint32_t)row_l * (uint32_t)pitch + (uint32_t)col_l) * (uint32_t)sizeof(OutT);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int row_c, col_c;
    Pos::reg_part(bt, 4 * q + e, row_c, col_c);
    char* p = reinterpret_cast<char*>(base) + ((uint32_t)row_c * (uint32_t)pitch + (uint32_t)col_c) * (uint32_t)sizeof(OutT);
    if (row_c + row_l < k_lim && col_c + col_l < m_lim) dw_put(reinterpret_cast<OutT*>(p + lane_off), s[e]);
  }
}

}  // namespace
}  // namespace pyg_hip
