// Item-ring kernels of segment_matmul / grouped_matmul: a relation's W lives in REGISTERS (each wave of a four-wave
// workgroup a column slice), LDS is a ring of 16 KiB slots filled by LDS-DMA that carries X tiles and -- on a relation
// change -- the chunks of the new W, which have the shape of an X tile.  Three instances: 16-bit K = M = 256 (the C4
// shape), 16-bit K = M = 128 for many short relations, fp32 K = M = 128 through split-bf16 MFMAs.  Dispatch, tile
// tables and the other kernels: matmul.hip.
#include "matmul_common.h"

#include <algorithm>
#include <type_traits>

namespace pyg_hip {
namespace {

// Four LDS-DMA instructions of 1 KiB each (one wave's share of a 16 KiB ring item): lane l of instruction i fetches the
// 16 bytes at base + off[i] into LDS byte lds + 1024 i + 16 l.  Issued from inline asm: the compiler must not know that
// LDS is written behind its back (it cannot tell the ring slots apart and would drain vmcnt before every LDS access),
// and the waits are placed by hand (ring_wait_younger).
__device__ __forceinline__ void ring_dma4(uint32_t lds, const char* base, const uint32_t (&off)[4]) {
  uint32_t sv;
  asm volatile(
      "s_nop 4\n\t"  // base / lds may come out of v_readfirstlane: VALU-written SGPR -> VMEM address / M0
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
      "s_mov_b32 m0, %[sv]"
      : [sv] "=&s"(sv)
      : [lds] "s"(lds), [base] "s"(base), [o0] "v"(off[0]), [o1] "v"(off[1]), [o2] "v"(off[2]), [o3] "v"(off[3])
      : "memory", "scc");
}

// vmcnt retires in order: waiting until at most `younger` vector-memory operations are outstanding leaves exactly the
// ones issued after the awaited DMA in flight (4 DMA per item, 4 stores per X item: multiples of four up to 20).
__device__ __forceinline__ void ring_wait_younger(int younger) {
  if (younger == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if (younger == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (younger == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (younger == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- 16-bit, K = M = 256: W in registers, X tiles by LDS-DMA (the C4 shape) -------------------------------------------
// The two kernels above keep the 128 KiB weight matrix in LDS: one workgroup of four waves per CU, one wave per SIMD,
// X through registers (64 KiB in flight per CU) -- their waves wait for memory more than half of the time, nothing
// overlaps a wave's phases, and every MFMA needs an LDS read for W next to the one for X.  Here the roles are swapped:
//   * wave w of a four-wave workgroup owns output columns 64 w ... 64 w + 63 and keeps its slice of W (256 k x 64
//     columns = 32 KiB) as 32 MFMA A fragments in 128 registers.
//   * LDS holds a ring of two 16 KiB buffers filled by LDS-DMA (one to two items in flight while one is consumed; a
//     third slot is slower, 1.29 instead of 1.23 ms on C4: the memory side prefers few bytes in flight per CU) plus one
//     16 KiB output tile: 48.5 KiB; the 128 W registers allow TWO workgroups per CU (two waves per SIMD, from different
//     workgroups: while one multiplies, the other stores / issues / waits at its barrier.  An eight-wave workgroup
//     with 32 columns per wave ran all its waves in lock-step through its barriers: 6.7k cycles per 64 rows with the
//     matrix pipes busy 2k of them).
//   * the ring carries ITEMS: a 32-row tile of X (32 x 512 bytes), or -- when the relation changes -- one of the eight
//     32-k-row chunks of the new W[g], which has exactly the shape of an X tile.  W therefore arrives through the same
//     pipeline, fully coalesced and without draining it; when a chunk has landed every wave takes its two k-steps x two
//     column blocks out of it with ds_read_b64_tr_b16 (gfx950's transposing LDS read: 8 consecutive k of one column).
//     (A first version let every wave DMA its own slice through a private 2 KiB area: 16 dependent round trips, 24 us
//     per relation change, and the workgroups whose range crosses many small relations finished 25 % late.)  A
//     transposed `other` ([M][K]) is read from memory in fragment order directly.
//   * an X item: every wave multiplies the whole tile against its columns (16 ds_read_b128 feed 32 MFMAs), writes its
//     32 x 64 results into the output tile, and after a barrier stores 8 whole 512-byte rows of it.
//   * two barriers per item: "item i has landed" (each wave has waited for its own four DMA instructions) and "the
//     output tile is complete / ring slot i % 2 is free".  The DMA goes through inline asm with hand-placed waits:
//     every wave issues exactly 4 DMA per item and 4 stores per X item (rows behind a segment's end are clamped to its
//     last row on both sides, items behind the workgroup's range re-read its last tile into a buffer nobody consumes),
//     so the wait for item i names 4 + 4 x (X items among i - 2, i - 1) younger operations that stay in flight.
//   * 16-byte chunk c of row r lies at slot (c & 16) | ((c ^ r) & 15) of its 512-byte LDS row (permuted on the source
//     side of the DMA): conflict-free B-fragment reads; the output tile uses the same permutation.
// A workgroup owns a contiguous range of 64-row tiles (tile_start3) and walks it in 32-row halves.
template <typename T>
__global__ __launch_bounds__(256, 2) void mfma_rows_k256_regw_kernel(const DevGroup* __restrict__ descs,
                                                                     const int32_t* __restrict__ tile_start, int B) {
  constexpr int NB = 2;           // ring slots (C4: 1.23 ms with two, 1.29 ms with three -- few bytes in flight per CU)
  static_assert(NB == 2 || NB == 3, "wait_item names the younger operations of a 2- or 3-slot ring");
  constexpr int XB = 32 * 512;    // bytes per item / output tile
  typedef short v4i16 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) u32x4 GU32x4;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = (int)threadIdx.x & 63;
  const int n = lane & 31, h = lane >> 5;
  char* obuf = smem + NB * XB;

  const int total = tile_start[B];
  const int G = (int)gridDim.x;
  // consecutive workgroup ids go to consecutive XCDs: XCD k takes the k-th eighth of the ranges, so the workgroups
  // that share a relation's W (neighbours in tile order) share an L2
  const int bid = (G & 7) == 0 ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_begin = (int)((int64_t)bid * total / G);
  const int nloc = 2 * ((int)((int64_t)(bid + 1) * total / G) - t_begin);  // 32-row halves
  if (nloc <= 0) return;
  int g_first;
  {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tile_start[mid] <= t_begin) lo = mid; else hi = mid;
    }
    g_first = lo;
  }

  // ---- issue side: the item sequence is, per relation met in the range: [8 chunks of W unless transposed] [its halves]
  int ig = g_first;                              // relation of the issue front
  int iu = 0;                                    // next half-tile to issue
  int ibuf = 0;                                  // ring slot of the next item
  // fields of relation `ig` (kept in registers: every asm block clobbers "memory", so the compiler would re-read them)
  const char* i_a;
  const char* i_w;
  int64_t i_rows;
  int i_ts0, i_ts1, iw;                          // its first tile, the next relation's first tile, W chunks to issue
  auto issue_enter = [&](int g) {
    const DevGroup* p = descs + g;
    i_a = p->a;
    i_w = p->w;
    i_rows = p->rows;
    iw = p->trans ? 0 : 8;
    i_ts0 = tile_start[g];
    i_ts1 = tile_start[g + 1];
  };
  issue_enter(ig);
  auto issue_item = [&]() {
    const char* base;
    int last = 31;
    if (iw > 0) {
      base = i_w + (8 - iw) * XB;
      --iw;
    } else {
      const int uu = iu < nloc ? iu : nloc - 1;  // behind the range: the last half again (nobody consumes it)
      const int t = t_begin + (uu >> 1);
      int64_t row0 = (int64_t)(t - i_ts0) * 64 + 32 * (uu & 1);
      int64_t left = i_rows - row0;
      if (left <= 0) {  // the second half of a segment's last tile is empty: 32 times the segment's last row
        row0 = i_rows - 1;
        left = 1;
      }
      if (left < 32) last = (int)left - 1;
      base = i_a + row0 * 512;
      if (iu < nloc) {
        ++iu;
        if (iu < nloc && t_begin + (iu >> 1) >= i_ts1) {
          do ++ig; while (t_begin + (iu >> 1) >= tile_start[ig + 1]);
          issue_enter(ig);
        }
      }
    }
    const uint32_t lds = (uint32_t)(size_t)(smem + ibuf * XB + wave * 4096);
    ibuf = ibuf + 1 == NB ? 0 : ibuf + 1;
    uint32_t off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 8 * wave + 2 * i + h;
      const int c = (n & 16) | ((n ^ r) & 15);
      const int rc = r > last ? last : r;
      off[i] = (uint32_t)(rc * 512 + c * 16);
    }
    ring_dma4(lds, base, off);
  };

  // ---- consume side ----
  u32x4 wreg[16][2];
  T* bias_lds = reinterpret_cast<T*>(smem + (NB + 1) * XB);  // the relation's 256 bias values
  int cbuf = 0;                // ring slot of the item consumed next
  int s1 = 0, s2 = 0, s3 = 0;  // stores issued with the last three items (0 or 4 each)
  int consumed = 0;            // items consumed so far (the first NB were issued back to back: wait for everything)
  auto wait_item = [&]() {
    // item i has landed; younger (NB = 2): [stores of i - 2] [DMA i + 1] [stores of i - 1]; (NB = 3): one more pair
    const int younger = consumed < NB ? 0 : 4 * (NB - 1) + s1 + s2 + (NB == 3 ? s3 : 0);
    ring_wait_younger(younger);
    ++consumed;
  };
  auto retire = [&](int stores) {
    s3 = s2;
    s2 = s1;
    s1 = stores;
    cbuf = cbuf + 1 == NB ? 0 : cbuf + 1;
  };

#pragma unroll
  for (int b = 0; b < NB; ++b) issue_item();
  int gc = g_first;
  int u = 0;
  while (u < nloc) {
    const DevGroup* p = descs + gc;
    // ---- this relation's W ----
    if (p->trans) {
      // `other` stored [M][K]: 8 consecutive k of output column 64 wave + 32 cb + n are 16 contiguous bytes
      const char* wl = p->w + (64 * wave + n) * 512 + 16 * h;
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) wreg[s][cb] = *reinterpret_cast<const u32x4*>(wl + cb * 32 * 512 + 32 * s);
    } else {
      // lane (q, half16, h) of a transposing read supplies row 8 h + 4 half + (q >> 2), columns 16 half16 + 4 (q & 3) ... + 3
      // of the 16 k-rows x 32 columns of one fragment pair, and receives column 16 half16 + q
      const int q = lane & 15, half16 = (lane >> 4) & 1;
      const int ccl = 8 * wave + 2 * half16 + ((q & 3) >> 1);  // 16-byte chunk of the row (column block 0)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        wait_item();
        __syncthreads();
        const char* wb = smem + cbuf * XB;
#pragma unroll
        for (int s2k = 0; s2k < 2; ++s2k)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            v4i16 a[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              const int kk = 16 * s2k + 8 * h + 4 * half + (q >> 2);
              const int cc = ccl + 4 * cb;
              a[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(
                  wb + kk * 512 + (((cc & 16) | ((cc ^ kk) & 15)) * 16) + 8 * (q & 1)));
            }
            wreg[2 * c + s2k][cb] = __builtin_bit_cast(u32x4, __builtin_shufflevector(a[0], a[1], 0, 1, 2, 3, 4, 5, 6, 7));
          }
        // the reads must have returned before the slot is refilled
#pragma unroll
        for (int s2k = 0; s2k < 2; ++s2k) {
          asm volatile("" : "+v"(wreg[2 * c + s2k][0]));
          asm volatile("" : "+v"(wreg[2 * c + s2k][1]));
        }
        __syncthreads();
        issue_item();
        retire(0);
      }
    }
    const bool has_bias = p->bias != nullptr;
    if (has_bias) {
      __syncthreads();  // everybody is done with the previous relation's bias
      bias_lds[threadIdx.x] = reinterpret_cast<const T*>(p->bias)[threadIdx.x];
      __syncthreads();
    }
    // "use" what was loaded HERE with ordinary loads: the compiler's wait for them then sits in this (rare) path --
    // left to the first MFMA of the tile loop it becomes an s_waitcnt vmcnt(0) in every iteration, which also drains
    // the DMA and the stores the hand-placed waits leave in flight
    if (p->trans) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        asm volatile("" : "+v"(wreg[s][0]));
        asm volatile("" : "+v"(wreg[s][1]));
      }
    }
    const int64_t c_rows = p->rows;
    char* const c_out = p->c;
    const int c_ts0 = tile_start[gc];

    // ---- this relation's halves inside the range ----
    const int t_rel_end = tile_start[gc + 1];
    for (; u < nloc && t_begin + (u >> 1) < t_rel_end; ++u) {
      const int t = t_begin + (u >> 1);
      const int64_t row0 = (int64_t)(t - c_ts0) * 64 + 32 * (u & 1);
      const int64_t left = c_rows - row0;
      wait_item();
      __syncthreads();
      const char* xb = smem + cbuf * XB;
      f32x16 acc[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
      {
        // B fragments four k-steps ahead of their MFMAs (one wave of this workgroup per SIMD: nothing else hides the
        // LDS round trip of a just-in-time read)
        const char* xrow = xb + n * 512;
        const int xsw = n & 15;
        u32x4 xf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) xf[s] = *reinterpret_cast<const u32x4*>(xrow + (((2 * s + h) ^ xsw) * 16));
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const u32x4 xa = xf[s & 3];
          asm volatile("" : "+v"(xf[s & 3]));  // the wait for this fragment goes here, in front of the next read
          __builtin_amdgcn_sched_barrier(0);
          if (s + 4 < 16) xf[s & 3] = *reinterpret_cast<const u32x4*>(xrow + (((2 * (s + 4) + h) ^ xsw) * 16));
          __builtin_amdgcn_sched_barrier(0);
          acc[0] = mfma_chunk(T{}, wreg[s][0], xa, acc[0]);
          acc[1] = mfma_chunk(T{}, wreg[s][1], xa, acc[1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // results -> output tile: lane (n, h) holds columns 64 wave + 32 cb + 8 g + 4 h + (0 ... 3), g = 0 ... 3, of row n
      {
        char* orow = obuf + n * 512 + 8 * h;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            f32x2_hw v01 = {acc[cb][4 * gq], acc[cb][4 * gq + 1]}, v23 = {acc[cb][4 * gq + 2], acc[cb][4 * gq + 3]};
            u32x2 o;
            if constexpr (std::is_same<T, bf16_t>::value) {
              if (has_bias) {
                const u32x2 bb = *reinterpret_cast<const u32x2*>(bias_lds + 64 * wave + 32 * cb + 8 * gq + 4 * h);
                const f32x2_hw r01 = __builtin_convertvector(__builtin_convertvector(v01, bf16x2_hw), f32x2_hw);
                const f32x2_hw r23 = __builtin_convertvector(__builtin_convertvector(v23, bf16x2_hw), f32x2_hw);
                v01[0] = r01[0] + __builtin_bit_cast(float, bb[0] << 16);
                v01[1] = r01[1] + __builtin_bit_cast(float, bb[0] & 0xffff0000u);
                v23[0] = r23[0] + __builtin_bit_cast(float, bb[1] << 16);
                v23[1] = r23[1] + __builtin_bit_cast(float, bb[1] & 0xffff0000u);
              }
              o[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v01, bf16x2_hw));
              o[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v23, bf16x2_hw));
            } else {
              typedef _Float16 f16x2_hw __attribute__((ext_vector_type(2)));
              if (has_bias) {
                const u32x2 bb = *reinterpret_cast<const u32x2*>(bias_lds + 64 * wave + 32 * cb + 8 * gq + 4 * h);
                const uint32_t bw0 = bb[0], bw1 = bb[1];  // (scalars: __builtin_bit_cast of a vector ELEMENT reads element 0)
                const f16x2_hw b01 = __builtin_bit_cast(f16x2_hw, bw0), b23 = __builtin_bit_cast(f16x2_hw, bw1);
                v01 = __builtin_convertvector(__builtin_convertvector(v01, f16x2_hw), f32x2_hw) + __builtin_convertvector(b01, f32x2_hw);
                v23 = __builtin_convertvector(__builtin_convertvector(v23, f16x2_hw), f32x2_hw) + __builtin_convertvector(b23, f32x2_hw);
              }
              o[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v01, f16x2_hw));
              o[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v23, f16x2_hw));
            }
            *reinterpret_cast<u32x2*>(orow + (((8 * wave + 4 * cb + gq) ^ (n & 15)) * 16)) = o;
          }
      }
      __syncthreads();
      issue_item();
      // always 4 stores (the waits count them): rows behind the segment end rewrite its last row with its own data.
      // An empty second half (`left` <= 0) has multiplied the segment's last row 32 times (issue_item clamps to it)
      // and stores it again: LDS row 0, global row `last` < 0 relative to the half.
      {
        const int last = left < 32 ? (int)left - 1 : 31;
        char* cbase = c_out + row0 * 512;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 8 * wave + 2 * i + h;
          const int rg = r > last ? last : r;
          const int rl = rg < 0 ? 0 : rg;
          const u32x4 ov = *reinterpret_cast<const u32x4*>(obuf + rl * 512 + n * 16);
          const int c = (n & 16) | ((n ^ rl) & 15);
          __builtin_nontemporal_store(ov, (GU32x4*)(cbase + (int64_t)rg * 512 + c * 16));
        }
      }
      retire(4);
    }
    if (u < nloc) {
      do ++gc; while (t_begin + (u >> 1) >= tile_start[gc + 1]);
    }
  }
}

// ---- 16-bit, K = M = 128: the item ring of the kernel above for many short relations ---------------------------------
// The ticket kernel keeps the whole W in every wave's registers and refills it through a staging area with two
// workgroup barriers per relation change: 6.1 TB/s on long relations, 4.3 at 4096 rows per relation, 3.8 at 512, 1.5 at 64
// (4 Mi rows).  Here a relation change is two ring items: wave w of a four-wave workgroup keeps its 32 columns of W as 8
// fragments (32 registers), the ring carries 16 KiB items -- a 64-row X tile or one of the two 64-k-row chunks of a new
// W -- and an X item is 16 ds_read_b128 + 16 MFMAs per wave, 8 ds_write_b64 into the output tile, barrier, 4 stores of
// four whole rows each.  48.3 KiB of LDS and ~100 registers: three workgroups per CU.  Same waits as above (4 DMA per
// item, 4 stores per X item).
template <typename T>
__global__ __launch_bounds__(256, 3) void mfma_rows_k128_ring_kernel(const DevGroup* __restrict__ descs,
                                                                     const int32_t* __restrict__ tile_start, int B) {
  constexpr int NB = 2;           // ring slots
  static_assert(NB == 2 || NB == 3, "wait_item names the younger operations of a 2- or 3-slot ring");
  constexpr int XB = 64 * 256;    // bytes per item / output tile
  typedef short v4i16 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) u32x4 GU32x4;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = (int)threadIdx.x & 63;
  const int n = lane & 31, h = lane >> 5;
  char* obuf = smem + NB * XB;

  const int total = tile_start[B];
  const int G = (int)gridDim.x;
  // consecutive workgroup ids go to consecutive XCDs: XCD k takes the k-th eighth of the ranges, so the workgroups
  // that share a relation's W (neighbours in tile order) share an L2
  const int bid = (G & 7) == 0 ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_begin = (int)((int64_t)bid * total / G);
  const int nloc = (int)((int64_t)(bid + 1) * total / G) - t_begin;  // 64-row tiles
  if (nloc <= 0) return;
  int g_first;
  {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tile_start[mid] <= t_begin) lo = mid; else hi = mid;
    }
    g_first = lo;
  }

  // ---- issue side: the item sequence is, per relation met in the range: [2 chunks of W unless transposed] [its tiles]
  int ig = g_first;                              // relation of the issue front
  int iu = 0;                                    // next tile to issue
  int ibuf = 0;                                  // ring slot of the next item
  // fields of relation `ig` (kept in registers: every asm block clobbers "memory", so the compiler would re-read them)
  const char* i_a;
  const char* i_w;
  int64_t i_rows;
  int i_ts0, i_ts1, iw;                          // its first tile, the next relation's first tile, W chunks to issue
  auto issue_enter = [&](int g) {
    const DevGroup* p = descs + g;
    i_a = p->a;
    i_w = p->w;
    i_rows = p->rows;
    iw = p->trans ? 0 : 2;
    i_ts0 = tile_start[g];
    i_ts1 = tile_start[g + 1];
  };
  issue_enter(ig);
  auto issue_item = [&]() {
    const char* base;
    int last = 63;
    if (iw > 0) {
      base = i_w + (2 - iw) * XB;
      --iw;
    } else {
      const int uu = iu < nloc ? iu : nloc - 1;  // behind the range: the last tile again (nobody consumes it)
      const int t = t_begin + uu;
      const int64_t row0 = (int64_t)(t - i_ts0) * 64;
      const int64_t left = i_rows - row0;
      if (left < 64) last = (int)left - 1;
      base = i_a + row0 * 256;
      if (iu < nloc) {
        ++iu;
        if (iu < nloc && t_begin + iu >= i_ts1) {
          do ++ig; while (t_begin + iu >= tile_start[ig + 1]);
          issue_enter(ig);
        }
      }
    }
    const uint32_t lds = (uint32_t)(size_t)(smem + ibuf * XB + wave * 4096);
    ibuf = ibuf + 1 == NB ? 0 : ibuf + 1;
    uint32_t off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 16 * wave + 4 * i + (lane >> 4);
      const int c = (lane ^ r) & 15;
      const int rc = r > last ? last : r;
      off[i] = (uint32_t)(rc * 256 + c * 16);
    }
    ring_dma4(lds, base, off);
  };

  // ---- consume side ----
  u32x4 wreg[8];  // A fragments: W[64 h + 8 s ... + 7][32 wave + n] -- the k order of the other K = 128 kernels: same bits
  T* bias_lds = reinterpret_cast<T*>(smem + (NB + 1) * XB);  // the relation's 128 bias values
  int cbuf = 0;                // ring slot of the item consumed next
  int s1 = 0, s2 = 0, s3 = 0;  // stores issued with the last three items (0 or 4 each)
  int consumed = 0;            // items consumed so far (the first NB were issued back to back: wait for everything)
  auto wait_item = [&]() {
    // item i has landed; younger (NB = 2): [stores of i - 2] [DMA i + 1] [stores of i - 1]; (NB = 3): one more pair
    const int younger = consumed < NB ? 0 : 4 * (NB - 1) + s1 + s2 + (NB == 3 ? s3 : 0);
    ring_wait_younger(younger);
    ++consumed;
  };
  auto retire = [&](int stores) {
    s3 = s2;
    s2 = s1;
    s1 = stores;
    cbuf = cbuf + 1 == NB ? 0 : cbuf + 1;
  };

#pragma unroll
  for (int b = 0; b < NB; ++b) issue_item();
  int gc = g_first;
  int u = 0;
  while (u < nloc) {
    const DevGroup* p = descs + gc;
    // ---- this relation's W ----
    if (p->trans) {
      // `other` stored [M][K]: 8 consecutive k of output column 32 wave + n are 16 contiguous bytes
      const char* wl = p->w + (32 * wave + n) * 256 + 128 * h;
#pragma unroll
      for (int s = 0; s < 8; ++s) wreg[s] = *reinterpret_cast<const u32x4*>(wl + 16 * s);
    } else {
      // a chunk = 64 k-rows of W as they lie in memory (256-byte rows, chunk-swizzled like an X tile): chunk c holds the
      // k of lane half h = c.  Lane (q, half16) of a transposing read supplies row 8 s + 4 half + (q >> 2), columns
      // 16 half16 + 4 (q & 3) ... + 3 of the 8 k-rows x 32 columns of one fragment, and receives column 16 half16 + q
      const int q = lane & 15, half16 = (lane >> 4) & 1;
      const int cc = 4 * wave + 2 * half16 + ((q & 3) >> 1);  // 16-byte chunk of the row
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        wait_item();
        __syncthreads();
        const char* wb = smem + cbuf * XB;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          v4i16 a[2];
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int kk = 8 * s + 4 * half + (q >> 2);
            a[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(
                wb + kk * 256 + (((cc ^ kk) & 15) * 16) + 8 * (q & 1)));
          }
          const u32x4 frag = __builtin_bit_cast(u32x4, __builtin_shufflevector(a[0], a[1], 0, 1, 2, 3, 4, 5, 6, 7));
          if (h == c) wreg[s] = frag;
        }
        // the reads must have returned before the slot is refilled
#pragma unroll
        for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(wreg[s]));
        __syncthreads();
        issue_item();
        retire(0);
      }
    }
    const bool has_bias = p->bias != nullptr;
    if (has_bias) {
      __syncthreads();  // everybody is done with the previous relation's bias
      if (threadIdx.x < 128) bias_lds[threadIdx.x] = reinterpret_cast<const T*>(p->bias)[threadIdx.x];
      __syncthreads();
    }
    // "use" what was loaded HERE with ordinary loads: the compiler's wait for them then sits in this (rare) path --
    // left to the first MFMA of the tile loop it becomes an s_waitcnt vmcnt(0) in every iteration, which also drains
    // the DMA and the stores the hand-placed waits leave in flight
    if (p->trans) {
#pragma unroll
      for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(wreg[s]));
    }
    const int64_t c_rows = p->rows;
    char* const c_out = p->c;
    const int c_ts0 = tile_start[gc];

    // ---- this relation's tiles inside the range ----
    const int t_rel_end = tile_start[gc + 1];
    for (; u < nloc && t_begin + u < t_rel_end; ++u) {
      const int t = t_begin + u;
      const int64_t row0 = (int64_t)(t - c_ts0) * 64;
      const int64_t left = c_rows - row0;
      wait_item();
      __syncthreads();
      const char* xb = smem + cbuf * XB;
      f32x16 acc[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
      {
        // B fragments two k-steps (four reads) ahead of their MFMAs
        const char* xrow0 = xb + n * 256;
        const char* xrow1 = xb + (32 + n) * 256;
        const int xsw = n & 15;  // (32 + n) & 15 == n & 15
        u32x4 xf[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          xf[s][0] = *reinterpret_cast<const u32x4*>(xrow0 + (((8 * h + s) ^ xsw) * 16));
          xf[s][1] = *reinterpret_cast<const u32x4*>(xrow1 + (((8 * h + s) ^ xsw) * 16));
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const u32x4 xa0 = xf[s & 1][0], xa1 = xf[s & 1][1];
          asm volatile("" : "+v"(xf[s & 1][1]));  // the wait for these fragments goes here, in front of the next reads
          __builtin_amdgcn_sched_barrier(0);
          if (s + 2 < 8) {
            xf[s & 1][0] = *reinterpret_cast<const u32x4*>(xrow0 + (((8 * h + s + 2) ^ xsw) * 16));
            xf[s & 1][1] = *reinterpret_cast<const u32x4*>(xrow1 + (((8 * h + s + 2) ^ xsw) * 16));
          }
          __builtin_amdgcn_sched_barrier(0);
          acc[0] = mfma_chunk(T{}, wreg[s], xa0, acc[0]);
          acc[1] = mfma_chunk(T{}, wreg[s], xa1, acc[1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // results -> output tile: lane (n, h) holds columns 32 wave + 8 g + 4 h + (0 ... 3), g = 0 ... 3, of rows n and 32 + n
      {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          char* orow = obuf + (32 * rb + n) * 256 + 8 * h;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            f32x2_hw v01 = {acc[rb][4 * gq], acc[rb][4 * gq + 1]}, v23 = {acc[rb][4 * gq + 2], acc[rb][4 * gq + 3]};
            u32x2 o;
            if constexpr (std::is_same<T, bf16_t>::value) {
              if (has_bias) {
                const u32x2 bb = *reinterpret_cast<const u32x2*>(bias_lds + 32 * wave + 8 * gq + 4 * h);
                const f32x2_hw r01 = __builtin_convertvector(__builtin_convertvector(v01, bf16x2_hw), f32x2_hw);
                const f32x2_hw r23 = __builtin_convertvector(__builtin_convertvector(v23, bf16x2_hw), f32x2_hw);
                v01[0] = r01[0] + __builtin_bit_cast(float, bb[0] << 16);
                v01[1] = r01[1] + __builtin_bit_cast(float, bb[0] & 0xffff0000u);
                v23[0] = r23[0] + __builtin_bit_cast(float, bb[1] << 16);
                v23[1] = r23[1] + __builtin_bit_cast(float, bb[1] & 0xffff0000u);
              }
              o[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v01, bf16x2_hw));
              o[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v23, bf16x2_hw));
            } else {
              typedef _Float16 f16x2_hw __attribute__((ext_vector_type(2)));
              if (has_bias) {
                const u32x2 bb = *reinterpret_cast<const u32x2*>(bias_lds + 32 * wave + 8 * gq + 4 * h);
                const uint32_t bw0 = bb[0], bw1 = bb[1];  // (scalars: __builtin_bit_cast of a vector ELEMENT reads element 0)
                const f16x2_hw b01 = __builtin_bit_cast(f16x2_hw, bw0), b23 = __builtin_bit_cast(f16x2_hw, bw1);
                v01 = __builtin_convertvector(__builtin_convertvector(v01, f16x2_hw), f32x2_hw) + __builtin_convertvector(b01, f32x2_hw);
                v23 = __builtin_convertvector(__builtin_convertvector(v23, f16x2_hw), f32x2_hw) + __builtin_convertvector(b23, f32x2_hw);
              }
              o[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v01, f16x2_hw));
              o[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v23, f16x2_hw));
            }
            *reinterpret_cast<u32x2*>(orow + ((((4 * wave + gq) ^ n) & 15) * 16)) = o;
          }
        }
      }
      __syncthreads();
      issue_item();
      // always 4 stores (the waits count them): rows behind the segment end rewrite its last row with its own data
      {
        const int last = left < 64 ? (int)left - 1 : 63;
        char* cbase = c_out + row0 * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int r = 16 * wave + 4 * i + (lane >> 4);
          r = r > last ? last : r;
          const u32x4 ov = *reinterpret_cast<const u32x4*>(obuf + r * 256 + (lane & 15) * 16);
          const int c = (lane ^ r) & 15;
          __builtin_nontemporal_store(ov, (GU32x4*)(cbase + r * 256 + c * 16));
        }
      }
      retire(4);
    }
    if (u < nloc) {
      do ++gc; while (t_begin + u >= tile_start[gc + 1]);
    }
  }
}

// ---- fp32, K = M = 128 by split-bf16: W planes in registers, X tiles through the LDS-DMA ring ---------------------------
// The split-bf16 arithmetic of mfma_rows_lds_kernel<float, ..., FLAGS bit 2> in the structure of the kernel above (that
// kernel's 96 KiB of W planes fill the LDS: one wave per SIMD, nothing overlaps its phases).  Wave w of a four-wave
// workgroup owns output columns 32 w ... 32 w + 31 and keeps the three bf16 terms of its slice of W (128 k x 32 columns)
// as 8 x 3 MFMA A fragments in 96 registers.  The ring carries 16 KiB fp32 items: a 32-row X tile, or one of the FOUR
// 32-k-row chunks of a new relation's W (every lane reads its column's 16 values per chunk and splits them).  An X
// item: barrier, every wave splits 8 rows of the tile ONCE into three bf16 planes in LDS (round-to-nearest, 72 VALU
// instructions per wave -- splitting inside the K loop would cost every wave the whole tile: 288 next to 48 MFMAs),
// barrier, 8 K steps of 3 ds_read_b128 + 6 MFMAs, results into the fp32 output tile, barrier, 4 stores of whole rows.
// LDS: ring 2 x 16 + output 16 + planes 24 KiB + bias = 72.5 KiB: two workgroups per CU.
__global__ __launch_bounds__(256, 2) void mfma_rows_f32x3_regw_kernel(const DevGroup* __restrict__ descs,
                                                                      const int32_t* __restrict__ tile_start, int B) {
  constexpr int NB = 2;           // ring slots
  static_assert(NB == 2 || NB == 3, "wait_item names the younger operations of a 2- or 3-slot ring");
  constexpr int XB = 32 * 512;    // bytes per item / output tile
  typedef __attribute__((address_space(1))) u32x4 GU32x4;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = (int)threadIdx.x & 63;
  const int n = lane & 31, h = lane >> 5;
  char* planes = smem + NB * XB;                   // 3 x [32 rows][128 k] bf16: the current X tile, split
  float* bias_lds = reinterpret_cast<float*>(smem + NB * XB + 3 * 8192);  // the relation's 128 bias values

  const int total = tile_start[B];
  const int G = (int)gridDim.x;
  // consecutive workgroup ids go to consecutive XCDs: XCD k takes the k-th eighth of the ranges, so the workgroups
  // that share a relation's W (neighbours in tile order) share an L2
  const int bid = (G & 7) == 0 ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_begin = (int)((int64_t)bid * total / G);
  const int nloc = 2 * ((int)((int64_t)(bid + 1) * total / G) - t_begin);  // 32-row halves
  if (nloc <= 0) return;
  int g_first;
  {
    int lo = 0, hi = B;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tile_start[mid] <= t_begin) lo = mid; else hi = mid;
    }
    g_first = lo;
  }

  // ---- issue side: the item sequence is, per relation met in the range: [4 chunks of W unless transposed] [its halves]
  int ig = g_first;                              // relation of the issue front
  int iu = 0;                                    // next half-tile to issue
  int ibuf = 0;                                  // ring slot of the next item
  // fields of relation `ig` (kept in registers: every asm block clobbers "memory", so the compiler would re-read them)
  const char* i_a;
  const char* i_w;
  int64_t i_rows;
  int i_ts0, i_ts1, iw;                          // its first tile, the next relation's first tile, W chunks to issue
  auto issue_enter = [&](int g) {
    const DevGroup* p = descs + g;
    i_a = p->a;
    i_w = p->w;
    i_rows = p->rows;
    iw = p->trans ? 0 : 4;
    i_ts0 = tile_start[g];
    i_ts1 = tile_start[g + 1];
  };
  issue_enter(ig);
  auto issue_item = [&]() {
    const char* base;
    int last = 31;
    if (iw > 0) {
      base = i_w + (4 - iw) * XB;
      --iw;
    } else {
      const int uu = iu < nloc ? iu : nloc - 1;  // behind the range: the last half again (nobody consumes it)
      const int t = t_begin + (uu >> 1);
      int64_t row0 = (int64_t)(t - i_ts0) * 64 + 32 * (uu & 1);
      int64_t left = i_rows - row0;
      if (left <= 0) {  // the second half of a segment's last tile is empty: 32 times the segment's last row
        row0 = i_rows - 1;
        left = 1;
      }
      if (left < 32) last = (int)left - 1;
      base = i_a + row0 * 512;
      if (iu < nloc) {
        ++iu;
        if (iu < nloc && t_begin + (iu >> 1) >= i_ts1) {
          do ++ig; while (t_begin + (iu >> 1) >= tile_start[ig + 1]);
          issue_enter(ig);
        }
      }
    }
    const uint32_t lds = (uint32_t)(size_t)(smem + ibuf * XB + wave * 4096);
    ibuf = ibuf + 1 == NB ? 0 : ibuf + 1;
    uint32_t off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 8 * wave + 2 * i + h;
      const int c = (n & 16) | ((n ^ r) & 15);
      const int rc = r > last ? last : r;
      off[i] = (uint32_t)(rc * 512 + c * 16);
    }
    ring_dma4(lds, base, off);
  };

  // ---- consume side ----
  u32x4 wreg[8][3];  // A fragments of K step s: the hi / mid / lo bf16 terms of W[16 s + 8 h ... + 7][32 wave + n]
  int cbuf = 0;                // ring slot of the item consumed next
  int s1 = 0, s2 = 0, s3 = 0;  // stores issued with the last three items (0 or 4 each)
  int consumed = 0;            // items consumed so far (the first NB were issued back to back: wait for everything)
  auto wait_item = [&]() {
    // item i has landed; younger (NB = 2): [stores of i - 2] [DMA i + 1] [stores of i - 1]; (NB = 3): one more pair
    const int younger = consumed < NB ? 0 : 4 * (NB - 1) + s1 + s2 + (NB == 3 ? s3 : 0);
    ring_wait_younger(younger);
    ++consumed;
  };
  auto retire = [&](int stores) {
    s3 = s2;
    s2 = s1;
    s1 = stores;
    cbuf = cbuf + 1 == NB ? 0 : cbuf + 1;
  };

#pragma unroll
  for (int b = 0; b < NB; ++b) issue_item();
  int gc = g_first;
  int u = 0;
  while (u < nloc) {
    const DevGroup* p = descs + gc;
    // ---- this relation's W ----
    auto split8 = [&](const float (&f)[8], u32x4 (&o)[3]) {
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        float a0 = f[2 * pr], a1 = f[2 * pr + 1];
        o[0][pr] = split2(a0, a1);
        o[1][pr] = split2(a0, a1);
        o[2][pr] = split2(a0, a1);
      }
    };
    if (p->trans) {
      // `other` stored [M][K]: 8 consecutive k of output column 32 wave + n are 32 contiguous bytes
      const char* wl = p->w + (32 * wave + n) * 512 + 32 * h;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(wl + 64 * s), v1 = *reinterpret_cast<const f32x4*>(wl + 64 * s + 16);
        const float f[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        split8(f, wreg[s]);
      }
    } else {
      // a chunk = k-rows 32 c ... 32 c + 31 of W as they lie in memory (512-byte rows, chunk-swizzled like an X tile);
      // lane (n, h) picks column 32 wave + n of rows 16 s2 + 8 h + e
      const int cc = 8 * wave + (n >> 2);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        wait_item();
        __syncthreads();
        const char* wb = smem + cbuf * XB;
#pragma unroll
        for (int s2k = 0; s2k < 2; ++s2k) {
          float f[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int kk = 16 * s2k + 8 * h + e;
            f[e] = *reinterpret_cast<const float*>(wb + kk * 512 + (((cc & 16) | ((cc ^ kk) & 15)) * 16) + 4 * (n & 3));
          }
          split8(f, wreg[2 * c + s2k]);
        }
        // the reads must have returned before the slot is refilled
#pragma unroll
        for (int s2k = 0; s2k < 2; ++s2k)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(wreg[2 * c + s2k][pl]));
        __syncthreads();
        issue_item();
        retire(0);
      }
    }
    const bool has_bias = p->bias != nullptr;
    if (has_bias) {
      __syncthreads();  // everybody is done with the previous relation's bias
      if (threadIdx.x < 128) bias_lds[threadIdx.x] = reinterpret_cast<const float*>(p->bias)[threadIdx.x];
      __syncthreads();
    }
    // "use" what was loaded HERE with ordinary loads: the compiler's wait for them then sits in this (rare) path --
    // left to the first MFMA of the tile loop it becomes an s_waitcnt vmcnt(0) in every iteration, which also drains
    // the DMA and the stores the hand-placed waits leave in flight
    if (p->trans) {
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(wreg[s][pl]));
    }
    const int64_t c_rows = p->rows;
    char* const c_out = p->c;
    const int c_ts0 = tile_start[gc];

    // ---- this relation's halves inside the range ----
    const int t_rel_end = tile_start[gc + 1];
    for (; u < nloc && t_begin + (u >> 1) < t_rel_end; ++u) {
      const int t = t_begin + (u >> 1);
      const int64_t row0 = (int64_t)(t - c_ts0) * 64 + 32 * (u & 1);
      const int64_t left = c_rows - row0;
      wait_item();
      // no barrier here: a wave splits the 8 rows it has loaded itself, and nobody reads the planes any more (the
      // previous item's last barrier lies behind everybody's K loop)
      char* const xb = smem + cbuf * XB;
      char* const obuf = xb;  // the fp32 tile is dead once it is split (barrier below): its slot takes the results
      // split this wave's 8 rows of the tile into the three bf16 planes: lane l takes 16 consecutive floats of row
      // 8 wave + (l >> 3) (two 16-byte bf16 chunks per plane; chunk c of row r lies at slot c ^ (r & 15) of its 256 bytes)
      {
        const int r = 8 * wave + (lane >> 3);
        const int c0 = 4 * (lane & 7);
        float f[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cc = c0 + j;
          const f32x4 v = *reinterpret_cast<const f32x4*>(xb + r * 512 + (((cc & 16) | ((cc ^ r) & 15)) * 16));
#pragma unroll
          for (int e = 0; e < 4; ++e) f[4 * j + e] = v[e];
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          u32x4 o[3];
#pragma unroll
          for (int pr = 0; pr < 4; ++pr) {
            float a0 = f[8 * half + 2 * pr], a1 = f[8 * half + 2 * pr + 1];
            o[0][pr] = split2(a0, a1);
            o[1][pr] = split2(a0, a1);
            o[2][pr] = split2(a0, a1);
          }
          const int bc = 2 * (lane & 7) + half;  // bf16 chunk of the row
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<u32x4*>(planes + pl * 8192 + r * 256 + ((bc ^ (r & 15)) * 16)) = o[pl];
        }
      }
      __syncthreads();
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      {
        // B fragments two K steps ahead of their MFMAs
        const char* prow = planes + n * 256;
        const int psw = n & 15;
        u32x4 xf[2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) xf[s][pl] = *reinterpret_cast<const u32x4*>(prow + pl * 8192 + (((2 * s + h) ^ psw) * 16));
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const u32x4 xh = xf[s & 1][0], xm = xf[s & 1][1], xl = xf[s & 1][2];
          asm volatile("" : "+v"(xf[s & 1][2]));  // the wait for these fragments goes here, in front of the next reads
          __builtin_amdgcn_sched_barrier(0);
          if (s + 2 < 8) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              xf[s & 1][pl] = *reinterpret_cast<const u32x4*>(prow + pl * 8192 + (((2 * (s + 2) + h) ^ psw) * 16));
          }
          __builtin_amdgcn_sched_barrier(0);
          // smallest terms first
          acc = mfma_chunk(bf16_t{}, wreg[s][2], xh, acc);
          acc = mfma_chunk(bf16_t{}, wreg[s][0], xl, acc);
          acc = mfma_chunk(bf16_t{}, wreg[s][1], xm, acc);
          acc = mfma_chunk(bf16_t{}, wreg[s][1], xh, acc);
          acc = mfma_chunk(bf16_t{}, wreg[s][0], xm, acc);
          acc = mfma_chunk(bf16_t{}, wreg[s][0], xh, acc);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // results -> output tile: lane (n, h) holds columns 32 wave + 8 g + 4 h + (0 ... 3), g = 0 ... 3, of row n
      {
        char* orow = obuf + n * 512;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          f32x4 v = {acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]};
          if (has_bias) v += *reinterpret_cast<const f32x4*>(bias_lds + 32 * wave + 8 * gq + 4 * h);
          const int oc = 8 * wave + 2 * gq + h;  // 16-byte chunk of the output row
          *reinterpret_cast<f32x4*>(orow + (((oc & 16) | ((oc ^ n) & 15)) * 16)) = v;
        }
      }
      __syncthreads();
      // always 4 stores (the waits count them): rows behind the segment end rewrite its last row with its own data.  An
      // empty second half (`left` <= 0) has multiplied the segment's last row 32 times (issue_item clamps to it) and
      // stores it again: LDS row 0, global row `last` < 0 relative to the half.  The results are read out of the ring
      // slot BEFORE its refill is issued.
      {
        const int last = left < 32 ? (int)left - 1 : 31;
        char* cbase = c_out + row0 * 512;
        u32x4 ov[4];
        int64_t goff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 8 * wave + 2 * i + h;
          const int rg = r > last ? last : r;
          const int rl = rg < 0 ? 0 : rg;
          ov[i] = *reinterpret_cast<const u32x4*>(obuf + rl * 512 + n * 16);
          const int c = (n & 16) | ((n ^ rl) & 15);
          goff[i] = (int64_t)rg * 512 + c * 16;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(ov[i]));  // the reads have returned
        // a whole tile: a wave's DMA writes exactly the 8 rows it has just read.  A partial tile clamps rows beyond the
        // segment end to row `last` (or 0), which lies in ANOTHER wave's DMA region: every wave's reads must have
        // returned before anybody refills (`left` is uniform over the workgroup)
        if (left < 32) __syncthreads();
        issue_item();
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_nontemporal_store(ov[i], (GU32x4*)(cbase + goff[i]));
      }
      retire(4);
    }
    if (u < nloc) {
      do ++gc; while (t_begin + (u >> 1) >= tile_start[gc + 1]);
    }
  }
}


template <typename T>
int launch_k256(const DevGroup* descs, const int32_t* tile_start3, int B, int64_t tiles3_upper, hipStream_t stream) {
  constexpr int lds = 3 * 32 * 512 + 512;  // a ring of two 16 KiB items + the output tile + the bias row = 48.5 KiB
  const void* kern = reinterpret_cast<const void*>(&mfma_rows_k256_regw_kernel<T>);
  if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
  const int64_t gx = std::min<int64_t>(std::max<int64_t>(tiles3_upper, 1), 2 * (int64_t)device_info().num_cus);
  hipLaunchKernelGGL((mfma_rows_k256_regw_kernel<T>), dim3((unsigned)gx), dim3(256), lds, stream, descs, tile_start3, B);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

template <typename T>
int launch_k128(const DevGroup* descs, const int32_t* tile_start3, int B, int64_t tiles3_upper, hipStream_t stream) {
  constexpr int lds = 3 * 64 * 256 + 256;  // ring of two items + the output tile + the bias row = 48.3 KiB
  const void* kern = reinterpret_cast<const void*>(&mfma_rows_k128_ring_kernel<T>);
  if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
  const int64_t gx = std::min<int64_t>(std::max<int64_t>(tiles3_upper, 1), 3 * (int64_t)device_info().num_cus);
  hipLaunchKernelGGL((mfma_rows_k128_ring_kernel<T>), dim3((unsigned)gx), dim3(256), lds, stream, descs, tile_start3, B);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

}  // namespace

int launch_ring_k256(int dtype, const void* descs, const int32_t* tile_start3, int B, int64_t tiles3_upper, hipStream_t stream) {
  const DevGroup* d = static_cast<const DevGroup*>(descs);
  return dtype == PYG_BF16 ? launch_k256<bf16_t>(d, tile_start3, B, tiles3_upper, stream)
                           : launch_k256<f16_t>(d, tile_start3, B, tiles3_upper, stream);
}

int launch_ring_k128(int dtype, const void* descs, const int32_t* tile_start3, int B, int64_t tiles3_upper, hipStream_t stream) {
  const DevGroup* d = static_cast<const DevGroup*>(descs);
  return dtype == PYG_BF16 ? launch_k128<bf16_t>(d, tile_start3, B, tiles3_upper, stream)
                           : launch_k128<f16_t>(d, tile_start3, B, tiles3_upper, stream);
}

int launch_ring_f32x3(const void* descs, const int32_t* tile_start3, int B, int64_t tiles3_upper, hipStream_t stream) {
  constexpr int lds = 2 * 32 * 512 + 3 * 8192 + 512;  // ring of two fp32 items + three bf16 planes + the bias row
  const void* kern = reinterpret_cast<const void*>(&mfma_rows_f32x3_regw_kernel);
  if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
  const int64_t gx = std::min<int64_t>(std::max<int64_t>(tiles3_upper, 1), 2 * (int64_t)device_info().num_cus);
  hipLaunchKernelGGL(mfma_rows_f32x3_regw_kernel, dim3((unsigned)gx), dim3(256), lds, stream,
                     static_cast<const DevGroup*>(descs), tile_start3, B);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

}  // namespace pyg_hip
