// Fused relational graph convolution over a sampled neighbourhood (BASELINE.json configs[4]):
//
//     out[scatter_index[e] + scatter_offset_r] += x[gather_index[e] + gather_offset_r] @ W_r      for every edge e of
//                                                                                                 every relation r
//
// i.e. the reference-op chain  gather_coo (ops/cuda/segment_coo_kernel.cu:1316-1360) -> segment_matmul
// (ops/cuda/matmul_kernel.cu:304-319) -> scatter_sum (ops/cuda/scatter_kernel.cu:56-71)  in ONE launch: neither the
// gathered features [E, K] nor the messages [E, M] ever exist in HBM, and the per-relation index vectors are read
// where the sampler left them (no concatenation).
//
// A persistent grid of four-wave workgroups; a tile is 128 edges of one relation (32 per wave), a workgroup walks a
// contiguous range of tiles (rgcn_fused_kernel below: the index / row pipeline):
//   * A-operand rows are GATHERED: lane l of load i fetches 16 bytes of row gather(edge (64 i + l) / 16) -- a whole
//     256-byte row per 16 lanes -- into registers one tile ahead, and parks them in the same XOR-swizzled per-wave
//     stage the segment_matmul kernels use;
//   * W_r is copied in its native [K][M] layout by LDS-DMA and read through ds_read_b64_tr_b16 (see
//     mfma_rows_cyc_kernel in matmul.hip for the block permutation that keeps those reads conflict free);
//   * epilogue = SCATTER: the 32 x 128 messages of a wave are rounded to the storage type (exactly what the
//     unfused chain materialises), parked in the stage, and every lane walks its two columns down the 32 rows,
//     summing runs of equal destination in fp32 -- the sampler emits a relation's edges grouped by source node, so
//     runs are ~fan-out long and run boundaries are the same for all lanes (wave-uniform branch) -- and flushes
//     each run with one packed atomic per lane (global_atomic_pk_add_bf16 / _f16: 64 lanes = one whole 256-byte
//     output row per instruction).  ~E / fan-out row atomics instead of E, no index sort, no second pass.
// Rounding: messages are rounded once to T (as in the unfused chain), run sums are exact fp32 and rounded once per
// flush -- at least as accurate as the reference's element-by-element rounding (ops/cpu/scatter_kernel.cpp:82-110).
#include "common.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <type_traits>
#include <vector>

namespace pyg_hip {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void LDSV;
// global_* instructions: a flat access also counts on lgkmcnt and would make every LDS wait conservative
typedef const __attribute__((address_space(1))) int64_t GI64;
typedef const __attribute__((address_space(1))) u32x4 GU32x4;

struct RelDev {
  const int64_t* gather_index;
  const int64_t* scatter_index;
  const char* weight;
  int64_t num_edges;
  int64_t gather_offset;
  int64_t scatter_offset;
  const char* x;              // feature rows this relation gathers from (the call's x, or the source type's own table)
  const int64_t* gather_map;  // nullptr, or: row = gather_map[gather_index[e]] (sampled local id -> global node id)
  int64_t x_rows;             // rows of `x` (checked mode)
  int64_t map_len;            // entries of gather_map (checked mode)
};

template <bool BF16>
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 acc) {
  if constexpr (BF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}

typedef float f32x2_v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_v __attribute__((ext_vector_type(2)));

template <bool BF16>
__device__ __forceinline__ uint32_t pack2(float a, float b) {  // one v_cvt_pk_* (element-wise casts cost four instructions)
  const f32x2_v v = {a, b};
  if constexpr (BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_v));
  else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_v));
}

template <bool BF16>
__device__ __forceinline__ void unpack2(uint32_t v, float* a, float* b) {
  if constexpr (BF16) {
    *a = __builtin_bit_cast(float, v << 16);
    *b = __builtin_bit_cast(float, v & 0xffff0000u);
  } else {
    *a = (float)__builtin_bit_cast(_Float16, (uint16_t)(v & 0xffffu));
    *b = (float)__builtin_bit_cast(_Float16, (uint16_t)(v >> 16));
  }
}

// out[addr] += (a, b) as one packed 16-bit pair, no value returned.  (The builtins, not inline asm: an asm with a memory
// clobber made the compiler re-read the relation records -- scalar loads with their waits -- behind every flush, and it
// cannot take the scalar-base addressing form.)
template <bool BF16>
__device__ __forceinline__ void atomic_add_pk(char* addr, float a, float b) {
  typedef short s16x2_v __attribute__((ext_vector_type(2)));
  const f32x2_v v = {a, b};
  if constexpr (BF16) {
    typedef __attribute__((address_space(1))) bf16x2_v GB;
    (void)__builtin_amdgcn_global_atomic_fadd_v2bf16((GB*)addr, __builtin_bit_cast(s16x2_v, __builtin_convertvector(v, bf16x2_v)));
  } else {
    typedef __attribute__((address_space(1))) f16x2_v GH;
    (void)__builtin_amdgcn_global_atomic_fadd_v2f16((GH*)addr, __builtin_convertvector(v, f16x2_v));
  }
}

template <bool BF16>
__device__ __forceinline__ void atomic_add_pk_bits(char* addr, uint32_t packed) {  // the same, operand already packed
  typedef short s16x2_v __attribute__((ext_vector_type(2)));
  if constexpr (BF16) {
    typedef __attribute__((address_space(1))) bf16x2_v GB;
    (void)__builtin_amdgcn_global_atomic_fadd_v2bf16((GB*)addr, __builtin_bit_cast(s16x2_v, packed));
  } else {
    typedef __attribute__((address_space(1))) f16x2_v GH;
    (void)__builtin_amdgcn_global_atomic_fadd_v2f16((GH*)addr, __builtin_bit_cast(f16x2_v, packed));
  }
}

// The same add as a compare-and-swap loop on the pair's word (PYG_HIP_RGCN_CAS / PYG_HIP_FLOAT_ATOMICS=cas: the diagnostic
// flavour that does not use the hardware's floating-point atomic unit).
template <bool BF16>
__device__ __forceinline__ void cas_add_pk_bits(char* addr, uint32_t packed) {
  unsigned int* p = reinterpret_cast<unsigned int*>(addr);
  float a, b;
  unpack2<BF16>(packed, &a, &b);
  unsigned int old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (true) {
    float c, d;
    unpack2<BF16>(old, &c, &d);
    const unsigned int want = pack2<BF16>(c + a, d + b);
    if (__hip_atomic_compare_exchange_strong(p, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  }
}

typedef uint32_t u32x32 __attribute__((ext_vector_type(32)));

// K = M = 128, 16-bit T
// CHECK: every gather / scatter index is validated; an offender sets *error and is redirected to row 0 (the host
// reports it): without the check a bad index is an out-of-bounds read or an atomic into foreign memory.
//
// Persistent launch: workgroup b owns the contiguous tile range [b T / G, (b + 1) T / G) of the tiles of all relations
// (tile_start = prefix of ceil(edges_r / 128)); W_r is copied into LDS when the range enters relation r (a handful of
// workgroups change relation at all).  A tile needs three DEPENDENT memory round trips before its MFMAs -- edge ->
// gather_index -> (gather_map) -> feature row, ~2.5 - 5 us each on this chip whatever its load -- so every wave runs a
// three-deep software pipeline over its tiles:
//     tile t + 2 : first level   (gather_index / scatter_index of the wave's 32 edges, coalesced)
//     tile t + 1 : second level  (gather_map lookup), then its 32 feature rows travel to REGISTERS (8 x 16 bytes per lane)
//     tile t     : rows -> LDS stage (the XOR-swizzled layout of the segment_matmul kernels), MFMAs, scatter
// and pays the round trips once per workgroup instead of three per tile.  (The first version issued one tile at a time by
// LDS-DMA: 51 % of a wave's time waiting for rows, 30 % for indices; phase clocks of an experiment build.)
// Where the time goes now (C5 batch, 565 k edges, ablations of an experiment build through `dbg`; operator us, the
// kernel alone is 57.6 under rocprofv3): everything 59; no atomics 53; no row gathers (compute on stale registers) 46 - 49; gathers and
// indices only 33; indices only 15.  The compute side is the larger one and is made of LDS traffic (64 KB per wave and
// tile: 32 KB of W fragments, 8 + 8 KB of X, 8 + 8 KB of messages -- 128 bytes per clock and CU = 1.9 us per round of
// eight wave-tiles), ~330 VALU instructions per wave-tile (4 clocks each: unpack / select / add / round of the scatter
// walk 230 of them, accumulator reads and packing 100) and 32 MFMAs; one MFMA k-step instead of eight saves 10 us, no
// scatter walk 18 - 20 us.
// BIG: some feature table is 4 GB or more (row offsets need 64 bits: two lane exchanges and 64-bit adds per 16-byte load
// instead of one exchange and a scalar-base load)
// The relation records and the tile prefix of a call.  Up to kRgcnInline relations travel IN THE KERNEL ARGUMENT (INL):
// staging them through pinned memory put a host-to-device copy (a 4 us blit kernel + 6 us of gap on the stream) in front
// of every launch of a 50 us kernel; longer lists are copied to the workspace as before.
constexpr int kRgcnInline = 24;
struct RgcnDesc {
  const RelDev* rels;
  const int32_t* tile_start;
  RelDev irels[kRgcnInline];
  int32_t itile[kRgcnInline + 1];
};
static_assert(sizeof(RgcnDesc) <= 3072, "kernel argument");

template <bool BF16, bool CHECK, bool BIG, bool INL>
__global__ __launch_bounds__(256) void rgcn_fused_kernel(const RgcnDesc desc, int R, char* __restrict__ out, int64_t out_rows,
                                                        int* __restrict__ error, int dbg) {
  auto tile_at = [&](int i) -> int {
    if constexpr (INL) return desc.itile[i];
    else return desc.tile_start[i];
  };
  auto rel_at = [&](int i) -> RelDev {
    if constexpr (INL) return desc.irels[i];
    else return desc.rels[i];
  };
  // dbg bit 5 (32): the packed adds as CAS loops (PYG_HIP_RGCN_CAS).  The other bits are 0 in the product
  // (PYG_HIP_RGCN_DBG of an experiment build): 1 no atomics, 2 no row gathers, 4 no MFMAs / scatter,
  // 8 no scatter walk, 16 one MFMA k-step instead of eight
  constexpr int NT = 4, NI = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, xl = lane & 31, h = lane >> 5;
  // (through readfirstlane: the compiler then knows that a tile's row count, relation and run structure are wave-uniform
  // and keeps them -- and the branches on them -- on the scalar unit)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = tile_at(R);
  const int G = (int)gridDim.x;
  const int t_beg = (int)((int64_t)blockIdx.x * total / G);
  const int t_end = (int)((int64_t)(blockIdx.x + 1) * total / G);
  if (t_beg >= t_end) return;
  char* xs = smem + 32768 + wave * 8192;

  // ---- index pipeline -------------------------------------------------------------------------------------------
  // What a stage needs from its relation's record travels with the tile (scalar registers): the record itself is read
  // when the walk enters a relation, not per tile (scalar loads + their waits, three stages per tile).
  struct Idx {
    int rel;      // relation of the tile (uniform)
    int nrows;    // edges of this wave in the tile (uniform; 0: nothing to do)
    int64_t si;   // lane l < 32: output row of edge l (set once its load has landed)
    const char* x;              // the relation's feature table, its row count, map and offset (uniform)
    const int64_t* gmap;
    int64_t goff, x_rows, map_len, soff;  // (soff: added to the loaded scatter index where it is consumed)
  };
  int walker;  // relation of the tile the first level is at (tiles ascend)
  {
    int lo = 0, hi = R;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tile_at(mid) <= t_beg) lo = mid; else hi = mid;
    }
    walker = lo;
  }
  RelDev wrel = rel_at(walker);
  int w_first = tile_at(walker), w_next = tile_at(walker + 1);
  // first level of tile t: the tile's scalars into `o`, the two index loads into g1 / s1 (lane l < 32: gather index and
  // output row of edge l).  What a load returns is NOT touched before the top of the next iteration (see the loop): the
  // compiler orders a use behind everything issued before it (vmcnt counts in order and it cannot see across the loop's
  // branches how many loads are younger), so a use in mid-iteration waits for the row gathers issued a moment earlier
  // -- which is what the first version of this pipeline did: 56 % of the wave cycles waiting (SQ_WAIT_ANY).
  auto first_level = [&](int t, Idx& o, int64_t& g1, int64_t& s1) {
    o.rel = walker;
    o.nrows = 0;
    o.si = 0;
    o.x = nullptr;
    o.gmap = nullptr;
    o.goff = o.x_rows = o.map_len = o.soff = 0;
    g1 = 0;
    s1 = 0;
    if (t >= t_end) return;
    while (t >= w_next) {
      ++walker;
      wrel = rel_at(walker);
      w_first = w_next;
      w_next = tile_at(walker + 1);
    }
    o.rel = walker;
    o.x = wrel.x;
    o.gmap = wrel.gather_map;
    o.goff = wrel.gather_offset;
    o.x_rows = wrel.x_rows;
    o.map_len = wrel.map_len;
    o.soff = wrel.scatter_offset;
    const int64_t e0 = (int64_t)(t - w_first) * 128 + wave * 32;   // first edge of this wave inside the relation
    const int64_t left = wrel.num_edges - e0;
    o.nrows = left >= 32 ? 32 : (left > 0 ? (int)left : 0);
    if (o.nrows > 0) {  // rows past the end repeat the last valid edge
      const uint32_t el = (uint32_t)(xl < o.nrows ? xl : o.nrows - 1);
      g1 = ((GI64*)(wrel.gather_index + e0))[el];
      s1 = ((GI64*)(wrel.scatter_index + e0))[el];
    }
  };
  // second level: the feature row of the lane's edge -- g1 + offset, or (a load) gather_map[g1]
  auto second_level = [&](const Idx& io, int64_t g1) -> int64_t {
    if (io.nrows == 0) return 0;
    if (!io.gmap) return g1 + io.goff;
    if (CHECK && (g1 < 0 || g1 >= io.map_len)) {  // double indirection done here: the gathered feature matrix never exists
      *error = 1;
      g1 = 0;
    }
    return ((GI64*)io.gmap)[g1];
  };
  // the 32 feature rows of the next tile on their way: lane's chunk i is slot p = 64 i + lane of the stage.  (Two
  // tiles ahead -- a second register set -- was no faster: 82 vs 79 us on the C5 batch; what is left is not latency.)
  u32x4 xr[NI];
  uint32_t coff[NI];  // byte offset of the lane's chunk of load i inside its row
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int p = i * 64 + lane;
    const int r = p >> 4, cs = p & 15;
    coff[i] = (uint32_t)((cs ^ (r & 15)) * 16);
  }
  auto issue_rows = [&](Idx& b, int64_t g2) {
    if (b.nrows == 0 || (dbg & 2)) return;  // (dbg: timing ablations of an experiment build, 0 in the product)
    if (CHECK) {
      if (g2 < 0 || g2 >= b.x_rows) {
        *error = 1;
        g2 = 0;
      }
      if (b.si < 0 || b.si >= out_rows) {
        *error = 2;
        b.si = 0;
      }
    }
    if constexpr (BIG) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int64_t row = __shfl(g2, 4 * i + (lane >> 4));
        xr[i] = *(GU32x4*)(b.x + row * 256 + coff[i]);
      }
    } else {
      const int rowb = (int)((uint32_t)g2 << 8);  // < 4 GB tables (checked on the host)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const uint32_t off = (uint32_t)__shfl(rowb, 4 * i + (lane >> 4)) + coff[i];
        xr[i] = *(GU32x4*)(b.x + off);
      }
    }
  };

  // W of relation `g`: this wave's 8 blocks of 4 k-rows, 16-byte chunks permuted inside a block
  auto load_w = [&](int g) {
    const int dma_r = (lane & 15) >> 2, dma_ii = lane & 3, dma_u = lane >> 4;
    const int dma_c = 2 * dma_u + (dma_ii & 1) + 8 * (dma_ii >> 1);
    const char* wsrc = rel_at(g).weight + dma_r * 256 + dma_c * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kb = wave * 8 + j;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + kb * 1024),
                                       (LDSV*)(smem + kb * 1024), 16, 0, 0);
    }
  };

  // prologue: W of the first relation and the three index chains of tiles t, t + 1, t + 2 are requested together
  int w_rel = walker;
  load_w(w_rel);
  Idx C, Bx, A;
  int64_t pend_g2, pend_g1, pend_si;  // in flight across an iteration: row index of Bx's edges, first level of A
  {
    int64_t g1c, g1b, sc, sb;
    first_level(t_beg, C, g1c, sc);
    first_level(t_beg + 1, Bx, g1b, sb);
    first_level(t_beg + 2, A, pend_g1, pend_si);
    C.si = sc + C.soff;
    Bx.si = sb + Bx.soff;
    const int64_t g2c = second_level(C, g1c);
    pend_g2 = second_level(Bx, g1b);
    issue_rows(C, g2c);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (W's DMA is invisible to the compiler's counters)
  __syncthreads();

  const int q = lane & 15, grp16 = lane >> 4;
  const char* wb = smem + 16384 * h + (4 * (q >> 2) + (grp16 & 1) + 2 * (q & 1)) * 16 + ((q & 3) >> 1) * 8;
  const int cch = lane >> 2, cdw = lane & 3;
  for (int t = t_beg; t < t_end; ++t) {
    if (C.rel != w_rel) {  // uniform over the workgroup: every wave walks the same tiles
      __syncthreads();     // everybody is done with the old W
      w_rel = C.rel;
      load_w(w_rel);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const int nrows = C.nrows;
    const int64_t si = C.si;
    if (nrows > 0) {
#pragma unroll
      for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(xs + (i * 64 + lane) * 16) = xr[i];
    }
    // advance the pipeline.  Everything outstanding here was issued an iteration ago: FIRST every use of it (the rows
    // above, the row index of tile t + 1, the first level of t + 2), THEN this iteration's loads -- second level of
    // t + 2, first level of t + 3, rows of t + 1 -- whose results rest until the next iteration's top.
    const int64_t g2b = pend_g2;
    A.si = pend_si + A.soff;
    pend_g2 = second_level(A, pend_g1);
    Idx Nx;
    first_level(t + 3, Nx, pend_g1, pend_si);
    issue_rows(Bx, g2b);
    C = Bx;
    Bx = A;
    A = Nx;
    if (nrows == 0 || (dbg & 4)) continue;
    f32x16 acc[NT];
#pragma unroll
    for (int s = 0; s < ((dbg & 16) ? 1 : NI); ++s) {
      const u32x4 xa = *reinterpret_cast<const u32x4*>(xs + (xl * 16 + ((NI * h + s) ^ (xl & 15))) * 16);
      u32x4 wa[NT];
#pragma unroll
      for (int t4 = 0; t4 < NT; ++t4) {
        const v4i16 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(wb + (2 * s) * 1024 + t4 * 256));
        const v4i16 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(wb + (2 * s + 1) * 1024 + t4 * 256));
        wa[t4] = __builtin_bit_cast(u32x4, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int t4 = 0; t4 < NT; ++t4) {
        // (the first step takes the constant 0 as its C operand: no 64 moves to clear the accumulators per tile)
        f32x16 c0;
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[r] = 0.f;
        acc[t4] = mfma16<BF16>(wa[t4], xa, s == 0 ? c0 : acc[t4]);
      }
    }
    // messages, rounded to T, into the stage (row xl, 16-byte chunks XOR-swizzled with the row)
#pragma unroll
    for (int t4 = 0; t4 < NT; ++t4) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        u32x4 pk;
#pragma unroll
        for (int i = 0; i < 4; ++i) pk[i] = pack2<BF16>(acc[t4][8 * j + 2 * i], acc[t4][8 * j + 2 * i + 1]);
        const int c = 8 * h + 2 * t4 + j;
        *reinterpret_cast<u32x4*>(xs + (xl * 16 + (c ^ (xl & 15))) * 16) = pk;
      }
    }
    // scatter: lane owns columns 2*lane, 2*lane + 1 (chunk lane / 4, dword lane % 4) of every row; runs of equal
    // destination are summed in fp32 and flushed with one packed atomic per lane (a whole 256-byte row per instruction).
    // The walk is scalar: the run ends of the wave's 32 rows are ONE ballot, a flush takes its destination from
    // v_readlane, and the 32 values of a lane are read from the stage up front -- the first version compared per row
    // through ds_bpermute under exec masks and waited for every row's LDS read before the next (two LDS round trips per
    // row: ~4 us of a tile's ~9).
    if (dbg & 8) continue;
    uint32_t ends;
    {
      const int64_t nxt = __shfl_down(si, 1);
      const uint32_t valid = nrows >= 32 ? 0xffffffffu : ((1u << nrows) - 1u);
      ends = ((uint32_t)__ballot(si != nxt) & 0x7fffffffu & valid) | (1u << (nrows - 1));
    }
    // Straight-line part: every row's running sum of its run, rounded to T, replaces the row's value (= what a flush at
    // that row writes); a row that opens a run drops the carried sum through a scalar select.  No branch per row, and the
    // flush code exists once (a loop over the run ends with a register-indexed read) instead of 32 times.
    u32x32 mv;
#pragma unroll
    for (int r = 0; r < 32; ++r) mv[r] = *reinterpret_cast<const uint32_t*>(xs + (r * 16 + (cch ^ (r & 15))) * 16 + cdw * 4);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const uint32_t keep = (r == 0 || ((ends >> (r - 1)) & 1u)) ? 0u : 0xffffffffu;  // wave-uniform
      float a, b;
      unpack2<BF16>(mv[r], &a, &b);
      s0 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s0) & keep) + a;
      s1 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s1) & keep) + b;
      mv[r] = pack2<BF16>(s0, s1);
    }
    // flushes: one per run end (bits of `ends`: rows behind the tile's last edge have none)
    const int si_lo = (int)(uint32_t)si, si_hi = (int)(uint32_t)((uint64_t)si >> 32);
    for (uint32_t m = ends; m != 0; m &= m - 1) {
      const int r = __builtin_ctz(m);
      const int64_t d = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(si_hi, r) << 32) |
                                  (uint32_t)__builtin_amdgcn_readlane(si_lo, r));
      if (dbg & 32) cas_add_pk_bits<BF16>(out + d * 256 + (uint32_t)(lane * 4), mv[r]);
      else if (!(dbg & 1)) atomic_add_pk_bits<BF16>(out + d * 256 + (uint32_t)(lane * 4), mv[r]);
    }
  }
}

// one pinned, device-visible error word per device (PYG_HIP_RGCN_DEFERRED)
int deferred_error_slot(int** out) {
  static std::mutex mu;
  static int* slots[64] = {nullptr};
  int dev = 0;
  PYG_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  if (!slots[dev]) {
    void* p = nullptr;
    PYG_HIP_CHECK(hipHostMalloc(&p, 64, hipHostMallocDefault));
    *static_cast<int*>(p) = 0;
    slots[dev] = static_cast<int*>(p);
  }
  *out = slots[dev];
  return PYG_HIP_OK;
}

#include "rgcn_grouped.h"

const char* rgcn_error_text(int code) {
  return code == 1 ? "a gather index out of range" : code == 2 ? "a scatter index out of range"
                                                                : "scatter indices that are not grouped (nondecreasing) although PYG_HIP_RGCN_GROUPED promised so";
}

// PYG_HIP_RGCN_GROUPED: the row-start launch, then the owner-computes kernel (rgcn_grouped.h).  `out` is WRITTEN.
int rgcn_grouped_launch(int dtype, const void* x, int64_t num_x_rows, const pyg_hip_rgcn_relation* rels, int64_t R, void* out,
                        int64_t num_out_rows, int64_t K, int64_t M, int checked, void* workspace, size_t workspace_bytes,
                        hipStream_t stream) {
  const size_t esz = dtype == PYG_F32 ? 4 : 2;
  // 16-bit: K, M both in {128, 256} have their own instances; every other pair of multiples of 8 up to 256 takes the instance
  // with run-time row sizes (rgcn_grouped_small_kernel: KC = MC = 2 slices, those that do not exist skipped)
  const bool small = esz == 2 && !((K == 128 || K == 256) && (M == 128 || M == 256));
  const int KC = small ? 2 : (int)(K * esz / 256), MC = small ? 2 : (int)(M * esz / 256);   // 256-byte slices of a feature row / of a row of `out`
  if (num_out_rows == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(out, "rgcn_fused: NULL tensor");
  PYG_HIP_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "rgcn_fused: 'out' must be 16-byte aligned in grouped mode");
  if (num_out_rows >= (1LL << 31) - 64)
    return fail(PYG_HIP_ERR_UNSUPPORTED, "rgcn_fused: grouped mode handles fewer than 2^31 output rows");
  if (R > kGroupedMaxRel)
    return fail(PYG_HIP_ERR_UNSUPPORTED, "rgcn_fused: grouped mode handles up to %d relations", kGroupedMaxRel);
  int64_t E = 0;
  for (int64_t r = 0; r < R; ++r) {
    if (rels[r].num_edges >= (1LL << 31) - 64)
      return fail(PYG_HIP_ERR_UNSUPPORTED, "rgcn_fused: grouped mode handles fewer than 2^31 edges per relation");
    if (rels[r].num_edges > 0) {
      PYG_HIP_REQUIRE(rels[r].scatter_offset >= 0 && rels[r].scatter_offset < num_out_rows && rels[r].scatter_rows >= 0,
                      "rgcn_fused: relation %lld: scatter_offset outside of 'out' (or negative scatter_rows)", (long long)r);
      PYG_HIP_REQUIRE((rels[r].x ? rels[r].x_rows : num_x_rows) > 0, "rgcn_fused: relation %lld gathers from an empty table", (long long)r);
    }
    E += rels[r].num_edges;
  }
  if (E == 0) {  // nothing arrives anywhere
    PYG_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)num_out_rows * (size_t)M * esz, stream));
    return PYG_HIP_OK;
  }
  const size_t need = grouped_workspace_bytes(rels, R, num_out_rows);
  if (workspace == nullptr || workspace_bytes < need)
    return fail(PYG_HIP_ERR_WORKSPACE, "rgcn_fused: grouped mode needs a workspace of %zu bytes (pyg_hip_rgcn_grouped_workspace_size), got %zu",
                need, workspace_bytes);
  const size_t rel_b = align_up(sizeof(RelDev) * (size_t)R, 256);
  const size_t vec_b = align_up(sizeof(int64_t) * (size_t)(3 * R + 1), 256);
  const size_t meta_b = align_up(sizeof(int32_t) * 2 * (size_t)R, 256);
  const bool inl = R <= kRgcnInline;
  GroupedDesc desc;
  ::memset(&desc, 0, sizeof(desc));
  RelDev* hr = desc.irels;
  int64_t* hp = desc.ieprefix;
  int64_t* ho = desc.irp_off;
  int64_t* hs = desc.ispan;
  if (!inl) {
    void* staged = nullptr;
    int rc = pinned_stage().acquire(rel_b + vec_b, &staged);
    if (rc != PYG_HIP_OK) return rc;
    hr = static_cast<RelDev*>(staged);
    hp = reinterpret_cast<int64_t*>(static_cast<char*>(staged) + rel_b);
    ho = hp + (R + 1);
    hs = ho + R;
  }
  bool big = false;
  int64_t e = 0, rp_entries = 0;
  for (int64_t r = 0; r < R; ++r) {
    hr[r].gather_index = rels[r].gather_index;
    hr[r].scatter_index = rels[r].scatter_index;
    hr[r].weight = static_cast<const char*>(rels[r].weight);
    hr[r].num_edges = rels[r].num_edges;
    hr[r].gather_offset = rels[r].gather_offset;
    hr[r].scatter_offset = rels[r].scatter_offset;
    hr[r].x = static_cast<const char*>(rels[r].x ? rels[r].x : x);
    hr[r].gather_map = rels[r].gather_map;
    hr[r].x_rows = rels[r].x ? rels[r].x_rows : num_x_rows;
    hr[r].map_len = rels[r].gather_map_len;
    big = big || hr[r].x_rows * (int64_t)(K * esz) >= (1LL << 32);
    hp[r] = e;
    e += rels[r].num_edges;
    const int64_t span = grouped_span(rels[r], num_out_rows);
    ho[r] = rp_entries;
    hs[r] = span;
    rp_entries += span;
  }
  hp[R] = e;
  char* w = static_cast<char*>(workspace);
  if (!inl) {
    PYG_HIP_CHECK(hipMemcpyAsync(w, hr, rel_b + vec_b, hipMemcpyHostToDevice, stream));
    int rc = pinned_stage().commit(stream);
    if (rc != PYG_HIP_OK) return rc;
    desc.rels = reinterpret_cast<const RelDev*>(w);
    desc.eprefix = reinterpret_cast<const int64_t*>(w + rel_b);
    desc.rp_off = desc.eprefix + (R + 1);
    desc.span = desc.rp_off + R;
  }
  desc.meta = reinterpret_cast<int32_t*>(w + rel_b + vec_b);
  int* err_dev = reinterpret_cast<int*>(w + rel_b + vec_b + meta_b);
  desc.rp = reinterpret_cast<int32_t*>(w + rel_b + vec_b + meta_b + 256);
  desc.long_rows = reinterpret_cast<uint64_t*>(w + rel_b + vec_b + meta_b + 64);
#if PYG_ABL_ & 16
  desc.dbg = reinterpret_cast<uint64_t*>(w + need - 65536);
#endif
  desc.row_bytes = (int)(K * esz);
  desc.out_bytes = (int)(M * esz);
  {  // an id that no earlier call of this process and no stale word of the workspace holds
    static std::atomic<uint64_t> counter{0};
    static const uint64_t salt = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() << 24;
    desc.call_id = (salt ^ 0x9e3779b97f4a7c15ull) + counter.fetch_add(1) + 1;
  }
  const bool deferred = (checked & PYG_HIP_RGCN_DEFERRED) != 0 && (checked & PYG_HIP_RGCN_CHECKED) == 0;
  const bool sync_check = (checked & PYG_HIP_RGCN_CHECKED) != 0;
  if (sync_check) PYG_HIP_CHECK(hipMemsetAsync(err_dev, 0, sizeof(int), stream));
  if (deferred) {
    int* slot = nullptr;
    int rc = deferred_error_slot(&slot);
    if (rc != PYG_HIP_OK) return rc;
    const int pending = *static_cast<volatile int*>(slot);
    if (pending != 0) {
      *static_cast<volatile int*>(slot) = 0;
      return fail(PYG_HIP_ERR_INVALID,
                  "rgcn_fused: an earlier call on this device had %s (set PYG_HIP_RGCN_CHECK=1 to fail in the call that has them)",
                  rgcn_error_text(pending));
    }
    err_dev = slot;
  }
  const bool bf = dtype == PYG_BF16, ck = sync_check || deferred;
  const void* prep = ck ? (inl ? (const void*)&rgcn_rowstart_kernel<true, true> : (const void*)&rgcn_rowstart_kernel<true, false>)
                        : (inl ? (const void*)&rgcn_rowstart_kernel<false, true> : (const void*)&rgcn_rowstart_kernel<false, false>);
  const void* kern;
  {
#define PYG_RGCN_GPICK(K, BF, CK, BG) (inl ? (const void*)&K<BF, CK, BG, true> : (const void*)&K<BF, CK, BG, false>)
#define PYG_RGCN_GPICK4(K)                                                                                                       \
  (bf ? (ck ? (big ? PYG_RGCN_GPICK(K, true, true, true) : PYG_RGCN_GPICK(K, true, true, false))                                \
            : (big ? PYG_RGCN_GPICK(K, true, false, true) : PYG_RGCN_GPICK(K, true, false, false)))                              \
      : (ck ? (big ? PYG_RGCN_GPICK(K, false, true, true) : PYG_RGCN_GPICK(K, false, true, false))                              \
            : (big ? PYG_RGCN_GPICK(K, false, false, true) : PYG_RGCN_GPICK(K, false, false, false))))
    kern = PYG_RGCN_GPICK4(rgcn_grouped_kernel);
#define PYG_RGCN_SPICK(KC_, MC_)                                                                                                        \
  (bf ? (big ? (inl ? (const void*)&rgcn_grouped_shape_kernel<true, true, true, KC_, MC_> : (const void*)&rgcn_grouped_shape_kernel<true, true, false, KC_, MC_>)    \
             : (inl ? (const void*)&rgcn_grouped_shape_kernel<true, false, true, KC_, MC_> : (const void*)&rgcn_grouped_shape_kernel<true, false, false, KC_, MC_>)) \
      : (big ? (inl ? (const void*)&rgcn_grouped_shape_kernel<false, true, true, KC_, MC_> : (const void*)&rgcn_grouped_shape_kernel<false, true, false, KC_, MC_>)  \
             : (inl ? (const void*)&rgcn_grouped_shape_kernel<false, false, true, KC_, MC_> : (const void*)&rgcn_grouped_shape_kernel<false, false, false, KC_, MC_>)))
    if (small)
      kern = bf ? (big ? (inl ? (const void*)&rgcn_grouped_small_kernel<true, true, true> : (const void*)&rgcn_grouped_small_kernel<true, true, false>)
                       : (inl ? (const void*)&rgcn_grouped_small_kernel<true, false, true> : (const void*)&rgcn_grouped_small_kernel<true, false, false>))
                : (big ? (inl ? (const void*)&rgcn_grouped_small_kernel<false, true, true> : (const void*)&rgcn_grouped_small_kernel<false, true, false>)
                       : (inl ? (const void*)&rgcn_grouped_small_kernel<false, false, true> : (const void*)&rgcn_grouped_small_kernel<false, false, false>));
    else if (dtype == PYG_F32 && (K != 128 || M != 128))
      kern = big ? (inl ? (const void*)&rgcn_grouped_f32_small_kernel<true, true> : (const void*)&rgcn_grouped_f32_small_kernel<true, false>)
                 : (inl ? (const void*)&rgcn_grouped_f32_small_kernel<false, true> : (const void*)&rgcn_grouped_f32_small_kernel<false, false>);
    else if (dtype == PYG_F32)
      kern = big ? (inl ? (const void*)&rgcn_grouped_f32_kernel<true, true> : (const void*)&rgcn_grouped_f32_kernel<true, false>)
                 : (inl ? (const void*)&rgcn_grouped_f32_kernel<false, true> : (const void*)&rgcn_grouped_f32_kernel<false, false>);
    else if (KC == 2 && MC == 2) kern = PYG_RGCN_SPICK(2, 2);
    else if (KC == 1 && MC == 2) kern = PYG_RGCN_SPICK(1, 2);
    else if (KC == 2 && MC == 1) kern = PYG_RGCN_SPICK(2, 1);
#undef PYG_RGCN_SPICK
#undef PYG_RGCN_GPICK4
#undef PYG_RGCN_GPICK
  }
  // 16-bit: 32 KB of W, 32-row A tiles, three workgroups per CU; fp32: all 64 KB of W, an A tile of 16 rows of 528 bytes, two
  // workgroups per CU (rgcn_grouped.h)
  const bool wide = dtype == PYG_F32;
  const int lds = (wide ? 65536 + 8704 : 32768 + 8192 * KC) + 2 * 4 * kGroupedMaxRel;
  if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
  int Ri = (int)R;
  {
    void* args[] = {(void*)&desc, (void*)&Ri, (void*)&err_dev};
    PYG_HIP_CHECK(hipLaunchKernel(prep, dim3((unsigned)((E + 255) / 256)), dim3(256), args, 0, stream));
  }
  {
    char* outc = static_cast<char*>(out);
    void* args[] = {(void*)&desc, (void*)&Ri, (void*)&outc, (void*)&num_out_rows, (void*)&err_dev};
    // persistent: three workgroups of 256 threads per CU (<= 168 VGPRs, 44 KB of LDS), blocks of 16 rows dealt round-robin (the rows with
    // edges are the first ones of every node type: neighbours in the grid, spread over the chip)
    const int64_t nblocks = (num_out_rows + 15) / 16;
    const int64_t grid = std::max<int64_t>(1, std::min<int64_t>(nblocks, (wide ? 2 : 3) * (int64_t)device_info().num_cus));
    PYG_HIP_CHECK(hipLaunchKernel(kern, dim3((unsigned)grid), dim3(256), args, lds, stream));
  }
  if (sync_check) {
    int host_err = 0;
    PYG_HIP_CHECK(hipMemcpyAsync(&host_err, err_dev, sizeof(int), hipMemcpyDeviceToHost, stream));
    PYG_HIP_CHECK(hipStreamSynchronize(stream));
    if (host_err != 0) return fail(PYG_HIP_ERR_INVALID, "rgcn_fused: the call has %s", rgcn_error_text(host_err));
  }
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

}  // namespace
}  // namespace pyg_hip

using namespace pyg_hip;

extern "C" {

int pyg_hip_rgcn_pending_error(void) {
  int* slot = nullptr;
  if (deferred_error_slot(&slot) != PYG_HIP_OK) return PYG_HIP_ERR_RUNTIME;
  const int pending = *static_cast<volatile int*>(slot);
  *static_cast<volatile int*>(slot) = 0;
  return pending;
}

size_t pyg_hip_rgcn_grouped_workspace_size(const pyg_hip_rgcn_relation* relations, int64_t num_relations, int64_t num_out_rows) {
  if (relations == nullptr || num_relations <= 0 || num_out_rows <= 0) return 256;
  return grouped_workspace_bytes(relations, num_relations, num_out_rows);
}

size_t pyg_hip_rgcn_fused_workspace_size(int64_t num_relations, int64_t num_edges) {
  if (num_relations < 0) num_relations = 0;
  (void)num_edges;
  return align_up(sizeof(RelDev) * (size_t)std::max<int64_t>(num_relations, 1), 256) +
         align_up(sizeof(int32_t) * (size_t)(num_relations + 1), 256) + 256;
}

int pyg_hip_rgcn_fused(int dtype, const void* x, int64_t num_x_rows, const pyg_hip_rgcn_relation* rels, int64_t R,
                       void* out, int64_t num_out_rows, int64_t K, int64_t M, int checked, void* workspace,
                       size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const bool grouped = (checked & PYG_HIP_RGCN_GROUPED) != 0;
  PYG_HIP_REQUIRE(dtype == PYG_BF16 || dtype == PYG_F16 || (grouped && dtype == PYG_F32),
                  "rgcn_fused: bfloat16 / float16 only (PYG_HIP_RGCN_GROUPED: float32 too)");
  const bool any8 = K >= 8 && K <= 256 && M >= 8 && M <= 256 && K % 8 == 0 && M % 8 == 0;
  const bool any4 = K >= 4 && K <= 128 && M >= 4 && M <= 128 && K % 4 == 0 && M % 4 == 0;   // (fp32: rows of multiples of 16 bytes up to 512)
  if (dtype == PYG_F32 ? !any4 : (grouped ? !any8 : (K != 128 || M != 128)))
    return fail(PYG_HIP_ERR_UNSUPPORTED,
                "rgcn_fused: K = M = 128 only (PYG_HIP_RGCN_GROUPED: K, M any multiples of 8 up to 256 for the 16-bit types, of 4 up to 128 for float32); got %lld x %lld",
                (long long)K, (long long)M);
  PYG_HIP_REQUIRE(R >= 0 && R < (1 << 30), "rgcn_fused: bad relation count");
  PYG_HIP_REQUIRE(num_x_rows >= 0 && num_out_rows >= 0, "rgcn_fused: negative size");
  if (R == 0) {
    if (grouped && out && num_out_rows > 0)
      PYG_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)num_out_rows * (size_t)M * (dtype == PYG_F32 ? 4 : 2), stream));
    return PYG_HIP_OK;
  }
  PYG_HIP_REQUIRE(rels != nullptr, "rgcn_fused: 'relations' is NULL");
  int64_t E = 0, tiles = 0;
  for (int64_t r = 0; r < R; ++r) {
    PYG_HIP_REQUIRE(rels[r].num_edges >= 0, "rgcn_fused: negative edge count");
    PYG_HIP_REQUIRE(rels[r].num_edges == 0 || (rels[r].gather_index && rels[r].scatter_index && rels[r].weight),
                    "rgcn_fused: NULL tensor in relation %lld", (long long)r);
    PYG_HIP_REQUIRE((reinterpret_cast<uintptr_t>(rels[r].weight) & 15) == 0, "rgcn_fused: weights must be 16-byte aligned");
    E += rels[r].num_edges;
    tiles += (rels[r].num_edges + 127) / 128;
  }
  if (E == 0 && !grouped) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(out, "rgcn_fused: NULL tensor");
  for (int64_t r = 0; r < R; ++r) {
    const void* xr = rels[r].x ? rels[r].x : x;
    PYG_HIP_REQUIRE(rels[r].num_edges == 0 || xr != nullptr, "rgcn_fused: relation %lld has no feature table", (long long)r);
    PYG_HIP_REQUIRE((reinterpret_cast<uintptr_t>(xr) & 15) == 0, "rgcn_fused: misaligned feature table");
  }
  if (grouped)
    return rgcn_grouped_launch(dtype, x, num_x_rows, rels, R, out, num_out_rows, K, M, checked, workspace, workspace_bytes, stream);
  PYG_HIP_REQUIRE((reinterpret_cast<uintptr_t>(out) & 3) == 0, "rgcn_fused: misaligned tensor");
  PYG_HIP_REQUIRE(tiles < (1LL << 31), "rgcn_fused: too many tiles");
  if (workspace == nullptr || workspace_bytes < pyg_hip_rgcn_fused_workspace_size(R, E))
    return fail(PYG_HIP_ERR_WORKSPACE, "rgcn_fused: workspace of %zu bytes needed, got %zu",
                pyg_hip_rgcn_fused_workspace_size(R, E), workspace_bytes);
  const size_t rel_b = align_up(sizeof(RelDev) * (size_t)R, 256);
  const size_t tile_b = align_up(sizeof(int32_t) * (size_t)(R + 1), 256);
  const bool inl = R <= kRgcnInline;
  RgcnDesc desc;
  ::memset(&desc, 0, sizeof(desc));
  RelDev* hr = desc.irels;
  int32_t* ht = desc.itile;
  if (!inl) {
    void* staged = nullptr;
    int rc = pinned_stage().acquire(rel_b + tile_b, &staged);
    if (rc != PYG_HIP_OK) return rc;
    hr = static_cast<RelDev*>(staged);
    ht = reinterpret_cast<int32_t*>(static_cast<char*>(staged) + rel_b);
  }
  int64_t t = 0;
  bool big = false;  // a feature table of 4 GB or more: 64-bit row offsets
  for (int64_t r = 0; r < R; ++r) {
    hr[r].gather_index = rels[r].gather_index;
    hr[r].scatter_index = rels[r].scatter_index;
    hr[r].weight = static_cast<const char*>(rels[r].weight);
    hr[r].num_edges = rels[r].num_edges;
    hr[r].gather_offset = rels[r].gather_offset;
    hr[r].scatter_offset = rels[r].scatter_offset;
    hr[r].x = static_cast<const char*>(rels[r].x ? rels[r].x : x);
    hr[r].gather_map = rels[r].gather_map;
    hr[r].x_rows = rels[r].x ? rels[r].x_rows : num_x_rows;
    hr[r].map_len = rels[r].gather_map_len;
    big = big || hr[r].x_rows >= (1LL << 24);
    ht[r] = (int32_t)t;
    t += (rels[r].num_edges + 127) / 128;
  }
  ht[R] = (int32_t)t;
  char* w = static_cast<char*>(workspace);
  if (!inl) {
    PYG_HIP_CHECK(hipMemcpyAsync(w, hr, rel_b + tile_b, hipMemcpyHostToDevice, stream));
    int rc = pinned_stage().commit(stream);
    if (rc != PYG_HIP_OK) return rc;
    desc.rels = reinterpret_cast<const RelDev*>(w);
    desc.tile_start = reinterpret_cast<const int32_t*>(w + rel_b);
  }
  constexpr int lds = 32768 + 4 * 8192;
  int* err_dev = reinterpret_cast<int*>(w + rel_b + tile_b);
  const bool cas = (checked & PYG_HIP_RGCN_CAS) != 0 || float_atomic_mode() == 1;
  const bool deferred = (checked & PYG_HIP_RGCN_DEFERRED) != 0 && (checked & PYG_HIP_RGCN_CHECKED) == 0;
  checked &= PYG_HIP_RGCN_CHECKED;
  if (checked) PYG_HIP_CHECK(hipMemsetAsync(err_dev, 0, sizeof(int), stream));
  if (deferred) {
    // validated without a synchronisation: offenders are redirected to row 0 (never an out-of-bounds access) and the
    // kernel leaves the code in a pinned word of this device; whoever looks next -- the next call here, or
    // pyg_hip_rgcn_pending_error -- reports it
    int* slot = nullptr;
    int rc = deferred_error_slot(&slot);
    if (rc != PYG_HIP_OK) return rc;
    const int pending = *static_cast<volatile int*>(slot);
    if (pending != 0) {
      *static_cast<volatile int*>(slot) = 0;
      return fail(PYG_HIP_ERR_INVALID,
                  "rgcn_fused: an earlier call on this device had %s (out-of-range edges were redirected to row 0 or "
                  "dropped; set PYG_HIP_RGCN_CHECK=1 to fail in the call that has them)",
                  rgcn_error_text(pending));
    }
    err_dev = slot;
  }
  const void* kern;
  {
#define PYG_RGCN_PICK3(BF, CK, BG) (inl ? (const void*)&rgcn_fused_kernel<BF, CK, BG, true> : (const void*)&rgcn_fused_kernel<BF, CK, BG, false>)
#define PYG_RGCN_PICK(BF, CK, BG) PYG_RGCN_PICK3(BF, CK, BG)
    const bool bf = dtype == PYG_BF16, ck = checked != 0 || deferred;
    kern = bf ? (ck ? (big ? PYG_RGCN_PICK(true, true, true) : PYG_RGCN_PICK(true, true, false))
                    : (big ? PYG_RGCN_PICK(true, false, true) : PYG_RGCN_PICK(true, false, false)))
              : (ck ? (big ? PYG_RGCN_PICK(false, true, true) : PYG_RGCN_PICK(false, true, false))
                    : (big ? PYG_RGCN_PICK(false, false, true) : PYG_RGCN_PICK(false, false, false)));
#undef PYG_RGCN_PICK
#undef PYG_RGCN_PICK3
  }
  if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
  // persistent grid: two workgroups per CU (64 KB of LDS each), every one a contiguous range of >= 2 tiles
  int64_t per_cu = 2;
  int64_t min_tiles = 2;
  int dbg = 0;
#ifdef PYG_HIP_EXPERIMENTS  // timing ablations: never part of the shipped library
  if (const char* e = getenv("PYG_HIP_RGCN_WGS")) per_cu = std::max(1, atoi(e));
  if (const char* e = getenv("PYG_HIP_RGCN_MIN_TILES")) min_tiles = std::max(1, atoi(e));
  if (const char* e = getenv("PYG_HIP_RGCN_DBG")) dbg = atoi(e);
#endif
  if (cas) dbg |= 32;
  note_accumulate("pyg_hip_rgcn_fused", out, (size_t)num_out_rows * 256, "the caller (`out` is accumulated into)", stream, cas ? 1 : 0);
  const int64_t grid = std::max<int64_t>(1, std::min<int64_t>((tiles + min_tiles - 1) / min_tiles, per_cu * (int64_t)device_info().num_cus));
  char* outc = static_cast<char*>(out);
  int Ri = (int)R;
  void* args[] = {(void*)&desc, (void*)&Ri, (void*)&outc, (void*)&num_out_rows, (void*)&err_dev, (void*)&dbg};
  PYG_HIP_CHECK(hipLaunchKernel(kern, dim3((unsigned)grid), dim3(256), args, lds, stream));
  if (checked) {  // checked mode synchronises: the caller asked for a verdict
    int host_err = 0;
    PYG_HIP_CHECK(hipMemcpyAsync(&host_err, err_dev, sizeof(int), hipMemcpyDeviceToHost, stream));
    PYG_HIP_CHECK(hipStreamSynchronize(stream));
    if (host_err == 1) return fail(PYG_HIP_ERR_INVALID, "rgcn_fused: gather index out of range");
    if (host_err == 2) return fail(PYG_HIP_ERR_INVALID, "rgcn_fused: scatter index out of range");
  }
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

}  // extern "C"
