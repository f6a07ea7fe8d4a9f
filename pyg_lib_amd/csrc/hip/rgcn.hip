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
// One workgroup (4 waves x 32 edges) per 128-edge tile of one relation, dispatched in tile order:
//   * A-operand rows are GATHERED: the X tile arrives by LDS-DMA (global_load_lds_dwordx4) whose per-lane source
//     address is x + (gather_index[e] + offset) * row_bytes + chunk * 16 -- a whole 256-byte row per 16 lanes --
//     into the same XOR-swizzled per-wave stage the segment_matmul kernels use;
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

#include <vector>

namespace pyg_hip {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void LDSV;

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

struct TileDev {
  int32_t rel;
  int32_t tile;   // first 128-edge tile (index inside the relation) of this workgroup's run
  int32_t count;  // consecutive tiles of the relation it processes with ONE copy of W_r in LDS
  int32_t pad;
};

template <bool BF16>
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 acc) {
  if constexpr (BF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}

template <bool BF16>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  if constexpr (BF16) {
    return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)a) | ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)b) << 16);
  } else {
    return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)a) | ((uint32_t)__builtin_bit_cast(uint16_t, (_Float16)b) << 16);
  }
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

template <bool BF16>
__device__ __forceinline__ void atomic_add_pk(char* addr, float a, float b) {
  const uint32_t v = pack2<BF16>(a, b);
  typedef __attribute__((address_space(1))) void GV;
  if constexpr (BF16)
    asm volatile("global_atomic_pk_add_bf16 %0, %1, off" ::"v"((GV*)addr), "v"(v) : "memory");
  else
    asm volatile("global_atomic_pk_add_f16 %0, %1, off" ::"v"((GV*)addr), "v"(v) : "memory");
}

// K = M = 128, 16-bit T
// CHECK: every gather / scatter index is validated; an offender sets *error and is redirected to row 0 (the host
// reports it): without the check a bad index is an out-of-bounds DMA read or an atomic into foreign memory.
template <bool BF16, bool CHECK>
__global__ __launch_bounds__(256) void rgcn_fused_kernel(const RelDev* __restrict__ rels, const TileDev* __restrict__ tiles,
                                                        char* __restrict__ out, int64_t out_rows, int* __restrict__ error,
                                                        int dbg) {  // dbg: 0 in the product (timing ablations, experiment builds)
  constexpr int NT = 4, NI = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, xl = lane & 31, h = lane >> 5;
#ifdef PYG_HIP_EXPERIMENTS
  // phase clocks (dbg & 4): cycles per wave summed into error[2 ...]: 0 W + first indices, 1 wait for the rows, 2 MFMAs,
  // 3 pack + stage, 4 scatter, 5 tiles
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#define PYG_RGCN_MARK(i)                                     \
  if (dbg & 4) {                                             \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    tph[i] += now_ - tlast;                                  \
    tlast = now_;                                            \
  }
#else
#define PYG_RGCN_MARK(i)
#endif
  const TileDev td = tiles[blockIdx.x];
  const RelDev rel = rels[td.rel];
  char* xs = smem + 32768 + wave * 8192;
  // W: this wave's 8 blocks of 4 k-rows, 16-byte chunks permuted inside a block (once per workgroup: its `count` tiles
  // are of one relation -- with one tile per workgroup the weight copy was as many bytes as the gathered rows)
  {
    const int dma_r = (lane & 15) >> 2, dma_ii = lane & 3, dma_u = lane >> 4;
    const int dma_c = 2 * dma_u + (dma_ii & 1) + 8 * (dma_ii >> 1);
    const char* wsrc = rel.weight + dma_r * 256 + dma_c * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kb = wave * 8 + j;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + kb * 1024),
                                       (LDSV*)(smem + kb * 1024), 16, 0, 0);
    }
  }
  const int q = lane & 15, grp16 = lane >> 4;
  const char* wb = smem + 16384 * h + (4 * (q >> 2) + (grp16 & 1) + 2 * (q & 1)) * 16 + ((q & 3) >> 1) * 8;
  const int cch = lane >> 2, cdw = lane & 3;
  // Index pipeline: a tile needs edge -> gather_index -> (gather_map) -> feature row, three dependent memory round
  // trips before its MFMAs.  The first level of tile t + 1 is requested together with tile t's row DMA, the second
  // right behind their common wait, so that from its second tile on a workgroup pays ONE round trip per tile.
  // lane l < 32 holds edge l of the wave's 32 (rows past the end repeat the last valid edge)
  auto first_level = [&](int tt, int& nrows, int64_t& g1, int64_t& si) {
    const int64_t e0 = (int64_t)(td.tile + tt) * 128 + wave * 32;   // first edge of this wave inside the relation
    const int64_t left = rel.num_edges - e0;
    nrows = left >= 32 ? 32 : (left > 0 ? (int)left : 0);
    g1 = 0;
    si = 0;
    if (nrows > 0) {
      const int64_t e = e0 + (xl < nrows ? xl : nrows - 1);
      g1 = rel.gather_index[e];
      si = rel.scatter_index[e] + rel.scatter_offset;
    }
  };
  auto second_level = [&](int nrows, int64_t g1) -> int64_t {
    if (nrows == 0) return 0;
    if (!rel.gather_map) return g1 + rel.gather_offset;
    if (CHECK && (g1 < 0 || g1 >= rel.map_len)) {  // double indirection done here: the gathered feature matrix never exists
      *error = 1;
      g1 = 0;
    }
    return rel.gather_map[g1];
  };
  int nrows_n;
  int64_t g1_n, si_n, gi_n;
  first_level(0, nrows_n, g1_n, si_n);
  gi_n = second_level(nrows_n, g1_n);
#ifdef PYG_HIP_EXPERIMENTS
  if (dbg & 4) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(gi_n), "+v"(si_n));
  }
#endif
  PYG_RGCN_MARK(0)
  for (int tt = 0; tt < td.count; ++tt) {
    const int nrows = nrows_n;
    int64_t gi = gi_n, si = si_n;
    if (nrows > 0) {
      if (CHECK) {
        if (gi < 0 || gi >= rel.x_rows) {
          *error = 1;
          gi = 0;
        }
        if (si < 0 || si >= out_rows) {
          *error = 2;
          si = 0;
        }
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int p = i * 64 + lane;
        const int r = p >> 4, cs = p & 15;
        const int c = cs ^ (r & 15);
        const int64_t row = (dbg & 2) ? (int64_t)((blockIdx.x * 4 + wave) * 32 + r) : __shfl(gi, r);  // (ablation: sequential rows)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rel.x + row * 256 + c * 16),
                                         (LDSV*)(xs + i * 1024), 16, 0, 0);
      }
    }
    const bool more = tt + 1 < td.count;
    if (more) first_level(tt + 1, nrows_n, g1_n, si_n);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tt == 0) __syncthreads();  // everybody's part of W has landed; the X stage is private to the wave
    if (more) gi_n = second_level(nrows_n, g1_n);
    PYG_RGCN_MARK(1)
    if (nrows == 0) continue;
    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int s = 0; s < NI; ++s) {
      const u32x4 xa = *reinterpret_cast<const u32x4*>(xs + (xl * 16 + ((NI * h + s) ^ (xl & 15))) * 16);
      u32x4 wa[NT];
#pragma unroll
      for (int t4 = 0; t4 < NT; ++t4) {
        const v4i16 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(wb + (2 * s) * 1024 + t4 * 256));
        const v4i16 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(wb + (2 * s + 1) * 1024 + t4 * 256));
        wa[t4] = __builtin_bit_cast(u32x4, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int t4 = 0; t4 < NT; ++t4) acc[t4] = mfma16<BF16>(wa[t4], xa, acc[t4]);
    }
#ifdef PYG_HIP_EXPERIMENTS
    if (dbg & 4) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#endif
    PYG_RGCN_MARK(2)
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
    // scatter: lane owns columns 2*lane, 2*lane + 1 (chunk lane / 4, dword lane % 4) of every row.  All 32 rows of the
    // lane's column pair are read first (one LDS round trip instead of 32 dependent ones: the atomics below are
    // `asm volatile` memory clobbers, the compiler would not move a read across them)
    uint32_t mv[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) mv[r] = *reinterpret_cast<const uint32_t*>(xs + (r * 16 + (cch ^ (r & 15))) * 16 + cdw * 4);
#ifdef PYG_HIP_EXPERIMENTS
    if (dbg & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    PYG_RGCN_MARK(3)
    float s0 = 0.f, s1 = 0.f;
    int64_t cur = __shfl(si, 0);
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      if (r < nrows) {
        const int64_t d = __shfl(si, r);
        if (d != cur) {  // wave-uniform
          if (!(dbg & 1)) atomic_add_pk<BF16>(out + cur * 256 + lane * 4, s0, s1);
          s0 = 0.f;
          s1 = 0.f;
          cur = d;
        }
        float a, b;
        unpack2<BF16>(mv[r], &a, &b);
        s0 += a;
        s1 += b;
      }
    }
    if (!(dbg & 1)) atomic_add_pk<BF16>(out + cur * 256 + lane * 4, s0, s1);
    PYG_RGCN_MARK(4)
#ifdef PYG_HIP_EXPERIMENTS
    if (dbg & 4) tph[5] += 1;
#endif
  }
#ifdef PYG_HIP_EXPERIMENTS
  if ((dbg & 4) && lane == 0)
    for (int i = 0; i < 6; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(error) + 1 + i, tph[i]);
#endif
}

}  // namespace
}  // namespace pyg_hip

using namespace pyg_hip;

extern "C" {

size_t pyg_hip_rgcn_fused_workspace_size(int64_t num_relations, int64_t num_edges) {
  if (num_relations < 0) num_relations = 0;
  if (num_edges < 0) num_edges = 0;
  const size_t tiles = (size_t)(num_edges / 128 + num_relations + 1);
  return align_up(sizeof(RelDev) * (size_t)std::max<int64_t>(num_relations, 1), 256) + align_up(sizeof(TileDev) * tiles, 256) + 256;
}

int pyg_hip_rgcn_fused(int dtype, const void* x, int64_t num_x_rows, const pyg_hip_rgcn_relation* rels, int64_t R,
                       void* out, int64_t num_out_rows, int64_t K, int64_t M, int checked, void* workspace,
                       size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(dtype == PYG_BF16 || dtype == PYG_F16, "rgcn_fused: bfloat16 / float16 only");
  if (K != 128 || M != 128) return fail(PYG_HIP_ERR_UNSUPPORTED, "rgcn_fused: K = M = 128 only (got %lld x %lld)", (long long)K, (long long)M);
  PYG_HIP_REQUIRE(R >= 0 && R < (1 << 30), "rgcn_fused: bad relation count");
  PYG_HIP_REQUIRE(num_x_rows >= 0 && num_out_rows >= 0, "rgcn_fused: negative size");
  if (R == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(rels != nullptr, "rgcn_fused: 'relations' is NULL");
  int64_t E = 0, tiles = 0;
  for (int64_t r = 0; r < R; ++r) E += std::max<int64_t>(rels[r].num_edges, 0);
  // tiles per workgroup: W_r (32 KB) is copied once per workgroup and every tile behind a workgroup's first has its
  // indices prefetched, so runs of a few tiles pay -- as long as the grid still has a couple of workgroups for each of
  // the chip's 2 x CUs slots
  const int64_t slots = 2 * (int64_t)device_info().num_cus;
  int64_t run = std::max<int64_t>(1, std::min<int64_t>(8, (E / 128) / (4 * slots)));
  int dbg = 0;
#ifdef PYG_HIP_EXPERIMENTS  // timing ablations (wrong results by construction): never part of the shipped library
  if (const char* e = getenv("PYG_HIP_RGCN_RUN")) run = std::max(1, atoi(e));
  if (const char* e = getenv("PYG_HIP_RGCN_DBG")) dbg = atoi(e);
#endif
  E = 0;
  for (int64_t r = 0; r < R; ++r) {
    PYG_HIP_REQUIRE(rels[r].num_edges >= 0, "rgcn_fused: negative edge count");
    PYG_HIP_REQUIRE(rels[r].num_edges == 0 || (rels[r].gather_index && rels[r].scatter_index && rels[r].weight),
                    "rgcn_fused: NULL tensor in relation %lld", (long long)r);
    PYG_HIP_REQUIRE((reinterpret_cast<uintptr_t>(rels[r].weight) & 15) == 0, "rgcn_fused: weights must be 16-byte aligned");
    E += rels[r].num_edges;
    tiles += ((rels[r].num_edges + 127) / 128 + run - 1) / run;
  }
  if (E == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(out, "rgcn_fused: NULL tensor");
  for (int64_t r = 0; r < R; ++r) {
    const void* xr = rels[r].x ? rels[r].x : x;
    PYG_HIP_REQUIRE(rels[r].num_edges == 0 || xr != nullptr, "rgcn_fused: relation %lld has no feature table", (long long)r);
    PYG_HIP_REQUIRE((reinterpret_cast<uintptr_t>(xr) & 15) == 0, "rgcn_fused: misaligned feature table");
  }
  PYG_HIP_REQUIRE((reinterpret_cast<uintptr_t>(out) & 3) == 0, "rgcn_fused: misaligned tensor");
  PYG_HIP_REQUIRE(tiles < (1LL << 31), "rgcn_fused: too many tiles");
  if (workspace == nullptr || workspace_bytes < pyg_hip_rgcn_fused_workspace_size(R, E))
    return fail(PYG_HIP_ERR_WORKSPACE, "rgcn_fused: workspace of %zu bytes needed, got %zu",
                pyg_hip_rgcn_fused_workspace_size(R, E), workspace_bytes);
  const size_t rel_b = align_up(sizeof(RelDev) * (size_t)R, 256);
  const size_t tile_b = align_up(sizeof(TileDev) * (size_t)tiles, 256);
  void* staged = nullptr;
  int rc = pinned_stage().acquire(rel_b + tile_b, &staged);
  if (rc != PYG_HIP_OK) return rc;
  RelDev* hr = static_cast<RelDev*>(staged);
  TileDev* ht = reinterpret_cast<TileDev*>(static_cast<char*>(staged) + rel_b);
  int64_t t = 0;
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
    const int64_t nt = (rels[r].num_edges + 127) / 128;
    for (int64_t k = 0; k < nt; k += run) ht[t++] = TileDev{(int32_t)r, (int32_t)k, (int32_t)std::min<int64_t>(run, nt - k), 0};
  }
  char* w = static_cast<char*>(workspace);
  PYG_HIP_CHECK(hipMemcpyAsync(w, staged, rel_b + tile_b, hipMemcpyHostToDevice, stream));
  rc = pinned_stage().commit(stream);
  if (rc != PYG_HIP_OK) return rc;
  const RelDev* drel = reinterpret_cast<const RelDev*>(w);
  const TileDev* dtile = reinterpret_cast<const TileDev*>(w + rel_b);
  constexpr int lds = 32768 + 4 * 8192;
  int* err_dev = reinterpret_cast<int*>(w + rel_b + tile_b);
  if (checked || dbg) PYG_HIP_CHECK(hipMemsetAsync(err_dev, 0, 64, stream));
  const void* kern = dtype == PYG_BF16 ? (checked ? (const void*)&rgcn_fused_kernel<true, true> : (const void*)&rgcn_fused_kernel<true, false>)
                                       : (checked ? (const void*)&rgcn_fused_kernel<false, true> : (const void*)&rgcn_fused_kernel<false, false>);
  if (int rc_ = ensure_dynamic_lds(kern, lds)) return rc_;
  char* outc = static_cast<char*>(out);
  void* args[] = {(void*)&drel, (void*)&dtile, (void*)&outc, (void*)&num_out_rows, (void*)&err_dev, (void*)&dbg};
  PYG_HIP_CHECK(hipLaunchKernel(kern, dim3((unsigned)tiles), dim3(256), args, lds, stream));
#ifdef PYG_HIP_EXPERIMENTS
  if (dbg & 4) {
    unsigned long long t[7];
    PYG_HIP_CHECK(hipStreamSynchronize(stream));
    PYG_HIP_CHECK(hipMemcpy(t, err_dev, sizeof(t), hipMemcpyDeviceToHost));
    fprintf(stderr, "rgcn phases (cycles per wave-tile; %llu wave-tiles): setup/tile %.0f  row wait %.0f  mfma %.0f  pack %.0f  scatter %.0f\n",
            t[6], (double)t[1] / t[6], (double)t[2] / t[6], (double)t[3] / t[6], (double)t[4] / t[6], (double)t[5] / t[6]);
  }
#endif
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
