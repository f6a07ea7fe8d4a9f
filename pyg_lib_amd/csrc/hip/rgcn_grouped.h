// The fused R-GCN layer WITHOUT atomics, for edge lists that are grouped by destination (PYG_HIP_RGCN_GROUPED) --
// included by rgcn.hip inside its namespace.
//
//     out[o] = sum over relations r, edges e of r with scatter_index_r[e] + scatter_offset_r == o of
//              x_r[gather(e)] @ W_r            = sum_r ( sum_e x_r[gather(e)] ) @ W_r
//
// The samplers emit every relation's edges grouped by the node they were sampled for (sampler/cpu/neighbor_kernel.cpp:
// 332-514: one frontier node after the other, a later hop's frontier has larger local ids), so `row` -- the scatter
// index of the layer -- is nondecreasing per relation.  Then a destination row can be OWNED: a workgroup takes 16
// consecutive rows of `out` (a block), one row per 16-lane group, and for every relation with edges into its rows
//   * every group finds its row's edges (one lookup in the row-start table a small launch in front of this one builds
//     from scatter_index: rgcn_rowstart_kernel), walks them 16 at a time -- lane c of the group fetches 16 bytes of each
//     source row, a whole 256-byte row per group and instruction -- and sums them in fp32 registers in edge order;
//   * the 16 sums, rounded to T once, are the A tile of ONE 32 x 128 x 128 MFMA product with W_r (half of its rows in
//     use; aggregate first, transform second: E / fan-out matrix rows instead of E), accumulated in fp32 across the
//     relations of the block;
//   * the block's rows are written once, rounded once -- rows without edges as zeros.
// No atomics, no zero fill of `out` in front, no read-modify-write of `out`, the same bits on every run (the sums of a
// row are taken in edge order, relations in list order).  Rounding: one rounding of the per-relation feature sum
// (relative 2^-9 for bf16, where the chain rounds every message) and one of the result (where scatter_sum of the chain
// rounds once per destination as well).
//
// `out` is written with non-temporal stores (it is not read again here, and 40 % of the kernel's traffic: 0.0666 -> 0.0589 ms
// on the C5 batch; non-temporal LOADS of the feature rows cost 4 %: neighbouring groups share lines).
// Shapes: bf16 / f16 with K, M any multiples of 8 up to 256 -- 128 x 128 is the pipeline described below, K, M in {128, 256} the
// same pipeline over 128-feature sub-items, every other pair one instance with run-time row sizes (item at a time); float32
// with K, M multiples of 4 up to 128 -- 128 x 128 with both slices of the rows and all of W per item and the product on fp32
// MFMAs, the others item at a time with FMAs.  HBM traffic by the counters (profiles/r5_pmc_ops.json): 169 MB fetched + 106 MB
// written per C5 batch = 1.07 x the formula's 263 MB.

// lane I of every 16-lane DPP row to all lanes of the row (v_mov_b32_dpp row_newbcast:I)
template <int I>
__device__ __forceinline__ int row_bcast(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + I, 0xf, 0xf, false);
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// (by value: __builtin_bit_cast applied to an ELEMENT of an ext_vector -- `bit_cast(float, v[k])` -- returned element 0
// for every k)
__device__ __forceinline__ float u2f(uint32_t v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ uint32_t f2u(float v) { return __builtin_bit_cast(uint32_t, v); }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Timing experiments only (tools/build_variant.sh <name> rgcn -DPYG_HIP_RGCN_ABLATE=<bits>; wrong results by construction; the
// shipped library is built without it): 1 = no W loads in the pipeline, 2 = no products, 4 = no feature-row loads, 8 = no zero
// rows for blocks without edges.
#ifdef PYG_HIP_RGCN_ABLATE
#define PYG_ABL_ PYG_HIP_RGCN_ABLATE
#else
#define PYG_ABL_ 0
#endif

struct GroupedDesc {
  const RelDev* rels;        // R > kRgcnInline: the records and the three vectors below live in the workspace
  const int64_t* eprefix;    // [R + 1] running edge count
  const int64_t* rp_off;     // [R] first entry of the relation's row starts in `rp`
  const int64_t* span;       // [R] rows of the relation's destination segment (scatter_rows; 0: `out` at and behind its scatter_offset)
  int32_t* rp;               // row starts: rp[rp_off[r] + d] = first edge of relation r with scatter index d (only for d that occur)
  int32_t* meta;             // [2 r] the first, [2 r + 1] the last scatter index of relation r (written by the row-start launch)
  uint64_t* long_rows;       // == call_id: some row of this call has more than 16 edges (set by the row-start launch; the
  uint64_t call_id;          //  workspace is not cleared: an id no earlier call and no stale word can hold)
  int row_bytes, out_bytes;  // SMALL instances (K or M = 64): bytes of a feature row / of a row of `out`
#if PYG_ABL_ & 16
  uint64_t* dbg;             // per workgroup: start, end, sub-items, iterations (the last 64 KB of the workspace)
#endif
  RelDev irels[kRgcnInline];
  int64_t ieprefix[kRgcnInline + 1];
  int64_t irp_off[kRgcnInline];
  int64_t ispan[kRgcnInline];
};
static_assert(sizeof(GroupedDesc) <= 3072, "kernel argument");
constexpr int kGroupedMaxRel = 512;    // relations whose row ranges the owner-computes kernel keeps in LDS

// One thread per edge: an edge whose scatter index differs from its predecessor's starts a row.  Rows that do not occur
// keep whatever the workspace held: the consumer verifies scatter_index[rp[d]] == d, which no stale value can satisfy for
// a row without edges.  CHECK: indices outside [0, span) -> *error = 2, a descent -> *error = 3 (not grouped).
template <bool CHECK, bool INL>
__global__ __launch_bounds__(256) void rgcn_rowstart_kernel(const GroupedDesc desc, int R, int* __restrict__ error) {
  auto ep = [&](int i) -> int64_t {
    if constexpr (INL) return desc.ieprefix[i];
    else return desc.eprefix[i];
  };
  // (every index into the argument arrays is wave-uniform: a per-lane index would copy the struct to scratch)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t w0 = (int64_t)blockIdx.x * 256 + wv * 64;
  const int64_t gid = w0 + (threadIdx.x & 63);
  const int64_t total = ep(R);
  if (w0 >= total) return;
  int lo = 0, hi = R;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ep(mid) <= w0) lo = mid; else hi = mid;
  }
  for (int r = lo; r < R; ++r) {
    const int64_t e0 = ep(r), e1 = ep(r + 1);
    if (e0 >= w0 + 64) break;
    if (gid < e0 || gid >= e1) continue;
    const int64_t* sidx;
    int64_t span, off;
    if constexpr (INL) sidx = desc.irels[r].scatter_index, span = desc.ispan[r], off = desc.irp_off[r];
    else sidx = desc.rels[r].scatter_index, span = desc.span[r], off = desc.rp_off[r];
    const int64_t e = gid - e0;
    const int64_t d = ((GI64*)sidx)[e];
    const int64_t prev = e > 0 ? ((GI64*)sidx)[e - 1] : -1;
    const bool ok = d >= 0 && d < span;
    if (CHECK) {
      if (!ok) *error = 2;
      else if (d < prev) *error = 3;
    }
    if (ok && d != prev) desc.rp[off + d] = (int32_t)e;
    if (e >= 16 && ((GI64*)sidx)[e - 16] == d) *desc.long_rows = desc.call_id;   // a row of 17 or more edges
    const int64_t dc = d < 0 ? 0 : (d >= span ? span - 1 : d);
    if (e == 0) desc.meta[2 * r] = (int32_t)dc;
    if (e == e1 - e0 - 1) desc.meta[2 * r + 1] = (int32_t)dc;
  }
}

// 256 threads = 16 groups of 16 lanes = 16 rows of `out` per block.
//
// Persistent launch (three workgroups per CU), blocks dealt round-robin.  The unit of work is an ITEM = (block, relation with
// edges into the block); a block without items costs its 4 KB of zero stores and nothing else.  An item needs four
// DEPENDENT memory round trips -- row start -> the two indices -> gather_map -> feature rows, 3 - 5 us each on this chip
// under load -- so the items of a workgroup run through a four-deep software pipeline: every iteration first uses what
// the previous iteration requested (one wait at its top), then issues, back to back,
//     rows of item i + 1 | gather_map lookups of item i + 2 | indices of item i + 3 | row starts of item i + 4 | W of item i + 1
// and only then does item i's arithmetic (A tile, barrier, 8 MFMAs per wave, barrier, W of item i + 1, the block's stores
// when its last item is done).  (The first version walked one block at a time, its stages one after the other: 104 us on
// the C5 batch where the atomic kernel + its zero fill take 66; this one 57.  An item-at-a-time walk with enough
// workgroups per CU is only 2 % behind at 128 x 128 -- it serves rows of more than 16 edges and the run-time-size instances.)
template <bool BF16, bool CHECK, bool BIG, bool INL, int NW, int KC, int MC, bool F32 = false, bool SMALL = false>
__device__ __forceinline__ void rgcn_grouped_body(const GroupedDesc& desc, int R, char* __restrict__ out, int64_t out_rows,
                                                  int* __restrict__ error) {
  constexpr int U = 16;   // edges of a row per batch (one per lane of the group)
  constexpr int ROWS = 4 * NW;          // rows of `out` per block: one per 16-lane group
  // (NW = 8 -- 32 rows per block, 512 threads, 128 registers with the accumulators parked in LDS between the items of a
  // block, half the items per workgroup -- was built and measured: 0.0673 ms on the C5 batch against 0.0657 - 0.0679 for
  // NW = 4.  Neither is latency bound any more: 145 MB of random 256-byte rows + 110 MB of zero rows + 10 MB of indices in
  // 57 us is what the chip delivers for this access mix -- random-row gathers run at 3.7 - 4.7 TB/s in every kernel of
  // this library.)
  // WIDE (fp32): both 256-byte slices of a feature row travel together (2 U loads per lane and batch in the pipeline), all of
  // W (64 KB) has room in LDS, and the kernel runs with TWO workgroups per CU (256 registers per lane).
  constexpr bool WIDE = F32;
  constexpr int kWBytes = WIDE ? 65536 : 32768;
  constexpr int KS = WIDE ? KC : 1;   // slices of a batch in registers at the same time (the pipeline; the item-at-a-time walk: 1)
  // K = 128 KC input features (rows of 256 KC bytes), M = 128 MC output features: the feature rows are walked once per
  // 128-feature slice, W travels through LDS in 128 x 128 chunks (the pipeline at the end of this function: one SUB-ITEM per
  // slice; the item-at-a-time walk: one pass over the row's edges per slice, the indices come from L2 the second time).
  constexpr int RB0 = 256 * KC, OB0 = 256 * MC;   // bytes of a feature row / of a row of `out`
  // SMALL (instantiated with KC = MC = 2): K and M are any multiples of 8 up to 256 -- rows of RB = 2 K / OB = 2 M bytes known at
  // run time, kcn / mcn slices of them.  Lanes whose 16 bytes lie behind the row's end load nothing
  // (their part of the A tile is zero), W's columns behind M and rows behind K are written as zeros instead of copied, the
  // stores behind M are dropped, slices that do not exist are skipped; the item-at-a-time walk.
  static_assert(!SMALL || (KC == 2 && MC == 2), "SMALL: 16-bit with K, M multiples of 8 up to 256; fp32 with multiples of 4 up to 128");
  const int RB = SMALL ? desc.row_bytes : RB0, OB = SMALL ? desc.out_bytes : OB0;
  const int kcn = SMALL ? (RB + 255) >> 8 : KC, mcn = SMALL ? (OB + 255) >> 8 : MC;   // slices of a feature row / of a row of `out` that exist
  // F32: K = M = 128 floats, i.e. KC = MC = 2 in BYTES (rows of 512 bytes, walked in two 256-byte slices like K = 256 of the
  // 16-bit types); sums, the A tile and the product stay fp32 -- plain FMAs, IEEE like the reference's fp32 (no MFMA: after
  // the aggregation the product is 16 x 128 x 128 per item, ~4 us of a CU's FMA and LDS time next to ~25 us of gathers) --
  // and every thread stores its own 8 results (no pass through the tile).
  static_assert(!F32 || (KC == 2 && MC == 2), "fp32: K = M = 128");
  constexpr int kARow = 528;   // F32: bytes between the rows of the fp32 A tile (512 + 16: the four rows a wave reads fall
                               // into different banks)
  // Rows of more than 16 edges (fan-outs above 16, full neighbourhoods) need further batches.  That loop inside the pipeline
  // costs ~80 registers next to the pipeline's own (a third of the occupancy of every call), so the pipeline does without
  // it, and a call in which the row-start launch has seen such a row takes the item-at-a-time walk at the end of this
  // function instead (same registers, no pipeline; the choice is uniform over the launch).
  const bool long_rows = SMALL || *desc.long_rows == desc.call_id;
  [[maybe_unused]] const uint64_t abl_t0 = (PYG_ABL_ & 16) ? __builtin_readcyclecounter() * 0 + wall_clock64() : 0;
  [[maybe_unused]] int abl_items = 0, abl_iters = 0;
  auto rel_at = [&](int i) __attribute__((always_inline)) -> const RelDev& {
    if constexpr (INL) return desc.irels[i];
    else return desc.rels[i];
  };
  auto rp_off_at = [&](int i) __attribute__((always_inline)) -> int64_t {
    if constexpr (INL) return desc.irp_off[i];
    else return desc.rp_off[i];
  };
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem + kWBytes;   // the A tile: 32 rows (ROWS used) x 16 chunks of 16 bytes, chunk index XOR-swizzled with the row
  int* rlo = reinterpret_cast<int*>(smem + kWBytes + (F32 ? 8704 : 8192 * KC));   // rows of `out` relation r has edges into: [rlo[r], rhi[r]] (empty: 1, 0)
  int* rhi = rlo + kGroupedMaxRel;
  const int tid = threadIdx.x, lane = tid & 63, xl = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = tid >> 4, c = tid & 15;
  const int gbase = lane & 48;  // first lane of the group inside its wave
  // the relations' row ranges, once per workgroup
  for (int r0 = 0; r0 < R; r0 += 64 * NW) {
    const int r = r0 + tid;
    int64_t so = 0;
    bool has = false;
    if constexpr (INL) {  // (a per-lane index into the kernel argument would copy it to scratch)
      for (int i = r0; i < R && i < r0 + 64 * NW; ++i) {
        const int64_t so_i = desc.irels[i].scatter_offset;
        const bool has_i = desc.irels[i].num_edges > 0;
        if (i == r) so = so_i, has = has_i;
      }
    } else if (r < R) {
      so = desc.rels[r].scatter_offset;
      has = desc.rels[r].num_edges > 0;
    }
    if (r < R) {
      int lo = 1, hi = 0;
      if (has) lo = (int)(so + desc.meta[2 * r]), hi = (int)(so + desc.meta[2 * r + 1]);
      rlo[r] = lo;
      rhi[r] = hi;
    }
  }
  if constexpr (ROWS < 32) {  // rows 16 ... 31 of the A tile are never written: zeros (their products are not stored)
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) *reinterpret_cast<u32x4*>(xs + kc * 8192 + 4096 + tid * 16) = z;
  }
  __syncthreads();

  // W of relation `g` into LDS buffer `buf`: the layout of rgcn_fused_kernel (32 blocks of 4 k-rows, 16-byte chunks
  // permuted inside a block), this wave's 4 blocks
  // (Lane constants of the phases behind the row loads -- W's DMA addresses, the MFMA operand base, the store addresses --
  // are recomputed from an opaque copy of the thread id where they are used: kept in registers over the loop they do not
  // fit, and as spills they are reloaded from scratch behind the row loads, i.e. the reload waits for the rows.)
  auto opaque_tid = [&]() __attribute__((always_inline)) -> int {
    int t = tid;
    asm volatile("" : "+v"(t));
    return t;
  };
  auto load_w = [&](int g, int kc, int mc) __attribute__((always_inline)) {
    const int lane = opaque_tid() & 63;
    const int dma_r = (lane & 15) >> 2, dma_ii = lane & 3, dma_u = lane >> 4;
    const int dma_c = 2 * dma_u + (dma_ii & 1) + 8 * (dma_ii >> 1);
    const char* wsrc = rel_at(g).weight + (size_t)(128 * kc + dma_r) * OB + mc * 256 + dma_c * 16;
#pragma unroll
    for (int j = 0; j < 32 / NW; ++j) {
      const int kb = wave * (32 / NW) + j;
      if constexpr (SMALL) {   // (K = RB / 2 rows of M = OB / 2 columns exist; what lies behind them is written as zeros: the
                               //  A tile is zero there, but 0 x a stale Inf of the previous chunk would be NaN)
        if (mc * 256 + dma_c * 16 < OB && 2 * (128 * kc + 4 * kb + dma_r) < RB) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + kb * 4 * OB),
                                           (LDSV*)(smem + kb * 1024), 16, 0, 0);
        } else {
          const u32x4 z = {0u, 0u, 0u, 0u};
          *reinterpret_cast<u32x4*>(smem + kb * 1024 + lane * 16) = z;
        }
      } else {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + kb * 4 * OB),
                                         (LDSV*)(smem + kb * 1024), 16, 0, 0);
      }
    }
  };
  // edges of the group's row among the 16 at `start`: lane c looks at edge start + c; the row's edges are a prefix
  auto prefix_count = [&](bool match) __attribute__((always_inline)) -> int {
    const uint64_t m = __ballot(match);
    const uint32_t g16 = (uint32_t)(m >> gbase) & 0xffffu;
    return __builtin_ctz(~g16 | 0x10000u);
  };
  typedef typename std::conditional<BIG, int64_t, int>::type RowT;   // (tables of fewer than 2^24 rows: 32 bits do)
  auto feature_row = [&](const RelDev& rel, int64_t g1, bool on) __attribute__((always_inline)) -> RowT {
    if (!on) return 0;
    int64_t g2;
    if (rel.gather_map) {
      if (CHECK && (g1 < 0 || g1 >= rel.map_len)) {
        *error = 1;
        g1 = 0;
      }
      g2 = ((GI64*)rel.gather_map)[g1];
    } else {
      g2 = g1 + rel.gather_offset;
    }
    // (validated here, on all 64 bits: narrowed to RowT first, an index of 2^32 + k would alias row k and pass issue_rows'
    // range check -- memory-safe, but a bad index the atomic kernel reports would go unnoticed)
    if (CHECK && (g2 < 0 || g2 >= rel.x_rows)) {
      *error = 1;
      g2 = 0;
    }
    return (RowT)g2;
  };
  u32x4 xr[KS][U];
  // the n <= 16 rows of one batch on their way (rows the batch does not have: zeros)
  auto issue_rows = [&](const RelDev& rel, int n, RowT g2, int kc, u32x4 (&xr)[U]) __attribute__((always_inline)) {
    if (CHECK && (g2 < 0 || g2 >= rel.x_rows)) {   // (what the map returned; lanes past n hold 0)
      *error = 1;
      g2 = 0;
    }
    // lane i of the group holds the row of edge i: a broadcast inside the 16-lane DPP row (v_mov_b32_dpp row_newbcast:i --
    // no LDS traffic and no per-edge address registers, which __shfl's ds_bpermute costs)
    if constexpr (BIG) {
      const int lo = (int)(uint32_t)g2, hi = (int)(uint32_t)((uint64_t)g2 >> 32);
      static_for<0, U>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const uint32_t rl = (uint32_t)row_bcast<i>(lo), rh = (uint32_t)row_bcast<i>(hi);
        const int64_t row = (int64_t)(((uint64_t)rh << 32) | rl);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (i < n && (!SMALL || kc * 256 + c * 16 < RB)) v = *(GU32x4*)(rel.x + row * RB + kc * 256 + c * 16);
        xr[i] = v;
      });
    } else {
      const int rowb = (int)((uint32_t)g2 * (uint32_t)RB);  // < 4 GB tables (checked on the host)
      static_for<0, U>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const uint32_t off = (uint32_t)row_bcast<i>(rowb) + (uint32_t)(kc * 256 + c * 16);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (i < n && (!SMALL || kc * 256 + c * 16 < RB)) v = *(GU32x4*)(rel.x + off);
        xr[i] = v;
      });
    }
  };
  float sum[KS][8];
  auto add_rows = [&](const u32x4 (&xr)[U], float (&sum)[8]) __attribute__((always_inline)) {  // in edge order
#pragma unroll
    for (int i = 0; i < U; ++i) {
      if constexpr (F32) {
#pragma unroll
        for (int k = 0; k < 4; ++k) sum[k] += u2f(xr[i][k]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float a, b;
          unpack2<BF16>(xr[i][k], &a, &b);
          sum[2 * k] += a;
          sum[2 * k + 1] += b;
        }
      }
    }
  };

  // ---- the items of this workgroup, in order (wave-uniform state) ---------------------------------------------------
  const int nblocks = (int)((out_rows + ROWS - 1) / ROWS);
  const int G = (int)gridDim.x;
  // (no flags in this state: two bools set in sibling branches were merged into one store through a selected pointer,
  // which kept them -- and with them the whole walk -- in scratch and in vector registers)
  // Workgroup w owns blocks w, w + G, w + 2 G, ... (the rows with edges are the first ones of every node type: contiguous
  // ranges, dealt out evenly this way) and walks them from a workgroup-specific START, wrapping around: walked from the
  // front by everybody, the whole chip alternated between phases that only gather (latency bound) and phases that only
  // write zeros (bandwidth bound).  (A scrambled block order mixes the phases too but deals the active blocks unevenly:
  // 0.152 ms instead of 0.070.)
  // (Dealing the blocks that can have items -- the union of the relations' row ranges -- by a dense number instead of by
  // position evens out the items per workgroup, 4 ... 8 instead of 3 ... 10 on the C5 sample, and made the launch SLOWER, 72 us
  // instead of 55: profiles/NOTES_r6.md section 4.  The launch is bound by what the memory system does with the random rows,
  // not by the longest chain of a workgroup.)
  const int it_n = nblocks > (int)blockIdx.x ? (nblocks - (int)blockIdx.x + G - 1) / G : 0;   // blocks of this workgroup
  const int it_k0 = it_n > 0 ? (int)(((uint32_t)blockIdx.x * 2654435761u >> 12) % (uint32_t)it_n) : 0;
  int it_j = -1, it_blk = -1, it_c0 = R;
  uint64_t it_mask = 0;
  auto chunk_mask = [&](int blk, int c0) __attribute__((always_inline)) -> uint64_t {
    const int row0 = blk * ROWS;
    bool relv = false;
    if (c0 + lane < R) relv = rhi[c0 + lane] >= row0 && rlo[c0 + lane] <= row0 + ROWS - 1;
    return __ballot(relv);
  };
  auto next_item = [&](int& blk, int& rel) __attribute__((always_inline)) -> bool {
    while (true) {
      if (it_mask != 0) {
        rel = it_c0 + __builtin_ctzll(it_mask);
        it_mask &= it_mask - 1;
        blk = it_blk;
        return true;
      }
      if (it_j >= it_n) return false;        // (stays there)
      if (it_j >= 0 && it_c0 + 64 < R) {      // the next 64 relations of the same block
        it_c0 += 64;
        it_mask = chunk_mask(it_blk, it_c0);
        continue;
      }
      if (++it_j >= it_n) return false;
      {
        int k = it_k0 + it_j;
        if (k >= it_n) k -= it_n;
        it_blk = (int)blockIdx.x + k * G;
      }
      it_c0 = 0;
      it_mask = chunk_mask(it_blk, 0);
      while (it_mask == 0 && it_c0 + 64 < R) {
        it_c0 += 64;
        it_mask = chunk_mask(it_blk, it_c0);
      }
      if (it_mask == 0) {  // nothing arrives in the block's rows
        const int t = opaque_tid(), grp = t >> 4, c = t & 15;
        const int64_t o = (int64_t)it_blk * ROWS + grp;
        uint32_t z0 = 0;
        asm volatile("" : "+v"(z0));   // (materialised here: hoisted, the four zero registers were spilled and reloaded)
        const u32x4 z = {z0, z0, z0, z0};
        if (!(PYG_ABL_ & 8) && o < out_rows) {
#pragma unroll
          for (int mc = 0; mc < MC; ++mc)
            if (!SMALL || mc * 256 + c * 16 < OB) __builtin_nontemporal_store(z, reinterpret_cast<u32x4*>(out + o * OB + mc * 256 + c * 16));
        }
        it_c0 = R;
      }
    }
  };

  // ---- pipeline registers -------------------------------------------------------------------------------------------
  bool v0 = false, v1 = false, v2 = false, v3 = false;   // item i, i + 1, i + 2, i + 3 exist (uniform)
  int blk0 = 0, blk1 = 0, blk2 = 0, blk3 = 0, rel0 = 0, rel1 = 0, rel2 = 0, rel3 = 0;
  int st2 = -1, st3 = -1;                                // first edge of the group's row (-1: the row has none), items i + 2, i + 3
  int n1 = 0;                                            // edges of the row (<= 16 in this walk), item i + 1
  RowT g2_1 = 0;                                         // feature row of lane c's edge of item i + 1
  int64_t g1_2 = 0;                                      // the two indices of lane c's edge of item i + 2 (the scatter index:
  int s1_2 = -1;                                         //  its low word -- the row-start launch has seen all 64 bits)
  // the group's row as relation `rel`'s scatter index (recomputed where it is needed: four pipeline registers less)
  auto row_of = [&](int blk, int rel) __attribute__((always_inline)) -> int {
    return (int)((int64_t)blk * ROWS + grp - rel_at(rel).scatter_offset);
  };
  auto issue_start = [&](int blk, int rel, int& st) __attribute__((always_inline)) {
    const int64_t o = (int64_t)blk * ROWS + grp;
    st = -1;
    if (o < out_rows && o >= rlo[rel] && o <= rhi[rel]) st = desc.rp[rp_off_at(rel) + (o - rel_at(rel).scatter_offset)];
  };

  f32x16 acc[MC];
#pragma unroll
  for (int mc = 0; mc < MC; ++mc)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mc][r] = 0.f;
  // A tile (rows = the block's 16 destinations) x W in LDS -> acc: wave t the 32 output columns of its n-tile
  auto product = [&](f32x16& acc, int kc) __attribute__((always_inline)) {
    if (NW > 4 && wave >= 4) return;
    const int lane = opaque_tid() & 63, xl = lane & 31, h = lane >> 5, q = lane & 15, grp16 = lane >> 4;
    const char* wbb = smem + 16384 * h + (4 * (q >> 2) + (grp16 & 1) + 2 * (q & 1)) * 16 + ((q & 3) >> 1) * 8;
    const char* xk = xs + kc * 8192;
#pragma unroll 2   // (not 8: the operands of all steps at once, next to the rows in flight, do not fit the register budget)
    for (int s = 0; s < 8; ++s) {
      const u32x4 xa = *reinterpret_cast<const u32x4*>(xk + (xl * 16 + ((8 * h + s) ^ (xl & 15))) * 16);
      const v4i16 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(wbb + (2 * s) * 1024 + wave * 256));
      const v4i16 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(wbb + (2 * s + 1) * 1024 + wave * 256));
      const u32x4 wa = __builtin_bit_cast(u32x4, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
      acc = mfma16<BF16>(wa, xa, acc);
    }
  };
  // the block's 16 x 128 results, rounded once, through the tile: wave t holds columns (8 h + 2 t + j) * 8 ... + 7 of row xl
  auto store_block = [&](f32x16 (&acc)[MC], int blk) __attribute__((always_inline)) {
    const int t = opaque_tid(), lane = t & 63, xl = lane & 31, h = lane >> 5, grp = t >> 4, c = t & 15;
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) {
      if (SMALL && mc >= mcn) break;
      if (NW == 4 || wave < 4) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          u32x4 pk;
#pragma unroll
          for (int i = 0; i < 4; ++i) pk[i] = pack2<BF16>(acc[mc][8 * j + 2 * i], acc[mc][8 * j + 2 * i + 1]);
          const int ch = 8 * h + 2 * wave + j;
          *reinterpret_cast<u32x4*>(xs + (xl * 16 + (ch ^ (xl & 15))) * 16) = pk;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mc][r] = 0.f;
      }
      __syncthreads();
      // (MC = 1: every lane reads back the very 16 bytes it writes as its group's part of the next A tile: no barrier behind it)
      const u32x4 v = *reinterpret_cast<const u32x4*>(xs + (grp * 16 + (c ^ (grp & 15))) * 16);
      const int64_t o = (int64_t)blk * ROWS + grp;
      if (o < out_rows && (!SMALL || mc * 256 + c * 16 < OB)) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out + o * OB + mc * 256 + c * 16));
      if (MC > 1) __syncthreads();   // (the next 128 columns go through the same tile)
    }
  };
  auto a_tile_from_sums = [&](int kc, const float (&sum)[8]) __attribute__((always_inline)) {
    const int t = opaque_tid(), grp = t >> 4, c = t & 15;
    u32x4 pk;
    if constexpr (F32) {   // 64 floats of row grp: slice kc, lane c's four
#pragma unroll
      for (int k = 0; k < 4; ++k) pk[k] = f2u(sum[k]);
      *reinterpret_cast<u32x4*>(xs + grp * kARow + kc * 256 + c * 16) = pk;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) pk[k] = pack2<BF16>(sum[2 * k], sum[2 * k + 1]);
      *reinterpret_cast<u32x4*>(xs + kc * 8192 + (grp * 16 + (c ^ (grp & 15))) * 16) = pk;
    }
  };
  // F32: W[32 q ... 32 q + 31][0 ... 127] (16 KB, contiguous in memory) into buffer q & 1, as it lies: four DMA instructions
  // per wave
  // (SMALL: rows of OB <= 512 bytes, K = RB / 4 of them: the piece is 32 OB bytes, its rows behind K are written as zeros)
  auto load_w32 = [&](int g, int q) __attribute__((always_inline)) {
    const int lane = opaque_tid() & 63;
    const int pbytes = SMALL ? 32 * OB : 16384;
    const char* wsrc = rel_at(g).weight + (size_t)q * pbytes + lane * 16;
    int valid = pbytes;
    if constexpr (SMALL) {
      const int krows = (RB >> 2) - 32 * q;
      valid = (krows >= 32 ? 32 : (krows > 0 ? krows : 0)) * OB;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int blk1k = wave * 4 + j;
      if (!SMALL || blk1k * 1024 + lane * 16 < valid) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + blk1k * 1024),
                                         (LDSV*)(smem + (q & 1) * 16384 + blk1k * 1024), 16, 0, 0);
      } else if (blk1k * 1024 + lane * 16 < pbytes) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(smem + (q & 1) * 16384 + blk1k * 1024 + lane * 16) = z;
      }
    }
  };
  float acc8[8];   // F32: row tid / 16, columns 8 (tid % 16) ... + 7 of the block's results
#pragma unroll
  for (int k = 0; k < 8; ++k) acc8[k] = 0.f;
  auto product32 = [&](int q) __attribute__((always_inline)) {
    const int t = opaque_tid(), r = t >> 4, cg = t & 15;
    const char* a = xs + r * kARow + q * 128;
    const char* w = smem + (q & 1) * 16384 + cg * 32;
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) {
      const float av = *reinterpret_cast<const float*>(a + kk * 4);
      const int wp = SMALL ? OB : 512;   // (a thread whose 8 columns lie behind M reads into the next row: results nobody stores)
      const u32x4 w0 = *reinterpret_cast<const u32x4*>(w + kk * wp), w1 = *reinterpret_cast<const u32x4*>(w + kk * wp + 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc8[k] = __builtin_fmaf(av, u2f(w0[k]), acc8[k]);
        acc8[4 + k] = __builtin_fmaf(av, u2f(w1[k]), acc8[4 + k]);
      }
    }
  };
  auto store_block32 = [&](int blk) __attribute__((always_inline)) {
    const int t = opaque_tid(), r = t >> 4, cg = t & 15;
    const int64_t o = (int64_t)blk * ROWS + r;
    u32x4 v0, v1;
#pragma unroll
    for (int k = 0; k < 4; ++k) v0[k] = f2u(acc8[k]), v1[k] = f2u(acc8[4 + k]);
    if (o < out_rows) {
      if (!SMALL || cg * 32 < OB) __builtin_nontemporal_store(v0, reinterpret_cast<u32x4*>(out + o * OB + cg * 32));
      if (!SMALL || cg * 32 + 16 < OB) __builtin_nontemporal_store(v1, reinterpret_cast<u32x4*>(out + o * OB + cg * 32 + 16));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc8[k] = 0.f;
  };

  // ---- fp32 in the pipeline: the product on v_mfma_f32_16x16x4_f32 (fp32 operands, fp32 accumulation) --------------------------
  // (product32 above reads 36 bytes of LDS per thread and k: 590 KB per 64 k-rows, ~2 us of the CU's LDS time per item -- with
  // three workgroups on the CU that was most of the kernel.  On the matrix pipe a wave reads its operands once: 48 KB.)
  // All of W (128 k-rows of 512 bytes) into the 64 KB buffer, 16 DMA instructions per wave, two k-rows each.  The 16-byte chunks
  // of row k are permuted -- chunk lc lies at lc ^ 4 ((k >> 2) & 3) -- so that the four k-rows one operand read touches fall
  // into different banks.
  auto load_w32m = [&](int g) __attribute__((always_inline)) {
    const int lane = opaque_tid() & 63;
    const int r = lane >> 5, pc = lane & 31;
    const char* wbase = rel_at(g).weight;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int i = wave * 16 + j;          // k-rows 2 i, 2 i + 1
      const int k = 2 * i + r;
      const int lc = pc ^ (4 * ((k >> 2) & 3));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + k * 512 + lc * 16),
                                       (LDSV*)(smem + i * 1024), 16, 0, 0);
    }
  };
  // lane (m = lane % 16, kq = lane / 16) of wave t: rows m of the A tile, columns 32 t + 16 nt + m of W; MFMA step j of a group of
  // 16 k-rows multiplies k = k0 + 4 kq + j (any assignment of k to the instruction's four slots does, as long as both operands
  // use it: this one makes the A operand of four steps ONE 16-byte read)
  f32x4 acc32[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc32[nt][k] = 0.f;
  auto product32m = [&]() __attribute__((always_inline)) {
    const int lane = opaque_tid() & 63, m = lane & 15, kq = lane >> 4;
    const char* ap = xs + m * kARow + kq * 16;
    // column 32 wave + 16 nt + m: chunk (8 wave + 4 nt + (m >> 2)) ^ 4 kq, float m & 3
    const char* wp = smem + (4 * kq) * 512 + (m & 3) * 4;
    const int ch0 = ((8 * wave + (m >> 2)) ^ (4 * kq)) * 16, ch1 = ((8 * wave + 4 + (m >> 2)) ^ (4 * kq)) * 16;
#pragma unroll 2
    for (int k0 = 0; k0 < 128; k0 += 16) {
      const u32x4 a4 = *reinterpret_cast<const u32x4*>(ap + k0 * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float b0 = *reinterpret_cast<const float*>(wp + (k0 + j) * 512 + ch0);
        const float b1 = *reinterpret_cast<const float*>(wp + (k0 + j) * 512 + ch1);
        const uint32_t aw = a4[j];
        acc32[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u2f(aw), b0, acc32[0], 0, 0, 0);
        acc32[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u2f(aw), b1, acc32[1], 0, 0, 0);
      }
    }
  };
  // results through the A tile's region (rows of kARow bytes): lane (m, kq) holds rows 4 kq ... 4 kq + 3 of column 32 t + 16 nt + m
  auto store_block32m = [&](int blk) __attribute__((always_inline)) {
    const int t = opaque_tid(), lane = t & 63, m = lane & 15, kq = lane >> 4, r = t >> 4, cg = t & 15;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        *reinterpret_cast<float*>(xs + (4 * kq + k) * kARow + (32 * wave + 16 * nt + m) * 4) = acc32[nt][k];
        acc32[nt][k] = 0.f;
      }
    __syncthreads();
    const u32x4 v0 = *reinterpret_cast<const u32x4*>(xs + r * kARow + cg * 32), v1 = *reinterpret_cast<const u32x4*>(xs + r * kARow + cg * 32 + 16);
    const int64_t o = (int64_t)blk * ROWS + r;
    if (o < out_rows) {
      __builtin_nontemporal_store(v0, reinterpret_cast<u32x4*>(out + o * OB + cg * 32));
      __builtin_nontemporal_store(v1, reinterpret_cast<u32x4*>(out + o * OB + cg * 32 + 16));
    }
    __syncthreads();   // (the next A tile is written by other lanes than the ones that read here)
  };

  if (long_rows) {
    // ---- item at a time: row start -> indices -> gather_map -> rows, 16 edges per batch, as many batches as the longest
    //      row of the wave needs ----------------------------------------------------------------------------------------
    int blk = 0, rel = 0, cur = -1;
    while (next_item(blk, rel)) {
      blk = __builtin_amdgcn_readfirstlane(blk);
      rel = __builtin_amdgcn_readfirstlane(rel);
      if (cur >= 0 && blk != cur) {
        if constexpr (F32) store_block32(cur);
        else store_block(acc, cur);
      }
      cur = blk;
      const RelDev& r = rel_at(rel);
      if constexpr (F32) load_w32(rel, 0);   // (buffer 0 was last read two barriers ago)
      else load_w(rel, 0, 0);   // (the previous item's products are behind a barrier)
      int st = -1;
      issue_start(blk, rel, st);
      const int d = row_of(blk, rel);
#pragma unroll 1
      for (int kc = 0; kc < KC; ++kc) {   // one walk over the row's edges per 128-feature slice
        if (SMALL && kc >= kcn) break;
#pragma unroll
        for (int k = 0; k < 8; ++k) sum[0][k] = 0.f;
        int sc = st;
        bool more = sc >= 0;
        do {
          const bool onc = more && (int64_t)sc + c < r.num_edges;
          int64_t s1c = -1, g1c = 0;
          if (onc) {
            s1c = ((GI64*)r.scatter_index)[sc + c];
            g1c = ((GI64*)r.gather_index)[sc + c];
          }
          const int nc = prefix_count(onc && s1c == d);
          const RowT g2c = feature_row(r, g1c, c < nc);
          issue_rows(r, nc, g2c, kc, xr[0]);
          add_rows(xr[0], sum[0]);
          more = more && nc == U;
          sc += U;
        } while (__any(more));
        a_tile_from_sums(kc, sum[0]);
      }
      if constexpr (F32) {
        // W in four pieces of 32 k-rows through two buffers: piece q + 1 travels while piece q is multiplied.  (A thread
        // reads the A rows its own wave wrote; the barrier is for W.)
        const int nq = SMALL ? ((RB >> 2) + 31) >> 5 : 4;
#pragma unroll 1
        for (int q = 0; q < nq; ++q) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (W's DMA is invisible to the compiler's counters)
          __syncthreads();   // piece q is there for everybody; everybody is done with piece q - 1
          if (q + 1 < nq) load_w32(rel, q + 1);
          product32(q);
        }
      } else {
#pragma unroll
        for (int mc = 0; mc < MC; ++mc) {
#pragma unroll 1
          for (int kc = 0; kc < KC; ++kc) {
            if (SMALL && (kc >= kcn || mc >= mcn)) continue;
            if (kc + mc > 0) load_w(rel, kc, mc);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (W's DMA is invisible to the compiler's counters)
            __syncthreads();
            product(acc[mc], kc);
            __syncthreads();
          }
        }
      }
    }
    if (cur >= 0) {
      if constexpr (F32) store_block32(cur);
      else store_block(acc, cur);
    }
    return;
  }
  if constexpr (!SMALL) {
  // The pipeline's unit is a SUB-ITEM (block, relation, 128-feature slice kc): KC of them per item, one after the other.  A
  // sub-item gathers slice kc of the row's feature rows (sub-items behind the first copy the row start, the indices and the
  // feature row from their predecessor -- one stage ahead of them in the pipeline, the same edge -- instead of fetching them
  // again) and multiplies its 16 x 128 sums with W[128 kc ... 128 kc + 127][:]: the 16-bit types in MC chunks of 128 columns
  // through the one W buffer (chunk 0 travels behind the previous sub-item's products like W of the K = M = 128 case, chunks
  // 1 ... MC - 1 are fetched between the products: one exposed trip to L2 each, which the row loads of the next sub-item share),
  // fp32 (WIDE) has no sub-items: both slices of the rows and all of W travel at once, one trip per item.
  bool primed = false;  // (the first iteration only fetches: one call site for the item walk, so that it is inlined and its
                        // state stays in scalar registers -- as a called function it lived in scratch, and every relation
                        // record was fetched with vector loads the row gathers had to wait behind)
  constexpr int KSUB = WIDE ? 1 : KC;        // sub-items of an item (WIDE: the slices travel together)
  int kc0 = 0, kc1 = 0, kc2 = 0, kc3 = 0;   // the slices of the four sub-items (uniform; KSUB = 1: always 0)
  int sub_blk = 0, sub_rel = 0, sub_kc = KSUB - 1;
  bool sub_on = false;
  auto next_sub = [&](int& blk, int& rel, int& kc) __attribute__((always_inline)) -> bool {
    if (KSUB > 1 && sub_on && sub_kc + 1 < KSUB) {
      ++sub_kc;
    } else {
      sub_on = next_item(sub_blk, sub_rel);
      sub_blk = __builtin_amdgcn_readfirstlane(sub_blk);
      sub_rel = __builtin_amdgcn_readfirstlane(sub_rel);
      sub_kc = 0;
    }
    blk = sub_blk, rel = sub_rel, kc = sub_kc;
    return sub_on;
  };
  // the first W chunk of sub-item (rel, kc) on its way
  auto load_w_first = [&](int rel, int kc) __attribute__((always_inline)) {
    if constexpr ((PYG_ABL_ & 1) != 0) return;
    if constexpr (F32) load_w32m(rel);
    else load_w(rel, kc, 0);
  };
  while (!primed || v0 || v1 || v2 || v3) {
    primed = true;
    if constexpr ((PYG_ABL_ & 16) != 0) abl_iters += 1, abl_items += v0 ? 1 : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // what the previous iteration asked for is here (W's DMA included)
    // (Every value the previous iteration loaded is USED here, in front of this iteration's first load: a use further down
    // would make the compiler wait for all loads issued in between -- it cannot count them across the conditional row
    // loads -- i.e. for the rows just requested.)
    asm volatile("" ::"v"(st3), "v"(g1_2), "v"(s1_2), "v"(g2_1));
    // ---- sub-item i: its rows have landed ---------------------------------------------------------------------------------
    if (v0) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int k = 0; k < 8; ++k) sum[ks][k] = 0.f;
        add_rows(xr[ks], sum[ks]);
        a_tile_from_sums(WIDE ? ks : 0, sum[ks]);   // (the previous sub-item's products are behind a barrier)
      }
    }
    // ---- sub-item i + 2: its indices have landed -> the row's edge count --------------------------------------------------
    const int n2 = prefix_count(v2 && st2 >= 0 && s1_2 == row_of(blk2, rel2));
    // ---- issue: nothing requested below is touched before the next iteration's top ----------------------------------------
    if (v1) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) issue_rows(rel_at(rel1), (PYG_ABL_ & 4) ? 0 : n1, g2_1, WIDE ? ks : kc1, xr[ks]);
    }
    RowT g2n = 0;
    if (v2) {
      if (KSUB > 1 && kc2 > 0) g2n = g2_1;   // (sub-item i + 1 is the same edge list's previous slice)
      else g2n = feature_row(rel_at(rel2), g1_2, c < n2);
    }
    int64_t g1n = 0;
    int s1n = -1;
    if (v3) {
      if (KSUB > 1 && kc3 > 0) {
        st3 = st2, s1n = s1_2, g1n = g1_2;
      } else {
        const RelDev& rel = rel_at(rel3);
        if (st3 >= 0 && (int64_t)st3 + c < rel.num_edges) {  // (a stale start of a row without edges may point anywhere)
          s1n = *(const __attribute__((address_space(1))) int*)(rel.scatter_index + (st3 + c));
          g1n = ((GI64*)rel.gather_index)[st3 + c];
        } else {
          st3 = -1;
        }
      }
    }
    int blk4 = 0, rel4 = 0, kc4 = 0, st4 = -1;
    const bool v4 = next_sub(blk4, rel4, kc4);
    if (v4 && (KSUB == 1 || kc4 == 0)) issue_start(blk4, rel4, st4);
    // ---- sub-item i: A tile, products; the block's rows when this was its last one ----------------------------------------
    if (!v0 && v1) load_w_first(rel1, kc1);   // (the pipeline is filling: nobody reads W)
    if (v0) {
      const bool last = !v1 || blk1 != blk0;   // of its block
      __syncthreads();
      if constexpr ((PYG_ABL_ & 2) != 0) {
      } else if constexpr (F32) {
        product32m();
      } else {
        product(acc[0], 0);
#pragma unroll
        for (int mc = 1; mc < MC; ++mc) {
          __syncthreads();   // everybody is done with the previous chunk
          if constexpr ((PYG_ABL_ & 1) == 0) load_w(rel0, kc0, mc);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          product(acc[mc], 0);
        }
      }
      __syncthreads();  // the A tile and W are free again
      if (v1) load_w_first(rel1, kc1);   // (one W buffer: three workgroups per CU; it lands behind this iteration's rows)
      if (last) {
        if constexpr (F32) store_block32m(blk0);
        else store_block(acc, blk0);
      }
    }
    // ---- rotate --------------------------------------------------------------------------------------------------------
    v0 = v1, blk0 = blk1, rel0 = rel1, kc0 = kc1;
    v1 = v2, blk1 = blk2, rel1 = rel2, kc1 = kc2, n1 = n2, g2_1 = g2n;
    v2 = v3, blk2 = blk3, rel2 = rel3, kc2 = kc3, st2 = st3, g1_2 = g1n, s1_2 = s1n;
    v3 = v4, blk3 = blk4, rel3 = rel4, kc3 = kc4, st3 = st4;
  }
#if PYG_ABL_ & 16
  {   // per workgroup: start, end (100 MHz), sub-items, iterations -- into the workspace behind `meta`
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
      uint64_t* dbg = desc.dbg + 4 * blockIdx.x;
      dbg[0] = abl_t0, dbg[1] = wall_clock64(), dbg[2] = (uint64_t)abl_items, dbg[3] = (uint64_t)abl_iters;
    }
  }
#endif
  }  // !SMALL
}

template <bool BF16, bool CHECK, bool BIG, bool INL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void rgcn_grouped_kernel(const GroupedDesc desc, int R, char* __restrict__ out,
                                                                                              int64_t out_rows, int* __restrict__ error) {
  rgcn_grouped_body<BF16, CHECK, BIG, INL, 4, 1, 1>(desc, R, out, out_rows, error);
}
// K = 128 KC, M = 128 MC (indices always validated: one instantiation less per shape)
template <bool BF16, bool BIG, bool INL, int KC, int MC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void rgcn_grouped_shape_kernel(const GroupedDesc desc, int R, char* __restrict__ out,
                                                                                                    int64_t out_rows, int* __restrict__ error) {
  rgcn_grouped_body<BF16, true, BIG, INL, 4, KC, MC>(desc, R, out, out_rows, error);
}

// 16-bit, K, M multiples of 8 up to 256 that the kernels above do not take
template <bool BF16, bool BIG, bool INL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void rgcn_grouped_small_kernel(const GroupedDesc desc, int R, char* __restrict__ out,
                                                                                                    int64_t out_rows, int* __restrict__ error) {
  rgcn_grouped_body<BF16, true, BIG, INL, 4, 2, 2, false, true>(desc, R, out, out_rows, error);
}

// fp32, K, M multiples of 4 up to 128 other than 128 x 128 (run-time row sizes, the item-at-a-time walk, plain FMAs)
template <bool BIG, bool INL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void rgcn_grouped_f32_small_kernel(const GroupedDesc desc, int R, char* __restrict__ out,
                                                                                                        int64_t out_rows, int* __restrict__ error) {
  rgcn_grouped_body<false, true, BIG, INL, 4, 2, 2, true, true>(desc, R, out, out_rows, error);
}

// fp32, K = M = 128
template <bool BIG, bool INL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void rgcn_grouped_f32_kernel(const GroupedDesc desc, int R, char* __restrict__ out,
                                                                                                  int64_t out_rows, int* __restrict__ error) {
  rgcn_grouped_body<false, true, BIG, INL, 4, 2, 2, true>(desc, R, out, out_rows, error);
}

// rows of `out` relation r may write: its destination segment (scatter_rows), cut at the end of `out`
inline int64_t grouped_span(const pyg_hip_rgcn_relation& rel, int64_t out_rows) {
  if (rel.num_edges <= 0) return 0;
  const int64_t to_end = out_rows - rel.scatter_offset;
  const int64_t span = rel.scatter_rows > 0 ? std::min(rel.scatter_rows, to_end) : to_end;
  return span > 0 ? span : 0;
}

size_t grouped_workspace_bytes(const pyg_hip_rgcn_relation* rels, int64_t R, int64_t out_rows) {
  size_t rp = 0;
  for (int64_t r = 0; r < R; ++r) rp += (size_t)grouped_span(rels[r], out_rows);
  const size_t Rz = (size_t)std::max<int64_t>(R, 1);
  return align_up(sizeof(RelDev) * Rz, 256) + align_up(sizeof(int64_t) * (3 * Rz + 1), 256) + align_up(sizeof(int32_t) * 2 * Rz, 256) + 256 +
         align_up(sizeof(int32_t) * rp, 256) + ((PYG_ABL_ & 16) ? 65536 : 0);
}
