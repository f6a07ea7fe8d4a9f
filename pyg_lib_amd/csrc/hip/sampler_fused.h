// Fused chain of the fully queued sampler mode (included by sampler.hip inside its anonymous namespace).
//
// Round 2's fully queued mode ran, per (hop, relation): count scan (2 launches) -> sample -> first-occurrence scan
// (2 launches) -> finalize, each a 5 - 10 us kernel behind a dependent launch boundary; 22 launches per C3 batch, ~90
// for a 7-relation hetero batch.  What the reference's sequential loop (neighbor_kernel.cpp:332-514, 518-841) really
// orders is less than that:
//   * the degree of a node is known WHERE IT IS APPENDED.  The first-occurrence scan of hop l therefore carries, next
//     to the rank (= local id), the (edges, RNG transition table) pair of the new node for every relation that will
//     expand its type in hop l + 1: its prefix IS hop l + 1's count scan (edge offset + engine position of every
//     frontier node), and its total the relation's emitted-edge count -- no separate count scan, no second gather of
//     rowptr[node];
//   * the engine position at which a relation starts is a fold of the totals of everything before it: all of them
//     are known when its hop starts, so every relation of a hop samples in ONE launch;
//   * relations that append to the same node type share its table: emission positions are ordered across them by a
//     per-relation base, so atomicMin(position) still finds the reference's first occurrence, and one scan per node
//     type (its relations' tiles concatenated in relation order) hands out the ids;
//   * finalize (local ids of every emitted edge) only reads table entries that already hold final ids, which a later
//     atomicMin(provisional position) never changes: it rides in the NEXT hop's sampling launch.
// Per hop: [finalize(l - 1) | sample(l)] -> reduce(l) -> apply(l): three launches whatever the number of relations;
// C3: 13 launches instead of 22.  State is write-once tables (sizes per hop and type, totals per hop and relation):
// nothing a running launch reads is written by it, so there is no versioning and no chain kernel.  Every launch is a
// list of work items whose records travel IN THE KERNEL ARGUMENT (<= 3 KB: no staging copy, and a block needs one
// dependent load -- its counts from the tables -- before its data, like round 2's kernels).
// Same outputs, same generator advance: every test of the fully queued mode runs through this path
// (PYG_HIP_SAMPLER_FUSED=0 selects round 2's chain for A/B timing).

constexpr int kMaxCons = 3;         // relations of the next hop that expand one node type; more: round 2's chain
                                    // (with 4 the reduce pass holds 64 row bounds per thread and spills)
constexpr int kMaxParts = 8;        // relations queued per hop; more: round 2's chain
constexpr int kMaxLaunchCons = 12;  // consumers over all node types of one scan launch
constexpr int kOnePassMaxTiles = 256;  // scans of at most this many tiles run as one launch (see fused_onepass)
constexpr int kAggStride = 8;       // one-pass scan: 64-bit words between two tiles' aggregates (one 64-byte line each: every
                                    // block reads every aggregate in front of it, and a dense array is a few KB, i.e. a
                                    // handful of memory channels for a few hundred thousand uncached reads)

template <int NC>
struct FusedAgg {
  int64_t rank;
  CountAgg next[NC];
};
template <>
struct FusedAgg<0> {
  int64_t rank;
};

template <int NC>
struct FusedOp {
  __device__ FusedAgg<NC> operator()(const FusedAgg<NC>& a, const FusedAgg<NC>& b) const {
    FusedAgg<NC> r;
    r.rank = a.rank + b.rank;
    if constexpr (NC > 0) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        // pure (16-bit draws only) tables compose by addition, inline; a table that holds a wider draw -- a row of degree
        // >= 2^16 -- takes the out-of-line general composition (NOT commutative: every reduction below keeps input order)
        r.next[c].edges = a.next[c].edges + b.next[c].edges;
        r.next[c].tab = rng_compose(a.next[c].tab, b.next[c].tab);
      }
    }
    return r;
  }
  __device__ static FusedAgg<NC> identity() {
    FusedAgg<NC> r;
    r.rank = 0;
    if constexpr (NC > 0) {
#pragma unroll
      for (int c = 0; c < NC; ++c) r.next[c] = CountOp::identity();
    }
    return r;
  }
};

// write-once tables of one call (device; copied to the host once, behind the last launch)
struct FTables {
  int64_t* size_at;   // [(L + 1) * T]: length of type t's node list at the start of hop h (h = 0: the seeds)
  int64_t* dup;       // [T]: seed positions that repeat an earlier seed (list positions - distinct nodes)
  CountAgg* tot;      // [L * R]: emitted edges + RNG transition table of (hop, relation)
  int32_t* overflow;  // [L * R]: 0 ok, 1 this relation lacked random words, 2 something before it did
  int32_t* wide;      // [1]: a sampled row of degree >= 2^16 was met (draws wider than 16 bits): the chain stops, the
                      // call is repeated through round 2's chain, which carries the general transition tables
  int64_t word0;      // engine position at the start of the call
  int units0;
  int L, R, T;
  int is32;           // some relation's rowptr / col are int32 (a call's relations share the index type)
};

struct FConsumer {  // a relation of the NEXT hop that expands the segment's node type
  RangeCtx range;
  int64_t count;
  int replace;
  int tot_index;      // (l + 1) * R + e'
  int64_t* edge_off;  // [frontier bound of (l + 1, e')]: exclusive prefix of the emitted edges
  RngTab* tabp;       //                                  exclusive prefix of the RNG tables
};

struct LastTile {
  u64 bits[16];        // bit b of word j: position 64 j + b of the tile is the first occurrence of a new node
  uint32_t before[16]; // first occurrences of the tile in front of word j
};

struct FSegHdr {  // one node type in one phase: the emissions of every relation that appends to it, in relation order
  u64* vals;
  u64 prov;                // value coding of the type's table (HashTable::prov / tag / idmask)
  u64 tag;
  u64 idmask;
  int64_t* nodes;
  int64_t* batch;          // disjoint: batch ids of the list, else nullptr
  const int64_t* size_in;  // hop: &size_at[l * T + t]; seeds: nullptr
  int64_t* size_out;       // hop: &size_at[(l + 1) * T + t]; seeds: &size_at[t]
  int64_t* dup;            // &dup[t]
  void* tile_agg;          // FusedAgg<NC>[tiles of all parts] (one-pass scan: zeroed at the start of the call)
  // last hop (fused_last_*): per tile of the segment the bitmap of its first occurrences + their count in front of every
  // 64-position word, and the tile's count -- written by the first pass, read by the second (no zeroing needed)
  struct LastTile* last_meta;
  uint32_t* last_cnt;
  int ntiles;              // tiles of the segment
  int seeds;
  int ncons;
  int pad;
};

struct FPart {  // one relation's emissions of the hop (or one seed set); carries its segment's header
  FSegHdr h;
  const u64* slots;
  const int64_t* e_node;
  const int64_t* e_batch;
  u64* cache;        // per emission: first-occurrence flag + the consumers' counts, kept from reduce for apply
  int64_t n_fixed;   // seeds: number of seeds; relations: -1 (n = tot[tot_index].edges)
  int64_t pos_base;  // emission position of this part's p = 0 within the segment
  int tot_index;
  int tile0;         // first tile of this part within the segment
  int last;          // last part of the segment: its last block publishes the segment's totals
  int cons0;         // first consumer of the segment in the launch's consumer array
};

struct FSeedFold {  // seeds launch of a call whose seeds are ONE tile: the table initialisation and the seeds' insertion
                    // (fused_init_kernel, seed_insert_kernel) run in front of the scan, in the same single block
  const int64_t* seed;  // nullptr: not folded
  int64_t batch0;
  int64_t num_batches;
  HashTable table;
  TypeState* ts;
  unsigned* sync;       // the call's tickets + tile aggregates
  int sync_words;
  int disjoint;
};

struct FFoldRec {  // what the fold role hands to the host, straight into pinned memory (no copies behind the last launch)
  MtHandBack* hb;            // engine hand-back (nullptr: the caller's engine is not device-resident)
  int64_t a0, generated32;   // mt_finish_chain_kernel's arguments
  char* tables_host;         // copy of the write-once tables
  const char* tables_dev;
  int tables_bytes;
  int pad;
  // completion word in pinned memory: set to `done_seq` when everything above has been written -- the host polls it
  // instead of waiting for the stream (the wake-up of hipStreamSynchronize costs 6 - 8 us of a ~180 us call)
  unsigned long long* done;
  unsigned long long done_seq;
};

struct FScanLaunch {  // kernel argument of a phase's scan (one pass; or the reduce / apply pair)
  FTables tb;
  unsigned* ticket;        // one-pass scan: blocks take their position in the launch from here (zeroed with the aggregates)
  FSeedFold fold;
  int n;                   // items: the parts, then (apply pass of a hop) the carry block
  int ell;                 // hop (carry)
  unsigned type_mask_lo, type_mask_hi;  // carry: bit t set = type t has a segment (its last block writes the size)
  int cum[kMaxParts + 1];  // inclusive prefix of the items' block counts
  int nc[kMaxParts + 1];   // consumers of the part's segment; -1: the carry block
  FPart part[kMaxParts];
  FConsumer cons[kMaxLaunchCons];
  int64_t* out_col[kMaxParts];   // last hop (fused_last_apply): base of the part's relation's `col` output
  // last hop: the closing role (engine position, tables and hand-back to pinned memory, completion word) rides in the
  // second pass as one more block (item nc == -2): it needs nothing the other blocks of that launch produce -- the hop's
  // node counts are the sums of the tile counts the FIRST pass wrote
  ChainState* chain;
  FFoldRec closing;
  const u64* words;
};
static_assert(sizeof(FScanLaunch) <= 4096, "launch records travel in the kernel argument");

struct FSampleRec {  // sampling of relation e in hop l: what does not depend on the tables
  const int64_t* nodes;  // src node list
  const int64_t* batch;  // src batch ids (disjoint) or nullptr
  RangeCtx range;
  int64_t count;
  int64_t num_batches;
  const int64_t* edge_off;
  const RngTab* tabp;
  int64_t* e_row;
  int64_t* e_node;
  int64_t* e_batch;
  int64_t* e_eid;
  u64* e_slot;
  HashTable table;
  int64_t pos_base;
  int replace;
  int ell, e, t_src;
};

struct FFinalRec {  // local ids of relation e's emissions of hop l
  const u64* slots;
  const u64* vals;
  u64 idmask;        // HashTable::idmask of the table
  int64_t* out_col;  // base of the relation's col output
  int ell, e;
};

enum FRole { kRoleSample8 = 0, kRoleSample16, kRoleSample32, kRoleSample64, kRoleFinalize, kRoleFold, kRoleSampleWave };


struct FSampleLaunch {  // kernel argument of [finalize(l - 1) | sample(l)] (+ the engine fold behind the last hop)
  FTables tb;
  ChainState* chain;  // fold
  FFoldRec fold;
  int n;
  int pad;
  int cum[2 * kMaxParts + 2];
  unsigned char role[2 * kMaxParts + 2];
  unsigned char idx[2 * kMaxParts + 2];  // index into s[] / f[]
  FSampleRec s[kMaxParts];
  FFinalRec f[kMaxParts];
};

static_assert(sizeof(FSampleLaunch) <= 4096, "launch records travel in the kernel argument");

// An argument record lies in device memory, where the old kernels had it as a kernel argument (= in SGPRs).  Read
// dword-wise through v_readfirstlane the compiler knows every field is wave-uniform: addresses and branch conditions
// built from it stay scalar (without this the sampling role needed 80 VGPRs instead of 35: 5 waves per SIMD instead of
// 8 on a kernel that is nothing but dependent gathers).
template <typename T>
__device__ __forceinline__ T uniform_record(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "dword records");
  T out;
  const uint32_t* s = reinterpret_cast<const uint32_t*>(p);
  uint32_t* d = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 4); ++i) d[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s[i]);
  return out;
}

// transition table of `count` draws from a row of `deg` >= 2^16 neighbours (CountLoad::operator(), sampler.hip)
__device__ __attribute__((noinline)) RngTab wide_row_table(int64_t deg, int64_t count, int replace) {
  RngTab t = rng_identity();
  if (replace) {
    const int n = need_units((u64)deg);
    for (int64_t j = 0; j < count; ++j) rng_push_draw(t, n);
  } else {
    for (int64_t j = deg - count; j < deg; ++j) rng_push_draw(t, need_units((u64)(j + 1)));
  }
  return t;
}

// (edges, RNG table) of node v for one consumer: CountLoad::operator() for a node that is being appended
__device__ __forceinline__ CountAgg consumer_count_bounds(const FConsumer& cs, int64_t rs0, int64_t re0, int64_t batch_id,
                                                          int32_t* wide) {
  CountAgg r;
  r.tab = rng_identity();
  r.edges = 0;
  int64_t rs, re;
  cs.range.narrow(rs0, re0, batch_id, cs.count, &rs, &re);
  const int64_t deg = re - rs;
  const int64_t count = cs.count;
  if (deg <= 0 || count == 0) return r;
  if (count < 0 || (!cs.replace && count >= deg)) {
    r.edges = deg;
    return r;
  }
  r.edges = count;
  // a row of degree >= 2^16 draws 32-bit numbers (rand_engine.h:44-50): its table is the packed 5-state form, built out
  // of line (inlined, the table arithmetic cost the scan kernels 250 registers + 2.4 KB of scratch per lane)
  r.tab = (u64)deg < (1ull << 16) ? tab_pure(count) : wide_row_table(deg, count, cs.replace);
  (void)wide;
  return r;
}

// cache word of an emission: bit 0 = first occurrence; consumer c at bits [1 + 15 c, 16 + 15 c): edges (12 bits, <=
// kMaxFusedCount), bit 12 = sampled (count 16-bit draws), bit 13 = sampled from a row of degree >= 2^16 (wider draws)
__device__ __forceinline__ u64 cons_encode(const CountAgg& a) {
  u64 code = (u64)a.edges & 0xfff;
  if (a.tab != rng_identity()) code |= 1u << 12;
  if (!tab_is_pure(a.tab)) code |= 1u << 13;  // a wide row: the apply pass rebuilds its table from the row bounds
  return code;
}

__device__ __forceinline__ int64_t part_count(const FScanLaunch& L, const FPart& pt) {
  if (pt.n_fixed >= 0) return pt.n_fixed;
  const int over = L.tb.overflow[pt.tot_index];
  const int64_t edges = L.tb.tot[pt.tot_index].edges;
  return __builtin_amdgcn_readfirstlane(over) ? 0 : edges;
}

// where relation e's `col` of hop `ell` starts inside its output: behind its edges of the hops before (every lane the same)
__device__ __forceinline__ int64_t part_col_offset(const FScanLaunch& L, const FPart& pt) {
  const int lane = threadIdx.x & 63;
  const int e = pt.tot_index - L.ell * L.tb.R;
  int64_t mine = 0;
  if (lane < L.ell) mine = L.tb.tot[lane * L.tb.R + e].edges;
  int64_t off = 0;
  for (int h = 0; h < L.ell; ++h) off += __shfl(mine, h);
  return off;
}

template <int NC>
struct FCons {
  FConsumer c[NC > 0 ? NC : 1];
  // I64: the call's graphs are int64 -- said to the compiler by overwriting the (then known to be 0) width flags of the
  // local copies.  IdxArr::operator[] otherwise is a branch per access whose 32-bit arm converts what it loaded right
  // away, i.e. waits for it: the 24 row-bound gathers of a 3-consumer scan ran one after the other (47 -> 24 us was the
  // scratch copy, the rest of the distance to the 1-consumer scan's 14 us was this).
  template <bool I64>
  __device__ __forceinline__ void load(const FScanLaunch& L, int first) {
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      c[i] = L.cons[first + i];
      if constexpr (I64) {
        c[i].range.rowptr.is32 = 0;
        c[i].range.col.is32 = 0;
      }
    }
  }
};

// ---- reduce: tile aggregates of (first-occurrence flag, next-hop counts) -----------------------------------------
template <int NC, bool I64>
__device__ __forceinline__ void fused_reduce(const FScanLaunch& L, const FPart& pt, int lt) {
  typedef FusedAgg<NC> T;
  typedef FusedOp<NC> Op;
  __shared__ T lds[8];
  FCons<NC> cons;
  cons.template load<I64>(L, pt.cons0);
  const int64_t n = part_count(L, pt);
  const int64_t base = (int64_t)lt * kScanTile + threadIdx.x * kScanItems;
  Op op;
  T agg = Op::identity();
  if (base < n) {
    // two rounds of dependent loads: [slot, node, batch] of every item, then [table value, the consumers' row bounds]
    // (the bounds are fetched whether or not the emission turns out to be a first occurrence: off the critical path)
    u64 sl[kScanItems], vv[kScanItems];
    int64_t nd[kScanItems], bt[kScanItems];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      const int64_t pc = base + k < n ? base + k : n - 1;
      sl[k] = pt.slots[pc];
      if constexpr (NC > 0) {
        nd[k] = pt.e_node[pc];
        bt[k] = pt.e_batch ? pt.e_batch[pc] : 0;
      }
    }
    T v[kScanItems];
    int64_t r0[kScanItems][NC > 0 ? NC : 1], r1[kScanItems][NC > 0 ? NC : 1];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      vv[k] = pt.h.vals[sl[k]];
      if constexpr (NC > 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          r0[k][c] = cons.c[c].range.rowptr[nd[k]];
          r1[k][c] = cons.c[c].range.rowptr[nd[k] + 1];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      v[k] = Op::identity();
      if constexpr (NC > 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) v[k].next[c] = consumer_count_bounds(cons.c[c], r0[k][c], r1[k][c], bt[k], L.tb.wide);
      }
    }
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      const int64_t p = base + k;
      if (p >= n) break;
      const bool flag = vv[k] == pt.h.prov + (u64)(pt.pos_base + p);
      v[k].rank = flag ? 1 : 0;
      u64 word = flag ? 1ull : 0ull;
      if constexpr (NC > 0) {
        if (flag || pt.h.seeds) {
#pragma unroll
          for (int c = 0; c < NC; ++c) word |= cons_encode(v[k].next[c]) << (1 + 15 * c);
        } else {
#pragma unroll
          for (int c = 0; c < NC; ++c) v[k].next[c] = CountOp::identity();
        }
      }
      pt.cache[p] = word;
      agg = op(agg, v[k]);
    }
  }
  T total;
  (void)block_exclusive<T, Op>(agg, lds, op, &total);
  if (threadIdx.x == 0) static_cast<T*>(pt.h.tile_agg)[pt.tile0 + lt] = total;
}

// ---- apply: ids, node-list append, the next hop's per-node prefixes; the segment's last block publishes its totals
template <int NC, bool I64>
__device__ __forceinline__ void fused_apply(const FScanLaunch& L, const FPart& pt, int lt, int nblocks) {
  typedef FusedAgg<NC> T;
  typedef FusedOp<NC> Op;
  __shared__ T lds[8];
  FCons<NC> cons;
  cons.template load<I64>(L, pt.cons0);
  const int64_t n = part_count(L, pt);
  const int64_t size0 = pt.h.seeds ? 0 : *pt.h.size_in;
  const int64_t id0 = pt.h.seeds ? 0 : size0 - *pt.h.dup;
  const int64_t base = (int64_t)lt * kScanTile + threadIdx.x * kScanItems;
  Op op;
  T v[kScanItems];
  u64 word[kScanItems];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t p = base + k;
    v[k] = Op::identity();
    word[k] = 0;
    if (p < n) {
      word[k] = pt.cache[p];
      v[k].rank = (int64_t)(word[k] & 1);
      if constexpr (NC > 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const u64 code = (word[k] >> (1 + 15 * c)) & 0x7fff;
          v[k].next[c].edges = (int64_t)(code & 0xfff);
          v[k].next[c].tab = (code & (1u << 12)) ? tab_pure(cons.c[c].count) : rng_identity();
          if (code & (1u << 13)) {  // rare: a wide row -- its table depends on the degree: fetch the row bounds again
            const int64_t nd = pt.e_node[p];
            const int64_t bt = pt.e_batch ? pt.e_batch[p] : 0;
            v[k].next[c] = consumer_count_bounds(cons.c[c], cons.c[c].range.rowptr[nd], cons.c[c].range.rowptr[nd + 1], bt,
                                                 L.tb.wide);
          }
        }
      }
    }
  }
  T agg = Op::identity();
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) agg = op(agg, v[k]);
  T total;
  T run = block_exclusive<T, Op>(agg, lds, op, &total);
  // every block reduces the aggregates of the tiles in front of it (of the whole segment) for itself: every thread a
  // CONTIGUOUS run of tiles, then ONE ordered block reduction (instead of a block scan per 256 tiles) -- tile order is
  // kept, because the composition of tables that hold wide draws does not commute
  T before = Op::identity();
  const int ntb = pt.tile0 + lt;
  const T* tiles = static_cast<const T*>(pt.h.tile_agg);
  {
    T part = Op::identity();
    const int per = (ntb + kScanThreads - 1) / kScanThreads;
    const int i0 = (int)threadIdx.x * per, i1 = min(i0 + per, ntb);
    for (int i = i0; i < i1; ++i) part = op(part, tiles[i]);
    (void)block_exclusive<T, Op>(part, lds, op, &before);
  }
  run = op(before, run);
  if (pt.last && lt == nblocks - 1 && threadIdx.x == 0) {
    const T grand = op(before, total);
    if (pt.h.seeds) {
      *pt.h.size_out = pt.n_fixed;
      *pt.h.dup = pt.n_fixed - grand.rank;
    } else {
      *pt.h.size_out = size0 + grand.rank;
    }
    if constexpr (NC > 0) {
#pragma unroll
      for (int c = 0; c < NC; ++c) L.tb.tot[cons.c[c].tot_index] = grand.next[c];
    }
  }
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t p = base + k;
    if (p < n) {
      const bool flag = (word[k] & 1) != 0;
      if (flag) {
        pt.h.vals[pt.slots[p]] = pt.h.tag | (u64)(id0 + run.rank);
        if (!pt.h.seeds) {
          pt.h.nodes[size0 + run.rank] = pt.e_node[p];
          if (pt.h.batch) pt.h.batch[size0 + run.rank] = pt.e_batch[p];
        }
      }
      if constexpr (NC > 0) {
        if (flag || pt.h.seeds) {
          const int64_t i = pt.h.seeds ? p : run.rank;  // index in the next hop's frontier
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            cons.c[c].edge_off[i] = run.next[c].edges;
            cons.c[c].tabp[i] = run.next[c].tab;
          }
        }
      }
    }
    run = op(run, v[k]);
  }
}

// ---- one pass: reduce and apply of a tile by the same block -----------------------------------------------------------
// The reduce / apply pair above costs a launch boundary and a second trip of every emission's (flag, counts) word
// through memory.  Here a block keeps its items in registers, publishes its tile aggregate and then looks back: it
// waits for and sums the aggregates of every tile in front of it in its segment.  The blocks take
// their position from a ticket counter, so a block only ever waits for blocks that started before it (no assumption on
// the dispatch order, nothing to deadlock on).  The waiting is all-pairs -- every block polls every aggregate in front of
// it, uncached -- which is the cheaper form up to a few hundred tiles (16 tiles: 11 - 13 us against 17 for the pair of
// launches, 151 tiles: 16.5 against 19) and the dearer one beyond (750 tiles: 32 - 37 us against 29.5: the polls compete
// with the gathers of the blocks that still reduce); the host picks per launch (kOnePassMaxTiles).  What apply writes while other blocks still reduce does not disturb them:
// a table entry turns from its owner's provisional position into a final id, and a non-owner's test
// `value == provisional + my position` fails on either.
// FOLD: the seeds of the call are this single block's tile -- it first initialises the call's tables and inserts the seeds
// (fused_init_kernel + seed_insert_kernel), slot handles, nodes and batch ids staying in registers.
__device__ __forceinline__ void fused_tables_init(const FTables& tb, int i) {
  if (i < (tb.L + 1) * tb.T) tb.size_at[i] = 0;
  if (i < tb.T) tb.dup[i] = 0;
  if (i < tb.L * tb.R) {
    tb.tot[i] = CountOp::identity();
    tb.overflow[i] = 0;
  }
  if (i == 0) *tb.wide = 0;
}

#ifdef PYG_HIP_FOLD_TIMING  // experiment builds only: 100 MHz stamps of the seeds launch's phases (thread 0)
__device__ unsigned long long g_fold_stamps[16];
__device__ unsigned long long g_kstamps[2 * 64];
__device__ unsigned int g_kstamp_n;
__device__ __forceinline__ void dbg_kstamp(unsigned long long id) {
  const unsigned i = atomicAdd(&g_kstamp_n, 1u) & 63u;
  g_kstamps[2 * i] = id;
  g_kstamps[2 * i + 1] = wall_clock64();
}
#define PYG_FOLD_STAMP(i) do { if (FOLD && threadIdx.x == 0) g_fold_stamps[i] = wall_clock64(); } while (0)
#else
#define PYG_FOLD_STAMP(i) do { } while (0)
#endif

template <int NC, bool FOLD, bool I64>
__device__ __forceinline__ void fused_onepass(const FScanLaunch& L, const FPart& pt, int lt, int nblocks) {
  typedef FusedAgg<NC> T;
  typedef FusedOp<NC> Op;
  __shared__ T lds[8];
  FCons<NC> cons;
  cons.template load<I64>(L, pt.cons0);
  const int tid = (int)threadIdx.x;
  const int64_t n = FOLD ? pt.n_fixed : part_count(L, pt);
  const int64_t base = (int64_t)lt * kScanTile + tid * kScanItems;
  Op op;
  u64 sl[kScanItems];
  int64_t nd[kScanItems], bt[kScanItems];
  T v[kScanItems];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    sl[k] = 0;
    nd[k] = 0;
    bt[k] = 0;
    v[k] = Op::identity();
  }
  PYG_FOLD_STAMP(0);
  if constexpr (FOLD) {
    const FSeedFold& f = L.fold;
    const int cells = max((L.tb.L + 1) * L.tb.T, L.tb.L * L.tb.R);
    for (int i = tid; i < cells; i += kScanThreads) fused_tables_init(L.tb, i);
    for (int i = tid; i < f.sync_words / 4; i += kScanThreads) reinterpret_cast<uint4*>(f.sync)[i] = make_uint4(0, 0, 0, 0);
    if (tid == 0) {
      f.ts->size = n;
      f.ts->slice_e = n;
    }
    PYG_FOLD_STAMP(1);
    if (base < n) {
#pragma unroll
      for (int k = 0; k < kScanItems; ++k) {
        const int64_t p = base + k < n ? base + k : n - 1;
        nd[k] = f.seed[p];
        bt[k] = f.disjoint ? f.batch0 + p : 0;
      }
#pragma unroll
      for (int k = 0; k < kScanItems; ++k) {
        const int64_t p = base + k;
        sl[k] = table_slot(f.table, make_key(nd[k], bt[k], f.num_batches));
        if (p < n) {
          pt.h.nodes[p] = nd[k];
          if (f.disjoint) pt.h.batch[p] = bt[k];
          __hip_atomic_fetch_min(&f.table.vals[sl[k]], f.table.prov + (u64)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    // every initialising store and every insertion has been performed before anybody (of this single block) reads a table
    // value or overwrites a table cell (the segment's totals go where the initialisation wrote).  Workgroup scope: the wait
    // for the stores' and atomics' acknowledgements in front of the barrier is what orders them -- the insertions are
    // agent-scope atomics and the reads below agent-scope atomic loads, both performed at L2; an AGENT-scope fence here
    // additionally wrote back and invalidated this XCD's whole L2 (whatever the previous kernels left dirty in it).
    PYG_FOLD_STAMP(2);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    PYG_FOLD_STAMP(3);
  }
  T agg = Op::identity();
  if (base < n) {
    if constexpr (!FOLD) {
#pragma unroll
      for (int k = 0; k < kScanItems; ++k) {
        const int64_t pc = base + k < n ? base + k : n - 1;
        sl[k] = pt.slots[pc];
        nd[k] = pt.e_node[pc];  // (a hop's last scan carries no consumers but still appends the node)
        bt[k] = pt.e_batch ? pt.e_batch[pc] : 0;
      }
    }
    u64 vv[kScanItems];
    int64_t r0[kScanItems][NC > 0 ? NC : 1], r1[kScanItems][NC > 0 ? NC : 1];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      vv[k] = FOLD ? __hip_atomic_load(&pt.h.vals[sl[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : pt.h.vals[sl[k]];
      if constexpr (NC > 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          r0[k][c] = cons.c[c].range.rowptr[nd[k]];
          r1[k][c] = cons.c[c].range.rowptr[nd[k] + 1];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      const int64_t p = base + k;
      if (p >= n) break;
      const bool flag = vv[k] == pt.h.prov + (u64)(pt.pos_base + p);
      v[k].rank = flag ? 1 : 0;
      if constexpr (NC > 0) {
        if (flag || pt.h.seeds) {
#pragma unroll
          for (int c = 0; c < NC; ++c) v[k].next[c] = consumer_count_bounds(cons.c[c], r0[k][c], r1[k][c], bt[k], L.tb.wide);
        }
      }
      agg = op(agg, v[k]);
    }
  }
  PYG_FOLD_STAMP(4);
  T total;
  T run = block_exclusive<T, Op>(agg, lds, op, &total);
  PYG_FOLD_STAMP(5);
  T before = Op::identity();
  if constexpr (!FOLD) {
    // Aggregates travel as relaxed agent-scope atomics of self-validating words (rank + 1, edges + 1, a pure table has
    // bit 63 set: never 0, and the array starts zeroed) -- per-location coherence is all the protocol needs, so there
    // is no release / acquire pair, which on this chip means an L2 write-back and an L2 invalidate per block.
    const int ti = pt.tile0 + lt;
    u64* words = static_cast<u64*>(pt.h.tile_agg);
    constexpr int W = 1 + 2 * NC;
    static_assert(W <= kAggStride, "a tile's aggregate has its own 64-byte line");
    if (tid == 0) {
      u64* w = words + (size_t)ti * kAggStride;
      __hip_atomic_store(&w[0], (u64)total.rank + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if constexpr (NC > 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          __hip_atomic_store(&w[1 + 2 * c], (u64)total.next[c].edges + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&w[2 + 2 * c], (u64)total.next[c].tab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (ti > 0) {
      T part = Op::identity();
      for (int i = tid; i < ti; i += kScanThreads) {
        const u64* w = words + (size_t)i * kAggStride;
        u64 x[W];
#pragma unroll
        for (int j = 0; j < W; ++j)
          while ((x[j] = __hip_atomic_load(&w[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(2);
        T t;
        t.rank = (int64_t)(x[0] - 1);
        if constexpr (NC > 0) {
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            t.next[c].edges = (int64_t)(x[1 + 2 * c] - 1);
            t.next[c].tab = (RngTab)x[2 + 2 * c];
          }
        }
        part = op(part, t);
      }
      (void)block_exclusive<T, Op>(part, lds, op, &before);
    }
  }
  const int64_t size0 = pt.h.seeds ? 0 : *pt.h.size_in;
  const int64_t id0 = pt.h.seeds ? 0 : size0 - *pt.h.dup;
  run = op(before, run);
  PYG_FOLD_STAMP(6);
  if (pt.last && lt == nblocks - 1 && tid == 0) {
    const T grand = op(before, total);
    if (pt.h.seeds) {
      *pt.h.size_out = pt.n_fixed;
      *pt.h.dup = pt.n_fixed - grand.rank;
    } else {
      *pt.h.size_out = size0 + grand.rank;
    }
    if constexpr (NC > 0) {
#pragma unroll
      for (int c = 0; c < NC; ++c) L.tb.tot[cons.c[c].tot_index] = grand.next[c];
    }
  }
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t p = base + k;
    if (p < n) {
      const bool flag = v[k].rank != 0;
      if (flag) {
        pt.h.vals[sl[k]] = pt.h.tag | (u64)(id0 + run.rank);
        if (!pt.h.seeds) {
          pt.h.nodes[size0 + run.rank] = nd[k];
          if (pt.h.batch) pt.h.batch[size0 + run.rank] = bt[k];
        }
      }
      if constexpr (NC > 0) {
        if (flag || pt.h.seeds) {
          const int64_t i = pt.h.seeds ? p : run.rank;  // index in the next hop's frontier
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            cons.c[c].edge_off[i] = run.next[c].edges;
            cons.c[c].tabp[i] = run.next[c].tab;
          }
        }
      }
    }
    run = op(run, v[k]);
  }
  PYG_FOLD_STAMP(7);
#ifdef PYG_HIP_FOLD_TIMING
  if (FOLD && threadIdx.x == 0) dbg_kstamp(999);
#endif
}

// The closing role: engine position of the call folded over all (hop, relation) totals, the write-once tables and the engine
// hand-back copied into pinned host memory (kernels write there directly: no copy launch behind the last hop) ...
__device__ __forceinline__ void fused_fold_tables(const FTables& tb, ChainState* chain, const FFoldRec& fold, const u64* words) {
  __shared__ int64_t fold_word;
  if (threadIdx.x == 0) {
    int64_t w = tb.word0;
    int u = tb.units0;
    bool ab = false;
    for (int q = 0; q < tb.L * tb.R; ++q) {
      const CountAgg t = tb.tot[q];
      if (t.edges > 0) {
        const int u0 = u;
        w += tab_dw(t.tab, u0);
        u = tab_nb(t.tab, u0);
      }
      ab = ab || tb.overflow[q] != 0;
    }
    chain->word = w;
    chain->units = u;
    chain->abort = ab ? 1 : 0;
    fold_word = w;
  }
  __syncthreads();
  if (fold.tables_host) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(fold.tables_dev);
    uint32_t* dst = reinterpret_cast<uint32_t*>(fold.tables_host);
    for (int i = threadIdx.x; i < fold.tables_bytes / 4; i += 256) dst[i] = src[i];
  }
  if (fold.hb) {  // mt_finish_chain_kernel (sampler_rng.hip), with the position just folded
    MtHandBack* hb = fold.hb;
    const uint32_t* out32 = reinterpret_cast<const uint32_t*>(words);
    const int64_t a0 = fold.a0;
    const int64_t n32 = (fold_word / 128 + 1) * 256;
    int status = 0;
    const int64_t mp = n32 - a0;
    const int64_t k = (mp + 623) / 624;
    if (n32 <= a0) status = 2;
    else if (a0 + 624 * k > fold.generated32) status = 1;
    if (status == 0) {
      const int64_t o0 = a0 + 624 * (k - 1);
      for (int i = threadIdx.x; i < 624; i += 256) hb->st.state[i] = mt_untemper(mt_output_at(out32, o0 + i));
      if (threadIdx.x == 0) {
        const int64_t nx = mp - 624 * (k - 1);
        hb->st.next = (uint32_t)nx;
        hb->st.left = (int32_t)(625 - nx);
      }
    }
    if (threadIdx.x == 0) {
      hb->n32 = n32;
      hb->status = status;
    }
  }
}
// ... and, when everything the host reads is written (system scope), the word it waits for
__device__ __forceinline__ void fused_fold_done(const FFoldRec& fold) {
  if (fold.done) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(fold.done, fold.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- the LAST hop: two passes, no table writes, no finalize ---------------------------------------------------------------
// The nodes the last hop discovers are never expanded: nobody looks them up in the table afterwards.  So the hop's
// bookkeeping needs neither the apply pass's ~E random table stores nor the finalize role's ~E random table gathers:
//   pass 1 (fused_last_reduce): one table gather per emission, as in fused_reduce; what it finds goes into the emission's
//     cache word -- 1 = first occurrence (the table holds this very position), (id << 2) | 2 = a node of an earlier hop (the
//     table holds its final id), (q << 2) = a node first seen at position q of THIS hop -- and, per tile, into a bitmap of
//     the first occurrences with their counts in front of every 64-position word (LastTile) and the tile's count;
//   pass 2 (fused_last_apply): every block scans the segment's tile counts (<= 8192 values, L2-resident), then knows the
//     rank of ANY position of the hop from two small reads -- rank(q) = tiles in front + LastTile::before[word] +
//     popcount(bits below q) -- and with it the local id of every emission: first occurrences append their node and
//     write `col`, nodes of earlier hops write `col`, in-hop duplicates compute their owner's rank.  ~200 bytes per tile
//     (C3: 140 KB) are what the duplicates read at random, instead of the 19.6 MB table.
// Tiles are STRIPED (thread t holds positions t, t + 256, ...): a wave's ballot over item i IS bitmap word 4 i + wave.
// C3: reduce 13.6 + apply 15.5 + finalize 14.3 us -> see profiles/NOTES_r6.md section 2f.
__global__ __launch_bounds__(256) void fused_last_reduce_kernel(const FScanLaunch L) {
  __shared__ uint32_t s_cnt[16];
  const int bx = (int)blockIdx.x;
  int k = 0;
#pragma unroll
  for (int j = 0; j < kMaxParts + 1; ++j) k += (j < L.n - 1 && bx >= L.cum[j]) ? 1 : 0;
  const int lt = bx - (k > 0 ? L.cum[k - 1] : 0);
  const FPart& pt = L.part[k];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n = part_count(L, pt);
  const int64_t base = (int64_t)lt * kScanTile;
  const u64 provbit = pt.h.prov ^ pt.h.tag;
  u64 sl[kScanItems], vv[kScanItems];
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) sl[i] = 0, vv[i] = 0;
  if (base < n) {
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
      const int64_t p = base + i * kScanThreads + tid;
      sl[i] = pt.slots[p < n ? p : n - 1];
    }
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) vv[i] = pt.h.vals[sl[i]];
  }
  LastTile* meta = pt.h.last_meta + (pt.tile0 + lt);
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    const int64_t p = base + i * kScanThreads + tid;
    const bool first = p < n && vv[i] == pt.h.prov + (u64)(pt.pos_base + p);
    const u64 m = __ballot(first);
    if (p < n) {
      u64 word = 1ull;
      if (!first) word = (vv[i] & provbit) == 0 ? (((vv[i] & pt.h.idmask) << 2) | 2ull) : ((vv[i] - pt.h.prov) << 2);
      pt.cache[p] = word;
    }
    if (lane == 0) {
      meta->bits[i * 4 + wave] = m;
      s_cnt[i * 4 + wave] = (uint32_t)__popcll(m);
    }
  }
  __syncthreads();
  if (tid < 16) {   // counts in front of every word: word order = position order
    const uint32_t v = s_cnt[tid];
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
      if (tid >= d) incl += up;
    }
    meta->before[tid] = incl - v;
    if (tid == 15) pt.h.last_cnt[pt.tile0 + lt] = incl;
  }
}

constexpr int kLastMaxTiles = 8192;   // tiles of a segment (fused_eligible): their counts are scanned in LDS by every block

__global__ __launch_bounds__(256) void fused_last_apply_kernel(const FScanLaunch L) {
  __shared__ uint32_t s_tp[kLastMaxTiles];   // tiles in front of tile t: first occurrences
  __shared__ uint32_t s_red[8];
  __shared__ uint32_t s_cnt[16], s_pre[16];
  const int bx = (int)blockIdx.x;
  int k = 0;
#pragma unroll
  for (int j = 0; j < kMaxParts + 1; ++j) k += (j < L.n - 1 && bx >= L.cum[j]) ? 1 : 0;
  const int first_blk = k > 0 ? L.cum[k - 1] : 0;
  const int lt = bx - first_blk;
  const int nblocks = L.cum[k] - first_blk;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (L.nc[k] == -2) {
    // the closing role.  The copy of the tables it sends to the host may catch this launch's other blocks half-way
    // through writing the hop's node counts: it overwrites those cells of the HOST copy with what it computes itself --
    // a segment's count = the sum of the tile counts of the first pass; a type without a segment keeps its size.
    fused_fold_tables(L.tb, L.chain, L.closing, L.words);
    __syncthreads();
    if (L.closing.tables_host) {
      int64_t* host_size = reinterpret_cast<int64_t*>(L.closing.tables_host);   // size_at is the first array of the tables
      for (int j = 0; j < L.n; ++j) {
        if (L.nc[j] < 0 || !L.part[j].last) continue;
        const FSegHdr& h = L.part[j].h;
        uint32_t sum = 0;
        for (int i = tid; i < h.ntiles; i += kScanThreads) sum += h.last_cnt[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += (uint32_t)__shfl_xor((int)sum, d);
        if (lane == 0) s_red[wave] = sum;
        __syncthreads();
        if (tid == 0) host_size[h.size_out - L.tb.size_at] = *h.size_in + (int64_t)(s_red[0] + s_red[1] + s_red[2] + s_red[3]);
        __syncthreads();
      }
      if (tid < L.tb.T) {
        const bool has = tid < 32 ? ((L.type_mask_lo >> tid) & 1u) != 0 : ((L.type_mask_hi >> (tid - 32)) & 1u) != 0;
        if (!has) host_size[(L.ell + 1) * L.tb.T + tid] = L.tb.size_at[L.ell * L.tb.T + tid];
      }
    }
    fused_fold_done(L.closing);
    return;
  }
  if (L.nc[k] < 0) {  // carry: node types nobody appended to in this hop keep their size
    if (tid < L.tb.T) {
      const bool has = tid < 32 ? ((L.type_mask_lo >> tid) & 1u) != 0 : ((L.type_mask_hi >> (tid - 32)) & 1u) != 0;
      if (!has) L.tb.size_at[(L.ell + 1) * L.tb.T + tid] = L.tb.size_at[L.ell * L.tb.T + tid];
    }
    return;
  }
  const FPart& pt = L.part[k];
  const int nt = pt.h.ntiles;
  // exclusive prefix of the segment's tile counts, by every block for itself: thread t scans a contiguous run
  uint32_t grand;
  {
    const int per = (nt + kScanThreads - 1) / kScanThreads;
    const int i0 = tid * per, i1 = min(i0 + per, nt);
    uint32_t sum = 0;
    for (int i = i0; i < i1; ++i) sum += pt.h.last_cnt[i];
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
      if (lane >= d) incl += up;
    }
    if (lane == 63) s_red[wave] = incl;
    __syncthreads();
    uint32_t before_w = 0;
    for (int w = 0; w < wave; ++w) before_w += s_red[w];
    grand = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    uint32_t run = before_w + incl - sum;
    for (int i = i0; i < i1; ++i) {
      s_tp[i] = run;
      run += pt.h.last_cnt[i];
    }
    __syncthreads();
  }
  const int64_t n = part_count(L, pt);
  const int64_t base = (int64_t)lt * kScanTile;
  const int64_t size0 = *pt.h.size_in;
  const int64_t id0 = size0 - *pt.h.dup;
  if (pt.last && lt == nblocks - 1 && tid == 0) *pt.h.size_out = size0 + grand;
  if (base >= n) return;
  const int ti = pt.tile0 + lt;
  int64_t* __restrict__ out_col = L.out_col[k] + part_col_offset(L, pt);
  u64 word[kScanItems];
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    const int64_t p = base + i * kScanThreads + tid;
    word[i] = p < n ? pt.cache[p] : 2ull;
  }
  int rk[kScanItems];
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    const u64 m = __ballot(word[i] == 1ull);
    rk[i] = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (lane == 0) s_cnt[i * 4 + wave] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  if (tid < 16) {
    const uint32_t v = s_cnt[tid];
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
      if (tid >= d) incl += up;
    }
    s_pre[tid] = incl - v;
  }
  __syncthreads();
  const bool slot_is_node = reinterpret_cast<const void*>(pt.slots) == reinterpret_cast<const void*>(pt.e_node);
  const uint32_t before = s_tp[ti];
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    const int64_t p = base + i * kScanThreads + tid;
    if (p >= n) continue;
    const u64 w = word[i];
    if (w == 1ull) {
      const int64_t r = (int64_t)before + s_pre[i * 4 + wave] + rk[i];
      pt.h.nodes[size0 + r] = slot_is_node ? (int64_t)pt.slots[p] : pt.e_node[p];
      if (pt.h.batch) pt.h.batch[size0 + r] = pt.e_batch[p];
      out_col[p] = id0 + r;
    } else if (w & 2ull) {
      out_col[p] = (int64_t)(w >> 2);
    } else {
      // first seen at position q of this hop's segment: the part that holds q, its tile, the word and the bit
      const int64_t q = (int64_t)(w >> 2);
      int64_t qb = -1;
      int qt0 = 0;
#pragma unroll
      for (int j = 0; j < kMaxParts; ++j) {   // (the part of this segment with the largest pos_base <= q; bases differ by the parts' bounds)
        if (j < L.n && L.nc[j] >= 0 && L.part[j].h.vals == pt.h.vals) {
          const int64_t pb = L.part[j].pos_base;
          if (pb <= q && pb > qb) {
            qb = pb;
            qt0 = L.part[j].tile0;
          }
        }
      }
      const int64_t ql = q - qb;
      const int t_o = qt0 + (int)(ql / kScanTile);
      const int o = (int)(ql % kScanTile);
      const LastTile* mt = pt.h.last_meta + t_o;
      const u64 bits = mt->bits[o >> 6];
      const int64_t r = (int64_t)s_tp[t_o] + mt->before[o >> 6] + __popcll(bits & ((1ull << (o & 63)) - 1ull));
      out_col[p] = id0 + r;
    }
  }
}

// ---- sample: relation e of hop l; everything position-dependent is resolved from the tables ------------------------
// Per block: ONE round of loads (lane k of the first wave reads total k and overflow flag k; L * R <= 64), then the fold
// over the relations in front -- finalize_kernel's chain advance and hop_overflow()'s sticky abort -- runs on shuffles.
struct FResolved {
  int64_t frontier;  // -1: this relation (or one before it) lacks random words
  int64_t begin;
  int64_t word;
  int units;
  int pad;
  int64_t rel_off;
};

__device__ __forceinline__ FResolved fused_resolve(const FTables& tb, int ell, int e, int t_src, int64_t avail_blocks,
                                                  bool publish) {
  const int lane = threadIdx.x & 63;
  const int n = tb.L * tb.R;
  CountAgg mine = CountOp::identity();
  int over_in = 0;
  if (lane < n) {
    mine = tb.tot[lane];
    over_in = lane < ell * tb.R ? tb.overflow[lane] : 0;
  }
  const int64_t begin = ell > 0 ? tb.size_at[(ell - 1) * tb.T + t_src] : 0;
  const int64_t end = tb.size_at[ell * tb.T + t_src];
  bool aborted = __ballot(over_in != 0) != 0 || *tb.wide != 0;
  bool own = false;
  int64_t w = tb.word0;
  int u = tb.units0;
  int64_t rel_off = 0;
  const int last = ell * tb.R + e;
  for (int k = 0; k <= last; ++k) {
    CountAgg t;
    t.edges = __shfl(mine.edges, k);
    t.tab = (RngTab)__shfl((unsigned long long)mine.tab, k);
    if (k < ell * tb.R && (k % tb.R) == e) rel_off += t.edges;
    if (t.edges > 0) {
      if (k >= ell * tb.R) {  // hop_overflow(): a relation of this hop that lacks words stops everything behind it
        const int64_t end_word = w + tab_dw(t.tab, u);
        if (end_word / 128 + 1 > avail_blocks) {
          if (k < last) aborted = true;
          else own = true;
        }
      }
      if (k < last) {
        const int u0 = u;
        w += tab_dw(t.tab, u0);
        u = tab_nb(t.tab, u0);
      }
    }
  }
  const bool over = aborted || own;
  if (publish && threadIdx.x == 0) tb.overflow[last] = over ? (aborted ? 2 : 1) : 0;
  FResolved r;  // every lane computed the same values: say so
  r.frontier = over ? -1 : end - begin;
  r.begin = begin;
  r.word = w;
  r.units = u;
  r.pad = 0;
  r.rel_off = rel_off;
  return uniform_record(&r);
}

template <int G, bool I64>
__device__ __forceinline__ void fused_sample(const FSampleLaunch& L, const FSampleRec& rec, int blk, int64_t avail_blocks,
                                             const u64* words) {
  const FResolved r = fused_resolve(L.tb, rec.ell, rec.e, rec.t_src, avail_blocks, blk == 0);
  if (r.frontier <= 0) return;
  HopArgs a;
  a.avail_blocks = avail_blocks;
  a.nodes = rec.nodes;
  a.batch = rec.batch;
  a.begin = r.begin;
  a.frontier = r.frontier;
  a.range = rec.range;
  a.col = rec.range.col;
  if constexpr (I64) {  // (see FCons::load)
    a.range.rowptr.is32 = 0;
    a.range.col.is32 = 0;
    a.col.is32 = 0;
  }
  a.count = rec.count;
  a.replace = rec.replace;
  a.num_batches = rec.num_batches;
  a.edge_off = rec.edge_off;
  a.rng_word = nullptr;
  a.rng_units = nullptr;
  a.words = words;
  a.e_row = rec.e_row + r.rel_off;
  a.e_node = rec.e_node;
  a.e_batch = rec.e_batch;
  a.e_eid = rec.e_eid ? rec.e_eid + r.rel_off : nullptr;
  a.e_slot = rec.e_slot;
  a.table = rec.table;
  a.pos_base = rec.pos_base;
  a.tab_prefix = rec.tabp;
  a.w0 = r.word;
  a.u0 = r.units;
  if constexpr (G == 0) sample_wave_body(a, ((int64_t)blk * blockDim.x + threadIdx.x) >> 6);  // fan-outs > 64: a wave per node
  else sample_group_body<G>(a, blk);
}

__device__ __forceinline__ void fused_finalize(const FSampleLaunch& L, const FFinalRec& ff, int blk) {
  const FTables& tb = L.tb;
  const int k = ff.ell * tb.R + ff.e;
  const int lane = threadIdx.x & 63;
  int64_t mine = 0;
  if (lane <= ff.ell) mine = tb.tot[lane * tb.R + ff.e].edges;  // hops 0 ... l of this relation
  const int over = tb.overflow[k];
  int64_t off = 0;
  for (int h = 0; h < ff.ell; ++h) off += __shfl(mine, h);
  const int64_t n = __shfl(mine, ff.ell);
  if (__builtin_amdgcn_readfirstlane(over)) return;
  const int64_t p = (int64_t)blk * blockDim.x + threadIdx.x;
  if (p < n) ff.out_col[off + p] = (int64_t)(ff.vals[ff.slots[p]] & ff.idmask);
}

// tables at the start of a call: sizes 0, totals identity, no overflow; tickets and tile aggregates of the one-pass scans 0
__global__ void fused_init_kernel(FTables tb, unsigned* __restrict__ sync, int sync_words) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  fused_tables_init(tb, i);
  if (i < sync_words) sync[i] = 0;
}

// launch kind 1: [finalize of the previous hop | sampling of this hop] (+ the engine fold behind the last hop)
// GMAX = the widest lane group of any sampling item of the launch (registers of the 64-lane variant only where needed)
// I64: the call's graphs are int64 (see FCons::load)
template <int GMAX, bool I64>
__global__ __launch_bounds__(256) void fused_sample_kernel(const FSampleLaunch L, int64_t avail_blocks,
                                                           const u64* __restrict__ words) {
  const int bx = (int)blockIdx.x;
#ifdef PYG_HIP_FOLD_TIMING
  if (bx == 0 && threadIdx.x == 0) dbg_kstamp(100 + GMAX);
#endif
  int k = 0;
#pragma unroll
  for (int j = 0; j < 2 * kMaxParts + 2; ++j) k += (j < L.n - 1 && bx >= L.cum[j]) ? 1 : 0;
  const int b = bx - (k > 0 ? L.cum[k - 1] : 0);
  const int role = L.role[k];
  const int idx = L.idx[k];
  switch (role) {
    case kRoleSample8: fused_sample<8, I64>(L, L.s[idx], b, avail_blocks, words); break;
    case kRoleSample16:
      if constexpr (GMAX >= 16) fused_sample<16, I64>(L, L.s[idx], b, avail_blocks, words);
      break;
    case kRoleSample32:
      if constexpr (GMAX >= 32) fused_sample<32, I64>(L, L.s[idx], b, avail_blocks, words);
      break;
    case kRoleSample64:
      if constexpr (GMAX >= 64) fused_sample<64, I64>(L, L.s[idx], b, avail_blocks, words);
      break;
    case kRoleSampleWave:
      if constexpr (GMAX >= 64) fused_sample<0, I64>(L, L.s[idx], b, avail_blocks, words);
      break;
    case kRoleFinalize: fused_finalize(L, L.f[idx], b); break;
    default:  // kRoleFold: the engine position, the tables and the engine hand-back, written where the host reads them
      fused_fold_tables(L.tb, L.chain, L.fold, words);
      fused_fold_done(L.fold);
      break;
  }
}

// launch kinds 2 / 3: the scans.  MAXNC = the most consumers of any segment of the launch (the registers of the widest
// aggregate are only paid where one occurs: the last hop carries none).
// MODE 0 = reduce, 1 = apply (the two-pass form: PYG_HIP_SAMPLER_ONEPASS=0), 2 = one pass, 3 = one pass of a single
// block with the call's initialisation and the seeds' insertion folded in
constexpr int kScanReduce = 0, kScanApply = 1, kScanOnePass = 2, kScanSeedFold = 3;
template <int MAXNC, int MODE, bool I64>
__global__ __launch_bounds__(256) void fused_scan_kernel(const FScanLaunch L) {
  int bx = (int)blockIdx.x;
#ifdef PYG_HIP_FOLD_TIMING
  if (bx == 0 && threadIdx.x == 0) dbg_kstamp(200 + 10 * MAXNC + MODE);
#endif
  if constexpr (MODE == kScanOnePass) {
    __shared__ int s_ticket;
    if (threadIdx.x == 0) s_ticket = (int)__hip_atomic_fetch_add(L.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    bx = __builtin_amdgcn_readfirstlane(s_ticket);
  }
  int k = 0;
#pragma unroll
  for (int j = 0; j < kMaxParts + 1; ++j) k += (j < L.n - 1 && bx >= L.cum[j]) ? 1 : 0;
  const int first = k > 0 ? L.cum[k - 1] : 0;
  const int b = bx - first;
  const int nblocks = L.cum[k] - first;
  const int nc = L.nc[k];
  if (nc < 0) {  // carry: node types nobody appended to in this hop keep their size
    const int t = (int)threadIdx.x;
    if (t < L.tb.T) {
      const bool has = t < 32 ? ((L.type_mask_lo >> t) & 1u) != 0 : ((L.type_mask_hi >> (t - 32)) & 1u) != 0;
      if (!has) L.tb.size_at[(L.ell + 1) * L.tb.T + t] = L.tb.size_at[L.ell * L.tb.T + t];
    }
    return;
  }
  const FPart& pt = L.part[k];
#define PYG_FUSED_CASE(N)                                                           \
  if (MAXNC >= N && nc == N) {                                                      \
    if constexpr (MAXNC >= N) {                                                     \
      if constexpr (MODE == kScanReduce) fused_reduce<N, I64>(L, pt, b);                 \
      else if constexpr (MODE == kScanApply) fused_apply<N, I64>(L, pt, b, nblocks);     \
      else fused_onepass<N, MODE == kScanSeedFold, I64>(L, pt, b, nblocks);              \
    }                                                                               \
    return;                                                                         \
  }
  PYG_FUSED_CASE(0)
  PYG_FUSED_CASE(1)
  PYG_FUSED_CASE(2)
  PYG_FUSED_CASE(3)
#undef PYG_FUSED_CASE
}
