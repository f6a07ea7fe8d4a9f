// neighbor_sample / hetero_neighbor_sample for gfx950 (MI355X), bit-exact with the reference's
// single-threaded CPU kernel (pyg_lib/csrc/sampler/cpu/neighbor_kernel.cpp).  The reference has
// no device sampler; its CPU kernel is the semantic specification restated in oracle/.
//
// How a strictly sequential algorithm is made parallel without changing a single output
// (DESIGN.md "neighbor_sample"):
//   * RNG.  The reference consumes one stream of 64-bit words 16/32/64 bits at a time
//     (rand_engine.h:41-76).  Which bits a draw uses depends only on how many bits every earlier
//     draw took, and that is known from degrees and fan-outs before any sampling happens.  Each
//     frontier node is summarised as a transition table over the five possible "bits left in the
//     current word" states {0,16,32,48,64}; an exclusive scan under table composition gives every
//     node its exact (word index, bit offset) start state.  The words themselves are drawn on the
//     host by the caller's generator (torch.manual_seed compatible) and shipped to the device.
//   * Sampling.  One wave per frontier node: full neighbourhoods are copied 64 edges at a time,
//     Floyd's without-replacement picks (neighbor_kernel.cpp:231-240) are evaluated in draw order
//     with the chosen set held across the wave's lanes.
//   * Relabelling.  Local ids are first-occurrence ranks in emission order (mapper.h:30-46).
//     Every emitted edge inserts its destination into an open-addressing hash table with
//     atomicMin(emission position); an edge owns its node iff it holds the minimum; an exclusive
//     scan of the owner flags yields the ranks, i.e. exactly the ids the sequential Mapper hands
//     out.  Duplicate seeds keep the reference's quirk (ids count distinct nodes, rows count
//     positions).
// Everything per hop is HBM/latency-bound integer work: 8-byte gathers of col[], 16-byte hash
// slots, coalesced row/col/edge-id streams.
#include "common.h"
#include "sampler_rng.h"
#include "scan.h"

#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <thread>
#include <atomic>
#include <functional>
#include <mutex>
#include <vector>

namespace pyg_hip {
namespace {

using namespace sampler;

// PYG_HIP_SAMPLER_TRACE=1 prints host wall time per phase of a sampler call (diagnostics only).
struct PhaseTimer {
  bool on;
  std::chrono::steady_clock::time_point last;
  double acc[8] = {0};
  PhaseTimer() : on(getenv("PYG_HIP_SAMPLER_TRACE") != nullptr), last(std::chrono::steady_clock::now()) {}
  void lap(int k) {
    if (!on) return;
    auto now = std::chrono::steady_clock::now();
    acc[k] += std::chrono::duration<double, std::micro>(now - last).count();
    last = now;
  }
  // finer trace: (label, microseconds since the call started)
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  std::vector<std::pair<const char*, double>> marks;
  void mark(const char* what) {
    if (!on) return;
    marks.emplace_back(what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  ~PhaseTimer() {
    if (on && !marks.empty()) {
      fprintf(stderr, "[pyg_hip sampler] trace us:");
      for (auto& m : marks) fprintf(stderr, " %s=%.0f", m.first, m.second);
      fprintf(stderr, "\n");
    }
    if (on)
      fprintf(stderr,
              "[pyg_hip sampler] us: seeds=%.0f count+scan+sync=%.0f rng=%.0f reserve=%.0f "
              "sample+dedup-launch=%.0f sync2=%.0f release=%.0f finish=%.0f\n",
              acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7]);
  }
};

constexpr u64 kEmpty = ~0ull;           // empty hash key / unset value (memset 0xFF)
constexpr u64 kProvisional = 1ull << 62;  // values >= this are emission positions, below: final ids

// ---- RNG state algebra ---------------------------------------------------------------------------
// State = number of unused 16-bit units in the current word (0..4).  A node's draws map a start state
// u to (words advanced dw[u], units left nb[u]).  Start states only differ until each has fetched its
// first fresh word (after which all hold 4 - n units), so dw[u] is D or D + 1 for one common D, and
// the whole table packs into ONE 64-bit value -- bits 4u..4u+2 = nb[u], bit 4u+3 = dw[u] - D,
// bits 20.. = D -- which keeps the scan's operator a handful of shifts (the scan kernels are
// latency-bound; the first version carried 5 x (int32, int32) tables and ran 3-5x longer).
typedef u64 RngTab;

// Almost every frontier node draws 16-bit numbers only (degree < 2^16), and c such draws in a row are described by
// the single number c (from `u` units left: nothing fetched while c <= u, else ceil((c - u) / 4) fresh words).  Such
// tables are kept as  kPureTab | c  and compose by ADDITION -- the count scans are latency-bound and their operator
// sits inside 6 shuffle steps per wave scan; the packed 5-state form above is only built when a draw wider than 16
// bits appears (a row of degree >= 2^16).
constexpr u64 kPureTab = 1ull << 63;

__host__ __device__ inline bool tab_is_pure(RngTab t) { return (t & kPureTab) != 0; }
__host__ __device__ inline RngTab tab_pure(int64_t c) { return kPureTab | (u64)c; }
__host__ __device__ inline RngTab rng_identity() { return kPureTab; }
// the identity in the packed 5-state form: the biased-sampling scans carry a plain counter in its word field
// (identity transitions compose additively there, and tab_dw() returns that counter)
__host__ __device__ inline RngTab rng_identity_packed() { return 0x43210ull; }
__host__ __device__ inline int64_t tab_dw(RngTab t, int u) {
  if (tab_is_pure(t)) {
    const int64_t c = (int64_t)(t & ~kPureTab);
    return c <= u ? 0 : (c - u + 3) / 4;
  }
  return (int64_t)(t >> 20) + (int64_t)((t >> (4 * u + 3)) & 1);
}
__host__ __device__ inline int tab_nb(RngTab t, int u) {
  if (tab_is_pure(t)) {
    const int64_t c = (int64_t)(t & ~kPureTab);
    if (c <= u) return u - (int)c;
    const int64_t c2 = c - u;
    return (int)(4 * ((c2 + 3) / 4) - c2);
  }
  return (int)((t >> (4 * u)) & 7);
}

__host__ __device__ inline RngTab tab_pack(const int64_t (&dw)[5], const int (&nb)[5]) {
  int64_t d = dw[0];
#pragma unroll
  for (int u = 1; u < 5; ++u) d = dw[u] < d ? dw[u] : d;
  RngTab t = (u64)d << 20;
#pragma unroll
  for (int u = 0; u < 5; ++u) t |= (u64)((nb[u] & 7) | ((int)(dw[u] - d) << 3)) << (4 * u);
  return t;
}

// packed 5-state form of any table
__host__ __device__ inline RngTab tab_general(RngTab t) {
  if (!tab_is_pure(t)) return t;
  int64_t dw[5];
  int nb[5];
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    dw[u] = tab_dw(t, u);
    nb[u] = tab_nb(t, u);
  }
  return tab_pack(dw, nb);
}

__host__ __device__ inline int need_units(u64 range) {
  // rand_engine.h:44-50: 16 bits below 2^16, 32 below 2^32, else 64
  return range < (1ull << 16) ? 1 : (range < (1ull << 32) ? 2 : 4);
}

// one more draw of n units appended to the table (only rows of degree >= 2^16 take the general path)
__host__ __device__ inline void rng_push_draw(RngTab& t, int n) {
  if (n == 1 && tab_is_pure(t)) {
    t += 1;
    return;
  }
  t = tab_general(t);
  int64_t dw[5];
  int nb[5];
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    dw[u] = tab_dw(t, u);
    nb[u] = tab_nb(t, u);
    if (nb[u] < n) {
      dw[u] += 1;
      nb[u] = 4 - n;
    } else {
      nb[u] -= n;
    }
  }
  t = tab_pack(dw, nb);
}

// g after f, at least one of them in the packed 5-state form (a draw wider than 16 bits somewhere: rare).  Kept out of
// line: the scans apply their operator ~20 times per thread, and the packed path inlined into each of them is what
// made the fused apply kernels need 244 registers.
__host__ __device__ __attribute__((noinline)) RngTab rng_compose_general(RngTab f, RngTab g);

// g after f
__host__ __device__ inline RngTab rng_compose(RngTab f, RngTab g) {
  if (tab_is_pure(f) && tab_is_pure(g)) return f + (g & ~kPureTab);
  return rng_compose_general(f, g);
}

__host__ __device__ __attribute__((noinline)) RngTab rng_compose_general(RngTab f, RngTab g) {
  f = tab_general(f);
  g = tab_general(g);
  u64 e[5];
  u64 mn = 3;
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const u64 ef = (f >> (4 * u)) & 0xF;
    const u64 eg = (g >> (4 * (ef & 7))) & 0xF;
    const u64 b = (ef >> 3) + (eg >> 3);
    e[u] = (eg & 7) | (b << 3);  // b in 0..2, temporarily 2 bits wide
    mn = b < mn ? b : mn;
  }
  RngTab h = ((f >> 20) + (g >> 20) + mn) << 20;
#pragma unroll
  for (int u = 0; u < 5; ++u) h |= (e[u] - (mn << 3)) << (4 * u);
  return h;
}

struct CountAgg {
  int64_t edges;
  RngTab tab;
};

struct CountOp {
  __host__ __device__ CountAgg operator()(const CountAgg& a, const CountAgg& b) const {
    CountAgg r;
    r.edges = a.edges + b.edges;
    r.tab = rng_compose(a.tab, b.tab);
    return r;
  }
  __host__ __device__ static CountAgg identity() {
    CountAgg r;
    r.edges = 0;
    r.tab = rng_identity();
    return r;
  }
};

// ---- hash table ------------------------------------------------------------------------------------
struct HashTable {
  u64* keys;  // slot s: keys[2 s], vals[2 s] with vals == keys + 1 (interleaved pairs)
  u64* vals;
  u64 mask;  // capacity - 1
  // Direct-address mode (the reference's Mapper does the same when the node count allows it, mapper.h:17-24):
  // vals[node] for node < the type's node count, no keys, no probing -- one atomic per insert instead of a key
  // load + CAS + atomic, and a table a third smaller than the hash table of a products-sized batch.
  int dense;
  // Value coding.  Classic tables (cleared per call): an emission position p is stored as kProvisional + p, a final
  // local id as the id itself (prov = kProvisional, tag = 0, idmask = ~0).  Epoch tables (direct-address tables the
  // library keeps between calls, see DenseCache): [63] 0 | [62:43] kEpochMax - epoch | [42] provisional | [41:0]
  // position or id -- a newer call's values are smaller than anything an older call left behind, so atomicMin treats
  // stale entries exactly like kEmpty and the table needs no clearing (prov = tag | 1 << 42, idmask = 2^42 - 1).
  u64 prov = 1ull << 62;  // kProvisional
  u64 tag = 0;
  u64 idmask = ~0ull;
};
static_assert(kProvisional == 1ull << 62, "HashTable::prov default");
constexpr int kEpochShift = 43;
constexpr u64 kEpochMax = (1ull << 20) - 1;
constexpr u64 kEpochProv = 1ull << 42;

__device__ __forceinline__ u64 hash64(u64 x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

// Find-or-claim the slot of `key`.  Key and value of a slot are interleaved (16 bytes: one memory sector
// per probe + value update instead of two); the returned handle is 2 * slot, to be used as
// t.keys[handle] / t.vals[handle] (vals = keys + 1).
__device__ __forceinline__ u64 table_slot(const HashTable& t, u64 key) {
  if (t.dense) return key;
  u64 s = hash64(key) & t.mask;
  while (true) {
    u64 k = __hip_atomic_load(&t.keys[2 * s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == key) return 2 * s;
    if (k == kEmpty) {
      u64 expected = kEmpty;
      if (__hip_atomic_compare_exchange_strong(&t.keys[2 * s], &expected, key, __ATOMIC_RELAXED,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        return 2 * s;
      if (expected == key) return 2 * s;
    }
    s = (s + 1) & t.mask;
  }
}

__device__ __forceinline__ u64 make_key(int64_t node, int64_t batch, int64_t num_batches) {
  // non-disjoint: num_batches == 1, batch == 0
  return (u64)node * (u64)num_batches + (u64)batch;
}

__global__ void rehash_kernel(HashTable oldt, HashTable newt) {
  const u64 n = oldt.mask + 1;
  for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    const u64 k = oldt.keys[2 * i];
    if (k == kEmpty) continue;
    const u64 s = table_slot(newt, k);
    newt.vals[s] = oldt.vals[2 * i];
  }
}

struct TypeState {
  int64_t size;      // length of the node list of this type
  int64_t distinct;  // Mapper::curr
  int64_t slice_b;   // frontier of the current hop = nodes [slice_b, slice_e) (fully queued mode)
  int64_t slice_e;
};

// ---- seeds -----------------------------------------------------------------------------------------
// mapper.fill(seed) / per-seed insert (neighbor_kernel.cpp:409-416, 667-683): emission position =
// seed index within its type; value = min position.
__global__ void seed_insert_kernel(const int64_t* __restrict__ seed, int64_t n, int64_t batch0,
                                   int disjoint, int64_t num_batches, HashTable t,
                                   int64_t* __restrict__ nodes, int64_t* __restrict__ batch,
                                   u64* __restrict__ slots, TypeState* __restrict__ ts) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (i == 0) {  // the seeds are the node list so far and the frontier of hop 0
    ts->size = n;
    ts->slice_e = n;
  }
  const int64_t v = seed[i];
  const int64_t b = disjoint ? batch0 + i : 0;
  nodes[i] = v;
  if (disjoint) batch[i] = b;
  const u64 s = table_slot(t, make_key(v, b, num_batches));
  slots[i] = s;
  __hip_atomic_fetch_min(&t.vals[s], t.prov + (u64)i, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
}

// seed_times[batch] = seed_time[i], or node_time[seed[i]] (neighbor_kernel.cpp:417-428, 684-699)
__global__ void seed_time_kernel(const int64_t* __restrict__ seed, const int64_t* __restrict__ seed_time,
                                 const int64_t* __restrict__ node_time, int64_t n, int64_t batch0,
                                 int64_t* __restrict__ seed_times) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) seed_times[batch0 + i] = seed_time ? seed_time[i] : node_time[seed[i]];
}

// ---- per-hop kernels -------------------------------------------------------------------------------
// Effective neighbourhood [rs, re) of a frontier node: the CSR row, narrowed by the temporal
// constraints of node_temporal_sample / edge_temporal_sample (neighbor_kernel.cpp:74-144):
// upper bound on time <= seed_time (neighbourhoods are time-sorted), and for the "last" strategy
// the `count` most recent ones.
// CSR index array of either width (the reference dispatches on the index dtype, neighbor_kernel.cpp:893,930; an
// int32 graph is read in place -- no widened copy of rowptr / col per call).  The width test is wave-uniform.
struct IdxArr {
  const void* p = nullptr;
  int is32 = 0;
  IdxArr() = default;
  __host__ __device__ IdxArr(const int64_t* q) : p(q), is32(0) {}
  __host__ __device__ IdxArr(const void* q, int narrow) : p(q), is32(narrow) {}
  __device__ __forceinline__ int64_t operator[](int64_t i) const {
    return is32 ? (int64_t) static_cast<const int32_t*>(p)[i] : static_cast<const int64_t*>(p)[i];
  }
  __host__ __device__ explicit operator bool() const { return p != nullptr; }
};

struct RangeCtx {
  IdxArr rowptr;
  IdxArr col;
  const int64_t* time;        // nullptr: no temporal constraint
  int edge_level;             // time indexed by edge (1) or by destination node (0)
  int last;                   // temporal_strategy == "last"
  const int64_t* seed_times;  // per batch id
  const int64_t* batch;       // batch id of every node of the src list
  int* error;                 // set to 1 on a non time-sorted neighbourhood
  __device__ void operator()(int64_t v, int64_t src_pos, int64_t count, int64_t* rs_out, int64_t* re_out) const {
    eval(v, time ? batch[src_pos] : 0, count, rs_out, re_out);
  }
  // the same for a node whose batch id is known directly (a node that is being appended: sampler_fused.h)
  __device__ void eval(int64_t v, int64_t batch_id, int64_t count, int64_t* rs_out, int64_t* re_out) const {
    narrow(rowptr[v], rowptr[v + 1], batch_id, count, rs_out, re_out);
  }
  // the temporal narrowing of a row whose bounds [rs, re) the caller has already fetched (sampler_fused.h issues the
  // row-bound loads of all items and consumers back to back before it looks at any of them)
  __device__ void narrow(int64_t rs, int64_t re, int64_t batch_id, int64_t count, int64_t* rs_out, int64_t* re_out) const {
    if (time && re > rs && count != 0) {
      const int64_t st = seed_times[batch_id];
      int64_t lo = rs, hi = re;  // first p in [rs, re) with st < time(p)
      while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        const int64_t tm = edge_level ? time[mid] : time[col[mid]];
        if (st < tm) hi = mid; else lo = mid + 1;
      }
      re = lo;
      if (last && count >= 0 && re - count > rs) rs = re - count;
      if (re - rs > 1) {
        const int64_t t0 = edge_level ? time[rs] : time[col[rs]];
        const int64_t t1 = edge_level ? time[re - 1] : time[col[re - 1]];
        if (!(t0 <= t1)) *error = 1;  // "Found invalid non-sorted temporal neighborhood"
      }
    }
    *rs_out = rs;
    *re_out = re;
  }
};

struct CountLoad {
  const int64_t* nodes;   // src node list
  int64_t begin;          // frontier begin
  RangeCtx range;
  int64_t count;          // fan-out (may be negative)
  int replace;
  const TypeState* fs;    // device-resident frontier [slice_b, slice_e) of the src type, or nullptr
  __device__ CountAgg operator()(int64_t i) const {
    CountAgg r;
    r.tab = rng_identity();
    int64_t b = begin;
    if (fs) {  // launched for an upper bound of the frontier size
      b = fs->slice_b;
      if (i >= fs->slice_e - b) {
        r.edges = 0;
        return r;
      }
    }
    const int64_t v = nodes[b + i];
    int64_t rs, re;
    range(v, b + i, count, &rs, &re);
    const int64_t deg = re - rs;
    if (deg <= 0 || count == 0) {
      r.edges = 0;
      return r;
    }
    if (count < 0 || (!replace && count >= deg)) {
      r.edges = deg;  // full neighbourhood, no draws (neighbor_kernel.cpp:188-193)
      return r;
    }
    r.edges = count;
    if ((u64)deg < (1ull << 16)) {
      r.tab = tab_pure(count);  // all draws take 16 bits
    } else if (replace) {
      const int n = need_units((u64)deg);
      for (int64_t j = 0; j < count; ++j) rng_push_draw(r.tab, n);
    } else {
      for (int64_t j = deg - count; j < deg; ++j) rng_push_draw(r.tab, need_units((u64)(j + 1)));
    }
    return r;
  }
};

struct CountStore {
  int64_t* edge_off;
  int64_t* rng_word;
  int32_t* rng_units;
  int64_t word0;   // hop start state (chain == nullptr)
  int32_t units0;
  const ChainState* chain;  // device-resident start state, or nullptr
  __device__ void operator()(int64_t i, const CountAgg& prefix, const CountAgg&) const {
    const int64_t w0 = chain ? chain->word : word0;
    const int u0 = chain ? chain->units : units0;
    edge_off[i] = prefix.edges;
    rng_word[i] = w0 + tab_dw(prefix.tab, u0);
    rng_units[i] = tab_nb(prefix.tab, u0);
  }
};


// Where the engine's 128-word blocks lie in the generated stream.  Normally block b = outputs [256 b, 256 b + 256);
// when weighted (biased) relations of the same call have drawn `shift` outputs straight from the generator
// before a block was fetched, that block lies `shift` outputs further on: blocks up to b0 (fetched before this
// relation started) at shift_old, later ones at shift_new.
struct WordSrc {
  const u64* words;
  int64_t b0;
  int64_t shift_old, shift_new;
  __device__ u64 at(int64_t word) const {
    const int64_t b = word >> 7;
    const int j = 127 - (int)(word & 127);
    const int64_t shift = b <= b0 ? shift_old : shift_new;
    u64 w;
    if (shift == 0) {
      w = words[b * 128 + j];
    } else {
      const uint32_t* o32 = reinterpret_cast<const uint32_t*>(words);
      const int64_t o = 256 * b + shift + 2 * j;
      w = ((u64)(mt_output_at(o32, o) ^ 0x80000000u) << 32) | mt_output_at(o32, o + 1);
    }
    // uniform_int_from_to's `x % (2^64 - 1)` differs from x only for x == 2^64 - 1 (-> 0); the device
    // generator stores x + INT64_MIN un-reduced (so the stream stays invertible, mt_finish_kernel)
    if (w == 0x7fffffffffffffffull) w = 0x8000000000000000ull;
    return w;
  }
};

// Device-resident results of one hop of one relation; published to pinned host memory by the hop's
// last kernel so that the host needs ONE synchronisation per hop.
struct HopInfo {
  CountAgg tot;      // count-scan total: emitted edges + RNG transition table of the whole frontier
  int64_t uniq;      // nodes seen for the first time
  int32_t overflow;  // the hop needs random words beyond the generated ones: nothing was sampled
  int32_t pad;
};

struct HopArgs {
  // run-ahead mode (info != nullptr): the kernel is launched before the host knows the count-scan
  // total; it checks on its own that every random word it may read has been generated
  HopInfo* info = nullptr;
  ChainState* chain = nullptr;  // device-resident engine position (start of this relation), sticky abort
  const TypeState* fs = nullptr;      // device-resident frontier of the src type (overrides begin / frontier)
  const int64_t* rel_size = nullptr;  // device-resident number of edges this relation has emitted so far
                                      // (offset of e_row / e_eid), or nullptr
  int64_t word0 = 0;          // engine position at the start of the hop (chain == nullptr)
  int units0 = 4;
  int64_t avail_blocks = 0;   // 128-word blocks readable by this launch
  const int64_t* nodes;       // src node list (positions are the emitted `row`)
  const int64_t* batch;       // src batch ids (disjoint) or nullptr
  int64_t begin;              // frontier begin (position of frontier node 0)
  int64_t frontier;           // frontier size
  RangeCtx range;
  IdxArr col;
  int64_t count;
  int replace;
  int64_t num_batches;
  const int64_t* edge_off;
  const int64_t* rng_word;
  const int32_t* rng_units;
  const u64* words;           // all prefetched RNG words, block-major
  int64_t word_b0 = INT64_MAX;  // WordSrc: blocks <= word_b0 lie at shift_old, later ones at shift_new
  int64_t shift_old = 0, shift_new = 0;
  __device__ WordSrc word_src() const { return WordSrc{words, word_b0, shift_old, shift_new}; }
  // emission buffers of this hop
  int64_t* e_row;
  int64_t* e_node;            // global dst node id
  int64_t* e_batch;           // (disjoint) or nullptr
  int64_t* e_eid;             // or nullptr
  u64* e_slot;
  HashTable table;
  // fused chain (sampler_fused.h): emission positions of relations that share a table are ordered by pos_base;
  // a node's engine position = (w0, u0) of the relation advanced by its prefix table
  int64_t pos_base = 0;
  const RngTab* tab_prefix = nullptr;
  int64_t w0 = 0;
  int u0 = 4;
};

struct RngCursor {
  int64_t word;
  int units;
  WordSrc words;
  // rand_engine.h:41-76 with the buffer laid out block-major: linear word index W lives at
  // words[(W / 128) * 128 + (127 - W % 128)] (the reference consumes each 128-word block from
  // its tail).
  __device__ u64 next(u64 range) {
    const int n = need_units(range);
    if (units < n) {
      ++word;
      units = 4;
    }
    const u64 w = words.at(word);
    const int shift = (4 - units) * 16;
    u64 v = w >> shift;
    if (n == 1) v &= 0xffffull;
    else if (n == 2) v &= 0xffffffffull;
    units -= n;
    return v % range;
  }
};

__device__ __forceinline__ void emit(const HopArgs& a, int64_t pos, int64_t edge, int64_t src_pos,
                                     int64_t src_batch) {
  const int64_t w = a.col[edge];
  a.e_row[pos] = src_pos;
  a.e_node[pos] = w;
  if (a.e_batch) a.e_batch[pos] = src_batch;
  if (a.e_eid) a.e_eid[pos] = edge;
  if (!a.table.keys) return;  // dist_neighbor_sample: no relabelling (neighbor_kernel.cpp:296-303)
  const u64 s = table_slot(a.table, make_key(w, src_batch, a.num_batches));
  a.e_slot[pos] = s;
#ifndef PYG_HIP_EXPERIMENT_NO_TABLE_ATOMIC  // (timing ablation of an experiment build: results are wrong without it)
  __hip_atomic_fetch_min(&a.table.vals[s], a.table.prov + (u64)(a.pos_base + pos), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// Run-ahead guard: true (for every thread of the launch alike) when the hop would read random words
// that do not exist yet; the host then generates them and repeats the hop.
__device__ __forceinline__ bool hop_overflow(const HopArgs& a) {
  if (!a.info) return false;
  const int64_t w0 = a.chain ? a.chain->word : a.word0;
  const int u0 = a.chain ? a.chain->units : a.units0;
  const int64_t end_word = w0 + tab_dw(a.info->tot.tab, u0);
  // an earlier relation of the hop already gave up: its engine position is not final, so neither is ours
  const bool aborted = a.chain && a.chain->abort;
  const bool over = aborted || (a.info->tot.edges > 0 && end_word / 128 + 1 > a.avail_blocks);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    a.info->overflow = over ? (aborted ? 2 : 1) : 0;
    if (over && a.chain) a.chain->abort = 1;
  }
  return over;
}

// fully queued mode: frontier and output offsets come from device memory
__device__ __forceinline__ void resolve_device_state(HopArgs& a) {
  if (a.fs) {
    a.begin = a.fs->slice_b;
    a.frontier = a.fs->slice_e - a.fs->slice_b;
  }
  if (a.rel_size) {
    const int64_t o = *a.rel_size;
    a.e_row += o;
    if (a.e_eid) a.e_eid += o;
  }
}

// One wave per frontier node (_sample, neighbor_kernel.cpp:177-243): any fan-out.  `i` = frontier index of this wave's node.
__device__ __forceinline__ void sample_wave_body(const HopArgs& a, int64_t i) {
  const int lane = threadIdx.x & 63;
  if (i >= a.frontier) return;
  const int64_t src_pos = a.begin + i;
  const int64_t v = a.nodes[src_pos];
  const int64_t src_batch = a.batch ? a.batch[src_pos] : 0;
  const int64_t count = a.count;
  int64_t rs, re_;
  a.range(v, src_pos, count, &rs, &re_);
  const int64_t deg = re_ - rs;
  if (deg <= 0 || count == 0) return;
  const int64_t off = a.edge_off[i];

  if (count < 0 || (!a.replace && count >= deg)) {
    for (int64_t t = lane; t < deg; t += 64) emit(a, off + t, rs + t, src_pos, src_batch);
    return;
  }

  int64_t w0;
  int u0;
  if (a.tab_prefix) {  // fused chain: the node's engine position = the relation's start advanced by its prefix table
    const RngTab tp = a.tab_prefix[i];
    w0 = a.w0 + tab_dw(tp, a.u0);
    u0 = tab_nb(tp, a.u0);
  } else {
    w0 = a.rng_word[i];
    u0 = a.rng_units[i];
  }
  RngCursor rng{w0, u0, a.word_src()};
  if (a.replace) {
    // `count` independent draws from [0, deg) (:196-210); lanes take turns holding a draw
    int64_t mine = 0;
    for (int64_t j = 0; j < count; ++j) {
      const int64_t r = (int64_t)rng.next((u64)deg);
      if (lane == (j & 63)) mine = r;
      if ((j & 63) == 63 || j == count - 1) {
        const int64_t jb = j & ~63ll;
        if (jb + lane <= j) emit(a, off + jb + lane, rs + mine, src_pos, src_batch);
      }
    }
    return;
  }

  // Floyd-style sampling without replacement (:231-240): for i in [deg-count, deg):
  //   r = rand[0, i]; if r already chosen: r = i.
  // The chosen set lives in the lanes (64 picks per round); earlier rounds are re-read from the
  // edge ids already written to global memory.
  int64_t mine = -1;
  for (int64_t j = 0; j < count; ++j) {
    const int64_t idx = deg - count + j;
    int64_t r = (int64_t)rng.next((u64)(idx + 1));
    const int64_t jb = j & ~63ll;
    bool dup = (jb + lane < j) && (mine == r);
    // picks of completed 64-draw rounds of this node were flushed to e_eid[off ..]
    for (int64_t q = lane; q < jb; q += 64) dup = dup || (a.e_eid[off + q] - rs == r);
    if (__any(dup)) r = idx;
    if (lane == (j & 63)) mine = r;
    if ((j & 63) == 63 || j == count - 1) {
      if (jb + lane <= j) emit(a, off + jb + lane, rs + mine, src_pos, src_batch);
      // make this round's edge ids visible to the wave's later history reads
      if (j != count - 1) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
  }
}

__global__ __launch_bounds__(256) void sample_kernel(HopArgs a) {
  if (hop_overflow(a)) return;
  resolve_device_state(a);
  sample_wave_body(a, (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
}

// G lanes per frontier node, 64 / G nodes per wave, for fan-outs 0 < count <= G: lane g owns draw g.
// The wave-per-node kernel above spends every lane on the same serial draw loop (64-bit modulo
// included); here each draw is computed once, and only the "already chosen?" resolution of Floyd's
// algorithm stays sequential -- `count` steps of one shuffle + one ballot.
//
// Position of draw g in the word stream when every draw of the node takes the same n units
// (rand_engine.h:41-76): the current word still serves u / n draws, every later word 4 / n.
template <int G>
__device__ __forceinline__ void sample_group_body(const HopArgs& a, int64_t blk) {
  const int lane = threadIdx.x & 63;
  const int g = lane & (G - 1);
  const int gbase = lane & ~(G - 1);
  const int64_t i = (blk * (int64_t)blockDim.x + threadIdx.x) / G;
  const bool live = i < a.frontier;
  const int64_t count = a.count;  // 0 < count <= G
  int64_t src_pos = 0, src_batch = 0, rs = 0, deg = 0, off = 0;
  if (live) {
    src_pos = a.begin + i;
    const int64_t v = a.nodes[src_pos];
    src_batch = a.batch ? a.batch[src_pos] : 0;
    int64_t re_;
    a.range(v, src_pos, count, &rs, &re_);
    deg = re_ - rs;
    off = a.edge_off[i];
  }
  const bool all = live && deg > 0 && !a.replace && count >= deg;  // whole neighbourhood, no draws
  const bool samp = live && deg > 0 && !all && g < count;
  int64_t pick = -1 - g;  // distinct negatives: never equal to a real draw
  int64_t idx = 0;
  if (samp) {
    idx = a.replace ? deg - 1 : deg - count + g;  // draw g is uniform in [0, idx]
    const u64 range = (u64)idx + 1;
    const int n = need_units(range);
    const int n0 = need_units(a.replace ? (u64)deg : (u64)(deg - count + 1));
    int64_t w0;
    int u0;
    if (a.tab_prefix) {
      const RngTab tp = a.tab_prefix[i];
      w0 = a.w0 + tab_dw(tp, a.u0);
      u0 = tab_nb(tp, a.u0);
    } else {
      w0 = a.rng_word[i];
      u0 = a.rng_units[i];
    }
    u64 val;
    if (n0 == need_units((u64)deg)) {
      // uniform draw width: closed-form position
      const int first = u0 / n;
      int64_t word;
      int unit;
      if (g < first) {
        word = w0;
        unit = (4 - u0) + g * n;
      } else {
        const int t = g - first;
        const int per = 4 / n;
        word = w0 + 1 + t / per;
        unit = (t % per) * n;
      }
      const u64 w = a.word_src().at(word);
      val = w >> (unit * 16);
      if (n == 1) val &= 0xffffull;
      else if (n == 2) val &= 0xffffffffull;
      if ((val >> 32) == 0 && (range >> 32) == 0) val = (u64)((uint32_t)val % (uint32_t)range);
      else val = val % range;
    } else {
      // the draw width changes inside this node's sequence (degree straddles 2^16 / 2^32): walk it
      RngCursor rng{w0, u0, a.word_src()};
      val = 0;
      for (int t = 0; t <= g; ++t) val = rng.next((u64)(deg - count + t) + 1);
    }
    pick = (int64_t)val;
  }
  if (!a.replace) {
    // Floyd (:231-240): draw j is replaced by idx_j when its value was already chosen
    for (int j = 1; j < (int)count; ++j) {
      const int64_t rj = __shfl(pick, gbase + j);
      const bool dup = samp && g < j && pick == rj;
      const u64 m = __ballot(dup);
      const bool hit = ((m >> gbase) & (G == 64 ? ~0ull : ((1ull << G) - 1))) != 0;
      if (samp && g == j && hit) pick = idx;
    }
  }
  if (samp) emit(a, off + g, rs + pick, src_pos, src_batch);
  else if (all && g < deg) emit(a, off + g, rs + g, src_pos, src_batch);
}

template <int G>
__global__ __launch_bounds__(256) void sample_group_kernel(HopArgs a) {
  if (hop_overflow(a)) return;
  resolve_device_state(a);
  sample_group_body<G>(a, blockIdx.x);
}

// owner flag of emission p: it holds the table minimum  <=>  first occurrence of a NEW node
struct FlagLoad {
  const u64* slots;
  const u64* vals;
  const HopInfo* info;  // run-ahead mode: emissions [0, info->tot.edges) exist (none after an overflow)
  __device__ int64_t operator()(int64_t p) const {
    if (info && (p >= info->tot.edges || info->overflow)) return 0;
    return vals[slots[p]] == kProvisional + (u64)p ? 1 : 0;
  }
};

struct AssignStore {
  const u64* slots;
  u64* vals;
  const int64_t* e_node;
  const int64_t* e_batch;
  int64_t* nodes;      // dst node list
  int64_t* batch;      // dst batch list (disjoint) or nullptr
  int64_t size_base;   // dst list size before this hop-relation  (ts == nullptr)
  int64_t id_base;     // dst mapper `curr` before this hop-relation
  int write_nodes;
  const TypeState* ts; // device-resident bases, or nullptr
  __device__ void operator()(int64_t p, const int64_t& rank, const int64_t& flag) const {
    if (!flag) return;
    const int64_t sb = ts ? ts->size : size_base;
    const int64_t ib = ts ? ts->distinct : id_base;
    vals[slots[p]] = (u64)(ib + rank);
    if (write_nodes) {
      nodes[sb + rank] = e_node[p];
      if (batch) batch[sb + rank] = e_batch[p];
    }
  }
};

// the work of hop_boundary_kernel (below), carried by the finalize launch of a hop's last relation
struct HopBoundary {
  TypeState* types;
  int num_types;
  int64_t* rel_sizes;
  int num_rel;
  int next_hop;  // < 0: none
};

// Local ids of every emitted edge.  Thread 0 also closes the relation: publishes its totals to pinned host
// memory and advances the device-resident engine position and node-list sizes for the next relation.
__global__ void finalize_kernel(const u64* __restrict__ slots, const u64* __restrict__ vals,
                                int64_t n, int64_t* __restrict__ out_col, const HopInfo* __restrict__ info,
                                HopInfo* __restrict__ publish, ChainState* __restrict__ chain,
                                TypeState* __restrict__ ts, const int64_t* __restrict__ rel_size,
                                int64_t* __restrict__ rel_size_next, HopBoundary hb = HopBoundary{nullptr, 0, nullptr, 0, -1}) {
  const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t col_off = rel_size ? *rel_size : 0;  // fully queued mode: edges this relation emitted before
  if (p == 0 && rel_size_next) *rel_size_next = col_off + (info->overflow ? 0 : info->tot.edges);
  if (p == 0 && publish) {
    *publish = *info;  // pinned host memory: read by the host after the hop's sync
    if (chain && !info->overflow) {
      if (info->tot.edges > 0) {
        const int u0 = chain->units;
        chain->word += tab_dw(info->tot.tab, u0);
        chain->units = tab_nb(info->tot.tab, u0);
      }
      ts->size += info->uniq;
      ts->distinct += info->uniq;
    }
  }
  if (p == 0 && hb.next_hop >= 0) {
    // last relation of a hop (fully queued mode): what hop_boundary_kernel does, without its launch -- nobody else in
    // this kernel reads the slices or the next hops' rows of rel_sizes, and this thread has just made its own updates
    for (int i = 0; i < hb.num_types; ++i) {
      hb.types[i].slice_b = hb.types[i].slice_e;
      hb.types[i].slice_e = hb.types[i].size;
    }
    for (int i = 0; i < hb.num_rel; ++i)
      hb.rel_sizes[(int64_t)(hb.next_hop + 1) * hb.num_rel + i] = hb.rel_sizes[(int64_t)hb.next_hop * hb.num_rel + i];
  }
  if (info && (p >= info->tot.edges || info->overflow)) return;
  if (p < n) out_col[col_off + p] = (int64_t)vals[slots[p]];
}

// Hop boundary (fully queued mode): the nodes appended during the hop become the next frontier, and every
// relation's output offset is carried over to the next hop's slot (finalize_kernel overwrites it for the
// relations that run in that hop).  rel_sizes is [hops + 1][num_rel].
__global__ void hop_boundary_kernel(TypeState* t, int n, int advance, int64_t* rel_sizes, int num_rel, int next_hop) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (advance && i < n) {
    t[i].slice_b = t[i].slice_e;
    t[i].slice_e = t[i].size;
  }
  if (i < num_rel) rel_sizes[(int64_t)(next_hop + 1) * num_rel + i] = rel_sizes[(int64_t)next_hop * num_rel + i];
}

__global__ void init_state_kernel(ChainState* c, int64_t word, int units, TypeState* t, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    c->word = word;
    c->units = units;
    c->abort = 0;
  }
  if (i < n) {
    t[i].size = 0;
    t[i].distinct = 0;
    t[i].slice_b = 0;
    t[i].slice_e = 0;
  }
}
__global__ void clear_abort_kernel(ChainState* c) { c->abort = 0; }

__global__ void interleave_kernel(const int64_t* __restrict__ batch,
                                  const int64_t* __restrict__ node, int64_t n,
                                  int64_t* __restrict__ out) {
  const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p < n) {
    out[2 * p] = batch[p];
    out[2 * p + 1] = node[p];
  }
}

// fan-outs up to 64 use sub-wave groups (8/16/32/64 lanes per node); "all neighbours" and larger
// fan-outs one wave per node
void launch_sample(const HopArgs& a, int64_t F, hipStream_t stream) {
  const int64_t c = a.count;
  if (c > 0 && c <= 8)
    hipLaunchKernelGGL(sample_group_kernel<8>, dim3((unsigned)((F + 31) / 32)), dim3(256), 0, stream, a);
  else if (c > 0 && c <= 16)
    hipLaunchKernelGGL(sample_group_kernel<16>, dim3((unsigned)((F + 15) / 16)), dim3(256), 0, stream, a);
  else if (c > 0 && c <= 32)
    hipLaunchKernelGGL(sample_group_kernel<32>, dim3((unsigned)((F + 7) / 8)), dim3(256), 0, stream, a);
  else if (c > 0 && c <= 64)
    hipLaunchKernelGGL(sample_group_kernel<64>, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, stream, a);
}

#include "sampler_biased.h"  // biased (edge_weight) sampling kernels

#include "sampler_fused.h"

// ---- host driver -----------------------------------------------------------------------------------
// growable device array of int64
struct DevVec {
  int64_t* p = nullptr;
  int64_t size = 0;
  int64_t cap = 0;
  int64_t live = 0;  // elements that may hold data of relations not yet committed by the host (>= size)
  // `hint`: expected final size (later hops included), so that one allocation usually lasts the call
  int reserve(Ctx& c, int64_t n, int64_t hint = 0) {
    if (n <= cap) return PYG_HIP_OK;
    const int64_t ncap = std::max<int64_t>(std::max<int64_t>(n, hint), std::max<int64_t>(2 * cap, 1024));
    int64_t* np;
    PYG_ALLOC(np, int64_t*, c, sizeof(int64_t) * (size_t)ncap);
    const int64_t keep = std::min<int64_t>(std::max(size, live), cap);
    if (keep > 0)
      PYG_HIP_CHECK(hipMemcpyAsync(np, p, sizeof(int64_t) * (size_t)keep, hipMemcpyDeviceToDevice,
                                   c.stream));
    c.release(p);
    p = np;
    cap = ncap;
    return PYG_HIP_OK;
  }
};

struct NodeSet {
  DevVec nodes, batch;
  int64_t distinct = 0;  // Mapper::curr
  int64_t slice_b = 0, slice_e = 0;
  HashTable table{nullptr, nullptr, 0, 0, kProvisional, 0, ~0ull};
  void* cached = nullptr;  // the table block belongs to the library's DenseCache (not to this call's allocations)
  int64_t entries_bound = 0;  // upper bound of keys present in the table
  int64_t dense_n = 0;        // > 0: every id of this type is < dense_n and keys are plain node ids (not disjoint)
};

typedef u64 u64x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void table_clear_kernel(u64x2* __restrict__ p, int64_t n16) {
  const u64x2 e = {kEmpty, kEmpty};
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) p[i] = e;
}

// Direct-address tables kept between calls (fused chain only).  A products-sized table is 19.6 MB: clearing it took
// 11 - 13 us and 7 % of a C3 batch's HBM traffic per call (C5: four tables, four launches).  The cache hands a call a
// table whose stale contents are invalidated by the epoch in the value coding (HashTable above); a block is cleared
// once, when it is allocated, and again when its 2^20 - 1 epochs are used up.  An entry serves one call at a time
// (`busy`; a second concurrent call on the same device and node count gets a classic table); it is handed back by
// Ctx::release_all, i.e. after the call's last synchronisation -- no kernel of the call touches it afterwards.  Blocks
// come from the caller's allocator (host->alloc) and stay allocated for the life of the process (at most
// kDenseCacheEntries x 128 MiB per device).
struct DenseCacheEntry {
  int device;
  int64_t dense_n;
  u64* ptr;
  u64 epoch;
  bool busy;
};
constexpr int kDenseCacheEntries = 8;
constexpr int64_t kDenseCacheMaxN = 1ll << 24;  // 128 MiB per table
inline std::mutex& dense_cache_mutex() {
  static std::mutex m;
  return m;
}
inline std::vector<DenseCacheEntry>& dense_cache() {
  static std::vector<DenseCacheEntry> v;
  return v;
}
inline bool dense_cache_enabled() {
  static const bool on = [] {
    const char* e = getenv("PYG_HIP_SAMPLER_TABLE_CACHE");
    return !(e != nullptr && e[0] == '0');
  }();
  return on;
}
std::atomic<u64> g_epoch_limit{kEpochMax};  // epochs per clear of a cached table (lowered by tests to exercise the wrap)
}  // namespace
namespace sampler {
void dense_cache_release(void* ptr) {  // declared in sampler_rng.h (Ctx::release_all)
  std::lock_guard<std::mutex> lock(dense_cache_mutex());
  for (DenseCacheEntry& e : dense_cache())
    if (e.ptr == ptr) e.busy = false;
}
}  // namespace sampler
namespace {

// keys and vals share one block (one allocation, one memset); `hint` = entries expected by the end of
// the call, so that the table is usually built once instead of being rehashed every hop.  `epoch_ok`: the caller's
// kernels all read the value coding from the HashTable (the fused chain), so a cached epoch table may be used.
int table_reserve(Ctx& c, NodeSet& ns, int64_t extra, int64_t hint = 0, bool epoch_ok = false) {
  const int64_t need = ns.entries_bound + extra;
  if (ns.table.keys && ns.table.dense) {
    ns.entries_bound = need;
    return PYG_HIP_OK;
  }
  u64 cap = ns.table.keys ? ns.table.mask + 1 : 0;
  if (cap >= 2 * (u64)need && cap > 0) {
    ns.entries_bound = need;
    return PYG_HIP_OK;
  }
  const int64_t want = std::max<int64_t>(need, std::min<int64_t>(hint, 1ll << 20));
  u64 ncap = 1024;
  while (ncap < 2 * (u64)need || ncap < 2 * (u64)want) ncap <<= 1;  // load factor <= 0.5
  if (!ns.table.keys && ns.dense_n > 0 && (u64)ns.dense_n <= 4 * ncap && ns.dense_n <= (1ll << 27)) {
    // direct-address table: at most twice the bytes of the hash table it replaces
    HashTable dt;
    dt.mask = 0;
    dt.dense = 1;
    dt.prov = kProvisional;
    dt.tag = 0;
    dt.idmask = ~0ull;
    auto clear = [&](u64* p) -> int {
      // cleared by an own kernel: the runtime's fill kernel takes 12 us for the 19.6 MB of a products-sized table (1.6 TB/s)
      const int64_t n16 = (ns.dense_n + 1) / 2;
      hipLaunchKernelGGL(table_clear_kernel, dim3((unsigned)std::min<int64_t>((n16 + 255) / 256, 1 << 20)), dim3(256), 0,
                         c.stream, reinterpret_cast<u64x2*>(p), n16);
      PYG_HIP_CHECK(hipGetLastError());
      return PYG_HIP_OK;
    };
    if (epoch_ok && dense_cache_enabled() && ns.dense_n <= kDenseCacheMaxN) {
      int dev = 0;
      PYG_HIP_CHECK(hipGetDevice(&dev));
      bool room = false;
      u64* block = nullptr;
      u64* evict = nullptr;
      u64 epoch = 0;
      {
        std::lock_guard<std::mutex> lock(dense_cache_mutex());
        int count = 0;
        for (DenseCacheEntry& e : dense_cache()) {
          count += e.device == dev ? 1 : 0;
          if (e.device == dev && e.dense_n == ns.dense_n && !e.busy && !block) {
            e.busy = true;
            if (e.epoch >= g_epoch_limit.load()) e.epoch = 0;
            epoch = ++e.epoch;
            block = e.ptr;
          }
        }
        room = count < kDenseCacheEntries;
        if (!block && !room) {
          // full, none of this size free: drop an idle table of ANOTHER node count (graphs that come and go must not
          // leave the cache full of dead blocks); tables of this size that are merely busy stay
          auto& v = dense_cache();
          for (size_t i = 0; i < v.size(); ++i)
            if (v[i].device == dev && !v[i].busy && v[i].dense_n != ns.dense_n) {
              evict = v[i].ptr;
              v.erase(v.begin() + (long)i);
              room = true;
              break;
            }
        }
      }
      if (evict) c.host->free(c.host->user, evict);  // idle: its last user synchronised before handing it back
      if (block) {
        if (epoch == 1) {  // wrapped around: stale entries of the old numbering would look current
          int rc = clear(block);
          if (rc != PYG_HIP_OK) return rc;
        }
      } else if (room) {
        PYG_ALLOC(block, u64*, c, sizeof(u64) * (size_t)(ns.dense_n + (ns.dense_n & 1)));
        c.keep(block);  // owned by the cache from here on
        int rc = clear(block);
        if (rc != PYG_HIP_OK) return rc;
        epoch = 1;
        std::lock_guard<std::mutex> lock(dense_cache_mutex());
        dense_cache().push_back(DenseCacheEntry{dev, ns.dense_n, block, epoch, true});
      }
      if (block) {
        dt.keys = block;
        dt.vals = block;
        dt.tag = (kEpochMax - epoch) << kEpochShift;
        dt.prov = dt.tag | kEpochProv;
        dt.idmask = kEpochProv - 1;
        ns.table = dt;
        ns.cached = block;
        c.cached_tables.push_back(block);
        ns.entries_bound = need;
        return PYG_HIP_OK;
      }
    }
    PYG_ALLOC(dt.keys, u64*, c, sizeof(u64) * (size_t)(ns.dense_n + (ns.dense_n & 1)));  // whole 16-byte stores
    dt.vals = dt.keys;
    {
      int rc = clear(dt.keys);
      if (rc != PYG_HIP_OK) return rc;
    }
    ns.table = dt;
    ns.entries_bound = need;
    return PYG_HIP_OK;
  }
  HashTable nt;
  nt.dense = 0;
  nt.prov = kProvisional;
  nt.tag = 0;
  nt.idmask = ~0ull;
  PYG_ALLOC(nt.keys, u64*, c, sizeof(u64) * 2 * ncap);
  nt.vals = nt.keys + 1;  // interleaved slots: (key, value) pairs
  nt.mask = ncap - 1;
  PYG_HIP_CHECK(hipMemsetAsync(nt.keys, 0xFF, sizeof(u64) * 2 * ncap, c.stream));
  if (ns.table.keys) {
    hipLaunchKernelGGL(rehash_kernel, dim3((unsigned)std::min<u64>((cap + 255) / 256, 4096)),
                       dim3(256), 0, c.stream, ns.table, nt);
    PYG_HIP_CHECK(hipGetLastError());
    c.release(ns.table.keys);
  }
  ns.table = nt;
  ns.entries_bound = need;
  return PYG_HIP_OK;
}


constexpr int kMaxFusedCount = 1024;  // largest fan-out the fused chain samples (12 bits of its cached word; > 64: one wave per node)
constexpr int kWideSlackWords = 64;  // random words per (hop, relation) the speculation adds for rows of degree >= 2^16
                                    // (their draws take 32 bits: two per word instead of four)
constexpr int kNeedSlow = 1000;    // internal: repeat the call in the synchronising mode
constexpr int kNeedQueued = 1001;  // internal: repeat the call through round 2's fully queued chain (no fused chain)

// which driver the calling thread's last sampler call ran (pyg_hip_sampler_last_mode): tests and benchmarks assert it
thread_local const char* g_sampler_mode = "none";

struct RelState {
  DevVec row, col, eid;
  std::vector<int64_t> edges_per_hop;
};

#include "sampler_fused_host.h"

int run_sampler(int num_node_types, int num_relations, const pyg_hip_relation* rels,
                int num_seed_sets, const pyg_hip_seed_set* seeds, const int64_t* const* node_time,
                int temporal_last, int L, int csc, int replace, int disjoint, int return_edge_id,
                Ctx& c, pyg_hip_sample_result* res, bool allow_fast, bool allow_fused = true) {
  hipStream_t stream = c.stream;
  std::vector<NodeSet> ns((size_t)num_node_types);
  std::vector<RelState> rs((size_t)num_relations);
  std::vector<std::vector<int64_t>> nodes_per_hop((size_t)num_node_types);
  RngHost rng;
  MtHandBack* hand_back = nullptr;  // pinned: engine state written by mt_finish_chain_kernel (fully queued mode)
  PhaseTimer pt;

  int64_t num_batches = 1;
  if (disjoint) {
    num_batches = 0;
    for (int s = 0; s < num_seed_sets; ++s) num_batches += seeds[s].num_seed;
    if (num_batches < 1) num_batches = 1;
    PYG_HIP_REQUIRE(num_batches < (1ll << 22), "sampler: too many seeds for disjoint sampling");
  } else {
    // node count of every type that some relation expands (rowptr has one row per node of that type): ids of that
    // type are < that count -- the precondition of the reference's own dense Mapper (mapper.h:17-24, 609-612) --
    // so its table can be direct-addressed (table_reserve decides by size)
    for (int e = 0; e < num_relations; ++e) {
      const int src = !csc ? rels[e].src_type : rels[e].dst_type;
      if (src >= 0 && src < num_node_types)
        ns[(size_t)src].dense_n = std::max<int64_t>(ns[(size_t)src].dense_n, rels[e].num_rows);
    }
  }

  const size_t hand_back_offset =
      align_up(1024 + sizeof(HopInfo) * (size_t)std::max(num_relations * std::max(L, 1), 96), 64);
  // fused chain (sampler_fused.h): host copy of its tables
  const size_t fused_tables_offset = align_up(hand_back_offset + sizeof(MtHandBack), 256) + 256;   // (the 64 bytes in front: its completion word)
  const size_t fused_tables_bytes =
      align_up(8 * (size_t)(L + 1) * num_node_types + 8 * (size_t)num_node_types + 20 * (size_t)std::max(L, 1) * num_relations + 64, 256);
  void* pinned = nullptr;
  {
    // scratch + one HopInfo per (hop, relation) + the engine hand-back of the fully queued mode + the fused chain's blocks
    int rc = get_pinned(&pinned, fused_tables_offset + fused_tables_bytes);
    if (rc != PYG_HIP_OK) return rc;
  }

  // temporal sampling: one seed time per disjoint subgraph (batch id)
  bool temporal = false;
  for (int t = 0; t < num_node_types && node_time; ++t) temporal = temporal || node_time[t] != nullptr;
  for (int e = 0; e < num_relations; ++e) temporal = temporal || rels[e].edge_time != nullptr;
  int64_t* seed_times = nullptr;
  int* err_flag = nullptr;
  if (temporal) {
    PYG_HIP_REQUIRE(disjoint, "Temporal sampling needs to create disjoint subgraphs");
    PYG_ALLOC(seed_times, int64_t*, c, sizeof(int64_t) * (size_t)num_batches);
    // kernels raise the flag straight in pinned host memory: no copy, no extra synchronisation
    err_flag = reinterpret_cast<int*>(static_cast<char*>(pinned) + 256);
    *err_flag = 0;
  }

  pt.mark("pinned");
  // device-resident engine position and node-list sizes (see the hop loop)
  ChainState* chain;
  TypeState* tstate;
  PYG_ALLOC(chain, ChainState*, c, sizeof(ChainState));
  PYG_ALLOC(tstate, TypeState*, c, sizeof(TypeState) * (size_t)num_node_types);

  // Upper bounds from the seeds and the fan-out products: frontier size per (hop, type), emitted edges per
  // (hop, relation).  With bounded fan-outs they size everything up front (fully queued mode below).
  bool fast = allow_fast && c.host->mt19937 != nullptr && L > 0 && num_relations > 0;
  bool any_biased = false;
  for (int e = 0; e < num_relations; ++e) any_biased = any_biased || rels[e].edge_weight != nullptr;
  if (any_biased) {
    // reference checks (neighbor_kernel.cpp:377-380,579-582)
    PYG_HIP_REQUIRE(!temporal, "Biased temporal sampling not yet supported");
    if (!c.host->mt19937) return fail(PYG_HIP_ERR_UNSUPPORTED, "sampler: biased sampling needs the mt19937 engine state (host->mt19937)");
    for (int e = 0; e < num_relations; ++e) {
      if (rels[e].edge_weight)
        PYG_HIP_REQUIRE(rels[e].edge_weight_dtype == PYG_F32 || rels[e].edge_weight_dtype == PYG_F64,
                        "sampler: edge_weight must be float32 or float64");
    }
    fast = false;  // the draws of a hop are only known after its degree scan
  }
  std::vector<std::vector<int64_t>> eb((size_t)L, std::vector<int64_t>((size_t)num_relations, 0));
  std::vector<std::vector<int64_t>> fbh((size_t)L, std::vector<int64_t>((size_t)num_node_types, 0));
  std::vector<int64_t> node_bound((size_t)num_node_types, 0), rel_bound((size_t)num_relations, 0);
  bool big_count = false;
  if (fast) {
    std::vector<int64_t> fb((size_t)num_node_types, 0);
    for (int s = 0; s < num_seed_sets; ++s)
      if (seeds[s].node_type >= 0 && seeds[s].node_type < num_node_types)
        fb[(size_t)seeds[s].node_type] += seeds[s].num_seed;
    int64_t total = 0;
    for (int ell = 0; ell < L && fast; ++ell) {
      fbh[(size_t)ell] = fb;
      std::vector<int64_t> nf((size_t)num_node_types, 0);
      for (int e = 0; e < num_relations; ++e) {
        const int src = !csc ? rels[e].src_type : rels[e].dst_type;
        const int dst = !csc ? rels[e].dst_type : rels[e].src_type;
        const int64_t count = rels[e].num_neighbors_host[ell];
        // (fan-outs of 65 ... kMaxFusedCount: one wave per node -- inside the fused chain only, decided below)
        if (count < 0 || count > kMaxFusedCount) fast = false;
        else if (count > 64) big_count = true;
        if (count <= 0 || fb[(size_t)src] == 0 || rels[e].num_cols == 0) continue;
        const int64_t b = fb[(size_t)src] * count;
        eb[(size_t)ell][(size_t)e] = b;
        nf[(size_t)dst] += b;
        node_bound[(size_t)dst] += b;
        rel_bound[(size_t)e] += b;
        total += b;
        if (total > (24ll << 20)) fast = false;  // bounds too loose to pre-size everything
      }
      fb.swap(nf);
    }
  }

  const bool fused = fast && allow_fused && fused_eligible(rels, num_relations, num_node_types, num_seed_sets, csc, L, eb);
  if (big_count && !fused) fast = false;  // the queued chain keeps its limit of 64 draws per node
  std::vector<FusedSeed> fseeds;
  if (!fused) {
    // the chains below read and advance the device-resident engine position / type states; the fused chain keeps its
    // state in write-once tables (fused_init_kernel) and only WRITES `chain` (fold) and `tstate` (seed insert)
    hipLaunchKernelGGL(init_state_kernel, dim3((unsigned)((num_node_types + 63) / 64)), dim3(64), 0, stream, chain,
                       rng.word, rng.units, tstate, num_node_types);
    PYG_HIP_CHECK(hipGetLastError());
  }

  // The word generation (side stream).  Round 2's chain queues it behind the seed kernels (which do not need it; its
  // launches would otherwise sit in front of them on the host); the fused chain, whose hops follow each other within
  // ~40 us, needs the round's later segments as early as possible and starts it first.
  auto start_rng = [&](bool prepare_only) -> int {
  if (c.host->mt19937) {
    // upper bounds of the words each hop can consume: every frontier node draws `count` 16-bit numbers
    std::vector<int64_t> spec;
    std::vector<double> fb((size_t)num_node_types, 0.0);
    for (int s = 0; s < num_seed_sets; ++s)
      if (seeds[s].node_type >= 0 && seeds[s].node_type < num_node_types)
        fb[(size_t)seeds[s].node_type] += (double)seeds[s].num_seed;
    double cum = 128.0;
    for (int ell = 0; ell < L; ++ell) {
      std::vector<double> nf((size_t)num_node_types, 0.0);
      double draws = 0.0;
      bool open_ended = false;
      for (int e = 0; e < num_relations; ++e) {
        const int src = !csc ? rels[e].src_type : rels[e].dst_type;
        const int dst = !csc ? rels[e].dst_type : rels[e].src_type;
        const int64_t count = rels[e].num_neighbors_host[ell];
        if (count < 0) {
          open_ended = open_ended || fb[(size_t)src] > 0;
          continue;
        }
        draws += fb[(size_t)src] * (double)count;
        nf[(size_t)dst] += fb[(size_t)src] * (double)count;
        if (fb[(size_t)src] > 0 && count > 0) cum += (double)kWideSlackWords;  // (the fused chain's allowance for 32-bit draws)
      }
      cum += draws / 4.0 + 128.0;
      spec.push_back((int64_t)std::min<double>(cum, (double)kSpecCapWords));
      if (open_ended || cum >= (double)kSpecCapWords) break;  // later frontiers are unbounded / too far ahead
      fb.swap(nf);
    }
    int rc = prepare_only ? rng_begin_prepare(c, rng, pinned, spec, true) : rng_begin(c, rng, pinned, spec);
    if (rc != PYG_HIP_OK) return rc;
  } else {
    // The engine constructor always prefetches one block (rand_engine.h:27-29), sampled or not.
    int rc = rng_ensure(c, rng, 0);
    if (rc != PYG_HIP_OK) return rc;
  }

    return PYG_HIP_OK;
  };
  pt.mark("bounds");
  // Fused chain: the round's launches (three kernels, two events: ~12 us of host time) go BEHIND the seeds' scan launch
  // (run_fused_chain) -- the scan is the head of the main stream's critical path and needs no random word, hop 0 needs
  // the prefix kernel only, and the later segments have ~30 us of slack before hop 1 reads them (kernel timeline,
  // profiles/NOTES_r6.md).  Here: engine adopted, blocks allocated, side stream ordered behind this point.
  const bool rng_late = fused && c.host->mt19937 != nullptr;
  if (fused) {
    int rc = start_rng(rng_late);
    if (rc != PYG_HIP_OK) return rc;
  }
  pt.mark("rng_started");

  // ---- seeds ----
  // fused chain, one seed set of one tile (the usual mini-batch): its insertion rides in the seeds' scan launch
  const bool fold_seeds = fused && fused_seed_foldable(seeds, num_seed_sets);
  int64_t batch0 = 0;
  for (int s = 0; s < num_seed_sets; ++s) {
    const pyg_hip_seed_set& ss = seeds[s];
    PYG_HIP_REQUIRE(ss.node_type >= 0 && ss.node_type < num_node_types, "sampler: bad seed type");
    NodeSet& n = ns[(size_t)ss.node_type];
    PYG_HIP_REQUIRE(n.nodes.size == 0, "sampler: node type seeded twice");
    const int64_t S = ss.num_seed;
    n.slice_b = 0;
    n.slice_e = S;
    if (S == 0) continue;
    // fully queued mode: the list and the table are created at their final (bound) size right away
    const int64_t nb = fast ? node_bound[(size_t)ss.node_type] : 0;
    int rc = n.nodes.reserve(c, S, S + nb);
    if (rc != PYG_HIP_OK) return rc;
    if (disjoint) {
      rc = n.batch.reserve(c, S, S + nb);
      if (rc != PYG_HIP_OK) return rc;
    }
    rc = table_reserve(c, n, fused ? S + nb : S, S + nb, fused);  // fused chain: sized once (the seeds' slot handles stay valid)
    if (rc != PYG_HIP_OK) return rc;
    u64* slots;
    PYG_ALLOC(slots, u64*, c, sizeof(u64) * (size_t)S);
    if (!fold_seeds) {
      hipLaunchKernelGGL(seed_insert_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, stream,
                         ss.seed, S, batch0, disjoint, num_batches, n.table, n.nodes.p,
                         disjoint ? n.batch.p : (int64_t*)nullptr, slots, tstate + ss.node_type);
      PYG_HIP_CHECK(hipGetLastError());
    }
    if (temporal) {
      const int64_t* nt = node_time ? node_time[ss.node_type] : nullptr;
      if (ss.seed_time || nt) {
        hipLaunchKernelGGL(seed_time_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, stream,
                           ss.seed, ss.seed_time, nt, S, batch0, seed_times);
        PYG_HIP_CHECK(hipGetLastError());
      } else {
        // the reference would index an empty seed_times vector here (undefined behaviour)
        return fail(PYG_HIP_ERR_INVALID, "Seed time needs to be specified");
      }
    }
    if (fused) {
      // the seeds' first-occurrence scan also carries hop 0's counts: queued with the chain (run_fused_chain)
      fseeds.push_back(FusedSeed{ss.node_type, S, slots, ss.seed, batch0, tstate + ss.node_type, fold_seeds});
      n.nodes.size = S;
      if (disjoint) n.batch.size = S;
      batch0 += S;
      continue;
    }
    // ranks of first occurrences = local ids (duplicates keep their first id)
    const int64_t ntiles = (S + kScanTile - 1) / kScanTile;
    int64_t* tile_buf;
    PYG_ALLOC(tile_buf, int64_t*, c, sizeof(int64_t) * (size_t)(ntiles + 1));
    FlagLoad fl{slots, n.table.vals};
    AssignStore as{slots, n.table.vals, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    // the number of distinct seeds (Mapper::curr) goes straight into the device-resident type state: the
    // host never needs it, so the seeds cost no synchronisation
    rc = device_scan<int64_t, SumOp>(fl, as, S, tile_buf, &tstate[ss.node_type].distinct, stream);
    if (rc != PYG_HIP_OK) return rc;
    n.nodes.size = S;
    if (disjoint) n.batch.size = S;
    c.release(slots);
    c.release(tile_buf);
    batch0 += S;
  }
  if (!fused) {  // (the fused chain prepared it before the seeds and launches it behind their scan)
    int rc = start_rng(false);
    if (rc != PYG_HIP_OK) return rc;
  }
  for (int t = 0; t < num_node_types; ++t) nodes_per_hop[(size_t)t].push_back(ns[(size_t)t].nodes.size);
  pt.lap(0);
  pt.mark("seeds_queued");

  // ---- hops ----
  // Engine position and node-list sizes live on the device (ChainState / TypeState): the relations of a
  // hop are queued back to back -- count scan, sample, dedup scan, finalize, each starting where the
  // previous one ended -- and the host synchronises ONCE per hop to read every relation's totals.
  // A relation is "committed" when the host has folded its totals into its own bookkeeping.
  volatile HopInfo* info_host = reinterpret_cast<volatile HopInfo*>(static_cast<char*>(pinned) + 1024);
  HopInfo* info_dev;
  PYG_ALLOC(info_dev, HopInfo*, c, sizeof(HopInfo) * (size_t)std::max(num_relations * std::max(L, 1), 1));

  // ---- fully queued mode -------------------------------------------------------------------------------
  // With bounded fan-outs everything is sized from the upper bounds computed above, the frontier of every
  // hop is read from device memory, and ALL hops are queued without a single intermediate
  // synchronisation; the host reads every (hop, relation) total at the end.  Scratch comes from ONE
  // allocation (the host is the bottleneck of this mode: ~40 allocator calls and ~45 launches otherwise).
  // A relation that lacks random words (draws wider than 16 bits) aborts the rest of the queue untouched
  // and the call is repeated in the synchronising mode below (kNeedSlow).
  bool done_fast = false;
  if (fused) {
    int rc = run_fused_chain(c, num_node_types, num_relations, rels, L, csc, replace, disjoint, num_batches, node_time,
                             temporal_last, seed_times, err_flag, ns, rs, eb, fbh, node_bound, rel_bound, fseeds, rng,
                             rng_late, chain,
                             static_cast<char*>(pinned) + fused_tables_offset,
                             reinterpret_cast<MtHandBack*>(static_cast<char*>(pinned) + hand_back_offset), &hand_back,
                             nodes_per_hop, pt);
    if (rc != PYG_HIP_OK) return rc;  // kNeedSlow included
    g_sampler_mode = "fused";
    done_fast = true;
  } else if (fast) {
    struct Queued {
      int ell, e;
    };
    std::vector<Queued> queued;
    size_t arena_bytes = align_up(sizeof(int64_t) * (size_t)(L + 1) * (size_t)num_relations, 256);
    auto tiles = [](int64_t n) { return (n + kScanTile - 1) / kScanTile + 1; };
    for (int ell = 0; ell < L; ++ell)
      for (int e = 0; e < num_relations; ++e) {
        const int64_t Eb = eb[(size_t)ell][(size_t)e];
        if (Eb == 0) continue;
        const int src = !csc ? rels[e].src_type : rels[e].dst_type;
        const int64_t Fb = fbh[(size_t)ell][(size_t)src];
        arena_bytes += align_up(sizeof(CountAgg) * (size_t)tiles(Fb), 256) + 2 * align_up(8 * (size_t)Fb, 256) +
                       align_up(4 * (size_t)Fb, 256) + (disjoint ? 3 : 2) * align_up(8 * (size_t)Eb, 256) +
                       align_up(8 * (size_t)tiles(Eb), 256) +
                       align_up(sizeof(CountAgg) * (size_t)Fb, 256) + align_up(8 * (size_t)Eb, 256);  // scan value caches
      }
    char* arena;
    PYG_ALLOC(arena, char*, c, arena_bytes);
    auto carve = [&](size_t bytes) {
      char* p = arena;
      arena += align_up(bytes, 256);
      return p;
    };
    int64_t* rel_sizes = reinterpret_cast<int64_t*>(carve(sizeof(int64_t) * (size_t)(L + 1) * (size_t)num_relations));
    // rows 0 and 1: hop 0 starts from empty relations (what hop_boundary_kernel(next_hop = 0) would copy over)
    PYG_HIP_CHECK(hipMemsetAsync(rel_sizes, 0, sizeof(int64_t) * 2 * (size_t)num_relations, stream));
    for (int t = 0; t < num_node_types; ++t) {
      NodeSet& n = ns[(size_t)t];
      if (node_bound[(size_t)t] == 0) continue;
      n.nodes.live = n.nodes.size;
      int rc = n.nodes.reserve(c, n.nodes.size + node_bound[(size_t)t]);
      if (rc != PYG_HIP_OK) return rc;
      if (disjoint) {
        n.batch.live = n.batch.size;
        rc = n.batch.reserve(c, n.batch.size + node_bound[(size_t)t]);
        if (rc != PYG_HIP_OK) return rc;
      }
      rc = table_reserve(c, n, node_bound[(size_t)t]);
      if (rc != PYG_HIP_OK) return rc;
    }
    for (int e = 0; e < num_relations; ++e) {
      RelState& st = rs[(size_t)e];
      if (rel_bound[(size_t)e] == 0) continue;
      int rc = st.row.reserve(c, rel_bound[(size_t)e]);
      if (rc == PYG_HIP_OK) rc = st.col.reserve(c, rel_bound[(size_t)e]);
      if (rc == PYG_HIP_OK) rc = st.eid.reserve(c, rel_bound[(size_t)e]);
      if (rc != PYG_HIP_OK) return rc;
    }
    int64_t avail_blocks = 0;
    int64_t spec_word = rng.word;
    bool boundary_done = true;  // hop 0 needs none (rows 0 and 1 of rel_sizes are zero, the slices are the seeds)
    for (int ell = 0; ell < L; ++ell) {
      if (!boundary_done) {  // the previous hop queued no relation that could carry it
        hipLaunchKernelGGL(hop_boundary_kernel, dim3((unsigned)((std::max(num_node_types, num_relations) + 63) / 64)),
                           dim3(64), 0, stream, tstate, num_node_types, 1, rel_sizes, num_relations, ell);
        PYG_HIP_CHECK(hipGetLastError());
      }
      boundary_done = ell + 1 >= L;  // nothing follows the last hop
      int last_rel = -1;
      for (int e = 0; e < num_relations; ++e)
        if (eb[(size_t)ell][(size_t)e] != 0) last_rel = e;
      for (int e = 0; e < num_relations; ++e) {
        const int64_t Eb = eb[(size_t)ell][(size_t)e];
        if (Eb == 0) continue;
        const pyg_hip_relation& r = rels[e];
        const int src = !csc ? r.src_type : r.dst_type;
        const int dst = !csc ? r.dst_type : r.src_type;
        NodeSet& sn = ns[(size_t)src];
        NodeSet& dn = ns[(size_t)dst];
        RelState& st = rs[(size_t)e];
        const int64_t count = r.num_neighbors_host[ell];
        const int64_t Fb = fbh[(size_t)ell][(size_t)src];
        const int slot = ell * num_relations + e;
        CountAgg* tile_buf = reinterpret_cast<CountAgg*>(carve(sizeof(CountAgg) * (size_t)tiles(Fb)));
        int64_t* edge_off = reinterpret_cast<int64_t*>(carve(8 * (size_t)Fb));
        int64_t* rng_word = reinterpret_cast<int64_t*>(carve(8 * (size_t)Fb));
        int32_t* rng_units = reinterpret_cast<int32_t*>(carve(4 * (size_t)Fb));
        int64_t* e_node = reinterpret_cast<int64_t*>(carve(8 * (size_t)Eb));
        int64_t* e_batch = disjoint ? reinterpret_cast<int64_t*>(carve(8 * (size_t)Eb)) : nullptr;
        u64* e_slot = reinterpret_cast<u64*>(carve(8 * (size_t)Eb));
        int64_t* ftile = reinterpret_cast<int64_t*>(carve(8 * (size_t)tiles(Eb)));
        // values of the scans' first pass, kept for the second (their loads are dependent random gathers)
        CountAgg* count_cache = reinterpret_cast<CountAgg*>(carve(sizeof(CountAgg) * (size_t)Fb));
        int64_t* flag_cache = reinterpret_cast<int64_t*>(carve(8 * (size_t)Eb));
        RangeCtx range;
        range.rowptr = IdxArr(r.rowptr, r.index_is32);
        range.col = IdxArr(r.col, r.index_is32);
        range.time = r.edge_time ? r.edge_time : (node_time ? node_time[dst] : nullptr);
        range.edge_level = r.edge_time ? 1 : 0;
        range.last = temporal_last;
        range.seed_times = seed_times;
        range.batch = disjoint ? sn.batch.p : nullptr;
        range.error = err_flag;
        CountLoad cl{sn.nodes.p, 0, range, count, replace, tstate + src};
        CountStore cs{edge_off, rng_word, rng_units, 0, 4, chain};
        int rc = device_scan<CountAgg, CountOp>(cl, cs, Fb, tile_buf, &info_dev[slot].tot, stream, count_cache);
        if (rc != PYG_HIP_OK) return rc;
        // order the words this relation may read (16-bit draws, cumulative bound) before its sample kernel:
        // the first hops only wait for the first segment of the round, not for all of it
        spec_word += (Eb + 3) / 4 + 1;
        rc = rng_wait(c, rng, spec_word, &avail_blocks);
        if (rc != PYG_HIP_OK) return rc;
        HopArgs a;
        a.info = info_dev + slot;
        a.chain = chain;
        a.fs = tstate + src;
        a.rel_size = rel_sizes + (int64_t)ell * num_relations + e;
        a.avail_blocks = avail_blocks;
        a.nodes = sn.nodes.p;
        a.batch = disjoint ? sn.batch.p : nullptr;
        a.begin = 0;
        a.frontier = Fb;
        a.range = range;
        a.col = IdxArr(r.col, r.index_is32);
        a.count = count;
        a.replace = replace;
        a.num_batches = num_batches;
        a.edge_off = edge_off;
        a.rng_word = rng_word;
        a.rng_units = rng_units;
        a.words = rng.dev;
        a.e_row = st.row.p;
        a.e_node = e_node;
        a.e_batch = e_batch;
        a.e_eid = st.eid.p;
        a.e_slot = e_slot;
        a.table = dn.table;
        launch_sample(a, Fb, stream);
        PYG_HIP_CHECK(hipGetLastError());
        FlagLoad fl{e_slot, dn.table.vals, info_dev + slot};
        AssignStore as{e_slot, dn.table.vals, e_node, e_batch, dn.nodes.p,
                       disjoint ? dn.batch.p : (int64_t*)nullptr, 0, 0, 1, tstate + dst};
        rc = device_scan<int64_t, SumOp>(fl, as, Eb, ftile, &info_dev[slot].uniq, stream, flag_cache);
        if (rc != PYG_HIP_OK) return rc;
        hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((Eb + 255) / 256)), dim3(256), 0, stream, e_slot,
                           dn.table.vals, Eb, st.col.p, info_dev + slot, const_cast<HopInfo*>(info_host) + slot, chain,
                           tstate + dst, (const int64_t*)a.rel_size, rel_sizes + (int64_t)(ell + 1) * num_relations + e,
                           (e == last_rel && ell + 1 < L)
                               ? HopBoundary{tstate, num_node_types, rel_sizes, num_relations, ell + 1}
                               : HopBoundary{nullptr, 0, nullptr, 0, -1});
        PYG_HIP_CHECK(hipGetLastError());
        if (e == last_rel) boundary_done = true;
        queued.push_back({ell, e});
      }
    }
    // the engine hand-back rides behind the last hop: its position is on the device, so no host round trip is
    // needed to know how many blocks were consumed (verified against the host's bookkeeping below)
    if (rng.engine) {
      // outputs that must exist: everything the hops may have read (spec_word bounds it) + the rest of the
      // 624-array the engine is left in; only a launch that is already under way is waited for
      const int64_t need32 = (spec_word / 128 + 1) * 256 + 624;
      size_t k = 0;
      while (k < rng.marks.size() && rng.marks[k].upto32 < need32) ++k;
      if (k < rng.marks.size()) {
        MtHandBack* hb_dev;
        PYG_ALLOC(hb_dev, MtHandBack*, c, sizeof(MtHandBack));
        hand_back = reinterpret_cast<MtHandBack*>(static_cast<char*>(pinned) + hand_back_offset);
        hand_back->status = -1;
        if (k >= rng.waited) {
          PYG_HIP_CHECK(hipStreamWaitEvent(stream, rng.marks[k].ev, 0));
          rng.waited = k + 1;
        }
        {
          int rc = rng_queue_hand_back(c, rng, chain, rng.marks[k].upto32, hb_dev);
          if (rc != PYG_HIP_OK) return rc;
        }
        PYG_HIP_CHECK(hipMemcpyAsync(hand_back, hb_dev, sizeof(MtHandBack), hipMemcpyDeviceToHost, stream));
      }
    }
    pt.lap(4);
    PYG_HIP_CHECK(hipStreamSynchronize(stream));
    pt.lap(5);
    if (temporal)
      PYG_HIP_REQUIRE(*static_cast<volatile int*>(err_flag) == 0, "Found invalid non-sorted temporal neighborhood");
    for (const Queued& q : queued)
      if (info_host[q.ell * num_relations + q.e].overflow) return kNeedSlow;
    // fold the totals into the host's bookkeeping, hop by hop
    size_t qi = 0;
    for (int ell = 0; ell < L; ++ell) {
      for (int e = 0; e < num_relations; ++e) rs[(size_t)e].edges_per_hop.push_back(0);
      std::vector<int64_t> before((size_t)num_node_types);
      for (int t = 0; t < num_node_types; ++t) before[(size_t)t] = ns[(size_t)t].nodes.size;
      for (; qi < queued.size() && queued[qi].ell == ell; ++qi) {
        const int e = queued[qi].e;
        volatile HopInfo* hi = info_host + (ell * num_relations + e);
        const pyg_hip_relation& r = rels[e];
        NodeSet& dn = ns[(size_t)(!csc ? r.dst_type : r.src_type)];
        RelState& st = rs[(size_t)e];
        const int64_t E = hi->tot.edges, U = hi->uniq;
        if (E > 0) {
          const RngTab tab = hi->tot.tab;
          const int64_t end_word = rng.word + tab_dw(tab, rng.units);
          rng.blocks = std::max(rng.blocks, end_word / 128 + 1);
          rng.word = end_word;
          rng.units = tab_nb(tab, rng.units);
        }
        dn.nodes.size += U;
        if (disjoint) dn.batch.size += U;
        st.row.size += E;
        st.col.size += E;
        st.eid.size += E;
        st.edges_per_hop.back() = E;
      }
      for (int t = 0; t < num_node_types; ++t)
        nodes_per_hop[(size_t)t].push_back(ns[(size_t)t].nodes.size - before[(size_t)t]);
    }
    g_sampler_mode = "queued";
    done_fast = true;
  }
  if (!done_fast) {
  g_sampler_mode = "synchronising";

  struct Pending {
    int e = 0;
    int64_t F = 0, Eb = 0, count = 0;
    RangeCtx range;
    CountAgg* tile_buf = nullptr;
    int64_t* edge_off = nullptr;
    int64_t* rng_word = nullptr;
    int32_t* rng_units = nullptr;
    int64_t* e_node = nullptr;
    int64_t* e_batch = nullptr;
    u64* e_slot = nullptr;
    int64_t* ftile = nullptr;
  };
  std::vector<Pending> pend;
  auto free_pending = [&](Pending& q) {
    c.release(q.tile_buf);
    c.release(q.edge_off);
    c.release(q.rng_word);
    c.release(q.rng_units);
    c.release(q.e_node);
    if (q.e_batch) c.release(q.e_batch);
    c.release(q.e_slot);
    c.release(q.ftile);
  };

  // queues sample + dedup scan + finalize of one relation behind its count scan
  auto enqueue_tail = [&](Pending& q, int64_t avail_blocks) -> int {
    const pyg_hip_relation& r = rels[q.e];
    const int src = !csc ? r.src_type : r.dst_type;
    const int dst = !csc ? r.dst_type : r.src_type;
    NodeSet& sn = ns[(size_t)src];
    NodeSet& dn = ns[(size_t)dst];
    RelState& st = rs[(size_t)q.e];
    HopArgs a;
    a.info = info_dev + q.e;
    a.chain = chain;
    a.avail_blocks = avail_blocks;
    a.nodes = sn.nodes.p;
    a.batch = disjoint ? sn.batch.p : nullptr;
    a.begin = sn.slice_b;
    a.frontier = q.F;
    q.range.batch = a.batch;  // a reserve may have moved the list (src type == dst type)
    a.range = q.range;
    a.col = IdxArr(r.col, r.index_is32);
    a.count = q.count;
    a.replace = replace;
    a.num_batches = num_batches;
    a.edge_off = q.edge_off;
    a.rng_word = q.rng_word;
    a.rng_units = q.rng_units;
    a.words = rng.dev;
    if (any_biased) {  // blocks may lie behind outputs that weighted relations drew in between (WordSrc)
      a.word_b0 = rng.word >> 7;
      a.shift_old = rng.cur_shift;
      a.shift_new = rng.raw_used;
    }
    a.e_row = st.row.p + st.row.size;
    a.e_node = q.e_node;
    a.e_batch = q.e_batch;
    a.e_eid = st.eid.p + st.eid.size;
    a.e_slot = q.e_slot;
    a.table = dn.table;
    launch_sample(a, q.F, stream);
    PYG_HIP_CHECK(hipGetLastError());
    // first occurrences -> ranks -> new local ids / appended nodes
    FlagLoad fl{q.e_slot, dn.table.vals, info_dev + q.e};
    AssignStore as{q.e_slot, dn.table.vals, q.e_node, q.e_batch, dn.nodes.p,
                   disjoint ? dn.batch.p : (int64_t*)nullptr, 0, 0, 1, tstate + dst};
    int rc = device_scan<int64_t, SumOp>(fl, as, q.Eb, q.ftile, &info_dev[q.e].uniq, stream);
    if (rc != PYG_HIP_OK) return rc;
    // local ids of every emitted edge; publishes the totals and advances the device state
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((q.Eb + 255) / 256)), dim3(256), 0, stream, q.e_slot,
                       dn.table.vals, q.Eb, st.col.p + st.col.size, info_dev + q.e,
                       const_cast<HopInfo*>(info_host) + q.e, chain, tstate + dst, (const int64_t*)nullptr,
                       (int64_t*)nullptr);
    PYG_HIP_CHECK(hipGetLastError());
    return PYG_HIP_OK;
  };

  // Synchronises and commits the queued relations in order.  Returns the index (into `pend`) of the
  // first relation that must be re-queued from scratch, or -1: a relation that ran out of random words
  // (draws wider than 16 bits, or beyond what the speculation may run ahead) is repeated right here with
  // exactly the words it needs; the relations queued behind it did nothing (sticky abort) and start over.
  auto commit = [&](int* restart) -> int {
    *restart = -1;
    if (pend.empty()) return PYG_HIP_OK;
    pt.lap(4);
    PYG_HIP_CHECK(hipStreamSynchronize(stream));
    pt.lap(5);
    if (temporal)
      PYG_HIP_REQUIRE(*static_cast<volatile int*>(err_flag) == 0, "Found invalid non-sorted temporal neighborhood");
    for (size_t k = 0; k < pend.size(); ++k) {
      Pending& q = pend[k];
      const pyg_hip_relation& r = rels[q.e];
      const int dst = !csc ? r.dst_type : r.src_type;
      NodeSet& dn = ns[(size_t)dst];
      RelState& st = rs[(size_t)q.e];
      volatile HopInfo* hi = info_host + q.e;
      if (hi->overflow == 2) {  // aborted behind an earlier relation: start over
        *restart = (int)k;
        break;
      }
      RngTab tab = hi->tot.tab;
      int64_t end_word = rng.word + tab_dw(tab, rng.units);
      if (hi->overflow == 1) {
        PYG_HIP_REQUIRE(rng.engine, "sampler: random words missing");
        if (pt.on)
          fprintf(stderr, "[pyg_hip sampler] relation %d repeated after a word top-up, %zu queued behind it start over\n",
                  q.e, pend.size() - k - 1);
        int64_t avail = 0;
        int rc = rng_wait(c, rng, end_word, &avail);
        if (rc != PYG_HIP_OK) return rc;
        hipLaunchKernelGGL(clear_abort_kernel, dim3(1), dim3(1), 0, stream, chain);
        rc = enqueue_tail(q, avail);
        if (rc != PYG_HIP_OK) return rc;
        PYG_HIP_CHECK(hipStreamSynchronize(stream));
        PYG_HIP_REQUIRE(hi->overflow == 0, "sampler: random words missing after regeneration");
        if (k + 1 < pend.size()) *restart = (int)(k + 1);
      }
      const int64_t E = hi->tot.edges, U = hi->uniq;
      if (E > 0) {
        if (rng.engine) rng.blocks = std::max(rng.blocks, end_word / 128 + 1);
        if ((end_word >> 7) > (rng.word >> 7)) rng.cur_shift = rng.raw_used;  // the current block is a new one
        rng.word = end_word;
        rng.units = tab_nb(tab, rng.units);
      }
      dn.nodes.size += U;
      if (disjoint) dn.batch.size += U;
      dn.distinct += U;
      dn.entries_bound = dn.entries_bound - q.Eb + U;
      st.row.size += E;
      st.col.size += E;
      st.eid.size += E;
      st.edges_per_hop.back() = E;
      if (*restart >= 0) break;
    }
    // relations that start over: undo their bound-based reservations
    for (size_t k = (*restart < 0 ? pend.size() : (size_t)*restart); k < pend.size(); ++k) {
      const pyg_hip_relation& r = rels[pend[k].e];
      NodeSet& dn = ns[(size_t)(!csc ? r.dst_type : r.src_type)];
      dn.entries_bound -= pend[k].Eb;
    }
    for (Pending& q : pend) free_pending(q);
    for (int t = 0; t < num_node_types; ++t) {
      ns[(size_t)t].nodes.live = ns[(size_t)t].nodes.size;
      ns[(size_t)t].batch.live = ns[(size_t)t].batch.size;
    }
    pt.lap(6);
    return PYG_HIP_OK;
  };

  // One weighted relation of one hop (biased_sample, neighbor_kernel.cpp:39-56,245-285): degree scan ->
  // host (emitted edges, generator outputs drawn) -> keys + top-k per row -> the usual dedup tail.
  auto run_biased = [&](int e, int64_t F, int64_t count) -> int {
    const pyg_hip_relation& r = rels[e];
    const int src = !csc ? r.src_type : r.dst_type;
    const int dst = !csc ? r.dst_type : r.src_type;
    NodeSet& sn = ns[(size_t)src];
    NodeSet& dn = ns[(size_t)dst];
    RelState& st = rs[(size_t)e];
    const bool f64 = r.edge_weight_dtype == PYG_F64;
    const int outputs = f64 ? 2 : 1;
    const int64_t ntiles = (F + kScanTile - 1) / kScanTile;
    CountAgg* tile_buf;
    int64_t *edge_off, *raw_off;
    int32_t* flag;
    PYG_ALLOC(tile_buf, CountAgg*, c, sizeof(CountAgg) * (size_t)(ntiles + 1));
    PYG_ALLOC(edge_off, int64_t*, c, sizeof(int64_t) * (size_t)F);
    PYG_ALLOC(raw_off, int64_t*, c, sizeof(int64_t) * (size_t)F);
    PYG_ALLOC(flag, int32_t*, c, sizeof(int32_t) * (size_t)F);
    const int64_t out_base = rng.blocks * 256 + rng.raw_used;
    const bool single = replace && count == 1;  // at::multinomial's single-draw route
    int rc;
    if (single) {
      BiasedSingleCountLoad cl{sn.nodes.p, sn.slice_b, IdxArr(r.rowptr, r.index_is32)};
      CountStore cs{edge_off, raw_off, flag, out_base, 4, nullptr};
      rc = device_scan<CountAgg, CountOp>(cl, cs, F, tile_buf, &info_dev[e].tot, stream);
    } else if (replace) {
      // the word field of the scan carries the cumulative-distribution scratch offset (in weights)
      BiasedReplaceCountLoad cl{sn.nodes.p, sn.slice_b, IdxArr(r.rowptr, r.index_is32), count};
      CountStore cs{edge_off, raw_off, flag, 0, 4, nullptr};
      rc = device_scan<CountAgg, CountOp>(cl, cs, F, tile_buf, &info_dev[e].tot, stream);
    } else {
      BiasedCountLoad cl{sn.nodes.p, sn.slice_b, IdxArr(r.rowptr, r.index_is32), count, outputs};
      CountStore cs{edge_off, raw_off, flag, out_base, 4, nullptr};
      rc = device_scan<CountAgg, CountOp>(cl, cs, F, tile_buf, &info_dev[e].tot, stream);
    }
    if (rc != PYG_HIP_OK) return rc;
    PYG_HIP_CHECK(hipMemsetAsync(&info_dev[e].overflow, 0, sizeof(int32_t), stream));
    PYG_HIP_CHECK(hipMemcpyAsync(const_cast<HopInfo*>(info_host) + e, info_dev + e, sizeof(HopInfo),
                                 hipMemcpyDeviceToHost, stream));
    PYG_HIP_CHECK(hipStreamSynchronize(stream));
    const int64_t E = info_host[e].tot.edges;
    const int64_t scratch_w = (replace && !single) ? (int64_t)((info_host[e].tot.tab & ~kPureTab) >> 20) : 0;  // cumulative-distribution entries
    // generator outputs drawn by this relation: one uniform_ value per neighbour of every drawing row, one double
    // per sampled edge (with replacement, count > 1), or one double per neighbour (single draw)
    const int64_t W = (replace && !single) ? (count > 0 ? 2 * E : 0) : (int64_t)((info_host[e].tot.tab & ~kPureTab) >> 20);
    auto cleanup = [&]() {
      c.release(tile_buf);
      c.release(edge_off);
      c.release(raw_off);
      c.release(flag);
    };
    if (E == 0) {
      cleanup();
      return PYG_HIP_OK;
    }
    PYG_HIP_REQUIRE(r.num_cols < (1ll << 31) || W == 0, "sampler: biased sampling supports rows below 2^31 neighbours");
    if (W > 0) {
      rc = rng_wait32(c, rng, out_base + W, nullptr);
      if (rc != PYG_HIP_OK) return rc;
    }
    double mult = 1.0, term = 1.0;
    for (int l2 = (int)st.edges_per_hop.size(); l2 < L && r.num_neighbors_host[l2] > 0; ++l2) {
      term *= (double)r.num_neighbors_host[l2];
      mult += term;
    }
    const int64_t grow = (int64_t)std::min<double>((double)E * mult, 16.0 * 1024 * 1024);
    rc = st.row.reserve(c, st.row.size + E, st.row.size + grow);
    if (rc != PYG_HIP_OK) return rc;
    rc = st.col.reserve(c, st.col.size + E, st.col.size + grow);
    if (rc != PYG_HIP_OK) return rc;
    rc = st.eid.reserve(c, st.eid.size + E, st.eid.size + grow);
    if (rc != PYG_HIP_OK) return rc;
    dn.nodes.live = std::max(dn.nodes.live, dn.nodes.size);
    rc = dn.nodes.reserve(c, dn.nodes.live + E, dn.nodes.live + grow);
    if (rc != PYG_HIP_OK) return rc;
    if (disjoint) {
      dn.batch.live = std::max(dn.batch.live, dn.batch.size);
      rc = dn.batch.reserve(c, dn.batch.live + E, dn.batch.live + grow);
      if (rc != PYG_HIP_OK) return rc;
    }
    rc = table_reserve(c, dn, E, dn.entries_bound + grow);
    if (rc != PYG_HIP_OK) return rc;
    int64_t *e_node, *e_batch = nullptr, *ftile;
    u64* e_slot;
    PYG_ALLOC(e_node, int64_t*, c, sizeof(int64_t) * (size_t)E);
    if (disjoint) PYG_ALLOC(e_batch, int64_t*, c, sizeof(int64_t) * (size_t)E);
    PYG_ALLOC(e_slot, u64*, c, sizeof(u64) * (size_t)E);
    const int64_t etiles = (E + kScanTile - 1) / kScanTile;
    PYG_ALLOC(ftile, int64_t*, c, sizeof(int64_t) * (size_t)(etiles + 1));
    const int64_t draws = single ? 0 : (replace ? scratch_w : W / outputs);
    const size_t ksz = f64 ? 8 : 4;
    void *skey, *selkey;
    int32_t *sidx, *selidx;
    PYG_ALLOC(skey, void*, c, ksz * (size_t)std::max<int64_t>(draws, 1));
    PYG_ALLOC(sidx, int32_t*, c, 4 * (size_t)std::max<int64_t>(draws, 1));
    PYG_ALLOC(selkey, void*, c, ksz * (size_t)E);
    PYG_ALLOC(selidx, int32_t*, c, 4 * (size_t)E);

    HopArgs a;
    a.nodes = sn.nodes.p;
    a.batch = disjoint ? sn.batch.p : nullptr;
    a.begin = sn.slice_b;
    a.frontier = F;
    a.range.rowptr = IdxArr(r.rowptr, r.index_is32);
    a.range.col = IdxArr(r.col, r.index_is32);
    a.range.time = nullptr;
    a.col = IdxArr(r.col, r.index_is32);
    a.count = count;
    a.replace = 0;
    a.num_batches = num_batches;
    a.edge_off = edge_off;
    a.rng_word = raw_off;
    a.rng_units = nullptr;
    a.words = nullptr;
    a.e_row = st.row.p + st.row.size;
    a.e_node = e_node;
    a.e_batch = e_batch;
    a.e_eid = st.eid.p + st.eid.size;
    a.e_slot = e_slot;
    a.table = dn.table;
    const unsigned wg = (unsigned)((F + 3) / 4), xg = (unsigned)((F + 63) / 64);
    if (single) {
      a.replace = 1;
      if (f64)
        hipLaunchKernelGGL(biased_single_kernel<double>, dim3(wg), dim3(256), 0, stream, a, info_dev + e,
                           static_cast<const double*>(r.edge_weight), reinterpret_cast<const uint32_t*>(rng.dev));
      else
        hipLaunchKernelGGL(biased_single_kernel<float>, dim3(wg), dim3(256), 0, stream, a, info_dev + e,
                           static_cast<const float*>(r.edge_weight), reinterpret_cast<const uint32_t*>(rng.dev));
    } else if (replace) {
      a.replace = 1;
      if (f64)
        hipLaunchKernelGGL(biased_replace_kernel<double>, dim3(wg), dim3(256), 0, stream, a, info_dev + e,
                           static_cast<const double*>(r.edge_weight), reinterpret_cast<const uint32_t*>(rng.dev), out_base,
                           static_cast<double*>(skey));
      else
        hipLaunchKernelGGL(biased_replace_kernel<float>, dim3(wg), dim3(256), 0, stream, a, info_dev + e,
                           static_cast<const float*>(r.edge_weight), reinterpret_cast<const uint32_t*>(rng.dev), out_base,
                           static_cast<float*>(skey));
    } else if (f64) {
      BiasedArgs<uint64_t> b{a, info_dev + e, r.edge_weight, reinterpret_cast<const uint32_t*>(rng.dev), out_base,
                             static_cast<uint64_t*>(skey), sidx, static_cast<uint64_t*>(selkey), selidx, flag};
      hipLaunchKernelGGL(biased_sample_kernel<true>, dim3(wg), dim3(256), 0, stream, b);
      hipLaunchKernelGGL(biased_exact_kernel<true>, dim3(xg), dim3(64), 0, stream, b);
    } else {
      BiasedArgs<uint32_t> b{a, info_dev + e, r.edge_weight, reinterpret_cast<const uint32_t*>(rng.dev), out_base,
                             static_cast<uint32_t*>(skey), sidx, static_cast<uint32_t*>(selkey), selidx, flag};
      hipLaunchKernelGGL(biased_sample_kernel<false>, dim3(wg), dim3(256), 0, stream, b);
      hipLaunchKernelGGL(biased_exact_kernel<false>, dim3(xg), dim3(64), 0, stream, b);
    }
    PYG_HIP_CHECK(hipGetLastError());
    FlagLoad fl{e_slot, dn.table.vals, info_dev + e};
    AssignStore as{e_slot, dn.table.vals, e_node, e_batch, dn.nodes.p, disjoint ? dn.batch.p : (int64_t*)nullptr,
                   0, 0, 1, tstate + dst};
    rc = device_scan<int64_t, SumOp>(fl, as, E, ftile, &info_dev[e].uniq, stream);
    if (rc != PYG_HIP_OK) return rc;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, stream, e_slot,
                       dn.table.vals, E, st.col.p + st.col.size, info_dev + e, const_cast<HopInfo*>(info_host) + e,
                       chain, tstate + dst, (const int64_t*)nullptr, (int64_t*)nullptr);
    PYG_HIP_CHECK(hipGetLastError());
    PYG_HIP_CHECK(hipStreamSynchronize(stream));
    if (info_host[e].overflow == 3)  // at::multinomial's checks (negative / non-finite weights, zero sum)
      return fail(PYG_HIP_ERR_INVALID, "invalid multinomial distribution (a sampled row has negative or non-finite weights, or they sum to zero)");
    const int64_t U = info_host[e].uniq;
    rng.raw_used += W;
    dn.nodes.size += U;
    if (disjoint) dn.batch.size += U;
    dn.distinct += U;
    dn.entries_bound = dn.entries_bound - E + U;
    dn.nodes.live = dn.nodes.size;
    dn.batch.live = dn.batch.size;
    st.row.size += E;
    st.col.size += E;
    st.eid.size += E;
    st.edges_per_hop.back() = E;
    cleanup();
    c.release(e_node);
    if (e_batch) c.release(e_batch);
    c.release(e_slot);
    c.release(ftile);
    c.release(skey);
    c.release(sidx);
    c.release(selkey);
    c.release(selidx);
    return PYG_HIP_OK;
  };

  for (int ell = 0; ell < L; ++ell) {
    std::vector<int> order;
    for (int e = 0; e < num_relations; ++e) {
      const pyg_hip_relation& r = rels[e];
      const NodeSet& sn = ns[(size_t)(!csc ? r.src_type : r.dst_type)];
      rs[(size_t)e].edges_per_hop.push_back(0);
      if (sn.slice_e - sn.slice_b <= 0 || r.num_neighbors_host[ell] == 0 || r.num_cols == 0) continue;
      order.push_back(e);
    }
    int64_t spec_word = rng.word;  // upper bound of the engine position behind the queued relations
    size_t pos = 0;
    while (pos < order.size() || !pend.empty()) {
      bool flush = pos >= order.size();
      if (!flush) {
        const int e = order[pos];
        const pyg_hip_relation& r = rels[e];
        const int src = !csc ? r.src_type : r.dst_type;
        const int dst = !csc ? r.dst_type : r.src_type;
        NodeSet& sn = ns[(size_t)src];
        NodeSet& dn = ns[(size_t)dst];
        RelState& st = rs[(size_t)e];
        const int64_t count = r.num_neighbors_host[ell];
        const int64_t F = sn.slice_e - sn.slice_b;
        if (r.edge_weight) {  // biased: runs alone (any_biased => nothing is ever pending)
          int rc = run_biased(e, F, count);
          if (rc != PYG_HIP_OK) return rc;
          ++pos;
          continue;
        }
        // Unbounded / large fan-outs and the host-callback word source (which must draw exactly what is
        // consumed) need the count-scan total on the host before anything can be sized: they run alone.
        // (with weighted relations in the call the engine's block positions depend on what ran before: alone too)
        const bool presync = count < 0 || count > 64 || !rng.engine || any_biased;
        if (presync && !pend.empty()) {
          flush = true;
        } else {
          Pending q;
          q.e = e;
          q.F = F;
          q.count = count;
          // 1. per-node edge counts + RNG transition tables -> exclusive scan (total -> info_dev[e].tot)
          const int64_t ntiles = (F + kScanTile - 1) / kScanTile;
          PYG_ALLOC(q.tile_buf, CountAgg*, c, sizeof(CountAgg) * (size_t)(ntiles + 1));
          PYG_ALLOC(q.edge_off, int64_t*, c, sizeof(int64_t) * (size_t)F);
          PYG_ALLOC(q.rng_word, int64_t*, c, sizeof(int64_t) * (size_t)F);
          PYG_ALLOC(q.rng_units, int32_t*, c, sizeof(int32_t) * (size_t)F);
          // edge-level time wins over node-level time of the destination type (neighbor_kernel.cpp:746-789)
          q.range.rowptr = IdxArr(r.rowptr, r.index_is32);
          q.range.col = IdxArr(r.col, r.index_is32);
          q.range.time = r.edge_time ? r.edge_time : (node_time ? node_time[dst] : nullptr);
          q.range.edge_level = r.edge_time ? 1 : 0;
          q.range.last = temporal_last;
          q.range.seed_times = seed_times;
          q.range.batch = disjoint ? sn.batch.p : nullptr;
          q.range.error = err_flag;
          CountLoad cl{sn.nodes.p, sn.slice_b, q.range, count, replace};
          CountStore cs{q.edge_off, q.rng_word, q.rng_units, 0, 4, chain};
          int rc = device_scan<CountAgg, CountOp>(cl, cs, F, q.tile_buf, &info_dev[e].tot, stream);
          if (rc != PYG_HIP_OK) return rc;

          int64_t avail_blocks = 0;
          q.Eb = presync ? 0 : F * count;
          if (presync) {
            PYG_HIP_CHECK(hipMemcpyAsync(const_cast<HopInfo*>(info_host) + e, info_dev + e, sizeof(HopInfo),
                                         hipMemcpyDeviceToHost, stream));
            PYG_HIP_CHECK(hipStreamSynchronize(stream));
            if (q.range.time)
              PYG_HIP_REQUIRE(*static_cast<volatile int*>(err_flag) == 0,
                              "Found invalid non-sorted temporal neighborhood");
            q.Eb = info_host[e].tot.edges;
            pt.lap(1);
            if (q.Eb == 0) {
              free_pending(q);
              ++pos;
              continue;
            }
            // make the consumed random words resident (drawn by the caller's generator)
            const int64_t end_word = rng.word + tab_dw(info_host[e].tot.tab, rng.units);
            if (any_biased) {
              rc = rng_wait32(c, rng, (end_word / 128 + 1) * 256 + rng.raw_used, nullptr);
              avail_blocks = end_word / 128 + 1;  // exactly the words this relation reads
            } else if (rng.engine) {
              rc = rng_wait(c, rng, end_word, &avail_blocks);
            } else {
              rc = rng_ensure(c, rng, end_word);
            }
            if (rc != PYG_HIP_OK) return rc;
            if (!rng.engine) avail_blocks = rng.blocks;
          } else {
            // run-ahead: the words this relation MAY read (16-bit draws) are already being generated
            spec_word += (F * count + 3) / 4 + 1;
            rc = rng_wait(c, rng, spec_word, &avail_blocks);
            if (rc != PYG_HIP_OK) return rc;
          }
          pt.lap(2);

          // 2. size outputs / node list / hash table for the bound Eb.  Expected growth over the remaining
          // hops (this relation's own fan-outs, every node expanding fully) sizes them once instead of
          // copying / rehashing them per hop.
          double mult = 1.0, term = 1.0;
          for (int l2 = ell + 1; l2 < L && r.num_neighbors_host[l2] > 0; ++l2) {
            term *= (double)r.num_neighbors_host[l2];
            mult += term;
          }
          const int64_t grow = (int64_t)std::min<double>((double)q.Eb * mult, 16.0 * 1024 * 1024);
          rc = st.row.reserve(c, st.row.size + q.Eb, st.row.size + grow);
          if (rc != PYG_HIP_OK) return rc;
          rc = st.col.reserve(c, st.col.size + q.Eb, st.col.size + grow);
          if (rc != PYG_HIP_OK) return rc;
          // edge ids are always produced: they double as the chosen-set history of large fan-outs
          rc = st.eid.reserve(c, st.eid.size + q.Eb, st.eid.size + grow);
          if (rc != PYG_HIP_OK) return rc;
          // the node list may already hold uncommitted nodes of relations queued before this one
          dn.nodes.live = std::max(dn.nodes.live, dn.nodes.size);
          rc = dn.nodes.reserve(c, dn.nodes.live + q.Eb, dn.nodes.live + grow);
          if (rc != PYG_HIP_OK) return rc;
          dn.nodes.live += q.Eb;
          if (disjoint) {
            dn.batch.live = std::max(dn.batch.live, dn.batch.size);
            rc = dn.batch.reserve(c, dn.batch.live + q.Eb, dn.batch.live + grow);
            if (rc != PYG_HIP_OK) return rc;
            dn.batch.live += q.Eb;
          }
          rc = table_reserve(c, dn, q.Eb, dn.entries_bound + grow);
          if (rc != PYG_HIP_OK) return rc;
          PYG_ALLOC(q.e_node, int64_t*, c, sizeof(int64_t) * (size_t)q.Eb);
          if (disjoint) PYG_ALLOC(q.e_batch, int64_t*, c, sizeof(int64_t) * (size_t)q.Eb);
          PYG_ALLOC(q.e_slot, u64*, c, sizeof(u64) * (size_t)q.Eb);
          const int64_t etiles = (q.Eb + kScanTile - 1) / kScanTile;
          PYG_ALLOC(q.ftile, int64_t*, c, sizeof(int64_t) * (size_t)(etiles + 1));
          pt.lap(3);
          // 3. sample + insert, dedup scan, finalize
          rc = enqueue_tail(q, avail_blocks);
          if (rc != PYG_HIP_OK) return rc;
          pend.push_back(q);
          ++pos;
          if (presync) flush = true;
        }
      }
      if (flush) {
        int restart = -1;
        const size_t first_pending = pos - pend.size();
        int rc = commit(&restart);
        if (rc != PYG_HIP_OK) return rc;
        if (restart >= 0) pos = first_pending + (size_t)restart;
        pend.clear();
        spec_word = rng.word;
      }
    }
    for (int t = 0; t < num_node_types; ++t) {
      NodeSet& n = ns[(size_t)t];
      n.slice_b = n.slice_e;
      n.slice_e = n.nodes.size;
      nodes_per_hop[(size_t)t].push_back(n.slice_e - n.slice_b);
    }
  }

  }  // synchronising mode

  // ---- hand the results over ----
  for (int t = 0; t < num_node_types; ++t) {
    NodeSet& n = ns[(size_t)t];
    res->num_nodes[t] = n.nodes.size;
    for (int l = 0; l <= L; ++l) res->nodes_per_hop_host[(size_t)t * (L + 1) + l] = nodes_per_hop[(size_t)t][(size_t)l];
    if (!disjoint) {
      if (!n.nodes.p) PYG_ALLOC(n.nodes.p, int64_t*, c, 16);
      res->node_id[t] = n.nodes.p;
      c.keep(n.nodes.p);
    } else {
      int64_t* out;
      PYG_ALLOC(out, int64_t*, c, sizeof(int64_t) * 2 * (size_t)std::max<int64_t>(n.nodes.size, 1));
      if (n.nodes.size > 0) {
        hipLaunchKernelGGL(interleave_kernel, dim3((unsigned)((n.nodes.size + 255) / 256)), dim3(256),
                           0, stream, n.batch.p, n.nodes.p, n.nodes.size, out);
        PYG_HIP_CHECK(hipGetLastError());
      }
      res->node_id[t] = out;
      c.keep(out);
    }
  }
  for (int e = 0; e < num_relations; ++e) {
    RelState& st = rs[(size_t)e];
    if (!st.row.p) PYG_ALLOC(st.row.p, int64_t*, c, 16);
    if (!st.col.p) PYG_ALLOC(st.col.p, int64_t*, c, 16);
    if (!st.eid.p) PYG_ALLOC(st.eid.p, int64_t*, c, 16);
    // csc only swaps the returned (row, col) (get_sampled_edges, neighbor_kernel.cpp:155-159)
    res->row[e] = !csc ? st.row.p : st.col.p;
    res->col[e] = !csc ? st.col.p : st.row.p;
    c.keep(st.row.p);
    c.keep(st.col.p);
    res->num_edges[e] = st.row.size;
    if (return_edge_id) {
      res->edge_id[e] = st.eid.p;
      c.keep(st.eid.p);
    } else {
      res->edge_id[e] = nullptr;
    }
    for (int l = 0; l < L; ++l) res->edges_per_hop_host[(size_t)e * L + l] = st.edges_per_hop[(size_t)l];
  }
  res->rng_blocks = rng.blocks;
  if (hand_back && hand_back->status == 0 && hand_back->n32 == rng.blocks * 256 + rng.raw_used) {
    // already on the host (it arrived with the hop totals): nothing left to wait for
    ::memcpy(c.host->mt19937, &hand_back->st, sizeof(MtDev));
    // the words generated beyond this call's consumption are the next call's words if it presents this very engine
    // (sampler_rng.h, RngCarry): kept, with everything still queued on the side stream completed instead of cancelled
    if (rng_late) rng_carry_commit(c, rng, hand_back->st, hand_back->n32);
    c.quiesce_side();
  } else {
    c.main_idle = false;          // (the hand-back below is queued on the main stream: the closing synchronisation stays)
    int rc = rng_finish(c, rng);  // hand the advanced engine back
    if (rc != PYG_HIP_OK) return rc;
  }
  // (fused chain, nothing queued behind its closing launch: that launch was seen to finish -- run_fused_chain)
  if (!(c.main_idle && !disjoint)) PYG_HIP_CHECK(hipStreamSynchronize(stream));
  pt.lap(7);
  return PYG_HIP_OK;
}

__global__ void iota_kernel(int64_t* out, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = i;
}

__global__ void dist_nodes_kernel(const int64_t* __restrict__ seed, int64_t S, const int64_t* __restrict__ e_node,
                                  const int64_t* __restrict__ e_batch, int64_t E, int disjoint,
                                  int64_t* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= S + E) return;
  const int64_t node = i < S ? seed[i] : e_node[i - S];
  if (disjoint) {
    out[2 * i] = i < S ? i : e_batch[i - S];
    out[2 * i + 1] = node;
  } else {
    out[i] = node;
  }
}

// dist_neighbor_sample (neighbor_kernel.cpp:957-978): one hop over the seeds, no relabelling.
int run_dist_sampler(const IdxArr rowptr, const IdxArr col, const int64_t* seed, int64_t S, int64_t count,
                     const int64_t* node_time, const int64_t* edge_time, const int64_t* seed_time,
                     const void* weight, int weight_dtype, int temporal_last, int replace, int disjoint, Ctx& c,
                     int64_t** out_node, int64_t** out_edge, int64_t* num_edges, int64_t* cumsum_host) {
  hipStream_t stream = c.stream;
  RngHost rng;
  void* pinned = nullptr;
  {
    int rc = get_pinned(&pinned, 4096);
    if (rc != PYG_HIP_OK) return rc;
  }
  const bool temporal = node_time || edge_time;
  if (temporal) PYG_HIP_REQUIRE(disjoint, "Temporal sampling needs to create disjoint subgraphs");
  if (edge_time) PYG_HIP_REQUIRE(seed_time != nullptr, "Seed time needs to be specified");
  if (weight) {  // biased_sample in distributed mode (neighbor_kernel.cpp:436-447, 296-303)
    PYG_HIP_REQUIRE(!temporal, "Biased temporal sampling not yet supported");
    PYG_HIP_REQUIRE(weight_dtype == PYG_F32 || weight_dtype == PYG_F64, "sampler: edge_weight must be float32 or float64");
    if (!c.host->mt19937) return fail(PYG_HIP_ERR_UNSUPPORTED, "sampler: biased sampling needs the mt19937 engine state (host->mt19937)");
  }
  const bool single = weight && replace && count == 1;  // at::multinomial's single-draw route
  if (c.host->mt19937) {
    std::vector<int64_t> spec;
    if (count > 0 && !weight) spec.push_back(std::min<int64_t>(kSpecCapWords, 256 + (int64_t)((double)S * (double)count / 4.0)));
    int rc = rng_begin(c, rng, pinned, spec);
    if (rc != PYG_HIP_OK) return rc;
  } else {
    int rc = rng_ensure(c, rng, 0);
    if (rc != PYG_HIP_OK) return rc;
  }
  cumsum_host[0] = S;
  int64_t E = 0;
  int64_t* e_node = nullptr;
  int64_t* e_batch = nullptr;
  int64_t* e_eid = nullptr;
  if (S > 0) {
    int64_t* batch = nullptr;
    int64_t* seed_times = nullptr;
    int* err_flag = nullptr;
    if (disjoint) {
      PYG_ALLOC(batch, int64_t*, c, sizeof(int64_t) * (size_t)S);
      hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, stream, batch, S);
    }
    if (temporal) {
      PYG_ALLOC(seed_times, int64_t*, c, sizeof(int64_t) * (size_t)S);
      PYG_ALLOC(err_flag, int*, c, sizeof(int));
      PYG_HIP_CHECK(hipMemsetAsync(err_flag, 0, sizeof(int), stream));
      hipLaunchKernelGGL(seed_time_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, stream, seed,
                         seed_time, node_time, S, (int64_t)0, seed_times);
    }
    PYG_HIP_CHECK(hipGetLastError());
    RangeCtx range;
    range.rowptr = rowptr;
    range.col = col;
    range.time = edge_time ? edge_time : node_time;
    range.edge_level = edge_time ? 1 : 0;
    range.last = temporal_last;
    range.seed_times = seed_times;
    range.batch = batch;
    range.error = err_flag;
    const int64_t ntiles = (S + kScanTile - 1) / kScanTile;
    CountAgg* tile_buf;
    int64_t* edge_off;
    int64_t* rng_word;
    int32_t* rng_units;
    PYG_ALLOC(tile_buf, CountAgg*, c, sizeof(CountAgg) * (size_t)(ntiles + 1));
    PYG_ALLOC(edge_off, int64_t*, c, sizeof(int64_t) * (size_t)S);
    PYG_ALLOC(rng_word, int64_t*, c, sizeof(int64_t) * (size_t)S);
    PYG_ALLOC(rng_units, int32_t*, c, sizeof(int32_t) * (size_t)S);
    const bool f64 = weight_dtype == PYG_F64;
    const int64_t out_base = rng.blocks * 256;  // the uniform_ draws follow the engine's first block
    int rc;
    if (single) {
      BiasedSingleCountLoad cl{seed, 0, rowptr};
      CountStore cs{edge_off, rng_word, rng_units, out_base, 4, nullptr};
      rc = device_scan<CountAgg, CountOp>(cl, cs, S, tile_buf, tile_buf + ntiles, stream);
    } else if (weight && replace) {
      BiasedReplaceCountLoad cl{seed, 0, rowptr, count};
      CountStore cs{edge_off, rng_word, rng_units, 0, 4, nullptr};
      rc = device_scan<CountAgg, CountOp>(cl, cs, S, tile_buf, tile_buf + ntiles, stream);
    } else if (weight) {
      BiasedCountLoad cl{seed, 0, rowptr, count, f64 ? 2 : 1};
      CountStore cs{edge_off, rng_word, rng_units, out_base, 4, nullptr};
      rc = device_scan<CountAgg, CountOp>(cl, cs, S, tile_buf, tile_buf + ntiles, stream);
    } else {
      CountLoad cl{seed, 0, range, count, replace};
      CountStore cs{edge_off, rng_word, rng_units, rng.word, rng.units};
      rc = device_scan<CountAgg, CountOp>(cl, cs, S, tile_buf, tile_buf + ntiles, stream);
    }
    if (rc != PYG_HIP_OK) return rc;
    PYG_HIP_CHECK(hipMemcpyAsync(pinned, tile_buf + ntiles, sizeof(CountAgg), hipMemcpyDeviceToHost, stream));
    // per-seed prefix of emitted neighbours -> cumsum_neighbors_per_node (:386-388,446-447)
    PYG_HIP_CHECK(hipMemcpyAsync(cumsum_host + 1, edge_off, sizeof(int64_t) * (size_t)S, hipMemcpyDeviceToHost,
                                 stream));
    int herr = 0;
    if (temporal) PYG_HIP_CHECK(hipMemcpyAsync(&herr, err_flag, sizeof(int), hipMemcpyDeviceToHost, stream));
    PYG_HIP_CHECK(hipStreamSynchronize(stream));
    PYG_HIP_REQUIRE(herr == 0, "Found invalid non-sorted temporal neighborhood");
    const CountAgg tot = *static_cast<CountAgg*>(pinned);
    E = tot.edges;
    // cumsum_host[1 + i] currently holds the EXCLUSIVE prefix of seed i; shift to "size after seed i"
    for (int64_t i = 0; i + 1 < S; ++i) cumsum_host[1 + i] = S + cumsum_host[2 + i];
    cumsum_host[S] = S + E;
    if (E > 0) {
      // generator outputs drawn directly: uniform_ values per neighbour, or one double per sampled edge
      const int64_t scratch_w = (weight && replace && !single) ? (int64_t)((tot.tab & ~kPureTab) >> 20) : 0;
      const int64_t W = !weight ? 0 : (replace && !single) ? (count > 0 ? 2 * E : 0) : (int64_t)((tot.tab & ~kPureTab) >> 20);
      if (weight) {
        if (W > 0) rc = rng_wait32(c, rng, out_base + W, nullptr);
      } else {
        const int64_t end_word = rng.word + tab_dw(tot.tab, rng.units);
        rc = rng_ensure(c, rng, end_word);
      }
      if (rc != PYG_HIP_OK) return rc;
      PYG_ALLOC(e_node, int64_t*, c, sizeof(int64_t) * (size_t)E);
      PYG_ALLOC(e_eid, int64_t*, c, sizeof(int64_t) * (size_t)E);
      int64_t* e_row;
      PYG_ALLOC(e_row, int64_t*, c, sizeof(int64_t) * (size_t)E);
      if (disjoint) PYG_ALLOC(e_batch, int64_t*, c, sizeof(int64_t) * (size_t)E);
      HopArgs a;
      a.nodes = seed;
      a.batch = batch;
      a.begin = 0;
      a.frontier = S;
      a.range = range;
      a.col = col;
      a.count = count;
      a.replace = replace;
      a.num_batches = 1;
      a.edge_off = edge_off;
      a.rng_word = rng_word;
      a.rng_units = rng_units;
      a.words = rng.dev;
      a.e_row = e_row;
      a.e_node = e_node;
      a.e_batch = e_batch;
      a.e_eid = e_eid;
      a.e_slot = nullptr;
      a.table = HashTable{nullptr, nullptr, 0, 0};
      if (!weight) {
        launch_sample(a, S, stream);
      } else if (replace) {
        HopInfo* info;
        void* cum;
        PYG_ALLOC(info, HopInfo*, c, sizeof(HopInfo));
        PYG_ALLOC(cum, void*, c, (f64 ? 8 : 4) * (size_t)std::max<int64_t>(scratch_w, 1));
        PYG_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(HopInfo), stream));
        a.replace = 1;
        const unsigned wg = (unsigned)((S + 3) / 4);
        if (single && f64)
          hipLaunchKernelGGL(biased_single_kernel<double>, dim3(wg), dim3(256), 0, stream, a, info,
                             static_cast<const double*>(weight), reinterpret_cast<const uint32_t*>(rng.dev));
        else if (single)
          hipLaunchKernelGGL(biased_single_kernel<float>, dim3(wg), dim3(256), 0, stream, a, info,
                             static_cast<const float*>(weight), reinterpret_cast<const uint32_t*>(rng.dev));
        else if (f64)
          hipLaunchKernelGGL(biased_replace_kernel<double>, dim3(wg), dim3(256), 0, stream, a, info,
                             static_cast<const double*>(weight), reinterpret_cast<const uint32_t*>(rng.dev), out_base,
                             static_cast<double*>(cum));
        else
          hipLaunchKernelGGL(biased_replace_kernel<float>, dim3(wg), dim3(256), 0, stream, a, info,
                             static_cast<const float*>(weight), reinterpret_cast<const uint32_t*>(rng.dev), out_base,
                             static_cast<float*>(cum));
        PYG_HIP_CHECK(hipGetLastError());
        PYG_HIP_CHECK(hipMemcpyAsync(pinned, info, sizeof(HopInfo), hipMemcpyDeviceToHost, stream));
        PYG_HIP_CHECK(hipStreamSynchronize(stream));
        if (static_cast<HopInfo*>(pinned)->overflow == 3)
          return fail(PYG_HIP_ERR_INVALID, "invalid multinomial distribution (a sampled row has negative or non-finite weights, or they sum to zero)");
        rng.raw_used += W;
      } else {
        const int64_t draws = W / (f64 ? 2 : 1);
        const size_t ksz = f64 ? 8 : 4;
        void *skey, *selkey;
        int32_t *sidx, *selidx;
        HopInfo* info;
        PYG_ALLOC(skey, void*, c, ksz * (size_t)std::max<int64_t>(draws, 1));
        PYG_ALLOC(sidx, int32_t*, c, 4 * (size_t)std::max<int64_t>(draws, 1));
        PYG_ALLOC(selkey, void*, c, ksz * (size_t)E);
        PYG_ALLOC(selidx, int32_t*, c, 4 * (size_t)E);
        PYG_ALLOC(info, HopInfo*, c, sizeof(HopInfo));
        const unsigned wg = (unsigned)((S + 3) / 4), xg = (unsigned)((S + 63) / 64);
        if (f64) {
          BiasedArgs<uint64_t> b{a, info, weight, reinterpret_cast<const uint32_t*>(rng.dev), out_base,
                                 static_cast<uint64_t*>(skey), sidx, static_cast<uint64_t*>(selkey), selidx, rng_units};
          hipLaunchKernelGGL(biased_sample_kernel<true>, dim3(wg), dim3(256), 0, stream, b);
          hipLaunchKernelGGL(biased_exact_kernel<true>, dim3(xg), dim3(64), 0, stream, b);
        } else {
          BiasedArgs<uint32_t> b{a, info, weight, reinterpret_cast<const uint32_t*>(rng.dev), out_base,
                                 static_cast<uint32_t*>(skey), sidx, static_cast<uint32_t*>(selkey), selidx, rng_units};
          hipLaunchKernelGGL(biased_sample_kernel<false>, dim3(wg), dim3(256), 0, stream, b);
          hipLaunchKernelGGL(biased_exact_kernel<false>, dim3(xg), dim3(64), 0, stream, b);
        }
        rng.raw_used += W;
      }
      PYG_HIP_CHECK(hipGetLastError());
    }
  }
  int64_t* nodes_out;
  PYG_ALLOC(nodes_out, int64_t*, c, sizeof(int64_t) * (size_t)std::max<int64_t>((S + E) * (disjoint ? 2 : 1), 1));
  if (S + E > 0) {
    hipLaunchKernelGGL(dist_nodes_kernel, dim3((unsigned)((S + E + 255) / 256)), dim3(256), 0, stream, seed, S,
                       e_node, e_batch, E, disjoint, nodes_out);
    PYG_HIP_CHECK(hipGetLastError());
  }
  if (!e_eid) PYG_ALLOC(e_eid, int64_t*, c, 16);
  *out_node = nodes_out;
  *out_edge = e_eid;
  *num_edges = E;
  c.keep(nodes_out);
  c.keep(e_eid);
  {
    int rc = rng_finish(c, rng);
    if (rc != PYG_HIP_OK) return rc;
  }
  PYG_HIP_CHECK(hipStreamSynchronize(stream));
  return PYG_HIP_OK;
}

// ---- distributed-sampling helpers ---------------------------------------------------------------------
// relabel_neighborhood (sampler/cpu/dist_relabel_kernel.cpp:30-94): the Mapper's insertion-ordered ids for
// seeds + an externally sampled node sequence -- the sampler's own first-occurrence machinery without
// sampling.  `emit` of position j: hash insert with atomicMin(j).
__global__ void relabel_insert_kernel(const int64_t* __restrict__ nodes, const int64_t* __restrict__ batch, int64_t n,
                                      int64_t num_batches, HashTable t, u64* __restrict__ slots) {
  const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= n) return;
  const u64 s = table_slot(t, make_key(nodes[j], batch ? batch[j] : 0, num_batches));
  slots[j] = s;
  __hip_atomic_fetch_min(&t.vals[s], kProvisional + (u64)j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// row[j] = the source node whose neighbour list contains position j: prefix[i] <= j < prefix[i + 1]
__global__ void expand_rows_kernel(const int64_t* __restrict__ prefix, int64_t n, int64_t total, int64_t* __restrict__ row) {
  const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= total) return;
  int64_t lo = 0, hi = n;  // first i with prefix[i + 1] > j
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (prefix[mid + 1] > j) hi = mid; else lo = mid + 1;
  }
  row[j] = lo;
}

// merge_sampler_outputs (sampler/cpu/dist_merge_outputs_kernel.cpp:17-138): out = the segments
// bases[part[j]][begin[j] .. begin[j] + len_j) concatenated in j order (dst_off = prefix of the lengths);
// with `fill`, out[i] = fill[j] instead (the batch vector of disjoint sampling).
__global__ void segment_concat_kernel(const int64_t* const* __restrict__ bases, const int64_t* __restrict__ part,
                                      const int64_t* __restrict__ begin, const int64_t* __restrict__ dst_off, int64_t n,
                                      const int64_t* __restrict__ fill, int64_t* __restrict__ out, int64_t total) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  int64_t lo = 0, hi = n;  // first j with dst_off[j + 1] > i
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (dst_off[mid + 1] > i) hi = mid; else lo = mid + 1;
  }
  out[i] = fill ? fill[lo] : bases[part[lo]][begin[lo] + (i - dst_off[lo])];
}

inline size_t relabel_ws_bytes(int64_t S, int64_t E) {
  u64 cap = 1024;
  while (cap < 2 * (u64)(S + E)) cap <<= 1;
  const int64_t tiles = (std::max(S, E) + kScanTile - 1) / kScanTile + 2;
  return align_up(sizeof(u64) * 2 * cap, 256) + align_up(sizeof(u64) * (size_t)(3 * S + E + 1), 256) +
         align_up(sizeof(int64_t) * (size_t)tiles, 256) + align_up(sizeof(TypeState), 256);
}

}  // namespace
}  // namespace pyg_hip

using namespace pyg_hip;

extern "C" const char* pyg_hip_sampler_last_mode(void) { return g_sampler_mode; }

extern "C" int pyg_hip_sampler_table_cache(int64_t limit) {
  g_epoch_limit.store(limit > 0 && (u64)limit < kEpochMax ? (u64)limit : kEpochMax);
  int dev = 0, n = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(dense_cache_mutex());
  for (const DenseCacheEntry& e : dense_cache()) n += e.device == dev ? 1 : 0;
  return n;
}

extern "C" int pyg_hip_sampler_table_cache_release(const pyg_hip_sampler_host* host) {
  PYG_HIP_REQUIRE(host && host->free, "sampler: table_cache_release needs the allocator the tables came from");
  int dev = 0;
  PYG_HIP_CHECK(hipGetDevice(&dev));
  std::vector<u64*> drop;
  int kept = 0;
  {
    std::lock_guard<std::mutex> lock(dense_cache_mutex());
    auto& v = dense_cache();
    for (size_t i = 0; i < v.size();) {
      if (v[i].device == dev && !v[i].busy) {
        drop.push_back(v[i].ptr);
        v.erase(v.begin() + (long)i);
      } else {
        kept += v[i].device == dev ? 1 : 0;
        ++i;
      }
    }
  }
  // (an idle entry's last user synchronised its stream before it handed the table back: nothing on the device uses it)
  for (u64* p : drop) host->free(host->user, p);
  kept += rng_carry_release_idle(host);   // the random-word stream kept between calls (sampler_rng.h, RngCarry)
  return kept;
}

extern "C" int pyg_hip_sampler_rng_carry_stats(int64_t* adopted, int64_t* cold) {
  PYG_HIP_REQUIRE(adopted && cold, "sampler: rng_carry_stats needs two counters");
  long long a = 0, k = 0;
  rng_carry_stats(&a, &k);
  *adopted = a;
  *cold = k;
  return PYG_HIP_OK;
}

static thread_local bool g_lane_call = false;   // set around the per-batch calls of pyg_hip_hetero_neighbor_sample_batched

extern "C" int pyg_hip_hetero_neighbor_sample(int num_node_types, int num_relations,
                                              const pyg_hip_relation* relations, int num_seed_sets,
                                              const pyg_hip_seed_set* seeds,
                                              const int64_t* const* node_time, int temporal_last,
                                              int L, int csc, int replace, int disjoint,
                                              int return_edge_id, const pyg_hip_sampler_host* host,
                                              pyg_hip_sample_result* result, void* stream_) {
  PYG_HIP_REQUIRE(num_node_types > 0 && num_relations >= 0 && num_seed_sets >= 0 && L >= 0,
                  "sampler: bad sizes");
  PYG_HIP_REQUIRE(host && host->alloc && host->free && (host->rng_blocks || host->mt19937),
                  "sampler: host callbacks missing");
  PYG_HIP_REQUIRE(result && result->node_id && result->num_nodes && result->nodes_per_hop_host &&
                      (num_relations == 0 ||
                       (result->row && result->col && result->edge_id && result->num_edges &&
                        result->edges_per_hop_host)),
                  "sampler: result arrays missing");
  for (int e = 0; e < num_relations; ++e) {
    const pyg_hip_relation& r = relations[e];
    PYG_HIP_REQUIRE(r.src_type >= 0 && r.src_type < num_node_types && r.dst_type >= 0 &&
                        r.dst_type < num_node_types,
                    "sampler: relation %d names an unknown node type", e);
    PYG_HIP_REQUIRE(r.rowptr && (r.col || r.num_cols == 0) && r.num_neighbors_host,
                    "sampler: relation %d has NULL arrays", e);
  }
  Ctx c;
  c.host = host;
  c.stream = static_cast<hipStream_t>(stream_);
  c.no_carry = g_lane_call;   // (a batched lane: every batch brings a fresh engine -- nothing to continue, nothing to keep)
  const bool allow_fast = getenv("PYG_HIP_SAMPLER_SYNC_MODE") == nullptr;  // experiment / test knob
  int rc = run_sampler(num_node_types, num_relations, relations, num_seed_sets, seeds, node_time,
                       temporal_last, L, csc, replace, disjoint, return_edge_id, c, result, allow_fast);
  if (rc == kNeedQueued) {
    // the fused chain met a sampled row of degree >= 2^16 (draws wider than 16 bits): nothing was handed out and the
    // caller's engine is untouched -- start over through round 2's chain, which carries the general RNG tables
    c.quiesce_side();
    (void)hipStreamSynchronize(c.stream);
    c.release_all();
    rc = run_sampler(num_node_types, num_relations, relations, num_seed_sets, seeds, node_time, temporal_last, L,
                     csc, replace, disjoint, return_edge_id, c, result, allow_fast, false);
  }
  if (rc == kNeedSlow) {
    // a hub row made its draws wider than the speculation assumed: nothing was handed out and the
    // caller's engine is untouched -- start over in the synchronising mode
    c.quiesce_side();
    (void)hipStreamSynchronize(c.stream);
    c.release_all();
    rc = run_sampler(num_node_types, num_relations, relations, num_seed_sets, seeds, node_time, temporal_last, L,
                     csc, replace, disjoint, return_edge_id, c, result, false);
  }
  c.quiesce_side();
  if (rc != PYG_HIP_OK) {
    (void)hipStreamSynchronize(c.stream);
  }
  c.release_all();  // scratch (and, on failure, everything)
  return rc;
}

#ifdef PYG_HIP_FOLD_TIMING
extern "C" __attribute__((visibility("default"))) int pyg_hip_debug_fold_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fold_stamps), sizeof(unsigned long long) * 16);
}
extern "C" __attribute__((visibility("default"))) int pyg_hip_debug_kstamps(unsigned long long* out) {
  unsigned n = 0;
  (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_kstamp_n), sizeof(unsigned));
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kstamps), sizeof(unsigned long long) * 128);
  unsigned z = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_kstamp_n), &z, sizeof(unsigned));
  return (int)n;
}
#endif

// ---- K independent calls at once ------------------------------------------------------------------------------------
// A mini-batch is a chain of ~12 dependent launches of 10 - 45 us, most of them far too small for 256 CUs, plus ~50 us of
// host work to queue them: one call after the other leaves the chip idle most of the time.  Independent batches (each its
// own seeds, its own generator stream -- the reference's benchmark loop reseeds per batch, benchmark/sampler/neighbor.py)
// have no such dependence: batches that were given different streams are driven by different host threads of a small
// persistent pool, so their chains overlap on the device and their host work overlaps on the cores.  Every batch runs the
// unchanged single-batch code: its results are bit for bit those of pyg_hip_hetero_neighbor_sample on that batch alone.
namespace {

class SamplePool {
 public:
  static SamplePool& get() {
    static SamplePool* p = new SamplePool();  // (never destroyed: its threads may outlive static destruction order)
    return *p;
  }
  // runs task(i) for i in [0, n) -- the caller's thread takes part -- and returns when all are done
  void run(int n, const std::function<void(int)>& task) {
    std::unique_lock<std::mutex> lock(mu_);
    while (busy_) idle_cv_.wait(lock);  // one batched call at a time
    busy_ = true;
    grow(n - 1);
    task_ = &task;
    n_ = n;
    next_ = 0;
    pending_ = n;
    ++generation_;
    lock.unlock();
    work_cv_.notify_all();
    drain();
    lock.lock();
    while (pending_ > 0) done_cv_.wait(lock);
    task_ = nullptr;
    busy_ = false;
    lock.unlock();
    idle_cv_.notify_one();
  }

 private:
  void grow(int want) {  // mu_ held
    while ((int)threads_.size() < want && threads_.size() < 15) {
      threads_.emplace_back([this] { loop(); });
      threads_.back().detach();
    }
  }
  void drain() {
    while (true) {
      int i;
      {
        std::lock_guard<std::mutex> lock(mu_);
        if (task_ == nullptr || next_ >= n_) return;
        i = next_++;
      }
      (*task_)(i);
      {
        std::lock_guard<std::mutex> lock(mu_);
        if (--pending_ == 0) done_cv_.notify_all();
      }
    }
  }
  void loop() {
    uint64_t seen = 0;
    while (true) {
      {
        std::unique_lock<std::mutex> lock(mu_);
        while (generation_ == seen || task_ == nullptr || next_ >= n_) {
          seen = generation_;
          work_cv_.wait(lock);
        }
        seen = generation_;
      }
      drain();
    }
  }
  std::mutex mu_;
  std::condition_variable work_cv_, done_cv_, idle_cv_;
  std::vector<std::thread> threads_;
  const std::function<void(int)>* task_ = nullptr;
  int n_ = 0, next_ = 0, pending_ = 0;
  uint64_t generation_ = 0;
  bool busy_ = false;
};

}  // namespace

extern "C" int pyg_hip_hetero_neighbor_sample_batched(int num_node_types, int num_relations,
                                                      const pyg_hip_relation* relations,
                                                      const int64_t* const* node_time, int temporal_last, int L, int csc,
                                                      int replace, int disjoint, int return_edge_id, int num_batches,
                                                      pyg_hip_sample_batch* batches, void* stream_) {
  PYG_HIP_REQUIRE(num_batches >= 0 && (num_batches == 0 || batches != nullptr), "sampler (batched): bad batch list");
  if (num_batches == 0) return PYG_HIP_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int dev = 0;
  PYG_HIP_CHECK(hipGetDevice(&dev));
  // lanes = batches that share a stream, in call order; one host thread per lane
  std::vector<std::vector<int>> lanes;
  {
    std::vector<void*> lane_stream;
    for (int b = 0; b < num_batches; ++b) {
      batches[b].status = PYG_HIP_OK;
      batches[b].error[0] = 0;
      batches[b].mode = "none";
      PYG_HIP_REQUIRE(batches[b].host && batches[b].result && (batches[b].num_seed_sets == 0 || batches[b].seeds_host),
                      "sampler (batched): batch %d is incomplete", b);
      size_t k = 0;
      while (k < lane_stream.size() && lane_stream[k] != batches[b].stream) ++k;
      if (k == lane_stream.size()) {
        lane_stream.push_back(batches[b].stream);
        lanes.emplace_back();
      }
      lanes[k].push_back(b);
    }
    // what the caller queued on `stream` (seeds, graph updates) is ordered in front of every lane
    hipEvent_t ev = nullptr;
    bool need = false;
    for (void* ls : lane_stream) need = need || ls != stream_;
    if (need) {
      PYG_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      PYG_HIP_CHECK(hipEventRecord(ev, stream));
      for (void* ls : lane_stream)
        if (ls != stream_) PYG_HIP_CHECK(hipStreamWaitEvent(static_cast<hipStream_t>(ls), ev, 0));
      PYG_HIP_CHECK(hipEventDestroy(ev));
    }
  }
  const std::function<void(int)> task = [&](int lane) {
    (void)hipSetDevice(dev);
    g_lane_call = true;
    for (int b : lanes[(size_t)lane]) {
      pyg_hip_sample_batch& B = batches[b];
      B.status = pyg_hip_hetero_neighbor_sample(num_node_types, num_relations, relations, B.num_seed_sets, B.seeds_host,
                                                node_time, temporal_last, L, csc, replace, disjoint, return_edge_id, B.host,
                                                B.result, B.stream);
      B.mode = g_sampler_mode;
      if (B.status != PYG_HIP_OK) snprintf(B.error, sizeof(B.error), "%s", last_error_buffer());
    }
    g_lane_call = false;
  };
  SamplePool::get().run((int)lanes.size(), task);
  for (int b = 0; b < num_batches; ++b)
    if (batches[b].status != PYG_HIP_OK) return fail(batches[b].status, "sampler (batched): batch %d: %s", b, batches[b].error);
  return PYG_HIP_OK;
}

extern "C" int pyg_hip_biased_log_f32(const float* in, float* out, int64_t n, void* stream) {
  using namespace pyg_hip;
  if (n <= 0) return PYG_HIP_OK;
  hipLaunchKernelGGL(biased_log_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     in, out, n);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

extern "C" int pyg_hip_dist_neighbor_sample(const int64_t* rowptr, const int64_t* col, const int64_t* seed,
                                            int64_t num_seed, int64_t num_neighbors, const int64_t* node_time,
                                            const int64_t* edge_time, const int64_t* seed_time,
                                            const void* edge_weight, int edge_weight_dtype, int temporal_last,
                                            int replace, int disjoint, int index_is32,
                                            const pyg_hip_sampler_host* host, int64_t** node_id, int64_t** edge_id,
                                            int64_t* num_edges, int64_t* cumsum_host, void* stream_) {
  PYG_HIP_REQUIRE(host && host->alloc && host->free && (host->rng_blocks || host->mt19937),
                  "dist sampler: host callbacks missing");
  PYG_HIP_REQUIRE(rowptr && (num_seed == 0 || seed) && node_id && edge_id && num_edges && cumsum_host,
                  "dist sampler: NULL argument");
  Ctx c;
  c.host = host;
  c.stream = static_cast<hipStream_t>(stream_);
  int rc = run_dist_sampler(IdxArr(rowptr, index_is32), IdxArr(col, index_is32), seed, num_seed, num_neighbors, node_time,
                            edge_time, seed_time, edge_weight,
                            edge_weight_dtype, temporal_last, replace, disjoint, c, node_id, edge_id, num_edges,
                            cumsum_host);
  c.quiesce_side();
  if (rc != PYG_HIP_OK) (void)hipStreamSynchronize(c.stream);
  c.release_all();
  return rc;
}

extern "C" size_t pyg_hip_relabel_workspace_size(int64_t num_seed, int64_t num_sampled) {
  return relabel_ws_bytes(num_seed < 0 ? 0 : num_seed, num_sampled < 0 ? 0 : num_sampled);
}

// Local ids (Mapper order) of an externally sampled node sequence: the seeds first (batch ids seed_batch0,
// seed_batch0 + 1, ... when disjoint), then the sequence; local_out[j] = id of sampled[j].
static int relabel_nodes_impl(const int64_t* seed, int64_t S, int64_t seed_batch0, int64_t num_batches,
                              const int64_t* sampled, int64_t E, const int64_t* batch, int disjoint, int64_t* local_out,
                              void* workspace, size_t workspace_bytes, hipStream_t stream) {
  using namespace pyg_hip;
  PYG_HIP_REQUIRE(S >= 0 && E >= 0, "relabel: negative size");
  if (E == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE((S == 0 || seed) && sampled && local_out, "relabel: NULL argument");
  PYG_HIP_REQUIRE(!disjoint || batch, "Batch needs to be specified to create disjoint subgraphs");
  if (workspace == nullptr || workspace_bytes < relabel_ws_bytes(S, E))
    return fail(PYG_HIP_ERR_WORKSPACE, "relabel: workspace of %zu bytes needed, got %zu", relabel_ws_bytes(S, E), workspace_bytes);
  if (!disjoint) num_batches = 1;
  PYG_HIP_REQUIRE(num_batches >= 1 && num_batches < (1ll << 22), "relabel: too many seeds for disjoint relabelling");
  u64 cap = 1024;
  while (cap < 2 * (u64)(S + E)) cap <<= 1;
  char* w = static_cast<char*>(workspace);
  HashTable t;
  t.keys = reinterpret_cast<u64*>(w);
  t.vals = t.keys + 1;
  t.mask = cap - 1;
  t.dense = 0;
  w += align_up(sizeof(u64) * 2 * cap, 256);
  u64* seed_slots = reinterpret_cast<u64*>(w);                 // [S] slots of the seeds
  int64_t* seed_copy = reinterpret_cast<int64_t*>(w) + S;       // [S] seed_insert_kernel also writes the node list
  int64_t* seed_batch = reinterpret_cast<int64_t*>(w) + 2 * S;  // [S] ... and its batch list (disjoint)
  u64* slots = reinterpret_cast<u64*>(w) + 3 * S;               // [E] slots of the sampled sequence
  w += align_up(sizeof(u64) * (size_t)(3 * S + E + 1), 256);
  int64_t* tile_buf = reinterpret_cast<int64_t*>(w);
  w += align_up(sizeof(int64_t) * (size_t)((std::max(S, E) + kScanTile - 1) / kScanTile + 2), 256);
  TypeState* ts = reinterpret_cast<TypeState*>(w);
  PYG_HIP_CHECK(hipMemsetAsync(t.keys, 0xFF, sizeof(u64) * 2 * cap, stream));
  PYG_HIP_CHECK(hipMemsetAsync(ts, 0, sizeof(TypeState), stream));
  if (S > 0) {
    // seeds: mapper.fill(seed) / insert({i, seed[i]}) -- ids of first occurrences in position order
    hipLaunchKernelGGL(seed_insert_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, stream, seed, S, seed_batch0,
                       disjoint, num_batches, t, seed_copy, seed_batch, seed_slots, ts);
    PYG_HIP_CHECK(hipGetLastError());
    FlagLoad fl{seed_slots, t.vals};
    AssignStore as{seed_slots, t.vals, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    int rc = device_scan<int64_t, SumOp>(fl, as, S, tile_buf, &ts->distinct, stream);
    if (rc != PYG_HIP_OK) return rc;
  }
  // the sampled sequence: ids continue from the number of distinct seeds (device-resident)
  hipLaunchKernelGGL(relabel_insert_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, stream, sampled,
                     disjoint ? batch : (const int64_t*)nullptr, E, num_batches, t, slots);
  PYG_HIP_CHECK(hipGetLastError());
  FlagLoad fl{slots, t.vals};
  AssignStore as{slots, t.vals, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, ts};
  int rc = device_scan<int64_t, SumOp>(fl, as, E, tile_buf, tile_buf + (E + kScanTile - 1) / kScanTile + 1, stream);
  if (rc != PYG_HIP_OK) return rc;
  hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, stream, slots, t.vals, E, local_out,
                     (const HopInfo*)nullptr, (HopInfo*)nullptr, (ChainState*)nullptr, (TypeState*)nullptr,
                     (const int64_t*)nullptr, (int64_t*)nullptr);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

extern "C" int pyg_hip_relabel_nodes(const int64_t* seed, int64_t num_seed, int64_t seed_batch0, int64_t num_batches,
                                     const int64_t* sampled, int64_t num_sampled, const int64_t* batch, int disjoint,
                                     int64_t* local_out, void* workspace, size_t workspace_bytes, void* stream) {
  return relabel_nodes_impl(seed, num_seed, seed_batch0, num_batches, sampled, num_sampled, batch, disjoint, local_out,
                            workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

extern "C" int pyg_hip_expand_rows(const int64_t* count_prefix, int64_t num_src, int64_t total, int64_t* row_out, void* stream) {
  using namespace pyg_hip;
  if (total <= 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(count_prefix && row_out && num_src >= 0, "expand_rows: bad argument");
  hipLaunchKernelGGL(expand_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     count_prefix, num_src, total, row_out);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

extern "C" int pyg_hip_relabel_neighborhood(const int64_t* seed, int64_t num_seed, const int64_t* sampled,
                                            int64_t num_sampled, const int64_t* count_prefix, int64_t num_src,
                                            const int64_t* batch, int disjoint, int64_t* row_out, int64_t* col_out,
                                            void* workspace, size_t workspace_bytes, void* stream_) {
  using namespace pyg_hip;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int64_t S = num_seed, E = num_sampled;
  PYG_HIP_REQUIRE(S >= 0 && E >= 0 && num_src >= 0, "relabel_neighborhood: negative size");
  if (E == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(count_prefix && row_out && col_out, "relabel_neighborhood: NULL argument");
  int rc = relabel_nodes_impl(seed, S, 0, std::max<int64_t>(S, 1), sampled, E, batch, disjoint, col_out, workspace,
                              workspace_bytes, stream);
  if (rc != PYG_HIP_OK) return rc;
  hipLaunchKernelGGL(expand_rows_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, stream, count_prefix, num_src, E,
                     row_out);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

extern "C" int pyg_hip_segment_concat(const int64_t* const* bases, const int64_t* part, const int64_t* begin,
                                      const int64_t* dst_off, int64_t n, const int64_t* fill, int64_t* out, int64_t total,
                                      void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(n >= 0 && total >= 0, "segment_concat: negative size");
  if (total == 0 || n == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(dst_off && out && (fill || (bases && part && begin)), "segment_concat: NULL argument");
  hipLaunchKernelGGL(segment_concat_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, bases, part, begin,
                     dst_off, n, fill, out, total);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}
