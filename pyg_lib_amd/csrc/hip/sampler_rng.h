// Random-word generation of the sampler: the caller's mt19937 engine continued on the device (sampler_rng.hip),
// plus the small host-side context both translation units of the sampler share.  Internal to libpyg_hip.so.
#pragma once

#include "common.h"

#include <algorithm>
#include <vector>

namespace pyg_hip {
namespace sampler {

typedef unsigned long long u64;

// Device-resident engine position and per-node-type list sizes: the relations of a hop are queued
// back to back (each one starts where the previous one ended) and the host reads the results of the
// whole hop with ONE synchronisation.
struct ChainState {
  int64_t word;   // engine position: linear word index
  int32_t units;  //                  16-bit units left in that word
  int32_t abort;  // sticky: a relation of this hop lacked random words -> everything after it is a no-op
};

// ---- device-side mt19937 (continues the caller's CPU engine) ---------------------------------------
// The reference draws its words with at::randint / Tensor.random_ on torch's CPU generator, i.e.
// at::mt19937 (ATen/core/MT19937RNGEngine.h) + random64() = (hi << 32 | lo) of two consecutive
// outputs + `% (2^64 - 1) + INT64_MIN` (ATen/core/DistributionsHelper.h:40-56).  Generating the
// ~2e5 words of a products-scale batch on the host costs more than all sampling kernels together, so
// the engine state (624 words + left/next) is handed to the device as a kernel argument, the device
// continues the very same stream, and the advanced engine is handed back to the caller afterwards.
//
// Stream coordinates of one call: output o = 0, 1, 2, ... counted from the engine position at call
// start.  The engine's current array A_0 still holds a0 = left - 1 outputs (o < a0 reads
// state[next + o]); array A_j (j >= 1, the j-th mt19937_engine::next_state()) covers
// o in [a0 + 624 (j-1), a0 + 624 j).  Output 2k is the HIGH half of 64-bit word k
// (CPUGeneratorImpl::random64), INT64_MIN is folded in by flipping bit 63; the words are stored
// un-reduced (see RngCursor::next), so the raw engine values can be recovered from them.
struct MtDev {
  uint32_t state[624];
  int32_t left;
  uint32_t next;
};

// The same, queued BEHIND the last hop without the host in between (fully queued mode): the number of
// consumed blocks comes from the device-resident engine position.  status: 0 = state written; 1 = the final
// 624-array is not fully generated yet; 2 = the position is still inside the caller's own array (the host
// adjusts left / next itself).  The host cross-checks n32 against its own bookkeeping before trusting it.
struct MtHandBack {
  MtDev st;
  int64_t n32;
  int32_t status;
  int32_t pad;
};

// inverse of the engine's output tempering (raw state value from a tempered output)
__device__ __forceinline__ uint32_t mt_untemper(uint32_t y) {
  y ^= y >> 18;
  y ^= (y << 15) & 0xefc60000u;
  uint32_t t = y;  // invert y ^= (y << 7) & B: 7 known low bits grow by 7 per round
  for (int i = 0; i < 4; ++i) t = y ^ ((t << 7) & 0x9d2c5680u);
  y = t;
  t = y;           // invert y ^= y >> 11
  for (int i = 0; i < 2; ++i) t = y ^ (t >> 11);
  return t;
}

// tempered engine output o of the device-generated stream (mt_emit's layout: little-endian u64 words, even
// output = high half with bit 63 flipped)
__device__ __forceinline__ uint32_t mt_output_at(const uint32_t* __restrict__ out32, int64_t o) {
  return (o & 1) == 0 ? (out32[o + 1] ^ 0x80000000u) : out32[o - 1];
}

// ---- host context ---------------------------------------------------------------------------------
void dense_cache_release(void* ptr);  // sampler.hip
void rng_carry_abandon(int device);   // sampler_rng.hip

struct Ctx {
  const pyg_hip_sampler_host* host;
  hipStream_t stream;
  hipStream_t side = nullptr;     // side stream with random-word generation in flight (or nullptr)
  volatile int* side_stop = nullptr;
  void quiesce_side() {           // cancel speculation and wait: scratch may be freed afterwards
    if (!side) return;
    if (side_stop) *side_stop = 1;
    (void)hipStreamSynchronize(side);
    side = nullptr;
  }
  void finish_side() {            // wait WITHOUT cancelling: what was queued is completed (the stream carry keeps it)
    if (!side) return;
    (void)hipStreamSynchronize(side);
    side = nullptr;
  }
  // Random-word stream carried over from the previous call on this device (sampler_rng.hip, RngCarry): -1 none, else the
  // device whose carry this call has adopted and must either commit (rng_carry_commit) or give back (release_all).
  int carry_device = -1;
  bool main_idle = false;         // the fused chain saw its closing launch finish (completion word) and queued nothing since:
                                  // the call's last hipStreamSynchronize would only pay the wake-up again
  bool no_carry = false;          // this call never adopts / leaves a carry (batched lanes: a fresh engine per batch)
  std::vector<void*> live;  // every block obtained from host->alloc and not yet handed out/freed
  void* alloc(size_t bytes) {
    void* p = host->alloc(host->user, bytes ? bytes : 16);
    if (p) live.push_back(p);
    return p;
  }
  void release(void* p) {
    if (!p) return;
    auto it = std::find(live.begin(), live.end(), p);
    if (it != live.end()) live.erase(it);
    host->free(host->user, p);
  }
  void keep(void* p) {  // ownership passes to the caller
    auto it = std::find(live.begin(), live.end(), p);
    if (it != live.end()) live.erase(it);
  }
  std::vector<void*> cached_tables;  // DenseCache blocks lent to this call (sampler.hip), handed back with the scratch
  void release_all() {
    for (void* p : live) host->free(host->user, p);
    live.clear();
    for (void* p : cached_tables) dense_cache_release(p);
    cached_tables.clear();
    if (carry_device >= 0) rng_carry_abandon(carry_device);   // adopted but not committed (a failed or repeated call)
    carry_device = -1;
  }
};

#define PYG_ALLOC(ptr, type, ctx, bytes)                                                   \
  do {                                                                                     \
    ptr = static_cast<type>((ctx).alloc(bytes));                                           \
    if (!ptr) return fail(PYG_HIP_ERR_RUNTIME, "sampler: device allocation of %zu bytes failed", \
                          (size_t)(bytes));                                                \
  } while (0)

// pinned host scratch of the calling thread (grown on demand)
int get_pinned(void** out, size_t bytes);

// Per-thread, per-device side stream + event pool for the speculative word generation.
struct SideStream {
  int device = -1;
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> events;
  size_t used = 0;
  int next_event(hipEvent_t* ev) {
    if (used == events.size()) {
      hipEvent_t e;
      PYG_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      events.push_back(e);
    }
    *ev = events[used++];
    return PYG_HIP_OK;
  }
};

int get_side_stream(SideStream** out);

struct RngHost {
  int64_t blocks = 0;         // 128-word blocks consumed (prefetched, in the reference's terms) so far
  u64* dev = nullptr;         // device copy of all blocks (word 0 of THIS call)
  u64* dev_base = nullptr;    // the allocation `dev` points into (== dev unless the stream was carried over)
  int64_t dev_cap_blocks = 0; // blocks from `dev` to the end of the allocation
  bool carried = false;       // this call continues the previous call's stream (RngCarry): outputs [0, marks[0].upto32)
                              // were complete before it started
  bool topup_deferred = false;  // carried and fully covered: the next round is queued behind the call's last launch
  int64_t word = 0;           // engine state: linear word index
  int units = 4;              //               16-bit units left in that word
  int64_t raw_used = 0;       // generator outputs consumed directly so far (biased sampling's uniform_)
  int64_t cur_shift = 0;      // raw_used at the time the engine's CURRENT block was fetched (WordSrc)
  // device continuation of the caller's mt19937 (fast path)
  bool engine = false;
  MtDev init;                 // the caller's engine at call start
  int64_t a0 = 0;             // outputs left in its current array
  int64_t o_r0 = 0;           // engine output index of raw position 0 of the current base window
  bool started = false;       // a round has been launched: the base window lives in `window`
  uint32_t* window = nullptr; // device: the last 624 raw values generated (base window of the next round)
  uint32_t* base_raw = nullptr;   // device: raw copy of a round's base window (jump kernel input)
  uint32_t* windows = nullptr;    // device: start windows of segments 1 .. kMtMaxSeg-1
  volatile int* stop = nullptr;
  SideStream* side = nullptr;
  struct Mark {
    int64_t upto32;           // outputs complete once `ev` has fired
    hipEvent_t ev;
  };
  std::vector<Mark> marks;
  size_t waited = 0;          // marks[0 .. waited) are already ordered before the main stream
  // rng_begin in two halves (fused chain): the first round is LAUNCHED behind the seeds' scan launch of the main stream
  hipEvent_t begin_ev = nullptr;   // main-stream point the side stream is ordered behind (block reuse)
  int64_t begin_target32 = 0;      // outputs the first round must cover
  // outputs [0, generated32) exist once the last launched round has finished
  int64_t generated32() const { return started ? o_r0 + 624 : 0; }
};

constexpr int64_t kSpecCapWords = (int64_t)kMtMaxSeg * kMtSeg / 2;  // one round: 655 k words (5 MB)

// Launches rounds on the side stream until at least `target32` outputs exist.
int rng_generate(Ctx& c, RngHost& r, int64_t target32);
// Start of a call: adopts the caller's engine (or the callback path) and queues the speculative first round.
int rng_begin(Ctx& c, RngHost& r, void* pinned, const std::vector<int64_t>& spec_words);
// The same in two halves: _prepare adopts the engine, allocates and records the main-stream point the side stream has to
// stay behind (cheap: no launch); _launch queues the first round.  Main-stream work queued in between runs CONCURRENTLY
// with the round (the fused chain's seeds launch: its 1024 seeds otherwise sat ~20 us of host time behind the round).
int rng_begin_prepare(Ctx& c, RngHost& r, void* pinned, const std::vector<int64_t>& spec_words, bool allow_carry = false);
int rng_begin_launch(Ctx& c, RngHost& r);
// The stream carry (RngCarry, sampler_rng.hip).  A data loader calls the sampler batch after batch on ONE generator: the
// engine a call hands back is the engine the next call presents, and the words the previous call generated beyond its
// own consumption ARE the next call's words.  rng_carry_commit keeps them (buffer, base windows, the handed-back engine)
// per device; rng_begin_prepare(allow_carry) adopts them when the presented engine equals the kept one bit for bit -- then
// no generation launch and no cross-stream wait lies in front of any hop; the next round is queued in the background.
// Any other engine (reseeded, used elsewhere in between) misses, and the call starts cold as before.
// PYG_HIP_SAMPLER_RNG_CARRY=0 disables it.
void rng_carry_commit(Ctx& c, RngHost& r, const MtDev& handed_back, int64_t n32);
int rng_topup_deferred(Ctx& c, RngHost& r);
int rng_carry_release_idle(const pyg_hip_sampler_host* host);
void rng_carry_stats(long long* adopted, long long* cold);
// Orders the main stream behind the generation of everything up to `last_word` / `need32` outputs.
int rng_wait(Ctx& c, RngHost& r, int64_t last_word, int64_t* avail_blocks);
int rng_wait32(Ctx& c, RngHost& r, int64_t need32, int64_t* avail_blocks);
// Makes sure the words up to `last_word` exist (generating more if the speculation fell short).
int rng_ensure(Ctx& c, RngHost& r, int64_t last_word);
// End of a call: hands the advanced engine back to the caller.
int rng_finish(Ctx& c, RngHost& r);
// Fully queued mode: the engine hand-back rides behind the last hop, its position read from `chain` on the device
// (mt_finish_chain_kernel); *hb_dev_out receives the device record the caller copies to pinned memory.
int rng_queue_hand_back(Ctx& c, RngHost& r, const ChainState* chain, int64_t generated32, MtHandBack* hb_dev);

}  // namespace sampler
}  // namespace pyg_hip
