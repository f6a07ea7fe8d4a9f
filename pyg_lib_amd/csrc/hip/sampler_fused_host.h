// Host side of the fused chain (sampler_fused.h): 3 launches per hop, every record in the launch's kernel argument (no
// staging copy), the write-once tables read back once behind the last launch.  Included by sampler.hip.

struct FusedSeed {  // one seed set, recorded by the seeds loop of run_sampler
  int type;
  int64_t S;
  u64* slots;
  const int64_t* seed;  // folded == true: the insert kernel was NOT queued -- the seeds' scan launch inserts them itself
  int64_t batch0;
  TypeState* ts;
  bool folded;
};

// PYG_HIP_SAMPLER_ONEPASS=0: the scans as reduce + apply launches, seeds inserted by their own kernel (A/B timing)
inline bool fused_onepass_enabled() {
  static const bool on = [] {
    const char* e = getenv("PYG_HIP_SAMPLER_ONEPASS");
    return !(e != nullptr && e[0] == '0');
  }();
  return on;
}
// the seeds of a call can be folded into their scan launch: one seed set of one tile
inline bool fused_seed_foldable(const pyg_hip_seed_set* seeds, int num_seed_sets) {
  if (!fused_onepass_enabled()) return false;
  int sets = 0;
  int64_t S = 0;
  for (int s = 0; s < num_seed_sets; ++s)
    if (seeds[s].num_seed > 0) {
      ++sets;
      S = seeds[s].num_seed;
    }
  return sets == 1 && S <= kScanTile;
}

// consumers of node type t for hop `next` (relations of that hop that expand t), in relation order
inline std::vector<int> fused_consumers(const pyg_hip_relation* rels, int num_relations, int csc,
                                        const std::vector<std::vector<int64_t>>& eb, int next, int L, int t) {
  std::vector<int> out;
  if (next >= L) return out;
  for (int e = 0; e < num_relations; ++e) {
    const int src = !csc ? rels[e].src_type : rels[e].dst_type;
    if (src == t && eb[(size_t)next][(size_t)e] != 0) out.push_back(e);
  }
  return out;
}

// can the call run the fused chain?  (every type has at most kMaxCons consumers per hop, tile counts stay small)
inline bool fused_eligible(const pyg_hip_relation* rels, int num_relations, int num_node_types, int num_seed_sets,
                           int csc, int L, const std::vector<std::vector<int64_t>>& eb) {
  if (num_node_types > 64 || L * num_relations > 64 || num_seed_sets > kMaxParts) return false;
  static const bool off = [] {
    const char* e = getenv("PYG_HIP_SAMPLER_FUSED");
    return e != nullptr && e[0] == '0';
  }();
  if (off) return false;
  for (int next = 0; next <= L; ++next) {  // next = the hop whose relations consume what phase next - 1 appends
    int launch_cons = 0;
    for (int t = 0; t < num_node_types; ++t) {
      const int nc = (int)fused_consumers(rels, num_relations, csc, eb, next, L, t).size();
      if (nc > kMaxCons) return false;
      launch_cons += nc;  // (upper bound: types without a segment in that phase carry none)
    }
    if (launch_cons > kMaxLaunchCons) return false;
  }
  for (int ell = 0; ell < L; ++ell) {
    std::vector<int64_t> tiles((size_t)num_node_types, 0);
    int queued = 0;
    for (int e = 0; e < num_relations; ++e) {
      const int dst = !csc ? rels[e].dst_type : rels[e].src_type;
      if (eb[(size_t)ell][(size_t)e] != 0) ++queued;
      tiles[(size_t)dst] += (eb[(size_t)ell][(size_t)e] + kScanTile - 1) / kScanTile;
      if (tiles[(size_t)dst] > 8192) return false;  // every apply block reduces the tiles in front of it
    }
    if (queued > kMaxParts) return false;
  }
  return true;
}

int run_fused_chain(Ctx& c, int num_node_types, int num_relations, const pyg_hip_relation* rels, int L, int csc,
                    int replace, int disjoint, int64_t num_batches, const int64_t* const* node_time, int temporal_last,
                    int64_t* seed_times, int* err_flag, std::vector<NodeSet>& ns, std::vector<RelState>& rs,
                    const std::vector<std::vector<int64_t>>& eb, const std::vector<std::vector<int64_t>>& fbh,
                    const std::vector<int64_t>& node_bound, const std::vector<int64_t>& rel_bound,
                    const std::vector<FusedSeed>& fseeds, RngHost& rng, bool rng_late, ChainState* chain, char* tables_host,
                    MtHandBack* hand_back_host, MtHandBack** hand_back_out,
                    std::vector<std::vector<int64_t>>& nodes_per_hop, PhaseTimer& pt) {
  hipStream_t stream = c.stream;
  const int R = num_relations, T = num_node_types;
  // completion word of the closing launch (pinned; polled at the end: see there).  A fresh sequence number per call.
  static std::atomic<unsigned long long> done_counter{0};
  unsigned long long* done_word = reinterpret_cast<unsigned long long*>(tables_host - 64);
  const unsigned long long done_seq = 0x5eed000000000000ull | (done_counter.fetch_add(1) + 1);
  *done_word = 0;
  auto tiles_of = [](int64_t n) { return (int)((n + kScanTile - 1) / kScanTile); };
  auto src_of = [&](int e) { return !csc ? rels[e].src_type : rels[e].dst_type; };
  auto dst_of = [&](int e) { return !csc ? rels[e].dst_type : rels[e].src_type; };

  // ---- lists, tables and outputs at their bound size (as in round 2's fully queued mode) ----
  for (int t = 0; t < T; ++t) {
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
    // a seeded type reserved its table for the whole call already (its seeds' slot handles must stay valid: no rehash)
    const int64_t want = n.nodes.size + node_bound[(size_t)t];
    if (n.entries_bound < want) {
      PYG_HIP_REQUIRE(n.nodes.size == 0 || n.table.dense, "sampler: internal error (seeded table would be rehashed)");
      rc = table_reserve(c, n, want - n.entries_bound, 0, true);
      if (rc != PYG_HIP_OK) return rc;
    }
  }
  for (int e = 0; e < R; ++e) {
    RelState& st = rs[(size_t)e];
    if (rel_bound[(size_t)e] == 0) continue;
    int rc = st.row.reserve(c, rel_bound[(size_t)e]);
    if (rc == PYG_HIP_OK) rc = st.col.reserve(c, rel_bound[(size_t)e]);
    if (rc == PYG_HIP_OK) rc = st.eid.reserve(c, rel_bound[(size_t)e]);
    if (rc != PYG_HIP_OK) return rc;
  }

  // ---- arena: staging mirror + per-step scratch ----
  const size_t tb_bytes =
      align_up(8 * (size_t)(L + 1) * T + 8 * (size_t)T + sizeof(CountAgg) * (size_t)L * R + 4 * (size_t)L * R + 4, 16);
  size_t arena_bytes = align_up(tb_bytes, 256);
  for (const FusedSeed& fsd : fseeds) {
    const size_t nc = fused_consumers(rels, R, csc, eb, 0, L, fsd.type).size();
    arena_bytes += align_up(8 * (size_t)fsd.S, 256) + align_up((8 + 16 * nc) * (size_t)(tiles_of(fsd.S) + 1), 256);
  }
  for (int ell = 0; ell < L; ++ell) {
    std::vector<size_t> seg_tiles((size_t)T, 0);
    for (int e = 0; e < R; ++e) {
      const int64_t Eb = eb[(size_t)ell][(size_t)e];
      if (Eb == 0) continue;
      const int64_t Fb = fbh[(size_t)ell][(size_t)src_of(e)];
      arena_bytes += 2 * align_up(8 * (size_t)Fb, 256) + (disjoint ? 4 : 3) * align_up(8 * (size_t)Eb, 256);
      seg_tiles[(size_t)dst_of(e)] += (size_t)tiles_of(Eb);
    }
    for (int t = 0; t < T; ++t)
      if (seg_tiles[(size_t)t]) {
        arena_bytes += align_up((8 + 16 * (size_t)kMaxCons) * (seg_tiles[(size_t)t] + 1), 256);
        if (ell == L - 1) arena_bytes += align_up(sizeof(LastTile) * seg_tiles[(size_t)t], 256) + align_up(4 * seg_tiles[(size_t)t], 256);
      }
  }
  // Which scans run as one launch (fused_onepass: up to kOnePassMaxTiles tiles)?  Their tickets (own 64-byte lines) and
  // tile aggregates are cleared per call.
  const bool onepass = fused_onepass_enabled();
  size_t sync_bytes = 64 * (size_t)(L + 1);
  bool seeds_onepass = onepass;
  {
    size_t tiles = 0;
    for (const FusedSeed& fsd : fseeds) tiles += (size_t)tiles_of(fsd.S);
    seeds_onepass = onepass && tiles <= (size_t)kOnePassMaxTiles;
    if (seeds_onepass) sync_bytes += 8 * kAggStride * (tiles + fseeds.size());
  }
  std::vector<char> hop_onepass((size_t)L, 0);
  // The last hop's bookkeeping runs as the two passes of fused_last_* (no table writes, no finalize role);
  // PYG_HIP_SAMPLER_LAST=0: reduce / apply / finalize as for the other hops (A/B timing).
  static const bool last_on = [] {
    const char* e = getenv("PYG_HIP_SAMPLER_LAST");
    return !(e != nullptr && e[0] == '0');
  }();
  const bool last_two_pass = last_on && L > 0;
  for (int ell = 0; ell < L; ++ell) {
    if (last_two_pass && ell == L - 1) continue;   // (neither the one-pass scan nor its sync words)
    size_t tiles = 0, parts = 0;
    for (int e = 0; e < R; ++e)
      if (eb[(size_t)ell][(size_t)e] != 0) {
        tiles += (size_t)tiles_of(eb[(size_t)ell][(size_t)e]);
        ++parts;
      }
    hop_onepass[(size_t)ell] = onepass && parts > 0 && tiles < (size_t)kOnePassMaxTiles;
    if (hop_onepass[(size_t)ell]) sync_bytes += 8 * kAggStride * (tiles + parts);
  }
  const size_t sync_words = sync_bytes / 4;
  arena_bytes += align_up(sync_bytes, 256);
  pt.mark("chain_entry");
  char* arena;
  PYG_ALLOC(arena, char*, c, arena_bytes);
  pt.mark("arena");
  auto carve = [&](size_t bytes) {
    char* p = arena;
    arena += align_up(bytes, 256);
    return p;
  };
  // ---- tables (device; initialised by a kernel) ----
  char* tb_dev = carve(tb_bytes);
  FTables tb;
  tb.size_at = reinterpret_cast<int64_t*>(tb_dev);
  tb.dup = tb.size_at + (size_t)(L + 1) * T;
  tb.tot = reinterpret_cast<CountAgg*>(tb.dup + T);
  tb.overflow = reinterpret_cast<int32_t*>(tb.tot + (size_t)L * R);
  tb.wide = tb.overflow + (size_t)L * R;
  tb.word0 = rng.word;
  tb.units0 = rng.units;
  tb.L = L;
  tb.R = R;
  tb.T = T;
  tb.is32 = 0;
  for (int e = 0; e < R; ++e) tb.is32 |= rels[e].index_is32 ? 1 : 0;
  unsigned* sync = reinterpret_cast<unsigned*>(carve(sync_bytes));
  char* sync_next = reinterpret_cast<char*>(sync) + 64 * (size_t)(L + 1);
  auto ticket_of = [&](int launch) { return sync + 16 * (size_t)launch; };  // 0: seeds, 1 + l: hop l
  auto tile_agg_of = [&](bool one_launch, int ncons, int tiles) {  // aggregates of a segment's tiles
    if (!one_launch) return carve((8 + 16 * (size_t)ncons) * (size_t)(tiles + 1));
    char* p = sync_next;
    sync_next += 8 * kAggStride * (size_t)(tiles + 1);
    return p;
  };
  const bool fold_seeds = fseeds.size() == 1 && fseeds[0].folded && sync_words <= 65536;
  if (!fold_seeds) {
    for (const FusedSeed& fsd : fseeds) {
      if (!fsd.folded) continue;  // left to the chain, but too much for one block to clear: the insert kernel after all
      NodeSet& n = ns[(size_t)fsd.type];
      hipLaunchKernelGGL(seed_insert_kernel, dim3((unsigned)((fsd.S + 255) / 256)), dim3(256), 0, stream, fsd.seed, fsd.S,
                         fsd.batch0, disjoint, num_batches, n.table, n.nodes.p, disjoint ? n.batch.p : (int64_t*)nullptr,
                         fsd.slots, fsd.ts);
      PYG_HIP_CHECK(hipGetLastError());
    }
    const int cells = std::max((int)sync_words, std::max((L + 1) * T, L * R));
    hipLaunchKernelGGL(fused_init_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, tb, sync, (int)sync_words);
    PYG_HIP_CHECK(hipGetLastError());
  }

  auto add_sample = [&](FSampleLaunch& l, int role, int64_t blocks, int idx) {
    if (blocks <= 0) return;
    const int k = l.n++;
    l.cum[k] = (k > 0 ? l.cum[k - 1] : 0) + (int)blocks;
    l.role[k] = (unsigned char)role;
    l.idx[k] = (unsigned char)idx;
  };
  auto add_part = [&](FScanLaunch& l, int nc, int64_t blocks, const FPart* part) {
    if (blocks <= 0) return;
    const int k = l.n++;
    l.cum[k] = (k > 0 ? l.cum[k - 1] : 0) + (int)blocks;
    l.nc[k] = nc;
    if (part) l.part[k] = *part;
  };
  int64_t avail_blocks = 0;
  auto launch_sample = [&](const FSampleLaunch& l) -> int {
    if (l.n == 0) return PYG_HIP_OK;
    int gmax = 8;
    for (int k = 0; k < l.n; ++k) {
      if (l.role[k] <= kRoleSample64) gmax = std::max(gmax, 8 << l.role[k]);
      if (l.role[k] == kRoleSampleWave) gmax = 64;
    }
    const dim3 grid((unsigned)l.cum[l.n - 1]), block(256);
    const u64* words = rng.dev;
#define PYG_FUSED_SAMPLE_LAUNCH(G)                                                                                  \
  do {                                                                                                              \
    if (tb.is32) hipLaunchKernelGGL((fused_sample_kernel<G, false>), grid, block, 0, stream, l, avail_blocks, words); \
    else hipLaunchKernelGGL((fused_sample_kernel<G, true>), grid, block, 0, stream, l, avail_blocks, words);         \
  } while (0)
    if (gmax <= 8) PYG_FUSED_SAMPLE_LAUNCH(8);
    else if (gmax <= 16) PYG_FUSED_SAMPLE_LAUNCH(16);
    else if (gmax <= 32) PYG_FUSED_SAMPLE_LAUNCH(32);
    else PYG_FUSED_SAMPLE_LAUNCH(64);
#undef PYG_FUSED_SAMPLE_LAUNCH
    PYG_HIP_CHECK(hipGetLastError());
    return PYG_HIP_OK;
  };
  auto launch_scan = [&](const FScanLaunch& l, int mode) -> int {  // kScanReduce / kScanApply / kScanOnePass / kScanSeedFold
    if (l.n == 0) return PYG_HIP_OK;
    int maxnc = 0;
    for (int k = 0; k < l.n; ++k) maxnc = std::max(maxnc, l.nc[k]);
    const dim3 grid((unsigned)l.cum[l.n - 1]), block(256);
#define PYG_FUSED_LAUNCH2(N, I64)                                                                                       \
  do {                                                                                                                  \
    if (mode == kScanReduce) hipLaunchKernelGGL((fused_scan_kernel<N, kScanReduce, I64>), grid, block, 0, stream, l);    \
    else if (mode == kScanApply) hipLaunchKernelGGL((fused_scan_kernel<N, kScanApply, I64>), grid, block, 0, stream, l); \
    else if (mode == kScanOnePass) hipLaunchKernelGGL((fused_scan_kernel<N, kScanOnePass, I64>), grid, block, 0, stream, l); \
    else hipLaunchKernelGGL((fused_scan_kernel<N, kScanSeedFold, I64>), grid, block, 0, stream, l);                     \
  } while (0)
#define PYG_FUSED_LAUNCH(N)                 \
  case N:                                   \
    if (tb.is32) PYG_FUSED_LAUNCH2(N, false); \
    else PYG_FUSED_LAUNCH2(N, true);        \
    break;
    switch (maxnc) {
      PYG_FUSED_LAUNCH(0)
      PYG_FUSED_LAUNCH(1)
      PYG_FUSED_LAUNCH(2)
      default:
        PYG_FUSED_LAUNCH(3)
    }
#undef PYG_FUSED_LAUNCH
#undef PYG_FUSED_LAUNCH2
    PYG_HIP_CHECK(hipGetLastError());
    return PYG_HIP_OK;
  };
  struct ConsArr {
    int64_t* edge_off = nullptr;
    RngTab* tabp = nullptr;
  };
  std::vector<ConsArr> cons_arr((size_t)std::max(L * R, 1));
  auto range_of = [&](int e, const NodeSet& sn) {
    const pyg_hip_relation& r = rels[e];
    RangeCtx range;
    range.rowptr = IdxArr(r.rowptr, r.index_is32);
    range.col = IdxArr(r.col, r.index_is32);
    range.time = r.edge_time ? r.edge_time : (node_time ? node_time[dst_of(e)] : nullptr);
    range.edge_level = r.edge_time ? 1 : 0;
    range.last = temporal_last;
    range.seed_times = seed_times;
    range.batch = disjoint ? sn.batch.p : nullptr;
    range.error = err_flag;
    return range;
  };
  // appends the consumers of type t for hop `next` to the launch's consumer array (allocating their per-node prefix
  // arrays) and returns the index of the first one
  auto fill_consumers = [&](FScanLaunch& l, int* used, FSegHdr& sh, int t, int next) -> int {
    const std::vector<int> cons = fused_consumers(rels, R, csc, eb, next, L, t);
    sh.ncons = (int)cons.size();
    const int first = *used;
    for (size_t q = 0; q < cons.size(); ++q) {
      const int e2 = cons[q];
      const int64_t Fb = fbh[(size_t)next][(size_t)t];
      ConsArr& ca = cons_arr[(size_t)next * R + e2];
      ca.edge_off = reinterpret_cast<int64_t*>(carve(8 * (size_t)Fb));
      ca.tabp = reinterpret_cast<RngTab*>(carve(8 * (size_t)Fb));
      FConsumer& fc = l.cons[(*used)++];
      fc.range = range_of(e2, ns[(size_t)t]);
      fc.count = rels[e2].num_neighbors_host[next];
      fc.replace = replace;
      fc.tot_index = next * R + e2;
      fc.edge_off = ca.edge_off;
      fc.tabp = ca.tabp;
    }
    return first;
  };
  auto seg_base = [&](FSegHdr& sh, int t) {
    NodeSet& n = ns[(size_t)t];
    ::memset(&sh, 0, sizeof(sh));
    sh.vals = n.table.vals;
    sh.prov = n.table.prov;
    sh.tag = n.table.tag;
    sh.idmask = n.table.idmask;
    sh.nodes = n.nodes.p;
    sh.batch = disjoint ? n.batch.p : nullptr;
    sh.dup = tb.dup + t;
  };

  // ---- seeds: one segment per seeded type; reduce + apply, queued right away ----
  {
    FScanLaunch sc;
    ::memset(&sc, 0, sizeof(sc));
    sc.tb = tb;
    int used = 0;
    for (const FusedSeed& fsd : fseeds) {
      if (fsd.S == 0) continue;
      FPart pp;
      ::memset(&pp, 0, sizeof(pp));
      seg_base(pp.h, fsd.type);
      pp.h.size_in = nullptr;
      pp.h.size_out = tb.size_at + fsd.type;
      pp.h.seeds = 1;
      pp.cons0 = fill_consumers(sc, &used, pp.h, fsd.type, 0);
      const int nt = tiles_of(fsd.S);
      pp.h.tile_agg = tile_agg_of(seeds_onepass, pp.h.ncons, nt);
      pp.slots = fsd.slots;
      pp.e_node = ns[(size_t)fsd.type].nodes.p;
      pp.e_batch = disjoint ? ns[(size_t)fsd.type].batch.p : nullptr;
      pp.cache = reinterpret_cast<u64*>(carve(8 * (size_t)fsd.S));
      pp.n_fixed = fsd.S;
      pp.pos_base = 0;
      pp.tot_index = 0;
      pp.tile0 = 0;
      pp.last = 1;
      add_part(sc, pp.h.ncons, nt, &pp);
    }
    pt.mark("seedscan_built");
    int rc;
    if (fold_seeds) {
      const FusedSeed& fsd = fseeds[0];
      PYG_HIP_REQUIRE(sc.n == 1 && sc.cum[0] == 1, "sampler: internal error (folded seeds are one tile)");
      sc.fold.seed = fsd.seed;
      sc.fold.batch0 = fsd.batch0;
      sc.fold.num_batches = num_batches;
      sc.fold.table = ns[(size_t)fsd.type].table;
      sc.fold.ts = fsd.ts;
      sc.fold.sync = sync;
      sc.fold.sync_words = (int)sync_words;
      sc.fold.disjoint = disjoint;
      rc = launch_scan(sc, kScanSeedFold);
    } else if (seeds_onepass) {
      sc.ticket = ticket_of(0);
      rc = launch_scan(sc, kScanOnePass);
    } else {
      rc = launch_scan(sc, kScanReduce);
      if (rc == PYG_HIP_OK) rc = launch_scan(sc, kScanApply);
    }
    if (rc != PYG_HIP_OK) return rc;
  }
  if (rng_late) {  // the first round of the word generation, behind the seeds' launch (run_sampler: rng_begin_prepare)
    int rc = rng_begin_launch(c, rng);
    if (rc != PYG_HIP_OK) return rc;
    pt.mark("rng_launched");
  }

  // The closing role's record: where the tables and the engine hand-back go in pinned host memory (kernels write there
  // directly, as they do for the temporal error flag: no launch and no copy behind the last hop), the completion word.
  // `spec_word` = the bound of the words the call may have consumed when the role runs.
  int64_t spec_word = rng.word;
  bool closed = false;   // the role rode in the last hop's second pass (fused_last_apply_kernel): no closing launch
  auto fill_fold = [&](FFoldRec& f) -> int {
    ::memset(&f, 0, sizeof(f));
    f.tables_host = tables_host;
    f.tables_dev = tb_dev;
    f.tables_bytes = (int)tb_bytes;
    f.done = done_word;
    f.done_seq = done_seq;
    *hand_back_out = nullptr;
    if (rng.engine) {
      const int64_t need32 = (spec_word / 128 + 1) * 256 + 624;
      size_t k = 0;
      while (k < rng.marks.size() && rng.marks[k].upto32 < need32) ++k;
      if (k < rng.marks.size()) {
        hand_back_host->status = -1;
        if (k >= rng.waited) {
          if (rng.marks[k].ev) PYG_HIP_CHECK(hipStreamWaitEvent(stream, rng.marks[k].ev, 0));
          rng.waited = k + 1;
        }
        f.hb = hand_back_host;
        f.a0 = rng.a0;
        f.generated32 = rng.marks[k].upto32;
        *hand_back_out = hand_back_host;
      }
    }
    return PYG_HIP_OK;
  };

  // ---- hops ----
  struct Step {
    int ell, e;
    FFinalRec fin;
    int64_t Eb = 0;
  };
  std::vector<Step> prev_steps;
  std::vector<std::vector<Step>> steps_by_hop((size_t)L);
  for (int ell = 0; ell < L; ++ell) {
    FSampleLaunch p1;
    FScanLaunch p2;
    ::memset(&p1, 0, sizeof(p1));
    ::memset(&p2, 0, sizeof(p2));
    p1.tb = tb;
    p1.chain = chain;
    p2.tb = tb;
    p2.ell = ell;
    int nf = 0, nsmp = 0, used = 0;
    for (const Step& s : prev_steps) {
      p1.f[nf] = s.fin;
      add_sample(p1, kRoleFinalize, (s.Eb + 255) / 256, nf++);
    }
    // segments of this hop
    std::vector<FSegHdr> segs((size_t)T);
    std::vector<int> seg_cons0((size_t)T, 0), seg_tiles((size_t)T, 0), seg_last((size_t)T, -1);
    std::vector<int64_t> seg_pos((size_t)T, 0);
    for (int e = 0; e < R; ++e)
      if (eb[(size_t)ell][(size_t)e] != 0) seg_last[(size_t)dst_of(e)] = e;
    for (int t = 0; t < T; ++t) {
      if (seg_last[(size_t)t] < 0) continue;
      if (t < 32) p2.type_mask_lo |= 1u << t;
      else p2.type_mask_hi |= 1u << (t - 32);
      FSegHdr& sh = segs[(size_t)t];
      seg_base(sh, t);
      sh.size_in = tb.size_at + (size_t)ell * T + t;
      sh.size_out = tb.size_at + (size_t)(ell + 1) * T + t;
      sh.seeds = 0;
      seg_cons0[(size_t)t] = fill_consumers(p2, &used, sh, t, ell + 1);
      int nt = 0;
      for (int e = 0; e < R; ++e)
        if (dst_of(e) == t) nt += tiles_of(eb[(size_t)ell][(size_t)e]);
      sh.ntiles = nt;
      if (last_two_pass && ell == L - 1) {
        sh.last_meta = reinterpret_cast<LastTile*>(carve(sizeof(LastTile) * (size_t)nt));
        sh.last_cnt = reinterpret_cast<uint32_t*>(carve(4 * (size_t)nt));
      } else {
        sh.tile_agg = tile_agg_of(hop_onepass[(size_t)ell] != 0, sh.ncons, nt);
      }
    }
    std::vector<Step> cur;
    for (int e = 0; e < R; ++e) {
      const int64_t Eb = eb[(size_t)ell][(size_t)e];
      if (Eb == 0) continue;
      const pyg_hip_relation& r = rels[e];
      const int src = src_of(e), dst = dst_of(e);
      NodeSet& sn = ns[(size_t)src];
      NodeSet& dn = ns[(size_t)dst];
      RelState& st = rs[(size_t)e];
      const int64_t count = r.num_neighbors_host[ell];
      const int64_t Fb = fbh[(size_t)ell][(size_t)src];
      int64_t* e_node = reinterpret_cast<int64_t*>(carve(8 * (size_t)Eb));
      int64_t* e_batch = disjoint ? reinterpret_cast<int64_t*>(carve(8 * (size_t)Eb)) : nullptr;
      // a direct-address table's slot IS the node id (table_slot; non-disjoint keys are plain ids): the slot array is the
      // node array -- one store, one line per 8 emissions instead of two in the sampling launch, and the scan's and the
      // finalize step's slot reads hit what the node reads fetched (24 - 32 bytes per emission less HBM traffic)
      u64* e_slot = (dn.table.dense && !disjoint) ? reinterpret_cast<u64*>(e_node) : reinterpret_cast<u64*>(carve(8 * (size_t)Eb));
      u64* cache = reinterpret_cast<u64*>(carve(8 * (size_t)Eb));
      const ConsArr& ca = cons_arr[(size_t)ell * R + e];
      PYG_HIP_REQUIRE(ca.edge_off != nullptr, "sampler: internal error (no producer for a queued relation)");
      FSampleRec& sr = p1.s[nsmp];
      sr.nodes = sn.nodes.p;
      sr.batch = disjoint ? sn.batch.p : nullptr;
      sr.range = range_of(e, sn);
      sr.count = count;
      sr.num_batches = num_batches;
      sr.edge_off = ca.edge_off;
      sr.tabp = ca.tabp;
      sr.e_row = st.row.p;
      sr.e_node = e_node;
      sr.e_batch = e_batch;
      sr.e_eid = st.eid.p;
      sr.e_slot = e_slot;
      sr.table = dn.table;
      sr.pos_base = seg_pos[(size_t)dst];
      sr.replace = replace;
      sr.ell = ell;
      sr.e = e;
      sr.t_src = src;
      const int role = count <= 8 ? kRoleSample8 : count <= 16 ? kRoleSample16 : count <= 32 ? kRoleSample32
                       : count <= 64 ? kRoleSample64 : kRoleSampleWave;
      const int64_t per = count <= 8 ? 32 : count <= 16 ? 16 : count <= 32 ? 8 : 4;  // frontier nodes per block
      add_sample(p1, role, (Fb + per - 1) / per, nsmp++);
      FPart pp;
      ::memset(&pp, 0, sizeof(pp));
      pp.h = segs[(size_t)dst];
      pp.cons0 = seg_cons0[(size_t)dst];
      pp.slots = e_slot;
      pp.e_node = e_node;
      pp.e_batch = e_batch;
      pp.cache = cache;
      pp.n_fixed = -1;
      pp.pos_base = seg_pos[(size_t)dst];
      pp.tot_index = ell * R + e;
      pp.tile0 = seg_tiles[(size_t)dst];
      pp.last = seg_last[(size_t)dst] == e ? 1 : 0;
      p2.out_col[p2.n] = st.col.p;   // (add_part below takes slot p2.n)
      add_part(p2, pp.h.ncons, tiles_of(Eb), &pp);
      seg_pos[(size_t)dst] += Eb;
      seg_tiles[(size_t)dst] += tiles_of(Eb);
      Step s;
      s.ell = ell;
      s.e = e;
      s.Eb = Eb;
      s.fin.slots = e_slot;
      s.fin.vals = dn.table.vals;
      s.fin.idmask = dn.table.idmask;
      s.fin.out_col = st.col.p;
      s.fin.ell = ell;
      s.fin.e = e;
      cur.push_back(s);
    }
    // ---- queue the hop: [finalize(l - 1) | sample(l)] -> reduce -> apply (+ carry) ----
    // (16-bit draws: four per word; a row of degree >= 2^16 draws 32-bit numbers, two per word -- the slack covers a few
    // hundred such draws per relation and hop, more of them end in the overflow flag and the synchronising repeat)
    for (const Step& s : cur) spec_word += (s.Eb + 3) / 4 + 1 + kWideSlackWords;
    int rc = PYG_HIP_OK;
    if (!cur.empty()) {  // order the words this hop may read (16-bit draws, cumulative bound) before its sampling launch
      pt.mark("hop_built");
      rc = rng_wait(c, rng, spec_word, &avail_blocks);
      pt.mark("rng_waited");
      if (rc != PYG_HIP_OK) return rc;
    }
    rc = launch_sample(p1);
    if (rc != PYG_HIP_OK) return rc;
    const bool last_hop = last_two_pass && ell == L - 1;
    if (last_hop) {
      if (p2.n > 0) {
        hipLaunchKernelGGL(fused_last_reduce_kernel, dim3((unsigned)p2.cum[p2.n - 1]), dim3(256), 0, stream, p2);
        PYG_HIP_CHECK(hipGetLastError());
      }
      add_part(p2, -1, 1, nullptr);  // carry block: second pass only
      add_part(p2, -2, 1, nullptr);  // the closing role (fold): one more block of the second pass
      p2.chain = chain;
      p2.words = rng.dev;
      rc = fill_fold(p2.closing);
      if (rc != PYG_HIP_OK) return rc;
      closed = true;
      hipLaunchKernelGGL(fused_last_apply_kernel, dim3((unsigned)p2.cum[p2.n - 1]), dim3(256), 0, stream, p2);
      PYG_HIP_CHECK(hipGetLastError());
    } else if (hop_onepass[(size_t)ell]) {
      add_part(p2, -1, 1, nullptr);  // carry block
      p2.ticket = ticket_of(1 + ell);
      rc = launch_scan(p2, kScanOnePass);
    } else {
      rc = launch_scan(p2, kScanReduce);
      if (rc != PYG_HIP_OK) return rc;
      add_part(p2, -1, 1, nullptr);  // carry block: apply pass only
      rc = launch_scan(p2, kScanApply);
    }
    pt.mark("hop_queued");
    if (rc != PYG_HIP_OK) return rc;
    steps_by_hop[(size_t)ell] = cur;
    prev_steps = cur;
    if (last_hop) prev_steps.clear();   // their `col` is written: nothing left for the finalize role of the closing launch
  }
  if (!closed) {
    FSampleLaunch fin;
    ::memset(&fin, 0, sizeof(fin));
    fin.tb = tb;
    fin.chain = chain;
    int nf = 0;
    for (const Step& s : prev_steps) {
      fin.f[nf] = s.fin;
      add_sample(fin, kRoleFinalize, (s.Eb + 255) / 256, nf++);
    }
    add_sample(fin, kRoleFold, 1, 0);
    int rc = fill_fold(fin.fold);
    if (rc != PYG_HIP_OK) return rc;
    rc = launch_sample(fin);
    if (rc != PYG_HIP_OK) return rc;
  }
  if (rng_late) {  // a carried stream that covers this call: its next round (for the NEXT call) goes out now, off the critical path
    int rc = rng_topup_deferred(c, rng);
    if (rc != PYG_HIP_OK) return rc;
  }
  pt.lap(4);
  pt.mark("all_queued");
  {
    // The closing launch writes `done_seq` behind the tables and the engine hand-back (pinned memory): seen here ~2 us after
    // it is written, where the wake-up out of hipStreamSynchronize takes 6 - 8.  Everything the call queued lies in front of
    // that launch on the stream; scratch goes back to a stream-ordered allocator.  A poll that outlasts 5 ms (a long
    // queue in front of this call, a fault) falls back to the synchronisation, which also surfaces the error.
    const auto t_poll = std::chrono::steady_clock::now();
    bool seen = false;
    unsigned spins = 0;
    while (!seen) {
      seen = __atomic_load_n(done_word, __ATOMIC_ACQUIRE) == done_seq;
      if (!seen && (++spins & 1023u) == 0 &&
          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_poll).count() > 5.0)
        break;
    }
    if (!seen) PYG_HIP_CHECK(hipStreamSynchronize(stream));
    c.main_idle = true;
  }
  pt.lap(5);
  pt.mark("synced");
  if (err_flag)
    PYG_HIP_REQUIRE(*static_cast<volatile int*>(err_flag) == 0, "Found invalid non-sorted temporal neighborhood");

  // ---- fold the tables into the host's bookkeeping ----
  const int64_t* h_size = reinterpret_cast<const int64_t*>(tables_host);
  const CountAgg* h_tot = reinterpret_cast<const CountAgg*>(tables_host + 8 * (size_t)(L + 1) * T + 8 * (size_t)T);
  const int32_t* h_over = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(h_tot) + sizeof(CountAgg) * (size_t)L * R);
  if (h_over[(size_t)L * R]) return kNeedQueued;  // a sampled row of degree >= 2^16: round 2's chain carries the wide tables
  for (int k = 0; k < L * R; ++k)
    if (h_over[k]) return kNeedSlow;
  for (int ell = 0; ell < L; ++ell) {
    for (int e = 0; e < R; ++e) rs[(size_t)e].edges_per_hop.push_back(0);
    for (const Step& s : steps_by_hop[(size_t)ell]) {
      const CountAgg t = h_tot[(size_t)ell * R + s.e];
      RelState& st = rs[(size_t)s.e];
      if (t.edges > 0) {
        const int64_t end_word = rng.word + tab_dw(t.tab, rng.units);
        rng.blocks = std::max(rng.blocks, end_word / 128 + 1);
        rng.word = end_word;
        rng.units = tab_nb(t.tab, rng.units);
      }
      st.row.size += t.edges;
      st.col.size += t.edges;
      st.eid.size += t.edges;
      st.edges_per_hop.back() = t.edges;
    }
    for (int t = 0; t < T; ++t) {
      const int64_t now = h_size[(size_t)(ell + 1) * T + t], before = h_size[(size_t)ell * T + t];
      nodes_per_hop[(size_t)t].push_back(now - before);
      ns[(size_t)t].nodes.size = now;
      if (disjoint) ns[(size_t)t].batch.size = now;
    }
  }
  return PYG_HIP_OK;
}
