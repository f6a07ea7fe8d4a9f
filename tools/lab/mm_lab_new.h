// candidate designs (included by mm_lab.hip after Ctx / bench)

// ------------------------------------------------------------------------------------------------
// v3: NWV waves per workgroup (WG tile = NWV*32 rows), cyclic WG-tile schedule (tile b, b + G, ...: the
//     chip sweeps one narrow window of X / out), W kept in LDS in its NATIVE [K][M] layout: copied by
//     LDS-DMA (global_load_lds_dwordx4, chunk order permuted inside every 1 KiB block so the transposing
//     reads are bank-conflict free), double buffered (the next relation's W is in flight while this one is
//     multiplied, one barrier per relation change), A fragments through ds_read_b64_tr_b16.
// ------------------------------------------------------------------------------------------------
template <int FLAGS, int NWV, int DBG, int WDB = 1, int SCHED = 0>
__global__ __launch_bounds__(NWV * 64) void v3_kernel(const DevGroup* __restrict__ descs, const int32_t* __restrict__ tile_start,
                                                      int B) {
  constexpr bool NT_LOAD = (FLAGS & 1) != 0;
  constexpr bool NT_STORE = (FLAGS & 2) != 0;
  constexpr int NT = 4, NI = 8, NO = 8, CPR = 16;
  constexpr int BM = NWV * 32;
  constexpr int WB = K * MC * 2;  // 32 KB per W buffer
  constexpr int BLK_PER_WAVE = (K / 4) / NWV;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 31, h = lane >> 5;
  const int bx = blockIdx.x, G = gridDim.x;
  char* stage = smem + (WDB ? 2 : 1) * WB + wave * 8192;
  const int total = tile_start[B];
  int nloc, cbase = 0, tstep = 1;
  if (SCHED == 0) {
    if (bx >= total) return;
    nloc = (total - 1 - bx) / G + 1;
    cbase = bx;
    tstep = G;
  } else if (SCHED == 1) {
    cbase = (int)((int64_t)bx * total / G);
    nloc = (int)((int64_t)(bx + 1) * total / G) - cbase;
    if (nloc <= 0) return;
  } else {
    // banded: workgroups of XCD k (ids k, k + 8, ...) sweep band k of the tiles cyclically
    const int nb = SCHED == 2 ? 8 : SCHED;            // number of bands
    const int band = bx % nb, w = bx / nb, per = G / nb;   // G must be a multiple of nb
    const int b0 = (int)((int64_t)band * total / nb), b1 = (int)((int64_t)(band + 1) * total / nb);
    cbase = b0 + w;
    tstep = per;
    if (cbase >= b1) return;
    nloc = (b1 - 1 - cbase) / per + 1;
  }
  auto tile_of = [&](int j) -> int { return cbase + j * tstep; };

  // group of the first tile
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_start[mid] <= tile_of(0)) lo = mid; else hi = mid;
  }
  int g = lo;  // group of the tile being prefetched

  // --- W buffers -----------------------------------------------------------------------------------
  // DMA side: lane i of a block's instruction fills LDS position i of the 1 KiB block
  const int dma_r = (lane & 15) >> 2, dma_ii = lane & 3, dma_u = lane >> 4;
  const int dma_c = 2 * dma_u + (dma_ii & 1) + 8 * (dma_ii >> 1);
  const int dma_src_off = dma_r * 256 + dma_c * 16;
  auto issue_w = [&](int grp_id, int buf) {
    const char* w = descs[grp_id].w;
#pragma unroll
    for (int j = 0; j < BLK_PER_WAVE; ++j) {
      const int kb = wave * BLK_PER_WAVE + j;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + kb * 1024 + dma_src_off),
                                       (LDSV*)(smem + buf * WB + kb * 1024), 16, 0, 0);
    }
  };
  // first tile of this workgroup's sequence in a group after `gc` (-1: none), and its group
  auto next_group = [&](int gc) -> int {
    const int ts = tile_start[gc + 1];
    if (ts >= total) return -1;
    const int j = ts > cbase ? (ts - cbase + tstep - 1) / tstep : 0;
    if (j >= nloc) return -1;
    const int t = cbase + j * tstep;
    int gg = gc + 1;
    while (tile_start[gg + 1] <= t) ++gg;
    return gg;
  };
  // reader side
  const int q = lane & 15, grp16 = lane >> 4;
  const int a_lane_off = 16384 * h + (4 * (q >> 2) + (grp16 & 1) + 2 * (q & 1)) * 16 + ((q & 3) >> 1) * 8;

  u32x4 xr[NI];
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
  int wnext = -1;
  if (WDB) {
    wnext = next_group(wcur);
    if (wnext >= 0) issue_w(wnext, 1);
  }
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
      if (!(DBG & 1)) {
#pragma unroll
        for (int s = 0; s < NI; ++s) {
          const u32x4 xa = *reinterpret_cast<const u32x4*>(stage + (x * CPR + ((NI * h + s) ^ (x & 15))) * 16);
          bf16x8 wa[NT];
#pragma unroll
          for (int tt = 0; tt < NT; ++tt) {
            const v4i16 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(wb + (2 * s) * 1024 + tt * 256));
            const v4i16 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(wb + (2 * s + 1) * 1024 + tt * 256));
            wa[tt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
          }
#pragma unroll
          for (int tt = 0; tt < NT; ++tt)
            acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[tt], __builtin_bit_cast(bf16x8, xa), acc[tt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[tt][r];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = NO * h + 2 * tt + j;
          *reinterpret_cast<u32x4*>(stage + (x * 16 + (c ^ (x & 15))) * 16) = pack8(v + 8 * j);
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
        // relation change: everything this wave has in flight (its part of the next W included) is older than
        // the X tile it has just staged
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        wcur = cg;
        if (WDB) {
          wbuf ^= 1;
          wnext = next_group(wcur);
          if (wnext >= 0) issue_w(wnext, wbuf ^ 1);
        } else {
          issue_w(wcur, 0);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        }
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


// ------------------------------------------------------------------------------------------------
// copyq: persistent pipelined copy with the features of the base kernel added one at a time
//   MODE 0  flat: workgroup b copies WG-tiles [b*T/G, (b+1)*T/G) of a flat buffer (hbm_copy's pat1 pipe1)
//   MODE 1  segment aware: tile walk over descs / tile_start, clamped rows, per-row store predicates
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void copyq_kernel(const DevGroup* __restrict__ descs, const int32_t* __restrict__ tile_start, int B,
                                                    const u32x4* __restrict__ in, u32x4* __restrict__ out, long flat_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bx = blockIdx.x, G = gridDim.x;
  if (smem && tid == 100000) smem[0] = 1;
  if (MODE == 0) {
    const long b0 = (long)bx * flat_tiles / G, b1 = (long)(bx + 1) * flat_tiles / G;
    u32x4 v[8], nx[8];
    if (b0 < b1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) nx[q] = __builtin_nontemporal_load((const GU32x4*)(in + ((b0 * 4 + wave) * 8 + q) * 64 + lane));
    }
    for (long t = b0; t < b1; ++t) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = nx[q];
      if (t + 1 < b1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) nx[q] = __builtin_nontemporal_load((const GU32x4*)(in + (((t + 1) * 4 + wave) * 8 + q) * 64 + lane));
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(out + ((t * 4 + wave) * 8 + q) * 64 + lane));
    }
    return;
  }
  const int total = tile_start[B];
  const int cbase = (int)((int64_t)bx * total / G);
  const int nloc = (int)((int64_t)(bx + 1) * total / G) - cbase;
  if (nloc <= 0) return;
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_start[mid] <= cbase) lo = mid; else hi = mid;
  }
  int g = lo;
  u32x4 xr[8];
  DevGroup dn = descs[g];
  int64_t n_row0 = 0, n_rows = 0;
  bool n_valid = false;
  auto prefetch = [&](int ti) {
    const int t = cbase + ti;
    while (t >= tile_start[g + 1]) {
      ++g;
      dn = descs[g];
    }
    n_rows = dn.rows;
    n_row0 = (int64_t)(t - tile_start[g]) * 128 + wave * 32;
    n_valid = n_row0 < n_rows;
    if (n_valid) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = i * 64 + lane;
        const int r = p / 16, cs = p % 16;
        int64_t row = n_row0 + r;
        if (row >= n_rows) row = n_rows - 1;
        xr[i] = __builtin_nontemporal_load((const GU32x4*)(dn.a + row * 256 + cs * 16));
      }
    }
  };
  prefetch(0);
  for (int t = 0; t < nloc; ++t) {
    u32x4 ov[8];
    const DevGroup d_out = dn;
    const int64_t row0_out = n_row0, rows_out = n_rows;
    const bool valid_out = n_valid;
#pragma unroll
    for (int i = 0; i < 8; ++i) ov[i] = xr[i];
    if (t + 1 < nloc) prefetch(t + 1);
    if (valid_out) {
      char* obase = d_out.c + (row0_out * 128) * 2;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = i * 64 + lane;
        const int r = p / 16, cs = p % 16;
        if (row0_out + r < rows_out) __builtin_nontemporal_store(ov[i], (GU32x4*)(obase + (int64_t)r * 256 + cs * 16));
      }
    }
  }
}


// plain cyclic read / write / copy over named buffers (is one of the buffers "slow memory"?)
template <int MODE>  // 0 read, 1 write, 2 copy
__global__ __launch_bounds__(256) void rw_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long W = (long)gridDim.x * 4;
  u32x4 acc = {0, 0, 0, 0};
  for (long t = (long)blockIdx.x * 4 + wave; t < ntiles; t += W) {
    u32x4 v[8];
    if (MODE != 1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = __builtin_nontemporal_load((const GU32x4*)(in + (t * 8 + q) * 64 + lane));
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = u32x4{(uint32_t)lane, 1, 2, 3};
    }
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) acc ^= v[q];
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(out + (t * 8 + q) * 64 + lane));
    }
  }
  if (MODE == 0 && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[0] = acc;
}


// non-persistent copy, one U KiB tile per wave, occupancy limited through the dynamic LDS size
template <int U>
__global__ __launch_bounds__(256) void bigc_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 100000) smem[0] = 1;
  const long t = (long)blockIdx.x * 4 + wave;
  if (t >= ntiles) return;
  u32x4 v[U];
#pragma unroll
  for (int q = 0; q < U; ++q) v[q] = __builtin_nontemporal_load((const GU32x4*)(in + (t * U + q) * 64 + lane));
#pragma unroll
  for (int q = 0; q < U; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(out + (t * U + q) * 64 + lane));
}

// persistent copy with DYNAMIC tile assignment: every wave pulls its next 8 KiB tile from one of 8 counters
// (counter = workgroup id % 8, i.e. per XCD; tile = 8 * ticket + counter), one tile ahead of the copy
__global__ __launch_bounds__(256) void dync_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long ntiles,
                                                   unsigned int* __restrict__ ctr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  if (threadIdx.x == 100000) smem[0] = 1;
  const int q8 = blockIdx.x & 7;
  auto pull = [&]() -> long {
    unsigned int k = 0;
    if (lane == 0) k = atomicAdd(&ctr[q8 * 32], 1u);
    k = __builtin_amdgcn_readfirstlane(k);
    return (long)k * 8 + q8;
  };
  u32x4 v[8], nx[8];
  long t = pull();
  if (t < ntiles) {
#pragma unroll
    for (int q = 0; q < 8; ++q) nx[q] = __builtin_nontemporal_load((const GU32x4*)(in + (t * 8 + q) * 64 + lane));
  }
  while (t < ntiles) {
    const long t2 = pull();
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = nx[q];
    if (t2 < ntiles) {
#pragma unroll
      for (int q = 0; q < 8; ++q) nx[q] = __builtin_nontemporal_load((const GU32x4*)(in + (t2 * 8 + q) * 64 + lane));
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(out + (t * 8 + q) * 64 + lane));
    t = t2;
  }
}


// persistent pipelined copy whose loads are inline asm (invisible to the compiler's waitcnt pass) and whose wait
// is an explicit vmcnt(8): the 8 stores issued after the loads stay in flight.  SCHED 0: contiguous WG ranges,
// 1: cyclic wave tiles.
template <int SCHED, int WAITN, int ILV = 0, int TLBPF = 0>
__global__ __launch_bounds__(256) void copyv_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long flat_tiles) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bx = blockIdx.x, G = gridDim.x;
  long j0, j1, step;  // wave tile index sequence j0, j0 + step, ... < j1 (8 KiB wave tiles)
  const long wtiles = flat_tiles * 4;
  if (SCHED == 0) {
    const long b0 = (long)bx * flat_tiles / G, b1 = (long)(bx + 1) * flat_tiles / G;
    j0 = b0 * 4 + wave; j1 = b1 * 4; step = 4;
  } else {
    j0 = (long)bx * 4 + wave; j1 = wtiles; step = (long)G * 4;
  }
  u32x4 nx[8];
  auto issue = [&](long j) {
    const u32x4* p = in + j * 512 + lane;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(nx[q]) : "v"(p + q * 64) : "memory");
  };
  if (j0 < j1) issue(j0);
  for (long j = j0; j < j1; j += step) {
    u32x4 v[8];
    if (WAITN == 8 && j != j0)  // steady state: 8 loads, then the previous tile's 8 stores
      asm volatile("s_waitcnt vmcnt(8)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]), "+v"(nx[4]), "+v"(nx[5]), "+v"(nx[6]), "+v"(nx[7])::"memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(nx[0]), "+v"(nx[1]), "+v"(nx[2]), "+v"(nx[3]), "+v"(nx[4]), "+v"(nx[5]), "+v"(nx[6]), "+v"(nx[7])::"memory");
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = nx[q];
    u32x4* o = out + j * 512 + lane;
    if (TLBPF) {
      // translation prefetch: one dword from the X tile TLBPF rounds ahead and from the out tile of the next round
      const long ja = j + (long)TLBPF * step, jo = j + step;
      uint32_t dummy;
      if (ja < j1) asm volatile("global_load_dword %0, %1, off" : "=v"(dummy) : "v"((const char*)in + ja * 8192) : "memory");
      if (jo < j1) asm volatile("global_load_dword %0, %1, off" : "=v"(dummy) : "v"((const char*)out + jo * 8192) : "memory");
    }
    if (ILV && j + step < j1) {
      // loads of the next tile and stores of this one alternate, 1 KiB each
      const u32x4* p = in + (j + step) * 512 + lane;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(nx[q]) : "v"(p + q * 64) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(o + q * 64), "v"(v[q]) : "memory");
      }
    } else {
      if (j + step < j1) issue(j + step);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(o + q * 64), "v"(v[q]) : "memory");
    }
  }
}


// memory-behaviour proxy of a NON-persistent matmul: one workgroup per NW*RPW-row tile; every wave DMAs its
// RPW rows of X (256 B each) and its share of a 32 KB weight (L2 resident) into LDS, barrier, reads X back and
// stores it as the "output".  No MFMAs: what would the memory system give such a kernel?
template <int NW, int RPW, int WLOAD>
__global__ __launch_bounds__(NW * 64) void proxy_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, const char* __restrict__ w,
                                                        long ntiles, int xg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XI = RPW * 256 / 1024;        // 1 KiB DMA instructions per wave for X
  constexpr int WI = 32768 / (NW * 1024);     // ... for the weight share
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long bk = blockIdx.x & 7, bj = blockIdx.x >> 3;
  const long t = xg > 1 ? (bj / xg) * (8L * xg) + bk * xg + bj % xg : (long)blockIdx.x;
  if (t >= ntiles) return;
  char* xs = smem + 32768 + wave * (RPW * 256);
  const char* src = (const char*)in + (t * NW + wave) * (RPW * 256);
#pragma unroll
  for (int q = 0; q < XI; ++q)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 1024 + lane * 16), (LDSV*)(xs + q * 1024), 16, 0, 2);
  if (WLOAD) {
    const char* wsrc = w + ((t * 7) % 154) * 32768 + wave * (WI * 1024);
#pragma unroll
    for (int q = 0; q < WI; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + q * 1024 + lane * 16), (LDSV*)(smem + wave * (WI * 1024) + q * 1024), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  u32x4 v[XI];
#pragma unroll
  for (int q = 0; q < XI; ++q) v[q] = *reinterpret_cast<const u32x4*>(xs + q * 1024 + lane * 16);
  if (WLOAD) v[0][0] ^= *reinterpret_cast<const uint32_t*>(smem + ((lane * 516 + wave * 36) & 32764)) & 1u;
  char* dst = (char*)out + (t * NW + wave) * (RPW * 256);
#pragma unroll
  for (int q = 0; q < XI; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(dst + q * 1024 + lane * 16));
}


// ------------------------------------------------------------------------------------------------
// v4: NON-persistent.  One workgroup (4 waves x 32 rows) per 128-row tile, dispatched in tile order by the
//     hardware (the chip sweeps X / out linearly with a narrow window of tiles in flight).  X tile and the
//     relation's W (native layout) arrive by LDS-DMA, A fragments through ds_read_b64_tr_b16; 64 KB of LDS,
//     two workgroups per CU cover each other's load phase.
// ------------------------------------------------------------------------------------------------
struct TileRec {
  const char* a;   // first row of the tile
  const char* w;
  char* c;
  int32_t rows;    // valid rows in this tile (1..128)
  int32_t pad;
};

template <int FLAGS, int DBG>
__global__ __launch_bounds__(256) void v4_kernel(const TileRec* __restrict__ recs, int ntiles) {
  constexpr bool NT_STORE = (FLAGS & 2) != 0;
  constexpr int XAUX = (FLAGS & 1) ? 2 : 0;
  constexpr int NT = 4, NI = 8, NO = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 31, h = lane >> 5;
  const TileRec rec = recs[blockIdx.x];
  const int row0 = wave * 32;
  const bool valid = row0 < rec.rows;
  char* xs = smem + 32768 + wave * 8192;
  // W: this wave's 8 blocks (4 k-rows each), chunk order permuted inside every block (see v3)
  {
    const int dma_r = (lane & 15) >> 2, dma_ii = lane & 3, dma_u = lane >> 4;
    const int dma_c = 2 * dma_u + (dma_ii & 1) + 8 * (dma_ii >> 1);
    const char* wsrc = rec.w + dma_r * 256 + dma_c * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kb = wave * 8 + j;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + kb * 1024), (LDSV*)(smem + kb * 1024), 16, 0, 0);
    }
  }
  if (valid) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = i * 64 + lane;
      const int r = p >> 4, cs = p & 15;
      const int c = cs ^ (r & 15);
      int row = row0 + r;
      if (row >= rec.rows) row = rec.rows - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec.a + row * 256 + c * 16), (LDSV*)(xs + i * 1024), 16, 0, XAUX);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (!valid) return;
  const int q = lane & 15, grp16 = lane >> 4;
  const char* wb = smem + 16384 * h + (4 * (q >> 2) + (grp16 & 1) + 2 * (q & 1)) * 16 + ((q & 3) >> 1) * 8;
  f32x16 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  if (!(DBG & 1)) {
#pragma unroll
    for (int s = 0; s < NI; ++s) {
      const u32x4 xa = *reinterpret_cast<const u32x4*>(xs + (x * 16 + ((NI * h + s) ^ (x & 15))) * 16);
      bf16x8 wa[NT];
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const v4i16 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(wb + (2 * s) * 1024 + tt * 256));
        const v4i16 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(wb + (2 * s + 1) * 1024 + tt * 256));
        wa[tt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[tt], __builtin_bit_cast(bf16x8, xa), acc[tt], 0, 0, 0);
    }
  } else {
    acc[0][0] = __builtin_bit_cast(float, *reinterpret_cast<const uint32_t*>(xs + lane * 16));
  }
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[tt][r];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = NO * h + 2 * tt + j;
      *reinterpret_cast<u32x4*>(xs + (x * 16 + (c ^ (x & 15))) * 16) = pack8(v + 8 * j);
    }
  }
  u32x4 ov[NO];
#pragma unroll
  for (int i = 0; i < NO; ++i) ov[i] = *reinterpret_cast<const u32x4*>(xs + (i * 64 + lane) * 16);
  char* obase = rec.c + row0 * 256;
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int p = i * 64 + lane;
    const int r = p >> 4, cs = p & 15;
    const int c = cs ^ (r & 15);
    if (row0 + r < rec.rows) {
      GU32x4* dst = (GU32x4*)(obase + r * 256 + c * 16);
      if (NT_STORE) __builtin_nontemporal_store(ov[i], dst); else *dst = ov[i];
    }
  }
}


// ------------------------------------------------------------------------------------------------
// v5: v3 with SIXTEEN waves of 16 rows (1024 threads, 256-row workgroup tiles, one workgroup per CU):
//     twice as many independent 4 KiB load / store streams per CU -- what the copy proxies say a badly placed
//     buffer needs -- v_mfma_f32_16x16x32_bf16, W double buffered + LDS-DMA + transposing reads as in v3.
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int FLAGS, int DBG>
__global__ __launch_bounds__(1024) void v5_kernel(const DevGroup* __restrict__ descs, const int32_t* __restrict__ tile_start, int B) {
  constexpr bool NT_LOAD = (FLAGS & 1) != 0;
  constexpr bool NT_STORE = (FLAGS & 2) != 0;
  constexpr int NWV = 16, BM = NWV * 16;
  constexpr int WB = K * MC * 2;
  constexpr int BLK_PER_WAVE = (K / 4) / NWV;  // 2
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 15, g4 = lane >> 4;
  const int bx = blockIdx.x, G = gridDim.x;
  char* stage = smem + 2 * WB + wave * 4096;
  const int total = tile_start[B];
  if (bx >= total) return;
  const int nloc = (total - 1 - bx) / G + 1;
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_start[mid] <= bx) lo = mid; else hi = mid;
  }
  int g = lo;
  const int pix = (x + 12) & 15;  // row permutation of the stage swizzle (see the service groups of ds_read_b128)
  // W DMA: LDS position i of a 1 KiB block (4 k-rows): line u = i >> 4, slot v = i & 15 holds row v >> 2, chunk 4 (v & 3) + u
  const int dma_src_off = ((lane & 15) >> 2) * 256 + (4 * (lane & 3) + (lane >> 4)) * 16;
  auto issue_w = [&](int grp_id, int buf) {
    const char* w = descs[grp_id].w;
#pragma unroll
    for (int j = 0; j < BLK_PER_WAVE; ++j) {
      const int kb = wave * BLK_PER_WAVE + j;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + kb * 1024 + dma_src_off),
                                       (LDSV*)(smem + buf * WB + kb * 1024), 16, 0, 0);
    }
  };
  auto next_group = [&](int gc) -> int {
    const int ts = tile_start[gc + 1];
    if (ts >= total) return -1;
    const int j = ts > bx ? (ts - bx + G - 1) / G : 0;
    const int t = bx + j * G;
    if (t >= total) return -1;
    int gg = gc + 1;
    while (tile_start[gg + 1] <= t) ++gg;
    return gg;
  };
  const int a_lane_off = 8192 * g4 + 16 * x;

  u32x4 xr[4];
  DevGroup dn = descs[g];
  int64_t n_row0 = 0, n_rows = 0;
  bool n_valid = false;
  auto prefetch = [&](int ti) {
    const int t = bx + ti * G;
    while (t >= tile_start[g + 1]) {
      ++g;
      dn = descs[g];
    }
    n_rows = dn.rows;
    n_row0 = (int64_t)(t - tile_start[g]) * BM + wave * 16;
    n_valid = n_row0 < n_rows;
    if (n_valid) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = i * 64 + lane;
        const int r = p >> 4, cs = p & 15;
        const int c = cs ^ ((r + 12) & 15);
        int64_t row = n_row0 + r;
        if (row >= n_rows) row = n_rows - 1;
        const GU32x4* src = (const GU32x4*)(dn.a + row * 256 + c * 16);
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
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(stage + (i * 64 + lane) * 16) = xr[i];
  }
  if (1 < nloc) prefetch(1);

  for (int t = 0; t < nloc; ++t) {
    u32x4 ov[4];
    if (valid) {
      f32x4 acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      const char* wb = smem + wbuf * WB + a_lane_off;
      if (!(DBG & 1)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const u32x4 xa = *reinterpret_cast<const u32x4*>(stage + (x * 16 + ((4 * g4 + s) ^ pix)) * 16);
#pragma unroll
          for (int cb = 0; cb < 8; ++cb) {
            const v4i16 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) v4i16*)(wb + 2048 * s + 256 * (cb >> 1) + 8 * (cb & 1)));
            const v4i16 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) v4i16*)(wb + 2048 * s + 1024 + 256 * (cb >> 1) + 8 * (cb & 1)));
            const bf16x8 wa = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, __builtin_bit_cast(bf16x8, xa), acc[cb], 0, 0, 0);
          }
        }
      } else {
        acc[0][0] = __builtin_bit_cast(float, *reinterpret_cast<const uint32_t*>(stage + lane * 16));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = acc[2 * j][i];
          v[4 + i] = acc[2 * j + 1][i];
        }
        *reinterpret_cast<u32x4*>(stage + (x * 16 + ((4 * g4 + j) ^ pix)) * 16) = pack8(v);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) ov[i] = *reinterpret_cast<const u32x4*>(stage + (i * 64 + lane) * 16);
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
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(stage + (i * 64 + lane) * 16) = xr[i];
      }
      if (cg != wcur) {
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
      char* obase = d_out.c + row0_out * 256;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = i * 64 + lane;
        const int r = p >> 4, cs = p & 15;
        const int c = cs ^ ((r + 12) & 15);
        if (row0_out + r < rows_out) {
          GU32x4* dst = (GU32x4*)(obase + (int64_t)r * 256 + c * 16);
          if (NT_STORE) __builtin_nontemporal_store(ov[i], dst); else *dst = ov[i];
        }
      }
    }
  }
}


// persistent copy, U KiB per wave tile: DYN 0 = static cyclic (wave w takes tiles w, w + W, ...), 1 = in-order tickets
// per wave (8 counters, one per XCD); DEPTH 1 = next tile's loads issued before this tile's stores, 0 = load, store, next.
template <int U, int DYN, int DEPTH>
__global__ __launch_bounds__(256) void pcopy_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long ntiles,
                                                    unsigned int* __restrict__ ctr, int slp, int wait) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 100000) smem[0] = 1;
  const int q8 = blockIdx.x & 7;
  const long W = (long)gridDim.x * 4;
  long stat = (long)blockIdx.x * 4 + wave;
  auto pull = [&]() -> long {
    if (!DYN) {
      const long t = stat;
      stat += W;
      return t;
    }
    unsigned int k = 0;
    if (lane == 0) k = atomicAdd(&ctr[q8 * 32], 1u);
    k = __builtin_amdgcn_readfirstlane(k);
    return (long)k * 8 + q8;
  };
  u32x4 v[U], nx[U];
  long t = pull();
  if (DEPTH) {
    if (t < ntiles) {
#pragma unroll
      for (int q = 0; q < U; ++q) nx[q] = __builtin_nontemporal_load((const GU32x4*)(in + (t * U + q) * 64 + lane));
    }
    while (t < ntiles) {
      const long t2 = pull();
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = nx[q];
      if (t2 < ntiles) {
#pragma unroll
        for (int q = 0; q < U; ++q) nx[q] = __builtin_nontemporal_load((const GU32x4*)(in + (t2 * U + q) * 64 + lane));
      }
#pragma unroll
      for (int q = 0; q < U; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(out + (t * U + q) * 64 + lane));
      for (int k = 0; k < slp; ++k) __builtin_amdgcn_s_sleep(4);
      t = t2;
    }
  } else {
    while (t < ntiles) {
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = __builtin_nontemporal_load((const GU32x4*)(in + (t * U + q) * 64 + lane));
#pragma unroll
      for (int q = 0; q < U; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(out + (t * U + q) * 64 + lane));
      if (wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int k = 0; k < slp; ++k) __builtin_amdgcn_s_sleep(4);
      t = pull();
    }
  }
}


// persistent copy with WORKGROUP tickets: a WG of NW waves takes 64 KiB * (NW*U/64) tiles in address order from a per-XCD
// counter; the first wave to reach step i pulls the ticket and publishes it through LDS (no barrier).
template <int U, int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void pcopy2_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long ntiles,
                                                         unsigned int* __restrict__ ctr, int dynamic, int cg, int wait, int early) {
  const int perm = wait >> 4;
  wait &= 15;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* seq = (int*)smem;  // 2048 entries
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2048; i += NW * 64) seq[i] = -1;
  __syncthreads();
  const int q8 = blockIdx.x & 7;
  int step = 0;
  long stat = blockIdx.x >> 3;
  auto pull = [&]() -> long {
    if (!dynamic) {
      const long sq = stat;
      stat += gridDim.x >> 3;
      return ((sq / cg) * 8 + q8) * cg + sq % cg;
    }
    int got = 0;
    if (lane == 0) {
      int v = atomicCAS(&seq[step], -1, -2);
      if (v == -1) {
        v = (int)atomicAdd(&ctr[q8 * 32], 1u);
        __hip_atomic_store(&seq[step], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        while (v < 0) v = __hip_atomic_load(&seq[step], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      got = v;
    }
    ++step;
    got = __builtin_amdgcn_readfirstlane(got);
    return (((long)got / cg) * 8 + q8) * cg + got % cg;
  };
  u32x4 v[U], nx[U];
  long t = pull();
  // perm: 16-byte chunk cs of row r goes to / comes from chunk cs ^ (r & 15) (the MFMA kernels' stage swizzle); row = 4 q + lane / 16
  auto base = [&](long tile) { return (tile * NW + wave) * U * 64 + (perm ? 0 : lane); };
  auto pl = [&](int q) { return perm ? ((lane & ~15) | ((lane & 15) ^ ((4 * q + (lane >> 4)) & 15))) : 0; };
  if (DEPTH) {
    if (t < ntiles) {
#pragma unroll
      for (int q = 0; q < U; ++q) nx[q] = __builtin_nontemporal_load((const GU32x4*)(in + base(t) + q * 64 + pl(q)));
    }
    while (t < ntiles) {
      const long t2 = pull();
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = nx[q];
      if (t2 < ntiles) {
#pragma unroll
        for (int q = 0; q < U; ++q) nx[q] = __builtin_nontemporal_load((const GU32x4*)(in + base(t2) + q * 64 + pl(q)));
      }
#pragma unroll
      for (int q = 0; q < U; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(out + base(t) + q * 64 + pl(q)));
      t = t2;
    }
  } else {
    while (t < ntiles) {
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = __builtin_nontemporal_load((const GU32x4*)(in + base(t) + q * 64 + pl(q)));
      long t2 = 0;
      if (early) t2 = pull();
#pragma unroll
      for (int q = 0; q < U; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(out + base(t) + q * 64 + pl(q)));
      if (wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t = early ? t2 : pull();
    }
  }
}


// bigc morphing towards persistence: every wave copies `rep` tiles one after the other; mode 0 = its tiles are adjacent,
// 1 = strided by the whole grid; wait = drain the stores before the next loads; slp = s_sleep units between tiles.
template <int U>
__global__ __launch_bounds__(256) void bigr_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long ntiles, int rep,
                                                   int mode, int wait, int slp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 100000) smem[0] = 1;
  const long w = (long)blockIdx.x * 4 + wave, W = (long)gridDim.x * 4;
  for (int r = 0; r < rep; ++r) {
    const long t = mode == 0 ? w * rep + r : (long)r * W + w;
    if (t >= ntiles) return;
    u32x4 v[U];
    const int op = slp >> 8;
    if (op == 1) {  // read only
      u32x4 acc = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = __builtin_nontemporal_load((const GU32x4*)(in + (t * U + q) * 64 + lane));
#pragma unroll
      for (int q = 0; q < U; ++q) acc ^= v[q];
      if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) out[t] = acc;
    } else if (op == 2) {  // write only
#pragma unroll
      for (int q = 0; q < U; ++q) {
        u32x4 z = {(unsigned)t, (unsigned)q, (unsigned)lane, 0u};
        __builtin_nontemporal_store(z, (GU32x4*)(out + (t * U + q) * 64 + lane));
      }
    } else {
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = __builtin_nontemporal_load((const GU32x4*)(in + (t * U + q) * 64 + lane));
#pragma unroll
      for (int q = 0; q < U; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(out + (t * U + q) * 64 + lane));
    }
    if (wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int k = 0; k < (slp & 255); ++k) __builtin_amdgcn_s_sleep(8);
  }
}


// XCD-interleave experiments.  bigx: non-persistent, one U KiB tile per wave, but XCD k (= blockIdx % 8) gets xg consecutive
// workgroup spans (xg * 4 * U KiB contiguous) instead of one.  pcx: persistent static, the U KiB tiles are dealt round-robin
// over XCDs (tile t -> XCD t % 8) when fine = 1, else in workgroup spans (4 tiles per XCD turn).
template <int U>
__global__ __launch_bounds__(256) void bigx_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long ntiles, int xg, int op) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 100000) smem[0] = 1;
  const long b = blockIdx.x, k = b & 7, j = b >> 3;
  const long wg = (j / xg) * (8L * xg) + k * xg + (j % xg);
  const long t = wg * 4 + wave;
  if (t >= ntiles) return;
  u32x4 v[U];
  if (op != 2) {
#pragma unroll
    for (int q = 0; q < U; ++q) v[q] = __builtin_nontemporal_load((const GU32x4*)(in + (t * U + q) * 64 + lane));
  } else {
#pragma unroll
    for (int q = 0; q < U; ++q) v[q] = u32x4{(unsigned)t, (unsigned)q, (unsigned)lane, 0u};
  }
  if (op == 1) {
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < U; ++q) acc ^= v[q];
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) out[t] = acc;
  } else {
#pragma unroll
    for (int q = 0; q < U; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(out + (t * U + q) * 64 + lane));
  }
}

template <int U>
__global__ __launch_bounds__(256) void pcx_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long ntiles, int fine, int op) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 100000) smem[0] = 1;
  const long k = blockIdx.x & 7, j = blockIdx.x >> 3;
  const long wpx = (long)(gridDim.x >> 3) * 4;  // waves per XCD
  const long W = (long)gridDim.x * 4;
  for (long r = 0;; ++r) {
    const long t = fine ? (r * wpx + j * 4 + wave) * 8 + k : r * W + (long)blockIdx.x * 4 + wave;
    if (t >= ntiles) break;
    u32x4 v[U];
    if (op != 2) {
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = __builtin_nontemporal_load((const GU32x4*)(in + (t * U + q) * 64 + lane));
    } else {
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = u32x4{(unsigned)t, (unsigned)q, (unsigned)lane, 0u};
    }
    if (op == 1) {
      u32x4 acc = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < U; ++q) acc ^= v[q];
      if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) out[t] = acc;
    } else {
#pragma unroll
      for (int q = 0; q < U; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(out + (t * U + q) * 64 + lane));
    }
  }
}


// persistent static copy with workgroup tiles (NW waves x U KiB) and a chunked XCD deal: the XCD-local tile sequence s
// (dealt cyclically to the XCD's workgroups) maps to global tile ((s / cg) * 8 + xcd) * cg + s % cg.
template <int U, int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void pcg_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long ntiles, int cg, int op) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 100000) smem[0] = 1;
  const long k = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;
  auto tile_of = [&](long r) { const long sq = r * per + j; return ((sq / cg) * 8 + k) * cg + sq % cg; };
  auto base = [&](long tile) { return (tile * NW + wave) * U * 64 + lane; };
  u32x4 v[U], nx[U];
  long r = 0;
  long t = tile_of(r);
  if (DEPTH && op == 0) {
    if (t < ntiles) {
#pragma unroll
      for (int q = 0; q < U; ++q) nx[q] = __builtin_nontemporal_load((const GU32x4*)(in + base(t) + q * 64));
    }
    while (t < ntiles) {
      const long t2 = tile_of(++r);
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = nx[q];
      if (t2 < ntiles) {
#pragma unroll
        for (int q = 0; q < U; ++q) nx[q] = __builtin_nontemporal_load((const GU32x4*)(in + base(t2) + q * 64));
      }
#pragma unroll
      for (int q = 0; q < U; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(out + base(t) + q * 64));
      t = t2;
    }
  } else {
    while (t < ntiles) {
      if (op != 2) {
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = __builtin_nontemporal_load((const GU32x4*)(in + base(t) + q * 64));
      } else {
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = u32x4{(unsigned)t, (unsigned)q, (unsigned)lane, 0u};
      }
      if (op == 1) {
        u32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < U; ++q) acc ^= v[q];
        if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) out[t] = acc;
      } else {
#pragma unroll
        for (int q = 0; q < U; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(out + base(t) + q * 64));
      }
      t = tile_of(++r);
    }
  }
}


// wave-independent persistent copy: every wave (its own 64-thread workgroup) pulls tickets from its XCD's counter; a ticket
// is REP consecutive 8 KiB tiles, copied one after the other with the next tile's loads in flight (DEPTH 1) or not (0);
// ticket s of XCD k covers tiles (((s / cg) * 8 + k) * cg + s % cg) * REP ...; the next ticket is pulled one ticket early.
template <int DEPTH>
__global__ __launch_bounds__(64) void pcw_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long ntiles,
                                                  unsigned int* __restrict__ ctr, int rep, int cg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int U = 8;
  const int lane = threadIdx.x;
  if (threadIdx.x == 100000) smem[0] = 1;
  const int k = blockIdx.x & 7;
  auto pull = [&]() -> long {
    unsigned int s = 0;
    if (lane == 0) s = atomicAdd(&ctr[k * 32], 1u);
    s = __builtin_amdgcn_readfirstlane(s);
    return (((long)(s / cg) * 8 + k) * cg + s % cg) * rep;
  };
  long cur = pull(), pend = pull();
  int r = 0;
  auto advance = [&]() -> long {  // next tile index (may be >= ntiles: the caller stops there)
    if (++r < rep) return cur + r;
    cur = pend;
    r = 0;
    if (cur < ntiles) pend = pull();
    return cur;
  };
  u32x4 v[U], nx[U];
  long t = cur;
  if (DEPTH) {
    if (t < ntiles) {
#pragma unroll
      for (int q = 0; q < U; ++q) nx[q] = __builtin_nontemporal_load((const GU32x4*)(in + (t * U + q) * 64 + lane));
    }
    while (t < ntiles) {
      const long t2 = advance();
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = nx[q];
      if (t2 < ntiles) {
#pragma unroll
        for (int q = 0; q < U; ++q) nx[q] = __builtin_nontemporal_load((const GU32x4*)(in + (t2 * U + q) * 64 + lane));
      }
#pragma unroll
      for (int q = 0; q < U; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(out + (t * U + q) * 64 + lane));
      t = t2;
    }
  } else {
    while (t < ntiles) {
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = __builtin_nontemporal_load((const GU32x4*)(in + (t * U + q) * 64 + lane));
#pragma unroll
      for (int q = 0; q < U; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(out + (t * U + q) * 64 + lane));
      t = advance();
    }
  }
}


// ------------------------------------------------------------------------------------------------
// v6: wave-pair workgroups (2 waves x 32 rows = 64-row tiles, three per CU), the relation's W held in REGISTERS (the 32
//     MFMA A fragments of a wave = 128 VGPRs, refilled through a 16 KiB LDS staging area on a relation change), X by
//     LDS-DMA into two 8 KiB stages per wave (next tile in flight while this one is multiplied), tiles handed out IN
//     ADDRESS ORDER by per-XCD ticket counters with the XCD deal in chunks of cg tiles
//     (tile = ((s / cg) * 8 + xcd) * cg + s % cg).  The first wave of the pair to need ticket i pulls it and publishes
//     it through a small LDS ring.
// ------------------------------------------------------------------------------------------------
template <int FLAGS, int DBG>
__global__ __launch_bounds__(128, 2) void v6_kernel(const DevGroup* __restrict__ descs, const int32_t* __restrict__ tile_start,
                                                    int B, unsigned int* __restrict__ ctr, int cg) {
  constexpr bool NT_STORE = (FLAGS & 2) != 0;
  constexpr int XAUX = (FLAGS & 1) ? 2 : 0;
  constexpr int NT = 4, NI = 8, NO = 8, R = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // lane-derived values are re-derived where they are used (two VALU ops) instead of living in VGPRs across the whole
  // kernel: with 128 registers of W the allocator otherwise spills them and reloads them behind an s_waitcnt vmcnt(0)
  auto lane_now = [&]() -> int {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  char* wst = smem;
  char* xs0 = smem + 16384 + wave * 16384;
  int* claim = (int*)(smem + 49152);
  int* ready = claim + R;
  int* val = ready + R;
  int* prog = val + R;
  for (int i = tid; i < 3 * R + 2; i += 128) claim[i] = 0;
  __syncthreads();
  const int k8 = blockIdx.x & 7;
  const int total = tile_start[B];
  int step = 0;
  auto pull = [&]() -> int {
    int v = 0;
    if (lane_now() == 0) {
      const int slot = step & (R - 1), lap = step / R;
      while (min(__hip_atomic_load(&prog[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP),
                 __hip_atomic_load(&prog[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) + R <= step)
        __builtin_amdgcn_s_sleep(1);
      const int old = atomicCAS(&claim[slot], lap, lap + 1);
      if (old == lap) {
        v = (int)atomicAdd(&ctr[k8 * 32], 1u);
        __hip_atomic_store(&val[slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&ready[slot], lap + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        while (__hip_atomic_load(&ready[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != lap + 1) __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(&val[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __hip_atomic_store(&prog[wave], step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    ++step;
    v = __builtin_amdgcn_readfirstlane(v);
    return ((v / cg) * 8 + k8) * cg + v % cg;
  };

  // W fragments: wreg[s][tt] = A operand of k-step s, column tile tt (see v3 for the LDS layout they are read from)
  bf16x8 wreg[NI][NT];
  auto load_w = [&](const char* w) {
    const int lane = lane_now(), h = lane >> 5;
    const int dma_r = (lane & 15) >> 2, dma_ii = lane & 3, dma_u = lane >> 4;
    const int dma_c = 2 * dma_u + (dma_ii & 1) + 8 * (dma_ii >> 1);
    const int dma_src_off = dma_r * 256 + dma_c * 16;
    const int q = lane & 15, grp16 = lane >> 4;
    const int a_lane_off = 8192 * h + (4 * (q >> 2) + (grp16 & 1) + 2 * (q & 1)) * 16 + ((q & 3) >> 1) * 8;
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      __syncthreads();
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
          const v4i16 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(wst + a_lane_off + (2 * s4) * 1024 + tt * 256));
          const v4i16 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(wst + a_lane_off + (2 * s4 + 1) * 1024 + tt * 256));
          wreg[r2 * 4 + s4][tt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
      }
    }
  };
  // per-lane pieces of the X / out addressing (16-byte chunk cs of row r sits at chunk cs ^ (r & 15) of the LDS row)
  auto issue_x = [&](const DevGroup& dg, int64_t row0, int buf) {
    char* xs = xs0 + buf * 8192;
    const char* base = dg.a + row0 * 256;                       // wave-uniform
    const int64_t left = dg.rows - row0;
    const int last = left < 32 ? (int)left - 1 : 31;            // wave-uniform
    const int l = lane_now();
    const int l4 = l >> 4, c0 = (l & 15) ^ l4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int r = 4 * i + l4;
      const int c = c0 ^ (4 * (i & 3));
      r = r > last ? last : r;
      const uint32_t off = (uint32_t)(r * 256 + c * 16);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off), (LDSV*)(xs + i * 1024), 16, 0, XAUX);
    }
  };

  int t_cur = pull();
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
  DevGroup d = descs[g];
  int64_t row0 = (int64_t)(t_cur - tile_start[g]) * 64 + wave * 32;
  bool valid = row0 < d.rows;
  if (valid) issue_x(d, row0, 0);
  int wcur = -1, buf = 0;
  while (true) {
    const int t_next = pull();
    const bool more = t_next < total;
    int gn = g;
    DevGroup dn = d;
    int64_t n_row0 = 0;
    bool n_valid = false;
    if (more) {
      if (t_next >= tile_start[gn + 1]) {
        do ++gn; while (t_next >= tile_start[gn + 1]);
        dn = descs[gn];
      }
      n_row0 = (int64_t)(t_next - tile_start[gn]) * 64 + wave * 32;
      n_valid = n_row0 < dn.rows;
      if (n_valid) issue_x(dn, n_row0, buf ^ 1);
    }
    if (g != wcur) {
      load_w(d.w);
      wcur = g;
    } else if (n_valid) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (valid) {
      const char* stage = xs0 + buf * 8192;
      char* scratch = wst + wave * 8192;
      const int lc = lane_now();
      const int xo = lc & 31, h = lc >> 5;
      const int cb = (NI * h) ^ (xo & 15);  // LDS addresses below: one or two VALU ops each, recomputed per tile
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        if (!(DBG & 1)) {
          const char* xrow = stage + xo * 256;
          u32x4 xa = *reinterpret_cast<const u32x4*>(xrow + cb * 16);
#pragma unroll
          for (int s = 0; s < NI; ++s) {
            u32x4 xn = xa;
            if (s + 1 < NI) xn = *reinterpret_cast<const u32x4*>(xrow + (cb ^ (s + 1)) * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[s][2 * half + j], __builtin_bit_cast(bf16x8, xa), acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            xa = xn;
          }
        }
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
          const int tt = 2 * half + j2;
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[j2][r];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<u32x4*>(scratch + xo * 256 + (cb ^ (2 * tt + j)) * 16) = pack8(v + 8 * j);
          }
        }
      }
      char* obase = d.c + row0 * 256;                            // wave-uniform
      const int64_t left = d.rows - row0;
      const int nrow = left < 32 ? (int)left : 32;
      const int l = lane_now();
      const int l4 = l >> 4, c0 = (l & 15) ^ l4;
#pragma unroll
      for (int i = 0; i < NO; ++i) {
        const u32x4 ov = *reinterpret_cast<const u32x4*>(scratch + i * 1024 + l * 16);
        const int r = 4 * i + l4;
        const int c = c0 ^ (4 * (i & 3));
        if (r < nrow) {
          GU32x4* dst = (GU32x4*)(obase + (uint32_t)(r * 256 + c * 16));
          if (NT_STORE) __builtin_nontemporal_store(ov, dst); else *dst = ov;
        }
      }
    }
    if (!more) break;
    t_cur = t_next;
    g = gn;
    d = dn;
    row0 = n_row0;
    valid = n_valid;
    buf ^= 1;
  }
}


// ------------------------------------------------------------------------------------------------
// v7: v6 with the ticket two tiles ahead and exact vmcnt bookkeeping.  Ticket s is requested by wave (s & 1) of the
//     pair with an asynchronous global atomic (inline asm: the compiler must not wait for it), collected one iteration
//     later and handed to the partner through a 4-slot LDS ring (the two waves can never be more than two tickets
//     apart).  Per iteration a wave issues, in this order: [atomic(i+2)] [8 DMA of tile i+1] ... [8 stores of tile i];
//     every wait names exactly the number of YOUNGER operations it may leave in flight (vmcnt retires in order on
//     gfx9), so neither the stores of the previous tile nor the next tile's DMA are ever waited for.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wait_vm(int n) {  // steady state: 16 or 17 younger operations; anything else drains
  if (n == 17) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
  else if (n == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int FLAGS, int DBG>
__global__ __launch_bounds__(128, 2) void v7_kernel(const DevGroup* __restrict__ descs, const int32_t* __restrict__ tile_start,
                                                    int B, unsigned int* __restrict__ ctr, int cg) {
  constexpr bool NT_STORE = (FLAGS & 2) != 0;
  constexpr int XAUX = (FLAGS & 1) ? 2 : 0;
  constexpr int NT = 4, NI = 8, NO = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  auto lane_now = [&]() -> int {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  char* wst = smem;
  char* xs0 = smem + 16384 + wave * 16384;
  int* ring_val = (int*)(smem + 49152);  // [4]
  int* ring_gen = ring_val + 4;          // [4]
  if (threadIdx.x < 8) ring_val[threadIdx.x] = 0;
  __syncthreads();
  const int k8 = blockIdx.x & 7;
  const int total = tile_start[B];
  unsigned int* my_ctr = ctr + k8 * 32;
  const int cgq = cg;
  auto tile_of = [&](int v) -> int { return ((v / cgq) * 8 + k8) * cgq + v % cgq; };

  unsigned int raw = 0;  // lane 0: the atomic's return value, valid once the matching wait has passed
  auto request = [&]() {
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "global_atomic_add %[ret], %[off], %[one], %[base] sc0\n\t"
        "s_mov_b64 exec, %[sv]"
        : [ret] "+v"(raw), [sv] "=&s"(sv)
        : [off] "v"(0), [one] "v"(1u), [base] "s"(my_ctr)
        : "memory");
  };
  auto publish = [&](int s) {  // after the wait for the atomic
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
      while (__hip_atomic_load(&ring_gen[s & 3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (s >> 2) + 1) __builtin_amdgcn_s_sleep(1);
      v = __hip_atomic_load(&ring_val[s & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return __builtin_amdgcn_readfirstlane(v);
  };

  bf16x8 wreg[NI][NT];
  auto load_w = [&](const char* w) {
    const int lane = lane_now(), h = lane >> 5;
    const int dma_r = (lane & 15) >> 2, dma_ii = lane & 3, dma_u = lane >> 4;
    const int dma_c = 2 * dma_u + (dma_ii & 1) + 8 * (dma_ii >> 1);
    const int dma_src_off = dma_r * 256 + dma_c * 16;
    const int q = lane & 15, grp16 = lane >> 4;
    const int a_lane_off = 8192 * h + (4 * (q >> 2) + (grp16 & 1) + 2 * (q & 1)) * 16 + ((q & 3) >> 1) * 8;
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      __syncthreads();
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
          const v4i16 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(wst + a_lane_off + (2 * s4) * 1024 + tt * 256));
          const v4i16 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(wst + a_lane_off + (2 * s4 + 1) * 1024 + tt * 256));
          wreg[r2 * 4 + s4][tt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
      }
    }
  };
  // The X DMA is issued from inline asm: the compiler then does not know that LDS is written behind its back and does not
  // put an s_waitcnt vmcnt(0) in front of every LDS access that might alias the stage (it cannot tell the two stage
  // buffers, or the ticket ring, apart) -- the waits are the explicit ones below.
  auto issue_x = [&](const DevGroup& dg, int64_t row0, int buf) {
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

  // prologue: ticket 0 synchronously (wave 0), ticket 1 requested (wave 1)
  int t_cur;
  if (wave == 0) {
    request();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t_cur = tile_of(publish(0));
  } else {
    t_cur = tile_of(consume(0));
    request();  // after ticket 0 has been drawn: a workgroup's tickets must ascend (the relation cursor only moves forward)
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
  DevGroup d = descs[g];
  int64_t row0 = (int64_t)(t_cur - tile_start[g]) * 64 + wave * 32;
  bool valid = row0 < d.rows;
  int d_cur = 0;  // DMA instructions of the current tile (issued last iteration)
  if (valid) {
    issue_x(d, row0, 0);
    d_cur = 8;
  }
  int s_prev = 0;  // store instructions of the previous tile that may still be in flight (0 when not known exactly)
  int buf = 0, i = 0;
  bool done = false;
  // outer loop: one pass per run of tiles of the same relation (W is loop-invariant inside, so the 128 registers stay put)
  while (!done) {
  if (!(DBG & 4) || i == 0) load_w(d.w);  // DBG & 4: W loaded once (wrong results): what do the refills cost?
  const int wcur = g;
  for (;; ++i) {
    // S0: ticket i + 1 (requested one iteration ago by wave (i + 1) & 1; younger operations: this tile's DMA, the previous stores)
    int v_next;
    if (((i + 1) & 1) == wave) {
      wait_vm(d_cur + s_prev);
      v_next = publish(i + 1);
    } else {
      v_next = consume(i + 1);
    }
    const int t_next = tile_of(v_next);
    const bool more = t_next < total;
    // S1: request ticket i + 2
    int a_now = 0;
    if (more && ((i + 2) & 1) == wave) {
      request();
      a_now = 1;
    }
    if (DBG & 2) {  // proxy order: this tile's X first, then the next tile's DMA (one tile of loads in flight)
      const int n = s_prev + a_now;
      if (n == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // S2: next tile's DMA
    int gn = g;
    DevGroup dn = d;
    int64_t n_row0 = 0;
    bool n_valid = false;
    int d_next = 0;
    if (more) {
      if (t_next >= tile_start[gn + 1]) {
        do ++gn; while (t_next >= tile_start[gn + 1]);
        dn = descs[gn];
      }
      n_row0 = (int64_t)(t_next - tile_start[gn]) * 64 + wave * 32;
      n_valid = n_row0 < dn.rows;
      if (n_valid) {
        issue_x(dn, n_row0, buf ^ 1);
        d_next = 8;
      }
    }
    // S3: this tile's X has landed (younger: previous stores, the request, the next DMA)
    if (!(DBG & 2)) wait_vm(s_prev + a_now + d_next);
    int s_now = 0;
    if (valid) {
      const char* stage = xs0 + buf * 8192;
      char* scratch = wst + wave * 8192;
      const int lc = lane_now();
      const int xo = lc & 31, h = lc >> 5;
      const int cb = (NI * h) ^ (xo & 15);
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {  // one 32-column block at a time: 16 accumulator registers next to the 128 of W
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (!(DBG & 1)) {
          const char* xrow = stage + xo * 256;
          u32x4 xa = *reinterpret_cast<const u32x4*>(xrow + cb * 16);
#pragma unroll
          for (int s = 0; s < NI; ++s) {
            u32x4 xn = xa;
            if (s + 1 < NI) xn = *reinterpret_cast<const u32x4*>(xrow + (cb ^ (s + 1)) * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[s][tt], __builtin_bit_cast(bf16x8, xa), acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            xa = xn;
          }
        }
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[r];
#pragma unroll
        for (int j = 0; j < 2; ++j)
          *reinterpret_cast<u32x4*>(scratch + xo * 256 + (cb ^ (2 * tt + j)) * 16) = pack8(v + 8 * j);
      }
      // 8 unconditional stores (the wait bookkeeping counts them): lanes whose row lies behind the segment end rewrite the
      // segment's last row with that row's own data
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
        if (NT_STORE) __builtin_nontemporal_store(ov, dst); else *dst = ov;
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


// pcopy2 (nw = 2, u = 8, depth 1, WG tickets) with the loads done by LDS-DMA from inline asm into two 8 KiB stages per wave
// and the stores fed from LDS: does the DMA path itself cost bandwidth against VGPR loads?
__global__ __launch_bounds__(128) void pcopyd_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long ntiles,
                                                     unsigned int* __restrict__ ctr, int cg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* seq = (int*)(smem + 32768);  // 2048 entries
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 2048; i += 128) seq[i] = -1;
  __syncthreads();
  const int q8 = blockIdx.x & 7;
  int step = 0;
  auto pull = [&]() -> long {
    int got = 0;
    if (lane == 0) {
      int v = atomicCAS(&seq[step], -1, -2);
      if (v == -1) {
        v = (int)atomicAdd(&ctr[q8 * 32], 1u);
        __hip_atomic_store(&seq[step], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        while (v < 0) v = __hip_atomic_load(&seq[step], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      got = v;
    }
    ++step;
    got = __builtin_amdgcn_readfirstlane(got);
    return (((long)got / cg) * 8 + q8) * cg + got % cg;
  };
  char* xs0 = smem + wave * 16384;
  auto issue = [&](long tile, int buf) {
    const uint32_t lds = (uint32_t)(size_t)(xs0 + buf * 8192);
    const char* base = (const char*)in + (tile * 2 + wave) * 8192;
    uint32_t off[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) off[q] = lane * 16 + q * 1024;
    uint32_t sv;
    asm volatile(
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
  long t = pull();
  int buf = 0;
  if (t < ntiles) issue(t, 0);
  while (t < ntiles) {
    const long t2 = pull();
    if (t2 < ntiles) {
      issue(t2, buf ^ 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const char* st = xs0 + buf * 8192;
    u32x4* dst = out + (t * 2 + wave) * 512 + lane;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(st + q * 1024 + lane * 16);
      __builtin_nontemporal_store(v, (GU32x4*)(dst + q * 64));
    }
    t = t2;
    buf ^= 1;
  }
}

template <int FLAGS, int DBG>
__global__ __launch_bounds__(128, 2) void v8_kernel(const DevGroup* __restrict__ descs, const int32_t* __restrict__ tile_start,
                                                    int B, unsigned int* __restrict__ ctr, int cg) {
  constexpr bool NT_STORE = (FLAGS & 2) != 0;
  constexpr int XAUX = (FLAGS & 1) ? 2 : 0;
  constexpr int NT = 4, NI = 8, NO = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  auto lane_now = [&]() -> int {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  char* wst = smem;
  char* xs0 = smem + 16384 + wave * 8192;
  int* ring_val = (int*)(smem + 32768);  // [4]
  int* ring_gen = ring_val + 4;          // [4]
  if (threadIdx.x < 8) ring_val[threadIdx.x] = 0;
  __syncthreads();
  const int k8 = blockIdx.x & 7;
  const int total = tile_start[B];
  unsigned int* my_ctr = ctr + k8 * 32;
  const int cgq = cg;
  auto tile_of = [&](int v) -> int { return ((v / cgq) * 8 + k8) * cgq + v % cgq; };

  unsigned int raw = 0;  // lane 0: the atomic's return value, valid once the matching wait has passed
  auto request = [&]() {
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "global_atomic_add %[ret], %[off], %[one], %[base] sc0\n\t"
        "s_mov_b64 exec, %[sv]"
        : [ret] "+v"(raw), [sv] "=&s"(sv)
        : [off] "v"(0), [one] "v"(1u), [base] "s"(my_ctr)
        : "memory");
  };
  auto publish = [&](int s) {  // after the wait for the atomic
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
      while (__hip_atomic_load(&ring_gen[s & 3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (s >> 2) + 1) __builtin_amdgcn_s_sleep(1);
      v = __hip_atomic_load(&ring_val[s & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return __builtin_amdgcn_readfirstlane(v);
  };

  bf16x8 wreg[NI][NT];
  auto load_w = [&](const char* w) {
    const int lane = lane_now(), h = lane >> 5;
    const int dma_r = (lane & 15) >> 2, dma_ii = lane & 3, dma_u = lane >> 4;
    const int dma_c = 2 * dma_u + (dma_ii & 1) + 8 * (dma_ii >> 1);
    const int dma_src_off = dma_r * 256 + dma_c * 16;
    const int q = lane & 15, grp16 = lane >> 4;
    const int a_lane_off = 8192 * h + (4 * (q >> 2) + (grp16 & 1) + 2 * (q & 1)) * 16 + ((q & 3) >> 1) * 8;
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      __syncthreads();
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
          const v4i16 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(wst + a_lane_off + (2 * s4) * 1024 + tt * 256));
          const v4i16 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(wst + a_lane_off + (2 * s4 + 1) * 1024 + tt * 256));
          wreg[r2 * 4 + s4][tt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
      }
    }
  };
  // X goes HBM -> registers (8 x 16 bytes per lane, whole rows per instruction) -> the wave's stage; the compiler places
  // the wait for the registers itself (it knows these loads and the stores; the asm atomic is older than the loads)
  u32x4 xr[NI];
  auto issue_x = [&](const DevGroup& dg, int64_t row0) {
    const char* base = dg.a + row0 * 256;
    const int64_t left = dg.rows - row0;
    const int last = left < 32 ? (int)left - 1 : 31;
    const int l = lane_now();
    const int l4 = l >> 4, c0 = (l & 15) ^ l4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int r = 4 * i + l4;
      const int c = c0 ^ (4 * (i & 3));
      r = r > last ? last : r;
      xr[i] = __builtin_nontemporal_load((const GU32x4*)(base + (uint32_t)(r * 256 + c * 16)));
    }
  };
  auto stage_x = [&]() {
    const int l = lane_now();
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(xs0 + i * 1024 + l * 16) = xr[i];
  };

  // prologue: ticket 0 synchronously (wave 0), ticket 1 requested (wave 1)
  int t_cur;
  if (wave == 0) {
    request();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t_cur = tile_of(publish(0));
  } else {
    t_cur = tile_of(consume(0));
    request();  // after ticket 0 has been drawn: a workgroup's tickets must ascend (the relation cursor only moves forward)
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
  DevGroup d = descs[g];
  int64_t row0 = (int64_t)(t_cur - tile_start[g]) * 64 + wave * 32;
  bool valid = row0 < d.rows;
  int d_cur = 0;  // DMA instructions of the current tile (issued last iteration)
  if (valid) {
    issue_x(d, row0);
    stage_x();
    d_cur = 0;
  }
  int s_prev = 0;  // store instructions of the previous tile that may still be in flight (0 when not known exactly)
  int buf = 0, i = 0;
  bool done = false;
  // outer loop: one pass per run of tiles of the same relation (W is loop-invariant inside, so the 128 registers stay put)
  while (!done) {
  load_w(d.w);
  const int wcur = g;
  for (;; ++i) {
    // S0: ticket i + 1 (requested one iteration ago by wave (i + 1) & 1; younger operations: this tile's DMA, the previous stores)
    int v_next;
    if (DBG & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (((i + 1) & 1) == wave) {
      if (s_prev == 8 && !(DBG & 4)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      v_next = publish(i + 1);
    } else {
      v_next = consume(i + 1);
    }
    const int t_next = tile_of(v_next);
    const bool more = t_next < total;
    // S1: request ticket i + 2
    int a_now = 0;
    if (more && ((i + 2) & 1) == wave) {
      request();
      a_now = 1;
    }
    // S2: next tile's DMA
    int gn = g;
    DevGroup dn = d;
    int64_t n_row0 = 0;
    bool n_valid = false;
    int d_next = 0;
    if (more) {
      if (t_next >= tile_start[gn + 1]) {
        do ++gn; while (t_next >= tile_start[gn + 1]);
        dn = descs[gn];
      }
      n_row0 = (int64_t)(t_next - tile_start[gn]) * 64 + wave * 32;
      n_valid = n_row0 < dn.rows;
      if (n_valid) issue_x(dn, n_row0);
    }
    // S3: this tile's X has landed (younger: previous stores, the request, the next DMA)
    int s_now = 0;
    if (valid) {
      const char* stage = xs0;
      char* scratch = wst + wave * 8192;
      const int lc = lane_now();
      const int xo = lc & 31, h = lc >> 5;
      const int cb = (NI * h) ^ (xo & 15);
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {  // one 32-column block at a time: 16 accumulator registers next to the 128 of W
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (!(DBG & 1)) {
          const char* xrow = stage + xo * 256;
          u32x4 xa = *reinterpret_cast<const u32x4*>(xrow + cb * 16);
#pragma unroll
          for (int s = 0; s < NI; ++s) {
            u32x4 xn = xa;
            if (s + 1 < NI) xn = *reinterpret_cast<const u32x4*>(xrow + (cb ^ (s + 1)) * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[s][tt], __builtin_bit_cast(bf16x8, xa), acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            xa = xn;
          }
        }
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[r];
#pragma unroll
        for (int j = 0; j < 2; ++j)
          *reinterpret_cast<u32x4*>(scratch + xo * 256 + (cb ^ (2 * tt + j)) * 16) = pack8(v + 8 * j);
      }
      // 8 unconditional stores (the wait bookkeeping counts them): lanes whose row lies behind the segment end rewrite the
      // segment's last row with that row's own data
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
        if (NT_STORE) __builtin_nontemporal_store(ov, dst); else *dst = ov;
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
    s_prev = s_now;
    if (valid) stage_x();  // waits for the registers (the stores just issued stay in flight)
    if (g != wcur) {
      ++i;
      break;
    }
  }
  }
}


static bool run_new(const Ctx& c, const std::string& spec, const std::string& name, std::map<std::string, int>& o) {
  auto opt = [&](const char* k, int dflt) { return o.count(k) ? o[k] : dflt; };

  if (name == "copyq") {
    const int mode = opt("mode", 0), lds = opt("lds", 0), wgs = opt("wgs", 2);
    const long ooff = (long)opt("ooff", 0) * 1024;  // shift of the output buffer (bytes), mode 0 only
    const int grid = c.cus * wgs;
    const long flat_tiles = c.rows * 256 / 32768;
    if (mode == 0) {
      CK(hipFuncSetAttribute((const void*)&copyq_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      bench(c, spec, [&] { hipLaunchKernelGGL((copyq_kernel<0>), dim3(grid), dim3(256), lds, 0, c.descs, c.tile128, c.B, (const u32x4*)c.x, (u32x4*)((opt("dst", 0) ? (char*)c.out : (char*)c.out2) + ooff), flat_tiles); });
    } else {
      CK(hipFuncSetAttribute((const void*)&copyq_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      bench(c, spec, [&] { hipLaunchKernelGGL((copyq_kernel<1>), dim3(grid), dim3(256), lds, 0, c.descs, c.tile128, c.B, (const u32x4*)c.x, (u32x4*)c.out, flat_tiles); });
    }
    return true;
  }


  if (name == "big" || name == "dyn") {
    const int u = opt("u", 8), lds = opt("lds", 0), wgs = opt("wgs", 2);
    const long nbytes = c.rows * 256;
    const long ntiles = nbytes / (u * 1024);
    static unsigned int* ctr = nullptr;
    if (!ctr) CK(hipMalloc(&ctr, 8 * 32 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    if (lds) {
      CK(hipFuncSetAttribute((const void*)&bigc_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      CK(hipFuncSetAttribute((const void*)&bigc_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      CK(hipFuncSetAttribute((const void*)&bigc_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      CK(hipFuncSetAttribute((const void*)&bigc_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      CK(hipFuncSetAttribute((const void*)&dync_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    const u32x4* in = (const u32x4*)c.x;
    u32x4* out = (u32x4*)c.out;
    for (int i = 0; i < 7; ++i) {
      if (name == "dyn") CK(hipMemsetAsync(ctr, 0, 8 * 32 * 4, 0));
      CK(hipEventRecord(e0));
      const unsigned gb = (unsigned)((ntiles + 3) / 4);
      if (name == "dyn") hipLaunchKernelGGL(dync_kernel, dim3(c.cus * wgs), dim3(256), lds, 0, in, out, nbytes / 8192, ctr);
      else if (u == 8) hipLaunchKernelGGL((bigc_kernel<8>), dim3(gb), dim3(256), lds, 0, in, out, ntiles);
      else if (u == 4) hipLaunchKernelGGL((bigc_kernel<4>), dim3(gb), dim3(256), lds, 0, in, out, ntiles);
      else if (u == 2) hipLaunchKernelGGL((bigc_kernel<2>), dim3(gb), dim3(256), lds, 0, in, out, ntiles);
      else hipLaunchKernelGGL((bigc_kernel<1>), dim3(gb), dim3(256), lds, 0, in, out, ntiles);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    const double bytes = 2.0 * nbytes;
    if (g_round == 0)
      printf("%-36s best %.3f ms %.2f TB/s | mean %.3f ms %.2f TB/s\n", spec.c_str(), best, bytes / best * 1e-9, sum / 5, bytes / (sum / 5) * 1e-9);
    fflush(stdout);
    return true;
  }

  if (name == "copyv") {
    const int sched = opt("sched", 0), waitn = opt("wait", 8), wgs = opt("wgs", 2), ilv = opt("ilv", 0), pf = opt("pf", 0);
    const int grid = c.cus * wgs;
    const long flat_tiles = c.rows * 256 / 32768;
    const u32x4* in = (const u32x4*)c.x;
    u32x4* out = (u32x4*)c.out;
#define CV(S, W, I, P) if (sched == S && waitn == W && ilv == I && pf == P) { bench(c, spec, [&] { hipLaunchKernelGGL((copyv_kernel<S, W, I, P>), dim3(grid), dim3(256), 0, 0, in, out, flat_tiles); }); return true; }
    CV(0, 8, 0, 0) CV(0, 0, 0, 0) CV(1, 8, 0, 0) CV(1, 0, 0, 0) CV(0, 0, 1, 0) CV(1, 0, 1, 0)
    CV(1, 0, 0, 2) CV(1, 0, 0, 3) CV(1, 0, 0, 4) CV(0, 0, 0, 8)
#undef CV
    return true;
  }

  if (name == "proxy") {
    const int nw = opt("nw", 8), rpw = opt("rpw", 16), wl = opt("w", 1), pad = opt("pad", 0), xg = opt("xg", 1);
    const long nbytes = c.rows * 256;
    const long ntiles = nbytes / ((long)nw * rpw * 256);
    const int lds = 32768 + nw * rpw * 256 + pad;
    const u32x4* in = (const u32x4*)c.x;
    u32x4* out = (u32x4*)c.out;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    for (int i = 0; i < 7; ++i) {
      CK(hipEventRecord(e0));
#define PX(N, R, W) if (nw == N && rpw == R && wl == W) { CK(hipFuncSetAttribute((const void*)&proxy_kernel<N, R, W>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); hipLaunchKernelGGL((proxy_kernel<N, R, W>), dim3((unsigned)((ntiles + 8 * xg - 1) / (8 * xg) * (8 * xg))), dim3(N * 64), lds, 0, in, out, (const char*)c.w, ntiles, xg); }
      PX(8, 16, 1) PX(8, 16, 0) PX(4, 32, 1) PX(4, 32, 0) PX(4, 16, 1) PX(8, 32, 1) PX(8, 8, 1) PX(16, 8, 1) PX(16, 16, 1) PX(2, 32, 1) PX(8, 4, 1) PX(16, 4, 1)
#undef PX
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    const double bytes = 2.0 * nbytes;
    printf("%-36s best %.3f ms %.2f TB/s | mean %.3f ms %.2f TB/s  (lds %d)\n", spec.c_str(), best, bytes / best * 1e-9, sum / 5, bytes / (sum / 5) * 1e-9, lds);
    fflush(stdout);
    return true;
  }

  if (name == "v4") {
    const int flags = opt("flags", 3), dbg = opt("dbg", 0), pad = opt("pad", 0);
    static TileRec* drec = nullptr;
    static int ntiles = 0, last_xg = -1;
    if (!drec || last_xg != opt("xg", 1)) {
      last_xg = opt("xg", 1);
      if (drec) CK(hipFree(drec));
      std::vector<TileRec> hr;
      for (int b = 0; b < c.B; ++b) {
        const int64_t n = c.ptr[b + 1] - c.ptr[b];
        for (int64_t r0 = 0; r0 < n; r0 += 128) {
          TileRec t;
          t.a = (const char*)c.x + (c.ptr[b] + r0) * 256;
          t.w = (const char*)c.w + (size_t)b * 32768;
          t.c = (char*)c.out + (c.ptr[b] + r0) * 256;
          t.rows = (int32_t)std::min<int64_t>(128, n - r0);
          t.pad = 0;
          hr.push_back(t);
        }
      }
      ntiles = (int)hr.size();
      const int xg = opt("xg", 1);
      if (xg > 1) {  // block b (XCD b % 8) takes tile (j / xg) * 8 * xg + (b % 8) * xg + j % xg, j = b / 8
        std::vector<TileRec> pr;
        TileRec none = hr[0];
        none.rows = 0;
        const long span = 8L * xg, nb = (ntiles + span - 1) / span * span;
        for (long b = 0; b < nb; ++b) {
          const long k = b & 7, j = b >> 3, t = (j / xg) * span + k * xg + j % xg;
          pr.push_back(t < ntiles ? hr[t] : none);
        }
        hr.swap(pr);
        ntiles = (int)hr.size();
      }
      CK(hipMalloc(&drec, hr.size() * sizeof(TileRec)));
      CK(hipMemcpy(drec, hr.data(), hr.size() * sizeof(TileRec), hipMemcpyHostToDevice));
    }
    const int lds = 65536 + pad;
    bool done = false;
#define V4_CASE(F, D)                                                                                          \
  if (flags == F && dbg == D) {                                                                                \
    CK(hipFuncSetAttribute((const void*)&v4_kernel<F, D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));   \
    bench(c, spec, [&] { hipLaunchKernelGGL((v4_kernel<F, D>), dim3(ntiles), dim3(256), lds, 0, drec, ntiles); }); \
    done = true;                                                                                               \
  }
    V4_CASE(3, 0) V4_CASE(2, 0) V4_CASE(1, 0) V4_CASE(0, 0) V4_CASE(3, 1)
#undef V4_CASE
    if (!done && g_round == 0) printf("%s: no such v4 variant\n", spec.c_str());
    return true;
  }

  if (name == "v5") {
    const int flags = opt("flags", 3), dbg = opt("dbg", 0);
    std::vector<int32_t> ht(c.B + 1);
    long tiles = 0;
    for (int b = 0; b < c.B; ++b) {
      ht[b] = (int32_t)tiles;
      tiles += (c.ptr[b + 1] - c.ptr[b] + 255) / 256;
    }
    ht[c.B] = (int32_t)tiles;
    static int32_t* dt = nullptr;
    if (!dt) {
      CK(hipMalloc(&dt, (c.B + 1) * 4));
      CK(hipMemcpy(dt, ht.data(), (c.B + 1) * 4, hipMemcpyHostToDevice));
    }
    const int lds = 2 * 32768 + 16 * 4096;
    const int grid = c.cus;
    bool done = false;
#define V5_CASE(F, D)                                                                                          \
  if (flags == F && dbg == D) {                                                                                \
    CK(hipFuncSetAttribute((const void*)&v5_kernel<F, D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));   \
    bench(c, spec, [&] { hipLaunchKernelGGL((v5_kernel<F, D>), dim3(grid), dim3(1024), lds, 0, c.descs, dt, c.B); }); \
    done = true;                                                                                               \
  }
    V5_CASE(3, 0) V5_CASE(0, 0) V5_CASE(3, 1)
#undef V5_CASE
    if (!done && g_round == 0) printf("%s: no such v5 variant\n", spec.c_str());
    return true;
  }

  if (name == "pcopy") {
    const int u = opt("u", 4), dyn = opt("dyn", 1), depth = opt("depth", 1), wgs = opt("wgs", 4), lds = opt("lds", 0), slp = opt("slp", 0), wait = opt("wait", 0);
    const long nbytes = c.rows * 256;
    const long ntiles = nbytes / (u * 1024);
    static unsigned int* ctr = nullptr;
    if (!ctr) CK(hipMalloc(&ctr, 8 * 32 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    const u32x4* in = (const u32x4*)c.x;
    u32x4* out = (u32x4*)c.out;
    for (int i = 0; i < 7; ++i) {
      CK(hipMemsetAsync(ctr, 0, 8 * 32 * 4, 0));
      CK(hipEventRecord(e0));
#define PC(UU, DD, PP) if (u == UU && dyn == DD && depth == PP) { if (lds) CK(hipFuncSetAttribute((const void*)&pcopy_kernel<UU, DD, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); hipLaunchKernelGGL((pcopy_kernel<UU, DD, PP>), dim3(c.cus * wgs), dim3(256), lds, 0, in, out, ntiles, ctr, slp, wait); }
      PC(2, 0, 0) PC(1, 0, 0) PC(8, 0, 0) PC(1, 0, 1) PC(4, 0, 1) PC(4, 1, 1) PC(4, 1, 0) PC(4, 0, 0) PC(8, 1, 1) PC(8, 0, 1) PC(2, 1, 1) PC(2, 1, 0) PC(1, 1, 0) PC(1, 1, 1) PC(2, 0, 1) PC(8, 1, 0)
#undef PC
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    const double bytes = 2.0 * nbytes;
    if (g_round == 0)
      printf("%-36s best %.3f ms %.2f TB/s | mean %.3f ms %.2f TB/s\n", spec.c_str(), best, bytes / best * 1e-9, sum / 5, bytes / (sum / 5) * 1e-9);
    fflush(stdout);
    return true;
  }

  if (name == "pcopy2") {
    const int u = opt("u", 4), nw = opt("nw", 16), depth = opt("depth", 1), wgs = opt("wgs", 1), dyn = opt("dyn", 1), lds = opt("lds", 8192), cg = opt("cg", 1), wait = opt("wait", 0) + 16 * opt("perm", 0), early = opt("early", 0);
    const long nbytes = c.rows * 256;
    const long ntiles = nbytes / ((long)u * 1024 * nw);
    static unsigned int* ctr = nullptr;
    if (!ctr) CK(hipMalloc(&ctr, 8 * 32 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    const u32x4* in = (const u32x4*)c.x;
    u32x4* out = (u32x4*)c.out;
    for (int i = 0; i < 7; ++i) {
      CK(hipMemsetAsync(ctr, 0, 8 * 32 * 4, 0));
      CK(hipEventRecord(e0));
#define PC(UU, NN, PP) if (u == UU && nw == NN && depth == PP) { CK(hipFuncSetAttribute((const void*)&pcopy2_kernel<UU, NN, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); hipLaunchKernelGGL((pcopy2_kernel<UU, NN, PP>), dim3(c.cus * wgs), dim3(NN * 64), lds, 0, in, out, ntiles, ctr, dyn, cg, wait, early); }
      PC(4, 16, 1) PC(4, 16, 0) PC(8, 8, 1) PC(8, 8, 0) PC(4, 8, 1) PC(4, 8, 0) PC(2, 16, 1) PC(2, 16, 0) PC(4, 4, 1) PC(4, 4, 0) PC(8, 4, 1) PC(8, 4, 0) PC(2, 8, 0) PC(2, 4, 0) PC(1, 4, 0) PC(1, 8, 0) PC(8, 2, 0) PC(8, 2, 1) PC(8, 12, 0) PC(8, 12, 1) PC(8, 6, 0) PC(8, 6, 1)
#undef PC
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    const double bytes = 2.0 * nbytes;
    if (g_round == 0)
      printf("%-36s best %.3f ms %.2f TB/s | mean %.3f ms %.2f TB/s\n", spec.c_str(), best, bytes / best * 1e-9, sum / 5, bytes / (sum / 5) * 1e-9);
    fflush(stdout);
    return true;
  }

  if (name == "bigr") {
    const int u = opt("u", 4), lds = opt("lds", 40960), rep = opt("rep", 1), mode = opt("mode", 0), wait = opt("wait", 0), slp = opt("slp", 0) + 256 * opt("op", 0);
    const long nbytes = c.rows * 256;
    const long ntiles = nbytes / (u * 1024);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    const u32x4* in = (const u32x4*)c.x;
    u32x4* out = (u32x4*)c.out;
    for (int i = 0; i < 7; ++i) {
      CK(hipEventRecord(e0));
      const unsigned gb = (unsigned)((ntiles + 4L * rep - 1) / (4L * rep));
#define PC(UU) if (u == UU) { CK(hipFuncSetAttribute((const void*)&bigr_kernel<UU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); hipLaunchKernelGGL((bigr_kernel<UU>), dim3(gb), dim3(256), lds, 0, in, out, ntiles, rep, mode, wait, slp); }
      PC(1) PC(2) PC(4) PC(8)
#undef PC
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    const double bytes = (opt("op", 0) ? 1.0 : 2.0) * nbytes;
    if (g_round == 0)
      printf("%-36s best %.3f ms %.2f TB/s | mean %.3f ms %.2f TB/s\n", spec.c_str(), best, bytes / best * 1e-9, sum / 5, bytes / (sum / 5) * 1e-9);
    fflush(stdout);
    return true;
  }

  if (name == "bigx" || name == "pcx") {
    const int u = opt("u", 4), lds = opt("lds", name == "bigx" ? 40960 : 0), xg = opt("xg", 1), op = opt("op", 0), fine = opt("fine", 1), wgs = opt("wgs", 4);
    const long nbytes = c.rows * 256;
    const long ntiles = nbytes / (u * 1024);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    const u32x4* in = (const u32x4*)c.x;
    u32x4* out = (u32x4*)c.out;
    for (int i = 0; i < 7; ++i) {
      CK(hipEventRecord(e0));
      unsigned gb = (unsigned)((ntiles + 3) / 4);
      gb = (gb + 8 * xg - 1) / (8 * xg) * (8 * xg);
#define PC(UU) if (u == UU) { if (name == "bigx") { CK(hipFuncSetAttribute((const void*)&bigx_kernel<UU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); hipLaunchKernelGGL((bigx_kernel<UU>), dim3(gb), dim3(256), lds, 0, in, out, ntiles, xg, op); } else { if (lds) CK(hipFuncSetAttribute((const void*)&pcx_kernel<UU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); hipLaunchKernelGGL((pcx_kernel<UU>), dim3(c.cus * wgs), dim3(256), lds, 0, in, out, ntiles, fine, op); } }
      PC(1) PC(2) PC(4) PC(8)
#undef PC
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    const double bytes = (op ? 1.0 : 2.0) * nbytes;
    if (g_round == 0)
      printf("%-36s best %.3f ms %.2f TB/s | mean %.3f ms %.2f TB/s\n", spec.c_str(), best, bytes / best * 1e-9, sum / 5, bytes / (sum / 5) * 1e-9);
    fflush(stdout);
    return true;
  }

  if (name == "pcg") {
    const int u = opt("u", 8), nw = opt("nw", 8), depth = opt("depth", 1), wgs = opt("wgs", 1), cg = opt("cg", 1), op = opt("op", 0), lds = opt("lds", 0);
    const long nbytes = c.rows * 256;
    const long ntiles = nbytes / ((long)u * 1024 * nw);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    const u32x4* in = (const u32x4*)c.x;
    u32x4* out = (u32x4*)c.out;
    for (int i = 0; i < 7; ++i) {
      CK(hipEventRecord(e0));
#define PC(UU, NN, PP) if (u == UU && nw == NN && depth == PP) { if (lds) CK(hipFuncSetAttribute((const void*)&pcg_kernel<UU, NN, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); hipLaunchKernelGGL((pcg_kernel<UU, NN, PP>), dim3(c.cus * wgs), dim3(NN * 64), lds, 0, in, out, ntiles, cg, op); }
      PC(8, 8, 1) PC(8, 8, 0) PC(4, 16, 1) PC(4, 16, 0) PC(4, 8, 1) PC(4, 8, 0) PC(4, 4, 1) PC(4, 4, 0) PC(8, 4, 1) PC(8, 4, 0)
#undef PC
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    const double bytes = (op ? 1.0 : 2.0) * nbytes;
    if (g_round == 0)
      printf("%-36s best %.3f ms %.2f TB/s | mean %.3f ms %.2f TB/s\n", spec.c_str(), best, bytes / best * 1e-9, sum / 5, bytes / (sum / 5) * 1e-9);
    fflush(stdout);
    return true;
  }

  if (name == "pcw") {
    const int depth = opt("depth", 1), wpc = opt("wpc", 6), rep = opt("rep", 4), cgb = opt("cgk", 512);
    const int lds = opt("lds", 160 * 1024 / wpc - 1024);
    const int cg = cgb / (8 * rep) > 0 ? cgb / (8 * rep) : 1;  // chunk of cgk KiB
    const long nbytes = c.rows * 256;
    const long ntiles = nbytes / 8192;
    static unsigned int* ctr = nullptr;
    if (!ctr) CK(hipMalloc(&ctr, 8 * 32 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    const u32x4* in = (const u32x4*)c.x;
    u32x4* out = (u32x4*)c.out;
    for (int i = 0; i < 7; ++i) {
      CK(hipMemsetAsync(ctr, 0, 8 * 32 * 4, 0));
      CK(hipEventRecord(e0));
      if (depth) { CK(hipFuncSetAttribute((const void*)&pcw_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); hipLaunchKernelGGL((pcw_kernel<1>), dim3(c.cus * wpc), dim3(64), lds, 0, in, out, ntiles, ctr, rep, cg); }
      else { CK(hipFuncSetAttribute((const void*)&pcw_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); hipLaunchKernelGGL((pcw_kernel<0>), dim3(c.cus * wpc), dim3(64), lds, 0, in, out, ntiles, ctr, rep, cg); }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    const double bytes = 2.0 * nbytes;
    if (g_round == 0)
      printf("%-36s best %.3f ms %.2f TB/s | mean %.3f ms %.2f TB/s\n", spec.c_str(), best, bytes / best * 1e-9, sum / 5, bytes / (sum / 5) * 1e-9);
    fflush(stdout);
    return true;
  }

  if (name == "v6" || name == "v7" || name == "v8") {
    const int flags = opt("flags", 3), dbg = opt("dbg", 0), wgs = opt("wgs", 3), cgk = opt("cgk", 512);
    const int cg = cgk / 16 > 0 ? cgk / 16 : 1;
    std::vector<int32_t> ht(c.B + 1);
    long tiles = 0;
    for (int b = 0; b < c.B; ++b) {
      ht[b] = (int32_t)tiles;
      tiles += (c.ptr[b + 1] - c.ptr[b] + 63) / 64;
    }
    ht[c.B] = (int32_t)tiles;
    int32_t* dt;
    CK(hipMalloc(&dt, (c.B + 1) * 4));
    CK(hipMemcpy(dt, ht.data(), (c.B + 1) * 4, hipMemcpyHostToDevice));
    static unsigned int* ctr = nullptr;
    if (!ctr) CK(hipMalloc(&ctr, 8 * 32 * 4));
    const int lds = opt("lds", 49152 + 1024);
    const int grid = c.cus * wgs;
    bool done = false;
#define V6_CASE(F, D)                                                                                          \
  if (flags == F && dbg == D) {                                                                                \
    CK(hipFuncSetAttribute((const void*)&v6_kernel<F, D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));   \
    bench(c, spec, [&] { CK(hipMemsetAsync(ctr, 0, 8 * 32 * 4, 0)); hipLaunchKernelGGL((v6_kernel<F, D>), dim3(grid), dim3(128), lds, 0, c.descs, dt, c.B, ctr, cg); }); \
    done = true;                                                                                               \
  }
    if (name == "v6") { V6_CASE(3, 0) V6_CASE(0, 0) V6_CASE(3, 1) V6_CASE(2, 0) V6_CASE(1, 0) }
#undef V6_CASE
#define V7_CASE(F, D)                                                                                          \
  if (name == "v7" && flags == F && dbg == D) {                                                                \
    CK(hipFuncSetAttribute((const void*)&v7_kernel<F, D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));   \
    bench(c, spec, [&] { CK(hipMemsetAsync(ctr, 0, 8 * 32 * 4, 0)); hipLaunchKernelGGL((v7_kernel<F, D>), dim3(grid), dim3(128), lds, 0, c.descs, dt, c.B, ctr, cg); }); \
    done = true;                                                                                               \
  }
    V7_CASE(3, 0) V7_CASE(0, 0) V7_CASE(3, 1) V7_CASE(2, 0) V7_CASE(1, 0) V7_CASE(3, 2) V7_CASE(3, 3) V7_CASE(3, 4)
#undef V7_CASE
    if (name == "v8") {
      const int lds8 = opt("lds", 32768 + 64);
      CK(hipFuncSetAttribute((const void*)&v8_kernel<3, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds8));
      CK(hipFuncSetAttribute((const void*)&v8_kernel<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds8));
      CK(hipFuncSetAttribute((const void*)&v8_kernel<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds8));
      CK(hipFuncSetAttribute((const void*)&v8_kernel<3, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds8));
      if (dbg == 4) bench(c, spec, [&] { CK(hipMemsetAsync(ctr, 0, 8 * 32 * 4, 0)); hipLaunchKernelGGL((v8_kernel<3, 4>), dim3(grid), dim3(128), lds8, 0, c.descs, dt, c.B, ctr, cg); });
      else if (dbg == 8) bench(c, spec, [&] { CK(hipMemsetAsync(ctr, 0, 8 * 32 * 4, 0)); hipLaunchKernelGGL((v8_kernel<3, 8>), dim3(grid), dim3(128), lds8, 0, c.descs, dt, c.B, ctr, cg); });
      else if (dbg == 0) bench(c, spec, [&] { CK(hipMemsetAsync(ctr, 0, 8 * 32 * 4, 0)); hipLaunchKernelGGL((v8_kernel<3, 0>), dim3(grid), dim3(128), lds8, 0, c.descs, dt, c.B, ctr, cg); });
      else bench(c, spec, [&] { CK(hipMemsetAsync(ctr, 0, 8 * 32 * 4, 0)); hipLaunchKernelGGL((v8_kernel<3, 1>), dim3(grid), dim3(128), lds8, 0, c.descs, dt, c.B, ctr, cg); });
      done = true;
    }
    CK(hipFree(dt));
    if (!done && g_round == 0) printf("%s: no such v6 variant\n", spec.c_str());
    return true;
  }

  if (name == "pcopyd") {
    const int wgs = opt("wgs", 3), cg = opt("cg", 16), lds = opt("lds", 32768 + 8192);
    const long nbytes = c.rows * 256;
    const long ntiles = nbytes / 16384;
    static unsigned int* ctr = nullptr;
    if (!ctr) CK(hipMalloc(&ctr, 8 * 32 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    const u32x4* in = (const u32x4*)c.x;
    u32x4* out = (u32x4*)c.out;
    CK(hipFuncSetAttribute((const void*)&pcopyd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int i = 0; i < 7; ++i) {
      CK(hipMemsetAsync(ctr, 0, 8 * 32 * 4, 0));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(pcopyd_kernel, dim3(c.cus * wgs), dim3(128), lds, 0, in, out, ntiles, ctr, cg);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    const double bytes = 2.0 * nbytes;
    if (g_round == 0)
      printf("%-36s best %.3f ms %.2f TB/s | mean %.3f ms %.2f TB/s\n", spec.c_str(), best, bytes / best * 1e-9, sum / 5, bytes / (sum / 5) * 1e-9);
    fflush(stdout);
    return true;
  }
  if (name == "rw") {
    // src / dst: 0 = x, 1 = out, 2 = out2, 3 = out2 + 32 MB
    const int mode = opt("mode", 2), src = opt("src", 0), dst = opt("dst", 1), wgs = opt("wgs", 8);
    char* bufs[4] = {(char*)c.x, (char*)c.out, c.out2, c.out2 + (32u << 20)};
    const long ntiles = c.rows * 256 / 8192;
    const u32x4* in = (const u32x4*)bufs[src];
    u32x4* out = (u32x4*)bufs[dst];
    const int grid = c.cus * wgs;
    // timing only (the data check of bench() is meaningless here)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    for (int i = 0; i < 7; ++i) {
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL((rw_kernel<0>), dim3(grid), dim3(256), 0, 0, in, out, ntiles);
      else if (mode == 1) hipLaunchKernelGGL((rw_kernel<1>), dim3(grid), dim3(256), 0, 0, in, out, ntiles);
      else hipLaunchKernelGGL((rw_kernel<2>), dim3(grid), dim3(256), 0, 0, in, out, ntiles);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    const double bytes = (mode == 2 ? 2.0 : 1.0) * c.rows * 256;
    printf("%-36s best %.3f ms %.2f TB/s | mean %.3f ms %.2f TB/s\n", spec.c_str(), best, bytes / best * 1e-9, sum / 5, bytes / (sum / 5) * 1e-9);
    fflush(stdout);
    return true;
  }
  if (name == "v3") {
    const int nwv = opt("nw", 8), flags = opt("flags", 3), wgs = opt("wgs", 1), dbg = opt("dbg", 0);
    const int wdb = opt("wdb", 1), sched = opt("sched", 0);
    const int bm = nwv * 32;
    // tile table for bm-row workgroup tiles
    std::vector<int32_t> ht(c.B + 1);
    long tiles = 0;
    for (int b = 0; b < c.B; ++b) {
      ht[b] = (int32_t)tiles;
      tiles += (c.ptr[b + 1] - c.ptr[b] + bm - 1) / bm;
    }
    ht[c.B] = (int32_t)tiles;
    int32_t* dt;
    CK(hipMalloc(&dt, (c.B + 1) * 4));
    CK(hipMemcpy(dt, ht.data(), (c.B + 1) * 4, hipMemcpyHostToDevice));
    const int lds = (wdb ? 2 : 1) * 32768 + nwv * 8192;
    const int grid = opt("g", 0) > 0 ? opt("g", 0) : c.cus * wgs;
    bool done = false;
#define V3_CASE(F, N, D, W, S)                                                                                       \
  if (flags == F && nwv == N && dbg == D && wdb == W && sched == S) {                                                \
    CK(hipFuncSetAttribute((const void*)&v3_kernel<F, N, D, W, S>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    bench(c, spec, [&] { hipLaunchKernelGGL((v3_kernel<F, N, D, W, S>), dim3(grid), dim3(N * 64), lds, 0, c.descs, dt, c.B); }); \
    done = true;                                                                                                     \
  }
    V3_CASE(3, 8, 0, 1, 0) V3_CASE(0, 8, 0, 1, 0) V3_CASE(3, 4, 0, 1, 0) V3_CASE(3, 8, 1, 1, 0)
    V3_CASE(3, 8, 0, 1, 1) V3_CASE(3, 8, 0, 0, 0) V3_CASE(3, 4, 0, 0, 0) V3_CASE(3, 4, 0, 0, 1) V3_CASE(3, 2, 0, 0, 0)
    V3_CASE(1, 8, 0, 1, 0) V3_CASE(2, 8, 0, 1, 0) V3_CASE(3, 8, 0, 1, 2) V3_CASE(0, 8, 0, 1, 2) V3_CASE(1, 8, 0, 1, 2) V3_CASE(2, 8, 0, 1, 2) V3_CASE(3, 8, 0, 0, 2) V3_CASE(3, 8, 0, 1, 4) V3_CASE(3, 8, 0, 1, 16) V3_CASE(3, 8, 0, 1, 32) V3_CASE(3, 8, 0, 1, 64)
#undef V3_CASE
    CK(hipFree(dt));
    if (!done) printf("%s: no such v3 variant\n", spec.c_str());
    return true;
  }
  return false;
}
