// General-shape MFMA kernel of segment_matmul / grouped_matmul (own translation unit; dispatch in matmul.hip).
//
// The reference's CUDA path hands CUTLASS one GemmCoord per group (pyg_lib/csrc/ops/cuda/matmul_kernel.cu:33-67), i.e.
// every group has its own (rows, K, M); its tests use K = 16 / 9 / 32 with M = 48 / 42 / 64
// (test/ops/test_matmul.py:56-72).  The specialised kernels of matmul.hip cover ONE (K, M) per launch with
// K in {32 ... 512}, M % 32 == 0 and 16-byte aligned rows; everything else -- K = 100 (ogbn-products), the K-per-type
// lists of a HeteroDictLinear, odd M, element-aligned views -- runs here:
//   * one workgroup (4 waves) per 128-row tile of ONE group, dispatched in address order (run-to-completion grid, the
//     blockIdx -> tile map deals every XCD runs of `run` consecutive tiles, ~256 KiB);
//   * the tile is multiplied 128 output columns at a time, the contraction in chunks of 128 bytes per row (64 16-bit
//     / 32 fp32 values): per chunk a [128 rows][chunk] image of X and a [chunk][128 columns] image of W[g] are staged
//     in LDS (double-buffered; the global loads of chunk s+1 are in registers while chunk s multiplies);
//   * alignment classes instead of shape specialisation: a group's X / W / out accesses use the widest vector (16, 8,
//     4 or 2 bytes) that divides both its base address and its row pitch, chosen per group on the host / by the plan
//     kernel (DevGroup::pad).  A row-major block is then read with whole vectors that never straddle a row end, the K
//     and M tails and the rows behind the group's end are ZERO-FILLED in the LDS image, and the MFMA loop is the same
//     for every shape; stores are masked by (row, column byte);
//   * W is read as it lies in memory: [K][M] images feed the MFMA "A" operand (8 consecutive k of one output column)
//     through the transposing LDS read ds_read_b64_tr_b16 (16-bit) / four ds_read_b32 (fp32, whose MFMA takes one
//     value per lane); a transposed `other` ([M][K] storage: the dX pass) is staged like X and read with ds_read_b128;
//   * the accumulators leave through a wave-private LDS stage (aliased on the stage buffer that is idle during the
//     tile's last chunk) so that global stores are whole row pieces, bias as in the other kernels.
// D = W^T X^T per 32x32 block as in the specialised kernels: lane (j, h) owns row j of the wave's 32 rows.

#include "matmul_common.h"

#include <algorithm>

namespace pyg_hip {
namespace {

constexpr int kGenPX = 144;                   // pitch of a [row][128-byte slice] image: 16-byte pad => conflict-free b128
constexpr int kGenXBytes = 128 * kGenPX;      // 18432
constexpr int kGenWBytes = 64 * (256 + 64);   // [64 k][128 col] 16-bit image (fp32: [32][128] x 576 = 18432)
constexpr int kGenBuf = kGenXBytes + kGenWBytes;
constexpr int kGenLds = kGenBuf;              // 38912 bytes: ONE stage buffer (the second one is the registers), 3 workgroups per CU
constexpr int kGenPS = 272;                   // epilogue stage pitch (256 bytes of one output row + pad)

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int V> struct GenVec;
template <> struct GenVec<16> { typedef u32x4 type; };
template <> struct GenVec<8> { typedef u32x2 type; };
template <> struct GenVec<4> { typedef uint32_t type; };
template <> struct GenVec<2> { typedef uint16_t type; };

// A block of 16384 / ROWBYTES rows x ROWBYTES bytes, 64 bytes per thread (256 threads), as V-byte vectors: vector idx
// = i * 256 + tid -> (row, v).  Vectors outside (rows_valid, bytes_valid) read as zero.
template <int V, int ROWBYTES>
__device__ __forceinline__ void gen_load(uint32_t (&r)[16], const char* base, uint32_t pitch, int rows_valid,
                                         int bytes_valid, int tid, bool stream) {
  constexpr int NV = ROWBYTES / V;
  constexpr int PER = 64 / V;
  constexpr int RPI = 256 / NV;  // rows per iteration: vector idx = i * 256 + tid -> row i * RPI + tid / NV, piece tid % NV
  static_assert(256 % NV == 0, "a row's pieces belong to one iteration");
  typedef typename GenVec<V>::type VT;
  typedef __attribute__((address_space(1))) VT GVT;
  // one multiply per call: the byte offset advances by the uniform RPI * pitch per iteration, the row test becomes a compare
  // of the thread's first row against a constant (the counters showed this kernel bound by its VALU instructions -- ~40 per
  // load with the row / piece arithmetic redone per vector --, not by HBM: profiles/NOTES_r6.md section 5)
  const int r0 = tid / NV, v = tid % NV;
  const bool col_ok = v * V < bytes_valid;
  const int rows_left = rows_valid - r0;   // row i * RPI + r0 exists  <=>  i * RPI < rows_left
  uint32_t off = (uint32_t)r0 * pitch + (uint32_t)(v * V);  // < 2^31: K, M bounded by the host
  const uint32_t step = (uint32_t)RPI * pitch;
  if (rows_valid >= RPI * PER && bytes_valid >= ROWBYTES) {
    // the whole block exists (every chunk but a K / M tail, every tile but a group's last): no predicates at all -- the
    // masked form below costs five VALU + five SALU instructions and a branch per vector
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const GVT* src = (const GVT*)(base + off);
      off += step;
      if constexpr (V == 16) {
        const u32x4 q = stream ? __builtin_nontemporal_load(src) : *src;
        r[4 * i] = q[0];
        r[4 * i + 1] = q[1];
        r[4 * i + 2] = q[2];
        r[4 * i + 3] = q[3];
      } else {
        const u32x2 q = stream ? __builtin_nontemporal_load(src) : *src;
        r[2 * i] = q[0];
        r[2 * i + 1] = q[1];
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const bool ok = col_ok && i * RPI < rows_left;
    const GVT* src = (const GVT*)(base + off);
    off += step;
    if constexpr (V == 16) {
      u32x4 q = {0u, 0u, 0u, 0u};
      if (ok) q = stream ? __builtin_nontemporal_load(src) : *src;
      r[4 * i] = q[0];
      r[4 * i + 1] = q[1];
      r[4 * i + 2] = q[2];
      r[4 * i + 3] = q[3];
    } else {
      static_assert(V == 8, "register path: 16- and 8-byte vectors");
      u32x2 q = {0u, 0u};
      if (ok) q = stream ? __builtin_nontemporal_load(src) : *src;
      r[2 * i] = q[0];
      r[2 * i + 1] = q[1];
    }
  }
}

template <int V, int ROWBYTES, int PITCH>
__device__ __forceinline__ void gen_stage(const uint32_t (&r)[16], char* img, int tid) {
  constexpr int NV = ROWBYTES / V;
  constexpr int PER = 64 / V;
  constexpr int RPI = 256 / NV;
  char* dst = img + (tid / NV) * PITCH + (tid % NV) * V;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    if constexpr (V == 16) {
      const u32x4 q = {r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]};
      *reinterpret_cast<u32x4*>(dst + i * (RPI * PITCH)) = q;
    } else {
      const u32x2 q = {r[2 * i], r[2 * i + 1]};
      *reinterpret_cast<u32x2*>(dst + i * (RPI * PITCH)) = q;
    }
  }
}

// Narrow classes (4- and 2-byte vectors: rows that are not even 8-byte multiples) are copied global -> LDS in place at
// staging time, a few vectors at a time: keeping 16 / 32 loads per thread in flight across the multiply costs more
// registers than such shapes are worth (with all four widths on the register path the kernel spilled 300 VGPRs).
template <int V, int ROWBYTES, int PITCH>
__device__ __forceinline__ void gen_copy(char* img, const char* base, uint32_t pitch, int rows_valid, int bytes_valid,
                                         int tid) {
  constexpr int NV = ROWBYTES / V;
  constexpr int PER = 64 / V;
  constexpr int RPI = 256 / NV;
  typedef typename GenVec<V>::type VT;
  typedef __attribute__((address_space(1))) VT GVT;
  const int r0 = tid / NV, v = tid % NV;
  const bool col_ok = v * V < bytes_valid;
  const int rows_left = rows_valid - r0;
  uint32_t off = (uint32_t)r0 * pitch + (uint32_t)(v * V);
  const uint32_t step = (uint32_t)RPI * pitch;
  char* dst = img + r0 * PITCH + v * V;
#pragma unroll 4
  for (int i = 0; i < PER; ++i) {
    VT q = 0;
    if (col_ok && i * RPI < rows_left) q = *(const GVT*)(base + off);
    off += step;
    *reinterpret_cast<VT*>(dst + i * (RPI * PITCH)) = q;
  }
}

// One operand block of a step: where it lies in memory and how much of it exists.
struct GenBlock {
  const char* base;
  uint32_t pitch;
  int rows_valid;
  int bytes_valid;
};

// CLS = the alignment class of the operand's accesses (log2 of the vector bytes; a template parameter: with the
// class as a run-time branch around the loads the arms write the same registers, the compiler copies them at the
// merge, and a copy needs the data -- the loads of chunk s + 1 were waited for where they were issued instead of
// behind the multiply of chunk s; the weight-gradient kernel lost 2.7x to the same pattern, profiles/NOTES_r4.md).
template <int ROWBYTES, int CLS>
__device__ __forceinline__ void gen_load_cls(uint32_t (&r)[16], const GenBlock& b, int tid, bool stream) {
  if constexpr (CLS >= 4) gen_load<16, ROWBYTES>(r, b.base, b.pitch, b.rows_valid, b.bytes_valid, tid, stream);
  else if constexpr (CLS == 3) gen_load<8, ROWBYTES>(r, b.base, b.pitch, b.rows_valid, b.bytes_valid, tid, stream);
}

// narrow classes (CLS < 3): the operand's own class `lg` picks 4- or 2-byte copies at staging time
template <int ROWBYTES, int PITCH, int CLS>
__device__ __forceinline__ void gen_stage_cls(int lg, const uint32_t (&r)[16], char* img, const GenBlock& b, int tid) {
  if constexpr (CLS >= 4) gen_stage<16, ROWBYTES, PITCH>(r, img, tid);
  else if constexpr (CLS == 3) gen_stage<8, ROWBYTES, PITCH>(r, img, tid);
  else if (lg >= 2) gen_copy<4, ROWBYTES, PITCH>(img, b.base, b.pitch, b.rows_valid, b.bytes_valid, tid);
  else gen_copy<2, ROWBYTES, PITCH>(img, b.base, b.pitch, b.rows_valid, b.bytes_valid, tid);
}

// Wave-private stage (32 rows x 256 bytes of output, pitch kGenPS) -> global rows, V-byte pieces masked by
// (row < rows_valid, column byte < bytes_valid).
template <int V>
__device__ __forceinline__ void gen_store(const char* st, char* dst, uint32_t pitch, int rows_valid, int bytes_valid,
                                          int lane) {
  constexpr int NV = 256 / V;
  constexpr int PER = NV / 2;
  typedef typename GenVec<V>::type VT;
  typedef __attribute__((address_space(1))) VT GVT;
  if constexpr (NV <= 64) {
    // vector idx = i * 64 + lane -> row i * RPI + lane / NV, piece lane % NV: one multiply, uniform steps (see gen_load)
    constexpr int RPI = 64 / NV;
    const int r0 = lane / NV, v = lane % NV;
    const bool col_ok = v * V < bytes_valid;
    const int rows_left = rows_valid - r0;
    uint32_t off = (uint32_t)r0 * pitch + (uint32_t)(v * V);
    const uint32_t step = (uint32_t)RPI * pitch;
    const char* src = st + r0 * kGenPS + v * V;
    if (rows_valid >= 32 && bytes_valid >= 256) {   // the whole 32 x 256-byte stage exists: no predicates
#pragma unroll 8
      for (int i = 0; i < PER; ++i) {
        const VT val = *reinterpret_cast<const VT*>(src + i * (RPI * kGenPS));
        GVT* p = (GVT*)(dst + off);
        if constexpr (V >= 8) __builtin_nontemporal_store(val, p);
        else *p = val;
        off += step;
      }
      return;
    }
#pragma unroll 8
    for (int i = 0; i < PER; ++i) {
      const VT val = *reinterpret_cast<const VT*>(src + i * (RPI * kGenPS));
      if (col_ok && i * RPI < rows_left) {
        GVT* p = (GVT*)(dst + off);
        if constexpr (V >= 8) __builtin_nontemporal_store(val, p);
        else *p = val;
      }
      off += step;
    }
  } else {   // 2-byte pieces: 128 per row, two iterations per row
#pragma unroll 8
    for (int i = 0; i < PER; ++i) {
      const int idx = i * 64 + lane;
      const int r = idx / NV;
      const int v = idx % NV;
      const VT val = *reinterpret_cast<const VT*>(st + r * kGenPS + v * V);
      if (r < rows_valid && v * V < bytes_valid) {
        GVT* p = (GVT*)(dst + ((uint32_t)r * pitch + (uint32_t)(v * V)));
        *p = val;
      }
    }
  }
}

__device__ __forceinline__ u32x2 gen_pack4(bf16_t, const float* v) {
  u32x2 q;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint16_t a = __builtin_bit_cast(uint16_t, (__bf16)v[2 * i]);
    const uint16_t b = __builtin_bit_cast(uint16_t, (__bf16)v[2 * i + 1]);
    q[i] = (uint32_t)a | ((uint32_t)b << 16);
  }
  return q;
}
__device__ __forceinline__ u32x2 gen_pack4(f16_t, const float* v) {
  u32x2 q;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint16_t a = __builtin_bit_cast(uint16_t, (_Float16)v[2 * i]);
    const uint16_t b = __builtin_bit_cast(uint16_t, (_Float16)v[2 * i + 1]);
    q[i] = (uint32_t)a | ((uint32_t)b << 16);
  }
  return q;
}

// One chunk of the contraction for one wave: acc[b] += W^T[32 b ... 32 b + 31][chunk] * X^T[chunk][32 rows], b < NBLK.
// `xp` = the lane's row of the X image (+ 16 h), `W` = the W image ([k][128 columns], or [128 columns][k] if TRANS).
// `first`: the tile's first chunk -- the accumulators start at zero, which the first MFMA of every block takes as its C operand
// (an inline constant) instead of 16 v_mov per block in front of the loop.
template <typename T, int NBLK, bool TRANS>
__device__ __forceinline__ void gen_mma(f32x16 (&acc)[4], const char* xp, const char* W, int kvalid, int lane, bool first) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  constexpr int SZ = Elem<T>::kSize;
  constexpr int PW = 128 * SZ + 64;
  typedef short v4i16 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) v4i16* lds_v4;
  const int j = lane & 31, h = lane >> 5;
  if constexpr (SZ == 2) {
    const int nks = (kvalid + 15) >> 4;
    if constexpr (!TRANS) {
      const int q = lane & 15, g1 = (lane >> 4) & 1;
      const char* wp = W + (h * 8 + (q >> 2)) * PW + (g1 * 16 + (q & 3) * 4) * 2;
      int ks0 = 0;
      if (first) {   // (kvalid >= 1: there is a step 0)
        const u32x4 xa = *reinterpret_cast<const u32x4*>(xp);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
          const v4i16 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(wp + b * 64));
          const v4i16 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(wp + 4 * PW + b * 64));
          const u32x4 wa = __builtin_bit_cast(u32x4, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
          acc[b] = mfma_chunk(T{}, wa, xa, zero);
        }
        ks0 = 1;
      }
      for (int ks = ks0; ks < nks; ++ks) {
        const u32x4 xa = *reinterpret_cast<const u32x4*>(xp + ks * 32);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
          const v4i16 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(wp + ks * 16 * PW + b * 64));
          const v4i16 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(wp + (ks * 16 + 4) * PW + b * 64));
          const u32x4 wa = __builtin_bit_cast(u32x4, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
          acc[b] = mfma_chunk(T{}, wa, xa, acc[b]);
        }
      }
    } else {
      const char* wp = W + j * kGenPX + h * 16;
      int ks0 = 0;
      if (first) {
        const u32x4 xa = *reinterpret_cast<const u32x4*>(xp);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
          const u32x4 wa = *reinterpret_cast<const u32x4*>(wp + b * 32 * kGenPX);
          acc[b] = mfma_chunk(T{}, wa, xa, zero);
        }
        ks0 = 1;
      }
      for (int ks = ks0; ks < nks; ++ks) {
        const u32x4 xa = *reinterpret_cast<const u32x4*>(xp + ks * 32);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
          const u32x4 wa = *reinterpret_cast<const u32x4*>(wp + b * 32 * kGenPX + ks * 32);
          acc[b] = mfma_chunk(T{}, wa, xa, acc[b]);
        }
      }
    }
  } else {
    // fp32: v_mfma_f32_32x32x2_f32 takes one value per lane and step; lane half h of 16-byte piece kq holds
    // k = 8 kq + 4 h + e, e = 0 ... 3 (the contraction order inside a chunk is permuted, A and B alike)
    const int nkq = (kvalid + 7) >> 3;
    if constexpr (!TRANS) {
      const char* wp = W + (h * 4) * PW + j * 4;
      for (int kq = 0; kq < nkq; ++kq) {
        const f32x4 xf = *reinterpret_cast<const f32x4*>(xp + kq * 32);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
          float a[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = *reinterpret_cast<const float*>(wp + (kq * 8 + e) * PW + b * 128);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], xf[e], acc[b], 0, 0, 0);
        }
      }
    } else {
      const char* wp = W + j * kGenPX + h * 16;
      for (int kq = 0; kq < nkq; ++kq) {
        const f32x4 xf = *reinterpret_cast<const f32x4*>(xp + kq * 32);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
          const f32x4 af = *reinterpret_cast<const f32x4*>(wp + b * 32 * kGenPX + kq * 32);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], xf[e], acc[b], 0, 0, 0);
        }
      }
    }
  }
}

// One 128-row tile of group `d` (rows row0 ...): all column passes and contraction chunks.  NBLK = 32-column blocks
// multiplied per pass (2 when the group has at most 64 columns), TRANS = `other` is stored [M][K].  One function
// per (NBLK, TRANS) so that every variant owns its accumulators: with the variants as branches inside one step loop
// the accumulators were copied between the branches' register assignments on every step (128 v_mov per chunk).
template <typename T, int NBLK, bool TRANS, int CLSX, int CLSW>
__device__ __forceinline__ void gen_tile(const DevGroup& d, const int64_t row0, char* smem, const int tid,
                                         const int lane, const int wave) {
  constexpr int SZ = Elem<T>::kSize;
  constexpr int KC = 128 / SZ;          // contraction values per chunk
  constexpr int PW = 128 * SZ + 64;     // pitch of the [k][128 columns] image of W
  constexpr int WROWB = 128 * SZ;       // bytes of one k-row of that image
  constexpr int PB = 64 / (16 * SZ) * 2;  // 32-column blocks per 256-byte epilogue pass: 4 (16-bit) / 2 (fp32)
  constexpr bool trans = TRANS;
  const int K = d.k, M = d.m;
  const int lgx = d.pad & 7, lgw = (d.pad >> 3) & 7, lgc = (d.pad >> 6) & 7;
  const int rows_here = (int)(d.rows - row0 < 128 ? d.rows - row0 : 128);
  const int nchunks = K > 0 ? (K + KC - 1) / KC : 1;
  const int ncb = (M + 127) >> 7;
  const int nsteps = ncb * nchunks;
  const uint32_t xpitch = (uint32_t)K * SZ, opitch = (uint32_t)M * SZ;  // K, M < 2^21 (host check): 128 rows < 2^31 bytes
  const char* xbase = d.a + row0 * xpitch;
  const bool xstream = ncb == 1;  // X is read once: streaming hint

  uint32_t xr[16], wr[16];
  // (chunk c of column pass nb: the step counter is carried as the pair -- `s / nchunks`, `s % nchunks` by a run-time
  // nchunks were three integer divisions per step, each a reciprocal sequence on the VALU)
  auto x_block = [&](int c) {
    return GenBlock{xbase + c * 128, xpitch, rows_here, K * SZ - c * 128};
  };
  auto w_block = [&](int c, int nb) {
    if (!trans) return GenBlock{d.w + ((int64_t)c * KC * M + nb * 128) * SZ, opitch, K - c * KC, (M - nb * 128) * SZ};
    return GenBlock{d.w + (int64_t)nb * 128 * xpitch + c * 128, xpitch, M - nb * 128, K * SZ - c * 128};
  };
  // The per-vector offsets of every (operand, class) pair are loop-invariant; hoisted out of the step loop they cost
  // ~150 registers (seen as 300 spilled VGPRs).  The thread id goes through an opaque asm per step instead: a handful
  // of integer instructions per chunk.
  auto issue = [&](int c, int nb) {
    int t2 = tid;
    asm volatile("" : "+v"(t2));
    gen_load_cls<128, CLSX>(xr, x_block(c), t2, xstream);
    if (!trans) gen_load_cls<WROWB, CLSW>(wr, w_block(c, nb), t2, false);
    else gen_load_cls<128, CLSW>(wr, w_block(c, nb), t2, false);
  };
  auto stage = [&](int c, int nb) {
    int t2 = tid;
    asm volatile("" : "+v"(t2));
    char* X = smem;
    gen_stage_cls<128, kGenPX, CLSX>(lgx, xr, X, x_block(c), t2);
    if (!trans) gen_stage_cls<WROWB, PW, CLSW>(lgw, wr, X + kGenXBytes, w_block(c, nb), t2);
    else gen_stage_cls<128, kGenPX, CLSW>(lgw, wr, X + kGenXBytes, w_block(c, nb), t2);
  };

  issue(0, 0);
  stage(0, 0);
  __syncthreads();

  const bool active = wave * 32 < rows_here;  // waves without rows still load, stage and meet the barriers
  f32x16 acc[4];

  int c = 0, nb = 0;   // chunk and column pass of step s
  for (int s = 0; s < nsteps; ++s) {
    int c1 = c + 1, nb1 = nb;   // ... of step s + 1
    if (c1 == nchunks) c1 = 0, nb1 = nb + 1;
    if (s + 1 < nsteps) issue(c1, nb1);
    const char* X = smem;
    const char* W = X + kGenXBytes;
    const int mvalid = M - nb * 128 < 128 ? M - nb * 128 : 128;
    const int nblk = (mvalid + 31) >> 5;
    if (active) {
      if constexpr (SZ == 4) {   // (fp32: zeroed here -- through gen_mma's first step the kernel spilled; the 16-bit types: see gen_mma)
        if (c == 0) {
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        }
      }
      const int kvalid = K - c * KC < KC ? K - c * KC : KC;
      int l1 = lane;
      asm volatile("" : "+v"(l1));  // as in issue(): keep the lane-derived LDS offsets of all variants out of registers
      const char* xp = X + (wave * 32 + (l1 & 31)) * kGenPX + (l1 >> 5) * 16;
      // column blocks behind M multiply the zero-filled part of the W image
      gen_mma<T, NBLK, TRANS>(acc, xp, W, kvalid, l1, c == 0);
    }
    if (s + 1 < nsteps || c == nchunks - 1) __syncthreads();  // everybody has multiplied: the buffer is free
    if (c == nchunks - 1) {
      if (active) {
        // epilogue through the (now idle) stage buffer; the next chunk is still in xr / wr
        char* st = smem + wave * (32 * kGenPS);
        int l2 = lane;
        asm volatile("" : "+v"(l2));
        const int j = l2 & 31, h = l2 >> 5;
        const int wrows = rows_here - wave * 32 < 32 ? rows_here - wave * 32 : 32;
        char* obase = d.c + (row0 + wave * 32) * (int64_t)opitch + (int64_t)nb * 128 * SZ;
        const T* bias = d.bias ? reinterpret_cast<const T*>(d.bias) + nb * 128 : nullptr;
#pragma unroll
        for (int p = 0; p < 4 / PB; ++p) {
          if (p * PB < nblk) {
#pragma unroll
            for (int bb = 0; bb < PB; ++bb) {
              const int b = p * PB + bb;
              if (b < nblk) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                  float v[4];
                  const int col = 32 * b + 8 * g4 + 4 * h;
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    v[e] = acc[b][4 * g4 + e];
                    if (bias && col + e < mvalid) v[e] = round_to(T{}, v[e]) + load_bias(bias + col + e);
                  }
                  char* dst = st + j * kGenPS + (32 * bb + 8 * g4 + 4 * h) * SZ;
                  if constexpr (SZ == 2) {
                    *reinterpret_cast<u32x2*>(dst) = gen_pack4(T{}, v);
                  } else {
                    const f32x4 o = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(dst) = o;
                  }
                }
              }
            }
            const int bvalid = mvalid * SZ - p * 256;
            char* od = obase + p * 256;
            if (lgc >= 4) gen_store<16>(st, od, opitch, wrows, bvalid, l2);
            else if (lgc == 3) gen_store<8>(st, od, opitch, wrows, bvalid, l2);
            else if (lgc == 2) gen_store<4>(st, od, opitch, wrows, bvalid, l2);
            else gen_store<2>(st, od, opitch, wrows, bvalid, l2);
          }
        }
      }
      if (s + 1 < nsteps) __syncthreads();  // everybody is done with its stage before the next chunk lands there
    }
    if (s + 1 < nsteps) {
      stage(c1, nb1);
      __syncthreads();
    }
    c = c1;
    nb = nb1;
  }
}

template <typename T>
__global__ __launch_bounds__(256, 3) void mfma_rows_gen_kernel(const DevGroup* __restrict__ descs,
                                                               const int32_t* __restrict__ tile_start, int B,
                                                               int run_log2) {
  static_assert((128 / Elem<T>::kSize) * (128 * Elem<T>::kSize + 64) <= kGenWBytes, "W image");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // blockIdx -> tile: consecutive workgroup ids go to consecutive XCDs; XCD k is dealt runs of 2^run_log2
  // consecutive tiles (tiles ((s >> r) * 8 + k) << r ... ), so every XCD streams ~256 KiB contiguous pieces
  const int bs = (int)blockIdx.x >> 3, bk = (int)blockIdx.x & 7;
  const int t = ((((bs >> run_log2) << 3) + bk) << run_log2) + (bs & ((1 << run_log2) - 1));
  const int total = tile_start[B];
  if (t >= total) return;

  // group of the tile: the last g with tile_start[g] <= t, by a 64-ary search (two dependent loads for B <= 4096)
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int step = (hi - lo + 63) >> 6;
    const int idx = lo + lane * step;
    const bool pred = idx < hi && tile_start[idx < hi ? idx : lo] <= t;
    const int n = __builtin_popcountll(__ballot(pred));  // lane 0 always holds (tile_start[lo] <= t)
    lo += (n - 1) * step;
    hi = lo + step < hi ? lo + step : hi;
  }
  const DevGroup d = descs[lo];
  const int64_t row0 = (int64_t)(t - tile_start[lo]) * 128;
  // one straight-line copy of the tile per (column blocks, W storage, alignment classes of the X and of the W accesses:
  // 16-byte, 8-byte or narrower)
  const int cx = d.pad & 7, cw = (d.pad >> 3) & 7;
#define PYG_GEN_TILE_W(NB, TR, CX)                                                       \
  do {                                                                                   \
    if (cw >= 4) gen_tile<T, NB, TR, CX, 4>(d, row0, smem, tid, lane, wave);             \
    else if (cw == 3) gen_tile<T, NB, TR, CX, 3>(d, row0, smem, tid, lane, wave);        \
    else gen_tile<T, NB, TR, CX, 2>(d, row0, smem, tid, lane, wave);                     \
  } while (0)
#define PYG_GEN_TILE(NB, TR)                                                             \
  do {                                                                                   \
    if (cx >= 4) PYG_GEN_TILE_W(NB, TR, 4);                                              \
    else if (cx == 3) PYG_GEN_TILE_W(NB, TR, 3);                                         \
    else PYG_GEN_TILE_W(NB, TR, 2);                                                      \
  } while (0)
  if (d.m > 64) {
    if (!d.trans) PYG_GEN_TILE(4, false);
    else PYG_GEN_TILE(4, true);
  } else {
    if (!d.trans) PYG_GEN_TILE(2, false);
    else PYG_GEN_TILE(2, true);
  }
#undef PYG_GEN_TILE_W
#undef PYG_GEN_TILE
}

template <typename T>
int launch_gen(const DevGroup* descs, const int32_t* tile_start, int B, int64_t tiles_upper, int64_t mean_k,
               hipStream_t stream) {
  const void* kern = reinterpret_cast<const void*>(&mfma_rows_gen_kernel<T>);
  if (int rc_ = ensure_dynamic_lds(kern, kGenLds)) return rc_;
  // every XCD is dealt runs of ~256 KiB of consecutive X tiles (32 / 64 KiB pieces per XCD are the slow ones)
  const int64_t tile_bytes = std::max<int64_t>(128 * mean_k * Elem<T>::kSize, 1);
  int run_log2 = 0;
  while (run_log2 < 4 && (tile_bytes << (run_log2 + 1)) <= 262144) ++run_log2;
  const int64_t per = 8LL << run_log2;
  const int64_t grid = (std::max<int64_t>(tiles_upper, 1) + per - 1) / per * per;
  hipLaunchKernelGGL((mfma_rows_gen_kernel<T>), dim3((unsigned)grid), dim3(256), kGenLds, stream, descs, tile_start, B,
                     run_log2);
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

}  // namespace

int launch_matmul_gen(int dtype, const void* descs, const int32_t* tile_start, int B, int64_t tiles_upper, int64_t mean_k,
                      hipStream_t stream) {
  const DevGroup* d = static_cast<const DevGroup*>(descs);
  if (dtype == PYG_BF16) return launch_gen<bf16_t>(d, tile_start, B, tiles_upper, mean_k, stream);
  if (dtype == PYG_F16) return launch_gen<f16_t>(d, tile_start, B, tiles_upper, mean_k, stream);
  if (dtype == PYG_F32) return launch_gen<float>(d, tile_start, B, tiles_upper, mean_k, stream);
  return fail(PYG_HIP_ERR_INVALID, "matmul (general shapes): unsupported dtype %d", dtype);
}

}  // namespace pyg_hip
