// Probe (not product): what a hand-written device copy of the C2 byte mix (5.4 GB read : 5.4 GB written)
// reaches on this box, by access pattern.  Answers "is the segment_matmul kernel at the box's streaming
// ceiling?" and "does the workgroup -> address schedule matter to the HBM controllers?".
//   hipcc --offload-arch=gfx950 -O3 tools/probe/hbm_copy.hip -o tools/probe/hbm_copy
// Patterns (a tile = U KiB per wave: U wave-instructions of 1 KiB, 16 B per lane):
//   cyc   wave w of the grid copies tiles w, w + W, w + 2W, ...        (chip sweeps one window)
//   blk   workgroup b owns one contiguous range of tiles                 (segment_matmul's schedule)
//   chk   blocked-cyclic, C consecutive tiles per workgroup and round
//   big   non-persistent: one tile per wave, grid covers the buffer
// Variants: nt / plain loads and stores; register prefetch of the next tile (pipe); LDS-DMA loads.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 GU32x4;

#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

__global__ void fill_kernel(uint32_t* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t h = (uint32_t)i * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (h & 0x3fff3fffu) | 0x3c003c00u;  // plausible bf16 pairs
  }
}

template <bool NT>
__device__ __forceinline__ u32x4 ld(const u32x4* p) {
  const GU32x4* g = (const GU32x4*)p;
  return NT ? __builtin_nontemporal_load(g) : *g;
}
template <bool NT>
__device__ __forceinline__ void st(u32x4* p, u32x4 v) {
  GU32x4* g = (GU32x4*)p;
  if (NT) __builtin_nontemporal_store(v, g); else *g = v;
}

// PAT: 0 cyc, 1 blk, 2 chk.  PIPE: prefetch the next tile into registers before storing this one.
template <int PAT, int U, bool NTL, bool NTS, bool PIPE>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out,
                                                   long ntiles, int chunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long W = (long)gridDim.x * 4;  // waves in the grid
  // tile sequence of this wave: index i -> tile id
  long n_mine, base = 0;
  if (PAT == 0) {
    const long gw = (long)blockIdx.x * 4 + wave;
    n_mine = ntiles > gw ? (ntiles - 1 - gw) / W + 1 : 0;
    base = gw;
  } else if (PAT == 1) {
    // workgroup range of WG-tiles (4 wave tiles each), waves take consecutive tiles inside
    const long wgt = (ntiles + 3) / 4;
    const long b0 = (long)blockIdx.x * wgt / gridDim.x, b1 = (long)(blockIdx.x + 1) * wgt / gridDim.x;
    n_mine = b1 - b0;
    base = b0;
  } else {
    const long wgt = (ntiles + 3) / 4;
    const long nch = (wgt + chunk - 1) / chunk;
    const long mine = nch > blockIdx.x ? (nch - 1 - blockIdx.x) / gridDim.x + 1 : 0;
    n_mine = mine * chunk;
    base = 0;
  }
  auto tile_of = [&](long i) -> long {
    if (PAT == 0) return base + i * W;
    if (PAT == 1) return (base + i) * 4 + wave;
    const long j = i / chunk;
    return ((j * gridDim.x + blockIdx.x) * chunk + (i - j * chunk)) * 4 + wave;
  };
  u32x4 v[U], nx[U];
  if (PIPE) {
    if (n_mine > 0) {
      const long t = tile_of(0);
      if (t < ntiles)
#pragma unroll
        for (int q = 0; q < U; ++q) nx[q] = ld<NTL>(in + (t * U + q) * 64 + lane);
    }
    for (long i = 0; i < n_mine; ++i) {
      const long t = tile_of(i);
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = nx[q];
      if (i + 1 < n_mine) {
        const long t2 = tile_of(i + 1);
        if (t2 < ntiles)
#pragma unroll
          for (int q = 0; q < U; ++q) nx[q] = ld<NTL>(in + (t2 * U + q) * 64 + lane);
      }
      if (t < ntiles)
#pragma unroll
        for (int q = 0; q < U; ++q) st<NTS>(out + (t * U + q) * 64 + lane, v[q]);
    }
  } else {
    for (long i = 0; i < n_mine; ++i) {
      const long t = tile_of(i);
      if (t >= ntiles) continue;
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = ld<NTL>(in + (t * U + q) * 64 + lane);
#pragma unroll
      for (int q = 0; q < U; ++q) st<NTS>(out + (t * U + q) * 64 + lane, v[q]);
    }
  }
}

// one tile per wave, grid covers everything
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_big_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long t = (long)blockIdx.x * 4 + wave;
  if (t >= ntiles) return;
  u32x4 v[U];
#pragma unroll
  for (int q = 0; q < U; ++q) v[q] = ld<NTL>(in + (t * U + q) * 64 + lane);
#pragma unroll
  for (int q = 0; q < U; ++q) st<NTS>(out + (t * U + q) * 64 + lane, v[q]);
}

// LDS-DMA loads (global_load_lds_dwordx4), double-buffered per wave: 2 x U KiB of LDS per wave.
template <int PAT, int U, bool NTS>
__global__ __launch_bounds__(256) void copy_dma_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long W = (long)gridDim.x * 4;
  long n_mine, base;
  if (PAT == 0) {
    const long gw = (long)blockIdx.x * 4 + wave;
    n_mine = ntiles > gw ? (ntiles - 1 - gw) / W + 1 : 0;
    base = gw;
  } else {
    const long wgt = (ntiles + 3) / 4;
    const long b0 = (long)blockIdx.x * wgt / gridDim.x, b1 = (long)(blockIdx.x + 1) * wgt / gridDim.x;
    n_mine = b1 - b0;
    base = b0;
  }
  auto tile_of = [&](long i) -> long { return PAT == 0 ? base + i * W : (base + i) * 4 + wave; };
  char* buf = smem + wave * (2 * U * 1024);
  typedef __attribute__((address_space(3))) void LDSV;
  auto issue = [&](long t, int slot) {
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const u32x4* src = in + (t * U + q) * 64 + lane;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (LDSV*)(buf + slot * U * 1024 + q * 1024), 16, 0, 0);
    }
  };
  if (n_mine > 0 && tile_of(0) < ntiles) issue(tile_of(0), 0);
  for (long i = 0; i < n_mine; ++i) {
    const long t = tile_of(i);
    const int slot = (int)(i & 1);
    const bool more = i + 1 < n_mine && tile_of(i + 1) < ntiles;
    if (more) issue(tile_of(i + 1), slot ^ 1);
    if (t >= ntiles) continue;
    // wait for tile i only (the U loads of tile i+1 may stay in flight; stores also count on vmcnt and
    // are older than both, so they are waited for as well)
    if (more) {
      if (U == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (U == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    u32x4 v[U];
#pragma unroll
    for (int q = 0; q < U; ++q) v[q] = *reinterpret_cast<const u32x4*>(buf + slot * U * 1024 + q * 1024 + lane * 16);
#pragma unroll
    for (int q = 0; q < U; ++q) st<NTS>(out + (t * U + q) * 64 + lane, v[q]);
  }
}

template <int U>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ in, uint32_t* sink, long ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long W = (long)gridDim.x * 4;
  u32x4 acc = {0, 0, 0, 0};
  for (long t = (long)blockIdx.x * 4 + wave; t < ntiles; t += W) {
#pragma unroll
    for (int q = 0; q < U; ++q) acc ^= ld<true>(in + (t * U + q) * 64 + lane);
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}
template <int U>
__global__ __launch_bounds__(256) void write_kernel(u32x4* __restrict__ out, long ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long W = (long)gridDim.x * 4;
  const u32x4 v = {(uint32_t)lane, 1, 2, 3};
  for (long t = (long)blockIdx.x * 4 + wave; t < ntiles; t += W) {
#pragma unroll
    for (int q = 0; q < U; ++q) st<true>(out + (t * U + q) * 64 + lane, v);
  }
}

static hipEvent_t e0, e1;
static double g_bytes;
static float g_best = 1e30f;
static char g_best_name[128];

template <typename F>
void timeit(const char* name, F&& launch, double bytes) {
  for (int i = 0; i < 2; ++i) launch();
  float best = 1e30f, sum = 0;
  const int reps = 5;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    sum += ms;
  }
  CK(hipGetLastError());
  printf("%-44s best %.3f ms  %.2f TB/s   mean %.3f ms  %.2f TB/s\n", name, best, bytes / best * 1e-9, sum / reps,
         bytes / (sum / reps) * 1e-9);
  fflush(stdout);
  if (bytes == g_bytes && sum / reps < g_best) {
    g_best = sum / reps;
    strncpy(g_best_name, name, sizeof(g_best_name) - 1);
  }
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz, mem clock %d kHz, bus %d bits\n", p.name, cus, p.clockRate, p.memoryClockRate,
         p.memoryBusWidth);
  const long rows = argc > 1 ? atol(argv[1]) : 21111007L;
  const long nbytes = rows * 256;                     // one [rows, 128] bf16 matrix
  const long ntile8 = nbytes / 8192, ntile4 = nbytes / 4096, ntile2 = nbytes / 2048, ntile1 = nbytes / 1024;
  u32x4 *in, *out;
  CK(hipMalloc(&in, nbytes + 65536));
  CK(hipMalloc(&out, nbytes + 65536));
  hipLaunchKernelGGL(fill_kernel, dim3(cus * 8), dim3(256), 0, 0, (uint32_t*)in, (size_t)nbytes / 4);
  CK(hipMemset(out, 0, nbytes));
  CK(hipDeviceSynchronize());
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double B2 = 2.0 * nbytes;
  g_bytes = B2;
  char nm[128];

  timeit("hipMemcpyDtoD", [&] { CK(hipMemcpyAsync(out, in, nbytes, hipMemcpyDeviceToDevice, 0)); }, B2);
  for (int g : {2, 4, 8}) {
    snprintf(nm, sizeof nm, "read-only  cyc U8 nt g=%dxCU", g);
    timeit(nm, [&] { hipLaunchKernelGGL((read_kernel<8>), dim3(cus * g), dim3(256), 0, 0, in, (uint32_t*)out, ntile8); }, (double)nbytes);
    snprintf(nm, sizeof nm, "write-only cyc U8 nt g=%dxCU", g);
    timeit(nm, [&] { hipLaunchKernelGGL((write_kernel<8>), dim3(cus * g), dim3(256), 0, 0, out, ntile8); }, (double)nbytes);
  }
#define RUN(PAT, U, NTL, NTS, PIPE, G, CH, NT_)                                                             \
  snprintf(nm, sizeof nm, "copy pat%d U%d ntl%d nts%d pipe%d g=%dxCU ch%d", PAT, U, NTL, NTS, PIPE, G, CH); \
  timeit(nm, [&] { hipLaunchKernelGGL((copy_kernel<PAT, U, NTL, NTS, PIPE>), dim3(cus * G), dim3(256), 0, 0, in, out, NT_, CH); }, B2);
  // pattern x grid, U = 8, nt both, no register pipelining
  RUN(0, 8, true, true, false, 2, 1, ntile8)
  RUN(0, 8, true, true, false, 4, 1, ntile8)
  RUN(0, 8, true, true, false, 8, 1, ntile8)
  RUN(1, 8, true, true, false, 2, 1, ntile8)
  RUN(1, 8, true, true, false, 4, 1, ntile8)
  RUN(1, 8, true, true, false, 8, 1, ntile8)
  RUN(2, 8, true, true, false, 2, 8, ntile8)
  RUN(2, 8, true, true, false, 4, 8, ntile8)
  RUN(2, 8, true, true, false, 2, 64, ntile8)
  // register-pipelined (segment_matmul's issue-early / store-late shape)
  RUN(0, 8, true, true, true, 2, 1, ntile8)
  RUN(0, 8, true, true, true, 4, 1, ntile8)
  RUN(1, 8, true, true, true, 2, 1, ntile8)
  RUN(1, 8, true, true, true, 4, 1, ntile8)
  RUN(2, 8, true, true, true, 2, 8, ntile8)
  // cache policy
  RUN(0, 8, false, false, false, 4, 1, ntile8)
  RUN(0, 8, true, false, false, 4, 1, ntile8)
  RUN(0, 8, false, true, false, 4, 1, ntile8)
  RUN(1, 8, false, false, true, 2, 1, ntile8)
  // bytes per wave per round
  RUN(0, 4, true, true, false, 4, 1, ntile4)
  RUN(0, 4, true, true, false, 8, 1, ntile4)
  RUN(0, 2, true, true, false, 8, 1, ntile2)
  RUN(0, 1, true, true, false, 8, 1, ntile1)
  RUN(1, 4, true, true, true, 4, 1, ntile4)
#undef RUN
#define RUNB(U, NTL, NTS, NT_)                                                    \
  snprintf(nm, sizeof nm, "copy big U%d ntl%d nts%d grid=%ld", U, NTL, NTS, (NT_ + 3) / 4); \
  timeit(nm, [&] { hipLaunchKernelGGL((copy_big_kernel<U, NTL, NTS>), dim3((unsigned)((NT_ + 3) / 4)), dim3(256), 0, 0, in, out, NT_); }, B2);
  RUNB(8, true, true, ntile8)
  RUNB(4, true, true, ntile4)
  RUNB(1, true, true, ntile1)
  RUNB(4, false, false, ntile4)
  RUNB(1, false, false, ntile1)
#undef RUNB
#define RUND(PAT, U, NTS, G, NT_)                                                            \
  {                                                                                          \
    const int lds = 4 * 2 * U * 1024;                                                        \
    CK(hipFuncSetAttribute((const void*)&copy_dma_kernel<PAT, U, NTS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    snprintf(nm, sizeof nm, "copy dma pat%d U%d nts%d g=%dxCU", PAT, U, NTS, G);             \
    timeit(nm, [&] { hipLaunchKernelGGL((copy_dma_kernel<PAT, U, NTS>), dim3(cus * G), dim3(256), lds, 0, in, out, NT_); }, B2); \
  }
  RUND(0, 8, true, 2, ntile8)
  RUND(1, 8, true, 2, ntile8)
  RUND(0, 4, true, 4, ntile4)
  RUND(1, 4, true, 4, ntile4)
  RUND(0, 8, false, 2, ntile8)
#undef RUND
  // verify the last copy
  {
    const size_t nchk = 1 << 20;
    uint32_t *ha = (uint32_t*)malloc(nchk * 4), *hb = (uint32_t*)malloc(nchk * 4);
    const size_t off = ((size_t)nbytes / 4 - nchk) & ~(size_t)1023;
    CK(hipMemcpy(ha, (uint32_t*)in + off, nchk * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb, (uint32_t*)out + off, nchk * 4, hipMemcpyDeviceToHost));
    printf("tail check: %s\n", memcmp(ha, hb, nchk * 4) == 0 ? "ok" : "MISMATCH");
  }
  printf("BEST (mean) %s: %.3f ms = %.2f TB/s\n", g_best_name, g_best, B2 / g_best * 1e-9);
  return 0;
}
