// Device-to-device streaming copy kernels: the yardstick bench.py prints next to the segment_matmul roofline
// figure ("what does a hand-written copy of the same 1 read : 1 write byte mix reach on THIS box, on THESE
// buffers?").  Measurement support only -- no operator uses them.
//   mode 0: fine-grained, non-persistent: one 1 KiB wave-instruction per wave, workgroups dispatched in address
//           order (the chip sweeps both buffers through a narrow window): 6.5 TB/s on every box / buffer
//           placement seen so far -- the ceiling quoted as `achievable`.
//   mode 1: persistent, one contiguous range per workgroup, 8 KiB per wave and round, next round's loads issued
//           before this round's stores: segment_matmul's contiguous schedule without the arithmetic (5.0 - 6.2
//           TB/s depending on where the allocator placed the two buffers).
//   mode 2: persistent, wave tiles taken cyclically: the cyclic schedule without the arithmetic (5.5 - 6.0).
#include "common.h"

namespace pyg_hip {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 GU32x4;

__global__ __launch_bounds__(256) void copy_fine_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int64_t n16) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) __builtin_nontemporal_store(__builtin_nontemporal_load((const GU32x4*)(in + i)), (GU32x4*)(out + i));
}

template <int CYCLIC>
__global__ __launch_bounds__(256) void copy_persistent_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out,
                                                              int64_t wave_tiles) {
  // a wave tile = 8 KiB = 8 wave-instructions of 1 KiB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t j0, j1, step;
  if (CYCLIC) {
    j0 = (int64_t)blockIdx.x * 4 + wave;
    j1 = wave_tiles;
    step = (int64_t)gridDim.x * 4;
  } else {
    const int64_t wg_tiles = wave_tiles / 4;
    const int64_t b0 = (int64_t)blockIdx.x * wg_tiles / gridDim.x, b1 = (int64_t)(blockIdx.x + 1) * wg_tiles / gridDim.x;
    j0 = b0 * 4 + wave;
    j1 = b1 * 4;
    step = 4;
  }
  u32x4 v[8], nx[8];
  if (j0 < j1) {
#pragma unroll
    for (int q = 0; q < 8; ++q) nx[q] = __builtin_nontemporal_load((const GU32x4*)(in + (j0 * 8 + q) * 64 + lane));
  }
  for (int64_t j = j0; j < j1; j += step) {
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = nx[q];
    if (j + step < j1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) nx[q] = __builtin_nontemporal_load((const GU32x4*)(in + ((j + step) * 8 + q) * 64 + lane));
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) __builtin_nontemporal_store(v[q], (GU32x4*)(out + (j * 8 + q) * 64 + lane));
  }
}

__global__ void copy_tail_kernel(const char* __restrict__ in, char* __restrict__ out, int64_t begin, int64_t end) {
  const int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < end) out[i] = in[i];
}

// One wave that watches both clocks for `real_ticks` ticks of the constant 100 MHz counter: out[0] = shader-clock cycles
// elapsed (s_memtime), out[1] = constant-clock ticks elapsed (s_memrealtime).  It sleeps between looks (a spinning wave
// would add to the load it is meant to observe).
__global__ void clock_probe_kernel(uint64_t* out, uint64_t real_ticks) {
  if (threadIdx.x != 0) return;
  const uint64_t r0 = wall_clock64();
  const uint64_t c0 = __builtin_readcyclecounter();
  uint64_t r1 = r0;
  while (r1 - r0 < real_ticks) {
    __builtin_amdgcn_s_sleep(127);
    r1 = wall_clock64();
  }
  out[0] = __builtin_readcyclecounter() - c0;
  out[1] = r1 - r0;
}

}  // namespace
}  // namespace pyg_hip

using namespace pyg_hip;

extern "C" int pyg_hip_clock_probe(uint64_t* out2_dev, double milliseconds, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(out2_dev != nullptr && milliseconds > 0 && milliseconds < 10000, "clock_probe: bad arguments");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, stream, out2_dev, (uint64_t)(milliseconds * 1e5));
  PYG_HIP_CHECK(hipGetLastError());
  return PYG_HIP_OK;
}

extern "C" int pyg_hip_stream_copy(const void* src, void* dst, size_t bytes, int mode, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(mode >= 0 && mode <= 2, "stream_copy: unknown mode %d", mode);
  if (bytes == 0) return PYG_HIP_OK;
  PYG_HIP_REQUIRE(src && dst, "stream_copy: NULL buffer");
  PYG_HIP_REQUIRE(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0,
                  "stream_copy: buffers must be 16-byte aligned");
  const DeviceInfo& di = device_info();
  int64_t done = 0;
  if (mode == 0) {
    const int64_t n16 = (int64_t)(bytes / 16);
    if (n16 > 0) {
      PYG_HIP_REQUIRE((n16 + 255) / 256 < (1LL << 31), "stream_copy: buffer too large");
      hipLaunchKernelGGL(copy_fine_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, stream,
                         static_cast<const u32x4*>(src), static_cast<u32x4*>(dst), n16);
    }
    done = n16 * 16;
  } else {
    const int64_t wave_tiles = (int64_t)(bytes / 8192) / 4 * 4;  // whole workgroup tiles
    if (wave_tiles > 0) {
      const int grid = di.num_cus * 2;
      if (mode == 1)
        hipLaunchKernelGGL((copy_persistent_kernel<0>), dim3(grid), dim3(256), 0, stream, static_cast<const u32x4*>(src),
                           static_cast<u32x4*>(dst), wave_tiles);
      else
        hipLaunchKernelGGL((copy_persistent_kernel<1>), dim3(grid), dim3(256), 0, stream, static_cast<const u32x4*>(src),
                           static_cast<u32x4*>(dst), wave_tiles);
    }
    done = wave_tiles * 8192;
  }
  PYG_HIP_CHECK(hipGetLastError());
  if ((size_t)done < bytes) {
    const int64_t rest = (int64_t)bytes - done;
    hipLaunchKernelGGL(copy_tail_kernel, dim3((unsigned)((rest + 255) / 256)), dim3(256), 0, stream,
                       static_cast<const char*>(src), static_cast<char*>(dst), done, (int64_t)bytes);
    PYG_HIP_CHECK(hipGetLastError());
  }
  return PYG_HIP_OK;
}
