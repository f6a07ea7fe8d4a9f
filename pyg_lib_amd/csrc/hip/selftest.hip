// In-process health check of "clear an accumulator, add to it from all over the chip, read it back" -- the pattern behind
// every kernel of this library that still accumulates through atomics (reduce.hip's small sums, the fused R-GCN kernel).
//
// Why it exists (profiles/NOTES_r4.md section 1): on one lease per round a full test pass lost, doubled or
// garbled updates in exactly the kernels that use the hardware's floating-point atomic adds, while a stand-alone probe
// on hipMalloc memory never reproduced it.  This check runs INSIDE the process that saw the failure, on the caller's
// memory (the torch caching allocator's blocks in the test suite) and stream, and crosses the three things that could be
// at fault so that the next occurrence names one:
//   flavour   hw f32 add | hw packed-bf16 add | hw f64 add | CAS f32 (no float-atomic unit) | integer add (control)
//   clearing  hipMemsetAsync | a fill kernel with plain stores | a fill kernel with write-through (sc0 sc1) stores
//   readback  a copy kernel with plain loads into a second buffer | hipMemcpyAsync of the accumulator itself
// Before every variant the accumulator is scribbled with NaN patterns, so a clear that did not arrive shows as NaN, a clear
// that arrived late (after the first adds) or a stale read as a small count, a doubled update as a large one.
#include "common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <string>

namespace pyg_hip {
namespace {

typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
typedef short s16x2_v __attribute__((ext_vector_type(2)));

constexpr int kAdds = 16;  // updates per accumulator cell, each from a different workgroup

__global__ void st_scribble_kernel(uint32_t* p, int64_t words) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = 0x7fc17fc1u;  // NaN as f32, (NaN, NaN) as bf16 pair, NaN-ish high word of an f64
}
__global__ void st_fill_plain_kernel(uint32_t* p, int64_t words) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}
__global__ void st_fill_wt_kernel(uint32_t* p, int64_t words) {  // system-scope stores: leave the L2 at once
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x)
    __hip_atomic_store(p + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Cell c receives +1 from kAdds workgroups: workgroup b adds to the cells of "stripe" (b + j * stride) -- consecutive
// workgroups (which land on different XCDs) hit the same cells.
template <int FLAVOUR>
__global__ void st_add_kernel(void* acc, int64_t cells) {
  const int64_t per = blockDim.x;  // cells per workgroup visit
  const int64_t groups = (cells + per - 1) / per;
  // workgroup b serves cell group (b / kAdds); the kAdds workgroups of one group are neighbours in the grid
  const int64_t g = blockIdx.x / kAdds;
  if (g >= groups) return;
  const int64_t c = g * per + threadIdx.x;
  if (c >= cells) return;
  if constexpr (FLAVOUR == 0) {
    unsafeAtomicAdd(static_cast<float*>(acc) + c, 1.0f);
  } else if constexpr (FLAVOUR == 1) {
    const bf16x2_v one = {(__bf16)1.0f, (__bf16)1.0f};
    (void)__builtin_amdgcn_global_atomic_fadd_v2bf16((__attribute__((address_space(1))) bf16x2_v*)(static_cast<uint32_t*>(acc) + c),
                                                     __builtin_bit_cast(s16x2_v, one));
  } else if constexpr (FLAVOUR == 2) {
    unsafeAtomicAdd(static_cast<double*>(acc) + c, 1.0);
  } else if constexpr (FLAVOUR == 3) {
    unsigned int* p = static_cast<unsigned int*>(acc) + c;
    unsigned int old = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (!__hip_atomic_compare_exchange_strong(p, &old, __builtin_bit_cast(unsigned int, __builtin_bit_cast(float, old) + 1.0f),
                                                 __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
    }
  } else {
    atomicAdd(static_cast<int*>(acc) + c, 1);
  }
}

__global__ void st_copy_kernel(const uint32_t* src, uint32_t* dst, int64_t words) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

const char* kFlavour[] = {"hw f32 add", "hw packed-bf16 add", "hw f64 add", "CAS f32 add", "int32 add"};
const char* kClear[] = {"hipMemsetAsync", "fill kernel (plain stores)", "fill kernel (write-through stores)"};
const char* kRead[] = {"copy kernel (plain loads) + D2H", "D2H of the accumulator"};

struct Verdict {
  int64_t wrong = 0, nan = 0, low = 0, high = 0;
  int64_t first = -1;
  double got = 0, want = 0;
};

Verdict judge(int flavour, const void* host, int64_t cells) {
  Verdict v;
  for (int64_t c = 0; c < cells; ++c) {
    double got[2];
    int n = 1;
    if (flavour == 0 || flavour == 3) got[0] = static_cast<const float*>(host)[c];
    else if (flavour == 2) got[0] = static_cast<const double*>(host)[c];
    else if (flavour == 4) got[0] = static_cast<const int32_t*>(host)[c];
    else {
      const uint32_t w = static_cast<const uint32_t*>(host)[c];
      uint32_t lo = w << 16, hi = w & 0xffff0000u;
      float a, b;
      ::memcpy(&a, &lo, 4);
      ::memcpy(&b, &hi, 4);
      got[0] = a, got[1] = b, n = 2;
    }
    for (int i = 0; i < n; ++i) {
      if (got[i] == (double)kAdds) continue;
      ++v.wrong;
      if (got[i] != got[i]) ++v.nan;
      else if (got[i] < kAdds) ++v.low;
      else ++v.high;
      if (v.first < 0) v.first = c, v.got = got[i], v.want = kAdds;
    }
  }
  return v;
}

}  // namespace
}  // namespace pyg_hip

using namespace pyg_hip;

extern "C" {

int pyg_hip_atomic_selftest(void* scratch, size_t scratch_bytes, int rounds, char* report, size_t report_cap, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PYG_HIP_REQUIRE(scratch != nullptr && scratch_bytes >= (1u << 16) && ((uintptr_t)scratch & 15) == 0,
                  "atomic_selftest: needs >= 64 KiB of 16-byte aligned device scratch");
  PYG_HIP_REQUIRE(report != nullptr && report_cap > 0, "atomic_selftest: no report buffer");
  if (rounds < 1) rounds = 1;
  // accumulator = first half, copy target = second half; 8-byte cells in the f64 flavour
  size_t half = scratch_bytes / 2 / 256 * 256;
  if (half > (8u << 20)) half = 8u << 20;
  char* acc = static_cast<char*>(scratch);
  char* mirror = acc + half;
  void* host = nullptr;
  PYG_HIP_CHECK(hipHostMalloc(&host, half, hipHostMallocDefault));
  std::string text;
  char line[512];
  int bad_variants = 0;
  const int64_t words = (int64_t)(half / 4);
  const unsigned fill_grid = (unsigned)std::min<int64_t>((words + 255) / 256, 4096);
  for (int flavour = 0; flavour < 5; ++flavour) {
    const int64_t cells = flavour == 2 ? words / 2 : words;
    const unsigned add_grid = (unsigned)(((cells + 255) / 256) * kAdds);
    for (int clear = 0; clear < 3; ++clear) {
      for (int read = 0; read < 2; ++read) {
        Verdict total;
        int bad_rounds = 0;
        for (int r = 0; r < rounds; ++r) {
          hipLaunchKernelGGL(st_scribble_kernel, dim3(fill_grid), dim3(256), 0, stream, reinterpret_cast<uint32_t*>(acc), words);
          if (clear == 0) PYG_HIP_CHECK(hipMemsetAsync(acc, 0, half, stream));
          else if (clear == 1)
            hipLaunchKernelGGL(st_fill_plain_kernel, dim3(fill_grid), dim3(256), 0, stream, reinterpret_cast<uint32_t*>(acc), words);
          else
            hipLaunchKernelGGL(st_fill_wt_kernel, dim3(fill_grid), dim3(256), 0, stream, reinterpret_cast<uint32_t*>(acc), words);
          switch (flavour) {
            case 0: hipLaunchKernelGGL(st_add_kernel<0>, dim3(add_grid), dim3(256), 0, stream, acc, cells); break;
            case 1: hipLaunchKernelGGL(st_add_kernel<1>, dim3(add_grid), dim3(256), 0, stream, acc, cells); break;
            case 2: hipLaunchKernelGGL(st_add_kernel<2>, dim3(add_grid), dim3(256), 0, stream, acc, cells); break;
            case 3: hipLaunchKernelGGL(st_add_kernel<3>, dim3(add_grid), dim3(256), 0, stream, acc, cells); break;
            default: hipLaunchKernelGGL(st_add_kernel<4>, dim3(add_grid), dim3(256), 0, stream, acc, cells); break;
          }
          const char* src = acc;
          if (read == 0) {
            hipLaunchKernelGGL(st_copy_kernel, dim3(fill_grid), dim3(256), 0, stream, reinterpret_cast<const uint32_t*>(acc),
                               reinterpret_cast<uint32_t*>(mirror), words);
            src = mirror;
          }
          PYG_HIP_CHECK(hipGetLastError());
          PYG_HIP_CHECK(hipMemcpyAsync(host, src, half, hipMemcpyDeviceToHost, stream));
          PYG_HIP_CHECK(hipStreamSynchronize(stream));
          const Verdict v = judge(flavour, host, cells);
          if (v.wrong) {
            ++bad_rounds;
            total.wrong += v.wrong, total.nan += v.nan, total.low += v.low, total.high += v.high;
            if (total.first < 0) total.first = v.first, total.got = v.got, total.want = v.want;
          }
        }
        if (bad_rounds) {
          ++bad_variants;
          snprintf(line, sizeof(line),
                   "  BAD  %-18s | %-34s | %-31s : %d of %d rounds, %lld wrong values (%lld NaN = clear missed, %lld low = "
                   "updates lost / clear late / stale read, %lld high), first cell %lld: %g != %g\n",
                   kFlavour[flavour], kClear[clear], kRead[read], bad_rounds, rounds, (long long)total.wrong, (long long)total.nan,
                   (long long)total.low, (long long)total.high, (long long)total.first, total.got, total.want);
          text += line;
        }
      }
    }
  }
  (void)hipHostFree(host);
  hipPointerAttribute_t attr;
  ::memset(&attr, 0, sizeof(attr));
  const hipError_t pa = hipPointerGetAttributes(&attr, scratch);
  snprintf(line, sizeof(line),
           "atomic self-test: %d of 30 variants bad (5 flavours x 3 clears x 2 readbacks, %d rounds, %zu KiB accumulator at %p: "
           "%s, device %d, managed %d, flags 0x%x; stream %p)\n",
           bad_variants, rounds, half >> 10, scratch,
           pa != hipSuccess ? "attributes unavailable" : (attr.type == hipMemoryTypeDevice ? "device memory" : "NOT plain device memory"),
           pa == hipSuccess ? attr.device : -1, pa == hipSuccess ? (int)attr.isManaged : -1,
           pa == hipSuccess ? attr.allocationFlags : 0u, (void*)stream);
  text = std::string(line) + text;
  snprintf(report, report_cap, "%s", text.c_str());
  return bad_variants;
}

}  // extern "C"
