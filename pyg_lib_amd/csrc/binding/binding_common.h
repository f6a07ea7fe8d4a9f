// Helpers shared by the binding translation units (libpyg.so).
#pragma once

#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPCachingAllocatorMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <ATen/record_function.h>
#include <dlfcn.h>

#include "pyg_hip.h"

namespace pyg_amd {

using at::Tensor;

inline int dtype_code(at::ScalarType t) {
  switch (t) {
    case at::kFloat: return PYG_F32;
    case at::kDouble: return PYG_F64;
    case at::kHalf: return PYG_F16;
    case at::kBFloat16: return PYG_BF16;
    case at::kChar: return PYG_I8;
    case at::kByte: return PYG_U8;
    case at::kShort: return PYG_I16;
    case at::kInt: return PYG_I32;
    case at::kLong: return PYG_I64;
    default: TORCH_CHECK(false, "pyg (HIP): unsupported dtype ", t); return -1;
  }
}

// Op-level tracing: every operator entry shows up as a RECORD_FUNCTION range in the PyTorch
// profiler and as a roctx range in rocprofv3 --marker-trace, named like the schema ("pyg::segment_matmul").
// roctx is OPTIONAL: the two entry points are looked up in librocprofiler-sdk-roctx.so at first use (dlopen), so
// libpyg.so carries no link-time dependency on the profiler SDK -- a host without it just gets no roctx ranges.
struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  RoctxApi() {
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_GLOBAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_LAZY | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_GLOBAL);
    if (h) {
      push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (!push || !pop) push = nullptr, pop = nullptr;
    }
  }
};
inline const RoctxApi& roctx_api() {
  static const RoctxApi api;
  return api;
}
struct RoctxRange {
  bool on;
  explicit RoctxRange(const char* name) : on(roctx_api().push != nullptr) {
    if (on) roctx_api().push(name);
  }
  ~RoctxRange() {
    if (on) roctx_api().pop();
  }
  RoctxRange(const RoctxRange&) = delete;
  RoctxRange& operator=(const RoctxRange&) = delete;
};
#define PYG_TRACE(name)                                                \
  at::RecordFunction pyg_record_fn_(at::RecordScope::FUNCTION);          \
  if (pyg_record_fn_.isActive()) pyg_record_fn_.before(name);            \
  ::pyg_amd::RoctxRange pyg_roctx_range_(name)

inline void check_status(int rc) {
  TORCH_CHECK(rc == PYG_HIP_OK, pyg_hip_last_error());
}

// PyTorch-ROCm types HIP devices/streams as "cuda" (masquerading), hence these spellings.
namespace alloc = c10::hip::HIPCachingAllocatorMasqueradingAsCUDA;
using DeviceGuard = c10::hip::HIPGuardMasqueradingAsCUDA;

inline hipStream_t current_hip_stream(c10::DeviceIndex index) {
  return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(index).stream();
}

inline void* current_stream(const Tensor& t) {
  return static_cast<void*>(current_hip_stream((c10::DeviceIndex)t.get_device()));
}

// `flags` of pyg_hip_segment_matmul / pyg_hip_grouped_matmul (include/pyg_hip.h) for a call made on this thread:
//  * the tile schedule is a thread-local of the binding (set through pyg_binding_set_matmul_schedule, a test /
//    measurement hook; SegmentMatmul carries the forward's value into its backward, which runs on an autograd thread);
//  * fp32 arithmetic follows torch's switch exactly as the reference does (ops/cuda/matmul_kernel.cu:158-165):
//    float32MatmulPrecision() == HIGHEST (torch's default) -> IEEE fp32 MFMAs, anything else -> the split-bf16 kernels.
int& matmul_schedule_tls();
int matmul_flags(at::ScalarType t);

}  // namespace pyg_amd
