// Shared host/device helpers for the gfx950 kernels behind include/pyg_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "pyg_hip.h"

namespace pyg_hip {

// ---- error reporting (thread local; see pyg_hip_last_error) ---------------------------------
char* last_error_buffer();
constexpr int kErrLen = 512;

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buffer(), kErrLen, fmt, ap);
  va_end(ap);
  return code;
}

#define PYG_HIP_CHECK(expr)                                                             \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess)                                                               \
      return ::pyg_hip::fail(PYG_HIP_ERR_RUNTIME, "%s failed: %s (%s:%d)", #expr,       \
                             hipGetErrorString(_e), __FILE__, __LINE__);                \
  } while (0)

#define PYG_HIP_REQUIRE(cond, ...)                                           \
  do {                                                                       \
    if (!(cond)) return ::pyg_hip::fail(PYG_HIP_ERR_INVALID, __VA_ARGS__);   \
  } while (0)

inline size_t dtype_size(int dtype) {
  switch (dtype) {
    case PYG_F32: return 4;
    case PYG_F64: return 8;
    case PYG_F16: return 2;
    case PYG_BF16: return 2;
    case PYG_I8: return 1;
    case PYG_U8: return 1;
    case PYG_I16: return 2;
    case PYG_I32: return 4;
    case PYG_I64: return 8;
    default: return 0;
  }
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Device properties, cached per device index (the reference keeps un-keyed process globals,
// pyg_lib/csrc/ops/cuda/matmul_kernel.cu:19,118-119 -- not replicated).
struct DeviceInfo {
  int device;
  int num_cus;
  int max_lds_per_block;
};
const DeviceInfo& device_info();

// hipFuncAttributeMaxDynamicSharedMemorySize for `kern` on the CURRENT device, applied once per (device, kernel)
// for the whole process (the attribute is per device: a per-thread "done" flag would skip the second GPU).
int ensure_dynamic_lds(const void* kern, int bytes);

// Pinned staging (thread local) for small host -> device descriptor copies: a ring of kSlots buffers, each with its own event.
// acquire() hands out the next slot (waiting for ITS last copy only -- with one slot every call waited for the
// previous call's copy, i.e. for the kernel in front of it: back-to-back short operators ran in lock-step with the
// device); commit() records the slot's event behind the copy the caller has just queued.
struct PinnedStage {
  static constexpr int kSlots = 4;
  struct Slot {
    void* ptr = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
  };
  Slot slots[kSlots];
  int next = 0;   // slot of the next acquire()
  int cur = 0;    // slot handed out last (commit() records its event)
  int acquire(size_t bytes, void** out);
  int commit(hipStream_t stream);
};
PinnedStage& pinned_stage();

// Float-atomic flavour of the kernels that still accumulate through atomics (reduce.hip's small / element-wise sums, the
// fused R-GCN kernel): 0 = hardware adds (global_atomic_add_f32 / _f64 / _pk_add_bf16 / _pk_add_f16), 1 = compare-and-swap
// loops on the containing 32- / 64-bit word.  Process-wide DEFAULT (environment PYG_HIP_FLOAT_ATOMICS=hw|cas, read once;
// pyg_hip_set_float_atomic_mode overrides it): a call's own flag bit (PYG_HIP_SCATTER_CAS, PYG_HIP_RGCN_CAS) also selects
// the CAS form.
int float_atomic_mode();
// remembered for pyg_hip_last_accumulate_info (failure reports): what the last atomically accumulating launch was
void note_accumulate(const char* op, const void* ptr, size_t bytes, const char* cleared, hipStream_t stream, int cas);

// int64 key sort shared between index_sort and the sort-based scatter (index_sort.hip).  `max_value`
// bounds the (non-negative) keys, so no device->host read is needed.
size_t index_sort_ws_bytes_i64(int64_t n);
int index_sort_i64(const int64_t* keys, int64_t n, int64_t max_value, int64_t* keys_out, int64_t* idx_out,
                   void* ws, size_t ws_bytes, hipStream_t stream);

// mt_jump.hip: jump-ahead tables of MT19937 for the sampler's parallel word generation.  The stream is
// cut into segments of kMtSeg raw values; list k holds the set coefficient positions of
// x^(k * kMtSeg) mod phi on the current device (immutable, created on first use); `max_span` = the
// largest window of raw values one of `parts` equal shares of the list touches.
constexpr int kMtSeg = 20480;
constexpr int kMtMaxSeg = 64;
constexpr int kMtStride = 4;  // long rounds: the jump segments start kMtStride * kMtSeg apart (and are that long)
int mt_jump_list(int k, int parts, const uint16_t** idx_dev, int* count, int* max_span);

// csr.hip: row sums seeded from `out` (the atomic-free, source-order back end of segment_sum_coo).
// (`fresh` != 0: `out` is uninitialised and every slot is written; else the sums are added to its values.  `perm`,
// optional: row r sums the source rows perm[indptr[r]] ... perm[indptr[r + 1] - 1] -- a scatter through its stable index sort,
// i.e. in source order: deterministic, the order of the reference's sequential CPU loop)
int segment_csr_sum(int dtype, const void* src, const int64_t* indptr, int64_t indptr_stride, const int64_t* perm, void* out,
                    int64_t leading, int64_t rows, int64_t E, int64_t K, int fresh, hipStream_t stream, void* hub_ws = nullptr,
                    size_t hub_ws_bytes = 0);
// (`hub_ws`: scratch for rows of more than 512 positions -- with it their chunks are dealt to workgroups, without it each such
// row is one workgroup's; any size: the chunk length adapts, too little of it means "without")
// csr.hip: min / max (+ first-match arg) over CSR rows, optionally reading source position perm[e]
// instead of e -- the atomic-free back end of sorted and sort-based scatter_min/max (reduce.hip).
int segment_csr_minmax(int is_min, int dtype, const void* src, const int64_t* indptr, int64_t indptr_stride,
                       const int64_t* perm, void* out, int64_t* arg, int fresh, int64_t leading, int64_t rows,
                       int64_t E, int64_t K, hipStream_t stream, void* hub_ws = nullptr, size_t hub_ws_bytes = 0);

}  // namespace pyg_hip
