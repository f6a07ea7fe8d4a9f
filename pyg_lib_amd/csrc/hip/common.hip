// Library-level entry points and shared host helpers (see include/pyg_hip.h).
#include "common.h"

#include <mutex>
#include <set>
#include <utility>
#include <vector>

namespace pyg_hip {

char* last_error_buffer() {
  static thread_local char buf[kErrLen] = {0};
  return buf;
}

const DeviceInfo& device_info() {
  static std::mutex mu;
  static std::vector<DeviceInfo> cache(64, DeviceInfo{0, 0, 0});
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  if (cache[dev].num_cus == 0) {
    hipDeviceProp_t prop;
    cache[dev].device = dev;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
      cache[dev].num_cus = prop.multiProcessorCount;
      cache[dev].max_lds_per_block = (int)prop.sharedMemPerBlock;
    } else {
      cache[dev].num_cus = 256;  // MI355X
      cache[dev].max_lds_per_block = 160 * 1024;
    }
  }
  return cache[dev];
}

int ensure_dynamic_lds(const void* kern, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  PYG_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({dev, kern})) return PYG_HIP_OK;
  PYG_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert({dev, kern});
  return PYG_HIP_OK;
}

int PinnedStage::acquire(size_t bytes, void** out) {
  cur = next;
  next = next + 1 == kSlots ? 0 : next + 1;
  Slot& s = slots[cur];
  if (s.pending) {
    PYG_HIP_CHECK(hipEventSynchronize(s.ev));
    s.pending = false;
  }
  if (bytes > s.cap) {
    if (s.ptr) PYG_HIP_CHECK(hipHostFree(s.ptr));
    s.ptr = nullptr;
    s.cap = 0;
    size_t want = align_up(bytes < 65536 ? 65536 : bytes, 4096);
    PYG_HIP_CHECK(hipHostMalloc(&s.ptr, want, hipHostMallocDefault));
    s.cap = want;
  }
  if (!s.ev) PYG_HIP_CHECK(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
  *out = s.ptr;
  return PYG_HIP_OK;
}

int PinnedStage::commit(hipStream_t stream) {
  Slot& s = slots[cur];
  PYG_HIP_CHECK(hipEventRecord(s.ev, stream));
  s.pending = true;
  return PYG_HIP_OK;
}

PinnedStage& pinned_stage() {
  // one staging buffer + event per (thread, device): an event belongs to the device it was created on
  static thread_local std::vector<PinnedStage> st(64);
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  return st[dev];
}

}  // namespace pyg_hip

extern "C" {

int64_t pyg_hip_version(void) { return (int64_t)HIP_VERSION; }

const char* pyg_hip_last_error(void) { return pyg_hip::last_error_buffer(); }

const char* pyg_hip_arch(void) { return "gfx950"; }

}  // extern "C"
