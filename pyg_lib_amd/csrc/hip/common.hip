// Library-level entry points and shared host helpers (see include/pyg_hip.h).
#include "common.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

namespace pyg_hip {

char* last_error_buffer() {
  static thread_local char buf[kErrLen] = {0};
  return buf;
}

namespace {
std::atomic<int> g_float_atomic_mode{-1};  // -1: not read yet
}
int float_atomic_mode() {
  int m = g_float_atomic_mode.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("PYG_HIP_FLOAT_ATOMICS");
    m = (e && (!strcmp(e, "cas") || !strcmp(e, "CAS") || !strcmp(e, "1"))) ? 1 : 0;
    g_float_atomic_mode.store(m, std::memory_order_relaxed);
  }
  return m;
}

namespace {
struct AccumNote {
  const char* op = nullptr;
  const void* ptr = nullptr;
  size_t bytes = 0;
  const char* cleared = nullptr;
  void* stream = nullptr;
  int cas = 0;
  unsigned long long serial = 0;   // launches noted so far: two reports are equal exactly when no launch lies between them
};
std::mutex g_note_mu;
AccumNote g_note;
unsigned long long g_note_serial = 0;
unsigned long long g_note_text_serial = 0;   // the launch the cached report text describes
char g_note_text[512];
}  // namespace
void note_accumulate(const char* op, const void* ptr, size_t bytes, const char* cleared, hipStream_t stream, int cas) {
  std::lock_guard<std::mutex> lock(g_note_mu);
  g_note.op = op, g_note.ptr = ptr, g_note.bytes = bytes, g_note.cleared = cleared, g_note.stream = (void*)stream, g_note.cas = cas;
  g_note.serial = ++g_note_serial;
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

int pyg_hip_abi_version(void) { return PYG_HIP_ABI_VERSION; }

int pyg_hip_set_float_atomic_mode(int mode) {
  const int before = pyg_hip::float_atomic_mode();
  pyg_hip::g_float_atomic_mode.store(mode ? 1 : 0, std::memory_order_relaxed);
  return before;
}

const char* pyg_hip_last_accumulate_info(void) {
  static thread_local char buf[512];
  std::lock_guard<std::mutex> lock(pyg_hip::g_note_mu);
  const pyg_hip::AccumNote n = pyg_hip::g_note;
  if (n.op == nullptr) {
    snprintf(buf, sizeof(buf), "no atomically accumulating kernel has been launched by this process");
    return buf;
  }
  // The text of a launch is made ONCE (its first report) and kept: the accumulator's pointer attributes are those of that
  // moment -- asked again later they describe whoever owns the address by then (freed, reallocated), and two reports of the
  // same launch would differ.
  if (pyg_hip::g_note_text_serial != n.serial) {
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof(attr));
    const hipError_t pa = hipPointerGetAttributes(&attr, n.ptr);
    if (pa != hipSuccess) (void)hipGetLastError();  // (a freed accumulator is not an error of the caller's next launch)
    snprintf(pyg_hip::g_note_text, sizeof(pyg_hip::g_note_text),
             "%s (accumulating launch #%llu): accumulator %p (%zu bytes; %s, device %d, managed %d, allocation flags 0x%x), cleared by %s, stream %p, %s adds",
             n.op, n.serial, n.ptr, n.bytes,
             pa != hipSuccess ? "attributes unavailable" : (attr.type == hipMemoryTypeDevice ? "device memory" : "NOT plain device memory"),
             pa == hipSuccess ? attr.device : -1, pa == hipSuccess ? (int)attr.isManaged : -1, pa == hipSuccess ? attr.allocationFlags : 0u,
             n.cleared, n.stream, n.cas ? "compare-and-swap" : "hardware floating-point atomic");
    pyg_hip::g_note_text_serial = n.serial;
  }
  memcpy(buf, pyg_hip::g_note_text, sizeof(buf));
  return buf;
}

const char* pyg_hip_last_error(void) { return pyg_hip::last_error_buffer(); }

const char* pyg_hip_arch(void) { return "gfx950"; }

}  // extern "C"
