// How fast are uncoalesced 8-byte accesses on a node-table-sized array?  (sampler: one col gather, one table atomicMin and
// two table reads per sampled edge.)   hipcc --offload-arch=gfx950 -O3 random_access.hip -o random_access && ./random_access
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
// mode 0: load, 1: atomicMin (no return), 2: store, 3: load then atomicMin only if it would lower the value,
// 4: atomicMin with return, 5: 32-bit atomicMin on a u32 table
template <int MODE>
__global__ void k(u64* table, u64 n, u64 ops, u64* sink, u64 salt) {
  u64 acc = 0;
  for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < ops; i += (u64)gridDim.x * blockDim.x) {
    const u64 j = mix(i + salt) % n;
    const u64 v = (i << 8) | 1;
    if (MODE == 0) acc += table[j];
    else if (MODE == 1) __hip_atomic_fetch_min(&table[j], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 2) table[j] = v;
    else if (MODE == 3) {
      const u64 cur = __builtin_nontemporal_load(&table[j]);
      if (v < cur) __hip_atomic_fetch_min(&table[j], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 4) acc += __hip_atomic_fetch_min(&table[j], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_min(reinterpret_cast<unsigned*>(table) + j, (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (acc == 0x1234567) *sink = acc;
}

int main() {
  const char* names[] = {"load u64", "atomicMin u64 (no return)", "store u64", "load, atomicMin if lower", "atomicMin u64 (returning)",
                         "atomicMin u32 (no return)"};
  const u64 sizes[] = {2449029ull, 16ull << 20, 128ull << 20};
  const u64 ops = 32ull << 20;
  u64 *table, *sink;
  hipMalloc(&table, sizes[2] * 8);
  hipMalloc(&sink, 8);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int s = 0; s < 3; ++s) {
    for (int mode = 0; mode < 6; ++mode) {
      hipMemset(table, 0xff, sizes[s] * 8);
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        if (mode == 3) hipMemset(table, 0xff, sizes[s] * 8);
        hipDeviceSynchronize();
        hipEventRecord(a);
        const dim3 g(4096), t(256);
        switch (mode) {
          case 0: hipLaunchKernelGGL(k<0>, g, t, 0, 0, table, sizes[s], ops, sink, (u64)rep); break;
          case 1: hipLaunchKernelGGL(k<1>, g, t, 0, 0, table, sizes[s], ops, sink, (u64)rep); break;
          case 2: hipLaunchKernelGGL(k<2>, g, t, 0, 0, table, sizes[s], ops, sink, (u64)rep); break;
          case 3: hipLaunchKernelGGL(k<3>, g, t, 0, 0, table, sizes[s], ops, sink, (u64)rep); break;
          case 4: hipLaunchKernelGGL(k<4>, g, t, 0, 0, table, sizes[s], ops, sink, (u64)rep); break;
          default: hipLaunchKernelGGL(k<5>, g, t, 0, 0, table, sizes[s], ops, sink, (u64)rep); break;
        }
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      printf("table %4llu MB  %-28s %7.3f ms  %6.1f G ops/s\n", sizes[s] * 8 >> 20, names[mode], best, ops / best / 1e6);
    }
  }
  return 0;
}
