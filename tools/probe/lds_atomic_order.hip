// Probe: does ds_add_rtn_u32 hand out its return values in ascending LANE order among lanes that hit the same address?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/lds_atomic_order.hip -o tools/probe/lds_atomic_order && tools/probe/lds_atomic_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
__global__ void probe(const unsigned char* digits, unsigned* bad, int rounds) {
  __shared__ unsigned cnt[8][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = lane; j < 256; j += 64) cnt[wave][j] = 0;
  __syncthreads();
  unsigned errors = 0;
  for (int r = 0; r < rounds; ++r) {
    const unsigned d = digits[((size_t)blockIdx.x * rounds + r) * blockDim.x + threadIdx.x];
    // reference: stable rank via match-any
    unsigned long long peers = ~0ull;
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const unsigned before = cnt[wave][d];
    const unsigned want = before + (unsigned)__popcll(peers & lt);
    __builtin_amdgcn_wave_barrier();
    const unsigned got = atomicAdd(&cnt[wave][d], 1u);
    __builtin_amdgcn_wave_barrier();
    errors += got != want;
  }
  if (errors) atomicAdd(bad, errors);
}
int main() {
  const int blocks = 2048, threads = 512, rounds = 64;
  size_t n = (size_t)blocks * threads * rounds;
  unsigned char* h = (unsigned char*)malloc(n);
  unsigned char *d; unsigned *bad;
  hipMalloc(&d, n); hipMalloc(&bad, 4);
  for (int mode = 0; mode < 4; ++mode) {
    for (size_t i = 0; i < n; ++i) {
      unsigned r = (unsigned)rand();
      h[i] = mode == 0 ? (r & 255) : mode == 1 ? (r & 3) : mode == 2 ? 7 : ((r & 15) * 16);  // spread, 4 values, constant, same-bank
    }
    hipMemcpy(d, h, n, hipMemcpyHostToDevice); hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d, bad, rounds);
    unsigned hb = 0; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("mode %d: %u of %zu ranks differ from lane order\n", mode, hb, n);
  }
  return 0;
}
