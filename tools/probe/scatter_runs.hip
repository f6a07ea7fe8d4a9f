// Write pattern of one LSD radix pass (index_sort.hip): every 8192-word tile is read in order and leaves as NB runs of
// 8192 / NB consecutive words, run d of every tile into region d of the output.  How does the pass rate depend on NB
// (256 = the 8-bit digits of the product, 2048 = an 11-bit digit that would save one of three passes)?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/scatter_runs.hip -o tools/probe/scatter_runs && tools/probe/scatter_runs
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int kTile = 8192;

__global__ __launch_bounds__(512) void pass_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int64_t n,
                                                   int nb, int64_t tiles_per_wg) {
  const int run = kTile / nb;
  const int64_t ntiles = n / kTile;
  const int64_t region = (int64_t)ntiles * run;  // words per bucket region
  for (int64_t k = 0; k < tiles_per_wg; ++k) {
    const int64_t tile = (int64_t)blockIdx.x * tiles_per_wg + k;
    if (tile >= ntiles) return;
    const uint64_t* src = in + tile * kTile;
#pragma unroll
    for (int r = 0; r < kTile / 512; ++r) {
      const int j = r * 512 + threadIdx.x;
      const uint64_t v = src[j];
      const int d = j / run;  // "digit": consecutive words of the sorted tile share it
      out[(int64_t)d * region + tile * run + (j - d * run)] = v;
    }
  }
}

int main() {
  const int64_t n = (int64_t)100000000 / kTile * kTile;
  uint64_t *a, *b;
  hipMalloc(&a, n * 8);
  hipMalloc(&b, n * 8);
  hipMemset(a, 1, n * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int wgs = 1024;
  const int64_t tiles_per_wg = (n / kTile + wgs - 1) / wgs;
  for (int nb : {1, 64, 256, 512, 1024, 2048, 4096}) {
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(pass_kernel, dim3(wgs), dim3(512), 0, 0, a, b, n, nb, tiles_per_wg);
    hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(pass_kernel, dim3(wgs), dim3(512), 0, 0, a, b, n, nb, tiles_per_wg);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("buckets %5d (runs of %4d words = %5d B): %.3f ms, %.2f TB/s read+write\n", nb, kTile / nb, kTile / nb * 8, ms,
           2.0 * n * 8 / ms / 1e9);
  }
  return 0;
}
