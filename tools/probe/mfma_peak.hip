// Probe (not product): sustained MFMA rate of gfx950 with nothing but MFMAs in the loop, to put a
// measured ceiling next to the data-sheet peak used in DESIGN.md / bench.py's roofline.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_peak.hip -o tools/probe/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, bool BF16, bool RANDOM>
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  if (RANDOM) {  // operands with random mantissas: the multiplier arrays toggle like on real data
    uint32_t hsh = (threadIdx.x + 1u) * 2654435761u ^ (blockIdx.x + 7u) * 40503u;
    hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
    a = __uint_as_float(0x3f000000u | (hsh & 0x7fffffu)) - 0.75f;
    hsh *= 3266489917u; hsh ^= hsh >> 16;
    b = __uint_as_float(0x3f000000u | (hsh & 0x7fffffu)) - 0.75f;
  }
  bf16x8 av, bv;
  for (int e = 0; e < 8; ++e) { av[e] = (__bf16)a; bv[e] = (__bf16)b; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if constexpr (BF16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) out[0] = s;
}

template <int NACC, bool BF16, bool RANDOM = false>
void run(const char* name, int wgs_per_cu, int cus) {
  float* out;
  hipMalloc(&out, 4);
  const int iters = BF16 ? 40000 : 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((spin<NACC, BF16, RANDOM>), dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop_per = BF16 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
    const double flops = (double)cus * wgs_per_cu * 4 * iters * 4.0 * NACC * flop_per;
    if (rep == 2) printf("%s%s nacc=%d wg/cu=%d: %.3f ms  %.1f TFLOP/s\n", name, RANDOM ? " random operands" : "", NACC, wgs_per_cu, ms, flops / ms * 1e-9);
  }
  hipFree(out);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("CUs %d clock %d MHz\n", cus, p.clockRate / 1000);
  run<4, false>("f32 32x32x2", 1, cus);
  run<4, false>("f32 32x32x2", 2, cus);
  run<1, false>("f32 32x32x2", 1, cus);
  run<4, true>("bf16 32x32x16", 1, cus);
  run<4, true>("bf16 32x32x16", 2, cus);
  run<1, true>("bf16 32x32x16", 1, cus);
  run<4, false, true>("f32 32x32x2", 1, cus);
  run<4, false, true>("f32 32x32x2", 2, cus);
  run<4, true, true>("bf16 32x32x16", 1, cus);
  run<4, true, true>("bf16 32x32x16", 2, cus);
  return 0;
}
