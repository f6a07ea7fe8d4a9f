// Probe (not product): times pyg_hip_segment_matmul on the C2 shape straight through the C-ABI (no torch),
// once per environment setting given on the command line.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/c2_mm.cpp -o tools/probe/c2_mm -Iinclude -Lpyg_lib_amd -lpyg_hip -Wl,-rpath,'$ORIGIN/../../pyg_lib_amd'
//   tools/probe/c2_mm [rows] [K] [M] [B] -- "" "PYG_HIP_MM_WGS=1" "PYG_HIP_MM_CHUNK=8" ...
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "pyg_hip.h"

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

__global__ void fill_bf16(uint16_t* p, size_t n, float scale, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t h = ((uint32_t)i + seed) * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    // sum of two uniforms, roughly bell shaped in [-2, 2) * scale
    float u = ((h & 0xffff) + (h >> 16)) * (1.0f / 32768.0f) - 2.0f;
    uint32_t b = __float_as_uint(u * scale);
    p[i] = (uint16_t)((b + 0x7fffu + ((b >> 16) & 1)) >> 16);
  }
}

int main(int argc, char** argv) {
  long rows = 21111007L, K = 128, M = 128, B = 154;
  int ai = 1;
  std::vector<long*> pos = {&rows, &K, &M, &B};
  for (size_t i = 0; i < pos.size() && ai < argc && strcmp(argv[ai], "--") != 0; ++i, ++ai) *pos[i] = atol(argv[ai]);
  if (ai < argc && strcmp(argv[ai], "--") == 0) ++ai;
  std::vector<std::string> envs;
  for (; ai < argc; ++ai) envs.push_back(argv[ai]);
  if (envs.empty()) envs.push_back("");

  uint16_t *x, *w, *out;
  CK(hipMalloc(&x, (size_t)rows * K * 2));
  CK(hipMalloc(&w, (size_t)B * K * M * 2));
  CK(hipMalloc(&out, (size_t)rows * M * 2));
  hipLaunchKernelGGL(fill_bf16, dim3(2048), dim3(256), 0, 0, x, (size_t)rows * K, 1.0f, 1u);
  hipLaunchKernelGGL(fill_bf16, dim3(256), dim3(256), 0, 0, w, (size_t)B * K * M, 0.09f, 77u);
  CK(hipDeviceSynchronize());
  // same split as bench.py would give is not needed here: sizes proportional to a fixed hash
  std::vector<int64_t> ptr(B + 1, 0);
  {
    std::vector<double> f(B);
    double s = 0;
    uint32_t h = 12345;
    for (long b = 0; b < B; ++b) {
      h = h * 1664525u + 1013904223u;
      f[b] = (h >> 8) / 16777216.0 + 1e-3;
      s += f[b];
    }
    long acc = 0;
    for (long b = 0; b < B; ++b) {
      acc += (long)(f[b] / s * rows);
      ptr[b + 1] = acc;
    }
    ptr[B] = rows;
  }
  const size_t wsb = pyg_hip_matmul_workspace_size(B);
  void* ws;
  CK(hipMalloc(&ws, wsb));
  const double bytes = 2.0 * ((double)rows * K + (double)rows * M + (double)B * K * M) + 8.0 * (B + 1);
  const double flops = 2.0 * rows * K * M;
  for (const std::string& e : envs) {
    // "A=1,B=2" -> setenv each; names are remembered so the next setting starts clean
    std::vector<std::string> names;
    size_t p0 = 0;
    while (p0 < e.size()) {
      size_t p1 = e.find(',', p0);
      if (p1 == std::string::npos) p1 = e.size();
      std::string kv = e.substr(p0, p1 - p0);
      size_t eq = kv.find('=');
      if (eq != std::string::npos) {
        setenv(kv.substr(0, eq).c_str(), kv.substr(eq + 1).c_str(), 1);
        names.push_back(kv.substr(0, eq));
      }
      p0 = p1 + 1;
    }
    for (int i = 0; i < 3; ++i) {
      int rc = pyg_hip_segment_matmul(PYG_BF16, x, ptr.data(), 0, w, nullptr, out, rows, K, M, B, ws, wsb, 0, nullptr);
      if (rc != 0) {
        printf("[%s] error %d: %s\n", e.c_str(), rc, pyg_hip_last_error());
        break;
      }
    }
    CK(hipDeviceSynchronize());
    pyg_hip_profile_enable(1);
    const int reps = 10;
    for (int i = 0; i < reps; ++i)
      pyg_hip_segment_matmul(PYG_BF16, x, ptr.data(), 0, w, nullptr, out, rows, K, M, B, ws, wsb, 0, nullptr);
    float ms[reps];
    int n = pyg_hip_profile_collect(ms, reps);
    pyg_hip_profile_enable(0);
    float best = 1e30f, sum = 0;
    for (int i = 0; i < n && i < reps; ++i) {
      best = ms[i] < best ? ms[i] : best;
      sum += ms[i];
    }
    const float mean = n ? sum / (n < reps ? n : reps) : 0.f;
    // checksum of a slice so variants can be compared for equality
    std::vector<uint16_t> hs(1 << 16);
    CK(hipMemcpy(hs.data(), out + (size_t)(rows / 2) * M, hs.size() * 2, hipMemcpyDeviceToHost));
    uint64_t ck = 1469598103934665603ull;
    for (uint16_t v : hs) ck = (ck ^ v) * 1099511628211ull;
    printf("[%-40s] %-28s best %.3f ms %.2f TB/s | mean %.3f ms %.2f TB/s %.1f TFLOP/s  ck %016llx\n", e.c_str(),
           pyg_hip_matmul_last_variant(), best, bytes / best * 1e-9, mean, bytes / mean * 1e-9, flops / mean * 1e-9,
           (unsigned long long)ck);
    fflush(stdout);
    for (auto& nme : names) unsetenv(nme.c_str());
  }
  return 0;
}
