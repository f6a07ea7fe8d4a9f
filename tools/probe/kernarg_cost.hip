// Host cost of hipLaunchKernelGGL as a function of the kernel-argument size (struct passed by value).
//   hipcc --offload-arch=gfx950 -O2 tools/probe/kernarg_cost.hip -o /tmp/kernarg_cost && /tmp/kernarg_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
template <int N> struct Blob { int v[N]; };
template <int N> __global__ void k(Blob<N> b, int* out) { if (threadIdx.x == 0 && b.v[N - 1] == 12345) out[0] = 1; }
template <int N> void run(int* d) {
  Blob<N> b; for (int i = 0; i < N; ++i) b.v[i] = i;
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, d);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  const int reps = 2000;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, d);
  auto t1 = std::chrono::steady_clock::now();
  hipDeviceSynchronize();
  auto t2 = std::chrono::steady_clock::now();
  printf("kernarg %5d B: issue %.2f us / launch, drained %.2f us / launch\n", (int)sizeof(Blob<N>),
         std::chrono::duration<double, std::micro>(t1 - t0).count() / reps,
         std::chrono::duration<double, std::micro>(t2 - t0).count() / reps);
}
int main() {
  int* d; hipMalloc(&d, 4);
  run<4>(d); run<64>(d); run<256>(d); run<512>(d); run<768>(d); run<1000>(d);
  return 0;
}
