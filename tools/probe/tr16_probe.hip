// Probe of gfx950's ds_read_b64_tr_b16 (LDS transpose read) lane/address semantics.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/tr16_probe.hip -o /tmp/tr16_probe && /tmp/tr16_probe
// Hypothesis: in every 16-lane group, lane q supplies the address of 4 contiguous b16 values = row (q>>2),
// columns (q&3)*4.. of a 4 x 16 matrix; lane q receives column q of that matrix (4 rows).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  const int q = l & 15, grp = l >> 4;
  const int pitch = 160;  // elements
  const short* p = lds + (q >> 2) * pitch + (q & 3) * 4 + grp * 16;
  v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d;
  short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int want = j * 160 + (l >> 4) * 16 + (l & 15);
      if (h[l * 4 + j] != want) ++bad;
    }
  printf("tr16 probe: %d mismatches\n", bad);
  for (int l = 0; l < 20; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
