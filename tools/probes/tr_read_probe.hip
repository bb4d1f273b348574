// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (which element lands in which lane).
// Build: hipcc --offload-arch=gfx950 -O2 tr_read_probe.hip -o tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short sm[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = (short)i;
  __syncthreads();
  // lane l points at elements 4l .. 4l+3 (8 bytes)
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm + threadIdx.x * 4));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) {
      printf(" %4d", h[l * 4 + j]);
      if (h[l * 4 + j] != (l & 15) + j * 16 + (l >> 4) * 64) ++bad;
    }
    printf("\n");
  }
  printf("mismatches vs lds[(l&15) + 16 j + 64 (l>>4)]: %d\n", bad);
  return 0;
}
