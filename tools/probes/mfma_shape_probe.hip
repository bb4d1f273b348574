// Probe: the power-limited matrix-pipe rate of bf16 MFMA shapes with nothing else running -- 32x32x16 (this repo's kernels)
// against 16x16x32 (the vendor GEMM's shape): same FLOPs, same accumulator footprint (256 registers per wave), one wave per
// SIMD, 256 workgroups, random operands.  TF/s = 2 * M * N * K per MFMA x count / time.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_shape_probe.hip -o mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ bf16x8 rnd(unsigned s) {
  bf16x8 v;
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u;
    v[i] = (__bf16)(((int)(s >> 20) - 2048) * (1.0f / 1024.0f));
  }
  return v;
}

__global__ __launch_bounds__(256) void k32(int iters, float* sink) {
  f32x16 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x16{0};
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = rnd(threadIdx.x * 7 + i); b[i] = rnd(threadIdx.x * 13 + 100 + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 * i + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[4 * i + j], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  if (s == 12345.678f) sink[0] = s;
}

__global__ __launch_bounds__(256) void k16(int iters, float* sink) {
  f32x4 acc[64];
  for (int i = 0; i < 64; ++i) acc[i] = f32x4{0};
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = rnd(threadIdx.x * 7 + i); b[i] = rnd(threadIdx.x * 13 + 100 + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[8 * i + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[8 * i + j], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 64; ++i) s += acc[i][0];
  if (s == 12345.678f) sink[0] = s;
}

int main() {
  float* sink; hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    for (int shape : {32, 16}) {
      const int iters = shape == 32 ? 20000 : 10000;  // 16 x 32768 vs 64 x 16384 FLOPs per iteration
      auto launch = [&] { if (shape == 32) hipLaunchKernelGGL(k32, dim3(256), dim3(256), 0, 0, iters, sink); else hipLaunchKernelGGL(k16, dim3(256), dim3(256), 0, 0, iters, sink); };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      const double flops = (shape == 32 ? 16.0 * 32768 : 64.0 * 16384) * iters * 4 * 256;
      printf("MFMA %dx%dx%d bf16 only: %7.2f ms  %7.1f TF/s (%.1f %% of 2500)\n", shape, shape, shape == 32 ? 16 : 32, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0);
    }
  }
  return 0;
}
