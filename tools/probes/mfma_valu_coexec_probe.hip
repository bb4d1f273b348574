// Probe (round 4): do MFMA and VALU work of DIFFERENT waves on one SIMD run concurrently on gfx950, and what does a transcendental cost?
// 256 workgroups x 512 threads = two waves per SIMD (wave w and w + 4 share SIMD w % 4).  Role A = waves 0-3, role B = waves 4-7;
// a role is one of: idle, a loop of independent v_mfma_f32_32x32x16_bf16 (4 accumulators), a loop of v_exp_f32, a loop of v_fma_f32,
// the softmax mix of the prefill attention (32 exp + 96 full-rate VALU per 22 MFMAs).  Prints each combination's time: if A+B together
// take max(A, B) the pipes overlap across waves, if they take A + B they do not.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_coexec_probe.hip -o mfma_valu_coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { IDLE = 0, MFMA = 1, EXP = 2, FMA = 3, MIX = 4, MFMA_AGPR = 5 };

__device__ __forceinline__ void role_mfma(int iters, float* sink) {
  f32x16 acc[4];
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x % 37 + i)); b[i] = (__bf16)(0.02f * (threadIdx.x % 29 + i)); }
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
  }
  float s = 0;
  for (int j = 0; j < 4; ++j) s += acc[j][0];
  if (s == 12345.678f) sink[0] = s;
}
// the same loop with the accumulators in the ACCUMULATION register file (constraint "a")
__device__ __forceinline__ void role_mfma_agpr(int iters, float* sink) {
  f32x16 acc[4];
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x % 37 + i)); b[i] = (__bf16)(0.02f * (threadIdx.x % 29 + i)); }
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a), "v"(b));
  }
  float s = 0;
  for (int j = 0; j < 4; ++j) s += acc[j][0];
  if (s == 12345.678f) sink[0] = s;
}
template <int NEXP, int NFMA>
__device__ __forceinline__ void role_valu(int iters, float* sink) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 0.001f * (threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < NEXP / 8; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
#pragma unroll
    for (int u = 0; u < NFMA / 8; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 12345.678f) sink[0] = s;
}
__device__ __forceinline__ void run_role(int role, int iters, float* sink) {
  if (role == MFMA) role_mfma(iters, sink);             // 16 MFMAs per iteration = 512 matrix-pipe cycles
  else if (role == MFMA_AGPR) role_mfma_agpr(iters, sink);
  else if (role == EXP) role_valu<64, 0>(iters, sink);  // 64 v_exp_f32 per iteration
  else if (role == FMA) role_valu<0, 128>(iters, sink); // 128 v_fma_f32 per iteration = 512 cycles at full rate
  else if (role == MIX) role_valu<24, 72>(iters, sink); // the attention loop's ratio (32 exp + 96 others per 22 MFMAs), scaled to 16 MFMAs
}
__global__ __launch_bounds__(512) void k(int role_a, int role_b, int iters, float* sink) {
  const int wave = threadIdx.x >> 6;
  run_role(wave < 4 ? role_a : role_b, iters, sink);
}

int main() {
  float* sink; hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* nm[6] = {"idle", "mfma", "exp", "fma", "mix", "mfmaA"};
  const int combos[][2] = {{MFMA, IDLE}, {EXP, IDLE}, {FMA, IDLE}, {MIX, IDLE}, {MFMA, MFMA}, {EXP, EXP}, {FMA, FMA}, {MFMA, EXP}, {MFMA, FMA}, {MFMA, MIX}, {MIX, MIX}, {MFMA_AGPR, IDLE}, {MFMA_AGPR, EXP}, {MFMA_AGPR, FMA}, {MFMA_AGPR, MIX}, {MFMA_AGPR, MFMA_AGPR}};
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep)
    for (auto& c : combos) {
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, c[0], c[1], iters, sink); hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, c[0], c[1], iters, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
      printf("A=%-5s B=%-5s %8.3f ms  = %6.1f ns per iteration\n", nm[c[0]], nm[c[1]], ms, ms * 1e6 / iters);
    }
  return 0;
}
