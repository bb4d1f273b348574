// Probe (round 4): what the chip's power budget gives the two bf16 MFMA shapes -- 32x32x16 (this repo's tile kernels) and
// 16x16x32 (the vendor GEMM's) -- alone and with the fragment traffic of a 128 x 128 wave tile (ds_read_b128 of random LDS data
// feeding the operands: 8 reads per 16 MFMAs 32x32x16 / 16 reads per 64 MFMAs 16x16x32, the same bytes).  One wave per SIMD, 256
// workgroups, 256 accumulator registers.  Prints TF/s and the shader clock the chip held (s_memtime delta / wall time).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_power_probe.hip -o mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// accumulators a[0:255] owned by inline asm (hipcc shuffles 64 f32x4 accumulators through v_accvgpr_mov otherwise)
#define MD_A16(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
__device__ __forceinline__ void acc_reserve() {
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", MD_A16(1), MD_A16(2), MD_A16(3), MD_A16(4),
               MD_A16(5), MD_A16(6), MD_A16(7), MD_A16(8), MD_A16(9), MD_A16(10), MD_A16(11), MD_A16(12), MD_A16(13), MD_A16(14),
               MD_A16(15), MD_A16(16), MD_A16(17), MD_A16(18), MD_A16(19), MD_A16(20), MD_A16(21), MD_A16(22), MD_A16(23),
               MD_A16(24), "a250", "a251", "a252", "a253", "a254", "a255");
}
template <int X, bool FIRST>
__device__ __forceinline__ void mfma32(const bf16x8& a, const bf16x8& b) {
  if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(a), "v"(b), "i"(16 * X), "i"(16 * X + 15));
  else asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(16 * X), "i"(16 * X + 15));
}
template <int X, bool FIRST>
__device__ __forceinline__ void mfma16(const bf16x8& a, const bf16x8& b) {
  if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(a), "v"(b), "i"(4 * X), "i"(4 * X + 3));
  else asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(4 * X), "i"(4 * X + 3));
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

__device__ __forceinline__ bf16x8 rnd(unsigned s) {
  bf16x8 v;
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u;
    v[i] = (__bf16)(((int)(s >> 20) - 2048) * (1.0f / 2048.0f));
  }
  return v;
}
__device__ __forceinline__ void fill_lds(char* smem) {
  for (int i = threadIdx.x; i < 65536 / 16; i += 256) ((bf16x8*)smem)[i] = rnd(i * 977 + blockIdx.x);
  __syncthreads();
}
__device__ __forceinline__ bf16x8 lds_frag(const char* smem, unsigned off) { return *(const bf16x8*)(smem + (off & 0xfff0)); }

template <bool LDS>
__global__ __launch_bounds__(256) void k32(int iters, unsigned long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (LDS) fill_lds(smem);
  acc_reserve();
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = rnd(threadIdx.x * 7 + i); b[i] = rnd(threadIdx.x * 13 + 100 + i); }
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned off = threadIdx.x * 16;
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = lds_frag(smem, off + i * 4096); b[i] = lds_frag(smem, off + 16384 + i * 4096); }
      off += 1040;
    }
    if (it == 0) static_for<0, 16>([&](auto x) { constexpr int X = decltype(x)::value; mfma32<X, true>(a[X / 4], b[X % 4]); });
    else static_for<0, 16>([&](auto x) { constexpr int X = decltype(x)::value; mfma32<X, false>(a[X / 4], b[X % 4]); });
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  float s;
  asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a0" : "=v"(s));
  if (s == 12345.678f) sink[0] = s;
}

template <bool LDS>
__global__ __launch_bounds__(256) void k16(int iters, unsigned long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (LDS) fill_lds(smem);
  acc_reserve();
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = rnd(threadIdx.x * 7 + i); b[i] = rnd(threadIdx.x * 13 + 100 + i); }
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned off = threadIdx.x * 16;
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { a[i] = lds_frag(smem, off + i * 2048); b[i] = lds_frag(smem, off + 16384 + i * 2048); }
      off += 1040;
    }
    if (it == 0) static_for<0, 64>([&](auto x) { constexpr int X = decltype(x)::value; mfma16<X, true>(a[X / 8], b[X % 8]); });
    else static_for<0, 64>([&](auto x) { constexpr int X = decltype(x)::value; mfma16<X, false>(a[X / 8], b[X % 8]); });
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  float s;
  asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a0" : "=v"(s));
  if (s == 12345.678f) sink[0] = s;
}

int main() {
  float* sink; hipMalloc(&sink, 4);
  unsigned long long* cyc; hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k32<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 3; ++rep) {
    for (int v = 0; v < 4; ++v) {
      const int shape = (v & 1) ? 16 : 32;
      const bool lds = v >= 2;
      const int iters = shape == 32 ? 20000 : 5000;  // 16 x 32768 vs 64 x 16384 FLOPs per iteration
      auto launch = [&] {
        if (v == 0) hipLaunchKernelGGL(k32<false>, dim3(256), dim3(256), 0, 0, iters, cyc, sink);
        if (v == 1) hipLaunchKernelGGL(k16<false>, dim3(256), dim3(256), 0, 0, iters, cyc, sink);
        if (v == 2) hipLaunchKernelGGL(k32<true>, dim3(256), dim3(256), 65536, 0, iters, cyc, sink);
        if (v == 3) hipLaunchKernelGGL(k16<true>, dim3(256), dim3(256), 65536, 0, iters, cyc, sink);
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      const double flops = (shape == 32 ? 16.0 * 32768 : 64.0 * 16384) * iters * 4 * 256;
      const double mf = (shape == 32 ? 16.0 : 64.0) * iters;  // MFMAs per wave
      printf("MFMA %dx%dx%d bf16 %-18s %7.2f ms  %7.1f TF/s  | %.1f cycles per MFMA, shader clock %.3f GHz\n", shape, shape, shape == 32 ? 16 : 32,
             lds ? "+ fragment reads:" : "only:", ms, flops / ms / 1e9, (double)c / mf, (double)c / (ms * 1e6));
    }
  }
  return 0;
}
