// Probe: per-CU throughput of L2-resident streams through (a) LDS-DMA (global_load_lds_dwordx4)
// and (b) plain global_load_dwordx4 into registers, 1 workgroup per CU.
// Build: hipcc --offload-arch=gfx950 -O3 load_path_probe.hip -o load_path_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(NT) void dma_stream(const char* __restrict__ src, size_t wg_bytes, int reps, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src + (size_t)blockIdx.x * wg_bytes;
  const int per_iter = NT * 16;  // bytes per wave-set per issue
  const int ring = 8;            // issues in flight
  unsigned acc = 0;
  for (int r = 0; r < reps; ++r) {
    for (size_t off = 0; off < wg_bytes; off += (size_t)per_iter * ring) {
#pragma unroll
      for (int j = 0; j < ring; ++j) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off + (size_t)j * per_iter + tid * 16),
                                         (__attribute__((address_space(3))) void*)(smem + j * per_iter + wave * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  acc = *(unsigned*)(smem + tid * 4);
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int NT>
__global__ __launch_bounds__(NT) void reg_stream(const char* __restrict__ src, size_t wg_bytes, int reps, unsigned* sink) {
  const int tid = threadIdx.x;
  const char* base = src + (size_t)blockIdx.x * wg_bytes;
  const int per_iter = NT * 16;
  const int ring = 8;
  u32x4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    for (size_t off = 0; off < wg_bytes; off += (size_t)per_iter * ring) {
      u32x4 v[ring];
#pragma unroll
      for (int j = 0; j < ring; ++j) v[j] = *(const u32x4*)(base + off + (size_t)j * per_iter + tid * 16);
#pragma unroll
      for (int j = 0; j < ring; ++j) acc ^= v[j];
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = acc[0];
}

template <class F>
float time_ms(F&& f, int iters) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main() {
  const int n_wg = 256;
  const size_t wg_bytes = 64 << 10;  // 64 KiB per workgroup -> 16 MiB total: L2 (4 MiB/XCD x 8) + MALL resident
  const int reps = 64;
  char* src; unsigned* sink;
  (void)hipMalloc(&src, n_wg * wg_bytes); (void)hipMemset(src, 1, n_wg * wg_bytes); (void)hipMalloc(&sink, 64);
  (void)hipFuncSetAttribute((const void*)dma_stream<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  (void)hipFuncSetAttribute((const void*)dma_stream<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  (void)hipFuncSetAttribute((const void*)dma_stream<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const double total = (double)n_wg * wg_bytes * reps;
  auto report = [&](const char* name, float ms) {
    printf("%-34s %8.3f ms  %7.2f TB/s  %6.1f GB/s per CU\n", name, ms, total / ms / 1e9, total / ms / 1e6 / n_wg);
  };
  report("lds-dma 128 thr (2 waves)", time_ms([&] { hipLaunchKernelGGL(dma_stream<128>, dim3(n_wg), dim3(128), 8 * 128 * 16, 0, src, wg_bytes, reps, sink); }, 5));
  report("lds-dma 256 thr (4 waves)", time_ms([&] { hipLaunchKernelGGL(dma_stream<256>, dim3(n_wg), dim3(256), 8 * 256 * 16, 0, src, wg_bytes, reps, sink); }, 5));
  report("lds-dma 512 thr (8 waves)", time_ms([&] { hipLaunchKernelGGL(dma_stream<512>, dim3(n_wg), dim3(512), 8 * 512 * 16, 0, src, wg_bytes, reps, sink); }, 5));
  report("global_load 128 thr", time_ms([&] { hipLaunchKernelGGL(reg_stream<128>, dim3(n_wg), dim3(128), 0, 0, src, wg_bytes, reps, sink); }, 5));
  report("global_load 256 thr", time_ms([&] { hipLaunchKernelGGL(reg_stream<256>, dim3(n_wg), dim3(256), 0, 0, src, wg_bytes, reps, sink); }, 5));
  report("global_load 512 thr", time_ms([&] { hipLaunchKernelGGL(reg_stream<512>, dim3(n_wg), dim3(512), 0, 0, src, wg_bytes, reps, sink); }, 5));
  return 0;
}
