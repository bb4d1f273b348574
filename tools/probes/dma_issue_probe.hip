// Probe (round 4): what an LDS-DMA issue costs a wave that is alone on its SIMD and feeding the matrix pipe with
// v_mfma_f32_16x16x32_bf16 (16-cycle gaps).  One "pair" = 128 MFMAs with 16 operand loads spread every 8th gap, as in the
// four-wave GEMM; 4 waves per workgroup, 256 workgroups, the loads hit a small L2-resident buffer.  Variants of the load:
//   0 none   1 s_add m0 + s_nop + buffer_load_dwordx4 lds (the GEMM's form)   2 M0 written once per pair, the piece selected by the
//   instruction offset   3 form 1 without the s_nop   4 buffer_load_dword lds (a quarter of the bytes)   5 buffer_load_dwordx4 into
//   registers (no LDS)   6 global_load_lds_dwordx4 (64-bit per-lane addresses)   7 form 1, two pieces back to back every 16th gap
//   8-11 global_load_lds_dwordx4 with a scalar base + 32-bit lane offset, M0 written per piece / per pair / per four pieces
// Prints shader cycles per pair (s_memtime) and the implied cost per load.
// Build: hipcc --offload-arch=gfx950 -O3 dma_issue_probe.hip -o dma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define MD_A16(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
__device__ __forceinline__ void acc_reserve() {
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", MD_A16(1), MD_A16(2), MD_A16(3), MD_A16(4),
               MD_A16(5), MD_A16(6), MD_A16(7), MD_A16(8), MD_A16(9), MD_A16(10), MD_A16(11), MD_A16(12), MD_A16(13), MD_A16(14),
               MD_A16(15), MD_A16(16), MD_A16(17), MD_A16(18), MD_A16(19), MD_A16(20), MD_A16(21), MD_A16(22), MD_A16(23),
               MD_A16(24), "a250", "a251", "a252", "a253", "a254", "a255");
}
template <int X>
__device__ __forceinline__ void mfma16_first(const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(a), "v"(b), "i"(4 * X), "i"(4 * X + 3));
}
template <int X>
__device__ __forceinline__ void mfma16(const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "i"(4 * X), "i"(4 * X + 3));
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
__device__ __forceinline__ bf16x8 rnd(unsigned s) {
  bf16x8 v;
  for (int i = 0; i < 8; ++i) { s = s * 1664525u + 1013904223u; v[i] = (__bf16)(((int)(s >> 20) - 2048) * (1.0f / 2048.0f)); }
  return v;
}

template <int V>
__global__ __launch_bounds__(256) void k(int iters, const char* src, unsigned long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  acc_reserve();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = rnd(threadIdx.x * 7 + i); b[i] = rnd(threadIdx.x * 13 + 100 + i); }
  static_for<0, 64>([&](auto x) { constexpr int X = decltype(x)::value; mfma16_first<X>(a[X / 8], b[X % 8]); });
  u32x4 rs;
  rs[0] = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)src);
  rs[1] = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)src >> 32));
  rs[2] = 0xffffffffu; rs[3] = 0x00020000u;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 1024;
  unsigned voff[16];
  for (int j = 0; j < 16; ++j) voff[j] = ((blockIdx.x * 4 + wave) * 16 + j) * 1024 % (4 << 20) + lane * 16;
  unsigned soff = 0;
  u32x4 r[4];
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (V == 2 || V == 9 || V == 10) asm volatile("s_mov_b32 m0, %0" ::"s"(lds0) : "memory");
    static_for<0, 128>([&](auto x) {
      constexpr int X = decltype(x)::value, M = X % 64;
      mfma16<M>(a[M / 8], b[M % 8]);
      __builtin_amdgcn_sched_barrier(0);
      constexpr bool SLOT = (V == 7) ? (X % 16 == 3) : (X % 8 == 3);
      if constexpr (SLOT && V != 0) {
        constexpr int Q = (V == 7) ? 2 * (X / 16) : X / 8;
        const unsigned vo = voff[Q], vo2 = voff[(Q + 1) & 15], so = soff, base = lds0;
        const u32x4 rsv = rs;
        if constexpr (V == 1) asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds" ::"s"(base), "i"(Q * 4096), "v"(vo), "s"(rsv), "s"(so) : "memory", "scc");
        if constexpr (V == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(vo), "s"(rsv), "s"(so), "i"((Q & 3) * 1024) : "memory");
        if constexpr (V == 3) asm volatile("s_add_u32 m0, %0, %1\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds" ::"s"(base), "i"(Q * 4096), "v"(vo), "s"(rsv), "s"(so) : "memory", "scc");
        if constexpr (V == 4) asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, %4 offen lds" ::"s"(base), "i"(Q * 4096), "v"(vo), "s"(rsv), "s"(so) : "memory", "scc");
        if constexpr (V == 5) { u32x4 t; asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(t) : "v"(vo), "s"(rsv), "s"(so) : "memory"); r[Q & 3] = t; }
        if constexpr (V == 6) {
          const char* gp = src + vo;
          asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off" ::"s"(base), "i"(Q * 4096), "v"(gp) : "memory", "scc");
        }
        if constexpr (V == 8) asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" ::"s"(base), "i"(Q * 4096), "v"(vo), "s"(src) : "memory", "scc");
        if constexpr (V == 9) asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(vo), "s"(src), "i"((Q & 3) * 1024) : "memory");
        if constexpr (V == 10) { const char* gp = src + vo; asm volatile("global_load_lds_dwordx4 %0, off offset:%1" ::"v"(gp), "i"((Q & 3) * 1024) : "memory"); }
        if constexpr (V == 11) {  // saddr form, M0 written once per four pieces
          if constexpr ((Q & 3) == 0) asm volatile("s_add_u32 m0, %0, %1" ::"s"(base), "i"(Q * 4096) : "memory", "scc");
          asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(vo), "s"(src), "i"((Q & 3) * 1024) : "memory");
        }
        if constexpr (V == 7) {
          asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds" ::"s"(base), "i"(Q * 4096), "v"(vo), "s"(rsv), "s"(so) : "memory", "scc");
          asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds" ::"s"(base), "i"((Q + 1) * 4096), "v"(vo2), "s"(rsv), "s"(so) : "memory", "scc");
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    soff = (soff + 128) & 0xfff;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  float s;
  asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a0" : "=v"(s));
  if (V == 5) s += __uint_as_float(r[0][0] ^ r[1][1] ^ r[2][2] ^ r[3][3]);
  if (s == 12345.678f) sink[0] = s;
}

template <int V>
void run(int iters, const char* src, unsigned long long* cyc, float* sink, double base, const char* name) {
  hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 65536 + 4096, 0, iters, src, cyc, sink); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 65536 + 4096, 0, iters, src, cyc, sink); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per = (double)c / iters;
  printf("%-78s %7.0f cycles per pair  (+%5.1f per load)  clock %.2f GHz\n", name, per, V ? (per - base) / 16.0 : 0.0, (double)c / (ms * 1e6));
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float* sink; hipMalloc(&sink, 4);
  unsigned long long* cyc; hipMalloc(&cyc, 8);
  char* src; hipMalloc(&src, 8 << 20); hipMemset(src, 1, 8 << 20);
  const int iters = 3000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 65536 + 4096, 0, iters, src, cyc, sink); hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double base = (double)c / iters;
    run<0>(iters, src, cyc, sink, base, "0 no loads");
    run<1>(iters, src, cyc, sink, base, "1 s_add m0 + s_nop + buffer_load_dwordx4 lds");
    run<2>(iters, src, cyc, sink, base, "2 m0 once per pair, piece by instruction offset");
    run<3>(iters, src, cyc, sink, base, "3 form 1 without s_nop");
    run<4>(iters, src, cyc, sink, base, "4 buffer_load_dword lds");
    run<5>(iters, src, cyc, sink, base, "5 buffer_load_dwordx4 to registers");
    run<6>(iters, src, cyc, sink, base, "6 global_load_lds_dwordx4");
    run<7>(iters, src, cyc, sink, base, "7 form 1, two pieces back to back every 16th gap");
    run<8>(iters, src, cyc, sink, base, "8 s_add m0 + s_nop + global_load_lds_dwordx4 v, s[base] (scalar base + 32-bit offset)");
    run<9>(iters, src, cyc, sink, base, "9 form 8, m0 once per pair, piece by instruction offset");
    run<10>(iters, src, cyc, sink, base, "10 global_load_lds_dwordx4 v[0:1], off, m0 once per pair, piece by instruction offset");
    run<11>(iters, src, cyc, sink, base, "11 form 8, m0 written once per FOUR pieces (s_add in the slot of the first)");
  }
  return 0;
}
