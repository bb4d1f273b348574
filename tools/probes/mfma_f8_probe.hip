// v_mfma_scale_f32_32x32x64_f8f6f4 with OCP e4m3 operands (the large-K fp8 MFMA of gfx950; there is no unscaled form):
//   1. operand layout: which (row, k) of A and which (k, column) of B byte b of lane l holds -- found by multiplying
//      one-hot operands against operands whose values encode the index (the layout is not in the guides of this image);
//   2. issue rate against v_mfma_f32_32x32x16_bf16 (same bytes per operand register pair, twice the K): a dependent
//      chain per accumulator block, 4 blocks per wave, 4 waves per CU -- the same shape as the four-wave GEMM's inner loop.
// Groundwork for an fp8-activation big-tile GEMM (BASELINE configs[4]); nothing in the product uses it yet.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f8_probe mfma_f8_probe.hip && ./mfma_f8_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// e4m3fn: value v = 2^(e-7) (1 + m/8); 1.0 = 0x38; small integers n = 1..15 are exact
static uint8_t e4m3_of_int(int n) {
  if (n == 0) return 0;
  int e = 0;
  while ((n >> (e + 1)) != 0) ++e;          // n in [2^e, 2^(e+1))
  const int m = ((n << 3) >> e) & 7;        // exact for n < 16
  return (uint8_t)(((e + 7) << 3) | m);
}

// C = A . B with both operands given per lane as 32 bytes; scales = 1.0 (E8M0 127 in byte 0)
__global__ void one_mfma(const uint8_t* a_bytes, const uint8_t* b_bytes, float* c) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = ((const int*)(a_bytes + lane * 32))[i];
    b[i] = ((const int*)(b_bytes + lane * 32))[i];
  }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 0x7f, 0, 0x7f);
  for (int i = 0; i < 16; ++i) c[lane * 16 + i] = acc[i];
}

template <bool F8>
__global__ __launch_bounds__(256) void rate_loop(float* out, int iters, unsigned seed) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) {  // random-looking bit patterns (finite e4m3 / bf16 values): the power state depends on the data
    a[i] = (int)(((threadIdx.x * 2654435761u) ^ (seed + i * 40503u)) & 0x3f3f3f3fu);
    b[i] = (int)(((threadIdx.x * 40503u) ^ (seed * 3u + i * 2654435761u)) & 0x3f3f3f3fu);
  }
  bf16x8 ah, bh, ah2, bh2;
  for (int i = 0; i < 8; ++i) {
    ah[i] = (short)(a[i >> 1] >> (16 * (i & 1)));
    bh[i] = (short)(b[i >> 1] >> (16 * (i & 1)));
    ah2[i] = (short)(a[4 + (i >> 1)] >> (16 * (i & 1)));
    bh2[i] = (short)(b[4 + (i >> 1)] >> (16 * (i & 1)));
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (F8) {
        acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 0, 0, 0, 0x7f, 0, 0x7f);
      } else {  // the same operand bytes as four bf16 K steps of 16
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah2, bh2, acc[j], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 16; ++i) s += acc[j][i];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  // ---- layout ---------------------------------------------------------------------------------------------------
  // C/D of a 32x32 block: lane l, register r -> column l & 31, row (r & 3) + 8 (r >> 2) + 4 (l >> 5)  (guide, dtype-independent)
  uint8_t *da, *db;
  float* dc;
  hipMalloc(&da, 64 * 32);
  hipMalloc(&db, 64 * 32);
  hipMalloc(&dc, 64 * 16 * 4);
  std::vector<uint8_t> ha(64 * 32), hb(64 * 32);
  std::vector<float> hc(64 * 16);
  auto run = [&]() {
    hipMemcpy(da, ha.data(), ha.size(), hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), hb.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, dc);
    hipMemcpy(hc.data(), dc, hc.size() * 4, hipMemcpyDeviceToHost);
  };
  auto C = [&](int row, int col) { return hc[(col + 32 * ((row >> 2) & 1)) * 16 + (row & 3) + 4 * (row >> 3)]; };
  // Hypothesis H: the FIRST operand's byte b of lane l is element (m = l & 31, k = 32 (l >> 5) + b) and likewise the
  // second operand's (k = 32 (l >> 5) + b, n = l & 31); D[m][n] = sum_k first[m][k] second[k][n] with the D map above
  // taken as (row = m? or n?).  One-hot first operand at (lane la, byte ba), second operand = value (byte index + 1)
  // in the low bytes / (lane >> 5) + 1 ... : print what comes out and where.
  printf("one-hot first operand (lane, byte) x second operand whose byte b of lane l holds ((b & 7) + 1) [and a second pass: (b >> 3) + 1 + 4 (l >> 5)]\n");
  for (int pass = 0; pass < 2; ++pass) {
    for (int la : {0, 1, 31, 32, 63}) {
      for (int ba : {0, 1, 7, 8, 15, 16, 31}) {
        std::fill(ha.begin(), ha.end(), 0);
        ha[la * 32 + ba] = 0x38;  // 1.0
        for (int l = 0; l < 64; ++l)
          for (int b = 0; b < 32; ++b) hb[l * 32 + b] = e4m3_of_int(pass == 0 ? (b & 7) + 1 : (b >> 3) + 1 + 4 * (l >> 5));
        run();
        // non-zero entries of D
        int nz = 0, r0 = -1, c0 = -1;
        float v0 = 0.f;
        bool same_row = true, same_val = true;
        for (int r = 0; r < 32; ++r)
          for (int c = 0; c < 32; ++c)
            if (C(r, c) != 0.f) {
              if (nz == 0) { r0 = r; c0 = c; v0 = C(r, c); }
              else { same_row &= (r == r0); same_val &= (C(r, c) == v0); }
              ++nz;
            }
        printf("  pass %d  first[lane %2d][byte %2d] = 1  ->  %3d non-zeros, first at D[%2d][%2d] = %g, all in one row: %d, all equal: %d\n",
               pass, la, ba, nz, r0, c0, v0, (int)same_row, (int)same_val);
      }
    }
  }
  printf("one-hot second operand (lane, byte) x first operand = all ones: which column / row lights up\n");
  for (int lb : {0, 1, 31, 32, 63}) {
    for (int bb : {0, 9, 31}) {
      std::fill(hb.begin(), hb.end(), 0);
      hb[lb * 32 + bb] = 0x38;
      std::fill(ha.begin(), ha.end(), 0x38);
      run();
      int nz = 0, r0 = -1, c0 = -1;
      bool same_col = true, same_row = true;
      for (int r = 0; r < 32; ++r)
        for (int c = 0; c < 32; ++c)
          if (C(r, c) != 0.f) {
            if (nz == 0) { r0 = r; c0 = c; }
            else { same_col &= (c == c0); same_row &= (r == r0); }
            ++nz;
          }
      printf("  second[lane %2d][byte %2d] = 1 -> %3d non-zeros, first at D[%2d][%2d], one column: %d, one row: %d\n", lb, bb, nz, r0, c0,
             (int)same_col, (int)same_row);
    }
  }
  // ---- rate -------------------------------------------------------------------------------------------------------
  int dev = 0, ncu = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  float* dout;
  hipMalloc(&dout, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int f8 = 0; f8 < 2; ++f8) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      if (f8) hipLaunchKernelGGL(rate_loop<true>, dim3(ncu), dim3(256), 0, 0, dout, iters, 17u + rep);
      else hipLaunchKernelGGL(rate_loop<false>, dim3(ncu), dim3(256), 0, 0, dout, iters, 17u + rep);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      // per iteration and wave: 4 blocks x 2 * 32 * 32 * 64 flop (both variants: fp8 one K = 64 step, bf16 two K = 16 steps = K 32!)
      const double k_per_iter = f8 ? 64.0 : 32.0;
      const double flop = (double)ncu * 4 /*waves*/ * iters * 4 /*blocks*/ * 2.0 * 32 * 32 * k_per_iter;
      printf("%s: %.3f ms  %.1f TFLOP/s  (%d CUs x 4 waves, one wave per SIMD, 4 independent accumulator blocks)\n",
             f8 ? "mfma_scale_f32_32x32x64_f8f6f4 (e4m3, scale 1)" : "mfma_f32_32x32x16_bf16 x 2", ms, flop / (ms * 1e-3) / 1e12, ncu);
    }
  }
  return 0;
}
