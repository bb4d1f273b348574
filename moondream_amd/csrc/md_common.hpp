// Shared device/host helpers for the gfx950 kernels.  CDNA4 only: wave64,
// v_cvt_pk_bf16_f32, MFMA 32x32x16 bf16, LDS-DMA (global_load_lds_dwordx4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <unordered_map>

#include "../../include/moondream_hip.h"

typedef uint16_t bf16_t;  // raw bf16 bits everywhere; arithmetic is fp32

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;  // one 32x32 accumulator tile
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define MD_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even; lowers to v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}

// two roundings in ONE v_cvt_pk_bf16_f32 (converting the halves separately costs
// two conversions plus a shift and an or per pair)
typedef __bf16 md_bf16x2 __attribute__((ext_vector_type(2)));
typedef float md_f32pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const md_f32pair v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, md_bf16x2));
}

__device__ __forceinline__ float lo_bf(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// GELU(tanh) in the algebraically equal sigmoid form
//   0.5 x (1 + tanh(u)) = x / (1 + exp(-2u)),  u = sqrt(2/pi) (x + 0.044715 x^3)
// (reference: layers.py:24-25, F.gelu(approximate="tanh") evaluated in fp32).  No
// cancellation for large |u|; -2 log2(e) sqrt(2/pi) is folded into the cubic so the
// whole thing is 3 multiplies/FMAs, one v_exp_f32, one add, one v_rcp_f32, one multiply
// (an IEEE division here costs ~10 more VALU instructions per element and showed up as
// ~40 % of the GELU layers' tile time).  |x| large: exp2 -> inf or 0, rcp -> 0 or 1.
constexpr float kGeluA = -2.0f * 1.4426950408889634f * 0.7978845608028654f;
constexpr float kGeluB = kGeluA * 0.044715f;
__device__ __forceinline__ float gelu_tanh_f32(float x) {
  const float z = x * __builtin_fmaf(kGeluB, x * x, kGeluA);
  const float e = __builtin_amdgcn_exp2f(z);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// two elements at a time: the polynomial and the final product in packed fp32 (v_pk_*)
typedef float md_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ md_f32x2 gelu_tanh_f32x2(md_f32x2 x) {
  const md_f32x2 a = {kGeluA, kGeluA}, b = {kGeluB, kGeluB}, one = {1.0f, 1.0f};
  const md_f32x2 z = x * __builtin_elementwise_fma(b, x * x, a);
  md_f32x2 d = {__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
  d = d + one;
  const md_f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  return x * r;
}

// One RoPE pair exactly as torch evaluates it (rope.py:40-45: xq_r * cos - xq_i * sin, xq_r * sin + xq_i * cos, each an
// elementwise fp32 op): FOUR separately rounded products, then one subtraction and one addition.  The files are built
// with -ffp-contract=fast, and __fmul_rn / __fsub_rn do not stop hipcc from contracting a product into the add (round 3
// found v_pk_fma_f32 in every RoPE site: one rounding fewer than the reference, visible as rare last-bit flips after the
// bf16 rounding); the empty asm makes each product a value of its own.
__device__ __forceinline__ float md_mul_rounded(float a, float b) {
  float p = a * b;
  asm volatile("" : "+v"(p));
  return p;
}
__device__ __forceinline__ void md_rope_pair(float re, float im, float cs, float sn, float& o_re, float& o_im) {
  const float a = md_mul_rounded(re, cs), b = md_mul_rounded(im, sn), c = md_mul_rounded(re, sn), d = md_mul_rounded(im, cs);
  o_re = a - b;
  o_im = c + d;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// bf16 -> OCP e4m3 (the opt-in FP8 mode's activation producers: quant_f8.hip, the prefill attention's fp8 output)
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
  const float lim = 448.0f;  // e4m3fn's largest finite value; saturate instead of producing NaN
  a = __builtin_amdgcn_fmed3f(a, -lim, lim);
  b = __builtin_amdgcn_fmed3f(b, -lim, lim);
  c = __builtin_amdgcn_fmed3f(c, -lim, lim);
  d = __builtin_amdgcn_fmed3f(d, -lim, lim);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}
__device__ __forceinline__ u32x2 quant8(const u32x4& q, float s) {
  u32x2 o;
  o[0] = pack_fp8x4(lo_bf(q[0]) * s, hi_bf(q[0]) * s, lo_bf(q[1]) * s, hi_bf(q[1]) * s);
  o[1] = pack_fp8x4(lo_bf(q[2]) * s, hi_bf(q[2]) * s, lo_bf(q[3]) * s, hi_bf(q[3]) * s);
  return o;
}

// XCD-aware remap of a linear workgroup id: hardware places block b on XCD b % 8,
// so give every XCD a contiguous chunk of the logical tile sequence (neighbouring
// tiles share operand panels -> hit the same private L2).  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int xcd = bid % nx, idx = bid / nx;
  int q = nwg / nx, r = nwg % nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

#define MD_CHECK_ARG(cond)                  \
  do {                                      \
    if (!(cond)) return MD_ERR_INVALID_ARG; \
  } while (0)

#define MD_TRY(expr)               \
  do {                             \
    md_status _s = (expr);         \
    if (_s != MD_OK) return _s;    \
  } while (0)

// Raise a kernel's dynamic-LDS limit once per (kernel, device): the attribute belongs to the function ON the
// current device, so a process that drives several GPUs needs it on each of them.
inline md_status md_ensure_dynamic_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::unordered_map<const void*, uint64_t> done;  // kernel -> bit per device ordinal
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return MD_ERR_LAUNCH;
  const uint64_t bit = 1ull << (dev & 63);
  std::lock_guard<std::mutex> g(mu);
  uint64_t& bits = done[fn];
  if (bits & bit) return MD_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return MD_ERR_LAUNCH;
  }
  bits |= bit;
  return MD_OK;
}

static inline md_status md_launch_status() {
  return hipGetLastError() == hipSuccess ? MD_OK : MD_ERR_LAUNCH;
}
