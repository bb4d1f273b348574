// Decode attention over an FP8 (OCP e4m3fn) copy of the KV cache -- part of the opt-in fp8 mode (BASELINE configs[4]).
//
// At B = 64 the decode step is bound by the K / V bytes it reads (the bf16 kernel runs at 5.6 TB/s and is 58 % of the
// step): only fewer bytes help.  The cache keeps its bf16 slabs (prefill attention, EncodedImage snapshots and the
// reference-compatible views read those); this mode adds e4m3 slabs of the same [layer][slot][head][position][64]
// shape with ONE static scale per layer for K and one for V (value ~= scale * fp8), written by
//   * kv_quantize_kernel after a prefill (positions pos0[b] .. pos0[b] + q_len - 1 of every layer), and
//   * the decode attention itself for the new token's row (RoPE + cache update fused in, like attn_decode_kernel<true>),
// and read by attn_decode_f8_kernel: one workgroup per (sequence, head), four waves, a 64-byte row = 4 lanes x 16 B
// (16-byte loads: 8-byte ones run at 0.54-0.70x the rate), two streaming passes (scores -> LDS, then P.V), the K scale
// folded into q and the V scale into the output.  The newest key / value enter as what the cache will hold for them
// (quantised, dequantised), so a step sees the same numbers as every later step.
// Tolerance-judged like the rest of the fp8 mode; no bit-compatibility with the bf16 kernel is claimed.
#include "md_common.hpp"

#include <algorithm>

namespace {

constexpr int F8_MAX_CTX = 2048;

typedef float f32x2v __attribute__((ext_vector_type(2)));
// 16 e4m3 bytes -> 16 floats
__device__ __forceinline__ void unpack16(const u32x4& w, float (&f)[16]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2v lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[i], false);
    const f32x2v hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[i], true);
    f[4 * i + 0] = lo[0];
    f[4 * i + 1] = lo[1];
    f[4 * i + 2] = hi[0];
    f[4 * i + 3] = hi[1];
  }
}
__device__ __forceinline__ float deq1(uint32_t byte) {  // one e4m3 byte (low 8 bits) -> float
  return __builtin_amdgcn_cvt_pk_f32_fp8((int)byte, false)[0];
}

// bf16 slab rows -> e4m3 slab rows of ONE layer: positions pos0[b] .. pos0[b] + n_pos - 1 of every (slot, head); one
// 16-byte bf16 chunk (8 features) per thread
__global__ __launch_bounds__(256) void kv_quantize_kernel(const bf16_t* __restrict__ ks, const bf16_t* __restrict__ vs, uint8_t* __restrict__ k8,
                                                          uint8_t* __restrict__ v8, float ik, float iv, const int32_t* __restrict__ pos0,
                                                          int pos_fixed, int64_t batch_stride, int ctx, int batch, int n_heads, int n_pos) {
  const int64_t total = (int64_t)batch * n_heads * n_pos * 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t r = i;
    const int ch = (int)(r & 7);
    r >>= 3;
    const int t = (int)(r % n_pos);
    r /= n_pos;
    const int h = (int)(r % n_heads), b = (int)(r / n_heads);
    const int pos = (pos0 ? pos0[b] : pos_fixed) + t;
    if (pos >= ctx) continue;
    const int64_t off = (int64_t)b * batch_stride + ((int64_t)h * ctx + pos) * 64 + ch * 8;
    const u32x4 kq = *(const u32x4*)(ks + off), vq = *(const u32x4*)(vs + off);
    u32x2 ko, vo;
    ko[0] = pack_fp8x4(lo_bf(kq[0]) * ik, hi_bf(kq[0]) * ik, lo_bf(kq[1]) * ik, hi_bf(kq[1]) * ik);
    ko[1] = pack_fp8x4(lo_bf(kq[2]) * ik, hi_bf(kq[2]) * ik, lo_bf(kq[3]) * ik, hi_bf(kq[3]) * ik);
    vo[0] = pack_fp8x4(lo_bf(vq[0]) * iv, hi_bf(vq[0]) * iv, lo_bf(vq[1]) * iv, hi_bf(vq[1]) * iv);
    vo[1] = pack_fp8x4(lo_bf(vq[2]) * iv, hi_bf(vq[2]) * iv, lo_bf(vq[3]) * iv, hi_bf(vq[3]) * iv);
    *(u32x2*)(k8 + off) = ko;
    *(u32x2*)(v8 + off) = vo;
  }
}

// One workgroup (4 waves) per (sequence, head).  Lane l: row group g = l >> 2 (16 per wave, 64 per workgroup), 16-byte
// chunk c = l & 3 of the 64-byte row.  Key j of round i, slot u (of UNR): j = 64 (i + u) + 16 wave + g.
constexpr int UNR = 8;  // rows requested together per lane: 8 x 16 B in flight
__global__ __launch_bounds__(256) void attn_decode_f8_kernel(const bf16_t* __restrict__ qkv, int64_t ld, bf16_t* __restrict__ o, int64_t ldo,
                                                             const float* __restrict__ freqs, bf16_t* __restrict__ kslab, bf16_t* __restrict__ vslab,
                                                             uint8_t* __restrict__ k8slab, uint8_t* __restrict__ v8slab, int64_t slab_bs, int ctx,
                                                             const int32_t* __restrict__ kv_len_p, int n_heads, float scale_log2, int rot,
                                                             float k_scale, float v_scale) {
  __shared__ float sc[F8_MAX_CTX];
  __shared__ float red[64][64 + 1];  // [row group][feature | sum of p]
  __shared__ float red_m[4];
  __shared__ float newq[64];                                      // rotated q (fp32 of its bf16 value)
  __shared__ __attribute__((aligned(16))) uint8_t new8[2][64];    // the new token's K and V rows as stored in the fp8 cache
  __shared__ __attribute__((aligned(16))) bf16_t newbf[2][64];    // and as stored in the bf16 cache

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 2, c = lane & 3;
  const int b = blockIdx.y, h = blockIdx.x;
  const int kv_len = kv_len_p[b], pos = kv_len - 1;
  const int64_t head_off = (int64_t)b * slab_bs + (int64_t)h * ctx * 64;
  const uint8_t* kb = k8slab + head_off;
  const uint8_t* vb = v8slab + head_off;

  {  // RoPE of q and k, pass-through features, v: the new token's rows (text.py:35-46, rope.py:20-48)
    const bf16_t* row = qkv + (int64_t)b * ld;
    const int half = rot >> 1;
    if (tid < 2 * half) {
      const int which = tid / half, j = tid % half;
      const bf16_t* hp = row + (which ? (n_heads + h) : h) * 64;
      const float re = bf2f(hp[j]), im = bf2f(hp[half + j]);
      const float cs = freqs[((int64_t)pos * half + j) * 2], sn = freqs[((int64_t)pos * half + j) * 2 + 1];
      float o_re, o_im;
      md_rope_pair(re, im, cs, sn, o_re, o_im);
      if (which == 0) {
        newq[2 * j] = bf2f(f2bf(o_re));
        newq[2 * j + 1] = bf2f(f2bf(o_im));
      } else {
        newbf[0][2 * j] = f2bf(o_re);
        newbf[0][2 * j + 1] = f2bf(o_im);
      }
    } else if (tid >= 64 && tid < 64 + 2 * (64 - rot)) {
      const int t2 = tid - 64, which = t2 / (64 - rot), i = rot + t2 % (64 - rot);
      const bf16_t x = row[(which ? (n_heads + h) : h) * 64 + i];
      if (which == 0) newq[i] = bf2f(x); else newbf[0][i] = x;
    } else if (tid >= 192 && tid < 256) {
      newbf[1][tid - 192] = row[(2 * n_heads + h) * 64 + tid - 192];
    }
    __syncthreads();
    if (tid < 32) {  // 2 rows x 16 words of 4 features: quantise, keep in LDS, write both caches
      const int which = tid >> 4, w4 = tid & 15;
      const float inv = 1.0f / (which ? v_scale : k_scale);
      const bf16_t* src = &newbf[which][4 * w4];
      const uint32_t q = pack_fp8x4(bf2f(src[0]) * inv, bf2f(src[1]) * inv, bf2f(src[2]) * inv, bf2f(src[3]) * inv);
      *(uint32_t*)(&new8[which][4 * w4]) = q;
      *(uint32_t*)((which ? v8slab : k8slab) + head_off + (int64_t)pos * 64 + 4 * w4) = q;
    } else if (tid >= 64 && tid < 128) {
      kslab[head_off + (int64_t)pos * 64 + tid - 64] = newbf[0][tid - 64];
    } else if (tid >= 128 && tid < 192) {
      vslab[head_off + (int64_t)pos * 64 + tid - 128] = newbf[1][tid - 128];
    }
    __syncthreads();
  }

  float qv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) qv[e] = newq[16 * c + e] * (scale_log2 * k_scale);

  // ---- pass 1: scores --------------------------------------------------------------------------------------------
  float mx = -INFINITY;
  for (int i0 = 0; i0 * 64 < kv_len; i0 += UNR) {
    u32x4 kq[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = 64 * (i0 + u) + 16 * wave + g;
      kq[u] = __builtin_nontemporal_load((const u32x4*)(kb + (int64_t)min(j, pos) * 64 + c * 16));  // read once per step (attention.hip: NT)
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = 64 * (i0 + u) + 16 * wave + g;
      // (the newest key is not yet visible in global memory to this CU: take it from LDS)
      const u32x4 kk = (j == pos) ? *(const u32x4*)(&new8[0][c * 16]) : kq[u];
      float kf[16];
      unpack16(kk, kf);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) s += qv[e] * kf[e];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      if (j < kv_len) {
        if (c == 0) sc[j] = s;
        mx = fmaxf(mx, s);
      }
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red_m[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));

  // ---- pass 2: probabilities and P.V -----------------------------------------------------------------------------
  float acc[16], l = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int i0 = 0; i0 * 64 < kv_len; i0 += UNR) {
    u32x4 vq[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = 64 * (i0 + u) + 16 * wave + g;
      vq[u] = __builtin_nontemporal_load((const u32x4*)(vb + (int64_t)min(j, pos) * 64 + c * 16));
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = 64 * (i0 + u) + 16 * wave + g;
      const u32x4 vv = (j == pos) ? *(const u32x4*)(&new8[1][c * 16]) : vq[u];
      const float pj = (j < kv_len) ? __builtin_amdgcn_exp2f(sc[min(j, pos)] - mx) : 0.f;
      l += pj;
      const float pr = bf2f(f2bf(pj));  // probabilities enter the second contraction as bf16, as in the bf16 kernel
      float vf[16];
      unpack16(vv, vf);
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] += pr * vf[e];
    }
  }
  const int rg = 16 * wave + g;
#pragma unroll
  for (int e = 0; e < 16; ++e) red[rg][16 * c + e] = acc[e];
  if (c == 0) red[rg][64] = l;
  __syncthreads();
  if (tid < 64) {
    float s = 0.f, lt = 0.f;
    for (int r = 0; r < 64; ++r) {
      s += red[r][tid];
      lt += red[r][64];
    }
    o[(int64_t)b * ldo + h * 64 + tid] = f2bf(lt > 0.f ? s * v_scale / lt : 0.f);
  }
}

}  // namespace

// internal (api.hip)
md_status md_kv_quantize_f8_layer(const md_kv_cache* kv, int layer, const int32_t* pos0, int pos_fixed, int batch, int n_heads, int n_pos,
                                  hipStream_t s) {
  MD_CHECK_ARG(kv && kv->k && kv->v && kv->k8 && kv->v8 && kv->k_scale && kv->v_scale && kv->k_scale[layer] > 0.f && kv->v_scale[layer] > 0.f);
  const int64_t total = (int64_t)batch * n_heads * n_pos * 8;
  if (total <= 0) return MD_OK;
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 32);
  const int64_t lo = (int64_t)layer * kv->layer_stride;
  hipLaunchKernelGGL(kv_quantize_kernel, dim3(blocks), dim3(256), 0, s, (const bf16_t*)kv->k + lo, (const bf16_t*)kv->v + lo,
                     (uint8_t*)kv->k8 + lo, (uint8_t*)kv->v8 + lo, 1.0f / kv->k_scale[layer], 1.0f / kv->v_scale[layer], pos0, pos_fixed,
                     kv->batch_stride, kv->ctx, batch, n_heads, n_pos);
  return md_launch_status();
}

extern "C" md_status md_kv_quantize_f8(const md_kv_cache* kv, int32_t n_layers, int32_t batch, int32_t n_heads, const int32_t* pos0,
                                       int32_t pos_fixed, int32_t n_pos, void* stream) {
  MD_CHECK_ARG(kv && n_layers > 0 && batch > 0 && n_heads > 0 && n_pos > 0 && pos_fixed >= 0);
  for (int l = 0; l < n_layers; ++l) MD_TRY(md_kv_quantize_f8_layer(kv, l, pos0, pos_fixed, batch, n_heads, n_pos, (hipStream_t)stream));
  return MD_OK;
}

md_status md_attention_decode_rope_f8_launch(const void* qkv, int64_t ld, void* o, int64_t ldo, const float* freqs, void* k_slab, void* v_slab,
                                             void* k8_slab, void* v8_slab, int64_t slab_batch_stride, int32_t ctx, const int32_t* kv_len,
                                             int32_t batch, int32_t n_heads, int32_t rot_dim, float scale, float k_scale, float v_scale,
                                             hipStream_t s) {
  MD_CHECK_ARG(qkv && o && freqs && k_slab && v_slab && k8_slab && v8_slab && kv_len);
  MD_CHECK_ARG(ctx <= F8_MAX_CTX && batch > 0 && n_heads > 0 && rot_dim % 2 == 0 && rot_dim > 0 && rot_dim <= 64);
  MD_CHECK_ARG(ld % 8 == 0 && ldo % 8 == 0 && ld >= 3 * n_heads * 64 && ldo >= n_heads * 64 && k_scale > 0.f && v_scale > 0.f);
  hipLaunchKernelGGL(attn_decode_f8_kernel, dim3(n_heads, batch), dim3(256), 0, s, (const bf16_t*)qkv, ld, (bf16_t*)o, ldo, freqs,
                     (bf16_t*)k_slab, (bf16_t*)v_slab, (uint8_t*)k8_slab, (uint8_t*)v8_slab, slab_batch_stride, ctx, kv_len, n_heads,
                     scale * 1.4426950408889634f, rot_dim, k_scale, v_scale);
  return md_launch_status();
}

extern "C" md_status md_attention_decode_rope_f8(const void* qkv, int64_t ld, void* o, int64_t ldo, const float* freqs, void* k_slab,
                                                 void* v_slab, void* k8_slab, void* v8_slab, int64_t slab_batch_stride, int32_t ctx,
                                                 const int32_t* kv_len, int32_t batch, int32_t n_heads, int32_t rot_dim, float scale,
                                                 float k_scale, float v_scale, void* stream) {
  return md_attention_decode_rope_f8_launch(qkv, ld, o, ldo, freqs, k_slab, v_slab, k8_slab, v8_slab, slab_batch_stride, ctx, kv_len, batch,
                                            n_heads, rot_dim, scale, k_scale, v_scale, (hipStream_t)stream);
}
