// HBM-bound helpers around the GEMMs: LayerNorm, patch extraction, partial RoPE
// + KV-slab write, token embedding gather, greedy argmax, crop stitch + pool.
// All of them move 16 bytes per lane per access (8 bf16) and do their
// arithmetic in fp32 with exactly one rounding to bf16 per reference op.
#include "md_common.hpp"

#include <algorithm>

namespace {

// ---------------------------------------------------------------------------
// LayerNorm: one wave per row, the row lives in registers between the two
// reductions (reference: layers.py:118-119 -> F.layer_norm; biased variance).
// ---------------------------------------------------------------------------
template <int NCH>  // 16-byte chunks per lane; covers dim <= 512 * NCH
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                        bf16_t* __restrict__ y, int64_t ldy,
                                                        const bf16_t* __restrict__ w,
                                                        const bf16_t* __restrict__ bia, int rows,
                                                        int dim, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = dim >> 3;
  const bf16_t* xr = x + (int64_t)row * ldx;
  float v[NCH][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    u32x4 q = {0, 0, 0, 0};
    if (ch < nchunk) q = *(const u32x4*)(xr + ch * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[i][2 * e] = lo_bf(q[e]);
      v[i][2 * e + 1] = hi_bf(q[e]);
      sum += v[i][2 * e] + v[i][2 * e + 1];
    }
  }
  const float mean = wave_sum(sum) / (float)dim;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[i][e] - mean;
        ss += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)dim + eps);
  bf16_t* yr = y + (int64_t)row * ldy;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nchunk) {
      const u32x4 wq = *(const u32x4*)(w + ch * 8);
      const u32x4 bq = *(const u32x4*)(bia + ch * 8);
      u32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = (v[i][2 * e] - mean) * rstd * lo_bf(wq[e]) + lo_bf(bq[e]);
        const float c = (v[i][2 * e + 1] - mean) * rstd * hi_bf(wq[e]) + hi_bf(bq[e]);
        out[e] = pack_bf16x2(a, c);
      }
      *(u32x4*)(yr + ch * 8) = out;
    }
  }
}

// ---------------------------------------------------------------------------
// Decode-regime block tail: the launch-boundary reduction of the two split-K
// linears that feed the residual stream, their bias / residual epilogues, and the
// NEXT block's layer norm, one workgroup per row:
//   t1 = bf16(sum_s A[s] + bias_a)   x1 = bf16(x + t1)        (proj,  text.py:53,157)
//   t2 = bf16(sum_s B[s] + bias_b)   x  = bf16(x1 + t2)       (fc2,   text.py:158)
//   y  = layer_norm(x) with the next block's weights         (text.py:145), optional
// -- the roundings of md_gemm_bf16's MD_EPI_RESIDUAL epilogue, in the same order; the
// slices are summed in index order, so the result depends on the layer shape only.
// ---------------------------------------------------------------------------
template <int NCH>  // 16-byte chunks per thread; covers dim <= 2048 * NCH
__global__ __launch_bounds__(256) void reduce_residual_ln_kernel(
    bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ pa, int sa,
    const bf16_t* __restrict__ bias_a, const float* __restrict__ pb, int sb,
    const bf16_t* __restrict__ bias_b, int64_t ldp, int64_t slice_stride, bf16_t* __restrict__ y,
    int64_t ldy, const bf16_t* __restrict__ lnw, const bf16_t* __restrict__ lnb, int dim, float eps) {
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = blockIdx.x;
  const int nchunk = dim >> 3;
  bf16_t* xr = x + (int64_t)row * ldx;
  float v[NCH][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = tid + 256 * i;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    if (ch < nchunk) {
      // every slice of both operands is requested before the first add (one exposed memory
      // latency instead of one per slice); the sums run in slice order
      constexpr int MAXS = 8;
      f32x4 la[MAXS][2], lb[MAXS][2];
      const float* qa = pa + (int64_t)row * ldp + ch * 8;
      const float* qb = pb + (int64_t)row * ldp + ch * 8;
#pragma unroll
      for (int s = 0; s < MAXS; ++s) {
        if (s < sa) {
          la[s][0] = *(const f32x4*)(qa + (int64_t)s * slice_stride);
          la[s][1] = *(const f32x4*)(qa + (int64_t)s * slice_stride + 4);
        }
        if (s < sb) {
          lb[s][0] = *(const f32x4*)(qb + (int64_t)s * slice_stride);
          lb[s][1] = *(const f32x4*)(qb + (int64_t)s * slice_stride + 4);
        }
      }
      float acc_a[8], acc_b[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc_a[e] = acc_b[e] = 0.f;
#pragma unroll
      for (int s = 0; s < MAXS; ++s) {
        if (s < sa) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc_a[e] += la[s][0][e];
            acc_a[4 + e] += la[s][1][e];
          }
        }
        if (s < sb) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc_b[e] += lb[s][0][e];
            acc_b[4 + e] += lb[s][1][e];
          }
        }
      }
      const u32x4 xq = *(const u32x4*)(xr + ch * 8);
      const u32x4 ba = *(const u32x4*)(bias_a + ch * 8);
      const u32x4 bb = *(const u32x4*)(bias_b + ch * 8);
      u32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t t1 = pack_bf16x2(acc_a[2 * e] + lo_bf(ba[e]), acc_a[2 * e + 1] + hi_bf(ba[e]));
        const uint32_t x1 = pack_bf16x2(lo_bf(xq[e]) + lo_bf(t1), hi_bf(xq[e]) + hi_bf(t1));
        const uint32_t t2 = pack_bf16x2(acc_b[2 * e] + lo_bf(bb[e]), acc_b[2 * e + 1] + hi_bf(bb[e]));
        out[e] = pack_bf16x2(lo_bf(x1) + lo_bf(t2), hi_bf(x1) + hi_bf(t2));
        v[i][2 * e] = lo_bf(out[e]);
        v[i][2 * e + 1] = hi_bf(out[e]);
        sum += v[i][2 * e] + v[i][2 * e + 1];
      }
      *(u32x4*)(xr + ch * 8) = out;
    }
  }
  if (y == nullptr) return;  // uniform
  sum = wave_sum(sum);
  if (lane == 0) red[0][wave] = sum;
  __syncthreads();
  const float mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)dim;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (tid + 256 * i < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[i][e] - mean;
        ss += d * d;
      }
    }
  }
  ss = wave_sum(ss);
  if (lane == 0) red[1][wave] = ss;
  __syncthreads();
  const float rstd = rsqrtf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)dim + eps);
  bf16_t* yr = y + (int64_t)row * ldy;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = tid + 256 * i;
    if (ch < nchunk) {
      const u32x4 wq = *(const u32x4*)(lnw + ch * 8);
      const u32x4 bq = *(const u32x4*)(lnb + ch * 8);
      u32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = (v[i][2 * e] - mean) * rstd * lo_bf(wq[e]) + lo_bf(bq[e]);
        const float c = (v[i][2 * e + 1] - mean) * rstd * hi_bf(wq[e]) + hi_bf(bq[e]);
        out[e] = pack_bf16x2(a, c);
      }
      *(u32x4*)(yr + ch * 8) = out;
    }
  }
}

// ---------------------------------------------------------------------------
// Patch extraction.  One workgroup per patch row of the GEMM operand:
// out[(n*G*G + gy*G + gx)][c*P*P + py*P + px]   (reference: vision.py:44-61)
// ---------------------------------------------------------------------------
template <bool FROM_U8>
__global__ __launch_bounds__(256) void patchify_kernel(const void* __restrict__ src,
                                                       const bf16_t* __restrict__ lut,
                                                       bf16_t* __restrict__ out, int64_t ld_out,
                                                       int crop, int patch, int grid) {
  const int prow = blockIdx.x;
  const int n = prow / (grid * grid), gy = (prow / grid) % grid, gx = prow % grid;
  const int pp = patch * patch, feat = 3 * pp;
  bf16_t* o = out + (int64_t)prow * ld_out;
  for (int f = threadIdx.x; f < ld_out; f += 256) {
    bf16_t val = 0;
    if (f < feat) {
      const int c = f / pp, py = (f % pp) / patch, px = f % patch;
      const int y = gy * patch + py, xx = gx * patch + px;
      if (FROM_U8) {
        const uint8_t* s = (const uint8_t*)src;
        val = lut[s[(((int64_t)n * crop + y) * crop + xx) * 3 + c]];
      } else {
        const bf16_t* s = (const bf16_t*)src;
        val = s[(((int64_t)n * 3 + c) * crop + y) * crop + xx];
      }
    }
    o[f] = val;
  }
}

// ---------------------------------------------------------------------------
// Partial RoPE + KV write.  One workgroup per token row of the fused qkv
// activation.  (reference: rope.py:20-48, text.py:35-46, moondream.py:74-78)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_kv_kernel(bf16_t* __restrict__ qkv, int64_t ld,
                                                      const float* __restrict__ freqs,
                                                      const int32_t* __restrict__ pos0,
                                                      bf16_t* __restrict__ kslab,
                                                      bf16_t* __restrict__ vslab, int64_t slab_bs,
                                                      int ctx, int q_len, int n_heads,
                                                      int n_kv_heads, int hd, int rot) {
  const int tok = blockIdx.x;
  const int b = tok / q_len, t = tok % q_len;
  const int pos = pos0[b] + t;
  bf16_t* row = qkv + (int64_t)tok * ld;
  const int half = rot >> 1;
  const float* fr = freqs + (int64_t)pos * half * 2;
  const int n_rot_heads = n_heads + n_kv_heads;

  // rotated pairs: thread -> (head, j).  Input is half-split (re = x[j],
  // im = x[half + j]); output is interleaved (x'[2j], x'[2j+1]).  All inputs of
  // a head are read before any output is written: the unit of work is a whole
  // head per group of `half` consecutive threads and the barrier below
  // separates the phases.
  const int items = n_rot_heads * half;
  float o_re[4], o_im[4];  // up to 4 items per thread (items <= 1024 checked on the host)

#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int it = threadIdx.x + 256 * u;
    if (it >= items) break;
    const int head = it / half, j = it % half;
    const bf16_t* hp = row + head * hd;
    const float re = bf2f(hp[j]), im = bf2f(hp[half + j]);
    const float c = fr[2 * j], s = fr[2 * j + 1];
    // separate fp32 roundings, as torch evaluates mul, mul, sub / add
    md_rope_pair(re, im, c, s, o_re[u], o_im[u]);
  }
  __syncthreads();

#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int it = threadIdx.x + 256 * u;
    if (it >= items) break;
    const int head = it / half, j = it % half;
    const uint32_t w = pack_bf16x2(o_re[u], o_im[u]);
    if (head < n_heads) {
      *(uint32_t*)(row + head * hd + 2 * j) = w;  // q stays in the activation, rotated in place
    } else {
      const int hk = head - n_heads;
      *(uint32_t*)(kslab + (int64_t)b * slab_bs + ((int64_t)hk * ctx + pos) * hd + 2 * j) = w;
    }
  }
  // pass-through half of k, and all of v, in 16-byte chunks
  const int cpk = (hd - rot) >> 3, cpv = hd >> 3;
  for (int it = threadIdx.x; it < n_kv_heads * cpk; it += 256) {
    const int hk = it / cpk, ch = it % cpk;
    const u32x4 q = *(const u32x4*)(row + (n_heads + hk) * hd + rot + ch * 8);
    *(u32x4*)(kslab + (int64_t)b * slab_bs + ((int64_t)hk * ctx + pos) * hd + rot + ch * 8) = q;
  }
  for (int it = threadIdx.x; it < n_kv_heads * cpv; it += 256) {
    const int hk = it / cpv, ch = it % cpv;
    const u32x4 q = *(const u32x4*)(row + (n_heads + n_kv_heads + hk) * hd + ch * 8);
    *(u32x4*)(vslab + (int64_t)b * slab_bs + ((int64_t)hk * ctx + pos) * hd + ch * 8) = q;
  }
}

// ---------------------------------------------------------------------------
// Row-wise bf16 helpers of the LoRA side path: out = bf16(a + b) (the reference's bf16 tensor adds,
// layers.py:132-143, text.py:31-32,55-56,158) and out = gelu_tanh(a) as a stand-alone op.
// rows x cols with leading dimensions; cols % 8 == 0.
template <int OP>  // 0: add, 1: gelu
__global__ __launch_bounds__(256) void rowwise_kernel(const bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ b,
                                                      int64_t ldb, bf16_t* __restrict__ out, int64_t ldo, int rows, int cols) {
  const int cpr = cols >> 3;
  const int64_t total = (int64_t)rows * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / cpr), c = (int)(i % cpr) * 8;
    const u32x4 x = *(const u32x4*)(a + (int64_t)r * lda + c);
    u32x4 o;
    if constexpr (OP == 0) {
      const u32x4 y = *(const u32x4*)(b + (int64_t)r * ldb + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(lo_bf(x[e]) + lo_bf(y[e]), hi_bf(x[e]) + hi_bf(y[e]));
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const md_f32x2 g = gelu_tanh_f32x2(md_f32x2{lo_bf(x[e]), hi_bf(x[e])});
        o[e] = pack_bf16x2(g[0], g[1]);
      }
    }
    *(u32x4*)(out + (int64_t)r * ldo + c) = o;
  }
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(const int32_t* __restrict__ ids,
                                                    const bf16_t* __restrict__ table, int64_t ldt,
                                                    bf16_t* __restrict__ out, int64_t ldo, int dim) {
  const int i = blockIdx.x;
  const bf16_t* src = table + (int64_t)ids[i] * ldt;
  bf16_t* dst = out + (int64_t)i * ldo;
  for (int ch = threadIdx.x; ch < (dim >> 3); ch += 256)
    *(u32x4*)(dst + ch * 8) = *(const u32x4*)(src + ch * 8);
}

// greedy argmax, ties -> lowest index; optionally bumps pos[b]
__global__ __launch_bounds__(1024) void argmax_kernel(const bf16_t* __restrict__ logits, int64_t ld,
                                                     int vocab, int suppress, int32_t* __restrict__ next,
                                                     int32_t* __restrict__ pos_inc) {
  __shared__ float sv[16];
  __shared__ int si[16];
  const int b = blockIdx.x;
  const bf16_t* lr = logits + (int64_t)b * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  // 16 waves per row: the single-sequence decode step waits on this kernel's dependent-load chain
  for (int ch = threadIdx.x; ch < (vocab >> 3); ch += 1024) {
    const u32x4 q = *(const u32x4*)(lr + ch * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = ch * 8 + e;
      float v = (e & 1) ? hi_bf(q[e >> 1]) : lo_bf(q[e >> 1]);
      if (idx == suppress) v = -INFINITY;
      if (v > best || (v == best && idx < bi)) {
        best = v;
        bi = idx;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = best;
    si[threadIdx.x >> 6] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
        best = sv[w];
        bi = si[w];
      }
    // an all-NaN row selects nothing: fall back to id 0 rather than an out-of-range id that the
    // next md_embed_tokens would use as a table index
    next[b] = (bi >= vocab) ? 0 : bi;
    if (pos_inc) pos_inc[b] += 1;
  }
}

// ---------------------------------------------------------------------------
// stitch + adaptive average pool + concat.  grid (g*g, n_images).
// stitched(y, x) comes from local crop (ty, tx) at (y - ty*inner, x - tx*inner)
// where a crop keeps its interior plus the outer margin on image borders
// (reference: image_crops.py:170-231 with patch_size=1); pooling bin i covers
// [floor(i*H/g), ceil((i+1)*H/g)) (reference: vision.py:83-86).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stitch_pool_kernel(const bf16_t* __restrict__ feats,
                                                          const bf16_t* __restrict__ grid_in,
                                                          bf16_t* __restrict__ out, int64_t ld_out,
                                                          int64_t out_img_stride, int dim, int g,
                                                          int margin, int th, int tw, int H_in,
                                                          int W_in) {
  // grid_in != nullptr: the local features are an ALREADY stitched [H_in][W_in][dim]
  // grid (the reference seam's _vis_proj(g, r) form) and feats is the global crop only.
  const int p = blockIdx.x, img = blockIdx.y;
  const int pi = p / g, pj = p % g;
  const int inner = g - 2 * margin;
  const int H = grid_in ? H_in : inner * th + 2 * margin;
  const int W = grid_in ? W_in : inner * tw + 2 * margin;
  const int y0 = (pi * H) / g, y1 = ((pi + 1) * H + g - 1) / g;
  const int x0 = (pj * W) / g, x1 = ((pj + 1) * W + g - 1) / g;
  const int64_t crop_sz = (int64_t)g * g * dim;
  const bf16_t* fimg = feats + (grid_in ? 0 : (int64_t)img * (1 + th * tw) * crop_sz);
  bf16_t* o = out + (int64_t)img * out_img_stride + (int64_t)p * ld_out;
  // sum / count, the form torch's adaptive_avg_pool2d uses
  const float cnt = (float)((y1 - y0) * (x1 - x0));
  for (int ch = threadIdx.x; ch < (dim >> 3); ch += 256) {
    // global crop features
    *(u32x4*)(o + ch * 8) = *(const u32x4*)(fimg + (int64_t)p * dim + ch * 8);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int y = y0; y < y1; ++y) {
      const int ty = (y < margin) ? 0 : min((y - margin) / inner, th - 1);
      const int ly = y - ty * inner;
      for (int x = x0; x < x1; ++x) {
        const bf16_t* s;
        if (grid_in) {
          s = grid_in + ((int64_t)y * W + x) * dim + ch * 8;
        } else {
          const int tx = (x < margin) ? 0 : min((x - margin) / inner, tw - 1);
          const int lx = x - tx * inner;
          s = fimg + (int64_t)(1 + ty * tw + tx) * crop_sz + ((int64_t)ly * g + lx) * dim + ch * 8;
        }
        const u32x4 q = *(const u32x4*)s;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[2 * e] += lo_bf(q[e]);
          acc[2 * e + 1] += hi_bf(q[e]);
        }
      }
    }
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = pack_bf16x2(acc[2 * e] / cnt, acc[2 * e + 1] / cnt);
    *(u32x4*)(o + dim + ch * 8) = w;
  }
}

}  // namespace

extern "C" md_status md_layernorm_bf16(const void* x, int64_t ldx, void* y, int64_t ldy,
                                       const md_layernorm* p, int32_t rows, int32_t dim, float eps,
                                       void* stream) {
  MD_CHECK_ARG(x && y && p && p->w && p->b && rows > 0);
  MD_CHECK_ARG(dim % 8 == 0 && dim > 0 && dim <= 4096 && ldx % 8 == 0 && ldy % 8 == 0);
  MD_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)p->w | (uintptr_t)p->b) & 15) == 0);
  dim3 grid((rows + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  const bf16_t* xx = (const bf16_t*)x;
  bf16_t* yy = (bf16_t*)y;
  const bf16_t* w = (const bf16_t*)p->w;
  const bf16_t* b = (const bf16_t*)p->b;
  if (dim <= 512)
    hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, xx, ldx, yy, ldy, w, b, rows, dim, eps);
  else if (dim <= 1536)
    hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, s, xx, ldx, yy, ldy, w, b, rows, dim, eps);
  else if (dim <= 2560)
    hipLaunchKernelGGL(layernorm_kernel<5>, grid, block, 0, s, xx, ldx, yy, ldy, w, b, rows, dim, eps);
  else
    hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, s, xx, ldx, yy, ldy, w, b, rows, dim, eps);
  return md_launch_status();
}

extern "C" md_status md_reduce_residual_layernorm(void* x, int64_t ldx, const float* partial_a, int32_t slices_a,
                                                  const void* bias_a, const float* partial_b, int32_t slices_b,
                                                  const void* bias_b, int64_t ld_partial, int64_t slice_stride,
                                                  void* y, int64_t ldy, const md_layernorm* ln, int32_t rows,
                                                  int32_t dim, float eps, void* stream) {
  MD_CHECK_ARG(x && partial_a && partial_b && bias_a && bias_b && rows > 0 && dim > 0 && dim % 8 == 0);
  MD_CHECK_ARG(slices_a >= 1 && slices_b >= 1 && slices_a <= 8 && slices_b <= 8 && ldx % 8 == 0 && ldx >= dim && ld_partial % 4 == 0 && ld_partial >= dim);
  MD_CHECK_ARG(slice_stride % 4 == 0 && dim <= 8192);
  MD_CHECK_ARG((((uintptr_t)x | (uintptr_t)partial_a | (uintptr_t)partial_b | (uintptr_t)bias_a | (uintptr_t)bias_b) & 15) == 0);
  const bf16_t *lw = nullptr, *lb = nullptr;
  if (y != nullptr) {
    MD_CHECK_ARG(ln && ln->w && ln->b && ldy % 8 == 0 && ldy >= dim && ((uintptr_t)y & 15) == 0);
    lw = (const bf16_t*)ln->w;
    lb = (const bf16_t*)ln->b;
  }
  hipStream_t s = (hipStream_t)stream;
#define MD_RRL(NCH)                                                                                          \
  hipLaunchKernelGGL(reduce_residual_ln_kernel<NCH>, dim3(rows), dim3(256), 0, s, (bf16_t*)x, ldx, partial_a, \
                     slices_a, (const bf16_t*)bias_a, partial_b, slices_b, (const bf16_t*)bias_b, ld_partial, \
                     slice_stride, (bf16_t*)y, ldy, lw, lb, dim, eps)
  if (dim <= 2048) MD_RRL(1);
  else if (dim <= 4096) MD_RRL(2);
  else MD_RRL(4);
#undef MD_RRL
  return md_launch_status();
}

extern "C" md_status md_patchify_u8(const void* crops_u8, const void* lut_bf16, void* out,
                                    int64_t ld_out, int32_t n_crops, int32_t crop, int32_t patch,
                                    void* stream) {
  MD_CHECK_ARG(crops_u8 && lut_bf16 && out && n_crops > 0 && patch > 0 && crop % patch == 0);
  MD_CHECK_ARG(ld_out >= 3 * patch * patch);
  const int g = crop / patch;
  hipLaunchKernelGGL(patchify_kernel<true>, dim3(n_crops * g * g), dim3(256), 0, (hipStream_t)stream,
                     crops_u8, (const bf16_t*)lut_bf16, (bf16_t*)out, ld_out, crop, patch, g);
  return md_launch_status();
}

extern "C" md_status md_patchify_bf16(const void* crops, void* out, int64_t ld_out, int32_t n_crops,
                                      int32_t crop, int32_t patch, void* stream) {
  MD_CHECK_ARG(crops && out && n_crops > 0 && patch > 0 && crop % patch == 0);
  MD_CHECK_ARG(ld_out >= 3 * patch * patch);
  const int g = crop / patch;
  hipLaunchKernelGGL(patchify_kernel<false>, dim3(n_crops * g * g), dim3(256), 0, (hipStream_t)stream,
                     crops, (const bf16_t*)nullptr, (bf16_t*)out, ld_out, crop, patch, g);
  return md_launch_status();
}

extern "C" md_status md_rope_kv_write(void* qkv, int64_t ld, const float* freqs, const int32_t* pos0,
                                      void* k_slab, void* v_slab, int64_t slab_batch_stride,
                                      int32_t ctx, int32_t batch, int32_t q_len, int32_t n_heads,
                                      int32_t n_kv_heads, int32_t head_dim, int32_t rot_dim,
                                      void* stream) {
  MD_CHECK_ARG(qkv && freqs && pos0 && k_slab && v_slab && batch > 0 && q_len > 0);
  MD_CHECK_ARG(head_dim % 8 == 0 && rot_dim % 8 == 0 && rot_dim <= head_dim && ld % 8 == 0);
  MD_CHECK_ARG((n_heads + n_kv_heads) * (rot_dim / 2) <= 1024);
  MD_CHECK_ARG(ld >= (int64_t)(n_heads + 2 * n_kv_heads) * head_dim);
  hipLaunchKernelGGL(rope_kv_kernel, dim3(batch * q_len), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)qkv, ld, freqs, pos0, (bf16_t*)k_slab, (bf16_t*)v_slab,
                     slab_batch_stride, ctx, q_len, n_heads, n_kv_heads, head_dim, rot_dim);
  return md_launch_status();
}

extern "C" md_status md_add_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo,
                                 int32_t rows, int32_t cols, void* stream) {
  MD_CHECK_ARG(a && b && out && rows > 0 && cols > 0 && cols % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0);
  const int64_t total = (int64_t)rows * (cols / 8);
  hipLaunchKernelGGL(rowwise_kernel<0>, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)out, ldo, rows, cols);
  return md_launch_status();
}

extern "C" md_status md_gelu_bf16(const void* a, int64_t lda, void* out, int64_t ldo, int32_t rows, int32_t cols, void* stream) {
  MD_CHECK_ARG(a && out && rows > 0 && cols > 0 && cols % 8 == 0 && lda % 8 == 0 && ldo % 8 == 0);
  const int64_t total = (int64_t)rows * (cols / 8);
  hipLaunchKernelGGL(rowwise_kernel<1>, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)a, lda, (const bf16_t*)nullptr, (int64_t)0, (bf16_t*)out, ldo, rows, cols);
  return md_launch_status();
}

extern "C" md_status md_embed_tokens(const int32_t* ids, const void* table, int64_t ld_table, void* out,
                                     int64_t ld_out, int32_t n, int32_t dim, void* stream) {
  MD_CHECK_ARG(ids && table && out && n > 0 && dim % 8 == 0 && ld_table % 8 == 0 && ld_out % 8 == 0);
  hipLaunchKernelGGL(embed_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, ids,
                     (const bf16_t*)table, ld_table, (bf16_t*)out, ld_out, dim);
  return md_launch_status();
}

// internal (api.hip): argmax that also advances pos
md_status md_argmax_advance(const void* logits, int64_t ld, int32_t batch, int32_t vocab,
                            int32_t suppress_id, int32_t* next, int32_t* pos, hipStream_t stream) {
  hipLaunchKernelGGL(argmax_kernel, dim3(batch), dim3(1024), 0, stream, (const bf16_t*)logits, ld, vocab,
                     suppress_id, next, pos);
  return md_launch_status();
}

extern "C" md_status md_argmax_bf16(const void* logits, int64_t ld, int32_t batch, int32_t vocab,
                                    int32_t suppress_id, int32_t* next, void* stream) {
  MD_CHECK_ARG(logits && next && batch > 0 && vocab > 0 && vocab % 8 == 0 && ld % 8 == 0);
  return md_argmax_advance(logits, ld, batch, vocab, suppress_id, next, nullptr, (hipStream_t)stream);
}

// internal: batched form used by md_vision_project
md_status md_stitch_pool_batched(const void* feats, void* out, int64_t ld_out, int64_t out_img_stride,
                                 int32_t n_images, int32_t dim, int32_t grid, int32_t margin,
                                 int32_t tiles_h, int32_t tiles_w, hipStream_t stream) {
  hipLaunchKernelGGL(stitch_pool_kernel, dim3(grid * grid, n_images), dim3(256), 0, stream,
                     (const bf16_t*)feats, (const bf16_t*)nullptr, (bf16_t*)out, ld_out, out_img_stride,
                     dim, grid, margin, tiles_h, tiles_w, 0, 0);
  return md_launch_status();
}

// internal: pool an already stitched [H][W][dim] grid (the _vis_proj(g, r) seam form)
md_status md_pool_grid_concat(const void* global_feats, const void* grid_feats, int32_t H, int32_t W,
                              void* out, int64_t ld_out, int32_t dim, int32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL(stitch_pool_kernel, dim3(grid * grid, 1), dim3(256), 0, stream,
                     (const bf16_t*)global_feats, (const bf16_t*)grid_feats, (bf16_t*)out, ld_out,
                     (int64_t)0, dim, grid, 0, 1, 1, H, W);
  return md_launch_status();
}

extern "C" md_status md_stitch_pool_concat(const void* feats, void* out, int64_t ld_out, int32_t dim,
                                           int32_t grid, int32_t margin, int32_t tiles_h,
                                           int32_t tiles_w, void* stream) {
  MD_CHECK_ARG(feats && out && dim % 8 == 0 && ld_out >= 2 * dim && ld_out % 8 == 0);
  MD_CHECK_ARG(grid > 2 * margin && tiles_h > 0 && tiles_w > 0);
  return md_stitch_pool_batched(feats, out, ld_out, 0, 1, dim, grid, margin, tiles_h, tiles_w,
                                (hipStream_t)stream);
}
