// FP8 (OCP e4m3fn) activation producers for the opt-in fp8 mode (md_gemm_f8's A operand) and the calibration
// reduction that sizes their per-tensor scales.  All HBM-bound, one pass over the data:
//   md_quantize_f8     bf16 rows -> fp8 rows (attention outputs, the projector's concatenated input)
//   md_layernorm_f8    layer norm (layers.py:118-119) with its bf16 rounding point kept, then quantised
//   md_amax_bf16       running max |x| of a tensor (atomic max on the fp32 bit pattern of a non-negative value)
#include "md_common.hpp"

namespace {

// one 8-element chunk (16 B in, 8 B out) per thread, grid-stride over rows x chunks
__global__ __launch_bounds__(256) void quantize_f8_kernel(const bf16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ y, int64_t ldy,
                                                          int rows, int nch, int nch_pad, float inv_scale) {
  const int64_t total = (int64_t)rows * nch_pad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / nch_pad), ch = (int)(i % nch_pad);
    u32x2 o = {0u, 0u};
    if (ch < nch) o = quant8(*(const u32x4*)(x + (int64_t)r * ldx + ch * 8), inv_scale);
    *(u32x2*)(y + (int64_t)r * ldy + ch * 8) = o;
  }
}

// one wave per row, the row in registers between the two reductions (as layernorm_kernel in elementwise.hip)
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_f8_kernel(const bf16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ y, int64_t ldy,
                                                           const bf16_t* __restrict__ w, const bf16_t* __restrict__ bia, int rows, int dim,
                                                           int dim_pad, float eps, float inv_scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = dim >> 3, nchunk_pad = dim_pad >> 3;
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
  uint8_t* yr = y + (int64_t)row * ldy;
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
        out[e] = pack_bf16x2(a, c);  // the reference's rounding point: the norm's output is a bf16 tensor
      }
      *(u32x2*)(yr + ch * 8) = quant8(out, inv_scale);
    } else if (ch < nchunk_pad) {
      *(u32x2*)(yr + ch * 8) = u32x2{0u, 0u};
    }
  }
}

__global__ __launch_bounds__(256) void amax_kernel(const bf16_t* __restrict__ x, int64_t ldx, int rows, int nch, float* __restrict__ amax) {
  const int64_t total = (int64_t)rows * nch;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / nch), ch = (int)(i % nch);
    const u32x4 q = *(const u32x4*)(x + (int64_t)r * ldx + ch * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = fabsf(lo_bf(q[e])), b = fabsf(hi_bf(q[e]));
      if (a < 3.0e38f) m = fmaxf(m, a);  // ignores inf / NaN
      if (b < 3.0e38f) m = fmaxf(m, b);
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned int*)amax, __float_as_uint(m));  // non-negative floats order like their bits
}

}  // namespace

extern "C" md_status md_quantize_f8(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t cols, int32_t cols_pad,
                                    float inv_scale, void* stream) {
  MD_CHECK_ARG(x && y && rows > 0 && cols > 0 && cols % 8 == 0 && cols_pad % 8 == 0 && cols_pad >= cols);
  MD_CHECK_ARG(ldx % 8 == 0 && ldx >= cols && ldy % 8 == 0 && ldy >= cols_pad && inv_scale > 0.f);
  MD_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0);
  const int64_t total = (int64_t)rows * (cols_pad / 8);
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(quantize_f8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (uint8_t*)y, ldy, rows,
                     cols / 8, cols_pad / 8, inv_scale);
  return md_launch_status();
}

extern "C" md_status md_layernorm_f8(const void* x, int64_t ldx, void* y, int64_t ldy, const md_layernorm* p, int32_t rows,
                                     int32_t dim, int32_t dim_pad, float eps, float inv_scale, void* stream) {
  MD_CHECK_ARG(x && y && p && p->w && p->b && rows > 0 && inv_scale > 0.f);
  MD_CHECK_ARG(dim % 8 == 0 && dim > 0 && dim <= 4096 && dim_pad % 8 == 0 && dim_pad >= dim && dim_pad <= 4096 && ldx % 8 == 0 &&
               ldy % 8 == 0 && ldy >= dim_pad);
  MD_CHECK_ARG((((uintptr_t)x | (uintptr_t)p->w | (uintptr_t)p->b) & 15) == 0 && ((uintptr_t)y & 7) == 0);
  dim3 grid((rows + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  const bf16_t* w = (const bf16_t*)p->w;
  const bf16_t* b = (const bf16_t*)p->b;
#define MD_LNF8(NCH) \
  hipLaunchKernelGGL(layernorm_f8_kernel<NCH>, grid, block, 0, s, (const bf16_t*)x, ldx, (uint8_t*)y, ldy, w, b, rows, dim, dim_pad, eps, inv_scale)
  if (dim_pad <= 512) MD_LNF8(1);
  else if (dim_pad <= 1536) MD_LNF8(3);
  else if (dim_pad <= 2560) MD_LNF8(5);
  else MD_LNF8(8);
#undef MD_LNF8
  return md_launch_status();
}

extern "C" md_status md_amax_bf16(const void* x, int64_t ldx, int32_t rows, int32_t cols, float* amax, void* stream) {
  MD_CHECK_ARG(x && amax && rows > 0 && cols > 0 && cols % 8 == 0 && ldx % 8 == 0 && ldx >= cols && ((uintptr_t)x & 15) == 0);
  const int64_t total = (int64_t)rows * (cols / 8);
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(amax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, rows, cols / 8, amax);
  return md_launch_status();
}
