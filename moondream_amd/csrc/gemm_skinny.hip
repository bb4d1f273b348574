// Decode-regime linear layer: C[M <= 64, N] = epilogue(X . W^T + b).
//
// With at most 64 activation rows (one per sequence in flight) the layer is a
// stream over the weights: every W byte is used by <= 64 rows, so the bound is
// HBM bandwidth (2.6 GB of weights per decode step), not
// the matrix cores.  The big-tile kernel would put N/256 workgroups on 256 CUs;
// this one instead
//   * gives every workgroup 32 weight rows and ALL of K, split over its 8 waves
//     (in-workgroup split-K, combined in a FIXED order through LDS, so results
//     are deterministic and independent of how many rows M are live -- the
//     batched decode must equal the sequential one bit for bit);
//   * streams W straight into MFMA operand registers (16 B per lane, row =
//     lane & 31; four consecutive K-steps cover a full 128-byte line per row),
//     eight loads in flight per wave, no LDS round trip for the operand that is
//     read exactly once;
//   * reads the activations (<= 64 x K bf16, L2-resident) the same way.
// MFMA is used because it is the cheapest way to do 64 rows x 32 cols x 16 k of
// FMAs per instruction, not because the kernel is compute bound.
#include "md_common.hpp"

namespace {

struct SkinnyK {
  const bf16_t* X;
  const bf16_t* W;
  const bf16_t* bias;
  const bf16_t* R;
  bf16_t* C;
  int64_t ldx, ldw, ldc, ldr;
  int M, n_store, n_pad, K;
  int res_row_mod;
};

constexpr int SK_WAVES = 8;
constexpr int SK_UNROLL = 4;

template <int MT, int EPI>
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_skinny_kernel(const SkinnyK p) {
  __shared__ float part[SK_WAVES][MT][16][64];  // [wave][m tile][acc reg][lane]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int n0 = blockIdx.x * 32;

  // K range of this wave: whole 16-wide steps, contiguous
  const int steps = p.K / 16;
  const int per = (steps + SK_WAVES - 1) / SK_WAVES;
  const int s0 = min(wave * per, steps), s1 = min(s0 + per, steps);

  const bf16_t* wrow = p.W + (int64_t)min(n0 + l31, p.n_pad - 1) * p.ldw + 8 * hi;
  const bf16_t* xrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) xrow[mt] = p.X + (int64_t)min(mt * 32 + l31, p.M - 1) * p.ldx + 8 * hi;

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  int s = s0;
  for (; s + SK_UNROLL <= s1; s += SK_UNROLL) {
    bf16x8 wf[SK_UNROLL], xf[SK_UNROLL][MT];
#pragma unroll
    for (int u = 0; u < SK_UNROLL; ++u) wf[u] = __builtin_nontemporal_load((const bf16x8*)(wrow + (s + u) * 16));
#pragma unroll
    for (int u = 0; u < SK_UNROLL; ++u)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xf[u][mt] = *(const bf16x8*)(xrow[mt] + (s + u) * 16);
#pragma unroll
    for (int u = 0; u < SK_UNROLL; ++u)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u], xf[u][mt], acc[mt], 0, 0, 0);
  }
  for (; s < s1; ++s) {
    const bf16x8 wf = *(const bf16x8*)(wrow + s * 16);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, *(const bf16x8*)(xrow[mt] + s * 16), acc[mt], 0, 0, 0);
  }

#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][mt][r][lane] = acc[mt][r];
  __syncthreads();

  // combine the 8 K-partials in wave order, then the reference's rounding points
  for (int slot = tid; slot < MT * 16 * 64; slot += SK_WAVES * 64) {
    const int ln = slot & 63, r = (slot >> 6) & 15, mt = slot >> 10;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) v += part[w][mt][r][ln];
    const int m = mt * 32 + (ln & 31);
    const int n = n0 + 8 * (r >> 2) + 4 * (ln >> 5) + (r & 3);
    if (m < p.M && n < p.n_store) {
      if (p.bias != nullptr) v += bf2f(p.bias[n]);
      float y = bf2f(f2bf(v));
      if constexpr (EPI == MD_EPI_GELU) {
        y = gelu_tanh_f32(y);
      } else if constexpr (EPI == MD_EPI_RESIDUAL) {
        const int64_t rrow = p.res_row_mod ? (m % p.res_row_mod) : m;
        y = bf2f(p.R[rrow * p.ldr + n]) + y;
      }
      p.C[(int64_t)m * p.ldc + n] = f2bf(y);
    }
  }
}

template <int EPI>
md_status launch(const SkinnyK& k, hipStream_t s) {
  dim3 grid((k.n_store + 31) / 32), block(SK_WAVES * 64);
  if (k.M <= 32)
    hipLaunchKernelGGL((gemm_skinny_kernel<1, EPI>), grid, block, 0, s, k);
  else
    hipLaunchKernelGGL((gemm_skinny_kernel<2, EPI>), grid, block, 0, s, k);
  return md_launch_status();
}

}  // namespace

// internal: called by md_gemm_bf16 for m <= 64
md_status md_gemm_skinny(const md_gemm_args* a, hipStream_t stream) {
  SkinnyK k;
  k.X = (const bf16_t*)a->a;
  k.W = (const bf16_t*)a->lin.w;
  k.bias = (const bf16_t*)a->lin.b;
  k.R = (const bf16_t*)a->r;
  k.C = (bf16_t*)a->c;
  k.ldx = a->lda;
  k.ldw = a->lin.k_pad;
  k.ldc = a->ldc;
  k.ldr = a->ldr;
  k.M = a->m;
  k.n_pad = a->lin.n_pad;
  k.n_store = a->store_pad_cols ? a->lin.n_pad : a->lin.n;
  k.K = a->lin.k_pad;
  k.res_row_mod = a->res_row_mod;
  switch (a->epilogue) {
    case MD_EPI_BIAS: return launch<MD_EPI_BIAS>(k, stream);
    case MD_EPI_GELU: return launch<MD_EPI_GELU>(k, stream);
    case MD_EPI_RESIDUAL: return launch<MD_EPI_RESIDUAL>(k, stream);
    default: return MD_ERR_INVALID_ARG;
  }
}
