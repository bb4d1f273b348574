// Decode-regime linear over FP8 weights (gfx950, OCP e4m3fn): C[m, n] = epilogue(scale[n] * sum_k A[m, k] q[n, k] + b[n])
// for m <= 64 rows of bf16 activations.  This is the weight stream of a decode step (reference: the F.linear calls of
// text.py:30,53 and layers.py:130,139 and text.py:166 at q_len 1) at HALF the bytes: the layer is HBM-bound, the
// activations stay bf16, the products are exact (e4m3 -> bf16 is lossless) and accumulate in fp32 on
// v_mfma_f32_32x32x16_bf16; the only approximation is the weight quantisation itself (per-output-channel scale).
// It is an opt-in numerical mode (BASELINE configs[4] names fp8 weights), judged by tolerance against the bf16 path,
// never the default.
//
// Weight layout (our own quantised copy, so free to choose): MFMA-fragment order.  Block (nb, kb) = 32 output channels
// x 32 input features = 64 lanes x 16 bytes = 1 KiB contiguous; lane l holds channel 32 nb + (l & 31), bytes 0..7 =
// features 32 kb + 8 (l >> 5) + j and bytes 8..15 = features 32 kb + 16 + 8 (l >> 5) + j, i.e. exactly the first-operand
// fragments of two consecutive 16-wide MFMA steps.  A wave load is one coalesced 1 KiB line burst, there is no LDS
// staging and no swizzle for the weights; a channel block's K range is one contiguous stream.
//
// Workgroup: 4 waves, 64 channels x (32 | 64) rows x one K slice.  Wave (nbl, kh) owns channel block nbl and the kh-th
// half of every 128-wide K step (two 1 KiB weight loads per step).  Two activation paths, chosen by the row count:
//   RESIDENT (m <= 8, the few-sequence decode step): the whole activation slice [m][K slice] is put into LDS once; the
//     loop is then a pure weight stream -- no barrier, 8 steps (16 KiB) of weight loads in flight per wave.
//   RING (m <= 64): the bf16 activation chunk [rows][128] of every step goes through a 3-stage LDS ring (row stride
//     272 B: conflict-free ds_read_b128), one barrier per step; activation and weight loads of a step are issued
//     together, 4 steps ahead (loads return in order: a wait for an activation chunk must not imply a wait for younger
//     weight loads).
// The two K halves are combined through LDS in a fixed order and both paths add the same products in the same order,
// so the result is a function of the layer shape only (row-subset invariant like every other kernel here).
//
// FORMAT 1 (round 4): the reference's OWN quantised checkpoint format as a weight stream -- 4-bit weights in groups of 128
// input features with one (scale, zero_point) pair per group (layers.py:38-74, QuantizedLinear / dequantize_tensor).  The
// loader used to expand such checkpoints to bf16 once (weights.dequantize_int4) and stream 4x the bytes; here the nibbles
// are streamed and every weight is rebuilt in registers with the reference's own arithmetic and roundings,
//   w = bf16( bf16( q - zero ) * scale ),
// so the bf16 operand the MFMA sees is bit for bit the weight the bf16 path multiplies with: no numerical mode, the same
// layer at a quarter of the weight bytes (sums in this kernel's K order: agrees with the bf16 decode kernels to fp32
// accumulation order).  Layout: block (nb, step, kh) = 32 channels x 64 features (the kh-th half of a 128-wide K step = of
// a quantisation group) = 64 lanes x 16 bytes; lane l holds channel 32 nb + (l & 31), dword t = the 8 features
// 128 step + 64 kh + 16 t + 8 (l >> 5) + j as nibble j.  qparams: fp32 (scale, zero) at [step][n_pad].
#include "md_common.hpp"

#include <algorithm>
#include <type_traits>

namespace {

struct Fp8K {
  const bf16_t* A;
  int64_t lda;
  int Ka;  // readable (zero-padded) columns of A
  const uint8_t* W;
  const float* scale;
  const bf16_t* bias;
  bf16_t* C;
  int64_t ldc;
  float* partial;
  int64_t partial_ld, partial_slice_stride;
  int M, n_store, n_pad;
  int total_steps, steps_per_slice, slices;  // 128-wide K steps
  int gelu_from;
};
struct Fp8Pair {
  Fp8K g[2];
};

constexpr int STEP_K = 128;
constexpr int A_ROW = STEP_K * 2 + 16;  // bytes; rows rotate by 4 banks
constexpr int A_STAGE = 64 * A_ROW;
constexpr int NS = 3;
constexpr int EPI_PARTIAL = 3;

// 8 nibbles (one dword) -> one bf16x8 MFMA operand with the reference's dequantisation (layers.py:38-44: W_r.sub_(zero).mul_(scale)
// on a bf16 tensor: one rounding after the subtraction, one after the product)
__device__ __forceinline__ bf16x8 int4x8_to_bf16(uint32_t w, float s, float z) {
  const uint32_t lo = w & 0x0F0F0F0Fu, hi = (w >> 4) & 0x0F0F0F0Fu;  // nibbles 0, 2, 4, 6 / 1, 3, 5, 7 as bytes
  // (uint8 -> float of a byte lane: hipcc selects v_cvt_f32_ubyte0..3)
  const float q[8] = {(float)(lo & 0xffu),         (float)(hi & 0xffu),         (float)((lo >> 8) & 0xffu), (float)((hi >> 8) & 0xffu),
                      (float)((lo >> 16) & 0xffu), (float)((hi >> 16) & 0xffu), (float)(lo >> 24),          (float)(hi >> 24)};
  u32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t t = pack_bf16x2(q[2 * e] - z, q[2 * e + 1] - z);
    r[e] = pack_bf16x2(lo_bf(t) * s, hi_bf(t) * s);
  }
  return __builtin_bit_cast(bf16x8, r);
}

// 8 fp8 (two dwords) -> one bf16x8 MFMA operand: v_cvt_scalef32_pk_bf16_fp8 converts a pair per instruction
// (scale 1.0; e4m3 -> bf16 is exact)
typedef __bf16 md_bf16pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8 fp8x8_to_bf16(uint32_t lo, uint32_t hi) {
  const md_bf16pair a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)lo, 1.0f, false);
  const md_bf16pair b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)lo, 1.0f, true);
  const md_bf16pair c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)hi, 1.0f, false);
  const md_bf16pair d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)hi, 1.0f, true);
  const u32x4 r = {__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, c),
                   __builtin_bit_cast(uint32_t, d)};
  return __builtin_bit_cast(bf16x8, r);
}

template <int EPI, int MF, bool RESIDENT, int FMT>
__device__ __forceinline__ void fp8w_body(const Fp8K& p, char* smem) {
  static_assert(!RESIDENT || MF == 1, "resident activations: at most 8 rows");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nbl = wave & 1, kh = wave >> 1;
  const int n0 = blockIdx.x * 64;
  const int slice = blockIdx.y;
  if (n0 >= p.n_pad || slice >= p.slices) return;  // pair launch: the grid is sized for the larger problem
  const int step0 = slice * p.steps_per_slice;
  const int nsteps = min(p.steps_per_slice, p.total_steps - step0);
  const int KB = p.total_steps * 4;
  // FMT 0: two 1 KiB blocks per step and K half; FMT 1: one, plus the group's (scale, zero) of this lane's channel
  const u32x4* wp = FMT == 0 ? (const u32x4*)p.W + ((int64_t)(n0 / 32 + nbl) * KB + step0 * 4 + 2 * kh) * 64 + lane
                             : (const u32x4*)p.W + (((int64_t)(n0 / 32 + nbl) * p.total_steps + step0) * 2 + kh) * 64 + lane;
  const u32x2* qpp = (const u32x2*)p.scale + (int64_t)step0 * p.n_pad + n0 + 32 * nbl + l31;

  f32x16 acc[MF];
#pragma unroll
  for (int f = 0; f < MF; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

  // one 128-wide step of this wave: 2 weight loads (already in registers) x 2 MFMA k-steps x MF row blocks
  auto compute = [&](const u32x4 (&wq2)[2], const char* st, int row_stride) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8 wf;
        if constexpr (FMT == 0) wf = fp8x8_to_bf16(wq2[j][2 * t], wq2[j][2 * t + 1]);
        else wf = int4x8_to_bf16(wq2[0][2 * j + t], __uint_as_float(wq2[1].x), __uint_as_float(wq2[1].y));
#pragma unroll
        for (int f = 0; f < MF; ++f) {
          const bf16x8 af = *(const bf16x8*)(st + f * 32 * row_stride + (32 * j + 16 * t) * 2);
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, af, acc[f], 0, 0, 0);
        }
      }
    }
  };

  // Both loops are branch-free inside a block of PF steps: every load is issued unconditionally (step indices past the
  // slice are clamped to its last step and their weights replaced by zeros), so hipcc can count the in-order returns
  // (s_waitcnt vmcnt(N)) instead of draining the queue at every block.
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto load_w = [&](int i, u32x4 (&dst)[2]) {
    const int ic = min(i, nsteps - 1);
    if constexpr (FMT == 0) {
      dst[0] = wp[(4 * ic) * 64];
      dst[1] = wp[(4 * ic + 1) * 64];
    } else {
      dst[0] = wp[(2 * ic) * 64];
      const u32x2 qp = qpp[(int64_t)ic * p.n_pad];  // a dead step's slot is replaced by zeros below: scale 0 makes its weights 0
      u32x4 t4 = {0u, 0u, 0u, 0u};
      t4.x = qp.x;  // scale
      t4.y = qp.y;  // zero point
      dst[1] = t4;
    }
  };
  if constexpr (RESIDENT) {
    // ---- few rows: activation slice resident in LDS, then a barrier-free weight stream -------------------------
    constexpr int PFW = 8;
    const int rs = nsteps * (STEP_K * 2) + 16;  // row stride: rotates 4 banks per row
    const int ppr = nsteps * 16;               // 16-byte pieces per row
    // 8 independent loads per thread before the first store: one L2 round trip for up to 8 rows of K = 2048
    const int total = p.M * ppr;
    for (int q0 = tid; q0 < total; q0 += 256 * 8) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = min(q0 + 256 * u, total - 1);
        const int r = q / ppr, pc = q - r * ppr;
        const int k = step0 * STEP_K + pc * 8;
        const u32x4 x = *(const u32x4*)(p.A + (int64_t)r * p.lda + (k < p.Ka ? k : 0));
        v[u] = (k < p.Ka) ? x : zero4;  // columns past the activation's zero padding
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0 + 256 * u;
        if (q < total) {
          const int r = q / ppr, pc = q - r * ppr;
          *(u32x4*)(smem + r * rs + pc * 16) = v[u];
        }
      }
    }
    u32x4 wreg[PFW][2];
#pragma unroll
    for (int u = 0; u < PFW; ++u) load_w(u, wreg[u]);
    __syncthreads();
    const char* base = smem + min(l31, p.M - 1) * rs + (64 * kh + 8 * hi) * 2;  // rows past m replay the last row
    for (int i0 = 0; i0 < nsteps; i0 += PFW) {
#pragma unroll
      for (int u = 0; u < PFW; ++u) {
        const int i = i0 + u;
        const bool live = i < nsteps;
        const u32x4 wq[2] = {live ? wreg[u][0] : zero4, live ? wreg[u][1] : zero4};
        compute(wq, base + min(i, nsteps - 1) * (STEP_K * 2), 0);
        load_w(i + PFW, wreg[u]);
      }
    }
    __syncthreads();  // the reduction below reuses the LDS
  } else {
    // ---- up to 64 rows: activation chunks through the LDS ring ---------------------------------------------------
    constexpr int PFR = 4;
    constexpr int APT = 2 * MF;  // 16-byte activation pieces per thread and step
    u32x4 areg[PFR][APT], wreg[PFR][2];
    auto load_a = [&](int i, u32x4 (&dst)[APT]) {
      const int ic = min(i, nsteps - 1);
#pragma unroll
      for (int u = 0; u < APT; ++u) {
        const int q = tid + 256 * u, r = q >> 4, pc = q & 15;
        const int k = (step0 + ic) * STEP_K + pc * 8;
        // rows past m replay the last row; columns past the activation's padding replay column 0 and are zeroed
        const u32x4 v = *(const u32x4*)(p.A + (int64_t)min(r, p.M - 1) * p.lda + (k < p.Ka ? k : 0));
        dst[u] = (k < p.Ka) ? v : zero4;
      }
    };
    auto store_a = [&](int stage, const u32x4 (&src)[APT]) {
#pragma unroll
      for (int u = 0; u < APT; ++u) {
        const int q = tid + 256 * u, r = q >> 4, pc = q & 15;
        *(u32x4*)(smem + stage * A_STAGE + r * A_ROW + pc * 16) = src[u];
      }
    };
#pragma unroll
    for (int u = 0; u < PFR; ++u) {
      load_a(u, areg[u]);
      load_w(u, wreg[u]);
    }
    store_a(0, areg[0]);
    __syncthreads();
    for (int i0 = 0; i0 < nsteps; i0 += PFR) {
#pragma unroll
      for (int u = 0; u < PFR; ++u) {
        const int i = i0 + u;
        const bool live = i < nsteps;
        store_a((i + 1) % NS, areg[(u + 1) % PFR]);  // chunk i + 1 (past the slice: a replay nobody reads)
        const u32x4 wq[2] = {live ? wreg[u][0] : zero4, live ? wreg[u][1] : zero4};
        compute(wq, smem + (i % NS) * A_STAGE + l31 * A_ROW + (64 * kh + 8 * hi) * 2, A_ROW);
        load_a(i + PFR, areg[u]);  // the chunk before the weights of the same step: both are old when needed
        load_w(i + PFR, wreg[u]);
        __syncthreads();
      }
    }
  }

  // ---- combine the two K halves (fixed order: first half + second half), then the epilogue --------------------
  float* red = (float*)smem;  // the activation ring is dead after the loop's last barrier
  if (kh == 1) {
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((nbl * MF + f) * 16 + r) * 64 + lane] = acc[f][r];
  }
  __syncthreads();
  if (kh == 1) return;
#pragma unroll
  for (int f = 0; f < MF; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] += red[((nbl * MF + f) * 16 + r) * 64 + lane];

  // acc[f][r]: row m = 32 f + l31, channel n = n0 + 32 nbl + 8 (r >> 2) + 4 hi + (r & 3)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = n0 + 32 * nbl + 8 * g + 4 * hi;
    if (n >= p.n_store) continue;
    f32x4 sc = {1.0f, 1.0f, 1.0f, 1.0f};  // FMT 1: the group scales are inside the operands
    if constexpr (FMT == 0) sc = *(const f32x4*)(p.scale + n);
    if constexpr (EPI == EPI_PARTIAL) {
      float* dst = p.partial + (int64_t)slice * p.partial_slice_stride;
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        const int m = 32 * f + l31;
        if (m < p.M)
          *(f32x4*)(dst + (int64_t)m * p.partial_ld + n) = f32x4{acc[f][4 * g] * sc[0], acc[f][4 * g + 1] * sc[1],
                                                                  acc[f][4 * g + 2] * sc[2], acc[f][4 * g + 3] * sc[3]};
      }
    } else {
      u32x2 bw = {0u, 0u};
      if (p.bias != nullptr) bw = *(const u32x2*)(p.bias + n);
      const float b0 = lo_bf(bw[0]), b1 = hi_bf(bw[0]), b2 = lo_bf(bw[1]), b3 = hi_bf(bw[1]);
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        const int m = 32 * f + l31;
        // one bf16 rounding of (acc + bias), a second one after the GELU: the rounding points of the bf16 kernels
        uint32_t w0 = pack_bf16x2(acc[f][4 * g] * sc[0] + b0, acc[f][4 * g + 1] * sc[1] + b1);
        uint32_t w1 = pack_bf16x2(acc[f][4 * g + 2] * sc[2] + b2, acc[f][4 * g + 3] * sc[3] + b3);
        if constexpr (EPI == MD_EPI_GELU) {
          if (n >= p.gelu_from) {
            const md_f32x2 ga = gelu_tanh_f32x2(md_f32x2{lo_bf(w0), hi_bf(w0)});
            const md_f32x2 gb = gelu_tanh_f32x2(md_f32x2{lo_bf(w1), hi_bf(w1)});
            w0 = pack_bf16x2(ga[0], ga[1]);
            w1 = pack_bf16x2(gb[0], gb[1]);
          }
        }
        if (m < p.M) *(u32x2*)(p.C + (int64_t)m * p.ldc + n) = u32x2{w0, w1};
      }
    }
  }
}

template <int EPI, int MF, bool RESIDENT, int FMT>
__global__ __launch_bounds__(256) void gemm_fp8w_kernel(const Fp8K p) {
  __shared__ __attribute__((aligned(16))) char smem[NS * A_STAGE];
  fp8w_body<EPI, MF, RESIDENT, FMT>(p, smem);
}
template <int MF, bool RESIDENT, int FMT>
__global__ __launch_bounds__(256) void gemm_fp8w_pair_kernel(const Fp8Pair pair) {
  __shared__ __attribute__((aligned(16))) char smem[NS * A_STAGE];
  fp8w_body<EPI_PARTIAL, MF, RESIDENT, FMT>(pair.g[blockIdx.z], smem);
}

// the activation slice of every workgroup fits the LDS array of the kernels: m rows x (steps x 256 + 16) bytes
bool resident_fits(int m, int steps_per_slice) {
  return m <= 8 && (int64_t)m * (steps_per_slice * STEP_K * 2 + 16) <= NS * A_STAGE;
}

md_status fill(Fp8K& k, const void* a, int64_t lda, const md_linear_fp8* lin, int m) {
  MD_CHECK_ARG(a && lin && lin->w && lin->scale);
  MD_CHECK_ARG(m > 0 && m <= 64 && lin->n > 0 && lin->k > 0);
  MD_CHECK_ARG(lin->n_pad % 64 == 0 && lin->n_pad >= lin->n && lin->k_pad % STEP_K == 0 && lin->k_pad >= lin->k);
  MD_CHECK_ARG(lda % 8 == 0 && lda >= lin->k && ((uintptr_t)a & 15) == 0 && ((uintptr_t)lin->w & 15) == 0);
  MD_CHECK_ARG(((uintptr_t)lin->scale & 15) == 0);
  MD_CHECK_ARG(lin->format == MD_WSTREAM_E4M3 || lin->format == MD_WSTREAM_INT4_G128);
  MD_CHECK_ARG(lin->format == MD_WSTREAM_E4M3 || lin->k_pad == lin->k);  // a quantisation group is a whole 128-wide K step
  k.A = (const bf16_t*)a;
  k.lda = lda;
  // Columns [k, round_up(k, 64)) of A are the producer's zero padding (the bf16 packing pads to 64); nothing past that is
  // promised to be zero or even to belong to the row (A may be a column slice of a wider fused activation, lda > k_pad),
  // while k_pad here is rounded to 128: the loader supplies zeros beyond Ka itself.
  k.Ka = (int)std::min<int64_t>(std::min<int64_t>(lda, lin->k_pad), ((int64_t)lin->k + 63) / 64 * 64) / 8 * 8;
  k.W = (const uint8_t*)lin->w;
  k.scale = lin->scale;
  k.bias = (const bf16_t*)lin->b;
  k.C = nullptr;
  k.ldc = 0;
  k.partial = nullptr;
  k.partial_ld = k.partial_slice_stride = 0;
  k.M = m;
  k.n_pad = lin->n_pad;
  k.n_store = lin->n_pad;
  k.total_steps = lin->k_pad / STEP_K;
  k.steps_per_slice = k.total_steps;
  k.slices = 1;
  k.gelu_from = 0;
  return MD_OK;
}

}  // namespace

// K slices of the launch-boundary split (same rule as md_gemm_partial_slices, in 128-wide steps)
extern "C" int32_t md_gemm_fp8w_partial_slices(const md_linear_fp8* lin) {
  if (!lin || lin->n <= 0 || lin->k_pad <= 0) return 0;
  const int tiles = (lin->n + 63) / 64, steps = lin->k_pad / STEP_K;
  int s = 256 / std::max(1, tiles);
  s = std::max(1, std::min(s, std::min(8, steps / 8)));  // at least 8 steps (one prefetch window) per slice
  const int per = (steps + s - 1) / s;
  return (steps + per - 1) / per;
}

extern "C" md_status md_gemm_fp8w(const void* a, int64_t lda, const md_linear_fp8* lin, void* c, int64_t ldc, int32_t m,
                                  int32_t epilogue, int32_t store_pad_cols, int32_t gelu_from_col, void* stream) {
  Fp8K k;
  MD_TRY(fill(k, a, lda, lin, m));
  MD_CHECK_ARG(c && ldc % 4 == 0 && ((uintptr_t)c & 7) == 0);
  MD_CHECK_ARG(epilogue == MD_EPI_BIAS || epilogue == MD_EPI_GELU);
  MD_CHECK_ARG(gelu_from_col >= 0 && gelu_from_col % 8 == 0);
  MD_CHECK_ARG(store_pad_cols || lin->n % 8 == 0);
  k.C = (bf16_t*)c;
  k.ldc = ldc;
  k.n_store = store_pad_cols ? lin->n_pad : lin->n;
  MD_CHECK_ARG(ldc >= k.n_store);
  k.gelu_from = gelu_from_col;
  const dim3 grid(lin->n_pad / 64, 1), block(256);
  hipStream_t s = (hipStream_t)stream;
  const bool res = resident_fits(m, k.steps_per_slice);
  auto launch = [&](auto fmt_c) {
    constexpr int FMT = decltype(fmt_c)::value;
    if (epilogue == MD_EPI_GELU) {
      if (res) hipLaunchKernelGGL((gemm_fp8w_kernel<MD_EPI_GELU, 1, true, FMT>), grid, block, 0, s, k);
      else if (m <= 32) hipLaunchKernelGGL((gemm_fp8w_kernel<MD_EPI_GELU, 1, false, FMT>), grid, block, 0, s, k);
      else hipLaunchKernelGGL((gemm_fp8w_kernel<MD_EPI_GELU, 2, false, FMT>), grid, block, 0, s, k);
    } else {
      if (res) hipLaunchKernelGGL((gemm_fp8w_kernel<MD_EPI_BIAS, 1, true, FMT>), grid, block, 0, s, k);
      else if (m <= 32) hipLaunchKernelGGL((gemm_fp8w_kernel<MD_EPI_BIAS, 1, false, FMT>), grid, block, 0, s, k);
      else hipLaunchKernelGGL((gemm_fp8w_kernel<MD_EPI_BIAS, 2, false, FMT>), grid, block, 0, s, k);
    }
  };
  if (lin->format == MD_WSTREAM_INT4_G128) launch(std::integral_constant<int, 1>{});
  else launch(std::integral_constant<int, 0>{});
  return md_launch_status();
}

extern "C" md_status md_gemm_fp8w_partial_f32_pair(const void* a0, int64_t lda0, const md_linear_fp8* lin0, float* partial0,
                                                   const void* a1, int64_t lda1, const md_linear_fp8* lin1, float* partial1,
                                                   int32_t m, int64_t ld_partial, int64_t slice_stride, void* stream) {
  Fp8Pair pair;
  const void* as[2] = {a0, a1};
  const int64_t ldas[2] = {lda0, lda1};
  const md_linear_fp8* lins[2] = {lin0, lin1};
  float* parts[2] = {partial0, partial1};
  int gx = 0, gy = 0;
  for (int i = 0; i < 2; ++i) {
    Fp8K& k = pair.g[i];
    MD_TRY(fill(k, as[i], ldas[i], lins[i], m));
    k.n_store = std::min(lins[i]->n_pad, (lins[i]->n + 3) / 4 * 4);
    MD_CHECK_ARG(parts[i] && ld_partial % 4 == 0 && ld_partial >= k.n_store && ((uintptr_t)parts[i] & 15) == 0);
    MD_CHECK_ARG(slice_stride >= (int64_t)m * ld_partial);
    k.partial = parts[i];
    k.partial_ld = ld_partial;
    k.partial_slice_stride = slice_stride;
    k.slices = md_gemm_fp8w_partial_slices(lins[i]);
    k.steps_per_slice = (k.total_steps + k.slices - 1) / k.slices;
    gx = std::max(gx, lins[i]->n_pad / 64);
    gy = std::max(gy, k.slices);
  }
  MD_CHECK_ARG(lin0->format == lin1->format);  // one launch, one weight format
  const dim3 grid(gx, gy, 2), block(256);
  hipStream_t s = (hipStream_t)stream;
  auto launch = [&](auto fmt_c) {
    constexpr int FMT = decltype(fmt_c)::value;
    if (resident_fits(m, std::max(pair.g[0].steps_per_slice, pair.g[1].steps_per_slice)))
      hipLaunchKernelGGL((gemm_fp8w_pair_kernel<1, true, FMT>), grid, block, 0, s, pair);
    else if (m <= 32) hipLaunchKernelGGL((gemm_fp8w_pair_kernel<1, false, FMT>), grid, block, 0, s, pair);
    else hipLaunchKernelGGL((gemm_fp8w_pair_kernel<2, false, FMT>), grid, block, 0, s, pair);
  };
  if (lin0->format == MD_WSTREAM_INT4_G128) launch(std::integral_constant<int, 1>{});
  else launch(std::integral_constant<int, 0>{});
  return md_launch_status();
}
