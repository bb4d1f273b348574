// Device-resident pieces of the generation loops above the seam:
//   * the region head's "pick a bin and re-encode it" step of detect / point
//     (reference: moondream.py:672-713 around region.py:12-71), so a points loop over B
//     images needs no host round trip per coordinate;
//   * temperature / top-p sampling (reference: moondream.py:270-278, 313-318, 521-528).
// HBM-bound row kernels: one workgroup per sequence, 16-byte loads, wave shuffles.
#include "md_common.hpp"

namespace {

// ---------------------------------------------------------------------------
// [cos(f) | sin(f)],  f = bf16( sum_g bf16(2 pi v_g) * W[g][j] )          (region.py:12-29)
// v_g bf16; the K = 1 / K = 2 contraction is an fp32 sum of exact products, rounded once.
__device__ __forceinline__ void fourier_row(const float* v, int n_groups, const bf16_t* __restrict__ w, int half,
                                            bf16_t* __restrict__ out) {
  float t[2];
  for (int g = 0; g < n_groups; ++g) t[g] = bf2f(f2bf(v[g] * 6.283185307179586f));  // 2 * math.pi * x, in bf16
  for (int j = threadIdx.x; j < half; j += blockDim.x) {
    float f = t[0] * bf2f(w[j]);
    if (n_groups == 2) f += t[1] * bf2f(w[half + j]);
    f = bf2f(f2bf(f));
    out[j] = f2bf(cosf(f));
    out[half + j] = f2bf(sinf(f));
  }
}

__global__ __launch_bounds__(256) void fourier_kernel(const bf16_t* __restrict__ x, int64_t ldx, int in_dim,
                                                      const bf16_t* __restrict__ w, int half,
                                                      bf16_t* __restrict__ out, int64_t ldo) {
  const int r = blockIdx.x;
  float v[2] = {bf2f(x[r * ldx]), in_dim == 2 ? bf2f(x[r * ldx + 1]) : 0.f};
  fourier_row(v, in_dim, w, half, out + (int64_t)r * ldo);
}

// one workgroup per sequence: argmax of each 1024-bin group (ties -> lowest index, like
// torch.argmax on the reference's CPU path), value = table[bin], then the Fourier features
__global__ __launch_bounds__(256) void region_pick_kernel(const bf16_t* __restrict__ logits, int64_t ld, int n_groups,
                                                          int n_bins, const bf16_t* __restrict__ table,
                                                          const bf16_t* __restrict__ w, int half,
                                                          int32_t* __restrict__ bins, int64_t ld_bins,
                                                          bf16_t* __restrict__ feats, int64_t ldf) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ float val[2];
  const int b = blockIdx.x;
  for (int g = 0; g < n_groups; ++g) {
    const bf16_t* lr = logits + (int64_t)b * ld + (int64_t)g * n_bins;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n_bins; i += 256) {
      const float x = bf2f(lr[i]);
      if (x > best || (x == best && i < bi)) {
        best = x;
        bi = i;
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
      for (int wv = 1; wv < 4; ++wv)
        if (sv[wv] > best || (sv[wv] == best && si[wv] < bi)) {
          best = sv[wv];
          bi = si[wv];
        }
      if (bi >= n_bins) bi = 0;  // all-NaN row
      bins[(int64_t)b * ld_bins + g] = bi;
      val[g] = bf2f(table[bi]);
    }
    __syncthreads();
  }
  float v[2] = {val[0], n_groups == 2 ? val[1] : 0.f};
  fourier_row(v, n_groups, w, half, feats + (int64_t)b * ldf);
}

// ---------------------------------------------------------------------------
// temperature + top-p sampling of one token per sequence.
//   z = bf16(logit / T);  p = bf16(softmax(z))                          moondream.py:526 (bf16 tensors, fp32 inside)
//   in descending-p order keep the tokens whose preceding mass is <= top_p (evaluated with the
//   reference's bf16 roundings, see below); renormalise                    moondream.py:270-278
//   draw from the kept distribution with the caller's uniform u          moondream.py:528 (torch.multinomial)
// The descending order is realised without a sort: bf16 probabilities take < 2^15 distinct
// positive values, so a histogram over the bit pattern (LDS, 128 KiB) gives every value's count
// and the mass above it; within the boundary value the lowest token ids are kept.
constexpr int SAMPLE_THREADS = 1024;
constexpr int HIST_BINS = 32768;

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < SAMPLE_THREADS / 64; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int i = 0; i < SAMPLE_THREADS / 64; ++i) t = fmaxf(t, red[i]);
  return t;
}

__global__ __launch_bounds__(SAMPLE_THREADS) void sample_top_p_kernel(const bf16_t* __restrict__ logits, int64_t ld, int vocab,
                                                                      int suppress, float temperature, float top_p,
                                                                      const float* __restrict__ uniforms,
                                                                      int32_t* __restrict__ next,
                                                                      bf16_t* __restrict__ probs_out, int64_t ldp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* hist = (unsigned*)smem;                    // [HIST_BINS] counts per bf16 bit pattern
  float* red = (float*)(smem + HIST_BINS * 4);         // [16]
  float* scan = red + 16;                              // [SAMPLE_THREADS] block scans
  int* shared_i = (int*)(scan + SAMPLE_THREADS);       // [8] boundary pattern, survivors in it, drawn token, ...
  const int b = blockIdx.x, tid = threadIdx.x;
  const bf16_t* lr = logits + (int64_t)b * ld;
  const float inv_t = temperature;  // division, not a reciprocal multiply: the reference divides

  // ---- z = bf16(logit / T), softmax statistics
  float mx = -INFINITY;
  for (int i = tid; i < vocab; i += SAMPLE_THREADS) {
    float z = (i == suppress) ? -INFINITY : bf2f(f2bf(bf2f(lr[i]) / inv_t));
    mx = fmaxf(mx, z);
  }
  mx = block_max(mx, red);
  float s = 0.f;
  for (int i = tid; i < vocab; i += SAMPLE_THREADS) {
    float z = (i == suppress) ? -INFINITY : bf2f(f2bf(bf2f(lr[i]) / inv_t));
    s += expf(z - mx);
  }
  s = block_sum(s, red);
  auto prob_bits = [&](int i) -> unsigned {
    float z = (i == suppress) ? -INFINITY : bf2f(f2bf(bf2f(lr[i]) / inv_t));
    return (unsigned)f2bf(expf(z - mx) / s);  // non-negative: the sign bit is clear
  };

  // ---- histogram over the 15-bit pattern
  for (int i = tid; i < HIST_BINS; i += SAMPLE_THREADS) hist[i] = 0u;
  __syncthreads();
  for (int i = tid; i < vocab; i += SAMPLE_THREADS) atomicAdd(&hist[prob_bits(i) & 0x7fffu], 1u);
  __syncthreads();

  // ---- walk the values from the largest down: thread t owns HIST_BINS / 1024 = 32 consecutive patterns
  constexpr int PER = HIST_BINS / SAMPLE_THREADS;
  const int top_pat = HIST_BINS - 1 - tid * PER;  // this thread's largest pattern
  float mass = 0.f;
  for (int k = 0; k < PER; ++k) {
    const int pat = top_pat - k;
    if (hist[pat]) mass += (float)hist[pat] * bf2f((bf16_t)pat);  // empty patterns include inf / NaN encodings
  }
  // exclusive prefix over threads (descending value order == ascending tid)
  scan[tid] = mass;
  __syncthreads();
  for (int off = 1; off < SAMPLE_THREADS; off <<= 1) {
    const float add = (tid >= off) ? scan[tid - off] : 0.f;
    __syncthreads();
    scan[tid] += add;
    __syncthreads();
  }
  float before = scan[tid] - mass;  // mass of all strictly larger values owned by earlier threads
  if (tid == 0) {
    shared_i[0] = HIST_BINS;  // boundary pattern: the SMALLEST value that still has a surviving token
    shared_i[1] = 0;          // how many tokens of that value survive
    shared_i[3] = -1;         // the drawn token
    shared_i[4] = -1;         // last survivor (fallback when rounding puts the target past the end)
  }
  __syncthreads();
  // `before` only grows as the values descend, so the values with survivors form a prefix of the
  // descending order and every value above the boundary survives whole
  int my_b = HIST_BINS, my_n = 0;
  const float tp16 = bf2f(f2bf(top_p));
  for (int k = 0; k < PER; ++k) {
    const int pat = top_pat - k;
    const unsigned c = hist[pat];
    const float v = bf2f((bf16_t)pat);
    if (c > 0u && v > 0.f) {
      // token j of this value (j = 0..c-1): the reference's cumsum is an fp32 running sum rounded to
      // bf16 per element, the subtraction and the comparison are bf16 too (the python scalar top_p is
      // cast to bf16):  keep iff  bf16(bf16(before + (j+1) v) - v) <= bf16(top_p).  Monotone in j.
      auto kept = [&](int j) { return bf2f(f2bf(bf2f(f2bf(before + (float)(j + 1) * v)) - v)) <= tp16; };
      if (kept(0)) {
        int lo = 0, hi = (int)c - 1;  // largest j that is kept
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (kept(mid)) lo = mid; else hi = mid - 1;
        }
        my_b = pat;
        my_n = lo + 1;
      }
    }
    if (c) before += (float)c * v;
  }
  if (my_b < HIST_BINS) atomicMin(&shared_i[0], my_b);
  __syncthreads();
  const int bpat = shared_i[0];
  if (my_b == bpat && bpat < HIST_BINS) shared_i[1] = my_n;  // every pattern has exactly one owner
  __syncthreads();
  const int n_boundary = shared_i[1];

  // ---- survivors: pattern > bpat, or pattern == bpat and among its n_boundary lowest ids.
  // Contiguous chunks per thread keep id order; two block scans (equal-count, kept mass).
  const int chunk = (vocab + SAMPLE_THREADS - 1) / SAMPLE_THREADS;
  const int i0 = tid * chunk, i1 = min(vocab, i0 + chunk);
  int eq = 0;
  for (int i = i0; i < i1; ++i) eq += ((int)(prob_bits(i) & 0x7fffu) == bpat);
  int* iscan = (int*)scan;
  __syncthreads();
  iscan[tid] = eq;
  __syncthreads();
  for (int off = 1; off < SAMPLE_THREADS; off <<= 1) {
    const int add = (tid >= off) ? iscan[tid - off] : 0;
    __syncthreads();
    iscan[tid] += add;
    __syncthreads();
  }
  int eq_before = iscan[tid] - eq;
  __syncthreads();
  // kept mass (sum of the surviving bf16 probabilities, fp32)
  float kept = 0.f;
  {
    int e = eq_before;
    for (int i = i0; i < i1; ++i) {
      const int pat = (int)(prob_bits(i) & 0x7fffu);
      bool keep = pat > bpat;
      if (pat == bpat) keep = (e++ < n_boundary);
      if (keep) kept += bf2f((bf16_t)pat);
    }
  }
  const float total = block_sum(kept, red);
  const float denom = bf2f(f2bf(total));  // probs_sort.sum(): a bf16 tensor
  // renormalised probabilities bf16(p / denom); chunk masses for the draw
  float cm = 0.f;
  {
    int e = eq_before;
    for (int i = i0; i < i1; ++i) {
      const int pat = (int)(prob_bits(i) & 0x7fffu);
      bool keep = pat > bpat;
      if (pat == bpat) keep = (e++ < n_boundary);
      const bf16_t q = keep ? f2bf(bf2f((bf16_t)pat) / denom) : (bf16_t)0;
      if (probs_out) probs_out[(int64_t)b * ldp + i] = q;
      cm += bf2f(q);
    }
  }
  __syncthreads();
  scan[tid] = cm;
  __syncthreads();
  for (int off = 1; off < SAMPLE_THREADS; off <<= 1) {
    const float add = (tid >= off) ? scan[tid - off] : 0.f;
    __syncthreads();
    scan[tid] += add;
    __syncthreads();
  }
  const float all = scan[SAMPLE_THREADS - 1];
  const float target = uniforms[b] * all;
  const float lo = scan[tid] - cm;
  int last = -1;
  {
    float acc = lo;
    int e = eq_before, pick = -1;
    const bool mine = cm > 0.f && target >= lo && target < scan[tid];
    for (int i = i0; i < i1; ++i) {
      const int pat = (int)(prob_bits(i) & 0x7fffu);
      bool keep = pat > bpat;
      if (pat == bpat) keep = (e++ < n_boundary);
      if (!keep) continue;
      last = i;
      if (mine && pick < 0) {
        acc += bf2f(f2bf(bf2f((bf16_t)pat) / denom));
        if (target < acc) pick = i;
      }
    }
    if (mine) shared_i[3] = pick >= 0 ? pick : last;  // the intervals [lo, scan) partition [0, all): one owner
  }
  if (last >= 0) atomicMax(&shared_i[4], last);
  __syncthreads();
  if (tid == 0) next[b] = shared_i[3] >= 0 ? shared_i[3] : (shared_i[4] > 0 ? shared_i[4] : 0);
}

}  // namespace

extern "C" md_status md_fourier_features(const void* x, int64_t ldx, int32_t rows, int32_t in_dim, const void* w,
                                         int32_t half, void* out, int64_t ld_out, void* stream) {
  MD_CHECK_ARG(x && w && out && rows > 0 && (in_dim == 1 || in_dim == 2) && half > 0 && ld_out >= 2 * half && ldx >= in_dim);
  hipLaunchKernelGGL(fourier_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, in_dim,
                     (const bf16_t*)w, half, (bf16_t*)out, ld_out);
  return md_launch_status();
}

extern "C" md_status md_region_pick_encode(const void* logits, int64_t ld, int32_t batch, int32_t n_groups,
                                           int32_t n_bins, const void* value_table, const void* feat_w, int32_t half,
                                           int32_t* bins, int64_t ld_bins, void* feats, int64_t ld_feats, void* stream) {
  MD_CHECK_ARG(logits && value_table && feat_w && bins && feats && batch > 0 && n_bins > 0);
  MD_CHECK_ARG((n_groups == 1 || n_groups == 2) && ld >= (int64_t)n_groups * n_bins && ld_bins >= n_groups && ld_feats >= 2 * half);
  hipLaunchKernelGGL(region_pick_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, n_groups,
                     n_bins, (const bf16_t*)value_table, (const bf16_t*)feat_w, half, bins, ld_bins, (bf16_t*)feats, ld_feats);
  return md_launch_status();
}

extern "C" md_status md_sample_top_p(const void* logits, int64_t ld, int32_t batch, int32_t vocab, int32_t suppress_id,
                                     float temperature, float top_p, const float* uniforms, int32_t* next,
                                     void* probs_out, int64_t ld_probs, void* stream) {
  MD_CHECK_ARG(logits && uniforms && next && batch > 0 && vocab > 0 && ld >= vocab && temperature > 0.f);
  MD_CHECK_ARG(probs_out == nullptr || ld_probs >= vocab);
  constexpr int lds = HIST_BINS * 4 + 16 * 4 + SAMPLE_THREADS * 4 + 16 * 4;
  MD_TRY(md_ensure_dynamic_lds((const void*)sample_top_p_kernel, lds));
  hipLaunchKernelGGL(sample_top_p_kernel, dim3(batch), dim3(SAMPLE_THREADS), lds, (hipStream_t)stream, (const bf16_t*)logits, ld,
                     vocab, suppress_id, temperature, top_p, uniforms, next, (bf16_t*)probs_out, ld_probs);
  return md_launch_status();
}
