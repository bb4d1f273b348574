// Single-sequence decode step as ONE persistent kernel (gfx950): all decoder blocks of one token -- layer norm, the fused
// qkv|fc1 linear with GELU, RoPE + KV-slab write, attention over the slab, proj and fc2 with both residual adds -- run
// inside one launch of one workgroup per CU, separated by grid barriers instead of kernel boundaries (reference: the
// body of _decode_one_tok, moondream.py:183-192 -> text.py:128-160, at batch 1).
//
// Why: at one sequence a decoder block is ~100 MB of weights (17 us at 6 TB/s) but four launches at ~4.5 us each
// (profiles/r02 B = 1 trace); a grid barrier with one counter per XCD costs 2.0 us (profiles/r02_grid_barrier_probe.txt).
//
// Shape of the computation (one row, so every linear is a matrix-vector product -- no MFMA, no LDS tiling):
//   phase A  every workgroup: x (residual stream) -> layer norm -> LDS; every WAVE owns pairs of adjacent output rows of
//            the fused [qkv | fc1] matrix (row pair p = wave id + n_waves * i): two 16-byte weight loads per lane and
//            512-feature chunk, fp32 FMAs against the activation chunk in LDS, wave reduction, bias, bf16 rounding,
//            GELU on the fc1 rows (the reference's rounding points).
//   phase B  workgroup (head h, slice s of the keys): RoPE of q and k (the slice that holds the new position also writes
//            K / V to the slab), scores, softmax statistics and P.V over its key slice -> partial (max, sum, out[64]).
//   phase C  every workgroup combines the partials of all heads into the attention row (LDS) and stages gelu(fc1);
//            every wave owns one pair of rows of proj AND the same pair of fc2: x' = bf16(bf16(x + bf16(proj + b)) +
//            bf16(fc2 + b)) -- the two sequential bf16 adds of text.py:157-158.
// Activations that cross workgroups (x, qkv|fc1 row, attention partials) move through global memory with agent-scope
// atomic loads / stores (L2-bypassing: the 8 XCDs have private L2s), so the barriers need no cache write-back.
//
// Numerics: fp32 dot products in a different association than the MFMA kernels (lane-strided partial sums + a wave
// butterfly), softmax with per-slice maxima (flash-decoding) -- within the same tolerance of the reference as the
// batched path, not bit-identical to it.  Weights: the SAME packed row-major bf16 matrices as every other kernel.
#include "md_common.hpp"

#include <cstdlib>

namespace {

constexpr int B1_MAX_LAYERS = 32;
constexpr int B1_SLICES = 8;        // key slices per head
constexpr int B1_PART = 66;         // floats per attention partial: max, sum, out[64]
constexpr unsigned B1_SPIN_LIMIT = 4000000u;

struct B1Layer {
  const bf16_t *ln_w, *ln_b, *w1, *b1, *wp, *bp, *w2, *b2;
};
struct B1Args {
  B1Layer layer[B1_MAX_LAYERS];
  int n_layers, dim, n_heads, ff, qkv_w;  // qkv_w = 3 * dim
  int ld1, ldp, ld2;                      // leading dimensions (k_pad) of the three matrices
  int rot, ctx;
  const float* freqs;
  bf16_t* kslab;
  bf16_t* vslab;
  int64_t layer_stride;
  const int32_t* pos;
  bf16_t* x[2];      // residual stream, ping-pong; x[0] holds the input embedding
  bf16_t* act;       // [qkv_w + ff]: the fused linear's output row
  float* part;       // [n_heads][B1_SLICES][B1_PART]
  unsigned* sync;    // [0] epoch, [64 (1 + xcd)] per-XCD arrivals, [64 * 9] top-level arrivals, [64 * 10] flag, [64 * 11] error
  float eps, scale_log2;
};

// ---- agent-scope (L2-bypassing) accesses for data that crosses workgroups -------------------------------------------
__device__ __forceinline__ uint32_t ld_coh32(const void* p) {
  return __hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t ld_coh64(const void* p) {
  return __hip_atomic_load((const uint64_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_coh32(void* p, uint32_t v) {
  __hip_atomic_store((uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Grid barrier: one arrival counter per XCD (workgroup b runs on XCD b % 8), the last arrival of an XCD arrives at the
// top-level counter, the last of those publishes the flag.  Counters only grow: targets are offsets from the epoch read
// at kernel start.  Spins are bounded: a barrier that times out raises the error word and lets the kernel finish.
struct GridBarrier {
  unsigned* sync;
  unsigned base;
  unsigned count = 0;
  __device__ void arrive_and_wait() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this workgroup's stores have been issued and acknowledged
    __syncthreads();
    ++count;
    if (threadIdx.x == 0) {
      const int nwg = gridDim.x, xcd = blockIdx.x & 7, per = (nwg + 7 - xcd) / 8, groups = nwg < 8 ? nwg : 8;
      const unsigned target = base + count;
      bool last = false;
      const unsigned a1 = __hip_atomic_fetch_add(sync + 64 * (1 + xcd), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a1 + 1u == (unsigned)per * target) {
        const unsigned a2 = __hip_atomic_fetch_add(sync + 64 * 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (a2 + 1u == (unsigned)groups * target);
      }
      if (last) {
        __hip_atomic_store(sync + 64 * 10, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(sync + 64 * 10, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
          if (++spins > B1_SPIN_LIMIT) {
            __hip_atomic_store(sync + 64 * 11, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    __syncthreads();
  }
};

// NWV waves per workgroup (8 or 16): one wave per SIMD leaves the load latency exposed
template <int NWV>
__device__ __forceinline__ float block_sum(float v, float* red) {  // red: NWV floats of LDS
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NWV; ++w) t += red[w];
  return t;
}
template <int NWV>
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int w = 1; w < NWV; ++w) t = fmaxf(t, red[w]);
  return t;
}

// R rows of a row-major bf16 matrix times an activation vector staged in LDS (bf16, K features, K % 8 == 0), by ONE wave:
// lane l takes the 8 features [512 c + 8 l, +8) of every chunk c; DEPTH chunks (R x 16 B per lane each) stay in flight.
// Branch-free around its loads (clamped addresses, zeroed operands) so that the compiler counts the in-order returns.
// prime() issues the first DEPTH chunks (the weights do not depend on the activations: it is called before the phase's
// prologue); run() returns the R dot products in every lane.
template <int R, int DEPTH>
struct Gemv {
  u32x4 w[DEPTH][R];
  __device__ __forceinline__ void issue(const bf16_t* const (&rows)[R], int K, int c, u32x4 (&dst)[R]) {
    const int lane = threadIdx.x & 63, nchunk = (K + 511) / 512;
    const int k = 512 * min(c, nchunk - 1) + 8 * lane;
    const int kk = k < K ? k : 0;
#pragma unroll
    for (int r = 0; r < R; ++r) dst[r] = *(const u32x4*)(rows[r] + kk);
  }
  __device__ __forceinline__ void prime(const bf16_t* const (&rows)[R], int K) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) issue(rows, K, u, w[u]);
  }
  __device__ __forceinline__ void run(const bf16_t* const (&rows)[R], int K, const char* act_lds, float (&d)[R]) {
    const int lane = threadIdx.x & 63, nchunk = (K + 511) / 512;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    float a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = 0.f;
    for (int c0 = 0; c0 < nchunk; c0 += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        const int c = c0 + u;
        const int k = 512 * min(c, nchunk - 1) + 8 * lane;
        const bool live = c < nchunk && k < K;
        const u32x4 av = live ? *(const u32x4*)(act_lds + k * 2) : zero4;  // dead chunks / lanes past K multiply by zero
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const u32x4 x = w[u][r];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a[r] = fmaf(lo_bf(av[e]), lo_bf(x[e]), a[r]);
            a[r] = fmaf(hi_bf(av[e]), hi_bf(x[e]), a[r]);
          }
        }
        issue(rows, K, c + DEPTH, w[u]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) d[r] = wave_sum(a[r]);
  }
};

constexpr int B1_MAX_DIM = 4096, B1_MAX_FF = 16384, B1_MAX_KEYS = 2048 / B1_SLICES;

template <int NWV>
__global__ __launch_bounds__(64 * NWV) void decode_b1_kernel(const B1Args p) {
  __shared__ __attribute__((aligned(16))) char lds_row[B1_MAX_DIM * 2];  // ln(x) in phase A, the attention row in phase C
  __shared__ __attribute__((aligned(16))) char lds_ff[B1_MAX_FF * 2];    // gelu(fc1) in phase C; phase B: P.V partials of the 32 key groups
  __shared__ float red[NWV];
  __shared__ float sc[B1_MAX_KEYS];
  __shared__ float head_m[64 * B1_SLICES], head_l[64 * B1_SLICES];  // phase C: slice statistics of every head
  __shared__ __attribute__((aligned(16))) bf16_t newrow[3][64];      // phase B: rotated q, rotated k, v of the new token

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NT = 64 * NWV;
  GridBarrier bar{p.sync, ld_coh32(p.sync)};
  // measurement hook: with word 768 of the sync state non-zero, workgroup 0 stamps the 100 MHz real-time counter at
  // every phase boundary into words 1024.. (two per stamp)
  const bool stamp_on = blockIdx.x == 0 && tid == 0 && ld_coh32(p.sync + 64 * 12) != 0u;
  int n_stamp = 0;
  auto stamp = [&]() {
    if (stamp_on && n_stamp < 1400) {
      const uint64_t t = __builtin_amdgcn_s_memrealtime();
      p.sync[1024 + 2 * n_stamp] = (unsigned)t;
      p.sync[1024 + 2 * n_stamp + 1] = (unsigned)(t >> 32);
      ++n_stamp;
    }
  };
  stamp();
  const int pos = p.pos[0];
  const int L = pos + 1;
  const int D = p.dim, FF = p.ff, QW = p.qkv_w;

  for (int l = 0; l < p.n_layers; ++l) {
    const B1Layer& w = p.layer[l];
    const bf16_t* xc = p.x[l & 1];
    bf16_t* xn = p.x[(l + 1) & 1];

    // ================= phase A: layer norm (every workgroup, redundantly) + fused [qkv | fc1] rows ==================
    {
      // row pairs of the fused matrix: workgroup b owns pairs b + gridDim q, its wave w the q = w + NWV i of them
      // (consecutive q land on the four SIMDs in turn)
      const int np = (QW + FF) >> 1;
      const int npw = blockIdx.x < np ? (np - 1 - blockIdx.x) / gridDim.x + 1 : 0;
      const bf16_t* rows_a[2];
      auto set_rows_a = [&](int q) {
        const int pp = blockIdx.x + gridDim.x * min(q, npw - 1);
        rows_a[0] = w.w1 + (int64_t)(2 * pp) * p.ld1;
        rows_a[1] = rows_a[0] + p.ld1;
      };
      Gemv<2, 4> ga;
      if (npw > 0) {
        set_rows_a(wave);
        ga.prime(rows_a, D);  // before the layer norm: the weights do not depend on it
      }
      float v[8];
      float sum = 0.f;
      const int nch = D >> 3;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
      if (tid < nch) {
        const uint64_t q0 = ld_coh64(xc + tid * 8), q1 = ld_coh64(xc + tid * 8 + 4);
        const uint32_t q[4] = {(uint32_t)q0, (uint32_t)(q0 >> 32), (uint32_t)q1, (uint32_t)(q1 >> 32)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] = lo_bf(q[e]);
          v[2 * e + 1] = hi_bf(q[e]);
          sum += v[2 * e] + v[2 * e + 1];
        }
      }
      const float mean = block_sum<NWV>(sum, red) / (float)D;
      float ss = 0.f;
      if (tid < nch) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dlt = v[e] - mean;
          ss += dlt * dlt;
        }
      }
      const float rstd = rsqrtf(block_sum<NWV>(ss, red) / (float)D + p.eps);
      if (tid < nch) {
        const u32x4 wq = *(const u32x4*)(w.ln_w + tid * 8), bq = *(const u32x4*)(w.ln_b + tid * 8);
        u32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          out[e] = pack_bf16x2((v[2 * e] - mean) * rstd * lo_bf(wq[e]) + lo_bf(bq[e]),
                               (v[2 * e + 1] - mean) * rstd * hi_bf(wq[e]) + hi_bf(bq[e]));
        *(u32x4*)(lds_row + tid * 16) = out;
      }
      __syncthreads();
      for (int q = wave; q < npw; q += NWV) {
        if (q != wave) {
          set_rows_a(q);
          ga.prime(rows_a, D);
        }
        float d[2];
        ga.run(rows_a, D, lds_row, d);
        if (lane == 0) {
          const int pp = blockIdx.x + gridDim.x * q;
          const uint32_t bw = *(const uint32_t*)(w.b1 + 2 * pp);
          uint32_t o = pack_bf16x2(d[0] + lo_bf(bw), d[1] + hi_bf(bw));  // F.linear's rounding point
          if (2 * pp >= QW) {  // fc1 rows: gelu on the bf16 value, rounded again (layers.py:24-25,137)
            const md_f32x2 g = gelu_tanh_f32x2(md_f32x2{lo_bf(o), hi_bf(o)});
            o = pack_bf16x2(g[0], g[1]);
          }
          st_coh32(p.act + 2 * pp, o);
        }
      }
    }
    stamp();
    bar.arrive_and_wait();
    stamp();

    // ================= phase B: attention partials, one (head, key slice) per workgroup ==============================
    {
      const int chunk = (L + B1_SLICES - 1) / B1_SLICES;
      for (int idx = blockIdx.x; idx < p.n_heads * B1_SLICES; idx += gridDim.x) {
        const int h = idx / B1_SLICES, sl = idx % B1_SLICES;
        const int j0 = sl * chunk, j1 = min(L, j0 + chunk);
        bf16_t* kb = p.kslab + (int64_t)l * p.layer_stride + (int64_t)h * p.ctx * 64;
        bf16_t* vb = p.vslab + (int64_t)l * p.layer_stride + (int64_t)h * p.ctx * 64;
        float* out = p.part + (int64_t)idx * B1_PART;
        __syncthreads();  // LDS reuse across iterations
        if (j0 >= j1) {   // no keys in this slice
          if (tid < B1_PART) st_coh32(out + tid, tid == 0 ? 0xff800000u : 0u);
          continue;
        }
        // rotated q (every slice needs it) and, for the slice that holds the new position, rotated k and v (rope.py:20-48)
        {
          const int half = p.rot >> 1;
          auto act_at = [&](int col) -> float {  // one bf16 of the fused row, through a coherent 4-byte load
            const uint32_t wv = ld_coh32(p.act + (col & ~1));
            return (col & 1) ? hi_bf(wv) : lo_bf(wv);
          };
          if (tid < 2 * half) {
            const int which = tid / half, j = tid % half;
            const int base = (which ? (p.n_heads + h) : h) * 64;
            const float re = act_at(base + j), im = act_at(base + half + j);
            const float cs = p.freqs[((int64_t)pos * half + j) * 2], sn = p.freqs[((int64_t)pos * half + j) * 2 + 1];
            newrow[which][2 * j] = f2bf(__fsub_rn(__fmul_rn(re, cs), __fmul_rn(im, sn)));
            newrow[which][2 * j + 1] = f2bf(__fadd_rn(__fmul_rn(re, sn), __fmul_rn(im, cs)));
          } else if (tid >= 64 && tid < 64 + 2 * (64 - p.rot)) {
            const int t2 = tid - 64, which = t2 / (64 - p.rot), i = p.rot + t2 % (64 - p.rot);
            newrow[which][i] = f2bf(act_at((which ? (p.n_heads + h) : h) * 64 + i));
          } else if (tid >= 192) {
            newrow[2][tid - 192] = f2bf(act_at((2 * p.n_heads + h) * 64 + tid - 192));
          }
        }
        __syncthreads();
        const bool has_new = pos >= j0 && pos < j1;
        if (has_new) {
          if (tid < 64) kb[(int64_t)pos * 64 + tid] = newrow[1][tid];
          else if (tid < 128) vb[(int64_t)pos * 64 + tid - 64] = newrow[2][tid - 64];
        }
        const int g = tid >> 3, c = tid & 7;  // 128 key groups x 8 lanes per 128-byte row
        float qv[8];
        {
          const u32x4 qq = *(const u32x4*)(&newrow[0][c * 8]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            qv[2 * e] = lo_bf(qq[e]) * p.scale_log2;
            qv[2 * e + 1] = hi_bf(qq[e]) * p.scale_log2;
          }
        }
        constexpr int NG = NT / 8, KPT = (B1_MAX_KEYS + NG - 1) / NG;  // key groups; keys per group at most
        u32x4 kq[KPT], vq[KPT];
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
          const int j = min(j0 + g + NG * i, j1 - 1);
          kq[i] = *(const u32x4*)(kb + (int64_t)j * 64 + c * 8);
          vq[i] = *(const u32x4*)(vb + (int64_t)j * 64 + c * 8);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
          const int j = j0 + g + NG * i;
          const u32x4 kk = (j == pos) ? *(const u32x4*)(&newrow[1][c * 8]) : kq[i];  // the new key is not in global memory for this CU yet
          float sv = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) sv += qv[2 * e] * lo_bf(kk[e]) + qv[2 * e + 1] * hi_bf(kk[e]);
          sv += __shfl_xor(sv, 1, 64);
          sv += __shfl_xor(sv, 2, 64);
          sv += __shfl_xor(sv, 4, 64);
          if (j < j1) {
            if (c == 0) sc[j - j0] = sv;
            mx = fmaxf(mx, sv);
          }
        }
        mx = block_max<NWV>(mx, red);  // (its barriers also publish sc[])
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        float lsum = 0.f;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
          const int j = j0 + g + NG * i;
          if (j < j1) {
            const u32x4 vv = (j == pos) ? *(const u32x4*)(&newrow[2][c * 8]) : vq[i];
            const float pj = __builtin_amdgcn_exp2f(sc[j - j0] - mx);
            if (c == 0) lsum += pj;
            const float pr = bf2f(f2bf(pj));  // probabilities enter the second contraction as bf16
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc[2 * e] += pr * lo_bf(vv[e]);
              acc[2 * e + 1] += pr * hi_bf(vv[e]);
            }
          }
        }
        const float ltot = block_sum<NWV>(lsum, red);
        // P.V: the 8 key groups of a wave through a lane butterfly, the 16 waves through LDS
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          acc[e] += __shfl_xor(acc[e], 8, 64);
          acc[e] += __shfl_xor(acc[e], 16, 64);
          acc[e] += __shfl_xor(acc[e], 32, 64);
        }
        float* pv = (float*)lds_ff;  // [16 waves][64]
        if (lane < 8) {
#pragma unroll
          for (int e = 0; e < 8; ++e) pv[wave * 64 + lane * 8 + e] = acc[e];
        }
        __syncthreads();
        if (tid < 64) {
          float o = 0.f;
#pragma unroll
          for (int ww = 0; ww < NWV; ++ww) o += pv[ww * 64 + tid];
          st_coh32(out + 2 + tid, __float_as_uint(o));
        } else if (tid == 64) {
          st_coh32(out, __float_as_uint(mx));
        } else if (tid == 65) {
          st_coh32(out + 1, __float_as_uint(ltot));
        }
      }
    }
    stamp();
    bar.arrive_and_wait();
    stamp();

    // ================= phase C: attention row + gelu(fc1) into LDS; proj and fc2 row pairs, both residual adds =========
    {
      // workgroup b owns output pairs b + gridDim q; TWO waves share a pair: the first takes its proj rows and the first
      // c_split chunks of its fc2 rows, the second the rest of fc2 (equal chunk counts), so all waves stream in one pass
      // at 2B / 256 CUs (4 pairs per workgroup, 8 waves)
      const int npc = D >> 1;
      const int npw = blockIdx.x < npc ? (npc - 1 - blockIdx.x) / gridDim.x + 1 : 0;
      const int nchd = (D + 511) / 512, nch2 = (FF + 511) / 512;
      const int c_split = max(0, (nch2 - nchd) / 2);
      constexpr int PPP = NWV / 2;  // pairs per pass
      const int qw = wave >> 1, half = wave & 1;
      const bf16_t *prow[2], *frow[2];
      int kf = 0;
      const char* actf = lds_ff;
      auto set_rows_c = [&](int q) {
        const int pp = blockIdx.x + gridDim.x * min(q, npw - 1);
        prow[0] = w.wp + (int64_t)(2 * pp) * p.ldp;
        prow[1] = prow[0] + p.ldp;
        const int k0 = half ? 512 * c_split : 0, k1 = half ? FF : min(FF, 512 * c_split);
        frow[0] = w.w2 + (int64_t)(2 * pp) * p.ld2 + k0;
        frow[1] = frow[0] + p.ld2;
        kf = max(k1 - k0, 0);
        actf = lds_ff + k0 * 2;
      };
      Gemv<2, 4> gp, gf;
      if (qw < npw) {  // before the prologue: the weights do not depend on it
        set_rows_c(qw);
        if (half == 0) gp.prime(prow, D);
        if (kf > 0) gf.prime(frow, kf);
      }
      const int nhs = p.n_heads * B1_SLICES;
      for (int i = tid; i < nhs; i += NT) {
        head_m[i] = __uint_as_float(ld_coh32(p.part + (int64_t)i * B1_PART));
        head_l[i] = __uint_as_float(ld_coh32(p.part + (int64_t)i * B1_PART + 1));
      }
      for (int i = tid; i < (FF >> 2); i += NT) {  // 4 bf16 per coherent load
        const uint64_t q = ld_coh64(p.act + QW + 4 * i);
        *(uint64_t*)(lds_ff + 8 * i) = q;
      }
      __syncthreads();
      for (int e0 = tid; e0 < (D >> 1); e0 += NT) {  // two features of one head per thread and pass
        const int e = 2 * e0, h = e >> 6, d = e & 63;
        float M = -INFINITY;
#pragma unroll
        for (int s_ = 0; s_ < B1_SLICES; ++s_) M = fmaxf(M, head_m[h * B1_SLICES + s_]);
        float Lt = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < B1_SLICES; ++s_) {
          const float f = __builtin_amdgcn_exp2f(head_m[h * B1_SLICES + s_] - M);  // empty slice: exp2(-inf) = 0
          Lt += head_l[h * B1_SLICES + s_] * f;
          const uint64_t q = ld_coh64(p.part + (int64_t)(h * B1_SLICES + s_) * B1_PART + 2 + d);
          o0 += __uint_as_float((uint32_t)q) * f;
          o1 += __uint_as_float((uint32_t)(q >> 32)) * f;
        }
        const float inv = Lt > 0.f ? 1.0f / Lt : 0.f;
        *(uint32_t*)(lds_row + e * 2) = pack_bf16x2(o0 * inv, o1 * inv);
      }
      __syncthreads();
      float* part_c = (float*)sc;  // [PPP][3][2]: proj, fc2 first part, fc2 second part  (sc[] is free outside phase B)
      for (int q0 = 0; q0 < npw; q0 += PPP) {
        const int q = q0 + qw;
        float dp[2] = {0.f, 0.f}, df[2] = {0.f, 0.f};
        if (q < npw) {
          if (q0 > 0) {
            set_rows_c(q);
            if (half == 0) gp.prime(prow, D);
            if (kf > 0) gf.prime(frow, kf);
          }
          if (half == 0) gp.run(prow, D, lds_row, dp);
          if (kf > 0) gf.run(frow, kf, actf, df);
        }
        if (lane == 0) {
          if (half == 0) {
            part_c[(qw * 3) * 2] = dp[0];
            part_c[(qw * 3) * 2 + 1] = dp[1];
          }
          part_c[(qw * 3 + 1 + half) * 2] = df[0];
          part_c[(qw * 3 + 1 + half) * 2 + 1] = df[1];
        }
        __syncthreads();
        if (tid < PPP && q0 + tid < npw) {
          const int pc = blockIdx.x + gridDim.x * (q0 + tid);
          const float dp0 = part_c[(3 * tid) * 2], dp1 = part_c[(3 * tid) * 2 + 1];
          const float df0 = part_c[(3 * tid + 1) * 2] + part_c[(3 * tid + 2) * 2];
          const float df1 = part_c[(3 * tid + 1) * 2 + 1] + part_c[(3 * tid + 2) * 2 + 1];
          const uint32_t xo = ld_coh32(xc + 2 * pc);
          const uint32_t bpw = *(const uint32_t*)(w.bp + 2 * pc), b2w = *(const uint32_t*)(w.b2 + 2 * pc);
          const uint32_t t1 = pack_bf16x2(dp0 + lo_bf(bpw), dp1 + hi_bf(bpw));
          const uint32_t x1 = pack_bf16x2(lo_bf(xo) + lo_bf(t1), hi_bf(xo) + hi_bf(t1));
          const uint32_t t2 = pack_bf16x2(df0 + lo_bf(b2w), df1 + hi_bf(b2w));
          st_coh32(xn + 2 * pc, pack_bf16x2(lo_bf(x1) + lo_bf(t2), hi_bf(x1) + hi_bf(t2)));
        }
        __syncthreads();
      }
    }
    stamp();
    if (l + 1 < p.n_layers) bar.arrive_and_wait();
    stamp();
  }
  // the next launch continues the barrier targets where this one stopped (every workgroup read the epoch at its start)
  if (blockIdx.x == 0 && tid == 0) st_coh32(p.sync, bar.base + (unsigned)(3 * p.n_layers - 1));
}

}  // namespace

extern "C" size_t md_decode_b1_workspace_bytes(const md_text_model* m) {
  if (!m || !m->blocks) return 0;
  const size_t dim = m->dim, ffp = m->blocks[0].fc1.n_pad;
  return 2 * ((dim * 2 + 255) / 256 * 256) + ((3 * dim + ffp) * 2 + 255) / 256 * 256 +
         ((size_t)m->n_heads * B1_SLICES * B1_PART * 4 + 255) / 256 * 256;
}

// sync_state: >= 16 KiB of device memory, ZERO when first used and never written by anything else; it carries the barrier
// counters from one launch to the next.  hidden: bf16 [dim].  x_in: bf16 [dim] (the token embedding).
extern "C" md_status md_decode_b1_layers(const md_text_model* m, const void* x_in, void* hidden, const int32_t* pos,
                                         const md_kv_cache* kv, void* workspace, size_t workspace_bytes, void* sync_state,
                                         void* stream) {
  MD_CHECK_ARG(m && x_in && hidden && pos && kv && kv->k && kv->v && workspace && sync_state && m->blocks);
  MD_CHECK_ARG(m->n_layers >= 1 && m->n_layers <= B1_MAX_LAYERS && m->n_heads == m->n_kv_heads && m->dim == m->n_heads * 64);
  MD_CHECK_ARG(m->dim % 8 == 0 && m->dim <= B1_MAX_DIM && m->n_heads <= 64 && kv->ctx <= 2048);
  const md_text_block& b0 = m->blocks[0];
  MD_CHECK_ARG(b0.qkv_fc1.w && b0.qkv_fc1.b && b0.proj.b && b0.fc2.b);
  const int ff = b0.fc1.n;
  MD_CHECK_ARG(ff % 8 == 0 && ff <= B1_MAX_FF && (3 * m->dim + ff) % 2 == 0 && b0.qkv.n == 3 * m->dim);
  if (workspace_bytes < md_decode_b1_workspace_bytes(m)) return MD_ERR_WORKSPACE;
  B1Args a;
  for (int l = 0; l < m->n_layers; ++l) {
    const md_text_block& b = m->blocks[l];
    MD_CHECK_ARG(b.qkv_fc1.w && b.qkv_fc1.k_pad == b0.qkv_fc1.k_pad && b.proj.k_pad == b0.proj.k_pad && b.fc2.k_pad == b0.fc2.k_pad);
    a.layer[l] = B1Layer{(const bf16_t*)b.ln.w,      (const bf16_t*)b.ln.b,   (const bf16_t*)b.qkv_fc1.w, (const bf16_t*)b.qkv_fc1.b,
                         (const bf16_t*)b.proj.w,    (const bf16_t*)b.proj.b, (const bf16_t*)b.fc2.w,     (const bf16_t*)b.fc2.b};
  }
  a.n_layers = m->n_layers;
  a.dim = m->dim;
  a.n_heads = m->n_heads;
  a.ff = ff;
  a.qkv_w = 3 * m->dim;
  a.ld1 = b0.qkv_fc1.k_pad;
  a.ldp = b0.proj.k_pad;
  a.ld2 = b0.fc2.k_pad;
  a.rot = m->rot_dim;
  a.ctx = kv->ctx;
  a.freqs = m->freqs;
  a.kslab = (bf16_t*)kv->k;
  a.vslab = (bf16_t*)kv->v;
  a.layer_stride = kv->layer_stride;
  a.pos = pos;
  char* ws = (char*)workspace;
  const size_t xb = ((size_t)m->dim * 2 + 255) / 256 * 256, ab = ((size_t)(3 * m->dim + b0.fc1.n_pad) * 2 + 255) / 256 * 256;
  a.x[0] = (bf16_t*)ws;
  a.x[1] = (bf16_t*)(ws + xb);
  a.act = (bf16_t*)(ws + 2 * xb);
  a.part = (float*)(ws + 2 * xb + ab);
  a.sync = (unsigned*)sync_state;
  a.eps = 1e-5f;
  a.scale_log2 = 0.125f * 1.4426950408889634f;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemcpyAsync(a.x[0], x_in, (size_t)m->dim * 2, hipMemcpyDeviceToDevice, s) != hipSuccess) return MD_ERR_LAUNCH;
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n >= 8 ? n : 8;
  }();
  // 8 waves per workgroup (two per SIMD, 256 registers each); MD_B1_WAVES=16: the four-per-SIMD variant (spills) for A/B runs
  static const int nwv = [] { const char* e = getenv("MD_B1_WAVES"); return (e && atoi(e) == 16) ? 16 : 8; }();
  if (nwv == 8) hipLaunchKernelGGL(decode_b1_kernel<8>, dim3(n_cu), dim3(512), 0, s, a);
  else hipLaunchKernelGGL(decode_b1_kernel<16>, dim3(n_cu), dim3(1024), 0, s, a);
  MD_TRY(md_launch_status());
  if (hipMemcpyAsync(hidden, a.x[m->n_layers & 1], (size_t)m->dim * 2, hipMemcpyDeviceToDevice, s) != hipSuccess) return MD_ERR_LAUNCH;
  return MD_OK;
}
