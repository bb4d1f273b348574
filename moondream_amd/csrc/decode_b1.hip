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
constexpr int B1_PART = 68;         // floats per attention partial: max, sum, two pad words, out[64] (16-byte aligned)
constexpr unsigned B1_SPIN_LIMIT = 400000u;

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
  const bf16_t* x_first;  // the first block's input (nullptr: row token[0] of wte)
  bf16_t* x[2];           // residual stream between blocks, ping-pong
  bf16_t* x_last;         // the last block's output (nullptr: x[n_layers & 1])
  // whole decode step (lm_w != nullptr): embedding lookup in front, final layer norm + lm_head + greedy choice behind
  const int32_t* token;
  const bf16_t *wte, *post_ln_w, *post_ln_b, *lm_w, *lm_b;
  int ld_lm, vocab, suppress;
  bf16_t* logits;
  int32_t *next, *pos_out;
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
// 16-byte form: a buffer load with the sc1 bit (what the agent-scope atomic loads above compile to), tracked by the
// compiler's wait counters like any other load
__device__ __forceinline__ __amdgpu_buffer_rsrc_t coh_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xffffffffu, 0x00020000);
}
__device__ __forceinline__ u32x4 ld_coh128(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
__device__ __forceinline__ void st_coh32(void* p, uint32_t v) {
  __hip_atomic_store((uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Grid barrier without atomics: workgroup b publishes the barrier's number in ITS word of a flag array once its stores
// are acknowledged; everybody polls the whole array (one 4-byte coherent load per lane, gridDim / 64 waves) until every
// word has reached the number.  Round trips on the critical path: store acknowledgement, flag store, one poll -- the
// counter tree this replaces (per-XCD counter, top-level counter, flag) had two atomic round trips more
// (profiles/r02_decode_b1_persistent_phase_times.txt).  Numbers only grow: targets are offsets from the epoch read at
// kernel start.  Spins are bounded: a barrier that times out raises the error word and lets the kernel finish.
// arrive() and wait() are separate calls: work that depends on neither side of the barrier (weight rows, K / V rows of
// earlier tokens) goes between them.
// Words of the sync state: [0] epoch, [64 * 11] error, [64 * 12] stamp switch, [1024 + b] flag of workgroup b,
// stamps from word 2048.
constexpr int B1_FLAG_WORD = 1024, B1_STAMP_WORD = 2048, B1_MAX_WG = 1024;
struct GridBarrier {
  unsigned* sync;
  unsigned base;
  unsigned count = 0;
  unsigned target = 0;
  __device__ void arrive() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this workgroup's stores have been issued and acknowledged
    __syncthreads();
    ++count;
    target = base + count;
    if (threadIdx.x == 0) __hip_atomic_store(sync + B1_FLAG_WORD + blockIdx.x, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ void wait() {
    const int nwg = gridDim.x;
    for (int i0 = (threadIdx.x >> 6) * 64; i0 < nwg; i0 += blockDim.x) {  // wave w: flags [64 w, 64 w + 64) (+ blockDim ...)
      const int i = i0 + (threadIdx.x & 63);
      unsigned spins = 0;
      while (true) {
        const bool ok = i >= nwg || (int)(__hip_atomic_load(sync + B1_FLAG_WORD + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0;
        if (__all(ok)) break;
        if (++spins > B1_SPIN_LIMIT) {
          __hip_atomic_store(sync + 64 * 11, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  }
};

// NWV waves per workgroup: one wave per SIMD leaves the load latency exposed
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
  // (a conditional prime() needs this on the other path: otherwise the ring is live around the whole layer loop)
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u)
#pragma unroll
      for (int r = 0; r < R; ++r) w[u][r] = u32x4{0u, 0u, 0u, 0u};
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

constexpr int B1_MAX_DIM = 4096, B1_MAX_FF = 8192, B1_MAX_KEYS = 2048 / B1_SLICES, B1_MAX_Q = 32;

template <int NWV>
__global__ __launch_bounds__(64 * NWV) void decode_b1_kernel(const B1Args p) {
  __shared__ __attribute__((aligned(16))) char lds_row[B1_MAX_DIM * 2];  // ln(x) in phase A / D, the attention row in phase C
  __shared__ __attribute__((aligned(16))) char lds_ff[B1_MAX_FF * 2];    // gelu(fc1), staged at the start of phase B
  __shared__ float red[NWV];
  __shared__ float pv_lds[NWV * 64];  // phase B: P.V partials of the waves
  __shared__ float sc[B1_MAX_KEYS];
  __shared__ float head_m[64 * B1_SLICES], head_l[64 * B1_SLICES];  // phase C: slice statistics of every head
  __shared__ __attribute__((aligned(16))) bf16_t newrow[3][64];      // phase B: rotated q, rotated k, v of the new token
  __shared__ float fc2_part[B1_MAX_Q][2][2];                         // fc2 dot products: [pair of the workgroup][K half][row]
  __shared__ float proj_part[NWV / 2][2][2];
  __shared__ int best_i[NWV];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NT = 64 * NWV;
  GridBarrier bar{p.sync, ld_coh32(p.sync)};
  // measurement hook: with word 768 of the sync state non-zero, workgroup 0 stamps the 100 MHz real-time counter at
  // every phase boundary into words 2048.. (two per stamp)
  const bool stamp_on = blockIdx.x == 0 && tid == 0 && ld_coh32(p.sync + 64 * 12) != 0u;
  int n_stamp = 0;
  auto stamp = [&]() {
    if (stamp_on && n_stamp < 1000) {
      const uint64_t t = __builtin_amdgcn_s_memrealtime();
      p.sync[B1_STAMP_WORD + 2 * n_stamp] = (unsigned)t;
      p.sync[B1_STAMP_WORD + 2 * n_stamp + 1] = (unsigned)(t >> 32);
      ++n_stamp;
    }
  };
  stamp();
  const int pos = p.pos[0];
  const int L = pos + 1;
  const int D = p.dim, FF = p.ff, QW = p.qkv_w;

  // ---- work lists (the same for every layer) ---------------------------------------------------------------------------
  // phase A: row pairs of the fused [qkv | fc1] matrix: workgroup b owns pairs b + gridDim q, its wave w the q = w + NWV i
  const int np_a = (QW + FF) >> 1;
  const int npw_a = blockIdx.x < np_a ? (np_a - 1 - blockIdx.x) / gridDim.x + 1 : 0;
  // phases B / C: output pairs b + gridDim q of proj and fc2; TWO waves share a pair, each takes half of the features
  const int np_c = D >> 1;
  const int npw_c = blockIdx.x < np_c ? (np_c - 1 - blockIdx.x) / gridDim.x + 1 : 0;
  const int nchd = (D + 511) / 512, nch2 = (FF + 511) / 512;
  constexpr int PPP = NWV / 2;  // pairs per pass
  const int qw = wave >> 1, half = wave & 1;
  // phase D: row pairs of lm_head
  const int np_d = p.vocab >> 1;
  const int npw_d = (p.lm_w && blockIdx.x < np_d) ? (np_d - 1 - blockIdx.x) / gridDim.x + 1 : 0;

  const bf16_t* rows_a[2];
  auto set_rows = [&](const bf16_t* wmat, int ld, int q) {
    const int pp = blockIdx.x + gridDim.x * q;
    rows_a[0] = wmat + (int64_t)(2 * pp) * ld;
    rows_a[1] = rows_a[0] + ld;
  };
  Gemv<2, 4> ga;  // phases A and D; primed across the barrier that precedes the phase (the weights depend on nothing)
  auto prime_first = [&](int npw, const bf16_t* wmat, int ld) {
    if (wave < npw) {
      set_rows(wmat, ld, wave);
      ga.prime(rows_a, D);
    } else {
      ga.zero();
    }
  };
  // a workgroup's npw row pairs, round-robin over its waves (pair q = wave + NWV i: consecutive pairs land on the four
  // SIMDs in turn); finish(pp, d0, d1) is called by every lane of the wave that holds the two dot products of row pair pp.
  // (Sharing the last npw % NWV pairs between waves by K ranges evens the waves out on paper and measured slower.)
  auto run_rows = [&](int npw, const bf16_t* wmat, int ld, auto finish) {
    for (int q = wave; q < npw; q += NWV) {
      if (q != wave) {
        set_rows(wmat, ld, q);
        ga.prime(rows_a, D);
      }
      float d[2];
      ga.run(rows_a, D, lds_row, d);
      finish(blockIdx.x + gridDim.x * q, d[0], d[1]);
    }
  };
  prime_first(npw_a, p.layer[0].w1, p.ld1);

  // layer norm of the residual stream (every workgroup, redundantly) -> lds_row
  auto layer_norm_to_lds = [&](const bf16_t* xsrc, const bf16_t* lnw, const bf16_t* lnb) {
    float v[8];
    float sum = 0.f;
    const int nch = D >> 3;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (tid < nch) {
      const u32x4 q = ld_coh128(coh_rsrc(xsrc), tid * 16);
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
      const u32x4 wq = *(const u32x4*)(lnw + tid * 8), bq = *(const u32x4*)(lnb + tid * 8);
      u32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        out[e] = pack_bf16x2((v[2 * e] - mean) * rstd * lo_bf(wq[e]) + lo_bf(bq[e]),
                             (v[2 * e + 1] - mean) * rstd * hi_bf(wq[e]) + hi_bf(bq[e]));
      *(u32x4*)(lds_row + tid * 16) = out;
    }
    __syncthreads();
  };

  // phase B's key / value rows of this workgroup's first (head, slice): they were written by earlier launches, so their
  // loads go out while the barrier in front of the phase is still collecting arrivals
  constexpr int NG = NT / 8, KPT = (B1_MAX_KEYS + NG - 1) / NG;  // key groups (8 lanes per 128-byte row); keys per group at most
  const int chunk = (L + B1_SLICES - 1) / B1_SLICES;
  const int g = tid >> 3, c = tid & 7;
  auto load_kv = [&](int l, int idx, u32x4 (&kq)[KPT], u32x4 (&vq)[KPT]) {
    const int h = idx / B1_SLICES, sl = idx % B1_SLICES;
    const int j0 = sl * chunk, j1 = max(min(L, j0 + chunk), j0 + 1);
    const bf16_t* kb = p.kslab + (int64_t)l * p.layer_stride + (int64_t)h * p.ctx * 64;
    const bf16_t* vb = p.vslab + (int64_t)l * p.layer_stride + (int64_t)h * p.ctx * 64;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int j = min(min(j0 + g + NG * i, j1 - 1), p.ctx - 1);
      kq[i] = *(const u32x4*)(kb + (int64_t)j * 64 + c * 8);
      vq[i] = *(const u32x4*)(vb + (int64_t)j * 64 + c * 8);
    }
  };
  const int n_items = p.n_heads * B1_SLICES;

  const bf16_t* xc = p.x_first;
  if (!xc) xc = p.wte + (int64_t)min(max(p.token[0], 0), p.vocab - 1) * D;
  for (int l = 0; l < p.n_layers; ++l) {
    const B1Layer& w = p.layer[l];
    bf16_t* xn = (l + 1 == p.n_layers && p.x_last) ? p.x_last : p.x[(l + 1) & 1];

    // ================= phase A: layer norm + fused [qkv | fc1] rows ===================================================
    layer_norm_to_lds(xc, w.ln_w, w.ln_b);
    run_rows(npw_a, w.w1, p.ld1, [&](int pp, float d0, float d1) {
      if (lane == 0) {
        const uint32_t bw = *(const uint32_t*)(w.b1 + 2 * pp);
        uint32_t o = pack_bf16x2(d0 + lo_bf(bw), d1 + hi_bf(bw));  // F.linear's rounding point
        if (2 * pp >= QW) {  // fc1 rows: gelu on the bf16 value, rounded again (layers.py:24-25,137)
          const md_f32x2 gl = gelu_tanh_f32x2(md_f32x2{lo_bf(o), hi_bf(o)});
          o = pack_bf16x2(gl[0], gl[1]);
        }
        st_coh32(p.act + 2 * pp, o);
      }
    });
    stamp();
    bar.arrive();
    u32x4 kq[KPT], vq[KPT];
    if ((int)blockIdx.x < n_items) {
      load_kv(l, blockIdx.x, kq, vq);
    } else {
#pragma unroll
      for (int i = 0; i < KPT; ++i) kq[i] = vq[i] = u32x4{0u, 0u, 0u, 0u};
    }
    bar.wait();
    stamp();

    // ================= phase B: attention partials, one (head, key slice) per workgroup; then the fc2 rows ============
    // gelu(fc1) -> LDS for the fc2 rows: requested now, parked in LDS after the attention slice (the latency of these
    // loads runs under it)
    constexpr int NFQ = B1_MAX_FF / 8 / NT > 0 ? B1_MAX_FF / 8 / NT : 1;
    u32x4 ffq[NFQ];
    {
      const __amdgpu_buffer_rsrc_t r_ff = coh_rsrc(p.act + QW);
#pragma unroll
      for (int i = 0; i < NFQ; ++i) {
        const int ch = tid + NT * i;
        ffq[i] = ld_coh128(r_ff, (ch < (FF >> 3) ? ch : 0) * 16);
      }
    }
    for (int idx = blockIdx.x; idx < n_items; idx += gridDim.x) {
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
      if (idx != (int)blockIdx.x) load_kv(l, idx, kq, vq);
      // rotated q (every slice needs it) and, for the slice that holds the new position, rotated k and v (rope.py:20-48)
      {
        const int hr = p.rot >> 1;
        auto act_at = [&](int col) -> float {  // one bf16 of the fused row, through a coherent 4-byte load
          const uint32_t wv = ld_coh32(p.act + (col & ~1));
          return (col & 1) ? hi_bf(wv) : lo_bf(wv);
        };
        if (tid < 2 * hr) {
          const int which = tid / hr, j = tid % hr;
          const int base = (which ? (p.n_heads + h) : h) * 64;
          const float re = act_at(base + j), im = act_at(base + hr + j);
          const float cs = p.freqs[((int64_t)pos * hr + j) * 2], sn = p.freqs[((int64_t)pos * hr + j) * 2 + 1];
          float o_re, o_im;
          md_rope_pair(re, im, cs, sn, o_re, o_im);
          newrow[which][2 * j] = f2bf(o_re);
          newrow[which][2 * j + 1] = f2bf(o_im);
        } else if (tid >= 64 && tid < 64 + 2 * (64 - p.rot)) {
          const int t2 = tid - 64, which = t2 / (64 - p.rot), i = p.rot + t2 % (64 - p.rot);
          newrow[which][i] = f2bf(act_at((which ? (p.n_heads + h) : h) * 64 + i));
        } else if (tid >= 192 && tid < 256) {
          newrow[2][tid - 192] = f2bf(act_at((2 * p.n_heads + h) * 64 + tid - 192));
        }
      }
      __syncthreads();
      const bool has_new = pos >= j0 && pos < j1;
      if (has_new) {
        if (tid < 64) kb[(int64_t)pos * 64 + tid] = newrow[1][tid];
        else if (tid < 128) vb[(int64_t)pos * 64 + tid - 64] = newrow[2][tid - 64];
      }
      float qv[8];
      {
        const u32x4 qq = *(const u32x4*)(&newrow[0][c * 8]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          qv[2 * e] = lo_bf(qq[e]) * p.scale_log2;
          qv[2 * e + 1] = hi_bf(qq[e]) * p.scale_log2;
        }
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
      // P.V: the 8 key groups of a wave through a lane butterfly, the waves through LDS
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        acc[e] += __shfl_xor(acc[e], 8, 64);
        acc[e] += __shfl_xor(acc[e], 16, 64);
        acc[e] += __shfl_xor(acc[e], 32, 64);
      }
      if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) pv_lds[wave * 64 + lane * 8 + e] = acc[e];
      }
      __syncthreads();
      if (tid < 64) {
        float o = 0.f;
#pragma unroll
        for (int ww = 0; ww < NWV; ++ww) o += pv_lds[ww * 64 + tid];
        st_coh32(out + 4 + tid, __float_as_uint(o));
      } else if (tid == 64) {
        st_coh32(out, __float_as_uint(mx));
      } else if (tid == 65) {
        st_coh32(out + 1, __float_as_uint(ltot));
      }
    }
#pragma unroll
    for (int i = 0; i < NFQ; ++i) {
      const int ch = tid + NT * i;
      if (ch < (FF >> 3)) *(u32x4*)(lds_ff + 16 * ch) = ffq[i];
    }
    stamp();
    bar.arrive();  // (also publishes lds_ff to every wave)
    // fc2 needs gelu(fc1), not the attention: its rows stream while the barrier collects the attention partials
    const bf16_t *prow[2], *frow[2];
    int kp = 0, kf = 0;
    const char *actp = lds_row, *actf = lds_ff;
    auto set_rows_c = [&](int q) {
      const int pp = blockIdx.x + gridDim.x * min(q, npw_c - 1);
      const int cp0 = half ? nchd / 2 : 0, cp1 = half ? nchd : nchd / 2;
      prow[0] = w.wp + (int64_t)(2 * pp) * p.ldp + 512 * cp0;
      prow[1] = prow[0] + p.ldp;
      kp = max(min(D, 512 * cp1) - 512 * cp0, 0);
      actp = lds_row + 512 * cp0 * 2;
      const int c0 = half ? nch2 / 2 : 0, c1 = half ? nch2 : nch2 / 2;
      frow[0] = w.w2 + (int64_t)(2 * pp) * p.ld2 + 512 * c0;
      frow[1] = frow[0] + p.ld2;
      kf = max(min(FF, 512 * c1) - 512 * c0, 0);
      actf = lds_ff + 512 * c0 * 2;
    };
    for (int q0 = 0; q0 < npw_c; q0 += PPP) {
      const int q = q0 + qw;
      if (q < npw_c) {
        set_rows_c(q);
        float df[2] = {0.f, 0.f};
        if (kf > 0) {
          Gemv<2, 4> gf;
          gf.prime(frow, kf);
          gf.run(frow, kf, actf, df);
        }
        if (lane == 0) {
          fc2_part[q][half][0] = df[0];
          fc2_part[q][half][1] = df[1];
        }
      }
    }
    Gemv<2, 2> gp;  // proj rows of the first pass: in flight across the wait and the combination of the partials
    gp.zero();
    if (qw < npw_c) {
      set_rows_c(qw);
      if (kp > 0) gp.prime(prow, kp);
    }
    // the residual operand and the biases of the first pass's pairs
    uint32_t xo0 = 0, bpw0 = 0, b2w0 = 0;
    if (tid < PPP && tid < npw_c) {
      const int pc = blockIdx.x + gridDim.x * tid;
      xo0 = ld_coh32(xc + 2 * pc);
      bpw0 = *(const uint32_t*)(w.bp + 2 * pc);
      b2w0 = *(const uint32_t*)(w.b2 + 2 * pc);
    }
    bar.wait();
    stamp();

    // ================= phase C: attention row into LDS; proj row pairs, both residual adds ============================
    {
      // one round trip: the slice statistics and this thread's four features of all slices of its head
      const __amdgpu_buffer_rsrc_t r_part = coh_rsrc(p.part);
      u32x4 po[B1_SLICES];
      for (int i = tid; i < n_items; i += NT) {
        const uint64_t q = ld_coh64(p.part + (int64_t)i * B1_PART);
        head_m[i] = __uint_as_float((uint32_t)q);
        head_l[i] = __uint_as_float((uint32_t)(q >> 32));
      }
      auto load_po = [&](int e0) {
        const int e = 4 * e0, h = e >> 6, d = e & 63;
#pragma unroll
        for (int s_ = 0; s_ < B1_SLICES; ++s_) po[s_] = ld_coh128(r_part, ((h * B1_SLICES + s_) * B1_PART + 4 + d) * 4);
      };
      if (tid < (D >> 2)) {
        load_po(tid);
      } else {
#pragma unroll
        for (int s_ = 0; s_ < B1_SLICES; ++s_) po[s_] = u32x4{0u, 0u, 0u, 0u};
      }
      __syncthreads();
      for (int e0 = tid; e0 < (D >> 2); e0 += NT) {  // four features of one head per thread and pass
        if (e0 != tid) load_po(e0);
        const int e = 4 * e0, h = e >> 6;
        float M = -INFINITY;
#pragma unroll
        for (int s_ = 0; s_ < B1_SLICES; ++s_) M = fmaxf(M, head_m[h * B1_SLICES + s_]);
        float Lt = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s_ = 0; s_ < B1_SLICES; ++s_) {
          const float f = __builtin_amdgcn_exp2f(head_m[h * B1_SLICES + s_] - M);  // empty slice: exp2(-inf) = 0
          Lt += head_l[h * B1_SLICES + s_] * f;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] += __uint_as_float(po[s_][k]) * f;
        }
        const float inv = Lt > 0.f ? 1.0f / Lt : 0.f;
        *(uint2*)(lds_row + e * 2) = uint2{pack_bf16x2(o[0] * inv, o[1] * inv), pack_bf16x2(o[2] * inv, o[3] * inv)};
      }
      __syncthreads();
      for (int q0 = 0; q0 < npw_c; q0 += PPP) {
        const int q = q0 + qw;
        float dp[2] = {0.f, 0.f};
        if (q < npw_c) {
          if (q0 > 0) {
            set_rows_c(q);
            if (kp > 0) gp.prime(prow, kp);
          }
          if (kp > 0) gp.run(prow, kp, actp, dp);
        }
        if (lane == 0) {
          proj_part[qw][half][0] = dp[0];
          proj_part[qw][half][1] = dp[1];
        }
        __syncthreads();
        if (tid < PPP && q0 + tid < npw_c) {
          const int pc = blockIdx.x + gridDim.x * (q0 + tid);
          const float dp0 = proj_part[tid][0][0] + proj_part[tid][1][0], dp1 = proj_part[tid][0][1] + proj_part[tid][1][1];
          const float df0 = fc2_part[q0 + tid][0][0] + fc2_part[q0 + tid][1][0];
          const float df1 = fc2_part[q0 + tid][0][1] + fc2_part[q0 + tid][1][1];
          const uint32_t xo = q0 == 0 ? xo0 : ld_coh32(xc + 2 * pc);
          const uint32_t bpw = q0 == 0 ? bpw0 : *(const uint32_t*)(w.bp + 2 * pc), b2w = q0 == 0 ? b2w0 : *(const uint32_t*)(w.b2 + 2 * pc);
          const uint32_t t1 = pack_bf16x2(dp0 + lo_bf(bpw), dp1 + hi_bf(bpw));
          const uint32_t x1 = pack_bf16x2(lo_bf(xo) + lo_bf(t1), hi_bf(xo) + hi_bf(t1));
          const uint32_t t2 = pack_bf16x2(df0 + lo_bf(b2w), df1 + hi_bf(b2w));
          st_coh32(xn + 2 * pc, pack_bf16x2(lo_bf(x1) + lo_bf(t2), hi_bf(x1) + hi_bf(t2)));
        }
        __syncthreads();
      }
    }
    stamp();
    const bool more = l + 1 < p.n_layers;
    if (more || p.lm_w) {
      bar.arrive();
      // the next phase's first weight rows go out while the barrier collects arrivals
      if (more) prime_first(npw_a, p.layer[l + 1].w1, p.ld1);
      else prime_first(npw_d, p.lm_w, p.ld_lm);
      bar.wait();
    }
    stamp();
    xc = xn;
  }

  // ================= phase D (whole decode step): final layer norm, lm_head rows, argmax ==============================
  if (p.lm_w) {
    layer_norm_to_lds(xc, p.post_ln_w, p.post_ln_b);
    float best = -INFINITY;
    int bi = 0x7fffffff;
    run_rows(npw_d, p.lm_w, p.ld_lm, [&](int pp, float d0, float d1) {
      const uint32_t bw = *(const uint32_t*)(p.lm_b + 2 * pp);
      const uint32_t o = pack_bf16x2(d0 + lo_bf(bw), d1 + hi_bf(bw));
      if (lane == 0) *(uint32_t*)(p.logits + 2 * pp) = o;
      // greedy choice on the bf16 logits, ties -> lowest index, the suppressed id never (argmax_kernel's rule)
      const float v0 = (2 * pp == p.suppress) ? -INFINITY : lo_bf(o), v1 = (2 * pp + 1 == p.suppress) ? -INFINITY : hi_bf(o);
      if (v0 > best || (v0 == best && 2 * pp < bi)) {
        best = v0;
        bi = 2 * pp;
      }
      if (v1 > best || (v1 == best && 2 * pp + 1 < bi)) {
        best = v1;
        bi = 2 * pp + 1;
      }
    });
    if (lane == 0) {
      red[wave] = best;
      best_i[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      for (int ww = 1; ww < NWV; ++ww)
        if (red[ww] > best || (red[ww] == best && best_i[ww] < bi)) {
          best = red[ww];
          bi = best_i[ww];
        }
      st_coh32(p.part + 2 * blockIdx.x, __float_as_uint(best));
      st_coh32(p.part + 2 * blockIdx.x + 1, (uint32_t)bi);
    }
    bar.arrive();
    bar.wait();
    if (blockIdx.x == 0) {
      best = -INFINITY;
      bi = 0x7fffffff;
      for (int i = tid; i < (int)gridDim.x; i += NT) {
        const float ov = __uint_as_float(ld_coh32(p.part + 2 * i));
        const int oi = (int)ld_coh32(p.part + 2 * i + 1);
        if (ov > best || (ov == best && oi < bi)) {
          best = ov;
          bi = oi;
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
      __syncthreads();
      if (lane == 0) {
        red[wave] = best;
        best_i[wave] = bi;
      }
      __syncthreads();
      if (tid == 0) {
        for (int ww = 1; ww < NWV; ++ww)
          if (red[ww] > best || (red[ww] == best && best_i[ww] < bi)) {
            best = red[ww];
            bi = best_i[ww];
          }
        p.next[0] = (bi >= p.vocab) ? 0 : bi;  // an all-NaN row selects nothing: id 0, as argmax_kernel does
        p.pos_out[0] = pos + 1;
      }
    }
  }
  // the next launch continues the barrier targets where this one stopped (every workgroup read the epoch at its start)
  if (blockIdx.x == 0 && tid == 0) st_coh32(p.sync, bar.base + bar.count);
}

}  // namespace

extern "C" size_t md_decode_b1_workspace_bytes(const md_text_model* m) {
  if (!m || !m->blocks) return 0;
  const size_t dim = m->dim, ffp = m->blocks[0].fc1.n_pad;
  return 2 * ((dim * 2 + 255) / 256 * 256) + ((3 * dim + ffp) * 2 + 255) / 256 * 256 +
         ((size_t)m->n_heads * B1_SLICES * B1_PART * 4 + 255) / 256 * 256;
}

namespace {

bool b1_supported(const md_text_model* m, const md_kv_cache* kv);

md_status b1_fill(const md_text_model* m, const md_kv_cache* kv, const int32_t* pos, void* workspace, void* sync_state, B1Args& a) {
  if (!b1_supported(m, kv)) return MD_ERR_UNSUPPORTED;  // callers ask md_decode_step_b1_supported() first
  const md_text_block& b0 = m->blocks[0];
  const int ff = b0.fc1.n;
  for (int l = 0; l < m->n_layers; ++l) {
    const md_text_block& b = m->blocks[l];
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
  a.x_first = nullptr;
  a.x_last = nullptr;
  a.token = nullptr;
  a.wte = a.post_ln_w = a.post_ln_b = a.lm_w = a.lm_b = nullptr;
  a.ld_lm = a.vocab = 0;
  a.suppress = -1;
  a.logits = nullptr;
  a.next = a.pos_out = nullptr;
  return MD_OK;
}

// Per device: the CU count and whether ONE workgroup per CU of decode_b1_kernel<8> can be resident at all (the software
// grid barriers need the whole grid on the chip; a plain launch of a grid the hardware cannot hold would spin to the
// bound and raise the error word).  A cooperative launch would only add this same check at +15-19 us per launch.
struct B1Device {
  int n_cu = 0;      // 0: not probed yet
  bool resident = false;
};
const B1Device& b1_device() {
  static B1Device cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  B1Device& d = cache[dev];
  if (d.n_cu == 0) {
    int n = 256, per_cu = 0;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    const bool ok = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_b1_kernel<8>, 512, 0) == hipSuccess;
    (void)hipGetLastError();
    d.resident = ok && per_cu >= 1;
    d.n_cu = n >= 8 ? n : 8;
  }
  return d;
}

// every static limit of the kernel for this model / cache / device, in one place (b1_fill and the public query)
bool b1_supported(const md_text_model* m, const md_kv_cache* kv) {
  if (!m || !kv || !m->blocks) return false;
  const B1Device& d = b1_device();
  const int n_cu = d.n_cu;
  if (!d.resident || n_cu > B1_MAX_WG) return false;
  if (!(m->n_layers >= 1 && m->n_layers <= B1_MAX_LAYERS && m->n_heads == m->n_kv_heads && m->dim == m->n_heads * 64)) return false;
  if (!(m->dim % 8 == 0 && m->dim <= B1_MAX_DIM && m->n_heads <= 64 && kv->ctx <= 2048)) return false;
  const md_text_block& b0 = m->blocks[0];
  if (!(b0.qkv_fc1.w && b0.qkv_fc1.b && b0.proj.b && b0.fc2.b)) return false;
  const int ff = b0.fc1.n;
  if (!(ff % 8 == 0 && ff <= B1_MAX_FF && (3 * m->dim + ff) % 2 == 0 && b0.qkv.n == 3 * m->dim)) return false;
  for (int l = 0; l < m->n_layers; ++l) {
    const md_text_block& b = m->blocks[l];
    if (!(b.qkv_fc1.w && b.qkv_fc1.b && b.proj.b && b.fc2.b && b.qkv_fc1.k_pad == b0.qkv_fc1.k_pad &&
          b.proj.k_pad == b0.proj.k_pad && b.fc2.k_pad == b0.fc2.k_pad))
      return false;
  }
  if ((m->dim / 2 + n_cu - 1) / n_cu > B1_MAX_Q) return false;          // fc2 partials of a workgroup's pairs live in LDS
  if (m->n_heads * B1_SLICES * B1_PART < 2 * n_cu) return false;        // the argmax candidates reuse the attention partials' buffer
  return true;
}

md_status b1_launch(const md_text_model* m, const B1Args& a, hipStream_t s) {
  // 8 waves per workgroup: two per SIMD, 256 registers each (16 waves at 128 registers spill the weight rings: 1.39 ms per
  // token against 0.84, profiles/r02_decode_b1_persistent_phase_times.txt)
  (void)m;
  hipLaunchKernelGGL(decode_b1_kernel<8>, dim3(b1_device().n_cu), dim3(512), 0, s, a);
  return md_launch_status();
}

}  // namespace

extern "C" int32_t md_decode_step_b1_supported(const md_text_model* m, const md_kv_cache* kv) {
  if (!m || !m->wte || !m->post_ln.w || !m->post_ln.b || !m->lm_head.w || !m->lm_head.b || m->vocab % 2 != 0 || m->lm_head.n != m->vocab) return 0;
  if (m->fp8) return 0;  // the persistent kernel streams the bf16 matrices only
  return b1_supported(m, kv) ? 1 : 0;
}

// sync_state: >= 16 KiB of device memory, ZERO when first used and never written by anything else; it carries the barrier
// counters from one launch to the next.  hidden: bf16 [dim].  x_in: bf16 [dim] (the token embedding; may equal hidden).
extern "C" md_status md_decode_b1_layers(const md_text_model* m, const void* x_in, void* hidden, const int32_t* pos,
                                         const md_kv_cache* kv, void* workspace, size_t workspace_bytes, void* sync_state,
                                         void* stream) {
  MD_CHECK_ARG(m && x_in && hidden && pos && kv && kv->k && kv->v && workspace && sync_state && m->blocks);
  if (workspace_bytes < md_decode_b1_workspace_bytes(m)) return MD_ERR_WORKSPACE;
  B1Args a;
  MD_TRY(b1_fill(m, kv, pos, workspace, sync_state, a));
  a.x_first = (const bf16_t*)x_in;
  // (x_in == hidden is allowed: with one block the in-place update would race, so only then go through the workspace)
  a.x_last = (m->n_layers > 1 || x_in != hidden) ? (bf16_t*)hidden : nullptr;
  hipStream_t s = (hipStream_t)stream;
  MD_TRY(b1_launch(m, a, s));
  if (!a.x_last && hipMemcpyAsync(hidden, a.x[m->n_layers & 1], (size_t)m->dim * 2, hipMemcpyDeviceToDevice, s) != hipSuccess)
    return MD_ERR_LAUNCH;
  return MD_OK;
}

// internal (api.hip): the whole greedy decode step of one sequence in the one launch
md_status md_decode_b1_step(const md_text_model* m, const int32_t* token, int32_t* next, int32_t* pos, const md_kv_cache* kv,
                            int32_t suppress_id, void* logits, void* workspace, size_t workspace_bytes, void* sync_state,
                            hipStream_t s) {
  MD_CHECK_ARG(m && token && next && pos && kv && kv->k && kv->v && logits && workspace && sync_state && m->blocks);
  MD_CHECK_ARG(m->wte && m->post_ln.w && m->post_ln.b && m->lm_head.w && m->lm_head.b && m->vocab % 2 == 0 && m->lm_head.n == m->vocab);
  if (workspace_bytes < md_decode_b1_workspace_bytes(m)) return MD_ERR_WORKSPACE;
  B1Args a;
  MD_TRY(b1_fill(m, kv, pos, workspace, sync_state, a));
  a.token = token;
  a.wte = (const bf16_t*)m->wte;
  a.post_ln_w = (const bf16_t*)m->post_ln.w;
  a.post_ln_b = (const bf16_t*)m->post_ln.b;
  a.lm_w = (const bf16_t*)m->lm_head.w;
  a.lm_b = (const bf16_t*)m->lm_head.b;
  a.ld_lm = m->lm_head.k_pad;
  a.vocab = m->vocab;
  a.suppress = suppress_id;
  a.logits = (bf16_t*)logits;
  a.next = next;
  a.pos_out = pos;
  return b1_launch(m, a, s);
}
