// FP8 tile GEMM (opt-in numerical mode, BASELINE configs[4] "CDNA4 fp8 MFMA"):
//   C = epilogue(a_scale * w_scale[n] * (A8 . W8^T) + b)          A8, W8: OCP e4m3fn, row-major, K contiguous
// for the MFMA-bound linears of the ViT, the projector and the decoder prefill (reference ops: layers.py:34-35,
// 129-146; vision.py:67; text.py:30,53 -- the reference itself has no fp8 path, so this mode is judged by tolerance
// against the bf16 path, never by bit parity).
//
// v_mfma_f32_32x32x64_f8f6f4 multiplies 64 K elements per instruction at twice the bf16 MFMA rate.  Its operand
// layout (tools/probes/mfma_f8_probe.hip): byte b of lane l of the first operand is X[l & 31][32 (l >> 5) + b], of
// the second Y[32 (l >> 5) + b][l & 31]; D in the bf16 32x32 map.  So a lane's operand is 32 CONSECUTIVE BYTES of
// a row-major fp8 row: a 64-byte row slice is exactly one K step, and a fragment is two ds_read_b128.
//
// Structure: the eight-wave alternating-wave-group kernel of gemm_bf16.hip (tile 15) with twice the K per slice:
//   * 256 x 256 tile, 8 waves as 2 (M) x 4 (N), 128 x 64 per wave = 4 x 2 accumulator blocks in VGPRs;
//   * K in 64-element slices (64-byte rows, 4 chunks of 16 B, chunk c of row r at physical chunk c ^ ((r >> 2) & 3)),
//     HBM/L2 -> LDS by LDS-DMA into a 4-deep ring, two slices in flight ahead of the one being multiplied;
//   * ONE phase per slice:  { 12 ds_read_b128 + 4 LDS-DMA pieces } barrier { 8 MFMAs (256 matrix-pipe cycles) } barrier,
//     the second wave group (one wave per SIMD, like the first) one barrier behind, so that on every SIMD one wave
//     owns the matrix pipe while the other fetches: the other group's MFMA phase hides the LDS latency;
//   * persistent workgroups (one per CU), the next tile's first two slices requested before the epilogue;
//   * epilogue: fp32 scale and bias, ONE rounding to bf16, transposition through a wave-private LDS tile, GELU /
//     residual on 16-byte row pieces; columns >= f8_from_col can be stored as fp8 (the next GEMM's operand) instead.
// At full matrix rate a CU would need 64 B / clk of operands through its vector-memory path (twice the bf16
// kernel's demand for the same tile): this kernel is bound by data movement, not by the matrix pipe.
#include "gemm_internal.hpp"

#include <algorithm>
#include <type_traits>

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NT = WM * WN * 64;
constexpr int ROW_BYTES = 64, CH = 4, STAGES = 4, AHEAD = STAGES - 2;
constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES, STAGE = A_BYTES + B_BYTES;
constexpr int NA = BM * CH / NT, NB = BN * CH / NT, PIECES = NA + NB;
constexpr int RING = STAGES * STAGE, EPI_LDS = WM * WN * 4096, LDS_BYTES = RING + EPI_LDS;
static_assert(NA == 2 && NB == 2 && MI == 4 && NI == 2 && LDS_BYTES <= 163840, "geometry");
static_assert((AHEAD - 1) * PIECES < 64, "vmcnt is a 6-bit counter");

struct F8K {
  const uint8_t* A;
  const uint8_t* W;
  const float* wscale;
  const bf16_t* bias;
  const bf16_t* R;
  bf16_t* C;
  uint8_t* C8;
  int64_t lda, ldw, ldc, ldc8, ldr;
  float a_scale, c8_inv_scale;
  int M, n_store, n_pad, K;
  int tiles_m, tiles_n, res_row_mod, group_m, gelu_from, f8_from;
  // MD_EPI_QKV_ROPE (the decoder's fused [q | k | v | fc1] layer at prefill): per-row (cos, sin) rows and slab byte offsets
  // (rope_rowinfo_kernel), the layer's bf16 K / V slabs and -- optionally -- their e4m3 copies with the layer's inverse scales
  const float* rope_cs;
  const uint32_t* rope_kv;
  bf16_t *kslab, *vslab;
  uint8_t *k8slab, *v8slab;
  float k8_inv, v8_inv;
  int rope_d, rope_ctx;
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
template <int OFF>
__device__ __forceinline__ void ds_read_b128(u32x4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
__device__ __forceinline__ void ds_write_b64_asm(uint32_t addr, u32x2 v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void ds_read_b128_plain(u32x4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}
__device__ __forceinline__ void ds_read_b64_plain(u32x2& dst, uint32_t addr) {
  asm volatile("ds_read_b64 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
__device__ __forceinline__ i32x8 join(const u32x4& lo, const u32x4& hi) {
  return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}
// 8 bf16 (as fp32 pairs) -> 8 e4m3fn bytes, saturating at +-448

template <int EPI>
__global__ __launch_bounds__(NT) void gemm_f8_kernel(const F8K p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int hi = lane >> 5, l31 = lane & 31;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int nwg = p.tiles_m * p.tiles_n;
  const int per_group = p.group_m * p.tiles_n;
  int m0, n0;
  const char* a_src[NA];
  const char* b_src[NB];
  auto set_tile = [&](int vv) {
    const int L = xcd_remap(vv, nwg);
    const int first_m = (L / per_group) * p.group_m;
    const int gsz = min(p.tiles_m - first_m, p.group_m);
    m0 = (first_m + (L % per_group) % gsz) * BM;
    n0 = ((L % per_group) / gsz) * BN;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int slot = j * NT + tid, r = slot >> 2, c = (slot & 3) ^ ((r >> 2) & 3);
      a_src[j] = (const char*)p.A + (int64_t)min(m0 + r, p.M - 1) * p.lda + c * 16;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int slot = j * NT + tid, r = slot >> 2, c = (slot & 3) ^ ((r >> 2) & 3);
      b_src[j] = (const char*)p.W + (int64_t)min(n0 + r, p.n_pad - 1) * p.ldw + c * 16;
    }
  };
  int vtile = blockIdx.x;
  if (vtile >= nwg) return;
  set_tile(vtile);

  // one 16-byte-per-lane LDS-DMA piece (1 KiB per wave) of the slice going into ring stage `stage`
  auto issue_piece = [&](auto piece_c, int stage) {
    constexpr int P = decltype(piece_c)::value;
    char* base = smem + stage * STAGE + wave * (64 * 16);
    if constexpr (P < NA) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_src[P],
                                       (__attribute__((address_space(3))) void*)(base + P * NT * 16), 16, 0, 0);
      a_src[P] += ROW_BYTES;
    } else {
      constexpr int Q = P - NA;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b_src[Q],
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + Q * NT * 16), 16, 0, 0);
      b_src[Q] += ROW_BYTES;
    }
  };

  // fragment read offsets: row = (multiple of 32) + l31, so the swizzle term depends on the lane only
  const uint32_t swz = (l31 >> 2) & 3;
  const uint32_t c_lo = ((2 * hi) ^ swz) * 16, c_hi = ((2 * hi + 1) ^ swz) * 16;  // this lane's 32 bytes of a 64-byte row
  const uint32_t a_row_off = (wm * TM + l31) * ROW_BYTES;
  const uint32_t b_row_off = A_BYTES + (wn * TN + l31) * ROW_BYTES;

  const int nk = p.K / 64;
  auto prologue = [&]() {
    static_for<0, AHEAD>([&](auto sc) {
      constexpr int SL0 = decltype(sc)::value;
      if (SL0 < nk) static_for<0, PIECES>([&](auto pc) { issue_piece(pc, SL0); });
    });
  };
  prologue();
  const int lag = (wave >= (WM * WN) / 2) ? 1 : 0;  // waves w and w + 4 share a SIMD: one of each group per SIMD

  f32x16 acc[MI][NI];
  for (;;) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk >= AHEAD) wait_vm<(AHEAD - 1) * PIECES>(); else wait_vm<0>();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();  // slice 0 visible to everybody
    if (lag) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int u = 0; u < nk; ++u) {
      const uint32_t st = lds_base + (u % STAGES) * STAGE;
      const int v = u + AHEAD;  // slice whose DMA is issued during slice u; its ring stage was last read in slice u - 2
      const bool has_next = v < nk;
      const int nstage = v % STAGES;
      u32x4 bl[NI], bh[NI], al[MI], ah[MI];
      static_for<0, NI>([&](auto j) {
        ds_read_b128<decltype(j)::value * 32 * ROW_BYTES>(bl[decltype(j)::value], st + b_row_off + c_lo);
        ds_read_b128<decltype(j)::value * 32 * ROW_BYTES>(bh[decltype(j)::value], st + b_row_off + c_hi);
      });
      static_for<0, MI>([&](auto i) {
        ds_read_b128<decltype(i)::value * 32 * ROW_BYTES>(al[decltype(i)::value], st + a_row_off + c_lo);
        ds_read_b128<decltype(i)::value * 32 * ROW_BYTES>(ah[decltype(i)::value], st + a_row_off + c_hi);
      });
      __builtin_amdgcn_sched_barrier(0);
      if (has_next) static_for<0, PIECES>([&](auto pc) { issue_piece(pc, nstage); });
      // this wave's pieces of slice u + 1 (first read in the NEXT phase) have landed; the younger slice stays in flight
      if (has_next) wait_vm<(AHEAD - 1) * PIECES>(); else wait_vm<0>();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      wait_lgkm<0>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      static_for<0, MI>([&](auto ic) {
        static_for<0, NI>([&](auto jc) {
          constexpr int I = decltype(ic)::value, J = decltype(jc)::value;
          // first operand = weight rows: a lane then holds one row m and runs of 4 consecutive columns n
          acc[I][J] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(join(bl[J], bh[J]), join(al[I], ah[I]), acc[I][J], 0, 0, 0, 0, 0, 0);
        });
      });
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (!lag) __builtin_amdgcn_s_barrier();  // the leading group owes the barrier the lagging one took first

    // ---- epilogue -----------------------------------------------------------------------------------
    // acc[i][j][r]: row m = 32 i + l31, col n = 32 j + 8 (r >> 2) + 4 hi + (r & 3)
    __syncthreads();  // every wave is done with the operand ring
    const int m0c = m0, n0c = n0;
    const uint32_t tile_lds = lds_base + RING + wave * 4096;  // wave-private 32 x 64 bf16 transposition tile
    const int wn0 = n0c + wn * TN;
    float sc_v[NI][4][4];
    u32x2 bias_p[NI][4];  // packed bf16: unpacked at use (32 registers fewer than fp32 copies, next to 128 accumulators)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = wn0 + 32 * j + 8 * g + 4 * hi;
        u32x2 bw = {0u, 0u};
        f32x4 sw = {0.f, 0.f, 0.f, 0.f};
        if (n < p.n_pad) {
          if (p.bias != nullptr) bw = *(const u32x2*)(p.bias + n);
          sw = *(const f32x4*)(p.wscale + n);
        }
        bias_p[j][g] = bw;
#pragma unroll
        for (int e = 0; e < 4; ++e) sc_v[j][g][e] = sw[e] * p.a_scale;
      }
    // residual operand: the four pieces of row block i are requested at the top of its pass and arrive while the block is
    // scaled and transposed (all MI x 4 pieces up front, or a double buffer, do not fit next to the 128 accumulators and
    // the scale / bias vectors: spills)
    u32x4 rres[4];
    auto load_residual = [&](int i, u32x4 (&rv)[4]) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = q * 64 + lane, row = idx >> 3, ch = idx & 7;
        const int m = m0c + wm * TM + 32 * i + row, n = wn0 + ch * 8;
        rv[q] = u32x4{0, 0, 0, 0};
        if (m < p.M && n < p.n_store) {
          const int64_t rrow = p.res_row_mod ? (m % p.res_row_mod) : m;
          rv[q] = *(const u32x4*)(p.R + rrow * p.ldr + n);
        }
      }
    };
    // the next tile's first slices are requested now (after this tile's bias / scale / residual loads) and land under
    // this epilogue; the transposition tiles live behind the ring
    vtile += gridDim.x;
    const bool more = vtile < nwg;
    if (more) {
      set_tile(vtile);
      prologue();
    }
    const bool to_f8 = (p.C8 != nullptr) && (wn0 >= p.f8_from);  // wave-uniform: f8_from is a multiple of 64
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if constexpr (EPI == MD_EPI_RESIDUAL) load_residual(i, rres);
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 w;
          w[0] = pack_bf16x2(__builtin_fmaf(acc[i][j][4 * g + 0], sc_v[j][g][0], lo_bf(bias_p[j][g][0])),
                             __builtin_fmaf(acc[i][j][4 * g + 1], sc_v[j][g][1], hi_bf(bias_p[j][g][0])));
          w[1] = pack_bf16x2(__builtin_fmaf(acc[i][j][4 * g + 2], sc_v[j][g][2], lo_bf(bias_p[j][g][1])),
                             __builtin_fmaf(acc[i][j][4 * g + 3], sc_v[j][g][3], hi_bf(bias_p[j][g][1])));
          const int ch = 4 * j + g;
          ds_write_b64_asm(tile_lds + l31 * 128 + ((ch ^ (l31 & 7)) * 16) + hi * 8, w);
        }
      if constexpr (EPI == MD_EPI_QKV_ROPE) {
        // q / k / v sections: a wave's 64 columns are ONE head.  Features 0..31 are rotated: the reference reads them
        // half-split (re = x[d], im = x[16 + d]) and writes them interleaved (rope.py:37-46), so output chunk ch < 4 (features
        // 8 ch .. 8 ch + 7 = pairs d = 4 ch .. 4 ch + 3) needs the two 8-byte halves x[4 ch ..] and x[16 + 4 ch ..] of the
        // transposed row -- two ds_read_b64 instead of one b128; chunks 4..7 and the v section pass through.  fp32 arithmetic on
        // the bf16-rounded layer output with separately rounded products (md_rope_pair), bit-equal to rope_kv_kernel.  q
        // stays in the activation, k / v go to the bf16 slab and, when the cache has one, to its e4m3 copy (text.py:45-46).
        const int sec = wn0 / p.rope_d;  // 0 q, 1 k, 2 v, >= 3: fc1 columns   (wave-uniform)
        if (sec < 3) {
          const int head = (wn0 - sec * p.rope_d) >> 6;
          u32x2 ta[4], tb[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int idx = q * 64 + lane, row = idx >> 3, ch = idx & 7;
            const bool rot = sec != 2 && ch < 4;
            const int ca = rot ? (ch >> 1) : ch, cb = rot ? 2 + (ch >> 1) : ch;
            const int ha = rot ? (ch & 1) : 0, hb = rot ? (ch & 1) : 1;
            ds_read_b64_plain(ta[q], tile_lds + row * 128 + ((ca ^ (row & 7)) * 16) + ha * 8);
            ds_read_b64_plain(tb[q], tile_lds + row * 128 + ((cb ^ (row & 7)) * 16) + hb * 8);
          }
          wait_lgkm<0>();
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int idx = q * 64 + lane, row = idx >> 3, ch = idx & 7;
            const int m = m0c + wm * TM + 32 * i + row;
            if (m >= p.M) continue;
            u32x4 v = {ta[q][0], ta[q][1], tb[q][0], tb[q][1]};
            if (sec != 2 && ch < 4) {
              const f32x4* cp = (const f32x4*)(p.rope_cs + (int64_t)m * 32 + 8 * ch);  // (cos, sin) of pairs 4 ch .. 4 ch + 3
              const f32x4 c0 = cp[0], c1 = cp[1];
              const float re[4] = {lo_bf(ta[q][0]), hi_bf(ta[q][0]), lo_bf(ta[q][1]), hi_bf(ta[q][1])};
              const float im[4] = {lo_bf(tb[q][0]), hi_bf(tb[q][0]), lo_bf(tb[q][1]), hi_bf(tb[q][1])};
              const float cs[4] = {c0[0], c0[2], c1[0], c1[2]}, sn[4] = {c0[1], c0[3], c1[1], c1[3]};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float o_re, o_im;
                md_rope_pair(re[e], im[e], cs[e], sn[e], o_re, o_im);
                v[e] = pack_bf16x2(o_re, o_im);
              }
            }
            if (sec == 0) {
              *(u32x4*)(p.C + (int64_t)m * p.ldc + wn0 + ch * 8) = v;
            } else {
              const uint32_t off = p.rope_kv[m] + (uint32_t)head * (uint32_t)p.rope_ctx * 128u + (uint32_t)ch * 16u;  // bytes in the bf16 slab
              *(u32x4*)((char*)(sec == 1 ? p.kslab : p.vslab) + off) = v;
              uint8_t* s8 = sec == 1 ? p.k8slab : p.v8slab;
              if (s8 != nullptr) *(u32x2*)(s8 + (off >> 1)) = quant8(v, sec == 1 ? p.k8_inv : p.v8_inv);
            }
          }
          continue;
        }
      }
      u32x4 tv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // same-wave LDS operations execute in order: the reads see the writes above
        const int idx = q * 64 + lane, row = idx >> 3, ch = idx & 7;
        ds_read_b128_plain(tv[q], tile_lds + row * 128 + ((ch ^ (row & 7)) * 16));
      }
      wait_lgkm<0>();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = q * 64 + lane, row = idx >> 3, ch = idx & 7;
        u32x4 v = tv[q];
        const int m = m0c + wm * TM + 32 * i + row;
        const int n = wn0 + ch * 8;
        if (m < p.M && n < p.n_store) {
          if constexpr (EPI == MD_EPI_GELU || EPI == MD_EPI_QKV_ROPE) {
            if (n >= p.gelu_from) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const md_f32x2 ge = gelu_tanh_f32x2(md_f32x2{lo_bf(v[e]), hi_bf(v[e])});
                v[e] = pack_bf16x2(ge[0], ge[1]);
              }
            }
          } else if constexpr (EPI == MD_EPI_RESIDUAL) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v[e] = pack_bf16x2(lo_bf(rres[q][e]) + lo_bf(v[e]), hi_bf(rres[q][e]) + hi_bf(v[e]));
          }
          if (to_f8) {
            const float s = p.c8_inv_scale;
            u32x2 o;
            o[0] = pack_fp8x4(lo_bf(v[0]) * s, hi_bf(v[0]) * s, lo_bf(v[1]) * s, hi_bf(v[1]) * s);
            o[1] = pack_fp8x4(lo_bf(v[2]) * s, hi_bf(v[2]) * s, lo_bf(v[3]) * s, hi_bf(v[3]) * s);
            *(u32x2*)(p.C8 + (int64_t)m * p.ldc8 + (n - p.f8_from)) = o;
          } else {
            *(u32x4*)(p.C + (int64_t)m * p.ldc + n) = v;
          }
        }
      }
    }
    if (!more) break;
    // the epilogue's stores share vmcnt with the DMA ring: drain both before the next tile's counted waits
    wait_vm<0>();
  }
}

template <int EPI>
md_status launch(const F8K& k, hipStream_t stream) {
  auto fn = gemm_f8_kernel<EPI>;
  MD_TRY(md_ensure_dynamic_lds((const void*)fn, LDS_BYTES));
  F8K kk = k;
  kk.tiles_m = (k.M + BM - 1) / BM;
  kk.tiles_n = (k.n_store + BN - 1) / BN;
  const int nwg = kk.tiles_m * kk.tiles_n;
  int dev = 0, n_cu = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
  n_cu = n_cu >= 8 ? (n_cu / 8) * 8 : 8;  // a multiple of 8 keeps (sequence number % 8) == XCD for the tile-order remap
  hipLaunchKernelGGL(fn, dim3(std::min(nwg, n_cu)), dim3(NT), LDS_BYTES, stream, kk);
  return md_launch_status();
}

}  // namespace

namespace {
md_status gemm_f8_dispatch(const md_gemm_f8_args* a, void* stream, const md_rope_fuse* rf, const md_rope_fuse_f8* rf8) {
  MD_CHECK_ARG(a && a->a && a->lin.w && a->lin.scale && a->m > 0 && a->lin.n > 0 && a->lin.k > 0);
  MD_CHECK_ARG(a->lin.k_pad % 64 == 0 && a->lin.k_pad >= a->lin.k && a->lin.n_pad % 64 == 0 && a->lin.n_pad >= a->lin.n);
  MD_CHECK_ARG(a->lda % 16 == 0 && a->lda >= a->lin.k_pad);
  MD_CHECK_ARG((((uintptr_t)a->a | (uintptr_t)a->lin.w | (uintptr_t)a->lin.scale) & 15) == 0);
  MD_CHECK_ARG(a->a_scale > 0.f);
  const int n_store = a->store_pad_cols ? a->lin.n_pad : a->lin.n;
  const bool all_f8 = a->c8 != nullptr && a->f8_from_col <= 0;
  MD_CHECK_ARG(all_f8 || (a->c != nullptr && a->ldc % 8 == 0 && ((uintptr_t)a->c & 15) == 0));
  if (a->c8 != nullptr) {
    MD_CHECK_ARG(a->f8_from_col >= 0 && a->f8_from_col % 64 == 0 && a->ldc8 % 8 == 0 && ((uintptr_t)a->c8 & 7) == 0);
    MD_CHECK_ARG(a->ldc8 >= n_store - a->f8_from_col && a->c8_inv_scale > 0.f);
  }
  if (a->c != nullptr && !all_f8) MD_CHECK_ARG(a->ldc >= (a->c8 ? std::min(n_store, a->f8_from_col) : n_store));
  if (a->lin.b) MD_CHECK_ARG(((uintptr_t)a->lin.b & 7) == 0);
  F8K k;
  k.A = (const uint8_t*)a->a;
  k.W = (const uint8_t*)a->lin.w;
  k.wscale = a->lin.scale;
  k.bias = (const bf16_t*)a->lin.b;
  k.R = (const bf16_t*)a->r;
  k.C = (bf16_t*)a->c;
  k.C8 = (uint8_t*)a->c8;
  k.lda = a->lda;
  k.ldw = a->lin.k_pad;
  k.ldc = a->ldc;
  k.ldc8 = a->ldc8;
  k.ldr = a->ldr;
  k.a_scale = a->a_scale;
  k.c8_inv_scale = a->c8 ? a->c8_inv_scale : 1.f;
  k.M = a->m;
  k.n_store = n_store;
  k.n_pad = a->lin.n_pad;
  k.K = a->lin.k_pad;
  k.tiles_m = k.tiles_n = 0;
  k.res_row_mod = a->res_row_mod;
  k.group_m = md_gemm_auto_group_m(n_store);
  k.gelu_from = a->gelu_from_col;
  k.f8_from = a->c8 ? a->f8_from_col : 0;
  k.rope_cs = nullptr; k.rope_kv = nullptr; k.kslab = k.vslab = nullptr; k.k8slab = k.v8slab = nullptr;
  k.k8_inv = k.v8_inv = 1.f; k.rope_d = 1 << 30; k.rope_ctx = 0;
  hipStream_t s = (hipStream_t)stream;
  if (rf != nullptr) {
    if (!md_gemm_knob_rope_fuse()) return MD_ERR_UNSUPPORTED;
    // RoPE + KV write in the epilogue: [q | k | v] sections of n_heads x 64 columns ending where the GELU (= fp8) columns start
    const int D = rf->n_heads * 64;
    if (a->epilogue != MD_EPI_GELU || a->gelu_from_col != 3 * D || a->c == nullptr || a->c8 == nullptr || a->f8_from_col != 3 * D ||
        rf->slab_bytes >= 0xfffff000ull)
      return MD_ERR_UNSUPPORTED;
    k.rope_cs = rf->row_cs; k.rope_kv = rf->row_kv;
    k.kslab = (bf16_t*)rf->kslab; k.vslab = (bf16_t*)rf->vslab;
    k.rope_d = D; k.rope_ctx = rf->ctx;
    if (rf8 != nullptr && rf8->k8slab != nullptr) {
      if (!(rf8->k_scale > 0.f && rf8->v_scale > 0.f)) return MD_ERR_INVALID_ARG;
      k.k8slab = (uint8_t*)rf8->k8slab; k.v8slab = (uint8_t*)rf8->v8slab;
      k.k8_inv = 1.0f / rf8->k_scale; k.v8_inv = 1.0f / rf8->v_scale;
    }
    return launch<MD_EPI_QKV_ROPE>(k, s);
  }
  switch (a->epilogue) {
    case MD_EPI_BIAS: return launch<MD_EPI_BIAS>(k, s);
    case MD_EPI_GELU: return launch<MD_EPI_GELU>(k, s);
    case MD_EPI_RESIDUAL:
      MD_CHECK_ARG(a->r && a->ldr % 8 == 0 && ((uintptr_t)a->r & 15) == 0 && a->c8 == nullptr);
      return launch<MD_EPI_RESIDUAL>(k, s);
    default: return MD_ERR_INVALID_ARG;
  }
}
}  // namespace

extern "C" md_status md_gemm_f8(const md_gemm_f8_args* a, void* stream) { return gemm_f8_dispatch(a, stream, nullptr, nullptr); }

md_status md_gemm_f8_qkv_rope(const md_gemm_f8_args* a, const md_rope_fuse* rf, const md_rope_fuse_f8* rf8, hipStream_t stream) {
  if (rf == nullptr) return MD_ERR_INVALID_ARG;
  return gemm_f8_dispatch(a, (void*)stream, rf, rf8);
}
