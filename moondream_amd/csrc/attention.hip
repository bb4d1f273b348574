// Softmax attention kernels.
//
//  attn_prefill_kernel<HD>  tiled (flash-style) attention for a block of query
//     rows on the matrix cores: the ViT encoder (729 x 729, head_dim 72, no
//     mask; reference layers.py:163) and the decoder prefill against the KV
//     slab with the prefix-LM visibility rule (reference text.py:48-50, mask of
//     moondream.py:138-146).
//  attn_decode_kernel       one query per (sequence, head) over keys [0, kv_len)
//     of the KV slab: the decode step (reference moondream.py:472-474 +
//     text.py:48-50); HBM-bandwidth-bound, no matrix cores.
//
// Layout of the prefill kernel (wave64, v_mfma_f32_32x32x16_bf16):
//   workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32
//   rows.  Keys/values stream through LDS in tiles of 64.
//   S^T = K Q^T is computed with K as the first MFMA operand, so a lane holds ONE
//   query row (lane & 31) and 16 of every 32 keys: row max / sum are in-register
//   plus a single lane^32 exchange, and the per-row rescale factors are per-lane
//   scalars for both S and the output accumulator.
//   O^T = V^T P^T: P goes straight from the S accumulator registers into the B
//   operand (registers 8u..8u+7 of a 32-key block become one bf16x8); the
//   K-slot -> key permutation this implies (slot (hi, j) <-> key 16u + 4hi +
//   (j&3) + 8(j>>2)) is applied to the V^T operand's LDS addresses instead of
//   shuffling P between lanes.
//   head_dim 72 is handled by LDS-side zero padding only: 80 for the QK^T
//   contraction (5 K-steps), 96 (3 x 32 rows of V^T) for PV; HBM layouts stay
//   dense.
#include "md_common.hpp"
#include <cstdlib>

// md_gemm_set_tuning("attn_skip_dead", 0..3) / MD_ATTN_SKIP_DEAD: A/B and test hook (every setting gives the same bits)
static int g_attn_skip_dead = [] { const char* e = getenv("MD_ATTN_SKIP_DEAD"); return (e && *e) ? atoi(e) & 3 : 3; }();
void md_attention_set_skip_dead(int v) { g_attn_skip_dead = v & 3; }

namespace {

struct AttnK {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  bf16_t* o;
  int64_t q_bs, q_ts, q_hs, k_bs, k_ts, k_hs, v_bs, v_ts, v_hs, o_bs, o_ts, o_hs;
  int q_len, kv_len_all, prefix, kv_group;  // kv_group = n_heads / n_kv_heads
  const int32_t* q_pos0;
  const int32_t* kv_len;
  float scale_log2;  // scale * log2(e)
  int n_qblk, n_bh, n_heads;  // XCD-aware 1-D grid of the LDS-DMA kernels
  uint8_t* o8;                // opt-in FP8 mode: the output row as e4m3 bytes as well (md_attn_args.o8), or nullptr
  int64_t o8_bs, o8_ts;
  float o8_inv_scale;
  int head_dim;
  int skip_dead;  // LDS-DMA prefill kernel, exact work skipping: bit 0 = the second 32-key half of a last tile with no live key in it, bit 1 = waves with no live query row
};

template <int HD>
struct Cfg {
  static constexpr int HDP = (HD + 15) / 16 * 16;  // QK^T contraction length (zero padded)
  static constexpr int KSTEPS = HDP / 16;
  static constexpr int ND = (HD + 31) / 32;        // 32-row blocks of V^T / O^T
  static constexpr int KSTR = HDP + 8;             // K tile row stride (elements): 16-B slots rotate by 11 (HD 72) / 9 (HD 64) per row -> conflict-free b128
  static constexpr int VSTR = 68;                  // V^T row stride: 8-B slots rotate by 17 per row
  static constexpr int CPR = HD / 8;               // 16-byte chunks per global row
  static constexpr int K_BYTES = 64 * KSTR * 2;
  static constexpr int V_BYTES = ND * 32 * VSTR * 2;
  static constexpr int O_BYTES = 4 * 32 * KSTR * 2;
  static constexpr int LDS = (K_BYTES + V_BYTES) > O_BYTES ? (K_BYTES + V_BYTES) : O_BYTES;
};

template <int HD>
__global__ __launch_bounds__(256) void attn_prefill_kernel(const AttnK p) {
  using C = Cfg<HD>;
  __shared__ __attribute__((aligned(16))) char smem[C::LDS];
  char* Ks = smem;
  char* Vt = smem + C::K_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z, h = blockIdx.y, hk = h / p.kv_group;
  const int q_pos0 = p.q_pos0 ? p.q_pos0[b] : 0;
  const int kv_len = p.kv_len ? p.kv_len[b] : p.kv_len_all;

  const int q_blk0 = blockIdx.x * 128;
  const int q_row0 = q_blk0 + wave * 32;
  const int q_row = min(q_row0 + l31, p.q_len - 1);  // clamp: padding rows replay a valid row
  const int qpos = q_pos0 + q_row;

  // keys any row of this workgroup may see
  const int blk_qpos_hi = q_pos0 + min(q_blk0 + 127, p.q_len - 1);
  const int blk_vis = (blk_qpos_hi < p.prefix) ? max(blk_qpos_hi + 1, p.prefix) : blk_qpos_hi + 1;
  const int kv_end = min(kv_len, blk_vis);
  // per-wave bounds for skipping the mask arithmetic on fully visible tiles
  const int w_qpos_lo = q_pos0 + min(q_row0, p.q_len - 1);
  const int w_qpos_hi = q_pos0 + min(q_row0 + 31, p.q_len - 1);

  // ---- Q fragments (B operand: column = query row, K-slot = 8 hi + j) ------
  bf16x8 qf[C::KSTEPS];
  {
    const bf16_t* qrow = p.q + (int64_t)b * p.q_bs + (int64_t)q_row * p.q_ts + (int64_t)h * p.q_hs;
#pragma unroll
    for (int s = 0; s < C::KSTEPS; ++s) {
      const int col = 16 * s + 8 * hi;
      bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      qf[s] = (col < HD) ? *(const bf16x8*)(qrow + col) : z;
    }
  }

  // zero the LDS padding that tile loads never touch
  if (C::HDP > HD) {
    for (int r = tid; r < 64; r += 256) *(u32x4*)(Ks + r * C::KSTR * 2 + HD * 2) = u32x4{0, 0, 0, 0};
  }
  for (int i = tid; i < (C::ND * 32 - HD) * (C::VSTR / 2); i += 256)
    *(uint32_t*)(Vt + HD * C::VSTR * 2 + i * 4) = 0u;

  f32x16 oacc[C::ND];
#pragma unroll
  for (int d = 0; d < C::ND; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const bf16_t* kbase = p.k + (int64_t)b * p.k_bs + (int64_t)hk * p.k_hs;
  const bf16_t* vbase = p.v + (int64_t)b * p.v_bs + (int64_t)hk * p.v_hs;

  // ---- K/V staging, split (cdna guide T14): the global loads of tile t+1 are issued
  // before tile t is computed and written to LDS after it, so their latency hides
  // under the MFMA/softmax work.  A thread owns (key-row PAIR, 16-byte chunk) items:
  // K goes to LDS row-major, V transposed -- two adjacent keys of one feature pack
  // into a single 4-byte LDS store.
  constexpr int ITEMS = 32 * C::CPR, NIT = (ITEMS + 255) / 256;
  u32x4 kreg[NIT][2], vreg[NIT][2];
  auto stage_load = [&](int kv0) {
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int it = min(tid + 256 * u, ITEMS - 1);  // surplus threads replay the last item (never written)
      const int rp = it / C::CPR, ch = it % C::CPR;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        // rows past kv_len replay the last valid row: their scores are masked to -inf
        // below, so P is exactly 0 and the (finite) V values never contribute
        const int j = min(kv0 + 2 * rp + e, kv_len - 1);
        kreg[u][e] = *(const u32x4*)(kbase + (int64_t)j * p.k_ts + ch * 8);
        vreg[u][e] = *(const u32x4*)(vbase + (int64_t)j * p.v_ts + ch * 8);
      }
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int it = tid + 256 * u;
      if (it < ITEMS) {
        const int rp = it / C::CPR, ch = it % C::CPR;
        *(u32x4*)(Ks + (2 * rp) * C::KSTR * 2 + ch * 16) = kreg[u][0];
        *(u32x4*)(Ks + (2 * rp + 1) * C::KSTR * 2 + ch * 16) = kreg[u][1];
        char* vt = Vt + (ch * 8) * C::VSTR * 2 + rp * 4;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const uint32_t a = vreg[u][0][w], bq = vreg[u][1][w];
          *(uint32_t*)(vt + (2 * w) * C::VSTR * 2) = (a & 0xffffu) | (bq << 16);
          *(uint32_t*)(vt + (2 * w + 1) * C::VSTR * 2) = (a >> 16) | (bq & 0xffff0000u);
        }
      }
    }
  };

  if (kv_end > 0) {
    stage_load(0);
    stage_write();
  }
  for (int kv0 = 0; kv0 < kv_end; kv0 += 64) {
    __syncthreads();  // tile kv0 is in LDS
    const bool more = kv0 + 64 < kv_end;
    if (more) stage_load(kv0 + 64);

    // ---- S^T = K Q^T : two 32-key blocks ------------------------------------
    f32x16 sacc[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[sub][r] = 0.f;
#pragma unroll
      for (int s = 0; s < C::KSTEPS; ++s) {
        const bf16x8 kf = *(const bf16x8*)(Ks + (32 * sub + l31) * C::KSTR * 2 + (16 * s + 8 * hi) * 2);
        sacc[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sacc[sub], 0, 0, 0);
      }
    }

    // ---- mask, online softmax (per lane = per query row) --------------------
    // Scores stay raw; the softmax scale is folded into the exp2 argument as one FMA:
    // p = 2^(s*c - m*c), c = scale*log2(e) > 0 (so the row max can be taken on raw s).
    const bool full_vis = (kv0 + 63 <= w_qpos_lo) || (w_qpos_hi < p.prefix && kv0 + 64 <= p.prefix);
    const bool need_mask = !(full_vis && kv0 + 64 <= kv_len);
    float mx = -INFINITY;
    if (need_mask) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = kv0 + 32 * sub + 8 * (r >> 2) + 4 * hi + (r & 3);
          const bool ok = (j < kv_len) && (j <= qpos || (qpos < p.prefix && j < p.prefix));
          sacc[sub][r] = ok ? sacc[sub][r] : -INFINITY;
        }
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[sub][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // Deferred rescale: the running reference m_run only moves when some row's max
    // has grown by more than 2^RESCALE_THR in the exp2 domain (wave-uniform branch).
    // Until then P = 2^((s - m_run) c) <= 2^RESCALE_THR, which bf16 P and the fp32
    // accumulators hold without loss; O, l and P always share one reference, so the
    // result is the same softmax.  Saves the 48-register accumulator rescale (and
    // its AGPR round trip) on almost every tile.
    constexpr float RESCALE_THR = 6.0f;
    if (__any((mx - m_run) * p.scale_log2 > RESCALE_THR)) {
      const float m_new = fmaxf(m_run, mx);  // finite from the first tile on: key 0 is visible to every query
      // (a row with no visible key yet keeps alpha = 1: it has nothing accumulated)
      const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < C::ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
    }
    const float mc = (m_run == -INFINITY) ? 0.f : -m_run * p.scale_log2;
    float psum = 0.f;
    bf16x8 pf[2][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p0 = __builtin_amdgcn_exp2f(fmaf(sacc[sub][8 * u + 2 * e], p.scale_log2, mc));
          const float p1 = __builtin_amdgcn_exp2f(fmaf(sacc[sub][8 * u + 2 * e + 1], p.scale_log2, mc));
          psum += p0 + p1;
          w[e] = pack_bf16x2(p0, p1);
        }
        pf[sub][u] = __builtin_bit_cast(bf16x8, w);
      }
    l_run += psum;

    // ---- O^T += V^T P^T ------------------------------------------------------
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int kcol = 32 * sub + 16 * u + 4 * hi;
#pragma unroll
        for (int d = 0; d < C::ND; ++d) {
          const char* vrow = Vt + (32 * d + l31) * C::VSTR * 2 + kcol * 2;
          const u32x2 lo = *(const u32x2*)(vrow);
          const u32x2 hi8 = *(const u32x2*)(vrow + 16);
          const u32x4 vv = {lo[0], lo[1], hi8[0], hi8[1]};
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf[sub][u],
                                                            oacc[d], 0, 0, 0);
        }
      }
    if (more) {
      __syncthreads();  // every wave is done reading tile kv0
      stage_write();
    }
  }

  // ---- finalize: O / l, transpose through LDS, coalesced row stores ---------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  __syncthreads();
  char* ot = smem + wave * 32 * C::KSTR * 2;
#pragma unroll
  for (int d = 0; d < C::ND; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = 32 * d + 8 * g + 4 * hi;
      if (col < HD) {
        u32x2 w;
        w[0] = pack_bf16x2(oacc[d][4 * g + 0] * inv, oacc[d][4 * g + 1] * inv);
        w[1] = pack_bf16x2(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv);
        *(u32x2*)(ot + l31 * C::KSTR * 2 + col * 2) = w;
      }
    }
  bf16_t* obase = p.o + (int64_t)b * p.o_bs + (int64_t)h * p.o_hs;
  uint8_t* o8base = p.o8 + (int64_t)b * p.o8_bs + (int64_t)h * HD;
  for (int idx = lane; idx < 32 * C::CPR; idx += 64) {
    const int row = idx / C::CPR, ch = idx % C::CPR;
    if (q_row0 + row < p.q_len) {
      const u32x4 v = *(const u32x4*)(ot + row * C::KSTR * 2 + ch * 16);
      if (p.o != nullptr) *(u32x4*)(obase + (int64_t)(q_row0 + row) * p.o_ts + ch * 8) = v;
      if (p.o8 != nullptr) *(u32x2*)(o8base + (int64_t)(q_row0 + row) * p.o8_ts + ch * 8) = quant8(v, p.o8_inv_scale);
    }
  }
}

// ---------------------------------------------------------------------------
// prefill, LDS-DMA staged (default): same math and fragment layouts as
// attn_prefill_kernel above, different plumbing --
//   * K and V tiles go global -> LDS by LDS-DMA (no staging registers, no VALU), both
//     ROW-major, double buffered: one barrier per 64-key tile, the DMA of tile t+1
//     runs under the MFMA/softmax work of tile t;
//   * the V^T operand of O^T += V^T P^T is gathered with ds_read_b64_tr_b16 (the
//     hardware 4x4 transpose read: in every 16-lane group, lane i points at 4
//     consecutive features of key (i >> 2) and receives 4 consecutive keys of feature
//     (i & 15)), so V is never transposed by ALU and never written with 4-byte stores.
//   LDS image of a tile: K rows of 176 B (head_dim 72: 9 data + 2 pad chunks) / 144 B
//   (head_dim 64) -- 16-byte slots rotate by 11 / 9 per row, b128 fragment reads are
//   conflict-free; V rows of 192 B (12 chunks) -- four consecutive keys x 64 B cover
//   the 256-byte bank row exactly once for the transpose read.  Pad chunks replay chunk
//   0 of their row (finite data): K pads only ever meet the zero pad of the Q
//   fragments, V pads only feed output rows >= head_dim, which are never stored.
// ---------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int HD>
struct CfgD {
  static constexpr int HDP = (HD + 15) / 16 * 16;
  static constexpr int KSTEPS = HDP / 16;
  static constexpr int ND = (HD + 31) / 32;
  static constexpr int CPR = HD / 8;                     // data chunks per global row
  static constexpr int KCH = CPR + (HD == 72 ? 2 : 1);   // 11 / 9 chunks per LDS row of K
  static constexpr int KROW = KCH * 16;
  static constexpr int VCH = 12, VROW = VCH * 16;
  static constexpr int K_BYTES = 64 * KROW;
  static constexpr int V_BYTES = 64 * VROW;
  static constexpr int BUF = K_BYTES + V_BYTES;
  static constexpr int K_LAST_WAVES = (64 * KCH - 512) / 64;  // waves that own a third K piece
  static constexpr int OSTR = HDP + 8;
  static constexpr int O_BYTES = 4 * 32 * OSTR * 2;
  static constexpr int LDS = (2 * BUF) > O_BYTES ? (2 * BUF) : O_BYTES;
  static_assert((64 * KCH - 512) % 64 == 0 && K_LAST_WAVES >= 1 && K_LAST_WAVES <= 4, "K piece split");
};

// PIPE: the K stream runs one tile ahead of the V stream and S(t+1) = K(t+1) Q^T is issued
// in the same basic block as the exp2 / pack work of tile t and the P V MFMAs of tile t, so
// one wave keeps the matrix pipe and the VALU busy at the same time (22 MFMAs x 32 cycles
// against ~110 VALU instructions per tile) instead of alternating between them.
template <int HD, bool PIPE>
__global__ __launch_bounds__(256) void attn_prefill_dma_kernel(const AttnK p) {
  using C = CfgD<HD>;
  __shared__ __attribute__((aligned(16))) char smem[C::LDS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: all query
  // blocks of one (batch, head) are mapped to ONE XCD (consecutive slots of it), so its K/V
  // rows come over the fabric once instead of once per query block.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / p.n_qblk) * 8 + xcd;
  if (bh >= p.n_bh) return;  // uniform per workgroup, before any barrier
  const int b = bh / p.n_heads, h = bh % p.n_heads, hk = h / p.kv_group;
  const int q_pos0 = p.q_pos0 ? p.q_pos0[b] : 0;
  const int kv_len = p.kv_len ? p.kv_len[b] : p.kv_len_all;

  const int q_blk0 = (slot % p.n_qblk) * 128;
  const int q_row0 = q_blk0 + wave * 32;
  const int q_row = min(q_row0 + l31, p.q_len - 1);
  const int qpos = q_pos0 + q_row;
  const int blk_qpos_hi = q_pos0 + min(q_blk0 + 127, p.q_len - 1);
  const int blk_vis = (blk_qpos_hi < p.prefix) ? max(blk_qpos_hi + 1, p.prefix) : blk_qpos_hi + 1;
  const int kv_end = min(kv_len, blk_vis);
  const int w_qpos_lo = q_pos0 + min(q_row0, p.q_len - 1);
  const int w_qpos_hi = q_pos0 + min(q_row0 + 31, p.q_len - 1);

  bf16x8 qf[C::KSTEPS];
  {
    const bf16_t* qrow = p.q + (int64_t)b * p.q_bs + (int64_t)q_row * p.q_ts + (int64_t)h * p.q_hs;
#pragma unroll
    for (int s = 0; s < C::KSTEPS; ++s) {
      const int col = 16 * s + 8 * hi;
      bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      qf[s] = (col < HD) ? *(const bf16x8*)(qrow + col) : z;
    }
  }

  f32x16 oacc[C::ND];
#pragma unroll
  for (int d = 0; d < C::ND; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const bf16_t* kbase = p.k + (int64_t)b * p.k_bs + (int64_t)hk * p.k_hs;
  const bf16_t* vbase = p.v + (int64_t)b * p.v_bs + (int64_t)hk * p.v_hs;

  // LDS-DMA pieces: chunk c = 256 j + tid of the tile image, lane-linear in LDS.  The source
  // pointers run ahead one tile per issue; only a tile that reaches past kv_len (the last one)
  // takes the clamped path.
  int krow[3], kcol[3], vrow[3], vcol[3];
  const bf16_t* ksrc[3];
  const bf16_t* vsrc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int c = 256 * j + tid;
    const int kc = c % C::KCH, vc = c % C::VCH;
    krow[j] = min(c / C::KCH, 63);
    kcol[j] = (kc < C::CPR ? kc : 0) * 8;
    vrow[j] = c / C::VCH;
    vcol[j] = (vc < C::CPR ? vc : 0) * 8;
    ksrc[j] = kbase + (int64_t)krow[j] * p.k_ts + kcol[j];
    vsrc[j] = vbase + (int64_t)vrow[j] * p.v_ts + vcol[j];
  }
  const int64_t k_step = 64 * p.k_ts, v_step = 64 * p.v_ts;
  // LDS-DMA issued through inline asm: with the builtin, hipcc orders every later LDS read it can
  // see (the transpose reads of the P V stage) behind the in-flight DMA of the NEXT tile with an
  // s_waitcnt vmcnt(0) in the middle of the iteration.  The buffers are disjoint by construction
  // (double buffering + the barrier below), so the wait belongs in front of that barrier only.
  auto dma16 = [&](const bf16_t* src, char* dst) {
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)dst;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :
                 : "v"(src), "s"(lds)
                 : "memory", "m0");
  };
  auto tile_barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own LDS-DMA pieces (the compiler does not count them)
    __syncthreads();
  };
  auto issue_k = [&](int kv0, int buf) {
    char* kb = smem + buf * C::BUF + 64 * wave * 16;
    if (kv0 + 64 <= kv_len) {  // wave-uniform: every row of the tile exists
      dma16(ksrc[0], kb);
      dma16(ksrc[1], kb + 4096);
      if (wave < C::K_LAST_WAVES) dma16(ksrc[2], kb + 8192);
    } else {
      // rows past kv_len replay the last valid row: their scores are masked to -inf, P is exactly 0
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (j < 2 || wave < C::K_LAST_WAVES)
          dma16(kbase + (int64_t)min(kv0 + krow[j], kv_len - 1) * p.k_ts + kcol[j], kb + 4096 * j);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) ksrc[j] += k_step;
  };
  auto issue_v = [&](int kv0, int buf) {
    char* vb = smem + buf * C::BUF + C::K_BYTES + 64 * wave * 16;
    if (kv0 + 64 <= kv_len) {
#pragma unroll
      for (int j = 0; j < 3; ++j) dma16(vsrc[j], vb + 4096 * j);
    } else {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        dma16(vbase + (int64_t)min(kv0 + vrow[j], kv_len - 1) * p.v_ts + vcol[j], vb + 4096 * j);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) vsrc[j] += v_step;
  };

  // per-lane offsets of the fragment reads
  const int g16 = lane >> 4, i16 = lane & 15;
  const int k_lane = l31 * C::KROW + hi * 16;
  const int v_lane = (4 * hi + (i16 >> 2)) * C::VROW + (16 * (g16 & 1) + 4 * (i16 & 3)) * 2;

  // both_subs == false (the LAST key tile of the block when at most 32 of its keys are live): the second 32-key sub-block holds only keys
  // >= kv_end (past kv_len, or past what the block's last row may see) -- masked to -inf, P exactly 0, a contribution of exact zeros to every sum -- so its S block is simply -inf and its
  // exponentials, packs and P V products are skipped; the result is bit for bit the one of the full computation.
  auto compute_s = [&](f32x16 (&sa)[2], const char* Ks, bool both_subs) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (sub == 1 && !both_subs) {  // wave-uniform
#pragma unroll
        for (int r = 0; r < 16; ++r) sa[1][r] = -INFINITY;
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[sub][r] = 0.f;
#pragma unroll
      for (int s = 0; s < C::KSTEPS; ++s) {
        const bf16x8 kf = *(const bf16x8*)(Ks + k_lane + 32 * sub * C::KROW + 32 * s);
        sa[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sa[sub], 0, 0, 0);
      }
    }
  };

  f32x16 sacc[2];
  // Pin the Q fragments before the loop: otherwise hipcc keeps its wait for those loads inside the
  // loop (s_waitcnt vmcnt(0) in front of the first MFMAs), where it also waits for the LDS-DMA
  // pieces it does not know about, every iteration.
#pragma unroll
  for (int s = 0; s < C::KSTEPS; ++s) asm volatile("" ::"v"(qf[s]));
  if (kv_end > 0) {
    issue_k(0, 0);
    issue_v(0, 0);
    if (PIPE) {
      if (64 < kv_end) issue_k(64, 1);
      tile_barrier();
      compute_s(sacc, smem, true);
    }
  }
  const bool wave_live = !(p.skip_dead & 2) || q_row0 < p.q_len;  // wave-uniform
  int buf = 0;
  for (int kv0 = 0; kv0 < kv_end; kv0 += 64, buf ^= 1) {
    // own DMA of the previous iteration landed (vmcnt 0), everybody's is visible, and the
    // buffers refilled below are no longer read by any wave
    tile_barrier();
    if (PIPE) {
      if (kv0 + 128 < kv_end) issue_k(kv0 + 128, buf);      // K(t+2) over K(t), consumed one iteration ago
      if (kv0 + 64 < kv_end) issue_v(kv0 + 64, buf ^ 1);    // V(t+1) over V(t-1)
    } else if (kv0 + 64 < kv_end) {
      issue_k(kv0 + 64, buf ^ 1);
      issue_v(kv0 + 64, buf ^ 1);
    }
    const char* Ks = smem + buf * C::BUF;
    const char* Vs = Ks + C::K_BYTES;
    const char* Kn = smem + (buf ^ 1) * C::BUF;  // PIPE: K(t+1) (stale but finite data after the last tile; result unused)

    // (PIPE keeps the full computation: its S runs one tile ahead)
    const bool both = PIPE || !(p.skip_dead & 1) || kv0 + 32 < kv_end;  // (kv_end <= kv_len: keys past it are visible to no row of this block)
    // A wave whose 32 query rows all lie past q_len (the fourth wave of the last query block: 729 / 735 rows = 5 blocks of 128 +
    // 89 / 95 rows) stores nothing: it keeps its share of the LDS-DMA stream and the barriers and skips the arithmetic, which
    // leaves its SIMD to the other workgroups' waves.
    if (!PIPE && !wave_live) continue;
    if (!PIPE) compute_s(sacc, Ks, both);

    const bool full_vis = (kv0 + 63 <= w_qpos_lo) || (w_qpos_hi < p.prefix && kv0 + 64 <= p.prefix);
    const bool need_mask = !(full_vis && kv0 + 64 <= kv_len);
    float mx = -INFINITY;
    if (need_mask) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = kv0 + 32 * sub + 8 * (r >> 2) + 4 * hi + (r & 3);
          const bool ok = (j < kv_len) && (j <= qpos || (qpos < p.prefix && j < p.prefix));
          sacc[sub][r] = ok ? sacc[sub][r] : -INFINITY;
        }
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[sub][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    constexpr float RESCALE_THR = 6.0f;  // deferred rescale, see attn_prefill_kernel
    if (__any((mx - m_run) * p.scale_log2 > RESCALE_THR)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < C::ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
    }
    const float mc = (m_run == -INFINITY) ? 0.f : -m_run * p.scale_log2;
    float psum = 0.f;
    bf16x8 pf[2][2];
    f32x16 snext[2];
    if (PIPE) compute_s(snext, Kn, true);  // independent of everything below: the scheduler interleaves it
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (sub == 1 && !both) break;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p0 = __builtin_amdgcn_exp2f(fmaf(sacc[sub][8 * u + 2 * e], p.scale_log2, mc));
          const float p1 = __builtin_amdgcn_exp2f(fmaf(sacc[sub][8 * u + 2 * e + 1], p.scale_log2, mc));
          psum += p0 + p1;
          w[e] = pack_bf16x2(p0, p1);
        }
        pf[sub][u] = __builtin_bit_cast(bf16x8, w);
      }
    }
    l_run += psum;

    // O^T += V^T P^T: MFMA K-slot (hi, j) <-> key 16u + 4hi + (j & 3) + 8 (j >> 2), so the two
    // transpose reads of a fragment start at keys 16u + 4hi and 16u + 8 + 4hi
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (sub == 1 && !both) break;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int d = 0; d < C::ND; ++d) {
          const char* va = Vs + v_lane + (32 * sub + 16 * u) * C::VROW + 64 * d;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)va);
          const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va + 8 * C::VROW));
          const bf16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vv, pf[sub][u], oacc[d], 0, 0, 0);
        }
      }
    }
    if (PIPE) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) sacc[sub] = snext[sub];
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  __syncthreads();
  char* ot = smem + wave * 32 * C::OSTR * 2;
#pragma unroll
  for (int d = 0; d < C::ND; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = 32 * d + 8 * g + 4 * hi;
      if (col < HD) {
        u32x2 w;
        w[0] = pack_bf16x2(oacc[d][4 * g + 0] * inv, oacc[d][4 * g + 1] * inv);
        w[1] = pack_bf16x2(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv);
        *(u32x2*)(ot + l31 * C::OSTR * 2 + col * 2) = w;
      }
    }
  bf16_t* obase = p.o + (int64_t)b * p.o_bs + (int64_t)h * p.o_hs;
  uint8_t* o8base = p.o8 + (int64_t)b * p.o8_bs + (int64_t)h * HD;
  for (int idx = lane; idx < 32 * C::CPR; idx += 64) {
    const int row = idx / C::CPR, ch = idx % C::CPR;
    if (q_row0 + row < p.q_len) {
      const u32x4 v = *(const u32x4*)(ot + row * C::OSTR * 2 + ch * 16);
      if (p.o != nullptr) *(u32x4*)(obase + (int64_t)(q_row0 + row) * p.o_ts + ch * 8) = v;
      if (p.o8 != nullptr) *(u32x2*)(o8base + (int64_t)(q_row0 + row) * p.o8_ts + ch * 8) = quant8(v, p.o8_inv_scale);
    }
  }
}

// ---------------------------------------------------------------------------
// decode: one query row per (sequence, head)
// ---------------------------------------------------------------------------
// 8 lanes cover one 128-byte key/value row (8 x 16 B); a wave covers 8 rows per
// load instruction.  Keys are partitioned into 128 classes (key index mod 128); class
// 32 u + r is accumulated on its own (fp32, ascending key order), the four classes of a
// residue r are combined as (u0 + u1) + (u2 + u3) and the 32 residues summed in
// ascending order.  That order is the DEFINITION of the result, so the two launch
// shapes below produce the same bits:
//   NW = 4   (large batches): thread group (wave, g) owns residue 8 wave + g and its
//            four classes -- four independent load streams per thread;
//   NW = 16  (few sequences): 16 waves, thread group owns ONE class -- a quarter of the
//            dependent iterations, for the latency-bound single-sequence decode step.
// Pass 1 streams K (scores -> LDS, running max), pass 2 streams V.
constexpr int DEC_MAX_CTX = 2048;

// FUSED: q points at the un-rotated fused activation row (q | k | v heads); the kernel
// applies the partial RoPE to this head's q and k itself (rope.py:20-48), stores the
// rotated k and v at slot pos = kv_len - 1 of the slab (moondream.py:74-78) and treats
// that newest key from LDS -- one launch instead of rope_kv_kernel + attention, same
// arithmetic (bf16-rounded rotated values), MHA only.
// NT: the K / V rows are requested with non-temporal loads.  A large batch's step reads every live cache row exactly once
// and nothing of it again before ~10 GB of other traffic has passed, so keeping the rows in the caches only evicts what
// could be reused; tools/probes/hbm_read_probe.hip: a 2048-workgroup streaming read runs at 5.7 TB/s with plain loads (this
// kernel: 5.6) and at 6.3 TB/s with non-temporal ones.  MHA only (with grouped heads a row is read by several workgroups).
template <bool NT>
__device__ __forceinline__ u32x4 load_kv_row(const bf16_t* p) {
  if constexpr (NT) return __builtin_nontemporal_load((const u32x4*)p);
  else return *(const u32x4*)p;
}

template <bool FUSED, int NW, bool NT = false>
__global__ __launch_bounds__(NW * 64) void attn_decode_kernel(const bf16_t* __restrict__ q, int64_t ldq,
                                                              bf16_t* __restrict__ o, int64_t ldo,
                                                              bf16_t* __restrict__ kslab,
                                                              bf16_t* __restrict__ vslab,
                                                              int64_t slab_bs, int ctx, const int32_t* kv_len_p,
                                                              int n_heads, int kv_group, float scale_log2,
                                                              const float* __restrict__ freqs, int rot) {
  static_assert(NW == 4 || NW == 16, "waves per workgroup");
  constexpr int CPT = 16 / NW;            // classes per thread group
  constexpr int NU = (NW == 16) ? 4 : 1;  // class quarters that live in different waves
  constexpr int UNR = (NW == 16) ? 8 : 2; // 128-key rounds whose loads are issued together
  constexpr int VPRE = (NW == 16) ? 8 : 0; // NW = 16: the V rows of the first VPRE rounds are requested WITH the K rows (one memory round trip for a
                                           // context of up to 1024 keys instead of two: the single-sequence step is latency-bound)
  __shared__ float sc[DEC_MAX_CTX];
  __shared__ float red[NU][32][64 + 1];   // [class quarter][residue][feature | sum of p]
  __shared__ float red_m[NW];
  __shared__ __attribute__((aligned(16))) bf16_t newrow[3][64];  // FUSED: rotated q, rotated k, v of the new token

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 3, c = lane & 7;
  const int b = blockIdx.y, h = blockIdx.x, hk = h / kv_group;
  const int kv_len = kv_len_p[b];
  const int pos = kv_len - 1;
  bf16_t* kb = kslab + (int64_t)b * slab_bs + (int64_t)hk * ctx * 64;
  bf16_t* vb = vslab + (int64_t)b * slab_bs + (int64_t)hk * ctx * 64;
  // residue and first class of this thread group
  const int res = (NW == 16) ? ((wave & 3) * 8 + g) : (wave * 8 + g);
  const int u0 = (NW == 16) ? (wave >> 2) : 0;

  if constexpr (FUSED) {
    // row layout: [q heads | k heads | v heads], 64 features per head
    const bf16_t* row = q + (int64_t)b * ldq;
    const int half = rot >> 1;
    if (tid < 2 * half) {  // rotated pairs of q (tid < half) and k
      const int which = tid / half, j = tid % half;
      const bf16_t* hp = row + (which ? (n_heads + h) : h) * 64;
      const float re = bf2f(hp[j]), im = bf2f(hp[half + j]);
      const float cs = freqs[((int64_t)pos * half + j) * 2], sn = freqs[((int64_t)pos * half + j) * 2 + 1];
      // separately rounded mul, mul, sub / add, as torch evaluates them; interleaved output
      float o_re, o_im;
      md_rope_pair(re, im, cs, sn, o_re, o_im);
      newrow[which][2 * j] = f2bf(o_re);
      newrow[which][2 * j + 1] = f2bf(o_im);
    } else if (tid >= 64 && tid < 64 + 2 * (64 - rot)) {  // pass-through features of q and k
      const int t2 = tid - 64, which = t2 / (64 - rot), i = rot + t2 % (64 - rot);
      newrow[which][i] = row[(which ? (n_heads + h) : h) * 64 + i];
    } else if (tid >= 192 && tid < 256) {  // v
      const int i = tid - 192;
      newrow[2][i] = row[(2 * n_heads + h) * 64 + i];
    }
    __syncthreads();
    if (tid < 64) kb[(int64_t)pos * 64 + tid] = newrow[1][tid];
    else if (tid < 128) vb[(int64_t)pos * 64 + tid - 64] = newrow[2][tid - 64];
  }

  float qv[8];
  {
    const u32x4 qq = FUSED ? *(const u32x4*)(&newrow[0][c * 8]) : *(const u32x4*)(q + (int64_t)b * ldq + h * 64 + c * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qv[2 * e] = lo_bf(qq[e]) * scale_log2;
      qv[2 * e + 1] = hi_bf(qq[e]) * scale_log2;
    }
  }

  // key of (round i, class slot u): 128 i + 32 (u0 + u) + res; rows past kv_len replay the last row, masked below
  // ---- pass 1: scores --------------------------------------------------------
  float mx = -INFINITY;
  u32x4 vpre[VPRE > 0 ? VPRE : 1];
  for (int i0 = 0; i0 * 128 < kv_len; i0 += UNR) {
    u32x4 kq[UNR][CPT];
#pragma unroll
    for (int r = 0; r < UNR; ++r)
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        const int j = 128 * (i0 + r) + 32 * (u0 + u) + res;
        kq[r][u] = load_kv_row<NT>(kb + (int64_t)min(j, pos) * 64 + c * 8);
      }
    if constexpr (VPRE > 0) {
      if (i0 == 0) {
#pragma unroll
        for (int r = 0; r < VPRE; ++r) {
          const int j = 128 * r + 32 * u0 + res;
          vpre[r] = load_kv_row<NT>(vb + (int64_t)min(j, pos) * 64 + c * 8);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < UNR; ++r)
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        const int j = 128 * (i0 + r) + 32 * (u0 + u) + res;
        // (FUSED: the newest key is not yet visible in global memory to this CU: take it from LDS)
        const u32x4 kk = (FUSED && j == pos) ? *(const u32x4*)(&newrow[1][c * 8]) : kq[r][u];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) s += qv[2 * e] * lo_bf(kk[e]) + qv[2 * e + 1] * hi_bf(kk[e]);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (j < kv_len) {
          if (c == 0) sc[j] = s;
          mx = fmaxf(mx, s);
        }
      }
  }
  mx = wave_max(mx);
  if (lane == 0) red_m[wave] = mx;
  __syncthreads();
  mx = red_m[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red_m[w]);

  // ---- pass 2: probabilities and P.V ----------------------------------------
  float acc[CPT][8], l[CPT];
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    l[u] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[u][e] = 0.f;
  }
  for (int i0 = 0; i0 * 128 < kv_len; i0 += UNR) {
    u32x4 vq[UNR][CPT];
    if (VPRE > 0 && i0 == 0) {
      static_assert(VPRE == 0 || (VPRE == UNR && CPT == 1), "prefetched V rows cover exactly the first batch");
#pragma unroll
      for (int r = 0; r < UNR; ++r) vq[r][0] = vpre[VPRE > 0 ? r : 0];
    } else {
#pragma unroll
      for (int r = 0; r < UNR; ++r)
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
          const int j = 128 * (i0 + r) + 32 * (u0 + u) + res;
          vq[r][u] = load_kv_row<NT>(vb + (int64_t)min(j, pos) * 64 + c * 8);
        }
    }
#pragma unroll
    for (int r = 0; r < UNR; ++r)
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        const int j = 128 * (i0 + r) + 32 * (u0 + u) + res;
        const u32x4 vv = (FUSED && j == pos) ? *(const u32x4*)(&newrow[2][c * 8]) : vq[r][u];
        const float pj = (j < kv_len) ? __builtin_amdgcn_exp2f(sc[min(j, pos)] - mx) : 0.f;
        l[u] += pj;
        const float pr = bf2f(f2bf(pj));  // probabilities enter the second contraction as bf16
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[u][2 * e] += pr * lo_bf(vv[e]);
          acc[u][2 * e + 1] += pr * hi_bf(vv[e]);
        }
      }
  }
  if constexpr (NW == 4) {
    // the four classes of this residue are in registers: (u0 + u1) + (u2 + u3)
#pragma unroll
    for (int e = 0; e < 8; ++e) red[0][res][c * 8 + e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
    if (c == 0) red[0][res][64] = (l[0] + l[1]) + (l[2] + l[3]);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[u0][res][c * 8 + e] = acc[0][e];
    if (c == 0) red[u0][res][64] = l[0];
  }
  __syncthreads();
  if (tid < 64) {
    float s = 0.f, lt = 0.f;
    for (int r = 0; r < 32; ++r) {
      if constexpr (NW == 4) {
        s += red[0][r][tid];
        lt += red[0][r][64];
      } else {
        s += (red[0][r][tid] + red[1][r][tid]) + (red[2][r][tid] + red[3][r][tid]);
        lt += (red[0][r][64] + red[1][r][64]) + (red[2][r][64] + red[3][r][64]);
      }
    }
    o[(int64_t)b * ldo + h * 64 + tid] = f2bf(lt > 0.f ? s / lt : 0.f);
  }
}

// waves per workgroup of the decode kernel: a function of the launch size only (same bits either way)
int decode_attn_waves(int batch, int n_heads) {
  static const int forced = [] { const char* e = getenv("MD_ATTN_DECODE_NW"); return e ? atoi(e) : 0; }();
  if (forced == 4 || forced == 16) return forced;
  return ((long)batch * n_heads <= 512) ? 16 : 4;
}
bool decode_attn_nt() {  // MD_ATTN_DECODE_NT=0: plain loads (A/B)
  static const bool on = [] { const char* e = getenv("MD_ATTN_DECODE_NT"); return !e || atoi(e) != 0; }();
  return on;
}

}  // namespace

extern "C" md_status md_attention_prefill(const md_attn_args* a, void* stream) {
  MD_CHECK_ARG(a && a->q && a->k && a->v && (a->o || a->o8));
  MD_CHECK_ARG(a->batch > 0 && a->n_heads > 0 && a->n_kv_heads > 0 && a->q_len > 0);
  MD_CHECK_ARG(a->n_heads % a->n_kv_heads == 0);
  MD_CHECK_ARG(a->head_dim == 64 || a->head_dim == 72);
  const int64_t strides[] = {a->q_bs, a->q_ts, a->q_hs, a->k_bs, a->k_ts, a->k_hs,
                             a->v_bs, a->v_ts, a->v_hs, a->o_bs, a->o_ts, a->o_hs};
  for (int64_t s : strides) MD_CHECK_ARG(s % 8 == 0);
  MD_CHECK_ARG((((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->o) & 15) == 0);
  if (a->o8 != nullptr)
    MD_CHECK_ARG(((uintptr_t)a->o8 & 7) == 0 && a->o8_bs % 8 == 0 && a->o8_ts % 8 == 0 && a->o8_ts >= (int64_t)a->n_heads * a->head_dim &&
                 a->o8_inv_scale > 0.f);
  AttnK k;
  k.q = (const bf16_t*)a->q;
  k.k = (const bf16_t*)a->k;
  k.v = (const bf16_t*)a->v;
  k.o = (bf16_t*)a->o;
  k.q_bs = a->q_bs; k.q_ts = a->q_ts; k.q_hs = a->q_hs;
  k.k_bs = a->k_bs; k.k_ts = a->k_ts; k.k_hs = a->k_hs;
  k.v_bs = a->v_bs; k.v_ts = a->v_ts; k.v_hs = a->v_hs;
  k.o_bs = a->o_bs; k.o_ts = a->o_ts; k.o_hs = a->o_hs;
  k.q_len = a->q_len;
  k.kv_len_all = a->kv_len_all;
  k.prefix = a->prefix_len;
  k.kv_group = a->n_heads / a->n_kv_heads;
  k.q_pos0 = a->q_pos0;
  k.kv_len = a->kv_len;
  k.scale_log2 = a->scale * 1.4426950408889634f;
  k.skip_dead = g_attn_skip_dead;
  k.o8 = (uint8_t*)a->o8;
  k.o8_bs = a->o8_bs;
  k.o8_ts = a->o8_ts;
  k.o8_inv_scale = a->o8_inv_scale;
  k.head_dim = a->head_dim;
  dim3 grid((a->q_len + 127) / 128, a->n_heads, a->batch);
  k.n_qblk = (a->q_len + 127) / 128;
  k.n_bh = a->batch * a->n_heads;
  k.n_heads = a->n_heads;
  const dim3 grid1(8 * k.n_qblk * ((k.n_bh + 7) / 8));
  hipStream_t s = (hipStream_t)stream;
  // MD_ATTN_VARIANT = reg | dma (default) | pipe for A/B runs
  static const int variant = [] {
    const char* e = getenv("MD_ATTN_VARIANT");
    if (e && e[0] == 'r') return 0;
    if (e && e[0] == 'p') return 2;
    return 1;
  }();
  if (variant == 0) {
    if (a->head_dim == 72)
      hipLaunchKernelGGL(attn_prefill_kernel<72>, grid, dim3(256), 0, s, k);
    else
      hipLaunchKernelGGL(attn_prefill_kernel<64>, grid, dim3(256), 0, s, k);
  } else if (variant == 1) {
    if (a->head_dim == 72)
      hipLaunchKernelGGL((attn_prefill_dma_kernel<72, false>), grid1, dim3(256), 0, s, k);
    else
      hipLaunchKernelGGL((attn_prefill_dma_kernel<64, false>), grid1, dim3(256), 0, s, k);
  } else {
    if (a->head_dim == 72)
      hipLaunchKernelGGL((attn_prefill_dma_kernel<72, true>), grid1, dim3(256), 0, s, k);
    else
      hipLaunchKernelGGL((attn_prefill_dma_kernel<64, true>), grid1, dim3(256), 0, s, k);
  }
  return md_launch_status();
}

extern "C" md_status md_attention_decode(const void* q, int64_t ldq, void* o, int64_t ldo,
                                         const void* k_slab, const void* v_slab,
                                         int64_t slab_batch_stride, int32_t ctx, const int32_t* kv_len,
                                         int32_t batch, int32_t n_heads, int32_t n_kv_heads,
                                         int32_t head_dim, float scale, void* stream) {
  MD_CHECK_ARG(q && o && k_slab && v_slab && kv_len);
  MD_CHECK_ARG(head_dim == 64 && ctx <= DEC_MAX_CTX && batch > 0 && n_heads % n_kv_heads == 0);
  MD_CHECK_ARG(ldq % 8 == 0 && ldo % 8 == 0 && ldq >= n_heads * 64 && ldo >= n_heads * 64);
  if (decode_attn_waves(batch, n_heads) == 16)
    hipLaunchKernelGGL((attn_decode_kernel<false, 16>), dim3(n_heads, batch), dim3(1024), 0, (hipStream_t)stream,
                       (const bf16_t*)q, ldq, (bf16_t*)o, ldo, (bf16_t*)k_slab, (bf16_t*)v_slab,
                       slab_batch_stride, ctx, kv_len, n_heads, n_heads / n_kv_heads,
                       scale * 1.4426950408889634f, (const float*)nullptr, 0);
  else if (n_heads == n_kv_heads && decode_attn_nt())
    hipLaunchKernelGGL((attn_decode_kernel<false, 4, true>), dim3(n_heads, batch), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)q, ldq, (bf16_t*)o, ldo, (bf16_t*)k_slab, (bf16_t*)v_slab,
                       slab_batch_stride, ctx, kv_len, n_heads, 1, scale * 1.4426950408889634f, (const float*)nullptr, 0);
  else
    hipLaunchKernelGGL((attn_decode_kernel<false, 4>), dim3(n_heads, batch), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)q, ldq, (bf16_t*)o, ldo, (bf16_t*)k_slab, (bf16_t*)v_slab,
                       slab_batch_stride, ctx, kv_len, n_heads, n_heads / n_kv_heads,
                       scale * 1.4426950408889634f, (const float*)nullptr, 0);
  return md_launch_status();
}

extern "C" md_status md_attention_decode_rope(const void* qkv, int64_t ld, void* o, int64_t ldo,
                                              const float* freqs, void* k_slab, void* v_slab,
                                              int64_t slab_batch_stride, int32_t ctx, const int32_t* kv_len,
                                              int32_t batch, int32_t n_heads, int32_t head_dim,
                                              int32_t rot_dim, float scale, void* stream) {
  MD_CHECK_ARG(qkv && o && freqs && k_slab && v_slab && kv_len);
  MD_CHECK_ARG(head_dim == 64 && ctx <= DEC_MAX_CTX && batch > 0 && n_heads > 0);
  MD_CHECK_ARG(rot_dim % 2 == 0 && rot_dim > 0 && rot_dim <= 64 && ld % 8 == 0 && ldo % 8 == 0);
  MD_CHECK_ARG(ld >= 3 * n_heads * 64 && ldo >= n_heads * 64);
  if (decode_attn_waves(batch, n_heads) == 16)
    hipLaunchKernelGGL((attn_decode_kernel<true, 16>), dim3(n_heads, batch), dim3(1024), 0, (hipStream_t)stream,
                       (const bf16_t*)qkv, ld, (bf16_t*)o, ldo, (bf16_t*)k_slab, (bf16_t*)v_slab,
                       slab_batch_stride, ctx, kv_len, n_heads, 1, scale * 1.4426950408889634f, freqs, rot_dim);
  else if (decode_attn_nt())
    hipLaunchKernelGGL((attn_decode_kernel<true, 4, true>), dim3(n_heads, batch), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)qkv, ld, (bf16_t*)o, ldo, (bf16_t*)k_slab, (bf16_t*)v_slab,
                       slab_batch_stride, ctx, kv_len, n_heads, 1, scale * 1.4426950408889634f, freqs, rot_dim);
  else
    hipLaunchKernelGGL((attn_decode_kernel<true, 4>), dim3(n_heads, batch), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)qkv, ld, (bf16_t*)o, ldo, (bf16_t*)k_slab, (bf16_t*)v_slab,
                       slab_batch_stride, ctx, kv_len, n_heads, 1, scale * 1.4426950408889634f, freqs, rot_dim);
  return md_launch_status();
}
