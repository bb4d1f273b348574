// Whole-stage entry points of the C ABI: the four seam functions of the
// reference (moondream.py:168-192) expressed as launch sequences of the kernels
// in this library.  Nothing here allocates or synchronises; every buffer,
// including the workspace, belongs to the caller.
#include "gemm_internal.hpp"
#include <cstdlib>

#include <algorithm>

md_status md_argmax_advance(const void* logits, int64_t ld, int32_t batch, int32_t vocab,
                            int32_t suppress_id, int32_t* next, int32_t* pos, hipStream_t stream);
md_status md_stitch_pool_batched(const void* feats, void* out, int64_t ld_out, int64_t out_img_stride,
                                 int32_t n_images, int32_t dim, int32_t grid, int32_t margin,
                                 int32_t tiles_h, int32_t tiles_w, hipStream_t stream);

md_status md_pool_grid_concat(const void* global_feats, const void* grid_feats, int32_t H, int32_t W,
                              void* out, int64_t ld_out, int32_t dim, int32_t grid, hipStream_t stream);

namespace {

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct Arena {
  char* base;
  size_t off;
  void* take(size_t bytes) {
    void* p = base ? base + off : nullptr;
    off += align_up(bytes);
    return p;
  }
};

// md_vit_model.tile_policy / md_text_model.tile_policy of the entry point this host thread is inside: every GEMM the entry
// point issues carries it in md_gemm_args.tile_policy.  Thread-local and scoped to the call, i.e. a property of the CALL
// (two models / threads in one process do not see each other's choice, unlike the process-wide knob of ABI <= 4).
thread_local int t_tile_policy = MD_TILE_BY_SHAPE;
struct TilePolicyScope {
  int saved;
  explicit TilePolicyScope(int p) : saved(t_tile_policy) { t_tile_policy = p; }
  ~TilePolicyScope() { t_tile_policy = saved; }
  TilePolicyScope(const TilePolicyScope&) = delete;
  TilePolicyScope& operator=(const TilePolicyScope&) = delete;
};
inline bool tile_policy_ok(int p) { return p == MD_TILE_BY_SHAPE || p == MD_TILE_PINNED || p == MD_TILE_DECODE_TALL; }

// Round 6: a decode step (q_len 1) of 65 .. 128 sequences as ONE pass over the weights -- the fused qkv|fc1 GEMM on the 128 x 64
// weight-streaming tile, proj / fc2 as K-slice partials of 128 rows + the fused tail, lm_head on its by-shape config -- instead of
// two passes of 64 rows.  Same K order per output element as the 64-row regime: a sequence gets the same bits either way.
// Measured (tools/bench_decode_gemm_m128.py): qkv|fc1 19.8 us vs 2 x 16.4, pair 18.6 (3-stage ring) vs 2 x 13.7, lm_head 57 vs 2 x 54.
// MD_DECODE_TALL=0: the round-5 behaviour (blocks of 64).  Not with the fp8 weight stream attached (its kernels are <= 64 rows).
inline bool decode_tall_model(const md_text_model* m) {
  static const bool allowed = [] { const char* e = getenv("MD_DECODE_TALL"); return !(e && e[0] == '0'); }();
  return allowed && m->blocks[0].qkv_fc1.w != nullptr && m->n_kv_heads == m->n_heads && !(m->fp8 && m->fp8->blocks) &&
         m->blocks[0].proj.b && m->blocks[0].fc2.b && m->blocks[0].proj.n == m->dim && m->blocks[0].fc2.n == m->dim && m->dim % 8 == 0;
}
inline bool decode_tall_rows(const md_text_model* m, int64_t rows, int q_len) { return q_len == 1 && rows > 64 && rows <= 128 && decode_tall_model(m); }

md_status gemm(const void* a, int64_t lda, const md_linear& lin, void* c, int64_t ldc, int m, int epi,
               const void* r, int64_t ldr, int res_row_mod, int store_pad, hipStream_t s,
               void* splitk_ws = nullptr, size_t splitk_bytes = 0) {
  md_gemm_args g;
  g.tile_policy = t_tile_policy;
  g.gelu_from_col = 0;
  g.splitk_ws = splitk_ws;
  g.splitk_ws_bytes = splitk_bytes;
  g.a = a;
  g.lda = lda;
  g.lin = lin;
  g.c = c;
  g.ldc = ldc;
  g.r = r;
  g.ldr = ldr;
  g.res_row_mod = res_row_mod;
  g.m = m;
  g.epilogue = epi;
  g.store_pad_cols = store_pad;
  return md_gemm_bf16(&g, s);
}


// ---- FP8 mode helpers (md_gemm_f8 and its activation producers; opt-in, see md_vit_f8 / md_text_f8) ----------------
md_status gemm_f8(const void* a8, int64_t lda, float a_scale, const md_linear_f8& lin, void* c, int64_t ldc, int m, int epi,
                  const void* r, int64_t ldr, int res_row_mod, int store_pad, hipStream_t s, void* c8 = nullptr, int64_t ldc8 = 0,
                  float c8_scale = 1.f, int f8_from = 0, int gelu_from = 0) {
  md_gemm_f8_args g;
  g.a = a8; g.lda = lda; g.a_scale = a_scale; g.lin = lin; g.c = c; g.ldc = ldc;
  g.c8 = c8; g.ldc8 = ldc8; g.c8_inv_scale = c8 ? 1.0f / c8_scale : 1.0f; g.f8_from_col = f8_from;
  g.r = r; g.ldr = ldr; g.res_row_mod = res_row_mod; g.m = m; g.epilogue = epi; g.store_pad_cols = store_pad;
  g.gelu_from_col = gelu_from;
  return md_gemm_f8(&g, s);
}
// calibration: running max |x| of a quantisation site
md_status calib_site(float* calib, int site, const void* x, int64_t ldx, int rows, int cols, hipStream_t s) {
  if (!calib) return MD_OK;
  return md_amax_bf16(x, ldx, rows, cols / 8 * 8, calib + site, s);
}
bool f8_scales_ok(float a, float b, float c, float d = 1.f) { return a > 0.f && b > 0.f && c > 0.f && d > 0.f; }

struct VitWs {
  void *patches, *x, *h, *qkv, *ff;
  size_t total;
};

VitWs vit_layout(const md_vit_model* m, int n_crops, void* base) {
  const size_t g = m->crop / m->patch, M = (size_t)n_crops * g * g;
  Arena a{(char*)base, 0};
  VitWs w;
  w.patches = a.take(M * m->patch_emb.k_pad * 2);
  w.x = a.take(M * m->dim * 2);
  w.h = a.take(M * m->blocks[0].qkv.k_pad * 2);  // GEMM A operand: padded leading dim, zero pad
  w.qkv = a.take(M * 3 * m->dim * 2);
  w.ff = a.take(M * m->blocks[0].fc1.n_pad * 2);
  w.total = a.off;
  return w;
}

// Zero fill as a KERNEL, never hipMemsetAsync: a memset captured into a hipGraph becomes a memset node, and from the
// second replay of a graph on this stack (ROCm 7.2) such a node is not reliably ordered against its neighbouring kernel
// nodes -- the arrival tickets of the in-launch split-K GEMM were re-zeroed while its K slices were arriving, and decode
// graphs replayed in a later generator run produced garbage (tools/debug_stale_graph.py; profiles/r03_stale_graph_replay.txt).
// A kernel node is ordered like every other kernel of the captured stream.
__global__ void zero_fill_kernel(u32x4* p, size_t n16) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16) p[i] = u32x4{0u, 0u, 0u, 0u};
}
md_status zero_fill(void* p, size_t bytes, hipStream_t s) {  // p 16-byte aligned, bytes a multiple of 16
  if (bytes == 0) return MD_OK;
  if (((uintptr_t)p & 15) != 0 || (bytes & 15) != 0) return MD_ERR_INVALID_ARG;
  const size_t n16 = bytes / 16;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, (u32x4*)p, n16);
  return md_launch_status();
}

// zero an A-operand buffer whose leading dimension is padded beyond its logical width
md_status zero_if_padded(void* p, size_t rows, int ld, int width, hipStream_t s) {
  if (ld == width) return MD_OK;
  return zero_fill(p, (rows * (size_t)ld * 2 + 15) / 16 * 16, s);
}

struct TextWs {
  void *h, *qkv, *att, *ff, *pos_kv, *splitk;
  float* rope_cs;      // prefill with the fused RoPE / KV-write epilogue: fp32 [M][32] + uint32 [M]
  uint32_t* rope_kv;
  size_t splitk_bytes;
  // decode regime: fp32 partial products of proj / fc2 (launch-boundary split-K)
  float *part_a, *part_b;
  int64_t part_ld, part_stride;
  size_t total;
};

TextWs text_layout(const md_text_model* m, int batch, int q_len, void* base) {
  const size_t M = (size_t)batch * q_len;
  const size_t hd = m->dim / m->n_heads;
  Arena a{(char*)base, 0};
  TextWs w;
  w.h = a.take(M * m->blocks[0].qkv.k_pad * 2);
  if (m->blocks[0].qkv_fc1.w) {
    // fused qkv|fc1 activation: one buffer, fc1's GELU output starts at column qkv_w
    w.qkv = a.take(M * (size_t)m->blocks[0].qkv_fc1.n_pad * 2);
    w.ff = (char*)w.qkv + (m->n_heads + 2 * m->n_kv_heads) * hd * 2;
  } else {
    w.qkv = a.take(M * (m->n_heads + 2 * m->n_kv_heads) * hd * 2);
    w.ff = a.take(M * m->blocks[0].fc1.n_pad * 2);
  }
  w.att = a.take(M * m->blocks[0].proj.k_pad * 2);
  w.pos_kv = a.take((size_t)batch * 4);
  w.rope_cs = nullptr;
  w.rope_kv = nullptr;
  if (M > 64 && q_len > 1 && m->rot_dim == 32) {
    w.rope_cs = (float*)a.take(M * 32 * sizeof(float));
    w.rope_kv = (uint32_t*)a.take(M * sizeof(uint32_t));
  }
  // decode regime: split-K scratch shared by the layer's four linears (stream-ordered)
  size_t sk = 0;
  const bool tall = decode_tall_rows(m, (int64_t)M, q_len);
  if (M <= 64 || tall) {
    const md_text_block& b0 = m->blocks[0];
    sk = std::max(std::max(md_gemm_workspace_bytes(&b0.qkv, (int)M, 0), md_gemm_workspace_bytes(&b0.proj, (int)M, 0)),
                  std::max(md_gemm_workspace_bytes(&b0.fc1, (int)M, 1), md_gemm_workspace_bytes(&b0.fc2, (int)M, 0)));
    if (b0.qkv_fc1.w) sk = std::max(sk, md_gemm_workspace_bytes(&b0.qkv_fc1, (int)M, 1));
  }
  w.splitk_bytes = sk;
  w.splitk = a.take(sk);
  w.part_a = w.part_b = nullptr;
  w.part_ld = w.part_stride = 0;
  if (M <= 64 || tall) {
    const md_text_block& b0 = m->blocks[0];
    w.part_ld = ((int64_t)m->dim + 3) / 4 * 4;
    w.part_stride = (int64_t)M * w.part_ld;
    w.part_a = (float*)a.take((size_t)md_gemm_partial_slices(&b0.proj) * w.part_stride * 4);
    w.part_b = (float*)a.take((size_t)md_gemm_partial_slices(&b0.fc2) * w.part_stride * 4);
  }
  w.total = a.off;
  return w;
}

// per token row of a prefill: the (cos, sin) row of its position and its byte offset in a layer's K / V slab -- what the fused
// RoPE / KV-write epilogue of the qkv GEMM (MD_EPI_QKV_ROPE) needs per row, computed once per forward for all layers
__global__ __launch_bounds__(256) void rope_rowinfo_kernel(const int32_t* __restrict__ pos0, const float* __restrict__ freqs, float* __restrict__ row_cs,
                                                           uint32_t* __restrict__ row_kv, int q_len, int rows, int64_t slab_bs, int hd, int half2) {
  const int m = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  if (m >= rows) return;
  const int b = m / q_len, t = m - b * q_len, pos = pos0[b] + t;
  if (l < half2) row_cs[(int64_t)m * half2 + l] = freqs[(int64_t)pos * half2 + l];
  if (l == 0) row_kv[m] = (uint32_t)(((int64_t)b * slab_bs + (int64_t)pos * hd) * 2);
}

__global__ void kv_len_kernel(const int32_t* pos0, int32_t* kv_len, int q_len, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) kv_len[i] = pos0[i] + q_len;
}

}  // namespace

extern "C" int md_abi_version(void) { return MD_ABI_VERSION; }

extern "C" const char* md_status_string(md_status s) {
  switch (s) {
    case MD_OK: return "ok";
    case MD_ERR_INVALID_ARG: return "invalid argument (shape / alignment / null pointer)";
    case MD_ERR_LAUNCH: return "HIP launch failed (is a gfx950 device visible?)";
    case MD_ERR_WORKSPACE: return "workspace too small";
    case MD_ERR_UNSUPPORTED: return "unsupported shape";
    default: return "unknown status";
  }
}

// ------------------------------------------------------------------ vision
extern "C" size_t md_vit_workspace_bytes(const md_vit_model* m, int32_t n_crops) {
  if (!m || !m->blocks || n_crops <= 0) return 0;
  return vit_layout(m, n_crops, nullptr).total;
}

// reference: vision.py:64-74
extern "C" md_status md_vit_encode(const md_vit_model* m, const void* crops, int32_t crops_kind,
                                   int32_t n_crops, void* out, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  MD_CHECK_ARG(m && crops && out && workspace && m->blocks && n_crops > 0);
  MD_CHECK_ARG(tile_policy_ok(m->tile_policy));
  TilePolicyScope tile_scope(m->tile_policy);
  MD_CHECK_ARG(m->dim % m->n_heads == 0 && m->crop % m->patch == 0);
  const int hd = m->dim / m->n_heads;
  if (hd != 72 && hd != 64) return MD_ERR_UNSUPPORTED;
  const VitWs w = vit_layout(m, n_crops, workspace);
  if (workspace_bytes < w.total) return MD_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int g = m->crop / m->patch, T = g * g, M = n_crops * T, D = m->dim;
  const int64_t kp = m->patch_emb.k_pad;
  const int Dp = m->blocks[0].qkv.k_pad;  // == round_up(D, 64) for every consumer of h
  MD_CHECK_ARG(Dp >= D && m->blocks[0].proj.k_pad == Dp && m->blocks[0].fc1.k_pad == Dp);
  MD_TRY(zero_if_padded(w.h, M, Dp, D, s));

  if (crops_kind == MD_CROPS_U8_HWC) {
    MD_CHECK_ARG(m->pixel_lut != nullptr);
    MD_TRY(md_patchify_u8(crops, m->pixel_lut, w.patches, kp, n_crops, m->crop, m->patch, s));
  } else {
    MD_TRY(md_patchify_bf16(crops, w.patches, kp, n_crops, m->crop, m->patch, s));
  }
  // x = patch_emb(patches) + pos_emb          (vision.py:67-68)
  MD_TRY(gemm(w.patches, kp, m->patch_emb, w.x, D, M, MD_EPI_RESIDUAL, m->pos_emb, D, T, 0, s));

  const float scale = 1.0f / sqrtf((float)hd);
  // FP8 mode (opt-in): LN -> fp8, the four linears on md_gemm_f8, attention output quantised for proj.  The fp8
  // activations live in the (then unused) bf16 ff buffer: [ff8: M x fc1.n_pad bytes | h8: M x Dp bytes].
  const md_vit_f8* f8 = m->f8;
  float* calib = f8 ? f8->calib : nullptr;
  const bool use_f8 = f8 && !calib && f8->blocks;
  uint8_t *ff8 = (uint8_t*)w.ff, *h8 = (uint8_t*)w.ff + (size_t)M * m->blocks[0].fc1.n_pad;
  if (use_f8) MD_CHECK_ARG(m->blocks[0].fc1.n_pad >= Dp);
  for (int l = 0; l < m->n_layers; ++l) {
    const md_vit_block& b = m->blocks[l];
    md_attn_args a = {};
    const bf16_t* qkv = (const bf16_t*)w.qkv;
    a.q = qkv;
    a.k = qkv + D;
    a.v = qkv + 2 * D;
    a.o = w.h;
    a.q_bs = a.k_bs = a.v_bs = (int64_t)T * 3 * D;
    a.q_ts = a.k_ts = a.v_ts = 3 * D;
    a.q_hs = a.k_hs = a.v_hs = hd;
    a.o_bs = (int64_t)T * Dp;
    a.o_ts = Dp;
    a.o_hs = hd;
    a.batch = n_crops;
    a.n_heads = a.n_kv_heads = m->n_heads;
    a.head_dim = hd;
    a.q_len = T;
    a.kv_len_all = T;
    a.q_pos0 = nullptr;
    a.kv_len = nullptr;
    a.prefix_len = T;  // everything visible: no mask
    a.scale = scale;
    if (use_f8) {
      const md_vit_block_f8& q = f8->blocks[l];
      MD_CHECK_ARG(f8_scales_ok(q.s_ln1, q.s_att, q.s_ln2, q.s_ff) && q.fc2.k_pad == q.fc1.n_pad && q.qkv.k_pad == Dp);
      MD_TRY(md_layernorm_f8(w.x, D, h8, Dp, &b.ln1, M, D, Dp, 1e-5f, 1.0f / q.s_ln1, s));
      MD_TRY(gemm_f8(h8, Dp, q.s_ln1, q.qkv, w.qkv, 3 * D, M, MD_EPI_BIAS, nullptr, 0, 0, 0, s));
      if (Dp == D) {  // the attention epilogue writes proj's e4m3 operand itself (h8 holds ln1's output until the qkv GEMM has read it)
        md_attn_args a8 = a;
        a8.o = nullptr;
        a8.o8 = h8;
        a8.o8_bs = (int64_t)T * Dp;
        a8.o8_ts = Dp;
        a8.o8_inv_scale = 1.0f / q.s_att;
        MD_TRY(md_attention_prefill(&a8, s));
      } else {
        MD_TRY(md_attention_prefill(&a, s));
        MD_TRY(md_quantize_f8(w.h, Dp, h8, Dp, M, D, Dp, 1.0f / q.s_att, s));
      }
      MD_TRY(gemm_f8(h8, Dp, q.s_att, q.proj, w.x, D, M, MD_EPI_RESIDUAL, w.x, D, 0, 0, s));
      MD_TRY(md_layernorm_f8(w.x, D, h8, Dp, &b.ln2, M, D, Dp, 1e-5f, 1.0f / q.s_ln2, s));
      MD_TRY(gemm_f8(h8, Dp, q.s_ln2, q.fc1, nullptr, 0, M, MD_EPI_GELU, nullptr, 0, 0, 1, s, ff8, q.fc1.n_pad, q.s_ff, 0, 0));
      MD_TRY(gemm_f8(ff8, q.fc1.n_pad, q.s_ff, q.fc2, w.x, D, M, MD_EPI_RESIDUAL, w.x, D, 0, 0, s));
      continue;
    }
    // x = x + attn(ln1(x))                    (vision.py:70, layers.py:155-166)
    MD_TRY(md_layernorm_bf16(w.x, D, w.h, Dp, &b.ln1, M, D, 1e-5f, s));
    MD_TRY(calib_site(calib, 4 * l + 0, w.h, Dp, M, D, s));
    MD_TRY(gemm(w.h, Dp, b.qkv, w.qkv, 3 * D, M, MD_EPI_BIAS, nullptr, 0, 0, 0, s));
    MD_TRY(md_attention_prefill(&a, s));
    MD_TRY(calib_site(calib, 4 * l + 1, w.h, Dp, M, D, s));
    MD_TRY(gemm(w.h, Dp, b.proj, w.x, D, M, MD_EPI_RESIDUAL, w.x, D, 0, 0, s));
    // x = x + mlp(ln2(x))                     (vision.py:71, layers.py:129-146)
    MD_TRY(md_layernorm_bf16(w.x, D, w.h, Dp, &b.ln2, M, D, 1e-5f, s));
    MD_TRY(calib_site(calib, 4 * l + 2, w.h, Dp, M, D, s));
    MD_TRY(gemm(w.h, Dp, b.fc1, w.ff, b.fc1.n_pad, M, MD_EPI_GELU, nullptr, 0, 0, 1, s));
    MD_TRY(calib_site(calib, 4 * l + 3, w.ff, b.fc1.n_pad, M, b.fc1.n, s));
    MD_CHECK_ARG(b.fc2.k_pad == b.fc1.n_pad);
    MD_TRY(gemm(w.ff, b.fc1.n_pad, b.fc2, w.x, D, M, MD_EPI_RESIDUAL, w.x, D, 0, 0, s));
  }
  return md_layernorm_bf16(w.x, D, out, D, &m->post_ln, M, D, 1e-5f, s);  // vision.py:72
}

extern "C" size_t md_vision_project_workspace_bytes(const md_vit_model* m, int32_t n_images) {
  if (!m || n_images <= 0) return 0;
  const size_t g = m->crop / m->patch, M = (size_t)n_images * g * g;
  return align_up(M * m->proj_fc1.k_pad * 2) + align_up(M * m->proj_fc1.n_pad * 2);
}

// the projector MLP over M rows of [global | pooled] features (vision.py:87-89); fp8 mode: the concatenated input is
// quantised into the second half of the (bf16-sized) ff buffer, the GELU output into its first half
static md_status projector_mlp(const md_vit_model* m, void* cat, int Cp, void* ff, int M, void* out, int64_t ld_out, hipStream_t s) {
  const md_vit_f8* f8 = m->f8;
  float* calib = f8 ? f8->calib : nullptr;
  const int nl = m->n_layers, D2 = m->proj_fc1.k;
  if (f8 && !calib && f8->blocks && f8->proj_fc1.w && f8->proj_fc2.w) {
    MD_CHECK_ARG(f8_scales_ok(f8->s_cat, f8->s_pff, 1.f) && f8->proj_fc1.k_pad == Cp && f8->proj_fc2.k_pad == f8->proj_fc1.n_pad &&
                 f8->proj_fc1.n_pad >= Cp);
    uint8_t *ff8 = (uint8_t*)ff, *cat8 = (uint8_t*)ff + (size_t)M * f8->proj_fc1.n_pad;
    MD_TRY(md_quantize_f8(cat, Cp, cat8, Cp, M, D2 / 8 * 8, Cp, 1.0f / f8->s_cat, s));
    MD_TRY(gemm_f8(cat8, Cp, f8->s_cat, f8->proj_fc1, nullptr, 0, M, MD_EPI_GELU, nullptr, 0, 0, 1, s, ff8, f8->proj_fc1.n_pad, f8->s_pff, 0, 0));
    return gemm_f8(ff8, f8->proj_fc1.n_pad, f8->s_pff, f8->proj_fc2, out, ld_out, M, MD_EPI_BIAS, nullptr, 0, 0, 0, s);
  }
  MD_TRY(calib_site(calib, 4 * nl + 0, cat, Cp, M, D2, s));
  MD_TRY(gemm(cat, Cp, m->proj_fc1, ff, m->proj_fc1.n_pad, M, MD_EPI_GELU, nullptr, 0, 0, 1, s));
  MD_TRY(calib_site(calib, 4 * nl + 1, ff, m->proj_fc1.n_pad, M, m->proj_fc1.n, s));
  return gemm(ff, m->proj_fc1.n_pad, m->proj_fc2, out, ld_out, M, MD_EPI_BIAS, nullptr, 0, 0, 0, s);
}

// reference: moondream.py:213-228 + vision.py:77-89
extern "C" md_status md_vision_project(const md_vit_model* m, const void* feats, int32_t n_images,
                                       int32_t tiles_h, int32_t tiles_w, int32_t margin, void* out,
                                       int64_t ld_out, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  MD_CHECK_ARG(m && feats && out && workspace && n_images > 0 && tiles_h > 0 && tiles_w > 0);
  MD_CHECK_ARG(tile_policy_ok(m->tile_policy));
  TilePolicyScope tile_scope(m->tile_policy);
  if (workspace_bytes < md_vision_project_workspace_bytes(m, n_images)) return MD_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int g = m->crop / m->patch, T = g * g, D = m->dim, M = n_images * T;
  MD_CHECK_ARG(m->proj_fc1.k == 2 * D && g > 2 * margin);
  MD_CHECK_ARG(m->proj_fc2.k_pad == m->proj_fc1.n_pad && ld_out >= m->proj_fc2.n);
  const int Cp = m->proj_fc1.k_pad;
  Arena a{(char*)workspace, 0};
  void* cat = a.take((size_t)M * Cp * 2);
  void* ff = a.take((size_t)M * m->proj_fc1.n_pad * 2);
  MD_TRY(zero_if_padded(cat, M, Cp, 2 * D, s));
  MD_TRY(md_stitch_pool_batched(feats, cat, Cp, (int64_t)T * Cp, n_images, D, g, margin, tiles_h,
                                tiles_w, s));
  return projector_mlp(m, cat, Cp, ff, M, out, ld_out, s);
}

// the seam form: _vis_proj(g, r) with r already stitched (moondream.py:171-172, vision.py:77-89)
extern "C" md_status md_vision_project_grid(const md_vit_model* m, const void* global_feats,
                                            const void* grid_feats, int32_t H, int32_t W, void* out,
                                            int64_t ld_out, void* workspace, size_t workspace_bytes,
                                            void* stream) {
  MD_CHECK_ARG(m && global_feats && grid_feats && out && workspace && H > 0 && W > 0);
  MD_CHECK_ARG(tile_policy_ok(m->tile_policy));
  TilePolicyScope tile_scope(m->tile_policy);
  if (workspace_bytes < md_vision_project_workspace_bytes(m, 1)) return MD_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int g = m->crop / m->patch, T = g * g, D = m->dim;
  MD_CHECK_ARG(m->proj_fc1.k == 2 * D && m->proj_fc2.k_pad == m->proj_fc1.n_pad && ld_out >= m->proj_fc2.n);
  const int Cp = m->proj_fc1.k_pad;
  Arena a{(char*)workspace, 0};
  void* cat = a.take((size_t)T * Cp * 2);
  void* ff = a.take((size_t)T * m->proj_fc1.n_pad * 2);
  MD_TRY(zero_if_padded(cat, T, Cp, 2 * D, s));
  MD_TRY(md_pool_grid_concat(global_feats, grid_feats, H, W, cat, Cp, D, g, s));
  return projector_mlp(m, cat, Cp, ff, T, out, ld_out, s);
}

// -------------------------------------------------------------------- text
extern "C" size_t md_text_workspace_bytes(const md_text_model* m, int32_t batch, int32_t q_len) {
  if (!m || !m->blocks || batch <= 0 || q_len <= 0) return 0;
  size_t need = text_layout(m, batch, q_len, nullptr).total;
  // decode steps of more than 64 (128) sequences run as blocks of 64 (128: decode_tall_model) rows
  if (q_len == 1 && batch > 64) need = std::max(need, std::max(text_layout(m, 64, 1, nullptr).total, text_layout(m, std::min(batch, 128), 1, nullptr).total));
  return need;
}

md_status md_kv_quantize_f8_layer(const md_kv_cache* kv, int layer, const int32_t* pos0, int pos_fixed, int batch, int n_heads, int n_pos,
                                  hipStream_t s);
md_status md_attention_decode_rope_f8_launch(const void* qkv, int64_t ld, void* o, int64_t ldo, const float* freqs, void* k_slab, void* v_slab,
                                             void* k8_slab, void* v8_slab, int64_t slab_batch_stride, int32_t ctx, const int32_t* kv_len,
                                             int32_t batch, int32_t n_heads, int32_t rot_dim, float scale, float k_scale, float v_scale,
                                             hipStream_t s);

// reference: text.py:128-160 (text_decoder) with text.py:16-60 (attn)
extern "C" md_status md_text_forward(const md_text_model* m, const void* x_in, void* hidden,
                                     int32_t batch, int32_t q_len, const int32_t* pos0,
                                     const md_kv_cache* kv, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  MD_CHECK_ARG(m && x_in && hidden && pos0 && kv && kv->k && kv->v && workspace && m->blocks);
  MD_CHECK_ARG(batch > 0 && q_len > 0 && m->dim % m->n_heads == 0);
  MD_CHECK_ARG(tile_policy_ok(m->tile_policy));
  TilePolicyScope tile_scope(m->tile_policy);
  // the e4m3 copy of the KV cache (fp8 mode) keeps the 64-row blocks: its attention kernel is part of an opt-in mode tuned there
  const bool tall_ok = decode_tall_model(m) && !(kv->k8 && kv->v8);
  const int block_rows = tall_ok ? 128 : 64;
  if (q_len == 1 && batch > block_rows) {
    // A decode step over more sequences than the decode regime takes in one pass: blocks of 64 (128) rows, each through the
    // decode-regime kernels (weight-streaming GEMMs, launch-boundary split-K, fused block tail).  The weights are
    // streamed once per block; the big-tile kernels this replaces ran the step ~1.5x slower at 128 rows.
    for (int b0 = 0; b0 < batch; b0 += block_rows) {
      const int nb = std::min(block_rows, batch - b0);
      md_kv_cache sub = *kv;
      sub.k = (char*)kv->k + (int64_t)b0 * kv->batch_stride * 2;
      sub.v = (char*)kv->v + (int64_t)b0 * kv->batch_stride * 2;
      // the e4m3 copy of the cache (fp8 mode) has the same slot layout at one byte per element
      if (kv->k8) sub.k8 = (char*)kv->k8 + (int64_t)b0 * kv->batch_stride;
      if (kv->v8) sub.v8 = (char*)kv->v8 + (int64_t)b0 * kv->batch_stride;
      MD_TRY(md_text_forward(m, (const char*)x_in + (int64_t)b0 * m->dim * 2, (char*)hidden + (int64_t)b0 * m->dim * 2, nb, 1,
                             pos0 + b0, &sub, workspace, workspace_bytes, stream));
    }
    return MD_OK;
  }
  const int hd = m->dim / m->n_heads;
  if (hd != 64) return MD_ERR_UNSUPPORTED;
  const TextWs w = text_layout(m, batch, q_len, workspace);
  if (workspace_bytes < w.total) return MD_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int D = m->dim, M = batch * q_len;
  const int qkv_w = (m->n_heads + 2 * m->n_kv_heads) * hd;
  const int Dp = m->blocks[0].qkv.k_pad;
  MD_CHECK_ARG(Dp >= D && m->blocks[0].proj.k_pad == Dp && m->blocks[0].fc1.k_pad == Dp);
  MD_TRY(zero_if_padded(w.h, M, Dp, D, s));
  MD_TRY(zero_if_padded(w.att, M, Dp, D, s));
  bf16_t* x = (bf16_t*)hidden;
  if (x_in != hidden) {
    if (hipMemcpyAsync(hidden, x_in, (size_t)M * D * 2, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return MD_ERR_LAUNCH;
  }
  if (w.splitk_bytes) {
    MD_TRY(zero_fill(w.splitk, 8192, s));  // arrival tickets
  }
  int32_t* kv_len = (int32_t*)w.pos_kv;
  hipLaunchKernelGGL(kv_len_kernel, dim3((batch + 255) / 256), dim3(256), 0, s, pos0, kv_len, q_len, batch);
  const float scale = 1.0f / sqrtf((float)hd);
  // decode regime (<= 64 rows): launch-boundary split-K for proj / fc2 + fused block tail.
  // A function of the row count only, like the choice of GEMM kernel.  MD_TEXT_TAIL=0: A/B runs.
  static const bool tail_allowed = [] { const char* e = getenv("MD_TEXT_TAIL"); return !(e && e[0] == '0'); }();
  const bool tall = tall_ok && decode_tall_rows(m, M, q_len);   // 65 .. 128 rows of a decode step: one pass (decode_tall_model)
  const bool tail_fused = tail_allowed && (M <= 64 || tall) && D % 8 == 0 && m->blocks[0].proj.b && m->blocks[0].fc2.b &&
                          m->blocks[0].proj.n == D && m->blocks[0].fc2.n == D;
  TilePolicyScope tall_scope(tall ? MD_TILE_DECODE_TALL : t_tile_policy);

  // Prefill: RoPE + KV write in the epilogue of the fused qkv|fc1 GEMM (MD_EPI_QKV_ROPE) when the four-wave kernel takes the
  // shape: MHA, head_dim 64, rot_dim 32, slab offsets in 32 bits.  Per-row positions / slab offsets once per forward.
  const uint64_t slab_bytes = (uint64_t)kv->layer_stride * 2;
  bool rope_in_gemm = w.rope_cs != nullptr && q_len > 1 && M > 64 && m->n_kv_heads == m->n_heads && hd == 64 && m->rot_dim == 32 &&
                      m->blocks[0].qkv_fc1.w != nullptr && slab_bytes < 0xfffff000ull && kv->layer_stride >= (int64_t)batch * kv->batch_stride;
  if (rope_in_gemm)
    hipLaunchKernelGGL(rope_rowinfo_kernel, dim3((M + 7) / 8), dim3(256), 0, s, pos0, m->freqs, w.rope_cs, w.rope_kv, q_len, M,
                       kv->batch_stride, hd, m->rot_dim);
  bool rope_done = false;  // set per block when the fused launch took it
  bool rope_done_kv8 = false;  // ... and also wrote the e4m3 copy of the rows (fp8 mode)

  // rope(q), rope(k), cache update (text.py:42-46) and attention over the slab (text.py:48-51) of block l
  auto rope_and_attention = [&](int l, int64_t qld, bool fuse_rope, uint8_t* att8 = nullptr, float att8_inv_scale = 0.f) -> md_status {
    bf16_t* kl = (bf16_t*)kv->k + (int64_t)l * kv->layer_stride;
    bf16_t* vl = (bf16_t*)kv->v + (int64_t)l * kv->layer_stride;
    if (!fuse_rope && !rope_done)
      MD_TRY(md_rope_kv_write(w.qkv, qld, m->freqs, pos0, kl, vl, kv->batch_stride, kv->ctx, batch,
                              q_len, m->n_heads, m->n_kv_heads, hd, m->rot_dim, s));
    // attention over the slab                                   (text.py:48-51)
    const bool kv8 = kv->k8 && kv->v8 && kv->k_scale && kv->v_scale && m->n_kv_heads == m->n_heads && hd == 64;
    if (fuse_rope && kv8) {
      // fp8 mode: the step attends over the e4m3 copy of the cache and writes the new row into both copies
      MD_TRY(md_attention_decode_rope_f8_launch(w.qkv, qld, w.att, Dp, m->freqs, kl, vl, (uint8_t*)kv->k8 + (int64_t)l * kv->layer_stride,
                                                (uint8_t*)kv->v8 + (int64_t)l * kv->layer_stride, kv->batch_stride, kv->ctx, kv_len, batch,
                                                m->n_heads, m->rot_dim, scale, kv->k_scale[l], kv->v_scale[l], s));
    } else if (fuse_rope) {
      MD_TRY(md_attention_decode_rope(w.qkv, qld, w.att, Dp, m->freqs, kl, vl, kv->batch_stride, kv->ctx, kv_len,
                                      batch, m->n_heads, hd, m->rot_dim, scale, s));
    } else if (q_len == 1) {
      MD_TRY(md_attention_decode(w.qkv, qld, w.att, Dp, kl, vl, kv->batch_stride, kv->ctx, kv_len, batch,
                                 m->n_heads, m->n_kv_heads, hd, scale, s));
    } else {
      md_attn_args a = {};
      a.q = w.qkv;
      a.q_bs = (int64_t)q_len * qld;
      a.q_ts = qld;
      a.q_hs = hd;
      a.k = kl;
      a.v = vl;
      a.k_bs = a.v_bs = kv->batch_stride;
      a.k_ts = a.v_ts = hd;
      a.k_hs = a.v_hs = (int64_t)kv->ctx * hd;
      a.o = w.att;
      a.o_bs = (int64_t)q_len * Dp;
      a.o_ts = Dp;
      a.o_hs = hd;
      a.batch = batch;
      a.n_heads = m->n_heads;
      a.n_kv_heads = m->n_kv_heads;
      a.head_dim = hd;
      a.q_len = q_len;
      a.kv_len_all = 0;
      a.q_pos0 = pos0;
      a.kv_len = kv_len;
      a.prefix_len = m->prefix_len;
      a.scale = scale;
      if (att8 != nullptr) {  // fp8 mode: proj reads e4m3 rows only -- written by the attention epilogue, no bf16 copy, no quantise pass
        a.o = nullptr;
        a.o8 = att8;
        a.o8_bs = (int64_t)q_len * Dp;
        a.o8_ts = Dp;
        a.o8_inv_scale = att8_inv_scale;
      }
      MD_TRY(md_attention_prefill(&a, s));
    }
    // fp8 mode: the rows this pass wrote (bf16) also go into the e4m3 copy the decode steps read
    if (kv8 && !fuse_rope && !rope_done_kv8) MD_TRY(md_kv_quantize_f8_layer(kv, l, pos0, 0, batch, m->n_heads, q_len, s));
    return MD_OK;
  };

  // FP8 prefill (opt-in, md_text_f8): launches of more than 64 rows with the fused qkv|fc1 packing
  const md_text_f8* f8p = m->f8;
  float* calib = (f8p && M > 64) ? f8p->calib : nullptr;  // calibration records the PREFILL's activation ranges
  const bool use_f8 = f8p && !f8p->calib && f8p->blocks && M > 64 && m->blocks[0].qkv_fc1.w != nullptr && qkv_w % 64 == 0;

  for (int l = 0; l < m->n_layers; ++l) {
    const md_text_block& b = m->blocks[l];
    if (use_f8) {
      const md_text_block_f8& q = f8p->blocks[l];
      const int64_t qld = b.qkv_fc1.n_pad;
      MD_CHECK_ARG(f8_scales_ok(q.s_ln, q.s_att, q.s_ff) && q.qkv_fc1.n_pad == b.qkv_fc1.n_pad && q.qkv_fc1.k_pad == Dp &&
                   q.fc2.k_pad == b.fc1.n_pad && q.proj.k_pad == Dp && b.qkv_fc1.n_pad == qkv_w + b.fc1.n_pad);
      // fp8 activations: ln(x) / the attention output in the (bf16-sized) h buffer, gelu(fc1) in the fc1 columns' own
      // (bf16-sized) slots of the fused activation rows
      uint8_t* h8 = (uint8_t*)w.h;
      uint8_t* ff8 = (uint8_t*)w.qkv + (size_t)qkv_w * 2;
      MD_TRY(md_layernorm_f8(x, D, h8, Dp, &b.ln, M, D, Dp, 1e-5f, 1.0f / q.s_ln, s));
      // [qkv | gelu(fc1) -> e4m3]; with per-row positions at hand the epilogue also rotates q / k and writes k / v to the
      // slab and to its e4m3 copy (MD_EPI_QKV_ROPE, as the bf16 kernel does): no rope_kv_kernel, no kv-quantise pass
      rope_done = rope_done_kv8 = false;
      if (rope_in_gemm) {
        md_gemm_f8_args g;
        g.a = h8; g.lda = Dp; g.a_scale = q.s_ln; g.lin = q.qkv_fc1; g.c = w.qkv; g.ldc = qld;
        g.c8 = ff8; g.ldc8 = qld * 2; g.c8_inv_scale = 1.0f / q.s_ff; g.f8_from_col = qkv_w;
        g.r = nullptr; g.ldr = 0; g.res_row_mod = 0; g.m = M; g.epilogue = MD_EPI_GELU; g.store_pad_cols = 1; g.gelu_from_col = qkv_w;
        md_rope_fuse rf;
        rf.row_cs = w.rope_cs; rf.row_kv = w.rope_kv;
        rf.kslab = (bf16_t*)kv->k + (int64_t)l * kv->layer_stride;
        rf.vslab = (bf16_t*)kv->v + (int64_t)l * kv->layer_stride;
        rf.slab_bytes = slab_bytes; rf.n_heads = m->n_heads; rf.ctx = kv->ctx;
        md_rope_fuse_f8 rf8 = {nullptr, nullptr, 1.f, 1.f};
        const bool has8 = kv->k8 && kv->v8 && kv->k_scale && kv->v_scale;
        if (has8) {
          rf8.k8slab = (uint8_t*)kv->k8 + (int64_t)l * kv->layer_stride;
          rf8.v8slab = (uint8_t*)kv->v8 + (int64_t)l * kv->layer_stride;
          rf8.k_scale = kv->k_scale[l];
          rf8.v_scale = kv->v_scale[l];
        }
        const md_status fs = md_gemm_f8_qkv_rope(&g, &rf, &rf8, s);
        if (fs == MD_OK) {
          rope_done = true;
          rope_done_kv8 = has8;
        } else if (fs != MD_ERR_UNSUPPORTED) {
          return fs;
        }
      }
      if (!rope_done)
        MD_TRY(gemm_f8(h8, Dp, q.s_ln, q.qkv_fc1, w.qkv, qld, M, MD_EPI_GELU, nullptr, 0, 0, 1, s, ff8, qld * 2, q.s_ff, qkv_w, qkv_w));
      if (q_len > 1 && Dp == D) {
        MD_TRY(rope_and_attention(l, qld, false, h8, 1.0f / q.s_att));
      } else {
        MD_TRY(rope_and_attention(l, qld, false));
        MD_TRY(md_quantize_f8(w.att, Dp, h8, Dp, M, D, Dp, 1.0f / q.s_att, s));
      }
      MD_TRY(gemm_f8(h8, Dp, q.s_att, q.proj, x, D, M, MD_EPI_RESIDUAL, x, D, 0, 0, s));
      MD_TRY(gemm_f8(ff8, qld * 2, q.s_ff, q.fc2, x, D, M, MD_EPI_RESIDUAL, x, D, 0, 0, s));
      continue;
    }
    // l_in = ln(x)                                            (text.py:145)
    // (decode regime: blocks > 0 get it from the previous block's tail kernel)
    if (!tail_fused || l == 0) MD_TRY(md_layernorm_bf16(x, D, w.h, Dp, &b.ln, M, D, 1e-5f, s));
    MD_TRY(calib_site(calib, 3 * l + 0, w.h, Dp, M, D, s));
    // qkv, rope(q), rope(k), cache update                      (text.py:30-46)
    const bool fused = b.qkv_fc1.w != nullptr;
    const int64_t qld = fused ? b.qkv_fc1.n_pad : qkv_w;  // leading dimension of the qkv activation
    const int64_t ffld = fused ? b.qkv_fc1.n_pad : b.fc1.n_pad;
    // decode regime with FP8 weight copies attached (opt-in): the same three launches over half the bytes
    const md_text_block_fp8* f8 = (m->fp8 && m->fp8->blocks && tail_fused && fused) ? &m->fp8->blocks[l] : nullptr;
    if (f8 && f8->qkv_fc1.w) {
      MD_CHECK_ARG(f8->qkv_fc1.n_pad == b.qkv_fc1.n_pad && qkv_w % 64 == 0);
      MD_TRY(md_gemm_fp8w(w.h, Dp, &f8->qkv_fc1, w.qkv, qld, M, MD_EPI_GELU, 1, qkv_w, s));
    } else if (fused) {
      // one GEMM for both consumers of l_in: [qkv | gelu(fc1)]   (text.py:30 and layers.py:130-138)
      MD_CHECK_ARG(b.qkv_fc1.n_pad == qkv_w + b.fc1.n_pad && qkv_w % 64 == 0);
      md_gemm_args g;
      g.a = w.h; g.lda = Dp; g.lin = b.qkv_fc1; g.c = w.qkv; g.ldc = qld; g.r = nullptr; g.ldr = 0;
      g.res_row_mod = 0; g.m = M; g.epilogue = MD_EPI_GELU; g.store_pad_cols = 1; g.gelu_from_col = qkv_w;
      g.splitk_ws = w.splitk; g.splitk_ws_bytes = w.splitk_bytes; g.tile_policy = t_tile_policy;
      rope_done = false;
      if (rope_in_gemm) {
        md_rope_fuse rf;
        rf.row_cs = w.rope_cs; rf.row_kv = w.rope_kv;
        rf.kslab = (bf16_t*)kv->k + (int64_t)l * kv->layer_stride;
        rf.vslab = (bf16_t*)kv->v + (int64_t)l * kv->layer_stride;
        rf.slab_bytes = slab_bytes; rf.n_heads = m->n_heads; rf.ctx = kv->ctx;
        const md_status fs = md_gemm_qkv_rope(&g, &rf, s);
        if (fs == MD_OK) rope_done = true;
        else if (fs == MD_ERR_UNSUPPORTED) rope_in_gemm = false;  // a function of the shape: the same answer for every block
        else return fs;
      }
      if (!rope_done) MD_TRY(md_gemm_bf16(&g, s));
    } else {
      MD_TRY(gemm(w.h, Dp, b.qkv, w.qkv, qkv_w, M, MD_EPI_BIAS, nullptr, 0, 0, 0, s, w.splitk, w.splitk_bytes));
    }
    const bool fuse_rope = (q_len == 1) && (m->n_kv_heads == m->n_heads);  // decode step: rope + KV write inside attention
    MD_TRY(rope_and_attention(l, qld, fuse_rope));
    MD_TRY(calib_site(calib, 3 * l + 1, w.att, Dp, M, D, s));
    // x = (x + proj(att)) + fc2(gelu(fc1(l_in)))               (text.py:53,157-158)
    if (!fused)
      MD_TRY(gemm(w.h, Dp, b.fc1, w.ff, ffld, M, MD_EPI_GELU, nullptr, 0, 0, 1, s, w.splitk, w.splitk_bytes));
    MD_TRY(calib_site(calib, 3 * l + 2, w.ff, ffld, M, b.fc1.n, s));
    MD_CHECK_ARG(b.fc2.k_pad == b.fc1.n_pad);
    if (tail_fused) {
      // decode regime: both linears leave fp32 K-slice partials; ONE tail kernel sums them, applies
      // bias / residual with the same roundings and writes the next block's ln(x)
      const bool f8_tail = f8 && f8->proj.w && f8->fc2.w;
      int sl_a = md_gemm_partial_slices(&b.proj), sl_b = md_gemm_partial_slices(&b.fc2);
      if (f8_tail) {
        MD_CHECK_ARG(md_gemm_fp8w_partial_slices(&f8->proj) <= sl_a && md_gemm_fp8w_partial_slices(&f8->fc2) <= sl_b);  // workspace is sized for the bf16 split
        sl_a = md_gemm_fp8w_partial_slices(&f8->proj);
        sl_b = md_gemm_fp8w_partial_slices(&f8->fc2);
        MD_TRY(md_gemm_fp8w_partial_f32_pair(w.att, Dp, &f8->proj, w.part_a, w.ff, ffld, &f8->fc2, w.part_b, M, w.part_ld,
                                             w.part_stride, s));
      } else {
        MD_TRY(md_gemm_partial_f32_pair(w.att, Dp, &b.proj, w.part_a, w.ff, ffld, &b.fc2, w.part_b, M, w.part_ld,
                                        w.part_stride, s));
      }
      const bool last = (l + 1 == m->n_layers);
      MD_TRY(md_reduce_residual_layernorm(x, D, w.part_a, sl_a, b.proj.b, w.part_b,
                                          sl_b, b.fc2.b, w.part_ld, w.part_stride,
                                          last ? nullptr : w.h, Dp, last ? nullptr : &m->blocks[l + 1].ln, M, D,
                                          1e-5f, s));
    } else {
      MD_TRY(gemm(w.att, Dp, b.proj, x, D, M, MD_EPI_RESIDUAL, x, D, 0, 0, s, w.splitk, w.splitk_bytes));
      MD_TRY(gemm(w.ff, ffld, b.fc2, x, D, M, MD_EPI_RESIDUAL, x, D, 0, 0, s, w.splitk, w.splitk_bytes));
    }
  }
  return MD_OK;
}

// ------------------------------------------------------------- LoRA side path
namespace {
struct LoraWs {
  void *h, *qkv, *att, *ff, *t, *d1, *d2, *pos_kv;
  size_t total;
};
LoraWs lora_layout(const md_text_model* m, int batch, int q_len, void* base) {
  const size_t M = (size_t)batch * q_len, hd = m->dim / m->n_heads;
  Arena a{(char*)base, 0};
  LoraWs w;
  w.h = a.take(M * m->blocks[0].qkv.k_pad * 2);
  w.qkv = a.take(M * (m->n_heads + 2 * m->n_kv_heads) * hd * 2);
  w.att = a.take(M * m->blocks[0].proj.k_pad * 2);
  w.ff = a.take(M * m->blocks[0].fc1.n_pad * 2);
  w.t = a.take(M * 256 * 2);  // x A^T, rank zero-padded (<= 256)
  w.d1 = a.take(M * m->dim * 2);
  w.d2 = a.take(M * m->dim * 2);
  w.pos_kv = a.take((size_t)batch * 4);
  w.total = a.off;
  return w;
}
// c = bf16(c + (x A^T) B^T), in place on c [M][ldc]; absent pair: nothing
md_status lora_add(const md_lora_pair& lp, const void* x, int64_t ldx, void* t, void* c, int64_t ldc, int M, hipStream_t s) {
  if (lp.a.w == nullptr) return MD_OK;
  if (lp.a.n_pad > 256 || lp.b.k_pad != lp.a.n_pad) return MD_ERR_UNSUPPORTED;
  MD_TRY(gemm(x, ldx, lp.a, t, lp.a.n_pad, M, MD_EPI_BIAS, nullptr, 0, 0, 1, s));   // pad columns written as zeros
  return gemm(t, lp.a.n_pad, lp.b, c, ldc, M, MD_EPI_RESIDUAL, c, ldc, 0, 0, s);
}
}  // namespace

extern "C" size_t md_text_lora_workspace_bytes(const md_text_model* m, int32_t batch, int32_t q_len) {
  if (!m || !m->blocks || batch <= 0 || q_len <= 0) return 0;
  return lora_layout(m, batch, q_len, nullptr).total;
}

extern "C" md_status md_text_forward_lora(const md_text_model* m, const md_text_block_lora* lora, const void* x_in, void* hidden,
                                          int32_t batch, int32_t q_len, const int32_t* pos0, const md_kv_cache* kv,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  if (lora == nullptr) return md_text_forward(m, x_in, hidden, batch, q_len, pos0, kv, workspace, workspace_bytes, stream);
  MD_CHECK_ARG(m && x_in && hidden && pos0 && kv && kv->k && kv->v && workspace && m->blocks);
  MD_CHECK_ARG(batch > 0 && q_len > 0 && m->dim % m->n_heads == 0);
  MD_CHECK_ARG(tile_policy_ok(m->tile_policy));
  TilePolicyScope tile_scope(m->tile_policy);
  const int hd = m->dim / m->n_heads;
  if (hd != 64) return MD_ERR_UNSUPPORTED;
  const LoraWs w = lora_layout(m, batch, q_len, workspace);
  if (workspace_bytes < w.total) return MD_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int D = m->dim, M = batch * q_len;
  const int qkv_w = (m->n_heads + 2 * m->n_kv_heads) * hd;
  const int Dp = m->blocks[0].qkv.k_pad;
  MD_CHECK_ARG(Dp == D && m->blocks[0].proj.k_pad == D && m->blocks[0].fc1.k_pad == D);  // no K padding on this path
  bf16_t* x = (bf16_t*)hidden;
  if (x_in != hidden && hipMemcpyAsync(hidden, x_in, (size_t)M * D * 2, hipMemcpyDeviceToDevice, s) != hipSuccess) return MD_ERR_LAUNCH;
  int32_t* kv_len = (int32_t*)w.pos_kv;
  hipLaunchKernelGGL(kv_len_kernel, dim3((batch + 255) / 256), dim3(256), 0, s, pos0, kv_len, q_len, batch);
  const float scale = 1.0f / sqrtf((float)hd);
  for (int l = 0; l < m->n_layers; ++l) {
    const md_text_block& b = m->blocks[l];
    const md_text_block_lora& lo = lora[l];
    bf16_t* kl = (bf16_t*)kv->k + (int64_t)l * kv->layer_stride;
    bf16_t* vl = (bf16_t*)kv->v + (int64_t)l * kv->layer_stride;
    const int64_t ffld = b.fc1.n_pad;
    MD_CHECK_ARG(b.fc2.k_pad == b.fc1.n_pad && b.fc1.n == b.fc1.n_pad);
    MD_TRY(md_layernorm_bf16(x, D, w.h, D, &b.ln, M, D, 1e-5f, s));                                     // text.py:145
    MD_TRY(gemm(w.h, D, b.qkv, w.qkv, qkv_w, M, MD_EPI_BIAS, nullptr, 0, 0, 0, s));                      // text.py:30
    MD_TRY(lora_add(lo.qkv, w.h, D, w.t, w.qkv, qkv_w, M, s));                                          // text.py:31-32
    MD_TRY(md_rope_kv_write(w.qkv, qkv_w, m->freqs, pos0, kl, vl, kv->batch_stride, kv->ctx, batch, q_len, m->n_heads,
                            m->n_kv_heads, hd, m->rot_dim, s));
    if (q_len == 1) {
      MD_TRY(md_attention_decode(w.qkv, qkv_w, w.att, D, kl, vl, kv->batch_stride, kv->ctx, kv_len, batch, m->n_heads,
                                 m->n_kv_heads, hd, scale, s));
    } else {
      md_attn_args a = {};
      a.q = w.qkv; a.q_bs = (int64_t)q_len * qkv_w; a.q_ts = qkv_w; a.q_hs = hd;
      a.k = kl; a.v = vl; a.k_bs = a.v_bs = kv->batch_stride; a.k_ts = a.v_ts = hd; a.k_hs = a.v_hs = (int64_t)kv->ctx * hd;
      a.o = w.att; a.o_bs = (int64_t)q_len * D; a.o_ts = D; a.o_hs = hd;
      a.batch = batch; a.n_heads = m->n_heads; a.n_kv_heads = m->n_kv_heads; a.head_dim = hd; a.q_len = q_len;
      a.kv_len_all = 0; a.q_pos0 = pos0; a.kv_len = kv_len; a.prefix_len = m->prefix_len; a.scale = scale;
      MD_TRY(md_attention_prefill(&a, s));
    }
    MD_TRY(gemm(w.att, D, b.proj, w.d1, D, M, MD_EPI_BIAS, nullptr, 0, 0, 0, s));                        // text.py:53
    MD_TRY(lora_add(lo.proj, w.h, D, w.t, w.d1, D, M, s));                                               // text.py:55: x = l_in
    MD_TRY(gemm(w.h, D, b.fc1, w.ff, ffld, M, MD_EPI_BIAS, nullptr, 0, 0, 1, s));                        // layers.py:130
    MD_TRY(lora_add(lo.fc1, w.h, D, w.t, w.ff, ffld, M, s));                                             // layers.py:131-133
    MD_TRY(md_gelu_bf16(w.ff, ffld, w.ff, ffld, M, (int32_t)ffld, s));                                   // layers.py:137
    MD_TRY(gemm(w.ff, ffld, b.fc2, w.d2, D, M, MD_EPI_BIAS, nullptr, 0, 0, 0, s));                       // layers.py:139
    MD_TRY(lora_add(lo.fc2, w.ff, ffld, w.t, w.d2, D, M, s));                                            // layers.py:140-142
    MD_TRY(md_add_bf16(x, D, w.d1, D, x, D, M, D, s));                                                   // text.py:158
    MD_TRY(md_add_bf16(x, D, w.d2, D, x, D, M, D, s));
  }
  return MD_OK;
}

extern "C" size_t md_lm_head_workspace_bytes(const md_text_model* m, int32_t batch) {
  if (!m || batch <= 0) return 0;
  return align_up((size_t)batch * m->lm_head.k_pad * 2);
}

// reference: text.py:163-167
extern "C" md_status md_lm_head(const md_text_model* m, const void* hidden, int32_t batch, int32_t q_len,
                                void* logits, int64_t ld_logits, void* workspace,
                                size_t workspace_bytes, void* stream) {
  MD_CHECK_ARG(m && hidden && logits && workspace && batch > 0 && q_len > 0);
  MD_CHECK_ARG(tile_policy_ok(m->tile_policy));
  TilePolicyScope tile_scope(m->tile_policy);
  if (workspace_bytes < md_lm_head_workspace_bytes(m, batch)) return MD_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int D = m->dim;
  const bf16_t* last = (const bf16_t*)hidden + (int64_t)(q_len - 1) * D;
  const int Dp = m->lm_head.k_pad;
  MD_TRY(zero_if_padded(workspace, batch, Dp, D, s));
  MD_TRY(md_layernorm_bf16(last, (int64_t)q_len * D, workspace, Dp, &m->post_ln, batch, D, 1e-5f, s));
  if (m->fp8 && m->fp8->lm_head.w && batch <= 64)
    return md_gemm_fp8w(workspace, Dp, &m->fp8->lm_head, logits, ld_logits, batch, MD_EPI_BIAS, 0, 0, s);
  return gemm(workspace, Dp, m->lm_head, logits, ld_logits, batch, MD_EPI_BIAS, nullptr, 0, 0, 0, s);
}

extern "C" size_t md_decode_workspace_bytes(const md_text_model* m, int32_t batch) {
  if (!m || !m->blocks || batch <= 0) return 0;
  return align_up((size_t)batch * m->dim * 2) + md_lm_head_workspace_bytes(m, batch) +
         md_text_workspace_bytes(m, batch, 1);
}

// reference: the generator loop body of moondream.py:512-530, device resident
extern "C" md_status md_decode_step(const md_text_model* m, const int32_t* tokens, int32_t* next,
                                    int32_t* pos, int32_t batch, const md_kv_cache* kv,
                                    int32_t suppress_id, void* logits, int64_t ld_logits,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  MD_CHECK_ARG(m && tokens && next && pos && kv && logits && workspace && batch > 0);
  if (workspace_bytes < md_decode_workspace_bytes(m, batch)) return MD_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  Arena a{(char*)workspace, 0};
  void* x = a.take((size_t)batch * m->dim * 2);
  void* lmws = a.take(md_lm_head_workspace_bytes(m, batch));
  void* tws = a.take(0);
  const size_t tws_bytes = workspace_bytes - a.off;
  MD_TRY(md_embed_tokens(tokens, m->wte, m->dim, x, m->dim, batch, m->dim, s));
  MD_TRY(md_text_forward(m, x, x, batch, 1, pos, kv, tws, tws_bytes, s));
  if (decode_tall_rows(m, batch, 1) && !(kv->k8 && kv->v8)) {
    // the step's lm_head at 65 .. 128 rows: the by-shape config of the same MFMA family as the <= 64-row regime (never the
    // pinned four-wave kernel: a sequence must get the same logits in a step of 128 as in a step of 64)
    md_text_model tallm = *m;
    tallm.tile_policy = MD_TILE_DECODE_TALL;
    MD_TRY(md_lm_head(&tallm, x, batch, 1, logits, ld_logits, lmws, md_lm_head_workspace_bytes(m, batch), s));
  } else {
    MD_TRY(md_lm_head(m, x, batch, 1, logits, ld_logits, lmws, md_lm_head_workspace_bytes(m, batch), s));
  }
  return md_argmax_advance(logits, ld_logits, batch, m->vocab, suppress_id, next, pos, s);
}

extern "C" size_t md_decode_step_b1_workspace_bytes(const md_text_model* m) {
  if (!m || !m->blocks) return 0;
  return align_up(md_decode_b1_workspace_bytes(m));
}

// decode_b1.hip: embedding lookup, every decoder block, final layer norm, lm_head, suppression, argmax and pos += 1 in ONE launch
md_status md_decode_b1_step(const md_text_model* m, const int32_t* token, int32_t* next, int32_t* pos, const md_kv_cache* kv,
                            int32_t suppress_id, void* logits, void* workspace, size_t workspace_bytes, void* sync_state,
                            hipStream_t s);

extern "C" md_status md_decode_step_b1(const md_text_model* m, const int32_t* token, int32_t* next, int32_t* pos,
                                       const md_kv_cache* kv, int32_t suppress_id, void* logits, int64_t ld_logits,
                                       void* workspace, size_t workspace_bytes, void* sync_state, void* stream) {
  MD_CHECK_ARG(m && token && next && pos && kv && logits && workspace && sync_state && ld_logits >= m->vocab);
  if (workspace_bytes < md_decode_step_b1_workspace_bytes(m)) return MD_ERR_WORKSPACE;
  return md_decode_b1_step(m, token, next, pos, kv, suppress_id, logits, workspace, workspace_bytes, sync_state, (hipStream_t)stream);
}
