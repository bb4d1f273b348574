/*
 * moondream_hip.h -- C ABI of libmoondream_hip.so, the MI355X (gfx950) compute
 * library behind the Moondream inference seam.
 *
 * The reference isolates its accelerated path behind four bound methods that
 * it rebinds itself in compile() (reference: moondream/torch/moondream.py:168-204):
 *
 *     _vis_enc(x)                           moondream.py:168-169  -> md_vit_encode
 *     _vis_proj(g, r)                       moondream.py:171-172  -> md_vision_project
 *     _prefill(x, attn_mask, pos_ids, lora) moondream.py:174-181  -> md_text_forward
 *     _decode_one_tok(x, mask, pos, lora)   moondream.py:183-192  -> md_text_forward + md_lm_head
 *
 * Conventions
 *   - plain C: device pointers travel as void*, the HIP stream as void*
 *     (a hipStream_t); no torch / C++ types cross this boundary.
 *   - the caller owns every buffer (weights, KV slabs, activations, workspace);
 *     nothing here allocates, frees or synchronises.  All work is enqueued on
 *     the caller's stream, so calls are hipGraph-capturable.
 *   - bf16 tensors are raw 16-bit words, row-major, with an explicit leading
 *     dimension in ELEMENTS.
 *   - every entry point returns an md_status; 0 is success.  There is no CPU
 *     fallback: without a gfx950 device the launches fail and say so.
 */
#ifndef MOONDREAM_HIP_H
#define MOONDREAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MD_ABI_VERSION 5 /* 2: md_text_model.fp8 (trailing, optional); 3: md_vit_model.f8, md_text_model.f8 (trailing, optional), md_gemm_f8 + fp8 producers, md_decode_step_b1_supported; 4: md_linear_fp8.format (int4 group-128 weight stream; a LAYOUT BREAK: md_linear_fp8 grew 40 -> 48 bytes, which moves every md_text_block_fp8 member); 5: tile_policy in md_gemm_args / md_vit_model / md_text_model (per call; replaces the process-wide "strict" tuning key).  Every version is a layout break for some struct: callers MUST compare md_abi_version() with the header they were built against */

typedef int md_status;
enum {
  MD_OK = 0,
  MD_ERR_INVALID_ARG = 1, /* shape / alignment / null-pointer contract violated */
  MD_ERR_LAUNCH = 2,      /* the HIP runtime rejected a launch (no gfx950 device, ...) */
  MD_ERR_WORKSPACE = 3,   /* workspace smaller than md_*_workspace_bytes() */
  MD_ERR_UNSUPPORTED = 4  /* a shape outside what the kernels are built for */
};

int md_abi_version(void);
const char* md_status_string(md_status s);

/* ------------------------------------------------------------------ *
 * Dense layers                                                        *
 * ------------------------------------------------------------------ */

/* y = x W^T + b in the nn.Linear layout (reference: layers.py:34-35).
 * w: bf16 [n_pad][k_pad] row-major, packed once by the host: k_pad = k rounded
 * up to 64 with zero fill; rows n..n_pad-1 (n_pad = n rounded up to 64) zero.
 * b: bf16 [n_pad] (zero-filled tail) or NULL. */
typedef struct {
  const void* w;
  const void* b;
  int32_t n;     /* logical out features */
  int32_t k;     /* logical in features  */
  int32_t n_pad; /* allocated rows       */
  int32_t k_pad; /* allocated row length = leading dimension */
} md_linear;

/* LayerNorm affine parameters, bf16 [dim] (reference: layers.py:118-119). */
typedef struct {
  const void* w;
  const void* b;
} md_layernorm;

enum {
  MD_EPI_BIAS = 0,     /* c = bf16(acc + b)                                       */
  MD_EPI_GELU = 1,     /* c = bf16(gelu_tanh(bf16(acc + b)))    layers.py:129-138 */
  MD_EPI_RESIDUAL = 2  /* c = bf16(r + bf16(acc + b))           vision.py:68,70-71, text.py:158 */
};

/* C[m, n] = epilogue(A[m, :k_pad] . W[n, :k_pad] + b[n]), fp32 accumulate on
 * MFMA, one bf16 rounding of (acc + bias) and one more for the epilogue op --
 * the reference's rounding points.  A must have lda >= k_pad with columns
 * k..k_pad-1 readable and ZERO.  r (MD_EPI_RESIDUAL) is read at row
 * (res_row_mod ? row % res_row_mod : row); it may alias c.  Columns n..n_pad-1
 * of C are written (zeros through the zero weight rows) when ldc >= n_pad,
 * which is what lets a padded activation feed the next layer's k_pad. */
typedef struct {
  const void* a;
  int64_t lda;
  md_linear lin;
  void* c;
  int64_t ldc;
  const void* r;
  int64_t ldr;
  int32_t res_row_mod;
  int32_t m;
  int32_t epilogue;
  int32_t store_pad_cols; /* 1: also store columns [n, n_pad) */
  int32_t gelu_from_col;  /* MD_EPI_GELU only: GELU applies to columns >= this (multiple of 8; 0 = all) */
  /* Scratch for the decode regime (m <= 64), where K is split over several
   * workgroups per output tile: md_gemm_workspace_bytes() bytes, or NULL (then K
   * is not split across workgroups: slower, and a different -- still
   * deterministic -- summation tree).  Its first 8 KiB are arrival tickets:
   * they must be ZERO before the first launch that uses the buffer; every launch
   * leaves them zero, so launches that follow each other on one stream can share
   * the buffer. */
  void* splitk_ws;
  size_t splitk_ws_bytes;
  int32_t tile_policy; /* MD_TILE_BY_SHAPE / MD_TILE_PINNED (below); ABI 5 */
} md_gemm_args;

/* Tile choice of a launch with more than 64 rows -- a PER-CALL property (ABI 5; up to ABI 4 a process-wide tuning key):
 *   MD_TILE_BY_SHAPE  the config that is fastest for (m, n, k): small launches (one image: 730 / 1458 rows) take the
 *                     64x64 / 128x128 / 256x128 configs (32x32x16 MFMAs), large ones the four-wave 256x256 kernel
 *                     (16x16x32 MFMAs).  The two families sum K in different associations: equal to fp32 rounding of the
 *                     accumulator, not bitwise -- so a row's bits can depend on how many rows travel with it.
 *   MD_TILE_PINNED    the tile config is a function of the layer (n, k) alone, never of m: every launch of more than 64
 *                     rows takes the 256x256 kernel, so a sequence gets the same bits alone and in any batch.  That
 *                     kernel addresses a launch with 32-bit byte offsets (m * lda * 2 < 4 GiB, (m + 256) * ldc * 2 <
 *                     0xfffff000: ~150 000 rows of the 2B fused qkv|fc1 layer); a larger launch is cut into row blocks
 *                     of the same kernel (same bits), and one that cannot be cut -- a broadcast residual
 *                     (res_row_mod != 0), the RoPE epilogue -- returns MD_ERR_UNSUPPORTED instead of changing family.
 *   MD_TILE_DECODE_TALL  (round 6; set by md_decode_step / md_text_forward themselves for a decode step of 65 .. 128 sequences, accepted
 *                     from a caller of md_gemm_bf16 too) a launch of 65 .. 128 rows is a WEIGHT STREAM like the <= 64-row regime: one
 *                     128 x 64 tile per weight panel (four compute waves + two DMA waves), the same K order per output element as
 *                     the 64-row configs -- a sequence gets the same bits in a decode step of 128 as in one of 64; layers of
 *                     >= 16384 output columns (lm_head) take the by-shape config (same MFMA family, same K order).  Other row
 *                     counts: as MD_TILE_BY_SHAPE.
 * md_vit_model.tile_policy / md_text_model.tile_policy apply it to every GEMM of md_vit_encode / md_vision_project* /
 * md_text_forward* / md_lm_head / md_decode_step made with that struct.  Launches of <= 64 rows (the decode regime) are
 * not affected: they always take the split-K weight-streaming configs. */
enum { MD_TILE_BY_SHAPE = 0, MD_TILE_PINNED = 1, MD_TILE_DECODE_TALL = 2 };

md_status md_gemm_bf16(const md_gemm_args* args, void* stream);
size_t md_gemm_workspace_bytes(const md_linear* lin, int32_t m, int32_t store_pad_cols);

/* Decode regime (m <= 64), launch-boundary split-K: instead of combining the K slices inside
 * the launch (tickets + an agent-scope release/acquire per tile), every slice stores its
 * fp32 partial products, partial[s][row][ld_partial] for s < md_gemm_partial_slices(lin)
 * (slice_stride floats apart; no bias), and the consumer kernel below sums them in slice
 * order.  Used for the two linears that feed the residual stream in a decode step
 * (proj: text.py:53, fc2: layers.py:139). */
int32_t md_gemm_partial_slices(const md_linear* lin);
md_status md_gemm_partial_f32(const void* a, int64_t lda, const md_linear* lin, int32_t m, float* partial,
                              int64_t ld_partial, int64_t slice_stride, void* stream);
/* The same for two independent layers over the same rows in ONE launch (a decode block's proj and
 * fc2): partial0 / partial1 get md_gemm_partial_slices(lin0) / (lin1) slices. */
md_status md_gemm_partial_f32_pair(const void* a0, int64_t lda0, const md_linear* lin0, float* partial0,
                                   const void* a1, int64_t lda1, const md_linear* lin1, float* partial1,
                                   int32_t m, int64_t ld_partial, int64_t slice_stride, void* stream);

/* Block tail of a decode step, one launch (text.py:53,157-158 and the next block's text.py:145):
 *   x = bf16(bf16(x + bf16(sum_s A[s] + bias_a)) + bf16(sum_s B[s] + bias_b));  y = layer_norm(x)
 * -- the same roundings, in the same order, as two md_gemm_bf16 MD_EPI_RESIDUAL launches
 * followed by md_layernorm_bf16.  y == NULL skips the layer norm (last block). */
md_status md_reduce_residual_layernorm(void* x, int64_t ldx, const float* partial_a, int32_t slices_a,
                                       const void* bias_a, const float* partial_b, int32_t slices_b,
                                       const void* bias_b, int64_t ld_partial, int64_t slice_stride,
                                       void* y, int64_t ldy, const md_layernorm* ln, int32_t rows,
                                       int32_t dim, float eps, void* stream);

/* ---- FP8 weights for the decode regime (opt-in numerical mode; BASELINE configs[4]) -------------------------------
 * w: OCP e4m3fn bytes in MFMA-fragment order: block (nb, kb) of 32 channels x 32 features is 1 KiB at
 * ((nb * k_pad / 32 + kb) * 1024); inside it lane l (0..63) owns 16 bytes: channel 32 nb + (l & 31), bytes 0..7 =
 * features 32 kb + 8 (l >> 5) + j, bytes 8..15 = features 32 kb + 16 + 8 (l >> 5) + j.  scale: fp32 [n_pad],
 * weight[n][k] ~= scale[n] * fp8[n][k].  b: bf16 [n_pad] or NULL.  n_pad % 64 == 0, k_pad % 128 == 0, zero padded.
 * The same op as md_gemm_bf16 at m <= 64 (F.linear of text.py:30,53, layers.py:130,139, text.py:166 at q_len 1):
 * bf16 activations, exact fp8 -> bf16 products, fp32 accumulation, epilogue(scale * acc + bias) with the bf16
 * kernels' rounding points.  MD_EPI_BIAS / MD_EPI_GELU (GELU on columns >= gelu_from_col). */
/* format MD_WSTREAM_INT4_G128 (ABI 4): the reference's own quantised checkpoint format as the weight stream (layers.py:38-74,
 * QuantizedLinear: 4-bit weights in groups of 128 input features, one (scale, zero_point) pair per group).  w: nibbles in
 * MFMA-fragment order: block (nb, step, kh) of 32 channels x 64 features (the kh-th half of the 128-wide K step = of the
 * group) is 1 KiB at (((nb * k_pad / 128 + step) * 2 + kh) * 1024); lane l owns 16 bytes: channel 32 nb + (l & 31), dword t =
 * features 128 step + 64 kh + 16 t + 8 (l >> 5) + j as nibble j (bits 4 j .. 4 j + 3).  scale: fp32 pairs (scale, zero_point)
 * at [k_pad / 128][n_pad].  Every weight is rebuilt as bf16(bf16(q - zero_point) * scale), the reference's dequantize_tensor
 * (layers.py:38-44) with its two roundings: the operand is bit for bit the bf16 weight the bf16 path multiplies with, so this is
 * NOT a numerical mode -- the same layer at a quarter of the weight bytes.  k_pad == k (k % 128 == 0). */
#define MD_WSTREAM_E4M3 0
#define MD_WSTREAM_INT4_G128 1
typedef struct {
  const void* w;
  const float* scale;
  const void* b;
  int32_t n, k, n_pad, k_pad;
  int32_t format; /* MD_WSTREAM_E4M3 | MD_WSTREAM_INT4_G128.  Added in ABI 4 -- a LAYOUT BREAK: the struct grew from 40 to 48
                   * bytes, which also moves every member of md_text_block_fp8; a caller built against ABI <= 3 does not pass
                   * 0 here, it passes garbage.  Only the md_abi_version() check protects against that. */
} md_linear_fp8;
md_status md_gemm_fp8w(const void* a, int64_t lda, const md_linear_fp8* lin, void* c, int64_t ldc, int32_t m,
                       int32_t epilogue, int32_t store_pad_cols, int32_t gelu_from_col, void* stream);
/* K-slice partial products of two layers in one launch, the fp8 form of md_gemm_partial_f32_pair (already scaled,
 * no bias); md_gemm_fp8w_partial_slices(lin) slices each. */
int32_t md_gemm_fp8w_partial_slices(const md_linear_fp8* lin);
md_status md_gemm_fp8w_partial_f32_pair(const void* a0, int64_t lda0, const md_linear_fp8* lin0, float* partial0,
                                        const void* a1, int64_t lda1, const md_linear_fp8* lin1, float* partial1,
                                        int32_t m, int64_t ld_partial, int64_t slice_stride, void* stream);

/* ---- FP8 operands for the MFMA-bound linears (opt-in numerical mode; BASELINE configs[4] "CDNA4 fp8 MFMA") ---------
 * Both operands OCP e4m3fn, ROW-MAJOR with K contiguous (the layout v_mfma_f32_32x32x64_f8f6f4 consumes: a lane's
 * operand is 32 consecutive bytes of a row), fp32 accumulation:
 *   value[m][n] = a_scale * scale[n] * sum_k a8[m][k] * w8[n][k] + b[n]
 * then the bf16 kernels' rounding points (ONE rounding of value to bf16, then GELU / residual add).  The reference has
 * no fp8 path (its only quantised format is int4, layers.py:38-109): this mode is judged by tolerance against the
 * bf16 path, never by bit parity.
 * w: [n_pad][k_pad] bytes, zero padded, k_pad % 64 == 0, n_pad % 64 == 0; scale: fp32 [n_pad] (weight[n][k] ~=
 * scale[n] * fp8[n][k]); b: bf16 [n_pad] or NULL. */
typedef struct {
  const void* w;
  const float* scale;
  const void* b;
  int32_t n, k, n_pad, k_pad;
} md_linear_f8;

typedef struct {
  const void* a;      /* e4m3fn [m][lda] bytes, lda >= k_pad, lda % 16 == 0, columns k..k_pad-1 ZERO (0x00) */
  int64_t lda;
  float a_scale;      /* activation[m][k] ~= a_scale * fp8 (one scale per tensor) */
  md_linear_f8 lin;
  void* c;            /* bf16 result [m][ldc] (columns < f8_from_col when c8 is given); may be NULL if everything goes to c8 */
  int64_t ldc;
  void* c8;           /* optional e4m3fn result of columns >= f8_from_col: c8[m][n - f8_from_col] = fp8(sat(value * c8_inv_scale)) */
  int64_t ldc8;       /* bytes between rows of c8 (% 8 == 0) */
  float c8_inv_scale;
  int32_t f8_from_col; /* multiple of 64 */
  const void* r;      /* MD_EPI_RESIDUAL operand (bf16), may alias c */
  int64_t ldr;
  int32_t res_row_mod;
  int32_t m;
  int32_t epilogue;       /* MD_EPI_* */
  int32_t store_pad_cols; /* 1: also store columns [n, n_pad) (zeros) */
  int32_t gelu_from_col;  /* MD_EPI_GELU: GELU applies to columns >= this */
} md_gemm_f8_args;
md_status md_gemm_f8(const md_gemm_f8_args* args, void* stream);

/* bf16 [rows][ldx] -> e4m3fn [rows][ldy bytes]: y = fp8(sat(x * inv_scale)) for columns < cols, 0x00 for columns
 * cols..cols_pad-1 (the consumer's K padding).  cols % 8 == 0, cols_pad % 8 == 0. */
md_status md_quantize_f8(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t cols, int32_t cols_pad,
                         float inv_scale, void* stream);
/* md_layernorm_bf16 with the result rounded to bf16 (the reference's rounding point) and then quantised like
 * md_quantize_f8; columns dim..dim_pad-1 of y are written as zero. */
md_status md_layernorm_f8(const void* x, int64_t ldx, void* y, int64_t ldy, const md_layernorm* p, int32_t rows,
                          int32_t dim, int32_t dim_pad, float eps, float inv_scale, void* stream);
/* *amax = max(*amax, max |x[r][c]|) over rows x cols (calibration of the activation scales; *amax is device memory the
 * caller zeroes; non-finite values are ignored). */
md_status md_amax_bf16(const void* x, int64_t ldx, int32_t rows, int32_t cols, float* amax, void* stream);

/* Measurement / test hook, not needed by the product path: overrides one of the GEMM dispatch
 * knobs at run time (the same knobs are read once from MD_GEMM_* / MD_DECODE_* environment
 * variables at first use).  PROCESS-WIDE and not thread-safe: for sweeps and A/B tests only; nothing that decides the
 * bits of a product call lives here (that is md_gemm_args.tile_policy, per call).  The complete key list:
 *   "tile"          -1 = automatic; 20 = four-wave 256x256, 11 / 15 = the eight-wave 256x256 baselines, 1 = 256x128,
 *                   2 = 128x128, 16 / 10 / 3 = decode-regime configs (forces the config for every launch)
 *   "small_m_rule"  tile rule of MD_TILE_BY_SHAPE launches: 1 (default) = round 5's (the 128x128 config for up to 512 tiles
 *                   of 128x128), 0 = round 2's (the cost model above 128 such tiles), n > 1 = n tiles instead of 512; same
 *                   bits either way
 *   "w4"            0: the eight-wave 256x256 kernels wherever the four-wave one would be picked (also under
 *                   MD_TILE_PINNED)
 *   "persist"       0: eight-wave kernels without their persistent tile loop
 *   "group_m"       row panels per tile-order group (0 = by shape)
 *   "decode_nt"     1: decode-regime weights streamed non-temporally
 *   "decode_cfg"    16 (default) / 10 / 3: decode-regime tile config
 *   "decode_slices" K slices per decode-regime tile (0 = by shape)
 *   "rope_fuse"     0: prefill RoPE + KV write as their own kernel instead of the qkv GEMM's epilogue
 *   "attn_skip_dead"  (not a GEMM key; the library's one tuning entry point) exact work skipping of md_attention_prefill:
 *                   bit 0 = the second 32-key half of a last key tile with no live key in it, bit 1 = waves with no live
 *                   query row; default 3, 0 = round 4's kernel; same bits whatever the value
 *   "w4_grid"       workgroups of the four-wave kernel's persistent grid (0 = one per CU)
 *   "w4_variant"    main-loop schedule variant of the four-wave kernel (0 = shipped; others: tools/sweep_w4_variants.py)
 *   "w4_dbg_*"      in-kernel cycle stamps of the four-wave kernel (tools/w4_probe.py)
 * Every tile config of one MFMA family accumulates K in the same order, so within a family outputs do not depend on
 * these.  Unknown key: MD_ERR_INVALID_ARG. */
md_status md_gemm_set_tuning(const char* key, int32_t value);

/* Live timing of the GEMM launches for the roofline report: while enabled,
 * md_gemm_bf16 brackets every launch with HIP events on the caller's stream.
 * md_profile_gemm(0|1) also resets the log.  md_profile_gemm_read waits for the
 * recorded events of one kernel family and returns its summed ALGORITHMIC work,
 * kernel time in ms and launch count: kind 0 = the MFMA tile kernel (m > 64),
 * work = flops 2 m n k; kind 1 = the decode-regime weight-streaming kernel
 * (m <= 64), work = weight bytes 2 n k (logical n, k).  Not for use inside
 * hipGraph capture. */
void md_profile_gemm(int32_t enable);
md_status md_profile_gemm_read(int32_t kind, double* work, double* ms, int64_t* launches);
/* Algorithmic HBM bytes of the logged launches of one family: operands read once (A rows x k_pad,
 * W n_pad x k_pad, residual rows) and results written once -- the yard-stick for the PMC traffic
 * counters (FETCH_SIZE / WRITE_SIZE) of the same step. */
md_status md_profile_gemm_bytes(int32_t kind, double* read_bytes, double* written_bytes);

/* y[r, :dim] = LN(x[r, :dim]) * w + b, fp32 statistics, eps as given
 * (reference: layers.py:118-119, default eps 1e-5).  dim % 8 == 0, dim <= 4096. */
md_status md_layernorm_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, const md_layernorm* p,
                            int32_t rows, int32_t dim, float eps, void* stream);

/* ------------------------------------------------------------------ *
 * Vision front end                                                    *
 * ------------------------------------------------------------------ */

/* uint8 HWC crops [n][crop][crop][3] -> bf16 patch rows [n*grid*grid][ld_out]
 * with feature order (c, py, px), columns >= 3*patch*patch zero-filled.  lut is
 * the 256-entry bf16 table of the reference's pixel normalisation
 * (vision.py:33-40 followed by create_patches, vision.py:44-61). */
md_status md_patchify_u8(const void* crops_u8, const void* lut_bf16, void* out, int64_t ld_out,
                         int32_t n_crops, int32_t crop, int32_t patch, void* stream);

/* same from the already-normalised bf16 CHW tensor the reference's _vis_enc
 * receives (moondream.py:168-169; create_patches vision.py:44-61). */
md_status md_patchify_bf16(const void* crops_bf16_chw, void* out, int64_t ld_out, int32_t n_crops,
                           int32_t crop, int32_t patch, void* stream);

/* ------------------------------------------------------------------ *
 * Attention                                                           *
 * ------------------------------------------------------------------ */

/* Tiled (flash-style) softmax attention on MFMA for a block of queries:
 *   ViT encoder, no mask (layers.py:163) and the decoder prefill with the
 *   prefix-LM rule (text.py:48-50 with the mask of moondream.py:138-146):
 *   key j is visible to the query at position p iff j <= p or (p < prefix and
 *   j < prefix); only keys [0, kv_len) exist.
 * Element (b, t, h, d) of q/o lives at base + b*bs + t*ts + h*hs + d; k/v alike
 * (strides in elements).  head_dim is 64 or 72.  q_pos0[b] (device int32, or
 * NULL for 0) is the position of query row 0; kv_len[b] (device int32, or NULL
 * for kv_len_all) the number of valid keys. */
typedef struct {
  const void* q;
  int64_t q_bs, q_ts, q_hs;
  const void* k;
  int64_t k_bs, k_ts, k_hs;
  const void* v;
  int64_t v_bs, v_ts, v_hs;
  void* o;
  int64_t o_bs, o_ts, o_hs;
  int32_t batch, n_heads, n_kv_heads, head_dim;
  int32_t q_len;      /* query rows per batch element */
  int32_t kv_len_all; /* used when kv_len == NULL */
  const int32_t* q_pos0;
  const int32_t* kv_len;
  int32_t prefix_len; /* bidirectional prefix; >= kv_len disables the causal rule */
  float scale;        /* 1/sqrt(head_dim) */
  /* opt-in FP8 mode (md_gemm_f8's A operand), trailing since ABI 3: when o8 != NULL the output row is ALSO written as OCP
   * e4m3 bytes -- element (b, t, h, d) at o8 + b * o8_bs + t * o8_ts + h * head_dim + d, the value being the bf16-rounded
   * output times o8_inv_scale, exactly what md_quantize_f8 makes of the bf16 output -- and o may then be NULL */
  void* o8;
  int64_t o8_bs, o8_ts; /* bytes */
  float o8_inv_scale;
} md_attn_args;

md_status md_attention_prefill(const md_attn_args* args, void* stream);

/* One query per (batch, head) against the KV slab: the decode step
 * (text.py:48-50 with the [1,1,2048] mask of moondream.py:472-474 == keys
 * [0, pos]).  q, o: [batch][n_heads*64]; slabs [batch][n_kv_heads][ctx][64];
 * kv_len[b] valid keys (device int32). */
md_status md_attention_decode(const void* q, int64_t ldq, void* o, int64_t ldo, const void* k_slab,
                              const void* v_slab, int64_t slab_batch_stride, int32_t ctx, const int32_t* kv_len,
                              int32_t batch, int32_t n_heads, int32_t n_kv_heads, int32_t head_dim,
                              float scale, void* stream);

/* The decode step's RoPE + KV-slab write + attention in ONE launch (MHA, head_dim 64):
 * qkv rows are the UN-rotated fused activation [q heads | k heads | v heads]; each
 * (sequence, head) workgroup rotates its q and k (rope.py:20-48), stores rotated k and v
 * at slot kv_len[b]-1 (moondream.py:74-78) and attends over keys [0, kv_len[b])
 * (text.py:48-50).  Bit-identical to md_rope_kv_write followed by md_attention_decode. */
md_status md_attention_decode_rope(const void* qkv, int64_t ld, void* o, int64_t ldo, const float* freqs,
                                   void* k_slab, void* v_slab, int64_t slab_batch_stride, int32_t ctx,
                                   const int32_t* kv_len, int32_t batch, int32_t n_heads, int32_t head_dim,
                                   int32_t rot_dim, float scale, void* stream);

/* Partial RoPE + KV-cache write (reference: rope.py:20-48, text.py:42-46,
 * moondream.py:74-78).  qkv: bf16 [batch*q_len][ld] rows laid out q|k|v.  The
 * first rot_dim features of every q and k head are read half-split, rotated in
 * fp32 with the fp32 table freqs[pos][rot_dim/2][2] (cos, sin) and written
 * INTERLEAVED; q is rewritten in place, rotated k and v go to
 * slab[b][h][pos][:].  pos = pos0[b] + t (pos0 device int32 [batch]). */
md_status md_rope_kv_write(void* qkv, int64_t ld, const float* freqs, const int32_t* pos0,
                           void* k_slab, void* v_slab, int64_t slab_batch_stride, int32_t ctx,
                           int32_t batch, int32_t q_len, int32_t n_heads, int32_t n_kv_heads,
                           int32_t head_dim, int32_t rot_dim, void* stream);

/* ------------------------------------------------------------------ *
 * Small ops                                                           *
 * ------------------------------------------------------------------ */

/* out[i, :dim] = table[ids[i], :dim]  (reference: text.py:12-13). */
md_status md_embed_tokens(const int32_t* ids, const void* table, int64_t ld_table, void* out,
                          int64_t ld_out, int32_t n, int32_t dim, void* stream);

/* next[b] = argmax_v logits[b, v] with logits[b, suppress_id] treated as -inf
 * when suppress_id >= 0 (reference: moondream.py:517,521-524); ties -> lowest
 * index (torch.argmax on CPU). */
md_status md_argmax_bf16(const void* logits, int64_t ld, int32_t batch, int32_t vocab,
                         int32_t suppress_id, int32_t* next, void* stream);

/* Temperature / top-p sampling of one token per sequence, device resident (reference:
 * moondream.py:521-528 with _apply_top_p, moondream.py:270-278):
 *   p = bf16(softmax(bf16(logits / temperature)));  in descending-p order keep the tokens whose
 *   preceding mass is <= top_p;  q = bf16(p / bf16(sum kept));  next[b] ~ q by inverse CDF over
 *   token ids with the caller's uniforms[b] in [0, 1)  (torch.multinomial draws from the same q;
 *   the random stream differs, the distribution does not).
 * logits[b, suppress_id] is treated as -inf when suppress_id >= 0 (moondream.py:517).
 * probs_out (optional, bf16 [batch][ld_probs]) receives q -- what _apply_top_p returns. */
md_status md_sample_top_p(const void* logits, int64_t ld, int32_t batch, int32_t vocab, int32_t suppress_id,
                          float temperature, float top_p, const float* uniforms, int32_t* next,
                          void* probs_out, int64_t ld_probs, void* stream);

/* Region head, device resident (reference: region.py:12-71 inside the loop of moondream.py:653-733).
 * md_fourier_features: out[r] = [cos(f) | sin(f)], f = bf16(bf16(2 pi x[r, :in_dim]) . w[in_dim][half])
 *   (region.py:12-29; in_dim 1 for coordinates, 2 for sizes) -- the input of coord_encoder /
 *   size_encoder.
 * md_region_pick_encode: one step of the points loop for `batch` sequences with no host round trip:
 *   bins[b][g] = argmax of logits[b][g*n_bins .. (g+1)*n_bins) (ties -> lowest index), value =
 *   value_table[bin] (bf16 [n_bins]: bin/1024 for coordinates, 2^(bin/1023*10-10) for sizes --
 *   moondream.py:673-674,696-701), then the Fourier features of the value(s) as above. */
md_status md_fourier_features(const void* x, int64_t ldx, int32_t rows, int32_t in_dim, const void* w,
                              int32_t half, void* out, int64_t ld_out, void* stream);
md_status md_region_pick_encode(const void* logits, int64_t ld, int32_t batch, int32_t n_groups,
                                int32_t n_bins, const void* value_table, const void* feat_w, int32_t half,
                                int32_t* bins, int64_t ld_bins, void* feats, int64_t ld_feats, void* stream);

/* Overlap-crop stitch + adaptive average pool + concat with the global crop's
 * features (reference: moondream.py:213-226, image_crops.py:170-231 with
 * patch_size=1, vision.py:83-88).  feats: bf16 [1 + th*tw][g*g][dim] for ONE
 * image (crop 0 = global).  out: bf16 [g*g][ld_out] = [global | pooled]. */
md_status md_stitch_pool_concat(const void* feats, void* out, int64_t ld_out, int32_t dim,
                                int32_t grid, int32_t margin, int32_t tiles_h, int32_t tiles_w,
                                void* stream);

/* ------------------------------------------------------------------ *
 * The seam: whole-stage entry points                                  *
 * ------------------------------------------------------------------ */

typedef struct {
  md_layernorm ln1;
  md_linear qkv, proj;
  md_layernorm ln2;
  md_linear fc1, fc2;
} md_vit_block;

/* Optional FP8 mode of the vision path (md_gemm_f8 above): e4m3 copies of every block's four linears and of the
 * projector MLP, and ONE static scale per quantised activation tensor (s_*: activation ~= s * fp8), found by a
 * calibration pass.  calib != NULL: md_vit_encode / md_vision_project run the bf16 path and record the running
 * max |x| of every quantisation site into calib (device fp32, zeroed by the caller: [4 l + {0 ln1, 1 attention,
 * 2 ln2, 3 gelu}] for block l, then [4 n_layers] the projector's concatenated input, [4 n_layers + 1] its GELU
 * output).  calib == NULL and blocks != NULL: the fp8 path runs (patch embedding, attention, layer norms'
 * statistics and the residual stream stay bf16 / fp32). */
typedef struct {
  md_linear_f8 qkv, proj, fc1, fc2;
  float s_ln1, s_att, s_ln2, s_ff;
} md_vit_block_f8;
typedef struct {
  const md_vit_block_f8* blocks; /* host array of n_layers, or NULL */
  md_linear_f8 proj_fc1, proj_fc2;
  float s_cat, s_pff;
  float* calib;
} md_vit_f8;

typedef struct {
  int32_t dim, n_heads, n_layers, ff_dim;
  int32_t patch, crop;     /* 14, 378 */
  md_linear patch_emb;
  const void* pos_emb;     /* bf16 [grid*grid][dim] */
  const md_vit_block* blocks; /* host array of n_layers */
  md_layernorm post_ln;
  md_linear proj_fc1, proj_fc2; /* vision projector MLP (vision.py:77-89) */
  const void* pixel_lut;   /* bf16 [256] */
  const md_vit_f8* f8;     /* NULL: bf16 everywhere (the reference's precision) */
  int32_t tile_policy;     /* MD_TILE_BY_SHAPE / MD_TILE_PINNED for every GEMM of this call (ABI 5) */
} md_vit_model;

enum { MD_CROPS_U8_HWC = 0, MD_CROPS_BF16_CHW = 1 };

size_t md_vit_workspace_bytes(const md_vit_model* m, int32_t n_crops);

/* vision_encoder (reference: vision.py:64-74 behind moondream.py:168-169):
 * crops -> out bf16 [n_crops][grid*grid][dim]. */
md_status md_vit_encode(const md_vit_model* m, const void* crops, int32_t crops_kind,
                        int32_t n_crops, void* out, void* workspace, size_t workspace_bytes,
                        void* stream);

size_t md_vision_project_workspace_bytes(const md_vit_model* m, int32_t n_images);

/* vision_projection for n_images images that all share one tiling
 * (reference: vision.py:77-89 behind moondream.py:171-172, plus the stitch of
 * moondream.py:213-226).  feats: [n_images][1 + th*tw][g*g][dim];
 * out: bf16 [n_images][g*g][ld_out] (proj_out_dim columns). */
md_status md_vision_project(const md_vit_model* m, const void* feats, int32_t n_images,
                            int32_t tiles_h, int32_t tiles_w, int32_t margin, void* out,
                            int64_t ld_out, void* workspace, size_t workspace_bytes, void* stream);

/* The seam form of the same stage: _vis_proj(g, r) with r the ALREADY stitched
 * [H][W][dim] grid, exactly what the reference passes (moondream.py:171-172,
 * vision.py:77-89).  global_feats: [g*g][dim]; out: [g*g][ld_out].  Workspace:
 * md_vision_project_workspace_bytes(m, 1). */
md_status md_vision_project_grid(const md_vit_model* m, const void* global_feats,
                                 const void* grid_feats, int32_t H, int32_t W, void* out,
                                 int64_t ld_out, void* workspace, size_t workspace_bytes, void* stream);

typedef struct {
  md_layernorm ln;
  md_linear qkv, proj, fc1, fc2;
  /* Optional fusion: qkv and fc1 read the same LayerNorm output (text.py:145-157), so
   * their weights may be packed as ONE matrix [qkv rows | fc1 rows] and run as one
   * GEMM (GELU on the fc1 columns only); every output element is computed exactly
   * as by the two separate layers.  w == NULL: not packed, the two layers run
   * separately. */
  md_linear qkv_fc1;
} md_text_block;

/* Optional FP8 copies of the decoder's weight stream: used by md_text_forward / md_lm_head / md_decode_step for
 * launches of <= 64 rows (the decode regime) when md_text_model.fp8 is set; prefill always runs the bf16 weights.
 * A member with w == NULL falls back to bf16 for that layer. */
typedef struct {
  md_linear_fp8 qkv_fc1, proj, fc2;
} md_text_block_fp8;
typedef struct {
  const md_text_block_fp8* blocks; /* host array of n_layers */
  md_linear_fp8 lm_head;
} md_text_fp8;

/* Optional FP8 mode of the decoder's PREFILL (launches of more than 64 rows; md_gemm_f8): e4m3 copies of the fused
 * qkv|fc1, proj and fc2 matrices of every block and one static scale per quantised activation tensor.  calib as in
 * md_vit_f8: [3 l + {0 ln, 1 attention, 2 gelu}].  Needs the fused qkv|fc1 packing.  RoPE, attention, the KV cache
 * and the residual stream stay bf16. */
typedef struct {
  md_linear_f8 qkv_fc1, proj, fc2;
  float s_ln, s_att, s_ff;
} md_text_block_f8;
typedef struct {
  const md_text_block_f8* blocks; /* host array of n_layers, or NULL */
  float* calib;
} md_text_f8;

typedef struct {
  int32_t dim, n_heads, n_kv_heads, n_layers, ff_dim, vocab, max_context, prefix_len, rot_dim;
  const md_text_block* blocks; /* host array of n_layers */
  md_layernorm post_ln;
  md_linear lm_head;
  const void* wte;       /* bf16 [vocab][dim] */
  const float* freqs;    /* fp32 [max_context][rot_dim/2][2] */
  const md_text_fp8* fp8; /* NULL: bf16 weights everywhere (the reference's precision) */
  const md_text_f8* f8;  /* NULL: bf16 prefill */
  int32_t tile_policy;   /* MD_TILE_BY_SHAPE / MD_TILE_PINNED for every GEMM of this call (ABI 5) */
} md_text_model;

/* KV slabs: layer l's keys at k + l*layer_stride, element (b,h,p,d) at
 * b*batch_stride + (h*ctx + p)*64 + d (the reference's [1,H,2048,64] slab per
 * sequence, moondream.py:62-72, batched on a leading axis). */
typedef struct {
  void* k;
  void* v;
  int64_t layer_stride, batch_stride; /* elements */
  int32_t ctx;
  /* Optional e4m3fn copy of the cache for the decode steps of the fp8 mode (NULL: bf16 only): slabs of the same shape with
   * one byte per element and ONE static scale per layer (HOST arrays of n_layers floats: value ~= scale * fp8).  When
   * set, md_text_forward keeps both copies current (prefill rows are quantised after the bf16 rows are written, a
   * decode step writes the new row into both) and the decode attention reads the e4m3 copy -- half the bytes of the
   * step's dominant stream.  MHA with head_dim 64 only; md_text_forward_lora and md_decode_step_b1 ignore it. */
  void* k8;
  void* v8;
  const float* k_scale;
  const float* v_scale;
} md_kv_cache;

/* md_attention_decode_rope over the e4m3 copy of ONE layer's slabs (k8 / v8: that layer's e4m3 K / V slab, k_scale /
 * v_scale its scales): RoPE of the new token's q / k, the new K / V row written into BOTH copies, attention of every
 * (sequence, head) over keys [0, kv_len[b]) of the e4m3 copy with the new row taken as the cache will hold it
 * (quantised, dequantised).  head_dim 64, MHA.  Tolerance-judged against md_attention_decode_rope. */
md_status md_attention_decode_rope_f8(const void* qkv, int64_t ld, void* o, int64_t ldo, const float* freqs, void* k_slab,
                                      void* v_slab, void* k8_slab, void* v8_slab, int64_t slab_batch_stride, int32_t ctx,
                                      const int32_t* kv_len, int32_t batch, int32_t n_heads, int32_t rot_dim, float scale,
                                      float k_scale, float v_scale, void* stream);

/* (Re)build the e4m3 copy from the bf16 slabs for positions pos .. pos + n_pos - 1 of `batch` slots, every layer and
 * head (pos = pos0[b], device int32, or pos_fixed when pos0 is NULL): after bf16 rows were written by something other
 * than md_text_forward (load_encoded_image's copy). */
md_status md_kv_quantize_f8(const md_kv_cache* kv, int32_t n_layers, int32_t batch, int32_t n_heads, const int32_t* pos0,
                            int32_t pos_fixed, int32_t n_pos, void* stream);

size_t md_text_workspace_bytes(const md_text_model* m, int32_t batch, int32_t q_len);

/* text_decoder over a block of q_len new tokens per sequence (reference:
 * text.py:128-160 behind moondream.py:174-192): x bf16 [batch*q_len][dim]
 * (embeddings) -> hidden bf16 [batch*q_len][dim]; writes K,V at positions
 * pos0[b] .. pos0[b]+q_len-1 of every layer's slab.  x may alias hidden. */
md_status md_text_forward(const md_text_model* m, const void* x, void* hidden, int32_t batch,
                          int32_t q_len, const int32_t* pos0, const md_kv_cache* kv,
                          void* workspace, size_t workspace_bytes, void* stream);

/* LoRA "variant" side path (reference: lora.py:54-79 -> text.py:31-32,55-56 and layers.py:129-146 with lora != None).
 * delta(x) = (x A^T) B^T with A [r][k] and B [n][r] packed as bias-free md_linear (r zero-padded to 64); a pair with
 * a.w == NULL is absent.  Per block, in the reference's order and with its bf16 roundings:
 *   qkv  = bf16(qkv(l_in) + delta_qkv(l_in))
 *   attn = bf16(proj(att) + delta_proj(l_in))        -- sic: the reference feeds the BLOCK INPUT to this pair (text.py:55)
 *   h    = gelu(bf16(fc1(l_in) + delta_fc1(l_in)));  mlp = bf16(fc2(h) + delta_fc2(h))
 *   x    = bf16(bf16(x + attn) + mlp) */
typedef struct {
  md_linear a, b;
} md_lora_pair;
typedef struct {
  md_lora_pair qkv, proj, fc1, fc2;
} md_text_block_lora;

/* md_text_forward with the side path: lora = host array of n_layers entries (NULL: plain md_text_forward).
 * Runs the unfused kernels (separate qkv / fc1, stand-alone GELU and adds). */
size_t md_text_lora_workspace_bytes(const md_text_model* m, int32_t batch, int32_t q_len);
md_status md_text_forward_lora(const md_text_model* m, const md_text_block_lora* lora, const void* x, void* hidden,
                               int32_t batch, int32_t q_len, const int32_t* pos0, const md_kv_cache* kv,
                               void* workspace, size_t workspace_bytes, void* stream);

/* out[r, :cols] = bf16(a[r, :cols] + b[r, :cols]);  out = gelu_tanh(a)  (bf16 rows, cols % 8 == 0). */
md_status md_add_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int32_t rows,
                      int32_t cols, void* stream);
md_status md_gelu_bf16(const void* a, int64_t lda, void* out, int64_t ldo, int32_t rows, int32_t cols, void* stream);

size_t md_lm_head_workspace_bytes(const md_text_model* m, int32_t batch);

/* lm_head on the LAST token of each sequence (reference: text.py:163-167):
 * hidden [batch*q_len][dim] -> logits bf16 [batch][ld_logits]. */
md_status md_lm_head(const md_text_model* m, const void* hidden, int32_t batch, int32_t q_len,
                     void* logits, int64_t ld_logits, void* workspace, size_t workspace_bytes,
                     void* stream);

/* Single-sequence decode as ONE persistent kernel (csrc/decode_b1.hip): all decoder blocks of one token inside one
 * launch of one workgroup per CU, grid barriers instead of kernel boundaries (the body of _decode_one_tok,
 * moondream.py:183-192 -> text.py:128-160, at batch 1).  x_in / hidden: bf16 [dim]; kv: the sequence's slot
 * (k / v pointing at its batch entry); sync_state: >= 16 KiB of device memory owned by the caller, ZERO before first
 * use and written by nothing else (barrier counters carry over from launch to launch; word 704 is raised if a
 * barrier ever timed out).  fp32 matrix-vector products and per-slice softmax maxima: the same tolerance against the
 * reference as the batched kernels, not bit-identical to them.  Needs the fused qkv|fc1 packing and n_kv_heads ==
 * n_heads, head_dim 64. */
size_t md_decode_b1_workspace_bytes(const md_text_model* m);
/* 1 when md_decode_step_b1 / md_decode_b1_layers can run this model with this cache on the current device: every static
 * limit of the kernel (dims, head layout, ctx <= 2048, fused packing, biases present, no fp8 weights attached) AND one
 * workgroup per CU of its grid resident at once (occupancy query; its software grid barriers need the whole grid on
 * the chip).  0: use md_decode_step.  Callers should also keep it off streams that run next to other persistent
 * kernels: co-residency with ANOTHER kernel's workgroups cannot be queried (a barrier then times out, raises word 704
 * of sync_state and the step's outputs are invalid -- repeat the step with md_decode_step). */
int32_t md_decode_step_b1_supported(const md_text_model* m, const md_kv_cache* kv);
md_status md_decode_b1_layers(const md_text_model* m, const void* x_in, void* hidden, const int32_t* pos,
                              const md_kv_cache* kv, void* workspace, size_t workspace_bytes, void* sync_state,
                              void* stream);
/* md_decode_step for ONE sequence on that kernel: embed -> md_decode_b1_layers -> lm_head -> suppress -> argmax;
 * pos[0] += 1.  Workspace: md_decode_step_b1_workspace_bytes(m). */
size_t md_decode_step_b1_workspace_bytes(const md_text_model* m);
md_status md_decode_step_b1(const md_text_model* m, const int32_t* token, int32_t* next, int32_t* pos,
                            const md_kv_cache* kv, int32_t suppress_id, void* logits, int64_t ld_logits,
                            void* workspace, size_t workspace_bytes, void* sync_state, void* stream);

/* One greedy decode step for the whole batch, device-resident (the body of the
 * reference's generator loop, moondream.py:512-530, without its per-token host
 * sync): embed tokens[b] -> decoder at pos[b] -> lm_head -> suppress
 * suppress_id -> argmax -> next[b]; pos[b] += 1.  done[b] != 0 sequences are
 * still computed (lockstep) but keep their token. */
md_status md_decode_step(const md_text_model* m, const int32_t* tokens, int32_t* next, int32_t* pos,
                         int32_t batch, const md_kv_cache* kv, int32_t suppress_id, void* logits,
                         int64_t ld_logits, void* workspace, size_t workspace_bytes, void* stream);

size_t md_decode_workspace_bytes(const md_text_model* m, int32_t batch);

#ifdef __cplusplus
}
#endif
#endif /* MOONDREAM_HIP_H */
