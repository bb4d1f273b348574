// Internal interface between the GEMM translation units (not part of the C ABI).
#pragma once
#include "md_common.hpp"

// One launch of  C = epilogue(A . W^T + b)  as the kernels see it.
struct GemmK {
  const bf16_t* A;
  const bf16_t* W;
  const bf16_t* bias;
  const bf16_t* R;
  bf16_t* C;
  int64_t lda, ldw, ldc, ldr;
  int M, n_store, n_pad, K;
  int tiles_m, tiles_n;
  int res_row_mod;
  int group_m;    // tile-order grouping (row panels per group)
  int gelu_from;  // EPI_GELU: columns >= gelu_from get the GELU
  int nt;         // decode regime: stream the weights with the non-temporal policy
  // launch-boundary split-K: every K slice stores its fp32 partial tile [slice][m][ldp] and exits;
  // the consumer kernel sums the slices (md_reduce_residual_layernorm)
  float* partial;
  int64_t partial_ld, partial_slice_stride;
  // split-K (decode regime only): K slices per output tile, fp32 slabs, arrival tickets
  int slices;
  float* slabs;
  unsigned* tickets;
  // MD_EPI_QKV_ROPE (four-wave kernel only): the decoder prefill's fused [q | k | v | gelu(fc1)] layer with the partial RoPE
  // (rope.py:20-48) and the KV-cache update (text.py:45-46, moondream.py:74-78) applied in the epilogue
  const float* rope_cs = nullptr;      // fp32 [M][32]: (cos, sin) of the row's position for the 16 rotated pairs
  const uint32_t* rope_kv = nullptr;   // [M]: byte offset of (slot, position) in a layer's K / V slab
  bf16_t* kslab = nullptr;
  bf16_t* vslab = nullptr;
  uint32_t slab_bytes = 0;             // bytes of one layer's K (= V) slab
  int rope_d = 0;                      // n_heads * 64: width of each of the q, k, v sections
  int rope_ctx = 0;                    // cache positions per head
};
constexpr int MD_EPI_QKV_ROPE = 3;     // internal to the library (md_gemm_qkv_rope)

// The fused layer of a decoder block at prefill with RoPE + KV write in its epilogue, when the four-wave kernel takes the
// shape (MD_ERR_UNSUPPORTED otherwise: the caller runs md_gemm_bf16 + md_rope_kv_write).  MHA, head_dim 64, rot_dim 32.
struct md_rope_fuse {
  const float* row_cs;
  const uint32_t* row_kv;
  void* kslab;
  void* vslab;
  uint64_t slab_bytes;
  int n_heads, ctx;
};
md_status md_gemm_qkv_rope(const md_gemm_args* a, const md_rope_fuse* rf, hipStream_t stream);
// the same for the opt-in FP8 mode's tile GEMM (gemm_f8.hip); rf8 (or its slabs) may be null: no e4m3 copy of the cache
struct md_rope_fuse_f8 {
  void* k8slab;
  void* v8slab;
  float k_scale, v_scale;
};
int md_gemm_auto_group_m(int n_store);  // tile-order grouping by shape (gemm_bf16.hip): round 3's rule (eight-wave kernels, fp8 kernel)
int md_gemm_auto_group_m_w4(int n_store, int k);  // the four-wave kernel's, re-swept in round 4
bool md_gemm_knob_rope_fuse();  // MD_ROPE_FUSE / md_gemm_set_tuning("rope_fuse"): A/B and tests
md_status md_gemm_f8_qkv_rope(const md_gemm_f8_args* a, const md_rope_fuse* rf, const md_rope_fuse_f8* rf8, hipStream_t stream);

// gemm_w4.hip: the 256x256 tile kernel with one wave per SIMD (4 waves x 128x128), persistent.
// epi = MD_EPI_*.  Fills tiles_m / tiles_n itself.
md_status md_gemm_w4_launch(const GemmK& k, int epi, hipStream_t stream);
bool md_gemm_w4_takes(const GemmK& k, int epi);  // shape limits of the four-wave kernel for this launch
void md_gemm_w4_set_grid(int v);     // persistent workgroups per launch of the four-wave kernel (0 = one per CU)
void md_gemm_w4_set_debug(int half, uint32_t v);  // measurement builds: device buffer for in-kernel stamps
void md_gemm_w4_set_variant(int v);  // measurement hook: schedule / ablation variant of the bias-epilogue kernel
void md_attention_set_skip_dead(int v);  // attention.hip: exact work skipping of the prefill kernel (bit 0: dead half of the last key tile, bit 1: waves without a live query row)
