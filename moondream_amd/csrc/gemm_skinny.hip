// Decode-regime linear layer: C[M <= 64, N] = epilogue(X . W^T + b).
//
// With at most 64 activation rows (one per sequence in flight) the layer is a
// stream over the weights: every W byte is used by <= 64 rows, so the bound is
// HBM bandwidth (2.6 GB of weights per decode step), not
// the matrix cores.  A CU streams ~10 B/clk at best, so the whole chip has to
// pull: the grid is (N / 32 weight-row tiles) x (S K-slices), S chosen so that
// ~1000 workgroups of 4 waves are resident, each wave streaming its own K range
// with 8 x 1 KiB loads in flight straight into MFMA operand registers (row =
// lane & 31, 16 B per lane; four consecutive K-steps consume a full 128-byte
// line per row; no LDS round trip for data that is read exactly once).
//
// Reduction over K is DETERMINISTIC and independent of M: the 4 waves of a
// workgroup combine through LDS in wave order; the S workgroups of a tile
// publish fp32 slabs and the last one to arrive (agent-scope release -> ticket
// -> acquire, cdna guide section 6 guideline 16) sums them in slice order and
// runs the epilogue.  So batched decode equals sequential decode bit for bit.
//
// MFMA is used because it is the cheapest way to issue 64 rows x 32 cols x 16 k
// of FMAs per instruction, not because the kernel is compute bound.
#include "md_common.hpp"

namespace {

struct SkinnyK {
  const bf16_t* X;
  const bf16_t* W;
  const bf16_t* bias;
  const bf16_t* R;
  bf16_t* C;
  float* slabs;        // [tile][slice][MT*16*64] fp32, slices > 1 only
  unsigned* tickets;   // [tile], zero on entry, left zero on exit
  int64_t ldx, ldw, ldc, ldr;
  int M, n_store, n_pad, K;
  int res_row_mod, slices;
};

constexpr int SK_WAVES = 4;
constexpr int SK_CHUNK = 256;                 // K elements of X staged in LDS per round (16 K-steps, 4 per wave)
constexpr int SK_XSTR = SK_CHUNK * 2 + 16;    // padded LDS row stride (bytes): 16-B slots rotate by one per row

template <int MT, int EPI>
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_skinny_kernel(const SkinnyK p) {
  // X chunk [MT*32 rows][256 k] during the K loop; afterwards reused as the
  // [wave][m tile][acc reg][lane] partial-sum exchange (slot 0 doubles as the "last" flag)
  constexpr int X_BYTES = MT * 32 * SK_XSTR, PART_BYTES = SK_WAVES * MT * 16 * 64 * 4;
  __shared__ __attribute__((aligned(16))) char lds[X_BYTES > PART_BYTES ? X_BYTES : PART_BYTES];
  float(*part)[MT][16][64] = reinterpret_cast<float(*)[MT][16][64]>(lds);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int tile = blockIdx.x, slice = blockIdx.y;
  const int n0 = tile * 32;

  // this workgroup's K range: whole chunks, contiguous
  const int chunks = (p.K + SK_CHUNK - 1) / SK_CHUNK;
  const int per = (chunks + p.slices - 1) / p.slices;
  const int c0 = min(slice * per, chunks), c1 = min(c0 + per, chunks);

  const bf16_t* wrow = p.W + (int64_t)min(n0 + l31, p.n_pad - 1) * p.ldw + 8 * hi;

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  for (int c = c0; c < c1; ++c) {
    const int k0 = c * SK_CHUNK;
    const int ksteps = min(SK_CHUNK, p.K - k0) / 16;  // K is a multiple of 64
    // weights first: four 1 KiB loads per wave (a full 128-byte line per row) fly
    // while the activations are staged
    bf16x8 wf[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int st = min(4 * wave + u, ksteps - 1);
      wf[u] = __builtin_nontemporal_load((const bf16x8*)(wrow + k0 + st * 16));
    }
    // X chunk -> LDS in whole 16-byte pieces, coalesced along K (L2-resident, shared
    // by all four waves; fragment-shaped global loads of X would triple the traffic
    // through the per-CU load path)
    __syncthreads();  // previous chunk fully consumed
    for (int i = tid; i < MT * 32 * (SK_CHUNK / 8); i += SK_WAVES * 64) {
      const int row = i / (SK_CHUNK / 8), ch = i % (SK_CHUNK / 8);
      u32x4 v = {0, 0, 0, 0};
      if (ch * 8 < ksteps * 16) v = *(const u32x4*)(p.X + (int64_t)min(row, p.M - 1) * p.ldx + k0 + ch * 8);
      *(u32x4*)(lds + row * SK_XSTR + ch * 16) = v;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int st = 4 * wave + u;
      if (st < ksteps) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const bf16x8 xf = *(const bf16x8*)(lds + (mt * 32 + l31) * SK_XSTR + st * 32 + hi * 16);
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u], xf, acc[mt], 0, 0, 0);
        }
      }
    }
  }

  __syncthreads();  // X buffer is dead; reuse it for the partials
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][mt][r][lane] = acc[mt][r];
  __syncthreads();

  constexpr int SLOTS = MT * 16 * 64, PER_T = SLOTS / (SK_WAVES * 64);
  float v[PER_T];
#pragma unroll
  for (int i = 0; i < PER_T; ++i) {
    const int slot = tid + i * SK_WAVES * 64;
    const int ln = slot & 63, r = (slot >> 6) & 15, mt = slot >> 10;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < SK_WAVES; ++w) a += part[w][mt][r][ln];
    v[i] = a;
  }

  if (p.slices > 1) {
    // publish this slice's slab, take a ticket; the last arriver reduces
    float* slab = p.slabs + ((int64_t)tile * p.slices + slice) * SLOTS;
#pragma unroll
    for (int i = 0; i < PER_T; ++i) slab[tid + i * SK_WAVES * 64] = v[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // also orders the reads of part[] above before the flag write below
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned t = __hip_atomic_fetch_add(p.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      part[0][0][0][0] = (t == (unsigned)p.slices - 1u) ? 1.f : 0.f;  // part[] was consumed before the barrier above
    }
    __syncthreads();
    if (part[0][0][0][0] == 0.f) return;
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(p.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    __syncthreads();
    const float* base = p.slabs + (int64_t)tile * p.slices * SLOTS;
#pragma unroll
    for (int i = 0; i < PER_T; ++i) v[i] = 0.f;
    for (int sl = 0; sl < p.slices; ++sl)  // fixed slice order: deterministic sum
#pragma unroll
      for (int i = 0; i < PER_T; ++i) v[i] += base[(int64_t)sl * SLOTS + tid + i * SK_WAVES * 64];
  }

  // the reference's rounding points: bf16(acc + bias), then the elementwise op
#pragma unroll
  for (int i = 0; i < PER_T; ++i) {
    const int slot = tid + i * SK_WAVES * 64;
    const int ln = slot & 63, r = (slot >> 6) & 15, mt = slot >> 10;
    const int m = mt * 32 + (ln & 31);
    const int n = n0 + 8 * (r >> 2) + 4 * (ln >> 5) + (r & 3);
    if (m < p.M && n < p.n_store) {
      float a = v[i];
      if (p.bias != nullptr) a += bf2f(p.bias[n]);
      float y = bf2f(f2bf(a));
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
  dim3 grid((k.n_store + 31) / 32, k.slices), block(SK_WAVES * 64);
  if (k.M <= 32)
    hipLaunchKernelGGL((gemm_skinny_kernel<1, EPI>), grid, block, 0, s, k);
  else
    hipLaunchKernelGGL((gemm_skinny_kernel<2, EPI>), grid, block, 0, s, k);
  return md_launch_status();
}

// K-slices so that ~1024 workgroups are resident; a function of the layer shape
// only (never of M), so the summation tree of a row never depends on the batch.
int pick_slices(int n_store, int k_pad) {
  const int tiles = (n_store + 31) / 32;
  int s = 1;
  const int chunks = (k_pad + SK_CHUNK - 1) / SK_CHUNK;
  while (s < 8 && tiles * s * 2 <= 1024 && s * 2 <= chunks) s *= 2;
  return s;
}

}  // namespace

constexpr size_t SK_TICKET_BYTES = 8192;  // fixed-size ticket area at the head of the scratch (<= 1024 tiles when sliced)

size_t md_gemm_skinny_ws_bytes(const md_linear* lin, int store_pad) {
  const int n_store = store_pad ? lin->n_pad : lin->n;
  const int s = pick_slices(n_store, lin->k_pad);
  if (s == 1) return 0;
  const size_t tiles = (n_store + 31) / 32;
  return SK_TICKET_BYTES + tiles * s * (2 * 16 * 64) * sizeof(float);
}

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
  k.slices = pick_slices(k.n_store, k.K);
  k.slabs = nullptr;
  k.tickets = nullptr;
  if (k.slices > 1) {
    const size_t need = md_gemm_skinny_ws_bytes(&a->lin, a->store_pad_cols);
    if (a->splitk_ws == nullptr || a->splitk_ws_bytes < need) {
      k.slices = 1;  // no scratch from the caller: single-slice (still deterministic, just slower)
    } else {
      k.tickets = (unsigned*)a->splitk_ws;
      k.slabs = (float*)((char*)a->splitk_ws + SK_TICKET_BYTES);
    }
  }
  switch (a->epilogue) {
    case MD_EPI_BIAS: return launch<MD_EPI_BIAS>(k, stream);
    case MD_EPI_GELU: return launch<MD_EPI_GELU>(k, stream);
    case MD_EPI_RESIDUAL: return launch<MD_EPI_RESIDUAL>(k, stream);
    default: return MD_ERR_INVALID_ARG;
  }
}
