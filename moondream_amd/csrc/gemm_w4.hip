// bf16 GEMM, big-tile kernel:  C = epilogue(A . W^T + b)  for M > 64 and K % 64 == 0.
//
// One workgroup = FOUR waves, one per SIMD, each owning a 128 x 128 quarter of a 256 x 256 tile
// of C with its 16 accumulator blocks (256 registers) in the accumulator half of the register
// file.  Per 16-wide K step a wave issues 16 v_mfma_f32_32x32x16_bf16 (512 matrix-pipe cycles) for
// 8 ds_read_b128 -- two thirds of the LDS reads per FLOP of an eight-wave 128 x 64 split, and no
// second wave on the SIMD to arbitrate with.  Because nothing else runs on the SIMD, everything
// that is not an MFMA is a FILLER placed by hand between two MFMAs (<= 2 per 32-cycle gap):
//
//   stream of 32-wide K slices, software pipelined over three levels
//     HBM/L2 -> registers   raw buffer loads, 16 B per lane, issued ~2 slices (2 x 1024 cycles) ahead
//     registers -> LDS      ds_write_b128 into a 2-stage ring, chunk-swizzled on the WRITE side
//     LDS -> fragments      ds_read_b128 one 16-wide K step ahead of the MFMAs that consume them
//   one s_barrier per slice, sitting between two MFMAs.
//
// (LDS-DMA is deliberately not used here: an LDS-DMA issue costs its wave 60-180 cycles, which a
// second wave on the SIMD can cover but a lone wave cannot; a buffer load and a ds_write cost a
// few issue cycles each.)
//
// The slice stream is CONTINUOUS across the tiles of a persistent workgroup (grid = one
// workgroup per CU, tiles blockIdx.x, blockIdx.x + gridDim.x, ...): while the last slices of a
// tile are multiplied, the first slices of the next tile are already being loaded and written, so
// the epilogue is the only part of a tile that does not overlap with MFMA work.
//
// Numerics: K is accumulated in the same order as every other tile config (sequential 16-wide
// steps into one fp32 accumulator), so results are bit-identical to them.
#include "gemm_internal.hpp"

#include <algorithm>
#include <type_traits>

namespace {

constexpr int BM = 256, BN = 256, BKT = 32;
constexpr int ROW_BYTES = BKT * 2;             // 64-byte rows: 4 chunks of 16 bytes
constexpr int A_BYTES = BM * ROW_BYTES;        // 16 KiB
constexpr int STAGE = (BM + BN) * ROW_BYTES;   // 32 KiB
constexpr int RING = 2 * STAGE;                // 64 KiB
constexpr int XPOSE_BYTES = 32 * 256;          // 32 rows x 128 bf16: the epilogue's transposition tile
constexpr int SCRATCH_PER_WAVE = XPOSE_BYTES + 256;  // + the wave's 128 bias values
constexpr int LDS_BYTES = RING + 4 * SCRATCH_PER_WAVE;

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int OFF>
__device__ __forceinline__ void ds_read_b128(bf16x8& dst, uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int OFF>
__device__ __forceinline__ void ds_write_b128(uint32_t addr, const u32x4& v) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "i"(OFF) : "memory");
}
__device__ __forceinline__ void ds_write_b64_asm(uint32_t addr, u32x2 v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void ds_read_b128_u32(u32x4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}
template <int OFF>
__device__ __forceinline__ void ds_read_b64_u32(u32x2& dst, uint32_t addr) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
}
#define MD_PIN() __builtin_amdgcn_sched_barrier(0)

// The 16 accumulator blocks of a wave (256 registers) live in a[0:255], OWNED BY INLINE ASM: block X is
// a[16X : 16X+15].  As compiler-visible f32x16 values they made the register allocator shuffle and spill
// around every control-flow join; named literally they cost it nothing.  acc_reserve() lists them as
// clobbers once (which also makes the kernel descriptor allocate them); the build audits that no
// compiler-generated v_accvgpr_* / scratch instruction appears in the kernel (see _lib.build_library).
#define MD_A16(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
__device__ __forceinline__ void acc_reserve() {
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", MD_A16(1), MD_A16(2), MD_A16(3), MD_A16(4),
               MD_A16(5), MD_A16(6), MD_A16(7), MD_A16(8), MD_A16(9), MD_A16(10), MD_A16(11), MD_A16(12), MD_A16(13), MD_A16(14),
               MD_A16(15), MD_A16(16), MD_A16(17), MD_A16(18), MD_A16(19), MD_A16(20), MD_A16(21), MD_A16(22), MD_A16(23),
               MD_A16(24), "a250", "a251", "a252", "a253", "a254", "a255");
}
// block X (+)= W-fragment . A-fragment^T      (first operand = weight rows, so a lane holds one row m and runs of 4 columns n)
template <int X, bool FIRST>
__device__ __forceinline__ void mfma_acc(const bf16x8& wf, const bf16x8& af) {
  if constexpr (FIRST)
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(wf), "v"(af), "i"(16 * X), "i"(16 * X + 15));
  else
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(wf), "v"(af), "i"(16 * X), "i"(16 * X + 15));
}
template <int N>
__device__ __forceinline__ float acc_read() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "i"(N));
  return v;
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_w4_kernel(const GemmK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int nwg = p.tiles_m * p.tiles_n;
  const int nk = p.K / BKT;  // even: K % 64 == 0

  // workgroup sequence number -> tile: XCD-contiguous remap, then grouped (group_m row panels x all
  // column panels) order, so the 32 workgroups of an XCD that run together cover a compact block
  const int per_group = p.group_m * p.tiles_n;
  auto tile_origin = [&](int vv, int& m0, int& n0) {
    const int L = xcd_remap(vv, nwg);
    const int first_m = (L / per_group) * p.group_m;
    const int gsz = min(p.tiles_m - first_m, p.group_m);
    m0 = (first_m + (L % per_group) % gsz) * BM;
    n0 = ((L % per_group) / gsz) * BN;
  };

  // ---- load cursor: runs ~3 slices ahead of the MFMAs, across tile boundaries -----------------
  // piece j of this thread: LDS slot j*256 + tid = row (slot >> 2), PHYSICAL chunk (slot & 3); it holds the
  // row's LOGICAL chunk (slot & 3) ^ ((row >> 2) & 3), so a ds_read_b128 of one chunk column over 16
  // consecutive-ish rows touches 16 distinct 16-byte bank slots.
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0xffffffffu, 0x00020000);
  uint32_t a_voff[4], b_voff[4];
  int ld_tile = blockIdx.x, ld_slice = 0;
  uint32_t ld_soff = 0;
  auto set_load_tile = [&](int vv) {
    int m0, n0;
    tile_origin(vv, m0, n0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int slot = j * 256 + tid, r = slot >> 2, c = (slot & 3) ^ ((r >> 2) & 3);
      a_voff[j] = (uint32_t)min(m0 + r, p.M - 1) * (uint32_t)(p.lda * 2) + c * 16;
      b_voff[j] = (uint32_t)min(n0 + r, p.n_pad - 1) * (uint32_t)(p.ldw * 2) + c * 16;
    }
  };
  set_load_tile(ld_tile);

  u32x4 R[2][8];  // two slices in registers: one landed / being written to LDS, one in flight
  auto load_piece = [&](auto rs_c, auto j_c) {
    constexpr int RS = decltype(rs_c)::value, J = decltype(j_c)::value;
    if constexpr (J < 4)
      R[RS][J] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, a_voff[J], ld_soff, 0));
    else
      R[RS][J] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, b_voff[J - 4], ld_soff, 0));
  };
  auto advance_load_cursor = [&]() {
    ld_soff += ROW_BYTES;
    if (++ld_slice == nk) {
      ld_slice = 0;
      ld_soff = 0;
      // past the end of the stream the cursor stays on the last tile: the loads keep going
      // (their data is never written anywhere that is read), which keeps the loop free of branches
      if (ld_tile + (int)gridDim.x < nwg) {
        ld_tile += gridDim.x;
        set_load_tile(ld_tile);
      }
    }
  };
  const uint32_t wr_addr = lds_base + tid * 16;
  auto write_piece = [&](auto rs_c, auto j_c, auto stage_c) {
    constexpr int RS = decltype(rs_c)::value, J = decltype(j_c)::value, ST = decltype(stage_c)::value;
    ds_write_b128<ST * STAGE + (J < 4 ? J * 4096 : A_BYTES + (J - 4) * 4096)>(wr_addr, R[RS][J]);
  };

  // ---- fragment reads: K step s of a slice = logical chunks 2s (lanes 0-31) and 2s+1 (lanes 32-63)
  const uint32_t swz = (l31 >> 2) & 3;
  const uint32_t a_row = lds_base + (wm * 128 + l31) * ROW_BYTES;
  const uint32_t b_row = lds_base + A_BYTES + (wn * 128 + l31) * ROW_BYTES;
  const uint32_t coff0 = ((0 + hi) ^ swz) * 16, coff1 = ((2 + hi) ^ swz) * 16;
  const uint32_t ra[2] = {a_row + coff0, a_row + coff1}, rb[2] = {b_row + coff0, b_row + coff1};
  bf16x8 fa[2][4], fb[2][4];  // [fragment set][32-row block]
  // read q (0..7) of K step S of the slice in ring stage ST into fragment set SET: B blocks first
  auto read_frag = [&](auto set_c, auto st_c, auto s_c, auto q_c) {
    constexpr int SET = decltype(set_c)::value, ST = decltype(st_c)::value, S = decltype(s_c)::value, Q = decltype(q_c)::value;
    if constexpr (Q < 4)
      ds_read_b128<ST * STAGE + Q * 32 * ROW_BYTES>(fb[SET][Q], rb[S]);
    else
      ds_read_b128<ST * STAGE + (Q - 4) * 32 * ROW_BYTES>(fa[SET][Q - 4], ra[S]);
  };

  acc_reserve();

  // ---- stream prologue: slices 0 and 1 requested, slice 0 written, slice 2 requested ----------
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  static_for<0, 8>([&](auto j) { load_piece(I0{}, j); });
  advance_load_cursor();
  static_for<0, 8>([&](auto j) { load_piece(I1{}, j); });
  advance_load_cursor();
  asm volatile("" ::: "memory");
  static_for<0, 8>([&](auto j) { write_piece(I0{}, j, I0{}); });
  asm volatile("" ::: "memory");
  static_for<0, 8>([&](auto j) { load_piece(I0{}, j); });
  advance_load_cursor();
  wait_lgkm<0>();
  __builtin_amdgcn_s_barrier();
  static_for<0, 8>([&](auto q) { read_frag(I0{}, I0{}, I0{}, q); });
  wait_lgkm<0>();
  MD_PIN();

  // One slice = two halves of 16 MFMAs.  P = parity of the stream slice g being multiplied:
  //   ring stage P holds slice g; stage 1-P receives slice g+1 from R[1-P] during the FIRST half, and
  //   R[1-P] is then re-requested for slice g+3; the barrier that publishes slice g+1 sits in the
  //   SECOND half, followed by the reads of its first fragments (slice g+2 stays in flight in R[P]).
  // There is ONE straight-line body per parity and no branch inside it: a second code path would
  // merge 256 live accumulator registers at its join (the compiler then shuffles them through
  // copies).  Past the end of the stream the fillers simply keep running on data nobody reads.
  auto slice_body = [&](auto p_c, auto first_c) {
    constexpr int P = decltype(p_c)::value;
    constexpr bool FIRST = decltype(first_c)::value;  // first K step of a tile: accumulate onto zero
    using PP = std::integral_constant<int, P>;
    using NP = std::integral_constant<int, 1 - P>;
    // ---- first half: K step 0 (fragment set 0)
    static_for<0, 16>([&](auto mc) {
      constexpr int X = decltype(mc)::value, I = X / 4, J = X % 4;
      mfma_acc<X, FIRST>(fb[0][J], fa[0][I]);
      MD_PIN();
      if constexpr (X < 8) {
        read_frag(I1{}, PP{}, I1{}, mc);   // set 1 <- slice g, K step 1
        write_piece(NP{}, mc, NP{});       // slice g+1 -> the other stage
      } else {
        load_piece(NP{}, std::integral_constant<int, X - 8>{});  // slice g+3
      }
      MD_PIN();
    });
    advance_load_cursor();
    wait_lgkm<0>();
    MD_PIN();
    // ---- second half: K step 1 (fragment set 1)
    static_for<0, 16>([&](auto mc) {
      constexpr int X = decltype(mc)::value, I = X / 4, J = X % 4;
      mfma_acc<X, false>(fb[1][J], fa[1][I]);
      MD_PIN();
      if constexpr (X == 1) {
        // every wave's writes of slice g+1 were waited for above; every wave's reads of stage 1-P
        // (slice g-1) finished an iteration ago
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if constexpr (X >= 2 && X < 10)
        read_frag(I0{}, NP{}, I0{}, std::integral_constant<int, X - 2>{});  // set 0 <- slice g+1, K step 0
      MD_PIN();
    });
    wait_lgkm<0>();
    MD_PIN();
  };

  // ---- tile loop ------------------------------------------------------------------------------
  const uint32_t tile_lds = lds_base + RING + wave * SCRATCH_PER_WAVE;
  for (int vtile = blockIdx.x; vtile < nwg; vtile += gridDim.x) {
    // the accumulators are (re)defined by the first K step of every tile: nothing is carried from
    // one tile to the next in them
    slice_body(I0{}, std::true_type{});
    slice_body(I1{}, std::false_type{});
    for (int u = 2; u < nk; u += 2) {
      slice_body(I0{}, std::false_type{});
      slice_body(I1{}, std::false_type{});
    }

    // ---- epilogue of tile vtile (the next tile's first slices are already in the ring / in flight)
    // block X = 4 i + j, register r: row m = 32 i + l31, col n = 32 j + 8 (r >> 2) + 4 hi + (r & 3)   within the wave's quarter
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results are not readable before they retire
    int m0c, n0c;
    tile_origin(vtile, m0c, n0c);
    const int wm0 = m0c + wm * 128, wn0 = n0c + wn * 128;
    // the wave's 128 bias values go to LDS once per tile and are re-read per 32-row block (16 broadcast
    // ds_read_b64): held in registers for the whole epilogue they cost 32 VGPRs that the residual
    // variant does not have
    const uint32_t bias_lds = tile_lds + XPOSE_BYTES;
    if (lane < 32) {
      const int n = wn0 + 4 * lane;
      u32x2 bw = {0u, 0u};
      if (p.bias != nullptr && n < p.n_pad) bw = *(const u32x2*)(p.bias + n);
      ds_write_b64_asm(bias_lds + lane * 8, bw);
    }
    // piece q of this lane in a 32 x 128 block: row (q*64 + lane) >> 4, 16-byte chunk (q*64 + lane) & 15
    u32x4 rres[8];
    auto load_residual = [&](int i) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int idx = q * 64 + lane, row = idx >> 4, ch = idx & 15;
        const int m = wm0 + 32 * i + row, n = wn0 + ch * 8;
        rres[q] = u32x4{0, 0, 0, 0};
        if (m < p.M && n < p.n_store) {
          const int64_t rrow = p.res_row_mod ? (m % p.res_row_mod) : m;
          rres[q] = *(const u32x4*)(p.R + rrow * p.ldr + n);
        }
      }
    };
    if constexpr (EPI == MD_EPI_RESIDUAL) load_residual(0);
    MD_PIN();
    static_for<0, 4>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      u32x2 bias_w[4][4];
      static_for<0, 16>([&](auto jq) {
        constexpr int j = decltype(jq)::value / 4, q = decltype(jq)::value % 4;
        ds_read_b64_u32<(32 * j + 8 * q) * 2>(bias_w[j][q], bias_lds + hi * 8);
      });
      wait_lgkm<0>();
      MD_PIN();
      static_for<0, 16>([&](auto jq) {
        constexpr int j = decltype(jq)::value / 4, q = decltype(jq)::value % 4, base = 16 * (4 * i + j) + 4 * q;
        u32x2 w;
        w[0] = pack_bf16x2(acc_read<base + 0>() + lo_bf(bias_w[j][q][0]), acc_read<base + 1>() + hi_bf(bias_w[j][q][0]));
        w[1] = pack_bf16x2(acc_read<base + 2>() + lo_bf(bias_w[j][q][1]), acc_read<base + 3>() + hi_bf(bias_w[j][q][1]));
        constexpr int ch = 4 * j + q;
        ds_write_b64_asm(tile_lds + l31 * 256 + ((ch ^ (l31 & 15)) * 16) + hi * 8, w);
      });
      MD_PIN();
      u32x4 tv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int idx = q * 64 + lane, row = idx >> 4, ch = idx & 15;
        ds_read_b128_u32(tv[q], tile_lds + row * 256 + ((ch ^ (row & 15)) * 16));
      }
      wait_lgkm<0>();
      MD_PIN();
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int idx = q * 64 + lane, row = idx >> 4, ch = idx & 15;
        const int m = wm0 + 32 * i + row, n = wn0 + ch * 8;
        u32x4 v = tv[q];
        if (m < p.M && n < p.n_store) {
          if constexpr (EPI == MD_EPI_GELU) {
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
          *(u32x4*)(p.C + (int64_t)m * p.ldc + n) = v;
        }
      }
      MD_PIN();
      // the next block's residual rows: requested now, used after its transposition
      if constexpr (EPI == MD_EPI_RESIDUAL && i + 1 < 4) load_residual(i + 1);
      MD_PIN();
    });
    // the next tile's first fragments again (the copy read before the epilogue was not kept: 32
    // registers the epilogue needs); its first slice was published by the last barrier above
    static_for<0, 8>([&](auto q) { read_frag(I0{}, I0{}, I0{}, q); });
    wait_lgkm<0>();
    MD_PIN();
  }
}

template <int EPI>
md_status launch(const GemmK& k, hipStream_t stream) {
  auto fn = gemm_w4_kernel<EPI>;
  static bool attr_set = false;  // per process: the library serves the process's current device
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      return MD_ERR_LAUNCH;
    }
    attr_set = true;
  }
  GemmK kk = k;
  kk.tiles_m = (k.M + BM - 1) / BM;
  kk.tiles_n = (k.n_store + BN - 1) / BN;
  const int nwg = kk.tiles_m * kk.tiles_n;
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? (n / 8) * 8 : 256;  // a multiple of 8 keeps (sequence number % 8) == XCD for the tile-order remap
  }();
  const int gx = std::min(nwg, n_cu);  // one persistent workgroup per CU
  hipLaunchKernelGGL(fn, dim3(gx), dim3(256), LDS_BYTES, stream, kk);
  return md_launch_status();
}

}  // namespace

md_status md_gemm_w4_launch(const GemmK& k, int epi, hipStream_t stream) {
  if (k.K % 64 != 0 || k.M <= 0) return MD_ERR_INVALID_ARG;
  // 32-bit byte offsets into A and W
  if ((uint64_t)k.M * (uint64_t)k.lda * 2 >= (1ull << 32) || (uint64_t)k.n_pad * (uint64_t)k.ldw * 2 >= (1ull << 32))
    return MD_ERR_UNSUPPORTED;
  switch (epi) {
    case MD_EPI_BIAS: return launch<MD_EPI_BIAS>(k, stream);
    case MD_EPI_GELU: return launch<MD_EPI_GELU>(k, stream);
    case MD_EPI_RESIDUAL: return launch<MD_EPI_RESIDUAL>(k, stream);
    default: return MD_ERR_INVALID_ARG;
  }
}
