// bf16 GEMM, big-tile kernel:  C = epilogue(A . W^T + b)  for M > 64 and K % 64 == 0.
//
// One workgroup = FOUR waves, one per SIMD, each owning a 128 x 128 quarter of a 256 x 256 tile of C with its 16
// accumulator blocks (256 registers) in the accumulator half of the register file, named literally by inline asm.  Per
// 16-wide K step a wave issues 16 v_mfma_f32_32x32x16_bf16 (512 matrix-pipe cycles) for 8 ds_read_b128.  Nothing else runs
// on the SIMD, so everything that is not an MFMA is a FILLER placed by hand between two MFMAs (tables below); the four
// waves run in lockstep between barriers, so fillers of one kind are spread out.
//
// Operand path (round 4): global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), no register round trip, no ds_write.
//   LDS image   TWO pair buffers of 64 KiB.  A buffer holds 64 K elements ("a pair of 32-wide slices") of the tile's 256
//               activation rows (32 KiB) and 256 weight rows (32 KiB) as 128-byte rows; logical 16-byte chunk c of row r sits
//               at physical chunk c ^ ((r >> 1) & 7), so the 16-lane groups of a ds_read_b128 touch 16 distinct bank slots.
//   LDS-DMA     an instruction moves 8 rows x 128 bytes = whole cache lines (the lane -> LDS mapping is linear: the
//               permutation sits in the per-lane SOURCE offset); a wave issues 16 per pair, two pairs ahead of the MFMAs
//               that consume them.  One 32-bit offset register per piece (constant per tile) + one scalar K offset.
//   fragments   ds_read_b128 one 16-wide K step ahead of the MFMAs that consume them, waited for with COUNTED lgkmcnt
//               (only the fragment an MFMA is first to use); LDS-DMA is counted by vmcnt, so the count sees reads only.
//   one s_barrier per pair (64 MFMAs): lgkmcnt(0) (this wave has finished reading the current buffer), vmcnt(0) (its pieces
//   of the next pair have landed), barrier: the next buffer is published, the current one is released to the DMA stream.
//
// The stream is CONTINUOUS across the tiles of a persistent workgroup (grid = one workgroup per CU, tiles blockIdx.x,
// blockIdx.x + gridDim.x, ...): the next tile's first pairs are requested under the current tile's last MFMAs and land
// under its epilogue.
//
// Epilogue (round 4): every layer kind leaves through a 4 KiB per-wave LDS transposition tile, so that a store
// instruction writes 8 rows x 128 bytes -- WHOLE lines.  (The register-only epilogue of rounds 2-3 wrote 32 rows x 32
// bytes per instruction; profiles/r04_gemm_w4_lds_dma_ablations_stamps.txt: its stores alone cost 19 % of a K <= 2048
// layer.)  Software pipelined: pass p + 1 is converted and written to the tile while pass p's 16-byte pieces are in flight
// back from it; LDS executes a wave's operations in order, which is all the ordering one tile needs.  The tile's bias
// slice (128 columns per wave) arrives by one small LDS-DMA per tile into a wave-private slot.
//
// Numerics: K is accumulated in the same order as every other tile config (sequential 16-wide steps into one fp32
// accumulator), so results are bit-identical to them.
#include "gemm_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BM = 256, BN = 256;
constexpr int PBUF = 65536;                    // one pair buffer: 256 A rows + 256 W rows of 128 bytes
constexpr int W_OFF = 32768;                   // the weight rows' half of a pair buffer
constexpr int RING = 2 * PBUF;
constexpr int XPOSE_BYTES = 32 * 128;          // per wave: 32 rows x 64 bf16
constexpr int BIAS_SLOT = 256;                 // per wave: the bias of its 128 columns
constexpr int LDS_BYTES = RING + 4 * XPOSE_BYTES + 4 * BIAS_SLOT;

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
__device__ __forceinline__ void ds_write_b64_asm(uint32_t addr, u32x2 v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void ds_write_b32_asm(uint32_t addr, uint32_t v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
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
// ablation builds: keep a value opaque / alive without emitting an instruction
template <class T>
__device__ __forceinline__ void opaque(T& v) {
  asm volatile("" : "+v"(v));
}
template <class T>
__device__ __forceinline__ void keep_alive(const T& v) {
  asm volatile("" ::"v"(v));
}

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

// ---- the filler schedule of one PAIR of slices (64 MFMAs, 64 gaps; gap g = right after MFMA g), as compile-time tables ----
// A pair is four HALVES of 16 MFMAs = the four 16-wide K steps of the pair buffer; MFMA m = 4 i + j of a half multiplies
// weight block j with activation block i.  The k-th read for a half fetches fragment kReadOrder[k] (0-3 = weight blocks j,
// 4-7 = activation blocks i) in the order the MFMAs first need them: B0 A0 B1 B2 B3 A1 A2 A3 are first used by MFMAs
// 0 0 1 2 3 4 8 12 of the consuming half.
//   ds_read      half 1's fragments in gaps 1, 3, .., 15, half 2's in 17, .., 31, half 3's in 32, 34, .., 46 (EVEN: the last
//                read of the buffer being multiplied is two MFMAs old at the barrier), the NEXT pair's half 0 in 49, .., 63
//   gap 48       lgkmcnt(0), vmcnt(0), s_barrier (see the head of the file)
//   LDS-DMA      the 16 pieces of the pair AFTER the next one, into the buffer the barrier released, at the stream positions
//                dma_pos() names (48 .. 63 = the rest of this pair, 64 .. = the first gaps of the next pair); a piece has 32+
//                gaps (1000+ cycles) to land.  MODE picks the placement (profiles/r04_gemm_w4_lds_dma_first_contact.txt).
constexpr int kReadOrder[8] = {0, 4, 1, 2, 3, 5, 6, 7};
constexpr int kFirstUse[8] = {0, 0, 1, 2, 3, 4, 8, 12};  // by read position k
constexpr int kAdvanceGap = 40;                 // the load cursor moves on here: after the last wrapped piece, before gap 48
constexpr int dma_pos(int mode, int q) {
  return mode == 1 ? (q < 8 ? 48 + 2 * q : 64 + 2 * (q - 8)) : mode == 2 ? 48 + q : mode == 3 ? 48 + 3 * q : 48 + (5 * q) / 2;
}
constexpr bool modes_ok() {
  for (int m = 1; m <= 4; ++m)
    for (int q = 0; q < 16; ++q)
      if (dma_pos(m, q) < 48 || dma_pos(m, q) - 64 >= kAdvanceGap || (q > 0 && dma_pos(m, q) <= dma_pos(m, q - 1))) return false;
  return true;
}
static_assert(modes_ok(), "pieces are issued in order, from gap 48 on, and the wrapped ones before the cursor advances");
constexpr bool is_read_gap(int g) {
  const int x = ((g % 64) + 64) % 64;
  return (x < 32 && x % 2 == 1) || (x >= 32 && x < 48 && x % 2 == 0) || (x > 48 && x % 2 == 1);
}
// stream position (relative to gap 0 of the consuming pair) of the k-th read of half h
constexpr int read_pos(int h, int k) { return h == 0 ? -15 + 2 * k : h == 1 ? 1 + 2 * k : h == 2 ? 17 + 2 * k : 32 + 2 * k; }
// which (half, k) is read in gap x of a body, encoded 8 h + k; -1: none
constexpr int read_slot(int x) {
  for (int h = 0; h < 4; ++h)
    for (int k = 0; k < 8; ++k)
      if (((read_pos(h, k) % 64) + 64) % 64 == x) return 8 * h + k;
  return -1;
}
// lgkmcnt to wait for before MFMA m of half h; -1: the MFMA introduces no new fragment
constexpr int frag_wait(int h, int m) {
  int w = -1;
  for (int k = 0; k < 8; ++k)
    if (kFirstUse[k] == m) {
      int c = 0;
      for (int g = read_pos(h, k) + 1; g < 16 * h + m; ++g) c += is_read_gap(g) ? 1 : 0;
      w = (w < 0 || c < w) ? c : w;
    }
  return w;
}
static_assert(frag_wait(0, 0) <= 15 && frag_wait(1, 12) <= 15 && frag_wait(2, 12) <= 15 && frag_wait(3, 12) <= 15, "lgkmcnt is a 4-bit counter");

// ABL: measurement builds (bit 1: no operand DMA, 2: no barrier, 3: no fragment reads, 4: no epilogue stores, 5: no
// epilogue, 6: shader-clock stamps around the waits of gap 48 and the epilogue, 7: every store into one 2 MiB window, 8: sc1
// stores, 7 + 8: sc0 sc1 stores); results are garbage with bits 1-5 or 7 alone set.
template <int EPI, int ABL = 0, int MODE = 1>
__global__ __launch_bounds__(256) void gemm_w4_kernel(const GemmK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int nwg = p.tiles_m * p.tiles_n;
  const int npair = p.K / 64;  // pairs of 32-wide slices per tile
  int grid_x = gridDim.x;      // pinned in a scalar register: re-read from the dispatch packet inside the stream it is an
  asm volatile("" : "+s"(grid_x));  // s_load + lgkmcnt(0) in the middle of a pair

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

  // ---- load cursor: one pair (64 K elements) at a time, across tile boundaries --------------------------------
  // Piece j (0-7: activations, 8-15: weights) of this wave: rows 32 (j & 7) + 8 wave + (lane >> 3) of the tile, all 128 bytes
  // of each.  The lane's LDS slot is fixed (row lane >> 3, physical chunk lane & 7), so it FETCHES the logical chunk that
  // belongs there.  Rows past M / n_pad are clamped (their products land in rows / columns nobody stores).
  uint32_t voff[16];
  int ld_tile = blockIdx.x, ld_pair = 0;
  uint32_t ld_soff = 0;
  auto set_load_tile = [&](int vv) {
    int m0, n0;
    tile_origin(vv, m0, n0);
    const int r8 = tid >> 3, c8 = (tid & 7) ^ ((r8 >> 1) & 7);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      voff[j] = (uint32_t)min(m0 + 32 * j + r8, p.M - 1) * (uint32_t)(p.lda * 2) + c8 * 16;
      voff[8 + j] = (uint32_t)min(n0 + 32 * j + r8, p.n_pad - 1) * (uint32_t)(p.ldw * 2) + c8 * 16;
    }
  };
  set_load_tile(ld_tile);
  auto advance_load_cursor = [&]() {
    ld_soff += 128;
    if (++ld_pair == npair) {
      ld_pair = 0;
      ld_soff = 0;
      // past the end of the stream the cursor stays on the last tile: the DMA keeps going into buffers
      // nobody reads any more, which keeps the loop free of "is there a next pair" tests
      if (ld_tile + grid_x < nwg) {
        ld_tile += grid_x;
        set_load_tile(ld_tile);
      }
    }
  };
  // buffer descriptors as four scalar words each (inline-asm operands); piece j lands at
  //   pair buffer + (j < 8 ? 0 : 32 KiB) + (j & 7) * 4 KiB + wave * 1 KiB + lane * 16
  auto make_rsrc = [](const void* ptr, uint32_t bytes) {
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ptr);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)ptr >> 32));
    r[2] = bytes;
    r[3] = 0x00020000u;
    return r;
  };
  const u32x4 rs_a = make_rsrc(p.A, 0xffffffffu), rs_w = make_rsrc(p.W, 0xffffffffu);
  const u32x4 rs_bias = make_rsrc(p.bias, (uint32_t)p.n_pad * 2u);  // columns past n_pad (last column tile) read as zero
  const uint32_t dma_lds = lds_base + (uint32_t)wave * 1024u;
  auto dma_piece = [&](auto j_c, uint32_t buf) {
    constexpr int J = decltype(j_c)::value;
    constexpr int OFF = J < 8 ? J * 4096 : W_OFF + (J - 8) * 4096;
    // (operands copied to locals first: clang does not capture variables that appear only as asm operands of a generic lambda)
    const uint32_t base = dma_lds + buf, vo = voff[J], so = ld_soff;  // base, so: wave-uniform
    const u32x4 rs = J < 8 ? rs_a : rs_w;
    if constexpr (!(ABL & 2))
      asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds" ::"s"(base), "i"(OFF), "v"(vo), "s"(rs), "s"(so) : "memory", "scc");
  };

  // ---- fragment reads: K step s (0-3) of a pair = logical chunks 2 s (lanes 0-31) and 2 s + 1 (lanes 32-63) ----------
  const uint32_t swz8 = (l31 >> 1) & 7;
  uint32_t da[4], db[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    da[st] = lds_base + (wm * 128 + l31) * 128 + (((2 * st + hi) ^ swz8) * 16);
    db[st] = lds_base + W_OFF + (wn * 128 + l31) * 128 + (((2 * st + hi) ^ swz8) * 16);
  }
  bf16x8 fa[2][4], fb[2][4];  // [fragment set][32-row block]
  if constexpr (ABL & 8) {
#pragma unroll
    for (int j = 0; j < 4; ++j) fa[0][j] = fa[1][j] = fb[0][j] = fb[1][j] = bf16x8{0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  }
  auto read_frag = [&](auto set_c, auto q_c, uint32_t a_addr, uint32_t b_addr) {
    constexpr int SET = decltype(set_c)::value, Q = decltype(q_c)::value;
    if constexpr (ABL & 8) {
      if constexpr (Q < 4) opaque(fb[SET][Q]); else opaque(fa[SET][Q - 4]);
    } else if constexpr (Q < 4)
      ds_read_b128<Q * 4096>(fb[SET][Q], b_addr);
    else
      ds_read_b128<(Q - 4) * 4096>(fa[SET][Q - 4], a_addr);
  };

  acc_reserve();

  // ---- per-wave LDS next to the ring: the transposition tile and the bias slot ------------------------------------
  const uint32_t tile_lds = lds_base + RING + wave * XPOSE_BYTES;
  const uint32_t bias_lds = lds_base + RING + 4 * XPOSE_BYTES + wave * BIAS_SLOT;
  const bool has_bias = p.bias != nullptr;  // uniform
  if (!has_bias) ds_write_b32_asm(bias_lds + lane * 4, 0u);  // a layer without bias adds the zeros of a slot that is never refilled
  uint32_t bias_soff = 0;  // byte offset of the current tile's (wave's) first bias column
  auto dma_bias = [&]() {
    const uint32_t base = bias_lds, vo = (uint32_t)lane * 4u, so = bias_soff;
    const u32x4 rs = rs_bias;
    if (has_bias) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(base), "v"(vo), "s"(rs), "s"(so) : "memory");
  };

  // ---- stream prologue: pair 0 whole, and the pieces of pair 1 that the steady state issues at the END of a body -------
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  uint32_t pair_cur = 0, pair_wr = PBUF;  // byte offsets of the buffer being multiplied / the other one
  // the first fragments (K step 0) of the pair in pair_cur, all eight, waited for: stream start and after every epilogue
  auto read_first_frags = [&]() {
    const uint32_t a0 = da[0] + pair_cur, b0 = db[0] + pair_cur;
    static_for<0, 8>([&](auto q) { read_frag(I0{}, std::integral_constant<int, kReadOrder[decltype(q)::value]>{}, a0, b0); });
    wait_lgkm<0>();
    MD_PIN();
  };
  static_for<0, 16>([&](auto j) { dma_piece(j, pair_cur); });
  advance_load_cursor();
  static_for<0, 16>([&](auto j) {
    if constexpr (dma_pos(MODE, decltype(j)::value) < 64) dma_piece(j, pair_wr);
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_first_frags();

  // measurement build (ABL & 64): shader-clock stamps around the waits of gap 48 (s_memtime is counted by lgkmcnt: only here,
  // where the counter is drained anyway)
  uint32_t st_lgkm = 0, st_vm = 0, st_bar = 0, st_n = 0;
  uint64_t st_first = 0, st_last = 0, st_epi = 0;

  // One pair = 64 MFMAs = 64 gaps.  pair_cur = the buffer being multiplied, pair_wr = the other one: it receives the late
  // pieces of the NEXT pair in the first gaps, is published by the barrier in gap 48 and read from gap 49 on; from gap 48 on
  // pair_cur receives the pair after that.  ONE straight-line body but for the cursor's once-per-tile branch in gap 40;
  // past the end of the stream the fillers keep running on data nobody reads.
  auto pair_body = [&](auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;  // first pair of a tile: accumulate onto zero, request the tile's bias slice
    const uint32_t a1 = da[1] + pair_cur, b1 = db[1] + pair_cur, a2 = da[2] + pair_cur, b2 = db[2] + pair_cur;
    const uint32_t a3 = da[3] + pair_cur, b3 = db[3] + pair_cur, an = da[0] + pair_wr, bn = db[0] + pair_wr;
    const uint32_t buf_cur = pair_cur, buf_nxt = pair_wr;
    static_for<0, 64>([&](auto xc) {
      constexpr int X = decltype(xc)::value, H = X / 16, M = X % 16, I = M / 4, J = M % 4, SET = H & 1;
      if constexpr (X == 48) {
        uint64_t t0 = 0;
        if constexpr (ABL & 64) t0 = __builtin_readcyclecounter();
        if constexpr (!(ABL & 8)) wait_lgkm<0>();
        if constexpr (ABL & 64) { const uint64_t t = __builtin_readcyclecounter(); st_lgkm += (uint32_t)(t - t0); t0 = t; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (ABL & 64) { const uint64_t t = __builtin_readcyclecounter(); st_vm += (uint32_t)(t - t0); t0 = t; }
        if constexpr (!(ABL & 4)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (ABL & 64) {
          const uint64_t t = __builtin_readcyclecounter();
          st_bar += (uint32_t)(t - t0);
          if (st_n == 0) st_first = t;
          st_last = t;
          ++st_n;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
      // counted wait: the fragments this MFMA is the first to use have landed, younger reads stay in flight
      if constexpr (frag_wait(H, M) >= 0 && !(ABL & 8)) wait_lgkm<frag_wait(H, M)>();
      mfma_acc<M, FIRST && H == 0>(fb[SET][J], fa[SET][I]);
      MD_PIN();
      constexpr int RS = read_slot(X);
      if constexpr (RS >= 0) {
        using Q = std::integral_constant<int, kReadOrder[RS % 8]>;
        if constexpr (RS / 8 == 1) read_frag(I1{}, Q{}, a1, b1);
        if constexpr (RS / 8 == 2) read_frag(I0{}, Q{}, a2, b2);
        if constexpr (RS / 8 == 3) read_frag(I1{}, Q{}, a3, b3);
        if constexpr (RS / 8 == 0) read_frag(I0{}, Q{}, an, bn);
      }
      static_for<0, 16>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        if constexpr (dma_pos(MODE, Q) == X) dma_piece(qc, buf_cur);        // pair p + 2 -> the buffer released in gap 48
        if constexpr (dma_pos(MODE, Q) - 64 == X) dma_piece(qc, buf_nxt);   // late pieces of pair p + 1
      });
      if constexpr (FIRST && X == 41) dma_bias();  // the slot's previous contents went to registers in the last epilogue
      if constexpr (X == kAdvanceGap) advance_load_cursor();
      MD_PIN();
    });
    const uint32_t t = pair_cur;
    pair_cur = pair_wr;
    pair_wr = t;
  };

  // ---- tile loop ------------------------------------------------------------------------------
  for (int vtile = blockIdx.x; vtile < nwg; vtile += grid_x) {
    int m0c, n0c;
    tile_origin(vtile, m0c, n0c);
    const int wm0 = m0c + wm * 128, wn0 = n0c + wn * 128;
    bias_soff = (uint32_t)wn0 * 2u;
    // the accumulators are (re)defined by the first K step of every tile: nothing is carried from
    // one tile to the next in them
    pair_body(std::true_type{});
    for (int u = 1; u < npair; ++u) pair_body(std::false_type{});
    wait_lgkm<0>();
    MD_PIN();
    uint64_t te0 = 0;
    if constexpr (ABL & 64) { te0 = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

    // ---- epilogue of tile vtile (the next tile's first pairs are already in the ring / in flight) ----------------------
    // block X = 4 i + j, register r: row m = 32 i + l31, col n = 32 j + 8 (r >> 2) + 4 hi + (r & 3)   within the wave's quarter
    if constexpr (!(ABL & 32)) {
    // This lane's 64 bias values (column 32 j + 8 q + 4 hi + e), unpacked ONCE per tile from the wave's slot.  (npair == 1: the
    // slot's DMA was waited for by gap 48's vmcnt(0) like every other piece.)
    md_f32x2 bias_f[4][4][2];
    {
      u32x2 bw[4][4];
      static_for<0, 16>([&](auto jq) {
        constexpr int j = decltype(jq)::value / 4, q = decltype(jq)::value % 4;
        ds_read_b64_u32<(32 * j + 8 * q) * 2>(bw[j][q], bias_lds + 8 * hi);
      });
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results are not readable before they retire
      wait_lgkm<0>();
      MD_PIN();
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bias_f[j][q][0] = md_f32x2{lo_bf(bw[j][q][0]), hi_bf(bw[j][q][0])};
          bias_f[j][q][1] = md_f32x2{lo_bf(bw[j][q][1]), hi_bf(bw[j][q][1])};
        }
    }
    // C (and a residual R) are addressed through buffer resources with num_records = M rows: rows past M read as zero / are
    // dropped by the range check (no exec masking, no branch); an address is one 32-bit offset.
    const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)min((uint64_t)p.M * (uint64_t)p.ldc * 2, (uint64_t)0xffffffffu), 0x00020000);
    // ---- the generic path: eight passes of 32 rows x 64 columns (row block i, column half jp) through the wave's LDS tile ----
    // write side, on the accumulator layout: bias add + ONE bf16 rounding (F.linear's rounding point), a lane's quad of 4
    // columns = 8 bytes at row l31, chunk (4 jj + q) ^ (l31 & 7), half hi.  Read side: piece q of this lane = row 8 q +
    // (lane >> 3), columns 8 (lane & 7) .. + 7 of the pass: 8 lanes cover a row's 128 bytes, an instruction 8 whole lines.
    // GELU / the residual add work on those 16-byte pieces.  Pass p + 1 is converted and written while pass p's pieces are
    // in flight back (LDS executes a wave's operations in order: read p, write p + 1, read p + 1 need no waits between them).
    auto lds_epilogue = [&]() {
      const int lane_row = lane >> 3, lane_col = (lane & 7) * 8;
      const bool col_ok0 = wn0 + lane_col < p.n_store, col_ok1 = wn0 + 64 + lane_col < p.n_store;  // columns past n_store: last column tile only
      const uint32_t col_bytes = (uint32_t)(wn0 + lane_col) * 2u;
      const uint32_t c_base = (uint32_t)(wm0 + lane_row) * (uint32_t)(p.ldc * 2) + col_bytes;
      const uint32_t c_step = (uint32_t)p.ldc * 16u;  // 8 rows
      auto store_off = [&](int i, int q, int jp) -> uint32_t {
        const uint32_t off = c_base + (uint32_t)(4 * i + q) * c_step;
        return ((jp ? col_ok1 : col_ok0) ? off : 0xfffff000u) + 128 * jp;  // out of range: dropped
      };
      // residual layers: the second operand in the same whole-line pieces, prefetched one pass ahead; a broadcast residual
      // (the ViT's position embedding) wraps at res_row_mod (>= 256: at most one wrap per tile)
      const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc((void*)p.R, 0, (int)min((uint64_t)(p.res_row_mod ? p.res_row_mod : p.M) * (uint64_t)p.ldr * 2, (uint64_t)0xffffffffu), 0x00020000);
      const uint32_t wrap = p.res_row_mod ? (uint32_t)p.res_row_mod : 0x7fffffffu;
      const uint32_t r_row0 = (uint32_t)(wm0 + lane_row) % wrap;
      auto load_residual = [&](int pass, u32x4 (&rv)[4]) {
        const int i = pass >> 1, jp = pass & 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t row = r_row0 + 8u * (uint32_t)(4 * i + q);
          row -= (row >= wrap) ? wrap : 0u;
          const uint32_t off = (jp ? col_ok1 : col_ok0) ? row * (uint32_t)(p.ldr * 2) + col_bytes : 0xfffff000u;
          rv[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, off + 128 * jp, 0, 0));
        }
      };
      auto convert_write = [&](auto pc) {
        constexpr int PASS = decltype(pc)::value, i = PASS >> 1, jp = PASS & 1;
        static_for<0, 8>([&](auto jq) {
          constexpr int jj = decltype(jq)::value / 4, q = decltype(jq)::value % 4, j = 2 * jp + jj, base = 16 * (4 * i + j) + 4 * q;
          const md_f32x2 x0 = md_f32x2{acc_read<base + 0>(), acc_read<base + 1>()} + bias_f[j][q][0];
          const md_f32x2 x1 = md_f32x2{acc_read<base + 2>(), acc_read<base + 3>()} + bias_f[j][q][1];
          constexpr int ch = 4 * jj + q;
          ds_write_b64_asm(tile_lds + l31 * 128 + ((ch ^ (l31 & 7)) * 16) + hi * 8, u32x2{pack_bf16x2(x0[0], x0[1]), pack_bf16x2(x1[0], x1[1])});
        });
      };
      auto read_pieces = [&](u32x4 (&tv)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int idx = q * 64 + lane, row = idx >> 3, ch = idx & 7;
          ds_read_b128_u32(tv[q], tile_lds + row * 128 + ((ch ^ (row & 7)) * 16));
        }
      };
      u32x4 tv[2][4], rres[2][4];
      if constexpr (EPI == MD_EPI_RESIDUAL) load_residual(0, rres[0]);
      convert_write(I0{});
      read_pieces(tv[0]);
      MD_PIN();
      static_for<0, 8>([&](auto pc) {
        constexpr int PASS = decltype(pc)::value, i = PASS >> 1, jp = PASS & 1;
        if constexpr (PASS + 1 < 8) {
          if constexpr (EPI == MD_EPI_RESIDUAL) load_residual(PASS + 1, rres[(PASS + 1) & 1]);
          convert_write(std::integral_constant<int, PASS + 1>{});
          wait_lgkm<8>();  // this pass's four reads are back; the next pass's eight writes may still be on their way
        } else {
          wait_lgkm<0>();
        }
        MD_PIN();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u32x4 v = tv[PASS & 1][q];
          if constexpr (EPI == MD_EPI_GELU || EPI == MD_EPI_QKV_ROPE) {
            if (wn0 + 64 * jp >= p.gelu_from) {  // wave-uniform: gelu_from is a multiple of 64 (fused [qkv | fc1] layers)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const md_f32x2 ge = gelu_tanh_f32x2(md_f32x2{lo_bf(v[e]), hi_bf(v[e])});
                v[e] = pack_bf16x2(ge[0], ge[1]);
              }
            }
          }
          if constexpr (EPI == MD_EPI_RESIDUAL) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v[e] = pack_bf16x2(lo_bf(rres[PASS & 1][q][e]) + lo_bf(v[e]), hi_bf(rres[PASS & 1][q][e]) + hi_bf(v[e]));
          }
          const uint32_t off = store_off(i, q, jp);
          if constexpr ((ABL & (128 | 256)) == 128) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_c, off & 0x1fffffu, 0, 0);  // every store into one 2 MiB window
          else if constexpr ((ABL & (128 | 256)) == 256) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_c, off, 0, 16);       // sc1
          else if constexpr ((ABL & (128 | 256)) == 384) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_c, off, 0, 17);       // sc0 sc1
          else if constexpr (!(ABL & 16)) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_c, off, 0, 0);
          else keep_alive(v);
        }
        MD_PIN();
        if constexpr (PASS + 1 < 8) read_pieces(tv[(PASS + 1) & 1]);
        MD_PIN();
      });
    };
    // ---- MD_EPI_QKV_ROPE: the q / k / v sections of the decoder's fused layer at prefill ------------------------------
    // A wave's 128 columns are two heads of ONE section (sections are n_heads x 64 wide, a multiple of 128).  Per head the
    // first 32 features are rotated: the reference reads them half-split (re = x[d], im = x[16 + d]) and writes them
    // interleaved (rope.py:37-46).  In the accumulator layout a lane holds quads q = 0..3 of column block j, i.e. features
    // 8 q + 4 hi + e: re (q = 0, 1) and im (q + 2) of a pair sit in the SAME lane, and the rotated pairs of quad q are the 8
    // consecutive output features 16 q + 8 hi ..: a 16-byte piece with no lane exchange.  fp32 arithmetic on the bf16-rounded
    // layer output with separately rounded mul, mul, sub / add, as torch evaluates it (bit-equal to rope_kv_kernel).
    // q stays in the activation (rotated), k and v go straight to the KV slab (text.py:45-46): one pass over the bytes
    // instead of the GEMM's stores + rope_kv_kernel's load and store of every q / k / v element.
    auto rope_tile = [&]() {
      const int sec = wn0 / p.rope_d;               // 0 q, 1 k, 2 v   (wave-uniform)
      const int head0 = (wn0 - sec * p.rope_d) >> 6;
      bf16_t* slab = (sec == 1) ? p.kslab : p.vslab;
      const __amdgpu_buffer_rsrc_t rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, (int)p.slab_bytes, 0x00020000);
      static_for<0, 4>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int m = wm0 + 32 * i + l31, mc = min(m, p.M - 1);
        const uint32_t row_off = (uint32_t)m * (uint32_t)(p.ldc * 2) + (uint32_t)(wn0 + 8 * hi) * 2u;
        uint32_t kv_off = 0xfffff000u;  // rows past M: dropped by the range check
        if (sec != 0 && m < p.M) kv_off = p.rope_kv[mc] + (uint32_t)hi * 16u;
        f32x4 cs[4];  // (cos, sin) of features 4 hi + {0,1}, {2,3}, 8 + 4 hi + {0,1}, {2,3}
        if (sec != 2) {
          const f32x4* cp = (const f32x4*)(p.rope_cs + (int64_t)mc * 32);
          cs[0] = cp[2 * hi];
          cs[1] = cp[2 * hi + 1];
          cs[2] = cp[4 + 2 * hi];
          cs[3] = cp[4 + 2 * hi + 1];
        }
        static_for<0, 2>([&](auto jc) {
          constexpr int jp = decltype(jc)::value;
          const uint32_t head_off = (uint32_t)(head0 + jp) * (uint32_t)p.rope_ctx * 128u;  // slab: [head][position][64] bf16
          auto put = [&](const u32x4& v, int col_in_head_bytes) {  // col_in_head_bytes: immediate-sized constant
            if (sec == 0) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_c, row_off + 128 * jp + col_in_head_bytes, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_s, kv_off + col_in_head_bytes, head_off, 0);
          };
          // features 0..31 of the head: column block j = 2 jp
          {
            constexpr int j = 2 * jp, base = 16 * (4 * i + j);
            float x[4][4];
            static_for<0, 4>([&](auto qc) {
              constexpr int q = decltype(qc)::value;
              const md_f32x2 u0 = md_f32x2{acc_read<base + 4 * q + 0>(), acc_read<base + 4 * q + 1>()} + bias_f[j][q][0];
              const md_f32x2 u1 = md_f32x2{acc_read<base + 4 * q + 2>(), acc_read<base + 4 * q + 3>()} + bias_f[j][q][1];
              const uint32_t w0 = pack_bf16x2(u0[0], u0[1]), w1 = pack_bf16x2(u1[0], u1[1]);  // the layer's bf16 output
              x[q][0] = lo_bf(w0); x[q][1] = hi_bf(w0); x[q][2] = lo_bf(w1); x[q][3] = hi_bf(w1);
            });
            if (sec != 2) {
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                u32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float re = x[q][e], im = x[q + 2][e];
                  const float c = cs[2 * q + (e >> 1)][2 * (e & 1)], sn = cs[2 * q + (e >> 1)][2 * (e & 1) + 1];
                  float o_re, o_im;
                  md_rope_pair(re, im, c, sn, o_re, o_im);
                  v[e] = pack_bf16x2(o_re, o_im);
                }
                put(v, q * 32);  // features 16 q + 8 hi .. + 7 (the 8 hi is part of row_off / kv_off)
              }
            } else {
              // v: no rotation; the standard 16-byte pieces (quads q0, q0 + 1 of the two lane halves side by side)
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                const uint32_t a0 = pack_bf16x2(x[2 * t][0], x[2 * t][1]), a1 = pack_bf16x2(x[2 * t][2], x[2 * t][3]);
                const uint32_t b0 = pack_bf16x2(x[2 * t + 1][0], x[2 * t + 1][1]), b1 = pack_bf16x2(x[2 * t + 1][2], x[2 * t + 1][3]);
                const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                put(u32x4{s0[0], s1[0], s0[1], s1[1]}, t * 32);
              }
            }
          }
          // features 32..63: column block j = 2 jp + 1, never rotated (rot_dim 32)
          {
            constexpr int j = 2 * jp + 1, base = 16 * (4 * i + j);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              uint32_t w[4];
              static_for<0, 2>([&](auto hc) {
                constexpr int hq = decltype(hc)::value;  // quads 2 t and 2 t + 1
                auto rd = [&](auto tt) {
                  constexpr int q = 2 * decltype(tt)::value + hq;
                  const md_f32x2 u0 = md_f32x2{acc_read<base + 4 * q + 0>(), acc_read<base + 4 * q + 1>()} + bias_f[j][q][0];
                  const md_f32x2 u1 = md_f32x2{acc_read<base + 4 * q + 2>(), acc_read<base + 4 * q + 3>()} + bias_f[j][q][1];
                  w[2 * hq] = pack_bf16x2(u0[0], u0[1]);
                  w[2 * hq + 1] = pack_bf16x2(u1[0], u1[1]);
                };
                if (t == 0) rd(std::integral_constant<int, 0>{}); else rd(std::integral_constant<int, 1>{});
              });
              const auto s0 = __builtin_amdgcn_permlane32_swap(w[0], w[2], false, false);
              const auto s1 = __builtin_amdgcn_permlane32_swap(w[1], w[3], false, false);
              put(u32x4{s0[0], s1[0], s0[1], s1[1]}, 64 + t * 32);
            }
          }
        });
        MD_PIN();
      });
    };
    if constexpr (EPI == MD_EPI_QKV_ROPE) {
      if (wn0 < 3 * p.rope_d) rope_tile(); else lds_epilogue();
    } else {
      lds_epilogue();
    }
    }
    // the next tile's first fragments again (the copy read before the epilogue was not kept: 32
    // registers the epilogue does not have to carry); its first pair was published by the last barrier above
    read_first_frags();
    if constexpr (ABL & 64) { st_epi += __builtin_readcyclecounter() - te0; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA of the run-on stream in flight at the end of the wave
  if constexpr (ABL & 64) {
    if (lane == 0 && p.slabs != nullptr) {
      float* o = p.slabs + (blockIdx.x * 4 + wave) * 8;
      o[0] = (float)st_n; o[1] = (float)(st_last - st_first); o[2] = (float)st_lgkm; o[3] = (float)st_vm; o[4] = (float)st_bar; o[5] = (float)st_epi;
    }
  }
}

int g_w4_grid = 0;     // md_gemm_set_tuning "w4_grid": persistent workgroups per launch (0 = one per CU); a multiple of 8
// md_gemm_set_tuning "w4_variant" (MD_W4_VARIANT): low 4 bits = placement of the LDS-DMA pieces (MODE, 0 = the default;
// the others exist for the bias epilogue only), the rest 16 * ABL (measurement builds, bias epilogue only)
int g_w4_variant = [] { const char* e = getenv("MD_W4_VARIANT"); return (e && *e) ? atoi(e) : 0; }();
uint64_t g_w4_debug = 0;  // measurement builds: device buffer for the in-kernel stamps (md_gemm_set_tuning "w4_dbg_lo" / "w4_dbg_hi")
constexpr int kDefaultMode = 1;

template <int EPI, int ABL = 0, int MODE = kDefaultMode>
md_status launch(const GemmK& k, hipStream_t stream) {
  auto fn = gemm_w4_kernel<EPI, ABL, MODE>;
  MD_TRY(md_ensure_dynamic_lds((const void*)fn, LDS_BYTES));
  GemmK kk = k;
  if constexpr (ABL & 64) kk.slabs = (float*)(uintptr_t)g_w4_debug;
  kk.tiles_m = (k.M + BM - 1) / BM;
  kk.tiles_n = (k.n_store + BN - 1) / BN;
  const int nwg = kk.tiles_m * kk.tiles_n;
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? (n / 8) * 8 : 256;  // a multiple of 8 keeps (sequence number % 8) == XCD for the tile-order remap
  }();
  // one persistent workgroup per CU -- or fewer ("w4_grid"): a workgroup owns its CU (145 KiB of LDS, the whole register file),
  // so a smaller grid leaves whole CUs to kernels of another stream (the pipelined engine's decode steps)
  const int gx = std::min(nwg, g_w4_grid > 0 ? std::min(g_w4_grid, n_cu) : n_cu);
  hipLaunchKernelGGL(fn, dim3(gx), dim3(256), LDS_BYTES, stream, kk);
  return md_launch_status();
}

}  // namespace

void md_gemm_w4_set_variant(int v) { g_w4_variant = v; }
void md_gemm_w4_set_debug(int half, uint32_t v) { g_w4_debug = half ? ((g_w4_debug & 0xffffffffull) | ((uint64_t)v << 32)) : ((g_w4_debug & ~0xffffffffull) | v); }
void md_gemm_w4_set_grid(int v) { g_w4_grid = v > 0 ? std::max(8, v / 8 * 8) : 0; }

bool md_gemm_w4_takes(const GemmK& k, int epi) {
  if (k.K % 64 != 0 || k.M <= 0) return false;
  // 32-bit byte offsets into A and W
  if ((uint64_t)k.M * (uint64_t)k.lda * 2 >= (1ull << 32) || (uint64_t)k.n_pad * (uint64_t)k.ldw * 2 >= (1ull << 32)) return false;
  // C (and a residual R) are addressed through buffer resources with 32-bit byte offsets: the rows of the last, partial
  // row tile must stay below the out-of-range sentinel without wrapping
  const uint64_t lim = 0xfffff000ull, rows = (uint64_t)k.M + 256;
  if (rows * (uint64_t)k.ldc * 2 >= lim) return false;
  if (epi == MD_EPI_RESIDUAL) {
    if (k.res_row_mod != 0 && k.res_row_mod < 256) return false;  // a broadcast residual wraps at most once inside a tile
    if ((k.res_row_mod ? (uint64_t)k.res_row_mod : rows) * (uint64_t)k.ldr * 2 >= lim) return false;
  }
  return true;
}

md_status md_gemm_w4_launch(const GemmK& k, int epi, hipStream_t stream) {
  if (k.K % 64 != 0 || k.M <= 0) return MD_ERR_INVALID_ARG;
  if (!md_gemm_w4_takes(k, epi)) return MD_ERR_UNSUPPORTED;
  const int mode = g_w4_variant & 15, abl = g_w4_variant >> 4;  // (measurement codes: MODE + 16 * ABL)
#ifdef MD_W4_ABLATIONS  // measurement builds only (MD_W4_ABLATIONS=1 python -c "import __graft_entry__ as g; g.build()")
  if (epi == MD_EPI_BIAS && abl != 0) {
    switch (256 * mode + abl) {
      case 256 * 1 + 2: return launch<MD_EPI_BIAS, 2, 1>(k, stream);
      case 256 * 1 + 4: return launch<MD_EPI_BIAS, 4, 1>(k, stream);
      case 256 * 1 + 8: return launch<MD_EPI_BIAS, 8, 1>(k, stream);
      case 256 * 1 + 14: return launch<MD_EPI_BIAS, 14, 1>(k, stream);
      case 256 * 1 + 16: return launch<MD_EPI_BIAS, 16, 1>(k, stream);
      case 256 * 1 + 32: return launch<MD_EPI_BIAS, 32, 1>(k, stream);
      case 256 * 1 + 46: return launch<MD_EPI_BIAS, 46, 1>(k, stream);
      case 256 * 1 + 64: return launch<MD_EPI_BIAS, 64, 1>(k, stream);
      case 256 * 1 + 128: return launch<MD_EPI_BIAS, 128, 1>(k, stream);
      case 256 * 1 + 256: return launch<MD_EPI_BIAS, 256, 1>(k, stream);
      case 256 * 1 + 384: return launch<MD_EPI_BIAS, 384, 1>(k, stream);
      default: return MD_ERR_INVALID_ARG;
    }
  }
#endif
  if (abl != 0) return MD_ERR_INVALID_ARG;
  if (mode != 0 && mode != kDefaultMode) {
    if (epi != MD_EPI_BIAS) return MD_ERR_INVALID_ARG;
    switch (mode) {
      case 2: return launch<MD_EPI_BIAS, 0, 2>(k, stream);
      case 3: return launch<MD_EPI_BIAS, 0, 3>(k, stream);
      case 4: return launch<MD_EPI_BIAS, 0, 4>(k, stream);
      default: return MD_ERR_INVALID_ARG;
    }
  }
  switch (epi) {
    case MD_EPI_BIAS: return launch<MD_EPI_BIAS>(k, stream);
    case MD_EPI_GELU: return launch<MD_EPI_GELU>(k, stream);
    case MD_EPI_RESIDUAL: return launch<MD_EPI_RESIDUAL>(k, stream);
    case MD_EPI_QKV_ROPE: return launch<MD_EPI_QKV_ROPE>(k, stream);
    default: return MD_ERR_INVALID_ARG;
  }
}
