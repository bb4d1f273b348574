// bf16 GEMM, big-tile kernel:  C = epilogue(A . W^T + b)  for M > 64 and K % 64 == 0.
//
// One workgroup = FOUR waves, one per SIMD, each owning a 128 x 128 quarter of a 256 x 256 tile of C as 8 x 8 accumulator
// blocks of v_mfma_f32_16x16x32_bf16 (256 registers) in the accumulator half of the register file, named literally by
// inline asm.  The 16x16x32 shape (round 4): with nothing else running the chip holds 2.07 GHz under it and 1.78 GHz under
// 32x32x16 (profiles/r04_mfma_power_probe.txt) -- it reads and writes each accumulator once per 32 K elements instead of
// once per 16, and on a power-limited chip energy per FLOP is rate.  Per 32-wide K step a wave issues 64 MFMAs (1024
// matrix-pipe cycles) for 16 ds_read_b128.  Nothing else runs on the SIMD, so everything that is not an MFMA is a FILLER
// placed by hand between two MFMAs (tables below); the four waves run in lockstep between barriers.
//
// Operand path: global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), no register round trip, no ds_write.
//   LDS image   TWO pair buffers of 64 KiB.  A buffer holds 64 K elements (two K steps) of the tile's 256 activation rows
//               (32 KiB) and 256 weight rows (32 KiB) as 128-byte rows; logical 16-byte chunk c of row r sits at physical
//               chunk c ^ ((r >> 1) & 7), so the 16-lane groups of a ds_read_b128 touch 16 distinct bank slots.
//   LDS-DMA     an instruction moves 8 rows x 128 bytes = whole cache lines (the lane -> LDS mapping is linear: the
//               permutation sits in the per-lane SOURCE offset); a wave issues 16 per pair, two pairs ahead of the MFMAs
//               that consume them.  One 32-bit offset register per piece (constant per tile) + one scalar K offset.  A
//               wave's pieces are grouped four to a 4 KiB LDS block: M0 is written once per block and the piece is picked
//               by the instruction's immediate offset (the M0 write and its wait state were 8-9 of a piece's ~20 issue
//               cycles: profiles/r04_dma_issue_probe.txt); the immediate also moves the global address, which the
//               piece's offset register takes back out.
//   fragments   a 16 x 32 fragment = one ds_read_b128 (lane: row lane & 15, chunk 4 s + (lane >> 4) of K step s), read one
//               K step ahead of the MFMAs that consume it and waited for with COUNTED lgkmcnt (only the fragment an MFMA is
//               first to use); LDS-DMA is counted by vmcnt, so the count sees reads only.
//   one s_barrier per pair (128 MFMAs), in the middle: every read of the current buffer is issued in the first 46 gaps, so
//   at gap 64 lgkmcnt(0) (this wave has finished reading it), vmcnt(0) (its pieces of the next pair have landed), barrier:
//   the next buffer is published, the current one is released to the DMA stream.
//
// The stream is CONTINUOUS across the tiles of a persistent workgroup (grid = one workgroup per CU, tiles blockIdx.x,
// blockIdx.x + gridDim.x, ...): the next tile's first pairs are requested under the current tile's last MFMAs and land
// under its epilogue.
//
// Epilogue: every layer kind leaves through a 4 KiB per-wave LDS transposition tile, so that a store instruction writes
// 8 rows x 128 bytes -- whole lines.  Software pipelined: pass p + 1 is converted and written to the tile while pass p's
// 16-byte pieces are in flight back from it; LDS executes a wave's operations in order, which is all the ordering one tile
// needs.  The tile's bias slice (128 columns per wave) arrives by one small LDS-DMA per tile into a wave-private slot.
// The decoder's fused [q | k | v | fc1] layer at prefill rotates q and k (rope.py:20-48) on the write side of the
// transposition and sends k / v pieces straight to the KV slab (text.py:45-46).
//
// Numerics: fp32 accumulation over K in 32-wide MFMA steps, bias add in fp32, ONE rounding to bf16 (F.linear's rounding
// point).  (Rounds 2-3 used 32x32x16 like the other tile configs and were bit-identical to them; the summation inside an
// MFMA differs between the shapes, so this kernel agrees with them to fp32 rounding of the accumulator, not bitwise.)
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
__device__ __forceinline__ void ds_write_b128_asm(uint32_t addr, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
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

// The 64 accumulator blocks of a wave (256 registers) live in a[0:255], OWNED BY INLINE ASM: block X is
// a[4X : 4X+3].  As compiler-visible values they made the register allocator shuffle and spill around every
// control-flow join; named literally they cost it nothing.  acc_reserve() lists them as clobbers once (which
// also makes the kernel descriptor allocate them); the build audits that no compiler-generated v_accvgpr_* /
// scratch instruction appears in the kernel (see _lib.build_library).
#define MD_A16(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
__device__ __forceinline__ void acc_reserve() {
  asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", MD_A16(1), MD_A16(2), MD_A16(3), MD_A16(4),
               MD_A16(5), MD_A16(6), MD_A16(7), MD_A16(8), MD_A16(9), MD_A16(10), MD_A16(11), MD_A16(12), MD_A16(13), MD_A16(14),
               MD_A16(15), MD_A16(16), MD_A16(17), MD_A16(18), MD_A16(19), MD_A16(20), MD_A16(21), MD_A16(22), MD_A16(23),
               MD_A16(24), "a250", "a251", "a252", "a253", "a254", "a255");
}
// block X (+)= W-fragment . A-fragment^T      (first operand = weight rows: a lane then holds token row lane & 15 and the
// four consecutive columns 4 (lane >> 4) .. + 3 of the block)
template <int X, bool FIRST>
__device__ __forceinline__ void mfma_acc(const bf16x8& wf, const bf16x8& af) {
  if constexpr (FIRST)
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(wf), "v"(af), "i"(4 * X), "i"(4 * X + 3));
  else
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(wf), "v"(af), "i"(4 * X), "i"(4 * X + 3));
}
template <int N>
__device__ __forceinline__ float acc_read() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(v) : "i"(N));
  return v;
}

// ---- the filler schedule of one PAIR (two K steps = 128 MFMAs, 128 gaps; gap g = right after MFMA g), as compile-time tables ----
// MFMA m of a K step multiplies activation block i = m / 8 with weight block j = m % 8 (odd i: 7 - m % 8, so that only one
// operand changes between neighbours).  Fragment f: 0-7 = weight blocks j, 8-15 = activation blocks i.  The k-th read of a
// step fetches kReadOrder[k], in the order the MFMAs first need them (kFirstUse[k]).
//   ds_read      K step 1's fragments (current buffer) in gaps 0, 3, .., 45; the NEXT pair's K step 0 (other buffer) in gaps
//                65, 68, .., 110
//   gap 64       lgkmcnt(0), vmcnt(0), s_barrier (see the head of the file)
//   LDS-DMA      the 16 pieces of the pair AFTER the next one, into the buffer the barrier released, at the stream positions
//                dma_pos() names (64 .. 127 = the rest of this pair, 128 .. = the first gaps of the next pair).  MODE picks
//                the placement.
constexpr int kReadOrder[16] = {0, 8, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15};
constexpr int kFirstUse[16] = {0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 16, 24, 32, 40, 48, 56};  // by read position k
constexpr int kAdvanceGap = 50;                 // the load cursor moves on here: after the last wrapped piece, before gap 64
constexpr int kBiasGap = 52;                    // first pair of a tile: the tile's bias slice is requested here
constexpr int dma_pos(int mode, int q) { return mode == 1 ? 66 + 4 * q : mode == 2 ? 64 + 6 * q : mode == 3 ? 66 + 2 * q : 67 + 3 * q; }
constexpr bool modes_ok() {
  for (int m = 1; m <= 4; ++m)
    for (int q = 0; q < 16; ++q)
      if (dma_pos(m, q) < 64 || dma_pos(m, q) - 128 >= kAdvanceGap || (q > 0 && dma_pos(m, q) <= dma_pos(m, q - 1))) return false;
  return true;
}
static_assert(modes_ok(), "pieces are issued in order, from gap 64 on, and the wrapped ones before the cursor advances");
// stream position (relative to gap 0 of the consuming pair) of the k-th read of K step h
constexpr int read_pos(int h, int k) { return h == 0 ? -63 + 3 * k : 3 * k; }
constexpr bool is_read_gap(int g) {
  const int x = ((g % 128) + 128) % 128;
  return (x < 48 && x % 3 == 0) || (x >= 65 && x < 113 && (x - 65) % 3 == 0);
}
// which (step, k) is read in gap x of a body, encoded 16 h + k; -1: none
constexpr int read_slot(int x) {
  for (int h = 0; h < 2; ++h)
    for (int k = 0; k < 16; ++k)
      if (((read_pos(h, k) % 128) + 128) % 128 == x) return 16 * h + k;
  return -1;
}
// lgkmcnt to wait for before MFMA m of K step 0 (step 1's fragments are all covered by gap 64's lgkmcnt(0)); -1: no new fragment
constexpr int frag_wait(int m) {
  int w = -1;
  for (int k = 0; k < 16; ++k)
    if (kFirstUse[k] == m) {
      int c = 0;
      for (int g = read_pos(0, k) + 1; g < m; ++g) c += is_read_gap(g) ? 1 : 0;
      w = (w < 0 || c < w) ? c : w;
    }
  return w > 15 ? 15 : w;  // lgkmcnt is a 4-bit counter; a smaller count only waits for more
}

// ABL: measurement builds (bit 1: no operand DMA, 2: no barrier, 3: no fragment reads, 4: no epilogue stores, 5: no
// epilogue, 6: shader-clock stamps around the waits of gap 64 and the epilogue, 8: every operand load from one L2-resident MiB); results are garbage with bits 1-5 set.
template <int EPI, int ABL = 0, int MODE = 1>
__global__ __launch_bounds__(256) void gemm_w4_kernel(const GemmK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g4 = lane >> 4, l15 = lane & 15;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int nwg = p.tiles_m * p.tiles_n;
  const int npair = p.K / 64;  // pairs of 32-wide K steps per tile
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
  // Piece j (0-7: activations, 8-15: weights) of this wave: rows 64 wave + 8 (j & 7) + (lane >> 3) of the tile, all 128 bytes of
  // each: a wave fills two 4 KiB LDS blocks per operand, four pieces each.  The lane's LDS slot is fixed (row lane >> 3, physical
  // chunk lane & 7), so it FETCHES the logical chunk that belongs there.  The instruction's immediate (j & 3) * 1024 picks the
  // piece inside its block and moves the global address along: the offset register holds (4096 - immediate) more and the buffer
  // descriptors start 4096 bytes early, so that no offset goes negative.  Rows past M / n_pad are clamped (their products land
  // in rows / columns nobody stores).
  uint32_t voff[16];
  int ld_tile = blockIdx.x, ld_pair = 0;
  uint32_t ld_soff = 0;
  auto set_load_tile = [&](int vv) {
    int m0, n0;
    tile_origin(vv, m0, n0);
    const int r8 = lane >> 3;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = 64 * wave + 8 * j + r8, c8 = (lane & 7) ^ ((row >> 1) & 7);
      const uint32_t fix = 4096u - (uint32_t)(j & 3) * 1024u + c8 * 16;
      voff[j] = (uint32_t)min(m0 + row, p.M - 1) * (uint32_t)(p.lda * 2) + fix;
      voff[8 + j] = (uint32_t)min(n0 + row, p.n_pad - 1) * (uint32_t)(p.ldw * 2) + fix;
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
  //   pair buffer + (j < 8 ? 0 : 32 KiB) + wave * 8 KiB + (j & 7) * 1 KiB + lane * 16
  auto make_rsrc = [](const void* ptr, uint32_t bytes) {
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ptr);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)ptr >> 32));
    r[2] = bytes;
    r[3] = 0x00020000u;
    return r;
  };
  const u32x4 rs_a = make_rsrc((const char*)p.A - 4096, 0xffffffffu), rs_w = make_rsrc((const char*)p.W - 4096, 0xffffffffu);
  const u32x4 rs_bias = make_rsrc(p.bias, (uint32_t)p.n_pad * 2u);  // columns past n_pad (last column tile) read as zero
  const uint32_t dma_lds = lds_base + (uint32_t)wave * 8192u;
  // SETM0: the piece opens a 4 KiB block, or is the first one a body issues (nothing but this kernel's own asm touches M0
  // inside a body -- the build audits that -- but an epilogue may lie between two bodies)
  auto dma_piece = [&](auto j_c, uint32_t buf, auto setm0_c) {
    constexpr int J = decltype(j_c)::value;
    constexpr bool SETM0 = decltype(setm0_c)::value || (J & 3) == 0;
    constexpr int BLOCK = (J < 8 ? 0 : W_OFF) + ((J & 7) >> 2) * 4096, IMM = (J & 3) * 1024;
    // (operands copied to locals first: clang does not capture variables that appear only as asm operands of a generic lambda)
    const uint32_t base = dma_lds + buf, vo = (ABL & 256) ? ((voff[J] & 0xfffffu) | 4096u) : voff[J], so = (ABL & 256) ? (ld_soff & 0xfffu) : ld_soff;  // base, so: wave-uniform (ABL & 256: every load from one L2-resident MiB)
    const u32x4 rs = J < 8 ? rs_a : rs_w;
    if constexpr (ABL & 2) return;
    if constexpr (SETM0)
      asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen offset:%5 lds" ::"s"(base), "i"(BLOCK), "v"(vo), "s"(rs), "s"(so), "i"(IMM) : "memory", "scc");
    else
      asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(vo), "s"(rs), "s"(so), "i"(IMM) : "memory");
  };

  // ---- fragment reads: K step s (0, 1) of a pair = logical chunks 4 s + (lane >> 4); block b of an operand = rows 16 b .. ----
  const uint32_t swz8 = (l15 >> 1) & 7;
  uint32_t da[2], db[2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    da[st] = lds_base + (wm * 128 + l15) * 128 + (((4 * st + g4) ^ swz8) * 16);
    db[st] = lds_base + W_OFF + (wn * 128 + l15) * 128 + (((4 * st + g4) ^ swz8) * 16);
  }
  bf16x8 fa[2][8], fb[2][8];  // [fragment set = K step parity][16-row block]
  if constexpr (ABL & 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[0][j] = fa[1][j] = fb[0][j] = fb[1][j] = bf16x8{0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  }
  auto read_frag = [&](auto set_c, auto f_c, uint32_t a_addr, uint32_t b_addr) {
    constexpr int SET = decltype(set_c)::value, F = decltype(f_c)::value;
    if constexpr (ABL & 8) {
      if constexpr (F < 8) opaque(fb[SET][F]); else opaque(fa[SET][F - 8]);
    } else if constexpr (F < 8)
      ds_read_b128<F * 2048>(fb[SET][F], b_addr);
    else
      ds_read_b128<(F - 8) * 2048>(fa[SET][F - 8], a_addr);
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
  // the fragments of K step 0 of the pair in pair_cur, all sixteen, waited for: stream start and after every epilogue
  auto read_first_frags = [&]() {
    const uint32_t a0 = da[0] + pair_cur, b0 = db[0] + pair_cur;
    static_for<0, 16>([&](auto k) { read_frag(I0{}, std::integral_constant<int, kReadOrder[decltype(k)::value]>{}, a0, b0); });
    wait_lgkm<0>();
    MD_PIN();
  };
  static_for<0, 16>([&](auto j) { dma_piece(j, pair_cur, std::false_type{}); });
  advance_load_cursor();
  static_for<0, 16>([&](auto j) {
    if constexpr (dma_pos(MODE, decltype(j)::value) < 128) dma_piece(j, pair_wr, std::false_type{});
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_first_frags();

  // measurement build (ABL & 64): shader-clock stamps around the waits of gap 64 (s_memtime is counted by lgkmcnt: only here,
  // where the counter is drained anyway)
  uint32_t st_lgkm = 0, st_vm = 0, st_bar = 0, st_n = 0, st_vm_first = 0;
  uint64_t st_first = 0, st_last = 0, st_epi = 0;

  // One pair = 128 MFMAs = 128 gaps.  pair_cur = the buffer being multiplied, pair_wr = the other one: it receives the late
  // pieces of the NEXT pair in the first gaps, is published by the barrier in gap 64 and read from gap 65 on; from gap 64 on
  // pair_cur receives the pair after that.  ONE straight-line body but for the cursor's once-per-tile branch in gap 50;
  // past the end of the stream the fillers keep running on data nobody reads.
  auto pair_body = [&](auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;  // first pair of a tile: accumulate onto zero, request the tile's bias slice
    const uint32_t a1 = da[1] + pair_cur, b1 = db[1] + pair_cur, an = da[0] + pair_wr, bn = db[0] + pair_wr;
    const uint32_t buf_cur = pair_cur, buf_nxt = pair_wr;
    static_for<0, 128>([&](auto xc) {
      constexpr int X = decltype(xc)::value, H = X / 64, M = X % 64, I = M / 8, J = (I & 1) ? 7 - M % 8 : M % 8;
      if constexpr (X == 64) {
        uint64_t t0 = 0;
        if constexpr (ABL & 64) t0 = __builtin_readcyclecounter();
        if constexpr (!(ABL & 8)) wait_lgkm<0>();
        if constexpr (ABL & 64) { const uint64_t t = __builtin_readcyclecounter(); st_lgkm += (uint32_t)(t - t0); t0 = t; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (ABL & 64) { const uint64_t t = __builtin_readcyclecounter(); st_vm += (uint32_t)(t - t0); if constexpr (FIRST) st_vm_first += (uint32_t)(t - t0); t0 = t; }
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
      if constexpr (H == 0 && frag_wait(M) >= 0 && !(ABL & 8)) wait_lgkm<frag_wait(M)>();
      // (the first-use table assumes ascending j in row i = 0 and that block (i, *) first needs activation fragment i: both hold
      // for the serpentine order)
      mfma_acc<8 * I + J, FIRST && H == 0>(fb[H][J], fa[H][I]);
      MD_PIN();
      constexpr int RS = read_slot(X);
      if constexpr (RS >= 0) {
        using F = std::integral_constant<int, kReadOrder[RS % 16]>;
        if constexpr (RS / 16 == 1) read_frag(I1{}, F{}, a1, b1);
        if constexpr (RS / 16 == 0) read_frag(I0{}, F{}, an, bn);
      }
      static_for<0, 16>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        if constexpr (dma_pos(MODE, Q) == X) dma_piece(qc, buf_cur, std::false_type{});  // pair p + 2 -> the buffer released in gap 64
        // late pieces of pair p + 1 (the first of them is the body's first DMA instruction: it writes M0 whatever its place in a block)
        if constexpr (dma_pos(MODE, Q) - 128 == X) dma_piece(qc, buf_nxt, std::integral_constant<bool, (Q == 0 || dma_pos(MODE, Q - 1) < 128)>{});
      });
      if constexpr (FIRST && X == kBiasGap) dma_bias();  // the slot's previous contents went to registers in the last epilogue
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
    // block X = 8 i + j, register r: row m = 16 i + l15, col n = 16 j + 4 g4 + r   within the wave's quarter
    if constexpr (!(ABL & 32)) {
    // This lane's 32 bias values (column 16 j + 4 g4 + r), unpacked ONCE per tile from the wave's slot.  (npair == 1: the slot's
    // DMA was waited for by gap 64's vmcnt(0) like every other piece.)
    md_f32x2 bias_f[8][2];
    {
      u32x2 bw[8];
      static_for<0, 8>([&](auto jc) { ds_read_b64_u32<32 * decltype(jc)::value>(bw[decltype(jc)::value], bias_lds + 8 * g4); });
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results are not readable before they retire
      wait_lgkm<0>();
      MD_PIN();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        bias_f[j][0] = md_f32x2{lo_bf(bw[j][0]), hi_bf(bw[j][0])};
        bias_f[j][1] = md_f32x2{lo_bf(bw[j][1]), hi_bf(bw[j][1])};
      }
    }
    // C (and a residual R, and the KV slabs) are addressed through buffer resources: rows past M read as zero / are dropped by
    // the range check (no exec masking, no branch); an address is one 32-bit offset.
    const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)min((uint64_t)p.M * (uint64_t)p.ldc * 2, (uint64_t)0xffffffffu), 0x00020000);
    // Eight passes of 32 rows x 64 columns (row pair ip = i / 2, column half jp) through the wave's LDS tile.
    // write side, on the accumulator layout: bias add + ONE bf16 rounding (F.linear's rounding point); a lane's quad of 4 columns
    // = 8 bytes at row 16 (i & 1) + l15, chunk 2 jj + (g4 >> 1), half g4 & 1 (chunks swizzled by row & 7).  Read side: piece q
    // of this lane = row 8 q + (lane >> 3), columns 8 (lane & 7) .. + 7 of the pass: 8 lanes cover a row's 128 bytes, an
    // instruction 8 whole lines.  GELU / the residual add work on those 16-byte pieces.  Pass p + 1 is converted and written
    // while pass p's pieces are in flight back (LDS executes a wave's operations in order: read p, write p + 1, read p + 1
    // need no waits between them).
    // KIND 0: bias / GELU / residual layers and the fc1 columns of the decoder's fused layer.  KIND 1-3 (MD_EPI_QKV_ROPE): a
    // pass is one head (64 features) of the q / k / v section.  q and k: the first 32 features are rotated ON THE WRITE SIDE --
    // the reference reads them half-split (re = x[d], im = x[16 + d]) and writes them interleaved (rope.py:37-46); in the
    // accumulator layout re (block jj = 0) and im (jj = 1) of pairs d = 4 g4 .. + 3 sit in the SAME lane and their interleaved
    // outputs are the 8 consecutive features 8 g4 ..: one 16-byte chunk, no lane exchange.  fp32 arithmetic on the bf16-rounded
    // layer output with separately rounded mul, mul, sub / add, as torch evaluates it (bit-equal to rope_kv_kernel).  q stays in
    // the activation; k and v pieces go straight to the KV slab (text.py:45-46) at the row's (slot, position) offset.
    auto lds_epilogue = [&](auto kind_c) {
      constexpr int KIND = decltype(kind_c)::value;
      constexpr bool ROT = KIND == 1 || KIND == 2, SLAB = KIND == 2 || KIND == 3;
      const int lane_row = lane >> 3, lane_col = (lane & 7) * 8;
      const bool col_ok0 = wn0 + lane_col < p.n_store, col_ok1 = wn0 + 64 + lane_col < p.n_store;  // columns past n_store: last column tile only
      const uint32_t col_bytes = (uint32_t)(wn0 + lane_col) * 2u;
      const uint32_t c_base = (uint32_t)(wm0 + lane_row) * (uint32_t)(p.ldc * 2) + col_bytes;
      const uint32_t c_step = (uint32_t)p.ldc * 16u;  // 8 rows
      auto store_off = [&](int ip, int q, int jp) -> uint32_t {
        const uint32_t off = c_base + (uint32_t)(4 * ip + q) * c_step;
        return ((jp ? col_ok1 : col_ok0) ? off : 0xfffff000u) + 128 * jp;  // out of range: dropped
      };
      // residual layers: the second operand in the same whole-line pieces, prefetched one pass ahead; a broadcast residual
      // (the ViT's position embedding) wraps at res_row_mod (>= 256: at most one wrap per tile)
      const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc((void*)p.R, 0, (int)min((uint64_t)(p.res_row_mod ? p.res_row_mod : p.M) * (uint64_t)p.ldr * 2, (uint64_t)0xffffffffu), 0x00020000);
      const uint32_t wrap = p.res_row_mod ? (uint32_t)p.res_row_mod : 0x7fffffffu;
      const uint32_t r_row0 = (uint32_t)(wm0 + lane_row) % wrap;
      auto load_residual = [&](int pass, u32x4 (&rv)[4]) {
        const int ip = pass >> 1, jp = pass & 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t row = r_row0 + 8u * (uint32_t)(4 * ip + q);
          row -= (row >= wrap) ? wrap : 0u;
          const uint32_t off = (jp ? col_ok1 : col_ok0) ? row * (uint32_t)(p.ldr * 2) + col_bytes : 0xfffff000u;
          rv[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, off + 128 * jp, 0, 0));
        }
      };
      // q / k / v sections: the wave's 128 columns are two heads of ONE section (sections are n_heads x 64 wide, a multiple of 128)
      const int sec = (KIND == 0) ? 0 : wn0 / p.rope_d;
      const int head0 = (KIND == 0) ? 0 : (wn0 - sec * p.rope_d) >> 6;
      const __amdgpu_buffer_rsrc_t rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void*)(KIND == 2 ? p.kslab : p.vslab), 0, (int)p.slab_bytes, 0x00020000);
      // write side, rows 32 ip + 16 ii + l15: (cos, sin) of pairs d = 4 g4 .. + 3; read side, rows 32 ip + 8 q + (lane >> 3): the
      // row's byte offset in a layer's slab (rows past M: out of range, dropped)
      // (all of a tile's row info is requested here, ahead of the first pass: 16 + 4 registers per row pair)
      f32x4 cs[4][2][2];
      uint32_t kv_off[4][4];
      if constexpr (ROT) {
#pragma unroll
        for (int ip = 0; ip < 4; ++ip)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const f32x4* cp = (const f32x4*)(p.rope_cs + (int64_t)min(wm0 + 32 * ip + 16 * ii + l15, p.M - 1) * 32);
            cs[ip][ii][0] = cp[2 * g4];
            cs[ip][ii][1] = cp[2 * g4 + 1];
          }
      }
      if constexpr (SLAB) {
#pragma unroll
        for (int ip = 0; ip < 4; ++ip)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int m = wm0 + 32 * ip + 8 * q + lane_row;
            kv_off[ip][q] = m < p.M ? p.rope_kv[m] + (uint32_t)(lane & 7) * 16u : 0xfffff000u;
          }
      }
      auto convert_write = [&](auto pc) {
        constexpr int PASS = decltype(pc)::value, ip = PASS >> 1, jp = PASS & 1;
        static_for<0, 2>([&](auto iic) {
          constexpr int ii = decltype(iic)::value, i = 2 * ip + ii;
          const uint32_t row = 16 * ii + l15;
          const uint32_t row_lds = tile_lds + row * 128;
          u32x2 w[4];
          static_for<0, 4>([&](auto jjc) {
            constexpr int jj = decltype(jjc)::value, j = 4 * jp + jj, base = 4 * (8 * i + j);
            const md_f32x2 x0 = md_f32x2{acc_read<base + 0>(), acc_read<base + 1>()} + bias_f[j][0];
            const md_f32x2 x1 = md_f32x2{acc_read<base + 2>(), acc_read<base + 3>()} + bias_f[j][1];
            w[jj] = u32x2{pack_bf16x2(x0[0], x0[1]), pack_bf16x2(x1[0], x1[1])};  // the layer's bf16 output
          });
          static_for<0, 4>([&](auto jjc) {
            constexpr int jj = decltype(jjc)::value;
            if constexpr (ROT && jj == 0) {
              u32x4 v;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float re = (r & 1) ? hi_bf(w[0][r >> 1]) : lo_bf(w[0][r >> 1]), im = (r & 1) ? hi_bf(w[1][r >> 1]) : lo_bf(w[1][r >> 1]);
                float o_re, o_im;
                md_rope_pair(re, im, cs[ip][ii][r >> 1][2 * (r & 1)], cs[ip][ii][r >> 1][2 * (r & 1) + 1], o_re, o_im);
                v[r] = pack_bf16x2(o_re, o_im);
              }
              ds_write_b128_asm(row_lds + ((g4 ^ (row & 7)) * 16), v);  // features 8 g4 .. + 7 of the head
            } else if constexpr (!(ROT && jj == 1)) {
              ds_write_b64_asm(row_lds + (((2 * jj + (g4 >> 1)) ^ (row & 7)) * 16) + (g4 & 1) * 8, w[jj]);
            }
          });
        });
      };
      auto read_pieces = [&](u32x4 (&tv)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int idx = q * 64 + lane, row = idx >> 3, ch = idx & 7;
          ds_read_b128_u32(tv[q], tile_lds + row * 128 + ((ch ^ (row & 7)) * 16));
        }
      };
      constexpr int WRITES = ROT ? 6 : 8;  // LDS writes of a pass
      u32x4 tv[2][4], rres[2][4];
      if constexpr (EPI == MD_EPI_RESIDUAL) load_residual(0, rres[0]);
      convert_write(I0{});
      read_pieces(tv[0]);
      MD_PIN();
      static_for<0, 8>([&](auto pc) {
        constexpr int PASS = decltype(pc)::value, ip = PASS >> 1, jp = PASS & 1;
        if constexpr (PASS + 1 < 8) {
          if constexpr (EPI == MD_EPI_RESIDUAL) load_residual(PASS + 1, rres[(PASS + 1) & 1]);
          convert_write(std::integral_constant<int, PASS + 1>{});
          wait_lgkm<WRITES>();  // this pass's four reads are back; the next pass's writes may still be on their way
        } else {
          wait_lgkm<0>();
        }
        MD_PIN();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u32x4 v = tv[PASS & 1][q];
          if constexpr (KIND == 0 && (EPI == MD_EPI_GELU || EPI == MD_EPI_QKV_ROPE)) {
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
          if constexpr (SLAB) {
            // slab: [head][position][64] bf16 per slot
            const uint32_t head_off = (uint32_t)(head0 + jp) * (uint32_t)p.rope_ctx * 128u;
            if constexpr (!(ABL & 16)) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_s, kv_off[ip][q], head_off, 0);
          } else {
            const uint32_t off = store_off(ip, q, jp);
            if constexpr (!(ABL & 16)) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_c, off, 0, 0);
            else keep_alive(v);
          }
        }
        MD_PIN();
        if constexpr (PASS + 1 < 8) read_pieces(tv[(PASS + 1) & 1]);
        MD_PIN();
      });
    };
    if constexpr (EPI == MD_EPI_QKV_ROPE) {
      const int sec = wn0 / p.rope_d;  // 0 q, 1 k, 2 v, >= 3: the fc1 columns   (wave-uniform)
      if (sec == 0) lds_epilogue(std::integral_constant<int, 1>{});
      else if (sec == 1) lds_epilogue(std::integral_constant<int, 2>{});
      else if (sec == 2) lds_epilogue(std::integral_constant<int, 3>{});
      else lds_epilogue(I0{});
    } else {
      lds_epilogue(I0{});
    }
    }
    // the next tile's first fragments again (the copy read before the epilogue was not kept: 64
    // registers the epilogue does not have to carry); its first pair was published by the last barrier above
    read_first_frags();
    if constexpr (ABL & 64) { st_epi += __builtin_readcyclecounter() - te0; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA of the run-on stream in flight at the end of the wave
  if constexpr (ABL & 64) {
    if (lane == 0 && p.slabs != nullptr) {
      float* o = p.slabs + (blockIdx.x * 4 + wave) * 8;
      o[0] = (float)st_n; o[1] = (float)(st_last - st_first); o[2] = (float)st_lgkm; o[3] = (float)st_vm; o[4] = (float)st_bar; o[5] = (float)st_epi; o[6] = (float)st_vm_first;
    }
  }
}

int g_w4_grid = 0;     // md_gemm_set_tuning "w4_grid": persistent workgroups per launch (0 = one per CU); a multiple of 8
// md_gemm_set_tuning "w4_variant" (MD_W4_VARIANT): low 4 bits = placement of the LDS-DMA pieces (MODE, 0 = the default;
// the others exist for the bias epilogue only), the rest 16 * ABL (measurement builds, bias epilogue only)
int g_w4_variant = [] { const char* e = getenv("MD_W4_VARIANT"); return (e && *e) ? atoi(e) : 0; }();
uint64_t g_w4_debug = 0;  // measurement builds: device buffer for the in-kernel stamps (md_gemm_set_tuning "w4_dbg_lo" / "w4_dbg_hi")
constexpr int kDefaultMode = 2;

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
    switch (256 * (mode == 0 ? kDefaultMode : mode) + abl) {
      case 256 * 2 + 2: return launch<MD_EPI_BIAS, 2, 2>(k, stream);
      case 256 * 2 + 4: return launch<MD_EPI_BIAS, 4, 2>(k, stream);
      case 256 * 2 + 8: return launch<MD_EPI_BIAS, 8, 2>(k, stream);
      case 256 * 2 + 14: return launch<MD_EPI_BIAS, 14, 2>(k, stream);
      case 256 * 2 + 16: return launch<MD_EPI_BIAS, 16, 2>(k, stream);
      case 256 * 2 + 32: return launch<MD_EPI_BIAS, 32, 2>(k, stream);
      case 256 * 2 + 46: return launch<MD_EPI_BIAS, 46, 2>(k, stream);
      case 256 * 2 + 64: return launch<MD_EPI_BIAS, 64, 2>(k, stream);
      case 256 * 2 + 80: return launch<MD_EPI_BIAS, 80, 2>(k, stream);
      case 256 * 2 + 256: return launch<MD_EPI_BIAS, 256, 2>(k, stream);
      case 256 * 2 + 272: return launch<MD_EPI_BIAS, 272, 2>(k, stream);
      case 256 * 2 + 320: return launch<MD_EPI_BIAS, 320, 2>(k, stream);
      default: return MD_ERR_INVALID_ARG;
    }
  }
#endif
  if (abl != 0) return MD_ERR_INVALID_ARG;
  if (mode != 0 && mode != kDefaultMode) {
    if (epi != MD_EPI_BIAS) return MD_ERR_INVALID_ARG;
    switch (mode) {
      case 1: return launch<MD_EPI_BIAS, 0, 1>(k, stream);
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
