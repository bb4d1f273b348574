// bf16 GEMM, big-tile kernel:  C = epilogue(A . W^T + b)  for M > 64 and K % 64 == 0.
//
// One workgroup = FOUR waves, one per SIMD, each owning a 128 x 128 quarter of a 256 x 256 tile
// of C with its 16 accumulator blocks (256 registers) in the accumulator half of the register
// file, named literally by inline asm.  Per 16-wide K step a wave issues 16
// v_mfma_f32_32x32x16_bf16 (512 matrix-pipe cycles) for 8 ds_read_b128 -- two thirds of the LDS
// reads per FLOP of an eight-wave 128 x 64 split, and no second wave on the SIMD to arbitrate with.
// Because nothing else runs on the SIMD, everything that is not an MFMA is a FILLER placed by hand
// between two MFMAs (tables below), and since the four waves run in lockstep between barriers the
// fillers of one kind are spread out evenly so that neither the LDS nor the vector-memory front
// end sees a burst:
//
//   stream of K in PAIRS of 32-wide slices, software pipelined over three levels
//     HBM/L2 -> registers   raw buffer loads of 8 rows x 128 bytes (whole cache lines; with 64-byte row
//                           pieces every line crossed L2 -> L1 twice), issued one pair (~2000 cycles) ahead
//     registers -> LDS      ds_write_b128 into a 4-stage ring (pair being multiplied + pair being written),
//                           chunk-swizzled on the WRITE side; each register is re-requested right after it
//                           is written out, so ONE pair's worth of registers carries the whole stream
//     LDS -> fragments      ds_read_b128 one 16-wide K step ahead of the MFMAs that consume them, waited
//                           for with COUNTED lgkmcnt (only the fragment an MFMA is first to use)
//   one s_barrier per pair (64 MFMAs).
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
//
// Measured (profiles/r02_gemm_w4_*): 1.37 PF/s at 8192^3, 1.09-1.26 PF/s on the decoder-prefill / projector
// shapes, 0.84-0.97 on the K = 1152 ViT shapes (random operands; the chip runs ~1.75 GHz under it: MFMA-busy
// 70 %, 18 % of wave cycles parked at waits / the barrier).
#include "gemm_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BM = 256, BN = 256, BKT = 32;
constexpr int ROW_BYTES = BKT * 2;             // 64-byte rows: 4 chunks of 16 bytes
constexpr int A_BYTES = BM * ROW_BYTES;        // 16 KiB
constexpr int STAGE = (BM + BN) * ROW_BYTES;   // 32 KiB
constexpr int RING = 4 * STAGE;                // 128 KiB: the pair of slices being multiplied + the pair being written
// next to the ring: the layer's whole bias vector (28 KiB), or -- residual layers, whose epilogue transposes through
// LDS -- four 4 KiB transposition tiles and a bias vector of up to 6144 columns
constexpr int XPOSE_BYTES = 32 * 128;          // 32 rows x 64 bf16
constexpr int LDS_BYTES = RING + 14336 * 2;
template <int EPI> constexpr int bias_max_cols() { return EPI == MD_EPI_RESIDUAL ? (LDS_BYTES - RING - 4 * XPOSE_BYTES) / 2 : (LDS_BYTES - RING) / 2; }

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

// ---- the filler schedule of one PAIR of slices (64 MFMAs, 64 gaps), as compile-time tables --------
// Fillers are issued right AFTER the MFMA of their gap, in the order read, write, load:
//   ds_read      every odd gap: 8 fragment reads per 16-gap half, for the half that follows
//   ds_write     gaps 0, 3, 6, .., 45: the 16 pieces of the NEXT pair of slices (published by the barrier in gap 48)
//   buffer_load  one gap after each write: the same registers are re-requested for the pair after that
// The k-th read of a half fetches fragment kReadOrder[k] (0-3 = weight blocks j, 4-7 = activation blocks i) in
// the order the MFMAs first need them: MFMA m = 4 i + j uses B_j and A_i, so B0 A0 B1 B2 B3 A1 A2 A3 are first
// used by MFMAs 0 0 1 2 3 4 8 12 of the consuming half.
constexpr int kReadOrder[8] = {0, 4, 1, 2, 3, 5, 6, 7};
constexpr int kFirstUse[8] = {0, 0, 1, 2, 3, 4, 8, 12};  // by read position k
constexpr bool gap_has_write(int g) { return ((g % 64) + 64) % 64 % 3 == 0 && ((g % 64) + 64) % 64 < 48; }
// LDS operations issued strictly after the read of gap g_issue and before MFMA x_need (gaps along the periodic stream)
constexpr int lds_ops_between(int g_issue, int x_need) {
  int n = gap_has_write(g_issue) ? 1 : 0;  // the write of the read's own gap is issued after the read
  for (int g = g_issue + 1; g < x_need; ++g) n += (((g % 2) + 2) % 2 == 1) + (gap_has_write(g) ? 1 : 0);
  return n;
}
// lgkmcnt to wait for before MFMA m of half h (0..3) of a pair; -1: the MFMA introduces no new fragment.  The
// fragments of half h are read in gaps 16 (h - 1) + 1 + 2 k  (half 0: at the end of the previous pair).
constexpr int frag_wait(int h, int m) {
  int w = -1;
  for (int k = 0; k < 8; ++k)
    if (kFirstUse[k] == m) {
      const int c = lds_ops_between(16 * (h - 1) + 1 + 2 * k, 16 * h + m);
      w = (w < 0 || c < w) ? c : w;
    }
  return w;
}
static_assert(frag_wait(0, 0) <= 15 && frag_wait(1, 12) <= 15 && frag_wait(2, 12) <= 15 && frag_wait(3, 0) <= 15, "lgkmcnt is a 4-bit counter");

// ---- MODE >= 1: operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), no register round trip ----------
// LDS image: TWO pair buffers of 64 KiB; a buffer holds 64 K elements of the tile's 256 activation rows (32 KiB) and 256
// weight rows (32 KiB) as 128-byte rows, logical 16-byte chunk c of row r at physical chunk c ^ ((r >> 1) & 7).  A DMA
// instruction moves 8 rows x 128 bytes (whole cache lines; the lane -> LDS mapping is linear, so the permutation sits in
// the per-lane SOURCE offset); a wave issues 16 of them per pair.  Fragment reads: K step s (0-3) of a pair = logical chunks
// 2 s (lanes 0-31) and 2 s + 1 (lanes 32-63).
//   filler schedule of one pair (gap g = right after MFMA g):
//     ds_read      half 1's fragments in gaps 1, 3, .., 15, half 2's in 17, .., 31, half 3's in 32, 34, .., 46 (EVEN: the last
//                  read of the buffer being multiplied is two MFMAs old at the barrier), the NEXT pair's half 0 in 49, .., 63
//     gap 48       s_waitcnt lgkmcnt(0) (this wave has finished reading the current buffer), vmcnt(0) (its DMA pieces of the
//                  next pair have landed), s_barrier: the next buffer is published, the current one is released
//     LDS-DMA      16 pieces of the pair AFTER the next one into the released buffer, at the gaps dma_pos() names: the
//                  stream runs two pairs ahead of the MFMAs at issue, a piece has 32+ gaps (1000+ cycles) to land
constexpr int PBUF = 65536;                     // one pair buffer
constexpr int kAdvanceGap = 40;                 // the load cursor moves on here: after the last wrapped piece, before gap 48
constexpr int dma_pos(int mode, int q) {        // stream gap (48 .. 111) at which piece q of pair p + 2 is issued, in body(p) / body(p + 1)
  return mode == 1 ? (q < 8 ? 48 + 2 * q : 64 + 2 * (q - 8)) : mode == 2 ? 48 + q : 48 + 3 * q;
}
constexpr bool dma_is_read_gap(int g) {
  const int x = ((g % 64) + 64) % 64;
  return (x < 32 && x % 2 == 1) || (x >= 32 && x < 48 && x % 2 == 0) || (x > 48 && x % 2 == 1);
}
// stream position (relative to gap 0 of the consuming pair) of the k-th read of half h
constexpr int dma_read_pos(int h, int k) { return h == 0 ? -15 + 2 * k : h == 1 ? 1 + 2 * k : h == 2 ? 17 + 2 * k : 32 + 2 * k; }
// which (half, k) is read in gap x of a body, encoded 8 h + k; -1: none
constexpr int dma_read_slot(int x) {
  for (int h = 0; h < 4; ++h)
    for (int k = 0; k < 8; ++k)
      if (((dma_read_pos(h, k) % 64) + 64) % 64 == x) return 8 * h + k;
  return -1;
}
constexpr int dma_frag_wait(int h, int m) {
  int w = -1;
  for (int k = 0; k < 8; ++k)
    if (kFirstUse[k] == m) {
      int c = 0;
      for (int g = dma_read_pos(h, k) + 1; g < 16 * h + m; ++g) c += dma_is_read_gap(g) ? 1 : 0;
      w = (w < 0 || c < w) ? c : w;
    }
  return w;
}
static_assert(dma_frag_wait(0, 0) <= 15 && dma_frag_wait(1, 12) <= 15 && dma_frag_wait(2, 12) <= 15 && dma_frag_wait(3, 12) <= 15, "lgkmcnt is a 4-bit counter");

// ABL: timing ablations for profiling (bit 0: no ds_write, 1: no global loads, 2: no barrier, 3: no
// fragment reads, 4: no epilogue stores); results are garbage with any bit set.
template <int EPI, int ABL = 0, int MODE = 1>
__global__ __launch_bounds__(256) void gemm_w4_kernel(const GemmK p) {
  constexpr bool DMA = MODE != 0;
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

  // ---- load cursor: one pair of slices (64 K elements) at a time, across tile boundaries --------
  // A load instruction covers 8 rows x 128 bytes: WHOLE cache lines (with 64-byte row pieces every line
  // would be fetched twice, once per slice, ~1000 cycles apart, and the 32 KiB L1 does not keep it).  Piece
  // j (0-7: activations, 8-15: weights) of this thread: row 32 (j & 7) + (tid >> 3), 16-byte chunk c8 = tid & 7
  // of the row's 128 bytes: chunks 0-3 belong to the even slice of the pair, 4-7 to the odd one.
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0xffffffffu, 0x00020000);
  uint32_t voff[16];
  int ld_tile = blockIdx.x, ld_pair = 0;
  uint32_t ld_soff = 0;
  auto set_load_tile = [&](int vv) {
    int m0, n0;
    tile_origin(vv, m0, n0);
    // r8: this thread's row within a 32-row block (wave w covers rows 8 w .. 8 w + 7 of every block).  DMA modes: the lane's LDS
    // slot is fixed (row r8, physical chunk tid & 7), so it FETCHES the logical chunk that belongs there
    const int r8 = tid >> 3, c8 = DMA ? ((tid & 7) ^ ((r8 >> 1) & 7)) : (tid & 7);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      voff[j] = (uint32_t)min(m0 + 32 * j + r8, p.M - 1) * (uint32_t)(p.lda * 2) + c8 * 16;
      voff[8 + j] = (uint32_t)min(n0 + 32 * j + r8, p.n_pad - 1) * (uint32_t)(p.ldw * 2) + c8 * 16;
    }
  };
  set_load_tile(ld_tile);
  // DMA modes: the buffer descriptors as four scalar words each (inline asm operand), and the wave's LDS-DMA base:
  // piece j lands at  pair buffer + (j < 8 ? 0 : 32 KiB) + (j & 7) * 4 KiB + wave * 1 KiB + lane * 16
  u32x4 rs_a, rs_w;
  rs_a[0] = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)p.A);
  rs_a[1] = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)p.A >> 32));
  rs_w[0] = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)p.W);
  rs_w[1] = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)p.W >> 32));
  rs_a[2] = rs_w[2] = 0xffffffffu;
  rs_a[3] = rs_w[3] = 0x00020000u;
  const uint32_t dma_lds = lds_base + (uint32_t)wave * 1024u;
  auto dma_piece = [&](auto j_c, uint32_t buf) {
    constexpr int J = decltype(j_c)::value;
    constexpr int OFF = J < 8 ? J * 4096 : 32768 + (J - 8) * 4096;
    // (operands copied to locals first: clang does not capture variables that appear only as asm operands of a generic lambda)
    const uint32_t base = dma_lds + buf, vo = voff[J], so = ld_soff;  // base, so: wave-uniform
    const u32x4 rs = J < 8 ? rs_a : rs_w;
    if constexpr (!(ABL & 2))
      asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds" ::"s"(base), "i"(OFF), "v"(vo), "s"(rs), "s"(so) : "memory", "scc");
  };

  u32x4 R[16];  // one pair of slices in registers; each piece is re-requested right after it is written out
  if constexpr (ABL != 0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) R[j] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  }
  auto load_piece = [&](auto j_c) {
    constexpr int J = decltype(j_c)::value;
    if constexpr (ABL & 2)
      opaque(R[J]);
    else if constexpr (J < 8)
      R[J] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff[J], ld_soff, 0));
    else
      R[J] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, voff[J], ld_soff, 0));
  };
  auto advance_load_cursor = [&]() {
    ld_soff += 128;
    if (++ld_pair == npair) {
      ld_pair = 0;
      ld_soff = 0;
      // past the end of the stream the cursor stays on the last tile: the loads keep going
      // (their data is never written anywhere that is read), which keeps the loop free of branches
      if (ld_tile + grid_x < nwg) {
        ld_tile += grid_x;
        set_load_tile(ld_tile);
      }
    }
  };

  // ---- the LDS ring: four 32 KiB stages = two pairs, roles swapping after every pair (byte offsets,
  // wave-uniform).  A stage holds one 32-wide slice: A rows then W rows, 64 bytes each, logical chunk c
  // of row r at physical chunk c ^ ((r >> 2) & 3) (a ds_read_b128 of one chunk column over a 16-lane
  // group's rows then touches 16 distinct 16-byte bank slots).  ODD slices additionally store row r at
  // row r ^ 1: the two halves of a ds_write_b128's 8-lane group (even slice | odd slice of one row)
  // would otherwise hit the same banks.
  uint32_t pair_cur = 0, pair_wr = 2 * STAGE;
  const uint32_t r8w = tid >> 3, c8w = tid & 7;
  const uint32_t wr_even = lds_base + r8w * ROW_BYTES + (((c8w & 3) ^ ((r8w >> 2) & 3)) * 16);
  const uint32_t wr_odd = lds_base + STAGE + (r8w ^ 1) * ROW_BYTES + (((c8w & 3) ^ ((r8w >> 2) & 3)) * 16);
  const uint32_t wr_lane = (c8w < 4) ? wr_even : wr_odd;
  auto write_piece = [&](auto j_c, uint32_t pair) {
    constexpr int J = decltype(j_c)::value;
    if constexpr (ABL & 1)
      keep_alive(R[J]);
    else
      ds_write_b128<(J < 8 ? J * 32 * ROW_BYTES : A_BYTES + (J - 8) * 32 * ROW_BYTES)>(wr_lane + pair, R[J]);
  };

  // ---- fragment reads: K step s of a slice = logical chunks 2s (lanes 0-31) and 2s+1 (lanes 32-63)
  const uint32_t swz = (l31 >> 2) & 3;
  const uint32_t coff[2] = {((0 + hi) ^ swz) * 16, ((2 + hi) ^ swz) * 16};
  // [slice parity within the pair][K step]
  uint32_t ra[2][2], rb[2][2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      ra[par][st] = lds_base + par * STAGE + ((wm * 128 + l31) ^ par) * ROW_BYTES + coff[st];
      rb[par][st] = lds_base + par * STAGE + A_BYTES + ((wn * 128 + l31) ^ par) * ROW_BYTES + coff[st];
    }
  // DMA modes: 128-byte rows, [K step 0-3]
  const uint32_t swz8 = (l31 >> 1) & 7;
  uint32_t da[4], db[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    da[st] = lds_base + (wm * 128 + l31) * 128 + (((2 * st + hi) ^ swz8) * 16);
    db[st] = lds_base + 32768 + (wn * 128 + l31) * 128 + (((2 * st + hi) ^ swz8) * 16);
  }
  bf16x8 fa[2][4], fb[2][4];  // [fragment set][32-row block]
  if constexpr (ABL != 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) fa[0][j] = fa[1][j] = fb[0][j] = fb[1][j] = bf16x8{0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  }
  auto read_frag = [&](auto set_c, auto q_c, uint32_t a_addr, uint32_t b_addr) {
    constexpr int SET = decltype(set_c)::value, Q = decltype(q_c)::value;
    if constexpr (ABL & 8) {
      if constexpr (Q < 4) opaque(fb[SET][Q]); else opaque(fa[SET][Q - 4]);
    } else if constexpr (Q < 4)
      ds_read_b128<Q * 32 * (DMA ? 128 : ROW_BYTES)>(fb[SET][Q], b_addr);
    else
      ds_read_b128<(Q - 4) * 32 * (DMA ? 128 : ROW_BYTES)>(fa[SET][Q - 4], a_addr);
  };

  acc_reserve();

  // ---- stream prologue: pair 0 written, pair 1 requested ---------------------------------------
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // the first fragments (K step 0) of the pair in pair_cur, all eight, waited for: stream start and after every epilogue
  auto read_first_frags = [&]() {
    const uint32_t a0 = (DMA ? da[0] : ra[0][0]) + pair_cur, b0 = (DMA ? db[0] : rb[0][0]) + pair_cur;
    static_for<0, 8>([&](auto q) { read_frag(I0{}, std::integral_constant<int, kReadOrder[decltype(q)::value]>{}, a0, b0); });
    wait_lgkm<0>();
    MD_PIN();
  };
  if constexpr (!DMA) {
    static_for<0, 16>([&](auto j) { load_piece(j); });
    advance_load_cursor();
    asm volatile("" ::: "memory");
    static_for<0, 16>([&](auto j) { write_piece(j, pair_cur); });
    asm volatile("" ::: "memory");
    static_for<0, 16>([&](auto j) { load_piece(j); });
    advance_load_cursor();
    wait_lgkm<0>();
  } else {
    // pair 0 whole, and the pieces of pair 1 that the steady state issues at the END of a body (dma_pos < 64)
    static_for<0, 16>([&](auto j) { dma_piece(j, pair_cur); });
    advance_load_cursor();
    static_for<0, 16>([&](auto j) {
      if constexpr (dma_pos(MODE, decltype(j)::value) < 64) dma_piece(j, pair_wr);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  read_first_frags();

  // One pair of slices = 64 MFMAs = 64 gaps (four halves of 16: even slice K steps 0, 1, odd slice K steps 0, 1;
  // fragment sets 0, 1, 0, 1).  The four waves run in lockstep between barriers, so whatever one wave does in a
  // gap all four do: fillers of one kind are spread out so that neither the LDS nor the vector-memory front end
  // sees a burst.  The pair being written was requested one pair (64 gaps, ~2000 cycles) earlier.  The barrier in
  // gap 48 publishes it (every wave drains its own writes first); its stages were last read in gap 47 of the
  // PREVIOUS pair, before that pair's barrier.  ONE straight-line body, no branch inside it (a second code path
  // would be a join over ~200 live registers); past the end of the stream the fillers keep running on data nobody reads.
  auto pair_body_reg = [&](auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;  // first K step of a tile: accumulate onto zero
    const uint32_t a_e1 = ra[0][1] + pair_cur, b_e1 = rb[0][1] + pair_cur;   // even slice, K step 1
    const uint32_t a_o0 = ra[1][0] + pair_cur, b_o0 = rb[1][0] + pair_cur;   // odd slice, K step 0
    const uint32_t a_o1 = ra[1][1] + pair_cur, b_o1 = rb[1][1] + pair_cur;   // odd slice, K step 1
    const uint32_t a_n0 = ra[0][0] + pair_wr, b_n0 = rb[0][0] + pair_wr;     // next pair's even slice, K step 0
    const uint32_t wr = pair_wr;
    static_for<0, 64>([&](auto xc) {
      constexpr int X = decltype(xc)::value, H = X / 16, M = X % 16, I = M / 4, J = M % 4, SET = H & 1;
      if constexpr (X == 48 && !(ABL & 4)) {
        // this wave's writes (the last one in gap 45; one younger read in gap 47) are complete
        wait_lgkm<1>();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      // counted wait: the fragments this MFMA is the first to use have landed, younger LDS operations stay in flight
      if constexpr (frag_wait(H, M) >= 0 && !(ABL & 8)) wait_lgkm<frag_wait(H, M)>();
      mfma_acc<M, FIRST && H == 0>(fb[SET][J], fa[SET][I]);
      MD_PIN();
      if constexpr (X % 2 == 1) {
        using Q = std::integral_constant<int, kReadOrder[(X % 16) / 2]>;
        if constexpr (H == 0) read_frag(I1{}, Q{}, a_e1, b_e1);
        if constexpr (H == 1) read_frag(I0{}, Q{}, a_o0, b_o0);
        if constexpr (H == 2) read_frag(I1{}, Q{}, a_o1, b_o1);
        if constexpr (H == 3) read_frag(I0{}, Q{}, a_n0, b_n0);
      }
      if constexpr (X % 3 == 0 && X < 48) write_piece(std::integral_constant<int, X / 3>{}, wr);
      if constexpr (X % 3 == 1 && X < 48) load_piece(std::integral_constant<int, X / 3>{});
      MD_PIN();
    });
    advance_load_cursor();
    const uint32_t t = pair_cur;  // swap the roles of the two pairs of stages
    pair_cur = pair_wr;
    pair_wr = t;
  };
  // The same pair with the operands arriving by LDS-DMA (schedule: the tables above).  pair_cur = the buffer being
  // multiplied, pair_wr = the other one: it receives the late pieces of the NEXT pair in the first gaps, is published by
  // the barrier in gap 48, and is read from gap 49 on; from gap 48 on pair_cur receives the pair after that.
  // measurement build (ABL & 64): shader-clock stamps around the waits of gap 48 (s_memtime is counted by lgkmcnt: only here,
  // where the counter is drained anyway)
  uint32_t st_lgkm = 0, st_vm = 0, st_bar = 0, st_n = 0;
  uint64_t st_first = 0, st_last = 0, st_epi = 0;
  auto pair_body_dma = [&](auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;
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
      if constexpr (dma_frag_wait(H, M) >= 0 && !(ABL & 8)) wait_lgkm<dma_frag_wait(H, M)>();
      mfma_acc<M, FIRST && H == 0>(fb[SET][J], fa[SET][I]);
      MD_PIN();
      constexpr int RS = dma_read_slot(X);
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
      if constexpr (X == kAdvanceGap) advance_load_cursor();
      MD_PIN();
    });
    const uint32_t t = pair_cur;
    pair_cur = pair_wr;
    pair_wr = t;
  };
  auto pair_body = [&](auto first_c) {
    if constexpr (DMA) pair_body_dma(first_c); else pair_body_reg(first_c);
  };

  // ---- tile loop ------------------------------------------------------------------------------
  // The layer's bias vector goes to LDS once per launch: a global load inside the epilogue would have to wait for
  // every older operand load of the running stream (loads return in order), i.e. drain the prefetch once per tile.
  const uint32_t tile_lds = lds_base + RING + wave * XPOSE_BYTES;  // residual epilogue only
  const uint32_t bias_lds = lds_base + RING + (EPI == MD_EPI_RESIDUAL ? 4 * XPOSE_BYTES : 0);
  const bool bias_in_lds = p.n_pad <= bias_max_cols<EPI>();
  if (bias_in_lds) {
    for (int c = tid * 4; c < p.n_pad; c += 256 * 4) {
      u32x2 bw = {0u, 0u};
      if (p.bias != nullptr) bw = *(const u32x2*)(p.bias + c);
      ds_write_b64_asm(bias_lds + c * 2, bw);
    }
    wait_lgkm<0>();
    __syncthreads();
  }
  for (int vtile = blockIdx.x; vtile < nwg; vtile += grid_x) {
    // the accumulators are (re)defined by the first K step of every tile: nothing is carried from
    // one tile to the next in them
    pair_body(std::true_type{});
    for (int u = 1; u < npair; ++u) pair_body(std::false_type{});
    wait_lgkm<0>();
    MD_PIN();
    uint64_t te0 = 0;
    if constexpr (ABL & 64) { te0 = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

    // ---- epilogue of tile vtile (the next tile's first slices are already in the ring / in flight)
    // block X = 4 i + j, register r: row m = 32 i + l31, col n = 32 j + 8 (r >> 2) + 4 hi + (r & 3)   within the wave's quarter
    int m0c, n0c;
    tile_origin(vtile, m0c, n0c);
    const int wm0 = m0c + wm * 128, wn0 = n0c + wn * 128;
    if constexpr (ABL & 32) {
      // measurement build: no epilogue at all (what a tile costs without one)
    } else if constexpr (EPI == MD_EPI_RESIDUAL) {
    // Residual layers keep the LDS transposition: their second operand is read in whole 128-byte row pieces (the
    // register-only path below reads / writes 32-byte pieces, which costs these short-K, narrow-N layers 5 % --
    // profiles/r02_gemm_w4_epilogue_variants.txt).
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results are not readable before they retire
    // eight passes of 32 rows x 64 columns (row block i, column half jp): bias add + ONE bf16 rounding on the
    // accumulator layout, transposition through the wave's 4 KiB LDS tile, then GELU / residual and the global
    // stores on whole 16-byte row pieces.  Piece q of this lane: row (q*64 + lane) >> 3, chunk (q*64 + lane) & 7.
    // Piece q of this lane in pass (i, jp): row 32 i + 8 q + (lane >> 3), columns 64 jp + 8 (lane & 7) .. + 7.  Residual
    // loads and result stores go through buffer resources over R and C (num_records = M rows): rows past M read as zero /
    // are dropped by the range check, the address is one 32-bit offset per (i, q) -- a scalar multiple of the 8-row step
    // added to the lane's base -- and the column half is an immediate (the per-piece m / n tests, 64-bit address arithmetic
    // and exec-mask branches were ~600 of this epilogue's VALU instructions per tile).
    const int lane_row = lane >> 3, lane_col = (lane & 7) * 8;
    const __amdgpu_buffer_rsrc_t rsrc_cr = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)min((uint64_t)p.M * (uint64_t)p.ldc * 2, (uint64_t)0xffffffffu), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc((void*)p.R, 0, (int)min((uint64_t)(p.res_row_mod ? p.res_row_mod : p.M) * (uint64_t)p.ldr * 2, (uint64_t)0xffffffffu), 0x00020000);
    const bool col_ok0 = wn0 + lane_col < p.n_store, col_ok1 = wn0 + 64 + lane_col < p.n_store;
    const uint32_t col_bytes = (uint32_t)(wn0 + lane_col) * 2u;
    const uint32_t c_base = (uint32_t)(wm0 + lane_row) * (uint32_t)(p.ldc * 2) + col_bytes;
    const uint32_t c_step = (uint32_t)p.ldc * 16u;  // 8 rows
    // residual row of this lane's first piece; a broadcast residual wraps at res_row_mod (>= 256: at most one wrap per tile)
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
    auto store_off = [&](int i, int q, int jp) -> uint32_t {
      const uint32_t off = c_base + (uint32_t)(4 * i + q) * c_step;
      return ((jp ? col_ok1 : col_ok0) ? off : 0xfffff000u) + 128 * jp;  // columns past n_store (last column tile): out of range
    };
    u32x4 rres[2][4];
    if constexpr (EPI == MD_EPI_RESIDUAL) load_residual(0, rres[0]);
    // this lane's 16 bias quads, once per tile (not per pass: one LDS round trip less on every pass)
    u32x2 bias_r[4][4];
    static_for<0, 16>([&](auto jq) {
      constexpr int j = decltype(jq)::value / 4, q = decltype(jq)::value % 4;
      ds_read_b64_u32<(32 * j + 8 * q) * 2>(bias_r[j][q], bias_lds + (wn0 + 4 * hi) * 2);
    });
    wait_lgkm<0>();
    MD_PIN();
    static_for<0, 8>([&](auto pc) {
      constexpr int PASS = decltype(pc)::value, i = PASS >> 1, jp = PASS & 1;
      if constexpr (EPI == MD_EPI_RESIDUAL && PASS + 1 < 8) load_residual(PASS + 1, rres[(PASS + 1) & 1]);
      MD_PIN();
      static_for<0, 8>([&](auto jq) {
        constexpr int jj = decltype(jq)::value / 4, q = decltype(jq)::value % 4, base = 16 * (4 * i + 2 * jp + jj) + 4 * q;
        u32x2 w;
        w[0] = pack_bf16x2(acc_read<base + 0>() + lo_bf(bias_r[2 * jp + jj][q][0]), acc_read<base + 1>() + hi_bf(bias_r[2 * jp + jj][q][0]));
        w[1] = pack_bf16x2(acc_read<base + 2>() + lo_bf(bias_r[2 * jp + jj][q][1]), acc_read<base + 3>() + hi_bf(bias_r[2 * jp + jj][q][1]));
        constexpr int ch = 4 * jj + q;
        ds_write_b64_asm(tile_lds + l31 * 128 + ((ch ^ (l31 & 7)) * 16) + hi * 8, w);
      });
      MD_PIN();
      u32x4 tv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = q * 64 + lane, row = idx >> 3, ch = idx & 7;
        ds_read_b128_u32(tv[q], tile_lds + row * 128 + ((ch ^ (row & 7)) * 16));
      }
      wait_lgkm<0>();
      MD_PIN();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u32x4 v = tv[q];
        if constexpr (EPI == MD_EPI_RESIDUAL) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = pack_bf16x2(lo_bf(rres[PASS & 1][q][e]) + lo_bf(v[e]), hi_bf(rres[PASS & 1][q][e]) + hi_bf(v[e]));
        }
        const uint32_t off = store_off(i, q, jp);
        if constexpr (ABL & 128) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_cr, off, 0, 2);
        else if constexpr (!(ABL & 16)) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_cr, off, 0, 0);
        else keep_alive(v);
      }
      MD_PIN();
    });
    } else {
    // ---- bias / GELU layers: registers only, no trip through LDS --------------------------------------------------
    // This lane's 64 bias values (column 32 j + 8 q + 4 hi + e), unpacked ONCE per tile from the LDS-resident vector (they
    // are reused by the four row blocks: unpacking at every use cost 256 of the epilogue's ~1150 VALU instructions).
    md_f32x2 bias_f[4][4][2];
    {
      u32x2 bw[4][4];
      static_for<0, 16>([&](auto jq) {
        constexpr int j = decltype(jq)::value / 4, q = decltype(jq)::value % 4;
        ds_read_b64_u32<(32 * j + 8 * q) * 2>(bw[j][q], bias_lds + (wn0 + 4 * hi) * 2);
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
    // Stores go through a buffer resource over C with num_records = M x ldc x 2 bytes: rows past M are dropped by the range
    // check (no exec masking, no branch), the address is ONE 32-bit row offset per row block plus an immediate per piece
    // (per-store 64-bit address arithmetic and an exec-mask branch were another ~320 instructions per tile).  Columns past
    // n_store exist only in a layer's last column tile: there the offset of an out-of-range piece is pushed out of range.
    const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)min((uint64_t)p.M * (uint64_t)p.ldc * 2, (uint64_t)0xffffffffu), 0x00020000);
    const bool full_cols = wn0 + 128 <= p.n_store;   // wave-uniform
    // Eight passes of 32 rows x 64 columns (row block i, column half jp).  Bias add + ONE bf16 rounding happen on the
    // accumulator layout (a lane: one row, quads of 4 consecutive columns; the two lane halves hold the two quads of an
    // 8-column group).  Two v_permlane32_swap per pair of groups hand every lane a full 16-byte row piece -- lanes
    // 0-31 the even group, lanes 32-63 the odd one, adjacent in memory -- so GELU and the global stores work on
    // 16-byte pieces: piece t of a pass: row l31, columns 64 jp + 32 (t >> 1) + 16 (t & 1) + 8 hi.
    auto store_tile = [&](auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;  // every column of this wave's 128 is inside n_store: no per-piece test
    static_for<0, 8>([&](auto pc) {
      constexpr int PASS = decltype(pc)::value, i = PASS >> 1, jp = PASS & 1;
      const uint32_t row_off = (uint32_t)(wm0 + 32 * i + l31) * (uint32_t)(p.ldc * 2) + (uint32_t)(wn0 + 8 * hi) * 2u;
      static_for<0, 4>([&](auto tc) {
        // piece t: groups q = 2 (t & 1) and q + 1 of column block j = 2 jp + (t >> 1)
        constexpr int T = decltype(tc)::value, j = 2 * jp + (T >> 1), q0 = 2 * (T & 1);
        constexpr int base0 = 16 * (4 * i + j) + 4 * q0, base1 = base0 + 4;
        const md_f32x2 x0 = md_f32x2{acc_read<base0 + 0>(), acc_read<base0 + 1>()} + bias_f[j][q0][0];
        const md_f32x2 x1 = md_f32x2{acc_read<base0 + 2>(), acc_read<base0 + 3>()} + bias_f[j][q0][1];
        const md_f32x2 y0 = md_f32x2{acc_read<base1 + 0>(), acc_read<base1 + 1>()} + bias_f[j][q0 + 1][0];
        const md_f32x2 y1 = md_f32x2{acc_read<base1 + 2>(), acc_read<base1 + 3>()} + bias_f[j][q0 + 1][1];
        const uint32_t a0 = pack_bf16x2(x0[0], x0[1]), a1 = pack_bf16x2(x1[0], x1[1]);
        const uint32_t b0 = pack_bf16x2(y0[0], y0[1]), b1 = pack_bf16x2(y1[0], y1[1]);
        // swap(x, y): x' = {x of lanes 0-31, y of lanes 0-31}, y' = {x of lanes 32-63, y of lanes 32-63}
        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
        if constexpr (EPI == MD_EPI_GELU || EPI == MD_EPI_QKV_ROPE) {
          if (wn0 + 32 * j >= p.gelu_from) {  // wave-uniform: gelu_from is a multiple of 64 (fused [qkv | fc1] layers)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const md_f32x2 ge = gelu_tanh_f32x2(md_f32x2{lo_bf(v[e]), hi_bf(v[e])});
              v[e] = pack_bf16x2(ge[0], ge[1]);
            }
          }
        }
        constexpr int col_off = (32 * j + 8 * q0) * 2;  // bytes from the row offset's column (wn0 + 8 hi)
        uint32_t off = row_off;
        if constexpr (!FULL) off = (wn0 + 32 * j + 8 * (q0 + hi) < p.n_store) ? row_off : 0xfffff000u;  // out of range: dropped
        if constexpr (ABL & 128) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_c, off + col_off, 0, 2);
        else if constexpr (!(ABL & 16)) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_c, off + col_off, 0, 0);
        else keep_alive(v);
      });
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
      if (wn0 < 3 * p.rope_d) rope_tile();
      else if (full_cols) store_tile(std::true_type{});
      else store_tile(std::false_type{});
    } else {
      if (full_cols) store_tile(std::true_type{}); else store_tile(std::false_type{});
    }
    }
    if constexpr (ABL & 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the next tile's first fragments again (the copy read before the epilogue was not kept: 32
    // registers the epilogue does not have to carry); its first pair was published by the last barrier above
    read_first_frags();
    if constexpr (ABL & 64) { st_epi += __builtin_readcyclecounter() - te0; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  }
  if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA of the run-on stream in flight at the end of the wave
  if constexpr (ABL & 64) {
    if (lane == 0 && p.slabs != nullptr) {
      float* o = p.slabs + (blockIdx.x * 4 + wave) * 8;
      o[0] = (float)st_n; o[1] = (float)(st_last - st_first); o[2] = (float)st_lgkm; o[3] = (float)st_vm; o[4] = (float)st_bar; o[5] = (float)st_epi;
    }
  }
}

int g_w4_grid = 0;     // md_gemm_set_tuning "w4_grid": persistent workgroups per launch (0 = one per CU); a multiple of 8
// md_gemm_set_tuning "w4_variant": low 4 bits = operand path / schedule (MODE: 0 register-staged, 1-3 LDS-DMA schedules; 2 and 3
// exist for the bias epilogue only), the rest 16 * ABL (measurement builds, bias epilogue only)
int g_w4_variant = [] { const char* e = getenv("MD_W4_VARIANT"); return (e && *e) ? atoi(e) : 0; }();

uint64_t g_w4_debug = 0;  // measurement builds: device buffer for the in-kernel stamps (md_gemm_set_tuning "w4_dbg_lo" / "w4_dbg_hi")

template <int EPI, int ABL = 0, int MODE = 1>
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
  // one persistent workgroup per CU -- or fewer ("w4_grid"): a workgroup owns its CU (156 KiB of LDS, the whole register file),
  // so a smaller grid leaves whole CUs to kernels of another stream (the pipelined engine's decode steps)
  const int gx = std::min(nwg, g_w4_grid > 0 ? std::min(g_w4_grid, n_cu) : n_cu);
  hipLaunchKernelGGL(fn, dim3(gx), dim3(256), LDS_BYTES, stream, kk);
  return md_launch_status();
}

template <int MODE>
md_status launch_mode(const GemmK& k, int epi, hipStream_t stream) {
  switch (epi) {
    case MD_EPI_BIAS: return launch<MD_EPI_BIAS, 0, MODE>(k, stream);
    case MD_EPI_GELU: return launch<MD_EPI_GELU, 0, MODE>(k, stream);
    case MD_EPI_RESIDUAL: return launch<MD_EPI_RESIDUAL, 0, MODE>(k, stream);
    case MD_EPI_QKV_ROPE: return launch<MD_EPI_QKV_ROPE, 0, MODE>(k, stream);
    default: return MD_ERR_INVALID_ARG;
  }
}

}  // namespace

int md_gemm_w4_residual_max_cols() { return bias_max_cols<MD_EPI_RESIDUAL>(); }
int md_gemm_w4_max_cols(int epi) { return epi == MD_EPI_RESIDUAL ? bias_max_cols<MD_EPI_RESIDUAL>() : bias_max_cols<MD_EPI_BIAS>(); }

void md_gemm_w4_set_variant(int v) { g_w4_variant = v; }
void md_gemm_w4_set_debug(int half, uint32_t v) { g_w4_debug = half ? ((g_w4_debug & 0xffffffffull) | ((uint64_t)v << 32)) : ((g_w4_debug & ~0xffffffffull) | v); }
void md_gemm_w4_set_grid(int v) { g_w4_grid = v > 0 ? std::max(8, v / 8 * 8) : 0; }

bool md_gemm_w4_takes(const GemmK& k, int epi) {
  if (k.K % 64 != 0 || k.M <= 0) return false;
  // 32-bit byte offsets into A and W
  if ((uint64_t)k.M * (uint64_t)k.lda * 2 >= (1ull << 32) || (uint64_t)k.n_pad * (uint64_t)k.ldw * 2 >= (1ull << 32)) return false;
  // every epilogue takes its bias from the LDS-resident vector only (md_gemm_bf16 sends wider layers to the eight-wave kernel)
  if (k.n_pad > md_gemm_w4_max_cols(epi)) return false;
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
  const int mode = g_w4_variant & 15, abl = g_w4_variant >> 4;  // (ablation codes: mode + 16 * ABL)
#ifdef MD_W4_ABLATIONS  // measurement builds only (MD_W4_ABLATIONS=1 python -c "import __graft_entry__ as g; g.build()")
  if (epi == MD_EPI_BIAS && abl != 0) {
    switch (256 * mode + abl) {
      case 256 * 0 + 2: return launch<MD_EPI_BIAS, 2, 0>(k, stream);
      case 256 * 0 + 8: return launch<MD_EPI_BIAS, 8, 0>(k, stream);
      case 256 * 0 + 16: return launch<MD_EPI_BIAS, 16, 0>(k, stream);
      case 256 * 0 + 32: return launch<MD_EPI_BIAS, 32, 0>(k, stream);
      case 256 * 1 + 2: return launch<MD_EPI_BIAS, 2, 1>(k, stream);
      case 256 * 1 + 4: return launch<MD_EPI_BIAS, 4, 1>(k, stream);
      case 256 * 1 + 8: return launch<MD_EPI_BIAS, 8, 1>(k, stream);
      case 256 * 1 + 14: return launch<MD_EPI_BIAS, 14, 1>(k, stream);
      case 256 * 1 + 16: return launch<MD_EPI_BIAS, 16, 1>(k, stream);
      case 256 * 1 + 32: return launch<MD_EPI_BIAS, 32, 1>(k, stream);
      case 256 * 1 + 46: return launch<MD_EPI_BIAS, 46, 1>(k, stream);
      case 256 * 1 + 64: return launch<MD_EPI_BIAS, 64, 1>(k, stream);
      case 256 * 3 + 64: return launch<MD_EPI_BIAS, 64, 3>(k, stream);
      default: return MD_ERR_INVALID_ARG;
    }
  }
#endif
  if (abl != 0) return MD_ERR_INVALID_ARG;
  switch (mode) {
    case 0: return launch_mode<0>(k, epi, stream);
    case 1: return launch_mode<1>(k, epi, stream);
    case 2: return epi == MD_EPI_BIAS ? launch<MD_EPI_BIAS, 0, 2>(k, stream) : MD_ERR_INVALID_ARG;
    case 3: return epi == MD_EPI_BIAS ? launch<MD_EPI_BIAS, 0, 3>(k, stream) : MD_ERR_INVALID_ARG;
    default: return MD_ERR_INVALID_ARG;
  }
}
