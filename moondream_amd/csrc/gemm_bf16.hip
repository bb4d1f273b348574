// bf16 GEMM on the CDNA4 matrix cores:  C = epilogue(A . W^T + b)
//
// This is the kernel that carries ~90 % of the encode FLOPs (SURVEY.md section
// 8a rows a4/a6/a9/a12: every nn.Linear of the ViT, the projector and the
// decoder prefill).  Both operands are K-contiguous (activations [M,K], weights
// in the nn.Linear layout [N,K]), so both MFMA fragments are plain 16-byte
// K-runs of one row.
//
// Structure (one workgroup = one BM x BN tile of C, 64-wide wavefronts):
//   * K is walked in 64-element slices.  A and W slices are copied HBM/L2 -> LDS
//     by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip) into a 2-deep
//     ring; slice t+1 is in flight while slice t feeds the matrix cores.
//   * LDS image per operand: [rows][8 x 16 B], 16-byte chunk c of row r stored
//     at physical chunk c ^ ((r >> 1) & 7).  LDS-DMA writes lane-linear, so the
//     permutation is applied on the per-lane SOURCE address and again on the
//     ds_read_b128 address; the 16-lane groups of a b128 read then touch 16
//     distinct 16-byte bank slots (conflict-free).
//   * v_mfma_f32_32x32x16_bf16 with the weight fragment as the first operand:
//     the accumulator then holds, per lane, one row m and runs of 4 consecutive
//     columns n, so (acc + bias) packs straight into 8-byte bf16 quads.
//   * epilogue: bias add in fp32, ONE rounding to bf16 (the reference's
//     F.linear rounding point), transposition through a wave-private LDS tile,
//     then GELU / residual add on whole 16-byte row segments with fully
//     coalesced global traffic.
//   * workgroup id -> tile: XCD-contiguous remap, then grouped (8 row panels x
//     n) ordering so the 32 CUs of one XCD work on a compact 2-D block of
//     tiles and share A/W slices through their private L2.
#include "gemm_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

namespace {

// Optional live timing (bench.py's roofline leg): HIP events recorded on the
// caller's stream around every GEMM launch, read back with md_profile_gemm_read.
struct ProfRec {
  hipEvent_t start, stop;
  double work;  // kind 0: algorithmic flops; kind 1: weight bytes streamed
  double rd, wr;  // algorithmic operand bytes read (A + W + residual, padded dims) / result bytes written
  int kind;     // 0 = MFMA tile kernel, 1 = decode-regime weight-streaming kernel
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;

// Experiment knobs, read ONCE (A/B runs; the product path never sets them).
struct Knobs {
  int tile = -1;       // MD_GEMM_TILE: force a tile config
  int group_m = 0;     // MD_GEMM_GROUP_M: row panels per tile-order group (0 = by shape, md_gemm_auto_group_m)
  int w4 = 1;          // MD_GEMM_W4=0: the eight-wave 256x256 kernels instead of the four-wave one
  int persist = 1;     // MD_GEMM_PERSIST=0 (eight-wave kernels only)
  int nt = 0;          // MD_DECODE_NT=1: stream decode-regime weights non-temporally
  int decode_cfg = 16; // MD_DECODE_CFG: d / 6 = alternatives to the 64x64 + helper-waves config
  int decode_slices = 0;  // MD_DECODE_SLICES
  int rope_fuse = 1;      // MD_ROPE_FUSE=0: prefill RoPE + KV write as their own kernel again (A/B, tests)
  int small_m_rule = 1;   // MD_SMALL_M_RULE=0: round 2's single-image tile rule (cost model for everything above 128 tiles of 128 x 128)
  Knobs() {
    auto geti = [](const char* n, int d) { const char* e = getenv(n); return (e && *e) ? atoi(e) : d; };
    tile = geti("MD_GEMM_TILE", -1);
    group_m = std::max(0, geti("MD_GEMM_GROUP_M", 0));
    w4 = geti("MD_GEMM_W4", 1);
    persist = geti("MD_GEMM_PERSIST", 1);
    nt = geti("MD_DECODE_NT", 0);
    decode_slices = geti("MD_DECODE_SLICES", 0);
    rope_fuse = geti("MD_ROPE_FUSE", 1);
    small_m_rule = geti("MD_SMALL_M_RULE", 1);
    if (const char* dc = getenv("MD_DECODE_CFG")) {
      if (dc[0] == 'd') decode_cfg = 3;
      else if (dc[0] == '6') decode_cfg = 10;
      else if (dc[0] == '1' && dc[1] >= '6' && dc[1] <= '9') decode_cfg = 10 + (dc[1] - '0');  // "16" .. "19"
    }
  }
};
Knobs& knobs() {
  static Knobs k;
  return k;
}

constexpr int BK = 64;  // K granularity of the packing contract (k_pad % 64 == 0) and default slice width

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// ds_read_b128 the compiler does not track: the caller waits (wait_lgkm) before use
template <int OFF>
__device__ __forceinline__ void ds_read_b128(bf16x8& dst, uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
// LDS store / load of the epilogue transposition the compiler does not see (persistent tile
// loop: with visible LDS accesses hipcc parks an s_waitcnt vmcnt(0) in front of them while the
// next tile's LDS-DMA prefetch is in flight)
__device__ __forceinline__ void ds_write_b64_asm(uint32_t addr, u32x2 v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void ds_read_b128_u32(u32x4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// STAGES = depth of the LDS operand ring: 2 for the MFMA-bound big tiles (one slice
// in flight under ~2000 cycles of MFMA work), deeper for the decode-regime config
// whose per-slice compute is far shorter than the DMA latency.
// BKT = K elements per slice (64: 128-byte rows, 8 chunks; 32: 64-byte rows, 4 chunks,
// which lets the 256x256 tile keep three 32 KiB slices in flight in a 4-deep ring).
// XW = extra DMA-only waves: they issue their share of every slice's LDS-DMA pieces and take
// part in the barriers, nothing else.  An LDS-DMA instruction costs its wave ~100+ cycles, so in
// the decode regime (tiny MFMA work per slice) a workgroup's stream is bound by how many waves
// issue it (probe: 67 / 108 / 120 GB/s per CU with 2 / 4 / 8 issuing waves).
template <int BM, int BN, int WM, int WN, int EPI, bool SPLITK, int STAGES, int BKT, int PP, int XW>
__device__ __forceinline__ void gemm_body(const GemmK& p) {
  constexpr int ROW_BYTES = BKT * 2;
  constexpr int CH = BKT / 8;                 // 16-byte chunks per row
  constexpr int CH_SHIFT = (CH == 16) ? 4 : (CH == 8) ? 3 : 2;
  constexpr int KSTEPS = BKT / 16;            // MFMA K-steps per slice
  // BKT = 128 (round 5, decode regime): 256-byte rows, 16 chunks XOR-swizzled by row & 15 -- a 16-lane group of a ds_read_b128
  // touches 16 distinct chunks = every bank once.  Twice the K per ring iteration: the iteration's fixed latencies (counted
  // vmcnt, barrier, the first fragments' LDS latency: ~300 of ~750 cycles at 64) are paid half as often.
  static_assert(BKT == 128 || BKT == 64 || BKT == 32, "slice width");
  constexpr int CT = WM * WN * 64;       // threads that compute
  constexpr int NT = XW > 0 ? XW * 64 : CT;  // threads that move data: the helper waves when there are any (round 3: the compute
                                             // waves' share of the LDS-DMA issue, ~40 cycles per K-step, sat on their MFMA chain while the
                                             // helpers idled ~400 cycles per slice at the barrier), else every wave
  static_assert(XW == 0 || PP == 0, "helper waves: lockstep schedule only");
  constexpr int TM = BM / WM, TN = BN / WN;
  static_assert(TN == 64 || TN == 32, "epilogue transposes 32 x 64 (or 32 x 32) wave tiles");
  constexpr int CPRW = TN / 8;               // 16-byte pieces per row of a wave tile
  constexpr int NQ = 32 * CPRW / 64;         // pieces per lane of a 32-row wave tile
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int NA = BM * CH / NT, NB = BN * CH / NT;  // 16-byte LDS-DMA pieces per thread
  static_assert(BM * CH % NT == 0 && BN * CH % NT == 0, "tile/threads mismatch");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_compute = (XW == 0) || wave < WM * WN;  // wave-uniform
  const bool is_dma = (XW == 0) || !is_compute;
  const int dtid = XW > 0 ? max(tid - CT, 0) : tid;      // index among the data-moving threads
  const int dwave = XW > 0 ? max(wave - WM * WN, 0) : wave;
  const int wm = wave / WN, wn = wave % WN;
  const int hi = lane >> 5, l31 = lane & 31;

  // ---- workgroup -> tile -------------------------------------------------
  // PP == 3: persistent workgroups (one per CU) walk the tile sequence with stride gridDim.x;
  // v % 8 == blockIdx.x % 8 (gridDim.x is a multiple of 8 or covers every tile), so the XCD
  // each tile lands on is the one xcd_remap assumes.
  constexpr bool PERSIST = (PP == 3);
  constexpr bool ALT = (PP == 2 || PP == 3);
  const int nwg = p.tiles_m * p.tiles_n;
  const int GROUP_M = p.group_m;
  const int per_group = GROUP_M * p.tiles_n;
  int tm, tn, m0, n0;
  // per-thread LDS-DMA source pointers (advance one slice width per K slice)
  const char* a_src[NA];
  const char* b_src[NB];
  auto set_tile = [&](int vv) {
    const int L = xcd_remap(vv, nwg);
    const int first_m = (L / per_group) * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    tm = first_m + (L % per_group) % gsz;
    tn = (L % per_group) / gsz;
    m0 = tm * BM;
    n0 = tn * BN;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int slot = j * NT + dtid, r = slot >> CH_SHIFT, c = (slot & (CH - 1)) ^ ((r >> (4 - CH_SHIFT)) & (CH - 1));
      const int64_t row = min(m0 + r, p.M - 1);
      a_src[j] = (const char*)(p.A + row * p.lda + c * 8);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int slot = j * NT + dtid, r = slot >> CH_SHIFT, c = (slot & (CH - 1)) ^ ((r >> (4 - CH_SHIFT)) & (CH - 1));
      const int64_t row = min(n0 + r, p.n_pad - 1);
      b_src[j] = (const char*)(p.W + row * p.ldw + c * 8);
    }
  };
  int vtile = blockIdx.x;
  if (vtile >= nwg) return;                                    // paired launches: grid covers the larger problem
  if constexpr (SPLITK) if ((int)blockIdx.y >= p.slices) return;
  set_tile(vtile);

  // one 16-byte-per-lane LDS-DMA piece (1 KiB per wave) of slice `stage`
  auto issue_piece = [&](auto piece_c, int stage) {
    constexpr int P = decltype(piece_c)::value;
    char* base = smem + stage * STAGE + dwave * (64 * 16);
    if constexpr (P < NA) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_src[P],
                                       (__attribute__((address_space(3))) void*)(base + P * NT * 16), 16, 0, 0);
      a_src[P] += ROW_BYTES;
    } else {
      constexpr int Q = P - NA;
      // (a non-temporal policy on the decode regime's weight stream was a runtime choice until round 3: it never changed
      // a step time, and the branch per piece sat in the loop)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b_src[Q],
                                       (__attribute__((address_space(3))) void*)(base + A_BYTES + Q * NT * 16), 16, 0, 0);
      b_src[Q] += ROW_BYTES;
    }
  };

  f32x16 acc[MI][NI];

  // fragment read offsets: row = (tile row base, multiple of 32) + l31, so the
  // swizzle term depends on the lane only: (row >> 1) & 7 for 128-byte rows, (row >> 2) & 3 for 64-byte rows
  const int swz = (l31 >> (4 - CH_SHIFT)) & (CH - 1);
  const uint32_t a_row_off = (wm * TM + l31) * ROW_BYTES;
  const uint32_t b_row_off = A_BYTES + (wn * TN + l31) * ROW_BYTES;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  int nk = p.K / BKT;
  if constexpr (SPLITK) {
    // this workgroup's contiguous range of K slices (a function of the layer shape only)
    const int per = (nk + p.slices - 1) / p.slices;
    const int t0 = min((int)blockIdx.y * per, nk);
    nk = min(t0 + per, nk) - t0;
#pragma unroll
    for (int j = 0; j < NA; ++j) a_src[j] += (int64_t)t0 * ROW_BYTES;
#pragma unroll
    for (int j = 0; j < NB; ++j) b_src[j] += (int64_t)t0 * ROW_BYTES;
  }
  // alternating schedule: the first AHEAD slices of a tile (PERSIST: issued for the NEXT tile
  // before the current tile's epilogue, so they land under it)
  auto alt_prologue = [&]() {
    if constexpr (ALT) {
      static_for<0, STAGES - 2>([&](auto sc) {
        constexpr int SL0 = decltype(sc)::value;
        if (SL0 < nk) static_for<0, NA + NB>([&](auto pc) { issue_piece(pc, SL0); });
      });
    }
  };
  if constexpr (PERSIST) alt_prologue();
  for (;;) {
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if constexpr (ALT) {
    // ---- alternating wave groups (cdna guide 5: "8-phase" discipline) ----------
    // 32-wide slices, two PHASES per slice (the wave's upper / lower 64 rows), eight
    // MFMAs (256 matrix-pipe cycles) per phase on four independent accumulators.
    // Every phase is   { ds_reads + 2 LDS-DMA pieces } barrier { MFMAs } barrier   and
    // the second wave group (one wave per SIMD, like the first) runs ONE BARRIER
    // behind: while one wave of a SIMD owns the matrix pipe (raised priority), the
    // other fetches its next fragments and feeds the DMA ring, and the barriers
    // enforce the alternation.  Nothing is software-pipelined inside a wave: the
    // other group's MFMA phase is what hides the LDS latency.
    //   slice u is read in phases (u,0) [B: 4 reads, A rows 0-63: 4 reads] and (u,1)
    //   [A rows 64-127: 4 reads]; its ring slot is refilled with slice u+STAGES from
    //   phase (u+2,0) on, i.e. >= 3 phases after its last read by either group.
    //   The DMA of slice v is waited for (counted vmcnt, then the phase's first
    //   barrier) in phase (v-1,1) and first read in phase (v,0), one phase later.
    static_assert(KSTEPS == 2 && STAGES >= 4 && !SPLITK && MI == 4 && NI == 2, "alternating config");
    constexpr int PIECES = NA + NB;  // 4 per thread and slice
    static_assert(PIECES == 4, "two LDS-DMA pieces per phase");
    constexpr int AHEAD = STAGES - 2;  // slices in flight ahead of the one being computed
    static_assert((AHEAD - 1) * PIECES < 64, "vmcnt is a 6-bit counter");
    const int lag = (wave >= (WM * WN) / 2) ? 1 : 0;  // SGPR: uniform per wave
    if constexpr (!PERSIST) alt_prologue();
    if (nk >= AHEAD) wait_vm<(AHEAD - 1) * PIECES>(); else wait_vm<0>();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();  // slice 0 visible to everybody
    if (lag) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bf16x8 af[2][2], bfr[2][NI];
    for (int u = 0; u < nk; ++u) {
      const uint32_t st = lds_base + (u % STAGES) * STAGE;
      const int v = u + AHEAD;              // slice whose DMA is issued during slice u
      const bool has_next = v < nk;
      const int nstage = v % STAGES;
      static_for<0, 2>([&](auto hc) {
        constexpr int H = decltype(hc)::value;
        // fragments: K-step s -> 16-byte chunk (2s + hi) of the row, swizzled
        static_for<0, 2>([&](auto sc) {
          constexpr int S = decltype(sc)::value;
          const uint32_t coff = (uint32_t)(((2 * S + hi) ^ swz) * 16);
          if constexpr (H == 0)
            static_for<0, NI>([&](auto j) { ds_read_b128<decltype(j)::value * 32 * ROW_BYTES>(bfr[S][decltype(j)::value], st + b_row_off + coff); });
          static_for<0, 2>([&](auto i) { ds_read_b128<(2 * H + decltype(i)::value) * 32 * ROW_BYTES>(af[S][decltype(i)::value], st + a_row_off + coff); });
        });
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) {
          issue_piece(std::integral_constant<int, 2 * H>{}, nstage);
          issue_piece(std::integral_constant<int, 2 * H + 1>{}, nstage);
        }
        if constexpr (H == 1) {
          // slice u+1 (first read in the next phase) has landed; the AHEAD-1 younger slices stay in flight
          if (has_next) wait_vm<(AHEAD - 1) * PIECES>(); else wait_vm<0>();
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        wait_lgkm<0>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        static_for<0, 2>([&](auto sc) {
          constexpr int S = decltype(sc)::value;
          static_for<0, 2>([&](auto i) {
            static_for<0, NI>([&](auto j) {
              constexpr int I = 2 * H + decltype(i)::value, J = decltype(j)::value;
              acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[S][J], af[S][decltype(i)::value], acc[I][J], 0, 0, 0);
            });
          });
        });
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      });
    }
    if (!lag) __builtin_amdgcn_s_barrier();  // the leading group owes the barrier the lagging one took first
  } else {
    static_assert((STAGES - 2) * (NA + NB) < 64, "vmcnt is a 6-bit counter");
    // prologue: slices 0 .. STAGES-2 in flight
    static_for<0, STAGES - 1>([&](auto sc) {
      constexpr int SL = decltype(sc)::value;
      if (SL < nk && is_dma) static_for<0, NA + NB>([&](auto pc) { issue_piece(pc, SL); });
    });
    // One iteration of the ring.  STEADY (compile time): slice t + STAGES - 1 exists and STAGES - 2 younger slices are in
    // flight -- no per-piece "is there a next slice" test, one fixed vmcnt: the steady-state loop carries no scalar branches
    // (a taken branch costs a wave ~16-20 cycles; the general form has ~15 of them per 64-wide slice, which at 8 MFMAs per
    // slice was a third of the decode-regime kernels' loop time).
    auto ring_iteration = [&](int t, auto steady_c) {
      constexpr bool STEADY = decltype(steady_c)::value;
      // slice t has landed: own DMA by a COUNTED vmcnt (the min(STAGES-2, nk-1-t)
      // younger slices stay in flight), everybody's by the barrier, which also fences
      // the previous iteration's reads of the ring slot refilled during this one
      if constexpr (STAGES == 2) {
        wait_vm<0>();
      } else if constexpr (STEADY) {
        wait_vm<(STAGES - 2) * (NA + NB)>();
      } else {
        const int ahead = min(STAGES - 2, nk - 1 - t);
        static_for<0, STAGES - 1>([&](auto ac) {
          constexpr int A = decltype(ac)::value;
          if (ahead == A) wait_vm<A*(NA + NB)>();
        });
      }
      // raw s_barrier: __syncthreads() would add a vmcnt(0) and drain the DMA ring.
      // Every ds_read of this wave was waited for (wait_lgkm<0>) before its last MFMAs.
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const bool has_next = STEADY || (t + STAGES - 1 < nk);
      const int nstage = (t + STAGES - 1) % STAGES;
      if constexpr (XW > 0) {
        if (!is_compute) {  // DMA-only wave: its pieces of slice t+STAGES-1, then the next barrier
          if (has_next) static_for<0, NA + NB>([&](auto pc) { issue_piece(pc, nstage); });
          return;
        }
      }
      // LDS -> register fragments, software pipelined by hand: the six ds_read_b128 of
      // K-step s+1 are issued BEFORE the eight MFMAs of step s and waited for with a
      // counted lgkmcnt (LDS returns in order), so the matrix pipe never waits on LDS
      // latency inside a slice.  (The two waves of a SIMD leave the barrier in
      // lockstep and cannot cover for each other; left to itself hipcc sinks every
      // read next to its use and waits lgkmcnt(0) in front of each MFMA group.)
      // Inline asm: the compiler neither counts these reads nor moves MFMAs across
      // the sched_barrier that follows each wait (cdna guide 5.7, form iii).
      const uint32_t st = lds_base + (t % STAGES) * STAGE;
      bf16x8 af[2][MI], bfr[2][NI];
      auto issue_reads = [&](auto set_c, auto step_c) {
        constexpr int SET = decltype(set_c)::value, S = decltype(step_c)::value;
        const uint32_t coff = (uint32_t)(((2 * S + hi) ^ swz) * 16);
        const uint32_t a_addr = st + a_row_off + coff, b_addr = st + b_row_off + coff;
        static_for<0, NI>([&](auto j) { ds_read_b128<decltype(j)::value * 32 * ROW_BYTES>(bfr[SET][decltype(j)::value], b_addr); });
        static_for<0, MI>([&](auto i) { ds_read_b128<decltype(i)::value * 32 * ROW_BYTES>(af[SET][decltype(i)::value], a_addr); });
      };
      issue_reads(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      static_for<0, KSTEPS>([&](auto sc) {
        constexpr int S = decltype(sc)::value, SET = S & 1;
        if constexpr (S + 1 < KSTEPS) {
          issue_reads(std::integral_constant<int, (S + 1) & 1>{}, std::integral_constant<int, S + 1>{});
          wait_lgkm<MI + NI>();  // everything but the reads just issued has landed
        } else {
          wait_lgkm<0>();
        }
        __builtin_amdgcn_sched_barrier(0);
        // The LDS-DMA pieces of slice t+1 are spread over the four K-steps and issued
        // BETWEEN MFMAs: an LDS-DMA issue costs the wave 60-180 cycles, which hides
        // under the 32-cycle-per-MFMA matrix pipe instead of idling it at the head
        // of the slice.  Piece p goes to step p % KSTEPS, slot p / KSTEPS.
        constexpr int PPS = (NA + NB + KSTEPS - 1) / KSTEPS;  // pieces per K-step
        static_for<0, MI * NI>([&](auto mc) {
          constexpr int Mx = decltype(mc)::value, I = Mx / NI, J = Mx % NI;
          acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[SET][J], af[SET][I], acc[I][J], 0, 0, 0);
          static_for<0, PPS>([&](auto qc) {
            constexpr int Q = decltype(qc)::value, P = S + KSTEPS * Q;
            // slot Q of this step sits behind MFMA number Q * (MI*NI) / PPS
            if constexpr (XW == 0 && (Q * MI * NI) / PPS == Mx && P < NA + NB) {  // (with helper waves the compute waves move nothing)
              __builtin_amdgcn_sched_barrier(0);
              if (has_next) issue_piece(std::integral_constant<int, P>{}, nstage);
              __builtin_amdgcn_sched_barrier(0);
            }
          });
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    const int n_steady = max(0, nk - (STAGES - 1));
    for (int t = 0; t < n_steady; ++t) ring_iteration(t, std::true_type{});
    for (int t = n_steady; t < nk; ++t) ring_iteration(t, std::false_type{});
  }

  if constexpr (SPLITK) {
    if (p.partial != nullptr) {
      // acc[i][j][r]: row m = 32 i + l31, col n = 32 j + 8 (r >> 2) + 4 hi + (r & 3)
      float* dst = p.partial + (int64_t)blockIdx.y * p.partial_slice_stride;
      if (is_compute)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * TM + 32 * i + l31;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * TN + 32 * j + 8 * g + 4 * hi;
            if (m < p.M && n < p.n_store)
              *(f32x4*)(dst + (int64_t)m * p.partial_ld + n) =
                  f32x4{acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          }
      }
      return;
    }
    // Cross-workgroup split-K, deterministic: every slice publishes its fp32
    // accumulators as a slab in accumulator-register order (coalesced, and each
    // lane later reads back exactly its own registers); the LAST workgroup to
    // arrive (agent-scope release -> ticket -> acquire, cdna guide 6 G16) sums the
    // slabs in slice order and runs the epilogue.  The summation tree depends on
    // the layer shape only, never on how many rows are live.
    constexpr int SLAB = CT * MI * NI * 16;  // floats per workgroup
    const int tile_id = tm * p.tiles_n + tn;
    float* slab = p.slabs + ((int64_t)tile_id * p.slices + blockIdx.y) * SLAB;
    if (is_compute) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) slab[((i * NI + j) * 16 + r) * CT + tid] = acc[i][j][r];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* flag = (unsigned*)smem;  // operand ring is dead after the barrier
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned tk = __hip_atomic_fetch_add(p.tickets + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = (tk == (unsigned)p.slices - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (*flag == 0u) return;
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(p.tickets + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // leave zero for the next launch
    }
    __syncthreads();
    const float* base = p.slabs + (int64_t)tile_id * p.slices * SLAB;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (is_compute) {
      for (int sl = 0; sl < p.slices; ++sl)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += base[(int64_t)sl * SLAB + ((i * NI + j) * 16 + r) * CT + tid];
    }
  }

  // ---- epilogue -----------------------------------------------------------
  // acc[i][j][r]: row m = 32 i + l31, col n = 32 j + 8 (r >> 2) + 4 hi + (r & 3)
  __syncthreads();  // all waves are done reading the operand ring
  if constexpr (XW > 0) {
    if (!is_compute) return;
  }
  const int m0c = m0, n0c = n0;  // the tile whose accumulators are being stored
  char* tile = smem + (PERSIST ? STAGES * STAGE : 0) + wave * 4096;  // wave-private 32 x 64 bf16 transposition tile
  const uint32_t tile_lds = lds_base + (PERSIST ? STAGES * STAGE : 0) + wave * 4096;
  const int wn0 = n0c + wn * TN;

  float bias_v[NI][4][4];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = wn0 + 32 * j + 8 * g + 4 * hi;
      u32x2 bw = {0u, 0u};
      if (p.bias != nullptr && n < p.n_pad) bw = *(const u32x2*)(p.bias + n);
      bias_v[j][g][0] = lo_bf(bw[0]);
      bias_v[j][g][1] = hi_bf(bw[0]);
      bias_v[j][g][2] = lo_bf(bw[1]);
      bias_v[j][g][3] = hi_bf(bw[1]);
    }

  // residual operand: all MI x 4 16-byte pieces of this lane are requested up front
  // (one exposed memory latency per tile instead of one per 32-row block); r may
  // alias c, each piece is read and later written by the same lane
  constexpr bool PREFETCH_R = (EPI == MD_EPI_RESIDUAL) && MI > 1;  // small decode tiles keep their register budget low
  u32x4 rres[PREFETCH_R ? MI : 1][4];
  if constexpr (PREFETCH_R) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int idx = q * 64 + lane, row = idx / CPRW, ch = idx % CPRW;
        const int m = m0c + wm * TM + 32 * i + row;
        const int n = wn0 + ch * 8;
        rres[i][q] = u32x4{0, 0, 0, 0};
        if (m < p.M && n < p.n_store) {
          const int64_t rrow = p.res_row_mod ? (m % p.res_row_mod) : m;
          rres[i][q] = *(const u32x4*)(p.R + rrow * p.ldr + n);
        }
      }
  }

  bool more = false;
  if constexpr (PERSIST) {
    // next tile's first slices are requested now (AFTER this tile's bias / residual loads, which
    // would otherwise queue behind them) and land under this epilogue; the transposition
    // scratch lives behind the ring so the two do not meet
    vtile += gridDim.x;
    more = vtile < nwg;
    if (more) {
      set_tile(vtile);
      alt_prologue();
    }
  }

#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 w;
        w[0] = pack_bf16x2(acc[i][j][4 * g + 0] + bias_v[j][g][0], acc[i][j][4 * g + 1] + bias_v[j][g][1]);
        w[1] = pack_bf16x2(acc[i][j][4 * g + 2] + bias_v[j][g][2], acc[i][j][4 * g + 3] + bias_v[j][g][3]);
        const int ch = 4 * j + g;
        if constexpr (PERSIST)
          ds_write_b64_asm(tile_lds + l31 * 128 + ((ch ^ (l31 & 7)) * 16) + hi * 8, w);
        else
          *(u32x2*)(tile + l31 * 128 + ((ch ^ (l31 & 7)) * 16) + hi * 8) = w;
      }
    u32x4 tv[4];
    if constexpr (PERSIST) {
      // same-wave LDS operations execute in order: the reads see the writes above
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int idx = q * 64 + lane, row = idx / CPRW, ch = idx % CPRW;
        ds_read_b128_u32(tv[q], tile_lds + row * 128 + ((ch ^ (row & 7)) * 16));
      }
      wait_lgkm<0>();
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int idx = q * 64 + lane, row = idx / CPRW, ch = idx % CPRW;
      u32x4 v;
      if constexpr (PERSIST)
        v = tv[q];
      else
        v = *(const u32x4*)(tile + row * 128 + ((ch ^ (row & 7)) * 16));
      const int m = m0c + wm * TM + 32 * i + row;
      const int n = wn0 + ch * 8;
      if (m < p.M && n < p.n_store) {
        if constexpr (EPI == MD_EPI_GELU) {
          if (n >= p.gelu_from) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
            {
              const md_f32x2 g = gelu_tanh_f32x2(md_f32x2{lo_bf(v[e]), hi_bf(v[e])});
              v[e] = pack_bf16x2(g[0], g[1]);
            }
          }
        } else if constexpr (EPI == MD_EPI_RESIDUAL) {
          u32x4 rv;
          if constexpr (PREFETCH_R) {
            rv = rres[i][q];
          } else {
            const int64_t rrow = p.res_row_mod ? (m % p.res_row_mod) : m;
            rv = *(const u32x4*)(p.R + rrow * p.ldr + n);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = pack_bf16x2(lo_bf(rv[e]) + lo_bf(v[e]), hi_bf(rv[e]) + hi_bf(v[e]));
        }
        *(u32x4*)(p.C + (int64_t)m * p.ldc + n) = v;
      }
    }
  }
  if (!more) break;
  // the epilogue's stores share vmcnt with the DMA ring: drain both before the next tile's counted waits
  wait_vm<0>();
  }  // tile loop
}

template <int BM, int BN, int WM, int WN, int EPI, bool SPLITK = false, int STAGES = 2, int BKT = 64, int PP = 0, int XW = 0>
__global__ __launch_bounds__((WM * WN + XW) * 64) void gemm_bf16_kernel(const GemmK p) {
  gemm_body<BM, BN, WM, WN, EPI, SPLITK, STAGES, BKT, PP, XW>(p);
}

// two independent problems in ONE launch (blockIdx.z picks the problem): the decode block's
// proj and fc2 partial-product GEMMs
struct GemmPair {
  GemmK g[2];
};
template <int BM, int BN, int WM, int WN, int EPI, bool SPLITK = false, int STAGES = 2, int BKT = 64, int PP = 0, int XW = 0>
__global__ __launch_bounds__((WM * WN + XW) * 64) void gemm_pair_kernel(const GemmPair pair) {
  gemm_body<BM, BN, WM, WN, EPI, SPLITK, STAGES, BKT, PP, XW>(pair.g[blockIdx.z]);
}

template <int BM, int BN, int WM, int WN, int EPI, bool SPLITK = false, int STAGES = 2, int BKT = 64, int PP = 0, int XW = 0>
md_status launch_cfg(const GemmK& k, hipStream_t stream) {
  constexpr int NT = (WM * WN + XW) * 64;
  constexpr int ring = STAGES * (BM + BN) * BKT * 2;
  constexpr int epi = WM * WN * 4096;
  constexpr int lds = (PP == 3) ? ring + epi : (ring > epi ? ring : epi);
  static_assert(lds <= 163840, "LDS budget");
  auto fn = gemm_bf16_kernel<BM, BN, WM, WN, EPI, SPLITK, STAGES, BKT, PP, XW>;
  MD_TRY(md_ensure_dynamic_lds((const void*)fn, lds));
  GemmK kk = k;
  kk.tiles_m = (k.M + BM - 1) / BM;
  kk.tiles_n = (k.n_store + BN - 1) / BN;
  const int nwg = kk.tiles_m * kk.tiles_n;
  const int gx = (PP == 3) ? std::min(nwg, 256) : nwg;  // persistent: one workgroup per CU
  hipLaunchKernelGGL(fn, dim3(gx, SPLITK ? kk.slices : 1), dim3(NT), lds, stream, kk);
  return md_launch_status();
}

constexpr int DEC_STAGES = 4;  // decode-regime ring: 3 x 24 KiB slices in flight per workgroup

template <int EPI>
md_status launch_epi(const GemmK& k, int tile, hipStream_t stream) {
  switch (tile) {
    case 20: return md_gemm_w4_launch(k, EPI, stream);                              // 256x256, four waves (gemm_w4.hip)
    case 1: return launch_cfg<256, 128, 4, 2, EPI>(k, stream);
    case 3: return k.slices > 1 ? launch_cfg<64, 128, 1, 2, EPI, true, DEC_STAGES>(k, stream)
                                : launch_cfg<64, 128, 1, 2, EPI, false, DEC_STAGES>(k, stream);
    // decode regime, 64x64 tiles: twice the tiles of 64x128 -> the wide fused layer needs no
    // split-K, the N = 2048 layers get 16 KiB slabs
    case 10: return k.slices > 1 ? launch_cfg<64, 64, 2, 1, EPI, true, 4>(k, stream)
                                 : launch_cfg<64, 64, 2, 1, EPI, false, 4>(k, stream);
    // the eight-wave 256x256 kernels the four-wave one replaced, kept as its A/B baseline (MD_GEMM_W4=0)
    case 11: return launch_cfg<256, 256, 2, 4, EPI, false, 4, 32, 2>(k, stream);  // alternating wave groups, 2 slices ahead
    case 15: return launch_cfg<256, 256, 2, 4, EPI, false, 4, 32, 3>(k, stream);  // alternating + persistent tile loop
    // decode regime, 64x64 tiles + two DMA-only helper waves (4 waves issue the stream)
    case 16: return k.slices > 1 ? launch_cfg<64, 64, 2, 1, EPI, true, 4, 64, 0, 2>(k, stream)
                                 : launch_cfg<64, 64, 2, 1, EPI, false, 4, 64, 0, 2>(k, stream);
    // round 5: FOUR compute waves (2 x 2 wave tiles of 32 x 32: one MFMA and two fragment reads per wave and K step instead of two and
    // three on two of the four SIMDs) + the two DMA-only helpers, and 128-wide K slices where the layer allows it (K and every
    // K-slice range a multiple of 128).  Same K order per output element as 16 / 10: bit-identical results.
    case 17:
      if (k.K % 128 == 0 && (k.slices <= 1 || (k.K / 64) % (2 * k.slices) == 0))
        return k.slices > 1 ? launch_cfg<64, 64, 2, 2, EPI, true, 4, 128, 0, 2>(k, stream)
                            : launch_cfg<64, 64, 2, 2, EPI, false, 4, 128, 0, 2>(k, stream);
      return k.slices > 1 ? launch_cfg<64, 64, 2, 2, EPI, true, 4, 64, 0, 2>(k, stream)
                          : launch_cfg<64, 64, 2, 2, EPI, false, 4, 64, 0, 2>(k, stream);
    // the two ingredients of 17 on their own (A/B): 18 = four compute waves, 64-wide slices; 19 = two compute waves, 128-wide slices
    case 18: return k.slices > 1 ? launch_cfg<64, 64, 2, 2, EPI, true, 4, 64, 0, 2>(k, stream)
                                 : launch_cfg<64, 64, 2, 2, EPI, false, 4, 64, 0, 2>(k, stream);
    case 19:
      if (k.K % 128 == 0 && (k.slices <= 1 || (k.K / 64) % (2 * k.slices) == 0))
        return k.slices > 1 ? launch_cfg<64, 64, 2, 1, EPI, true, 4, 128, 0, 2>(k, stream)
                            : launch_cfg<64, 64, 2, 1, EPI, false, 4, 128, 0, 2>(k, stream);
      return k.slices > 1 ? launch_cfg<64, 64, 2, 1, EPI, true, 4, 64, 0, 2>(k, stream)
                          : launch_cfg<64, 64, 2, 1, EPI, false, 4, 64, 0, 2>(k, stream);
    // round 6 (probe): the decode-regime config with a 128-row tile -- four compute waves x 32 rows + the two DMA helpers; one
    // weight panel per 128 rows instead of one per 64.  Same K order per output element as 16.
    case 21: return k.slices > 1 ? launch_cfg<128, 64, 4, 1, EPI, true, 4, 64, 0, 2>(k, stream)
                                 : launch_cfg<128, 64, 4, 1, EPI, false, 4, 64, 0, 2>(k, stream);
    default: return launch_cfg<128, 128, 2, 2, EPI>(k, stream);
  }
}

// Decode regime (m <= 64): the layer is a weight stream.  64 x 128 tiles keep the
// activation traffic at half the weight traffic through the per-CU load path; K is
// split over S workgroups per tile so that ~256+ workgroups pull from HBM.  S is a
// function of (n, k) only.
constexpr size_t TICKET_BYTES = 8192;

// decode-regime config: 16 = 64x64 tiles + helper waves (default), 10 = 64x64 without helpers, 3 = 64x128 / 4-deep ring
int decode_cfg() { return knobs().decode_cfg; }
bool decode_is64(int cfg) { return cfg == 10 || cfg == 16 || cfg == 17 || cfg == 18 || cfg == 19; }
int decode_bn() { return decode_is64(decode_cfg()) ? 64 : 128; }
int decode_slab_floats() { return decode_is64(decode_cfg()) ? 64 * 64 : 128 * 2 * 2 * 16; }  // CT * MI * NI * 16 = the tile's outputs

int decode_slices(int n_store, int k_pad) {
  const int DEC_BN = decode_bn();
  const int tiles = (n_store + DEC_BN - 1) / DEC_BN, nk = k_pad / BK;
  if (knobs().decode_slices >= 1) return std::min(knobs().decode_slices, std::max(1, nk));  // experiments
  // measured model (profiles/r01_decode_gemm_sweep.txt): one workgroup saturates its
  // CU's load path (~40 GB/s), the last arriver pays ~1 us per 32 KiB slab, so
  // t ~ 2 us + bytes / (tiles * S * 40 GB/s) + S * 1 us: aim for ~256 workgroups, S <= 8
  // (profiles/r01_decode_gemm_sweep_deep_ring_nt.txt: deeper rings do not help, the per-CU
  // LDS-DMA rate is the limit, so the lever is the number of CUs pulling)
  int s = 1;
  while (s < 8 && tiles * s < 192 && s * 2 <= nk / 2) s *= 2;
  return s;
}

// Tile choice: every CU works through ceil(tiles / 256) tiles, a tile costs its area
// over the config's measured efficiency (256x256 alternating schedule = 1; 256x128
// ~0.78; 128x128 ~0.62 -- profiles/r01_gemm_alternating_sweep*.txt).  A function of the shape
// (M, N, K) only, so a given layer always runs the same kernel -- and every config
// accumulates K in the same order (sequential 16-wide MFMA steps), so results do
// not depend on it.
int pick_tile(int M, int n_store, int K, int policy) {
  if (knobs().tile >= 0) return knobs().tile;  // experiments / tests: force a tile config
  // MD_TILE_PINNED (md_gemm_args.tile_policy, per call): the config is a function of the layer, never of the row count --
  // the 256x256 kernel (16x16x32 MFMAs; the small-shape configs below multiply with 32x32x16: equal to fp32 rounding, not
  // bitwise), so that a sequence gets the same bits alone and in a batch
  if (policy == MD_TILE_PINNED && M > 64) return knobs().w4 ? 20 : 11;
  // MD_TILE_DECODE_TALL: 65 .. 128 rows of a decode step -- the 128 x 64 weight-streaming tile (21), except for the widest layers
  // (lm_head: 800 column panels keep every CU busy with the 128 x 128 two-stage config, measured 57 vs 73 us); every config
  // involved multiplies with 32x32x16 in the same K order as the 64-row decode configs
  if (policy == MD_TILE_DECODE_TALL && M > 64 && M <= 128 && n_store < 16384) return 21;
  // Single-image regime (1458 ViT rows / 735 decoder rows; every config named here multiplies with 32x32x16 in the same K
  // order: the choice never changes a bit).
  //  * at most 128 tiles of 128 x 128 and a long K (proj / fc2 of both towers): half the chip is idle whatever the tile and the
  //    launch runs at the latency of its K loop -- the 64 x 64 tiles with the 4-deep ring and DMA helper waves (the decode-regime
  //    config) quadruple the workgroups (tools/sweep_gemm_b1.py, round 2: text fc2 at 730 rows 81 -> 61 us, ViT fc2 50 -> 30 us).
  //    Round 5 re-swept it: a 128 x 128 tile on a four-stage ring wins the back-to-back sweep (text fc2 63.7 -> 52.0 us) and LOSES
  //    inside a caption, where the layer's weights come from HBM and 96 workgroups pull them through 96 CUs (ViT proj / fc2 23.7 ->
  //    24.9 us, text 42.0 -> 42.5 us per launch: profiles/r05_b1_tile_config_sweep.txt) -- the 64 x 64 config stays.
  //  * up to 512 tiles of 128 x 128 (ViT qkv / fc1, projector fc1): the two-stage 128 x 128 config keeps TWO workgroups per CU, so
  //    all of them are resident at once; the cost model below counts one workgroup per CU and preferred 256 x 128.  Inside a
  //    caption: ViT qkv 25.9 -> 21.5 us, fc1 29.2 -> 24.9 us per launch (round 5).  Wider layers (the decoder's fused qkv|fc1: 672
  //    such tiles) stay with the cost model, i.e. the four-wave kernel.
  if (M > 64) {
    const long t128 = (long)((M + 127) / 128) * ((n_store + 127) / 128);
    if (K >= 1024 && t128 <= 128) return 16;
    // ("small_m_rule": 0 = round 2's rule, 1 = the default below, n > 1 = that many tiles instead of 512 -- A/B)
    const int rule = knobs().small_m_rule;
    if (rule != 0 && t128 <= (rule > 1 ? rule : 512)) return 2;
  }
  const int bm[3] = {256, 256, 128}, bn[3] = {256, 128, 128};
  const double eff[3] = {1.0, 0.70, 0.55};
  int best = 2;
  double best_cost = 1e300;
  for (int c = 0; c < 3; ++c) {
    const long tiles = (long)((M + bm[c] - 1) / bm[c]) * ((n_store + bn[c] - 1) / bn[c]);
    const long per_cu = (tiles + 255) / 256;
    const double cost = (double)per_cu * bm[c] * bn[c] / eff[c];
    if (cost < best_cost) {
      best_cost = cost;
      best = c;
    }
  }
  if (best == 0) best = knobs().w4 ? 20 : 11;
  return best;
}

}  // namespace

namespace {
md_status gemm_dispatch(const md_gemm_args* a, void* stream, const md_rope_fuse* rf);
}
extern "C" md_status md_gemm_bf16(const md_gemm_args* a, void* stream) { return gemm_dispatch(a, stream, nullptr); }
md_status md_gemm_qkv_rope(const md_gemm_args* a, const md_rope_fuse* rf, hipStream_t stream) {
  MD_CHECK_ARG(rf && rf->row_cs && rf->row_kv && rf->kslab && rf->vslab && rf->n_heads > 0 && rf->ctx > 0);
  if (!knobs().rope_fuse) return MD_ERR_UNSUPPORTED;
  return gemm_dispatch(a, (void*)stream, rf);
}
namespace {
md_status gemm_dispatch(const md_gemm_args* a, void* stream, const md_rope_fuse* rf) {
  MD_CHECK_ARG(a && a->a && a->c && a->lin.w);
  MD_CHECK_ARG(a->m > 0 && a->lin.n > 0 && a->lin.k > 0);
  MD_CHECK_ARG(a->lin.k_pad % BK == 0 && a->lin.k_pad >= a->lin.k);
  MD_CHECK_ARG(a->lin.n_pad % 64 == 0 && a->lin.n_pad >= a->lin.n);
  MD_CHECK_ARG(a->store_pad_cols || a->lin.n % 8 == 0);  // rows are stored in 16-byte pieces
  MD_CHECK_ARG(a->gelu_from_col >= 0 && a->gelu_from_col % 8 == 0);
  MD_CHECK_ARG(a->lda >= a->lin.k_pad && a->lda % 8 == 0 && a->ldc % 8 == 0);
  MD_CHECK_ARG(((uintptr_t)a->a & 15) == 0 && ((uintptr_t)a->c & 15) == 0 && ((uintptr_t)a->lin.w & 15) == 0);
  GemmK k;
  k.A = (const bf16_t*)a->a;
  k.W = (const bf16_t*)a->lin.w;
  k.bias = (const bf16_t*)a->lin.b;
  k.R = (const bf16_t*)a->r;
  k.C = (bf16_t*)a->c;
  k.lda = a->lda;
  k.ldw = a->lin.k_pad;
  k.ldc = a->ldc;
  k.ldr = a->ldr;
  k.M = a->m;
  k.n_pad = a->lin.n_pad;
  k.n_store = a->store_pad_cols ? a->lin.n_pad : a->lin.n;
  MD_CHECK_ARG(a->ldc >= k.n_store);
  k.K = a->lin.k_pad;
  k.res_row_mod = a->res_row_mod;
  k.tiles_m = k.tiles_n = 0;
  k.group_m = knobs().group_m > 0 ? knobs().group_m : knobs().group_m < 0 ? md_gemm_auto_group_m(k.n_store) : md_gemm_auto_group_m_w4(k.n_store, k.K);
  k.gelu_from = a->gelu_from_col;
  k.partial = nullptr;
  k.partial_ld = k.partial_slice_stride = 0;
  k.nt = knobs().nt;  // decode regime: non-temporal weight stream (kernel-level +2..10 %, nothing end to end)
  hipStream_t s = (hipStream_t)stream;
  MD_CHECK_ARG(a->tile_policy == MD_TILE_BY_SHAPE || a->tile_policy == MD_TILE_PINNED || a->tile_policy == MD_TILE_DECODE_TALL);
  int tile = pick_tile(k.M, k.n_store, k.K, a->tile_policy);
  k.slices = 1;
  k.slabs = nullptr;
  k.tickets = nullptr;
  const bool forced = knobs().tile >= 0;
  if (tile == 20 && !md_gemm_w4_takes(k, a->epilogue)) {
    if (a->tile_policy != MD_TILE_PINNED) {
      tile = 11;
    } else {
      // MD_TILE_PINNED promises that the tile config is a function of the layer alone; the four-wave kernel's 32-bit offsets
      // bound the ROWS of a launch (M * lda * 2 < 4 GiB, (M + 256) * ldc * 2 < 0xfffff000: ~150 k rows of the 2B fused qkv|fc1
      // layer).  Falling back to the 32x32x16 family here would give a very large batch other bits than the same sequence
      // alone (advisor, round 5): the launch is cut into row blocks the kernel accepts -- rows are independent, so the bits
      // are those of one launch -- and what cannot be cut (a broadcast residual, the RoPE epilogue) is refused loudly.
      if (rf != nullptr || (a->epilogue == MD_EPI_RESIDUAL && a->res_row_mod != 0) || k.K % 64 != 0 ||
          (uint64_t)k.n_pad * (uint64_t)k.ldw * 2 >= (1ull << 32))
        return MD_ERR_UNSUPPORTED;
      uint64_t rows = (1ull << 32) / ((uint64_t)k.lda * 2) - 1;
      rows = std::min<uint64_t>(rows, 0xfffff000ull / ((uint64_t)k.ldc * 2) - 257);
      if (a->epilogue == MD_EPI_RESIDUAL) rows = std::min<uint64_t>(rows, 0xfffff000ull / ((uint64_t)k.ldr * 2) - 257);
      rows = rows / 256 * 256;
      if (rows < 256 || rows >= (uint64_t)a->m) return MD_ERR_UNSUPPORTED;
      for (int64_t r0 = 0; r0 < a->m; r0 += (int64_t)rows) {
        md_gemm_args blk = *a;
        blk.m = (int32_t)std::min<int64_t>((int64_t)rows, a->m - r0);
        blk.a = (const char*)a->a + r0 * a->lda * 2;
        blk.c = (char*)a->c + r0 * a->ldc * 2;
        if (a->r) blk.r = (const char*)a->r + r0 * a->ldr * 2;
        if (blk.m <= 64) blk.splitk_ws = nullptr;  // a short last block: no in-launch split-K (another K association)
        const md_status st = gemm_dispatch(&blk, stream, nullptr);
        if (st != MD_OK) return st;
      }
      return MD_OK;
    }
  }
  if (rf != nullptr) {
    // RoPE + KV write in the epilogue: four-wave kernel only, [q | k | v] sections of n_heads x 64 columns ending where the
    // GELU columns start, slab offsets in 32 bits
    const int D = rf->n_heads * 64;
    if (tile != 20 || a->epilogue != MD_EPI_GELU || a->gelu_from_col != 3 * D || D % 128 != 0 || a->m <= 64 ||
        rf->slab_bytes >= 0xfffff000ull)
      return MD_ERR_UNSUPPORTED;
    k.rope_cs = rf->row_cs;
    k.rope_kv = rf->row_kv;
    k.kslab = (bf16_t*)rf->kslab;
    k.vslab = (bf16_t*)rf->vslab;
    k.slab_bytes = (uint32_t)rf->slab_bytes;
    k.rope_d = D;
    k.rope_ctx = rf->ctx;
  }
  if (tile == 11 && a->epilogue != MD_EPI_RESIDUAL && !forced) {
    // eight-wave baseline: bias / GELU layers with more tiles than CUs run its persistent tile loop
    const long tiles = (long)((k.M + 255) / 256) * ((k.n_store + 255) / 256);
    if (knobs().persist && tiles > 256) tile = 15;
  }
  const bool tall = (tile == 21 && a->m > 64 && a->m <= 128 && !forced);  // MD_TILE_DECODE_TALL: the 128-row weight-streaming tile
  if ((a->m <= 64 || tall) && !forced) {
    if (!tall) tile = decode_cfg();
    // in-launch split-K exactly where the 64-row regime splits (a function of the layer shape): the same slices, the same
    // slab summation order -- a row gets the same bits in a 128-row launch as in a 64-row one
    const int sl = decode_slices(k.n_store, k.K);
    const size_t tiles = (k.n_store + decode_bn() - 1) / decode_bn();
    const size_t need = TICKET_BYTES + tiles * sl * (tall ? 128 * 64 : decode_slab_floats()) * sizeof(float);
    if (sl > 1 && a->splitk_ws != nullptr && a->splitk_ws_bytes >= need && tiles * 4 <= TICKET_BYTES) {
      k.slices = sl;
      k.tickets = (unsigned*)a->splitk_ws;
      k.slabs = (float*)((char*)a->splitk_ws + TICKET_BYTES);
    }
  }
  if (a->epilogue == MD_EPI_RESIDUAL)
    MD_CHECK_ARG(a->r != nullptr && a->ldr % 8 == 0 && ((uintptr_t)a->r & 15) == 0);
  ProfRec rec;
  const bool prof = g_prof_on;
  if (prof) {
    if (hipEventCreate(&rec.start) != hipSuccess || hipEventCreate(&rec.stop) != hipSuccess) return MD_ERR_LAUNCH;
    rec.kind = (a->m <= 64 || (a->tile_policy == MD_TILE_DECODE_TALL && a->m <= 128)) ? 1 : 0;   // a weight stream
    rec.work = rec.kind ? 2.0 * (double)a->lin.n * (double)a->lin.k           // bf16 weight bytes, logical n, k
                        : 2.0 * a->m * (double)a->lin.n * (double)a->lin.k;  // algorithmic flops
    rec.rd = 2.0 * ((double)a->m * a->lin.k_pad + (double)a->lin.n_pad * a->lin.k_pad +
                    (a->epilogue == MD_EPI_RESIDUAL ? (double)(a->res_row_mod ? a->res_row_mod : a->m) * k.n_store : 0.0));
    rec.wr = 2.0 * (double)a->m * k.n_store;
    (void)hipEventRecord(rec.start, s);
  }
  md_status st;
  switch (a->epilogue) {
    case MD_EPI_BIAS: st = launch_epi<MD_EPI_BIAS>(k, tile, s); break;
    case MD_EPI_GELU: st = rf ? md_gemm_w4_launch(k, MD_EPI_QKV_ROPE, s) : launch_epi<MD_EPI_GELU>(k, tile, s); break;
    case MD_EPI_RESIDUAL: st = launch_epi<MD_EPI_RESIDUAL>(k, tile, s); break;
    default: st = MD_ERR_INVALID_ARG;
  }
  if (prof) {
    (void)hipEventRecord(rec.stop, s);
    g_prof.push_back(rec);
  }
  return st;
}
}  // namespace

// Launch-boundary split-K (decode regime, m <= 64): S = md_gemm_partial_slices() workgroups per
// 64-column tile, each over a contiguous range of K, store fp32 partial products
// partial[s][row][ld_partial]; no bias, no tickets, no fences -- the kernel boundary publishes
// them and md_reduce_residual_layernorm sums the slices in index order.
extern "C" int32_t md_gemm_partial_slices(const md_linear* lin) {
  if (!lin || lin->n <= 0 || lin->k_pad <= 0) return 0;
  const int tiles = (lin->n + 63) / 64, nk = lin->k_pad / BK;
  int s = 256 / std::max(1, tiles);
  s = std::max(1, std::min(s, std::min(8, nk)));
  return s;
}

namespace {
md_status fill_partial(GemmK& k, const void* a, int64_t lda, const md_linear* lin, int32_t m, float* partial,
                       int64_t ld_partial, int64_t slice_stride) {
  MD_CHECK_ARG(a && lin && lin->w && partial && m > 0 && m <= 128);  // (65 .. 128 rows: the 128 x 64 tile, round 6)
  MD_CHECK_ARG(lin->k_pad % BK == 0 && lin->k_pad >= lin->k && lin->n % 8 == 0 && lin->n_pad % 64 == 0);
  MD_CHECK_ARG(lda >= lin->k_pad && lda % 8 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)lin->w & 15) == 0);
  MD_CHECK_ARG(ld_partial >= lin->n && ld_partial % 4 == 0 && ((uintptr_t)partial & 15) == 0);
  MD_CHECK_ARG(slice_stride >= (int64_t)m * ld_partial && slice_stride % 4 == 0);
  k.A = (const bf16_t*)a;
  k.W = (const bf16_t*)lin->w;
  k.bias = nullptr;
  k.R = nullptr;
  k.C = nullptr;
  k.lda = lda;
  k.ldw = lin->k_pad;
  k.ldc = k.ldr = 0;
  k.M = m;
  k.n_pad = lin->n_pad;
  k.n_store = lin->n;
  k.K = lin->k_pad;
  k.res_row_mod = 0;
  k.tiles_m = 1;
  k.tiles_n = (lin->n + 63) / 64;
  k.group_m = 8;
  k.gelu_from = 0;
  k.nt = knobs().nt;
  k.slices = md_gemm_partial_slices(lin);
  k.slabs = nullptr;
  k.tickets = nullptr;
  k.partial = partial;
  k.partial_ld = ld_partial;
  k.partial_slice_stride = slice_stride;
  return MD_OK;
}

struct ProfScope {  // HIP-event bracket of one launch for md_profile_gemm (kind 1: weight bytes streamed)
  ProfRec rec;
  bool on;
  hipStream_t s;
  md_status begin(double work, hipStream_t stream) {
    on = g_prof_on;
    s = stream;
    if (!on) return MD_OK;
    if (hipEventCreate(&rec.start) != hipSuccess || hipEventCreate(&rec.stop) != hipSuccess) return MD_ERR_LAUNCH;
    rec.kind = 1;
    rec.work = work;
    rec.rd = work;
    rec.wr = 0;
    (void)hipEventRecord(rec.start, s);
    return MD_OK;
  }
  void end() {
    if (!on) return;
    (void)hipEventRecord(rec.stop, s);
    g_prof.push_back(rec);
  }
};
}  // namespace

extern "C" md_status md_gemm_partial_f32(const void* a, int64_t lda, const md_linear* lin, int32_t m,
                                         float* partial, int64_t ld_partial, int64_t slice_stride,
                                         void* stream) {
  GemmK k;
  const md_status chk = fill_partial(k, a, lda, lin, m, partial, ld_partial, slice_stride);
  if (chk != MD_OK) return chk;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof;
  if (prof.begin(2.0 * (double)lin->n * (double)lin->k, s) != MD_OK) return MD_ERR_LAUNCH;
  const int cfg = decode_cfg();
  const md_status st = m > 64 ? launch_cfg<128, 64, 4, 1, MD_EPI_BIAS, true, 4, 64, 0, 2>(k, s)
                       : (cfg == 17 || cfg == 18) ? launch_cfg<64, 64, 2, 2, MD_EPI_BIAS, true, 4, 64, 0, 2>(k, s)
                       : (cfg == 16 || cfg == 19) ? launch_cfg<64, 64, 2, 1, MD_EPI_BIAS, true, 4, 64, 0, 2>(k, s)
                                                  : launch_cfg<64, 64, 2, 1, MD_EPI_BIAS, true, 4>(k, s);
  prof.end();
  return st;
}

// Both partial-product GEMMs of a decode block (proj over the attention output, fc2 over
// gelu(fc1)) in ONE launch: they are independent, and a launch costs as much as the stream.
extern "C" md_status md_gemm_partial_f32_pair(const void* a0, int64_t lda0, const md_linear* lin0, float* partial0,
                                              const void* a1, int64_t lda1, const md_linear* lin1, float* partial1,
                                              int32_t m, int64_t ld_partial, int64_t slice_stride, void* stream) {
  GemmPair pair;
  md_status chk = fill_partial(pair.g[0], a0, lda0, lin0, m, partial0, ld_partial, slice_stride);
  if (chk != MD_OK) return chk;
  chk = fill_partial(pair.g[1], a1, lda1, lin1, m, partial1, ld_partial, slice_stride);
  if (chk != MD_OK) return chk;
  hipStream_t s = (hipStream_t)stream;
  constexpr int lds = 4 * (64 + 64) * 64 * 2;
  // (64-wide slices here whatever the config: 64 KiB of ring = TWO workgroups per CU, and the pair makes 512)
  const int cfg = decode_cfg();
  const bool four = cfg == 17 || cfg == 18, helpers = four || cfg == 16 || cfg == 19;
  // 65 .. 128 rows (round 6): one 128 x 64 tile per weight panel -- four compute waves x 32 rows + the two DMA helpers, 96 KiB
  // of ring (one workgroup per CU); the same K slices and the same K order per output element as the 64-row configs
  const bool tall = m > 64;
  const int NT = tall ? 384 : four ? 384 : helpers ? 256 : 128;
  static const int tall_stages = [] { const char* e = getenv("MD_TALL_PAIR_STAGES"); return e ? atoi(e) : 3; }();   // 3 stages = 72 KiB of ring: TWO workgroups per CU, all 512 of the pair resident at once (18.6 us against 20.3 with 4 stages; MD_TALL_PAIR_STAGES=4: A/B)
  const int lds_used = tall ? tall_stages * (128 + 64) * 64 * 2 : lds;
  auto fn = tall ? (tall_stages == 3 ? gemm_pair_kernel<128, 64, 4, 1, MD_EPI_BIAS, true, 3, 64, 0, 2>
                                     : gemm_pair_kernel<128, 64, 4, 1, MD_EPI_BIAS, true, 4, 64, 0, 2>)
            : four ? gemm_pair_kernel<64, 64, 2, 2, MD_EPI_BIAS, true, 4, 64, 0, 2>
            : helpers ? gemm_pair_kernel<64, 64, 2, 1, MD_EPI_BIAS, true, 4, 64, 0, 2>
                      : gemm_pair_kernel<64, 64, 2, 1, MD_EPI_BIAS, true, 4>;
  MD_TRY(md_ensure_dynamic_lds((const void*)fn, lds_used));
  ProfScope prof;
  if (prof.begin(2.0 * ((double)lin0->n * lin0->k + (double)lin1->n * lin1->k), s) != MD_OK) return MD_ERR_LAUNCH;
  const int gx = std::max(pair.g[0].tiles_m * pair.g[0].tiles_n, pair.g[1].tiles_m * pair.g[1].tiles_n);
  const int gy = std::max(pair.g[0].slices, pair.g[1].slices);
  hipLaunchKernelGGL(fn, dim3(gx, gy, 2), dim3(NT), lds_used, s, pair);
  prof.end();
  return md_launch_status();
}

extern "C" size_t md_gemm_workspace_bytes(const md_linear* lin, int32_t m, int32_t store_pad_cols) {
  if (!lin || m > 128) return 0;   // (65 .. 128 rows: what a MD_TILE_DECODE_TALL launch needs; other policies ignore the workspace there)
  const int n_store = store_pad_cols ? lin->n_pad : lin->n;
  const int sl = decode_slices(n_store, lin->k_pad);
  if (sl == 1) return 0;
  const size_t tiles = (n_store + decode_bn() - 1) / decode_bn();
  return TICKET_BYTES + tiles * sl * (m > 64 ? 128 * 64 : decode_slab_floats()) * sizeof(float);
}

bool md_gemm_knob_rope_fuse() { return knobs().rope_fuse != 0; }

// Row panels per group of the tile order (a group = group_m row panels x all column panels, rows fastest; an XCD's 32
// concurrent workgroups walk it contiguously).  Measured on the model's shapes (profiles/r03_gemm_tile_order_group_m.txt):
// layers of at most eight 256-wide column panels (N <= 2048: proj, fc2) run 2-3.5 % faster when an XCD covers ALL column
// panels of a few row panels at a time (group_m = 1: every activation panel crosses the fabric once), wide layers want a
// squarer block (4; 8 was the value until round 3 and is 2-4 % slower on the long-K narrow layers, never faster).
int md_gemm_auto_group_m(int n_store) { return (n_store + 255) / 256 <= 8 ? 1 : 4; }
// Round 4, re-swept on the LDS-DMA / 16x16x32 four-wave kernel (profiles/r04_gemm_tile_order_group_m.txt): the narrowest layers
// (N = 1152: 4.5 column panels) want 2 row panels per group, 8-panel layers 8 when K is short (text proj) and 1 when it is
// long (fc2: the weight panel of a long-K layer is what an XCD's L2 should keep), 9-17 panels (ViT qkv / fc1) 8, wider ones 4.
// md_gemm_set_tuning("group_m", -1) selects round 3's rule above.
int md_gemm_auto_group_m_w4(int n_store, int k) {
  const int panels = (n_store + 255) / 256;
  if (panels <= 5) return 2;
  if (panels <= 8) return k <= 4096 ? 8 : 1;
  if (panels <= 17) return 8;
  return 4;
}

extern "C" md_status md_gemm_set_tuning(const char* key, int32_t value) {
  MD_CHECK_ARG(key != nullptr);
  Knobs& k = knobs();
  const std::string s(key);
  if (s == "tile") k.tile = value;
  else if (s == "group_m") k.group_m = std::max(-1, (int)value);
  else if (s == "w4") k.w4 = value;
  else if (s == "persist") k.persist = value;
  else if (s == "decode_nt") k.nt = value;
  else if (s == "decode_cfg") k.decode_cfg = value;
  else if (s == "decode_slices") k.decode_slices = value;
  else if (s == "w4_variant") md_gemm_w4_set_variant(value);
  else if (s == "rope_fuse") k.rope_fuse = value;
  else if (s == "small_m_rule") k.small_m_rule = value;
  else if (s == "attn_skip_dead") md_attention_set_skip_dead(value);
  else if (s == "w4_grid") md_gemm_w4_set_grid(value);
  else if (s == "w4_dbg_lo") md_gemm_w4_set_debug(0, (uint32_t)value);
  else if (s == "w4_dbg_hi") md_gemm_w4_set_debug(1, (uint32_t)value);
  else return MD_ERR_INVALID_ARG;
  return MD_OK;
}

extern "C" void md_profile_gemm(int32_t enable) {
  for (auto& r : g_prof) {
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  g_prof.clear();
  g_prof_on = enable != 0;
}

extern "C" md_status md_profile_gemm_bytes(int32_t kind, double* read_bytes, double* written_bytes) {
  MD_CHECK_ARG(read_bytes && written_bytes);
  double r = 0, w = 0;
  for (auto& rec : g_prof)
    if (rec.kind == kind) {
      r += rec.rd;
      w += rec.wr;
    }
  *read_bytes = r;
  *written_bytes = w;
  return MD_OK;
}

extern "C" md_status md_profile_gemm_read(int32_t kind, double* work, double* ms, int64_t* launches) {
  MD_CHECK_ARG(work && ms && launches);
  double f = 0, t = 0;
  int64_t n = 0;
  for (auto& r : g_prof) {
    if (r.kind != kind) continue;
    float e = 0;
    if (hipEventSynchronize(r.stop) != hipSuccess || hipEventElapsedTime(&e, r.start, r.stop) != hipSuccess)
      return MD_ERR_LAUNCH;
    f += r.work;
    t += e;
    ++n;
  }
  *work = f;
  *ms = t;
  *launches = n;
  return MD_OK;
}
