// Implicit-GEMM tile routine shared by the per-launch kernel (kernels_gemm.hip) and the persistent per-XCD schedule
// (persist.hip).  See kernels_gemm.hip for the design notes.
#pragma once
#ifndef DV_GNX_STATS_FIRST
// 1: a fragment's 32x16 block statistics are computed and published before its result stores are issued; 0 (default): behind them.
// Round 5, same-box A/B (profiles/r05_ab_stats_first.txt): 62.29 k vs 62.42 k mel-frames/s - the publishing workgroup is rarely the
// one the launch waits for; not adopted
#define DV_GNX_STATS_FIRST 0
#endif
#ifndef DV_GEMM_EXP
// development knob (trace experiments on the plain tile's k-loop, WRONG results): 1 = B fragments read from LDS once, not per
// k-tile; 2 = 1 + no B DMA; 3 = 2 + A fragments read once; 4 = no DMA inside the loop at all.  0 in every shipped build
#define DV_GEMM_EXP 0
#endif
#include "dv_common.h"
#include "dv_device.h"
#include "gnx_device.h"

#include <type_traits>

#if defined(DV_GEMM_TRACE) && defined(DV_GEMM_TRACE_OWNER)
#define DV_GEMM_TRACING 1
// development build only (make trace): per-workgroup s_memtime stamps of the kernel's phases
// (DV_TR_W slots per workgroup; p.trace: launch_gemm stamps every launch, or only the one chosen by dv_debug_gemm_trace_select)
#define DV_TR_W 32
__device__ unsigned long long g_gemm_trace[8192 * DV_TR_W];
#define DV_TRACE(i) do { if (p.trace && threadIdx.x == 0 && blockIdx.x < 8192) g_gemm_trace[blockIdx.x * DV_TR_W + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define DV_TRACE_P(i, first) do { if (p.trace && (int)threadIdx.x == (first) && blockIdx.x < 8192) g_gemm_trace[blockIdx.x * DV_TR_W + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int dv_debug_gemm_trace(unsigned long long* host, int n_wg) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_trace), (size_t)n_wg * DV_TR_W * sizeof(unsigned long long));
}
extern "C" int dv_debug_gemm_trace_clear() {
  void* d = nullptr;
  hipError_t e = hipGetSymbolAddress(&d, HIP_SYMBOL(g_gemm_trace));
  return (int)(e != hipSuccess ? e : hipMemset(d, 0, sizeof(g_gemm_trace)));
}
#else
#define DV_TRACE(i) do {} while (0)
#define DV_TRACE_P(i, first) do {} while (0)
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) { return dv_cvt_pk_bf16(lo, hi); }
// KS = 2 doubles the waves of a workgroup (two per SIMD): both groups stage every k-tile together and each
// multiplies half of its 16-deep k-steps, so one wave's MFMAs overlap the other's address math and DMA issue;
// the two partial accumulators are added through LDS before the epilogue.
// SC1: every load of MUTABLE data (A planes, residual, LayerNorm partials) bypasses this CU's L1 (`sc1`: served by the
// XCD's L2).  Needed when producer and consumer run inside ONE launch on different CUs (persist.hip); weights and
// biases are never written on the GPU and keep the default policy.
// Split-K over two launches (p.sk_mode; long-K GEMMs with too few tiles to occupy 256 CUs): mode 1 runs slice ksel of
// p.sk_split equal k-tile ranges and dumps the raw accumulators lane-linearly (coalesced 16-byte stores) to p.sk_buf;
// mode 2 (same tile shape) starts from the sum of the slices, skips the k-loop and runs the ordinary epilogue.
// Mode 3 (p.sk_ticket given, two slices) does both in ONE launch: each workgroup dumps its half with write-through
// stores and takes a ticket from the tile's agent-scope counter; the first to arrive leaves, the second adds the
// partner's dump (system-scope loads, after the atomic) and runs the epilogue.  No workgroup ever waits for another.
// Measured (docs/HISTORY.md §4): worth ~25 % on the K >= 1536 GEMMs of the 128-frame level, ~1 % of a forward.
template <int BM, int BN, int BK, int WM, int WN, int NSPLIT, int KS, bool SC1>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, const int m0, const int n0, char* smem, const int ksel = 0) {
  constexpr int FM = BM / (WM * 32), FN = BN / (WN * 32);
  constexpr int NWQ = WM * WN;                       // waves per k-group (1, 2 or 4)
  constexpr int NWV = NWQ * KS;                      // waves per workgroup
  constexpr bool SPLIT = NSPLIT == 3;
  constexpr int NPL = SPLIT ? 2 : 1;                 // planes per operand
  constexpr int ROWB = BK * 2;                       // LDS row pitch (bytes), unpadded
  constexpr int CPR = ROWB / 16;                     // 16-byte chunks per row: 4 (BK=32) or 8 (BK=64)
  constexpr int RPI = 64 / CPR;                      // rows per wave-instruction
  constexpr int A_PL = BM * ROWB, B_PL = BN * ROWB;  // bytes per plane tile in the ring
  constexpr int STAGE = (A_PL + B_PL) * NPL;
#ifndef DV_NSTAGE_64
#define DV_NSTAGE_64 4
#endif
  // LDS ring depth: NSTAGE-1 tiles in flight (DV_NSTAGE_64: experiment knob for the 64x64 tiles' occupancy)
  constexpr int NSTAGE = (BM == 64 && BN == 64 && !SC1) ? DV_NSTAGE_64 : ((4 * STAGE <= 160 * 1024) ? 4 : 3);
  constexpr int A_IPW = BM / RPI / NWV, B_IPW = BN / RPI / NWV;   // DMA instructions per wave per plane
  constexpr int A_IPW1 = A_IPW, B_IPW1 = B_IPW;
  constexpr int LPT = (A_IPW + B_IPW) * NPL;         // DMA instructions per thread per k-tile
  constexpr int LPT_W = DV_GEMM_EXP >= 2 ? A_IPW * NPL : LPT;   // ... that the counted waits see (experiments)
  static_assert(BM % (RPI * NWV) == 0 && BN % (RPI * NWV) == 0, "tile rows must split over the waves");
  DV_TRACE(0);
#ifdef DV_GEMM_TRACING
  if (p.trace && threadIdx.x == 0 && blockIdx.x < 8192) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    g_gemm_trace[blockIdx.x * DV_TR_W + 6] = ((unsigned long long)xcc << 32) | hw;
    g_gemm_trace[blockIdx.x * DV_TR_W + 7] = wall_clock64();
  }
#endif
  // touch one field of every 64-byte line of the argument block up front: the scalar loads go out together and
  // miss once in parallel, instead of one dependent miss per line as the prologue reaches each field
  asm volatile("" ::"s"(p.seg[0].a0_hi), "s"(p.seg[1].a0_hi), "s"(p.seg[1].pad), "s"(p.T_in), "s"(p.w_hi), "s"(p.M),
               "s"(p.res), "s"(p.out_hi), "s"(p.zero_page), "s"(p.ln_u));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kgrp = wave / NWQ, wq = wave % NWQ;
  const int wm = wq / WN, wn = wq % WN;
  const unsigned smem_base = (unsigned)(size_t)smem;   // LDS byte address of the ring

#ifdef DV_GEMM_TRACING
  if (m0 + n0 >= 0) DV_TRACE(8);    // first kernel-argument dependent value is available
#endif

  // ---- per-lane DMA geometry: lane -> (row within the instruction's RPI rows, LDS chunk slot) ----
  const int l_row = lane / CPR, l_slot = lane % CPR;
  auto swz = [](int row) { return CPR == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

  // row m -> utterance m / T_out without a division (GemmParams tout_magic: exact for every row of the launch)
  auto utt_of = [&](int m) { return p.tout_magic ? (int)__umulhi((unsigned)m, p.tout_magic) : m; };
  int arow_b[A_IPW1], arow_t[A_IPW1], a_chunk[A_IPW1];
  unsigned arow_ok = 0;
#pragma unroll
  for (int q = 0; q < A_IPW; ++q) {
    const int r = (q * NWV + wave) * RPI + l_row;     // row within the A tile
    int m = m0 + r;
    bool ok = m < p.M;
    m = ok ? m : 0;
    arow_b[q] = utt_of(m);
    arow_t[q] = m - arow_b[q] * p.T_out;
    ok = ok && arow_t[q] < p.Tv_out;                  // (padding rows of a padded row space: zeros in, zeros out)
    arow_ok |= (ok ? 1u : 0u) << q;
    a_chunk[q] = l_slot ^ swz(r);                     // source chunk that lands in this lane's slot
  }
  // (plain tiles: 32-bit per-lane byte offset beside a wave-uniform base that moves with the k-tile - the DMA is then
  // "scalar base + vector offset", no 64-bit vector arithmetic per instruction; the weight planes are < 4 GiB)
  size_t b_off[B_IPW1];
  unsigned b_off32[B_IPW1];
#pragma unroll
  for (int q = 0; q < B_IPW; ++q) {
    const int r = (q * NWV + wave) * RPI + l_row;
    b_off[q] = ((size_t)(n0 + r) * p.Kp + (l_slot ^ swz(r)) * 8) * 2;   // byte offset at kt = 0
    b_off32[q] = (unsigned)b_off[q];
  }
  unsigned b_kbase = 0;                              // byte offset of this range's first k-tile inside a weight row

  // Source of the k-tile being issued, kept in wave-uniform registers and advanced incrementally: the segment
  // descriptor (kernel-argument memory) is only re-read when a source tensor / tap is exhausted, never on the
  // per-tile path (a scalar load there sits on every wave's critical path right after the barrier).
  const int total_kt = p.seg[0].nkt + (p.nseg > 1 ? p.seg[1].nkt : 0);
  int kt0 = 0, nk = total_kt;                        // this launch's k-tiles: [kt0, kt0 + nk)
  if (p.sk_mode == 1 || p.sk_mode == 3) { kt0 = ksel * total_kt / p.sk_split; nk = (ksel + 1) * total_kt / p.sk_split - kt0; }
  if (p.sk_mode == 2) nk = 0;
  int ld_seg = 0, ld_tap = 0, ld_half = 0;
  const bf16_t* cur_hi; const bf16_t* cur_lo;
  int cur_ld, cur_col, cur_toff;
  auto enter = [&]() {
    // (readfirstlane: the index is wave-uniform by construction; where hipcc cannot prove it, a vector-indexed p.seg[] sends the
    // whole argument block through scratch memory)
    const GemmSeg& sg = p.seg[__builtin_amdgcn_readfirstlane(ld_seg)];
    cur_hi = ld_half ? sg.a1_hi : sg.a0_hi;
    cur_lo = ld_half ? sg.a1_lo : sg.a0_lo;
    cur_ld = ld_half ? sg.c1 : sg.c0;
    cur_col = 0;
    cur_toff = ld_tap - sg.pad;
  };
  enter();
  DV_TRACE(9);     // row geometry done

  // One k-tile's DMA = LPT wave-instructions per thread ("units"): A rows (hi, lo plane) then B rows (hi, lo).
  // prep_a() forms the A source addresses of the tile being issued; issue_unit() sends one unit; advance()
  // moves the source state to the next k-tile.  The main loop spreads the units between its MFMA groups.
  const void* asrc[A_IPW1 * NPL];
  // Inside one run (same source tensor, same tap) the next k-tile's rows are the same rows BK channels on: the pointers just
  // move by BK * 2 bytes - two vector instructions per pointer instead of the ~30 of the full row arithmetic below, which
  // runs only when a run begins (a_fresh).  [The k-loop is bound by instruction ISSUE - a SIMD issues one instruction of a
  // wave every four cycles, two waves share it: ~95 instructions per k-tile per wave were ~800 of the ~1050 cycles per
  // k-tile; profiles/r03_gemm_kloop_ablation_*.txt.]  Lanes that read the zero page (conv padding, rows >= M) move along
  // inside it: it is DV_ZERO_PAGE_BYTES long, longer than the widest row.
  bool a_fresh = true;
  auto prep_a_next = [&]() {
#pragma unroll
    for (int u = 0; u < A_IPW * NPL; ++u) asrc[u] = reinterpret_cast<const char*>(asrc[u]) + BK * 2;
  };
  auto prep_a = [&]() {
#pragma unroll
    for (int q = 0; q < A_IPW; ++q) {
      const int ts = arow_t[q] * p.stride + cur_toff;
      const bool ok = ((arow_ok >> q) & 1u) && ts >= 0 && ts < p.T_virt;
      int st = ts;
      st = p.up_mode == UP_X2 ? (ts >> 1) : st;
      st = p.up_mode == UP_SIZE ? min((int)floorf((float)ts * p.up_scale), p.Tv_in - 1) : st;
      const size_t e = ((size_t)arow_b[q] * p.T_in + st) * cur_ld + cur_col + a_chunk[q] * 8;
      // conv zero padding / rows >= M read the zero page instead (DV_ZERO_PAGE_BYTES: the lane then walks the row's k-tiles inside it)
      asrc[q * NPL] = ok ? (const void*)(cur_hi + e) : (const void*)p.zero_page;
      if (SPLIT) asrc[q * NPL + 1] = ok ? (const void*)(cur_lo + e) : (const void*)p.zero_page;
    }
  };
  auto issue_unit = [&](int kt, int u) {
    const unsigned st_base = smem_base + (unsigned)((kt % NSTAGE) * STAGE);
    if (u < A_IPW * NPL) {
      const int q = u / NPL, pl = u % NPL;
      if (SC1) glds16_sc1(asrc[u], st_base + (unsigned)(((q * NWV + wave) * RPI) * ROWB + pl * A_PL));
      else glds16(asrc[u], st_base + (unsigned)(((q * NWV + wave) * RPI) * ROWB + pl * A_PL));
    } else {
      if (DV_GEMM_EXP >= 2) return;
      const int q = (u - A_IPW * NPL) / NPL, pl = (u - A_IPW * NPL) % NPL;
      const unsigned dst = st_base + (unsigned)(NPL * A_PL + ((q * NWV + wave) * RPI) * ROWB + pl * B_PL);
      glds16_s(reinterpret_cast<const char*>(pl ? p.w_lo : p.w_hi) + (b_kbase + (unsigned)kt * (BK * 2)), b_off32[q], dst);
    }
  };
  auto advance = [&]() {
    cur_col += BK;
    a_fresh = cur_col == cur_ld;
    if (cur_col == cur_ld) {     // wave-uniform, once per (source tensor, tap)
      const GemmSeg& sg = p.seg[__builtin_amdgcn_readfirstlane(ld_seg)];
      if (ld_half == 0 && sg.c1 > 0) ld_half = 1;
      else {
        ld_half = 0;
        if (++ld_tap == sg.taps) { ld_tap = 0; ++ld_seg; }
      }
      if (ld_seg < p.nseg) enter();
    }
  };
  auto issue = [&](int kt) {     // whole tile at once (prologue)
    if (a_fresh) prep_a();
    else prep_a_next();
#pragma unroll
    for (int u = 0; u < LPT; ++u) issue_unit(kt, u);
    advance();
  };
  if (kt0 > 0) {                                     // second k-half: move the source state to its first tile
    for (int t = 0; t < kt0; ++t) advance();
#pragma unroll
    for (int q = 0; q < B_IPW; ++q) b_off[q] += (size_t)kt0 * (BK * 2);
    b_kbase = (unsigned)kt0 * (BK * 2);
    a_fresh = true;                                  // (the range may begin in the middle of a run: full row arithmetic first)
  }

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  constexpr bool SPLIT_EPI = KS == 2 && FM % 2 == 0;
#ifndef DV_T1_COLSPLIT
#define DV_T1_COLSPLIT 1
#endif
  // ... and tiles with ONE row fragment and two column fragments per wave (128x64: four row waves x two k-groups): k-group g keeps
  // column fragment j = g.  [Until round 5 the second k-group of such a tile handed both fragments over and left: the epilogue of the
  // two most expensive launches of the forward - the up-path resamplers, 65 k and 53 k cycles - ran on four waves, one per SIMD:
  // statistics 6.8 k + in-launch GroupNorm pass 10.7 k cycles against 0.7 k + 1.5 k in the 64x64 tiles, r05_gemm_per_launch_trace.txt.]
  // Wave-uniform run-time flag: GEGLU needs both fragments of a 64-column block in one wave, the split-K forms keep their layout.
  const bool own_col = DV_T1_COLSPLIT && !SC1 && KS == 2 && !SPLIT_EPI && FN % 2 == 0 && p.epi != EPI_GEGLU && p.sk_mode == 0;
  auto own = [&](int i, int j) { return SPLIT_EPI ? ((i & 1) == kgrp) : (own_col ? ((j & 1) == kgrp) : (kgrp == 0)); };
  const int l31 = lane & 31, lh = lane >> 5;
  // split-K dump: [slice][tile][fragment][4 column groups][64 * NWQ lanes] float4
  const size_t sk_tile = ((size_t)(m0 / BM) * ((p.N + BN - 1) / BN) + n0 / BN) * (FM * FN * 4) * (64 * NWQ);
  const size_t sk_slice = (size_t)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * (FM * FN * 4) * (64 * NWQ);
  if (p.sk_mode == 2) {                              // accumulators = sum of the k-slices' dumps
    const float4* s0 = reinterpret_cast<const float4*>(p.sk_buf) + sk_tile + wq * 64 + lane;
    for (int sl = 0; sl < p.sk_split; ++sl, s0 += sk_slice) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        if (SPLIT_EPI ? ((i & 1) == kgrp) : (kgrp == 0)) {
#pragma unroll
          for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 a = s0[(size_t)((i * FN + j) * 4 + g) * (64 * NWQ)];
              acc[i][j][4 * g] += a.x; acc[i][j][4 * g + 1] += a.y; acc[i][j][4 * g + 2] += a.z; acc[i][j][4 * g + 3] += a.w;
            }
        }
      }
    }
  }

  // One k-tile of work for this wave: all operand fragments are read first, then the MFMAs go out in groups of
  // FM*FN (one product term of one 16-deep k-step: consecutive MFMAs write different accumulators) and, when
  // ISSUE, the DMA units of tile kt+NSTAGE-1 are spread between the groups: a wave blocked in the (slow, 64 B/clk
  // per CU) vector-memory issue then has MFMAs in flight instead of serialising a DMA phase after an MFMA phase.
  // Weights are the FIRST MFMA operand: the accumulator is the TRANSPOSED tile, lane = output row m,
  // registers = 16 output columns in runs of 4 -> 16-byte epilogue loads / stores per lane.
  constexpr int NKS = BK / 16 / KS;                  // 16-deep k-steps per wave per k-tile
  constexpr int NTERM = SPLIT ? 3 : 1;
  constexpr int NCH = NKS * NTERM;                   // MFMA groups per k-tile
  bf16x8 ahK[NKS][FM], alK[NKS][FM], bhK[NKS][FN], blK[NKS][FN];   // (DV_GEMM_EXP: fragments read once)
  bool exp_first = true;
  auto step = [&](int kt, auto issue_tag) {
    constexpr bool ISSUE = decltype(issue_tag)::value && DV_GEMM_EXP != 4;
    const char* base = smem + (kt % NSTAGE) * STAGE;
    const char* a_hi = base;
    const char* a_lo = base + A_PL;
    const char* b_hi = base + NPL * A_PL;
    const char* b_lo = b_hi + B_PL;
    bf16x8 ah[NKS][FM], al[NKS][FM], bh[NKS][FN], bl[NKS][FN];
#pragma unroll
    for (int ks0 = 0; ks0 < NKS; ++ks0) {
      const int chunk = (kgrp * NKS + ks0) * 2 + lh;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = (wm * FM + i) * 32 + l31;
        const int off = row * ROWB + ((chunk ^ swz(row)) << 4), off_lo = off;
        if (DV_GEMM_EXP == 3) {
          if (exp_first) { ahK[ks0][i] = *reinterpret_cast<const bf16x8*>(a_hi + off); if (SPLIT) alK[ks0][i] = *reinterpret_cast<const bf16x8*>(a_lo + off_lo); }
          ah[ks0][i] = ahK[ks0][i]; if (SPLIT) al[ks0][i] = alK[ks0][i];
        } else {
        ah[ks0][i] = *reinterpret_cast<const bf16x8*>(a_hi + off);
        if (SPLIT) al[ks0][i] = *reinterpret_cast<const bf16x8*>(a_lo + off_lo);
        }
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = (wn * FN + j) * 32 + l31;
        const int off = row * ROWB + ((chunk ^ swz(row)) << 4);
        if (DV_GEMM_EXP >= 1 && DV_GEMM_EXP <= 3) {
          if (exp_first) { bhK[ks0][j] = *reinterpret_cast<const bf16x8*>(b_hi + off); if (SPLIT) blK[ks0][j] = *reinterpret_cast<const bf16x8*>(b_lo + off); }
          bh[ks0][j] = bhK[ks0][j]; if (SPLIT) bl[ks0][j] = blK[ks0][j];
        } else {
        bh[ks0][j] = *reinterpret_cast<const bf16x8*>(b_hi + off);
        if (SPLIT) bl[ks0][j] = *reinterpret_cast<const bf16x8*>(b_lo + off);
        }
      }
    }
    exp_first = false;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // (independent accumulators per k-step: consecutive MFMAs alternate chains)
      const int ks0 = c / NTERM, term = c % NTERM;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          f32x16& ac = acc[i][j];
          if (SPLIT && term == 0)
            ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[ks0][j], al[ks0][i], ac, 0, 0, 0);
          else if (SPLIT && term == 1)
            ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[ks0][j], ah[ks0][i], ac, 0, 0, 0);
          else
            ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[ks0][j], ah[ks0][i], ac, 0, 0, 0);
        }
      if (ISSUE) {
        __builtin_amdgcn_sched_barrier(0);
        if (c == 0) {
          if (a_fresh) prep_a(); else prep_a_next();
        }
#pragma unroll
        for (int u = c * LPT / NCH; u < (c + 1) * LPT / NCH; ++u) issue_unit(kt + NSTAGE - 1, u);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (ISSUE) advance();
  };

  // accumulator fragment (i, j), lane (l31, lh), register r = 4*g + e  holds
  //   row m = m0 + (wm*FM+i)*32 + l31,  column n = n0 + (wn*FN+j)*32 + 8*g + 4*lh + e
  // vector (16-byte) epilogue accesses need every row pitch and N to be a multiple of 4
  const bool vec4 = ((p.N | p.ldo | (p.epi == EPI_RESIDUAL ? p.ldres : 0)) & 3) == 0;
  // Half-fragment epilogue (64x64-class tiles with two k-groups and ONE fragment per wave): instead of k-group 1 handing its
  // sums over and leaving - the epilogue then runs on one wave per SIMD, where nothing hides a latency - the groups exchange
  // column halves and EACH finishes 16 of the fragment's 32 columns (registers 0-7 of group 0, 8-15 of group 1: exactly the two
  // 16-column statistics blocks).  Workgroup-uniform; the cases it does not cover take the full path below.
  const bool al8 = ((p.N | p.ldo) & 7) == 0;        // row pitches of the bf16 planes are multiples of 16 bytes: 16-byte plane stores are aligned
  constexpr bool HALF_OK = !SC1 && KS == 2 && FM == 1 && FN == 1;
  const bool half_mode = HALF_OK && p.sk_mode == 0 && vec4 && al8 && n0 + BN <= p.N && !p.stats && !p.rowstat_out &&
                         (p.epi == EPI_STORE || p.epi == EPI_RESIDUAL) && nk > 0;
  // padded row space (GemmParams Tv_out): rows [Tv_out, T_out) of every utterance do not exist - stored as zeros, kept out of
  // the statistics.  `padded` is wave-uniform; the per-lane division only runs then.
  const bool padded = p.Tv_out != p.T_out;
  auto row_ok = [&](int m) { return m < p.M && (!padded || m - utt_of(m) * p.T_out < p.Tv_out); };
  // frames that exist in the 32-row block that starts at row mr (32 unless the utterance's last, partial block)
  auto blk_cnt = [&](int mr) { return padded ? min(32, p.Tv_out - (mr - utt_of(mr) * p.T_out)) : 32; };
  auto load4 = [&](const float* base, size_t row_off, int nb, float* dst) {   // dst[0..3] = base[row_off + nb + e]
    if (vec4) {
      const float4 v = nb < p.N ? ld_mut4<SC1>(base + row_off + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[e] = nb + e < p.N ? ld_mut1<SC1>(base + row_off + nb + e) : 0.f;
    }
  };

  // 16 floats of one row: columns nf + 8 g + e (g, e < 4).  SC1: four 16-byte L1-bypassing loads and ONE wait
  auto load_row16 = [&](const float* base, size_t row_off, int nf, float* dst) {
    if (SC1 && vec4 && nf + 28 < p.N) {
      float4 a, b, c, d;
      const float* q = base + row_off + nf;
      asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:32 sc1\n\t"
                   "global_load_dwordx4 %2, %4, off offset:64 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:96 sc1\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(q) : "memory");
      dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w; dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
      dst[8] = c.x; dst[9] = c.y; dst[10] = c.z; dst[11] = c.w; dst[12] = d.x; dst[13] = d.y; dst[14] = d.z; dst[15] = d.w;
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) load4(base, row_off, nf + 8 * g, dst + 4 * g);
    }
  };

  // residual operand of small tiles: fetched before the k-loop so its latency hides under it
  constexpr bool PRE_RES = FM * FN <= 2;
  float rpre[PRE_RES ? FM * FN * 16 : 1];
  auto res_prefetch = [&]() __attribute__((always_inline)) {
  if (HALF_OK && half_mode) {                      // this wave's 16 columns: two 16-byte loads
    if (p.epi == EPI_RESIDUAL) {
      const float* rp = p.res + (size_t)min(m0 + wm * 32 + l31, p.M - 1) * p.ldres + n0 + wn * 32 + kgrp * 16 + 4 * lh;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const float4 a = *reinterpret_cast<const float4*>(rp + 8 * g);
        rpre[4 * g] = a.x; rpre[4 * g + 1] = a.y; rpre[4 * g + 2] = a.z; rpre[4 * g + 3] = a.w;
      }
    }
    return;
  }
  if (PRE_RES && p.epi == EPI_RESIDUAL && p.sk_mode != 1) {   // (mode 3: half of the workgroups prefetch in vain)
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      // (only the waves that will run this fragment's epilogue: with two k-groups and one row fragment the second group hands
      // its sums over and leaves - its prefetch was 16 KiB of loads per workgroup for nothing)
      if (KS == 2 && !own_col && ((FM % 2 == 0) ? ((i & 1) != kgrp) : (kgrp != 0))) continue;
      const size_t ro = (size_t)min(m0 + (wm * FM + i) * 32 + l31, p.M - 1) * p.ldres;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if (KS == 2 && own_col && (j & 1) != kgrp) continue;
        const int nfj = n0 + (wn * FN + j) * 32;
        float* dst = &rpre[(j * FM + i) * 16];
        if (!SC1 && vec4 && nfj + 32 <= p.N) {       // (wave-uniform) whole fragment inside N: four unguarded 16-byte loads
          const float* rp = p.res + ro + nfj + 4 * lh;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(rp + 8 * g);
            dst[4 * g] = a.x; dst[4 * g + 1] = a.y; dst[4 * g + 2] = a.z; dst[4 * g + 3] = a.w;
          }
        } else {
          load_row16(p.res, ro, nfj + 4 * lh, dst);
        }
      }
    }
  }
  };

  // fused LayerNorm (consumer side): per-row mean / rstd of this tile's rows from the producer's partials
  __shared__ float2 s_ln[BM];
  if (p.ln_stat && p.sk_mode != 1) {
    for (int r = tid; r < BM; r += 64 * NWV) {
      const int m = min(m0 + r, p.M - 1);
      const float2* src = reinterpret_cast<const float2*>(p.ln_stat) + (size_t)m * p.ln_nblk;
      // partials per 32-column block: (sum, sum of squared deviations from the block's own mean); combined with the
      // parallel-variance formula, so no E[x^2] - mean^2 cancellation for rows whose mean is large against their spread.
      // All of a row's partials are requested at once (launch_gemm: at most 16 blocks): ONE memory round trip at the head of
      // the workgroup instead of two dependent ones (round 4: 5000-9000 of these GEMMs' prologue cycles were here; +1.5 % end
      // to end.  Reducing them BEHIND the first k-tiles' DMAs instead - no barrier, the round trip under the issue phase - was
      // -1.6 %: the compiler's wait for these loads then also waits for every DMA behind them.)
      float2 v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = k < p.ln_nblk ? ld_mut2<SC1>(src + k) : make_float2(0.f, 0.f);
      float s1 = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s1 += v[k].x;
      const float inv_c = 1.0f / (float)(p.ln_nblk * 32);
      const float mean = s1 * inv_c;
      float m2 = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float dm = v[k].x * (1.0f / 32.0f) - mean;
        m2 += k < p.ln_nblk ? v[k].y + 32.0f * dm * dm : 0.f;
      }
      s_ln[r] = make_float2(mean, 1.0f / sqrtf(m2 * inv_c + p.ln_eps));
    }
    __syncthreads();
  }

  // bias of this tile's columns -> LDS by DMA, oldest in the queue: landed before the first tile's wait returns
  __shared__ __attribute__((aligned(16))) float s_bias[BN < 64 ? 64 : BN];
  // LayerNorm consumer: u[n] of this tile's columns the same way (a global load at the head of the epilogue would be a
  // dependent L2 round trip on every workgroup's critical path)
  __shared__ __attribute__((aligned(16))) float s_u[BN < 64 ? 64 : BN];
  auto aux_loads = [&]() __attribute__((always_inline)) {
    if (wave < (BN + 63) / 64) {
      const int n = min(n0 + wave * 64 + lane, p.N - 1);
      glds4(p.bias ? (const void*)(p.bias + n) : (const void*)p.zero_page, (unsigned)(size_t)s_bias + wave * 256);
    }
    if (p.ln_stat && wave < (BN + 63) / 64) {
      const int n = min(n0 + wave * 64 + lane, p.N - 1);
      glds4((const void*)(p.ln_u + n), (unsigned)(size_t)s_u + wave * 256);
    }
    res_prefetch();
  };
  aux_loads();   // (ahead of the k-tiles: issued right behind the first tile's DMAs instead, prologue + first-tile wait grew by ~5 % in the per-launch trace)

  // ---- main loop: wait(tile kt) -> barrier -> multiply tile kt with the DMA of tile kt+NSTAGE-1 interleaved ----
  // NSTAGE-1 tiles are in flight; the counted vmcnt leaves the younger ones outstanding across the barrier
  DV_TRACE(10);    // residual prefetch / LayerNorm rows / bias DMA issued
#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t) {
    if (t < nk) issue(t);
#ifdef DV_GEMM_TRACING
    if (t == 0) DV_TRACE(11);
#endif
  }
  DV_TRACE(1);
  // steady state: tile kt+NSTAGE-1 is issued while tile kt is multiplied; NSTAGE-2 younger tiles stay in flight
  const int n_steady = nk - (NSTAGE - 1);
  int kt = 0;
#ifdef DV_GEMM_TRACING
  unsigned long long tr_vm = 0, tr_bar = 0, tr_step = 0;
#endif
  for (; kt < n_steady; ++kt) {
#ifdef DV_GEMM_TRACING
    const unsigned long long tr0 = __builtin_amdgcn_s_memtime();
#endif
    wait_vmcnt<DV_GEMM_EXP == 4 ? 0 : (NSTAGE - 2) * LPT_W>();
#ifdef DV_GEMM_TRACING
    const unsigned long long tr1 = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_barrier();                    // tile kt visible; every wave is done with tile kt-1
#ifdef DV_GEMM_TRACING
    const unsigned long long tr2 = __builtin_amdgcn_s_memtime();
    if (kt == 0) DV_TRACE(2);
#endif
    step(kt, std::true_type{});                      // DMA overwrites the stage tile kt-1 was read from
#ifdef DV_GEMM_TRACING
    tr_vm += tr1 - tr0; tr_bar += tr2 - tr1; tr_step += __builtin_amdgcn_s_memtime() - tr2;
#endif
  }
#ifdef DV_GEMM_TRACING
  if (p.trace && threadIdx.x == 0 && blockIdx.x < 8192) {   // the steady loop's wait sums
    g_gemm_trace[blockIdx.x * DV_TR_W + 12] = tr_vm; g_gemm_trace[blockIdx.x * DV_TR_W + 13] = tr_bar; g_gemm_trace[blockIdx.x * DV_TR_W + 14] = tr_step;
  }
#endif
  for (; kt < nk; ++kt) {                            // drain: nothing left to issue
    const int younger = min(NSTAGE - 2, nk - 1 - kt);
    if (younger >= 2) wait_vmcnt<2 * LPT_W>();
    else if (younger == 1) wait_vmcnt<LPT_W>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#ifdef DV_GEMM_TRACING
    if (kt == 0) DV_TRACE(2);
#endif
    step(kt, std::false_type{});
  }

  // ---- in-epilogue GroupNorm, first half (GnxParams, dv_common.h): wait for the statistics of the groups this tile's columns
  // belong to and build the tile's per-column affine in LDS (s_gA, s_gB); `nwa` = waves of the workgroup that are still here ----
  // (GroupNorm of this GEMM's own output - GnxParams - and of a concatenated consumer's skip slice: gnx_device.h)
  __shared__ GnxShared<BN> s_gnx;
  float* const s_gA = s_gnx.gA; float* const s_gB = s_gnx.gB;
  auto gnx_table = [&](const int nwa) __attribute__((always_inline)) {
    GnxTile t;
    t.M = p.M; t.N = p.N; t.T_out = p.T_out; t.Tv_out = p.Tv_out; t.m0 = m0; t.n0 = n0; t.bm = BM; t.bn = BN; t.bq = utt_of(m0);
    gnx_finish_table<BN>(p.gnx, t, s_gnx, tid, lane, wave, nwa, [&](int k) { DV_TRACE(k); });
  };

  if (nk == 0) { wait_vmcnt<0>(); __syncthreads(); }   // epilogue-only launch: the bias DMA has landed
  DV_TRACE(3);
  // ---- half-fragment epilogue (half_mode above): both k-groups stay, each finishes 16 columns of its wave pair's fragment ----
  if constexpr (HALF_OK) {
    if (half_mode) {
      __builtin_amdgcn_s_barrier();                  // every wave is done reading the ring
      float4* red4 = reinterpret_cast<float4*>(smem);
      // hand the OTHER group's column half over: group 0 its registers 8-15, group 1 its registers 0-7
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int a = 4 * g, b = 8 + 4 * g;
        red4[((kgrp * NWQ + wq) * 2 + g) * 64 + lane] =
            kgrp ? make_float4(acc[0][0][a], acc[0][0][a + 1], acc[0][0][a + 2], acc[0][0][a + 3])
                 : make_float4(acc[0][0][b], acc[0][0][b + 1], acc[0][0][b + 2], acc[0][0][b + 3]);
      }
      __syncthreads();
      float vv[8];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const float4 o4 = red4[(((kgrp ^ 1) * NWQ + wq) * 2 + g) * 64 + lane];
        const int a = 4 * g, b = 8 + 4 * g;
        // (group 0's sums first, as on the full path: acc(group 0) + acc(group 1))
        vv[a] = kgrp ? o4.x + acc[0][0][b] : acc[0][0][a] + o4.x;
        vv[a + 1] = kgrp ? o4.y + acc[0][0][b + 1] : acc[0][0][a + 1] + o4.y;
        vv[a + 2] = kgrp ? o4.z + acc[0][0][b + 2] : acc[0][0][a + 2] + o4.z;
        vv[a + 3] = kgrp ? o4.w + acc[0][0][b + 3] : acc[0][0][a + 3] + o4.w;
      }
      DV_TRACE(22);
      DV_TRACE(4);
      const bool gnx_h = p.gnx.xchg != nullptr;
      const int coff = kgrp * 16;                    // this wave's columns inside the fragment
      const int ncol = n0 + wn * 32 + coff;          // first of them
      const int rl = wm * 32 + l31, m = m0 + rl, mrow0 = m0 + wm * 32;
      const bool m_in = m < p.M, m_ok = row_ok(m);   // inside the tensor / a frame that exists (padded row spaces: dv_common.h)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int cl = wn * 32 + coff + 4 * lh + 8 * g;                    // tile-local column of e = 0
        if (p.ln_stat) {
          const float2 st = s_ln[rl];
          const float4 u4 = *reinterpret_cast<const float4*>(s_u + cl);
          vv[4 * g] = st.y * (vv[4 * g] - st.x * u4.x); vv[4 * g + 1] = st.y * (vv[4 * g + 1] - st.x * u4.y);
          vv[4 * g + 2] = st.y * (vv[4 * g + 2] - st.x * u4.z); vv[4 * g + 3] = st.y * (vv[4 * g + 3] - st.x * u4.w);
        }
        const float4 b4 = *reinterpret_cast<const float4*>(s_bias + cl);
        vv[4 * g] += b4.x; vv[4 * g + 1] += b4.y; vv[4 * g + 2] += b4.z; vv[4 * g + 3] += b4.w;
      }
      if (p.epi == EPI_RESIDUAL) {
#pragma unroll
        for (int r = 0; r < 8; ++r) vv[r] += rpre[r];
      }
      if (p.relu) {
#pragma unroll
        for (int r = 0; r < 8; ++r) vv[r] = fmaxf(vv[r], 0.f);
      }
      if (p.rowmask) {
        const float rmask = p.rowmask[m < p.M ? m : p.M - 1];
#pragma unroll
        for (int r = 0; r < 8; ++r) vv[r] *= rmask;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) vv[r] = m_ok ? vv[r] : 0.f;
      DV_TRACE(16);
#if DV_GNX_STATS_FIRST
      // (round 5: the block statistics are published BEFORE this wave's result stores are issued - the other workgroups of the
      // launch wait for the words, nobody waits for the stores; k_chain_ff has had this order since round 4)
      if (p.stats16) {                               // this wave's 32 x 16 block: (sum, squared deviations about its own mean)
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) a1 += vv[r];
        a1 = wave_sum64(a1);
        const float mb = padded ? a1 / (float)(16 * blk_cnt(mrow0)) : a1 * (1.0f / 512.0f);
#pragma unroll
        for (int r = 0; r < 8; ++r) { const float dv = (!padded || m_ok) ? vv[r] - mb : 0.f; a2 = fmaf(dv, dv, a2); }
        a2 = wave_sum64(a2);
        if (lane == 0 && mrow0 < p.M) {
          const size_t e = (size_t)(mrow0 >> 5) * (p.N >> 4) + (ncol >> 4);
          reinterpret_cast<float2*>(p.stats16)[e] = make_float2(a1, a2);
          if (gnx_h)
            __hip_atomic_store(p.gnx.xchg + e, (unsigned long long)__float_as_uint(a1) | ((unsigned long long)__float_as_uint(a2) << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
#endif
      if (m_in) {                                    // (padding rows are stored as zeros)
        const size_t ob = (size_t)m * p.ldo + ncol;
        if (p.out) {
#pragma unroll
          for (int g = 0; g < 2; ++g)
            dv_st16(p.out + ob + 4 * lh + 8 * g, make_float4(vv[4 * g], vv[4 * g + 1], vv[4 * g + 2], vv[4 * g + 3]));
        }
        if (p.out_hi) store_planes8(p.out_hi, p.out_lo, ob, lh, vv);
      }
      DV_TRACE(17);
#if !DV_GNX_STATS_FIRST
      if (p.stats16) {                               // this wave's 32 x 16 block: (sum, squared deviations about its own mean)
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) a1 += vv[r];
        a1 = wave_sum64(a1);
        const float mb = padded ? a1 / (float)(16 * blk_cnt(mrow0)) : a1 * (1.0f / 512.0f);
#pragma unroll
        for (int r = 0; r < 8; ++r) { const float dv = (!padded || m_ok) ? vv[r] - mb : 0.f; a2 = fmaf(dv, dv, a2); }
        a2 = wave_sum64(a2);
        if (lane == 0 && mrow0 < p.M) {
          const size_t e = (size_t)(mrow0 >> 5) * (p.N >> 4) + (ncol >> 4);
          reinterpret_cast<float2*>(p.stats16)[e] = make_float2(a1, a2);
          if (gnx_h)
            __hip_atomic_store(p.gnx.xchg + e, (unsigned long long)__float_as_uint(a1) | ((unsigned long long)__float_as_uint(a2) << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
#endif
      DV_TRACE(18);
      if (gnx_h) {
        gnx_table(NWV);
        if (m_in) {
          float y[8];
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const int cl = wn * 32 + coff + 4 * lh + 8 * g;
            const float4 sa = *reinterpret_cast<const float4*>(s_gA + cl);
            const float4 sb = *reinterpret_cast<const float4*>(s_gB + cl);
            y[4 * g] = fmaf(vv[4 * g], sa.x, sb.x); y[4 * g + 1] = fmaf(vv[4 * g + 1], sa.y, sb.y);
            y[4 * g + 2] = fmaf(vv[4 * g + 2], sa.z, sb.z); y[4 * g + 3] = fmaf(vv[4 * g + 3], sa.w, sb.w);
          }
          if (p.gnx.silu) {
#pragma unroll
            for (int r = 0; r < 8; ++r) y[r] = y[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-y[r]));
          }
          store_planes8(p.gnx.y_hi, p.gnx.y_lo, (size_t)m * p.N + ncol, lh, y);
        }
      }
      DV_TRACE(5);
      return;
    }
  }
  // KS == 2: add the two k-groups' partial accumulators through LDS (lane-linear, conflict-free).  With an even
  // number of row fragments each k-group keeps the sums of ITS fragments (i % 2 == kgrp) and both run the epilogue on
  // their half, so all eight waves share the (store- and GELU-bound) epilogue; otherwise group 1 hands everything over.
  if (KS == 2) {
    __builtin_amdgcn_s_barrier();                    // every wave is done reading the ring
    float4* red4 = reinterpret_cast<float4*>(smem);  // (16-byte LDS accesses, lane-linear: a quarter of the instructions of the dword form)
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if (!own(i, j)) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            red4[((i * FN + j) * 4 + g) * (64 * NWQ) + wq * 64 + lane] =
                make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        }
      }
    }
    __syncthreads();
    if (!SPLIT_EPI && !own_col && kgrp == 1) return;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if (own(i, j)) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 v = red4[((i * FN + j) * 4 + g) * (64 * NWQ) + wq * 64 + lane];
            // (k-group 0's sums first, whichever group adds: the same bits as the one-finisher form)
            if (kgrp == 0) { acc[i][j][4 * g] += v.x; acc[i][j][4 * g + 1] += v.y; acc[i][j][4 * g + 2] += v.z; acc[i][j][4 * g + 3] += v.w; }
            else { acc[i][j][4 * g] = v.x + acc[i][j][4 * g]; acc[i][j][4 * g + 1] = v.y + acc[i][j][4 * g + 1]; acc[i][j][4 * g + 2] = v.z + acc[i][j][4 * g + 2]; acc[i][j][4 * g + 3] = v.w + acc[i][j][4 * g + 3]; }
          }
        }
      }
    }
  }
  auto my_frag_row = [&](int i) { return !SPLIT_EPI || ((i & 1) == kgrp); };
  auto my_frag = [&](int i, int j) { return KS != 2 || own(i, j); };
  DV_TRACE(22);                                      // k-groups added up
  if (p.sk_mode == 3) {                              // fused split-K pair: hand over, or finish
    float4* d0 = reinterpret_cast<float4*>(p.sk_buf) + (size_t)ksel * sk_slice + sk_tile + wq * 64 + lane;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      if (!my_frag_row(i)) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          st_handover16(d0 + (size_t)((i * FN + j) * 4 + g) * (64 * NWQ),
                        make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]));
    }
    wait_vmcnt<0>();                                 // this thread's dump has been written through
    __shared__ unsigned s_arrival;
    __syncthreads();                                 // (the waves of a k-group that handed everything over have left)
    unsigned* const ticket = p.sk_ticket + (m0 / BM) * ((p.N + BN - 1) / BN) + n0 / BN;
    if (tid == 0) s_arrival = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_arrival == 0) return;                      // the partner finishes this tile
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    const float4* s0 = reinterpret_cast<const float4*>(p.sk_buf) + (size_t)(ksel ^ 1) * sk_slice + sk_tile + wq * 64 + lane;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      if (!my_frag_row(i)) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        float4 v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = ld_handover16(s0 + (size_t)((i * FN + j) * 4 + g) * (64 * NWQ));
        wait_vmcnt<0>();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          acc[i][j][4 * g] += v[g].x; acc[i][j][4 * g + 1] += v[g].y; acc[i][j][4 * g + 2] += v[g].z; acc[i][j][4 * g + 3] += v[g].w;
        }
      }
    }
  }
  if (p.sk_mode == 1) {                              // first pass of a split-K pair: dump this k-slice and leave
    float4* d0 = reinterpret_cast<float4*>(p.sk_buf) + (size_t)ksel * sk_slice + sk_tile + wq * 64 + lane;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      if (!my_frag_row(i)) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          d0[(size_t)((i * FN + j) * 4 + g) * (64 * NWQ)] =
              make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
    }
    return;
  }

  // ---- epilogue ----
  DV_TRACE(4);
  const bool gnx = !SC1 && p.gnx.xchg != nullptr;
  // store 4 consecutive columns of one row: fp32 and/or split bf16 planes
  auto store4 = [&](size_t o, int nb, const float* v) {
    if (vec4) {
      if (nb >= p.N) return;
      if (p.out) *reinterpret_cast<float4*>(p.out + o) = make_float4(v[0], v[1], v[2], v[3]);
      if (p.out_hi) {
        const unsigned h01 = cvt_pk_bf16(v[0], v[1]), h23 = cvt_pk_bf16(v[2], v[3]);
        *reinterpret_cast<uint2*>(p.out_hi + o) = make_uint2(h01, h23);
        if (p.out_lo) {
          const unsigned l01 = cvt_pk_bf16(v[0] - __uint_as_float(h01 << 16), v[1] - __uint_as_float(h01 & 0xffff0000u));
          const unsigned l23 = cvt_pk_bf16(v[2] - __uint_as_float(h23 << 16), v[3] - __uint_as_float(h23 & 0xffff0000u));
          *reinterpret_cast<uint2*>(p.out_lo + o) = make_uint2(l01, l23);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (nb + e >= p.N) continue;
        if (p.out) p.out[o + e] = v[e];
        if (p.out_hi) {
          const unsigned hb = cvt_pk_bf16(v[e], 0.f);
          p.out_hi[o + e] = (bf16_t)(hb & 0xffffu);
          if (p.out_lo) p.out_lo[o + e] = (bf16_t)(cvt_pk_bf16(v[e] - __uint_as_float(hb << 16), 0.f) & 0xffffu);
        }
      }
    }
  };

  if (p.epi == EPI_GEGLU) {
    // packed column order: per 64-column block, [32 x a | 32 x gate]  (needs FN == 2 per wave)
    if constexpr (FN == 2) {
      const int blk = (n0 + wn * 64) >> 6;          // 64-column block index
      const int nca = n0 + wn * 64 + 4 * lh;        // packed column of `a` for g = 0, e = 0
      float ba[16], bg[16], ua[16], ug[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nb = nca + 8 * g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = nb + e < p.N;
          ba[4 * g + e] = s_bias[wn * 64 + 8 * g + 4 * lh + e];
          bg[4 * g + e] = s_bias[wn * 64 + 32 + 8 * g + 4 * lh + e];
          ua[4 * g + e] = (p.ln_stat && ok) ? s_u[wn * 64 + 8 * g + 4 * lh + e] : 0.f;
          ug[4 * g + e] = (p.ln_stat && ok) ? s_u[wn * 64 + 32 + 8 * g + 4 * lh + e] : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        if (!my_frag_row(i)) continue;
        const int rl = (wm * FM + i) * 32 + l31;
        const int m = m0 + rl;
        if (m >= p.M) continue;
        const float2 st = p.ln_stat ? s_ln[rl] : make_float2(0.f, 1.f);
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float a = acc[i][0][r], gt = acc[i][1][r];
          if (p.ln_stat) {
            a = st.y * (a - st.x * ua[r]);
            gt = st.y * (gt - st.x * ug[r]);
          }
          v[r] = (a + ba[r]) * gelu_erf(gt + bg[r]);
        }
        // output column of packed `a` column nb: 32 per 64-column block; nb < N  <=>  oc-run inside N/2
        if (vec4 && al8 && p.out_hi && !p.out && n0 + wn * 64 + 64 <= p.N) {   // (wave-uniform) whole block inside N: 16-byte plane stores
          store_planes16(p.out_hi, p.out_lo, (size_t)m * p.ldo + blk * 32, lh, v);
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) store4((size_t)m * p.ldo + blk * 32 + 8 * g + 4 * lh, nca + 8 * g, &v[4 * g]);
        }
      }
    }
    DV_TRACE(5);
    return;
  }
  // `full` (wave-uniform): all 32 columns of the fragment lie inside N and 16-byte accesses are allowed - the stores and the
  // residual loads then run without per-column guards / per-store branches (the guarded store4 form costs ~40 instructions and 8
  // branches per 16-byte store, ~1.5 k cycles per fragment with one wave per SIMD left to hide nothing: tools/gemm_trace_fwd.py).
  // [Two full instantiations of this loop body (guards compiled out) made hipcc lose the wave-uniformity of the k-range state in
  // several tiles - the argument block went through scratch memory; the runtime flag keeps one body.]
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const bool full = vec4 && al8 && n0 + (wn * FN + j) * 32 + 32 <= p.N;
    const int nf = n0 + (wn * FN + j) * 32 + 4 * lh;   // column of g = 0, e = 0
    float bv[16], un[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {                    // (16-byte LDS reads; columns >= N hold the bias of column N - 1 / are masked below)
      const float4 b4 = *reinterpret_cast<const float4*>(s_bias + (wn * FN + j) * 32 + 4 * lh + 8 * g);
      bv[4 * g] = b4.x; bv[4 * g + 1] = b4.y; bv[4 * g + 2] = b4.z; bv[4 * g + 3] = b4.w;
    }
    if (p.ln_stat) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nf + 8 * (r >> 2) + (r & 3);
        un[r] = (p.ln_stat && n < p.N) ? s_u[(wn * FN + j) * 32 + 4 * lh + 8 * (r >> 2) + (r & 3)] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      if (!my_frag(i, j)) continue;
      const int mrow0 = m0 + (wm * FM + i) * 32;
      const int rl = (wm * FM + i) * 32 + l31;
      const int m = m0 + rl;
      const bool m_in = m < p.M, m_ok = row_ok(m);   // inside the tensor / a frame that exists
      const int mc = m < p.M ? m : p.M - 1;
      // residual operand: 4 x 16-byte loads issued back to back (clamped row), one wait
      float rv[16];
      if (p.epi == EPI_RESIDUAL) {
        if (PRE_RES) {
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = rpre[(j * FM + i) * 16 + r];
        } else if (full && !SC1) {
          const float* rp = p.res + (size_t)mc * p.ldres + nf;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(rp + 8 * g);
            rv[4 * g] = a.x; rv[4 * g + 1] = a.y; rv[4 * g + 2] = a.z; rv[4 * g + 3] = a.w;
          }
        } else {
          load_row16(p.res, (size_t)mc * p.ldres, nf, rv);
        }
      }
      // (one wave-uniform branch per step instead of sixteen selects each; same operations in the same order)
      float vv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) vv[r] = acc[i][j][r];
      if (p.ln_stat) {
        const float2 st = s_ln[rl];
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[r] = st.y * (vv[r] - st.x * un[r]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) vv[r] += bv[r];
      if (p.epi == EPI_RESIDUAL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[r] += rv[r];
      }
      if (p.relu) {
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[r] = fmaxf(vv[r], 0.f);
      }
      if (p.rowmask) {
        const float rmask = p.rowmask[mc];
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[r] *= rmask;
      }
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[r] = m_ok ? vv[r] : 0.f;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[r] = (m_ok && nf + 8 * (r >> 2) + (r & 3) < p.N) ? vv[r] : 0.f;
      }
      if (i == 0 && j == 0) DV_TRACE(16);            // bias / residual / LayerNorm applied (first fragment)
      if (gnx) {                                     // the values stay in registers for the normalising second pass
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = vv[r];
      }
#if DV_GNX_STATS_FIRST
      // (the 32x16 block statistics - published to the other workgroups of the launch when a GroupNorm is finished here - go out
      // BEFORE this fragment's result stores are issued: the words are what the launch waits for)
      if (p.stats16) {
        // per (32-row, 16-column) block: (sum, squared deviations from the block's OWN mean) - no E[x^2] - mean^2
        // cancellation when |mean| >> spread; the consumer combines blocks with the parallel-variance formula in fp64.
        // Registers 0-7 / 8-15 are the fragment's first / second 16 columns (column = 8g + 4lh + e, r = 4g + e);
        // every lane holds columns of both halves, so both sums run over all 64 lanes.
        float a1[2] = {0.f, 0.f}, a2[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) a1[r >> 3] += vv[r];
        auto wsum = [&](float v) { return wave_sum64(v); };   // (DPP + scalar registers: dv_device.h)
        a1[0] = wsum(a1[0]); a1[1] = wsum(a1[1]);
        const float inv_n = padded ? 1.0f / (float)(16 * blk_cnt(mrow0)) : 1.0f / 512.0f;
        const float mb[2] = {a1[0] * inv_n, a1[1] * inv_n};
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float dv = (!padded || m_ok) ? vv[r] - mb[r >> 3] : 0.f; a2[r >> 3] = fmaf(dv, dv, a2[r >> 3]); }
        a2[0] = wsum(a2[0]); a2[1] = wsum(a2[1]);
        const int cb0 = (n0 + (wn * FN + j) * 32) >> 4;
        if (lane < 2 && mrow0 < p.M && (cb0 + lane) * 16 < p.N) {
          float2* const dst = reinterpret_cast<float2*>(p.stats16) + (size_t)(mrow0 >> 5) * (p.N >> 4) + cb0 + lane;
          const float2 val = make_float2(lane ? a1[1] : a1[0], lane ? a2[1] : a2[0]);
          *dst = val;
          if (gnx)     // for the other workgroups of this launch: one 8-byte word (sum, M2), written through
            __hip_atomic_store(p.gnx.xchg + (size_t)(mrow0 >> 5) * (p.N >> 4) + cb0 + lane,
                               (unsigned long long)__float_as_uint(val.x) | ((unsigned long long)__float_as_uint(val.y) << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
#endif
      if (p.epi == EPI_STORE_NCT) {
        // [B, N, T_out]: the 32 lanes of a half-wave write 32 consecutive frames of one channel
        if (m_ok) {
          const int b = utt_of(m), t = m - b * p.T_out;     // (row pitch T_out; the [B, N, Tv_out] result has no padding)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = nf + 8 * (r >> 2) + (r & 3);
            if (n < p.N) p.out[((size_t)b * p.N + n) * p.Tv_out + t] = vv[r];
          }
        }
      } else if (m_in) {                              // (padding rows are stored as zeros)
        if (full) {                                  // planes as 16-byte stores: dv_device.h store_planes16
          const size_t ob = (size_t)m * p.ldo + nf;
          if (p.out) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
              dv_st16(p.out + ob + 8 * g, make_float4(vv[4 * g], vv[4 * g + 1], vv[4 * g + 2], vv[4 * g + 3]));
          }
          if (p.out_hi) store_planes16(p.out_hi, p.out_lo, ob - 4 * lh, lh, vv);
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) store4((size_t)m * p.ldo + nf + 8 * g, nf + 8 * g, &vv[4 * g]);
        }
      }
      if (i == 0 && j == 0) DV_TRACE(17);            // stores issued (first fragment)
      if (p.rowstat_out) {   // row partials over this fragment's 32 columns (LayerNorm of the consumer)
        const int nblk_total = (p.N + 31) >> 5;
        const int cb = (n0 + (wn * FN + j) * 32) >> 5;
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += vv[r];
        a = pair_sum32(a);
        const float mb = a * (1.0f / 32.0f);           // block mean; q = squared deviations from it
#pragma unroll
        for (int r = 0; r < 16; ++r) q += (vv[r] - mb) * (vv[r] - mb);
        q = pair_sum32(q);
        if (lh == 0 && m_in && cb < nblk_total)
          reinterpret_cast<float2*>(p.rowstat_out)[(size_t)m * nblk_total + cb] = make_float2(a, q);
      }
      if (p.stats) {
        // column sums over this 32-row block: 16 values per lane summed across the 32 lanes of each half by a
        // halving butterfly (8 + 4 + 2 + 1 exchanges, then one add with the neighbour lane)
        float s1[16], s2[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { s1[r] = vv[r]; s2[r] = vv[r] * vv[r]; }
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1) {
          const bool up = (l31 & (2 * w)) != 0;      // lane keeps the upper half of its 2w values
#pragma unroll
          for (int k = 0; k < w; ++k) {
            const float k1 = up ? s1[k + w] : s1[k], x1 = up ? s1[k] : s1[k + w];
            const float k2 = up ? s2[k + w] : s2[k], x2 = up ? s2[k] : s2[k + w];
            s1[k] = k1 + __shfl_xor(x1, 2 * w);
            s2[k] = k2 + __shfl_xor(x2, 2 * w);
          }
        }
        s1[0] += __shfl_xor(s1[0], 1);
        s2[0] += __shfl_xor(s2[0], 1);
        const int r = (l31 >> 1) & 15;               // bit 4 -> r bit 3, ... bit 1 -> r bit 0
        const int n = nf + 8 * (r >> 2) + (r & 3);
        if ((l31 & 1) == 0 && n < p.N && mrow0 < p.M)
          reinterpret_cast<float2*>(p.stats)[(size_t)(mrow0 >> 5) * p.N + n] = make_float2(s1[0], s2[0]);
      }
#if !DV_GNX_STATS_FIRST
      if (p.stats16) {
        // per (32-row, 16-column) block: (sum, squared deviations from the block's OWN mean) - no E[x^2] - mean^2
        // cancellation when |mean| >> spread; the consumer combines blocks with the parallel-variance formula in fp64.
        // Registers 0-7 / 8-15 are the fragment's first / second 16 columns (column = 8g + 4lh + e, r = 4g + e);
        // every lane holds columns of both halves, so both sums run over all 64 lanes.
        float a1[2] = {0.f, 0.f}, a2[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) a1[r >> 3] += vv[r];
        auto wsum = [&](float v) { return wave_sum64(v); };   // (DPP + scalar registers: dv_device.h)
        a1[0] = wsum(a1[0]); a1[1] = wsum(a1[1]);
        const float inv_n = padded ? 1.0f / (float)(16 * blk_cnt(mrow0)) : 1.0f / 512.0f;
        const float mb[2] = {a1[0] * inv_n, a1[1] * inv_n};
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float dv = (!padded || m_ok) ? vv[r] - mb[r >> 3] : 0.f; a2[r >> 3] = fmaf(dv, dv, a2[r >> 3]); }
        a2[0] = wsum(a2[0]); a2[1] = wsum(a2[1]);
        const int cb0 = (n0 + (wn * FN + j) * 32) >> 4;
        if (lane < 2 && mrow0 < p.M && (cb0 + lane) * 16 < p.N) {
          float2* const dst = reinterpret_cast<float2*>(p.stats16) + (size_t)(mrow0 >> 5) * (p.N >> 4) + cb0 + lane;
          const float2 val = make_float2(lane ? a1[1] : a1[0], lane ? a2[1] : a2[0]);
          *dst = val;
          if (gnx)     // for the other workgroups of this launch: one 8-byte word (sum, M2), written through
            __hip_atomic_store(p.gnx.xchg + (size_t)(mrow0 >> 5) * (p.N >> 4) + cb0 + lane,
                               (unsigned long long)__float_as_uint(val.x) | ((unsigned long long)__float_as_uint(val.y) << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
#endif
    }
    }
  DV_TRACE(18);                                      // statistics of every fragment written
  if (gnx) {
    // ---- GroupNorm of this GEMM's own output (GnxParams, dv_common.h) ----
    gnx_table((KS == 2 && !SPLIT_EPI && !own_col) ? NWQ : NWV);   // (waves still here)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int cl = (wn * FN + j) * 32 + 4 * lh;    // tile-local column of g = 0, e = 0
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        if (!my_frag(i, j)) continue;
        const int m = m0 + (wm * FM + i) * 32 + l31;
        if (m >= p.M) continue;
        const int nfr = n0 + (wn * FN + j) * 32;
        if (al8 && nfr + 32 <= p.N) {                  // (wave-uniform) the whole fragment: 16-byte plane stores
          float y[16];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 sa = *reinterpret_cast<const float4*>(s_gA + cl + 8 * g);
            const float4 sb = *reinterpret_cast<const float4*>(s_gB + cl + 8 * g);
            y[4 * g] = fmaf(acc[i][j][4 * g], sa.x, sb.x); y[4 * g + 1] = fmaf(acc[i][j][4 * g + 1], sa.y, sb.y);
            y[4 * g + 2] = fmaf(acc[i][j][4 * g + 2], sa.z, sb.z); y[4 * g + 3] = fmaf(acc[i][j][4 * g + 3], sa.w, sb.w);
          }
          if (p.gnx.silu) {
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] = y[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-y[r]));
          }
          store_planes16(p.gnx.y_hi, p.gnx.y_lo, (size_t)m * p.N + nfr, lh, y);
          continue;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (n0 + cl + 8 * g >= p.N) continue;
          const float4 sa = *reinterpret_cast<const float4*>(s_gA + cl + 8 * g);
          const float4 sb = *reinterpret_cast<const float4*>(s_gB + cl + 8 * g);
          float y[4] = {fmaf(acc[i][j][4 * g], sa.x, sb.x), fmaf(acc[i][j][4 * g + 1], sa.y, sb.y),
                        fmaf(acc[i][j][4 * g + 2], sa.z, sb.z), fmaf(acc[i][j][4 * g + 3], sa.w, sb.w)};
          if (p.gnx.silu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = y[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-y[e]));
          }
          const size_t o = (size_t)m * p.N + n0 + cl + 8 * g;
          const unsigned h01 = cvt_pk_bf16(y[0], y[1]), h23 = cvt_pk_bf16(y[2], y[3]);
          *reinterpret_cast<uint2*>(p.gnx.y_hi + o) = make_uint2(h01, h23);
          if (p.gnx.y_lo) {
            const unsigned l01 = cvt_pk_bf16(y[0] - __uint_as_float(h01 << 16), y[1] - __uint_as_float(h01 & 0xffff0000u));
            const unsigned l23 = cvt_pk_bf16(y[2] - __uint_as_float(h23 << 16), y[3] - __uint_as_float(h23 & 0xffff0000u));
            *reinterpret_cast<uint2*>(p.gnx.y_lo + o) = make_uint2(l01, l23);
          }
        }
      }
    }
  }
  DV_TRACE(5);
}

