// Implicit-GEMM contraction kernel for gfx950 (MI355X): every conv1d (k=1/3, stride 1/2,
// nearest-upsample folded into the gather, channel-concat of two sources, 1x1 shortcut folded
// in as a second K-segment) and every Linear of the denoiser (reference unet1d/resnet.py:591-641,
// transformer_1d.py:264-300, attention.py:130-203, attention_processor.py:1008-1046) is one
// launch of this kernel.
//
//   A operand : activations PRE-SPLIT into bf16 hi / lo planes [rows, C] by their producer
//               (norm/activation kernels, GEMM / attention epilogues), gathered by row
//   B operand : weights pre-packed as bf16 hi / lo, [N][K] with K contiguous
//   staging   : global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip),
//               3-stage ring, prefetch distance 2, one raw s_barrier per k-tile, counted
//               s_waitcnt vmcnt(N) so a tile stays in flight across the barrier
//   LDS image : lane-linear rows of BK bf16 (the DMA cannot pad); bank conflicts of the 16-byte
//               MFMA operand reads are removed by XOR-swizzling the 16-byte chunk index with
//               the row on the SOURCE address and again on the read (same involution)
//   MFMA      : v_mfma_f32_32x32x16_bf16, fp32 accumulate; bf16x3 mode issues
//               lo*hi + hi*lo + hi*hi (1.3e-5 rel. error vs fp32), bf16 mode hi*hi
//   epilogue  : + bias, + residual | GEGLU a*gelu_erf(g) | transposed [B,N,T] store; output as
//               fp32 and/or split bf16 planes; optional per-(32-row block, channel) sum / sum of
//               squares of the result (GroupNorm statistics for the consumer, no extra pass)
#include "dv_common.h"
#define DV_GEMM_TRACE_OWNER   // the trace build's stamp buffer lives in this translation unit
#include "gemm_tile.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

template <int BM, int BN, int BK, int WM, int WN, int NSPLIT, int KS>
__global__ __launch_bounds__(64 * WM * WN * KS) void k_gemm(const GemmParams p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  // XCD-aware tile order (workgroup b runs on XCD b % 8; each XCD has its own L2, which starts cold): an XCD's workgroups
  // form a rectangle of the tile grid so that its L2 fetches (rows / xm) of A and (columns / xn) of W once.
  // every 64-byte line of the argument block is requested NOW, together: the tile mapping below branches on a field of
  // the last line, and the loads of the tile routine would otherwise only be issued once that branch has resolved (two
  // dependent scalar-cache misses at the head of every workgroup: measured +300 cycles)
  asm volatile("" ::"s"(p.seg[0].a0_hi), "s"(p.seg[1].a0_hi), "s"(p.seg[1].pad), "s"(p.T_in), "s"(p.w_hi), "s"(p.M), "s"(p.res),
               "s"(p.out_hi), "s"(p.zero_page), "s"(p.ln_u), "s"(p.gnx.xchg), "s"(p.gnx.y_hi), "s"(p.xcd_n), "s"(p.xcd_inv_tn));
  // ONE call site of the tile routine.  [Rounds 1-3 called it from both branches of the mapping: two inlined copies.  Waves
  // leave the routine at different points (k-groups that hand their sums over), so the structurised control flow runs from the
  // first copy's epilogue THROUGH the second copy under an empty exec mask - and hipcc's wait-count insertion, which knows
  // nothing about exec, carried the first epilogue's pending loads (GroupNorm hand-over polls, affine parameters) into the
  // second copy's k-loop: whenever the register allocator happened to reuse one of their destination registers for the
  // k-loop's pointers it inserted `s_waitcnt vmcnt(0)` there - a full drain of the LDS-DMA queue per k-tile.  Round 4 hit
  // that by adding two epilogue scalars: -10 % end to end, found by a same-box A/B.  tools/kloop_waits.py checks the
  // assembly for it; tests/test_kernel_resources.py runs the check.]
  const int n_tiles_n = (p.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  int tm0, tn0, ksel = 0;
  if (p.xcd_n > 0) {
    // launch_gemm checked: 8 | nwg; (k-slices) x xm x xn = 8; xm | row tiles, xn | column tiles
    // (xs, xm, xn are powers of two; the one real division - by the rectangle's width - is a multiply by the host's
    // 16-bit reciprocal, exact for i < 1024: this runs before the first DMA can be issued)
    const int x = bid & 7, i = bid >> 3;
    const int ks_i = x >> p.xcd_sh_mn, r = x & ((1 << p.xcd_sh_mn) - 1), xm_i = r >> p.xcd_sh_n, xn_i = r & (p.xcd_n - 1);
    const int lm = (i * p.xcd_inv_tn) >> 16, ln = i - lm * p.xcd_tn;
    tm0 = (xm_i * p.xcd_tm + lm) * BM; tn0 = (xn_i * p.xcd_tn + ln) * BN; ksel = ks_i;
  } else {
    {
      const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    if (p.sk_mode == 1 || p.sk_mode == 3) {              // XCD-contiguous ids share a k-slice: an L2 holds one slice of A and W
      const int tiles = nwg / p.sk_split;
      ksel = bid / tiles;
      bid -= ksel * tiles;
    }
    tm0 = (bid / n_tiles_n) * BM; tn0 = (bid % n_tiles_n) * BN;
  }
  gemm_tile<BM, BN, BK, WM, WN, NSPLIT, KS, false>(p, tm0, tn0, smem, ksel);
}

template <int BM, int BN, int BK, int WM, int WN, int NSPLIT, int KS>
struct GemmCfg {
  static constexpr int STAGE_B = (BM + BN) * BK * 2 * (NSPLIT == 3 ? 2 : 1);
  static constexpr int SMEM = ((BM == 64 && BN == 64) ? DV_NSTAGE_64 : ((4 * STAGE_B <= 160 * 1024) ? 4 : 3)) * STAGE_B;
  // > 64 KiB of dynamic LDS needs the attribute; set once, outside any stream capture
  static hipError_t init() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm<BM, BN, BK, WM, WN, NSPLIT, KS>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  }
  static hipError_t launch(const GemmParams& p, hipStream_t st) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * (p.sk_mode == 1 || p.sk_mode == 3 ? p.sk_split : 1);
    hipLaunchKernelGGL((k_gemm<BM, BN, BK, WM, WN, NSPLIT, KS>), dim3(tiles), dim3(64 * WM * WN * KS), SMEM, st, p);
    return hipGetLastError();
  }
};

template <int BM, int BN, int BK, int WM, int WN, int KS = 1>
struct GemmTile {
  static hipError_t init() {
    hipError_t e = GemmCfg<BM, BN, BK, WM, WN, 3, KS>::init();
    return e != hipSuccess ? e : GemmCfg<BM, BN, BK, WM, WN, 1, KS>::init();
  }
  static hipError_t launch(const GemmParams& p, bool x3, hipStream_t st) {
    return x3 ? GemmCfg<BM, BN, BK, WM, WN, 3, KS>::launch(p, st) : GemmCfg<BM, BN, BK, WM, WN, 1, KS>::launch(p, st);
  }
};

// tile menu, largest first; *G variants keep a 64-column block inside one wave (FN == 2) for GEGLU
#ifndef DV_T2_WM
#define DV_T2_WM 2   // wave grid of the 64x64 tile (experiment builds: 1x2 / 2x1 - 64x32 / 32x64 per wave, fewer LDS operand reads)
#define DV_T2_WN 2
#endif
template <int BK> struct Tiles {
  using T0 = GemmTile<128, 128, 32, 2, 2, 2>;
  using T0S = GemmTile<128, 128, 32, 2, 2, 1>;
  using T1 = GemmTile<128, 64, BK, 4, 1, (BK == 64 ? 2 : 1)>;
  using T2 = GemmTile<64, 64, BK, DV_T2_WM, DV_T2_WN, (BK == 64 ? 2 : 1)>;   // 8 waves at BK = 64 (k-split pairs)
  using T2S = GemmTile<64, 64, BK, 2, 2, 1>;
  using T2G = GemmTile<64, 64, BK, 2, 1>;
  // (small tiles with ONE k-group - 2 / 1 waves - measured 1500 cycles per k-tile against 1030 for the 8-wave 64x64 tile
  // although they move fewer bytes: too few waves to keep the LDS-DMA queue full.  With 64-deep k-tiles they run two k-groups.)
  using T3 = GemmTile<64, 32, BK, 2, 1, (BK == 64 ? 2 : 1)>;
  using T4 = GemmTile<32, 32, BK, 1, 1, (BK == 64 ? 2 : 1)>;
  using T4G = GemmTile<32, 64, BK, 1, 1>;
  static hipError_t init() {
    hipError_t e;
    if ((e = T0::init()) != hipSuccess) return e;
    if ((e = T0S::init()) != hipSuccess) return e;
    if ((e = T1::init()) != hipSuccess) return e;
    if ((e = T2::init()) != hipSuccess) return e;
    if ((e = T2S::init()) != hipSuccess) return e;
    if ((e = T2G::init()) != hipSuccess) return e;
    if ((e = T3::init()) != hipSuccess) return e;
    if ((e = T4::init()) != hipSuccess) return e;
    return T4G::init();
  }
  static hipError_t launch(const GemmParams& p, bool x3, int min_wg, bool ksplit, hipStream_t st) {
    auto cnt = [&](int bm, int bn) { return ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
    if (p.epi == EPI_GEGLU) {
      if (cnt(128, 64) >= min_wg) return T1::launch(p, x3, st);
      if (cnt(64, 64) >= min_wg) return T2G::launch(p, x3, st);
      return T4G::launch(p, x3, st);
    }
    static const int t1_min = [] { const char* e = getenv("DVITS_GEMM_T1"); return e ? atoi(e) : 192; }();   // 128x64 tiles from 192 of them: 384+ 64x64 tiles would run as two rounds, and would not fit the in-launch GroupNorm hand-over (round 4: 256 -> 192, +0.4 % same box)
    if (BK == 64 && cnt(128, 64) >= t1_min) return T1::launch(p, x3, st);
    if (cnt(64, 64) >= min_wg) return ksplit ? T2::launch(p, x3, st) : T2S::launch(p, x3, st);
    if (cnt(64, 32) >= min_wg) return T3::launch(p, x3, st);
    return T4::launch(p, x3, st);
  }
  static hipError_t launch_forced(const GemmParams& p, bool x3, int tile, hipStream_t st) {
    switch (tile) {
      case GT_T0: return T0::launch(p, x3, st);
      case GT_T1: return T1::launch(p, x3, st);
      case GT_T2: return T2::launch(p, x3, st);
      case GT_T2S: return T2S::launch(p, x3, st);
      case GT_T2G: return T2G::launch(p, x3, st);
      case GT_T3: return T3::launch(p, x3, st);
      case GT_T4: return T4::launch(p, x3, st);
      case GT_T4G: return T4G::launch(p, x3, st);
      default: return hipErrorInvalidValue;
    }
  }
};

hipError_t gemm_init() {
  hipError_t e = Tiles<32>::init();
  if (e != hipSuccess) return e;
  if ((e = Tiles<64>::init()) != hipSuccess) return e;
  return conv3_init();
}

// Tunables (env DVITS_GEMM_CFG="big,min_wg,bk64"): workgroup-count threshold for the 128x128x32 tile,
// minimum workgroups wanted from the smaller tiles, and whether 64-deep k-tiles are used when the channel
// counts allow.  The denoiser's GEMMs are small (M = B*T_l <= 8192, N = 128..4096): filling 256 CUs with
// >= 1-2 workgroups each matters more than tile efficiency.
struct GemmTune { int big = 144, min_wg = 128, bk64 = 1, ksplit = 1; };
static const GemmTune& gemm_tune() {
  static GemmTune t = [] {
    GemmTune v;
    if (const char* e = getenv("DVITS_GEMM_CFG")) sscanf(e, "%d,%d,%d,%d", &v.big, &v.min_wg, &v.bk64, &v.ksplit);
    return v;
  }();
  return t;
}

// Split-K pair (env DVITS_SPLITK="max_tiles,min_k,split", "0" disables): a GEMM with at most max_tiles 64x64 tiles
// and K >= min_k runs as two launches - `split` k-slices per tile on split x the workgroups (raw accumulators dumped),
// then an epilogue-only pass that sums the slices.  Only pays when the single launch leaves CUs idle for a long k-loop:
// the 128-frame level (128 tiles) with K >= 1536 gains ~25 %; at 192 tiles the 1.5 rounds of workgroups lose.
// A 128x128-tile variant with 4-8 slices was measured slower everywhere (the dump and the second epilogue are both
// store-bound: ~10 us + ~9 us per GEMM regardless of K).
struct SplitKTune { int max_tiles = 160, min_k = 1536, split = 2; };
static SplitKTune splitk_tune() {   // read when an engine is prepared, not per launch
  SplitKTune v;
  if (const char* e = getenv("DVITS_SPLITK")) { v.max_tiles = 0; sscanf(e, "%d,%d,%d", &v.max_tiles, &v.min_k, &v.split); }
  return v;
}
int gemm_splitk_plan(int M, int N, int K, int epi) {
  const SplitKTune t = splitk_tune();
  const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
  if (tiles > t.max_tiles || K < t.min_k || t.split < 2 || (epi != EPI_STORE && epi != EPI_RESIDUAL)) return 0;
  return t.split;
}

// Tiles that can run this GEMM, for the prepare-time tuner (engine.hip): it times each on the real operands.
// GEGLU needs a 64-column block inside one wave (FN == 2 tiles); 64-deep k-tiles need channel counts % 64 == 0.
int gemm_candidates(const GemmParams& p, int* out, int cap) {
  bool k64 = true;
  for (int s = 0; s < p.nseg; ++s) k64 = k64 && p.seg[s].c0 % 64 == 0 && p.seg[s].c1 % 64 == 0;
  auto cnt = [&](int bm, int bn) { return ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  int n = 0;
  auto add = [&](int id) { if (n < cap) out[n++] = id; };
  if (cnt(128, 128) >= 24) add(GT_T0);
  if (p.epi == EPI_GEGLU) {
    if (cnt(128, 64) >= 32) add(k64 ? (GT_T1 | GT_BK64) : GT_T1);
    add(k64 ? (GT_T2G | GT_BK64) : GT_T2G);
    if (cnt(64, 64) < 64) add(k64 ? (GT_T4G | GT_BK64) : GT_T4G);
    return n;
  }
  if (k64 && cnt(128, 64) >= 64) add(GT_T1 | GT_BK64);
  if (k64) { add(GT_T2 | GT_BK64); add(GT_T2S | GT_BK64); }
  else add(GT_T2);
  if (cnt(64, 64) < 256) add(k64 ? (GT_T3 | GT_BK64) : GT_T3);
  if (cnt(64, 64) < 64) add(k64 ? (GT_T4 | GT_BK64) : GT_T4);
  return n;
}

// environment knobs of launch_gemm that tests flip inside one process: re-read whenever an engine is prepared
static int g_env_xn = [] { const char* e = getenv("DVITS_XCD_N"); return e ? atoi(e) : -1; }();
void gemm_env_refresh() {
  const char* e = getenv("DVITS_XCD_N"); g_env_xn = e ? atoi(e) : -1;
  conv3_env_refresh();
}

// Tile (BM x BN) the shape heuristic of launch_gemm picks for a non-GEGLU GEMM (kept in step with
// Tiles<BK>::launch below; used to plan the in-epilogue GroupNorm: its tiles must not span utterances and must all be resident)
static void gemm_pick_tile(const GemmParams& p, int& bm, int& bn) {
  const GemmTune& tune = gemm_tune();
  auto cnt = [&](int a, int b) { return ((p.M + a - 1) / a) * ((p.N + b - 1) / b); };
  bool k64 = tune.bk64 != 0;
  for (int s = 0; s < p.nseg; ++s) k64 = k64 && p.seg[s].c0 % 64 == 0 && p.seg[s].c1 % 64 == 0;
  if (cnt(128, 128) >= tune.big) { bm = 128; bn = 128; return; }
  const int min_wg = k64 ? tune.min_wg : (tune.min_wg > 0 ? tune.min_wg : 1);
  if (p.epi == EPI_GEGLU) {
    if (cnt(128, 64) >= min_wg) { bm = 128; bn = 64; return; }
    if (cnt(64, 64) >= min_wg) { bm = 64; bn = 64; return; }
    bm = 32; bn = 64; return;
  }
  static const int t1_min = [] { const char* e = getenv("DVITS_GEMM_T1"); return e ? atoi(e) : 192; }();
  if (k64 && cnt(128, 64) >= t1_min) { bm = 128; bn = 64; return; }
  if (cnt(64, 64) >= min_wg) { bm = 64; bn = 64; return; }
  if (cnt(64, 32) >= min_wg) { bm = 64; bn = 32; return; }
  bm = 32; bn = 32;
}

// XCD rectangle of a tm x tn tile grid run as xs k-slices (fills p.xcd_*; xcd_n = 0: row bands of the tile grid)
static void gemm_xcd_rect(GemmParams& p, int tm, int tn, int xs) {
  const int env_xn = g_env_xn;
  long chans = 0, ktot = 0;
  for (int s2 = 0; s2 < p.nseg; ++s2) { chans += p.seg[s2].c0 + p.seg[s2].c1; ktot += (long)p.seg[s2].taps * (p.seg[s2].c0 + p.seg[s2].c1); }
  double best = 0; int best_xn = 0;
  p.xcd_n = 0;
  for (int xn = 1; xn * xs <= 8; xn *= 2) {
    const int xm = 8 / (xs * xn);
    if ((tm * tn * xs) % 8 != 0 || tm % xm != 0 || tn % xn != 0 || (xs != 1 && xs != 2)) continue;
    if (env_xn >= 0 && xn != env_xn) continue;
    const double cost = (double)p.N / xn * ktot + (double)p.M / xm * chans;
    if (best_xn == 0 || cost < best * 0.97) { best = cost; best_xn = xn; }
  }
  if (env_xn != 0 && best_xn > 0 && tm * tn * xs / 8 <= 1024) {
    const int xm = 8 / (xs * best_xn);
    auto lg2 = [](int v) { int s2 = 0; while ((1 << s2) < v) ++s2; return s2; };
    p.xcd_n = best_xn; p.xcd_sh_n = lg2(best_xn); p.xcd_sh_mn = lg2(xm * best_xn);
    p.xcd_tn = tn / best_xn; p.xcd_tm = tm / xm; p.xcd_inv_tn = (65536 + p.xcd_tn - 1) / p.xcd_tn;
  }
}
// Most workgroups of ONE utterance that one XCD runs under the order the kernels map workgroup ids to tiles in (the rectangle / the
// row bands above): tu row tiles per utterance, tm x tn tiles in all.  Workgroup b runs on XCD b % 8 and an XCD starts its workgroups
// in id order; both mappings walk an XCD's tiles utterance by utterance.
static int gemm_utt_tiles_per_xcd(const GemmParams& p, int tu, int tm, int tn) {
  if (p.xcd_n > 0) return (tu < p.xcd_tm ? tu : p.xcd_tm) * p.xcd_tn;
  const int q = (tm * tn + 7) / 8;                   // band of an XCD (row-major tiles)
  return tu * tn < q ? tu * tn : q;
}

// In-launch hand-overs that WAIT (the GroupNorm exchange here, k_ff_split's partial sums) need the partners of a waiting workgroup
// on the chip.  Rounds 2-5 asked for the whole grid to be resident (grid <= CUs): B = 16 or T = 2048 batches fell back to separate
// GroupNorm launches.  What is actually needed is less: workgroup b runs on XCD b % 8 and every XCD starts its workgroups in id
// order, a workgroup only waits for workgroups of ITS OWN utterance, and the kernels walk an XCD's tiles utterance by utterance - so
// the unfinished utterance at the head of an XCD's list is always fully started as long as its share of ONE utterance fits the
// XCD's CUs (one workgroup per CU is always possible), whatever the size of the grid: the utterances behind it simply run in later
// rounds.  [The time-outs round 2 found at 2 x 192 workgroups were this rule broken - 48 tiles of one utterance on one XCD of 32
// CUs - not the grid size.]  DVITS_GNX_ROUNDS=0 restores "whole grid resident".  The waits stay bounded and flagged either way.
bool gemm_handover_rounds() {
  static const bool on = [] { const char* e = getenv("DVITS_GNX_ROUNDS"); return !(e && e[0] == '0'); }();
  return on;
}
int gemm_gnx_plan(const GemmParams& p, int n_cu) {
  if (p.force_tile != GT_AUTO || (p.epi != EPI_STORE && p.epi != EPI_RESIDUAL) || !p.stats16 || p.rowmask || p.relu) return 0;
  const int skc = p.gnx.sk_c;                          // concatenated consumer: groups of [output | skip]
  if (p.M != p.B * p.T_out || p.gnx.groups <= 0 || p.gnx.groups > 64 || skc < 0 || (p.N + skc) % p.gnx.groups != 0) return 0;
  const int cpg = (p.N + skc) / p.gnx.groups;
  if (cpg % 16 != 0 || p.N % 16 != 0 || skc % 16 != 0) return 0;
  int bm, bn, tu;
  if (p.c3_route) {                                    // the convolution kernels: row tiles per utterance, the last one may be short
    bm = p.c3_route == 2 ? 128 : 64; bn = 64;
    if (p.T_out % 32 != 0 || p.N % bn != 0) return 0;
    tu = (p.T_out + bm - 1) / bm;
  } else {
    gemm_pick_tile(p, bm, bn);
    if (p.T_out % bm != 0 || p.N % bn != 0) return 0;
    tu = p.T_out / bm;
  }
  if (skc > 0 && ((skc / 16 + p.N / bn - 1) / (p.N / bn)) * 16 > 128) return 0;   // skip slice per workgroup (gnx_device.h DV_GSK)
  if ((p.T_out / 32) * (cpg / 16) > 256) return 0;     // entries of one group: four per lane of the reducing wave
  const int tm = p.B * tu, tn = p.N / bn, tiles = tm * tn;
  const bool pair = p.sk_buf && p.sk_split == 2 && p.sk_ticket;
  if (p.sk_buf && !pair) return 0;                   // two-launch split-K: not supported with the in-epilogue GroupNorm
  if (pair) {
    // both halves of a fused split-K pair count, although the first arriver of a pair leaves without waiting; the ids of a pair's
    // halves are far apart (an XCD holds one k-half): the plain bound
    if (2 * tiles > n_cu) return 0;
  } else if (tiles > n_cu) {
    if (!gemm_handover_rounds() || n_cu < 8) return 0;
    GemmParams t = p;
    if (p.c3_route == 2) t.xcd_n = 0; else gemm_xcd_rect(t, tm, tn, 1);
    if (gemm_utt_tiles_per_xcd(t, tu, tm, tn) > n_cu / 8) return 0;
  }
  return (p.M / 32) * (p.N / 16);
}

#ifdef DV_GEMM_TRACE
// development build: which launch_gemm call stamps its phases (-1: every one; n: the n-th call since the selection), and what it was
static int g_trace_sel = -1, g_trace_no = 0;
static char g_trace_desc[256] = "";
extern "C" int dv_debug_gemm_trace_select(int n) { g_trace_sel = n; g_trace_no = 0; g_trace_desc[0] = 0; return 0; }
extern "C" int dv_debug_gemm_trace_desc(char* out, int cap) { snprintf(out, (size_t)cap, "%s", g_trace_desc); return g_trace_no; }
#endif

hipError_t launch_gemm(const GemmParams& pin, int precision, hipStream_t st) {
  GemmParams p = pin;
#ifdef DV_GEMM_TRACE
  if (pin.sk_mode == 0) {                            // (not the split-K recursion below)
    p.trace = g_trace_sel < 0 || g_trace_no == g_trace_sel;
    if (p.trace) {
      int k = 0;
      for (int s2 = 0; s2 < p.nseg; ++s2) k += p.seg[s2].taps * (p.seg[s2].c0 + p.seg[s2].c1);
      snprintf(g_trace_desc, sizeof(g_trace_desc), "M=%d N=%d K=%d taps=%d nseg=%d epi=%d stride=%d up=%d%s%s%s%s%s%s split=%d%s",
               p.M, p.N, k, p.seg[0].taps, p.nseg, p.epi, p.stride, p.up_mode, p.stats ? " +colstats" : "", p.stats16 ? " +stats16" : "",
               p.gnx.xchg ? " +gnx" : "", p.out_hi ? " +planes" : "", p.rowstat_out ? " +rowstat" : "", p.ln_stat ? " +ln" : "",
               p.sk_buf ? p.sk_split : 0, (p.sk_buf && p.sk_ticket && p.sk_split == 2) ? "f" : "");
    }
    ++g_trace_no;
  }
#endif
  const bool x3 = precision == 0;
  if (!p.zero_page || (x3 && !p.w_lo)) return hipErrorInvalidValue;
  // padded row spaces (GemmParams): 0 = no padding
  if (p.Tv_out <= 0) p.Tv_out = p.T_out;
  if (p.Tv_in <= 0) p.Tv_in = p.T_in;
  if (p.Tv_out > p.T_out || p.Tv_in > p.T_in || ((p.stats16 || p.gnx.xchg) && p.Tv_out <= p.T_out - 32)) return hipErrorInvalidValue;
  if (p.T_out < 1 || (unsigned long long)(p.M > 0 ? p.M : 1) * (unsigned long long)p.T_out >= (1ull << 32)) return hipErrorInvalidValue;
  p.tout_magic = gemm_tout_magic(p.T_out);
  if (p.ln_stat && (p.ln_nblk < 1 || p.ln_nblk > 16)) return hipErrorInvalidValue;   // (a row's LayerNorm partials are held in registers: gemm_tile.h)
  for (int s2 = 0; s2 < p.nseg; ++s2)     // (a lane on the zero page walks a row's k-tiles inside it: gemm_tile.h prep_a_next)
    if (2 * (size_t)p.seg[s2].c0 + 256 > DV_ZERO_PAGE_BYTES || 2 * (size_t)p.seg[s2].c1 + 256 > DV_ZERO_PAGE_BYTES) return hipErrorInvalidValue;
  // which kernel family runs this launch (GemmParams c3_route: the in-launch GroupNorm is planned for ITS tile grid)
  p.c3_route = 0;
  if (x3 && p.wf_hi && p.wf_lo) {
    if (gemm_conv3_shape_ok(p) && p.Kp == gemm_conv3_k(p)) p.c3_route = 1;
    else if (gemm_conv3_up_ok(p) && p.Kp == 3 * (p.seg[0].c0 + p.seg[0].c1)) p.c3_route = 2;
  }
  if (p.gnx.xchg && p.sk_mode == 0) {            // (checked once, before the split-K recursion)
    static const int n_cu = [] { int d = 0, n = 0; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
    if (!p.gnx.status || !p.gnx.y_hi || !p.gnx.gamma || !p.gnx.beta || gemm_gnx_plan(p, n_cu) <= 0) return hipErrorInvalidValue;
    if (p.gnx.sk_c > 0 && (!p.gnx.sk_x || !p.gnx.sk_stat16 || !p.gnx.sk_y_hi || (x3 && !p.gnx.sk_y_lo) || p.gnx.tscale)) return hipErrorInvalidValue;
  }
  if (p.sk_buf && p.sk_split == 2 && p.sk_ticket && p.sk_mode == 0) {
    p.sk_mode = 3;                                     // both k-halves and the epilogue in one launch
    return launch_gemm(p, precision, st);
  }
  if (p.sk_buf && p.sk_split >= 2 && p.sk_mode == 0) {
    p.sk_mode = 1;
    hipError_t e = launch_gemm(p, precision, st);
    if (e != hipSuccess) return e;
    p.sk_mode = 2;
    return launch_gemm(p, precision, st);
  }
  if (!p.sk_buf || p.sk_split < 2) { p.sk_buf = nullptr; p.sk_split = 0; p.sk_mode = 0; p.sk_ticket = nullptr; }
  // XCD rectangle (k_gemm): split the columns over xn of the 8 XCDs where that lowers what one L2 has to fetch -
  // (N / xn) x K of W plus (M / xm) x channels of A.  DVITS_XCD_N=<1|2|4|8> forces xn where it divides, 0 = row bands.
  const bool c3 = p.c3_route == 1 && (p.sk_mode == 0 || p.sk_mode == 3);
  p.xcd_n = 0;
  if (p.force_tile == GT_AUTO) {
    int bm, bn;
    gemm_pick_tile(p, bm, bn);
    // (the convolution kernels: 64 x 64 tiles laid out per utterance - kernels_conv.hip C3Tile)
    const int tm = c3 ? gemm_conv3_row_tiles(p, 64) : (p.M + bm - 1) / bm, tn = c3 ? p.N / 64 : (p.N + bn - 1) / bn;
    const int xs = (p.sk_mode == 1 || p.sk_mode == 3) ? p.sk_split : 1;
    gemm_xcd_rect(p, tm, tn, xs);
  }
  // stride-1 three-tap convolutions whose input channels fit the LDS: the resident-operand kernel (kernels_conv.hip), on its own
  // per-utterance 64 x 64 tile grid with k_gemm's XCD rectangle / exchange-word layout
  // (the caller - engine.hip conv3_takes - gives the fragment-major weights only where this kernel is the better one)
  if (c3) return launch_conv3(p, st);
  if (p.c3_route == 2 && p.sk_mode == 0) return launch_conv3_up(p, st);
  const GemmTune& tune = gemm_tune();
  const int big_tiles = ((p.M + 127) / 128) * ((p.N + 127) / 128);
  bool k64 = tune.bk64 != 0;
  for (int s = 0; s < p.nseg; ++s) k64 = k64 && p.seg[s].c0 % 64 == 0 && p.seg[s].c1 % 64 == 0;
  const int ft = p.force_tile & 0xff;
  if (ft != GT_AUTO && (p.force_tile & GT_BK64) && !k64) return hipErrorInvalidValue;
  const bool big = ft != GT_AUTO ? ft == GT_T0 : big_tiles >= tune.big;
  const int bk = ft != GT_AUTO ? ((p.force_tile & GT_BK64) ? 64 : 32) : ((!big && k64) ? 64 : 32);
  for (int s = 0; s < p.nseg; ++s) {
    if (p.seg[s].c0 % 32 != 0 || p.seg[s].c1 % 32 != 0) return hipErrorInvalidValue;
    p.seg[s].nkt = p.seg[s].taps * (p.seg[s].c0 + p.seg[s].c1) / bk;
  }
  if (ft != GT_AUTO) return bk == 64 ? Tiles<64>::launch_forced(p, x3, ft, st) : Tiles<32>::launch_forced(p, x3, ft, st);
  if (big) return tune.ksplit ? Tiles<32>::T0::launch(p, x3, st) : Tiles<32>::T0S::launch(p, x3, st);
  if (bk == 64) return Tiles<64>::launch(p, x3, tune.min_wg, tune.ksplit != 0, st);
  return Tiles<32>::launch(p, x3, tune.min_wg > 0 ? tune.min_wg : 1, false, st);
}
