// Feed-forward tail of a transformer block at C = 256 / 384 / 512 for gfx950 (MI355X) as ONE launch: LN3 -> GEGLU -> merged
// ff.net.2 + proj_out + block residual (-> the consumer's GroupNorm), FFSplitParams in dv_common.h.  Reference:
// unet1d/attention.py:189-203 (norm3, ff, residual), :206-255 / :280-301 (FeedForward, GEGLU), transformer_1d.py:300-326.
//
// Why not two GEMMs (rounds 1-4): 31.9 + 25.0 us (C = 256, M = 4096) and 35.4 + 28.8 us (C = 384, M = 2048) at the bench
// shape, 0.59 ms = 19 % of a forward, with the 4C-wide GEGLU product making an HBM round trip between them.  Why not the
// 32-row row-block kernel of the C = 128 blocks (k_chain_ff): a workgroup there streams ALL weights of the block - 3.3 / 7.6 MB
// at these widths - for 32 rows.  This kernel keeps what makes the row-block form fast (the A operand resident in LDS, the
// weights fragment-major straight into registers, no barrier in a k-loop, the product never leaves the CU) and fixes its
// weight traffic twice over:
//   * 64 rows per workgroup - two row fragments per weight fragment, half the weight bytes per MFMA (32 rows at C = 512, where
//     64 rows of h3 alone would fill the LDS, for row pitches that are no multiple of 64, and for mid-size inputs: RF below);
//   * the product columns are SPLIT over `nspl` workgroups per row block (4 at C = 256, 8 at C = 384 / 512: 256 workgroups at the
//     bench shape), and workgroup id = row block * nspl + slice puts slice s on the XCDs x = s (mod nspl) only: an XCD's L2
//     holds 1 / nspl of the weights and every CU of the XCD walks the same 0.8-0.9 MB (1.7 MB at C = 512).
// The price is a reduction over the slices: every workgroup writes its partial ffproj sums through, raises a flag word per wave,
// and finishes C / nspl of the output columns once the flags of the waves that wrote its blocks are up (reduce-scatter; partials summed in slice order:
// deterministic).  That wait needs every workgroup of the launch resident at once - the planner checks (engine.hip), the wait
// is bounded and flagged exactly like the in-launch GroupNorm's (gnx_device.h), and the engine's recovery path is the same.
// Measured (profiles/r05_*): both k-loops run at the MFMA rate (192 + 120 MFMAs x 32 cycles x two waves per SIMD at C = 256);
// per launch 32.5-44 us against 53-64 us for the two GEMMs, +7.8 % on the 50-step run; planned from 96 workgroups (one utterance
// keeps the two GEMMs: there a launch with two in-launch hand-overs costs what the two launches did).
#include "dv_common.h"
#include "dv_device.h"
#include "gnx_device.h"

#include <cstdlib>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#ifdef DV_GEMM_TRACE
// development build only (make trace): per-workgroup s_memtime stamps of the phases (tools/ffsplit_trace.py)
__device__ unsigned long long g_ffs_trace[1024 * 16];
__device__ int g_ffs_sel = 0;                      // which launches stamp: C (0 = any) - set by the tool
#define DV_FTRACE(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024 && (g_ffs_sel == 0 || g_ffs_sel == p.C)) g_ffs_trace[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int dv_debug_ffs_trace_select(int C) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_ffs_sel), &C, sizeof(C));
  void* d = nullptr;
  if (e == hipSuccess) e = hipGetSymbolAddress(&d, HIP_SYMBOL(g_ffs_trace));
  return (int)(e != hipSuccess ? e : hipMemset(d, 0, sizeof(g_ffs_trace)));
}
extern "C" int dv_debug_ffs_trace(unsigned long long* host, int n_wg) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ffs_trace), (size_t)n_wg * 16 * sizeof(unsigned long long));
}
#else
#define DV_FTRACE(i) do {} while (0)
#endif

namespace {

constexpr int NWV = 8, NT = 64 * NWV;
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ unsigned pk(float lo, float hi) { return dv_cvt_pk_bf16(lo, hi); }
struct BFrag { bf16x8 h, l; };

// RF = row fragments per workgroup: 2 (64 rows: C = 256 / 384) or 1 (32 rows: C = 512, where 64 rows of h3 alone would fill the LDS)
template <int C, int NSPL, int RF>
struct FFGeom {
  static constexpr int BM = 32 * RF;
  static constexpr int CHP = BM * 128;                              // bytes of one 64-channel chunk of one plane of a resident operand
  static constexpr int KSA = C / 16;                               // k-steps of stage A (K = C)
  static constexpr int PS = 4 * C / NSPL, UNITS = PS / 32;          // product columns of a slice; 32-column units ([32 a | 32 gate] packed blocks)
  static constexpr int HS = C / NSPL, HKS = HS / 16, PKS = PS / 16; // the slice's share of h3's channels / k-steps of either part of stage B
  static constexpr int KSB = HKS + PKS;                             // k-steps of stage B per workgroup
  static constexpr int NF = C / 32, KSBW = 5 * C / 16;              // output column fragments; k-steps per row of the merged weights
  static constexpr int KGB = NF > NWV ? 2 : 1;                      // k-groups of stage B (C = 384: 4 column groups x 2 k-groups)
  static constexpr int NFW = NF / (NWV / KGB);                      // column fragments per wave in stage B
  static constexpr int NSRC = NSPL;                                 // partial sums per output element (the k-groups of stage B are added through LDS first)
  static constexpr int BNF = C / NSPL, HFT = BNF / 16;              // finishing tile: columns / 16-column blocks
  static constexpr int A_CH = C / 64, A_PL = A_CH * CHP, G_CH = PS / 64, G_PL = G_CH * CHP;
  static constexpr int SMEM = 2 * A_PL + 2 * G_PL;
  static constexpr int NIT = NFW * RF;                              // (fragment, row fragment) items a wave accumulates in stage B
  static_assert(KGB == 1 || (NIT % 2 == 0 && 2 * 4 * (NIT / 2) * 4 * 64 * 16 <= 2 * A_PL), "k-group exchange of stage B fits the h3 region");
  static constexpr int DA = 8, DB = NFW == 3 ? 6 : 8;               // weight units (hi + lo fragment: 8 VGPRs) in flight per wave
  static constexpr int CPG = 8 / (BM / 8);                          // chunks of h3 one round of LDS-DMAs (one instruction per wave and plane) covers
  static constexpr int KS_G1 = CPG * 4;                             // first k-step that reads h3 beyond the first round
  static_assert(UNITS <= NWV && PS % 64 == 0 && HS % 16 == 0 && NF % (NWV / KGB) == 0 && BNF % 16 == 0 && HFT * RF <= NWV && A_CH % CPG == 0 && A_CH > CPG, "geometry");
};

template <int C, int NSPL, int RF>
__global__ __launch_bounds__(NT) void k_ff_split(const FFSplitParams p) {
  using G = FFGeom<C, NSPL, RF>;
  constexpr int BM = G::BM, CHP = G::CHP;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* const a_reg = smem;                                  // h3:      [2 planes][C / 64 chunks][BM rows][128 B]
  char* const g_reg = smem + 2 * G::A_PL;                    // product: [2 planes][PS / 64 chunks][BM rows][128 B]
  constexpr int NB = C / 32;                                 // LayerNorm row partials per row
  __shared__ __attribute__((aligned(1024))) float2 s_rs[BM * NB];           // the rows' raw partials (sum, M2 about the block mean)
  __shared__ __attribute__((aligned(256))) float s_ug[G::UNITS * 64], s_bg[G::UNITS * 64];
  __shared__ GnxShared<64> s_gnx;
  __shared__ __attribute__((aligned(256))) unsigned s_pf[64];
  // (every 64-byte line of the argument block is requested at once: see k_gemm)
  asm volatile("" ::"s"(p.M), "s"(p.rowstat), "s"(p.wg_lo), "s"(p.wm_hi), "s"(p.res), "s"(p.out_lo), "s"(p.flags), "s"(p.gnx.xchg),
               "s"(p.gnx.y_hi), "s"(p.gnx.sk_x), "s"(p.gnx.sk_raw_lo));
  DV_FTRACE(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int rb = (int)blockIdx.x / NSPL, s = (int)blockIdx.x - rb * NSPL;
  const int m0 = rb * BM;
  const unsigned a_base = (unsigned)(size_t)a_reg;

  // ---- prologue: everything the kernel reads before its first hand-over is REQUESTED here, and only the first 64 channels of h3
  //      are WAITED for.  Issue order = queue order (a wave's vector-memory operations return in order):
  //        1. the small LDS-DMAs (LayerNorm vectors, the rows' partials) and this workgroup's share of the L2 prefetch
  //        2. chunk 0 of h3 (two instructions per wave)      <- the first barrier waits for these
  //        3. weight units 0-3 of stage A
  //        4. chunks 1.. of h3                               <- waited for inside the k-loop, before k-step 4 is read
  //        5. weight units 4-7
  //      [v1 waited for all of h3 and sixteen weight fragments per wave - 192 KiB per CU through a path that delivers ~20 B/clk
  //      while every CU of the chip starts cold: 10.6 k of 62 k cycles before the first MFMA.] ----
  constexpr int UA = 2 * G::KSA;
  const bool a_wave = wave < G::UNITS;
  const int unit = s * G::UNITS + (a_wave ? wave : 0);       // packed 64-column block of the GEGLU weights
  // stage A weights: unit U = k-step * 2 + f (f: 0 = the `a` fragment, 1 = the gate fragment of this wave's 32 product columns)
  auto load_a_unit = [&](int U) __attribute__((always_inline)) {
    const int nf = 2 * unit + (U & 1), ks = U >> 1;
    const size_t e = ((size_t)(nf * G::KSA + ks) * 64 + lane) * 8;
    BFrag f;
    f.h = *reinterpret_cast<const bf16x8*>(p.wg_hi + e);
    f.l = *reinterpret_cast<const bf16x8*>(p.wg_lo + e);
    return f;
  };
  BFrag bq[G::DA > G::DB ? G::DA : G::DB];
  static_assert(G::DA == 8, "the counted waits below assume eight weight units in flight");
  // 1. LayerNorm finish vectors of the slice's packed columns and the rows' raw LayerNorm partials: by LDS-DMA as well.  [First
  // version: plain loads -> registers -> LDS, (mean, rstd) per row computed here: two dependent cold round trips at the head of
  // every workgroup, and the compiler's wait for the first also drains every DMA issued before it
  // (profiles/r05_ffsplit_phase_trace_v1.txt).  Nothing in the prologue has a register destination now; the rows' statistics
  // are formed from the LDS copy in the GEGLU epilogue.]
  if (a_wave) {
    glds4(p.ug + (size_t)(s * G::UNITS + wave) * 64 + lane, (unsigned)(size_t)s_ug + wave * 256);
    glds4(p.bg + (size_t)(s * G::UNITS + wave) * 64 + lane, (unsigned)(size_t)s_bg + wave * 256);
  }
  if (wave < BM * NB * 8 / 1024)                              // 64 rows x NB float2, contiguous: 1 KiB per instruction
    glds16(reinterpret_cast<const char*>(p.rowstat) + ((size_t)m0 * NB * 8 + wave * 1024 + lane * 16), (unsigned)(size_t)s_rs + wave * 1024);
  // L2 prefetch: the workgroups of an XCD (id % 8 - a speed assumption only) all read slice s of both weight matrices; each
  // touches a share of its lines at launch (one 128-byte line per lane, LDS-DMA into a scratch word), so the L2 fills with
  // thousands of requests in flight while the compute waves start on their first fragments (see k_chain2)
  {
    constexpr int LG = G::UNITS * 2 * G::KSA * 8, LM = G::NF * G::KSB * 8;    // 128-byte lines per plane of this slice
    const int xw = (int)blockIdx.x >> 3, nxw = ((int)gridDim.x + 7) >> 3;
    const int total = 2 * (LG + LM), per = (total + nxw - 1) / nxw, end = min(total, (xw + 1) * per);
    for (int ln = xw * per + tid; ln < end; ln += NT) {
      const char* src;
      if (ln < 2 * LG) {
        const int pl = ln >= LG, r = ln - pl * LG;
        src = reinterpret_cast<const char*>(pl ? p.wg_lo : p.wg_hi) + ((size_t)s * LG + r) * 128;
      } else {
        const int l2 = ln - 2 * LG, pl = l2 >= LM, r = l2 - pl * LM;
        const int nf = r / (G::KSB * 8), q = r - nf * (G::KSB * 8), kk = q >> 3, sub = q & 7;
        const int ksw = kk < G::HKS ? s * G::HKS + kk : G::KSA + s * G::PKS + (kk - G::HKS);
        src = reinterpret_cast<const char*>(pl ? p.wm_lo : p.wm_hi) + ((size_t)(nf * G::KSBW + ksw) * 8 + sub) * 128;
      }
      glds4(src, (unsigned)(size_t)s_pf);
    }
  }
  // h3 planes -> LDS: instruction = (64-channel chunk, 8 rows) of one plane; one round = one instruction per wave and plane =
  // CPG chunks (64 rows: wave w sends rows 8 w .. 8 w + 7 of one chunk; 32 rows: waves 0-3 / 4-7 send two chunks)
  auto h3_chunk = [&](int cg) __attribute__((always_inline)) {
    const int d_row = lane >> 3, d_slot = lane & 7, r8 = wave % (BM / 8), c = cg * G::CPG + wave / (BM / 8), row = r8 * 8 + d_row;
    const size_t e = (size_t)(m0 + row) * C + c * 64 + ((d_slot ^ swz(row)) << 3);
    const unsigned dst = a_base + (unsigned)(c * CHP + r8 * 1024);
    glds16(p.a_hi + e, dst);
    glds16(p.a_lo + e, dst + G::A_PL);
  };
  h3_chunk(0);                                               // 2.
  if (a_wave) {                                              // 3.
#pragma unroll
    for (int j = 0; j < 4; ++j) bq[j] = load_a_unit(j);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 1; c < G::A_CH / G::CPG; ++c) h3_chunk(c);    // 4.
  constexpr int NC = (G::A_CH / G::CPG - 1) * 2;             // ... LDS-DMA instructions per wave
  if (a_wave) {                                              // 5.
#pragma unroll
    for (int j = 4; j < 8; ++j) bq[j] = load_a_unit(j);
  }
  __builtin_amdgcn_sched_barrier(0);
  DV_FTRACE(1);
  // [v4 put steps 3b-5 BEHIND the barrier (two weight units ahead of it): first MFMA at 8.0 k instead of 9.0 k cycles, but the
  // k-loop then waits for what it had not asked for in time: 13.7 k instead of 12.3 k cycles, nothing gained - the start-up is
  // bound by how fast a cold chip delivers the first ~200 KiB per CU, not by the order of the requests.]
  if (a_wave) wait_vmcnt<8 + NC + 8>();                       // chunk 0 (and everything older) has landed
  else wait_vmcnt<NC>();
  __builtin_amdgcn_s_barrier();
  DV_FTRACE(2);

  // A fragments (both row fragments, both planes) of k-step `ks` of a resident operand
  auto read_frag = [&](const char* reg, int pl_bytes, int ks, bf16x8 (&h)[RF], bf16x8 (&l)[RF]) __attribute__((always_inline)) {
    const int c16 = ks * 2 + lh;
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) {
      const int row = rf * 32 + l31;
      const int off = (c16 >> 3) * CHP + row * 128 + (((c16 & 7) ^ swz(row)) << 4);
      h[rf] = *reinterpret_cast<const bf16x8*>(reg + off);
      l[rf] = *reinterpret_cast<const bf16x8*>(reg + pl_bytes + off);
    }
  };
  auto mfma3 = [&](f32x16& acc, const BFrag& f, const bf16x8& ah, const bf16x8& al) __attribute__((always_inline)) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, al, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.l, ah, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, ah, acc, 0, 0, 0);
  };


  // ================= stage B: partial ffproj over [this slice's channels of h3 | its product columns], all C output columns ==========
  // wave -> NFW column fragments from cf0 and k-group kg (KGB = 2: 4 column groups x 2 halves of the k-steps)
  const int kg = G::KGB == 1 ? 0 : wave >> 2;
  const int cf0 = G::KGB == 1 ? wave : (wave & 3) * G::NFW;
  f32x16 accb[G::NFW][RF];
  auto load_b_unit = [&](int j, int i) __attribute__((always_inline)) {        // k-step j of the workgroup's range, fragment cf0 + i
    const int ksw = j < G::HKS ? s * G::HKS + j : G::KSA + s * G::PKS + (j - G::HKS);
    const size_t e = ((size_t)((cf0 + i) * G::KSBW + ksw) * 64 + lane) * 8;
    BFrag f;
    f.h = *reinterpret_cast<const bf16x8*>(p.wm_hi + e);
    f.l = *reinterpret_cast<const bf16x8*>(p.wm_lo + e);
    return f;
  };
  auto b_prologue = [&](auto kg_tag) __attribute__((always_inline)) {
    constexpr int KG = decltype(kg_tag)::value;
    constexpr int J0 = KG == 0 ? 0 : (G::KSB + 1) / 2;
    constexpr int JN = G::KGB == 1 ? G::KSB : (KG == 0 ? (G::KSB + 1) / 2 : G::KSB / 2);
#pragma unroll
    for (int u = 0; u < G::DB; ++u)
      if (u < JN * G::NFW) bq[u] = load_b_unit(J0 + u / G::NFW, u % G::NFW);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto b_loop = [&](auto kg_tag) __attribute__((always_inline)) {
    constexpr int KG = decltype(kg_tag)::value;
    constexpr int J0 = KG == 0 ? 0 : (G::KSB + 1) / 2;
    constexpr int JN = G::KGB == 1 ? G::KSB : (KG == 0 ? (G::KSB + 1) / 2 : G::KSB / 2);
    constexpr int UB = JN * G::NFW;
    auto read_b = [&](int j, bf16x8 (&h)[RF], bf16x8 (&l)[RF]) __attribute__((always_inline)) {
      if (j < G::HKS) read_frag(a_reg, G::A_PL, s * G::HKS + j, h, l);
      else read_frag(g_reg, G::G_PL, j - G::HKS, h, l);
    };
    bf16x8 ah[2][RF], al[2][RF];
    read_b(J0, ah[0], al[0]);
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int jj = u / G::NFW, i = u % G::NFW, cur = jj & 1;
      if (i == 0 && jj + 1 < JN) read_b(J0 + jj + 1, ah[cur ^ 1], al[cur ^ 1]);
      const BFrag w = bq[u % G::DB];
      // (the row fragments alternate: consecutive MFMAs write different accumulators)
#pragma unroll
      for (int rf = 0; rf < RF; ++rf) accb[i][rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, al[cur][rf], accb[i][rf], 0, 0, 0);
#pragma unroll
      for (int rf = 0; rf < RF; ++rf) accb[i][rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.l, ah[cur][rf], accb[i][rf], 0, 0, 0);
#pragma unroll
      for (int rf = 0; rf < RF; ++rf) accb[i][rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, ah[cur][rf], accb[i][rf], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (u + G::DB < UB) bq[u % G::DB] = load_b_unit(J0 + (u + G::DB) / G::NFW, (u + G::DB) % G::NFW);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  const bool fin = wave < G::HFT * RF;
  const int f_rf = wave % RF, f_hfl = wave / RF, f_hf = s * G::HFT + f_hfl;   // finishing unit: row fragment, 16-column block of the output (tile-local / global)
  const int f_m = m0 + f_rf * 32 + l31;
  float4 rres[2], rbias[2];
  // The first weight fragments of stage B and the finishing waves' bias / residual rows (cold: written a whole transformer block
  // ago): requested behind a wave's GEGLU epilogue, ahead of the barrier that waits for everybody's
  auto early_requests = [&]() __attribute__((always_inline)) {
    if (kg == 0) b_prologue(std::integral_constant<int, 0>{});
    else b_prologue(std::integral_constant<int, 1>{});
    if (fin) {
      const float* rp = p.res + (size_t)f_m * C + f_hf * 16 + 4 * lh;
      rres[0] = *reinterpret_cast<const float4*>(rp); rres[1] = *reinterpret_cast<const float4*>(rp + 8);
      rbias[0] = *reinterpret_cast<const float4*>(p.bm + f_hf * 16 + 4 * lh); rbias[1] = *reinterpret_cast<const float4*>(p.bm + f_hf * 16 + 4 * lh + 8);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // [Requested BEFORE the GEGLU epilogue (C = 256, v2): the epilogue grew by 2.5 k cycles - 249 VGPRs - and the barrier behind it is
  // the wait for the slowest wave either way: nothing gained.]
  if (!a_wave) {       // (C = 384: waves 6 and 7 have no stage-A unit; they meet the others at the in-loop barrier)
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    early_requests();
  }

  // ================= stage A: GEGLU of this wave's 32 product columns, both row fragments =================
  if (a_wave) {
    f32x16 acc[2][RF];                                       // [a | gate][row fragment]
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int rf = 0; rf < RF; ++rf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][rf][r] = 0.f;
    bf16x8 ah[2][RF], al[2][RF];
    read_frag(a_reg, G::A_PL, 0, ah[0], al[0]);
#pragma unroll
    for (int U = 0; U < UA; ++U) {
      const int ks = U >> 1, f = U & 1, cur = ks & 1;
      if (U == 2 * (G::KS_G1 - 1) && G::A_CH > G::CPG) {   // the k-step read ahead below is the first beyond the first round of h3 DMAs: the rest has landed - here and in every wave
        wait_vmcnt<16>();  // (younger than the DMAs: the eight weight units in flight, U .. U + 7)
        __builtin_amdgcn_s_barrier();
      }
      if (f == 0 && ks + 1 < G::KSA) read_frag(a_reg, G::A_PL, ks + 1, ah[cur ^ 1], al[cur ^ 1]);
      const BFrag w = bq[U % G::DA];
      // (the row fragments alternate: consecutive MFMAs write different accumulators)
#pragma unroll
      for (int rf = 0; rf < RF; ++rf) acc[f][rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, al[cur][rf], acc[f][rf], 0, 0, 0);
#pragma unroll
      for (int rf = 0; rf < RF; ++rf) acc[f][rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.l, ah[cur][rf], acc[f][rf], 0, 0, 0);
#pragma unroll
      for (int rf = 0; rf < RF; ++rf) acc[f][rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, ah[cur][rf], acc[f][rf], 0, 0, 0);
      // pinned: without the scheduling barriers hipcc sinks every prefetch down to its use (load -> vmcnt(0) -> MFMA)
      __builtin_amdgcn_sched_barrier(0);
      if (U + G::DA < UA) bq[U % G::DA] = load_a_unit(U + G::DA);
      __builtin_amdgcn_sched_barrier(0);
    }
    DV_FTRACE(3);
    // LayerNorm finish + bias, a * gelu(gate) -> product columns wave * 32 + (8g + 4lh + e) of the slice as split planes in LDS
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) {
      const int row = rf * 32 + l31;
      float2 st;
      {   // (mean, rstd) of this lane's row from the producer's partials per 32-column block, parallel-variance form (gemm_tile.h)
        float4 v[NB / 2];
#pragma unroll
        for (int k = 0; k < NB / 2; ++k) v[k] = *reinterpret_cast<const float4*>(s_rs + row * NB + 2 * k);
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < NB / 2; ++k) { s1 += v[k].x; s1 += v[k].z; }
        const float inv_c = 1.0f / (float)C, mean = s1 * inv_c;
        float m2 = 0.f;
#pragma unroll
        for (int k = 0; k < NB / 2; ++k) {
          const float d0 = v[k].x * (1.0f / 32.0f) - mean, d1 = v[k].z * (1.0f / 32.0f) - mean;
          m2 += v[k].y + 32.0f * d0 * d0;
          m2 += v[k].w + 32.0f * d1 * d1;
        }
        st = make_float2(mean, 1.0f / sqrtf(m2 * inv_c + p.ln_eps));
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cl = 8 * g + 4 * lh;
        const float4 ua = *reinterpret_cast<const float4*>(s_ug + wave * 64 + cl), ugt = *reinterpret_cast<const float4*>(s_ug + wave * 64 + 32 + cl);
        const float4 ba = *reinterpret_cast<const float4*>(s_bg + wave * 64 + cl), bgt = *reinterpret_cast<const float4*>(s_bg + wave * 64 + 32 + cl);
        const float ua_[4] = {ua.x, ua.y, ua.z, ua.w}, ug_[4] = {ugt.x, ugt.y, ugt.z, ugt.w};
        const float ba_[4] = {ba.x, ba.y, ba.z, ba.w}, bg_[4] = {bgt.x, bgt.y, bgt.z, bgt.w};
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = st.y * (acc[0][rf][4 * g + e] - st.x * ua_[e]) + ba_[e];
          const float gt = st.y * (acc[1][rf][4 * g + e] - st.x * ug_[e]) + bg_[e];
          v[e] = a * gelu_erf(gt);
        }
        uint2 hw, lw;
        hw.x = pk(v[0], v[1]); hw.y = pk(v[2], v[3]);
        lw.x = pk(v[0] - __uint_as_float(hw.x << 16), v[1] - __uint_as_float(hw.x & 0xffff0000u));
        lw.y = pk(v[2] - __uint_as_float(hw.y << 16), v[3] - __uint_as_float(hw.y & 0xffff0000u));
        const int n = wave * 32 + cl, c = n >> 6, s16 = (n & 63) >> 3;
        const int off = c * CHP + row * 128 + ((s16 ^ swz(row)) << 4) + ((n & 7) >> 2) * 8;
        *reinterpret_cast<uint2*>(g_reg + off) = hw;
        *reinterpret_cast<uint2*>(g_reg + G::G_PL + off) = lw;
      }
    }
  }
  if (a_wave) early_requests();
  DV_FTRACE(4);
#pragma unroll
  for (int i = 0; i < G::NFW; ++i)
#pragma unroll
    for (int rf = 0; rf < RF; ++rf)
#pragma unroll
      for (int r = 0; r < 16; ++r) accb[i][rf][r] = 0.f;
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();                                           // product planes complete
  DV_FTRACE(5);
  if (kg == 0) b_loop(std::integral_constant<int, 0>{});
  else b_loop(std::integral_constant<int, 1>{});
  DV_FTRACE(6);

  // ================= hand-over: partial sums written through, flag, wait for the row block, finish C / nspl columns =================
  // xbuf (float4 units): [row block][16-column block hf][source][row fragment][gg][64 lanes]; register group g of column fragment cf
  // is 16-column block 2 cf + (g >> 1), half gg = g & 1 (columns 8 gg + 4 lh + e of the block)
  float4* const xb = reinterpret_cast<float4*>(p.xbuf) + (size_t)rb * (2 * G::NF) * G::NSRC * RF * 2 * 64 + lane;
  auto xslot = [&](int hf, int rf, int gg) { return xb + ((size_t)((hf * G::NSRC + s) * RF + rf) * 2 + gg) * 64; };
  if constexpr (G::KGB == 2) {
    // two k-groups (C = 384, 512): added through LDS first.  Item q = fragment * RF + row fragment BELONGS to k-group q & 1: each group
    // hands the other's items over and keeps its own, so both directions cross the LDS at once and every wave writes half of
    // the partial sums (first version: both groups wrote everything through and the finishing waves summed 16 sources - 25 k of 88 k cycles)
    __syncthreads();                                         // every wave is done reading the resident operands
    constexpr int HO = G::NIT / 2;                           // items handed over per wave
    float4* red4 = reinterpret_cast<float4*>(a_reg) + lane;  // [writer's k-group][column group][item slot q >> 1][g][64 lanes]
#pragma unroll
    for (int i = 0; i < G::NFW; ++i)
#pragma unroll
      for (int rf = 0; rf < RF; ++rf) {
        const int q = i * RF + rf;
        // (a wave-uniform branch per item, not a run-time index: an indexed register array lives in scratch memory)
        if ((q & 1) != kg) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            red4[(size_t)(((kg * 4 + (wave & 3)) * HO + (q >> 1)) * 4 + g) * 64] =
                make_float4(accb[i][rf][4 * g], accb[i][rf][4 * g + 1], accb[i][rf][4 * g + 2], accb[i][rf][4 * g + 3]);
        }
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < G::NFW; ++i)
#pragma unroll
      for (int rf = 0; rf < RF; ++rf) {
        const int q = i * RF + rf;
        if ((q & 1) == kg) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 v = red4[(size_t)((((kg ^ 1) * 4 + (wave & 3)) * HO + (q >> 1)) * 4 + g) * 64];
            float4 a = make_float4(accb[i][rf][4 * g], accb[i][rf][4 * g + 1], accb[i][rf][4 * g + 2], accb[i][rf][4 * g + 3]);
            // (k-group 0's sums first, whichever group adds)
            a = kg == 0 ? make_float4(a.x + v.x, a.y + v.y, a.z + v.z, a.w + v.w) : make_float4(v.x + a.x, v.y + a.y, v.z + a.z, v.w + a.w);
            st_handover16(xslot(2 * (cf0 + i) + (g >> 1), rf, g & 1), a);
          }
        }
      }
    wait_vmcnt<0>();
  } else {
#pragma unroll
    for (int i = 0; i < G::NFW; ++i)
#pragma unroll
      for (int rf = 0; rf < RF; ++rf)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          st_handover16(xslot(2 * (cf0 + i) + (g >> 1), rf, g & 1),
                        make_float4(accb[i][rf][4 * g], accb[i][rf][4 * g + 1], accb[i][rf][4 * g + 2], accb[i][rf][4 * g + 3]));
    wait_vmcnt<0>();                                         // this thread's partial sums have been written through
  }
  // one flag word per WAVE: a finishing wave needs the partial sums of exactly one wave of every workgroup of its row block (the one
  // that multiplied its column fragment) - it waits for those four / eight words, not for whole workgroups, and no barrier sits
  // between a wave's last store and its flag [v3: one flag per workgroup behind a __syncthreads]
  if (lane == 0) __hip_atomic_store(p.flags + ((size_t)rb * NSPL + s) * NWV + wave, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  DV_FTRACE(7);
  const int b_item = m0 / p.T, Tv = p.Tv > 0 ? p.Tv : p.T;
  const int t_row = f_m - b_item * p.T;
  const bool m_ok = t_row < Tv;                              // a frame that exists (padded row spaces: dv_common.h)
  float vv[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) vv[r] = 0.f;
  if (fin) {
    // wait for the flags of this row block (every wave polls for itself: no barrier behind the wait); bounded and flagged
    {
      // (the wave of workgroup `lane` that wrote this unit's block: column fragment f_hf / 2 - and, with two k-groups, the group
      // that owns the item (fragment, row fragment))
      const int cfw = f_hf >> 1, ww = G::KGB == 1 ? cfw : ((((cfw % G::NFW) * RF + f_rf) & 1) * 4 + cfw / G::NFW);
      const unsigned long long* fl = p.flags + (size_t)rb * NSPL * NWV + ww;
      for (int spins = 0;; ++spins) {
        bool ok = true;
        if (lane < NSPL) ok = __hip_atomic_load(fl + lane * NWV, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != ~0ull;
        if (__all(ok)) break;
        const bool lost = (spins & 63) == 63 && __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
        if (lost) break;
        if (spins > p.spin_max) {
          if (lane == 0) {
            p.status[1] = (unsigned)(size_t)p.flags; p.status[2] = blockIdx.x; p.status[3] = 0xff5u; p.status[4] = (unsigned)__builtin_popcountll(__ballot(!ok));
            __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");
    }
    DV_FTRACE(8);
    // partial sums of this unit's 32 x 16 block, in source order (slice, k-group): the same bits on every run
    const float4* src0 = xb + (size_t)(f_hf * G::NSRC * RF + f_rf) * 2 * 64;
    constexpr int BATCH = G::NSRC < 8 ? G::NSRC : 8;
#pragma unroll
    for (int s0 = 0; s0 < G::NSRC; s0 += BATCH) {
      float4 t[BATCH][2];
#pragma unroll
      for (int k = 0; k < BATCH; ++k)
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) t[k][gg] = ld_handover16(src0 + ((size_t)(s0 + k) * RF * 2 + gg) * 64);
#pragma unroll
      for (int k = 0; k < BATCH; ++k)
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          vv[4 * gg] += t[k][gg].x; vv[4 * gg + 1] += t[k][gg].y; vv[4 * gg + 2] += t[k][gg].z; vv[4 * gg + 3] += t[k][gg].w;
        }
    }
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
      vv[4 * gg] += rbias[gg].x + rres[gg].x; vv[4 * gg + 1] += rbias[gg].y + rres[gg].y;
      vv[4 * gg + 2] += rbias[gg].z + rres[gg].z; vv[4 * gg + 3] += rbias[gg].w + rres[gg].w;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) vv[r] = m_ok ? vv[r] : 0.f;     // (padding rows are stored as zeros)
    DV_FTRACE(9);
    if (p.stats16) {   // this unit's 32 x 16 block: (sum, squared deviations about its own mean) over the frames that exist
      const int cnt = min(32, Tv - (m0 + f_rf * 32 - b_item * p.T));
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) a1 += vv[r];
      a1 = wave_sum64(a1);
      const float mb = cnt == 32 ? a1 * (1.0f / 512.0f) : a1 / (float)(16 * cnt);
#pragma unroll
      for (int r = 0; r < 8; ++r) { const float dv = m_ok ? vv[r] - mb : 0.f; a2 = fmaf(dv, dv, a2); }
      a2 = wave_sum64(a2);
      if (lane == 0) {
        const size_t e = (size_t)((m0 >> 5) + f_rf) * (C >> 4) + f_hf;
        reinterpret_cast<float2*>(p.stats16)[e] = make_float2(a1, a2);
        if (p.gnx.xchg)    // for the other workgroups of the launch: one 8-byte word (sum, M2), written through
          __hip_atomic_store(p.gnx.xchg + e, (unsigned long long)__float_as_uint(a1) | ((unsigned long long)__float_as_uint(a2) << 32),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    // (the statistics go out FIRST: other workgroups wait for them, nobody waits for these stores)
    const size_t ob = (size_t)f_m * C + f_hf * 16;
    if (p.out) {
#pragma unroll
      for (int gg = 0; gg < 2; ++gg)
        dv_st16(p.out + ob + 4 * lh + 8 * gg, make_float4(vv[4 * gg], vv[4 * gg + 1], vv[4 * gg + 2], vv[4 * gg + 3]));
    }
    if (p.out_hi) store_planes8(p.out_hi, p.out_lo, ob, lh, vv);
  }
  DV_FTRACE(10);
  if (p.gnx.xchg) {
    // ---- the consumer's GroupNorm (+ SiLU) of this block's output, a concatenated skip tensor included (gnx_device.h): the
    //      workgroup's finishing tile [64 rows x C / nspl columns] is treated exactly like a GEMM tile ----
    GnxTile t;
    t.M = p.M; t.N = C; t.T_out = p.T; t.Tv_out = Tv; t.m0 = m0; t.n0 = s * G::BNF; t.bm = BM; t.bn = G::BNF; t.bq = b_item;
    gnx_finish_table<64>(p.gnx, t, s_gnx, tid, lane, wave, NWV, [](int) {});
    if (fin) {
      const int cl0 = f_hfl * 16 + 4 * lh;                   // tile-local column of gg = 0, e = 0
      float y[8];
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const float4 sa = *reinterpret_cast<const float4*>(s_gnx.gA + cl0 + 8 * gg);
        const float4 sb = *reinterpret_cast<const float4*>(s_gnx.gB + cl0 + 8 * gg);
        y[4 * gg] = fmaf(vv[4 * gg], sa.x, sb.x); y[4 * gg + 1] = fmaf(vv[4 * gg + 1], sa.y, sb.y);
        y[4 * gg + 2] = fmaf(vv[4 * gg + 2], sa.z, sb.z); y[4 * gg + 3] = fmaf(vv[4 * gg + 3], sa.w, sb.w);
      }
      if (p.gnx.silu) {
#pragma unroll
        for (int r = 0; r < 8; ++r) y[r] = y[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-y[r]));
      }
      store_planes8(p.gnx.y_hi, p.gnx.y_lo, (size_t)f_m * C + f_hf * 16, lh, y);
    }
  }
  DV_FTRACE(11);
}

template <int C, int NSPL, int RF>
hipError_t ffs_init_one() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_ff_split<C, NSPL, RF>), hipFuncAttributeMaxDynamicSharedMemorySize, FFGeom<C, NSPL, RF>::SMEM);
}
template <int C, int NSPL, int RF>
hipError_t ffs_launch_one(const FFSplitParams& p, hipStream_t st) {
  constexpr int smem = FFGeom<C, NSPL, RF>::SMEM;
  hipLaunchKernelGGL((k_ff_split<C, NSPL, RF>), dim3((p.M / (32 * RF)) * NSPL), dim3(NT), smem, st, p);
  return hipGetLastError();
}

}  // namespace

hipError_t ff_split_init() {
  hipError_t e = ffs_init_one<256, 4, 2>();
  if (e == hipSuccess) e = ffs_init_one<384, 8, 2>();
  if (e == hipSuccess) e = ffs_init_one<256, 4, 1>();
  return e != hipSuccess ? e : ffs_init_one<512, 8, 1>();
}

// Rows per workgroup.  64 (two row fragments per weight fragment) wherever that fills the chip; 32 at C = 512 (64 rows of h3 alone
// - 128 KiB of split planes - would leave no room for the product), for row pitches that are no multiple of 64 (utterances of any
// length: 32-frame padding), and for small inputs, where twice the workgroups matter more than half the weight bytes per MFMA
// (one utterance of 300 frames: 5 / 3 / 2 row blocks of 32 at the three levels)
int ff_split_rows(int C, int M, int T, int nspl, int n_cu) {
  if (C == 384) return 64;            // (three column fragments per wave and k-group in stage B: no even split of the items at 32 rows)
  if (C == 512 || T % 64 != 0) return 32;
  return (M / 32) * nspl * 2 <= n_cu ? 32 : 64;
}

bool ff_split_supported(const FFSplitParams& p, int precision) {
  if (precision != 0) return false;                                  // split-bf16 mode only
  if (!((p.C == 256 && p.nspl == 4) || (p.C == 384 && p.nspl == 8) || (p.C == 512 && p.nspl == 8))) return false;
  if (!(p.rows == 32 && p.C != 384) && !(p.rows == 64 && p.C != 512)) return false;
  if (p.M % p.rows != 0 || p.T % p.rows != 0 || p.M % p.T != 0) return false;   // (a row block never spans two utterances)
  if (p.Tv < 0 || p.Tv > p.T || (p.Tv > 0 && p.Tv <= p.T - 32)) return false;
  return true;
}
size_t ff_split_xbuf_floats(int M, int C, int nspl, int rows) {
  return (size_t)(M / rows) * (2 * (C / 32)) * nspl * (rows / 32) * 2 * 64 * 4;
}
// In-launch GroupNorm of the output: the finishing tiles ([rows x C / nspl columns], all resident) behave like GEMM tiles
// (the conditions of gemm_gnx_plan, kernels_gemm.hip)
int ff_split_gnx_plan(const FFSplitParams& p, int n_cu) {
  const GnxParams& gx = p.gnx;
  if (!p.stats16 || gx.groups <= 0 || gx.groups > 64 || gx.sk_c < 0 || gx.sk_c % 16 != 0 || gx.tscale) return 0;
  if ((p.C + gx.sk_c) % gx.groups != 0) return 0;
  const int cpg = (p.C + gx.sk_c) / gx.groups;
  // (workgroup id = row block * nspl + slice, plain order: an utterance's workgroups are consecutive ids, spread evenly over the
  // XCDs - with the launch-by-rounds rule of gemm_gnx_plan the bound is one utterance's workgroups, not the grid)
  const int wg_wait = gemm_handover_rounds() ? (p.T / p.rows) * p.nspl : (p.M / p.rows) * p.nspl;
  if (cpg % 16 != 0 || (p.T / 32) * (cpg / 16) > 256 || wg_wait > n_cu) return 0;
  if (gx.sk_c > 0 && ((gx.sk_c / 16 + p.nspl - 1) / p.nspl) * 16 > DV_GSK) return 0;   // skip slice per workgroup
  return (p.M / 32) * (p.C / 16);
}

hipError_t launch_ff_split(const FFSplitParams& p, int precision, hipStream_t st) {
  if (!ff_split_supported(p, precision)) return hipErrorInvalidValue;
  if (!p.a_hi || !p.a_lo || !p.rowstat || !p.wg_hi || !p.wg_lo || !p.bg || !p.ug || !p.wm_hi || !p.wm_lo || !p.bm || !p.res || !p.xbuf ||
      !p.flags || !p.status || (p.out_hi && !p.out_lo))
    return hipErrorInvalidValue;
  static const int n_cu = [] { int d = 0, n = 0; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
  // (the partial-sum hand-over waits for the nspl <= 8 workgroups of its own row block - consecutive ids; kernels_gemm.hip
  // gemm_handover_rounds - else every workgroup of the launch resident at once)
  if (!gemm_handover_rounds() && (p.M / p.rows) * p.nspl > n_cu) return hipErrorInvalidValue;
  if (p.gnx.xchg) {
    if (!p.gnx.status || !p.gnx.y_hi || !p.gnx.y_lo || !p.gnx.gamma || !p.gnx.beta || ff_split_gnx_plan(p, n_cu) <= 0) return hipErrorInvalidValue;
    if (p.gnx.sk_c > 0 && (!p.gnx.sk_x || !p.gnx.sk_stat16 || !p.gnx.sk_y_hi || !p.gnx.sk_y_lo)) return hipErrorInvalidValue;
  } else if (!p.out) return hipErrorInvalidValue;
  if (p.C == 256) return p.rows == 64 ? ffs_launch_one<256, 4, 2>(p, st) : ffs_launch_one<256, 4, 1>(p, st);
  if (p.C == 384) return ffs_launch_one<384, 8, 2>(p, st);
  return ffs_launch_one<512, 8, 1>(p, st);
}
