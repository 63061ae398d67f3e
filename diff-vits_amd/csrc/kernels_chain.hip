// Row-block chain kernel for gfx950 (MI355X): two dependent K = C contractions of a transformer block in one launch
// (dv_common.h ChainParams).  The denoiser's K = C GEMMs (M = 8192..2048 rows, N = K = 128..384) take 12-15 us each as
// separate launches although their MFMA work is ~1 us: argument fetch, cold first loads, a 2-6 tile k-loop and a
// store-bound epilogue per launch, and every intermediate makes an HBM/L2 round trip.  Here a workgroup owns 32 rows and
// ALL channels: the A operand is resident in LDS, the first GEMM's result goes back into the same LDS region as split
// planes (+ LayerNorm row partials), the second GEMM reads it from there.
//
// Only the weights stream, and they never touch LDS: they are stored FRAGMENT-MAJOR (k_relayout_frag: the 64 lanes'
// 16-byte MFMA operand pieces of one (32-row, 16-k) fragment are 1 KiB contiguous), every wave loads exactly the
// fragments it multiplies - one fully coalesced global_load_dwordx4 per fragment plane, DEPTH fragments ahead in
// registers - so the k-loop has no barrier at all and the eight waves of a workgroup drift freely.  [First version: the
// weight tiles went through a three-stage LDS-DMA ring with one barrier per [128 x 64] tile: ~1900 cycles per 32 KiB tile,
// bound by the DMA latency with only two tiles in flight beside the resident A operand - profiles/r02_ops_profile_chain_v1.txt.]
//
//   waves      : 8 = 4 column-fragment owners (fragments wn, wn + 4, ... of the N = C output columns) x 2 k-groups (each
//                multiplies one half of K); accumulators of all its fragments stay live (NS = C / 128 per wave)
//   order      : 16-deep k-step outer, fragment inner (the A fragment of a k-step is read from LDS once)
//   after GEMM : k-group 1 hands its accumulators over through LDS, k-group 0 runs the epilogue; the next GEMM's first
//                weight fragments are already in flight
#include "dv_common.h"
#include "dv_device.h"
#include "gnx_device.h"

#include <cstdlib>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#ifdef DV_GEMM_TRACE
// development build only (make trace): per-workgroup s_memtime stamps of the chain kernel's phases (tools/chain_trace.py)
__device__ unsigned long long g_chain_trace[8192 * 16];
// which launches stamp: C (0 = any), amode (-1 = any), with the cross attention inside (-1 = any) - set by the tool
__device__ int g_chain_sel[3] = {0, -1, -1};
#define DV_CTRACE(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192 && (g_chain_sel[0] == 0 || g_chain_sel[0] == p.C) && \
    (g_chain_sel[1] < 0 || g_chain_sel[1] == p.amode) && (g_chain_sel[2] < 0 || g_chain_sel[2] == (p.xa_kf_hi != nullptr))) \
    g_chain_trace[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int dv_debug_chain_trace_select(int C, int amode, int xa) {
  const int h[3] = {C, amode, xa};
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_chain_sel), h, sizeof(h));
  void* d = nullptr;
  if (e == hipSuccess) e = hipGetSymbolAddress(&d, HIP_SYMBOL(g_chain_trace));
  return (int)(e != hipSuccess ? e : hipMemset(d, 0, sizeof(g_chain_trace)));
}
extern "C" int dv_debug_chain_trace(unsigned long long* host, int n_wg) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_chain_trace), (size_t)n_wg * 16 * sizeof(unsigned long long));
}
#else
#define DV_CTRACE(i) do {} while (0)
#endif

namespace {

constexpr int BM = 32, NWV = 8, NT = 64 * NWV, NT_LAUNCH = NT + 64;   // + one L2-prefetch wave
constexpr int CHUNK_PL = BM * 128;                 // one 64-channel chunk of the resident A operand, one plane
constexpr int DEPTH = 6;                           // weight fragments (hi + lo: 8 VGPRs each) in flight per wave
#ifndef DV_DEPTH_FF
#define DV_DEPTH_FF 6
#endif
constexpr int DEPTH_FF = DV_DEPTH_FF;              // k_chain_ff (one accumulator fragment per wave: registers to spare)

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ unsigned pk(float lo, float hi) { return dv_cvt_pk_bf16(lo, hi); }

#ifndef DV_CHAIN_PFMODE
// 0: ninth (L2-prefetch) wave wherever round 2 had it (some instantiations then spill at the 168-VGPR cap of a 576-thread
// workgroup); 1: ninth wave only where the kernel fits 168 VGPRs, no prefetch elsewhere; 2: like 1, and the instantiations
// without a ninth wave issue the prefetch from their eight compute waves (behind their first operand loads)
#define DV_CHAIN_PFMODE 2
#endif
// Does this instantiation run the ninth wave?  A 576-thread workgroup puts three waves on one SIMD: 168 VGPRs per lane.
// The cross-attention variants (~230 VGPRs) and every instantiation that spilled at 168 (amode 1 at C = 256, everything at
// C = 384: 12-144 bytes of scratch per lane, tools/kernel_resources.py) run eight waves with the full 256.
template <int NS, int AMODE, bool XA, bool SA, bool CS>
constexpr bool chain_pf_wave() {
  if (XA) return false;
  if (DV_CHAIN_PFMODE == 0) return !(SA && NS >= 3);
  return NS == 1 || (NS == 2 && AMODE == 0);
}

template <int NS, int AMODE, bool XA = false, bool SA = false, bool CS = false>
__global__ __launch_bounds__((chain_pf_wave<NS, AMODE, XA, SA, CS>() ? NT_LAUNCH : NT)) void k_chain2(const ChainParams p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int CH = 2 * NS;                       // 64-channel chunks of the A operand
  constexpr int A_PL = CH * CHUNK_PL;              // bytes per plane of the resident A operand
  constexpr int C = 128 * NS;
  constexpr bool PFW = chain_pf_wave<NS, AMODE, XA, SA, CS>();
  char* const a_reg = smem;                        // [2 planes][CH][32 rows][128 B]
  char* const red_reg = smem + 2 * A_PL;           // k-group hand-over (NS * 16 KiB); GroupNorm entries before stage 1
  __shared__ float2 s_rowp[BM][16];                // LayerNorm row partials per 32-column block (sum, M2 about the block mean)
  __shared__ __attribute__((aligned(16))) float2 s_ln[BM];   // per row (mean, rstd)
  __shared__ __attribute__((aligned(16))) float s_gscale[AMODE ? C : 4], s_gshift[AMODE ? C : 4];
  __shared__ __attribute__((aligned(256))) unsigned s_pf[64];

  // (every 64-byte line of the argument block is requested at once: see k_gemm)
  asm volatile("" ::"s"(p.M), "s"(p.gamma), "s"(p.w1_lo), "s"(p.out1), "s"(p.u2), "s"(p.xa_kf_hi), "s"(p.xa_bias), "s"(p.w3_lo),
               "s"(p.out3_lo), "s"(p.sa_vf_lo), "s"(p.nsplit));
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- L2 prefetch ----
  // Every workgroup streams ALL weights, the workgroups of an XCD walk the same addresses at the same pace, and an L2 does
  // not survive a kernel boundary: without help every fragment load of every wave waits for an HBM / fabric fill
  // (measured 40-50 GB/s per CU).  The workgroups of an XCD (observed placement: block b on XCD b % 8 - a speed
  // assumption only) each touch one slice of the weight planes at launch, one 128-byte line per lane through LDS-DMA into
  // a scratch word (no VGPR destination, the wave's own vmcnt), so the L2 fills with thousands of requests in flight
  // while the compute waves start on their first fragments.  Issued by a ninth wave that then leaves (instantiations that
  // fit the 168 VGPRs a 576-thread workgroup allows), or shared out over the eight compute waves (w of nw) behind their
  // first operand loads.
  auto l2_prefetch = [&](int w, int nw) __attribute__((always_inline)) {
    const int xw = blockIdx.x >> 3, nxw = (gridDim.x + 7) >> 3;
    const int l1 = C * C / 64, l2 = p.passes * l1, l3 = XA ? l1 : 0;   // 128-byte lines per plane of stage 1 / 2 / 3
    const int total = 2 * (l1 + l2 + l3), per = (total + nxw - 1) / nxw;
    const int end = min(total, (xw + 1) * per);
    for (int ln = xw * per + w * 64 + lane; ln < end; ln += 64 * nw) {
      const char* src;
      if (ln < l1) src = reinterpret_cast<const char*>(p.w1_hi) + (size_t)ln * 128;
      else if (ln < 2 * l1) src = reinterpret_cast<const char*>(p.w1_lo) + (size_t)(ln - l1) * 128;
      else if (ln < 2 * l1 + l2) src = reinterpret_cast<const char*>(p.w2_hi) + (size_t)(ln - 2 * l1) * 128;
      else if (ln < 2 * l1 + 2 * l2) src = reinterpret_cast<const char*>(p.w2_lo) + (size_t)(ln - 2 * l1 - l2) * 128;
      else if (ln < 2 * l1 + 2 * l2 + l3) src = reinterpret_cast<const char*>(p.w3_hi) + (size_t)(ln - 2 * l1 - 2 * l2) * 128;
      else src = reinterpret_cast<const char*>(p.w3_lo) + (size_t)(ln - 2 * l1 - 2 * l2 - l3) * 128;
      glds4(src, (unsigned)(size_t)s_pf);
    }
    if (p.res) {                                   // this workgroup's residual rows (read by the first epilogue)
      const char* r0 = reinterpret_cast<const char*>(p.res + (size_t)(blockIdx.x / (p.nsplit > 1 && AMODE == 1 && !XA ? p.nsplit : 1)) * BM * C);
      for (int ln = w * 64 + lane; ln < BM * C / 32; ln += 64 * nw) glds4(r0 + (size_t)ln * 128, (unsigned)(size_t)s_pf);
    }
    // the epilogues' bias / LayerNorm-u vectors (their first touch would be a dependent cold miss inside the epilogue)
    if (w == 0) {
      for (int ln = lane; ln < C / 32; ln += 64) glds4(reinterpret_cast<const char*>(p.b1) + (size_t)ln * 128, (unsigned)(size_t)s_pf);
      for (int ln = lane; ln < p.passes * C / 32; ln += 64) {
        glds4(reinterpret_cast<const char*>(p.b2) + (size_t)ln * 128, (unsigned)(size_t)s_pf);
        glds4(reinterpret_cast<const char*>(p.u2) + (size_t)ln * 128, (unsigned)(size_t)s_pf);
      }
    }
  };
  if (PFW && wave == NWV) { l2_prefetch(0, 1); return; }
  DV_CTRACE(0);
  const int wn = wave & 3, kg = wave >> 2, l31 = lane & 31, lh = lane >> 5;
  // nsplit workgroups per row block (amode 1 with several stage-2 passes, few row blocks): each repeats stage 1 (cheap:
  // the kernel is bound by the weight stream, not by MFMA) and runs its share of the passes - a third of the q | k | v
  // weights per workgroup on three times the CUs.  Part 0 writes out1.
  // CS (amode 0, one stage-2 pass): the parts share out the stage-2 COLUMNS instead - part k computes fragment group k
  // (128 columns) of the second GEMM.
  const int nsp = (((AMODE == 1 && !XA) || CS) && p.nsplit > 1) ? p.nsplit : 1;
  const int rb = nsp == 1 ? (int)blockIdx.x : (int)blockIdx.x / nsp, part = (int)blockIdx.x - rb * nsp;
  const int npass_all = SA ? 3 : p.passes;
  const int ps_lo = CS ? 0 : (part * npass_all + nsp - 1) / nsp, ps_hi = CS ? npass_all : ((part + 1) * npass_all + nsp - 1) / nsp;
  const int m0 = rb * BM;
  const unsigned a_base = (unsigned)(size_t)a_reg;
  const int d_row = lane >> 3, d_slot = lane & 7;

  // ---- one GEMM stage over the resident A: acc[ns] += W[frag0 + ns*4 + wn][k-half kg] x A^T ----
  // unit u = (k-step u / NS of this wave's k-half, fragment u % NS); its two operand planes are one coalesced 16-byte
  // load per lane each from the fragment-major weights
  constexpr int KSTEPS = C / 16, KH = KSTEPS / 2, U = NS * KH;
  struct BFrag { bf16x8 h, l; };
  // (kh_tag / ks0: the stage's k-range is k-steps [ks0, ks0 + 2 * KHX), KHX per k-group - the whole K (KHX = KH, ks0 = 0)
  // except in stage 3 of a head-split cross-attention launch, where a workgroup multiplies its own heads' half of K)
  using KHfull = std::integral_constant<int, KH>;
  auto load_unit = [&](const bf16_t* wf_hi, const bf16_t* wf_lo, int frag0, int u, auto nsx_tag, auto kh_tag, int ks0) {
    constexpr int NSX = decltype(nsx_tag)::value;   // fragments per wave in this stage (NS, or 1 in a column-split stage)
    constexpr int KHX = decltype(kh_tag)::value;
    const int ks = ks0 + kg * KHX + u / NSX, nf = frag0 + (u % NSX) * 4 + wn;
    const size_t e = ((size_t)(nf * KSTEPS + ks) * 64 + lane) * 8;
    BFrag f;
    f.h = *reinterpret_cast<const bf16x8*>(wf_hi + e);
    f.l = *reinterpret_cast<const bf16x8*>(wf_lo + e);
    return f;
  };
  auto stage_prologue = [&](const bf16_t* wf_hi, const bf16_t* wf_lo, int frag0, BFrag (&bq)[DEPTH], auto nsx_tag, auto kh_tag, int ks0) __attribute__((always_inline)) {
    constexpr int NSX = decltype(nsx_tag)::value, KHX = decltype(kh_tag)::value;
#pragma unroll
    for (int j = 0; j < DEPTH; ++j)
      if (j < NSX * KHX) bq[j] = load_unit(wf_hi, wf_lo, frag0, j, nsx_tag, kh_tag, ks0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // swap_tag: the activations are the FIRST MFMA operand - the accumulator is then the tile itself, lane = output column,
  // registers = rows 8g + 4lh + e (used for V: that register image IS the V^T operand fragment of the attention kernel)
  auto stage_loop = [&](const bf16_t* wf_hi, const bf16_t* wf_lo, int frag0, BFrag (&bq)[DEPTH], f32x16 (&acc)[NS], auto swap_tag, auto nsx_tag, auto kh_tag, int ks0) __attribute__((always_inline)) {
    constexpr bool SWAP = decltype(swap_tag)::value;
    constexpr int NSX = decltype(nsx_tag)::value, KHX = decltype(kh_tag)::value, UX = NSX * KHX;
    // A fragment of k-step ksl (this wave's k-half): read one k-step ahead of its MFMAs (LDS latency off the chain)
    auto read_a = [&](int ksl, bf16x8& h, bf16x8& l) {
      const int c16 = (ks0 + kg * KHX + ksl) * 2 + lh;             // 16-byte chunk of the row: k-step * 2 + half
      const int off = (c16 >> 3) * CHUNK_PL + l31 * 128 + (((c16 & 7) ^ swz(l31)) << 4);
      h = *reinterpret_cast<const bf16x8*>(a_reg + off);
      l = *reinterpret_cast<const bf16x8*>(a_reg + A_PL + off);
    };
    bf16x8 ah[2], al[2];
    read_a(0, ah[0], al[0]);
#pragma unroll
    for (int u = 0; u < UX; ++u) {
      const int ksl = u / NSX, cur = ksl & 1;
      if (u % NSX == 0 && ksl + 1 < KHX) read_a(ksl + 1, ah[cur ^ 1], al[cur ^ 1]);
      const BFrag f = bq[u % DEPTH];
      if (SWAP) {
        acc[u % NSX] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cur], f.h, acc[u % NSX], 0, 0, 0);
        acc[u % NSX] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur], f.l, acc[u % NSX], 0, 0, 0);
        acc[u % NSX] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur], f.h, acc[u % NSX], 0, 0, 0);
      } else {
        acc[u % NSX] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, al[cur], acc[u % NSX], 0, 0, 0);
        acc[u % NSX] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.l, ah[cur], acc[u % NSX], 0, 0, 0);
        acc[u % NSX] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, ah[cur], acc[u % NSX], 0, 0, 0);
      }
      // pinned: without the scheduling barriers hipcc sinks every prefetch down to its use (load -> vmcnt(0) -> MFMA)
      __builtin_amdgcn_sched_barrier(0);
      if (u + DEPTH < UX) bq[u % DEPTH] = load_unit(wf_hi, wf_lo, frag0, u + DEPTH, nsx_tag, kh_tag, ks0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // The two k-groups add their accumulators through `buf` (NS * 16 KiB).  own_tag false: k-group 1 hands everything to k-group
  // 0, which runs the epilogue alone.  own_tag true (stages with NS >= 2 fragments per wave - round 4): fragment ns BELONGS to
  // k-group ns & 1 - each group hands over the fragments of the other and keeps its own, so both directions cross the LDS at
  // once and both groups run the epilogue, on half the fragments each (same two addends per sum: the same bits).
  auto owns = [&](int ns, auto own_tag) { return decltype(own_tag)::value ? (ns & 1) == kg : kg == 0; };
  auto kgroup_reduce = [&](f32x16 (&acc)[NS], char* buf, auto own_tag) __attribute__((always_inline)) {
    float* red = reinterpret_cast<float*>(buf);
    __syncthreads();                               // the buffer's previous readers are done; every wave has left its k-loop
    if constexpr (PFW) {                           // (nine-wave instantiations sit at the 168-VGPR cap: dword form, no temporaries)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
        if (!owns(ns, own_tag)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((ns * 4 + wn) * 16 + r) * 64 + lane] = acc[ns][r];
        }
      __syncthreads();
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
        if (owns(ns, own_tag)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ns][r] += red[((ns * 4 + wn) * 16 + r) * 64 + lane];
        }
      return;
    }
    // (16-byte LDS accesses, lane-linear: a quarter of the instructions of the dword form)
    float4* red4 = reinterpret_cast<float4*>(red);
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
      if (!owns(ns, own_tag)) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          red4[((ns * 4 + wn) * 4 + g) * 64 + lane] = make_float4(acc[ns][4 * g], acc[ns][4 * g + 1], acc[ns][4 * g + 2], acc[ns][4 * g + 3]);
      }
    __syncthreads();
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
      if (owns(ns, own_tag)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = red4[((ns * 4 + wn) * 4 + g) * 64 + lane];
          acc[ns][4 * g] += v.x; acc[ns][4 * g + 1] += v.y; acc[ns][4 * g + 2] += v.z; acc[ns][4 * g + 3] += v.w;
        }
      }
  };
  using OwnSplit = std::integral_constant<bool, (NS >= 2) && DV_CHAIN_OWN>;   // stage 1 and the whole-width stage-2 passes
  using OwnNone = std::false_type;

  // ================= A operand of stage 1 =================
  if (AMODE == 0) {
    // split planes from global by LDS-DMA: instruction = (chunk, 8 rows) of one plane
    for (int idx = wave; idx < CH * 4; idx += NWV) {
      const int c = idx >> 2, r8 = idx & 3, row = r8 * 8 + d_row;
      const size_t e = (size_t)(m0 + row) * C + c * 64 + ((d_slot ^ swz(row)) << 3);
      const unsigned dst = a_base + (unsigned)(c * CHUNK_PL + r8 * 1024);
      glds16(p.a_hi + e, dst);
      glds16(p.a_lo + e, dst + A_PL);
    }
  }
  BFrag bq[DEPTH];
  stage_prologue(p.w1_hi, p.w1_lo, 0, bq, std::integral_constant<int, NS>{}, KHfull{}, 0);   // the first weight fragments fly under the A operand's arrival / conversion
  if (!PFW && !XA && DV_CHAIN_PFMODE == 2) l2_prefetch(wave, NWV);   // (XA kernels: measured +-0 with it, round 4)
  if (AMODE == 1) {
    // GroupNorm of the fp32 rows, once per row-block: table of this utterance, then convert
    const int T = p.T, b_item = m0 / T, Tv = p.Tv > 0 ? p.Tv : T;   // row pitch / frames that exist (padded row spaces)
    // (the rows are requested FIRST: their cold-miss latency runs under the table's loads, reductions and barriers)
    constexpr int TASKS = BM * (C / 4), RNDS = (TASKS + NT - 1) / NT;    // (row, 4-channel group); NT % (C/4) may be != 0
    float4 rv[RNDS];
#pragma unroll
    for (int j = 0; j < RNDS; ++j) {
      const int id = min(j * NT + tid, TASKS - 1), row = id / (C / 4), c4 = id - row * (C / 4);
      rv[j] = *reinterpret_cast<const float4*>(p.x + (size_t)(m0 + row) * C + c4 * 4);
    }
    {
      const int G = p.groups, cg = C / G, nvb = cg >> 4, RB = T >> 5, nblk = C >> 4, n_ent = RB * nblk;
      const int cc = min(tid, C - 1);
      const float pg = p.gamma[cc], pb = p.beta[cc];
      float2* s_ent = reinterpret_cast<float2*>(red_reg);                          // NS * 16 KiB: n_ent <= NS * 2048
      for (int e0 = 0; e0 < n_ent; e0 += 2 * NT) {
        float2 ev[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int e = min(e0 + k * NT + tid, n_ent - 1);
          ev[k] = reinterpret_cast<const float2*>(p.stat16)[(size_t)b_item * n_ent + e];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int e = e0 + k * NT + tid;
          if (e < n_ent) s_ent[e] = ev[k];
        }
      }
      __syncthreads();
      const int lpg = 64 / G, g = lane / lpg, sub = lane - g * lpg;
      double s1 = 0, q = 0;
      for (int i0 = sub; i0 < RB * nvb; i0 += 4 * lpg) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k * lpg, rb = i / nvb;
          const float2 v = s_ent[min(rb * nblk + g * nvb + (i - rb * nvb), n_ent - 1)];
          if (i < RB * nvb) { s1 += (double)v.x; q += (double)v.y + (double)v.x * (double)v.x / (double)(16 * min(32, Tv - 32 * rb)); }
        }
      }
      for (int o = lpg >> 1; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); q += __shfl_xor(q, o); }
      const double n = (double)cg * (double)Tv, mean_d = s1 / n;
      double var = q / n - mean_d * mean_d;
      var = var > 0 ? var : 0;
      const float mean = (float)mean_d, rstd = 1.0f / sqrtf((float)var + p.gn_eps);
      const int src_lane = cc / cg * lpg;
      const float gm = __shfl(mean, src_lane), gr = __shfl(rstd, src_lane);
      if (tid < C) { const float a = gr * pg; s_gscale[tid] = a; s_gshift[tid] = pb - gm * a; }
      __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < RNDS; ++j) {
      const int id = j * NT + tid;
      if (id >= TASKS) continue;
      const int row = id / (C / 4), c4 = id - row * (C / 4);
      const float4 sc = *reinterpret_cast<const float4*>(s_gscale + c4 * 4);
      const float4 sh = *reinterpret_cast<const float4*>(s_gshift + c4 * 4);
      const float v0 = fmaf(rv[j].x, sc.x, sh.x), v1 = fmaf(rv[j].y, sc.y, sh.y), v2 = fmaf(rv[j].z, sc.z, sh.z),
                  v3 = fmaf(rv[j].w, sc.w, sh.w);
      uint2 hw, lw;
      hw.x = pk(v0, v1); hw.y = pk(v2, v3);
      lw.x = pk(v0 - __uint_as_float(hw.x << 16), v1 - __uint_as_float(hw.x & 0xffff0000u));
      lw.y = pk(v2 - __uint_as_float(hw.y << 16), v3 - __uint_as_float(hw.y & 0xffff0000u));
      const int c = c4 >> 4, s16 = (c4 & 15) >> 1;
      const int off = c * CHUNK_PL + row * 128 + ((s16 ^ swz(row)) << 4) + (c4 & 1) * 8;
      *reinterpret_cast<uint2*>(a_reg + off) = hw;
      *reinterpret_cast<uint2*>(a_reg + A_PL + off) = lw;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  DV_CTRACE(1);
  if (AMODE == 0) wait_vmcnt<0>();                  // this wave's A pieces have landed
  __syncthreads();                                 // A operand complete
  DV_CTRACE(2);

  // ================= stage 1 =================
  f32x16 acc[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ns][r] = 0.f;
  stage_loop(p.w1_hi, p.w1_lo, 0, bq, acc, std::false_type{}, std::integral_constant<int, NS>{}, KHfull{}, 0);
  DV_CTRACE(3);
  // the second GEMM's first fragments fly during the hand-over and the epilogue
  using Ns2 = std::integral_constant<int, CS ? 1 : NS>;
  const int cs_frag0 = CS ? part * 4 : 0;          // column split: this part's fragment group
  stage_prologue(p.w2_hi, p.w2_lo, ps_lo * (C / 32) + cs_frag0, bq, Ns2{}, KHfull{}, 0);
  kgroup_reduce(acc, red_reg, OwnSplit{});         // (its leading barrier: every wave is done reading the A operand)
  DV_CTRACE(4);
  // epilogue 1 (the fragment's owner): x1 = acc + b1 (+ res) -> out1 fp32, raw split planes into the A region, row partials
  {
    const int m = m0 + l31;
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      if (!owns(ns, OwnSplit{})) continue;
      const int nf = ns * 128 + wn * 32 + 4 * lh;           // column of g = 0, e = 0
      float vv[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4*>(p.b1 + nf + 8 * g);
        float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.res) rr = *reinterpret_cast<const float4*>(p.res + (size_t)m * C + nf + 8 * g);
        vv[4 * g] = acc[ns][4 * g] + bv.x + rr.x; vv[4 * g + 1] = acc[ns][4 * g + 1] + bv.y + rr.y;
        vv[4 * g + 2] = acc[ns][4 * g + 2] + bv.z + rr.z; vv[4 * g + 3] = acc[ns][4 * g + 3] + bv.w + rr.w;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nf + 8 * g;
        // (XA with two workgroups per row block: BOTH write x1 - identical bits - because whichever finishes reads it back as
        // the stage-3 residual)
        if (part == 0 || (XA && CS)) dv_st16(p.out1 + (size_t)m * C + n, make_float4(vv[4 * g], vv[4 * g + 1], vv[4 * g + 2], vv[4 * g + 3]));
        uint2 hw, lw;
        hw.x = pk(vv[4 * g], vv[4 * g + 1]); hw.y = pk(vv[4 * g + 2], vv[4 * g + 3]);
        lw.x = pk(vv[4 * g] - __uint_as_float(hw.x << 16), vv[4 * g + 1] - __uint_as_float(hw.x & 0xffff0000u));
        lw.y = pk(vv[4 * g + 2] - __uint_as_float(hw.y << 16), vv[4 * g + 3] - __uint_as_float(hw.y & 0xffff0000u));
        const int c = n >> 6, s16 = (n & 63) >> 3;
        const int off = c * CHUNK_PL + l31 * 128 + ((s16 ^ swz(l31)) << 4) + ((n & 7) >> 2) * 8;
        *reinterpret_cast<uint2*>(a_reg + off) = hw;
        *reinterpret_cast<uint2*>(a_reg + A_PL + off) = lw;
      }
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) a += vv[r];
      a = pair_sum32(a);
      const float mb = a * (1.0f / 32.0f);
#pragma unroll
      for (int r = 0; r < 16; ++r) q += (vv[r] - mb) * (vv[r] - mb);
      q = pair_sum32(q);
      if (lh == 0) s_rowp[l31][ns * 4 + wn] = make_float2(a, q);
    }
  }
  __syncthreads();
  if (tid < BM) {
    float s1 = 0.f;
    for (int k = 0; k < 4 * NS; ++k) s1 += s_rowp[tid][k].x;
    const float inv_c = 1.0f / (float)C, mean = s1 * inv_c;
    float m2 = 0.f;
    for (int k = 0; k < 4 * NS; ++k) {
      const float2 v = s_rowp[tid][k];
      const float dm = v.x * (1.0f / 32.0f) - mean;
      m2 += v.y + 32.0f * dm * dm;
    }
    s_ln[tid] = make_float2(mean, 1.0f / sqrtf(m2 * inv_c + p.ln_eps));
  }
  __syncthreads();                                 // planes of x1 and the row statistics are visible
  DV_CTRACE(5);

  // ================= stage 2: `passes` contractions of C output columns each =================
  // SA (self-attention operands as MFMA fragments: sa_kf / sa_vf, passes = q | k | v): the three passes are separate code
  // - pass 1 stores K as K fragments, pass 2 runs with swapped operands and stores its accumulator as V^T fragments
  auto do_pass = [&](const int ps, const int npass, auto mode_tag) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_tag)::value;    // 0: fp32 (or the in-kernel query), 1: K fragments, 2: V^T fragments
    constexpr bool sa = MODE != 0;
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ns][r] = 0.f;
    stage_loop(p.w2_hi, p.w2_lo, ps * (C / 32) + cs_frag0, bq, acc, std::integral_constant<bool, MODE == 2>{}, Ns2{}, KHfull{}, 0);
    if (ps == 0) DV_CTRACE(6);
    if (ps + 1 < npass) stage_prologue(p.w2_hi, p.w2_lo, (ps + 1) * (C / 32) + cs_frag0, bq, Ns2{}, KHfull{}, 0);
    using OwnP = std::integral_constant<bool, OwnSplit::value && !CS>;   // (a column-split pass has one fragment per wave)
    kgroup_reduce(acc, red_reg, OwnP{});
    if (ps == 0) DV_CTRACE(7);
    if (MODE == 2) {
      {
        // V: lane = channel ns*128 + wn*32 + l31, registers = rows (keys) 8g + 4lh + e.  Registers 8kb .. 8kb+7 are the
        // 8 keys a lane of the S^T accumulator holds for k-block kb: written as they are, they form the V^T fragment
        // (channel block f = ns*4 + wn, k-block kb) of this 32-key tile
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          if (!owns(ns, OwnP{})) continue;
          const int n = 2 * C + ns * 128 + wn * 32 + l31;
          const float bv = p.b2[n], uv = p.u2[n];
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            uint4 hw, lw;
            unsigned* hp = &hw.x; unsigned* lp = &lw.x;
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
              // rows 8g + 4lh + e, g = 2kb + gg: their (mean, rstd) are 32 contiguous bytes of s_ln
              const int g = 2 * kb + gg;
              const float4 s01 = *reinterpret_cast<const float4*>(&s_ln[8 * g + 4 * lh]);
              const float4 s23 = *reinterpret_cast<const float4*>(&s_ln[8 * g + 4 * lh + 2]);
              const float x0 = s01.y * (acc[ns][4 * g] - s01.x * uv) + bv, x1 = s01.w * (acc[ns][4 * g + 1] - s01.z * uv) + bv;
              const float x2 = s23.y * (acc[ns][4 * g + 2] - s23.x * uv) + bv, x3 = s23.w * (acc[ns][4 * g + 3] - s23.z * uv) + bv;
              if (DV_ATTN_PF16) {                    // V as split fp16 (dv_device.h)
                dv_split_pk_f16(x0, x1, hp[2 * gg], lp[2 * gg]);
                dv_split_pk_f16(x2, x3, hp[2 * gg + 1], lp[2 * gg + 1]);
              } else {
                const unsigned h01 = pk(x0, x1), h23 = pk(x2, x3);
                hp[2 * gg] = h01; hp[2 * gg + 1] = h23;
                lp[2 * gg] = pk(x0 - __uint_as_float(h01 << 16), x1 - __uint_as_float(h01 & 0xffff0000u));
                lp[2 * gg + 1] = pk(x2 - __uint_as_float(h23 << 16), x3 - __uint_as_float(h23 & 0xffff0000u));
              }
            }
            const size_t eo = ((((size_t)rb * (C / 32) + ns * 4 + wn) * 2 + kb) * 64 + lane) * 8;
            dv_st16(p.sa_vf_hi + eo, hw);
            dv_st16(p.sa_vf_lo + eo, lw);
          }
        }
      }
    } else {
      const int m = m0 + l31;
      const float2 st = s_ln[l31];
#pragma unroll
      for (int ns = 0; ns < (CS ? 1 : NS); ++ns) {
        if (!owns(ns, OwnP{})) continue;
        const int nf = ps * C + (CS ? part : ns) * 128 + wn * 32 + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nf + 8 * g;
          const float4 bv = *reinterpret_cast<const float4*>(p.b2 + n);
          const float4 uv = *reinterpret_cast<const float4*>(p.u2 + n);
          float4 o;
          o.x = st.y * (acc[ns][4 * g] - st.x * uv.x) + bv.x;
          o.y = st.y * (acc[ns][4 * g + 1] - st.x * uv.y) + bv.y;
          o.z = st.y * (acc[ns][4 * g + 2] - st.x * uv.z) + bv.z;
          o.w = st.y * (acc[ns][4 * g + 3] - st.x * uv.w) + bv.w;
          if (MODE == 1) {
            // K: 4 channels of key row l31 -> 8 bytes of the K fragment (tile = this row block, 16-channel group n / 16):
            // its lane (g & 1, l31) holds channels (n & ~7) .. + 8, this lane's four at byte 8 lh
            uint2 hw, lw;
            hw.x = pk(o.x, o.y); hw.y = pk(o.z, o.w);
            lw.x = pk(o.x - __uint_as_float(hw.x << 16), o.y - __uint_as_float(hw.x & 0xffff0000u));
            lw.y = pk(o.z - __uint_as_float(hw.y << 16), o.w - __uint_as_float(hw.y & 0xffff0000u));
            const int nk = n - C;
            const size_t eo = (((size_t)rb * (C / 16) + (nk >> 4)) * 64 + (g & 1) * 32 + l31) * 8 + lh * 4;
            dv_st8(p.sa_kf_hi + eo, hw);
            dv_st8(p.sa_kf_lo + eo, lw);
          } else if (!XA) dv_st16(p.out2 + (size_t)m * p.ldo2 + n, o);
          else {
            // the query of the cross attention stays in the workgroup: pre-scaled (d^-1/2 log2 e) split planes in the A
            // region (the planes of x1 are dead: every wave has left the stage-2 k-loop)
            const float q0 = o.x * p.xa_qscale, q1 = o.y * p.xa_qscale, q2 = o.z * p.xa_qscale, q3 = o.w * p.xa_qscale;
            uint2 hw, lw;
            hw.x = pk(q0, q1); hw.y = pk(q2, q3);
            lw.x = pk(q0 - __uint_as_float(hw.x << 16), q1 - __uint_as_float(hw.x & 0xffff0000u));
            lw.y = pk(q2 - __uint_as_float(hw.y << 16), q3 - __uint_as_float(hw.y & 0xffff0000u));
            const int c = n >> 6, s16 = (n & 63) >> 3;
            const int off = c * CHUNK_PL + l31 * 128 + ((s16 ^ swz(l31)) << 4) + ((n & 7) >> 2) * 8;
            *reinterpret_cast<uint2*>(a_reg + off) = hw;
            *reinterpret_cast<uint2*>(a_reg + A_PL + off) = lw;
          }
        }
      }
    }
    if (ps == 0) DV_CTRACE(8);
  };
  if constexpr (SA) {
    if (ps_lo <= 0 && 0 < ps_hi) do_pass(0, ps_hi, std::integral_constant<int, 0>{});
    if (ps_lo <= 1 && 1 < ps_hi) do_pass(1, ps_hi, std::integral_constant<int, 1>{});
    if (ps_lo <= 2 && 2 < ps_hi) do_pass(2, ps_hi, std::integral_constant<int, 2>{});
  } else {
    for (int ps = ps_lo; ps < ps_hi; ++ps) do_pass(ps, ps_hi, std::integral_constant<int, 0>{});
  }
  if constexpr (XA) {
    // ================= cross attention: wave h = head h, 32 queries, keys / values as MFMA fragments from global =================
    // Same arithmetic as k_attention (attn_tile.h): S^T = K Q^T and O^T += V^T P^T with split-bf16 operands (3 products),
    // scores in the log2 domain, online softmax lane-local (lane = query, registers = keys), P never leaves registers.
    // XS (XA with CS: two workgroups per row block, `part` = 0 / 1): this workgroup runs heads 4 part .. 4 part + 3 - the columns
    // its stage 2 produced - with TWO waves per head (wave w: head w & 3, key tiles of parity w >> 2; merged through LDS
    // behind the loop), then multiplies ITS heads' half of K in stage 3 and hands the partial sums over to its partner (or
    // takes the partner's and finishes): the launch fills 256 CUs at M = 4096 and every workgroup streams 2/3 of the weights
    constexpr bool XS = CS;
    using KH3 = std::integral_constant<int, XS ? KH / 2 : KH>;
    const int ks3 = XS ? part * KH : 0;              // first k-step of this workgroup's stage-3 range
    constexpr int d = 16 * NS, KSq = NS, NBv = NS == 3 ? 2 : 1;     // 8 heads: d = C / 8
    const int nT = p.xa_nT;
    const int h = XS ? part * 4 + (wave & 3) : wave, b_item = m0 / p.T;
    const int kh = XS ? wave >> 2 : 0, tstep = XS ? 2 : 1;           // key-tile parity of this wave / tile stride
    const size_t bh = (size_t)b_item * NWV + h;
    const bf16x8* kfh = reinterpret_cast<const bf16x8*>(p.xa_kf_hi) + bh * nT * KSq * 64 + lane;
    const bf16x8* kfl = reinterpret_cast<const bf16x8*>(p.xa_kf_lo) + bh * nT * KSq * 64 + lane;
    const bf16x8* vfh = reinterpret_cast<const bf16x8*>(p.xa_vf_hi) + bh * nT * 2 * NBv * 64 + lane;
    const bf16x8* vfl = reinterpret_cast<const bf16x8*>(p.xa_vf_lo) + bh * nT * 2 * NBv * 64 + lane;
    const float* bias = p.xa_bias + (size_t)b_item * nT * 32 + 4 * lh;
    // K / V fragments and the key bias of tile t + 1 are requested before tile t is multiplied (pinned like the weight
    // prefetch: the loads have no consumer in the current iteration and would otherwise sink to their use)
    struct KVT { bf16x8 kh[KSq], kl[KSq], vh[2][NBv], vl[2][NBv]; float4 bv[4]; };
    auto load_kv = [&](int t) {
      KVT f;
      const int tc = min(t, nT - 1);
#pragma unroll
      for (int ks = 0; ks < KSq; ++ks) { f.kh[ks] = kfh[(size_t)(tc * KSq + ks) * 64]; f.kl[ks] = kfl[(size_t)(tc * KSq + ks) * 64]; }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int nb = 0; nb < NBv; ++nb) {
          f.vh[kb][nb] = vfh[(size_t)((tc * 2 + kb) * NBv + nb) * 64];
          f.vl[kb][nb] = vfl[(size_t)((tc * 2 + kb) * NBv + nb) * 64];
        }
#pragma unroll
      for (int g = 0; g < 4; ++g) f.bv[g] = *reinterpret_cast<const float4*>(bias + tc * 32 + 8 * g);
      return f;
    };
    // The FIRST key tile is requested here - the prompt's fragments depend on nothing this launch computes - so that its
    // round trip (the fragments were written a whole sampler run ago: L2 misses) runs under the two barriers and the query
    // fragment reads below instead of at the head of the key loop [round 4: requested behind the second barrier]
    KVT cur = load_kv(kh);
    __builtin_amdgcn_sched_barrier(0);
    stage_prologue(p.w3_hi, p.w3_lo, 0, bq, std::integral_constant<int, NS>{}, KH3{}, ks3);   // the output projection's first weight fragments fly under the attention
    __syncthreads();                               // query planes complete
    bf16x8 qh[KSq], ql[KSq];
#pragma unroll
    for (int ks = 0; ks < KSq; ++ks) {
      const int c16 = ((h * d + ks * 16) >> 3) + lh;
      const int off = (c16 >> 3) * CHUNK_PL + l31 * 128 + (((c16 & 7) ^ swz(l31)) << 4);
      qh[ks] = *reinterpret_cast<const bf16x8*>(a_reg + off);
      ql[ks] = *reinterpret_cast<const bf16x8*>(a_reg + A_PL + off);
    }
    f32x16 o[NBv];
#pragma unroll
    for (int nb = 0; nb < NBv; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    __syncthreads();                               // every wave holds its query fragments: the A region may be rewritten
    DV_CTRACE(10);
    for (int t = kh; t < nT; t += tstep) {
      __builtin_amdgcn_sched_barrier(0);
      KVT nxt = load_kv(t + tstep);
      __builtin_amdgcn_sched_barrier(0);
      f32x16 sc;
      // (the key bias IS the initial accumulator: no add behind the MFMAs, 12 VGPRs fewer)
#pragma unroll
      for (int g = 0; g < 4; ++g) { sc[4 * g] = cur.bv[g].x; sc[4 * g + 1] = cur.bv[g].y; sc[4 * g + 2] = cur.bv[g].z; sc[4 * g + 3] = cur.bv[g].w; }
#pragma unroll
      for (int ks = 0; ks < KSq; ++ks) {
        sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.kl[ks], qh[ks], sc, 0, 0, 0);
        sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.kh[ks], ql[ks], sc, 0, 0, 0);
        sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.kh[ks], qh[ks], sc, 0, 0, 0);
      }
      float tmax = m_run;
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sc[r]);
      const float m_new = pair_max32(tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[r] = __builtin_amdgcn_exp2f(sc[r] - m_new); psum += sc[r]; }
      l_run = l_run * alpha + psum;
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int nb = 0; nb < NBv; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[nb][r] *= alpha;
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 hw, lw;
        if (DV_ATTN_PF16) {                            // P as one fp16 plane, V as split fp16: two products (dv_device.h)
#pragma unroll
          for (int e = 0; e < 4; ++e) hw[e] = dv_cvt_pk_f16(sc[kb * 8 + 2 * e], sc[kb * 8 + 2 * e + 1]);
          const dv_f16x8 ph16 = __builtin_bit_cast(dv_f16x8, hw);
#pragma unroll
          for (int nb = 0; nb < NBv; ++nb) {
            o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dv_f16x8, cur.vl[kb][nb]), ph16, o[nb], 0, 0, 0);
            o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dv_f16x8, cur.vh[kb][nb]), ph16, o[nb], 0, 0, 0);
          }
          continue;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x0 = sc[kb * 8 + 2 * e], x1 = sc[kb * 8 + 2 * e + 1];
          const unsigned h2 = pk(x0, x1);
          hw[e] = h2;
          lw[e] = DV_ATTN_PLO ? pk(x0 - __uint_as_float(h2 << 16), x1 - __uint_as_float(h2 & 0xffff0000u)) : 0u;
        }
        const bf16x8 ph = __builtin_bit_cast(bf16x8, hw), pl = __builtin_bit_cast(bf16x8, lw);
#pragma unroll
        for (int nb = 0; nb < NBv; ++nb) {
          o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.vl[kb][nb], ph, o[nb], 0, 0, 0);
          if (DV_ATTN_PLO) o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.vh[kb][nb], pl, o[nb], 0, 0, 0);
          o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.vh[kb][nb], ph, o[nb], 0, 0, 0);
        }
      }
      cur = nxt;
    }
    if (XS) {
      // merge the two key halves of every head: wave w + 4 hands (m, l, O) over, wave w combines (log2 domain)
      float* mg = reinterpret_cast<float*>(red_reg);               // [4 heads][2 + 16 NBv][64 lanes]
      constexpr int MW = 2 + 16 * NBv;
      if (kh == 1) {
        float* q = mg + (size_t)(wave & 3) * MW * 64 + lane;
        q[0] = m_run; q[64] = l_run;
#pragma unroll
        for (int nb = 0; nb < NBv; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) q[(2 + nb * 16 + r) * 64] = o[nb][r];
      }
      __syncthreads();
      if (kh == 0) {
        const float* q = mg + (size_t)(wave & 3) * MW * 64 + lane;
        const float m1 = q[0], l1 = q[64];
        const float mm = fmaxf(m_run, m1);
        const float a0 = __builtin_amdgcn_exp2f(m_run - mm), a1 = __builtin_amdgcn_exp2f(m1 - mm);
        l_run = l_run * a0 + l1 * a1;
#pragma unroll
        for (int nb = 0; nb < NBv; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[nb][r] = o[nb][r] * a0 + q[(2 + nb * 16 + r) * 64] * a1;
      }
    }
    if (!XS || kh == 0) {
      const float inv = 1.0f / pair_sum32(l_run);
      // O (lane = query row, registers = channels 8g + 4lh + e of block nb) -> split planes in the A region, columns of head h
#pragma unroll
      for (int nb = 0; nb < NBv; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dv = nb * 32 + 8 * g + 4 * lh;
          if (dv < d) {
            const float v0 = o[nb][4 * g] * inv, v1 = o[nb][4 * g + 1] * inv, v2 = o[nb][4 * g + 2] * inv, v3 = o[nb][4 * g + 3] * inv;
            uint2 hw, lw;
            hw.x = pk(v0, v1); hw.y = pk(v2, v3);
            lw.x = pk(v0 - __uint_as_float(hw.x << 16), v1 - __uint_as_float(hw.x & 0xffff0000u));
            lw.y = pk(v2 - __uint_as_float(hw.y << 16), v3 - __uint_as_float(hw.y & 0xffff0000u));
            const int n = h * d + dv, c = n >> 6, s16 = (n & 63) >> 3;
            const int off = c * CHUNK_PL + l31 * 128 + ((s16 ^ swz(l31)) << 4) + ((n & 7) >> 2) * 8;
            *reinterpret_cast<uint2*>(a_reg + off) = hw;
            *reinterpret_cast<uint2*>(a_reg + A_PL + off) = lw;
          }
        }
    }
    DV_CTRACE(11);
    __syncthreads();                               // attention output complete = A operand of stage 3
    DV_CTRACE(12);
    // ================= stage 3: x3 = O W3^T + b3 + x1 -> fp32, raw planes and LayerNorm row partials in global =================
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ns][r] = 0.f;
    stage_loop(p.w3_hi, p.w3_lo, 0, bq, acc, std::false_type{}, std::integral_constant<int, NS>{}, KH3{}, ks3);
    DV_CTRACE(13);
    kgroup_reduce(acc, red_reg, OwnNone{});
    DV_CTRACE(14);
    if (XS) {
      // hand-over between the two workgroups of the row block (the fused split-K pair's protocol, gemm_tile.h): dump the
      // partial sums written through, take a ticket; the first to arrive leaves, the second adds the partner's dump and
      // runs the epilogue.  Nobody waits for anybody.  [NS fragments][4 column owners][4 float4][64 lanes] per part.
      float4* dump = reinterpret_cast<float4*>(p.xs_buf) + ((size_t)rb * 2 + part) * (NS * 4 * 4 * 64);
      if (kg == 0) {
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            st_handover16(dump + ((ns * 4 + wn) * 4 + g) * 64 + lane, make_float4(acc[ns][4 * g], acc[ns][4 * g + 1], acc[ns][4 * g + 2], acc[ns][4 * g + 3]));
        wait_vmcnt<0>();
      }
      __shared__ unsigned s_arrival;
      __syncthreads();
      if (tid == 0) s_arrival = __hip_atomic_fetch_add(p.xs_ticket + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (s_arrival == 0) return;                    // the partner finishes this row block
      if (tid == 0) __hip_atomic_store(p.xs_ticket + rb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      if (kg == 0) {
        const float4* other = reinterpret_cast<const float4*>(p.xs_buf) + ((size_t)rb * 2 + (part ^ 1)) * (NS * 4 * 4 * 64);
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          float4 v[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) v[g] = ld_handover16(other + ((ns * 4 + wn) * 4 + g) * 64 + lane);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            acc[ns][4 * g] += v[g].x; acc[ns][4 * g + 1] += v[g].y; acc[ns][4 * g + 2] += v[g].z; acc[ns][4 * g + 3] += v[g].w;
          }
        }
      }
    }
    if (kg == 0) {
      const int m = m0 + l31;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        const int nf = ns * 128 + wn * 32 + 4 * lh;
        float vv[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = *reinterpret_cast<const float4*>(p.b3 + nf + 8 * g);
          const float4 rr = *reinterpret_cast<const float4*>(p.out1 + (size_t)m * C + nf + 8 * g);   // x1, written by this workgroup
          vv[4 * g] = acc[ns][4 * g] + bv.x + rr.x; vv[4 * g + 1] = acc[ns][4 * g + 1] + bv.y + rr.y;
          vv[4 * g + 2] = acc[ns][4 * g + 2] + bv.z + rr.z; vv[4 * g + 3] = acc[ns][4 * g + 3] + bv.w + rr.w;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
          dv_st16(p.out3 + (size_t)m * C + nf + 8 * g, make_float4(vv[4 * g], vv[4 * g + 1], vv[4 * g + 2], vv[4 * g + 3]));
        // split planes as 16-byte stores (lane pairs exchange halves: dv_device.h store_planes16)
        store_planes16(p.out3_hi, p.out3_lo, (size_t)m * C + nf - 4 * lh, lh, vv);
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += vv[r];
        a = pair_sum32(a);
        const float mb = a * (1.0f / 32.0f);
#pragma unroll
        for (int r = 0; r < 16; ++r) q += (vv[r] - mb) * (vv[r] - mb);
        q = pair_sum32(q);
        if (lh == 0) reinterpret_cast<float2*>(p.rowstat3)[(size_t)m * (C / 32) + ns * 4 + wn] = make_float2(a, q);
      }
    }
  }
  DV_CTRACE(9);
}

// ---------------------------------------------------------------------------------------
// k_chain_ff (ChainFFParams): LN3 -> GEGLU -> merged ff.net.2 + proj_out + residual of a C = 128 transformer block as one
// row-block launch.  As separate GEMMs the two take 29 + 18 us at the bench shape (M = 8192: a [M, 8C] GEMM whose
// 4C-wide product makes an HBM round trip, then a K = 5C GEMM); a workgroup that owns 32 rows streams 0.85 MB of weights
// and keeps the product in LDS.
//   stage A: 8 waves = 8 packed 64-column blocks per pass ([32 a | 32 gate]: both fragments in one wave, full K = C, no
//            k-group hand-over), two passes -> a * gelu(gate) as split planes [32 rows x 4C] in LDS
//   stage B: 4 column-fragment owners x 2 k-groups over K = 5C = [h3 planes | product planes], then bias + residual,
//            fp32 rows + 32x16 block statistics for the next GroupNorm
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT_LAUNCH) void k_chain_ff(const ChainFFParams p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int C = 128, A_CH = 2, G_CH = 8;                 // 64-channel chunks of h3 / of the GEGLU product
  constexpr int A_PL = A_CH * CHUNK_PL, G_PL = G_CH * CHUNK_PL;
  char* const a_reg = smem;                                  // [2 planes][2 chunks][32 rows][128 B]
  char* const g_reg = smem + 2 * A_PL;                       // [2 planes][8 chunks][32 rows][128 B]
  char* const red_reg = g_reg + 2 * G_PL;                    // 16 KiB k-group hand-over
  __shared__ __attribute__((aligned(16))) float2 s_ln[BM];
  asm volatile("" ::"s"(p.M), "s"(p.rowstat), "s"(p.wg_lo), "s"(p.wm_hi), "s"(p.res), "s"(p.out_lo));
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave == NWV) {                                         // ninth wave: L2 prefetch of the weight planes (see k_chain2)
    __shared__ __attribute__((aligned(256))) unsigned s_pf[64];
    const int xw = blockIdx.x >> 3, nxw = (gridDim.x + 7) >> 3;
    const int lg = 8 * C * C / 64, lm = 5 * C * C / 64;      // 128-byte lines per plane
    const int total = 2 * (lg + lm), per = (total + nxw - 1) / nxw, end = min(total, (xw + 1) * per);
    for (int ln = xw * per + lane; ln < end; ln += 64) {
      const char* src;
      if (ln < lg) src = reinterpret_cast<const char*>(p.wg_hi) + (size_t)ln * 128;
      else if (ln < 2 * lg) src = reinterpret_cast<const char*>(p.wg_lo) + (size_t)(ln - lg) * 128;
      else if (ln < 2 * lg + lm) src = reinterpret_cast<const char*>(p.wm_hi) + (size_t)(ln - 2 * lg) * 128;
      else src = reinterpret_cast<const char*>(p.wm_lo) + (size_t)(ln - 2 * lg - lm) * 128;
      glds4(src, (unsigned)(size_t)s_pf);
    }
    const char* r0 = reinterpret_cast<const char*>(p.res + (size_t)blockIdx.x * BM * C);
    for (int ln = lane; ln < BM * C / 32; ln += 64) glds4(r0 + (size_t)ln * 128, (unsigned)(size_t)s_pf);
    for (int ln = lane; ln < 8 * C / 32; ln += 64) {
      glds4(reinterpret_cast<const char*>(p.bg) + (size_t)ln * 128, (unsigned)(size_t)s_pf);
      glds4(reinterpret_cast<const char*>(p.ug) + (size_t)ln * 128, (unsigned)(size_t)s_pf);
    }
    return;
  }
  const int l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * BM;
  const unsigned a_base = (unsigned)(size_t)a_reg;
  const int d_row = lane >> 3, d_slot = lane & 7;
  struct BFrag { bf16x8 h, l; };

  // ---- h3 planes -> LDS (LDS-DMA), LayerNorm row statistics from the producer's partials ----
  for (int idx = wave; idx < A_CH * 4; idx += NWV) {
    const int c = idx >> 2, r8 = idx & 3, row = r8 * 8 + d_row;
    const size_t e = (size_t)(m0 + row) * C + c * 64 + ((d_slot ^ swz(row)) << 3);
    const unsigned dst = a_base + (unsigned)(c * CHUNK_PL + r8 * 1024);
    glds16(p.a_hi + e, dst);
    glds16(p.a_lo + e, dst + A_PL);
  }
  // stage A weights: unit U = pass * 16 + ks * 2 + f  (f: 0 = the `a` fragment, 1 = the gate fragment of this wave's block)
  constexpr int UA = 32;
  auto load_a_unit = [&](int U) __attribute__((always_inline)) {
    const int nf = 2 * ((U >> 4) * 8 + wave) + (U & 1), ks = (U & 15) >> 1;
    const size_t e = ((size_t)(nf * 8 + ks) * 64 + lane) * 8;
    BFrag f;
    f.h = *reinterpret_cast<const bf16x8*>(p.wg_hi + e);
    f.l = *reinterpret_cast<const bf16x8*>(p.wg_lo + e);
    return f;
  };
  BFrag bq[DEPTH_FF];
#pragma unroll
  for (int j = 0; j < DEPTH_FF; ++j) bq[j] = load_a_unit(j);
  __builtin_amdgcn_sched_barrier(0);
  if (tid < BM) {
    const float2* src = reinterpret_cast<const float2*>(p.rowstat) + (size_t)(m0 + tid) * (C / 32);
    float2 v[C / 32];
#pragma unroll
    for (int k = 0; k < C / 32; ++k) v[k] = src[k];
    float s1 = 0.f;
#pragma unroll
    for (int k = 0; k < C / 32; ++k) s1 += v[k].x;
    const float inv_c = 1.0f / (float)C, mean = s1 * inv_c;
    float m2 = 0.f;
#pragma unroll
    for (int k = 0; k < C / 32; ++k) { const float dm = v[k].x * (1.0f / 32.0f) - mean; m2 += v[k].y + 32.0f * dm * dm; }
    s_ln[tid] = make_float2(mean, 1.0f / sqrtf(m2 * inv_c + p.ln_eps));
  }
  wait_vmcnt<0>();
  __syncthreads();

  // ================= stage A: GEGLU =================
  {
    auto read_a = [&](int ks, bf16x8& h, bf16x8& l) __attribute__((always_inline)) {
      const int c16 = ks * 2 + lh;
      const int off = (c16 >> 3) * CHUNK_PL + l31 * 128 + (((c16 & 7) ^ swz(l31)) << 4);
      h = *reinterpret_cast<const bf16x8*>(a_reg + off);
      l = *reinterpret_cast<const bf16x8*>(a_reg + A_PL + off);
    };
    const float2 st = s_ln[l31];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      f32x16 acc[2];
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
      bf16x8 ah, al;
#pragma unroll
      for (int uu = 0; uu < 16; ++uu) {
        const int U = ps * 16 + uu;
        if ((uu & 1) == 0) read_a(uu >> 1, ah, al);
        const BFrag f = bq[U % DEPTH_FF];
        acc[uu & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, al, acc[uu & 1], 0, 0, 0);
        acc[uu & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.l, ah, acc[uu & 1], 0, 0, 0);
        acc[uu & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, ah, acc[uu & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (U + DEPTH_FF < UA) bq[U % DEPTH_FF] = load_a_unit(U + DEPTH_FF);
        __builtin_amdgcn_sched_barrier(0);
      }
      // LayerNorm finish + bias, a * gelu(gate) -> product columns blk * 32 + (8g + 4lh + e) as split planes in LDS
      const int blk = ps * 8 + wave;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cl = 8 * g + 4 * lh;
        const float4 ua = *reinterpret_cast<const float4*>(p.ug + blk * 64 + cl), ugt = *reinterpret_cast<const float4*>(p.ug + blk * 64 + 32 + cl);
        const float4 ba = *reinterpret_cast<const float4*>(p.bg + blk * 64 + cl), bgt = *reinterpret_cast<const float4*>(p.bg + blk * 64 + 32 + cl);
        const float ua_[4] = {ua.x, ua.y, ua.z, ua.w}, ug_[4] = {ugt.x, ugt.y, ugt.z, ugt.w};
        const float ba_[4] = {ba.x, ba.y, ba.z, ba.w}, bg_[4] = {bgt.x, bgt.y, bgt.z, bgt.w};
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = st.y * (acc[0][4 * g + e] - st.x * ua_[e]) + ba_[e];
          const float gt = st.y * (acc[1][4 * g + e] - st.x * ug_[e]) + bg_[e];
          v[e] = a * gelu_erf(gt);
        }
        uint2 hw, lw;
        hw.x = pk(v[0], v[1]); hw.y = pk(v[2], v[3]);
        lw.x = pk(v[0] - __uint_as_float(hw.x << 16), v[1] - __uint_as_float(hw.x & 0xffff0000u));
        lw.y = pk(v[2] - __uint_as_float(hw.y << 16), v[3] - __uint_as_float(hw.y & 0xffff0000u));
        const int n = blk * 32 + cl, c = n >> 6, s16 = (n & 63) >> 3;
        const int off = c * CHUNK_PL + l31 * 128 + ((s16 ^ swz(l31)) << 4) + ((n & 7) >> 2) * 8;
        *reinterpret_cast<uint2*>(g_reg + off) = hw;
        *reinterpret_cast<uint2*>(g_reg + G_PL + off) = lw;
      }
    }
  }
  // ================= stage B: [h3 | product] x merged weights =================
  const int wn = wave & 3, kg = wave >> 2;
  constexpr int KSB = 5 * C / 16, KHB = KSB / 2;              // 40 k-steps, 20 per k-group
  auto load_b_unit = [&](int u) __attribute__((always_inline)) {
    const size_t e = ((size_t)(wn * KSB + kg * KHB + u) * 64 + lane) * 8;
    BFrag f;
    f.h = *reinterpret_cast<const bf16x8*>(p.wm_hi + e);
    f.l = *reinterpret_cast<const bf16x8*>(p.wm_lo + e);
    return f;
  };
#pragma unroll
  for (int j = 0; j < DEPTH_FF; ++j) bq[j] = load_b_unit(j);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();                                           // product planes complete
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  {
    auto read_b = [&](int u, bf16x8& h, bf16x8& l) __attribute__((always_inline)) {
      const int ks = kg * KHB + u;
      if (ks < C / 16) {
        const int c16 = ks * 2 + lh;
        const int off = (c16 >> 3) * CHUNK_PL + l31 * 128 + (((c16 & 7) ^ swz(l31)) << 4);
        h = *reinterpret_cast<const bf16x8*>(a_reg + off);
        l = *reinterpret_cast<const bf16x8*>(a_reg + A_PL + off);
      } else {
        const int c16 = (ks - C / 16) * 2 + lh;
        const int off = (c16 >> 3) * CHUNK_PL + l31 * 128 + (((c16 & 7) ^ swz(l31)) << 4);
        h = *reinterpret_cast<const bf16x8*>(g_reg + off);
        l = *reinterpret_cast<const bf16x8*>(g_reg + G_PL + off);
      }
    };
    bf16x8 ah[2], al[2];
    read_b(0, ah[0], al[0]);
#pragma unroll
    for (int u = 0; u < KHB; ++u) {
      const int cur = u & 1;
      if (u + 1 < KHB) read_b(u + 1, ah[cur ^ 1], al[cur ^ 1]);
      const BFrag f = bq[u % DEPTH_FF];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, al[cur], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.l, ah[cur], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, ah[cur], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (u + DEPTH_FF < KHB) bq[u % DEPTH_FF] = load_b_unit(u + DEPTH_FF);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  {
    float* red = reinterpret_cast<float*>(red_reg);
    float4* red4 = reinterpret_cast<float4*>(red);   // (16-byte LDS accesses, lane-linear)
    if (kg == 1) {
#pragma unroll
      for (int g = 0; g < 4; ++g) red4[(wn * 4 + g) * 64 + lane] = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = red4[(wn * 4 + g) * 64 + lane];
      acc[4 * g] += v.x; acc[4 * g + 1] += v.y; acc[4 * g + 2] += v.z; acc[4 * g + 3] += v.w;
    }
  }
  // epilogue: + bias + block residual -> fp32 (+ planes), 32x16 block statistics (sum, M2 about the block mean)
  {
    const int m = m0 + l31, nf = wn * 32 + 4 * lh;
    float vv[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bv = *reinterpret_cast<const float4*>(p.bm + nf + 8 * g);
      const float4 rr = *reinterpret_cast<const float4*>(p.res + (size_t)m * C + nf + 8 * g);
      vv[4 * g] = acc[4 * g] + bv.x + rr.x; vv[4 * g + 1] = acc[4 * g + 1] + bv.y + rr.y;
      vv[4 * g + 2] = acc[4 * g + 2] + bv.z + rr.z; vv[4 * g + 3] = acc[4 * g + 3] + bv.w + rr.w;
    }
    const bool gnx = p.gnx.xchg != nullptr;
    if (p.stats16) {
      // (padded row spaces: the utterance's last row block holds cnt < 32 frames that exist - the others stay out)
      const int Tv = p.Tv > 0 ? p.Tv : p.T, cnt = min(32, Tv - (m0 - (m0 / p.T) * p.T));
      const bool r_ok = l31 < cnt;
      float a1[2] = {0.f, 0.f}, a2[2] = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[r >> 3] += r_ok ? vv[r] : 0.f;
      a1[0] = wave_sum64(a1[0]); a1[1] = wave_sum64(a1[1]);
      const float inv_n = cnt == 32 ? 1.0f / 512.0f : 1.0f / (float)(16 * cnt);
      const float mb[2] = {a1[0] * inv_n, a1[1] * inv_n};
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float dv = r_ok ? vv[r] - mb[r >> 3] : 0.f; a2[r >> 3] = fmaf(dv, dv, a2[r >> 3]); }
      a2[0] = wave_sum64(a2[0]); a2[1] = wave_sum64(a2[1]);
      if (lane < 2) {
        const float2 val = make_float2(lane ? a1[1] : a1[0], lane ? a2[1] : a2[0]);
        reinterpret_cast<float2*>(p.stats16)[(size_t)blockIdx.x * (C / 16) + wn * 2 + lane] = val;
        if (gnx)       // for the other row blocks of the utterance: one 8-byte word (sum, M2), written through
          __hip_atomic_store(p.gnx.xchg + (size_t)blockIdx.x * (C / 16) + wn * 2 + lane,
                             (unsigned long long)__float_as_uint(val.x) | ((unsigned long long)__float_as_uint(val.y) << 32),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    // (the statistics go out FIRST: the other row blocks of the utterance wait for them, nobody waits for these stores)
    if (p.out) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        dv_st16(p.out + (size_t)m * C + nf + 8 * g, make_float4(vv[4 * g], vv[4 * g + 1], vv[4 * g + 2], vv[4 * g + 3]));
    }
    if (p.out_hi) store_planes16(p.out_hi, p.out_lo, (size_t)m * C + nf - 4 * lh, lh, vv);   // 16-byte plane stores (dv_device.h)
    if (gnx) {
      // ---- the consumer's GroupNorm (+ SiLU) of this block's output, a concatenated skip tensor included (gnx_device.h):
      //      the four waves that are left reduce the groups, then normalise their own fragment ----
      __shared__ GnxShared<C> s_gnx;
      GnxTile t;
      t.M = p.M; t.N = C; t.T_out = p.T; t.Tv_out = p.Tv > 0 ? p.Tv : p.T; t.m0 = m0; t.n0 = 0; t.bm = BM; t.bn = C; t.bq = m0 / p.T;
      gnx_finish_table<C>(p.gnx, t, s_gnx, tid, lane, wave, 4, [](int) {});
      float y[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 sa = *reinterpret_cast<const float4*>(s_gnx.gA + nf + 8 * g);
        const float4 sb = *reinterpret_cast<const float4*>(s_gnx.gB + nf + 8 * g);
        y[4 * g] = fmaf(vv[4 * g], sa.x, sb.x); y[4 * g + 1] = fmaf(vv[4 * g + 1], sa.y, sb.y);
        y[4 * g + 2] = fmaf(vv[4 * g + 2], sa.z, sb.z); y[4 * g + 3] = fmaf(vv[4 * g + 3], sa.w, sb.w);
      }
      if (p.gnx.silu) {
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = y[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-y[r]));
      }
      store_planes16(p.gnx.y_hi, p.gnx.y_lo, (size_t)m * C + nf - 4 * lh, lh, y);
    }
  }
}

// cross-attention K / V of one block, fp32 rows [B*L, 2C] (k | v), head h = columns [h*d, h*d + d) -> split-bf16 MFMA
// fragments: K fragment (tile t, k-step ks): lane (l31, lh) = K[key t*32 + l31][ks*16 + lh*8 .. +8];  V^T fragment (tile t,
// k-block kb, channel block nb): lane (l31, lh) = V[key(j)][nb*32 + l31], j = 0..7, key(j) = t*32 + kb*16 + (j < 4 ? 4 lh + j :
// 8 + 4 lh + j - 4) - the order in which a lane of the S^T accumulator holds its keys.  Keys >= L and channels >= d are zero.
__global__ __launch_bounds__(64) void k_kv_frag(const float* __restrict__ kv, bf16_t* __restrict__ kf_hi, bf16_t* __restrict__ kf_lo,
                                                bf16_t* __restrict__ vf_hi, bf16_t* __restrict__ vf_lo, int L, int C, int H, int nT) {
  const int t = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
  const int d = C / H, KSq = d >> 4, NBv = (d + 31) >> 5;
  const size_t bh = (size_t)b * H + h;
  auto split8 = [&](const float (&v)[8], bf16_t* hi, bf16_t* lo) {
    uint4 hw, lw;
    unsigned* hp = &hw.x; unsigned* lp = &lw.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned h2 = pk(v[2 * e], v[2 * e + 1]);
      hp[e] = h2;
      lp[e] = pk(v[2 * e] - __uint_as_float(h2 << 16), v[2 * e + 1] - __uint_as_float(h2 & 0xffff0000u));
    }
    *reinterpret_cast<uint4*>(hi) = hw;
    *reinterpret_cast<uint4*>(lo) = lw;
  };
  for (int ks = 0; ks < KSq; ++ks) {
    const int key = t * 32 + l31;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = key < L ? kv[((size_t)b * L + key) * 2 * C + h * d + ks * 16 + lh * 8 + j] : 0.f;
    const size_t e = ((bh * nT + t) * KSq + ks) * 64 + lane;
    split8(v, kf_hi + e * 8, kf_lo + e * 8);
  }
  for (int kb = 0; kb < 2; ++kb)
    for (int nb = 0; nb < NBv; ++nb) {
      const int ch = nb * 32 + l31;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int key = t * 32 + kb * 16 + (j < 4 ? 4 * lh + j : 8 + 4 * lh + j - 4);
        v[j] = (key < L && ch < d) ? kv[((size_t)b * L + key) * 2 * C + C + h * d + ch] : 0.f;
      }
      const size_t e = (((bh * nT + t) * 2 + kb) * NBv + nb) * 64 + lane;
      if (DV_ATTN_PF16) {                              // V as split fp16 (dv_device.h)
        uint4 hw, lw;
        dv_split_pk_f16(v[0], v[1], hw.x, lw.x); dv_split_pk_f16(v[2], v[3], hw.y, lw.y);
        dv_split_pk_f16(v[4], v[5], hw.z, lw.z); dv_split_pk_f16(v[6], v[7], hw.w, lw.w);
        *reinterpret_cast<uint4*>(vf_hi + e * 8) = hw;
        *reinterpret_cast<uint4*>(vf_lo + e * 8) = lw;
      } else split8(v, vf_hi + e * 8, vf_lo + e * 8);
    }
}
__global__ void k_xbias(const float* __restrict__ mb, float* __restrict__ out, int B, int L, int Lp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Lp) return;
  const int b = i / Lp, k = i - b * Lp;
  out[i] = k < L ? (mb ? mb[(size_t)b * L + k] * 1.44269504088896340736f : 0.f) : -1e30f;
}

// packed weights [rows][Kp] (k contiguous) -> fragment-major: the 16-byte operand pieces of the 64 lanes of one
// (32-row, 16-k) MFMA fragment contiguous (lane (l31, lh) holds W[f*32 + l31][ks*16 + lh*8 .. +8])
__global__ __launch_bounds__(64) void k_relayout_frag(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int Kp, int ksteps) {
  const int nf = blockIdx.x, ks = blockIdx.y, lane = threadIdx.x;
  const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)(nf * 32 + (lane & 31)) * Kp + ks * 16 + (lane >> 5) * 8);
  *reinterpret_cast<uint4*>(dst + ((size_t)(nf * ksteps + ks) * 64 + lane) * 8) = v;
}

template <int NS, int AMODE, bool XA = false, bool SA = false, bool CS = false>
hipError_t init_one() {
  const int smem = 2 * 2 * NS * CHUNK_PL + NS * 16384;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain2<NS, AMODE, XA, SA, CS>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
}
template <int NS, int AMODE, bool XA = false, bool SA = false, bool CS = false>
hipError_t launch_one(const ChainParams& p, hipStream_t st) {
  const int smem = 2 * 2 * NS * CHUNK_PL + NS * 16384;
  static const bool no_pf = [] { const char* e = getenv("DVITS_CHAIN_PF"); return e && e[0] == '0'; }();   // experiment knob: no L2-prefetch wave
  hipLaunchKernelGGL((k_chain2<NS, AMODE, XA, SA, CS>), dim3((p.M / BM) * ((((AMODE == 1 && !XA) || CS) && p.nsplit > 1) ? p.nsplit : 1)),
                     dim3((!chain_pf_wave<NS, AMODE, XA, SA, CS>() || no_pf) ? NT : NT_LAUNCH), smem, st, p);
  return hipGetLastError();
}

}  // namespace

static constexpr int FF_SMEM = 2 * 2 * CHUNK_PL + 2 * 8 * CHUNK_PL + 16384;
bool chain_ff_supported(const ChainFFParams& p, int precision) {
  return precision == 0 && p.C == 128 && p.M % 32 == 0 && p.T % 32 == 0 && p.M % p.T == 0 && p.Tv >= 0 && p.Tv <= p.T && (p.Tv == 0 || p.Tv > p.T - 32);
}
// In-launch GroupNorm of k_chain_ff's output: one workgroup per row block and all of them resident (the conditions of
// gemm_gnx_plan, kernels_gemm.hip, for a launch whose tiles are whole rows)
int chain_ff_gnx_plan(const ChainFFParams& p, int n_cu) {
  const GnxParams& gx = p.gnx;
  if (!p.stats16 || gx.groups <= 0 || gx.groups > 64 || gx.sk_c < 0 || gx.sk_c > DV_GSK || gx.sk_c % 16 != 0 || gx.tscale) return 0;
  if ((p.C + gx.sk_c) % gx.groups != 0) return 0;
  const int cpg = (p.C + gx.sk_c) / gx.groups;
  // (workgroup id = row block: an utterance's workgroups are consecutive ids - gemm_gnx_plan's launch-by-rounds rule)
  if (cpg % 16 != 0 || (p.T / 32) * (cpg / 16) > 256 || (gemm_handover_rounds() ? p.T / 32 : p.M / 32) > n_cu) return 0;
  return (p.M / 32) * (p.C / 16);
}
hipError_t launch_chain_ff(const ChainFFParams& p, int precision, hipStream_t st) {
  if (!chain_ff_supported(p, precision)) return hipErrorInvalidValue;
  if (!p.a_hi || !p.a_lo || !p.rowstat || !p.wg_hi || !p.wg_lo || !p.bg || !p.ug || !p.wm_hi || !p.wm_lo || !p.bm || !p.res ||
      (p.out_hi && !p.out_lo))
    return hipErrorInvalidValue;
  if (p.gnx.xchg) {
    static const int n_cu = [] { int d = 0, n = 0; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
    if (!p.gnx.status || !p.gnx.y_hi || !p.gnx.y_lo || !p.gnx.gamma || !p.gnx.beta || chain_ff_gnx_plan(p, n_cu) <= 0) return hipErrorInvalidValue;
    if (p.gnx.sk_c > 0 && (!p.gnx.sk_x || !p.gnx.sk_stat16 || !p.gnx.sk_y_hi || !p.gnx.sk_y_lo)) return hipErrorInvalidValue;
  } else if (!p.out) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_chain_ff, dim3(p.M / BM), dim3(NT_LAUNCH), FF_SMEM, st, p);
  return hipGetLastError();
}

hipError_t chain_init() {
  hipError_t e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_ff), hipFuncAttributeMaxDynamicSharedMemorySize, FF_SMEM)) != hipSuccess) return e;
  if ((e = init_one<1, 0>()) != hipSuccess) return e;
  if ((e = init_one<2, 0>()) != hipSuccess) return e;
  if ((e = init_one<3, 0>()) != hipSuccess) return e;
  if ((e = init_one<1, 1>()) != hipSuccess) return e;
  if ((e = init_one<2, 1>()) != hipSuccess) return e;
  if ((e = init_one<3, 1>()) != hipSuccess) return e;
  if ((e = init_one<1, 1, false, true>()) != hipSuccess) return e;
  if ((e = init_one<2, 1, false, true>()) != hipSuccess) return e;
  if ((e = init_one<3, 1, false, true>()) != hipSuccess) return e;
  if ((e = init_one<2, 0, false, false, true>()) != hipSuccess) return e;
  if ((e = init_one<3, 0, false, false, true>()) != hipSuccess) return e;
  if ((e = init_one<1, 0, true>()) != hipSuccess) return e;
  if ((e = init_one<2, 0, true, false, true>()) != hipSuccess) return e;
  return init_one<2, 0, true>();   // (C = 384 with the attention inside needs > 256 VGPRs: it keeps the separate launches)
}

hipError_t launch_kv_frag(const float* kv, bf16_t* kf_hi, bf16_t* kf_lo, bf16_t* vf_hi, bf16_t* vf_lo, int B, int L, int C, int H,
                          hipStream_t st) {
  if (H <= 0 || C % H != 0 || (C / H) % 16 != 0 || C / H > 64) return hipErrorInvalidValue;
  const int nT = (L + 31) / 32;
  hipLaunchKernelGGL(k_kv_frag, dim3(nT, H, B), dim3(64), 0, st, kv, kf_hi, kf_lo, vf_hi, vf_lo, L, C, H, nT);
  return hipGetLastError();
}
hipError_t launch_xbias(const float* mask_bias, float* out, int B, int L, int nT, hipStream_t st) {
  const int n = B * nT * 32;
  hipLaunchKernelGGL(k_xbias, dim3((n + 255) / 256), dim3(256), 0, st, mask_bias, out, B, L, nT * 32);
  return hipGetLastError();
}

hipError_t launch_relayout_frag(const bf16_t* src, bf16_t* dst, int rows, int Kp, hipStream_t st) {
  if (rows % 32 != 0 || Kp % 16 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_relayout_frag, dim3(rows / 32, Kp / 16), dim3(64), 0, st, src, dst, Kp, Kp / 16);
  return hipGetLastError();
}

bool chain2_supported(const ChainParams& p, int precision) {
  if (precision != 0) return false;                                  // split-bf16 mode only
  if (p.C != 128 && p.C != 256 && p.C != 384) return false;   // (C = 512 was measured slower than one launch per GEMM; its instantiations were removed in round 3)
  if (p.M % 32 != 0 || p.T % 32 != 0 || p.M % p.T != 0 || p.passes < 1) return false;
  if (p.Tv < 0 || p.Tv > p.T || (p.Tv > 0 && p.Tv <= p.T - 32)) return false;   // (padded row spaces: every row block holds a frame that exists)
  if (p.Kp1 != p.C || p.Kp2 != p.C) return false;
  if (p.xa_kf_hi) {                                                    // cross-attention tail: wave = head
    if (p.amode != 0 || p.passes != 1 || p.C % 8 != 0) return false;
    const int d = p.C / 8;
    if (d != p.xa_d || (d != 16 && d != 32) || p.xa_nT < 1) return false;
    if (!p.xa_kf_lo || !p.xa_vf_hi || !p.xa_vf_lo || !p.xa_bias || !p.w3_hi || !p.w3_lo || !p.b3 || !p.out3 || !p.out3_hi || !p.out3_lo ||
        !p.rowstat3)
      return false;
  }
  if (p.nsplit < 0 || p.nsplit > 3) return false;
  if (p.nsplit > 1 && p.amode == 1 && p.nsplit > p.passes) return false;
  // amode 0: the parts share out the 128-column groups of the one stage-2 pass
  if (p.nsplit > 1 && p.amode == 0 && p.xa_kf_hi) {                    // cross-attention tail on two workgroups per row block
    if (p.C != 256 || p.nsplit != 2 || p.passes != 1 || !p.xs_buf || !p.xs_ticket) return false;
  } else
  if (p.nsplit > 1 && p.amode == 0 && (p.passes != 1 || p.nsplit != p.C / 128 || p.C > 384)) return false;
  if (p.sa_kf_hi) {                                                    // q | k | v with K / V as attention fragments
    if (p.amode != 1 || p.passes != 3 || p.C > 384 || !p.sa_kf_lo || !p.sa_vf_hi || !p.sa_vf_lo) return false;
  }
  if (p.amode == 1) {
    const int G = p.groups;
    if (G <= 0 || G > 64 || (G & (G - 1)) != 0 || p.C % G != 0 || (p.C / G) % 16 != 0) return false;
    if ((p.T / 32) * (p.C / 16) > (p.C / 128) * 2048) return false;  // block entries of one utterance are staged in the hand-over region
  }
  return true;
}

hipError_t launch_chain2(const ChainParams& p, int precision, hipStream_t st) {
  if (!chain2_supported(p, precision)) return hipErrorInvalidValue;
  if (!p.w1_hi || !p.w1_lo || !p.w2_hi || !p.w2_lo || !p.b1 || !p.b2 || !p.u2 || !p.out1 || (!p.out2 && !p.xa_kf_hi)) return hipErrorInvalidValue;
  if (p.amode == 0 ? (!p.a_hi || !p.a_lo) : (!p.x || !p.stat16 || !p.gamma || !p.beta)) return hipErrorInvalidValue;
  const int ns = p.C / 128;
  if (p.xa_kf_hi && p.nsplit == 2) return launch_one<2, 0, true, false, true>(p, st);   // (two workgroups per row block: heads 0-3 / 4-7)
  if (p.xa_kf_hi) return ns == 1 ? launch_one<1, 0, true>(p, st) : launch_one<2, 0, true>(p, st);
  if (p.sa_kf_hi) return ns == 1 ? launch_one<1, 1, false, true>(p, st) : (ns == 2 ? launch_one<2, 1, false, true>(p, st) : launch_one<3, 1, false, true>(p, st));
  if (p.amode == 0 && p.nsplit > 1) return ns == 2 ? launch_one<2, 0, false, false, true>(p, st) : launch_one<3, 0, false, false, true>(p, st);
  if (p.amode == 0)
    return ns == 1 ? launch_one<1, 0>(p, st) : (ns == 2 ? launch_one<2, 0>(p, st) : launch_one<3, 0>(p, st));
  return ns == 1 ? launch_one<1, 1>(p, st) : (ns == 2 ? launch_one<2, 1>(p, st) : launch_one<3, 1>(p, st));
}
