// Head of a transformer block at C = 128 / 256 / 384 for gfx950 (MI355X) as ONE launch of 64-row blocks whose COLUMNS are split over
// the workgroups: GroupNorm -> proj_in -> LayerNorm1 -> to_q | to_k | to_v (reference unet1d/transformer_1d.py:262-268: norm, proj_in;
// attention.py:157-160: norm1, attn1's projections) - the work of k_chain2<NS, 1, false, true> (kernels_chain.hip, "chain 1"), same
// ChainParams, same results layout (h fp32, q fp32, K / V^T as MFMA fragments of 32-key tiles).  The same launch form also runs the
// two tails of the self attention (template parameter MODE, below): attn1.to_out + residual -> LN2 -> attn2.to_q (MODE 1, C = 384) and
// the whole cross-attention chain ... -> cross attention -> attn2.to_out + residual -> LN3 partials (MODE 2, C = 256: the slice's heads
// attend inside the launch, the attention output is a second hand-over).
//
// Why.  The row-block chain owns 32 rows and ALL channels: a workgroup streams every weight of both contractions - 4 C^2 x 4 bytes,
// 2.4 MB at C = 384 - for 32 rows, each weight fragment feeds ONE row fragment, and the k-loops run at the rate the weights arrive
// (40-60 GB/s per CU: 60-85 TFLOP/s, 15 launches, 0.37 ms of a 2.48 ms forward at the bench shape; VERDICT r5 missing #4).  Sharing
// the stage-2 passes out over three workgroups (round 3) cut the stream per workgroup but repeated stage 1 in each.  Here, as in
// k_ff_split (kernels_ffsplit.hip):
//   * 64 rows per workgroup: two row fragments per weight fragment (half the weight bytes per MFMA);
//   * the OUTPUT COLUMNS of both contractions are split over C / 64 workgroups per row block (workgroup id = row block * nspl +
//     slice: XCD x only ever touches slices x mod nspl of the weights): slice s computes columns [64 s, 64 s + 64) of h = proj_in(GN(x))
//     and, once the row block's h is complete, the same 64 columns of q, of k and of v - a workgroup streams 4 C x 64 x 4 bytes
//     (0.39 MB at C = 384) instead of 4 C^2 x 4;
//   * every contraction is the 64 x 64 tile of k_conv3 (kernels_conv.hip): the operand rows resident in LDS, the weights fragment-
//     major straight into registers, 8 waves = 2 column fragments x 4 k-quarters, no barrier inside a k-loop, the quarters added
//     through LDS.
// The price is an all-gather of h inside the launch: every workgroup writes its 64 columns of h through (it is the block's residual
// stream anyway: ChainParams out1), raises one flag word, waits for the flags of its row block (C / 64 consecutive workgroup ids:
// gemm_handover_rounds, kernels_gemm.hip; bounded and flagged like every in-launch hand-over) and reads the 64 x C rows back - fp32,
// so that LayerNorm1's row statistics are formed from exactly the values the reference normalises.
#include "dv_common.h"
#include "dv_device.h"

#include <cstdlib>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#ifdef DV_GEMM_TRACE
// development build only (make trace): per-workgroup s_memtime stamps of the phases
__device__ unsigned long long g_qkv_trace[1024 * 16];
__device__ int g_qkv_sel = 0;                      // which launches stamp: C (0 = any)
#define DV_QTRACE(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024 && (g_qkv_sel == 0 || g_qkv_sel == C)) g_qkv_trace[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int dv_debug_qkv_trace_select(int c) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_qkv_sel), &c, sizeof(c));
  void* d = nullptr;
  if (e == hipSuccess) e = hipGetSymbolAddress(&d, HIP_SYMBOL(g_qkv_trace));
  return (int)(e != hipSuccess ? e : hipMemset(d, 0, sizeof(g_qkv_trace)));
}
extern "C" int dv_debug_qkv_trace(unsigned long long* host, int n_wg) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_qkv_trace), (size_t)n_wg * 16 * sizeof(unsigned long long));
}
#else
#define DV_QTRACE(i) do {} while (0)
#endif

namespace {

constexpr int NWV = 8, NT = 64 * NWV, BM = 64, BN = 64;
constexpr int CHP = BM * 128;                        // bytes of one 64-channel chunk of one plane of the resident rows
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ unsigned pk(float lo, float hi) { return dv_cvt_pk_bf16(lo, hi); }
struct BFrag { bf16x8 h, l; };

template <int C>
struct QGeom {
  static constexpr int NSPL = C / BN;                // workgroups per row block
  static constexpr int KS = C / 16, U = KS / 4;      // 16-deep k-steps of a contraction (K = C), per k-quarter (wave)
  static constexpr int DEPTH = U < 6 ? U : 6;        // weight units (hi + lo fragment: 8 VGPRs) in flight per wave
  static constexpr int A_CH = C / 64, A_PL = A_CH * CHP;
  // k-quarter exchange: [column fragment][quarter][row fragment][register group][64 lanes] float4 - both row fragments at once
  // (64 KiB) where the resident rows leave room (C <= 256), one row fragment per round (32 KiB) at C = 384
  static constexpr bool ONE = 2 * A_PL + 65536 + 8192 <= 160 * 1024;
  static constexpr int RED = ONE ? 65536 : 32768;
  static constexpr int SMEM = 2 * A_PL + RED;
  static constexpr int D3 = 3 * U < 9 ? 3 * U : 9;   // weight units in flight per wave in the merged q | k | v loop
  static constexpr int JT = C / 32;                  // float4 pieces of a row per thread (thread = (row, eighth): 8 threads per row)
  static constexpr int ENT_MAX = RED / 8;            // GroupNorm block-statistics entries of one utterance that fit the exchange region
  static_assert(C % 128 == 0 && KS % 4 == 0 && SMEM + 8192 <= 160 * 1024, "geometry");
};

// MODE 0: the block head (GroupNorm of the fp32 rows -> proj_in -> LN1 -> q | K | V fragments; ChainParams amode 1 with sa_*).
// MODE 1: the self-attention tail (reference attention.py:157-189: attn1.to_out + residual -> LN2 -> attn2.to_q; ChainParams amode 0,
//         one pass): the rows arrive as split planes (LDS-DMA), stage 1 adds the residual, stage 2 is the query contraction alone.
template <int C, int MODE>
__global__ __launch_bounds__(NT) void k_qkv_split(const ChainParams p) {
  using G = QGeom<C>;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* const a_reg = smem;                          // the resident rows: [2 planes][C / 64 chunks][64 rows][128 B]
  char* const red_reg = smem + 2 * G::A_PL;          // k-quarter exchange; the utterance's GroupNorm entries before stage 1
  __shared__ __attribute__((aligned(16))) float s_gscale[C], s_gshift[C];
  __shared__ __attribute__((aligned(16))) float2 s_ln[BM];   // per row (mean, rstd) of h
  // (every 64-byte line of the argument block is requested at once: see k_gemm)
  asm volatile("" ::"s"(p.M), "s"(p.gamma), "s"(p.w1_lo), "s"(p.out1), "s"(p.u2), "s"(p.sa_kf_hi), "s"(p.sa_vf_lo), "s"(p.qs_flags));
  DV_QTRACE(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int cf = wave & 1, kq = wave >> 1;           // k-loop role: column fragment of the slice, k-quarter
  const int fg = wave >> 1;                          // finishing role: register group fg of fragment cf (a 32-row x 8-column piece per row fragment)
  // Workgroup -> (row block, slice).  XCD-local form (p.qs_xcd, whole multiples of 8 row blocks): workgroup b runs on XCD b % 8, and
  // the C / 64 slices of a row block are given ids that differ by multiples of 8 - they share an L2: the rows of x are fetched into
  // it once, and h is handed over THROUGH it (plain stores, L1-bypassing `sc1` loads: the coherence rule of persist.hip) instead of
  // through memory.  The placement is verified, not assumed: every flag carries its writer's XCC id (see the wait below).
  const bool xl = p.qs_xcd != 0;
  int rb, s;
  if (xl) { const int x = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3, rbl = i / G::NSPL; s = i - rbl * G::NSPL; rb = rbl * 8 + x; }
  else { rb = (int)blockIdx.x / G::NSPL; s = (int)blockIdx.x - rb * G::NSPL; }
  const int m0 = rb * BM;
  unsigned xcc = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 0xfu;
  const int r_row = tid >> 3, r_e8 = tid & 7;        // row pass role: row of the block, eighth of its float4 pieces (coalesced: 8 lanes = 128 bytes)

  // weight unit j of this wave for column fragment nf (fragment-major [nf][k-step][64 lanes][8])
  auto load_unit = [&](const bf16_t* wf_hi, const bf16_t* wf_lo, int nf, int j) {
    const size_t e = ((size_t)(nf * G::KS + kq * G::U + j) * 64 + lane) * 8;
    BFrag f;
    f.h = *reinterpret_cast<const bf16x8*>(wf_hi + e);
    f.l = *reinterpret_cast<const bf16x8*>(wf_lo + e);
    return f;
  };
  BFrag bq[G::DEPTH];
  auto job_prologue = [&](const bf16_t* wf_hi, const bf16_t* wf_lo, int nf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < G::DEPTH; ++j) bq[j] = load_unit(wf_hi, wf_lo, nf, j);
    __builtin_amdgcn_sched_barrier(0);
  };
  // one 64 x 64 tile over the resident rows: acc[rf] += W[nf][k-quarter kq] x A^T[row fragment rf] (SWAP: the rows are the first
  // operand - the accumulator is then the tile itself: lane = output column, registers = rows 8g + 4lh + e; used for V)
  f32x16 acc[2];
  auto job_loop = [&](const bf16_t* wf_hi, const bf16_t* wf_lo, int nf, auto swap_tag) __attribute__((always_inline)) {
    constexpr bool SWAP = decltype(swap_tag)::value;
#pragma unroll
    for (int rf = 0; rf < 2; ++rf)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rf][r] = 0.f;
    auto read_a = [&](int u, bf16x8 (&h)[2], bf16x8 (&l)[2]) {
      const int c16 = (kq * G::U + u) * 2 + lh;      // 16-byte piece of the row: k-step * 2 + half
#pragma unroll
      for (int rf = 0; rf < 2; ++rf) {
        const int row = rf * 32 + l31;
        const int off = (c16 >> 3) * CHP + row * 128 + (((c16 & 7) ^ swz(row)) << 4);
        h[rf] = *reinterpret_cast<const bf16x8*>(a_reg + off);
        l[rf] = *reinterpret_cast<const bf16x8*>(a_reg + G::A_PL + off);
      }
    };
    bf16x8 ah[2][2], al[2][2];
    read_a(0, ah[0], al[0]);
#pragma unroll
    for (int u = 0; u < G::U; ++u) {
      const int cur = u & 1;
      if (u + 1 < G::U) read_a(u + 1, ah[cur ^ 1], al[cur ^ 1]);
      const BFrag f = bq[u % G::DEPTH];
      if (SWAP) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cur][0], f.h, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cur][1], f.h, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][0], f.l, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][1], f.l, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][0], f.h, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][1], f.h, acc[1], 0, 0, 0);
      } else {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, al[cur][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, al[cur][1], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.l, ah[cur][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.l, ah[cur][1], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, ah[cur][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, ah[cur][1], acc[1], 0, 0, 0);
      }
      // pinned: without the scheduling barriers hipcc sinks every prefetch down to its use (load -> vmcnt(0) -> MFMA)
      __builtin_amdgcn_sched_barrier(0);
      if (u + G::DEPTH < G::U) bq[u % G::DEPTH] = load_unit(wf_hi, wf_lo, nf, u + G::DEPTH);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // The four k-quarters are added through LDS (quarters in order: deterministic); wave (cf, fg) receives register group fg of
  // fragment cf for both row fragments: v[rf] = rows l31 x columns 8 fg + 4 lh + e of the fragment (SWAP: column l31 x rows ...)
  float4* const red4 = reinterpret_cast<float4*>(red_reg);
  auto pieces = [&](const f32x16& a0, const f32x16& a1, float4 (&v)[2]) __attribute__((always_inline)) {
    if constexpr (G::ONE) {
      __syncthreads();                               // the exchange region is free; every wave has left the k-loop
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        red4[((((cf * 4 + kq) * 2 + 0) * 4 + g) << 6) + lane] = make_float4(a0[4 * g], a0[4 * g + 1], a0[4 * g + 2], a0[4 * g + 3]);
        red4[((((cf * 4 + kq) * 2 + 1) * 4 + g) << 6) + lane] = make_float4(a1[4 * g], a1[4 * g + 1], a1[4 * g + 2], a1[4 * g + 3]);
      }
      __syncthreads();
#pragma unroll
      for (int rf = 0; rf < 2; ++rf) {
        float4 t = red4[((((cf * 4 + 0) * 2 + rf) * 4 + fg) << 6) + lane];
#pragma unroll
        for (int k = 1; k < 4; ++k) {
          const float4 w = red4[((((cf * 4 + k) * 2 + rf) * 4 + fg) << 6) + lane];
          t.x += w.x; t.y += w.y; t.z += w.z; t.w += w.w;
        }
        v[rf] = t;
      }
    } else {
#pragma unroll
      for (int rf = 0; rf < 2; ++rf) {
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g)
          red4[(((cf * 4 + kq) * 4 + g) << 6) + lane] = rf == 0 ? make_float4(a0[4 * g], a0[4 * g + 1], a0[4 * g + 2], a0[4 * g + 3])
                                                                 : make_float4(a1[4 * g], a1[4 * g + 1], a1[4 * g + 2], a1[4 * g + 3]);
        __syncthreads();
        float4 t = red4[(((cf * 4 + 0) * 4 + fg) << 6) + lane];
#pragma unroll
        for (int k = 1; k < 4; ++k) {
          const float4 w = red4[(((cf * 4 + k) * 4 + fg) << 6) + lane];
          t.x += w.x; t.y += w.y; t.z += w.z; t.w += w.w;
        }
        v[rf] = t;
      }
    }
  };
  // fp32 rows -> split planes of the resident operand: thread (r_row, r_e8) owns the float4 pieces 4 (r_e8 + 8 j) .. + 3 of its row
  auto put_planes = [&](int j, float v0, float v1, float v2, float v3) __attribute__((always_inline)) {
    const int ch = 4 * (r_e8 + 8 * j);
    uint2 hw, lw;
    hw.x = pk(v0, v1); hw.y = pk(v2, v3);
    lw.x = pk(v0 - __uint_as_float(hw.x << 16), v1 - __uint_as_float(hw.x & 0xffff0000u));
    lw.y = pk(v2 - __uint_as_float(hw.y << 16), v3 - __uint_as_float(hw.y & 0xffff0000u));
    const int off = (ch >> 6) * CHP + r_row * 128 + ((((ch & 63) >> 3) ^ swz(r_row)) << 4) + ((ch & 7) >> 2) * 8;
    *reinterpret_cast<uint2*>(a_reg + off) = hw;
    *reinterpret_cast<uint2*>(a_reg + G::A_PL + off) = lw;
  };

  // ================= the rows of the block -> LDS (MODE 0: GroupNorm of the fp32 rows; MODE 1: planes by DMA), stage-1 weights =================
  float4 rv[G::JT];
  // bias of this lane's finished stage-1 columns (cold: requested at the head, not in front of its use)
  const float4 b1v = *reinterpret_cast<const float4*>(p.b1 + s * BN + cf * 32 + 8 * fg + 4 * lh);
  float4 rres[2];                                    // MODE 1 / 2: the residual rows of this lane's finished pieces (cold: requested first)
  float4 h2v[2];                                     // MODE 2: this lane's finished columns of x1 = to_out + residual (stage 3 adds them)
  if constexpr (MODE >= 1) {
    const int ncol_ = s * BN + cf * 32 + 8 * fg + 4 * lh;
#pragma unroll
    for (int rf = 0; rf < 2; ++rf)
      rres[rf] = p.res ? *reinterpret_cast<const float4*>(p.res + (size_t)(m0 + rf * 32 + l31) * C + ncol_) : make_float4(0.f, 0.f, 0.f, 0.f);
    // the rows as split planes by LDS-DMA: instruction = (64-channel chunk, 8 rows) of one plane, wave w sends rows 8 w .. 8 w + 7
    const unsigned a_base = (unsigned)(size_t)a_reg;
    const int d_row = wave * 8 + (lane >> 3), d_slot = lane & 7;
#pragma unroll
    for (int c = 0; c < G::A_CH; ++c) {
      const size_t e = (size_t)(m0 + d_row) * C + c * 64 + ((d_slot ^ swz(d_row)) << 3);
      const unsigned dst = a_base + (unsigned)(c * CHP + wave * 1024);
      glds16(p.a_hi + e, dst);
      glds16(p.a_lo + e, dst + G::A_PL);
    }
    job_prologue(p.w1_hi, p.w1_lo, s * 2 + cf);
    DV_QTRACE(10);
    wait_vmcnt<2 * G::DEPTH>();                      // everything older than the weight units: this wave's rows have landed
  } else {
  // ================= rows of x, GroupNorm table of the utterance, stage-1 weights: requested together =================
  const int T = p.T, b_item = m0 / T, Tv = p.Tv > 0 ? p.Tv : T;   // row pitch / frames that exist (padded row spaces)
  {
    const float* xr = p.x + (size_t)(m0 + r_row) * C + 4 * r_e8;
#pragma unroll
    for (int j = 0; j < G::JT; ++j) rv[j] = *reinterpret_cast<const float4*>(xr + 32 * j);
  }
  job_prologue(p.w1_hi, p.w1_lo, s * 2 + cf);
  DV_QTRACE(10);
  {
    // GroupNorm (eps 1e-6, no activation) of this utterance from its 32 x 16 block statistics (k_chain2, amode 1)
    const int Gn = p.groups, cg = C / Gn, nvb = cg >> 4, RB = T >> 5, nblk = C >> 4, n_ent = RB * nblk;
    const int cc = min(tid, C - 1);
    const float pg = p.gamma[cc], pb = p.beta[cc];
    float2* s_ent = reinterpret_cast<float2*>(red_reg);
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
    DV_QTRACE(11);
    const int lpg = 64 / Gn, g = lane / lpg, sub = lane - g * lpg;
    const double inv_last = 1.0 / (double)(16 * min(32, Tv - 32 * (RB - 1)));
    double s1 = 0, q = 0;
    for (int i0 = sub; i0 < RB * nvb; i0 += 4 * lpg) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * lpg, rbk = i / nvb;
        const float2 v = s_ent[min(rbk * nblk + g * nvb + (i - rbk * nvb), n_ent - 1)];
        // (a block's sum of squares = M2 + sum^2 / count; the count is 512 except in an utterance's last, partial block: no fp64 division per entry)
        if (i < RB * nvb) { s1 += (double)v.x; q += (double)v.y + (double)v.x * (double)v.x * (rbk == RB - 1 ? inv_last : (1.0 / 512.0)); }
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
    DV_QTRACE(12);
  }
#pragma unroll
  for (int j = 0; j < G::JT; ++j) {
    const int ch = 4 * (r_e8 + 8 * j);
    const float4 sc = *reinterpret_cast<const float4*>(s_gscale + ch);
    const float4 sh = *reinterpret_cast<const float4*>(s_gshift + ch);
    put_planes(j, fmaf(rv[j].x, sc.x, sh.x), fmaf(rv[j].y, sc.y, sh.y), fmaf(rv[j].z, sc.z, sh.z), fmaf(rv[j].w, sc.w, sh.w));
  }
  }
  DV_QTRACE(1);
  __syncthreads();                                   // GN(x) complete in LDS
  DV_QTRACE(2);

  // ================= stage 1: columns [64 s, 64 s + 64) of h = GN(x) W1^T + b1 =================
  job_loop(p.w1_hi, p.w1_lo, s * 2 + cf, std::false_type{});
  DV_QTRACE(3);
  const int ncol = s * BN + cf * 32 + 8 * fg + 4 * lh;   // this lane's four columns of a finished piece (normal orientation)
  {
    const float4 b4 = b1v;
    float4 pv[2];
    pieces(acc[0], acc[1], pv);
#pragma unroll
    for (int rf = 0; rf < 2; ++rf) {
      float4 v = pv[rf];
      v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
      if constexpr (MODE >= 1) {
        const float4 r4 = rf == 0 ? rres[0] : rres[1];
        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
        if (rf == 0) h2v[0] = v; else h2v[1] = v;      // (MODE 2: the residual of stage 3 - this lane's own columns of x1)
      }
      // the other workgroups of the row block read it back below (XCD-local: from the shared L2 - a plain store reaches it, the
      // L1 is write-through; else written through to memory), later launches read it as the residual stream
      float4* const dst = reinterpret_cast<float4*>(p.out1 + (size_t)(m0 + rf * 32 + l31) * C + ncol);
      if (xl) *dst = v; else st_handover16(dst, v);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's part of h has reached the L2 / memory
  __syncthreads();
  if (tid == 0) {
    const unsigned long long fv = 1ull + xcc;        // (the flag says where it was written)
    if (xl) *reinterpret_cast<volatile unsigned long long*>(p.qs_flags + (size_t)rb * G::NSPL + s) = fv;
    else __hip_atomic_store(p.qs_flags + (size_t)rb * G::NSPL + s, fv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // stage 2's first weight fragments fly during the hand-over (MODE 0: of the merged q | k | v loop below)
  const int nfq = s * 2 + cf, nfk = C / 32 + s * 2 + cf, nfv = 2 * (C / 32) + s * 2 + cf;
  BFrag b3[MODE == 0 ? G::D3 : 1];
  auto load_unit3 = [&](int j) {                     // unit j of the merged loop: k-step j / 3, pass j % 3 (q, k, v)
    const int u = j / 3, ps = j - 3 * u;
    return load_unit(p.w2_hi, p.w2_lo, ps == 0 ? nfq : (ps == 1 ? nfk : nfv), u);
  };
  if constexpr (MODE == 0) {
#pragma unroll
    for (int j = 0; j < G::D3; ++j) b3[j] = load_unit3(j);
    __builtin_amdgcn_sched_barrier(0);
  } else job_prologue(p.w2_hi, p.w2_lo, nfq);
  DV_QTRACE(4);
  // LayerNorm vectors of this lane's finished columns (cold; independent of the hand-over)
  const float4 uq = *reinterpret_cast<const float4*>(p.u2 + ncol), bq4 = *reinterpret_cast<const float4*>(p.b2 + ncol);
  float4 uk = uq, bk4 = bq4;
  float uvv = 0.f, bvv = 0.f;
  if constexpr (MODE == 0) {
    uk = *reinterpret_cast<const float4*>(p.u2 + C + ncol); bk4 = *reinterpret_cast<const float4*>(p.b2 + C + ncol);
    const int vch = s * BN + cf * 32 + l31;          // this lane's channel of a finished V piece (swapped orientation)
    uvv = p.u2[2 * C + vch]; bvv = p.b2[2 * C + vch];
  }

  // ================= all-gather of h: wait for the row block's flags (every wave polls for itself), rows back as fp32 =================
  auto wait_flags = [&](const unsigned long long* fl, unsigned code) __attribute__((always_inline)) {
    for (int spins = 0;; ++spins) {
      bool ok = true;
      unsigned long long fv = 0;
      if (lane < G::NSPL) {
        if (xl) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(fv) : "v"(fl + lane) : "memory");
        else fv = __hip_atomic_load(fl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok = fv != ~0ull;
      }
      if (__all(ok)) {
        // XCD-local hand-over: every partner must have run on THIS XCD (its stores sit in this L2).  A placement other than
        // "workgroup b on XCD b % 8" is reported like a timed-out wait: the run is repeated on the fallback schedule
        if (xl && __any(lane < G::NSPL && fv != 1ull + xcc)) {
          if (lane == 0) {
            p.qs_status[1] = (unsigned)(size_t)p.qs_flags; p.qs_status[2] = blockIdx.x; p.qs_status[3] = 0xc2u; p.qs_status[4] = xcc;
            __hip_atomic_store(p.qs_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
        break;
      }
      const bool lost = (spins & 63) == 63 && __hip_atomic_load(p.qs_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
      if (lost) break;
      if (spins > p.qs_spin) {
        if (lane == 0) {
          p.qs_status[1] = (unsigned)(size_t)p.qs_flags; p.qs_status[2] = blockIdx.x; p.qs_status[3] = code; p.qs_status[4] = (unsigned)__builtin_popcountll(__ballot(!ok));
          __hip_atomic_store(p.qs_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
  };
  wait_flags(p.qs_flags + (size_t)rb * G::NSPL, 0xc1u);
  DV_QTRACE(5);
  {
    const float4* hr = reinterpret_cast<const float4*>(p.out1 + (size_t)(m0 + r_row) * C + 4 * r_e8);
    if (xl) {
      // L1-bypassing loads served by the XCD's L2, all of the row's pieces in flight behind ONE wait (the destination registers of an
      // asm load are only safe to read behind a wait inside the same block)
      if constexpr (G::JT == 4)
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:128 sc1\n\t"
                     "global_load_dwordx4 %2, %4, off offset:256 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:384 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(rv[0]), "=&v"(rv[1]), "=&v"(rv[2]), "=&v"(rv[3]) : "v"(hr) : "memory");
      else if constexpr (G::JT == 8)
        asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:128 sc1\n\t"
                     "global_load_dwordx4 %2, %8, off offset:256 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:384 sc1\n\t"
                     "global_load_dwordx4 %4, %8, off offset:512 sc1\n\tglobal_load_dwordx4 %5, %8, off offset:640 sc1\n\t"
                     "global_load_dwordx4 %6, %8, off offset:768 sc1\n\tglobal_load_dwordx4 %7, %8, off offset:896 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(rv[0]), "=&v"(rv[1]), "=&v"(rv[2]), "=&v"(rv[3]), "=&v"(rv[4]), "=&v"(rv[5]), "=&v"(rv[6]), "=&v"(rv[7]) : "v"(hr) : "memory");
      else
        asm volatile("global_load_dwordx4 %0, %12, off sc1\n\tglobal_load_dwordx4 %1, %12, off offset:128 sc1\n\t"
                     "global_load_dwordx4 %2, %12, off offset:256 sc1\n\tglobal_load_dwordx4 %3, %12, off offset:384 sc1\n\t"
                     "global_load_dwordx4 %4, %12, off offset:512 sc1\n\tglobal_load_dwordx4 %5, %12, off offset:640 sc1\n\t"
                     "global_load_dwordx4 %6, %12, off offset:768 sc1\n\tglobal_load_dwordx4 %7, %12, off offset:896 sc1\n\t"
                     "global_load_dwordx4 %8, %12, off offset:1024 sc1\n\tglobal_load_dwordx4 %9, %12, off offset:1152 sc1\n\t"
                     "global_load_dwordx4 %10, %12, off offset:1280 sc1\n\tglobal_load_dwordx4 %11, %12, off offset:1408 sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(rv[0]), "=&v"(rv[1]), "=&v"(rv[2]), "=&v"(rv[3]), "=&v"(rv[4]), "=&v"(rv[5]), "=&v"(rv[6]), "=&v"(rv[7]),
                       "=&v"(rv[8 % G::JT]), "=&v"(rv[9 % G::JT]), "=&v"(rv[10 % G::JT]), "=&v"(rv[11 % G::JT])
                     : "v"(hr) : "memory");
    } else {
#pragma unroll
      for (int j = 0; j < G::JT; ++j) rv[j] = ld_handover16(hr + 8 * j);
    }
    // LayerNorm1 statistics of the row from the fp32 values (two passes over registers; 8 threads per row)
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < G::JT; ++j) s1 += (rv[j].x + rv[j].y) + (rv[j].z + rv[j].w);
    s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2); s1 += __shfl_xor(s1, 4);
    const float mean = s1 * (1.0f / (float)C);
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < G::JT; ++j) {
      const float d0 = rv[j].x - mean, d1 = rv[j].y - mean, d2 = rv[j].z - mean, d3 = rv[j].w - mean;
      m2 = fmaf(d0, d0, m2); m2 = fmaf(d1, d1, m2); m2 = fmaf(d2, d2, m2); m2 = fmaf(d3, d3, m2);
    }
    m2 += __shfl_xor(m2, 1); m2 += __shfl_xor(m2, 2); m2 += __shfl_xor(m2, 4);
    if (r_e8 == 0) s_ln[r_row] = make_float2(mean, 1.0f / sqrtf(m2 * (1.0f / (float)C) + p.ln_eps));
    // (raw planes: LayerNorm's gamma / beta are folded into W2 / b2 / u2, its mean / rstd are applied in the epilogues)
#pragma unroll
    for (int j = 0; j < G::JT; ++j) put_planes(j, rv[j].x, rv[j].y, rv[j].z, rv[j].w);
  }
  __syncthreads();                                   // h complete in LDS, row statistics visible
  DV_QTRACE(6);

  // ================= stage 2: the same 64 columns of q (MODE 0: and of k and of v) =================
  auto finish_q = [&](const float4 (&pv)[2]) __attribute__((always_inline)) {     // fp32 [M, ldo2]
#pragma unroll
    for (int rf = 0; rf < 2; ++rf) {
      const float4 v = pv[rf];
      const float2 st = s_ln[rf * 32 + l31];
      float4 o;
      o.x = st.y * (v.x - st.x * uq.x) + bq4.x; o.y = st.y * (v.y - st.x * uq.y) + bq4.y;
      o.z = st.y * (v.z - st.x * uq.z) + bq4.z; o.w = st.y * (v.w - st.x * uq.w) + bq4.w;
      dv_st16(p.out2 + (size_t)(m0 + rf * 32 + l31) * p.ldo2 + ncol, o);
    }
  };
  if constexpr (MODE == 1) {
    job_loop(p.w2_hi, p.w2_lo, nfq, std::false_type{});
    float4 pv[2];
    pieces(acc[0], acc[1], pv);
    finish_q(pv);
    DV_QTRACE(7); DV_QTRACE(8); DV_QTRACE(9);
    return;
  } else if constexpr (MODE == 2) {
    // ================= MODE 2: the cross attention of the block inside the launch (reference attention.py:176-189: attn2 over the
    // prompt's keys / values - hoisted by set_cond into MFMA-fragment order, ChainParams xa_* - then attn2.to_out + residual) =================
    // The slice's 64 query columns are 64 / d whole heads (d = C / 8 = 16 / 32).  Jobs = (head of the slice, row fragment): eight at
    // C = 128 - one per wave - four at C = 256, two waves per job on the key tiles of either parity, merged through LDS.  The
    // arithmetic is k_chain2's (S^T = K Q^T and O^T += V^T P^T, split operands, scores in the log2 domain, online softmax lane-local).
    constexpr int d = C / 8, KSq = d / 16, HS = BN / d, KP = (2 * HS < NWV) ? 2 : 1;    // head dim, k-steps of a score, heads per slice, key-parity waves per job
    static_assert(C <= 256 && 2 * HS * KP == NWV, "MODE 2 geometry");
    job_loop(p.w2_hi, p.w2_lo, nfq, std::false_type{});
    job_prologue(p.w3_hi, p.w3_lo, s * 2 + cf);      // the output projection's first weight fragments fly under the attention
    {
      float4 pv[2];
      pieces(acc[0], acc[1], pv);                    // (its barriers: every wave has left the k-loop - the rows of x1 are dead)
      // the slice's queries, pre-scaled (d^-1/2 log2 e), as split planes in chunk 0 of the row region
#pragma unroll
      for (int rf = 0; rf < 2; ++rf) {
        const float4 v = pv[rf];
        const float2 st = s_ln[rf * 32 + l31];
        const float q0 = (st.y * (v.x - st.x * uq.x) + bq4.x) * p.xa_qscale, q1 = (st.y * (v.y - st.x * uq.y) + bq4.y) * p.xa_qscale;
        const float q2 = (st.y * (v.z - st.x * uq.z) + bq4.z) * p.xa_qscale, q3 = (st.y * (v.w - st.x * uq.w) + bq4.w) * p.xa_qscale;
        uint2 hw, lw;
        hw.x = pk(q0, q1); hw.y = pk(q2, q3);
        lw.x = pk(q0 - __uint_as_float(hw.x << 16), q1 - __uint_as_float(hw.x & 0xffff0000u));
        lw.y = pk(q2 - __uint_as_float(hw.y << 16), q3 - __uint_as_float(hw.y & 0xffff0000u));
        const int nl = cf * 32 + 8 * fg + 4 * lh, row = rf * 32 + l31;        // column inside the slice
        const int off = row * 128 + (((nl >> 3) ^ swz(row)) << 4) + ((nl & 7) >> 2) * 8;
        *reinterpret_cast<uint2*>(a_reg + off) = hw;
        *reinterpret_cast<uint2*>(a_reg + G::A_PL + off) = lw;
      }
    }
    // this wave's job
    const int job = KP == 2 ? (wave & 3) : wave, jh = job >> 1, jrf = job & 1, kh = KP == 2 ? wave >> 2 : 0;
    const int head = s * HS + jh, nT = p.xa_nT;
    const int b_item = m0 / p.T;
    const size_t bh = (size_t)b_item * 8 + head;
    const bf16x8* kfh = reinterpret_cast<const bf16x8*>(p.xa_kf_hi) + bh * nT * KSq * 64 + lane;
    const bf16x8* kfl = reinterpret_cast<const bf16x8*>(p.xa_kf_lo) + bh * nT * KSq * 64 + lane;
    const bf16x8* vfh = reinterpret_cast<const bf16x8*>(p.xa_vf_hi) + bh * nT * 2 * 64 + lane;
    const bf16x8* vfl = reinterpret_cast<const bf16x8*>(p.xa_vf_lo) + bh * nT * 2 * 64 + lane;
    const float* kbias = p.xa_bias + (size_t)b_item * nT * 32 + 4 * lh;
    struct KVT { bf16x8 kh[KSq], kl[KSq], vh[2], vl[2]; float4 bv[4]; };
    auto load_kv = [&](int t) {
      KVT f;
      const int tc = min(t, nT - 1);
#pragma unroll
      for (int ks = 0; ks < KSq; ++ks) { f.kh[ks] = kfh[(size_t)(tc * KSq + ks) * 64]; f.kl[ks] = kfl[(size_t)(tc * KSq + ks) * 64]; }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) { f.vh[kb] = vfh[(size_t)(tc * 2 + kb) * 64]; f.vl[kb] = vfl[(size_t)(tc * 2 + kb) * 64]; }
#pragma unroll
      for (int g = 0; g < 4; ++g) f.bv[g] = *reinterpret_cast<const float4*>(kbias + tc * 32 + 8 * g);
      return f;
    };
    KVT cur = load_kv(kh);                           // (the prompt's fragments depend on nothing this launch computes)
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                 // query planes complete
    bf16x8 qh[KSq], ql[KSq];
#pragma unroll
    for (int ks = 0; ks < KSq; ++ks) {
      const int c16 = ((jh * d + ks * 16) >> 3) + lh, row = jrf * 32 + l31;
      const int off = row * 128 + ((c16 ^ swz(row)) << 4);
      qh[ks] = *reinterpret_cast<const bf16x8*>(a_reg + off);
      ql[ks] = *reinterpret_cast<const bf16x8*>(a_reg + G::A_PL + off);
    }
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    DV_QTRACE(7);
    for (int t = kh; t < nT; t += KP) {
      __builtin_amdgcn_sched_barrier(0);
      KVT nxt = load_kv(t + KP);
      __builtin_amdgcn_sched_barrier(0);
      f32x16 sc;
      // (the key bias IS the initial accumulator)
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
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 hw, lw;
        if (DV_ATTN_PF16) {                          // P as one fp16 plane, V as split fp16: two products (dv_device.h)
#pragma unroll
          for (int e = 0; e < 4; ++e) hw[e] = dv_cvt_pk_f16(sc[kb * 8 + 2 * e], sc[kb * 8 + 2 * e + 1]);
          const dv_f16x8 ph16 = __builtin_bit_cast(dv_f16x8, hw);
          o = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dv_f16x8, cur.vl[kb]), ph16, o, 0, 0, 0);
          o = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dv_f16x8, cur.vh[kb]), ph16, o, 0, 0, 0);
          continue;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x0 = sc[kb * 8 + 2 * e], x1 = sc[kb * 8 + 2 * e + 1];
          const unsigned h2 = pk(x0, x1);
          hw[e] = h2;
          lw[e] = pk(x0 - __uint_as_float(h2 << 16), x1 - __uint_as_float(h2 & 0xffff0000u));
        }
        const bf16x8 ph = __builtin_bit_cast(bf16x8, hw), pl = __builtin_bit_cast(bf16x8, lw);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.vl[kb], ph, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.vh[kb], pl, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.vh[kb], ph, o, 0, 0, 0);
      }
      cur = nxt;
    }
    if constexpr (KP == 2) {
      // merge the two key halves of a job: wave w + 4 hands (m, l, O) over, wave w combines (log2 domain)
      float* mg = reinterpret_cast<float*>(red_reg);               // [4 jobs][2 + 16][64 lanes]
      __syncthreads();                               // (the exchange region's last readers - the q pieces - are done)
      if (kh == 1) {
        float* q = mg + (size_t)job * 18 * 64 + lane;
        q[0] = m_run; q[64] = l_run;
#pragma unroll
        for (int r = 0; r < 16; ++r) q[(2 + r) * 64] = o[r];
      }
      __syncthreads();
      if (kh == 0) {
        const float* q = mg + (size_t)job * 18 * 64 + lane;
        const float m1 = q[0], l1 = q[64];
        const float mm = fmaxf(m_run, m1);
        const float a0 = __builtin_amdgcn_exp2f(m_run - mm), a1 = __builtin_amdgcn_exp2f(m1 - mm);
        l_run = l_run * a0 + l1 * a1;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = o[r] * a0 + q[(2 + r) * 64] * a1;
      }
    }
    DV_QTRACE(8);
    // O (lane = query row, registers = channels 8g + 4lh + e of the head) -> split planes of the attention output in memory: the
    // other slices of the row block need them (the output projection contracts over ALL heads) - the second hand-over of the launch
    if (kh == 0) {
      const float inv = 1.0f / pair_sum32(l_run);
      const size_t ro = (size_t)(m0 + jrf * 32 + l31) * C + head * d + 4 * lh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (8 * g < d) {
          const float v0 = o[4 * g] * inv, v1 = o[4 * g + 1] * inv, v2 = o[4 * g + 2] * inv, v3 = o[4 * g + 3] * inv;
          uint2 hw, lw;
          hw.x = pk(v0, v1); hw.y = pk(v2, v3);
          lw.x = pk(v0 - __uint_as_float(hw.x << 16), v1 - __uint_as_float(hw.x & 0xffff0000u));
          lw.y = pk(v2 - __uint_as_float(hw.y << 16), v3 - __uint_as_float(hw.y & 0xffff0000u));
          *reinterpret_cast<uint2*>(p.qs_o_hi + ro + 8 * g) = hw;       // (plain stores: the XCD's L2 - MODE 2 runs XCD-local only)
          *reinterpret_cast<uint2*>(p.qs_o_lo + ro + 8 * g) = lw;
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long* fl2 = p.qs_flags + (size_t)(p.M / BM) * G::NSPL + (size_t)rb * G::NSPL;
    if (tid == 0) *reinterpret_cast<volatile unsigned long long*>(const_cast<unsigned long long*>(fl2) + s) = 1ull + xcc;
    // stage-3 vectors of this lane's finished columns
    const float4 b3v = *reinterpret_cast<const float4*>(p.b3 + ncol);
    wait_flags(fl2, 0xc3u);
    {
      // the row block's attention output (all heads) -> the row region, by L1-bypassing LDS-DMA (served by the XCD's L2)
      const unsigned a_base = (unsigned)(size_t)a_reg;
      const int d_row = wave * 8 + (lane >> 3), d_slot = lane & 7;
#pragma unroll
      for (int c = 0; c < G::A_CH; ++c) {
        const size_t e = (size_t)(m0 + d_row) * C + c * 64 + ((d_slot ^ swz(d_row)) << 3);
        const unsigned dst = a_base + (unsigned)(c * CHP + wave * 1024);
        glds16_sc1(p.qs_o_hi + e, dst);
        glds16_sc1(p.qs_o_lo + e, dst + G::A_PL);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    DV_QTRACE(9);
    // ================= stage 3: x3 = O W3^T + b3 + x1 -> fp32, raw planes and LayerNorm row partials (the GEGLU GEMM's inputs) =================
    job_loop(p.w3_hi, p.w3_lo, s * 2 + cf, std::false_type{});
    {
      float4 pv[2];
      pieces(acc[0], acc[1], pv);
      __shared__ float2 s_rp[2 * 32 * 2 * 4];        // [row][column fragment][register group] (sum, M2) of 8 columns
#pragma unroll
      for (int rf = 0; rf < 2; ++rf) {
        const float4 r4 = rf == 0 ? h2v[0] : h2v[1];
        float4 v = pv[rf];
        v.x += b3v.x + r4.x; v.y += b3v.y + r4.y; v.z += b3v.z + r4.z; v.w += b3v.w + r4.w;
        const size_t ob = (size_t)(m0 + rf * 32 + l31) * C + ncol;
        dv_st16(p.out3 + ob, v);
        uint2 hw, lw;
        hw.x = pk(v.x, v.y); hw.y = pk(v.z, v.w);
        lw.x = pk(v.x - __uint_as_float(hw.x << 16), v.y - __uint_as_float(hw.x & 0xffff0000u));
        lw.y = pk(v.z - __uint_as_float(hw.y << 16), v.w - __uint_as_float(hw.y & 0xffff0000u));
        dv_st8(p.out3_hi + ob, hw);
        dv_st8(p.out3_lo + ob, lw);
        // this wave's 8 columns of the row: (sum, M2 about their own mean); the 32-column block's partial is combined below
        const float a = pair_sum32((v.x + v.y) + (v.z + v.w));
        const float mb = a * (1.0f / 8.0f);
        const float q = pair_sum32((v.x - mb) * (v.x - mb) + (v.y - mb) * (v.y - mb) + (v.z - mb) * (v.z - mb) + (v.w - mb) * (v.w - mb));
        if (lh == 0) s_rp[((rf * 32 + l31) * 2 + cf) * 4 + fg] = make_float2(a, q);
      }
      __syncthreads();
      if (tid < 2 * BM) {                            // (row, column fragment): LayerNorm3's row partial of the 32-column block (parallel-variance form)
        const int row = tid >> 1, f = tid & 1;
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) s1 += s_rp[(row * 2 + f) * 4 + k].x;
        const float mean = s1 * (1.0f / 32.0f);
        float m2 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float2 t2 = s_rp[(row * 2 + f) * 4 + k];
          const float dm = t2.x * (1.0f / 8.0f) - mean;
          m2 += t2.y + 8.0f * dm * dm;
        }
        reinterpret_cast<float2*>(p.rowstat3)[(size_t)(m0 + row) * (C / 32) + s * 2 + f] = make_float2(s1, m2);
      }
    }
    DV_QTRACE(13);
    return;
  } else {
    // ONE k-loop for the three contractions: the row fragments of a k-step are read from LDS once and meet the q, the k and the v
    // fragment of this wave's columns (v with swapped operands: its accumulator is the transposed tile)
    f32x16 aq[2], ak[2], av[2];
#pragma unroll
    for (int rf = 0; rf < 2; ++rf)
#pragma unroll
      for (int r = 0; r < 16; ++r) { aq[rf][r] = 0.f; ak[rf][r] = 0.f; av[rf][r] = 0.f; }
    {
      auto read_a = [&](int u, bf16x8 (&h)[2], bf16x8 (&l)[2]) {
        const int c16 = (kq * G::U + u) * 2 + lh;
#pragma unroll
        for (int rf = 0; rf < 2; ++rf) {
          const int row = rf * 32 + l31;
          const int off = (c16 >> 3) * CHP + row * 128 + (((c16 & 7) ^ swz(row)) << 4);
          h[rf] = *reinterpret_cast<const bf16x8*>(a_reg + off);
          l[rf] = *reinterpret_cast<const bf16x8*>(a_reg + G::A_PL + off);
        }
      };
      bf16x8 ah[2][2], al[2][2];
      read_a(0, ah[0], al[0]);
#pragma unroll
      for (int j = 0; j < 3 * G::U; ++j) {
        const int u = j / 3, ps = j - 3 * u, cur = u & 1;
        if (ps == 0 && u + 1 < G::U) read_a(u + 1, ah[cur ^ 1], al[cur ^ 1]);
        const BFrag f = b3[j % G::D3];
        if (ps == 2) {
#pragma unroll
          for (int rf = 0; rf < 2; ++rf) av[rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cur][rf], f.h, av[rf], 0, 0, 0);
#pragma unroll
          for (int rf = 0; rf < 2; ++rf) av[rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][rf], f.l, av[rf], 0, 0, 0);
#pragma unroll
          for (int rf = 0; rf < 2; ++rf) av[rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][rf], f.h, av[rf], 0, 0, 0);
        } else if (ps == 1) {
#pragma unroll
          for (int rf = 0; rf < 2; ++rf) ak[rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, al[cur][rf], ak[rf], 0, 0, 0);
#pragma unroll
          for (int rf = 0; rf < 2; ++rf) ak[rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.l, ah[cur][rf], ak[rf], 0, 0, 0);
#pragma unroll
          for (int rf = 0; rf < 2; ++rf) ak[rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, ah[cur][rf], ak[rf], 0, 0, 0);
        } else {
#pragma unroll
          for (int rf = 0; rf < 2; ++rf) aq[rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, al[cur][rf], aq[rf], 0, 0, 0);
#pragma unroll
          for (int rf = 0; rf < 2; ++rf) aq[rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.l, ah[cur][rf], aq[rf], 0, 0, 0);
#pragma unroll
          for (int rf = 0; rf < 2; ++rf) aq[rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, ah[cur][rf], aq[rf], 0, 0, 0);
        }
        // pinned: without the scheduling barriers hipcc sinks every prefetch down to its use (load -> vmcnt(0) -> MFMA)
        __builtin_amdgcn_sched_barrier(0);
        if (j + G::D3 < 3 * G::U) b3[j % G::D3] = load_unit3(j + G::D3);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    DV_QTRACE(7);
    float4 pv[2];
    // ---- q ----
    pieces(aq[0], aq[1], pv);
    finish_q(pv);
    // ---- k: K fragments of the two 32-key tiles of this block (lane (half, key) holds 8 channels of a 16-channel group; this
    //      lane's four at byte 8 lh - k_chain2's layout) ----
    pieces(ak[0], ak[1], pv);
#pragma unroll
    for (int rf = 0; rf < 2; ++rf) {
      const float4 v = pv[rf];
      const float2 st = s_ln[rf * 32 + l31];
      const float o0 = st.y * (v.x - st.x * uk.x) + bk4.x, o1 = st.y * (v.y - st.x * uk.y) + bk4.y;
      const float o2 = st.y * (v.z - st.x * uk.z) + bk4.z, o3 = st.y * (v.w - st.x * uk.w) + bk4.w;
      uint2 hw, lw;
      hw.x = pk(o0, o1); hw.y = pk(o2, o3);
      lw.x = pk(o0 - __uint_as_float(hw.x << 16), o1 - __uint_as_float(hw.x & 0xffff0000u));
      lw.y = pk(o2 - __uint_as_float(hw.y << 16), o3 - __uint_as_float(hw.y & 0xffff0000u));
      const size_t eo = (((size_t)((m0 >> 5) + rf) * (C / 16) + (ncol >> 4)) * 64 + (fg & 1) * 32 + l31) * 8 + lh * 4;
      dv_st8(p.sa_kf_hi + eo, hw);
      dv_st8(p.sa_kf_lo + eo, lw);
    }
    DV_QTRACE(8);
    // ---- v: lane = channel, registers = keys 8 fg + 4 lh + e: that register image is half of the lane's 16-byte piece of the V^T
    //      fragment (channel block, k-block fg >> 1) of the tile ----
    pieces(av[0], av[1], pv);
#pragma unroll
    for (int rf = 0; rf < 2; ++rf) {
      const float4 v = pv[rf];
      // (mean, rstd) of the four key rows rf * 32 + 8 fg + 4 lh + e: 32 contiguous bytes of s_ln
      const float4 s01 = *reinterpret_cast<const float4*>(&s_ln[rf * 32 + 8 * fg + 4 * lh]);
      const float4 s23 = *reinterpret_cast<const float4*>(&s_ln[rf * 32 + 8 * fg + 4 * lh + 2]);
      const float x0 = s01.y * (v.x - s01.x * uvv) + bvv, x1 = s01.w * (v.y - s01.z * uvv) + bvv;
      const float x2 = s23.y * (v.z - s23.x * uvv) + bvv, x3 = s23.w * (v.w - s23.z * uvv) + bvv;
      uint2 hw, lw;
      if (DV_ATTN_PF16) {                            // V as split fp16 (dv_device.h)
        dv_split_pk_f16(x0, x1, hw.x, lw.x);
        dv_split_pk_f16(x2, x3, hw.y, lw.y);
      } else {
        hw.x = pk(x0, x1); hw.y = pk(x2, x3);
        lw.x = pk(x0 - __uint_as_float(hw.x << 16), x1 - __uint_as_float(hw.x & 0xffff0000u));
        lw.y = pk(x2 - __uint_as_float(hw.y << 16), x3 - __uint_as_float(hw.y & 0xffff0000u));
      }
      const size_t eo = (((((size_t)((m0 >> 5) + rf) * (C / 32) + s * 2 + cf) * 2 + (fg >> 1)) * 64 + lane) * 8) + (fg & 1) * 4;
      dv_st8(p.sa_vf_hi + eo, hw);
      dv_st8(p.sa_vf_lo + eo, lw);
    }
    DV_QTRACE(9);
  }
}

template <int C>
hipError_t qkv_init_one() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_split<C, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, QGeom<C>::SMEM);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_split<C, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, QGeom<C>::SMEM);
  if constexpr (C <= 256) {
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_qkv_split<C, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, QGeom<C>::SMEM);
  }
  return e;
}
template <int C>
hipError_t qkv_launch_one(const ChainParams& pin, hipStream_t st) {
  ChainParams p = pin;
  p.qs_xcd = qkv_split_xcd_local(p.M) ? 1 : 0;
  if (p.amode == 1) hipLaunchKernelGGL((k_qkv_split<C, 0>), dim3((p.M / BM) * QGeom<C>::NSPL), dim3(NT), QGeom<C>::SMEM, st, p);
  else if (!p.xa_kf_hi) hipLaunchKernelGGL((k_qkv_split<C, 1>), dim3((p.M / BM) * QGeom<C>::NSPL), dim3(NT), QGeom<C>::SMEM, st, p);
  else {
    if constexpr (C <= 256) {
      if (!p.qs_xcd) return hipErrorInvalidValue;    // (the cross-attention form hands over through the XCD's L2 only)
      hipLaunchKernelGGL((k_qkv_split<C, 2>), dim3((p.M / BM) * QGeom<C>::NSPL), dim3(NT), QGeom<C>::SMEM, st, p);
    } else return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace

// Do the slices of a row block share an XCD (hand-overs through its L2)?  Whole multiples of 8 row blocks; DVITS_QKV_XCD=0: never
bool qkv_split_xcd_local(int M) {
  static const bool xcd_on = [] { const char* e = getenv("DVITS_QKV_XCD"); return !(e && e[0] == '0'); }();
  return xcd_on && (M / BM) % 8 == 0;
}
hipError_t qkv_split_init() {
  hipError_t e = qkv_init_one<128>();
  if (e == hipSuccess) e = qkv_init_one<256>();
  return e != hipSuccess ? e : qkv_init_one<384>();
}
// chain 1 with its self-attention operands as fragments (amode 1, passes q | k | v, sa_*) or chain 2 without the cross attention
// inside (amode 0, one pass), whole 64-row blocks per utterance
bool qkv_split_supported(const ChainParams& p, int precision) {
  if (precision != 0 || !p.out1 || (p.ldo2 & 3) != 0) return false;
  if (p.C != 128 && p.C != 256 && p.C != 384) return false;
  if (p.T % BM != 0 || p.M % p.T != 0 || p.Tv < 0 || p.Tv > p.T || (p.Tv > 0 && p.Tv <= p.T - 32)) return false;
  if (p.amode == 1) {
    if (p.passes != 3 || !p.sa_kf_hi || !p.sa_kf_lo || !p.sa_vf_hi || !p.sa_vf_lo || p.res || p.xa_kf_hi || !p.out2 || p.ldo2 < p.C) return false;
    const int G = p.groups;
    if (G <= 0 || G > 64 || (G & (G - 1)) != 0 || p.C % G != 0 || (p.C / G) % 16 != 0) return false;
    if ((p.T / 32) * (p.C / 16) > 2 * 4 * 4 * 64 * 16 / 8) return false;   // the utterance's GroupNorm entries fit the exchange region
    return true;
  }
  if (p.amode != 0 || p.passes != 1 || !p.a_hi || !p.a_lo || p.sa_kf_hi) return false;
  if (!p.xa_kf_hi) return p.out2 && p.ldo2 >= p.C;                         // MODE 1: the query leaves as fp32
  // MODE 2: the cross attention inside (8 heads of d = C / 8 = 16 / 32, whole multiples of 8 row blocks: XCD-local hand-overs only)
  return p.C <= 256 && p.xa_kf_lo && p.xa_vf_hi && p.xa_vf_lo && p.xa_bias && p.xa_nT > 0 && p.xa_d == p.C / 8 && p.w3_hi && p.w3_lo && p.b3 &&
         p.out3 && p.out3_hi && p.out3_lo && p.rowstat3 && p.qs_o_hi && p.qs_o_lo && qkv_split_xcd_local(p.M);
}
int qkv_split_flags(const ChainParams& p) { return (p.M / BM) * (p.C / BN) * (p.xa_kf_hi ? 2 : 1); }   // (the cross-attention form hands over twice)
hipError_t launch_qkv_split(const ChainParams& p, int precision, hipStream_t st) {
  if (!qkv_split_supported(p, precision)) return hipErrorInvalidValue;
  if ((p.amode == 1 && (!p.x || !p.stat16 || !p.gamma || !p.beta)) || !p.w1_hi || !p.w1_lo || !p.b1 || !p.w2_hi || !p.w2_lo || !p.b2 || !p.u2 ||
      !p.qs_flags || !p.qs_status)
    return hipErrorInvalidValue;
  if (!gemm_handover_rounds()) {                     // (else: the wait is for C / 64 consecutive workgroup ids)
    static const int n_cu = [] { int d = 0, n = 0; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
    if ((p.M / BM) * (p.C / BN) > n_cu) return hipErrorInvalidValue;
  }
  if (p.C == 128) return qkv_launch_one<128>(p, st);
  if (p.C == 256) return qkv_launch_one<256>(p, st);
  return qkv_launch_one<384>(p, st);
}
