// Flash-attention tile routine shared by the per-launch kernel (kernels_attn.hip) and the persistent per-XCD schedule
// (persist.hip).  See kernels_attn.hip for the design notes.
#pragma once
#include "dv_common.h"
#include "dv_device.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef DV_ATTN_KVSPLIT_MAX
#define DV_ATTN_KVSPLIT_MAX 16
#endif

__device__ __forceinline__ unsigned apk(float lo, float hi) { return dv_cvt_pk_bf16(lo, hi); }
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

template <int DP, int NW>
struct AttnGeom {
  static constexpr int KP = DP * 2 + 16, VP = 80, NB = (DP + 31) / 32;
  static constexpr int K_PL = 32 * KP, V_PL = NB * 32 * VP;
  static constexpr int NST = 4;                                  // raw tiles in flight
  static constexpr int RAWK = 32 * DP * 4;                       // raw fp32 K (or V) tile bytes
  static constexpr int KV_INSTR = 2 * 32 * (DP / 4) / 64;         // DMA wave-instructions per tile for K and V (= DP / 4)
  static constexpr int KPW = (KV_INSTR + NW - 1) / NW;           // per wave (at most)
  static constexpr int RAW = (2 * RAWK + 256 + 1023) / 1024 * 1024;   // + the tile's key bias (32 floats, by wave 0)
  static constexpr int buf_bytes(int npl) { return (K_PL + V_PL) * npl + 128; }
  static constexpr int planes_bytes(int npl) { return (4 * buf_bytes(npl) + 1023) / 1024 * 1024; }   // 2 sub-tiles x double buffer
  static constexpr int smem_bytes(int npl) { return planes_bytes(npl) + NST * RAW; }
};

#if defined(DV_GEMM_TRACE) && defined(DV_ATTN_TRACE_OWNER)
// development build only (make trace): per-workgroup s_memtime stamps of the attention kernel's phases
__device__ unsigned long long g_attn_trace[4096 * 16];
#define DV_ATRACE(i) do { const unsigned w_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
    if (threadIdx.x == 0 && w_ < 4096) g_attn_trace[w_ * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int dv_debug_attn_trace(unsigned long long* host, int n_wg) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_trace), (size_t)n_wg * 16 * sizeof(unsigned long long));
}
extern "C" int dv_debug_attn_trace_clear() {
  void* d = nullptr;
  hipError_t e = hipGetSymbolAddress(&d, HIP_SYMBOL(g_attn_trace));
  return (int)(e != hipSuccess ? e : hipMemset(d, 0, sizeof(g_attn_trace)));
}
#else
#define DV_ATRACE(i) do {} while (0)
#endif

// One (query block of 32*NW queries, head h, batch item b).  SC1: loads of q / k / v / bias bypass this CU's L1 (they
// were written earlier in the SAME launch by other CUs: persist.hip).
template <int DP, int NW, int NSPLIT, bool SC1>
__device__ __forceinline__ void attn_tile(const AttnParams& p, const int qblk, const int h, const int b, char* lds) {
  constexpr bool SPLIT = NSPLIT == 3;
  constexpr int NPL = SPLIT ? 2 : 1;
  constexpr int KS = DP / 16;                      // k-steps of K Q^T
  constexpr int NB = (DP + 31) / 32;               // 32-wide output-channel blocks
  constexpr int KP = DP * 2 + 16;                  // K row pitch (bytes): odd number of 16-byte slots
  constexpr int VP = 80;                           // V^T row pitch: 32 keys * 2 B + 16
  constexpr int K_PL = 32 * KP, V_PL = NB * 32 * VP;
  constexpr int BUF = (K_PL + V_PL) * NPL + 128;   // + 32 floats of key bias
  constexpr int NT = 64 * NW;
  constexpr int TASKS = 4 * DP;                    // (key pair, 4-channel group) conversion tasks per tile
  constexpr bool KVSPLIT = DP <= DV_ATTN_KVSPLIT_MAX;   // conversion task granularity (see convert_pair)
  using G = AttnGeom<DP, NW>;
  constexpr int NST = G::NST, RAWK = G::RAWK, RAW = G::RAW;
  constexpr int CPR = DP / 4;                      // 16-byte chunks per raw row
  constexpr int KV_INSTR = G::KV_INSTR, KPW = G::KPW;
  char* const raw0 = lds + G::planes_bytes(NPL);

  DV_ATRACE(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int d = p.d;
  const int qi = qblk * (32 * NW) + wave * 32 + l31;
  const bool q_ok = qi < p.Tq;

  // scores are kept in the log2 domain (q pre-scaled by d^-1/2 * log2 e, key bias by log2 e): the softmax
  // exponentials are bare v_exp_f32
  constexpr float LOG2E = 1.44269504088896340736f;
  const float qscale = p.scale * LOG2E;
  // ---- Q fragments (B operand of K Q^T): lane (query, lh) holds channels ks*16 + lh*8 .. +8 ----
  // (loaded AFTER the K / V ring has been filled: the two cold-miss round trips overlap)
  bf16x8 qh[KS], ql[KS];
  auto load_q = [&]() {
    const float* qp = p.q + ((size_t)b * p.Tq + (q_ok ? qi : 0)) * p.ldq + h * d;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks * 16 + lh * 8;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c2 = a;
      if (q_ok && c < d) a = ld_mut4<SC1>(qp + c);
      if (q_ok && c + 4 < d) c2 = ld_mut4<SC1>(qp + c + 4);
      a.x *= qscale; a.y *= qscale; a.z *= qscale; a.w *= qscale;
      c2.x *= qscale; c2.y *= qscale; c2.z *= qscale; c2.w *= qscale;
      u32x4 hw, lw;
      hw.x = apk(a.x, a.y); hw.y = apk(a.z, a.w); hw.z = apk(c2.x, c2.y); hw.w = apk(c2.z, c2.w);
      lw.x = apk(a.x - bf_lo(hw.x), a.y - bf_hi(hw.x)); lw.y = apk(a.z - bf_lo(hw.y), a.w - bf_hi(hw.y));
      lw.z = apk(c2.x - bf_lo(hw.z), c2.y - bf_hi(hw.z)); lw.w = apk(c2.z - bf_lo(hw.w), c2.w - bf_hi(hw.w));
      qh[ks] = __builtin_bit_cast(bf16x8, hw);
      ql[ks] = __builtin_bit_cast(bf16x8, lw);
    }
  };

  f32x16 o[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // ---- K / V tile staging ----
  // DMA: instruction ii of a tile moves raw chunks [64 ii, 64 ii + 64) of the tile image [K: 32 x DP | V: 32 x DP]
  // (row-major fp32, 16-byte chunks; chunks past d read a valid dummy and are zeroed at conversion)
  const unsigned raw_base = (unsigned)(size_t)raw0;
  // per-lane source geometry of this wave's DMA instructions, computed once (the key loop is vector-ALU bound: per
  // tile only the key clamp and one 64-bit multiply-add per instruction remain).  Kept in registers when a wave has
  // few instructions per tile (the wide workgroups of the large shapes), recomputed per tile otherwise.
  constexpr bool GEO_REGS = KPW <= 4;
  const float* dma_src[GEO_REGS ? KPW : 1];
  int dma_row[GEO_REGS ? KPW : 1], dma_ld[GEO_REGS ? KPW : 1];
  const int tk_pitch = p.Tk_pitch > 0 ? p.Tk_pitch : p.Tk;   // key rows per utterance in k / v (padded row spaces: > Tk)
  auto dma_geo = [&](int j, const float*& base, int& row, int& ld) {
    const int task = (wave + j * NW) * 64 + lane;
    const int which = task / (32 * CPR), rem = task - which * (32 * CPR);
    row = rem / CPR;
    const int c4 = (rem - row * CPR) * 4;
    const int cc = c4 < d ? c4 : 0;
    ld = which ? p.ldv : p.ldk;
    base = (which ? p.v + (size_t)b * tk_pitch * p.ldv : p.k + (size_t)b * tk_pitch * p.ldk) + h * d + cc;
  };
  if (GEO_REGS) {
#pragma unroll
    for (int j = 0; j < KPW; ++j) dma_geo(j, dma_src[j], dma_row[j], dma_ld[j]);
  }
  const float* const bias_b = p.bias ? p.bias + (size_t)b * p.Tk : nullptr;
  auto issue_tile = [&](int t) {
    const int kt0 = t * 32;
    const unsigned st = raw_base + (unsigned)((t % NST) * RAW);
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
      const int ii = wave + j * NW;
      if (ii >= KV_INSTR) continue;      // every wait below is vmcnt(0): the waves need not issue equal counts
      const float* base; int row, ld;
      if (GEO_REGS) { base = dma_src[j]; row = dma_row[j]; ld = dma_ld[j]; }
      else dma_geo(j, base, row, ld);
      const float* src = base + (size_t)min(kt0 + row, p.Tk - 1) * ld;   // key clamped: masked by the bias
      if (SC1) glds16_sc1(src, st + (unsigned)(ii * 1024));
      else glds16(src, st + (unsigned)(ii * 1024));
    }
    if (wave == 0) {   // key bias of the tile
      const int key = min(kt0 + l31, p.Tk - 1);
      const void* src = bias_b ? (const void*)(bias_b + key) : (const void*)p.k;
      if (SC1) glds4_sc1(src, st + (unsigned)(2 * RAWK));
      else glds4(src, st + (unsigned)(2 * RAWK));
    }
  };
  // conversion of the raw sub-tile pair (t0, t0 + 1) -> split bf16 planes buffers (buf0, buf0 + 1).  The key loop is
  // bound by vector-ALU work (softmax + operand splitting, two waves per SIMD), and the waves meet at one barrier per
  // iteration: the conversion is therefore cut into small tasks = (sub-tile, K or V, key pair kp, 4-channel group c4)
  // of 8 elements, spread over as many waves as there are (4 * TASKS = 16 DP tasks per pair: 256 .. 1024), instead of
  // 16-element tasks that kept only DP / 16 waves busy while the others waited.  TASKS is a multiple of 64, so the
  // K / V choice is wave-uniform.  Measured: the 8-element tasks pay at d = 16 only (more, narrower LDS writes cost
  // more than the better balance gains at d >= 48); wider heads keep K and V of a (kp, c4) in one 16-element task,
  // still spread over both sub-tiles.
  auto convert_pair = [&](int t0, int buf0) {
    constexpr int NPART = KVSPLIT ? 2 : 1;              // K and V of a (kp, c4) as two tasks, or as one
    constexpr int TASKS2 = 2 * NPART * TASKS;
    // only the first wave of each SIMD converts (waves w and w + 4 share a SIMD): its partner starts the MFMAs at
    // once, which de-phases the pair - one in the vector ALU while the other is in the matrix pipe
    constexpr int CT = 64 * (NW < 4 ? NW : 4);
    constexpr int TPT2 = (TASKS2 + CT - 1) / CT;
#pragma unroll
    for (int i = 0; i < TPT2; ++i) {
      const int task = tid + i * CT;
      if (tid >= CT || task >= TASKS2) continue;
      const int tl = task / (NPART * TASKS), r1 = task - tl * (NPART * TASKS);
      const int which = KVSPLIT ? __builtin_amdgcn_readfirstlane(r1 / TASKS) : 0, r2 = r1 - which * TASKS;
      const int kp = r2 / (DP / 4), c4 = (r2 - kp * (DP / 4)) * 4;
      const char* rs = raw0 + ((t0 + tl) % NST) * RAW + which * RAWK;
      char* base = lds + (buf0 + tl) * BUF;
      float4 a = *reinterpret_cast<const float4*>(rs + ((2 * kp) * DP + c4) * 4);
      float4 c = *reinterpret_cast<const float4*>(rs + ((2 * kp + 1) * DP + c4) * 4);
      if (c4 >= d) { a = make_float4(0.f, 0.f, 0.f, 0.f); c = a; }   // zero the padded channels
      if (which == 0) {
        // K rows 2kp, 2kp+1: 4 channels -> 8 bytes per plane
        char* k_hi = base;
        char* k_lo = base + K_PL;
        uint2 h0, h1, l0, l1;
        h0.x = apk(a.x, a.y); h0.y = apk(a.z, a.w);
        h1.x = apk(c.x, c.y); h1.y = apk(c.z, c.w);
        *reinterpret_cast<uint2*>(k_hi + (2 * kp) * KP + c4 * 2) = h0;
        *reinterpret_cast<uint2*>(k_hi + (2 * kp + 1) * KP + c4 * 2) = h1;
        if (SPLIT) {
          l0.x = apk(a.x - bf_lo(h0.x), a.y - bf_hi(h0.x)); l0.y = apk(a.z - bf_lo(h0.y), a.w - bf_hi(h0.y));
          l1.x = apk(c.x - bf_lo(h1.x), c.y - bf_hi(h1.x)); l1.y = apk(c.z - bf_lo(h1.y), c.w - bf_hi(h1.y));
          *reinterpret_cast<uint2*>(k_lo + (2 * kp) * KP + c4 * 2) = l0;
          *reinterpret_cast<uint2*>(k_lo + (2 * kp + 1) * KP + c4 * 2) = l1;
        }
      }
      if (!KVSPLIT) {
        a = *reinterpret_cast<const float4*>(rs + RAWK + ((2 * kp) * DP + c4) * 4);
        c = *reinterpret_cast<const float4*>(rs + RAWK + ((2 * kp + 1) * DP + c4) * 4);
        if (c4 >= d) { a = make_float4(0.f, 0.f, 0.f, 0.f); c = a; }
      }
      if (!KVSPLIT || which == 1) {
        // V^T: channel rows, key slot = key index with bits 2 and 3 swapped; keys 2kp, 2kp+1 are adjacent slots
        char* v_hi = base + NPL * K_PL;
        char* v_lo = v_hi + V_PL;
        const int j = 2 * kp;
        const int slot = (j & 0x13) | ((j & 4) << 1) | ((j & 8) >> 1);
        const float ve0[4] = {a.x, a.y, a.z, a.w}, ve1[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (SPLIT && DV_ATTN_PF16) {               // V as split fp16 (P enters P V as one fp16 plane: dv_device.h)
            unsigned hw, lw;
            dv_split_pk_f16(ve0[e], ve1[e], hw, lw);
            *reinterpret_cast<unsigned*>(v_hi + (c4 + e) * VP + slot * 2) = hw;
            *reinterpret_cast<unsigned*>(v_lo + (c4 + e) * VP + slot * 2) = lw;
            continue;
          }
          const unsigned hw = apk(ve0[e], ve1[e]);
          *reinterpret_cast<unsigned*>(v_hi + (c4 + e) * VP + slot * 2) = hw;
          if (SPLIT) *reinterpret_cast<unsigned*>(v_lo + (c4 + e) * VP + slot * 2) = apk(ve0[e] - bf_lo(hw), ve1[e] - bf_hi(hw));
        }
      }
    }
    if (wave == NW - 1) {        // key bias of both sub-tiles (the last wave has the fewest conversion tasks)
      const int tl = lane >> 5, t = t0 + tl;
      const int key = t * 32 + l31;
      const float raw = reinterpret_cast<const float*>(raw0 + (t % NST) * RAW + 2 * RAWK)[l31];
      float* bl = reinterpret_cast<float*>(lds + (buf0 + tl) * BUF + (K_PL + V_PL) * NPL);
      bl[l31] = (key < p.Tk) ? (p.bias ? raw * LOG2E : 0.f) : -1e30f;
    }
  };

  // Two 32-key sub-tiles per iteration: one barrier, two independent S^T MFMA chains, one softmax pass over 32 scores per
  // lane.  Sub-tile t lives in raw stage t % NST and, converted, in planes buffer 2 * (iteration & 1) + (t & 1).  A
  // trailing odd sub-tile is processed as fully masked (its keys clamp to Tk - 1, its bias is -1e30).
  const int nsub = (p.Tk + 31) / 32, nit = (nsub + 1) / 2, nsub2 = 2 * nit;
  DV_ATRACE(1);
#pragma unroll
  for (int t = 0; t < NST; ++t)
    if (t < nsub2) issue_tile(t);
  DV_ATRACE(2);                  // ring filled
  load_q();                      // behind the DMAs in the memory queue: when it has returned, so have they
  wait_vmcnt<0>();
  __syncthreads();
  DV_ATRACE(3);                  // first two sub-tiles landed
  convert_pair(0, 0);
  DV_ATRACE(4);                  // and converted
  for (int it = 0; it < nit; ++it) {
#ifdef DV_GEMM_TRACE
    if (it == 1) DV_ATRACE(5);   // first iteration done
#endif
    const bool more = it + 1 < nit;
    if (more) wait_vmcnt<0>();   // sub-tiles 2it+2, 2it+3 (issued one iteration ago) have landed for this wave
    __syncthreads();            // planes of this iteration written, next raw sub-tiles landed, previous MFMAs done
    if (2 * it + NST < nsub2) { issue_tile(2 * it + NST); issue_tile(2 * it + NST + 1); }   // stages of sub-tiles 2it, 2it+1
    if (more) convert_pair(2 * it + 2, 2 * ((it + 1) & 1));
    const char* base0 = lds + (2 * (it & 1)) * BUF;

    // ---- S^T = K Q^T for both sub-tiles (independent accumulators: the MFMA chains interleave) ----
    f32x16 s[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int off = l31 * KP + (ks * 2 + lh) * 16;
      bf16x8 kh[2], kl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        kh[u] = *reinterpret_cast<const bf16x8*>(base0 + u * BUF + off);
        if (SPLIT) kl[u] = *reinterpret_cast<const bf16x8*>(base0 + u * BUF + K_PL + off);
      }
      if (SPLIT) {
#pragma unroll
        for (int u = 0; u < 2; ++u) s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl[u], qh[ks], s[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 2; ++u) s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[u], ql[ks], s[u], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[u], qh[ks], s[u], 0, 0, 0);
    }
    // ---- online softmax (lane = query; registers = keys) ----
    // the key bias (attention mask / keys past Tk) is only added where there is one: wave-uniform branch
    if (p.bias != nullptr || (!more && (p.Tk & 63) != 0)) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float* bl = reinterpret_cast<const float*>(base0 + u * BUF + (K_PL + V_PL) * NPL);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = *reinterpret_cast<const float4*>(bl + 8 * g + 4 * lh);
          s[u][4 * g] += bv.x; s[u][4 * g + 1] += bv.y; s[u][4 * g + 2] += bv.z; s[u][4 * g + 3] += bv.w;
        }
      }
    }
    // vector-ALU economy (this loop is VALU-bound): three-input max, packed fp32 subtract / add
    float tmax = m_run;
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = __builtin_fmaxf(__builtin_fmaxf(tmax, s[0][r]), s[1][r]);   // v_max3_f32
    const float m_new = pair_max32(tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    const f32x2 mneg = {-m_new, -m_new};
    f32x2 psum = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 v = {s[u][r], s[u][r + 1]};
        v += mneg;                                            // v_pk_add_f32
        v.x = __builtin_amdgcn_exp2f(v.x);
        v.y = __builtin_amdgcn_exp2f(v.y);
        psum += v;
        s[u][r] = v.x; s[u][r + 1] = v.y;
      }
    l_run = l_run * alpha + (psum.x + psum.y);
    if (__any(alpha != 1.0f)) {        // the running maximum moved for some query of this wave
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nb][r] *= alpha;
    }
    // ---- O^T += V^T P^T: P^T operand of k-block kb = this lane's score registers 8kb..8kb+7 ----
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const char* v_hi = base0 + u * BUF + NPL * K_PL;
      const char* v_lo = v_hi + V_PL;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        u32x4 hw, lw;
        if (SPLIT && DV_ATTN_PF16) {
          hw.x = dv_cvt_pk_f16(s[u][kb * 8 + 0], s[u][kb * 8 + 1]); hw.y = dv_cvt_pk_f16(s[u][kb * 8 + 2], s[u][kb * 8 + 3]);
          hw.z = dv_cvt_pk_f16(s[u][kb * 8 + 4], s[u][kb * 8 + 5]); hw.w = dv_cvt_pk_f16(s[u][kb * 8 + 6], s[u][kb * 8 + 7]);
          const dv_f16x8 ph16 = __builtin_bit_cast(dv_f16x8, hw);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const int off = (nb * 32 + l31) * VP + (kb * 2 + lh) * 16;
            const dv_f16x8 vh = *reinterpret_cast<const dv_f16x8*>(v_hi + off);
            const dv_f16x8 vl = *reinterpret_cast<const dv_f16x8*>(v_lo + off);
            o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph16, o[nb], 0, 0, 0);
            o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph16, o[nb], 0, 0, 0);
          }
          continue;
        }
        hw.x = apk(s[u][kb * 8 + 0], s[u][kb * 8 + 1]); hw.y = apk(s[u][kb * 8 + 2], s[u][kb * 8 + 3]);
        hw.z = apk(s[u][kb * 8 + 4], s[u][kb * 8 + 5]); hw.w = apk(s[u][kb * 8 + 6], s[u][kb * 8 + 7]);
        const bf16x8 ph = __builtin_bit_cast(bf16x8, hw);
        bf16x8 pl;
        if (SPLIT) {
          auto lo_pair = [&](unsigned h2, float x0, float x1) {   // residual of two scores: one packed subtract
            const f32x2 x = {x0, x1}, hf = {bf_lo(h2), bf_hi(h2)};
            const f32x2 dlt = x - hf;
            return apk(dlt.x, dlt.y);
          };
          lw.x = lo_pair(hw.x, s[u][kb * 8 + 0], s[u][kb * 8 + 1]);
          lw.y = lo_pair(hw.y, s[u][kb * 8 + 2], s[u][kb * 8 + 3]);
          lw.z = lo_pair(hw.z, s[u][kb * 8 + 4], s[u][kb * 8 + 5]);
          lw.w = lo_pair(hw.w, s[u][kb * 8 + 6], s[u][kb * 8 + 7]);
          pl = __builtin_bit_cast(bf16x8, lw);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int off = (nb * 32 + l31) * VP + (kb * 2 + lh) * 16;
          const bf16x8 vh = *reinterpret_cast<const bf16x8*>(v_hi + off);
          if (SPLIT) {
            const bf16x8 vl = *reinterpret_cast<const bf16x8*>(v_lo + off);
            o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, o[nb], 0, 0, 0);
            o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, o[nb], 0, 0, 0);
          }
          o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o[nb], 0, 0, 0);
        }
      }
    }
  }

  DV_ATRACE(6);                  // key loop done
  const float l_tot = pair_sum32(l_run);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
    const size_t obase = ((size_t)b * p.Tq + qi) * p.ldo + h * d;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float v16[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v16[r] = o[nb][r] * inv;
      if (p.o) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dv = nb * 32 + 8 * g + 4 * lh;
          if (dv < d) *reinterpret_cast<float4*>(p.o + obase + dv) = make_float4(v16[4 * g], v16[4 * g + 1], v16[4 * g + 2], v16[4 * g + 3]);
        }
      }
      // split bf16 planes for the to_out GEMM (hi = rne(v), lo = rne(v - hi)) as 16-byte stores: the lanes of a pair exchange halves
      // (dv_device.h store_planes16; d is a multiple of 16: the head's last fragment may hold 16 columns only)
      if (p.o_hi && (d & 15) == 0) store_planes16(p.o_hi, p.o_lo, obase + nb * 32, lh, v16, d - nb * 32 >= 32 ? 3 : 1);
      else if (p.o_hi) {                              // heads of 8 / 24 ... channels: 8-byte stores, guarded per 4 columns
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dv = nb * 32 + 8 * g + 4 * lh;
          if (dv < d) {
            uint2 hh, ll;
            hh.x = apk(v16[4 * g], v16[4 * g + 1]); hh.y = apk(v16[4 * g + 2], v16[4 * g + 3]);
            *reinterpret_cast<uint2*>(p.o_hi + obase + dv) = hh;
            if (p.o_lo) {
              ll.x = apk(v16[4 * g] - bf_lo(hh.x), v16[4 * g + 1] - bf_hi(hh.x));
              ll.y = apk(v16[4 * g + 2] - bf_lo(hh.y), v16[4 * g + 3] - bf_hi(hh.y));
              *reinterpret_cast<uint2*>(p.o_lo + obase + dv) = ll;
            }
          }
        }
      }
    }
  }
  DV_ATRACE(7);
}

