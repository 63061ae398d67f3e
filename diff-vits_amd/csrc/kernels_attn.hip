// Flash-style multi-head attention for gfx950 (MI355X): softmax(q k^T * d^-1/2 + bias) v
// without materialising the score matrix (reference unet1d/attention_processor.py:1019-1035
// = F.scaled_dot_product_attention on [B,H,T,d] views; additive key bias [B,Tk] from
// prepare_attention_mask :309-336).
//
// Both contractions run on v_mfma_f32_32x32x16_bf16 with split-bf16 operands (hi*hi + lo*hi +
// hi*lo, fp32 accumulate: the same 1e-5-class accuracy as the GEMMs; bf16 mode issues hi*hi
// only).  One wave owns 32 queries and both products are computed TRANSPOSED so that every
// softmax quantity is lane-local:
//   S^T[key][query] = K Q^T   -> lane (query = lane&31) holds 16 of the tile's 32 keys (the other
//                               16 sit in lane^32): row max / row sum need one 64-lane shuffle
//   O^T[dv][query]  = V^T P^T -> the P^T operand of k-block kb is the lane's OWN score registers
//                               8kb..8kb+7 (converted to bf16 in place, no cross-lane traffic);
//                               V^T is stored in LDS with its key axis permuted (bits 2 and 3 of
//                               the key index swapped) so that the matching 8 keys are one 16-byte
//                               read; the per-query rescale and the final 1/l are per-lane scalars
// K / V tiles of 32 keys travel global -> LDS by LDS-DMA (global_load_lds, raw fp32, 4-stage ring; a wave only
// issues the instructions that carry data, every wait is vmcnt(0)): the producer kernel's output is never in this
// kernel's L2 (kernel boundary), so every tile is a ~2 us fabric round trip and a short key loop is bound by how many
// tiles are in flight, not by bandwidth.  Two 32-key sub-tiles are processed per iteration (one barrier).  Each pair is
// converted fp32 -> split bf16 (K rows; V^T with the permuted key axis) once per workgroup, one iteration ahead of
// the MFMAs, by ONE wave of each SIMD pair, and shared by the NW waves (32*NW queries).  The key loop is bound by
// vector-ALU work (softmax, P -> hi + lo, the conversion), not by MFMA: see docs/HISTORY.md and tools/attn_trace.py.
#include "dv_common.h"
#define DV_ATTN_TRACE_OWNER   // the trace build's stamp buffer lives in this translation unit
#include "attn_tile.h"

#include <cstdlib>

// 1-D grid -> (query block, head, batch item).  Workgroup id runs on XCD id % 8 and every XCD has its own (cold) L2: all
// query blocks of one (batch item, head) are given to ONE XCD, so its K / V cross the fabric once instead of once per
// query block (up to 4x at the bench shape).
__device__ __forceinline__ void attn_block_of(int id, int nq, int H, int B, int& qblk, int& h, int& b, bool by_xcd = true) {
  const int pairs = H * B;
  int pair;
  if (by_xcd && (pairs & 7) == 0) {
    const int x = id & 7, i = id >> 3;
    pair = (i / nq) * 8 + x;
    qblk = i - (i / nq) * nq;
  } else {
    pair = id / nq;
    qblk = id - pair * nq;
  }
  b = pair / H;
  h = pair - b * H;
}

template <int DP, int NW, int NSPLIT>
__global__ __launch_bounds__(64 * NW) void k_attention(const AttnParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];   // [2 x BUF planes | NST x RAW ring]
  int qblk, h, b;
  attn_block_of(blockIdx.x, (p.Tq + 32 * NW - 1) / (32 * NW), p.H, p.B, qblk, h, b, p.no_xcd_map == 0);
  attn_tile<DP, NW, NSPLIT, false>(p, qblk, h, b, lds);
}

template <int DP, int NW>
static void launch_att(const AttnParams& p, dim3 grid, hipStream_t st) {
  using G = AttnGeom<DP, NW>;
  const int smem3 = G::smem_bytes(2), smem1 = G::smem_bytes(1);
  if (p.nsplit == 3) hipLaunchKernelGGL((k_attention<DP, NW, 3>), grid, dim3(64 * NW), smem3, st, p);
  else hipLaunchKernelGGL((k_attention<DP, NW, 1>), grid, dim3(64 * NW), smem1, st, p);
}
template <int DP, int NW>
static hipError_t init_att() {
  using G = AttnGeom<DP, NW>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attention<DP, NW, 3>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, G::smem_bytes(2));
  return e != hipSuccess ? e : hipFuncSetAttribute(reinterpret_cast<const void*>(k_attention<DP, NW, 1>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, G::smem_bytes(1));
}
static hipError_t attn_frag_init();
// > 64 KiB of dynamic LDS needs the attribute; set once, outside any stream capture
hipError_t attn_init() {
  static bool done = false;
  if (done) return hipSuccess;
  hipError_t e = hipSuccess;
#define ATT_INIT(DP)                                               \
  if (e == hipSuccess) e = init_att<DP, 8>();                      \
  if (e == hipSuccess) e = init_att<DP, 4>();                      \
  if (e == hipSuccess) e = init_att<DP, 2>();                      \
  if (e == hipSuccess) e = init_att<DP, 1>();
  ATT_INIT(16) ATT_INIT(32) ATT_INIT(48) ATT_INIT(64)
#undef ATT_INIT
  if (e == hipSuccess) e = attn_frag_init();
  done = e == hipSuccess;
  return e;
}

// ---------------------------------------------------------------------------------------
// k_attention_frag: the same attention over K / V that arrive as split-bf16 MFMA fragments (AttnFragParams).  The
// fragment blocks of a 32-key tile are lane-linear, so they travel global -> LDS by LDS-DMA exactly as they are multiplied
// (one ds_read_b128 per lane, conflict-free, no swizzle, no padding) and the whole fp32 -> planes conversion stage of
// k_attention - its vector-ALU work, its LDS round trip, its pipeline stage - is gone.  Ring of three 64-key pairs,
// counted waits (a wave knows how many DMA instructions of a pair are its own), one barrier per pair.
// KSP = 2 (the short levels: T <= 256 frames leave half of the CUs without a workgroup, and a wave's key loop - bound by
// the softmax's vector-ALU issue - is the whole critical path): two waves share a query block, wave half kh takes the
// sub-tile kh of every 64-key pair (same ring, same DMAs, half the work per wave per pair), and the two partial
// (m, l, O) are merged through LDS once, behind the loop.  A workgroup then covers 32 NW / 2 queries: twice the workgroups.
// ---------------------------------------------------------------------------------------
template <int DP, int NW, int NSPLIT, int KSP = 1>
__global__ __launch_bounds__(64 * NW) void k_attention_frag(const AttnFragParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  constexpr bool SPLIT = NSPLIT == 3;
  constexpr int NPL = SPLIT ? 2 : 1;
  constexpr int KS = DP / 16, NB = (DP + 31) / 32;
  constexpr int KBL = KS * NPL, VBL = 2 * NB * NPL;      // 1 KiB blocks per 32-key sub-tile
  constexpr int SUB = (KBL + VBL) * 1024;
  constexpr int PAIR = 2 * SUB + 256;                   // + the pair's key bias (64 floats)
  constexpr int NSTG = 3;
  constexpr int NI = 2 * (KBL + VBL);                   // DMA instructions per pair
  constexpr int CPW = (NI + NW - 1) / NW;               // per wave (at most)
  constexpr int NQ = NW / KSP;                          // query blocks (of 32) per workgroup
  constexpr int NU = 2 / KSP;                           // sub-tiles of a pair a wave multiplies
  static_assert(KSP == 1 || (KSP == 2 && NW % 2 == 0), "key split: two wave halves");
  asm volatile("" ::"s"(p.q), "s"(p.vf_lo), "s"(p.v_t), "s"(p.bias), "s"(p.o_lo), "s"(p.no_xcd_map));   // all argument lines at once
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  int qblk, h, b;
  attn_block_of(blockIdx.x, (p.Tq + 32 * NQ - 1) / (32 * NQ), p.H, p.B, qblk, h, b, p.no_xcd_map == 0);
  const int d = p.d;
  const int qg = KSP == 2 ? wave % NQ : wave, kh = KSP == 2 ? wave / NQ : 0;   // query block of this wave, its key half
  const int qi = qblk * (32 * NQ) + qg * 32 + l31;
  const bool q_ok = qi < p.Tq;
  constexpr float LOG2E = 1.44269504088896340736f;
  const float qscale = p.scale * LOG2E;
  const int row0 = p.self_layout ? (h * d) & 31 : 0;    // first channel row of this head inside its first V^T fragment
  const int nsub = (p.Tk + 31) / 32, nit = (nsub + 1) / 2;

  // ---- this wave's DMA instructions of a pair: ii = wave + j * NW -> (sub-tile u, block blk) ----
  const unsigned lds_base = (unsigned)(size_t)lds;
  const bf16_t* src[CPW]; int step[CPW], tsel[CPW]; unsigned dst[CPW];
  int my_n = 0;
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    const int ii = wave + j * NW;
    src[j] = nullptr; step[j] = 0; tsel[j] = 0; dst[j] = 0;
    if (ii >= NI) continue;
    ++my_n;
    const int u = ii / (KBL + VBL), blk = ii - u * (KBL + VBL);
    tsel[j] = u;
    dst[j] = (unsigned)(u * SUB + blk * 1024);
    if (blk < KBL) {
      const int pl = blk / KS, ks = blk - pl * KS;
      src[j] = (pl ? p.kf_lo : p.kf_hi) + ((size_t)b * p.k_b + (size_t)h * p.k_h + ks) * 512 + lane * 8;
      step[j] = p.k_t;
    } else {
      const int vb = blk - KBL, pl = vb / (2 * NB), r = vb - pl * (2 * NB), kb = r / NB, nb = r - kb * NB;
      const size_t hv = p.self_layout ? (size_t)((h * d) >> 5) * p.v_nb : (size_t)h * p.v_h;
      src[j] = (pl ? p.vf_lo : p.vf_hi) + ((size_t)b * p.v_b + hv + (size_t)kb * p.v_kb + (size_t)nb * p.v_nb) * 512 + lane * 8;
      step[j] = p.v_t;
    }
  }
  const bool bias_wave = p.bias != nullptr && wave == 0;
  if (bias_wave) ++my_n;
  auto issue_pair = [&](int pr) {
    const unsigned st = lds_base + (unsigned)((pr % NSTG) * PAIR);
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      if (wave + j * NW >= NI) continue;
      const int t = min(2 * pr + tsel[j], nsub - 1);      // a trailing odd sub-tile re-reads the last one (masked below)
      glds16(src[j] + (size_t)t * step[j] * 512, st + dst[j]);
    }
    if (bias_wave) glds4(p.bias + (size_t)b * p.bias_ld + min(pr * 64 + lane, p.bias_ld - 1), st + 2 * SUB);
  };
  auto wait_pair = [&]() {     // at most this wave's instructions of ONE pair still in flight
    if (my_n >= CPW + 1) wait_vmcnt<CPW + 1>();
    else if (my_n == CPW) wait_vmcnt<CPW>();
    else wait_vmcnt<(CPW > 1 ? CPW - 1 : 0)>();
  };

  issue_pair(0);
  if (nit > 1) issue_pair(1);
  // ---- Q fragments (B operand of K Q^T), behind the DMAs in the memory queue ----
  bf16x8 qh[KS], ql[KS];
  {
    const float* qp = p.q + ((size_t)b * p.Tq + (q_ok ? qi : 0)) * p.ldq + h * d;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks * 16 + lh * 8;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c2 = a;
      if (q_ok && c < d) a = *reinterpret_cast<const float4*>(qp + c);
      if (q_ok && c + 4 < d) c2 = *reinterpret_cast<const float4*>(qp + c + 4);
      a.x *= qscale; a.y *= qscale; a.z *= qscale; a.w *= qscale;
      c2.x *= qscale; c2.y *= qscale; c2.z *= qscale; c2.w *= qscale;
      u32x4 hw, lw;
      hw.x = apk(a.x, a.y); hw.y = apk(a.z, a.w); hw.z = apk(c2.x, c2.y); hw.w = apk(c2.z, c2.w);
      lw.x = apk(a.x - bf_lo(hw.x), a.y - bf_hi(hw.x)); lw.y = apk(a.z - bf_lo(hw.y), a.w - bf_hi(hw.y));
      lw.z = apk(c2.x - bf_lo(hw.z), c2.y - bf_hi(hw.z)); lw.w = apk(c2.z - bf_lo(hw.w), c2.w - bf_hi(hw.w));
      qh[ks] = __builtin_bit_cast(bf16x8, hw);
      ql[ks] = __builtin_bit_cast(bf16x8, lw);
    }
  }
  f32x16 o[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  for (int it = 0; it < nit; ++it) {
    if (it + 1 < nit) wait_pair(); else wait_vmcnt<0>();   // this wave's part of pair `it` has landed
    __syncthreads();                                       // everyone's has; everyone is done with pair it - 1
    if (it + 2 < nit) issue_pair(it + 2);                  // into the stage of pair it - 1
    const char* base0 = lds + (it % NSTG) * PAIR;
    const bool last = it + 1 == nit;

    // (KSP == 2: this wave multiplies sub-tile kh of the pair only; s[] / the loops below are indexed by uu, u = u0 + uu)
    const int u0 = KSP == 2 ? kh : 0;
    f32x16 s[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 kfh[NU], kfl[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        kfh[u] = *reinterpret_cast<const bf16x8*>(base0 + (u0 + u) * SUB + ks * 1024 + lane * 16);
        if (SPLIT) kfl[u] = *reinterpret_cast<const bf16x8*>(base0 + (u0 + u) * SUB + (KS + ks) * 1024 + lane * 16);
      }
      if (SPLIT) {
#pragma unroll
        for (int u = 0; u < NU; ++u) s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfl[u], qh[ks], s[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NU; ++u) s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh[u], ql[ks], s[u], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh[u], qh[ks], s[u], 0, 0, 0);
    }
    if (p.bias != nullptr) {                   // key bias (log2 domain; -1e30 beyond Tk)
      const float* bl = reinterpret_cast<const float*>(base0 + 2 * SUB);
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = *reinterpret_cast<const float4*>(bl + (u0 + u) * 32 + 8 * g + 4 * lh);
          s[u][4 * g] += bv.x; s[u][4 * g + 1] += bv.y; s[u][4 * g + 2] += bv.z; s[u][4 * g + 3] += bv.w;
        }
    }
    if (last && (p.Tk & 63) != 0) {            // keys past Tk of the last pair (a trailing odd sub-tile re-reads the last one)
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = (2 * it + u0 + u) * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
          if (key >= p.Tk) s[u][r] = -1e30f;
        }
    }
    float tmax = m_run;
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = NU == 2 ? __builtin_fmaxf(__builtin_fmaxf(tmax, s[0][r]), s[NU - 1][r]) : __builtin_fmaxf(tmax, s[0][r]);
    const float m_new = pair_max32(tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    const f32x2 mneg = {-m_new, -m_new};
    f32x2 psum = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 v = {s[u][r], s[u][r + 1]};
        v += mneg;
        v.x = __builtin_amdgcn_exp2f(v.x);
        v.y = __builtin_amdgcn_exp2f(v.y);
        psum += v;
        s[u][r] = v.x; s[u][r + 1] = v.y;
      }
    l_run = l_run * alpha + (psum.x + psum.y);
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nb][r] *= alpha;
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const char* vbase = base0 + (u0 + u) * SUB + KBL * 1024;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        u32x4 hw, lw;
        if (SPLIT && DV_ATTN_PF16) {
          // P as ONE fp16 plane, V as split fp16: O^T += Vlo^T P^T + Vhi^T P^T (dv_device.h DV_ATTN_PF16)
          hw.x = dv_cvt_pk_f16(s[u][kb * 8 + 0], s[u][kb * 8 + 1]); hw.y = dv_cvt_pk_f16(s[u][kb * 8 + 2], s[u][kb * 8 + 3]);
          hw.z = dv_cvt_pk_f16(s[u][kb * 8 + 4], s[u][kb * 8 + 5]); hw.w = dv_cvt_pk_f16(s[u][kb * 8 + 6], s[u][kb * 8 + 7]);
          const dv_f16x8 ph = __builtin_bit_cast(dv_f16x8, hw);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const dv_f16x8 vh = *reinterpret_cast<const dv_f16x8*>(vbase + (kb * NB + nb) * 1024 + lane * 16);
            const dv_f16x8 vl = *reinterpret_cast<const dv_f16x8*>(vbase + (2 * NB + kb * NB + nb) * 1024 + lane * 16);
            o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[nb], 0, 0, 0);
            o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[nb], 0, 0, 0);
          }
          continue;
        }
        hw.x = apk(s[u][kb * 8 + 0], s[u][kb * 8 + 1]); hw.y = apk(s[u][kb * 8 + 2], s[u][kb * 8 + 3]);
        hw.z = apk(s[u][kb * 8 + 4], s[u][kb * 8 + 5]); hw.w = apk(s[u][kb * 8 + 6], s[u][kb * 8 + 7]);
        const bf16x8 ph = __builtin_bit_cast(bf16x8, hw);
        bf16x8 pl;
        if (SPLIT && DV_ATTN_PLO) {
          auto lo_pair = [&](unsigned h2, float x0, float x1) {
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
          const bf16x8 vh = *reinterpret_cast<const bf16x8*>(vbase + (kb * NB + nb) * 1024 + lane * 16);
          if (SPLIT) {
            const bf16x8 vl = *reinterpret_cast<const bf16x8*>(vbase + (2 * NB + kb * NB + nb) * 1024 + lane * 16);
            o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, o[nb], 0, 0, 0);
            if (DV_ATTN_PLO) o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, o[nb], 0, 0, 0);
          }
          o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o[nb], 0, 0, 0);
        }
      }
    }
  }
  if (KSP == 2) {
    // merge the two key halves of a query block: half 1 hands (m, l, O) over through LDS (the ring is free now), half 0
    // rescales both to the common maximum, adds, and goes on to the store
    __syncthreads();                                       // every wave has left the key loop: the ring is free
    float* mg = reinterpret_cast<float*>(lds) + (size_t)qg * (2 + 16 * NB) * 64;
    if (kh == 1) {
      mg[lane] = m_run; mg[64 + lane] = l_run;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mg[(2 + nb * 16 + r) * 64 + lane] = o[nb][r];
    }
    __syncthreads();
    if (kh == 1) return;
    const float m1 = mg[lane], l1 = mg[64 + lane];
    const float m = fmaxf(m_run, m1);
    const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
    l_run = l_run * a0 + l1 * a1;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[nb][r] = o[nb][r] * a0 + mg[(2 + nb * 16 + r) * 64 + lane] * a1;
  }
  const float inv = 1.0f / pair_sum32(l_run);
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
          const int dv = nb * 32 + 8 * g + 4 * lh - row0;   // channel of the head held by registers 4g .. 4g+3
          if (dv >= 0 && dv < d) *reinterpret_cast<float4*>(p.o + obase + dv) = make_float4(v16[4 * g], v16[4 * g + 1], v16[4 * g + 2], v16[4 * g + 3]);
        }
      }
      // split bf16 planes for the to_out GEMM (hi = rne(v), lo = rne(v - hi)) as 16-byte stores: the lanes of a pair exchange halves
      // (dv_device.h store_planes16).  The head occupies fragment columns row0 .. row0 + d - 1 (multiples of 16): per 16-column half
      if (p.o_hi && ((d | row0) & 15) != 0) {         // heads of 8 / 24 ... channels: 8-byte stores, guarded per 4 columns
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dv = nb * 32 + 8 * g + 4 * lh - row0;
          if (dv >= 0 && dv < d) {
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
      } else if (p.o_hi) {
        const int c0 = nb * 32 - row0, c1 = c0 + 16;   // head channel of the fragment's columns 0 / 16
        const int pairs = ((c0 >= 0 && c0 < d) ? 1 : 0) | ((c1 >= 0 && c1 < d) ? 2 : 0);
        store_planes16(p.o_hi, p.o_lo, (size_t)((long long)obase + c0), lh, v16, pairs);
      }
    }
  }
}

template <int DP, int NW>
static constexpr int frag_smem(int npl) { return 3 * (2 * (DP / 16 + 2 * ((DP + 31) / 32)) * npl * 1024 + 256); }
template <int DP, int NW, int KSP = 1>
static void launch_att_frag(const AttnFragParams& p, dim3 grid, hipStream_t st) {
  const int smem3 = frag_smem<DP, NW>(2), smem1 = frag_smem<DP, NW>(1);
  if (p.nsplit == 3) hipLaunchKernelGGL((k_attention_frag<DP, NW, 3, KSP>), grid, dim3(64 * NW), smem3, st, p);
  else hipLaunchKernelGGL((k_attention_frag<DP, NW, 1, KSP>), grid, dim3(64 * NW), smem1, st, p);
}
template <int DP, int NW, int KSP = 1>
static hipError_t init_att_frag() {
  static_assert(KSP == 1 || (NW / 2) * (2 + 16 * ((DP + 31) / 32)) * 256 <= frag_smem<DP, NW>(1), "merge buffer fits the ring");
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attention_frag<DP, NW, 3, KSP>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, frag_smem<DP, NW>(2));
  return e != hipSuccess ? e : hipFuncSetAttribute(reinterpret_cast<const void*>(k_attention_frag<DP, NW, 1, KSP>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, frag_smem<DP, NW>(1));
}
static hipError_t attn_frag_init() {
  hipError_t e = hipSuccess;
#define ATTF_INIT(DP)                                                  \
  if (e == hipSuccess) e = init_att_frag<DP, 8>();                     \
  if (e == hipSuccess) e = init_att_frag<DP, 4>();                     \
  if (e == hipSuccess) e = init_att_frag<DP, 2>();                     \
  if (e == hipSuccess) e = init_att_frag<DP, 8, 2>();                  \
  if (e == hipSuccess) e = init_att_frag<DP, 4, 2>();
  ATTF_INIT(16) ATTF_INIT(32) ATTF_INIT(48) ATTF_INIT(64)
#undef ATTF_INIT
  return e;
}

static int attn_no_xcd_map() { static const int v = [] { const char* e = getenv("DVITS_ATTN_XCD"); return (e && e[0] == '0') ? 1 : 0; }(); return v; }

hipError_t launch_attention_frag(const AttnFragParams& pin, hipStream_t st) {
  AttnFragParams p = pin;
  p.no_xcd_map = attn_no_xcd_map();
  if (p.d % 16 != 0 || p.d > 64 || p.d <= 0 || (p.nsplit != 1 && p.nsplit != 3) || !p.q || !p.kf_hi || !p.vf_hi) return hipErrorInvalidValue;
  if (p.nsplit == 3 && (!p.kf_lo || !p.vf_lo)) return hipErrorInvalidValue;
  if (p.self_layout && (p.d & 15)) return hipErrorInvalidValue;
  // keys beyond Tk inside the last 32-key tile (zero fragments of the hoisted prompt; finite padding rows of a padded self-
  // attention row space) are kept out of the softmax by the kernel itself (key >= Tk -> -1e30 in the last pair); a bias row,
  // if given, must cover whole tiles
  if (p.bias && p.bias_ld < (p.Tk + 31) / 32 * 32) return hipErrorInvalidValue;
  const long waves = (long)p.B * p.H * ((p.Tq + 31) / 32);
  static const int nw8_min = [] { const char* e = getenv("DVITS_ATTNF_NW8"); return e ? atoi(e) : 1100; }();
  static const int nw4_min = [] { const char* e = getenv("DVITS_ATTNF_NW4"); return e ? atoi(e) : 64; }();
  const int nw = waves >= nw8_min ? 8 : (waves >= nw4_min ? 4 : 2);
  // Key split inside the workgroup (KSP = 2) where the plain grid leaves CUs without a workgroup and there are at least two
  // 64-key pairs to share out: the short levels (T <= 256: 64-128 workgroups of four waves).  DVITS_ATTNF_KSP=<max plain
  // workgroups> (default 160; 0: never).
  static const int ksp_max = [] { const char* e = getenv("DVITS_ATTNF_KSP"); return e ? atoi(e) : 160; }();
  const int plain_wgs = ((p.Tq + 32 * nw - 1) / (32 * nw)) * p.H * p.B;
  const bool ksp = nw >= 4 && plain_wgs <= ksp_max && p.Tk > 64;
  const int nq = ksp ? nw / 2 : nw;
  dim3 grid(((p.Tq + 32 * nq - 1) / (32 * nq)) * p.H * p.B);
#define ATTF(DP)                                                            \
  if (nw == 8) { if (ksp) launch_att_frag<DP, 8, 2>(p, grid, st); else launch_att_frag<DP, 8>(p, grid, st); }          \
  else if (nw == 4) { if (ksp) launch_att_frag<DP, 4, 2>(p, grid, st); else launch_att_frag<DP, 4>(p, grid, st); }     \
  else launch_att_frag<DP, 2>(p, grid, st);
  if (p.d == 16) { ATTF(16) } else if (p.d == 32) { ATTF(32) } else if (p.d == 48) { ATTF(48) } else { ATTF(64) }
#undef ATTF
  return hipGetLastError();
}

hipError_t launch_attention(const AttnParams& pin, hipStream_t st) {
  AttnParams p = pin;
  p.no_xcd_map = attn_no_xcd_map();
  if (p.d % 4 != 0 || p.d > 64 || p.d <= 0 || (p.nsplit != 1 && p.nsplit != 3)) return hipErrorInvalidValue;
  // queries per workgroup: the K / V tiles are converted once per workgroup, so more waves per workgroup amortise
  // that, fewer spread the launch over more CUs.  Measured at the bench shape (after the conversion was made cheap):
  // 8 waves (256 queries) for the 1024-frame level, 4 waves below it (DVITS_ATTN_NW8 / DVITS_ATTN_NW4 = wave-count
  // thresholds for 8 / 4 waves per workgroup)
  const long waves = (long)p.B * p.H * ((p.Tq + 31) / 32);
  static const int nw8_min = [] { const char* e = getenv("DVITS_ATTN_NW8"); return e ? atoi(e) : 1100; }();
  static const int nw4_min = [] { const char* e = getenv("DVITS_ATTN_NW4"); return e ? atoi(e) : 64; }();
  const int nw = waves >= nw8_min ? 8 : (waves >= nw4_min ? 4 : (waves >= 16 ? 2 : 1));
  dim3 grid(((p.Tq + 32 * nw - 1) / (32 * nw)) * p.H * p.B);
  const int dp = (p.d + 15) / 16 * 16;
#define ATT(DP)                                         \
  if (nw == 8) launch_att<DP, 8>(p, grid, st);          \
  else if (nw == 4) launch_att<DP, 4>(p, grid, st);     \
  else if (nw == 2) launch_att<DP, 2>(p, grid, st);     \
  else launch_att<DP, 1>(p, grid, st);
  if (dp == 16) { ATT(16) } else if (dp == 32) { ATT(32) } else if (dp == 48) { ATT(48) } else { ATT(64) }
#undef ATT
  return hipGetLastError();
}
