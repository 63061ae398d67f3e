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
// K / V^T tiles of 32 keys are converted fp32 -> split bf16 once per workgroup and shared by its NW
// waves (32*NW queries); two LDS buffers, global loads of tile t+1 in flight during tile t.
#include "dv_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned apk(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

template <int DP, int NW, int NSPLIT>
__global__ __launch_bounds__(64 * NW) void k_attention(const AttnParams p) {
  constexpr bool SPLIT = NSPLIT == 3;
  constexpr int NPL = SPLIT ? 2 : 1;
  constexpr int KS = DP / 16;                      // k-steps of K Q^T
  constexpr int NB = (DP + 31) / 32;               // 32-wide output-channel blocks
  constexpr int KP = DP * 2 + 16;                  // K row pitch (bytes): odd number of 16-byte slots
  constexpr int VP = 80;                           // V^T row pitch: 32 keys * 2 B + 16
  constexpr int K_PL = 32 * KP, V_PL = NB * 32 * VP;
  constexpr int BUF = (K_PL + V_PL) * NPL + 128;   // + 32 floats of key bias
  constexpr int NT = 64 * NW;
  constexpr int TASKS = 4 * DP;                    // (key pair, 4-channel group) load tasks per tile
  constexpr int TPT = (TASKS + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int d = p.d;
  const int qi = blockIdx.x * (32 * NW) + wave * 32 + l31;
  const bool q_ok = qi < p.Tq;

  // scores are kept in the log2 domain (q pre-scaled by d^-1/2 * log2 e, key bias by log2 e): the softmax
  // exponentials are bare v_exp_f32
  constexpr float LOG2E = 1.44269504088896340736f;
  const float qscale = p.scale * LOG2E;
  // ---- Q fragments (B operand of K Q^T): lane (query, lh) holds channels ks*16 + lh*8 .. +8 ----
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

  // ---- K / V tile staging: task = (key pair kp, channel group c4) ----
  float4 rk0[TPT], rk1[TPT], rv0[TPT], rv1[TPT];
  float rbias = 0.f;
  auto load_tile = [&](int kt0) {
#pragma unroll
    for (int i = 0; i < TPT; ++i) {
      const int task = tid + i * NT;
      const int kp = task / (DP / 4), c4 = (task - kp * (DP / 4)) * 4;
      const int k0 = min(kt0 + 2 * kp, p.Tk - 1), k1 = min(kt0 + 2 * kp + 1, p.Tk - 1);   // clamped: masked by the bias
      const bool live = task < TASKS && c4 < d;
      const int cc = live ? c4 : 0;
      const size_t o0 = ((size_t)b * p.Tk + k0), o1 = ((size_t)b * p.Tk + k1);
      rk0[i] = *reinterpret_cast<const float4*>(p.k + o0 * p.ldk + h * d + cc);
      rk1[i] = *reinterpret_cast<const float4*>(p.k + o1 * p.ldk + h * d + cc);
      rv0[i] = *reinterpret_cast<const float4*>(p.v + o0 * p.ldv + h * d + cc);
      rv1[i] = *reinterpret_cast<const float4*>(p.v + o1 * p.ldv + h * d + cc);
    }
    if (tid < 32) {
      const int key = kt0 + tid;
      rbias = (key < p.Tk) ? (p.bias ? p.bias[(size_t)b * p.Tk + key] * LOG2E : 0.f) : -1e30f;
    }
  };
  auto store_tile = [&](int buf) {
    char* base = lds + buf * BUF;
    char* k_hi = base;
    char* k_lo = base + K_PL;
    char* v_hi = base + NPL * K_PL;
    char* v_lo = v_hi + V_PL;
    float* bl = reinterpret_cast<float*>(base + (K_PL + V_PL) * NPL);
#pragma unroll
    for (int i = 0; i < TPT; ++i) {
      const int task = tid + i * NT;
      if (task >= TASKS) continue;
      const int kp = task / (DP / 4), c4 = (task - kp * (DP / 4)) * 4;
      float4 a = rk0[i], c = rk1[i], va = rv0[i], vc = rv1[i];
      if (c4 >= d) { a = make_float4(0.f, 0.f, 0.f, 0.f); c = a; va = a; vc = a; }   // zero the padded channels
      // K rows 2kp, 2kp+1: 4 channels -> 8 bytes per plane
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
      // V^T: channel rows, key slot = key index with bits 2 and 3 swapped; keys 2kp, 2kp+1 are adjacent slots
      const int j = 2 * kp;
      const int slot = (j & 0x13) | ((j & 4) << 1) | ((j & 8) >> 1);
      const float ve0[4] = {va.x, va.y, va.z, va.w}, ve1[4] = {vc.x, vc.y, vc.z, vc.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned hw = apk(ve0[e], ve1[e]);
        *reinterpret_cast<unsigned*>(v_hi + (c4 + e) * VP + slot * 2) = hw;
        if (SPLIT) *reinterpret_cast<unsigned*>(v_lo + (c4 + e) * VP + slot * 2) = apk(ve0[e] - bf_lo(hw), ve1[e] - bf_hi(hw));
      }
    }
    if (tid < 32) bl[tid] = rbias;
  };

  const int ntile = (p.Tk + 31) / 32;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const bool more = t + 1 < ntile;
    if (more) load_tile((t + 1) * 32);
    const char* base = lds + (t & 1) * BUF;
    const char* k_hi = base;
    const char* k_lo = base + K_PL;
    const char* v_hi = base + NPL * K_PL;
    const char* v_lo = v_hi + V_PL;
    const float* bl = reinterpret_cast<const float*>(base + (K_PL + V_PL) * NPL);

    // ---- S^T = K Q^T ----
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int off = l31 * KP + (ks * 2 + lh) * 16;
      const bf16x8 kh = *reinterpret_cast<const bf16x8*>(k_hi + off);
      if (SPLIT) {
        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(k_lo + off);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[ks], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[ks], s, 0, 0, 0);
      }
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[ks], s, 0, 0, 0);
    }
    // ---- online softmax (lane = query; registers = keys) ----
    // the key bias (attention mask / keys past Tk) is only added where there is one: wave-uniform branch
    if (p.bias != nullptr || (!more && (p.Tk & 31) != 0)) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4*>(bl + 8 * g + 4 * lh);
        s[4 * g] += bv.x; s[4 * g + 1] += bv.y; s[4 * g + 2] += bv.z; s[4 * g + 3] += bv.w;
      }
    }
    float tmax = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
      psum += s[r];
    }
    l_run = l_run * alpha + psum;
    if (__any(alpha != 1.0f)) {        // the running maximum moved for some query of this wave
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nb][r] *= alpha;
    }
    // ---- O^T += V^T P^T: P^T operand of k-block kb = this lane's score registers 8kb..8kb+7 ----
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      u32x4 hw, lw;
      hw.x = apk(s[kb * 8 + 0], s[kb * 8 + 1]); hw.y = apk(s[kb * 8 + 2], s[kb * 8 + 3]);
      hw.z = apk(s[kb * 8 + 4], s[kb * 8 + 5]); hw.w = apk(s[kb * 8 + 6], s[kb * 8 + 7]);
      const bf16x8 ph = __builtin_bit_cast(bf16x8, hw);
      bf16x8 pl;
      if (SPLIT) {
        lw.x = apk(s[kb * 8 + 0] - bf_lo(hw.x), s[kb * 8 + 1] - bf_hi(hw.x));
        lw.y = apk(s[kb * 8 + 2] - bf_lo(hw.y), s[kb * 8 + 3] - bf_hi(hw.y));
        lw.z = apk(s[kb * 8 + 4] - bf_lo(hw.z), s[kb * 8 + 5] - bf_hi(hw.z));
        lw.w = apk(s[kb * 8 + 6] - bf_lo(hw.w), s[kb * 8 + 7] - bf_hi(hw.w));
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
    if (more) store_tile((t + 1) & 1);
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
    const size_t obase = ((size_t)b * p.Tq + qi) * p.ldo + h * d;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dv = nb * 32 + 8 * g + 4 * lh;
        if (dv < d) {
          const float4 v = make_float4(o[nb][4 * g] * inv, o[nb][4 * g + 1] * inv, o[nb][4 * g + 2] * inv,
                                       o[nb][4 * g + 3] * inv);
          if (p.o) *reinterpret_cast<float4*>(p.o + obase + dv) = v;
          if (p.o_hi) {   // split bf16 planes for the to_out GEMM: hi = rne(v), lo = rne(v - hi)
            uint2 hh, ll;
            hh.x = apk(v.x, v.y); hh.y = apk(v.z, v.w);
            *reinterpret_cast<uint2*>(p.o_hi + obase + dv) = hh;
            if (p.o_lo) {
              ll.x = apk(v.x - bf_lo(hh.x), v.y - bf_hi(hh.x));
              ll.y = apk(v.z - bf_lo(hh.y), v.w - bf_hi(hh.y));
              *reinterpret_cast<uint2*>(p.o_lo + obase + dv) = ll;
            }
          }
        }
      }
  }
}

template <int DP, int NW>
static void launch_att(const AttnParams& p, dim3 grid, hipStream_t st) {
  if (p.nsplit == 3) hipLaunchKernelGGL((k_attention<DP, NW, 3>), grid, dim3(64 * NW), 0, st, p);
  else hipLaunchKernelGGL((k_attention<DP, NW, 1>), grid, dim3(64 * NW), 0, st, p);
}

hipError_t launch_attention(const AttnParams& p, hipStream_t st) {
  if (p.d % 4 != 0 || p.d > 64 || p.d <= 0 || (p.nsplit != 1 && p.nsplit != 3)) return hipErrorInvalidValue;
  // queries per workgroup: enough workgroups to cover the 256 CUs even at the short levels
  const long waves = (long)p.B * p.H * ((p.Tq + 31) / 32);
  const int nw = waves >= 2048 ? 4 : (waves >= 512 ? 2 : 1);
  dim3 grid((p.Tq + 32 * nw - 1) / (32 * nw), p.H, p.B);
  const int dp = (p.d + 15) / 16 * 16;
#define ATT(DP)                                         \
  if (nw == 4) launch_att<DP, 4>(p, grid, st);          \
  else if (nw == 2) launch_att<DP, 2>(p, grid, st);     \
  else launch_att<DP, 1>(p, grid, st);
  if (dp == 16) { ATT(16) } else if (dp == 32) { ATT(32) } else if (dp == 48) { ATT(48) } else { ATT(64) }
#undef ATT
  return hipGetLastError();
}
