// Flash-style multi-head attention for gfx950 (MI355X): softmax(q k^T * d^-1/2 + bias) v
// without materialising the score matrix (reference unet1d/attention_processor.py:1019-1035
// = F.scaled_dot_product_attention on [B,H,T,d] views; additive key bias [B,Tk] from
// prepare_attention_mask :309-336).
//
// Exact-fp32 contractions on the matrix cores: v_mfma_f32_32x32x2_f32 (f32 in, f32
// accumulate; bitwise an fmaf chain).  One wave owns 32 queries.  Both products are
// computed TRANSPOSED so that every softmax quantity is lane-local:
//   S^T[key][query] = K Q^T   -> lane (query = lane&31) holds 16 of the tile's 32 keys
//                               (the other 16 sit in lane^32): row max / row sum need one
//                               64-lane shuffle, no LDS
//   O^T[dv][query]  = V^T P^T -> the P^T operand of step r IS the score register r
//                               (keys {r', r'+4} for lane halves 0/1), so P never moves;
//                               the per-query rescale exp(m_old-m_new) and the final 1/l
//                               are per-lane scalars
// K/V tiles of 32 keys are staged through LDS by the whole workgroup (NW waves = 32*NW
// queries share them); latency is hidden by occupancy (16 KB LDS / workgroup).
#include "dv_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int DP, int NW>
__global__ __launch_bounds__(64 * NW) void k_attention(const AttnParams p) {
  constexpr int NB = DP / 32 > 0 ? DP / 32 : 1;      // 32-wide output-channel blocks
  constexpr int KLD = DP + 1;                        // K tile pitch (floats): odd -> conflict-free column reads
  constexpr int VLD = NB * 32;                       // V tile pitch
  __shared__ float Kl[32 * KLD];
  __shared__ float Vl[32 * VLD];
  __shared__ float Bl[32];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int d = p.d;
  const int qi = blockIdx.x * (32 * NW) + wave * 32 + l31;
  const bool q_ok = qi < p.Tq;

  // Q fragment: q[s] = Q[qi][2s + lh] * scale
  float q[DP / 2];
  {
    const float* qp = p.q + ((size_t)b * p.Tq + (q_ok ? qi : 0)) * p.ldq + h * d;
#pragma unroll
    for (int s = 0; s < DP / 2; ++s) {
      const int c = 2 * s + lh;
      q[s] = (q_ok && c < d) ? qp[c] * p.scale : 0.f;
    }
  }

  f32x16 o[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int nchunk = 32 * (DP / 4);                  // float4 chunks per tile
  for (int kt0 = 0; kt0 < p.Tk; kt0 += 32) {
    // ---- stage K, V (and bias) tile ----
    for (int c = tid; c < nchunk; c += 64 * NW) {
      const int row = c / (DP / 4), c4 = (c % (DP / 4)) * 4;
      const int key = kt0 + row;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (key < p.Tk && c4 < d) {
        kv = *reinterpret_cast<const float4*>(p.k + ((size_t)b * p.Tk + key) * p.ldk + h * d + c4);
        vv = *reinterpret_cast<const float4*>(p.v + ((size_t)b * p.Tk + key) * p.ldv + h * d + c4);
      }
      float* kd = Kl + row * KLD + c4;
      kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
      if (c4 < VLD) *reinterpret_cast<float4*>(Vl + row * VLD + c4) = vv;
    }
    if (tid < 32) {
      const int key = kt0 + tid;
      Bl[tid] = (key < p.Tk) ? (p.bias ? p.bias[(size_t)b * p.Tk + key] : 0.f) : -1e30f;
    }
    __syncthreads();

    // ---- S^T = K Q^T ----
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < DP / 2; ++st) {
      if (2 * st < d) {
        const float a = Kl[l31 * KLD + 2 * st + lh];
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(a, q[st], s, 0, 0, 0);
      }
    }
    // ---- online softmax (lane = query; registers = keys) ----
    float tmax = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kl = (r & 3) + 8 * (r >> 2) + 4 * lh;
      s[r] += Bl[kl];
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __expf(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __expf(s[r] - m_new);
      psum += s[r];
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[nb][r] *= alpha;
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kl = (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if (nb * 32 < d) {
          const float a = Vl[kl * VLD + nb * 32 + l31];
          o[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], o[nb], 0, 0, 0);
        }
      }
    }
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
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hh.x) : "v"(v.x), "v"(v.y));
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hh.y) : "v"(v.z), "v"(v.w));
            *reinterpret_cast<uint2*>(p.o_hi + obase + dv) = hh;
            if (p.o_lo) {
              const float rx = v.x - __uint_as_float(hh.x << 16), ry = v.y - __uint_as_float(hh.x & 0xffff0000u);
              const float rz = v.z - __uint_as_float(hh.y << 16), rw = v.w - __uint_as_float(hh.y & 0xffff0000u);
              asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(ll.x) : "v"(rx), "v"(ry));
              asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(ll.y) : "v"(rz), "v"(rw));
              *reinterpret_cast<uint2*>(p.o_lo + obase + dv) = ll;
            }
          }
        }
      }
  }
}

hipError_t launch_attention(const AttnParams& p, hipStream_t st) {
  if (p.d % 4 != 0 || p.d > 64 || p.d <= 0) return hipErrorInvalidValue;
  const bool wide = p.Tq >= 512;
  const int qpb = wide ? 128 : 64;
  dim3 grid((p.Tq + qpb - 1) / qpb, p.H, p.B);
  const int dp = p.d <= 16 ? 16 : (p.d <= 32 ? 32 : 64);
#define ATT(DP)                                                                        \
  if (wide) hipLaunchKernelGGL((k_attention<DP, 4>), grid, dim3(256), 0, st, p);       \
  else hipLaunchKernelGGL((k_attention<DP, 2>), grid, dim3(128), 0, st, p);
  if (dp == 16) { ATT(16) } else if (dp == 32) { ATT(32) } else { ATT(64) }
#undef ATT
  return hipGetLastError();
}
