// Device bodies of the streaming kernels that also run inside the persistent per-XCD schedule (persist.hip):
// GroupNorm apply and the fp32 -> split-plane copy, plus the small helpers of kernels_misc.hip.
#pragma once
#include "dv_common.h"
#include "dv_device.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// (all call sites run with the whole wave active: DPP adds inside each row of 16 lanes + four row sums through scalar registers,
// dv_device.h wave_sum64 - the twelve dependent ds_bpermute pairs of the butterfly were the head of every k_gn_apply workgroup)
__device__ __forceinline__ double wave_sum_d(double v) { return wave_sum64(v); }
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) { return dv_cvt_pk_bf16(lo, hi); }
// 4 floats -> 4 bf16 hi (uint2) + 4 bf16 lo (uint2), hi = rne(x), lo = rne(x - hi)
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  hi.x = pk_bf16(v.x, v.y);
  hi.y = pk_bf16(v.z, v.w);
  lo.x = pk_bf16(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xffff0000u));
  lo.y = pk_bf16(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xffff0000u));
}
__device__ __forceinline__ void split1(float v, bf16_t& hi, bf16_t& lo) {
  const unsigned h = pk_bf16(v, 0.f);
  hi = (bf16_t)(h & 0xffffu);
  lo = (bf16_t)(pk_bf16(v - __uint_as_float(h << 16), 0.f) & 0xffffu);
}

// fp32 -> split planes over float4 index range [i0, i1), NT*stride threads cooperating
template <bool SC1>
__device__ __forceinline__ void split_body(const float* in, bf16_t* hi, bf16_t* lo, int64_t i0, int64_t i1, int64_t first,
                                           int64_t stride) {
  for (int64_t i = i0 + first; i < i1; i += stride) {
    uint2 h, l;
    split4(ld_mut4<SC1>(in + 4 * i), h, l);
    reinterpret_cast<uint2*>(hi)[i] = h;
    if (lo) reinterpret_cast<uint2*>(lo)[i] = l;
  }
}

// NT threads; (chunk, g, b) = the launch's (blockIdx.x, .y, .z).  SC1: loads of activations / statistics / temb rows
// bypass this CU's L1 (written earlier in the same launch by other CUs: persist.hip).
template <int NT, bool SC1>
__device__ __forceinline__ void gn_apply_body(const GnApplyParams& p, int rows_per_block, const int chunk, const int g,
                                              const int b) {
  // grid = (frame chunks, groups, batch): a workgroup normalises `rows_per_block` frames of ONE group
  // (cg channels) of one batch item, so it only reduces that group's slice of the statistics slab
  __shared__ float s_scale[512], s_shift[512];
  __shared__ double s_red[2 * (NT / 64)];
  const int ctot = p.c0 + p.c1, G = p.groups, cg = ctot / G;
  const int tid = threadIdx.x;
  const int cbase = g * cg;
  // the elements this thread normalises do not depend on the statistics: fetch the first two items now, so their
  // (cold-L2) latency overlaps the slab reduction instead of following it
  const int ncol4 = cg >> 2;
  const int t0 = chunk * rows_per_block, t1r = min(p.T, t0 + rows_per_block);
  const int total = (t1r - t0) * ncol4;
  auto load_item = [&](int i) {
    const int r = i / ncol4, c = cbase + (i - r * ncol4) * 4;
    const size_t row = (size_t)b * p.T + t0 + r;
    return ld_mut4<SC1>(c < p.c0 ? p.a0 + row * p.c0 + c : p.a1 + row * p.c1 + (c - p.c0));
  };
  constexpr int PF = 2;
  float4 pv[PF];
#pragma unroll
  for (int k = 0; k < PF; ++k) pv[k] = tid + k * NT < total ? load_item(tid + k * NT) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.scale_in) {
    for (int c = tid; c < cg; c += NT) {
      s_scale[c] = ld_mut1<SC1>(p.scale_in + (size_t)b * ctot + cbase + c);
      s_shift[c] = ld_mut1<SC1>(p.shift_in + (size_t)b * ctot + cbase + c);
    }
  } else if (p.st16_0) {
    // block statistics: the group's RB x (cg / 16) entries are reduced by the first wave in one pass (fp64), while every
    // thread already holds the affine parameters of its channels (fetched before the statistics arrive)
    const int RB = p.T >> 5, nvb = cg >> 4;
    const int Tv = p.Tv ? p.Tv : p.T;                  // frames that exist: the last block of a padded utterance is partial
    float pg[2], pb[2], pts[2], ptb[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int cc = cbase + min(tid + k * NT, cg - 1);
      pg[k] = p.gamma[cc];
      pb[k] = p.beta[cc];
      pts[k] = p.tscale ? 1.0f + ld_mut1<SC1>(p.tscale + (size_t)b * p.ld_t + cc) : 1.0f;
      ptb[k] = p.tshift ? ld_mut1<SC1>(p.tshift + (size_t)b * p.ld_t + cc) : 0.0f;
    }
    if (tid < 64) {
      double s1 = 0, q = 0;
      for (int e = tid; e < RB * nvb; e += 64) {
        const int rb = e / nvb, vb = g * nvb + (e - rb * nvb);
        const bool first = vb * 16 < p.c0;
        const float2* st = reinterpret_cast<const float2*>(first ? p.st16_0 : p.st16_1);
        const int nb = (first ? p.c0 : p.c1) >> 4, vbl = first ? vb : vb - (p.c0 >> 4);
        const float2 v = ld_mut2<SC1>(st + (size_t)(b * RB + rb) * nb + vbl);
        s1 += (double)v.x;
        q += (double)v.y + (double)v.x * (double)v.x / (double)(16 * min(32, Tv - 32 * rb));   // = the block's sum of squares
      }
      s1 = wave_sum_d(s1);
      q = wave_sum_d(q);
      if (tid == 0) { s_red[0] = s1; s_red[1] = q; }
    }
    __syncthreads();
    const double n = (double)cg * (double)Tv;
    const double mean = s_red[0] / n;
    double var = s_red[1] / n - mean * mean;
    var = var > 0 ? var : 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = tid + k * NT;
      if (c < cg) {
        const float a = rstd * pg[k];
        s_scale[c] = a * pts[k];
        s_shift[c] = fmaf(pb[k] - (float)mean * a, pts[k], ptb[k]);
      }
    }
  } else {
    const int RB = (p.T + 31) >> 5;      // 32-row blocks per batch item (T % 32 == 0, or B == 1 with a partial last block)
    double s1 = 0, s2 = 0;
    for (int item = tid; item < cg * RB; item += NT) {
      const int rb = item / cg, c = cbase + (item - rb * cg);
      const bool first = c < p.c0;
      const float2* slab = reinterpret_cast<const float2*>(first ? p.slab0 : p.slab1);
      const int ld = first ? p.c0 : p.c1, cc = first ? c : c - p.c0;
      const float2 v = ld_mut2<SC1>(slab + (size_t)(b * RB + rb) * ld + cc);
      s1 += v.x;
      s2 += v.y;
    }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if ((tid & 63) == 0) { s_red[(tid >> 6) * 2] = s1; s_red[(tid >> 6) * 2 + 1] = s2; }
    __syncthreads();
    double t1 = 0, t2 = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { t1 += s_red[2 * w]; t2 += s_red[2 * w + 1]; }
    const double n = (double)cg * (double)p.T;
    const double mean = t1 / n;
    double var = t2 / n - mean * mean;
    var = var > 0 ? var : 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
    for (int c = tid; c < cg; c += NT) {
      const int cc = cbase + c;
      const float a = rstd * p.gamma[cc];
      const float sh = p.beta[cc] - (float)mean * a;
      const float ts = p.tscale ? 1.0f + ld_mut1<SC1>(p.tscale + (size_t)b * p.ld_t + cc) : 1.0f;
      const float tb = p.tshift ? ld_mut1<SC1>(p.tshift + (size_t)b * p.ld_t + cc) : 0.0f;
      s_scale[c] = a * ts;
      s_shift[c] = fmaf(sh, ts, tb);
    }
  }
  __syncthreads();
  for (int i = tid, k = 0; i < total; i += NT, ++k) {
    const int r = i / ncol4, j = i - r * ncol4, c = cbase + j * 4;
    const size_t row = (size_t)b * p.T + t0 + r;
    const float4 v = k == 0 ? pv[0] : (k == 1 ? pv[1] : load_item(i));
    const float4 sc = *reinterpret_cast<const float4*>(s_scale + j * 4);
    const float4 sh = *reinterpret_cast<const float4*>(s_shift + j * 4);
    float4 y;
    y.x = fmaf(v.x, sc.x, sh.x); y.y = fmaf(v.y, sc.y, sh.y); y.z = fmaf(v.z, sc.z, sh.z); y.w = fmaf(v.w, sc.w, sh.w);
    if (p.silu) {
      y.x = y.x / (1.0f + __expf(-y.x)); y.y = y.y / (1.0f + __expf(-y.y));
      y.z = y.z / (1.0f + __expf(-y.z)); y.w = y.w / (1.0f + __expf(-y.w));
    }
    uint2 h, l;
    split4(y, h, l);
    const size_t o = (row * ctot + c) >> 2;
    dv_st8(reinterpret_cast<uint2*>(p.out_hi) + o, h);
    if (p.out_lo) dv_st8(reinterpret_cast<uint2*>(p.out_lo) + o, l);
    if (p.raw_hi) {
      split4(v, h, l);
      dv_st8(reinterpret_cast<uint2*>(p.raw_hi) + o, h);
      if (p.raw_lo) dv_st8(reinterpret_cast<uint2*>(p.raw_lo) + o, l);
    }
  }
}

