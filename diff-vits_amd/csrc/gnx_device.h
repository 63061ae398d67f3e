// GroupNorm of a producer's OWN output finished inside its launch (GnxParams, dv_common.h): the part every producer kernel
// shares - gemm_tile.h (the contraction kernel) and kernels_chain.hip (k_chain_ff).  The caller has written its tile's 32x16
// block statistics (sum, M2 about the block mean) to the exchange words; gnx_finish_table
//   1. reduces, one wave per group, the statistics of the groups its tile's columns belong to (polling the exchange words of
//      the other workgroups of the launch - all resident, the planner checked; bounded and flagged, never a hang),
//   2. leaves the per-column affine (scale, shift) of its tile in shared memory (sh.gA / sh.gB) for the caller's own
//      normalise-and-store pass, and
//   3. for a concatenated consumer normalises a slice of the skip tensor's columns for the tile's rows itself.
#pragma once
#include "dv_common.h"
#include "dv_device.h"

struct GnxTile {
  int M, N;            // the producer's output [M, N]
  int T_out, Tv_out;   // row pitch per utterance, frames that exist
  int m0, n0;          // the tile's first row / column
  int bm, bn;          // its extent
  int bq;              // utterance of row m0 (a tile never spans two)
};
constexpr int DV_GSK = 128;                          // widest skip slice per workgroup (gemm_gnx_plan / chain_ff_gnx_plan check)
template <int BN>
struct GnxShared {
  float2 gst[65];                                    // (mean, rstd) by group index - g_lo (at most 64 groups)
  __attribute__((aligned(16))) float gA[BN < 64 ? 64 : BN], gB[BN < 64 ? 64 : BN];
  __attribute__((aligned(16))) float gA2[DV_GSK], gB2[DV_GSK];
};

// Concatenated consumer (GnxParams sk_*; reference unet_1d_blocks.py:2085,2187 -> resnet.py:594: norm1 of an up-path resnet
// runs over [h | skip]): this GEMM produces h, the skip tensor and ITS block statistics have been in memory since the down
// path.  Groups are those of the concatenation ((N + sk_c) / groups channels each, h first): a group's entries come from the
// exchange words (h's blocks, polled) and / or from the skip's stored statistics (loaded), so a group that straddles the
// boundary needs no special case.  Besides its own tile every workgroup normalises a SLICE of the skip's columns for its rows
// (and writes the raw planes the folded 1x1 shortcut reads): the k_gn_apply launch of the concatenation disappears.
// `nwa` waves (threads 0 .. 64 nwa - 1) are still in the kernel and call this together; `trace(k)` stamps a phase (trace build).
template <int BN, typename TR>
__device__ __forceinline__ void gnx_finish_table(const GnxParams& gx, const GnxTile& t, GnxShared<BN>& sh, const int tid, const int lane,
                                                 const int wave, const int nwa, TR trace) {
  const int skc = gx.sk_c;                      // 0: no concatenated consumer
  const int cpg = (t.N + skc) / gx.groups, bq = t.bq;
  // this workgroup's slice of the skip's 16-channel blocks: [sb0, sb1)
  int sb0 = 0, sb1 = 0;
  if (skc > 0) {
    const int nbs = skc >> 4, tn = (t.N + t.bn - 1) / t.bn, per = (nbs + tn - 1) / tn, j = t.n0 / t.bn;
    sb0 = min(j * per, nbs); sb1 = min(sb0 + per, nbs);
  }
  const int sw = (sb1 - sb0) * 16;                 // slice width in channels
  float pg = 0.f, pb = 0.f, pts = 1.f, ptb = 0.f;  // this thread's column: affine + temb scale / shift (independent of the statistics)
  if (tid < t.bn) {
    const int c = min(t.n0 + tid, t.N - 1);
    pg = gx.gamma[c]; pb = gx.beta[c];
    if (gx.tscale) pts = 1.0f + gx.tscale[(size_t)bq * gx.ld_t + c];
    if (gx.tshift) ptb = gx.tshift[(size_t)bq * gx.ld_t + c];
  }
  float pg2 = 0.f, pb2 = 0.f;                      // ... and its column of the skip slice (concatenated channel N + ...)
  if (tid < sw) { pg2 = gx.gamma[t.N + sb0 * 16 + tid]; pb2 = gx.beta[t.N + sb0 * 16 + tid]; }
  // statistics of the groups this tile's columns (and the skip slice's) belong to: one wave per group, fp64, fixed order
  const int g_lo = t.n0 / cpg, g_hi = (min(t.n0 + t.bn, t.N) - 1) / cpg;
  const int n1 = g_hi - g_lo + 1;
  const int g2_lo = sw > 0 ? max((t.N + sb0 * 16) / cpg, g_hi + 1) : 0, g2_hi = sw > 0 ? (t.N + sb1 * 16 - 1) / cpg : -1;
  const int ng = n1 + max(g2_hi - g2_lo + 1, 0);
  const int RB = t.T_out >> 5, nvb = cpg >> 4, ncb = t.N >> 4, ncs = skc >> 4;
  // the skip slice of this tile's rows, 8 channels per item: what a thread will normalise does not depend on the statistics -
  // its first items are requested NOW (cold: written a whole down / up path ago), so their latency runs under the polls below
  constexpr int PF = 2;
  const int sw8 = sw >> 3, total = t.bm * sw8, nthr = nwa * 64;
  auto item = [&](int i, int& cl) -> size_t {      // offset of item i in the skip tensor (its planes), ~0: a row beyond M
    const int r = i / sw8, m = t.m0 + r;
    cl = (i - r * sw8) * 8;
    return m < t.M ? (size_t)m * skc + sb0 * 16 + cl : ~(size_t)0;
  };
  float4 pv[PF][2];
#pragma unroll
  for (int k = 0; k < PF; ++k) {
    int cl;
    const size_t o = sw > 0 && tid + k * nthr < total ? item(tid + k * nthr, cl) : ~(size_t)0;
    if (o != ~(size_t)0) { pv[k][0] = *reinterpret_cast<const float4*>(gx.sk_x + o); pv[k][1] = *reinterpret_cast<const float4*>(gx.sk_x + o + 4); }
    else pv[k][0] = pv[k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  trace(19);
  // (last group first: the skip slice's own groups are plain - cold - loads, which run while the partners' words arrive)
  for (int gi = ng - 1 - wave; gi >= 0; gi -= nwa) {
    const int g = gi < n1 ? g_lo + gi : g2_lo + (gi - n1);
    // poll the group's entries until none is EMPTY (all ones: the forward's first kernel resets the exchange words;
    // a published (sum, M2) is finite).  All tiles of the utterance are resident and arrive within the spread of the
    // workgroups' k-loops; a lane re-reads only what it has not seen yet; bounded and flagged, never a hang
    constexpr int EPL = 4;                         // entries per lane: up to 256 per group (gemm_gnx_plan checks)
    unsigned long long w[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) w[k] = ~0ull;
    const int ne = RB * nvb;
    for (int spins = 0;; ++spins) {
      bool ok = true;
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        const int e = lane + 64 * k;
        if (e < ne && w[k] == ~0ull) {
          const int rb = e / nvb, cb = g * nvb + (e - rb * nvb);     // 16-channel block of the concatenation
          if (cb < ncb)
            w[k] = __hip_atomic_load(gx.xchg + (size_t)(bq * RB + rb) * ncb + cb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          else {                                   // the skip's block: in memory since an earlier launch
            const float2 sv = reinterpret_cast<const float2*>(gx.sk_stat16)[(size_t)(bq * RB + rb) * ncs + (cb - ncb)];
            w[k] = (unsigned long long)__float_as_uint(sv.x) | ((unsigned long long)__float_as_uint(sv.y) << 32);
          }
          ok = ok && w[k] != ~0ull;
        }
      }
      if (__all(ok)) break;
      // (another launch has already given up: the run is lost and will be repeated on the fallback schedule - do not
      // spend ~0.4 s per GEMM waiting for partners that a foreign kernel keeps off the CUs)
      const bool lost = (spins & 63) == 63 && __hip_atomic_load(gx.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
      if (lost) break;
      if (spins > gx.spin_max) {
        if (lane == 0) {               // which GEMM, which workgroup, which group: reported by the next host call
          gx.status[1] = (unsigned)(size_t)gx.xchg; gx.status[2] = blockIdx.x; gx.status[3] = (unsigned)g;
          gx.status[4] = (unsigned)__builtin_popcountll(__ballot(!ok));
          __hip_atomic_store(gx.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    double s1 = 0.0, q = 0.0;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      if (lane + 64 * k < ne) {
        const double sx = (double)__uint_as_float((unsigned)w[k]), m2 = (double)__uint_as_float((unsigned)(w[k] >> 32));
        s1 += sx;
        const int e = lane + 64 * k, cnt = min(32, t.Tv_out - 32 * (e / nvb));   // frames of row block e / nvb that exist
        q += m2 + sx * sx / (double)(16 * cnt);    // = the block's sum of squares
      }
    }
    s1 = wave_sum64(s1); q = wave_sum64(q);
    if (lane == 0) {
      const double n = (double)cpg * (double)t.Tv_out, mean = s1 / n;
      double var = q / n - mean * mean;
      var = var > 0 ? var : 0;
      sh.gst[g - g_lo] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)gx.eps)));
    }
  }
  trace(20);                                    // this wave's groups are reduced
  __syncthreads();
  trace(21);                                    // ... every wave's
  if (tid < t.bn) {
    const float2 st = sh.gst[min(t.n0 + tid, t.N - 1) / cpg - g_lo];
    const float a = st.y * pg;
    sh.gA[tid] = a * pts;
    sh.gB[tid] = fmaf(pb - st.x * a, pts, ptb);
  }
  if (tid < sw) {
    const float2 st = sh.gst[(t.N + sb0 * 16 + tid) / cpg - g_lo];
    const float a = st.y * pg2;
    sh.gA2[tid] = a;
    sh.gB2[tid] = pb2 - st.x * a;
  }
  __syncthreads();
  if (sw > 0) {
    for (int i = tid, k = 0; i < total; i += nthr, ++k) {
      int cl;
      const size_t o = item(i, cl);
      if (o == ~(size_t)0) continue;
      float4 v0, v1;
      if (k < PF) { v0 = k == 0 ? pv[0][0] : pv[1][0]; v1 = k == 0 ? pv[0][1] : pv[1][1]; }
      else { v0 = *reinterpret_cast<const float4*>(gx.sk_x + o); v1 = *reinterpret_cast<const float4*>(gx.sk_x + o + 4); }
      const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        y[e] = fmaf(v[e], sh.gA2[cl + e], sh.gB2[cl + e]);
        if (gx.silu) y[e] = y[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-y[e]));
      }
      auto put8 = [&](bf16_t* hi, bf16_t* lo, const float* x) {
        uint4 h, l;
        h.x = dv_cvt_pk_bf16(x[0], x[1]); h.y = dv_cvt_pk_bf16(x[2], x[3]); h.z = dv_cvt_pk_bf16(x[4], x[5]); h.w = dv_cvt_pk_bf16(x[6], x[7]);
        dv_st16(hi + o, h);
        if (lo) {
          l.x = dv_cvt_pk_bf16(x[0] - __uint_as_float(h.x << 16), x[1] - __uint_as_float(h.x & 0xffff0000u));
          l.y = dv_cvt_pk_bf16(x[2] - __uint_as_float(h.y << 16), x[3] - __uint_as_float(h.y & 0xffff0000u));
          l.z = dv_cvt_pk_bf16(x[4] - __uint_as_float(h.z << 16), x[5] - __uint_as_float(h.z & 0xffff0000u));
          l.w = dv_cvt_pk_bf16(x[6] - __uint_as_float(h.w << 16), x[7] - __uint_as_float(h.w & 0xffff0000u));
          dv_st16(lo + o, l);
        }
      };
      put8(gx.sk_y_hi, gx.sk_y_lo, y);
      if (gx.sk_raw_hi) put8(gx.sk_raw_hi, gx.sk_raw_lo, v);
    }
  }
}
