// Bandwidth-bound helper kernels of the denoiser for gfx950: layout packing, GroupNorm /
// LayerNorm statistics, the small conditioning GEMVs, weight packing (fp32 -> split bf16)
// and the sampler's fused linear-combination update.  All are wave64 kernels with 16-byte
// accesses where the layout allows.
#include "dv_common.h"
#include "misc_body.h"

#include <cstdlib>

// ---------------------------------------------------------------------------------------
// (B, C, T) channels-first inputs x | cond  ->  (B*T, cpad) channels-last, zero padded.
// The boundary tensors of UNet1DConditionModel.forward are channels-first
// (reference unet_1d_condition.py:943 conv_in on [B, C+128, T]); everything inside the
// engine is channels-last.  32x32 LDS-transposed tiles: reads coalesced along T, writes
// along C.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_input(const float* __restrict__ x, int cx,
                                                     const float* __restrict__ cond, int cc,
                                                     bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                     int cpad, int T, int Tp, unsigned long long* __restrict__ reset, size_t reset_words) {
  // (this kernel precedes every GEMM of a forward: the exchange words of the in-epilogue GroupNorms - GnxParams - start EMPTY)
  {
    const size_t nthr = (size_t)gridDim.x * gridDim.y * gridDim.z * blockDim.x;
    const size_t me = ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    for (size_t k = me * 2; k < reset_words; k += nthr * 2) *reinterpret_cast<ulonglong2*>(reset + k) = make_ulonglong2(~0ull, ~0ull);
  }
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // ty 0..7
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + i * 8, t = t0 + tx;
    float v = 0.f;
    if (t < T) {
      if (c < cx) v = x[((size_t)b * cx + c) * T + t];
      else if (c < cx + cc) v = cond[((size_t)b * cc + (c - cx)) * T + t];
    }
    tile[ty + i * 8][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + i * 8, c = c0 + tx;
    if (t < Tp && c < cpad) {                           // (rows [T, Tp): zeros)
      bf16_t h, l;
      split1(tile[tx][ty + i * 8], h, l);
      const size_t o = ((size_t)b * Tp + t) * cpad + c;
      out_hi[o] = h;
      if (out_lo) out_lo[o] = l;
    }
  }
}

hipError_t launch_pack_input(const float* x, int cx, const float* cond, int cc, bf16_t* out_hi, bf16_t* out_lo, int cpad,
                             int B, int T, hipStream_t st, unsigned long long* reset, size_t reset_words, int Tp) {
  if (reset_words & 1) return hipErrorInvalidValue;
  if (Tp <= 0) Tp = T;
  if (Tp < T) return hipErrorInvalidValue;
  dim3 grid((Tp + 31) / 32, (cpad + 31) / 32, B);
  hipLaunchKernelGGL(k_pack_input, grid, dim3(256), 0, st, x, cx, cond, cc, out_hi, out_lo, cpad, T, Tp, reset, reset ? reset_words : (size_t)0);
  return hipGetLastError();
}

__global__ void k_split(const float* __restrict__ in, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int64_t n4) {
  split_body<false>(in, hi, lo, 0, n4, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}
hipError_t launch_split(const float* in, bf16_t* hi, bf16_t* lo, int64_t n, hipStream_t st) {
  if (n % 4 != 0) return hipErrorInvalidValue;
  const int64_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_split, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, in, hi, lo, n4);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// GroupNorm apply (reference F.group_norm + temb scale/shift + SiLU: resnet.py:594-631,
// transformer_1d.py:257, unet_1d_condition.py:1030-1031) -> split bf16 planes for the consumer GEMM.
// Each workgroup handles a run of frames of one batch item: (1) it derives the per-channel affine of
// that batch item - from the producers' per-32-row-block column sums (fp64 reduction over row blocks
// and over the group's channels) or from a precomputed table - into LDS; (2) it streams its rows:
// y = act(x*scale + shift) -> hi/lo planes, 16-byte loads, 8-byte stores.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gn_apply(const GnApplyParams p, int rows_per_block) {
  gn_apply_body<256, false>(p, rows_per_block, blockIdx.x, blockIdx.y, blockIdx.z);
}

hipError_t launch_gn_apply(const GnApplyParams& p, hipStream_t st) {
  const int ctot = p.c0 + p.c1;
  if (ctot % p.groups != 0) return hipErrorInvalidValue;
  const int cg = ctot / p.groups;
  if (cg % 4 != 0 || p.c0 % 4 != 0 || cg > 512 || p.groups > 64) return hipErrorInvalidValue;
  // slab statistics: 32-row blocks must not span utterances (T % 32 == 0, or a single utterance whose last block is partial)
  if (p.Tv < 0 || p.Tv > p.T) return hipErrorInvalidValue;
  if (!p.scale_in && p.st16_0) {
    if (p.T % 32 != 0 || cg % 16 != 0 || p.c0 % 16 != 0 || (p.c1 && !p.st16_1) || cg > 512) return hipErrorInvalidValue;
    if (p.Tv && p.Tv <= p.T - 32) return hipErrorInvalidValue;   // (every 32-row block holds at least one frame that exists)
  } else if (p.Tv && p.Tv != p.T) return hipErrorInvalidValue;     // (padded row spaces carry block statistics)
  else if (!p.scale_in && ((p.T % 32 != 0 && p.B != 1) || !p.slab0 || (p.c1 && !p.slab1))) return hipErrorInvalidValue;
  // ~1024 workgroups: (frame chunks) x groups x batch  (DVITS_GN_WGS: experiment knob)
  static const int target = [] { const char* e = getenv("DVITS_GN_WGS"); return e ? atoi(e) : 1024; }();
  int chunks = target / (p.groups * p.B);
  chunks = chunks < 1 ? 1 : chunks;
  int rpb = (p.T + chunks - 1) / chunks;
  rpb = rpb < 4 ? 4 : rpb;
  hipLaunchKernelGGL(k_gn_apply, dim3((p.T + rpb - 1) / rpb, p.groups, p.B), dim3(256), 0, st, p, rpb);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// GroupNorm statistics (reference F.group_norm calls: resnet.py:594,621; transformer_1d.py:257;
// unet_1d_condition.py:1030) over the channel-concat [a0 | a1] of channels-last tensors.
// Stage 1: per (batch, frame-chunk, group) partial sum / sum of squares, accumulated in
// fp32 per thread (<= 64 values) and combined in fp64.  Stage 2 (k_gn_finalize) reduces the
// chunks in fp64 and emits the per-(batch, channel) affine the consumer GEMM applies:
//   y = x*scale + shift,  scale = rstd*gamma*(1+ts),  shift = (beta - mean*rstd*gamma)*(1+ts) + tb
// where ts/tb are the resnet's timestep scale/shift (resnet.py:627-629) or absent.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gn_partial(const float* __restrict__ a0, int c0,
                                                     const float* __restrict__ a1, int c1,
                                                     double* __restrict__ part, int T, int G, int nchunk,
                                                     int rows_per_chunk) {
  __shared__ double s_sum[256], s_sq[256];
  __shared__ int s_grp[256];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int ctot = c0 + c1, ncol4 = ctot >> 2, cg = ctot / G;
  const int tid = threadIdx.x;
  const int t0 = chunk * rows_per_chunk, t1 = min(T, t0 + rows_per_chunk);
  // thread -> (row lane, float4 column); columns beyond 256 are looped
  const int ncol_thr = min(ncol4, 256);
  const int nrl = 256 / ncol_thr;
  const int rl = tid / ncol_thr, j0 = tid % ncol_thr;
  double gs[4] = {0, 0, 0, 0}, gq[4] = {0, 0, 0, 0};   // up to 4 column slots per thread (ctot <= 4096)
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    const int j = j0 + sl * ncol_thr;
    if (rl < nrl && j < ncol4) {
      const int c = j * 4;
      const float* src = (c < c0) ? a0 : a1;
      const int ld = (c < c0) ? c0 : c1;
      const int cc = (c < c0) ? c : c - c0;
      float s = 0.f, q = 0.f;
      for (int t = t0 + rl; t < t1; t += nrl) {
        const float4 v = *reinterpret_cast<const float4*>(src + ((size_t)b * T + t) * ld + cc);
        s += (v.x + v.y) + (v.z + v.w);
        q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
      gs[sl] = s;
      gq[sl] = q;
    }
  }
  // one LDS pass per column slot keeps the reduction order fixed (deterministic)
  double* out = part + (((size_t)b * nchunk + chunk) * G) * 2;
  double acc_s = 0, acc_q = 0;   // used by threads g < G
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    if (sl * ncol_thr >= ncol4) break;
    const int j = j0 + sl * ncol_thr;
    const bool live = rl < nrl && j < ncol4;
    s_sum[tid] = live ? gs[sl] : 0.0;
    s_sq[tid] = live ? gq[sl] : 0.0;
    s_grp[tid] = live ? (j * 4) / cg : -1;
    __syncthreads();
    if (tid < G) {
      for (int i = 0; i < 256; ++i)
        if (s_grp[i] == tid) { acc_s += s_sum[i]; acc_q += s_sq[i]; }
    }
    __syncthreads();
  }
  if (tid < G) { out[tid * 2] = acc_s; out[tid * 2 + 1] = acc_q; }
}

hipError_t launch_gn_partial(const float* a0, int c0, const float* a1, int c1, double* part, int B, int T, int G,
                             int nchunk, hipStream_t st) {
  const int rows = (T + nchunk - 1) / nchunk;
  hipLaunchKernelGGL(k_gn_partial, dim3(nchunk, B), dim3(256), 0, st, a0, c0, a1, c1, part, T, G, nchunk, rows);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_gn_finalize(const double* __restrict__ part, int nchunk,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      const float* __restrict__ tscale,
                                                      const float* __restrict__ tshift, int ld_t,
                                                      float* __restrict__ scale, float* __restrict__ shift,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                      int T, int C, int G, float eps) {
  __shared__ float s_mean[64], s_rstd[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int cg = C / G;
  if (tid < G) {
    double s = 0, q = 0;
    for (int ch = 0; ch < nchunk; ++ch) {
      const double* p = part + (((size_t)b * nchunk + ch) * G + tid) * 2;
      s += p[0];
      q += p[1];
    }
    const double n = (double)cg * (double)T;
    const double mean = s / n;
    double var = q / n - mean * mean;
    var = var > 0 ? var : 0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    s_mean[tid] = (float)mean;
    s_rstd[tid] = (float)rstd;
    if (mean_out) { mean_out[b * G + tid] = (float)mean; rstd_out[b * G + tid] = (float)rstd; }
  }
  __syncthreads();
  if (scale) {
    for (int c = tid; c < C; c += 256) {
      const int g = c / cg;
      const float a = s_rstd[g] * gamma[c];
      const float sh = beta[c] - s_mean[g] * a;
      const float ts = tscale ? 1.0f + tscale[(size_t)b * ld_t + c] : 1.0f;
      const float tb = tshift ? tshift[(size_t)b * ld_t + c] : 0.0f;
      scale[(size_t)b * C + c] = a * ts;
      shift[(size_t)b * C + c] = fmaf(sh, ts, tb);
    }
  }
}

hipError_t launch_gn_finalize(const double* part, int nchunk, const float* gamma, const float* beta,
                              const float* tscale, const float* tshift, int ld_t, float* scale, float* shift,
                              float* mean_out, float* rstd_out, int B, int T, int C, int G, float eps,
                              hipStream_t st) {
  hipLaunchKernelGGL(k_gn_finalize, dim3(B), dim3(256), 0, st, part, nchunk, gamma, beta, tscale, tshift, ld_t,
                     scale, shift, mean_out, rstd_out, T, C, G, eps);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// LayerNorm row statistics (reference attention.py:157,176,189 nn.LayerNorm eps 1e-5):
// one wave per row, two-pass (mean, then centred variance) on register-resident values.
// gamma/beta are folded into the consumer GEMM's weights/bias at pack time.
// ---------------------------------------------------------------------------------------
template <bool WRITE_NORM>
__global__ __launch_bounds__(256) void k_ln_rows(const float* __restrict__ x, float* __restrict__ mean,
                                                  float* __restrict__ rstd, const float* __restrict__ g,
                                                  const float* __restrict__ bta, float* __restrict__ out,
                                                  int M, int C, float eps, int rows_per_batch, int out_rows_per_batch,
                                                  int out_row_off) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (size_t)row * C;
  float4 v[8];
  const int n4 = C >> 2;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + i * 64;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < n4) {
      v[i] = *reinterpret_cast<const float4*>(xr + j * 4);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + i * 64;
    if (j < n4) {
      const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  if (!WRITE_NORM) {
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  } else {
    const int bb = row / rows_per_batch, rr = row - bb * rows_per_batch;
    float* orow = out + ((size_t)bb * out_rows_per_batch + out_row_off + rr) * C;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = lane + i * 64;
      if (j < n4) {
        const float4 gg = *reinterpret_cast<const float4*>(g + j * 4);
        const float4 bb4 = *reinterpret_cast<const float4*>(bta + j * 4);
        float4 o;
        o.x = (v[i].x - mu) * rs * gg.x + bb4.x;
        o.y = (v[i].y - mu) * rs * gg.y + bb4.y;
        o.z = (v[i].z - mu) * rs * gg.z + bb4.z;
        o.w = (v[i].w - mu) * rs * gg.w + bb4.w;
        *reinterpret_cast<float4*>(orow + j * 4) = o;
      }
    }
  }
}

hipError_t launch_ln_stats(const float* x, float* mean, float* rstd, int M, int C, float eps, hipStream_t st) {
  if (C % 4 != 0 || C > 2048) return hipErrorInvalidValue;
  hipLaunchKernelGGL((k_ln_rows<false>), dim3((M + 3) / 4), dim3(256), 0, st, x, mean, rstd, nullptr, nullptr,
                     nullptr, M, C, eps, 1, 1, 0);
  return hipGetLastError();
}

// out row mapping: row (b, r) of x -> out row b*out_rows_per_batch + out_row_off + r
static hipError_t launch_layernorm_rows_ex(const float* x, const float* g, const float* b, float* out, int M, int C,
                                           float eps, int rows_per_batch, int out_rows_per_batch, int out_row_off,
                                           hipStream_t st) {
  if (C % 4 != 0 || C > 2048) return hipErrorInvalidValue;
  hipLaunchKernelGGL((k_ln_rows<true>), dim3((M + 3) / 4), dim3(256), 0, st, x, nullptr, nullptr, g, b, out, M, C,
                     eps, rows_per_batch, out_rows_per_batch, out_row_off);
  return hipGetLastError();
}

hipError_t launch_layernorm_rows(const float* x, const float* g, const float* b, float* out, int M, int C,
                                 float eps, hipStream_t st) {
  return launch_layernorm_rows_ex(x, g, b, out, M, C, eps, M, M, 0, st);
}

// LayerNorm rows -> split planes (reference attention.py:157,176,189; gamma/beta live in the consumer
// GEMM's packed weights/bias): one wave per row, two-pass statistics on register-resident values.
__global__ __launch_bounds__(256) void k_ln_apply(const float* __restrict__ x, bf16_t* __restrict__ hi,
                                                   bf16_t* __restrict__ lo, int M, int C, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (size_t)row * C;
  float4 v[8];
  const int n4 = C >> 2;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + i * 64;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < n4) {
      v[i] = *reinterpret_cast<const float4*>(xr + j * 4);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + i * 64;
    if (j < n4) {
      const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + i * 64;
    if (j < n4) {
      float4 y;
      y.x = (v[i].x - mu) * rs; y.y = (v[i].y - mu) * rs; y.z = (v[i].z - mu) * rs; y.w = (v[i].w - mu) * rs;
      uint2 h, l;
      split4(y, h, l);
      const size_t o = ((size_t)row * C >> 2) + j;
      reinterpret_cast<uint2*>(hi)[o] = h;
      if (lo) reinterpret_cast<uint2*>(lo)[o] = l;
    }
  }
}

hipError_t launch_ln_apply(const float* x, bf16_t* hi, bf16_t* lo, int M, int C, float eps, hipStream_t st) {
  if (C % 4 != 0 || C > 2048) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_ln_apply, dim3((M + 3) / 4), dim3(256), 0, st, x, hi, lo, M, C, eps);
  return hipGetLastError();
}

// LayerNorm rows with affine (+ per-row keep mask) -> fp32 and/or split planes.  Prompt encoder: the input of the
// k = 9 feed-forward (its zero padding must see gamma/beta applied, so they cannot be folded into the weights;
// reference operations.py:816-817, 676-681) and the final LayerNorm (model3.py:428-430).  One wave per row.
__global__ __launch_bounds__(256) void k_ln_affine(const float* __restrict__ x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ rowmask,
                                                    float* __restrict__ out, bf16_t* __restrict__ hi,
                                                    bf16_t* __restrict__ lo, int M, int C, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (size_t)row * C;
  float4 v[8];
  const int n4 = C >> 2;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + i * 64;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < n4) {
      v[i] = *reinterpret_cast<const float4*>(xr + j * 4);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + i * 64;
    if (j < n4) {
      const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  const float rm = rowmask ? rowmask[row] : 1.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = lane + i * 64;
    if (j < n4) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + j * 4);
      const float4 b = *reinterpret_cast<const float4*>(beta + j * 4);
      float4 y;
      y.x = fmaf((v[i].x - mu) * rs, g.x, b.x) * rm; y.y = fmaf((v[i].y - mu) * rs, g.y, b.y) * rm;
      y.z = fmaf((v[i].z - mu) * rs, g.z, b.z) * rm; y.w = fmaf((v[i].w - mu) * rs, g.w, b.w) * rm;
      const size_t o = ((size_t)row * C >> 2) + j;
      if (out) reinterpret_cast<float4*>(out)[o] = y;
      if (hi) {
        uint2 h, l;
        split4(y, h, l);
        reinterpret_cast<uint2*>(hi)[o] = h;
        if (lo) reinterpret_cast<uint2*>(lo)[o] = l;
      }
    }
  }
}

hipError_t launch_ln_affine(const float* x, const float* gamma, const float* beta, const float* rowmask, float* out,
                            bf16_t* hi, bf16_t* lo, int M, int C, float eps, hipStream_t st) {
  if (C % 4 != 0 || C > 2048 || !gamma || !beta) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_ln_affine, dim3((M + 3) / 4), dim3(256), 0, st, x, gamma, beta, rowmask, out, hi, lo, M, C, eps);
  return hipGetLastError();
}

// Prompt-encoder input stage (reference model3.py:409-420 + model.py:164-169): one wave per frame (b, t) gathers
// its C channels from the channels-first prompt, zeroes padding frames, LayerNorm(C) with affine, writes split
// planes [B*L, cpad] (channels C..cpad-1 zero) for the k = 1 `pre` contraction, and the additive key bias of the
// frame for the attention kernels (0 / -1e30: padded keys get exactly zero weight, operations.py:405-416).
__global__ __launch_bounds__(256) void k_prompt_pre(const float* __restrict__ prompt, const float* __restrict__ keep,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                                                     float* __restrict__ key_bias, int B, int C, int L, int cpad, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= B * L) return;
  const int b = row / L, t = row - b * L;
  const float kp = keep[row];
  float v[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + i * 64;
    v[i] = (c < C && kp != 0.f) ? prompt[((size_t)b * C + c) * L + t] : 0.f;
    s += v[i];
  }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + i * 64;
    if (c < C) q += (v[i] - mu) * (v[i] - mu);
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + i * 64;
    if (c < cpad) {
      const float y = c < C ? fmaf((v[i] - mu) * rs, gamma[c], beta[c]) : 0.f;
      const unsigned hb = pk_bf16(y, 0.f);
      hi[(size_t)row * cpad + c] = (bf16_t)(hb & 0xffffu);
      if (lo) lo[(size_t)row * cpad + c] = (bf16_t)(pk_bf16(y - __uint_as_float(hb << 16), 0.f) & 0xffffu);
    }
  }
  if (lane == 0) key_bias[row] = kp != 0.f ? 0.f : -1e30f;
}

hipError_t launch_prompt_pre(const float* prompt, const float* keep, const float* gamma, const float* beta, bf16_t* hi,
                             bf16_t* lo, float* key_bias, int B, int C, int L, int cpad, float eps, hipStream_t st) {
  if (C > 512 || cpad > 512 || cpad < C) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_prompt_pre, dim3((B * L + 3) / 4), dim3(256), 0, st, prompt, keep, gamma, beta, hi, lo, key_bias, B, C,
                     L, cpad, eps);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Small-M fp32 linear (the conditioning path: TimestepEmbedding MLP, the 22 batched
// time_emb_proj GEMVs, the pooled-text projection; reference embeddings.py:186-201,
// resnet.py:615-617).  One wave per output column n, all M (<= 64) rows: W row read once,
// coalesced; exact fp32 FMA dot + wave reduction.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_small_linear(const float* __restrict__ in, int ldin,
                                                       const float* __restrict__ W, const float* __restrict__ bias,
                                                       const float* __restrict__ add, float* __restrict__ out,
                                                       int ldo, int M, int K, int N, int silu_in, int silu_out,
                                                       int cols_per_wave) {
  extern __shared__ float s_in[];        // [M][K] activated input rows, staged once per workgroup
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < M * K; i += 256) {
    const int m = i / K, k = i - m * K;
    float xv = in[(size_t)m * ldin + k];
    if (silu_in) xv = xv / (1.0f + __expf(-xv));
    s_in[i] = xv;
  }
  __syncthreads();
  const int n_base = (blockIdx.x * 4 + wave) * cols_per_wave;
  for (int cw = 0; cw < cols_per_wave; ++cw) {
    const int n = n_base + cw;
    if (n >= N) return;
    const float* w = W + (size_t)n * K;
    for (int m0 = 0; m0 < M; m0 += 8) {
      float acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = 0.f;
      for (int k = lane; k < K; k += 64) {
        const float wv = w[k];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (m0 + i < M) acc[i] = fmaf(s_in[(m0 + i) * K + k], wv, acc[i]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float sum = wave_sum(acc[i]);
        if (lane == 0 && m0 + i < M) {
          float v = sum + (bias ? bias[n] : 0.f);
          if (silu_out) v = v / (1.0f + __expf(-v));
          if (add) v += add[(size_t)(m0 + i) * ldo + n];
          out[(size_t)(m0 + i) * ldo + n] = v;
        }
      }
    }
  }
}

// Wide small-M fp32 linear on TRANSPOSED weights Wt [K, N] (the 22 batched time_emb_proj GEMVs: [B, 512] x [512, 14848]):
// lane = output column, so the weight stream is read once, fully coalesced, and no cross-lane reduction is needed;
// the (activated) input rows sit in LDS and are broadcast.  M <= 16.
template <int MR>
__global__ __launch_bounds__(512) void k_small_linear_t(const float* __restrict__ in, int ldin, const float* __restrict__ Wt,
                                                         const float* __restrict__ bias, const float* __restrict__ add,
                                                         float* __restrict__ out, int ldo, int M, int K, int N, int silu_in,
                                                         int silu_out, int add_rows) {
  // workgroup = 64 output columns x 8 k-eighths (one wave each): 232 workgroups for N = 14848, 64 sequential
  // k-steps per lane with 16 weight loads in flight (the 30 MB table is streamed every step: bytes in flight per CU
  // are what sets the rate - four waves per workgroup reached 1.5 TB/s); the eight partial sums meet in LDS
  extern __shared__ float s_in[];        // [K][MR] activated input, row index fastest; then [8][64][MR] partials
  float* s_part = s_in + K * MR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // blockIdx.y: chunk of MR rows (a batch of evaluations: every row's sum runs in the same order as in a one-chunk launch)
  const int row0 = blockIdx.y * MR;
  in += (size_t)row0 * ldin; out += (size_t)row0 * ldo;
  M = min(MR, M - row0);
  for (int i = tid; i < K * MR; i += 512) {
    const int k = i / MR, m = i - k * MR;
    float xv = m < M ? in[(size_t)m * ldin + k] : 0.f;
    if (silu_in) xv = xv / (1.0f + __expf(-xv));
    s_in[i] = xv;
  }
  __syncthreads();
  const int n = blockIdx.x * 64 + lane;
  const int nc = n < N ? n : N - 1;
  float acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = 0.f;
  const int kq = (K + 7) / 8, k0 = wave * kq, k1 = min(K, k0 + kq);
  const float* w = Wt + nc;
#pragma unroll 16
  for (int k = k0; k < k1; ++k) {
    const float wv = w[(size_t)k * N];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = fmaf(s_in[k * MR + m], wv, acc[m]);
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) s_part[(wave * 64 + lane) * MR + m] = acc[m];
  __syncthreads();
  if (wave == 0 && n < N) {
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += s_part[(w * 64 + lane) * MR + m];
      if (m < M) {
        v += bv;
        if (silu_out) v = v / (1.0f + __expf(-v));
        if (add) v += add[(size_t)(add_rows > 0 ? (row0 + m) % add_rows : row0 + m) * ldo + n];
        out[(size_t)m * ldo + n] = v;
      }
    }
  }
}

hipError_t launch_small_linear_t(const float* in, int ldin, const float* Wt, const float* b, const float* add, float* out,
                                 int ldo, int M, int K, int N, int silu_in, int silu_out, hipStream_t st, int add_rows) {
  if (((size_t)K * 16 + 512 * 16) * sizeof(float) > 64 * 1024) return hipErrorInvalidValue;
  // more than 16 rows (a batch of evaluations): chunks of 8 rows over blockIdx.y - per row the same arithmetic, bit for bit
  const dim3 grid((N + 63) / 64, M > 16 ? (M + 7) / 8 : 1);
  if (M <= 8 || M > 16) hipLaunchKernelGGL(k_small_linear_t<8>, grid, dim3(512), ((size_t)K * 8 + 512 * 8) * 4, st, in, ldin, Wt, b, add, out, ldo, M, K, N, silu_in, silu_out, add_rows);
  else hipLaunchKernelGGL(k_small_linear_t<16>, grid, dim3(512), ((size_t)K * 16 + 512 * 16) * 4, st, in, ldin, Wt, b, add, out, ldo, M, K, N, silu_in, silu_out, add_rows);
  return hipGetLastError();
}

// dst[c, r] = src[r, c]  (one-off weight re-layout at prepare time)
__global__ __launch_bounds__(256) void k_transpose_f32(const float* __restrict__ src, float* __restrict__ dst, int R, int Cc) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)R * Cc) return;
  const int r = (int)(i / Cc), c = (int)(i - (int64_t)r * Cc);
  dst[(size_t)c * R + r] = src[i];
}
hipError_t launch_transpose_f32(const float* src, float* dst, int R, int Cc, hipStream_t st) {
  hipLaunchKernelGGL(k_transpose_f32, dim3((unsigned)(((int64_t)R * Cc + 255) / 256)), dim3(256), 0, st, src, dst, R, Cc);
  return hipGetLastError();
}

hipError_t launch_small_linear(const float* in, int ldin, const float* W, const float* b, const float* add,
                               float* out, int ldo, int M, int K, int N, int silu_in, int silu_out,
                               hipStream_t st) {
  const size_t smem = (size_t)M * K * sizeof(float);
  if (smem > 64 * 1024) return hipErrorInvalidValue;
  // ~1024 workgroups at most: wide outputs (the batched time_emb_proj) take several columns per wave
  int cpw = (N + 4095) / 4096;
  cpw = cpw < 1 ? 1 : cpw;
  hipLaunchKernelGGL(k_small_linear, dim3((N + 4 * cpw - 1) / (4 * cpw)), dim3(256), smem, st, in, ldin, W, b, add,
                     out, ldo, M, K, N, silu_in, silu_out, cpw);
  return hipGetLastError();
}

// Sinusoidal timestep embedding [cos | sin] (reference embeddings.py:24-64 with
// flip_sin_to_cos=True, freq_shift=0): same fp32 operation order as the reference.
__global__ void k_timestep_sincos(const float* __restrict__ t, float* __restrict__ out, int B, int dim,
                                  unsigned long long* __restrict__ reset, size_t reset_words) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  (void)reset; (void)reset_words;
  const int half = dim >> 1;
  if (i >= B * half) return;
  const int b = i / half, j = i - b * half;
  float e = -9.210340371976184f * (float)j;   // -ln(10000) * arange, fp32
  e = e / (float)half;
  const float arg = t[b] * expf(e);
  out[(size_t)b * dim + j] = cosf(arg);
  out[(size_t)b * dim + half + j] = sinf(arg);
}

hipError_t launch_timestep_sincos(const float* t, float* out, int B, int dim, hipStream_t st, unsigned long long* reset,
                                  size_t reset_words) {
  const int n = B * (dim / 2);
  if (reset_words & 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_timestep_sincos, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, t, out, B, dim, reset, reset ? reset_words : 0);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Attention pooling of the text/prompt states (reference embeddings.py:499-546): one query
// (the mean token) against L+1 keys, `heads` heads of D/heads dims.  Step-invariant: runs
// once per set_cond.  seq = [cls | LN(enc)] rows; kv = seq @ [Wk;Wv]^T computed by the
// GEMM kernel; this kernel does the per-(batch, head) softmax-weighted sum.
// ---------------------------------------------------------------------------------------
__global__ void k_mean_token(const float* __restrict__ seq, const float* __restrict__ pos, float* __restrict__ seq_out,
                             int L, int D) {
  // seq rows 1..L of batch b hold LN(enc); write row 0 = mean over rows + pos
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float s = 0.f;
    for (int r = 1; r <= L; ++r) s += seq[((size_t)b * (L + 1) + r) * D + c];
    seq_out[((size_t)b * (L + 1)) * D + c] = s / (float)L + pos[c];
  }
}

__global__ __launch_bounds__(64) void k_pool_attn(const float* __restrict__ q, const float* __restrict__ kv,
                                                   float* __restrict__ out, int S, int D, int heads) {
  // q [B, D]; kv [B*S, 2D] (k | v); out [B, D]
  const int b = blockIdx.y, h = blockIdx.x, lane = threadIdx.x;
  const int dph = D / heads;
  const float scale = 1.0f / sqrtf(sqrtf((float)dph));
  float mx = -1e30f;
  for (int s = lane; s < S; s += 64) {
    float sc = 0.f;
    for (int c = 0; c < dph; ++c)
      sc += (q[(size_t)b * D + h * dph + c] * scale) * (kv[((size_t)b * S + s) * 2 * D + h * dph + c] * scale);
    mx = fmaxf(mx, sc);
  }
  mx = wave_max(mx);
  float den = 0.f;
  float accv[8];
  for (int c = 0; c < 8; ++c) accv[c] = 0.f;
  for (int s = lane; s < S; s += 64) {
    float sc = 0.f;
    for (int c = 0; c < dph; ++c)
      sc += (q[(size_t)b * D + h * dph + c] * scale) * (kv[((size_t)b * S + s) * 2 * D + h * dph + c] * scale);
    const float w = expf(sc - mx);
    den += w;
    for (int c = 0; c < dph && c < 8; ++c) accv[c] += w * kv[((size_t)b * S + s) * 2 * D + D + h * dph + c];
  }
  den = wave_sum(den);
  for (int c = 0; c < dph && c < 8; ++c) {
    const float a = wave_sum(accv[c]);
    if (lane == 0) out[(size_t)b * D + h * dph + c] = a / den;
  }
}

hipError_t launch_mean_token(float* seq, const float* pos, int B, int L, int D, hipStream_t st) {
  hipLaunchKernelGGL(k_mean_token, dim3(B), dim3(128), 0, st, seq, pos, seq, L, D);
  return hipGetLastError();
}
hipError_t launch_pool_attn(const float* q, const float* kv, float* out, int B, int S, int D, int heads,
                            hipStream_t st) {
  if (D / heads > 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_pool_attn, dim3(heads, B), dim3(64), 0, st, q, kv, out, S, D, heads);
  return hipGetLastError();
}
hipError_t launch_layernorm_rows_into(const float* x, const float* g, const float* b, float* out, int M, int C,
                                      float eps, int rows_per_batch, int out_rows_per_batch, int out_row_off,
                                      hipStream_t st) {
  return launch_layernorm_rows_ex(x, g, b, out, M, C, eps, rows_per_batch, out_rows_per_batch, out_row_off, st);
}

// ---------------------------------------------------------------------------------------
// Weight packing: fp32 state-dict tensors -> bf16 hi / lo operand matrices [N_pad, Kp]
// (k contiguous) in the K order the implicit GEMM walks: segment, tap, channel.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bf16_t f2bf_rne(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

__global__ void k_pack_weight(const PackSpec s, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int Kp) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = s.C * s.taps;
  if (idx >= (size_t)s.N * per_row) return;
  const int n = (int)(idx / per_row), rem = (int)(idx - (size_t)n * per_row);
  int c, tap;
  if (s.kind == 1) { c = rem / s.taps; tap = rem - c * s.taps; }   // src [N, C, taps]
  else { c = rem; tap = 0; }
  float w = s.src[idx];
  if (s.kscale) w *= s.kscale[c];
  int row = n;
  if (s.geglu) {
    const int half = s.N >> 1;
    const int j = n < half ? n : n - half;
    row = (j >> 5) * 64 + (n < half ? 0 : 32) + (j & 31);
  }
  const size_t o = (size_t)(s.n_off + row) * Kp + s.k_off + (size_t)tap * s.c_pad + c;
  const bf16_t h = f2bf_rne(w);
  hi[o] = h;
  if (lo) lo[o] = f2bf_rne(w - __uint_as_float((unsigned)h << 16));
}

hipError_t launch_pack_weight(const PackSpec& s, bf16_t* hi, bf16_t* lo, int Kp, hipStream_t st) {
  const size_t n = (size_t)s.N * s.C * s.taps;
  hipLaunchKernelGGL(k_pack_weight, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s, hi, lo, Kp);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_fold_bias(const float* __restrict__ W, const float* __restrict__ bias,
                                                    const float* __restrict__ beta, float* __restrict__ out, int N,
                                                    int C, int n_off, int geglu) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  double s = 0;
  if (beta)
    for (int c = lane; c < C; c += 64) s += (double)W[(size_t)n * C + c] * (double)beta[c];
  s = wave_sum_d(s);
  if (lane == 0) {
    int row = n;
    if (geglu) {
      const int half = N >> 1;
      const int j = n < half ? n : n - half;
      row = (j >> 5) * 64 + (n < half ? 0 : 32) + (j & 31);
    }
    out[n_off + row] = (float)((bias ? (double)bias[n] : 0.0) + s);
  }
}

hipError_t launch_fold_bias(const float* W, const float* bias, const float* beta, float* out, int N, int C,
                            int n_off, int geglu, hipStream_t st) {
  hipLaunchKernelGGL(k_fold_bias, dim3((N + 3) / 4), dim3(256), 0, st, W, bias, beta, out, N, C, n_off, geglu);
  return hipGetLastError();
}

// out[n, k] = sum_c A[n, c] * Bm[c, k]   (fp64 accumulate; one-off weight products at prepare time)
__global__ __launch_bounds__(256) void k_matmul_f32(const float* __restrict__ A, const float* __restrict__ Bm,
                                                     float* __restrict__ out, int N, int Cc, int K) {
  const int k = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (k >= K || n >= N) return;
  double s = 0;
  for (int c = 0; c < Cc; ++c) s += (double)A[(size_t)n * Cc + c] * (double)Bm[(size_t)c * K + k];
  out[(size_t)n * K + k] = (float)s;
}
hipError_t launch_matmul_f32(const float* A, const float* Bm, float* out, int N, int Cc, int K, hipStream_t st) {
  hipLaunchKernelGGL(k_matmul_f32, dim3((K + 255) / 256, N), dim3(256), 0, st, A, Bm, out, N, Cc, K);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
__global__ void k_copy_f32(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}
__global__ void k_fill_f32(float* __restrict__ dst, float v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = v;
}
hipError_t launch_copy_f32(const float* src, float* dst, int64_t n, hipStream_t st) {
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_copy_f32, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, src, dst, n);
  return hipGetLastError();
}
hipError_t launch_fill_f32(float* dst, float v, int64_t n, hipStream_t st) {
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_fill_f32, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, dst, v, n);
  return hipGetLastError();
}

// Sampler update (reference dpm_solver.py:574-577, 815-819, 873-889; uni_pc.py:529-556): every
// multistep predictor/corrector update is a fixed linear combination of the state and the
// model-output history,
//   out = c0*x + c1*m0 + c2*m1 + c3*m2 + c4*m3,
// with per-step coefficients precomputed on the host in fp64 and read from a device row
// (8 floats).  `out` may alias `x`.
__global__ void k_lincomb(float* out, const float* x, const float* m0, const float* m1, const float* m2,
                          const float* m3, const float* __restrict__ coef, int64_t n4) {
  const float c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3], c4 = coef[4];
  if (coef[7] != 0.f) {     // x0 -> noise prediction: (x - alpha m0) / sigma, every step rounded as torch rounds it (sampler.hip)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
      const float4 v = reinterpret_cast<const float4*>(x)[i], a = reinterpret_cast<const float4*>(m0)[i];
      reinterpret_cast<float4*>(out)[i] = make_float4(__fdiv_rn(__fsub_rn(v.x, __fmul_rn(c1, a.x)), c0), __fdiv_rn(__fsub_rn(v.y, __fmul_rn(c1, a.y)), c0),
                                                      __fdiv_rn(__fsub_rn(v.z, __fmul_rn(c1, a.z)), c0), __fdiv_rn(__fsub_rn(v.w, __fmul_rn(c1, a.w)), c0));
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x *= c0; v.y *= c0; v.z *= c0; v.w *= c0;
#define DV_ACC(ptr, c)                                                                   \
    if (ptr) {                                                                           \
      const float4 a = reinterpret_cast<const float4*>(ptr)[i];                          \
      v.x = fmaf(c, a.x, v.x); v.y = fmaf(c, a.y, v.y); v.z = fmaf(c, a.z, v.z); v.w = fmaf(c, a.w, v.w); \
    }
    DV_ACC(m0, c1) DV_ACC(m1, c2) DV_ACC(m2, c3) DV_ACC(m3, c4)
#undef DV_ACC
    reinterpret_cast<float4*>(out)[i] = v;
  }
}
__global__ void k_lincomb_tail(float* out, const float* x, const float* m0, const float* m1, const float* m2,
                               const float* m3, const float* __restrict__ coef, int64_t start, int64_t n) {
  const int64_t i = start + threadIdx.x;
  if (i >= n) return;
  if (coef[7] != 0.f) { out[i] = __fdiv_rn(__fsub_rn(x[i], __fmul_rn(coef[1], m0[i])), coef[0]); return; }
  float v = coef[0] * x[i];
  if (m0) v = fmaf(coef[1], m0[i], v);
  if (m1) v = fmaf(coef[2], m1[i], v);
  if (m2) v = fmaf(coef[3], m2[i], v);
  if (m3) v = fmaf(coef[4], m3[i], v);
  out[i] = v;
}

hipError_t launch_lincomb(float* out, const float* x, const float* m0, const float* m1, const float* m2,
                          const float* m3, const float* coef, int64_t n, hipStream_t st) {
  const int64_t n4 = n / 4;
  if (n4 > 0) {
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_lincomb, dim3(blocks), dim3(256), 0, st, out, x, m0, m1, m2, m3, coef, n4);
  }
  if (n4 * 4 < n) hipLaunchKernelGGL(k_lincomb_tail, dim3(1), dim3(4), 0, st, out, x, m0, m1, m2, m3, coef, n4 * 4, n);
  return hipGetLastError();
}


// ---- 32x16-block statistics of a channels-last fp32 tensor [M, C] -> stat16 [M/32, C/16, 2] = (sum, squared deviations
// from the block's own mean): what a GEMM epilogue writes beside its output for the GroupNorm of its consumer.
// Standalone form for tensors that did not come out of such an epilogue.
__global__ __launch_bounds__(64) void k_stat16(const float* __restrict__ x, float* __restrict__ stat16, int M, int C) {
  const int rb = blockIdx.x, cb = blockIdx.y, lane = threadIdx.x;
  const int row = rb * 32 + (lane >> 1), col = cb * 16 + (lane & 1) * 8;
  float v[8];
  const bool ok = row < M;
  const float4 a = ok ? *reinterpret_cast<const float4*>(x + (size_t)row * C + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 b = ok ? *reinterpret_cast<const float4*>(x + (size_t)row * C + col + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  float s1 = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) s1 += v[e];
  s1 = wave_sum(s1);
  const float mb = s1 * (1.0f / 512.0f);
  float s2 = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { const float d = v[e] - mb; s2 = fmaf(d, d, s2); }
  s2 = wave_sum(s2);
  if (lane == 0) reinterpret_cast<float2*>(stat16)[(size_t)rb * (C >> 4) + cb] = make_float2(s1, s2);
}
hipError_t launch_stat16(const float* x, float* stat16, int M, int C, hipStream_t st) {
  if (M % 32 != 0 || C % 16 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_stat16, dim3(M / 32, C / 16), dim3(64), 0, st, x, stat16, M, C);
  return hipGetLastError();
}
