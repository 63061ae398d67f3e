// Implicit-GEMM contraction kernel for gfx950 (MI355X): every conv1d (k=1/3, stride 1/2,
// nearest-upsample folded into the gather, channel-concat of two sources) and every Linear
// of the denoiser (reference unet1d/resnet.py:591-641, transformer_1d.py:264-300,
// attention.py:130-203, attention_processor.py:1008-1046) is one launch of this kernel on
// channels-last fp32 activations.
//
//   A operand : gathered fp32 rows -> prologue (GroupNorm/temb affine + SiLU, LayerNorm)
//               -> split into bf16 hi + bf16 lo in registers -> LDS
//   B operand : weights pre-packed as bf16 hi / lo, [N][K] with K contiguous
//   MFMA      : v_mfma_f32_32x32x16_bf16, fp32 accumulate;  bf16x3 mode issues
//               hi*hi + lo*hi + hi*lo (1.3e-5 rel. error vs fp32),
//               bf16 mode issues hi*hi only
//   epilogue  : + bias, (+ residual | GEGLU a*gelu_erf(g) | transposed [B,N,T] store)
//
// Tile: BM x BN x 32 per 256-thread workgroup (4 waves of 64 lanes); LDS rows are padded
// to 80 bytes so the 16-byte MFMA operand reads (ds_read_b128) are bank-conflict free;
// two LDS buffers, one barrier per k-tile; global loads of tile k+1 are in flight while
// tile k is multiplied.
#include "dv_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define ROWB 80
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// split 4 floats into packed bf16 hi (2 dwords) and lo (2 dwords)
template <bool SPLIT>
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  hi.x = cvt_pk_bf16(v.x, v.y);
  hi.y = cvt_pk_bf16(v.z, v.w);
  if (SPLIT) {
    float rx = v.x - __uint_as_float(hi.x << 16);
    float ry = v.y - __uint_as_float(hi.x & 0xffff0000u);
    float rz = v.z - __uint_as_float(hi.y << 16);
    float rw = v.w - __uint_as_float(hi.y & 0xffff0000u);
    lo.x = cvt_pk_bf16(rx, ry);
    lo.y = cvt_pk_bf16(rz, rw);
  }
}

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// PRO = prologue family of segment 0, a compile-time choice so that the hot loop has no
// control flow around its loads (every global load of a k-tile is issued back to back and
// waited for once): 0 none, 1 per-(batch,channel) affine (+SiLU when seg.pro says so),
// 3 LayerNorm (per-row mean/rstd, loaded once before the loop).  Segment 1 (the 1x1 shortcut
// folded into conv2) never has a prologue.
template <int WM, int WN, int FM, int FN, int NSPLIT, int PRO>
__global__ __launch_bounds__(256) void k_gemm(const GemmParams p) {
  constexpr int BM = WM * FM * 32, BN = WN * FN * 32;
  constexpr bool SPLIT = NSPLIT == 3;
  constexpr int APASS = BM / 32;        // float4 loads per thread per A tile
  constexpr int BPASS = BN / 64;        // 16-byte loads per thread per B array per tile
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
  constexpr int BUF_BYTES = (A_BYTES + B_BYTES) * (SPLIT ? 2 : 1);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware tile order: consecutive tile ids (same A rows, neighbouring N) share an L2.
  const int n_tiles_n = (p.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int m0 = (bid / n_tiles_n) * BM;
  const int n0 = (bid % n_tiles_n) * BN;

  // ---- per-thread A-gather rows (fixed for the whole K loop) ----
  const int a_row = tid >> 3;           // 0..31 within a pass
  const int a_kq = tid & 7;             // float4 column within the 32-wide k-tile
  int row_b[APASS], row_t[APASS];
  unsigned row_ok = 0;
  float ln_sc[APASS], ln_sh[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    int m = m0 + i * 32 + a_row;
    const bool ok = m < p.M;
    row_ok |= (ok ? 1u : 0u) << i;
    m = ok ? m : 0;
    row_b[i] = m / p.T_out;
    row_t[i] = m - row_b[i] * p.T_out;
    if (PRO == PRO_LN) {                // 1x1, stride 1: source row == output row
      const float mu = p.seg[0].p0[m], rs = p.seg[0].p1[m];
      ln_sc[i] = rs;
      ln_sh[i] = -mu * rs;
    }
  }
  const int b_row = tid >> 2;           // 0..63 within a pass
  const int b_ch = tid & 3;             // 16-byte chunk within the 64-byte k-tile row

  // ---- staging registers ----
  float4 ra[APASS], rsc[APASS], rsh[APASS];
  u32x4 rbh[BPASS], rbl[BPASS];
  unsigned valid_mask = 0;
  bool cur_seg0 = true;
  // running decode of the k-tile being loaded
  int ld_seg = 0, ld_tap = 0, ld_cc = 0;
  const int total_kt = p.seg[0].nkt + (p.nseg > 1 ? p.seg[1].nkt : 0);
  const bool act_silu = p.seg[0].pro == PRO_AFFINE_SILU;

  auto load_tile = [&](int kt) {
    const GemmSeg& s = p.seg[ld_seg];
    const int ctot = s.c0 + s.c1;
    const bool first = ld_cc < s.c0;
    const float* src = first ? s.a0 : s.a1;
    const int ld = first ? s.c0 : s.c1;
    const int col = (first ? ld_cc : ld_cc - s.c0) + a_kq * 4;
    cur_seg0 = ld_seg == 0;
    // affine table offset; for segment-1 tiles a valid dummy offset (values ignored at convert time)
    const int aff_c = (cur_seg0 ? ld_cc : 0) + a_kq * 4;
    const int ctot0 = p.seg[0].c0 + p.seg[0].c1;
    valid_mask = 0;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int ts = row_t[i] * p.stride + ld_tap - s.pad;
      const bool ok = ((row_ok >> i) & 1u) && ts >= 0 && ts < p.T_virt;
      int st = ts;
      st = p.up_mode == UP_X2 ? (ts >> 1) : st;
      st = p.up_mode == UP_SIZE ? min((int)floorf((float)ts * p.up_scale), p.T_in - 1) : st;
      st = ok ? st : 0;                                  // clamp: the load is unconditional
      const size_t srow = (size_t)row_b[i] * p.T_in + st;
      ra[i] = *reinterpret_cast<const float4*>(src + srow * ld + col);
      valid_mask |= (ok ? 1u : 0u) << i;
      if (PRO == PRO_AFFINE_SILU) {
        const size_t o = (size_t)row_b[i] * ctot0 + aff_c;
        rsc[i] = *reinterpret_cast<const float4*>(p.seg[0].p0 + o);
        rsh[i] = *reinterpret_cast<const float4*>(p.seg[0].p1 + o);
      }
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const size_t o = (size_t)(n0 + i * 64 + b_row) * p.Kp + (size_t)kt * 32 + b_ch * 8;
      rbh[i] = *reinterpret_cast<const u32x4*>(p.w_hi + o);
      if (SPLIT) rbl[i] = *reinterpret_cast<const u32x4*>(p.w_lo + o);
    }
    // advance the decode state to the next k-tile
    ld_cc += 32;
    if (ld_cc == ctot) {
      ld_cc = 0;
      if (++ld_tap == s.taps) { ld_tap = 0; ++ld_seg; }
    }
  };

  auto store_tile = [&](int buf) {
    char* base = smem + buf * BUF_BYTES;
    char* a_hi = base;
    char* a_lo = base + A_BYTES;
    char* b_hi = base + (SPLIT ? 2 : 1) * A_BYTES;
    char* b_lo = b_hi + B_BYTES;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      float4 v = ra[i];
      if (PRO == PRO_AFFINE_SILU) {
        if (cur_seg0) {
          v.x = fmaf(v.x, rsc[i].x, rsh[i].x);
          v.y = fmaf(v.y, rsc[i].y, rsh[i].y);
          v.z = fmaf(v.z, rsc[i].z, rsh[i].z);
          v.w = fmaf(v.w, rsc[i].w, rsh[i].w);
          if (act_silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
        }
      } else if (PRO == PRO_LN) {
        v.x = fmaf(v.x, ln_sc[i], ln_sh[i]);
        v.y = fmaf(v.y, ln_sc[i], ln_sh[i]);
        v.z = fmaf(v.z, ln_sc[i], ln_sh[i]);
        v.w = fmaf(v.w, ln_sc[i], ln_sh[i]);
      }
      if (!((valid_mask >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);   // conv zero padding / rows >= M
      uint2 hi, lo;
      split4<SPLIT>(v, hi, lo);
      const int off = (i * 32 + a_row) * ROWB + a_kq * 8;
      *reinterpret_cast<uint2*>(a_hi + off) = hi;
      if (SPLIT) *reinterpret_cast<uint2*>(a_lo + off) = lo;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const int off = (i * 64 + b_row) * ROWB + b_ch * 16;
      *reinterpret_cast<u32x4*>(b_hi + off) = rbh[i];
      if (SPLIT) *reinterpret_cast<u32x4*>(b_lo + off) = rbl[i];
    }
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute_tile = [&](int buf) {
    const char* base = smem + buf * BUF_BYTES;
    const char* a_hi = base;
    const char* a_lo = base + A_BYTES;
    const char* b_hi = base + (SPLIT ? 2 : 1) * A_BYTES;
    const char* b_lo = b_hi + B_BYTES;
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int koff = ks * 32 + lh * 16;   // bytes within the row
      bf16x8 ah[FM], al[FM], bh[FN], bl[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int off = ((wm * FM + i) * 32 + l31) * ROWB + koff;
        ah[i] = *reinterpret_cast<const bf16x8*>(a_hi + off);
        if (SPLIT) al[i] = *reinterpret_cast<const bf16x8*>(a_lo + off);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int off = ((wn * FN + j) * 32 + l31) * ROWB + koff;
        bh[j] = *reinterpret_cast<const bf16x8*>(b_hi + off);
        if (SPLIT) bl[j] = *reinterpret_cast<const bf16x8*>(b_lo + off);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          if (SPLIT) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  // ---- main loop (iteration -1 is the pipeline fill: load + store of tile 0 only) ----
  for (int kt = -1; kt < total_kt; ++kt) {
    const bool more = kt + 1 < total_kt;
    if (more) load_tile(kt + 1);
    if (kt >= 0) compute_tile(kt & 1);
    if (more) store_tile((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue ----
  const int l31 = lane & 31, lh = lane >> 5;
  if (p.epi == EPI_GEGLU) {
    // packed column order: per 64-column block, [32 x a | 32 x gate]  (FN == 2 per wave)
    if constexpr (FN == 2) {
      const int blk = (n0 + wn * 64) >> 6;          // 64-column block index
      const int oc = blk * 32 + l31;                // output column
      const int ncol = n0 + wn * 64 + l31;          // packed column of `a`
      if (ncol < p.N) {
        const float ba = p.bias ? p.bias[ncol] : 0.f;
        const float bg = p.bias ? p.bias[ncol + 32] : 0.f;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * FM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < p.M) {
              const float a = acc[i][0][r] + ba;
              const float g = acc[i][1][r] + bg;
              p.out[(size_t)m * p.ldo + oc] = a * gelu_erf(g);
            }
          }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = n0 + (wn * FN + j) * 32 + l31;
    if (n >= p.N) continue;
    const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * FM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= p.M) continue;
        float v = acc[i][j][r] + bv;
        if (p.epi == EPI_RESIDUAL) v += p.res[(size_t)m * p.ldres + n];
        if (p.epi == EPI_STORE_NCT) {
          const int b = m / p.T_out, t = m - b * p.T_out;
          p.out[((size_t)b * p.N + n) * p.T_out + t] = v;
        } else {
          p.out[(size_t)m * p.ldo + n] = v;
        }
      }
  }
}

template <int WM, int WN, int FM, int FN, int NSPLIT, int PRO>
struct GemmCfg {
  static constexpr int BM = WM * FM * 32, BN = WN * FN * 32;
  static constexpr int SMEM = 2 * (BM + BN) * ROWB * (NSPLIT == 3 ? 2 : 1);
  // > 64 KiB of dynamic LDS needs the attribute; set once, outside any stream capture
  static hipError_t init() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm<WM, WN, FM, FN, NSPLIT, PRO>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
  }
  static hipError_t launch(const GemmParams& p, hipStream_t st) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL((k_gemm<WM, WN, FM, FN, NSPLIT, PRO>), dim3(tiles), dim3(256), SMEM, st, p);
    return hipGetLastError();
  }
};

template <int WM, int WN, int FM, int FN>
struct GemmTile {
  static hipError_t init() {
    hipError_t e;
    if ((e = GemmCfg<WM, WN, FM, FN, 3, PRO_NONE>::init()) != hipSuccess) return e;
    if ((e = GemmCfg<WM, WN, FM, FN, 3, PRO_AFFINE_SILU>::init()) != hipSuccess) return e;
    if ((e = GemmCfg<WM, WN, FM, FN, 3, PRO_LN>::init()) != hipSuccess) return e;
    if ((e = GemmCfg<WM, WN, FM, FN, 1, PRO_NONE>::init()) != hipSuccess) return e;
    if ((e = GemmCfg<WM, WN, FM, FN, 1, PRO_AFFINE_SILU>::init()) != hipSuccess) return e;
    return GemmCfg<WM, WN, FM, FN, 1, PRO_LN>::init();
  }
  static hipError_t launch(const GemmParams& p, bool x3, hipStream_t st) {
    const int pro = p.seg[0].pro == PRO_AFFINE ? PRO_AFFINE_SILU : p.seg[0].pro;   // affine family
    if (x3) {
      if (pro == PRO_NONE) return GemmCfg<WM, WN, FM, FN, 3, PRO_NONE>::launch(p, st);
      if (pro == PRO_LN) return GemmCfg<WM, WN, FM, FN, 3, PRO_LN>::launch(p, st);
      return GemmCfg<WM, WN, FM, FN, 3, PRO_AFFINE_SILU>::launch(p, st);
    }
    if (pro == PRO_NONE) return GemmCfg<WM, WN, FM, FN, 1, PRO_NONE>::launch(p, st);
    if (pro == PRO_LN) return GemmCfg<WM, WN, FM, FN, 1, PRO_LN>::launch(p, st);
    return GemmCfg<WM, WN, FM, FN, 1, PRO_AFFINE_SILU>::launch(p, st);
  }
};

hipError_t gemm_init() {
  hipError_t e;
  if ((e = GemmTile<2, 2, 2, 2>::init()) != hipSuccess) return e;
  if ((e = GemmTile<4, 1, 1, 2>::init()) != hipSuccess) return e;
  return GemmTile<2, 2, 1, 1>::init();
}

// Tile choice: the denoiser's GEMMs are small (M = B*T_l <= 8192, N = 128..4096), so the
// first concern is filling 256 CUs; 128x128 tiles only when they still give >= 1.5 workgroups
// per CU, else 64x64.  GEGLU needs both halves of a 64-column block in one wave (FN == 2):
// 128x128 or 128x64 (4x1 waves).
hipError_t launch_gemm(const GemmParams& p, int precision, hipStream_t st) {
  const int big_tiles = ((p.M + 127) / 128) * ((p.N + 127) / 128);
  const bool x3 = precision == 0;
  if (p.nseg > 1 && p.seg[1].pro != PRO_NONE) return hipErrorInvalidValue;
  if (big_tiles >= 384) return GemmTile<2, 2, 2, 2>::launch(p, x3, st);
  if (p.epi == EPI_GEGLU) return GemmTile<4, 1, 1, 2>::launch(p, x3, st);
  return GemmTile<2, 2, 1, 1>::launch(p, x3, st);
}
