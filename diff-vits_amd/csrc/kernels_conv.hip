// The ResnetBlock2D / Upsample2D convolutions of the denoiser on gfx950 (MI355X) with the A operand RESIDENT in LDS (reference
// unet1d/resnet.py:591-641: conv1 / conv2 over [B, C, T], kernel 3, padding 1, stride 1, conv2 with its 1x1 shortcut; resnet.py:138-187:
// nearest x2 + the same convolution).  Three kernels, one tile scheme:
//   k_conv3<CIN>  - all C_in = 128 / 256 / 384 / 512 channels of the tile's rows resident (one tensor of planes, or the concatenation of two)
//   k_conv3s      - any width up to 1024 channels and / or a second, one-tap K-segment (the folded shortcut): the rows streamed through
//                   a ring of 64-channel chunks by a producer wave; fused split-K pairs on 128-tile grids
//   k_conv3u<CIN> - the upsampling form: 128 output rows per tile over the same 64 + 2 source rows
//
// Why not k_gemm (kernels_gemm.hip).  It stages BOTH operands of every 64-deep k-tile through an LDS ring with one barrier per
// k-tile; its k-loop sits on three per-CU bounds at once - instruction issue, LDS bytes and the vector-memory path - at 1050-1170
// cycles per 64x64x64 k-tile against 384 cycles of MFMA (docs/HISTORY.md, round 3).  A 3-tap convolution reads every activation row
// three times (once per tap), and the 64 + 2 rows a 64-row tile touches are 34-135 KiB as split planes at up to 512 channels: they
// fit.  So, exactly like k_ff_split (kernels_ffsplit.hip):
//   * the tile's rows [m0 - 1, m0 + 64] x all C_in channels are DMA'd into LDS ONCE (the three taps read them at row offsets
//     -1 / 0 / +1; the two halo rows are zeros at an utterance's ends);
//   * the weights never touch LDS: fragment-major (launch_relayout_frag of the packed [N][3 C_in] planes), each wave loads the
//     fragments it multiplies with coalesced 16-byte loads, 6-8 ahead in registers;
//   * 8 waves = 2 column fragments x 4 k-quarters, each wave multiplies BOTH row fragments with every weight fragment it loads
//     (half the weight bytes per MFMA of a one-fragment wave); no barrier inside the resident k-loop;
//   * the four k-quarters are added through LDS and every wave finishes a 32-row x 16-column half fragment - the layout of
//     k_gemm's half-fragment epilogue, whose steps follow unchanged: bias, residual, fp32 / split-plane stores, 32x16 block
//     statistics, the consumer's GroupNorm finished in the launch (gnx_device.h).
// The tile grid, the XCD rectangle and the exchange-word layout are those of k_gemm's 64x64 tile: launch_gemm (kernels_gemm.hip)
// dispatches here when gemm_conv3_shape_ok() / gemm_conv3_up_ok() hold and the caller gives the fragment-major weights (GemmParams
// wf_hi / wf_lo, c3_route; engine.hip conv3_route: from 64 tiles, any grid size).  Round 6: the row tiles are laid out PER UTTERANCE
// (C3Tile below) - any row pitch of whole 32-frame blocks, the padded row space of real utterance lengths included.  Measured (profiles/r05_*conv3*): the resident k-loops run AT the
// MFMA rate, the streamed one at ~1900 cycles per 64-channel chunk (1152 of MFMA: the per-CU vector-memory path carries the rows AND
// the weights); 48 of the forward's 62 GEMM launches, family 1.24 -> 1.06 ms per forward, +4.9 % on the 50-step run.
#include "dv_common.h"
#include "dv_device.h"
#include "gnx_device.h"

#include <cstdlib>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#ifdef DV_GEMM_TRACE
// development build only (make trace): per-workgroup s_memtime stamps of the phases (tools/conv3_trace.py)
__device__ unsigned long long g_c3_trace[1024 * 16];
__device__ int g_c3_sel = 0;                       // which launches stamp: C_in (0 = any) - set by the tool
#define DV_C3TRACE(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024 && (g_c3_sel == 0 || g_c3_sel == CIN)) g_c3_trace[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int dv_debug_c3_trace_select(int cin) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_c3_sel), &cin, sizeof(cin));
  void* d = nullptr;
  if (e == hipSuccess) e = hipGetSymbolAddress(&d, HIP_SYMBOL(g_c3_trace));
  return (int)(e != hipSuccess ? e : hipMemset(d, 0, sizeof(g_c3_trace)));
}
extern "C" int dv_debug_c3_trace(unsigned long long* host, int n_wg) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_c3_trace), (size_t)n_wg * 16 * sizeof(unsigned long long));
}
#else
#define DV_C3TRACE(i) do {} while (0)
#endif

namespace {

constexpr int BM = 64, BN = 64, NWV = 8, NT = 64 * NWV;
constexpr int ROWS = BM + 2;                         // LDS rows of a chunk: the tile's 64, then row m0 + 64 (index 64), then row m0 - 1 (index 65)
constexpr int CHP = ROWS * 128;                      // bytes of one 64-channel chunk of one plane
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }
struct BFrag { bf16x8 h, l; };
constexpr int RED_BYTES = 2 * 4 * 2 * 4 * 64 * 16;      // k-quarter exchange: [column fragment][quarter][row fragment][4 column groups][64 lanes] float4

template <int CIN>
struct ConvGeom {
  static constexpr int NCH = CIN / 64;               // 64-channel chunks
  static constexpr int KPT = CIN / 16;               // 16-deep k-steps per tap
  static constexpr int KS = 3 * KPT;                 // ... of the whole contraction (K order: tap, channel - the packed weights' order)
  static constexpr int U = KS / 4;                   // ... per k-quarter (wave)
  static constexpr int DEPTH = U < 8 ? U : 8;        // weight units (hi + lo fragment: 8 VGPRs) in flight per wave
  static constexpr int A_PL = NCH * CHP;             // bytes per plane of the resident operand
  static constexpr int RED = RED_BYTES;   // k-quarter exchange: [column fragment][quarter][row fragment][4 column groups][64 lanes] float4
  static constexpr int SMEM = 2 * A_PL > RED ? 2 * A_PL : RED;
  static_assert(CIN % 64 == 0 && KS % 4 == 0 && SMEM + 4096 <= 160 * 1024, "geometry");
};

}  // namespace

// Tile of this workgroup: k_gemm's XCD-aware order (workgroup b runs on XCD b % 8) over a grid of ROW TILES THAT NEVER SPAN TWO
// UTTERANCES.  The row space may be padded (GemmParams: row pitch T_out = a multiple of 32, Tv_out <= T_out frames exist - the real
// inference call has any length, reference tts_infer.py:46-74): an utterance has c3_tu = ceil(T_out / bm) row tiles, the last one
// `rows` < bm rows (a multiple of 32) where the pitch is no multiple of bm.
// (ksel: the k-half of a fused split-K pair - GemmParams sk_mode 3 - else 0)
struct C3Tile { int m0, n0, t0, rows, ksel, bq; };   // first row, first column, frame of row m0 inside its utterance, rows of the tile, utterance
__device__ __forceinline__ void conv3_tile(const GemmParams& p, C3Tile& t, const int bm) {
  const int n_tiles_n = p.N / BN, nwg = gridDim.x;
  int bid = blockIdx.x, tmi;
  t.ksel = 0;
  if (p.xcd_n > 0) {
    const int x = bid & 7, i = bid >> 3;
    const int r = x & ((1 << p.xcd_sh_mn) - 1), xm_i = r >> p.xcd_sh_n, xn_i = r & (p.xcd_n - 1);
    const int lm = (i * p.xcd_inv_tn) >> 16, ln = i - lm * p.xcd_tn;
    tmi = xm_i * p.xcd_tm + lm; t.n0 = (xn_i * p.xcd_tn + ln) * BN; t.ksel = x >> p.xcd_sh_mn;
  } else {
    if (p.xcd_n == 0) {                              // (xcd_n < 0: plain order - an utterance's tiles spread over all XCDs)
      const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    if (p.sk_mode == 3) {                            // XCD-contiguous ids share a k-half
      const int tiles = nwg >> 1;
      t.ksel = bid / tiles;
      bid -= t.ksel * tiles;
    }
    tmi = bid / n_tiles_n; t.n0 = (bid - tmi * n_tiles_n) * BN;
  }
  t.bq = p.c3_tu_magic ? (int)__umulhi((unsigned)tmi, p.c3_tu_magic) : tmi;   // (magic 0: one tile per utterance)
  t.t0 = (tmi - t.bq * p.c3_tu) * bm;
  t.m0 = t.bq * p.T_out + t.t0;
  t.rows = min(bm, p.T_out - t.t0);
}

// The part both kernels share: the four k-quarters' accumulators (wave = column fragment cf = wave & 1, quarter kq = wave >> 1, both
// row fragments) are added through LDS - `smem` is free: the caller's waves are past their last operand read only after the barrier
// here - and every wave finishes a 32-row x 16-column half fragment: the steps of gemm_tile.h's half-fragment epilogue.
__device__ __forceinline__ void conv3_finish(const GemmParams& p, char* smem, const float* s_bias, GnxShared<BN>& s_gnx, f32x16 (&acc)[2],
                                             const float (&rpre)[8], const C3Tile& tl, const int tid, const int lane, const int wave,
                                             const int CIN) {
  (void)CIN;                                         // (trace builds select launches by it)
  const int m0 = tl.m0, n0 = tl.n0, ksel = tl.ksel;
  const int l31 = lane & 31, lh = lane >> 5;
  const int cf = wave & 1, kq = wave >> 1;
  const int e_wn = cf, e_wm = (wave >> 1) & 1, e_half = wave >> 2;   // epilogue role: fragment (e_wm, e_wn), its columns [16 e_half, +16)
  // ---- the four k-quarters are added through LDS; this wave keeps a 32-row x 16-column half fragment ----
  __syncthreads();                                   // every wave is done reading the resident operand
  float4* const red4 = reinterpret_cast<float4*>(smem);
#pragma unroll
  for (int rf = 0; rf < 2; ++rf)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      red4[((((cf * 4 + kq) * 2 + rf) * 4 + g) << 6) + lane] = make_float4(acc[rf][4 * g], acc[rf][4 * g + 1], acc[rf][4 * g + 2], acc[rf][4 * g + 3]);
  __syncthreads();
  float vv[8];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    float4 s = red4[((((e_wn * 4 + 0) * 2 + e_wm) * 4 + e_half * 2 + g) << 6) + lane];
#pragma unroll
    for (int k = 1; k < 4; ++k) {                    // (quarters in order: deterministic)
      const float4 v = red4[((((e_wn * 4 + k) * 2 + e_wm) * 4 + e_half * 2 + g) << 6) + lane];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    vv[4 * g] = s.x; vv[4 * g + 1] = s.y; vv[4 * g + 2] = s.z; vv[4 * g + 3] = s.w;
  }
  DV_C3TRACE(4);
  if (p.sk_mode == 3) {
    // fused split-K pair (the protocol of gemm_tile.h's sk_mode 3): this workgroup multiplied one half of the chunks.  It writes
    // its half-fragment sums through and takes a ticket; the first of the pair to arrive leaves, the second adds the partner's
    // sums (a + b = b + a: the same bits whoever finishes) and runs the epilogue.  Nobody waits for anybody.
    const int tmi = tl.bq * p.c3_tu + tl.t0 / BM;    // row-tile index
    const int tile = tmi * (p.N / BN) + n0 / BN, tiles = (int)(gridDim.x >> 1);
    float4* const d0 = reinterpret_cast<float4*>(p.sk_buf) + ((size_t)(ksel * tiles + tile) * NWV + wave) * 128 + lane;
    st_handover16(d0, make_float4(vv[0], vv[1], vv[2], vv[3]));
    st_handover16(d0 + 64, make_float4(vv[4], vv[5], vv[6], vv[7]));
    wait_vmcnt<0>();                                 // this thread's sums have been written through
    __shared__ unsigned s_arrival;
    __syncthreads();
    unsigned* const ticket = p.sk_ticket + tile;
    if (tid == 0) s_arrival = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_arrival == 0) return;                      // the partner finishes this tile
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    const float4* const s0 = reinterpret_cast<const float4*>(p.sk_buf) + ((size_t)((ksel ^ 1) * tiles + tile) * NWV + wave) * 128 + lane;
    const float4 v0 = ld_handover16(s0), v1 = ld_handover16(s0 + 64);
    vv[0] += v0.x; vv[1] += v0.y; vv[2] += v0.z; vv[3] += v0.w;
    vv[4] += v1.x; vv[5] += v1.y; vv[6] += v1.z; vv[7] += v1.w;
  }

  // ---- epilogue of the half fragment (the steps of gemm_tile.h's half-fragment epilogue) ----
  // Padded row space: a fragment beyond the tile's rows (the short last tile of an utterance) belongs to the NEXT utterance and is
  // not touched; rows [Tv_out, T_out) of this one are stored as zeros and kept out of the statistics (dv_common.h GemmParams).
  const bool gnx_h = p.gnx.xchg != nullptr;
  const int coff = e_half * 16;                      // this wave's columns inside the fragment
  const int ncol = n0 + e_wn * 32 + coff;            // first of them
  const int m = m0 + e_wm * 32 + l31, mrow0 = m0 + e_wm * 32;
  const bool live = e_wm * 32 < tl.rows;             // (wave-uniform)
  const bool padded = p.Tv_out != p.T_out;
  const int tfr = tl.t0 + e_wm * 32;                 // frame of the fragment's first row
  const bool m_ok = tfr + l31 < p.Tv_out;
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const float4 b4 = *reinterpret_cast<const float4*>(s_bias + e_wn * 32 + coff + 4 * lh + 8 * g);
    vv[4 * g] += b4.x; vv[4 * g + 1] += b4.y; vv[4 * g + 2] += b4.z; vv[4 * g + 3] += b4.w;
  }
  if (p.epi == EPI_RESIDUAL) {
#pragma unroll
    for (int r = 0; r < 8; ++r) vv[r] += rpre[r];
  }
  if (padded) {
#pragma unroll
    for (int r = 0; r < 8; ++r) vv[r] = m_ok ? vv[r] : 0.f;
  }
  if (live) {
    const size_t ob = (size_t)m * p.ldo + ncol;
    if (p.out) {
#pragma unroll
      for (int g = 0; g < 2; ++g)
        dv_st16(p.out + ob + 4 * lh + 8 * g, make_float4(vv[4 * g], vv[4 * g + 1], vv[4 * g + 2], vv[4 * g + 3]));
    }
    if (p.out_hi) store_planes8(p.out_hi, p.out_lo, ob, lh, vv);
  }
  DV_C3TRACE(5);
  if (p.stats16 && live) {                           // this wave's 32 x 16 block: (sum, squared deviations about its own mean)
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) a1 += vv[r];
    a1 = wave_sum64(a1);
    const float mb = padded ? a1 / (float)(16 * min(32, p.Tv_out - tfr)) : a1 * (1.0f / 512.0f);
#pragma unroll
    for (int r = 0; r < 8; ++r) { const float dv = (!padded || m_ok) ? vv[r] - mb : 0.f; a2 = fmaf(dv, dv, a2); }
    a2 = wave_sum64(a2);
    if (lane == 0) {
      const size_t e = (size_t)(mrow0 >> 5) * (p.N >> 4) + (ncol >> 4);
      reinterpret_cast<float2*>(p.stats16)[e] = make_float2(a1, a2);
      if (gnx_h)
        __hip_atomic_store(p.gnx.xchg + e, (unsigned long long)__float_as_uint(a1) | ((unsigned long long)__float_as_uint(a2) << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  DV_C3TRACE(6);
  if (gnx_h) {
    GnxTile t;
    t.M = p.M; t.N = p.N; t.T_out = p.T_out; t.Tv_out = p.Tv_out; t.m0 = m0; t.n0 = n0; t.bm = tl.rows; t.bn = BN;
    t.bq = tl.bq;
    gnx_finish_table<BN>(p.gnx, t, s_gnx, tid, lane, wave, NWV, [&](int) {});
    DV_C3TRACE(7);
    float y[8];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int cl = e_wn * 32 + coff + 4 * lh + 8 * g;
      const float4 sa = *reinterpret_cast<const float4*>(s_gnx.gA + cl);
      const float4 sb = *reinterpret_cast<const float4*>(s_gnx.gB + cl);
      y[4 * g] = fmaf(vv[4 * g], sa.x, sb.x); y[4 * g + 1] = fmaf(vv[4 * g + 1], sa.y, sb.y);
      y[4 * g + 2] = fmaf(vv[4 * g + 2], sa.z, sb.z); y[4 * g + 3] = fmaf(vv[4 * g + 3], sa.w, sb.w);
    }
    if (p.gnx.silu) {
#pragma unroll
      for (int r = 0; r < 8; ++r) y[r] = y[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-y[r]));
    }
    if (padded) {                                    // (rows that do not exist: zeros, like every producer of planes)
#pragma unroll
      for (int r = 0; r < 8; ++r) y[r] = m_ok ? y[r] : 0.f;
    }
    if (live) store_planes8(p.gnx.y_hi, p.gnx.y_lo, (size_t)m * p.N + ncol, lh, y);
  }
  DV_C3TRACE(8);
}

template <int CIN>
__global__ __launch_bounds__(NT) void k_conv3(const GemmParams p) {
  using G = ConvGeom<CIN>;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  __shared__ __attribute__((aligned(16))) float s_bias[BN];
  __shared__ GnxShared<BN> s_gnx;
  // (every 64-byte line of the argument block is requested at once: see k_gemm)
  asm volatile("" ::"s"(p.seg[0].a0_hi), "s"(p.seg[1].a0_hi), "s"(p.seg[1].pad), "s"(p.T_in), "s"(p.w_hi), "s"(p.M), "s"(p.res),
               "s"(p.out_hi), "s"(p.zero_page), "s"(p.ln_u), "s"(p.gnx.xchg), "s"(p.gnx.y_hi), "s"(p.xcd_n), "s"(p.xcd_inv_tn), "s"(p.wf_lo));
  DV_C3TRACE(0);
  C3Tile tl;
  conv3_tile(p, tl, BM);
  const int m0 = tl.m0, n0 = tl.n0, t0 = tl.t0;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int cf = wave & 1, kq = wave >> 1;           // k-loop role: column fragment, k-quarter
  const int e_wn = cf, e_wm = (wave >> 1) & 1, e_half = wave >> 2;   // epilogue role (conv3_finish): this wave's residual rows
  const unsigned a_base = (unsigned)(size_t)smem;
  const GemmSeg& sg = p.seg[0];
  const int c0 = sg.c0, c1 = sg.c1;
  const int m_last = p.M - 1;                        // (a short last tile: the rows behind it belong to the next utterance - read, never used - or lie beyond M)

  // ---- requests, oldest first: residual rows of this wave's half fragment, halo rows, bias, the tile's rows, first weights ----
  float rpre[8];
  if (p.epi == EPI_RESIDUAL) {
    const float* rp = p.res + (size_t)min(m0 + e_wm * 32 + l31, m_last) * p.ldres + n0 + e_wn * 32 + e_half * 16 + 4 * lh;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float4 a = *reinterpret_cast<const float4*>(rp + 8 * g);
      rpre[4 * g] = a.x; rpre[4 * g + 1] = a.y; rpre[4 * g + 2] = a.z; rpre[4 * g + 3] = a.w;
    }
  }
  // halo rows: 2 rows x 2 planes x CIN / 8 pieces of 16 bytes, one per thread (register round trip: 4 KiB at most)
  constexpr int HITEMS = CIN / 2;
  uint4 hv = make_uint4(0u, 0u, 0u, 0u);
  int h_dst = -1;
  if (tid < HITEMS) {
    const int pl = tid / (CIN / 4), rem = tid - pl * (CIN / 4), which = rem / (CIN / 8), piece = rem - which * (CIN / 8);
    const int ch = piece * 8, c = piece >> 3, slot = piece & 7;
    const bool ok = which == 0 ? (t0 + BM < p.Tv_out) : (t0 > 0);                // which 0: row m0 + 64 (LDS row 64), 1: row m0 - 1 (LDS row 65); frames that exist
    const long srow = which == 0 ? (long)m0 + BM : (long)m0 - 1;
    const bool first = ch < c0;
    const bf16_t* src = first ? (pl ? sg.a0_lo : sg.a0_hi) : (pl ? sg.a1_lo : sg.a1_hi);
    if (ok) hv = *reinterpret_cast<const uint4*>(src + (size_t)srow * (first ? c0 : c1) + (first ? ch : ch - c0));
    h_dst = pl * G::A_PL + c * CHP + (BM + which) * 128 + ((slot ^ swz(BM + which)) << 4);
  }
  if (wave == 0) glds4(p.bias ? (const void*)(p.bias + n0 + lane) : (const void*)p.zero_page, (unsigned)(size_t)s_bias);
  {
    // wave w brings rows 8w .. 8w + 7 of every chunk, both planes: lane = (row, 16-byte slot), the source chunk is swizzled
    const int row = wave * 8 + (lane >> 3), slot = lane & 7;
    const int sc8 = (slot ^ swz(row)) << 3;
#pragma unroll
    for (int c = 0; c < G::NCH; ++c) {
      const int cb = c * 64;
      const bool first = cb < c0;
      const size_t e = (size_t)min(m0 + row, m_last) * (first ? c0 : c1) + (first ? cb : cb - c0) + sc8;
      const unsigned dst = a_base + (unsigned)(c * CHP + wave * 1024);
      glds16((first ? sg.a0_hi : sg.a1_hi) + e, dst);
      glds16((first ? sg.a0_lo : sg.a1_lo) + e, dst + G::A_PL);
    }
  }
  // weights of this wave: fragment nf = n0 / 32 + cf, k-steps [kq U, kq U + U); unit j = k-step kq U + j (1 KiB per plane)
  const size_t w_e0 = ((size_t)((n0 >> 5) + cf) * G::KS + (size_t)kq * G::U) * 512 + (size_t)lane * 8;
  const bf16_t* const wh = p.wf_hi + w_e0;
  const bf16_t* const wl = p.wf_lo + w_e0;
  auto load_unit = [&](int j) {
    BFrag f;
    f.h = *reinterpret_cast<const bf16x8*>(wh + (size_t)j * 512);
    f.l = *reinterpret_cast<const bf16x8*>(wl + (size_t)j * 512);
    return f;
  };
  BFrag bq[G::DEPTH];
#pragma unroll
  for (int j = 0; j < G::DEPTH; ++j) bq[j] = load_unit(j);
  __builtin_amdgcn_sched_barrier(0);
  if (h_dst >= 0) *reinterpret_cast<uint4*>(smem + h_dst) = hv;
  DV_C3TRACE(1);
  wait_vmcnt<2 * G::DEPTH>();                        // everything older than the weight units: the tile's rows have landed
  {
    // padded row space: the first frame that does not exist (Tv_out) is read by the +1 tap of the last one that does and must be
    // zeros (nothing else of the padding is ever used: the outputs of rows that do not exist are discarded).  Its producers write
    // whatever their GroupNorm made of it, so the wave that brought that row clears it - behind ITS wait, ahead of the barrier.
    const int zr = p.Tv_out - t0;                    // tile row of that frame (row 64 is the halo row: `ok` above)
    if (zr < BM && (zr >> 3) == wave && (lane >> 3) == (zr & 7)) {
#pragma unroll
      for (int c = 0; c < G::NCH; ++c) {
        *reinterpret_cast<uint4*>(smem + c * CHP + wave * 1024 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(smem + G::A_PL + c * CHP + wave * 1024 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  }
  __syncthreads();
  DV_C3TRACE(2);

  // ---- k-loop: acc[rf] += W[fragment][k-step] x A^T[row fragment rf][k-step], no barrier ----
  // LDS row of (row fragment rf, tap): frame rf * 32 + l31 + tap - 1 of the tile; -1 is LDS row 65, 64 is LDS row 64
  int rowb[2][3], swl[2][3];
#pragma unroll
  for (int rf = 0; rf < 2; ++rf)
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
      const int idx = rf * 32 + l31 + tap - 1, row = idx < 0 ? BM + 1 : idx;
      rowb[rf][tap] = row * 128;
      swl[rf][tap] = swz(row) ^ lh;                  // (the 16-byte slot of k-step cs, half lh, is ((cs & 3) * 2 + lh) ^ swz(row))
    }
  f32x16 acc[2];
#pragma unroll
  for (int rf = 0; rf < 2; ++rf)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rf][r] = 0.f;
  auto run = [&](auto kq_tag) __attribute__((always_inline)) {
    constexpr int KQ = decltype(kq_tag)::value;
    auto read_a = [&](int u, bf16x8 (&h)[2], bf16x8 (&l)[2]) {
      const int ks = KQ * G::U + u, tap = ks / G::KPT, cs = ks - tap * G::KPT;
#pragma unroll
      for (int rf = 0; rf < 2; ++rf) {
        const int off = (cs >> 2) * CHP + rowb[rf][tap] + ((((cs & 3) << 1) ^ swl[rf][tap]) << 4);
        h[rf] = *reinterpret_cast<const bf16x8*>(smem + off);
        l[rf] = *reinterpret_cast<const bf16x8*>(smem + G::A_PL + off);
      }
    };
    bf16x8 ah[2][2], al[2][2];
    read_a(0, ah[0], al[0]);
#pragma unroll
    for (int u = 0; u < G::U; ++u) {
      const int cur = u & 1;
      if (u + 1 < G::U) read_a(u + 1, ah[cur ^ 1], al[cur ^ 1]);
      const BFrag f = bq[u % G::DEPTH];
      // (consecutive MFMAs alternate accumulators)
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, al[cur][0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, al[cur][1], acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.l, ah[cur][0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.l, ah[cur][1], acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, ah[cur][0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, ah[cur][1], acc[1], 0, 0, 0);
      // pinned: without the scheduling barriers hipcc sinks every prefetch down to its use (load -> vmcnt(0) -> MFMA)
      __builtin_amdgcn_sched_barrier(0);
      if (u + G::DEPTH < G::U) bq[u % G::DEPTH] = load_unit(u + G::DEPTH);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (kq == 0) run(std::integral_constant<int, 0>{});
  else if (kq == 1) run(std::integral_constant<int, 1>{});
  else if (kq == 2) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 3>{});
  DV_C3TRACE(3);

  conv3_finish(p, smem, s_bias, s_gnx, acc, rpre, tl, tid, lane, wave, CIN);
}

// ---------------------------------------------------------------------------------------
// Upsample2D (reference resnet.py:138-187: nearest x2, then a 3-tap convolution): the same resident-operand scheme on a tile of
// 128 OUTPUT rows.  Output frame t reads source frames (t + tap - 1) >> 1, so 128 output rows touch 64 + 2 source rows - the LDS
// image of k_conv3 exactly - and every weight fragment a wave loads is multiplied with FOUR row fragments: the k-loop needs
// 21 B/clk of weights per CU and runs at the MFMA rate.  8 waves = 2 column fragments x 4 k-quarters as above; after the exchange
// every wave finishes one whole 32 x 32 fragment (the steps of gemm_tile.h's full-fragment epilogue).  These were the two most
// expensive launches of the forward on k_gemm (128 x 64 tiles whose row gather reads every source row twice).
// ---------------------------------------------------------------------------------------
namespace {
constexpr int BMU = 128, RFU = 4;
template <int CIN>
struct ConvGeomU {
  static constexpr int NCH = CIN / 64, KPT = CIN / 16, KS = 3 * KPT, U = KS / 4;
  static constexpr int DEPTH = U < 6 ? U : 6;        // weight units in flight per wave (four row fragments of operands are live too)
  static constexpr int A_PL = NCH * CHP;
  static constexpr int RED = 2 * 4 * RFU * 4 * 64 * 16;   // k-quarter exchange: [column fragment][quarter][row fragment][4 column groups][64 lanes] float4
  static constexpr int SMEM = 2 * A_PL > RED ? 2 * A_PL : RED;
  static_assert(CIN % 64 == 0 && KS % 4 == 0 && SMEM + 4096 <= 160 * 1024, "geometry");
};
}  // namespace

template <int CIN>
__global__ __launch_bounds__(NT) void k_conv3u(const GemmParams p) {
  using G = ConvGeomU<CIN>;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  __shared__ __attribute__((aligned(16))) float s_bias[BN];
  __shared__ GnxShared<BN> s_gnx;
  asm volatile("" ::"s"(p.seg[0].a0_hi), "s"(p.seg[1].a0_hi), "s"(p.seg[1].pad), "s"(p.T_in), "s"(p.w_hi), "s"(p.M), "s"(p.res),
               "s"(p.out_hi), "s"(p.zero_page), "s"(p.ln_u), "s"(p.gnx.xchg), "s"(p.gnx.y_hi), "s"(p.xcd_n), "s"(p.xcd_inv_tn), "s"(p.wf_lo));
  DV_C3TRACE(0);
  // tile: XCD x takes a band of row tiles (all their column tiles: an L2 fetches its rows of the source once); row tiles never span
  // two utterances (C3Tile: the last one of an utterance may be short)
  C3Tile tl;
  {
    const int n_tiles_n = p.N / BN, nwg = gridDim.x;
    int bid = blockIdx.x;
    if (p.xcd_n == 0) {                              // (xcd_n < 0: plain order)
      const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tmi = bid / n_tiles_n;
    tl.n0 = (bid - tmi * n_tiles_n) * BN; tl.ksel = 0;
    tl.bq = p.c3_tu_magic ? (int)__umulhi((unsigned)tmi, p.c3_tu_magic) : tmi;
    tl.t0 = (tmi - tl.bq * p.c3_tu) * BMU;
    tl.m0 = tl.bq * p.T_out + tl.t0;
    tl.rows = min(BMU, p.T_out - tl.t0);
  }
  const int m0 = tl.m0, n0 = tl.n0, t0 = tl.t0;      // t0: output frame of the tile's first row inside its utterance
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int cf = wave & 1, kq = wave >> 1;           // k-loop role: column fragment, k-quarter; epilogue role: fragment (row kq, column cf)
  const unsigned a_base = (unsigned)(size_t)smem;
  const GemmSeg& sg = p.seg[0];
  const int c0 = sg.c0, c1 = sg.c1;
  const int ts0 = t0 >> 1;                           // source frame of output frame t0
  const int ms0 = tl.bq * p.T_in + ts0;              // ... its row (the pitches of the two levels are independent)
  const int ms_last = p.B * p.T_in - 1;              // (a short last tile: the source rows behind it are read, never used)

  // ---- requests, oldest first: halo rows, bias, the tile's 64 source rows, first weights ----
  constexpr int HITEMS = CIN / 2;
  uint4 hv = make_uint4(0u, 0u, 0u, 0u);
  int h_dst = -1;
  if (tid < HITEMS) {
    const int pl = tid / (CIN / 4), rem = tid - pl * (CIN / 4), which = rem / (CIN / 8), piece = rem - which * (CIN / 8);
    const int ch = piece * 8, c = piece >> 3, slot = piece & 7;
    const bool ok = which == 0 ? (ts0 + 64 < p.Tv_in) : (t0 > 0);               // which 0: source row ms0 + 64 (LDS row 64), 1: row ms0 - 1 (LDS row 65); frames that exist
    const long srow = which == 0 ? (long)ms0 + 64 : (long)ms0 - 1;
    const bool first = ch < c0;
    const bf16_t* src = first ? (pl ? sg.a0_lo : sg.a0_hi) : (pl ? sg.a1_lo : sg.a1_hi);
    if (ok) hv = *reinterpret_cast<const uint4*>(src + (size_t)srow * (first ? c0 : c1) + (first ? ch : ch - c0));
    h_dst = pl * G::A_PL + c * CHP + (64 + which) * 128 + ((slot ^ swz(64 + which)) << 4);
  }
  if (wave == 0) glds4(p.bias ? (const void*)(p.bias + n0 + lane) : (const void*)p.zero_page, (unsigned)(size_t)s_bias);
  {
    const int row = wave * 8 + (lane >> 3), slot = lane & 7;
    const int sc8 = (slot ^ swz(row)) << 3;
#pragma unroll
    for (int c = 0; c < G::NCH; ++c) {
      const int cb = c * 64;
      const bool first = cb < c0;
      const size_t e = (size_t)min(ms0 + row, ms_last) * (first ? c0 : c1) + (first ? cb : cb - c0) + sc8;
      const unsigned dst = a_base + (unsigned)(c * CHP + wave * 1024);
      glds16((first ? sg.a0_hi : sg.a1_hi) + e, dst);
      glds16((first ? sg.a0_lo : sg.a1_lo) + e, dst + G::A_PL);
    }
  }
  const size_t w_e0 = ((size_t)((n0 >> 5) + cf) * G::KS + (size_t)kq * G::U) * 512 + (size_t)lane * 8;
  const bf16_t* const wh = p.wf_hi + w_e0;
  const bf16_t* const wl = p.wf_lo + w_e0;
  auto load_unit = [&](int j) {
    BFrag f;
    f.h = *reinterpret_cast<const bf16x8*>(wh + (size_t)j * 512);
    f.l = *reinterpret_cast<const bf16x8*>(wl + (size_t)j * 512);
    return f;
  };
  BFrag bq[G::DEPTH];
#pragma unroll
  for (int j = 0; j < G::DEPTH; ++j) bq[j] = load_unit(j);
  __builtin_amdgcn_sched_barrier(0);
  if (h_dst >= 0) *reinterpret_cast<uint4*>(smem + h_dst) = hv;
  DV_C3TRACE(1);
  wait_vmcnt<2 * G::DEPTH>();                        // everything older than the weight units: the source rows have landed
  {
    // padded row space (see k_conv3): the first source frame that does not exist is cleared by the wave that brought it
    const int zr = p.Tv_in - ts0;
    if (zr < 64 && (zr >> 3) == wave && (lane >> 3) == (zr & 7)) {
#pragma unroll
      for (int c = 0; c < G::NCH; ++c) {
        *reinterpret_cast<uint4*>(smem + c * CHP + wave * 1024 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(smem + G::A_PL + c * CHP + wave * 1024 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  }
  __syncthreads();
  DV_C3TRACE(2);

  // ---- k-loop: acc[rf] += W[fragment][k-step] x A^T[row fragment rf][k-step], no barrier ----
  // LDS row of (row fragment rf, tap): source frame (rf * 32 + l31 + tap - 1) >> 1 of the tile; -1 is LDS row 65, 64 is LDS row 64
  int rowb[RFU][3], swl[RFU][3];
#pragma unroll
  for (int rf = 0; rf < RFU; ++rf)
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
      const int idx = (rf * 32 + l31 + tap - 1) >> 1, row = idx < 0 ? 65 : idx;
      rowb[rf][tap] = row * 128;
      swl[rf][tap] = swz(row) ^ lh;
    }
  f32x16 acc[RFU];
#pragma unroll
  for (int rf = 0; rf < RFU; ++rf)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rf][r] = 0.f;
  auto run = [&](auto kq_tag) __attribute__((always_inline)) {
    constexpr int KQ = decltype(kq_tag)::value;
    auto read_a = [&](int u, bf16x8 (&h)[RFU], bf16x8 (&l)[RFU]) {
      const int ks = KQ * G::U + u, tap = ks / G::KPT, cs = ks - tap * G::KPT;
#pragma unroll
      for (int rf = 0; rf < RFU; ++rf) {
        const int off = (cs >> 2) * CHP + rowb[rf][tap] + ((((cs & 3) << 1) ^ swl[rf][tap]) << 4);
        h[rf] = *reinterpret_cast<const bf16x8*>(smem + off);
        l[rf] = *reinterpret_cast<const bf16x8*>(smem + G::A_PL + off);
      }
    };
    bf16x8 ah[2][RFU], al[2][RFU];
    read_a(0, ah[0], al[0]);
#pragma unroll
    for (int u = 0; u < G::U; ++u) {
      const int cur = u & 1;
      if (u + 1 < G::U) read_a(u + 1, ah[cur ^ 1], al[cur ^ 1]);
      const BFrag f = bq[u % G::DEPTH];
#pragma unroll
      for (int rf = 0; rf < RFU; ++rf) acc[rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, al[cur][rf], acc[rf], 0, 0, 0);
#pragma unroll
      for (int rf = 0; rf < RFU; ++rf) acc[rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.l, ah[cur][rf], acc[rf], 0, 0, 0);
#pragma unroll
      for (int rf = 0; rf < RFU; ++rf) acc[rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, ah[cur][rf], acc[rf], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (u + G::DEPTH < G::U) bq[u % G::DEPTH] = load_unit(u + G::DEPTH);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (kq == 0) run(std::integral_constant<int, 0>{});
  else if (kq == 1) run(std::integral_constant<int, 1>{});
  else if (kq == 2) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 3>{});
  DV_C3TRACE(3);

  // ---- the four k-quarters are added through LDS; wave (kq, cf) keeps the whole fragment (row fragment kq, column fragment cf) ----
  __syncthreads();                                   // every wave is done reading the resident operand
  float4* const red4 = reinterpret_cast<float4*>(smem);
#pragma unroll
  for (int rf = 0; rf < RFU; ++rf)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      red4[((((cf * 4 + kq) * RFU + rf) * 4 + g) << 6) + lane] = make_float4(acc[rf][4 * g], acc[rf][4 * g + 1], acc[rf][4 * g + 2], acc[rf][4 * g + 3]);
  __syncthreads();
  const int e_wm = kq, e_wn = cf;
  float vv[16];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float4 s = red4[((((e_wn * 4 + 0) * RFU + e_wm) * 4 + g) << 6) + lane];
#pragma unroll
    for (int k = 1; k < 4; ++k) {                    // (quarters in order: deterministic)
      const float4 v = red4[((((e_wn * 4 + k) * RFU + e_wm) * 4 + g) << 6) + lane];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    vv[4 * g] = s.x; vv[4 * g + 1] = s.y; vv[4 * g + 2] = s.z; vv[4 * g + 3] = s.w;
  }
  DV_C3TRACE(4);

  // ---- epilogue of the fragment: vv[4g + e] = column n0 + 32 e_wn + 8g + 4lh + e of row m0 + 32 e_wm + l31 ----
  // (padded row space: a fragment beyond the tile's rows belongs to the next utterance and is not touched; rows that do not exist
  // are stored as zeros and kept out of the statistics - conv3_finish)
  const bool gnx_h = p.gnx.xchg != nullptr;
  const int ncol0 = n0 + e_wn * 32;
  const int m = m0 + e_wm * 32 + l31, mrow0 = m0 + e_wm * 32;
  const bool live = e_wm * 32 < tl.rows;             // (wave-uniform)
  const bool padded = p.Tv_out != p.T_out;
  const int tfr = t0 + e_wm * 32;
  const bool m_ok = tfr + l31 < p.Tv_out;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 b4 = *reinterpret_cast<const float4*>(s_bias + e_wn * 32 + 4 * lh + 8 * g);
    vv[4 * g] += b4.x; vv[4 * g + 1] += b4.y; vv[4 * g + 2] += b4.z; vv[4 * g + 3] += b4.w;
  }
  if (p.epi == EPI_RESIDUAL) {
    const float* rp = p.res + (size_t)min(m, p.M - 1) * p.ldres + ncol0 + 4 * lh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 a = *reinterpret_cast<const float4*>(rp + 8 * g);
      vv[4 * g] += a.x; vv[4 * g + 1] += a.y; vv[4 * g + 2] += a.z; vv[4 * g + 3] += a.w;
    }
  }
  if (padded) {
#pragma unroll
    for (int r = 0; r < 16; ++r) vv[r] = m_ok ? vv[r] : 0.f;
  }
  if (live) {
    const size_t ob = (size_t)m * p.ldo + ncol0;
    if (p.out) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        dv_st16(p.out + ob + 4 * lh + 8 * g, make_float4(vv[4 * g], vv[4 * g + 1], vv[4 * g + 2], vv[4 * g + 3]));
    }
    if (p.out_hi) store_planes16(p.out_hi, p.out_lo, ob, lh, vv);
  }
  DV_C3TRACE(5);
  if (p.stats16 && live) {
    // per (32-row, 16-column) block: (sum, squared deviations from the block's own mean); registers 0-7 / 8-15 are the fragment's
    // first / second 16 columns
    float a1[2] = {0.f, 0.f}, a2[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; ++r) a1[r >> 3] += vv[r];
    a1[0] = wave_sum64(a1[0]); a1[1] = wave_sum64(a1[1]);
    const float inv = padded ? 1.0f / (float)(16 * min(32, p.Tv_out - tfr)) : (1.0f / 512.0f);
    const float mb[2] = {a1[0] * inv, a1[1] * inv};
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float dv = (!padded || m_ok) ? vv[r] - mb[r >> 3] : 0.f; a2[r >> 3] = fmaf(dv, dv, a2[r >> 3]); }
    a2[0] = wave_sum64(a2[0]); a2[1] = wave_sum64(a2[1]);
    if (lane < 2) {
      const size_t e = (size_t)(mrow0 >> 5) * (p.N >> 4) + (ncol0 >> 4) + lane;
      const float2 val = make_float2(lane ? a1[1] : a1[0], lane ? a2[1] : a2[0]);
      reinterpret_cast<float2*>(p.stats16)[e] = val;
      if (gnx_h)
        __hip_atomic_store(p.gnx.xchg + e, (unsigned long long)__float_as_uint(val.x) | ((unsigned long long)__float_as_uint(val.y) << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  DV_C3TRACE(6);
  if (gnx_h) {
    GnxTile t;
    t.M = p.M; t.N = p.N; t.T_out = p.T_out; t.Tv_out = p.Tv_out; t.m0 = m0; t.n0 = n0; t.bm = tl.rows; t.bn = BN;
    t.bq = tl.bq;
    gnx_finish_table<BN>(p.gnx, t, s_gnx, tid, lane, wave, NWV, [&](int) {});
    DV_C3TRACE(7);
    float y[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cl = e_wn * 32 + 4 * lh + 8 * g;
      const float4 sa = *reinterpret_cast<const float4*>(s_gnx.gA + cl);
      const float4 sb = *reinterpret_cast<const float4*>(s_gnx.gB + cl);
      y[4 * g] = fmaf(vv[4 * g], sa.x, sb.x); y[4 * g + 1] = fmaf(vv[4 * g + 1], sa.y, sb.y);
      y[4 * g + 2] = fmaf(vv[4 * g + 2], sa.z, sb.z); y[4 * g + 3] = fmaf(vv[4 * g + 3], sa.w, sb.w);
    }
    if (p.gnx.silu) {
#pragma unroll
      for (int r = 0; r < 16; ++r) y[r] = y[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-y[r]));
    }
    if (padded) {
#pragma unroll
      for (int r = 0; r < 16; ++r) y[r] = m_ok ? y[r] : 0.f;
    }
    if (live) store_planes16(p.gnx.y_hi, p.gnx.y_lo, (size_t)m * p.N + ncol0, lh, y);
  }
  DV_C3TRACE(8);
}

// ---------------------------------------------------------------------------------------
// Any input width (multiples of 128 channels up to 1024), and the ResnetBlock2D's second convolution WITH its folded 1x1 shortcut
// (a second, one-tap K-segment over the block's raw input - reference resnet.py:627-641): the same tile with the rows STREAMED
// through a ring of 64-channel chunks.
//   K order  : chunk-major.  Three-tap segment: chunk -> tap -> the chunk's four 16-deep k-steps, wave quarter q multiplies
//              k-step q of every tap (three weight units per chunk and wave: the same 16 channels of the rows at offsets
//              -1 / 0 / +1).  One-tap segment: two chunks per step, k-step q of each (two units).
//   ring     : RS slots of [2 planes][64 rows][128 B]; the two halo rows of every three-tap chunk are brought once, at the head,
//              into their own 4 KiB per plane.
//   producer : a NINTH wave issues every LDS-DMA (16 per chunk) and nothing else: its vector-memory queue holds DMAs only, so
//              "all but the two youngest chunks have landed" is vmcnt(32) whatever the segment mix, and the compute waves' queues
//              hold nothing but compiler-visible weight loads (hipcc counts those exactly; it cannot see a DMA).  [First version:
//              every compute wave issued two DMAs per chunk with hand-counted waits - the same speed for one segment, and no
//              constant that fits two.]
//   barriers : one per step.  Barrier s says: the chunks of step s + 1 are in LDS (the producer waited), and every wave is done
//              with step s - 1 (the producer refills its slots).  Operand fragments are read one unit ahead - those of step
//              s + 1's first unit under step s's last MFMAs - so nobody arrives behind a barrier with nothing to multiply.
// Measured (profiles/r05_conv3_*): ~1850 cycles per three-tap chunk against 1152 of MFMA and ~3500 on k_gemm's ring - the per-CU
// vector-memory path carries 48 KiB of weights AND 17 KiB of rows per chunk here (ablation: 1230 cycles with neither, 1510 / 1400
// with one of them); the resident kernel above streams the weights only and runs at the MFMA rate, so it keeps the widths it fits.
// ---------------------------------------------------------------------------------------
#ifndef DV_C3_EXP
// development knob (trace experiments on the streaming k-loop, WRONG results): bit 0 = no operand DMA inside the loop, bit 1 = no
// weight loads inside the loop.  0 in every shipped build
#define DV_C3_EXP 0
#endif
namespace {
constexpr int RS = 8;                                // ring slots
constexpr int SLOT_PL = BM * 128, SLOT = 2 * SLOT_PL;   // one plane / both planes of a 64-channel chunk: 16 KiB per slot
constexpr int S_HALO = RS * SLOT;                    // halo rows: [16 chunks][row m0 + 64 | row m0 - 1][128 B], the lo plane SLOT_PL further on
constexpr int S_ZERO = S_HALO + 4096;                // 128 bytes of zeros (the lo plane's SLOT_PL further on, like everything else)
constexpr int S_TOTAL = S_HALO + SLOT_PL + 4096 + 256;
constexpr int DWS = 6;                               // weight units in flight per wave
constexpr int NT_S = NT + 64;                        // + the producer wave
static_assert(S_TOTAL + 4096 <= 160 * 1024 && RED_BYTES <= RS * SLOT && RS == 8, "ring geometry");
}  // namespace

__global__ __launch_bounds__(NT_S) void k_conv3s(const GemmParams p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  __shared__ __attribute__((aligned(16))) float s_bias[BN];
  __shared__ GnxShared<BN> s_gnx;
  asm volatile("" ::"s"(p.seg[0].a0_hi), "s"(p.seg[1].a0_hi), "s"(p.seg[1].pad), "s"(p.T_in), "s"(p.w_hi), "s"(p.M), "s"(p.res),
               "s"(p.out_hi), "s"(p.zero_page), "s"(p.ln_u), "s"(p.gnx.xchg), "s"(p.gnx.y_hi), "s"(p.xcd_n), "s"(p.xcd_inv_tn), "s"(p.wf_lo));
  const GemmSeg& sg = p.seg[0];
  const GemmSeg& sh = p.seg[1];
  const int c0 = sg.c0, c1 = sg.c1, CIN = c0 + c1;
  const int d0 = p.nseg > 1 ? sh.c0 : 0, d1 = p.nseg > 1 ? sh.c1 : 0;   // the one-tap segment's channels (0: none)
  DV_C3TRACE(0);
  C3Tile tl;
  conv3_tile(p, tl, BM);
  const int m0 = tl.m0, n0 = tl.n0, ksel = tl.ksel, t0 = tl.t0;
  const int rmax = min(BM - 1, p.M - 1 - m0);        // last tile row that is inside the tensor (a short last tile: the rows behind it are read, never used)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned a_base = (unsigned)(size_t)smem;
  const int NCH_ALL = CIN >> 6, KPT = CIN >> 4;      // three-tap chunks of the whole contraction (even); 16-deep k-steps per tap
  const int NC2_ALL = (d0 + d1) >> 6;                // one-tap chunks
  // this workgroup's chunks: all of them, or - fused split-K pair - one half of either list (three-tap halves stay even: the
  // parities of the weight ring and of the fragment buffers below rely on it)
  const bool half = p.sk_mode == 3;
  const int h1 = half ? (((NCH_ALL >> 1) + 1) & ~1) : NCH_ALL, h2 = half ? ((NC2_ALL + 1) >> 1) : NC2_ALL;
  const int CA = ksel ? h1 : 0, NCH = ksel ? NCH_ALL - h1 : h1;      // three-tap chunks CA .. CA + NCH - 1 (list positions 0 .. NCH - 1)
  const int JA = ksel ? h2 : 0, NC2 = ksel ? NC2_ALL - h2 : h2;      // one-tap chunks JA .. JA + NC2 - 1 (list positions NCH ..)
  const int NS2 = (NC2 + 1) >> 1;                    // one-tap steps (two chunks each; the last may hold one)
  const int NCT = NCH + NC2, NSTEP = NCH + NS2;

  if (wave == NWV) {
    // ================= producer =================
    glds4(p.bias ? (const void*)(p.bias + n0 + lane) : (const void*)p.zero_page, (unsigned)(size_t)s_bias);
    // a chunk: 8 instructions per plane, instruction r8 = rows 8 r8 .. 8 r8 + 7 (lane = (row, 16-byte slot); the source chunk is swizzled)
    const int lr = lane >> 3;
    const unsigned sw_e = (unsigned)(((lane & 7) ^ swz(lr)) << 4), sw_o = (unsigned)(((lane & 7) ^ swz(8 + lr)) << 4);
    auto issue_chunk = [&](int cc) {
      // chunk cc of the list [three-tap segment's chunks | one-tap segment's chunks] -> slot cc % RS
      const bool seg0 = cc < NCH;
      const int cb = (seg0 ? CA + cc : JA + cc - NCH) * 64;
      const int e0 = seg0 ? c0 : d0;
      const bool first = cb < e0;
      const int ld = first ? e0 : (seg0 ? c1 : d1);
      const bf16_t* bh = seg0 ? (first ? sg.a0_hi : sg.a1_hi) : (first ? sh.a0_hi : sh.a1_hi);
      const bf16_t* bl = seg0 ? (first ? sg.a0_lo : sg.a1_lo) : (first ? sh.a0_lo : sh.a1_lo);
      const size_t col = (size_t)(first ? cb : cb - e0);
      const unsigned dst = a_base + (unsigned)((cc & (RS - 1)) * SLOT);
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        const unsigned vo = (unsigned)((m0 + min(lr + 8 * r8, rmax)) * ld * 2) + ((r8 & 1) ? sw_o : sw_e);
        glds16_s(bh + col, vo, dst + r8 * 1024);
        glds16_s(bl + col, vo, dst + SLOT_PL + r8 * 1024);
      }
    };
    // (chunks 0 and 1 - steps 0 and 1 - and the halo rows are brought by the eight compute waves together, 5 instructions each:
    // one wave issuing the first 73 instructions held barrier 0 back by ~4 k cycles)
    int next = 2;
    for (; next < min(4, NCT); ++next) issue_chunk(next);
    wait_vmcnt<32>();                                // (the bias; chunks 2 and 3 are not needed before barrier 1)
    __builtin_amdgcn_s_barrier();                    // barrier 0
    int done = 0;                                    // chunks of the steps every wave is through
    for (int s2 = 1; s2 < NSTEP; ++s2) {
      // (at most two chunks per step: the ring fills over the first steps instead of holding barrier 1 back)
      if (!(DV_C3_EXP & 1)) {
        const int lim = min(min(NCT, done + RS), next + 2);
        for (; next < lim; ++next) issue_chunk(next);
      }
      // the chunks of step s2 + 1 must be in LDS: everything but the chunks issued behind its last one (at most three: the queue
      // holds 63 operations) may still be under way
      {
        const int s3 = s2 + 1;
        const int need = s3 < NCH ? s3 : min(NCH + 2 * (s3 - NCH) + 1, NCT - 1);
        const int k = next - 1 - need;
        if (k >= 3) wait_vmcnt<48>();
        else if (k == 2) wait_vmcnt<32>();
        else if (k == 1) wait_vmcnt<16>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();                  // barrier s2: everybody is through step s2 - 1
      done += (s2 - 1) < NCH ? 1 : 2;
    }
    return;
  }

  // ================= compute waves =================
  const int l31 = lane & 31, lh = lane >> 5;
  const int cf = wave & 1, kq = wave >> 1;           // k-loop role: column fragment, k-quarter (k-step kq of every 64-channel chunk and tap)
  const int e_wn = cf, e_wm = (wave >> 1) & 1, e_half = wave >> 2;   // epilogue role (conv3_finish): this wave's residual rows
  float rpre[8];
  if (p.epi == EPI_RESIDUAL) {
    const float* rp = p.res + (size_t)(m0 + min(e_wm * 32 + l31, rmax)) * p.ldres + n0 + e_wn * 32 + e_half * 16 + 4 * lh;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float4 a = *reinterpret_cast<const float4*>(rp + 8 * g);
      rpre[4 * g] = a.x; rpre[4 * g + 1] = a.y; rpre[4 * g + 2] = a.z; rpre[4 * g + 3] = a.w;
    }
  }
  {
    // halo rows: one instruction per wave = one plane of four chunks, lane = (chunk, row m0 + 64 | row m0 - 1, 16-byte slot); rows
    // beyond the utterance's ends and chunks beyond the last come from the zero page
    const int pl = wave & 1, c = (wave >> 1) * 4 + (lane >> 4), which = (lane >> 3) & 1, slot = lane & 7;
    const bool ok = c < NCH_ALL && (which == 0 ? (t0 + BM < p.Tv_out) : (t0 > 0));   // (frames that exist)
    const long srow = which == 0 ? (long)m0 + BM : (long)m0 - 1;
    const int ch = c * 64 + slot * 8;
    const bool first = ch < c0;
    const bf16_t* src = first ? (pl ? sg.a0_lo : sg.a0_hi) : (pl ? sg.a1_lo : sg.a1_hi);
    const void* g = ok ? (const void*)(src + (size_t)srow * (first ? c0 : c1) + (first ? ch : ch - c0)) : (const void*)p.zero_page;
    glds16(g, a_base + (unsigned)(S_HALO + pl * SLOT_PL + (wave >> 1) * 1024));
    // (a row of zeros per plane: what the lanes of the first frame that does not exist read - see rb / sm below)
    if (wave == 0 && lane < 16) *reinterpret_cast<uint4*>(smem + S_ZERO + (lane >> 3) * SLOT_PL + (lane & 7) * 16) = make_uint4(0u, 0u, 0u, 0u);
    // chunks 0 and 1 (both of the three-tap segment: NCH >= 2): wave w brings rows 8w .. 8w + 7, both planes
    const int d_row = wave * 8 + (lane >> 3);
    const unsigned d_sc = (unsigned)(((lane & 7) ^ swz(d_row)) << 4);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int cb = (CA + cc) * 64;
      const bool f0 = cb < c0;
      const unsigned vo = (unsigned)((size_t)(m0 + min(d_row, rmax)) * (f0 ? c0 : c1) * 2) + d_sc;
      const unsigned dst = a_base + (unsigned)(cc * SLOT + wave * 1024);
      glds16_s((f0 ? sg.a0_hi : sg.a1_hi) + (f0 ? cb : cb - c0), vo, dst);
      glds16_s((f0 ? sg.a0_lo : sg.a1_lo) + (f0 ? cb : cb - c0), vo, dst + SLOT_PL);
    }
  }
  // weights of this wave: fragment nf = n0 / 32 + cf of the packed [N][3 C_in | one-tap channels] rows, k-steps of quarter kq.
  // Unit list: (chunk c, tap t) = k-step t KPT + 4 c + kq for the three-tap chunks (three per chunk), then k-step 3 KPT + 4 j + kq
  // for one-tap chunk j (one per chunk).
  const int KSW = 3 * KPT + 4 * NC2_ALL;             // k-steps per weight row
  const bf16_t* const wh = p.wf_hi + ((size_t)((n0 >> 5) + cf) * KSW + kq) * 512 + (size_t)lane * 8;
  const bf16_t* const wl = p.wf_lo + ((size_t)((n0 >> 5) + cf) * KSW + kq) * 512 + (size_t)lane * 8;
  auto load_ks = [&](int ks) {
    BFrag f;
    f.h = *reinterpret_cast<const bf16x8*>(wh + (size_t)ks * 512);
    f.l = *reinterpret_cast<const bf16x8*>(wl + (size_t)ks * 512);
    return f;
  };
  // unit (c, t) of the three-tap list, continued into the one-tap list behind its end (past the very end: the last unit again -
  // every refill is unconditional, hipcc's wait counts stay exact)
  auto unit_ks = [&](int c, int t) {
    if (c < NCH) return t * KPT + 4 * (CA + c);
    const int j = 3 * (c - NCH) + t;
    return NC2 > 0 ? 3 * KPT + 4 * (JA + min(j, NC2 - 1)) : 2 * KPT + 4 * (CA + NCH - 1);
  };
  BFrag bq[DWS];
#pragma unroll
  for (int j = 0; j < DWS; ++j) bq[j] = load_ks(unit_ks(j / 3, j % 3));
  __builtin_amdgcn_sched_barrier(0);
  DV_C3TRACE(1);

  // operand reads: LDS byte offset of (row fragment rf, tap) inside a slot for this lane - frame rf * 32 + l31 + tap - 1, k-step kq
  // of the chunk, half lh - or, for the one lane pair whose frame is the row before / behind the tile, inside the halo area
  // Padded row space: the first frame that does not exist (tile row zr = Tv_out - t0, when it lies inside the tile) is read by the
  // +1 tap of the last one that does and must be zeros - whatever its producers wrote there: the lane that would read it reads the
  // row of zeros instead (sm = 0: no slot / halo offset is added).  Nothing else of the padding is ever used.
  const int zr = p.Tv_out - t0;
  int rb[2][3], sm[2][3];
  bool hs[2][3];
#pragma unroll
  for (int rf = 0; rf < 2; ++rf)
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
      const int idx = rf * 32 + l31 + tap - 1, c16 = kq * 2 + lh;
      const bool zero = idx == zr && zr < BM;
      hs[rf][tap] = idx < 0 || idx >= BM;
      rb[rf][tap] = zero ? S_ZERO + (c16 << 4) : hs[rf][tap] ? S_HALO + (idx < 0 ? 128 : 0) + (c16 << 4) : idx * 128 + ((c16 ^ swz(idx)) << 4);
      sm[rf][tap] = zero ? 0 : -1;
    }
  f32x16 acc[2];
#pragma unroll
  for (int rf = 0; rf < 2; ++rf)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rf][r] = 0.f;
  // operand fragments (both row fragments, both planes) of tap t of three-tap chunk c / of one-tap chunk j (list position NCH + j)
  auto read_a = [&](int c, int t, bf16x8 (&h)[2], bf16x8 (&l)[2]) __attribute__((always_inline)) {
    const int s_slot = (c & (RS - 1)) * SLOT, s_halo = (CA + c) * 256;   // (c: list position)
#pragma unroll
    for (int rf = 0; rf < 2; ++rf) {
      // (only (rf 0, tap 0) and (rf 1, tap 2) have a halo lane pair: the select folds away elsewhere)
      const bool any_halo = (rf == 0 && t == 0) || (rf == 1 && t == 2);
      const int off = rb[rf][t] + (((any_halo && hs[rf][t]) ? s_halo : s_slot) & sm[rf][t]);
      h[rf] = *reinterpret_cast<const bf16x8*>(smem + off);
      l[rf] = *reinterpret_cast<const bf16x8*>(smem + off + SLOT_PL);
    }
  };
  auto mma = [&](const BFrag& f, bf16x8 (&ah)[2], bf16x8 (&al)[2]) __attribute__((always_inline)) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, al[0], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, al[1], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.l, ah[0], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.l, ah[1], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, ah[0], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.h, ah[1], acc[1], 0, 0, 0);
  };
  bf16x8 ahb[2][2], alb[2][2];                       // fragments of the unit at list position i sit in buffer i & 1
  wait_vmcnt<2 * DWS>();                             // this wave's part of chunks 0 / 1 and of the halo rows (everything older than the weight units)
  __builtin_amdgcn_s_barrier();                      // barrier 0: chunks 0 and 1, halo rows, bias are in LDS
  DV_C3TRACE(2);
  read_a(0, 0, ahb[0], alb[0]);

  // ---- three-tap chunks: step c = chunk c ----
  auto chunk = [&](const int c, auto par_tag) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_tag)::value;    // c & 1: units 3c .. 3c + 2 sit in bq[3 PAR ..]
    if (c > 0) __builtin_amdgcn_s_barrier();         // barrier c
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int cur = (PAR + t) & 1;
      if (t + 1 < 3) read_a(c, t + 1, ahb[cur ^ 1], alb[cur ^ 1]);
      else if (c + 1 < NCH) read_a(c + 1, 0, ahb[cur ^ 1], alb[cur ^ 1]);
      else read_a(c + 1, 1, ahb[cur ^ 1], alb[cur ^ 1]);           // the first one-tap chunk: list position NCH, centre tap (none: a slot nobody multiplies)
      mma(bq[3 * PAR + t], ahb[cur], alb[cur]);
      // pinned: without the scheduling barriers hipcc sinks every prefetch down to its use (load -> vmcnt(0) -> MFMA)
      __builtin_amdgcn_sched_barrier(0);
      if (!(DV_C3_EXP & 2)) bq[3 * PAR + t] = load_ks(unit_ks(c + 2, t));
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int c = 0; c < NCH; c += 2) {                 // (NCH is even: C_in is a multiple of 128)
    chunk(c, std::integral_constant<int, 0>{});
    chunk(c + 1, std::integral_constant<int, 1>{});
  }
  // ---- one-tap chunks: step = chunks 2 j2, 2 j2 + 1 (list positions NCH + ...: even NCH keeps the buffer / weight-ring parities) ----
  // (unit j sits in bq[j % 6]: three steps per round of the ring)
  auto pair = [&](const int j2, auto pos_tag) __attribute__((always_inline)) {
    constexpr int POS = decltype(pos_tag)::value;    // j2 % 3
    __builtin_amdgcn_s_barrier();                    // barrier NCH + j2
    const int j = 2 * j2;
    read_a(NCH + j + 1, 1, ahb[1], alb[1]);
    mma(bq[2 * POS], ahb[0], alb[0]);
    __builtin_amdgcn_sched_barrier(0);
    if (!(DV_C3_EXP & 2)) bq[2 * POS] = load_ks(3 * KPT + 4 * (JA + min(j + 6, NC2 - 1)));
    __builtin_amdgcn_sched_barrier(0);
    read_a(NCH + j + 2, 1, ahb[0], alb[0]);          // the next step's first chunk (behind the last: a slot nobody multiplies)
    if (j + 1 < NC2) mma(bq[2 * POS + 1], ahb[1], alb[1]);
    __builtin_amdgcn_sched_barrier(0);
    if (!(DV_C3_EXP & 2)) bq[2 * POS + 1] = load_ks(3 * KPT + 4 * (JA + min(j + 7, NC2 - 1)));
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int j2 = 0; j2 < NS2; j2 += 3) {
    pair(j2, std::integral_constant<int, 0>{});
    if (j2 + 1 < NS2) pair(j2 + 1, std::integral_constant<int, 1>{});
    if (j2 + 2 < NS2) pair(j2 + 2, std::integral_constant<int, 2>{});
  }
  DV_C3TRACE(3);
  conv3_finish(p, smem, s_bias, s_gnx, acc, rpre, tl, tid, lane, wave, CIN);
}

// ---- host side ----
static bool g_conv3_on = true;
static int g_conv3_stream = 512;                     // widths above this run the streaming kernel (DVITS_CONV3_STREAM=<channels>; 0: every width)
void conv3_env_refresh() {
  const char* e = getenv("DVITS_CONV3"); g_conv3_on = !(e && e[0] == '0');
  const char* e2 = getenv("DVITS_CONV3_STREAM"); g_conv3_stream = e2 ? atoi(e2) : 512;
}
static const bool g_conv3_env_once = [] { conv3_env_refresh(); return true; }();

// Shapes these kernels take (everything else stays with k_gemm): one 3-tap segment over 128 .. 1024 input channels in steps of 128
// (one tensor or the concatenation of two, each a multiple of 64 channels), stride 1, no resampling, a row pitch of whole 32-frame
// blocks (padded row spaces included: Tv_out <= T_out frames exist; the row tiles are laid out per utterance - C3Tile), the plain /
// residual epilogue without LayerNorm, column-slab statistics, ReLU or row mask.
bool gemm_conv3_shape_ok(const GemmParams& p) {
  if (!g_conv3_on || p.nseg < 1 || p.nseg > 2 || p.seg[0].taps != 3 || p.seg[0].pad != 1 || p.stride != 1 || p.up_mode != UP_NONE) return false;
  const int cin = p.seg[0].c0 + p.seg[0].c1;
  if (cin < 128 || cin > 1024 || cin % 128 != 0) return false;
  if (p.seg[0].c0 % 64 != 0 || p.seg[0].c1 % 64 != 0 || (p.seg[0].c1 > 0 && !p.seg[0].a1_hi)) return false;
  if (p.nseg == 2) {   // the folded 1x1 shortcut: one tap over the block's raw input (one tensor or two), any multiple of 64 channels
    const GemmSeg& s1 = p.seg[1];
    if (s1.taps != 1 || s1.pad != 0 || s1.c0 <= 0 || s1.c0 % 64 != 0 || s1.c1 % 64 != 0 || (s1.c1 > 0 && !s1.a1_hi) || s1.c0 + s1.c1 > 2048) return false;
  }
  const int tv_out = p.Tv_out > 0 ? p.Tv_out : p.T_out, tv_in = p.Tv_in > 0 ? p.Tv_in : p.T_in;
  if (p.T_in != p.T_out || tv_in != tv_out || p.T_virt != tv_out || tv_out > p.T_out || tv_out <= p.T_out - 32) return false;
  if (p.T_out % 32 != 0 || p.N % BN != 0 || p.B <= 0 || p.M != p.B * p.T_out) return false;
  if ((p.epi != EPI_STORE && p.epi != EPI_RESIDUAL) || p.stats || p.rowstat_out || p.ln_stat || p.relu || p.rowmask) return false;
  if ((p.ldo & 7) != 0 || (p.epi == EPI_RESIDUAL && (p.ldres & 3) != 0) || p.force_tile != GT_AUTO) return false;
  return true;
}
// row tiles of such a launch (64 rows; 128 output rows for the upsampling form): per utterance, the last one may be short
int gemm_conv3_row_tiles(const GemmParams& p, int bm) { return p.B * ((p.T_out + bm - 1) / bm); }
// Fused split-K pair on the streaming kernel (GemmParams sk_buf / sk_ticket, sk_split 2): 2 when the launch should run as two
// workgroups per tile - at most half the CUs' worth of tiles and a k-loop of at least eight steps (the 128-frame level's
// K >= 1920 convolutions: 128 tiles) - else 0.  The three-tap list must split into two even halves.
int gemm_conv3_split(const GemmParams& p, int n_cu) {
  static const bool on = [] { const char* e = getenv("DVITS_CONV3_SPLIT"); return !(e && e[0] == '0'); }();
  if (!on || n_cu <= 0 || !gemm_conv3_shape_ok(p)) return 0;
  const int cin = p.seg[0].c0 + p.seg[0].c1, nch = cin >> 6, nc2 = p.nseg > 1 ? (p.seg[1].c0 + p.seg[1].c1) >> 6 : 0;
  if (cin <= 512 && cin <= g_conv3_stream && p.nseg == 1) return 0;       // (the resident kernel has no split form)
  if (2 * gemm_conv3_row_tiles(p, BM) * (p.N / BN) > n_cu || nch < 4 || nch + ((nc2 + 1) >> 1) < 8) return 0;
  return 2;
}
// scratch of such a pair: the half-fragment sums of every tile (the short tiles of a padded row space included)
size_t gemm_conv3_split_bytes(const GemmParams& p) { return (size_t)2 * gemm_conv3_row_tiles(p, BM) * (p.N / BN) * BM * BN * sizeof(float); }
// ... and the upsampling form (k_conv3u): nearest x2 folded into the row gather, 128 x 64 tiles, the widths the resident image fits
bool gemm_conv3_up_ok(const GemmParams& p) {
  if (!g_conv3_on || p.nseg != 1 || p.seg[0].taps != 3 || p.seg[0].pad != 1 || p.stride != 1 || p.up_mode != UP_X2) return false;
  const int cin = p.seg[0].c0 + p.seg[0].c1;
  if (cin != 128 && cin != 256 && cin != 384 && cin != 512) return false;
  if (p.seg[0].c0 % 64 != 0 || p.seg[0].c1 % 64 != 0 || (p.seg[0].c1 > 0 && !p.seg[0].a1_hi)) return false;
  const int tv_out = p.Tv_out > 0 ? p.Tv_out : p.T_out, tv_in = p.Tv_in > 0 ? p.Tv_in : p.T_in;
  if (tv_out != 2 * tv_in || p.T_virt != tv_out || tv_out > p.T_out || tv_out <= p.T_out - 32 || tv_in > p.T_in) return false;
  if (p.T_out % 32 != 0 || p.N % BN != 0 || p.B <= 0 || p.M != p.B * p.T_out) return false;
  if ((p.epi != EPI_STORE && p.epi != EPI_RESIDUAL) || p.stats || p.rowstat_out || p.ln_stat || p.relu || p.rowmask) return false;
  if ((p.ldo & 7) != 0 || (p.epi == EPI_RESIDUAL && (p.ldres & 3) != 0) || p.force_tile != GT_AUTO || p.sk_buf) return false;
  return true;
}
// packed K (weight row length) such a launch expects: 3 C_in (+ the one-tap segment's channels)
int gemm_conv3_k(const GemmParams& p) {
  return 3 * (p.seg[0].c0 + p.seg[0].c1) + (p.nseg > 1 ? p.seg[1].c0 + p.seg[1].c1 : 0);
}
// the per-utterance row-tile geometry of a launch (GemmParams c3_tu / c3_tu_magic)
static void conv3_geometry(GemmParams& p, int bm) {
  p.c3_tu = (p.T_out + bm - 1) / bm;
  p.c3_tu_magic = p.c3_tu <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)p.c3_tu - 1) / (unsigned)p.c3_tu);
}

template <int CIN>
static hipError_t conv3_launch(const GemmParams& p, hipStream_t st) {
  constexpr int smem = ConvGeom<CIN>::SMEM;
  hipLaunchKernelGGL((k_conv3<CIN>), dim3(gemm_conv3_row_tiles(p, BM) * (p.N / BN)), dim3(NT), smem, st, p);
  return hipGetLastError();
}
template <int CIN>
static hipError_t conv3_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3<CIN>), hipFuncAttributeMaxDynamicSharedMemorySize, ConvGeom<CIN>::SMEM);
}
template <int CIN>
static hipError_t conv3u_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3u<CIN>), hipFuncAttributeMaxDynamicSharedMemorySize, ConvGeomU<CIN>::SMEM);
}
template <int CIN>
static hipError_t conv3u_launch(const GemmParams& p, hipStream_t st) {
  constexpr int smem = ConvGeomU<CIN>::SMEM;
  hipLaunchKernelGGL((k_conv3u<CIN>), dim3(gemm_conv3_row_tiles(p, BMU) * (p.N / BN)), dim3(NT), smem, st, p);
  return hipGetLastError();
}
hipError_t launch_conv3_up(const GemmParams& pin, hipStream_t st) {
  GemmParams p = pin;
  if (!p.wf_hi || !p.wf_lo || !gemm_conv3_up_ok(p) || p.Kp != 3 * (p.seg[0].c0 + p.seg[0].c1) || p.sk_mode != 0) return hipErrorInvalidValue;
  if (p.Tv_out <= 0) p.Tv_out = p.T_out;
  if (p.Tv_in <= 0) p.Tv_in = p.T_in;
  conv3_geometry(p, BMU);
  if (p.xcd_n > 0) p.xcd_n = 0;                      // (row bands; launch_gemm's rectangle belongs to another tile)
  switch (p.seg[0].c0 + p.seg[0].c1) {
    case 128: return conv3u_launch<128>(p, st);
    case 256: return conv3u_launch<256>(p, st);
    case 384: return conv3u_launch<384>(p, st);
    default: return conv3u_launch<512>(p, st);
  }
}
hipError_t conv3_init() {
  hipError_t e;
  if ((e = conv3u_attr<128>()) != hipSuccess) return e;
  if ((e = conv3u_attr<256>()) != hipSuccess) return e;
  if ((e = conv3u_attr<384>()) != hipSuccess) return e;
  if ((e = conv3u_attr<512>()) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3s), hipFuncAttributeMaxDynamicSharedMemorySize, S_TOTAL)) != hipSuccess) return e;
  if ((e = conv3_attr<128>()) != hipSuccess) return e;
  if ((e = conv3_attr<256>()) != hipSuccess) return e;
  if ((e = conv3_attr<384>()) != hipSuccess) return e;
  return conv3_attr<512>();
}
// (called by launch_gemm with p validated, tout_magic and the XCD rectangle of the 64x64 tile grid set)
hipError_t launch_conv3(const GemmParams& pin, hipStream_t st) {
  GemmParams p = pin;
  if (!p.wf_hi || !p.wf_lo || !gemm_conv3_shape_ok(p) || p.Kp != gemm_conv3_k(p) || (p.sk_mode != 0 && p.sk_mode != 3)) return hipErrorInvalidValue;
  if (p.Tv_out <= 0) p.Tv_out = p.T_out;
  if (p.Tv_in <= 0) p.Tv_in = p.T_in;
  conv3_geometry(p, BM);
  const int tiles = gemm_conv3_row_tiles(p, BM) * (p.N / BN);
  if (p.sk_mode == 3) {
    if (!p.sk_buf || !p.sk_ticket || p.sk_split != 2 || p.seg[0].c0 + p.seg[0].c1 < 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_conv3s, dim3(2 * tiles), dim3(NT_S), S_TOTAL, st, p);
    return hipGetLastError();
  }
  const int cin = p.seg[0].c0 + p.seg[0].c1;
  if (cin > 512 || cin > g_conv3_stream || p.nseg > 1) {
    hipLaunchKernelGGL(k_conv3s, dim3(tiles), dim3(NT_S), S_TOTAL, st, p);
    return hipGetLastError();
  }
  switch (cin) {
    case 128: return conv3_launch<128>(p, st);
    case 256: return conv3_launch<256>(p, st);
    case 384: return conv3_launch<384>(p, st);
    default: return conv3_launch<512>(p, st);
  }
}
