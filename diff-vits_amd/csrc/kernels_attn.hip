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
// vector-ALU work (softmax, P -> hi + lo, the conversion), not by MFMA: see DESIGN.md and tools/attn_trace.py.
#include "dv_common.h"
#define DV_ATTN_TRACE_OWNER   // the trace build's stamp buffer lives in this translation unit
#include "attn_tile.h"

#include <cstdlib>

template <int DP, int NW, int NSPLIT>
__global__ __launch_bounds__(64 * NW) void k_attention(const AttnParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];   // [2 x BUF planes | NST x RAW ring]
  attn_tile<DP, NW, NSPLIT, false>(p, blockIdx.x, blockIdx.y, blockIdx.z, lds);
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
  done = e == hipSuccess;
  return e;
}

hipError_t launch_attention(const AttnParams& p, hipStream_t st) {
  if (p.d % 4 != 0 || p.d > 64 || p.d <= 0 || (p.nsplit != 1 && p.nsplit != 3)) return hipErrorInvalidValue;
  // queries per workgroup: the K / V tiles are converted once per workgroup, so more waves per workgroup amortise
  // that, fewer spread the launch over more CUs.  Measured at the bench shape (after the conversion was made cheap):
  // 8 waves (256 queries) for the 1024-frame level, 4 waves below it (DVITS_ATTN_NW8 / DVITS_ATTN_NW4 = wave-count
  // thresholds for 8 / 4 waves per workgroup)
  const long waves = (long)p.B * p.H * ((p.Tq + 31) / 32);
  static const int nw8_min = [] { const char* e = getenv("DVITS_ATTN_NW8"); return e ? atoi(e) : 1100; }();
  static const int nw4_min = [] { const char* e = getenv("DVITS_ATTN_NW4"); return e ? atoi(e) : 64; }();
  const int nw = waves >= nw8_min ? 8 : (waves >= nw4_min ? 4 : (waves >= 16 ? 2 : 1));
  dim3 grid((p.Tq + 32 * nw - 1) / (32 * nw), p.H, p.B);
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
