// Internal declarations shared by the kernel translation units and the engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;   // raw bfloat16 bits

// ---------------------------------------------------------------------------------------
// Implicit-GEMM parameters.  One launch computes
//     out[m, n] = epilogue( sum over K-segments/taps/channels of  A(m, k) * W[n, k] )
// Row m = (b, t) of the output; the A operand is gathered on the fly (conv taps, stride,
// nearest upsample, channel-concat of two source tensors) from activations that their
// producers already stored as split bf16 planes (hi = bf16(x), lo = bf16(x - hi)).
// ---------------------------------------------------------------------------------------
enum { DV_ZERO_PAGE_BYTES = 32768 };   // >= 2 bytes x the widest A row (channels of one source tensor) + 16
enum { EPI_STORE = 0, EPI_RESIDUAL = 1, EPI_GEGLU = 2, EPI_STORE_NCT = 3 };
enum { UP_NONE = 0, UP_X2 = 1, UP_SIZE = 2 };

struct GemmSeg {
  const bf16_t* a0_hi; const bf16_t* a0_lo;   // [B*T_in, c0] planes
  const bf16_t* a1_hi; const bf16_t* a1_lo;   // [B*T_in, c1] planes or null (channel concat [a0 | a1])
  int c0, c1;          // channel counts, multiples of 32
  int taps;            // 1 or 3
  int pad;             // left padding in frames
  int nkt;             // k-tiles of this segment = taps * (c0+c1) / BK   (set by the launcher)
};

// GroupNorm of the GEMM's OWN output finished in its epilogue (gemm_tile.h "GNX"): every workgroup publishes the
// 32x16-block statistics of its tile in an exchange buffer (8-byte words, written through), polls the words of its
// groups until all of the utterance's tiles have published (EMPTY = all ones; the first kernel of every forward resets
// the buffer), reduces them (fp64, fixed order) and writes y = SiLU(GN(v) * (1 + tscale) + tshift) as split planes
// straight from the accumulator registers: the consumer's k_gn_apply launch and the fp32 round trip of the
// intermediate disappear.  No counters and no read-modify-write atomics: a launch inside a hipGraph cannot be told its
// sequence number, and a counter round trip sits on every workgroup's critical path.  ALL workgroups of the launch must
// be resident at once: launch_gemm refuses grids larger than the device's CU count (gemm_gnx_plan()).
struct GnxParams {
  unsigned long long* xchg;      // null: off.  [M / 32][N / 16] (sum | M2 << 32) words of THIS op, EMPTY before the launch
  unsigned* status;              // set to 1 if a wait timed out (results invalid; the host checks after the run)
  const float* gamma; const float* beta;   // [N]
  const float* tscale; const float* tshift; int ld_t;   // temb scale / shift rows [B, ld_t] or null
  int groups; float eps; int silu;
  int spin_max;                  // polls before a wait gives up (default 1 << 18, ~0.4 s; tests force time-outs with 1)
  bf16_t* y_hi; bf16_t* y_lo;    // normalised split planes [M, N]
  // Concatenated consumer (norm1 of an up-path resnet over [this GEMM's output | skip], reference unet_1d_blocks.py:2085,2187):
  // sk_c > 0 skip channels; gamma / beta then hold N + sk_c entries and `groups` are those of the concatenation.  The skip's
  // fp32 rows [M, sk_c] and 32x16 block statistics [M / 32, sk_c / 16, 2] are read, its normalised planes sk_y and (optional)
  // raw planes sk_raw [M, sk_c] are written - a slice per workgroup (gemm_tile.h).
  const float* sk_x; const float* sk_stat16; int sk_c;
  bf16_t* sk_y_hi; bf16_t* sk_y_lo; bf16_t* sk_raw_hi; bf16_t* sk_raw_lo;
};

struct GemmParams {
  GemmSeg seg[2];
  int nseg;
  int B, T_out, T_in, T_virt;   // T_virt: length after the (virtual) nearest upsample
  // Utterances whose length is not a multiple of 32 (the real inference call: any T) live in a row space PADDED to whole
  // 32-frame blocks per level, so that the row-block kernels, the 32x16 block statistics and the in-epilogue GroupNorm run
  // for any T: T_out / T_in are then the row PITCH per utterance (frames incl. padding: row m = b * T_out + t) and Tv_out /
  // Tv_in the frames that exist (0: same as the pitch).  Padding rows are excluded everywhere a frame count enters: a conv
  // tap that lands on one reads zeros (T_virt bounds the gather), their outputs are written as zeros, the block statistics
  // of an utterance's last block cover its valid rows only, self-attention masks them as keys.
  int Tv_out, Tv_in;
  unsigned tout_magic;          // internal (launch_gemm): ceil(2^32 / T_out) - row m's utterance is umulhi(m, magic), exact for m * T_out < 2^32;
                                // 0 when T_out == 1.  (A division by a run-time value is ~25 vector instructions on every lane's critical path.)
  int stride;
  int up_mode;                  // UP_*
  float up_scale;               // UP_SIZE: (float)T_in / T_virt, as ATen computes it
  const bf16_t* w_hi;           // [N_pad, Kp] bf16, row n = output channel, k contiguous
  const bf16_t* w_lo;           // low-order split (null in bf16 mode)
  int Kp;                       // packed K
  int N_pad;                    // rows present in w_hi/w_lo (multiple of 128)
  const float* bias;            // [N] (GEGLU: packed order) or null
  int M, N;                     // output rows, GEMM columns (GEGLU: packed 2*N_out)
  int epi;                      // EPI_*
  const float* res;             // EPI_RESIDUAL: [M, ldres] fp32
  int ldres;
  float* out;                   // fp32 output [M, ldo] or null  (EPI_STORE_NCT: [B, N, T_out], required)
  bf16_t* out_hi;               // split-plane output [M, ldo] or null
  bf16_t* out_lo;
  int ldo;
  float* stats;                 // [ceil(M/32), N, 2] per 32-row block column (sum, sumsq) or null
  float* stats16;               // [M/32, N/16, 2] per (32-row, 16-column) block (sum, sumsq) or null (N % 16 == 0)
  const bf16_t* zero_page;      // DV_ZERO_PAGE_BYTES of zeros (source of padded / out-of-range rows: a lane that reads it walks
                                // along a row's k-tiles inside it - 2 bytes per channel of the widest source tensor)
  // LayerNorm fused across two GEMMs (reference attention.py:157,176,189): the PRODUCER of x writes per-row
  // partial (sum, squared deviations from the block mean) over each 32-column block of its output; the CONSUMER multiplies the raw x by the
  // gamma-folded weights and finishes y = rstd*(acc - mean*u[n]) + bias'[n] in its epilogue.
  float* rowstat_out;           // producer: [M, N/32, 2] or null
  const float* ln_stat;         // consumer: the A operand's [M, ln_nblk, 2] row partials, or null (no LayerNorm)
  int ln_nblk;                  // = C/32 of the normalised rows
  const float* ln_u;            // consumer: u[n] = sum_k gamma[k]*W[n,k]  (packed column order)
  float ln_eps;
  // prompt-encoder epilogue options (reference operations.py:687, 813, 820): ReLU, then a per-row keep mask
  int relu;                     // 1: result = max(result, 0) after bias / residual
  const float* rowmask;         // [M] multiplier of every output row (padding frames -> 0) or null
  // split-K over two launches (long K, too few tiles to occupy the chip): scratch for the k-slices' raw accumulators
  // (>= gemm_splitk_bytes(M, N, sk_split)), or null.  The caller sets sk_buf / sk_split from gemm_splitk_plan() and
  // leaves sk_mode 0; launch_gemm runs the pair with its ordinary tile choice.
  float* sk_buf;
  int sk_split;
  int sk_mode;                  // internal: 0 single launch, 1 k-slice pass (dump), 2 epilogue pass, 3 fused pair
  unsigned* sk_ticket;          // per-tile arrival counters (zero between launches) for the fused pair, or null: two launches
  int force_tile;               // 0: launch_gemm's shape heuristic; else a GT_* tile of the menu (set by the prepare-time tuner)
  GnxParams gnx;                // GroupNorm of the output in the epilogue (needs stats16)
  int xcd_n;                    // internal (launch_gemm): XCDs the columns are split over (0: row bands of the tile grid)
  int xcd_sh_n, xcd_sh_mn, xcd_tn, xcd_tm, xcd_inv_tn;   // internal: log2 xn, log2 (xm xn), rectangle width / height in tiles, ceil(2^16 / width)
  // the same weights FRAGMENT-MAJOR (launch_relayout_frag of w_hi / w_lo), or null: with them a stride-1 three-tap convolution whose
  // input channels fit the LDS runs on k_conv3 (kernels_conv.hip, gemm_conv3_shape_ok) instead of k_gemm
  const bf16_t* wf_hi; const bf16_t* wf_lo;
  // internal (launch_conv3 / launch_conv3_up): row tiles of the convolution kernels never span two utterances - every utterance has
  // c3_tu = ceil(T_out / tile rows) of them, the last one short (a multiple of 32 rows) where the row pitch is no multiple of the
  // tile; row-tile index i -> utterance umulhi(i, c3_tu_magic), tile i - utterance * c3_tu of it
  int c3_tu; unsigned c3_tu_magic;
  // planning aid (engine.hip conv3_takes -> gemm_gnx_plan): 1 = this launch will run on k_conv3 / k_conv3s (64 x 64 tiles),
  // 2 = on k_conv3u (128 x 64 tiles); 0 = k_gemm's tile menu
  int c3_route;
#ifdef DV_GEMM_TRACE
  int trace;                    // development build (make trace): this launch stamps its phases (gemm_tile.h DV_TRACE)
#endif
};
inline unsigned gemm_tout_magic(int T_out) { return T_out <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)T_out - 1) / (unsigned)T_out); }
// exchange words a GNX GEMM needs (M / 32 * N / 16), or 0 if launch_gemm would refuse it (tile shape vs T_out / groups,
// more workgroups than `n_cu` compute units, unsupported epilogue)
int gemm_gnx_plan(const GemmParams& p, int n_cu);
// grids above the CU count may carry waiting hand-overs when one utterance's share of an XCD fits its CUs (kernels_gemm.hip)
bool gemm_handover_rounds();
// tile menu ids (kernels_gemm.hip); GT_BK64 is or-ed in when the tile runs 64-deep k-tiles
enum { GT_AUTO = 0, GT_T0 = 1, GT_T1 = 2, GT_T2 = 3, GT_T2S = 4, GT_T2G = 5, GT_T3 = 6, GT_T4 = 7, GT_T4G = 8,
       GT_BK64 = 0x100 };
// candidate tiles (force_tile values) that can run this GEMM; returns the count written to out[cap]
int gemm_candidates(const GemmParams& p, int* out, int cap);
// number of k-slices launch_gemm should run this GEMM in (0: single launch); env DVITS_SPLITK tunes / disables
int gemm_splitk_plan(int M, int N, int K, int epi);
inline size_t gemm_splitk_bytes(int M, int N, int split) {
  return (size_t)split * ((M + 127) / 128 * 128) * ((N + 127) / 128 * 128) * sizeof(float);
}

// ---------------------------------------------------------------------------------------
// Row-block chain (kernels_chain.hip): two dependent K = C contractions of a transformer block in ONE launch.  A
// workgroup owns 32 frames and ALL their channels, so everything between the two GEMMs (bias, residual, LayerNorm) is
// row-local and the intermediate never leaves LDS; only the weights stream (LDS-DMA ring).
//   stage 1:  x1 = A W1^T + b1 (+ res)          -> out1 fp32 [M, C]; raw split planes of x1 + LayerNorm row partials in LDS
//   stage 2:  y  = LN(x1) W2^T + b2 (LayerNorm gamma/beta folded into W2 / b2 / u2, finished in the epilogue)
//             -> out2 fp32 [M, ldo2], N2 = passes * C columns
// A of stage 1: split planes [M, C] written by the attention kernel (amode 0: attention -> to_out + residual -> LN ->
// to_q of the next attention, reference attention.py:157-189), or GroupNorm of the fp32 block input, normalised once per
// row-block (amode 1: norm -> proj_in -> LN -> to_q/to_k/to_v, transformer_1d.py:264-268 + attention.py:157-160).
// ---------------------------------------------------------------------------------------
struct ChainParams {
  int M, C, T;                   // rows (= B * T, T % 32 == 0: the row pitch per utterance), channels (128, 256, 384 or 512)
  int Tv;                        // frames that exist per utterance (0: T); amode 1: the GroupNorm statistics cover these only
  int amode;
  const bf16_t* a_hi; const bf16_t* a_lo;                       // amode 0: [M, C] planes
  const float* x; const float* stat16;                          // amode 1: fp32 [M, C] and its 32x16-block statistics
  const float* gamma; const float* beta; float gn_eps; int groups;
  // weights FRAGMENT-MAJOR (launch_relayout_frag of the packed [rows][Kp = C] planes)
  const bf16_t* w1_hi; const bf16_t* w1_lo; int Kp1; const float* b1;
  const float* res;              // [M, C] fp32 or null
  float* out1;                   // [M, C] fp32
  const bf16_t* w2_hi; const bf16_t* w2_lo; int Kp2; const float* b2; const float* u2;
  int passes;                    // N2 = passes * C
  float* out2; int ldo2;         // (null with the cross-attention tail: the query never leaves the workgroup)
  float ln_eps;
  // Optional tail (xa_kf_hi != null; amode 0, passes 1, 8 heads): the cross attention of the block and its output
  // projection run inside the same launch - wave h is head h for the workgroup's 32 queries (keys / values of the
  // utterance's prompt, hoisted by set_cond into MFMA-fragment order: launch_kv_frag) - then
  //   stage 3:  x3 = O W3^T + b3 + x1 -> out3 fp32, raw split planes out3_hi / out3_lo and LayerNorm row partials
  //             rowstat3 [M, C/32, 2] for the GEGLU GEMM (reference attention.py:176-203)
  const bf16_t* xa_kf_hi; const bf16_t* xa_kf_lo;   // [B][H][nT][d/16][64 lanes][8]   K fragments (32 keys x 16 channels)
  const bf16_t* xa_vf_hi; const bf16_t* xa_vf_lo;   // [B][H][nT][2][NB][64 lanes][8]  V^T fragments (32 channels x 16 keys, keys permuted)
  const float* xa_bias;          // [B][nT * 32] key bias in the log2 domain (mask bias * log2 e; -1e30 beyond L)
  int xa_nT, xa_d;               // 32-key tiles, head dim (16, 32 or 48)
  float xa_qscale;               // d^-1/2 * log2 e
  const bf16_t* w3_hi; const bf16_t* w3_lo; const float* b3;
  float* out3; bf16_t* out3_hi; bf16_t* out3_lo; float* rowstat3;
  // Optional (sa_kf_hi != null; amode 1, passes 3 = to_q | to_k | to_v): the self-attention's K and V leave the kernel as
  // MFMA fragments of 32-key tiles (tile = this row block) for k_attention_frag - no fp32 K / V, no conversion there:
  //   K  [M/32][C/16][64 lanes][8]: lane (l31, lh) = K[tile*32 + l31][grp*16 + lh*8 .. +8]
  //   V^T[M/32][C/32][2 k-blocks][64 lanes][8]: lane (l31, lh) = V[tile*32 + key(kb, lh, j)][blk*32 + l31], j = 0..7,
  //      key(kb, lh, j) = kb*16 + (j < 4 ? 4 lh + j : 8 + 4 lh + j - 4) - the order of a score accumulator's registers
  // out2 then receives the query columns only (ldo2 >= C).
  bf16_t* sa_kf_hi; bf16_t* sa_kf_lo; bf16_t* sa_vf_hi; bf16_t* sa_vf_lo;
  // Cross-attention tail on TWO workgroups per row block (nsplit = 2 with xa_kf_hi, C = 256): heads 0-3 / 4-7, two waves per
  // head (key tiles of either parity), stage 3 as a 2-way K split handed over like the fused split-K pair of k_gemm:
  // xs_buf >= (M / 32) * 2 * C * 32 floats of scratch, xs_ticket = M / 32 arrival counters (zero between launches)
  float* xs_buf; unsigned* xs_ticket;
  // amode 1: workgroups per row block (0 / 1: one).  2 or 3: the stage-2 passes are shared out between them, each
  // repeating stage 1 - for launches with fewer row blocks than CUs (every workgroup streams the weights it multiplies)
  int nsplit;
  // k_qkv_split (kernels_qkv.hip: amode 1 with sa_*, 64-row blocks, the output columns split over C / 64 workgroups per row block):
  // qkv_split_flags(p) exchange words, EMPTY (all ones) before the launch; time-out flag / bound of the wait (GnxParams)
  unsigned long long* qs_flags; unsigned* qs_status; int qs_spin;
  bf16_t* qs_o_hi; bf16_t* qs_o_lo;   // MODE 2 (with xa_*): scratch planes [M, C] of the attention output (handed over between the slices)
  int qs_xcd;                    // internal (launcher): 1 = the slices of a row block share an XCD and hand h over through its L2
};
bool qkv_split_supported(const ChainParams& p, int precision);
bool qkv_split_xcd_local(int M);   // the slices of a row block share an XCD (required by the cross-attention form)
int qkv_split_flags(const ChainParams& p);
hipError_t qkv_split_init();
hipError_t launch_qkv_split(const ChainParams& p, int precision, hipStream_t st);
// Feed-forward tail of a transformer block as ONE row-block launch (k_chain_ff, C = 128): LayerNorm3 (finished from the
// producer's row partials) -> GEGLU (8C columns, packed [32 a | 32 gate] blocks; the 4C product stays in LDS as split
// planes) -> merged ff.net.2 + proj_out over [h3 | product] (K = 5C) + bias + block residual -> fp32 output + 32x16
// GroupNorm block statistics (+ split planes).  Reference attention.py:189-203 + transformer_1d.py:300-326.
struct ChainFFParams {
  int M, C, T;                                    // rows (T % 32 == 0: row pitch per utterance), C = 128
  int Tv;                                         // frames that exist per utterance (0: T): the output statistics cover these only
  const bf16_t* a_hi; const bf16_t* a_lo;         // raw split planes of h3 [M, C]
  const float* rowstat; float ln_eps;             // LayerNorm row partials of h3 [M, C/32, 2]
  const bf16_t* wg_hi; const bf16_t* wg_lo; const float* bg; const float* ug;   // GEGLU [8C][Kp = C] fragment-major (gamma folded), bias', u
  const bf16_t* wm_hi; const bf16_t* wm_lo; const float* bm;                    // merged [C][Kp = 5C] fragment-major, bias
  const float* res;                               // block input [M, C] fp32
  float* out; float* stats16;                     // [M, C] fp32 (null: nobody reads it); [M/32, C/16, 2] or null
  bf16_t* out_hi; bf16_t* out_lo;                 // optional split planes of the output
  GnxParams gnx;                                  // xchg != null: the consumer's GroupNorm is finished by this launch (gnx_device.h)
};
bool chain_ff_supported(const ChainFFParams& p, int precision);
// exchange words the in-launch GroupNorm of k_chain_ff needs (gnx.groups / gnx.sk_c set), 0: not possible (see gemm_gnx_plan)
int chain_ff_gnx_plan(const ChainFFParams& p, int n_cu);
hipError_t launch_chain_ff(const ChainFFParams& p, int precision, hipStream_t st);

// Feed-forward tail of a transformer block at C = 256 / 384 / 512 as ONE launch (k_ff_split, kernels_ffsplit.hip; reference
// attention.py:189-203, 206-255, 280-301 + transformer_1d.py:300-326).  As two GEMMs the block costs 55-62 us at the bench
// shape: a [M, 8C] GEGLU GEMM whose 4C-wide product makes an HBM round trip, then the K = 5C merged ff.net.2 + proj_out GEMM.
// A row-block kernel of 32 rows (k_chain_ff) does not scale to these widths: every workgroup would stream 3.3 / 7.6 MB of
// weights.  Here a workgroup owns 64 (or 32: `rows`) rows (two row fragments per weight fragment: half the weight bytes per MFMA) and ONE
// SLICE of the product columns: `nspl` workgroups per row block, workgroup id = row block * nspl + slice, so that XCD x
// (= id % 8) only ever touches slice x % nspl of the weights (its L2 holds 1 / nspl of them).
//   stage A: LN3 (from the producer's row partials) -> GEGLU for the slice's 4C / nspl product columns, product as split
//            planes in LDS
//   stage B: partial ffproj over the slice's K range [C / nspl channels of h3 | its product columns] for ALL C output columns
//   hand-over: every workgroup writes its partial sums through (8 x 16-column half fragments per 32-row fragment), publishes a
//            flag word, waits for the flags of its row block (all workgroups of the launch are resident: the planner checks,
//            the wait is bounded and flagged like the in-launch GroupNorm's) and FINISHES C / nspl output columns: partials
//            summed in slice order (deterministic), + bias + block residual -> fp32, 32x16 block statistics, optional planes,
//            optional GroupNorm of the consumer (GnxParams) - the tile [rows x C / nspl columns] behaves like a GEMM tile.
struct FFSplitParams {
  int M, C, T;                                    // rows (T % rows == 0: row pitch per utterance), C = 256, 384 or 512
  int rows;                                       // rows per workgroup: 64 or 32 (ff_split_rows; C = 512: 32)
  int Tv;                                         // frames that exist per utterance (0: T)
  int nspl;                                       // workgroups per row block (4 at C = 256, 8 at C = 384 / 512)
  const bf16_t* a_hi; const bf16_t* a_lo;         // raw split planes of h3 [M, C]
  const float* rowstat; float ln_eps;             // LayerNorm row partials of h3 [M, C/32, 2]
  const bf16_t* wg_hi; const bf16_t* wg_lo; const float* bg; const float* ug;   // GEGLU [8C][Kp = C] fragment-major (gamma folded), bias', u
  const bf16_t* wm_hi; const bf16_t* wm_lo; const float* bm;                    // merged [C][Kp = 5C] fragment-major, bias
  const float* res;                               // block input [M, C] fp32
  float* out; float* stats16;                     // [M, C] fp32 (null: nobody reads it); [M/32, C/16, 2] or null
  bf16_t* out_hi; bf16_t* out_lo;                 // optional split planes of the output
  float* xbuf;                                    // partial sums: ff_split_xbuf_floats(M, C, nspl) floats of scratch
  unsigned long long* flags;                      // (M / rows) * nspl * 8 exchange words (one per wave), EMPTY (all ones) before the launch
  unsigned* status; int spin_max;                 // time-out flag / bound of the waits (GnxParams)
  GnxParams gnx;                                  // xchg != null: the consumer's GroupNorm is finished by this launch
};
bool ff_split_supported(const FFSplitParams& p, int precision);
size_t ff_split_xbuf_floats(int M, int C, int nspl, int rows);
int ff_split_rows(int C, int M, int T, int nspl, int n_cu);   // rows per workgroup the planner should ask for: 64 or 32 (kernels_ffsplit.hip)
// exchange words of the in-launch GroupNorm ((M / 32) * (C / 16); gnx.groups / gnx.sk_c set), 0: not possible
int ff_split_gnx_plan(const FFSplitParams& p, int n_cu);
hipError_t ff_split_init();
hipError_t launch_ff_split(const FFSplitParams& p, int precision, hipStream_t st);

// cross-attention K/V of one block, fp32 [B*L, 2C] (k | v) -> MFMA-fragment-major split planes (ChainParams xa_*)
hipError_t launch_kv_frag(const float* kv, bf16_t* kf_hi, bf16_t* kf_lo, bf16_t* vf_hi, bf16_t* vf_lo, int B, int L, int C, int H,
                          hipStream_t st);
// key bias of the fused cross attention: out[b][k] = k < L ? (mask_bias ? mask_bias[b][k] : 0) * log2 e : -1e30, k < nT * 32
hipError_t launch_xbias(const float* mask_bias, float* out, int B, int L, int nT, hipStream_t st);
bool chain2_supported(const ChainParams& p, int precision);
hipError_t chain_init();
hipError_t launch_chain2(const ChainParams& p, int precision, hipStream_t st);
// packed weight plane [rows][Kp] -> fragment-major (rows % 32 == 0, Kp % 16 == 0), same size
hipError_t launch_relayout_frag(const bf16_t* src, bf16_t* dst, int rows, int Kp, hipStream_t st);

struct AttnParams {
  const float* q; const float* k; const float* v; const float* bias;
  float* o;                     // fp32 output or null
  bf16_t* o_hi; bf16_t* o_lo;   // split-plane output (consumed by the to_out GEMM) or null
  int ldq, ldk, ldv, ldo;       // row strides (floats); head h occupies columns [h*d, h*d+d)
  int B, H, Tq, Tk, d;
  int Tk_pitch;                 // key rows per utterance in k / v (0: Tk) - padded row spaces: keys [Tk, Tk_pitch) do not exist
  float scale;
  int nsplit;                   // 3: split-bf16 (hi*hi + lo*hi + hi*lo), 1: single bf16 product
  int no_xcd_map;               // internal (launcher): 1 = plain block order (DVITS_ATTN_XCD=0, experiments)
};

// Attention over K / V that arrive as MFMA fragments of 32-key tiles (k_attention_frag, kernels_attn.hip): written by
// the q|k|v chain kernel's epilogue (ChainParams sa_*: self attention) or by launch_kv_frag (the prompt's keys / values,
// hoisted out of the sampler loop: cross attention).  Blocks are 64 lanes x 8 bf16 = 1 KiB per plane:
//   K block   (b, h, t, ks):      kf + (b*k_b + h*k_h + t*k_t + ks) * 512
//   V^T block (b, h, t, kb, nb):  vf + (b*v_b + hv + t*v_t + kb*v_kb + nb*v_nb) * 512,  hv = h*v_h, channel rows from 0
//                                 (self_layout: hv = (h*d / 32) * v_nb, rows from (h*d) % 32 - fragments span heads)
struct AttnFragParams {
  const float* q; int ldq;      // fp32 queries [B*Tq, ldq], head h = columns [h*d, h*d + d)
  const bf16_t* kf_hi; const bf16_t* kf_lo; const bf16_t* vf_hi; const bf16_t* vf_lo;
  int k_b, k_h, k_t;            // K block strides (blocks)
  int v_b, v_h, v_t, v_kb, v_nb;
  int self_layout;
  const float* bias;            // [B, bias_ld] additive key bias in the LOG2 domain (-1e30 beyond Tk), or null (no mask)
  int bias_ld;
  float* o; bf16_t* o_hi; bf16_t* o_lo; int ldo;
  int B, H, Tq, Tk, d;
  float scale;
  int nsplit;
  int no_xcd_map;               // internal (launcher)
};
hipError_t launch_attention_frag(const AttnFragParams& p, hipStream_t st);

// ---------------------------------------------------------------------------------------
// Persistent per-XCD schedule (persist.hip): operation table in device memory, executed by one launch.
// ---------------------------------------------------------------------------------------
enum { POP_GEMM = 0, POP_ATTN = 1, POP_GN = 2, POP_SPLIT = 3 };
struct PersistSync {
  unsigned arrive[8][32];       // per-XCD arrival counter (one cache line each)
  unsigned rank[8][32];         // per-XCD workgroup numbering
  unsigned error;               // 1: barrier timed out, 2: unbalanced dispatch
  unsigned pad_;
  unsigned long long ticks[1024];   // s_memtime of XCD 0 / workgroup 0 after each operation (profiling aid)
};

// launchers (each enqueues on `st` and returns hipGetLastError())
hipError_t launch_gemm(const GemmParams& p, int precision, hipStream_t st);
hipError_t gemm_init();   // one-time kernel attribute setup (call outside stream capture)
// k_conv3 (kernels_conv.hip): the resident-operand kernel of the stride-1 three-tap convolutions; launch_gemm dispatches to it
bool gemm_conv3_shape_ok(const GemmParams& p);   // shape / epilogue test only (weights and precision are the caller's)
int gemm_conv3_k(const GemmParams& p);           // the packed K such a launch expects
int gemm_conv3_split(const GemmParams& p, int n_cu);   // 2: run it as a fused split-K pair (sk_buf / sk_ticket needed), 0: one workgroup per tile
size_t gemm_conv3_split_bytes(const GemmParams& p);    // ... the scratch of such a pair
int gemm_conv3_row_tiles(const GemmParams& p, int bm); // row tiles of a launch on the convolution kernels (bm = 64; 128 for the upsampling form): per utterance
hipError_t conv3_init();
void conv3_env_refresh();                        // DVITS_CONV3=0: off
hipError_t launch_conv3(const GemmParams& p, hipStream_t st);
bool gemm_conv3_up_ok(const GemmParams& p);      // the nearest-x2 upsampling convolutions (k_conv3u: 128-row tiles)
hipError_t launch_conv3_up(const GemmParams& p, hipStream_t st);
void gemm_env_refresh();  // re-read launch_gemm's environment knobs (called by every prepare)
hipError_t attn_init();
hipError_t launch_attention(const AttnParams& p, hipStream_t st);

// misc kernels (kernels_misc.hip)
// (B,C,T) x | cond -> channels-last split planes [B*T, cpad] (zero padded)
// (Tp >= T: row pitch per utterance of the planes, rows [T, Tp) zero)
hipError_t launch_pack_input(const float* x, int cx, const float* cond, int cc, bf16_t* out_hi, bf16_t* out_lo, int cpad,
                             int B, int T, hipStream_t st,
                             unsigned long long* reset = nullptr, size_t reset_words = 0, int Tp = 0);   // (+ the GnxParams exchange words set to all ones)
// fp32 [n] -> split planes
hipError_t launch_split(const float* in, bf16_t* hi, bf16_t* lo, int64_t n, hipStream_t st);
// GroupNorm (+temb scale/shift) (+SiLU) of the channel concat [a0 | a1] -> split planes [B*T, c0+c1].
// Statistics come either from the producers' per-32-row-block column sums (slab0/slab1, T % 32 == 0) or
// from a precomputed per-(batch,channel) affine (scale_in/shift_in).  raw_hi/raw_lo (optional): split
// planes of the un-normalised input (operand of the folded 1x1 shortcut).
struct GnApplyParams {
  const float* a0; const float* a1; int c0, c1;
  const float* slab0; const float* slab1;
  // alternative statistics: per (32-frame, 16-channel) block (sum, M2 about the block mean), [B*T/32][c/16][2] per source
  // (gemm_tile.h stats16): 16x fewer entries to reduce, no E[x^2] - mean^2 cancellation
  const float* st16_0; const float* st16_1;
  const float* scale_in; const float* shift_in;
  const float* gamma; const float* beta; float eps; int groups;
  const float* tscale; const float* tshift; int ld_t;
  int silu;
  bf16_t* out_hi; bf16_t* out_lo; bf16_t* raw_hi; bf16_t* raw_lo;
  int B, T;            // T: row pitch per utterance (frames incl. padding)
  int Tv;              // frames that exist (0: T) - the statistics of the last 32-row block cover these only
};
hipError_t launch_gn_apply(const GnApplyParams& p, hipStream_t st);

struct PersistOp {
  int type;                     // POP_*
  int cfg;                      // GEMM: tile configuration 0..2 (persist.hip); attention: padded head dim 16/32/48/64
  int gn_rpb, gn_chunks;        // GroupNorm apply: frames per task, tasks per (item, group)
  int64_t n4_per_item;          // split: float4 elements per batch item
  const float* sp_in; bf16_t* sp_hi; bf16_t* sp_lo;
  GemmParams g;
  AttnParams a;
  GnApplyParams gn;
};
hipError_t persist_init();      // attribute / co-residency check (call outside stream capture)
hipError_t launch_persist(const PersistOp* ops_dev, int n_ops, PersistSync* sync, int B, hipStream_t st);
// LayerNorm rows (no affine: gamma/beta are folded into the consumer's weights) -> split planes
hipError_t launch_ln_apply(const float* x, bf16_t* hi, bf16_t* lo, int M, int C, float eps, hipStream_t st);
// LayerNorm rows WITH affine (+ optional per-row mask) -> fp32 [M, C] and/or split planes [M, C]
hipError_t launch_ln_affine(const float* x, const float* gamma, const float* beta, const float* rowmask, float* out,
                            bf16_t* hi, bf16_t* lo, int M, int C, float eps, hipStream_t st);
// prompt-encoder input stage (reference model3.py:409-420, model.py:164-169): [B, C, L] channels-first ->
// masked_fill(padding, 0) -> LayerNorm(C) with affine -> split planes [B*L, cpad] (zero padded);
// also key_bias[b, t] = keep ? 0 : -1e30 for the attention kernels
hipError_t launch_prompt_pre(const float* prompt, const float* keep, const float* gamma, const float* beta, bf16_t* hi,
                             bf16_t* lo, float* key_bias, int B, int C, int L, int cpad, float eps, hipStream_t st);
// GroupNorm statistics of the channel-concat [a0 | a1] -> part[B, nchunk, G, 2] (double sum, sumsq)
hipError_t launch_gn_partial(const float* a0, int c0, const float* a1, int c1, double* part, int B, int T, int G,
                             int nchunk, hipStream_t st);
// combine partials -> per-(b,c) affine: scale = rstd*gamma*(1+ts), shift = (beta-mean*rstd*gamma)*(1+ts)+tb
// ts/tb = temb scale/shift rows [B, ld_t] (null -> 0); optional raw mean/rstd outputs [B,G]
hipError_t launch_gn_finalize(const double* part, int nchunk, const float* gamma, const float* beta,
                              const float* tscale, const float* tshift, int ld_t, float* scale, float* shift,
                              float* mean_out, float* rstd_out, int B, int T, int C, int G, float eps,
                              hipStream_t st);
hipError_t launch_ln_stats(const float* x, float* mean, float* rstd, int M, int C, float eps, hipStream_t st);
// fp32 [M, C] -> per (32-row, 16-column) block (sum, M2 about the block mean) [M/32, C/16, 2]
hipError_t launch_stat16(const float* x, float* stat16, int M, int C, hipStream_t st);
// out[m,n] = act_out( sum_k act_in(in[m,k]) * W[n,k] + b[n] ) + add[m,n];  fp32, small M
hipError_t launch_small_linear(const float* in, int ldin, const float* W, const float* b, const float* add,
                               float* out, int ldo, int M, int K, int N, int silu_in, int silu_out,
                               hipStream_t st);
// the same on transposed weights Wt [K, N] (lane = output column; wide N, M <= 16), and the one-off transpose
hipError_t launch_small_linear_t(const float* in, int ldin, const float* Wt, const float* b, const float* add, float* out,
                                 int ldo, int M, int K, int N, int silu_in, int silu_out, hipStream_t st, int add_rows = 0);   // add_rows > 0: `add` has that many rows, row r uses add[r % add_rows]
hipError_t launch_transpose_f32(const float* src, float* dst, int R, int Cc, hipStream_t st);
hipError_t launch_timestep_sincos(const float* t, float* out, int B, int dim, hipStream_t st, unsigned long long* reset = nullptr,
                                  size_t reset_words = 0);
hipError_t launch_layernorm_rows(const float* x, const float* g, const float* b, float* out, int M, int C,
                                 float eps, hipStream_t st);
// attention pooling pieces (reference embeddings.py:499-546)
hipError_t launch_mean_token(float* seq, const float* pos, int B, int L, int D, hipStream_t st);
hipError_t launch_pool_attn(const float* q, const float* kv, float* out, int B, int S, int D, int heads,
                            hipStream_t st);
hipError_t launch_layernorm_rows_into(const float* x, const float* g, const float* b, float* out, int M, int C,
                                      float eps, int rows_per_batch, int out_rows_per_batch, int out_row_off,
                                      hipStream_t st);
// weight packing: src fp32 -> bf16 hi/lo [N_pad, Kp]
struct PackSpec {
  const float* src;   // source weight
  int N;              // rows (output channels) in src
  int kind;           // 0: linear [N, C];  1: conv [N, C, taps]
  int C, taps;        // source channel count / taps
  int c_pad;          // channels per tap in the packed layout (>= C, multiple of 32)
  int k_off;          // first packed k of this piece
  int n_off;          // first packed row of this piece
  const float* kscale; // optional per-source-channel multiplier (LayerNorm gamma fold) or null
  int geglu;          // 1: permute rows so [a(32) | gate(32)] alternate per 64-row block
};
hipError_t launch_pack_weight(const PackSpec& s, bf16_t* hi, bf16_t* lo, int Kp, hipStream_t st);
// bias'[n] = bias[n] + sum_c W[n,c]*beta[c]   (LayerNorm beta fold), rows permuted like geglu packing
hipError_t launch_fold_bias(const float* W, const float* bias, const float* beta, float* out, int N, int C,
                            int n_off, int geglu, hipStream_t st);
// out[N, K] = A[N, C] * Bm[C, K] in fp64 accumulation (weight products at prepare time)
hipError_t launch_matmul_f32(const float* A, const float* Bm, float* out, int N, int Cc, int K, hipStream_t st);
hipError_t launch_copy_f32(const float* src, float* dst, int64_t n, hipStream_t st);
hipError_t launch_fill_f32(float* dst, float v, int64_t n, hipStream_t st);
// sampler update: out = c[0]*x + c[1]*m0 + c[2]*m1 + c[3]*m2 + c[4]*m3 (coefficient row of 8 floats on device)
hipError_t launch_lincomb(float* out, const float* x, const float* m0, const float* m1, const float* m2,
                          const float* m3, const float* coef, int64_t n, hipStream_t st);
