/*
 * dvits_hip.h — C ABI of libdvits_hip.so, the MI355X (gfx950) engine under the
 * diff-vits diffusion-sampling path.
 *
 * The reference (adelacvg/diff-vits) has no FFI layer: its boundary for this path is
 * the Python surface
 *     unet1d/unet_1d_condition.py:743-1037   UNet1DConditionModel.forward
 *     sampler/dpm_solver.py:1047-1245         DPM_Solver.sample  (multistep, dpmsolver++)
 *     sampler/uni_pc.py:590-672               UniPC.sample       (multistep, bh1/bh2)
 * The Python mirror classes in diff-vits_amd/{unet1d,sampler}/ keep that surface and
 * bind the entry points below through ctypes (diff-vits_amd/_lib.py); INTEGRATION.md
 * shows the stub.  Plain pointers and sizes only: no torch types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success, a negative dv_status otherwise;
 *     dv_last_error() returns a thread-local message for the last failure.
 *   - all tensor pointers are DEVICE pointers to contiguous float32 unless noted.
 *     The caller owns inputs/outputs; the library owns packed weights and workspace.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Work is
 *     enqueued asynchronously on it; nothing here synchronises the device except
 *     dv_unet_prepare / dv_penc_prepare / dv_*_destroy, the single-operator dv_op_* entry points (they
 *     return results) and the first dv_sampler_run of a plan (it captures the loop into a hipGraph).
 *   - handles are not thread-safe; distinct handles are independent.
 */
#ifndef DVITS_HIP_H
#define DVITS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  DV_OK = 0,
  DV_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
  DV_ERR_HIP = -2,          /* a HIP runtime call failed */
  DV_ERR_STATE = -3,        /* call order violated (e.g. forward before prepare) */
  DV_ERR_MISSING_WEIGHT = -4
} dv_status;

/* Contraction precision (SURVEY.md §7 "Accuracy vs bf16"):
 *   DV_PREC_BF16X3: split-bf16, 3 MFMA products per tile, fp32 accumulate — the parity mode
 *   DV_PREC_BF16  : single bf16 product — fast mode, ~7e-3 rel. error, reported separately */
typedef enum { DV_PREC_BF16X3 = 0, DV_PREC_BF16 = 1 } dv_precision;

/* Constructor keywords of UNet1DConditionModel that shape the network
 * (reference unet_1d_condition.py:151-203; diffusion values model3.py:887-896). */
typedef struct {
  int32_t in_channels;            /* C + cond channels, e.g. 208 */
  int32_t out_channels;           /* C, e.g. 80 */
  int32_t n_levels;               /* len(block_out_channels), 2..6 */
  int32_t block_out_channels[6];
  int32_t layers_per_block;       /* 2 */
  int32_t num_heads;              /* `attention_head_dim` keyword (= head count) */
  int32_t cross_attention_dim;    /* encoder_hidden_states feature dim */
  int32_t norm_num_groups;        /* 8 */
  int32_t add_embed_heads;        /* addition_embed_type_num_heads, 64 */
  float   norm_eps;               /* 1e-5 */
} dv_unet_cfg;

/* PromptEncoder constructor arguments (reference model3.py:382-406 as built by Diffusion_Encoder, model3.py:898:
 * PromptEncoder(100, hidden, hidden, 4, 0.2); every layer is OPERATIONS_ENCODER[8] = EncSALayer(c, 8 heads,
 * kernel_size 9, 'SAME'), operations.py:961-964). */
typedef struct {
  int32_t in_channels;            /* mel channels of the prompt, 100 */
  int32_t hidden_channels;        /* 128 (config.json diffusion_encoder.hidden_channels) */
  int32_t out_channels;           /* = hidden_channels at the reference call site */
  int32_t n_layers;               /* 4 */
  int32_t num_heads;              /* 8 */
  int32_t ffn_kernel;             /* 9 */
} dv_penc_cfg;

typedef struct dv_unet dv_unet;
typedef struct dv_plan dv_plan;
typedef struct dv_penc dv_penc;

const char* dv_last_error(void);
/* Library/version probe: returns e.g. "dvits_hip 0.1 gfx950". */
const char* dv_version(void);

/* ---- denoiser: UNet1DConditionModel ------------------------------------------------ */

/* Replaces UNet1DConditionModel.__init__ (unet_1d_condition.py:151-607). */
int dv_unet_create(const dv_unet_cfg* cfg, dv_unet** out);
void dv_unet_destroy(dv_unet* u);

/* Hand over one state-dict tensor under its reference name, e.g.
 * "down_blocks.1.attentions.0.transformer_blocks.0.attn2.to_k.weight"
 * (replaces load_state_dict, tts_infer.py:75-81).  `dev_ptr` is float32, contiguous,
 * shape[ndim]; the data is copied (the caller keeps ownership). */
int dv_unet_set_weight(dv_unet* u, const char* name, const void* dev_ptr, const int64_t* shape, int32_t ndim);

/* Pack weights for `precision`, build the kernel schedule for (B, T, L) and allocate the
 * workspace arena (T is arbitrary: levels whose frame count is no multiple of 32 are laid out in a padded row space
 * inside the engine).  Must be called after all weights are set and again whenever B/T/L,
 * the precision or the weights change.  `force_upsample_size` = the reference's
 * forward_upsample_size flag (unet_1d_condition.py:789-797), computed by the caller from
 * the shape of `sample`. */
int dv_unet_prepare(dv_unet* u, int32_t B, int32_t T, int32_t L, int32_t precision, int32_t force_upsample_size);

/* Step-invariant conditioning (hoisted out of the sampler loop): pooled-text embedding
 * add_embedding(enc) (unet_1d_condition.py:869-870), the 16 cross-attention K/V
 * projections (attention_processor.py:1016-1017) and the additive mask bias (:816-818).
 * enc: [B, L, cross_attention_dim]; mask_bias: [B, L] additive (0 / -10000) or NULL. */
int dv_unet_set_cond(dv_unet* u, const float* enc, const float* mask_bias, void* stream);

/* One denoiser evaluation = UNet1DConditionModel.forward (unet_1d_condition.py:743-1037).
 *   x    : [B, cx, T]            first cx input channels (channels-first, as the reference)
 *   cond : [B, in_channels-cx, T] remaining input channels, or NULL when cx == in_channels
 *   t    : [B] float32 timesteps (fractional allowed)
 *   y    : [B, out_channels, T]  output
 * Passing x and cond separately avoids materialising torch.cat([x, cond]) (model3.py:908). */
int dv_unet_forward(dv_unet* u, const float* x, int32_t cx, const float* cond, const float* t, float* y,
                    void* stream);

/* Number of kernel launches one forward enqueues, and algorithmic FLOPs (2*MAC of all
 * contractions) of one forward at the prepared shape — for the roofline report. */
int dv_unet_stats(dv_unet* u, int64_t* n_launch, double* flops);
/* Number of schedule OPERATIONS of one forward (the index range of dv_unet_op_info / dv_unet_forward_timed).  An
 * operation is one kernel launch, except a split-K GEMM pair (k-slice pass + epilogue pass), which is one operation
 * of two launches. */
int dv_unet_op_count(dv_unet* u, int32_t* n_ops);

/* Measurement aid for bench.py: one forward with a HIP event pair around every launch of the
 * schedule (eager, not graph-replayed); ms_per_op[i] = elapsed ms of operation i (capacity >= dv_unet_op_count).
 * dv_unet_op_info gives operation i's kernel family ("gemm", "attn", "gn_partial", "gn_finalize",
 * "ln_stats", "misc"), its algorithmic FLOPs (0 for non-contractions) and a shape description. */
int dv_unet_forward_timed(dv_unet* u, const float* x, int32_t cx, const float* cond, const float* t, float* y,
                          void* stream, float* ms_per_op, int32_t capacity);
int dv_unet_op_info(dv_unet* u, int32_t index, char* kind16, double* flops, char* desc128);
/* Re-launches only the schedule's launches of one kernel family (e.g. "gemm"), in schedule order with their own
 * arguments, `reps` times back to back between ONE HIP event pair on `stream` (after one untimed pass);
 * *ms_total / *launches = that family's average launch duration without per-launch event overhead.  Needs a completed
 * dv_unet_forward (the buffers then hold valid data; outputs are overwritten with the same values). */
int dv_unet_time_family(dv_unet* u, const char* kind, int32_t reps, void* stream, float* ms_total, int32_t* launches);

/* Persistent per-XCD schedule (environment DVITS_PERSIST=1 at prepare time; csrc/persist.hip): *n_ops = number of
 * schedule operations that run inside the one persistent launch (0 = off / not applicable to this shape);
 * *error_flag = 0 ok, 1 = an in-launch barrier timed out, 2 = the workgroups were not spread evenly over the XCDs
 * (synchronises the device). */
int dv_unet_persist_status(dv_unet* u, int32_t* n_ops, int32_t* error_flag);
/* In-launch hand-overs (default; environment DVITS_GNX=0 at prepare time restores the separate k_gn_apply launches and the
 * two-GEMM feed-forward).  GroupNorm finished inside the producer GEMM: the workgroups of such a GEMM exchange their tile
 * statistics inside the launch and wait for one another (bounded).  Feed-forward block of the C = 256 / 384 / 512 transformer
 * blocks as one launch (round 5, csrc/kernels_ffsplit.hip; reference unet1d/attention.py:189-203, 206-255; DVITS_FF_SPLIT=0
 * restores the two GEMMs): the 4 / 8 workgroups of a row block hand their partial sums over the same way.  *n_ops = schedule
 * operations that wait inside their launch; *timed_out = 1 if any of them ever gave up waiting - the
 * results of that launch are invalid and every later dv_unet_forward / dv_sampler_run on this handle fails with
 * DV_ERR_HIP.  Does not synchronise: meaningful once the stream has drained. */
int dv_unet_handover_status(dv_unet* u, int32_t* n_ops, int32_t* timed_out);
/* Recovery from such a time-out (a foreign kernel kept some workgroups off the CUs past the bounded wait): synchronises the
 * device and clears the flag, so that the handle works again.  The caller then re-plans it with dv_unet_set_exclusive(u, 0) -
 * GroupNorm as separate launches, no in-launch waits - and repeats the lost run; diff_vits_amd/engine.py does exactly that
 * (UNetEngine.recover_handover; WHEN it checks - lazily since round 5, UNetEngine.wait() being the explicit point - is
 * described in INTEGRATION.md section 4) and reports the downgrade once. */
int dv_unet_handover_reset(dv_unet* u);
/* The in-launch waits above assume that the launch has the device to itself (every workgroup resident at once): true
 * for one stream, or for several streams that never run kernels of such handles side by side.  A host that drives
 * several handles CONCURRENTLY on different streams of one device must call this with exclusive = 0 before
 * dv_unet_prepare (GroupNorm then runs as separate launches on that handle); two such GEMMs sharing the CUs could wait
 * for each other's queued workgroups - the bounded wait turns that into DV_ERR_HIP, not a hang, but the run is lost. */
int dv_unet_set_exclusive(dv_unet* u, int32_t exclusive);
/* Profiling aid: s_memtime stamps taken by one workgroup of XCD 0 after every operation of the last persistent launch;
 * returns the number of stamps written (operations + 1) or a negative error; *first_op = schedule index of the first
 * operation inside the launch (pairs with dv_unet_op_info). */
int dv_unet_persist_ticks(dv_unet* u, int32_t* first_op, uint64_t* ticks, int32_t capacity);

/* Debug/parity probe: copy a named intermediate activation (channels-last [B, T, C]) of the
 * last forward to the host.  Available only when dv_unet_prepare ran with the environment
 * variable DVITS_KEEP_INTERMEDIATES=1 (buffers are then never reused).  dims[3] = {B, T, C};
 * host_out may be NULL to query dims.  Names follow the reference module paths, e.g.
 * "conv_in", "emb", "down_blocks.0.resnets.0", "down_blocks.0.attentions.0",
 * "mid_block.resnets.1", "up_blocks.1.upsamplers.0". */
int dv_unet_probe(dv_unet* u, const char* name, float* host_out, int64_t capacity, int64_t* dims);

/* ---- sampler: DPM-Solver++ / UniPC multistep loops --------------------------------- */

/* DV_SOLVER_UNIPC_VARY: variant='vary_coeff' (multistep_uni_pc_vary_update, uni_pc.py:368-469).  DPM-Solver++ takes
 * orders 1-3 (the reference's three closed forms); the UniPC variants any order 1..8 (the reference solves the
 * order x order systems numerically, uni_pc.py:545-560 / :410-420). */
/* DV_SOLVER_DPM: algorithm_type='dpmsolver' - the multistep updates on the NOISE prediction (dpm_solver.py:581-592, 841-847,
 * 895-904); the network still predicts x0: every evaluation is followed by eps = (x - alpha_t x0) / sigma_t on its history
 * slot (model_wrapper's 'x_start' branch, dpm_solver.py:290-292).
 * DV_SOLVER_*_TAYLOR: solver_type='taylor' - the second-order update's Taylor form (dpm_solver.py:825-829, 848-851); orders 1
 * and 3 are the same as without it.
 * DV_SOLVER_UNIPC_*_NOISE: UniPC(algorithm_type='noise_prediction') - the same three variants on the noise prediction
 * (uni_pc.py:448-468, 569-587), with the in-place x0 -> noise conversion of DV_SOLVER_DPM. */
typedef enum { DV_SOLVER_DPMPP = 0, DV_SOLVER_UNIPC_BH1 = 1, DV_SOLVER_UNIPC_BH2 = 2, DV_SOLVER_UNIPC_VARY = 3, DV_SOLVER_DPM = 4,
               DV_SOLVER_DPMPP_TAYLOR = 5, DV_SOLVER_DPM_TAYLOR = 6, DV_SOLVER_UNIPC_BH1_NOISE = 7, DV_SOLVER_UNIPC_BH2_NOISE = 8,
               DV_SOLVER_UNIPC_VARY_NOISE = 9 } dv_solver;
typedef enum { DV_SKIP_TIME_UNIFORM = 0, DV_SKIP_TIME_QUADRATIC = 1, DV_SKIP_LOGSNR = 2 } dv_skip;
/* NoiseScheduleVP(schedule=...): 'discrete' (betas; dpm_solver.py:98-107, uni_pc.py:59-66), or the continuous-time VP
 * schedules 'linear' (beta_0, beta_1; dpm_solver.py:108-111,133-134,160-163) and 'cosine' (uni_pc.py:73-100; UniPC only -
 * dpm_solver.py:94 knows 'discrete' and 'linear').  Continuous schedules: total_N = 1000, T = 1 (0.9946 for 'cosine'), and
 * the network is called with t itself instead of (t - 1/N) * N (dpm_solver.py:271-280). */
typedef enum { DV_SCHEDULE_DISCRETE = 0, DV_SCHEDULE_LINEAR = 1, DV_SCHEDULE_COSINE = 2 } dv_schedule;

/* Host fp64 precompute of every schedule scalar of the loop
 * (NoiseScheduleVP + get_time_steps + the multistep coefficient algebra:
 * dpm_solver.py:6-167, 453-480, 547-592, 796-904; uni_pc.py:471-588).
 * betas: HOST float32[n_betas] (the reference's `self.betas`, model3.py:990). */
int dv_sampler_plan(int32_t solver, const float* betas, int32_t n_betas, int32_t steps, int32_t order,
                    int32_t skip_type, int32_t lower_order_final, dv_plan** out);
/* The same with the remaining multistep options of DPM_Solver.sample / UniPC.sample (dpm_solver.py:1047-1050,
 * uni_pc.py:590-591): integrate from t_start (<= 0: T = 1) down to t_end (<= 0: 1/N); denoise_to_zero appends the
 * data prediction at t_end as one more model evaluation (dpm_solver.py:1234-1240). */
int dv_sampler_plan_ex(int32_t solver, const float* betas, int32_t n_betas, int32_t steps, int32_t order,
                       int32_t skip_type, int32_t lower_order_final, double t_start, double t_end,
                       int32_t denoise_to_zero, dv_plan** out);
/* ... for any schedule kind: betas / n_betas are read for DV_SCHEDULE_DISCRETE only, beta_0 / beta_1 for DV_SCHEDULE_LINEAR
 * only (the reference's defaults: 0.1, 20). */
int dv_sampler_plan_sched(int32_t solver, int32_t schedule, const float* betas, int32_t n_betas, double beta_0, double beta_1,
                          int32_t steps, int32_t order, int32_t skip_type, int32_t lower_order_final, double t_start,
                          double t_end, int32_t denoise_to_zero, dv_plan** out);
/* ... and for the singlestep methods of DPM_Solver.sample ("DPM-Solver-fast", dpm_solver.py:482-539, 594-794, 1214-1232):
 * `steps` model evaluations shared out over outer steps of order <= `order`; DV_METHOD_SINGLESTEP_FIXED: steps / order outer
 * steps of exactly `order`.  DPM-Solver(++) solvers only. */
typedef enum { DV_METHOD_MULTISTEP = 0, DV_METHOD_SINGLESTEP = 1, DV_METHOD_SINGLESTEP_FIXED = 2 } dv_method;
int dv_sampler_plan_method(int32_t solver, int32_t schedule, const float* betas, int32_t n_betas, double beta_0, double beta_1,
                           int32_t method, int32_t steps, int32_t order, int32_t skip_type, int32_t lower_order_final,
                           double t_start, double t_end, int32_t denoise_to_zero, dv_plan** out);
void dv_plan_destroy(dv_plan* p);

/* Introspection for tests: number of model evaluations, and the plan's tables.
 * t_input[nfe] = timestep fed to the network at each evaluation;
 * timesteps[steps+1] = continuous-time grid. */
int dv_plan_info(const dv_plan* p, int32_t* nfe, double* t_input, double* timesteps);
/* n_timesteps = entries of `timesteps` (steps + 1 for the multistep loops, outer steps + 1 for the singlestep methods);
 * eval_times[nfe] = continuous time of each evaluation. */
int dv_plan_times(const dv_plan* p, int32_t* n_timesteps, double* eval_times);

/* The compiled loop, for host-side execution with an arbitrary Python callable and for tests:
 * coefficient rows (8 floats each: c0 for x, c1..c4 for the history terms) and events
 * (9 int32 each: type 0=EVAL/1=COMB, src 0=x/1=x_pred (EVAL: the network input; COMB: the
 * tensor the c0 term multiplies - sums of more than four history terms are chained through
 * x_pred), eval_idx (EVAL; -1 for COMB), dst (EVAL: history slot; COMB: 0=x/1=x_pred), coef row,
 * slot0..slot3 (-1 = unused)).  *n_slots = number of history buffers the loop needs. */
int dv_plan_coefs(const dv_plan* p, int32_t* n_rows, float* rows8);
int dv_plan_events(const dv_plan* p, int32_t* n_events, int32_t* ev9, int32_t* n_slots);

/* Run the whole loop: x_inout [B, C, T] is x_T on entry and x_0 on exit; cond
 * [B, in_channels-C, T] is the channel-concat condition.  dv_unet_set_cond must have been
 * called.  The first call for a given (plan, unet shape) captures the loop into a
 * hipGraph; later calls replay it. */
int dv_sampler_run(dv_plan* p, dv_unet* u, float* x_inout, const float* cond, void* stream);

/* Same loop with a caller-supplied model instead of the UNet, for sampler known-answer
 * tests: model(user, x_dev, t_input_host, out_dev) must enqueue out = f(x, t) on `stream`. */
typedef int (*dv_model_fn)(void* user, const float* x, double t_input, float* out, void* stream);
int dv_sampler_run_custom(dv_plan* p, dv_model_fn fn, void* user, float* x_inout, int64_t numel, void* stream);

/* ---- prompt encoder: PromptEncoder.forward, reference model3.py:408-433 ------------------
 * The reference recomputes it inside every denoiser call (Diffusion_Encoder.forward, model3.py:902-906) although it
 * does not depend on the step; the host mirror calls this once per sampler run and feeds the result to
 * dv_unet_set_cond. */
int dv_penc_create(const dv_penc_cfg* cfg, dv_penc** out);
void dv_penc_destroy(dv_penc* p);
/* State-dict tensors under their reference names relative to the PromptEncoder ("pre.layer_norm.weight",
 * "layers.0.op.self_attn.in_proj_weight" [3H,H], "layers.0.op.ffn.ffn_1.3.weight" [4H,H], "out_proj.conv.bias",
 * "layer_norm.weight", ...), float32, copied.  The two ConvTBC weights (model.py:145-146, stored [1, C_in, C_out])
 * are handed over as [C_out, C_in]: "pre.conv.weight", "out_proj.conv.weight". */
int dv_penc_set_weight(dv_penc* p, const char* name, const void* dev_ptr, const int64_t* shape, int32_t ndim);
int dv_penc_prepare(dv_penc* p, int32_t B, int32_t L, int32_t precision);
/* prompt: [B, in_channels, L] float32 (the reference's src_tokens); keep: [B, L] float32, 1 = frame < length,
 * 0 = padding (commons.sequence_mask, model3.py:416); out: [B, L, out_channels] float32 = the reference's return
 * value transposed (model3.py:912 feeds prompt.transpose(1,2) to the UNet), padding frames zero. */
int dv_penc_forward(dv_penc* p, const float* prompt, const float* keep, float* out, void* stream);
int dv_penc_stats(dv_penc* p, int64_t* n_launch, double* flops);

/* ---- single-operator entry points (parity tests of each kernel through the C ABI) ---- */

/* y[B,Cout,T_out] = conv1d(act(x)) on channels-first tensors, through the implicit-GEMM
 * kernel: k = 1 or 3, stride 1/2, padding (k-1)/2, optional nearest upsample to `up_T`
 * frames first (0 = none).  w: [Cout, Cin, k] float32, bias [Cout] or NULL. */
int dv_op_conv1d(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t Cin, int32_t T,
                 int32_t Cout, int32_t k, int32_t stride, int32_t up_T, int32_t precision, void* stream);
/* y[M,N] = x[M,K] @ w[N,K]^T + bias */
int dv_op_linear(const float* x, const float* w, const float* bias, float* y, int32_t M, int32_t K, int32_t N,
                 int32_t precision, void* stream);
/* The same contraction with the result leaving the GEMM epilogue as split bf16 planes (hi = bf16(y), lo = bf16(y - hi);
 * uint16 bit patterns, [M, ldo] each; y_lo ignored in bf16 mode) - the form every contraction of the denoiser hands to the
 * next one - and optionally as fp32 y [M, ldo] as well (NULL: planes only).  Any N and any pitch ldo >= N are legal (the
 * 16-byte plane stores of the epilogue are taken only when N and ldo are multiples of 8: tests/test_gpu_ops.py drives the
 * other paths).  geglu = 1: w = [value rows | gate rows] (N = 2 x N_out, N % 64 == 0; reference unet1d/attention.py GEGLU),
 * the planes receive N_out columns of value * gelu(gate), y must be NULL. */
int dv_op_linear_planes(const float* x, const float* w, const float* bias, float* y, uint16_t* y_hi, uint16_t* y_lo, int32_t M,
                        int32_t K, int32_t N, int32_t ldo, int32_t geglu, int32_t precision, void* stream);
/* GroupNorm statistics over a channels-last tensor x[B*T, C]: mean/rstd [B, groups]. */
int dv_op_group_stats(const float* x, float* mean, float* rstd, int32_t B, int32_t T, int32_t C, int32_t groups,
                      float eps, void* stream);
/* softmax(q k^T * d^-0.5 + bias) v on [B, T, H*d] tensors (heads interleaved on the last
 * axis as the reference's .view(B,-1,H,d)); bias [B, Tk] additive or NULL. */
int dv_op_attention(const float* q, const float* k, const float* v, const float* bias, float* o, int32_t B,
                    int32_t H, int32_t Tq, int32_t Tk, int32_t d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DVITS_HIP_H */
