/* mi355_flow.h -- C ABI of libmi355flow.so, the MI355X-native GRPO rollout engine for
 * SD3.5-medium (MMDiT-X) that drops in under X-GenGroup/Flow-Factory's adapter API.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; `mi355_last_error()` then holds a
 *     message (thread-local).  Nothing falls back silently (reference constraints.md:144-145).
 *   - all `const void*`/`void*` data arguments are DEVICE pointers unless the name ends in
 *     `_host`; they are borrowed for the duration of the call's stream work and never aliased
 *     across calls (outputs are written into caller-allocated buffers).
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises.
 *   - dtypes: MI355_F32 / MI355_BF16 / MI355_F16; activations and weights inside the engine are
 *     bf16 with fp32 accumulation, the scheduler math is fp32.
 *
 * Reference interfaces replaced (X-GenGroup/Flow-Factory @ 2026-05-01, paths under
 * src/flow_factory/):
 *   mi355_transformer_forward  <- `self.transformer(hidden_states, timestep, encoder_hidden_states,
 *                                 pooled_projections, return_dict=False)[0]`
 *                                 models/stable_diffusion/sd3_5.py:421-428 (diffusers SD3Transformer2DModel)
 *   mi355_sde_step             <- FlowMatchEulerDiscreteSDEScheduler.step
 *                                 scheduler/flow_match_euler_discrete.py:243-438 (+ CFG combine
 *                                 sd3_5.py:431-433, cast_latents models/abc.py:172-182)
 *   mi355_denoise_step         <- SD3_5Adapter.forward  models/stable_diffusion/sd3_5.py:352-448
 *   mi355_rollout              <- the N-step loop of SD3_5Adapter.inference  sd3_5.py:258-304
 *   mi355_engine_bind_weight   <- the torch Parameters of `pipeline.transformer`
 *                                 (models/abc.py:314-320), HF state-dict names
 */
#ifndef MI355_FLOW_H
#define MI355_FLOW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_FLOW_VERSION 2

enum { MI355_F32 = 0, MI355_BF16 = 1, MI355_F16 = 2 };
enum { MI355_ODE = 0, MI355_FLOW_SDE = 1, MI355_DANCE_SDE = 2, MI355_CPS = 3 };

typedef struct mi355_engine mi355_engine;
typedef struct mi355_plan mi355_plan;

/* SD3Transformer2DModel config (diffusers `transformer/config.json`).  head_dim must be 64. */
typedef struct mi355_model_cfg {
    int32_t in_channels, out_channels, patch_size;
    int32_t num_layers, num_heads, head_dim;
    int32_t joint_attention_dim, pooled_projection_dim;
    int32_t pos_embed_max_size, time_proj_dim, ff_mult;
    uint64_t dual_layer_mask; /* bit i set: block i carries attn2 (SD3.5-medium: bits 0..12) */
    float eps;
} mi355_model_cfg;

int mi355_version(void);
const char* mi355_last_error(void);

/* ---- engine: owns a packed bf16 copy of the transformer weights --------------------------- */
int mi355_engine_create(const mi355_model_cfg* cfg, mi355_engine** out);
int mi355_engine_destroy(mi355_engine* e);
/* Copy (and convert / re-pack) one named parameter, e.g. "transformer_blocks.3.attn.to_q.weight".
 * `src` is a contiguous device tensor of dtype `dtype` and shape `shape[ndim]`.  Call again after
 * an optimizer step / EMA swap / LoRA merge to refresh (weights are live in GRPO). */
int mi355_engine_bind_weight(mi355_engine* e, const char* name, const void* src, int dtype, int ndim,
                             const int64_t* shape, void* stream);
/* 0 if every parameter of the config has been bound at least once, else error listing the first missing. */
int mi355_engine_weights_ready(mi355_engine* e);
/* With the weights bound now: how many of the forward's `n_total` attention launches run the static-bound softmax kernel (selected
 * per layer when the q/k RMSNorm weights prove |score| <= 60 in the log2 domain; the others keep the running-max kernel), and the
 * largest proven bound.  Synchronises `stream` once if a norm weight changed since the last query (never inside a rollout). */
int mi355_engine_attention_info(mi355_engine* e, void* stream, int* n_static, int* n_total, float* max_bound);
/* number of parameter tensors the engine expects, and the i-th expected name (for binding loops) */
int mi355_engine_num_params(mi355_engine* e);
const char* mi355_engine_param_name(mi355_engine* e, int i);

/* ---- plan: workspace for one (batch, n_cfg, latent_h, latent_w, n_text, max_steps) shape -- */
int mi355_plan_create(mi355_engine* e, int batch, int n_cfg, int latent_h, int latent_w, int n_text_tokens,
                      int max_steps, mi355_plan** out);
int mi355_plan_destroy(mi355_plan* p);
int64_t mi355_plan_workspace_bytes(mi355_plan* p);

/* ---- denoiser forward (K0-K13) -------------------------------------------------------------
 * latents : [batch][C][h][w] of `lat_dtype`; with n_cfg == 2 the batch is used twice
 *           (reference: latents_input = cat([latents, latents]), sd3_5.py:409-413).
 * t       : [batch*n_cfg] fp32 timesteps in [0,1000]; rounded to `t_round_dtype` before the
 *           sinusoidal embedding (reference: t.expand(B).to(latents.dtype), sd3_5.py:394).
 * enc_a/pooled_a : first half of the forward batch (negative prompt when n_cfg == 2, else the prompt)
 * enc_b/pooled_b : second half (the prompt) when n_cfg == 2, else NULL.  bf16,
 *           [batch][n_text][joint_attention_dim] and [batch][pooled_projection_dim].
 * v_out   : [batch*n_cfg][C][h][w] bf16 (order [negative, positive], sd3_5.py:432). */
int mi355_transformer_forward(mi355_plan* p, void* stream, const void* latents, int lat_dtype, const float* t,
                              int t_round_dtype, const void* enc_a, const void* pooled_a, const void* enc_b,
                              const void* pooled_b, void* v_out);

/* ---- fused CFG-combine + SDE/ODE step + log-prob (K14-K17), usable standalone ---------------
 * v_text/v_uncond : [batch][n] of `v_dtype` (the engine's own network output is bf16; fp32 / fp16 predictions are read exactly,
 *                   like the reference's `noise_pred.float()`, and CFG is combined op by op in that dtype); v_uncond NULL => no CFG.
 *                   latents: storage dtype.
 * noise           : fp32 eps [batch][n] (rollout) -- ignored when next_in != NULL (replay).
 * sigma/sigma_next/eta : device fp32, one value (scalar_stride 0) or one per sample (stride 1).
 * outputs (any may be NULL): next_out (storage dtype `lat_dtype`), next_f32 (value-rounded fp32,
 * what the reference's step() returns), mean_out fp32, noise_pred_out fp32 (CFG-combined),
 * log_prob/std_dev_t/dt [batch] fp32. */
int mi355_sde_step(void* stream, int batch, int64_t n, const void* v_text, const void* v_uncond, int v_dtype, float guidance,
                   const void* latents, int lat_dtype, const float* noise, const void* next_in, int next_in_dtype,
                   const float* sigma, const float* sigma_next, const float* eta, int scalar_stride, float sigma_max,
                   int dynamics, int compute_log_prob, void* next_out, float* next_f32, float* mean_out,
                   float* noise_pred_out, float* log_prob, float* std_dev_t, float* dt);

/* adjoint of mi355_sde_step w.r.t. the network prediction(s) (autograd through scheduler.step + the CFG combine): upstream gradients of
 * (log_prob [batch], noise_pred [batch][n], next_latents_mean [batch][n]) fp32, any may be NULL -> dv [n_cfg*batch][n] fp32, order [uncond, text];
 * v_text / v_uncond = the bf16 predictions the forward step consumed, next_in = the stored next state (replay) */
int mi355_sde_step_bwd(void* stream, int batch, int64_t n, const void* v_text, const void* v_uncond, float guidance, const void* latents,
                       int lat_dtype, const void* next_in, int next_in_dtype, const float* sigma, const float* sigma_next, const float* eta,
                       int scalar_stride, float sigma_max, int dynamics, int compute_log_prob, const float* g_log_prob,
                       const float* g_noise_pred, const float* g_mean, float* dv);

/* ---- one denoise step = forward + CFG + step (SD3_5Adapter.forward); replay when next_in != NULL */
int mi355_denoise_step(mi355_plan* p, void* stream, const void* latents, int lat_dtype, const float* t,
                       const void* enc_a, const void* pooled_a, const void* enc_b, const void* pooled_b,
                       float guidance, const float* noise, const void* next_in, int next_in_dtype,
                       const float* sigma, const float* sigma_next, const float* eta, int scalar_stride,
                       float sigma_max, int dynamics, int compute_log_prob, void* next_out, float* next_f32,
                       float* mean_out, float* noise_pred_out, float* log_prob, float* std_dev_t, float* dt);

/* ---- whole rollout: N denoise steps with no host sync ----------------------------------------
 * timesteps_host[N], sigmas_host[N+1] (scheduler.sigmas, last = 0), noise_levels_host[N]: host arrays.
 * init_latents : [batch][C][h][w] of `init_dtype` (prepare_latents output), cast to `storage_dtype`
 *                with the fp16 clamp of cast_latents.
 * step_noise   : fp32 [N][batch][C][h][w], drawn by the caller in the reference's RNG order.
 * keep_slot_host[N+1] : trajectory position -> slot in out_latents, or -1 (TrajectoryCollector).
 * out_latents  : [n_kept][batch][C][h][w] storage dtype;  out_log_probs : fp32 [N][batch]
 *                (written only for steps with noise_level > 0 when compute_log_prob);
 * out_final    : [batch][C][h][w] storage dtype (x_N). */
int mi355_rollout(mi355_plan* p, void* stream, int n_steps, const float* timesteps_host, const float* sigmas_host,
                  const float* noise_levels_host, int dynamics, float guidance, const void* init_latents,
                  int init_dtype, int storage_dtype, const float* step_noise, const void* prompt_embeds,
                  const void* pooled, const void* neg_embeds, const void* neg_pooled, const int32_t* keep_slot_host,
                  void* out_latents, float* out_log_probs, void* out_final, int compute_log_prob);

/* ---- differentiable denoise step: the `optimize()` replay -------------------
 * Replaces the grad-mode `SD3_5Adapter.forward(..., next_latents=x_{i+1})` + `loss.backward()` of GRPOTrainer.optimize (reference
 * src/flow_factory/trainers/grpo.py:229-263, :330): the forward is the launch sequence of mi355_denoise_step on per-block activation
 * buffers (log-prob bit-identical to the rollout's), the backward turns the upstream gradients of (log_prob [batch], noise_pred,
 * next_latents_mean [batch][C][h][w] fp32; any may be NULL) into fp32 weight gradients, written (overwritten) into the buffers
 * registered with mi355_engine_set_grad (same shape as the parameter; NULL un-registers).  Supported parameters: weights and biases of
 * the linear layers inside the transformer blocks (mi355_engine_grad_supported == 0); the data gradient covers the whole network.
 * The caller re-binds the CURRENT weights before the backward if they were swapped after the forward. */
int mi355_engine_set_grad(mi355_engine* e, const char* name, float* grad);
/* the same with the buffer's dtype (MI355_F32 / MI355_BF16).  bf16 is accepted for the weights and biases of the linear layers inside the
 * transformer blocks (grad_supported == 0): the kernels that finish those gradients round their fp32 sums to bf16 on the way out -- the values
 * an fp32 buffer converted to bf16 holds, without 8 bytes per parameter of HBM round trip -- what a bf16 parameter's `.grad` wants. */
int mi355_engine_set_grad_typed(mi355_engine* e, const char* name, void* grad, int dtype);
/* Gradient scope of the NEXT training-mode forward: 0 (default) = the blocks' linear layers; 1 = every parameter (`target_modules: all`):
 * AdaLN modulation linears, q/k RMSNorm weights, timestep / pooled-text MLPs, context_embedder, patch embedding, proj_out as well.  The
 * forward then stashes the un-gated projections too.  mi355_engine_grad_supported: 0 = default scope, 2 = full scope only, 1 = never. */
int mi355_engine_set_train_scope(mi355_engine* e, int full);
int mi355_engine_clear_grads(mi355_engine* e);
int mi355_engine_grad_supported(mi355_engine* e, const char* name);
int64_t mi355_plan_training_bytes(mi355_plan* p);
int mi355_denoise_step_train(mi355_plan* p, void* stream, const void* latents, int lat_dtype, const float* t, const void* enc_a,
                             const void* pooled_a, const void* enc_b, const void* pooled_b, float guidance, const void* next_in,
                             int next_in_dtype, const float* sigma, const float* sigma_next, const float* eta, int scalar_stride,
                             float sigma_max, int dynamics, int compute_log_prob, float* next_f32, float* mean_out,
                             float* noise_pred_out, float* log_prob, float* std_dev_t, float* dt);
int mi355_denoise_step_backward(mi355_plan* p, void* stream, const void* latents, int lat_dtype, float guidance, const void* next_in,
                                int next_in_dtype, const float* sigma, const float* sigma_next, const float* eta, int scalar_stride,
                                float sigma_max, int dynamics, int compute_log_prob, const float* g_log_prob, const float* g_noise_pred,
                                const float* g_mean);
/* ---- UniPC multistep predictor-corrector: evaluation-mode sampling of the Wan adapters ------------------------------------------------
 * Replaces: `UniPCMultistepSDEScheduler.step` in evaluation mode (reference src/flow_factory/scheduler/unipc_multistep.py:282-285), which
 * hands the step to diffusers' `UniPCMultistepScheduler.step` (convert_model_output -> multistep_uni_c_bh_update -> multistep_uni_p_bh_update;
 * solver body not in the reference tree: restated in oracle/unipc_ref.py, parity unpinned).  With flow sigmas and x0-prediction every update
 * is a linear combination of stored tensors with schedule-only coefficients (computed on the host, mi355_flow/unipc.py).
 *   mi355_unipc_convert: x0_out[n] (fp32) = sample - round_v(sigma * v), v = v_text, or `v_uncond + g * (v_text - v_uncond)` evaluated op by op
 *                        in `v_dtype` (as torch does on the network's bf16 output; the product with the 0-dim sigma stays in that dtype too)
 *   mi355_op_lincomb:    out[n] (`out_dtype`) = sum_{i < n_terms} round_i(coefs[i] * tensors[i]),  round_i = to dtypes[i] (a 0-dim fp32 scalar
 *                        times a half-precision tensor stays half precision in torch), fp32 accumulation in term order; n_terms <= 5
 * n must be a multiple of 4; all pointers device memory. */
int mi355_unipc_convert(void* stream, const void* v_text, const void* v_uncond, int v_dtype, float guidance, const void* sample, int sample_dtype,
                        float sigma, float* x0_out, int64_t n);
int mi355_op_lincomb(void* stream, int n_terms, const void* const* tensors, const int* dtypes, const float* coefs, void* out, int out_dtype,
                     int64_t n);
/* unit-test helper: attention forward (q pre-scaled by log2(e)/8) + flash backward; d_o / o token-major [B*S][H*64]; synchronises.
 * CONTRACT of the backward passes (round 6: the software-pipelined loops carry no tail masks): the padding -- rows [S, S_pad) of q and k,
 * columns [S, S_pad) of vT -- must be ZERO (finite was required before: a masked probability times a NaN is a NaN).  The engines' stashes are
 * zero-initialised and only written below S; a caller of this helper zero-fills its tensors. */
int mi355_op_attention_fwd_bwd(void* stream, const void* q, const void* k, const void* vT, const void* d_o, void* o, void* dq, void* dk,
                               void* dv, int B, int H, int S, int S_pad);

/* ---- operator-level entry points (unit tests, per-kernel profiling) ------------------------- */
/* out[M][N] (bf16, ld = N) = A[M][K] . W[N][K]^T + bias[N] (fp32 bias); act: 0 none, 1 silu, 2 gelu-tanh */
int mi355_op_linear(void* stream, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                    int act);
/* x[M][N] (bf16, in place) += gate[m / rows_per_sample][N] (bf16) * (A . W^T + bias): the gated-residual GEMM epilogue of the attention
 * out-projections and the second MLP linears (reference: the `hidden_states + gate.unsqueeze(1) * attn_output` lines of diffusers'
 * JointTransformerBlock.forward, reached from models/stable_diffusion/sd3_5.py:421-428) as an operator */
int mi355_op_linear_gate_res(void* stream, const void* A, const void* W, const float* bias, const void* gate, void* x, int M, int N, int K,
                             int rows_per_sample);
/* Weight-gradient product on ROW-MAJOR operands (round 6, csrc/gemm_tn.hip): out[s][n][k] (fp32, [k_split][N][K]) = sum over the s-th slice of
 * the M rows of dY[m][n] * X[m][k]; dY [M][ld_dy] and X [M][ld_x] bf16 as the backward leaves them in HBM -- no transposed copies (reference: the
 * weight gradients `accelerator.backward(loss)` produces for the trainable linear layers, trainers/grpo.py:326-330).  N % 128, K % 128 must be
 * 0; any M (a ragged last 64-row tile reads zeros for the missing rows; the slices are taken on M rounded up to 64).  variant 1 = 128 x 128 tiles,
 * 2 = 256 x 256 tiles (N % 256, K % 256 == 0), 0 = the transposed-copy path of rounds 2-5 (two zero-padded transposes + the K-contiguous GEMM;
 * scratch: (N + K) * M_pad bf16) for A/B
 * tests: both variants give the same bits. */
int mi355_op_wgrad(void* stream, const void* dY, int64_t ld_dy, const void* X, int64_t ld_x, float* out, int M, int N, int K, int k_split,
                   int variant, void* scratch, float* colsum);
/* (colsum: optional [k_split][N] fp32 -- the per-slice column sums of dY = the bias gradient's partials, taken by variant 1 from the operand
 *  fragments it holds; NULL = not taken; variant 0 ignores it) */
/* debug: mi355_op_linear (act 0) that also records s_memtime stamps per workgroup / tile / wave-group:
 * trace[((wg*16 + tile_iter)*2 + group)*4 + {0: tile start, 1: main loop start, 2: main loop end, 3: stores drained}] */
int mi355_op_linear_trace(void* stream, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                          void* trace);
/* measurement tool (scripts/clock_under_load.py): one wave writes n {s_memtime core-clock count, s_memrealtime 100 MHz count} int64 pairs to
 * `out`, ~sleep_iters * 8 k core clocks apart; launched on a side stream beside the kernels under test it shows the core clock they are
 * delivered under the package power cap */
int mi355_clock_probe(void* stream, void* out, int n, int sleep_iters);
/* q,k : [B][H][S_pad][64] bf16, vT : [B][H][64][S_pad] bf16 -> o_img [B*n_img][H*64], o_ctx [B*(S-n_img)][H*64] */
int mi355_op_attention(void* stream, const void* q, const void* k, const void* vT, void* o_img, void* o_ctx, int B,
                       int H, int S, int S_pad, int n_img);
/* out = LayerNorm(x)*(1+scale[b]) + shift[b]; x,out [M][D] bf16; shift,scale [M/rows_per_sample][D] bf16 */
int mi355_op_ln_modulate(void* stream, const void* x, const void* shift, const void* scale, void* out, int M, int D,
                         int rows_per_sample, float eps);

/* ---- VAE decode ----------------------------------------------
 * Replaces `pipeline.vae.decode(latents / scaling_factor + shift_factor)` + `image_processor.postprocess(.., 'pt')`
 * of SD3_5Adapter.decode_latents (reference src/flow_factory/models/stable_diffusion/sd3_5.py:161-172); the decoder is
 * diffusers' AutoencoderKL (HF state-dict names `decoder.*`).  Weights are bound like the transformer's: any of
 * fp32 / bf16 / fp16 device tensors in torch layout ([Co][Ci][3][3] convs, [out][in] linears), repacked on bind. */
typedef struct mi355_vae mi355_vae;
typedef struct mi355_vae_plan mi355_vae_plan;
typedef struct mi355_vae_cfg {
    int32_t latent_channels, out_channels;
    int32_t num_blocks, layers_per_block, norm_num_groups;
    int32_t block_out_channels[8];   /* encoder order, as in the HF config (decoder walks it reversed) */
    float eps, scaling_factor, shift_factor;
} mi355_vae_cfg;
int mi355_vae_create(const mi355_vae_cfg* cfg, mi355_vae** out);
int mi355_vae_destroy(mi355_vae* vae);
int mi355_vae_bind_weight(mi355_vae* vae, const char* name, const void* src, int dtype, int ndim, const int64_t* shape,
                          void* stream);
int mi355_vae_weights_ready(mi355_vae* vae);
int mi355_vae_num_params(mi355_vae* vae);
const char* mi355_vae_param_name(mi355_vae* vae, int i);
/* workspace for up to max_batch images decoded from latent_h x latent_w latents (latent_h*latent_w % 64 == 0) */
int mi355_vae_plan_create(mi355_vae* vae, int max_batch, int latent_h, int latent_w, mi355_vae_plan** out);
int mi355_vae_plan_destroy(mi355_vae_plan* plan);
int64_t mi355_vae_plan_workspace_bytes(mi355_vae_plan* plan);
/* latents [batch][latent_channels][h][w] (lat_dtype) -> images [batch][out_channels][8h][8w] (img_dtype: 0 fp32, 1 bf16;
 * values are bf16-representable either way: the reference decodes in the VAE's bf16);
 * postprocess = 1 applies (x/2 + 0.5).clamp(0, 1).  Enqueues on `stream`, never synchronises. */
int mi355_vae_decode(mi355_vae_plan* plan, void* stream, const void* latents, int lat_dtype, int batch, void* images,
                     int img_dtype, int postprocess);

/* ---- FLUX.1 rollout ------------------------------------------------------
 * Replaces `self.transformer(hidden_states=packed latents, timestep=t/1000, guidance, pooled_projections,
 * encoder_hidden_states, txt_ids=0, img_ids)` + `self.scheduler.step(...)` in Flux1Adapter.inference / .forward
 * (reference src/flow_factory/models/flux/flux1.py:151-289 loop, :294-346 forward).  Weights bind by the HF names of
 * diffusers' FluxTransformer2DModel.  Latents are PACKED: [B][(h/2)*(w/2)][in_channels = 64]. */
typedef struct mi355_flux mi355_flux;
typedef struct mi355_flux_plan mi355_flux_plan;
typedef struct mi355_flux_cfg {
    int32_t in_channels, num_layers, num_single_layers, num_heads, head_dim;
    int32_t joint_attention_dim, pooled_projection_dim, guidance_embeds, time_proj_dim;
    int32_t axes_dims_rope[3];
    float eps;
} mi355_flux_cfg;
int mi355_flux_create(const mi355_flux_cfg* cfg, mi355_flux** out);
int mi355_flux_destroy(mi355_flux* e);
int mi355_flux_bind_weight(mi355_flux* e, const char* name, const void* src, int dtype, int ndim, const int64_t* shape,
                           void* stream);
int mi355_flux_weights_ready(mi355_flux* e);
int mi355_flux_num_params(mi355_flux* e);
const char* mi355_flux_param_name(mi355_flux* e, int i);
/* latent_h x latent_w = the UNPACKED latent grid (e.g. 128 x 128 for 1024^2); n_text = T5 sequence length */
int mi355_flux_plan_create(mi355_flux* e, int batch, int latent_h, int latent_w, int n_text, int max_steps,
                           mi355_flux_plan** out);
int mi355_flux_plan_destroy(mi355_flux_plan* plan);
int64_t mi355_flux_plan_workspace_bytes(mi355_flux_plan* plan);
/* transformer only: t_model / guidance_model = the values the network embeds (device fp32 [B]); v_out bf16 packed */
int mi355_flux_forward(mi355_flux_plan* plan, void* stream, const void* latents, int lat_dtype, const float* t_model,
                       const float* guidance_model, const void* prompt_embeds, const void* pooled, void* v_out);
/* the whole N-step loop, zero host syncs; arguments as mi355_rollout (no negative prompt: guidance is embedded) */
int mi355_flux_rollout(mi355_flux_plan* plan, void* stream, int n_steps, const float* timesteps_host, const float* sigmas_host,
                       const float* noise_levels_host, int dynamics, float guidance_scale, const void* init_latents,
                       int init_dtype, int storage_dtype, const float* step_noise, const void* prompt_embeds, const void* pooled,
                       const int32_t* keep_slot_host, void* out_latents, float* out_log_probs, void* out_final,
                       int compute_log_prob);
/* ---- FLUX.1 `optimize()` replay with gradients (SURVEY.md 8(f) N1 over N3) --------------------------------------------------
 * Replaces the grad-mode `Flux1Adapter.forward(..., next_latents=x_{i+1})` + `accelerator.backward(loss)` of GRPOTrainer.optimize (reference
 * src/flow_factory/trainers/grpo.py:263, :326-330 over models/flux/flux1.py:294-346):
 *   mi355_flux_forward_train  = mi355_flux_forward on per-block activation buffers (same kernel binaries: v_out bit-identical), keeping what
 *                               the backward needs in the plan's training stash (ONE per plan, overwritten by every call);
 *   mi355_sde_step            = the scheduler step on v_out (unchanged: log-prob bit-identical to the rollout's, ratio == 1);
 *   mi355_sde_step_bwd        = its adjoint: upstream gradients of (log_prob, noise_pred, next_latents_mean) -> d v (one op, every family);
 *   mi355_flux_backward       = d v [B][Ni][C] fp32 -> fp32 weight gradients, OVERWRITING the buffers registered with mi355_flux_set_grad
 *                               (same shape as the parameter; NULL un-registers).
 * Gradient scope (mi355_flux_grad_supported == 0): weights and biases of every linear layer inside the transformer blocks --
 * transformer_blocks.N.{attn.{to_q,to_k,to_v,to_out.0,add_q_proj,add_k_proj,add_v_proj,to_add_out},ff.net.{0.proj,2},ff_context.net.{0.proj,2}},
 * single_transformer_blocks.N.{attn.{to_q,to_k,to_v},proj_mlp,proj_out}: a superset of the reference's FLUX.1 default target modules
 * (models/flux/flux1.py:76-84).  Everything else (modulation linears, norm weights, embedders, conditioning MLPs, final proj_out): 1 = never. */
int mi355_flux_set_grad(mi355_flux* e, const char* name, float* grad);
int mi355_flux_set_grad_typed(mi355_flux* e, const char* name, void* grad, int dtype);
int mi355_flux_clear_grads(mi355_flux* e);
int mi355_flux_grad_supported(mi355_flux* e, const char* name);
int64_t mi355_flux_plan_training_bytes(mi355_flux_plan* plan);
int mi355_flux_forward_train(mi355_flux_plan* plan, void* stream, const void* latents, int lat_dtype, const float* t_model,
                             const float* guidance_model, const void* prompt_embeds, const void* pooled, void* v_out);
int mi355_flux_backward(mi355_flux_plan* plan, void* stream, const float* dv);
/* unit-test helpers of the head_dim-128 backward kernels: attention forward (q pre-scaled by log2(e)/sqrt(128)) + flash backward, d_o / o
 * token-major [B*S][H*128], dq / dk / dv head-major (synchronises); q | k producer (per-head RMSNorm + RoPE) forward with its 1/rms output
 * and backward: dq / dk / dv head-major -> out [M][3*H*128] = [dq_pre | dk_pre | dv] */
int mi355_op_attention128_fwd_bwd(void* stream, const void* q, const void* k, const void* vT, const void* d_o, void* o, void* dq, void* dk,
                                  void* dv, int B, int H, int S, int S_pad);
int mi355_op_rope_norm_fwd_bwd(void* stream, const void* src, int64_t src_ld, int q_col, int k_col, const float* nw_q, const float* nw_k,
                               const float* cos_sin, void* q_out, void* k_out, float* rstd, const void* dq, const void* dk, const void* dv,
                               void* out, int M, int H, int rows_per_sample, int s_off, int S_pad, float eps, float q_scale);
/* head_dim-128 attention / RMSNorm+RoPE operators (unit tests) */
int mi355_op_attention128(void* stream, const void* q, const void* k, const void* vT, void* o_first, int64_t ld_first,
                          int n_first, void* o_rest, int64_t ld_rest, int B, int H, int S, int S_pad, int q_prescaled);
int mi355_op_rope_norm(void* stream, const void* src, int64_t src_ld, int q_col, int k_col, const float* nw_q,
                       const float* nw_k, const float* cos_sin, void* q_out, void* k_out, int M, int H, int rows_per_sample,
                       int s_off, int S_pad, float eps, float q_scale);
/* Wan's q / k producer (reference models/wan/wan2_t2v.py:426-543 -> diffusers WanAttnProcessor: RMSNorm ACROSS heads, 3-D RoPE) as an
 * operator: src [M][src_ld] bf16 (columns col .. col + H*128), weight fp32 [H*128], cos_sin fp32 [S][64][2] or NULL, out head-major
 * [B][H][S_pad][128] bf16.  max2 (optional, device uint32 [B*H], overwritten; M must be whole samples) receives the maximum of the squared norm
 * of every stored row per (batch, head), as float bits: the data-dependent score bound of the self-attention (mi355_tune_set key 24). */
int mi355_op_norm_rope_full(void* stream, const void* src, int64_t src_ld, int col, const float* weight, const float* cos_sin,
                            void* out, int M, int H, int rows_per_sample, int S_pad, float eps, float out_scale, void* max2);

/* ---- Wan2.1 text-to-video rollout ------------------------------------------
 * Replaces the cond / uncond `transformer(hidden_states=latents (B,16,T,h,w), timestep=t.expand(B), encoder_hidden_states)` passes,
 * the CFG combine and `scheduler.step(...)` in Wan2_T2V_Adapter.inference / .forward (reference
 * src/flow_factory/models/wan/wan2_t2v.py:344-376, :426-543), single-transformer Wan2.1 configuration.  Weights bind by the HF names
 * of diffusers' WanTransformer3DModel.  CFG is one forward over the batch [negative, positive]. */
typedef struct mi355_wan mi355_wan;
typedef struct mi355_wan_plan mi355_wan_plan;
typedef struct mi355_wan_cfg {
    int32_t in_channels, out_channels, num_layers, num_heads, head_dim, ffn_dim, text_dim, freq_dim;
    int32_t patch_t, patch_h, patch_w;
    float eps;
} mi355_wan_cfg;
int mi355_wan_create(const mi355_wan_cfg* cfg, mi355_wan** out);
int mi355_wan_destroy(mi355_wan* e);
int mi355_wan_bind_weight(mi355_wan* e, const char* name, const void* src, int dtype, int ndim, const int64_t* shape, void* stream);
int mi355_wan_weights_ready(mi355_wan* e);
int mi355_wan_num_params(mi355_wan* e);
const char* mi355_wan_param_name(mi355_wan* e, int i);
/* latent grid (T, h, w) = ((frames-1)/4+1, H/8, W/8); n_cfg 1 or 2 */
int mi355_wan_plan_create(mi355_wan* e, int batch, int n_cfg, int latent_t, int latent_h, int latent_w, int n_text, int max_steps,
                          mi355_wan_plan** out);
int mi355_wan_plan_destroy(mi355_wan_plan* plan);
int64_t mi355_wan_plan_workspace_bytes(mi355_wan_plan* plan);
/* transformer only: t [batch*n_cfg] device fp32; enc_b NULL when n_cfg == 1, else forward batch = [enc_a, enc_b]; v_out bf16 */
int mi355_wan_forward(mi355_wan_plan* plan, void* stream, const void* latents, int lat_dtype, const float* t, const void* enc_a,
                      const void* enc_b, void* v_out);
/* the whole N-step loop (arguments as mi355_rollout, no pooled embeddings; sigma of a step = t / 1000) */
int mi355_wan_rollout(mi355_wan_plan* plan, void* stream, int n_steps, const float* timesteps_host, const float* sigmas_host,
                      const float* noise_levels_host, int dynamics, float guidance, const void* init_latents, int init_dtype,
                      int storage_dtype, const float* step_noise, const void* prompt_embeds, const void* neg_embeds,
                      const int32_t* keep_slot_host, void* out_latents, float* out_log_probs, void* out_final, int compute_log_prob);

/* ---- Wan2.1 / Wan2.2 T2V `optimize()` replay with gradients (SURVEY.md 8(f) N1 over N4) ----------------------------------------
 * Replaces the grad-mode `Wan2_T2V_Adapter.forward(...)` + `accelerator.backward(loss)` (trainers/grpo.py:263, :326-330 over
 * models/wan/wan2_t2v.py:426-543).  Same contract as mi355_flux_forward_train / _backward: the training-mode forward is mi355_wan_forward's own
 * launch sequence on per-block buffers (prediction bit-identical), the backward writes the gradients of blocks.N.{attn1.{to_q,to_k,to_v,
 * to_out.0},attn2.{to_q,to_k,to_v,to_out.0},ffn.net.{0.proj,2}} (the reference's Wan default target modules, wan2_t2v.py:74-85) into the
 * buffers registered with mi355_wan_set_grad[_typed]; dv = d loss / d v_out for BOTH CFG halves [uncond | text], as mi355_sde_step_bwd
 * produces it.  Validated by tests/test_gpu_wan_backward.py (profiles/r04u_*, r04v_*); MI355_WAN_NATIVE_BACKWARD=0 keeps the host side off it. */
int mi355_wan_set_grad(mi355_wan* e, const char* name, float* grad);
int mi355_wan_set_grad_typed(mi355_wan* e, const char* name, void* grad, int dtype);
int mi355_wan_clear_grads(mi355_wan* e);
int mi355_wan_grad_supported(mi355_wan* e, const char* name);
int64_t mi355_wan_plan_training_bytes(mi355_wan_plan* plan);
int mi355_wan_forward_train(mi355_wan_plan* plan, void* stream, const void* latents, int lat_dtype, const float* t, const void* enc_a,
                            const void* enc_b, void* v_out);
int mi355_wan_backward(mi355_wan_plan* plan, void* stream, const float* dv);
/* unit-test helper: full-row RMSNorm [+ RoPE] producer (Wan q / k) forward with its 1 / rms output, then its backward */
int mi355_op_norm_rope_full_fwd_bwd(void* stream, const void* src, int64_t src_ld, int col, const float* weight, const float* cos_sin, void* y,
                                    float* rstd, const void* dy, void* dx, int M, int H, int rows_per_sample, int S_pad, float eps, float out_scale);

/* ---- Qwen-Image -----------------------------------------------
 * Replaces, inside QwenImageAdapter.inference / .forward (reference models/qwen_image/qwen_image.py:372-423, :476-600), the two
 * `self.transformer(...)` calls (cond / uncond), the norm-rescaled true-CFG combine (:579-587) and `self.scheduler.step(...)`.
 * Parameter names are the state_dict() keys of diffusers' QwenImageTransformer2DModel.  Latents are PACKED (B, Ni, 64).
 * Text: prompt_embeds bf16 [n_cfg*batch][n_text][joint_attention_dim], negative prompts first when n_cfg == 2, zero-padded to
 * n_text; txt_lens_host int32 [n_cfg*batch] = valid tokens per sample (NULL: all n_text) -- keys past a sample's length are masked. */
typedef struct mi355_qwen mi355_qwen;
typedef struct mi355_qwen_plan mi355_qwen_plan;
typedef struct mi355_qwen_cfg {
    int32_t in_channels, num_layers, num_heads, head_dim, joint_attention_dim, time_proj_dim;
    int32_t axes_dims_rope[3];
    int32_t scale_rope;
    float eps;
} mi355_qwen_cfg;
int mi355_qwen_create(const mi355_qwen_cfg* cfg, mi355_qwen** out);
int mi355_qwen_destroy(mi355_qwen* e);
int mi355_qwen_bind_weight(mi355_qwen* e, const char* name, const void* src, int dtype, int ndim, const int64_t* shape, void* stream);
int mi355_qwen_weights_ready(mi355_qwen* e);
int mi355_qwen_num_params(mi355_qwen* e);
const char* mi355_qwen_param_name(mi355_qwen* e, int i);
/* latent_h x latent_w = the UNPACKED latent grid (128 x 128 for 1024^2); n_cfg = 2 runs [negative | positive] as one forward batch */
int mi355_qwen_plan_create(mi355_qwen* e, int batch, int n_cfg, int latent_h, int latent_w, int n_text, int max_steps,
                           mi355_qwen_plan** out);
int mi355_qwen_plan_destroy(mi355_qwen_plan* plan);
int64_t mi355_qwen_plan_workspace_bytes(mi355_qwen_plan* plan);
/* one evaluation incl. the CFG combine: t_model device fp32 [batch] = 1000 * (t/1000 rounded to the latents' dtype), the angle base
 * of Timesteps(scale=1000); v_out bf16 [batch][Ni][64] = the prediction handed to the scheduler; v_raw (optional) bf16
 * [n_cfg*batch][Ni][64] = the raw network outputs */
int mi355_qwen_forward(mi355_qwen_plan* plan, void* stream, const void* latents, int lat_dtype, const float* t_model,
                       const void* prompt_embeds, const int32_t* txt_lens_host, float guidance_scale, void* v_out, void* v_raw);
/* the whole N-step loop, zero host syncs; arguments as mi355_flux_rollout; guidance_scale = true-CFG scale (n_cfg == 2) */
int mi355_qwen_rollout(mi355_qwen_plan* plan, void* stream, int n_steps, const float* timesteps_host, const float* sigmas_host,
                       const float* noise_levels_host, int dynamics, float guidance_scale, const void* init_latents, int init_dtype,
                       int storage_dtype, const float* step_noise, const void* prompt_embeds, const int32_t* txt_lens_host,
                       const int32_t* keep_slot_host, void* out_latents, float* out_log_probs, void* out_final, int compute_log_prob);

/* ---- Qwen-Image `optimize()` replay with gradients (SURVEY.md 8(f) N1 over N4) ----------------------------------------------
 * Replaces the grad-mode `QwenImageAdapter.forward(...)` + `accelerator.backward(loss)` of the reference's trainers (src/flow_factory/
 * trainers/grpo.py:263, :326-330; trainers/dgpo.py:352-364 -- BASELINE.json configs[4]) over models/qwen_image/qwen_image.py:476-600:
 *   mi355_qwen_forward_train = mi355_qwen_forward (same arguments, same kernel binaries: v_out bit-identical) keeping the activations the
 *                              backward needs in the plan's training stash (ONE per plan, overwritten by every call);
 *   mi355_sde_step / mi355_sde_step_bwd = the scheduler step and its adjoint, as for every family;
 *   mi355_qwen_backward      = d v_out [B][Ni][C] fp32 -> fp32 weight gradients, OVERWRITING the buffers registered with mi355_qwen_set_grad;
 *                              with n_cfg == 2 the norm-rescaled true-CFG combine (qwen_image.py:579-587) is differentiated through, norms
 *                              included, and both branches' forward batches are back-propagated together.
 * Gradient scope (mi355_qwen_grad_supported == 0): weights and biases of every linear layer inside the transformer blocks --
 * transformer_blocks.N.{attn.{to_q,to_k,to_v,to_out.0,add_q_proj,add_k_proj,add_v_proj,to_add_out},img_mlp.net.{0.proj,2},txt_mlp.net.{0.proj,2}}:
 * a superset of the reference's Qwen-Image default target modules (models/qwen_image/qwen_image.py:81-89).  img_mod / txt_mod, norm weights,
 * img_in / txt_in / txt_norm, the timestep MLP, norm_out and proj_out: 1 = never. */
int mi355_qwen_set_grad(mi355_qwen* e, const char* name, float* grad);
int mi355_qwen_set_grad_typed(mi355_qwen* e, const char* name, void* grad, int dtype);
int mi355_qwen_clear_grads(mi355_qwen* e);
int mi355_qwen_grad_supported(mi355_qwen* e, const char* name);
int64_t mi355_qwen_plan_training_bytes(mi355_qwen_plan* plan);
int mi355_qwen_forward_train(mi355_qwen_plan* plan, void* stream, const void* latents, int lat_dtype, const float* t_model,
                             const void* prompt_embeds, const int32_t* txt_lens_host, float guidance_scale, void* v_out, void* v_raw);
int mi355_qwen_backward(mi355_qwen_plan* plan, void* stream, const float* dv);
/* unit-test helper: adjoint of mi355_op_cfg_rescale -- d_out fp32 [rows][64] -> d_neg, d_pos bf16 */
int mi355_op_cfg_rescale_bwd(void* stream, const void* v_neg, const void* v_pos, float guidance_scale, const float* d_out, void* d_neg, void* d_pos,
                             int64_t rows, int channels);
/* operator-level (unit tests): comb = neg + g (pos - neg); out = comb * ||pos|| / ||comb|| per token of `channels` = 64 bf16 values */
int mi355_op_cfg_rescale(void* stream, const void* v_neg, const void* v_pos, float guidance_scale, void* out, int64_t rows, int channels);
/* RMSNorm over whole rows: x bf16 [rows][dim], weight fp32 [dim] -> out bf16 [rows][dim] */
int mi355_op_rms_rows(void* stream, const void* x, const float* weight, void* out, int rows, int dim, float eps);

/* ---- causal 3-D video VAE decode (Wan2.1 / Wan2.2-A14B / Qwen-Image VAE) ----------------------------
 * Replaces `pipeline.vae.decode` + `video_processor.postprocess_video` in Wan2_T2V_Adapter.decode_latents (reference
 * models/wan/wan2_t2v.py:215-230) and, with one latent frame, `vae.decode(...)[:, :, 0]` + `image_processor.postprocess` in
 * QwenImageAdapter.decode_latents (models/qwen_image/qwen_image.py:197-213).  Parameter names are the state_dict() keys of diffusers'
 * AutoencoderKLWan (`post_quant_conv.*`, `decoder.*`).  dim_mult as in the HF config (ascending); temporal_upsample[i] = up block i
 * doubles the frames (HF `temperal_downsample` reversed). */
typedef struct mi355_wvae mi355_wvae;
typedef struct mi355_wvae_plan mi355_wvae_plan;
typedef struct mi355_wvae_cfg {
    int32_t z_dim, base_dim, num_res_blocks, out_channels;
    int32_t dim_mult[4];
    int32_t temporal_upsample[3];
    float latents_mean[16], latents_std[16];
} mi355_wvae_cfg;
int mi355_wvae_create(const mi355_wvae_cfg* cfg, mi355_wvae** out);
int mi355_wvae_destroy(mi355_wvae* v);
int mi355_wvae_bind_weight(mi355_wvae* v, const char* name, const void* src, int dtype, int ndim, const int64_t* shape, void* stream);
int mi355_wvae_weights_ready(mi355_wvae* v);
int mi355_wvae_num_params(mi355_wvae* v);
const char* mi355_wvae_param_name(mi355_wvae* v, int i);
/* latent grid (T, h, w); h*w must be a multiple of 8 */
int mi355_wvae_plan_create(mi355_wvae* v, int max_batch, int latent_t, int latent_h, int latent_w, mi355_wvae_plan** out);
int mi355_wvae_plan_destroy(mi355_wvae_plan* plan);
int64_t mi355_wvae_plan_workspace_bytes(mi355_wvae_plan* plan);
/* latents (batch, z_dim, T, h, w) in lat_dtype; denormalise != 0: z = latents / (1/std) + mean first.  video
 * [batch][F = 1 + 4 (T-1)][out_channels][8h][8w], fp32 (0) or bf16 (1); postprocess != 0: (x/2 + 0.5) clamped to [0, 1], else the
 * decoder output clamped to [-1, 1] */
int mi355_wvae_decode(mi355_wvae_plan* plan, void* stream, const void* latents, int lat_dtype, int batch, void* video, int vid_dtype,
                      int postprocess, int denormalise);
/* operator-level (unit tests): causal convolution over x [B][T_in][H>>up][W>>up][Cin] bf16 (Cin % 64 == 0) with kt (1 | 3) temporal taps
 * reaching back in time and ks x ks (1 | 3) spatial taps; w_packed [Cout][kt*ks*ks][Cin] (mi355_op_conv_repack); out [B*T*H*W][Cout] */
int mi355_op_conv3d_causal(void* stream, const void* x, const void* w_packed, const float* bias, const void* residual, void* out,
                           int B, int T, int T_in, int H, int W, int Cin, int Cout, int kt, int ks, int upsample);
/* WanRMS_norm (+ SiLU): rows of C_pad bf16 channels, the first C real; gamma fp32 [C_pad] */
int mi355_op_wan_rms(void* stream, const void* x, const float* gamma, void* out, int64_t rows, int C, int C_pad, int silu);

/* VAE operator-level entry points (unit tests / microbenchmarks).  NHWC bf16 activations.
 * conv3x3: x [B][H>>up][W>>up][Cin] (Cin % 64 == 0), w_packed [Cout][9][Cin] bf16 (mi355_op_conv_repack), padding 1,
 * optional nearest-2x upsample of x folded in, optional residual [B*H*W][Cout] added (may alias out) -> out [B*H*W][Cout] */
int mi355_op_conv3x3(void* stream, const void* x, const void* w_packed, const float* bias, const void* residual, void* out,
                     int B, int H, int W, int Cin, int Cout, int upsample);
/* torch conv weight [Cout][Cin][taps] (dtype) -> bf16 [Cout][taps][Cin_pad] */
int mi355_op_conv_repack(void* stream, const void* w, int dtype, void* w_packed, int Cout, int Cin, int Cin_pad, int taps);
/* GroupNorm (+SiLU) over x [B][HW][C] bf16; scratch: >= B*(2048*C + 2*C) floats */
int mi355_op_group_norm(void* stream, const void* x, const float* gamma, const float* beta, void* out, float* scratch,
                        int B, int64_t HW, int C, int groups, float eps, int silu);

/* ---- launch-schedule trace (test infrastructure; csrc/sched_trace.hip) -----------------------
 * mi355_sched_trace(1) clears the buffer and makes every kernel launch / event record / stream wait of the engines append one text line:
 *   "L <stream> <kernel> R:<ptr>:<bytes>:<stride>:<count> ... W:<ptr>:<bytes>:<stride>:<count> ..."   (hex ptr; count blocks of bytes, stride apart)
 *   "E <stream> <event>"  (record)      "T <stream> <event>"  (the stream waits for the event)
 * mi355_sched_trace_read copies the text (NUL-terminated) into `out` when `cap` suffices and returns the size needed.  The GPU test
 * tests/test_gpu_schedules.py feeds it to a happens-before checker (tests/_sched_check.py): the multi-stream forwards must order every pair of
 * launches that touch the same bytes.  Not thread-safe; eager launches only (inside a stream capture events become graph edges). */
int mi355_sched_trace(int on);
long long mi355_sched_trace_read(char* out, long long cap);

/* ---- measurement: hipEvent brackets per kernel class, recorded on the launch stream ---------
 * enable(1) starts recording every launch of {attention, gemm, ln_modulate, sde_step, misc};
 * collect() waits for the events and returns summed elapsed milliseconds and launch counts (5 each). */
int mi355_profile_enable(int on);
/* A/B knob for kernel variants (key 0 = large-grid 256x256 GEMM kernel: 0 simple 2-stage; 1 (default) the 4-wave kernel with the
 *         hand-scheduled main loop where it applies (whole 256x256 tiles, K % 128 == 0) up to K = <key 19> (default 3072), else the 8-wave
 *         ping-pong kernel; 2 the 4-wave kernel wherever it applies; 3 the ping-pong kernel only.  Results are bit-identical for 1, 2, 3;
 * key 1 = attention softmax: 0 plain online softmax, 1 deferred rescale (default);
 * key 2 = hipGraph replay of the rollout loop: 0 eager launches, 1 captured graph (default);
 * key 3 = smallest 256x256-tile grid that takes the ping-pong kernel (default 128);
 * key 4 = VAE conv tile shape: 0 auto (default), 1 128x128, 2 256x128, 3 256x256, 4 512x128;
 * key 5 = head_dim-128 attention: 0 (default) = the 4-wave kernel with the hand-scheduled key loop where it applies (static softmax,
 *         self-attention), else 8-wave workgroups; 1 = 4-wave compiler-scheduled workgroups; 5 = never the hand-scheduled kernel (A/B);
 *         key 21 = the |score| bound mi355_op_attention128 asserts (0 = none: running-max kernel; unit tests and A/B);
 * key 6 = static-bound softmax (no running max when the q/k norm weights prove |score| <= 60): 1 on (default), 0 off,
 *         v >= 2: mi355_op_attention asserts the bound v itself (unit tests));
 * key 7 = GEMM tile raster: tile rows per band, walked column by column so that the tiles an XCD holds at any time form a near-square block
 *         (default 6; 0 = plain row-major order).  Results are bit-identical for every value;
 * key 8 = SD3.5 forward: the text-stream chain of every block (out-projection, LN-modulate, MLP, next q|k / V^T projections) on a
 *         plan-owned second stream, forked after each joint attention and joined before the next (graph edges inside the captured
 *         rollout): 0 = single stream, 1 = always, 2 (default) = when the image stream has at most <key 9> rows (default 32768).
 *         key 10 = fork point in dual-attention blocks: 1 after the block's last attention, 0 right after the joint attention, 2 (default)
 *         = 1 for plans with more than 16384 image rows, else 0.
 *         Results are bit-identical for every value.
 *  12/13  the same for the Qwen-Image engine (mi355_qwen_*): 0 = single stream, 1 = text chain of every block on a plan-owned side stream,
 *         2 (default) = when the image stream has at most <key 13> rows (default 16384).  Measured +3 ... +52 % (profiles/r03a_*).
 *  14/15  the same for the double blocks of the FLUX.1 engine (mi355_flux_*); default 2.  Measured +3 ... +22 %.
 *  16     FLUX.1 engine: 1 (default) = mi355_flux_rollout replays its N-step loop as one hipGraph (captured on the second call of a
 *         configuration, like key 2 for the SD3.5 engine); 0 = eager launches.  A failed capture is an error, not a fallback.
 *  17     the same for mi355_qwen_rollout (the prompt preparation, which uploads the per-sample key lengths, stays in front of the graph).
 *  19     largest K that key 0 = 1 gives to the 4-wave GEMM kernel (default 3072).
 *  22     optimize() replay (mi355_denoise_step_train / _backward): 1 (default) = the context-stream chain of every block on a side stream owned
 *         by the training state (the backward: in the default gradient scope; the context stream's weight gradients run there too, on
 *         their own scratch), 0 = in line.  Results are bit-identical for either value.
 *  24     Wan self-attention: 1 (default) = the kernel that stores q and k also measures their largest row norm per (batch, head), and every
 *         (batch, head) whose |q| |k| bound stays <= 60 runs the static-softmax hand-scheduled kernel (the others keep the running max);
 *         0 = the weight-side bound only (never satisfied by Wan's across-head RMSNorm: running-max kernel everywhere).
 *  25     optimize() backward: 1 (default since round 4) = the 16-byte-access forms of the attention-backward prep kernel and of the
 *         default-scope RMSNorm-backward gather (step 94.2 -> 91.2 ms at B = 2, 1024^2; verified at full width against the oracle's autograd
 *         and its bf16 band, profiles/r04a_*), 0 = the general forms.
 *  26-28  optimize() backward: weight-gradient GEMMs on a side stream (26), modelled split-K factor (27), text chain of the Qwen-Image / FLUX.1
 *         double-block backward on the plan's side stream (28): csrc/backward.hip.
 *  31     smallest 256x256-tile grid that key 0 = 1 gives to the 4-wave GEMM kernel (default 512: below it the ping-pong kernel, +2.5 % on the
 *         reference's 512^2 example rollouts, profiles/r05j_*).  Results are bit-identical for every value.
 *  29     MEASUREMENT ONLY (scripts/gpu_r5_call2.sh): bit mask of launches the SD3.5 forward SKIPS -- 1 the text-stream chain, 2 every
 *         LayerNorm-modulate, 4 the V^T projections -- to put a measured ceiling on what fusing them away could buy.  Results
 *         are WRONG by construction; 0 (default) = nothing skipped.
 *  32-37  mid-size GEMM kernel (128x192 / 192x128 tiles, round 6): 32 = 0 off, 1 by the cost rule (default), 2 wherever it applies; 33 margin of the
 *         rule in percent (100); 34 / 37 smallest / largest grid of its tiles (160 / 256: one-round grids); 35, 36 accepted and ignored (measurement
 *         knobs of round 6, removed).  Results are bit-identical for every value.
 *  38     optimize() backward: 1 (default) = the bias gradient's column-sum finish rides in the weight gradient's split-K reduction launch.
 *  39     optimize() backward: weight-gradient GEMMs on ROW-MAJOR operands (csrc/gemm_tn.hip; N, K multiples of 128, any M): 0 = transposed copies
 *         everywhere; 1 (default) = SD3.5: 128x128 tiles (image and text stream), the head_dim-128 engines: each engine's measured default
 *         (Wan on, FLUX.1 / Qwen-Image off: train_common.h); 2 = SD3.5: 256x256 tiles where they give a one-round grid (measured equal in the
 *         step), the other engines: row-major wherever it applies.  Same products; SD3.5 0 / 1 bit-identical.
 *  40-42  256x192-tile GEMM kernel (gemm_w6_kernel): 40 = 0 off (default: 16-20 % faster back to back, 2.4 % slower inside the two-stream forward),
 *         1 by its cost rule, 2 wherever it applies; 41 margin in percent (105); 42 smallest grid (200).  Bit-identical for every value.
 *  43     head_dim-64 attention backward: 1 (default) = the software-pipelined passes (csrc/gen_attn_bwd64.py), 0 = the round-3 kernels
 *         (bit-identical); 2..5 = ablation builds of the dK/dV loop (WRONG results; refused without MI355_ALLOW_ABLATION=1); 6 = the two-row-block
 *         form (64 keys / queries per wave, one wave per SIMD: bit-identical, measured equal -- csrc/gen_attn_bwd64x2.py).
 *  44     head_dim-128 attention backward (FLUX.1 / Qwen-Image / Wan): 1 (default) = the software-pipelined passes (csrc/gen_attn_bwd128.py),
 *         0 = the round-4 kernels (bit-identical).
 * The environment variable MI355_TUNE="key=value,..." applies these settings when the Python binding loads the library. */
int mi355_tune_set(int key, int value);
int mi355_profile_collect(double* ms_out, int64_t* count_out);

#ifdef __cplusplus
}
#endif
#endif /* MI355_FLOW_H */
