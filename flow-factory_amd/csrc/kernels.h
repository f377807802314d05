// mi355_flow -- host-visible kernel launchers (internal C++ API; the drop-in boundary is
// include/mi355_flow.h).  Every launcher enqueues on `stream` and never synchronises.
#pragma once
#include <initializer_list>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

namespace mi355 {

// ------------------------------------------------------------------------------------ GEMM
// C[m][n] = sum_k A[m][k] * W[n][k]   (both operands K-contiguous bf16, fp32 accumulate on
// v_mfma_f32_16x16x32_bf16); the epilogue decides what is written.  K % 64 == 0.
enum GemmEpi : int {
    EPI_BIAS = 0,       // out[m][n] = bf16(acc + bias[n])
    EPI_BIAS_SILU,      // out = bf16(silu(bf16(acc + bias)))
    EPI_BIAS_GELU,      // out = bf16(gelu_tanh(acc + bias))
    EPI_POSADD,         // out = bf16(acc + bias[n] + aux[(m % rows_per_sample)][n])   (patch-embed + pos-embed)
    EPI_ADDSRC_SILU,    // out = bf16(silu(bf16(acc + bias[n] + aux[m % rows_per_sample][n])))  (temb = t_emb + pooled_emb)
    EPI_GATE_RES,       // out[m][n] = bf16(out[m][n] + gate[m / rows_per_sample][n] * (acc + bias[n]))  (in place)
    EPI_QK_NORM,        // per-head RMSNorm of (acc + bias) then scatter to q/k [b][h][s][64]
    EPI_VT,             // rows = features, cols = tokens: vT[b][h][d][s] = bf16(acc + bias[m])
    EPI_UNPATCH,        // proj_out: scatter (token, (p,q,c)) -> latent [b][c][y][x] (bf16)
    // ---- VAE decode (vae_engine.hip); these three always run on the simple 2-stage kernel
    EPI_BIAS_ROW,       // out[m][n] = bf16(acc + bias[m])              (row-major; swapped-operand V^T of the mid attention)
    EPI_F32,            // out_f32[m][n] = acc * q_scale                (attention scores, no bias)
    EPI_IMG,            // conv_out: image[b][n][y][x] = post(acc + bias[n]), n < N <= 4, fp32 or bf16 NCHW
    // ---- backward (engine_train.inc)
    EPI_DGELU,          // out[m][n] = bf16((acc + bias[n]) * gelu_tanh'(aux[m][n]))   (aux = stashed pre-activation, row stride ld_aux)
    EPI_QK_NORM_RSTD,   // EPI_QK_NORM that also stores the per-(row, head) 1/rms to rstd_out (training-mode forward; selected by launch_gemm)
    EPI_COUNT
};

struct GemmParams {
    const bf16_t* A; long lda;
    const bf16_t* W; long ldw;
    int M, N, K;
    int epi;
    const float* bias;        // [N], or [M] for EPI_VT
    bf16_t* out; long ldo;
    const bf16_t* aux; long ld_aux;   // POSADD: table; ADDSRC: src; GATE_RES: gate base (row stride ld_aux per sample)
    int rows_per_sample;      // tokens per sample of the row (or column, EPI_VT) dimension
    // EPI_QK_NORM / EPI_VT scatter
    bf16_t* q; bf16_t* k;     // q,k: [B][H][S_pad][64];  EPI_VT: q = vT [B][H][64][S_pad]
    const float* nw_q; const float* nw_k;   // RMSNorm weights [64]
    int H, S_pad, s_off; float eps;
    int hd_shift;             // EPI_VT: log2(head_dim) (0 = 6, head_dim 64)
    float q_scale;            // EPI_QK_NORM: extra factor on the normalised q (softmax scale folded in); 0 = 1.0
    // EPI_UNPATCH
    int hp, wp, patch, out_ch;
    // EPI_F32 / EPI_IMG
    float* out_f32;           // EPI_F32 target; EPI_IMG: fp32 image (or nullptr -> bf16 image in `out`)
    int img_post;             // EPI_IMG: 1 = (x/2 + 0.5).clamp(0,1)
    int img_clamp;            // EPI_IMG: 1 = clamp the decoder output to [-1, 1] first (video VAE)
    // implicit 3x3 convolution, padding 1 (conv_cin > 0): A is an NHWC activation tensor [B][Hin][Win][conv_cin], row m of the
    // GEMM is output pixel (b, y, x) of a conv_h x conv_w image, K = 9*conv_cin with k = tap*conv_cin + c (tap = ky*3 + kx);
    // conv_up = 1: the input is (conv_h/2) x (conv_w/2) and is nearest-2x upsampled on the fly (Upsample2D + conv)
    int conv_cin, conv_h, conv_w, conv_up;
    // causal 3-D extension (video VAE): rows are (b, t, y, x) with conv_t output frames per sample; the input holds conv_t_in frames per
    // sample (>= conv_t; the A pointer may be advanced by whole frames); conv_kt temporal taps reach BACK in time (frame t + j - (kt-1),
    // zeros before frame 0); conv_ks = spatial kernel size (3 or 1).  K = kt*ks*ks*conv_cin, k = ((j*ks + ky)*ks + kx)*conv_cin + c.
    // All zero = the plain 2-D 3x3 convolution above.
    int conv_t, conv_t_in, conv_kt, conv_ks;
    const bf16_t* zero_page;  // >= 128 B of zeros: source of the padding taps
    // training-mode forward (the main output is bit-identical to the rollout's): EPI_BIAS_GELU also stores the pre-activation
    // bf16(acc + bias) to `stash` (row stride ld_stash); EPI_QK_NORM stores the per-(row, head) 1/rms to rstd_out[m * 2H + head]
    bf16_t* stash; long ld_stash;
    float* rstd_out;
    // split-K (EPI_F32 on the 2-stage kernel only: weight gradients, output tiles << CUs): split s of k_split accumulates K-tiles
    // [s*nt/k_split, (s+1)*nt/k_split) into out_f32 + s * split_stride (deterministic second-stage reduction by the caller)
    int k_split; long split_stride;
    int raster_gm;            // tile rows per raster band (set by launch_gemm from the global knob)
    // optional s_memtime trace (debug): per workgroup, per tile 4 stamps {tile start, main loop start, main loop end, epilogue end}
    long long* trace;
    int dbg_skip_prefetch;    // debug ablation: the K-loop prefetches are not issued (results are garbage)
};

// weight-gradient GEMM on row-major operands (gemm_tn.hip): out[n][k] (fp32, + split * split_stride) = sum_m A[m][n] * B[m][k]
struct GemmTnParams {
    const bf16_t* A; long lda;        // dY [M][lda], the N columns of interest starting at A
    const bf16_t* B; long ldb;        // X  [M][ldb], K columns
    int M, N, K;
    float* out; long ldo;             // fp32 [N][ldo]
    int k_split; long split_stride;   // split over m, like GemmParams
    float* colsum;                    // optional [k_split][N] fp32: per-split column sums of A (the bias gradient's partials); nullptr = not taken
    int tile256;                      // 1 = the 256 x 256-tile form (8 waves, one workgroup per CU) where N and K are multiples of 256
    int M_pad = 0;                    // 0 = M rounded up to 64; else the padded length (a multiple of 64, >= M) the split boundaries are taken on
    const void* zeros = nullptr;      // >= 16 bytes of zeros, 16-byte aligned, for the rows [M, M_pad) (nullptr: the launcher's own block)
};
bool gemm_tn_ok(const GemmTnParams& p);
void set_wgrad_tn_mode(int v);      // mi355_tune_set(39, .) for the head_dim-128 engines' shared weight-gradient path (train_common.h): 0 = transposed copies
int get_wgrad_tn_mode();
hipError_t launch_gemm_tn(const GemmTnParams& p, hipStream_t stream);

// ---- launch-schedule trace (sched_trace.hip; off unless mi355_sched_trace(1)): every launch_* / event call reports (stream, regions)
struct TraceRegion { const void* p; size_t len; size_t stride; size_t count; };     // `count` blocks of `len` bytes, `stride` apart
inline TraceRegion treg(const void* p, size_t len) { return TraceRegion{p, len, 0, 1}; }
inline TraceRegion tregs(const void* p, size_t len, size_t stride, size_t count) { return TraceRegion{p, len, stride, count}; }
bool sched_trace_on();
void sched_trace_launch(const char* name, hipStream_t st, std::initializer_list<TraceRegion> reads, std::initializer_list<TraceRegion> writes);
void sched_trace_event(int kind, hipStream_t st, hipEvent_t ev);      // kind 0: record, 1: wait
// event calls of the engines go through these two (same semantics, plus the trace)
inline hipError_t ev_record(hipEvent_t ev, hipStream_t st) { sched_trace_event(0, st, ev); return hipEventRecord(ev, st); }
inline hipError_t ev_wait(hipStream_t st, hipEvent_t ev) { sched_trace_event(1, st, ev); return hipStreamWaitEvent(st, ev, 0); }
inline hipError_t copy_d2d(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (sched_trace_on()) sched_trace_launch("copy", st, {treg(src, bytes)}, {treg(dst, bytes)});
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
}

// bumped by every mi355_tune_set: part of the captured-graph key of every engine, so that an A/B through the knobs never replays a graph
// captured under the previous kernel selection (engine.hip)
int tune_epoch();
hipError_t launch_gemm(const GemmParams& p, hipStream_t stream);
void set_gemm_variant(int v);
void set_w4_max_k(int k);      // largest K that the default dispatch gives to the 4-wave hand-scheduled kernel (mi355_tune_set(19, k))
int get_gemm_variant();
void set_pp_min_tiles(int v);
void set_w4_min_tiles(int v);
void set_w6_mode(int v);            // 256x192 GEMM kernel: 0 off (default), 1 cost rule, 2 wherever it applies
void set_w6_alpha_percent(int v);
void set_w6_min_tiles(int v);
void set_mid_mode(int v);
void set_mid_alpha_percent(int v);
void set_mid_min_tiles(int v);
void set_mid_max_tiles(int v);
void set_mid_plan_hint(int v);      // engine.hip forward_core: whether this plan takes the mid-size GEMM kernel at all
void set_raster_gm(int v);      // GEMM tile raster: tile rows per band (0 = row-major)
void set_conv_cfg(int v);       // VAE conv tile shape A/B knob (0 auto)
void set_attn_variant(int v);  // 0 = plain online softmax, 1 = deferred-rescale (default)
int get_attn_variant();  // tuning / A-B knob: 0 = simple 2-stage 256x256 kernel, 1 = ping-pong (default)

// ------------------------------------------------------------------------------- attention
// Non-causal softmax(q k^T / 8) v over S keys, head_dim 64.  q,k: [B][H][S_pad][64],
// vT: [B][H][64][S_pad] (keys >= S must be finite).  Output rows: query s < n_img goes to
// o_img[(b*n_img + s)][h*64 + d], the rest to o_ctx[(b*(S-n_img) + s-n_img)][h*64 + d].
struct AttnParams {
    const bf16_t* q; const bf16_t* k; const bf16_t* vT;
    bf16_t* o_img; bf16_t* o_ctx;
    int B, H, S, S_pad, n_img;
    int q_prescaled;   // 1: q already carries the softmax scale 0.125*log2(e) (folded into the q RMSNorm epilogue)
    float score_bound; // > 0: proven bound on |q.k| * scale * log2(e); <= 60 selects the no-running-max kernel (0 = unknown)
    float* lse;        // optional [B][H][S_pad] fp32: log2-sum-exp of the scaled scores per query (training-mode forward; nullptr = not written)
};
hipError_t launch_attention(const AttnParams& p, hipStream_t stream);

// head_dim 128 (FLUX.1, attention128.hip): q,k [B][H][S_pad][128], vT [B][H][128][S_pad].  Query s < n_first of sample b is
// written to o_first[(b*n_first + s)*ld_first + h*128 + d], the others to o_rest[(b*(S-n_first) + s-n_first)*ld_rest + ...].
struct Attn128Params {
    const bf16_t* q; const bf16_t* k; const bf16_t* vT;
    bf16_t* o_first; long ld_first; int n_first;
    bf16_t* o_rest; long ld_rest;
    int B, H, S, S_pad;
    int q_prescaled;   // 1: q carries log2(e)/sqrt(128) already (rope_norm kernel)
    float score_bound; // > 0: proven bound on |score| (log2 domain); <= 60 selects the no-running-max kernel
    int S_kv, S_kv_pad; // cross-attention: keys / values are a different sequence (k [B][H][S_kv_pad][128], vT [B][H][128][S_kv_pad]); 0 = self
    const int* kv_len;  // optional device [B]: sample b attends to keys [0, kv_len[b]) only (ragged text at the END of the joint sequence)
    // optional device [B*H] each: the largest squared row norm of this launch's stored q (k) per (batch, head), as its producer measured it
    // (norm_rope_full).  |score| <= |q| |k| then bounds every score of a (b, h) from the DATA: launch_attention128 launches the static-softmax
    // kernel and the running-max kernel, and each workgroup runs or exits by its own (b, h)'s bound (`mode` below).
    const unsigned* qmax2; const unsigned* kmax2;
    int mode;           // set by launch_attention128: 0 = always run, 1 = run only where the measured bound <= 60, 2 = only where it is larger
    float* lse;         // optional [B][H][S_pad] fp32: log2-sum-exp of the scaled scores per query (training-mode forward).  A RUNTIME branch in the
                        // kernels' epilogues, not a second instantiation: rollout and training run the same binary, so their outputs agree bit for bit
};
hipError_t launch_attention128(const Attn128Params& p, hipStream_t stream);
void set_attn128_variant(int v);   // 0 (default): 4-wave hand-scheduled kernel where it applies, else 8-wave workgroups; 1: 4-wave compiler-scheduled; 5: never hand-scheduled
int get_attn128_variant();
void set_attn128_op_bound(int v);  // mi355_op_attention128 passes this as the proven |score| bound (0 = none); mi355_tune_set(21, v)
int get_attn128_op_bound();
void set_wan_data_bound(int v);            // wan_engine.hip: data-dependent static / running-max split of the self-attention (1 = default)
void set_qwen_two_stream(int mode);        // qwen_engine.hip: text chain on a side stream (0 off = default, 1 on, 2 auto by image rows)
void set_qwen_two_stream_rows(int rows);
void set_qwen_graph(int on);               // qwen_engine.hip: replay the N-step loop of mi355_qwen_rollout as one hipGraph (0 = default: eager)
void set_flux_two_stream(int mode);        // flux_engine.hip: the same for the FLUX.1 double blocks
void set_flux_two_stream_rows(int rows);
void set_flux_graph(int on);               // flux_engine.hip: replay the N-step loop of mi355_flux_rollout as one hipGraph (0 = default: eager)

// per-head RMSNorm (weight, eps) + rotary embedding of the q and k projections (flux_ops.hip):
//   src rows [M][src_ld]: q at column q_col + h*128, k at k_col + h*128 (bf16, bias already added);
//   row m = (sample b, token m % rows_per_sample) -> joint position s = s_off + token; rot = cs[s][pair] = (cos, sin);
//   q_out / k_out [B][H][S_pad][128]; q additionally multiplied by q_scale (softmax scale * log2 e).
struct RopeNormParams {
    const bf16_t* src; long src_ld; int q_col, k_col;
    const float* nw_q; const float* nw_k;     // [128]
    const float2* cs;                         // [S_joint][64]
    bf16_t* q_out; bf16_t* k_out;
    int M, H, rows_per_sample, s_off, S_pad;
    float eps, q_scale;
    float* rstd_out;                          // optional [M][2H] fp32: 1 / rms per (row, head) of q ([0, H)) and k ([H, 2H)) (training-mode forward; runtime branch)
};
hipError_t launch_rope_norm(const RopeNormParams& p, hipStream_t stream);

// Wan: RMSNorm ACROSS heads (over the whole row of H*128 features, weight [H*128]) + optional rotary embedding (cs == nullptr: none)
// of ONE projection: src rows [M][src_ld] at column `col` -> out [B][H][S_pad][128] (row m = sample m / rows_per_sample, position
// s_off + m % rows_per_sample), multiplied by out_scale.
struct NormRopeFullParams {
    const bf16_t* src; long src_ld; int col;
    const float* weight;                      // [H*128]
    const float2* cs;                         // [S][64] (cos, sin) or nullptr
    bf16_t* out;
    int M, H, rows_per_sample, s_off, S_pad;
    float eps, out_scale;
    unsigned* max2;                           // optional [B*H]: largest squared norm of the stored (bf16-rounded) output rows per (batch, head), as float bits
    float* max2_part;                         // scratch for it: [B][norm_rope_parts(rows_per_sample)][H] floats (s_off must be 0)
    float* rstd_out = nullptr;                // optional [M]: 1 / rms of every row (training-mode forward: the backward needs it)
};
hipError_t launch_norm_rope_full(const NormRopeFullParams& p, hipStream_t stream);
// backward of launch_norm_rope_full for ONE tensor (Wan q / k: RMSNorm over the whole row of H * 128 features, then RoPE, then out_scale):
// from the STORED output y~ [B][H][S_pad][128] (= the forward's `out`), its gradient dy~ (same layout), the row's 1 / rms and the weight:
//   z = y~ / out_scale;  y = R^T z;  xhat = y / w;  g = (R^T dy~ * out_scale) * w;  dx = rstd * (g - xhat * mean_row(g * xhat))
// written token-major to out[m][col .. col + H * 128).  weight == nullptr: plain gather (dx = dy~: the V gradient).
struct NormRopeFullBwdParams {
    const bf16_t* y; const bf16_t* dy;         // stored output and its gradient, head-major
    const float* weight; const float* rstd;    // [H*128], [M]
    const float2* cs;                          // rotary table or nullptr
    bf16_t* out; long out_ld; int col;
    int M, H, rows_per_sample, s_off, S_pad;
    float out_scale;
};
hipError_t launch_norm_rope_full_bwd(const NormRopeFullBwdParams& p, hipStream_t stream);
int norm_rope_parts(int rows_per_sample);
// Qwen-Image: RMSNorm over whole rows (weight fp32 [D]) and the norm-rescaled true-CFG combine over C = 64 channel tokens (flux_ops.hip)
hipError_t launch_rms_rows(const bf16_t* x, long ldx, const float* w, bf16_t* out, long ldo, int M, int D, float eps, hipStream_t stream);
hipError_t launch_cfg_rescale(const bf16_t* neg, const bf16_t* pos, float g, bf16_t* out, long rows, int C, hipStream_t stream);
hipError_t launch_cfg_rescale_bwd(const bf16_t* neg, const bf16_t* pos, float g, const float* dout, bf16_t* dneg, bf16_t* dpos, long rows, int C,
                                  hipStream_t stream);
// out[m][j*W + x] = bf16(a[m][x] + table[j][x])   (a bf16 [rows][W], table fp32 [J][W]): Wan modulation = scale_shift_table + time_proj
hipError_t launch_bcast_add(const bf16_t* a, const float* table, bf16_t* out, long out_ld, int rows, int W, int J, hipStream_t stream);
// LayerNorm affine (weight, bias fp32 [D]) -> the (shift, scale) row layout of ln_mod: out[0][d] = bias, out[1][d] = weight - 1
hipError_t launch_affine_to_mod(const float* weight, const float* bias, bf16_t* out, int D, hipStream_t stream);

// ------------------------------------------------------------------------------ elementwise
// LayerNorm(no affine, eps) + AdaLN modulate: out = LN(x)*(1+scale[b]) + shift[b]; optional
// second modulated copy (dual blocks).  x,out: [M][D] bf16; mod vectors bf16 at
// mod + b*mod_ld + {shift,scale}_off.
struct LnModParams {
    const bf16_t* x; bf16_t* out; bf16_t* out2;
    const bf16_t* mod; long mod_ld;
    int shift_off, scale_off, shift2_off, scale2_off;
    int M, D, rows_per_sample; float eps;
};
hipError_t launch_ln_mod(const LnModParams& p, hipStream_t stream);

// latents [B][C][h][w] (storage dtype) -> patches [B*hp*wp][C*p*p] bf16 (k = c*p*p + py*p + px);
// `rep` > 1 replicates the batch (CFG: latents_input = cat([latents, latents])).
hipError_t launch_patchify(const void* lat, int dt, bf16_t* patches, int B, int rep, int C, int h, int w, int p,
                           hipStream_t stream);
// pos_embed buffer [max*max][D] bf16 -> centre crop [hp*wp][D]
hipError_t launch_pos_crop(const bf16_t* pos, bf16_t* out, int max_size, int hp, int wp, int D, hipStream_t stream);
// sinusoidal timestep projection (cos first): out[r][dim] bf16, t rounded to `t_round_dt` first
hipError_t launch_time_proj(const float* t, int rows, int dim, int t_round_dt, bf16_t* out, hipStream_t stream);
// generic dtype conversion (weights binding); n elements
hipError_t launch_convert(const void* src, int src_dt, void* dst, int dst_dt, long n, hipStream_t stream);
hipError_t launch_clock_probe(long long* out, int n, int sleep_iters, hipStream_t stream);   // measurement tool: {core clock, wall clock} samples

// ------------------------------------------------------------------------------ VAE decode (vae.hip)
// latents [B][C][HW] (storage dtype) -> NHWC bf16 [B*HW][Cpad] = bf16(lat / scale + shift), channels >= C zero
hipError_t launch_vae_ingest(const void* lat, int dt, bf16_t* out, int B, int C, int Cpad, long HW, float scale, float shift,
                             hipStream_t st);
// GroupNorm over NHWC bf16 [B][HW][C] (+ optional SiLU) -> y (may alias x).  part: >= B*gn_num_chunks*C*2 floats of scratch,
// ad: B*2*C floats of scratch (per-channel affine).  Deterministic (fixed-order partial sums).
int gn_num_chunks(long HW, int C);
hipError_t launch_group_norm(const bf16_t* x, bf16_t* y, const float* gamma, const float* beta, float* part, float* ad, int B, long HW,
                             int C, int groups, float eps, bool silu, hipStream_t st);
// p[r][:] = softmax(scale * s[r][:]) for `rows` rows of n fp32 scores (n % 4 == 0), bf16 out
hipError_t launch_softmax_rows(const float* s, bf16_t* p, long rows, int n, float scale, hipStream_t st);
// conv weight [Co][Ci][taps] (any dtype) -> bf16 [Co][taps][Cpad] (k = tap*Cpad + ci; ci >= Ci zero); taps = 1 for 1x1 / linear
hipError_t launch_conv_repack(const void* src, int dt, bf16_t* dst, int Co, int Ci, int Cpad, int taps, hipStream_t st);
// ---- video VAE (causal 3-D; vae.hip)
hipError_t launch_softmax_rows_ld(const float* s, long ld_s, bf16_t* p, long ld_p, long rows, int n, float scale, hipStream_t st);
struct WvaeIngestParams {
    int B, C, T, Cpad, denorm; long HW;
    float mean[16], std[16];
    const float* w_pq; const float* b_pq;     // post_quant_conv [C][C], [C] fp32
};
hipError_t launch_wvae_ingest(const void* lat, int dt, bf16_t* out, const WvaeIngestParams& q, hipStream_t st);
hipError_t launch_wan_rms(const bf16_t* x, bf16_t* y, const float* gamma, long M, int C, int Cpad, bool silu, hipStream_t st);
hipError_t launch_frame_interleave(const bf16_t* x, const bf16_t* tc, bf16_t* out, int B, int T, long HW, int C, hipStream_t st);
// shared error sink of the C ABI (engine.hip): formats into mi355_last_error(), returns 1
int errorf(const char* fmt, ...);

// ------------------------------------------------------------------------------ backward (backward.hip, attention_bwd.hip)
// dres (+)= d/dx of [LN(x)*(1+scale)+shift](dy) [+ the same for a second modulated copy (dy2, scale2)]; x, dy, dres [M][D] bf16
struct LnModBwdParams {
    const bf16_t* x; const bf16_t* dy; const bf16_t* dy2; bf16_t* dres;
    const bf16_t* mod; long mod_ld; int scale_off, scale2_off;
    int M, D, rows_per_sample; float eps; int accumulate;
    // optional: gradients of the modulation vectors, ADDED (fp32 atomics) into dmod[b * mod_ld + {shift,scale}_off + d]:
    // dshift = sum_tokens dy, dscale = sum_tokens dy * xhat  (and the same for the second modulated copy)
    float* dmod; int shift_off, shift2_off;
};
hipError_t launch_ln_mod_bwd(const LnModBwdParams& p, hipStream_t stream);
// dy[m][:] = gate[m / rps][:] * dx[m][:]
hipError_t launch_gate_mul(const bf16_t* dx, const bf16_t* gate, long gate_ld, bf16_t* dy, long M, int D, int rps, hipStream_t stream);
// the same plus dgate[b][n] += sum_{m in sample b} dx[m][n] * y[m][n]  (y = the stashed un-gated projection; fp32 atomics into dgate, row stride dg_ld)
hipError_t launch_gate_bwd(const bf16_t* dx, const bf16_t* gate, long gate_ld, const bf16_t* y, bf16_t* dy, float* dgate, long dg_ld, long M, int D,
                           int rps, hipStream_t stream);
// dx = dy * silu'(pre)  (bf16, n % 8 == 0)
hipError_t launch_silu_bwd(const bf16_t* dy, const bf16_t* pre, bf16_t* dx, long n, hipStream_t stream);
// fp32 -> bf16 rows with zero padding: out[r][c] = r < rows ? in[r][c] : 0 for r < rows_pad
hipError_t launch_f32_to_bf16_pad(const float* in, bf16_t* out, long rows, long rows_pad, long cols, hipStream_t stream);
hipError_t launch_gelu_fwd(const bf16_t* pre, bf16_t* out, long n, hipStream_t stream);
// out[z][c][r] = in[z][r][c] for r < rows (0 for rows <= r < rows_pad), 64 x 64 tiles through LDS
hipError_t launch_transpose(const bf16_t* in, long ld_in, long bs_in, bf16_t* out, long ld_out, long bs_out, int rows, int cols, int rows_pad,
                            int batch, hipStream_t stream);
hipError_t launch_transpose_colsum(const bf16_t* in, long ld_in, bf16_t* out, long ld_out, int rows, int cols, int rows_pad, float* scratch,
                                   float* colsum, hipStream_t stream, bool finish = true);
hipError_t launch_colsum_finish(const float* scratch, int nslab, int cols, float* colsum, hipStream_t stream);
hipError_t launch_splitk_reduce_colsum(const float* part, long stride, int nsplit, float* out, long n, const float* cs_scratch, int nslab, int cols,
                                       float* colsum, hipStream_t stream);
struct AttnBwdPrepParams {
    const bf16_t* o_img; const bf16_t* o_ctx; const bf16_t* do_img; const bf16_t* do_ctx;   // token-major [.][H*64]; do_ctx may be null (zeros)
    const float* lse;                                                                     // [B][H][S_pad] from the forward
    bf16_t* doh; float* delta; float* nld;                                                // [B][H][S_pad][64], [B][H][S_pad], [B][H][S_pad/64][2][64]
    int B, H, S, S_pad, n_img;
};
hipError_t launch_attn_bwd_prep(const AttnBwdPrepParams& p, hipStream_t stream);
struct RmsBwdParams {
    const bf16_t* q; const bf16_t* k;                      // stored (normalised) q~, k [B][H][S_pad][64]
    const bf16_t* dq; const bf16_t* dk; const bf16_t* dv;  // gradients in the same layout
    const float* rstd_img; const float* rstd_ctx;          // [rows][2H] from the forward's q|k epilogue
    const float* nw_q; const float* nw_k; const float* nw_cq; const float* nw_ck;   // RMSNorm weights [64] (image / context stream)
    float q_scale;                                         // factor folded into the stored q (softmax scale * log2 e)
    bf16_t* out_img; bf16_t* out_ctx;                      // [rows][3*H*64] = [dq_pre | dk_pre | dv]
    int B, H, S, S_pad, n_img;
    float* dw_part;                                        // optional [gridDim.x][4][64]: per-workgroup partial RMSNorm-weight gradients
                                                           // (norm_q, norm_k, norm_added_q, norm_added_k), summed by launch_rms_dw_finish
};
// dw[which][64] = sum over workgroups of part[wg][which][64] (fixed order); outputs may be null; q gradients are w.r.t. the UNSCALED weight
void set_rms_bwd_fast(int v);
hipError_t launch_rms_dw_finish(const float* part, int nwg, float q_scale, float* dw_q, float* dw_k, float* dw_cq, float* dw_ck, hipStream_t stream);
int rms_bwd_grid(int B, int S);
hipError_t launch_rms_bwd_gather(const RmsBwdParams& p, hipStream_t stream);
// out[n] (+)= sum_m dy[m][n]; scratch >= 64 * N floats
hipError_t launch_colsum(const bf16_t* dy, long ld, long M, int N, float* scratch, float* out, int accumulate, hipStream_t stream,
                         int* deferred_nslab = nullptr);      // non-null: only the slab partials; *deferred_nslab slabs are left for the caller's finish
hipError_t launch_splitk_reduce(const float* part, long stride, int nsplit, float* out, long n, int accumulate, hipStream_t stream);
// gradient-buffer dtype registry (backward.hip): launch_splitk_reduce / launch_colsum / launch_transpose_colsum write bf16 into an output
// pointer marked DT_BF16 (the pointer is still passed as float*); DT_F32 un-marks
void grad_buf_mark(const void* p, int dt);
int grad_buf_dtype(const void* p);
int grad_buf_esize(const void* p);
void set_wgrad_side(int v);
int get_wgrad_side();
// split-K factor of a weight-gradient GEMM dW[N][K] = aT[N][M_pad] . xT[K][M_pad]^T (EPI_F32 through launch_simple) -- backward.hip
void set_train_text_side(int v);      // key 28
int get_train_text_side();
void set_wgrad_split_model(int v);
int wgrad_split(int N, int K, int M_pad, size_t part_floats, bool overlapped);
hipError_t launch_unpatch_bwd(const float* dv, bf16_t* out, int Bp, int C, int hp, int wp, int patch, hipStream_t stream);
struct SdeBwdParams {
    const bf16_t* v_text; const bf16_t* v_uncond; float guidance;
    const void* latents; int lat_dt; const void* next_in; int next_in_dt;
    const float* sigma; const float* sigma_next; const float* eta; int scalar_stride; float sigma_max;
    int dynamics; int compute_log_prob; int B; long n;
    const float* g_lp; const float* g_np; const float* g_mean;     // upstream gradients (any may be null)
    float* dv;                                                     // [n_cfg*B][n] fp32, order [uncond, text]
};
hipError_t launch_sde_step_bwd(const SdeBwdParams& p, hipStream_t stream);
// flash-attention backward, head_dim 64 (attention_bwd.hip).  q (pre-scaled by log2(e)/8), k, v, doh: [B][H][S_pad][64];
// lse (log2 domain), delta: [B][H][S_pad] fp32.  Outputs dq (w.r.t. the stored, pre-scaled q), dk, dv [B][H][S_pad][64] bf16.  Two deterministic passes: key-block-outer (dk, dv) and query-block-outer (dq).
struct AttnBwdParams {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* doh;     // [B][H][S_pad][64] each (q pre-scaled; v row-major; doh = dO head-major)
    const float* lse; const float* delta;                                     // [B][H][S_pad]
    const float* nld;                                                         // [B][H][S_pad / 64][2][64] = -lse | -delta per 64-query tile
    bf16_t* dq; bf16_t* dk; bf16_t* dv;
    int B, H, S, S_pad;
    const int* kv_len;        // head_dim-128 kernels only: optional device [B], sample b attends to keys [0, kv_len[b]) (ragged text at the END of
                              // the joint sequence, like Attn128Params::kv_len); dk / dv of the masked keys are written as zeros
    int S_kv = 0, S_kv_pad = 0;   // head_dim-128 kernels only: cross-attention -- k, v, dk, dv are [B][H][S_kv_pad][128] over S_kv keys (0 = self)
};
hipError_t launch_attention_bwd(const AttnBwdParams& p, hipStream_t stream);
void set_attn_bwd_pipe(int v);
void set_attn128_bwd_pipe(int v);   // head_dim-128 backward: 1 = software-pipelined passes (default), 0 = the round-4 kernels      // head_dim-64 backward: 1 = software-pipelined passes (default), 0 = the round-3 kernels
// head_dim 128 (attention128_bwd.hip): the same contract with [B][H][S_pad][128] operands
hipError_t launch_attention128_bwd(const AttnBwdParams& p, hipStream_t stream);
struct Attn128BwdPrepParams {
    // o / dO token-major, split like Attn128Params' output: positions s < n_first of sample b at (b * n_first + s) * ld + h * 128, the others
    // at (b * (S - n_first) + s - n_first) * ld + h * 128; do_first / do_rest may be null (that part of the output has no consumer: zeros)
    const bf16_t* o_first; long ld_o_first; const bf16_t* o_rest; long ld_o_rest;
    const bf16_t* do_first; long ld_do_first; const bf16_t* do_rest; long ld_do_rest;
    int n_first;
    const float* lse;                                                                     // [B][H][S_pad] from the forward
    bf16_t* doh; float* delta; float* nld;                                                // [B][H][S_pad][128], [B][H][S_pad], [B][H][S_pad/64][2][64]
    int B, H, S, S_pad;
};
hipError_t launch_attn128_bwd_prep(const Attn128BwdPrepParams& p, hipStream_t stream);
// backward of rope_norm (per-head RMSNorm + RoPE of q, k) + gather of (dq~, dk, dv) [B][H][S_pad][128] to token-major rows
// out[m][0 .. 3 H 128) = [dq_pre | dk_pre | dv] (row stride ld_out) of the M rows of one stream (row m = sample m / rows_per_sample,
// joint position s_off + m % rows_per_sample)
struct RopeRmsBwdParams {
    const bf16_t* q; const bf16_t* k;                      // stored q~ (normalised, rotated, scaled), k
    const bf16_t* dq; const bf16_t* dk; const bf16_t* dv;  // gradients in the same layout (dq w.r.t. the stored q~)
    const float* rstd;                                     // [M][2H] from the forward (RopeNormParams::rstd_out)
    const float* nw_q; const float* nw_k;                  // RMSNorm weights [128]
    const float2* cs;                                      // [S_joint][64] (cos, sin)
    float q_scale;
    bf16_t* out; long ld_out;
    int M, H, rows_per_sample, s_off, S_pad;
};
hipError_t launch_rope_rms_bwd128(const RopeRmsBwdParams& p, hipStream_t stream);

// --------------------------------------------------------------------------- SDE step (K15)
enum Dynamics : int { DYN_ODE = 0, DYN_FLOW_SDE = 1, DYN_DANCE_SDE = 2, DYN_CPS = 3 };
struct SdeStepParams {
    const bf16_t* v_text;     // network output (bf16) [B][n]; with CFG: text branch
    const bf16_t* v_uncond;   // nullptr => no CFG
    int v_dt;                 // dtype of v_text / v_uncond (DT_BF16 for the engine's own output; fp32 / fp16 predictions
                              // of standalone scheduler.step() callers are read exactly, like the reference's noise_pred.float())
    float guidance;
    const void* latents; int lat_dt;          // x_i, storage dtype
    const float* noise;                       // eps [B][n] fp32 (ignored when next_in != nullptr)
    const void* next_in; int next_in_dt;      // replay: provided x_{i+1} (not re-rounded)
    const float* sigma; const float* sigma_next; const float* eta;   // device scalars
    int scalar_stride;                        // 0: one value for the whole batch, 1: per sample
    float sigma_max;
    int dynamics; int compute_log_prob;
    int B; long n;                            // elements per sample
    void* next_out; int next_out_dt;          // x_{i+1} in the storage dtype (may be nullptr)
    float* next_f32;                          // optional fp32 (value-rounded) copy
    float* mean_out;                          // optional next_latents_mean fp32
    float* noise_pred_out;                    // optional CFG-combined noise_pred as fp32
    float* log_prob; float* std_dev_t; float* dt_out;   // [B] (optional)
};
hipError_t launch_sde_step(const SdeStepParams& p, hipStream_t stream);
// UniPC multistep update (sde_step.hip): out = sum_i round_i(c_i * t_i), at most 5 terms, n % 4 == 0; x0 = sample - round(sigma * CFG-combine(v))
hipError_t launch_lincomb(int n_terms, const void* const* t, const int* dt, const float* c, void* out, int out_dt, long n, hipStream_t stream);
hipError_t launch_unipc_convert(const void* v_text, const void* v_uncond, int v_dt, float guidance, const void* sample, int sample_dt, float sigma,
                                float* x0, long n, hipStream_t stream);

}  // namespace mi355
