// mi355_flow -- Wan2.1 text-to-video rollout engine behind the C ABI (include/mi355_flow.h, mi355_wan_*): SURVEY.md 8(f) row N4.
// Replaces the cond / uncond `transformer(...)` passes, the CFG combine and `scheduler.step(...)` inside the denoising loop of
// Wan2_T2V_Adapter.inference / .forward (reference src/flow_factory/models/wan/wan2_t2v.py:344-376, :426-543) for the single-
// transformer Wan2.1 configuration (no boundary_ratio / transformer_2, no expand_timesteps).
//
// Layout: the video latents (B, 16, T, h, w) are patchified per frame (k = s = (1,2,2)) to S = T*(h/2)*(w/2) tokens of 64 features --
// the image patchify / un-patchify kernels on a (T*h) x w "image".  Per rollout (step-invariant, hoisted out of the loop): the text
// embedder AND every block's cross-attention keys / values (norm_k(to_k(ctx)), to_v(ctx) depend on the prompt only; the reference
// recomputes them in all N steps), the time MLP, time_proj and the per-block modulation tables for all N steps.  Per step and block:
// modulated LayerNorm -> q|k GEMM -> RMSNorm across heads + 3-D RoPE -> V^T GEMM -> head_dim-128 attention -> gated out-projection;
// affine LayerNorm -> q GEMM -> RMSNorm -> cross-attention on the cached text K / V^T -> out-projection (+residual);
// modulated LayerNorm -> gelu-tanh feed-forward, gated.  CFG runs as one forward over the batch [negative, positive].
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/mi355_flow.h"
#include "engine_common.h"
#include "train_common.h"

using namespace mi355;


namespace {

struct WSlot { void* dst; int dst_dt; int64_t numel; bool bound; };

// per-block activation stash of the training-mode forward (wan_train.inc): what the backward of the block needs
struct WTrainBlk {
    bf16_t *x_in, *xn1, *o1, *x_mid1, *xn2, *o2, *x_mid2, *xn3, *pre;     // [M][D] each; pre [M][F]
    bf16_t *q, *k, *vT, *q2;                                              // [Bp][H][S_pad][128] (vT: [Bp][H][128][S_pad])
    float *lse1, *lse2, *rstd_q, *rstd_k, *rstd_q2;                       // [Bp][H][S_pad] x 2, [M] x 3
};

struct WanBlockW {
    bf16_t *w_qk, *w_v, *w_o, *w_q2, *w_kv2, *w_o2, *w_ff1, *w_ff2;
    float *b_qk, *b_v, *b_o, *b_q2, *b_kv2, *b_o2, *b_ff1, *b_ff2;
    float *nq, *nk, *nq2, *nk2, *ln2_w, *ln2_b;
    float* table;          // scale_shift_table [6][D] fp32
    bf16_t* ln2_mod;       // (bias, weight - 1) rows for ln_mod, built at weights_ready time
};

}  // namespace

struct mi355_wan {
    mi355_wan_cfg cfg;
    int D, F, L, H, KP, NO;
    char* arena16 = nullptr;
    char* arena32 = nullptr;
    size_t used16 = 0, used32 = 0;
    bf16_t *w_patch, *w_t1, *w_t2, *w_tp, *w_x1, *w_x2, *w_proj;
    float *b_patch, *b_t1, *b_t2, *b_tp, *b_x1, *b_x2, *b_proj, *table_out;
    std::vector<WanBlockW> blk;
    std::map<std::string, WSlot> slots;
    std::vector<std::string> names;
    bool derived_dirty = true;   // ln2_mod tables / score bounds need a rebuild after a (re)bind
    int derived_ver = 0;         // bumped by every rebuild (the bounds are baked into a captured graph)
    std::vector<float> bound_self, bound_cross;

    bf16_t* a16(int64_t n) {
        size_t bytes = ((size_t)n * 2 + 255) & ~(size_t)255;
        char* p = arena16 ? arena16 + used16 : nullptr;
        used16 += bytes;
        return (bf16_t*)p;
    }
    float* a32(int64_t n) {
        size_t bytes = ((size_t)n * 4 + 255) & ~(size_t)255;
        char* p = arena32 ? arena32 + used32 : nullptr;
        used32 += bytes;
        return (float*)p;
    }
    void reg(const std::string& name, void* dst, int dt, int64_t numel) {
        if (!arena16) return;
        slots[name] = WSlot{dst, dt, numel, false};
        names.push_back(name);
    }
    void lin(const std::string& name, bf16_t* w, float* b, int out_f, int in_f) {
        reg(name + ".weight", w, DT_BF16, (int64_t)out_f * in_f);
        reg(name + ".bias", b, DT_F32, out_f);
    }
    void layout();
};

void mi355_wan::layout() {
    used16 = used32 = 0;
    slots.clear(); names.clear();
    blk.assign(L, WanBlockW());
    const int T = cfg.freq_dim, J = cfg.text_dim;
    const int64_t DD = (int64_t)D * D;
    w_patch = a16((int64_t)D * KP); b_patch = a32(D);
    reg("patch_embedding.weight", w_patch, DT_BF16, (int64_t)D * KP);
    reg("patch_embedding.bias", b_patch, DT_F32, D);
    w_t1 = a16((int64_t)D * T); b_t1 = a32(D); lin("condition_embedder.time_embedder.linear_1", w_t1, b_t1, D, T);
    w_t2 = a16(DD); b_t2 = a32(D); lin("condition_embedder.time_embedder.linear_2", w_t2, b_t2, D, D);
    w_tp = a16(6 * DD); b_tp = a32(6 * D); lin("condition_embedder.time_proj", w_tp, b_tp, 6 * D, D);
    w_x1 = a16((int64_t)D * J); b_x1 = a32(D); lin("condition_embedder.text_embedder.linear_1", w_x1, b_x1, D, J);
    w_x2 = a16(DD); b_x2 = a32(D); lin("condition_embedder.text_embedder.linear_2", w_x2, b_x2, D, D);
    for (int i = 0; i < L; ++i) {
        WanBlockW& b = blk[i];
        const std::string pre = "blocks." + std::to_string(i);
        b.table = a32(6 * D); reg(pre + ".scale_shift_table", b.table, DT_F32, 6 * D);
        b.w_qk = a16(2 * DD); b.b_qk = a32(2 * D);
        lin(pre + ".attn1.to_q", b.w_qk, b.b_qk, D, D); lin(pre + ".attn1.to_k", b.w_qk + DD, b.b_qk + D, D, D);
        b.w_v = a16(DD); b.b_v = a32(D); lin(pre + ".attn1.to_v", b.w_v, b.b_v, D, D);
        b.w_o = a16(DD); b.b_o = a32(D); lin(pre + ".attn1.to_out.0", b.w_o, b.b_o, D, D);
        b.nq = a32(D); b.nk = a32(D);
        reg(pre + ".attn1.norm_q.weight", b.nq, DT_F32, D); reg(pre + ".attn1.norm_k.weight", b.nk, DT_F32, D);
        b.w_q2 = a16(DD); b.b_q2 = a32(D); lin(pre + ".attn2.to_q", b.w_q2, b.b_q2, D, D);
        b.w_kv2 = a16(2 * DD); b.b_kv2 = a32(2 * D);
        lin(pre + ".attn2.to_k", b.w_kv2, b.b_kv2, D, D); lin(pre + ".attn2.to_v", b.w_kv2 + DD, b.b_kv2 + D, D, D);
        b.w_o2 = a16(DD); b.b_o2 = a32(D); lin(pre + ".attn2.to_out.0", b.w_o2, b.b_o2, D, D);
        b.nq2 = a32(D); b.nk2 = a32(D);
        reg(pre + ".attn2.norm_q.weight", b.nq2, DT_F32, D); reg(pre + ".attn2.norm_k.weight", b.nk2, DT_F32, D);
        b.ln2_w = a32(D); b.ln2_b = a32(D);
        reg(pre + ".norm2.weight", b.ln2_w, DT_F32, D); reg(pre + ".norm2.bias", b.ln2_b, DT_F32, D);
        b.ln2_mod = a16(2 * D);
        b.w_ff1 = a16((int64_t)F * D); b.b_ff1 = a32(F); lin(pre + ".ffn.net.0.proj", b.w_ff1, b.b_ff1, F, D);
        b.w_ff2 = a16((int64_t)D * F); b.b_ff2 = a32(D); lin(pre + ".ffn.net.2", b.w_ff2, b.b_ff2, D, F);
    }
    table_out = a32(2 * D); reg("scale_shift_table", table_out, DT_F32, 2 * D);
    w_proj = a16((int64_t)NO * D); b_proj = a32(NO); lin("proj_out", w_proj, b_proj, NO, D);
}

// training-mode state (wan_train.inc, included at the end of this file)
struct mi355_wan_plan;
static void wan_train_release(mi355_wan_plan* p);
static void wan_train_release_engine(mi355_wan* e);
static void wan_train_mark_dirty(mi355_wan* e);

extern "C" int mi355_wan_create(const mi355_wan_cfg* cfg, mi355_wan** out) {
    if (!cfg || !out) return errorf("mi355_wan_create: null argument");
    if (cfg->head_dim != 128) return errorf("mi355_wan_create: head_dim must be 128 (got %d)", cfg->head_dim);
    if (cfg->num_heads < 1 || cfg->num_heads > 48) return errorf("mi355_wan_create: num_heads must be 1..48");
    if (cfg->num_layers < 1 || cfg->num_layers > 256) return errorf("mi355_wan_create: num_layers out of range");
    if (cfg->patch_t != 1 || cfg->patch_h != 2 || cfg->patch_w != 2) return errorf("mi355_wan_create: patch_size must be (1, 2, 2)");
    const int KP = cfg->in_channels * 4;
    if (KP % 64 || cfg->text_dim % 64 || cfg->freq_dim % 64 || cfg->ffn_dim % 64)
        return errorf("mi355_wan_create: every GEMM K dim must be a multiple of 64");
    mi355_wan* e = new mi355_wan();
    e->cfg = *cfg;
    e->H = cfg->num_heads; e->D = cfg->num_heads * 128; e->F = cfg->ffn_dim; e->L = cfg->num_layers; e->KP = KP;
    e->NO = cfg->out_channels * 4;
    e->layout();
    const size_t cap16 = e->used16, cap32 = e->used32;
    hipError_t e1 = hipMalloc((void**)&e->arena16, cap16);
    hipError_t e2 = hipMalloc((void**)&e->arena32, cap32);
    if (e1 != hipSuccess || e2 != hipSuccess) {
        int r = errorf("mi355_wan_create: hipMalloc of %zu + %zu bytes failed", cap16, cap32);
        if (e->arena16) (void)hipFree(e->arena16);
        if (e->arena32) (void)hipFree(e->arena32);
        delete e;
        return r;
    }
    e->layout();
    *out = e;
    return 0;
}

extern "C" int mi355_wan_destroy(mi355_wan* e) {
    if (!e) return 0;
    wan_train_release_engine(e);
    if (e->arena16) (void)hipFree(e->arena16);
    if (e->arena32) (void)hipFree(e->arena32);
    delete e;
    return 0;
}
extern "C" int mi355_wan_num_params(mi355_wan* e) { return e ? (int)e->names.size() : 0; }
extern "C" const char* mi355_wan_param_name(mi355_wan* e, int i) {
    if (!e || i < 0 || i >= (int)e->names.size()) return nullptr;
    return e->names[i].c_str();
}
extern "C" int mi355_wan_bind_weight(mi355_wan* e, const char* name, const void* src, int dtype, int ndim, const int64_t* shape,
                                     void* stream) {
    if (!e || !name || !src) return errorf("mi355_wan_bind_weight: null argument");
    auto it = e->slots.find(name);
    if (it == e->slots.end()) return errorf("mi355_wan_bind_weight: unknown parameter '%s'", name);
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    if (n != it->second.numel)
        return errorf("mi355_wan_bind_weight: '%s' has %lld elements, expected %lld", name, (long long)n, (long long)it->second.numel);
    if (dtype < 0 || dtype > 2) return errorf("mi355_wan_bind_weight: bad dtype %d", dtype);
    HIPCHK(launch_convert(src, dtype, it->second.dst, it->second.dst_dt, n, (hipStream_t)stream));
    it->second.bound = true;
    e->derived_dirty = true;
    wan_train_mark_dirty(e);           // the transposed copies the backward's dgrad GEMMs read are stale
    return 0;
}
extern "C" int mi355_wan_weights_ready(mi355_wan* e) {
    if (!e) return errorf("null engine");
    for (auto& kv : e->slots)
        if (!kv.second.bound) return errorf("parameter '%s' has not been bound", kv.first.c_str());
    return 0;
}

// -------------------------------------------------------------------------------------- plan
struct mi355_wan_plan {
    mi355_wan* e;
    int B, ncfg, Bp, T, h, w, hp, wp, S, S_pad, Nt, Nt_pad, M, Mc, max_steps;
    int64_t n_lat;
    char* ws = nullptr;
    size_t ws_bytes = 0;
    bf16_t *patches, *x, *xn, *qkbuf, *q, *k, *vT, *o, *hid, *v, *ctx, *c1, *kvbuf, *kx, *vTx;
    bf16_t *tproj_in, *h1, *temb, *semb, *tp6, *mod_all;
    float2* cs;
    float *t_dev, *scal;
    char *io_init, *io_traj;
    float *io_noise, *io_lp;
    bf16_t *io_pe, *io_ne;
    unsigned* max2;            // [2][Bp*H]: largest squared stored row norm of q / k per (batch, head) of the current self-attention (float bits)
    float* max2_part;          // scratch of the kernel that measures them
    std::vector<float> host_t, host_sc;
    int mod_cols;
};

extern "C" int mi355_wan_plan_create(mi355_wan* e, int batch, int n_cfg, int latent_t, int latent_h, int latent_w, int n_text,
                                     int max_steps, mi355_wan_plan** out) {
    if (!e || !out) return errorf("mi355_wan_plan_create: null argument");
    if (batch < 1 || (n_cfg != 1 && n_cfg != 2) || latent_t < 1 || latent_h < 2 || latent_w < 2 || ((latent_h | latent_w) & 1) ||
        n_text < 1 || max_steps < 1)
        return errorf("mi355_wan_plan_create: bad shape (latent height / width must be even)");
    mi355_wan_plan* p = new mi355_wan_plan();
    p->e = e; p->B = batch; p->ncfg = n_cfg; p->Bp = batch * n_cfg; p->T = latent_t; p->h = latent_h; p->w = latent_w;
    p->hp = latent_h / 2; p->wp = latent_w / 2; p->S = latent_t * p->hp * p->wp; p->S_pad = (p->S + 63) / 64 * 64;
    p->Nt = n_text; p->Nt_pad = (n_text + 63) / 64 * 64; p->M = p->Bp * p->S; p->Mc = p->Bp * n_text; p->max_steps = max_steps;
    p->n_lat = (int64_t)e->cfg.in_channels * latent_t * latent_h * latent_w;
    if ((int64_t)p->Bp * p->S > 0x7fffffffLL / 2) { delete p; return errorf("mi355_wan_plan_create: too many tokens"); }
    const int D = e->D, F = e->F;
    p->mod_cols = e->L * 6 * D + 2 * D;
    const int64_t rows_cond = (int64_t)max_steps * p->Bp;
    size_t off = 0;
    auto take = [&](int64_t elems, int esz) {
        size_t o = off;
        off += (((size_t)elems * esz) + 255) & ~(size_t)255;
        return o;
    };
    const int64_t qk_el = (int64_t)p->Bp * e->H * p->S_pad * 128;
    const int64_t kx_el = (int64_t)p->Bp * e->H * p->Nt_pad * 128;
    const int64_t nl = (int64_t)batch * p->n_lat;
    size_t o_pat = take((int64_t)p->M * e->KP, 2), o_x = take((int64_t)p->M * D, 2), o_xn = take((int64_t)p->M * D, 2);
    size_t o_qkb = take((int64_t)p->M * 2 * D, 2), o_q = take(qk_el, 2), o_k = take(qk_el, 2), o_vT = take(qk_el, 2);
    size_t o_o = take((int64_t)p->M * D, 2), o_hid = take((int64_t)p->M * F, 2), o_v = take((int64_t)p->Bp * p->n_lat, 2);
    size_t o_ctx = take((int64_t)p->Mc * D, 2), o_c1 = take((int64_t)p->Mc * D, 2), o_kvb = take((int64_t)p->Mc * 2 * D, 2);
    size_t o_kx = take(kx_el * e->L, 2), o_vTx = take(kx_el * e->L, 2);
    size_t o_tpi = take(rows_cond * e->cfg.freq_dim, 2), o_h1 = take(rows_cond * D, 2), o_temb = take(rows_cond * D, 2);
    size_t o_semb = take(rows_cond * D, 2), o_tp6 = take(rows_cond * 6 * D, 2), o_mod = take(rows_cond * p->mod_cols, 2);
    size_t o_cs = take((int64_t)p->S * 64, 8), o_t = take(rows_cond, 4), o_sc = take(3 * (int64_t)max_steps, 4);
    size_t o_ii = take(nl, 4), o_it = take((int64_t)(max_steps + 1) * nl, 4), o_in = take((int64_t)max_steps * nl, 4);
    size_t o_il = take((int64_t)max_steps * batch, 4);
    size_t o_ipe = take((int64_t)batch * n_text * e->cfg.text_dim, 2), o_ine = take((int64_t)batch * n_text * e->cfg.text_dim, 2);
    size_t o_max2 = take((int64_t)2 * p->Bp * e->H, 4), o_m2p = take((int64_t)p->Bp * norm_rope_parts(p->S) * e->H, 4);
    p->ws_bytes = off;
    if (hipMalloc((void**)&p->ws, off) != hipSuccess) {
        int r = errorf("mi355_wan_plan_create: hipMalloc of %zu bytes failed", off);
        delete p;
        return r;
    }
    if (hipMemset(p->ws, 0, off) != hipSuccess) {   // padded key rows / columns of q, k, vT, kx, vTx must stay finite
        (void)hipFree(p->ws);
        delete p;
        return errorf("mi355_wan_plan_create: hipMemset failed");
    }
    char* w = p->ws;
    p->patches = (bf16_t*)(w + o_pat); p->x = (bf16_t*)(w + o_x); p->xn = (bf16_t*)(w + o_xn); p->qkbuf = (bf16_t*)(w + o_qkb);
    p->q = (bf16_t*)(w + o_q); p->k = (bf16_t*)(w + o_k); p->vT = (bf16_t*)(w + o_vT); p->o = (bf16_t*)(w + o_o);
    p->hid = (bf16_t*)(w + o_hid); p->v = (bf16_t*)(w + o_v); p->ctx = (bf16_t*)(w + o_ctx); p->c1 = (bf16_t*)(w + o_c1);
    p->kvbuf = (bf16_t*)(w + o_kvb); p->kx = (bf16_t*)(w + o_kx); p->vTx = (bf16_t*)(w + o_vTx);
    p->tproj_in = (bf16_t*)(w + o_tpi); p->h1 = (bf16_t*)(w + o_h1); p->temb = (bf16_t*)(w + o_temb); p->semb = (bf16_t*)(w + o_semb);
    p->tp6 = (bf16_t*)(w + o_tp6); p->mod_all = (bf16_t*)(w + o_mod);
    p->cs = (float2*)(w + o_cs); p->t_dev = (float*)(w + o_t); p->scal = (float*)(w + o_sc);
    p->io_init = w + o_ii; p->io_traj = w + o_it; p->io_noise = (float*)(w + o_in); p->io_lp = (float*)(w + o_il);
    p->io_pe = (bf16_t*)(w + o_ipe); p->io_ne = (bf16_t*)(w + o_ine);
    p->max2 = (unsigned*)(w + o_max2); p->max2_part = (float*)(w + o_m2p);
    // rotary table (WanRotaryPosEmbed): head_dim 128 -> t / h / w axes of 44 / 42 / 42 features, float64 angles, adjacent pairs
    {
        const int hw = 2 * (128 / 6), ax[3] = {128 - 2 * hw, hw, hw};
        std::vector<float> cs((size_t)p->S * 128);
        for (int s = 0; s < p->S; ++s) {
            const int t = s / (p->hp * p->wp), r = s % (p->hp * p->wp);
            const double pos[3] = {(double)t, (double)(r / p->wp), (double)(r % p->wp)};
            int pair = 0;
            for (int a = 0; a < 3; ++a)
                for (int j = 0; j < ax[a] / 2; ++j, ++pair) {
                    const double ang = pos[a] / pow(10000.0, (2.0 * j) / ax[a]);
                    cs[((size_t)s * 64 + pair) * 2 + 0] = (float)cos(ang);
                    cs[((size_t)s * 64 + pair) * 2 + 1] = (float)sin(ang);
                }
        }
        if (hipMemcpy(p->cs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(p->ws);
            delete p;
            return errorf("mi355_wan_plan_create: rotary table upload failed");
        }
    }
    *out = p;
    return 0;
}

extern "C" int mi355_wan_plan_destroy(mi355_wan_plan* p) {
    if (!p) return 0;
    wan_train_release(p);
    if (p->ws) (void)hipFree(p->ws);
    delete p;
    return 0;
}
extern "C" int64_t mi355_wan_plan_workspace_bytes(mi355_wan_plan* p) { return p ? (int64_t)p->ws_bytes : 0; }

// ---------------------------------------------------------------------------------- forward
namespace {

constexpr float kScale = 0.08838834764831845f * 1.4426950408889634f;   // log2(e) / sqrt(128)

// key 24: self-attention score bound from the data where the weights prove none (1 = default).  Wan's q / k RMSNorm runs ACROSS heads, so the
// weight-side bound is 200 * max|w_q| * max|w_k| -- never <= 60 -- and every self-attention ran the running-max kernel.  The kernel that
// stores q and k now also measures them (largest squared row norm per (batch, head): per-wave partial maxima + a tiny reduction); |q . k| <= |q| |k| then
// bounds the scores of each (b, h) from what is actually there, and the static-softmax 4-wave kernel takes every (b, h) that passes.
int g_wan_data_bound = 1;

// after a (re)bind: LayerNorm-affine rows for ln_mod, and the |score| bounds from the across-head norm weights:
// ||q_hat|| <= sqrt(H*128) * max|w| for the whole row, so per head |q_h . k_h| <= ||q_hat|| ||k_hat|| (Cauchy-Schwarz on the sub-vectors)
int refresh_derived(mi355_wan* e, hipStream_t st) {
    if (!e->derived_dirty) return 0;
    for (auto& b : e->blk) HIPCHK(launch_affine_to_mod(b.ln2_w, b.ln2_b, b.ln2_mod, e->D, st));
    std::vector<float> host(e->used32 / 4);
    HIPCHK(hipMemcpyAsync(host.data(), e->arena32, e->used32, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    auto amax = [&](const float* dev) {
        const float* hh = host.data() + (dev - (const float*)e->arena32);
        float m = 0.f;
        for (int i = 0; i < e->D; ++i) m = fmaxf(m, fabsf(hh[i]));
        return m;
    };
    e->bound_self.assign(e->L, 0.f); e->bound_cross.assign(e->L, 0.f);
    const float c = (float)e->D * kScale * 1.02f;
    for (int i = 0; i < e->L; ++i) {
        e->bound_self[i] = c * amax(e->blk[i].nq) * amax(e->blk[i].nk);
        e->bound_cross[i] = c * amax(e->blk[i].nq2) * amax(e->blk[i].nk2);
    }
    e->derived_dirty = false;
    ++e->derived_ver;
    return 0;
}

int ln_mod(mi355_wan_plan* p, hipStream_t st, const bf16_t* x, bf16_t* out, const bf16_t* mod, long mod_ld, int M, int rps, int shift_off,
           int scale_off) {
    LnModParams l;
    memset(&l, 0, sizeof(l));
    l.x = x; l.out = out; l.mod = mod; l.mod_ld = mod_ld; l.shift_off = shift_off; l.scale_off = scale_off;
    l.M = M; l.D = p->e->D; l.rows_per_sample = rps; l.eps = p->e->cfg.eps;
    HIPCHK(launch_ln_mod(l, st));
    return 0;
}

int norm_rope(mi355_wan_plan* p, hipStream_t st, const bf16_t* src, long ld, int col, const float* w, bool rope, bf16_t* out, int M, int rps,
              int S_pad, float scale, unsigned* max2 = nullptr, float* rstd = nullptr) {
    NormRopeFullParams r;
    memset(&r, 0, sizeof(r));
    r.src = src; r.src_ld = ld; r.col = col; r.weight = w; r.cs = rope ? p->cs : nullptr; r.out = out; r.M = M; r.H = p->e->H;
    r.rows_per_sample = rps; r.s_off = 0; r.S_pad = S_pad; r.eps = p->e->cfg.eps; r.out_scale = scale; r.max2 = max2; r.max2_part = p->max2_part;
    r.rstd_out = rstd;
    HIPCHK(launch_norm_rope_full(r, st));
    return 0;
}

int vt_proj(mi355_wan_plan* p, hipStream_t st, const bf16_t* w_v, const float* b_v, const bf16_t* xin, int M, int rps, bf16_t* vT, int S_pad) {
    const int D = p->e->D;
    GemmParams gv = make_gemm(w_v, D, xin, D, D, M, D, EPI_VT, b_v, nullptr, 0);
    gv.q = vT; gv.H = p->e->H; gv.S_pad = S_pad; gv.s_off = 0; gv.rows_per_sample = rps; gv.hd_shift = 7;
    HIPCHK(launch_gemm(gv, st));
    return 0;
}

int gate_res(mi355_wan_plan* p, hipStream_t st, const bf16_t* A, int K, const bf16_t* W, const float* bias, bf16_t* x, int M, int rps,
             const bf16_t* mod, int gate_off) {
    GemmParams g = make_gemm(A, K, W, K, M, p->e->D, K, EPI_GATE_RES, bias, x, p->e->D);
    g.aux = mod + gate_off; g.ld_aux = p->mod_cols; g.rows_per_sample = rps;
    HIPCHK(launch_gemm(g, st));
    return 0;
}

// step-invariant prompt work: text embedder, then every block's cross-attention K / V^T.  Prompt halves: [negative, positive] for CFG.
int prepare_prompt(mi355_wan_plan* p, hipStream_t st, const void* enc_a, const void* enc_b, float* rstd_kx = nullptr) {
    mi355_wan* e = p->e;
    const int D = e->D, J = e->cfg.text_dim;
    const void* encs[2] = {enc_a, enc_b};
    if (p->ncfg == 2 && !enc_b) return errorf("n_cfg == 2 needs both prompt halves");
    for (int half = 0; half < p->ncfg; ++half) {
        const int rows = p->B * p->Nt;
        GemmParams g1 = make_gemm((const bf16_t*)encs[half], J, e->w_x1, J, rows, D, J, EPI_BIAS_GELU, e->b_x1, p->c1 + (int64_t)half * rows * D, D);
        HIPCHK(launch_gemm(g1, st));
    }
    GemmParams g2 = make_gemm(p->c1, D, e->w_x2, D, p->Mc, D, D, EPI_BIAS, e->b_x2, p->ctx, D);
    HIPCHK(launch_gemm(g2, st));
    const int64_t kx_el = (int64_t)p->Bp * e->H * p->Nt_pad * 128;
    for (int i = 0; i < e->L; ++i) {
        const WanBlockW& b = e->blk[i];
        GemmParams gk = make_gemm(p->ctx, D, b.w_kv2, D, p->Mc, D, D, EPI_BIAS, b.b_kv2, p->kvbuf, D);     // to_k
        HIPCHK(launch_gemm(gk, st));
        CHK(norm_rope(p, st, p->kvbuf, D, 0, b.nk2, false, p->kx + i * kx_el, p->Mc, p->Nt, p->Nt_pad, 1.0f, nullptr,
                      rstd_kx ? rstd_kx + (int64_t)i * p->Mc : nullptr));      // (training mode: 1 / rms of every text row for the k-norm backward)
        CHK(vt_proj(p, st, b.w_kv2 + (int64_t)D * D, b.b_kv2 + D, p->ctx, p->Mc, p->Nt, p->vTx + i * kx_el, p->Nt_pad));
    }
    return 0;
}

// conditioning of `nsteps` steps: temb, time_proj(silu(temb)), modulation tables = scale_shift_table + time_proj
int prepare_conditioning(mi355_wan_plan* p, hipStream_t st, int nsteps) {
    mi355_wan* e = p->e;
    const int D = e->D, T = e->cfg.freq_dim;
    const int rows = nsteps * p->Bp;
    HIPCHK(launch_time_proj(p->t_dev, rows, T, DT_F32, p->tproj_in, st));
    GemmParams g1 = make_gemm(p->tproj_in, T, e->w_t1, T, rows, D, T, EPI_BIAS_SILU, e->b_t1, p->h1, D);
    HIPCHK(launch_gemm(g1, st));
    GemmParams g2 = make_gemm(p->h1, D, e->w_t2, D, rows, D, D, EPI_BIAS, e->b_t2, p->temb, D);
    HIPCHK(launch_gemm(g2, st));
    GemmParams g3 = make_gemm(p->h1, D, e->w_t2, D, rows, D, D, EPI_BIAS_SILU, e->b_t2, p->semb, D);       // silu(temb) for time_proj
    HIPCHK(launch_gemm(g3, st));
    GemmParams g4 = make_gemm(p->semb, D, e->w_tp, D, rows, 6 * D, D, EPI_BIAS, e->b_tp, p->tp6, 6 * D);
    HIPCHK(launch_gemm(g4, st));
    for (int i = 0; i < e->L; ++i)
        HIPCHK(launch_bcast_add(p->tp6, e->blk[i].table, p->mod_all + (int64_t)i * 6 * D, p->mod_cols, rows, 6 * D, 1, st));
    HIPCHK(launch_bcast_add(p->temb, e->table_out, p->mod_all + (int64_t)e->L * 6 * D, p->mod_cols, rows, D, 2, st));
    return 0;
}

// one transformer forward over the forward batch Bp (latents replicated n_cfg times): v_out [Bp][16][T][h][w] bf16.
// Training mode (`tb` = the plan's per-block stash, wan_train.inc): the SAME launches with the activations the backward needs written to
// per-block buffers (+ the residual stream copied aside at the three sub-layer boundaries, the log-sum-exp / 1/rms side outputs switched on,
// the FFN pre-activation stashed by the GEMM epilogue); `kx` / `vTx` = the cross-attention keys / values to read (the training state's copy).
int forward_core(mi355_wan_plan* p, hipStream_t st, const void* latents, int lat_dt, const bf16_t* mod, bf16_t* v_out, const WTrainBlk* tb = nullptr,
                 bf16_t* x_final = nullptr, const bf16_t* kx = nullptr, const bf16_t* vTx = nullptr) {
    mi355_wan* e = p->e;
    const int D = e->D, F = e->F, M = p->M, S = p->S;
    const int64_t kx_el = (int64_t)p->Bp * e->H * p->Nt_pad * 128;
    const size_t x_b = (size_t)M * D * 2;
    if (!kx) { kx = p->kx; vTx = p->vTx; }
    HIPCHK(launch_patchify(latents, lat_dt, p->patches, p->B, p->ncfg, e->cfg.in_channels, p->T * p->h, p->w, 2, st));
    GemmParams g0 = make_gemm(p->patches, e->KP, e->w_patch, e->KP, M, D, e->KP, EPI_BIAS, e->b_patch, p->x, D);
    HIPCHK(launch_gemm(g0, st));
    for (int i = 0; i < e->L; ++i) {
        const WanBlockW& b = e->blk[i];
        const WTrainBlk* k = tb ? tb + i : nullptr;
        const int m0 = i * 6 * D;        // chunks: shift, scale, gate, c_shift, c_scale, c_gate
        // ---- self-attention
        bf16_t* xn1 = k ? k->xn1 : p->xn;
        if (k) HIPCHK(copy_d2d(k->x_in, p->x, x_b, st));
        CHK(ln_mod(p, st, p->x, xn1, mod, p->mod_cols, M, S, m0, m0 + D));
        GemmParams gq = make_gemm(xn1, D, b.w_qk, D, M, 2 * D, D, EPI_BIAS, b.b_qk, p->qkbuf, 2 * D);
        HIPCHK(launch_gemm(gq, st));
        // self-attention score bound from the data (the weights prove none: RMSNorm across heads): largest stored row norm per (b, h)
        const bool dyn_bound = g_wan_data_bound && !(e->bound_self[i] > 0.f && e->bound_self[i] <= 60.f);
        bf16_t* q1 = k ? k->q : p->q;
        bf16_t* k1 = k ? k->k : p->k;
        bf16_t* v1 = k ? k->vT : p->vT;
        CHK(norm_rope(p, st, p->qkbuf, 2 * D, 0, b.nq, true, q1, M, S, p->S_pad, kScale, dyn_bound ? p->max2 : nullptr, k ? k->rstd_q : nullptr));
        CHK(norm_rope(p, st, p->qkbuf, 2 * D, D, b.nk, true, k1, M, S, p->S_pad, 1.0f, dyn_bound ? p->max2 + p->Bp * e->H : nullptr,
                      k ? k->rstd_k : nullptr));
        CHK(vt_proj(p, st, b.w_v, b.b_v, xn1, M, S, v1, p->S_pad));
        bf16_t* o1 = k ? k->o1 : p->o;
        {
            Attn128Params a;
            memset(&a, 0, sizeof(a));
            a.q = q1; a.k = k1; a.vT = v1; a.o_first = o1; a.ld_first = D; a.n_first = S; a.o_rest = o1; a.ld_rest = D;
            a.B = p->Bp; a.H = e->H; a.S = S; a.S_pad = p->S_pad; a.q_prescaled = 1; a.score_bound = e->bound_self[i];
            if (dyn_bound) { a.qmax2 = p->max2; a.kmax2 = p->max2 + p->Bp * e->H; }
            a.lse = k ? k->lse1 : nullptr;
            HIPCHK(launch_attention128(a, st));
        }
        CHK(gate_res(p, st, o1, D, b.w_o, b.b_o, p->x, M, S, mod, m0 + 2 * D));
        // ---- cross-attention to the cached text keys / values
        bf16_t* xn2 = k ? k->xn2 : p->xn;
        if (k) HIPCHK(copy_d2d(k->x_mid1, p->x, x_b, st));
        CHK(ln_mod(p, st, p->x, xn2, b.ln2_mod, 0, M, S, 0, D));
        GemmParams gq2 = make_gemm(xn2, D, b.w_q2, D, M, D, D, EPI_BIAS, b.b_q2, p->qkbuf, D);
        HIPCHK(launch_gemm(gq2, st));
        bf16_t* q2 = k ? k->q2 : p->q;
        CHK(norm_rope(p, st, p->qkbuf, D, 0, b.nq2, false, q2, M, S, p->S_pad, kScale, nullptr, k ? k->rstd_q2 : nullptr));
        bf16_t* o2 = k ? k->o2 : p->o;
        {
            Attn128Params a;
            memset(&a, 0, sizeof(a));
            a.q = q2; a.k = kx + i * kx_el; a.vT = vTx + i * kx_el; a.o_first = o2; a.ld_first = D; a.n_first = S;
            a.o_rest = o2; a.ld_rest = D; a.B = p->Bp; a.H = e->H; a.S = S; a.S_pad = p->S_pad; a.q_prescaled = 1;
            a.S_kv = p->Nt; a.S_kv_pad = p->Nt_pad; a.score_bound = e->bound_cross[i];
            a.lse = k ? k->lse2 : nullptr;
            HIPCHK(launch_attention128(a, st));
        }
        GemmParams go2 = make_gemm(o2, D, b.w_o2, D, M, D, D, EPI_POSADD, b.b_o2, p->x, D);
        go2.aux = p->x; go2.ld_aux = D; go2.rows_per_sample = M;
        HIPCHK(launch_gemm(go2, st));
        // ---- feed-forward
        bf16_t* xn3 = k ? k->xn3 : p->xn;
        if (k) HIPCHK(copy_d2d(k->x_mid2, p->x, x_b, st));
        CHK(ln_mod(p, st, p->x, xn3, mod, p->mod_cols, M, S, m0 + 3 * D, m0 + 4 * D));
        GemmParams f1 = make_gemm(xn3, D, b.w_ff1, D, M, F, D, EPI_BIAS_GELU, b.b_ff1, p->hid, F);
        if (k) { f1.stash = k->pre; f1.ld_stash = F; }
        HIPCHK(launch_gemm(f1, st));
        CHK(gate_res(p, st, p->hid, F, b.w_ff2, b.b_ff2, p->x, M, S, mod, m0 + 5 * D));
    }
    if (x_final) HIPCHK(copy_d2d(x_final, p->x, x_b, st));
    const int mo = e->L * 6 * D;         // output modulation: shift, scale
    CHK(ln_mod(p, st, p->x, p->xn, mod, p->mod_cols, M, S, mo, mo + D));
    GemmParams g = make_gemm(p->xn, D, e->w_proj, D, M, e->NO, D, EPI_UNPATCH, e->b_proj, v_out, 0);
    g.hp = p->T * p->hp; g.wp = p->wp; g.patch = 2; g.out_ch = e->cfg.out_channels;
    HIPCHK(launch_gemm(g, st));
    return 0;
}

}  // namespace

namespace mi355 { void set_wan_data_bound(int v) { g_wan_data_bound = v != 0; } }

// transformer only (tests / replay): t[Bp] device fp32 = the timestep values the network embeds; enc_b == NULL when n_cfg == 1.
// With n_cfg == 2 the forward batch is [enc_a (negative), enc_b (positive)] on replicated latents, v_out holds both halves.
extern "C" int mi355_wan_forward(mi355_wan_plan* p, void* stream, const void* latents, int lat_dtype, const float* t, const void* enc_a,
                                 const void* enc_b, void* v_out) {
    if (!p || !latents || !t || !enc_a || !v_out) return errorf("mi355_wan_forward: null argument");
    if (lat_dtype < 0 || lat_dtype > 2) return errorf("mi355_wan_forward: bad latent dtype %d", lat_dtype);
    CHK(mi355_wan_weights_ready(p->e));
    hipStream_t st = (hipStream_t)stream;
    CHK(refresh_derived(p->e, st));
    HIPCHK(hipMemcpyAsync(p->t_dev, t, (size_t)p->Bp * 4, hipMemcpyDeviceToDevice, st));
    CHK(prepare_prompt(p, st, enc_a, enc_b));
    CHK(prepare_conditioning(p, st, 1));
    return forward_core(p, st, latents, lat_dtype, p->mod_all, (bf16_t*)v_out);
}

// The loop runs as eager launches: replaying it as ONE hipGraph (the SD3.5 / FLUX.1 / Qwen-Image engines do) was measured in round 3 at the
// reference's example shape (240 x 240 x 5 frames, B = 1, CFG) and at 480 x 832 x 17 -- bit-identical, -0.1 % / -0.3 %
// (profiles/r03a_wan_graph_ab.txt): even the smallest Wan clip is not launch-bound (per-step cross-attention K / V are cached, 30 blocks of
// ~10 launches each).  Removed rather than kept as a dead option.
// the whole N-step loop; timesteps_host: the scheduler's (integer-valued) timesteps; sigma of a step = t / 1000
// (scheduler/unipc_multistep.py:288-291); neg_embeds == NULL <=> the plan has n_cfg == 1.
extern "C" int mi355_wan_rollout(mi355_wan_plan* p, void* stream, int n_steps, const float* timesteps_host, const float* sigmas_host,
                                 const float* noise_levels_host, int dynamics, float guidance, const void* init_latents, int init_dtype,
                                 int storage_dtype, const float* step_noise, const void* prompt_embeds, const void* neg_embeds,
                                 const int32_t* keep_slot_host, void* out_latents, float* out_log_probs, void* out_final,
                                 int compute_log_prob) {
    if (!p || !timesteps_host || !sigmas_host || !noise_levels_host || !init_latents || !prompt_embeds)
        return errorf("mi355_wan_rollout: null argument");
    if (n_steps < 1 || n_steps > p->max_steps) return errorf("mi355_wan_rollout: n_steps %d exceeds the plan's max_steps %d", n_steps, p->max_steps);
    if (storage_dtype < 0 || storage_dtype > 2 || init_dtype < 0 || init_dtype > 2) return errorf("mi355_wan_rollout: bad dtype");
    if (p->ncfg == 2 && !neg_embeds) return errorf("mi355_wan_rollout: plan has n_cfg == 2 but no negative prompt embeddings");
    if (!step_noise && dynamics != MI355_ODE) return errorf("mi355_wan_rollout: step_noise is NULL");
    if (dynamics < 0 || dynamics > 3) return errorf("mi355_wan_rollout: unknown dynamics %d", dynamics);
    CHK(mi355_wan_weights_ready(p->e));
    hipStream_t st = (hipStream_t)stream;
    CHK(refresh_derived(p->e, st));
    const int B = p->B, Bp = p->Bp;
    std::vector<float>& tt = p->host_t;
    std::vector<float>& sc = p->host_sc;
    tt.assign((size_t)n_steps * Bp, 0.f);
    sc.assign(3 * (size_t)p->max_steps, 0.f);
    for (int i = 0; i < n_steps; ++i) {
        for (int j = 0; j < Bp; ++j) tt[(size_t)i * Bp + j] = timesteps_host[i];     // timestep = t.expand(B) (wan2_t2v.py:499)
        const float t_next = (i + 1 < n_steps) ? timesteps_host[i + 1] : 0.0f;
        sc[i] = timesteps_host[i] / 1000.0f;
        sc[p->max_steps + i] = t_next / 1000.0f;
        sc[2 * p->max_steps + i] = noise_levels_host[i];
    }
    HIPCHK(hipMemcpyAsync(p->t_dev, tt.data(), tt.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(p->scal, sc.data(), sc.size() * 4, hipMemcpyHostToDevice, st));
    const int64_t nl = (int64_t)B * p->n_lat;
    const size_t in_esz = init_dtype == MI355_F32 ? 4 : 2;
    const size_t emb_bytes = (size_t)B * p->Nt * p->e->cfg.text_dim * 2;
    HIPCHK(hipMemcpyAsync(p->io_init, init_latents, nl * in_esz, hipMemcpyDeviceToDevice, st));
    if (step_noise) HIPCHK(hipMemcpyAsync(p->io_noise, step_noise, (size_t)n_steps * nl * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(p->io_pe, prompt_embeds, emb_bytes, hipMemcpyDeviceToDevice, st));
    if (p->ncfg == 2) HIPCHK(hipMemcpyAsync(p->io_ne, neg_embeds, emb_bytes, hipMemcpyDeviceToDevice, st));
    const float sigma_max = sigmas_host[1];
    const int clp = compute_log_prob && out_log_probs;
    const size_t esz = storage_dtype == MI355_F32 ? 4 : 2;
    const size_t lat_bytes = (size_t)nl * esz;
    // everything below reads / writes plan-owned buffers at fixed addresses (staged inputs, t_dev / scal, io_traj, io_lp)
    auto body = [&](hipStream_t sx) -> int {
        if (p->ncfg == 2) CHK(prepare_prompt(p, sx, p->io_ne, p->io_pe));
        else CHK(prepare_prompt(p, sx, p->io_pe, nullptr));
        CHK(prepare_conditioning(p, sx, n_steps));
        HIPCHK(launch_convert(p->io_init, init_dtype, p->io_traj, storage_dtype, (long)nl, sx));      // cast_latents(init)
        for (int i = 0; i < n_steps; ++i) {
            const bf16_t* mod = p->mod_all + (int64_t)i * Bp * p->mod_cols;
            char* cur = p->io_traj + (size_t)i * lat_bytes;
            char* nxt = p->io_traj + (size_t)(i + 1) * lat_bytes;
            CHK(forward_core(p, sx, cur, storage_dtype, mod, p->v));
            SdeStepParams s;
            memset(&s, 0, sizeof(s));
            s.v_uncond = p->ncfg == 2 ? p->v : nullptr;
            s.v_text = p->ncfg == 2 ? p->v + nl : p->v;
            s.v_dt = DT_BF16; s.guidance = guidance; s.latents = cur; s.lat_dt = storage_dtype;
            s.noise = step_noise ? p->io_noise + (int64_t)i * nl : nullptr;
            s.sigma = p->scal + i; s.sigma_next = p->scal + p->max_steps + i; s.eta = p->scal + 2 * p->max_steps + i; s.scalar_stride = 0;
            s.sigma_max = sigma_max; s.dynamics = dynamics; s.compute_log_prob = clp ? 2 : 0; s.B = B; s.n = p->n_lat;
            s.next_out = nxt; s.next_out_dt = storage_dtype; s.log_prob = clp ? p->io_lp + (int64_t)i * B : nullptr;
            HIPCHK(launch_sde_step(s, sx));
        }
        return 0;
    };
    CHK(body(st));
    if (keep_slot_host && out_latents)
        for (int i = 0; i <= n_steps; ++i)
            if (keep_slot_host[i] >= 0)
                HIPCHK(hipMemcpyAsync((char*)out_latents + (size_t)keep_slot_host[i] * lat_bytes, p->io_traj + (size_t)i * lat_bytes,
                                      lat_bytes, hipMemcpyDeviceToDevice, st));
    if (clp)
        for (int i = 0; i < n_steps; ++i)
            if (noise_levels_host[i] > 0.f)
                HIPCHK(hipMemcpyAsync(out_log_probs + (int64_t)i * B, p->io_lp + (int64_t)i * B, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    if (out_final)
        HIPCHK(hipMemcpyAsync(out_final, p->io_traj + (size_t)n_steps * lat_bytes, lat_bytes, hipMemcpyDeviceToDevice, st));
    return 0;
}

#include "wan_train.inc"
