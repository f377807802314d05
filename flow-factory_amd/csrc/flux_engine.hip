// mi355_flow -- FLUX.1 rollout engine behind the C ABI (include/mi355_flow.h, mi355_flux_*): SURVEY.md 8(f) row N3.
// Replaces `self.transformer(...)` + `self.scheduler.step(...)` inside the denoising loop of Flux1Adapter.inference /
// .forward (reference src/flow_factory/models/flux/flux1.py:151-346): packed latents (B, Ni, 64), timestep t/1000,
// embedded guidance (no CFG), txt_ids = 0, img_ids = (0, row, col), then the same FlowMatchEulerDiscreteSDEScheduler.step.
//
// Same design as the SD3.5 engine (engine.hip): all AdaLN modulation linears of all blocks concatenated and evaluated for
// all N steps in ONE GEMM before the loop (the conditioning depends on (t, guidance, pooled) only), the context embedder
// hoisted, q/k/v projections with the V^T scatter fused, gated residuals in place, one fused scheduler-step kernel.
// FLUX specifics: head_dim 128 (attention128.hip), per-head RMSNorm + RoPE in one pass (flux_ops.hip), context tokens
// first in the joint sequence, 38 single-stream blocks whose attention output and GELU(MLP) halves are written side by
// side into one [M][5D] buffer so that proj_out is a single K = 5D GEMM with the gated residual fused.
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

struct FSlot { void* dst; int dst_dt; int64_t numel; bool bound; };

struct DoubleW {
    bf16_t *w_qk, *w_v, *w_o, *w_cqk, *w_cv, *w_co, *w_ff1, *w_ff2, *w_cff1, *w_cff2;
    float *b_qk, *b_v, *b_o, *b_cqk, *b_cv, *b_co, *b_ff1, *b_ff2, *b_cff1, *b_cff2;
    float *nq, *nk, *ncq, *nck;
    int mod_img, mod_ctx;
    float bound = 0.f;     // proven |score| bound of the block's attention (update_score_bounds)
};
struct SingleW {
    bf16_t *w_qk, *w_v, *w_mlp, *w_out;
    float *b_qk, *b_v, *b_mlp, *b_out;
    float *nq, *nk;
    int mod;
    float bound = 0.f;
};

// value-round an fp32 to a storage dtype on the host (bf16 / fp16 round-to-nearest-even)
float host_round(float v, int dt) {
    if (dt == DT_F32) return v;
    if (dt == DT_BF16) {
        unsigned u;
        memcpy(&u, &v, 4);
        u += 0x7fffu + ((u >> 16) & 1u);
        u &= 0xffff0000u;
        memcpy(&v, &u, 4);
        return v;
    }
    return (float)(_Float16)v;
}

}  // namespace

struct mi355_flux {
    mi355_flux_cfg cfg;
    int D, F, L, LS, H;
    int mod_cols, mod_out;
    char* arena16 = nullptr;
    char* arena32 = nullptr;
    size_t used16 = 0, used32 = 0;
    bf16_t *w_x, *w_ctx, *w_t1, *w_t2, *w_g1, *w_g2, *w_p1, *w_p2, *w_mod, *w_proj;
    float *b_x, *b_ctx, *b_t1, *b_t2, *b_g1, *b_g2, *b_p1, *b_p2, *b_mod, *b_proj;
    std::vector<DoubleW> dbl;
    std::vector<SingleW> sgl;
    std::map<std::string, FSlot> slots;
    std::vector<std::string> names;
    bool bounds_dirty = true;
    int bounds_ver = 0;          // bumped whenever the per-block score bounds are recomputed (they are baked into a captured graph)

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
        slots[name] = FSlot{dst, dt, numel, false};
        names.push_back(name);
    }
    void lin(const std::string& name, bf16_t* w, float* b, int out_f, int in_f) {
        reg(name + ".weight", w, DT_BF16, (int64_t)out_f * in_f);
        reg(name + ".bias", b, DT_F32, out_f);
    }
    void layout();
};

struct mi355_flux_plan;
// training-mode state (flux_train.inc, included at the end of this file)
static void flux_train_release(mi355_flux_plan* p);
static void flux_train_release_engine(mi355_flux* e);
static void flux_train_mark_dirty(mi355_flux* e);

void mi355_flux::layout() {
    used16 = used32 = 0;
    slots.clear(); names.clear();
    dbl.assign(L, DoubleW()); sgl.assign(LS, SingleW());
    const int T = cfg.time_proj_dim, J = cfg.joint_attention_dim, P = cfg.pooled_projection_dim, C = cfg.in_channels;
    const int64_t DD = (int64_t)D * D;
    w_x = a16((int64_t)D * C); b_x = a32(D); lin("x_embedder", w_x, b_x, D, C);
    w_ctx = a16((int64_t)D * J); b_ctx = a32(D); lin("context_embedder", w_ctx, b_ctx, D, J);
    w_t1 = a16((int64_t)D * T); b_t1 = a32(D); lin("time_text_embed.timestep_embedder.linear_1", w_t1, b_t1, D, T);
    w_t2 = a16(DD); b_t2 = a32(D); lin("time_text_embed.timestep_embedder.linear_2", w_t2, b_t2, D, D);
    if (cfg.guidance_embeds) {
        w_g1 = a16((int64_t)D * T); b_g1 = a32(D); lin("time_text_embed.guidance_embedder.linear_1", w_g1, b_g1, D, T);
        w_g2 = a16(DD); b_g2 = a32(D); lin("time_text_embed.guidance_embedder.linear_2", w_g2, b_g2, D, D);
    }
    w_p1 = a16((int64_t)D * P); b_p1 = a32(D); lin("time_text_embed.text_embedder.linear_1", w_p1, b_p1, D, P);
    w_p2 = a16(DD); b_p2 = a32(D); lin("time_text_embed.text_embedder.linear_2", w_p2, b_p2, D, D);
    int cols = 0;
    for (int i = 0; i < L; ++i) { dbl[i].mod_img = cols; cols += 6 * D; dbl[i].mod_ctx = cols; cols += 6 * D; }
    for (int i = 0; i < LS; ++i) { sgl[i].mod = cols; cols += 3 * D; }
    mod_out = cols; cols += 2 * D;
    mod_cols = cols;
    w_mod = a16((int64_t)mod_cols * D); b_mod = a32(mod_cols);
    for (int i = 0; i < L; ++i) {
        const std::string pre = "transformer_blocks." + std::to_string(i);
        lin(pre + ".norm1.linear", w_mod + (int64_t)dbl[i].mod_img * D, b_mod + dbl[i].mod_img, 6 * D, D);
        lin(pre + ".norm1_context.linear", w_mod + (int64_t)dbl[i].mod_ctx * D, b_mod + dbl[i].mod_ctx, 6 * D, D);
    }
    for (int i = 0; i < LS; ++i)
        lin("single_transformer_blocks." + std::to_string(i) + ".norm.linear", w_mod + (int64_t)sgl[i].mod * D, b_mod + sgl[i].mod, 3 * D, D);
    lin("norm_out.linear", w_mod + (int64_t)mod_out * D, b_mod + mod_out, 2 * D, D);
    for (int i = 0; i < L; ++i) {
        DoubleW& b = dbl[i];
        const std::string pre = "transformer_blocks." + std::to_string(i);
        b.w_qk = a16(2 * DD); b.b_qk = a32(2 * D);
        lin(pre + ".attn.to_q", b.w_qk, b.b_qk, D, D); lin(pre + ".attn.to_k", b.w_qk + DD, b.b_qk + D, D, D);
        b.w_v = a16(DD); b.b_v = a32(D); lin(pre + ".attn.to_v", b.w_v, b.b_v, D, D);
        b.w_o = a16(DD); b.b_o = a32(D); lin(pre + ".attn.to_out.0", b.w_o, b.b_o, D, D);
        b.w_cqk = a16(2 * DD); b.b_cqk = a32(2 * D);
        lin(pre + ".attn.add_q_proj", b.w_cqk, b.b_cqk, D, D); lin(pre + ".attn.add_k_proj", b.w_cqk + DD, b.b_cqk + D, D, D);
        b.w_cv = a16(DD); b.b_cv = a32(D); lin(pre + ".attn.add_v_proj", b.w_cv, b.b_cv, D, D);
        b.w_co = a16(DD); b.b_co = a32(D); lin(pre + ".attn.to_add_out", b.w_co, b.b_co, D, D);
        const int hd = cfg.head_dim;
        b.nq = a32(hd); b.nk = a32(hd); b.ncq = a32(hd); b.nck = a32(hd);
        reg(pre + ".attn.norm_q.weight", b.nq, DT_F32, hd); reg(pre + ".attn.norm_k.weight", b.nk, DT_F32, hd);
        reg(pre + ".attn.norm_added_q.weight", b.ncq, DT_F32, hd); reg(pre + ".attn.norm_added_k.weight", b.nck, DT_F32, hd);
        b.w_ff1 = a16((int64_t)F * D); b.b_ff1 = a32(F); lin(pre + ".ff.net.0.proj", b.w_ff1, b.b_ff1, F, D);
        b.w_ff2 = a16((int64_t)D * F); b.b_ff2 = a32(D); lin(pre + ".ff.net.2", b.w_ff2, b.b_ff2, D, F);
        b.w_cff1 = a16((int64_t)F * D); b.b_cff1 = a32(F); lin(pre + ".ff_context.net.0.proj", b.w_cff1, b.b_cff1, F, D);
        b.w_cff2 = a16((int64_t)D * F); b.b_cff2 = a32(D); lin(pre + ".ff_context.net.2", b.w_cff2, b.b_cff2, D, F);
    }
    for (int i = 0; i < LS; ++i) {
        SingleW& b = sgl[i];
        const std::string pre = "single_transformer_blocks." + std::to_string(i);
        b.w_qk = a16(2 * DD); b.b_qk = a32(2 * D);
        lin(pre + ".attn.to_q", b.w_qk, b.b_qk, D, D); lin(pre + ".attn.to_k", b.w_qk + DD, b.b_qk + D, D, D);
        b.w_v = a16(DD); b.b_v = a32(D); lin(pre + ".attn.to_v", b.w_v, b.b_v, D, D);
        b.w_mlp = a16((int64_t)F * D); b.b_mlp = a32(F); lin(pre + ".proj_mlp", b.w_mlp, b.b_mlp, F, D);
        b.w_out = a16((int64_t)D * (D + F)); b.b_out = a32(D); lin(pre + ".proj_out", b.w_out, b.b_out, D, D + F);
        b.nq = a32(cfg.head_dim); b.nk = a32(cfg.head_dim);
        reg(pre + ".attn.norm_q.weight", b.nq, DT_F32, cfg.head_dim); reg(pre + ".attn.norm_k.weight", b.nk, DT_F32, cfg.head_dim);
    }
    w_proj = a16((int64_t)C * D); b_proj = a32(C); lin("proj_out", w_proj, b_proj, C, D);
}

extern "C" int mi355_flux_create(const mi355_flux_cfg* cfg, mi355_flux** out) {
    if (!cfg || !out) return errorf("mi355_flux_create: null argument");
    if (cfg->head_dim != 128) return errorf("mi355_flux_create: head_dim must be 128 (got %d)", cfg->head_dim);
    if (cfg->axes_dims_rope[0] + cfg->axes_dims_rope[1] + cfg->axes_dims_rope[2] != 128 ||
        (cfg->axes_dims_rope[0] | cfg->axes_dims_rope[1] | cfg->axes_dims_rope[2]) & 1)
        return errorf("mi355_flux_create: axes_dims_rope must be even and sum to head_dim");
    if (cfg->num_layers < 0 || cfg->num_single_layers < 0 || cfg->num_layers + cfg->num_single_layers < 1 || cfg->num_layers > 256 ||
        cfg->num_single_layers > 256)
        return errorf("mi355_flux_create: layer counts out of range");
    if (cfg->in_channels % 64 || cfg->joint_attention_dim % 64 || cfg->pooled_projection_dim % 64 || cfg->time_proj_dim % 64)
        return errorf("mi355_flux_create: every GEMM K dim must be a multiple of 64");
    mi355_flux* e = new mi355_flux();
    e->cfg = *cfg;
    e->H = cfg->num_heads; e->D = cfg->num_heads * cfg->head_dim; e->F = 4 * e->D;
    e->L = cfg->num_layers; e->LS = cfg->num_single_layers;
    e->layout();
    const size_t cap16 = e->used16, cap32 = e->used32;
    hipError_t e1 = hipMalloc((void**)&e->arena16, cap16);
    hipError_t e2 = hipMalloc((void**)&e->arena32, cap32);
    if (e1 != hipSuccess || e2 != hipSuccess) {
        int r = errorf("mi355_flux_create: hipMalloc of %zu + %zu bytes failed", cap16, cap32);
        if (e->arena16) (void)hipFree(e->arena16);
        if (e->arena32) (void)hipFree(e->arena32);
        delete e;
        return r;
    }
    e->layout();
    *out = e;
    return 0;
}

extern "C" int mi355_flux_destroy(mi355_flux* e) {
    if (!e) return 0;
    flux_train_release_engine(e);
    if (e->arena16) (void)hipFree(e->arena16);
    if (e->arena32) (void)hipFree(e->arena32);
    delete e;
    return 0;
}
extern "C" int mi355_flux_num_params(mi355_flux* e) { return e ? (int)e->names.size() : 0; }
extern "C" const char* mi355_flux_param_name(mi355_flux* e, int i) {
    if (!e || i < 0 || i >= (int)e->names.size()) return nullptr;
    return e->names[i].c_str();
}
extern "C" int mi355_flux_bind_weight(mi355_flux* e, const char* name, const void* src, int dtype, int ndim, const int64_t* shape,
                                      void* stream) {
    if (!e || !name || !src) return errorf("mi355_flux_bind_weight: null argument");
    auto it = e->slots.find(name);
    if (it == e->slots.end()) return errorf("mi355_flux_bind_weight: unknown parameter '%s'", name);
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    if (n != it->second.numel)
        return errorf("mi355_flux_bind_weight: '%s' has %lld elements, expected %lld", name, (long long)n, (long long)it->second.numel);
    if (dtype < 0 || dtype > 2) return errorf("mi355_flux_bind_weight: bad dtype %d", dtype);
    HIPCHK(launch_convert(src, dtype, it->second.dst, it->second.dst_dt, n, (hipStream_t)stream));
    it->second.bound = true;
    if (strstr(name, ".norm_")) e->bounds_dirty = true;
    flux_train_mark_dirty(e);          // the transposed copies the backward's dgrad GEMMs read are stale
    return 0;
}
extern "C" int mi355_flux_weights_ready(mi355_flux* e) {
    if (!e) return errorf("null engine");
    for (auto& kv : e->slots)
        if (!kv.second.bound) return errorf("parameter '%s' has not been bound", kv.first.c_str());
    return 0;
}

// -------------------------------------------------------------------------------------- plan
struct mi355_flux_plan {
    mi355_flux* e;
    int B, h, w, hp, wp, Ni, Nt, S, S_pad, Mi, Mc, M, max_steps;
    int64_t n_lat;      // packed elements per sample = Ni * in_channels
    char* ws = nullptr;
    size_t ws_bytes = 0;
    bf16_t *lat16, *x, *c, *c0, *xn, *cn, *y, *yn, *qkbuf, *q, *k, *vT, *o_img, *o_ctx, *big, *v;
    bf16_t *tproj, *gproj, *h1, *p1, *pemb, *gemb, *semb, *mod_all;
    float2* cs;
    float *t_dev, *g_dev, *scal;
    char *io_init, *io_traj;
    float *io_noise, *io_lp;
    bf16_t *io_pe, *io_pp;
    std::vector<float> host_t, host_sc;
    // two-stream double blocks (opt-in, mi355_tune_set key 14): the text chain on a plan-owned side stream with its own q|k staging and
    // MLP-hidden buffers (the single-stream path shares `qkbuf` / `big` between the two chains)
    hipStream_t side = nullptr;
    char* ws_side = nullptr;
    bf16_t *qkbuf_c = nullptr, *big_c = nullptr;
    std::vector<hipEvent_t> ev_join, ev_fork;   // per double block: text q|k|v ready (side -> main), attention done (main -> side); [L] = start / end
    // hipGraph of the N-step loop (opt-in, mi355_tune_set key 16): captured on a plan-owned stream on the second call of a configuration
    hipGraphExec_t gexec = nullptr;
    hipStream_t cap_stream = nullptr;
    bool warmed = false;
    int g_steps = -1, g_dyn = -1, g_storage = -1, g_init = -1, g_clp = -1, g_noise = -1, g_bounds = -1, g_two = -1, g_gemm = -1, g_attn = -1, g_tune = -1;
    float g_sigma_max = 0.f;
};

extern "C" int mi355_flux_plan_create(mi355_flux* e, int batch, int latent_h, int latent_w, int n_text, int max_steps,
                                      mi355_flux_plan** out) {
    if (!e || !out) return errorf("mi355_flux_plan_create: null argument");
    if (batch < 1 || n_text < 1 || max_steps < 1 || latent_h < 2 || latent_w < 2 || (latent_h | latent_w) & 1)
        return errorf("mi355_flux_plan_create: bad shape (latent size must be even)");
    mi355_flux_plan* p = new mi355_flux_plan();
    p->e = e; p->B = batch; p->h = latent_h; p->w = latent_w; p->hp = latent_h / 2; p->wp = latent_w / 2;
    p->Ni = p->hp * p->wp; p->Nt = n_text; p->S = p->Ni + p->Nt; p->S_pad = (p->S + 63) / 64 * 64;
    p->Mi = batch * p->Ni; p->Mc = batch * p->Nt; p->M = batch * p->S; p->max_steps = max_steps;
    p->n_lat = (int64_t)p->Ni * e->cfg.in_channels;
    const int D = e->D, F = e->F;
    const int64_t rows_cond = (int64_t)max_steps * batch;
    size_t off = 0;
    auto take = [&](int64_t elems, int esz) {
        size_t o = off;
        off += (((size_t)elems * esz) + 255) & ~(size_t)255;
        return o;
    };
    const int64_t qk_el = (int64_t)batch * e->H * p->S_pad * 128;
    const int64_t nl = (int64_t)batch * p->n_lat;
    size_t o_l16 = take(nl, 2), o_x = take((int64_t)p->Mi * D, 2), o_c = take((int64_t)p->Mc * D, 2), o_c0 = take((int64_t)p->Mc * D, 2);
    size_t o_xn = take((int64_t)p->Mi * D, 2), o_cn = take((int64_t)p->Mc * D, 2);
    size_t o_y = take((int64_t)p->M * D, 2), o_yn = take((int64_t)p->M * D, 2), o_qkb = take((int64_t)p->M * 2 * D, 2);
    size_t o_q = take(qk_el, 2), o_k = take(qk_el, 2), o_vT = take(qk_el, 2);
    size_t o_oi = take((int64_t)p->Mi * D, 2), o_oc = take((int64_t)p->Mc * D, 2);
    size_t o_big = take((int64_t)p->M * (D + F), 2);        // single blocks: [attn | gelu(mlp)]; double blocks: ff hidden (img, then ctx)
    size_t o_v = take(nl, 2);
    size_t o_tp = take(rows_cond * e->cfg.time_proj_dim, 2), o_gp = take((int64_t)batch * e->cfg.time_proj_dim, 2);
    size_t o_h1 = take(rows_cond * D, 2), o_p1 = take((int64_t)batch * D, 2), o_pemb = take((int64_t)batch * D, 2);
    size_t o_gemb = take((int64_t)batch * D, 2), o_semb = take(rows_cond * D, 2), o_mod = take(rows_cond * e->mod_cols, 2);
    size_t o_cs = take((int64_t)p->S * 64, 8);
    size_t o_t = take(rows_cond, 4), o_g = take(batch, 4), o_sc = take(3 * (int64_t)max_steps, 4);
    size_t o_ii = take(nl, 4), o_it = take((int64_t)(max_steps + 1) * nl, 4), o_in = take((int64_t)max_steps * nl, 4);
    size_t o_il = take((int64_t)max_steps * batch, 4);
    size_t o_ipe = take((int64_t)batch * p->Nt * e->cfg.joint_attention_dim, 2), o_ipp = take((int64_t)batch * e->cfg.pooled_projection_dim, 2);
    p->ws_bytes = off;
    if (hipMalloc((void**)&p->ws, off) != hipSuccess) {
        int r = errorf("mi355_flux_plan_create: hipMalloc of %zu bytes failed", off);
        delete p;
        return r;
    }
    if (hipMemset(p->ws, 0, off) != hipSuccess) {   // padded key rows / columns of q,k,vT must stay finite
        (void)hipFree(p->ws);
        delete p;
        return errorf("mi355_flux_plan_create: hipMemset failed");
    }
    char* w = p->ws;
    p->lat16 = (bf16_t*)(w + o_l16); p->x = (bf16_t*)(w + o_x); p->c = (bf16_t*)(w + o_c); p->c0 = (bf16_t*)(w + o_c0);
    p->xn = (bf16_t*)(w + o_xn); p->cn = (bf16_t*)(w + o_cn); p->y = (bf16_t*)(w + o_y); p->yn = (bf16_t*)(w + o_yn);
    p->qkbuf = (bf16_t*)(w + o_qkb); p->q = (bf16_t*)(w + o_q); p->k = (bf16_t*)(w + o_k); p->vT = (bf16_t*)(w + o_vT);
    p->o_img = (bf16_t*)(w + o_oi); p->o_ctx = (bf16_t*)(w + o_oc); p->big = (bf16_t*)(w + o_big); p->v = (bf16_t*)(w + o_v);
    p->tproj = (bf16_t*)(w + o_tp); p->gproj = (bf16_t*)(w + o_gp); p->h1 = (bf16_t*)(w + o_h1); p->p1 = (bf16_t*)(w + o_p1);
    p->pemb = (bf16_t*)(w + o_pemb); p->gemb = (bf16_t*)(w + o_gemb); p->semb = (bf16_t*)(w + o_semb); p->mod_all = (bf16_t*)(w + o_mod);
    p->cs = (float2*)(w + o_cs); p->t_dev = (float*)(w + o_t); p->g_dev = (float*)(w + o_g); p->scal = (float*)(w + o_sc);
    p->io_init = w + o_ii; p->io_traj = w + o_it; p->io_noise = (float*)(w + o_in); p->io_lp = (float*)(w + o_il);
    p->io_pe = (bf16_t*)(w + o_ipe); p->io_pp = (bf16_t*)(w + o_ipp);
    // rotary table of the joint sequence [txt (ids 0) | img (0, row, col)]: FluxPosEmbed evaluates the angles in float64
    // (get_1d_rotary_pos_embed(freqs_dtype=float64)) and casts cos / sin to fp32
    {
        std::vector<float> cs((size_t)p->S * 128);
        const int* ax = e->cfg.axes_dims_rope;
        for (int s = 0; s < p->S; ++s) {
            double pos[3] = {0.0, 0.0, 0.0};
            if (s >= p->Nt) { const int t = s - p->Nt; pos[1] = t / p->wp; pos[2] = t % p->wp; }
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
            return errorf("mi355_flux_plan_create: rotary table upload failed");
        }
    }
    *out = p;
    return 0;
}

extern "C" int mi355_flux_plan_destroy(mi355_flux_plan* p) {
    if (!p) return 0;
    flux_train_release(p);
    if (p->gexec) (void)hipGraphExecDestroy(p->gexec);
    if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
    for (hipEvent_t ev : p->ev_join) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : p->ev_fork) (void)hipEventDestroy(ev);
    if (p->side) (void)hipStreamDestroy(p->side);
    if (p->ws_side) (void)hipFree(p->ws_side);
    if (p->ws) (void)hipFree(p->ws);
    delete p;
    return 0;
}
extern "C" int64_t mi355_flux_plan_workspace_bytes(mi355_flux_plan* p) { return p ? (int64_t)p->ws_bytes : 0; }

// ---------------------------------------------------------------------------------- forward
namespace {

// |score| <= 128 / sqrt(128) * log2(e) * max|w_q| * max|w_k| (RoPE preserves norms; see engine.hip update_score_bounds)
int update_score_bounds(mi355_flux* e, hipStream_t st) {
    if (!e->bounds_dirty) return 0;
    std::vector<float> host(e->used32 / 4);
    HIPCHK(hipMemcpyAsync(host.data(), e->arena32, e->used32, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    auto amax = [&](const float* dev) {
        const float* h = host.data() + (dev - (const float*)e->arena32);
        float m = 0.f;
        for (int i = 0; i < 128; ++i) m = fmaxf(m, fabsf(h[i]));
        return m;
    };
    const float c = 11.313708f * 1.4426950408889634f * 1.02f;
    for (auto& b : e->dbl) b.bound = c * fmaxf(amax(b.nq), amax(b.ncq)) * fmaxf(amax(b.nk), amax(b.nck));
    for (auto& b : e->sgl) b.bound = c * amax(b.nq) * amax(b.nk);
    e->bounds_dirty = false;
    ++e->bounds_ver;
    return 0;
}

// step-invariant work: context embedder, pooled-text MLP, guidance MLP -> pemb = text_emb + guidance_emb
int prepare_prompt(mi355_flux_plan* p, hipStream_t st, const void* enc, const void* pooled) {
    mi355_flux* e = p->e;
    const int D = e->D, J = e->cfg.joint_attention_dim, P = e->cfg.pooled_projection_dim, T = e->cfg.time_proj_dim;
    GemmParams g = make_gemm((const bf16_t*)enc, J, e->w_ctx, J, p->Mc, D, J, EPI_BIAS, e->b_ctx, p->c0, D);
    HIPCHK(launch_gemm(g, st));
    GemmParams g1 = make_gemm((const bf16_t*)pooled, P, e->w_p1, P, p->B, D, P, EPI_BIAS_SILU, e->b_p1, p->p1, D);
    HIPCHK(launch_gemm(g1, st));
    GemmParams g2 = make_gemm(p->p1, D, e->w_p2, D, p->B, D, D, EPI_BIAS, e->b_p2, p->pemb, D);
    HIPCHK(launch_gemm(g2, st));
    if (e->cfg.guidance_embeds) {
        HIPCHK(launch_time_proj(p->g_dev, p->B, T, DT_F32, p->gproj, st));
        GemmParams g3 = make_gemm(p->gproj, T, e->w_g1, T, p->B, D, T, EPI_BIAS_SILU, e->b_g1, p->p1, D);
        HIPCHK(launch_gemm(g3, st));
        GemmParams g4 = make_gemm(p->p1, D, e->w_g2, D, p->B, D, D, EPI_POSADD, e->b_g2, p->gemb, D);    // + text_emb
        g4.aux = p->pemb; g4.ld_aux = D; g4.rows_per_sample = p->B;
        HIPCHK(launch_gemm(g4, st));
    } else {
        HIPCHK(copy_d2d(p->gemb, p->pemb, (size_t)p->B * D * 2, st));
    }
    return 0;
}

// conditioning of `nsteps` steps at once: temb = t_emb + (g_emb + p_emb), semb = silu(temb), mod_all = every AdaLN linear
int prepare_conditioning(mi355_flux_plan* p, hipStream_t st, int nsteps) {
    mi355_flux* e = p->e;
    const int D = e->D, T = e->cfg.time_proj_dim;
    const int rows = nsteps * p->B;
    HIPCHK(launch_time_proj(p->t_dev, rows, T, DT_F32, p->tproj, st));
    GemmParams g1 = make_gemm(p->tproj, T, e->w_t1, T, rows, D, T, EPI_BIAS_SILU, e->b_t1, p->h1, D);
    HIPCHK(launch_gemm(g1, st));
    GemmParams g2 = make_gemm(p->h1, D, e->w_t2, D, rows, D, D, EPI_ADDSRC_SILU, e->b_t2, p->semb, D);
    g2.aux = p->gemb; g2.ld_aux = D; g2.rows_per_sample = p->B;
    HIPCHK(launch_gemm(g2, st));
    GemmParams g3 = make_gemm(p->semb, D, e->w_mod, D, rows, e->mod_cols, D, EPI_BIAS, e->b_mod, p->mod_all, e->mod_cols);
    HIPCHK(launch_gemm(g3, st));
    return 0;
}

int ln_mod(mi355_flux_plan* p, hipStream_t st, const bf16_t* x, bf16_t* out, const bf16_t* mod, int M, int rps, int shift_off,
           int scale_off) {
    LnModParams l;
    memset(&l, 0, sizeof(l));
    l.x = x; l.out = out; l.out2 = nullptr; l.mod = mod; l.mod_ld = p->e->mod_cols;
    l.shift_off = shift_off; l.scale_off = scale_off;
    l.M = M; l.D = p->e->D; l.rows_per_sample = rps; l.eps = p->e->cfg.eps;
    HIPCHK(launch_ln_mod(l, st));
    return 0;
}

// q|k projection (one GEMM) -> RMSNorm + RoPE + scatter; V^T projection with the scatter fused (operands swapped)
int qkv(mi355_flux_plan* p, hipStream_t st, const bf16_t* xin, int M, int rps, int s_off, const bf16_t* w_qk, const float* b_qk,
        const bf16_t* w_v, const float* b_v, const float* nq, const float* nk, bf16_t* qkb = nullptr) {
    mi355_flux* e = p->e;
    const int D = e->D;
    if (!qkb) qkb = p->qkbuf;
    GemmParams g = make_gemm(xin, D, w_qk, D, M, 2 * D, D, EPI_BIAS, b_qk, qkb, 2 * D);
    HIPCHK(launch_gemm(g, st));
    RopeNormParams r;
    memset(&r, 0, sizeof(r));
    r.src = qkb; r.src_ld = 2 * D; r.q_col = 0; r.k_col = D; r.nw_q = nq; r.nw_k = nk; r.cs = p->cs;
    r.q_out = p->q; r.k_out = p->k; r.M = M; r.H = e->H; r.rows_per_sample = rps; r.s_off = s_off; r.S_pad = p->S_pad;
    r.eps = e->cfg.eps; r.q_scale = 0.08838834764831845f * 1.4426950408889634f;
    HIPCHK(launch_rope_norm(r, st));
    GemmParams gv = make_gemm(w_v, D, xin, D, D, M, D, EPI_VT, b_v, nullptr, 0);
    gv.q = p->vT; gv.H = e->H; gv.S_pad = p->S_pad; gv.s_off = s_off; gv.rows_per_sample = rps; gv.hd_shift = 7;
    HIPCHK(launch_gemm(gv, st));
    return 0;
}

int gate_res(mi355_flux_plan* p, hipStream_t st, const bf16_t* A, long lda, int K, const bf16_t* W, const float* bias, bf16_t* x,
             int M, int rps, const bf16_t* mod, int gate_off) {
    GemmParams g = make_gemm(A, lda, W, K, M, p->e->D, K, EPI_GATE_RES, bias, x, p->e->D);
    g.aux = mod + gate_off; g.ld_aux = p->e->mod_cols; g.rows_per_sample = rps;
    HIPCHK(launch_gemm(g, st));
    return 0;
}

int attention(mi355_flux_plan* p, hipStream_t st, bf16_t* o_first, long ld_first, int n_first, bf16_t* o_rest, long ld_rest, float bound) {
    Attn128Params a;
    memset(&a, 0, sizeof(a));
    a.q = p->q; a.k = p->k; a.vT = p->vT; a.o_first = o_first; a.ld_first = ld_first; a.n_first = n_first;
    a.o_rest = o_rest; a.ld_rest = ld_rest; a.B = p->B; a.H = p->e->H; a.S = p->S; a.S_pad = p->S_pad; a.q_prescaled = 1;
    a.score_bound = bound;
    HIPCHK(launch_attention128(a, st));
    return 0;
}

// one transformer forward: packed latents (storage dtype) -> packed velocity v_out [B][Ni][C] bf16.  `mod` = this step's
// rows of mod_all; c0 / conditioning prepared.
// Two-stream double blocks (tune key 14; ON by default for plans of up to 16 384 image rows since round 3): as in the
// SD3.5 and Qwen-Image engines, the text chain of a double block runs on a plan-owned side stream beside the image chain (the reference's
// own FLUX.1 examples sample at B = 1-2 and 384^2 / 512^2: 576-1024 image tokens next to 512 text tokens, every grid a fraction of the
// chip).  Join before the joint attention, fork after it, last join before the two streams are concatenated for the single blocks.
// Bit-identical to the single-stream order.  Measured on MI355X (profiles/r03a_flux_two_stream_ab.txt, hipGraph replay, denoise-steps/s single ->
// two streams): 384^2 B = 1 30.9 -> 37.7 (+22 %), 512^2 B = 2 37.6 -> 39.8 (+6 %), 1024^2 B = 1 13.2 -> 13.9 (+5 %), B = 2 14.1 -> 14.5 (+3 %),
// B = 8 15.39 -> 15.39 (32 768 rows: above the threshold, single stream).
int g_flux_two_stream = 2;          // 0 off, 1 on, 2 (default) on for plans with at most g_flux_two_stream_rows image rows
int g_flux_two_stream_rows = 16384;

// key 16: replay the N-step loop of mi355_flux_rollout as ONE hipGraph, like the SD3.5 engine's captured rollout (engine.hip, key 2).  ON by
// default since round 3; measured bit-identical and +0.1 ... +0.8 % over eager launches at the reference's example shapes, +-0.1 % at
// B = 8 (same file): the host keeps up with ~700 launches of 5-15 us, the graph mainly takes the CPU out of the loop.
int g_flux_graph = 1;

bool flux_two_stream_wanted(const mi355_flux_plan* p) {
    return g_flux_two_stream == 1 || (g_flux_two_stream == 2 && p->Mi <= g_flux_two_stream_rows);
}

int flux_two_stream_init(mi355_flux_plan* p) {
    if (p->side) return 0;
    const size_t qkb = (((size_t)p->Mc * 2 * p->e->D * 2) + 255) & ~(size_t)255, big = (((size_t)p->Mc * p->e->F * 2) + 255) & ~(size_t)255;
    HIPCHK(hipMalloc((void**)&p->ws_side, qkb + big));
    p->qkbuf_c = (bf16_t*)p->ws_side;
    p->big_c = (bf16_t*)(p->ws_side + qkb);
    for (int i = 0; i <= p->e->L; ++i) {
        hipEvent_t a, b;
        HIPCHK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        p->ev_join.push_back(a);
        HIPCHK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        p->ev_fork.push_back(b);
    }
    HIPCHK(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
    return 0;
}

int forward_core(mi355_flux_plan* p, hipStream_t st, const void* latents, int lat_dt, const bf16_t* mod, bf16_t* v_out) {
    mi355_flux* e = p->e;
    const int D = e->D, F = e->F, C = e->cfg.in_channels;
    const int Ni = p->Ni, Nt = p->Nt, S = p->S;
    const bool two = flux_two_stream_wanted(p) && e->L > 0;
    hipStream_t ts = st;                               // carries the text chain of the double blocks
    bf16_t *qkb_c = p->qkbuf, *big_c = p->big;
    if (two) {
        CHK(flux_two_stream_init(p));
        ts = p->side; qkb_c = p->qkbuf_c; big_c = p->big_c;
    }
    const bf16_t* lat = (const bf16_t*)latents;
    if (lat_dt != DT_BF16) {
        HIPCHK(launch_convert(latents, lat_dt, p->lat16, DT_BF16, (long)p->B * p->n_lat, st));
        lat = p->lat16;
    }
    GemmParams gx = make_gemm(lat, C, e->w_x, C, p->Mi, D, C, EPI_BIAS, e->b_x, p->x, D);
    HIPCHK(launch_gemm(gx, st));
    HIPCHK(copy_d2d(p->c, p->c0, (size_t)p->Mc * D * 2, st));
    if (two) {          // c, the conditioning (modulation table, prompt) and the previous forward are complete on `st`
        HIPCHK(ev_record(p->ev_fork[e->L], st));
        HIPCHK(ev_wait(ts, p->ev_fork[e->L]));
    }
    for (int i = 0; i < e->L; ++i) {
        const DoubleW& b = e->dbl[i];
        const int mi = b.mod_img, mc = b.mod_ctx;     // chunks: shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
        CHK(ln_mod(p, ts, p->c, p->cn, mod, p->Mc, Nt, mc, mc + D));
        CHK(qkv(p, ts, p->cn, p->Mc, Nt, 0, b.w_cqk, b.b_cqk, b.w_cv, b.b_cv, b.ncq, b.nck, qkb_c));
        CHK(ln_mod(p, st, p->x, p->xn, mod, p->Mi, Ni, mi, mi + D));
        CHK(qkv(p, st, p->xn, p->Mi, Ni, Nt, b.w_qk, b.b_qk, b.w_v, b.b_v, b.nq, b.nk));
        if (two) {      // join: the attention reads the text rows of q / k / vT
            HIPCHK(ev_record(p->ev_join[i], ts));
            HIPCHK(ev_wait(st, p->ev_join[i]));
        }
        CHK(attention(p, st, p->o_ctx, D, Nt, p->o_img, D, b.bound));
        if (two) {      // fork: o_ctx is written, and the text rows of q / k / vT are free for the next block's text projections
            HIPCHK(ev_record(p->ev_fork[i], st));
            HIPCHK(ev_wait(ts, p->ev_fork[i]));
        }
        CHK(gate_res(p, st, p->o_img, D, D, b.w_o, b.b_o, p->x, p->Mi, Ni, mod, mi + 2 * D));
        CHK(gate_res(p, ts, p->o_ctx, D, D, b.w_co, b.b_co, p->c, p->Mc, Nt, mod, mc + 2 * D));
        CHK(ln_mod(p, st, p->x, p->xn, mod, p->Mi, Ni, mi + 3 * D, mi + 4 * D));
        GemmParams f1 = make_gemm(p->xn, D, b.w_ff1, D, p->Mi, F, D, EPI_BIAS_GELU, b.b_ff1, p->big, F);
        HIPCHK(launch_gemm(f1, st));
        CHK(gate_res(p, st, p->big, F, F, b.w_ff2, b.b_ff2, p->x, p->Mi, Ni, mod, mi + 5 * D));
        CHK(ln_mod(p, ts, p->c, p->cn, mod, p->Mc, Nt, mc + 3 * D, mc + 4 * D));
        GemmParams c1 = make_gemm(p->cn, D, b.w_cff1, D, p->Mc, F, D, EPI_BIAS_GELU, b.b_cff1, big_c, F);
        HIPCHK(launch_gemm(c1, ts));
        CHK(gate_res(p, ts, big_c, F, F, b.w_cff2, b.b_cff2, p->c, p->Mc, Nt, mod, mc + 5 * D));
    }
    if (two) {          // the text stream is complete before it is concatenated with the image stream
        HIPCHK(ev_record(p->ev_join[e->L], ts));
        HIPCHK(ev_wait(st, p->ev_join[e->L]));
    }
    // joint stream y = cat([c, x], dim=1) per sample
    const size_t rowb = (size_t)D * 2;
    if (sched_trace_on()) {
        sched_trace_launch("copy2d.c->y", st, {treg(p->c, (size_t)p->B * Nt * rowb)}, {tregs(p->y, (size_t)Nt * rowb, (size_t)S * rowb, (size_t)p->B)});
        sched_trace_launch("copy2d.x->y", st, {treg(p->x, (size_t)p->B * Ni * rowb)}, {tregs(p->y + (size_t)Nt * D, (size_t)Ni * rowb, (size_t)S * rowb, (size_t)p->B)});
    }
    HIPCHK(hipMemcpy2DAsync(p->y, (size_t)S * rowb, p->c, (size_t)Nt * rowb, (size_t)Nt * rowb, p->B, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpy2DAsync(p->y + (size_t)Nt * D, (size_t)S * rowb, p->x, (size_t)Ni * rowb, (size_t)Ni * rowb, p->B,
                            hipMemcpyDeviceToDevice, st));
    for (int i = 0; i < e->LS; ++i) {
        const SingleW& b = e->sgl[i];
        const int m0 = b.mod;                          // chunks: shift, scale, gate
        CHK(ln_mod(p, st, p->y, p->yn, mod, p->M, S, m0, m0 + D));
        CHK(qkv(p, st, p->yn, p->M, S, 0, b.w_qk, b.b_qk, b.w_v, b.b_v, b.nq, b.nk));
        GemmParams gm = make_gemm(p->yn, D, b.w_mlp, D, p->M, F, D, EPI_BIAS_GELU, b.b_mlp, p->big + D, D + F);
        HIPCHK(launch_gemm(gm, st));
        CHK(attention(p, st, p->big, D + F, S, p->big, D + F, b.bound));
        CHK(gate_res(p, st, p->big, D + F, D + F, b.w_out, b.b_out, p->y, p->M, S, mod, m0 + 2 * D));
    }
    // image rows back to a contiguous stream, AdaLayerNormContinuous (scale first), proj_out
    if (sched_trace_on())
        sched_trace_launch("copy2d.y->x", st, {tregs(p->y + (size_t)Nt * D, (size_t)Ni * rowb, (size_t)S * rowb, (size_t)p->B)}, {treg(p->x, (size_t)p->B * Ni * rowb)});
    HIPCHK(hipMemcpy2DAsync(p->x, (size_t)Ni * rowb, p->y + (size_t)Nt * D, (size_t)S * rowb, (size_t)Ni * rowb, p->B,
                            hipMemcpyDeviceToDevice, st));
    CHK(ln_mod(p, st, p->x, p->xn, mod, p->Mi, Ni, e->mod_out + D, e->mod_out));
    GemmParams go = make_gemm(p->xn, D, e->w_proj, D, p->Mi, C, D, EPI_BIAS, e->b_proj, v_out, C);
    HIPCHK(launch_gemm(go, st));
    return 0;
}

int sde_call(hipStream_t st, int batch, int64_t n, const bf16_t* v, const void* latents, int lat_dtype, const float* noise,
             const float* sigma, const float* sigma_next, const float* eta, float sigma_max, int dynamics, int compute_log_prob,
             void* next_out, float* log_prob) {
    SdeStepParams s;
    memset(&s, 0, sizeof(s));
    s.v_text = v; s.v_uncond = nullptr; s.v_dt = DT_BF16; s.guidance = 1.0f;
    s.latents = latents; s.lat_dt = lat_dtype; s.noise = noise;
    s.sigma = sigma; s.sigma_next = sigma_next; s.eta = eta; s.scalar_stride = 0; s.sigma_max = sigma_max;
    s.dynamics = dynamics; s.compute_log_prob = compute_log_prob; s.B = batch; s.n = n;
    s.next_out = next_out; s.next_out_dt = lat_dtype; s.log_prob = log_prob;
    HIPCHK(launch_sde_step(s, st));
    return 0;
}

}  // namespace

namespace mi355 {
void set_flux_two_stream(int mode) { g_flux_two_stream = mode; }
void set_flux_two_stream_rows(int rows) { g_flux_two_stream_rows = rows; }
void set_flux_graph(int on) { g_flux_graph = on; }
}  // namespace mi355

// transformer only (replay / tests): t_model[B] and guidance_model[B] are the values the network embeds (device fp32):
// the adapter passes t/1000 and the model multiplies by 1000 in the latents' dtype (see mi355_flux_rollout)
extern "C" int mi355_flux_forward(mi355_flux_plan* p, void* stream, const void* latents, int lat_dtype, const float* t_model,
                                  const float* guidance_model, const void* prompt_embeds, const void* pooled, void* v_out) {
    if (!p || !latents || !t_model || !prompt_embeds || !pooled || !v_out) return errorf("mi355_flux_forward: null argument");
    if (p->e->cfg.guidance_embeds && !guidance_model) return errorf("mi355_flux_forward: this model embeds guidance, guidance_model is NULL");
    if (lat_dtype < 0 || lat_dtype > 2) return errorf("mi355_flux_forward: bad latent dtype %d", lat_dtype);
    CHK(mi355_flux_weights_ready(p->e));
    hipStream_t st = (hipStream_t)stream;
    CHK(update_score_bounds(p->e, st));
    HIPCHK(copy_d2d(p->t_dev, t_model, (size_t)p->B * 4, st));
    if (guidance_model) HIPCHK(copy_d2d(p->g_dev, guidance_model, (size_t)p->B * 4, st));
    CHK(prepare_prompt(p, st, prompt_embeds, pooled));
    CHK(prepare_conditioning(p, st, 1));
    return forward_core(p, st, latents, lat_dtype, p->mod_all, (bf16_t*)v_out);
}

// The whole N-step rollout (flux1.py:222-259) with zero host syncs.  timesteps_host: scheduler timesteps in [0, 1000];
// sigmas_host: scheduler sigmas (sigma_max = sigmas[1]); noise_levels_host: eta per step; guidance_scale: embedded guidance.
// Latents are PACKED [B][Ni][in_channels]; step_noise fp32 [n_steps][B][Ni*in_channels].
extern "C" int mi355_flux_rollout(mi355_flux_plan* p, void* stream, int n_steps, const float* timesteps_host, const float* sigmas_host,
                                  const float* noise_levels_host, int dynamics, float guidance_scale, const void* init_latents,
                                  int init_dtype, int storage_dtype, const float* step_noise, const void* prompt_embeds,
                                  const void* pooled, const int32_t* keep_slot_host, void* out_latents, float* out_log_probs,
                                  void* out_final, int compute_log_prob) {
    if (!p || !timesteps_host || !sigmas_host || !noise_levels_host || !init_latents || !prompt_embeds || !pooled)
        return errorf("mi355_flux_rollout: null argument");
    if (n_steps < 1 || n_steps > p->max_steps) return errorf("mi355_flux_rollout: n_steps %d exceeds the plan's max_steps %d", n_steps, p->max_steps);
    if (storage_dtype < 0 || storage_dtype > 2 || init_dtype < 0 || init_dtype > 2) return errorf("mi355_flux_rollout: bad dtype");
    if (!step_noise && dynamics != MI355_ODE) return errorf("mi355_flux_rollout: step_noise is NULL");
    if (dynamics < 0 || dynamics > 3) return errorf("mi355_flux_rollout: unknown dynamics %d", dynamics);
    CHK(mi355_flux_weights_ready(p->e));
    hipStream_t st = (hipStream_t)stream;
    CHK(update_score_bounds(p->e, st));
    const int B = p->B;
    std::vector<float>& tt = p->host_t;
    std::vector<float>& sc = p->host_sc;
    tt.assign((size_t)n_steps * B + B, 0.f);
    sc.assign(3 * (size_t)p->max_steps, 0.f);
    for (int i = 0; i < n_steps; ++i) {
        // flux1.py:325 passes t/1000 (fp32); FluxTransformer2DModel does `timestep.to(hidden_states.dtype) * 1000` with the
        // hidden states in the latent storage dtype
        const float tm = host_round(host_round(timesteps_host[i] / 1000.0f, storage_dtype) * 1000.0f, storage_dtype);
        for (int j = 0; j < B; ++j) tt[(size_t)i * B + j] = tm;
        const float t_next = (i + 1 < n_steps) ? timesteps_host[i + 1] : 0.0f;
        sc[i] = timesteps_host[i] / 1000.0f;
        sc[p->max_steps + i] = t_next / 1000.0f;
        sc[2 * p->max_steps + i] = noise_levels_host[i];
    }
    // guidance = as_tensor(guidance_scale, dtype=latents.dtype) (flux1.py:319), then `* 1000` in that dtype
    const float gm = host_round(host_round(guidance_scale, storage_dtype) * 1000.0f, storage_dtype);
    for (int j = 0; j < B; ++j) tt[(size_t)n_steps * B + j] = gm;
    HIPCHK(hipMemcpyAsync(p->t_dev, tt.data(), (size_t)n_steps * B * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(p->g_dev, tt.data() + (size_t)n_steps * B, (size_t)B * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(p->scal, sc.data(), sc.size() * 4, hipMemcpyHostToDevice, st));
    const int64_t nl = (int64_t)B * p->n_lat;
    const size_t in_esz = init_dtype == MI355_F32 ? 4 : 2;
    HIPCHK(copy_d2d(p->io_init, init_latents, nl * in_esz, st));
    if (step_noise) HIPCHK(copy_d2d(p->io_noise, step_noise, (size_t)n_steps * nl * 4, st));
    HIPCHK(copy_d2d(p->io_pe, prompt_embeds, (size_t)B * p->Nt * p->e->cfg.joint_attention_dim * 2, st));
    HIPCHK(copy_d2d(p->io_pp, pooled, (size_t)B * p->e->cfg.pooled_projection_dim * 2, st));
    const float sigma_max = sigmas_host[1];
    const int clp = compute_log_prob && out_log_probs;
    const size_t esz = storage_dtype == MI355_F32 ? 4 : 2;
    const size_t lat_bytes = (size_t)nl * esz;
    // everything below reads / writes plan-owned buffers at fixed addresses (staged inputs, t_dev / g_dev / scal, io_traj, io_lp)
    auto body = [&](hipStream_t s) -> int {
        CHK(prepare_prompt(p, s, p->io_pe, p->io_pp));
        CHK(prepare_conditioning(p, s, n_steps));
        HIPCHK(launch_convert(p->io_init, init_dtype, p->io_traj, storage_dtype, (long)nl, s));      // cast_latents(init)
        for (int i = 0; i < n_steps; ++i) {
            const bf16_t* mod = p->mod_all + (int64_t)i * B * p->e->mod_cols;
            char* cur = p->io_traj + (size_t)i * lat_bytes;
            char* nxt = p->io_traj + (size_t)(i + 1) * lat_bytes;
            CHK(forward_core(p, s, cur, storage_dtype, mod, p->v));
            CHK(sde_call(s, B, p->n_lat, p->v, cur, storage_dtype, step_noise ? p->io_noise + (int64_t)i * nl : nullptr, p->scal + i,
                         p->scal + p->max_steps + i, p->scal + 2 * p->max_steps + i, sigma_max, dynamics, clp ? 2 : 0, nxt,
                         clp ? p->io_lp + (int64_t)i * B : nullptr));
        }
        return 0;
    };
    bool launched = false;
    if (g_flux_graph && p->warmed) {
        const int two = (int)flux_two_stream_wanted(p);
        const bool same = p->gexec && p->g_steps == n_steps && p->g_dyn == dynamics && p->g_storage == storage_dtype && p->g_init == init_dtype &&
                          p->g_clp == clp && p->g_noise == (int)(step_noise != nullptr) && p->g_sigma_max == sigma_max &&
                          p->g_bounds == p->e->bounds_ver && p->g_two == two && p->g_gemm == get_gemm_variant() && p->g_attn == get_attn128_variant() && p->g_tune == tune_epoch();
        if (!same) {
            if (two) CHK(flux_two_stream_init(p));              // streams / events / buffers are created outside the capture
            if (p->gexec) { (void)hipGraphExecDestroy(p->gexec); p->gexec = nullptr; }
            hipGraph_t graph = nullptr;
            hipError_t ce = hipSuccess;
            if (!p->cap_stream) ce = hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking);
            if (ce == hipSuccess) ce = hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeRelaxed);
            if (ce == hipSuccess) {
                const int rc = body(p->cap_stream);             // nothing executes: launches / D2D copies become graph nodes
                ce = hipStreamEndCapture(p->cap_stream, &graph);
                if (rc != 0 || ce != hipSuccess || !graph) {
                    if (graph) (void)hipGraphDestroy(graph);
                    graph = nullptr;
                }
            }
            if (graph) {
                ce = hipGraphInstantiate(&p->gexec, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (ce != hipSuccess) p->gexec = nullptr;
            }
            if (!p->gexec) {                                    // no silent fallback: the caller chooses eager launches with key 16 = 0
                const hipError_t last = hipGetLastError();
                return errorf("mi355_flux_rollout: hipGraph capture / instantiation of the %d-step loop failed (%s); mi355_tune_set(16, 0) "
                              "selects eager launches", n_steps, hipGetErrorString(ce != hipSuccess ? ce : last));
            }
            p->g_steps = n_steps; p->g_dyn = dynamics; p->g_storage = storage_dtype; p->g_init = init_dtype; p->g_clp = clp;
            p->g_noise = (int)(step_noise != nullptr); p->g_sigma_max = sigma_max; p->g_bounds = p->e->bounds_ver; p->g_two = two;
            p->g_gemm = get_gemm_variant(); p->g_attn = get_attn128_variant(); p->g_tune = tune_epoch();
        }
        HIPCHK(hipGraphLaunch(p->gexec, st));
        launched = true;
    }
    if (!launched) {
        CHK(body(st));
        p->warmed = true;
    }
    if (keep_slot_host && out_latents)
        for (int i = 0; i <= n_steps; ++i)
            if (keep_slot_host[i] >= 0)
                HIPCHK(hipMemcpyAsync((char*)out_latents + (size_t)keep_slot_host[i] * lat_bytes, p->io_traj + (size_t)i * lat_bytes,
                                      lat_bytes, hipMemcpyDeviceToDevice, st));
    if (clp)
        for (int i = 0; i < n_steps; ++i)
            if (noise_levels_host[i] > 0.f)
                HIPCHK(copy_d2d(out_log_probs + (int64_t)i * B, p->io_lp + (int64_t)i * B, (size_t)B * 4, st));
    if (out_final)
        HIPCHK(copy_d2d(out_final, p->io_traj + (size_t)n_steps * lat_bytes, lat_bytes, st));
    return 0;
}

// ----------------------------------------------------------------------- operator-level API
extern "C" int mi355_op_attention128(void* stream, const void* q, const void* k, const void* vT, void* o_first, int64_t ld_first,
                                     int n_first, void* o_rest, int64_t ld_rest, int B, int H, int S, int S_pad, int q_prescaled) {
    if (!q || !k || !vT || (n_first > 0 && !o_first)) return errorf("mi355_op_attention128: null argument");
    if (n_first < S && !o_rest) return errorf("mi355_op_attention128: o_rest is NULL but n_first < S");
    Attn128Params a;
    memset(&a, 0, sizeof(a));
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.vT = (const bf16_t*)vT; a.o_first = (bf16_t*)o_first; a.ld_first = ld_first;
    a.n_first = n_first; a.o_rest = (bf16_t*)o_rest; a.ld_rest = ld_rest; a.B = B; a.H = H; a.S = S; a.S_pad = S_pad;
    a.q_prescaled = q_prescaled;
    a.score_bound = (float)get_attn128_op_bound();
    HIPCHK(launch_attention128(a, (hipStream_t)stream));
    return 0;
}

extern "C" int mi355_op_rope_norm(void* stream, const void* src, int64_t src_ld, int q_col, int k_col, const float* nw_q,
                                  const float* nw_k, const float* cos_sin, void* q_out, void* k_out, int M, int H,
                                  int rows_per_sample, int s_off, int S_pad, float eps, float q_scale) {
    if (!src || !nw_q || !nw_k || !cos_sin || !q_out || !k_out) return errorf("mi355_op_rope_norm: null argument");
    RopeNormParams r;
    memset(&r, 0, sizeof(r));
    r.src = (const bf16_t*)src; r.src_ld = src_ld; r.q_col = q_col; r.k_col = k_col; r.nw_q = nw_q; r.nw_k = nw_k;
    r.cs = (const float2*)cos_sin; r.q_out = (bf16_t*)q_out; r.k_out = (bf16_t*)k_out; r.M = M; r.H = H;
    r.rows_per_sample = rows_per_sample; r.s_off = s_off; r.S_pad = S_pad; r.eps = eps; r.q_scale = q_scale;
    HIPCHK(launch_rope_norm(r, (hipStream_t)stream));
    return 0;
}

// Wan q / k producer (across-head RMSNorm, optional 3-D RoPE, head-major scatter) as an operator: unit tests of the row-norm output
extern "C" int mi355_op_norm_rope_full(void* stream, const void* src, int64_t src_ld, int col, const float* weight, const float* cos_sin,
                                       void* out, int M, int H, int rows_per_sample, int S_pad, float eps, float out_scale, void* max2) {
    if (!src || !weight || !out) return errorf("mi355_op_norm_rope_full: null argument");
    NormRopeFullParams r;
    memset(&r, 0, sizeof(r));
    r.src = (const bf16_t*)src; r.src_ld = src_ld; r.col = col; r.weight = weight; r.cs = (const float2*)cos_sin; r.out = (bf16_t*)out;
    r.M = M; r.H = H; r.rows_per_sample = rows_per_sample; r.s_off = 0; r.S_pad = S_pad; r.eps = eps; r.out_scale = out_scale;
    r.max2 = (unsigned*)max2;
    float* part = nullptr;
    if (max2) {
        if (rows_per_sample <= 0 || M % rows_per_sample) return errorf("mi355_op_norm_rope_full: max2 needs whole samples");
        HIPCHK(hipMalloc((void**)&part, (size_t)(M / rows_per_sample) * norm_rope_parts(rows_per_sample) * H * 4));
        r.max2_part = part;
    }
    hipError_t e = launch_norm_rope_full(r, (hipStream_t)stream);
    if (part) { if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream); (void)hipFree(part); }
    if (e != hipSuccess) return errorf("mi355_op_norm_rope_full: %s", hipGetErrorString(e));
    return 0;
}

#include "flux_train.inc"
