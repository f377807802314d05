// mi355_flow -- Qwen-Image rollout engine behind the C ABI (include/mi355_flow.h, mi355_qwen_*): SURVEY.md 8(f) row N4 (config E).
// Replaces the two `self.transformer(...)` calls (cond / uncond), the norm-rescaled true-CFG combine and `self.scheduler.step(...)`
// inside the denoising loop of QwenImageAdapter.inference / .forward (reference src/flow_factory/models/qwen_image/qwen_image.py:
// 288-438 loop, :476-600 forward, :579-587 CFG): packed latents (B, Ni, 64), timestep t/1000, text embeddings (B, Nt, 3584) with a
// per-sample valid length, then the same FlowMatchEulerDiscreteSDEScheduler.step.
//
// MI355X layout decisions:
//   * 20 B parameters = 41 GB bf16 stay RESIDENT in one arena (288 GB HBM): the rollout never gathers shards; an FSDP2-wrapped
//     trainable module is bound through DTensor.full_tensor() once per optimiser epoch (mi355_flow/binding.py).
//   * negative and positive prompts run as ONE forward batch [neg | pos] (2B samples) so every GEMM sees twice the rows; ragged text
//     lengths are handled by ordering the joint sequence [image | text] and masking keys past Ni + len[b] in the attention kernel.
//   * the 121 AdaLN modulation linears (6.8 B parameters, a third of the model) are one concatenated GEMM evaluated for all N steps
//     before the loop; in a rollout every sample shares t, so that GEMM has N rows and the modulation row stride is 0.
//   * txt_norm + txt_in are step-invariant and hoisted out of the loop.
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

struct QSlot { void* dst; int dst_dt; int64_t numel; bool bound; };

struct QBlockW {
    bf16_t *w_qk, *w_v, *w_o, *w_cqk, *w_cv, *w_co, *w_ff1, *w_ff2, *w_cff1, *w_cff2;
    float *b_qk, *b_v, *b_o, *b_cqk, *b_cv, *b_co, *b_ff1, *b_ff2, *b_cff1, *b_cff2;
    float *nq, *nk, *ncq, *nck;
    int mod_img, mod_ctx;
    float bound = 0.f;
};

// per-block activation stash of the training-mode forward (qwen_train.inc): what the backward of the block needs
struct QTrainBlk {
    bf16_t *x_in, *xn, *x_mid, *xn_mlp, *pre, *o_img;
    bf16_t *c_in, *cn, *c_mid, *cn_mlp, *cpre, *o_ctx;
    bf16_t *q, *k, *vT;
    float *lse, *rstd_img, *rstd_ctx;
};

float q_host_round(float v, int dt) {
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

struct mi355_qwen {
    mi355_qwen_cfg cfg;
    int D, F, L, H;
    int mod_cols, mod_out;
    char* arena16 = nullptr;
    char* arena32 = nullptr;
    size_t used16 = 0, used32 = 0;
    bf16_t *w_x, *w_ctx, *w_t1, *w_t2, *w_mod, *w_proj;
    float *b_x, *b_ctx, *b_t1, *b_t2, *b_mod, *b_proj, *txt_norm;
    std::vector<QBlockW> blk;
    std::map<std::string, QSlot> slots;
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
        slots[name] = QSlot{dst, dt, numel, false};
        names.push_back(name);
    }
    void lin(const std::string& name, bf16_t* w, float* b, int out_f, int in_f) {
        reg(name + ".weight", w, DT_BF16, (int64_t)out_f * in_f);
        reg(name + ".bias", b, DT_F32, out_f);
    }
    void layout();
};

// parameter names = diffusers QwenImageTransformer2DModel.state_dict()
void mi355_qwen::layout() {
    used16 = used32 = 0;
    slots.clear(); names.clear();
    blk.assign(L, QBlockW());
    const int T = cfg.time_proj_dim, J = cfg.joint_attention_dim, C = cfg.in_channels;
    const int64_t DD = (int64_t)D * D;
    w_x = a16((int64_t)D * C); b_x = a32(D); lin("img_in", w_x, b_x, D, C);
    txt_norm = a32(J); reg("txt_norm.weight", txt_norm, DT_F32, J);
    w_ctx = a16((int64_t)D * J); b_ctx = a32(D); lin("txt_in", w_ctx, b_ctx, D, J);
    w_t1 = a16((int64_t)D * T); b_t1 = a32(D); lin("time_text_embed.timestep_embedder.linear_1", w_t1, b_t1, D, T);
    w_t2 = a16(DD); b_t2 = a32(D); lin("time_text_embed.timestep_embedder.linear_2", w_t2, b_t2, D, D);
    int cols = 0;
    for (int i = 0; i < L; ++i) { blk[i].mod_img = cols; cols += 6 * D; blk[i].mod_ctx = cols; cols += 6 * D; }
    mod_out = cols; cols += 2 * D;
    mod_cols = cols;
    w_mod = a16((int64_t)mod_cols * D); b_mod = a32(mod_cols);
    for (int i = 0; i < L; ++i) {
        const std::string pre = "transformer_blocks." + std::to_string(i);
        lin(pre + ".img_mod.1", w_mod + (int64_t)blk[i].mod_img * D, b_mod + blk[i].mod_img, 6 * D, D);
        lin(pre + ".txt_mod.1", w_mod + (int64_t)blk[i].mod_ctx * D, b_mod + blk[i].mod_ctx, 6 * D, D);
    }
    lin("norm_out.linear", w_mod + (int64_t)mod_out * D, b_mod + mod_out, 2 * D, D);
    for (int i = 0; i < L; ++i) {
        QBlockW& b = blk[i];
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
        b.w_ff1 = a16((int64_t)F * D); b.b_ff1 = a32(F); lin(pre + ".img_mlp.net.0.proj", b.w_ff1, b.b_ff1, F, D);
        b.w_ff2 = a16((int64_t)D * F); b.b_ff2 = a32(D); lin(pre + ".img_mlp.net.2", b.w_ff2, b.b_ff2, D, F);
        b.w_cff1 = a16((int64_t)F * D); b.b_cff1 = a32(F); lin(pre + ".txt_mlp.net.0.proj", b.w_cff1, b.b_cff1, F, D);
        b.w_cff2 = a16((int64_t)D * F); b.b_cff2 = a32(D); lin(pre + ".txt_mlp.net.2", b.w_cff2, b.b_cff2, D, F);
    }
    w_proj = a16((int64_t)C * D); b_proj = a32(C); lin("proj_out", w_proj, b_proj, C, D);
}

// training-mode state (qwen_train.inc, included at the end of this file)
struct mi355_qwen_plan;
static void qwen_train_release(mi355_qwen_plan* p);
static void qwen_train_release_engine(mi355_qwen* e);
static void qwen_train_mark_dirty(mi355_qwen* e);

extern "C" int mi355_qwen_create(const mi355_qwen_cfg* cfg, mi355_qwen** out) {
    if (!cfg || !out) return errorf("mi355_qwen_create: null argument");
    if (cfg->head_dim != 128) return errorf("mi355_qwen_create: head_dim must be 128 (got %d)", cfg->head_dim);
    if (cfg->axes_dims_rope[0] + cfg->axes_dims_rope[1] + cfg->axes_dims_rope[2] != 128 ||
        (cfg->axes_dims_rope[0] | cfg->axes_dims_rope[1] | cfg->axes_dims_rope[2]) & 1)
        return errorf("mi355_qwen_create: axes_dims_rope must be even and sum to head_dim");
    if (cfg->num_layers < 1 || cfg->num_layers > 256) return errorf("mi355_qwen_create: layer count out of range");
    if (cfg->in_channels != 64) return errorf("mi355_qwen_create: in_channels must be 64 (2x2 patches of 16 latent channels)");
    if (cfg->joint_attention_dim % 64 || cfg->time_proj_dim % 64) return errorf("mi355_qwen_create: every GEMM K dim must be a multiple of 64");
    mi355_qwen* e = new mi355_qwen();
    e->cfg = *cfg;
    e->H = cfg->num_heads; e->D = cfg->num_heads * cfg->head_dim; e->F = 4 * e->D; e->L = cfg->num_layers;
    e->layout();
    const size_t cap16 = e->used16, cap32 = e->used32;
    hipError_t e1 = hipMalloc((void**)&e->arena16, cap16);
    hipError_t e2 = hipMalloc((void**)&e->arena32, cap32);
    if (e1 != hipSuccess || e2 != hipSuccess) {
        int r = errorf("mi355_qwen_create: hipMalloc of %zu + %zu bytes failed", cap16, cap32);
        if (e->arena16) (void)hipFree(e->arena16);
        if (e->arena32) (void)hipFree(e->arena32);
        delete e;
        return r;
    }
    e->layout();
    *out = e;
    return 0;
}

extern "C" int mi355_qwen_destroy(mi355_qwen* e) {
    if (!e) return 0;
    qwen_train_release_engine(e);
    if (e->arena16) (void)hipFree(e->arena16);
    if (e->arena32) (void)hipFree(e->arena32);
    delete e;
    return 0;
}
extern "C" int mi355_qwen_num_params(mi355_qwen* e) { return e ? (int)e->names.size() : 0; }
extern "C" const char* mi355_qwen_param_name(mi355_qwen* e, int i) {
    if (!e || i < 0 || i >= (int)e->names.size()) return nullptr;
    return e->names[i].c_str();
}
extern "C" int mi355_qwen_bind_weight(mi355_qwen* e, const char* name, const void* src, int dtype, int ndim, const int64_t* shape,
                                      void* stream) {
    if (!e || !name || !src) return errorf("mi355_qwen_bind_weight: null argument");
    auto it = e->slots.find(name);
    if (it == e->slots.end()) return errorf("mi355_qwen_bind_weight: unknown parameter '%s'", name);
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    if (n != it->second.numel)
        return errorf("mi355_qwen_bind_weight: '%s' has %lld elements, expected %lld", name, (long long)n, (long long)it->second.numel);
    if (dtype < 0 || dtype > 2) return errorf("mi355_qwen_bind_weight: bad dtype %d", dtype);
    HIPCHK(launch_convert(src, dtype, it->second.dst, it->second.dst_dt, n, (hipStream_t)stream));
    it->second.bound = true;
    qwen_train_mark_dirty(e);          // the transposed copies the backward's dgrad GEMMs read are stale
    if (strstr(name, ".norm_")) e->bounds_dirty = true;
    return 0;
}
extern "C" int mi355_qwen_weights_ready(mi355_qwen* e) {
    if (!e) return errorf("null engine");
    for (auto& kv : e->slots)
        if (!kv.second.bound) return errorf("parameter '%s' has not been bound", kv.first.c_str());
    return 0;
}

// -------------------------------------------------------------------------------------- plan
struct mi355_qwen_plan {
    mi355_qwen* e;
    int B, ncfg, FB, h, w, hp, wp, Ni, Nt, S, S_pad, Mi, Mc, M, max_steps;
    int64_t n_lat;      // packed elements per sample = Ni * in_channels
    long mod_ld = 0;    // modulation row stride of the current call: 0 in a rollout (all samples share t), mod_cols in a replay step
    char* ws = nullptr;
    size_t ws_bytes = 0;
    bf16_t *lat16, *x, *c, *c0, *xn, *cn, *qkbuf, *q, *k, *vT, *o_img, *o_ctx, *big, *v2, *v;
    bf16_t *tproj, *h1, *semb, *mod_all, *txtn;
    float2* cs;
    float *t_dev, *scal;
    int* kvlen;
    char *io_init, *io_traj;
    float *io_noise, *io_lp;
    bf16_t* io_pe;
    std::vector<float> host_t, host_sc;
    std::vector<int> host_len;
    // two-stream forward (opt-in, mi355_tune_set key 12): the text chain of every block on a plan-owned side stream with its own q|k
    // staging and MLP-hidden buffers (the single-stream path shares `qkbuf` / `big` between the two chains)
    hipStream_t side = nullptr;
    char* ws_side = nullptr;
    bf16_t *qkbuf_c = nullptr, *big_c = nullptr;
    std::vector<hipEvent_t> ev_join, ev_fork;   // per block: text q|k|v ready (side -> main), attention done (main -> side); [L] = forward start / end
    // hipGraph of the N-step loop (opt-in, mi355_tune_set key 17): captured on a plan-owned stream on the second call of a configuration
    hipGraphExec_t gexec = nullptr;
    hipStream_t cap_stream = nullptr;
    bool warmed = false;
    int g_steps = -1, g_dyn = -1, g_storage = -1, g_init = -1, g_clp = -1, g_noise = -1, g_bounds = -1, g_two = -1, g_gemm = -1, g_attn = -1, g_tune = -1;
    float g_sigma_max = 0.f, g_guidance = 0.f;
};

extern "C" int mi355_qwen_plan_create(mi355_qwen* e, int batch, int n_cfg, int latent_h, int latent_w, int n_text, int max_steps,
                                      mi355_qwen_plan** out) {
    if (!e || !out) return errorf("mi355_qwen_plan_create: null argument");
    if (batch < 1 || n_text < 1 || max_steps < 1 || latent_h < 2 || latent_w < 2 || (latent_h | latent_w) & 1 || n_cfg < 1 || n_cfg > 2)
        return errorf("mi355_qwen_plan_create: bad shape (latent size must be even, n_cfg 1 or 2)");
    mi355_qwen_plan* p = new mi355_qwen_plan();
    p->e = e; p->B = batch; p->ncfg = n_cfg; p->FB = batch * n_cfg;
    p->h = latent_h; p->w = latent_w; p->hp = latent_h / 2; p->wp = latent_w / 2;
    p->Ni = p->hp * p->wp; p->Nt = n_text; p->S = p->Ni + p->Nt; p->S_pad = (p->S + 63) / 64 * 64;
    p->Mi = p->FB * p->Ni; p->Mc = p->FB * p->Nt; p->M = p->FB * p->S; p->max_steps = max_steps;
    p->n_lat = (int64_t)p->Ni * e->cfg.in_channels;
    const int D = e->D, F = e->F, J = e->cfg.joint_attention_dim, FB = p->FB;
    const int64_t rows_cond = (int64_t)(max_steps > FB ? max_steps : FB);
    size_t off = 0;
    auto take = [&](int64_t elems, int esz) {
        size_t o = off;
        off += (((size_t)elems * esz) + 255) & ~(size_t)255;
        return o;
    };
    const int64_t qk_el = (int64_t)FB * e->H * p->S_pad * 128;
    const int64_t nl = (int64_t)batch * p->n_lat;
    const int64_t big_rows = p->Mi > p->Mc ? p->Mi : p->Mc;
    size_t o_l16 = take(nl, 2), o_x = take((int64_t)p->Mi * D, 2), o_c = take((int64_t)p->Mc * D, 2), o_c0 = take((int64_t)p->Mc * D, 2);
    size_t o_xn = take((int64_t)p->Mi * D, 2), o_cn = take((int64_t)p->Mc * D, 2), o_qkb = take(big_rows * 2 * D, 2);
    size_t o_q = take(qk_el, 2), o_k = take(qk_el, 2), o_vT = take(qk_el, 2);
    size_t o_oi = take((int64_t)p->Mi * D, 2), o_oc = take((int64_t)p->Mc * D, 2), o_big = take(big_rows * F, 2);
    size_t o_v2 = take((int64_t)FB * p->n_lat, 2), o_v = take(nl, 2);
    size_t o_tp = take(rows_cond * e->cfg.time_proj_dim, 2), o_h1 = take(rows_cond * D, 2), o_semb = take(rows_cond * D, 2);
    size_t o_mod = take(rows_cond * e->mod_cols, 2), o_txtn = take((int64_t)p->Mc * J, 2);
    size_t o_cs = take((int64_t)p->S * 64, 8);
    size_t o_t = take(rows_cond, 4), o_sc = take(3 * (int64_t)max_steps, 4), o_kl = take(FB, 4);
    size_t o_ii = take(nl, 4), o_it = take((int64_t)(max_steps + 1) * nl, 4), o_in = take((int64_t)max_steps * nl, 4);
    size_t o_il = take((int64_t)max_steps * batch, 4);
    size_t o_ipe = take((int64_t)p->Mc * J, 2);
    p->ws_bytes = off;
    if (hipMalloc((void**)&p->ws, off) != hipSuccess) {
        int r = errorf("mi355_qwen_plan_create: hipMalloc of %zu bytes failed", off);
        delete p;
        return r;
    }
    if (hipMemset(p->ws, 0, off) != hipSuccess) {   // padded key rows / columns of q, k, vT must stay finite
        (void)hipFree(p->ws);
        delete p;
        return errorf("mi355_qwen_plan_create: hipMemset failed");
    }
    char* w = p->ws;
    p->lat16 = (bf16_t*)(w + o_l16); p->x = (bf16_t*)(w + o_x); p->c = (bf16_t*)(w + o_c); p->c0 = (bf16_t*)(w + o_c0);
    p->xn = (bf16_t*)(w + o_xn); p->cn = (bf16_t*)(w + o_cn); p->qkbuf = (bf16_t*)(w + o_qkb);
    p->q = (bf16_t*)(w + o_q); p->k = (bf16_t*)(w + o_k); p->vT = (bf16_t*)(w + o_vT);
    p->o_img = (bf16_t*)(w + o_oi); p->o_ctx = (bf16_t*)(w + o_oc); p->big = (bf16_t*)(w + o_big);
    p->v2 = (bf16_t*)(w + o_v2); p->v = (bf16_t*)(w + o_v);
    p->tproj = (bf16_t*)(w + o_tp); p->h1 = (bf16_t*)(w + o_h1); p->semb = (bf16_t*)(w + o_semb); p->mod_all = (bf16_t*)(w + o_mod);
    p->txtn = (bf16_t*)(w + o_txtn);
    p->cs = (float2*)(w + o_cs); p->t_dev = (float*)(w + o_t); p->scal = (float*)(w + o_sc); p->kvlen = (int*)(w + o_kl);
    p->io_init = w + o_ii; p->io_traj = w + o_it; p->io_noise = (float*)(w + o_in); p->io_lp = (float*)(w + o_il);
    p->io_pe = (bf16_t*)(w + o_ipe);
    // rotary table of the joint sequence in THIS engine's order [img | txt] (attention is order-invariant once the angles are applied).
    // QwenEmbedRope (diffusers transformer_qwenimage.py): fp32 angles pos * theta^(-2j/dim); image token (frame 0, row r, col c) sits at
    // (0, r - (hp - hp/2), c - (wp - wp/2)) when scale_rope (rows/cols centred: negative indices first), else (0, r, c); text token j at
    // (m + j, m + j, m + j) with m = max(hp/2, wp/2) (scale_rope) or max(hp, wp).
    {
        std::vector<float> cs((size_t)p->S * 128);
        const int* ax = e->cfg.axes_dims_rope;
        const int sr = e->cfg.scale_rope;
        const int m0 = sr ? (p->hp / 2 > p->wp / 2 ? p->hp / 2 : p->wp / 2) : (p->hp > p->wp ? p->hp : p->wp);
        for (int s = 0; s < p->S; ++s) {
            float pos[3];
            if (s < p->Ni) {
                const int r = s / p->wp, cc = s % p->wp;
                pos[0] = 0.f;
                pos[1] = (float)(sr ? r - (p->hp - p->hp / 2) : r);
                pos[2] = (float)(sr ? cc - (p->wp - p->wp / 2) : cc);
            } else {
                pos[0] = pos[1] = pos[2] = (float)(m0 + (s - p->Ni));
            }
            int pair = 0;
            for (int a = 0; a < 3; ++a)
                for (int j = 0; j < ax[a] / 2; ++j, ++pair) {
                    const float inv = 1.0f / powf(10000.0f, (float)(2 * j) / (float)ax[a]);
                    const float ang = pos[a] * inv;
                    cs[((size_t)s * 64 + pair) * 2 + 0] = cosf(ang);
                    cs[((size_t)s * 64 + pair) * 2 + 1] = sinf(ang);
                }
        }
        if (hipMemcpy(p->cs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(p->ws);
            delete p;
            return errorf("mi355_qwen_plan_create: rotary table upload failed");
        }
    }
    *out = p;
    return 0;
}

extern "C" int mi355_qwen_plan_destroy(mi355_qwen_plan* p) {
    if (!p) return 0;
    qwen_train_release(p);
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
extern "C" int64_t mi355_qwen_plan_workspace_bytes(mi355_qwen_plan* p) { return p ? (int64_t)p->ws_bytes : 0; }

// ---------------------------------------------------------------------------------- forward
namespace {

// |score| <= 128 / sqrt(128) * log2(e) * max|w_q| * max|w_k| (RoPE preserves norms; see engine.hip update_score_bounds)
int update_score_bounds(mi355_qwen* e, hipStream_t st) {
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
    for (auto& b : e->blk) b.bound = c * fmaxf(amax(b.nq), amax(b.ncq)) * fmaxf(amax(b.nk), amax(b.nck));
    e->bounds_dirty = false;
    ++e->bounds_ver;
    return 0;
}

// step-invariant work: per-sample valid lengths, txt_norm (RMSNorm over 3584) and txt_in
int prepare_prompt(mi355_qwen_plan* p, hipStream_t st, const void* enc, const int32_t* txt_lens_host) {
    mi355_qwen* e = p->e;
    const int D = e->D, J = e->cfg.joint_attention_dim;
    p->host_len.assign(p->FB, p->Ni + p->Nt);
    if (txt_lens_host)
        for (int i = 0; i < p->FB; ++i) {
            if (txt_lens_host[i] < 1 || txt_lens_host[i] > p->Nt)
                return errorf("mi355_qwen: txt_lens[%d] = %d is outside [1, n_text = %d]", i, txt_lens_host[i], p->Nt);
            p->host_len[i] = p->Ni + txt_lens_host[i];
        }
    HIPCHK(hipMemcpyAsync(p->kvlen, p->host_len.data(), (size_t)p->FB * 4, hipMemcpyHostToDevice, st));
    HIPCHK(launch_rms_rows((const bf16_t*)enc, J, e->txt_norm, p->txtn, J, p->Mc, J, e->cfg.eps, st));
    GemmParams g = make_gemm(p->txtn, J, e->w_ctx, J, p->Mc, D, J, EPI_BIAS, e->b_ctx, p->c0, D);
    HIPCHK(launch_gemm(g, st));
    return 0;
}

// conditioning of `rows` (t values in t_dev) at once: temb = timestep_embedder(time_proj(t)), semb = silu(temb), mod_all = every AdaLN linear
int prepare_conditioning(mi355_qwen_plan* p, hipStream_t st, int rows) {
    mi355_qwen* e = p->e;
    const int D = e->D, T = e->cfg.time_proj_dim;
    HIPCHK(launch_time_proj(p->t_dev, rows, T, DT_F32, p->tproj, st));
    GemmParams g1 = make_gemm(p->tproj, T, e->w_t1, T, rows, D, T, EPI_BIAS_SILU, e->b_t1, p->h1, D);
    HIPCHK(launch_gemm(g1, st));
    GemmParams g2 = make_gemm(p->h1, D, e->w_t2, D, rows, D, D, EPI_BIAS_SILU, e->b_t2, p->semb, D);
    HIPCHK(launch_gemm(g2, st));
    GemmParams g3 = make_gemm(p->semb, D, e->w_mod, D, rows, e->mod_cols, D, EPI_BIAS, e->b_mod, p->mod_all, e->mod_cols);
    HIPCHK(launch_gemm(g3, st));
    return 0;
}

int ln_mod(mi355_qwen_plan* p, hipStream_t st, const bf16_t* x, bf16_t* out, const bf16_t* mod, int M, int rps, int shift_off, int scale_off) {
    LnModParams l;
    memset(&l, 0, sizeof(l));
    l.x = x; l.out = out; l.out2 = nullptr; l.mod = mod; l.mod_ld = p->mod_ld;
    l.shift_off = shift_off; l.scale_off = scale_off;
    l.M = M; l.D = p->e->D; l.rows_per_sample = rps; l.eps = p->e->cfg.eps;
    HIPCHK(launch_ln_mod(l, st));
    return 0;
}

// q|k projection (one GEMM) -> per-head RMSNorm + RoPE + scatter; V^T projection with the scatter fused (operands swapped)
int qkv(mi355_qwen_plan* p, hipStream_t st, const bf16_t* xin, int M, int rps, int s_off, const bf16_t* w_qk, const float* b_qk,
        const bf16_t* w_v, const float* b_v, const float* nq, const float* nk, bf16_t* qkb, const QTrainBlk* tb = nullptr, float* rstd = nullptr) {
    mi355_qwen* e = p->e;
    const int D = e->D;
    GemmParams g = make_gemm(xin, D, w_qk, D, M, 2 * D, D, EPI_BIAS, b_qk, qkb, 2 * D);
    HIPCHK(launch_gemm(g, st));
    RopeNormParams r;
    memset(&r, 0, sizeof(r));
    r.src = qkb; r.src_ld = 2 * D; r.q_col = 0; r.k_col = D; r.nw_q = nq; r.nw_k = nk; r.cs = p->cs;
    r.q_out = tb ? tb->q : p->q; r.k_out = tb ? tb->k : p->k; r.M = M; r.H = e->H; r.rows_per_sample = rps; r.s_off = s_off; r.S_pad = p->S_pad;
    r.eps = e->cfg.eps; r.q_scale = 0.08838834764831845f * 1.4426950408889634f;
    r.rstd_out = rstd;                                 // (training mode: 1 / rms per (token, head) of q and k for the producer's backward)
    HIPCHK(launch_rope_norm(r, st));
    GemmParams gv = make_gemm(w_v, D, xin, D, D, M, D, EPI_VT, b_v, nullptr, 0);
    gv.q = tb ? tb->vT : p->vT; gv.H = e->H; gv.S_pad = p->S_pad; gv.s_off = s_off; gv.rows_per_sample = rps; gv.hd_shift = 7;
    HIPCHK(launch_gemm(gv, st));
    return 0;
}

int gate_res(mi355_qwen_plan* p, hipStream_t st, const bf16_t* A, long lda, int K, const bf16_t* W, const float* bias, bf16_t* x,
             int M, int rps, const bf16_t* mod, int gate_off) {
    GemmParams g = make_gemm(A, lda, W, K, M, p->e->D, K, EPI_GATE_RES, bias, x, p->e->D);
    g.aux = mod + gate_off; g.ld_aux = p->mod_ld; g.rows_per_sample = rps;
    HIPCHK(launch_gemm(g, st));
    return 0;
}

// one transformer forward over the FB = n_cfg * B samples: packed latents (storage dtype, B samples, replicated per CFG branch) ->
// packed velocity v2 [FB][Ni][C] bf16.  `mod` = this call's first modulation row; c0 / kvlen prepared.
// Two-stream forward (tune key 12; ON by default for plans of up to 16 384 image rows since round 3): the text
// chain of a block (LN-modulate -> q|k|v projections ... out-projection -> LN-modulate -> MLP, M = FB * Nt rows: a fraction of a wave of
// workgroups) runs on a plan-owned side stream beside the image chain, as in the SD3.5 engine (engine.hip, keys 8-10: +6 ... +31 % on
// small forward batches).  Join before the joint attention (it reads the text rows of q / k / vT), fork after it (the text out-projection
// reads o_ctx; the NEXT block's text projections overwrite the text rows of q / k / vT, so they must also come after this attention).
// Results are bit-identical to the single-stream order: the kernels and their inputs are the same, only `qkbuf` / `big` are not shared.
// Measured on MI355X (profiles/r03a_qwen_two_stream_ab.txt, full 60-layer geometry, true CFG, s per 2-step rollout single -> two streams):
// 512^2 B = 1 0.1191 -> 0.0785 (+52 %), 384^2 B = 1 0.1015 -> 0.0705 (+44 %), 1024^2 B = 2 0.5243 -> 0.5083 (+3 %).
int g_qwen_two_stream = 2;          // 0 off, 1 on, 2 (default) on for plans with at most g_qwen_two_stream_rows image rows
int g_qwen_two_stream_rows = 16384;

// key 17: replay the N-step loop of mi355_qwen_rollout as ONE hipGraph (ON by default since round 3: bit-identical, +0.9 % at 512^2 and +2 % at
// 384^2 on top of the two-stream forward, +-0.1 % at 1024^2; same file).  The prompt preparation (it uploads the per-sample key lengths from
// host memory) stays in front of the graph.
int g_qwen_graph = 1;

bool qwen_two_stream_wanted(const mi355_qwen_plan* p) {
    return g_qwen_two_stream == 1 || (g_qwen_two_stream == 2 && p->Mi <= g_qwen_two_stream_rows);
}

int qwen_two_stream_init(mi355_qwen_plan* p) {
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

// Training mode (`tb` = the plan's per-block stash, qwen_train.inc): the SAME launches with the activations the backward needs written to per-block
// buffers instead of the plan's shared ones (+ the block inputs copied aside, the log-sum-exp / 1/rms side outputs switched on, the MLP
// pre-activations stashed by the GEMM epilogue): the same kernel binaries on the same values -- the prediction is bit-identical.
int forward_core(mi355_qwen_plan* p, hipStream_t st, const void* latents, int lat_dt, const bf16_t* mod, const QTrainBlk* tb = nullptr,
                 bf16_t* x_final = nullptr) {
    mi355_qwen* e = p->e;
    const int D = e->D, F = e->F, C = e->cfg.in_channels;
    const int Ni = p->Ni, Nt = p->Nt;
    const size_t xi_b = (size_t)p->Mi * D * 2, xc_b = (size_t)p->Mc * D * 2;
    const bool two = qwen_two_stream_wanted(p);
    hipStream_t ts = st;                               // carries the text chain
    bf16_t *qkb_c = p->qkbuf, *big_c = p->big;
    if (two) {
        CHK(qwen_two_stream_init(p));
        ts = p->side; qkb_c = p->qkbuf_c; big_c = p->big_c;
    }
    const bf16_t* lat = (const bf16_t*)latents;
    if (lat_dt != DT_BF16) {
        HIPCHK(launch_convert(latents, lat_dt, p->lat16, DT_BF16, (long)p->B * p->n_lat, st));
        lat = p->lat16;
    }
    // img_in on the B distinct samples, replicated for the second CFG branch
    GemmParams gx = make_gemm(lat, C, e->w_x, C, p->B * Ni, D, C, EPI_BIAS, e->b_x, p->x, D);
    HIPCHK(launch_gemm(gx, st));
    if (p->ncfg == 2)
        HIPCHK(copy_d2d(p->x + (size_t)p->B * Ni * D, p->x, (size_t)p->B * Ni * D * 2, st));
    HIPCHK(copy_d2d(p->c, p->c0, xc_b, st));
    if (two) {          // c, the conditioning (modulation table, prompt, key lengths) and the previous forward are complete on `st`
        HIPCHK(ev_record(p->ev_fork[e->L], st));
        HIPCHK(ev_wait(ts, p->ev_fork[e->L]));
    }
    for (int i = 0; i < e->L; ++i) {
        const QBlockW& b = e->blk[i];
        const QTrainBlk* k = tb ? tb + i : nullptr;
        const int mi = b.mod_img, mc = b.mod_ctx;     // chunks: shift1, scale1, gate1, shift2, scale2, gate2
        bf16_t* cn = k ? k->cn : p->cn;
        bf16_t* xn = k ? k->xn : p->xn;
        if (k) {
            HIPCHK(copy_d2d(k->c_in, p->c, xc_b, ts));
            HIPCHK(copy_d2d(k->x_in, p->x, xi_b, st));
        }
        CHK(ln_mod(p, ts, p->c, cn, mod, p->Mc, Nt, mc, mc + D));
        CHK(qkv(p, ts, cn, p->Mc, Nt, Ni, b.w_cqk, b.b_cqk, b.w_cv, b.b_cv, b.ncq, b.nck, qkb_c, k, k ? k->rstd_ctx : nullptr));
        CHK(ln_mod(p, st, p->x, xn, mod, p->Mi, Ni, mi, mi + D));
        CHK(qkv(p, st, xn, p->Mi, Ni, 0, b.w_qk, b.b_qk, b.w_v, b.b_v, b.nq, b.nk, p->qkbuf, k, k ? k->rstd_img : nullptr));
        if (two) {      // join: the attention reads the text rows of q / k / vT
            HIPCHK(ev_record(p->ev_join[i], ts));
            HIPCHK(ev_wait(st, p->ev_join[i]));
        }
        bf16_t* o_img = k ? k->o_img : p->o_img;
        bf16_t* o_ctx = k ? k->o_ctx : p->o_ctx;
        Attn128Params a;
        memset(&a, 0, sizeof(a));
        a.q = k ? k->q : p->q; a.k = k ? k->k : p->k; a.vT = k ? k->vT : p->vT; a.o_first = o_img; a.ld_first = D; a.n_first = Ni;
        a.o_rest = o_ctx; a.ld_rest = D; a.B = p->FB; a.H = e->H; a.S = p->S; a.S_pad = p->S_pad; a.q_prescaled = 1;
        a.score_bound = b.bound; a.kv_len = p->kvlen; a.lse = k ? k->lse : nullptr;
        HIPCHK(launch_attention128(a, st));
        if (two) {      // fork: o_ctx is written, and the text rows of q / k / vT are free for the next block's text projections
            HIPCHK(ev_record(p->ev_fork[i], st));
            HIPCHK(ev_wait(ts, p->ev_fork[i]));
        }
        CHK(gate_res(p, st, o_img, D, D, b.w_o, b.b_o, p->x, p->Mi, Ni, mod, mi + 2 * D));
        CHK(gate_res(p, ts, o_ctx, D, D, b.w_co, b.b_co, p->c, p->Mc, Nt, mod, mc + 2 * D));
        bf16_t* xn2 = k ? k->xn_mlp : p->xn;
        if (k) HIPCHK(copy_d2d(k->x_mid, p->x, xi_b, st));
        CHK(ln_mod(p, st, p->x, xn2, mod, p->Mi, Ni, mi + 3 * D, mi + 4 * D));
        GemmParams f1 = make_gemm(xn2, D, b.w_ff1, D, p->Mi, F, D, EPI_BIAS_GELU, b.b_ff1, p->big, F);
        if (k) { f1.stash = k->pre; f1.ld_stash = F; }
        HIPCHK(launch_gemm(f1, st));
        CHK(gate_res(p, st, p->big, F, F, b.w_ff2, b.b_ff2, p->x, p->Mi, Ni, mod, mi + 5 * D));
        if (i + 1 < e->L) {       // the text stream of the last block feeds nothing
            bf16_t* cn2 = k ? k->cn_mlp : p->cn;
            if (k) HIPCHK(copy_d2d(k->c_mid, p->c, xc_b, ts));
            CHK(ln_mod(p, ts, p->c, cn2, mod, p->Mc, Nt, mc + 3 * D, mc + 4 * D));
            GemmParams c1 = make_gemm(cn2, D, b.w_cff1, D, p->Mc, F, D, EPI_BIAS_GELU, b.b_cff1, big_c, F);
            if (k) { c1.stash = k->cpre; c1.ld_stash = F; }
            HIPCHK(launch_gemm(c1, ts));
            CHK(gate_res(p, ts, big_c, F, F, b.w_cff2, b.b_cff2, p->c, p->Mc, Nt, mod, mc + 5 * D));
        }
    }
    if (two) {          // the side stream's tail (last block's text out-projection) completes before `st` goes on: the next forward's
                        // conditioning and its copy into `c` are ordered behind it
        HIPCHK(ev_record(p->ev_join[e->L], ts));
        HIPCHK(ev_wait(st, p->ev_join[e->L]));
    }
    if (x_final) HIPCHK(copy_d2d(x_final, p->x, xi_b, st));
    // AdaLayerNormContinuous (scale first), proj_out
    CHK(ln_mod(p, st, p->x, p->xn, mod, p->Mi, Ni, e->mod_out + D, e->mod_out));
    GemmParams go = make_gemm(p->xn, D, e->w_proj, D, p->Mi, C, D, EPI_BIAS, e->b_proj, p->v2, C);
    HIPCHK(launch_gemm(go, st));
    return 0;
}

// v2 = [neg | pos] -> the prediction the scheduler sees (qwen_image.py:579-587); n_cfg == 1: the network output itself
const bf16_t* combine(mi355_qwen_plan* p, hipStream_t st, float guidance_scale, bf16_t* dst, int* rc) {
    *rc = 0;
    if (p->ncfg == 1) return p->v2;
    hipError_t e = launch_cfg_rescale(p->v2, p->v2 + (int64_t)p->B * p->n_lat, guidance_scale, dst, (long)p->B * p->Ni, p->e->cfg.in_channels, st);
    if (e != hipSuccess) *rc = errorf("launch_cfg_rescale failed: %s", hipGetErrorString(e));
    return dst;
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
void set_qwen_two_stream(int mode) { g_qwen_two_stream = mode; }
void set_qwen_two_stream_rows(int rows) { g_qwen_two_stream_rows = rows; }
void set_qwen_graph(int on) { g_qwen_graph = on; }
}  // namespace mi355

// One transformer evaluation incl. the CFG combine (replay / tests).  t_model [B] device fp32 = the angle base of the sinusoidal
// projection = 1000 * (the value the network receives, t / 1000 rounded to the latents' dtype): Timesteps(scale=1000).  prompt_embeds bf16 [n_cfg*B][n_text][J]
// (negative prompts first when n_cfg == 2, zero-padded to n_text), txt_lens_host int32 [n_cfg*B] valid lengths (NULL = n_text each).
// v_out bf16 [B][Ni][C] = the prediction handed to the scheduler; v_raw (optional) bf16 [n_cfg*B][Ni][C] = the raw network outputs.
extern "C" int mi355_qwen_forward(mi355_qwen_plan* p, void* stream, const void* latents, int lat_dtype, const float* t_model,
                                  const void* prompt_embeds, const int32_t* txt_lens_host, float guidance_scale, void* v_out, void* v_raw) {
    if (!p || !latents || !t_model || !prompt_embeds || !v_out) return errorf("mi355_qwen_forward: null argument");
    if (lat_dtype < 0 || lat_dtype > 2) return errorf("mi355_qwen_forward: bad latent dtype %d", lat_dtype);
    CHK(mi355_qwen_weights_ready(p->e));
    hipStream_t st = (hipStream_t)stream;
    CHK(update_score_bounds(p->e, st));
    // per-sample t: one modulation row per forward sample ([neg | pos] share the B values)
    for (int r = 0; r < p->ncfg; ++r)
        HIPCHK(copy_d2d(p->t_dev + (size_t)r * p->B, t_model, (size_t)p->B * 4, st));
    CHK(prepare_prompt(p, st, prompt_embeds, txt_lens_host));
    p->mod_ld = p->e->mod_cols;
    CHK(prepare_conditioning(p, st, p->FB));
    CHK(forward_core(p, st, latents, lat_dtype, p->mod_all));
    int rc = 0;
    const bf16_t* v = combine(p, st, guidance_scale, p->v, &rc);
    CHK(rc);
    HIPCHK(copy_d2d(v_out, v, (size_t)p->B * p->n_lat * 2, st));
    if (v_raw) HIPCHK(copy_d2d(v_raw, p->v2, (size_t)p->FB * p->n_lat * 2, st));
    return 0;
}

// The whole N-step rollout (qwen_image.py:372-423) with zero host syncs.  timesteps_host: scheduler timesteps in [0, 1000];
// sigmas_host: scheduler sigmas (sigma_max = sigmas[1]); noise_levels_host: eta per step; guidance_scale: true-CFG scale (used when the
// plan has n_cfg == 2).  Latents are PACKED [B][Ni][in_channels]; step_noise fp32 [n_steps][B][Ni*in_channels].
extern "C" int mi355_qwen_rollout(mi355_qwen_plan* p, void* stream, int n_steps, const float* timesteps_host, const float* sigmas_host,
                                  const float* noise_levels_host, int dynamics, float guidance_scale, const void* init_latents,
                                  int init_dtype, int storage_dtype, const float* step_noise, const void* prompt_embeds,
                                  const int32_t* txt_lens_host, const int32_t* keep_slot_host, void* out_latents, float* out_log_probs,
                                  void* out_final, int compute_log_prob) {
    if (!p || !timesteps_host || !sigmas_host || !noise_levels_host || !init_latents || !prompt_embeds)
        return errorf("mi355_qwen_rollout: null argument");
    if (n_steps < 1 || n_steps > p->max_steps) return errorf("mi355_qwen_rollout: n_steps %d exceeds the plan's max_steps %d", n_steps, p->max_steps);
    if (storage_dtype < 0 || storage_dtype > 2 || init_dtype < 0 || init_dtype > 2) return errorf("mi355_qwen_rollout: bad dtype");
    if (!step_noise && dynamics != MI355_ODE) return errorf("mi355_qwen_rollout: step_noise is NULL");
    if (dynamics < 0 || dynamics > 3) return errorf("mi355_qwen_rollout: unknown dynamics %d", dynamics);
    CHK(mi355_qwen_weights_ready(p->e));
    hipStream_t st = (hipStream_t)stream;
    CHK(update_score_bounds(p->e, st));
    const int B = p->B;
    std::vector<float>& tt = p->host_t;
    std::vector<float>& sc = p->host_sc;
    tt.assign((size_t)n_steps, 0.f);
    sc.assign(3 * (size_t)p->max_steps, 0.f);
    for (int i = 0; i < n_steps; ++i) {
        // qwen_image.py:497, 534: timestep = t.to(latents.dtype); the model receives timestep / 1000 (same dtype) and its sinusoidal
        // projection multiplies by 1000 in fp32 (Timesteps(scale=1000)): the angle base is (t_model * 1000)
        const float tm = q_host_round(q_host_round(timesteps_host[i], storage_dtype) / 1000.0f, storage_dtype);
        tt[i] = tm * 1000.0f;
        const float t_next = (i + 1 < n_steps) ? timesteps_host[i + 1] : 0.0f;
        sc[i] = timesteps_host[i] / 1000.0f;
        sc[p->max_steps + i] = t_next / 1000.0f;
        sc[2 * p->max_steps + i] = noise_levels_host[i];
    }
    HIPCHK(hipMemcpyAsync(p->t_dev, tt.data(), (size_t)n_steps * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(p->scal, sc.data(), sc.size() * 4, hipMemcpyHostToDevice, st));
    const int64_t nl = (int64_t)B * p->n_lat;
    const size_t in_esz = init_dtype == MI355_F32 ? 4 : 2;
    HIPCHK(copy_d2d(p->io_init, init_latents, nl * in_esz, st));
    if (step_noise) HIPCHK(copy_d2d(p->io_noise, step_noise, (size_t)n_steps * nl * 4, st));
    HIPCHK(copy_d2d(p->io_pe, prompt_embeds, (size_t)p->Mc * p->e->cfg.joint_attention_dim * 2, st));
    const float sigma_max = sigmas_host[1];
    const int clp = compute_log_prob && out_log_probs;
    CHK(prepare_prompt(p, st, p->io_pe, txt_lens_host));
    p->mod_ld = 0;                                    // every sample of a step shares its one modulation row
    const size_t esz = storage_dtype == MI355_F32 ? 4 : 2;
    const size_t lat_bytes = (size_t)nl * esz;
    // everything below reads / writes plan-owned buffers at fixed addresses (staged inputs, c0 / kvlen, t_dev / scal, io_traj, io_lp)
    auto body = [&](hipStream_t s) -> int {
        CHK(prepare_conditioning(p, s, n_steps));
        HIPCHK(launch_convert(p->io_init, init_dtype, p->io_traj, storage_dtype, (long)nl, s));      // cast_latents(init)
        for (int i = 0; i < n_steps; ++i) {
            const bf16_t* mod = p->mod_all + (int64_t)i * p->e->mod_cols;
            char* cur = p->io_traj + (size_t)i * lat_bytes;
            char* nxt = p->io_traj + (size_t)(i + 1) * lat_bytes;
            CHK(forward_core(p, s, cur, storage_dtype, mod));
            int rc = 0;
            const bf16_t* v = combine(p, s, guidance_scale, p->v, &rc);
            CHK(rc);
            CHK(sde_call(s, B, p->n_lat, v, cur, storage_dtype, step_noise ? p->io_noise + (int64_t)i * nl : nullptr, p->scal + i,
                         p->scal + p->max_steps + i, p->scal + 2 * p->max_steps + i, sigma_max, dynamics, clp ? 2 : 0, nxt,
                         clp ? p->io_lp + (int64_t)i * B : nullptr));
        }
        return 0;
    };
    bool launched = false;
    if (g_qwen_graph && p->warmed) {
        const int two = (int)qwen_two_stream_wanted(p);
        const bool same = p->gexec && p->g_steps == n_steps && p->g_dyn == dynamics && p->g_storage == storage_dtype && p->g_init == init_dtype &&
                          p->g_clp == clp && p->g_noise == (int)(step_noise != nullptr) && p->g_sigma_max == sigma_max && p->g_guidance == guidance_scale &&
                          p->g_bounds == p->e->bounds_ver && p->g_two == two && p->g_gemm == get_gemm_variant() && p->g_attn == get_attn128_variant() && p->g_tune == tune_epoch();
        if (!same) {
            if (two) CHK(qwen_two_stream_init(p));              // streams / events / buffers are created outside the capture
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
            if (!p->gexec) {                                    // no silent fallback: the caller chooses eager launches with key 17 = 0
                const hipError_t last = hipGetLastError();
                return errorf("mi355_qwen_rollout: hipGraph capture / instantiation of the %d-step loop failed (%s); mi355_tune_set(17, 0) "
                              "selects eager launches", n_steps, hipGetErrorString(ce != hipSuccess ? ce : last));
            }
            p->g_steps = n_steps; p->g_dyn = dynamics; p->g_storage = storage_dtype; p->g_init = init_dtype; p->g_clp = clp;
            p->g_noise = (int)(step_noise != nullptr); p->g_sigma_max = sigma_max; p->g_guidance = guidance_scale; p->g_bounds = p->e->bounds_ver;
            p->g_two = two; p->g_gemm = get_gemm_variant(); p->g_attn = get_attn128_variant(); p->g_tune = tune_epoch();
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

// ----------------------------------------------------------------------- operator-level API (tests)
extern "C" int mi355_op_cfg_rescale(void* stream, const void* v_neg, const void* v_pos, float guidance_scale, void* out, int64_t rows, int channels) {
    if (!v_neg || !v_pos || !out) return errorf("mi355_op_cfg_rescale: null argument");
    HIPCHK(launch_cfg_rescale((const bf16_t*)v_neg, (const bf16_t*)v_pos, guidance_scale, (bf16_t*)out, (long)rows, channels, (hipStream_t)stream));
    return 0;
}

extern "C" int mi355_op_rms_rows(void* stream, const void* x, const float* weight, void* out, int rows, int dim, float eps) {
    if (!x || !weight || !out) return errorf("mi355_op_rms_rows: null argument");
    HIPCHK(launch_rms_rows((const bf16_t*)x, dim, weight, (bf16_t*)out, dim, rows, dim, eps, (hipStream_t)stream));
    return 0;
}

#include "qwen_train.inc"
