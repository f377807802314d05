// mi355_flow -- engine / plan / rollout orchestration behind the C ABI (include/mi355_flow.h).
// Host code only launches kernels on the caller's stream; no device synchronisation anywhere
// on the rollout path (the reference syncs 3x per step: flow_match_euler_discrete.py:187-194,:344,
// models/abc.py:177-181).
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/mi355_flow.h"
#include "kernels.h"

using namespace mi355;

static thread_local std::string g_err;
static int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}
namespace mi355 {
int errorf(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}
}  // namespace mi355
#define HIPCHK(x)                                                                          \
    do {                                                                                   \
        hipError_t _e = (x);                                                               \
        if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define CHK(x)                 \
    do {                       \
        int _r = (x);          \
        if (_r) return _r;     \
    } while (0)

// ------------------------------------------------------------------------ kernel-class timing
// Optional hipEvent brackets around every launch of a kernel class, recorded on the launch stream
// (bench.py's roofline leg).  Off by default: zero cost on the normal path.
enum ProfClass : int { PC_ATTN = 0, PC_GEMM, PC_LNMOD, PC_SDE, PC_MISC, PC_COUNT };
struct ProfState {
    bool on = false;
    int mask = 0;                 // bit c set: bracket launches of class c
    std::vector<hipEvent_t> ev;   // pairs
    std::vector<int> cls;
    size_t used = 0;              // pairs used
    double ms[PC_COUNT] = {0};
    long count[PC_COUNT] = {0};
};
static ProfState g_prof;

struct ProfScope {
    hipStream_t st; size_t idx; bool active;
    ProfScope(int cls, hipStream_t s) : st(s), idx(0), active(g_prof.on && ((g_prof.mask >> cls) & 1)) {
        if (!active) return;
        if (g_prof.used * 2 + 2 > g_prof.ev.size()) {
            for (int i = 0; i < 2; ++i) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) { active = false; return; }
                g_prof.ev.push_back(e);
            }
            g_prof.cls.push_back(cls);
        }
        idx = g_prof.used++;
        g_prof.cls[idx] = cls;
        (void)ev_record(g_prof.ev[2 * idx], st);
    }
    ~ProfScope() {
        if (active) (void)ev_record(g_prof.ev[2 * idx + 1], st);
    }
};

// on: 0 = off; 1 = every kernel class; otherwise a bit mask of classes (bit 0 attention, 1 gemm, 2 ln_modulate, 3 sde_step)
// shifted left by one (e.g. 2 = attention only: ~4 k events per rollout instead of ~31 k).
extern "C" int mi355_profile_enable(int on) {
    g_prof.on = on != 0;
    g_prof.mask = on == 1 ? 0x1f : (on >> 1);
    g_prof.used = 0;
    for (int i = 0; i < PC_COUNT; ++i) { g_prof.ms[i] = 0; g_prof.count[i] = 0; }
    return 0;
}

// Waits for the recorded events, accumulates elapsed ms / launch counts per class
// (order: attention, gemm, ln_modulate, sde_step, misc) and resets the event pool.
extern "C" int mi355_profile_collect(double* ms_out, int64_t* count_out) {
    for (size_t i = 0; i < g_prof.used; ++i) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(g_prof.ev[2 * i + 1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
        if (e != hipSuccess) return fail("mi355_profile_collect: %s", hipGetErrorString(e));
        g_prof.ms[g_prof.cls[i]] += ms;
        g_prof.count[g_prof.cls[i]] += 1;
    }
    g_prof.used = 0;
    for (int i = 0; i < PC_COUNT; ++i) {
        if (ms_out) ms_out[i] = g_prof.ms[i];
        if (count_out) count_out[i] = g_prof.count[i];
    }
    return 0;
}

static hipError_t gemm_p(const GemmParams& g, hipStream_t st) { ProfScope ps(PC_GEMM, st); return launch_gemm(g, st); }
static hipError_t attn_p(const AttnParams& a, hipStream_t st) { ProfScope ps(PC_ATTN, st); return launch_attention(a, st); }
static hipError_t lnmod_p(const LnModParams& l, hipStream_t st) { ProfScope ps(PC_LNMOD, st); return launch_ln_mod(l, st); }
static hipError_t sde_p(const SdeStepParams& s, hipStream_t st) { ProfScope ps(PC_SDE, st); return launch_sde_step(s, st); }

// training-mode state (engine_train.inc, included at the end of this file)
static int g_train_two_stream = 1;   // key 22, see engine_train.inc
static int g_wgrad_tn = 1;           // key 39: weight gradients of whole-tile shapes on the row-major-operand kernels (gemm_tn.hip): no dY^T / X^T copies
                                     // (1 = 128 x 128 tiles; 2 = 256 x 256 tiles where they give a one-round grid, else 128 x 128: measured equal; 0 = transposed copies)
static int g_fuse_colsum = 1;        // key 38: bias-gradient column-sum finish inside the weight gradient's split-K reduction launch (engine_train.inc: wgrad)
struct mi355_engine;
struct mi355_plan;
static void train_release(mi355_plan* p);
static void train_release_engine(mi355_engine* e);
static void train_mark_dirty(mi355_engine* e);

// ------------------------------------------------------------------------------------ engine
struct Slot {
    void* dst;      // device destination (bf16_t* or float*)
    int dst_dt;     // DT_BF16 / DT_F32
    int64_t numel;  // elements expected
    bool bound;
};

struct BlockW {
    bool dual, last;
    bf16_t *w_qk, *w_v, *w_o, *w_cqk, *w_cv, *w_co, *w_qk2, *w_v2, *w_o2, *w_ff1, *w_ff2, *w_cff1, *w_cff2;
    float *b_qk, *b_v, *b_o, *b_cqk, *b_cv, *b_co, *b_qk2, *b_v2, *b_o2, *b_ff1, *b_ff2, *b_cff1, *b_cff2;
    float *nq, *nk, *ncq, *nck, *nq2, *nk2;
    int mod_img, mod_ctx;  // column offsets into a mod_all row
    float bound_joint = 0.f, bound_dual = 0.f;   // proven |score| bounds of the two attentions (update_score_bounds)
};

struct mi355_engine {
    mi355_model_cfg cfg;
    int D, F, L, KP;  // model dim, ff inner, layers, patch K
    int mod_cols, mod_out;
    char* arena16 = nullptr;
    char* arena32 = nullptr;
    size_t used16 = 0, used32 = 0, cap16 = 0, cap32 = 0;
    bf16_t *w_patch, *pos_embed, *w_t1, *w_t2, *w_p1, *w_p2, *w_ctx, *w_mod, *w_proj;
    float *b_patch, *b_t1, *b_t2, *b_p1, *b_p2, *b_ctx, *b_mod, *b_proj;
    std::vector<BlockW> blk;
    std::map<std::string, Slot> slots;
    std::vector<std::string> names;
    bool bounds_dirty = true;   // a norm weight was (re)bound since the last update_score_bounds
    int bounds_ver = 0;

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
        if (arena16) {
            slots[name] = Slot{dst, dt, numel, false};
            names.push_back(name);
        }
    }
    // linear: weight [out][in] bf16 at w (row offset allowed by caller), bias fp32
    void lin(const std::string& name, bf16_t* w, float* b, int out_f, int in_f) {
        reg(name + ".weight", w, DT_BF16, (int64_t)out_f * in_f);
        reg(name + ".bias", b, DT_F32, out_f);
    }
    void layout();  // called twice: sizing pass (arena null), then real pass
};

void mi355_engine::layout() {
    used16 = used32 = 0;
    slots.clear();
    names.clear();
    blk.assign(L, BlockW());
    const int P = cfg.pooled_projection_dim, J = cfg.joint_attention_dim, T = cfg.time_proj_dim;
    w_patch = a16((int64_t)D * KP); b_patch = a32(D);
    reg("pos_embed.proj.weight", w_patch, DT_BF16, (int64_t)D * KP);
    reg("pos_embed.proj.bias", b_patch, DT_F32, D);
    pos_embed = a16((int64_t)cfg.pos_embed_max_size * cfg.pos_embed_max_size * D);
    reg("pos_embed.pos_embed", pos_embed, DT_BF16, (int64_t)cfg.pos_embed_max_size * cfg.pos_embed_max_size * D);
    w_t1 = a16((int64_t)D * T); b_t1 = a32(D); lin("time_text_embed.timestep_embedder.linear_1", w_t1, b_t1, D, T);
    w_t2 = a16((int64_t)D * D); b_t2 = a32(D); lin("time_text_embed.timestep_embedder.linear_2", w_t2, b_t2, D, D);
    w_p1 = a16((int64_t)D * P); b_p1 = a32(D); lin("time_text_embed.text_embedder.linear_1", w_p1, b_p1, D, P);
    w_p2 = a16((int64_t)D * D); b_p2 = a32(D); lin("time_text_embed.text_embedder.linear_2", w_p2, b_p2, D, D);
    w_ctx = a16((int64_t)D * J); b_ctx = a32(D); lin("context_embedder", w_ctx, b_ctx, D, J);
    // all AdaLN modulation linears concatenated along the output dim: one GEMM for every block (K3)
    int cols = 0;
    for (int i = 0; i < L; ++i) {
        BlockW& b = blk[i];
        b.dual = (cfg.dual_layer_mask >> i) & 1;
        b.last = (i == L - 1);
        b.mod_img = cols; cols += (b.dual ? 9 : 6) * D;
        b.mod_ctx = cols; cols += (b.last ? 2 : 6) * D;
    }
    mod_out = cols; cols += 2 * D;
    mod_cols = cols;
    w_mod = a16((int64_t)mod_cols * D); b_mod = a32(mod_cols);
    for (int i = 0; i < L; ++i) {
        BlockW& b = blk[i];
        const std::string pre = "transformer_blocks." + std::to_string(i);
        lin(pre + ".norm1.linear", w_mod + (int64_t)b.mod_img * D, b_mod + b.mod_img, (b.dual ? 9 : 6) * D, D);
        lin(pre + ".norm1_context.linear", w_mod + (int64_t)b.mod_ctx * D, b_mod + b.mod_ctx, (b.last ? 2 : 6) * D, D);
    }
    lin("norm_out.linear", w_mod + (int64_t)mod_out * D, b_mod + mod_out, 2 * D, D);
    const int64_t DD = (int64_t)D * D;
    for (int i = 0; i < L; ++i) {
        BlockW& b = blk[i];
        const std::string pre = "transformer_blocks." + std::to_string(i);
        b.w_qk = a16(2 * DD); b.b_qk = a32(2 * D);
        lin(pre + ".attn.to_q", b.w_qk, b.b_qk, D, D);
        lin(pre + ".attn.to_k", b.w_qk + DD, b.b_qk + D, D, D);
        b.w_v = a16(DD); b.b_v = a32(D); lin(pre + ".attn.to_v", b.w_v, b.b_v, D, D);
        b.w_o = a16(DD); b.b_o = a32(D); lin(pre + ".attn.to_out.0", b.w_o, b.b_o, D, D);
        b.w_cqk = a16(2 * DD); b.b_cqk = a32(2 * D);
        lin(pre + ".attn.add_q_proj", b.w_cqk, b.b_cqk, D, D);
        lin(pre + ".attn.add_k_proj", b.w_cqk + DD, b.b_cqk + D, D, D);
        b.w_cv = a16(DD); b.b_cv = a32(D); lin(pre + ".attn.add_v_proj", b.w_cv, b.b_cv, D, D);
        if (!b.last) { b.w_co = a16(DD); b.b_co = a32(D); lin(pre + ".attn.to_add_out", b.w_co, b.b_co, D, D); }
        b.nq = a32(64); b.nk = a32(64); b.ncq = a32(64); b.nck = a32(64);
        reg(pre + ".attn.norm_q.weight", b.nq, DT_F32, 64);
        reg(pre + ".attn.norm_k.weight", b.nk, DT_F32, 64);
        reg(pre + ".attn.norm_added_q.weight", b.ncq, DT_F32, 64);
        reg(pre + ".attn.norm_added_k.weight", b.nck, DT_F32, 64);
        if (b.dual) {
            b.w_qk2 = a16(2 * DD); b.b_qk2 = a32(2 * D);
            lin(pre + ".attn2.to_q", b.w_qk2, b.b_qk2, D, D);
            lin(pre + ".attn2.to_k", b.w_qk2 + DD, b.b_qk2 + D, D, D);
            b.w_v2 = a16(DD); b.b_v2 = a32(D); lin(pre + ".attn2.to_v", b.w_v2, b.b_v2, D, D);
            b.w_o2 = a16(DD); b.b_o2 = a32(D); lin(pre + ".attn2.to_out.0", b.w_o2, b.b_o2, D, D);
            b.nq2 = a32(64); b.nk2 = a32(64);
            reg(pre + ".attn2.norm_q.weight", b.nq2, DT_F32, 64);
            reg(pre + ".attn2.norm_k.weight", b.nk2, DT_F32, 64);
        }
        b.w_ff1 = a16((int64_t)F * D); b.b_ff1 = a32(F); lin(pre + ".ff.net.0.proj", b.w_ff1, b.b_ff1, F, D);
        b.w_ff2 = a16((int64_t)D * F); b.b_ff2 = a32(D); lin(pre + ".ff.net.2", b.w_ff2, b.b_ff2, D, F);
        if (!b.last) {
            b.w_cff1 = a16((int64_t)F * D); b.b_cff1 = a32(F); lin(pre + ".ff_context.net.0.proj", b.w_cff1, b.b_cff1, F, D);
            b.w_cff2 = a16((int64_t)D * F); b.b_cff2 = a32(D); lin(pre + ".ff_context.net.2", b.w_cff2, b.b_cff2, D, F);
        }
    }
    const int NO = cfg.patch_size * cfg.patch_size * cfg.out_channels;
    w_proj = a16((int64_t)NO * D); b_proj = a32(NO); lin("proj_out", w_proj, b_proj, NO, D);
}

extern "C" int mi355_version(void) { return MI355_FLOW_VERSION; }
extern "C" const char* mi355_last_error(void) { return g_err.c_str(); }

extern "C" int mi355_engine_create(const mi355_model_cfg* cfg, mi355_engine** out) {
    if (!cfg || !out) return fail("mi355_engine_create: null argument");
    if (cfg->head_dim != 64) return fail("mi355_engine_create: head_dim must be 64 (got %d)", cfg->head_dim);
    if (cfg->num_layers < 1 || cfg->num_layers > 64) return fail("mi355_engine_create: num_layers out of range");
    const int KP = cfg->in_channels * cfg->patch_size * cfg->patch_size;
    if (KP % 64 || cfg->joint_attention_dim % 64 || cfg->pooled_projection_dim % 64 || cfg->time_proj_dim % 64)
        return fail("mi355_engine_create: every GEMM K dim must be a multiple of 64");
    if ((cfg->patch_size * cfg->patch_size * cfg->out_channels) % 4) return fail("proj_out width must be a multiple of 4");
    mi355_engine* e = new mi355_engine();
    e->cfg = *cfg;
    e->D = cfg->num_heads * cfg->head_dim;
    e->F = cfg->ff_mult * e->D;
    e->L = cfg->num_layers;
    e->KP = KP;
    e->layout();  // sizing pass
    e->cap16 = e->used16;
    e->cap32 = e->used32;
    hipError_t e1 = hipMalloc((void**)&e->arena16, e->cap16);
    hipError_t e2 = hipMalloc((void**)&e->arena32, e->cap32);
    if (e1 != hipSuccess || e2 != hipSuccess) {
        int r = fail("mi355_engine_create: hipMalloc of %zu + %zu bytes failed (%s)", e->cap16, e->cap32, hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        if (e->arena16) (void)hipFree(e->arena16);
        if (e->arena32) (void)hipFree(e->arena32);
        delete e;
        return r;
    }
    e->layout();  // real pass
    *out = e;
    return 0;
}

extern "C" int mi355_engine_destroy(mi355_engine* e) {
    if (!e) return 0;
    train_release_engine(e);
    if (e->arena16) (void)hipFree(e->arena16);
    if (e->arena32) (void)hipFree(e->arena32);
    delete e;
    return 0;
}

extern "C" int mi355_engine_num_params(mi355_engine* e) { return e ? (int)e->names.size() : 0; }
extern "C" const char* mi355_engine_param_name(mi355_engine* e, int i) {
    if (!e || i < 0 || i >= (int)e->names.size()) return nullptr;
    return e->names[i].c_str();
}

extern "C" int mi355_engine_bind_weight(mi355_engine* e, const char* name, const void* src, int dtype, int ndim,
                                        const int64_t* shape, void* stream) {
    if (!e || !name || !src) return fail("mi355_engine_bind_weight: null argument");
    auto it = e->slots.find(name);
    if (it == e->slots.end()) return fail("mi355_engine_bind_weight: unknown parameter '%s'", name);
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    if (n != it->second.numel)
        return fail("mi355_engine_bind_weight: '%s' has %lld elements, expected %lld", name, (long long)n,
                    (long long)it->second.numel);
    if (dtype < 0 || dtype > 2) return fail("mi355_engine_bind_weight: bad dtype %d", dtype);
    HIPCHK(launch_convert(src, dtype, it->second.dst, it->second.dst_dt, n, (hipStream_t)stream));
    it->second.bound = true;
    train_mark_dirty(e);
    if (strstr(name, ".norm_")) e->bounds_dirty = true;
    return 0;
}

extern "C" int mi355_engine_weights_ready(mi355_engine* e) {
    if (!e) return fail("null engine");
    for (auto& kv : e->slots)
        if (!kv.second.bound) return fail("parameter '%s' has not been bound", kv.first.c_str());
    return 0;
}

// -------------------------------------------------------------------------------------- plan
struct mi355_plan {
    mi355_engine* e;
    int B, ncfg, Bp, h, w, hp, wp, Ni, Nt, S, S_pad, Mi, Mc, C, max_steps;
    int64_t n_lat;  // elements per latent sample
    char* ws = nullptr;
    size_t ws_bytes = 0;
    bool pos_ready = false;
    // workspace pointers
    bf16_t *pe, *patches, *x, *c, *c0, *xn, *xn2, *cn, *q, *k, *vT, *q2, *k2, *vT2, *o_img, *o_ctx, *hid, *chid;
    bf16_t *tproj, *h1, *p1, *pemb, *semb, *mod_all, *v;
    float *t_dev, *scal;  // t per (step, sample); scalars [3][max_steps] (sigma, sigma_next, eta)
    // rollout I/O staging (fixed addresses: the captured hipGraph bakes pointers in)
    char *io_init, *io_traj;            // init latents (<= fp32), trajectory [max_steps+1][B][n_lat] storage dtype
    float *io_noise, *io_lp;            // step noise [max_steps][B][n_lat] fp32, log-probs [max_steps][B]
    bf16_t *io_pe, *io_pp, *io_ne, *io_np;
    std::vector<float> host_t, host_sc;
    // hipGraph of the whole N-step loop
    hipGraphExec_t gexec = nullptr;
    hipStream_t cap_stream = nullptr;   // capture happens on a plan-owned stream: the caller's may be the legacy default stream (torch's
                                        // current stream unless the user switched), which cannot be captured; the graph is LAUNCHED on the caller's
    bool warmed = false;
    int g_steps = -1, g_dyn = -1, g_storage = -1, g_init = -1, g_clp = -1, g_attn = -1, g_gemm = -1, g_bounds = -1, g_two = -1, g_tune = -1;
    // text-stream chain of a forward on a second stream (forward_core): plan-owned, created on first use
    hipStream_t side = nullptr, side_v = nullptr;   // side_v: the image stream's V^T (and dual-attention q|k / V^T) projections
    std::vector<hipEvent_t> ev_vfork, ev_vjoin, ev_djoin;   // per block: xn ready (main -> side_v), V^T ready, dual q|k|V^T ready (side_v -> main)
    std::vector<hipEvent_t> ev_join, ev_fork;   // per block: text q|k|v ready (side -> main), attention done (main -> side); [L] = forward start
    float g_guidance = 0.f, g_sigma_max = 0.f;
};

extern "C" int mi355_plan_create(mi355_engine* e, int batch, int n_cfg, int latent_h, int latent_w, int n_text,
                                 int max_steps, mi355_plan** out) {
    if (!e || !out) return fail("mi355_plan_create: null argument");
    if (batch < 1 || (n_cfg != 1 && n_cfg != 2) || n_text < 1 || max_steps < 1) return fail("mi355_plan_create: bad shape");
    const int ps = e->cfg.patch_size;
    if (latent_h % ps || latent_w % ps) return fail("latent size must be a multiple of patch_size");
    mi355_plan* p = new mi355_plan();
    p->e = e;
    p->B = batch; p->ncfg = n_cfg; p->Bp = batch * n_cfg;
    p->h = latent_h; p->w = latent_w; p->hp = latent_h / ps; p->wp = latent_w / ps;
    p->Ni = p->hp * p->wp; p->Nt = n_text; p->S = p->Ni + p->Nt;
    p->S_pad = (p->S + 63) / 64 * 64;
    p->Mi = p->Bp * p->Ni; p->Mc = p->Bp * p->Nt;
    p->C = e->cfg.in_channels; p->max_steps = max_steps;
    p->n_lat = (int64_t)p->C * latent_h * latent_w;
    if (p->hp > e->cfg.pos_embed_max_size || p->wp > e->cfg.pos_embed_max_size) {
        const int hp = p->hp, wp = p->wp;
        delete p;
        return fail("latent grid %dx%d exceeds pos_embed_max_size %d", hp, wp, e->cfg.pos_embed_max_size);
    }
    const int D = e->D, F = e->F;
    const int64_t rows_cond = (int64_t)max_steps * p->Bp;
    size_t off = 0;
    auto take = [&](int64_t elems, int esz) {
        size_t o = off;
        off += (((size_t)elems * esz) + 255) & ~(size_t)255;
        return o;
    };
    const int64_t qk_el = (int64_t)p->Bp * e->cfg.num_heads * p->S_pad * 64;
    const int Ni_pad = (p->Ni + 63) / 64 * 64;
    const int64_t qk2_el = (int64_t)p->Bp * e->cfg.num_heads * Ni_pad * 64;
    size_t o_pe = take((int64_t)p->Ni * D, 2), o_patch = take((int64_t)p->Mi * e->KP, 2);
    size_t o_x = take((int64_t)p->Mi * D, 2), o_c = take((int64_t)p->Mc * D, 2), o_c0 = take((int64_t)p->Mc * D, 2);
    size_t o_xn = take((int64_t)p->Mi * D, 2), o_xn2 = take((int64_t)p->Mi * D, 2), o_cn = take((int64_t)p->Mc * D, 2);
    size_t o_q = take(qk_el, 2), o_k = take(qk_el, 2), o_vT = take(qk_el, 2);
    size_t o_q2 = take(qk2_el, 2), o_k2 = take(qk2_el, 2), o_vT2 = take(qk2_el, 2);
    size_t o_oi = take((int64_t)p->Mi * D, 2), o_oc = take((int64_t)p->Mc * D, 2);
    size_t o_hid = take((int64_t)p->Mi * F, 2), o_chid = take((int64_t)p->Mc * F, 2);
    size_t o_tp = take(rows_cond * e->cfg.time_proj_dim, 2), o_h1 = take(rows_cond * D, 2);
    size_t o_p1 = take((int64_t)p->Bp * D, 2), o_pemb = take((int64_t)p->Bp * D, 2), o_semb = take(rows_cond * D, 2);
    size_t o_mod = take(rows_cond * e->mod_cols, 2), o_v = take((int64_t)p->Bp * p->n_lat, 2);
    size_t o_t = take(rows_cond, 4), o_sc = take(3 * (int64_t)max_steps, 4);
    size_t o_ii = take((int64_t)p->B * p->n_lat, 4), o_it = take((int64_t)(max_steps + 1) * p->B * p->n_lat, 4);
    size_t o_in = take((int64_t)max_steps * p->B * p->n_lat, 4), o_il = take((int64_t)max_steps * p->B, 4);
    size_t o_ipe = take((int64_t)p->B * p->Nt * e->cfg.joint_attention_dim, 2), o_ipp = take((int64_t)p->B * e->cfg.pooled_projection_dim, 2);
    size_t o_ine = take((int64_t)p->B * p->Nt * e->cfg.joint_attention_dim, 2), o_inp = take((int64_t)p->B * e->cfg.pooled_projection_dim, 2);
    p->ws_bytes = off;
    if (hipMalloc((void**)&p->ws, off) != hipSuccess) {
        int r = fail("mi355_plan_create: hipMalloc of %zu bytes failed", off);
        delete p;
        return r;
    }
    // zero once: the padded key rows / columns of q,k,vT must stay finite (attention masks them)
    if (hipMemset(p->ws, 0, off) != hipSuccess) {
        (void)hipFree(p->ws);
        delete p;
        return fail("mi355_plan_create: hipMemset failed");
    }
    char* w = p->ws;
    p->pe = (bf16_t*)(w + o_pe); p->patches = (bf16_t*)(w + o_patch);
    p->x = (bf16_t*)(w + o_x); p->c = (bf16_t*)(w + o_c); p->c0 = (bf16_t*)(w + o_c0);
    p->xn = (bf16_t*)(w + o_xn); p->xn2 = (bf16_t*)(w + o_xn2); p->cn = (bf16_t*)(w + o_cn);
    p->q = (bf16_t*)(w + o_q); p->k = (bf16_t*)(w + o_k); p->vT = (bf16_t*)(w + o_vT);
    p->q2 = (bf16_t*)(w + o_q2); p->k2 = (bf16_t*)(w + o_k2); p->vT2 = (bf16_t*)(w + o_vT2);
    p->o_img = (bf16_t*)(w + o_oi); p->o_ctx = (bf16_t*)(w + o_oc);
    p->hid = (bf16_t*)(w + o_hid); p->chid = (bf16_t*)(w + o_chid);
    p->tproj = (bf16_t*)(w + o_tp); p->h1 = (bf16_t*)(w + o_h1); p->p1 = (bf16_t*)(w + o_p1);
    p->pemb = (bf16_t*)(w + o_pemb); p->semb = (bf16_t*)(w + o_semb); p->mod_all = (bf16_t*)(w + o_mod);
    p->v = (bf16_t*)(w + o_v);
    p->t_dev = (float*)(w + o_t); p->scal = (float*)(w + o_sc);
    p->io_init = w + o_ii; p->io_traj = w + o_it; p->io_noise = (float*)(w + o_in); p->io_lp = (float*)(w + o_il);
    p->io_pe = (bf16_t*)(w + o_ipe); p->io_pp = (bf16_t*)(w + o_ipp); p->io_ne = (bf16_t*)(w + o_ine); p->io_np = (bf16_t*)(w + o_inp);
    *out = p;
    return 0;
}

extern "C" int mi355_plan_destroy(mi355_plan* p) {
    if (!p) return 0;
    train_release(p);
    if (p->gexec) (void)hipGraphExecDestroy(p->gexec);
    if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
    for (hipEvent_t ev : p->ev_join) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : p->ev_fork) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : p->ev_vfork) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : p->ev_vjoin) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : p->ev_djoin) (void)hipEventDestroy(ev);
    if (p->side) (void)hipStreamDestroy(p->side);
    if (p->side_v) (void)hipStreamDestroy(p->side_v);
    if (p->ws) (void)hipFree(p->ws);
    delete p;
    return 0;
}
extern "C" int64_t mi355_plan_workspace_bytes(mi355_plan* p) { return p ? (int64_t)p->ws_bytes : 0; }

// ---- static score bounds ----------------------------------------------------------------------
// q and k reach the attention kernel RMS-normalised per head: ||q_hat|| <= 8 max|w_q|, ||k_hat|| <= 8 max|w_k| (64 dims), so with
// the folded scale |score| <= 64 * 0.125 * log2(e) * max|w_q| * max|w_k| (Cauchy-Schwarz; 2 % slack for the bf16 roundings).  When
// that is <= 60 the softmax needs no running max (attention.hip, STATIC).  Recomputed (one small D2H copy + stream sync) only after
// a norm weight was re-bound, i.e. once per weight refresh, never inside a rollout.
static int g_attn_static = 1;
static int update_score_bounds(mi355_engine* e, hipStream_t st) {
    if (!e->bounds_dirty) return 0;
    std::vector<float> host(e->used32 / 4);
    HIPCHK(hipMemcpyAsync(host.data(), e->arena32, e->used32, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // |q . k| = |sum_d qn_d a_d kn_d b_d| <= max_d |a_d b_d| * sum_d |qn_d kn_d| <= max_d |a_d b_d| * |qn| |kn|, and an RMS-normalised head has
    // |qn| <= sqrt(64): the bound needs the largest PRODUCT of the two norm weights at one channel, not the product of their largest entries
    // (a checkpoint whose q and k weights peak at different channels keeps the static kernel)
    auto pmax = [&](const float* dq, const float* dk) {
        const float* a = host.data() + (dq - (const float*)e->arena32);
        const float* b = host.data() + (dk - (const float*)e->arena32);
        float m = 0.f;
        for (int i = 0; i < 64; ++i) m = fmaxf(m, fabsf(a[i] * b[i]));
        return m;
    };
    const float c = 64.0f * 0.125f * 1.4426950408889634f * 1.02f;
    for (auto& b : e->blk) {
        // joint attention: image / text queries against image / text keys
        b.bound_joint = c * fmaxf(fmaxf(pmax(b.nq, b.nk), pmax(b.nq, b.nck)), fmaxf(pmax(b.ncq, b.nk), pmax(b.ncq, b.nck)));
        b.bound_dual = b.dual ? c * pmax(b.nq2, b.nk2) : 0.f;
    }
    e->bounds_dirty = false;
    ++e->bounds_ver;
    return 0;
}

// how many attention launches of a forward take the static-bound (no running max) kernel with the weights bound now
extern "C" int mi355_engine_attention_info(mi355_engine* e, void* stream, int* n_static, int* n_total, float* max_bound) {
    if (!e) return fail("mi355_engine_attention_info: null engine");
    CHK(mi355_engine_weights_ready(e));
    CHK(update_score_bounds(e, (hipStream_t)stream));
    int ns = 0, nt = 0;
    float mb = 0.f;
    for (auto& b : e->blk) {
        ++nt; if (g_attn_static && b.bound_joint > 0.f && b.bound_joint <= 60.f) ++ns;
        mb = fmaxf(mb, b.bound_joint);
        if (b.dual) { ++nt; if (g_attn_static && b.bound_dual > 0.f && b.bound_dual <= 60.f) ++ns; mb = fmaxf(mb, b.bound_dual); }
    }
    if (n_static) *n_static = ns;
    if (n_total) *n_total = nt;
    if (max_bound) *max_bound = mb;
    return 0;
}

// ---------------------------------------------------------------------------------- forward
static GemmParams gp(const bf16_t* A, long lda, const bf16_t* W, long ldw, int M, int N, int K, int epi,
                     const float* bias, bf16_t* out, long ldo) {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = M; g.N = N; g.K = K; g.epi = epi; g.bias = bias;
    g.out = out; g.ldo = ldo; g.rows_per_sample = M > 0 ? M : 1; g.eps = 1e-6f;
    return g;
}

// step-invariant prompt work: context embedder (K2) + pooled-text MLP.  enc_b == NULL => n_cfg 1.
static int prepare_prompt(mi355_plan* p, hipStream_t st, const void* enc_a, const void* pooled_a, const void* enc_b,
                          const void* pooled_b) {
    mi355_engine* e = p->e;
    const int D = e->D, J = e->cfg.joint_attention_dim, P = e->cfg.pooled_projection_dim;
    const void* encs[2] = {enc_a, enc_b};
    const void* pools[2] = {pooled_a, pooled_b};
    if (p->ncfg == 2 && (!enc_b || !pooled_b)) return fail("n_cfg == 2 needs both prompt halves");
    if (!enc_a || !pooled_a) return fail("prompt embeddings are NULL");
    for (int half = 0; half < p->ncfg; ++half) {
        const int rows = p->B * p->Nt;
        GemmParams g = gp((const bf16_t*)encs[half], J, e->w_ctx, J, rows, D, J, EPI_BIAS, e->b_ctx,
                          p->c0 + (int64_t)half * rows * D, D);
        HIPCHK(gemm_p(g, st));
        GemmParams g1 = gp((const bf16_t*)pools[half], P, e->w_p1, P, p->B, D, P, EPI_BIAS_SILU, e->b_p1,
                           p->p1 + (int64_t)half * p->B * D, D);
        HIPCHK(gemm_p(g1, st));
    }
    GemmParams g2 = gp(p->p1, D, e->w_p2, D, p->Bp, D, D, EPI_BIAS, e->b_p2, p->pemb, D);
    HIPCHK(gemm_p(g2, st));
    return 0;
}

// conditioning for `nsteps` steps at once (K1 + K3 hoisted): t_dev holds nsteps*Bp timesteps
static int prepare_conditioning(mi355_plan* p, hipStream_t st, int nsteps, int t_round_dt) {
    mi355_engine* e = p->e;
    const int D = e->D, T = e->cfg.time_proj_dim;
    const int rows = nsteps * p->Bp;
    HIPCHK(launch_time_proj(p->t_dev, rows, T, t_round_dt, p->tproj, st));
    GemmParams g1 = gp(p->tproj, T, e->w_t1, T, rows, D, T, EPI_BIAS_SILU, e->b_t1, p->h1, D);
    HIPCHK(gemm_p(g1, st));
    // semb = silu(bf16(timestep_emb + pooled_emb)): every AdaLN consumes silu(temb)
    GemmParams g2 = gp(p->h1, D, e->w_t2, D, rows, D, D, EPI_ADDSRC_SILU, e->b_t2, p->semb, D);
    g2.aux = p->pemb; g2.ld_aux = D; g2.rows_per_sample = p->Bp;
    HIPCHK(gemm_p(g2, st));
    GemmParams g3 = gp(p->semb, D, e->w_mod, D, rows, e->mod_cols, D, EPI_BIAS, e->b_mod, p->mod_all, e->mod_cols);
    HIPCHK(gemm_p(g3, st));
    return 0;
}

// key 29: MEASUREMENT ONLY (scripts/ablate_forward.py) -- a bit mask of launches the forward SKIPS, to put a measured ceiling on what fusing them
// away could buy before building the fusion: 1 = the whole text-stream chain (what a grouped image + text launch would absorb), 2 = every
// LayerNorm-modulate launch (what a GEMM-prologue fusion would absorb), 4 = the V^T projections (what a fused q|k|v weight would absorb).
// The results are WRONG by construction; nothing but the ablation script sets it.
static int g_ablate = 0;

static int ln_mod(mi355_plan* p, hipStream_t st, const bf16_t* x, bf16_t* out, bf16_t* out2, const bf16_t* mod, int M,
                  int rps, int shift_off, int scale_off, int shift2_off, int scale2_off) {
    LnModParams l;
    l.x = x; l.out = out; l.out2 = out2; l.mod = mod; l.mod_ld = p->e->mod_cols;
    l.shift_off = shift_off; l.scale_off = scale_off; l.shift2_off = shift2_off; l.scale2_off = scale2_off;
    l.M = M; l.D = p->e->D; l.rows_per_sample = rps; l.eps = p->e->cfg.eps;
    if (g_ablate & 2) return 0;
    HIPCHK(lnmod_p(l, st));
    return 0;
}

// q/k projection with fused bias + per-head RMSNorm + scatter, and V^T projection (operands swapped)
static int qkv_proj(mi355_plan* p, hipStream_t st, const bf16_t* xin, int M, int rps, const bf16_t* w_qk,
                    const float* b_qk, const bf16_t* w_v, const float* b_v, const float* nq, const float* nk,
                    bf16_t* q, bf16_t* k, bf16_t* vT, int S_pad, int s_off, hipStream_t st_v = nullptr, bool v_only = false,
                    bool qk_only = false) {
    if (!st_v) st_v = st;                 // the V^T GEMM reads the same input as the q|k GEMM and may run beside it
    mi355_engine* e = p->e;
    const int D = e->D;
    GemmParams g = gp(xin, D, w_qk, D, M, 2 * D, D, EPI_QK_NORM, b_qk, nullptr, 0);
    g.q = q; g.k = k; g.nw_q = nq; g.nw_k = nk; g.H = e->cfg.num_heads; g.S_pad = S_pad; g.s_off = s_off;
    g.rows_per_sample = rps; g.eps = e->cfg.eps;
    // deferred-rescale attention consumes q with the softmax scale 0.125*log2(e) folded in (same single bf16 rounding)
    if (get_attn_variant() >= 1) g.q_scale = 0.125f * 1.4426950408889634f;
    if (!v_only) HIPCHK(gemm_p(g, st));
    GemmParams gv = gp(w_v, D, xin, D, D, M, D, EPI_VT, b_v, nullptr, 0);
    gv.q = vT; gv.H = e->cfg.num_heads; gv.S_pad = S_pad; gv.s_off = s_off; gv.rows_per_sample = rps;
    if (!qk_only && !(g_ablate & 4)) HIPCHK(gemm_p(gv, st_v));
    return 0;
}

static int gate_res(mi355_plan* p, hipStream_t st, const bf16_t* A, int K, const bf16_t* W, const float* bias, bf16_t* x,
                    int M, int rps, const bf16_t* mod, int gate_off) {
    GemmParams g = gp(A, K, W, K, M, p->e->D, K, EPI_GATE_RES, bias, x, p->e->D);
    g.aux = mod + gate_off; g.ld_aux = p->e->mod_cols; g.rows_per_sample = rps;
    HIPCHK(gemm_p(g, st));
    return 0;
}

// ---- two-stream forward ------------------------------------------------------------------------------------------------------
// Between two joint attentions the text-stream chain of an MMDiT block (out-projection, LN-modulate, MLP, the next block's LN-modulate and
// q|k / V^T projections) is independent of the image-stream chain.  Its grids are small (M = B'.Nt rows: 11-132 workgroups) and, in
// small-batch configurations, so are the image grids (M <= 8192: 96-384 tiles on 256 CUs): launched back to back on one stream each of
// them leaves most of the chip idle.  With the text chain on a plan-owned side stream (fork after every attention, join before the next
// one) the two chains' workgroups share the CUs; inside the captured rollout the fork / join events become graph edges, so one hipGraph
// launch replays a two-branch DAG per block.  Same kernels, same operands, same arithmetic: results are bit-identical to the
// single-stream order.  Mode (mi355_tune_set key 8): 0 = single stream, 1 = always, 2 (default) = when the image stream has at most
// `g_two_stream_rows` rows (key 9; default 32 768 = the largest shape measured, B = 8 at 1024^2: +2.4 %; B = 1 / 2 / 4: +31 / +12 / +9 %,
// the reference's 512^2 examples +6 ... +22 %: profiles/r02b_two_stream_ab.txt).
static int g_two_stream = 2;
static int g_two_stream_rows = 32768;
// key 10: fork point in dual-attention blocks: 0 = right after the joint attention (the text chain also runs beside the dual attention:
// ~1 % more at small batches), 1 = after the block's last attention (at B' = 8, 1024^2 the dual attention fills every CU and text GEMMs
// squeezed in beside it only stretch it: 840 -> 900 us per launch), 2 (default) = 1 for plans with more than 16 384 image rows, else 0
static int g_two_stream_late_fork = 2;
static bool late_fork_wanted(const mi355_plan* p) {
    return g_two_stream_late_fork == 1 || (g_two_stream_late_fork == 2 && p->Mi > 16384);
}
// key 11: a THIRD stream for the image stream's V^T projection (same input as the q|k projection) and, in dual-attention blocks, the
// dual attention's q|k / V^T projections hoisted from behind the joint attention to beside it (xn2 comes out of the same LN-modulate
// launch).  Small grids only: at S = 1357 (512^2) the joint attention's grid is 576 workgroups on 512 slots -- its second round leaves
// 7/8 of the chip idle -- and a 128x128-tile GEMM at M = 4096 fills 0.75 or 1.5 rounds.  0 = off, 1 = on whenever two streams are,
// 2 = on for plans with at most 16 384 image rows (beside a chip-filling attention the extra GEMMs only stretch it).
static int g_three_stream = 0;
static bool two_stream_wanted(const mi355_plan* p);
static bool three_stream_wanted(const mi355_plan* p) {
    return two_stream_wanted(p) && (g_three_stream == 1 || (g_three_stream == 2 && p->Mi <= 16384));
}
static bool two_stream_wanted(const mi355_plan* p) {
    return g_two_stream == 1 || (g_two_stream == 2 && p->Mi <= g_two_stream_rows);
}
static int two_stream_init(mi355_plan* p) {
    if (p->side) return 0;
    HIPCHK(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
    const size_t n = (size_t)p->e->L + 1;
    for (size_t i = 0; i < n; ++i) {
        hipEvent_t a, b;
        HIPCHK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        p->ev_join.push_back(a);
        HIPCHK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        p->ev_fork.push_back(b);
    }
    HIPCHK(hipStreamCreateWithFlags(&p->side_v, hipStreamNonBlocking));
    for (size_t i = 0; i < n; ++i) {
        hipEvent_t a, b, c;
        HIPCHK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        p->ev_vfork.push_back(a);
        HIPCHK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        p->ev_vjoin.push_back(b);
        HIPCHK(hipEventCreateWithFlags(&c, hipEventDisableTiming));
        p->ev_djoin.push_back(c);
    }
    return 0;
}

// One transformer forward.  `mod` = this step's rows of mod_all; c0 / pe / conditioning prepared.
static int forward_core(mi355_plan* p, hipStream_t st, const void* latents, int lat_dt, const bf16_t* mod, bf16_t* v_out) {
    mi355_engine* e = p->e;
    const int D = e->D, F = e->F, H = e->cfg.num_heads;
    const int Mi = p->Mi, Mc = p->Mc, Ni = p->Ni, Nt = p->Nt;
    const int Ni_pad = (Ni + 63) / 64 * 64;
    // The mid-size GEMM kernel (gemm.hip, one 160-KiB workgroup per CU) keeps the text chain's workgroups off the CUs it holds.  Measured
    // in-model (profiles/r06d_*): that pays where the text chain is a real share of the work -- the reference's 512^2 B = 2 CFG example,
    // 1332 text rows beside 4096 image rows: +2.7 % -- and costs where it is a sliver (1024^2 B = 1, 333 rows: -1.6 %).  The choice depends on
    // the PLAN's shape only, never on data, and the kernels are bit-identical: a sample's result does not depend on it.
    struct MidHint { explicit MidHint(int v) { set_mid_plan_hint(v); } ~MidHint() { set_mid_plan_hint(1); } } mid_hint(4L * Mc >= Mi ? 1 : 0);
    if (!p->pos_ready) {
        HIPCHK(launch_pos_crop(e->pos_embed, p->pe, e->cfg.pos_embed_max_size, p->hp, p->wp, D, st));
        p->pos_ready = true;
    }
    // K0: patch-embed = im2col + GEMM + bias + pos-embed
    HIPCHK(launch_patchify(latents, lat_dt, p->patches, p->B, p->ncfg, p->C, p->h, p->w, e->cfg.patch_size, st));
    {
        GemmParams g = gp(p->patches, e->KP, e->w_patch, e->KP, Mi, D, e->KP, EPI_POSADD, e->b_patch, p->x, D);
        g.aux = p->pe; g.ld_aux = D; g.rows_per_sample = Ni;
        HIPCHK(gemm_p(g, st));
    }
    HIPCHK(copy_d2d(p->c, p->c0, (size_t)Mc * D * 2, st));
    // `ts` carries the text-stream chain: the caller's stream, or the plan's side stream (see above)
    const bool two = two_stream_wanted(p);
    hipStream_t ts = st;
    if (two) {
        CHK(two_stream_init(p));
        ts = p->side;
        HIPCHK(ev_record(p->ev_fork[e->L], st));           // c, the conditioning and the previous forward are complete on `st`
        HIPCHK(ev_wait(ts, p->ev_fork[e->L]));
    }
    const bool three = two && three_stream_wanted(p);
    hipStream_t vs = three ? p->side_v : st;
    bool text_open = false;      // the side stream holds work that `st` has not waited for yet
    for (int i = 0; i < e->L; ++i) {
        const BlockW& b = e->blk[i];
        const int mi = b.mod_img, mc = b.mod_ctx;
        // AdaLN-Zero chunk order: shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp[, shift2, scale2, gate2]
        CHK(ln_mod(p, st, p->x, p->xn, b.dual ? p->xn2 : nullptr, mod, Mi, Ni, mi + 0 * D, mi + 1 * D, mi + 6 * D, mi + 7 * D));
        const bool text = !(g_ablate & 1);
        if (!text) {}
        else if (b.last)  // AdaLayerNormContinuous: scale first, then shift
            CHK(ln_mod(p, ts, p->c, p->cn, nullptr, mod, Mc, Nt, mc + 1 * D, mc + 0 * D, 0, 0));
        else
            CHK(ln_mod(p, ts, p->c, p->cn, nullptr, mod, Mc, Nt, mc + 0 * D, mc + 1 * D, 0, 0));
        // joint attention: image tokens first, then text
        if (three) {             // xn / xn2 are complete on `st`; the previous block's attentions (readers of q2 / k2 / vT2) are behind them
            HIPCHK(ev_record(p->ev_vfork[i], st));
            HIPCHK(ev_wait(vs, p->ev_vfork[i]));
        }
        CHK(qkv_proj(p, st, p->xn, Mi, Ni, b.w_qk, b.b_qk, b.w_v, b.b_v, b.nq, b.nk, p->q, p->k, p->vT, p->S_pad, 0, vs));
        if (text) CHK(qkv_proj(p, ts, p->cn, Mc, Nt, b.w_cqk, b.b_cqk, b.w_cv, b.b_cv, b.ncq, b.nck, p->q, p->k, p->vT, p->S_pad, Ni));
        if (three) {
            HIPCHK(ev_record(p->ev_vjoin[i], vs));
            HIPCHK(ev_wait(st, p->ev_vjoin[i]));
            if (b.dual) {        // the dual attention's projections: beside the joint attention instead of behind it
                CHK(qkv_proj(p, vs, p->xn2, Mi, Ni, b.w_qk2, b.b_qk2, b.w_v2, b.b_v2, b.nq2, b.nk2, p->q2, p->k2, p->vT2, Ni_pad, 0));
                HIPCHK(ev_record(p->ev_djoin[i], vs));
            }
        }
        if (two) {               // join: the attention reads the text rows of q / k / vT and overwrites o_ctx
            HIPCHK(ev_record(p->ev_join[i], ts));
            HIPCHK(ev_wait(st, p->ev_join[i]));
            text_open = false;
        }
        {
            AttnParams a{p->q, p->k, p->vT, p->o_img, p->o_ctx, p->Bp, H, p->S, p->S_pad, Ni, get_attn_variant() >= 1,
                         g_attn_static ? b.bound_joint : 0.f};
            HIPCHK(attn_p(a, st));
        }
        // fork: text work follows (never after the final attention: nothing would join it).  The fork point is the block's LAST attention
        // launch -- the dual (image-only) attention of a dual block also fills every CU, and text-stream GEMMs squeezed in beside it only
        // stretch it (measured: attention 840 -> 900 us per launch with the fork before it); it touches neither o_ctx nor c.
        const bool fork_here = two && (!b.last || i + 1 < e->L);
        auto fork = [&]() -> int {
            HIPCHK(ev_record(p->ev_fork[i], st));
            HIPCHK(ev_wait(ts, p->ev_fork[i]));
            text_open = true;
            return 0;
        };
        const bool late = b.dual && late_fork_wanted(p);
        if (fork_here && !late) CHK(fork());
        CHK(gate_res(p, st, p->o_img, D, b.w_o, b.b_o, p->x, Mi, Ni, mod, mi + 2 * D));
        if (!two && !b.last && text) CHK(gate_res(p, st, p->o_ctx, D, b.w_co, b.b_co, p->c, Mc, Nt, mod, mc + 2 * D));
        if (b.dual) {
            if (three) HIPCHK(ev_wait(st, p->ev_djoin[i]));
            else CHK(qkv_proj(p, st, p->xn2, Mi, Ni, b.w_qk2, b.b_qk2, b.w_v2, b.b_v2, b.nq2, b.nk2, p->q2, p->k2, p->vT2, Ni_pad, 0));
            // S == n_img: this launch writes o_img only, never o_ctx (which the text chain reads after the fork)
            AttnParams a{p->q2, p->k2, p->vT2, p->o_img, p->o_ctx, p->Bp, H, Ni, Ni_pad, Ni, get_attn_variant() >= 1,
                         g_attn_static ? b.bound_dual : 0.f};
            HIPCHK(attn_p(a, st));
            if (fork_here && late) CHK(fork());
            CHK(gate_res(p, st, p->o_img, D, b.w_o2, b.b_o2, p->x, Mi, Ni, mod, mi + 8 * D));
        }
        if (two && !b.last && text) CHK(gate_res(p, ts, p->o_ctx, D, b.w_co, b.b_co, p->c, Mc, Nt, mod, mc + 2 * D));
        // MLP (image stream)
        CHK(ln_mod(p, st, p->x, p->xn, nullptr, mod, Mi, Ni, mi + 3 * D, mi + 4 * D, 0, 0));
        {
            GemmParams g = gp(p->xn, D, b.w_ff1, D, Mi, F, D, EPI_BIAS_GELU, b.b_ff1, p->hid, F);
            HIPCHK(gemm_p(g, st));
        }
        CHK(gate_res(p, st, p->hid, F, b.w_ff2, b.b_ff2, p->x, Mi, Ni, mod, mi + 5 * D));
        if (!b.last && text) {
            CHK(ln_mod(p, ts, p->c, p->cn, nullptr, mod, Mc, Nt, mc + 3 * D, mc + 4 * D, 0, 0));
            GemmParams g = gp(p->cn, D, b.w_cff1, D, Mc, F, D, EPI_BIAS_GELU, b.b_cff1, p->chid, F);
            HIPCHK(gemm_p(g, ts));
            CHK(gate_res(p, ts, p->chid, F, b.w_cff2, b.b_cff2, p->c, Mc, Nt, mod, mc + 5 * D));
        }
    }
    if (two && text_open) {      // a model whose last block keeps its text stream: rejoin before the caller's stream goes on
        HIPCHK(ev_record(p->ev_join[e->L], ts));
        HIPCHK(ev_wait(st, p->ev_join[e->L]));
    }
    // norm_out (scale first) + proj_out + unpatchify
    CHK(ln_mod(p, st, p->x, p->xn, nullptr, mod, Mi, Ni, e->mod_out + 1 * D, e->mod_out + 0 * D, 0, 0));
    {
        const int NO = e->cfg.patch_size * e->cfg.patch_size * e->cfg.out_channels;
        GemmParams g = gp(p->xn, D, e->w_proj, D, Mi, NO, D, EPI_UNPATCH, e->b_proj, v_out, 0);
        g.hp = p->hp; g.wp = p->wp; g.patch = e->cfg.patch_size; g.out_ch = e->cfg.out_channels;
        HIPCHK(gemm_p(g, st));
    }
    return 0;
}

extern "C" int mi355_transformer_forward(mi355_plan* p, void* stream, const void* latents, int lat_dtype, const float* t,
                                         int t_round_dtype, const void* enc_a, const void* pooled_a, const void* enc_b,
                                         const void* pooled_b, void* v_out) {
    if (!p || !latents || !t || !v_out) return fail("mi355_transformer_forward: null argument");
    CHK(mi355_engine_weights_ready(p->e));
    hipStream_t st = (hipStream_t)stream;
    CHK(update_score_bounds(p->e, st));
    CHK(prepare_prompt(p, st, enc_a, pooled_a, enc_b, pooled_b));
    HIPCHK(copy_d2d(p->t_dev, t, (size_t)p->Bp * 4, st));
    CHK(prepare_conditioning(p, st, 1, t_round_dtype));
    return forward_core(p, st, latents, lat_dtype, p->mod_all, (bf16_t*)v_out);
}

static int sde_call(hipStream_t st, int batch, int64_t n, const void* v_text, const void* v_uncond, int v_dtype, float guidance,
                    const void* latents, int lat_dtype, const float* noise, const void* next_in, int next_in_dtype,
                    const float* sigma, const float* sigma_next, const float* eta, int scalar_stride, float sigma_max,
                    int dynamics, int compute_log_prob, void* next_out, float* next_f32, float* mean_out,
                    float* noise_pred_out, float* log_prob, float* std_dev_t, float* dt) {
    if (!v_text || !latents || !sigma || !sigma_next || !eta) return fail("mi355_sde_step: null argument");
    if (!next_in && !noise && dynamics != MI355_ODE) return fail("mi355_sde_step: SDE rollout step needs `noise`");
    if (dynamics < 0 || dynamics > 3) return fail("mi355_sde_step: unknown dynamics %d", dynamics);
    if (lat_dtype < 0 || lat_dtype > 2) return fail("mi355_sde_step: bad latent dtype %d", lat_dtype);
    if (v_dtype < 0 || v_dtype > 2) return fail("mi355_sde_step: bad noise_pred dtype %d", v_dtype);
    SdeStepParams s;
    memset(&s, 0, sizeof(s));
    s.v_text = (const bf16_t*)v_text; s.v_uncond = (const bf16_t*)v_uncond; s.v_dt = v_dtype; s.guidance = guidance;
    s.latents = latents; s.lat_dt = lat_dtype; s.noise = noise; s.next_in = next_in; s.next_in_dt = next_in_dtype;
    s.sigma = sigma; s.sigma_next = sigma_next; s.eta = eta; s.scalar_stride = scalar_stride; s.sigma_max = sigma_max;
    s.dynamics = dynamics; s.compute_log_prob = compute_log_prob; s.B = batch; s.n = n;
    s.next_out = next_out; s.next_out_dt = lat_dtype; s.next_f32 = next_f32; s.mean_out = mean_out;
    s.noise_pred_out = noise_pred_out; s.log_prob = log_prob; s.std_dev_t = std_dev_t; s.dt_out = dt;
    HIPCHK(sde_p(s, st));
    return 0;
}

extern "C" int mi355_sde_step(void* stream, int batch, int64_t n, const void* v_text, const void* v_uncond, int v_dtype, float guidance,
                              const void* latents, int lat_dtype, const float* noise, const void* next_in,
                              int next_in_dtype, const float* sigma, const float* sigma_next, const float* eta,
                              int scalar_stride, float sigma_max, int dynamics, int compute_log_prob, void* next_out,
                              float* next_f32, float* mean_out, float* noise_pred_out, float* log_prob, float* std_dev_t,
                              float* dt) {
    return sde_call((hipStream_t)stream, batch, n, v_text, v_uncond, v_dtype, guidance, latents, lat_dtype, noise, next_in,
                    next_in_dtype, sigma, sigma_next, eta, scalar_stride, sigma_max, dynamics, compute_log_prob, next_out,
                    next_f32, mean_out, noise_pred_out, log_prob, std_dev_t, dt);
}

// adjoint of mi355_sde_step w.r.t. the network output(s): upstream gradients of (log_prob [B], noise_pred [B][n], next_latents_mean [B][n])
// (any may be NULL) -> dv [n_cfg * B][n] fp32, order [uncond, text] (reference: autograd through FlowMatchEulerDiscreteSDEScheduler.step,
// scheduler/flow_match_euler_discrete.py:305-426, and the CFG combine sd3_5.py:431-433).  v_text / v_uncond: the bf16 predictions the
// forward step consumed.  One op for every model family (the FLUX.1 replay: mi355_flux_forward_train -> mi355_sde_step -> ... -> this ->
// mi355_flux_backward).
// UniPC multistep update of the evaluation-mode Wan sampler (kernels: sde_step.hip; coefficients: mi355_flow/unipc.py)
extern "C" int mi355_unipc_convert(void* stream, const void* v_text, const void* v_uncond, int v_dtype, float guidance, const void* sample,
                                   int sample_dtype, float sigma, float* x0_out, int64_t n) {
    if (!v_text || !sample || !x0_out) return fail("mi355_unipc_convert: null argument");
    if (n <= 0 || (n & 3)) return fail("mi355_unipc_convert: n must be a positive multiple of 4 (got %lld)", (long long)n);
    HIPCHK(launch_unipc_convert(v_text, v_uncond, v_dtype, guidance, sample, sample_dtype, sigma, x0_out, (long)n, (hipStream_t)stream));
    return 0;
}
extern "C" int mi355_op_lincomb(void* stream, int n_terms, const void* const* tensors, const int* dtypes, const float* coefs, void* out, int out_dtype,
                                int64_t n) {
    if (!tensors || !dtypes || !coefs || !out) return fail("mi355_op_lincomb: null argument");
    if (n_terms < 1 || n_terms > 5) return fail("mi355_op_lincomb: 1..5 terms (got %d)", n_terms);
    if (n <= 0 || (n & 3)) return fail("mi355_op_lincomb: n must be a positive multiple of 4 (got %lld)", (long long)n);
    HIPCHK(launch_lincomb(n_terms, tensors, dtypes, coefs, out, out_dtype, (long)n, (hipStream_t)stream));
    return 0;
}

extern "C" int mi355_sde_step_bwd(void* stream, int batch, int64_t n, const void* v_text, const void* v_uncond, float guidance, const void* latents,
                                  int lat_dtype, const void* next_in, int next_in_dtype, const float* sigma, const float* sigma_next, const float* eta,
                                  int scalar_stride, float sigma_max, int dynamics, int compute_log_prob, const float* g_log_prob,
                                  const float* g_noise_pred, const float* g_mean, float* dv) {
    if (!v_text || !latents || !next_in || !sigma || !sigma_next || !eta || !dv) return fail("mi355_sde_step_bwd: null argument");
    if (dynamics < 0 || dynamics > 3) return fail("mi355_sde_step_bwd: unknown dynamics %d", dynamics);
    if (lat_dtype < 0 || lat_dtype > 2 || next_in_dtype < 0 || next_in_dtype > 2) return fail("mi355_sde_step_bwd: bad dtype");
    SdeBwdParams s;
    memset(&s, 0, sizeof(s));
    s.v_text = (const bf16_t*)v_text; s.v_uncond = (const bf16_t*)v_uncond; s.guidance = guidance;
    s.latents = latents; s.lat_dt = lat_dtype; s.next_in = next_in; s.next_in_dt = next_in_dtype;
    s.sigma = sigma; s.sigma_next = sigma_next; s.eta = eta; s.scalar_stride = scalar_stride; s.sigma_max = sigma_max;
    s.dynamics = dynamics; s.compute_log_prob = compute_log_prob; s.B = batch; s.n = n;
    s.g_lp = g_log_prob; s.g_np = g_noise_pred; s.g_mean = g_mean; s.dv = dv;
    HIPCHK(launch_sde_step_bwd(s, (hipStream_t)stream));
    return 0;
}

extern "C" int mi355_denoise_step(mi355_plan* p, void* stream, const void* latents, int lat_dtype, const float* t,
                                  const void* enc_a, const void* pooled_a, const void* enc_b, const void* pooled_b,
                                  float guidance, const float* noise, const void* next_in, int next_in_dtype,
                                  const float* sigma, const float* sigma_next, const float* eta, int scalar_stride,
                                  float sigma_max, int dynamics, int compute_log_prob, void* next_out, float* next_f32,
                                  float* mean_out, float* noise_pred_out, float* log_prob, float* std_dev_t, float* dt) {
    if (!p) return fail("mi355_denoise_step: null plan");
    CHK(mi355_transformer_forward(p, stream, latents, lat_dtype, t, lat_dtype, enc_a, pooled_a, enc_b, pooled_b, p->v));
    const bf16_t* vu = p->ncfg == 2 ? p->v : nullptr;
    const bf16_t* vt = p->ncfg == 2 ? p->v + (int64_t)p->B * p->n_lat : p->v;
    return sde_call((hipStream_t)stream, p->B, p->n_lat, vt, vu, MI355_BF16, guidance, latents, lat_dtype, noise, next_in,
                    next_in_dtype, sigma, sigma_next, eta, scalar_stride, sigma_max, dynamics, compute_log_prob, next_out,
                    next_f32, mean_out, noise_pred_out, log_prob, std_dev_t, dt);
}

static int g_use_graph = 1;

// The whole rollout on plan-owned buffers only (fixed addresses): this is what gets captured.
static int rollout_body(mi355_plan* p, hipStream_t st, int n_steps, int dynamics, float guidance, int init_dtype,
                        int storage_dtype, float sigma_max, int compute_log_prob) {
    const int Bp = p->Bp;
    if (p->ncfg == 2) CHK(prepare_prompt(p, st, p->io_ne, p->io_np, p->io_pe, p->io_pp));
    else CHK(prepare_prompt(p, st, p->io_pe, p->io_pp, nullptr, nullptr));
    CHK(prepare_conditioning(p, st, n_steps, storage_dtype));
    const size_t esz = storage_dtype == MI355_F32 ? 4 : 2;
    const size_t lat_bytes = (size_t)p->B * p->n_lat * esz;
    // cast_latents(init) -> trajectory position 0 (sd3_5.py:266-267)
    HIPCHK(launch_convert(p->io_init, init_dtype, p->io_traj, storage_dtype, (long)p->B * p->n_lat, st));
    for (int i = 0; i < n_steps; ++i) {
        const bf16_t* mod = p->mod_all + (int64_t)i * Bp * p->e->mod_cols;
        char* cur = p->io_traj + (size_t)i * lat_bytes;
        char* nxt = p->io_traj + (size_t)(i + 1) * lat_bytes;
        CHK(forward_core(p, st, cur, storage_dtype, mod, p->v));
        const bf16_t* vu = p->ncfg == 2 ? p->v : nullptr;
        const bf16_t* vt = p->ncfg == 2 ? p->v + (int64_t)p->B * p->n_lat : p->v;
        // the log-prob of a step is only meaningful (and only read) when its noise level is > 0; the kernel
        // checks eta on the device so that the launch sequence does not depend on which steps are SDE steps
        CHK(sde_call(st, p->B, p->n_lat, vt, vu, MI355_BF16, guidance, cur, storage_dtype, p->io_noise + (int64_t)i * p->B * p->n_lat,
                     nullptr, 0, p->scal + i, p->scal + p->max_steps + i, p->scal + 2 * p->max_steps + i, 0, sigma_max,
                     dynamics, compute_log_prob ? 2 : 0, nxt, nullptr, nullptr, nullptr,
                     compute_log_prob ? p->io_lp + (int64_t)i * p->B : nullptr, nullptr, nullptr));
    }
    return 0;
}

extern "C" int mi355_rollout(mi355_plan* p, void* stream, int n_steps, const float* timesteps_host, const float* sigmas_host,
                             const float* noise_levels_host, int dynamics, float guidance, const void* init_latents,
                             int init_dtype, int storage_dtype, const float* step_noise, const void* prompt_embeds,
                             const void* pooled, const void* neg_embeds, const void* neg_pooled,
                             const int32_t* keep_slot_host, void* out_latents, float* out_log_probs, void* out_final,
                             int compute_log_prob) {
    if (!p || !timesteps_host || !sigmas_host || !noise_levels_host || !init_latents || !prompt_embeds || !pooled)
        return fail("mi355_rollout: null argument");
    if (n_steps < 1 || n_steps > p->max_steps) return fail("mi355_rollout: n_steps %d exceeds the plan's max_steps %d", n_steps, p->max_steps);
    if (storage_dtype < 0 || storage_dtype > 2 || init_dtype < 0 || init_dtype > 2) return fail("mi355_rollout: bad dtype");
    if (p->ncfg == 2 && (!neg_embeds || !neg_pooled)) return fail("mi355_rollout: plan has n_cfg == 2 but no negative prompt embeddings");
    if (!step_noise && dynamics != MI355_ODE) return fail("mi355_rollout: step_noise is NULL");
    if (dynamics < 0 || dynamics > 3) return fail("mi355_rollout: unknown dynamics %d", dynamics);
    CHK(mi355_engine_weights_ready(p->e));
    hipStream_t st = (hipStream_t)stream;
    CHK(update_score_bounds(p->e, st));
    const int Bp = p->Bp;
    // ---- host-side per-step scalars (the reference computes them with .item() syncs inside the loop)
    std::vector<float>& tt = p->host_t;   // plan-owned: must outlive the async H2D copies
    std::vector<float>& sc = p->host_sc;
    tt.assign((size_t)n_steps * Bp, 0.f);
    sc.assign(3 * (size_t)p->max_steps, 0.f);
    for (int i = 0; i < n_steps; ++i) {
        for (int j = 0; j < Bp; ++j) tt[(size_t)i * Bp + j] = timesteps_host[i];
        const float t_next = (i + 1 < n_steps) ? timesteps_host[i + 1] : 0.0f;
        sc[i] = timesteps_host[i] / 1000.0f;                 // sigma      (flow_match_euler_discrete.py:302)
        sc[p->max_steps + i] = t_next / 1000.0f;             // sigma_next (:303)
        sc[2 * p->max_steps + i] = noise_levels_host[i];
    }
    HIPCHK(hipMemcpyAsync(p->t_dev, tt.data(), tt.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(p->scal, sc.data(), sc.size() * 4, hipMemcpyHostToDevice, st));
    // ---- inputs -> fixed-address staging (device-to-device, enqueued)
    const int64_t nl = (int64_t)p->B * p->n_lat;
    const size_t in_esz = init_dtype == MI355_F32 ? 4 : 2;
    const size_t emb_bytes = (size_t)p->B * p->Nt * p->e->cfg.joint_attention_dim * 2;
    const size_t pool_bytes = (size_t)p->B * p->e->cfg.pooled_projection_dim * 2;
    HIPCHK(copy_d2d(p->io_init, init_latents, nl * in_esz, st));
    if (step_noise) HIPCHK(copy_d2d(p->io_noise, step_noise, (size_t)n_steps * nl * 4, st));
    HIPCHK(copy_d2d(p->io_pe, prompt_embeds, emb_bytes, st));
    HIPCHK(copy_d2d(p->io_pp, pooled, pool_bytes, st));
    if (p->ncfg == 2) {
        HIPCHK(copy_d2d(p->io_ne, neg_embeds, emb_bytes, st));
        HIPCHK(copy_d2d(p->io_np, neg_pooled, pool_bytes, st));
    }
    const float sigma_max = sigmas_host[1];
    const int clp = compute_log_prob && out_log_probs;
    // ---- the N-step loop: one hipGraph launch (captured on the second call of a configuration), else eager
    bool launched = false;
    if (g_use_graph && !g_prof.on && p->warmed) {
        const bool same = p->gexec && p->g_steps == n_steps && p->g_dyn == dynamics && p->g_storage == storage_dtype &&
                          p->g_init == init_dtype && p->g_clp == clp && p->g_guidance == guidance && p->g_sigma_max == sigma_max &&
                          p->g_attn == get_attn_variant() && p->g_gemm == get_gemm_variant() && p->g_tune == tune_epoch() &&
                          p->g_bounds == p->e->bounds_ver * 2 + (g_attn_static != 0) && p->g_two == (int)two_stream_wanted(p) * (1 + (int)late_fork_wanted(p) + 2 * (int)three_stream_wanted(p));
        if (!same) {
            if (two_stream_wanted(p)) CHK(two_stream_init(p));   // streams / events are created outside the capture
            if (p->gexec) { (void)hipGraphExecDestroy(p->gexec); p->gexec = nullptr; }
            hipGraph_t graph = nullptr;
            hipError_t ce = hipSuccess;
            if (!p->cap_stream) ce = hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking);
            if (ce == hipSuccess) ce = hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeRelaxed);
            if (ce == hipSuccess) {
                // nothing executes here: the body's launches / D2D copies on plan-owned buffers become graph nodes
                const int rc = rollout_body(p, p->cap_stream, n_steps, dynamics, guidance, init_dtype, storage_dtype, sigma_max, clp);
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
            if (p->gexec) {
                p->g_steps = n_steps; p->g_dyn = dynamics; p->g_storage = storage_dtype; p->g_init = init_dtype; p->g_clp = clp;
                p->g_guidance = guidance; p->g_sigma_max = sigma_max; p->g_attn = get_attn_variant(); p->g_gemm = get_gemm_variant(); p->g_tune = tune_epoch();
                p->g_bounds = p->e->bounds_ver * 2 + (g_attn_static != 0);
                p->g_two = (int)two_stream_wanted(p) * (1 + (int)late_fork_wanted(p) + 2 * (int)three_stream_wanted(p));
            } else {
                // no silent fallback (header convention, reference constraints.md:144-145): the caller decides whether to
                // retry with eager launches (mi355_tune_set(2, 0))
                const hipError_t last = hipGetLastError();
                return fail("mi355_rollout: hipGraph capture / instantiation of the %d-step loop failed (%s); "
                            "mi355_tune_set(2, 0) selects eager launches", n_steps, hipGetErrorString(ce != hipSuccess ? ce : last));
            }
        }
        if (p->gexec) {
            HIPCHK(hipGraphLaunch(p->gexec, st));
            launched = true;
        }
    }
    if (!launched) {
        CHK(rollout_body(p, st, n_steps, dynamics, guidance, init_dtype, storage_dtype, sigma_max, clp));
        p->warmed = true;
    }
    // ---- outputs: kept trajectory positions, log-probs of the SDE steps, final latents
    const size_t esz = storage_dtype == MI355_F32 ? 4 : 2;
    const size_t lat_bytes = (size_t)nl * esz;
    if (keep_slot_host && out_latents)
        for (int i = 0; i <= n_steps; ++i)
            if (keep_slot_host[i] >= 0)
                HIPCHK(hipMemcpyAsync((char*)out_latents + (size_t)keep_slot_host[i] * lat_bytes, p->io_traj + (size_t)i * lat_bytes,
                                      lat_bytes, hipMemcpyDeviceToDevice, st));
    if (clp)
        for (int i = 0; i < n_steps; ++i)
            if (noise_levels_host[i] > 0.f)
                HIPCHK(hipMemcpyAsync(out_log_probs + (int64_t)i * p->B, p->io_lp + (int64_t)i * p->B, (size_t)p->B * 4,
                                      hipMemcpyDeviceToDevice, st));
    if (out_final)
        HIPCHK(copy_d2d(out_final, p->io_traj + (size_t)n_steps * lat_bytes, lat_bytes, st));
    return 0;
}

// A/B knob for kernel variants (key 0: large-grid GEMM kernel, 0 = simple 2-stage, 1 = default (4-wave hand-scheduled loop up to K = key 19,
// else ping-pong), 2 = 4-wave wherever it applies, 3 = ping-pong only)
static int g_tune_epoch = 0;
namespace mi355 { int tune_epoch() { return g_tune_epoch; } }

extern "C" int mi355_tune_set(int key, int value) {
    ++g_tune_epoch;
    // keys of variants that were measured, dropped and deleted (18: hipGraph replay of the Wan loop; 11 stays; 20 / 23: never shipped):
    // accepted as no-ops, so that an MI355_TUNE string from an older round does not make the library refuse to load
    if (key == 18 || key == 20 || key == 23) {
        static bool warned = false;
        if (!warned) { fprintf(stderr, "mi355_flow: mi355_tune_set(%d, .) names a removed variant: ignored\n", key); warned = true; }
        return 0;
    }
    if (key == 0) { set_gemm_variant(value); return 0; }
    if (key == 1) { set_attn_variant(value); return 0; }
    if (key == 2) { g_use_graph = value; return 0; }
    if (key == 3) { set_pp_min_tiles(value); return 0; }
    if (key == 4) { set_conv_cfg(value); return 0; }
    if (key == 5) { set_attn128_variant(value); return 0; }
    if (key == 6) { g_attn_static = value; return 0; }
    if (key == 7) { set_raster_gm(value); return 0; }
    if (key == 8) { g_two_stream = value; return 0; }          // text-stream chain on a side stream: 0 off, 1 on, 2 auto (rows <= key 9)
    if (key == 9) { g_two_stream_rows = value; return 0; }
    if (key == 10) { g_two_stream_late_fork = value; return 0; }
    if (key == 11) { g_three_stream = value; return 0; }
    if (key == 12) { set_qwen_two_stream(value); return 0; }   // Qwen-Image engine: text chain on a side stream (0 off = default, 1 on, 2 auto)
    if (key == 13) { set_qwen_two_stream_rows(value); return 0; }
    if (key == 14) { set_flux_two_stream(value); return 0; }   // FLUX.1 engine, double blocks: the same
    if (key == 15) { set_flux_two_stream_rows(value); return 0; }
    if (key == 16) { set_flux_graph(value); return 0; }        // FLUX.1 engine: hipGraph replay of the rollout loop (0 = default: eager)
    if (key == 17) { set_qwen_graph(value); return 0; }        // Qwen-Image engine: the same
    if (key == 19) { set_w4_max_k(value); return 0; }
    if (key == 24) { set_wan_data_bound(value); return 0; }    // Wan self-attention: static / running-max softmax per (batch, head) by the measured q / k norms (1 = default)
    if (key == 25) { set_rms_bwd_fast(value); return 0; }      // optimize() backward: 1 = 16-byte forms of attn_bwd_prep and of the default-scope RMSNorm-backward gather (0 = default)
    if (key == 26) { set_wgrad_side(value); return 0; }        // weight-gradient GEMMs of the FLUX.1 / Qwen-Image / Wan backward on a side stream: 1 = on (default), 0 = off
    if (key == 27) { set_wgrad_split_model(value); return 0; } // split-K factor of the weight-gradient GEMMs: 1 = modelled-time minimum (default), 0 = the round-2 rule
    if (key == 28) { set_train_text_side(value); return 0; }   // Qwen-Image backward: the text chain on the plan's side stream (1 = default)
    if (key == 31) { set_w4_min_tiles(value); return 0; }      // default GEMM dispatch: smallest 256 x 256-tile grid for the 4-wave hand-scheduled kernel (default 512)
    if (key == 29) {      // MEASUREMENT ONLY: launches the SD3.5 forward skips (WRONG results; scripts/ablate_forward.py)
        // refused unless the process opted in: a stale MI355_TUNE=29=... in a training environment must not corrupt rollouts silently
        const char* ok = getenv("MI355_ALLOW_ABLATION");
        if (value != 0 && !(ok && ok[0] == '1'))
            return fail("mi355_tune_set(29, %d): the forward-ablation mask produces wrong results by construction; set MI355_ALLOW_ABLATION=1 "
                        "to use it for a timing measurement", value);
        if (value != 0) fprintf(stderr, "mi355_flow: FORWARD ABLATION MASK %d ACTIVE -- results of the SD3.5 forward are WRONG (measurement only)\n", value);
        g_ablate = value;
        return 0;
    }
    if (key == 32) { set_mid_mode(value); return 0; }          // mid-size GEMM kernel (128x192 / 192x128 tiles): 0 off, 1 cost rule (default), 2 wherever it applies
    if (key == 33) { set_mid_alpha_percent(value); return 0; } // margin of that rule in percent (default 100)
    if (key == 34) { set_mid_min_tiles(value); return 0; }     // smallest grid of its tiles (default 160)
    if (key == 37) { set_mid_max_tiles(value); return 0; }     // largest grid of its tiles
    if (key == 35 || key == 36) return 0;                      // (round-6 measurement knobs of that kernel -- staggered LDS-DMA slots, launch-class mask -- measured and removed: accepted as no-ops)
    if (key == 39) { g_wgrad_tn = value; set_wgrad_tn_mode(value); return 0; }           // optimize() backward: 1 (default) = weight-gradient GEMMs read dY / X row-major through transposed LDS reads (2: 256 x 256 tiles where they fit), 0 = transposed copies
    if (key == 43) {      // (values 2..5: ablation builds of the dK/dV loop -- no VALU / no LDS reads / no barrier + loads / MFMAs only; WRONG results, measurement only)
        const char* ok = getenv("MI355_ALLOW_ABLATION");
        if (value > 1 && value < 6 && !(ok && ok[0] == '1')) return fail("mi355_tune_set(43, %d): ablation builds produce wrong gradients; set MI355_ALLOW_ABLATION=1 for a timing measurement", value);
        set_attn_bwd_pipe(value); return 0;
    }         if (key == 44) { set_attn128_bwd_pipe(value); return 0; }  // head_dim-128 attention backward: 1 (default) = software-pipelined passes (gen_attn_bwd128.py), 0 = the round-4 kernels
    if (key == 40) { set_w6_mode(value); return 0; }           // 256x192-tile GEMM kernel (round 6): 0 off (default: measured slower inside the two-stream forward), 1 cost rule, 2 wherever it applies
    if (key == 41) { set_w6_alpha_percent(value); return 0; }  // margin of that rule in percent (default 105)
    if (key == 42) { set_w6_min_tiles(value); return 0; }      // smallest grid of its tiles (default 200)
    if (key == 38) { g_fuse_colsum = value; return 0; }        // optimize() backward: 1 (default) = column-sum finish fused into the split-K reduction launch, 0 = two launches
    if (key == 22) { g_train_two_stream = value; return 0; }   // optimize() replay: the context-stream chain of the training forward / backward on a side stream (1 = default)
    if (key == 21) { set_attn128_op_bound(value); return 0; }  // mi355_op_attention128: the |score| bound the caller asserts (0 = none)          // default GEMM dispatch: largest K for the 4-wave hand-scheduled kernel
    return fail("mi355_tune_set: unknown key %d", key);
}

// ----------------------------------------------------------------------- operator-level API
extern "C" int mi355_op_linear(void* stream, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                               int act) {
    if (!A || !W || !bias || !out) return fail("mi355_op_linear: null argument");
    if (K % 64 || N % 4) return fail("mi355_op_linear: K %% 64 and N %% 4 must be 0");
    GemmParams g = gp((const bf16_t*)A, K, (const bf16_t*)W, K, M, N, K, act == 1 ? EPI_BIAS_SILU : act == 2 ? EPI_BIAS_GELU : EPI_BIAS,
                      bias, (bf16_t*)out, N);
    HIPCHK(gemm_p(g, (hipStream_t)stream));
    return 0;
}

// x[M][N] += gate[m / rows_per_sample][N] * (A . W^T + bias), in place: the gated-residual epilogue of the out-projections / second MLP
// linears (K8 / K11 of SURVEY.md 2.3) as an operator, for unit tests and the GEMM A/B scripts
extern "C" int mi355_op_linear_gate_res(void* stream, const void* A, const void* W, const float* bias, const void* gate, void* x, int M, int N,
                                        int K, int rows_per_sample) {
    if (!A || !W || !bias || !gate || !x) return fail("mi355_op_linear_gate_res: null argument");
    if (K % 64 || N % 8 || rows_per_sample < 1) return fail("mi355_op_linear_gate_res: K %% 64, N %% 8 must be 0 and rows_per_sample >= 1");
    GemmParams g = gp((const bf16_t*)A, K, (const bf16_t*)W, K, M, N, K, EPI_GATE_RES, bias, (bf16_t*)x, N);
    g.aux = (const bf16_t*)gate; g.ld_aux = N; g.rows_per_sample = rows_per_sample;
    HIPCHK(launch_gemm(g, (hipStream_t)stream));
    return 0;
}

// weight-gradient product as an operator (unit tests / A/B scripts): variant 1 = gemm_tn.hip on the row-major operands, variant 0 = the
// transposed-copy path (launch_transpose x 2 + the K-contiguous EPI_F32 GEMM).  Output: k_split fp32 partial slabs [N][K].
extern "C" int mi355_op_wgrad(void* stream, const void* dY, int64_t ld_dy, const void* X, int64_t ld_x, float* out, int M, int N, int K, int k_split,
                              int variant, void* scratch, float* colsum) {
    if (!dY || !X || !out || M <= 0 || N <= 0 || K <= 0 || k_split < 1) return fail("mi355_op_wgrad: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (variant == 1 || variant == 2) {          // 2 = the 256 x 256-tile form (N, K multiples of 256; no column sums)
        GemmTnParams tp{(const bf16_t*)dY, (long)ld_dy, (const bf16_t*)X, (long)ld_x, M, N, K, out, (long)K, k_split, (long)N * K, variant == 1 ? colsum : nullptr,
                        variant == 2 ? 1 : 0};
        if (variant == 2 && (N % 256 || K % 256)) return fail("mi355_op_wgrad: variant 2 needs N %% 256 == 0 and K %% 256 == 0");
        if (!gemm_tn_ok(tp)) return fail("mi355_op_wgrad: the row-major-operand kernel needs N %% 128 == 0, K %% 128 == 0, 16-byte aligned rows");
        HIPCHK(launch_gemm_tn(tp, st));
        return 0;
    }
    if (!scratch) return fail("mi355_op_wgrad: variant 0 needs (N + K) * M_pad bf16 of scratch (M_pad = M rounded up to 64)");
    const int Mp = (M + 63) / 64 * 64;                  // the copies are zero-padded to whole 64-row tiles (what the engines' path does)
    bf16_t* aT = (bf16_t*)scratch;
    bf16_t* xT = aT + (size_t)N * Mp;
    HIPCHK(launch_transpose((const bf16_t*)dY, (long)ld_dy, 0, aT, Mp, 0, M, N, Mp, 1, st));
    HIPCHK(launch_transpose((const bf16_t*)X, (long)ld_x, 0, xT, Mp, 0, M, K, Mp, 1, st));
    GemmParams g = gp(aT, Mp, xT, Mp, N, K, Mp, EPI_F32, nullptr, nullptr, K);
    g.q_scale = 1.0f; g.out_f32 = out; g.k_split = k_split; g.split_stride = (long)N * K;
    HIPCHK(launch_gemm(g, st));
    return 0;
}

// debug: same as mi355_op_linear (act 0) with an s_memtime trace buffer (device, >= 256*16*2*4 int64)
extern "C" int mi355_op_linear_trace(void* stream, const void* A, const void* W, const float* bias, void* out, int M, int N, int K,
                                     void* trace) {
    GemmParams g = gp((const bf16_t*)A, K, (const bf16_t*)W, K, M, N, K, EPI_BIAS, bias, (bf16_t*)out, N);
    g.trace = (long long*)trace;     // (may be null: then this is the plain launch with the ablation knobs below)
    g.dbg_skip_prefetch = getenv("MI355_DBG_SKIP_PREFETCH") ? 1 : 0;                       // ablations (results are garbage)
    if (getenv("MI355_DBG_MASK")) g.dbg_skip_prefetch = atoi(getenv("MI355_DBG_MASK"));   // bit 0: no K-loop prefetch, bit 1: no LDS fragment reads
    if (trace && g.dbg_skip_prefetch == 0) g.dbg_skip_prefetch = 16;                        // the trace lives in its own build of the kernel
    HIPCHK(launch_gemm(g, (hipStream_t)stream));
    return 0;
}

// measurement tool: n samples of {s_memtime, s_memrealtime} (int64 pairs) taken by one wave, ~sleep_iters * 8 k core clocks apart
extern "C" int mi355_clock_probe(void* stream, void* out, int n, int sleep_iters) {
    if (!out || n <= 0 || sleep_iters < 0) return fail("mi355_clock_probe: bad argument");
    HIPCHK(launch_clock_probe((long long*)out, n, sleep_iters, (hipStream_t)stream));
    return 0;
}

extern "C" int mi355_op_attention(void* stream, const void* q, const void* k, const void* vT, void* o_img, void* o_ctx, int B,
                                  int H, int S, int S_pad, int n_img) {
    if (!q || !k || !vT || !o_img) return fail("mi355_op_attention: null argument");
    if (n_img < S && !o_ctx) return fail("mi355_op_attention: o_ctx is NULL but S > n_img");
    AttnParams a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)vT, (bf16_t*)o_img, (bf16_t*)o_ctx, B, H, S, S_pad, n_img, 0,
                 g_attn_static >= 2 ? (float)g_attn_static : 0.f};
    HIPCHK(attn_p(a, (hipStream_t)stream));
    return 0;
}

extern "C" int mi355_op_ln_modulate(void* stream, const void* x, const void* shift, const void* scale, void* out, int M, int D,
                                    int rows_per_sample, float eps) {
    if (!x || !shift || !scale || !out) return fail("mi355_op_ln_modulate: null argument");
    LnModParams l;
    memset(&l, 0, sizeof(l));
    // shift / scale are separate [nb][D] tensors: address them relative to `shift`
    l.x = (const bf16_t*)x; l.out = (bf16_t*)out; l.out2 = nullptr; l.mod = (const bf16_t*)shift; l.mod_ld = D;
    l.shift_off = 0;
    const long delta = (const bf16_t*)scale - (const bf16_t*)shift;
    if (delta < -2147483647L || delta > 2147483647L) return fail("mi355_op_ln_modulate: shift/scale too far apart");
    l.scale_off = (int)delta;
    l.M = M; l.D = D; l.rows_per_sample = rows_per_sample; l.eps = eps;
    HIPCHK(lnmod_p(l, (hipStream_t)stream));
    return 0;
}

#include "engine_train.inc"
