// mi355_flow -- causal 3-D video VAE decode behind the C ABI (include/mi355_flow.h, mi355_wvae_*): the decode that closes a Wan
// rollout and, with one latent frame, a Qwen-Image rollout (reference Wan2_T2V_Adapter.decode_latents,
// models/wan/wan2_t2v.py:215-230 -> diffusers AutoencoderKLWan.decode + VideoProcessor.postprocess_video('pt');
// QwenImageAdapter.decode_latents, models/qwen_image/qwen_image.py:197-213 -> AutoencoderKLQwenImage.decode(...)[:, :, 0]).
//
// The published decoder runs one latent frame at a time through a feature cache; this engine evaluates the equivalent whole-sequence
// form (oracle/wan_vae_ref.py proves the two equal on CPU): activations are [B][T][H][W][C] bf16 (channels padded to a multiple of 64),
// every causal 3x3x3 convolution is ONE implicit GEMM on the MFMA kernel of gemm.hip over all frames (M = B*T*H*W pixels, N = C_out,
// K = 27*C_in; the A operand is gathered tap by tap, temporal taps reach back in time and read zeros before frame 0), the temporal
// upsampler keeps frame 0 and runs its (3,1,1) time_conv over frames 1.. as a sequence of their own, nearest-2x + Conv2d is the 2-D
// gather with the upsample folded in, RMS-norm + SiLU is one HBM pass (vae.hip), the mid-block attention is per frame as in
// vae_engine.hip.  With a single latent frame every causal conv only ever sees its last temporal slice: a second packed copy of the
// weights holding just that slice (9 taps) is used and the decode is 2-D end to end.
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/mi355_flow.h"
#include "engine_common.h"

using namespace mi355;

namespace {

inline int pad64(int c) { return (c + 63) / 64 * 64; }

// w: [copad][taps][cipad]; w_last (3x3x3 only): the taps of the LAST temporal slice, [copad][9][cipad]
struct WConv { bf16_t* w = nullptr; bf16_t* w_last = nullptr; float* b = nullptr; int co = 0, ci = 0, copad = 0, cipad = 0, taps = 0; };
struct WNorm { float* g = nullptr; int c = 0, cpad = 0; };
struct WRes { WNorm n1, n2; WConv c1, c2, sc; bool has_sc = false; int ci = 0, co = 0; };
struct WUp { std::vector<WRes> res; WConv resample, time_conv; int mode = 0; };     // mode 0 none, 2 upsample2d, 3 upsample3d

struct WSlot {
    void* dst; int kind;        // 0: fp32 vector copy, 1: conv repack, 2: causal 3x3x3 repack (+ last-slice copy)
    int64_t numel; int co, ci, cipad, taps; void* dst2; bool bound;
};

}  // namespace

struct mi355_wvae {
    mi355_wvae_cfg cfg;
    int dims[5];
    char* arena = nullptr;
    size_t used = 0, cap = 0;
    float *w_pq = nullptr, *b_pq = nullptr;
    WConv conv_in, conv_out, to_qkv, proj;
    WNorm attn_norm, norm_out;
    WRes mid[2];
    std::vector<WUp> up;
    bf16_t* zero_page = nullptr;
    float* zero_bias = nullptr;
    std::map<std::string, WSlot> slots;
    std::vector<std::string> names;

    void* take(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        char* p = arena ? arena + used : nullptr;
        used += bytes;
        return p;
    }
    void reg(const std::string& n, void* dst, int kind, int64_t numel, int co = 0, int ci = 0, int cipad = 0, int taps = 0, void* dst2 = nullptr) {
        if (!arena) return;
        slots[n] = WSlot{dst, kind, numel, co, ci, cipad, taps, dst2, false};
        names.push_back(n);
    }
    WConv conv(const std::string& n, int co, int ci, int taps) {
        WConv c; c.co = co; c.ci = ci; c.copad = pad64(co); c.cipad = pad64(ci); c.taps = taps;
        c.w = (bf16_t*)take((size_t)c.copad * taps * c.cipad * 2);
        if (taps == 27) c.w_last = (bf16_t*)take((size_t)c.copad * 9 * c.cipad * 2);
        c.b = (float*)take((size_t)c.copad * 4);
        reg(n + ".weight", c.w, taps == 27 ? 2 : 1, (int64_t)co * ci * taps, co, ci, c.cipad, taps, c.w_last);
        reg(n + ".bias", c.b, 0, co);
        return c;
    }
    WNorm norm(const std::string& n, int c) {
        WNorm w; w.c = c; w.cpad = pad64(c);
        w.g = (float*)take((size_t)w.cpad * 4);
        reg(n + ".gamma", w.g, 0, c);
        return w;
    }
    WRes resnet(const std::string& n, int ci, int co) {
        WRes r; r.ci = ci; r.co = co;
        r.n1 = norm(n + ".norm1", ci); r.c1 = conv(n + ".conv1", co, ci, 27);
        r.n2 = norm(n + ".norm2", co); r.c2 = conv(n + ".conv2", co, co, 27);
        r.has_sc = ci != co;
        if (r.has_sc) r.sc = conv(n + ".conv_shortcut", co, ci, 1);
        return r;
    }
    void layout();
};

// parameter names = diffusers AutoencoderKLWan.state_dict() (decoder half + post_quant_conv)
void mi355_wvae::layout() {
    used = 0; slots.clear(); names.clear(); up.clear();
    const int top = dims[0];
    zero_page = (bf16_t*)take(256);
    zero_bias = (float*)take((size_t)pad64(3 * top) * 4);
    w_pq = (float*)take((size_t)cfg.z_dim * cfg.z_dim * 4); b_pq = (float*)take((size_t)cfg.z_dim * 4);
    reg("post_quant_conv.weight", w_pq, 0, (int64_t)cfg.z_dim * cfg.z_dim);
    reg("post_quant_conv.bias", b_pq, 0, cfg.z_dim);
    conv_in = conv("decoder.conv_in", top, cfg.z_dim, 27);
    mid[0] = resnet("decoder.mid_block.resnets.0", top, top);
    const std::string at = "decoder.mid_block.attentions.0";
    attn_norm = norm(at + ".norm", top);
    to_qkv = conv(at + ".to_qkv", 3 * top, top, 1);
    proj = conv(at + ".proj", top, top, 1);
    mid[1] = resnet("decoder.mid_block.resnets.1", top, top);
    for (int i = 0; i < 4; ++i) {
        int ci = dims[i], co = dims[i + 1];
        if (i > 0) ci /= 2;
        WUp u;
        const std::string pre = "decoder.up_blocks." + std::to_string(i);
        for (int j = 0; j < cfg.num_res_blocks + 1; ++j) u.res.push_back(resnet(pre + ".resnets." + std::to_string(j), j == 0 ? ci : co, co));
        u.mode = i == 3 ? 0 : (cfg.temporal_upsample[i] ? 3 : 2);
        if (u.mode) u.resample = conv(pre + ".upsamplers.0.resample.1", co / 2, co, 9);
        if (u.mode == 3) u.time_conv = conv(pre + ".upsamplers.0.time_conv", 2 * co, co, 3);
        up.push_back(u);
    }
    norm_out = norm("decoder.norm_out", dims[4]);
    conv_out = conv("decoder.conv_out", cfg.out_channels, dims[4], 27);
}

extern "C" int mi355_wvae_create(const mi355_wvae_cfg* cfg, mi355_wvae** out) {
    if (!cfg || !out) return errorf("mi355_wvae_create: null argument");
    if (cfg->z_dim < 1 || cfg->z_dim > 16) return errorf("mi355_wvae_create: z_dim must be 1..16");
    if (cfg->out_channels < 1 || cfg->out_channels > 4) return errorf("mi355_wvae_create: out_channels must be 1..4");
    if (cfg->num_res_blocks < 1 || cfg->num_res_blocks > 8) return errorf("mi355_wvae_create: num_res_blocks out of range");
    if (cfg->base_dim < 8 || cfg->base_dim % 8) return errorf("mi355_wvae_create: base_dim must be a positive multiple of 8");
    mi355_wvae* v = new mi355_wvae();
    v->cfg = *cfg;
    v->dims[0] = cfg->base_dim * cfg->dim_mult[3];
    for (int i = 0; i < 4; ++i) v->dims[i + 1] = cfg->base_dim * cfg->dim_mult[3 - i];
    if (v->dims[0] % 64) {
        int r = errorf("mi355_wvae_create: the mid-block width %d must be a multiple of 64 (attention head dim = GEMM K)", v->dims[0]);
        delete v;
        return r;
    }
    for (int i = 0; i < 3; ++i)
        if (cfg->temporal_upsample[i] && v->dims[i + 1] % 64) {
            int r = errorf("mi355_wvae_create: temporal upsampling at width %d needs a multiple of 64", v->dims[i + 1]);
            delete v;
            return r;
        }
    for (int i = 0; i < 5; ++i)
        if (v->dims[i] < 8 || v->dims[i] % 8 || v->dims[i] > 1024 || (i > 0 && i < 4 && (v->dims[i] / 2) % 8)) {
            int r = errorf("mi355_wvae_create: stage width %d unsupported (need multiples of 16, <= 1024)", v->dims[i]);
            delete v;
            return r;
        }
    v->layout();
    v->cap = v->used;
    if (hipMalloc((void**)&v->arena, v->cap) != hipSuccess) {
        int r = errorf("mi355_wvae_create: hipMalloc of %zu bytes failed", v->cap);
        delete v;
        return r;
    }
    if (hipMemset(v->arena, 0, v->cap) != hipSuccess) {      // padded weight rows / channels / biases / gammas are zero
        (void)hipFree(v->arena);
        delete v;
        return errorf("mi355_wvae_create: hipMemset failed");
    }
    v->layout();
    *out = v;
    return 0;
}

extern "C" int mi355_wvae_destroy(mi355_wvae* v) {
    if (!v) return 0;
    if (v->arena) (void)hipFree(v->arena);
    delete v;
    return 0;
}
extern "C" int mi355_wvae_num_params(mi355_wvae* v) { return v ? (int)v->names.size() : 0; }
extern "C" const char* mi355_wvae_param_name(mi355_wvae* v, int i) {
    if (!v || i < 0 || i >= (int)v->names.size()) return nullptr;
    return v->names[i].c_str();
}

extern "C" int mi355_wvae_bind_weight(mi355_wvae* v, const char* name, const void* src, int dtype, int ndim, const int64_t* shape,
                                      void* stream) {
    if (!v || !name || !src) return errorf("mi355_wvae_bind_weight: null argument");
    auto it = v->slots.find(name);
    if (it == v->slots.end()) return errorf("mi355_wvae_bind_weight: unknown parameter '%s'", name);
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    WSlot& s = it->second;
    if (n != s.numel)
        return errorf("mi355_wvae_bind_weight: '%s' has %lld elements, expected %lld", name, (long long)n, (long long)s.numel);
    if (dtype < 0 || dtype > 2) return errorf("mi355_wvae_bind_weight: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    if (s.kind == 0) HIPCHK(launch_convert(src, dtype, s.dst, DT_F32, n, st));
    else {
        if (ndim >= 2 && (shape[0] != s.co || shape[1] != s.ci))
            return errorf("mi355_wvae_bind_weight: '%s' is [%lld][%lld]..., expected [%d][%d]...", name, (long long)shape[0],
                          (long long)shape[1], s.co, s.ci);
        HIPCHK(launch_conv_repack(src, dtype, (bf16_t*)s.dst, s.co, s.ci, s.cipad, s.taps, st));
        if (s.kind == 2) {      // last temporal slice = taps 18..26 of every output row
            const size_t row27 = (size_t)27 * s.cipad * 2, row9 = (size_t)9 * s.cipad * 2;
            HIPCHK(hipMemcpy2DAsync(s.dst2, row9, (const char*)s.dst + (size_t)18 * s.cipad * 2, row27, row9, (size_t)s.co,
                                    hipMemcpyDeviceToDevice, st));
        }
    }
    s.bound = true;
    return 0;
}

extern "C" int mi355_wvae_weights_ready(mi355_wvae* v) {
    if (!v) return errorf("null vae");
    for (auto& kv : v->slots)
        if (!kv.second.bound) return errorf("parameter '%s' has not been bound", kv.first.c_str());
    return 0;
}

// -------------------------------------------------------------------------------------- plan
struct mi355_wvae_plan {
    mi355_wvae* v;
    int B, T, h, w;
    long S, S_pad;
    char* ws = nullptr;
    size_t ws_bytes = 0;
    bf16_t *X, *Y, *T1, *T2, *P, *VT;
    float* SC;
};

extern "C" int mi355_wvae_plan_create(mi355_wvae* v, int max_batch, int latent_t, int latent_h, int latent_w, mi355_wvae_plan** out) {
    if (!v || !out) return errorf("mi355_wvae_plan_create: null argument");
    if (max_batch < 1 || latent_t < 1 || latent_h < 1 || latent_w < 1) return errorf("mi355_wvae_plan_create: bad shape");
    const long S = (long)latent_h * latent_w;
    if (S % 8) return errorf("mi355_wvae_plan_create: latent_h*latent_w = %ld must be a multiple of 8 (mid-block attention)", S);
    // largest activation any stage holds, in elements per sample
    size_t act = (size_t)latent_t * S * pad64(3 * v->dims[0]);      // q|k|v projection
    {
        long frames = latent_t, hw = S;
        for (int i = 0; i < 4; ++i) {
            const WUp& u = v->up[i];
            const int ci = u.res[0].ci, co = u.res[0].co;
            size_t a = (size_t)frames * hw * pad64(ci > co ? ci : co);
            if (a > act) act = a;
            if (u.mode == 3 && frames > 1) {
                a = (size_t)(frames - 1) * hw * pad64(2 * co);       // time_conv output
                if (a > act) act = a;
                frames = 2 * frames - 1;
                a = (size_t)frames * hw * pad64(co);
                if (a > act) act = a;
            }
            if (u.mode) {
                hw *= 4;
                a = (size_t)frames * hw * pad64(co / 2);
                if (a > act) act = a;
            }
        }
    }
    mi355_wvae_plan* p = new mi355_wvae_plan();
    p->v = v; p->B = max_batch; p->T = latent_t; p->h = latent_h; p->w = latent_w;
    p->S = S; p->S_pad = (S + 63) / 64 * 64;
    const int top = v->dims[0];
    const size_t buf = ((size_t)max_batch * act * 2 + 255) & ~(size_t)255;
    const size_t sc = ((size_t)S * p->S_pad * 4 + 255) & ~(size_t)255, pb = ((size_t)S * p->S_pad * 2 + 255) & ~(size_t)255;
    const size_t vt = ((size_t)pad64(top) * p->S_pad * 2 + 255) & ~(size_t)255;
    p->ws_bytes = 4 * buf + sc + pb + vt;
    if (hipMalloc((void**)&p->ws, p->ws_bytes) != hipSuccess) {
        int r = errorf("mi355_wvae_plan_create: hipMalloc of %zu bytes failed", p->ws_bytes);
        delete p;
        return r;
    }
    char* q = p->ws;
    p->X = (bf16_t*)q; q += buf; p->Y = (bf16_t*)q; q += buf; p->T1 = (bf16_t*)q; q += buf; p->T2 = (bf16_t*)q; q += buf;
    p->SC = (float*)q; q += sc; p->P = (bf16_t*)q; q += pb; p->VT = (bf16_t*)q; q += vt;
    if (hipMemset(p->VT, 0, vt) != hipSuccess) {       // key columns S..S_pad of V^T stay zero (the softmax writes zero probabilities there)
        (void)hipFree(p->ws);
        delete p;
        return errorf("mi355_wvae_plan_create: hipMemset failed");
    }
    *out = p;
    return 0;
}

extern "C" int mi355_wvae_plan_destroy(mi355_wvae_plan* p) {
    if (!p) return 0;
    if (p->ws) (void)hipFree(p->ws);
    delete p;
    return 0;
}
extern "C" int64_t mi355_wvae_plan_workspace_bytes(mi355_wvae_plan* p) { return p ? (int64_t)p->ws_bytes : 0; }

// ------------------------------------------------------------------------------------ decode
namespace {

struct Ctx {
    mi355_wvae_plan* p; hipStream_t st; int B;
    bf16_t *X, *Y, *T1, *T2;
};

// convolution over `frames` frames per sample of H x W (OUTPUT size; up = 1: the input frames are H/2 x W/2):
//   cw.taps 27: causal 3x3x3 (last-slice 3x3 when the sample has a single frame), 9: per-frame 3x3, 3: (3,1,1) temporal, 1: 1x1x1.
// frames_in = frames per sample in the input buffer (>= frames; `in` may be advanced by whole frames).  res: residual added (may alias out).
int conv(const Ctx& c, const WConv& cw, const bf16_t* in, bf16_t* out, int frames, int frames_in, int H, int W, int up, const bf16_t* res) {
    const long M = (long)c.B * frames * H * W;
    if (M > 0x7fffffffL) return errorf("mi355_wvae_decode: %ld pixels exceed the GEMM row range (decode fewer samples per call)", M);
    if (cw.taps == 1) {
        GemmParams g = make_gemm(in, cw.cipad, cw.w, cw.cipad, M, cw.copad, cw.cipad, res ? EPI_POSADD : EPI_BIAS, cw.b, out, cw.copad);
        g.aux = res; g.ld_aux = cw.copad;
        HIPCHK(launch_gemm(g, c.st));
        return 0;
    }
    int kt = 1, ks = 3;
    const bf16_t* w = cw.w;
    if (cw.taps == 27) {
        if (frames == 1 && frames_in == 1) w = cw.w_last;      // only the last temporal slice ever meets data
        else kt = 3;
    } else if (cw.taps == 3) { kt = 3; ks = 1; }
    const int K = kt * ks * ks * cw.cipad;
    GemmParams g = make_gemm(in, cw.cipad, w, K, M, cw.copad, K, res ? EPI_POSADD : EPI_BIAS, cw.b, out, cw.copad);
    g.conv_cin = cw.cipad; g.conv_h = H; g.conv_w = W; g.conv_up = up; g.zero_page = c.p->v->zero_page;
    g.conv_t = frames; g.conv_t_in = frames_in; g.conv_kt = kt; g.conv_ks = ks;
    g.aux = res; g.ld_aux = cw.copad;
    HIPCHK(launch_gemm(g, c.st));
    return 0;
}

int rms(const Ctx& c, const WNorm& n, const bf16_t* x, bf16_t* y, long M, bool silu) {
    HIPCHK(launch_wan_rms(x, y, n.g, M, n.c, n.cpad, silu, c.st));
    return 0;
}

// x in c.X on entry ([M][pad64(r.ci)]) and on exit ([M][pad64(r.co)])
int resnet(Ctx& c, const WRes& r, int frames, int H, int W) {
    const long M = (long)c.B * frames * H * W;
    CHK(rms(c, r.n1, c.X, c.T1, M, true));
    CHK(conv(c, r.c1, c.T1, c.T2, frames, frames, H, W, 0, nullptr));
    CHK(rms(c, r.n2, c.T2, c.T1, M, true));
    if (r.has_sc) {
        CHK(conv(c, r.sc, c.X, c.T2, frames, frames, H, W, 0, nullptr));
        CHK(conv(c, r.c2, c.T1, c.Y, frames, frames, H, W, 0, c.T2));
        bf16_t* t = c.X; c.X = c.Y; c.Y = t;
    } else {
        CHK(conv(c, r.c2, c.T1, c.X, frames, frames, H, W, 0, c.X));
    }
    return 0;
}

// WanAttentionBlock: per frame, single head of dim C over the S = h*w positions; scores materialised in fp32 (vae_engine.hip mid_attention)
int mid_attention(Ctx& c, int frames) {
    mi355_wvae_plan* p = c.p;
    mi355_wvae* v = p->v;
    const int C = v->attn_norm.c, Cp = v->attn_norm.cpad;
    const long S = p->S, Sp = p->S_pad, M = (long)c.B * frames * S;
    CHK(rms(c, v->attn_norm, c.X, c.T1, M, false));
    // q | k (rows 0..2C of to_qkv) in one GEMM; V^T per frame with the operands swapped (rows 2C..3C)
    const int QK = 2 * C;
    GemmParams gqk = make_gemm(c.T1, Cp, v->to_qkv.w, v->to_qkv.cipad, M, QK, Cp, EPI_BIAS, v->to_qkv.b, c.T2, QK);
    HIPCHK(launch_gemm(gqk, c.st));
    const float scale = 1.0f / sqrtf((float)C);
    for (long f = 0; f < (long)c.B * frames; ++f) {
        const bf16_t* hn = c.T1 + (size_t)f * S * Cp;
        const bf16_t* q = c.T2 + (size_t)f * S * QK;
        GemmParams gv = make_gemm(v->to_qkv.w + (size_t)QK * v->to_qkv.cipad, v->to_qkv.cipad, hn, Cp, C, (int)S, Cp, EPI_BIAS_ROW,
                                  v->to_qkv.b + QK, p->VT, Sp);
        HIPCHK(launch_gemm(gv, c.st));
        // q / k hold C real columns; K of the scores GEMM is C rounded up to 64 -- when C % 64 != 0 the extra columns belong to k / the
        // next row, so widths that are not multiples of 64 are rejected at create time for the attention (dims[0] % 64 == 0)
        GemmParams gs = make_gemm(q, QK, q + C, QK, S, (int)S, C, EPI_F32, nullptr, nullptr, Sp);
        gs.out_f32 = p->SC; gs.q_scale = 1.0f;
        HIPCHK(launch_gemm(gs, c.st));
        HIPCHK(launch_softmax_rows_ld(p->SC, Sp, p->P, Sp, S, (int)S, scale, c.st));
        GemmParams go = make_gemm(p->P, Sp, p->VT, Sp, S, C, (int)Sp, EPI_BIAS, v->zero_bias, c.Y + (size_t)f * S * Cp, Cp);
        HIPCHK(launch_gemm(go, c.st));
    }
    GemmParams gout = make_gemm(c.Y, Cp, v->proj.w, v->proj.cipad, M, v->proj.copad, Cp, EPI_POSADD, v->proj.b, c.X, Cp);
    gout.aux = c.X; gout.ld_aux = Cp;
    HIPCHK(launch_gemm(gout, c.st));
    return 0;
}

}  // namespace

// latents: (batch, z_dim, T, h, w) in lat_dtype.  denormalise != 0: z = latents / (1 / std) + mean first (what the adapters do before
// vae.decode).  video: [batch][F][out_channels][8h][8w] (F = 1 + 4 (T - 1)), fp32 (0) or bf16 (1); postprocess != 0: (x / 2 + 0.5) clamped
// to [0, 1] (VideoProcessor.postprocess_video(..., 'pt') layout and range), else the raw decoder output clamped to [-1, 1].
extern "C" int mi355_wvae_decode(mi355_wvae_plan* p, void* stream, const void* latents, int lat_dtype, int batch, void* video,
                                 int vid_dtype, int postprocess, int denormalise) {
    if (!p || !latents || !video) return errorf("mi355_wvae_decode: null argument");
    if (batch < 1 || batch > p->B) return errorf("mi355_wvae_decode: batch %d outside the plan's 1..%d", batch, p->B);
    if (lat_dtype < 0 || lat_dtype > 2) return errorf("mi355_wvae_decode: bad latent dtype %d", lat_dtype);
    if (vid_dtype != DT_F32 && vid_dtype != DT_BF16) return errorf("mi355_wvae_decode: video dtype must be fp32 (0) or bf16 (1)");
    mi355_wvae* v = p->v;
    CHK(mi355_wvae_weights_ready(v));
    const mi355_wvae_cfg& cfg = v->cfg;
    Ctx c{p, (hipStream_t)stream, batch, p->X, p->Y, p->T1, p->T2};
    int H = p->h, W = p->w, frames = p->T;
    WvaeIngestParams q;
    memset(&q, 0, sizeof(q));
    q.B = batch; q.C = cfg.z_dim; q.T = p->T; q.Cpad = v->conv_in.cipad; q.denorm = denormalise; q.HW = p->S;
    for (int i = 0; i < cfg.z_dim; ++i) { q.mean[i] = cfg.latents_mean[i]; q.std[i] = cfg.latents_std[i]; }
    q.w_pq = v->w_pq; q.b_pq = v->b_pq;
    HIPCHK(launch_wvae_ingest(latents, lat_dtype, c.T1, q, c.st));
    CHK(conv(c, v->conv_in, c.T1, c.X, frames, frames, H, W, 0, nullptr));
    CHK(resnet(c, v->mid[0], frames, H, W));
    CHK(mid_attention(c, frames));
    CHK(resnet(c, v->mid[1], frames, H, W));
    for (int i = 0; i < 4; ++i) {
        const WUp& u = v->up[i];
        for (const WRes& r : u.res) CHK(resnet(c, r, frames, H, W));
        if (u.mode == 3 && frames > 1) {
            // frames 1.. through time_conv as their own causal sequence, then interleave behind the untouched frame 0
            const int co = u.res[0].co, cp = pad64(co);
            const long hw = (long)H * W;
            CHK(conv(c, u.time_conv, c.X + (size_t)hw * cp, c.T2, frames - 1, frames, H, W, 0, nullptr));
            HIPCHK(launch_frame_interleave(c.X, c.T2, c.T1, batch, frames, hw, cp, c.st));
            frames = 2 * frames - 1;
            bf16_t* t = c.X; c.X = c.T1; c.T1 = t;
        }
        if (u.mode) {
            H *= 2; W *= 2;
            CHK(conv(c, u.resample, c.X, c.Y, frames, frames, H, W, 1, nullptr));
            bf16_t* t = c.X; c.X = c.Y; c.Y = t;
        }
    }
    const long M = (long)batch * frames * H * W;
    CHK(rms(c, v->norm_out, c.X, c.T1, M, true));
    {
        const WConv& cw = v->conv_out;
        const bool one = frames == 1;
        const int kt = one ? 1 : 3, K = kt * 9 * cw.cipad;
        if (M > 0x7fffffffL) return errorf("mi355_wvae_decode: %ld pixels exceed the GEMM row range", M);
        GemmParams g = make_gemm(c.T1, cw.cipad, one ? cw.w_last : cw.w, K, M, cw.co, K, EPI_IMG, cw.b, nullptr, 0);
        g.conv_cin = cw.cipad; g.conv_h = H; g.conv_w = W; g.zero_page = v->zero_page;
        g.conv_t = frames; g.conv_t_in = frames; g.conv_kt = kt; g.conv_ks = 3;
        g.img_post = postprocess;
        g.img_clamp = 1;
        if (vid_dtype == DT_F32) g.out_f32 = (float*)video; else g.out = (bf16_t*)video;
        HIPCHK(launch_gemm(g, c.st));
    }
    return 0;
}

// ----------------------------------------------------------------------- operator-level API (unit tests)
static bf16_t* g_wvae_zero = nullptr;

// causal convolution over x [B][T_in][H>>up][W>>up][Cin] bf16 (Cin % 64 == 0): kt temporal taps reaching back in time (zeros before
// frame 0), ks x ks spatial taps with zero padding; w_packed [Cout][kt*ks*ks][Cin] (mi355_op_conv_repack, taps = kt*ks*ks);
// x may be advanced by whole frames with T_in > T.  out [B*T*H*W][Cout] (+ residual, may alias out)
extern "C" int mi355_op_conv3d_causal(void* stream, const void* x, const void* w_packed, const float* bias, const void* residual, void* out,
                                      int B, int T, int T_in, int H, int W, int Cin, int Cout, int kt, int ks, int upsample) {
    if (!x || !w_packed || !bias || !out) return errorf("mi355_op_conv3d_causal: null argument");
    if (Cin % 64) return errorf("mi355_op_conv3d_causal: Cin must be a multiple of 64 (pad the channels)");
    if (!g_wvae_zero) {
        HIPCHK(hipMalloc((void**)&g_wvae_zero, 256));
        HIPCHK(hipMemset(g_wvae_zero, 0, 256));
    }
    const long M = (long)B * T * H * W;
    const int K = kt * ks * ks * Cin;
    GemmParams g = make_gemm((const bf16_t*)x, Cin, (const bf16_t*)w_packed, K, M, Cout, K, residual ? EPI_POSADD : EPI_BIAS, bias,
                             (bf16_t*)out, Cout);
    g.conv_cin = Cin; g.conv_h = H; g.conv_w = W; g.conv_up = upsample; g.zero_page = g_wvae_zero;
    g.conv_t = T; g.conv_t_in = T_in; g.conv_kt = kt; g.conv_ks = ks;
    g.aux = (const bf16_t*)residual; g.ld_aux = Cout;
    HIPCHK(launch_gemm(g, (hipStream_t)stream));
    return 0;
}

// WanRMS_norm (+ SiLU) over rows of C_pad channels of which the first C are real (gamma fp32 [C_pad], zero beyond C)
extern "C" int mi355_op_wan_rms(void* stream, const void* x, const float* gamma, void* out, int64_t rows, int C, int C_pad, int silu) {
    if (!x || !gamma || !out) return errorf("mi355_op_wan_rms: null argument");
    HIPCHK(launch_wan_rms((const bf16_t*)x, (bf16_t*)out, gamma, (long)rows, C, C_pad, silu != 0, (hipStream_t)stream));
    return 0;
}
