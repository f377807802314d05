// mi355_flow -- VAE decode behind the C ABI (include/mi355_flow.h, mi355_vae_*): the image decode that closes a rollout
// (reference SD3_5Adapter.decode_latents, sd3_5.py:161-172 ->
// diffusers AutoencoderKL.decode + VaeImageProcessor.postprocess('pt')).
//
// Activations are NHWC bf16 ([B*H*W][C] row-major), so every 3x3 convolution is one implicit GEMM on the MFMA kernel of
// gemm.hip (M = B*H*W pixels, N = C_out, K = 9*C_in, A-operand gathered tap by tap with zero padding; the nearest-2x
// upsample of Upsample2D is folded into the gather), 1x1 shortcuts and the attention projections are plain GEMMs, residual
// adds are fused into the conv / projection epilogue, GroupNorm+SiLU is the 3-pass HBM-bound kernel of vae.hip.
// Host code only launches kernels on the caller's stream.
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/mi355_flow.h"
#include "engine_common.h"

using namespace mi355;


namespace {

struct ConvW { bf16_t* w = nullptr; float* b = nullptr; int co = 0, ci = 0, cipad = 0, taps = 0; };
struct NormW { float* g = nullptr; float* b = nullptr; int c = 0; };
struct Resnet { NormW n1, n2; ConvW c1, c2, sc; bool has_sc = false; int ci = 0, co = 0; };

struct VSlot {
    void* dst; int kind;        // 0: fp32 vector copy, 1: conv / linear repack
    int64_t numel; int co, ci, cipad, taps; bool bound;
};

inline int pad64(int c) { return (c + 63) / 64 * 64; }

}  // namespace

struct mi355_vae {
    mi355_vae_cfg cfg;
    char* arena = nullptr;
    size_t used = 0, cap = 0;
    ConvW conv_in, conv_out, to_qk, to_v, to_out;
    NormW attn_gn, norm_out;
    Resnet mid[2];
    std::vector<std::vector<Resnet>> up;
    std::vector<ConvW> upconv;     // co == 0: no upsampler
    bf16_t* zero_page = nullptr;
    float* zero_bias = nullptr;
    std::map<std::string, VSlot> slots;
    std::vector<std::string> names;

    void* take(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        char* p = arena ? arena + used : nullptr;
        used += bytes;
        return p;
    }
    void reg(const std::string& n, void* dst, int kind, int64_t numel, int co = 0, int ci = 0, int cipad = 0, int taps = 0) {
        if (!arena) return;
        slots[n] = VSlot{dst, kind, numel, co, ci, cipad, taps, false};
        names.push_back(n);
    }
    ConvW conv(const std::string& n, int co, int ci, int taps) {
        ConvW c; c.co = co; c.ci = ci; c.cipad = pad64(ci); c.taps = taps;
        c.w = (bf16_t*)take((size_t)co * taps * c.cipad * 2);
        c.b = (float*)take((size_t)co * 4);
        reg(n + ".weight", c.w, 1, (int64_t)co * ci * taps, co, ci, c.cipad, taps);
        reg(n + ".bias", c.b, 0, co);
        return c;
    }
    NormW norm(const std::string& n, int c) {
        NormW w; w.c = c;
        w.g = (float*)take((size_t)c * 4); w.b = (float*)take((size_t)c * 4);
        reg(n + ".weight", w.g, 0, c); reg(n + ".bias", w.b, 0, c);
        return w;
    }
    Resnet resnet(const std::string& n, int ci, int co) {
        Resnet r; r.ci = ci; r.co = co;
        r.n1 = norm(n + ".norm1", ci); r.c1 = conv(n + ".conv1", co, ci, 9);
        r.n2 = norm(n + ".norm2", co); r.c2 = conv(n + ".conv2", co, co, 9);
        r.has_sc = ci != co;
        if (r.has_sc) r.sc = conv(n + ".conv_shortcut", co, ci, 1);
        return r;
    }
    void layout();
};

void mi355_vae::layout() {
    used = 0; slots.clear(); names.clear(); up.clear(); upconv.clear();
    const int nb = cfg.num_blocks;
    const int top = cfg.block_out_channels[nb - 1];
    zero_page = (bf16_t*)take(256);
    zero_bias = (float*)take((size_t)top * 4);
    conv_in = conv("decoder.conv_in", top, cfg.latent_channels, 9);
    mid[0] = resnet("decoder.mid_block.resnets.0", top, top);
    const std::string at = "decoder.mid_block.attentions.0";
    attn_gn = norm(at + ".group_norm", top);
    // to_q | to_k share one [2*top][top] weight: a single projection GEMM, the scores GEMM reads both halves strided
    to_qk.co = 2 * top; to_qk.ci = to_qk.cipad = top; to_qk.taps = 1;
    to_qk.w = (bf16_t*)take((size_t)2 * top * top * 2);
    to_qk.b = (float*)take((size_t)2 * top * 4);
    reg(at + ".to_q.weight", to_qk.w, 1, (int64_t)top * top, top, top, top, 1);
    reg(at + ".to_q.bias", to_qk.b, 0, top);
    reg(at + ".to_k.weight", to_qk.w + (size_t)top * top, 1, (int64_t)top * top, top, top, top, 1);
    reg(at + ".to_k.bias", to_qk.b + top, 0, top);
    to_v = conv(at + ".to_v", top, top, 1);
    to_out = conv(at + ".to_out.0", top, top, 1);
    mid[1] = resnet("decoder.mid_block.resnets.1", top, top);
    int prev = top;
    for (int i = 0; i < nb; ++i) {
        const int co = cfg.block_out_channels[nb - 1 - i];
        std::vector<Resnet> rs;
        for (int j = 0; j < cfg.layers_per_block + 1; ++j)
            rs.push_back(resnet("decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? prev : co, co));
        up.push_back(rs);
        upconv.push_back(i != nb - 1 ? conv("decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", co, co, 9) : ConvW());
        prev = co;
    }
    norm_out = norm("decoder.conv_norm_out", prev);
    conv_out = conv("decoder.conv_out", cfg.out_channels, prev, 9);
}

extern "C" int mi355_vae_create(const mi355_vae_cfg* cfg, mi355_vae** out) {
    if (!cfg || !out) return errorf("mi355_vae_create: null argument");
    if (cfg->num_blocks < 1 || cfg->num_blocks > 8) return errorf("mi355_vae_create: num_blocks out of range");
    if (cfg->out_channels < 1 || cfg->out_channels > 4) return errorf("mi355_vae_create: out_channels must be 1..4");
    if (cfg->latent_channels < 1 || cfg->latent_channels > 64) return errorf("mi355_vae_create: latent_channels must be 1..64");
    for (int i = 0; i < cfg->num_blocks; ++i) {
        const int c = cfg->block_out_channels[i];
        if (c % 64 || c > 2048 || 256 % (c / 8) || c % cfg->norm_num_groups)
            return errorf("mi355_vae_create: block_out_channels[%d] = %d unsupported (need a multiple of 64 dividing 2048, divisible "
                          "by norm_num_groups)", i, c);
    }
    mi355_vae* v = new mi355_vae();
    v->cfg = *cfg;
    v->layout();
    v->cap = v->used;
    if (hipMalloc((void**)&v->arena, v->cap) != hipSuccess) {
        int r = errorf("mi355_vae_create: hipMalloc of %zu bytes failed", v->cap);
        delete v;
        return r;
    }
    if (hipMemset(v->arena, 0, v->cap) != hipSuccess) {
        (void)hipFree(v->arena);
        delete v;
        return errorf("mi355_vae_create: hipMemset failed");
    }
    v->layout();
    *out = v;
    return 0;
}

extern "C" int mi355_vae_destroy(mi355_vae* v) {
    if (!v) return 0;
    if (v->arena) (void)hipFree(v->arena);
    delete v;
    return 0;
}

extern "C" int mi355_vae_num_params(mi355_vae* v) { return v ? (int)v->names.size() : 0; }
extern "C" const char* mi355_vae_param_name(mi355_vae* v, int i) {
    if (!v || i < 0 || i >= (int)v->names.size()) return nullptr;
    return v->names[i].c_str();
}

extern "C" int mi355_vae_bind_weight(mi355_vae* v, const char* name, const void* src, int dtype, int ndim, const int64_t* shape,
                                     void* stream) {
    if (!v || !name || !src) return errorf("mi355_vae_bind_weight: null argument");
    auto it = v->slots.find(name);
    if (it == v->slots.end()) return errorf("mi355_vae_bind_weight: unknown parameter '%s'", name);
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    VSlot& s = it->second;
    if (n != s.numel)
        return errorf("mi355_vae_bind_weight: '%s' has %lld elements, expected %lld", name, (long long)n, (long long)s.numel);
    if (dtype < 0 || dtype > 2) return errorf("mi355_vae_bind_weight: bad dtype %d", dtype);
    if (s.kind == 0) HIPCHK(launch_convert(src, dtype, s.dst, DT_F32, n, (hipStream_t)stream));
    else {
        if (ndim >= 2 && (shape[0] != s.co || shape[1] != s.ci))
            return errorf("mi355_vae_bind_weight: '%s' is [%lld][%lld]..., expected [%d][%d]...", name, (long long)shape[0],
                          (long long)shape[1], s.co, s.ci);
        HIPCHK(launch_conv_repack(src, dtype, (bf16_t*)s.dst, s.co, s.ci, s.cipad, s.taps, (hipStream_t)stream));
    }
    s.bound = true;
    return 0;
}

extern "C" int mi355_vae_weights_ready(mi355_vae* v) {
    if (!v) return errorf("null vae");
    for (auto& kv : v->slots)
        if (!kv.second.bound) return errorf("parameter '%s' has not been bound", kv.first.c_str());
    return 0;
}

// -------------------------------------------------------------------------------------- plan
struct mi355_vae_plan {
    mi355_vae* v;
    int B, h, w;
    char* ws = nullptr;
    size_t ws_bytes = 0;
    bf16_t *X, *Y, *T1, *T2, *P, *VT;
    float *SC, *part, *ad;
};

extern "C" int mi355_vae_plan_create(mi355_vae* v, int max_batch, int latent_h, int latent_w, mi355_vae_plan** out) {
    if (!v || !out) return errorf("mi355_vae_plan_create: null argument");
    if (max_batch < 1 || latent_h < 1 || latent_w < 1) return errorf("mi355_vae_plan_create: bad shape");
    const long S = (long)latent_h * latent_w;
    if (S % 64) return errorf("mi355_vae_plan_create: latent_h*latent_w = %ld must be a multiple of 64 (mid-block attention GEMM K)", S);
    const mi355_vae_cfg& c = v->cfg;
    const int nb = c.num_blocks, top = c.block_out_channels[nb - 1];
    // largest activation ([pixels][channels]) any stage holds
    size_t act = (size_t)S * (2 * top);   // q|k projection
    long hw = S; int ch = top;
    for (int i = 0; i < nb; ++i) {
        const int co = c.block_out_channels[nb - 1 - i];
        if ((size_t)hw * ch > act) act = (size_t)hw * ch;       // GN of the first resnet's input (ch may exceed co)
        ch = co;
        if (i != nb - 1) hw *= 4;
        if ((size_t)hw * ch > act) act = (size_t)hw * ch;
    }
    int cmax = 0;
    for (int i = 0; i < nb; ++i) cmax = c.block_out_channels[i] > cmax ? c.block_out_channels[i] : cmax;
    mi355_vae_plan* p = new mi355_vae_plan();
    p->v = v; p->B = max_batch; p->h = latent_h; p->w = latent_w;
    const size_t buf = ((size_t)max_batch * act * 2 + 255) & ~(size_t)255;
    const size_t sc = ((size_t)S * S * 4 + 255) & ~(size_t)255, pb = ((size_t)S * S * 2 + 255) & ~(size_t)255;
    const size_t vt = ((size_t)top * S * 2 + 255) & ~(size_t)255;
    const size_t part = ((size_t)max_batch * 1024 * cmax * 2 * 4 + 255) & ~(size_t)255;
    const size_t ad = ((size_t)max_batch * 2 * cmax * 4 + 255) & ~(size_t)255;
    p->ws_bytes = 4 * buf + sc + pb + vt + part + ad;
    if (hipMalloc((void**)&p->ws, p->ws_bytes) != hipSuccess) {
        int r = errorf("mi355_vae_plan_create: hipMalloc of %zu bytes failed", p->ws_bytes);
        delete p;
        return r;
    }
    char* q = p->ws;
    p->X = (bf16_t*)q; q += buf; p->Y = (bf16_t*)q; q += buf; p->T1 = (bf16_t*)q; q += buf; p->T2 = (bf16_t*)q; q += buf;
    p->SC = (float*)q; q += sc; p->P = (bf16_t*)q; q += pb; p->VT = (bf16_t*)q; q += vt;
    p->part = (float*)q; q += part; p->ad = (float*)q; q += ad;
    *out = p;
    return 0;
}

extern "C" int mi355_vae_plan_destroy(mi355_vae_plan* p) {
    if (!p) return 0;
    if (p->ws) (void)hipFree(p->ws);
    delete p;
    return 0;
}
extern "C" int64_t mi355_vae_plan_workspace_bytes(mi355_vae_plan* p) { return p ? (int64_t)p->ws_bytes : 0; }

// ------------------------------------------------------------------------------------ decode
namespace {

struct Ctx {
    mi355_vae_plan* p; hipStream_t st; int B;
};

// 3x3 conv (padding 1) over NHWC `in` ([B][H>>up][W>>up][cw.cipad]) -> out [B*H*W][cw.co]; res != nullptr: + res (may alias out)
int conv3(const Ctx& c, const ConvW& cw, const bf16_t* in, bf16_t* out, int H, int W, int up, const bf16_t* res) {
    const long M = (long)c.B * H * W;
    if (M > 0x7fffffffL) return errorf("mi355_vae_decode: %ld pixels exceed the GEMM row range", M);
    GemmParams g = make_gemm(in, cw.cipad, cw.w, 9L * cw.cipad, M, cw.co, 9 * cw.cipad, res ? EPI_POSADD : EPI_BIAS, cw.b, out, cw.co);
    g.conv_cin = cw.cipad; g.conv_h = H; g.conv_w = W; g.conv_up = up; g.zero_page = c.p->v->zero_page;
    g.aux = res; g.ld_aux = cw.co;
    HIPCHK(launch_gemm(g, c.st));
    return 0;
}

int group_norm(const Ctx& c, const NormW& n, const bf16_t* x, bf16_t* y, long HW, bool silu) {
    const mi355_vae_cfg& cfg = c.p->v->cfg;
    HIPCHK(launch_group_norm(x, y, n.g, n.b, c.p->part, c.p->ad, c.B, HW, n.c, cfg.norm_num_groups, cfg.eps, silu, c.st));
    return 0;
}

// x: [B*H*W][r.ci] in p->X on entry, [B*H*W][r.co] in p->X on exit
int resnet(const Ctx& c, const Resnet& r, int H, int W) {
    mi355_vae_plan* p = c.p;
    const long HW = (long)H * W, M = c.B * HW;
    CHK(group_norm(c, r.n1, p->X, p->T1, HW, true));
    CHK(conv3(c, r.c1, p->T1, p->T2, H, W, 0, nullptr));
    CHK(group_norm(c, r.n2, p->T2, p->T1, HW, true));
    if (r.has_sc) {
        GemmParams g = make_gemm(p->X, r.ci, r.sc.w, r.ci, M, r.co, r.ci, EPI_BIAS, r.sc.b, p->T2, r.co);
        HIPCHK(launch_gemm(g, c.st));
        CHK(conv3(c, r.c2, p->T1, p->X, H, W, 0, p->T2));
    } else {
        CHK(conv3(c, r.c2, p->T1, p->X, H, W, 0, p->X));
    }
    return 0;
}

// single-head self-attention over the S = H*W tokens of each image, head dim C (diffusers Attention with group_norm,
// residual_connection=True): scores are materialised per image in fp32, soft-maxed to bf16, then P.V on the MFMA GEMM
int mid_attention(const Ctx& c, int H, int W) {
    mi355_vae_plan* p = c.p;
    mi355_vae* v = p->v;
    const int C = v->attn_gn.c;
    const long S = (long)H * W, M = c.B * S;
    CHK(group_norm(c, v->attn_gn, p->X, p->T1, S, false));
    GemmParams gqk = make_gemm(p->T1, C, v->to_qk.w, C, M, 2 * C, C, EPI_BIAS, v->to_qk.b, p->T2, 2 * C);
    HIPCHK(launch_gemm(gqk, c.st));
    const float scale = 1.0f / sqrtf((float)C);
    for (int b = 0; b < c.B; ++b) {
        const bf16_t* hn = p->T1 + (size_t)b * S * C;
        const bf16_t* q = p->T2 + (size_t)b * S * 2 * C;
        GemmParams gv = make_gemm(v->to_v.w, C, hn, C, C, (int)S, C, EPI_BIAS_ROW, v->to_v.b, p->VT, S);
        HIPCHK(launch_gemm(gv, c.st));
        GemmParams gs = make_gemm(q, 2 * C, q + C, 2 * C, S, (int)S, C, EPI_F32, nullptr, nullptr, S);
        gs.out_f32 = p->SC; gs.q_scale = 1.0f;
        HIPCHK(launch_gemm(gs, c.st));
        HIPCHK(launch_softmax_rows(p->SC, p->P, S, (int)S, scale, c.st));
        GemmParams go = make_gemm(p->P, S, p->VT, S, S, C, (int)S, EPI_BIAS, v->zero_bias, p->Y + (size_t)b * S * C, C);
        HIPCHK(launch_gemm(go, c.st));
    }
    GemmParams gout = make_gemm(p->Y, C, v->to_out.w, C, M, C, C, EPI_POSADD, v->to_out.b, p->X, C);
    gout.aux = p->X; gout.ld_aux = C;
    HIPCHK(launch_gemm(gout, c.st));
    return 0;
}

}  // namespace

extern "C" int mi355_vae_decode(mi355_vae_plan* p, void* stream, const void* latents, int lat_dtype, int batch, void* images,
                                int img_dtype, int postprocess) {
    if (!p || !latents || !images) return errorf("mi355_vae_decode: null argument");
    if (batch < 1 || batch > p->B) return errorf("mi355_vae_decode: batch %d outside the plan's 1..%d", batch, p->B);
    if (lat_dtype < 0 || lat_dtype > 2) return errorf("mi355_vae_decode: bad latent dtype %d", lat_dtype);
    if (img_dtype != DT_F32 && img_dtype != DT_BF16) return errorf("mi355_vae_decode: image dtype must be fp32 (0) or bf16 (1)");
    mi355_vae* v = p->v;
    CHK(mi355_vae_weights_ready(v));
    const mi355_vae_cfg& cfg = v->cfg;
    Ctx c{p, (hipStream_t)stream, batch};
    int H = p->h, W = p->w;
    const long S = (long)H * W;
    // z = latents / scaling_factor + shift_factor, NHWC, channels padded to 64                 (sd3_5.py:165-166)
    HIPCHK(launch_vae_ingest(latents, lat_dtype, p->T1, batch, cfg.latent_channels, v->conv_in.cipad, S, cfg.scaling_factor,
                             cfg.shift_factor, c.st));
    CHK(conv3(c, v->conv_in, p->T1, p->X, H, W, 0, nullptr));
    CHK(resnet(c, v->mid[0], H, W));
    CHK(mid_attention(c, H, W));
    CHK(resnet(c, v->mid[1], H, W));
    for (int i = 0; i < cfg.num_blocks; ++i) {
        for (const Resnet& r : v->up[i]) CHK(resnet(c, r, H, W));
        if (v->upconv[i].co) {
            H *= 2; W *= 2;
            CHK(conv3(c, v->upconv[i], p->X, p->Y, H, W, 1, nullptr));
            bf16_t* t = p->X; p->X = p->Y; p->Y = t;
        }
    }
    CHK(group_norm(c, v->norm_out, p->X, p->T1, (long)H * W, true));
    {
        const ConvW& cw = v->conv_out;
        const long M = (long)batch * H * W;
        GemmParams g = make_gemm(p->T1, cw.cipad, cw.w, 9L * cw.cipad, M, cw.co, 9 * cw.cipad, EPI_IMG, cw.b, nullptr, 0);
        g.conv_cin = cw.cipad; g.conv_h = H; g.conv_w = W; g.zero_page = v->zero_page;
        g.img_post = postprocess;
        if (img_dtype == DT_F32) g.out_f32 = (float*)images; else g.out = (bf16_t*)images;
        HIPCHK(launch_gemm(g, c.st));
    }
    return 0;
}

// ----------------------------------------------------------------------- operator-level API
static bf16_t* g_zero_page = nullptr;

extern "C" int mi355_op_conv3x3(void* stream, const void* x, const void* w_packed, const float* bias, const void* residual, void* out,
                                int B, int H, int W, int Cin, int Cout, int upsample) {
    if (!x || !w_packed || !bias || !out) return errorf("mi355_op_conv3x3: null argument");
    if (Cin % 64) return errorf("mi355_op_conv3x3: Cin must be a multiple of 64 (pad the channels)");
    if (!g_zero_page) {
        HIPCHK(hipMalloc((void**)&g_zero_page, 256));
        HIPCHK(hipMemset(g_zero_page, 0, 256));
    }
    const long M = (long)B * H * W;
    GemmParams g = make_gemm((const bf16_t*)x, Cin, (const bf16_t*)w_packed, 9L * Cin, M, Cout, 9 * Cin,
                             residual ? EPI_POSADD : EPI_BIAS, bias, (bf16_t*)out, Cout);
    g.conv_cin = Cin; g.conv_h = H; g.conv_w = W; g.conv_up = upsample; g.zero_page = g_zero_page;
    g.aux = (const bf16_t*)residual; g.ld_aux = Cout;
    HIPCHK(launch_gemm(g, (hipStream_t)stream));
    return 0;
}

extern "C" int mi355_op_conv_repack(void* stream, const void* w, int dtype, void* w_packed, int Cout, int Cin, int Cin_pad, int taps) {
    if (!w || !w_packed) return errorf("mi355_op_conv_repack: null argument");
    if (Cin_pad < Cin || Cin_pad % 64) return errorf("mi355_op_conv_repack: Cin_pad must be a multiple of 64 and >= Cin");
    HIPCHK(launch_conv_repack(w, dtype, (bf16_t*)w_packed, Cout, Cin, Cin_pad, taps, (hipStream_t)stream));
    return 0;
}

extern "C" int mi355_op_group_norm(void* stream, const void* x, const float* gamma, const float* beta, void* out, float* scratch,
                                   int B, int64_t HW, int C, int groups, float eps, int silu) {
    if (!x || !gamma || !beta || !out || !scratch) return errorf("mi355_op_group_norm: null argument");
    const int nchunk = gn_num_chunks(HW, C);
    float* part = scratch;
    float* ad = scratch + (size_t)B * nchunk * C * 2;
    hipError_t e = launch_group_norm((const bf16_t*)x, (bf16_t*)out, gamma, beta, part, ad, B, HW, C, groups, eps, silu != 0,
                                     (hipStream_t)stream);
    if (e != hipSuccess) return errorf("mi355_op_group_norm: %s (C must be a multiple of 8 dividing 2048 and of groups)", hipGetErrorString(e));
    return 0;
}
