// mi355_flow -- FLUX.1-specific HBM-bound kernels.
//   rope_norm: per-head RMSNorm of the q / k projections, rotary embedding (adjacent pairs share one angle, diffusers
//   apply_rotary_emb(use_real_unbind_dim=-1)), softmax scale folded into q, scatter to the attention layout.
//   One wave per (token, head): a lane owns exactly one rotary pair of q and one of k (head_dim 128 = 64 lanes x 2).
#include "kernels.h"

namespace mi355 {
namespace {

// (narrow form: 4 bytes per lane and load; kept for sources that are not 16-byte aligned)
__global__ __launch_bounds__(256) void rope_norm_narrow_kernel(RopeNormParams p) {
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);      // (row m, head h)
    if (item >= (long)p.M * p.H) return;
    const int m = (int)(item / p.H), h = (int)(item - (long)m * p.H);
    const int b = m / p.rows_per_sample;
    const int s = m - b * p.rows_per_sample + p.s_off;
    const bf16_t* row = p.src + (long)m * p.src_ld + h * 128 + 2 * lane;
    const unsigned uq = *(const unsigned*)(row + p.q_col);
    const unsigned uk = *(const unsigned*)(row + p.k_col);
    float q0 = bf_lo(uq), q1 = bf_hi(uq), k0 = bf_lo(uk), k1 = bf_hi(uk);
    const float rq = rsqrtf(wave_sum(q0 * q0 + q1 * q1) * (1.0f / 128.0f) + p.eps);
    const float rk = rsqrtf(wave_sum(k0 * k0 + k1 * k1) * (1.0f / 128.0f) + p.eps);
    const float2 wq = *(const float2*)(p.nw_q + 2 * lane);
    const float2 wk = *(const float2*)(p.nw_k + 2 * lane);
    q0 *= rq * wq.x; q1 *= rq * wq.y; k0 *= rk * wk.x; k1 *= rk * wk.y;
    const float2 cs = p.cs[(long)s * 64 + lane];
    // out = x*cos + rot(x)*sin, rot((a, b)) = (-b, a)
    const float qa = (q0 * cs.x - q1 * cs.y) * p.q_scale, qb = (q1 * cs.x + q0 * cs.y) * p.q_scale;
    const float ka = k0 * cs.x - k1 * cs.y, kb = k1 * cs.x + k0 * cs.y;
    const long o = (((long)b * p.H + h) * p.S_pad + s) * 128 + 2 * lane;
    *(unsigned*)(p.q_out + o) = pack_bf16(qa, qb);
    *(unsigned*)(p.k_out + o) = pack_bf16(ka, kb);
}

// Wan: one wave per token row; lane l owns rotary pair l of EVERY head (pairs l, l+64, l+128, ... of the row), so the row RMS is one
// wave reduction and every load / store instruction of the wave covers 256 contiguous bytes.
// sum over the 64 lanes on the VALU (DPP), result wave-uniform: quad swaps, half-row and row mirrors leave every lane its 16-lane row sum,
// row_bcast15 / row_bcast31 carry the row sums up to lane 63.  (wave_sum's six ds_bpermute round trips per sum, twelve sums per row, made the
// q / k kernel 3 x slower when it also measured the row norms.)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    auto dpp = [](float x, auto ctrl, auto rmask) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(rmask)::value, 0xf, false));
    };
    using std::integral_constant;
    v += dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{});      // quad_perm [1,0,3,2]
    v += dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{});      // quad_perm [2,3,0,1]
    v += dpp(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{});     // row_half_mirror
    v += dpp(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{});     // row_mirror
    v += dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});     // row_bcast15 into rows 1, 3
    v += dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});     // row_bcast31 into rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Lane l of a wave owns 8 consecutive features of 4 heads per pass: pass `it` covers heads 4 it .. 4 it + 3, 16 lanes per head, one 16-byte
// load / store per lane and pass (the first version moved 4 bytes per lane and load: 1.5 TB/s).  The rotary pairs (2 i, 2 i + 1) of a lane are
// the same in every pass, so its 4 (cos, sin) pairs are fetched once per row.
// MEASURE: also record the largest squared norm of the rows AS STORED per (batch, head) -- the self-attention's data-dependent score bound.
// A head's 128 features sit in one 16-lane DPP row: 4 DPP steps leave every lane its head's sum.  No atomics (tens of thousands of rows on
// B*H addresses queue: measured 3-10 x the kernel's time): a wave walks NR_RPW consecutive rows of one sample with its running maxima in
// registers and writes ONE partial row [H]; max_finalize_kernel reduces the partial rows.
constexpr int NR_RPW = 8;
__device__ __forceinline__ void unpack8(const uint4 u, float (&v)[8]) {
    v[0] = bf_lo(u.x); v[1] = bf_hi(u.x); v[2] = bf_lo(u.y); v[3] = bf_hi(u.y);
    v[4] = bf_lo(u.z); v[5] = bf_hi(u.z); v[6] = bf_lo(u.w); v[7] = bf_hi(u.w);
}
__device__ __forceinline__ float row16_sum_dpp(float v) {      // sum over the 16 lanes of a DPP row, in every lane
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    using std::integral_constant;
    v += dpp(v, integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v += dpp(v, integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    v += dpp(v, integral_constant<int, 0x141>{});     // row_half_mirror
    v += dpp(v, integral_constant<int, 0x140>{});     // row_mirror
    return v;
}
// FLUX.1 / Qwen-Image q | k producer: per-head RMSNorm + RoPE + head-major scatter.  One wave per token row, 4 heads per pass, a head per
// 16-lane DPP row, 16 bytes per lane and access (the narrow form above -- one wave per (token, head), 4-byte accesses, two ds_bpermute
// reductions -- ran at ~1.5 TB/s: 2.6-2.8 % of the FLUX.1 / Qwen-Image rollouts).  Same arithmetic per element; the per-head sums of
// squares are formed in another order.
__global__ __launch_bounds__(256) void rope_norm_kernel(RopeNormParams p) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= p.M) return;
    const int hl = lane >> 4, d0 = (lane & 15) * 8;
    const int b = m / p.rows_per_sample;
    const int s = m - b * p.rows_per_sample + p.s_off;
    const float4 wq0 = *(const float4*)(p.nw_q + d0), wq1 = *(const float4*)(p.nw_q + d0 + 4);
    const float4 wk0 = *(const float4*)(p.nw_k + d0), wk1 = *(const float4*)(p.nw_k + d0 + 4);
    const float wq[8] = {wq0.x, wq0.y, wq0.z, wq0.w, wq1.x, wq1.y, wq1.z, wq1.w};
    const float wk[8] = {wk0.x, wk0.y, wk0.z, wk0.w, wk1.x, wk1.y, wk1.z, wk1.w};
    const float4 c01 = *(const float4*)(p.cs + (long)s * 64 + (lane & 15) * 4), c23 = *(const float4*)(p.cs + (long)s * 64 + (lane & 15) * 4 + 2);
    const float cs[4][2] = {{c01.x, c01.y}, {c01.z, c01.w}, {c23.x, c23.y}, {c23.z, c23.w}};
    const bf16_t* row = p.src + (long)m * p.src_ld + lane * 8;
    for (int h0 = 0; h0 < p.H; h0 += 4) {
        const int h = h0 + hl;
        if (h >= p.H) continue;                 // (uniform inside a 16-lane DPP row)
        float q[8], k[8];
        unpack8(*(const uint4*)(row + p.q_col + h0 * 128), q);
        unpack8(*(const uint4*)(row + p.k_col + h0 * 128), k);
        float sq = 0.f, sk = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sq += q[e] * q[e]; sk += k[e] * k[e]; }
        const float rq = rsqrtf(row16_sum_dpp(sq) * (1.0f / 128.0f) + p.eps);
        const float rk = rsqrtf(row16_sum_dpp(sk) * (1.0f / 128.0f) + p.eps);
        if (p.rstd_out && (lane & 15) == 0) {       // training-mode forward only (same binary as the rollout: a store, no arithmetic)
            p.rstd_out[(long)m * 2 * p.H + h] = rq;
            p.rstd_out[(long)m * 2 * p.H + p.H + h] = rk;
        }
        unsigned uq[4], uk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float q0 = q[2 * j] * (rq * wq[2 * j]), q1 = q[2 * j + 1] * (rq * wq[2 * j + 1]);
            const float k0 = k[2 * j] * (rk * wk[2 * j]), k1 = k[2 * j + 1] * (rk * wk[2 * j + 1]);
            // out = x*cos + rot(x)*sin, rot((a, b)) = (-b, a)
            uq[j] = pack_bf16((q0 * cs[j][0] - q1 * cs[j][1]) * p.q_scale, (q1 * cs[j][0] + q0 * cs[j][1]) * p.q_scale);
            uk[j] = pack_bf16(k0 * cs[j][0] - k1 * cs[j][1], k1 * cs[j][0] + k0 * cs[j][1]);
        }
        const long o = (((long)b * p.H + h) * p.S_pad + s) * 128 + d0;
        *(uint4*)(p.q_out + o) = make_uint4(uq[0], uq[1], uq[2], uq[3]);
        *(uint4*)(p.k_out + o) = make_uint4(uk[0], uk[1], uk[2], uk[3]);
    }
}

template <int MAXH, bool MEASURE>
__global__ __launch_bounds__(256) void norm_rope_full_kernel(NormRopeFullParams p) {
    constexpr int NI = MAXH / 4;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int hl = lane >> 4, d0 = (lane & 15) * 8;          // head within the pass, first feature inside the head
    int m_first, n_rows, b0 = 0;
    if constexpr (MEASURE) {        // grid (chunks of 4 * RPW rows, batch)
        b0 = blockIdx.y;
        const int s_first = (blockIdx.x * 4 + wave) * NR_RPW;
        n_rows = min(NR_RPW, p.rows_per_sample - s_first);
        m_first = b0 * p.rows_per_sample + s_first;
    } else {
        m_first = blockIdx.x * 4 + wave;
        n_rows = m_first < p.M ? 1 : 0;
    }
    float mx[MEASURE ? NI : 1];
    if constexpr (MEASURE) {
#pragma unroll
        for (int it = 0; it < NI; ++it) mx[it] = 0.f;
    }
    for (int r = 0; r < n_rows; ++r) {
        const int m = m_first + r;
        const int b = MEASURE ? b0 : m / p.rows_per_sample;
        const int s = m - b * p.rows_per_sample + p.s_off;
        const bf16_t* row = p.src + (long)m * p.src_ld + p.col + lane * 8;
        float x[NI][8];
        float ss = 0.f;
#pragma unroll
        for (int it = 0; it < NI; ++it)
            if (it * 4 + hl < p.H) {
                unpack8(*(const uint4*)(row + it * 512), x[it]);
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += x[it][e] * x[it][e];
            }
        const float rstd = rsqrtf(wave_sum_dpp(ss) / (float)(p.H * 128) + p.eps);
        if (p.rstd_out && lane == 0) p.rstd_out[m] = rstd;       // (training-mode forward; a runtime branch: same binary as the rollout's)
        float cs[4][2] = {{1.f, 0.f}, {1.f, 0.f}, {1.f, 0.f}, {1.f, 0.f}};
        if (p.cs) {
            const float4 c01 = *(const float4*)(p.cs + (long)s * 64 + (lane & 15) * 4), c23 = *(const float4*)(p.cs + (long)s * 64 + (lane & 15) * 4 + 2);
            cs[0][0] = c01.x; cs[0][1] = c01.y; cs[1][0] = c01.z; cs[1][1] = c01.w;
            cs[2][0] = c23.x; cs[2][1] = c23.y; cs[3][0] = c23.z; cs[3][1] = c23.w;
        }
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int h = it * 4 + hl;
            if (h < p.H) {
                const float4 w0 = *(const float4*)(p.weight + h * 128 + d0), w1 = *(const float4*)(p.weight + h * 128 + d0 + 4);
                const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                unsigned u[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = x[it][2 * j] * rstd * w[2 * j], bb = x[it][2 * j + 1] * rstd * w[2 * j + 1];
                    u[j] = pack_bf16((a * cs[j][0] - bb * cs[j][1]) * p.out_scale, (bb * cs[j][0] + a * cs[j][1]) * p.out_scale);
                }
                *(uint4*)(p.out + (((long)b * p.H + h) * p.S_pad + s) * 128 + d0) = make_uint4(u[0], u[1], u[2], u[3]);
                if constexpr (MEASURE) {
                    float n2 = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) n2 += bf_lo(u[j]) * bf_lo(u[j]) + bf_hi(u[j]) * bf_hi(u[j]);
                    mx[it] = fmaxf(mx[it], row16_sum_dpp(n2));
                }
            }          // (h < p.H is uniform inside a 16-lane DPP row: hl = lane >> 4)
        }
    }
    if constexpr (MEASURE) {        // partial row of this wave: the first lane of every 16-lane row holds its head's maximum
        float* dst = p.max2_part + (((long)b0 * gridDim.x + blockIdx.x) * 4 + wave) * p.H;
#pragma unroll
        for (int it = 0; it < NI; ++it)
            if ((lane & 15) == 0 && it * 4 + hl < p.H) dst[it * 4 + hl] = mx[it];
    }
}

// backward of norm_rope_full_kernel for one tensor (see NormRopeFullBwdParams): one wave per token, the forward's lane mapping (a head per
// 16-lane DPP row, 8 features per lane, H / 4 passes), 16-byte accesses.  Forward per element pair (a, b) of a head, x = projection + bias:
//   (a', b') = (a, b) * rstd * (w_a, w_b);  stored (a' c - b' s, b' c + a' s) * out_scale.
template <int MAXH>
__global__ __launch_bounds__(256) void norm_rope_full_bwd_kernel(NormRopeFullBwdParams p) {
    constexpr int NI = MAXH / 4;
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= p.M) return;
    const int hl = lane >> 4, d0 = (lane & 15) * 8;
    const int b = m / p.rows_per_sample;
    const int s = m - b * p.rows_per_sample + p.s_off;
    bf16_t* orow = p.out + (long)m * p.out_ld + p.col + lane * 8;
    if (!p.weight) {            // gather only (the V gradient)
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int h = it * 4 + hl;
            if (h < p.H) *(uint4*)(orow + it * 512) = *(const uint4*)(p.dy + (((long)b * p.H + h) * p.S_pad + s) * 128 + d0);
        }
        return;
    }
    float cs[4][2] = {{1.f, 0.f}, {1.f, 0.f}, {1.f, 0.f}, {1.f, 0.f}};
    if (p.cs) {
        const float4 c01 = *(const float4*)(p.cs + (long)s * 64 + (lane & 15) * 4), c23 = *(const float4*)(p.cs + (long)s * 64 + (lane & 15) * 4 + 2);
        cs[0][0] = c01.x; cs[0][1] = c01.y; cs[1][0] = c01.z; cs[1][1] = c01.w;
        cs[2][0] = c23.x; cs[2][1] = c23.y; cs[3][0] = c23.z; cs[3][1] = c23.w;
    }
    const float inv_scale = 1.0f / p.out_scale;
    float xh[NI][8], g[NI][8];
    float dot = 0.f;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int h = it * 4 + hl;
        if (h < p.H) {
            const long src = (((long)b * p.H + h) * p.S_pad + s) * 128 + d0;
            float yv[8], dv[8];
            unpack8(*(const uint4*)(p.y + src), yv);
            unpack8(*(const uint4*)(p.dy + src), dv);
            const float4 w0 = *(const float4*)(p.weight + h * 128 + d0), w1 = *(const float4*)(p.weight + h * 128 + d0 + 4);
            const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float c = cs[j][0], sn = cs[j][1];
                const float z0 = yv[2 * j] * inv_scale, z1 = yv[2 * j + 1] * inv_scale;          // un-scale, un-rotate: y = R^T z
                const float y0 = z0 * c + z1 * sn, y1 = z1 * c - z0 * sn;
                const float e0 = dv[2 * j] * p.out_scale, e1 = dv[2 * j + 1] * p.out_scale;      // d z = d y~ * out_scale; d y = R^T d z
                const float dy0 = e0 * c + e1 * sn, dy1 = e1 * c - e0 * sn;
                xh[it][2 * j] = y0 / w[2 * j]; xh[it][2 * j + 1] = y1 / w[2 * j + 1];
                g[it][2 * j] = dy0 * w[2 * j]; g[it][2 * j + 1] = dy1 * w[2 * j + 1];
                dot += g[it][2 * j] * xh[it][2 * j] + g[it][2 * j + 1] * xh[it][2 * j + 1];
            }
        }
    }
    const float mean = wave_sum_dpp(dot) / (float)(p.H * 128);
    const float rstd = p.rstd[m];
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int h = it * 4 + hl;
        if (h < p.H) {
            unsigned u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                u[j] = pack_bf16(rstd * (g[it][2 * j] - xh[it][2 * j] * mean), rstd * (g[it][2 * j + 1] - xh[it][2 * j + 1] * mean));
            *(uint4*)(orow + it * 512) = make_uint4(u[0], u[1], u[2], u[3]);
        }
    }
}

// max2[b][h] = float bits of the maximum over the `nparts` partial rows of sample b (one wave per (b, h))
__global__ __launch_bounds__(256) void max_finalize_kernel(const float* part, int nparts, int H, int BH, unsigned* out) {
    const int lane = threadIdx.x & 63;
    const int bh = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bh >= BH) return;
    const int b = bh / H, h = bh - b * H;
    float v = 0.f;
    for (int i = lane; i < nparts; i += 64) v = fmaxf(v, part[((long)b * nparts + i) * H + h]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if (lane == 0) out[bh] = __float_as_uint(v);
}

__global__ __launch_bounds__(256) void bcast_add_kernel(const bf16_t* a, const float* table, bf16_t* out, long out_ld, int rows, int W, int J) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // over rows * J * W / 2 pairs
    const long per_row = (long)J * (W / 2);
    if (i >= rows * per_row) return;
    const int m = (int)(i / per_row);
    const long r = i - m * per_row;
    const int j = (int)(r / (W / 2)), x = (int)(r - (long)j * (W / 2)) * 2;
    const unsigned u = *(const unsigned*)(a + (long)m * W + x);
    const float2 t = *(const float2*)(table + (long)j * W + x);
    *(unsigned*)(out + (long)m * out_ld + (long)j * W + x) = pack_bf16(bf_lo(u) + t.x, bf_hi(u) + t.y);
}

__global__ __launch_bounds__(256) void affine_to_mod_kernel(const float* weight, const float* bias, bf16_t* out, int D) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    out[d] = f2bf(bias[d]);
    out[D + d] = f2bf(weight[d] - 1.0f);
}

// Qwen-Image txt_norm: RMSNorm over a whole row of D features with a weight (diffusers RMSNorm: fp32 variance, x * rsqrt(var + eps) * w).
// One wave per row; a lane walks the row in 4-byte pairs (every load instruction of the wave covers 256 contiguous bytes).
__global__ __launch_bounds__(256) void rms_rows_kernel(const bf16_t* x, long ldx, const float* w, bf16_t* out, long ldo, int M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const bf16_t* row = x + (long)m * ldx;
    float ss = 0.f;
    for (int d = 2 * lane; d < D; d += 128) {
        const unsigned u = *(const unsigned*)(row + d);
        ss += bf_lo(u) * bf_lo(u) + bf_hi(u) * bf_hi(u);
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
    for (int d = 2 * lane; d < D; d += 128) {
        const unsigned u = *(const unsigned*)(row + d);
        const float2 ww = *(const float2*)(w + d);
        *(unsigned*)(out + (long)m * ldo + d) = pack_bf16(bf_lo(u) * rstd * ww.x, bf_hi(u) * rstd * ww.y);
    }
}

// Qwen-Image true-CFG with norm rescale (reference models/qwen_image/qwen_image.py:579-587) on bf16 predictions, every torch op rounding
// to bf16 as the reference's bf16 tensors do:  comb = neg + g * (pos - neg);  out = comb * (||pos|| / ||comb||), norms over the C = 64
// channels of a token.  One wave per token, one channel per lane.
__global__ __launch_bounds__(256) void cfg_rescale_kernel(const bf16_t* neg, const bf16_t* pos, float g, bf16_t* out, long rows) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= rows) return;
    const float n = bf2f(neg[m * 64 + lane]), c = bf2f(pos[m * 64 + lane]);
    const float comb = round_bf16(n + round_bf16(g * round_bf16(c - n)));
    const float cn = round_bf16(sqrtf(wave_sum(c * c)));
    const float nn = round_bf16(sqrtf(wave_sum(comb * comb)));
    out[m * 64 + lane] = f2bf(comb * round_bf16(cn / nn));
}


// adjoint of cfg_rescale_kernel (the norms are differentiated through, as autograd does in the reference: qwen_image.py:579-587):
//   s = ||pos|| / ||comb||;  dcomb = s * dout - (dout . comb) * s * comb / ||comb||^2;  dneg = (1 - g) * dcomb;
//   dpos = g * dcomb + (dout . comb) * pos / (||pos|| * ||comb||).       dout fp32 [rows][64] -> dneg, dpos bf16.
__global__ __launch_bounds__(256) void cfg_rescale_bwd_kernel(const bf16_t* neg, const bf16_t* pos, float g, const float* dout, bf16_t* dneg, bf16_t* dpos,
                                                              long rows) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= rows) return;
    const float n = bf2f(neg[m * 64 + lane]), c = bf2f(pos[m * 64 + lane]), d = dout[m * 64 + lane];
    const float comb = round_bf16(n + round_bf16(g * round_bf16(c - n)));
    const float cn = sqrtf(wave_sum(c * c)), nn = sqrtf(wave_sum(comb * comb));
    const float dot = wave_sum(d * comb);
    const float s = cn / nn;
    const float dcomb = s * d - dot * s * comb / (nn * nn);
    dneg[m * 64 + lane] = f2bf((1.0f - g) * dcomb);
    dpos[m * 64 + lane] = f2bf(g * dcomb + dot * c / (cn * nn));
}

}  // namespace

hipError_t launch_rms_rows(const bf16_t* x, long ldx, const float* w, bf16_t* out, long ldo, int M, int D, float eps, hipStream_t stream) {
    if (M <= 0 || D <= 0 || (D & 1) || (ldx & 1) || (ldo & 1)) return hipErrorInvalidValue;      // (validation first: the trace's region arithmetic assumes M >= 1)
    if (sched_trace_on())
        sched_trace_launch("rms_rows", stream, {treg(x, ((size_t)(M - 1) * ldx + D) * 2), treg(w, (size_t)D * 4)}, {treg(out, ((size_t)(M - 1) * ldo + D) * 2)});
    hipLaunchKernelGGL(rms_rows_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, stream, x, ldx, w, out, ldo, M, D, eps);
    return hipGetLastError();
}

hipError_t launch_cfg_rescale(const bf16_t* neg, const bf16_t* pos, float g, bf16_t* out, long rows, int C, hipStream_t stream) {
    if (rows <= 0 || C != 64) return hipErrorInvalidValue;
    if (sched_trace_on())
        sched_trace_launch("cfg_rescale", stream, {treg(neg, (size_t)rows * C * 2), treg(pos, (size_t)rows * C * 2)}, {treg(out, (size_t)rows * C * 2)});
    hipLaunchKernelGGL(cfg_rescale_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, neg, pos, g, out, rows);
    return hipGetLastError();
}

hipError_t launch_cfg_rescale_bwd(const bf16_t* neg, const bf16_t* pos, float g, const float* dout, bf16_t* dneg, bf16_t* dpos, long rows, int C,
                                  hipStream_t stream) {
    if (rows <= 0 || C != 64) return hipErrorInvalidValue;
    if (sched_trace_on())
        sched_trace_launch("cfg_rescale_bwd", stream, {treg(neg, (size_t)rows * C * 2), treg(pos, (size_t)rows * C * 2), treg(dout, (size_t)rows * C * 4)},
                           {treg(dneg, (size_t)rows * C * 2), treg(dpos, (size_t)rows * C * 2)});
    hipLaunchKernelGGL(cfg_rescale_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, neg, pos, g, dout, dneg, dpos, rows);
    return hipGetLastError();
}

int norm_rope_parts(int rows_per_sample) { return (rows_per_sample + 4 * NR_RPW - 1) / (4 * NR_RPW) * 4; }   // partial rows per sample
hipError_t launch_norm_rope_full(const NormRopeFullParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.H <= 0 || p.H > 48 || p.rows_per_sample <= 0 || (p.src_ld & 7) || (p.col & 7) || ((size_t)p.src & 15) || ((size_t)p.out & 15))
        return hipErrorInvalidValue;
    if (sched_trace_on()) {       // (regions: the source rows, the weight; the head-major output rows of every (sample, head), 1 / rms, the measured maxima)
        const size_t blocks = (size_t)((p.M + p.rows_per_sample - 1) / p.rows_per_sample) * p.H, len = (size_t)p.rows_per_sample * 256, stride = (size_t)p.S_pad * 256;
        const size_t nb = (size_t)((p.M + p.rows_per_sample - 1) / p.rows_per_sample);
        sched_trace_launch("norm_rope_full", stream, {treg(p.src + p.col, ((size_t)(p.M - 1) * p.src_ld + (size_t)p.H * 128) * 2), treg(p.weight, (size_t)p.H * 512)},
                           {tregs(p.out + (size_t)p.s_off * 128, len, stride, blocks), treg(p.rstd_out, p.rstd_out ? (size_t)p.M * 4 : 0),
                            treg(p.max2, p.max2 ? nb * p.H * 4 : 0), treg(p.max2_part, p.max2 ? nb * (size_t)norm_rope_parts(p.rows_per_sample) * p.H * 4 : 0)});
    }
    if (p.max2) {
        if (!p.max2_part || p.s_off != 0 || p.M % p.rows_per_sample) return hipErrorInvalidValue;
        const int B = p.M / p.rows_per_sample;
        const dim3 grid((unsigned)(norm_rope_parts(p.rows_per_sample) / 4), (unsigned)B);
        if (p.H <= 12) hipLaunchKernelGGL((norm_rope_full_kernel<12, true>), grid, dim3(256), 0, stream, p);
        else if (p.H <= 24) hipLaunchKernelGGL((norm_rope_full_kernel<24, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((norm_rope_full_kernel<48, true>), grid, dim3(256), 0, stream, p);
        hipLaunchKernelGGL(max_finalize_kernel, dim3((unsigned)((B * p.H + 3) / 4)), dim3(256), 0, stream, p.max2_part, norm_rope_parts(p.rows_per_sample),
                           p.H, B * p.H, p.max2);
        return hipGetLastError();
    }
    const dim3 grid((unsigned)((p.M + 3) / 4));
    if (p.H <= 12) hipLaunchKernelGGL((norm_rope_full_kernel<12, false>), grid, dim3(256), 0, stream, p);
    else if (p.H <= 24) hipLaunchKernelGGL((norm_rope_full_kernel<24, false>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((norm_rope_full_kernel<48, false>), grid, dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_norm_rope_full_bwd(const NormRopeFullBwdParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.H <= 0 || p.H > 48 || p.rows_per_sample <= 0 || (p.out_ld & 7) || (p.col & 7) || ((size_t)p.y & 15) || ((size_t)p.dy & 15) ||
        ((size_t)p.out & 15) || (p.weight && (!p.rstd || p.out_scale == 0.f)))
        return hipErrorInvalidValue;
    if (sched_trace_on()) {
        const size_t blocks = (size_t)((p.M + p.rows_per_sample - 1) / p.rows_per_sample) * p.H, len = (size_t)p.rows_per_sample * 256, stride = (size_t)p.S_pad * 256;
        sched_trace_launch("norm_rope_full_bwd", stream, {tregs(p.y + (size_t)p.s_off * 128, p.weight ? len : 0, stride, blocks), tregs(p.dy + (size_t)p.s_off * 128, len, stride, blocks),
                                                          treg(p.rstd, p.weight ? (size_t)p.M * 4 : 0)},
                           {tregs(p.out + p.col, (size_t)p.H * 256, (size_t)p.out_ld * 2, (size_t)p.M)});
    }
    const dim3 grid((unsigned)((p.M + 3) / 4));
    if (p.H <= 12) hipLaunchKernelGGL((norm_rope_full_bwd_kernel<12>), grid, dim3(256), 0, stream, p);
    else if (p.H <= 24) hipLaunchKernelGGL((norm_rope_full_bwd_kernel<24>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((norm_rope_full_bwd_kernel<48>), grid, dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_bcast_add(const bf16_t* a, const float* table, bf16_t* out, long out_ld, int rows, int W, int J, hipStream_t stream) {
    if (rows <= 0 || W <= 0 || J <= 0 || (W & 1)) return hipErrorInvalidValue;
    if (sched_trace_on())
        sched_trace_launch("bcast_add", stream, {treg(a, (size_t)rows * W * 2), treg(table, (size_t)J * W * 4)}, {treg(out, ((size_t)(rows - 1) * out_ld + (size_t)J * W) * 2)});
    const long n = (long)rows * J * (W / 2);
    hipLaunchKernelGGL(bcast_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, table, out, out_ld, rows, W, J);
    return hipGetLastError();
}

hipError_t launch_affine_to_mod(const float* weight, const float* bias, bf16_t* out, int D, hipStream_t stream) {
    hipLaunchKernelGGL(affine_to_mod_kernel, dim3((D + 255) / 256), dim3(256), 0, stream, weight, bias, out, D);
    return hipGetLastError();
}

hipError_t launch_rope_norm(const RopeNormParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.H <= 0 || p.rows_per_sample <= 0 || (p.src_ld & 1) || (p.q_col & 1) || (p.k_col & 1)) return hipErrorInvalidValue;
    if (sched_trace_on()) {
        const int hi = p.q_col > p.k_col ? p.q_col : p.k_col;
        const size_t blocks = (size_t)((p.M + p.rows_per_sample - 1) / p.rows_per_sample) * p.H, len = (size_t)p.rows_per_sample * 256, stride = (size_t)p.S_pad * 256;
        sched_trace_launch("rope_norm", stream, {treg(p.src, ((size_t)(p.M - 1) * p.src_ld + hi + (size_t)p.H * 128) * 2), treg(p.nw_q, 512), treg(p.nw_k, 512)},
                           {tregs(p.q_out + (size_t)p.s_off * 128, len, stride, blocks), tregs(p.k_out + (size_t)p.s_off * 128, len, stride, blocks),
                            treg(p.rstd_out, p.rstd_out ? (size_t)p.M * 2 * p.H * 4 : 0)});
    }
    if (!((p.src_ld | p.q_col | p.k_col) & 7) && !(((size_t)p.src | (size_t)p.q_out | (size_t)p.k_out) & 15)) {
        hipLaunchKernelGGL(rope_norm_kernel, dim3((unsigned)((p.M + 3) / 4)), dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    if (p.rstd_out) return hipErrorInvalidValue;      // the training-mode forward needs the 16-byte form (aligned operands)
    const long items = (long)p.M * p.H;
    hipLaunchKernelGGL(rope_norm_narrow_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace mi355
