// mi355_flow -- FLUX.1-specific HBM-bound kernels.
//   rope_norm: per-head RMSNorm of the q / k projections, rotary embedding (adjacent pairs share one angle, diffusers
//   apply_rotary_emb(use_real_unbind_dim=-1)), softmax scale folded into q, scatter to the attention layout.
//   One wave per (token, head): a lane owns exactly one rotary pair of q and one of k (head_dim 128 = 64 lanes x 2).
#include "kernels.h"

namespace mi355 {
namespace {

__global__ __launch_bounds__(256) void rope_norm_kernel(RopeNormParams p) {
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

}  // namespace

hipError_t launch_rope_norm(const RopeNormParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.H <= 0 || (p.src_ld & 1) || (p.q_col & 1) || (p.k_col & 1)) return hipErrorInvalidValue;
    const long items = (long)p.M * p.H;
    hipLaunchKernelGGL(rope_norm_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace mi355
