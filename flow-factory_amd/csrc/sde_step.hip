// mi355_flow -- fused SDE/ODE scheduler step + Gaussian log-prob (op K14-K17 of SURVEY.md 2.3).
//
// One kernel replaces the reference's ~25 elementwise torch kernels, its full-tensor mean and
// its three host syncs (reference src/flow_factory/scheduler/flow_match_euler_discrete.py:305-426,
// CFG combine sd3_5.py:431-433, cast_latents models/abc.py:172-182):
//   v      = CFG-combine(uncond, text) evaluated op-by-op in bf16 (as torch does on bf16 tensors)
//   mean   = dynamics-specific drift (ODE / Flow-SDE / Dance-SDE / CPS), all in fp32
//   x'     = mean + std * eps, value-rounded to the latent storage dtype (fp16 clamp included)
//   logp_b = mean over (C,H,W) of the Gaussian log-density of x' (or of the provided x' on replay)
// The fp32 arithmetic follows the reference's operation ORDER and this file is compiled with
// -ffp-contract=off, so mean / x' are bit-identical to torch CPU fp32; only the log-prob
// reduction order (and logf/sinf last-ulp) differs.
//
// HBM-bound and tiny (about 4 MB per 1024^2 sample) next to the 11 TFLOP transformer forward, so
// the launch shape is one 1024-thread workgroup per sample: deterministic reduction, no atomics.
#include "kernels.h"

namespace mi355 {
namespace {

constexpr int NT = 1024;

__device__ __forceinline__ float cfg_bf16(float vu, float vt, float g) {
    const float d = round_bf16(vt - vu);
    const float s = round_bf16(g * d);
    return round_bf16(vu + s);
}

__global__ __launch_bounds__(NT) void sde_step_kernel(SdeStepParams p) {
    __shared__ float red[NT / 64];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const float sigma = p.sigma[b * p.scalar_stride];
    const float sigma_next = p.sigma_next[b * p.scalar_stride];
    float eta = p.eta[b * p.scalar_stride];
    const int dyn = p.dynamics;
    if (dyn == DYN_ODE) eta = 0.0f;
    const float dt = sigma_next - sigma;

    // per-sample scalars, same fp32 op order as the reference's (B,1,1,1) tensors
    float std_dev = 0.f, c1 = 0.f, c2 = 0.f, sv = 0.f, den = 1.f, log_sv = 0.f;
    float dance_k = 0.f, one_m_sigma = 1.0f - sigma, cps_a = 0.f, cps_b = 0.f;
    const float LOG_SQRT_2PI = logf(sqrtf(2.0f * 3.14159274101257324f));
    if (dyn == DYN_FLOW_SDE) {
        const float sden = (sigma == 1.0f) ? p.sigma_max : sigma;
        std_dev = sqrtf(sigma / (1.0f - sden)) * eta;
        const float s2 = std_dev * std_dev;
        c1 = 1.0f + s2 / (2.0f * sigma) * dt;
        c2 = 1.0f + s2 * (1.0f - sigma) / (2.0f * sigma);
        sv = std_dev * sqrtf(-1.0f * dt);
    } else if (dyn == DYN_DANCE_SDE) {
        std_dev = eta;
        dance_k = 0.5f * (eta * eta);
        sv = std_dev * sqrtf(-1.0f * dt);
    } else if (dyn == DYN_CPS) {
        // sin through fp64 so the fp32 result is correctly rounded (torch CPU's sinf is <= 1 ulp off that)
        std_dev = sigma_next * (float)sin((double)(eta * 3.14159274101257324f / 2.0f));
        cps_a = 1.0f - sigma_next;
        cps_b = sqrtf(sigma_next * sigma_next - std_dev * std_dev);
        sv = std_dev;
    }
    if (dyn == DYN_FLOW_SDE || dyn == DYN_DANCE_SDE) {
        den = 2.0f * (sv * sv);
        log_sv = logf(sv);
    }

    // compute_log_prob == 2: only on steps with noise (decided on the device: keeps a captured rollout's launch
    // sequence independent of which steps the epoch's seed made SDE steps)
    const bool do_lp = p.compute_log_prob == 1 || (p.compute_log_prob == 2 && eta > 0.f);
    const long base = (long)b * p.n;
    float lp_sum = 0.f;
    for (long i = tid; i < p.n; i += NT) {
        const long gi = base + i;
        float v = bf2f(p.v_text[gi]);
        if (p.v_uncond) v = cfg_bf16(bf2f(p.v_uncond[gi]), v, p.guidance);
        const float x = load_as_f32(p.latents, gi, p.lat_dt);
        float mean;
        if (dyn == DYN_ODE) {
            mean = x + v * dt;
        } else if (dyn == DYN_FLOW_SDE) {
            mean = x * c1 + v * c2 * dt;
        } else if (dyn == DYN_DANCE_SDE) {
            const float x0 = x - sigma * v;
            const float log_term = dance_k * (x - x0 * one_m_sigma) / (sigma * sigma);
            mean = x + (v + log_term) * dt;
        } else {
            const float x0 = x - sigma * v;
            const float x1 = x + v * one_m_sigma;
            mean = x0 * cps_a + x1 * cps_b;
        }
        float nxt;
        if (p.next_in) {
            nxt = load_as_f32(p.next_in, gi, p.next_in_dt);
        } else if (dyn == DYN_ODE) {
            nxt = mean;   // reference returns the unrounded mean; cast_latents rounds on store
        } else {
            nxt = mean + sv * p.noise[gi];
            nxt = round_to_dtype(nxt, p.lat_dt);
        }
        if (p.next_out) store_from_f32(p.next_out, gi, p.next_out_dt, nxt);
        if (p.next_f32) p.next_f32[gi] = nxt;
        if (p.mean_out) p.mean_out[gi] = mean;
        if (p.noise_pred_out) p.noise_pred_out[gi] = v;
        if (do_lp) {
            const float d = nxt - mean;
            float lp;
            if (dyn == DYN_CPS) lp = -(d * d);
            else if (dyn == DYN_ODE) lp = 0.f;
            else lp = -(d * d) / den - log_sv - LOG_SQRT_2PI;
            lp_sum += lp;
        }
    }
    if (do_lp && p.log_prob) {
        lp_sum = wave_sum(lp_sum);
        if ((tid & 63) == 0) red[tid >> 6] = lp_sum;
        __syncthreads();
        if (tid < 64) {
            float s = tid < NT / 64 ? red[tid] : 0.f;
            s = wave_sum(s);
            if (tid == 0) p.log_prob[b] = s / (float)p.n;
        }
    }
    if (tid == 0) {
        if (p.std_dev_t) p.std_dev_t[b] = std_dev;
        if (p.dt_out) p.dt_out[b] = dt;
    }
}

}  // namespace

hipError_t launch_sde_step(const SdeStepParams& p, hipStream_t stream) {
    if (p.B <= 0 || p.n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sde_step_kernel, dim3(p.B), dim3(NT), 0, stream, p);
    return hipGetLastError();
}

}  // namespace mi355
