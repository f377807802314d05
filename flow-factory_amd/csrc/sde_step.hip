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
// HBM-bound (about 4 MB per 1024^2 sample).  Launch shape: NCHUNK workgroups per sample, 4 consecutive elements per
// thread and iteration (8/16-byte accesses); the log-prob mean is reduced DETERMINISTICALLY: every workgroup writes its
// partial sum, the last one to arrive (agent-scope ticket) adds the NCHUNK partials in index order, so the replay of a
// rollout step reproduces the rollout log-prob bit for bit (ratio == 1 invariant).
#include <string.h>
#include "kernels.h"

namespace mi355 {
namespace {

constexpr int NT = 256;
constexpr int NCHUNK = 64;          // workgroups per sample
constexpr int SCRATCH_SLOTS = 16;   // launches in flight that may share the scratch ring
constexpr int MAX_B = 4096;

// CFG combine `u + g * (c - u)` evaluated op by op in the tensors' dtype, as torch does (sd3_5.py:431-433)
__device__ __forceinline__ float cfg_combine(float vu, float vt, float g, int dt) {
    if (dt == DT_F32) return vu + g * (vt - vu);
    const float d = round_to_dtype(vt - vu, dt);
    const float s = round_to_dtype(g * d, dt);
    return round_to_dtype(vu + s, dt);
}

__device__ __forceinline__ void load4(const void* p, long i, int dt, float (&o)[4]) {
    if (dt == DT_F32) { const float4 v = *(const float4*)((const float*)p + i); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    else if (dt == DT_BF16) { const uint2 v = *(const uint2*)((const bf16_t*)p + i); o[0] = bf_lo(v.x); o[1] = bf_hi(v.x); o[2] = bf_lo(v.y); o[3] = bf_hi(v.y); }
    else { typedef __attribute__((ext_vector_type(4))) _Float16 h4; const h4 v = *(const h4*)((const _Float16*)p + i);
           o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3]; }
}
__device__ __forceinline__ void store4(void* p, long i, int dt, const float (&v)[4]) {
    if (dt == DT_F32) *(float4*)((float*)p + i) = make_float4(v[0], v[1], v[2], v[3]);
    else if (dt == DT_BF16) *(uint2*)((bf16_t*)p + i) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
    else { typedef __attribute__((ext_vector_type(4))) _Float16 h4; h4 o;
#pragma unroll
           for (int e = 0; e < 4; ++e) o[e] = (_Float16)fminf(fmaxf(v[e], -65504.0f), 65504.0f);
           *(h4*)((_Float16*)p + i) = o; }
}

// partial[b][chunk] + ticket[b] live in `scratch` (zeroed once; the last workgroup of a sample resets its ticket)
__global__ __launch_bounds__(NT) void sde_step_kernel(SdeStepParams p, float* partial, unsigned* ticket) {
    __shared__ float red[NT / 64];
    __shared__ int is_last;
    const int b = blockIdx.y;
    const int chunk = blockIdx.x;
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
    const long per = ((p.n + NCHUNK - 1) / NCHUNK + 3) & ~3L;   // elements per chunk, multiple of 4
    const long lo = (long)chunk * per;
    const long hi = lo + per < p.n ? lo + per : p.n;
    float lp_sum = 0.f;
    for (long i = lo + 4 * tid; i < hi; i += 4 * NT) {
        const long gi = base + i;
        const int cnt = hi - i >= 4 ? 4 : (int)(hi - i);
        float v[4], x[4], nz[4] = {0.f, 0.f, 0.f, 0.f}, nin[4], mean[4], nxt[4];
        if (cnt == 4 && (p.n & 3) == 0) {
            load4(p.v_text, gi, p.v_dt, v);
            if (p.v_uncond) {
                float u[4];
                load4(p.v_uncond, gi, p.v_dt, u);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = cfg_combine(u[e], v[e], p.guidance, p.v_dt);
            }
            load4(p.latents, gi, p.lat_dt, x);
            if (p.next_in) load4(p.next_in, gi, p.next_in_dt, nin);
            else if (dyn != DYN_ODE) load4(p.noise, gi, DT_F32, nz);
        } else {
            for (int e = 0; e < 4; ++e) {
                const long g = gi + (e < cnt ? e : 0);
                v[e] = load_as_f32(p.v_text, g, p.v_dt);
                if (p.v_uncond) v[e] = cfg_combine(load_as_f32(p.v_uncond, g, p.v_dt), v[e], p.guidance, p.v_dt);
                x[e] = load_as_f32(p.latents, g, p.lat_dt);
                nin[e] = p.next_in ? load_as_f32(p.next_in, g, p.next_in_dt) : 0.f;
                nz[e] = (!p.next_in && dyn != DYN_ODE) ? p.noise[g] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (dyn == DYN_ODE) {
                mean[e] = x[e] + v[e] * dt;
            } else if (dyn == DYN_FLOW_SDE) {
                mean[e] = x[e] * c1 + v[e] * c2 * dt;
            } else if (dyn == DYN_DANCE_SDE) {
                const float x0 = x[e] - sigma * v[e];
                const float log_term = dance_k * (x[e] - x0 * one_m_sigma) / (sigma * sigma);
                mean[e] = x[e] + (v[e] + log_term) * dt;
            } else {
                const float x0 = x[e] - sigma * v[e];
                const float x1 = x[e] + v[e] * one_m_sigma;
                mean[e] = x0 * cps_a + x1 * cps_b;
            }
            if (p.next_in) nxt[e] = nin[e];
            else if (dyn == DYN_ODE) nxt[e] = mean[e];   // reference returns the unrounded mean; cast_latents rounds on store
            else nxt[e] = round_to_dtype(mean[e] + sv * nz[e], p.lat_dt);
            if (do_lp && e < cnt) {
                const float d = nxt[e] - mean[e];
                float lp;
                if (dyn == DYN_CPS) lp = -(d * d);
                else if (dyn == DYN_ODE) lp = 0.f;
                else lp = -(d * d) / den - log_sv - LOG_SQRT_2PI;
                lp_sum += lp;
            }
        }
        if (cnt == 4 && (p.n & 3) == 0) {
            if (p.next_out) store4(p.next_out, gi, p.next_out_dt, nxt);
            if (p.next_f32) store4(p.next_f32, gi, DT_F32, nxt);
            if (p.mean_out) store4(p.mean_out, gi, DT_F32, mean);
            if (p.noise_pred_out) store4(p.noise_pred_out, gi, DT_F32, v);
        } else {
            for (int e = 0; e < cnt; ++e) {
                if (p.next_out) store_from_f32(p.next_out, gi + e, p.next_out_dt, nxt[e]);
                if (p.next_f32) p.next_f32[gi + e] = nxt[e];
                if (p.mean_out) p.mean_out[gi + e] = mean[e];
                if (p.noise_pred_out) p.noise_pred_out[gi + e] = v[e];
            }
        }
    }
    if (chunk == 0 && tid == 0) {
        if (p.std_dev_t) p.std_dev_t[b] = std_dev;
        if (p.dt_out) p.dt_out[b] = dt;
    }
    if (!(do_lp && p.log_prob)) return;   // uniform per workgroup
    lp_sum = wave_sum(lp_sum);
    if ((tid & 63) == 0) red[tid >> 6] = lp_sum;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) s += red[w];
        // publish the partial, then take a ticket (agent-scope release / acquire; cdna guide G16 counter form)
        __hip_atomic_store(partial + (long)b * NCHUNK + chunk, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tk = __hip_atomic_fetch_add(ticket + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (tk == NCHUNK - 1);
        if (is_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            float tot = 0.f;
            for (int c = 0; c < NCHUNK; ++c)
                tot += __hip_atomic_load(partial + (long)b * NCHUNK + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p.log_prob[b] = tot / (float)p.n;
            __hip_atomic_store(ticket + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch on this slot
        }
    }
}

}  // namespace

namespace {
// ---------------------------------------------------------------------------------------------------------------------------
// UniPC multistep predictor-corrector (evaluation-mode sampling of the Wan adapters: reference scheduler/unipc_multistep.py:282-285 hands the
// step to diffusers' UniPCMultistepScheduler.step -- convert_model_output, multistep_uni_c_bh_update, multistep_uni_p_bh_update; the
// solver body is NOT in the reference tree, restated from its published algorithm: oracle/unipc_ref.py, parity unpinned).
// With flow sigmas and x0-prediction every update of the solver is a LINEAR combination of at most five tensors whose coefficients depend on
// the sigma schedule alone (mi355_flow/unipc.py computes them on the host): the stored sample, up to two stored x0-predictions, the new one.
//   unipc_convert:  x0 = sample - round_v(sigma * v),  v = CFG-combine(uncond, text) op by op in the network's dtype   (convert_model_output)
//   lincomb:        out = sum_i round_i(c_i * t_i)      round_i = to t_i's dtype (torch: a 0-dim fp32 scalar times a half tensor stays half);
//                   every partial sum is rounded to torch's PROMOTED dtype of the terms added so far (bf16 + bf16 stays bf16 and rounds,
//                   half + fp32 or fp16 + bf16 is fp32): with bf16 latent storage diffusers accumulates the update in bf16
// HBM-bound streaming kernels, 4 elements per thread, fp32 arithmetic in term order (no contraction: this file's compile flag).
struct LinCombParams { const void* t[5]; int dt[5]; float c[5]; int n_terms; void* out; int out_dt; long n4; };

__global__ __launch_bounds__(NT) void lincomb_kernel(LinCombParams p) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < p.n4; i += (long)gridDim.x * NT) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int prom = p.dt[0];                      // torch.promote_types over the terms so far: equal dtypes stay, anything else is fp32
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if (k < p.n_terms) {
                float v[4];
                load4(p.t[k], i * 4, p.dt[k], v);
                if (p.dt[k] != prom) prom = DT_F32;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float term = round_to_dtype(p.c[k] * v[e], p.dt[k]);
                    acc[e] = k == 0 ? term : round_to_dtype(acc[e] + term, prom);
                }
            }
        }
        store4(p.out, i * 4, p.out_dt, acc);
    }
}

struct UniPcConvertParams { const void* v_text; const void* v_uncond; int v_dt; float guidance; const void* sample; int sample_dt; float sigma; float* x0; long n4; };

__global__ __launch_bounds__(NT) void unipc_convert_kernel(UniPcConvertParams p) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < p.n4; i += (long)gridDim.x * NT) {
        float vt[4], vu[4], x[4], o[4];
        load4(p.v_text, i * 4, p.v_dt, vt);
        if (p.v_uncond) load4(p.v_uncond, i * 4, p.v_dt, vu);
        load4(p.sample, i * 4, p.sample_dt, x);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = p.v_uncond ? cfg_combine(vu[e], vt[e], p.guidance, p.v_dt) : vt[e];
            o[e] = x[e] - round_to_dtype(p.sigma * v, p.v_dt);
        }
        *(float4*)(p.x0 + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace

static float* g_scratch = nullptr;      // SCRATCH_SLOTS x (MAX_B*NCHUNK partials + MAX_B tickets)
static unsigned g_scratch_next = 0;

hipError_t launch_sde_step(const SdeStepParams& p, hipStream_t stream) {
    if (sched_trace_on()) {
        const size_t n = (size_t)p.B * p.n, vb = n * (p.v_dt == DT_F32 ? 4 : 2), lb = n * (p.lat_dt == DT_F32 ? 4 : 2);
        sched_trace_launch("sde_step", stream, {treg(p.v_text, vb), treg(p.v_uncond, p.v_uncond ? vb : 0), treg(p.latents, lb), treg(p.noise, p.noise ? n * 4 : 0),
                                                treg(p.next_in, p.next_in ? n * (p.next_in_dt == DT_F32 ? 4 : 2) : 0)},
                           {treg(p.next_out, p.next_out ? n * (p.next_out_dt == DT_F32 ? 4 : 2) : 0), treg(p.next_f32, p.next_f32 ? n * 4 : 0),
                            treg(p.mean_out, p.mean_out ? n * 4 : 0), treg(p.noise_pred_out, p.noise_pred_out ? n * 4 : 0),
                            treg(p.log_prob, p.log_prob ? (size_t)p.B * 4 : 0)});
    }
    if (p.B <= 0 || p.n <= 0 || p.B > MAX_B || p.v_dt < 0 || p.v_dt > 2) return hipErrorInvalidValue;
    constexpr size_t slot_words = (size_t)MAX_B * NCHUNK + MAX_B;
    if (!g_scratch) {
        // first use (never inside a stream capture: the rollout warms up eagerly before it is captured)
        hipError_t e = hipMalloc((void**)&g_scratch, SCRATCH_SLOTS * slot_words * 4);
        if (e != hipSuccess) return e;
        e = hipMemset(g_scratch, 0, SCRATCH_SLOTS * slot_words * 4);
        if (e != hipSuccess) return e;
    }
    // launches that may overlap (different streams) use different slots; same-stream launches serialise anyway.
    // A captured graph keeps the slot it was captured with: the tickets are self-resetting.
    float* slot = g_scratch + (size_t)(g_scratch_next++ % SCRATCH_SLOTS) * slot_words;
    hipLaunchKernelGGL(sde_step_kernel, dim3(NCHUNK, p.B), dim3(NT), 0, stream, p, slot, (unsigned*)(slot + (size_t)MAX_B * NCHUNK));
    return hipGetLastError();
}

hipError_t launch_lincomb(int n_terms, const void* const* t, const int* dt, const float* c, void* out, int out_dt, long n, hipStream_t stream) {
    if (n_terms < 1 || n_terms > 5 || n <= 0 || (n & 3) || !out || out_dt < 0 || out_dt > 2) return hipErrorInvalidValue;
    LinCombParams p;
    memset(&p, 0, sizeof(p));
    for (int k = 0; k < n_terms; ++k) {
        if (!t[k] || dt[k] < 0 || dt[k] > 2) return hipErrorInvalidValue;
        p.t[k] = t[k]; p.dt[k] = dt[k]; p.c[k] = c[k];
    }
    p.n_terms = n_terms; p.out = out; p.out_dt = out_dt; p.n4 = n >> 2;
    if (sched_trace_on()) {
        auto rg = [&](int k) { return treg(p.t[k], k < n_terms ? (size_t)n * (p.dt[k] == DT_F32 ? 4 : 2) : 0); };
        sched_trace_launch("lincomb", stream, {rg(0), rg(1), rg(2), rg(3), rg(4)}, {treg(out, (size_t)n * (out_dt == DT_F32 ? 4 : 2))});
    }
    const long g = (p.n4 + NT - 1) / NT;
    hipLaunchKernelGGL(lincomb_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(NT), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_unipc_convert(const void* v_text, const void* v_uncond, int v_dt, float guidance, const void* sample, int sample_dt, float sigma,
                                float* x0, long n, hipStream_t stream) {
    if (!v_text || !sample || !x0 || n <= 0 || (n & 3) || v_dt < 0 || v_dt > 2 || sample_dt < 0 || sample_dt > 2) return hipErrorInvalidValue;
    UniPcConvertParams p{v_text, v_uncond, v_dt, guidance, sample, sample_dt, sigma, x0, n >> 2};
    if (sched_trace_on())
        sched_trace_launch("unipc_convert", stream, {treg(v_text, (size_t)n * (v_dt == DT_F32 ? 4 : 2)), treg(v_uncond, v_uncond ? (size_t)n * (v_dt == DT_F32 ? 4 : 2) : 0),
                                                     treg(sample, (size_t)n * (sample_dt == DT_F32 ? 4 : 2))}, {treg(x0, (size_t)n * 4)});
    const long g = (p.n4 + NT - 1) / NT;
    hipLaunchKernelGGL(unipc_convert_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(NT), 0, stream, p);
    return hipGetLastError();
}

}  // namespace mi355
