// mi355_flow -- HBM-bound helper kernels of the MMDiT forward (ops K0-prologue, K1, K4, K12 of
// SURVEY.md 2.3): LayerNorm+AdaLN-modulate, patchify (im2col of the k2/s2 conv), pos-embed crop,
// sinusoidal timestep projection, dtype conversion.  All 16-byte vectorised, one wave per row
// where a row reduction is needed.
#include "kernels.h"

namespace mi355 {
namespace {

// ------------------------------------------------------------------ LayerNorm + modulate
// One wave per row; the row (D <= 8*64*MAXC) stays in registers: exact two-pass mean/variance.
// LN_MAXC 16-byte chunks per lane: 4 -> D <= 2048 (SD3.5, D = 1536), 8 -> D <= 4096 (FLUX.1, D = 3072), 12 -> D <= 6144 (the 40-head Wan
// transformers, D = 5120: Wan2.1-14B and both Wan2.2-A14B experts)
template <int LN_MAXC>
__global__ __launch_bounds__(256) void ln_mod_kernel(LnModParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const int nchunk = p.D >> 3;  // 8 bf16 per 16-byte chunk
    float v[LN_MAXC][8];
    float sum = 0.f;
    const bf16_t* xr = p.x + (long)row * p.D;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            const uint4 u = *(const uint4*)(xr + ch * 8);
            v[c][0] = bf_lo(u.x); v[c][1] = bf_hi(u.x); v[c][2] = bf_lo(u.y); v[c][3] = bf_hi(u.y);
            v[c][4] = bf_lo(u.z); v[c][5] = bf_hi(u.z); v[c][6] = bf_lo(u.w); v[c][7] = bf_hi(u.w);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += v[c][e];
        }
    }
    const float mean = wave_sum(sum) / (float)p.D;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; sq += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)p.D + p.eps);
    const int b = row / p.rows_per_sample;
    const bf16_t* mod = p.mod + (long)b * p.mod_ld;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
            float nv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) nv[e] = (v[c][e] - mean) * rstd;
            {
                const uint4 sc = *(const uint4*)(mod + p.scale_off + ch * 8);
                const uint4 sh = *(const uint4*)(mod + p.shift_off + ch * 8);
                uint4 o;
                o.x = pack_bf16(nv[0] * (1.f + bf_lo(sc.x)) + bf_lo(sh.x), nv[1] * (1.f + bf_hi(sc.x)) + bf_hi(sh.x));
                o.y = pack_bf16(nv[2] * (1.f + bf_lo(sc.y)) + bf_lo(sh.y), nv[3] * (1.f + bf_hi(sc.y)) + bf_hi(sh.y));
                o.z = pack_bf16(nv[4] * (1.f + bf_lo(sc.z)) + bf_lo(sh.z), nv[5] * (1.f + bf_hi(sc.z)) + bf_hi(sh.z));
                o.w = pack_bf16(nv[6] * (1.f + bf_lo(sc.w)) + bf_lo(sh.w), nv[7] * (1.f + bf_hi(sc.w)) + bf_hi(sh.w));
                *(uint4*)(p.out + (long)row * p.D + ch * 8) = o;
            }
            if (p.out2) {
                const uint4 sc = *(const uint4*)(mod + p.scale2_off + ch * 8);
                const uint4 sh = *(const uint4*)(mod + p.shift2_off + ch * 8);
                uint4 o;
                o.x = pack_bf16(nv[0] * (1.f + bf_lo(sc.x)) + bf_lo(sh.x), nv[1] * (1.f + bf_hi(sc.x)) + bf_hi(sh.x));
                o.y = pack_bf16(nv[2] * (1.f + bf_lo(sc.y)) + bf_lo(sh.y), nv[3] * (1.f + bf_hi(sc.y)) + bf_hi(sh.y));
                o.z = pack_bf16(nv[4] * (1.f + bf_lo(sc.z)) + bf_lo(sh.z), nv[5] * (1.f + bf_hi(sc.z)) + bf_hi(sh.z));
                o.w = pack_bf16(nv[6] * (1.f + bf_lo(sc.w)) + bf_lo(sh.w), nv[7] * (1.f + bf_hi(sc.w)) + bf_hi(sh.w));
                *(uint4*)(p.out2 + (long)row * p.D + ch * 8) = o;
            }
        }
    }
}

// ------------------------------------------------------------------ patchify (im2col, k = c*p*p + py*p + px)
__global__ void patchify_kernel(const void* lat, int dt, bf16_t* out, int B, int rep, int C, int h, int w, int p) {
    const int hp = h / p, wp = w / p, K = C * p * p;
    const long total = (long)B * rep * hp * wp * K;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const long tok = i / K;
        const int tx = (int)(tok % wp);
        const int ty = (int)((tok / wp) % hp);
        const int bb = (int)(tok / ((long)wp * hp));
        const int b = bb % B;  // cat([latents, latents]): replica r uses sample bb - r*B
        const int c = k / (p * p), py = (k / p) % p, px = k % p;
        const long src = (((long)b * C + c) * h + ty * p + py) * w + tx * p + px;
        out[i] = f2bf(load_as_f32(lat, src, dt));
    }
}

__global__ void pos_crop_kernel(const bf16_t* pos, bf16_t* out, int max_size, int hp, int wp, int D) {
    const int top = (max_size - hp) / 2, left = (max_size - wp) / 2;
    const int chunks = D >> 3;
    const long total = (long)hp * wp * chunks;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % chunks);
        const long tok = i / chunks;
        const int tx = (int)(tok % wp), ty = (int)(tok / wp);
        const long src = ((long)(top + ty) * max_size + left + tx) * D + ch * 8;
        *(uint4*)(out + tok * D + ch * 8) = *(const uint4*)(pos + src);
    }
}

// diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]
__global__ void time_proj_kernel(const float* t, int rows, int dim, int t_round_dt, bf16_t* out) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * half) return;
    const int r = i / half, j = i % half;
    const float tv = round_to_dtype(t[r], t_round_dt);
    const float freq = expf(-9.210340371976184f * (float)j / (float)half);  // -ln(10000) * j / half
    const float a = tv * freq;
    out[(long)r * dim + j] = f2bf(cosf(a));
    out[(long)r * dim + half + j] = f2bf(sinf(a));
}

__global__ void convert_kernel(const void* src, int sdt, void* dst, int ddt, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = load_as_f32(src, i, sdt);
        store_from_f32(dst, i, ddt, v);  // fp16 destination clamps at +-65504 (cast_latents)
    }
}

inline int grid_for(long total, int block) {
    long g = (total + block - 1) / block;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

hipError_t launch_ln_mod(const LnModParams& p, hipStream_t stream) {
    if (sched_trace_on()) {
        const size_t xb = (size_t)p.M * p.D * 2, nb = (size_t)((p.M + p.rows_per_sample - 1) / (p.rows_per_sample > 0 ? p.rows_per_sample : 1));
        const size_t mb = nb ? ((nb - 1) * p.mod_ld + p.D) * 2 : 0;
        sched_trace_launch("ln_mod", stream, {treg(p.x, xb), treg(p.mod + p.shift_off, mb), treg(p.mod + p.scale_off, mb),
                                              treg(p.out2 ? p.mod + p.shift2_off : nullptr, mb), treg(p.out2 ? p.mod + p.scale2_off : nullptr, mb)},
                           {treg(p.out, xb), treg(p.out2, p.out2 ? xb : 0)});
    }
    if (p.D % 8 != 0 || p.D > 12 * 64 * 8 || p.M <= 0) return hipErrorInvalidValue;
    if (p.D <= 2048) hipLaunchKernelGGL(ln_mod_kernel<4>, dim3((p.M + 3) / 4), dim3(256), 0, stream, p);
    else if (p.D <= 4096) hipLaunchKernelGGL(ln_mod_kernel<8>, dim3((p.M + 3) / 4), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(ln_mod_kernel<12>, dim3((p.M + 3) / 4), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_patchify(const void* lat, int dt, bf16_t* patches, int B, int rep, int C, int h, int w, int p,
                           hipStream_t stream) {
    if (sched_trace_on())
        sched_trace_launch("patchify", stream, {treg(lat, (size_t)B * C * h * w * (dt == DT_F32 ? 4 : 2))}, {treg(patches, (size_t)B * rep * C * h * w * 2)});
    const long total = (long)B * rep * (h / p) * (w / p) * C * p * p;
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, lat, dt, patches, B, rep, C, h,
                       w, p);
    return hipGetLastError();
}

hipError_t launch_pos_crop(const bf16_t* pos, bf16_t* out, int max_size, int hp, int wp, int D, hipStream_t stream) {
    if (hp > max_size || wp > max_size || D % 8) return hipErrorInvalidValue;
    const long total = (long)hp * wp * (D >> 3);
    hipLaunchKernelGGL(pos_crop_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, pos, out, max_size, hp, wp, D);
    return hipGetLastError();
}

hipError_t launch_time_proj(const float* t, int rows, int dim, int t_round_dt, bf16_t* out, hipStream_t stream) {
    if (sched_trace_on()) sched_trace_launch("time_proj", stream, {treg(t, (size_t)rows * 4)}, {treg(out, (size_t)rows * dim * 2)});
    const int total = rows * (dim / 2);
    hipLaunchKernelGGL(time_proj_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, t, rows, dim, t_round_dt, out);
    return hipGetLastError();
}

// One wave that samples {s_memtime (core clocks, slows down with the power throttle), s_memrealtime (100 MHz wall clock)} every
// ~sleep_iters * 8 k core clocks: launched on a side stream it sits in one wave slot beside the kernels under test and shows the core
// clock they are actually delivered (profiles/r02_power_clock_notes.txt: 1.44 GHz inside the bf16 GEMM at the 1.35 kW cap, not the 2.4 GHz
// the roofline peaks are quoted at).  Measurement tool (scripts/clock_under_load.py), never on the product path.
__global__ void clock_probe_kernel(long long* out, int n, int sleep_iters) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < n; ++i) {
        const long long c = clock64(), w = wall_clock64();
        out[2 * i] = c;
        out[2 * i + 1] = w;
        for (int k = 0; k < sleep_iters; ++k) __builtin_amdgcn_s_sleep(127);
    }
}

hipError_t launch_clock_probe(long long* out, int n, int sleep_iters, hipStream_t stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, stream, out, n, sleep_iters);
    return hipGetLastError();
}

hipError_t launch_convert(const void* src, int src_dt, void* dst, int dst_dt, long n, hipStream_t stream) {
    if (sched_trace_on())
        sched_trace_launch("convert", stream, {treg(src, (size_t)n * (src_dt == DT_F32 ? 4 : 2))}, {treg(dst, (size_t)n * (dst_dt == DT_F32 ? 4 : 2))});
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(convert_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, src, src_dt, dst, dst_dt, n);
    return hipGetLastError();
}

}  // namespace mi355
