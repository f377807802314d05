// mi355_flow -- HBM-bound kernels of the VAE decode: NHWC activations, bf16.
//   * latent ingest: NCHW storage dtype -> NHWC bf16 (channels zero-padded to 64), z / scaling + shift fused
//   * GroupNorm(+SiLU): two deterministic reduction passes (per-chunk channel partials -> per-(sample, channel) affine)
//     and one apply pass; every pass reads / writes whole 16-byte channel vectors of consecutive pixels (full lines)
//   * row softmax of the mid-block attention scores (fp32 in, bf16 probabilities out)
//   * conv / linear weight repack at bind time ([Co][Ci][kh][kw] -> [Co][tap][Ci_pad], K-contiguous for the implicit GEMM)
#include "kernels.h"

namespace mi355 {

namespace {

// ------------------------------------------------------------------------------------ ingest
__global__ __launch_bounds__(256) void vae_ingest_kernel(const void* lat, int dt, bf16_t* out, int C, int Cpad, long HW,
                                                         float scale, float shift) {
    // one thread per (pixel, 8-channel chunk); grid.y = sample
    const long b = blockIdx.y;
    const int chunks = Cpad / 8;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= HW * chunks) return;
    const long pix = idx / chunks;
    const int c0 = (int)(idx - pix * chunks) * 8;
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        y[e] = c < C ? load_as_f32(lat, (b * C + c) * HW + pix, dt) / scale + shift : 0.f;
    }
    *(uint4*)(out + (b * HW + pix) * Cpad + c0) =
        make_uint4(pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]), pack_bf16(y[4], y[5]), pack_bf16(y[6], y[7]));
}

// --------------------------------------------------------------------------------- GroupNorm
// pass 1: part[b][chunk][c][2] = (sum, sum of squares) of channel c over the chunk's pixels
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* x, float* part, long HW, int C, long rows_per_chunk) {
    __shared__ float red[256 * 16];
    const int lpr = C >> 3;                  // lanes per pixel row (C <= 2048)
    const int rpi = 256 / lpr;               // pixel rows per iteration
    const int slot = threadIdx.x % lpr, r0 = threadIdx.x / lpr;
    const long b = blockIdx.y, chunk = blockIdx.x;
    const long lo = chunk * rows_per_chunk;
    long hi = lo + rows_per_chunk; hi = hi < HW ? hi : HW;
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
    if (r0 < rpi) {
        const bf16_t* base = x + b * HW * C + slot * 8;
        auto acc8 = [&](const uint4 v) {
            const float f[8] = {bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y), bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w)};
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += f[e]; ss[e] += f[e] * f[e]; }
        };
        long r = lo + r0;
        // 4 independent 16-byte loads in flight per lane (the row order of the sums is unchanged)
        for (; r + 3L * rpi < hi; r += 4L * rpi) {
            const uint4 v0 = *(const uint4*)(base + r * C);
            const uint4 v1 = *(const uint4*)(base + (r + rpi) * C);
            const uint4 v2 = *(const uint4*)(base + (r + 2L * rpi) * C);
            const uint4 v3 = *(const uint4*)(base + (r + 3L * rpi) * C);
            acc8(v0); acc8(v1); acc8(v2); acc8(v3);
        }
        for (; r < hi; r += rpi) acc8(*(const uint4*)(base + r * C));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[threadIdx.x * 16 + e] = s[e]; red[threadIdx.x * 16 + 8 + e] = ss[e]; }
    __syncthreads();
    // fixed-order sum over the rpi row-lanes of each (channel, stat): deterministic
    float* dst = part + (b * gridDim.x + chunk) * (long)C * 2;
    for (int o = threadIdx.x; o < C * 2; o += 256) {
        const int c = o >> 1, st = o & 1;
        const int sl = c >> 3, e = c & 7;
        float a = 0.f;
        for (int r = 0; r < rpi; ++r) a += red[(r * lpr + sl) * 16 + st * 8 + e];
        dst[o] = a;
    }
}

// pass 2: one workgroup per (sample, group): mean / rstd in double (fixed summation order), then the per-channel affine
// a = rstd*gamma, d = beta - mean*a
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* part, int nchunk, int C, int groups, long HW, float eps,
                                                          const float* gamma, const float* beta, float* ad /*[B][2][C]*/) {
    __shared__ double red[2][4];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cpg = C / groups;
    double s = 0.0, ss = 0.0;
    const int n = nchunk * cpg;
    for (int i = tid; i < n; i += 256) {
        const int chunk = i / cpg, c = g * cpg + (i - chunk * cpg);
        const float2 v = *(const float2*)(part + (((long)b * nchunk + chunk) * C + c) * 2);
        s += (double)v.x; ss += (double)v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
    if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; }
    __syncthreads();
    s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    ss = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double cnt = (double)HW * cpg;
    const double mean = s / cnt;
    double var = ss / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = g * cpg + tid; c < (g + 1) * cpg; c += 256) {
        const float a = rstd * gamma[c];
        ad[((long)b * 2 + 0) * C + c] = a;
        ad[((long)b * 2 + 1) * C + c] = beta[c] - (float)mean * a;
    }
}

// pass 3: y = x*a + d, optional SiLU, bf16
template <bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* x, bf16_t* y, const float* ad, long HW, int C, long rows_per_chunk) {
    const int lpr = C >> 3, rpi = 256 / lpr;
    const int slot = threadIdx.x % lpr, r0 = threadIdx.x / lpr;
    if (r0 >= rpi) return;
    const long b = blockIdx.y, chunk = blockIdx.x;
    const long lo = chunk * rows_per_chunk;
    long hi = lo + rows_per_chunk; hi = hi < HW ? hi : HW;
    float a[8], d[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = ad[(b * 2 + 0) * C + slot * 8 + e]; d[e] = ad[(b * 2 + 1) * C + slot * 8 + e]; }
    const long off = b * HW * C + slot * 8;
    auto one = [&](const uint4 v) -> uint4 {
        float f[8] = {bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y), bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w)};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            f[e] = f[e] * a[e] + d[e];
            if (SILU) f[e] = silu_f(f[e]);
        }
        return make_uint4(pack_bf16(f[0], f[1]), pack_bf16(f[2], f[3]), pack_bf16(f[4], f[5]), pack_bf16(f[6], f[7]));
    };
    long r = lo + r0;
    for (; r + 3L * rpi < hi; r += 4L * rpi) {
        const uint4 v0 = *(const uint4*)(x + off + r * C);
        const uint4 v1 = *(const uint4*)(x + off + (r + rpi) * C);
        const uint4 v2 = *(const uint4*)(x + off + (r + 2L * rpi) * C);
        const uint4 v3 = *(const uint4*)(x + off + (r + 3L * rpi) * C);
        *(uint4*)(y + off + r * C) = one(v0);
        *(uint4*)(y + off + (r + rpi) * C) = one(v1);
        *(uint4*)(y + off + (r + 2L * rpi) * C) = one(v2);
        *(uint4*)(y + off + (r + 3L * rpi) * C) = one(v3);
    }
    for (; r < hi; r += rpi) *(uint4*)(y + off + r * C) = one(*(const uint4*)(x + off + r * C));
}

// --------------------------------------------------------------------------------- softmax
// one workgroup per row: p[row][:] = softmax(scale * s[row][:]) (bf16).  n % 4 == 0.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* s, bf16_t* p, int n, float scale_log2e) {
    __shared__ float red[8];
    const long row = blockIdx.x;
    const float4* src = (const float4*)(s + row * n);
    const int n4 = n >> 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 v = src[i];
        m = fmaxf(fmaxf(fmaxf(m, v.x), fmaxf(v.y, v.z)), v.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float mb = m * scale_log2e;
    float sum = 0.f;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 v = src[i];
        sum += exp2f(v.x * scale_log2e - mb) + exp2f(v.y * scale_log2e - mb) + exp2f(v.z * scale_log2e - mb) + exp2f(v.w * scale_log2e - mb);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    uint2* dst = (uint2*)(p + row * n);
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 v = src[i];
        dst[i] = make_uint2(pack_bf16(exp2f(v.x * scale_log2e - mb) * inv, exp2f(v.y * scale_log2e - mb) * inv),
                            pack_bf16(exp2f(v.z * scale_log2e - mb) * inv, exp2f(v.w * scale_log2e - mb) * inv));
    }
}

// softmax over the first n entries of rows with stride ld_s (fp32 scores) -> bf16 probabilities with stride ld_p; columns [n, ld_p) are
// written as zeros (they are the zero-padded K range of the following P.V GEMM).  ld_s % 4 == 0, ld_p % 4 == 0, any n <= ld_s.
__global__ __launch_bounds__(256) void softmax_rows_ld_kernel(const float* s, long ld_s, bf16_t* p, long ld_p, int n, float scale_log2e) {
    __shared__ float red[8];
    const long row = blockIdx.x;
    const float4* src = (const float4*)(s + row * ld_s);
    const int n4 = (n + 3) >> 2, p4 = (int)(ld_p >> 2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto load = [&](int i) {
        float4 v = src[i];
        const int c = i * 4;
        if (c + 1 >= n) v.y = -INFINITY;
        if (c + 2 >= n) v.z = -INFINITY;
        if (c + 3 >= n) v.w = -INFINITY;
        return v;
    };
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 v = load(i);
        m = fmaxf(fmaxf(fmaxf(m, v.x), fmaxf(v.y, v.z)), v.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float mb = m * scale_log2e;
    float sum = 0.f;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 v = load(i);
        sum += exp2f(v.x * scale_log2e - mb) + exp2f(v.y * scale_log2e - mb) + exp2f(v.z * scale_log2e - mb) + exp2f(v.w * scale_log2e - mb);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    uint2* dst = (uint2*)(p + row * ld_p);
    for (int i = threadIdx.x; i < p4; i += 256) {
        if (i < n4) {
            const float4 v = load(i);
            dst[i] = make_uint2(pack_bf16(exp2f(v.x * scale_log2e - mb) * inv, exp2f(v.y * scale_log2e - mb) * inv),
                                pack_bf16(exp2f(v.z * scale_log2e - mb) * inv, exp2f(v.w * scale_log2e - mb) * inv));
        } else {
            dst[i] = make_uint2(0u, 0u);
        }
    }
}

// ------------------------------------------------------------------------------------ video VAE (causal 3-D, Wan / Qwen-Image)
// latent ingest: (B, 16, T, h, w) storage dtype -> [B][T][h*w][Cpad] bf16 with the de-normalisation z*std + mean (wan2_t2v.py:217-226) and
// the 1x1x1 post_quant_conv (a 16x16 matrix per pixel, fp32) fused; channels >= C are zero.  One thread per pixel.
__global__ __launch_bounds__(256) void wvae_ingest_kernel(const void* lat, int dt, bf16_t* out, WvaeIngestParams q) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;        // over B*T*HW
    const long n = (long)q.B * q.T * q.HW;
    if (idx >= n) return;
    const long pix = idx % q.HW;
    const long bt = idx / q.HW;
    const int t = (int)(bt % q.T);
    const long b = bt / q.T;
    float z[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float v = 0.f;
        if (c < q.C) {
            v = load_as_f32(lat, ((b * q.C + c) * q.T + t) * q.HW + pix, dt);
            if (q.denorm) v = v / (1.0f / q.std[c]) + q.mean[c];
        }
        z[c] = v;
    }
    bf16_t* o = out + idx * q.Cpad;
    for (int co = 0; co < q.Cpad; co += 2) {
        float y[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 2; ++e)
            if (co + e < q.C) {
                float a = q.b_pq[co + e];
                for (int c = 0; c < q.C; ++c) a += q.w_pq[(co + e) * q.C + c] * z[c];
                y[e] = a;
            }
        *(unsigned*)(o + co) = pack_bf16(y[0], y[1]);
    }
}

// WanRMS_norm (+ SiLU): y = x / max(||x||_2, 1e-12) * sqrt(C) * gamma over the C real channels of one pixel row; gamma is zero on the
// padded channels.  One wave per row, 4-byte accesses (every wave instruction covers 256 contiguous bytes).
template <bool SILU>
__global__ __launch_bounds__(256) void wan_rms_kernel(const bf16_t* x, bf16_t* y, const float* gamma, long M, int C, int Cpad) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const bf16_t* row = x + m * Cpad;
    float ss = 0.f;
    for (int d = 2 * lane; d < Cpad; d += 128) {
        const unsigned u = *(const unsigned*)(row + d);
        ss += bf_lo(u) * bf_lo(u) + bf_hi(u) * bf_hi(u);
    }
    const float sc = sqrtf((float)C) / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
    for (int d = 2 * lane; d < Cpad; d += 128) {
        const unsigned u = *(const unsigned*)(row + d);
        const float2 g = *(const float2*)(gamma + d);
        float a = bf_lo(u) * sc * g.x, b = bf_hi(u) * sc * g.y;
        if (SILU) { a = a / (1.0f + __expf(-a)); b = b / (1.0f + __expf(-b)); }
        *(unsigned*)(y + m * Cpad + d) = pack_bf16(a, b);
    }
}

// temporal upsampler (WanResample 'upsample3d'): out[b][0] = x[b][0]; out[b][1 + 2t + s][pix][c] = tc[b][t][pix][s*C + c]
// x [B][T][HW][C], tc [B][T-1][HW][2C], out [B][2T-1][HW][C]; one thread per 16-byte chunk of out
__global__ __launch_bounds__(256) void frame_interleave_kernel(const bf16_t* x, const bf16_t* tc, bf16_t* out, int B, int T, long HW, int C) {
    const int cc = C >> 3;
    const long n = (long)B * (2 * T - 1) * HW * cc;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c8 = (int)(i % cc);
    long r = i / cc;
    const long pix = r % HW; r /= HW;
    const int f = (int)(r % (2 * T - 1));
    const long b = r / (2 * T - 1);
    const uint4* src;
    if (f == 0) src = (const uint4*)(x + ((b * T) * HW + pix) * C) + c8;
    else {
        const int t = (f - 1) >> 1, sx = (f - 1) & 1;
        src = (const uint4*)(tc + ((b * (T - 1) + t) * HW + pix) * (2L * C) + (long)sx * C) + c8;
    }
    ((uint4*)out)[i] = *src;
}

// ------------------------------------------------------------------------------ weight repack
__global__ __launch_bounds__(256) void conv_repack_kernel(const void* src, int dt, bf16_t* dst, int Co, int Ci, int Cpad, int taps) {
    const long n = (long)Co * taps * Cpad;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int ci = (int)(i % Cpad);
    const long t = i / Cpad;
    const int tap = (int)(t % taps), co = (int)(t / taps);
    dst[i] = ci < Ci ? f2bf(load_as_f32(src, ((long)co * Ci + ci) * taps + tap, dt)) : f2bf(0.f);
}

}  // namespace

hipError_t launch_vae_ingest(const void* lat, int dt, bf16_t* out, int B, int C, int Cpad, long HW, float scale, float shift,
                             hipStream_t st) {
    const long n = HW * (Cpad / 8);
    hipLaunchKernelGGL(vae_ingest_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, lat, dt, out, C, Cpad, HW, scale, shift);
    return hipGetLastError();
}

int gn_num_chunks(long HW, int C) {
    // ~128 KiB of activations per workgroup (>= 8 workgroups per CU once a sample exceeds ~256 MiB / B), at most 1024 chunks
    // per sample (the partial buffer is sized for that)
    const int rpi = 256 / (C >> 3);
    long rows = (1L << 17) / (C * 2);
    rows = (rows + rpi - 1) / rpi * rpi;
    long n = (HW + rows - 1) / rows;
    if (n > 1024) n = 1024;
    return (int)n;
}

hipError_t launch_group_norm(const bf16_t* x, bf16_t* y, const float* gamma, const float* beta, float* part, float* ad, int B, long HW,
                             int C, int groups, float eps, bool silu, hipStream_t st) {
    if (C % 8 || C > 2048 || 256 % (C >> 3) || C % groups) return hipErrorInvalidValue;
    const int nchunk = gn_num_chunks(HW, C);
    const int rpi = 256 / (C >> 3);
    long rpc = (HW + nchunk - 1) / nchunk;
    rpc = (rpc + rpi - 1) / rpi * rpi;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, B), dim3(256), 0, st, x, part, HW, C, rpc);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, B), dim3(256), 0, st, part, nchunk, C, groups, HW, eps, gamma, beta, ad);
    if (silu) hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(nchunk, B), dim3(256), 0, st, x, y, ad, HW, C, rpc);
    else hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(nchunk, B), dim3(256), 0, st, x, y, ad, HW, C, rpc);
    return hipGetLastError();
}

hipError_t launch_softmax_rows(const float* s, bf16_t* p, long rows, int n, float scale, hipStream_t st) {
    if (n % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, st, s, p, n, scale * 1.44269504088896340736f);
    return hipGetLastError();
}

hipError_t launch_softmax_rows_ld(const float* s, long ld_s, bf16_t* p, long ld_p, long rows, int n, float scale, hipStream_t st) {
    if ((ld_s & 3) || (ld_p & 3) || n < 1 || n > ld_s || n > ld_p) return hipErrorInvalidValue;
    hipLaunchKernelGGL(softmax_rows_ld_kernel, dim3((unsigned)rows), dim3(256), 0, st, s, ld_s, p, ld_p, n, scale * 1.44269504088896340736f);
    return hipGetLastError();
}

hipError_t launch_wvae_ingest(const void* lat, int dt, bf16_t* out, const WvaeIngestParams& q, hipStream_t st) {
    if (q.C < 1 || q.C > 16 || (q.Cpad & 1) || q.Cpad < q.C) return hipErrorInvalidValue;
    const long n = (long)q.B * q.T * q.HW;
    hipLaunchKernelGGL(wvae_ingest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, lat, dt, out, q);
    return hipGetLastError();
}

hipError_t launch_wan_rms(const bf16_t* x, bf16_t* y, const float* gamma, long M, int C, int Cpad, bool silu, hipStream_t st) {
    if (M <= 0 || (Cpad & 1) || C > Cpad) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((M + 3) / 4));
    if (silu) hipLaunchKernelGGL(wan_rms_kernel<true>, grid, dim3(256), 0, st, x, y, gamma, M, C, Cpad);
    else hipLaunchKernelGGL(wan_rms_kernel<false>, grid, dim3(256), 0, st, x, y, gamma, M, C, Cpad);
    return hipGetLastError();
}

hipError_t launch_frame_interleave(const bf16_t* x, const bf16_t* tc, bf16_t* out, int B, int T, long HW, int C, hipStream_t st) {
    if (T < 2 || (C & 7)) return hipErrorInvalidValue;
    const long n = (long)B * (2 * T - 1) * HW * (C >> 3);
    hipLaunchKernelGGL(frame_interleave_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, tc, out, B, T, HW, C);
    return hipGetLastError();
}

hipError_t launch_conv_repack(const void* src, int dt, bf16_t* dst, int Co, int Ci, int Cpad, int taps, hipStream_t st) {
    const long n = (long)Co * taps * Cpad;
    hipLaunchKernelGGL(conv_repack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dt, dst, Co, Ci, Cpad, taps);
    return hipGetLastError();
}

}  // namespace mi355
