// mi355_flow -- HBM-bound kernels of the MMDiT BACKWARD (the `optimize()` replay of
// reference src/flow_factory/trainers/grpo.py:185-342 needs d loss / d weights of the transformer the rollout ran).
// Everything here is a row / tile streaming kernel: 16-byte accesses, one wave per row where a row reduction is
// needed, fp32 arithmetic on bf16 activations / gradients.  The matrix work of the backward (dgrad / wgrad GEMMs,
// the two flash-attention backward passes) lives in gemm.hip / attention_bwd.hip.
#include <unordered_map>
#include "kernels.h"

namespace mi355 {
namespace {

__device__ __forceinline__ void unpack8(const uint4 u, float (&v)[8]) {
    v[0] = bf_lo(u.x); v[1] = bf_hi(u.x); v[2] = bf_lo(u.y); v[3] = bf_hi(u.y);
    v[4] = bf_lo(u.z); v[5] = bf_hi(u.z); v[6] = bf_lo(u.w); v[7] = bf_hi(u.w);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    return make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
}

// ------------------------------------------------------------------ LayerNorm + AdaLN-modulate backward
// forward (ln_mod_kernel):  xn = LN(x) * (1 + scale[b]) + shift[b]   [, xn2 = LN(x) * (1 + scale2[b]) + shift2[b]]
// backward w.r.t. x:        g  = dxn * (1 + scale) [+ dxn2 * (1 + scale2)]
//                           dx = rstd * (g - mean(g) - xhat * mean(g * xhat))          (no affine LN weight)
// dx is ADDED to dres (the residual-stream gradient arriving from later layers) when accumulate != 0.
// DMOD = false drops the modulation-gradient partials (4 x LN_MAXC x 8 registers) from the instantiation: the D = 5120 form (Wan2.1-14B,
// both Wan2.2-A14B experts; their modulation table is not trained through this kernel) would not fit its registers otherwise.
template <int LN_MAXC, bool DMOD = true>
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(LnModBwdParams p, int rows_per_wave) {
    const int lane = threadIdx.x & 63;
    const int first = (blockIdx.x * 4 + (threadIdx.x >> 6)) * rows_per_wave;
    if (first >= p.M) return;
    const int last = first + rows_per_wave < p.M ? first + rows_per_wave : p.M;
    const int nchunk = p.D >> 3;
    // optional per-sample gradients of the modulation vectors: column partials over this wave's rows live in registers and are flushed
    // with fp32 atomics when the sample changes / at the end (rows_per_wave x fewer atomics than elements; summation order not fixed)
    constexpr int AC = DMOD ? LN_MAXC : 1;
    float ash[AC][8], asc[AC][8], ash2[AC][8], asc2[AC][8];
    int cur_b = first / p.rows_per_sample;
    auto zero_acc = [&]() {
#pragma unroll
        for (int c = 0; c < AC; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) { ash[c][e] = 0.f; asc[c][e] = 0.f; ash2[c][e] = 0.f; asc2[c][e] = 0.f; }
    };
    auto flush = [&](int bb) {
        float* dm = p.dmod + (long)bb * p.mod_ld;
#pragma unroll
        for (int c = 0; c < AC; ++c) {
            const int ch = lane + c * 64;
            if (ch < nchunk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    unsafeAtomicAdd(dm + p.shift_off + ch * 8 + e, ash[c][e]);
                    unsafeAtomicAdd(dm + p.scale_off + ch * 8 + e, asc[c][e]);
                    if (p.dy2) {
                        unsafeAtomicAdd(dm + p.shift2_off + ch * 8 + e, ash2[c][e]);
                        unsafeAtomicAdd(dm + p.scale2_off + ch * 8 + e, asc2[c][e]);
                    }
                }
            }
        }
    };
    if (DMOD && p.dmod) zero_acc();
    for (int row = first; row < last; ++row) {
        const int b = row / p.rows_per_sample;
        if (DMOD && p.dmod && b != cur_b) { flush(cur_b); zero_acc(); cur_b = b; }
        const bf16_t* mod = p.mod + (long)b * p.mod_ld;
        float v[LN_MAXC][8], g[LN_MAXC][8];
        float sum = 0.f;
        const long ro = (long)row * p.D;
#pragma unroll
        for (int c = 0; c < LN_MAXC; ++c) {
            const int ch = lane + c * 64;
            if (ch < nchunk) {
                unpack8(*(const uint4*)(p.x + ro + ch * 8), v[c]);
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
                for (int e = 0; e < 8; ++e) { v[c][e] -= mean; sq += v[c][e] * v[c][e]; }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)p.D + p.eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int c = 0; c < LN_MAXC; ++c) {
            const int ch = lane + c * 64;
            if (ch < nchunk) {
                float d1[8], s1[8];
                unpack8(*(const uint4*)(p.dy + ro + ch * 8), d1);
                unpack8(*(const uint4*)(mod + p.scale_off + ch * 8), s1);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[c][e] *= rstd;                         // xhat
                    g[c][e] = d1[e] * (1.f + s1[e]);
                    if (DMOD && p.dmod) { ash[c % AC][e] += d1[e]; asc[c % AC][e] += d1[e] * v[c][e]; }
                }
                if (p.dy2) {
                    unpack8(*(const uint4*)(p.dy2 + ro + ch * 8), d1);
                    unpack8(*(const uint4*)(mod + p.scale2_off + ch * 8), s1);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        g[c][e] += d1[e] * (1.f + s1[e]);
                        if (DMOD && p.dmod) { ash2[c % AC][e] += d1[e]; asc2[c % AC][e] += d1[e] * v[c][e]; }
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sg += g[c][e];
                    sgx += g[c][e] * v[c][e];
                }
            }
        }
        const float mg = wave_sum(sg) / (float)p.D, mgx = wave_sum(sgx) / (float)p.D;
#pragma unroll
        for (int c = 0; c < LN_MAXC; ++c) {
            const int ch = lane + c * 64;
            if (ch < nchunk) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rstd * (g[c][e] - mg - v[c][e] * mgx);
                if (p.accumulate) {
                    float r[8];
                    unpack8(*(const uint4*)(p.dres + ro + ch * 8), r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += r[e];
                }
                *(uint4*)(p.dres + ro + ch * 8) = pack8(o);
            }
        }
    }
    if (DMOD && p.dmod) flush(cur_b);
}

// ------------------------------------------------------------------ gated residual backward
// forward (EPI_GATE_RES): x' = x + gate[b] * y.   dy[m][n] = gate[m / rps][n] * dx'[m][n]   (dx = dx' passes through)
__global__ __launch_bounds__(256) void gate_mul_kernel(const bf16_t* dx, const bf16_t* gate, long gate_ld, bf16_t* dy, long M, int D, int rps) {
    const int chunks = D >> 3;
    const long total = M * chunks;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / chunks;
        const int ch = (int)(i - m * chunks);
        float a[8], gt[8];
        unpack8(*(const uint4*)(dx + m * D + ch * 8), a);
        unpack8(*(const uint4*)(gate + (m / rps) * gate_ld + ch * 8), gt);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] *= gt[e];
        *(uint4*)(dy + m * D + ch * 8) = pack8(a);
    }
}

// gate_mul + d gate: workgroup = 64 rows x 256 columns-of-8 ... thread t owns 8 columns (one 16-byte chunk) of a 64-row slab of ONE sample
__global__ __launch_bounds__(256) void gate_bwd_kernel(const bf16_t* dx, const bf16_t* gate, long gate_ld, const bf16_t* y, bf16_t* dy, float* dgate,
                                                       long dg_ld, long M, int D, int rps) {
    const int chunks = D >> 3;
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= chunks) return;
    const long r0 = (long)blockIdx.y * 64, r1 = r0 + 64 < M ? r0 + 64 : M;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    long cur_b = r0 / rps;
    for (long m = r0; m < r1; ++m) {
        const long b = m / rps;
        if (b != cur_b) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { unsafeAtomicAdd(dgate + cur_b * dg_ld + ch * 8 + e, acc[e]); acc[e] = 0.f; }
            cur_b = b;
        }
        float a[8], gt[8], yy[8];
        unpack8(*(const uint4*)(dx + m * D + ch * 8), a);
        unpack8(*(const uint4*)(gate + b * gate_ld + ch * 8), gt);
        unpack8(*(const uint4*)(y + m * D + ch * 8), yy);
#pragma unroll
        for (int e = 0; e < 8; ++e) { acc[e] += a[e] * yy[e]; a[e] *= gt[e]; }
        *(uint4*)(dy + m * D + ch * 8) = pack8(a);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) unsafeAtomicAdd(dgate + cur_b * dg_ld + ch * 8 + e, acc[e]);
}

__global__ __launch_bounds__(256) void silu_bwd_kernel(const bf16_t* dy, const bf16_t* pre, bf16_t* dx, long n8) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float a[8], x[8];
        unpack8(*(const uint4*)(dy + i * 8), a);
        unpack8(*(const uint4*)(pre + i * 8), x);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sg = fast_rcp(1.0f + __expf(-x[e]));
            a[e] *= sg * (1.0f + x[e] * (1.0f - sg));
        }
        *(uint4*)(dx + i * 8) = pack8(a);
    }
}

__global__ __launch_bounds__(256) void f32_to_bf16_pad_kernel(const float* in, bf16_t* out, long rows, long rows_pad, long cols) {
    const long total = rows_pad * cols;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        out[i] = (i / cols) < rows ? f2bf(in[i]) : (bf16_t)0;
}

// hid = gelu_tanh(pre): recomputes the MLP hidden activation (the A operand of the ff2 weight gradient) from the stashed pre-activation
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* pre, bf16_t* out, long n8) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float a[8];
        unpack8(*(const uint4*)(pre + i * 8), a);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = gelu_tanh_f(a[e]);
        *(uint4*)(out + i * 8) = pack8(a);
    }
}

// ------------------------------------------------------------------ tile transposes (64 x 64 bf16 through LDS, +2 B row pad)
// in: rows x cols (row stride ld_in), batch stride bs_in;  out[c][r] (row stride ld_out), rows >= `rows` up to rows_pad are written as 0
// colpart != nullptr: the column sums of this 64-row tile are written to colpart[blockIdx.x][cols] as well (the bias gradient of a
// linear layer = column sums of dY, taken while dY streams through for its transpose; summed over the row tiles by colsum_finish)
// VEC: 16-byte global loads and stores (ld_in, ld_out multiples of 8 elements, 16-byte aligned bases, cols % 8 == 0 not required:
// a chunk past `cols` reads as zero); the scalar form covers everything else.
template <bool VEC>
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* in, long ld_in, long bs_in, bf16_t* out, long ld_out, long bs_out,
                                                        int rows, int cols, int rows_pad, float* colpart) {
    __shared__ bf16_t tile[64][72];           // 144-byte row pitch: 16-byte aligned rows, column reads spread over the banks
    __shared__ float csum[4][64];
    const int tr = blockIdx.x * 64, tc = blockIdx.y * 64;
    const bf16_t* ib = in + (long)blockIdx.z * bs_in;
    bf16_t* ob = out + (long)blockIdx.z * bs_out;
    if constexpr (VEC) {
        const int ch = threadIdx.x & 7, r0 = threadIdx.x >> 3;            // 8 chunks of 8 columns x 32 rows per pass
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = r0 + 32 * pass, gr = tr + r, gc = tc + ch * 8;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (gr < rows && gc + 8 <= cols) v = *(const uint4*)(ib + (long)gr * ld_in + gc);
            else if (gr < rows && gc < cols) {
                unsigned u[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (gc + e < cols) u[e >> 1] |= (unsigned)ib[(long)gr * ld_in + gc + e] << ((e & 1) * 16);
                v = make_uint4(u[0], u[1], u[2], u[3]);
            }
            // Round 6: the 8-column chunk index is XORed with the row's group (r >> 3).  The transposed read below has the 8 lanes of an output
            // line walk rows 8 apart -- 8 x 144 bytes = 288 dwords = 0 mod 32 banks: an 8-way conflict on every 2-byte read
            // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.78, LDS issue stalls 22 % of the kernel's wave cycles: profiles/r06f_pmc_*);
            // with the swizzle those lanes sit 4 dwords apart.  Pure layout: same values to the same places.
            *(uint4*)&tile[r][(ch ^ (r >> 3)) * 8] = v;
        }
        __syncthreads();
        if (colpart) {      // 4 threads per column, 16 rows each
            const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
            float cs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) cs += bf2f(tile[q * 16 + r][(c & 7) | ((((c >> 3) ^ ((q * 16 + r) >> 3)) & 7) << 3)]);
            csum[q][c] = cs;
        }
        // out row = input column c, 8 consecutive input rows per 16-byte store
        const int rc = threadIdx.x & 7, c0 = threadIdx.x >> 3;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int c = c0 + 32 * pass, gc = tc + c, gr = tr + rc * 8;
            if (gc < cols && gr < rows_pad) {
                unsigned u[4];
                const int cs_ = (c & 7) | ((((c >> 3) ^ rc) & 7) << 3);          // rows rc * 8 .. rc * 8 + 7 all have row group rc
#pragma unroll
                for (int e = 0; e < 4; ++e) u[e] = (unsigned)tile[rc * 8 + 2 * e][cs_] | ((unsigned)tile[rc * 8 + 2 * e + 1][cs_] << 16);
                if (gr + 8 <= rows_pad) *(uint4*)(ob + (long)gc * ld_out + gr) = make_uint4(u[0], u[1], u[2], u[3]);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (gr + e < rows_pad) ob[(long)gc * ld_out + gr + e] = (bf16_t)(u[e >> 1] >> ((e & 1) * 16));
                }
            }
        }
        if (colpart) {
            __syncthreads();
            if (threadIdx.x < 64 && tc + threadIdx.x < cols)
                colpart[(long)blockIdx.x * cols + tc + threadIdx.x] =
                    (csum[0][threadIdx.x] + csum[1][threadIdx.x]) + (csum[2][threadIdx.x] + csum[3][threadIdx.x]);
        }
    } else {
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 4 rows per pass
        float cs = 0.f;
#pragma unroll 4
        for (int r = ty; r < 64; r += 4) {
            const int gr = tr + r, gc = tc + tx;
            const bf16_t v = (gr < rows && gc < cols) ? ib[(long)gr * ld_in + gc] : (bf16_t)0;
            tile[r][tx] = v;
            cs += bf2f(v);
        }
        if (colpart) csum[ty][tx] = cs;
        __syncthreads();
        if (colpart && ty == 0 && tc + tx < cols)
            colpart[(long)blockIdx.x * cols + tc + tx] = (csum[0][tx] + csum[1][tx]) + (csum[2][tx] + csum[3][tx]);
#pragma unroll 4
        for (int c = ty; c < 64; c += 4) {
            const int gc = tc + c, gr = tr + tx;
            if (gc < cols && gr < rows_pad) ob[(long)gc * ld_out + gr] = tile[tx][c];
        }
    }
}

// ------------------------------------------------------------------ attention backward: prologue
// sum over the 8 lanes of a half DPP row (lanes 8 k .. 8 k + 7), in every lane: two quad swaps + the half-row mirror
__device__ __forceinline__ float half8_sum_dpp(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    using std::integral_constant;
    v += dpp(v, integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v += dpp(v, integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    v += dpp(v, integral_constant<int, 0x141>{});     // row_half_mirror
    return v;
}

// token-major o / do (image rows, then context rows, as the forward wrote o) -> per (b, h):
//   doh [B][H][S_pad][64] = do rows, delta [B][H][S_pad] = sum_d do * o (fp32), nld = -lse | -delta per 64-query tile (the dK/dV pass
//   moves a tile's 128 floats into LDS with one LDS-DMA instruction per wave and feeds them to its MFMA chains as C operands)
// one workgroup = 64 tokens of one (b, h); padded rows (S <= s < S_pad) are written as ZEROS (round 6: the scratch these three buffers live in is
// shared by attentions of different lengths -- SD3.5's joint and dual attention -- so "zero-initialised" did not stay true for the padding, and
// the software-pipelined backward passes carry no tail masks: attention_bwd.hip states the contract)
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(AttnBwdPrepParams p) {
    const int s0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int D = p.H * 64, n_ctx = p.S - p.n_img;
    const long bh = (long)b * p.H + h;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // 4 waves x 16 tokens: lane = d
    for (int r = w; r < 64; r += 4) {
        const int s = s0 + r;
        if (s >= p.S) {
            if (s < p.S_pad) {
                p.doh[(bh * p.S_pad + s) * 64 + lane] = 0;
                if (lane == 0) {
                    p.delta[bh * p.S_pad + s] = 0.f;
                    float* nl = p.nld + (bh * p.S_pad + (s & ~63)) * 2 + (s & 63);
                    nl[0] = 0.f; nl[64] = 0.f;
                }
            }
            continue;
        }
        const long row = (s < p.n_img) ? ((long)b * p.n_img + s) : ((long)b * n_ctx + (s - p.n_img));
        const bf16_t* dop = (s < p.n_img) ? p.do_img : p.do_ctx;
        const bf16_t* op = (s < p.n_img) ? p.o_img : p.o_ctx;
        const float dov = dop ? bf2f(dop[row * D + h * 64 + lane]) : 0.f;      // do_ctx == nullptr: the context output is unused (last block)
        const float ov = bf2f(op[row * D + h * 64 + lane]);
        p.doh[(bh * p.S_pad + s) * 64 + lane] = f2bf(dov);
        const float dl = wave_sum(dov * ov);
        if (lane == 0) {
            p.delta[bh * p.S_pad + s] = dl;
            float* nl = p.nld + (bh * p.S_pad + (s & ~63)) * 2 + (s & 63);
            nl[0] = -p.lse[bh * p.S_pad + s];
            nl[64] = -dl;
        }
    }
}

// (16-byte form of the kernel above: 8 lanes per token, Delta by a half-row DPP sum; selected with the fast gather below by mi355_tune_set(25, 1))
__global__ __launch_bounds__(256) void attn_bwd_prep_fast_kernel(AttnBwdPrepParams p) {
    const int s0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int D = p.H * 64, n_ctx = p.S - p.n_img;
    const long bh = (long)b * p.H + h;
    // 8 lanes per token (16 bytes = 8 features each), 32 tokens per pass: whole 128-byte head rows per half DPP row
    const int c = threadIdx.x & 7;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int s = s0 + pass * 32 + (threadIdx.x >> 3);
        if (s >= p.S) {                       // (uniform inside a half DPP row)
            if (s < p.S_pad) {
                *(uint4*)(p.doh + (bh * p.S_pad + s) * 64 + c * 8) = make_uint4(0u, 0u, 0u, 0u);
                if (c == 0) {
                    p.delta[bh * p.S_pad + s] = 0.f;
                    float* nl = p.nld + (bh * p.S_pad + (s & ~63)) * 2 + (s & 63);
                    nl[0] = 0.f; nl[64] = 0.f;
                }
            }
            continue;
        }
        const long row = (s < p.n_img) ? ((long)b * p.n_img + s) : ((long)b * n_ctx + (s - p.n_img));
        const bf16_t* dop = (s < p.n_img) ? p.do_img : p.do_ctx;
        const bf16_t* op = (s < p.n_img) ? p.o_img : p.o_ctx;
        const uint4 du = dop ? *(const uint4*)(dop + row * D + h * 64 + c * 8) : make_uint4(0u, 0u, 0u, 0u);   // do_ctx == nullptr: unused (last block)
        const uint4 ou = *(const uint4*)(op + row * D + h * 64 + c * 8);
        *(uint4*)(p.doh + (bh * p.S_pad + s) * 64 + c * 8) = du;
        float dv[8], ov[8];
        unpack8(du, dv); unpack8(ou, ov);
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) part += dv[e] * ov[e];
        const float dl = half8_sum_dpp(part);
        if (c == 0) {
            p.delta[bh * p.S_pad + s] = dl;
            float* nl = p.nld + (bh * p.S_pad + (s & ~63)) * 2 + (s & 63);
            nl[0] = -p.lse[bh * p.S_pad + s];
            nl[64] = -dl;
        }
    }
}

// ------------------------------------------------------------------ attention backward: epilogue
// per-head RMSNorm backward of q, k + gather of (dq, dk, dv) [B][H][S_pad][64] into token-major rows [tokens][3D] = [dq_pre | dk_pre | dv]
// of the image stream (s < n_img) and the context stream.  forward: y = x * r * w  (x = projection + bias, r = 1/rms over the head's 64
// features; for q, w already carries the folded softmax scale):  xhat = y / w,  g = dy * w,  dx = r * (g - xhat * mean(g * xhat)).
// One wave per token, lane = feature d, loop over heads.
// the same without the norm-weight partials (default gradient scope): one wave per token, 8 heads per pass, a head's 64 features = 8 lanes x
// 16 bytes = one half DPP row (the general kernel below moves 2 bytes per lane and access and pays two 64-lane ds_bpermute reductions per
// head: 73 us at B = 2, 1024^2).  Same arithmetic per element; the per-head sums are formed in another order.
__global__ __launch_bounds__(256) void rms_bwd_gather_fast_kernel(RmsBwdParams p) {
    const int lane = threadIdx.x & 63;
    const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= (long)p.B * p.S) return;
    const int hl = lane >> 3, d0 = (lane & 7) * 8;
    const int D = p.H * 64, n_ctx = p.S - p.n_img;
    const int b = (int)(tok / p.S), s = (int)(tok - (long)b * p.S);
    const bool img = s < p.n_img;
    const long row = img ? ((long)b * p.n_img + s) : ((long)b * n_ctx + (s - p.n_img));
    bf16_t* out = (img ? p.out_img : p.out_ctx) + row * (3L * D);
    const float* rstd = (img ? p.rstd_img : p.rstd_ctx) + row * (2L * p.H);
    const float* nq = (img ? p.nw_q : p.nw_cq) + d0;
    const float* nk = (img ? p.nw_k : p.nw_ck) + d0;
    float wq[8], wk[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { wq[e] = nq[e] * p.q_scale; wk[e] = nk[e]; }
    for (int h0 = 0; h0 < p.H; h0 += 8) {
        const int h = h0 + hl;
        if (h >= p.H) continue;               // (uniform inside a half DPP row)
        const long src = (((long)b * p.H + h) * p.S_pad + s) * 64 + d0;
        auto one = [&](const bf16_t* yv, const bf16_t* dyv, const float (&w)[8], float r, bf16_t* dst) {
            float y[8], dy[8], xh[8], g[8];
            unpack8(*(const uint4*)(yv + src), y);
            unpack8(*(const uint4*)(dyv + src), dy);
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xh[e] = fabsf(w[e]) > 1e-20f ? y[e] / w[e] : 0.f;
                g[e] = dy[e] * w[e];
                part += g[e] * xh[e];
            }
            const float mgx = half8_sum_dpp(part) * (1.0f / 64.0f);
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = r * (g[e] - xh[e] * mgx);
            *(uint4*)dst = pack8(o);
        };
        one(p.q, p.dq, wq, rstd[h], out + h * 64 + d0);
        one(p.k, p.dk, wk, rstd[p.H + h], out + D + h * 64 + d0);
        *(uint4*)(out + 2 * D + h * 64 + d0) = *(const uint4*)(p.dv + src);
    }
}

__global__ __launch_bounds__(256) void rms_bwd_gather_kernel(RmsBwdParams p, int tokens_per_wave) {
    __shared__ float red[4][4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long total = (long)p.B * p.S;
    const long t0 = ((long)blockIdx.x * 4 + wv) * tokens_per_wave;
    float dwq = 0.f, dwk = 0.f, dwcq = 0.f, dwck = 0.f;         // lane = feature d: sum of dy * xhat over this wave's tokens and all heads
    const int D = p.H * 64;
    const int n_ctx = p.S - p.n_img;
    for (long tok = t0; tok < t0 + tokens_per_wave && tok < total; ++tok) {
        const int b = (int)(tok / p.S), s = (int)(tok - (long)b * p.S);
        const bool img = s < p.n_img;
        const long row = img ? ((long)b * p.n_img + s) : ((long)b * n_ctx + (s - p.n_img));
        bf16_t* out = (img ? p.out_img : p.out_ctx) + row * (3L * D);
        const float* rstd = (img ? p.rstd_img : p.rstd_ctx) + row * (2L * p.H);
        const float wq = (img ? p.nw_q : p.nw_cq)[lane] * p.q_scale, wk = (img ? p.nw_k : p.nw_ck)[lane];
        float aq = 0.f, ak = 0.f;
        for (int h = 0; h < p.H; ++h) {
            const long src = (((long)b * p.H + h) * p.S_pad + s) * 64 + lane;
            {
                const float y = bf2f(p.q[src]), dy = bf2f(p.dq[src]);
                const float xh = fabsf(wq) > 1e-20f ? y / wq : 0.f;
                const float g = dy * wq;
                const float mgx = wave_sum(g * xh) * (1.0f / 64.0f);
                out[h * 64 + lane] = f2bf(rstd[h] * (g - xh * mgx));
                aq += dy * xh;
            }
            {
                const float y = bf2f(p.k[src]), dy = bf2f(p.dk[src]);
                const float xh = fabsf(wk) > 1e-20f ? y / wk : 0.f;
                const float g = dy * wk;
                const float mgx = wave_sum(g * xh) * (1.0f / 64.0f);
                out[D + h * 64 + lane] = f2bf(rstd[p.H + h] * (g - xh * mgx));
                ak += dy * xh;
            }
            out[2 * D + h * 64 + lane] = p.dv[src];
        }
        if (img) { dwq += aq; dwk += ak; } else { dwcq += aq; dwck += ak; }
    }
    if (p.dw_part) {      // (uniform per launch) fixed-order workgroup partials -> launch_rms_dw_finish
        red[wv][0][lane] = dwq; red[wv][1][lane] = dwk; red[wv][2][lane] = dwcq; red[wv][3][lane] = dwck;
        __syncthreads();
        const int which = threadIdx.x >> 6;
        p.dw_part[((long)blockIdx.x * 4 + which) * 64 + lane] = (red[0][which][lane] + red[1][which][lane]) + (red[2][which][lane] + red[3][which][lane]);
    }
}

__global__ __launch_bounds__(256) void rms_dw_finish_kernel(const float* part, int nwg, float q_scale, float* dw_q, float* dw_k, float* dw_cq, float* dw_ck) {
    const int which = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float s = 0.f;
    for (int i = 0; i < nwg; ++i) s += part[((long)i * 4 + which) * 64 + lane];
    // y = xhat * (q_scale * w): d/dw = q_scale * sum dy * xhat for the q weights
    float* dst = which == 0 ? dw_q : which == 1 ? dw_k : which == 2 ? dw_cq : dw_ck;
    if (dst) dst[lane] = (which == 0 || which == 2) ? s * q_scale : s;
}

// ------------------------------------------------------------------ bias gradient: db[n] (+)= sum_m dY[m][n]
// workgroup = 32 column groups (8 columns, one 16-byte load each) x 8 row lanes over one row slab; slab partials are summed in a fixed
// order by a second launch (colsum_finish): deterministic.  N % 8 == 0 and ld % 8 == 0 take the vector path.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* dy, long ld, long M, int N, float* part, int nslab) {
    __shared__ float red[8][256];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int n0 = blockIdx.x * 256 + cg * 8;
    const long per = (M + nslab - 1) / nslab;
    const long lo = blockIdx.y * per, hi = lo + per < M ? lo + per : M;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (n0 + 8 <= N && (ld & 7) == 0) {
        // four rows in flight per lane, added in row order (round 6: with the transposed copies gone this kernel takes the bias gradients of
        // the image stream -- one dependent 16-byte load per iteration left it latency-bound at 1.8 TB/s; same sums in the same order)
        long m = lo + rl;
        for (; m + 24 < hi; m += 32) {
            const uint4 u0 = *(const uint4*)(dy + m * ld + n0), u1 = *(const uint4*)(dy + (m + 8) * ld + n0);
            const uint4 u2 = *(const uint4*)(dy + (m + 16) * ld + n0), u3 = *(const uint4*)(dy + (m + 24) * ld + n0);
            float a[8];
            unpack8(u0, a);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += a[e];
            unpack8(u1, a);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += a[e];
            unpack8(u2, a);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += a[e];
            unpack8(u3, a);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += a[e];
        }
        for (; m < hi; m += 8) {
            float a[8];
            unpack8(*(const uint4*)(dy + m * ld + n0), a);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += a[e];
        }
    } else {
        for (long m = lo + rl; m < hi; m += 8)
            for (int e = 0; e < 8; ++e)
                if (n0 + e < N) s[e] += bf2f(dy[m * ld + n0 + e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = s[e];
    __syncthreads();
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n < N) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += red[r][threadIdx.x];
        part[(long)blockIdx.y * N + n] = t;
    }
}
// 32 columns x 8 slab lanes per workgroup: a lane sums every 8th partial (4 independent chains), the 8 lanes of a column combine in a
// fixed order through LDS.  (One thread per column walking all partials was a chain of `nslab` dependent L2 round trips: 31 us for the
// 128 row tiles of an 8192-row dY, 148 times per optimize() step.)
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* part, int nslab, int N, void* out_, int accumulate) {
    __shared__ float red[8][32];
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < N) {
        int i = r;
        for (; i + 24 < nslab; i += 32) {
            s0 += part[(long)i * N + n];
            s1 += part[(long)(i + 8) * N + n];
            s2 += part[(long)(i + 16) * N + n];
            s3 += part[(long)(i + 24) * N + n];
        }
        for (; i < nslab; i += 8) s0 += part[(long)i * N + n];
    }
    red[r][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (r == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][c];
        if (OUT_BF16) {
            bf16_t* out = (bf16_t*)out_;
            out[n] = f2bf(accumulate ? bf2f(out[n]) + t : t);
        } else {
            float* out = (float*)out_;
            out[n] = accumulate ? out[n] + t : t;
        }
    }
}

// split-K second stage: out[i] (+)= sum_s part[s][i], fixed order; 4 elements per thread and access (n % 4 == 0, 16-byte aligned parts) or
// one.  OUT_BF16: the sum is rounded to bf16 on the way out (a gradient buffer registered with mi355_*_set_grad_typed(.., MI355_BF16): the
// same value `fp32 buffer -> .to(bfloat16)` gives, without the fp32 round trip through HBM).
template <bool OUT_BF16, bool VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* part, long stride, int nsplit, void* out_, long n, int accumulate) {
    if (VEC) {
        const long n4 = n >> 2;
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < nsplit; ++k) {
                const float4 v = *(const float4*)(part + (long)k * stride + i * 4);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            if (OUT_BF16) {
                bf16_t* out = (bf16_t*)out_ + i * 4;
                if (accumulate) { s.x += bf2f(out[0]); s.y += bf2f(out[1]); s.z += bf2f(out[2]); s.w += bf2f(out[3]); }
                uint2 o;
                o.x = (unsigned)f2bf(s.x) | ((unsigned)f2bf(s.y) << 16);
                o.y = (unsigned)f2bf(s.z) | ((unsigned)f2bf(s.w) << 16);
                *(uint2*)out = o;
            } else {
                float4* out = (float4*)out_ + i;
                if (accumulate) { const float4 v = *out; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
                *out = s;
            }
        }
        return;
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += part[(long)k * stride + i];
        if (OUT_BF16) {
            bf16_t* out = (bf16_t*)out_;
            out[i] = f2bf(accumulate ? bf2f(out[i]) + s : s);
        } else {
            float* out = (float*)out_;
            out[i] = accumulate ? out[i] + s : s;
        }
    }
}

// Round 6: the two finishing kernels of one weight gradient in ONE launch -- the split-K reduction of dW (workgroups [0, nred)) and the
// fixed-order finish of the bias gradient's row-tile partials (the workgroups behind them).  The optimize() step of the default target set
// issues 191 such pairs; as two dependent launches each pair paid a second ~8 us launch for 1.5 K floats of output.  Same arithmetic, same
// order per output element as splitk_reduce_kernel<.., true> / colsum_finish_kernel: results are bit-identical to the two-launch form.
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void splitk_reduce_colsum_kernel(const float* part, long stride, int nsplit, void* out_, long n, int nred,
                                                                   const float* cs_part, int nslab, int N, void* cs_out_) {
    if ((int)blockIdx.x >= nred) {
        __shared__ float red[8][32];
        const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
        const int col = ((int)blockIdx.x - nred) * 32 + c;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (col < N) {
            int i = r;
            for (; i + 24 < nslab; i += 32) {
                s0 += cs_part[(long)i * N + col];
                s1 += cs_part[(long)(i + 8) * N + col];
                s2 += cs_part[(long)(i + 16) * N + col];
                s3 += cs_part[(long)(i + 24) * N + col];
            }
            for (; i < nslab; i += 8) s0 += cs_part[(long)i * N + col];
        }
        red[r][c] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (r == 0 && col < N) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k][c];
            if (OUT_BF16) ((bf16_t*)cs_out_)[col] = f2bf(t);
            else ((float*)cs_out_)[col] = t;
        }
        return;
    }
    const long n4 = n >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)nred * blockDim.x) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < nsplit; ++k) {
            const float4 v = *(const float4*)(part + (long)k * stride + i * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (OUT_BF16) {
            uint2 o;
            o.x = (unsigned)f2bf(s.x) | ((unsigned)f2bf(s.y) << 16);
            o.y = (unsigned)f2bf(s.z) | ((unsigned)f2bf(s.w) << 16);
            *(uint2*)((bf16_t*)out_ + i * 4) = o;
        } else {
            *((float4*)out_ + i) = s;
        }
    }
}

// ------------------------------------------------------------------ proj_out backward prologue: un-patchify transposed
// dv [B'][C][hp*p][wp*p] (fp32) -> dproj [B'*hp*wp][p*p*C] bf16, feature f = (pp*p + qq)*C + c  (the forward's EPI_UNPATCH order)
__global__ void unpatch_bwd_kernel(const float* dv, bf16_t* out, int Bp, int C, int hp, int wp, int patch) {
    const int NO = patch * patch * C;
    const long total = (long)Bp * hp * wp * NO;
    const int Himg = hp * patch, Wimg = wp * patch;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)(i % NO);
        const long tok = i / NO;
        const int tx = (int)(tok % wp), ty = (int)((tok / wp) % hp), b = (int)(tok / ((long)wp * hp));
        const int c = f % C, pq = f / C, pp = pq / patch, qq = pq - pp * patch;
        out[i] = f2bf(dv[(((long)b * C + c) * Himg + ty * patch + pp) * Wimg + tx * patch + qq]);
    }
}

// ------------------------------------------------------------------ SDE step backward (adjoint of sde_step_kernel w.r.t. the network output)
// inputs: upstream gradients g_lp [B] (d loss / d log_prob), g_np / g_mean [B][n] fp32 (d loss / d noise_pred, d next_latents_mean; optional);
// the step's own inputs (v_text / v_uncond bf16, x, x' = next_in) to re-evaluate x' - mean.  Output dv [n_cfg*B][n] fp32 in the forward-batch
// order [uncond, text]:  v = vu + g (vt - vu)  =>  dvt = g dv, dvu = (1 - g) dv  (bf16 roundings of the combine are treated as identity).
//   ODE:       mean = x + v dt                                   dmean/dv = dt,                     lp = 0
//   Flow-SDE:  mean = x c1 + v c2 dt                             dmean/dv = c2 dt,                  lp = mean_i[-(x'-mean)^2 / (2 sv^2)] + const
//   Dance-SDE: mean = x + (v + k (x sigma + sigma v (1-sigma)) / sigma^2) dt   dmean/dv = dt (1 + k (1-sigma)/sigma),  lp as Flow-SDE
//   CPS:       mean = (x - sigma v) a + (x + v (1-sigma)) bq     dmean/dv = -sigma a + (1-sigma) bq,  lp = mean_i[-(x'-mean)^2]
__global__ __launch_bounds__(256) void sde_step_bwd_kernel(SdeBwdParams p) {
    const int b = blockIdx.y;
    const float sigma = p.sigma[b * p.scalar_stride], sigma_next = p.sigma_next[b * p.scalar_stride];
    float eta = p.eta[b * p.scalar_stride];
    const int dyn = p.dynamics;
    if (dyn == DYN_ODE) eta = 0.f;
    const float dt = sigma_next - sigma;
    float c1 = 1.f, dm = dt, lpk = 0.f;       // mean = x * c1x(x-part, only needed to rebuild mean) ...; dm = dmean/dv; lpk: dlp/dmean_i = lpk * (x' - mean_i)
    float std_dev = 0.f, c2dt = dt, dance_k = 0.f, cps_a = 0.f, cps_b = 0.f;
    const float inv_n = 1.0f / (float)p.n;
    if (dyn == DYN_FLOW_SDE) {
        const float sden = (sigma == 1.0f) ? p.sigma_max : sigma;
        std_dev = sqrtf(sigma / (1.0f - sden)) * eta;
        const float s2 = std_dev * std_dev;
        c1 = 1.0f + s2 / (2.0f * sigma) * dt;
        c2dt = (1.0f + s2 * (1.0f - sigma) / (2.0f * sigma)) * dt;
        dm = c2dt;
        const float sv = std_dev * sqrtf(-dt);
        lpk = sv > 0.f ? 1.0f / (sv * sv) : 0.f;          // d/dmean [-(x'-mean)^2 / (2 sv^2)] = (x'-mean) / sv^2
    } else if (dyn == DYN_DANCE_SDE) {
        std_dev = eta;
        dance_k = 0.5f * eta * eta;
        dm = dt * (1.0f + dance_k * (1.0f - sigma) / sigma);
        const float sv = std_dev * sqrtf(-dt);
        lpk = sv > 0.f ? 1.0f / (sv * sv) : 0.f;
    } else if (dyn == DYN_CPS) {
        std_dev = sigma_next * sinf(eta * 1.57079632679f);
        cps_a = 1.0f - sigma_next;
        cps_b = sqrtf(fmaxf(sigma_next * sigma_next - std_dev * std_dev, 0.f));
        dm = -sigma * cps_a + (1.0f - sigma) * cps_b;
        lpk = 2.0f;                                        // d/dmean [-(x'-mean)^2] = 2 (x'-mean)
    }
    const float glp = (p.g_lp && p.compute_log_prob) ? p.g_lp[b] * inv_n : 0.f;
    const long base = (long)b * p.n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < p.n; i += (long)gridDim.x * blockDim.x) {
        const long gi = base + i;
        float v = bf2f(p.v_text[gi]);
        if (p.v_uncond) {
            const float u = bf2f(p.v_uncond[gi]);
            v = round_bf16(u + round_bf16(p.guidance * round_bf16(v - u)));
        }
        const float x = load_as_f32(p.latents, gi, p.lat_dt);
        float mean;
        if (dyn == DYN_ODE) mean = x + v * dt;
        else if (dyn == DYN_FLOW_SDE) mean = x * c1 + v * c2dt;
        else if (dyn == DYN_DANCE_SDE) {
            const float x0 = x - sigma * v;
            mean = x + (v + dance_k * (x - x0 * (1.0f - sigma)) / (sigma * sigma)) * dt;
        } else {
            mean = (x - sigma * v) * cps_a + (x + v * (1.0f - sigma)) * cps_b;
        }
        float gmean = p.g_mean ? p.g_mean[gi] : 0.f;
        if (glp != 0.f && p.next_in) gmean += glp * lpk * (load_as_f32(p.next_in, gi, p.next_in_dt) - mean);
        float dv = gmean * dm + (p.g_np ? p.g_np[gi] : 0.f);
        if (p.v_uncond) {
            p.dv[gi] = (1.0f - p.guidance) * dv;                       // uncond half first: forward batch order [negative, positive]
            p.dv[(long)p.B * p.n + gi] = p.guidance * dv;
        } else {
            p.dv[gi] = dv;
        }
    }
}

inline int grid_for(long total, int block) {
    long g = (total + block - 1) / block;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace

// ---- gradient-buffer dtype registry -------------------------------------------------------------------------------------------
// The engines' training files keep `float*` gradient pointers (fp32 is the default contract of mi355_*_set_grad).  A buffer registered with
// mi355_*_set_grad_typed(.., MI355_BF16) is marked here, and the kernels that FINISH a weight / bias gradient (split-K reduce, column-sum
// finish) round their fp32 sum to bf16 on the way out: the value `fp32 buffer -> .to(bfloat16)` gives, without 10 bytes per parameter of HBM
// round trip (fp32 write, fp32 read, bf16 write -> 2 bytes).  Host-side state, touched only by the thread that launches the backward.
static std::unordered_map<const void*, int> g_grad_buf_dt;
void grad_buf_mark(const void* p, int dt) {
    if (!p) return;
    if (dt == DT_F32) g_grad_buf_dt.erase(p);
    else g_grad_buf_dt[p] = dt;
}
int grad_buf_dtype(const void* p) {
    if (g_grad_buf_dt.empty()) return DT_F32;
    auto it = g_grad_buf_dt.find(p);
    return it == g_grad_buf_dt.end() ? DT_F32 : it->second;
}
int grad_buf_esize(const void* p) { return grad_buf_dtype(p) == DT_F32 ? 4 : 2; }

// mi355_tune_set(26, .): the weight-gradient GEMMs (+ their split-K reductions) of the backward on a side stream of the plan's training state.
// 1 (default) = in the head_dim-128 engines (FLUX.1 / Qwen-Image / Wan, train_common.h; read when a plan's training state is created), 0 = nowhere
// (a third value, the SD3.5 engine as well, was an opt-in through round 4 and is gone: see the end of this comment).  Measured on MI355X, same box, optimize() step of the reference's default target modules at B = 1, 1024^2
// (profiles/r04i_*): FLUX.1 271.3 -> 249.1 ms, Qwen-Image (true CFG) 521.2 -> 468.7 ms -- their 3072 x 3072 gradients are 144 output tiles x 2
// splits = 288 workgroups on 256 CUs, two half-empty rounds that the dgrad / attention-backward chain fills.  SD3.5-medium (1536 x 1536: 144
// tiles of 128 x 128, two per CU) at B = 2, 1024^2 (profiles/r04j_*): attention projections trainable 91.6 -> 89.1 ms, every block linear
// trainable 110.2 -> 112.6 ms -- no consistent gain; re-measured in round 5 on the reference's default target set (86.5 / 86.7 vs 87.2 / 86.2 ms; 512^2:
// 31.4 vs 32.5 ms, profiles/r05h_*) and removed from that engine.  Bit-identical either way (same kernels, operands, order).
static int g_wgrad_side = 1;
void set_wgrad_side(int v) { g_wgrad_side = v != 0 ? 1 : 0; }      // (value 2 -- the SD3.5 engine too -- measured neutral twice and was removed in round 5)
int get_wgrad_side() { return g_wgrad_side; }

// mi355_tune_set(28, .): 1 (default) = the text chain of the Qwen-Image backward and of the FLUX.1 double blocks' backward (MLP backward, out-projection dgrad | join | joint attention
// backward | fork | q|k|v producer backward, dgrad, norm backward, and its weight-gradient operand transposes) on the plan's side stream -- the
// stream and events that carry the text chain of the forward (key 12, same size rule) -- beside the image chain; 0 = in line.  Measured on
// MI355X (profiles/r04l_*, Qwen-Image 60 layers, true CFG, B = 1, optimize() step in line -> side): 512^2 164.4 -> 141.1 ms, 1024^2 470.2 ->
// 470.0 ms; FLUX.1-dev B = 1 (profiles/r04q_*): 512^2 120.2 -> 116.0 ms, 1024^2 256.5 -> 253.2 ms.
// Bit-identical either way.
static int g_train_text_side = 1;
void set_train_text_side(int v) { g_train_text_side = v != 0; }
int get_train_text_side() { return g_train_text_side; }

// Split-K factor of a weight-gradient GEMM.  mi355_tune_set(27, .): 0 = the round-2 rule (about 768 tiles of 128 x 128 in flight), 1 (default)
// = the smallest modelled time over s = 1 .. 16: launch_simple runs 256 x 256 tiles, one workgroup per CU, when N >= 256 and there are >= 128
// of them, else 128 x 128 tiles, two per CU; a launch takes ceil(tiles * s / slots) rounds of M_pad / s contraction rows (20.75 ns per row and
// 256 x 256 tile, 7.2 ns per 128 x 128 tile: measured, profiles/r04h_qwen_train_step_kernel_stats.txt) and its reduction streams s partial
// copies at ~5 TB/s.  The model describes a GEMM that has the GPU to itself: `overlapped` launches (the side-stream schedule of key 26, whose
// partial rounds are filled by the other stream) keep the round-2 rule.  A/B on MI355X (profiles/r04k_*, optimize() step, ms, rule -> model):
// SD3.5 B = 2 1024^2 serial: attention projections 92.4 -> 91.7, every block linear 111.5 -> 108.3; overlapped FLUX.1 249.5 -> 252.6 and
// Qwen-Image 471.2 -> 479.6 (more partial-sum traffic for rounds that were not idle): hence the exception.
static int g_wgrad_split_model = 1;      // 0 = the round-2 rule, 1 = modelled-time minimum (default), 2 = never split (diagnostic: tests/test_gpu_fullsize.py)
void set_wgrad_split_model(int v) { g_wgrad_split_model = v; }
int wgrad_split(int N, int K, int M_pad, size_t part_floats, bool overlapped) {
    if (g_wgrad_split_model == 2) return 1;
    const int nt = M_pad / 64;
    int cap = nt / 2 < 16 ? nt / 2 : 16;
    if (cap < 1) cap = 1;
    while (cap > 1 && (size_t)cap * N * K > part_floats) --cap;
    if (!g_wgrad_split_model || overlapped) {
        const long tiles = (long)((N + 127) / 128) * ((K + 127) / 128);
        int split = (int)((768 + tiles - 1) / tiles);
        if (split > cap) split = cap;
        return split < 1 ? 1 : split;
    }
    const long t256 = (long)((N + 255) / 256) * ((K + 255) / 256), t128 = (long)((N + 127) / 128) * ((K + 127) / 128);
    const bool big = K >= 256 && t256 >= 128;
    const double tiles = big ? (double)t256 : (double)t128, slots = big ? 256.0 : 512.0, ns_row = big ? 20.75 : 7.2;
    int best = 1;
    double best_t = 1e30;
    for (int s = 1; s <= cap; ++s) {
        const double rounds = (double)(long)((tiles * s + slots - 1) / slots);
        const double gemm = rounds * ns_row * (double)M_pad / s;                                    // ns
        const double red = s > 1 ? 3000.0 + (double)s * N * K * 4.0 / 5000.0 : 0.0;                // ns (5 TB/s = 5000 bytes / ns)
        const double t = gemm + red;
        if (t < best_t * 0.999) { best_t = t; best = s; }
    }
    return best;
}

hipError_t launch_ln_mod_bwd(const LnModBwdParams& p, hipStream_t st) {
    if (sched_trace_on()) {
        const size_t xb = (size_t)p.M * p.D * 2, nb = (size_t)((p.M + p.rows_per_sample - 1) / p.rows_per_sample);
        const TraceRegion mod = treg(p.mod, ((nb - 1) * p.mod_ld + (size_t)(p.scale_off > p.scale2_off ? p.scale_off : p.scale2_off) + p.D) * 2);
        sched_trace_launch("ln_mod_bwd", st, {treg(p.x, xb), treg(p.dy, xb), treg(p.dy2, p.dy2 ? xb : 0), mod, treg(p.dres, p.accumulate ? xb : 0)},
                           {treg(p.dres, xb), treg(p.dmod, p.dmod ? (nb * p.mod_ld) * 4 : 0)});
    }
    // rows up to 4096 wide with modulation gradients; up to 6144 wide without (D = 5120: the 40-head Wan transformers)
    if (p.D % 8 != 0 || p.D > (p.dmod ? 8 : 12) * 64 * 8 || p.M <= 0) return hipErrorInvalidValue;
    const int rpw = p.dmod ? 16 : 1;         // rows per wave: 16 with the modulation-gradient partials in registers
    const int grid = (p.M + 4 * rpw - 1) / (4 * rpw);
    // Without modulation gradients (the reference's default target sets: no trainable tensor upstream of the modulation table) the
    // DMOD = false instantiation: the 4 x LN_MAXC x 8 partial-sum registers of the other form cost it its occupancy even when unused
    // (round 6: one wave per row, 8192 latency-bound waves per launch -- resident waves per SIMD are what hides their four reductions)
    if (!p.dmod && p.D <= 2048) hipLaunchKernelGGL((ln_mod_bwd_kernel<4, false>), dim3(grid), dim3(256), 0, st, p, rpw);
    else if (!p.dmod && p.D <= 4096) hipLaunchKernelGGL((ln_mod_bwd_kernel<8, false>), dim3(grid), dim3(256), 0, st, p, rpw);
    else if (p.D <= 2048) hipLaunchKernelGGL(ln_mod_bwd_kernel<4>, dim3(grid), dim3(256), 0, st, p, rpw);
    else if (p.D <= 4096) hipLaunchKernelGGL(ln_mod_bwd_kernel<8>, dim3(grid), dim3(256), 0, st, p, rpw);
    else hipLaunchKernelGGL((ln_mod_bwd_kernel<12, false>), dim3(grid), dim3(256), 0, st, p, rpw);
    return hipGetLastError();
}

hipError_t launch_gate_bwd(const bf16_t* dx, const bf16_t* gate, long gate_ld, const bf16_t* y, bf16_t* dy, float* dgate, long dg_ld, long M, int D,
                           int rps, hipStream_t st) {
    if (sched_trace_on()) {
        const size_t nb = (size_t)((M + rps - 1) / rps);
        sched_trace_launch("gate_bwd", st, {treg(dx, (size_t)M * D * 2), treg(gate, ((nb - 1) * gate_ld + D) * 2), treg(y, (size_t)M * D * 2)},
                           {treg(dy, (size_t)M * D * 2), treg(dgate, ((nb - 1) * dg_ld + D) * 4)});
    }
    if (D % 8 || M <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(((D >> 3) + 255) / 256, (unsigned)((M + 63) / 64)), dim3(256), 0, st, dx, gate, gate_ld, y, dy, dgate, dg_ld,
                       M, D, rps);
    return hipGetLastError();
}

hipError_t launch_silu_bwd(const bf16_t* dy, const bf16_t* pre, bf16_t* dx, long n, hipStream_t st) {
    if (n % 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(grid_for(n >> 3, 256)), dim3(256), 0, st, dy, pre, dx, n >> 3);
    return hipGetLastError();
}

hipError_t launch_f32_to_bf16_pad(const float* in, bf16_t* out, long rows, long rows_pad, long cols, hipStream_t st) {
    hipLaunchKernelGGL(f32_to_bf16_pad_kernel, dim3(grid_for(rows_pad * cols, 256)), dim3(256), 0, st, in, out, rows, rows_pad, cols);
    return hipGetLastError();
}

hipError_t launch_gate_mul(const bf16_t* dx, const bf16_t* gate, long gate_ld, bf16_t* dy, long M, int D, int rps, hipStream_t st) {
    if (sched_trace_on()) sched_trace_launch("gate_mul", st, {treg(dx, (size_t)M * D * 2), treg(gate, ((size_t)((M + rps - 1) / rps - 1) * gate_ld + D) * 2)}, {treg(dy, (size_t)M * D * 2)});
    if (D % 8 || M <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gate_mul_kernel, dim3(grid_for(M * (D >> 3), 256)), dim3(256), 0, st, dx, gate, gate_ld, dy, M, D, rps);
    return hipGetLastError();
}

hipError_t launch_gelu_fwd(const bf16_t* pre, bf16_t* out, long n, hipStream_t st) {
    if (sched_trace_on()) sched_trace_launch("gelu_fwd", st, {treg(pre, (size_t)n * 2)}, {treg(out, (size_t)n * 2)});
    if (n % 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n >> 3, 256)), dim3(256), 0, st, pre, out, n >> 3);
    return hipGetLastError();
}

static bool transpose_vec_ok(const bf16_t* in, long ld_in, long bs_in, const bf16_t* out, long ld_out, long bs_out) {
    return ((ld_in | ld_out | bs_in | bs_out) & 7) == 0 && (((size_t)in | (size_t)out) & 15) == 0;
}

hipError_t launch_transpose(const bf16_t* in, long ld_in, long bs_in, bf16_t* out, long ld_out, long bs_out, int rows, int cols, int rows_pad,
                            int batch, hipStream_t st) {
    if (sched_trace_on()) sched_trace_launch("transpose", st, {tregs(in, ((size_t)(rows - 1) * ld_in + cols) * 2, (size_t)bs_in * 2, (size_t)batch)},
                                             {tregs(out, ((size_t)(cols - 1) * ld_out + rows_pad) * 2, (size_t)bs_out * 2, (size_t)batch)});
    if (rows <= 0 || cols <= 0 || batch <= 0 || rows_pad < rows) return hipErrorInvalidValue;
    const dim3 grid((rows_pad + 63) / 64, (cols + 63) / 64, batch);
    if (transpose_vec_ok(in, ld_in, bs_in, out, ld_out, bs_out))
        hipLaunchKernelGGL(transpose_kernel<true>, grid, dim3(256), 0, st, in, ld_in, bs_in, out, ld_out, bs_out, rows, cols, rows_pad, (float*)nullptr);
    else
        hipLaunchKernelGGL(transpose_kernel<false>, grid, dim3(256), 0, st, in, ld_in, bs_in, out, ld_out, bs_out, rows, cols, rows_pad, (float*)nullptr);
    return hipGetLastError();
}

// transpose + column sums in one pass over `in`: out = in^T, colsum[cols] = sum over rows (scratch: ((rows_pad + 63) / 64) * cols floats)
hipError_t launch_transpose_colsum(const bf16_t* in, long ld_in, bf16_t* out, long ld_out, int rows, int cols, int rows_pad, float* scratch,
                                   float* colsum, hipStream_t st, bool finish) {
    if (sched_trace_on()) sched_trace_launch("transpose_colsum", st, {treg(in, ((size_t)(rows - 1) * ld_in + cols) * 2)},
                                             {treg(out, ((size_t)(cols - 1) * ld_out + rows_pad) * 2), treg(scratch, (size_t)((rows_pad + 63) / 64) * cols * 4),
                                              treg(colsum, finish ? (size_t)cols * grad_buf_esize(colsum) : 0)});
    if (rows <= 0 || cols <= 0 || rows_pad < rows || !scratch || !colsum) return hipErrorInvalidValue;
    const int ntile = (rows_pad + 63) / 64;
    if (transpose_vec_ok(in, ld_in, 0, out, ld_out, 0))
        hipLaunchKernelGGL(transpose_kernel<true>, dim3(ntile, (cols + 63) / 64, 1), dim3(256), 0, st, in, ld_in, 0L, out, ld_out, 0L, rows, cols,
                           rows_pad, scratch);
    else
        hipLaunchKernelGGL(transpose_kernel<false>, dim3(ntile, (cols + 63) / 64, 1), dim3(256), 0, st, in, ld_in, 0L, out, ld_out, 0L, rows, cols,
                           rows_pad, scratch);
    if (!finish) return hipGetLastError();      // the caller finishes the column sums inside its split-K reduction launch (launch_splitk_reduce_colsum)
    if (grad_buf_dtype(colsum) == DT_BF16) hipLaunchKernelGGL(colsum_finish_kernel<true>, dim3((cols + 31) / 32), dim3(256), 0, st, scratch, ntile, cols, (void*)colsum, 0);
    else hipLaunchKernelGGL(colsum_finish_kernel<false>, dim3((cols + 31) / 32), dim3(256), 0, st, scratch, ntile, cols, (void*)colsum, 0);
    return hipGetLastError();
}

// mi355_tune_set(25, .): 1 (default since round 4) = the 16-byte-access forms of attn_bwd_prep and of the default-scope RMSNorm-backward gather
// (optimize() step 94.2 -> 91.2 ms at B = 2, 1024^2).  They round in another order than the general kernels; verified at full width against the
// oracle's autograd AND its bf16-emulating band (profiles/r04a_*: worst tensor 5.14e-2 vs fp32 with them, 5.25e-2 without; band 3.1-3.3e-2).
// 0 = the general kernels (always used when norm-weight gradients are requested).
static int g_rms_bwd_fast = 1;
void set_rms_bwd_fast(int v) { g_rms_bwd_fast = v != 0; }
hipError_t launch_attn_bwd_prep(const AttnBwdPrepParams& p, hipStream_t st) {
    if (sched_trace_on()) {
        const size_t D2 = (size_t)p.H * 128, ri = (size_t)p.B * p.n_img, rc = (size_t)p.B * (p.S - p.n_img), bhs = (size_t)p.B * p.H * p.S_pad;
        sched_trace_launch("attn_bwd_prep", st, {treg(p.o_img, ri * D2), treg(p.o_ctx, p.o_ctx ? rc * D2 : 0), treg(p.do_img, ri * D2), treg(p.do_ctx, p.do_ctx ? rc * D2 : 0), treg(p.lse, bhs * 4)},
                           {treg(p.doh, bhs * 128), treg(p.delta, bhs * 4), treg(p.nld, bhs * 8)});
    }
    if (g_rms_bwd_fast) hipLaunchKernelGGL(attn_bwd_prep_fast_kernel, dim3((p.S + 63) / 64, p.H, p.B), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((p.S + 63) / 64, p.H, p.B), dim3(256), 0, st, p);
    return hipGetLastError();
}

// tokens per wave: 1 normally, 8 when the norm-weight partials are wanted (8 x fewer partial rows to sum)
int rms_bwd_grid(int B, int S) { return (int)(((long)B * S + 31) / 32); }
hipError_t launch_rms_bwd_gather(const RmsBwdParams& p, hipStream_t st) {
    if (sched_trace_on()) {
        const size_t bhs = (size_t)p.B * p.H * p.S_pad * 128, D6 = (size_t)p.H * 64 * 3 * 2, ri = (size_t)p.B * p.n_img, rc = (size_t)p.B * (p.S - p.n_img);
        sched_trace_launch("rms_bwd_gather", st, {treg(p.q, bhs), treg(p.k, bhs), treg(p.dq, bhs), treg(p.dk, bhs), treg(p.dv, bhs), treg(p.rstd_img, ri * 2 * p.H * 4),
                                                  treg(p.rstd_ctx, p.rstd_ctx ? rc * 2 * p.H * 4 : 0)},
                           {treg(p.out_img, ri * D6), treg(p.out_ctx, p.out_ctx ? rc * D6 : 0), treg(p.dw_part, p.dw_part ? (size_t)rms_bwd_grid(p.B, p.S) * 4 * 64 * 4 : 0)});
    }
    const long tokens = (long)p.B * p.S;
    if (p.dw_part) hipLaunchKernelGGL(rms_bwd_gather_kernel, dim3((unsigned)rms_bwd_grid(p.B, p.S)), dim3(256), 0, st, p, 8);
    else if (g_rms_bwd_fast) hipLaunchKernelGGL(rms_bwd_gather_fast_kernel, dim3((unsigned)((tokens + 3) / 4)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(rms_bwd_gather_kernel, dim3((unsigned)((tokens + 3) / 4)), dim3(256), 0, st, p, 1);
    return hipGetLastError();
}

hipError_t launch_rms_dw_finish(const float* part, int nwg, float q_scale, float* dw_q, float* dw_k, float* dw_cq, float* dw_ck, hipStream_t st) {
    hipLaunchKernelGGL(rms_dw_finish_kernel, dim3(1), dim3(256), 0, st, part, nwg, q_scale, dw_q, dw_k, dw_cq, dw_ck);
    return hipGetLastError();
}

hipError_t launch_colsum(const bf16_t* dy, long ld, long M, int N, float* scratch, float* out, int accumulate, hipStream_t st, int* deferred_nslab) {
    if (M <= 0 || N <= 0 || !scratch || !out) return hipErrorInvalidValue;
    if (sched_trace_on())       // accumulate: `out` is read as well (a read-after-write dependency the happens-before checker must see)
        sched_trace_launch("colsum", st, {treg(dy, ((size_t)(M - 1) * ld + N) * 2), treg(out, accumulate && !deferred_nslab ? (size_t)N * grad_buf_esize(out) : 0)},
                           {treg(scratch, (size_t)64 * N * 4), treg(out, deferred_nslab ? 0 : (size_t)N * grad_buf_esize(out))});
    const int nslab = (int)(M >= 8192 ? 64 : (M + 127) / 128);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 255) / 256, nslab), dim3(256), 0, st, dy, ld, M, N, scratch, nslab);
    if (deferred_nslab) { *deferred_nslab = nslab; return hipGetLastError(); }
    if (grad_buf_dtype(out) == DT_BF16) hipLaunchKernelGGL(colsum_finish_kernel<true>, dim3((N + 31) / 32), dim3(256), 0, st, scratch, nslab, N, (void*)out, accumulate);
    else hipLaunchKernelGGL(colsum_finish_kernel<false>, dim3((N + 31) / 32), dim3(256), 0, st, scratch, nslab, N, (void*)out, accumulate);
    return hipGetLastError();
}

hipError_t launch_splitk_reduce(const float* part, long stride, int nsplit, float* out, long n, int accumulate, hipStream_t st) {
    if (nsplit <= 0 || n <= 0 || !part || !out) return hipErrorInvalidValue;
    if (sched_trace_on())
        sched_trace_launch("splitk_reduce", st, {treg(part, ((size_t)(nsplit - 1) * stride + n) * 4), treg(out, accumulate ? (size_t)n * grad_buf_esize(out) : 0)}, {treg(out, (size_t)n * grad_buf_esize(out))});
    const bool b16 = grad_buf_dtype(out) == DT_BF16;
    const bool vec = !(n & 3) && !(stride & 3) && !((size_t)part & 15) && !((size_t)out & 15);
    const dim3 grid(grid_for(vec ? n / 4 : n, 256));
    if (b16 && vec) hipLaunchKernelGGL((splitk_reduce_kernel<true, true>), grid, dim3(256), 0, st, part, stride, nsplit, (void*)out, n, accumulate);
    else if (b16) hipLaunchKernelGGL((splitk_reduce_kernel<true, false>), grid, dim3(256), 0, st, part, stride, nsplit, (void*)out, n, accumulate);
    else if (vec) hipLaunchKernelGGL((splitk_reduce_kernel<false, true>), grid, dim3(256), 0, st, part, stride, nsplit, (void*)out, n, accumulate);
    else hipLaunchKernelGGL((splitk_reduce_kernel<false, false>), grid, dim3(256), 0, st, part, stride, nsplit, (void*)out, n, accumulate);
    return hipGetLastError();
}

hipError_t launch_colsum_finish(const float* scratch, int nslab, int cols, float* colsum, hipStream_t st) {
    if (!scratch || !colsum || nslab <= 0 || cols <= 0) return hipErrorInvalidValue;
    if (sched_trace_on()) sched_trace_launch("colsum_finish", st, {treg(scratch, (size_t)nslab * cols * 4)}, {treg(colsum, (size_t)cols * grad_buf_esize(colsum))});
    if (grad_buf_dtype(colsum) == DT_BF16) hipLaunchKernelGGL(colsum_finish_kernel<true>, dim3((cols + 31) / 32), dim3(256), 0, st, scratch, nslab, cols, (void*)colsum, 0);
    else hipLaunchKernelGGL(colsum_finish_kernel<false>, dim3((cols + 31) / 32), dim3(256), 0, st, scratch, nslab, cols, (void*)colsum, 0);
    return hipGetLastError();
}

// out[n] = sum_s part[s][n] (overwritten) AND colsum[cols] = sum over the `nslab` row-tile partials launch_transpose_colsum(.., finish = false)
// left in `cs_scratch`, one launch.  Returns hipErrorNotSupported when the pair does not qualify (ragged sizes, mixed buffer dtypes): the caller
// then issues the two launches.
hipError_t launch_splitk_reduce_colsum(const float* part, long stride, int nsplit, float* out, long n, const float* cs_scratch, int nslab, int cols,
                                       float* colsum, hipStream_t st) {
    if (nsplit <= 0 || n <= 0 || !part || !out || !cs_scratch || !colsum || nslab <= 0 || cols <= 0) return hipErrorInvalidValue;
    const bool b16 = grad_buf_dtype(out) == DT_BF16;
    const bool vec = !(n & 3) && !(stride & 3) && !((size_t)part & 15) && !((size_t)out & 15);
    if (!vec || (grad_buf_dtype(colsum) == DT_BF16) != b16) return hipErrorNotSupported;
    if (sched_trace_on())
        sched_trace_launch("splitk_reduce_colsum", st, {treg(part, ((size_t)(nsplit - 1) * stride + n) * 4), treg(cs_scratch, (size_t)nslab * cols * 4)},
                           {treg(out, (size_t)n * grad_buf_esize(out)), treg(colsum, (size_t)cols * grad_buf_esize(colsum))});
    const int nred = (int)grid_for(n / 4, 256);
    const dim3 grid(nred + (cols + 31) / 32);
    if (b16) hipLaunchKernelGGL(splitk_reduce_colsum_kernel<true>, grid, dim3(256), 0, st, part, stride, nsplit, (void*)out, n, nred, cs_scratch, nslab, cols, (void*)colsum);
    else hipLaunchKernelGGL(splitk_reduce_colsum_kernel<false>, grid, dim3(256), 0, st, part, stride, nsplit, (void*)out, n, nred, cs_scratch, nslab, cols, (void*)colsum);
    return hipGetLastError();
}

hipError_t launch_unpatch_bwd(const float* dv, bf16_t* out, int Bp, int C, int hp, int wp, int patch, hipStream_t st) {
    if (sched_trace_on()) sched_trace_launch("unpatch_bwd", st, {treg(dv, (size_t)Bp * hp * wp * patch * patch * C * 4)}, {treg(out, (size_t)Bp * hp * wp * patch * patch * C * 2)});
    const long total = (long)Bp * hp * wp * patch * patch * C;
    hipLaunchKernelGGL(unpatch_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, dv, out, Bp, C, hp, wp, patch);
    return hipGetLastError();
}

hipError_t launch_sde_step_bwd(const SdeBwdParams& p, hipStream_t st) {
    if (sched_trace_on()) sched_trace_launch("sde_step_bwd", st, {treg(p.v_text, (size_t)p.B * p.n * 2), treg(p.v_uncond, p.v_uncond ? (size_t)p.B * p.n * 2 : 0)},
                                             {treg(p.dv, (size_t)(p.v_uncond ? 2 : 1) * p.B * p.n * 4)});
    if (p.B <= 0 || p.n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sde_step_bwd_kernel, dim3(grid_for(p.n, 256) > 256 ? 256 : grid_for(p.n, 256), p.B), dim3(256), 0, st, p);
    return hipGetLastError();
}

}  // namespace mi355
