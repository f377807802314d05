// mi355_flow -- bf16 MFMA GEMM for gfx950 with fused epilogues (ops K0,K2,K3,K5,K6,K8,K9,K11,K13
// of SURVEY.md 2.3).
//
//   C[m][n] = sum_k A[m][k] * W[n][k]      A:[M][K], W:[N][K] (torch Linear weight layout), bf16
//
// Structure (cdna_hip_programming.md section 5, "glds, 2 LDS buffers, BK=64"):
//   * block tile BM x BN x 64, WM x WN waves, each wave owns a (BM/WM) x 64 sub-tile;
//   * operands go HBM/L2 -> LDS with global_load_lds_dwordx4 (1 KiB per wave-instruction, LDS
//     image lane-linear); the 16-byte chunk index inside each 128-byte row is XOR-swizzled with
//     (row>>1)&7 on the SOURCE address and again on the ds_read_b128 address, which makes every
//     ds_read_b128 lane group hit 16 distinct 16-byte bank slots (conflict-free);
//   * 2 LDS stages: tile t+1 is in flight while tile t feeds v_mfma_f32_16x16x32_bf16;
//   * MFMA operands are swapped (A-operand = W fragment, B-operand = activation fragment), so a
//     lane ends up holding 4 CONSECUTIVE output columns of one row: 8-byte packed stores, and the
//     per-head (64 column) RMSNorm of the q/k epilogue needs only 2 cross-lane adds per row;
//   * workgroup -> tile mapping is XCD-aware (8 XCDs, private L2 each): consecutive tiles of one
//     activation row-panel stay on one XCD.
#include "kernels.h"

namespace mi355 {

namespace {

constexpr int BK = 64;

template <int BM, int BN, int WM, int WN>
struct Cfg {
    static constexpr int NW = WM * WN;
    static constexpr int NT = NW * 64;
    static constexpr int TM = BM / WM;
    static constexpr int TN = BN / WN;
    static constexpr int MI = TM / 16;
    static constexpr int NI = TN / 16;
    static constexpr int A_BYTES = BM * BK * 2;
    static constexpr int W_BYTES = BN * BK * 2;
    static constexpr int STAGE = A_BYTES + W_BYTES;
    static constexpr int GA = BM / 8 / NW;  // 8-row glds groups per wave (A)
    static constexpr int GW = BN / 8 / NW;
    static_assert(TN == 64, "wave N extent must equal head_dim (q/k RMSNorm epilogue)");
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "glds group split");
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    // 64 lanes x 16 B -> LDS [base + lane*16]; base must be wave-uniform
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// ---- epilogue: called once per (row m, 4 consecutive columns n..n+3) ------------------------
template <int EPI>
__device__ __forceinline__ void epi_store(const GemmParams& p, int m, int n, const float (&v)[4], float rstd) {
    if (m >= p.M) return;
    if constexpr (EPI == EPI_VT) {
        // m = feature (row bias), n = token
        const float b = p.bias[m];
        const int h = m >> 6, d = m & 63;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tok = n + r;
            if (tok < p.N) {
                const int bi = tok / p.rows_per_sample;
                const int s = tok - bi * p.rows_per_sample + p.s_off;
                p.q[(((long)bi * p.H + h) * 64 + d) * p.S_pad + s] = f2bf(v[r] + b);
            }
        }
        return;
    } else {
        if (n >= p.N) return;  // N % 4 == 0 for all column-bias epilogues
        const float4 bb = *(const float4*)(p.bias + n);
        float y[4] = {v[0] + bb.x, v[1] + bb.y, v[2] + bb.z, v[3] + bb.w};
        if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_SILU || EPI == EPI_BIAS_GELU) {
            if constexpr (EPI == EPI_BIAS_SILU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = silu_f(round_bf16(y[r]));
            }
            if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = gelu_tanh_f(y[r]);
            }
            uint2 o = {pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3])};
            *(uint2*)(p.out + (long)m * p.ldo + n) = o;
        } else if constexpr (EPI == EPI_POSADD || EPI == EPI_ADDSRC_SILU) {
            const long arow = (long)(m % p.rows_per_sample);
            const uint2 a = *(const uint2*)(p.aux + arow * p.ld_aux + n);
            y[0] += bf_lo(a.x); y[1] += bf_hi(a.x); y[2] += bf_lo(a.y); y[3] += bf_hi(a.y);
            if constexpr (EPI == EPI_ADDSRC_SILU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = silu_f(round_bf16(y[r]));
            }
            uint2 o = {pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3])};
            *(uint2*)(p.out + (long)m * p.ldo + n) = o;
        } else if constexpr (EPI == EPI_GATE_RES) {
            const int bi = m / p.rows_per_sample;
            const uint2 g = *(const uint2*)(p.aux + (long)bi * p.ld_aux + n);
            bf16_t* xp = p.out + (long)m * p.ldo + n;
            const uint2 x = *(const uint2*)xp;
            y[0] = bf_lo(x.x) + bf_lo(g.x) * y[0];
            y[1] = bf_hi(x.x) + bf_hi(g.x) * y[1];
            y[2] = bf_lo(x.y) + bf_lo(g.y) * y[2];
            y[3] = bf_hi(x.y) + bf_hi(g.y) * y[3];
            uint2 o = {pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3])};
            *(uint2*)xp = o;
        } else if constexpr (EPI == EPI_QK_NORM) {
            const int D = p.H * 64;
            const bool is_k = n >= D;
            const int nn = is_k ? n - D : n;
            const int h = nn >> 6, d = nn & 63;
            const float4 w = *(const float4*)((is_k ? p.nw_k : p.nw_q) + d);
            const int bi = m / p.rows_per_sample;
            const int s = m - bi * p.rows_per_sample + p.s_off;
            bf16_t* dst = (is_k ? p.k : p.q) + (((long)bi * p.H + h) * p.S_pad + s) * 64 + d;
            uint2 o = {pack_bf16(y[0] * rstd * w.x, y[1] * rstd * w.y), pack_bf16(y[2] * rstd * w.z, y[3] * rstd * w.w)};
            *(uint2*)dst = o;
        } else if constexpr (EPI == EPI_UNPATCH) {
            // token m -> (b, py, px); feature n..n+3 -> ((pp*patch + qq)*C + c)
            const int tok_per = p.hp * p.wp;
            const int bi = m / tok_per;
            const int t = m - bi * tok_per;
            const int ty = t / p.wp, tx = t - ty * p.wp;
            const int Himg = p.hp * p.patch, Wimg = p.wp * p.patch;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = n + r;
                const int c = f % p.out_ch;
                const int pq = f / p.out_ch;
                const int pp = pq / p.patch, qq = pq - pp * p.patch;
                p.out[(((long)bi * p.out_ch + c) * Himg + ty * p.patch + pp) * Wimg + tx * p.patch + qq] = f2bf(y[r]);
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(GemmParams p) {
    using C = Cfg<BM, BN, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware tile mapping: consecutive tiles -> same XCD (blockIdx.x % 8 is the XCD)
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int nblk = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    const int tm = bid / ntn, tn = bid - tm * ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging addresses: lane -> (row in 8-row group, physical 16B chunk)
    const int srow = lane >> 3, spc = lane & 7;
    const bf16_t* srcA[C::GA];
    const bf16_t* srcW[C::GW];
#pragma unroll
    for (int i = 0; i < C::GA; ++i) {
        const int row = (wave + i * C::NW) * 8 + srow;   // row inside the A tile
        const int c = spc ^ ((row >> 1) & 7);             // logical chunk stored at physical chunk spc
        int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;  // clamp: rows beyond M are never stored
        srcA[i] = p.A + (long)gm * p.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < C::GW; ++i) {
        const int row = (wave + i * C::NW) * 8 + srow;
        const int c = spc ^ ((row >> 1) & 7);
        int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
        srcW[i] = p.W + (long)gn * p.ldw + c * 8;
    }

    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * C::STAGE;
        const long ko = (long)kt * BK;
#pragma unroll
        for (int i = 0; i < C::GA; ++i) glds16(srcA[i] + ko, base + (wave + i * C::NW) * 1024);
#pragma unroll
        for (int i = 0; i < C::GW; ++i) glds16(srcW[i] + ko, base + C::A_BYTES + (wave + i * C::NW) * 1024);
    };

    // ---- fragment read addresses (bytes inside a stage)
    const int frow = lane & 15, fkg = lane >> 4;
    const int fsw = frow >> 1;  // (row>>1)&7 for rows whose tile offset is a multiple of 16
    int offX[2], offW[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int pc = (kk * 4 + fkg) ^ fsw;
        offX[kk] = (wm * C::TM + frow) * 128 + pc * 16;
        offW[kk] = C::A_BYTES + (wn * C::TN + frow) * 128 + pc * 16;
    }

    f32x4 acc[C::MI][C::NI];
#pragma unroll
    for (int i = 0; i < C::MI; ++i)
#pragma unroll
        for (int j = 0; j < C::NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = p.K / BK;
    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // tile t landed for every wave; everyone is done reading the other buffer
        if (t + 1 < nt) stage(t + 1, (t + 1) & 1);
        const char* sb = smem + (t & 1) * C::STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 xf[C::MI], wf[C::NI];
#pragma unroll
            for (int j = 0; j < C::NI; ++j) wf[j] = *(const bf16x8*)(sb + offW[kk] + j * 2048);
#pragma unroll
            for (int i = 0; i < C::MI; ++i) xf[i] = *(const bf16x8*)(sb + offX[kk] + i * 2048);
#pragma unroll
            for (int i = 0; i < C::MI; ++i)
#pragma unroll
                for (int j = 0; j < C::NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue.  Lane holds, for row m = .. + (lane&15): columns n = .. + ni*16 + 4*(lane>>4) + r
    const int mrow = m0 + wm * C::TM + frow;
    const int ncol = n0 + wn * C::TN + 4 * fkg;
#pragma unroll
    for (int i = 0; i < C::MI; ++i) {
        const int m = mrow + i * 16;
        float rstd = 1.0f;
        if constexpr (EPI == EPI_QK_NORM) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < C::NI; ++j) {
                const int n = ncol + j * 16;
                const int nc = n < p.N ? n : 0;
                const float4 bb = *(const float4*)(p.bias + nc);
                const float a0 = acc[i][j][0] + bb.x, a1 = acc[i][j][1] + bb.y, a2 = acc[i][j][2] + bb.z,
                            a3 = acc[i][j][3] + bb.w;
                ss += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
            }
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            rstd = rsqrtf(ss * (1.0f / 64.0f) + p.eps);
        }
#pragma unroll
        for (int j = 0; j < C::NI; ++j) {
            const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            epi_store<EPI>(p, m, ncol + j * 16, v, rstd);
        }
    }
}

template <int BM, int BN, int WM, int WN, int EPI>
hipError_t launch_cfg(const GemmParams& p, hipStream_t stream) {
    using C = Cfg<BM, BN, WM, WN>;
    auto kern = gemm_kernel<BM, BN, WM, WN, EPI>;
    constexpr int smem = 2 * C::STAGE;
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(ntm * ntn), dim3(C::NT), smem, stream, p);
    return hipGetLastError();
}

template <int EPI>
hipError_t launch_epi(const GemmParams& p, hipStream_t stream) {
    // 256x256 tiles (8 waves, 1 block/CU) once they fill the chip, else 128x128 (4 waves, 2 blocks/CU)
    const long big = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    if (big >= 200) return launch_cfg<256, 256, 2, 4, EPI>(p, stream);
    return launch_cfg<128, 128, 2, 2, EPI>(p, stream);
}

}  // namespace

hipError_t launch_gemm(const GemmParams& p, hipStream_t stream) {
    if (p.K % BK != 0 || p.M <= 0 || p.N <= 0) return hipErrorInvalidValue;
    switch (p.epi) {
        case EPI_BIAS: return launch_epi<EPI_BIAS>(p, stream);
        case EPI_BIAS_SILU: return launch_epi<EPI_BIAS_SILU>(p, stream);
        case EPI_BIAS_GELU: return launch_epi<EPI_BIAS_GELU>(p, stream);
        case EPI_POSADD: return launch_epi<EPI_POSADD>(p, stream);
        case EPI_ADDSRC_SILU: return launch_epi<EPI_ADDSRC_SILU>(p, stream);
        case EPI_GATE_RES: return launch_epi<EPI_GATE_RES>(p, stream);
        case EPI_QK_NORM: return launch_epi<EPI_QK_NORM>(p, stream);
        case EPI_VT: return launch_epi<EPI_VT>(p, stream);
        case EPI_UNPATCH: return launch_epi<EPI_UNPATCH>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mi355
