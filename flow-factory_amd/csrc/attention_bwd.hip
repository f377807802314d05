// mi355_flow -- flash-attention BACKWARD, head_dim 64, non-causal, for gfx950 (the gradient of the joint / dual
// attention of the MMDiT for the `optimize()` replay, reference src/flow_factory/trainers/grpo.py:263-330).
//
// Same building blocks as the forward (attention.hip): v_mfma_f32_32x32x16_bf16 with the row permutation that makes the first
// product's accumulator registers, packed to bf16, the B fragments of the second product (nothing round-trips through LDS), 64 x 64
// bf16 tiles staged by global_load_lds into XOR-swizzled 128-byte rows, one query (or key) per lane.  The stored q carries the softmax
// scale log2(e)/8, so with s = q~.k (log2 domain) and L = log2 sum_j 2^s_j from the forward:
//       P = 2^(s - L)            dP = dO . V^T            dZ = P o (dP - Delta),  Delta_i = sum_d dO_id O_id
//       dV = P^T dO              dK = ln2 * dZ^T q~       dq~ = ln2 * dZ k
// Two deterministic passes instead of one pass with fp32 atomics on dQ (7 instead of 5 tile products, but bit-reproducible and no
// fp32 dQ buffer):
//   attn_bwd_dkv_kernel : one KEY per lane; loops over query tiles; S = Q K^T and dP = dO V^T arrive with lane = key and 8 consecutive
//                         queries per register group, so P and dZ feed  dV^T += dO^T . P  and  dK^T += Q^T . dZ  straight from registers;
//   attn_bwd_dq_kernel  : one QUERY per lane (the forward's orientation); S^T = K Q^T, dP^T = V dO^T, then  dQ^T += K^T . dZ^T.
// Transposed operand copies (qT, kT, doT: [64][S_pad]) and v in [S_pad][64] are produced by HBM-bound kernels in backward.hip.
#include "kernels.h"

namespace mi355 {
namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int TB = 64;                     // rows of the streamed dimension per tile
constexpr int TILE = TB * 64 * 2;          // 8 KiB
// NWAVE waves x 32 lanes-of-interest keys (pass 1) / queries (pass 2) per workgroup.  4 waves, two workgroups per CU: 8-wave workgroups
// (one per CU, half the L2 -> LDS operand traffic: every workgroup streams ALL tiles of its (b, h)) measured 4-8 % SLOWER
// (profiles/r03u_attn_bwd_waves_ab.txt) -- two independent barriers per CU overlap better than one, and the traffic is not the bound.
constexpr int NWAVES = 4;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ int row_perm(int i) {     // MFMA output row i (0..31) -> row offset inside the 32-row block (attention.hip key_perm)
    const int a = i >> 3, g = (i >> 2) & 1, b = i & 3;
    return 16 * (a >> 1) + 8 * g + 4 * (a & 1) + b;
}

// stage one 64-row x 128-byte tile (rows `row_stride` elements apart in HBM) into LDS at `dst`; wave w copies 8-row groups w, w + NWAVE
template <int NWAVE>
__device__ __forceinline__ void stage_tile(const bf16_t* src, long row_stride, char* dst, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 8 / NWAVE; ++i) {
        const int grp = wave + i * NWAVE;
        const int row = grp * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        __builtin_amdgcn_global_load_lds((gptr_t)(src + (long)row * row_stride + c * 8), (lptr_t)(dst + grp * 1024), 16, 0, 0);
    }
}

// accumulators acc[db][r] = X^T[d = 32 db + 8 (r>>2) + 4 lg + (r&3)][column = lane & 31] -> rows dst[(row0 + column)][0..64) bf16, scaled
__device__ __forceinline__ void store_rows(const f32x16 (&acc)[2], float scale, char* ob, bf16_t* dst, int row0, int row_limit, int lane) {
    const int lq = lane & 31, lg = lane >> 5;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int d0 = 32 * db + 8 * a + 4 * lg;
            uint2 w = {pack_bf16(acc[db][4 * a] * scale, acc[db][4 * a + 1] * scale),
                       pack_bf16(acc[db][4 * a + 2] * scale, acc[db][4 * a + 3] * scale)};
            const int chunk = (d0 >> 3) ^ (lq & 7);
            *(uint2*)(ob + lq * 128 + chunk * 16 + (d0 & 7) * 2) = w;
        }
    __builtin_amdgcn_wave_barrier();      // wave-private region, in-order LDS
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + (lane >> 3), c = lane & 7;
        const uint4 val = *(const uint4*)(ob + r * 128 + ((c ^ (r & 7)) << 4));
        if (row0 + r < row_limit) *(uint4*)(dst + (long)(row0 + r) * 64 + c * 8) = val;
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bf16x8 frag4(unsigned a, unsigned b, unsigned c, unsigned d) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const u32x4 u = {a, b, c, d};
    return __builtin_bit_cast(bf16x8, u);
}

// ----------------------------------------------------------------------------------------------- pass 1: dK, dV
// LDS stage: Q [64 q][64 d] | dO [64 q][64 d] | Q^T [64 d][64 q] | dO^T [64 d][64 q] | L[64], Delta[64]
constexpr int ST1 = 4 * TILE + 512;
template <int NWAVE>
__global__ __launch_bounds__(NWAVE * 64, 2) void attn_bwd_dkv_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane & 31, lg = lane >> 5;
    constexpr int KB = 32 * NWAVE;
    const int nkb = (p.S + KB - 1) / KB;
    const int nwg = nkb * p.H * p.B;
    int wid = blockIdx.x;
    {   // XCD-aware order: the key blocks of one (b, h) share its Q / dO stream in one private L2
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = wid & 7;
        wid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (wid >> 3);
    }
    const int kblk = wid % nkb;
    const long bh = wid / nkb;
    const bf16_t* Qg = p.q + bh * p.S_pad * 64;
    const bf16_t* Og = p.doh + bh * p.S_pad * 64;
    const bf16_t* QTg = p.qT + bh * 64 * p.S_pad;
    const bf16_t* OTg = p.doT + bh * 64 * p.S_pad;
    const float* Lg = p.lse + bh * p.S_pad;
    const float* Dg = p.delta + bh * p.S_pad;

    // own key's K / V fragments (B operands): lane holds row[key][kk*16 + lg*8 .. +8]
    const int key = kblk * KB + wave * 32 + lk;
    const int key_ld = key < p.S ? key : p.S - 1;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        kf[kk] = *(const bf16x8*)(p.k + (bh * p.S_pad + key_ld) * 64 + kk * 16 + lg * 8);
        vf[kk] = *(const bf16x8*)(p.v + (bh * p.S_pad + key_ld) * 64 + kk * 16 + lg * 8);
    }

    auto stage = [&](int t, int buf) {
        char* base = smem + buf * ST1;
        stage_tile<NWAVE>(Qg + (long)t * TB * 64, 64, base, wave, lane);
        stage_tile<NWAVE>(Og + (long)t * TB * 64, 64, base + TILE, wave, lane);
        stage_tile<NWAVE>(QTg + (long)t * TB, p.S_pad, base + 2 * TILE, wave, lane);
        stage_tile<NWAVE>(OTg + (long)t * TB, p.S_pad, base + 3 * TILE, wave, lane);
    };
    // L / Delta of a query tile (2 x 64 floats): loaded into a register early, written to LDS after the tile's MFMA work
    const float* ldsrc = tid < 64 ? Lg + tid : Dg + (tid - 64);

    // row-major tiles as A operand: row = 32*qb + perm(lk), chunk = 2*kk + lg
    const int prow = row_perm(lk);
    int offR[4], offT[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) offR[kk] = prow * 128 + (((2 * kk + lg) ^ ((prow >> 1) & 7)) << 4);
    // transposed tiles as A operand: row d = 32*db + lk, chunk (queries) = 2*c + lg
#pragma unroll
    for (int c = 0; c < 4; ++c) offT[c] = lk * 128 + (((2 * c + lg) ^ ((lk >> 1) & 7)) << 4);

    f32x16 dk[2], dv[2];
    dk[0] = (f32x16){0}; dk[1] = (f32x16){0}; dv[0] = (f32x16){0}; dv[1] = (f32x16){0};
    const int nt = (p.S + TB - 1) / TB;
    stage(0, 0);
    if (tid < 128) ((float*)(smem + 4 * TILE))[tid] = -ldsrc[0];      // LDS holds -L and -Delta: they enter the MFMA chains as their C operand
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float ld_next = 0.f;
        if (t + 1 < nt) {
            stage(t + 1, (t + 1) & 1);
            if (tid < 128) ld_next = -ldsrc[(t + 1) * TB];
        }
        const char* sb = smem + (t & 1) * ST1;
        const float* sL = (const float*)(sb + 4 * TILE);
        const float* sD = sL + 64;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            // register r <-> query t*64 + 32*qb + 16*(r>>3) + 8*lg + (r&7): the accumulators start at -L[query] / -Delta[query] (8 consecutive
            // floats per half: four 16-byte LDS reads each), so the chains deliver s - L and dP - Delta and no subtraction is issued
            f32x16 s, dp;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const float4 a0 = *(const float4*)(sL + 32 * qb + 16 * hh + 8 * lg), a1 = *(const float4*)(sL + 32 * qb + 16 * hh + 8 * lg + 4);
                const float4 b0 = *(const float4*)(sD + 32 * qb + 16 * hh + 8 * lg), b1 = *(const float4*)(sD + 32 * qb + 16 * hh + 8 * lg + 4);
                s[8 * hh + 0] = a0.x; s[8 * hh + 1] = a0.y; s[8 * hh + 2] = a0.z; s[8 * hh + 3] = a0.w;
                s[8 * hh + 4] = a1.x; s[8 * hh + 5] = a1.y; s[8 * hh + 6] = a1.z; s[8 * hh + 7] = a1.w;
                dp[8 * hh + 0] = b0.x; dp[8 * hh + 1] = b0.y; dp[8 * hh + 2] = b0.z; dp[8 * hh + 3] = b0.w;
                dp[8 * hh + 4] = b1.x; dp[8 * hh + 5] = b1.y; dp[8 * hh + 6] = b1.z; dp[8 * hh + 7] = b1.w;
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 qa = *(const bf16x8*)(sb + offR[kk] + qb * 4096);
                const bf16x8 oa = *(const bf16x8*)(sb + TILE + offR[kk] + qb * 4096);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vf[kk], dp, 0, 0, 0);
            }
            unsigned pk[8], zk[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float p0 = __builtin_amdgcn_exp2f(s[r]), p1 = __builtin_amdgcn_exp2f(s[r + 1]);
                if (t == nt - 1) {
                    const int ql = 32 * qb + 16 * (r >> 3) + 8 * lg + (r & 7);
                    if (t * TB + ql >= p.S) p0 = 0.f;
                    if (t * TB + ql + 1 >= p.S) p1 = 0.f;
                }
                pk[r >> 1] = pack_bf16(p0, p1);
                zk[r >> 1] = pack_bf16(p0 * dp[r], p1 * dp[r + 1]);
            }
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {          // two k-steps of 16 queries: c = 2*qb + hs
                const int c = 2 * qb + hs;
                const bf16x8 pf = frag4(pk[4 * hs], pk[4 * hs + 1], pk[4 * hs + 2], pk[4 * hs + 3]);
                const bf16x8 zf = frag4(zk[4 * hs], zk[4 * hs + 1], zk[4 * hs + 2], zk[4 * hs + 3]);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 ot = *(const bf16x8*)(sb + 3 * TILE + offT[c] + db * 4096);
                    const bf16x8 qt = *(const bf16x8*)(sb + 2 * TILE + offT[c] + db * 4096);
                    dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ot, pf, dv[db], 0, 0, 0);
                    dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt, zf, dk[db], 0, 0, 0);
                }
            }
        }
        // the other buffer's L / Delta slot was last read in iteration t-1 (all waves are past this iteration's barrier)
        if (t + 1 < nt && tid < 128) ((float*)(smem + ((t + 1) & 1) * ST1 + 4 * TILE))[tid] = ld_next;
    }
    __syncthreads();     // every wave is done with the ring: reuse it for the output transposes (4 KiB per wave)
    char* ob = smem + wave * 4096;
    const int row0 = kblk * KB + wave * 32;
    store_rows(dk, LN2, ob, p.dk + bh * p.S_pad * 64, row0, p.S, lane);
    store_rows(dv, 1.0f, ob, p.dv + bh * p.S_pad * 64, row0, p.S, lane);
}

// ----------------------------------------------------------------------------------------------- pass 2: dQ
// LDS stage: K [64 keys][64 d] | V [64 keys][64 d] | K^T [64 d][64 keys]
constexpr int ST2 = 3 * TILE;
template <int NWAVE>
__global__ __launch_bounds__(NWAVE * 64, 2) void attn_bwd_dq_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, lg = lane >> 5;
    constexpr int QB = 32 * NWAVE;
    const int nqb = (p.S + QB - 1) / QB;
    const int nwg = nqb * p.H * p.B;
    int wid = blockIdx.x;
    {
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = wid & 7;
        wid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (wid >> 3);
    }
    const int qblk = wid % nqb;
    const long bh = wid / nqb;
    const bf16_t* Kg = p.k + bh * p.S_pad * 64;
    const bf16_t* Vg = p.v + bh * p.S_pad * 64;
    const bf16_t* KTg = p.kT + bh * 64 * p.S_pad;

    const int q_row = qblk * QB + wave * 32 + lq;
    const int q_ld = q_row < p.S ? q_row : p.S - 1;
    bf16x8 qf[4], of[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        qf[kk] = *(const bf16x8*)(p.q + (bh * p.S_pad + q_ld) * 64 + kk * 16 + lg * 8);
        of[kk] = *(const bf16x8*)(p.doh + (bh * p.S_pad + q_ld) * 64 + kk * 16 + lg * 8);
    }
    const float L = p.lse[bh * p.S_pad + q_ld], Dl = p.delta[bh * p.S_pad + q_ld];

    auto stage = [&](int t, int buf) {
        char* base = smem + buf * ST2;
        stage_tile<NWAVE>(Kg + (long)t * TB * 64, 64, base, wave, lane);
        stage_tile<NWAVE>(Vg + (long)t * TB * 64, 64, base + TILE, wave, lane);
        stage_tile<NWAVE>(KTg + (long)t * TB, p.S_pad, base + 2 * TILE, wave, lane);
    };
    const int prow = row_perm(lq);
    int offR[4], offT[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) offR[kk] = prow * 128 + (((2 * kk + lg) ^ ((prow >> 1) & 7)) << 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) offT[c] = lq * 128 + (((2 * c + lg) ^ ((lq >> 1) & 7)) << 4);

    f32x16 dq[2];
    dq[0] = (f32x16){0}; dq[1] = (f32x16){0};
    const int nt = (p.S + TB - 1) / TB;
    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) stage(t + 1, (t + 1) & 1);
        const char* sb = smem + (t & 1) * ST2;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s = (f32x16){0}, dp = (f32x16){0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 ka = *(const bf16x8*)(sb + offR[kk] + kb * 4096);
                const bf16x8 va = *(const bf16x8*)(sb + TILE + offR[kk] + kb * 4096);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, of[kk], dp, 0, 0, 0);
            }
            // register r <-> key t*64 + 32*kb + 16*(r>>3) + 8*lg + (r&7)
            unsigned zk[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float p0 = __builtin_amdgcn_exp2f(s[r] - L), p1 = __builtin_amdgcn_exp2f(s[r + 1] - L);
                if (t == nt - 1) {
                    const int kl = t * TB + 32 * kb + 16 * (r >> 3) + 8 * lg + (r & 7);
                    if (kl >= p.S) p0 = 0.f;
                    if (kl + 1 >= p.S) p1 = 0.f;
                }
                zk[r >> 1] = pack_bf16(p0 * (dp[r] - Dl), p1 * (dp[r + 1] - Dl));
            }
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                const int c = 2 * kb + hs;
                const bf16x8 zf = frag4(zk[4 * hs], zk[4 * hs + 1], zk[4 * hs + 2], zk[4 * hs + 3]);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 kt = *(const bf16x8*)(sb + 2 * TILE + offT[c] + db * 4096);
                    dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt, zf, dq[db], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();
    char* ob = smem + wave * 4096;
    store_rows(dq, LN2, ob, p.dq + bh * p.S_pad * 64, qblk * QB + wave * 32, p.S, lane);
}

}  // namespace

hipError_t launch_attention_bwd(const AttnBwdParams& p, hipStream_t stream) {
    if (p.S <= 0 || p.S_pad % TB != 0 || p.S_pad < p.S) return hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<NWAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ST1);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nb = (p.S + 32 * NWAVES - 1) / (32 * NWAVES);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<NWAVES>, dim3(nb * p.H * p.B), dim3(NWAVES * 64), 2 * ST1, stream, p);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<NWAVES>, dim3(nb * p.H * p.B), dim3(NWAVES * 64), 2 * ST2, stream, p);
    return hipGetLastError();
}

}  // namespace mi355
