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

constexpr bool is_qk_epi(int e) { return e == EPI_QK_NORM || e == EPI_QK_NORM_RSTD; }


constexpr int BK = 64;
int g_raster_gm = 6;          // tile rows per raster band (0 = plain row-major order); mi355_tune_set(7, v)
int g_pp_min_tiles = 128;     // smallest 256x256-tile grid for gemm_pp_kernel
int g_w4_min_tiles = 512;     // smallest 256x256-tile grid the DEFAULT dispatch (key 0 = 1) gives to the 4-wave kernel; mi355_tune_set(31, v)
int g_mid_mode = 1;           // mid-size kernel (128x192 / 192x128 tiles): 0 off, 1 by the cost rule of launch_epi, 2 wherever it applies; mi355_tune_set(32, v)
double g_mid_alpha = 1.0;     // margin of that rule: the mid-size kernel's estimated cost is multiplied by it; mi355_tune_set(33, percent)
int g_mid_max_tiles = 256;    // largest grid of its tiles: ONE round, one workgroup per CU (measured in-model, profiles/r06b / r06d: two-round grids lose
                              // to the ping-pong kernel's 192 tiles + the CUs it leaves to the text chain: 1024^2 B = 2 -7 %); mi355_tune_set(37, v)
int g_mid_plan_hint = 1;      // set by the SD3.5 engine around a forward (set_mid_plan_hint): 0 = the plan's text chain is too small for the kernel to pay
int g_w6_mode = 0;            // 256x192 kernel: 0 off (default), 1 by the cost rule of launch_epi, 2 wherever it applies; mi355_tune_set(40, v).
                              // Measured (profiles/r06n_*): back to back it is 16-20 % faster than the 0.75- / 1.5-round 256x256 grids it replaces
                              // (8192 x 1536 x 6144: 132 -> 106 us = 1.46 PFLOP/s), bit-identical -- and the SD3.5 forward gets 2.4 % SLOWER with
                              // it (B = 2, 1024^2: 24.05 -> 24.63 ms; optimize() 83.1 -> 83.6 ms): the quarter of the CUs that a 192-tile grid
                              // leaves free is where the text stream's GEMMs run (two-stream forward), and a 144 KiB workgroup on every CU
                              // shuts them out.  Kept for single-stream callers and as a tested tile shape; off in the shipped dispatch.
double g_w6_alpha = 1.05;     // its margin; mi355_tune_set(41, percent)
int g_w6_min_tiles = 200;     // no launch below this many of its tiles; mi355_tune_set(42, v)
int g_mid_min_tiles = 160;    // no mid-size launch below this many of its tiles (sub-chip grids: a lone 128x128 tile per CU is quicker); mi355_tune_set(34, v)

// linear tile id -> (tm, tn).  Bands of `gm` tile rows are walked column by column, so the ~32 consecutive ids that the workgroups of one
// XCD hold at any time form a near-square block: gm A-panels + ~32/gm W-panels stream through that XCD's L2 per round instead of ~1 + 32
// (row-major order on a wide N).  The mapping does not touch the arithmetic of a tile: results are bit-identical for every gm.
__device__ __forceinline__ void tile_coords(int tile, int ntm, int ntn, int gm, int& tm, int& tn) {
    if (gm <= 0) { tm = tile / ntn; tn = tile - tm * ntn; return; }
    const int band = tile / (gm * ntn);
    const int first = band * gm;
    const int rows = ntm - first < gm ? ntm - first : gm;
    const int r = tile - band * gm * ntn;
    tn = r / rows;
    tm = first + (r - tn * rows);
}
int g_gemm_variant = 1;  // large grids: 0 simple 2-stage kernel; 1 (default) the 4-wave kernel with the hand-scheduled loop where it applies and
                         // K <= g_w4_max_k, else the persistent ping-pong kernel; 2 the 4-wave kernel wherever it applies; 3 ping-pong only
int g_w4_max_k = 3072;   // microbenchmark (profiles/r03e_gemm_w4_ab.txt): w4 +5 ... +10 % at K = 1536, -8 % at K = 6144 (16 loads per wave and
                         // K-tile concentrated in half an interval: too little flight time for rows 12 KiB apart streaming from HBM)
// (a 32x32x16-MFMA / 2-phases-per-K-tile ping-pong variant was measured 6-10 % SLOWER than the 16x16x32 / 4-phase one
//  on every shape of this model and was dropped: profiles/r01_gemm_variants.txt)

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

// ---- unpatchify epilogue (proj_out, N = p*p*C = 64): scalar scatter straight from the accumulator layout, called
// once per (token row m, 4 consecutive feature columns n..n+3); feature f = (pp*patch + qq)*C + c
__device__ __forceinline__ void unpatch_store(const GemmParams& p, int m, int n, const float (&v)[4]) {
    if (m >= p.M || n >= p.N) return;
    const float4 bb = *(const float4*)(p.bias + n);
    const float y[4] = {v[0] + bb.x, v[1] + bb.y, v[2] + bb.z, v[3] + bb.w};
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

// ---- LDS-staged epilogue ----------------------------------------------------------------------
// The MFMA accumulator layout gives a lane 4 consecutive columns of 16 different rows; written to
// HBM directly that is 32-byte partial-line stores (4 per 128-byte line, store-issue bound: the
// store tail of a 256x256 tile cost ~19 us of a 52 us K=1536 tile).  Instead each wave transposes its
// 64x64 half-tile through a private 8 KiB LDS region (bf16, 16-byte chunks XOR-swizzled by row) and
// writes / read-modify-writes whole 128-byte lines, 16 bytes per lane.
//   phase 1 (accumulator layout, fp32): + bias, per-head RMSNorm (q/k), SiLU / GELU  -> bf16 -> LDS
//   phase 2 (row layout, 8 columns per lane): + pos-embed / + src, gated residual, scatter -> HBM
template <int EPI, bool FULL>
__device__ __forceinline__ void store_row8(const GemmParams& p, int m, int n, int n_wave, uint4 val) {
    const bool full = FULL || (n + 8 <= p.N);
    float y[8] = {bf_lo(val.x), bf_hi(val.x), bf_lo(val.y), bf_hi(val.y), bf_lo(val.z), bf_hi(val.z), bf_lo(val.w), bf_hi(val.w)};
    if constexpr (is_qk_epi(EPI)) {
        const int D = p.H * 64;
        const bool is_k = n_wave >= D;
        const int h = ((is_k ? n_wave - D : n_wave) >> 6);
        const int bi = m / p.rows_per_sample;
        const int s = m - bi * p.rows_per_sample + p.s_off;
        bf16_t* dst = (is_k ? p.k : p.q) + (((long)bi * p.H + h) * p.S_pad + s) * 64 + (n - n_wave);
        *(uint4*)dst = val;
        return;
    } else if constexpr (EPI == EPI_VT) {
        // m = feature, n.. = 8 tokens; head_dim 64 (hd_shift 0) or 1 << hd_shift
        const int hs = p.hd_shift ? p.hd_shift : 6;
        const int h = m >> hs, d = m & ((1 << hs) - 1);
        if (full && ((p.rows_per_sample | p.s_off) & 7) == 0) {
            const int bi = n / p.rows_per_sample;
            const int s0 = n - bi * p.rows_per_sample + p.s_off;
            *(uint4*)(p.q + ((((long)bi * p.H + h) << hs) + d) * p.S_pad + s0) = val;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int tok = n + e;
                if (tok < p.N) {
                    const int bi = tok / p.rows_per_sample;
                    const int s = tok - bi * p.rows_per_sample + p.s_off;
                    p.q[((((long)bi * p.H + h) << hs) + d) * p.S_pad + s] = f2bf(y[e]);
                }
            }
        }
        return;
    } else {
        bf16_t* op = p.out + (long)m * p.ldo + n;
        const bool vec = FULL || (full && ((p.ldo & 7) == 0));
        if constexpr (EPI == EPI_DGELU) {
            const bf16_t* ap = p.aux + (long)m * p.ld_aux + n;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (n + e < p.N) y[e] *= gelu_tanh_grad_f(bf2f(ap[e]));
        } else if constexpr (EPI == EPI_POSADD || EPI == EPI_ADDSRC_SILU) {
            const bf16_t* ap = p.aux + (long)(m % p.rows_per_sample) * p.ld_aux + n;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (n + e < p.N) {
                    y[e] += bf2f(ap[e]);
                    if constexpr (EPI == EPI_ADDSRC_SILU) y[e] = silu_f(round_bf16(y[e]));
                }
        } else if constexpr (EPI == EPI_GATE_RES) {
            const int bi = m / p.rows_per_sample;
            const bf16_t* gp = p.aux + (long)bi * p.ld_aux + n;
            if (vec) {
                const uint4 g = *(const uint4*)gp;
                const uint4 x = *(const uint4*)op;
                // explicit fma: every path that forms x + g * y (this one, the ragged one, the batched one of epilogue_part) must round alike
                y[0] = __builtin_fmaf(bf_lo(g.x), y[0], bf_lo(x.x)); y[1] = __builtin_fmaf(bf_hi(g.x), y[1], bf_hi(x.x));
                y[2] = __builtin_fmaf(bf_lo(g.y), y[2], bf_lo(x.y)); y[3] = __builtin_fmaf(bf_hi(g.y), y[3], bf_hi(x.y));
                y[4] = __builtin_fmaf(bf_lo(g.z), y[4], bf_lo(x.z)); y[5] = __builtin_fmaf(bf_hi(g.z), y[5], bf_hi(x.z));
                y[6] = __builtin_fmaf(bf_lo(g.w), y[6], bf_lo(x.w)); y[7] = __builtin_fmaf(bf_hi(g.w), y[7], bf_hi(x.w));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (n + e < p.N) y[e] = __builtin_fmaf(bf2f(gp[e]), y[e], bf2f(op[e]));
            }
        }
        if (vec) {
            uint4 o;
            if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_SILU || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_ROW) o = val;
            else o = make_uint4(pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]), pack_bf16(y[4], y[5]), pack_bf16(y[6], y[7]));
            *(uint4*)op = o;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (n + e < p.N) op[e] = f2bf(y[e]);
        }
    }
}

// acc: NR*16 rows x 64 columns of the wave's tile, acc[i][j][e] = C[m_base + 16 i + (lane&15)][n_base + 16 j + 4 (lane>>4) + e];
// stg: NR*2 KiB of wave-private LDS
// FULL: the whole 256-wide tile lies inside [0,M) x [0,N) and rows are 16-byte aligned (no guards)
// bpre: the 4 bias float4 of this lane's columns (n_base + 16 j + 4 (lane >> 4)), loaded ONCE per output tile by the caller -- the compiler
//       cannot hoist these loads out of the per-chunk calls itself (the chunks' global stores may alias p.bias for all it knows), and each
//       call then pays one L2 round trip before its first arithmetic
struct BiasPre { float4 v[4]; };      // by value: a pointer to a caller's array keeps that array in scratch memory
// NJ (round 6): 16-column blocks of the chunk (4 = the 64-column chunk every kernel used so far; 2 = the 32-column tail of the mid-size
//       kernel's 96-column wave tile).  The staging rows stay 128 bytes; with NJ < 4 only the first 2 NJ 16-byte slots of a row carry data and
//       the row-layout phase masks the lanes of the others (they would touch the NEIGHBOURING wave's columns).
template <int EPI, int NR, bool FULL = false, bool PRE = false, int NJ = 4>
__device__ __forceinline__ void epilogue_part(const GemmParams& p, const f32x4 (&acc)[NR][NJ], int m_base, int n_base,
                                              char* stg, int lane, BiasPre bpre = BiasPre()) {
    static_assert(NJ == 4 || (NJ == 2 && !is_qk_epi(EPI)), "a q/k head is one whole 64-column chunk");
    const int frow = lane & 15, fkg = lane >> 4;
    const bool cok = NJ == 4 || (lane & 7) < 2 * NJ;          // row-layout phase: this lane's 8 columns belong to the chunk
    float4 bcol[NJ];
    float4 nw[NJ];
    if constexpr (EPI == EPI_BIAS_GELU || EPI == EPI_GATE_RES) {
        // training-mode forward: the pre-activation (GELU) / the un-gated projection (gated residual) goes to the stash first (same wave-private staging, LDS ops of a wave are in order)
        if (p.stash) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int r = i * 16 + frow;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int n = n_base + j * 16 + 4 * fkg;
                    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (FULL || n < p.N) bb = *(const float4*)(p.bias + n);
                    const int chunk = (j * 2 + (fkg >> 1)) ^ (r & 7);
                    uint2 o = {pack_bf16(acc[i][j][0] + bb.x, acc[i][j][1] + bb.y), pack_bf16(acc[i][j][2] + bb.z, acc[i][j][3] + bb.w)};
                    *(uint2*)(stg + r * 128 + chunk * 16 + (fkg & 1) * 8) = o;
                }
            }
#pragma unroll
            for (int it = 0; it < NR * 2; ++it) {
                const int r = it * 8 + (lane >> 3), c = lane & 7;
                const uint4 val = *(const uint4*)(stg + r * 128 + ((c ^ (r & 7)) << 4));
                const int m = m_base + r, n = n_base + c * 8;
                if (cok && (FULL || m < p.M)) {
                    bf16_t* sp = p.stash + (long)m * p.ld_stash + n;
                    if (FULL || (n + 8 <= p.N && (p.ld_stash & 7) == 0)) *(uint4*)sp = val;
                    else {
                        const unsigned w4[4] = {val.x, val.y, val.z, val.w};
                        for (int e = 0; e < 8; ++e)
                            if (n + e < p.N) sp[e] = (bf16_t)((w4[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n_base + j * 16 + 4 * fkg;
        bcol[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (EPI != EPI_VT && EPI != EPI_BIAS_ROW) {
            if constexpr (PRE) bcol[j] = bpre.v[j];
            else if (FULL || n < p.N) bcol[j] = *(const float4*)(p.bias + n);
        }
        if constexpr (is_qk_epi(EPI)) {
            const bool is_k = n_base >= p.H * 64;
            nw[j] = *(const float4*)((is_k ? p.nw_k : p.nw_q) + j * 16 + 4 * fkg);
            if (!is_k && p.q_scale != 0.f) { nw[j].x *= p.q_scale; nw[j].y *= p.q_scale; nw[j].z *= p.q_scale; nw[j].w *= p.q_scale; }
        }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = i * 16 + frow;
        float y[NJ][4];
        float brow = 0.f;
        constexpr bool ROWB = (EPI == EPI_VT || EPI == EPI_BIAS_ROW);
        if constexpr (ROWB) {
            const int m = m_base + r;
            brow = p.bias[m < p.M ? m : p.M - 1];
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            y[j][0] = acc[i][j][0] + (ROWB ? brow : bcol[j].x);
            y[j][1] = acc[i][j][1] + (ROWB ? brow : bcol[j].y);
            y[j][2] = acc[i][j][2] + (ROWB ? brow : bcol[j].z);
            y[j][3] = acc[i][j][3] + (ROWB ? brow : bcol[j].w);
        }
        if constexpr (is_qk_epi(EPI)) {
            // The rollout and the training-mode forward run DIFFERENT instantiations of this epilogue (EPI_QK_NORM / EPI_QK_NORM_RSTD) and
            // must agree bit for bit (ratio == 1): every multiply-add below is an explicit fma and contraction is off, so that hipcc's
            // SLP packing cannot fuse in one instantiation what it leaves unfused in the other (measured: v_pk_fma vs v_pk_mul + v_pk_add).
            float rstd;
            {
#pragma clang fp contract(off)
                float sj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    sj[j] = __builtin_fmaf(y[j][3], y[j][3], __builtin_fmaf(y[j][2], y[j][2], __builtin_fmaf(y[j][1], y[j][1], y[j][0] * y[j][0])));
                float ss = (sj[0] + sj[1]) + (sj[2] + sj[3]);
                ss += __shfl_xor(ss, 16, 64);
                ss += __shfl_xor(ss, 32, 64);
                rstd = rsqrtf(__builtin_fmaf(ss, 1.0f / 64.0f, p.eps));
            }
            if constexpr (EPI == EPI_QK_NORM_RSTD) {      // training-mode forward: 1/rms of this (row, head) for the RMSNorm backward
                // (its own epilogue kind: as a runtime branch it cost the rollout kernel 13 spilled VGPRs)
                const int m = m_base + r;
                if (fkg == 0 && (FULL || m < p.M)) p.rstd_out[(long)m * (2 * p.H) + (n_base >> 6)] = rstd;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                y[j][0] *= rstd * nw[j].x; y[j][1] *= rstd * nw[j].y; y[j][2] *= rstd * nw[j].z; y[j][3] *= rstd * nw[j].w;
            }
        }
        if constexpr (EPI == EPI_BIAS_SILU) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) y[j][e] = silu_f(round_bf16(y[j][e]));
        }
        if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) y[j][e] = gelu_tanh_f(y[j][e]);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int chunk = (j * 2 + (fkg >> 1)) ^ (r & 7);
            uint2 o = {pack_bf16(y[j][0], y[j][1]), pack_bf16(y[j][2], y[j][3])};
            *(uint2*)(stg + r * 128 + chunk * 16 + (fkg & 1) * 8) = o;
        }
    }
    // wave-private region: a wave's LDS operations execute in order, no barrier needed
    if constexpr (FULL && EPI == EPI_GATE_RES) {
        // x += gate * y in place.  All LDS reads, then all global loads (the residual rows and the gate vectors), then arithmetic + stores:
        // written as one loop the compiler keeps load -> store -> load order (a row's store may alias the next row's load for all it
        // knows), i.e. NR * 2 exposed L2 round trips per call with nothing else to run on a SIMD whose waves are all in their epilogue.
        uint4 val[NR * 2], xr[NR * 2], gr[NR * 2];
        const int c = lane & 7;
#pragma unroll
        for (int it = 0; it < NR * 2; ++it) {
            const int r = it * 8 + (lane >> 3);
            val[it] = *(const uint4*)(stg + r * 128 + ((c ^ (r & 7)) << 4));
        }
#pragma unroll
        for (int it = 0; it < NR * 2; ++it) {
            const int m = m_base + it * 8 + (lane >> 3), n = n_base + c * 8;
            if (cok) {
                xr[it] = *(const uint4*)(p.out + (long)m * p.ldo + n);
                gr[it] = *(const uint4*)(p.aux + (long)(m / p.rows_per_sample) * p.ld_aux + n);
            }
        }
#pragma unroll
        for (int it = 0; it < NR * 2; ++it) {
            const int m = m_base + it * 8 + (lane >> 3), n = n_base + c * 8;
            if (!cok) continue;
            const uint4 v = val[it], x = xr[it], g = gr[it];
            // (same operations, in the same order, as store_row8's vector path)
            const float y0 = __builtin_fmaf(bf_lo(g.x), bf_lo(v.x), bf_lo(x.x)), y1 = __builtin_fmaf(bf_hi(g.x), bf_hi(v.x), bf_hi(x.x));
            const float y2 = __builtin_fmaf(bf_lo(g.y), bf_lo(v.y), bf_lo(x.y)), y3 = __builtin_fmaf(bf_hi(g.y), bf_hi(v.y), bf_hi(x.y));
            const float y4 = __builtin_fmaf(bf_lo(g.z), bf_lo(v.z), bf_lo(x.z)), y5 = __builtin_fmaf(bf_hi(g.z), bf_hi(v.z), bf_hi(x.z));
            const float y6 = __builtin_fmaf(bf_lo(g.w), bf_lo(v.w), bf_lo(x.w)), y7 = __builtin_fmaf(bf_hi(g.w), bf_hi(v.w), bf_hi(x.w));
            *(uint4*)(p.out + (long)m * p.ldo + n) = make_uint4(pack_bf16(y0, y1), pack_bf16(y2, y3), pack_bf16(y4, y5), pack_bf16(y6, y7));
        }
    } else if constexpr (FULL && EPI == EPI_DGELU) {
        // dpre = dy * gelu'(pre): all LDS reads, then all loads of the stashed pre-activation, then arithmetic + stores (as above: inside
        // store_row8 every row's load waits behind the previous row's store)
        uint4 val[NR * 2], ar[NR * 2];
        const int c = lane & 7;
#pragma unroll
        for (int it = 0; it < NR * 2; ++it) {
            const int r = it * 8 + (lane >> 3);
            val[it] = *(const uint4*)(stg + r * 128 + ((c ^ (r & 7)) << 4));
        }
#pragma unroll
        for (int it = 0; it < NR * 2; ++it)
            if (cok) ar[it] = *(const uint4*)(p.aux + (long)(m_base + it * 8 + (lane >> 3)) * p.ld_aux + n_base + c * 8);
#pragma unroll
        for (int it = 0; it < NR * 2; ++it) {
            if (!cok) continue;
            const uint4 v = val[it], a = ar[it];
            // (same operations as store_row8's path: y * gelu'(pre) on the bf16-rounded product sum)
            const float y0 = bf_lo(v.x) * gelu_tanh_grad_f(bf_lo(a.x)), y1 = bf_hi(v.x) * gelu_tanh_grad_f(bf_hi(a.x));
            const float y2 = bf_lo(v.y) * gelu_tanh_grad_f(bf_lo(a.y)), y3 = bf_hi(v.y) * gelu_tanh_grad_f(bf_hi(a.y));
            const float y4 = bf_lo(v.z) * gelu_tanh_grad_f(bf_lo(a.z)), y5 = bf_hi(v.z) * gelu_tanh_grad_f(bf_hi(a.z));
            const float y6 = bf_lo(v.w) * gelu_tanh_grad_f(bf_lo(a.w)), y7 = bf_hi(v.w) * gelu_tanh_grad_f(bf_hi(a.w));
            *(uint4*)(p.out + (long)(m_base + it * 8 + (lane >> 3)) * p.ldo + n_base + c * 8) =
                make_uint4(pack_bf16(y0, y1), pack_bf16(y2, y3), pack_bf16(y4, y5), pack_bf16(y6, y7));
        }
    } else if constexpr (FULL && !is_qk_epi(EPI)) {      // (the q / k scatter epilogues sit at their register limit in the ping-pong kernel)
        uint4 val[NR * 2];                   // every LDS read in flight before the first store needs its data
#pragma unroll
        for (int it = 0; it < NR * 2; ++it) {
            const int r = it * 8 + (lane >> 3), c = lane & 7;
            val[it] = *(const uint4*)(stg + r * 128 + ((c ^ (r & 7)) << 4));
        }
#pragma unroll
        for (int it = 0; it < NR * 2; ++it)
            if (cok) store_row8<EPI, FULL>(p, m_base + it * 8 + (lane >> 3), n_base + (lane & 7) * 8, n_base, val[it]);
    } else {
#pragma unroll
        for (int it = 0; it < NR * 2; ++it) {
            const int r = it * 8 + (lane >> 3), c = lane & 7;
            const uint4 val = *(const uint4*)(stg + r * 128 + ((c ^ (r & 7)) << 4));
            const int m = m_base + r;
            if (cok && (FULL || m < p.M)) store_row8<EPI, FULL>(p, m, n_base + c * 8, n_base, val);
        }
    }
}

template <int BM, int BN, int WM, int WN, int EPI, bool CONV = false>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(GemmParams p) {
    using C = Cfg<BM, BN, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware tile mapping: consecutive tiles -> same XCD (blockIdx.x % 8 is the XCD)
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int nsplit = (EPI == EPI_F32 && p.k_split > 1) ? p.k_split : 1;
    const int nblk = ntm * ntn * nsplit;
    int bid = blockIdx.x;
    {
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    const int split = bid / (ntm * ntn);          // (splits of one tile run on different XCDs: they share no operand bytes)
    bid -= split * (ntm * ntn);
    int tm, tn;
    tile_coords(bid, ntm, ntn, p.raster_gm, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging addresses: lane -> (row in 8-row group, physical 16B chunk)
    const int srow = lane >> 3, spc = lane & 7;
    const bf16_t* srcA[C::GA];
    const bf16_t* srcW[C::GW];
    int cy[C::GA], cx[C::GA], ct[C::GA];                  // CONV: output pixel (and frame) of the row; srcA = frame base + chunk
#pragma unroll
    for (int i = 0; i < C::GA; ++i) {
        const int row = (wave + i * C::NW) * 8 + srow;   // row inside the A tile
        const int c = spc ^ ((row >> 1) & 7);             // logical chunk stored at physical chunk spc
        int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;  // clamp: rows beyond M are never stored
        if constexpr (CONV) {
            const int hw = p.conv_h * p.conv_w;
            const int bi = gm / hw, pix = gm - bi * hw;       // bi = output frame (b * conv_t + t), or the image index in 2-D
            cy[i] = pix / p.conv_w; cx[i] = pix - cy[i] * p.conv_w;
            int fi = bi;
            ct[i] = 0;
            if (p.conv_t > 0) {
                const int b = bi / p.conv_t;
                ct[i] = bi - b * p.conv_t;
                fi = b * p.conv_t_in + ct[i];
            }
            srcA[i] = p.A + (long)fi * ((hw >> (2 * p.conv_up)) * (long)p.conv_cin) + c * 8;
        } else {
            srcA[i] = p.A + (long)gm * p.lda + c * 8;
        }
    }
    const bf16_t* zsrc = nullptr;
    if constexpr (CONV) zsrc = p.zero_page + spc * 8;
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
        if constexpr (CONV) {
            // K-tile kt lies inside one tap (conv_cin % 64 == 0): gather the shifted pixel's channels, zeros outside
            const int k0 = kt * BK;
            const int tap = k0 / p.conv_cin, c0 = k0 - tap * p.conv_cin;
            const int ks = p.conv_ks ? p.conv_ks : 3, nsp = ks * ks;
            const int jt = tap / nsp, sp = tap - jt * nsp;                 // temporal tap, spatial tap
            const int df = p.conv_kt > 1 ? jt - (p.conv_kt - 1) : 0;        // frame offset <= 0 (causal)
            const int ky = sp / ks, dy = ky - (ks >> 1), dx = sp - ky * ks - (ks >> 1);
            const int win = p.conv_w >> p.conv_up;
            const long foff = (long)df * ((long)(p.conv_h >> p.conv_up) * win * p.conv_cin);
#pragma unroll
            for (int i = 0; i < C::GA; ++i) {
                const int yy = cy[i] + dy, xx = cx[i] + dx;
                const bool ok = (unsigned)yy < (unsigned)p.conv_h && (unsigned)xx < (unsigned)p.conv_w && ct[i] + df >= 0;
                const bf16_t* src = srcA[i] + foff + ((long)((yy >> p.conv_up) * win + (xx >> p.conv_up)) * p.conv_cin + c0);
                glds16(ok ? src : zsrc, base + (wave + i * C::NW) * 1024);
            }
        } else {
#pragma unroll
            for (int i = 0; i < C::GA; ++i) glds16(srcA[i] + ko, base + (wave + i * C::NW) * 1024);
        }
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

    const int nt_all = p.K / BK;
    const int kt0 = (int)((long)nt_all * split / nsplit);
    const int nt = (int)((long)nt_all * (split + 1) / nsplit) - kt0;      // K-tiles of this split (all of them without split-K)
    if (nt > 0) stage(kt0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // tile t landed for every wave; everyone is done reading the other buffer
        if (t + 1 < nt) stage(kt0 + t + 1, (t + 1) & 1);
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

    // ---- epilogue
    if constexpr (EPI == EPI_F32 || EPI == EPI_IMG) {
        // straight from the accumulator layout: fp32 scores (16 B per lane) / the <= 4 image channels of conv_out
        const int mrow = m0 + wm * C::TM + frow;
        const int ncol = n0 + wn * C::TN + 4 * fkg;
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
            for (int j = 0; j < C::NI; ++j) {
                const int m = mrow + i * 16, n = ncol + j * 16;
                if (m >= p.M || n >= p.N) continue;
                if constexpr (EPI == EPI_F32) {
                    const float sc = p.q_scale;
                    float* op = p.out_f32 + (long)split * p.split_stride + (long)m * p.ldo + n;
                    if (n + 4 <= p.N && (p.ldo & 3) == 0) {
                        *(float4*)op = make_float4(acc[i][j][0] * sc, acc[i][j][1] * sc, acc[i][j][2] * sc, acc[i][j][3] * sc);
                    } else {
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.N) op[e] = acc[i][j][e] * sc;
                    }
                } else {
                    const int hw = p.conv_h * p.conv_w;
                    const int bi = m / hw, pix = m - bi * hw;
                    for (int e = 0; e < 4; ++e) {
                        if (n + e >= p.N) break;
                        // bf16 VAE semantics: the conv output and the denormalised image are each rounded to bf16
                        float y = round_bf16(acc[i][j][e] + p.bias[n + e]);
                        if (p.img_clamp) y = fminf(fmaxf(y, -1.f), 1.f);
                        if (p.img_post) y = fminf(fmaxf(round_bf16(y * 0.5f + 0.5f), 0.f), 1.f);
                        const long o = ((long)bi * p.N + n + e) * hw + pix;
                        if (p.out_f32) p.out_f32[o] = y; else p.out[o] = f2bf(y);
                    }
                }
            }
    } else if constexpr (EPI == EPI_UNPATCH) {
        // tiny N (64): scalar scatter straight from the accumulator layout
        const int mrow = m0 + wm * C::TM + frow;
        const int ncol = n0 + wn * C::TN + 4 * fkg;
#pragma unroll
        for (int i = 0; i < C::MI; ++i)
#pragma unroll
            for (int j = 0; j < C::NI; ++j) {
                const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                unpatch_store(p, mrow + i * 16, ncol + j * 16, v);
            }
    } else {
        __syncthreads();  // every wave is done reading the operand ring: reuse it as epilogue staging
        char* stg = smem + wave * 8192;
#pragma unroll
        for (int hh = 0; hh < C::MI / 4; ++hh) {
            f32x4 a4[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) a4[i][j] = acc[hh * 4 + i][j];
            epilogue_part<EPI, 4>(p, a4, m0 + wm * C::TM + hh * 64, n0 + wn * C::TN, stg, lane);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Mid-size kernel (round 6): 128x192 (MI = 4, NJ = 6) or 192x128 (MI = 6, NJ = 4) output tiles, ONE 4-wave workgroup per CU, one wave per SIMD,
// each a 64x96 (96x64) register tile.  Why this shape: the GEMMs of a 4096-row image stream (512^2 at forward batch 4 -- the reference's
// shipped example, examples/grpo/full/sd3_5/default.yaml:49-54 -- and 1024^2 at B = 1) are 6.3 M outputs at N = 1536 = 24 576 per CU.  The
// 128x128 kernel cuts that into 384 tiles on 512 slots: half the CUs run a co-resident pair, the other half one tile, and on the busy half
// every SIMD carries two 64x64 wave tiles = 8192 outputs.  128x192 tiles are 256 tiles = one per CU with 6144 outputs per SIMD -- a quarter
// less work on the critical SIMD -- and N = 3072 / 6144 (and 8192 rows) give whole multiples of 256 as well; the transposed V^T projection
// (M = 1536 features x N = 4096 tokens) gets the same from 192x128.  (Round 5's 128x192 attempt kept the 64x64 wave tile on SIX waves: two
// SIMDs carried two waves, i.e. the same 8192 outputs on the critical SIMD -- measured +-0 and dropped)
//   * LDS bytes per MFMA: a 64x96 wave tile reads 4 + 6 fragments per 24 MFMAs (0.42 ds_read_b128 per MFMA against 0.5 for 64x64);
//     per K-tile and CU: 80 ds_read_b128 + 40 KiB of LDS-DMA against 192 MFMAs.
//   * 4-slot operand ring (4 x 40 KiB = the CU's whole 160 KiB; the epilogue staging reuses it): the loads of K-tile t + 3 are issued during
//     K-tile t and retired with a COUNTED s_waitcnt vmcnt(10) one K-tile before they are read -- two K-tiles (>= 1500 cycles) of flight,
//     never vmcnt(0) in the steady state; ONE barrier per K-tile.
//   * fragments are double-buffered across the two k-steps of a K-tile: the ds_reads of step kk + 1 are issued in front of the 24 MFMAs of
//     step kk; `sched_group_barrier` interleaves reads / LDS-DMA issues with the MFMAs (one wave per SIMD: nothing else hides them).
//   * same v_mfma_f32_16x16x32_bf16 operand order and ascending-k accumulation per output element as every other kernel of this file, and the
//     shared fused epilogues: BIT-IDENTICAL to the 128x128 / ping-pong / 4-wave kernels (the batch-invariance tests mix them freely).
template <int MI, int NJ, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_mid_kernel(GemmParams p) {
    constexpr int TM = MI * 16, TN = NJ * 16, BM = 2 * TM, BN = 2 * TN;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128, NS = 4;
    constexpr int GA = BM / 8, GT = (BM + BN) / 8, GPW = GT / 4;          // 8-row LDS-DMA groups: per tile side, per K-tile, per wave
    static_assert(GT % 4 == 0 && NS * STAGE <= 160 * 1024 && (NJ == 4 || NJ == 6) && (MI == 4 || MI == 6), "mid-size tile");
    static_assert(EPI != EPI_F32 && EPI != EPI_IMG && EPI != EPI_UNPATCH, "accumulator-layout epilogues stay on gemm_kernel");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int nblk = ntm * ntn;
    int bid = blockIdx.x;
    {   // consecutive tile ids -> one XCD (blockIdx % 8 is the XCD), as gemm_kernel
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    int tm, tn;
    tile_coords(bid, ntm, ntn, p.raster_gm, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- LDS-DMA: group g (8 rows = 1 KiB of a slot) of a K-tile belongs to wave g & 3; groups [0, GA) are activation rows, the rest
    // weight rows.  Per lane: 32-bit byte offset from p.A / p.W (launcher: operands < 4 GiB) of its 16-byte chunk, swizzled at the source.
    const int srow = lane >> 3, spc = lane & 7;
    unsigned soff[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
        const int g = wave + 4 * i;
        const bool is_a = g < GA;
        const int row = (is_a ? g : g - GA) * 8 + srow;
        const int c = spc ^ ((row >> 1) & 7);
        if (is_a) {
            int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;          // clamp: rows beyond M are never stored
            soff[i] = ((unsigned)gm * (unsigned)p.lda + (unsigned)(c * 8)) * 2u;
        } else {
            int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
            soff[i] = ((unsigned)gn * (unsigned)p.ldw + (unsigned)(c * 8)) * 2u;
        }
    }
    // ---- fragment read offsets inside a slot
    const int frow = lane & 15, fkg = lane >> 4, fsw = frow >> 1;
    int offX[2], offW[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int pc = (kk * 4 + fkg) ^ fsw;
        offX[kk] = (wm * TM + frow) * 128 + pc * 16;
        offW[kk] = A_BYTES + (wn * TN + frow) * 128 + pc * 16;
    }

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[2][MI], wf[2][NJ];

    // fragment q of a k-step, in the order the MFMAs (i-major) need them: x[0], w[0 .. NJ-1], x[1 .. MI-1]
    auto read_frag = [&](int buf, const char* sb, int kk, int q) {
        if (q == 0) xf[buf][0] = *(const bf16x8*)(sb + offX[kk]);
        else if (q <= NJ) wf[buf][q - 1] = *(const bf16x8*)(sb + offW[kk] + (q - 1) * 2048);
        else xf[buf][q - NJ] = *(const bf16x8*)(sb + offX[kk] + (q - NJ) * 2048);
    };
    // LDS-DMA group i (of GPW) of K-tile kt into ring slot `slot`: SGPR base + 32-bit VGPR offset (hipcc's builtin forms a 64-bit address
    // per lane with v_lshl_add_u64 in front of every load: twice the address bytes on the way to the texture-address unit)
    const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem);
    auto stage_one = [&](int kt, int slot, int i) {
        const int g = wave + 4 * i;
        const char* gb = (g < GA ? (const char*)p.A : (const char*)p.W) + (long)kt * (BK * 2);      // wave-uniform
        const unsigned dst = lds0 + slot * STAGE + g * 1024;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(dst), "v"(soff[i]), "s"(gb) : "memory");
    };
    // One half of a K-tile: the MI * NJ MFMAs of k-step `buf` in a FIXED issue order with, pinned between them (`sched_barrier(0)`: nothing
    // crosses; hipcc's own scheduler bunches the loads in front of the MFMAs, and its sched_group_barrier pipeline did not hold for the
    // LDS-DMA half), one fragment read of the NEXT k-step behind every second MFMA and one LDS-DMA issue behind every fourth -- the wave is
    // alone on its SIMD, so whatever it issues between two MFMAs must fit the 16 cycles the first one occupies the matrix pipe.
    // (Measured and removed, profiles/r06c_*: per-wave STAGGERED LDS-DMA slots -- wave w issuing behind MFMA 4 q + w so that the four waves, which run
    // in lockstep between barriers, do not reach the CU's one texture-address unit in the same cycle -- moved nothing, +-1 %: four copies of the loop.)
#define MID_HALF(buf, rd_sb, rd_kk, ld_kt, ld_slot, ld_first)                                                    \
    do {                                                                                                         \
        _Pragma("unroll") for (int n = 0; n < MI * NJ; ++n) {                                                    \
            acc[n / NJ][n % NJ] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[buf][n % NJ], xf[buf][n / NJ], acc[n / NJ][n % NJ], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            if ((n & 1) == 0 && n / 2 < MI + NJ) read_frag((buf) ^ 1, rd_sb, rd_kk, n / 2);                      \
            if ((n & 3) == 1 && n / 4 < GPW / 2) stage_one(ld_kt, ld_slot, (ld_first) + n / 4);                   \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
        }                                                                                                        \
    } while (0)
    static_assert(MI + NJ == GPW && GPW == 10 && MI * NJ >= 2 * GPW, "10 fragment reads and 5 LDS-DMA issues per half");

    const int nt = p.K / BK;
    auto ktile = [&](int kt) { return kt < nt ? kt : nt - 1; };      // (the tail's surplus prefetches re-load the last K-tile: never read)
    // prologue: K-tiles 0, 1 and the first half of K-tile 2 in flight, K-tile 0 retired
#pragma unroll
    for (int i = 0; i < GPW; ++i) stage_one(0, 0, i);
#pragma unroll
    for (int i = 0; i < GPW; ++i) stage_one(ktile(1), 1, i);
#pragma unroll
    for (int i = 0; i < GPW / 2; ++i) stage_one(ktile(2), 2, i);
    asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < MI + NJ; ++q) read_frag(0, smem, 0, q);
    // One K-tile = ONE basic block, the same for every t (a branch around the LDS-DMA issues or the fragment reads would fence them off
    // from the MFMAs they are interleaved with, and peeled tail copies made the register allocator shuffle the accumulators between the
    // copies): the last K-tiles, which have nothing left to prefetch, re-load K-tile nt - 1 into the slots that became free and read
    // fragments nobody uses (as the ping-pong kernel's last prefetch does); <= 3 x 40 KiB of L2 hits per tile.
    for (int t = 0; t < nt; ++t) {
        const char* sb = smem + (t & 3) * STAGE;
        // first half: k-step 0 of K-tile t; fragments of k-step 1; second half of the LDS-DMA of K-tile t + 2 (slot of K-tile t - 2: free
        // since the barrier of the previous iteration)
        MID_HALF(0, sb, 1, ktile(t + 2), (t + 2) & 3, GPW / 2);
        // K-tile t + 1 has landed (this wave's share; the barrier makes it everybody's); K-tile t + 2 stays in flight
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // ... and every wave is done with the slot of K-tile t - 1
        __builtin_amdgcn_sched_barrier(0);
        // second half: k-step 1; fragments of k-step 0 of K-tile t + 1; first half of the LDS-DMA of K-tile t + 3 into the slot K-tile t - 1 left
        MID_HALF(1, smem + ((t + 1) & 3) * STAGE, 0, ktile(t + 3), (t + 3) & 3, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the surplus prefetches of the tail have landed before the ring becomes staging
#undef MID_HALF

    // ---- epilogue: the wave's tile in chunks of <= 64 rows x {64, 32} columns through the shared fused epilogues
    __syncthreads();  // every wave is done reading the ring (and no LDS-DMA is in flight: the last waits were vmcnt(0)): reuse it as staging
    char* stg = smem + wave * 8192;
    const bool full_tile = (m0 + BM <= p.M) && (n0 + BN <= p.N) && ((p.ldo & 7) == 0 || EPI == EPI_VT) &&
                           (EPI != EPI_VT || ((p.rows_per_sample | p.s_off) & 7) == 0);
    const int mw = m0 + wm * TM, nw_ = n0 + wn * TN;
#define MID_CHUNK(R0, NR, C0, NC)                                                                              \
    {                                                                                                          \
        f32x4 a_[NR][NC];                                                                                      \
        _Pragma("unroll") for (int i = 0; i < NR; ++i) _Pragma("unroll") for (int j = 0; j < NC; ++j) a_[i][j] = acc[R0 + i][C0 + j]; \
        if (full_tile) epilogue_part<EPI, NR, true, false, NC>(p, a_, mw + (R0) * 16, nw_ + (C0) * 16, stg, lane); \
        else epilogue_part<EPI, NR, false, false, NC>(p, a_, mw + (R0) * 16, nw_ + (C0) * 16, stg, lane);        \
    }
    MID_CHUNK(0, 4, 0, 4)
    if constexpr (NJ == 6) MID_CHUNK(0, 4, 4, 2)
    if constexpr (MI == 6) MID_CHUNK(4, 2, 0, 4)
    static_assert(!(MI == 6 && NJ == 6), "one of the two tile sides is 64 wide per wave");
#undef MID_CHUNK
}

template <int MI, int NJ, int EPI>
hipError_t launch_mid(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm_mid_kernel<MI, NJ, EPI>;
    constexpr int BM = MI * 32, BN = NJ * 32;
    constexpr int smem = 4 * (BM + BN) * 128;             // the whole 160 KiB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3(ntm * ntn), dim3(256), smem, stream, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// 256x256x64 "ping-pong" kernel (the large-M GEMMs of the image stream).
//
// 8 waves = 2 groups of 4 (group g owns tile rows [128g, 128g+128), wave (g, wn) the 128x64 sub-tile
// at columns 64*wn).  Waves w and w+4 share a SIMD; the two groups run the same 4-phase-per-K-tile
// schedule staggered by one barrier interval, so that on every SIMD one wave is inside a 16-MFMA
// cluster (one 64x32 quadrant x K=64, s_setprio 1) while its partner issues the ds_read_b128s of its
// next quadrant and the global_load_lds prefetches -- matrix beside memory, never matrix beside
// matrix (MI355X_MICROARCH.md "Two waves per SIMD").  Loads for K-tile t+1 are issued one unit
// (128 rows x 128 B = 16 KiB, 2 wave-instructions per wave) per phase during K-tile t and retired with
// COUNTED s_waitcnt vmcnt(4) (never 0 in the main loop): every unit has >= 4 barrier intervals to
// land, and a unit is first read one phase after the wait + barrier that retires it for BOTH groups.
//   unit order U0 = A rows of quadrant-row 0 (both groups), U2/U3 = W rows of quadrant-col 0/1, U1 = A
//   rows of quadrant-row 1;  phase P0: (0,0) reads A0,B0 | P1: (0,1) reads B1 | P2: (1,1) reads A1 |
//   P3: (1,0) reads nothing;  prefetch order for tile t+1: U0, U2, U3, U1.
// DBG (ablation builds only, EPI_BIAS, results are garbage): bit 0 no K-loop prefetch, bit 1 no LDS fragment reads after the first K-tile,
// bit 2 no epilogue at all (accumulators just reset), bit 3 epilogue without its global stores; bit 4 (alone): the trace build (valid results)
template <int EPI, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_pp_kernel(GemmParams p) {
    constexpr int BM = 256, BN = 256, TM = 128, TN = 64;
    constexpr int A_BYTES = BM * BK * 2, STAGE = 2 * A_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;

    // ---- persistent tile schedule: the grid is min(#tiles, #CUs) workgroups; workgroup b (XCD b % 8) walks
    // tiles  chunk(b % 8) + (b / 8) + i * (gridDim / 8): at any time the workgroups of one XCD hold consecutive
    // tiles (same activation row panel / neighbouring weight panels) in that XCD's private L2.
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int nblk = ntm * ntn;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_xcd = (int)(gridDim.x >> 3);            // gridDim is a multiple of 8 (launcher)
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int chunk_lo = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int chunk_n = q8 + (xcd < r8 ? 1 : 0);

    // ---- staging: unit u in {0:A q-row 0, 1:A q-row 1, 2:W q-col 0, 3:W q-col 1}; 16 groups of 8 rows per
    // unit, wave w stages groups j = w and w + 8 of every unit.  Per-lane source = uniform base (+ k offset,
    // scalar) + 32-bit byte offset (VGPR): 8 VGPRs of addressing state for the whole kernel.
    const int srow = lane >> 3, spc = lane & 7;
    unsigned soff[4][2];   // byte offset of this lane's 16-byte chunk from p.A (u < 2) / p.W (u >= 2)
    int ldsoff[4][2];
    int m0 = 0, n0 = 0;
    auto set_tile = [&](int tile) {
        int tm, tn;
        tile_coords(tile, ntm, ntn, p.raster_gm, tm, tn);
        m0 = tm * BM; n0 = tn * BN;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = wave + 8 * i;
                int g8;  // 8-row group index inside the 256-row operand tile
                if (u == 0) g8 = (j < 8) ? j : 16 + (j - 8);
                else if (u == 1) g8 = (j < 8) ? 8 + j : 24 + (j - 8);
                else g8 = (j >> 2) * 8 + (u == 3 ? 4 : 0) + (j & 3);
                const int row = g8 * 8 + srow;
                const int c = spc ^ ((row >> 1) & 7);
                if (u < 2) {
                    int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;
                    soff[u][i] = ((unsigned)gm * (unsigned)p.lda + (unsigned)(c * 8)) * 2u;
                    ldsoff[u][i] = g8 * 1024;
                } else {
                    int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
                    soff[u][i] = ((unsigned)gn * (unsigned)p.ldw + (unsigned)(c * 8)) * 2u;
                    ldsoff[u][i] = A_BYTES + g8 * 1024;
                }
            }
    };
    auto stage_unit = [&](int u, long ko, char* base) {
        if constexpr ((DBG & 1) != 0) { if (ko != 0) return; }
        const char* gb = (u < 2 ? (const char*)p.A : (const char*)p.W) + ko * 2;   // uniform
        glds16(gb + soff[u][0], base + ldsoff[u][0]);
        glds16(gb + soff[u][1], base + ldsoff[u][1]);
    };

    // ---- fragment read offsets inside a stage
    const int frow = lane & 15, fkg = lane >> 4, fsw = frow >> 1;
    int offX[2], offW[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int pc = (kk * 4 + fkg) ^ fsw;
        offX[kk] = (grp * TM + frow) * 128 + pc * 16;
        offW[kk] = A_BYTES + (wn * TN + frow) * 128 + pc * 16;
    }

    f32x4 acc[8][4];
    bf16x8 af[4][2], bfr[2][2][2];

#define PP_BARRIER()                              \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)
    bool dbg_noread = false;
#define PP_READ_A(qm)                                                                                   \
    if (!((DBG & 2) && dbg_noread)) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)     \
        af[mi][kk] = *(const bf16x8*)(sb + offX[kk] + (qm) * 8192 + mi * 2048)
#define PP_READ_B(qn)                                                                                   \
    if (!((DBG & 2) && dbg_noread)) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)     \
        bfr[qn][ni][kk] = *(const bf16x8*)(sb + offW[kk] + (qn) * 4096 + ni * 2048)
#define PP_MFMA(qm, qn)                                                                                 \
    do {                                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                              \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                  \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) \
            _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                             \
                acc[(qm) * 4 + mi][(qn) * 2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(              \
                    bfr[qn][ni][kk], af[mi][kk], acc[(qm) * 4 + mi][(qn) * 2 + ni], 0, 0, 0);            \
        __builtin_amdgcn_s_setprio(0);                                                                  \
    } while (0)

    const int nt = p.K / BK;
    if (slot >= chunk_n) return;  // (only when there are fewer tiles than workgroups)
    const int my_tiles = (chunk_n - slot + per_xcd - 1) / per_xcd;
    // K-tile 0 of the first tile completely, then ONE continuous (tile, K-tile) stream: the prefetch of the
    // step after a tile's last K-tile already belongs to the next tile, so the cold first K-tile of a tile
    // lands under the previous tile's last MFMA phases and epilogue instead of in front of an idle CU.
    set_tile(chunk_lo + slot);
#pragma unroll
    for (int u = 0; u < 4; ++u) stage_unit(u, 0, smem);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BARRIER();
    if (grp == 1) PP_BARRIER();  // stagger: group 1 runs one barrier interval behind group 0

    // epilogue staging: a dedicated 32 KiB behind the 128 KiB operand ring (the CU has 160 KiB): no interplay with the
    // LDS-DMA prefetches that waves already past their epilogue issue into either ring stage
    const int stg_off = wave * 4096;
    int cur_m0 = m0, cur_n0 = n0;
    int tile_i = 0, t = 0, par = 0;
    long long* trc = nullptr;
    if constexpr ((DBG & 16) != 0) {      // trace build only (scripts/gemm_trace.py): production kernels carry no trace state
        if (p.trace && (wave == 0 || wave == 4) && lane == 0) trc = p.trace + ((long)blockIdx.x * 16 * 2 + grp) * 4;
    }
    // slot 15 of the trace: s_memrealtime (100 MHz, one time base for the whole device) at WG start / end; addressed from p.trace at the
    // two use sites (a second per-lane pointer held across the main loop costs VGPRs the epilogues do not have)
    if (trc) { trc[0] = __builtin_amdgcn_s_memtime(); trc[1] = trc[0]; trc[15 * 8] = __builtin_amdgcn_s_memrealtime(); }
#define PP_EPILOGUE(SB)                                                                                        \
    do {                                                                                                          \
        /* no wave has an LDS read of this stage pending here (P3 reads nothing; every wave's P2 reads were  */  \
        /* waited for before its P2 MFMAs, which precede the barrier this wave just passed).                 */  \
        /* Retire this wave's share of the next step's operands NOW (issued >= 2 intervals ago), before the  */  \
        /* epilogue queues stores behind them: the step after an epilogue then needs no wait in P0 / P1.     */  \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
        if (trc) trc[2] = __builtin_amdgcn_s_memtime();                                                          \
        char* stg = smem + 2 * STAGE + stg_off; (void)(SB);                                                       \
        const bool full_tile = (em0 + BM <= p.M) && (en0 + BN <= p.N) && ((p.ldo & 7) == 0 || is_qk_epi(EPI) || EPI == EPI_VT) && \
                               (EPI != EPI_VT || ((p.rows_per_sample | p.s_off) & 7) == 0);                       \
        _Pragma("unroll") for (int qq = 0; qq < 4; ++qq) {                                                        \
            f32x4 a2[2][4];                                                                                       \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) {          \
                a2[i][j] = acc[qq * 2 + i][j];                                                                    \
                acc[qq * 2 + i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};                                                 \
            }                                                                                                     \
            if constexpr ((DBG & 4) != 0) { float sink = 0.f;                                                                  \
                _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                     \
                    sink += (a2[i][j][0] + a2[i][j][1]) + (a2[i][j][2] + a2[i][j][3]);                                        \
                if (sink == 12345.678f) p.out[lane] = 1; }                                                                    \
            else if constexpr ((DBG & 8) != 0) { GemmParams pd = p; pd.M = 0; pd.N = 0;                                        \
                epilogue_part<EPI, 2, false>(pd, a2, em0 + grp * TM + qq * 32, en0 + wn * TN, stg, lane); }                    \
            else if (full_tile) epilogue_part<EPI, 2, true>(p, a2, em0 + grp * TM + qq * 32, en0 + wn * TN, stg, lane); \
            else epilogue_part<EPI, 2, false>(p, a2, em0 + grp * TM + qq * 32, en0 + wn * TN, stg, lane);          \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
        }                                                                                                         \
        if (trc) {                                                                                                \
            trc[3] = __builtin_amdgcn_s_memtime();                                                                \
            trc += 8;                                                                                             \
            trc[0] = trc[-8 + 3]; trc[1] = trc[0];                                                                \
        }                                                                                                         \
    } while (0)

    const int nsteps = my_tiles * nt;
    bool after_epi = false;  // this step's operands were fully retired by the wait that opens PP_EPILOGUE
    for (int sidx = 0; sidx < nsteps - 1; ++sidx) {
        const bool last_k = (t == nt - 1);
        const char* sb = smem + par * STAGE;
        char* nb = smem + (par ^ 1) * STAGE;
        const int em0 = cur_m0, en0 = cur_n0;  // the tile the accumulators belong to
        long ko = (long)(t + 1) * BK;
        if (last_k) {
            ++tile_i;
            set_tile(chunk_lo + slot + tile_i * per_xcd);
            ko = 0; cur_m0 = m0; cur_n0 = n0;
        }
        // P0: quadrant (0,0)
        PP_READ_A(0); PP_READ_B(0);
        stage_unit(0, ko, nb);
        if (!after_epi) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // retires U3 of this step: first read in P1
        PP_BARRIER(); PP_MFMA(0, 0); PP_BARRIER();
        // P1: quadrant (0,1)
        PP_READ_B(1);
        stage_unit(2, ko, nb);
        if (!after_epi) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // retires U1 of this step: first read in P2
        PP_BARRIER(); PP_MFMA(0, 1); PP_BARRIER();
        // P2: quadrant (1,1)
        PP_READ_A(1);
        stage_unit(3, ko, nb);
        PP_BARRIER(); PP_MFMA(1, 1); PP_BARRIER();
        // P3: quadrant (1,0)
        stage_unit(1, ko, nb);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // retires U0,U2 of the next step (and any epilogue stores)
        PP_BARRIER(); PP_MFMA(1, 0); PP_BARRIER();
        after_epi = false;
        if constexpr ((DBG & 2) != 0) dbg_noread = true;
        if (last_k) {
            PP_EPILOGUE(sb);
            t = 0;
            after_epi = true;
        } else {
            ++t;
        }
        par ^= 1;
    }
    {   // very last step of this workgroup: nothing left to prefetch
        const char* sb = smem + par * STAGE;
        const int em0 = cur_m0, en0 = cur_n0;
        PP_READ_A(0); PP_READ_B(0);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        PP_BARRIER(); PP_MFMA(0, 0); PP_BARRIER();
        PP_READ_B(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP_BARRIER(); PP_MFMA(0, 1); PP_BARRIER();
        PP_READ_A(1);
        PP_BARRIER(); PP_MFMA(1, 1); PP_BARRIER();
        PP_BARRIER(); PP_MFMA(1, 0);
        if (grp == 0) PP_BARRIER();  // pairs with group 1's barrier in front of its last MFMA cluster
        PP_EPILOGUE(sb);
        if constexpr ((DBG & 16) != 0) {
            if (p.trace && (wave == 0 || wave == 4) && lane == 0)
                p.trace[((long)blockIdx.x * 16 * 2 + grp) * 4 + 15 * 8 + 3] = __builtin_amdgcn_s_memrealtime();
        }
    }
#undef PP_EPILOGUE
#undef PP_BARRIER
#undef PP_READ_A
#undef PP_READ_B
#undef PP_MFMA
}

template <int EPI, int DBG = 0>
hipError_t launch_pp(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm_pp_kernel<EPI, DBG>;
    constexpr int smem = 2 * 2 * 256 * BK * 2 + 8 * 4096;   // operand ring + epilogue staging = 160 KiB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int ntm = (p.M + 255) / 256, ntn = (p.N + 255) / 256;
    int grid = ntm * ntn;
    if (grid > 256) grid = 256;            // one 512-thread workgroup (128 KiB LDS) per CU, persistent
    grid = (grid + 7) / 8 * 8;             // whole XCD rounds (surplus workgroups exit at once)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, stream, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// 256x256x64 kernel with 4 waves, one per SIMD, each a 128x128 register tile (256 AGPR accumulators), and a HAND-SCHEDULED main loop
// (gemm_w4_asm.inc, generated by gen_gemm_w4.py: its docstring has the schedule).  Against the 8-wave ping-pong kernel: LDS bytes read per
// MFMA -33 % (a wave's fragment feeds 8 MFMAs instead of 4 / 8), one barrier per K-tile instead of eight, 16 LDS-DMA instructions per wave and
// K-tile interleaved one per four MFMAs.  This is the structure hipBLASLt's own gfx950 bf16 kernels use (MT256x256x64, MIWaveTile 8x8, 256
// threads), which were 10 % ahead of the ping-pong kernel on every model shape (profiles/r03a_clock_under_load.txt).
// The C++ shell owns tile scheduling, addressing and the shared fused epilogues; the asm statement owns a[0:255], v[120:247], s[80:91]
// (clobber lists) -- hipcc's register allocator does not terminate on 64 "+a" operands, so the accumulators are handed over by register
// NUMBER: v_accvgpr_read statements right behind the loop.  The shell must therefore never make hipcc touch an AGPR itself (it has no MFMA
// and stays far below 248 VGPRs; `make check-w4` greps the ISA for stray AGPR writes).
// Requirements (launcher): M % 256 == 0, N % 256 == 0, K % 128 == 0.  ABL: ablation builds for the microbenchmark only (results garbage).
#include "gemm_w4_asm.inc"

template <int EPI, int ABL = 0>
__global__ __launch_bounds__(256) void gemm_w4_kernel(GemmParams p) {
    constexpr int BM = 256, BN = 256;
    constexpr int STAGE = 65536;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // persistent tile schedule (as gemm_pp_kernel)
    const int ntm = p.M / BM, ntn = p.N / BN;
    const int nblk = ntm * ntn;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_xcd = (int)(gridDim.x >> 3);
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int chunk_lo = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int chunk_n = q8 + (xcd < r8 ? 1 : 0);
    if (slot >= chunk_n) return;
    const int my_tiles = (chunk_n - slot + per_xcd - 1) / per_xcd;

    // LDS-DMA: wave w fills rows [64 w, 64 w + 64) of both operands, 8 rows (1 KiB) per instruction; lane -> (row = l >> 3, LDS chunk
    // position l & 7); the position holds source chunk  pos ^ ((row >> 1) & 7)
    unsigned ga[8], gw[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int row = wave * 64 + g * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        ga[g] = ((unsigned)row * (unsigned)p.lda + (unsigned)(c * 8)) * 2u;
        gw[g] = ((unsigned)row * (unsigned)p.ldw + (unsigned)(c * 8)) * 2u;
    }
    // fragment reads: lane -> (row l & 15 of a 16-row block, k-chunk kk * 4 + (l >> 4)), position = chunk ^ ((row >> 1) & 7)
    const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem);
    unsigned lx[2][2], lw[2][2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const unsigned in_blk = (unsigned)((lane & 15) * 128 + (((kk * 4 + (lane >> 4)) ^ ((lane & 15) >> 1)) << 4));
            lx[st][kk] = lds0 + st * STAGE + wm * 16384 + in_blk;
            lw[st][kk] = lds0 + st * STAGE + 32768 + wn * 16384 + in_blk;
        }
    const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + wave * 8192);       // this wave's LDS-DMA destination inside a region

    auto tile_bases = [&](int tile, unsigned long long& a, unsigned long long& w, int& m0, int& n0) {
        int tm, tn;
        tile_coords(tile, ntm, ntn, p.raster_gm, tm, tn);
        m0 = tm * BM; n0 = tn * BN;
        a = (unsigned long long)p.A + (unsigned long long)m0 * (unsigned long long)p.lda * 2ull;
        w = (unsigned long long)p.W + (unsigned long long)n0 * (unsigned long long)p.ldw * 2ull;
    };
    unsigned long long cA, cW, nA, nW;
    int m0, n0, m0n, n0n;
    tile_bases(chunk_lo + slot, cA, cW, m0, n0);
    {   // K-tile 0 of the first tile -> stage 0
        char* dst = smem + wave * 8192;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            glds16((const char*)cA + ga[g], dst + g * 1024);
            glds16((const char*)cW + gw[g], dst + 32768 + g * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const int pairs = p.K / 128 - 1;
    char* stg = smem + 2 * STAGE + wave * 8192;
    for (int ti = 0; ti < my_tiles; ++ti) {
        if (ti + 1 < my_tiles) tile_bases(chunk_lo + slot + (ti + 1) * per_xcd, nA, nW, m0n, n0n);
        else { nA = cA; nW = cW; m0n = m0; n0n = n0; }       // nothing follows: the last prefetch re-reads this tile's first K-tile (unused)
#define W4_OPERANDS                                                                                                                      \
        : : [ga0] "v"(ga[0]), [ga1] "v"(ga[1]), [ga2] "v"(ga[2]), [ga3] "v"(ga[3]), [ga4] "v"(ga[4]), [ga5] "v"(ga[5]), [ga6] "v"(ga[6]),    \
            [ga7] "v"(ga[7]), [gw0] "v"(gw[0]), [gw1] "v"(gw[1]), [gw2] "v"(gw[2]), [gw3] "v"(gw[3]), [gw4] "v"(gw[4]), [gw5] "v"(gw[5]),    \
            [gw6] "v"(gw[6]), [gw7] "v"(gw[7]), [lx00] "v"(lx[0][0]), [lx01] "v"(lx[0][1]), [lx10] "v"(lx[1][0]), [lx11] "v"(lx[1][1]),      \
            [lw00] "v"(lw[0][0]), [lw01] "v"(lw[0][1]), [lw10] "v"(lw[1][0]), [lw11] "v"(lw[1][1]), [cA] "s"(cA), [cW] "s"(cW), [nA] "s"(nA), \
            [nW] "s"(nW), [pairs] "s"(pairs), [ldsw] "s"(ldsw)                                                                             \
        : W4_CLOBBERS
        if constexpr (ABL == 0) asm volatile(W4_LOOP_ASM W4_OPERANDS);
        else if constexpr (ABL == 1) asm volatile(W4_LOOP_ASM_NOLOAD W4_OPERANDS);
        else if constexpr (ABL == 2) asm volatile(W4_LOOP_ASM_NOREAD W4_OPERANDS);
        else if constexpr (ABL == 3) asm volatile(W4_LOOP_ASM_MFMA_ONLY W4_OPERANDS);
        else if constexpr (ABL == 4) asm volatile(W4_LOOP_ASM_S1 W4_OPERANDS);
        else asm volatile(W4_LOOP_ASM_S3 W4_OPERANDS);
#undef W4_OPERANDS
        // ---- epilogue: eight 32x64 chunks of the wave's 128x128 tile through the shared fused epilogues (wave-private LDS staging)
#define W4_CHUNK(Q, C)                                                                                         \
        {                                                                                                      \
            f32x4 a2[2][4];                                                                                    \
            W4_READ_CHUNK_##Q##_##C(a2)                                                                        \
            const int mb = m0 + wm * 128 + Q * 32, nb = n0 + wn * 128 + C * 64;                                \
            epilogue_part<EPI, 2, true, HAS_COLB>(p, a2, mb, nb, stg, lane_e, bpre);                         \
        }
        if constexpr (ABL == 0 || ABL >= 4) {
            // Everything the epilogue derives from the lane id is recomputed per tile from an OPAQUE copy of it: hipcc would otherwise hoist
            // that address arithmetic out of the tile loop and keep it alive across the asm statement, where only v0-v119 are free -- measured:
            // 176-320 bytes of scratch per lane, every reload followed by s_waitcnt vmcnt(0) with no second wave on the SIMD to hide it
            // (w4 1113 vs ping-pong 1302 TFLOP/s; profiles/r03d_gemm_w4_second_ab.txt).
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            constexpr bool HAS_COLB = EPI != EPI_VT && EPI != EPI_BIAS_ROW;
#define W4_COLUMN_HALF(C)                                                                                                \
            {                                                                                                            \
                BiasPre bpre;           /* one bias load per tile and column half: its four row chunks share it */       \
                if constexpr (HAS_COLB)                                                                                   \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                         \
                        bpre.v[j] = *(const float4*)(p.bias + n0 + wn * 128 + C * 64 + j * 16 + 4 * (lane_e >> 4));        \
                W4_CHUNK(0, C) W4_CHUNK(1, C) W4_CHUNK(2, C) W4_CHUNK(3, C)                                              \
            }
            W4_COLUMN_HALF(0) W4_COLUMN_HALF(1)
#undef W4_COLUMN_HALF
        }
#undef W4_CHUNK
        cA = nA; cW = nW; m0 = m0n; n0 = n0n;
    }
}

template <int EPI, int ABL = 0>
hipError_t launch_w4(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm_w4_kernel<EPI, ABL>;
    constexpr int smem = 2 * 65536 + 4 * 8192;            // operand stages + epilogue staging = 160 KiB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    int grid = (p.M / 256) * (p.N / 256);
    if (grid > 256) grid = 256;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, p);
    return hipGetLastError();
}
template <int EPI>
inline bool w4_ok(const GemmParams& p) {      // whole tiles only (the kernel has the guard-free epilogue path alone)
    return p.M % 256 == 0 && p.N % 256 == 0 && p.K % 128 == 0 && p.K >= 128 && ((p.ldo & 7) == 0 || is_qk_epi(EPI) || EPI == EPI_VT) &&
           (EPI != EPI_VT || ((p.rows_per_sample | p.s_off) & 7) == 0);
}

// ---------------------------------------------------------------------------------------------
// 256x192x64 kernel (round 6): the hand-scheduled 4-wave loop on a 4 x 1 wave grid, each wave a 64 x 192 register tile (192 AGPR accumulators;
// W6_LOOP_ASM of gemm_w4_asm.inc).  For grids that 256x256 tiles cut into 0.75 / 1.5 rounds of the 256 CUs -- 8192 rows x N = 1536 / 3072: the
// image stream of SD3.5 at B = 2, 1024^2 (192 / 384 tiles) -- 256x192 tiles are whole rounds (256 / 512) with a quarter less work on every CU
// than the 256x256 kernels' busy ones.  The wave tile spans the tile's whole width, so every 64-column chunk of the shared epilogues (a q/k head)
// lies inside one wave.  Same MFMA, operand order and ascending-k accumulation as every kernel of this file: bit-identical outputs.
// Requirements (launcher): M % 256 == 0, N % 192 == 0, K % 128 == 0.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_w6_kernel(GemmParams p) {
    constexpr int BM = 256, BN = 192;
    constexpr int STAGE = 57344, XW_OFF = 32768;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // persistent tile schedule (as gemm_pp_kernel)
    const int ntm = p.M / BM, ntn = p.N / BN;
    const int nblk = ntm * ntn;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_xcd = (int)(gridDim.x >> 3);
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int chunk_lo = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int chunk_n = q8 + (xcd < r8 ? 1 : 0);
    if (slot >= chunk_n) return;
    const int my_tiles = (chunk_n - slot + per_xcd - 1) / per_xcd;

    // LDS-DMA: wave w fills X rows [64 w, 64 w + 64) and W rows [48 w, 48 w + 48), 8 rows (1 KiB) per instruction; lane -> (row = l >> 3, LDS
    // chunk position l & 7); the position holds source chunk  pos ^ ((row >> 1) & 7)
    unsigned ga[8], gw[6];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int row = wave * 64 + g * 8 + (lane >> 3);
        ga[g] = ((unsigned)row * (unsigned)p.lda + (unsigned)((((lane & 7) ^ ((row >> 1) & 7))) * 8)) * 2u;
    }
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        const int row = wave * 48 + g * 8 + (lane >> 3);
        gw[g] = ((unsigned)row * (unsigned)p.ldw + (unsigned)((((lane & 7) ^ ((row >> 1) & 7))) * 8)) * 2u;
    }
    // fragment reads: lane -> (row l & 15 of a 16-row block, k-chunk kk * 4 + (l >> 4)), position = chunk ^ ((row >> 1) & 7); X blocks of this
    // wave's 64 rows, W blocks of the whole 192-row tile
    const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem);
    unsigned lx[2][2], lw[2][2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const unsigned in_blk = (unsigned)((lane & 15) * 128 + (((kk * 4 + (lane >> 4)) ^ ((lane & 15) >> 1)) << 4));
            lx[st][kk] = lds0 + st * STAGE + wave * 8192 + in_blk;
            lw[st][kk] = lds0 + st * STAGE + XW_OFF + in_blk;
        }
    const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + wave * 8192);                 // this wave's LDS-DMA destinations inside a stage
    const unsigned ldsww = __builtin_amdgcn_readfirstlane(lds0 + XW_OFF + wave * 6144);

    auto tile_bases = [&](int tile, unsigned long long& a, unsigned long long& w, int& m0, int& n0) {
        int tm, tn;
        tile_coords(tile, ntm, ntn, p.raster_gm, tm, tn);
        m0 = tm * BM; n0 = tn * BN;
        a = (unsigned long long)p.A + (unsigned long long)m0 * (unsigned long long)p.lda * 2ull;
        w = (unsigned long long)p.W + (unsigned long long)n0 * (unsigned long long)p.ldw * 2ull;
    };
    unsigned long long cA, cW, nA, nW;
    int m0, n0, m0n, n0n;
    tile_bases(chunk_lo + slot, cA, cW, m0, n0);
    {   // K-tile 0 of the first tile -> stage 0
#pragma unroll
        for (int g = 0; g < 8; ++g) glds16((const char*)cA + ga[g], smem + wave * 8192 + g * 1024);
#pragma unroll
        for (int g = 0; g < 6; ++g) glds16((const char*)cW + gw[g], smem + XW_OFF + wave * 6144 + g * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const int pairs = p.K / 128 - 1;
    char* stg = smem + 2 * STAGE + wave * 8192;
    for (int ti = 0; ti < my_tiles; ++ti) {
        if (ti + 1 < my_tiles) tile_bases(chunk_lo + slot + (ti + 1) * per_xcd, nA, nW, m0n, n0n);
        else { nA = cA; nW = cW; m0n = m0; n0n = n0; }       // nothing follows: the last prefetch re-reads this tile's first K-tile (unused)
        asm volatile(W6_LOOP_ASM
                     : : [ga0] "v"(ga[0]), [ga1] "v"(ga[1]), [ga2] "v"(ga[2]), [ga3] "v"(ga[3]), [ga4] "v"(ga[4]), [ga5] "v"(ga[5]), [ga6] "v"(ga[6]),
                         [ga7] "v"(ga[7]), [gw0] "v"(gw[0]), [gw1] "v"(gw[1]), [gw2] "v"(gw[2]), [gw3] "v"(gw[3]), [gw4] "v"(gw[4]), [gw5] "v"(gw[5]),
                         [lx00] "v"(lx[0][0]), [lx01] "v"(lx[0][1]), [lx10] "v"(lx[1][0]), [lx11] "v"(lx[1][1]),
                         [lw00] "v"(lw[0][0]), [lw01] "v"(lw[0][1]), [lw10] "v"(lw[1][0]), [lw11] "v"(lw[1][1]), [cA] "s"(cA), [cW] "s"(cW), [nA] "s"(nA),
                         [nW] "s"(nW), [pairs] "s"(pairs), [ldsw] "s"(ldsw), [ldsww] "s"(ldsww)
                     : W4_CLOBBERS);
        // ---- epilogue: six 32x64 chunks of the wave's 64x192 tile through the shared fused epilogues (wave-private LDS staging); the lane id is
        //      laundered per tile for the reason given in gemm_w4_kernel
#define W6_CHUNK(Q, C)                                                                                         \
        {                                                                                                      \
            f32x4 a2[2][4];                                                                                    \
            W6_READ_CHUNK_##Q##_##C(a2)                                                                        \
            epilogue_part<EPI, 2, true, HAS_COLB>(p, a2, m0 + wave * 64 + Q * 32, n0 + C * 64, stg, lane_e, bpre); \
        }
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        constexpr bool HAS_COLB = EPI != EPI_VT && EPI != EPI_BIAS_ROW;
#define W6_COLUMN_PART(C)                                                                                                \
        {                                                                                                            \
            BiasPre bpre;                                                                                            \
            if constexpr (HAS_COLB)                                                                                   \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                         \
                    bpre.v[j] = *(const float4*)(p.bias + n0 + C * 64 + j * 16 + 4 * (lane_e >> 4));                  \
            W6_CHUNK(0, C) W6_CHUNK(1, C)                                                                            \
        }
        W6_COLUMN_PART(0) W6_COLUMN_PART(1) W6_COLUMN_PART(2)
#undef W6_COLUMN_PART
#undef W6_CHUNK
        cA = nA; cW = nW; m0 = m0n; n0 = n0n;
    }
}

template <int EPI>
hipError_t launch_w6(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm_w6_kernel<EPI>;
    constexpr int smem = 2 * 57344 + 4 * 8192;            // operand stages + epilogue staging = 144 KiB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    int grid = (p.M / 256) * (p.N / 192);
    if (grid > 256) grid = 256;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, p);
    return hipGetLastError();
}
template <int EPI>
inline bool w6_ok(const GemmParams& p) {      // whole tiles only
    return p.M % 256 == 0 && p.N % 192 == 0 && p.K % 128 == 0 && p.K >= 128 && ((p.ldo & 7) == 0 || is_qk_epi(EPI) || EPI == EPI_VT) &&
           (EPI != EPI_VT || ((p.rows_per_sample | p.s_off) & 7) == 0);
}

template <int BM, int BN, int WM, int WN, int EPI, bool CONV = false>
hipError_t launch_cfg(const GemmParams& p, hipStream_t stream) {
    using C = Cfg<BM, BN, WM, WN>;
    auto kern = gemm_kernel<BM, BN, WM, WN, EPI, CONV>;
    constexpr int smem = 2 * C::STAGE;
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int nsplit = (EPI == EPI_F32 && p.k_split > 1) ? p.k_split : 1;
    hipLaunchKernelGGL(kern, dim3(ntm * ntn * nsplit), dim3(C::NT), smem, stream, p);
    return hipGetLastError();
}

template <int EPI>
hipError_t launch_epi(const GemmParams& p, hipStream_t stream) {
    // grid-size dispatch: 256x256 persistent ping-pong once >= g_pp_min_tiles of its tiles exist (1 workgroup per CU),
    // else 128x128 tiles with 2 workgroups per CU (a 128x256 ping-pong variant for mid-size grids measured slower than
    // this on every text-stream / small-batch shape: profiles/r01_gemm_variants.txt)
    const long big = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    if (g_gemm_variant == 0) {
        if (big >= 200 && EPI != EPI_UNPATCH) return launch_cfg<256, 256, 2, 4, EPI>(p, stream);
        return launch_cfg<128, 128, 2, 2, EPI>(p, stream);
    }
    // the ping-pong kernel addresses its operands with 32-bit byte offsets from the base pointers
    const bool fits32 = ((size_t)p.M * (size_t)p.lda + (size_t)p.K) * 2 < (1ull << 32) &&
                        ((size_t)p.N * (size_t)p.ldw + (size_t)p.K) * 2 < (1ull << 32);
    // wave quantisation: the ping-pong kernel runs ceil(tiles256 / 256 CUs) rounds of one 256x256 tile (= 4 units of 128x128 work);
    // the 128x128 kernel ceil(tiles128 / 512) rounds of two co-resident tiles at ~72 % of the ping-pong per-tile rate (measured,
    // profiles/r01_gemm_variants.txt).  E.g. the text stream at B=8 (M = 2664): q|k 132 big tiles = half the CUs idle -> 128x128.
    const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    const double cost_pp = 4.0 * (double)((big + 255) / 256), cost_128 = 2.78 * (double)((t128 + 511) / 512);
    if constexpr (EPI != EPI_UNPATCH && !is_qk_epi(EPI)) {
        // mid-size kernel (round 6): 128x192 tiles (192x128 for the transposed V^T scatter), one 4-wave workgroup per CU.  In units of one
        // 128x128 tile's work at the ping-pong kernel's per-CU rate a round of it costs 1.5; it is taken when that, times the margin
        // g_mid_alpha, beats what the dispatch below would run (a lone 128x128 tile per CU -- grids of at most 256 -- runs at ~1.7, not at
        // the co-resident pair's 2.78).  Only the GRID enters the decision, never the data: every kernel accumulates an output element in
        // the same order, so a sample's result does not depend on the kernel its batch size selects.
        constexpr bool vt = EPI == EPI_VT;
        const long tmid = (long)((p.M + (vt ? 191 : 127)) / (vt ? 192 : 128)) * ((p.N + (vt ? 127 : 191)) / (vt ? 128 : 192));
        const double cost_mid = 1.5 * (double)((tmid + 255) / 256) * g_mid_alpha;
        const double cost_else = (big >= g_pp_min_tiles && fits32 && cost_pp <= cost_128) ? cost_pp : (t128 <= 256 ? 1.7 : cost_128);
        if (g_mid_mode != 0 && (g_mid_mode == 2 || g_mid_plan_hint) && fits32 && g_gemm_variant != 0 && tmid >= g_mid_min_tiles && tmid <= g_mid_max_tiles &&
            (g_mid_mode == 2 || cost_mid < cost_else))
            return vt ? launch_mid<6, 4, EPI>(p, stream) : launch_mid<4, 6, EPI>(p, stream);
    }
    if constexpr (EPI != EPI_UNPATCH) {   // (proj_out, N = 64: scalar-scatter epilogue, always the 128x128 kernel)
        // default dispatch, from IN-MODEL per-kernel durations (profiles/r03g_*: the rollout runs at the package power cap, where the 4-wave
        // kernel's +5 ... +10 % of the back-to-back microbenchmark shrink to -2.8 % time on the wide MLP projection, +-0 on q|k, and turn into
        // +2 ... +4 % on the N = 1536 gated-residual and V^T shapes, whose read-modify-write / scatter epilogues its single wave per SIMD exposes)
        // ... and only for grids of at least g_w4_min_tiles tiles (round 5, profiles/r05j_knob_sweep.txt): on the 192 / 384-tile grids of the
        // reference's 512^2 examples (forward batch 4: q|k, MLP projection) its single wave per SIMD has no second round to hide a tile's
        // epilogue behind -- the ping-pong kernel runs that rollout 2.5 % faster (154.5 vs 150.7 denoise-steps/s), +-0.3 % at 8192 rows
        // 256x192 tiles (round 6) where they turn a fractional last round of 256x256 tiles into whole rounds: a round of them costs 3 units of
        // 128x128 work against 4 (8192 rows x N = 1536: 192 tiles of 256x256 = 0.75 rounds -> 256 tiles of 256x192 = 1 round, 3 vs 4 units;
        // N = 3072: 1.5 -> 2 rounds, 6 vs 8).  Only the grid decides (see the mid-size kernel above); g_w6_mode: mi355_tune_set(40, v).
        if (g_w6_mode != 0 && g_gemm_variant != 0 && fits32 && w6_ok<EPI>(p)) {
            const long t192 = (long)(p.M / 256) * (p.N / 192);
            const double cost_w6 = 3.0 * (double)((t192 + 255) / 256) * g_w6_alpha;
            if (g_w6_mode == 2 || (big >= g_pp_min_tiles && cost_pp <= cost_128 && cost_w6 < cost_pp && t192 >= g_w6_min_tiles)) return launch_w6<EPI>(p, stream);
        }
        const bool w4_default = p.K <= g_w4_max_k && p.N >= 3072 && EPI != EPI_VT && EPI != EPI_GATE_RES && big >= g_w4_min_tiles;
        if ((g_gemm_variant == 2 || (g_gemm_variant == 1 && w4_default)) && big >= g_pp_min_tiles && w4_ok<EPI>(p) && cost_pp <= cost_128) {
            if constexpr (EPI == EPI_BIAS) {      // ablation builds of the hand-scheduled loop (scripts/gemm_ab.py): 33 no loads, 34 no fragment reads, 35 MFMA only
                switch (p.dbg_skip_prefetch) {
                    case 33: return launch_w4<EPI, 1>(p, stream);
                    case 34: return launch_w4<EPI, 2>(p, stream);
                    case 35: return launch_w4<EPI, 3>(p, stream);
                    case 36: return launch_w4<EPI, 4>(p, stream);      // schedule s1 (valid results)
                    case 37: return launch_w4<EPI, 5>(p, stream);      // schedule s3 (valid results)
                    default: break;
                }
            }
            return launch_w4<EPI>(p, stream);
        }
        if (big >= g_pp_min_tiles && fits32 && cost_pp <= cost_128) {
            if constexpr (EPI == EPI_BIAS) {      // ablation builds (scripts/gemm_ablate.py)
                switch (p.dbg_skip_prefetch) {
                    case 0: break;
                    case 1: return launch_pp<EPI, 1>(p, stream);
                    case 2: return launch_pp<EPI, 2>(p, stream);
                    case 3: return launch_pp<EPI, 3>(p, stream);
                    case 4: return launch_pp<EPI, 4>(p, stream);
                    case 7: return launch_pp<EPI, 7>(p, stream);
                    case 8: return launch_pp<EPI, 8>(p, stream);
                    case 16: return launch_pp<EPI, 16>(p, stream);      // s_memtime / s_memrealtime trace build
                    default: return hipErrorInvalidValue;
                }
            }
            return launch_pp<EPI>(p, stream);
        }
        if (big >= 200 && !fits32) return launch_cfg<256, 256, 2, 4, EPI>(p, stream);    // > 4 GiB operand (FLUX modulation table)
    }
    return launch_cfg<128, 128, 2, 2, EPI>(p, stream);
}

// VAE decode shapes (simple 2-stage kernel; CONV = implicit 3x3 gather on the A operand)
int g_conv_cfg = 0;   // A/B knob: 0 auto, 1 force 128x128, 2 force 256x128, 3 force 256x256, 4 force 512x128
template <int EPI, bool CONV>
hipError_t launch_simple(const GemmParams& p, hipStream_t stream) {
    int cfg = g_conv_cfg;
    if (cfg == 0) {
        const long t256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
        // N = 128 (the full-resolution convolutions): a 512x128 tile gives every wave the 128x64 sub-tile of the 256x256
        // kernel (LDS bytes per MFMA halve against 256x128's 64x64 sub-tiles, which are LDS-read bound)
        cfg = (p.N >= 256 && t256 >= 128) ? 3 : (p.N >= 128 && (long)((p.M + 511) / 512) * ((p.N + 127) / 128) >= 256) ? 4 : 1;
    }
    if (cfg == 4) return launch_cfg<512, 128, 4, 2, EPI, CONV>(p, stream);
    if (cfg == 3) return launch_cfg<256, 256, 2, 4, EPI, CONV>(p, stream);
    if (cfg == 2) return launch_cfg<256, 128, 4, 2, EPI, CONV>(p, stream);
    return launch_cfg<128, 128, 2, 2, EPI, CONV>(p, stream);
}

}  // namespace

void set_conv_cfg(int v) { g_conv_cfg = v; }
void set_gemm_variant(int v) { g_gemm_variant = v; }
void set_w4_max_k(int v) { g_w4_max_k = v; }
void set_pp_min_tiles(int v) { g_pp_min_tiles = v; }
void set_w4_min_tiles(int v) { g_w4_min_tiles = v; }
void set_w6_mode(int v) { g_w6_mode = v; }
void set_w6_alpha_percent(int v) { g_w6_alpha = v / 100.0; }
void set_w6_min_tiles(int v) { g_w6_min_tiles = v; }
void set_mid_mode(int v) { g_mid_mode = v; }
void set_mid_alpha_percent(int v) { g_mid_alpha = v / 100.0; }
void set_mid_min_tiles(int v) { g_mid_min_tiles = v; }
void set_mid_max_tiles(int v) { g_mid_max_tiles = v; }
void set_mid_plan_hint(int v) { g_mid_plan_hint = v; }
int get_gemm_variant() { return g_gemm_variant; }

void set_raster_gm(int v) { g_raster_gm = v; }

// regions of one GEMM launch for the schedule trace (sched_trace.hip)
static void trace_gemm(const GemmParams& p, hipStream_t stream) {
    const size_t a_bytes = p.conv_cin > 0 ? 0 : ((size_t)(p.M - 1) * p.lda + p.K) * 2, w_bytes = ((size_t)(p.N - 1) * p.ldw + p.K) * 2;
    const size_t out_bytes = p.out ? ((size_t)(p.M - 1) * p.ldo + p.N) * 2 : 0;
    const TraceRegion A = treg(p.A, a_bytes), W = treg(p.W, w_bytes), none = treg(nullptr, 0);
    const bool rowb = p.epi == EPI_VT || p.epi == EPI_BIAS_ROW;
    const TraceRegion bias = treg(p.bias, (size_t)(rowb ? p.M : p.N) * 4);
    const TraceRegion stash = treg(p.stash, p.stash ? ((size_t)(p.M - 1) * p.ld_stash + p.N) * 2 : 0);
    const int rps = p.rows_per_sample > 0 ? p.rows_per_sample : 1;
    switch (p.epi) {
        case EPI_GATE_RES: {
            const TraceRegion gate = treg(p.aux, ((size_t)((p.M + rps - 1) / rps - 1) * p.ld_aux + p.N) * 2), out = treg(p.out, out_bytes);
            sched_trace_launch("gemm.gate_res", stream, {A, W, bias, gate, out}, {out, stash});
            break;
        }
        case EPI_POSADD: case EPI_ADDSRC_SILU: case EPI_DGELU: {
            const size_t rows = p.epi == EPI_DGELU ? (size_t)p.M : (size_t)(rps < p.M ? rps : p.M);
            sched_trace_launch("gemm.aux", stream, {A, W, bias, treg(p.aux, ((rows - 1) * p.ld_aux + p.N) * 2)}, {treg(p.out, out_bytes)});
            break;
        }
        case EPI_QK_NORM: case EPI_QK_NORM_RSTD: {
            // rows of sample b, head h land at q / k [(b*H + h)*S_pad + s_off + s][64]: B*H blocks of rps rows, S_pad rows apart
            const size_t blocks = (size_t)(p.M / rps) * p.H, len = (size_t)rps * 128, stride = (size_t)p.S_pad * 128;
            sched_trace_launch("gemm.qk_norm", stream, {A, W, bias, treg(p.nw_q, 256), treg(p.nw_k, 256)},
                               {tregs(p.q + (size_t)p.s_off * 64, len, stride, blocks), tregs(p.k + (size_t)p.s_off * 64, len, stride, blocks),
                                treg(p.rstd_out, p.rstd_out ? (size_t)p.M * 2 * p.H * 4 : 0)});
            break;
        }
        case EPI_VT: {
            // feature m = (h, d), token n: vT [((b*H + h)*hd + d)*S_pad + s_off + s]: (N/rps)*M blocks of rps tokens, S_pad tokens apart
            const size_t blocks = (size_t)(p.N / rps) * p.M;
            sched_trace_launch("gemm.vT", stream, {A, W, bias}, {tregs(p.q + p.s_off, (size_t)rps * 2, (size_t)p.S_pad * 2, blocks)});
            break;
        }
        case EPI_UNPATCH:
            sched_trace_launch("gemm.unpatch", stream, {A, W, bias},
                               {treg(p.out, (size_t)(p.M / (p.hp * p.wp)) * p.out_ch * p.hp * p.patch * p.wp * p.patch * 2)});
            break;
        case EPI_F32:
            sched_trace_launch("gemm.f32", stream, {A, W}, {treg(p.out_f32, ((size_t)(p.M - 1) * p.ldo + p.N) * 4 + (size_t)(p.k_split > 1 ? p.k_split - 1 : 0) * p.split_stride * 4)});
            break;
        default:
            sched_trace_launch("gemm", stream, {A, W, bias}, {treg(p.out, out_bytes), stash, none});
    }
}

hipError_t launch_gemm(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    p.raster_gm = g_raster_gm;
    if (sched_trace_on()) trace_gemm(p, stream);
    if (p.K % BK != 0 || p.M <= 0 || p.N <= 0) return hipErrorInvalidValue;
    if (p.conv_cin > 0) {
        const int ks = p.conv_ks ? p.conv_ks : 3, kt3 = p.conv_kt ? p.conv_kt : 1;
        if (p.conv_cin % BK != 0 || p.K != kt3 * ks * ks * p.conv_cin || !p.zero_page || p.M % (p.conv_h * p.conv_w) != 0 ||
            (p.conv_up && ((p.conv_h | p.conv_w) & 1)) || (ks != 1 && ks != 3) || (kt3 != 1 && kt3 != 3) ||
            (kt3 > 1 && (p.conv_t < 1 || p.conv_t_in < p.conv_t)) || (p.conv_t > 0 && (p.M / (p.conv_h * p.conv_w)) % p.conv_t != 0))
            return hipErrorInvalidValue;
        switch (p.epi) {
            case EPI_BIAS: return launch_simple<EPI_BIAS, true>(p, stream);
            case EPI_POSADD: return launch_simple<EPI_POSADD, true>(p, stream);
            case EPI_IMG: return p.N <= 4 ? launch_cfg<128, 128, 2, 2, EPI_IMG, true>(p, stream) : hipErrorInvalidValue;
            default: return hipErrorInvalidValue;
        }
    }
    switch (p.epi) {
        case EPI_BIAS: return launch_epi<EPI_BIAS>(p, stream);
        case EPI_BIAS_SILU: return launch_epi<EPI_BIAS_SILU>(p, stream);
        case EPI_BIAS_GELU: return launch_epi<EPI_BIAS_GELU>(p, stream);
        case EPI_POSADD: return launch_epi<EPI_POSADD>(p, stream);
        case EPI_ADDSRC_SILU: return launch_epi<EPI_ADDSRC_SILU>(p, stream);
        case EPI_GATE_RES: return launch_epi<EPI_GATE_RES>(p, stream);
        case EPI_QK_NORM: return p.rstd_out ? launch_epi<EPI_QK_NORM_RSTD>(p, stream) : launch_epi<EPI_QK_NORM>(p, stream);
        case EPI_VT: return launch_epi<EPI_VT>(p, stream);
        case EPI_UNPATCH: return launch_epi<EPI_UNPATCH>(p, stream);
        case EPI_DGELU: return launch_epi<EPI_DGELU>(p, stream);
        case EPI_BIAS_ROW: return launch_simple<EPI_BIAS_ROW, false>(p, stream);
        case EPI_F32: return launch_simple<EPI_F32, false>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mi355
