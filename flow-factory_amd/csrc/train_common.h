// mi355_flow -- helpers shared by the native backward passes of the head_dim-128 engines (flux_train.inc, qwen_train.inc): the dgrad / wgrad
// GEMM wrappers, the attention-backward and q | k producer backward sequences, and the scratch they work in.  (The SD3.5 engine's training
// file, engine_train.inc, predates this header and keeps its own copies: head_dim 64, per-block dual attention, side-stream scratch pairs.)
#pragma once
#include "engine_common.h"

namespace mi355 {

// backward scratch of one plan (device pointers into the plan's training workspace).  Row counts: M = every token row of the forward batch,
// Mc = its context rows; wide / widec hold [rows][3D + F] (the fused q|k|v|mlp gradient of a single-stream block, or [rows][3D] + [rows][F]).
struct TrainScratch {
    bf16_t *g1, *g1c, *g2, *g2c, *g3, *g3c, *wide, *widec, *hid;
    bf16_t *doh, *v, *dq, *dk, *dvh;          // attention backward: dO head-major, V row-major, dq~ / dk / dv head-major
    float *delta, *nld;
    bf16_t *aT, *xT;                          // wgrad operands, contraction-contiguous: [N_max][M_pad], [K_max][M_pad]
    float *part, *csum, *zero_bias;
    size_t part_floats;
};

struct AttnGeom {
    int B, H, D, S, S_pad;                    // forward batch, heads, H * 128, joint sequence length (padded to 64)
    const float2* cs;                         // rotary table [S][64]
    const int* kv_len;                        // optional device [B]: valid keys per sample (ragged text at the END of the joint sequence)
};

// out[M][N] (row stride ldo) = A[M][K] (row stride lda) . WT[N][K]^T  (WT = transposed weight: a dgrad), optional fused gelu'(pre)
static inline int t_dgrad(const TrainScratch& t, hipStream_t st, const bf16_t* A, long lda, int K, const bf16_t* WT, int M, int N, bf16_t* out, long ldo,
                          const bf16_t* gelu_pre, long ld_pre) {
    GemmParams g = make_gemm(A, lda, WT, K, M, N, K, gelu_pre ? EPI_DGELU : EPI_BIAS, t.zero_bias, out, ldo);
    if (gelu_pre) { g.aux = gelu_pre; g.ld_aux = ld_pre; }
    HIPCHK(launch_gemm(g, st));
    return 0;
}

// dW[N][K] (fp32, overwritten) = dY[:, col0 : col0 + N]^T . X;  db[N] = column sums.  x_ready: t.xT already holds X^T.
// Operands are transposed to contraction-contiguous form (64 x 64 LDS tile transposes, rows zero-padded to M_pad), the GEMM is split-K over the
// token dimension into partial buffers + a fixed-order reduction (deterministic); the bias gradient is taken by the dY transpose.
static inline int t_wgrad(const TrainScratch& t, hipStream_t st, const bf16_t* dY, long ldY, int col0, int N, const bf16_t* X, long ldX, int K, int M,
                          int M_pad, float* gw, float* gb, bool x_ready) {
    if (gb && !gw) HIPCHK(launch_colsum(dY + col0, ldY, M, N, t.csum, gb, 0, st));
    if (!gw) return 0;
    if (!x_ready) HIPCHK(launch_transpose(X, ldX, 0, t.xT, M_pad, 0, M, K, M_pad, 1, st));
    if (gb) HIPCHK(launch_transpose_colsum(dY + col0, ldY, t.aT, M_pad, M, N, M_pad, t.csum, gb, st));
    else HIPCHK(launch_transpose(dY + col0, ldY, 0, t.aT, M_pad, 0, M, N, M_pad, 1, st));
    const long tiles = (long)((N + 127) / 128) * ((K + 127) / 128);
    int split = (int)((768 + tiles - 1) / tiles);
    const int nt = M_pad / 64;
    if (split > nt / 2) split = nt / 2;
    if (split > 16) split = 16;
    if (split < 1) split = 1;
    while (split > 1 && (size_t)split * N * K > t.part_floats) --split;
    GemmParams g = make_gemm(t.aT, M_pad, t.xT, M_pad, N, K, M_pad, EPI_F32, nullptr, nullptr, K);
    g.q_scale = 1.0f;
    if (split <= 1 && grad_buf_dtype(gw) == DT_F32) {
        g.out_f32 = gw; g.k_split = 1;
        HIPCHK(launch_gemm(g, st));
        return 0;
    }
    if ((size_t)N * K > t.part_floats) return errorf("wgrad: a %d x %d gradient does not fit the split-K scratch", N, K);   // (bf16 output, single split)
    g.out_f32 = t.part; g.k_split = split; g.split_stride = (long)N * K;
    HIPCHK(launch_gemm(g, st));
    HIPCHK(launch_splitk_reduce(t.part, (long)N * K, split, gw, (long)N * K, 0, st));
    return 0;
}

// flash-attention backward of one head_dim-128 attention: o / dO token-major (row stride D), the first n_first positions of a sample in
// (o_first, do_first), the rest in (o_rest, do_rest); a null dO = that part of the output has no consumer.  Results: head-major dq~, dk, dv
// in t.dq / t.dk / t.dvh.
static inline int t_attention128_backward(const TrainScratch& t, const AttnGeom& a, hipStream_t st, const bf16_t* q, const bf16_t* k, const bf16_t* vT,
                                          const float* lse, const bf16_t* o_first, const bf16_t* o_rest, const bf16_t* do_first, const bf16_t* do_rest,
                                          int n_first) {
    Attn128BwdPrepParams pp;
    memset(&pp, 0, sizeof(pp));
    pp.o_first = o_first; pp.ld_o_first = a.D; pp.o_rest = o_rest; pp.ld_o_rest = a.D;
    pp.do_first = do_first; pp.ld_do_first = a.D; pp.do_rest = do_rest; pp.ld_do_rest = a.D; pp.n_first = n_first;
    pp.lse = lse; pp.doh = t.doh; pp.delta = t.delta; pp.nld = t.nld; pp.B = a.B; pp.H = a.H; pp.S = a.S; pp.S_pad = a.S_pad;
    HIPCHK(launch_attn128_bwd_prep(pp, st));
    const long hs = (long)a.S_pad * 128;
    HIPCHK(launch_transpose(vT, a.S_pad, hs, t.v, 128, hs, 128, a.S_pad, 128, a.B * a.H, st));      // V^T [128][S_pad] -> V [S_pad][128] per (b, h)
    AttnBwdParams ab{q, k, t.v, t.doh, lse, t.delta, t.nld, t.dq, t.dk, t.dvh, a.B, a.H, a.S, a.S_pad, a.kv_len};
    HIPCHK(launch_attention128_bwd(ab, st));
    return 0;
}

// backward of the q | k producer (per-head RMSNorm + RoPE) of ONE stream's M rows (rps rows per sample, joint positions s_off ..) + gather of
// (dq~, dk, dv) to token-major [dq_pre | dk_pre | dv] rows (row stride ld_out)
static inline int t_rope_back(const TrainScratch& t, const AttnGeom& a, hipStream_t st, const bf16_t* q, const bf16_t* k, const float* rstd,
                              const float* nq, const float* nk, bf16_t* out, long ld_out, int M, int rps, int s_off) {
    RopeRmsBwdParams r;
    memset(&r, 0, sizeof(r));
    r.q = q; r.k = k; r.dq = t.dq; r.dk = t.dk; r.dv = t.dvh; r.rstd = rstd; r.nw_q = nq; r.nw_k = nk; r.cs = a.cs;
    r.q_scale = 0.08838834764831845f * 1.4426950408889634f;
    r.out = out; r.ld_out = ld_out; r.M = M; r.H = a.H; r.rows_per_sample = rps; r.s_off = s_off; r.S_pad = a.S_pad;
    HIPCHK(launch_rope_rms_bwd128(r, st));
    return 0;
}

}  // namespace mi355
