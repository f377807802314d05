// mi355_flow -- helpers shared by the native backward passes of the head_dim-128 engines (flux_train.inc, qwen_train.inc): the dgrad / wgrad
// GEMM wrappers, the attention-backward and q | k producer backward sequences, and the scratch they work in.  (The SD3.5 engine's training
// file, engine_train.inc, predates this header and keeps its own copies: head_dim 64, per-block dual attention, side-stream scratch pairs.)
#pragma once
#include "engine_common.h"

namespace mi355 {

// One in-flight weight-gradient group: its operands in contraction-contiguous form (aT: the dY columns, [rows <= aT_rows][M_pad]; xT: the
// layer input, [rows <= xT_rows][M_pad]) + the events that hand it to the side stream and back.
struct WgradSlot {
    bf16_t *aT = nullptr, *xT = nullptr;
    hipEvent_t ready = nullptr, done = nullptr;
    bool busy = false;
};
constexpr int WG_SLOTS = 3;

// backward scratch of one plan (device pointers into the plan's training workspace).  Row counts: M = every token row of the forward batch,
// Mc = its context rows; wide / widec hold [rows][3D + F] (the fused q|k|v|mlp gradient of a single-stream block, or [rows][3D] + [rows][F]).
struct TrainScratch {
    bf16_t *g1, *g1c, *g2, *g2c, *g3, *g3c, *wide, *widec, *hid;
    bf16_t *doh, *v, *dq, *dk, *dvh;          // attention backward: dO head-major, V row-major, dq~ / dk / dv head-major
    float *delta, *nld;
    float *part, *csum, *zero_bias;
    size_t part_floats;
    bf16_t* hid_c = nullptr;                  // the text chain's own gelu(pre) / column-sum scratch when it runs on a second stream
    float* csum_c = nullptr;
    // weight gradients: the operand transposes (+ the bias column sums) run on the backward's own stream, the split-K GEMMs and their
    // reductions on `wside` when it exists (mi355_tune_set(26, .)): they are off the critical path -- nothing in the backward reads a weight
    // gradient -- and their partial last rounds (144 output tiles x 2 splits on 256 CUs) fill with the dgrad chain's kernels and vice versa
    WgradSlot slot[WG_SLOTS];
    int n_slots = 1, next_slot = 0;
    long aT_rows = 0, xT_rows = 0;
    hipStream_t wside = nullptr;
    bool tn_default = false;                  // weight gradients on row-major operands by default (set per engine from measurements: Wan yes, FLUX.1 / Qwen-Image no)
};

struct AttnGeom {
    int B, H, D, S, S_pad;                    // forward batch, heads, H * 128, joint sequence length (padded to 64)
    const float2* cs;                         // rotary table [S][64]
    const int* kv_len;                        // optional device [B]: valid keys per sample (ragged text at the END of the joint sequence)
    int S_kv = 0, S_kv_pad = 0;               // cross-attention: keys / values are another sequence ([B][H][S_kv_pad][128]); 0 = self
};

// out[M][N] (row stride ldo) = A[M][K] (row stride lda) . WT[N][K]^T  (WT = transposed weight: a dgrad), optional fused gelu'(pre)
static inline int t_dgrad(const TrainScratch& t, hipStream_t st, const bf16_t* A, long lda, int K, const bf16_t* WT, int M, int N, bf16_t* out, long ldo,
                          const bf16_t* gelu_pre, long ld_pre) {
    GemmParams g = make_gemm(A, lda, WT, K, M, N, K, gelu_pre ? EPI_DGELU : EPI_BIAS, t.zero_bias, out, ldo);
    if (gelu_pre) { g.aux = gelu_pre; g.ld_aux = ld_pre; }
    HIPCHK(launch_gemm(g, st));
    return 0;
}

// ---- weight gradients -------------------------------------------------------------------------------------------------------------
struct WgradSeg { int col0, N; float* gw; float* gb; };      // dW[N][K] = dY[:, col0 : col0 + N]^T . X (overwritten), db[N] = column sums
struct WgradX { const bf16_t* p; long ld; int K; };          // X = [X0 | X1 ...] column blocks (a single block for every layer but FLUX.1's proj_out)

static inline int t_wgrad_streams_init(TrainScratch& t, bool side) {
    t.n_slots = 1;
    if (!side) return 0;
    HIPCHK(hipStreamCreateWithFlags(&t.wside, hipStreamNonBlocking));
    for (int i = 0; i < WG_SLOTS; ++i) {
        HIPCHK(hipEventCreateWithFlags(&t.slot[i].ready, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&t.slot[i].done, hipEventDisableTiming));
    }
    t.n_slots = WG_SLOTS;
    return 0;
}
static inline void t_wgrad_streams_destroy(TrainScratch& t) {
    for (int i = 0; i < WG_SLOTS; ++i) {
        if (t.slot[i].ready) (void)hipEventDestroy(t.slot[i].ready);
        if (t.slot[i].done) (void)hipEventDestroy(t.slot[i].done);
    }
    if (t.wside) (void)hipStreamDestroy(t.wside);
    t.wside = nullptr;
}
// every gradient written so far is complete on `st` (end of the backward)
static inline int t_wgrad_join(TrainScratch& t, hipStream_t st) {
    for (int i = 0; i < t.n_slots; ++i)
        if (t.slot[i].busy) { HIPCHK(ev_wait(st, t.slot[i].done)); t.slot[i].busy = false; }
    return 0;
}

// One GEMM of a group on stream `ws`: dW = aT[N][M_pad] . xT[K][M_pad]^T, split-K over the token dimension into partial buffers + a fixed-order
// reduction (deterministic; the reduction rounds to bf16 itself for a bf16 gradient buffer)
static inline int t_wgrad_gemm(const TrainScratch& t, hipStream_t ws, const bf16_t* aT, const bf16_t* xT, int N, int K, int M_pad, float* gw) {
    const int split = wgrad_split(N, K, M_pad, t.part_floats, ws == t.wside && t.wside != nullptr);
    GemmParams g = make_gemm(aT, M_pad, xT, M_pad, N, K, M_pad, EPI_F32, nullptr, nullptr, K);
    g.q_scale = 1.0f;
    if (split <= 1 && grad_buf_dtype(gw) == DT_F32) {
        g.out_f32 = gw; g.k_split = 1;
        HIPCHK(launch_gemm(g, ws));
        return 0;
    }
    if ((size_t)N * K > t.part_floats) return errorf("wgrad: a %d x %d gradient does not fit the split-K scratch", N, K);   // (bf16 output, single split)
    g.out_f32 = t.part; g.k_split = split; g.split_stride = (long)N * K;
    HIPCHK(launch_gemm(g, ws));
    HIPCHK(launch_splitk_reduce(t.part, (long)N * K, split, gw, (long)N * K, 0, ws));
    return 0;
}

// The weight (and bias) gradients of the linear layers that share the input X: dY [M][..] holds their output gradients side by side.
// On `st`: X^T once, the dY column blocks transposed (rows zero-padded to M_pad; the bias gradient = the column sums taken on the way).
// On the side stream (or `st`): one split-K GEMM + reduction per layer.  A group larger than a slot's aT is cut (X^T is taken again).
static inline int t_wgrad_group(TrainScratch& t, hipStream_t st, const bf16_t* dY, long ldY, const WgradSeg* seg, int nseg, const WgradX* xs, int nx,
                                int M, int M_pad, bool ctx_chain = false) {
    float* csum = ctx_chain && t.csum_c ? t.csum_c : t.csum;          // (column-sum partials: one scratch per stream that takes them)
    int K = 0;
    for (int i = 0; i < nx; ++i) K += xs[i].K;
    // Round 6: operands as they lie in HBM (csrc/gemm_tn.hip: dY [M][ldY], X [M][ldX] row-major, MFMA fragments through transposed LDS reads;
    // a ragged last m-tile reads zeros) -- neither dY^T nor X^T is made, and the 256 x 256-tile form runs the large weights (N x K >= 160 tiles)
    // at the ping-pong kernel's bytes per FLOP instead of the 2-stage kernel's.  Same products, same split boundaries over M_pad, same fixed-order
    // reduction.  These GEMMs read dY itself, which the backward overwrites next: they run on `st`, not on the side stream that the copies made
    // safe.  Measured per engine (profiles/r06w_*, optimize() step, row-major vs copies): Wan2.1 at 20 280 tokens 1033 vs 1053 ms (its 1536-wide
    // weights need 7-way split-K on the copies' path and its ragged M a padded transpose of 40 560 rows) -- ON; FLUX.1 251 vs 239 ms and Qwen-Image
    // 472 vs 468 ms (3072-wide weights: the 2-stage 256 x 256 kernel beside the dgrad chain on the side stream is the better schedule) -- OFF.
    // mi355_tune_set(39, 2) = row-major wherever it applies, 0 = never, 1 (default) = the engine's measured default.
    if ((get_wgrad_tn_mode() == 2 || (get_wgrad_tn_mode() == 1 && t.tn_default)) && nx == 1 && M_pad % 64 == 0) {
        bool ok = true;
        for (int i = 0; i < nseg && ok; ++i)
            if (seg[i].gw) {
                GemmTnParams tp{dY + seg[i].col0, ldY, xs[0].p, xs[0].ld, M, seg[i].N, K, t.part, (long)K, 1, (long)seg[i].N * K, nullptr, 0};
                tp.M_pad = M_pad;
                ok = gemm_tn_ok(tp) && (size_t)seg[i].N * K <= t.part_floats;
            }
        if (ok) {
            CHK(t_wgrad_join(t, st));                  // earlier side-stream GEMMs may still be reading / writing t.part
            for (int i = 0; i < nseg; ++i) {
                const int N = seg[i].N;
                if (seg[i].gb) HIPCHK(launch_colsum(dY + seg[i].col0, ldY, M, N, csum, seg[i].gb, 0, st));
                if (!seg[i].gw) continue;
                int split = wgrad_split(N, K, M_pad, t.part_floats, false);
                int tile256 = 0;
                if (N % 256 == 0 && K % 256 == 0) {
                    const long t256 = (long)(N / 256) * (K / 256);
                    int s256 = (int)(256 / t256 > 0 ? 256 / t256 : 1);
                    const int cap = (M_pad / 64) / 2 > 0 ? (M_pad / 64) / 2 : 1;
                    if (s256 > cap) s256 = cap;
                    while (s256 > 1 && (size_t)s256 * N * K > t.part_floats) --s256;
                    if (t256 * s256 >= 160) { tile256 = 1; split = s256; }
                }
                if (split < 1) split = 1;
                const bool direct = split == 1 && grad_buf_dtype(seg[i].gw) == DT_F32;
                GemmTnParams tp{dY + seg[i].col0, ldY, xs[0].p, xs[0].ld, M, N, K, direct ? seg[i].gw : t.part, (long)K, split, (long)N * K, nullptr, tile256};
                tp.M_pad = M_pad;
                HIPCHK(launch_gemm_tn(tp, st));
                if (!direct) HIPCHK(launch_splitk_reduce(t.part, (long)N * K, split, seg[i].gw, (long)N * K, 0, st));
            }
            return 0;
        }
    }
    if (K > t.xT_rows) return errorf("wgrad: K = %d exceeds the transposed-input scratch (%ld rows)", K, t.xT_rows);
    int i = 0;
    while (i < nseg) {
        if (!seg[i].gw) {                              // bias only (or nothing): no GEMM
            if (seg[i].gb) HIPCHK(launch_colsum(dY + seg[i].col0, ldY, M, seg[i].N, csum, seg[i].gb, 0, st));
            ++i;
            continue;
        }
        WgradSlot& s = t.slot[t.next_slot];
        t.next_slot = (t.next_slot + 1) % t.n_slots;
        if (s.busy) { HIPCHK(ev_wait(st, s.done)); s.busy = false; }       // the slot's previous GEMMs have read aT / xT
        int r0 = 0;
        for (int x = 0; x < nx; ++x) {
            HIPCHK(launch_transpose(xs[x].p, xs[x].ld, 0, s.xT + (size_t)r0 * M_pad, M_pad, 0, M, xs[x].K, M_pad, 1, st));
            r0 += xs[x].K;
        }
        const int first = i;
        long rows = 0;
        for (; i < nseg && seg[i].gw && rows + seg[i].N <= t.aT_rows; ++i) {
            bf16_t* aT = s.aT + (size_t)rows * M_pad;
            if (seg[i].gb) HIPCHK(launch_transpose_colsum(dY + seg[i].col0, ldY, aT, M_pad, M, seg[i].N, M_pad, csum, seg[i].gb, st));
            else HIPCHK(launch_transpose(dY + seg[i].col0, ldY, 0, aT, M_pad, 0, M, seg[i].N, M_pad, 1, st));
            rows += seg[i].N;
        }
        if (i == first) return errorf("wgrad: N = %d exceeds the transposed-gradient scratch (%ld rows)", seg[i].N, t.aT_rows);
        hipStream_t ws = st;
        if (t.wside) {
            HIPCHK(ev_record(s.ready, st));
            HIPCHK(ev_wait(t.wside, s.ready));
            ws = t.wside;
        }
        rows = 0;
        for (int j = first; j < i; ++j) {
            CHK(t_wgrad_gemm(t, ws, s.aT + (size_t)rows * M_pad, s.xT, seg[j].N, K, M_pad, seg[j].gw));
            rows += seg[j].N;
        }
        if (t.wside) { HIPCHK(ev_record(s.done, t.wside)); s.busy = true; }
    }
    return 0;
}

// single layer
static inline int t_wgrad(TrainScratch& t, hipStream_t st, const bf16_t* dY, long ldY, int col0, int N, const bf16_t* X, long ldX, int K, int M,
                          int M_pad, float* gw, float* gb, bool ctx_chain = false) {
    if (!gw && !gb) return 0;
    const WgradSeg seg{col0, N, gw, gb};
    const WgradX x{X, ldX, K};
    return t_wgrad_group(t, st, dY, ldY, &seg, 1, &x, 1, M, M_pad, ctx_chain);
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
    const int Skp = a.S_kv > 0 ? a.S_kv_pad : a.S_pad;
    const long hs = (long)Skp * 128;
    HIPCHK(launch_transpose(vT, Skp, hs, t.v, 128, hs, 128, Skp, 128, a.B * a.H, st));      // V^T [128][S_pad] -> V [S_pad][128] per (b, h)
    AttnBwdParams ab{q, k, t.v, t.doh, lse, t.delta, t.nld, t.dq, t.dk, t.dvh, a.B, a.H, a.S, a.S_pad, a.kv_len, a.S_kv, a.S_kv_pad};
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
