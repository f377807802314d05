// mi355_flow -- flash-attention BACKWARD, head_dim 128, non-causal, for gfx950: the gradient of the FLUX.1 joint (double-stream) and
// single-stream attention for the native `optimize()` replay (SURVEY.md 8(f) N1 over N3; reference
// src/flow_factory/trainers/grpo.py:263-330 over models/flux/flux1.py:294-346).
//
// The head_dim-64 construction of attention_bwd.hip carried to 128 the way attention128.hip carries the forward: a [64 rows][128 d]
// operand tile is kept as TWO [64][64] sub-tiles (d halves), so every LDS row stays 128 bytes and the swizzle, the fragment offsets and the
// lane mapping of the transposed reads (`ds_read_b64_tr_b16`, probed in scripts/mb/tr_b16_probe.hip) carry over unchanged; the first
// products run 8 k-steps instead of 4, the second products write 4 d-blocks instead of 2.  With the stored q~ = q * log2(e) / sqrt(128):
//       P = 2^(s - L)            dP = dO . V^T            dZ = P o (dP - Delta),  Delta_i = sum_d dO_id O_id
//       dV = P^T dO              dK = ln2 * dZ^T q~       dq~ = ln2 * dZ k
// Two deterministic passes (no fp32 atomics on dQ):
//   attn128_bwd_dkv_kernel : one KEY per lane, loops over 64-query tiles (Q | dO | -L | -Delta per stage); dK^T, dV^T (8 accumulator
//                            blocks = 128 registers) live in AGPRs for the whole loop;
//   attn128_bwd_dq_kernel  : one QUERY per lane, loops over 64-key tiles (K | V per stage); dQ^T (4 blocks) in AGPRs.
// One workgroup of 4 waves per CU (one wave per SIMD, the 512-register budget), 4-stage LDS-DMA rings with counted waits.  With a single
// wave per SIMD nothing hides a stall, so the loop is ordered by hand: the first products of BOTH 32-row halves are issued back to back
// (their loop-invariant B fragments in AGPRs, the A fragments of k-step j + 1 read under k-step j's MFMAs), then each half's exp / pack
// arithmetic runs while the matrix pipe works on the other half's products (round 4: 745 -> see profiles/r04d_* us per dK / dV launch).
// Also here: the prep kernel (dO head-major, Delta, -L | -Delta per tile) for token-major o / dO with leading dimensions (the single-stream
// blocks keep the attention output inside the [M][5D] concat buffer), and the backward of the q | k producer (per-head RMSNorm + RoPE,
// flux_ops.hip rope_norm_kernel) gathered to token-major [dq_pre | dk_pre | dv] rows.
#include "kernels.h"

namespace mi355 {
namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int HD = 128;
constexpr int TB = 64;                     // rows of the streamed dimension per tile
constexpr int SUB = TB * 64 * 2;           // one [64][64] bf16 sub-tile: 8 KiB
constexpr int TILE = 2 * SUB;              // one [64][128] operand tile: 16 KiB
constexpr int NWAVES = 4;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ int row_perm(int i) {     // MFMA output row i (0..31) -> row offset inside the 32-row block (attention.hip key_perm)
    const int a = i >> 3, g = (i >> 2) & 1, b = i & 3;
    return 16 * (a >> 1) + 8 * g + 4 * (a & 1) + b;
}
__device__ __forceinline__ int swz2(int row) {       // attention_bwd.hip: conflict-free for ds_read_b128 AND for the transposed 4-row reads
    const int f = (row >> 1) & 7;
    return f ^ ((f & 1) << 2);
}
__device__ __forceinline__ bf16x8 frag4(unsigned a, unsigned b, unsigned c, unsigned d) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const u32x4 u = {a, b, c, d};
    return __builtin_bit_cast(bf16x8, u);
}

// one [64 rows][64 d] sub-tile of a row-major [.][128] operand -> LDS (8 KiB, 128-byte rows, chunks XOR-swizzled by swz2(row))
__device__ __forceinline__ void stage_sub(const bf16_t* src, char* dst, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 8 / NWAVES; ++i) {
        const int grp = wave + i * NWAVES;
        const int row = grp * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz2(row);
        __builtin_amdgcn_global_load_lds((gptr_t)(src + (long)row * HD + c * 8), (lptr_t)(dst + grp * 1024), 16, 0, 0);
    }
}

// `ahead` tiles were issued after the one about to be read, PER VM operations each (vmcnt retires in order), then the workgroup barrier
template <int PER>
__device__ __forceinline__ void wait_tiles_ahead(int ahead) {
    static_assert(PER == 8 || PER == 9, "per-tile VM operation count");
    if (PER == 9) {
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// accumulators acc[db][r] = X^T[d = 32 db + 8 (r>>2) + 4 lg + (r&3)][column = lane & 31] -> rows dst[(row0 + column)][0..128) bf16, scaled
// (wave-private 8 KiB of LDS: 32 rows x 256 B, 16-byte chunks XOR-swizzled by row & 7 -- attention128.hip's output staging)
__device__ __forceinline__ void store_rows128(const f32x16 (&acc)[4], float scale, char* ob, bf16_t* dst, int row0, int row_limit, int lane) {
    const int lq = lane & 31, lg = lane >> 5;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int d0 = 32 * db + 8 * a + 4 * lg;
            uint2 w = {pack_bf16(acc[db][4 * a] * scale, acc[db][4 * a + 1] * scale),
                       pack_bf16(acc[db][4 * a + 2] * scale, acc[db][4 * a + 3] * scale)};
            const int chunk = (d0 >> 3) ^ (lq & 7);
            *(uint2*)(ob + lq * 256 + chunk * 16 + (d0 & 7) * 2) = w;
        }
    __builtin_amdgcn_wave_barrier();      // wave-private region, in-order LDS
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int r = it * 4 + (lane >> 4), c = lane & 15;
        const uint4 val = *(const uint4*)(ob + r * 256 + ((c ^ (r & 7)) << 4));
        if (row0 + r < row_limit) *(uint4*)(dst + (long)(row0 + r) * HD + c * 8) = val;
    }
    __builtin_amdgcn_wave_barrier();
}

// The transposed reads and the MFMAs they feed: one inline-assembly block per (32-row half tile, d half), fragments in FIXED registers
// (attention_bwd.hip explains why: hipcc guards C-level LDS reads with vmcnt(0) while LDS-DMA is in flight, and a fragment is two 64-bit reads
// into the halves of one operand tuple).  Accumulators are AGPR operands ("+a").  Per-lane addresses a0..a3 = pieces (db, jj) = (0,0), (0,1),
// (1,0), (1,1) of the 16-row block; the immediates select (operand tile, d half, 16-row step).
#define TR_RD(dst, a, off) "ds_read_b64_tr_b16 " dst ", " a " offset:" #off "\n\t"
#define MFMA32(acc, afrag, b) "v_mfma_f32_32x32x16_bf16 " acc ", " afrag ", " b ", " acc "\n\t"
#define TR_CLOBBER16 "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239"
#define TR_CLOBBER32 TR_CLOBBER16, "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"

// dV^T[2 blocks] += dO^T . P, dK^T[2 blocks] += Q^T . dZ over the half tile's 32 queries (two k-steps of 16): OO0 / OQ0 = byte offsets of the
// dO / Q sub-tile (k-step 0), OO1 / OQ1 = the same + 2048 (16 rows)
#define DKV_BLOCK(OO0, OQ0, OO1, OQ1, DV0, DV1, DK0, DK1)                                                                                          \
    asm volatile(                                                                                                                                  \
        TR_RD("v[224:225]", "%[a0]", OO0) TR_RD("v[226:227]", "%[a1]", OO0) TR_RD("v[228:229]", "%[a2]", OO0) TR_RD("v[230:231]", "%[a3]", OO0)    \
        TR_RD("v[232:233]", "%[a0]", OQ0) TR_RD("v[234:235]", "%[a1]", OQ0) TR_RD("v[236:237]", "%[a2]", OQ0) TR_RD("v[238:239]", "%[a3]", OQ0)    \
        TR_RD("v[240:241]", "%[a0]", OO1) TR_RD("v[242:243]", "%[a1]", OO1) TR_RD("v[244:245]", "%[a2]", OO1) TR_RD("v[246:247]", "%[a3]", OO1)    \
        "s_waitcnt lgkmcnt(4)\n\t"                                                                                                                 \
        MFMA32("%[dv0]", "v[224:227]", "%[pf0]") MFMA32("%[dk0]", "v[232:235]", "%[zf0]")                                                          \
        TR_RD("v[248:249]", "%[a0]", OQ1) TR_RD("v[250:251]", "%[a1]", OQ1) TR_RD("v[252:253]", "%[a2]", OQ1) TR_RD("v[254:255]", "%[a3]", OQ1)    \
        MFMA32("%[dv1]", "v[228:231]", "%[pf0]") MFMA32("%[dk1]", "v[236:239]", "%[zf0]")                                                          \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                 \
        MFMA32("%[dv0]", "v[240:243]", "%[pf1]") MFMA32("%[dk0]", "v[248:251]", "%[zf1]")                                                          \
        MFMA32("%[dv1]", "v[244:247]", "%[pf1]") MFMA32("%[dk1]", "v[252:255]", "%[zf1]")                                                          \
        : [dv0] "+a"(DV0), [dv1] "+a"(DV1), [dk0] "+a"(DK0), [dk1] "+a"(DK1)                                                                       \
        : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [pf0] "v"(pf0), [pf1] "v"(pf1), [zf0] "v"(zf0), [zf1] "v"(zf1)                   \
        : "memory", TR_CLOBBER32)

// dQ^T[2 blocks] += K^T . dZ^T over the half tile's 32 keys: OK0 = byte offset of the K sub-tile (k-step 0), OK1 = + 2048
#define DQ_BLOCK(OK0, OK1, DQ0, DQ1)                                                                                                               \
    asm volatile(                                                                                                                                  \
        TR_RD("v[224:225]", "%[a0]", OK0) TR_RD("v[226:227]", "%[a1]", OK0) TR_RD("v[228:229]", "%[a2]", OK0) TR_RD("v[230:231]", "%[a3]", OK0)    \
        TR_RD("v[232:233]", "%[a0]", OK1) TR_RD("v[234:235]", "%[a1]", OK1) TR_RD("v[236:237]", "%[a2]", OK1) TR_RD("v[238:239]", "%[a3]", OK1)    \
        "s_waitcnt lgkmcnt(4)\n\t"                                                                                                                 \
        MFMA32("%[dq0]", "v[224:227]", "%[zf0]") MFMA32("%[dq1]", "v[228:231]", "%[zf0]")                                                          \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                                                 \
        MFMA32("%[dq0]", "v[232:235]", "%[zf1]") MFMA32("%[dq1]", "v[236:239]", "%[zf1]")                                                          \
        : [dq0] "+a"(DQ0), [dq1] "+a"(DQ1)                                                                                                         \
        : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [zf0] "v"(zf0), [zf1] "v"(zf1)                                                   \
        : "memory", TR_CLOBBER16)

// First products of one k-step for BOTH 32-row halves: 4 MFMAs whose B operands (this lane's K / V -- or Q / dO -- fragments, loop-invariant)
// live in AGPRs ("a"): as compiler builtins they sat in 64 VGPRs and, with both halves' S / dP accumulators live, pushed the allocator into
// ~250 v_accvgpr moves per tile.  The compiler does not know these are MFMAs: the XDL-write -> VALU-read wait states of the LAST k-step are
// inserted by hand (MFMA_DRAIN; 8-pass MFMA: 11 wait states, CDNA3 ISA 4.5) before the softmax arithmetic reads the accumulators.
#define CHAIN4(S0, D0, S1, D1, FA0, FB0, FA1, FB1, BK, BV)                                                  \
    asm volatile("s_nop 1\n\t"            /* (a VALU-written accumulator / fragment is read at once: no hazard bookkeeping by hipcc here) */ \
                 "v_mfma_f32_32x32x16_bf16 %[s0], %[fa0], %[bk], %[s0]\n\t"                                  \
                 "v_mfma_f32_32x32x16_bf16 %[d0], %[fb0], %[bv], %[d0]\n\t"                                  \
                 "v_mfma_f32_32x32x16_bf16 %[s1], %[fa1], %[bk], %[s1]\n\t"                                  \
                 "v_mfma_f32_32x32x16_bf16 %[d1], %[fb1], %[bv], %[d1]\n\t"                                  \
                 : [s0] "+v"(S0), [d0] "+v"(D0), [s1] "+v"(S1), [d1] "+v"(D1)                                \
                 : [fa0] "v"(FA0), [fb0] "v"(FB0), [fa1] "v"(FA1), [fb1] "v"(FB1), [bk] "a"(BK), [bv] "a"(BV))
#define MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 3" ::: "memory")

// ---- pass 1: dK, dV.  LDS stage: Q [2 subs][64 q][64 d] | dO [2 subs][64 q][64 d] | -L[64] | -Delta[64]; ring of 4 (130 KiB)
constexpr int ST1 = 2 * TILE + 512;
constexpr int NST1 = 4;
__global__ __launch_bounds__(NWAVES * 64, 1) void attn128_bwd_dkv_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane & 31, lg = lane >> 5;
    constexpr int KB = 32 * NWAVES;
    const int Sk = p.S_kv > 0 ? p.S_kv : p.S;                       // keys: the queries' own sequence, or another one (cross-attention)
    const long Skp = p.S_kv > 0 ? p.S_kv_pad : p.S_pad;
    const int nkb = (Sk + KB - 1) / KB;
    const int nwg = nkb * p.H * p.B;
    int wid = blockIdx.x;
    {   // XCD-aware work order: the key blocks of one (b, h) share an L2
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = wid & 7;
        wid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (wid >> 3);
    }
    const int kblk = wid % nkb;
    const long bh = wid / nkb;
    const bf16_t* Qg = p.q + bh * p.S_pad * HD;
    const bf16_t* Og = p.doh + bh * p.S_pad * HD;
    // -L | -Delta of a 64-query tile: 128 contiguous floats in p.nld ([b h][tile][2][64]); wave w moves floats 32 w .. 32 w + 31 with the
    // first 8 lanes of one 16-byte LDS-DMA instruction (every wave: 9 VM operations per tile)
    const float* NLg = p.nld + bh * p.S_pad * 2 + wave * 32 + (lane & 7) * 4;
    const int key = kblk * KB + wave * 32 + lk;
    const int key_ld = key < Sk ? key : Sk - 1;
    // ragged text (Qwen-Image): keys >= kv_len[b] are masked in the forward; their P is zero here, so their dK / dV rows come out as zeros
    int Skv = Sk;
    if (p.kv_len) Skv = min(Skv, max(1, __builtin_amdgcn_readfirstlane(p.kv_len[bh / p.H])));
    const bool key_ok = key < Skv;
    bf16x8 kf[8], vf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        kf[j] = *(const bf16x8*)(p.k + (bh * Skp + key_ld) * HD + j * 16 + lg * 8);
        vf[j] = *(const bf16x8*)(p.v + (bh * Skp + key_ld) * HD + j * 16 + lg * 8);
    }
    auto stage = [&](int t, int buf) {
        char* base = smem + buf * ST1;
        const bf16_t* qs = Qg + (long)t * TB * HD;
        const bf16_t* os = Og + (long)t * TB * HD;
        stage_sub(qs, base, wave, lane);
        stage_sub(qs + 64, base + SUB, wave, lane);
        stage_sub(os, base + TILE, wave, lane);
        stage_sub(os + 64, base + TILE + SUB, wave, lane);
        if (lane < 8) __builtin_amdgcn_global_load_lds((gptr_t)(NLg + (long)t * 2 * TB), (lptr_t)(base + 2 * TILE + wave * 128), 16, 0, 0);
    };
    // row-major sub-tiles as A operand: row = 32*qb + perm(lk), logical chunk = 2*kk + lg
    const int prow = row_perm(lk);
    int offR[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) offR[kk] = prow * 128 + (((2 * kk + lg) ^ swz2(prow)) << 4);
    // transposed reads: this lane supplies the 8-byte piece (row 8 lg + 4 jj + (i >> 2), columns 32 db + 16 g1 + 4 (i & 3) ..) of the block
    int trb[2][2];
    {
        const int i = lane & 15, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int ql = 8 * lg + 4 * jj + (i >> 2);
                const int ch = 4 * db + 2 * g1 + ((i & 3) >> 1);
                trb[db][jj] = ql * 128 + ((ch ^ swz2(ql)) << 4) + (i & 1) * 8;
            }
    }
    f32x16 dk[4], dv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { dk[j] = (f32x16){0}; dv[j] = (f32x16){0}; }
    const int nt = (p.S + TB - 1) / TB;
    stage(0, 0);
    if (nt > 1) stage(1, 1);
    if (nt > 2) stage(2, 2);
    for (int t = 0; t < nt; ++t) {
        wait_tiles_ahead<9>(nt - 1 - t);
        if (t + 3 < nt) stage(t + 3, (t + 3) & (NST1 - 1));      // its buffer held tile t - 1: every wave is past it (the barrier above)
        const char* sb = smem + (t & (NST1 - 1)) * ST1;
        const unsigned stg = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + (t & (NST1 - 1)) * ST1);
        const unsigned sLa = stg + 2 * TILE + 32 * lg;
        const int q_lim = p.S - t * TB;
        // Both 32-query halves' first products are issued back to back (4 independent accumulation chains, 32 MFMAs), THEN each half's
        // softmax arithmetic + second products: with one wave per SIMD nothing else hides the VALU phase -- half 0's exp / pack work now runs
        // while the matrix pipe still executes half 1's chains, half 1's while it executes half 0's second products.
        f32x16 s[2], dp[2];
        {   // -L / -Delta of the tile's queries as the chains' C operands (inline assembly: see attention_bwd.hip)
            f32x4 a[4], b[4];
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:64\n\tds_read_b128 %3, %8 offset:80\n\t"
                         "ds_read_b128 %4, %8 offset:256\n\tds_read_b128 %5, %8 offset:272\n\tds_read_b128 %6, %8 offset:320\n\t"
                         "ds_read_b128 %7, %8 offset:336\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]) : "v"(sLa) : "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[0][4 * j + e] = a[j][e]; dp[0][4 * j + e] = b[j][e]; }
            asm volatile("ds_read_b128 %0, %8 offset:128\n\tds_read_b128 %1, %8 offset:144\n\tds_read_b128 %2, %8 offset:192\n\tds_read_b128 %3, %8 offset:208\n\t"
                         "ds_read_b128 %4, %8 offset:384\n\tds_read_b128 %5, %8 offset:400\n\tds_read_b128 %6, %8 offset:448\n\t"
                         "ds_read_b128 %7, %8 offset:464\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]) : "v"(sLa) : "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[1][4 * j + e] = a[j][e]; dp[1][4 * j + e] = b[j][e]; }
        }
        {   // 8 k-steps x (2 halves x {S, dP}) = 32 MFMAs; the 4 A fragments of step j + 1 are read while step j's MFMAs run (hipcc puts a
            // full lgkmcnt(0) between a read and the MFMA that consumes it when nothing else is in flight: one LDS latency per MFMA)
            bf16x8 fq[2][2], fo[2][2];
            auto rd = [&](int j, int buf) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const int off = (j >> 2) * SUB + offR[j & 3] + qb * 4096;
                    fq[buf][qb] = *(const bf16x8*)(sb + off);
                    fo[buf][qb] = *(const bf16x8*)(sb + TILE + off);
                }
            };
            rd(0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j + 1 < 8) rd(j + 1, (j + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                CHAIN4(s[0], dp[0], s[1], dp[1], fq[j & 1][0], fo[j & 1][0], fq[j & 1][1], fo[j & 1][1], kf[j], vf[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            MFMA_DRAIN();
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            unsigned pk[8], zk[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float p0 = __builtin_amdgcn_exp2f(s[qb][r]), p1 = __builtin_amdgcn_exp2f(s[qb][r + 1]);
                const int ql = 32 * qb + 16 * (r >> 3) + 8 * lg + (r & 7);      // branch-free tail mask: q_lim = valid queries of this tile (64 except the last)
                p0 = (ql < q_lim && key_ok) ? p0 : 0.f;
                p1 = (ql + 1 < q_lim && key_ok) ? p1 : 0.f;
                pk[r >> 1] = pack_bf16(p0, p1);
                zk[r >> 1] = pack_bf16(p0 * dp[qb][r], p1 * dp[qb][r + 1]);
            }
            {
                const bf16x8 pf0 = frag4(pk[0], pk[1], pk[2], pk[3]), pf1 = frag4(pk[4], pk[5], pk[6], pk[7]);
                const bf16x8 zf0 = frag4(zk[0], zk[1], zk[2], zk[3]), zf1 = frag4(zk[4], zk[5], zk[6], zk[7]);
                const unsigned a0 = stg + qb * 4096 + trb[0][0], a1 = stg + qb * 4096 + trb[0][1];
                const unsigned a2 = stg + qb * 4096 + trb[1][0], a3 = stg + qb * 4096 + trb[1][1];
                // d half 0: dO sub-tile at 16384, Q sub-tile at 0; d half 1: dO at 24576, Q at 8192; the second k-step is 16 rows (2048 B) on
                DKV_BLOCK(16384, 0, 18432, 2048, dv[0], dv[1], dk[0], dk[1]);
                DKV_BLOCK(24576, 8192, 26624, 10240, dv[2], dv[3], dk[2], dk[3]);
            }
        }
    }
    __syncthreads();     // every wave is done with the ring: reuse it for the output transposes (8 KiB per wave)
    char* ob = smem + wave * 8192;
    const int row0 = kblk * KB + wave * 32;
    store_rows128(dk, LN2, ob, p.dk + bh * Skp * HD, row0, Sk, lane);
    store_rows128(dv, 1.0f, ob, p.dv + bh * Skp * HD, row0, Sk, lane);
}

// ---- pass 2: dQ.  LDS stage: K [2 subs][64 keys][64 d] | V [2 subs][64 keys][64 d]; ring of 4 (128 KiB)
constexpr int ST2 = 2 * TILE;
constexpr int NST2 = 4;
__global__ __launch_bounds__(NWAVES * 64, 1) void attn128_bwd_dq_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, lg = lane >> 5;
    constexpr int QB = 32 * NWAVES;
    const int nqb = (p.S + QB - 1) / QB;
    const int nwg = nqb * p.H * p.B;
    int wid = blockIdx.x;
    {
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = wid & 7;
        wid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (wid >> 3);
    }
    const int qblk = wid % nqb;
    const long bh = wid / nqb;
    const long Skp = p.S_kv > 0 ? p.S_kv_pad : p.S_pad;               // (cross-attention: the keys are another sequence)
    const bf16_t* Kg = p.k + bh * Skp * HD;
    const bf16_t* Vg = p.v + bh * Skp * HD;
    const int q_row = qblk * QB + wave * 32 + lq;
    const int q_ld = q_row < p.S ? q_row : p.S - 1;
    bf16x8 qf[8], of[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        qf[j] = *(const bf16x8*)(p.q + (bh * p.S_pad + q_ld) * HD + j * 16 + lg * 8);
        of[j] = *(const bf16x8*)(p.doh + (bh * p.S_pad + q_ld) * HD + j * 16 + lg * 8);
    }
    // -L and -Delta of this lane's query as 16-register splats: the C operand of the first MFMA of every S^T / dP^T chain
    f32x16 nL, nD;
    {
        const float l = -p.lse[bh * p.S_pad + q_ld], d = -p.delta[bh * p.S_pad + q_ld];
#pragma unroll
        for (int r = 0; r < 16; ++r) { nL[r] = l; nD[r] = d; }
    }
    auto stage = [&](int t, int buf) {
        char* base = smem + buf * ST2;
        const bf16_t* ks = Kg + (long)t * TB * HD;
        const bf16_t* vs = Vg + (long)t * TB * HD;
        stage_sub(ks, base, wave, lane);
        stage_sub(ks + 64, base + SUB, wave, lane);
        stage_sub(vs, base + TILE, wave, lane);
        stage_sub(vs + 64, base + TILE + SUB, wave, lane);
    };
    const int prow = row_perm(lq);
    int offR[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) offR[kk] = prow * 128 + (((2 * kk + lg) ^ swz2(prow)) << 4);
    int trb[2][2];
    {
        const int i = lane & 15, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int kl = 8 * lg + 4 * jj + (i >> 2);
                const int ch = 4 * db + 2 * g1 + ((i & 3) >> 1);
                trb[db][jj] = kl * 128 + ((ch ^ swz2(kl)) << 4) + (i & 1) * 8;
            }
    }
    f32x16 dq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) dq[j] = (f32x16){0};
    int Skv = p.S_kv > 0 ? p.S_kv : p.S;      // ragged text: this sample's keys end at kv_len[b]
    if (p.kv_len) Skv = min(Skv, max(1, __builtin_amdgcn_readfirstlane(p.kv_len[bh / p.H])));
    const int nt = (Skv + TB - 1) / TB;
    stage(0, 0);
    if (nt > 1) stage(1, 1);
    if (nt > 2) stage(2, 2);
    for (int t = 0; t < nt; ++t) {
        wait_tiles_ahead<8>(nt - 1 - t);
        if (t + 3 < nt) stage(t + 3, (t + 3) % NST2);
        const char* sb = smem + (t % NST2) * ST2;
        const unsigned stg = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + (t % NST2) * ST2);
        const int k_lim = Skv - t * TB;
        f32x16 s[2], dp[2];          // both 32-key halves' chains first (see the dK / dV pass)
        {   // fragments of k-step j + 1 are read while step j's MFMAs run (see the dK / dV pass)
            bf16x8 fk[2][2], fv[2][2];
            auto rd = [&](int j, int buf) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const int off = (j >> 2) * SUB + offR[j & 3] + kb * 4096;
                    fk[buf][kb] = *(const bf16x8*)(sb + off);
                    fv[buf][kb] = *(const bf16x8*)(sb + TILE + off);
                }
            };
            rd(0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j + 1 < 8) rd(j + 1, (j + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                if (j == 0) { s[0] = nL; s[1] = nL; dp[0] = nD; dp[1] = nD; }      // -L / -Delta of this lane's query: the chains' C operands
                CHAIN4(s[0], dp[0], s[1], dp[1], fk[j & 1][0], fv[j & 1][0], fk[j & 1][1], fv[j & 1][1], qf[j], of[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            MFMA_DRAIN();
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            unsigned zk[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float p0 = __builtin_amdgcn_exp2f(s[kb][r]), p1 = __builtin_amdgcn_exp2f(s[kb][r + 1]);
                const int kl = 32 * kb + 16 * (r >> 3) + 8 * lg + (r & 7);      // branch-free tail mask
                p0 = kl < k_lim ? p0 : 0.f;
                p1 = kl + 1 < k_lim ? p1 : 0.f;
                zk[r >> 1] = pack_bf16(p0 * dp[kb][r], p1 * dp[kb][r + 1]);
            }
            {
                const bf16x8 zf0 = frag4(zk[0], zk[1], zk[2], zk[3]), zf1 = frag4(zk[4], zk[5], zk[6], zk[7]);
                const unsigned a0 = stg + kb * 4096 + trb[0][0], a1 = stg + kb * 4096 + trb[0][1];
                const unsigned a2 = stg + kb * 4096 + trb[1][0], a3 = stg + kb * 4096 + trb[1][1];
                DQ_BLOCK(0, 2048, dq[0], dq[1]);            // d half 0: K sub-tile at 0
                DQ_BLOCK(8192, 10240, dq[2], dq[3]);        // d half 1: K sub-tile at 8192
            }
        }
    }
    __syncthreads();
    char* ob = smem + wave * 8192;
    store_rows128(dq, LN2, ob, p.dq + bh * p.S_pad * HD, qblk * QB + wave * 32, p.S, lane);
}

// =============================================================================================== round 6: software-pipelined passes
// The two passes with their main loops as ONE generated asm statement each (attn_bwd128_asm.inc <- gen_attn_bwd128.py; the schedule of
// gen_attn_bwd64.py at head_dim 128): B(h - 1) || V(h) || A(h + 1) interleaved MFMA by MFMA -- with one wave per SIMD nothing else puts the
// exp / mask / pack arithmetic under matrix work.  Same arithmetic and masks in the same order per output element as the kernels above
// (kept: mi355_tune_set(44, 0), and the A/B of tests/test_gpu_flux_backward.py): bit-identical.
#include "attn_bwd128_asm.inc"

__global__ __launch_bounds__(NWAVES * 64, 1) void attn128_bwd_dkv_pipe_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane & 31, lg = lane >> 5;
    constexpr int KB = 32 * NWAVES;
    const int Sk = p.S_kv > 0 ? p.S_kv : p.S;
    const long Skp = p.S_kv > 0 ? p.S_kv_pad : p.S_pad;
    const int nkb = (Sk + KB - 1) / KB;
    const int nwg = nkb * p.H * p.B;
    int wid = blockIdx.x;
    {
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = wid & 7;
        wid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (wid >> 3);
    }
    const int kblk = wid % nkb;
    const long bh = wid / nkb;
    if ((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();      // the asm addresses the ring from LDS byte 0
    const bf16_t* Qg = p.q + bh * p.S_pad * HD;
    const bf16_t* Og = p.doh + bh * p.S_pad * HD;
    const float* NLt = p.nld + bh * p.S_pad * 2;
    const int key = kblk * KB + wave * 32 + lk;
    const int key_ld = key < Sk ? key : Sk - 1;
    int Skv = Sk;
    if (p.kv_len) Skv = min(Skv, max(1, __builtin_amdgcn_readfirstlane(p.kv_len[bh / p.H])));
    const int nt = (p.S + TB - 1) / TB;
    const int srow = wave * 8 + (lane >> 3);
    const unsigned g0 = (unsigned)(srow * 256 + (((lane & 7) ^ swz2(srow)) << 4));
    const unsigned g2 = (unsigned)(wave * 128 + (lane & 7) * 16);
    const unsigned grow = (unsigned)(key_ld * 256 + lg * 16);
    auto stage = [&](int t, int buf) {                   // (tiles 0..2; the loop stages the rest)
        const int tt = t < nt ? t : nt - 1;
        char* base = smem + buf * ST1;
        const bf16_t* qs = Qg + (long)tt * TB * HD;
        const bf16_t* os = Og + (long)tt * TB * HD;
        stage_sub(qs, base, wave, lane);
        stage_sub(qs + 64, base + SUB, wave, lane);
        stage_sub(os, base + TILE, wave, lane);
        stage_sub(os + 64, base + TILE + SUB, wave, lane);
        if (lane < 8) __builtin_amdgcn_global_load_lds((gptr_t)(NLt + (long)tt * 2 * TB + wave * 32 + (lane & 7) * 4), (lptr_t)(base + 2 * TILE + wave * 128), 16, 0, 0);
    };
    const int prow = row_perm(lk);
    unsigned la = (unsigned)(2 * TILE + 32 * lg), r0, a0, a1;
    {
        r0 = (unsigned)(prow * 128 + ((lg ^ swz2(prow)) << 4));                  // fragment kk = 0 (offR[0]); kk = 1..3: r0 ^ 32 kk, formed by the asm
        const int i = lane & 15, g1 = (lane >> 4) & 1;
        unsigned a[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {                                         // pieces (db = 0, jj) (trb[0][jj]); db = 1: ^ 64
            const int ql = 8 * lg + 4 * jj + (i >> 2);
            const int ch = 2 * g1 + ((i & 3) >> 1);
            a[jj] = (unsigned)(ql * 128 + ((ch ^ swz2(ql)) << 4) + (i & 1) * 8);
        }
        a0 = a[0]; a1 = a[1];
    }
    // valid queries of the current tile minus this half-wave's row offset; a lane whose key is masked (beyond the sample's keys) never has any
    int vrem = key < Skv ? p.S - 8 * lg : -(1 << 30);
    stage(0, 0); stage(1, 1); stage(2, 2);
    const int lt = nt - 1 < 3 ? nt - 1 : 3;
    const unsigned long long b0 = (unsigned long long)(Qg + (long)lt * TB * HD), b1 = (unsigned long long)(Og + (long)lt * TB * HD);
    const unsigned long long b2 = (unsigned long long)(NLt + (long)lt * 2 * TB);
    const unsigned long long p0 = (unsigned long long)(p.k + bh * Skp * HD), p1 = (unsigned long long)(p.v + bh * Skp * HD);
    asm volatile(ABWD128_DKV_ASM
                 : [la] "+v"(la), [r0] "+v"(r0), [a0] "+v"(a0), [a1] "+v"(a1), [vrem] "+v"(vrem)
                 : [g0] "v"(g0), [g2] "v"(g2), [grow] "v"(grow), [b0] "s"(b0), [b1] "s"(b1), [b2] "s"(b2), [p0] "s"(p0), [p1] "s"(p1), [nt] "s"(nt),
                   [wv] "s"(wave)
                 : ABWD128_DKV_CLOBBERS);
    f32x16 dk[4], dv[4];
    ABWD128_READ_ACC_0(dv[0]) ABWD128_READ_ACC_16(dv[1]) ABWD128_READ_ACC_32(dv[2]) ABWD128_READ_ACC_48(dv[3])
    ABWD128_READ_ACC_64(dk[0]) ABWD128_READ_ACC_80(dk[1]) ABWD128_READ_ACC_96(dk[2]) ABWD128_READ_ACC_112(dk[3])
    // a key beyond its sample's key count (ragged text): its P column is zero in every tile -- the loop masks the last tile only, so the lane's
    // sums are dropped here instead (the round-4 kernel's rows for such keys are exact zeros too)
    if (key >= Skv) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { dk[j] = (f32x16){0}; dv[j] = (f32x16){0}; }
    }
    __syncthreads();     // every wave is done with the ring (the asm ends with vmcnt(0)): reuse it for the output transposes (8 KiB per wave)
    int lane_e = lane;   // (an opaque copy: the epilogue's lane arithmetic must not be formed in front of the loop -- see attention_bwd.hip)
    asm volatile("" : "+v"(lane_e));
    char* ob = smem + wave * 8192;
    const int row0 = kblk * KB + wave * 32;
    store_rows128(dk, LN2, ob, p.dk + bh * Skp * HD, row0, Sk, lane_e);
    store_rows128(dv, 1.0f, ob, p.dv + bh * Skp * HD, row0, Sk, lane_e);
}

__global__ __launch_bounds__(NWAVES * 64, 1) void attn128_bwd_dq_pipe_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, lg = lane >> 5;
    constexpr int QB = 32 * NWAVES;
    const int nqb = (p.S + QB - 1) / QB;
    const int nwg = nqb * p.H * p.B;
    int wid = blockIdx.x;
    {
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = wid & 7;
        wid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (wid >> 3);
    }
    const int qblk = wid % nqb;
    const long bh = wid / nqb;
    if ((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
    const long Skp = p.S_kv > 0 ? p.S_kv_pad : p.S_pad;
    const bf16_t* Kg = p.k + bh * Skp * HD;
    const bf16_t* Vg = p.v + bh * Skp * HD;
    const int q_row = qblk * QB + wave * 32 + lq;
    const int q_ld = q_row < p.S ? q_row : p.S - 1;
    int Skv = p.S_kv > 0 ? p.S_kv : p.S;      // ragged text: this sample's keys end at kv_len[b]
    if (p.kv_len) Skv = min(Skv, max(1, __builtin_amdgcn_readfirstlane(p.kv_len[bh / p.H])));
    const int nt = (Skv + TB - 1) / TB;
    const int srow = wave * 8 + (lane >> 3);
    const unsigned g0 = (unsigned)(srow * 256 + (((lane & 7) ^ swz2(srow)) << 4));
    const unsigned grow = (unsigned)(q_ld * 256 + lg * 16);
    const float nl = -p.lse[bh * p.S_pad + q_ld], nd = -p.delta[bh * p.S_pad + q_ld];
    auto stage = [&](int t, int buf) {
        const int tt = t < nt ? t : nt - 1;
        char* base = smem + buf * ST2;
        const bf16_t* ks = Kg + (long)tt * TB * HD;
        const bf16_t* vs = Vg + (long)tt * TB * HD;
        stage_sub(ks, base, wave, lane);
        stage_sub(ks + 64, base + SUB, wave, lane);
        stage_sub(vs, base + TILE, wave, lane);
        stage_sub(vs + 64, base + TILE + SUB, wave, lane);
    };
    const int prow = row_perm(lq);
    unsigned r0, a0, a1;
    {
        r0 = (unsigned)(prow * 128 + ((lg ^ swz2(prow)) << 4));
        const int i = lane & 15, g1 = (lane >> 4) & 1;
        unsigned a[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int kl = 8 * lg + 4 * jj + (i >> 2);
            const int ch = 2 * g1 + ((i & 3) >> 1);
            a[jj] = (unsigned)(kl * 128 + ((ch ^ swz2(kl)) << 4) + (i & 1) * 8);
        }
        a0 = a[0]; a1 = a[1];
    }
    int vrem = Skv - 8 * lg;                  // valid keys of the current tile minus this half-wave's row offset
    stage(0, 0); stage(1, 1); stage(2, 2);
    const int lt = nt - 1 < 3 ? nt - 1 : 3;
    const unsigned long long b0 = (unsigned long long)(Kg + (long)lt * TB * HD), b1 = (unsigned long long)(Vg + (long)lt * TB * HD);
    const unsigned long long p0 = (unsigned long long)(p.q + bh * p.S_pad * HD), p1 = (unsigned long long)(p.doh + bh * p.S_pad * HD);
    asm volatile(ABWD128_DQ_ASM
                 : [r0] "+v"(r0), [a0] "+v"(a0), [a1] "+v"(a1), [vrem] "+v"(vrem)
                 : [g0] "v"(g0), [grow] "v"(grow), [nl] "v"(nl), [nd] "v"(nd), [b0] "s"(b0), [b1] "s"(b1), [p0] "s"(p0), [p1] "s"(p1), [nt] "s"(nt),
                   [wv] "s"(wave)
                 : ABWD128_DQ_CLOBBERS);
    f32x16 dq[4];
    ABWD128_READ_ACC_0(dq[0]) ABWD128_READ_ACC_16(dq[1]) ABWD128_READ_ACC_32(dq[2]) ABWD128_READ_ACC_48(dq[3])
    __syncthreads();
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    char* ob = smem + wave * 8192;
    store_rows128(dq, LN2, ob, p.dq + bh * p.S_pad * HD, qblk * QB + wave * 32, p.S, lane_e);
}

__device__ __forceinline__ void unpack8(const uint4 u, float (&v)[8]) {
    v[0] = bf_lo(u.x); v[1] = bf_hi(u.x); v[2] = bf_lo(u.y); v[3] = bf_hi(u.y);
    v[4] = bf_lo(u.z); v[5] = bf_hi(u.z); v[6] = bf_lo(u.w); v[7] = bf_hi(u.w);
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    return make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
}
__device__ __forceinline__ float row16_sum_dpp(float v) {      // sum over the 16 lanes of a DPP row, in every lane
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
    };
    using std::integral_constant;
    v += dpp(v, integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v += dpp(v, integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    v += dpp(v, integral_constant<int, 0x141>{});     // row_half_mirror
    v += dpp(v, integral_constant<int, 0x140>{});     // row_mirror
    return v;
}

// ---- prep: token-major o / dO (split like the forward's output: the first n_first positions of a sample in one buffer, the rest in another,
// each with its own leading dimension) -> doh [B][H][S_pad][128], delta [B][H][S_pad] = sum_d dO * O, nld = -lse | -delta per 64-query tile.
// One workgroup = 64 positions of one (b, h); 16 lanes (16 bytes each) per position; padded rows stay zero (zero-initialised buffers).
__global__ __launch_bounds__(256) void attn128_bwd_prep_kernel(Attn128BwdPrepParams p) {
    const int s0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const long bh = (long)b * p.H + h;
    const int c = threadIdx.x & 15;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int s = s0 + pass * 16 + (threadIdx.x >> 4);
        if (s >= p.S) continue;               // (uniform inside a 16-lane DPP row)
        const bool first = s < p.n_first;
        const long row = first ? ((long)b * p.n_first + s) : ((long)b * (p.S - p.n_first) + (s - p.n_first));
        const bf16_t* op = first ? p.o_first + row * p.ld_o_first : p.o_rest + row * p.ld_o_rest;
        const bf16_t* dop = first ? (p.do_first ? p.do_first + row * p.ld_do_first : nullptr) : (p.do_rest ? p.do_rest + row * p.ld_do_rest : nullptr);
        const uint4 du = dop ? *(const uint4*)(dop + h * HD + c * 8) : make_uint4(0u, 0u, 0u, 0u);
        const uint4 ou = *(const uint4*)(op + h * HD + c * 8);
        *(uint4*)(p.doh + (bh * p.S_pad + s) * HD + c * 8) = du;
        float dv[8], ov[8];
        unpack8(du, dv); unpack8(ou, ov);
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) part += dv[e] * ov[e];
        const float dl = row16_sum_dpp(part);
        if (c == 0) {
            p.delta[bh * p.S_pad + s] = dl;
            float* nl = p.nld + (bh * p.S_pad + (s & ~63)) * 2 + (s & 63);
            nl[0] = -p.lse[bh * p.S_pad + s];
            nl[64] = -dl;
        }
    }
}

// ---- backward of the q | k producer (flux_ops.hip rope_norm_kernel) + gather.  Forward per (token, head), x = projection + bias:
//        y = x * r * w   (r = 1 / rms over the head's 128 features),   z = RoPE(y) [adjacent pairs: (y0 c - y1 s, y1 c + y0 s)],   q~ = z * q_scale
//      Backward from dq~ (w.r.t. the STORED q~) and the stored q~ itself:  dz = dq~ * q_scale;  dy = R^T dz = (dz0 c + dz1 s, dz1 c - dz0 s);
//      y = R^T (q~ / q_scale);  xhat = y / w;  g = dy * w;  dx = r * (g - xhat * mean(g * xhat)).  k alike with q_scale = 1; v passes through.
//      One wave per token, 4 heads per pass, a head per 16-lane DPP row, 16 bytes per lane and access (the forward kernel's mapping).
//      Output rows [M][ld_out]: dq_pre at column 0, dk_pre at D, dv at 2 D (D = H * 128).
__global__ __launch_bounds__(256) void rope_rms_bwd128_kernel(RopeRmsBwdParams p) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= p.M) return;
    const int hl = lane >> 4, d0 = (lane & 15) * 8;
    const int D = p.H * HD;
    const int b = m / p.rows_per_sample;
    const int s = m - b * p.rows_per_sample + p.s_off;
    float wq[8], wk[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { wq[e] = p.nw_q[d0 + e]; wk[e] = p.nw_k[d0 + e]; }
    const float4 c01 = *(const float4*)(p.cs + (long)s * 64 + (lane & 15) * 4), c23 = *(const float4*)(p.cs + (long)s * 64 + (lane & 15) * 4 + 2);
    const float cs[4][2] = {{c01.x, c01.y}, {c01.z, c01.w}, {c23.x, c23.y}, {c23.z, c23.w}};
    const float inv_qs = 1.0f / p.q_scale;
    bf16_t* out = p.out + (long)m * p.ld_out;
    const float* rstd = p.rstd + (long)m * (2L * p.H);
    for (int h0 = 0; h0 < p.H; h0 += 4) {
        const int h = h0 + hl;
        if (h >= p.H) continue;               // (uniform inside a 16-lane DPP row)
        const long src = (((long)b * p.H + h) * p.S_pad + s) * HD + d0;
        auto one = [&](const bf16_t* zv, const bf16_t* dzv, const float (&w)[8], float r, float z_scale, float dz_scale, bf16_t* dst) {
            float z[8], dz[8], o[8], xh[8], g[8];
            unpack8(*(const uint4*)(zv + src), z);
            unpack8(*(const uint4*)(dzv + src), dz);
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float cc = cs[j][0], ss = cs[j][1];
                const float z0 = z[2 * j] * z_scale, z1 = z[2 * j + 1] * z_scale;
                const float g0 = dz[2 * j] * dz_scale, g1 = dz[2 * j + 1] * dz_scale;
                const float y0 = z0 * cc + z1 * ss, y1 = z1 * cc - z0 * ss;         // y = R^T z
                const float dy0 = g0 * cc + g1 * ss, dy1 = g1 * cc - g0 * ss;       // dy = R^T dz
                xh[2 * j] = fabsf(w[2 * j]) > 1e-20f ? y0 / w[2 * j] : 0.f;
                xh[2 * j + 1] = fabsf(w[2 * j + 1]) > 1e-20f ? y1 / w[2 * j + 1] : 0.f;
                g[2 * j] = dy0 * w[2 * j];
                g[2 * j + 1] = dy1 * w[2 * j + 1];
                part += g[2 * j] * xh[2 * j] + g[2 * j + 1] * xh[2 * j + 1];
            }
            const float mgx = row16_sum_dpp(part) * (1.0f / 128.0f);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = r * (g[e] - xh[e] * mgx);
            *(uint4*)dst = pack8(o);
        };
        one(p.q, p.dq, wq, rstd[h], inv_qs, p.q_scale, out + h * HD + d0);
        one(p.k, p.dk, wk, rstd[p.H + h], 1.0f, 1.0f, out + D + h * HD + d0);
        *(uint4*)(out + 2 * D + h * HD + d0) = *(const uint4*)(p.dv + src);
    }
}

}  // namespace

static int g_attn128_bwd_pipe = 1;      // mi355_tune_set(44, .): 1 = the software-pipelined passes (round 6), 0 = the round-4 kernels
void set_attn128_bwd_pipe(int v) { g_attn128_bwd_pipe = v; }

hipError_t launch_attention128_bwd(const AttnBwdParams& p, hipStream_t stream) {
    const int Sk = p.S_kv > 0 ? p.S_kv : p.S;
    const long Skp = p.S_kv > 0 ? p.S_kv_pad : p.S_pad;
    if (p.S <= 0 || p.S_pad % TB != 0 || p.S_pad < p.S || Skp % TB != 0 || Skp < Sk || !p.nld) return hipErrorInvalidValue;
    if (sched_trace_on()) {
        const size_t bhs = (size_t)p.B * p.H * p.S_pad, bhk = (size_t)p.B * p.H * Skp;
        sched_trace_launch("attention128_bwd", stream, {treg(p.q, bhs * 256), treg(p.k, bhk * 256), treg(p.v, bhk * 256), treg(p.doh, bhs * 256),
                                                        treg(p.lse, bhs * 4), treg(p.delta, bhs * 4), treg(p.nld, bhs * 8)},
                           {treg(p.dq, bhs * 256), treg(p.dk, bhk * 256), treg(p.dv, bhk * 256)});
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attn128_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NST1 * ST1);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)attn128_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NST2 * ST2);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nbq = (p.S + 32 * NWAVES - 1) / (32 * NWAVES), nbk = (Sk + 32 * NWAVES - 1) / (32 * NWAVES);
    if (g_attn128_bwd_pipe) {
        static bool attr_set2 = false;
        if (!attr_set2) {
            hipError_t e = hipFuncSetAttribute((const void*)attn128_bwd_dkv_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NST1 * ST1);
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute((const void*)attn128_bwd_dq_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NST2 * ST2);
            if (e != hipSuccess) return e;
            attr_set2 = true;
        }
        hipLaunchKernelGGL(attn128_bwd_dkv_pipe_kernel, dim3(nbk * p.H * p.B), dim3(NWAVES * 64), NST1 * ST1, stream, p);
        hipLaunchKernelGGL(attn128_bwd_dq_pipe_kernel, dim3(nbq * p.H * p.B), dim3(NWAVES * 64), NST2 * ST2, stream, p);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(attn128_bwd_dkv_kernel, dim3(nbk * p.H * p.B), dim3(NWAVES * 64), NST1 * ST1, stream, p);
    hipLaunchKernelGGL(attn128_bwd_dq_kernel, dim3(nbq * p.H * p.B), dim3(NWAVES * 64), NST2 * ST2, stream, p);
    return hipGetLastError();
}

hipError_t launch_attn128_bwd_prep(const Attn128BwdPrepParams& p, hipStream_t stream) {
    if (p.S <= 0 || p.S_pad % 64 != 0 || p.S_pad < p.S || p.n_first < 0 || p.n_first > p.S) return hipErrorInvalidValue;
    if (sched_trace_on()) {
        const size_t bhs = (size_t)p.B * p.H * p.S_pad;
        const size_t r1 = (size_t)p.B * p.n_first, r2 = (size_t)p.B * (p.S - p.n_first), hb = (size_t)p.H * 256;
        sched_trace_launch("attn128_bwd_prep", stream,
                           {treg(p.o_first, r1 ? (r1 - 1) * p.ld_o_first * 2 + hb : 0), treg(p.o_rest, r2 ? (r2 - 1) * p.ld_o_rest * 2 + hb : 0),
                            treg(p.do_first, (p.do_first && r1) ? (r1 - 1) * p.ld_do_first * 2 + hb : 0),
                            treg(p.do_rest, (p.do_rest && r2) ? (r2 - 1) * p.ld_do_rest * 2 + hb : 0), treg(p.lse, bhs * 4)},
                           {treg(p.doh, bhs * 256), treg(p.delta, bhs * 4), treg(p.nld, bhs * 8)});
    }
    hipLaunchKernelGGL(attn128_bwd_prep_kernel, dim3((p.S + 63) / 64, p.H, p.B), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_rope_rms_bwd128(const RopeRmsBwdParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.rows_per_sample <= 0 || p.q_scale == 0.f) return hipErrorInvalidValue;
    if (sched_trace_on()) {
        const size_t hm = (size_t)(p.M / p.rows_per_sample) * p.H * p.S_pad * 256;
        sched_trace_launch("rope_rms_bwd128", stream, {treg(p.q, hm), treg(p.k, hm), treg(p.dq, hm), treg(p.dk, hm), treg(p.dv, hm),
                                                       treg(p.rstd, (size_t)p.M * 2 * p.H * 4)},
                           {tregs(p.out, (size_t)3 * p.H * 256, (size_t)p.ld_out * 2, (size_t)p.M)});
    }
    hipLaunchKernelGGL(rope_rms_bwd128_kernel, dim3((p.M + 3) / 4), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace mi355
