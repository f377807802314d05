// mi355_flow -- weight-gradient GEMM on ROW-MAJOR operands (round 6):
//
//   dW[n][k] = sum_m dY[m][n] * X[m][k]        dY: [M][lda] (n contiguous), X: [M][ldb] (k contiguous), bf16;  dW fp32 [N][ldo]
//
// The reduction index m is the ROW index of both operands.  Rounds 2-5 fed this product to the K-contiguous kernel of gemm.hip through two
// HBM-bound transposes per weight (dY^T, X^T: `transpose_kernel`, 392 launches = 4.6 ms of the 87 ms SD3.5 optimize() step, 3.8 %;
// profiles/r05_final_sd3_train_step_default_set_kernel_stats.txt).  Here the row-major [64 m][128 cols] tiles are staged as they lie in HBM
// and the MFMA fragments -- 8 consecutive m for one column -- are read out of them with `ds_read_b64_tr_b16` (per 16-lane group: a
// [4 rows][16 columns] block, lane i receives column i; two reads per fragment), the idiom of attention_bwd.hip.  Same
// v_mfma_f32_16x16x32_bf16, same operand order (X fragment first), same ascending-m accumulation and the same split boundaries as the
// transposed-copy path: the fp32 partial sums are BIT-IDENTICAL to it (tests/test_gpu_backward.py).
//
//   * tile 128 (n) x 128 (k) x 64 (m), 4 waves 2 x 2, wave tile 64 x 64 (acc[4][4] f32x4), 2 LDS stages of 32 KiB, two workgroups per CU;
//   * LDS rows are 256 B; the 32-byte chunk index of a row is XORed with s(row) = (row & 3) | ((row >> 3) & 1) << 2, applied on the
//     SOURCE address of the LDS-DMA and on the read address: the 8 rows a 32-lane half of a transposed read touches (4 rows of two k-groups
//     8 rows apart) land in 8 distinct 32-byte bank groups;
//   * the transposed reads and their MFMAs are one inline-assembly block per k-step with the fragments in FIXED registers (a fragment is two
//     64-bit reads into the halves of one 4-register operand; and hipcc guards every LDS read it recognises with `s_waitcnt vmcnt(0)` while an
//     LDS-DMA is in flight, which would drain the prefetch of the next m-tile);
//   * split over m (k_split) into fp32 partial buffers like gemm_kernel<.., EPI_F32>; the caller reduces in fixed order (backward.hip);
//   * optional bias gradient (colsum != nullptr): the column sums of dY per split, taken from the fragments the k-tile-0 blocks hold
//     (v_dot2c_f32_bf16 against a constant (1, 1): 16 VALU per k-step in 1 / (K / 128) of the blocks) -- no extra pass over dY.
// Requirements (launcher): N % 128 == 0, K % 128 == 0, lda / ldb multiples of 8, operands below 4 GiB; any M (a ragged last m-tile reads zeros).
#include "kernels.h"

namespace mi355 {
namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int TM = 64;                       // m rows per tile
constexpr int OP_BYTES = TM * 256;           // one operand tile: 64 rows x 128 columns bf16
constexpr int STAGE = 2 * OP_BYTES;          // 32 KiB

__device__ __forceinline__ int swz_row(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

#define TN_RD(dst, a, off) "ds_read_b64_tr_b16 " dst ", " a " offset:" #off "\n\t"
#define TN_MFMA(c, w, x) "v_mfma_f32_16x16x32_bf16 " c ", " w ", " x ", " c "\n\t"
// one k-step (32 m): 16 transposed reads into v[224:255] (x fragments i = 0..3 at v[224 + 4 i ..], w fragments j = 0..3 at v[240 + 4 j ..]),
// then the 16 MFMAs of the 64 x 64 wave tile.  KOFF = byte offset of the k-step inside the operand tile (32 rows x 256 B).
// CS: "" or TN_CS -- the column sums of dY (the bias gradient) taken from the x fragments the step already holds: v_dot2c_f32_bf16 with a
// constant (1, 1) operand adds both bf16 halves of a register to an fp32 accumulator; the four registers of fragment i are 8 consecutive m of
// column 16 i + r.  Issued behind the last wait of the step, i.e. beside its last MFMAs.
#define TN_CS                                                                                                                    \
    "v_dot2c_f32_bf16 %[s0], %[one], v224\n\tv_dot2c_f32_bf16 %[s0], %[one], v225\n\tv_dot2c_f32_bf16 %[s0], %[one], v226\n\t"    \
    "v_dot2c_f32_bf16 %[s0], %[one], v227\n\tv_dot2c_f32_bf16 %[s1], %[one], v228\n\tv_dot2c_f32_bf16 %[s1], %[one], v229\n\t"    \
    "v_dot2c_f32_bf16 %[s1], %[one], v230\n\tv_dot2c_f32_bf16 %[s1], %[one], v231\n\tv_dot2c_f32_bf16 %[s2], %[one], v232\n\t"    \
    "v_dot2c_f32_bf16 %[s2], %[one], v233\n\tv_dot2c_f32_bf16 %[s2], %[one], v234\n\tv_dot2c_f32_bf16 %[s2], %[one], v235\n\t"    \
    "v_dot2c_f32_bf16 %[s3], %[one], v236\n\tv_dot2c_f32_bf16 %[s3], %[one], v237\n\tv_dot2c_f32_bf16 %[s3], %[one], v238\n\t"    \
    "v_dot2c_f32_bf16 %[s3], %[one], v239\n\t"
#define TN_STEP_G(KOFF, KOFF4, CS, A0)                                                                                               \
    asm volatile(                                                                                                                \
        TN_RD("v[224:225]", "%[xa0]", KOFF) TN_RD("v[226:227]", "%[xa0]", KOFF4)                                                  \
        TN_RD("v[240:241]", "%[wa0]", KOFF) TN_RD("v[242:243]", "%[wa0]", KOFF4)                                                  \
        TN_RD("v[244:245]", "%[wa1]", KOFF) TN_RD("v[246:247]", "%[wa1]", KOFF4)                                                  \
        TN_RD("v[248:249]", "%[wa2]", KOFF) TN_RD("v[250:251]", "%[wa2]", KOFF4)                                                  \
        TN_RD("v[252:253]", "%[wa3]", KOFF) TN_RD("v[254:255]", "%[wa3]", KOFF4)                                                  \
        TN_RD("v[228:229]", "%[xa1]", KOFF) TN_RD("v[230:231]", "%[xa1]", KOFF4)                                                  \
        TN_RD("v[232:233]", "%[xa2]", KOFF) TN_RD("v[234:235]", "%[xa2]", KOFF4)                                                  \
        TN_RD("v[236:237]", "%[xa3]", KOFF) TN_RD("v[238:239]", "%[xa3]", KOFF4)                                                  \
        "s_waitcnt lgkmcnt(6)\n\t"                                                                                               \
        TN_MFMA("%[c00]", "v[240:243]", "v[224:227]") TN_MFMA("%[c01]", "v[244:247]", "v[224:227]")                               \
        TN_MFMA("%[c02]", "v[248:251]", "v[224:227]") TN_MFMA("%[c03]", "v[252:255]", "v[224:227]")                               \
        "s_waitcnt lgkmcnt(4)\n\t"                                                                                               \
        TN_MFMA("%[c10]", "v[240:243]", "v[228:231]") TN_MFMA("%[c11]", "v[244:247]", "v[228:231]")                               \
        TN_MFMA("%[c12]", "v[248:251]", "v[228:231]") TN_MFMA("%[c13]", "v[252:255]", "v[228:231]")                               \
        "s_waitcnt lgkmcnt(2)\n\t"                                                                                               \
        TN_MFMA("%[c20]", "v[240:243]", "v[232:235]") TN_MFMA("%[c21]", "v[244:247]", "v[232:235]")                               \
        TN_MFMA("%[c22]", "v[248:251]", "v[232:235]") TN_MFMA("%[c23]", "v[252:255]", "v[232:235]")                               \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                               \
        TN_MFMA("%[c30]", "v[240:243]", "v[236:239]") TN_MFMA("%[c31]", "v[244:247]", "v[236:239]")                               \
        TN_MFMA("%[c32]", "v[248:251]", "v[236:239]") TN_MFMA("%[c33]", "v[252:255]", "v[236:239]")                               \
        CS                                                                                                                       \
        : [c00] "+v"(acc[A0 + 0][0]), [c01] "+v"(acc[A0 + 0][1]), [c02] "+v"(acc[A0 + 0][2]), [c03] "+v"(acc[A0 + 0][3]),             \
          [c10] "+v"(acc[A0 + 1][0]), [c11] "+v"(acc[A0 + 1][1]), [c12] "+v"(acc[A0 + 1][2]), [c13] "+v"(acc[A0 + 1][3]),             \
          [c20] "+v"(acc[A0 + 2][0]), [c21] "+v"(acc[A0 + 2][1]), [c22] "+v"(acc[A0 + 2][2]), [c23] "+v"(acc[A0 + 2][3]),             \
          [c30] "+v"(acc[A0 + 3][0]), [c31] "+v"(acc[A0 + 3][1]), [c32] "+v"(acc[A0 + 3][2]), [c33] "+v"(acc[A0 + 3][3]),             \
          [s0] "+v"(cs[A0 + 0]), [s1] "+v"(cs[A0 + 1]), [s2] "+v"(cs[A0 + 2]), [s3] "+v"(cs[A0 + 3])                                 \
        : [one] "v"(ones), [xa0] "v"(xa[A0 + 0]), [xa1] "v"(xa[A0 + 1]), [xa2] "v"(xa[A0 + 2]), [xa3] "v"(xa[A0 + 3]), [wa0] "v"(wa[0]),   \
          [wa1] "v"(wa[1]), [wa2] "v"(wa[2]), [wa3] "v"(wa[3])                                                                       \
        : "memory", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237",  \
          "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252",  \
          "v253", "v254", "v255")
// (the same step without the column-sum operands: the 256 x 256 form has no registers to spare for them)
#define TN_STEP_NC(KOFF, KOFF4, A0)                                                                                                      \
    asm volatile(                                                                                                                \
        TN_RD("v[224:225]", "%[xa0]", KOFF) TN_RD("v[226:227]", "%[xa0]", KOFF4)                                                  \
        TN_RD("v[240:241]", "%[wa0]", KOFF) TN_RD("v[242:243]", "%[wa0]", KOFF4)                                                  \
        TN_RD("v[244:245]", "%[wa1]", KOFF) TN_RD("v[246:247]", "%[wa1]", KOFF4)                                                  \
        TN_RD("v[248:249]", "%[wa2]", KOFF) TN_RD("v[250:251]", "%[wa2]", KOFF4)                                                  \
        TN_RD("v[252:253]", "%[wa3]", KOFF) TN_RD("v[254:255]", "%[wa3]", KOFF4)                                                  \
        TN_RD("v[228:229]", "%[xa1]", KOFF) TN_RD("v[230:231]", "%[xa1]", KOFF4)                                                  \
        TN_RD("v[232:233]", "%[xa2]", KOFF) TN_RD("v[234:235]", "%[xa2]", KOFF4)                                                  \
        TN_RD("v[236:237]", "%[xa3]", KOFF) TN_RD("v[238:239]", "%[xa3]", KOFF4)                                                  \
        "s_waitcnt lgkmcnt(6)\n\t"                                                                                               \
        TN_MFMA("%[c00]", "v[240:243]", "v[224:227]") TN_MFMA("%[c01]", "v[244:247]", "v[224:227]")                               \
        TN_MFMA("%[c02]", "v[248:251]", "v[224:227]") TN_MFMA("%[c03]", "v[252:255]", "v[224:227]")                               \
        "s_waitcnt lgkmcnt(4)\n\t"                                                                                               \
        TN_MFMA("%[c10]", "v[240:243]", "v[228:231]") TN_MFMA("%[c11]", "v[244:247]", "v[228:231]")                               \
        TN_MFMA("%[c12]", "v[248:251]", "v[228:231]") TN_MFMA("%[c13]", "v[252:255]", "v[228:231]")                               \
        "s_waitcnt lgkmcnt(2)\n\t"                                                                                               \
        TN_MFMA("%[c20]", "v[240:243]", "v[232:235]") TN_MFMA("%[c21]", "v[244:247]", "v[232:235]")                               \
        TN_MFMA("%[c22]", "v[248:251]", "v[232:235]") TN_MFMA("%[c23]", "v[252:255]", "v[232:235]")                               \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                               \
        TN_MFMA("%[c30]", "v[240:243]", "v[236:239]") TN_MFMA("%[c31]", "v[244:247]", "v[236:239]")                               \
        TN_MFMA("%[c32]", "v[248:251]", "v[236:239]") TN_MFMA("%[c33]", "v[252:255]", "v[236:239]")                               \
        : [c00] "+v"(acc[A0 + 0][0]), [c01] "+v"(acc[A0 + 0][1]), [c02] "+v"(acc[A0 + 0][2]), [c03] "+v"(acc[A0 + 0][3]),             \
          [c10] "+v"(acc[A0 + 1][0]), [c11] "+v"(acc[A0 + 1][1]), [c12] "+v"(acc[A0 + 1][2]), [c13] "+v"(acc[A0 + 1][3]),             \
          [c20] "+v"(acc[A0 + 2][0]), [c21] "+v"(acc[A0 + 2][1]), [c22] "+v"(acc[A0 + 2][2]), [c23] "+v"(acc[A0 + 2][3]),             \
          [c30] "+v"(acc[A0 + 3][0]), [c31] "+v"(acc[A0 + 3][1]), [c32] "+v"(acc[A0 + 3][2]), [c33] "+v"(acc[A0 + 3][3])              \
        : [xa0] "v"(xa[A0 + 0]), [xa1] "v"(xa[A0 + 1]), [xa2] "v"(xa[A0 + 2]), [xa3] "v"(xa[A0 + 3]), [wa0] "v"(wa[0]),   \
          [wa1] "v"(wa[1]), [wa2] "v"(wa[2]), [wa3] "v"(wa[3])                                                                       \
        : "memory", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237",  \
          "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252",  \
          "v253", "v254", "v255")
#define TN_STEP(KOFF, KOFF4, CS) TN_STEP_G(KOFF, KOFF4, CS, 0)


__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;          // wave tile: output rows (n of dY) 64 wm .., output columns (k of X) 64 wn ..

    // ---- tile mapping (XCD-aware, as gemm_kernel): blocks of one split first
    const int ntn = p.N / 128, ntk = p.K / 128;
    const int nsplit = p.k_split > 1 ? p.k_split : 1;
    const int nblk = ntn * ntk * nsplit;
    int bid = blockIdx.x;
    {
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    const int split = bid / (ntn * ntk);
    bid -= split * (ntn * ntk);
    const int tn = bid / ntk, tk = bid - tn * ntk;
    const int n0 = tn * 128, k0 = tk * 128;

    // ---- LDS-DMA: one instruction = 4 rows x 256 B; group g (16 per operand tile) belongs to wave g & 3.  Lane -> (row l >> 4 of the
    // group, physical 16-byte chunk l & 15); the physical 32-byte chunk pc holds the logical chunk pc ^ s(row).
    unsigned soffA[4], soffB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = wave + 4 * i;
        const int row = g * 4 + (lane >> 4);
        const int c32 = ((lane & 15) >> 1) ^ swz_row(row);
        const int col = c32 * 16 + (lane & 1) * 8;
        soffA[i] = ((unsigned)row * (unsigned)p.lda + (unsigned)(n0 + col)) * 2u;
        soffB[i] = ((unsigned)row * (unsigned)p.ldb + (unsigned)(k0 + col)) * 2u;
    }
    // ragged M (round 6b: the other engines' token counts -- 20 280, 6889 + text -- are not multiples of 64): the rows [M, M_pad) of the LAST m-tile
    // are read from a 16-byte block of zeros instead (what the transposed-copy path's zero padding contributes: nothing)
    auto stage = [&](int mt, int buf) {
        char* base = smem + buf * STAGE;
        const char* ga = (const char*)p.A + (long)mt * TM * p.lda * 2;      // wave-uniform
        const char* gb = (const char*)p.B + (long)mt * TM * p.ldb * 2;
        const bool ragged = (mt + 1) * TM > p.M;                            // wave-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = wave + 4 * i;
            const char* sa = ga + soffA[i];
            const char* sb = gb + soffB[i];
            if (ragged && mt * TM + g * 4 + (lane >> 4) >= p.M) { sa = (const char*)p.zeros; sb = (const char*)p.zeros; }
            __builtin_amdgcn_global_load_lds((gptr_t)sa, (lptr_t)(base + g * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)sb, (lptr_t)(base + OP_BYTES + g * 1024), 16, 0, 0);
        }
    };

    // ---- transposed-read addresses: lane (r = l & 15, kg = l >> 4) of 16-column block b supplies the 8-byte piece
    // (row kg * 8 [+ 4 for the second read] + (r >> 2), columns 16 b + 4 (r & 3) ..) and receives column r of the 4-row block.
    const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem);
    const int r = lane & 15, kg = lane >> 4;
    const int rrow = kg * 8 + (r >> 2);
    const int sw = swz_row(rrow);                    // (unchanged by the + 4 of the second read and the + 32 of the second k-step)
    unsigned xa_s[2][4], wa_s[2][4];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            xa_s[st][b] = lds0 + st * STAGE + rrow * 256 + (((wm * 4 + b) ^ sw) << 5) + (r & 3) * 8;
            wa_s[st][b] = lds0 + st * STAGE + OP_BYTES + rrow * 256 + (((wn * 4 + b) ^ sw) << 5) + (r & 3) * 8;
        }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned ones = 0x3f803f80u;                              // bf16 (1, 1)
    const bool do_cs = p.colsum != nullptr && tk == 0 && wn == 0;     // (wave-uniform) the k-tile-0 blocks' left waves: every dY column once per split

    const int nt_all = (p.M_pad > 0 ? p.M_pad : p.M + TM - 1) / TM;      // (M_pad: the transposed-copy path's padded length -- same split boundaries)
    const int mt0 = (int)((long)nt_all * split / nsplit);
    const int nt = (int)((long)nt_all * (split + 1) / nsplit) - mt0;
    if (nt > 0) stage(mt0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();          // tile t landed for every wave; everyone is done reading the other buffer
        if (t + 1 < nt) stage(mt0 + t + 1, (t + 1) & 1);
        unsigned xa[4], wa[4];
        if (t & 1) {
#pragma unroll
            for (int b = 0; b < 4; ++b) { xa[b] = xa_s[1][b]; wa[b] = wa_s[1][b]; }
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b) { xa[b] = xa_s[0][b]; wa[b] = wa_s[0][b]; }
        }
        if (do_cs) {
            TN_STEP(0, 1024, TN_CS);        // m rows  0 .. 31 of the tile (second read of a fragment: + 4 rows = 1024 B)
            TN_STEP(8192, 9216, TN_CS);     // m rows 32 .. 63
        } else {
            TN_STEP(0, 1024, "");
            TN_STEP(8192, 9216, "");
        }
    }
    // ---- bias gradient partials: cs[i] holds this lane's share (its kg's 8 of every 32 m) of column n0 + 64 wm + 16 i + r: combine the four
    // kg lane groups in a fixed order, one slab per split (colsum[split][N]; the caller's finishing launch adds the slabs in order)
    if (do_cs) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = cs[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kg == 0) p.colsum[(long)split * p.N + n0 + wm * 64 + i * 16 + r] = v;
        }
    }

    // ---- epilogue: fp32 partial sums straight from the accumulator layout: acc[i][j][e] = C[n0 + 64 wm + 16 i + r][k0 + 64 wn + 16 j + 4 kg + e]
    float* ob = p.out + (long)split * p.split_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wm * 64 + i * 16 + r, k = k0 + wn * 64 + j * 16 + 4 * kg;
            *(float4*)(ob + (long)n * p.ldo + k) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
}


// ---- 256 (n) x 256 (k) x 64 (m) tiles, 8 waves (2 x 4, wave tile 128 x 64), one workgroup per CU.  The 128 x 128 form stages 32 KiB per
// 128 x 128 x 64 step: two co-resident workgroups ask the CU's L2 -> LDS path for ~140 GB/s where it delivers ~90 -- measured 54.7 us
// for the N = K = 1536, M = 8192 weight gradient against a 17 us MFMA floor.  This form moves half the bytes per FLOP (64 KiB per 256 x 256 x 64
// step, the ping-pong kernel's ratio).  LDS rows are 512 B (one LDS-DMA instruction = 2 rows); same chunk swizzle (the low 3 bits of the 32-byte
// chunk index); a k-step is TWO assembly blocks of 16 MFMAs -- inline asm takes at most 30 operands -- each reading its own four x fragments
// and the wave's four w fragments (re-read by the second block: nothing may be assumed to survive in registers between two asm statements).
// Same MFMA, operand order and m order per output element as the 128 x 128 form: bit-identical for equal split boundaries.  No column sums
// here (`colsum` must be null: the accumulators leave no registers for them; the caller takes the bias gradient with the slab kernel).
constexpr int OP_BYTES_B = TM * 512;          // 64 rows x 256 columns bf16
constexpr int STAGE_B = 2 * OP_BYTES_B;       // 64 KiB

__global__ __launch_bounds__(512) void gemm_tn256_kernel(GemmTnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;          // wave tile: output rows 128 wm .., output columns 64 wn ..

    const int ntn = p.N / 256, ntk = p.K / 256;
    const int nsplit = p.k_split > 1 ? p.k_split : 1;
    const int nblk = ntn * ntk * nsplit;
    int bid = blockIdx.x;
    {
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    const int split = bid / (ntn * ntk);
    bid -= split * (ntn * ntk);
    const int tn = bid / ntk, tk = bid - tn * ntk;
    const int n0 = tn * 256, k0 = tk * 256;

    // LDS-DMA: one instruction = 2 rows x 512 B; group g (32 per operand tile) belongs to wave g & 7; lane -> (row l >> 5, 16-byte chunk l & 31)
    unsigned soffA[4], soffB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = wave + 8 * i;
        const int row = g * 2 + (lane >> 5);
        const int c32 = ((lane & 31) >> 1) ^ swz_row(row);
        const int col = c32 * 16 + (lane & 1) * 8;
        soffA[i] = ((unsigned)row * (unsigned)p.lda + (unsigned)(n0 + col)) * 2u;
        soffB[i] = ((unsigned)row * (unsigned)p.ldb + (unsigned)(k0 + col)) * 2u;
    }
    auto stage = [&](int mt, int buf) {
        char* base = smem + buf * STAGE_B;
        const char* ga = (const char*)p.A + (long)mt * TM * p.lda * 2;
        const char* gb = (const char*)p.B + (long)mt * TM * p.ldb * 2;
        const bool ragged = (mt + 1) * TM > p.M;                            // (see gemm_tn_kernel)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = wave + 8 * i;
            const char* sa = ga + soffA[i];
            const char* sb = gb + soffB[i];
            if (ragged && mt * TM + g * 2 + (lane >> 5) >= p.M) { sa = (const char*)p.zeros; sb = (const char*)p.zeros; }
            __builtin_amdgcn_global_load_lds((gptr_t)sa, (lptr_t)(base + g * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)sb, (lptr_t)(base + OP_BYTES_B + g * 1024), 16, 0, 0);
        }
    };

    const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem);
    const int r = lane & 15, kg = lane >> 4;
    const int rrow = kg * 8 + (r >> 2);
    const int sw = swz_row(rrow);
    const unsigned xbase = lds0 + rrow * 512 + (r & 3) * 8, wbase = xbase + OP_BYTES_B;
    unsigned xa[8], wa[4];                            // stage 0 to start with
#pragma unroll
    for (int b = 0; b < 8; ++b) xa[b] = xbase + (((wm * 8 + b) ^ sw) << 5);
#pragma unroll
    for (int b = 0; b < 4; ++b) wa[b] = wbase + (((wn * 4 + b) ^ sw) << 5);

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt_all = (p.M_pad > 0 ? p.M_pad : p.M + TM - 1) / TM;      // (M_pad: the transposed-copy path's padded length -- same split boundaries)
    const int mt0 = (int)((long)nt_all * split / nsplit);
    const int nt = (int)((long)nt_all * (split + 1) / nsplit) - mt0;
    if (nt > 0) stage(mt0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) stage(mt0 + t + 1, (t + 1) & 1);
        // (xa / wa ARE the address registers: they flip between the two stages in place at the end of the iteration -- copies would not fit
        //  beside the 128 accumulator registers)
        // second read of a fragment: + 4 rows = 2048 B; second k-step: + 32 rows = 16384 B
        TN_STEP_NC(0, 2048, 0); TN_STEP_NC(0, 2048, 4);
        TN_STEP_NC(16384, 18432, 0); TN_STEP_NC(16384, 18432, 4);
        const unsigned flip = (t & 1) ? (unsigned)-STAGE_B : (unsigned)STAGE_B;
#pragma unroll
        for (int b = 0; b < 8; ++b) xa[b] += flip;
#pragma unroll
        for (int b = 0; b < 4; ++b) wa[b] += flip;
    }
    float* ob = p.out + (long)split * p.split_stride;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wm * 128 + i * 16 + r, k = k0 + wn * 64 + j * 16 + 4 * kg;
            *(float4*)(ob + (long)n * p.ldo + k) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
}

}  // namespace

// the block of zeros a ragged last m-tile reads (one per process: allocated on first use, never freed)
static const void* tn_zero_block() {
    static void* z = nullptr;
    if (!z) {
        if (hipMalloc(&z, 256) != hipSuccess) { z = nullptr; return nullptr; }
        // (hipMemset on device memory may return before the fill has run, and it runs on the NULL stream, which the engines' non-blocking streams
        //  do not wait for: synchronise once -- the first model-level run read the block before it was zero)
        if (hipMemset(z, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(z); z = nullptr; return nullptr; }
    }
    return z;
}

bool gemm_tn_ok(const GemmTnParams& p) {
    return p.M > 0 && (p.M_pad == 0 || (p.M_pad >= p.M && p.M_pad % TM == 0)) && p.N % 128 == 0 && p.K % 128 == 0 && (p.lda & 7) == 0 && (p.ldb & 7) == 0 && (p.ldo & 3) == 0 &&
           ((size_t)p.M * (size_t)p.lda) * 2 < (1ull << 32) && ((size_t)p.M * (size_t)p.ldb) * 2 < (1ull << 32) &&
           (((size_t)p.A | (size_t)p.B | (size_t)p.out) & 15) == 0;
}

static int g_wgrad_tn_mode = 1;
void set_wgrad_tn_mode(int v) { g_wgrad_tn_mode = v; }
int get_wgrad_tn_mode() { return g_wgrad_tn_mode; }

hipError_t launch_gemm_tn(const GemmTnParams& p_in, hipStream_t stream) {
    GemmTnParams p = p_in;
    if (!gemm_tn_ok(p) || !p.A || !p.B || !p.out) return hipErrorInvalidValue;
    if (p.M % TM != 0 || (p.M_pad > 0 && p.M_pad != p.M)) {      // ragged: the kernels need the zero block
        if (!p.zeros) p.zeros = tn_zero_block();
        if (!p.zeros) return hipErrorOutOfMemory;
    }
    if (sched_trace_on())
        sched_trace_launch("gemm_tn", stream, {treg(p.A, ((size_t)(p.M - 1) * p.lda + p.N) * 2), treg(p.B, ((size_t)(p.M - 1) * p.ldb + p.K) * 2)},
                           {treg(p.out, (((size_t)(p.N - 1) * p.ldo + p.K) + (size_t)(p.k_split > 1 ? p.k_split - 1 : 0) * p.split_stride) * 4),
                            treg(p.colsum, p.colsum ? (size_t)(p.k_split > 1 ? p.k_split : 1) * p.N * 4 : 0)});
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nsplit = p.k_split > 1 ? p.k_split : 1;
    if (p.tile256 && p.N % 256 == 0 && p.K % 256 == 0) {
        if (p.colsum) return hipErrorInvalidValue;
        static bool attr_b = false;
        if (!attr_b) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_tn256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_B);
            if (e != hipSuccess) return e;
            attr_b = true;
        }
        hipLaunchKernelGGL(gemm_tn256_kernel, dim3((p.N / 256) * (p.K / 256) * nsplit), dim3(512), 2 * STAGE_B, stream, p);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((p.N / 128) * (p.K / 128) * nsplit), dim3(256), 2 * STAGE, stream, p);
    return hipGetLastError();
}

}  // namespace mi355
