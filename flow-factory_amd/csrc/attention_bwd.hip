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
//   attn_bwd_dkv_tr_kernel : one KEY per lane; loops over query tiles; S = Q K^T and dP = dO V^T arrive with lane = key and 8 consecutive
//                            queries per register group, so P and dZ feed  dV^T += dO^T . P  and  dK^T += Q^T . dZ  straight from registers;
//                            -L and -Delta enter the first two chains as their C operand (no subtraction is issued);
//   attn_bwd_dq_tr_kernel  : one QUERY per lane (the forward's orientation); S^T = K Q^T, dP^T = V dO^T, then  dQ^T += K^T . dZ^T.
// Round 3: the transposed A operands of the second products (dO^T, Q^T, K^T) are read out of the ROW-MAJOR tiles with
// `ds_read_b64_tr_b16` -- no transposed copies in HBM or LDS (see below).  v in [S_pad][64] comes from the forward's V^T by one HBM-bound
// transpose (backward.hip); -L | -Delta per 64-query tile from the prep kernel.
#include "kernels.h"
#include <type_traits>

namespace mi355 {
namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int TB = 64;                     // rows of the streamed dimension per tile
constexpr int TILE = TB * 64 * 2;          // 8 KiB
// NWAVES waves x 32 lanes-of-interest keys (pass 1) / queries (pass 2) per workgroup.  4 waves, two (three) workgroups per CU: 8-wave
// workgroups (one per CU, half the L2 -> LDS operand traffic: every workgroup streams ALL tiles of its (b, h)) measured 4-8 % SLOWER
// (profiles/r03u_attn_bwd_waves_ab.txt) -- independent barriers per CU overlap better than one.
constexpr int NWAVES = 4;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ int row_perm(int i) {     // MFMA output row i (0..31) -> row offset inside the 32-row block (attention.hip key_perm)
    const int a = i >> 3, g = (i >> 2) & 1, b = i & 3;
    return 16 * (a >> 1) + 8 * g + 4 * (a & 1) + b;
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

// =============================================================================================== the two passes
// The second product of each pass contracts over the STREAMED dimension
// (dV^T += dO^T . P and dK^T += Q^T . dZ over queries; dQ^T += K^T . dZ^T over keys), so its A operand is a column walk through a
// row-major tile: `ds_read_b64_tr_b16` delivers exactly that (per 16-lane group: a [4 rows][16 columns] block, lane i receives column
// i; checked lane by lane in scripts/mb/tr_b16_probe.hip).  Against round 2's kernels, which staged Q^T / dO^T (K^T) tiles from transposed
// copies in HBM: half (a third) of the L2 -> LDS traffic and of the LDS footprint, which pays for a 4-stage (3-stage) ring with loads two
// tiles ahead and counted waits instead of `vmcnt(0)` per tile, and no q^T / k^T / dO^T transpose launches.  Measured inside the
// optimize() step (profiles/r03v_attn_bwd_tr_ab.txt, joint + dual average): dK/dV pass 558 -> 492 us, dQ pass 383 -> 380 us (that pass was
// VALU-bound: see its C-operand splats below).
// Swizzle: physical 16-byte chunk = logical ^ swz2(row), swz2 = f ^ ((f & 1) << 2) with f = (row >> 1) & 7: still a permutation of the 8
// even (odd) rows a ds_read_b128 lane group touches (conflict-free as before), and rows r, r + 2 of a transposed read's 4-row block now
// sit in different 64-byte halves of their 128-byte rows (4 rows x 64 bytes = 64 distinct banks).
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s* lds_v4s_t;

__device__ __forceinline__ int swz2(int row) {
    const int f = (row >> 1) & 7;
    return f ^ ((f & 1) << 2);
}

template <int NWAVE>
__device__ __forceinline__ void stage_tile2(const bf16_t* src, long row_stride, char* dst, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 8 / NWAVE; ++i) {
        const int grp = wave + i * NWAVE;
        const int row = grp * 8 + (lane >> 3);
        const int c = (lane & 7) ^ swz2(row);
        __builtin_amdgcn_global_load_lds((gptr_t)(src + (long)row * row_stride + c * 8), (lptr_t)(dst + grp * 1024), 16, 0, 0);
    }
}

// The transposed reads and the MFMAs they feed are one inline-assembly block per 32-row half tile, with the fragments in FIXED registers:
//  * hipcc guards every C-level LDS read it knows to be an LDS read (the `ds_read_tr16_b64` builtin, address-space-3 pointers) with
//    `s_waitcnt vmcnt(0)` while any LDS-DMA is in flight -- it cannot tell the ring stages apart -- which would drain the tiles just put in
//    flight (the plain fragment reads go through generic pointers and are not guarded);
//  * a fragment is two 64-bit reads into the halves of one 4-register MFMA operand: as separate asm outputs they would need packing moves.
// Per-lane addresses a0..a3 = piece (db, jj) = (0,0), (0,1), (1,0), (1,1) of the 16-row block at the tile's row 32 half (+ hs * 2048 bytes).
// LDS returns in order, so `lgkmcnt(n)` retires all but the last n reads (compiler-issued fragment reads in flight only make it wait longer).
#define TR_RD(dst, a, off) "ds_read_b64_tr_b16 " dst ", " a " offset:" #off "\n\t"
#define MFMA32(acc, afrag, b) "v_mfma_f32_32x32x16_bf16 " acc ", " afrag ", " b ", " acc "\n\t"

__device__ __forceinline__ void wait_tiles_ahead(int ahead, int per_tile) {
    // `ahead` tiles were issued after the one about to be read, `per_tile` VM operations each (vmcnt retires in order)
    if (per_tile == 5) {
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// ---- pass 1: dK, dV.  LDS stage: Q [64 q][64 d] | dO [64 q][64 d] | -L[64] | -Delta[64]; ring of 4
constexpr int ST1T = 2 * TILE + 512;
constexpr int NST1 = 4;
__global__ __launch_bounds__(NWAVES * 64, 2) void attn_bwd_dkv_tr_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane & 31, lg = lane >> 5;
    constexpr int KB = 32 * NWAVES;
    const int nkb = (p.S + KB - 1) / KB;
    const int nwg = nkb * p.H * p.B;
    int wid = blockIdx.x;
    {
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = wid & 7;
        wid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (wid >> 3);
    }
    const int kblk = wid % nkb;
    const long bh = wid / nkb;
    const bf16_t* Qg = p.q + bh * p.S_pad * 64;
    const bf16_t* Og = p.doh + bh * p.S_pad * 64;
    // -L | -Delta of a 64-query tile are 128 contiguous floats in p.nld ([b h][tile][2][64]): wave w moves floats 32 w .. 32 w + 31 with
    // the first 8 lanes of one 16-byte LDS-DMA instruction (every wave: the same VM-operation count per tile, 5)
    const float* NLg = p.nld + bh * p.S_pad * 2 + wave * 32 + (lane & 7) * 4;
    const int key = kblk * KB + wave * 32 + lk;
    const int key_ld = key < p.S ? key : p.S - 1;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        kf[kk] = *(const bf16x8*)(p.k + (bh * p.S_pad + key_ld) * 64 + kk * 16 + lg * 8);
        vf[kk] = *(const bf16x8*)(p.v + (bh * p.S_pad + key_ld) * 64 + kk * 16 + lg * 8);
    }
    auto stage = [&](int t, int buf) {
        char* base = smem + buf * ST1T;
        stage_tile2<NWAVES>(Qg + (long)t * TB * 64, 64, base, wave, lane);
        stage_tile2<NWAVES>(Og + (long)t * TB * 64, 64, base + TILE, wave, lane);
        if (lane < 8) __builtin_amdgcn_global_load_lds((gptr_t)(NLg + (long)t * 2 * TB), (lptr_t)(base + 2 * TILE + wave * 128), 16, 0, 0);
    };
    // row-major tiles as A operand: row = 32*qb + perm(lk), logical chunk = 2*kk + lg
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
    f32x16 dk[2], dv[2];
    dk[0] = (f32x16){0}; dk[1] = (f32x16){0}; dv[0] = (f32x16){0}; dv[1] = (f32x16){0};
    const int nt = (p.S + TB - 1) / TB;
    stage(0, 0);
    if (nt > 1) stage(1, 1);
    if (nt > 2) stage(2, 2);
    // one half tile (32 queries): reads -> S / dP chains -> P, dZ -> second products.  Round 6: (1) ALL sixteen fragment / C-operand reads of the
    // half are issued by one asm statement and waited for once -- hipcc had paired every ds_read_b128 with an `s_waitcnt lgkmcnt(0)` in front of
    // its MFMA (eight exposed LDS round trips per half); (2) the query-tail mask is compiled into the LAST tile's copy of the body only (MASK) --
    // inside the loop it was a uniform branch behind every pair of v_exp, which also cut the schedule into sixteen regions.
    auto half = [&](int t, auto qb_c, auto mask_c) {
        constexpr int qb = decltype(qb_c)::value;
        constexpr bool MASK = decltype(mask_c)::value;
        f32x16 (&dv_)[2] = dv;      // (named here: an asm operand inside a generic lambda does not capture by itself)
        f32x16 (&dk_)[2] = dk;
        const unsigned stg = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + (t & (NST1 - 1)) * ST1T);
        const unsigned sLa = stg + 2 * TILE + 32 * lg;
        const unsigned r0 = stg + offR[0], r1 = stg + offR[1], r2 = stg + offR[2], r3 = stg + offR[3];
        f32x16 s, dp;
        bf16x8 qa[4], oa[4];
        {
            f32x4 a[4], b[4];
#define DKV_READS(L0, L1, L2, L3, D0, D1, D2, D3, QO, OO)                                                                                          \
            asm volatile("ds_read_b128 %0, %16 offset:" #L0 "\n\tds_read_b128 %1, %16 offset:" #L1 "\n\tds_read_b128 %2, %16 offset:" #L2 "\n\t"     \
                         "ds_read_b128 %3, %16 offset:" #L3 "\n\tds_read_b128 %4, %16 offset:" #D0 "\n\tds_read_b128 %5, %16 offset:" #D1 "\n\t"     \
                         "ds_read_b128 %6, %16 offset:" #D2 "\n\tds_read_b128 %7, %16 offset:" #D3 "\n\t"                                          \
                         "ds_read_b128 %8, %17 offset:" #QO "\n\tds_read_b128 %12, %17 offset:" #OO "\n\t"                                         \
                         "ds_read_b128 %9, %18 offset:" #QO "\n\tds_read_b128 %13, %18 offset:" #OO "\n\t"                                         \
                         "ds_read_b128 %10, %19 offset:" #QO "\n\tds_read_b128 %14, %19 offset:" #OO "\n\t"                                        \
                         "ds_read_b128 %11, %20 offset:" #QO "\n\tds_read_b128 %15, %20 offset:" #OO "\n\t"                                        \
                         "s_waitcnt lgkmcnt(0)"                                                                                                  \
                         : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]),                 \
                           "=&v"(qa[0]), "=&v"(qa[1]), "=&v"(qa[2]), "=&v"(qa[3]), "=&v"(oa[0]), "=&v"(oa[1]), "=&v"(oa[2]), "=&v"(oa[3])          \
                         : "v"(sLa), "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory")
            if constexpr (qb == 0) DKV_READS(0, 16, 64, 80, 256, 272, 320, 336, 0, 8192);
            else DKV_READS(128, 144, 192, 208, 384, 400, 448, 464, 4096, 12288);
#undef DKV_READS
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[4 * j + e] = a[j][e]; dp[4 * j + e] = b[j][e]; }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[kk], kf[kk], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa[kk], vf[kk], dp, 0, 0, 0);
        }
        unsigned pk[8], zk[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            float p0 = __builtin_amdgcn_exp2f(s[r]), p1 = __builtin_amdgcn_exp2f(s[r + 1]);
            if constexpr (MASK) {
                const int ql = 32 * qb + 16 * (r >> 3) + 8 * lg + (r & 7);
                if (t * TB + ql >= p.S) p0 = 0.f;
                if (t * TB + ql + 1 >= p.S) p1 = 0.f;
            }
            pk[r >> 1] = pack_bf16(p0, p1);
            zk[r >> 1] = pack_bf16(p0 * dp[r], p1 * dp[r + 1]);
        }
        // dV^T += dO^T . P and dK^T += Q^T . dZ over this half tile's 32 queries (two k-steps hs of 16): A fragments = transposed reads of the
        // dO tile (offset 8192) and the Q tile, v[224:239] for hs = 0, v[240:255] for hs = 1: [dO db0 | dO db1 | Q db0 | Q db1] x 4 registers
        {
            const bf16x8 pf0 = frag4(pk[0], pk[1], pk[2], pk[3]), pf1 = frag4(pk[4], pk[5], pk[6], pk[7]);
            const bf16x8 zf0 = frag4(zk[0], zk[1], zk[2], zk[3]), zf1 = frag4(zk[4], zk[5], zk[6], zk[7]);
            const unsigned a0 = stg + qb * 4096 + trb[0][0], a1 = stg + qb * 4096 + trb[0][1];
            const unsigned a2 = stg + qb * 4096 + trb[1][0], a3 = stg + qb * 4096 + trb[1][1];
            asm volatile(
                TR_RD("v[224:225]", "%[a0]", 8192) TR_RD("v[226:227]", "%[a1]", 8192) TR_RD("v[228:229]", "%[a2]", 8192) TR_RD("v[230:231]", "%[a3]", 8192)
                TR_RD("v[232:233]", "%[a0]", 0) TR_RD("v[234:235]", "%[a1]", 0) TR_RD("v[236:237]", "%[a2]", 0) TR_RD("v[238:239]", "%[a3]", 0)
                TR_RD("v[240:241]", "%[a0]", 10240) TR_RD("v[242:243]", "%[a1]", 10240) TR_RD("v[244:245]", "%[a2]", 10240) TR_RD("v[246:247]", "%[a3]", 10240)
                "s_waitcnt lgkmcnt(4)\n\t"
                MFMA32("%[dv0]", "v[224:227]", "%[pf0]") MFMA32("%[dk0]", "v[232:235]", "%[zf0]")
                TR_RD("v[248:249]", "%[a0]", 2048) TR_RD("v[250:251]", "%[a1]", 2048) TR_RD("v[252:253]", "%[a2]", 2048) TR_RD("v[254:255]", "%[a3]", 2048)
                MFMA32("%[dv1]", "v[228:231]", "%[pf0]") MFMA32("%[dk1]", "v[236:239]", "%[zf0]")
                "s_waitcnt lgkmcnt(0)\n\t"
                MFMA32("%[dv0]", "v[240:243]", "%[pf1]") MFMA32("%[dk0]", "v[248:251]", "%[zf1]")
                MFMA32("%[dv1]", "v[244:247]", "%[pf1]") MFMA32("%[dk1]", "v[252:255]", "%[zf1]")
                : [dv0] "+v"(dv_[0]), [dv1] "+v"(dv_[1]), [dk0] "+v"(dk_[0]), [dk1] "+v"(dk_[1])
                : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [pf0] "v"(pf0), [pf1] "v"(pf1), [zf0] "v"(zf0), [zf1] "v"(zf1)
                : "memory", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239",
                  "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
        }
    };
    using std::integral_constant;
    for (int t = 0; t < nt - 1; ++t) {
        wait_tiles_ahead(nt - 1 - t, 5);
        if (t + 3 < nt) stage(t + 3, (t + 3) & (NST1 - 1));      // its buffer held tile t - 1: every wave is past it (the barrier above)
        half(t, integral_constant<int, 0>{}, integral_constant<bool, false>{});
        half(t, integral_constant<int, 1>{}, integral_constant<bool, false>{});
    }
    wait_tiles_ahead(0, 5);                                      // the last tile, behind the loop (no accumulator merge inside it)
    half(nt - 1, integral_constant<int, 0>{}, integral_constant<bool, true>{});
    half(nt - 1, integral_constant<int, 1>{}, integral_constant<bool, true>{});
    __syncthreads();     // every wave is done with the ring: reuse it for the output transposes (4 KiB per wave)
    char* ob = smem + wave * 4096;
    const int row0 = kblk * KB + wave * 32;
    store_rows(dk, LN2, ob, p.dk + bh * p.S_pad * 64, row0, p.S, lane);
    store_rows(dv, 1.0f, ob, p.dv + bh * p.S_pad * 64, row0, p.S, lane);
}

// ---- pass 2: dQ.  LDS stage: K [64 keys][64 d] | V [64 keys][64 d]; ring of 4, two workgroups per CU
constexpr int ST2T = 2 * TILE;
constexpr int NST2 = 4;
__global__ __launch_bounds__(NWAVES * 64, 2) void attn_bwd_dq_tr_kernel(AttnBwdParams p) {
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
    const bf16_t* Kg = p.k + bh * p.S_pad * 64;
    const bf16_t* Vg = p.v + bh * p.S_pad * 64;
    const int q_row = qblk * QB + wave * 32 + lq;
    const int q_ld = q_row < p.S ? q_row : p.S - 1;
    bf16x8 qf[4], of[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        qf[kk] = *(const bf16x8*)(p.q + (bh * p.S_pad + q_ld) * 64 + kk * 16 + lg * 8);
        of[kk] = *(const bf16x8*)(p.doh + (bh * p.S_pad + q_ld) * 64 + kk * 16 + lg * 8);
    }
    // -L and -Delta of this lane's query as 16-register splats: the C operand of the first MFMA of every S^T / dP^T chain (vdst != src2),
    // so the chains deliver s - L and dP - Delta and the pass issues no subtraction (it is VALU-bound: 32 v_exp + 96 other VALU per 24 MFMAs
    // before, 32 + 32 now; the 32 registers cost the third wave per SIMD, which the VALU port could not feed anyway)
    f32x16 nL, nD;
    {
        const float l = -p.lse[bh * p.S_pad + q_ld], d = -p.delta[bh * p.S_pad + q_ld];
#pragma unroll
        for (int r = 0; r < 16; ++r) { nL[r] = l; nD[r] = d; }
    }
    auto stage = [&](int t, int buf) {
        char* base = smem + buf * ST2T;
        stage_tile2<NWAVES>(Kg + (long)t * TB * 64, 64, base, wave, lane);
        stage_tile2<NWAVES>(Vg + (long)t * TB * 64, 64, base + TILE, wave, lane);
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
    f32x16 dq[2];
    dq[0] = (f32x16){0}; dq[1] = (f32x16){0};
    const int nt = (p.S + TB - 1) / TB;
    stage(0, 0);
    if (nt > 1) stage(1, 1);
    if (nt > 2) stage(2, 2);
    // one half tile (32 keys); round 6: all eight fragment reads up front, one wait; the key-tail mask only in the last tile's copy (see pass 1)
    auto half = [&](int t, auto kb_c, auto mask_c) {
        constexpr int kb = decltype(kb_c)::value;
        constexpr bool MASK = decltype(mask_c)::value;
        f32x16 (&dq_)[2] = dq;      // (see pass 1)
        const unsigned stg = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + (t % NST2) * ST2T);
        const unsigned r0 = stg + offR[0], r1 = stg + offR[1], r2 = stg + offR[2], r3 = stg + offR[3];
        bf16x8 ka[4], va[4];
#define DQ_READS(KO, VO)                                                                                                                       \
        asm volatile("ds_read_b128 %0, %8 offset:" #KO "\n\tds_read_b128 %4, %8 offset:" #VO "\n\tds_read_b128 %1, %9 offset:" #KO "\n\t"            \
                     "ds_read_b128 %5, %9 offset:" #VO "\n\tds_read_b128 %2, %10 offset:" #KO "\n\tds_read_b128 %6, %10 offset:" #VO "\n\t"          \
                     "ds_read_b128 %3, %11 offset:" #KO "\n\tds_read_b128 %7, %11 offset:" #VO "\n\ts_waitcnt lgkmcnt(0)"                          \
                     : "=&v"(ka[0]), "=&v"(ka[1]), "=&v"(ka[2]), "=&v"(ka[3]), "=&v"(va[0]), "=&v"(va[1]), "=&v"(va[2]), "=&v"(va[3])              \
                     : "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory")
        if constexpr (kb == 0) DQ_READS(0, 8192);
        else DQ_READS(4096, 12288);
#undef DQ_READS
        f32x16 s, dp;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[kk], qf[kk], kk == 0 ? nL : s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[kk], of[kk], kk == 0 ? nD : dp, 0, 0, 0);
        }
        unsigned zk[8];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            float p0 = __builtin_amdgcn_exp2f(s[r]), p1 = __builtin_amdgcn_exp2f(s[r + 1]);
            if constexpr (MASK) {
                const int kl = t * TB + 32 * kb + 16 * (r >> 3) + 8 * lg + (r & 7);
                if (kl >= p.S) p0 = 0.f;
                if (kl + 1 >= p.S) p1 = 0.f;
            }
            zk[r >> 1] = pack_bf16(p0 * dp[r], p1 * dp[r + 1]);
        }
        // dQ^T += K^T . dZ^T over this half tile's 32 keys: A fragments = transposed reads of the K tile, v[224:231] (hs = 0), v[232:239] (hs = 1)
        {
            const bf16x8 zf0 = frag4(zk[0], zk[1], zk[2], zk[3]), zf1 = frag4(zk[4], zk[5], zk[6], zk[7]);
            const unsigned a0 = stg + kb * 4096 + trb[0][0], a1 = stg + kb * 4096 + trb[0][1];
            const unsigned a2 = stg + kb * 4096 + trb[1][0], a3 = stg + kb * 4096 + trb[1][1];
            asm volatile(
                TR_RD("v[224:225]", "%[a0]", 0) TR_RD("v[226:227]", "%[a1]", 0) TR_RD("v[228:229]", "%[a2]", 0) TR_RD("v[230:231]", "%[a3]", 0)
                TR_RD("v[232:233]", "%[a0]", 2048) TR_RD("v[234:235]", "%[a1]", 2048) TR_RD("v[236:237]", "%[a2]", 2048) TR_RD("v[238:239]", "%[a3]", 2048)
                "s_waitcnt lgkmcnt(4)\n\t"
                MFMA32("%[dq0]", "v[224:227]", "%[zf0]") MFMA32("%[dq1]", "v[228:231]", "%[zf0]")
                "s_waitcnt lgkmcnt(0)\n\t"
                MFMA32("%[dq0]", "v[232:235]", "%[zf1]") MFMA32("%[dq1]", "v[236:239]", "%[zf1]")
                : [dq0] "+v"(dq_[0]), [dq1] "+v"(dq_[1])
                : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [zf0] "v"(zf0), [zf1] "v"(zf1)
                : "memory", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239");
        }
    };
    using std::integral_constant;
    for (int t = 0; t < nt - 1; ++t) {
        wait_tiles_ahead(nt - 1 - t, 4);
        if (t + 3 < nt) stage(t + 3, (t + 3) % NST2);
        half(t, integral_constant<int, 0>{}, integral_constant<bool, false>{});
        half(t, integral_constant<int, 1>{}, integral_constant<bool, false>{});
    }
    wait_tiles_ahead(0, 4);
    half(nt - 1, integral_constant<int, 0>{}, integral_constant<bool, true>{});
    half(nt - 1, integral_constant<int, 1>{}, integral_constant<bool, true>{});
    __syncthreads();
    char* ob = smem + wave * 4096;
    store_rows(dq, LN2, ob, p.dq + bh * p.S_pad * 64, qblk * QB + wave * 32, p.S, lane);
}

// =============================================================================================== round 6: software-pipelined passes
// The same two passes with their main loops as ONE hand-scheduled asm statement each (attn_bwd64_asm.inc, generated by gen_attn_bwd64.py: its
// docstring has the schedule): the second products of half h - 1, the exp / scale / pack of half h and the S / dP chains of half h + 1 are
// interleaved MFMA by MFMA, so a single wave keeps the matrix pipe fed.  Same arithmetic in the same order per output element: bit-identical to
// the kernels above (kept: mi355_tune_set(43, 0), and the A/B of tests/test_gpu_backward.py).
// CONTRACT (both pairs of kernels relied on it already -- a masked probability times a NaN is a NaN): rows [S, S_pad) of q, k, v and dO are
// ZERO.  The engines' workspaces are zero-initialised and only rows < S are ever written; the pipelined loops carry no tail masks at all (a zero
// query row adds exactly 0 to dK^T / dV^T, a zero key row exactly 0 to dQ^T).
#include "attn_bwd64_asm.inc"

// ABL (timing only, results garbage; mi355_tune_set(43, 2..5) under MI355_ALLOW_ABLATION=1): 1 no exp / scale / pack, 2 no LDS reads, 3 no barrier /
// waits / tile loads, 4 MFMAs only
template <int ABL>
__global__ __launch_bounds__(NWAVES * 64, 2) void attn_bwd_dkv_pipe_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane & 31, lg = lane >> 5;
    constexpr int KB = 32 * NWAVES;
    const int nkb = (p.S + KB - 1) / KB;
    const int nwg = nkb * p.H * p.B;
    int wid = blockIdx.x;
    {
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = wid & 7;
        wid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (wid >> 3);
    }
    const int kblk = wid % nkb;
    const long bh = wid / nkb;
    if ((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();      // the asm addresses the ring from LDS byte 0
    const bf16_t* Qg = p.q + bh * p.S_pad * 64;
    const bf16_t* Og = p.doh + bh * p.S_pad * 64;
    const float* NLt = p.nld + bh * p.S_pad * 2;
    const int key = kblk * KB + wave * 32 + lk;
    const int key_ld = key < p.S ? key : p.S - 1;
    const int nt = (p.S + TB - 1) / TB;
    // LDS-DMA: this lane's 16-byte piece of rows 8 w + (l >> 3) (and + 32) of a 64 x 64 tile; 16 bytes of the tile's -L | -Delta floats
    const int srow = wave * 8 + (lane >> 3);
    const unsigned g0 = (unsigned)(srow * 128 + (((lane & 7) ^ swz2(srow)) << 4));
    const unsigned g2 = (unsigned)(wave * 128 + (lane & 7) * 16);
    const unsigned grow = (unsigned)(key_ld * 128 + lg * 16);
    auto stage = [&](int t, int buf) {                   // (tiles 0..2; the loop stages the rest)
        const int tt = t < nt ? t : nt - 1;
        char* base = smem + buf * ST1T;
        stage_tile2<NWAVES>(Qg + (long)tt * TB * 64, 64, base, wave, lane);
        stage_tile2<NWAVES>(Og + (long)tt * TB * 64, 64, base + TILE, wave, lane);
        if (lane < 8) __builtin_amdgcn_global_load_lds((gptr_t)(NLt + (long)tt * 2 * TB + wave * 32 + (lane & 7) * 4), (lptr_t)(base + 2 * TILE + wave * 128), 16, 0, 0);
    };
    const int prow = row_perm(lk);
    unsigned la = (unsigned)(2 * TILE + 32 * lg), r0, a0, a1;
    {
        r0 = (unsigned)(prow * 128 + ((lg ^ swz2(prow)) << 4));                  // fragment kk = 0 (pass 1's offR[0])
        const int i = lane & 15, g1 = (lane >> 4) & 1;
        unsigned a[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {                                         // pieces (db = 0, jj) (pass 1's trb[0][jj])
            const int ql = 8 * lg + 4 * jj + (i >> 2);
            const int ch = 2 * g1 + ((i & 3) >> 1);
            a[jj] = (unsigned)(ql * 128 + ((ch ^ swz2(ql)) << 4) + (i & 1) * 8);
        }
        a0 = a[0]; a1 = a[1];
    }
    stage(0, 0); stage(1, 1); stage(2, 2);
    const int lt = nt - 1 < 3 ? nt - 1 : 3;
    const unsigned long long b0 = (unsigned long long)(Qg + (long)lt * TB * 64), b1 = (unsigned long long)(Og + (long)lt * TB * 64);
    const unsigned long long b2 = (unsigned long long)(NLt + (long)lt * 2 * TB);
    const unsigned long long p0 = (unsigned long long)(p.k + bh * p.S_pad * 64), p1 = (unsigned long long)(p.v + bh * p.S_pad * 64);
#define DKV_OPERANDS                                                                                                                          \
                 : [la] "+v"(la), [r0] "+v"(r0), [a0] "+v"(a0), [a1] "+v"(a1)       /* (r1..r3, a2, a3 are r0 ^ 32 kk, a0 / a1 ^ 64: formed at their use) */ \
                 : [g0] "v"(g0), [g2] "v"(g2), [grow] "v"(grow), [b0] "s"(b0), [b1] "s"(b1), [b2] "s"(b2), [p0] "s"(p0), [p1] "s"(p1), [nt] "s"(nt), \
                   [wv] "s"(wave)                                                                                                                \
                 : ABWD64_DKV_CLOBBERS
    if constexpr (ABL == 0) asm volatile(ABWD64_DKV_ASM DKV_OPERANDS);
    else if constexpr (ABL == 1) asm volatile(ABWD64_DKV_ASM_NOVALU DKV_OPERANDS);
    else if constexpr (ABL == 2) asm volatile(ABWD64_DKV_ASM_NOLDS DKV_OPERANDS);
    else if constexpr (ABL == 3) asm volatile(ABWD64_DKV_ASM_NOSYNC DKV_OPERANDS);
    else asm volatile(ABWD64_DKV_ASM_MFMAONLY DKV_OPERANDS);
#undef DKV_OPERANDS
    f32x16 dk[2], dv[2];
    ABWD64_READ_ACC_0(dv[0]) ABWD64_READ_ACC_16(dk[0]) ABWD64_READ_ACC_32(dv[1]) ABWD64_READ_ACC_48(dk[1])
    __syncthreads();     // every wave is done with the ring (and its LDS-DMA writes: the asm ends with vmcnt(0)): reuse it for the output transposes
    // the epilogue's lane arithmetic hangs off an OPAQUE copy of the lane id: hipcc would otherwise form those addresses in front of the loop and
    // keep them alive across it, where sixteen VGPRs are all there is (one more register and the kernel drops to one wave per SIMD)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    char* ob = smem + wave * 4096;
    const int row0 = kblk * KB + wave * 32;
    store_rows(dk, LN2, ob, p.dk + bh * p.S_pad * 64, row0, p.S, lane_e);
    store_rows(dv, 1.0f, ob, p.dv + bh * p.S_pad * 64, row0, p.S, lane_e);
}

__global__ __launch_bounds__(NWAVES * 64, 2) void attn_bwd_dq_pipe_kernel(AttnBwdParams p) {
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
    const bf16_t* Kg = p.k + bh * p.S_pad * 64;
    const bf16_t* Vg = p.v + bh * p.S_pad * 64;
    const int q_row = qblk * QB + wave * 32 + lq;
    const int q_ld = q_row < p.S ? q_row : p.S - 1;
    const int nt = (p.S + TB - 1) / TB;
    const int srow = wave * 8 + (lane >> 3);
    const unsigned g0 = (unsigned)(srow * 128 + (((lane & 7) ^ swz2(srow)) << 4));
    const unsigned grow = (unsigned)(q_ld * 128 + lg * 16);
    const float nl = -p.lse[bh * p.S_pad + q_ld], nd = -p.delta[bh * p.S_pad + q_ld];
    auto stage = [&](int t, int buf) {
        const int tt = t < nt ? t : nt - 1;
        char* base = smem + buf * ST2T;
        stage_tile2<NWAVES>(Kg + (long)tt * TB * 64, 64, base, wave, lane);
        stage_tile2<NWAVES>(Vg + (long)tt * TB * 64, 64, base + TILE, wave, lane);
    };
    const int prow = row_perm(lq);
    unsigned r0, r1, r2, r3, a0, a1, a2, a3;
    {
        unsigned r[4], a[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) r[kk] = (unsigned)(prow * 128 + (((2 * kk + lg) ^ swz2(prow)) << 4));
        const int i = lane & 15, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int kl = 8 * lg + 4 * jj + (i >> 2);
                const int ch = 4 * db + 2 * g1 + ((i & 3) >> 1);
                a[db * 2 + jj] = (unsigned)(kl * 128 + ((ch ^ swz2(kl)) << 4) + (i & 1) * 8);
            }
        r0 = r[0]; r1 = r[1]; r2 = r[2]; r3 = r[3]; a0 = a[0]; a1 = a[1]; a2 = a[2]; a3 = a[3];
    }
    stage(0, 0); stage(1, 1); stage(2, 2);
    const int lt = nt - 1 < 3 ? nt - 1 : 3;
    const unsigned long long b0 = (unsigned long long)(Kg + (long)lt * TB * 64), b1 = (unsigned long long)(Vg + (long)lt * TB * 64);
    const unsigned long long p0 = (unsigned long long)(p.q + bh * p.S_pad * 64), p1 = (unsigned long long)(p.doh + bh * p.S_pad * 64);
    asm volatile(ABWD64_DQ_ASM
                 : [r0] "+v"(r0), [r1] "+v"(r1), [r2] "+v"(r2), [r3] "+v"(r3), [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3)
                 : [g0] "v"(g0), [grow] "v"(grow), [nl] "v"(nl), [nd] "v"(nd), [b0] "s"(b0), [b1] "s"(b1), [p0] "s"(p0), [p1] "s"(p1), [nt] "s"(nt),
                   [wv] "s"(wave)
                 : ABWD64_DQ_CLOBBERS);
    f32x16 dq[2];
    ABWD64_READ_ACC_0(dq[0]) ABWD64_READ_ACC_16(dq[1])
    __syncthreads();
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    char* ob = smem + wave * 4096;
    store_rows(dq, LN2, ob, p.dq + bh * p.S_pad * 64, qblk * QB + wave * 32, p.S, lane_e);
}

// ---- two 32-row blocks per wave, one wave per SIMD (attn_bwd64x2_asm.inc <- gen_attn_bwd64x2.py: its docstring has the why)
#include "attn_bwd64x2_asm.inc"

__global__ __launch_bounds__(NWAVES * 64, 1) void attn_bwd_dkv_pipe2_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lk = lane & 31, lg = lane >> 5;
    constexpr int KB = 64 * NWAVES;
    const int nkb = (p.S + KB - 1) / KB;
    const int nwg = nkb * p.H * p.B;
    int wid = blockIdx.x;
    {
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = wid & 7;
        wid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (wid >> 3);
    }
    const int kblk = wid % nkb;
    const long bh = wid / nkb;
    if ((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem != 0u) __builtin_trap();
    const bf16_t* Qg = p.q + bh * p.S_pad * 64;
    const bf16_t* Og = p.doh + bh * p.S_pad * 64;
    const float* NLt = p.nld + bh * p.S_pad * 2;
    const int key0 = kblk * KB + wave * 64 + lk, key1 = key0 + 32;
    const int kl0 = key0 < p.S ? key0 : p.S - 1, kl1 = key1 < p.S ? key1 : p.S - 1;
    const int nt = (p.S + TB - 1) / TB;
    const int srow = wave * 8 + (lane >> 3);
    const unsigned g0 = (unsigned)(srow * 128 + (((lane & 7) ^ swz2(srow)) << 4));
    const unsigned g2 = (unsigned)(wave * 128 + (lane & 7) * 16);
    const unsigned grow0 = (unsigned)(kl0 * 128 + lg * 16), grow1 = (unsigned)(kl1 * 128 + lg * 16);
    auto stage = [&](int t, int buf) {
        const int tt = t < nt ? t : nt - 1;
        char* base = smem + buf * ST1T;
        stage_tile2<NWAVES>(Qg + (long)tt * TB * 64, 64, base, wave, lane);
        stage_tile2<NWAVES>(Og + (long)tt * TB * 64, 64, base + TILE, wave, lane);
        if (lane < 8) __builtin_amdgcn_global_load_lds((gptr_t)(NLt + (long)tt * 2 * TB + wave * 32 + (lane & 7) * 4), (lptr_t)(base + 2 * TILE + wave * 128), 16, 0, 0);
    };
    const int prow = row_perm(lk);
    unsigned la = (unsigned)(2 * TILE + 32 * lg), r0, a0, a1;
    {
        r0 = (unsigned)(prow * 128 + ((lg ^ swz2(prow)) << 4));
        const int i = lane & 15, g1 = (lane >> 4) & 1;
        unsigned a[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int ql = 8 * lg + 4 * jj + (i >> 2);
            const int ch = 2 * g1 + ((i & 3) >> 1);
            a[jj] = (unsigned)(ql * 128 + ((ch ^ swz2(ql)) << 4) + (i & 1) * 8);
        }
        a0 = a[0]; a1 = a[1];
    }
    stage(0, 0); stage(1, 1); stage(2, 2);
    const int lt = nt - 1 < 3 ? nt - 1 : 3;
    const unsigned long long b0 = (unsigned long long)(Qg + (long)lt * TB * 64), b1 = (unsigned long long)(Og + (long)lt * TB * 64);
    const unsigned long long b2 = (unsigned long long)(NLt + (long)lt * 2 * TB);
    const unsigned long long p0 = (unsigned long long)(p.k + bh * p.S_pad * 64), p1 = (unsigned long long)(p.v + bh * p.S_pad * 64);
    asm volatile(ABWD64X2_DKV2_ASM
                 : [la] "+v"(la), [r0] "+v"(r0), [a0] "+v"(a0), [a1] "+v"(a1)
                 : [g0] "v"(g0), [g2] "v"(g2), [grow0] "v"(grow0), [grow1] "v"(grow1), [b0] "s"(b0), [b1] "s"(b1), [b2] "s"(b2), [p0] "s"(p0),
                   [p1] "s"(p1), [nt] "s"(nt), [wv] "s"(wave)
                 : ABWD64X2_DKV2_CLOBBERS);
    f32x16 dk0[2], dv0[2], dk1[2], dv1[2];
    ABWD64X2_READ_ACC_0(dv0[0]) ABWD64X2_READ_ACC_16(dv0[1]) ABWD64X2_READ_ACC_32(dk0[0]) ABWD64X2_READ_ACC_48(dk0[1])
    ABWD64X2_READ_ACC_64(dv1[0]) ABWD64X2_READ_ACC_80(dv1[1]) ABWD64X2_READ_ACC_96(dk1[0]) ABWD64X2_READ_ACC_112(dk1[1])
    __syncthreads();
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    char* ob = smem + wave * 4096;
    const int row0 = kblk * KB + wave * 64;
    store_rows(dk0, LN2, ob, p.dk + bh * p.S_pad * 64, row0, p.S, lane_e);
    store_rows(dv0, 1.0f, ob, p.dv + bh * p.S_pad * 64, row0, p.S, lane_e);
    store_rows(dk1, LN2, ob, p.dk + bh * p.S_pad * 64, row0 + 32, p.S, lane_e);
    store_rows(dv1, 1.0f, ob, p.dv + bh * p.S_pad * 64, row0 + 32, p.S, lane_e);
}

__global__ __launch_bounds__(NWAVES * 64, 1) void attn_bwd_dq_pipe2_kernel(AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, lg = lane >> 5;
    constexpr int QB = 64 * NWAVES;
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
    const bf16_t* Kg = p.k + bh * p.S_pad * 64;
    const bf16_t* Vg = p.v + bh * p.S_pad * 64;
    const int q0 = qblk * QB + wave * 64 + lq, q1 = q0 + 32;
    const int ql0 = q0 < p.S ? q0 : p.S - 1, ql1 = q1 < p.S ? q1 : p.S - 1;
    const int nt = (p.S + TB - 1) / TB;
    const int srow = wave * 8 + (lane >> 3);
    const unsigned g0 = (unsigned)(srow * 128 + (((lane & 7) ^ swz2(srow)) << 4));
    const unsigned grow0 = (unsigned)(ql0 * 128 + lg * 16), grow1 = (unsigned)(ql1 * 128 + lg * 16);
    const float nl0 = -p.lse[bh * p.S_pad + ql0], nd0 = -p.delta[bh * p.S_pad + ql0];
    const float nl1 = -p.lse[bh * p.S_pad + ql1], nd1 = -p.delta[bh * p.S_pad + ql1];
    auto stage = [&](int t, int buf) {
        const int tt = t < nt ? t : nt - 1;
        char* base = smem + buf * ST2T;
        stage_tile2<NWAVES>(Kg + (long)tt * TB * 64, 64, base, wave, lane);
        stage_tile2<NWAVES>(Vg + (long)tt * TB * 64, 64, base + TILE, wave, lane);
    };
    const int prow = row_perm(lq);
    unsigned r0, a0, a1, a2, a3;
    {
        r0 = (unsigned)(prow * 128 + ((lg ^ swz2(prow)) << 4));
        const int i = lane & 15, g1 = (lane >> 4) & 1;
        unsigned a[4];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int kl = 8 * lg + 4 * jj + (i >> 2);
                const int ch = 4 * db + 2 * g1 + ((i & 3) >> 1);
                a[db * 2 + jj] = (unsigned)(kl * 128 + ((ch ^ swz2(kl)) << 4) + (i & 1) * 8);
            }
        a0 = a[0]; a1 = a[1]; a2 = a[2]; a3 = a[3];
    }
    stage(0, 0); stage(1, 1); stage(2, 2);
    const int lt = nt - 1 < 3 ? nt - 1 : 3;
    const unsigned long long b0 = (unsigned long long)(Kg + (long)lt * TB * 64), b1 = (unsigned long long)(Vg + (long)lt * TB * 64);
    const unsigned long long p0 = (unsigned long long)(p.q + bh * p.S_pad * 64), p1 = (unsigned long long)(p.doh + bh * p.S_pad * 64);
    asm volatile(ABWD64X2_DQ2_ASM
                 : [r0] "+v"(r0), [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3)
                 : [g0] "v"(g0), [grow0] "v"(grow0), [grow1] "v"(grow1), [nl0] "v"(nl0), [nd0] "v"(nd0), [nl1] "v"(nl1), [nd1] "v"(nd1), [b0] "s"(b0),
                   [b1] "s"(b1), [p0] "s"(p0), [p1] "s"(p1), [nt] "s"(nt), [wv] "s"(wave)
                 : ABWD64X2_DQ2_CLOBBERS);
    f32x16 dqa[2], dqb[2];
    ABWD64X2_READ_ACC_0(dqa[0]) ABWD64X2_READ_ACC_16(dqa[1]) ABWD64X2_READ_ACC_32(dqb[0]) ABWD64X2_READ_ACC_48(dqb[1])
    __syncthreads();
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    char* ob = smem + wave * 4096;
    store_rows(dqa, LN2, ob, p.dq + bh * p.S_pad * 64, qblk * QB + wave * 64, p.S, lane_e);
    store_rows(dqb, LN2, ob, p.dq + bh * p.S_pad * 64, qblk * QB + wave * 64 + 32, p.S, lane_e);
}

}  // namespace

static int g_attn_bwd_pipe = 1;      // mi355_tune_set(43, .): 1 = the software-pipelined passes (round 6), 0 = the round-3 kernels
void set_attn_bwd_pipe(int v) { g_attn_bwd_pipe = v; }

hipError_t launch_attention_bwd(const AttnBwdParams& p, hipStream_t stream) {
    if (sched_trace_on()) {
        const size_t bhs = (size_t)p.B * p.H * p.S_pad;
        sched_trace_launch("attention_bwd", stream, {treg(p.q, bhs * 128), treg(p.k, bhs * 128), treg(p.v, bhs * 128), treg(p.doh, bhs * 128), treg(p.lse, bhs * 4),
                                                     treg(p.delta, bhs * 4), treg(p.nld, bhs * 8)},
                           {treg(p.dq, bhs * 128), treg(p.dk, bhs * 128), treg(p.dv, bhs * 128)});
    }
    if (p.S <= 0 || p.S_pad % TB != 0 || p.S_pad < p.S || !p.nld) return hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dkv_tr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NST1 * ST1T);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nb = (p.S + 32 * NWAVES - 1) / (32 * NWAVES);
    if (g_attn_bwd_pipe == 6) {      // two 32-row blocks per wave, one workgroup per CU
        static bool attr_set3 = false;
        if (!attr_set3) {
            hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dkv_pipe2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NST1 * ST1T);
            if (e != hipSuccess) return e;
            attr_set3 = true;
        }
        const int nb2 = (p.S + 64 * NWAVES - 1) / (64 * NWAVES);
        hipLaunchKernelGGL(attn_bwd_dkv_pipe2_kernel, dim3(nb2 * p.H * p.B), dim3(NWAVES * 64), NST1 * ST1T, stream, p);
        hipLaunchKernelGGL(attn_bwd_dq_pipe2_kernel, dim3(nb2 * p.H * p.B), dim3(NWAVES * 64), NST2 * ST2T, stream, p);
        return hipGetLastError();
    }
    if (g_attn_bwd_pipe) {
        void (*dkv)(AttnBwdParams) = g_attn_bwd_pipe == 2 ? attn_bwd_dkv_pipe_kernel<1> : g_attn_bwd_pipe == 3 ? attn_bwd_dkv_pipe_kernel<2>
                                     : g_attn_bwd_pipe == 4 ? attn_bwd_dkv_pipe_kernel<3> : g_attn_bwd_pipe == 5 ? attn_bwd_dkv_pipe_kernel<4>
                                                                                                                  : attn_bwd_dkv_pipe_kernel<0>;
        static bool attr_set2[6] = {false, false, false, false, false, false};      // per instantiation (66 KiB of dynamic LDS)
        const int vi = g_attn_bwd_pipe >= 1 && g_attn_bwd_pipe <= 5 ? g_attn_bwd_pipe : 1;
        if (!attr_set2[vi]) {
            hipError_t e = hipFuncSetAttribute((const void*)dkv, hipFuncAttributeMaxDynamicSharedMemorySize, NST1 * ST1T);
            if (e != hipSuccess) return e;
            attr_set2[vi] = true;
        }
        hipLaunchKernelGGL(dkv, dim3(nb * p.H * p.B), dim3(NWAVES * 64), NST1 * ST1T, stream, p);
        hipLaunchKernelGGL(attn_bwd_dq_pipe_kernel, dim3(nb * p.H * p.B), dim3(NWAVES * 64), NST2 * ST2T, stream, p);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(attn_bwd_dkv_tr_kernel, dim3(nb * p.H * p.B), dim3(NWAVES * 64), NST1 * ST1T, stream, p);
    hipLaunchKernelGGL(attn_bwd_dq_tr_kernel, dim3(nb * p.H * p.B), dim3(NWAVES * 64), NST2 * ST2T, stream, p);
    return hipGetLastError();
}

}  // namespace mi355
