// mi355_flow -- EXPERIMENTAL bf16 GEMM main loop for gfx950 (A/B candidate against gemm_pp_kernel; unit-test entry point only, not on the
// rollout path yet).  Motivation (profiles/r02_power_clock_notes.txt): the ping-pong kernel is power-capped (the package sits at its ~1.35 kW
// limit at an effective 1.44 GHz) and hipBLASLt delivers 10-20 % more FLOP/s at the same socket power, i.e. fewer joules per FLOP.  Two
// sources of energy per FLOP are structural in gemm_pp_kernel and are removed here:
//
//   * per-wave register tile 128 x 128 (4 waves x 1 per SIMD) instead of 128 x 64 (8 waves x 2 per SIMD): LDS bytes read per MFMA-FLOP
//     drop by a third (per K = 64 tile: 4 waves x 32 KiB = 128 KiB instead of 8 x 24 KiB = 192 KiB);
//   * v_mfma_f32_32x32x16_bf16 instead of 16x16x32: half the operand-register reads per FLOP (a 32 x 32 x 16 product reads 1024 operand
//     elements for 32 768 FLOP, a 16 x 16 x 32 one reads 1024 for 16 384).
//
// Structure: 256 x 256 x 64 block tile, 256 threads, 1 workgroup per CU; operands HBM/L2 -> LDS by global_load_lds_dwordx4 into the
// same XOR-swizzled 128-byte rows as gemm.hip (16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7): conflict-free
// ds_read_b128 for 32-row fragments, as in attention.hip); 2 LDS stages (128 KiB); ONE barrier per K-tile, placed before the LAST k-step:
//
//     step 0..2 : ds_read fragments of step s+1 | 16 MFMA of step s
//     vmcnt(0) + lgkmcnt(0) + s_barrier        -> tile t+1 has landed in the other stage AND every wave has read its last fragments
//                                                 of tile t, so stage (t & 1) is free
//     issue the 16 global_load_lds of tile t+2 into stage (t & 1)
//     step 3    : ds_read fragments of (tile t+1, step 0) | 16 MFMA of step 3
//
// so the prefetch distance is a full K-tile and no fragment read is ever exposed behind the barrier.  MFMA operands are swapped like
// gemm.hip (A-operand = W rows, B-operand = activation rows): a lane owns one output row and groups of 4 consecutive columns.
// Epilogue of this experiment: + bias, direct 8-byte stores (no LDS staging yet).  Requires M % 256 == 0, N % 256 == 0, K % 64 == 0.
#include "kernels.h"

namespace mi355 {
namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 256, BN = 256, BKW = 64;
constexpr int A_BYTES = BM * BKW * 2;       // 32 KiB
constexpr int STAGE = 2 * A_BYTES;          // A tile + W tile

struct W4Params {
    const bf16_t* A; const bf16_t* W; const float* bias; bf16_t* out;
    int M, N, K;
};

__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(W4Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;          // 2 x 2 waves, 128 x 128 each
    const int l31 = lane & 31, lg = lane >> 5;

    // XCD-aware tile order: workgroup id w runs on XCD (w % 8); give each XCD a contiguous range of tiles (row-major: one A row panel
    // and consecutive W panels stay in one private L2)
    const int ntn = p.N / BN, ntm = p.M / BM;
    const int nwg = ntn * ntm;
    int wid = blockIdx.x;
    {
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = wid & 7;
        wid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (wid >> 3);
    }
    const int tm = wid / ntn, tn = wid - tm * ntn;
    const bf16_t* Ag = p.A + (long)tm * BM * p.K;
    const bf16_t* Wg = p.W + (long)tn * BN * p.K;

    // ---- staging: a 256-row operand tile = 32 groups of 8 rows (1 KiB each); wave w stages groups w, w+4, ..., of A and of W.
    // lane -> row (lane >> 3) of the group, 16-byte chunk (lane & 7) of the LDS image = source chunk (lane & 7) ^ ((row >> 1) & 7)
    const bf16_t* srcA[8];
    const bf16_t* srcW[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (wave + 4 * i) * 8 + (lane >> 3);
        const int sc = (lane & 7) ^ ((row >> 1) & 7);
        srcA[i] = Ag + (long)row * p.K + sc * 8;
        srcW[i] = Wg + (long)row * p.K + sc * 8;
    }
    auto stage = [&](int t, int buf) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(srcA[i] + (long)t * BKW), (lptr_t)(base + (wave + 4 * i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(srcW[i] + (long)t * BKW), (lptr_t)(base + A_BYTES + (wave + 4 * i) * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets: operand row r = 32 * blk + l31, k-step s: logical chunk 2 s + lg, stored at chunk ^ ((r >> 1) & 7)
    // (rows + 32: (r >> 1) & 7 unchanged, byte offset + 4096)
    int offX[4], offW_[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int sw = (((2 * s + lg) ^ ((l31 >> 1) & 7)) << 4);
        offX[s] = (wm * 128 + l31) * 128 + sw;                 // activation rows (B operand): output rows
        offW_[s] = A_BYTES + (wn * 128 + l31) * 128 + sw;      // weight rows (A operand): output columns
    }

    f32x16 acc[4][4];            // [column block j (W rows)][row block i (activation rows)]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = (f32x16){0};

    bf16x8 fx[2][4], fw[2][4];   // double-buffered fragments: [parity][block]
    auto load_frags = [&](int par, const char* sb, int s) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            fw[par][b] = *(const bf16x8*)(sb + offW_[s] + b * 4096);
            fx[par][b] = *(const bf16x8*)(sb + offX[s] + b * 4096);
        }
    };
    auto mfma_step = [&](int par) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[par][j], fx[par][i], acc[j][i], 0, 0, 0);
    };

    // scheduling directives (LLVM SchedGroupMask: MFMA = 0x8, VMEM read = 0x20, DS read = 0x100): one LDS fragment read (and, in the last
    // k-step, one global_load_lds) issued behind each MFMA -- the wave is alone on its SIMD, so the overlap has to be in its own stream
#define W4_INTERLEAVE_DS8()                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    \
    }                                                                         \
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0)
    const int nt = p.K / BKW;
    stage(0, 0);
    stage(nt > 1 ? 1 : 0, 1);
    // tile 0 must have landed before its first fragments are read (tile 1's 16 loads may still be in flight: counted wait)
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __syncthreads();
    load_frags(0, smem, 0);

    for (int t = 0; t < nt; ++t) {
        const char* sb = smem + (t & 1) * STAGE;
        const char* sn = smem + ((t + 1) & 1) * STAGE;
        const int tl = t + 2 < nt ? t + 2 : nt - 1;     // (the last two iterations re-load the last tile into a stage nobody reads again)
        load_frags(1, sb, 1);
        mfma_step(0);
        W4_INTERLEAVE_DS8();
        load_frags(0, sb, 2);
        mfma_step(1);
        W4_INTERLEAVE_DS8();
        load_frags(1, sb, 3);
        mfma_step(0);
        W4_INTERLEAVE_DS8();
        // tile t+1 complete (this wave's loads: vmcnt(0); the other waves': the barrier) and stage (t & 1) free (every wave holds its
        // step-3 fragments in registers: lgkmcnt(0) before the barrier).  The 16 MFMAs of step 2 are in the pipe while the wave waits.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");      // (the barrier builtin is IntrNoMem: keep the LDS traffic below it)
        __builtin_amdgcn_sched_barrier(0);
        stage(tl, t & 1);
        load_frags(0, sn, 0);
        mfma_step(1);
#pragma unroll
        for (int q_ = 0; q_ < 8; ++q_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#pragma unroll
        for (int q_ = 0; q_ < 8; ++q_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef W4_INTERLEAVE_DS8
    // the redundant tail loads target this workgroup's LDS: they must have landed before the LDS can be handed to another workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: + bias, bf16, direct stores.  acc[j][i] reg r: column n = 32 j + 8 (r >> 2) + 4 lg + (r & 3), row m = 32 i + l31
    const long m0 = (long)tm * BM + wm * 128;
    const int n0 = tn * BN + wn * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int n = n0 + 32 * j + 8 * a + 4 * lg;
            const float4 bb = *(const float4*)(p.bias + n);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long m = m0 + 32 * i + l31;
                uint2 w = {pack_bf16(acc[j][i][4 * a] + bb.x, acc[j][i][4 * a + 1] + bb.y),
                           pack_bf16(acc[j][i][4 * a + 2] + bb.z, acc[j][i][4 * a + 3] + bb.w)};
                *(uint2*)(p.out + m * p.N + n) = w;
            }
        }
}

}  // namespace

// unit-test / A-B entry point: out[M][N] (bf16, ld = N) = A[M][K] . W[N][K]^T + bias[N]
hipError_t launch_gemm_w4(const bf16_t* A, const bf16_t* W, const float* bias, bf16_t* out, int M, int N, int K, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || M % BM || N % BN || K % BKW) return hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_w4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    W4Params p{A, W, bias, out, M, N, K};
    hipLaunchKernelGGL(gemm_w4_kernel, dim3((M / BM) * (N / BN)), dim3(256), 2 * STAGE, st, p);
    return hipGetLastError();
}

}  // namespace mi355
