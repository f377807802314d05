// Microbenchmark: cycles per v_mfma_f32_32x32x16_bf16 when K VALU "fillers" (softmax-like: exp2 / cvt_pk / add) are interleaved
// between the MFMAs of ONE wave, with 1, 2 or 4 waves per SIMD.  Registers only (no LDS / global traffic in the loop).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16v2;
typedef __attribute__((ext_vector_type(2))) float f32v2;

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    f32v2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16v2));
}

// FILL: number of softmax element-pairs processed per MFMA (each pair = 2 v_exp + 1 v_add(+1) + 1 v_cvt_pk ~ 5 issue slots)
template <int FILL, int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x % 3); b[i] = (short)(0x3f00 + i); }
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = (f32x16){0};
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = seed * (i + 1) * 1e-3f;
    float sum = 0.f;
    unsigned pk = 0;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % CHAINS], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < FILL; ++f) {
                const int i0 = (m * FILL + f) * 2 % 16;
                const float p0 = __builtin_amdgcn_exp2f(x[i0]);
                const float p1 = __builtin_amdgcn_exp2f(x[i0 + 1]);
                sum += p0 + p1;
                pk ^= pack_bf16(p0, p1);
                x[i0] = p0 * 1e-3f - 1.0f;        // keep the chain data dependent but cheap (1 fma each)
                x[i0 + 1] = p1 * 1e-3f - 1.0f;
            }
            if (FILL > 0) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, FILL * 7, 0);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = sum + (float)pk;
    for (int c = 0; c < CHAINS; ++c) r += acc[c][0] + acc[c][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int FILL, int CHAINS>
void run(int waves_per_simd, const char* tag) {
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (waves_per_simd workgroups of 4 waves)
    float* out; long long* cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k<FILL, CHAINS>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<FILL, CHAINS>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long* h = (long long*)malloc(blocks * 8);
    hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < blocks; ++i) avg += h[i]; avg /= blocks;
    const double mfma_per_wave = 16.0 * iters;
    const double tflops = (double)blocks * 4 * mfma_per_wave * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-28s waves/SIMD %d  fill %d pairs/MFMA (%2d VALU)  chains %d : %6.1f memtime-ticks/MFMA/wave  %7.1f TFLOP/s  (%.3f ms)\n", tag, waves_per_simd,
           FILL, FILL * 7, CHAINS, avg / mfma_per_wave, tflops, ms);
    hipFree(out); hipFree(cyc); free(h);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0, 2>(w, "mfma only");
        run<1, 2>(w, "softmax mix");
        run<2, 2>(w, "softmax mix");
        run<0, 4>(w, "mfma only, 4 chains");
        run<1, 4>(w, "softmax mix, 4 chains");
    }
    return 0;
}
