// calibrates s_memtime (clock64) and s_memrealtime (wall_clock64) against hipEvent time, idle and beside MFMA load
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void spin(long long* out, int iters, int mfma) {
    const long long t0 = clock64(), w0 = wall_clock64();
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int it = 0; it < iters; ++it) {
        if (mfma) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        } else {
            __builtin_amdgcn_s_sleep(10);
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
    if (acc[0][0] == 12345.f) out[2] = 1;
}
int main() {
    long long* d; hipMalloc(&d, 64);
    long long h[3];
    for (int mode = 0; mode < 2; ++mode) {
        const int grid = mode ? 256 * 8 : 1, iters = mode ? 2000000 : 2000000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        spin<<<grid, 256>>>(d, 1000, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        spin<<<grid, 256>>>(d, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("mode %s: event %.3f ms | clock64 delta %lld -> %.1f MHz | wall_clock64 delta %lld -> %.1f MHz", mode ? "MFMA x2048 WGs" : "idle 1 WG", ms,
               h[0], h[0] / ms / 1e3, h[1], h[1] / ms / 1e3);
        if (mode) printf(" | %.1f TFLOP/s (8 indep. 16x16x32 per wave, 4 waves/WG)", 2048.0 * 4 * 8 * 16384.0 * iters / (ms * 1e-3) / 1e12);
        printf("\n");
    }
    return 0;
}
