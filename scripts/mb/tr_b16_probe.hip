#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
// probe of ds_read_b64_tr_b16: every lane supplies its own 8-byte-aligned address; print what each lane receives
__global__ void k(short* out, int mode) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int el;
    if (mode == 0) el = l * 4;                                   // lane-linear: 16 lanes cover one 128-byte row
    else if (mode == 1) el = (l & 15) * 64 + (l >> 4) * 4;       // lane i of a group at row i (64-element rows), group g at columns 4g..4g+3
    else el = ((l & 15) >> 2) * 64 + (l & 3) * 4 + (l >> 4) * 16; // 4 rows x 16 columns per group (rows 64 elements apart), groups side by side
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + el));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main() {
    short* d; (void)hipMalloc(&d, 512);
    for (int mode = 0; mode < 3; ++mode) {
        k<<<1, 64>>>(d, mode);
        short h[256]; (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
