// mi355_flow -- shared device helpers (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355 {

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(2))) __bf16 bf16v2;
typedef __attribute__((ext_vector_type(2))) float f32v2;

enum DType : int { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };

__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// two floats -> packed bf16x2 (round-to-nearest-even; lowers to v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    f32v2 v = {lo, hi};
    bf16v2 r = __builtin_convertvector(v, bf16v2);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf16(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float round_bf16(float f) { return bf2f(f2bf(f)); }

// 1/x by v_rcp_f32 (1 ulp): every use below is followed by a bf16 rounding (2^-9), an IEEE division would cost ~10 VALU ops
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float silu_f(float x) { return x * fast_rcp(1.0f + __expf(-x)); }
// GELU(tanh approximation), as torch.nn.functional.gelu(approximate="tanh"):
//   0.5 x (1 + tanh(u)) = x * sigmoid(2u),  u = sqrt(2/pi) (x + 0.044715 x^3)
// evaluated as x / (1 + 2^(-x (c0 + c1 x^2))) with log2(e) folded into the constants: 5 VALU + v_exp + v_rcp per element
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float c0 = 2.0f * 0.7978845608028654f * 1.4426950408889634f;
    const float c1 = c0 * 0.044715f;
    const float t = x * (c0 + c1 * x * x);
    return x * fast_rcp(1.0f + __builtin_amdgcn_exp2f(-t));
}

// d/dx of gelu_tanh_f: s + x s (1 - s) 2u'(x), s = sigmoid(2u), 2u = x (k0 + k1 x^2), (2u)' = k0 + 3 k1 x^2
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
    const float k0 = 2.0f * 0.7978845608028654f, k1 = k0 * 0.044715f;
    const float x2 = x * x;
    const float s = fast_rcp(1.0f + __expf(-x * (k0 + k1 * x2)));
    return s + x * s * (1.0f - s) * (k0 + 3.0f * k1 * x2);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// storage-dtype load/store of a latent element (fp32 / bf16 / fp16)
__device__ __forceinline__ float load_as_f32(const void* p, long i, int dt) {
    if (dt == DT_F32) return ((const float*)p)[i];
    if (dt == DT_BF16) return bf2f(((const bf16_t*)p)[i]);
    return (float)(((const _Float16*)p)[i]);
}
// value-round an fp32 to the storage dtype (the reference's `.to(dtype).float()`), with the
// fp16 clamp of cast_latents (reference models/abc.py:172-182)
__device__ __forceinline__ float round_to_dtype(float v, int dt) {
    if (dt == DT_F32) return v;
    if (dt == DT_BF16) return round_bf16(v);
    v = fminf(fmaxf(v, -65504.0f), 65504.0f);
    return (float)((_Float16)v);
}
__device__ __forceinline__ void store_from_f32(void* p, long i, int dt, float v) {
    if (dt == DT_F32) ((float*)p)[i] = v;
    else if (dt == DT_BF16) ((bf16_t*)p)[i] = f2bf(v);
    else ((_Float16*)p)[i] = (_Float16)fminf(fmaxf(v, -65504.0f), 65504.0f);
}

}  // namespace mi355
