// mi355_flow -- helpers shared by the engine translation units (engine.hip, vae_engine.hip, flux_engine.hip, wan_engine.hip):
// error propagation into mi355_last_error() and the zero-initialised GEMM descriptor.
#pragma once
#include <string.h>

#include "kernels.h"

#define HIPCHK(x)                                                                                                            \
    do {                                                                                                                     \
        hipError_t _e = (x);                                                                                                 \
        if (_e != hipSuccess) return mi355::errorf("%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__);  \
    } while (0)
#define CHK(x)             \
    do {                   \
        int _r = (x);      \
        if (_r) return _r; \
    } while (0)

namespace mi355 {

// C[M][N] = A[M][K] . W[N][K]^T with epilogue `epi`; every optional field zero, rows_per_sample = M, eps = 1e-6
inline GemmParams make_gemm(const bf16_t* A, long lda, const bf16_t* W, long ldw, long M, int N, int K, int epi, const float* bias,
                            bf16_t* out, long ldo) {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = (int)M; g.N = N; g.K = K; g.epi = epi; g.bias = bias;
    g.out = out; g.ldo = ldo; g.rows_per_sample = M > 0 ? (int)M : 1; g.eps = 1e-6f;
    return g;
}

}  // namespace mi355
