// Transcendental-free GELU shared by the GEMM epilogues (gemm_bf16.hip) and the fused Swin MLP (swin_mlp.hip).
#pragma once
#include "common.h"

namespace vscgelu {
// GELU(x) = x Phi(x) without transcendentals, on packed fp32 (v_pk_fma_f32: two elements per instruction):
//     Phi(x) = 1/2 + t Q(z),   t = clamp(x, -U, U),   z = 2 t^2 / U^2 - 1 in [-1, 1],   U = 4.5,
// Q = the degree-8 polynomial fitted (Chebyshev nodes, weights t^2, then converted to powers of z -- the powers of t^2
// itself cancel badly in fp32) to (Phi(t) - 1/2) / t by tools/micro/gelu_poly_fit.py 4.5 8.  Error of the fp32 evaluation
// against erf in float64: |GELU error| <= 4.3e-5 + 3.5e-6 |x| (tests/test_gelu_poly.py) -- 1/25 of half a bf16 ulp of the
// stored value where the polynomial error peaks (|x| ~ 0.6).  Round 3: degree 11 -> 8 (it was 2.2e-6, forty times finer than
// anything downstream resolves): the vector pipe's time ADDS to the matrix pipe's on this part (tools/micro/pipe_overlap.hip:
// MFMAs and v_pk_fma_f32 of two waves of one SIMD take the sum of their times, not the maximum), so a write-out's
// instruction count is paid in full: 16 packed/scalar instructions per TWO elements (19 before).
constexpr int GELU_DEG = 8;
constexpr float GELU_U = 4.5f, GELU_ZS = 2.0f / (4.5f * 4.5f);
constexpr float GELU_C[GELU_DEG + 1] = {1.569020897e-01f, -7.717858255e-02f, 5.482625961e-02f, -4.047540203e-02f, 2.754251473e-02f,
                                        -1.697185636e-02f, 1.244884357e-02f, -9.152771905e-03f, 3.170517040e-03f};
__device__ __forceinline__ f32x2_t gelu2(f32x2_t x) {
    f32x2_t t;
    t[0] = __builtin_amdgcn_fmed3f(x[0], -GELU_U, GELU_U);
    t[1] = __builtin_amdgcn_fmed3f(x[1], -GELU_U, GELU_U);
    const f32x2_t z = __builtin_elementwise_fma(t * t, (f32x2_t){GELU_ZS, GELU_ZS}, (f32x2_t){-1.0f, -1.0f});
    f32x2_t q = (f32x2_t){GELU_C[GELU_DEG], GELU_C[GELU_DEG]};
#pragma unroll
    for (int i = GELU_DEG - 1; i >= 0; --i) q = __builtin_elementwise_fma(q, z, (f32x2_t){GELU_C[i], GELU_C[i]});
    return x * __builtin_elementwise_fma(t, q, (f32x2_t){0.5f, 0.5f});
}
__device__ __forceinline__ void gelu4(f32x4_t &v) {
    const f32x2_t lo = gelu2((f32x2_t){v[0], v[1]}), hi = gelu2((f32x2_t){v[2], v[3]});
    v = (f32x4_t){lo[0], lo[1], hi[0], hi[1]};
}
}  // namespace vscgelu
