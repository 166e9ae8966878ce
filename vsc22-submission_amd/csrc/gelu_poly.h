// Transcendental-free GELU shared by the GEMM epilogues (gemm_bf16.hip) and the fused Swin MLP (swin_mlp.hip).
#pragma once
#include "common.h"

namespace vscgelu {
// GELU(x) = x Phi(x) without transcendentals, on packed fp32 (v_pk_fma_f32: two elements per instruction):
//     Phi(x) = 1/2 + t Q(z),   t = clamp(x, -5, 5),   z = 0.08 t^2 - 1 in [-1, 1],
// Q = the degree-11 polynomial fitted (Chebyshev nodes, weights t^2, then converted to powers of z -- the powers of t^2
// itself cancel badly in fp32) to (Phi(t) - 1/2) / t by tools/micro/gelu_poly_fit.py.  Error of the fp32 evaluation against
// erf in float64: |GELU error| <= 2.2e-6 + 6.6e-7 |x| (tests/test_gelu_poly.py) -- under half a bf16 ulp of the stored
// value wherever |GELU| > 2e-3.  17 packed/scalar instructions per TWO elements against ~19 per ONE (two of them
// quarter-rate: v_rcp_f32, v_exp_f32) for the Abramowitz-Stegun 7.1.26 form used before: the fc1 write-out was
// VALU-bound on it (18 k of the 48 k cycles a tile took).
__device__ __forceinline__ f32x2_t gelu2(f32x2_t x) {
    f32x2_t t;
    t[0] = __builtin_amdgcn_fmed3f(x[0], -5.0f, 5.0f);
    t[1] = __builtin_amdgcn_fmed3f(x[1], -5.0f, 5.0f);
    const f32x2_t z = __builtin_elementwise_fma(t * t, (f32x2_t){0.08f, 0.08f}, (f32x2_t){-1.0f, -1.0f});
    constexpr float C[12] = {1.413637698e-01f, -7.029826939e-02f, 5.152343214e-02f, -4.038983583e-02f, 3.137785569e-02f,
                             -2.364724688e-02f, 1.683344319e-02f, -1.008572429e-02f, 5.223751534e-03f, -4.000799730e-03f,
                             3.139984794e-03f, -1.040469273e-03f};
    f32x2_t q = (f32x2_t){C[11], C[11]};
#pragma unroll
    for (int i = 10; i >= 0; --i) q = __builtin_elementwise_fma(q, z, (f32x2_t){C[i], C[i]});
    return x * __builtin_elementwise_fma(t, q, (f32x2_t){0.5f, 0.5f});
}
// NP independent pairs in lock-step: every Horner step is issued for all pairs before the next one, so the dependent
// v_pk_fma_f32 of one pair are NP instructions apart (one pair at a time the chain runs at its latency, ~2.5 x slower: the
// GEMM write-outs interleave their pairs with LDS and store traffic, a register-resident caller has only this)
template <int NP>
__device__ __forceinline__ void gelu_pairs(f32x2_t (&x)[NP]) {
    constexpr float C[12] = {1.413637698e-01f, -7.029826939e-02f, 5.152343214e-02f, -4.038983583e-02f, 3.137785569e-02f,
                             -2.364724688e-02f, 1.683344319e-02f, -1.008572429e-02f, 5.223751534e-03f, -4.000799730e-03f,
                             3.139984794e-03f, -1.040469273e-03f};
    f32x2_t t[NP], z[NP], q[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        t[p][0] = __builtin_amdgcn_fmed3f(x[p][0], -5.0f, 5.0f);
        t[p][1] = __builtin_amdgcn_fmed3f(x[p][1], -5.0f, 5.0f);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) z[p] = __builtin_elementwise_fma(t[p] * t[p], (f32x2_t){0.08f, 0.08f}, (f32x2_t){-1.0f, -1.0f});
#pragma unroll
    for (int p = 0; p < NP; ++p) q[p] = __builtin_elementwise_fma((f32x2_t){C[11], C[11]}, z[p], (f32x2_t){C[10], C[10]});
#pragma unroll
    for (int i = 9; i >= 0; --i) {
#pragma unroll
        for (int p = 0; p < NP; ++p) q[p] = __builtin_elementwise_fma(q[p], z[p], (f32x2_t){C[i], C[i]});
        __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise re-serialises the chains, depth first)
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) x[p] = x[p] * __builtin_elementwise_fma(t[p], q[p], (f32x2_t){0.5f, 0.5f});
}
__device__ __forceinline__ void gelu4(f32x4_t &v) {
    const f32x2_t lo = gelu2((f32x2_t){v[0], v[1]}), hi = gelu2((f32x2_t){v[2], v[3]});
    v = (f32x4_t){lo[0], lo[1], hi[0], hi[1]};
}
}  // namespace vscgelu
