// Which bf16 MFMA shape gives more FLOPs per joule?  Register-only MFMA streams on every CU (no LDS, no memory), random
// bf16 operands, long enough for the power limit to settle; prints sustained TFLOP/s and the cycle-counted clock.
// hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 rnd(unsigned s) {
    bf16x8 v;
    for (int i = 0; i < 8; ++i) {
        s = s * 1664525u + 1013904223u;
        v[i] = (short)(0x3f00 | ((s >> 9) & 0x80ff));  // ~ +-[0.5, 1): random sign and mantissa
    }
    return v;
}

// 8 waves per CU (2 per SIMD), each: `iters` x 32 MFMAs on 32 independent 16x16 accumulators (128 VGPRs), 12 operand fragments
__global__ __launch_bounds__(512, 2) void k16(int iters, float *sink, unsigned long long *ticks) {
    bf16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = rnd(threadIdx.x * 97 + i * 13 + blockIdx.x);
    for (int j = 0; j < 4; ++j) b[j] = rnd(threadIdx.x * 31 + j * 7 + 5);
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0];
    if (s == 1.2345f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

// same FLOPs per iteration: 16 MFMAs of 32x32x16 on 8 independent 32x32 accumulators (128 VGPRs), two k-halves
__global__ __launch_bounds__(512, 2) void k32(int iters, float *sink, unsigned long long *ticks) {
    bf16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = rnd(threadIdx.x * 97 + i * 13 + blockIdx.x);
    for (int j = 0; j < 4; ++j) b[j] = rnd(threadIdx.x * 31 + j * 7 + 5);
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[2 * h + j], a[2 * i + h], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0];
    if (s == 1.2345f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

template <typename K>
void run(const char *name, K kern, double flops_per_iter_per_wave) {
    float *sink;
    unsigned long long *d, h;
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMalloc(&d, 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 60000, grid = 256;
    for (int rep = 0; rep < 3; ++rep) {   // the power limit needs a moment: report each repetition
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, iters, sink, d);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
        const double fl = flops_per_iter_per_wave * iters * 8.0 * grid;
        printf("%s rep %d: %.1f ms  %.0f TFLOP/s  clock %.2f GHz  (%.2f cycles per MFMA-equivalent of 16 kFLOP)\n", name, rep, ms,
               fl / (ms * 1e-3) / 1e12, h / (ms * 1e-3) / 1e9, (double)h / (iters * 32.0));
    }
}

int main() {
    run("v_mfma_f32_16x16x32_bf16", k16, 32.0 * 16384.0);
    run("v_mfma_f32_32x32x16_bf16", k32, 16.0 * 32768.0);
    return 0;
}
