// What is one s_memtime tick?  hipcc --offload-arch=gfx950 -O3 tick_calib.hip -o tick_calib
// (1) ticks per microsecond of wall time (HIP events around a long spin);
// (2) ticks taken by N back-to-back independent / dependent v_mfma_f32_16x16x32_bf16 on one wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__global__ void spin(unsigned long long ticks, unsigned long long *out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t = t0;
    while (t - t0 < ticks) { __builtin_amdgcn_s_sleep(16); t = __builtin_amdgcn_s_memtime(); }
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = t - t0;
}

template <int INDEP>
__global__ __launch_bounds__(256) void mfma_chain(int iters, unsigned long long *out, float *sink) {
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {2, 3, 4, 5, 6, 7, 8, 9};
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[INDEP ? i : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[INDEP ? i : 0], 0, 0, 0);
    }
    asm volatile("s_nop 15\ns_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = t1 - t0;
}

int main() {
    unsigned long long *d, h;
    float *sink;
    CHECK(hipMalloc(&d, 8));
    CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (unsigned long long ticks : {1000000ull, 10000000ull}) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, ticks, d);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
        printf("spin %llu ticks: %.3f ms wall -> %.1f ticks/us\n", h, ms, h / (ms * 1e3));
    }
    const int iters = 4000;
    for (int grid : {1, 256, 1024}) {
        hipLaunchKernelGGL(mfma_chain<1>, dim3(grid), dim3(256), 0, 0, iters, d, sink);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
        printf("grid %4d, 4 waves/CU-block, 8 independent accumulators: %.2f ticks per MFMA\n", grid, (double)h / (iters * 8.0));
        hipLaunchKernelGGL(mfma_chain<0>, dim3(grid), dim3(256), 0, 0, iters, d, sink);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
        printf("grid %4d, one dependent accumulator chain:               %.2f ticks per MFMA\n", grid, (double)h / (iters * 8.0));
    }
    return 0;
}
