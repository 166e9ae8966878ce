// Do the matrix pipe and the vector pipe of a SIMD overlap?  One workgroup of 8 waves per CU (two waves per SIMD), register-only:
//   mode 0  every wave: M MFMAs (16x16x32 bf16, 16 independent accumulators)                         -> t_mfma
//   mode 1  every wave: V packed fp32 FMAs (16 independent chains)                                   -> t_valu
//   mode 2  every wave: both, interleaved in program order (one MFMA, V/M packed FMAs, ...)
//   mode 3  waves 0-3 (one per SIMD): the MFMAs, waves 4-7 (the other wave of each SIMD): the FMAs
//   mode 4  as 1 with scalar v_fma_f32 (2 V of them: the same arithmetic)
//   mode 5  as 2 with scalar v_fma_f32
// Prints cycles per iteration (s_memtime of wave 0): overlap shows as t(2), t(3) ~ max(t_mfma, t_valu), none as their sum.
// hipcc --offload-arch=gfx950 -O3 pipe_overlap.hip -o /tmp/pipe_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(int iters, float *sink, unsigned long long *ticks) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + ((threadIdx.x * 7 + i) & 63)); b[i] = (short)(0x3f00 + ((threadIdx.x * 3 + i) & 63)); }
    f32x4 acc[16];
    f32x2 q[16], z[16];
    for (int i = 0; i < 16; ++i) { acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; q[i] = (f32x2){1.f + i, 2.f}; z[i] = (f32x2){0.999f, 1.001f}; }
    constexpr bool SCALAR = MODE == 4 || MODE == 5;
    auto body = [&](auto dm, auto dv) {
        constexpr bool do_m = decltype(dm)::value, do_v = decltype(dv)::value;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {   // 4 rounds of (16 MFMAs, 16 x 4 packed FMAs): V / M = 4
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (do_m) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
                    if (do_v) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int c = (i * 4 + u) & 15;
                            if (SCALAR) {
                                q[c][0] = __builtin_fmaf(q[c][0], z[c][0], 0.25f);
                                q[c][1] = __builtin_fmaf(q[c][1], z[c][1], 0.25f);
                                asm volatile("" : "+v"(q[c][0]), "+v"(q[c][1]));
                            } else {
                                q[c] = __builtin_elementwise_fma(q[c], z[c], (f32x2){0.25f, 0.25f});
                                asm volatile("" : "+v"(q[c]));
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) body(std::true_type{}, std::false_type{});
    else if (MODE == 1 || MODE == 4) body(std::false_type{}, std::true_type{});
    else if (MODE == 2 || MODE == 5) body(std::true_type{}, std::true_type{});
    else if (wave < 4) body(std::true_type{}, std::false_type{});   // MODE 3: the role is fixed per wave, outside the loop
    else body(std::false_type{}, std::true_type{});
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + q[i][0] + q[i][1];
    if (s == 1.2345f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

template <int MODE>
void run(const char *what, int iters, float *sink, unsigned long long *ticks) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, iters, sink, ticks);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, iters, sink, ticks);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long t;
    CHECK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
    printf("mode %d %-58s %8.1f cycles / iteration (64 MFMAs and / or 256 packed FMAs per wave)   %.3f ms\n", MODE, what, (double)t / iters, ms);
}

int main() {
    float *sink; unsigned long long *ticks;
    CHECK(hipMalloc(&sink, 4)); CHECK(hipMalloc(&ticks, 8));
    const int iters = 2000;
    run<0>("MFMAs only (all 8 waves)", iters, sink, ticks);
    run<1>("packed FMAs only (all 8 waves)", iters, sink, ticks);
    run<2>("both, interleaved in every wave", iters, sink, ticks);
    run<3>("waves 0-3 MFMAs, waves 4-7 packed FMAs", iters, sink, ticks);
    run<4>("scalar FMAs only (2 per packed one)", iters, sink, ticks);
    run<5>("MFMAs + scalar FMAs interleaved in every wave", iters, sink, ticks);
    return 0;
}
