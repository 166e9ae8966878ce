// Can a light kernel of stream B run on CUs that a persistent kernel of stream A occupies with 8 waves x VREGS registers and
// LDS_A bytes of LDS?  A spins for ~2 ms on every CU; B (64-thread workgroups, ~20 VGPRs, no LDS) streams 200 MB.  If B's event
// time is close to its time alone, the two were co-resident; if it is ~A's duration, B waited for A to leave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/_bin/coresident_probe tools/micro/coresident_probe.hip && tools/micro/_bin/coresident_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int VREGS>
__global__ __launch_bounds__(512, 2) void occupy(unsigned long long cycles, float *sink) {
    extern __shared__ char lds[];
    float r[VREGS - 4];
#pragma unroll
    for (int i = 0; i < VREGS - 4; ++i) r[i] = threadIdx.x * 0.5f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) {
#pragma unroll
        for (int i = 0; i < VREGS - 4; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(r[i]));
        __builtin_amdgcn_s_sleep(8);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VREGS - 4; ++i) s += r[i];
    if (s == 12345.678f) sink[0] = s + lds[threadIdx.x];
}

__global__ __launch_bounds__(64) void stream_rows(const float4 *x, float4 *y, int per_row) {
    const float4 *p = x + (size_t)blockIdx.x * per_row * 64 + threadIdx.x;
    float4 *q = y + (size_t)blockIdx.x * per_row * 64 + threadIdx.x;
    float4 v[3];
    for (int i = 0; i < 3; ++i) v[i] = p[i * 64];
    for (int i = 0; i < 3; ++i) { v[i].x += 1.f; q[i * 64] = v[i]; }
}

template <int VREGS>
int run(int lds_a, const char *tag) {
    float *sink; CK(hipMalloc(&sink, 4));
    const int rows = 65404, per_row = 3;
    float4 *x, *y;
    CK(hipMalloc(&x, (size_t)rows * per_row * 64 * 16)); CK(hipMalloc(&y, (size_t)rows * per_row * 64 * 16));
    CK(hipMemset(x, 0, (size_t)rows * per_row * 64 * 16));
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t a0, a1, b0, b1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    CK(hipFuncSetAttribute((const void *)occupy<VREGS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
    // B alone
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(b0, b)); hipLaunchKernelGGL(stream_rows, dim3(rows), dim3(64), 0, b, x, y, per_row); CK(hipEventRecord(b1, b));
        CK(hipStreamSynchronize(b));
    }
    float alone = 0; CK(hipEventElapsedTime(&alone, b0, b1));
    // A (2 ms at ~2 GHz = 4e6 cycles of the 100 MHz s_memtime?  s_memtime counts shader cycles here) and B inside it
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(a0, a));
        hipLaunchKernelGGL(occupy<VREGS>, dim3(256), dim3(512), lds_a, a, 4000000ull, sink);
        CK(hipEventRecord(a1, a));
        CK(hipEventRecord(b0, b)); hipLaunchKernelGGL(stream_rows, dim3(rows), dim3(64), 0, b, x, y, per_row); CK(hipEventRecord(b1, b));
        CK(hipDeviceSynchronize());
    }
    float ta = 0, tb = 0, skew = 0;
    CK(hipEventElapsedTime(&ta, a0, a1)); CK(hipEventElapsedTime(&tb, b0, b1)); CK(hipEventElapsedTime(&skew, a0, b1));
    printf("%s: occupier %d VGPR request, %d KB LDS: A %.3f ms | B alone %.3f ms, B beside A %.3f ms (B ended %.3f ms after A started)\n", tag, VREGS,
           lds_a >> 10, ta, alone, tb, skew);
    return 0;
}

int main() {
    run<232>(130 * 1024, "like v4<0>");
    run<200>(130 * 1024, "200 regs ");
    run<128>(130 * 1024, "128 regs ");
    run<128>(32 * 1024, "128 regs, small LDS");
    run<250>(130 * 1024, "like v4<3>");
    return 0;
}
