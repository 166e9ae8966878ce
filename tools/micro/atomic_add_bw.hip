// Micro-benchmark: in-place fp32 residual update x += v three ways, on a [rows, 768] fp32 matrix walked the way a GEMM
// write-out walks it (each wave owns 64-column row segments of a 128-row tile):
//   rmw     float4 load + add + float4 store (what the RESADD epilogue does)
//   atomic  global_atomic_add_f32 without return, one dword per lane, 256-B row segments per instruction
//   store   float4 store only (lower bound for the CU side)
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o atomic_add_bw atomic_add_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(float *x, int rows, int n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles_n = n / 256;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const long m0 = (long)tm * 256 + wm * 128;
    const int n0 = tn * 256 + wn * 64;
    if (MODE == 1) {
        for (int r = 0; r < 128; ++r) {
            const long m = m0 + r;
            if (m < rows) __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float *)(x + m * n + n0 + lane), 1.0f + r);
        }
    } else {
        for (int r = 0; r < 128; r += 4) {
            const long m = m0 + r + (lane >> 4);
            float4 *p = (float4 *)(x + m * n + n0 + (lane & 15) * 4);
            if (m < rows) {
                float4 v = make_float4(1.f + r, 1.f + r, 1.f + r, 1.f + r);
                if (MODE == 0) { float4 o = *p; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *p = v;
            }
        }
    }
}

int main(int argc, char **argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 65404, n = argc > 2 ? atoi(argv[2]) : 768;
    float *x;
    CK(hipMalloc(&x, (size_t)rows * n * 4));
    CK(hipMemset(x, 0, (size_t)rows * n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = ((rows + 255) / 256) * (n / 256);
    const char *names[3] = {"rmw", "atomic", "store"};
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipEventRecord(e0));
            for (int it = 0; it < 10; ++it) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, x, rows, n);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, x, rows, n);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, x, rows, n);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-7s %8.1f us/launch  %7.1f GB/s (matrix bytes x%d)\n", names[mode], ms * 100, (double)rows * n * 4 * (mode == 2 ? 1 : 2) / (ms / 10 * 1e-3) / 1e9, mode == 2 ? 1 : 2);
        }
    float h[4]; CK(hipMemcpy(h, x + 5 * n + 3, 16, hipMemcpyDeviceToHost));
    printf("check %g\n", h[0]);
    return 0;
}
