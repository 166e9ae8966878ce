// Operand-delivery micro-benchmark (run on the GPU box):  hipcc --offload-arch=gfx950 -O3 fill_bench.hip -o fill_bench
// Streams the bf16 GEMM's tile traffic (256 A rows + 256 W rows x 32 k per step, row stride K) into LDS with
//   mode 0: LDS-DMA (global_load_lds, 16 B per lane), 4-stage ring, counted vmcnt + one barrier per step
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128, two steps in flight, 3-stage ring, one barrier per step
// and nothing else (no MFMA, one ds_read per step so the writes are live).  Prints bytes/clk/CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int ROWB, int THREADS = 512>  // ROWB: bytes of one tile row per step (64 = BK 32, 128 = BK 64)
__global__ __launch_bounds__(THREADS, 1) void fill_kernel(const char *a, const char *w, int64_t kbytes, int nk, int tiles_n, u32x4 *sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int ROWS = 512, STAGE = ROWS * ROWB, STAGES = ROWB == 128 ? 2 : (MODE == 0 ? 4 : 3);
    constexpr int PIECES = STAGE / (THREADS * 16);  // 16-B pieces per thread per step
    constexpr int NWV = THREADS / 64;
    constexpr int LPR = ROWB / 16;              // lanes per row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const char *src[PIECES];
    int dst[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
        const int e = (p * NWV + wave) * 64 + lane;  // piece p of this wave: 64 lanes -> 64/LPR rows
        const int row = e / LPR, c = e % LPR;
        const char *base = row < 256 ? a + (int64_t)(tm * 256 + row) * kbytes : w + (int64_t)(tn * 256 + row - 256) * kbytes;
        src[p] = base + c * 16;
        dst[p] = row * ROWB + ((c ^ ((row >> 1) & (LPR - 1))) << 4);
    }
    u32x4 acc = {0, 0, 0, 0};
    if (MODE == 0) {
        auto issue = [&](int kt, int st) {
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                char *l = lds + st * STAGE + ((p * NWV + wave) * 64) * 16;  // wave-uniform base, lane * 16 implied
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[p] + (int64_t)kt * ROWB),
                                                 (__attribute__((address_space(3))) void *)l, 16, 0, 0);
            }
        };
        for (int s = 0; s < STAGES - 1; ++s) issue(s, s);
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
            if (kt + STAGES - 1 < nk) vmcnt<PIECES *(STAGES - 1)>(); else vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            acc ^= *(const u32x4 *)(lds + (kt % STAGES) * STAGE + tid * 16);
            __builtin_amdgcn_s_barrier();
        }
    } else if (THREADS == 512) {
        u32x4 g[2][PIECES];
        auto load = [&](int kt, int b) {
#pragma unroll
            for (int p = 0; p < PIECES; ++p) g[b][p] = *(const u32x4 *)(src[p] + (int64_t)kt * ROWB);
        };
        auto store = [&](int st, int b) {
#pragma unroll
            for (int p = 0; p < PIECES; ++p) *(u32x4 *)(lds + st * STAGE + dst[p]) = g[b][p];
        };
        load(0, 0);
        load(1, 1);
        vmcnt<PIECES>();
        store(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {  // nk even
            if (kt + 2 < nk) load(kt + 2, 0);
            if (kt + 2 < nk) vmcnt<PIECES>(); else vmcnt<0>();
            store((kt + 1) % STAGES, 1);
            acc ^= *(const u32x4 *)(lds + (kt % STAGES) * STAGE + tid * 16);
            __builtin_amdgcn_s_barrier();
            if (kt + 3 < nk) load(kt + 3, 1);
            if (kt + 2 < nk) {
                if (kt + 3 < nk) vmcnt<PIECES>(); else vmcnt<0>();
                store((kt + 2) % STAGES, 0);
            }
            acc ^= *(const u32x4 *)(lds + ((kt + 1) % STAGES) * STAGE + tid * 16);
            __builtin_amdgcn_s_barrier();
        }
    }
    if (acc[0] == 0x12345678 && acc[1] == 0x9abcdef0) sink[blockIdx.x * THREADS + tid] = acc;
}

template <int MODE, int ROWB, int THREADS = 512>
void run(const char *name, const char *a, const char *w, int m, int n, int k, u32x4 *sink, double mhz) {
    const int tiles_m = m / 256, tiles_n = n / 256, nk = k * 2 / ROWB;
    const int smem = (ROWB == 128 ? 2 : (MODE == 0 ? 4 : 3)) * 512 * ROWB;
    CHECK(hipFuncSetAttribute((const void *)fill_kernel<MODE, ROWB, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((fill_kernel<MODE, ROWB, THREADS>), dim3(tiles_m * tiles_n), dim3(THREADS), smem, 0, a, w, (int64_t)k * 2, nk, tiles_n, sink);
    CHECK(hipEventRecord(e0));
    const int it = 10;
    for (int i = 0; i < it; ++i)
        hipLaunchKernelGGL((fill_kernel<MODE, ROWB, THREADS>), dim3(tiles_m * tiles_n), dim3(THREADS), smem, 0, a, w, (int64_t)k * 2, nk, tiles_n, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / it, bytes = (double)tiles_m * tiles_n * 512.0 * k * 2;
    printf("%-28s m=%d n=%d k=%d: %8.1f us  %7.2f TB/s into LDS  %6.1f B/clk/CU  (GEMM-equivalent %.0f TF/s)\n", name, m, n, k, us,
           bytes / us / 1e6, bytes / us / 1e6 * 1e12 / 256 / (mhz * 1e6), 2.0 * m * n * k / us / 1e6);
}

int main() {
    const int m = 8192, n = 8192, kmax = 4096;
    char *a, *w;
    u32x4 *sink;
    CHECK(hipMalloc(&a, (size_t)m * kmax * 2));
    CHECK(hipMalloc(&w, (size_t)n * kmax * 2));
    CHECK(hipMalloc(&sink, (size_t)1024 * 512 * 16));
    CHECK(hipMemset(a, 1, (size_t)m * kmax * 2));
    CHECK(hipMemset(w, 2, (size_t)n * kmax * 2));
    int khz = 0;
    CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double mhz = khz / 1e3;
    printf("clock %.0f MHz\n", mhz);
    for (int k : {768, 4096}) {
        run<0, 64>("lds-dma  64-B rows", a, w, m, n, k, sink, mhz);
        run<0, 128>("lds-dma 128-B rows", a, w, m, n, k, sink, mhz);
        run<1, 64>("vgpr+ds_write  64-B rows", a, w, m, n, k, sink, mhz);
        run<1, 128>("vgpr+ds_write 128-B rows", a, w, m, n, k, sink, mhz);
        run<0, 128, 256>("lds-dma 128-B rows, 4 waves", a, w, m, n, k, sink, mhz);
        run<0, 64, 256>("lds-dma  64-B rows, 4 waves", a, w, m, n, k, sink, mhz);
    }
    return 0;
}
