// Does an LDS-DMA instruction hold a wave's issue slot?  (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O3 dma_issue.hip -o dma_issue && ./dma_issue
// One workgroup per CU, WAVES waves (4 = one per SIMD, 8 = two).  Each wave runs STEPS steps of
// a step of the 256 x 256 x 64 bf16 tile -- per CU 64 LDS-DMA pieces (8 rows x 128 B each, whole cache lines, the same
// L2-resident 4 MiB for every CU) and 256 v_mfma_f32_32x32x16_bf16 (register operands only), split over the waves:
//   form 0: the pieces only        form 1: the MFMAs only
//   form 2: both, one piece after every 4 MFMAs
//   form 3: both, the pieces first, then the MFMAs
//   form 4: both, one piece per 4 MFMAs, wave w issuing its piece after MFMA (w & 3) of the group (staggered)
//   form 5: the pieces only, issued by wave 0 alone (16 per step): the cost of one piece without contention
// with the DMA issued as global_load_lds (KIND 0) or raw_buffer_load_lds (KIND 1), and prints shader cycles per step
// (s_memtime ticks == shader cycles on this part).  If form 2 ~ max(form 0, form 1) the two overlap inside a wave;
// if form 2 ~ form 0 + form 1 the DMA issue blocks the MFMA stream.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int FORM, int KIND, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void k(const char *src, int64_t row_bytes, int steps, float *sink, long long *ticks) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // this wave's pieces: rows (q * WAVES + wave) * 8 .. + 7 of the 512-row tile, 128 B per row per step
    constexpr int NP = 64 / WAVES;  // pieces per wave per step
    const char *base = src;
    unsigned loff[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) loff[q] = (unsigned)(((q * WAVES + wave) * 8 + (lane >> 3)) * row_bytes + (lane & 7) * 16);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7fffffff, 0x00020000);
    constexpr int AI = WAVES == 4 ? 4 : 2;  // accumulator rows: 256 (4 waves) or 128 (8 waves) registers
    f32x16 acc[AI][4];
#pragma unroll
    for (int i = 0; i < AI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (__bf16)(float)(lane + i + e);
            b[i][e] = (__bf16)(float)(lane - i - e);
        }
    auto piece = [&](int kt, int q) {
        char *l = lds + (kt & 1) * 65536 + (q * WAVES + wave) * 1024;
        if (KIND == 0)
            __builtin_amdgcn_global_load_lds((gptr_t)(base + kt * 128 + loff[q]), (lptr_t)l, 16, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)l, 16, (int)loff[q], kt * 128, 0, 0);
    };
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int kt = 0; kt < steps; ++kt) {
        if (FORM == 0 || FORM == 3) {
#pragma unroll
            for (int q = 0; q < NP; ++q) piece(kt, q);
        }
        if (FORM == 5 && wave == 0) {
#pragma unroll
            for (int q = 0; q < NP; ++q) piece(kt, q);
        }
        if (FORM != 0 && FORM != 5) {
#pragma unroll
            for (int g = 0; g < NP; ++g) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[g % AI][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(g / AI) & 3], b[j], acc[g % AI][j], 0, 0, 0);
                    if (FORM == 4 && j == (wave & 3)) piece(kt, g);
                }
                if (FORM == 2) piece(kt, g);
            }
        }
        if (FORM != 1) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");  // previous step's pieces have landed
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < AI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    s += (float)lds[threadIdx.x * 16];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int FORM, int KIND, int WAVES>
void run(const char *src, int64_t row_bytes, int steps, float *sink, long long *ticks, const char *label) {
    auto kern = k<FORM, KIND, WAVES>;
    const int smem = 2 * 65536;
    CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(WAVES * 64), smem, 0, src, row_bytes, steps, sink, ticks);
    CHECK(hipDeviceSynchronize());
    long long h[256];
    CHECK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
    double sum = 0;
    for (int i = 0; i < 256; ++i) sum += (double)h[i];
    const double per = sum / 256 / steps;
    printf("%-46s %8.0f cycles/step", label, per);
    if (FORM == 5) printf("  (%5.1f cycles/piece, one wave)", per / (64 / WAVES));
    else if (FORM != 1) printf("  (%5.1f B/clk/CU DMA)", 65536.0 / per);
    printf("\n");
}

int main() {
    const int steps = 64;
    const int64_t row_bytes = steps * 128;
    const size_t bytes = (size_t)512 * row_bytes;  // 512 rows x 8 KiB = 4 MiB, shared by every CU
    char *src;
    float *sink;
    long long *ticks;
    CHECK(hipMalloc(&src, bytes));
    CHECK(hipMemset(src, 1, bytes));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMalloc(&ticks, 256 * 8));
#define ROW(F, K, W, L) run<F, K, W>(src, row_bytes, steps, sink, ticks, L)
    ROW(1, 0, 4, "4 waves  MFMA only");
    ROW(0, 0, 4, "4 waves  DMA only   global_load_lds");
    ROW(0, 1, 4, "4 waves  DMA only   buffer_load_lds");
    ROW(2, 0, 4, "4 waves  interleave global_load_lds");
    ROW(2, 1, 4, "4 waves  interleave buffer_load_lds");
    ROW(3, 0, 4, "4 waves  burst      global_load_lds");
    ROW(3, 1, 4, "4 waves  burst      buffer_load_lds");
    ROW(4, 0, 4, "4 waves  staggered  global_load_lds");
    ROW(4, 1, 4, "4 waves  staggered  buffer_load_lds");
    ROW(5, 0, 4, "4 waves  solo wave  global_load_lds");
    ROW(1, 0, 8, "8 waves  MFMA only");
    ROW(0, 0, 8, "8 waves  DMA only   global_load_lds");
    ROW(0, 1, 8, "8 waves  DMA only   buffer_load_lds");
    ROW(2, 0, 8, "8 waves  interleave global_load_lds");
    ROW(2, 1, 8, "8 waves  interleave buffer_load_lds");
    ROW(3, 0, 8, "8 waves  burst      global_load_lds");
    ROW(4, 0, 8, "8 waves  staggered  global_load_lds");
    return 0;
}
