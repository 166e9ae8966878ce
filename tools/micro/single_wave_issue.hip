// What ONE wave per SIMD can issue (gfx950): cycles per v_mfma_f32_16x16x32_bf16 when a single wave feeds the matrix pipe, alone
// and with the other instruction kinds of swin_mlp512_kernel's loop between the MFMAs.  Register-only apart from the LDS reads.
//   hipcc --offload-arch=gfx950 -O3 -o _bin/single_wave_issue single_wave_issue.hip && _bin/single_wave_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

#define MF(k) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
#define PK(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q[k]) : "v"(z), "v"(c));
#define DS(k) asm volatile("ds_read_b128 %0, %1" : "=v"(w[k]) : "v"(ldsaddr));
#define WT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")");
#define SA() asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
// MFMA whose A operand is a fragment register that a ds_read refills right behind it (write-after-read), as in the kernel's fragment ring
#define MFW(k, f) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(w[f]), "v"(b));

// MODE 0: MFMAs only   1: MFMA + ds_read + s_waitcnt per pair   2: MFMA + one v_pk_fma per MFMA   3: MFMA + ds_read + waitcnt + 2 SALU per pair
// 4: v_pk_fma only     5: MFMA + one v_pk_fma per 2 MFMAs + ds_read + waitcnt per pair (the loop's mix)
// 6: the fragment ring: the pair's MFMAs read w[p & 3], the ds_read behind them refills it     7: MFMAs on FOUR accumulators only (dependent every 4th)
// 8: ring of 8 fragment registers
template <int MODE>
__global__ void k(int iters, float *sink, unsigned long long *ticks) {
    __shared__ char lds[4096];
    f32x4_t acc[16];
    f32x2_t q[8];
    bf16x8_t w[8];
    const bf16x8_t a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {2, 3, 4, 5, 6, 7, 8, 9};
    const f32x2_t z = {0.5f, 0.25f}, c = {0.1f, 0.2f};
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; ++i) q[i] = (f32x2_t){(float)threadIdx.x, 1.f};
    for (int i = 0; i < 8; ++i) w[i] = a;
    const unsigned ldsaddr = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds + (threadIdx.x & 63) * 16;
    unsigned sc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {      // 64 MFMAs per iteration
#pragma unroll
            for (int p = 0; p < 8; ++p) {  // pairs
                if (MODE == 6) { WT(3) MFW(2 * p, p & 3) MFW(2 * p + 1, p & 3) DS(p & 3) }
                else if (MODE == 8) { WT(7) MFW(2 * p, p & 7) MFW(2 * p + 1, p & 7) DS(p & 7) }
                else if (MODE == 7) { MF((2 * p) & 3) MF((2 * p + 1) & 3) }
                else if (MODE != 4) { MF(2 * p) MF(2 * p + 1) }
                if (MODE == 1 || MODE == 3 || MODE == 5) { WT(3) DS(p & 3) }
                if (MODE == 3) { SA() SA() }
                if (MODE == 2) { PK(p) PK((p + 4) & 7) }
                if (MODE == 5) { PK(p) }
                if (MODE == 4) { PK(p) PK((p + 4) & 7) }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)sc;
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    for (int i = 0; i < 8; ++i) s += q[i][0];
    for (int i = 0; i < 8; ++i) s += (float)w[i][0];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE>
void run(const char *name, int threads, int iters, float *sink, unsigned long long *ticks, double per_iter_mfma, double per_iter_pk) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, 10, sink, ticks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, iters, sink, ticks);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t;
    hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double cyc = (double)t / iters;
    printf("%-62s %d waves/SIMD: %8.1f ticks per iteration (%5.2f per MFMA, %5.2f per pk_fma)  %.3f ms  -> %.0f MHz if ticks are cycles\n", name, threads / 256, cyc,
           per_iter_mfma ? cyc / per_iter_mfma : 0.0, per_iter_pk ? cyc / per_iter_pk : 0.0, ms, (double)t / ms / 1e3);
}

int main() {
    float *sink;
    unsigned long long *ticks;
    hipMalloc(&sink, 4);
    hipMalloc(&ticks, 8);
    const int it = 20000;
    for (int threads : {256}) {
        run<0>("64 MFMA", threads, it, sink, ticks, 64, 0);
        run<1>("64 MFMA + 32 x (s_waitcnt, ds_read_b128)", threads, it, sink, ticks, 64, 0);
        run<3>("64 MFMA + 32 x (s_waitcnt, ds_read_b128, 2 SALU)", threads, it, sink, ticks, 64, 0);
        run<2>("64 MFMA + 64 v_pk_fma_f32", threads, it, sink, ticks, 64, 64);
        run<4>("64 v_pk_fma_f32", threads, it, sink, ticks, 0, 64);
        run<5>("64 MFMA + 32 x (s_waitcnt, ds_read_b128, v_pk_fma_f32)", threads, it, sink, ticks, 64, 32);
        run<6>("64 MFMA reading a 4-register fragment ring refilled behind them", threads, it, sink, ticks, 64, 0);
        run<8>("64 MFMA reading an 8-register fragment ring refilled behind them", threads, it, sink, ticks, 64, 0);
        run<7>("64 MFMA on 4 accumulators (every 4th dependent)", threads, it, sink, ticks, 64, 0);
    }
    return 0;
}
