// Does MODE.FP16_OVFL (bit 23) make v_cvt_pk_f16_f32 clamp an overflowing value to +-65504 instead of returning inf (gfx950)?
//   hipcc --offload-arch=gfx950 -O2 tools/micro/fp16_ovfl_probe.hip -o /tmp/fp16_ovfl_probe && /tmp/fp16_ovfl_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
typedef __attribute__((ext_vector_type(2))) float f2_t;
__global__ void k(const float *in, unsigned *out, int ovfl) {
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    f2_t v = {in[2 * threadIdx.x], in[2 * threadIdx.x + 1]};
    h2_t b = __builtin_convertvector(v, h2_t);
    out[threadIdx.x] = *(unsigned *)&b;
}
int main() {
    float h[8] = {1.0f, 65504.0f, 65520.0f, 1e6f, -1e6f, INFINITY, 70000.0f, 6e-8f};
    float *d; unsigned *o, r[4];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int ovfl = 0; ovfl < 2; ++ovfl) {
        hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, o, ovfl);
        hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d:", ovfl);
        for (int i = 0; i < 4; ++i) printf(" %04x %04x", r[i] & 0xffff, r[i] >> 16);
        printf("   (7bff = 65504, 7c00 = inf, fbff = -65504)\n");
    }
    return 0;
}
