// Is a raw buffer access range-checked on voffset + soffset, or on voffset alone?  And does voffset + soffset wrap?
// num_records = 1024 bytes inside a 1 MiB allocation filled with 0xAA.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/_bin/buffer_oob_probe tools/micro/buffer_oob_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint32_t *buf, uint32_t *res) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, 1024, 0x00020000);
    // case 1: voffset in range (0), soffset beyond the extent (4096): value 0x11111111
    __builtin_amdgcn_raw_buffer_store_b32(0x11111111u, r, 0, 4096, 0);
    // case 2: voffset beyond the extent (8192), soffset 0: value 0x22222222
    __builtin_amdgcn_raw_buffer_store_b32(0x22222222u, r, 8192, 0, 0);
    // case 3: voffset 0xfffffff0, soffset 16 + 512 -> wraps to 512: value 0x33333333
    __builtin_amdgcn_raw_buffer_store_b32(0x33333333u, r, 0xfffffff0u, 16 + 512, 0);
    // case 4: in range: voffset 256 + soffset 128
    __builtin_amdgcn_raw_buffer_store_b32(0x44444444u, r, 256, 128, 0);
    // loads: same cases
    res[0] = __builtin_amdgcn_raw_buffer_load_b32(r, 0, 4096 + 64, 0);     // voffset ok, soffset beyond
    res[1] = __builtin_amdgcn_raw_buffer_load_b32(r, 8192 + 64, 0, 0);     // voffset beyond
    res[2] = __builtin_amdgcn_raw_buffer_load_b32(r, 0xfffffff0u, 16 + 640, 0);   // wraps to 640
}
int main() {
    uint32_t *buf, *res, h[262144], hr[4];
    hipMalloc(&buf, 1 << 20); hipMalloc(&res, 16);
    hipMemset(buf, 0xAA, 1 << 20);
    hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, 0, buf, res);
    hipMemcpy(h, buf, 1 << 20, hipMemcpyDeviceToHost); hipMemcpy(hr, res, 12, hipMemcpyDeviceToHost);
    printf("store voffset 0 + soffset 4096 (extent 1024): word at 4096 = %08x  (%s)\n", h[4096 / 4], h[4096 / 4] == 0x11111111u ? "WRITTEN: soffset is NOT range-checked" : "dropped");
    printf("store voffset 8192 + soffset 0:               word at 8192 = %08x  (%s)\n", h[8192 / 4], h[8192 / 4] == 0x22222222u ? "WRITTEN" : "dropped");
    printf("store voffset 0xfffffff0 + soffset 528:       word at  512 = %08x  (%s)\n", h[512 / 4], h[512 / 4] == 0x33333333u ? "WRITTEN: the sum wraps into range" : "dropped");
    printf("store voffset 256 + soffset 128:              word at  384 = %08x\n", h[384 / 4]);
    printf("load  voffset 0 + soffset 4160: %08x   load voffset 8256: %08x   load voffset 0xfffffff0 + soffset 656 (-> 640): %08x\n", hr[0], hr[1], hr[2]);
    return 0;
}
