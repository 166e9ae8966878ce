// Is the scalar offset of a buffer instruction part of the range check on gfx950?  A descriptor of 4 KiB over an 8-KiB buffer of ones;
// dword loads at (voffset, soffset) = (4096 + 4 lane, 0), (4 lane, 4096), (2048 + 4 lane, 2048), (4 lane, 0):
// 0 = the hardware refused the access (out of range), 1 = it read past the descriptor's extent.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/bsr tools/micro/buffer_soffset_range.hip && /tmp/bsr
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const int *buf, int *out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, 4096, 0x00020000);
    const int l = threadIdx.x;
    out[l] = __builtin_amdgcn_raw_buffer_load_b32(r, 4096 + 4 * l, 0, 0);
    out[64 + l] = __builtin_amdgcn_raw_buffer_load_b32(r, 4 * l, 4096, 0);
    out[128 + l] = __builtin_amdgcn_raw_buffer_load_b32(r, 2048 + 4 * l, 2048, 0);
    out[192 + l] = __builtin_amdgcn_raw_buffer_load_b32(r, 4 * l, 0, 0);
}
int main() {
    int *buf, *out, h[256], ones[2048];
    for (int &v : ones) v = 1;
    hipMalloc(&buf, 8192); hipMalloc(&out, 1024);
    hipMemcpy(buf, ones, 8192, hipMemcpyHostToDevice);
    k<<<1, 64>>>(buf, out);
    hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
    const char *names[] = {"voffset 4096 + 4 lane, soffset 0   ", "voffset 4 lane,        soffset 4096", "voffset 2048 + 4 lane, soffset 2048", "voffset 4 lane,        soffset 0   "};
    for (int c = 0; c < 4; ++c) printf("%s: lane 0 -> %d, lane 63 -> %d\n", names[c], h[64 * c], h[64 * c + 63]);
    return 0;
}
