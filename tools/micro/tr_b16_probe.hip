// What does ds_read_b64_tr_b16 return?  LDS holds u16 element e at byte 2 e (value = e); every lane reads 8 bytes at its own address
// under several address schemes; the four u16 a lane receives are printed as element indices.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/_bin/tr_b16_probe tools/micro/tr_b16_probe.hip && tools/micro/_bin/tr_b16_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void probe(int scheme, uint16_t *out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    uint32_t addr = 0;
    switch (scheme) {
        case 0: addr = 0; break;                                              // every lane the same address
        case 1: addr = l * 8; break;                                          // lane-linear 8-byte pieces
        case 2: addr = (l & 15) * 128 + (l >> 4) * 8; break;                  // 16 rows of 128 B, 4 column groups
        case 3: addr = (l >> 2 & 3) * 128 + (l & 3) * 8 + (l >> 4) * 512; break;   // per 16 lanes: 4 rows x 4 col groups (32 B per row), rows 128 B apart
        case 4: addr = (l & 3) * 128 + (l >> 2 & 3) * 8 + (l >> 4) * 512; break;   // per 16 lanes: row = l & 3, col group = (l >> 2) & 3
    }
    addr += (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t *)lds;   // LDS byte address of the array
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)(v >> (16 * j));
}
int main() {
    uint16_t *d, h[256];
    hipMalloc(&d, 512);
    const char *names[] = {"same address 0", "addr = 8 l", "addr = (l & 15) * 128 + (l >> 4) * 8", "16 lanes = 4 rows x 4 col groups (row = (l >> 2) & 3)",
                           "16 lanes: row = l & 3, col group = (l >> 2) & 3"};
    for (int s = 0; s < 5; ++s) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, s, d);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("scheme %d (%s): element index = byte / 2; row of a 128-byte-row image = index / 64, column = index %% 64\n", s, names[s]);
        for (int l = 0; l < 64; ++l) {
            printf("  l%2d:", l);
            for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 4 + j]);
            if ((l & 3) == 3) printf("\n");
        }
    }
    return 0;
}
