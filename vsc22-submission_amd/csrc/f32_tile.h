// Exact fp32 tile product on v_mfma_f32_32x32x2_f32, shared by the similarity sweeps (knn.hip) and the fp32 convolution
// GEMM of the matching-track networks (conv.hip): acc = R_tile . Q_tile^T for a 128 x 128 tile, one accumulator per output
// over the whole K range, ascending k -- bit for bit an fmaf chain (oracle/knn_oracle.c).
//
// Operands are "packed": rows padded to a multiple of 32 floats and, inside every group of 8, stored as
// k = {0,2,4,6 | 1,3,5,7}, so that a lane's ds_read_b128 yields the operands of four consecutive MFMAs whose k pairs are
// (0,1),(2,3),(4,5),(6,7): wide LDS reads AND ascending chain order.
#pragma once
#include "common.h"

namespace f32tile {
namespace {   // internal linkage: this header is compiled into more than one translation unit

constexpr int TQ = 128, TR = 128, KS = 32;          // tile: queries, refs, floats per K-step
constexpr int TILE_BYTES = 128 * KS * 4;            // 16 KiB per operand per stage
constexpr int LDS_STAGE = 4 * TILE_BYTES;           // R0 Q0 R1 Q1

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// ---- layout pre-pass ------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_pack_kernel(const float *__restrict__ src,
                                                       float *__restrict__ dst, int64_t n, int d,
                                                       int dpad) {
    const int64_t total = n * (dpad >> 2);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / (dpad >> 2);
        const int c4 = (int)(e - row * (dpad >> 2));  // 16-byte chunk within the row
        const int base = (c4 >> 1) * 8, half = c4 & 1;
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = base + 2 * t + half;
            v[t] = k < d ? src[row * d + k] : 0.f;
        }
        *(float4 *)(dst + row * dpad + c4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---- staging (128 rows x 32 floats, 128-byte rows, chunk ^= (row >> 1) & 7) -------------
__device__ __forceinline__ void stage_tile(const float *src, int64_t ld, int64_t row0,
                                           int64_t row_last, int k0, char *tile, int wave,
                                           int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = j * 4 + wave;
        const int r = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int64_t gr = row0 + r;
        gr = gr > row_last ? row_last : gr;
        const float *g = src + gr * ld + k0 + c * 4;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(tile + piece * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ f32x4_t lds_frag(const char *tile, int row, int chunk) {
    return *(const f32x4_t *)(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// acc[a][b][reg] = <ref r0 + wm*64 + a*32 + row(reg, lane>>5), query q0 + wn*64 + b*32 + (lane&31)>
// for one 128 x 128 tile: double-buffered LDS-DMA staging + 32x32x2 f32 MFMA, ascending k.
// The stream of K-slabs is continuous ACROSS ref tiles: during the last K-step of this tile the
// first slab of the next one (next_r0 >= 0) is already in flight, so a tile does not start with
// an exposed HBM round trip.  `cur` is the LDS buffer holding this tile's first slab; `primed`
// says whether it is already there.  Ends with a workgroup barrier.
__device__ __forceinline__ void score_tile(f32x16_t (&acc)[2][2], const float *rp, const float *qp,
                                           int64_t nr, int64_t nq, int dpad, int64_t r0, int64_t next_r0,
                                           int64_t q0, char *lds, int wave, int lane, int &cur,
                                           bool primed) {
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nks = dpad / KS;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (!primed) {
        stage_tile(rp, dpad, r0, nr - 1, 0, lds + cur * 2 * TILE_BYTES, wave, lane);
        stage_tile(qp, dpad, q0, nq - 1, 0, lds + cur * 2 * TILE_BYTES + TILE_BYTES, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (int ks = 0; ks < nks; ++ks) {
        char *nxt = lds + (cur ^ 1) * 2 * TILE_BYTES;
        if (ks + 1 < nks) {
            stage_tile(rp, dpad, r0, nr - 1, (ks + 1) * KS, nxt, wave, lane);
            stage_tile(qp, dpad, q0, nq - 1, (ks + 1) * KS, nxt + TILE_BYTES, wave, lane);
        } else if (next_r0 >= 0) {
            stage_tile(rp, dpad, next_r0, nr - 1, 0, nxt, wave, lane);
            stage_tile(qp, dpad, q0, nq - 1, 0, nxt + TILE_BYTES, wave, lane);
        }
        const char *rtile = lds + cur * 2 * TILE_BYTES;
        const char *qtile = rtile + TILE_BYTES;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {  // 8 k per chunk pair
            f32x4_t af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = lds_frag(rtile, wm * 64 + a * 32 + l31, 2 * pr + hi);
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = lds_frag(qtile, wn * 64 + b * 32 + l31, 2 * pr + hi);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][t], bf[b][t], acc[a][b], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
}

}  // namespace
}  // namespace f32tile
