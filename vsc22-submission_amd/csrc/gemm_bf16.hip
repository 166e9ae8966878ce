// out[M,N] = epilogue(A[M,K] . W[N,K]^T + bias)   bf16 inputs, fp32 accumulate (MFMA).
//
// Every Linear / patch-conv of the encoder goes through here (the reference runs
// them as torch nn.Linear / nn.Conv2d inside the TorchScript backbone,
// infer/src/extractor.py:23).
//
// CDNA4 mapping
//   * block = 256 threads = 4 waves (2x2), tile 128(M) x 128(N) x 64(K), each wave a
//     64x64 sub-tile = 4x4 v_mfma_f32_16x16x32_bf16 accumulators (64 fp32 VGPRs).
//   * operands are staged HBM -> LDS by global_load_lds_dwordx4 (LDS-DMA, no VGPR round
//     trip), two LDS buffers (64 KiB static), next K-step in flight while this one is
//     multiplied.
//   * LDS image is row-major [128][64] bf16 (128-byte rows).  A ds_read_b128 fragment
//     read has 16 lanes on 16 different rows at the same 16-B chunk, which would hit 2
//     of the 16 slots of the 256-B bank row (8-way conflict); chunk c of row r is
//     therefore stored at chunk c ^ ((r >> 1) & 7).  LDS-DMA writes lane-linear, so the
//     permutation is applied to the per-lane SOURCE address and again on the read.
//   * the MFMA is issued with W as the "A" operand and A as the "B" operand, so a lane's 4
//     accumulator registers are 4 consecutive N columns of one output row: the epilogue
//     stores 8 B (bf16) / 16 B (fp32) per lane instead of four scalars.
//   * block id -> tile is XCD-aware (common.h xcd_remap): each XCD walks a contiguous
//     range of tiles, N fastest, so an A row-panel is fetched into one L2 only.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB, same for the W tile (BN == BM)

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

struct GemmArgs {
    const uint16_t *a;
    const uint16_t *w;
    const float *bias;
    const float *aux;
    void *out;
    int64_t m;
    int n, k, tokens, tiles_n;
};

// Stage rows [row0, row0+128) x k in [k0, k0+64) of a row-major bf16 matrix into one
// LDS tile.  16 pieces of 1 KiB (8 rows each); wave w issues pieces w, w+4, w+8, w+12.
__device__ __forceinline__ void stage_tile(const uint16_t *src, int64_t ld, int64_t row0,
                                           int64_t row_last, int k0, char *tile, int wave,
                                           int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = j * 4 + wave;
        const int r = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);  // source chunk for LDS chunk (lane & 7)
        int64_t gr = row0 + r;
        gr = gr > row_last ? row_last : gr;  // rows past the edge re-read the last row; never stored
        const uint16_t *g = src + gr * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(tile + piece * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8_t lds_frag(const char *tile, int row, int chunk) {
    return *(const bf16x8_t *)(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
}
__device__ __forceinline__ float quick_gelu(float x) {
    return x / (1.0f + __expf(-1.702f * x));
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES];  // A0 W0 A1 W1

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const int64_t a_last = p.m - 1, w_last = p.n - 1;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = p.k / BK;
    stage_tile(p.a, p.k, m0, a_last, 0, lds, wave, lane);
    stage_tile(p.w, p.k, n0, w_last, 0, lds + TILE_BYTES, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            char *nxt = lds + (cur ^ 1) * 2 * TILE_BYTES;
            stage_tile(p.a, p.k, m0, a_last, (kt + 1) * BK, nxt, wave, lane);
            stage_tile(p.w, p.k, n0, w_last, (kt + 1) * BK, nxt + TILE_BYTES, wave, lane);
        }
        const char *at = lds + cur * 2 * TILE_BYTES;
        const char *wt = at + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = lds_frag(at, wm * 64 + i * 16 + fr, fq + 4 * kk);
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = lds_frag(wt, wn * 64 + j * 16 + fr, fq + 4 * kk);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // Epilogue.  acc[i][j][r]: row m = m0 + wm*64 + i*16 + (lane & 15),
    //                          col n = n0 + wn*64 + j*16 + (lane >> 4)*4 + r.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + wm * 64 + i * 16 + fr;
        if (m >= p.m) continue;
        int64_t orow = m;
        const float *auxrow = nullptr;
        if (EPI == VSC_EPI_PATCH_F32) {
            const int pt = p.tokens - 1;
            const int64_t f = m / pt;
            const int tok = (int)(m - f * pt) + 1;
            orow = f * p.tokens + tok;
            auxrow = p.aux + (int64_t)tok * p.n;
        } else if (EPI == VSC_EPI_RESADD_F32) {
            auxrow = p.aux + m * p.n;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + fq * 4;
            if (n >= p.n) continue;
            f32x4_t v = acc[i][j];
            if (p.bias) v += *(const f32x4_t *)(p.bias + n);
            if (EPI == VSC_EPI_BF16 || EPI == VSC_EPI_GELU_BF16 || EPI == VSC_EPI_QGELU_BF16) {
                if (EPI == VSC_EPI_GELU_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                } else if (EPI == VSC_EPI_QGELU_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
                }
                uint2 pk;
                pk.x = pack_bf16x2(v[0], v[1]);
                pk.y = pack_bf16x2(v[2], v[3]);
                *(uint2 *)((uint16_t *)p.out + orow * p.n + n) = pk;
            } else {
                v += *(const f32x4_t *)(auxrow + n);
                *(f32x4_t *)((float *)p.out + orow * p.n + n) = v;
            }
        }
    }
}

template <int EPI>
int launch_t(const GemmArgs &p, int tiles_m, hipStream_t stream) {
    hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(tiles_m * p.tiles_n), dim3(256), 0, stream, p);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

}  // namespace

int launch_gemm_bf16(const uint16_t *a, const uint16_t *w, const float *bias, const float *aux,
                     void *out, int64_t m, int n, int k, int epilogue, int tokens,
                     hipStream_t stream) {
    VSC_REQUIRE(a && w && out, "gemm: null operand");
    VSC_REQUIRE(m > 0 && n > 0 && k > 0, "gemm: empty problem m=%lld n=%d k=%d", (long long)m, n, k);
    VSC_REQUIRE(k % BK == 0, "gemm: K=%d must be a multiple of %d", k, BK);
    VSC_REQUIRE(n % 4 == 0, "gemm: N=%d must be a multiple of 4", n);
    const int64_t tiles_m = (m + BM - 1) / BM;
    const int tiles_n = (n + BN - 1) / BN;
    VSC_REQUIRE(tiles_m * tiles_n < (1ll << 31), "gemm: grid too large");
    GemmArgs p{a, w, bias, aux, out, m, n, k, tokens, tiles_n};
    switch (epilogue) {
        case VSC_EPI_BF16: return launch_t<VSC_EPI_BF16>(p, (int)tiles_m, stream);
        case VSC_EPI_GELU_BF16: return launch_t<VSC_EPI_GELU_BF16>(p, (int)tiles_m, stream);
        case VSC_EPI_QGELU_BF16: return launch_t<VSC_EPI_QGELU_BF16>(p, (int)tiles_m, stream);
        case VSC_EPI_RESADD_F32:
            VSC_REQUIRE(aux, "gemm: RESADD needs the residual pointer");
            return launch_t<VSC_EPI_RESADD_F32>(p, (int)tiles_m, stream);
        case VSC_EPI_PATCH_F32:
            VSC_REQUIRE(aux && tokens > 1, "gemm: PATCH needs pos and tokens");
            VSC_REQUIRE(m % (tokens - 1) == 0, "gemm: PATCH rows %lld not a multiple of %d patches",
                        (long long)m, tokens - 1);
            return launch_t<VSC_EPI_PATCH_F32>(p, (int)tiles_m, stream);
        default: VSC_REQUIRE(false, "gemm: unknown epilogue %d", epilogue);
    }
    return VSC_OK;
}
