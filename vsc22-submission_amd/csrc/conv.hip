// fp32 convolution building blocks of the matching track's networks: the MobileNetV3 pair classifier and the HRNet
// refinement net that score / localise copies on frame x frame similarity maps
// (VSC22-Matching-Track-1st/infer/infer_matching.py:158-204 match_classify / match_refine; train/models.py:6-40).
// The reference runs them as TorchScript fp32 modules; here every layer is one of
//
//   vsc_conv2d_f32        dense convolution (any kernel / stride / padding), BatchNorm folded into weight + bias on the
//                         host, optional residual and activation fused:  out = act(conv(x) + bias [+ residual])
//   vsc_dwconv2d_f32      depthwise convolution + bias + activation
//   vsc_global_avgpool_f32, vsc_channel_scale_f32 (squeeze-excite gate), vsc_upsample_add_f32 (HRNet fuse / concat)
//
// Activations are NHWC float32 (a pixel's channels are contiguous: a convolution is a GEMM whose rows are pixels).
//
// CDNA4 mapping of the dense convolution: implicit GEMM on the EXACT fp32 matrix pipe.  out[pixel, co] = sum_k
// patch[pixel, k] w[co, k], k = (kh, kw, ci).  An im2col pass writes the patches as rows padded to 32 floats -- in plain k
// order: the k-interleave of f32_tile.h only fixes the ORDER of the fmaf chain (bit-exactness against the search oracle),
// which a convolution does not need, and plain order lets a 16-byte chunk of a patch be one 16-byte load of 4 channels, and
// lets a 1 x 1 / stride-1 layer whose channel count is a multiple of 32 read its input in place, with no patch matrix at
// all -- and the product runs on the 128 x 128 v_mfma_f32_32x32x2_f32
// tiles of the similarity sweep (157 TF/s peak, ascending-k fmaf chains: results match a float32 reference to rounding
// of the summation order, ~1e-6, where a bf16 pipeline through ~60 layers would not hold the 1e-3 of the probability maps).
// The weights are the MFMA row operand and the pixels the column operand, so a lane owns one pixel and 4 consecutive
// output channels per register group: 16-byte stores along the channel axis.
#include <mutex>

#define VSC_TU_BF16 1   // this file is bf16 by construction in every build of the library (common.h, "the encoders' 16-bit operand type")
#include "common.h"
#include "f32_tile.h"

namespace {

using namespace f32tile;

__device__ __forceinline__ float activate(float v, int act) {
    switch (act) {
        case VSC_ACT_RELU: return fmaxf(v, 0.f);
        case VSC_ACT_HARDSWISH: return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.0f / 6.0f);
        case VSC_ACT_HARDSIGMOID: return fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.0f / 6.0f);
        case VSC_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));   // exact (erf) GELU, torch's default
        default: return v;
    }
}

// patches of x [n, h, w, ldx >= c] -> packed [n * ho * wo, kpad], k = (i * kw + j) * c + ci, zero outside the image
__global__ __launch_bounds__(256) void im2col_pack_kernel(const float *__restrict__ x, float *__restrict__ dst, int64_t rows,
                                                          int h, int w, int c, int ldx, int kh, int kw, int stride, int pad,
                                                          int ho, int wo, int k, int kpad) {
    const int64_t total = rows * (kpad >> 2);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / (kpad >> 2);
        const int c4 = (int)(e - row * (kpad >> 2));
        const int ox = (int)(row % wo);
        const int64_t t = row / wo;
        const int oy = (int)(t % ho);
        const int64_t img = t / ho;
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = c4 * 4 + u;
            float val = 0.f;
            if (kk < k) {
                const int ci = kk % c, ij = kk / c;
                const int j = ij % kw, i = ij / kw;
                const int iy = oy * stride + i - pad, ix = ox * stride + j - pad;
                if (iy >= 0 && iy < h && ix >= 0 && ix < w) val = x[((img * h + iy) * w + ix) * ldx + ci];
            }
            v[u] = val;
        }
        *(float4 *)(dst + row * kpad + c4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// The same pass without integer divisions in the inner loop (they were most of its time: 35 % of an HRNet pass): one
// workgroup row per output image row (blockIdx.x = img * ho + oy), the (kh, kw, ci) decomposition of every k in an LDS
// table built once per workgroup, threads over (ox, 16-byte chunk).
constexpr int IM2COL_TABLE = 4096;   // k values the table holds (16 KiB); larger K takes the generic kernel
__global__ __launch_bounds__(256) void im2col_pack_rows_kernel(const float *__restrict__ x, float *__restrict__ dst, int h, int w,
                                                               int c, int ldx, int kh, int kw, int stride, int pad, int ho,
                                                               int wo, int k, int kpad) {
    __shared__ unsigned tab[IM2COL_TABLE];   // (i << 28) | (j << 24) | ci, 0xFFFFFFFF past k
    for (int kk = threadIdx.x; kk < kpad; kk += 256) {
        unsigned v = 0xFFFFFFFFu;
        if (kk < k) {
            const int ci = kk % c, ij = kk / c;
            v = ((unsigned)(ij / kw) << 28) | ((unsigned)(ij % kw) << 24) | (unsigned)ci;
        }
        tab[kk] = v;
    }
    __syncthreads();
    const int oy = blockIdx.x % ho;
    const int64_t img = blockIdx.x / ho;
    const int chunks = kpad >> 2;
    const int64_t row0 = (int64_t)blockIdx.x * wo;
    const float *ximg = x + img * h * w * ldx;
    const bool vec = (c & 3) == 0 && (ldx & 3) == 0 && (((uintptr_t)x) & 15) == 0;   // a chunk = 4 channels of one tap: one 16-byte load
    for (int e = blockIdx.y * 256 + threadIdx.x; e < wo * chunks; e += gridDim.y * 256) {
        const int ox = e / chunks, c4 = e - ox * chunks;   // one division per 16 bytes written (chunks is small)
        const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
        float4 out4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vec) {
            const unsigned t = tab[c4 * 4];
            const int iy = iy0 + (int)(t >> 28), ix = ix0 + (int)((t >> 24) & 15u);
            if (t != 0xFFFFFFFFu && iy >= 0 && iy < h && ix >= 0 && ix < w)
                out4 = *(const float4 *)(ximg + ((int64_t)iy * w + ix) * ldx + (t & 0xFFFFFFu));
        } else {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned t = tab[c4 * 4 + u];
                const int iy = iy0 + (int)(t >> 28), ix = ix0 + (int)((t >> 24) & 15u);
                float val = 0.f;
                if (t != 0xFFFFFFFFu && iy >= 0 && iy < h && ix >= 0 && ix < w) val = ximg[((int64_t)iy * w + ix) * ldx + (t & 0xFFFFFFu)];
                v[u] = val;
            }
            out4 = make_float4(v[0], v[1], v[2], v[3]);
        }
        *(float4 *)(dst + (row0 + ox) * kpad + c4 * 4) = out4;
    }
}

// w [rows, k] -> [rows, kpad], zero padded (plain k order)
__global__ __launch_bounds__(256) void pad_rows_kernel(const float *__restrict__ src, float *__restrict__ dst, int64_t rows, int k,
                                                       int kpad) {
    const int64_t total = rows * kpad;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / kpad;
        const int kk = (int)(e - r * kpad);
        dst[e] = kk < k ? src[r * k + kk] : 0.f;
    }
}

struct ConvGemmArgs {
    const float *wp;     // packed weights [cout, kpad]
    const float *ap;     // packed patches [rows, kpad]
    const float *bias;   // [cout] or null
    const float *res;    // residual [rows, ldr] or null
    float *out;          // [rows, ldo]
    int64_t rows;
    int cout, kpad, ldo, ldr, act, tiles_c;
    int64_t tiles_p;     // pixel tiles (the narrow kernel walks tiles_p * tiles_c tiles persistently)
    // implicit patch gathering (conv_gemm_narrow_kernel<true>): the convolution's own geometry instead of `ap`
    const float *x;      // input [n, h, w, ldx]
    const float *zeros;  // >= 16 bytes of zeros: where padding taps and k >= K point
    int h, w, cin, ldx, kh, kw, stride, pad, ho, wo, k;
    int no_remap;        // diagnostic (VSC_CONV_REMAP=0)
};

__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvGemmArgs p) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
    const int64_t tile = blockIdx.x;
    const int tc = (int)(tile % p.tiles_c);
    const int64_t tp = tile / p.tiles_c;
    const int64_t c0 = (int64_t)tc * TR, p0 = tp * TQ;
    f32x16_t acc[2][2];
    int cur = 0;
    score_tile(acc, p.wp, p.ap, p.cout, p.rows, p.kpad, c0, -1, p0, lds, wave, lane, cur, false);
    // acc[a][b][reg] = <w[c0 + wm*64 + a*32 + 8*(reg>>2) + 4*hi + (reg&3)], patch[p0 + wn*64 + b*32 + l31]>
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int64_t pix = p0 + wn * 64 + b * 32 + l31;
        if (pix >= p.rows) continue;
        float *orow = p.out + pix * p.ldo;
        const float *rrow = p.res ? p.res + pix * p.ldr : nullptr;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = (int)c0 + wm * 64 + a * 32 + 8 * g + 4 * hi;
                if (co >= p.cout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[a][b][4 * g + r];
                    if (co + r < p.cout) {
                        if (p.bias) v[r] += p.bias[co + r];
                        if (rrow) v[r] += rrow[co + r];
                        v[r] = activate(v[r], p.act);
                    }
                }
                if (co + 3 < p.cout && ((p.ldo | co) & 3) == 0) {
                    *(float4 *)(orow + co) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < p.cout) orow[co + r] = v[r];
                }
            }
    }
}

// Narrow variant for the thin layers (HRNet's 18- / 36- / 72-wide branches, the SE and head convolutions): 32 output
// channels x 256 pixels per workgroup, wave w owns pixels [64 w, 64 w + 64) as two 32 x 32 accumulators.  An 18-channel
// layer fills 56 % of this tile against 14 % of the 128-row one.  Same packed operands, same ascending-k chains.
constexpr int NARROW_C = 32;
constexpr int NARROW_W_BYTES = 4096;                         // 32 weight rows x 32 floats
constexpr int narrow_stage(int nt) { return NARROW_W_BYTES + nt * TILE_BYTES; }   // W | P0 [| P1]: 36 KiB at 256 pixels (two workgroups per CU)

// IMPLICIT: the patch operand is gathered from the input feature map while it is staged (no materialised im2col matrix:
// the pack pass and the GEMM's read of its output were 616 MB each way for one 3 x 3 layer of HRNet's 224 x 224 branch, and
// 37 % of a refinement pass).  A 16-byte chunk of a patch row is 4 consecutive channels of one tap (cin % 4 == 0, plain
// k = (i * kw + j) * cin + ci), so every lane of the LDS-DMA points at x[img, iy0 + i, ix0 + j, ci .. ci+3] -- or at a
// zero line when the tap falls into the padding or k >= K.  The pixel part of the address is decoded once per tile
// (divisions), the tap part comes from a per-workgroup LDS table indexed by the chunk.  Same values in the same LDS image as
// the packed path: results are bit-identical.
// NT = 128-pixel tiles per workgroup.  NT = 1 (four waves, 32 pixels each; 44 KiB of LDS: three workgroups per CU) is for the layers
// whose tile list at 256 pixels does not fill the chip or fills it in 1.x rounds -- HRNet's 72 @ 56 x 56 (588 tiles on 512 slots:
// a second round for 76 of them) and 144 @ 28 x 28 (245 tiles) branches.
template <bool IMPLICIT, int STAGES, int NW, int NT = 2>   // NW waves: 4 (wave = 64 pixels, two accumulators) or 8 (wave = 32 pixels, one)
__global__ __launch_bounds__(NW * 64, NT == 1 ? (STAGES == 3 ? 2 : 3) : 1) void conv_gemm_narrow_kernel(ConvGemmArgs p) {
    constexpr int NARROW_STAGE = narrow_stage(NT), NARROW_P = NT * 128;
    __shared__ __attribute__((aligned(16))) char lds[STAGES * NARROW_STAGE + (IMPLICIT ? IM2COL_TABLE : 0)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    unsigned *tab = (unsigned *)(lds + STAGES * NARROW_STAGE);   // per 16-byte chunk of k: (i << 28) | (j << 24) | ci, ~0u past K
    if (IMPLICIT) {
        for (int ch = tid; ch < p.kpad / 4; ch += NW * 64) {
            const int kk = ch * 4;
            unsigned v = 0xFFFFFFFFu;
            if (kk < p.k) {
                const int ci = kk % p.cin, ij = kk / p.cin;
                v = ((unsigned)(ij / p.kw) << 28) | ((unsigned)(ij % p.kw) << 24) | (unsigned)ci;
            }
            tab[ch] = v;
        }
        __syncthreads();
    }
    constexpr int PJ = 16 / NW;   // 1-KiB pieces of a 128-row pixel tile per wave
    // this lane's patch rows of a tile (2 pixel halves x 4 pieces): element offset of x[img, iy0, ix0, 0] and (iy0 << 16) | ix0
    struct Rows {
        int64_t pbase[NT][PJ];
        int pyx[NT][PJ];
    };
    auto decode = [&](int64_t p0, Rows &rw) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < PJ; ++j) {
                const int r = (j * NW + wave) * 8 + (lane >> 3);
                int64_t pix = p0 + t * 128 + r;
                pix = pix > p.rows - 1 ? p.rows - 1 : pix;
                const int64_t img = pix / ((int64_t)p.ho * p.wo);
                const int rem = (int)(pix - img * p.ho * p.wo);
                const int oy = rem / p.wo, ox = rem - oy * p.wo;
                const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
                rw.pbase[t][j] = ((img * p.h + iy0) * p.w + ix0) * (int64_t)p.ldx;
                rw.pyx[t][j] = (iy0 << 16) | (ix0 & 0xffff);
            }
    };
    auto stage = [&](int ks, char *st, int64_t c0, int64_t p0, const Rows &rw) {
        // weights: pieces 0..3 (8 rows each) of a 128-row tile image, one per wave (waves 0-3)
        if (wave < 4) {
            const int r = wave * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int64_t gr = c0 + r;
            gr = gr > p.cout - 1 ? p.cout - 1 : gr;
            __builtin_amdgcn_global_load_lds((gptr_t)(p.wp + gr * p.kpad + ks * KS + c * 4), (lptr_t)(st + wave * 1024), 16, 0, 0);
        }
        if (!IMPLICIT) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < PJ; ++j) {
                    const int piece = j * NW + wave;
                    const int r = piece * 8 + (lane >> 3);
                    const int c = (lane & 7) ^ ((r >> 1) & 7);
                    int64_t gr = p0 + t * 128 + r;
                    gr = gr > p.rows - 1 ? p.rows - 1 : gr;
                    __builtin_amdgcn_global_load_lds((gptr_t)(p.ap + gr * p.kpad + ks * KS + c * 4),
                                                     (lptr_t)(st + NARROW_W_BYTES + t * TILE_BYTES + piece * 1024), 16, 0, 0);
                }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < PJ; ++j) {
                    const int piece = j * NW + wave;
                    const int r = piece * 8 + (lane >> 3);
                    const int c = (lane & 7) ^ ((r >> 1) & 7);
                    const unsigned tv = tab[ks * (KS / 4) + c];
                    const int i = (int)(tv >> 28), jj = (int)((tv >> 24) & 15u), ci = (int)(tv & 0xFFFFFFu);
                    const int iy = (rw.pyx[t][j] >> 16) + i, ix = (int)(short)(rw.pyx[t][j] & 0xffff) + jj;
                    const bool ok = tv != 0xFFFFFFFFu && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
                    const float *g = ok ? p.x + rw.pbase[t][j] + ((int64_t)i * p.w + jj) * p.ldx + ci : p.zeros;
                    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(st + NARROW_W_BYTES + t * TILE_BYTES + piece * 1024), 16, 0, 0);
                }
        }
    };
    // Persistent over the tile list, with the operand stream running STAGES - 1 K-steps AHEAD of the MFMAs and straight across
    // tile boundaries (a staging cursor of its own): with two stages and a full vmcnt(0) + barrier per 32-float K-step a wave
    // spent ~11 k cycles per K-step for 2 k cycles of MFMA issue (18 -> 18 at 224 x 224: 195 us against 63 us); three stages
    // keep two K-steps of loads in flight behind the one being multiplied (counted vmcnt: one stage = 9 loads per wave).
    const int nks = p.kpad / KS;
    const int64_t ntiles = p.tiles_p * p.tiles_c;
    // XCD-aware order: virtual tile vt runs on XCD vt % 8 (gridDim is a multiple of 8 or the whole tile list); xcd_remap gives every
    // XCD a contiguous range of pixel tiles, so the rows a 3 x 3 tile shares with its neighbours are fetched into ONE L2
    if ((int64_t)blockIdx.x >= ntiles) return;
    const bool remap = ntiles < (1ll << 31) && !p.no_remap;
    auto origin = [&](int64_t vt, int64_t &c0, int64_t &p0) {
        const int64_t tile = remap ? (int64_t)xcd_remap((int)vt, (int)ntiles) : vt;
        c0 = (tile % p.tiles_c) * NARROW_C;
        p0 = (tile / p.tiles_c) * NARROW_P;
    };
    // staging cursor
    int64_t s_vt = blockIdx.x, s_c0, s_p0;
    int s_ks = 0, s_buf = 0;
    bool s_valid = true;
    Rows s_rw;
    origin(s_vt, s_c0, s_p0);
    if (IMPLICIT) decode(s_p0, s_rw);
    auto stage_next = [&]() {   // workgroup-uniform
        stage(s_ks, lds + s_buf * NARROW_STAGE, s_c0, s_p0, s_rw);
        s_buf = s_buf + 1 == STAGES ? 0 : s_buf + 1;
        if (++s_ks == nks) {
            s_ks = 0;
            s_vt += gridDim.x;
            s_valid = s_vt < ntiles;
            if (s_valid) {
                origin(s_vt, s_c0, s_p0);
                if (IMPLICIT) decode(s_p0, s_rw);
            }
        }
    };
    int beyond = -1;   // stages issued beyond the one being multiplied
#pragma unroll
    for (int i = 0; i < STAGES - 1; ++i)
        if (s_valid) {
            stage_next();
            ++beyond;
        }
    constexpr int LOADS = 1 + NT * PJ;   // LDS-DMA instructions per wave and stage when every wave also stages weights (NW == 4)
    if (STAGES == 3 && NW == 4 && beyond == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int64_t vt = blockIdx.x; vt < ntiles; vt += gridDim.x) {
        int64_t c0, p0;
        origin(vt, c0, p0);
        constexpr int NB = 4 * NT / NW;                              // 32-pixel accumulators per wave
        constexpr int WPT = NW / NT;                                 // waves per 128-pixel tile
        f32x16_t acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        for (int ks = 0; ks < nks; ++ks) {
            if (s_valid) {
                stage_next();
                ++beyond;
            }
            const char *wt = lds + cur * NARROW_STAGE;
            const char *pt = wt + NARROW_W_BYTES + (wave / WPT) * TILE_BYTES;
            const int prow = (wave % WPT) * (32 * NB);
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const f32x4_t af = lds_frag(wt, l31, 2 * pr + hi);
                f32x4_t bf[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) bf[b] = lds_frag(pt, prow + b * 32 + l31, 2 * pr + hi);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t], bf[b][t], acc[b], 0, 0, 0);
            }
            // the next stage must have landed; the ones behind it may stay in flight (loads complete in order, and the epilogue's
            // stores, which sit in the same queue, only make the count more conservative)
            if (STAGES == 3 && NW == 4 && beyond == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur = cur + 1 == STAGES ? 0 : cur + 1;
            --beyond;
        }
        // acc[b][reg] = <w[c0 + 8*(reg>>2) + 4*hi + (reg&3)], patch[p0 + wave*(NARROW_P/NW) + b*32 + l31]>
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int64_t pix = p0 + wave * (NARROW_P / NW) + b * 32 + l31;
            if (pix >= p.rows) continue;
            float *orow = p.out + pix * p.ldo;
            const float *rrow = p.res ? p.res + pix * p.ldr : nullptr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = (int)c0 + 8 * g + 4 * hi;
                if (co >= p.cout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[b][4 * g + r];
                    if (co + r < p.cout) {
                        if (p.bias) v[r] += p.bias[co + r];
                        if (rrow) v[r] += rrow[co + r];
                        v[r] = activate(v[r], p.act);
                    }
                }
                if (co + 3 < p.cout && ((p.ldo | co) & 3) == 0) {
                    *(float4 *)(orow + co) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < p.cout) orow[co + r] = v[r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Direct 3 x 3 / stride 1 / pad 1 convolution for the thin high-resolution layers (<= 32 output channels, cin = 4 CQ): the
// implicit-GEMM kernel above fetches every input pixel once per tap -- nine 16-byte L2 requests per 4 channels and pixel, 18
// B/clk/CU asked of a path that sustains ~12 -- and waits for them (PMC: matrix pipe 43 % busy, vector pipe 10 %).  Here a
// workgroup stages the (8 + 2) x (32 + 2)-pixel HALO of its 8 x 32 output tile ONCE (dense [pixel][cin] image in LDS, by
// LDS-DMA; pixels outside the image point past the buffer descriptor and arrive as zeros: the padding) and reads the nine
// taps' operands from it at shifted pixel positions: 1/9 of the requests.  The weights (<= 32 rows x 9 cin) sit in LDS for
// the workgroup's whole walk over tiles; the next tile's halo streams in under the current tile's MFMAs (two halo buffers).
//   * wave w = output row w of the tile, lane & 31 = output column, one 32 x 32 accumulator (channels x pixels);
//   * k runs over (tap, channel) in the packed weights' order; a K-step is a PAIR of 16-byte chunks (2 p, 2 p + 1), the
//     lane half (lane >> 5) takes chunk 2 p + half of both operands -- four MFMAs per pair, as in the implicit kernel.  With
//     5 chunks per tap a pair may straddle two taps: the halo offset of a chunk is a compile-time constant, the half picks one
//     of two (one v_cndmask per four MFMAs); an odd chunk count gets a zero weight chunk at the end;
//   * LDS strides: pixels 16 CQ bytes apart (an odd multiple of 16 for CQ = 5, 9: conflict-free ds_read_b128 over 32
//     consecutive pixels), weight rows (2 NPAIR + 1) 16-byte chunks.
// Same products as the implicit kernel, summed in a different order (its K-steps pair the chunks of 32-float slabs).
struct DirectArgs {
    const float *x, *wp, *bias, *res;
    float *out;
    int n, h, w, ldx, cout, kpad, ldo, ldr, act;
    int tiles_x, tiles_y;
    int64_t ntiles;
    unsigned xbytes;
    int cin4;   // input channels / 4 (the split-bf16 kernel pads them to chunk pairs itself)
};

// NACC accumulators per wave (32 output channels each), WR weight rows kept in LDS (>= cout; a lane whose channel row does not
// exist reads the last one -- its accumulator rows are never stored).  <5, 1, 32> leaves room for two workgroups per CU; the
// 36-channel layers (<9, 2, 40>: 52 KiB of weights + two 48-KiB halos) run one.
template <int CQ, int NACC, int WR>
__global__ __launch_bounds__(512, NACC == 1 && CQ <= 5 ? 2 : 1) void conv3x3_direct_kernel(DirectArgs p) {
    constexpr int TH = 8, TW = 32, HW = TW + 2, HH = TH + 2, HPIX = HW * HH;
    constexpr int NCH = 9 * CQ, NPAIR = (NCH + 1) / 2, WROW = (2 * NPAIR + 1) * 16;   // bytes per weight row in LDS
    constexpr int PIECES = (HPIX * CQ + 63) / 64, HALO_BYTES = PIECES * 1024, W_BYTES = (WR * WROW + 1023) / 1024 * 1024;
    __shared__ __attribute__((aligned(16))) char lds[W_BYTES + 2 * HALO_BYTES];
    char *wl = lds, *halo = lds + W_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    // ---- weights: row r (output channel), chunk c (k = 4 c .. 4 c + 3 of the packed row), zeros past cout / past K
    for (int e = tid; e < WR * 2 * NPAIR; e += 512) {
        const int r = e / (2 * NPAIR), c = e - r * (2 * NPAIR);
        f32x4_t v = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (r < p.cout && c < NCH) v = *(const f32x4_t *)(p.wp + (int64_t)r * p.kpad + c * 4);
        *(f32x4_t *)(wl + r * WROW + c * 16) = v;
    }
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.xbytes, 0x00020000);
    // halo of tile t into buffer b: chunk e of the dense image = pixel e / CQ (row-major over the HH x HW halo), channels 4 (e % CQ)..
    auto stage = [&](int64_t t, int b) {
        const int tx = (int)(t % p.tiles_x);
        const int64_t t2 = t / p.tiles_x;
        const int ty = (int)(t2 % p.tiles_y), img = (int)(t2 / p.tiles_y);
#pragma unroll
        for (int qq = 0; qq < (PIECES + 7) / 8; ++qq) {
            const int q = qq * 8 + wave;
            if (q >= PIECES) break;   // wave-uniform
            const int e = q * 64 + lane;
            const int pix = e / CQ, ch = e - pix * CQ;
            const int hy = pix / HW, hx = pix - hy * HW;
            const int iy = ty * TH - 1 + hy, ix = tx * TW - 1 + hx;
            const bool ok = pix < HPIX && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
            const unsigned voff = ok ? (unsigned)((((img * p.h + iy) * p.w + ix) * p.ldx + ch * 4) * 4) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lptr_t)(halo + b * HALO_BYTES + q * 1024), 16, voff, 0, 0, 0);
        }
    };
    int64_t t = blockIdx.x;
    if (t >= p.ntiles) return;
    stage(t, 0);
    int buf = 0;
    const char *wrow[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
        const int r = a * 32 + l31;
        wrow[a] = wl + (r < WR ? r : WR - 1) * WROW + hi * 16;
    }
    for (; t < p.ntiles; t += gridDim.x, buf ^= 1) {
        // this wave's pieces of the tile's halo have landed (explicit vmcnt wait: an LDS-DMA is tracked by the VM counter only and
        // __syncthreads() compiles to a bare s_barrier); behind the barrier everyone's have (and, the first time, the weights
        // are written), and every wave is done reading the other buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + gridDim.x < p.ntiles) stage(t + gridDim.x, buf ^ 1);
        const char *hb = halo + buf * HALO_BYTES + (wave * HW + l31) * (CQ * 16);
        // this lane's output pixel; its residual values are requested now and arrive under the MFMAs
        const int tx = (int)(t % p.tiles_x);
        const int64_t t2 = t / p.tiles_x;
        const int ty = (int)(t2 % p.tiles_y);
        const int64_t img = t2 / p.tiles_y;
        const int oy = ty * TH + wave, ox = tx * TW + l31;
        const bool inside = oy < p.h && ox < p.w;
        const int64_t pix = (img * p.h + (inside ? oy : 0)) * p.w + (inside ? ox : 0);
        f32x4_t rs[NACC][4];
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                rs[a][g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                const int co = a * 32 + 8 * g + 4 * hi;
                if (p.res && inside && co + 3 < p.cout && ((p.ldr | co) & 3) == 0) rs[a][g] = *(const f32x4_t *)(p.res + pix * p.ldr + co);
                else if (p.res && inside) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < p.cout) rs[a][g][r] = p.res[pix * p.ldr + co + r];
                }
            }
        f32x16_t acc[NACC];
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
#pragma unroll
        for (int pr = 0; pr < NPAIR; ++pr) {
            // halo byte offset of chunk c relative to the lane's pixel: tap (c / CQ) -> rows (tap / 3), columns (tap % 3)
            constexpr auto off_of = [](int c) { return (((c / CQ) / 3) * HW + (c / CQ) % 3) * (CQ * 16) + (c % CQ) * 16; };
            const int c0 = 2 * pr, c1 = 2 * pr + 1 < NCH ? 2 * pr + 1 : 2 * pr;   // the pad chunk re-reads finite data (its weights are 0)
            const int bo = hi ? off_of(c1) : off_of(c0);
            const f32x4_t bf = *(const f32x4_t *)(hb + bo);
            f32x4_t af[NACC];
#pragma unroll
            for (int a = 0; a < NACC; ++a) af[a] = *(const f32x4_t *)(wrow[a] + pr * 32);
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][k4], bf[k4], acc[a], 0, 0, 0);
        }
        // acc[a][reg] = <w[32 a + 8 (reg >> 2) + 4 hi + (reg & 3)], patch of output pixel (row wave, column l31) of the tile>
        if (inside) {
            float *orow = p.out + pix * p.ldo;
#pragma unroll
            for (int ag = 0; ag < NACC * 4; ++ag) {
                const int a = ag >> 2, g = ag & 3;
                const int co = a * 32 + 8 * g + 4 * hi;
                if (co >= p.cout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[a][4 * g + r];
                    if (co + r < p.cout) {
                        if (p.bias) v[r] += p.bias[co + r];
                        v[r] += rs[a][g][r];
                        v[r] = activate(v[r], p.act);
                    }
                }
                if (co + 3 < p.cout && ((p.ldo | co) & 3) == 0) {
                    *(float4 *)(orow + co) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < p.cout) orow[co + r] = v[r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// 1 x 1 expansion from 64 channels (HRNet layer1: 64 -> 256 at 224 x 224 with the 256-wide residual, 4 + 1 calls per pass, 12 % of
// it on the 128 x 128 tile kernel: two K-steps per tile, a serial load -> multiply -> scattered 16-byte epilogue per workgroup,
// 1.7 TB/s).  The layer is a stream: 64 floats in, cout out (+ cout of residual) per pixel, 26 GF of products under 1.85 GB.
//   * persistent workgroups of 8 waves, one per CU; a tile = 128 pixels x all channels; wave w owns channels 32 w .. 32 w + 31 and
//     keeps its 32 x 64 weights in 32 VGPRs for the whole kernel;
//   * the tile's 128 x 64 input is brought in by LDS-DMA (32 KiB, two buffers: tile t + 1 lands under tile t's MFMAs), rows
//     256 B with the 16-byte chunk index xor-ed with the row (conflict-free ds_read_b128 over consecutive pixels);
//   * PIXELS are the MFMA row operand and channels the column operand: a lane owns one channel and the accumulator registers
//     walk over pixels, so every store / residual load instruction covers two full 128-byte lines (32 consecutive channels of
//     one pixel per lane half) -- no transposition, no partial lines;
//   * the residual of tile t is requested before its MFMAs; the stores of tile t are not waited for until tile t + 1's MFMAs
//     are done.
struct ExpandArgs {
    const float *x, *wp, *bias, *res;
    float *out;
    int64_t rows, ntiles;
    int cout, ldo, ldr, act;
};

__global__ __launch_bounds__(512, 1) void conv1x1_expand64_kernel(ExpandArgs p) {
    constexpr int CIN = 64, TP = 128, BUF = TP * CIN * 4;
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const bool active = wave * 32 < p.cout;   // wave-uniform: waves past cout only help with the staging
    const int ch = wave * 32 + l31;
    // weights of this lane's channel: chunk 2 pr + hi of the packed row (k = 8 pr + 4 hi .. + 3), as in lds_frag()'s pairing
    f32x4_t wf[8];
#pragma unroll
    for (int pr = 0; pr < 8; ++pr)
        wf[pr] = active ? *(const f32x4_t *)(p.wp + (int64_t)ch * CIN + (2 * pr + hi) * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const float bias = active && p.bias ? p.bias[ch] : 0.f;
    // residual / output through buffer descriptors: one per-lane offset (pixel 4 hi, channel ch) + a wave-uniform pixel offset per
    // accumulator register; pixels past the last row fall outside the descriptor (loads give 0, stores are dropped)
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, (int)(p.rows * p.ldo * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(p.res ? p.res : p.out), 0, (int)(p.rows * (p.res ? p.ldr : p.ldo) * 4), 0x00020000);
    const uint32_t ovoff = (uint32_t)(4 * hi * p.ldo + ch) * 4u, rvoff = (uint32_t)(4 * hi * p.ldr + ch) * 4u;
    auto stage = [&](int64_t t, int b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = j * 8 + wave;                 // 1 KiB = 4 rows
            const int r = piece * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (r & 15);
            int64_t gr = t * TP + r;
            gr = gr > p.rows - 1 ? p.rows - 1 : gr;
            __builtin_amdgcn_global_load_lds((gptr_t)(p.x + gr * CIN + c * 4), (lptr_t)(lds + b * BUF + piece * 1024), 16, 0, 0);
        }
    };
    int64_t t = blockIdx.x;
    if (t >= p.ntiles) return;
    stage(t, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (; t < p.ntiles; t += gridDim.x, buf ^= 1) {
        if (t + gridDim.x < p.ntiles) stage(t + gridDim.x, buf ^ 1);
        const int64_t p0 = t * TP;
        // residual: rs[blk][reg] belongs to pixel p0 + 32 blk + 8 (reg >> 2) + 4 hi + (reg & 3), channel ch
        float rs[4][16];
        if (p.res && active) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const uint32_t soff = (uint32_t)(p0 + blk * 32 + 8 * (reg >> 2) + (reg & 3)) * (uint32_t)(p.ldr * 4);
                    rs[blk][reg] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, rvoff, soff, 0));
                }
        }
        f32x16_t acc[4];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[blk][r] = 0.f;
        if (active) {
            const char *xb = lds + buf * BUF;
#pragma unroll
            for (int pr = 0; pr < 8; ++pr)
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) {
                    const int row = blk * 32 + l31;
                    const f32x4_t af = *(const f32x4_t *)(xb + row * 256 + (((2 * pr + hi) ^ (row & 15)) << 4));
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k4], wf[pr][k4], acc[blk], 0, 0, 0);
                }
        }
        // the next tile's input and this tile's residual have landed; behind the barrier every wave is done with `buf`
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (active) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const uint32_t soff = (uint32_t)(p0 + blk * 32 + 8 * (reg >> 2) + (reg & 3)) * (uint32_t)(p.ldo * 4);
                    float v = acc[blk][reg] + bias;
                    if (p.res) v += rs[blk][reg];
                    v = activate(v, p.act);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, ovoff, soff, 0);
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The same streaming form for the classifier's pointwise layers with few input channels (16 -> 72, 24 -> 88 / 96, 40 -> 120 / 240,
// 48 -> 144 / 288, 96 -> 576: on the 32-channel x 256-pixel tile kernel every channel tile re-gathers the pixels and writes 16-byte
// pieces of 288..2304-byte rows at its own time: 1.1-2.0 TB/s).  CP = chunk pairs per pixel (input channels rounded up to 8);
// a tile = TP pixels x the workgroup's 128 channels (grid.y groups: all of a pixel's channels are written within one tile time); weights of a wave's 32 channels in CP x 4 VGPRs; a pixel's pad chunk (cin % 8 == 4) reads the next pixel's
// first channels -- finite values against zero weights.  The k order is the tile kernels' (chunk pairs of 32-float slabs): same bits.
template <int CP>
__global__ __launch_bounds__(256, CP <= 3 ? 4 : 3) void conv1x1_stream_kernel(ExpandArgs p, int cin, int kpad) {
    // four waves = 128 channels per workgroup, 128- / 64-pixel tiles: 16-48 KiB of LDS and <= 128 / 168 VGPRs, so three or four
    // workgroups share a CU.  The K loop of these layers is a few MFMAs; what a tile costs is the round trip of its input and the
    // drain of its stores, and only other workgroups' tiles can fill that time (one 8-wave workgroup per CU: 16 -> 72 at 2.0 TB/s)
    constexpr int PITCH = 32 * CP;                       // bytes per pixel in LDS
    constexpr int TP = CP <= 6 ? 128 : 64, NB = TP / 32, BUF = TP * PITCH;
    constexpr int PIECES = BUF / 1024;                   // 1-KiB LDS-DMA pieces per tile
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int ch = (blockIdx.y * 4 + wave) * 32 + l31;
    const bool active = (blockIdx.y * 4 + wave) * 32 < p.cout;   // wave-uniform
    const bool mine = ch < p.cout;                                // this lane's channel exists
    f32x4_t wf[CP];
#pragma unroll
    for (int pr = 0; pr < CP; ++pr)
        wf[pr] = mine ? *(const f32x4_t *)(p.wp + (int64_t)ch * kpad + (2 * pr + hi) * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const float bias = mine && p.bias ? p.bias[ch] : 0.f;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)(p.rows * cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, (int)(p.rows * p.ldo * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(p.res ? p.res : p.out), 0, (int)(p.rows * (p.res ? p.ldr : p.ldo) * 4), 0x00020000);
    const uint32_t ovoff = mine ? (uint32_t)(4 * hi * p.ldo + ch) * 4u : 0xfffffff0u, rvoff = mine ? (uint32_t)(4 * hi * p.ldr + ch) * 4u : 0xfffffff0u;
    auto stage = [&](int64_t t, int b) {
#pragma unroll
        for (int j = 0; j < (PIECES + 3) / 4; ++j) {
            const int piece = j * 4 + wave;
            if (piece >= PIECES) break;   // wave-uniform
            const int o = piece * 1024 + lane * 16;
            const int r = o / PITCH, cb = o - r * PITCH;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lptr_t)(lds + b * BUF + piece * 1024), 16,
                                                     (uint32_t)((t * TP + r) * cin * 4 + cb), 0, 0, 0);
        }
    };
    int64_t t = blockIdx.x;
    if (t >= p.ntiles) return;
    stage(t, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (; t < p.ntiles; t += gridDim.x, buf ^= 1) {
        if (t + gridDim.x < p.ntiles) stage(t + gridDim.x, buf ^ 1);
        const int64_t p0 = t * TP;
        f32x16_t acc[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[blk][r] = 0.f;
        if (active) {
            const char *xb = lds + buf * BUF;
#pragma unroll
            for (int pr = 0; pr < CP; ++pr)
#pragma unroll
                for (int blk = 0; blk < NB; ++blk) {
                    const f32x4_t af = *(const f32x4_t *)(xb + (blk * 32 + l31) * PITCH + (2 * pr + hi) * 16);
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k4], wf[pr][k4], acc[blk], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile's input has landed (and the previous tile's stores are out)
        __syncthreads();
        if (active) {
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                float rs[16];
                if (p.res) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        rs[reg] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rrsrc, rvoff, (uint32_t)(p0 + blk * 32 + 8 * (reg >> 2) + (reg & 3)) * (uint32_t)(p.ldr * 4), 0));
                }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    float v = acc[blk][reg] + bias;
                    if (p.res) v += rs[reg];
                    v = activate(v, p.act);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, ovoff,
                                                          (uint32_t)(p0 + blk * 32 + 8 * (reg >> 2) + (reg & 3)) * (uint32_t)(p.ldo * 4), 0);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Stem of the classifier: 3 x 3 convolution from 3 input channels (K = 27) to <= 32 channels at 160 x 160, stride 2.  On the GEMM
// path it materialises a [pixels, 32] patch matrix (1.7 GB written and read for 0.6 GB of input; 1229 us).  Direct form: a thread
// owns one output pixel -- its 27 inputs in registers (neighbouring threads' patches overlap: L1 hits), all COUT channels, the
// weights through scalar loads (uniform addresses: v_fmac with an SGPR operand, no LDS) -- and writes COUT x 4 contiguous bytes.
// One fmaf chain per output over k = (i, j, ci) in the order the matrix path multiplies a 32-float slab (k, k + 4 alternating per
// chunk pair), then the bias; padding taps multiply zeros there too: identical bits.
template <int COUT>
__global__ __launch_bounds__(256) void conv_stem3_kernel(const float *__restrict__ x, const float *__restrict__ wp, const float *__restrict__ bias,
                                                         float *__restrict__ out, int64_t total, int h, int w, int kpad, int stride, int pad,
                                                         int ho, int wo, int ldo, int act) {
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= total) return;
    const int ox = (int)(pix % wo);
    const int64_t t = pix / wo;
    const int oy = (int)(t % ho);
    const int64_t img = t / ho;
    const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
    float xv[27];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int iy = iy0 + i, ix = ix0 + j;
            const bool ok = iy >= 0 && iy < h && ix >= 0 && ix < w;
            const float *src = x + ((img * h + (ok ? iy : 0)) * (int64_t)w + (ok ? ix : 0)) * 3;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) xv[(i * 3 + j) * 3 + ci] = ok ? src[ci] : 0.f;
        }
    float *o = out + pix * ldo;
    // the weights through the constant address space: uniform addresses become s_load (rows of 27 consecutive floats), the products
    // v_fmac with an SGPR operand -- one channel's chain at a time, so at most a few rows of weights are live in SGPRs
    const __attribute__((address_space(4))) float *wc = (const __attribute__((address_space(4))) float *)wp;
#pragma unroll
    for (int cq = 0; cq < COUT / 4; ++cq) {
        f32x4_t a;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                // the matrix path's order inside a 32-float slab: MFMA t of chunk pair pr multiplies k = 8 pr + t, then 8 pr + 4 + t
                const int k = (s >> 3) * 8 + ((s & 7) >> 1) + (s & 1) * 4;
                if (k < 27) acc = fmaf(xv[k], wc[(cq * 4 + q) * kpad + k], acc);
            }
            a[q] = acc;
        }
        if (bias) a += *(const f32x4_t *)(bias + cq * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = activate(a[q], act);
        *(f32x4_t *)(o + cq * 4) = a;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The direct 3 x 3 convolution on the bf16 matrix pipe with fp32-level accuracy (split operands): every fp32 value is written as
// x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (24 significant bits together), likewise the
// weights, and a product is taken as x1 w1 + x1 w2 + x2 w1 + x1 w3 + x3 w1 + x2 w2 -- the dropped terms are below 2^-24 |x w|, the
// size of an fp32 rounding -- on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 6 instructions of 16 cycles per 16 x 16 x 32
// block against 16 fp32 instructions of 64 cycles per 32 x 32 x 32: 2.7 x the fp32 pipe's rate on the same products.
//   * halo tile and weights live in LDS as three bf16 planes each; the split happens on the way in (the halo passes through
//     registers: 6 VALU instructions per value, each value then feeds 9 taps x cout products);
//   * M = output channels (NRT tiles of 16), N = pixels (a wave = one output row of 32 = two tiles), K = (tap, channel) flattened in
//     4-channel chunks, 8 chunks per MFMA; a lane's two chunks may come from different taps (two 8-byte halo reads);
//   * a lane ends up with 4 consecutive channels of one pixel: 16-byte residual loads and stores;
//   * 78 KiB of LDS (one halo buffer): two workgroups per CU; the next tile's halo is loaded into registers before the MFMAs and
//     split into LDS after them.
__device__ __forceinline__ void split_bf16x3(f32x4_t v, uint2 &p1, uint2 &p2, uint2 &p3) {
    auto hi = [](uint32_t pk, int i) { return __builtin_bit_cast(float, i ? (pk & 0xffff0000u) : (pk << 16)); };
    p1.x = pack_bf16x2(v[0], v[1]);
    p1.y = pack_bf16x2(v[2], v[3]);
    const float r0 = v[0] - hi(p1.x, 0), r1 = v[1] - hi(p1.x, 1), r2 = v[2] - hi(p1.y, 0), r3 = v[3] - hi(p1.y, 1);
    p2.x = pack_bf16x2(r0, r1);
    p2.y = pack_bf16x2(r2, r3);
    p3.x = pack_bf16x2(r0 - hi(p2.x, 0), r1 - hi(p2.x, 1));
    p3.y = pack_bf16x2(r2 - hi(p2.y, 0), r3 - hi(p2.y, 1));
}

// CP = 8-channel chunk pairs per pixel (input channels rounded up to 8: 20 -> 24, 36 -> 40), WR = weight rows kept in LDS (>= cout),
// NRT = 16-row tiles of output channels, NPT = 16-pixel tiles per wave: 2 -> a wave is an output row of an 8 x 32 tile (20 channels:
// 77 KiB, two workgroups per CU); 1 -> a wave is half a row of a 4 x 32 tile (36 channels: the three weight planes alone are 85
// KiB, the smaller halo keeps the workgroup at 134 KiB).
template <int CP, int WR, int NRT, int NPT>
__global__ __launch_bounds__(512, NPT == 2 ? 4 : 2) void conv3x3_direct_x3_kernel(DirectArgs p) {
    constexpr int TH = 4 * NPT, TW = 32, HW = TW + 2, HH = TH + 2, HPIX = HW * HH;
    constexpr int NPAIR = 9 * CP, NBLK = (NPAIR + 3) / 4;              // chunk pairs of K (8 channels of one tap), MFMA blocks of 4 pairs
    constexpr int PIXB = CP * 16, PLANE = HPIX * PIXB;                 // halo: bytes per pixel and per plane
    constexpr int WROWB = NBLK * 64 + 16, WPLANE = WR * WROWB;         // weights: bytes per row (16 B of padding: conflict-free b128 reads) and per plane
    constexpr int ITEMS = (HPIX * CP + 511) / 512;
    __shared__ __attribute__((aligned(16))) char lds[3 * WPLANE + 3 * PLANE];
    char *wl = lds, *halo = lds + 3 * WPLANE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, g = lane >> 4;
    const int cin4 = p.kpad;   // (unused name guard)
    (void)cin4;
    const int cq_real = p.cin4;   // real 4-channel chunks per pixel (5 for 20 channels)
    // ---- weights: row r, pair q = (tap, 8 channels) of the packed fp32 row (k = tap cin + ci) -> three planes; zeros past cin / K
    for (int e = tid; e < WR * NBLK * 8; e += 512) {
        const int r = e / (NBLK * 8), hc = e - r * (NBLK * 8);          // hc: half-chunk index = 2 pair + half
        const int pair = hc >> 1, tap = pair / CP, cq = (pair - tap * CP) * 2 + (hc & 1);
        f32x4_t v = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (r < p.cout && pair < NPAIR && cq < cq_real) v = *(const f32x4_t *)(p.wp + (int64_t)r * p.kpad + (tap * cq_real + cq) * 4);
        uint2 w1, w2, w3;
        split_bf16x3(v, w1, w2, w3);
        *(uint2 *)(wl + r * WROWB + hc * 8) = w1;
        *(uint2 *)(wl + WPLANE + r * WROWB + hc * 8) = w2;
        *(uint2 *)(wl + 2 * WPLANE + r * WROWB + hc * 8) = w3;
    }
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.xbytes, 0x00020000);
    // this thread's halo items (pixel, pair): position inside the halo and byte offset relative to the tile's first halo pixel
    int hyx[ITEMS], rel[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = j * 512 + tid;
        const int pix = e / CP, pr = e - pix * CP;
        const int hy = pix / HW, hx = pix - hy * HW;
        hyx[j] = pix < HPIX ? (hy << 16) | hx : -1;
        rel[j] = ((hy * p.w + hx) * p.ldx + pr * 8) * 4;
    }
    f32x4_t nxt[ITEMS][2];
    auto fetch = [&](int64_t t) {   // the halo of tile t into registers (pixels outside the image: past the descriptor, zeros)
        const int tx = (int)(t % p.tiles_x);
        const int64_t t2 = t / p.tiles_x;
        const int ty = (int)(t2 % p.tiles_y), img = (int)(t2 / p.tiles_y);
        const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
        const int base = (((img * p.h + iy0) * p.w + ix0) * p.ldx) * 4;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int iy = iy0 + (hyx[j] >> 16), ix = ix0 + (hyx[j] & 0xffff);
            const bool ok = hyx[j] >= 0 && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
            const int pr = (j * 512 + tid) % CP;
            const unsigned v0 = ok ? (unsigned)(base + rel[j]) : 0x80000000u;
            const unsigned v1 = ok && pr * 2 + 1 < cq_real ? v0 + 16u : 0x80000000u;   // the pair's second half past cin: zeros
            nxt[j][0] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, v0, 0, 0));
            nxt[j][1] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, v1, 0, 0));
        }
    };
    auto deposit = [&]() {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int e = j * 512 + tid;
            if (e >= HPIX * CP) break;
            uint2 a1, a2, a3, b1, b2, b3;
            split_bf16x3(nxt[j][0], a1, a2, a3);
            split_bf16x3(nxt[j][1], b1, b2, b3);
            *(uint4 *)(halo + e * 16) = make_uint4(a1.x, a1.y, b1.x, b1.y);
            *(uint4 *)(halo + PLANE + e * 16) = make_uint4(a2.x, a2.y, b2.x, b2.y);
            *(uint4 *)(halo + 2 * PLANE + e * 16) = make_uint4(a3.x, a3.y, b3.x, b3.y);
        }
    };
    // halo byte offset of this lane group's pair in every MFMA block (pair q -> tap q / CP, channels 8 (q % CP) ..); the pad pairs of
    // the last block re-read the last real one (their weights are zero)
    int offx[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        int q = 4 * b + g;
        q = q < NPAIR ? q : NPAIR - 1;
        const int tap = q / CP, pr = q - tap * CP;
        offx[b] = ((tap / 3) * HW + tap % 3) * PIXB + pr * 16;
    }
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, (int)((int64_t)p.n * p.h * p.w * p.ldo * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(p.res ? p.res : p.out), 0, (int)((int64_t)p.n * p.h * p.w * (p.res ? p.ldr : p.ldo) * 4), 0x00020000);
    int64_t t = blockIdx.x;
    if (t >= p.ntiles) return;
    fetch(t);
    int wrow[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
        const int r = rt * 16 + lc;
        wrow[rt] = (r < WR ? r : WR - 1) * WROWB + g * 16;
    }
    for (; t < p.ntiles; t += gridDim.x) {
        __syncthreads();   // every wave is done reading the halo of the previous tile
        deposit();
        __syncthreads();
        if (t + gridDim.x < p.ntiles) fetch(t + gridDim.x);
        const int tx = (int)(t % p.tiles_x);
        const int64_t t2 = t / p.tiles_x;
        const int ty = (int)(t2 % p.tiles_y);
        const int img = (int)(t2 / p.tiles_y);
        const int wrow_ = NPT == 2 ? wave : wave >> 1, px0 = NPT == 2 ? 0 : (wave & 1) * 16;   // this wave's output row and first pixel in the tile
        const int oy = ty * TH + wrow_;
        f32x4_t acc[NRT][NPT];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) acc[rt][pt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const char *hb = halo + (wrow_ * HW + px0 + lc) * PIXB;
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            bf16x8_t xo[3][NPT], wo[3][NRT];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) xo[pl][pt] = *(const bf16x8_t *)(hb + offx[b] + pl * PLANE + pt * 16 * PIXB);
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) wo[pl][rt] = *(const bf16x8_t *)(wl + wrow[rt] + pl * WPLANE + b * 64);
            }
            constexpr int TW_[6] = {1, 2, 0, 1, 0, 0}, TX_[6] = {1, 0, 2, 0, 1, 0};   // (w plane, x plane), smallest terms first
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt)
                        acc[rt][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo[TW_[term]][rt], xo[TX_[term]][pt], acc[rt][pt], 0, 0, 0);
        }
        // a lane holds channels 16 rt + 4 g .. + 3 of pixel (oy, tx TW + px0 + 16 pt + lc): 16-byte residual loads and stores through buffer
        // descriptors (pixels / channel groups that do not exist: an offset past the extent)
        const bool vec = ((p.ldo | p.ldr) & 3) == 0;
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                const int ox = tx * TW + px0 + pt * 16 + lc, co = rt * 16 + 4 * g;
                const bool ok = oy < p.h && ox < p.w && co < p.cout;
                const int pix = (img * p.h + oy) * p.w + ox;
                f32x4_t v = acc[rt][pt];
                if (vec && co + 3 < p.cout) {
                    const unsigned ro = ok ? (unsigned)((pix * p.ldr + co) * 4) : 0x80000000u, oo = ok ? (unsigned)((pix * p.ldo + co) * 4) : 0x80000000u;
                    if (p.bias) v += *(const f32x4_t *)(p.bias + co);
                    if (p.res) v += __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ro, 0, 0));
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = activate(v[r], p.act);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vsc_u32x4_t, v), orsrc, oo, 0, 0);
                } else if (ok) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (co + r < p.cout) {
                            float s = v[r] + (p.bias ? p.bias[co + r] : 0.f);
                            if (p.res) s += p.res[(int64_t)pix * p.ldr + co + r];
                            p.out[(int64_t)pix * p.ldo + co + r] = activate(s, p.act);
                        }
                }
            }
    }
}

// The same arithmetic for layers whose weights do not fit LDS as three planes (64 -> 64 at 224 x 224: 221 KB): the weights are
// streamed PER TAP -- the tap's [cout][cin] slab is read from the packed fp32 rows into registers while the previous tap is
// multiplied, split, and written into the other of two 28-KiB plane buffers -- under a resident halo tile (4 x 32 outputs, 78 KiB as
// three planes).  A wave = (output row, 16-pixel tile), all NRT row tiles of output channels.  8 CP channels per pass (a multiple
// of 32: a tap is CP / 4 whole MFMA blocks); wider inputs (256 -> 20: NCC = 4) take NCC passes over the tile, one per 8 CP-channel
// slice of the input, the accumulators staying in registers.
template <int CP, int NRT, int NCC>
__global__ __launch_bounds__(512, 2) void conv3x3_tap_x3_kernel(DirectArgs p) {
    static_assert(CP % 4 == 0, "a tap must be whole 32-channel MFMA blocks");
    constexpr int TH = 4, TW = 32, HW = TW + 2, HH = TH + 2, HPIX = HW * HH;
    constexpr int BPT = CP / 4;                                         // MFMA blocks per tap
    constexpr int PIXB = CP * 16, PLANE = HPIX * PIXB;
    constexpr int WR = NRT * 16, WROWB = BPT * 64 + 16, WPLANE = WR * WROWB, WBUF = 3 * WPLANE;
    constexpr int ITEMS = (HPIX * CP + 511) / 512;                      // halo chunk pairs per thread
    constexpr int WITEMS = (WR * CP * 2 + 511) / 512;                   // weight 4-float chunks of one tap per thread
    __shared__ __attribute__((aligned(16))) char lds[2 * WBUF + 3 * PLANE];
    char *wl = lds, *halo = lds + 2 * WBUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)p.xbytes, 0x00020000);
    int hyx[ITEMS], rel[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int e = j * 512 + tid;
        const int pix = e / CP, pr = e - pix * CP;
        const int hy = pix / HW, hx = pix - hy * HW;
        hyx[j] = pix < HPIX ? (hy << 16) | hx : -1;
        rel[j] = ((hy * p.w + hx) * p.ldx + pr * 8) * 4;
    }
    // weights of tap `tap`: this thread's chunks (row r, 4 channels cq) -> registers; deposit: split into buffer b
    f32x4_t wreg[WITEMS];
    auto wfetch = [&](int tap, int cc) {
#pragma unroll
        for (int j = 0; j < WITEMS; ++j) {
            const int e = j * 512 + tid;
            const int r = e / (2 * CP), cq = e - r * (2 * CP);
            wreg[j] = (e < WR * 2 * CP && r < p.cout) ? *(const f32x4_t *)(p.wp + (int64_t)r * p.kpad + ((tap * NCC + cc) * 2 * CP + cq) * 4) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto wdeposit = [&](int b) {
#pragma unroll
        for (int j = 0; j < WITEMS; ++j) {
            const int e = j * 512 + tid;
            if (e >= WR * 2 * CP) break;
            const int r = e / (2 * CP), cq = e - r * (2 * CP);
            uint2 w1, w2, w3;
            split_bf16x3(wreg[j], w1, w2, w3);
            char *dst = wl + b * WBUF + r * WROWB + cq * 8;
            *(uint2 *)dst = w1;
            *(uint2 *)(dst + WPLANE) = w2;
            *(uint2 *)(dst + 2 * WPLANE) = w3;
        }
    };
    f32x4_t nxt[ITEMS][2];
    auto fetch = [&](int64_t t, int cc) {   // the halo of tile t, input channels 8 CP cc ..
        const int tx = (int)(t % p.tiles_x);
        const int64_t t2 = t / p.tiles_x;
        const int ty = (int)(t2 % p.tiles_y), img = (int)(t2 / p.tiles_y);
        const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
        const int base = (((img * p.h + iy0) * p.w + ix0) * p.ldx + cc * 8 * CP) * 4;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int iy = iy0 + (hyx[j] >> 16), ix = ix0 + (hyx[j] & 0xffff);
            const bool ok = hyx[j] >= 0 && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
            const unsigned v0 = ok ? (unsigned)(base + rel[j]) : 0x80000000u;
            nxt[j][0] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, v0, 0, 0));
            nxt[j][1] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? v0 + 16u : 0x80000000u, 0, 0));
        }
    };
    auto deposit = [&]() {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int e = j * 512 + tid;
            if (e >= HPIX * CP) break;
            uint2 a1, a2, a3, b1, b2, b3;
            split_bf16x3(nxt[j][0], a1, a2, a3);
            split_bf16x3(nxt[j][1], b1, b2, b3);
            *(uint4 *)(halo + e * 16) = make_uint4(a1.x, a1.y, b1.x, b1.y);
            *(uint4 *)(halo + PLANE + e * 16) = make_uint4(a2.x, a2.y, b2.x, b2.y);
            *(uint4 *)(halo + 2 * PLANE + e * 16) = make_uint4(a3.x, a3.y, b3.x, b3.y);
        }
    };
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, 0, (int)((int64_t)p.n * p.h * p.w * p.ldo * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(p.res ? p.res : p.out), 0, (int)((int64_t)p.n * p.h * p.w * (p.res ? p.ldr : p.ldo) * 4), 0x00020000);
    int64_t t = blockIdx.x;
    if (t >= p.ntiles) return;
    fetch(t, 0);
    wfetch(0, 0);
    const int wrow_ = wave >> 1, px0 = (wave & 1) * 16;
    const int woff = lc * WROWB + g * 16;
    const int hoff = (wrow_ * HW + px0 + lc) * PIXB + g * 16;
    f32x4_t acc[NRT];
    for (int cc = 0; t < p.ntiles;) {
        const int ncc = cc + 1 < NCC ? cc + 1 : 0;                  // the pass after this one: the next channel slice of the tile, or
        const int64_t nt = cc + 1 < NCC ? t : t + gridDim.x;        // slice 0 of the workgroup's next tile
        __syncthreads();   // every wave is done with the previous pass's halo and with weight buffer 0 (tap 8 lives in it)
        deposit();
        wdeposit(0);
        wfetch(1, cc);
        __syncthreads();
        if (nt < p.ntiles) fetch(nt, ncc);
        const int tx = (int)(t % p.tiles_x);
        const int64_t t2 = t / p.tiles_x;
        const int ty = (int)(t2 % p.tiles_y);
        const int img = (int)(t2 / p.tiles_y);
        const int oy = ty * TH + wrow_;
        if (cc == 0) {
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) acc[rt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const char *wb = wl + (tap & 1) * WBUF + woff;
            const char *hb = halo + hoff + ((tap / 3) * HW + tap % 3) * PIXB;
#pragma unroll
            for (int b = 0; b < BPT; ++b) {
                bf16x8_t xo[3], wo[3][NRT];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    xo[pl] = *(const bf16x8_t *)(hb + pl * PLANE + b * 64);
#pragma unroll
                    for (int rt = 0; rt < NRT; ++rt) wo[pl][rt] = *(const bf16x8_t *)(wb + pl * WPLANE + rt * 16 * WROWB + b * 64);
                }
                constexpr int TW_[6] = {1, 2, 0, 1, 0, 0}, TX_[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int rt = 0; rt < NRT; ++rt)
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo[TW_[term]][rt], xo[TX_[term]], acc[rt], 0, 0, 0);
            }
            if (tap < 8) {
                // the next tap's weights go into the other buffer (its readers -- tap - 1 -- are behind the previous barrier), then the
                // tap after that is requested
                wdeposit((tap + 1) & 1);
                if (tap + 2 < 9) wfetch(tap + 2, cc);
                else wfetch(0, ncc);   // tap 0 again: the next pass's first
                __syncthreads();
            }
        }
        const int cc_done = cc;
        const int64_t t_done = t;
        (void)t_done;
        cc = ncc;
        t = nt;
        if (cc_done + 1 < NCC) continue;   // more channel slices of this tile to go
        const bool vec = ((p.ldo | p.ldr) & 3) == 0;
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            const int ox = tx * TW + px0 + lc, co = rt * 16 + 4 * g;
            const bool ok = oy < p.h && ox < p.w && co < p.cout;
            const int pix = (img * p.h + oy) * p.w + ox;
            f32x4_t v = acc[rt];
            if (vec && co + 3 < p.cout) {
                const unsigned ro = ok ? (unsigned)((pix * p.ldr + co) * 4) : 0x80000000u, oo = ok ? (unsigned)((pix * p.ldo + co) * 4) : 0x80000000u;
                if (p.bias) v += *(const f32x4_t *)(p.bias + co);
                if (p.res) v += __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ro, 0, 0));
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = activate(v[r], p.act);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vsc_u32x4_t, v), orsrc, oo, 0, 0);
            } else if (ok) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co + r < p.cout) {
                        float sx = v[r] + (p.bias ? p.bias[co + r] : 0.f);
                        if (p.res) sx += p.res[(int64_t)pix * p.ldr + co + r];
                        p.out[(int64_t)pix * p.ldo + co + r] = activate(sx, p.act);
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Split-bf16 IMPLICIT GEMM for the wide 3 x 3 layers on small maps (72 @ 56 x 56, 144 @ 28 x 28): their weights do not fit LDS as
// planes, and streamed per tap under a small halo tile they cost more than the fp32 tile kernel.  Here both operands
// are split ONCE per call into bf16 planes in the per-device scratch (the activations of these layers are 7-14 MB) and the product
// is the narrow kernel's implicit GEMM on plain bf16 data: out[pixel, co] = sum_k patch[pixel, k] w[co, k], k = (tap, ci) flattened,
// a K-step = 32 k = four 8-channel chunks gathered by LDS-DMA (16 bytes = one chunk of one plane; a tap outside the image or k >= K
// points past the buffer descriptor: zeros), six MFMAs per 16 x 16 x 32 block.  128 pixels x up to 80 channels per workgroup, two
// 39-KiB stages: two workgroups per CU.  Wave = (32 pixels, half of the row tiles).
__global__ __launch_bounds__(256) void split_planes_kernel(const float *__restrict__ x, char *__restrict__ dst, int64_t chunks, int64_t plane_bytes) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < chunks; e += (int64_t)gridDim.x * 256) {
        uint2 a, b, c;
        split_bf16x3(*(const f32x4_t *)(x + e * 4), a, b, c);
        *(uint2 *)(dst + e * 8) = a;
        *(uint2 *)(dst + plane_bytes + e * 8) = b;
        *(uint2 *)(dst + 2 * plane_bytes + e * 8) = c;
    }
}

// packed fp32 rows [cout][kpad] (k = (tap, ci)) -> planes [3][wr][kp] bf16, zero rows past cout, zero columns past K
__global__ __launch_bounds__(256) void split_weight_rows_kernel(const float *__restrict__ wp, char *__restrict__ dst, int cout, int kpad, int k, int wr, int kp) {
    const int per_row = kp / 4;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < wr * per_row; e += gridDim.x * 256) {
        const int r = e / per_row, c4 = e - r * per_row;
        f32x4_t v = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (r < cout && c4 * 4 < k) v = *(const f32x4_t *)(wp + (int64_t)r * kpad + c4 * 4);   // (K is a multiple of 8 here: whole chunks)
        uint2 a, b, c;
        split_bf16x3(v, a, b, c);
        const size_t plane = (size_t)wr * kp * 2;
        *(uint2 *)(dst + (size_t)e * 8) = a;
        *(uint2 *)(dst + plane + (size_t)e * 8) = b;
        *(uint2 *)(dst + 2 * plane + (size_t)e * 8) = c;
    }
}

struct X3GemmArgs {
    const char *xpl, *wpl;      // activation planes [3][rows_in][cin] bf16, weight planes [3][wr_total][kp] bf16
    const float *bias, *res;
    float *out;
    int64_t rows;               // output pixels
    int n, h, w, cin, cout, ldo, ldr, act;
    int nks, kp, chunks_per_tap, wr_total;
    unsigned xplane_bytes, wplane_bytes;
};

template <int NRT>   // 16-row tiles of output channels per workgroup (grid.y channel groups of NRT x 16 rows); the two wave groups take
                      // tiles 0 .. NRTG - 1 and NRTG .. NRT - 1
__global__ __launch_bounds__(512, 4) void conv_x3_gemm_kernel(X3GemmArgs p) {
    constexpr int TP = 128, WRG = NRT * 16, NRTG = (NRT + 1) / 2;
    constexpr int STAGE = 3 * (WRG + TP) * 64;   // W planes | P planes, 64-byte rows, chunk ^= (row >> 2) & 3
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, g = lane >> 4;
    const int64_t p0 = (int64_t)blockIdx.x * TP;
    const int r0 = blockIdx.y * WRG;   // first weight row of this workgroup
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.xpl, 0, (int)(3u * p.xplane_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.wpl, 0, (int)(3u * p.wplane_bytes), 0x00020000);
    // this lane's gather row: pixel p0 + 16 wave + (lane >> 2), stored chunk lane & 3 holding source chunk (lane & 3) ^ ((row >> 2) & 3)
    const int prow = wave * 16 + (lane >> 2);
    const int pch = (lane & 3) ^ ((prow >> 2) & 3);
    int64_t pix = p0 + prow;
    const bool pin = pix < p.rows;
    pix = pin ? pix : p.rows - 1;
    const int img = (int)(pix / (p.h * p.w));
    const int rem = (int)(pix - (int64_t)img * p.h * p.w);
    const int oy = rem / p.w, ox = rem - oy * p.w;
    // weights: waves 0 .. WRG / 16 - 1 stage 16 rows each
    const int wrow = wave * 16 + (lane >> 2);
    const int wch = (lane & 3) ^ ((wrow >> 2) & 3);
    const bool wstager = wave * 16 < WRG;
    auto stage = [&](int ks, int b) {
        char *st = lds + b * STAGE;
        {
            const int kc = ks * 4 + pch;                       // 8-channel chunk of K
            const int tap = kc / p.chunks_per_tap, c8 = kc - tap * p.chunks_per_tap;
            const int i = tap / 3, j = tap - i * 3;
            const int iy = oy + i - 1, ix = ox + j - 1;
            const bool ok = pin && tap < 9 && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
            const unsigned voff = ok ? (unsigned)((((img * p.h + iy) * p.w + ix) * p.cin + c8 * 8) * 2) : 0x80000000u;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lptr_t)(st + 3 * WRG * 64 + pl * TP * 64 + wave * 1024), 16, voff, pl * p.xplane_bytes, 0, 0);
        }
        if (wstager) {
            const unsigned voff = (unsigned)(((r0 + wrow) * p.kp + (ks * 4 + wch) * 8) * 2);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lptr_t)(st + pl * WRG * 64 + wave * 1024), 16, voff, pl * p.wplane_bytes, 0, 0);
        }
    };
    const int pg = wave & 3, rg = wave >> 2;   // this wave: pixels 32 pg .. + 31, row tiles rg NRTG .. (nrt of them)
    const int nrt = NRT - rg * NRTG < NRTG ? NRT - rg * NRTG : NRTG;   // wave-uniform
    f32x4_t acc[NRTG][2];
#pragma unroll
    for (int rt = 0; rt < NRTG; ++rt)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) acc[rt][pt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // operand rows of this lane: weights row 16 (rg NRTG + rt) + lc, pixels 32 pg + 16 pt + lc; chunk g of the K-step
    int wofs[NRTG], pofs[2];
#pragma unroll
    for (int rt = 0; rt < NRTG; ++rt) {
        const int r = (rg * NRTG + (rt < nrt ? rt : 0)) * 16 + lc;
        wofs[rt] = r * 64 + ((g ^ ((r >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int r = pg * 32 + pt * 16 + lc;
        pofs[pt] = 3 * WRG * 64 + r * 64 + ((g ^ ((r >> 2) & 3)) << 4);
    }
    int cur = 0;
    for (int ks = 0; ks < p.nks; ++ks) {
        if (ks + 1 < p.nks) stage(ks + 1, cur ^ 1);
        const char *st = lds + cur * STAGE;
        bf16x8_t xo[3][2], wo[3][NRTG];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) xo[pl][pt] = *(const bf16x8_t *)(st + pofs[pt] + pl * TP * 64);
#pragma unroll
            for (int rt = 0; rt < NRTG; ++rt) wo[pl][rt] = *(const bf16x8_t *)(st + wofs[rt] + pl * WRG * 64);
        }
        constexpr int TW_[6] = {1, 2, 0, 1, 0, 0}, TX_[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int rt = 0; rt < NRTG; ++rt)
                if (rt < nrt) {
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt)
                        acc[rt][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo[TW_[term]][rt], xo[TX_[term]][pt], acc[rt][pt], 0, 0, 0);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    const bool vec = ((p.ldo | p.ldr) & 3) == 0;
#pragma unroll
    for (int rt = 0; rt < NRTG; ++rt)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int64_t px = p0 + pg * 32 + pt * 16 + lc;
            const int co = r0 + (rg * NRTG + rt) * 16 + 4 * g;
            if (rt >= nrt || px >= p.rows || co >= p.cout) continue;
            f32x4_t v = acc[rt][pt];
            if (vec && co + 3 < p.cout) {
                if (p.bias) v += *(const f32x4_t *)(p.bias + co);
                if (p.res) v += *(const f32x4_t *)(p.res + px * p.ldr + co);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = activate(v[r], p.act);
                *(f32x4_t *)(p.out + px * p.ldo + co) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co + r < p.cout) {
                        float sx = v[r] + (p.bias ? p.bias[co + r] : 0.f);
                        if (p.res) sx += p.res[px * p.ldr + co + r];
                        p.out[px * p.ldo + co + r] = activate(sx, p.act);
                    }
            }
        }
}

// depthwise: one thread per (pixel, channel); w [c, kh * kw]
__global__ __launch_bounds__(256) void dwconv_kernel(const float *__restrict__ x, const float *__restrict__ wgt,
                                                     const float *__restrict__ bias, float *__restrict__ out, int64_t total,
                                                     int h, int w, int c, int kh, int kw, int stride, int pad, int ho, int wo,
                                                     int act) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int ch = (int)(e % c);
        int64_t t = e / c;
        const int ox = (int)(t % wo);
        t /= wo;
        const int oy = (int)(t % ho);
        const int64_t img = t / ho;
        float a = 0.f;
        for (int i = 0; i < kh; ++i) {
            const int iy = oy * stride + i - pad;
            if (iy < 0 || iy >= h) continue;
            for (int j = 0; j < kw; ++j) {
                const int ix = ox * stride + j - pad;
                if (ix < 0 || ix >= w) continue;
                a = fmaf(x[((img * h + iy) * w + ix) * c + ch], wgt[ch * kh * kw + i * kw + j], a);
            }
        }
        if (bias) a += bias[ch];
        out[e] = activate(a, act);
    }
}

// The same for c % 4 == 0 (every depthwise layer of the classifier): a thread owns 4 channels of one output pixel -- one float4
// of x per tap instead of four scalar loads -- the weights sit transposed ([tap][c]) in LDS so a tap's 4 weights are one
// ds_read_b128, and a workgroup row is one output image row (blockIdx.x = img * ho + oy), so the index math is one 32-bit
// division per output instead of three 64-bit ones per channel.  Same tap order and fmaf chain per channel: identical bits.
// 2048 x 160 x 160 classifier maps: 16.2 -> 12.3 ms.
__global__ __launch_bounds__(256) void dwconv_rows_kernel(const float *__restrict__ x, const float *__restrict__ wgt,
                                                          const float *__restrict__ bias, float *__restrict__ out, int h, int w,
                                                          int c, int kh, int kw, int stride, int pad, int ho, int wo, int act) {
    extern __shared__ __attribute__((aligned(16))) float wt[];   // [kh * kw][c]
    const int taps = kh * kw;
    for (int e = threadIdx.x; e < taps * c; e += 256) {
        const int ch = e / taps, tp = e - ch * taps;
        wt[tp * c + ch] = wgt[e];
    }
    __syncthreads();
    const int oy = blockIdx.x % ho;
    const int64_t img = blockIdx.x / ho;
    const int c4n = c >> 2;
    const float *ximg = x + img * h * w * c;
    float *orow = out + ((int64_t)blockIdx.x * wo) * c;
    const int iy0 = oy * stride - pad;
    for (int e = blockIdx.y * 256 + threadIdx.x; e < wo * c4n; e += gridDim.y * 256) {
        const int ox = e / c4n, c4 = e - ox * c4n;
        const int ix0 = ox * stride - pad;
        f32x4_t a = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < kh; ++i) {
            const int iy = iy0 + i;
            if (iy < 0 || iy >= h) continue;
            for (int j = 0; j < kw; ++j) {
                const int ix = ix0 + j;
                if (ix < 0 || ix >= w) continue;
                const f32x4_t xv = *(const f32x4_t *)(ximg + ((int64_t)iy * w + ix) * c + c4 * 4);
                const f32x4_t wv = *(const f32x4_t *)(wt + (i * kw + j) * c + c4 * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = fmaf(xv[r], wv[r], a[r]);
            }
        }
        if (bias) a += *(const f32x4_t *)(bias + c4 * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = activate(a[r], act);
        *(f32x4_t *)(orow + (int64_t)ox * c + c4 * 4) = a;
    }
}

// Depthwise layers on SMALL maps (the classifier's 5 x 5 and 10 x 10 stages: 576 / 240 / 288 / 144 / 120 channels, 5 x 5 taps).  The
// row kernel above restages the whole [taps][c] weight table per output image row -- 57.6 KB for 3 outputs per thread at 576 x 5 x 5:
// 783 us for 236 MB (0.3 TB/s).  Here a workgroup owns a 64-channel slab: its weights (taps x 64) are staged once, then it walks over
// images, bringing each image's slab (h w x 64 floats <= 25.6 KB) into LDS once and computing every output of it from LDS -- the
// input is read from memory exactly once, in 256-byte runs.  Same taps in the same order per channel: identical bits.
constexpr int DW_SLAB = 64;   // channels per workgroup (16 chunks of 4)
__global__ __launch_bounds__(256) void dwconv_small_kernel(const float *__restrict__ x, const float *__restrict__ wgt,
                                                           const float *__restrict__ bias, float *__restrict__ out, int n, int h, int w,
                                                           int c, int kh, int kw, int stride, int pad, int ho, int wo, int act) {
    extern __shared__ __attribute__((aligned(16))) float dw_lds[];
    const int taps = kh * kw;
    float *wt = dw_lds;                       // [taps][64]
    float *xs = dw_lds + taps * DW_SLAB;      // [h w][64]
    const int c0 = blockIdx.x * DW_SLAB;
    const int cs = c - c0 < DW_SLAB ? c - c0 : DW_SLAB;   // channels of this slab (a multiple of 4)
    for (int e = threadIdx.x; e < taps * DW_SLAB; e += 256) {
        const int tp = e >> 6, ch = e & 63;
        wt[e] = ch < cs ? wgt[(int64_t)(c0 + ch) * taps + tp] : 0.f;
    }
    const int q = threadIdx.x & 15;           // this thread's chunk of the slab, for staging and for its outputs
    const bool qok = q * 4 < cs;
    f32x4_t bz = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (bias && qok) bz = *(const f32x4_t *)(bias + c0 + q * 4);
    const int hw = h * w, howo = ho * wo;
    for (int img = blockIdx.y; img < n; img += gridDim.y) {
        const float *ximg = x + (int64_t)img * hw * c + c0;
        __syncthreads();   // the previous image's readers are done (and, the first time, the weights are written)
        for (int pix = threadIdx.x >> 4; pix < hw; pix += 16)
            if (qok) *(f32x4_t *)(xs + pix * DW_SLAB + q * 4) = *(const f32x4_t *)(ximg + (int64_t)pix * c + q * 4);
        __syncthreads();
        float *oimg = out + (int64_t)img * howo * c + c0;
        for (int op = threadIdx.x >> 4; op < howo; op += 16) {
            const int oy = op / wo, ox = op - oy * wo;
            const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
            f32x4_t a = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < kh; ++i) {
                const int iy = iy0 + i;
                if (iy < 0 || iy >= h) continue;
                for (int j = 0; j < kw; ++j) {
                    const int ix = ix0 + j;
                    if (ix < 0 || ix >= w) continue;
                    const f32x4_t xv = *(const f32x4_t *)(xs + (iy * w + ix) * DW_SLAB + q * 4);
                    const f32x4_t wv = *(const f32x4_t *)(wt + (i * kw + j) * DW_SLAB + q * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] = fmaf(xv[r], wv[r], a[r]);
                }
            }
            a += bz;
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = activate(a[r], act);
            if (qok) *(f32x4_t *)(oimg + (int64_t)op * c + q * 4) = a;
        }
    }
}

// x [n, hw, c] -> out [n, c] (mean over hw); one workgroup per (image, 64-channel slab)
__global__ __launch_bounds__(256) void avgpool_kernel(const float *__restrict__ x, float *__restrict__ out, int hw, int c) {
    __shared__ float part[4][64];
    const int img = blockIdx.y, c0 = blockIdx.x * 64, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float s = 0.f;
    if (c0 + lane < c)
        for (int i = wv; i < hw; i += 4) s += x[((int64_t)img * hw + i) * c + c0 + lane];
    part[wv][lane] = s;
    __syncthreads();
    if (wv == 0 && c0 + lane < c)
        out[(int64_t)img * c + c0 + lane] = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) / (float)hw;
}

// x [n, hw, c] *= s [n, c]
__global__ __launch_bounds__(256) void channel_scale_kernel(float *__restrict__ x, const float *__restrict__ s, int64_t total,
                                                            int hw, int c) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int ch = (int)(e % c);
        const int64_t img = e / ((int64_t)hw * c);
        x[e] *= s[img * c + ch];
    }
}

// One squeeze-excite block per launch for maps that fit LDS (every SE of the classifier: <= 100 KB per image): a workgroup brings one
// image in (16-byte loads), takes the channel means from LDS, runs the gate's two small Linears (a wave per output row, lanes over the
// input, 6-step shuffle reduction; weights straight from their packed rows, L2-resident), scales the image in LDS and writes it back:
// x is read once and written once where avgpool + two convolution launches + channel_scale read it twice and wrote it once.
//   gate = act2(W2 act1(W1 mean_hw(x) + b1) + b2),  x *= gate
__global__ __launch_bounds__(512) void se_block_kernel(float *__restrict__ x, const float *__restrict__ w1, const float *__restrict__ b1, int kpad1,
                                                       const float *__restrict__ w2, const float *__restrict__ b2, int kpad2, int hw, int c, int cr,
                                                       int act1, int act2) {
    extern __shared__ __attribute__((aligned(16))) float se_lds[];
    float *img = se_lds;                       // [hw][c]
    float *pooled = se_lds + hw * c;           // [c]
    float *hidden = pooled + c;                // [cr]
    float *gate = hidden + cr;                 // [c]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *xi = x + (int64_t)blockIdx.x * hw * c;
    const int n4 = hw * c / 4;
    for (int e = tid; e < n4; e += 512) *(f32x4_t *)(img + e * 4) = *(const f32x4_t *)(xi + e * 4);
    __syncthreads();
    const float inv = 1.0f / (float)hw;
    for (int ch = tid; ch < c; ch += 512) {
        float sacc = 0.f;
        for (int pz = 0; pz < hw; ++pz) sacc += img[pz * c + ch];
        pooled[ch] = sacc * inv;
    }
    __syncthreads();
    for (int j = wave; j < cr; j += 8) {
        float sacc = 0.f;
        for (int k = lane; k < c; k += 64) sacc = fmaf(w1[(int64_t)j * kpad1 + k], pooled[k], sacc);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sacc += __shfl_xor(sacc, m, 64);
        if (lane == 0) hidden[j] = activate(sacc + (b1 ? b1[j] : 0.f), act1);
    }
    __syncthreads();
    for (int ch = wave; ch < c; ch += 8) {
        float sacc = 0.f;
        for (int k = lane; k < cr; k += 64) sacc = fmaf(w2[(int64_t)ch * kpad2 + k], hidden[k], sacc);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sacc += __shfl_xor(sacc, m, 64);
        if (lane == 0) gate[ch] = activate(sacc + (b2 ? b2[ch] : 0.f), act2);
    }
    __syncthreads();
    const int c4 = c / 4;
    for (int e = tid; e < n4; e += 512) {
        const int q = e % c4;
        *(f32x4_t *)(xi + e * 4) = *(const f32x4_t *)(img + e * 4) * *(const f32x4_t *)(gate + q * 4);
    }
}

// out[n, y, x, coff + ch] (op)= src[n, y / f, x / f, ch]   (nearest upsample by f >= 1); accumulate ? += : =; then act
__global__ __launch_bounds__(256) void upsample_add_kernel(const float *__restrict__ src, float *__restrict__ out, int64_t total,
                                                           int h, int w, int c, int f, int ldo, int coff, int accumulate, int act) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int ch = (int)(e % c);
        int64_t t = e / c;
        const int ox = (int)(t % w);
        t /= w;
        const int oy = (int)(t % h);
        const int64_t img = t / h;
        const float v = src[((img * (h / f) + oy / f) * (w / f) + ox / f) * c + ch];
        float *o = out + ((img * h + oy) * w + ox) * ldo + coff + ch;
        *o = activate(accumulate ? *o + v : v, act);
    }
}

// The same on 16-byte chunks, with up to three upsampled terms in one pass (an HRNet fuse node is y_i + sum_j up(conv(y_j)); as
// three read-modify-write passes over the 64-MB branch-0 map plus a clone it cost 1.5 ms per refinement pass):
//   out[n, y, x, coff + ch] = act(((base[n, y, x, ch] + up_f0(s0)) + up_f1(s1)) + up_f2(s2))      (terms added in this order)
// base may be null (0) or the output window itself.  One workgroup row per output image row: the only per-thread division is by
// the chunk count.  Factors are powers of two (shifts).
struct UpsampleSumArgs {
    const float *base, *src[3];
    float *out;
    int shift[3], nsrc;
    int h, w, c4, ldb, ldo, act;
};
__global__ __launch_bounds__(256) void upsample_sum4_kernel(UpsampleSumArgs p) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= p.w * p.c4) return;
    const int ox = e / p.c4, q = e - ox * p.c4;
    const int64_t row = blockIdx.y;
    const int oy = (int)(row % p.h);
    const int64_t img = row / p.h;
    const int64_t pix = row * p.w + ox;
    f32x4_t v = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (p.base) v = *(const f32x4_t *)(p.base + pix * p.ldb + q * 4);
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k < p.nsrc) {
            const int sh = p.shift[k];
            const f32x4_t t = *(const f32x4_t *)(p.src[k] + ((img * (p.h >> sh) + (oy >> sh)) * (p.w >> sh) + (ox >> sh)) * (p.c4 * 4) + q * 4);
            v[0] += t[0], v[1] += t[1], v[2] += t[2], v[3] += t[3];
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = activate(v[r], p.act);
    *(f32x4_t *)(p.out + pix * p.ldo + q * 4) = v;
}

// softmax(q k^T / sqrt(dh)) v for one (head, query) pair per workgroup, all in fp32 (the video-score head: <= 258 tokens,
// one video at a time -- 0.1 GFLOP per layer, latency- not throughput-bound).  qkv [tokens, 3 * heads * dh] as q | k | v.
// row_off (optional): sequences of DIFFERENT lengths back to back, sequence z = rows row_off[z] .. row_off[z + 1] (vsc_attention_f32_varlen;
// `tokens` is then the longest one: the grid's x extent and the LDS size); a row's arithmetic is the same in every form.
__global__ __launch_bounds__(256) void attention_f32_kernel(const float *__restrict__ qkv, float *__restrict__ out, int tokens,
                                                            int heads, int dh, const int32_t *__restrict__ row_off = nullptr) {
    extern __shared__ float sm[];            // [tokens] probabilities, then [4][dh] partial outputs, [dh] query
    float *prob = sm, *part = sm + tokens, *qs = part + 4 * dh;
    __shared__ float red[8];
    const int q = blockIdx.x, hd = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int width = heads * dh, ld = 3 * width;
    if (row_off) {
        const int r0 = row_off[blockIdx.z];
        tokens = row_off[blockIdx.z + 1] - r0;
        if (q >= tokens) return;                   // (workgroup-uniform)
        qkv += (int64_t)r0 * ld;
        out += (int64_t)r0 * width;
    } else {
    qkv += (int64_t)blockIdx.z * tokens * ld;      // blockIdx.z: one of several sequences of the same length, back to back
    out += (int64_t)blockIdx.z * tokens * width;
    }
    for (int d = tid; d < dh; d += 256) qs[d] = qkv[(int64_t)q * ld + hd * dh + d];
    __syncthreads();
    const float scale = rsqrtf((float)dh);
    float mx = -INFINITY;
    for (int t = tid; t < tokens; t += 256) {
        const float *kr = qkv + (int64_t)t * ld + width + hd * dh;
        float a = 0.f;
        for (int d = 0; d < dh; ++d) a = fmaf(qs[d], kr[d], a);
        a *= scale;
        prob[t] = a;
        mx = fmaxf(mx, a);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int t = tid; t < tokens; t += 256) {
        const float e = expf(prob[t] - mx);
        prob[t] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wv] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
    // out[d] = sum_t prob[t] v[t][d]: wave wv takes keys wv, wv + 4, ...; lane = head-dim column (dh <= 64 per pass)
    for (int d0 = 0; d0 < dh; d0 += 64) {
        const int d = d0 + lane;
        float a = 0.f;
        if (d < dh)
            for (int t = wv; t < tokens; t += 4) a = fmaf(prob[t], qkv[(int64_t)t * ld + 2 * width + hd * dh + d], a);
        if (d < dh) part[wv * dh + d] = a;
    }
    __syncthreads();
    for (int d = tid; d < dh; d += 256)
        out[(int64_t)q * width + hd * dh + d] = ((part[d] + part[dh + d]) + (part[2 * dh + d] + part[3 * dh + d])) * inv;
}

struct Scratch {
    void *ptr = nullptr;
    size_t bytes = 0;
};
Scratch g_patch[16];   // per device, grow-only: the packed patch matrix of the convolution in flight
std::mutex g_conv_mutex;
float *g_zero_line[16] = {};   // per device: 256 bytes of zeros (implicit gathering points padding taps at it)
Scratch g_x3[16];               // per device, grow-only: split-bf16 planes (activations | weights) of the implicit GEMM in flight

inline int blocks_for(int64_t items) {
    int64_t b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

}  // namespace

extern "C" int vsc_conv_packed_k(int32_t cin, int32_t kh, int32_t kw) { return (cin * kh * kw + KS - 1) / KS * KS; }

extern "C" int vsc_conv_pack_weight_f32(const float *w_dev, float *packed_dev, int32_t cout, int32_t k, void *stream_) {
    VSC_REQUIRE(w_dev && packed_dev && cout > 0 && k > 0, "conv_pack_weight: bad arguments");
    const int kpad = (k + KS - 1) / KS * KS;
    hipLaunchKernelGGL(pad_rows_kernel, dim3(blocks_for((int64_t)cout * kpad)), dim3(256), 0, (hipStream_t)stream_, w_dev, packed_dev,
                       (int64_t)cout, k, kpad);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

static int g_conv_last_pipe = 0;   // 0: fp32 matrix / vector pipe, 1: bf16 matrix pipe on split operands (six products per multiply)
extern "C" int vsc_conv_last_pipe(void) { return g_conv_last_pipe; }

extern "C" int vsc_conv2d_f32(const float *x_dev, int64_t n, int32_t h, int32_t w, int32_t cin, int32_t ldx,
                              const float *w_packed_dev, const float *bias_dev, int32_t cout, int32_t kh, int32_t kw,
                              int32_t stride, int32_t pad, const float *res_dev, int32_t ldr, int32_t act, float *out_dev,
                              int32_t ldo, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    g_conv_last_pipe = 0;
    VSC_REQUIRE(x_dev && w_packed_dev && out_dev, "conv2d: null operand");
    VSC_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0, "conv2d: bad shape");
    VSC_REQUIRE(ldx >= cin && ldo >= cout && (!res_dev || ldr >= cout), "conv2d: leading dimensions smaller than the channel counts");
    VSC_REQUIRE(act >= VSC_ACT_NONE && act <= VSC_ACT_GELU, "conv2d: unknown activation %d", act);
    const int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
    VSC_REQUIRE(ho > 0 && wo > 0, "conv2d: empty output");
    const int64_t rows = n * ho * wo;
    const int k = cin * kh * kw, kpad = (k + KS - 1) / KS * KS;
    int dev = 0;
    VSC_CHECK_HIP(hipGetDevice(&dev));
    VSC_REQUIRE(dev >= 0 && dev < 16, "conv2d: device %d out of range", dev);
    // thin 3 x 3 / stride 1 / pad 1 layers: direct convolution from a halo tile in LDS (conv3x3_direct_kernel)
    {
        const char *de = vsc_opt(OPT_CONV_DIRECT);   // diagnostic / test switch: 0 = the implicit-GEMM path
        const int64_t xbytes = n * (int64_t)h * w * ldx * 4;
        const char *x3e = vsc_opt(OPT_CONV_X3);
        // the epilogues of the direct / tap / plane kernels move out, res and bias 16 bytes at a time whenever ldo and ldr are multiples
        // of 4: an output window at an offset (Conv.__call__(out=..., coff=...)) or a caller's unaligned buffer takes the tile kernels
        // (with ldo or ldr not a multiple of 4 those epilogues use scalar accesses anyway: any alignment is fine then)
        const bool vec_io = ((ldo | (res_dev ? ldr : 0)) & 3) == 0;
        const bool io16 = !vec_io || (((uintptr_t)out_dev | (uintptr_t)(res_dev ? res_dev : out_dev) | (uintptr_t)(bias_dev ? bias_dev : out_dev)) & 15) == 0;
        const bool tap_x3 = io16 && !(de && de[0] == '0') && !(x3e && x3e[0] == '0') && kh == 3 && kw == 3 && stride == 1 && pad == 1 && ((cin == 64 && cout <= 64) || (cin == 256 && cout <= 32)) && ldx == cin &&
                            (((uintptr_t)x_dev | (uintptr_t)w_packed_dev) & 15) == 0 && xbytes < (1ll << 31) && (!res_dev || ldr >= cout) &&
                            n * (int64_t)h * w * (ldo > ldr ? ldo : ldr) * 4 < (1ll << 31) && n * (int64_t)h * w >= 65536;
        // wide 3 x 3 layers on small maps: both operands split into planes once, implicit GEMM on the bf16 pipe (conv_x3_gemm_kernel)
        const int64_t in_elems = n * (int64_t)h * w * cin;
        const bool x3gemm = io16 && !(de && de[0] == '0') && !(x3e && x3e[0] == '0') && kh == 3 && kw == 3 && stride == 1 && pad == 1 && (cin == 72 || cin == 144) &&
                            ldx == cin && cout >= 48 && cout <= 160 && (((uintptr_t)x_dev | (uintptr_t)w_packed_dev) & 15) == 0 && in_elems * 6 < (1ll << 31) &&
                            in_elems <= (16ll << 20) && (!res_dev || ldr >= cout) && n * (int64_t)h * w >= 8192 && n * (int64_t)h * w < (1ll << 31);
        if (x3gemm) {
            const int k = 9 * cin, nks = (k + 31) / 32, kp = nks * 32;
            const int groups = (cout + 79) / 80, wr_total = groups * 80;   // channel groups of 80 rows (grid.y)
            const size_t xplane = (size_t)in_elems * 2, wplane = (size_t)wr_total * kp * 2;
            const size_t xbytes3 = (3 * xplane + 255) / 256 * 256, need = xbytes3 + 3 * wplane;
            std::lock_guard<std::mutex> scratch_lock(g_conv_mutex);   // one plane scratch per device: one stream at a time (vsc_hip.h)
            Scratch &sx = g_x3[dev];
            if (sx.bytes < need) {
                if (sx.ptr) {
                    VSC_CHECK_HIP(hipDeviceSynchronize());
                    VSC_CHECK_HIP(hipFree(sx.ptr));
                    sx.ptr = nullptr;
                    sx.bytes = 0;
                }
                hipError_t e = hipMalloc(&sx.ptr, need);
                if (e != hipSuccess) {
                    sx.ptr = nullptr;
                    vsc_set_error("conv2d: hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
                    return VSC_ERR_NOMEM;
                }
                sx.bytes = need;
            }
            char *xpl = (char *)sx.ptr, *wpl = xpl + xbytes3;
            hipLaunchKernelGGL(split_planes_kernel, dim3(blocks_for(in_elems / 4)), dim3(256), 0, stream, x_dev, xpl, in_elems / 4, (int64_t)xplane);
            hipLaunchKernelGGL(split_weight_rows_kernel, dim3(blocks_for((int64_t)wr_total * kp / 4)), dim3(256), 0, stream, w_packed_dev, wpl, cout, kpad, k, wr_total, kp);
            X3GemmArgs a{xpl, wpl, bias_dev, res_dev, out_dev, n * (int64_t)h * w, (int)n, h, w, cin, cout, ldo, ldr, act, nks, kp, cin / 8, wr_total,
                         (unsigned)xplane, (unsigned)wplane};
            const dim3 grid((unsigned)((a.rows + 127) / 128), groups);
            g_conv_last_pipe = 1;
            hipLaunchKernelGGL((conv_x3_gemm_kernel<5>), grid, dim3(512), 0, stream, a);
            VSC_CHECK_LAUNCH();
            return VSC_OK;
        }
        if (tap_x3) {
            DirectArgs a{x_dev, w_packed_dev, bias_dev, res_dev, out_dev, (int)n, h, w, ldx, cout, kpad, ldo, ldr, act,
                         (w + 31) / 32, (h + 3) / 4, 0, (unsigned)xbytes, cin / 4};
            a.ntiles = (int64_t)a.tiles_x * a.tiles_y * n;
            static int cus_tap[16] = {};
            if (!cus_tap[dev]) VSC_CHECK_HIP(hipDeviceGetAttribute(&cus_tap[dev], hipDeviceAttributeMultiprocessorCount, dev));
            const unsigned grid = (unsigned)(a.ntiles < cus_tap[dev] ? a.ntiles : cus_tap[dev]);
            g_conv_last_pipe = 1;
            if (cin == 256) hipLaunchKernelGGL((conv3x3_tap_x3_kernel<8, 2, 4>), dim3(grid), dim3(512), 0, stream, a);
            else hipLaunchKernelGGL((conv3x3_tap_x3_kernel<8, 4, 1>), dim3(grid), dim3(512), 0, stream, a);
            VSC_CHECK_LAUNCH();
            return VSC_OK;
        }
        const bool direct = io16 && !(de && de[0] == '0') && kh == 3 && kw == 3 && stride == 1 && pad == 1 && cout <= 40 && (cin == 20 || cin == 36) &&
                            (ldx & 3) == 0 && (((uintptr_t)x_dev | (uintptr_t)w_packed_dev) & 15) == 0 && xbytes < (1ll << 31) &&
                            (!res_dev || ldr >= cout);
        if (direct) {
            DirectArgs a{x_dev, w_packed_dev, bias_dev, res_dev, out_dev, (int)n, h, w, ldx, cout, kpad, ldo, ldr, act,
                         (w + 31) / 32, (h + 7) / 8, 0, (unsigned)xbytes, cin / 4};
            a.ntiles = (int64_t)a.tiles_x * a.tiles_y * n;
            static int cus_direct[16] = {};
            if (!cus_direct[dev]) VSC_CHECK_HIP(hipDeviceGetAttribute(&cus_direct[dev], hipDeviceAttributeMultiprocessorCount, dev));
            const bool two = cin == 20 && cout <= 32;   // workgroups per CU (LDS)
            const int64_t resident = (two ? 2ll : 1ll) * cus_direct[dev];
            const unsigned grid = (unsigned)(a.ntiles < resident ? a.ntiles : resident);
            const char *x3 = vsc_opt(OPT_CONV_X3);   // diagnostic / test switch: 0 = the fp32-pipe kernels everywhere
            if (!(x3 && x3[0] == '0') && cin == 36 && cout <= 36 && n * (int64_t)h * w * (ldo > ldr ? ldo : ldr) * 4 < (1ll << 31)) {
                DirectArgs b = a;   // 4 x 32 tiles, one workgroup per CU
                b.tiles_y = (h + 3) / 4;
                b.ntiles = (int64_t)b.tiles_x * b.tiles_y * n;
                const unsigned g36 = (unsigned)(b.ntiles < cus_direct[dev] ? b.ntiles : cus_direct[dev]);
                g_conv_last_pipe = 1;
                hipLaunchKernelGGL((conv3x3_direct_x3_kernel<5, 36, 3, 1>), dim3(g36), dim3(512), 0, stream, b);
                VSC_CHECK_LAUNCH();
                return VSC_OK;
            }
            if (!(x3 && x3[0] == '0') && cin == 20 && cout <= 20 && n * (int64_t)h * w * (ldo > ldr ? ldo : ldr) * 4 < (1ll << 31)) {
                g_conv_last_pipe = 1;
                hipLaunchKernelGGL((conv3x3_direct_x3_kernel<3, 20, 2, 2>), dim3(grid), dim3(512), 0, stream, a);
            }
            else if (cin == 20 && cout <= 32) hipLaunchKernelGGL((conv3x3_direct_kernel<5, 1, 32>), dim3(grid), dim3(512), 0, stream, a);
            else if (cin == 20) hipLaunchKernelGGL((conv3x3_direct_kernel<5, 2, 40>), dim3(grid), dim3(512), 0, stream, a);
            else if (cout <= 32) hipLaunchKernelGGL((conv3x3_direct_kernel<9, 1, 32>), dim3(grid), dim3(512), 0, stream, a);
            else hipLaunchKernelGGL((conv3x3_direct_kernel<9, 2, 40>), dim3(grid), dim3(512), 0, stream, a);
            VSC_CHECK_LAUNCH();
            return VSC_OK;
        }
    }
    // 3 -> 8 / 16 / 32 channels, 3 x 3: the direct stem kernel (conv_stem3_kernel)
    {
        const char *se = vsc_opt(OPT_CONV_STEM);   // diagnostic / test switch: 0 = the GEMM path
        const bool stem = !(se && se[0] == '0') && cin == 3 && ldx == 3 && kh == 3 && kw == 3 && (cout == 8 || cout == 16 || cout == 32) && !res_dev &&
                          (ldo & 3) == 0 && (((uintptr_t)out_dev | (uintptr_t)(bias_dev ? bias_dev : out_dev)) & 15) == 0 && rows < (1ll << 31) * 256;
        if (stem) {
            const dim3 grid((unsigned)((rows + 255) / 256));
#define VSC_STEM(C) hipLaunchKernelGGL(conv_stem3_kernel<C>, grid, dim3(256), 0, stream, x_dev, w_packed_dev, bias_dev, out_dev, rows, h, w, kpad, stride, pad, ho, wo, ldo, act)
            if (cout == 8) VSC_STEM(8); else if (cout == 16) VSC_STEM(16); else VSC_STEM(32);
#undef VSC_STEM
            VSC_CHECK_LAUNCH();
            return VSC_OK;
        }
    }
    // 1 x 1 expansion from 64 channels: the streaming kernel (conv1x1_expand64_kernel)
    {
        const char *ee = vsc_opt(OPT_CONV_EXPAND);   // diagnostic / test switch: 0 = the tile kernels
        const bool expand = !(ee && ee[0] == '0') && kh == 1 && kw == 1 && stride == 1 && pad == 0 && cin == 64 && ldx == 64 && cout >= 128 &&
                            cout <= 256 && (cout & 31) == 0 && rows >= 128 * 256 && rows * (int64_t)ldo * 4 < (1ll << 31) &&
                            (!res_dev || rows * (int64_t)ldr * 4 < (1ll << 31)) && (((uintptr_t)x_dev | (uintptr_t)w_packed_dev) & 15) == 0;
        if (expand) {
            static int cus_expand[16] = {};
            if (!cus_expand[dev]) VSC_CHECK_HIP(hipDeviceGetAttribute(&cus_expand[dev], hipDeviceAttributeMultiprocessorCount, dev));
            ExpandArgs a{x_dev, w_packed_dev, bias_dev, res_dev, out_dev, rows, (rows + 127) / 128, cout, ldo, ldr, act};
            const unsigned grid = (unsigned)(a.ntiles < cus_expand[dev] ? a.ntiles : cus_expand[dev]);
            hipLaunchKernelGGL(conv1x1_expand64_kernel, dim3(grid), dim3(512), 0, stream, a);
            VSC_CHECK_LAUNCH();
            return VSC_OK;
        }
    }
    // pointwise layers with <= 96 input channels and >= 2 x as many outputs on many pixels: the streaming kernel for any cin % 4 == 0
    {
        const char *ee = vsc_opt(OPT_CONV_EXPAND);
        const char *nm2 = vsc_opt(OPT_CONV_STREAM_MIN_COUT);   // diagnostic: smallest cout taken (default 2 cin)
        const int cp = (cin + 7) / 8;
        const bool strm = !(ee && ee[0] == '0') && kh == 1 && kw == 1 && stride == 1 && pad == 0 && ldx == cin && (cin & 3) == 0 && cin >= 16 && cin <= 96 &&
                          cout >= (nm2 ? atoi(nm2) : 2 * cin) && rows >= 128 * 256 && rows * (int64_t)ldo * 4 < (1ll << 31) && rows * (int64_t)cin * 4 < (1ll << 31) &&
                          (!res_dev || rows * (int64_t)ldr * 4 < (1ll << 31)) && (((uintptr_t)x_dev | (uintptr_t)w_packed_dev) & 15) == 0 &&
                          (cp == 2 || cp == 3 || cp == 5 || cp == 6 || cp == 11 || cp == 12);
        if (strm) {
            static int cus_strm[16] = {};
            if (!cus_strm[dev]) VSC_CHECK_HIP(hipDeviceGetAttribute(&cus_strm[dev], hipDeviceAttributeMultiprocessorCount, dev));
            const int tp = cp <= 6 ? 128 : 64, groups = (cout + 127) / 128;
            ExpandArgs a{x_dev, w_packed_dev, bias_dev, res_dev, out_dev, rows, (rows + tp - 1) / tp, cout, ldo, ldr, act};
            const int resident = cus_strm[dev] * (cp <= 3 ? 4 : 3);
            const int per = resident / groups > 0 ? resident / groups : 1;
            const dim3 grid((unsigned)(a.ntiles < per ? a.ntiles : per), groups);
#define VSC_STRM(C) hipLaunchKernelGGL(conv1x1_stream_kernel<C>, grid, dim3(256), 0, stream, a, cin, kpad)
            switch (cp) { case 2: VSC_STRM(2); break; case 3: VSC_STRM(3); break; case 5: VSC_STRM(5); break; case 6: VSC_STRM(6); break;
                          case 11: VSC_STRM(11); break; default: VSC_STRM(12); }
#undef VSC_STRM
            VSC_CHECK_LAUNCH();
            return VSC_OK;
        }
    }
    // 1 x 1, stride 1, dense rows of a multiple of 32 channels: the input IS the patch matrix
    const bool in_place = kh == 1 && kw == 1 && stride == 1 && pad == 0 && ldx == cin && (cin % KS) == 0 && (((uintptr_t)x_dev) & 15) == 0;
    const char *nm = vsc_opt(OPT_CONV_NARROW_MAX);
    const bool narrow = cout <= (nm ? atoi(nm) : 160);   // <= 5 channel tiles of 32 (VSC_CONV_NARROW_MAX): the thin and the 144-wide layers (see conv_gemm_narrow_kernel)
    // patches gathered inside the GEMM's staging (narrow kernel): 4-channel chunks, table-sized K, 16-bit image coordinates
    const char *imp_env = vsc_opt(OPT_CONV_IMPLICIT);   // diagnostic / test switch, read per call
    const bool no_implicit = imp_env && imp_env[0] == '0';
    const bool implicit = !in_place && narrow && !no_implicit && (cin & 3) == 0 && (ldx & 3) == 0 && (((uintptr_t)x_dev) & 15) == 0 &&
                          kpad <= IM2COL_TABLE && kh < 16 && kw < 16 && h < 32000 && w < 32000 && pad < 1000;
    // The per-device scratch (zero line, patch matrix, CU count) is guarded; the patch matrix itself is ONE buffer per
    // device, so convolutions on one device must be issued from one stream at a time (documented in vsc_hip.h).
    std::lock_guard<std::mutex> scratch_lock(g_conv_mutex);
    if (implicit && !g_zero_line[dev]) {
        VSC_CHECK_HIP(hipMalloc((void **)&g_zero_line[dev], 256));
        // filled on the NULL stream; callers' streams may be non-blocking (PyTorch side streams), so wait here, once
        VSC_CHECK_HIP(hipMemset(g_zero_line[dev], 0, 256));
        VSC_CHECK_HIP(hipDeviceSynchronize());
    }
    Scratch &s = g_patch[dev];
    const size_t need = in_place || implicit ? 0 : (size_t)rows * kpad * 4;
    if (s.bytes < need) {
        if (s.ptr) {
            VSC_CHECK_HIP(hipDeviceSynchronize());
            VSC_CHECK_HIP(hipFree(s.ptr));
            s.ptr = nullptr;
            s.bytes = 0;
        }
        hipError_t e = hipMalloc(&s.ptr, need);
        if (e != hipSuccess) {
            s.ptr = nullptr;
            vsc_set_error("conv2d: hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
            return VSC_ERR_NOMEM;
        }
        s.bytes = need;
    }
    if (in_place || implicit) {
        // nothing to materialise
    } else if (kpad <= IM2COL_TABLE && kh < 16 && kw < 16 && n * ho < (1ll << 31)) {
        const int per_row = wo * (kpad / 4);
        int bx = (per_row + 255) / 256;
        bx = bx > 64 ? 64 : bx;
        hipLaunchKernelGGL(im2col_pack_rows_kernel, dim3((unsigned)(n * ho), bx), dim3(256), 0, stream, x_dev, (float *)s.ptr, h, w, cin,
                           ldx, kh, kw, stride, pad, ho, wo, k, kpad);
    } else {
        hipLaunchKernelGGL(im2col_pack_kernel, dim3(blocks_for(rows * (kpad / 4))), dim3(256), 0, stream, x_dev, (float *)s.ptr, rows,
                           h, w, cin, ldx, kh, kw, stride, pad, ho, wo, k, kpad);
    }
    VSC_CHECK_LAUNCH();
    static int cus_of[16] = {};
    if (!cus_of[dev]) VSC_CHECK_HIP(hipDeviceGetAttribute(&cus_of[dev], hipDeviceAttributeMultiprocessorCount, dev));
    const char *pe = vsc_opt(OPT_CONV_PERSIST);   // diagnostic: 0 = one tile per workgroup
    const char *se = vsc_opt(OPT_CONV_STAGES);    // diagnostic: 3 = three LDS stages, one workgroup per CU
    const char *we = vsc_opt(OPT_CONV_WAVES);     // diagnostic: 4 = four waves per workgroup
    const char *te = vsc_opt(OPT_CONV_NARROW_NT); // diagnostic: 1 / 2 = 128- / 256-pixel tiles for every narrow layer
    const int stages = se && se[0] == '3' ? 3 : 2;
    const int nw = we && we[0] == '4' ? 4 : 8;
    const int tiles_c = narrow ? (cout + NARROW_C - 1) / NARROW_C : (cout + TR - 1) / TR;
    // 128-pixel tiles (four waves, three workgroups per CU) when the 256-pixel list spills a partial second round over the chip's
    // 2 x CUs slots (72 @ 56 x 56: 588 tiles, 100 -> 95 us) or leaves most CUs without a workgroup (144 -> 20 @ 28 x 28: 49 tiles,
    // 18 -> 13 us); measured worse for lists of 245 / 392 (one round either way) and >= 1568 tiles
    const int64_t tiles256 = ((rows + 255) / 256) * tiles_c;
    const bool small = narrow && nw == 8 &&
                       (te ? te[0] == '1' : ((tiles256 > 2ll * cus_of[dev] && tiles256 < 4ll * cus_of[dev]) || 2 * tiles256 <= cus_of[dev]));
    const int narrow_p = small ? 128 : 256;
    const int64_t tiles_p = narrow ? (rows + narrow_p - 1) / narrow_p : (rows + TQ - 1) / TQ;
    VSC_REQUIRE(tiles_p * tiles_c < (1ll << 31), "conv2d: grid too large");
    ConvGemmArgs a{w_packed_dev, in_place ? x_dev : (const float *)s.ptr, bias_dev, res_dev, out_dev, rows, cout, kpad, ldo, ldr, act, tiles_c, tiles_p,
                   x_dev, g_zero_line[dev], h, w, cin, ldx, kh, kw, stride, pad, ho, wo, k, 0};
    if (const char *e = vsc_opt(OPT_CONV_REMAP)) a.no_remap = e[0] == '0';
    const int64_t resident = (small ? (stages == 3 ? 2ll : 3ll) : stages == 2 ? 2ll : 1ll) * cus_of[dev];   // LDS per workgroup: 44 / 64 KiB (128 pixels), 76 / 112 KiB
    const unsigned ngrid = (unsigned)((pe && pe[0] == '0') || tiles_p * tiles_c < resident ? tiles_p * tiles_c : resident);
    if (narrow) {
#define VSC_NARROW(I, S, W) hipLaunchKernelGGL((conv_gemm_narrow_kernel<I, S, W>), dim3(ngrid), dim3(W * 64), 0, stream, a)
        if (small && stages == 3) { if (implicit) hipLaunchKernelGGL((conv_gemm_narrow_kernel<true, 3, 4, 1>), dim3(ngrid), dim3(256), 0, stream, a);
                                    else hipLaunchKernelGGL((conv_gemm_narrow_kernel<false, 3, 4, 1>), dim3(ngrid), dim3(256), 0, stream, a); }
        else if (small) { if (implicit) hipLaunchKernelGGL((conv_gemm_narrow_kernel<true, 2, 4, 1>), dim3(ngrid), dim3(256), 0, stream, a);
                          else hipLaunchKernelGGL((conv_gemm_narrow_kernel<false, 2, 4, 1>), dim3(ngrid), dim3(256), 0, stream, a); }
        else if (implicit) { if (stages == 3) VSC_NARROW(true, 3, 4); else if (nw == 4) VSC_NARROW(true, 2, 4); else VSC_NARROW(true, 2, 8); }
        else { if (stages == 3) VSC_NARROW(false, 3, 4); else if (nw == 4) VSC_NARROW(false, 2, 4); else VSC_NARROW(false, 2, 8); }
#undef VSC_NARROW
    } else {
        hipLaunchKernelGGL(conv_gemm_kernel, dim3((unsigned)(tiles_p * tiles_c)), dim3(256), 0, stream, a);
    }
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

extern "C" int vsc_dwconv2d_f32(const float *x_dev, int64_t n, int32_t h, int32_t w, int32_t c, const float *w_dev,
                                const float *bias_dev, int32_t kh, int32_t kw, int32_t stride, int32_t pad, int32_t act,
                                float *out_dev, void *stream_) {
    VSC_REQUIRE(x_dev && w_dev && out_dev, "dwconv2d: null operand");
    VSC_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0, "dwconv2d: bad shape");
    VSC_REQUIRE(act >= VSC_ACT_NONE && act <= VSC_ACT_GELU, "dwconv2d: unknown activation %d", act);
    const int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
    VSC_REQUIRE(ho > 0 && wo > 0, "dwconv2d: empty output");
    const int64_t total = n * ho * wo * c;
    const size_t wbytes = (size_t)kh * kw * c * 4;
    const bool aligned = ((((uintptr_t)x_dev) | ((uintptr_t)out_dev) | (bias_dev ? (uintptr_t)bias_dev : 0)) & 15) == 0;
    const size_t small_bytes = ((size_t)kh * kw + (size_t)h * w) * DW_SLAB * 4;
    const char *se = vsc_opt(OPT_DWCONV_SMALL);   // diagnostic / test switch: 0 = the row kernel
    if ((c & 3) == 0 && aligned && small_bytes <= 40 * 1024 && n < (1ll << 31) && !(se && se[0] == '0')) {
        // small maps: a workgroup per (64-channel slab, image group), about eight workgroups per CU in flight
        const int slabs = (c + DW_SLAB - 1) / DW_SLAB;
        int64_t gy = 2048 / slabs;
        gy = gy < 1 ? 1 : (gy > n ? n : gy);
        hipLaunchKernelGGL(dwconv_small_kernel, dim3(slabs, (unsigned)gy), dim3(256), small_bytes, (hipStream_t)stream_, x_dev, w_dev, bias_dev,
                           out_dev, (int)n, h, w, c, kh, kw, stride, pad, ho, wo, act);
    } else if ((c & 3) == 0 && aligned && wbytes <= 64 * 1024 && n * ho < (1ll << 31)) {
        static bool attr_set[16] = {};
        int dev = 0;
        VSC_CHECK_HIP(hipGetDevice(&dev));
        if (dev >= 16 || !attr_set[dev]) {
            VSC_CHECK_HIP(hipFuncSetAttribute((const void *)dwconv_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            if (dev < 16) attr_set[dev] = true;
        }
        int bx = (wo * (c / 4) + 255) / 256;
        bx = bx > 64 ? 64 : bx;
        hipLaunchKernelGGL(dwconv_rows_kernel, dim3((unsigned)(n * ho), bx), dim3(256), wbytes, (hipStream_t)stream_, x_dev, w_dev, bias_dev,
                           out_dev, h, w, c, kh, kw, stride, pad, ho, wo, act);
    } else
    hipLaunchKernelGGL(dwconv_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream_, x_dev, w_dev, bias_dev, out_dev,
                       total, h, w, c, kh, kw, stride, pad, ho, wo, act);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

extern "C" int vsc_global_avgpool_f32(const float *x_dev, int64_t n, int32_t hw, int32_t c, float *out_dev, void *stream_) {
    VSC_REQUIRE(x_dev && out_dev && n > 0 && n < 65536 && hw > 0 && c > 0, "global_avgpool: bad arguments");
    hipLaunchKernelGGL(avgpool_kernel, dim3((c + 63) / 64, (unsigned)n), dim3(256), 0, (hipStream_t)stream_, x_dev, out_dev, hw, c);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

extern "C" int vsc_channel_scale_f32(float *x_dev, const float *scale_dev, int64_t n, int32_t hw, int32_t c, void *stream_) {
    VSC_REQUIRE(x_dev && scale_dev && n > 0 && hw > 0 && c > 0, "channel_scale: bad arguments");
    const int64_t total = n * hw * c;
    hipLaunchKernelGGL(channel_scale_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream_, x_dev, scale_dev, total, hw, c);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

extern "C" int vsc_se_block_f32(float *x_dev, int64_t n, int32_t hw, int32_t c, const float *w1_packed_dev, const float *b1_dev, int32_t cr,
                                const float *w2_packed_dev, const float *b2_dev, int32_t act1, int32_t act2, void *stream_) {
    VSC_REQUIRE(x_dev && w1_packed_dev && w2_packed_dev && n > 0 && n < (1ll << 31) && hw > 0 && c > 0 && cr > 0, "se_block: bad arguments");
    VSC_REQUIRE(act1 >= VSC_ACT_NONE && act1 <= VSC_ACT_GELU && act2 >= VSC_ACT_NONE && act2 <= VSC_ACT_GELU, "se_block: unknown activation");
    VSC_REQUIRE((c & 3) == 0 && (((uintptr_t)x_dev) & 15) == 0, "se_block: channels must be a multiple of 4 and x 16-byte aligned");
    const size_t lds = ((size_t)hw * c + 2 * c + cr) * 4;
    VSC_REQUIRE(lds <= 150 * 1024, "se_block: a %d x %d map does not fit LDS (%zu bytes): use avgpool + conv2d + channel_scale", hw, c, lds);
    static bool attr_set[16] = {};
    int dev = 0;
    VSC_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        VSC_CHECK_HIP(hipFuncSetAttribute((const void *)se_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    const int kpad1 = vsc_conv_packed_k(c, 1, 1), kpad2 = vsc_conv_packed_k(cr, 1, 1);
    hipLaunchKernelGGL(se_block_kernel, dim3((unsigned)n), dim3(512), lds, (hipStream_t)stream_, x_dev, w1_packed_dev, b1_dev, kpad1, w2_packed_dev, b2_dev,
                       kpad2, hw, c, cr, act1, act2);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

extern "C" int vsc_upsample_sum_f32(const float *base_dev, int32_t ldb, const float *src0_dev, int32_t factor0, const float *src1_dev,
                                    int32_t factor1, const float *src2_dev, int32_t factor2, int64_t n, int32_t h, int32_t w, int32_t c,
                                    int32_t act, float *out_dev, int32_t ldo, void *stream_) {
    VSC_REQUIRE(out_dev && n > 0 && h > 0 && w > 0 && c > 0 && ldo >= c && (!base_dev || ldb >= c), "upsample_sum: bad arguments");
    VSC_REQUIRE(act >= VSC_ACT_NONE && act <= VSC_ACT_GELU, "upsample_sum: unknown activation %d", act);
    VSC_REQUIRE((c & 3) == 0 && (ldo & 3) == 0 && (!base_dev || (ldb & 3) == 0), "upsample_sum: channel counts / row pitches must be multiples of 4");
    VSC_REQUIRE(n * h < (1ll << 31) && (int64_t)w * c < (1ll << 31), "upsample_sum: grid too large");
    UpsampleSumArgs a{};
    a.base = base_dev, a.out = out_dev, a.h = h, a.w = w, a.c4 = c / 4, a.ldb = ldb, a.ldo = ldo, a.act = act;
    const float *srcs[3] = {src0_dev, src1_dev, src2_dev};
    const int factors[3] = {factor0, factor1, factor2};
    uintptr_t align = (uintptr_t)out_dev | (uintptr_t)base_dev;
    for (int k = 0; k < 3; ++k) {
        if (!srcs[k]) continue;
        const int f = factors[k];
        VSC_REQUIRE(f >= 1 && (f & (f - 1)) == 0 && h % f == 0 && w % f == 0, "upsample_sum: factor %d must be a power of two dividing %d x %d", f, h, w);
        a.src[a.nsrc] = srcs[k];
        a.shift[a.nsrc++] = __builtin_ctz((unsigned)f);
        align |= (uintptr_t)srcs[k];
    }
    VSC_REQUIRE((align & 15) == 0, "upsample_sum: operands must be 16-byte aligned");
    VSC_REQUIRE(a.base || a.nsrc, "upsample_sum: nothing to sum");
    hipLaunchKernelGGL(upsample_sum4_kernel, dim3((unsigned)((w * a.c4 + 255) / 256), (unsigned)(n * h)), dim3(256), 0, (hipStream_t)stream_, a);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

extern "C" int vsc_upsample_add_f32(const float *src_dev, int64_t n, int32_t h, int32_t w, int32_t c, int32_t factor,
                                    float *out_dev, int32_t ldo, int32_t coff, int32_t accumulate, int32_t act, void *stream_) {
    VSC_REQUIRE(src_dev && out_dev && n > 0 && h > 0 && w > 0 && c > 0 && factor >= 1, "upsample_add: bad arguments");
    VSC_REQUIRE(h % factor == 0 && w % factor == 0, "upsample_add: %d x %d is not a multiple of the factor %d", h, w, factor);
    VSC_REQUIRE(ldo >= coff + c && coff >= 0, "upsample_add: channel window outside the output row");
    VSC_REQUIRE(act >= VSC_ACT_NONE && act <= VSC_ACT_GELU, "upsample_add: unknown activation %d", act);
    if ((factor & (factor - 1)) == 0 && (c & 3) == 0 && (ldo & 3) == 0 && (coff & 3) == 0 && (((uintptr_t)src_dev | (uintptr_t)out_dev) & 15) == 0 &&
        n * h < (1ll << 31) && (int64_t)w * c < (1ll << 31))
        return vsc_upsample_sum_f32(accumulate ? out_dev + coff : nullptr, ldo, src_dev, factor, nullptr, 0, nullptr, 0, n, h, w, c, act,
                                    out_dev + coff, ldo, stream_);
    const int64_t total = n * h * w * c;
    hipLaunchKernelGGL(upsample_add_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream_, src_dev, out_dev, total, h,
                       w, c, factor, ldo, coff, accumulate, act);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

extern "C" int vsc_attention_f32_batch(const float *qkv_dev, float *out_dev, int32_t tokens, int32_t heads, int32_t head_dim, int32_t seqs,
                                       void *stream_) {
    VSC_REQUIRE(qkv_dev && out_dev && tokens > 0 && heads > 0 && head_dim > 0, "attention_f32: bad arguments");
    VSC_REQUIRE(tokens <= 8192 && heads < 65536 && seqs >= 1 && seqs < 65536, "attention_f32: %d tokens / %d heads / %d sequences unsupported", tokens, heads, seqs);
    const size_t smem = (size_t)(tokens + 5 * head_dim) * 4;
    VSC_REQUIRE(smem <= 48 * 1024, "attention_f32: %d tokens x head_dim %d exceeds the kernel's LDS budget", tokens, head_dim);
    hipLaunchKernelGGL(attention_f32_kernel, dim3(tokens, heads, seqs), dim3(256), smem, (hipStream_t)stream_, qkv_dev, out_dev, tokens, heads,
                       head_dim);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

extern "C" int vsc_attention_f32_varlen(const float *qkv_dev, float *out_dev, const int32_t *row_offsets_dev, int32_t seqs, int32_t max_tokens,
                                        int32_t heads, int32_t head_dim, void *stream_) {
    VSC_REQUIRE(qkv_dev && out_dev && row_offsets_dev && max_tokens > 0 && heads > 0 && head_dim > 0, "attention_f32_varlen: bad arguments");
    VSC_REQUIRE(max_tokens <= 8192 && heads < 65536 && seqs >= 1 && seqs < 65536, "attention_f32_varlen: %d tokens / %d heads / %d sequences unsupported", max_tokens, heads, seqs);
    const size_t smem = (size_t)(max_tokens + 5 * head_dim) * 4;
    VSC_REQUIRE(smem <= 48 * 1024, "attention_f32_varlen: %d tokens x head_dim %d exceeds the kernel's LDS budget", max_tokens, head_dim);
    hipLaunchKernelGGL(attention_f32_kernel, dim3(max_tokens, heads, seqs), dim3(256), smem, (hipStream_t)stream_, qkv_dev, out_dev, max_tokens, heads,
                       head_dim, row_offsets_dev);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

extern "C" int vsc_attention_f32(const float *qkv_dev, float *out_dev, int32_t tokens, int32_t heads, int32_t head_dim, void *stream_) {
    return vsc_attention_f32_batch(qkv_dev, out_dev, tokens, heads, head_dim, 1, stream_);
}
