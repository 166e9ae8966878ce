// Fused MLP of a Swin-V2 block at C = 512 (stage 2 of Swin-V2-B: 18 of its 24 blocks), res-post-norm
// (train/train_v115/torch2scripts.py:18-35 Mlp, :297-300 the block's second half):
//     x += LayerNorm(GELU(xb W1^T + b1) W2^T + b2) * gamma + beta,     xb = bf16(x)
// in ONE kernel.  As two persistent GEMM launches (fc1 with the GELU write-out, fc2 with the LN_RES write-out) the hidden
// tensor [M, 2048] (268 MB per 256 frames) goes out to HBM and comes back, and both launches are MFMA time PLUS byte time.
//
// Why the narrow-stage kernel (swin_mlp.hip) does not stretch to 512: a wave that owns 16 rows needs a W fragment read from
// LDS for every MFMA (LDS-bound at 100 % matrix rate), and 32 rows x 512 output columns are 256 accumulator registers -- the
// whole budget of a wave at two waves per SIMD.  Hence ONE wave per SIMD with the full 512-register file:
//   * 4 waves per workgroup, a wave owns 32 rows from the first load to the last store (nothing is exchanged between waves);
//     oacc[2][32] (32 rows x 512 columns, 256 registers -- the AGPR half of the file), the rows' xb as MFMA operands (128
//     VGPRs), everything else in the remaining 128;
//   * the hidden axis is walked in chunks of 32: W1[chunk, :] (32 x 512) and W2[:, chunk] (512 x 32), 32 KiB each, in two
//     two-slot LDS-DMA rings (128 KiB); every fragment read from LDS feeds two MFMAs (the two 16-row tiles of the wave);
//   * software pipeline across chunks inside the wave, one workgroup barrier per chunk:
//         iteration i:  [ GELU(i): one run of vector instructions ]  then  [ GEMM 2 of chunk i, GEMM 1 of chunk i+1: one run of MFMAs ]
//     -- the matrix and the vector pipe do not overlap on this part and a vector instruction right behind an MFMA costs 16 cycles
//     where it costs 5 behind another vector instruction (tools/micro/single_wave_issue.hip), so the two kinds are kept apart; the
//     first LDS round trip behind the barrier lands under the GELU, the LDS-DMA's scalar set-up sits in MFMA shadows;
//   * GEMM 1's accumulator layout (row = lane & 15, hidden 16 j + 4 (lane >> 4) + r) is GEMM 2's B operand once the contraction
//     slots are assigned as in swin_mlp.hip; fc2.weight is packed chunk-major in that order by the host
//     (swin_mlp512_pack_w2, at model load), so a chunk is one contiguous 32 KiB piece;
//   * LayerNorm over the wave's own rows (two-pass, a row sits in four lanes), residual rows fetched in blocks under the
//     statistics, x / shadow written as whole 128- / 64-byte pieces per row.
// Variants of the body (template parameter V; csrc/gen_mlp512_loop.py): 1 = the attention projection + LayerNorm + residual in front
// (sixteen 32-column chunks of Wp on GEMM 1's machinery; x1 once through memory as fp32, its shadow in registers only); 9 = that + the
// NEXT block's qkv Linear behind (24 double chunks of Wqkv; the new shadow is the B operand where it sits, it is not written at all).
// The kernel is a loop over 128-row tiles (grid = tiles by default).
// LDS layouts, conflict-free for ds_read_b128 (tools/micro/lds_swizzle_check.py, tests/test_capi_symbols.py):
//   W1 chunk  [32 rows][1024 B]   16-byte piece index ^= row & 15
//   W2 chunk  [512 rows][64 B]    16-byte piece index ^= (row & 1) | ((row >> 3) & 1) << 1   (ds_read_b128 serves lanes 0-3, 12-15, 20-27 / 4-11,
//                                 16-19, 28-31 of a half-wave together: the rule has to hold across quads)
// LDS-DMA writes lane-linear, so both permutations are applied to the per-lane SOURCE address.
#include "common.h"
#include "gelu_poly.h"
#include "swin_mlp512_loop.inc"

namespace {

typedef __attribute__((address_space(3))) void *lptr_t;
typedef __attribute__((address_space(3))) const char *ldsc_t;            // 32-bit LDS address
typedef __attribute__((address_space(3))) const bf16x8_t *ldsfrag_t;

struct Mlp512Args {
    const uint16_t *w1;    // [2048, 512]
    const float *b1;       // [2048]
    const uint16_t *w2c;   // [64 chunks][512][32]: fc2.weight chunk-major, hidden axis in consumption order (swin_mlp512_pack_w2)
    const float *b2, *gamma, *beta;   // [512]
    float *x;              // [m, 512] residual stream, updated in place
    uint16_t *xb;          // [m, 512] its bf16 shadow: the MLP's input, replaced by the shadow of the new x
    int64_t m;
    float eps;
    uint32_t *dbg;         // timing variant: [workgroups][4 waves][8] cycle counters (nullptr otherwise)
    // PROJ (variant 1): the attention projection and its LayerNorm in front, in the same kernel --
    //     x1 = x + LayerNorm(att Wp^T + bp) * gamma1 + beta1;  then the MLP block on x1
    const uint16_t *att;   // [m, 512] attention output
    const uint16_t *wp;    // [512, 512] attn.proj.weight
    const float *bp, *gamma1, *beta1;   // [512]
    // QKV (variant 9, with PROJ): the NEXT block's qkv Linear behind the block -- qkv = bf16(x) Wq^T + bq; the shadow xb is then not written
    const uint16_t *wq;    // [1536, 512] the next block's attn.qkv.weight
    const float *bq;       // [1536] (q_bias | 0 | v_bias)
    uint16_t *qkv;         // [m, 1536]
};

constexpr int C = 512, H = 4 * C, NW = 4, MT = 2, RW = 16 * MT, R = NW * RW, HC = 32;
constexpr int SLOT = HC * C * 2;   // one chunk of either matrix: 32 KiB
constexpr int LDS_W1 = 0, LDS_W2 = 2 * SLOT, LDS_B1 = 4 * SLOT, LDS_VEC = LDS_B1 + H * 4, LDS_BYTES = LDS_VEC + 9 * C * 4;   // vectors: b2 | gamma | beta | bp | gamma1 | beta1 | bq [1536]
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

// V: variant of the generated body (gen_mlp512_loop.py VARIANTS): 0 the MLP, 1 proj + LayerNorm + MLP; 2 .. 8 (ablations, cycle counters)
// exist in -DVSC_MLP_ABLATION builds only (make EXTRA=-DVSC_MLP_ABLATION; tools/micro/mlp512_variants.py)
template <int V>
__global__ __launch_bounds__(NW * 64, 1) void swin_mlp512_kernel(Mlp512Args p) {
    lp_kernel_entry();
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float *b1s = (float *)(lds + LDS_B1), *b2s = (float *)(lds + LDS_VEC), *gs = b2s + C, *bs = gs + C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, quad = lane >> 4;

    // ---- operands of the kernel body (swin_mlp512_loop.inc, generated by gen_mlp512_loop.py: the register map is there)
    // LDS-DMA of one chunk (32 KiB, contiguous in memory): instruction q = 4 qq + wave covers the chunk's LDS bytes [1024 q, 1024 q + 1024);
    // a per-lane byte offset from five registers + a scalar offset, no address arithmetic in the loop
    const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc((void *)p.w1, 0, H * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc((void *)p.w2c, 0, H * C * 2, 0x00020000);
    // x / xb rows through descriptors over the m rows that exist: rows past m return zeros and their stores are dropped
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)(uint32_t)(p.m * C * 4), 0x00020000);
    // (QKV: the qkv rows [m, 1536] stand in the shadow's descriptor, which that variant never writes)
    const __amdgpu_buffer_rsrc_t xbr = V >= 9 ? __builtin_amdgcn_make_buffer_rsrc((void *)p.qkv, 0, (int)(uint32_t)(p.m * 3 * C * 2), 0x00020000)
                                              : __builtin_amdgcn_make_buffer_rsrc((void *)p.xb, 0, (int)(uint32_t)(p.m * C * 2), 0x00020000);
    uint32_t v1[4];   // W1: LDS row q = chunk row q, LDS piece `lane` <- source piece lane ^ (q & 15), q & 15 = 4 (qq & 3) + wave
#pragma unroll
    for (int a = 0; a < 4; ++a) v1[a] = (uint32_t)((lane ^ (4 * a + wave)) << 4);
    // W2: sixteen 64-byte rows per instruction: row n = 16 q + (lane >> 2), LDS piece lane & 3 <- source piece (lane & 3) ^ swz(n),
    // swz(n) = (n & 1) | ((n >> 3) & 1) << 1 = ((lane >> 2) & 1) | ((lane >> 5) & 1) << 1 whatever q
    const uint32_t v2 = (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ (((lane >> 2) & 1) | (((lane >> 5) & 1) << 1))) << 4));
    const uint32_t swave = (uint32_t)wave * 1024u;
    const uint32_t ldsbase = (uint32_t)(uintptr_t)(ldsc_t)lds;
    const uint32_t sldsw = __builtin_amdgcn_readfirstlane(ldsbase) + swave;
    // Fragment read bases of the slot pair in use (both rings toggle together: iteration i reads W2 slot (i - 1) & 1 and W1 slot
    // (i + 1) & 1); every fragment address is one of these five registers plus a ds_read immediate:
    //   W1 fragment (j, ks): row 16 j + fr, piece (4 ks + quad) ^ fr = 16 (ks >> 2) + ((4 (ks & 3) + quad) ^ fr)  -> w1p[ks & 3] + 16384 j + 256 (ks >> 2)
    //   W2 fragment jo:      row n = 32 (jo >> 1) + 4 (jo & 1) + 8 (fr >> 2) + (fr & 3), piece quad ^ swz(n), swz(n) = (fr & 1) | ((fr >> 2) & 1) << 1
    //                                                                                                            -> w2p + 2048 (jo >> 1) + 256 (jo & 1)
    const uint32_t vecp = ldsbase + (uint32_t)(LDS_VEC + 32 * quad);   // b2 | gamma | beta (2 KiB apart), this lane's column offset
    const uint32_t bp16 = (uint32_t)((lane ^ 16) << 2), bp32 = (uint32_t)((lane ^ 32) << 2);
    const uint32_t eps = __builtin_amdgcn_readfirstlane(__float_as_uint(p.eps));
    // (QKV: the next block's qkv weight [1536, 512] stands in the timing buffer's descriptor -- the asm has no operand to spare)
    const __amdgpu_buffer_rsrc_t dbgr = V >= 9 ? __builtin_amdgcn_make_buffer_rsrc((void *)p.wq, 0, 3 * C * C * 2, 0x00020000)
                                               : __builtin_amdgcn_make_buffer_rsrc((void *)p.dbg, 0, p.dbg ? (int)(((p.m + R - 1) / R) * NW * 32) : 0, 0x00020000);

    for (int i = tid; i < H; i += NW * 64) b1s[i] = p.b1[i];
    for (int i = tid; i < C; i += NW * 64) {
        b2s[i] = p.b2[i];
        gs[i] = p.gamma[i];
        bs[i] = p.beta[i];
        if (V == 1 || V >= 9) {
            bs[C + i] = p.bp[i];
            bs[2 * C + i] = p.gamma1[i];
            bs[3 * C + i] = p.beta1[i];
        }
        if (V >= 9) {
            bs[4 * C + i] = p.bq[i];
            bs[5 * C + i] = p.bq[C + i];
            bs[6 * C + i] = p.bq[2 * C + i];
        }
    }
    // PROJ: rows of the attention output / attn.proj.weight (other variants: empty descriptors, never used)
    const __amdgpu_buffer_rsrc_t attr = __builtin_amdgcn_make_buffer_rsrc((void *)p.att, 0, p.att ? (int)(uint32_t)(p.m * C * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t wpr = __builtin_amdgcn_make_buffer_rsrc((void *)p.wp, 0, p.wp ? C * C * 2 : 0, 0x00020000);
    // ---- everything else: rows in, the hidden-axis loop, LayerNorm + residual + shadow out -- per 128-row tile; a workgroup walks the
    // tiles blockIdx, blockIdx + gridDim, ... (grid = tiles: one each; a smaller grid (VSC_SWIN_MLP512_GRID) makes the workgroups
    // persistent).  What the body toggles or advances (the fragment bases, the bias pointer) is set up again per tile.
    const int ntiles = (int)((p.m + R - 1) / R);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = (int64_t)tile * R + wave * RW;
    uint32_t w1p[4], w2p, b1p;
#pragma unroll
    for (int a = 0; a < 4; ++a) w1p[a] = ldsbase + (uint32_t)(LDS_W1 + fr * 1024 + (((4 * a + quad) ^ fr) << 4));
    w2p = ldsbase + (uint32_t)(LDS_W2 + (8 * (fr >> 2) + (fr & 3)) * 64 + ((quad ^ ((fr & 1) | (((fr >> 2) & 1) << 1))) << 4));
    b1p = ldsbase + (uint32_t)(LDS_B1 + 16 * quad);    // this lane's four bias values of hidden tile 0 of the chunk at hand
    const uint32_t xboff = (uint32_t)(row0 + fr) * (uint32_t)(C * 2) + 16u * (uint32_t)quad;   // byte offset of row (0, fr), columns 8 quad .., in xb
    const uint32_t dbgoff = ((uint32_t)tile * NW + (uint32_t)wave) * 32u;
    const uint32_t bid = (uint32_t)tile;
    if (tile != (int)blockIdx.x) __syncthreads();   // every wave is done with the previous tile's last ring slots (and the body ends on s_waitcnt vmcnt(0))
#define VSC_MLP512_BODY(K) asm volatile(VSC_MLP512_LOOP_ASM_##K : VSC_MLP512_LOOP_OUTS : VSC_MLP512_LOOP_INS : VSC_MLP512_LOOP_CLOBBERS)
    static_assert(VSC_MLP512_VARIANTS == 10 || VSC_MLP512_VARIANTS == 19, "variant dispatch below (19: VSC_GEN_QKV_ABL=1 python gen_mlp512_loop.py)");
    if (V == 0) VSC_MLP512_BODY(0);
    else if (V == 1) VSC_MLP512_BODY(1);
    else if (V == 9) VSC_MLP512_BODY(9);
#ifdef VSC_MLP_ABLATION   // the ablation variants compute WRONG results and the timing variant writes counters: diagnostic builds only
    else if (V == 2) VSC_MLP512_BODY(2);
    else if (V == 3) VSC_MLP512_BODY(3);
    else if (V == 4) VSC_MLP512_BODY(4);
    else if (V == 5) VSC_MLP512_BODY(5);
    else if (V == 6) VSC_MLP512_BODY(6);
    else if (V == 7) VSC_MLP512_BODY(7);
    else if (V == 8) VSC_MLP512_BODY(8);
#if VSC_MLP512_VARIANTS == 19
    else if (V == 10) VSC_MLP512_BODY(10);
    else if (V == 11) VSC_MLP512_BODY(11);
    else if (V == 12) VSC_MLP512_BODY(12);
    else if (V == 13) VSC_MLP512_BODY(13);
    else if (V == 14) VSC_MLP512_BODY(14);
    else if (V == 15) VSC_MLP512_BODY(15);
    else if (V == 16) VSC_MLP512_BODY(16);
    else if (V == 17) VSC_MLP512_BODY(17);
    else if (V == 18) VSC_MLP512_BODY(18);
#endif
#endif
#undef VSC_MLP512_BODY
    }
}

uint32_t *g_mlp512_dbg = nullptr;   // vsc_debug_mlp512_timing

template <int V>
int launch_k(const Mlp512Args &a, hipStream_t stream) {
    static bool attr_set[16] = {};
    int dev = 0;
    VSC_CHECK_HIP(hipGetDevice(&dev));
    if (dev >= 16 || !attr_set[dev]) {
        VSC_CHECK_HIP(hipFuncSetAttribute((const void *)swin_mlp512_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        if (dev < 16) attr_set[dev] = true;
    }
    int64_t grid = (a.m + R - 1) / R;
    VSC_REQUIRE(grid < (1ll << 31), "swin_mlp512: grid too large");
    if (const char *g = vsc_opt(OPT_SWIN_MLP512_GRID)) {   // experiment: persistent workgroups (e.g. 128: half the chip per launch, the other lane's kernel beside it)
        const int64_t want = atoll(g);
        if (want > 0 && want < grid) grid = want;
    }
    hipLaunchKernelGGL((swin_mlp512_kernel<V>), dim3((unsigned)grid), dim3(NW * 64), LDS_BYTES, stream, a);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

}  // namespace

// fc2.weight [512, 2048] -> [64 chunks][512][32]: chunk ch holds the hidden units 32 ch .. 32 ch + 31 of every output row, in the
// contraction-slot order of the kernel's GEMM 2 (the order swin_mlp_permute_hidden uses inside a 32-block):
//     dst[ch][n][8 g + 4 t + i] = src[n][32 ch + 16 t + 4 g + i]      (g = 0..3, t = 0, 1, i = 0..3)
void swin_mlp512_set_timing_buffer(uint32_t *buf) { g_mlp512_dbg = buf; }

void swin_mlp512_pack_w2(const float *src, float *dst) {
    for (int n = 0; n < C; ++n)
        for (int k = 0; k < H; ++k) {
            const int ch = k >> 5, t = (k >> 4) & 1, g = (k >> 2) & 3, i = k & 3;
            dst[((size_t)ch * C + n) * HC + 8 * g + 4 * t + i] = src[(size_t)n * H + k];
        }
}

int launch_swin_mlp512(const uint16_t *w1, const float *b1, const uint16_t *w2c, const float *b2, const float *gamma, const float *beta,
                       float *x, uint16_t *xb, int64_t m, float eps, hipStream_t stream) {
    VSC_REQUIRE(w1 && b1 && w2c && b2 && gamma && beta && x && xb && m > 0, "swin_mlp512: null/empty");
    VSC_REQUIRE(m < (1ll << 21), "swin_mlp512: %lld rows (x is addressed through one 4-GiB buffer descriptor: < 2^21 rows per call)", (long long)m);
    const Mlp512Args a{w1, b1, w2c, b2, gamma, beta, x, xb, m, eps, g_mlp512_dbg, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
#ifdef VSC_MLP_ABLATION
    if (const char *e = vsc_opt(OPT_SWIN_MLP_ABL)) {   // diagnostic build: another variant of the generated body (ablations give wrong results)
        switch (atoi(e)) {
            case 2: return launch_k<2>(a, stream);
            case 3: return launch_k<3>(a, stream);
            case 4: return launch_k<4>(a, stream);
            case 5: return launch_k<5>(a, stream);
            case 6: return launch_k<6>(a, stream);
            case 7: return launch_k<7>(a, stream);
            case 8: return launch_k<8>(a, stream);
            default: break;
        }
    }
#endif
    return launch_k<0>(a, stream);
}

// proj + LayerNorm + residual + the MLP block of one Swin-V2 block at C = 512 in one launch (variant 1 of the generated body):
//     x1 = x + LayerNorm(att Wp^T + bp) gamma1 + beta1;   x = x1 + LayerNorm(GELU(bf16(x1) W1^T + b1) W2^T + b2) gamma2 + beta2;   xb = bf16(x)
// x1 passes through memory once as fp32 (the second LayerNorm's residual); its bf16 shadow never leaves the registers.
int launch_swin_proj_mlp512(const uint16_t *att, const uint16_t *wp, const float *bp, const float *gamma1, const float *beta1, const uint16_t *w1,
                            const float *b1, const uint16_t *w2c, const float *b2, const float *gamma2, const float *beta2, float *x, uint16_t *xb,
                            int64_t m, float eps, hipStream_t stream) {
    VSC_REQUIRE(att && wp && bp && gamma1 && beta1 && w1 && b1 && w2c && b2 && gamma2 && beta2 && x && xb && m > 0, "swin_proj_mlp512: null/empty");
    VSC_REQUIRE(m < (1ll << 21), "swin_proj_mlp512: %lld rows (x is addressed through one 4-GiB buffer descriptor: < 2^21 rows per call)", (long long)m);
    const Mlp512Args a{w1, b1, w2c, b2, gamma2, beta2, x, xb, m, eps, nullptr, att, wp, bp, gamma1, beta1, nullptr, nullptr, nullptr};
    return launch_k<1>(a, stream);
}

// ... and the NEXT block's qkv Linear behind it (variant 9): qkv_next = bf16(x) Wq^T + bq straight from the registers that hold the new
// shadow, which is then not written to memory at all (xb is untouched).  m * 3072 must fit 32 bits.
int launch_swin_proj_mlp_qkv512(const uint16_t *att, const uint16_t *wp, const float *bp, const float *gamma1, const float *beta1, const uint16_t *w1,
                                const float *b1, const uint16_t *w2c, const float *b2, const float *gamma2, const float *beta2, const uint16_t *wq,
                                const float *bq, float *x, uint16_t *qkv_next, int64_t m, float eps, hipStream_t stream) {
    VSC_REQUIRE(att && wp && bp && gamma1 && beta1 && w1 && b1 && w2c && b2 && gamma2 && beta2 && wq && bq && x && qkv_next && m > 0,
                "swin_proj_mlp_qkv512: null/empty");
    VSC_REQUIRE(m * 3072 < (1ll << 32), "swin_proj_mlp_qkv512: %lld rows (the qkv rows are addressed through one 4-GiB buffer descriptor)", (long long)m);
    const Mlp512Args a{w1, b1, w2c, b2, gamma2, beta2, x, nullptr, m, eps, nullptr, att, wp, bp, gamma1, beta1, wq, bq, qkv_next};
#if defined(VSC_MLP_ABLATION) && VSC_MLP512_VARIANTS == 19
    if (const char *e = vsc_opt(OPT_SWIN_MLP_ABL)) {   // diagnostic build: the QKV phase's ablations (wrong results)
        switch (atoi(e)) {
            case 10: return launch_k<10>(a, stream);
            case 11: return launch_k<11>(a, stream);
            case 12: return launch_k<12>(a, stream);
            case 13: return launch_k<13>(a, stream);
            case 14: return launch_k<14>(a, stream);
            case 15: return launch_k<15>(a, stream);
            case 16: return launch_k<16>(a, stream);
            case 17: return launch_k<17>(a, stream);
            case 18: return launch_k<18>(a, stream);
            default: break;
        }
    }
#endif
    return launch_k<9>(a, stream);
}
