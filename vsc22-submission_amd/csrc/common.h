// Shared device/host helpers for libvsc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/vsc_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // 8 bf16 = 4 VGPRs (MFMA A/B operand)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;   // 8 bytes
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define VSC_WAVE 64

// ---- error plumbing --------------------------------------------------------
void vsc_set_error(const char *fmt, ...);

#define VSC_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            vsc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                          __LINE__);                                                     \
            return VSC_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

#define VSC_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            vsc_set_error(__VA_ARGS__);   \
            return VSC_ERR_INVALID;       \
        }                                 \
    } while (0)

#define VSC_CHECK_LAUNCH() VSC_CHECK_HIP(hipGetLastError())

// ---- diagnostic / test switches ------------------------------------------------------------------
// Every switch that used to be an environment lookup on the launch path.  The environment is read ONCE per process (first use of any
// switch); afterwards a switch changes only through vsc_set_option() (include/vsc_hip.h).  vsc_opt() is an array read.
#define VSC_OPT_LIST(X)                                                                                                  \
    X(ATTN_SKEW) X(ATTN_ABL) X(ATTN_NI) X(ATTN_DMA) X(CONV_IMPLICIT) X(CONV_DIRECT) X(CONV_REMAP) X(CONV_PERSIST) X(CONV_STAGES) X(CONV_WAVES) X(CONV_NARROW_MAX) X(CONV_NARROW_NT) X(CONV_EXPAND) X(DWCONV_SMALL) X(CONV_STEM) X(CONV_STREAM_MIN_COUT) X(CONV_X3)      \
    X(GEMM_GROUP_N) X(GEMM_V4_SKEW) X(GEMM_TIMING_PRINT) X(GEMM_V4) X(GEMM_V4_GRID) X(GEMM_SKEW_NS_PER_K) X(GEMM_CFG)    \
    X(GEMM_V3) X(GEMM_ABL) X(GEMM_V1) X(KNN_TRIG) X(KNN_ABL) X(KNN_PATH) X(KNN_XCD_MAP) X(KNN_DELTA) X(KNN_TAIL) X(RANGE_PATH) X(PAIRMAX_PATH) X(WATTN_ABL)      \
    X(SWIN_SPLIT_LN) X(SWIN_SPLIT_K) X(GEMM_LN_V4) X(SWIN_FUSED_MLP) X(SWIN_MLP512) X(SWIN_PROJ512) X(SWIN_QKV512) X(SWIN_MLP512_GRID) X(SWIN_FUSED_PROJ) X(SWIN_MLP_ABL) X(SWIN_MLP_SEQ) X(SWIN_MLP_NW4) X(SWIN_FUSED_MERGE) X(SWIN_ROW_MAX) X(LN_LIGHT) X(WATTN_STREAM)
enum VscOpt {
#define X(n) OPT_##n,
    VSC_OPT_LIST(X)
#undef X
    OPT_COUNT
};
const char *vsc_opt(VscOpt o);   // value of VSC_<name> (environment at first use, or the last vsc_set_option), nullptr when unset

// ---- bf16 <-> f32 ------------------------------------------------------------
__device__ __host__ inline float bf16_to_f32(uint16_t h) {
    union { uint32_t u; float f; } v;
    v.u = ((uint32_t)h) << 16;
    return v.f;
}

// round-to-nearest-even, NaN kept quiet
__device__ __host__ inline uint16_t f32_to_bf16(float f) {
    union { uint32_t u; float f; } v;
    v.f = f;
    if ((v.u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((v.u >> 16) | 0x40);
    uint32_t lsb = (v.u >> 16) & 1u;
    v.u += 0x7fffu + lsb;
    return (uint16_t)(v.u >> 16);
}

// device-side packing: one v_cvt_pk_bf16_f32 (RNE) per pair
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
    f32x2_t v = {lo, hi};
    hw_bf16x2_t b = __builtin_convertvector(v, hw_bf16x2_t);
    return *(uint32_t *)&b;
}

// ---- the encoders' 16-bit operand type ("lp") ------------------------------------------------------------------------------
// Every MFMA operand of the frame -> descriptor path (weights, LayerNorm outputs, qkv, probabilities, attention output, MLP hidden)
// is ONE 16-bit type per build of the library: bf16 (libvsc_hip.so: the configuration BASELINE.json names) or, with
// -DVSC_OPERAND_F16, IEEE fp16 (libvsc_hip_f16.so).  Same MFMA rate (v_mfma_f32_16x16x32_{bf16,f16}), same bytes, fp32 accumulation
// either way; fp16 carries 11 significand bits against 8, which is what the end-to-end uAP parity needs (the weights'
// bf16 rounding alone moves ViT-B/16 descriptors by 1.3e-4 on average, fp16 by 1.6e-5).  Range: the residual stream, LayerNorm,
// softmax and pooling stay fp32; what is rounded is bounded by LayerNorm gains / GELU / V rows -- the regime these networks were
// trained in (the reference runs its CLIP tower under fp16 autocast: extract_query_feats.py:159).  Values past 65504 saturate at
// +-65504 (lp_kernel_entry below; under autocast they would become inf).  The similarity search (knn.hip) and the matching-track convolutions (conv.hip) are bf16 by CONSTRUCTION (their
// error bounds are derived for it): they define VSC_TU_BF16 and are the same objects in both libraries.
#if defined(VSC_OPERAND_F16) && !defined(VSC_TU_BF16)
#define VSC_LP_F16 1
#define VSC_LP_NAME "fp16"
#define VSC_LP_ASM "f16"      // mnemonic suffix in hand-written / generated asm: v_mfma_f32_16x16x32_<>, v_cvt_pk_<>_f32
typedef __attribute__((ext_vector_type(2))) _Float16 hw_f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 hw_f16x8_t;
__device__ inline uint32_t lp_pack2(float lo, float hi) {          // one v_cvt_pk_f16_f32 (RNE)
    f32x2_t v = {lo, hi};
    hw_f16x2_t b = __builtin_convertvector(v, hw_f16x2_t);
#ifdef VSC_LP_ACT_MANT
    // EXPERIMENT (tools/micro/operand_speed_ab.py act): ACTIVATIONS keep only VSC_LP_ACT_MANT explicit significand bits inside their fp16
    // containers (weights keep all ten: they are rounded by f32_to_lp) -- does the multiplier array's power follow the operands' bit width?
    constexpr uint32_t DROP = 10 - VSC_LP_ACT_MANT;
    return (*(uint32_t *)&b + (0x00010001u << (DROP - 1))) & ~(((1u << DROP) - 1u) * 0x00010001u);
#else
    return *(uint32_t *)&b;
#endif
}
__device__ __host__ inline float lp_to_f32(uint16_t h) {
    union { uint16_t u; _Float16 f; } v;
    v.u = h;
    return (float)v.f;
}
__device__ __host__ inline uint16_t f32_to_lp(float f) {           // RNE, overflow -> inf
    union { uint16_t u; _Float16 f; } v;
    v.f = (_Float16)f;
    return v.u;
}
__device__ __forceinline__ f32x4_t lp_mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(*(hw_f16x8_t *)&a, *(hw_f16x8_t *)&b, c, 0, 0, 0);
}
__device__ __forceinline__ float lp_dot2(uint32_t a, uint32_t b, float c) {   // c + a.lo * b.lo + a.hi * b.hi, exact products
    return __builtin_amdgcn_fdot2(*(hw_f16x2_t *)&a, *(hw_f16x2_t *)&b, c, false);
}
#define LP_ONE_BITS 0x3C00
// First statement of every kernel that rounds to the operand type: MODE.FP16_OVFL = 1 -- a conversion that overflows fp16 then yields
// +-65504 instead of inf (true infinities stay; measured: tools/micro/fp16_ovfl_probe.hip).  A checkpoint whose activations leave
// fp16's range at some rounding point thereby costs accuracy in that element instead of a NaN descriptor.
__device__ __forceinline__ void lp_kernel_entry() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }
#else
#define VSC_LP_F16 0
#define VSC_LP_NAME "bf16"
#define VSC_LP_ASM "bf16"
__device__ inline uint32_t lp_pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
__device__ __host__ inline float lp_to_f32(uint16_t h) { return bf16_to_f32(h); }
__device__ __host__ inline uint16_t f32_to_lp(float f) { return f32_to_bf16(f); }
__device__ __forceinline__ f32x4_t lp_mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float lp_dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(*(hw_bf16x2_t *)&a, *(hw_bf16x2_t *)&b, c, false);
}
#define LP_ONE_BITS 0x3F80
__device__ __forceinline__ void lp_kernel_entry() {}
#endif
#define LP_ONES ((bf16x8_t){LP_ONE_BITS, LP_ONE_BITS, LP_ONE_BITS, LP_ONE_BITS, LP_ONE_BITS, LP_ONE_BITS, LP_ONE_BITS, LP_ONE_BITS})

// ---- 16-byte stores through a buffer descriptor with a scalar offset ---------------------------------------------------
// buffer_store_dwordx4 with an SGPR soffset: on gfx950 the instruction is still reading its four data registers when the next
// instruction issues, and a VALU write to one of them in that slot reaches memory instead of the value stored (measured: the
// persistent GEMM's GELU write-out stored garbage in the first dword of a row segment whenever the next pass's v_med3 reused the
// register right behind the store -- ~1 % of the tiles, different ones every run; tools/micro/gemm_determinism.py).  The
// compiler's hazard recognizer pads this store-data hazard only for stores WITHOUT a register soffset.  Hence: the data stays
// live through one s_nop 1 (two wait states) behind every such store.  8-byte stores are not affected.
typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int vsc_u32x4_t;
__device__ __forceinline__ void buffer_store_b128_soff(vsc_u32x4_t data, __amdgpu_buffer_rsrc_t rsrc, uint32_t voffset, uint32_t soffset) {
    __builtin_amdgcn_raw_buffer_store_b128(data, rsrc, voffset, soffset, 0);
    asm volatile("s_nop 1" : : "v"(data));
}

// ---- wave reductions (64 lanes) --------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware block id remap (8 XCDs, block b runs on XCD b % 8): give every XCD a
// contiguous range of logical tiles so neighbouring tiles share its L2.
// Bijective for any grid size.
__device__ inline int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, slot = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// ---- launchers implemented in the .hip files ----------------------------------------------------
int launch_gemm_bf16(const uint16_t *a, const uint16_t *w, const float *bias, const float *aux,
                     void *out, int64_t m, int n, int k, int epilogue, int tokens,
                     hipStream_t stream);
// pair_ws: VSC_GEMM_LN_WS_BYTES of device memory private to `stream` (nullptr: a per-device buffer -- one call at a time then)
int launch_gemm_ln_bf16(const uint16_t *a, const uint16_t *w, const float *bias, const float *gamma,
                        const float *beta, const float *x_in, float *x_out, uint16_t *xb_out, int64_t m, int n,
                        int k, float eps, hipStream_t stream, void *pair_ws = nullptr, int merge_res = 0, int merge_c = 0);
bool gemm_ln_supported(int n, int k);
// Internal epilogue kinds of the encoder's LayerNorm folding (not part of the public enum in vsc_hip.h):
//   LNF_*            the A operand is bf16(x) itself and gamma is folded into W: out = act(rstd_m * (acc - mu_m * colsum_n)
//                    + bias_n) with (mu_m, rstd_m) = rowstats[m], colsum_n = sum_k W'[n,k], bias_n = b_n + sum_k beta_k W[n,k]
//   RESADD_STATS_F32 RESADD_F32 that also stores bf16(out) to xb and, per row and 64-column slice, (mean, centred sum of
//                    squares) to stats[slice][m]; ln_stats_merge turns the slices into rowstats
//   LN_RES_F32       Swin-V2's res-post-norm update on the persistent kernel: out = (aux ? aux : 0) + LayerNorm(acc + bias) * gamma + beta
//                    over the whole row (N = 256: one tile; N = 512: the two workgroups holding a row's two tiles exchange their
//                    (mean, M2) through L2), plus xb = bf16(out)
enum { VSC_EPI_LNF_BF16 = 6, VSC_EPI_LNF_GELU_BF16 = 7, VSC_EPI_LNF_QGELU_BF16 = 8, VSC_EPI_RESADD_STATS_F32 = 9, VSC_EPI_LN_RES_F32 = 10 };
struct GemmExtra {
    uint16_t *xb = nullptr;          // RESADD_STATS: bf16 copy of out [m, n]
    float *stats = nullptr;          // RESADD_STATS: [n / 64][m][2]
    const float *rowstats = nullptr; // LNF: [m][2] = (mean, rstd)
    const float *slices = nullptr;   // LNF: the [nslices][m][2] (mean, M2) partials RESADD_STATS wrote; given (with eps), the persistent kernel merges
    int nslices = 0;                 //      them per tile itself, other kernels get them merged into rowstats (then a scratch buffer) first
    const float *colsum = nullptr;   // LNF: [n]
    const float *gamma = nullptr, *beta = nullptr;   // LN_RES: [n]
    float eps = 0.f;                 // LN_RES
    float *xch = nullptr;            // LN_RES, N = 512: [2][256 workgroups][256 rows][2] partial row statistics
    int *xflags = nullptr;           // LN_RES, N = 512: [256 workgroups][2] tiles published, counting up across launches from `epoch`
    int epoch = 0;
};
// bytes of the pair-exchange workspace of launch_gemm_ln_bf16 (one per stream that may run it).
// Contract: the workspace's flag words count up from launch to launch; the count ("epoch") is kept by the launcher in a host map keyed
// by the workspace pointer and passed to the kernel as an argument.  Hence (1) one workspace serves ONE stream at a time; (2) whoever
// frees a workspace calls gemm_ln_workspace_forget(ws) first -- a later allocation at the same address would otherwise inherit a stale
// epoch over whatever the new memory holds; (3) launches that use it cannot be captured into a HIP graph (the launcher refuses).
constexpr size_t VSC_GEMM_LN_WS_BYTES = 2 * 256 * 256 * 2 * 4 + 256 * 2 * 4;
void gemm_ln_workspace_forget(const void *ws);   // call before freeing a workspace that was passed to launch_gemm_ln_bf16
int launch_gemm_bf16_ex(const uint16_t *a, const uint16_t *w, const float *bias, const float *aux, void *out, int64_t m,
                        int n, int k, int epilogue, int tokens, const GemmExtra &ex, hipStream_t stream);
int launch_ln_stats_merge(const float *stats, float *rowstats, int64_t rows, int slices, int width, float eps,
                          hipStream_t stream);
int launch_attention_bf16(const uint16_t *qkv, uint16_t *out, int frames, int tokens, int heads,
                          hipStream_t stream);
int launch_layernorm(const float *x, const float *g, const float *b, void *out, int64_t rows,
                     int width, float eps, int out_f32, hipStream_t stream);
int launch_patchify(const float *frames, uint16_t *patches, int64_t n, int channels, int image,
                    int patch, int kpad, hipStream_t stream);
int launch_cls_rows(float *x, const float *cls, const float *pos, int64_t frames, int tokens,
                    int width, hipStream_t stream);
int launch_ln_pool(const float *x, const float *g, const float *b, float *pooled, float *tokens_out,
                   int64_t frames, int tokens, int width, float eps, int pool, float gem_p,
                   hipStream_t stream);
int launch_gem_pool_bf16(const uint16_t *x, float *pooled, int64_t frames, int tokens, int channels, float gem_p,
                         hipStream_t stream);
int launch_head(const float *pooled, const float *w, const float *bias, float *desc, int64_t frames,
                int width, int out_dim, int l2, hipStream_t stream);
int launch_window_attention(const uint16_t *qkv, uint16_t *out, const float *bias, const float *scale,
                            int frames, int res, int ws, int shift, int heads, hipStream_t stream);
int launch_ln_residual(const float *t, const float *gamma, const float *beta, const float *x_in, float *x_out,
                       uint16_t *xb, int64_t rows, int width, float eps, hipStream_t stream);
// fused Swin MLP (swin_mlp.hip): x += LN(GELU(xb W1^T + b1) W2^T + b2) gamma + beta, xb = bf16(x), widths 128 / 256;
// w2p = fc2.weight with its hidden axis reordered by swin_mlp_permute_hidden (host, at model load)
bool swin_mlp_supported(int c);
bool swin_proj_mlp_supported(int c);
int launch_swin_proj_mlp(const uint16_t *att, const uint16_t *wp, const float *bp, const float *gamma1, const float *beta1, const uint16_t *w1,
                         const float *b1, const uint16_t *w2p, const float *b2, const float *gamma2, const float *beta2, float *x, uint16_t *xb,
                         int64_t m, int c, float eps, hipStream_t stream);
void swin_mlp_permute_hidden(const float *src, float *dst, int c);
// C = 512 (swin_mlp512.hip): one wave per SIMD, 32-unit hidden chunks; w2c = fc2.weight chunk-major [64][512][32] (swin_mlp512_pack_w2)
void swin_mlp512_pack_w2(const float *src, float *dst);
int launch_swin_proj_mlp512(const uint16_t *att, const uint16_t *wp, const float *bp, const float *gamma1, const float *beta1, const uint16_t *w1,
                            const float *b1, const uint16_t *w2c, const float *b2, const float *gamma2, const float *beta2, float *x, uint16_t *xb,
                            int64_t m, float eps, hipStream_t stream);
int launch_swin_proj_mlp_qkv512(const uint16_t *att, const uint16_t *wp, const float *bp, const float *gamma1, const float *beta1, const uint16_t *w1,
                                const float *b1, const uint16_t *w2c, const float *b2, const float *gamma2, const float *beta2, const uint16_t *wq,
                                const float *bq, float *x, uint16_t *qkv_next, int64_t m, float eps, hipStream_t stream);
void swin_mlp512_set_timing_buffer(uint32_t *buf);   // diagnostic: [workgroups][4][8] cycle counters of the timing variant (VSC_SWIN_MLP_ABL=5)
int launch_swin_mlp512(const uint16_t *w1, const float *b1, const uint16_t *w2c, const float *b2, const float *gamma, const float *beta,
                       float *x, uint16_t *xb, int64_t m, float eps, hipStream_t stream);
int launch_swin_mlp(const uint16_t *w1, const float *b1, const uint16_t *w2p, const float *b2, const float *gamma, const float *beta,
                    float *x, uint16_t *xb, int64_t m, int c, float eps, hipStream_t stream);
int launch_merge_gather(const uint16_t *xb, uint16_t *out, int64_t frames, int res, int c, hipStream_t stream);
int launch_l2_normalize(float *x, int64_t n, int d, hipStream_t stream);
int launch_patchify_u8(const uint8_t *frames, uint16_t *patches, int64_t n, int channels, int image, int patch, int kpad,
                       const float *mean, const float *std, hipStream_t stream);
int launch_f32_to_bf16(const float *src, uint16_t *dst, int64_t rows, int cols, int cols_pad,
                       hipStream_t stream);
