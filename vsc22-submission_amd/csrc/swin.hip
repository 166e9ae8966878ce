// Swin-Transformer-V2 kernels (reference model: train/train_v115/torch2scripts.py:70-366).
//   window_attention_kernel  windowed cosine multi-head attention, head_dim 32
//   ln_residual_kernel       x (+)= LayerNorm(t) (res-post-norm) + bf16 shadow of x for the next GEMM
//   merge_gather_kernel      PatchMerging's 2x2 gather (x0|x1|x2|x3) on the bf16 shadow
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr int HD = 32;  // head dim of every Swin-V2 stage (C / heads)

// sum of the squares of 8 bf16 values: four v_dot2c_f32_bf16 (exact products, fp32 accumulation) instead of 8 conversions'
// worth of multiplies and adds -- the kernel is bound by its vector instructions, and the matrix pipe does not hide them
// The P . V product runs on bf16 operands in BOTH builds of the library (fp16 operands everywhere else in libvsc_hip_f16.so): a
// probability of the bounded softmax, exp(logit - the head's upper bound), may be as small as e^-69 -- bf16 holds it (fp32's exponent
// range), fp16 flushes it to zero, and a row whose logits all lie far below the bound would sum to 0.  Keeping P in bf16 keeps the
// one-pass bounded softmax (the streamed kernels otherwise need a first pass over the keys for the row maximum: Swin-V2-B 17.3 ->
// 16.0 k frames/s); the MFMA needs both operands in one type, so V is rounded to bf16 once while it is staged.  What that costs in
// accuracy: P's bf16 rounding moves a ViT-B/16 descriptor by 2e-6 on average, V's by < 1e-5 (tools/precision_budget.py: "p", "qkv").
__device__ __forceinline__ uint32_t v_pair(uint16_t lo, uint16_t hi) {      // two V values of the build's operand type -> a bf16 pair
#if VSC_LP_F16
    return pack_bf16x2(lp_to_f32(lo), lp_to_f32(hi));
#else
    return (uint32_t)lo | ((uint32_t)hi << 16);
#endif
}
#define pv_pack2 pack_bf16x2
#define pv_mfma16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define PV_ONES ((bf16x8_t){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80})

__device__ __forceinline__ float sumsq8(bf16x8_t raw) {
    union { bf16x8_t v; uint32_t p[4]; } u;
    u.v = raw;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) ss = lp_dot2(u.p[j], u.p[j], ss);
    return ss;
}

// ------------------------------------------------------------------------------------------
// One workgroup per (frame, window, head).  The cyclic shift, the window partition and their
// inverses (torch.roll + window_partition / window_reverse, torch2scripts.py:37-67,275-296) are
// pure index math here: window token i sits at shifted coords (sy, sx) and is read from / written
// to image position ((sy + shift) % res, (sx + shift) % res); nothing is materialised.
//   * K rows are L2-normalised in fp32 while staged (F.normalize(k), :159) and stored bf16 with
//     the 64-byte-row swizzle; V is staged transposed; Q is normalised in registers (the four
//     lanes holding a query's 32 dims reduce with two xor-shuffles).
//   * scores use the swapped MFMA (A = K-hat, B = Q-hat) so a lane owns one query: logits =
//     cos * exp(min(logit_scale, ln 100)) + 16*sigmoid(cpb) bias (+ -100 where the shift mask
//     separates the two tokens' regions, :236-254), whole row in registers, exp2 softmax.
//   * the position bias is NOT the expanded [N, N] matrix of the reference (256 KiB per head, 16
//     dependent global loads per query tile): bias[i][j] = table[(yi - yj + w-1) * (2w-1) + (xi - xj + w-1)],
//     so the head's compact (2w-1)^2 table (3.8 KiB) sits in LDS and is gathered with ds_read_b32.
//   * P (bf16) is directly the B operand of the PV MFMA (k-slot permutation as attention.hip).
//   * where the time goes (ablations of the -DVSC_ATTN_ABLATION build, tools/micro/wattn_bench.py, stage-3 launch of 256 frames,
//     warm): whole kernel 115 us; without MFMAs and exp2 86; without the K / V loads 95; without the stores 100; without loads
//     and stores 87; with only the Q loads, the LDS traffic and the remaining VALU work (bias-table gathers, masks, K-hat
//     normalisation, index math) 63 -- unlike the ViT kernel (attention.hip) this one IS bound by its VALU / LDS work; requesting
//     V together with K (one memory round trip instead of two) changes nothing with four resident workgroups (116 us), and a
//     start skew of the residents loses (4 generations per launch only).
template <int NT, int ABL = 0>  // 16-key tiles per window: 4 (8x8 window) or 16 (16x16 window); ABL: ablation bits of the diagnostic build
__global__ __launch_bounds__(NT * 32, 4) void window_attention_kernel(
    const uint16_t *__restrict__ qkv, uint16_t *__restrict__ out, const float *__restrict__ bias,
    const float *__restrict__ scale, int res, int ws, int shift, int heads) {
    lp_kernel_entry();
    constexpr int N = NT * 16, NWAVES = NT / 2, NTHREADS = NWAVES * 64;
    // V^T rows: 16-byte aligned, +32 B of padding (conflict-free ds_read_b128 over a 16-lane group's rows); inside a 32-key block
    // the 4-key groups are stored in the order a lane consumes them (group g of the even 16-key tile, then group g of the odd
    // one), so a lane's 8 keys of a PV step are ONE ds_read_b128 -- as two 8-byte pieces 32 B apart the compiler fused them
    // into ds_read2_b64, which moves 128 B/clk against 256
    constexpr int VSTRIDE = N * 2 + 32;
    constexpr int WS = NT == 16 ? 16 : 8, SIDE = 2 * WS - 1;
    // bias table in LDS: columns mirrored (a lane's 4 consecutive keys then read 4 ASCENDING floats), rows padded to TSTRIDE
    // floats, four copies shifted by 0..3 floats -- the copy whose start is 16-byte aligned for this lane's query column gives
    // the 4 bias values of 4 scores in one ds_read_b128 (four ds_read_b32 gathers per 4 scores were 30 % of the kernel's LDS
    // cycles; 8-byte pairs get fused into the half-rate ds_read2_b64)
    constexpr int TSTRIDE = WS == 16 ? 36 : 20, TCOPY = SIDE * TSTRIDE;
    // + 9 rows [N] of mask addends (0 or -100 log2 e), one per region a query can lie in: a masked tile costs one ds_read_b128
    // and two v_pk_add_f32 per 4 scores (compare + select + add per SCORE before: the masked body issued 75 % more VALU)
    __shared__ __attribute__((aligned(16))) char smem[N * 64 + HD * VSTRIDE + N * 4 + N + 4 * TCOPY * 4 + 9 * N * 4];
    char *klds = smem;                                   // [N][32] bf16, 64-B rows, chunk ^= (-(row>>2)) & 3
    char *vt = smem + N * 64;                            // [32][VSTRIDE]
    int *rowmap = (int *)(smem + N * 64 + HD * VSTRIDE);  // image token index of window token i
    unsigned char *region = (unsigned char *)(rowmap + N);
    float *tbl = (float *)(smem + N * 64 + HD * VSTRIDE + N * 4 + N);  // this head's bias table, * log2(e)
    float *maskrow = tbl + 4 * TCOPY;                                    // [9][N]

    const float LOG2E = 1.44269504088896340736f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwx = res / ws, nw = nwx * nwx;
    // head_dim 32 = 64 B per token row: two heads share every 128-B line of qkv and out.  Consecutive block ids run on
    // different XCDs, so with b = blockIdx.x each line was fetched into two L2s (PMC: 2.07x the algorithmic reads);
    // the remap gives every XCD a contiguous range, i.e. all heads of a window next to each other in one L2.
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int head = b % heads; b /= heads;
    const int win = b % nw;
    const int frame = b / nw;
    const int wh = win / nwx, wwx = win - wh * nwx;
    const int C = heads * HD;
    const int64_t ld = 3 * (int64_t)C;
    const uint16_t *base = qkv + (int64_t)frame * res * res * ld + head * HD;

    // (ws == WS: the launcher picks the instantiation by the window size, so the window decomposition is shifts)
    for (int i = tid; i < N; i += NTHREADS) {
        const int wy = i / WS, wx = i - wy * WS;
        const int sy = wh * ws + wy, sx = wwx * ws + wx;
        int y = sy + shift, x = sx + shift;
        y = y >= res ? y - res : y;
        x = x >= res ? x - res : x;
        rowmap[i] = y * res + x;
        const int hr = sy < res - ws ? 0 : (sy < res - shift ? 1 : 2);
        const int wr = sx < res - ws ? 0 : (sx < res - shift ? 1 : 2);
        region[i] = (unsigned char)(3 * hr + wr);
    }
    for (int i = tid; i < SIDE * SIDE; i += NTHREADS) {
        const float v = bias[(int64_t)head * SIDE * SIDE + i] * 1.44269504088896340736f;
        const int at = (i / SIDE) * TSTRIDE + (SIDE - 1 - i % SIDE);
#pragma unroll
        for (int c = 0; c < 4; ++c) tbl[c * TCOPY + at + c] = v;
    }
    __syncthreads();

    const int fr = lane & 15, g = lane >> 4;
    // ---- Q fragments (normalised) of this wave's two query tiles, issued before the staging
    bf16x8_t qf[2];
    int qrow[2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const int q = (wave * 2 + qi) * 16 + fr;
        qrow[qi] = rowmap[q];
        const bf16x8_t raw = *(const bf16x8_t *)(base + qrow[qi] * ld + g * 8);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = lp_to_f32((uint16_t)raw[j]);
        float ss = sumsq8(raw);
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float inv = __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f));   // 1 / max(|x|, 1e-12): F.normalize's eps
        union { uint32_t w[4]; bf16x8_t v; } pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) pk.w[j] = lp_pack2(v[2 * j] * inv, v[2 * j + 1] * inv);
        qf[qi] = pk.v;
    }
    // ---- stage K-hat: 4 threads per key row (16 B each), norm over the row by two shuffles
    constexpr int KIT = (N * 4) / NTHREADS;  // = 2: both rows of a thread are requested before either is normalised
    static_assert(KIT * NTHREADS == N * 4, "K staging covers the window exactly");
    bf16x8_t kraw[KIT];
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
        const int e = tid + it * NTHREADS;
        kraw[it] = (ABL & 8) ? (bf16x8_t){1, 2, 3, 4, 5, 6, 7, 8} : *(const bf16x8_t *)(base + C + rowmap[e >> 2] * ld + (e & 3) * 8);
    }
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
        const int e = tid + it * NTHREADS;
        const int i = e >> 2, c = e & 3;
        const bf16x8_t raw = kraw[it];
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = lp_to_f32((uint16_t)raw[j]);
        float ss = sumsq8(raw);
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        const float inv = __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f));   // 1 / max(|x|, 1e-12): F.normalize's eps
        uint4 pk;
        pk.x = lp_pack2(v[0] * inv, v[1] * inv);
        pk.y = lp_pack2(v[2] * inv, v[3] * inv);
        pk.z = lp_pack2(v[4] * inv, v[5] * inv);
        pk.w = lp_pack2(v[6] * inv, v[7] * inv);
        *(uint4 *)(klds + i * 64 + ((c ^ ((-(i >> 2)) & 3)) << 4)) = pk;
    }
    // ---- stage V transposed: task = (4 keys) x (8 dims); 16 consecutive lanes = 16 key groups
    for (int e = tid; e < (N / 4) * 4; e += NTHREADS) {
        const int blk = e >> 6, c8 = (e >> 4) & 3, kg = blk * 16 + (e & 15);
        const int slot = (kg & ~7) | ((kg & 3) << 1) | ((kg >> 2) & 1);   // consumption order inside the 32-key block
        bf16x8_t r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = (ABL & 8) ? (bf16x8_t){1, 2, 3, 4, 5, 6, 7, 8} : *(const bf16x8_t *)(base + 2 * C + rowmap[kg * 4 + i] * ld + c8 * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint2 pk;
            pk.x = v_pair((uint16_t)r[0][j], (uint16_t)r[1][j]);
            pk.y = v_pair((uint16_t)r[2][j], (uint16_t)r[3][j]);
            // head dim d sits in row 16 ((d >> 2) & 1) + 4 (d >> 3) + (d & 3): the PV accumulators of a lane are then 8
            // CONSECUTIVE head dims of its query (8 g + 4 ct + r) -- one 16-byte store per lane, 64 contiguous bytes per query
            const int d = c8 * 8 + j;
            *(uint2 *)(vt + (16 * ((d >> 2) & 1) + 4 * (d >> 3) + (d & 3)) * VSTRIDE + slot * 8) = pk;
        }
    }
    __syncthreads();

    // scale[head] < 0 (see vsc_window_attention_bf16): |scale| is the logit scale and the caller has subtracted the head's upper
    // bound |scale| + max(bias) from its bias table -- cosine logits are bounded, so every shifted logit is <= 0 and, the caller
    // guarantees, >= -100 log2 units: the softmax needs no row maximum (a v_max3 per pair of scores, two shuffles) and no
    // subtraction (a v_pk_add per pair): a sixth of this kernel's vector instructions.  The same quotient either way.
    const float scraw = scale[head];
    const bool nomax = scraw < 0.f;   // workgroup-uniform
    const float sc = fabsf(scraw) * LOG2E;

    // Only windows in the last window row / column of a shifted layer hold more than one mask region
    // (torch2scripts.py:236-254): every other workgroup runs the body without the mask compares/selects
    // (3 of the ~15 VALU instructions per score; the kernel is VALU-bound).
    const bool need_mask = shift > 0 && (wh == nwx - 1 || wwx == nwx - 1);
    if (need_mask) {   // workgroup-uniform
        for (int i = tid; i < 9 * N; i += NTHREADS) maskrow[i] = region[i % N] != i / N ? -100.0f * LOG2E : 0.f;
        __syncthreads();
    }
    auto rows = [&](auto masked_c, auto nomax_c) {
        constexpr bool MASKED = decltype(masked_c)::value, NOMAX = decltype(nomax_c)::value;
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            const int q = (wave * 2 + qi) * 16 + fr;
            const int rq = region[q];
            // bias of (query q, key j) = table[yq - yj + WS-1][xq - xj + WS-1]; this lane's keys of a tile are xj0 .. xj0 + 3 of one
            // window row: mirrored columns WS-1 - xq + xj0 .. + 3, which start at a multiple of 4 in copy (xq + 1) % 4
            const int tcopy = (q + 1) & 3;
            const float *lt = tbl + tcopy * (TCOPY + 1) + (q / WS + WS - 1) * TSTRIDE + WS - 1 - (q % WS);
            f32x2_t s[NT][2];
            float mx = -INFINITY;
            const f32x2_t sc2 = (f32x2_t){sc, sc};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int krow = t * 16 + fr;
                const bf16x8_t kf = *(const bf16x8_t *)(klds + krow * 64 + ((g ^ ((-(krow >> 2)) & 3)) << 4));
                f32x4_t z = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                if (ABL & 2) z[0] = (float)kf[0] + (float)qf[qi][1];
                else z = lp_mfma16(kf, qf[qi], z);
                float tb[4];
                {
                    const int j0 = t * 16 + g * 4;
                    const f32x4_t b4 = *(const f32x4_t *)(lt - (j0 / WS) * TSTRIDE + (j0 % WS));
                    tb[0] = b4[0], tb[1] = b4[1], tb[2] = b4[2], tb[3] = b4[3];
                }
                if (MASKED) {
                    const f32x4_t m4 = *(const f32x4_t *)(maskrow + rq * N + t * 16 + g * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) tb[r] += m4[r];
                }
                s[t][0] = (f32x2_t){z[0], z[1]} * sc2 + (f32x2_t){tb[0], tb[1]};  // v_pk_fma_f32
                s[t][1] = (f32x2_t){z[2], z[3]} * sc2 + (f32x2_t){tb[2], tb[3]};
                if (!NOMAX) {
                    mx = fmaxf(fmaxf(mx, s[t][0][0]), s[t][0][1]);                // v_max3_f32
                    mx = fmaxf(fmaxf(mx, s[t][1][0]), s[t][1][1]);
                }
                if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            if (!NOMAX) {
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            }
            const f32x2_t nmx = (f32x2_t){-mx, -mx};
            bf16x8_t pb[NT / 2];
#pragma unroll
            for (int u = 0; u < NT / 2; ++u) {
                f32x2_t e[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const f32x2_t d = NOMAX ? s[2 * u + (h >> 1)][h & 1] : s[2 * u + (h >> 1)][h & 1] + nmx;  // v_pk_add_f32
                    e[h] = (ABL & 1) ? d : (f32x2_t){__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
                }
                union { uint32_t w[4]; bf16x8_t v; } pk;
#pragma unroll
                for (int h = 0; h < 4; ++h) pk.w[h] = pv_pack2(e[h][0], e[h][1]);
                pb[u] = pk.v;
            }
            // O^T[dh][query] += V^T[dh][key] . P^T[key][query]; a third A operand of ones gives the row sums of the bf16 P the
            // products use (every row of that tile = sum over the keys: no VALU adds, no cross-lane reduction; the matrix pipe
            // is 15 % busy in this kernel, the vector pipe 80 %)
            const bf16x8_t ones = PV_ONES;
            f32x4_t o[2], osum = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            o[0] = o[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NT / 2; ++u) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const bf16x8_t vf = *(const bf16x8_t *)(vt + (ct * 16 + fr) * VSTRIDE + (32 * u + 8 * g) * 2);
                    if (ABL & 2) o[ct][u & 3] += (float)vf[0] + (float)pb[u][ct];
                    else o[ct] = pv_mfma16(vf, pb[u], o[ct]);
                }
                if (ABL & 2) osum[0] += (float)pb[u][0];
                else osum = pv_mfma16(ones, pb[u], osum);
                if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            const float inv = __builtin_amdgcn_rcpf(osum[0]);
            uint16_t *orow = out + ((int64_t)frame * res * res + qrow[qi]) * C + head * HD + g * 8;
            uint4 pk;
            pk.x = lp_pack2(o[0][0] * inv, o[0][1] * inv);
            pk.y = lp_pack2(o[0][2] * inv, o[0][3] * inv);
            pk.z = lp_pack2(o[1][0] * inv, o[1][1] * inv);
            pk.w = lp_pack2(o[1][2] * inv, o[1][3] * inv);
            if (!((ABL & 16) && inv != 12345.f)) *(uint4 *)orow = pk;
        }
    };
    if (need_mask) {
        if (nomax) rows(std::true_type{}, std::true_type{});
        else rows(std::true_type{}, std::false_type{});
    } else {
        if (nomax) rows(std::false_type{}, std::true_type{});
        else rows(std::false_type{}, std::false_type{});
    }
}

// ------------------------------------------------------------------------------------------
// Unshifted 16 x 16 windows (every block of the 512- and 1024-wide stages, every other block of the first two): the same
// attention with the softmax STREAMED over pairs of key tiles instead of held as a 256-key row per lane.  The kernel above is
// bound by its residency -- 128 VGPRs and 62 KiB of LDS allow two workgroups per CU, each of which loads, computes and stores in
// turn -- not by any pipe (PMC: matrix pipe 17 % busy; its VALU / LDS work adds up to half the launch).  Without shift there are
// no masks (9 KiB of mask rows gone: 52 KiB), and with the bounded softmax (scale < 0: every logit <= 0, no row maximum) a
// probability can be packed and fed to the PV MFMA as soon as its score exists: no score row, ~70 VGPRs, THREE workgroups per
// CU.  Heads whose logit span is too large for the bound (scale >= 0) take a first pass over the key tiles for the row maximum
// (scores recomputed: MFMAs are not what this kernel lacks).  Layouts, operand order and the ones-operand row sum as above.
__global__ __launch_bounds__(512, 6) void window_attention_stream_kernel(const uint16_t *__restrict__ qkv, uint16_t *__restrict__ out,
                                                                         const float *__restrict__ bias, const float *__restrict__ scale,
                                                                         int res, int heads) {
    lp_kernel_entry();
    constexpr int NT = 16, N = 256, NTHREADS = 512, WS = 16, SIDE = 31;
    constexpr int VSTRIDE = N * 2 + 32, TSTRIDE = 36, TCOPY = SIDE * TSTRIDE;
    __shared__ __attribute__((aligned(16))) char smem[N * 64 + HD * VSTRIDE + N * 4 + 4 * TCOPY * 4];
    char *klds = smem;
    char *vt = smem + N * 64;
    int *rowmap = (int *)(smem + N * 64 + HD * VSTRIDE);
    float *tbl = (float *)(smem + N * 64 + HD * VSTRIDE + N * 4);
    const float LOG2E = 1.44269504088896340736f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwx = res / WS, nw = nwx * nwx;
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int head = b % heads; b /= heads;
    const int win = b % nw, frame = b / nw;
    const int wh = win / nwx, wwx = win - wh * nwx;
    const int C = heads * HD;
    const int64_t ld = 3 * (int64_t)C;
    const uint16_t *base = qkv + (int64_t)frame * res * res * ld + head * HD;
    for (int i = tid; i < N; i += NTHREADS) rowmap[i] = (wh * WS + (i >> 4)) * res + wwx * WS + (i & 15);
    for (int i = tid; i < SIDE * SIDE; i += NTHREADS) {
        const float v = bias[(int64_t)head * SIDE * SIDE + i] * LOG2E;
        const int at = (i / SIDE) * TSTRIDE + (SIDE - 1 - i % SIDE);
#pragma unroll
        for (int c = 0; c < 4; ++c) tbl[c * TCOPY + at + c] = v;
    }
    __syncthreads();
    const int fr = lane & 15, g = lane >> 4;
    bf16x8_t qf[2];
    int qrow[2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const int q = (wave * 2 + qi) * 16 + fr;
        qrow[qi] = rowmap[q];
        const bf16x8_t raw = *(const bf16x8_t *)(base + qrow[qi] * ld + g * 8);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = lp_to_f32((uint16_t)raw[j]);
        float ss = sumsq8(raw);
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float inv = __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f));
        union { uint32_t w[4]; bf16x8_t v; } pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) pk.w[j] = lp_pack2(v[2 * j] * inv, v[2 * j + 1] * inv);
        qf[qi] = pk.v;
    }
    {   // K-hat: 4 threads per key row, two rows per thread
        bf16x8_t kraw[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int e = tid + it * NTHREADS;
            kraw[it] = *(const bf16x8_t *)(base + C + rowmap[e >> 2] * ld + (e & 3) * 8);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int e = tid + it * NTHREADS;
            const int i = e >> 2, c = e & 3;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = lp_to_f32((uint16_t)kraw[it][j]);
            float ss = sumsq8(kraw[it]);
            ss += __shfl_xor(ss, 1, 64);
            ss += __shfl_xor(ss, 2, 64);
            const float inv = __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f));
            uint4 pk;
            pk.x = lp_pack2(v[0] * inv, v[1] * inv);
            pk.y = lp_pack2(v[2] * inv, v[3] * inv);
            pk.z = lp_pack2(v[4] * inv, v[5] * inv);
            pk.w = lp_pack2(v[6] * inv, v[7] * inv);
            *(uint4 *)(klds + i * 64 + ((c ^ ((-(i >> 2)) & 3)) << 4)) = pk;
        }
    }
    if (tid < (N / 4) * 4) {   // V^T: task = (4 keys) x (8 dims), consumption order inside each 32-key block (see the kernel above)
        const int e = tid;
        const int blk = e >> 6, c8 = (e >> 4) & 3, kg = blk * 16 + (e & 15);
        const int slot = (kg & ~7) | ((kg & 3) << 1) | ((kg >> 2) & 1);
        bf16x8_t r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *(const bf16x8_t *)(base + 2 * C + rowmap[kg * 4 + i] * ld + c8 * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint2 pk;
            pk.x = v_pair((uint16_t)r[0][j], (uint16_t)r[1][j]);
            pk.y = v_pair((uint16_t)r[2][j], (uint16_t)r[3][j]);
            const int d = c8 * 8 + j;
            *(uint2 *)(vt + (16 * ((d >> 2) & 1) + 4 * (d >> 3) + (d & 3)) * VSTRIDE + slot * 8) = pk;
        }
    }
    __syncthreads();
    const float scraw = scale[head];
    const bool nomax = scraw < 0.f;   // workgroup-uniform
    const float sc = fabsf(scraw) * LOG2E;
    const f32x2_t sc2 = (f32x2_t){sc, sc};
    const bf16x8_t ones = PV_ONES;
    auto rows = [&](auto nomax_c) {
        constexpr bool NOMAX = decltype(nomax_c)::value;
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            const int q = (wave * 2 + qi) * 16 + fr;
            const int tcopy = (q + 1) & 3;
            const float *lt = tbl + tcopy * (TCOPY + 1) + (q / WS + WS - 1) * TSTRIDE + WS - 1 - (q % WS);
            auto scores = [&](int t, f32x2_t (&s)[2]) {   // the 4 logits (log2 units) of this lane for key tile t
                const int krow = t * 16 + fr;
                const bf16x8_t kf = *(const bf16x8_t *)(klds + krow * 64 + ((g ^ ((-(krow >> 2)) & 3)) << 4));
                f32x4_t z = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                z = lp_mfma16(kf, qf[qi], z);
                const int j0 = t * 16 + g * 4;
                const f32x4_t b4 = *(const f32x4_t *)(lt - (j0 / WS) * TSTRIDE + (j0 % WS));
                s[0] = (f32x2_t){z[0], z[1]} * sc2 + (f32x2_t){b4[0], b4[1]};
                s[1] = (f32x2_t){z[2], z[3]} * sc2 + (f32x2_t){b4[2], b4[3]};
            };
            float nmx = 0.f;
            if (!NOMAX) {   // unbounded head: the row maximum first
                float mx = -INFINITY;
#pragma unroll 4
                for (int t = 0; t < NT; ++t) {
                    f32x2_t s[2];
                    scores(t, s);
                    mx = fmaxf(fmaxf(mx, s[0][0]), s[0][1]);
                    mx = fmaxf(fmaxf(mx, s[1][0]), s[1][1]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                nmx = -mx;
            }
            const f32x2_t nm2 = (f32x2_t){nmx, nmx};
            f32x4_t o[2], osum = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            o[0] = o[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NT / 2; ++u) {
                union { uint32_t w[4]; bf16x8_t v; } pk;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x2_t s[2];
                    scores(2 * u + h, s);
#pragma unroll
                    for (int x = 0; x < 2; ++x) {
                        const f32x2_t d = NOMAX ? s[x] : s[x] + nm2;
                        pk.w[2 * h + x] = pv_pack2(__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1]));
                    }
                }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const bf16x8_t vf = *(const bf16x8_t *)(vt + (ct * 16 + fr) * VSTRIDE + (32 * u + 8 * g) * 2);
                    o[ct] = pv_mfma16(vf, pk.v, o[ct]);
                }
                osum = pv_mfma16(ones, pk.v, osum);
                if ((u & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
            const float inv = __builtin_amdgcn_rcpf(osum[0]);
            uint16_t *orow = out + ((int64_t)frame * res * res + qrow[qi]) * C + head * HD + g * 8;
            uint4 pk;
            pk.x = lp_pack2(o[0][0] * inv, o[0][1] * inv);
            pk.y = lp_pack2(o[0][2] * inv, o[0][3] * inv);
            pk.z = lp_pack2(o[1][0] * inv, o[1][1] * inv);
            pk.w = lp_pack2(o[1][2] * inv, o[1][3] * inv);
            *(uint4 *)orow = pk;
        }
    };
    if (nomax) rows(std::true_type{});
    else rows(std::false_type{});
}

// ------------------------------------------------------------------------------------------
// The same attention for the window sizes the reference's own models do not use but BASELINE.json's configs[4] names
// (Swin-V2-L at 384 x 384: window 24, clipped to 12 in the last stage): 576- / 144-token windows.  Written for coverage,
// not tuned like the 8 / 16 kernel above: the bias is gathered per score from the plain (2w-1)^2 table, shift masks are
// compares, the row maximum is always taken (scale < 0 -- see vsc_window_attention_bf16 -- only says the caller folded a
// bound into the table: subtracting the row maximum as well is the same quotient), V^T keeps the plain key order.
// MFMA operand layouts, k-slot permutation of P and the ones-operand row sum are those of the kernel above.
template <int WS>
__global__ __launch_bounds__(256, 1) void window_attention_wide_kernel(
    const uint16_t *__restrict__ qkv, uint16_t *__restrict__ out, const float *__restrict__ bias,
    const float *__restrict__ scale, int res, int shift, int heads) {
    lp_kernel_entry();
    constexpr int N = WS * WS, NT = N / 16, NTP = (NT + 1) & ~1, NP = NTP * 16;   // keys padded to whole 32-key PV steps
    constexpr int NTHREADS = 256, NWAVES = 4;   // one wave per SIMD: a 576-key score row is 144 registers of a lane (up to 512 are its own)
    constexpr int SIDE = 2 * WS - 1, VSTRIDE = NP * 2 + 32;
    static_assert(N % 16 == 0 && WS % 4 == 0, "whole 16-key tiles; a lane's 4 keys stay inside one window row");
    extern __shared__ __attribute__((aligned(16))) char wsmem[];
    char *klds = wsmem;                                       // [NP][32] bf16, 64-B rows, chunk ^= (-(row >> 2)) & 3
    char *vt = klds + NP * 64;                                // [32][VSTRIDE]
    int *rowmap = (int *)(vt + HD * VSTRIDE);                 // image token of window token i
    float *tbl = (float *)(rowmap + N);                       // this head's bias table * log2 e
    unsigned char *region = (unsigned char *)(tbl + SIDE * SIDE);
    const float LOG2E = 1.44269504088896340736f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwx = res / WS, nw = nwx * nwx;
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int head = b % heads; b /= heads;
    const int win = b % nw, frame = b / nw;
    const int wh = win / nwx, wwx = win - wh * nwx;
    const int C = heads * HD;
    const int64_t ld = 3 * (int64_t)C;
    const uint16_t *base = qkv + (int64_t)frame * res * res * ld + head * HD;
    for (int i = tid; i < N; i += NTHREADS) {
        const int wy = i / WS, wx = i - wy * WS;
        const int sy = wh * WS + wy, sx = wwx * WS + wx;
        int y = sy + shift, x = sx + shift;
        y = y >= res ? y - res : y;
        x = x >= res ? x - res : x;
        rowmap[i] = y * res + x;
        const int hr = sy < res - WS ? 0 : (sy < res - shift ? 1 : 2);
        const int wr = sx < res - WS ? 0 : (sx < res - shift ? 1 : 2);
        region[i] = (unsigned char)(3 * hr + wr);
    }
    for (int i = tid; i < SIDE * SIDE; i += NTHREADS) tbl[i] = bias[(int64_t)head * SIDE * SIDE + i] * LOG2E;
    __syncthreads();
    // K-hat (4 threads per key row) and V^T (task = 4 keys x 8 dims); keys past N are zero rows / zero columns
    for (int e = tid; e < NP * 4; e += NTHREADS) {
        const int i = e >> 2, c = e & 3;
        bf16x8_t raw = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
        if (i < N) raw = *(const bf16x8_t *)(base + C + rowmap[i] * ld + c * 8);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = lp_to_f32((uint16_t)raw[j]);
        float ss = sumsq8(raw);
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        const float inv = __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f));
        uint4 pk;
        pk.x = lp_pack2(v[0] * inv, v[1] * inv);
        pk.y = lp_pack2(v[2] * inv, v[3] * inv);
        pk.z = lp_pack2(v[4] * inv, v[5] * inv);
        pk.w = lp_pack2(v[6] * inv, v[7] * inv);
        *(uint4 *)(klds + i * 64 + ((c ^ ((-(i >> 2)) & 3)) << 4)) = pk;
    }
    for (int e = tid; e < (NP / 4) * 4; e += NTHREADS) {
        const int kg = e >> 2, c8 = e & 3;
        bf16x8_t r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r[i] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
            if (kg * 4 + i < N) r[i] = *(const bf16x8_t *)(base + 2 * C + rowmap[kg * 4 + i] * ld + c8 * 8);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint2 pk;
            pk.x = v_pair((uint16_t)r[0][j], (uint16_t)r[1][j]);
            pk.y = v_pair((uint16_t)r[2][j], (uint16_t)r[3][j]);
            *(uint2 *)(vt + (c8 * 8 + j) * VSTRIDE + kg * 8) = pk;
        }
    }
    __syncthreads();
    const float sc = fabsf(scale[head]) * LOG2E;
    const bool need_mask = shift > 0 && (wh == nwx - 1 || wwx == nwx - 1);
    const int fr = lane & 15, g = lane >> 4;
    for (int qt = wave; qt < NT; qt += NWAVES) {
        const int q = qt * 16 + fr;
        const int qrow = rowmap[q];
        bf16x8_t qf;
        {
            const bf16x8_t raw = *(const bf16x8_t *)(base + qrow * ld + g * 8);
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = lp_to_f32((uint16_t)raw[j]);
            float ss = sumsq8(raw);
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            const float inv = __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f));
            union { uint32_t w[4]; bf16x8_t v; } pk;
#pragma unroll
            for (int j = 0; j < 4; ++j) pk.w[j] = lp_pack2(v[2 * j] * inv, v[2 * j + 1] * inv);
            qf = pk.v;
        }
        const int yq = q / WS, xq = q - yq * WS, rq = region[q];
        const float *lt = tbl + (yq + WS - 1) * SIDE + xq + WS - 1;   // bias(q, key j) = lt[-(yj * SIDE + xj)]
        float s[NTP][4];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int krow = t * 16 + fr;
            const bf16x8_t kf = *(const bf16x8_t *)(klds + krow * 64 + ((g ^ ((-(krow >> 2)) & 3)) << 4));
            f32x4_t z = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            z = lp_mfma16(kf, qf, z);
            const int j0 = t * 16 + g * 4, yj = j0 / WS, xj = j0 - yj * WS;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = fmaf(z[r], sc, lt[-(yj * SIDE + xj + r)]);
                if (need_mask && region[j0 + r] != rq) v += -100.0f * LOG2E;
                s[t][r] = v;
                mx = fmaxf(mx, v);
            }
            if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = NT; t < NTP; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[t][r] = -INFINITY;   // the padding tile: probability 0
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const bf16x8_t ones = PV_ONES;
        f32x4_t o[2], osum = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        o[0] = o[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NTP / 2; ++u) {
            union { uint32_t w[4]; bf16x8_t v; } pk;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const float *sv = s[2 * u + (h >> 1)] + 2 * (h & 1);
                pk.w[h] = pv_pack2(__builtin_amdgcn_exp2f(sv[0] - mx), __builtin_amdgcn_exp2f(sv[1] - mx));
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const char *vrow = vt + (ct * 16 + fr) * VSTRIDE + (32 * u + 4 * g) * 2;
                union { uint2 h[2]; bf16x8_t v; } vf;
                vf.h[0] = *(const uint2 *)vrow;            // keys 32 u + 4 g .. + 3       (k slots j < 4)
                vf.h[1] = *(const uint2 *)(vrow + 32);     // keys 32 u + 16 + 4 g .. + 3  (k slots j >= 4)
                o[ct] = pv_mfma16(vf.v, pk.v, o[ct]);
            }
            osum = pv_mfma16(ones, pk.v, osum);
            if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        const float inv = __builtin_amdgcn_rcpf(osum[0]);
        uint16_t *orow = out + ((int64_t)frame * res * res + qrow) * C + head * HD + g * 4;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {   // o[ct][r] = context[query fr][dim ct * 16 + 4 g + r]
            uint2 pk;
            pk.x = lp_pack2(o[ct][0] * inv, o[ct][1] * inv);
            pk.y = lp_pack2(o[ct][2] * inv, o[ct][3] * inv);
            *(uint2 *)(orow + ct * 16) = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------
// 24 x 24 windows (Swin-V2-L at 384: 44 % of its encoder time on the kernel above, which keeps a 576-key score row in 144
// registers of a lane and therefore runs four waves per CU, gathers the bias per score and reads the shift regions per key):
// the streamed softmax of window_attention_stream_kernel for wide windows, shifted ones included.
//   * a probability is packed and fed to the PV MFMA as soon as its score exists (bounded heads: scale < 0, every unmasked logit
//     <= 0; the others take a first pass over the key tiles for the row maximum, scores recomputed): ~100 VGPRs, TWELVE waves
//     per workgroup -- 36 query tiles, three per wave;
//   * the bias of a lane's 4 consecutive keys is one ds_read_b128 from a column-reversed table kept in four copies shifted by
//     0..3 floats (the copy is picked by the query's column so that the read is aligned), as in the 16 x 16 kernels;
//   * shift masks: the regions of a lane's 4 keys are one 4-byte read, compared byte-wise with the query's region, only in the
//     windows that touch the wrapped edge.
// K-hat / V^T staging, MFMA operand layouts, the k-slot order of P and the ones-operand row sum are those of the kernel above.
template <int WS, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, NWAVES == 12 ? 3 : 6) void window_attention_wide_stream_kernel(
    const uint16_t *__restrict__ qkv, uint16_t *__restrict__ out, const float *__restrict__ bias,
    const float *__restrict__ scale, int res, int shift, int heads) {
    lp_kernel_entry();
    constexpr int N = WS * WS, NT = N / 16, NTHREADS = NWAVES * 64, QPW = NT / NWAVES;
    constexpr int SIDE = 2 * WS - 1, VSTRIDE = N * 2 + 32, TSTRIDE = (SIDE + 3 + 3) & ~3, TCOPY = SIDE * TSTRIDE;
    static_assert(N % 32 == 0 && NT % NWAVES == 0 && WS % 4 == 0, "whole 32-key PV steps, whole query tiles per wave, a lane's 4 keys inside one window row");
    extern __shared__ __attribute__((aligned(16))) char wsmem[];
    char *klds = wsmem;                                       // [N][32] bf16, 64-B rows, chunk ^= (-(row >> 2)) & 3
    char *vt = klds + N * 64;                                 // [32][VSTRIDE]
    int *rowmap = (int *)(vt + HD * VSTRIDE);                 // image token of window token i
    unsigned char *region = (unsigned char *)(rowmap + N);    // shift region of window token i (N bytes, 4-byte aligned)
    float *tbl = (float *)(region + N);                       // 4 x [SIDE][TSTRIDE] column-reversed bias * log2 e, copy c shifted by c
    const float LOG2E = 1.44269504088896340736f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwx = res / WS, nw = nwx * nwx;
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int head = b % heads; b /= heads;
    const int win = b % nw, frame = b / nw;
    const int wh = win / nwx, wwx = win - wh * nwx;
    const int C = heads * HD;
    const int64_t ld = 3 * (int64_t)C;
    const uint16_t *base = qkv + (int64_t)frame * res * res * ld + head * HD;
    for (int i = tid; i < N; i += NTHREADS) {
        const int wy = i / WS, wx = i - wy * WS;
        const int sy = wh * WS + wy, sx = wwx * WS + wx;
        int y = sy + shift, x = sx + shift;
        y = y >= res ? y - res : y;
        x = x >= res ? x - res : x;
        rowmap[i] = y * res + x;
        const int hr = sy < res - WS ? 0 : (sy < res - shift ? 1 : 2);
        const int wr = sx < res - WS ? 0 : (sx < res - shift ? 1 : 2);
        region[i] = (unsigned char)(3 * hr + wr);
    }
    for (int i = tid; i < SIDE * SIDE; i += NTHREADS) {
        const float v = bias[(int64_t)head * SIDE * SIDE + i] * LOG2E;
        const int at = (i / SIDE) * TSTRIDE + (SIDE - 1 - i % SIDE);
#pragma unroll
        for (int c = 0; c < 4; ++c) tbl[c * TCOPY + at + c] = v;
    }
    __syncthreads();
    for (int e = tid; e < N * 4; e += NTHREADS) {   // K-hat: 4 threads per key row
        const int i = e >> 2, c = e & 3;
        const bf16x8_t raw = *(const bf16x8_t *)(base + C + rowmap[i] * ld + c * 8);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = lp_to_f32((uint16_t)raw[j]);
        float ss = sumsq8(raw);
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        const float inv = __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f));
        uint4 pk;
        pk.x = lp_pack2(v[0] * inv, v[1] * inv);
        pk.y = lp_pack2(v[2] * inv, v[3] * inv);
        pk.z = lp_pack2(v[4] * inv, v[5] * inv);
        pk.w = lp_pack2(v[6] * inv, v[7] * inv);
        *(uint4 *)(klds + i * 64 + ((c ^ ((-(i >> 2)) & 3)) << 4)) = pk;
    }
    for (int e = tid; e < (N / 4) * 4; e += NTHREADS) {   // V^T: task = 4 keys x 8 dims, plain key order
        const int kg = e >> 2, c8 = e & 3;
        bf16x8_t r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *(const bf16x8_t *)(base + 2 * C + rowmap[kg * 4 + i] * ld + c8 * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint2 pk;
            pk.x = v_pair((uint16_t)r[0][j], (uint16_t)r[1][j]);
            pk.y = v_pair((uint16_t)r[2][j], (uint16_t)r[3][j]);
            *(uint2 *)(vt + (c8 * 8 + j) * VSTRIDE + kg * 8) = pk;
        }
    }
    __syncthreads();
    const float scraw = scale[head];
    const bool need_mask = shift > 0 && (wh == nwx - 1 || wwx == nwx - 1);   // workgroup-uniform
    const bool nomax = scraw < 0.f;                                          // workgroup-uniform
    const float sc = fabsf(scraw) * LOG2E;
    const f32x2_t sc2 = (f32x2_t){sc, sc};
    const bf16x8_t ones = PV_ONES;
    const int fr = lane & 15, g = lane >> 4;
    auto rows = [&](auto nomax_c, auto mask_c) {
        constexpr bool NOMAX = decltype(nomax_c)::value, MASK = decltype(mask_c)::value;
#pragma unroll 1
        for (int qi = 0; qi < QPW; ++qi) {
            const int q = (wave * QPW + qi) * 16 + fr;
            const int qrow = rowmap[q];
            bf16x8_t qf;
            {
                const bf16x8_t raw = *(const bf16x8_t *)(base + qrow * ld + g * 8);
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = lp_to_f32((uint16_t)raw[j]);
                float ss = sumsq8(raw);
                ss += __shfl_xor(ss, 16, 64);
                ss += __shfl_xor(ss, 32, 64);
                const float inv = __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f));
                union { uint32_t w[4]; bf16x8_t v; } pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk.w[j] = lp_pack2(v[2 * j] * inv, v[2 * j + 1] * inv);
                qf = pk.v;
            }
            const int tcopy = (q + 1) & 3;
            const float *lt = tbl + tcopy * (TCOPY + 1) + (q / WS + WS - 1) * TSTRIDE + WS - 1 - (q % WS);
            const uint32_t rq4 = 0x01010101u * region[q];
            auto scores = [&](int t, f32x2_t (&sv)[2]) {   // the 4 logits (log2 units) of this lane for key tile t
                const int krow = t * 16 + fr;
                const bf16x8_t kf = *(const bf16x8_t *)(klds + krow * 64 + ((g ^ ((-(krow >> 2)) & 3)) << 4));
                f32x4_t z = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                z = lp_mfma16(kf, qf, z);
                const int a = t * 4 + g, yj = a / (WS / 4), xj = (a - yj * (WS / 4)) * 4;   // keys j0 = 16 t + 4 g .. + 3 = (yj, xj ..)
                const f32x4_t b4 = *(const f32x4_t *)(lt - yj * TSTRIDE + xj);
                sv[0] = (f32x2_t){z[0], z[1]} * sc2 + (f32x2_t){b4[0], b4[1]};
                sv[1] = (f32x2_t){z[2], z[3]} * sc2 + (f32x2_t){b4[2], b4[3]};
                if (MASK) {
                    const uint32_t df = *(const uint32_t *)(region + a * 4) ^ rq4;   // a byte is non-zero where the regions differ
                    const float m = -100.0f * LOG2E;
                    sv[0][0] += (df & 0x000000ffu) ? m : 0.f;
                    sv[0][1] += (df & 0x0000ff00u) ? m : 0.f;
                    sv[1][0] += (df & 0x00ff0000u) ? m : 0.f;
                    sv[1][1] += (df & 0xff000000u) ? m : 0.f;
                }
            };
            float nmx = 0.f;
            if (!NOMAX) {
                float mx = -INFINITY;
#pragma unroll 4
                for (int t = 0; t < NT; ++t) {
                    f32x2_t sv[2];
                    scores(t, sv);
                    mx = fmaxf(fmaxf(mx, sv[0][0]), sv[0][1]);
                    mx = fmaxf(fmaxf(mx, sv[1][0]), sv[1][1]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                nmx = -mx;
            }
            const f32x2_t nm2 = (f32x2_t){nmx, nmx};
            f32x4_t o[2], osum = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            o[0] = o[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
            for (int u = 0; u < NT / 2; ++u) {
                union { uint32_t w[4]; bf16x8_t v; } pk;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x2_t sv[2];
                    scores(2 * u + h, sv);
#pragma unroll
                    for (int x = 0; x < 2; ++x) {
                        const f32x2_t d = NOMAX ? sv[x] : sv[x] + nm2;
                        pk.w[2 * h + x] = pv_pack2(__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1]));
                    }
                }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const char *vrow = vt + (ct * 16 + fr) * VSTRIDE + (32 * u + 4 * g) * 2;
                    union { uint2 h[2]; bf16x8_t v; } vf;
                    vf.h[0] = *(const uint2 *)vrow;            // keys 32 u + 4 g .. + 3       (k slots j < 4)
                    vf.h[1] = *(const uint2 *)(vrow + 32);     // keys 32 u + 16 + 4 g .. + 3  (k slots j >= 4)
                    o[ct] = pv_mfma16(vf.v, pk.v, o[ct]);
                }
                osum = pv_mfma16(ones, pk.v, osum);
            }
            const float inv = __builtin_amdgcn_rcpf(osum[0]);
            uint16_t *orow = out + ((int64_t)frame * res * res + qrow) * C + head * HD + g * 4;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {   // o[ct][r] = context[query fr][dim ct * 16 + 4 g + r]
                uint2 pk;
                pk.x = lp_pack2(o[ct][0] * inv, o[ct][1] * inv);
                pk.y = lp_pack2(o[ct][2] * inv, o[ct][3] * inv);
                *(uint2 *)(orow + ct * 16) = pk;
            }
        }
    };
    if (nomax) {
        if (need_mask) rows(std::true_type{}, std::true_type{});
        else rows(std::true_type{}, std::false_type{});
    } else {
        if (need_mask) rows(std::false_type{}, std::true_type{});
        else rows(std::false_type{}, std::false_type{});
    }
}

// ------------------------------------------------------------------------------------------
// y = LayerNorm(t) * gamma + beta;  x = (x_in ? x_in : 0) + y;  writes x (fp32) and its bf16
// shadow.  One wave per row, row in registers (two-pass statistics).
constexpr int MAXV = 8;
__global__ __launch_bounds__(256) void ln_residual_kernel(const float *__restrict__ t,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ beta,
                                                          const float *x_in, float *x_out,
                                                          uint16_t *__restrict__ xb, int64_t rows,
                                                          int width, float eps) {
    lp_kernel_entry();
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = width >> 8, tail = width & 255;
    const float *tr = t + row * width;
    float4 v[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const bool on = i < nv || (i == nv && lane * 4 < tail);
        v[i] = on ? *(const float4 *)(tr + i * 256 + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(sum) / (float)width;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const bool on = i < nv || (i == nv && lane * 4 < tail);
        if (on) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            sq += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)width + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int col = i * 256 + lane * 4;
        const bool on = i < nv || (i == nv && lane * 4 < tail);
        if (!on) continue;
        const float4 gm = *(const float4 *)(gamma + col), bt = *(const float4 *)(beta + col);
        float4 y = make_float4((v[i].x - mean) * rstd * gm.x + bt.x, (v[i].y - mean) * rstd * gm.y + bt.y,
                               (v[i].z - mean) * rstd * gm.z + bt.z, (v[i].w - mean) * rstd * gm.w + bt.w);
        if (x_in) {
            const float4 r = *(const float4 *)(x_in + row * width + col);
            y = make_float4(r.x + y.x, r.y + y.y, r.z + y.z, r.w + y.w);
        }
        *(float4 *)(x_out + row * width + col) = y;
        uint2 pk;
        pk.x = lp_pack2(y.x, y.y);
        pk.y = lp_pack2(y.z, y.w);
        *(uint2 *)(xb + row * width + col) = pk;
    }
}

// ------------------------------------------------------------------------------------------
// PatchMerging gather (torch2scripts.py:353-358): out[b, (Y, X), :] = xb[b, 2Y+dy, 2X+dx, :] for
// (dy, dx) = (0,0), (1,0), (0,1), (1,1) concatenated.  16-byte chunks.
__global__ __launch_bounds__(256) void merge_gather_kernel(const uint16_t *__restrict__ xb,
                                                           uint16_t *__restrict__ out, int64_t total,
                                                           int res, int c) {
    const int c8 = c >> 3, half = res >> 1;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int chunk = (int)(e % (4 * c8));
        const int64_t tok = e / (4 * c8);
        const int part = chunk / c8, cc = chunk - part * c8;
        const int X = (int)(tok % half), Y = (int)((tok / half) % half);
        const int64_t b = tok / ((int64_t)half * half);
        const int dy = part & 1, dx = part >> 1;
        const int64_t src = ((b * res + 2 * Y + dy) * res + 2 * X + dx) * c + cc * 8;
        *(uint4 *)(out + e * 8) = *(const uint4 *)(xb + src);
    }
}

}  // namespace

int launch_window_attention(const uint16_t *qkv, uint16_t *out, const float *bias, const float *scale,
                            int frames, int res, int ws, int shift, int heads, hipStream_t stream) {
    VSC_REQUIRE(qkv && out && bias && scale, "window_attention: null operand");  // bias: [heads, (2w-1)^2] compact table
    VSC_REQUIRE(res % ws == 0 && shift >= 0 && shift < ws, "window_attention: res %d window %d shift %d", res, ws,
                shift);
    const int nw = (res / ws) * (res / ws);
    const int64_t grid = (int64_t)frames * nw * heads;
    VSC_REQUIRE(grid > 0 && grid < (1ll << 31), "window_attention: grid");
#ifdef VSC_ATTN_ABLATION
    if (ws == 16)
        if (const char *e = vsc_opt(OPT_WATTN_ABL)) {
#define VSC_WABL_CASE(A) case A: hipLaunchKernelGGL((window_attention_kernel<16, A>), dim3((unsigned)grid), dim3(512), 0, stream, qkv, out, bias, scale, res, ws, shift, heads); VSC_CHECK_LAUNCH(); return VSC_OK;
            switch (atoi(e)) { VSC_WABL_CASE(1) VSC_WABL_CASE(2) VSC_WABL_CASE(3) VSC_WABL_CASE(8) VSC_WABL_CASE(16) VSC_WABL_CASE(24) VSC_WABL_CASE(27) default: break; }
        }
#endif
    const char *so = vsc_opt(OPT_WATTN_STREAM);   // diagnostic: 0 = the row-in-registers kernel for unshifted windows as well
    if (ws == 16 && shift == 0 && !(so && so[0] == '0'))
        hipLaunchKernelGGL(window_attention_stream_kernel, dim3((unsigned)grid), dim3(512), 0, stream, qkv, out, bias, scale, res, heads);
    else if (ws == 16 && !(so && so[0] == '0')) {   // shifted 16 x 16 windows: the streamed kernel with masks (eight waves, 52 KiB: three workgroups per CU)
        constexpr int n = 256, side = 31, tstride = (side + 3 + 3) & ~3;
        constexpr int smem = n * 64 + HD * (n * 2 + 32) + n * 4 + n + (4 * side * tstride + 4) * 4;
        hipLaunchKernelGGL((window_attention_wide_stream_kernel<16, 8>), dim3((unsigned)grid), dim3(512), smem, stream, qkv, out, bias, scale, res, shift, heads);
    } else if (ws == 16)
        hipLaunchKernelGGL(window_attention_kernel<16>, dim3((unsigned)grid), dim3(512), 0, stream, qkv, out, bias,
                           scale, res, ws, shift, heads);
    else if (ws == 8)
        hipLaunchKernelGGL(window_attention_kernel<4>, dim3((unsigned)grid), dim3(128), 0, stream, qkv, out, bias,
                           scale, res, ws, shift, heads);
    else if (ws == 24 && !(so && so[0] == '0')) {
        constexpr int n = 576, side = 47, tstride = (side + 3 + 3) & ~3;
        constexpr int smem = n * 64 + HD * (n * 2 + 32) + n * 4 + n + (4 * side * tstride + 4) * 4;
        static bool attr_set[16] = {};
        int dev = 0;
        VSC_CHECK_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 16 || !attr_set[dev]) {
            VSC_CHECK_HIP(hipFuncSetAttribute((const void *)window_attention_wide_stream_kernel<24, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            if (dev >= 0 && dev < 16) attr_set[dev] = true;
        }
        hipLaunchKernelGGL((window_attention_wide_stream_kernel<24, 12>), dim3((unsigned)grid), dim3(768), smem, stream, qkv, out, bias, scale, res, shift, heads);
    } else if (ws == 24 || ws == 12) {
        auto smem_of = [](int w) {
            const int n = w * w, np = ((n / 16 + 1) & ~1) * 16, side = 2 * w - 1;
            return np * 64 + HD * (np * 2 + 32) + n * 4 + side * side * 4 + n + 16;
        };
        const int smem = smem_of(ws);
        static bool attr_set[16][2] = {};
        int dev = 0;
        VSC_CHECK_HIP(hipGetDevice(&dev));
        const int wi = ws == 24;
        if (dev < 0 || dev >= 16 || !attr_set[dev][wi]) {
            if (wi) VSC_CHECK_HIP(hipFuncSetAttribute((const void *)window_attention_wide_kernel<24>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            else VSC_CHECK_HIP(hipFuncSetAttribute((const void *)window_attention_wide_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            if (dev >= 0 && dev < 16) attr_set[dev][wi] = true;
        }
        if (wi) hipLaunchKernelGGL(window_attention_wide_kernel<24>, dim3((unsigned)grid), dim3(256), smem, stream, qkv, out, bias, scale, res, shift, heads);
        else hipLaunchKernelGGL(window_attention_wide_kernel<12>, dim3((unsigned)grid), dim3(256), smem, stream, qkv, out, bias, scale, res, shift, heads);
    } else
        VSC_REQUIRE(false, "window_attention: window %d unsupported (8, 12, 16 or 24)", ws);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

int launch_ln_residual(const float *t, const float *gamma, const float *beta, const float *x_in, float *x_out,
                       uint16_t *xb, int64_t rows, int width, float eps, hipStream_t stream) {
    VSC_REQUIRE(t && gamma && beta && x_out && xb && rows > 0, "ln_residual: null/empty");
    VSC_REQUIRE(width % 4 == 0 && width <= MAXV * 256, "ln_residual: width %d unsupported", width);
    hipLaunchKernelGGL(ln_residual_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, t, gamma, beta,
                       x_in, x_out, xb, rows, width, eps);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

int launch_merge_gather(const uint16_t *xb, uint16_t *out, int64_t frames, int res, int c, hipStream_t stream) {
    VSC_REQUIRE(xb && out && res % 2 == 0 && c % 8 == 0, "merge_gather: res %d c %d", res, c);
    const int64_t total = frames * (res / 2) * (res / 2) * 4 * (c / 8);
    int64_t blocks = (total + 255) / 256;
    blocks = blocks > 16384 ? 16384 : blocks;
    hipLaunchKernelGGL(merge_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, xb, out, total, res, c);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}
