// Fused MLP of a Swin-V2 block for the narrow stages (C = 128, 256), res-post-norm (train/train_v115/torch2scripts.py:297-300):
//     x += LayerNorm(GELU(xb W1^T + b1) W2^T + b2) * gamma + beta,     xb = bf16(x)
// in ONE kernel: the hidden activations [M, 4C] never exist in memory.  As two GEMM launches (fc1 with the GELU write-out, fc2
// with the LayerNorm write-out) the block moved 4.8 GB per 256 frames of stage 0, 2.1 GB of it the hidden tensor out and
// back in, at the HBM roof throughout (fc1 405 us + fc2 485 us); what must move is x in / out and the shadow: 1.6 GB.
//
// CDNA4 mapping.  Eight waves per workgroup split the ROWS only: a wave owns RW = 32 (C = 128) or 16 (C = 256) rows from the
// first load to the last store, so nothing is ever exchanged between waves and the only workgroup barrier is the one that
// publishes a weight chunk.
//   * the wave's rows of xb are loaded ONCE, straight from global memory into MFMA operand registers (32 VGPRs);
//   * the hidden axis is walked in chunks of 64 units.  Per chunk: W1[chunk, :] (64 x C) and W2[:, chunk] (C x 64) arrive by
//     LDS-DMA in a two-slot ring (the next chunk's DMA is issued behind the barrier that publishes the current one);
//     GEMM 1: hacc[row][64] = xb . W1c^T (swapped operands: a lane holds 4 consecutive hidden units of one row);
//     bias + GELU + bf16 rounding in registers -- the same rounding point as the unfused fc1 write-out;
//     GEMM 2: oacc[row][C] += h . W2c^T with h taken FROM THOSE REGISTERS: GEMM 1's accumulator layout (row = lane & 15,
//     hidden 16 j + 4 (lane >> 4) + r) is a legal B operand of the 16x16x32 MFMA once the contraction slots are assigned as
//     slot (g, i < 4) = hidden 32 s + 4 g + i, slot (g, i >= 4) = hidden 32 s + 16 + 4 g + i - 4 (the attention kernels' trick);
//     W2's hidden axis is stored in that order by the host (vsc_swin_finalize), so its operand is one ds_read_b128;
//   * LayerNorm over the wave's own rows: a row's C values sit in the four lanes (lane & 15, 0..3): two-pass statistics in
//     registers + two xor-shuffles.  The output columns of a lane are made 8 consecutive ones per pair of 16-column MFMA
//     tiles (row 4 q + r of tile jo <-> column 32 (jo >> 1) + 8 q + 4 (jo & 1) + r: only W2's LDS ROW a lane reads changes),
//     so x is read and written as whole 128-byte lines per row and the shadow as 64-byte halves.
// LDS rows are swizzled against ds_read_b128's 16-lane groups: W1c (256- / 512-byte rows) chunk ^= row & 15; W2c (128-byte
// rows, read in the permuted row order above) chunk ^= ((row >> 1) & 1) | ((row >> 3) & 3) << 1 -- both conflict-free
// (checked exhaustively, tools/micro/lds_swizzle_check.py).  LDS-DMA writes lane-linear, so the permutation is applied to the
// per-lane SOURCE address.
#include "common.h"
#include "gelu_poly.h"

namespace {

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

struct MlpArgs {
    const uint16_t *w1;    // [4C, C]
    const float *b1;       // [4C]
    const uint16_t *w2p;   // [C, 4C], hidden axis in consumption order (swin_mlp_permute_hidden)
    const float *b2, *gamma, *beta;   // [C]
    float *x;              // [m, C] residual stream, updated in place
    uint16_t *xb;          // [m, C] its bf16 shadow: the MLP's input, replaced by the shadow of the new x
    int64_t m;
    float eps;
    // PROJ (swin_mlp_kernel<C, NW, ABL, true>): the attention projection and its LayerNorm in front, in the same kernel --
    //     x1 = x + LayerNorm(att Wp^T + bp) * gamma1 + beta1;  then the MLP block on x1 (its shadow never leaves the registers)
    const uint16_t *att;   // [m, C] attention output
    const uint16_t *wp;    // [C, C]
    const float *bp, *gamma1, *beta1;   // [C]
    int stagger;           // NW = 4 (two workgroups per CU): the second resident workgroup of every CU starts this many x 8 128 cycles late
};

// Empty volatile asm through which every element of a step's results passes: the step's arithmetic cannot be sunk below it nor
// the next step's hoisted above it -- __builtin_amdgcn_sched_barrier alone only binds the machine scheduler, and the IR
// passes before it re-serialise the independent polynomial chains (one pair at a time, every v_pk_fma_f32 waiting for the
// one before: 2.5 x the cycles of the lock-step order).
template <int NP>
__device__ __forceinline__ void pin(f32x2_t (&q)[NP]) {
#pragma unroll
    for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(q[i]));
}

__device__ __forceinline__ int w2_swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }

// ABL: ablation bits of the diagnostic build (-DVSC_MLP_ABLATION, VSC_SWIN_MLP_ABL): 1 no GELU polynomial, 2 no MFMAs, 4 weight
// chunks staged once (no LDS-DMA in the loop), 8 no residual loads / stores, 16 no workgroup barrier in the loop
// NW waves per workgroup: eight.  One workgroup fits a CU (registers), so its phases add up -- ablations at 256 frames of
// stage 0 (705 us): GELU 235, MFMAs 162, residual loads + stores 273 (at the HBM roof while they run), weight DMA + barriers
// 35, everything else 100.  Measured and dropped: four waves per workgroup so that two workgroups share a CU and one
// computes while the other moves its rows -- 710 us, no gain (and 882 us when the launch bounds let the compiler spread to one
// workgroup per CU): the two pipes do not overlap in this instruction mix whichever waves issue them.
// PROJ: the block's first half fused in front (round 4): a wave multiplies its rows of the attention output by Wp (C x C, resident in
// LDS beside the ring), normalises, adds x -- the accumulator layout of that product is the layout of GEMM 2's, so the LayerNorm code
// is shared, and its bf16 shadow (8 consecutive columns per lane and column pair) IS the B operand of GEMM 1: x1 and its shadow
// never go to memory (1.6 of the 3.2 GB the two launches moved at 256 frames of stage 0), the residual of the second LayerNorm is
// taken from the registers.
// SEQ: the chunk's vector work as ONE run between two runs of MFMAs (GEMM 1 | GELU of all four tiles | GEMM 2) instead of the polynomial's
// steps between MFMA groups: a vector instruction behind an MFMA costs ~16 cycles where it costs 5 behind another vector instruction
// (tools/micro/single_wave_issue.hip)
template <int C, int NW, int ABL = 0, bool PROJ = false, bool SEQ = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void swin_mlp_kernel(MlpArgs p) {
    lp_kernel_entry();
#define MFMA(a, b, c) ((ABL & 2) ? (c) + (f32x4_t){(float)(a)[0], (float)(b)[0], 0.f, 0.f} : lp_mfma16(a, b, c))
    constexpr int RW = C == 128 ? 32 : 16, MT = RW / 16, R = NW * RW;
    constexpr int KS1 = C / 32, JO = C / 16, H = 4 * C, HC = 64, NCH = H / HC;
    constexpr int W1B = HC * C * 2, W2B = C * HC * 2, CHUNK = W1B + W2B;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *ring = lds;                          // 2 x CHUNK
    float *b1s = (float *)(lds + 2 * CHUNK);   // [H]
    float *b2s = b1s + H, *gs = b2s + C, *bs = gs + C;
    float *bps = bs + C, *g1s = bps + C, *be1s = g1s + C;   // PROJ: bp, gamma1, beta1
    // PROJ: Wp [C][C] bf16, 2 C-byte rows, chunk ^= row & 15 -- beside the ring at C = 128 (32 KiB); at C = 256 it is as large as the ring
    // (128 KiB) and lives IN it until the projection is done, the MLP's first weight chunk being requested only then
    // (NW = 4: two workgroups per CU -- 2 x (ring + bias rows) = 138 KiB at C = 128 leaves no room for a resident Wp either)
    // At C = 128 Wp is exactly one ring slot (32 KiB): it takes slot 1 while chunk 0 arrives in slot 0, and chunk 1's request -- issued behind
    // the first chunk barrier, which every wave reaches with its projection done -- overwrites it.  No extra barrier, no exposed request.
    constexpr bool WP_SLOT1 = PROJ && NW == 4 && C * C * 2 == CHUNK;
    constexpr bool WP_IN_RING = PROJ && !WP_SLOT1 && (C * C * 2 > 32 * 1024 || NW == 4);
    char *wps = WP_SLOT1 ? ring + CHUNK : WP_IN_RING ? ring : (char *)(be1s + C);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, quad = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * R + wave * RW;
    if (NW == 4 && p.stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
        // Two workgroups share a CU; they are dispatched together, take the same time and would stay in phase for the whole launch -- both
        // moving rows, then both computing.  The first round's second workgroups (workgroups 256 .. 511: the dispatcher gives every CU one
        // workgroup before any gets two) start half a tile late; their successors inherit the offset.
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }

    // ---- LDS-DMA of hidden chunk ch into ring slot `slot`: instruction q of a matrix covers its bytes [1024 q, 1024 q + 1024)
    auto stage = [&](int ch, int slot) {
        char *dst = ring + slot * CHUNK;
#pragma unroll
        for (int qq = 0; qq < W1B / 1024 / NW; ++qq) {
            const int q = qq * NW + wave;
            const int off = q * 1024 + lane * 16;
            const int row = off / (2 * C), cp = (off % (2 * C)) >> 4;
            const int c = cp ^ (row & 15);
            __builtin_amdgcn_global_load_lds((gptr_t)(p.w1 + ((int64_t)ch * HC + row) * C + c * 8), (lptr_t)(dst + q * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int qq = 0; qq < W2B / 1024 / NW; ++qq) {
            const int q = qq * NW + wave;
            const int off = q * 1024 + lane * 16;
            const int row = off >> 7, cp = (off & 127) >> 4;
            const int c = cp ^ w2_swz(row);
            __builtin_amdgcn_global_load_lds((gptr_t)(p.w2p + (int64_t)row * H + ch * HC + c * 8), (lptr_t)(dst + W1B + q * 1024), 16, 0, 0);
        }
    };
    if (!WP_IN_RING) stage(0, 0);
    for (int i = tid; i < H; i += NW * 64) b1s[i] = p.b1[i];
    for (int i = tid; i < C; i += NW * 64) {
        b2s[i] = p.b2[i];
        gs[i] = p.gamma[i];
        bs[i] = p.beta[i];
    }
    // ---- this wave's rows of xb as B operands of GEMM 1: xf[mt][ks] = xb[row0 + 16 mt + fr][32 ks + 8 quad .. + 7]
    bf16x8_t xf[MT][KS1];
    f32x4_t xres[PROJ ? MT : 1][PROJ ? JO : 1];   // PROJ: x1 (fp32), the residual of the second LayerNorm
    // LayerNorm of the wave's rows of `acc` (+ bias), times gamma, plus beta, plus the residual rows `res`: y[mt][jo] in acc's layout
    // -- lane (fr, quad) holds, of row fr, the columns 32 p + 8 quad + 4 t + r in [mt][2 p + t][r]
    auto layer_norm = [&](f32x4_t (&acc)[MT][JO], const float *bias_s, const float *gamma_s, const float *beta_s, auto &&emit) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float sum = 0.f;
#pragma unroll
            for (int jo = 0; jo < JO; ++jo) {
                acc[mt][jo] += *(const f32x4_t *)(bias_s + 32 * (jo >> 1) + 8 * quad + 4 * (jo & 1));
                sum += (acc[mt][jo][0] + acc[mt][jo][1]) + (acc[mt][jo][2] + acc[mt][jo][3]);
            }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.0f / C);
            float sq = 0.f;
#pragma unroll
            for (int jo = 0; jo < JO; ++jo) {
                acc[mt][jo] -= (f32x4_t){mean, mean, mean, mean};
                sq += (acc[mt][jo][0] * acc[mt][jo][0] + acc[mt][jo][1] * acc[mt][jo][1]) +
                      (acc[mt][jo][2] * acc[mt][jo][2] + acc[mt][jo][3] * acc[mt][jo][3]);
            }
            sq += __shfl_xor(sq, 16, 64);
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = rsqrtf(sq * (1.0f / C) + p.eps);
#pragma unroll
            for (int pp = 0; pp < JO / 2; ++pp) {
                const int col = 32 * pp + 8 * quad;
                f32x4_t nrm[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x4_t g4 = *(const f32x4_t *)(gamma_s + col + 4 * t), b4 = *(const f32x4_t *)(beta_s + col + 4 * t);
#pragma unroll
                    for (int r = 0; r < 4; ++r) nrm[t][r] = acc[mt][2 * pp + t][r] * rstd * g4[r] + b4[r];
                }
                emit(mt, pp, col, nrm, rstd);
            }
        }
    };
    if (!PROJ) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int64_t row = row0 + mt * 16 + fr;
            row = row < p.m ? row : p.m - 1;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) xf[mt][ks] = *(const bf16x8_t *)(p.xb + row * C + ks * 32 + quad * 8);
        }
    } else {
        // Wp into LDS (2 C-byte rows, chunk ^= row & 15), the wave's attention rows and residual rows into registers
#pragma unroll
        for (int qq = 0; qq < C * C * 2 / 1024 / NW; ++qq) {
            const int q = qq * NW + wave;
            const int off = q * 1024 + lane * 16;
            const int row = off / (2 * C), cp = (off % (2 * C)) >> 4;
            const int c = cp ^ (row & 15);
            __builtin_amdgcn_global_load_lds((gptr_t)(p.wp + (int64_t)row * C + c * 8), (lptr_t)(wps + q * 1024), 16, 0, 0);
        }
        for (int i = tid; i < C; i += NW * 64) {
            bps[i] = p.bp[i];
            g1s[i] = p.gamma1[i];
            be1s[i] = p.beta1[i];
        }
        bf16x8_t af[MT][KS1];
        f32x4_t xin1[MT][JO];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int64_t row = row0 + mt * 16 + fr;
            row = row < p.m ? row : p.m - 1;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) af[mt][ks] = *(const bf16x8_t *)(p.att + row * C + ks * 32 + quad * 8);
#pragma unroll
            for (int jo = 0; jo < JO; ++jo) xin1[mt][jo] = *(const f32x4_t *)(p.x + row * C + 32 * (jo >> 1) + 8 * quad + 4 * (jo & 1));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the MLP's first weight chunk, issued above, is waited for here as well)
        __syncthreads();
        f32x4_t pacc[MT][JO];
#pragma unroll
        for (int jo = 0; jo < JO; ++jo) {
            const int n = 32 * (jo >> 1) + 8 * (fr >> 2) + 4 * (jo & 1) + (fr & 3);   // Wp row = output column (GEMM 2's row order)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const bf16x8_t wf = *(const bf16x8_t *)(wps + n * (2 * C) + (((4 * ks + quad) ^ (n & 15)) << 4));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    pacc[mt][jo] = MFMA(wf, af[mt][ks], (ks == 0 ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : pacc[mt][jo]));
            }
        }
        if (WP_IN_RING) {
            __syncthreads();   // every wave is done reading Wp: the ring is the MLP's from here on
            stage(0, 0);
        }
        layer_norm(pacc, bps, g1s, be1s, [&](int mt, int pp, int col, f32x4_t (&nrm)[2], float) {
            (void)col;
            union { uint32_t w[4]; bf16x8_t v; } pk;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                xres[mt][2 * pp + t] = xin1[mt][2 * pp + t] + nrm[t];
                pk.w[2 * t] = lp_pack2(xres[mt][2 * pp + t][0], xres[mt][2 * pp + t][1]);
                pk.w[2 * t + 1] = lp_pack2(xres[mt][2 * pp + t][2], xres[mt][2 * pp + t][3]);
            }
            xf[mt][pp] = pk.v;   // columns 32 pp + 8 quad .. + 7 of row fr: GEMM 1's B operand of k-step pp
        });
    }
    f32x4_t oacc[MT][JO];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jo = 0; jo < JO; ++jo) oacc[mt][jo] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

#pragma nounroll
    for (int ch = 0; ch < NCH; ++ch) {
        // this wave's pieces of chunk ch have landed (and, the first time round, its bias rows are written); behind the barrier
        // every wave's have, and every wave is past its last read of the other slot
        // (the explicit vmcnt wait is essential: an LDS-DMA is a pending LDS write that only the VM counter tracks, and
        //  __syncthreads() compiles to a bare s_barrier here -- without it a wave can read a chunk whose pieces are still in flight)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(ABL & 16) || ch == 0) __syncthreads();
        if (ch + 1 < NCH && (!(ABL & 4) || ch == 0)) stage(ch + 1, (ch + 1) & 1);
        const char *w1s = ring + (ch & 1) * CHUNK, *w2s = w1s + W1B;
        auto w1f = [&](int j, int ks) {   // W1 chunk rows 16 j + fr, input channels 32 ks + 8 quad .. + 7
            return *(const bf16x8_t *)(w1s + (16 * j + fr) * (2 * C) + (((4 * ks + quad) ^ fr) << 4));
        };
        auto w2f = [&](int jo, int sk) {  // W2 row = output column 32 (jo >> 1) + 8 (fr >> 2) + 4 (jo & 1) + (fr & 3), k-step sk
            const int n = 32 * (jo >> 1) + 8 * (fr >> 2) + 4 * (jo & 1) + (fr & 3);
            return *(const bf16x8_t *)(w2s + n * 128 + (((4 * sk + quad) ^ w2_swz(n)) << 4));
        };
        // The chunk as a four-phase software pipeline inside the wave (two waves share a SIMD and leave every barrier in
        // phase, so matrix work and vector work overlap only if each wave interleaves them itself):
        //   A  GEMM 1 for hidden tiles j = 0, 1                         (matrix pipe)
        //   B  bias + GELU of tiles 0, 1   ||  GEMM 1 for tiles 2, 3    (MFMA groups spread over the polynomial steps)
        //   C  bias + GELU of tiles 2, 3   ||  GEMM 2, k-step 0         (h of tiles 0, 1)
        //   D  GEMM 2, k-step 1                                         (matrix pipe)
        // Operand fragments are read one MFMA group ahead of their use.
        f32x4_t hacc[MT][4];
        union { uint32_t w[4]; bf16x8_t v; } hf[MT][2];
        bf16x8_t wq[2];   // fragment double buffer
        // ---- A
        wq[0] = w1f(0, 0);
#pragma unroll
        for (int e = 0; e < 2 * KS1; ++e) {
            // k-step outermost: consecutive MFMAs go to different accumulators (ks innermost made every MFMA wait for the
            // one before it: 39 % of the wave cycles were issue stalls)
            const int j = e & 1, ks = e >> 1;
            wq[(e + 1) & 1] = e + 1 < 2 * KS1 ? w1f((e + 1) & 1, (e + 1) >> 1) : w1f(2, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                hacc[mt][j] = MFMA(wq[e & 1], xf[mt][ks], (ks == 0 ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : hacc[mt][j]));
        }
        // the lock-step GELU of NP pairs with `fill(step)` issued behind each of its NSTEP steps
        constexpr int NP = MT * 4, NSTEP = vscgelu::GELU_DEG + 3;
        auto gelu_with = [&](f32x2_t (&x)[NP], auto &&fill) {
            if (ABL & 1) {
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) {
                    fill(st);
                    __builtin_amdgcn_sched_barrier(0);
                }
                return;
            }
            // (measured and dropped: the same polynomial on scalar v_fma_f32 -- twice the instructions, and the vector pipe is the
            //  limiter: GELU 235 -> 474 us of the stage-0 launch)
            using vscgelu::GELU_C;
            using vscgelu::GELU_DEG;
            using vscgelu::GELU_U;
            using vscgelu::GELU_ZS;
            f32x2_t t[NP], z[NP], q[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                t[i][0] = __builtin_amdgcn_fmed3f(x[i][0], -GELU_U, GELU_U);
                t[i][1] = __builtin_amdgcn_fmed3f(x[i][1], -GELU_U, GELU_U);
            }
            pin(t);
            fill(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NP; ++i) z[i] = __builtin_elementwise_fma(t[i] * t[i], (f32x2_t){GELU_ZS, GELU_ZS}, (f32x2_t){-1.0f, -1.0f});
            pin(z);
            fill(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma((f32x2_t){GELU_C[GELU_DEG], GELU_C[GELU_DEG]}, z[i], (f32x2_t){GELU_C[GELU_DEG - 1], GELU_C[GELU_DEG - 1]});
            pin(q);
            fill(2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = GELU_DEG - 2; c >= 0; --c) {
#pragma unroll
                for (int i = 0; i < NP; ++i) q[i] = __builtin_elementwise_fma(q[i], z[i], (f32x2_t){GELU_C[c], GELU_C[c]});
                pin(q);
                fill(GELU_DEG + 1 - c);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) x[i] = x[i] * __builtin_elementwise_fma(t[i], q[i], (f32x2_t){0.5f, 0.5f});
            pin(x);
            fill(NSTEP - 1);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto bias_of = [&](int jp, f32x2_t (&v)[NP]) {   // tiles 2 jp, 2 jp + 1 of every 16-row tile, + bias
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * jp + jj;
                    const f32x4_t bz = *(const f32x4_t *)(b1s + ch * HC + 16 * j + 4 * quad);
                    v[mt * 4 + jj * 2] = (f32x2_t){hacc[mt][j][0] + bz[0], hacc[mt][j][1] + bz[1]};
                    v[mt * 4 + jj * 2 + 1] = (f32x2_t){hacc[mt][j][2] + bz[2], hacc[mt][j][3] + bz[3]};
                }
        };
        auto pack_to = [&](int jp, const f32x2_t (&v)[NP]) {   // -> hf[mt][jp]: the B operand of GEMM 2's k-step jp
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    hf[mt][jp].w[jj * 2 + 0] = lp_pack2(v[mt * 4 + jj * 2][0], v[mt * 4 + jj * 2][1]);
                    hf[mt][jp].w[jj * 2 + 1] = lp_pack2(v[mt * 4 + jj * 2 + 1][0], v[mt * 4 + jj * 2 + 1][1]);
                }
        };
        // ---- B: GELU(tiles 0, 1) || GEMM 1 (tiles 2, 3): 2 KS1 MFMA groups over the NSTEP steps
        {
            f32x2_t v[NP];
            bias_of(0, v);
            constexpr int NE = 2 * KS1;
            auto fill_b = [&](int step) {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    if (e * NSTEP / NE != step) continue;
                    const int j = 2 + (e & 1), ks = e >> 1;
                    // entry e sits in wq[e & 1] (phase A left entry 0 of this list in wq[(2 KS1) & 1] = wq[0])
                    wq[(e + 1) & 1] = e + 1 < NE ? w1f(2 + ((e + 1) & 1), (e + 1) >> 1) : w2f(0, 0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        hacc[mt][j] = MFMA(wq[e & 1], xf[mt][ks], (ks == 0 ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : hacc[mt][j]));
                }
            };
            if (SEQ) {
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) fill_b(st);
                __builtin_amdgcn_sched_barrier(0);
                gelu_with(v, [&](int) {});
            } else gelu_with(v, fill_b);
            pack_to(0, v);
        }
        // ---- C: GELU(tiles 2, 3) || GEMM 2, k-step 0: JO MFMA groups
        {
            f32x2_t v[NP];
            bias_of(1, v);
            auto fill_c = [&](int step) {
#pragma unroll
                for (int e = 0; e < JO; ++e) {
                    if (e * NSTEP / JO != step) continue;
                    wq[(e + 1) & 1] = e + 1 < JO ? w2f(e + 1, 0) : w2f(0, 1);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) oacc[mt][e] = MFMA(wq[e & 1], hf[mt][0].v, oacc[mt][e]);
                }
            };
            if (SEQ) {
                gelu_with(v, [&](int) {});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) fill_c(st);
            } else gelu_with(v, fill_c);
            pack_to(1, v);
        }
        // ---- D: GEMM 2, k-step 1
#pragma unroll
        for (int e = 0; e < JO; ++e) {
            if (e + 1 < JO) wq[(e + 1) & 1] = w2f(e + 1, 1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) oacc[mt][e] = MFMA(wq[e & 1], hf[mt][1].v, oacc[mt][e]);
        }
    }

    // ---- LayerNorm of the wave's rows + residual + shadow.  Lane (fr, quad) holds, of row fr, the columns
    //      32 p + 8 quad + 4 t + r (p = 0 .. C/32 - 1, t = 0, 1) in oacc[mt][2 p + t][r].
    // The residual rows are requested first, all of them, and arrive under the statistics (requested where they are used,
    // between the stores of the same rows, every 16-byte piece was its own memory round trip); PROJ: they are xres.
    f32x4_t xin[PROJ ? 1 : MT][PROJ ? 1 : JO];
    if (!PROJ) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int64_t row = row0 + mt * 16 + fr;
            row = row < p.m ? row : p.m - 1;
#pragma unroll
            for (int jo = 0; jo < JO; ++jo)
                xin[mt][jo] = (ABL & 8) ? (f32x4_t){1.f, 2.f, 3.f, 4.f} : *(const f32x4_t *)(p.x + row * C + 32 * (jo >> 1) + 8 * quad + 4 * (jo & 1));
        }
    }
    layer_norm(oacc, b2s, gs, bs, [&](int mt, int pp, int col, f32x4_t (&nrm)[2], float rstd) {
        const int64_t row = row0 + mt * 16 + fr;
        if (row < p.m && !((ABL & 8) && rstd != 12345.f)) {
            float *xr = p.x + row * C;
            uint16_t *xbr = p.xb + row * C;
            f32x4_t y[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                y[t] = (PROJ ? xres[mt][2 * pp + t] : xin[mt][2 * pp + t]) + nrm[t];
                *(f32x4_t *)(xr + col + 4 * t) = y[t];
            }
            uint4 pk;
            pk.x = lp_pack2(y[0][0], y[0][1]);
            pk.y = lp_pack2(y[0][2], y[0][3]);
            pk.z = lp_pack2(y[1][0], y[1][1]);
            pk.w = lp_pack2(y[1][2], y[1][3]);
            *(uint4 *)(xbr + col) = pk;
        }
    });
#undef MFMA
}

template <int C>
int launch_proj_c(const MlpArgs &a, hipStream_t stream) {
    constexpr int NW = 8, R = NW * (C == 128 ? 32 : 16);
    constexpr int smem = 2 * (2 * 64 * C * 2) + (4 * C + 3 * C + 3 * C) * 4 + (C * C * 2 > 32 * 1024 ? 0 : C * C * 2);
    static_assert(smem <= 160 * 1024 && C * C * 2 <= 2 * (2 * 64 * C * 2), "Wp fits neither beside the ring nor in it");
    static bool attr_set[16] = {};
    int dev = 0;
    VSC_CHECK_HIP(hipGetDevice(&dev));
    if (dev >= 16 || !attr_set[dev]) {
        VSC_CHECK_HIP(hipFuncSetAttribute((const void *)swin_mlp_kernel<C, NW, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        if (dev < 16) attr_set[dev] = true;
    }
    const int64_t grid = (a.m + R - 1) / R;
    VSC_REQUIRE(grid < (1ll << 31), "swin_mlp: grid too large");
    // C = 128 (round 6): two 4-wave workgroups per CU instead of one 8-wave one -- Wp rides in ring slot 1 (2 x 69 KiB of LDS), 128 rows per
    // workgroup.  One workgroup's row loads, its first weight request and its stores now sit under the other's chunk loop: 705 -> 638 us per
    // launch of 256 frames of stage 0 on one stream, the same bits (tools/micro/swin_mlp_nw4_ab.py, profiles/r06_swin_mlp_nw4_ab.txt).  A start
    // stagger between the two (VSC_SWIN_MLP_NW4=<n> x 8 128 cycles) changes nothing: they are not in lockstep to begin with.  In the
    // encoder's two-lane step the other lane's launches were already filling those gaps: frames/s unchanged there; one-lane calls gain.
    // VSC_SWIN_MLP_NW4=-1: the 8-wave form.
    const char *nw4 = vsc_opt(OPT_SWIN_MLP_NW4);
    if (C == 128 && !(nw4 && nw4[0] == '-')) {
        constexpr int R4 = 4 * 32, smem4 = 2 * (2 * 64 * C * 2) + (4 * C + 3 * C + 3 * C) * 4;
        static bool a4[16] = {};
        if (dev >= 16 || !a4[dev]) {
            VSC_CHECK_HIP(hipFuncSetAttribute((const void *)swin_mlp_kernel<128, 4, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem4));
            if (dev < 16) a4[dev] = true;
        }
        MlpArgs b = a;
        b.stagger = nw4 ? atoi(nw4) : 0;
        hipLaunchKernelGGL((swin_mlp_kernel<128, 4, 0, true>), dim3((unsigned)((a.m + R4 - 1) / R4)), dim3(256), smem4, stream, b);
        VSC_CHECK_LAUNCH();
        return VSC_OK;
    }
    const char *sq = vsc_opt(OPT_SWIN_MLP_SEQ);   // diagnostic: 1 = the vector work of a chunk as one run (SEQ)
    if (sq && sq[0] == '1') {
        static bool seq_attr[16] = {};
        if (dev >= 16 || !seq_attr[dev]) {
            VSC_CHECK_HIP(hipFuncSetAttribute((const void *)swin_mlp_kernel<C, NW, 0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            if (dev < 16) seq_attr[dev] = true;
        }
        hipLaunchKernelGGL((swin_mlp_kernel<C, NW, 0, true, true>), dim3((unsigned)grid), dim3(NW * 64), smem, stream, a);
    } else
        hipLaunchKernelGGL((swin_mlp_kernel<C, NW, 0, true>), dim3((unsigned)grid), dim3(NW * 64), smem, stream, a);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

template <int C>
int launch_c(const MlpArgs &a, hipStream_t stream) {
    constexpr int NW = 8, R = NW * (C == 128 ? 32 : 16);
    constexpr int smem = 2 * (2 * 64 * C * 2) + (4 * C + 3 * C) * 4;
    static bool attr_set[16] = {};
    int dev = 0;
    VSC_CHECK_HIP(hipGetDevice(&dev));
    if (dev >= 16 || !attr_set[dev]) {
        VSC_CHECK_HIP(hipFuncSetAttribute((const void *)swin_mlp_kernel<C, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        if (dev < 16) attr_set[dev] = true;
    }
    const int64_t grid = (a.m + R - 1) / R;
    VSC_REQUIRE(grid < (1ll << 31), "swin_mlp: grid too large");
#ifdef VSC_MLP_ABLATION
    if (const char *e = vsc_opt(OPT_SWIN_MLP_ABL)) {
#define VSC_MLP_CASE(A) case A: { auto k = swin_mlp_kernel<C, NW, A>; VSC_CHECK_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem)); \
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NW * 64), smem, stream, a); VSC_CHECK_LAUNCH(); return VSC_OK; }
        switch (atoi(e)) { VSC_MLP_CASE(1) VSC_MLP_CASE(2) VSC_MLP_CASE(3) VSC_MLP_CASE(4) VSC_MLP_CASE(8) VSC_MLP_CASE(16) VSC_MLP_CASE(20) VSC_MLP_CASE(31) default: break; }
    }
#endif
    hipLaunchKernelGGL((swin_mlp_kernel<C, NW>), dim3((unsigned)grid), dim3(NW * 64), smem, stream, a);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

}  // namespace

bool swin_mlp_supported(int c) { return c == 128 || c == 256 || c == 512; }

// fc2.weight [c, 4c] with the hidden axis of every 32-block reordered to the kernel's contraction slots:
//     dst[n][32 S + 8 g + 4 t + i] = src[n][32 S + 16 t + 4 g + i]      (g = 0..3, t = 0, 1, i = 0..3)
// (c = 512: the chunk-major packing of swin_mlp512.hip -- same element count, same order inside a 32-block)
void swin_mlp_permute_hidden(const float *src, float *dst, int c) {
    if (c == 512) return swin_mlp512_pack_w2(src, dst);
    const int h = 4 * c;
    for (int n = 0; n < c; ++n)
        for (int k = 0; k < h; ++k) {
            const int S = k >> 5, t = (k >> 4) & 1, g = (k >> 2) & 3, i = k & 3;
            dst[(size_t)n * h + 32 * S + 8 * g + 4 * t + i] = src[(size_t)n * h + k];
        }
}

int launch_swin_mlp(const uint16_t *w1, const float *b1, const uint16_t *w2p, const float *b2, const float *gamma, const float *beta,
                    float *x, uint16_t *xb, int64_t m, int c, float eps, hipStream_t stream) {
    VSC_REQUIRE(w1 && b1 && w2p && b2 && gamma && beta && x && xb && m > 0, "swin_mlp: null/empty");
    VSC_REQUIRE(swin_mlp_supported(c), "swin_mlp: width %d unsupported (128, 256 or 512)", c);
    if (c == 512) return launch_swin_mlp512(w1, b1, w2p, b2, gamma, beta, x, xb, m, eps, stream);
    const MlpArgs a{w1, b1, w2p, b2, gamma, beta, x, xb, m, eps, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    return c == 128 ? launch_c<128>(a, stream) : launch_c<256>(a, stream);
}

bool swin_proj_mlp_supported(int c) { return c == 128 || c == 256 || c == 512; }

// proj + LayerNorm + residual + the MLP block of one Swin-V2 block in one launch (swin_mlp_kernel<..., PROJ>): x, xb updated in place
int launch_swin_proj_mlp(const uint16_t *att, const uint16_t *wp, const float *bp, const float *gamma1, const float *beta1, const uint16_t *w1,
                         const float *b1, const uint16_t *w2p, const float *b2, const float *gamma2, const float *beta2, float *x, uint16_t *xb,
                         int64_t m, int c, float eps, hipStream_t stream) {
    VSC_REQUIRE(att && wp && bp && gamma1 && beta1 && w1 && b1 && w2p && b2 && gamma2 && beta2 && x && xb && m > 0, "swin_proj_mlp: null/empty");
    VSC_REQUIRE(swin_proj_mlp_supported(c), "swin_proj_mlp: width %d unsupported (128, 256, 512)", c);
    if (c == 512) return launch_swin_proj_mlp512(att, wp, bp, gamma1, beta1, w1, b1, w2p, b2, gamma2, beta2, x, xb, m, eps, stream);
    const MlpArgs a{w1, b1, w2p, b2, gamma2, beta2, x, xb, m, eps, att, wp, bp, gamma1, beta1, 0};
    return c == 128 ? launch_proj_c<128>(a, stream) : launch_proj_c<256>(a, stream);
}

