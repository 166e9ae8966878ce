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
#include <stdlib.h>

#include <type_traits>

#include <mutex>
#include <unordered_map>

#include "common.h"
#include "gelu_poly.h"
#include "mainloop64.h"

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
    int n, k, tokens, tiles_n, tiles_m, group_n;
    int skew;  // first-round start skew in shader cycles (see phase_skew); v4: cycles per skew group
    int skew_groups;  // v4: workgroups of an XCD start in this many groups, `skew` cycles apart (launch_v4)
#ifdef VSC_GEMM_TIMING
    unsigned long long *dbg;  // [8 waves][8] cycle sums of workgroup 0 (build with -DVSC_GEMM_TIMING)
#endif
    int abl;  // diagnostic ablation bits (VSC_GEMM_ABL): 1 no loop DMA, 2 no MFMA, 8 no frag reads
    GemmExtra ex;  // LayerNorm-folding operands (common.h)
};

constexpr bool epi_bf16_out(int e) {
    return e == VSC_EPI_BF16 || e == VSC_EPI_GELU_BF16 || e == VSC_EPI_QGELU_BF16 || e == VSC_EPI_LNF_BF16 ||
           e == VSC_EPI_LNF_GELU_BF16 || e == VSC_EPI_LNF_QGELU_BF16;
}
constexpr bool epi_lnf(int e) { return e >= VSC_EPI_LNF_BF16 && e <= VSC_EPI_LNF_QGELU_BF16; }
constexpr bool epi_gelu(int e) { return e == VSC_EPI_GELU_BF16 || e == VSC_EPI_LNF_GELU_BF16; }
constexpr bool epi_qgelu(int e) { return e == VSC_EPI_QGELU_BF16 || e == VSC_EPI_LNF_QGELU_BF16; }

// sum over the 16 lanes of a DPP row (one matrix row of the fp32 write-out), result in every lane: two quad
// permutes, then the half-row and the row mirror -- four v_add_f32 with a DPP operand, no LDS round trip
__device__ __forceinline__ float row16_sum(float x) {
    auto dpp = [](float v, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    x += dpp(x, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    x += dpp(x, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    x += dpp(x, std::integral_constant<int, 0x141>{});  // row_half_mirror
    x += dpp(x, std::integral_constant<int, 0x140>{});  // row_mirror
    return x;
}

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

using vscgelu::gelu2;
using vscgelu::gelu4;
__device__ __forceinline__ float quick_gelu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -2.45546696f));  // 1.702 * log2 e
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs p) {
    lp_kernel_entry();
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
                    acc[i][j] = lp_mfma16(wf[j], af[i], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // Epilogue.  acc[i][j][r]: row m = m0 + wm*64 + i*16 + (lane & 15),
    //                          col n = n0 + wn*64 + j*16 + (lane >> 4)*4 + r.
    // The residual / position rows of all 16 fragments are requested first: `out` may alias the residual, and a load
    // issued between the stores is waited for at once (see epilogue_via_lds).
    constexpr bool HAS_AUX = EPI == VSC_EPI_RESADD_F32 || EPI == VSC_EPI_PATCH_F32;
    f32x4_t axv[HAS_AUX ? 4 : 1][4];
    int64_t orows[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t m = m0 + wm * 64 + i * 16 + fr;
        m = m < p.m ? m : p.m - 1;
        orows[i] = m;
        const float *auxrow = nullptr;
        if (EPI == VSC_EPI_PATCH_F32) {
            const int pt = p.tokens - 1;
            const int64_t f = m / pt;
            const int tok = (int)(m - f * pt) + 1;
            orows[i] = f * p.tokens + tok;
            auxrow = p.aux + (int64_t)tok * p.n;
        } else if (EPI == VSC_EPI_RESADD_F32) {
            auxrow = p.aux + m * p.n;
        }
        if (HAS_AUX) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + fq * 4;
                axv[i][j] = *(const f32x4_t *)(auxrow + (n < p.n ? n : 0));
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + wm * 64 + i * 16 + fr;
        if (m >= p.m) continue;
        const int64_t orow = orows[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + fq * 4;
            if (n >= p.n) continue;
            f32x4_t v = acc[i][j];
            if (p.bias) v += *(const f32x4_t *)(p.bias + n);
            if (EPI == VSC_EPI_BF16 || EPI == VSC_EPI_GELU_BF16 || EPI == VSC_EPI_QGELU_BF16) {
                if (EPI == VSC_EPI_GELU_BF16) {
gelu4(v);
                } else if (EPI == VSC_EPI_QGELU_BF16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
                }
                uint2 pk;
                pk.x = lp_pack2(v[0], v[1]);
                pk.y = lp_pack2(v[2], v[3]);
                *(uint2 *)((uint16_t *)p.out + orow * p.n + n) = pk;
            } else {
                if (HAS_AUX) v += axv[i][j];
                *(f32x4_t *)((float *)p.out + orow * p.n + n) = v;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// v2: 256-row tiles, 8 waves, BK = 32, STAGES-deep LDS-DMA ring with counted vmcnt.
//   * tile 256 x BN (BN = 256: waves 2x4, wave tile 128x64; BN = 128: waves 4x2, wave tile 64x64)
//     -> 128 / 85 FLOP per byte of LDS-DMA traffic instead of 64: the v1 tile was bound by
//     L2->LDS bandwidth, not by the matrix pipe.
//   * BK = 32 = one v_mfma_f32_16x16x32_bf16 k-slice; LDS rows are 64 B, 16-B chunk c of row r is
//     stored at c ^ ((-(r >> 2)) & 3): conflict-free for ds_read_b128's real lane groups
//     ({0-3,12-15,20-27}, ...) where four rows share one 256-B bank row.
//   * ring of STAGES stages; tile kt+STAGES-1 is issued right after the barrier of step kt, and
//     the wait before that barrier is `s_waitcnt vmcnt(L * (STAGES-2))` -- the newest STAGES-2
//     tiles stay in flight ACROSS the barrier (raw s_barrier: __syncthreads would drain them).
//     RAW: a wave waits for its own pieces of tile kt, the barrier covers the other waves'
//     pieces.  WAR: the stage refilled after barrier kt was last read in step kt-1, which every
//     wave finished before arriving at barrier kt.
template <int ROWS, int NW>  // rows of a 64-byte-row tile region, staged by NW waves
__device__ __forceinline__ void stage_rows32(const uint16_t *src, int64_t ld, int64_t row0,
                                             int64_t row_last, int k0, char *region, int wave,
                                             int lane) {
    constexpr int PIECES = ROWS / 16;  // 1 KiB pieces of 16 rows
#pragma unroll
    for (int j = 0; j < PIECES / NW; ++j) {
        const int piece = j * NW + wave;
        const int r = piece * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((-(r >> 2)) & 3);
        int64_t gr = row0 + r;
        gr = gr > row_last ? row_last : gr;
        const uint16_t *g = src + gr * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(region + piece * 1024), 16, 0, 0);
    }
}

// The same for Swin's PatchMerging GEMM with the 2 x 2 gather done here instead of by a copy kernel: row r of the operand is
// the merged token (b, Y, X) of a [frames, res, res, c] bf16 tensor, its columns part * c + cc the channels cc of token
// (b, 2 Y + (part & 1), 2 X + (part >> 1)) (torch2scripts.py:353-358).  res is a power of two (lh = log2(res / 2)), c a
// multiple of 32, so a 32-column slab lies inside one part: a uniform offset per K-step on top of a per-row base.
template <int ROWS, int NW>
__device__ __forceinline__ void stage_rows32_merge(const uint16_t *src, int res, int lh, int c, int64_t row0, int64_t row_last,
                                                   int k0, char *region, int wave, int lane) {
    constexpr int PIECES = ROWS / 16;
    const int part = k0 / c, cc = k0 - part * c;                       // wave-uniform
    const int64_t poff = ((int64_t)(part & 1) * res + (part >> 1)) * c + cc;
    const int hmask = (1 << lh) - 1;
#pragma unroll
    for (int j = 0; j < PIECES / NW; ++j) {
        const int piece = j * NW + wave;
        const int r = piece * 16 + (lane >> 2);
        const int ch = (lane & 3) ^ ((-(r >> 2)) & 3);
        int64_t gr = row0 + r;
        gr = gr > row_last ? row_last : gr;
        const int X = (int)gr & hmask, Y = (int)(gr >> lh) & hmask;
        const int64_t b = gr >> (2 * lh);
        const int64_t tok = (b * res + 2 * Y) * res + 2 * X;
        __builtin_amdgcn_global_load_lds((gptr_t)(src + tok * c + poff + ch * 8), (lptr_t)(region + piece * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8_t lds_frag32(const char *region, int row, int g) {
    return *(const bf16x8_t *)(region + row * 64 + ((g ^ ((-(row >> 2)) & 3)) << 4));
}

// All 256 CUs start their first tile together and every tile takes the same time, so the
// chip runs in lockstep rounds: compute, then 256 workgroups write 32 MiB at once while no
// MFMA issues.  Delaying first-round workgroup b by (b / 256) of a tile time spreads the CUs'
// phases for the rest of the launch, so one CU's write-out overlaps the others' K loops -- in
// principle; measured, the delay is never recovered (see skew_cycles), so the skew is 0 unless
// VSC_GEMM_SKEW_NS_PER_K is set.  Speed only: results never depend on it.
__device__ __forceinline__ void phase_skew(int skew_cycles) {
    if (skew_cycles > 0 && blockIdx.x < 256 && gridDim.x > 256) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const unsigned long long wait = ((unsigned long long)skew_cycles * blockIdx.x) >> 8;
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wait until at most `tiles` whole tiles (L LDS-DMA instructions each) of this wave are in flight
template <int L, int STAGES>
__device__ __forceinline__ void wait_tiles(int tiles) {
    if (STAGES >= 5 && tiles >= 3) wait_vmcnt<3 * L>();
    else if (STAGES >= 4 && tiles >= 2) wait_vmcnt<2 * L>();
    else if (tiles >= 1) wait_vmcnt<L>();
    else wait_vmcnt<0>();
}

// one lane's 4 consecutive columns of one output row
template <int EPI>
__device__ __forceinline__ void epilogue_frag(const GemmArgs &p, f32x4_t v, int64_t orow,
                                              const float *auxrow, int n) {
    if (p.bias) v += *(const f32x4_t *)(p.bias + n);
    if (EPI == VSC_EPI_BF16 || EPI == VSC_EPI_GELU_BF16 || EPI == VSC_EPI_QGELU_BF16) {
        if (EPI == VSC_EPI_GELU_BF16) {
gelu4(v);
        } else if (EPI == VSC_EPI_QGELU_BF16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
        }
        uint2 pk;
        pk.x = lp_pack2(v[0], v[1]);
        pk.y = lp_pack2(v[2], v[3]);
        *(uint2 *)((uint16_t *)p.out + orow * p.n + n) = pk;
    } else {
        if (EPI != VSC_EPI_F32) v += *(const f32x4_t *)(auxrow + n);
        *(f32x4_t *)((float *)p.out + orow * p.n + n) = v;
    }
}

// Shared LDS-staged write-out of a wave's TM x 4 fragment tile (64 columns wide); see the
// comment inside.  `lds2` must be free (no reader, no DMA in flight) and hold NW x 16 KiB.
template <int EPI, int TM, int TN>
__device__ __forceinline__ void epilogue_via_lds(const GemmArgs &p, f32x4_t (&acc)[TM][TN], char *lds2,
                                                 int wave, int lane, int wm, int wn, int64_t m0,
                                                 int n0, const float2 (&rs)[TM], const f32x4_t (&bzp)[TN],
                                                 const f32x4_t (&csp)[TN]) {
    const int fr = lane & 15, fq = lane >> 4;
    // ---- epilogue through LDS.  A lane's accumulators are 4 columns of 16 different rows
    // per fragment: stored directly that is 32-byte pieces of 16 rows per instruction, and the
    // partial-line writes cost 30-40 % of the kernel.  The ring is idle now (every wave is
    // past its last fragment read and every DMA has landed), so each wave transposes its
    // 64-column tile through a private 16 KiB region and writes whole 128-B (bf16) / 256-B
    // (fp32) row segments with 16-byte stores; the residual / position rows are read the
    // same way.  16-B chunks are XOR-swizzled by the row so both the fragment-layout
    // writes and the row-layout reads are conflict-free (<= 2-way for the 8-B bf16 writes).
    static_assert(TN == 4, "wave tile is 64 columns wide");
    char *reg = lds2 + wave * 16384;  // NW x 16 KiB <= the ring (checked in launch_v2)
    const int ncol0 = n0 + wn * 64;
    if (epi_bf16_out(EPI)) {
        float mu[TM], rstd[TM];  // LayerNorm folding: this lane's rows (loaded before the K loop)
        if (epi_lnf(EPI)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                mu[i] = rs[i].x;
                rstd[i] = rs[i].y;
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = ncol0 + j * 16 + fq * 4;
            const f32x4_t bz = bzp[j];
            (void)n;
            const f32x4_t cs = csp[j];   // LayerNorm folding: column sums of W' (zeros otherwise)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                f32x4_t v;
                if (epi_lnf(EPI)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(fmaf(-mu[i], cs[r], acc[i][j][r]), rstd[i], bz[r]);
                } else {
                    v = acc[i][j] + bz;
                }
                if (epi_gelu(EPI)) {
gelu4(v);
                } else if (epi_qgelu(EPI)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
                }
                uint2 pk;
                pk.x = lp_pack2(v[0], v[1]);
                pk.y = lp_pack2(v[2], v[3]);
                const int row = i * 16 + fr, chunk = 2 * j + (fq >> 1);
                // rows r and r+8 share (r & 7): give them opposite 8-byte halves of the chunk so
                // the 16 lanes of a ds_write_b64 group hit 16 different bank pairs
                *(uint2 *)(reg + row * 128 + ((chunk ^ (row & 7)) << 4) + ((fq ^ (row >> 3)) & 1) * 8) = pk;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int c = lane & 7;
        const int n = ncol0 + c * 8;
#pragma unroll
        for (int it = 0; it < TM * 2; ++it) {
            const int row = it * 8 + (lane >> 3);
            uint4 d = *(const uint4 *)(reg + row * 128 + ((c ^ (row & 7)) << 4));
            if (it & 1) d = make_uint4(d.z, d.w, d.x, d.y);  // (row >> 3) & 1 == it & 1: halves were swapped
            const int64_t m = m0 + wm * TM * 16 + row;
            if (m < p.m && n < p.n)
                *(uint4 *)((uint16_t *)p.out + m * p.n + n) = d;
        }
    } else {
        const int c = lane & 15;
        const int n = ncol0 + c * 4;
        f32x4_t ax[8];  // residual / position rows of the next eight write-out steps
        int orow8[8];   // PATCH: output row (frame * tokens + token) of those steps, from the same division
        auto load_aux = [&](int g) {  // g = pass * 2 + group
            const int nc = n < p.n ? n : 0;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                int64_t m = m0 + wm * TM * 16 + (g >> 1) * 64 + ((g & 1) * 8 + it) * 4 + (lane >> 4);
                m = m < p.m ? m : p.m - 1;
                const float *auxrow;
                if (EPI == VSC_EPI_PATCH_F32) {
                    const int pt = p.tokens - 1;
                    const int f = (int)m / pt, tok = (int)m - f * pt + 1;  // rows < 2^31 (checked by the launcher)
                    orow8[it] = f * p.tokens + tok;
                    auxrow = p.aux + (int64_t)tok * p.n;
                } else {
                    auxrow = p.aux + m * p.n;
                }
                ax[it] = *(const f32x4_t *)(auxrow + nc);
            }
        };
        if (EPI != VSC_EPI_F32) load_aux(0);
#pragma unroll
        for (int pass = 0; pass < TM / 4; ++pass) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nb = ncol0 + j * 16 + fq * 4;
                const f32x4_t bz = bzp[j];
                (void)nb;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int row = ii * 16 + fr, chunk = 4 * j + fq;
                    *(f32x4_t *)(reg + row * 256 + ((chunk ^ (row & 15)) << 4)) = acc[pass * 4 + ii][j] + bz;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // `out` may alias the residual (x is updated in place), so the compiler orders every residual load behind
            // the previous step's store and waits for it at once -- one exposed memory round trip per 4-row step, which
            // was most of what the residual write-out cost.  The rows of eight steps are requested together instead
            // (ax, filled by load_aux before the staging / after the previous group's stores).
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = (grp * 8 + it) * 4 + (lane >> 4);
                    f32x4_t v = *(const f32x4_t *)(reg + row * 256 + ((c ^ (row & 15)) << 4));
                    const int64_t m = m0 + wm * TM * 16 + pass * 64 + row;
                    if (EPI != VSC_EPI_F32) v += ax[it];
                    if (m < p.m && n < p.n) {
                        const int64_t orow = EPI == VSC_EPI_PATCH_F32 ? (int64_t)orow8[it] : m;
                        *(f32x4_t *)((float *)p.out + orow * p.n + n) = v;
                        if (EPI == VSC_EPI_RESADD_STATS_F32) {
                            uint2 pk;
                            pk.x = lp_pack2(v[0], v[1]);
                            pk.y = lp_pack2(v[2], v[3]);
                            *(uint2 *)(p.ex.xb + m * p.n + n) = pk;
                        }
                    }
                    if (EPI == VSC_EPI_RESADD_STATS_F32) {
                        // statistics of this row's 64-column slice (n % 64 == 0 is required, so a slice is whole or absent;
                        // rows past m are computed on stale lanes and not stored): two passes on the registers
                        const float mean = row16_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 64.0f);
                        const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                        const float m2 = row16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
                        if (c == 0 && m < p.m && ncol0 < p.n)
                            *(float2 *)(p.ex.stats + ((int64_t)(ncol0 >> 6) * p.m + m) * 2) = make_float2(mean, m2);
                    }
                }
                if (EPI != VSC_EPI_F32 && pass * 2 + grp + 1 < (TM / 4) * 2) load_aux(pass * 2 + grp + 1);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int EPI, int WAVES_M, int WAVES_N, int TM, int TN, int STAGES>
__global__ __launch_bounds__(WAVES_M *WAVES_N * 64, 2) void gemm_bf16_v2_kernel(GemmArgs p) {
    lp_kernel_entry();
    constexpr int NW = WAVES_M * WAVES_N;  // 8: one block per CU, two staggered wave groups;
                                           // 4: two independent blocks per CU (their phases
                                           //    drift apart, so one block's write-out overlaps
                                           //    the other's K loop)
    constexpr int BM2 = WAVES_M * TM * 16, BN2 = WAVES_N * TN * 16, BK2 = 32;
    constexpr int A_BYTES = BM2 * 64, W_BYTES = BN2 * 64, STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int L = (BM2 + BN2) / (16 * NW);  // LDS-DMA instructions per thread per tile
    static_assert(NW == 8 || NW == 4, "4 or 8 waves");
    static_assert(L * (STAGES - 2) < 64, "vmcnt field");
    extern __shared__ __attribute__((aligned(16))) char lds2[];
    phase_skew(p.skew);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    // group-of-G ordering along N inside the XCD's range keeps a W panel L2-resident
    int tm, tn;
    {
        const int G = p.group_n;
        const int per_group = G * p.tiles_m;
        const int grp = t / per_group;
        const int first = grp * G;
        const int width = (p.tiles_n - first) < G ? (p.tiles_n - first) : G;
        const int rem = t - grp * per_group;
        tm = rem / width;
        tn = first + (rem - tm * width);
    }
    const int64_t m0 = (int64_t)tm * BM2;
    const int n0 = tn * BN2;
    const int64_t a_last = p.m - 1, w_last = p.n - 1;

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // Two-phase, two-group schedule.  Every K-step is an L phase (issue the LDS-DMA of tile
    // kt+STAGES-1, read the fragments of tile kt into registers) and a C phase (32 MFMAs), each
    // closed by a workgroup barrier.  Waves 4-7 (the second wave of every SIMD) run one phase
    // behind waves 0-3, so on each SIMD one wave is in C while the other is in L: the matrix
    // pipe is fed while DMA issue / LDS reads proceed.
    //   RAW: a wave waits for ITS pieces of tile kt+1 at the end of L(kt); the barrier closing
    //        that phase precedes every read of tile kt+1 by either group.
    //   WAR: the stage refilled in L(kt) was last read in L(kt-1) of both groups, whose
    //        ds_reads were drained (lgkmcnt(0)) before the barrier that precedes L(kt).
    const int nk = p.k / BK2;
    const int group = NW == 8 ? wave >> 2 : 0;
    // LayerNorm folding: (mean, rstd) of this lane's TM rows, requested before the first DMA so they are in registers
    // long before the write-out needs them (in-order return: they land ahead of every counted tile)
    float2 rs[TM];
    if (epi_lnf(EPI)) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int64_t m = m0 + wm * TM * 16 + i * 16 + (lane & 15);
            m = m < p.m ? m : p.m - 1;
            rs[i] = *(const float2 *)(p.ex.rowstats + 2 * m);
        }
    }
    // the bias of this lane's TN column groups, requested before the first DMA as well (it was the first thing the
    // write-out waited for)
    f32x4_t bzp[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nb = n0 + wn * TN * 16 + j * 16 + (lane >> 4) * 4;
        bzp[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (p.bias && nb < p.n) bzp[j] = *(const f32x4_t *)(p.bias + nb);
    }
#ifdef VSC_GEMM_TIMING
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) {
        if (s < nk) {
            char *st = lds2 + s * STAGE_BYTES;
            stage_rows32<BM2, NW>(p.a, p.k, m0, a_last, s * BK2, st, wave, lane);
            stage_rows32<BN2, NW>(p.w, p.k, n0, w_last, s * BK2, st + A_BYTES, wave, lane);
        }
    }
    {
        const int issued = nk < STAGES - 1 ? nk : STAGES - 1;
        wait_tiles<L, STAGES>(issued - 1);  // tile 0 landed
    }
    __builtin_amdgcn_s_barrier();
    if (NW == 8 && group == 1) __builtin_amdgcn_s_barrier();  // stagger
    const int fr = lane & 15, fq = lane >> 4;
    int cur = 0;  // stage of tile kt
#ifdef VSC_GEMM_TIMING
    const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0};
#define VSC_T(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tsum[i] += now_ - tprev; tprev = now_; }
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#else
#define VSC_T(i)
#endif
    for (int kt = 0; kt < nk; ++kt) {
        // ---- L phase
        if (kt + STAGES - 1 < nk && !(p.abl & 1)) {
            int ns = cur + STAGES - 1;
            ns = ns >= STAGES ? ns - STAGES : ns;
            char *st = lds2 + ns * STAGE_BYTES;
            stage_rows32<BM2, NW>(p.a, p.k, m0, a_last, (kt + STAGES - 1) * BK2, st, wave, lane);
            stage_rows32<BN2, NW>(p.w, p.k, n0, w_last, (kt + STAGES - 1) * BK2, st + A_BYTES, wave, lane);
        }
        VSC_T(0)  // DMA issue
        const char *at = lds2 + cur * STAGE_BYTES;
        const char *wt = at + A_BYTES;
        bf16x8_t wf[TN], af[TM];
        if (!(p.abl & 8)) {
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = lds_frag32(wt, wn * TN * 16 + j * 16 + fr, fq);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = lds_frag32(at, wm * TM * 16 + i * 16 + fr, fq);
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = (bf16x8_t){1, 2, 3, 4, 5, 6, 7, (short)kt};
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = (bf16x8_t){1, 2, 3, 4, 5, 6, 7, (short)i};
        }
        {
            int allowed = nk - 2 - kt;  // tiles issued after tile kt+1: may stay in flight across the barrier
            allowed = allowed > STAGES - 2 ? STAGES - 2 : (allowed < 0 ? 0 : allowed);
            wait_tiles<L, STAGES>(allowed);
        }
        // pin: every fragment is live (and its ds_read retired) before the phase barrier
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(wf[j]));
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(af[i]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        VSC_T(1)  // fragment reads + landing wait
        __builtin_amdgcn_s_barrier();
        VSC_T(2)  // barrier closing L
        // ---- C phase
        __builtin_amdgcn_s_setprio(1);
        if (!(p.abl & 2)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = lp_mfma16(wf[j], af[i], acc[i][j]);
        }
        __builtin_amdgcn_s_setprio(0);
        VSC_T(3)  // MFMA issue
        if (NW == 8) __builtin_amdgcn_s_barrier();  // single group: the L-phase barrier alone orders RAW and WAR
        VSC_T(4)  // barrier closing C
        cur = cur + 1 == STAGES ? 0 : cur + 1;
    }
#ifdef VSC_GEMM_TIMING
    const unsigned long long t_epi = __builtin_amdgcn_s_memtime();
#endif
    if (NW == 8 && group == 0) __builtin_amdgcn_s_barrier();
    if (NW == 4) __builtin_amdgcn_s_barrier();  // every wave is past its last fragment read

    f32x4_t csp[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nb = n0 + wn * TN * 16 + j * 16 + (lane >> 4) * 4;
        csp[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (epi_lnf(EPI) && nb < p.n) csp[j] = *(const f32x4_t *)(p.ex.colsum + nb);
    }
    epilogue_via_lds<EPI, TM, TN>(p, acc, lds2, wave, lane, wm, wn, m0, n0, rs, bzp, csp);
#ifdef VSC_GEMM_TIMING
    if (blockIdx.x == 300 % gridDim.x && lane == 0 && p.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 5; ++i) p.dbg[wave * 8 + i] = tsum[i];
        p.dbg[wave * 8 + 5] = t_loop - t_start;
        p.dbg[wave * 8 + 6] = t_epi - t_loop;
        p.dbg[wave * 8 + 7] = t_end - t_epi;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// v3: the 256 x 256 x 64 four-phase main loop of mainloop64.h (full 128-byte line fetches, 16-MFMA segments,
// counted vmcnt that keeps four 16-KiB units in flight) under the same tile order, staggered wave groups and
// LDS-staged write-out as v2.  K % 64 == 0.
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_v3_kernel(GemmArgs p) {
    lp_kernel_entry();
    constexpr int TM = 8, TN = 4;
    extern __shared__ __attribute__((aligned(16))) char lds2[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    {
        const int G = p.group_n;
        const int per_group = G * p.tiles_m;
        const int grp = t / per_group;
        const int first = grp * G;
        const int width = (p.tiles_n - first) < G ? (p.tiles_n - first) : G;
        const int rem = t - grp * per_group;
        tm = rem / width;
        tn = first + (rem - tm * width);
    }
    const int64_t m0 = (int64_t)tm * 256;
    const int n0 = tn * 256;

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // Per-tile epilogue constants -- the bias, and for LayerNorm folding the column sums and the rows' (mean, rstd) -- go to
    // the 4 KiB of LDS behind the ring by LDS-DMA before the first operand unit is requested: they are older than every
    // counted unit, so they have landed by the first barrier, and the epilogue reads them with ds_reads.  Held in registers
    // across the K loop instead (v2) they cost 16-32 VGPRs of a 254-VGPR kernel (spilled in the folding variants), fetched
    // at the end they cost an exposed global round trip per tile.
    char *ext = lds2 + ml64::RING_BYTES;   // [0, 1 KiB) bias | [1, 2 KiB) colsum | [2, 4 KiB) rowstats of the tile's 256 rows
    {
        typedef __attribute__((address_space(3))) void *lptr_t;
        const int piece = wave & 3;
        const float *vec = wave < 4 ? p.bias : (epi_lnf(EPI) ? p.ex.colsum : nullptr);
        if (vec) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)vec, 0, p.n * 4, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(ext + (wave < 4 ? 0 : 1024) + piece * 256), 4,
                                                     (n0 + piece * 64 + lane) * 4, 0, 0, 0);
        } else {
            *(float *)(ext + (wave < 4 ? 0 : 1024) + piece * 256 + lane * 4) = 0.f;
        }
        if (epi_lnf(EPI)) {
            const int64_t rows_left = p.m - m0;
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)(p.ex.rowstats + 2 * m0), 0,
                                                                              (int)(rows_left < 256 ? rows_left : 256) * 8, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(ext + 2048 + wave * 256), 4, (wave * 64 + lane) * 4, 0, 0, 0);
        }
    }
    ml64::Ctx c;
    const int64_t a_rows = p.m - m0;
    const int w_rows = p.n - n0;
    ml64::init(c, p.a + m0 * p.k, p.k, (int)(a_rows < 256 ? a_rows : 256), p.w + (int64_t)n0 * p.k, p.k,
               w_rows < 256 ? w_rows : 256, lds2, wave, lane);
    ml64::run(c, acc, p.k / 64);
    float2 rs[TM];
    f32x4_t bzp[TN], csp[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
        rs[i] = epi_lnf(EPI) ? *(const float2 *)(ext + 2048 + (wm * 128 + i * 16 + (lane & 15)) * 8) : make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = wn * 64 + j * 16 + (lane >> 4) * 4;
        bzp[j] = *(const f32x4_t *)(ext + col * 4);
        csp[j] = *(const f32x4_t *)(ext + 1024 + col * 4);
    }
    epilogue_via_lds<EPI, TM, TN>(p, acc, lds2, wave, lane, wm, wn, m0, n0, rs, bzp, csp);
}

template <int EPI>
int launch_v3(GemmArgs p, hipStream_t stream) {
    constexpr int smem = ml64::RING_BYTES + 4096;   // ring (== 8 waves x 16 KiB of write-out staging) + the epilogue constants
    auto kern = gemm_bf16_v3_kernel<EPI>;
    static bool attr_set[16] = {};
    int dev = 0;
    VSC_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 16 && !attr_set[dev]) {
        VSC_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set[dev] = true;
    } else if (dev >= 16) {
        VSC_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    }
    p.tiles_m = (int)((p.m + 255) / 256);
    p.tiles_n = (p.n + 255) / 256;
    // N-group width (tile order): groups of 4 N-tiles whatever K is.  Swept on the ViT shapes with this loop (G = 1..12,
    // gpurun_out/gemm_ab_groups.txt -> profiles/r02_gemm_ab_groups.txt): 4 is fastest for qkv / fc1, equal for the
    // N = 768 GEMMs (3 N-tiles: one group), and the L2-capacity rule of v2 picked 1 at K = 8192 (1186 vs 1561 TF/s).
    // groups of 4 N-tiles; of 3 where that divides the row of tiles and 4 does not (qkv: 9 = 3 + 3 + 3 instead of 4 + 4 + 1:
    // 204 -> 199.5 us on the persistent kernel, tools/micro/gemm_v4_groups.py)
    int g = (p.tiles_n % 4 != 0 && p.tiles_n % 3 == 0) ? 3 : 4;
    if (const char *e = vsc_opt(OPT_GEMM_GROUP_N)) g = atoi(e);
    p.group_n = g < 1 ? 1 : (g > p.tiles_n ? p.tiles_n : g);
    p.skew = 0;
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(512), smem, stream, p);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

// ---------------------------------------------------------------------------------------------
// v4: v3's K loop in a PERSISTENT kernel.  One workgroup per CU walks the tile order with stride gridDim (same XCD ranges
// and N-groups as v3); ml64::tile_p keeps the operand ring streaming across output tiles, so what a tile boundary costs is
// the write-out alone -- not a workgroup launch, a prologue round trip, a pipeline ramp and a drain (v3: ~7 us per tile
// against 17 us of K loop at K = 768).  The write-out stages through the two ring slots that are free at a tile boundary
// (slots 6, 7: 4 KiB per wave) in passes of 32 (bf16) / 16 (fp32) rows, with the next tile's first six units in flight;
// the per-tile bias sits in a double-buffered 1-KiB row behind the ring, fetched during the previous write-out.
// Serves the plain, GELU and residual epilogues; K % 128 == 0 (an even number of K-tiles), K >= 256.
constexpr bool epi_v4(int e) {
    return e == VSC_EPI_BF16 || e == VSC_EPI_GELU_BF16 || e == VSC_EPI_QGELU_BF16 || e == VSC_EPI_RESADD_F32 || e == VSC_EPI_F32 ||
           e == VSC_EPI_LN_RES_F32 || epi_lnf(e) || e == VSC_EPI_RESADD_STATS_F32;
}

#ifdef VSC_GEMM_TIMING
static __device__ unsigned long long g_v4_t_mid[16];   // (per-wave scratch of the instrumented build; racy by design, read by the same wave)
#endif
// write-out of one wave's 128 x 64 tile through its 4 KiB of staging; bias: this tile's 256 floats in LDS
template <int EPI>
__device__ __forceinline__ void epilogue_small(const GemmArgs &p, f32x4_t (&acc)[8][4], char *reg, const char *bias_lds,
                                               int lane, int wm, int wn, int64_t m0, int n0, const char *rs_lds = nullptr) {
    const int fr = lane & 15, fq = lane >> 4;
    const int ncol0 = n0 + wn * 64;
    const int64_t mrow0 = m0 + wm * 128;
    // (bf16 write-outs keep the wave's 16 bias values in registers; the fp32 ones re-read them from LDS per row group: the residual
    //  variant then fits 232 VGPRs like the others, which leaves room on every SIMD for a light kernel of the other lane --
    //  see layernorm_light_kernel)
    f32x4_t bz[epi_bf16_out(EPI) ? 4 : 1];
    if (epi_bf16_out(EPI)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bz[j] = *(const f32x4_t *)(bias_lds + (wn * 64 + j * 16 + fq * 4) * 4);
    }
    const char *bias_lane = bias_lds + (wn * 64 + fq * 4) * 4;
    if (epi_bf16_out(EPI)) {
        // four passes of 32 rows: bias + activation + pack -> staging -> whole 128-byte row segments.  The activation of a
        // pass is VALU work between the store bursts of its neighbours (GELU: ~2 k cycles per pass), so the 128 KiB of a
        // workgroup's tile leave spread over the write-out instead of in one burst behind it.
        const int c = lane & 7;
        const int n = ncol0 + c * 8;
        // stores through a buffer descriptor over [m, n] bf16: one per-lane offset for the tile + a scalar offset per 8-row step;
        // rows past m are past the extent (dropped), lanes past n get an offset past every extent (launcher: m n 2 < 2^32)
        const uint32_t row_bytes_h = (uint32_t)p.n * 2u;
        const __amdgpu_buffer_rsrc_t outh_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(uint32_t)((uint64_t)p.m * row_bytes_h), 0x00020000);
        const uint32_t offh = n < p.n ? (uint32_t)(mrow0 + (lane >> 3)) * row_bytes_h + (uint32_t)n * 2u : 0xfffffff0u;
        // LayerNorm folding: the column sums of the lane's 16 columns (behind the bias in this tile's epilogue constants) for the
        // whole tile, (mean, rstd) of a pass's two rows at its start -- one LDS round trip per pass, not one per value
        f32x4_t cs[epi_lnf(EPI) ? 4 : 1];
        if (epi_lnf(EPI)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) cs[j] = *(const f32x4_t *)(bias_lds + 1024 + (wn * 64 + j * 16 + fq * 4) * 4);
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            float2 rs[2];
            if (epi_lnf(EPI)) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) rs[ii] = *(const float2 *)(rs_lds + (wm * 128 + (pass * 2 + ii) * 16 + fr) * 8);
            }
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4_t v;
                    if (epi_lnf(EPI)) {
                        // LayerNorm folding: v = rstd_row (acc - mean_row colsum) + bias'
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaf(fmaf(-rs[ii].x, cs[j][r], acc[pass * 2 + ii][j][r]), rs[ii].y, bz[j][r]);
                    } else {
                        v = acc[pass * 2 + ii][j] + bz[j];
                    }
                    if (epi_gelu(EPI)) gelu4(v);
                    else if (epi_qgelu(EPI)) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
                    }
                    uint2 pk;
                    pk.x = lp_pack2(v[0], v[1]);
                    pk.y = lp_pack2(v[2], v[3]);
                    const int row = ii * 16 + fr, chunk = 2 * j + (fq >> 1);
                    // rows r and r+8 share (r & 7): opposite 8-byte halves of the chunk (see epilogue_via_lds)
                    *(uint2 *)(reg + row * 128 + ((chunk ^ (row & 7)) << 4) + ((fq ^ (row >> 3)) & 1) * 8) = pk;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (pass == 0) {
                ml64::wait_vmcnt<0>();   // this wave's pieces of the next tile's units 0..5 (tile_p: `first` contract)
#ifdef VSC_GEMM_TIMING
                if (blockIdx.x == 37) g_v4_t_mid[wm * 4 + wn] = __builtin_amdgcn_s_memtime();
#endif
            }
            uint4 d[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3);
                d[it] = *(const uint4 *)(reg + row * 128 + ((c ^ (row & 7)) << 4));
                if (it & 1) d[it] = make_uint4(d[it].z, d[it].w, d[it].x, d[it].y);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const vsc_u32x4_t dv = {d[it].x, d[it].y, d[it].z, d[it].w};
                buffer_store_b128_soff(dv, outh_rsrc, offh, (uint32_t)(pass * 32 + it * 8) * row_bytes_h);
            }
        }
    } else {
        const int c = lane & 15, rq = lane >> 4;
        const int n = ncol0 + c * 4;
        // Residual rows are requested two passes ahead.  (Loads and stores share one in-order counter on gfx9 and the
        // compiler's waits count only the loads issued since, so with the stores of this read-modify-write loop in between
        // its vmcnt(n) forces more than the row it needs.  Hand-counted waits around untracked inline-asm loads were
        // measured: proj 114.6 -> 111.1 us, fc2 unchanged -- the write-out is bound by the fabric's read + write rate, not
        // by this latency -- and they need unconditional stores to keep the count exact on ragged tiles.  Not kept.)
        //
        // Addresses: ONE per-lane byte offset for the whole tile (row mrow0 + rq, column n) plus a wave-uniform scalar offset
        // per 4-row step, through buffer descriptors over the whole [m, n] fp32 matrices (launcher: m n 4 < 2^32).  Rows past
        // m lie past the descriptor's extent -- their loads return zeros and their stores are dropped by the hardware -- and
        // lanes whose columns lie past n get an offset past every extent: no compares, no 64-bit address arithmetic and no
        // per-row registers in the write-out (the residual variant: 250 -> 232 VGPRs).
        const uint32_t row_bytes = (uint32_t)p.n * 4u;
        const uint32_t extent = (uint32_t)((uint64_t)p.m * row_bytes);
        const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)extent, 0x00020000);
        const __amdgpu_buffer_rsrc_t aux_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.aux, 0, p.aux ? (int)extent : 0, 0x00020000);
        const uint32_t off0 = n < p.n ? (uint32_t)(mrow0 + rq) * row_bytes + (uint32_t)n * 4u : 0xfffffff0u;
        // RESADD_STATS (LayerNorm folding): the bf16 shadow of the new x at half the offsets, and (mean, M2) of every row's
        // 64-column slice into stats[slice][m] (lane c == 0 of the row's 16; rows past m fall outside the descriptor)
        typedef __attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int u32x2_t;
        constexpr bool STATS = EPI == VSC_EPI_RESADD_STATS_F32;
        const __amdgpu_buffer_rsrc_t xb_rsrc = __builtin_amdgcn_make_buffer_rsrc(STATS ? (void *)p.ex.xb : p.out, 0, (int)(extent >> 1), 0x00020000);
        const __amdgpu_buffer_rsrc_t st_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            STATS ? (void *)(p.ex.stats + (int64_t)(ncol0 >> 6) * p.m * 2) : p.out, 0, (int)((uint32_t)p.m * 8u), 0x00020000);
        const uint32_t st_off = (c == 0 && ncol0 < p.n) ? (uint32_t)(mrow0 + rq) * 8u : 0xfffffff0u;
        f32x4_t ax[3][4];
        auto load_aux = [&](int i, f32x4_t (&dst)[4]) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                dst[it] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, off0, (uint32_t)(i * 16 + it * 4) * row_bytes, 0));
        };
        if (EPI == VSC_EPI_F32) {
            ml64::wait_vmcnt<0>();   // plain fp32 out: no residual loads to wait for (tile_p's `first` contract)
        } else {
            load_aux(0, ax[0]);   // queued behind the next tile's DMA: waiting for these covers tile_p's `first` contract
            load_aux(1, ax[1]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (EPI != VSC_EPI_F32 && i + 2 < 8) load_aux(i + 2, ax[(i + 2) % 3]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(f32x4_t *)(reg + fr * 256 + (((4 * j + fq) ^ fr) << 4)) = acc[i][j] + *(const f32x4_t *)(bias_lane + j * 64);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 4 + rq;
                f32x4_t v = *(const f32x4_t *)(reg + row * 256 + ((c ^ row) << 4));
                if (EPI != VSC_EPI_F32) v += ax[i % 3][it];
                buffer_store_b128_soff(__builtin_bit_cast(vsc_u32x4_t, v), out_rsrc, off0, (uint32_t)(i * 16 + it * 4) * row_bytes);
                if (STATS) {
                    const u32x2_t pk = {lp_pack2(v[0], v[1]), lp_pack2(v[2], v[3])};
                    __builtin_amdgcn_raw_buffer_store_b64(pk, xb_rsrc, off0 >> 1, ((uint32_t)(i * 16 + it * 4) * row_bytes) >> 1, 0);
                    // two passes on the registers, as the one-tile kernel (epilogue_via_lds)
                    const float mean = row16_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 64.0f);
                    const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                    const float m2 = row16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
                    const u32x2_t sv = {__builtin_bit_cast(unsigned, mean), __builtin_bit_cast(unsigned, m2)};
                    __builtin_amdgcn_raw_buffer_store_b64(sv, st_rsrc, st_off, (uint32_t)(i * 16 + it * 4) * 8u, 0);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// LN_RES write-out of one 256 x 256 tile of the persistent kernel (Swin-V2 res-post-norm, torch2scripts.py:297-300, 361-362):
//   x_out = (x_in ? x_in : 0) + LayerNorm_row(acc + bias) * gamma + beta,   xb = bf16(x_out)
// Row statistics are two-pass on the registers (mean, then centred squares), combined over the four column waves through
// the two free ring slots.  With N = 512 a row's other half lives in the workgroup that runs the neighbouring tile in the
// same round (t ^ 1 -> blockIdx ^ 8: same XCD, same L2): both publish their 256 rows' (mean, M2) with L2-scope stores
// and a flag word, read the partner's and merge (Chan, equal counts).  No fence: writer and reader share the XCD's L2, the
// stores are waited for (vmcnt) before the flag, the loads bypass L1 (agent-scope atomics).  Slots are double-buffered by
// round parity: a workgroup cannot publish round i + 2 before its partner has published i + 1, i.e. finished reading i.
__device__ __forceinline__ void epilogue_ln(const GemmArgs &p, f32x4_t (&acc)[8][4], char *slots, char *reg, const char *bias_lds,
                                            int lane, int wm, int wn, int64_t m0, int n0, int iter) {
    const int fr = lane & 15, fq = lane >> 4;
    const int ncol0 = n0 + wn * 64;
    const int64_t mrow0 = m0 + wm * 128;
    ml64::wait_vmcnt<0>();   // this wave's pieces of the next tile's units 0..5 (tile_p's `first` contract)
    float *part = (float *)slots;           // [4][256] row sums of the column waves
    float *part2 = part + 4 * 256;          // [4][256] centred squares
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4_t bz = *(const f32x4_t *)(bias_lds + (wn * 64 + j * 16 + fq * 4) * 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i][j] += bz;
    }
    float mean[8], rstd[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) sm += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        if (fq == 0) part[wn * 256 + wm * 128 + i * 16 + fr] = sm;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wm * 128 + i * 16 + fr;
        mean[i] = ((part[row] + part[256 + row]) + (part[512 + row] + part[768 + row])) * (1.0f / 256.0f);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = acc[i][j][r] - mean[i];
                sq = fmaf(d, d, sq);
            }
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);
        if (fq == 0) part2[wn * 256 + row] = sq;
    }
    __syncthreads();
    const bool paired = p.tiles_n == 2;
    float m2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wm * 128 + i * 16 + fr;
        m2[i] = (part2[row] + part2[256 + row]) + (part2[512 + row] + part2[768 + row]);
    }
    if (paired) {
        float2 *mine = (float2 *)p.ex.xch + ((size_t)(iter & 1) * 256 + blockIdx.x) * 256;
        const float2 *theirs = (const float2 *)p.ex.xch + ((size_t)(iter & 1) * 256 + (blockIdx.x ^ 8)) * 256;
        if (wn == 0) {   // wave-uniform: the two waves of column 0 publish their 128 rows each
            if (fq == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    union { float2 f; unsigned long long u; } v;
                    v.f = make_float2(mean[i], m2[i]);
                    __hip_atomic_store((unsigned long long *)(mine + wm * 128 + i * 16 + fr), v.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            ml64::wait_vmcnt<0>();   // the statistics are in L2 ...
            if (lane == 0) __hip_atomic_store(p.ex.xflags + blockIdx.x * 2 + wm, p.ex.epoch + iter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before the flag
        }
        // Forward progress: the partner (blockIdx ^ 8) runs the row's other tile in the SAME round of a grid with one workgroup
        // per CU on every CU (launch_v4 checks grid == CUs and whole tile pairs; launch_gemm_ln_bf16 checks the CU count), so it
        // is resident and reaches its own publish without waiting for anybody.  Should that assumption ever break (a CU mask,
        // a partitioned device) the wait is BOUNDED: ~2^25 polls of >= 128 cycles (seconds against a tile time of 80 us -- long enough
        // for a preempted or profiler-serialised partner, which a 0.1-s bound was not), then a trap: the launch fails with a HIP error
        // at the next synchronisation instead of hanging the queue.  (A trap ends the whole HIP context; it is the last resort behind
        // the launcher's residency checks, not a flow-control path.)
        const int *flag = p.ex.xflags + (blockIdx.x ^ 8) * 2 + wm;
        int polls = 0;   // (the flag is read as a wave-uniform value, so the loop and its counter stay on the scalar side: no VGPR)
        // (flags count up across launches -- p.ex.epoch is this launch's base -- so nothing has to be zeroed between launches)
        while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - (p.ex.epoch + iter + 1) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++polls > (1 << 25)) __builtin_trap();
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            union { float2 f; unsigned long long u; } v;
            v.u = __hip_atomic_load((const unsigned long long *)(theirs + wm * 128 + i * 16 + fr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float d = mean[i] - v.f.x;
            m2[i] = (m2[i] + v.f.y) + d * d * 128.0f;      // Chan: n_a n_b / (n_a + n_b) = 128
            mean[i] = 0.5f * (mean[i] + v.f.x);
        }
    }
    // normalisation as one FMA per value at staging time (row scalars a = rstd, b = -mean rstd); gamma / beta are applied
    // after the transposition through the staging rows, where a lane owns 4 fixed columns for all passes (8 registers
    // instead of 32) -- normalising all 128 accumulators up front spilled 40 of them
    const float inv_n = paired ? 1.0f / 512.0f : 1.0f / 256.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        rstd[i] = rsqrtf(m2[i] * inv_n + p.ex.eps);
        mean[i] = -mean[i] * rstd[i];
    }
    __syncthreads();   // every wave is done with the partials: the slots become the waves' private staging
    // write-out as the residual epilogue of epilogue_small (16-row passes through 4 KiB per wave), plus the bf16 shadow
    const int c = lane & 15, rq = lane >> 4;
    const int n = ncol0 + c * 4;
    const f32x4_t gm = *(const f32x4_t *)(p.ex.gamma + n), bt = *(const f32x4_t *)(p.ex.beta + n);
    // addresses as in epilogue_small's fp32 path: one per-lane byte offset for the tile, a scalar offset per 4-row step, buffer
    // descriptors over [m, n] (rows past m: loads return zeros, stores are dropped); the bf16 shadow at half the offsets
    typedef __attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int u32x2_t;
    const uint32_t row_bytes = (uint32_t)p.n * 4u;
    const uint32_t extent = (uint32_t)((uint64_t)p.m * row_bytes);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)extent, 0x00020000);
    const __amdgpu_buffer_rsrc_t aux_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.aux, 0, p.aux ? (int)extent : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t xb_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.ex.xb, 0, (int)(extent >> 1), 0x00020000);
    const uint32_t off0 = (uint32_t)(mrow0 + rq) * row_bytes + (uint32_t)n * 4u;
    f32x4_t ax[3][4];
    auto load_aux = [&](int i, f32x4_t (&dst)[4]) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
            dst[it] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, off0, (uint32_t)(i * 16 + it * 4) * row_bytes, 0));
    };
    const bool res = p.aux != nullptr;
    if (res) {
        load_aux(0, ax[0]);
        load_aux(1, ax[1]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (res && i + 2 < 8) load_aux(i + 2, ax[(i + 2) % 3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4_t *)(reg + fr * 256 + (((4 * j + fq) ^ fr) << 4)) = acc[i][j] * rstd[i] + mean[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 4 + rq;
            f32x4_t v = *(const f32x4_t *)(reg + row * 256 + ((c ^ row) << 4));
            v = v * gm + bt;
            if (res) v += ax[i % 3][it];
            const uint32_t soff = (uint32_t)(i * 16 + it * 4) * row_bytes;
            buffer_store_b128_soff(__builtin_bit_cast(vsc_u32x4_t, v), out_rsrc, off0, soff);
            const u32x2_t pk = {lp_pack2(v[0], v[1]), lp_pack2(v[2], v[3])};
            __builtin_amdgcn_raw_buffer_store_b64(pk, xb_rsrc, off0 >> 1, soff >> 1, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_v4_kernel(GemmArgs p) {
    lp_kernel_entry();
    typedef __attribute__((address_space(3))) void *lptr_t;
    extern __shared__ __attribute__((aligned(16))) char lds2[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int total = p.tiles_m * p.tiles_n;
    const int nk = p.k / 64;
    auto tile_of = [&](int vt, int &tm, int &tn) {
        const int t = xcd_remap(vt, total);
        const int G = p.group_n;
        const int per_group = G * p.tiles_m;
        const int grp = t / per_group;
        const int first = grp * G;
        const int width = (p.tiles_n - first) < G ? (p.tiles_n - first) : G;
        const int rem = t - grp * per_group;
        tm = rem / width;
        tn = first + (rem - tm * width);
    };
    // epilogue constants behind the ring: the bias of this tile / of the next one (2 x 1 KiB); with LayerNorm folding 2 x (bias |
    // column sums), then (mean, rstd) of the current tile's 256 rows (2 KiB) and, when the kernel merges the statistics itself,
    // the [nslices][256] (mean, M2) partials of those rows (2 KiB per slice)
    constexpr int EXT = epi_lnf(EPI) ? 2048 : 1024;
    char *ext = lds2 + ml64::RING_BYTES;
    char *rs_lds = ext + 2 * EXT, *sl_lds = rs_lds + 2048;
    const int nsl = epi_lnf(EPI) ? p.ex.nslices : 0;   // > 0: merge in the kernel
    const __amdgpu_buffer_rsrc_t bias_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.bias, 0, p.bias ? p.n * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t cs_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(epi_lnf(EPI) ? p.ex.colsum : p.bias), 0, epi_lnf(EPI) ? p.n * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(epi_lnf(EPI) ? (nsl ? p.ex.slices : p.ex.rowstats) : p.bias), 0, epi_lnf(EPI) ? (int)((uint32_t)p.m * 8u * (uint32_t)(nsl ? nsl : 1)) : 0, 0x00020000);
    auto stage_bias = [&](int buf, int n0) {   // waves 0-3, one 256-byte piece each (zeros without a bias / past n)
        if (wave < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(bias_rsrc, (lptr_t)(ext + buf * EXT + wave * 256), 4,
                                                     (n0 + wave * 64 + lane) * 4, 0, 0, 0);
        else if (epi_lnf(EPI))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(cs_rsrc, (lptr_t)(ext + buf * EXT + 1024 + (wave - 4) * 256), 4,
                                                     (n0 + (wave - 4) * 64 + lane) * 4, 0, 0, 0);
    };
    // row statistics of the tile about to be multiplied (single buffer: its previous readers are behind the tile-end barrier):
    // (mean, rstd) of rows m0 .. m0 + 255, or their nslices partials (1-KiB pieces of 128 rows; rows past m read zeros or the next
    // slice's rows -- their outputs are never stored)
    auto stage_rows = [&](int64_t m0) {
        if (!epi_lnf(EPI)) return;
        if (nsl == 0) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_rsrc, (lptr_t)(rs_lds + wave * 256), 4, (uint32_t)m0 * 8u + (wave * 64 + lane) * 4, 0, 0, 0);
        } else {
            for (int q = wave; q < 2 * nsl; q += 8)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_rsrc, (lptr_t)(sl_lds + q * 1024), 16,
                                                         (uint32_t)(((int64_t)(q >> 1) * p.m + m0 + (q & 1) * 128) * 8) + lane * 16, 0, 0, 0);
        }
    };
    int vt = blockIdx.x, tm, tn;
    tile_of(vt, tm, tn);
    stage_bias(0, tn * 256);
    stage_rows((int64_t)tm * 256);
    ml64::Ctx c;
    ml64::init_whole(c, p.a, p.k, p.m, p.w, p.k, p.n, (int64_t)tm * 256, (int64_t)tn * 256, lds2, wave, lane);
    // Start skew.  All workgroups of a persistent launch run in lockstep (same start, same tile times), so the 32 CUs of an
    // XCD would hit their write-outs together: 4 MiB of stores against the XCD's L2 / fabric write rate (~13 B/clk per CU
    // when everyone stores: tools/micro/atomic_add_bw) -- measured 5-7 k cycles per plain bf16 tile where the CU's own
    // store path needs 2 k.  The workgroups of an XCD therefore start in `skew_groups` groups `skew` cycles apart (slot =
    // blockIdx / 8 is the workgroup's index inside its XCD) and keep that distance to the end, so only one group is in its
    // write-out at a time.  The delay sits between the issue of the first operand units and their first use.
    const int skew_wait = p.skew * ((int)(blockIdx.x >> 3) % p.skew_groups);
    const unsigned long long t_launch = __builtin_amdgcn_s_memtime();
    ml64::prologue(c, nk);
    if (skew_wait > 0)
        while (__builtin_amdgcn_s_memtime() - t_launch < (unsigned long long)skew_wait) __builtin_amdgcn_s_sleep(4);
    ml64::Frags f;
    char *reg = lds2 + 6 * ml64::UNIT_BYTES + wave * 4096;
#ifdef VSC_GEMM_TIMING
    const bool rec = blockIdx.x == 37 && lane == 0 && (wave == 0 || wave == 4) && p.dbg;
    unsigned long long *tb = p.dbg + (wave >> 2) * 32;
    if (rec) tb[0] = __builtin_amdgcn_s_memtime();
#endif
    for (int i = 0;; ++i) {
        const int vnext = vt + (int)gridDim.x;
        const bool last = vnext >= total;
        int tm2 = tm, tn2 = tn;
        if (!last) tile_of(vnext, tm2, tn2);
        const uint32_t da = (uint32_t)((int64_t)(tm2 - tm) * 256 * p.k * 2), dw = (uint32_t)((int64_t)(tn2 - tn) * 256 * p.k * 2);
        f32x4_t acc[8][4];
#pragma unroll
        for (int ii = 0; ii < 8; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[ii][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                asm volatile("" : "+v"(acc[ii][j]));   // real zeros in real registers: folded into the first MFMAs' C operand, the
                                                       // first K-tile gets a register assignment of its own and the kernel spills
            }
        if (!last) stage_bias((i + 1) & 1, tn2 * 256);
        if (i > 0) stage_rows((int64_t)tm * 256);   // a tile ahead: landed long before the write-out that reads it
#ifdef VSC_GEMM_TIMING
        if (rec && i < 6) tb[1 + i * 5] = __builtin_amdgcn_s_memtime();
#endif
        ml64::tile_p(c, acc, f, nk, i == 0, da, dw);
#ifdef VSC_GEMM_TIMING
        if (rec && i < 6) tb[2 + i * 5] = __builtin_amdgcn_s_memtime();
#endif
        // the write-out's per-lane addresses hang off a lane id the compiler cannot see through: computed here, per tile,
        // instead of hoisted out of the tile loop and carried (spilled) across the K loop
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        if constexpr (EPI == VSC_EPI_LN_RES_F32)
            epilogue_ln(p, acc, lds2 + 6 * ml64::UNIT_BYTES, reg, ext + (i & 1) * EXT, lane_e, wm, wn, (int64_t)tm * 256, tn * 256, i);
        else
        {
            if (epi_lnf(EPI) && nsl) {
                // (mean, M2) of the row's 64-column slices -> (mean, rstd) of the row (Chan, equal counts), one row per thread of waves
                // 0-3; every wave's pieces have landed (counted waits + barriers of the K loop behind them)
                if (tid < 256) {
                    float mean = 0.f, m2 = 0.f;
                    for (int sl = 0; sl < nsl; ++sl) mean += *(const float *)(sl_lds + sl * 2048 + tid * 8);
                    mean /= (float)nsl;
                    for (int sl = 0; sl < nsl; ++sl) {
                        const float2 v = *(const float2 *)(sl_lds + sl * 2048 + tid * 8);
                        const float d = v.x - mean;
                        m2 += v.y + 64.0f * d * d;
                    }
                    *(float2 *)(rs_lds + tid * 8) = make_float2(mean, rsqrtf(m2 / (float)p.k + p.ex.eps));
                }
                __syncthreads();
            }
            epilogue_small<EPI>(p, acc, reg, ext + (i & 1) * EXT, lane_e, wm, wn, (int64_t)tm * 256, tn * 256, rs_lds);
        }
#ifdef VSC_GEMM_TIMING
        if (rec && i < 6) {
            tb[3 + i * 5] = g_v4_t_mid[wave];
            tb[4 + i * 5] = __builtin_amdgcn_s_memtime();
        }
#endif
        if (last) break;
        __builtin_amdgcn_s_barrier();   // every wave is done with slots 6, 7 and has waited for its DMA pieces
        __builtin_amdgcn_sched_barrier(0);
#ifdef VSC_GEMM_TIMING
        if (rec && i < 6) tb[5 + i * 5] = __builtin_amdgcn_s_memtime();
#endif
        vt = vnext;
        tm = tm2;
        tn = tn2;
    }
}

template <int EPI>
int launch_v4(GemmArgs p, int cus, hipStream_t stream) {
    constexpr int smem_max = ml64::RING_BYTES + (epi_lnf(EPI) ? 2 * 2048 + 2048 + 12 * 2048 : 2048);
    const int smem = ml64::RING_BYTES + (epi_lnf(EPI) ? 2 * 2048 + 2048 + p.ex.nslices * 2048 : 2048);
    auto kern = gemm_bf16_v4_kernel<EPI>;
    static bool attr_set[16] = {};
    int dev = 0;
    VSC_CHECK_HIP(hipGetDevice(&dev));
    if (dev >= 16 || !attr_set[dev]) {
        VSC_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem_max));
        if (dev < 16) attr_set[dev] = true;
    }
    // groups of 4 N-tiles; of 3 where that divides the row of tiles and 4 does not (qkv: 9 = 3 + 3 + 3 instead of 4 + 4 + 1:
    // 204 -> 199.5 us on the persistent kernel, tools/micro/gemm_v4_groups.py)
    int g = (p.tiles_n % 4 != 0 && p.tiles_n % 3 == 0) ? 3 : 4;
    if (const char *e = vsc_opt(OPT_GEMM_GROUP_N)) g = atoi(e);
    p.group_n = g < 1 ? 1 : (g > p.tiles_n ? p.tiles_n : g);
    p.skew = 0;
    p.skew_groups = 1;
    constexpr bool pair_exchange = EPI == VSC_EPI_LN_RES_F32;
    if (pair_exchange) {
        // the row's two tiles (t, t ^ 1) must land on workgroups (b, b ^ 8) of the same round: N-groups spanning the whole row
        // of tiles, no diagnostic regrouping, one workgroup on every CU, whole pairs per round
        p.group_n = p.tiles_n;
        int dev_cus = 0, dev = 0;
        VSC_CHECK_HIP(hipGetDevice(&dev));
        VSC_CHECK_HIP(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, dev));
        // (grid < CUs -- the VSC_GEMM_V4_GRID diagnostic -- keeps every workgroup resident as well; the exchange slots are sized for 256)
        VSC_REQUIRE(cus <= dev_cus && cus <= 256 && cus % 16 == 0, "gemm LN_RES: the pair exchange needs every workgroup resident, in whole pairs per XCD (grid %d, device %d CUs)", cus, dev_cus);
        VSC_REQUIRE(p.tiles_n == 1 || (p.tiles_n == 2 && ((int64_t)p.tiles_m * p.tiles_n) % 16 == 0),
                    "gemm LN_RES: %d x %d tiles do not form whole pairs per XCD round", p.tiles_m, p.tiles_n);
    }
    if (const char *e = pair_exchange ? nullptr : vsc_opt(OPT_GEMM_V4_SKEW)) {   // "cycles,groups" (diagnostic sweep)
        int cyc = 0, grp = 1;
        if (sscanf(e, "%d,%d", &cyc, &grp) == 2 && cyc >= 0 && grp >= 1) {
            p.skew = cyc;
            p.skew_groups = grp;
        }
    }
#ifdef VSC_GEMM_TIMING
    static unsigned long long *dbg = nullptr;
    if (!dbg) VSC_CHECK_HIP(hipMalloc(&dbg, 64 * 8));
    VSC_CHECK_HIP(hipMemsetAsync(dbg, 0, 64 * 8, stream));
    p.dbg = dbg;
#endif
    hipLaunchKernelGGL(kern, dim3(cus), dim3(512), smem, stream, p);
    VSC_CHECK_LAUNCH();
#ifdef VSC_GEMM_TIMING
    if (vsc_opt(OPT_GEMM_TIMING_PRINT)) {
        unsigned long long h[64];
        VSC_CHECK_HIP(hipStreamSynchronize(stream));
        VSC_CHECK_HIP(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
        for (int g = 0; g < 2; ++g) {
            const unsigned long long *t = h + g * 32;
            fprintf(stderr, "v4 timing m=%lld n=%d k=%d epi=%d wave %d (10 ns ticks from kernel start):", (long long)p.m, p.n, p.k, EPI, g * 4);
            for (int i = 0; i < 6 && t[1 + i * 5]; ++i)
                fprintf(stderr, "  | tile %d: K loop %llu..%llu, DMA wait until %llu, write-out until %llu, barrier %llu", i, t[1 + i * 5] - t[0],
                        t[2 + i * 5] - t[0], t[3 + i * 5] - t[0], t[4 + i * 5] - t[0], t[5 + i * 5] ? t[5 + i * 5] - t[0] : 0ull);
            fprintf(stderr, "\n");
        }
    }
#endif
    return VSC_OK;
}

// v3 or its persistent form: v4 wherever a workgroup gets more than one tile and the 32-bit source offsets hold
template <int EPI>
int launch_v34(GemmArgs p, hipStream_t stream) {
    p.tiles_m = (int)((p.m + 255) / 256);
    p.tiles_n = (p.n + 255) / 256;
    if constexpr (epi_v4(EPI)) {
        const char *v4e = vsc_opt(OPT_GEMM_V4);   // diagnostic A/B switch, read per launch
        const bool off = v4e && v4e[0] == '0';
        static int cus_of[16] = {};
        int dev = 0;
        VSC_CHECK_HIP(hipGetDevice(&dev));
        int cus = dev < 16 ? cus_of[dev] : 0;
        if (!cus) {
            VSC_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            if (dev < 16) cus_of[dev] = cus;
        }
        const int64_t a_span = (int64_t)p.tiles_m * 256 * p.k * 2, w_span = (int64_t)p.tiles_n * 256 * p.k * 2;
        // (K > 3072: the per-tile costs v4 removes are < 1 % of a tile and its lockstep costs ~3 % -- 8192^3 680 vs 701 us)
        // fp32 write-outs address [m, n] by 32-bit byte offsets from the matrix origin, rows of the last (ragged) tile included
        const bool out_span_ok = (int64_t)p.tiles_m * 256 * p.n * (epi_bf16_out(EPI) ? 2 : 4) < (1ll << 32);
        if (!off && p.k % 128 == 0 && p.k >= 128 && p.k <= 3072 && cus % 8 == 0 && (int64_t)p.tiles_m * p.tiles_n > cus &&
            a_span < (1ll << 32) && w_span < (1ll << 32) && out_span_ok && (EPI != VSC_EPI_RESADD_F32 || p.aux) &&
            (!epi_lnf(EPI) || p.m * 8 < (1ll << 32)))
        {
            int grid = cus;
            // diagnostic: persistent workgroups per launch (a multiple of 8).  Measured with two lanes, so that the two chunks' GEMMs run
            // side by side on half the chip each instead of one after the other: 128 -> 23.4 k, 192 -> 23.4 k, 256 -> 23.9 k frames/s
            if (const char *e = vsc_opt(OPT_GEMM_V4_GRID)) {
                const int g = atoi(e);
                if (g >= 8 && g <= cus && g % 8 == 0) grid = g;
            }
            if constexpr (epi_lnf(EPI)) {
                // statistics given as slice partials: merged per tile inside the kernel when they fit behind the ring (<= 12 slices)
                if (p.ex.slices && p.ex.nslices > 12) {
                    int rc = launch_ln_stats_merge(p.ex.slices, const_cast<float *>(p.ex.rowstats), p.m, p.ex.nslices, p.k, p.ex.eps, stream);
                    if (rc) return rc;
                    p.ex.nslices = 0;
                } else if (!p.ex.slices) {
                    p.ex.nslices = 0;
                }
            }
            return launch_v4<EPI>(p, grid, stream);
        }
    }
    if constexpr (epi_lnf(EPI)) {
        if (p.ex.slices) {
            int rc = launch_ln_stats_merge(p.ex.slices, const_cast<float *>(p.ex.rowstats), p.m, p.ex.nslices, p.k, p.ex.eps, stream);
            if (rc) return rc;
        }
        p.ex.nslices = 0;
    }
    return launch_v3<EPI>(p, stream);
}

// first-round start skew (diagnostic, off by default): VSC_GEMM_SKEW_NS_PER_K * K / 10 shader cycles spread over the
// first 256 workgroups.  Re-measured in the ViT step with skews from 0.6 us to a whole tile time: 0 is as fast as any
// (19.7 k frames/s), a tile time costs 6 % -- the start-up delay is never recovered.
inline int skew_cycles(int k) {
    const char *e = vsc_opt(OPT_GEMM_SKEW_NS_PER_K);
    const int ns_per_k = e ? atoi(e) : 0;
    return (int)((int64_t)ns_per_k * k / 10);
}

template <int EPI, int WAVES_M, int WAVES_N, int TM, int TN, int STAGES>
int launch_v2(GemmArgs p, hipStream_t stream) {
    constexpr int BM2 = WAVES_M * TM * 16, BN2 = WAVES_N * TN * 16;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int ring = STAGES * (BM2 + BN2) * 64;
    constexpr int smem = ring > NW * 16384 ? ring : NW * 16384;
    auto kern = gemm_bf16_v2_kernel<EPI, WAVES_M, WAVES_N, TM, TN, STAGES>;
    static bool attr_set[16] = {};   // per device
    int dev = 0;
    VSC_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        VSC_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    p.tiles_m = (int)((p.m + BM2 - 1) / BM2);
    p.tiles_n = (p.n + BN2 - 1) / BN2;
    // W panel of one N-group (G tiles x K) should sit in a 4 MiB XCD L2 next to the A panels: every N-group is one
    // more pass over A.  With few N tiles the whole width is one group whatever K is -- fc2 (N = 768, K = 3072) ran
    // 3 passes over its 402 MB A (larger than the Infinity Cache) under the L2 rule: 395 -> 330 us with one.
    int g = (int)((int64_t)(3 << 19) / ((int64_t)BN2 * p.k * 2));
    if (p.tiles_n <= 4) g = p.tiles_n;
    if (const char *e = vsc_opt(OPT_GEMM_GROUP_N)) g = atoi(e);
    p.group_n = g < 1 ? 1 : (g > p.tiles_n ? p.tiles_n : g);
    p.skew = skew_cycles(p.k);
#ifdef VSC_GEMM_TIMING
    static unsigned long long *dbg = nullptr;
    if (!dbg) VSC_CHECK_HIP(hipMalloc(&dbg, 64 * 8));
    VSC_CHECK_HIP(hipMemsetAsync(dbg, 0, 64 * 8, stream));
    p.dbg = dbg;
#endif
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(NW * 64), smem, stream, p);
    VSC_CHECK_LAUNCH();
#ifdef VSC_GEMM_TIMING
    if (vsc_opt(OPT_GEMM_TIMING_PRINT)) {
        unsigned long long h[64];
        VSC_CHECK_HIP(hipStreamSynchronize(stream));
        VSC_CHECK_HIP(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
        const double nk = p.k / 32.0;
        for (int w = 0; w < NW; ++w)
            fprintf(stderr, "timing m=%lld n=%d k=%d wave %d: per K-step cycles  dma-issue %.0f  reads+landing %.0f  barrier-L %.0f  mfma-issue %.0f  barrier-C %.0f\n",
                    (long long)p.m, p.n, p.k, w, h[w * 8] / nk, h[w * 8 + 1] / nk, h[w * 8 + 2] / nk, h[w * 8 + 3] / nk, h[w * 8 + 4] / nk);
        fprintf(stderr, "timing tile (wave 0): prologue %llu  K loop %llu  write-out incl. store drain %llu ticks\n", h[5], h[6], h[7]);
    }
#endif
    return VSC_OK;
}



template <int EPI>
int launch_v2_pick(const GemmArgs &p, hipStream_t stream) {
    // wide N: 256x256 tile; N <= 1024 (proj / fc2 / patch): 256x128 so the grid still fills 256 CUs
    // A: 8 waves 256x256 (1 block/CU)   B: 8 waves 256x128
    // C: 4 waves 128x256 (2 blocks/CU)  D: 4 waves 256x128 (2 blocks/CU)
    // Tried and dropped: 64-k stages with full 128-B line fetches (+0.6 %),
    // a 5-deep ring (+0.4 %), register staging instead of LDS-DMA (0.47x, spills), issuing the
    // LDS-DMA between the MFMA rows of the C phase (-1 %), hybrid staging with A by LDS-DMA and W
    // through registers (-3 %), an intra-wave software pipeline (A fragments refilled from the next stage right
    // after their MFMA group, W fragments double-buffered, one barrier per K-step: equal with DMA, 0.79x with DMA
    // off -- anything issued between back-to-back MFMAs costs more than the phase barriers it removes), a
    // persistent workgroup per CU with a 3-stage ring and the write-out staged behind it so the next tile's first
    // stages fly during the write-out (+-1 %: launch, fill and drain were already hidden; 3 stages are as fast
    // as 4).  tools/micro/fill_bench.hip: LDS-DMA alone delivers 21.6 B/clk/CU with 64-B row pieces (34-40 with
    // full 128-B lines), VGPR staging no more; the K loop is issue/phase-bound, not delivery-bound.
    const char *force = vsc_opt(OPT_GEMM_CFG);   // diagnostic, read per launch
    // measured: A wins on every ViT shape (K >= 768).  With K <= 512 (Swin) the K loop is only 4-16 stages long and
    // the epilogue is a large share of a tile: the 4-wave tiles run two workgroups per CU, so one's write-out
    // overlaps the other's K loop (stage-3 qkv 765 -> 875 TF/s); D when N is a multiple of 128 but not of 256.
    char cfg = p.n > 128 ? 'A' : 'B';
    if (p.k <= 512 && p.n > 128) {
        cfg = (p.n % 256 != 0 && p.n % 128 == 0) ? 'D' : 'C';
        // ... unless the persistent kernel takes it (v4: K = 256 / 384 / 512, more 256 x 256 tiles than CUs): without the per-tile
        // launch / prologue / drain the big tile wins here too (Swin-V2-B batch 256: 12.15 k -> 12.95 k frames/s)
        if (epi_v4(EPI) && p.k % 128 == 0 && p.k >= 128 && p.n % 256 == 0 && ((p.m + 255) / 256) * (p.n / 256) > 256) cfg = 'A';
    }
    if (force) cfg = force[0];
    const char *v3e = vsc_opt(OPT_GEMM_V3);   // diagnostic A/B switch, read per launch
    const bool no_v3 = v3e && v3e[0] == '0';
    if (cfg == 'A' && p.k % 64 == 0 && !no_v3) return launch_v34<EPI>(p, stream);
    switch (cfg) {
        case 'A': return launch_v2<EPI, 2, 4, 8, 4, 4>(p, stream);
        case 'B': return launch_v2<EPI, 4, 2, 4, 4, 4>(p, stream);
        case 'C': return launch_v2<EPI, 1, 4, 8, 4, 3>(p, stream);
        default: return launch_v2<EPI, 2, 2, 8, 4, 3>(p, stream);
    }
}

template <int EPI>
int launch_t(const GemmArgs &p, int tiles_m, hipStream_t stream) {
    hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(tiles_m * p.tiles_n), dim3(256), 0, stream, p);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}


// ---------------------------------------------------------------------------------------------
// gemm_ln: out rows are OWNED by one workgroup (BN == N), so the res-post-norm update of Swin-V2
//   x_out = (x_in ? x_in : 0) + LayerNorm(A W^T + bias) * gamma + beta,   xb_out = bf16(x_out)
// happens in the epilogue: no fp32 round trip of the GEMM result and no separate LayerNorm pass
// (torch2scripts.py:297-300 `shortcut + norm1(attn)`, `x + norm2(mlp(x))`; :361-362 reduction+norm).
// Same ring / two-phase two-group schedule as v2; wave tile 64 x 128 (4 x 8 fragments), block
// WAVES_M*64 x WAVES_N*128: 512x128 (N=128), 256x256 (N=256), 128x512 (N=512).
// Row statistics: per-lane partial sums -> xor-shuffles over the 4 lanes of a row -> per-wave
// partials in LDS -> combined across the WAVES_N waves; two passes (mean, centred squares).
struct GemmLnArgs {
    const uint16_t *a, *w;
    const float *bias, *gamma, *beta, *x_in;
    float *x_out;
    uint16_t *xb_out;
    int64_t m;
    int n, k;
    float eps;
    int g_res = 0, g_lh = 0, g_c = 0;   // PatchMerging gather on the A operand (stage_rows32_merge); g_res = 0: plain rows
};

#ifdef VSC_GEMM_TIMING
__device__ unsigned long long g_ln_dbg[8];
#endif

template <int WAVES_M, int WAVES_N, int STAGES>
__global__ __launch_bounds__(512, 2) void gemm_ln_kernel(GemmLnArgs p) {
    lp_kernel_entry();
#ifdef VSC_GEMM_TIMING
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
    constexpr int TM = 4, TN = 8, NW = 8, BK2 = 32;
    constexpr int BM2 = WAVES_M * 64, BN2 = WAVES_N * 128;
    constexpr int A_BYTES = BM2 * 64, W_BYTES = BN2 * 64, STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int L = (BM2 + BN2) / (16 * NW);
    constexpr int OUT_BYTES = 8 * 16384;  // write-out regions; the row-statistic partials sit behind them
    static_assert(WAVES_M * WAVES_N == 8, "8 waves");
    static_assert(STAGES * STAGE_BYTES <= 160 * 1024, "ring fits the CU's LDS");  // launch_ln_t allocates max(ring, write-out)
    extern __shared__ __attribute__((aligned(16))) char lds2[];
    // (a first-round start skew as in gemm_bf16_v2_kernel was measured here: no gain at 0.3 tile times, a loss beyond)
    float *part = (float *)(lds2 + OUT_BYTES);             // [WAVES_N][BM2] row sums
    float *part2 = part + WAVES_N * BM2;                   // [WAVES_N][BM2] centred squares

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int64_t m0 = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * BM2;
    const int64_t a_last = p.m - 1, w_last = p.n - 1;

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = p.k / BK2;
    const int group = wave >> 2;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) {
        if (s < nk) {
            char *st = lds2 + s * STAGE_BYTES;
            if (p.g_res) stage_rows32_merge<BM2, NW>(p.a, p.g_res, p.g_lh, p.g_c, m0, a_last, s * BK2, st, wave, lane);
            else stage_rows32<BM2, NW>(p.a, p.k, m0, a_last, s * BK2, st, wave, lane);
            stage_rows32<BN2, NW>(p.w, p.k, 0, w_last, s * BK2, st + A_BYTES, wave, lane);
        }
    }
    {
        const int issued = nk < STAGES - 1 ? nk : STAGES - 1;
        wait_tiles<L, STAGES>(issued - 1);
    }
    __builtin_amdgcn_s_barrier();
    if (group == 1) __builtin_amdgcn_s_barrier();  // stagger
    const int fr = lane & 15, fq = lane >> 4;
    int cur = 0;
#ifdef VSC_GEMM_TIMING
    const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + STAGES - 1 < nk) {
            int ns = cur + STAGES - 1;
            ns = ns >= STAGES ? ns - STAGES : ns;
            char *st = lds2 + ns * STAGE_BYTES;
            if (p.g_res) stage_rows32_merge<BM2, NW>(p.a, p.g_res, p.g_lh, p.g_c, m0, a_last, (kt + STAGES - 1) * BK2, st, wave, lane);
            else stage_rows32<BM2, NW>(p.a, p.k, m0, a_last, (kt + STAGES - 1) * BK2, st, wave, lane);
            stage_rows32<BN2, NW>(p.w, p.k, 0, w_last, (kt + STAGES - 1) * BK2, st + A_BYTES, wave, lane);
        }
        const char *at = lds2 + cur * STAGE_BYTES;
        const char *wt = at + A_BYTES;
        bf16x8_t wf[TN], af[TM];
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[j] = lds_frag32(wt, wn * 128 + j * 16 + fr, fq);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = lds_frag32(at, wm * 64 + i * 16 + fr, fq);
        {
            int allowed = nk - 2 - kt;
            allowed = allowed > STAGES - 2 ? STAGES - 2 : (allowed < 0 ? 0 : allowed);
            wait_tiles<L, STAGES>(allowed);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(wf[j]));
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(af[i]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = lp_mfma16(wf[j], af[i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        cur = cur + 1 == STAGES ? 0 : cur + 1;
    }
    if (group == 0) __builtin_amdgcn_s_barrier();
#ifdef VSC_GEMM_TIMING
    const unsigned long long t_epi = __builtin_amdgcn_s_memtime();
#endif

    // ---- bias, then row statistics over the full N = BN2 columns
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = wn * 128 + j * 16 + fq * 4;
        if (p.bias) {
            const f32x4_t bz = *(const f32x4_t *)(p.bias + n);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] += bz;
        }
    }
    float mean[TM], rstd[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) sm += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        if (fq == 0) part[wn * BM2 + wm * 64 + i * 16 + fr] = sm;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float sm = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_N; ++w) sm += part[w * BM2 + wm * 64 + i * 16 + fr];
        mean[i] = sm / (float)BN2;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = acc[i][j][r] - mean[i];
                sq += d * d;
            }
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);
        if (fq == 0) part2[wn * BM2 + wm * 64 + i * 16 + fr] = sq;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float sq = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_N; ++w) sq += part2[w * BM2 + wm * 64 + i * 16 + fr];
        rstd[i] = rsqrtf(sq / (float)BN2 + p.eps);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = wn * 128 + j * 16 + fq * 4;
        const f32x4_t gm = *(const f32x4_t *)(p.gamma + n), bt = *(const f32x4_t *)(p.beta + n);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = (acc[i][j][r] - mean[i]) * rstd[i] * gm[r] + bt[r];
    }
#ifdef VSC_GEMM_TIMING
    const unsigned long long t_norm = __builtin_amdgcn_s_memtime();
#endif
    // ---- write-out through LDS, one 64-column half at a time (64 rows x 64 cols fp32 = 16 KiB per wave)
    f32x4_t xin[8];
    auto load_xin = [&](int grp) {  // group = (column half, row half): 8 write-out steps of 4 rows
        const int nn = wn * 128 + (grp >> 1) * 64 + (lane & 15) * 4;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            int64_t m = m0 + wm * 64 + ((grp & 1) * 8 + it) * 4 + (lane >> 4);
            m = m < p.m ? m : p.m - 1;
            xin[it] = *(const f32x4_t *)(p.x_in + m * p.n + nn);
        }
    };
    if (p.x_in) load_xin(0);
    char *reg = lds2 + wave * 16384;
    const int c = lane & 15;
#pragma unroll
    for (int hj = 0; hj < 2; ++hj) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = i * 16 + fr, chunk = 4 * jj + fq;
                *(f32x4_t *)(reg + row * 256 + ((chunk ^ (row & 15)) << 4)) = acc[i][hj * 4 + jj];
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int n = wn * 128 + hj * 64 + c * 4;
        // x_out may alias x_in, so the compiler keeps every residual load behind the previous step's stores and waits
        // for it at once: one exposed HBM round trip per 4-row step, 65 % of a K = 512 tile's cycles.  The rows of a
        // group of eight steps are therefore requested together (they are distinct from anything stored so far).
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = (rh * 8 + it) * 4 + (lane >> 4);
                f32x4_t v = *(const f32x4_t *)(reg + row * 256 + ((c ^ (row & 15)) << 4));
                const int64_t m = m0 + wm * 64 + row;
                if (p.x_in) v += xin[it];
                if (m < p.m) {
                    *(f32x4_t *)(p.x_out + m * p.n + n) = v;
                    uint2 pk;
                    pk.x = lp_pack2(v[0], v[1]);
                    pk.y = lp_pack2(v[2], v[3]);
                    *(uint2 *)(p.xb_out + m * p.n + n) = pk;
                }
            }
            if (p.x_in && hj * 2 + rh < 3) load_xin(hj * 2 + rh + 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#ifdef VSC_GEMM_TIMING
    if (blockIdx.x == 300 % gridDim.x && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        g_ln_dbg[0] = t_loop - t_start;
        g_ln_dbg[1] = t_epi - t_loop;
        g_ln_dbg[2] = t_end - t_epi;
        g_ln_dbg[3] = t_norm - t_epi;
    }
#endif
}

template <int WAVES_M, int WAVES_N, int STAGES>
int launch_ln_t(const GemmLnArgs &p, hipStream_t stream) {
    // the write-out regions + row-statistic partials alias the ring once the K loop is done
    constexpr int ring = STAGES * (WAVES_M * 64 + WAVES_N * 128) * 64;
    constexpr int smem = ring > 8 * 16384 + 8192 ? ring : 8 * 16384 + 8192;
    auto kern = gemm_ln_kernel<WAVES_M, WAVES_N, STAGES>;
    static bool attr_set[16] = {};   // per device
    int dev = 0;
    VSC_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        VSC_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    const int64_t blocks = (p.m + WAVES_M * 64 - 1) / (WAVES_M * 64);
    VSC_REQUIRE(blocks < (1ll << 31), "gemm_ln: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), smem, stream, p);
    VSC_CHECK_LAUNCH();
#ifdef VSC_GEMM_TIMING
    if (vsc_opt(OPT_GEMM_TIMING_PRINT)) {
        unsigned long long h[8];
        VSC_CHECK_HIP(hipStreamSynchronize(stream));
        VSC_CHECK_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ln_dbg), sizeof(h)));
        fprintf(stderr, "timing gemm_ln m=%lld n=%d k=%d (%d K-steps): prologue %llu  K loop %llu (%.0f per step)  LN + write-out %llu ticks (statistics + normalise %llu)\n",
                (long long)p.m, p.n, p.k, p.k / 32, h[0], h[1], (double)h[1] / (p.k / 32), h[2], h[3]);
    }
#endif
    return VSC_OK;
}

}  // namespace

int launch_gemm_bf16(const uint16_t *a, const uint16_t *w, const float *bias, const float *aux,
                     void *out, int64_t m, int n, int k, int epilogue, int tokens,
                     hipStream_t stream) {
    VSC_REQUIRE(epilogue >= VSC_EPI_BF16 && epilogue <= VSC_EPI_F32, "gemm: unknown epilogue %d", epilogue);
    return launch_gemm_bf16_ex(a, w, bias, aux, out, m, n, k, epilogue, tokens, GemmExtra{}, stream);
}

int launch_gemm_bf16_ex(const uint16_t *a, const uint16_t *w, const float *bias, const float *aux, void *out, int64_t m,
                        int n, int k, int epilogue, int tokens, const GemmExtra &ex, hipStream_t stream) {
    VSC_REQUIRE(a && w && out, "gemm: null operand");
    VSC_REQUIRE(m > 0 && n > 0 && k > 0, "gemm: empty problem m=%lld n=%d k=%d", (long long)m, n, k);
    VSC_REQUIRE(k % BK == 0, "gemm: K=%d must be a multiple of %d", k, BK);
    VSC_REQUIRE(n % 4 == 0, "gemm: N=%d must be a multiple of 4", n);
    const int64_t tiles_m = (m + BM - 1) / BM;
    const int tiles_n = (n + BN - 1) / BN;
    VSC_REQUIRE(tiles_m * tiles_n < (1ll << 31), "gemm: grid too large");
    GemmArgs p{a, w, bias, aux, out, m, n, k, tokens, tiles_n, (int)tiles_m, 1, 0};
    p.abl = 0;
    p.ex = ex;
    if (const char *e = vsc_opt(OPT_GEMM_ABL)) p.abl = atoi(e);
    const bool force_v1 = vsc_opt(OPT_GEMM_V1) != nullptr;
    const bool force_v2 = vsc_opt(OPT_GEMM_CFG) != nullptr;
    // A launch that cannot put a 256-row tile on at least half the CUs runs the 128 x 128 kernel instead (four times
    // the workgroups, two per CU): at 8 frames (M = 1576) fc2 takes 44 instead of 74 us and proj 16 instead of 27,
    // at 32 frames the N = 768 GEMMs 26 / 57 instead of 35 / 77 us; from ~130 tiles up the big tile wins.
    const int64_t big_tiles = ((m + 255) / 256) * ((n + 255) / 256);
    const bool fills = big_tiles > (k <= 512 ? 64 : 128);
    const bool fused = epilogue >= VSC_EPI_LNF_BF16;  // LayerNorm-folding kinds exist in the 256-row kernels only
    if (fused) {
        VSC_REQUIRE(n % 64 == 0 && k % 32 == 0, "gemm: LayerNorm-folding epilogue needs N %% 64 == 0 (N=%d)", n);
        if (epilogue == VSC_EPI_RESADD_STATS_F32)
            VSC_REQUIRE(aux && ex.xb && ex.stats, "gemm: RESADD_STATS needs residual, xb and stats");
        else
            VSC_REQUIRE(ex.rowstats && ex.colsum && bias && (!ex.slices || (ex.nslices * 64 == k && ex.eps > 0.f)),
                        "gemm: LNF needs rowstats, colsum and the folded bias (and K / 64 slices + eps when given partials)");
    }
    const bool v2 = fused || (!force_v1 && m >= 1024 && k % 32 == 0 && n % 8 == 0 && (fills || force_v2));
    if (epilogue == VSC_EPI_RESADD_F32) VSC_REQUIRE(aux, "gemm: RESADD needs the residual pointer");
    if (epilogue == VSC_EPI_PATCH_F32) {
        VSC_REQUIRE(aux && tokens > 1, "gemm: PATCH needs pos and tokens");
        VSC_REQUIRE(m % (tokens - 1) == 0, "gemm: PATCH rows %lld not a multiple of %d patches",
                    (long long)m, tokens - 1);
        VSC_REQUIRE(m / (tokens - 1) * tokens < (1ll << 31), "gemm: PATCH output rows exceed 2^31");
    }
    if (v2) {
        switch (epilogue) {
            case VSC_EPI_BF16: return launch_v2_pick<VSC_EPI_BF16>(p, stream);
            case VSC_EPI_GELU_BF16: return launch_v2_pick<VSC_EPI_GELU_BF16>(p, stream);
            case VSC_EPI_QGELU_BF16: return launch_v2_pick<VSC_EPI_QGELU_BF16>(p, stream);
            case VSC_EPI_RESADD_F32: return launch_v2_pick<VSC_EPI_RESADD_F32>(p, stream);
            case VSC_EPI_PATCH_F32: return launch_v2_pick<VSC_EPI_PATCH_F32>(p, stream);
            case VSC_EPI_F32: return launch_v2_pick<VSC_EPI_F32>(p, stream);
            case VSC_EPI_LNF_BF16: return p.k % 64 == 0 ? launch_v34<VSC_EPI_LNF_BF16>(p, stream) : launch_v2<VSC_EPI_LNF_BF16, 2, 4, 8, 4, 4>(p, stream);
            case VSC_EPI_LNF_GELU_BF16: return p.k % 64 == 0 ? launch_v34<VSC_EPI_LNF_GELU_BF16>(p, stream) : launch_v2<VSC_EPI_LNF_GELU_BF16, 2, 4, 8, 4, 4>(p, stream);
            case VSC_EPI_LNF_QGELU_BF16: return p.k % 64 == 0 ? launch_v34<VSC_EPI_LNF_QGELU_BF16>(p, stream) : launch_v2<VSC_EPI_LNF_QGELU_BF16, 2, 4, 8, 4, 4>(p, stream);
            case VSC_EPI_RESADD_STATS_F32: return p.k % 64 == 0 ? launch_v34<VSC_EPI_RESADD_STATS_F32>(p, stream) : launch_v2<VSC_EPI_RESADD_STATS_F32, 2, 4, 8, 4, 4>(p, stream);
            default: VSC_REQUIRE(false, "gemm: unknown epilogue %d", epilogue);
        }
    }
    switch (epilogue) {
        case VSC_EPI_BF16: return launch_t<VSC_EPI_BF16>(p, (int)tiles_m, stream);
        case VSC_EPI_GELU_BF16: return launch_t<VSC_EPI_GELU_BF16>(p, (int)tiles_m, stream);
        case VSC_EPI_QGELU_BF16: return launch_t<VSC_EPI_QGELU_BF16>(p, (int)tiles_m, stream);
        case VSC_EPI_RESADD_F32: return launch_t<VSC_EPI_RESADD_F32>(p, (int)tiles_m, stream);
        case VSC_EPI_PATCH_F32: return launch_t<VSC_EPI_PATCH_F32>(p, (int)tiles_m, stream);
        case VSC_EPI_F32: return launch_t<VSC_EPI_F32>(p, (int)tiles_m, stream);
        default: VSC_REQUIRE(false, "gemm: unknown epilogue %d", epilogue);
    }
    return VSC_OK;
}

// (mean, M2) of the width / 64 column slices of every row -> (mean, rstd) of the row (Chan's pairwise update with
// equal counts): what the LayerNorm-folding GEMM epilogues read
__global__ __launch_bounds__(256) void ln_stats_merge_kernel(const float *__restrict__ stats, float *__restrict__ rowstats,
                                                             int64_t rows, int slices, int width, float eps) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= rows) return;
    float mean = 0.f, m2 = 0.f;
    for (int s = 0; s < slices; ++s) mean += stats[((int64_t)s * rows + m) * 2];
    mean /= (float)slices;
    for (int s = 0; s < slices; ++s) {
        const float2 v = *(const float2 *)(stats + ((int64_t)s * rows + m) * 2);
        const float d = v.x - mean;
        m2 += v.y + 64.0f * d * d;
    }
    *(float2 *)(rowstats + 2 * m) = make_float2(mean, rsqrtf(m2 / (float)width + eps));
}

int launch_ln_stats_merge(const float *stats, float *rowstats, int64_t rows, int slices, int width, float eps,
                          hipStream_t stream) {
    VSC_REQUIRE(stats && rowstats && rows > 0 && slices * 64 == width, "ln_stats_merge: %d slices for width %d", slices, width);
    hipLaunchKernelGGL(ln_stats_merge_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, stats, rowstats, rows,
                       slices, width, eps);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

// pair-exchange workspaces seen so far -> the value their flags have reached (see launch_gemm_ln_bf16)
static std::mutex g_ln_ws_mutex;
static std::unordered_map<const void *, int> g_ln_ws_epoch;
void gemm_ln_workspace_forget(const void *ws) {   // before the owner frees it: a later allocation at this address starts from zeroed flags
    std::lock_guard<std::mutex> lock(g_ln_ws_mutex);
    g_ln_ws_epoch.erase(ws);
}

int launch_gemm_ln_bf16(const uint16_t *a, const uint16_t *w, const float *bias, const float *gamma,
                        const float *beta, const float *x_in, float *x_out, uint16_t *xb_out, int64_t m, int n,
                        int k, float eps, hipStream_t stream, void *pair_ws, int merge_res, int merge_c) {
    VSC_REQUIRE(a && w && gamma && beta && x_out && xb_out, "gemm_ln: null operand");
    // merge_res > 0: a is the [frames, merge_res, merge_res, merge_c] token tensor, the operand its 2 x 2 PatchMerging gather
    VSC_REQUIRE(merge_res == 0 || (merge_res >= 2 && (merge_res & (merge_res - 1)) == 0 && merge_c % 32 == 0 && k == 4 * merge_c &&
                                   a != xb_out && m % ((int64_t)(merge_res / 2) * (merge_res / 2)) == 0),
                "gemm_ln: merge gather needs a power-of-two map (%d), channels %% 32 (%d), k = 4 c and an output outside the input", merge_res, merge_c);
    VSC_REQUIRE(m > 0 && k > 0 && k % 32 == 0, "gemm_ln: m=%lld k=%d (k must be a multiple of 32)", (long long)m, k);
    // Widths 256 / 512 with more 256 x 256 tiles than CUs: the persistent kernel (v4 K loop, ring streaming across tiles)
    // with the LN_RES write-out -- the row-owning 128 x 512 tile below stages 25 % more operand bytes per MFMA on the
    // BK = 32 loop and pays a prologue per tile.  N = 512 needs whole tile pairs per XCD range (tiles_m % 8 == 0).
    {
        const int64_t tiles_m = (m + 255) / 256;
        const int tiles_n = n / 256;
        const char *opt = vsc_opt(OPT_GEMM_LN_V4);
        int dev = 0, cus = 0;
        VSC_CHECK_HIP(hipGetDevice(&dev));
        static int cus_of[16] = {};
        if (dev >= 0 && dev < 16 && cus_of[dev]) cus = cus_of[dev];
        else {
            VSC_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            if (dev >= 0 && dev < 16) cus_of[dev] = cus;
        }
        const bool shape_ok = (n == 256 || (n == 512 && tiles_m % 8 == 0)) && k % 128 == 0 && k >= 128 && k <= 3072 && cus == 256 &&
                              tiles_m * tiles_n > cus && tiles_m * 256 * k * 2 < (1ll << 32) && tiles_m * 256 * n * 4 < (1ll << 32) &&
                              dev >= 0 && dev < 16;
        // Where it pays (tools/micro/gemm_ln_ab.py, 256 Swin-V2-B frames; round 4, with the write-out on buffer operations and no
        // spill left): N = 512 from K = 512 on -- s2 proj 96 vs 100 us, s2 fc2 166 vs 195, the un-gathered merge shape 114 vs
        // 128; at N = 256 the short-K launch ties (s1 proj 168 vs 168: bound by its bytes whichever kernel runs it) and stays
        // on the row-owning tile.  VSC_GEMM_LN_V4=1 forces the persistent kernel on every shape it supports (tests), 0
        // switches it off.
        const bool pays = n == 512 && k >= 512;
        if (shape_ok && merge_res == 0 && !(opt && opt[0] == '0') && (pays || (opt && opt[0] == '1'))) {
            static void *dev_ws[16] = {};   // callers without a workspace of their own: one call at a time per device
            if (!pair_ws) {
                if (!dev_ws[dev]) VSC_CHECK_HIP(hipMalloc(&dev_ws[dev], VSC_GEMM_LN_WS_BYTES));
                pair_ws = dev_ws[dev];
            }
            GemmArgs p{a, w, bias, x_in, x_out, m, n, k, 0, tiles_n, (int)tiles_m, 1, 0};
            p.abl = 0;
            p.ex.xb = xb_out;
            p.ex.gamma = gamma;
            p.ex.beta = beta;
            p.ex.eps = eps;
            p.ex.xch = (float *)pair_ws;
            p.ex.xflags = (int *)((char *)pair_ws + 2 * 256 * 256 * 2 * 4);
            int grid = cus;
            if (const char *e = vsc_opt(OPT_GEMM_V4_GRID)) {   // diagnostic, as in launch_v34
                const int g = atoi(e);
                if (g >= 16 && g <= cus && g % 16 == 0 && tiles_m * tiles_n >= g) grid = g;
            }
            if (tiles_n == 2) {
                // The epoch lives on the host and is baked into the kernel arguments: a captured launch replayed from a HIP graph would
                // wait for flag values of the capture, not of the replay -- refuse stream capture (common.h: the workspace contract).
                hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
                VSC_CHECK_HIP(hipStreamIsCapturing(stream, &cap));
                VSC_REQUIRE(cap == hipStreamCaptureStatusNone, "gemm_ln (N = 512 pair exchange): not capturable into a HIP graph (per-launch epoch)");
                // the flags of a workspace count up from launch to launch (zeroed once, when the workspace is first seen): no memset
                // node between the GEMMs of a block.  One launch advances them by its tiles per workgroup.
                std::lock_guard<std::mutex> lock(g_ln_ws_mutex);
                auto it = g_ln_ws_epoch.find(pair_ws);
                if (it == g_ln_ws_epoch.end()) {
                    VSC_CHECK_HIP(hipMemsetAsync(p.ex.xflags, 0, 256 * 2 * 4, stream));
                    it = g_ln_ws_epoch.emplace(pair_ws, 0).first;
                }
                p.ex.epoch = it->second;
                const int per_wg = (int)((tiles_m * tiles_n + grid - 1) / grid);
                it->second = (it->second + per_wg + 1) & 0x3fffffff;
                if (it->second < p.ex.epoch) {   // wrapped (after ~10^8 launches): start over from zeroed flags
                    VSC_CHECK_HIP(hipMemsetAsync(p.ex.xflags, 0, 256 * 2 * 4, stream));
                    p.ex.epoch = 0;
                    it->second = per_wg + 1;
                }
            }
            return launch_v4<VSC_EPI_LN_RES_F32>(p, grid, stream);
        }
    }
    GemmLnArgs p{a, w, bias, gamma, beta, x_in, x_out, xb_out, m, n, k, eps};
    if (merge_res) {
        p.g_res = merge_res;
        p.g_c = merge_c;
        for (int half = merge_res / 2; half > 1; half >>= 1) ++p.g_lh;
    }
    switch (n) {
        case 128: return launch_ln_t<8, 1, 4>(p, stream);  // 4 x 40 KiB = the whole 160 KiB: two tiles stay in flight across a barrier
        case 256: return launch_ln_t<4, 2, 4>(p, stream);
        case 512: return launch_ln_t<2, 4, 4>(p, stream);
        default: VSC_REQUIRE(false, "gemm_ln: N=%d unsupported (row-owning tiles exist for 128, 256, 512)", n);
    }
    return VSC_OK;
}

bool gemm_ln_supported(int n, int k) { return (n == 128 || n == 256 || n == 512) && k % 32 == 0; }
