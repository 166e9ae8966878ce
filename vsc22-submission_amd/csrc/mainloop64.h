// 256 x 256 x 64 bf16 MFMA main loop shared by the encoder GEMM (gemm_bf16.hip, v3) and the similarity
// pre-filter sweep (knn.hip): acc[8][4] (wave tile 128 x 64, 16x16x32 MFMAs) += A[256, K] . W[256, K]^T.
//
// Eight waves (2 along M x 4 along N), one workgroup per CU.  K advances 64 per tile (full 128-byte lines
// from L2, where the BK = 32 loop of v2 fetched every line in two halves), and a K-tile is worked off in
// FOUR phases of 16 MFMAs -- one quadrant (64 x 32) of the wave tile each:
//
//     phase  fragment reads (ds_read_b128)              LDS-DMA issued (2 x 1 KiB per wave)   MFMAs
//       0    A-sub0 (8)                                 unit (t+1, W-h1)                      A0 x W0
//       1    W-sub1 (4)                                 unit (t+1, A-h1)                      A0 x W1
//       2    A-sub1 (8)                                 unit (t+2, W-h0)                      A1 x W1
//       3    W-sub0 of tile t+1 (4), a phase early      unit (t+2, A-h0)                      A1 x W0
// (W-sub0 is kept in registers from phase 0 to phase 3; reading the next tile's W-sub0 in phase 3 into a second register
//  set balances the L segments at 8 / 4 / 8 / 4 reads where 12 / 4 / 8 / 0 left phase 0 the longest: 8192^3 1561 -> 1609
//  TF/s, the K = 768 shapes within noise.)
//
// Every phase is an L segment (reads + DMA issue + counted vmcnt) and a C segment (16 MFMAs under s_setprio),
// each closed by a raw s_barrier; waves 4-7 (the M-half wm = 1, the second wave of every SIMD) run one
// segment behind waves 0-3, so on each SIMD one wave computes while the other loads.
//
// LDS: a ring of 8 units of 16 KiB (128 rows x 128 B); a K-tile is 4 units in first-use order
//     j = 0: W-h0 (columns wn*64 + 0..31 of every wn)      j = 1: A-h0 (rows wm*128 + 0..63 of every wm)
//     j = 2: W-h1 (columns wn*64 + 32..63)                 j = 3: A-h1 (rows wm*128 + 64..127)
// unit u = 4 t + j sits in slot u & 7 and is issued SIX units ahead of the phase that first needs it:
//   RAW  unit u is first read in phase >= u - 1; the L segment of phase g ends with a vmcnt that leaves only the
//        four newest units (8 instructions) of this wave in flight, i.e. units <= g + 2 have landed, and the
//        barrier(s) that follow cover the other waves' pieces (one barrier more for the staggered group);
//   WAR  phase g overwrites the slot of unit g - 2, last read in phase <= g - 2 by either group: at least two
//        barriers lie between those reads' lgkmcnt wait (ahead of their MFMAs) and the DMA issue.
// 16-byte chunk c of row r is stored at chunk c ^ ((r >> 1) & 7): conflict-free for ds_read_b128's lane groups
// on 128-byte rows; LDS-DMA writes lane-linear, so the permutation is applied to the per-lane SOURCE address
// and again on the read.
#pragma once
#include "common.h"

namespace ml64 {
#ifndef NOWAIT_TEST
#define NOWAIT_TEST true
#endif

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef __attribute__((address_space(3))) const char *ldsc_t;                 // 32-bit LDS address
typedef __attribute__((address_space(3))) const bf16x8_t *ldsfrag_t;

constexpr int UNIT_BYTES = 16384;
constexpr int RING_BYTES = 8 * UNIT_BYTES;   // 128 KiB

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct Ctx {
    __amdgpu_buffer_rsrc_t a_rsrc;   // A tile: base = row origin of the tile, extent = its existing rows (reads past it return 0)
    __amdgpu_buffer_rsrc_t w_rsrc;
    uint32_t a_off[2][2];  // [half][piece]: this lane's source byte offset inside the tile (k = 0)
    uint32_t w_off[2][2];
    char *lds;             // ring base
    uint32_t rd_a;         // LDS byte offset of this lane's A fragment chunk (kh = 0) inside its unit: wm * 8192 + lane part
    uint32_t rd_w;         // same for W: wn * 4096 + lane part
    // fragment read bases [A / W][slot >> 2][kh]: ring base + 64 KiB * (slot >> 2) + (rd ^ 64 kh).  Every fragment address is one
    // of these eight registers plus a compile-time offset below 64 KiB ((slot & 3) * 16 KiB + f * 2 KiB), i.e. the ds_read's
    // immediate: formed as ring + slot * 16 KiB + ((rd + 2048 f) ^ 64 kh) the slot part did not fit the 16-bit immediate and
    // cost a v_add_u32 per fragment pair -- 17 vector instructions per K-tile and wave, and the vector pipe's time adds to the
    // matrix pipe's (tools/micro/pipe_overlap.hip).  (rd < 16 KiB - 6 KiB and bit 6 of rd belongs to the chunk index alone, so
    // (rd + 2048 f) ^ 64 = (rd ^ 64) + 2048 f.)
    ldsc_t fr[2][2][2];
    int wave;
    // K-tile index t -> source offsets.  One tile pair (GEMM): A and W both advance 128 bytes per K-tile.  Stream
    // (similarity sweep): t = (W tile index << kt_shift) | k-tile inside it; A restarts with every W tile, W moves on by
    // w_tile_stride bytes (256 rows).
    int kt_shift = 31;
    uint32_t kt_mask = 0x7fffffffu;
    uint32_t w_tile_stride = 0;
    // ... or, for a K-tile count per row that is not a power of two (kt_inv != 0): W tile index = t / kt_n = (t * kt_inv) >> 32 with
    // kt_inv = ceil(2^32 / kt_n) (exact for t < 2^32 / kt_n), k-tile = t - that * kt_n.  Scalar arithmetic (t is wave-uniform).
    uint32_t kt_inv = 0, kt_n = 1;
    bool kt_general = false;   // set from a template parameter by the caller: the pow2 form must not carry the other form's scalar work in its K loop
};

// a_rows / w_rows: rows of the tile that exist (>= 1).  Rows past the edge are outside the buffer descriptor's
// extent: the LDS-DMA delivers zeros for them and the write-out never stores them.
__device__ __forceinline__ void init(Ctx &c, const uint16_t *a_tile, int64_t lda, int a_rows, const uint16_t *w_tile,
                                     int64_t ldw, int w_rows, char *lds, int wave, int lane) {
    c.a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a_tile, 0, (int)(a_rows * lda * 2), 0x00020000);
    c.w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)w_tile, 0, (int)(w_rows * ldw * 2), 0x00020000);
    c.lds = lds;
    c.wave = wave;
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int piece = wave + 8 * jj;
            const int lr = piece * 8 + (lane >> 3);                 // row inside the unit
            const int ch = (lane & 7) ^ ((lr >> 1) & 7);            // source chunk for LDS chunk (lane & 7)
            const int ar = (lr >> 6) * 128 + half * 64 + (lr & 63);
            const int wr = (lr >> 5) * 64 + half * 32 + (lr & 31);
            c.a_off[half][jj] = (uint32_t)(ar * (int)lda * 2 + ch * 16);
            c.w_off[half][jj] = (uint32_t)(wr * (int)ldw * 2 + ch * 16);
        }
    const int wm = wave >> 2, wn = wave & 3;
    const uint32_t lane_part = (uint32_t)((lane & 15) * 128 + (((lane >> 4) ^ ((lane >> 1) & 7)) << 4));
    c.rd_a = wm * 8192 + lane_part;
    c.rd_w = wn * 4096 + lane_part;
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            // opaque values: the compiler otherwise keeps four of the eight and re-derives the other parity's by a v_add_u32 per use
            ldsc_t pa = (ldsc_t)(lds + par * 4 * UNIT_BYTES + (c.rd_a ^ (uint32_t)(kh * 64)));
            ldsc_t pw = (ldsc_t)(lds + par * 4 * UNIT_BYTES + (c.rd_w ^ (uint32_t)(kh * 64)));
            asm volatile("" : "+v"(pa), "+v"(pw));
            c.fr[0][par][kh] = pa;
            c.fr[1][par][kh] = pw;
        }
}

template <int KIND, int HALF>   // KIND 0 = A, 1 = W
__device__ __forceinline__ void stage_unit(const Ctx &c, int slot, int ktile) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const uint32_t off = KIND == 0 ? c.a_off[HALF][jj] : c.w_off[HALF][jj];
        uint32_t soff;
        if (c.kt_general) {
            const uint32_t wt = __builtin_amdgcn_readfirstlane((uint32_t)(((uint64_t)(uint32_t)ktile * c.kt_inv) >> 32));
            soff = ((uint32_t)ktile - wt * c.kt_n) * 128u + (KIND == 0 ? 0u : wt * c.w_tile_stride);
        } else {
            soff = ((uint32_t)ktile & c.kt_mask) * 128u + (KIND == 0 ? 0u : ((uint32_t)ktile >> c.kt_shift) * c.w_tile_stride);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(KIND == 0 ? c.a_rsrc : c.w_rsrc,
                                                 (lptr_t)(c.lds + slot * UNIT_BYTES + (c.wave + 8 * jj) * 1024), 16, off,
                                                 soff, 0, 0);
    }
}

template <int KIND>   // 0 = A, 1 = W
__device__ __forceinline__ bf16x8_t frag(const Ctx &c, int slot, int f, int kh) {
    return *(ldsfrag_t)(c.fr[KIND][slot >> 2][kh] + (slot & 3) * UNIT_BYTES + f * 2048);
}

// fragments: af[i][kh] (i = 0..3: the A sub-tile in use), w1[j][kh] (W sub-tile 1 of the K-tile at hand) and, by tile parity,
// w0[B][j][kh]: W sub-tile 0 of the tile at hand lives in w0[t & 1] -- it is read one phase EARLY, in the otherwise read-free
// phase 3 of the tile before, into the other parity's registers (fragment reads per phase 8 / 4 / 8 / 4 instead of 12 / 4 / 8 / 0)
struct Frags {
    bf16x8_t af[4][2];
    bf16x8_t w1[2][2];
    bf16x8_t w0[2][2][2];
};

template <int B, int ASUB, int WSUB>
__device__ __forceinline__ void mfma_quadrant(f32x4_t (&acc)[8][4], const Frags &f) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[ASUB * 4 + i][WSUB * 2 + j] = lp_mfma16(WSUB == 0 ? f.w0[B][j][kh] : f.w1[j][kh], f.af[i][kh], acc[ASUB * 4 + i][WSUB * 2 + j]);
    __builtin_amdgcn_s_setprio(0);
}

// closes an L segment: counted wait for this wave's DMA pieces (steady state: the four newest units stay in flight;
// the last two tiles drain: `tail_vm` instructions may stay), then the workgroup barrier
template <bool STEADY, int TAIL1, int TAIL2>
__device__ __forceinline__ void end_l(int rem, bool nowait = false) {
    if (STEADY) {
        if (!nowait) wait_vmcnt<8>();
    } else if (rem > 2) wait_vmcnt<8>();
    else if (rem == 2) wait_vmcnt<TAIL1>();
    else wait_vmcnt<TAIL2>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void end_c() {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// One K-tile (four phases).  B = slot parity of the tile; kt1 / kt2: K-tile indices (source offsets) of the units staged
// for the next tile (j = 2, 3) and the one after (j = 0, 1) -- t + 1 and t + 2 inside one operand pair, something else where
// a persistent sequence moves on to its next output tile; rem = K-tiles left including this one; STEADY: rem > 2 is known (no
// tail branches in the body); nowait (STEADY only): every unit this tile and the next one read has landed already (the first
// K-tile behind a write-out), so the L segments close without their vmcnt.
template <int B, bool STEADY>
__device__ __forceinline__ void ktile_g(const Ctx &c, f32x4_t (&acc)[8][4], Frags &f, int kt1, int kt2, int rem, bool nowait) {
    constexpr int S_W0 = 4 * B + 0, S_A0 = 4 * B + 1, S_W1 = 4 * B + 2, S_A1 = 4 * B + 3;   // this tile's slots
    constexpr int N_W0 = 4 * (B ^ 1) + 0, N_W1 = 4 * (B ^ 1) + 2, N_A1 = 4 * (B ^ 1) + 3;  // tile t+1, j = 0, 2, 3
    (void)S_W0;
    // ---- phase 0 (W sub-tile 0 is already in f.w0[B]: read in phase 3 of the previous tile, or by tiles() for tile 0)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) f.af[i][kh] = frag<0>(c, S_A0, i, kh);
    if (STEADY || rem > 1) stage_unit<1, 1>(c, N_W1, kt1);
    end_l<STEADY, 8, 2>(rem, nowait);
    mfma_quadrant<B, 0, 0>(acc, f);
    end_c();
    // ---- phase 1
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) f.w1[j][kh] = frag<1>(c, S_W1, j, kh);
    if (STEADY || rem > 1) stage_unit<0, 1>(c, N_A1, kt1);
    end_l<STEADY, 8, 0>(rem, nowait);
    mfma_quadrant<B, 0, 1>(acc, f);
    end_c();
    // ---- phase 2
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) f.af[i][kh] = frag<0>(c, S_A1, i, kh);
    if (STEADY || rem > 2) stage_unit<1, 0>(c, S_W0, kt2);
    end_l<STEADY, 6, 0>(rem, nowait);
    mfma_quadrant<B, 1, 1>(acc, f);
    end_c();
    // ---- phase 3: unit (t+1, W-h0) landed with the wait of phase 2 (units <= g + 2) and the barriers since
    if (STEADY || rem > 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) f.w0[B ^ 1][j][kh] = frag<1>(c, N_W0, j, kh);
    }
    if (STEADY || rem > 2) stage_unit<0, 0>(c, S_A0, kt2);
    end_l<STEADY, 4, 0>(rem, nowait);
    mfma_quadrant<B, 1, 0>(acc, f);
    end_c();
}
template <int B, bool STEADY>
__device__ __forceinline__ void ktile(const Ctx &c, f32x4_t (&acc)[8][4], Frags &f, int t, int rem) {
    ktile_g<B, STEADY>(c, acc, f, t + 1, t + 2, rem, false);
}

// staging of units 0..5 (tile 0 whole, tile 1 j = 0, 1), wait for the first two, workgroup barrier
__device__ __forceinline__ void prologue(const Ctx &c, int nk) {
    stage_unit<1, 0>(c, 0, 0);
    stage_unit<0, 0>(c, 1, 0);
    stage_unit<1, 1>(c, 2, 0);
    stage_unit<0, 1>(c, 3, 0);
    if (nk > 1) {
        stage_unit<1, 0>(c, 4, 1);
        stage_unit<0, 0>(c, 5, 1);
        wait_vmcnt<8>();   // units 0, 1 landed
    } else {
        wait_vmcnt<4>();
    }
    __builtin_amdgcn_s_barrier();
}

// K-tiles [t0, t0 + n) of a sequence of `total` (t0 even).  Opens with the stagger barrier of waves 4-7 and closes with
// the matching barrier of waves 0-3, so on return every wave is past its last fragment read of these tiles; LDS-DMA
// for the tiles behind them (if any) stays in flight.
__device__ __forceinline__ void tiles(const Ctx &c, f32x4_t (&acc)[8][4], Frags &f, int t0, int n, int total) {
    const int group = c.wave >> 2;
    if (group == 1) __builtin_amdgcn_s_barrier();   // stagger: waves 4-7 run one segment behind
    __builtin_amdgcn_sched_barrier(0);
    if (t0 == 0) {   // W sub-tile 0 of the very first K-tile (every later one is read a phase early by its predecessor)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) f.w0[0][j][kh] = frag<1>(c, 0, j, kh);
    }
    int t = t0;
    const int end = t0 + n;
    for (; t + 2 <= end && t + 4 <= total; t += 2) {   // both tiles have at least two more behind them
        ktile<0, true>(c, acc, f, t, 0);
        ktile<1, true>(c, acc, f, t + 1, 0);
    }
    for (; t < end; t += 2) {                          // the last 1..3 tiles of the sequence: staging stops, the waits drain
        ktile<0, false>(c, acc, f, t, total - t);
        if (t + 1 < end) ktile<1, false>(c, acc, f, t + 1, total - t - 1);
    }
    if (group == 0) __builtin_amdgcn_s_barrier();
}

// acc += A_tile . W_tile^T over nk K-tiles of 64.  Ends with every wave past its last fragment read and every
// DMA landed, so the ring may be reused at once.
__device__ __forceinline__ void run(const Ctx &c, f32x4_t (&acc)[8][4], int nk) {
    prologue(c, nk);
    Frags f;
    tiles(c, acc, f, 0, nk, nk);
}

// ---- persistent sequences of output tiles (gemm_bf16.hip, v4) -------------------------------------------------------
// One workgroup works off several output tiles back to back.  The ring never drains between them: the last two K-tiles of
// an output tile stage units 0..5 of the NEXT one (kt1 / kt2 of ktile_g wrap to 0 / 1 and the per-lane source offsets are
// moved to the next tile's origin), so the next K loop starts on landed data -- no prologue round trip, no ramp -- and the
// write-out in between runs with that DMA in flight.  Both operands use ONE descriptor each for the whole matrix (the
// lane offsets carry the tile origin; rows past the matrix edge are past the descriptor's extent and read as zeros), so a
// tile change is four v_add per half.

// as init(), with whole-matrix descriptors: a [m, lda], w [n, ldw], this workgroup's first tile at rows m0 / n0.
// (tiles_m * 256) * lda * 2 and (tiles_n * 256) * ldw * 2 must fit 32 bits (checked by the launcher).
__device__ __forceinline__ void init_whole(Ctx &c, const uint16_t *a, int64_t lda, int64_t m, const uint16_t *w, int64_t ldw,
                                           int64_t n, int64_t m0, int64_t n0, char *lds, int wave, int lane) {
    init(c, a, lda, 1, w, ldw, 1, lds, wave, lane);
    c.a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a, 0, (int)(uint32_t)(m * lda * 2), 0x00020000);
    c.w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, (int)(uint32_t)(n * ldw * 2), 0x00020000);
    const uint32_t ba = (uint32_t)(m0 * lda * 2), bw = (uint32_t)(n0 * ldw * 2);
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            c.a_off[half][jj] += ba;
            c.w_off[half][jj] += bw;
        }
}

template <int HALF>
__device__ __forceinline__ void bump(Ctx &c, uint32_t da, uint32_t dw) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        c.a_off[HALF][jj] += da;
        c.w_off[HALF][jj] += dw;
    }
}

// One output tile of a persistent sequence: nk K-tiles (even, >= 4).
//   first  the ring holds units 0..5 from prologue(); otherwise the previous tile_p staged them, every wave has waited for
//          its own pieces (vmcnt(0) in the write-out) and a workgroup barrier has passed since
//   da/dw  byte distance from this tile's A / W origin to the next tile's (wrapping 32-bit adds); 0 / 0 behind the last
//          tile of the sequence: its last two K-tiles then stage units 0..5 of the SAME tile once more (96 KiB of reads per
//          workgroup and launch that nobody uses) -- cheaper than a draining copy of the two K-tile bodies, which cost
//          registers (spills) in every tile
// On return every wave is past its last fragment read; units 0..5 of the next tile are in flight or landed in slots 0..5
// and slots 6, 7 are free until the next tile_p (the write-out stages through them).  The caller waits for its DMA
// (vmcnt(0)) in the write-out in any case, so nothing is in flight when the kernel ends.
__device__ __forceinline__ void tile_p(Ctx &c, f32x4_t (&acc)[8][4], Frags &f, int nk, bool first, uint32_t da, uint32_t dw) {
    const int group = c.wave >> 2;
    if (group == 1) __builtin_amdgcn_s_barrier();   // stagger: waves 4-7 run one segment behind
    __builtin_amdgcn_sched_barrier(0);
    if (first) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) f.w0[0][j][kh] = frag<1>(c, 0, j, kh);
    }
#pragma nounroll   // (also keeps the t == 0 iteration from being peeled into a third copy of the two bodies: that spills)
    for (int t = 0; t < nk; t += 2) {
        const bool wrap = t + 2 == nk;               // the two K-tiles that stage the next output tile
        // (workgroup-uniform branches: the eight v_add_u32 of the two bumps are vector instructions, paid in every iteration
        //  when they were unconditional adds of zero)
        if (wrap) bump<0>(c, da, dw);                // half-0 units (j = 0, 1) are staged for kt2: next tile from here on
        ktile_g<0, true>(c, acc, f, t + 1, wrap ? 0 : t + 2, 0, NOWAIT_TEST && !first && t == 0);
        if (wrap) bump<1>(c, da, dw);                // half-1 units (j = 2, 3) are staged for kt1
        ktile_g<1, true>(c, acc, f, wrap ? 0 : t + 2, wrap ? 1 : t + 3, 0, false);
    }
    if (group == 0) __builtin_amdgcn_s_barrier();
}

}  // namespace ml64
