// Fused multi-head self-attention for ViT token counts (T <= 320), head_dim 64:
//   out[f, t, h*64:(h+1)*64] = softmax(q k^T / 8) v          per frame f and head h
// (the reference runs this inside the TorchScript backbone as HF ViTSelfAttention /
//  nn.MultiheadAttention, train/train_vid_score/video/clip.py:31-47).
//
// CDNA4 mapping: one workgroup (8 waves) per (frame, head).
//   * K [T,64] is staged row-major into LDS with the same 16-byte-chunk XOR swizzle as the
//     GEMM tiles (conflict-free ds_read_b128 fragment reads); V is staged TRANSPOSED
//     ([64][Tpad] bf16) so the PV contraction index (key) is contiguous per lane.
//   * a wave owns 16 queries at a time.  Scores are computed "swapped":
//     mfma(A = K fragment, B = Q fragment) gives D[key][query], so a lane holds scores of ONE
//     query (lane & 15) for 4 keys per 16-key tile.  Row max / row sum are then in-lane
//     reductions plus two xor-shuffles (16, 32) -- no LDS round trip, no online rescaling
//     because a whole score row (<= 320 keys) lives in registers.
//   * the same registers, packed to bf16, are directly the B operand of the PV MFMA
//     (O^T[dh][query] = V^T[dh][key] . P^T[key][query]); the k-slot -> key assignment of
//     that MFMA is permuted to match (slot (g,j<4) = key 32u+4g+j, slot (g,j>=4) = key
//     32u+16+4g+j-4), which only changes WHICH V elements a lane loads.
//   * O^T puts 4 consecutive head-dim columns of one query in a lane: 8-byte bf16 stores.
//   * waves per workgroup: measured again in round 2 with one wave per query tile (13 waves for 197 tokens, all chains
//     side by side): 139 us per launch against 121 with these 8 waves, and a runtime tile loop with fewer waves is slower
//     too (tools/micro/attn_bench.py).  The launch moves 392 MB (qkv in, context out): ~71 us at HBM speed.
//     Also measured and dropped: persistent workgroups that request the NEXT (frame, head)'s Q / K / V into registers
//     while the current one is computed -- 173 VGPRs halve the residency (159 vs 148 us), capped at 128 VGPRs it spills
//     (239 us).
//   * where the time goes (ablations, -DVSC_ATTN_ABLATION + tools/micro/attn_bench.py, standalone launch, warm): whole
//     kernel 135 us; without MFMAs and exp2 109; without the K / V loads 78; without the stores 92; with neither 64 -- the
//     launch is bound by its memory phases, not by compute, although it moves only 3 TB/s.  Two things followed:
//     the context is written out through 2 KiB of wave-private LDS as whole 128-byte rows (the fragment layout stores
//     32-byte pieces of 16 rows: 135 -> 126 us), and the second resident workgroup of every CU starts half a lifetime late
//     (all workgroups last equally long, so the two residents of a CU otherwise load together and compute together for
//     the whole launch: 126 -> 104 us warm, 149 -> 122 cold).  In the encoder step: 124 -> 107 us per launch.
//   * softmax runs in fp32 with exp2 and a folded scale (1/8 * log2 e); probabilities are
//     rounded to bf16 for the PV MFMA, and the row sum is the sum of those bf16 values, taken by the matrix pipe (an A
//     operand of ones): the context is an exact weighted mean of the V rows.
#include "common.h"

namespace {

constexpr int DH = 64;
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// fmaxf(fmaxf(a, b), c) on MFMA outputs compiles to three v_max_f32 plus a canonicalising v_max x, x per operand; the scores
// are never signalling NaNs, so one v_max3_f32
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int KT, int NI, int ABL = 0>  // key tiles of 32 -> padded token count 32*KT; NI: (frame, head) items per workgroup;
                                         // ABL: diagnostic ablation bits (VSC_ATTN_ABL, KT = 7 only)
__global__ __launch_bounds__(512, 2) void attention_kernel(const uint16_t *__restrict__ qkv,
                                                           uint16_t *__restrict__ out, int tokens,
                                                           int heads, int total, int skew, int ncu) {
    lp_kernel_entry();
    constexpr int TP = KT * 32;
    // Start skew of the SECOND workgroup of every CU (the first 2 x ncu workgroups start together, two per CU; all have the same
    // duration, so without it the two residents of a CU load together and compute together for the whole launch).
    if (skew > 0 && (int)blockIdx.x >= ncu && (int)blockIdx.x < 2 * ncu) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)skew) __builtin_amdgcn_s_sleep(8);
    }
    // bytes per head-dim row of V^T: 16-byte aligned, +32 B (conflict-free ds_read_b128 over a 16-lane group's rows).  Inside a
    // 32-key block the 4-key groups sit in the order a lane consumes them (group g of the even 16-key tile, then group g of the
    // odd one): a lane's 8 keys of a PV step are ONE ds_read_b128.  (As two 8-byte reads 32 B apart the compiler fused them
    // into ds_read2_b64, which moves 128 B/clk against the 256 of b64 / b128.)
    constexpr int VSTRIDE = TP * 2 + 32;
    constexpr int QT_MAX = (2 * KT + 7) / 8;  // 16-query tiles per wave (8 waves)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *klds = smem;             // [TP][64] bf16, 128-B rows, chunk ^= (row >> 1) & 7
    char *vt = smem + TP * 128;    // [64][VSTRIDE]
    char *ost = vt + 64 * VSTRIDE; // 8 waves x 2 KiB: write-out transposition

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int width = heads * DH;
    const int64_t ld = 3 * (int64_t)width;
    const int fr = lane & 15, g = lane >> 4;
    const int qtiles = (tokens + 15) >> 4;
    const float scale = 0.125f * 1.44269504088896340736f;  // 1/sqrt(64) * log2(e)

    constexpr int KIT = (TP * 8 + 511) / 512;
    constexpr int VTASKS = ((TP / 4 + 15) / 16) * 128;   // 16 key groups x 8 column blocks per 128 tasks
    constexpr int VIT = (VTASKS + 511) / 512;
    struct KV {
        uint4 kv[KIT];
        bf16x8_t vr[VIT][4];
    };
    auto q_of = [&](int item) { return qkv + (int64_t)(item / heads) * tokens * ld + (item % heads) * DH; };

    // ---- Q fragments of every query tile this wave owns, issued first: their HBM latency
    //      hides behind the K/V staging instead of stalling each tile.
    auto load_q = [&](int item, bf16x8_t (&qf)[QT_MAX][2]) {
        const uint16_t *qptr = q_of(item);
#pragma unroll
        for (int i = 0; i < QT_MAX; ++i) {
            int qrow = (wave + 8 * i) * 16 + fr;
            qrow = qrow < tokens ? qrow : tokens - 1;
            qf[i][0] = *(const bf16x8_t *)(qptr + qrow * ld + g * 8);
            qf[i][1] = *(const bf16x8_t *)(qptr + qrow * ld + g * 8 + 32);
        }
    };
    // ---- K (row-major, swizzled; pad rows are zero) and V (transposed): ALL of a thread's K and V rows are requested
    //      before the first LDS write, so the workgroup pays one memory round trip for its 50 KB instead of two (K, then V).
    //      V task = (4 keys) x (8 head-dim columns); 16 consecutive lanes take 16 consecutive key groups of one column
    //      block, so every ds_write_b64 of a 16-lane group lands on 128 contiguous bytes (conflict-free).
    auto load_kv = [&](int item, KV &r) {
        const uint16_t *kptr = q_of(item) + width, *vptr = q_of(item) + 2 * width;
#pragma unroll
        for (int i = 0; i < KIT; ++i) {
            const int e = tid + i * 512, row = e >> 3, c = e & 7;
            r.kv[i] = make_uint4(0, 0, 0, 0);
            if (e < TP * 8 && row < tokens && !(ABL & 8)) r.kv[i] = *(const uint4 *)(kptr + row * ld + c * 8);
        }
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
            const int e = tid + it * 512;
            const int blk = e >> 7, c8 = (e >> 4) & 7, kg = blk * 16 + (e & 15);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = kg * 4 + i;
                r.vr[it][i] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
                if (e < VTASKS && kg < TP / 4 && row < tokens && !(ABL & 8)) r.vr[it][i] = *(const bf16x8_t *)(vptr + row * ld + c8 * 8);
            }
        }
    };
    auto store_kv = [&](const KV &r) {
#pragma unroll
        for (int i = 0; i < KIT; ++i) {
            const int e = tid + i * 512, row = e >> 3, c = e & 7;
            if (e < TP * 8) *(uint4 *)(klds + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = r.kv[i];
        }
#pragma unroll
        for (int it = 0; it < VIT; ++it) {
            const int e = tid + it * 512;
            const int blk = e >> 7, c8 = (e >> 4) & 7, kg = blk * 16 + (e & 15);
            if (e >= VTASKS || kg >= TP / 4) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint2 pk;
                pk.x = (uint32_t)(uint16_t)r.vr[it][0][j] | ((uint32_t)(uint16_t)r.vr[it][1][j] << 16);
                pk.y = (uint32_t)(uint16_t)r.vr[it][2][j] | ((uint32_t)(uint16_t)r.vr[it][3][j] << 16);
                *(uint2 *)(vt + (c8 * 8 + j) * VSTRIDE + ((kg & ~7) | ((kg & 3) << 1) | ((kg >> 2) & 1)) * 8) = pk;
            }
        }
    };
    auto compute = [&](int item, const bf16x8_t (&qf)[QT_MAX][2]) {
        const int frame = item / heads, head = item - frame * heads;
#pragma unroll
        for (int qi = 0; qi < QT_MAX; ++qi) {
            const int qt = wave + 8 * qi;
            if (qt >= qtiles) break;

            // scores: s[t][r] = <q[query = fr], k[key = 16 t + 4 g + r]>
            f32x4_t s[2 * KT];
#pragma unroll
            for (int t = 0; t < 2 * KT; ++t) {
                s[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                const int krow = t * 16 + fr;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8_t kf =
                        *(const bf16x8_t *)(klds + krow * 128 + (((g + 4 * kk) ^ ((krow >> 1) & 7)) << 4));
                    if (ABL & 4) s[t][kk] += (float)kf[0] + (float)qf[qi][kk][1];
                    else s[t] = lp_mfma16(kf, qf[qi][kk], s[t]);
                }
                // keep the scheduler from hoisting every tile's K fragments (register blow-up)
                if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            // row max on the raw scores (scale > 0); only the ragged tail tile needs masking
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2 * KT; ++t) {
                if (t >= 2 * KT - 2) {  // KT = ceil(tokens / 32): only the last two 16-key tiles can hold padding
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (t * 16 + g * 4 + r >= tokens) s[t][r] = -INFINITY;
                }
                mx = max3(max3(mx, s[t][0], s[t][1]), s[t][2], s[t][3]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mxs = mx * scale;
            const f32x2_t sc2 = (f32x2_t){scale, scale}, nm2 = (f32x2_t){-mxs, -mxs};
            bf16x8_t pb[KT];
#pragma unroll
            for (int u = 0; u < KT; ++u) {
                float e[8];
#pragma unroll
                for (int h = 0; h < 4; ++h) {   // v_pk_fma_f32: two scores per instruction
                    const f32x4_t &sv = s[2 * u + (h >> 1)];
                    const f32x2_t d = (f32x2_t){sv[2 * (h & 1)], sv[2 * (h & 1) + 1]} * sc2 + nm2;
                    e[2 * h] = (ABL & 1) ? d[0] : __builtin_amdgcn_exp2f(d[0]);
                    e[2 * h + 1] = (ABL & 1) ? d[1] : __builtin_amdgcn_exp2f(d[1]);
                }
                union { uint32_t w[4]; bf16x8_t v; } pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk.w[r] = lp_pack2(e[2 * r], e[2 * r + 1]);
                pb[u] = pk.v;
            }
            // O^T[dh][query] += V^T[dh][key] . P^T[key][query]; a fifth A operand of ones gives the row sums of the bf16 P the
            // products use (every row of that tile = the sum over the keys: no VALU adds, no cross-lane reduction)
            const bf16x8_t ones = LP_ONES;
            f32x4_t osum = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            f32x4_t o[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) o[ct] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < KT; ++u) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const bf16x8_t vf = *(const bf16x8_t *)(vt + (ct * 16 + fr) * VSTRIDE + (32 * u + 8 * g) * 2);
                    if (ABL & 2) o[ct][u & 3] += (float)vf[0] + (float)pb[u][ct];
                    else o[ct] = lp_mfma16(vf, pb[u], o[ct]);
                }
                if (ABL & 2) osum[0] += (float)pb[u][0];
                else osum = lp_mfma16(ones, pb[u], osum);
                if (u & 1) __builtin_amdgcn_sched_barrier(0);
            }
            const float inv = __builtin_amdgcn_rcpf(osum[0]);
            // write-out through 2 KiB of wave-private LDS: a lane's accumulators are 4 head-dim columns of 16 different queries,
            // stored directly that is 32-byte pieces of 16 rows per instruction (four instructions per 128-byte row: 4 x the
            // store requests -- the stores cost 43 of the launch's 135 us).  Transposed, eight lanes write one whole 128-byte
            // (token, head) row with 16-byte stores.  Same swizzle as the GEMM write-out (gemm_bf16.hip, epilogue_via_lds).
            {
                char *reg = ost + wave * 2048;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    uint2 pk;
                    pk.x = lp_pack2(o[ct][0] * inv, o[ct][1] * inv);
                    pk.y = lp_pack2(o[ct][2] * inv, o[ct][3] * inv);
                    const int chunk = 2 * ct + (g >> 1);
                    *(uint2 *)(reg + fr * 128 + ((chunk ^ (fr & 7)) << 4) + ((g ^ (fr >> 3)) & 1) * 8) = pk;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int c = lane & 7;
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int row = it * 8 + (lane >> 3);
                    uint4 d = *(const uint4 *)(reg + row * 128 + ((c ^ (row & 7)) << 4));
                    if (it & 1) d = make_uint4(d.z, d.w, d.x, d.y);
                    const int q = qt * 16 + row;
                    if (q < tokens && !(ABL & 16)) *(uint4 *)(out + ((int64_t)frame * tokens + q) * width + head * DH + c * 8) = d;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    };

    // ---- NI items per workgroup: the K / V rows of item i+1 are requested (into registers) before item i is computed and
    //      staged behind it, so only the first item's load latency is exposed.
    const int item0 = blockIdx.x * NI;
    bf16x8_t qf[QT_MAX][2];
    {
        KV cur;
        load_q(item0, qf);
        load_kv(item0, cur);
        store_kv(cur);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int item = item0 + it;
        const bool more = it + 1 < NI && item + 1 < total;   // workgroup-uniform
        KV nxt;
        if (more) load_kv(item + 1, nxt);
        compute(item, qf);
        if (!more) break;
        __syncthreads();   // every wave is done with this item's K / V
        load_q(item + 1, qf);
        store_kv(nxt);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The same attention as a PERSISTENT kernel fed by LDS-DMA (round 4): one 16-wave workgroup per CU walks the (frame, head) items;
// K and V of item i + 1 travel HBM -> LDS by buffer_load ... lds (no registers) into the second half of a double buffer while
// item i is computed, so a CU computes all the time instead of alternating two workgroups between a load and a compute phase
// (the kernel above moves 3 TB/s and keeps the matrix pipe 22 % busy: it is bound by that alternation, not by a pipe).
//   * LDS-DMA writes lane-linear images, so V cannot be transposed on the way as above.  It is stored ROW-major like K
//     ([key][64] bf16, 128-byte rows) and the PV operand V^T[dim][key] is read with ds_read_b64_tr_b16: each lane reads the
//     8 bytes V[key0 + ((l >> 2) & 3)][dim0 + 4 (l & 3) .. + 3] and receives V[key0 .. key0 + 3][dim0 + (l & 15)] -- 4 consecutive
//     keys of its own head dim (semantics measured with tools/micro/tr_b16_probe.hip: lane l of a 16-lane group gets element
//     l & 3 of the pieces read by lanes (l >> 2) + 4 j, j = 0..3).  Two reads (keys 32 u + 4 g .. and 32 u + 16 + 4 g ..) make
//     the 8 k-slots of a lane, in exactly the slot -> key order the packed P uses.
//   * swizzles (applied to the DMA's SOURCE address, and again on the read): K 16-byte chunk c -> c ^ ((row >> 1) & 7) as in
//     the GEMM tiles; V 32-byte chunk c -> c ^ ((row >> 1) & 3): the 8 rows x 32 bytes a 32-lane half reads in one
//     transpose-read then cover all 64 banks once.
//   * one 16-query tile per wave (tokens <= 256: 13 of the 16 waves at 197 tokens), Q fragments of the next item requested
//     during the current one; rows past the last token lie past the DMA descriptors' extent and arrive as zeros.
// Scores, softmax, P packing, ones-operand row sum and the write-out are those of the kernel above.
template <int KT>
__global__ __launch_bounds__(1024) void attention_dma_kernel(const uint16_t *__restrict__ qkv, uint16_t *__restrict__ out, int tokens,
                                                             int heads, int total) {
    lp_kernel_entry();
    typedef __attribute__((address_space(3))) void *lptr_t;
    typedef __attribute__((ext_vector_type(4))) short s16x4_t;
    typedef __attribute__((address_space(3))) s16x4_t *ldstr_t;
    constexpr int TP = KT * 32, OP_BYTES = TP * 128, PIECES = TP / 8;   // one operand of one item: TP rows of 128 B = PIECES KiB
    extern __shared__ __attribute__((aligned(16))) char smem[];         // [2 buffers][K | V] + 16 x 2 KiB write-out staging
    char *ost = smem + 4 * OP_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int width = heads * DH;
    const int64_t ld = 3 * (int64_t)width;
    const uint32_t ld_bytes = (uint32_t)ld * 2u;
    const int fr = lane & 15, g = lane >> 4;
    const int qtiles = (tokens + 15) >> 4;
    const float scale = 0.125f * 1.44269504088896340736f;
    auto q_of = [&](int item) { return qkv + (int64_t)(item / heads) * tokens * ld + (item % heads) * DH; };
    const uint32_t op_extent = (uint32_t)(tokens - 1) * ld_bytes + 128u;

    auto stage = [&](int item, int buf) {
        const uint16_t *kp = q_of(item) + width;
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)kp, 0, (int)op_extent, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(kp + width), 0, (int)op_extent, 0x00020000);
#pragma unroll
        for (int j = 0; j < (2 * PIECES + 15) / 16; ++j) {
            const int pc = wave + 16 * j;            // wave-uniform
            if (pc >= 2 * PIECES) break;
            const int op = pc >= PIECES, pp = pc - op * PIECES;
            const int row = pp * 8 + (lane >> 3), chunk = lane & 7;
            const int src = op ? ((((chunk >> 1) ^ ((row >> 1) & 3)) << 1) | (chunk & 1)) : (chunk ^ ((row >> 1) & 7));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(op ? rv : rk, (lptr_t)(smem + (buf * 2 + op) * OP_BYTES + pp * 1024), 16,
                                                     (uint32_t)row * ld_bytes + (uint32_t)src * 16u, 0, 0, 0);
        }
    };
    auto load_q = [&](int item, bf16x8_t (&qf)[2]) {
        const uint16_t *qptr = q_of(item);
        int qrow = wave * 16 + fr;
        qrow = qrow < tokens ? qrow : tokens - 1;
        qf[0] = *(const bf16x8_t *)(qptr + qrow * ld + g * 8);
        qf[1] = *(const bf16x8_t *)(qptr + qrow * ld + g * 8 + 32);
    };
    // per-lane part of the transpose-read address: row 4 g + r4 of a 32-key block, its 32-byte chunk swizzle, 8-byte column group
    const int r4 = (lane >> 2) & 3, sw = (2 * g + (r4 >> 1)) & 3;
    const uint32_t vlane = (uint32_t)(4 * g + r4) * 128u + (uint32_t)(lane & 3) * 8u;

    auto compute = [&](int item, const char *klds, const char *vlds, const bf16x8_t (&qf)[2]) {
        const int frame = item / heads, head = item - frame * heads;
        const int qt = wave;
        f32x4_t s[2 * KT];
#pragma unroll
        for (int t = 0; t < 2 * KT; ++t) {
            s[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            const int krow = t * 16 + fr;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8_t kf = *(const bf16x8_t *)(klds + krow * 128 + (((g + 4 * kk) ^ ((krow >> 1) & 7)) << 4));
                s[t] = lp_mfma16(kf, qf[kk], s[t]);
            }
            if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2 * KT; ++t) {
            if (t >= 2 * KT - 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (t * 16 + g * 4 + r >= tokens) s[t][r] = -INFINITY;
            }
            mx = max3(max3(mx, s[t][0], s[t][1]), s[t][2], s[t][3]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mxs = mx * scale;
        const f32x2_t sc2 = (f32x2_t){scale, scale}, nm2 = (f32x2_t){-mxs, -mxs};
        bf16x8_t pb[KT];
#pragma unroll
        for (int u = 0; u < KT; ++u) {
            float e[8];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const f32x4_t &sv = s[2 * u + (h >> 1)];
                const f32x2_t d = (f32x2_t){sv[2 * (h & 1)], sv[2 * (h & 1) + 1]} * sc2 + nm2;
                e[2 * h] = __builtin_amdgcn_exp2f(d[0]);
                e[2 * h + 1] = __builtin_amdgcn_exp2f(d[1]);
            }
            union { uint32_t w[4]; bf16x8_t v; } pk;
#pragma unroll
            for (int r = 0; r < 4; ++r) pk.w[r] = lp_pack2(e[2 * r], e[2 * r + 1]);
            pb[u] = pk.v;
        }
        const bf16x8_t ones = LP_ONES;
        f32x4_t osum = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        f32x4_t o[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) o[ct] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KT; ++u) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const char *va = vlds + vlane + ((ct ^ sw) << 5) + u * 4096;
                union { s16x4_t h[2]; bf16x8_t v; } vf;
                vf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ldstr_t)va);            // keys 32 u + 4 g .. + 3
                vf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ldstr_t)(va + 2048));   // keys 32 u + 16 + 4 g .. + 3
                o[ct] = lp_mfma16(vf.v, pb[u], o[ct]);
            }
            osum = lp_mfma16(ones, pb[u], osum);
            if (u & 1) __builtin_amdgcn_sched_barrier(0);
        }
        const float inv = __builtin_amdgcn_rcpf(osum[0]);
        char *reg = ost + wave * 2048;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            uint2 pk;
            pk.x = lp_pack2(o[ct][0] * inv, o[ct][1] * inv);
            pk.y = lp_pack2(o[ct][2] * inv, o[ct][3] * inv);
            const int chunk = 2 * ct + (g >> 1);
            *(uint2 *)(reg + fr * 128 + ((chunk ^ (fr & 7)) << 4) + ((g ^ (fr >> 3)) & 1) * 8) = pk;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int c = lane & 7;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = it * 8 + (lane >> 3);
            uint4 d = *(const uint4 *)(reg + row * 128 + ((c ^ (row & 7)) << 4));
            if (it & 1) d = make_uint4(d.z, d.w, d.x, d.y);
            const int q = qt * 16 + row;
            if (q < tokens) *(uint4 *)(out + ((int64_t)frame * tokens + q) * width + head * DH + c * 8) = d;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    int item = blockIdx.x;
    if (item >= total) return;
    bf16x8_t qf[2];
    stage(item, 0);
    load_q(item, qf);
    for (int it = 0;; ++it) {
        const int nxt = item + (int)gridDim.x;
        const bool more = nxt < total;    // workgroup-uniform
        // this wave's DMA pieces of `item` have landed (and nothing of the previous item's write-out is in flight); the barrier
        // publishes every wave's pieces and tells that every wave is done with the buffer the next stage() overwrites
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        bf16x8_t qn[2];
        if (more) {
            stage(nxt, (it + 1) & 1);
            load_q(nxt, qn);
        }
        if (wave < qtiles) compute(item, smem + (it & 1) * 2 * OP_BYTES, smem + ((it & 1) * 2 + 1) * OP_BYTES, qf);
        if (!more) break;
        qf[0] = qn[0];
        qf[1] = qn[1];
        item = nxt;
    }
}

template <int KT>
int launch_kt(const uint16_t *qkv, uint16_t *out, int frames, int tokens, int heads,
              hipStream_t stream) {
    constexpr int TP = KT * 32;
    constexpr int smem = TP * 128 + 64 * (TP * 2 + 32) + 8 * 2048;
    static bool attr_set[16] = {};   // per device (one process may drive several)
    int dev = 0;
    VSC_CHECK_HIP(hipGetDevice(&dev));
    if (dev >= 16 || !attr_set[dev]) {
        VSC_CHECK_HIP(hipFuncSetAttribute((const void *)attention_kernel<KT, 1>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        VSC_CHECK_HIP(hipFuncSetAttribute((const void *)attention_kernel<KT, 2>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        if (dev < 16) attr_set[dev] = true;
    }
    // start skew of each CU's second resident (see the kernel): half a workgroup lifetime.  Measured at 197 tokens (lifetime
    // ~32 k cycles): 0 / 4 / 8 / 12 / 16 / 24 / 32 k cycles -> 148.6 / 140.3 / 137.6 / 132.4 / 121.8 / 136.4 / 139.4 us cold,
    // 116.9 / 114.1 / 107.6 / 104.9 / 104.2 / 116.9 / 114.7 us warm (tools/micro/attn_bench.py).  Two workgroups are resident
    // from 5 key tiles up (LDS); the lifetime goes with the square of the token count.
    static int cus_of[16] = {};
    if (dev < 16 && !cus_of[dev]) VSC_CHECK_HIP(hipDeviceGetAttribute(&cus_of[dev], hipDeviceAttributeMultiprocessorCount, dev));
    const int ncu = dev < 16 && cus_of[dev] > 0 ? cus_of[dev] : 256;
    int skew = KT >= 5 && frames * heads > 2 * ncu ? 16000 * KT * KT / 49 : 0;
    if (const char *e = vsc_opt(OPT_ATTN_SKEW)) skew = atoi(e);
#ifdef VSC_ATTN_ABLATION
    if (KT == 7)
        if (const char *e = vsc_opt(OPT_ATTN_ABL)) {
            const int abl = atoi(e);
#define VSC_ABL_CASE(A) case A: { auto k = attention_kernel<7, 1, A>; VSC_CHECK_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem)); \
            hipLaunchKernelGGL(k, dim3(frames * heads), dim3(512), smem, stream, qkv, out, tokens, heads, frames * heads, skew, ncu); VSC_CHECK_LAUNCH(); return VSC_OK; }
            switch (abl) { VSC_ABL_CASE(1) VSC_ABL_CASE(2) VSC_ABL_CASE(4) VSC_ABL_CASE(6) VSC_ABL_CASE(7) VSC_ABL_CASE(8) VSC_ABL_CASE(16) VSC_ABL_CASE(24) VSC_ABL_CASE(31) default: break; }
        }
#endif
    const int total = frames * heads;
    // Persistent LDS-DMA form (attention_dma_kernel; one query tile per wave: tokens <= 256, two K/V buffers + staging within 160 KiB:
    // KT <= 8): opt-in by VSC_ATTN_DMA=1.  Bit-identical results; measured 110 us per launch against 103 us for the kernel below in the
    // ViT-B step (four alternating runs, same box) -- with all loads off the critical path the item time is the compute phase's
    // instruction stream (MFMA + softmax VALU + LDS issue add up on a SIMD), which 13 waves side by side do not shorten.
    if constexpr (KT <= 8) {
        const char *dm = vsc_opt(OPT_ATTN_DMA);
        if (tokens <= 256 && dm && dm[0] == '1') {
            constexpr int smem_dma = 4 * TP * 128 + 16 * 2048;
            static bool dma_attr[16] = {};
            if (dev >= 16 || !dma_attr[dev]) {
                VSC_CHECK_HIP(hipFuncSetAttribute((const void *)attention_dma_kernel<KT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_dma));
                if (dev < 16) dma_attr[dev] = true;
            }
            const int grid = total < ncu ? total : ncu;
            hipLaunchKernelGGL((attention_dma_kernel<KT>), dim3(grid), dim3(1024), smem_dma, stream, qkv, out, tokens, heads, total);
            VSC_CHECK_LAUNCH();
            return VSC_OK;
        }
    }
    // items per workgroup: 1.  Two (the second one's K / V rows requested into registers before the first is computed) were
    // measured again on this kernel (tools/micro/attn_ni.py): 166 VGPRs, i.e. one resident workgroup per CU, or capped at
    // 128 VGPRs 120 bytes of scratch -- 160 us per launch against 110-119.  VSC_ATTN_NI=2 keeps the variant reachable.
    int ni = 1;
    if (const char *e = vsc_opt(OPT_ATTN_NI)) ni = atoi(e) == 2 ? 2 : 1;
    if (ni == 2)
        hipLaunchKernelGGL((attention_kernel<KT, 2>), dim3((total + 1) / 2), dim3(512), smem, stream, qkv, out, tokens, heads,
                           total, 2 * skew, ncu);
    else
        hipLaunchKernelGGL((attention_kernel<KT, 1>), dim3(total), dim3(512), smem, stream, qkv, out, tokens, heads, total, skew, ncu);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

}  // namespace

int launch_attention_bf16(const uint16_t *qkv, uint16_t *out, int frames, int tokens, int heads,
                          hipStream_t stream) {
    VSC_REQUIRE(qkv && out, "attention: null operand");
    VSC_REQUIRE(frames > 0 && tokens > 0 && heads > 0, "attention: empty problem");
    VSC_REQUIRE((int64_t)frames * heads < (1ll << 31), "attention: grid too large");
    const int kt = (tokens + 31) / 32;
    switch (kt) {
        case 1: return launch_kt<1>(qkv, out, frames, tokens, heads, stream);
        case 2: return launch_kt<2>(qkv, out, frames, tokens, heads, stream);
        case 3: return launch_kt<3>(qkv, out, frames, tokens, heads, stream);
        case 4: return launch_kt<4>(qkv, out, frames, tokens, heads, stream);
        case 5: return launch_kt<5>(qkv, out, frames, tokens, heads, stream);
        case 6: return launch_kt<6>(qkv, out, frames, tokens, heads, stream);
        case 7: return launch_kt<7>(qkv, out, frames, tokens, heads, stream);
        case 8: return launch_kt<8>(qkv, out, frames, tokens, heads, stream);
        case 9: return launch_kt<9>(qkv, out, frames, tokens, heads, stream);
        case 10: return launch_kt<10>(qkv, out, frames, tokens, heads, stream);
        default:
            VSC_REQUIRE(false, "attention: %d tokens unsupported (max 320; windowed/long sequences are a later row)",
                        tokens);
    }
    return VSC_OK;
}
