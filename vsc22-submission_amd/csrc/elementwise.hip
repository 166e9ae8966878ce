// HBM-bound row kernels of the encoder: patch extraction, LayerNorm, CLS rows, the fused
// final-LayerNorm + GeM/CLS pooling, the descriptor head and the L2 normalisation.
// One wave (64 lanes) owns a row; rows are read as float4 per lane, reductions are
// wave shuffles; nothing goes through LDS except the per-frame pooling combine.
#include <type_traits>

#include "common.h"

namespace {

constexpr int MAXV = 8;  // float4 per lane -> width <= 2048

// ------------------------------------------------------------------ patchify
// frames f32 [n,C,H,W] -> patches bf16 [n*G*G, kpad], k = c*p*p + py*p + px.
// One thread produces 8 consecutive k (one 16-byte store).  When p % 8 == 0 the 8 source
// pixels are contiguous (two float4 loads); otherwise (p = 14) they are gathered.
__global__ __launch_bounds__(256) void patchify_kernel(const float *__restrict__ frames,
                                                       uint16_t *__restrict__ patches,
                                                       int64_t total_chunks, int channels,
                                                       int image, int patch, int kpad) {
    lp_kernel_entry();
    const int grid = image / patch;
    const int kc = kpad >> 3;
    const int pp = patch * patch;
    const int kreal = channels * pp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total_chunks;
         e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / kc;
        const int k0 = (int)(e - row * kc) * 8;
        const int64_t f = row / (grid * grid);
        const int pidx = (int)(row - f * grid * grid);
        const int gy = pidx / grid, gx = pidx - gy * grid;
        const float *fb = frames + f * (int64_t)channels * image * image;
        float v[8];
        if ((patch & 7) == 0 && k0 + 8 <= kreal) {
            const int c = k0 / pp, rem = k0 - c * pp;
            const int py = rem / patch, px = rem - py * patch;
            const float *src = fb + ((int64_t)c * image + gy * patch + py) * image + gx * patch + px;
            const float4 a = *(const float4 *)src, b = *(const float4 *)(src + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
            v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                if (k < kreal) {
                    const int c = k / pp, rem = k - c * pp;
                    const int py = rem / patch, px = rem - py * patch;
                    v[j] = fb[((int64_t)c * image + gy * patch + py) * image + gx * patch + px];
                } else {
                    v[j] = 0.f;
                }
            }
        }
        uint4 pk;
        pk.x = lp_pack2(v[0], v[1]);
        pk.y = lp_pack2(v[2], v[3]);
        pk.z = lp_pack2(v[4], v[5]);
        pk.w = lp_pack2(v[6], v[7]);
        *(uint4 *)(patches + row * kpad + k0) = pk;
    }
}

// ------------------------------------------------------------------ f32 -> bf16 (weights)
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float *__restrict__ src,
                                                          uint16_t *__restrict__ dst, int64_t rows,
                                                          int cols, int cols_pad) {
    lp_kernel_entry();
    const int64_t total = rows * cols_pad;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / cols_pad;
        const int c = (int)(e - r * cols_pad);
        dst[e] = c < cols ? f32_to_lp(src[r * cols + c]) : (uint16_t)0;
    }
}

// ------------------------------------------------------------------ cls rows
// x[f*T + 0, :] = cls + pos[0, :]
__global__ __launch_bounds__(256) void cls_rows_kernel(float *__restrict__ x,
                                                       const float *__restrict__ cls,
                                                       const float *__restrict__ pos, int64_t frames,
                                                       int tokens, int width) {
    const int64_t total = frames * (width >> 2);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * 256) {
        const int64_t f = e / (width >> 2);
        const int c = (int)(e - f * (width >> 2)) * 4;
        const float4 a = *(const float4 *)(cls + c), b = *(const float4 *)(pos + c);
        *(float4 *)(x + f * tokens * width + c) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

// ------------------------------------------------------------------ LayerNorm
// A wave normalises one row held in registers (two-pass: mean, then centred variance).
template <bool OUT_F32>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x,
                                                        const float *__restrict__ gamma,
                                                        const float *__restrict__ beta,
                                                        void *__restrict__ out, int64_t rows,
                                                        int width, float eps) {
    lp_kernel_entry();
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = width >> 8;            // full float4 rounds of 256 columns
    const int tail = width & 255;         // remaining columns (multiple of 4)
    const float *xr = x + row * width;
    float4 v[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int col = i * 256 + lane * 4;
        const bool on = i < nv || (i == nv && lane * 4 < tail);
        v[i] = on ? *(const float4 *)(xr + col) : make_float4(0.f, 0.f, 0.f, 0.f);
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
        const float4 g = *(const float4 *)(gamma + col), b = *(const float4 *)(beta + col);
        const float y0 = (v[i].x - mean) * rstd * g.x + b.x;
        const float y1 = (v[i].y - mean) * rstd * g.y + b.y;
        const float y2 = (v[i].z - mean) * rstd * g.z + b.z;
        const float y3 = (v[i].w - mean) * rstd * g.w + b.w;
        if (OUT_F32) {
            *(float4 *)((float *)out + row * width + col) = make_float4(y0, y1, y2, y3);
        } else {
            uint2 pk;
            pk.x = lp_pack2(y0, y1);
            pk.y = lp_pack2(y2, y3);
            *(uint2 *)((uint16_t *)out + row * width + col) = pk;
        }
    }
}

// The same for widths <= 1024 (NV float4 per lane, compile-time) in at most 24 VGPRs.  Why 24: the persistent bf16 GEMMs hold
// every CU with 8 waves of 232 VGPRs (plain / GELU / fp32-store write-outs), which leaves 48 of a SIMD's 512 registers per
// lane unused -- two waves of a 24-register kernel.  A LayerNorm of the OTHER lane's chunk then runs beside the GEMM (it needs
// no LDS, and it is bound by HBM while the GEMM's K loop is not) instead of waiting for the GEMM to leave the CUs: the
// encoder's two lanes overlap a memory-bound kernel with a matrix-bound one.  One wave per workgroup so that any single free
// slot can take one.  gamma / beta are fetched per 256-column round (not hoisted) to stay inside the budget.
// sum over the 64 lanes without index registers: four DPP adds inside each row of 16 (as row16_sum of the GEMM write-out), then
// the four row sums by v_readlane (__shfl_xor costs an address VGPR per butterfly step, kept live for the second reduction)
__device__ __forceinline__ float wave_sum_dpp(float x) {
    auto dpp = [](float v, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    x += dpp(x, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    x += dpp(x, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    x += dpp(x, std::integral_constant<int, 0x141>{});  // row_half_mirror
    x += dpp(x, std::integral_constant<int, 0x140>{});  // row_mirror
    const int xi = __builtin_bit_cast(int, x);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
    return (r0 + r1) + (r2 + r3);
}

// widths of exactly NV x 256 columns (768, 1024): no tails, no masks; every access is a buffer operation with ONE per-lane offset
// register (the row base travels in the descriptor, the 256-column round in the instruction's scalar offset)
template <bool OUT_F32, int NV>
__global__ __launch_bounds__(64) void layernorm_light_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, void *__restrict__ out, float eps) {
    lp_kernel_entry();
    constexpr int W = NV * 256;
    const uint32_t off = threadIdx.x * 16u;
    const int64_t row = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)(x + row * W), 0, W * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)gamma, 0, W * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)beta, 0, W * 4, 0x00020000);
    f32x4_t v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, off, i * 1024, 0));
    // (the empty asm statements below keep the reductions sequential: left alone, the compiler packs them into v_pk_add_f32 /
    //  interleaved chains whose operand copies and temporaries are a dozen registers -- this kernel's budget is 24)
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sum += v[i][r];
            asm volatile("" : "+v"(sum));
        }
    const float mean = wave_sum_dpp(sum) * (1.0f / W);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float d;   // one temporary, dead after its statement: (x - mean)^2 accumulated in order
            asm volatile("v_sub_f32 %1, %2, %3\n\tv_fmac_f32 %0, %1, %1" : "+v"(sq), "=&v"(d) : "v"(v[i][r]), "v"(mean));
        }
    const float rstd = rsqrtf(wave_sum_dpp(sq) * (1.0f / W) + eps);
    const float shift = -mean * rstd;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(OUT_F32 ? (void *)((float *)out + row * W) : (void *)((uint16_t *)out + row * W), 0,
                                                                        W * (OUT_F32 ? 4 : 2), 0x00020000);
    // gamma / beta one round at a time, in place: their offset register is made to depend on rstd (and on the previous round)
    // by an empty asm -- independent loads are otherwise all hoisted to the top of the kernel (36-44 registers)
    uint32_t off_gb = off;
    asm volatile("" : "+v"(off_gb) : "v"(rstd));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        f32x4_t y = v[i];
        f32x4_t g = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, off_gb, i * 1024, 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = fmaf(y[r], rstd, shift) * g[r];
        asm volatile("" : "+v"(off_gb) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]));   // gamma is dead: beta lands in its registers
        g = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rb, off_gb, i * 1024, 0));
        y += g;
        asm volatile("" : "+v"(off_gb) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]));
            typedef __attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int u32x2_t;
        if (OUT_F32) {
            buffer_store_b128_soff(__builtin_bit_cast(vsc_u32x4_t, y), ro, off, i * 1024);
        } else {
            u32x2_t pk = {lp_pack2(y[0], y[1]), lp_pack2(y[2], y[3])};
            uint32_t off_o;   // lane * 8, formed per round in a register that is free by now (as a value of the whole kernel it is the 25th)
            asm volatile("v_lshrrev_b32 %0, 1, %1" : "=v"(off_o) : "v"(off_gb));
            __builtin_amdgcn_raw_buffer_store_b64(pk, ro, off_o, i * 512, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------ final LN + pooling
// One workgroup per frame.  Each wave normalises tokens w, w+4, ...; because a lane owns the
// same columns for every token, GeM's per-column sum of clamp(y,1e-6)^p accumulates in
// registers; the 4 waves combine through LDS.  pool = 1 (CLS): token 0 only.
// tokens_out (optional) receives the normalised tokens (parity tests).
template <int NW>   // waves per frame: 16 (width <= 1024) or 8 -- one frame's 197 tokens on 4 waves left the launch latency-bound (146 us for 201 MB)
__global__ __launch_bounds__(NW * 64) void ln_pool_kernel(const float *__restrict__ x,
                                                      const float *__restrict__ gamma,
                                                      const float *__restrict__ beta,
                                                      float *__restrict__ pooled,
                                                      float *__restrict__ tokens_out, int tokens,
                                                      int width, float eps, int pool, float gem_p) {
    extern __shared__ float comb[];   // [NW][width]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t f = blockIdx.x;
    const int nv = width >> 8, tail = width & 255;
    float4 acc[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool cube = gem_p == 3.0f;
    const int tend = (pool == 1 && tokens_out == nullptr) ? 1 : tokens;
    for (int t = wave; t < tend; t += NW) {
        const float *xr = x + (f * tokens + t) * width;
        float4 v[MAXV];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const bool on = i < nv || (i == nv && lane * 4 < tail);
            v[i] = on ? *(const float4 *)(xr + i * 256 + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
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
            const float4 g = *(const float4 *)(gamma + col), b = *(const float4 *)(beta + col);
            float y[4] = {(v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                          (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w};
            if (tokens_out)
                *(float4 *)(tokens_out + (f * tokens + t) * width + col) = make_float4(y[0], y[1], y[2], y[3]);
            if (pool == 1) {
                if (t == 0) acc[i] = make_float4(y[0], y[1], y[2], y[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float c = fmaxf(y[j], 1e-6f);
                    y[j] = cube ? c * c * c : __powf(c, gem_p);
                }
                acc[i].x += y[0]; acc[i].y += y[1]; acc[i].z += y[2]; acc[i].w += y[3];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int col = i * 256 + lane * 4;
        if (col < width) *(float4 *)(&comb[wave * width + col]) = acc[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < width; c += NW * 64) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w += 4)
            s += (comb[w * width + c] + comb[(w + 1) * width + c]) + (comb[(w + 2) * width + c] + comb[(w + 3) * width + c]);
        float r;
        if (pool == 1) r = s;  // only wave 0 / token 0 contributed
        else {
            const float m = s / (float)tokens;
            r = cube ? cbrtf(m) : __powf(m, 1.0f / gem_p);
        }
        pooled[f * width + c] = r;
    }
}

// ------------------------------------------------------------------ GeM over tokens, bf16 input
// pooled[f, c] = (mean_t clamp(x[f, t, c], 1e-6)^p)^(1/p)   (sscd.py:40-42).  A thread owns 8
// consecutive channels (one 16-byte load per token); a block covers 2048 channels of one frame.
__global__ __launch_bounds__(256) void gem_pool_bf16_kernel(const uint16_t *__restrict__ x,
                                                            float *__restrict__ pooled, int tokens,
                                                            int channels, float gem_p) {
    const int64_t f = blockIdx.x;
    const int c0 = (blockIdx.y * 256 + threadIdx.x) * 8;
    if (c0 >= channels) return;
    const bool cube = gem_p == 3.0f;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint16_t *base = x + f * tokens * (int64_t)channels + c0;
    for (int t = 0; t < tokens; ++t) {
        const bf16x8_t v = *(const bf16x8_t *)(base + (int64_t)t * channels);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float c = fmaxf(lp_to_f32((uint16_t)v[j]), 1e-6f);
            acc[j] += cube ? c * c * c : __powf(c, gem_p);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float m = acc[j] / (float)tokens;
        pooled[f * channels + c0 + j] = cube ? cbrtf(m) : __powf(m, 1.0f / gem_p);
    }
}

// ------------------------------------------------------------------ descriptor head
// desc[f,:] = pooled[f,:] . W^T + b (fp32); the L2 normalisation, when asked for, is l2_normalize_kernel behind it.
// One workgroup per (HF = 4 frames, quarter of the outputs).  With one workgroup per frame every frame streamed the
// whole weight matrix (1.5 MiB for 768 -> 512) through four waves: 87 us for 332 frames, the time ONE workgroup needs
// for that stream; now a workgroup reads a quarter of it once for four frames.  A wave computes outputs o0 .. o0+7 of
// the four frames from one pass over the coalesced weight rows; every (frame, output) is one explicit fmaf chain, so
// the bits do not depend on how the frames are grouped (tests/test_gpu_encoder.py::test_encoder_batching_is_invisible).
constexpr int HF = 4, HSPLIT = 4;
__global__ __launch_bounds__(256) void head_kernel(const float *__restrict__ pooled,
                                                   const float *__restrict__ w,
                                                   const float *__restrict__ bias,
                                                   float *__restrict__ desc, int64_t frames, int width, int out_dim) {
    __shared__ __attribute__((aligned(16))) float xs[HF][2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t f0 = (int64_t)blockIdx.x * HF;
    const int nf = frames - f0 < HF ? (int)(frames - f0) : HF;
    const int per = ((out_dim + HSPLIT - 1) / HSPLIT + 31) / 32 * 32;   // outputs per workgroup, whole wave rounds
    const int o_begin = blockIdx.y * per, o_end = o_begin + per < out_dim ? o_begin + per : out_dim;
    for (int q = 0; q < HF; ++q)
        for (int c = threadIdx.x; c < width; c += 256) xs[q][c] = q < nf ? pooled[(f0 + q) * width + c] : 0.f;
    __syncthreads();
    for (int o0 = o_begin + wave * 8; o0 < o_end; o0 += 32) {
        float a[8][HF];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int q = 0; q < HF; ++q) a[j][q] = 0.f;
        for (int c = lane * 4; c < width; c += 256) {
            float4 xv[HF];
#pragma unroll
            for (int q = 0; q < HF; ++q) xv[q] = *(const float4 *)(&xs[q][c]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int o = o0 + j < out_dim ? o0 + j : out_dim - 1;
                const float4 wv = *(const float4 *)(w + (int64_t)o * width + c);
#pragma unroll
                for (int q = 0; q < HF; ++q)
                    a[j][q] = fmaf(wv.w, xv[q].w, fmaf(wv.z, xv[q].z, fmaf(wv.y, xv[q].y, fmaf(wv.x, xv[q].x, a[j][q]))));
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int q = 0; q < HF; ++q) {
                const float r = wave_sum(a[j][q]);
                if (lane == 0 && o0 + j < o_end && q < nf) desc[(f0 + q) * out_dim + o0 + j] = r + (bias ? bias[o0 + j] : 0.f);
            }
    }
}

// ------------------------------------------------------------------ L2 normalise rows in place
__global__ __launch_bounds__(256) void l2_normalize_kernel(float *__restrict__ x, int64_t n, int d) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float *xr = x + row * d;
    float ss = 0.f;
    for (int c = lane; c < d; c += 64) ss += xr[c] * xr[c];
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    if (nrm == 0.f) return;
    for (int c = lane; c < d; c += 64) xr[c] = xr[c] / nrm;
}

inline int grid_for(int64_t work_items) {
    int64_t b = (work_items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

int launch_patchify(const float *frames, uint16_t *patches, int64_t n, int channels, int image,
                    int patch, int kpad, hipStream_t stream) {
    VSC_REQUIRE(frames && patches && n > 0, "patchify: null/empty");
    VSC_REQUIRE(image % patch == 0, "patchify: image %d not a multiple of patch %d", image, patch);
    VSC_REQUIRE(kpad % 64 == 0 && kpad >= channels * patch * patch, "patchify: bad kpad %d", kpad);
    VSC_REQUIRE(image % 4 == 0, "patchify: image width must be a multiple of 4");
    const int g = image / patch;
    const int64_t chunks = n * g * g * (kpad / 8);
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(chunks)), dim3(256), 0, stream, frames, patches,
                       chunks, channels, image, patch, kpad);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

// The same patches from DECODED frames: uint8 [n, H, W, C] (the layout PIL / numpy hand out), with torchvision's
// ToTensor + Normalize (infer/src/transform.py:37-42; extract_query_feats.py:97-105 for the CLIP statistics) applied
// here in their fp32 op order -- u8 / 255 (IEEE divide), minus mean, IEEE divide by std -- so the bf16 patches are
// bit-identical to patchify_kernel on the fp32 tensor the reference would have built, from a quarter of the bytes
// over PCIe and HBM.
struct Norm3 {
    float mean[4], std[4];
};
__global__ __launch_bounds__(256) void patchify_u8_kernel(const uint8_t *__restrict__ frames, uint16_t *__restrict__ patches,
                                                          int64_t total_chunks, int channels, int image, int patch,
                                                          int kpad, Norm3 nm) {
    lp_kernel_entry();
    const int grid = image / patch;
    const int kc = kpad >> 3;
    const int pp = patch * patch;
    const int kreal = channels * pp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total_chunks; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / kc;
        const int k0 = (int)(e - row * kc) * 8;
        const int64_t f = row / (grid * grid);
        const int pidx = (int)(row - f * grid * grid);
        const int gy = pidx / grid, gx = pidx - gy * grid;
        const uint8_t *fb = frames + f * (int64_t)channels * image * image;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            if (k < kreal) {
                const int c = k / pp, rem = k - c * pp;
                const int py = rem / patch, px = rem - py * patch;
                const float t = __fdiv_rn((float)fb[((int64_t)(gy * patch + py) * image + gx * patch + px) * channels + c], 255.0f);
                v[j] = __fdiv_rn(t - nm.mean[c], nm.std[c]);
            } else {
                v[j] = 0.f;
            }
        }
        uint4 pk;
        pk.x = lp_pack2(v[0], v[1]);
        pk.y = lp_pack2(v[2], v[3]);
        pk.z = lp_pack2(v[4], v[5]);
        pk.w = lp_pack2(v[6], v[7]);
        *(uint4 *)(patches + row * kpad + k0) = pk;
    }
}

int launch_patchify_u8(const uint8_t *frames, uint16_t *patches, int64_t n, int channels, int image, int patch, int kpad,
                       const float *mean, const float *std, hipStream_t stream) {
    VSC_REQUIRE(frames && patches && mean && std, "patchify_u8: null pointer");
    VSC_REQUIRE(channels >= 1 && channels <= 4, "patchify_u8: %d channels", channels);
    VSC_REQUIRE(patch > 0 && image % patch == 0 && kpad % 8 == 0 && kpad >= channels * patch * patch,
                "patchify_u8: image %d patch %d kpad %d", image, patch, kpad);
    Norm3 nm{};
    for (int c = 0; c < channels; ++c) {
        VSC_REQUIRE(std[c] > 0.f, "patchify_u8: std[%d] = %g", c, (double)std[c]);
        nm.mean[c] = mean[c];
        nm.std[c] = std[c];
    }
    const int g = image / patch;
    const int64_t chunks = n * g * g * (kpad / 8);
    hipLaunchKernelGGL(patchify_u8_kernel, dim3(grid_for(chunks)), dim3(256), 0, stream, frames, patches, chunks, channels,
                       image, patch, kpad, nm);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

int launch_f32_to_bf16(const float *src, uint16_t *dst, int64_t rows, int cols, int cols_pad,
                       hipStream_t stream) {
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(rows * cols_pad)), dim3(256), 0, stream, src,
                       dst, rows, cols, cols_pad);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

int launch_cls_rows(float *x, const float *cls, const float *pos, int64_t frames, int tokens,
                    int width, hipStream_t stream) {
    hipLaunchKernelGGL(cls_rows_kernel, dim3(grid_for(frames * (width / 4))), dim3(256), 0, stream, x,
                       cls, pos, frames, tokens, width);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

int launch_layernorm(const float *x, const float *g, const float *b, void *out, int64_t rows,
                     int width, float eps, int out_f32, hipStream_t stream) {
    VSC_REQUIRE(x && g && b && out && rows > 0, "layernorm: null/empty");
    VSC_REQUIRE(width % 4 == 0 && width <= MAXV * 256, "layernorm: width %d unsupported", width);
    VSC_REQUIRE((rows + 3) / 4 < (1ll << 31), "layernorm: too many rows");
    const char *lt = vsc_opt(OPT_LN_LIGHT);   // diagnostic: 0 = the 50-register kernel for every width
    if ((width == 768 || width == 1024) && rows < (1ll << 31) && !(lt && lt[0] == '0')) {
        const dim3 grid1((unsigned)rows);
        if (width == 768) {
            if (out_f32) hipLaunchKernelGGL((layernorm_light_kernel<true, 3>), grid1, dim3(64), 0, stream, x, g, b, out, eps);
            else hipLaunchKernelGGL((layernorm_light_kernel<false, 3>), grid1, dim3(64), 0, stream, x, g, b, out, eps);
        } else {
            if (out_f32) hipLaunchKernelGGL((layernorm_light_kernel<true, 4>), grid1, dim3(64), 0, stream, x, g, b, out, eps);
            else hipLaunchKernelGGL((layernorm_light_kernel<false, 4>), grid1, dim3(64), 0, stream, x, g, b, out, eps);
        }
        VSC_CHECK_LAUNCH();
        return VSC_OK;
    }
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (out_f32)
        hipLaunchKernelGGL(layernorm_kernel<true>, grid, dim3(256), 0, stream, x, g, b, out, rows, width, eps);
    else
        hipLaunchKernelGGL(layernorm_kernel<false>, grid, dim3(256), 0, stream, x, g, b, out, rows, width, eps);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

int launch_ln_pool(const float *x, const float *g, const float *b, float *pooled, float *tokens_out,
                   int64_t frames, int tokens, int width, float eps, int pool, float gem_p,
                   hipStream_t stream) {
    VSC_REQUIRE(width % 4 == 0 && width <= 2048, "ln_pool: width %d unsupported", width);
    static bool attr_set[16] = {};
    int dev = 0;
    VSC_CHECK_HIP(hipGetDevice(&dev));
    if (dev >= 16 || !attr_set[dev]) {
        VSC_CHECK_HIP(hipFuncSetAttribute((const void *)ln_pool_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 1024 * 4));
        VSC_CHECK_HIP(hipFuncSetAttribute((const void *)ln_pool_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2048 * 4));
        if (dev < 16) attr_set[dev] = true;
    }
    if (width <= 1024)
        hipLaunchKernelGGL(ln_pool_kernel<16>, dim3((unsigned)frames), dim3(1024), 16 * width * 4, stream, x, g, b, pooled, tokens_out,
                           tokens, width, eps, pool, gem_p);
    else
        hipLaunchKernelGGL(ln_pool_kernel<8>, dim3((unsigned)frames), dim3(512), 8 * width * 4, stream, x, g, b, pooled, tokens_out,
                           tokens, width, eps, pool, gem_p);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

int launch_gem_pool_bf16(const uint16_t *x, float *pooled, int64_t frames, int tokens, int channels, float gem_p,
                         hipStream_t stream) {
    VSC_REQUIRE(channels % 8 == 0, "gem_pool: channels %d not a multiple of 8", channels);
    hipLaunchKernelGGL(gem_pool_bf16_kernel, dim3((unsigned)frames, (channels + 2047) / 2048), dim3(256), 0, stream,
                       x, pooled, tokens, channels, gem_p);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}

int launch_head(const float *pooled, const float *w, const float *bias, float *desc, int64_t frames,
                int width, int out_dim, int l2, hipStream_t stream) {
    VSC_REQUIRE(width <= 2048 && out_dim <= 2048 && width % 4 == 0, "head: dims unsupported");
    if (w) {
        hipLaunchKernelGGL(head_kernel, dim3((unsigned)((frames + HF - 1) / HF), HSPLIT), dim3(256), 0, stream, pooled, w, bias,
                           desc, frames, width, out_dim);
        VSC_CHECK_LAUNCH();
    } else {   // no projection: the pooled vector is the descriptor
        VSC_CHECK_HIP(hipMemcpyAsync(desc, pooled, (size_t)frames * width * 4, hipMemcpyDeviceToDevice, stream));
    }
    return l2 ? launch_l2_normalize(desc, frames, w ? out_dim : width, stream) : VSC_OK;
}

int launch_l2_normalize(float *x, int64_t n, int d, hipStream_t stream) {
    VSC_REQUIRE(x && n > 0 && d > 0, "l2_normalize: null/empty");
    hipLaunchKernelGGL(l2_normalize_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, x, n, d);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}
