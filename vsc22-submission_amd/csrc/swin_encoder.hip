// vsc_swin: weights, workspace and launch sequence of the Swin-Transformer-V2 frame encoder
// (reference model: train/train_v115/torch2scripts.py, SwinTransformerV2 :480-657).
//
// HBM layout for one step of B frames; stage s has res_s x res_s tokens of width C_s (M = B res^2,
// M*C halves from stage to stage, so stage-0 sizes bound every buffer):
//   patches bf16 [B L0, 64]     x  f32 [M, C]  residual stream       xb bf16 [M, C]  its shadow
//   t       f32  [M, C]   fp32 GEMM outputs awaiting their post-LayerNorm (proj, fc2, reduction)
//   qkv     bf16 [M, 3C]        att bf16 [M, C]      h bf16 [M, 4C]   merged bf16 [M/4, 4C]
// Per block (res-post-norm):  qkv = xb Wqkv^T + (q_bias|0|v_bias);  att = window attention;
//   t = att Wproj^T + b;  x += LN1(t);  h = gelu(xb W1^T + b1);  t = h W2^T + b2;  x += LN2(t).
// The continuous-position-bias tables 16*sigmoid(cpb_mlp(coords))[index] depend only on the
// weights: they are evaluated once on the host in finalize (model loading) and kept as
// fp32 [heads, (2w-1)^2] per block (the kernel applies the relative_position_index map itself).
#include <math.h>

#include <map>
#include <string>
#include <vector>

#include "common.h"

struct SwinBlockW {
    uint16_t *qkv_w, *proj_w, *fc1_w, *fc2_w;
    uint16_t *fc2_wp = nullptr;   // widths 128 / 256: fc2.weight with the hidden axis in the fused MLP's contraction order
    float *qkv_b, *proj_b, *fc1_b, *fc2_b, *n1_g, *n1_b, *n2_g, *n2_b, *bias, *scale;
};
struct SwinStageW {
    std::vector<SwinBlockW> blocks;
    uint16_t *red_w = nullptr;
    float *dn_g = nullptr, *dn_b = nullptr;
};

struct vsc_swin {
    vsc_swin_config cfg;
    bool finalized = false;
    std::map<std::string, std::vector<float>> host_w;
    std::map<std::string, size_t> expect;
    std::vector<void *> allocs;
    std::vector<SwinStageW> stages;
    uint16_t *pe_w = nullptr;
    float *pe_b = nullptr, *pe_g = nullptr, *pe_beta = nullptr, *norm_g = nullptr, *norm_b = nullptr,
          *out_w = nullptr, *out_b = nullptr;
    int kpad = 0;
    // One workspace per lane: the max_batch chunks of a forward call alternate over two internal streams (as the ViT
    // encoder's lanes do, encoder.hip), so the tail of one chunk's kernel overlaps the head of the other's: +4 % at
    // 2 x 256 frames (tools/micro/swin_two_lanes.py: 12.77 k -> 13.28 k frames/s).
    struct Workspace {
        uint16_t *patches = nullptr, *xb = nullptr, *qkv = nullptr, *att = nullptr, *h = nullptr, *merged = nullptr;
        float *x = nullptr, *t = nullptr, *pooled = nullptr;
        void *lnws = nullptr;   // pair-exchange workspace of the persistent gemm_ln (private to the lane's stream)
    } ws[2];
    hipStream_t lane_stream[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    int64_t ws_bytes = 0;
    size_t ws_sizes[9] = {0};
    bool lanes_ready = false;   // second workspace + lane streams: made by the first call that has more than one chunk
    // per-launch HIP events (vsc_swin_set_profiling), as in encoder.hip
    bool profile = false;
    struct Span { int cls; size_t e0, e1; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    double prof_ms[VSC_SWIN_PROF_CLASSES] = {0};
    int64_t prof_n[VSC_SWIN_PROF_CLASSES] = {0};

    int res(int s) const { return cfg.image_size / cfg.patch_size >> s; }
    int dim(int s) const { return cfg.embed_dim << s; }
    int window(int s) const { return cfg.window_size < res(s) ? cfg.window_size : res(s); }
    int shift(int s, int b) const { return res(s) <= cfg.window_size ? 0 : ((b & 1) ? cfg.window_size / 2 : 0); }
};

namespace {

int sw_alloc(vsc_swin *e, size_t bytes, void **out) {
    hipError_t err = hipMalloc(out, bytes);
    if (err != hipSuccess) {
        vsc_set_error("swin: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
        return VSC_ERR_NOMEM;
    }
    e->allocs.push_back(*out);
    return VSC_OK;
}

int sw_upload_f32v(vsc_swin *e, const std::vector<float> &v, float **out) {
    int rc = sw_alloc(e, v.size() * 4, (void **)out);
    if (rc) return rc;
    VSC_CHECK_HIP(hipMemcpy(*out, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return VSC_OK;
}
int sw_upload_f32(vsc_swin *e, const std::string &name, float **out) { return sw_upload_f32v(e, e->host_w.at(name), out); }

int sw_upload_bf16v(vsc_swin *e, const std::vector<float> &v, const std::string &name, int64_t rows, int cols, int cols_pad, uint16_t **out) {
    float *tmp = nullptr;
    VSC_CHECK_HIP(hipMalloc((void **)&tmp, v.size() * 4));
    hipError_t err = hipMemcpy(tmp, v.data(), v.size() * 4, hipMemcpyHostToDevice);
    int rc = err == hipSuccess ? sw_alloc(e, (size_t)rows * cols_pad * 2, (void **)out) : VSC_ERR_HIP;
    if (!rc) rc = launch_f32_to_bf16(tmp, *out, rows, cols, cols_pad, nullptr);
    hipError_t e2 = hipDeviceSynchronize();
    (void)hipFree(tmp);
    if (err != hipSuccess || e2 != hipSuccess) {
        vsc_set_error("swin: uploading %s failed", name.c_str());
        return VSC_ERR_HIP;
    }
    return rc;
}
int sw_upload_bf16(vsc_swin *e, const std::string &name, int64_t rows, int cols, int cols_pad, uint16_t **out) {
    return sw_upload_bf16v(e, e->host_w.at(name), name, rows, cols, cols_pad, out);
}

// 16 * sigmoid(cpb_mlp(log-spaced relative coords)) -> compact table [heads, (2w-1)^2]
// (torch2scripts.py:100-128, 166-171).
std::vector<float> position_bias(const std::vector<float> &w0, const std::vector<float> &b0,
                                 const std::vector<float> &w2, int window, int pretrained, int heads) {
    const int side = 2 * window - 1;
    const float denom = (float)((pretrained > 0 ? pretrained : window) - 1);
    std::vector<float> table((size_t)side * side * heads);
    std::vector<float> hid(512);
    for (int a = 0; a < side; ++a)
        for (int b = 0; b < side; ++b) {
            float c[2] = {(float)(a - (window - 1)) / denom * 8.f, (float)(b - (window - 1)) / denom * 8.f};
            for (int k = 0; k < 2; ++k) {
                const float s = c[k] > 0.f ? 1.f : (c[k] < 0.f ? -1.f : 0.f);
                c[k] = s * log2f(fabsf(c[k]) + 1.0f) / 3.0f;  // log2(8) = 3
            }
            for (int j = 0; j < 512; ++j) {
                const float v = c[0] * w0[j * 2] + c[1] * w0[j * 2 + 1] + b0[j];
                hid[j] = v > 0.f ? v : 0.f;
            }
            for (int hh = 0; hh < heads; ++hh) {
                float acc = 0.f;
                for (int j = 0; j < 512; ++j) acc += hid[j] * w2[(size_t)hh * 512 + j];
                table[((size_t)a * side + b) * heads + hh] = acc;
            }
        }
    // compact [heads, side*side]: the kernel gathers bias[i][j] = table[(yi-yj+w-1)*side + xi-xj+w-1]
    std::vector<float> bias((size_t)heads * side * side);
    for (size_t idx = 0; idx < (size_t)side * side; ++idx)
        for (int hh = 0; hh < heads; ++hh)
            bias[(size_t)hh * side * side + idx] = 16.0f / (1.0f + expf(-table[idx * heads + hh]));
    return bias;
}

}  // namespace

extern "C" int vsc_swin_create(const vsc_swin_config *cfg, vsc_swin **out) {
    VSC_REQUIRE(cfg && out, "swin_create: null argument");
    const vsc_swin_config &c = *cfg;
    VSC_REQUIRE(c.stages >= 1 && c.stages <= 4, "swin: stages %d", c.stages);
    VSC_REQUIRE(c.image_size % (c.patch_size << (c.stages - 1)) == 0, "swin: image %d / patch %d over %d stages",
                c.image_size, c.patch_size, c.stages);
    VSC_REQUIRE(c.image_size % 4 == 0, "swin: image size must be a multiple of 4");
    VSC_REQUIRE(c.embed_dim % 64 == 0, "swin: embed_dim %d must be a multiple of 64", c.embed_dim);
    VSC_REQUIRE(c.mlp_ratio == 4, "swin: mlp_ratio %d (only 4)", c.mlp_ratio);
    VSC_REQUIRE(c.max_batch >= 1 && c.out_dim >= 1 && c.out_dim <= 2048, "swin: max_batch / out_dim");
    vsc_swin *e = new vsc_swin();
    e->cfg = c;
    for (int s = 0; s < c.stages; ++s) {
        if (e->dim(s) != c.heads[s] * 32) {
            vsc_set_error("swin: stage %d width %d with %d heads -- only head_dim 32 is supported", s, e->dim(s),
                          c.heads[s]);
            delete e;
            return VSC_ERR_INVALID;
        }
        const int w = e->window(s);
        if (!((w == 8 || w == 16 || w == 12 || w == 24) && e->res(s) % w == 0) || c.depths[s] < 1) {
            vsc_set_error("swin: stage %d window %d on a %d x %d map unsupported (8, 12, 16 or 24)", s, w, e->res(s), e->res(s));
            delete e;
            return VSC_ERR_INVALID;
        }
    }
    if (e->dim(c.stages - 1) > 2048) {
        vsc_set_error("swin: last-stage width %d > 2048", e->dim(c.stages - 1));
        delete e;
        return VSC_ERR_INVALID;
    }
    const int kp = c.channels * c.patch_size * c.patch_size;
    e->kpad = (kp + 63) / 64 * 64;
    const size_t C0 = c.embed_dim;
    e->expect["patch_embed.proj.weight"] = C0 * kp;
    e->expect["patch_embed.proj.bias"] = e->expect["patch_embed.norm.weight"] = e->expect["patch_embed.norm.bias"] = C0;
    for (int s = 0; s < c.stages; ++s) {
        const size_t C = e->dim(s), H = c.heads[s];
        for (int b = 0; b < c.depths[s]; ++b) {
            const std::string p = "layers." + std::to_string(s) + ".blocks." + std::to_string(b) + ".";
            e->expect[p + "attn.qkv.weight"] = 3 * C * C;
            e->expect[p + "attn.q_bias"] = e->expect[p + "attn.v_bias"] = C;
            e->expect[p + "attn.logit_scale"] = H;
            e->expect[p + "attn.cpb_mlp.0.weight"] = 1024;
            e->expect[p + "attn.cpb_mlp.0.bias"] = 512;
            e->expect[p + "attn.cpb_mlp.2.weight"] = H * 512;
            e->expect[p + "attn.proj.weight"] = C * C;
            e->expect[p + "attn.proj.bias"] = C;
            e->expect[p + "norm1.weight"] = e->expect[p + "norm1.bias"] = C;
            e->expect[p + "norm2.weight"] = e->expect[p + "norm2.bias"] = C;
            e->expect[p + "mlp.fc1.weight"] = 4 * C * C;
            e->expect[p + "mlp.fc1.bias"] = 4 * C;
            e->expect[p + "mlp.fc2.weight"] = 4 * C * C;
            e->expect[p + "mlp.fc2.bias"] = C;
        }
        if (s + 1 < c.stages) {
            const std::string p = "layers." + std::to_string(s) + ".downsample.";
            e->expect[p + "reduction.weight"] = 8 * C * C;
            e->expect[p + "norm.weight"] = e->expect[p + "norm.bias"] = 2 * C;
        }
    }
    const size_t CL = e->dim(c.stages - 1);
    e->expect["norm.weight"] = e->expect["norm.bias"] = CL;
    e->expect["output_proj.weight"] = (size_t)c.out_dim * CL;
    e->expect["output_proj.bias"] = c.out_dim;
    *out = e;
    return VSC_OK;
}

extern "C" void vsc_swin_destroy(vsc_swin *e) {
    if (!e) return;
    for (int l = 0; l < 2; ++l)
        if (e->ws[l].lnws) gemm_ln_workspace_forget(e->ws[l].lnws);   // its flag counters die with the buffer
    for (void *p : e->allocs) (void)hipFree(p);
    for (int l = 0; l < 2; ++l) {
        if (e->lane_stream[l]) (void)hipStreamDestroy(e->lane_stream[l]);
        if (e->ev_join[l]) (void)hipEventDestroy(e->ev_join[l]);
    }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
    delete e;
}

extern "C" int vsc_swin_set_weight(vsc_swin *e, const char *name, const float *host, size_t count) {
    VSC_REQUIRE(e && name && host, "swin set_weight: null argument");
    if (e->finalized) {
        vsc_set_error("swin set_weight(%s) after finalize", name);
        return VSC_ERR_STATE;
    }
    auto it = e->expect.find(name);
    VSC_REQUIRE(it != e->expect.end(), "swin set_weight: unknown tensor '%s' for this config", name);
    VSC_REQUIRE(it->second == count, "swin set_weight: '%s' has %zu elements, expected %zu", name, count, it->second);
    e->host_w[name].assign(host, host + count);
    return VSC_OK;
}

// one lane's workspace (sizes fixed by finalize)
static int swin_alloc_workspace(vsc_swin *e, int l) {
    vsc_swin::Workspace &w = e->ws[l];
    void **dst[] = {(void **)&w.patches, (void **)&w.x, (void **)&w.xb, (void **)&w.t, (void **)&w.qkv,
                    (void **)&w.att, (void **)&w.h, (void **)&w.merged, (void **)&w.pooled};
    // idempotent per buffer: a call that failed part-way is resumed, not repeated (nothing is allocated or counted twice)
    for (int i = 0; i < 9; ++i) {
        if (*dst[i]) continue;
        int rc = sw_alloc(e, e->ws_sizes[i], dst[i]);
        if (rc) return rc;
        e->ws_bytes += (int64_t)e->ws_sizes[i];
    }
    if (!w.lnws) {
        int rc = sw_alloc(e, VSC_GEMM_LN_WS_BYTES, &w.lnws);
        if (rc) return rc;
        e->ws_bytes += (int64_t)VSC_GEMM_LN_WS_BYTES;
    }
    return VSC_OK;
}
// The two lanes (second workspace, two internal streams, fork / join events) exist from the first call with more than one
// chunk on: callers that never exceed max_batch per call hold one workspace only.
static int swin_make_lanes(vsc_swin *e) {
    if (e->lanes_ready) return VSC_OK;
    int rc = swin_alloc_workspace(e, 1);
    if (rc) return rc;
    for (int l = 0; l < 2; ++l) {
        if (!e->lane_stream[l]) VSC_CHECK_HIP(hipStreamCreateWithFlags(&e->lane_stream[l], hipStreamNonBlocking));
        if (!e->ev_join[l]) VSC_CHECK_HIP(hipEventCreateWithFlags(&e->ev_join[l], hipEventDisableTiming));
    }
    if (!e->ev_fork) VSC_CHECK_HIP(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    e->lanes_ready = true;
    return VSC_OK;
}

struct SwinProfScope {
    vsc_swin *e;
    hipStream_t st;
    size_t e0 = 0;
    int cls;
    SwinProfScope(vsc_swin *enc, int c, hipStream_t s) : e(enc), st(s), cls(c) {
        if (e->profile) e0 = rec();
    }
    ~SwinProfScope() {
        if (!e->profile || e0 == BAD) return;
        const size_t e1 = rec();
        if (e1 != BAD) e->spans.push_back({cls, e0, e1});
    }
    static constexpr size_t BAD = ~(size_t)0;
    // One event from the pool (recycled by get_profile / set_profiling).  A failed create / record, or a pool that has
    // grown past MAX_EVENTS because nobody collects the profile, switches profiling off instead of recording garbage.
    static constexpr size_t MAX_EVENTS = 1 << 16;
    size_t rec() {
        if (e->ev_used == e->ev_pool.size()) {
            hipEvent_t ev;
            if (e->ev_pool.size() >= MAX_EVENTS || hipEventCreate(&ev) != hipSuccess) {
                e->profile = false;
                return BAD;
            }
            e->ev_pool.push_back(ev);
        }
        if (hipEventRecord(e->ev_pool[e->ev_used], st) != hipSuccess) {
            e->profile = false;
            return BAD;
        }
        return e->ev_used++;
    }
};

extern "C" int vsc_swin_finalize(vsc_swin *e) {
    VSC_REQUIRE(e, "swin finalize: null");
    if (e->finalized) return VSC_OK;
    for (auto &kv : e->expect)
        if (!e->host_w.count(kv.first)) {
            vsc_set_error("swin finalize: weight '%s' was never set", kv.first.c_str());
            return VSC_ERR_STATE;
        }
    const vsc_swin_config &c = e->cfg;
    int rc;
#define TRY(x) do { if ((rc = (x))) return rc; } while (0)
    const int kp = c.channels * c.patch_size * c.patch_size;
    TRY(sw_upload_bf16(e, "patch_embed.proj.weight", c.embed_dim, kp, e->kpad, &e->pe_w));
    TRY(sw_upload_f32(e, "patch_embed.proj.bias", &e->pe_b));
    TRY(sw_upload_f32(e, "patch_embed.norm.weight", &e->pe_g));
    TRY(sw_upload_f32(e, "patch_embed.norm.bias", &e->pe_beta));
    e->stages.resize(c.stages);
    for (int s = 0; s < c.stages; ++s) {
        const int C = e->dim(s), H = c.heads[s], W = e->window(s);
        e->stages[s].blocks.resize(c.depths[s]);
        for (int b = 0; b < c.depths[s]; ++b) {
            const std::string p = "layers." + std::to_string(s) + ".blocks." + std::to_string(b) + ".";
            SwinBlockW &B = e->stages[s].blocks[b];
            TRY(sw_upload_bf16(e, p + "attn.qkv.weight", 3 * C, C, C, &B.qkv_w));
            std::vector<float> qb(3 * (size_t)C, 0.f);  // (q_bias | 0 | v_bias), :151-153
            const std::vector<float> &q = e->host_w.at(p + "attn.q_bias"), &v = e->host_w.at(p + "attn.v_bias");
            for (int i = 0; i < C; ++i) {
                qb[i] = q[i];
                qb[2 * C + i] = v[i];
            }
            TRY(sw_upload_f32v(e, qb, &B.qkv_b));
            std::vector<float> sc(H);
            const std::vector<float> &ls = e->host_w.at(p + "attn.logit_scale");
            for (int i = 0; i < H; ++i) sc[i] = expf(fminf(ls[i], logf(100.0f)));  // clamp(max = ln(1/0.01)).exp(), :161
            std::vector<float> pb = position_bias(e->host_w.at(p + "attn.cpb_mlp.0.weight"), e->host_w.at(p + "attn.cpb_mlp.0.bias"),
                                                  e->host_w.at(p + "attn.cpb_mlp.2.weight"), W, c.pretrained_window_sizes[s], H);
            // bounded softmax (vsc_window_attention_bf16): heads whose logits span <= 69 get their upper bound scale + max(bias)
            // folded into the table and a negative scale.  (Its probabilities can be as small as e^-69: they are packed to bf16 -- fp32's
            // exponent range -- in BOTH builds of the library; swin.hip "The P . V product runs on bf16 operands".)
            const size_t side2 = (size_t)(2 * W - 1) * (2 * W - 1);
            for (int hh = 0; hh < H && !vsc_opt(OPT_SWIN_ROW_MAX); ++hh) {
                float bmax = -INFINITY, bmin = INFINITY;
                for (size_t i = 0; i < side2; ++i) {
                    bmax = fmaxf(bmax, pb[hh * side2 + i]);
                    bmin = fminf(bmin, pb[hh * side2 + i]);
                }
                if (!(2.0f * sc[hh] + (bmax - bmin) <= 69.0f)) continue;   // (also skips NaN / inf)
                for (size_t i = 0; i < side2; ++i) pb[hh * side2 + i] -= bmax + sc[hh];
                sc[hh] = -sc[hh];
            }
            TRY(sw_upload_f32v(e, sc, &B.scale));
            TRY(sw_upload_f32v(e, pb, &B.bias));
            TRY(sw_upload_bf16(e, p + "attn.proj.weight", C, C, C, &B.proj_w));
            TRY(sw_upload_f32(e, p + "attn.proj.bias", &B.proj_b));
            TRY(sw_upload_f32(e, p + "norm1.weight", &B.n1_g));
            TRY(sw_upload_f32(e, p + "norm1.bias", &B.n1_b));
            TRY(sw_upload_bf16(e, p + "mlp.fc1.weight", 4 * C, C, C, &B.fc1_w));
            TRY(sw_upload_f32(e, p + "mlp.fc1.bias", &B.fc1_b));
            TRY(sw_upload_bf16(e, p + "mlp.fc2.weight", C, 4 * C, 4 * C, &B.fc2_w));
            if (swin_mlp_supported(C)) {
                const std::vector<float> &w2 = e->host_w.at(p + "mlp.fc2.weight");
                std::vector<float> w2p(w2.size());
                swin_mlp_permute_hidden(w2.data(), w2p.data(), C);
                TRY(sw_upload_bf16v(e, w2p, p + "mlp.fc2.weight (hidden axis reordered)", C, 4 * C, 4 * C, &B.fc2_wp));
            }
            TRY(sw_upload_f32(e, p + "mlp.fc2.bias", &B.fc2_b));
            TRY(sw_upload_f32(e, p + "norm2.weight", &B.n2_g));
            TRY(sw_upload_f32(e, p + "norm2.bias", &B.n2_b));
        }
        if (s + 1 < c.stages) {
            const std::string p = "layers." + std::to_string(s) + ".downsample.";
            TRY(sw_upload_bf16(e, p + "reduction.weight", 2 * C, 4 * C, 4 * C, &e->stages[s].red_w));
            TRY(sw_upload_f32(e, p + "norm.weight", &e->stages[s].dn_g));
            TRY(sw_upload_f32(e, p + "norm.bias", &e->stages[s].dn_b));
        }
    }
    TRY(sw_upload_f32(e, "norm.weight", &e->norm_g));
    TRY(sw_upload_f32(e, "norm.bias", &e->norm_b));
    TRY(sw_upload_f32(e, "output_proj.weight", &e->out_w));
    TRY(sw_upload_f32(e, "output_proj.bias", &e->out_b));
    const size_t B = c.max_batch, M0 = B * e->res(0) * e->res(0), MC = M0 * c.embed_dim;
    const size_t sz[] = {M0 * (size_t)e->kpad * 2, MC * 4, MC * 2, MC * 4, MC * 3 * 2, MC * 2, MC * 4 * 2, MC * 2,
                         B * (size_t)e->dim(c.stages - 1) * 4};
    for (int i = 0; i < 9; ++i) e->ws_sizes[i] = sz[i];
    TRY(swin_alloc_workspace(e, 0));
#undef TRY
    e->host_w.clear();
    e->finalized = true;
    return VSC_OK;
}

extern "C" int64_t vsc_swin_workspace_bytes(const vsc_swin *e) { return e ? e->ws_bytes : 0; }

// x_out = (x_in ? x_in : 0) + LayerNorm(A W^T + bias): one row-owning GEMM when a tile can hold the whole row
// (widths 128/256/512), otherwise GEMM to fp32 scratch + the row kernel (width 1024: the last stage).
static int gemm_ln(vsc_swin *e, vsc_swin::Workspace &ws, const uint16_t *a, const uint16_t *w, const float *bias, const float *g, const float *b,
                   const float *x_in, float *x_out, uint16_t *xb_out, int64_t m, int n, int k, hipStream_t st) {
    const bool split = vsc_opt(OPT_SWIN_SPLIT_LN) != nullptr;
    const char *split_k_opt = vsc_opt(OPT_SWIN_SPLIT_K);
    const int split_k = split_k_opt ? atoi(split_k_opt) : 1 << 30;
    if (!split && k < split_k && gemm_ln_supported(n, k))
        return launch_gemm_ln_bf16(a, w, bias, g, b, x_in, x_out, xb_out, m, n, k, e->cfg.ln_eps, st, ws.lnws);
    int rc = launch_gemm_bf16(a, w, bias, nullptr, ws.t, m, n, k, VSC_EPI_F32, 0, st);
    if (rc) return rc;
    return launch_ln_residual(ws.t, g, b, x_in, x_out, xb_out, m, n, e->cfg.ln_eps, st);
}

// the chunks of one call; `fork`: alternate them over the two lanes.  Returns at the first failing launch (the caller joins
// the lanes in every case).
static int swin_run_chunks(vsc_swin *e, const float *frames, const uint8_t *frames_u8, const float *mean, const float *std,
                           int64_t n, float *desc, float *tokens_out, hipStream_t user, bool fork) {
    const vsc_swin_config &c = e->cfg;
    const int64_t frame_elems = (int64_t)c.channels * c.image_size * c.image_size;
    const int SL = c.stages - 1, TL = e->res(SL) * e->res(SL), CL = e->dim(SL);
    int rc;
#define TRY(x) do { if ((rc = (x))) return rc; } while (0)
#define PROF(cls) SwinProfScope _ps(e, (cls), st)
    const char *fm = vsc_opt(OPT_SWIN_FUSED_MLP);   // diagnostic / test switch: 0 = fc1 and fc2 as two GEMM launches
    const bool unfused_mlp = fm && fm[0] == '0';
    const char *f5 = vsc_opt(OPT_SWIN_MLP512);      // diagnostic / test switch: 0 = the 512-wide stage keeps fc1 and fc2 as two GEMM launches,
    const bool unfused_mlp512 = f5 && f5[0] == '0';  // 1 = the fused kernel at every size (default: where its 128-row tiles fill the chip)
    const bool forced_mlp512 = f5 && f5[0] == '1';
    int cus512 = 256, dev512 = 0;
    if (hipGetDevice(&dev512) != hipSuccess || hipDeviceGetAttribute(&cus512, hipDeviceAttributeMultiprocessorCount, dev512) != hipSuccess || cus512 <= 0)
        cus512 = 256;
    const char *f6 = vsc_opt(OPT_SWIN_PROJ512);     // diagnostic / test switch: 0 = the 512-wide stage keeps proj + LayerNorm as their own launch
    const bool unfused_proj512 = f6 && f6[0] == '0';
    const char *f7 = vsc_opt(OPT_SWIN_QKV512);      // diagnostic / test switch: 0 = every block of the 512-wide stage launches its own qkv GEMM
    const bool unfused_qkv512 = f7 && f7[0] == '0';
    const char *fp = vsc_opt(OPT_SWIN_FUSED_PROJ);   // diagnostic / test switch: 0 = proj + LayerNorm as their own launch
    const bool unfused_proj = fp && fp[0] == '0';
    const char *fg = vsc_opt(OPT_SWIN_FUSED_MERGE);   // diagnostic / test switch: 0 = PatchMerging as a gather kernel + GEMM
    const bool unfused_merge = fg && fg[0] == '0';
    int chunk = 0;
    for (int64_t off = 0; off < n; off += c.max_batch, ++chunk) {
        const int lane = fork ? (chunk & 1) : 0;
        hipStream_t st = fork ? e->lane_stream[lane] : user;
        vsc_swin::Workspace &w = e->ws[lane];
        const int64_t B = (n - off) < c.max_batch ? (n - off) : c.max_batch;
        int64_t M = B * e->res(0) * e->res(0);
        {
            PROF(VSC_SWIN_PROF_PATCHIFY);
            if (frames)
                TRY(launch_patchify(frames + off * frame_elems, w.patches, B, c.channels, c.image_size, c.patch_size, e->kpad, st));
            else
                TRY(launch_patchify_u8(frames_u8 + off * frame_elems, w.patches, B, c.channels, c.image_size, c.patch_size,
                                       e->kpad, mean, std, st));
        }
        { PROF(VSC_SWIN_PROF_PATCH_EMBED); TRY(gemm_ln(e, w, w.patches, e->pe_w, e->pe_b, e->pe_g, e->pe_beta, nullptr, w.x, w.xb, M, c.embed_dim, e->kpad, st)); }
        for (int s = 0; s < c.stages; ++s) {
            const int C = e->dim(s), R = e->res(s), W = e->window(s), H = c.heads[s];
            const int pc = VSC_SWIN_PROF_STAGE0 + (s < 4 ? s : 3) * VSC_SWIN_PROF_PER_STAGE;
            M = B * R * R;
            const int64_t Bs = B, Ms = M;
            float *x = w.x;
            uint16_t *xb = w.xb;
            bool qkv_ready = false;   // the previous block's kernel already left this block's qkv in w.qkv
            for (int b = 0; b < c.depths[s]; ++b) {
                const SwinBlockW &K = e->stages[s].blocks[b];
                if (!qkv_ready) { PROF(pc + VSC_SWIN_PROF_QKV); TRY(launch_gemm_bf16(xb, K.qkv_w, K.qkv_b, nullptr, w.qkv, Ms, 3 * C, C, VSC_EPI_BF16, 0, st)); }
                qkv_ready = false;
                { PROF(pc + VSC_SWIN_PROF_ATTENTION); TRY(launch_window_attention(w.qkv, w.att, K.bias, K.scale, (int)Bs, R, W, e->shift(s, b), H, st)); }
                // (the 512-wide kernel addresses x through one 4-GiB buffer descriptor: chunks of >= 2^21 rows keep the GEMM launches)
                // ... and small chunks too: the fused kernel gives a CU one 128-row tile at a time (one workgroup per CU, ~200 us per tile
                // whatever else the chip does), so 40 frames = 80 tiles keep 80 of 256 CUs busy where the GEMMs' 256 x 256 tiles also split
                // the output width: below 0.6 of a whole number of rounds the launches win (8 frames 2.63 vs 1.96 k frames/s, 40 frames 8.5 vs
                // 7.3 k, 64 frames 11.1 vs 10.6 k; 96 frames 12.5 vs 13.6 k, 128 frames 13.6 vs 15.8 k: tools/micro/swin_small_batch_ab.sh)
                const int64_t tiles512 = (Ms + 127) / 128, rounds512 = (tiles512 + cus512 - 1) / cus512;
                const bool fills512 = forced_mlp512 || 10 * tiles512 >= 6 * rounds512 * cus512;
                const bool mlp512_ok = C != 512 || (!unfused_mlp512 && fills512 && Ms < (1ll << 21));
                if (K.fc2_wp && !unfused_mlp && !unfused_proj && swin_proj_mlp_supported(C) && mlp512_ok && !(C == 512 && unfused_proj512)) {
                    // the whole second half of the block -- proj, LayerNorm, residual, MLP, LayerNorm, residual -- in one kernel; its time
                    // is booked under fc2_ln, proj_ln and fc1 stay empty
                    PROF(pc + VSC_SWIN_PROF_FC2_LN);
                    if (C == 512 && b + 1 < c.depths[s] && !unfused_qkv512 && Ms * 3072 < (1ll << 32)) {
                        // ... and the NEXT block's qkv Linear behind it, from the registers that hold the new shadow: the shadow is not
                        // written (the stage's last block writes it for the PatchMerging), the next qkv launch does not happen
                        const SwinBlockW &N = e->stages[s].blocks[b + 1];
                        TRY(launch_swin_proj_mlp_qkv512(w.att, K.proj_w, K.proj_b, K.n1_g, K.n1_b, K.fc1_w, K.fc1_b, K.fc2_wp, K.fc2_b, K.n2_g, K.n2_b,
                                                        N.qkv_w, N.qkv_b, x, w.qkv, Ms, e->cfg.ln_eps, st));
                        qkv_ready = true;
                        continue;
                    }
                    TRY(launch_swin_proj_mlp(w.att, K.proj_w, K.proj_b, K.n1_g, K.n1_b, K.fc1_w, K.fc1_b, K.fc2_wp, K.fc2_b, K.n2_g, K.n2_b, x, xb,
                                             Ms, C, e->cfg.ln_eps, st));
                    continue;
                }
                { PROF(pc + VSC_SWIN_PROF_PROJ_LN); TRY(gemm_ln(e, w, w.att, K.proj_w, K.proj_b, K.n1_g, K.n1_b, x, x, xb, Ms, C, C, st)); }
                if (K.fc2_wp && !unfused_mlp && mlp512_ok) {
                    // both Linears, the GELU between them and the LayerNorm behind them in one kernel (swin_mlp.hip); its time is
                    // booked under fc2_ln, fc1 stays empty
                    PROF(pc + VSC_SWIN_PROF_FC2_LN);
                    TRY(launch_swin_mlp(K.fc1_w, K.fc1_b, K.fc2_wp, K.fc2_b, K.n2_g, K.n2_b, x, xb, Ms, C, e->cfg.ln_eps, st));
                } else {
                    { PROF(pc + VSC_SWIN_PROF_FC1); TRY(launch_gemm_bf16(xb, K.fc1_w, K.fc1_b, nullptr, w.h, Ms, 4 * C, C, VSC_EPI_GELU_BF16, 0, st)); }
                    { PROF(pc + VSC_SWIN_PROF_FC2_LN); TRY(gemm_ln(e, w, w.h, K.fc2_w, K.fc2_b, K.n2_g, K.n2_b, x, x, xb, Ms, C, 4 * C, st)); }
                }
            }
            if (s + 1 < c.stages) {
                PROF(pc + VSC_SWIN_PROF_MERGE);
                if (!unfused_merge && gemm_ln_supported(2 * C, 4 * C) && (R & (R - 1)) == 0 && C % 32 == 0) {
                    // the 2 x 2 gather inside the GEMM's operand staging (no [M/4, 4C] copy out and back in).  The shadow of the
                    // merged tokens cannot overwrite the tensor other workgroups are still gathering from: it goes to the
                    // buffer the copy used to fill, and the two (equally sized) buffers change roles.
                    TRY(launch_gemm_ln_bf16(w.xb, e->stages[s].red_w, nullptr, e->stages[s].dn_g, e->stages[s].dn_b, nullptr, w.x, w.merged,
                                            M / 4, 2 * C, 4 * C, e->cfg.ln_eps, st, w.lnws, R, C));
                    uint16_t *t = w.xb;
                    w.xb = w.merged;
                    w.merged = t;
                } else {
                    TRY(launch_merge_gather(w.xb, w.merged, B, R, C, st));
                    TRY(gemm_ln(e, w, w.merged, e->stages[s].red_w, nullptr, e->stages[s].dn_g, e->stages[s].dn_b, nullptr, w.x, w.xb,
                                M / 4, 2 * C, 4 * C, st));
                }
            }
        }
        {
            PROF(VSC_SWIN_PROF_POOL_HEAD);
            TRY(launch_ln_pool(w.x, e->norm_g, e->norm_b, w.pooled, tokens_out ? tokens_out + off * TL * CL : nullptr, B,
                               TL, CL, c.ln_eps, 0, c.gem_p, st));
            TRY(launch_head(w.pooled, e->out_w, e->out_b, desc + off * c.out_dim, B, CL, c.out_dim, c.l2_normalize, st));
        }
    }
#undef PROF
#undef TRY
    return VSC_OK;
}

static int swin_forward_impl(vsc_swin *e, const float *frames, const uint8_t *frames_u8, const float *mean, const float *std,
                             int64_t n, float *desc, float *tokens_out, void *stream_) {
    VSC_REQUIRE(e && (frames || frames_u8) && desc && n >= 0, "swin forward: bad argument");
    if (!e->finalized) {
        vsc_set_error("swin forward before finalize");
        return VSC_ERR_STATE;
    }
    hipStream_t user = (hipStream_t)stream_;
    // >= 2 chunks: alternate them over the two lanes.  Not while profiling: per-launch events are meant to time one kernel alone.
    const bool fork = n > e->cfg.max_batch && !e->profile;
    if (fork) {
        int rc = swin_make_lanes(e);
        if (rc) return rc;
        VSC_CHECK_HIP(hipEventRecord(e->ev_fork, user));
        for (int l = 0; l < 2; ++l) VSC_CHECK_HIP(hipStreamWaitEvent(e->lane_stream[l], e->ev_fork, 0));
    }
    const int rc = swin_run_chunks(e, frames, frames_u8, mean, std, n, desc, tokens_out, user, fork);
    if (fork) {
        // also after a failed launch: whatever the lanes already hold is ordered before the caller's next work on `user`,
        // so the caller may free or reuse frames / desc once its stream has drained
        for (int l = 0; l < 2; ++l) {
            const hipError_t e1 = hipEventRecord(e->ev_join[l], e->lane_stream[l]);
            const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(user, e->ev_join[l], 0) : e1;
            if (e2 != hipSuccess && !rc) {
                vsc_set_error("swin forward: joining lane %d failed: %s", l, hipGetErrorString(e2));
                return VSC_ERR_HIP;
            }
        }
    }
    return rc;
}

extern "C" int vsc_swin_set_profiling(vsc_swin *e, int32_t on) {
    VSC_REQUIRE(e, "swin set_profiling: null encoder");
    e->profile = on != 0;
    e->spans.clear();
    e->ev_used = 0;
    for (int i = 0; i < VSC_SWIN_PROF_CLASSES; ++i) {
        e->prof_ms[i] = 0;
        e->prof_n[i] = 0;
    }
    return VSC_OK;
}

extern "C" int vsc_swin_get_profile(vsc_swin *e, double *ms_out, int64_t *launches_out) {
    VSC_REQUIRE(e && ms_out && launches_out, "swin get_profile: null argument");
    VSC_CHECK_HIP(hipDeviceSynchronize());
    for (const vsc_swin::Span &sp : e->spans) {
        float ms = 0.f;
        VSC_CHECK_HIP(hipEventElapsedTime(&ms, e->ev_pool[sp.e0], e->ev_pool[sp.e1]));
        e->prof_ms[sp.cls] += ms;
        e->prof_n[sp.cls] += 1;
    }
    e->spans.clear();
    e->ev_used = 0;
    for (int i = 0; i < VSC_SWIN_PROF_CLASSES; ++i) {
        ms_out[i] = e->prof_ms[i];
        launches_out[i] = e->prof_n[i];
    }
    return VSC_OK;
}

extern "C" int vsc_swin_forward_debug(vsc_swin *e, const float *frames, int64_t n, float *desc, float *tokens_out,
                                      void *stream) {
    VSC_REQUIRE(frames, "swin forward: null frames");
    return swin_forward_impl(e, frames, nullptr, nullptr, nullptr, n, desc, tokens_out, stream);
}

extern "C" int vsc_swin_forward(vsc_swin *e, const float *frames, int64_t n, float *desc, void *stream) {
    return vsc_swin_forward_debug(e, frames, n, desc, nullptr, stream);
}

extern "C" int vsc_swin_forward_u8(vsc_swin *e, const uint8_t *frames_u8, int64_t n, const float *mean, const float *std,
                                   float *desc, void *stream) {
    VSC_REQUIRE(frames_u8 && mean && std, "swin forward_u8: null argument");
    return swin_forward_impl(e, nullptr, frames_u8, mean, std, n, desc, nullptr, stream);
}
