// C-ABI glue: error state, device probe, and the per-kernel entry points of include/vsc_hip.h.
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void vsc_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *vsc_last_error(void) { return g_err; }

#ifndef VSC_SRC_HASH
#define VSC_SRC_HASH "unhashed"
#endif
// "... src <hash>": first 16 hex digits of the SHA-256 over the sources the Makefile lists (HASHED) -- tests/test_capi_symbols.py recomputes it
extern "C" const char *vsc_version(void) { return "vsc_hip 0.1 (gfx950) src " VSC_SRC_HASH; }
// "bf16" (libvsc_hip.so) or "fp16" (libvsc_hip_f16.so, built with -DVSC_OPERAND_F16): the 16-bit type of every MFMA operand of the encoders
extern "C" const char *vsc_operand_dtype(void) { return VSC_LP_NAME; }

extern "C" int vsc_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        vsc_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return VSC_ERR_NO_DEVICE;
    }
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) == hipSuccess && strstr(prop.gcnArchName, "gfx950")) ++ok;
    }
    return ok;
}

// ---- diagnostic / test switches (common.h: VSC_OPT_LIST) -------------------------------------------------------------
#include <atomic>
#include <mutex>
#include <stdlib.h>
static const char *const g_opt_names[OPT_COUNT] = {
#define X(n) "VSC_" #n,
    VSC_OPT_LIST(X)
#undef X
};
static std::atomic<const char *> g_opt_values[OPT_COUNT];
static std::once_flag g_opt_once;
static void opt_load_env() {
    for (int i = 0; i < OPT_COUNT; ++i) {
        const char *e = getenv(g_opt_names[i]);   // the only getenv calls of the library: once per process
        g_opt_values[i].store(e ? strdup(e) : nullptr, std::memory_order_relaxed);
    }
}
const char *vsc_opt(VscOpt o) {
    std::call_once(g_opt_once, opt_load_env);
    return g_opt_values[o].load(std::memory_order_acquire);
}
// value == NULL (or "") clears the switch.  Superseded strings are not freed: a reader may still hold them, and switches
// change a handful of times per process (tests, A/B tools).
extern "C" int vsc_set_option(const char *name, const char *value) {
    VSC_REQUIRE(name, "set_option: null name");
    std::call_once(g_opt_once, opt_load_env);
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(name, g_opt_names[i]) || !strcmp(name, g_opt_names[i] + 4)) {
            g_opt_values[i].store(value && value[0] ? strdup(value) : nullptr, std::memory_order_release);
            return VSC_OK;
        }
    vsc_set_error("set_option: unknown switch '%s'", name);
    return VSC_ERR_INVALID;
}

// current value of a switch (nullptr when unset or unknown): lets a caller save and restore around a temporary change
extern "C" const char *vsc_get_option(const char *name) {
    if (!name) return nullptr;
    std::call_once(g_opt_once, opt_load_env);
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(name, g_opt_names[i]) || !strcmp(name, g_opt_names[i] + 4))
            return g_opt_values[i].load(std::memory_order_acquire);
    return nullptr;
}

extern "C" int vsc_gemm_bf16(const uint16_t *a, const uint16_t *w, const float *bias, const float *aux,
                             void *out, int64_t m, int32_t n, int32_t k, int32_t epilogue,
                             int32_t tokens, void *stream) {
    return launch_gemm_bf16(a, w, bias, aux, out, m, n, k, epilogue, tokens, (hipStream_t)stream);
}

extern "C" int vsc_attention_bf16(const uint16_t *qkv, uint16_t *out, int32_t frames, int32_t tokens,
                                  int32_t heads, void *stream) {
    return launch_attention_bf16(qkv, out, frames, tokens, heads, (hipStream_t)stream);
}

extern "C" int vsc_layernorm_f32(const float *x, const float *g, const float *b, void *out,
                                 int64_t rows, int32_t width, float eps, int32_t out_f32,
                                 void *stream) {
    return launch_layernorm(x, g, b, out, rows, width, eps, out_f32, (hipStream_t)stream);
}

extern "C" int vsc_patchify_bf16(const float *frames, uint16_t *patches, int64_t n, int32_t channels,
                                 int32_t image, int32_t patch, int32_t kpad, void *stream) {
    return launch_patchify(frames, patches, n, channels, image, patch, kpad, (hipStream_t)stream);
}

extern "C" int vsc_l2_normalize_f32(float *x, int64_t n, int32_t d, void *stream) {
    return launch_l2_normalize(x, n, d, (hipStream_t)stream);
}

extern "C" int vsc_window_attention_bf16(const uint16_t *qkv, uint16_t *out, const float *bias, const float *scale,
                                         int32_t frames, int32_t res, int32_t window, int32_t shift,
                                         int32_t heads, void *stream) {
    return launch_window_attention(qkv, out, bias, scale, frames, res, window, shift, heads, (hipStream_t)stream);
}

extern "C" int vsc_ln_residual_f32(const float *t, const float *g, const float *b, const float *x_in, float *x_out,
                                   uint16_t *xb, int64_t rows, int32_t width, float eps, void *stream) {
    return launch_ln_residual(t, g, b, x_in, x_out, xb, rows, width, eps, (hipStream_t)stream);
}

extern "C" int vsc_gemm_ln_bf16(const uint16_t *a, const uint16_t *w, const float *bias, const float *g, const float *b,
                                const float *x_in, float *x_out, uint16_t *xb, int64_t m, int32_t n, int32_t k,
                                float eps, void *stream) {
    return launch_gemm_ln_bf16(a, w, bias, g, b, x_in, x_out, xb, m, n, k, eps, (hipStream_t)stream);
}

extern "C" int vsc_swin_mlp_bf16(const uint16_t *w1, const float *b1, const uint16_t *w2p, const float *b2, const float *g, const float *b,
                                 float *x, uint16_t *xb, int64_t m, int32_t c, float eps, void *stream) {
    return launch_swin_mlp(w1, b1, w2p, b2, g, b, x, xb, m, c, eps, (hipStream_t)stream);
}

extern "C" int vsc_swin_proj_mlp_bf16(const uint16_t *att, const uint16_t *wp, const float *bp, const float *g1, const float *be1, const uint16_t *w1,
                                      const float *b1, const uint16_t *w2p, const float *b2, const float *g2, const float *be2, float *x, uint16_t *xb,
                                      int64_t m, int32_t c, float eps, void *stream) {
    return launch_swin_proj_mlp(att, wp, bp, g1, be1, w1, b1, w2p, b2, g2, be2, x, xb, m, c, eps, (hipStream_t)stream);
}

extern "C" int vsc_swin_proj_mlp_qkv_bf16(const uint16_t *att, const uint16_t *wp, const float *bp, const float *g1, const float *be1, const uint16_t *w1,
                                          const float *b1, const uint16_t *w2p, const float *b2, const float *g2, const float *be2, const uint16_t *wq,
                                          const float *bq, float *x, uint16_t *qkv_next, int64_t m, int32_t c, float eps, void *stream) {
    VSC_REQUIRE(c == 512, "swin_proj_mlp_qkv: width %d unsupported (512)", c);
    return launch_swin_proj_mlp_qkv512(att, wp, bp, g1, be1, w1, b1, w2p, b2, g2, be2, wq, bq, x, qkv_next, m, eps, (hipStream_t)stream);
}

extern "C" int vsc_swin_mlp_permute_hidden_f32(const float *w2, float *w2p, int32_t c) {
    VSC_REQUIRE(w2 && w2p && w2 != w2p, "swin_mlp_permute_hidden: null / aliased arrays");
    VSC_REQUIRE(swin_mlp_supported(c), "swin_mlp_permute_hidden: width %d unsupported (128, 256 or 512)", c);
    swin_mlp_permute_hidden(w2, w2p, c);
    return VSC_OK;
}

extern "C" int vsc_debug_mlp512_timing(uint32_t *buf_dev) {
    swin_mlp512_set_timing_buffer(buf_dev);
    return VSC_OK;
}

extern "C" int vsc_merge_gather_bf16(const uint16_t *xb, uint16_t *out, int64_t frames, int32_t res, int32_t c,
                                     void *stream) {
    return launch_merge_gather(xb, out, frames, res, c, (hipStream_t)stream);
}

// Measurement aid (tools/micro/clock_under_load.py): one wave spins for `ticks` s_memtime ticks (= shader cycles) and
// stores the elapsed count; timed from the host with events it gives the shader clock while other streams are busy.
__global__ void spin_ticks_kernel(unsigned long long ticks, unsigned long long *out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t = t0;
    while (t - t0 < ticks) {
        __builtin_amdgcn_s_sleep(16);
        t = __builtin_amdgcn_s_memtime();
    }
    if (threadIdx.x == 0) *out = t - t0;
}

extern "C" int vsc_debug_spin_ticks(uint64_t ticks, uint64_t *out_dev, void *stream) {
    VSC_REQUIRE(out_dev, "spin: null pointer");
    hipLaunchKernelGGL(spin_ticks_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)ticks,
                       (unsigned long long *)out_dev);
    VSC_CHECK_LAUNCH();
    return VSC_OK;
}
