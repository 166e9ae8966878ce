/* vsc_hip.h -- C ABI of libvsc_hip.so, the MI355X (gfx950) hot path of the VSC22
 * descriptor track.  Plain pointers and sizes only; every pointer named *_dev is
 * device memory on the current HIP device, `stream` is a hipStream_t passed as
 * void* (NULL = default stream).  Every entry point returns 0 on success and a
 * negative vsc_status otherwise; vsc_last_error() gives the message of the last
 * failure on the calling thread.  Nothing here falls back to the CPU.
 *
 * Concurrency: one process per GPU is the deployment model.  An encoder handle, and the
 * search entry points as a group (vsc_knn_ip_f32, vsc_range_search_ip_f32, vsc_pair_similarity_f32,
 * vsc_video_pair_max_f32 share grow-only device scratch), must not be driven from two host threads at once, and
 * consecutive calls that share a handle or that scratch must be stream-ordered (same stream, or
 * ordered by events).  Different encoder handles are independent.
 *
 * Each entry point names the reference interface it replaces
 * (paths relative to /root/reference/VSC22-Descriptor-Track-1st).
 */
#ifndef VSC_HIP_H
#define VSC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vsc_status {
    VSC_OK = 0,
    VSC_ERR_INVALID = -1,     /* bad argument / unsupported shape */
    VSC_ERR_HIP = -2,         /* a HIP runtime call failed */
    VSC_ERR_NO_DEVICE = -3,   /* no gfx950 device visible */
    VSC_ERR_STATE = -4,       /* call order (e.g. forward before finalize) */
    VSC_ERR_NOMEM = -5
} vsc_status;

const char *vsc_last_error(void);
/* Number of visible HIP devices whose arch is gfx950; <0 on runtime failure. */
int vsc_device_count(void);
const char *vsc_version(void);
/* The 16-bit operand type of this build of the library: "bf16" (libvsc_hip.so, the configuration BASELINE.json names) or "fp16"
 * (libvsc_hip_f16.so: same kernels, same MFMA rate and bytes, 11 significand bits instead of 8; the infer/ entry points default to it
 * because the end-to-end uAP parity needs it).  Every `uint16_t *` tensor of the `*_bf16` kernel-level entry points
 * below holds THIS type's bit patterns; the entry points keep their names in both builds.  The similarity search and the fp32
 * convolutions are identical in both libraries (their bf16 stages are bf16 by construction). */
const char *vsc_operand_dtype(void);
/* Diagnostic / test switches (path forcing for the parity tests, A/B knobs of tools/micro).  Each switch VSC_<NAME> takes
 * its initial value from the environment variable of the same name, read ONCE per process; after that it changes only
 * through this call (name with or without the VSC_ prefix; value NULL or "" clears it).  No entry point reads the
 * environment on its launch path.  Unknown names return VSC_ERR_INVALID.  Not a reference interface: the reference has no
 * counterpart. */
int vsc_set_option(const char *name, const char *value);
/* Current value of a switch (NULL when unset or unknown); the string stays valid for the life of the process.  For callers
 * that change a switch temporarily and must restore what the environment or an enclosing scope had set. */
const char *vsc_get_option(const char *name);

/* ------------------------------------------------------------------------ *
 * Frame encoder: replaces `flat_features = model(flat_frames)` on the
 * TorchScript backbones -- infer/src/extractor.py:23, infer/extract_query_feats.py:150
 * (single_infer) and :171 (clip_model) -- for ViT-family backbones
 * (train/train_v115/vsc/baseline/model_factory/backbones/vit.py:10-54,
 *  train/train_vid_score/video/clip.py:82-161).
 * ------------------------------------------------------------------------ */
typedef struct vsc_encoder vsc_encoder;

typedef struct vsc_encoder_config {
    int32_t image_size;   /* square input, pixels */
    int32_t patch_size;
    int32_t channels;     /* 3 */
    int32_t width;        /* multiple of 64; head_dim must be 64 */
    int32_t layers;
    int32_t heads;
    int32_t mlp_dim;      /* multiple of 64 */
    int32_t out_dim;      /* Linear head outputs; 0 = emit the pooled feature */
    float ln_eps;
    int32_t act;          /* 0 = exact GELU (HF/timm), 1 = QuickGELU (clip.py:22) */
    int32_t pre_ln;       /* 1 = CLIP ln_pre after the position embedding */
    int32_t patch_bias;   /* 0 = bias-free patch conv (CLIP) */
    int32_t pool;         /* 0 = GeM over all tokens (vit.py:52), 1 = CLS token */
    float gem_p;
    int32_t max_batch;    /* frames per internal step; workspace is sized for it */
    int32_t l2_normalize; /* 1 = emit sklearn-style L2-normalised descriptors */
    int32_t head_conv_dim; /* >0: SSCD head (sscd.py:25-42): tokens -> Conv1d(width, head_conv_dim, 1)
                              -> GeM over tokens -> Linear(head_conv_dim, out_dim); needs pool = 0 */
    int32_t lanes;        /* 2: a forward call of more than max_batch frames alternates its chunks over
                             two internal streams (two workspaces) so memory-bound kernels of one
                             chunk overlap the GEMMs of the other; anything else = 1 */
    int32_t fuse_ln;      /* 1: LayerNorm folding -- LN2 / the next layer's LN1 applied inside the fc1 / qkv GEMM
                             epilogues from statistics the proj / fc2 write-out emits (no LN pass);
                             same results within the bf16 rounding noise, throughput-neutral on MI355X.  0: off */
} vsc_encoder_config;

int vsc_encoder_create(const vsc_encoder_config *cfg, vsc_encoder **out);
void vsc_encoder_destroy(vsc_encoder *enc);

/* Upload one weight tensor (float32, host memory, canonical names and layouts of
 * vsc_hip/weights.py: "patch.weight", "blocks.3.qkv.bias", "head.weight" ...).
 * `count` is the number of float32 elements and is checked against the config. */
int vsc_encoder_set_weight(vsc_encoder *enc, const char *name, const float *host, size_t count);
/* Check completeness, build the bf16 device copies.  Required before forward. */
int vsc_encoder_finalize(vsc_encoder *enc);

/* frames_dev: float32 [n, channels, image, image], already normalised as
 * infer/src/transform.py does.  desc_dev: float32 [n, desc_dim] where desc_dim =
 * out_dim ? out_dim : width.  Asynchronous on `stream`. */
int vsc_encoder_forward(vsc_encoder *enc, const float *frames_dev, int64_t n, float *desc_dev,
                        void *stream);
/* The same from DECODED frames: frames_u8_dev uint8 [n, image, image, channels] (PIL / numpy layout); torchvision's
 * ToTensor + Normalize(mean, std) (infer/src/transform.py:37-42, extract_query_feats.py:97-105) run inside the
 * patchify kernel in the reference's fp32 op order, so the result is bit-identical to vsc_encoder_forward on the fp32
 * tensor -- with a quarter of the bytes over PCIe and HBM.  mean / std: `channels` floats in HOST memory. */
int vsc_encoder_forward_u8(vsc_encoder *enc, const uint8_t *frames_u8_dev, int64_t n, const float *mean,
                           const float *std, float *desc_dev, void *stream);
/* Same, but also copies the last hidden state (after the final LayerNorm),
 * float32 [n, tokens, width], for parity tests.  tokens_dev may be NULL. */
int vsc_encoder_forward_debug(vsc_encoder *enc, const float *frames_dev, int64_t n,
                              float *desc_dev, float *tokens_dev, void *stream);
int64_t vsc_encoder_workspace_bytes(const vsc_encoder *enc);

/* Per-kernel-class timing for bench.py's roofline: when on, every launch of
 * vsc_encoder_forward is bracketed by HIP events on the caller's stream.
 * vsc_encoder_get_profile synchronises the device, then returns the accumulated
 * milliseconds and launch counts per class since profiling was switched on. */
typedef enum vsc_prof_class {
    VSC_PROF_PATCHIFY = 0, VSC_PROF_GEMM_PATCH = 1, VSC_PROF_LAYERNORM = 2, VSC_PROF_GEMM_QKV = 3,
    VSC_PROF_ATTENTION = 4, VSC_PROF_GEMM_PROJ = 5, VSC_PROF_GEMM_FC1 = 6, VSC_PROF_GEMM_FC2 = 7,
    VSC_PROF_POOL_HEAD = 8, VSC_PROF_MISC = 9, VSC_PROF_CLASSES = 10
} vsc_prof_class;
int vsc_encoder_set_profiling(vsc_encoder *enc, int32_t on);
int vsc_encoder_get_profile(vsc_encoder *enc, double ms_out[VSC_PROF_CLASSES],
                            int64_t launches_out[VSC_PROF_CLASSES]);

/* ------------------------------------------------------------------------ *
 * Swin-Transformer-V2 frame encoder: replaces `model(flat_frames)` for the swinv2_v106/v107/v115
 * TorchScript backbones (infer/infer_ref.sh; model = train/train_v115/torch2scripts.py:480-657:
 * patch embed + norm, stages of res-post-norm blocks with windowed cosine attention and continuous
 * relative position bias, patch merging, final norm, GeM(p) over tokens, output_proj).
 * Weight names are the reference's state-dict names ("layers.2.blocks.5.attn.qkv.weight" ...).
 * Supported: head_dim 32, window 8, 12, 16 or 24 (after clipping to the feature map; 8 and 16 run the tuned kernel), mlp_ratio 4.
 * ------------------------------------------------------------------------ */
typedef struct vsc_swin vsc_swin;

typedef struct vsc_swin_config {
    int32_t image_size, patch_size, channels, embed_dim, stages;
    int32_t depths[4], heads[4];
    int32_t window_size;
    int32_t pretrained_window_sizes[4];
    int32_t mlp_ratio, out_dim;
    float ln_eps, gem_p;
    int32_t max_batch, l2_normalize;
} vsc_swin_config;

int vsc_swin_create(const vsc_swin_config *cfg, vsc_swin **out);
void vsc_swin_destroy(vsc_swin *enc);
int vsc_swin_set_weight(vsc_swin *enc, const char *name, const float *host, size_t count);
int vsc_swin_finalize(vsc_swin *enc);
/* frames_dev f32 [n, channels, image, image] -> desc_dev f32 [n, out_dim]; asynchronous on `stream`. */
int vsc_swin_forward(vsc_swin *enc, const float *frames_dev, int64_t n, float *desc_dev, void *stream);
/* uint8 [n, image, image, channels] input, normalisation fused as in vsc_encoder_forward_u8 */
int vsc_swin_forward_u8(vsc_swin *enc, const uint8_t *frames_u8_dev, int64_t n, const float *mean, const float *std,
                        float *desc_dev, void *stream);
/* also returns the last-stage tokens after the final LayerNorm, f32 [n, tokens_last, width_last] */
int vsc_swin_forward_debug(vsc_swin *enc, const float *frames_dev, int64_t n, float *desc_dev,
                           float *tokens_dev, void *stream);
int64_t vsc_swin_workspace_bytes(const vsc_swin *enc);

/* Per-kernel-class timing for bench.py's Swin roofline, as vsc_encoder_set_profiling: when on, every launch of
 * vsc_swin_forward is bracketed by HIP events on the stream it runs on, and the chunks of a call run back to back on the
 * caller's stream (no lanes), so every kernel is timed alone.  Classes: patchify, patch embedding (GEMM + LayerNorm), then
 * per stage s (0..3) at VSC_SWIN_PROF_STAGE0 + s * VSC_SWIN_PROF_PER_STAGE: qkv GEMM, window attention, proj GEMM with the
 * res-post-norm LayerNorm, fc1 GEMM (+GELU), fc2 GEMM with its LayerNorm, patch merging (gather + reduction GEMM +
 * LayerNorm); last the final LayerNorm + GeM + head.  vsc_swin_get_profile synchronises the device. */
enum {
    VSC_SWIN_PROF_PATCHIFY = 0, VSC_SWIN_PROF_PATCH_EMBED = 1, VSC_SWIN_PROF_POOL_HEAD = 2, VSC_SWIN_PROF_STAGE0 = 3,
    VSC_SWIN_PROF_QKV = 0, VSC_SWIN_PROF_ATTENTION = 1, VSC_SWIN_PROF_PROJ_LN = 2, VSC_SWIN_PROF_FC1 = 3,
    VSC_SWIN_PROF_FC2_LN = 4, VSC_SWIN_PROF_MERGE = 5, VSC_SWIN_PROF_PER_STAGE = 6, VSC_SWIN_PROF_CLASSES = 27
};
int vsc_swin_set_profiling(vsc_swin *enc, int32_t on);
int vsc_swin_get_profile(vsc_swin *enc, double ms_out[VSC_SWIN_PROF_CLASSES], int64_t launches_out[VSC_SWIN_PROF_CLASSES]);

/* ------------------------------------------------------------------------ *
 * Flat inner-product search: replaces faiss.IndexFlat(d, METRIC_INNER_PRODUCT)
 *   .search(x, k)        infer/vsc/index.py:167-175, infer/vsc/baseline/score_normalization.py:95,141,
 *                        infer/vsc/exhaustive_search.py:66 (the k = 1024 probe of range_search_gpu)
 *   .range_search(x, r)  infer/vsc/exhaustive_search.py:78,250
 * Scores are the ascending-k float32 fmaf chain (bit-identical to
 * oracle/knn_oracle.c); ties rank the lower reference index first.
 * ------------------------------------------------------------------------ */

/* q_dev [nq,d], r_dev [nr,d] float32 row-major, 1 <= d <= 4096, 1 <= k <= 1024.  out_scores_dev [nq,k] float32 descending, out_ids_dev [nq,k]
 * int64; slots beyond nr hold (-FLT_MAX, -1).  ref_id_offset is added to every
 * reported id (a shard of a larger bank).
 * Workspace: allocated internally, grow-only, cached per device (bf16 copies of both banks, candidate lists, partial
 * results: ~7 GB after a 1M x 1M call); vsc_search_release_scratch() returns it.  One search call per device at a time.
 * Host synchronisation: on the bf16 pre-filter path (nq * nr >= 2^24, nr >= 4096, k <= 512) the call waits for `stream`
 * once, after the merge, to read the per-block fallback flags -- on return the results are complete; on the exact path
 * (everything smaller) the call only enqueues. */
int vsc_knn_ip_f32(const float *q_dev, int64_t nq, const float *r_dev, int64_t nr, int32_t d,
                   int32_t k, int64_t ref_id_offset, float *out_scores_dev,
                   int64_t *out_ids_dev, void *stream);

/* vsc_knn_ip_f32 with a per-query floor: the k best of {r : <q, r> >= floor_dev[q]}, unused slots (-FLT_MAX, -1).  For a bank swept
 * shard by shard (the pipelined form of the sharded search, infer/vsc/baseline/score_normalization.py:107-150 at configs[3]'s
 * size): floor = the k-th best score of the shards merged so far -- nothing below it can enter the final list, ties at the floor
 * still can (lower id first) -- so a later shard's lists start with a threshold instead of paying their warm-up appends again
 * (a 1M x 125k sweep runs at 870 TFLOP/s, the same rows as one of eight shards behind a floor at the whole bank's rate).
 * floor_dev NULL: vsc_knn_ip_f32.  -FLT_MAX / -inf entries: no floor for that query. */
int vsc_knn_ip_floor_f32(const float *q_dev, int64_t nq, const float *r_dev, int64_t nr, int32_t d, int32_t k,
                         int64_t ref_id_offset, const float *floor_dev, float *out_scores_dev, int64_t *out_ids_dev,
                         void *stream);

/* Merge of per-shard results of vsc_knn_ip_f32 (a bank swept shard by shard, each with its ref_id_offset -- the pipelined form of
 * the sharded search, where shard s is swept while shard s + 1 is still arriving over xGMI): scores / ids [parts][nq][k], every
 * list in the search's order (score descending, equal scores by ascending id; unused slots (-FLT_MAX, -1)) -> the k best of the
 * union in that same order.  parts <= 64.  Equals one vsc_knn_ip_f32 call over the concatenated bank bit for bit
 * (reference: faiss merges its shards' heaps the same way; infer/vsc/index.py:167-175 searches one flat index). */
int vsc_knn_merge_parts_f32(const float *scores_dev, const int64_t *ids_dev, int32_t parts, int64_t nq, int32_t k,
                            float *out_scores_dev, int64_t *out_ids_dev, void *stream);

/* Frees the search scratch (top-k, range search, video-pair maxima) of the current device after waiting for the device;
 * returns the bytes released.  The next search call allocates again. */
int64_t vsc_search_release_scratch(void);

/* Diagnostic: which sweep the last vsc_knn_ip_f32 call of this process took -- 1 the exact fp32 MFMA sweep, 2 the
 * bf16 pre-filter sweep + exact re-scoring (large problems, k <= 512; results are bit-identical to 1; synchronises
 * `stream` once to read its fallback flag), 3 the pre-filter ran, but some 256-query blocks (a candidate band did not fit,
 * or non-finite operands) were redone on the exact sweep.  vsc_set_option("VSC_KNN_PATH", "exact"|"bf16") forces a path. */
int vsc_knn_last_path(void);

/* Diagnostic (bench.py): with profiling on, every vsc_knn_ip_f32 call records HIP events on its stream around its phases;
 * vsc_knn_last_profile waits for the last call and returns ms_out = {pack, sweep (the dominant kernel: knn_sweep_bf16_kernel
 * on path 2, knn_kernel on path 1), exact re-scoring (0 on path 1), merge}. */
void vsc_knn_set_profiling(int on);
int vsc_knn_last_profile(float ms_out[4]);

/* Range search: every pair with <q,r> > radius -- faiss IndexFlat.range_search
 * (infer/vsc/exhaustive_search.py:78,250; the radius sweep behind
 * infer/vsc/index.py:145-165).  lims_dev [nq+1] int64 receives the CSR offsets, *total_out
 * (host) the number of hits.  Hits of query i go to [lims[i], lims[i+1]) of out_scores_dev /
 * out_ids_dev in ascending reference id, but only if total <= capacity; otherwise nothing is
 * written and the caller calls again with capacity >= *total_out (capacity 0 = count only).
 * Synchronises `stream` once (to read the total). */
int vsc_range_search_ip_f32(const float *q_dev, int64_t nq, const float *r_dev, int64_t nr,
                            int32_t d, float radius, int64_t ref_id_offset, int64_t *lims_dev,
                            float *out_scores_dev, int64_t *out_ids_dev, int64_t capacity,
                            int64_t *total_out, void *stream);
/* Which path the last vsc_range_search_ip_f32 call took: 1 = exact (two fp32 sweeps: count, fill), 2 = bf16 pre-filter (one bf16
 * sweep with the radius as a fixed threshold, survivors re-scored with the exact fp32 chain, scan, emit: same CSR output bit for
 * bit), 3 = pre-filter abandoned because a (query, reference split) list held more than 1024 survivors, then exact.  Chosen like
 * vsc_knn_ip_f32's path; VSC_RANGE_PATH=exact|bf16 forces one. */
int vsc_range_search_last_path(void);

/* Per-candidate-pair frame similarity matrices -- the temporal alignment input of the matching track
 * (VSC22-Matching-Track-1st/infer/src/utils.py:29-51,66: np.matmul(qfeat, rfeat.T) per candidate).
 * q_dev [nq, d], r_dev [nr, d]: row banks (all query / reference videos' frames, concatenated).
 * pairs_host [n_pairs][4] = {q_row0, q_rows, r_row0, r_rows} (HOST memory).  Writes out_offsets_host
 * [n_pairs + 1] (element offsets; HOST memory) and out_dev[out_offsets[p] + i * r_rows + j] =
 * <q[q_row0 + i], r[r_row0 + j]> as the ascending-k fp32 fma chain of vsc_knn_ip_f32, so a matrix equals any
 * slice of a larger product bit for bit.  capacity = floats available at out_dev (>= sum q_rows * r_rows;
 * call with out_dev = NULL, capacity = 0 is an error unless the total is 0). */
int vsc_pair_similarity_f32(const float *q_dev, int64_t nq, const float *r_dev, int64_t nr, int32_t d,
                            const int64_t *pairs_host, int64_t n_pairs, int64_t *out_offsets_host,
                            float *out_dev, int64_t capacity, void *stream);

/* Video-pair maxima -- the candidate retrieval of the matching track
 * (VSC22-Matching-Track-1st/infer/infer_matching.py:229-262: per query frame top-1024, range_search where the
 * 1024th score still clears SEARCH_THRESHOLD, then max per (query video, reference video) in a dict).  The
 * union of the two faiss branches is {(qf, rf): <qf, rf> > threshold}; this entry point sweeps all pairs once
 * and keeps, per video pair, the largest frame score above `threshold` (strict, as in the reference).
 * q_video_dev [nq] / r_video_dev [nr]: int32 video index of every row (0 <= index < n_*_videos).
 * lims_dev [n_q_videos + 1] int64 receives CSR offsets over query videos, *total_out (host) the number of
 * video pairs.  Pairs of query video v go to [lims[v], lims[v+1]) of out_rvideo_dev / out_score_dev in
 * ascending reference video, but only if total <= capacity; otherwise nothing is written and the caller
 * calls again with capacity >= *total_out (capacity 0 = count only).  Scores are the fp32 chains of
 * vsc_knn_ip_f32 (bit-exact).  Uses a dense n_q_videos x n_r_videos uint32 table in scratch.
 * Synchronises `stream` once (to read the total). */
int vsc_video_pair_max_f32(const float *q_dev, int64_t nq, const int32_t *q_video_dev, int32_t n_q_videos,
                           const float *r_dev, int64_t nr, const int32_t *r_video_dev, int32_t n_r_videos,
                           int32_t d, float threshold, int64_t *lims_dev, int32_t *out_rvideo_dev,
                           float *out_score_dev, int64_t capacity, int64_t *total_out, void *stream);
/* Which sweep the last vsc_video_pair_max_f32 call ran: 1 = exact fp32 sweep, 2 = bf16 pre-filter (fixed-threshold variant of the
 * top-k pre-filter: survivors of s~ >= threshold - eps are re-scored with the exact fp32 chain before they reach the table, so
 * the result is the same table), 3 = pre-filter with some blocks of 256 queries redone on the exact sweep (a (query, reference
 * split) list held more than 1024 survivors).  Chosen like vsc_knn_ip_f32's path; VSC_PAIRMAX_PATH=exact|bf16 forces one. */
int vsc_video_pair_max_last_path(void);

/* sklearn.preprocessing.normalize(x) in place (l2, axis=1; zero rows untouched):
 * infer/extract_query_feats.py:178, infer/vsc/baseline/score_normalization.py:84-88. */
int vsc_l2_normalize_f32(float *x_dev, int64_t n, int32_t d, void *stream);

/* ------------------------------------------------------------------------ *
 * Building blocks, exported so the parity tests can check each kernel alone.
 * bf16 tensors are raw uint16 bit patterns.
 * ------------------------------------------------------------------------ */
typedef enum vsc_epilogue {
    VSC_EPI_BF16 = 0,        /* out bf16 = acc + bias */
    VSC_EPI_GELU_BF16 = 1,   /* out bf16 = gelu(acc + bias): erf GELU to 2.2e-6 + 6.6e-7 |x| absolute (polynomial) */
    VSC_EPI_QGELU_BF16 = 2,  /* out bf16 = quick_gelu(acc + bias) */
    VSC_EPI_RESADD_F32 = 3,  /* out f32  = residual + acc + bias (out may alias residual) */
    VSC_EPI_PATCH_F32 = 4,   /* out f32 row n*T+1+p = acc + bias + pos[1+p] (row = n*(T-1)+p) */
    VSC_EPI_F32 = 5          /* out f32  = acc + bias */
} vsc_epilogue;

/* out[M,N] = epi(A[M,K] . W[N,K]^T + bias[N]);  A, W bf16 row-major, K % 64 == 0,
 * N % 4 == 0.  bias may be NULL.  `aux_dev` = residual (RESADD) or pos (PATCH),
 * `tokens` = T for PATCH. */
int vsc_gemm_bf16(const uint16_t *a_dev, const uint16_t *w_dev, const float *bias_dev,
                  const float *aux_dev, void *out_dev, int64_t m, int32_t n, int32_t k,
                  int32_t epilogue, int32_t tokens, void *stream);

/* qkv_dev bf16 [frames*tokens, 3*width] (q | k | v column blocks, head-major inside
 * each) -> out_dev bf16 [frames*tokens, width]; softmax(q k^T / 8) v per head;
 * head_dim 64. */
int vsc_attention_bf16(const uint16_t *qkv_dev, uint16_t *out_dev, int32_t frames, int32_t tokens,
                       int32_t heads, void *stream);

/* Row LayerNorm over `width` of float32 x [rows,width]; out is bf16 (out_f32 = 0)
 * or float32 (out_f32 = 1; may alias x). */
int vsc_layernorm_f32(const float *x_dev, const float *gamma_dev, const float *beta_dev,
                      void *out_dev, int64_t rows, int32_t width, float eps, int32_t out_f32,
                      void *stream);

/* frames float32 [n,C,H,W] -> patches bf16 [n*grid*grid, kpad], k = c*p*p + py*p + px,
 * zero-filled up to kpad (kpad % 64 == 0). */
int vsc_patchify_bf16(const float *frames_dev, uint16_t *patches_dev, int64_t n, int32_t channels,
                      int32_t image, int32_t patch, int32_t kpad, void *stream);

/* Swin-V2 windowed cosine attention, head_dim 32.  qkv_dev bf16 [frames*res*res, 3*heads*32] in
 * image token order; the cyclic shift and window partition are index math.  bias_dev f32
 * [heads, (2*window-1)^2]: the compact relative-position table 16*sigmoid(cpb_mlp(coords)),
 * bias(i, j) = table[(yi-yj+window-1)*(2*window-1) + xi-xj+window-1]; scale_dev f32 [heads] =
 * exp(min(logit_scale, ln 100)).
 * Bounded form (optional, per head): cosine logits cannot exceed U = scale + max(table).  A caller that has subtracted U from a
 * head's table and knows 2 scale + max(table) - min(table) <= 69 (so that no probability underflows) passes -scale for that
 * head: the kernel then skips the row maximum of the softmax -- the same quotient, a sixth fewer vector instructions
 * (vsc_swin_finalize does this for its own tables). */
int vsc_window_attention_bf16(const uint16_t *qkv_dev, uint16_t *out_dev, const float *bias_dev,
                              const float *scale_dev, int32_t frames, int32_t res, int32_t window,
                              int32_t shift, int32_t heads, void *stream);
/* x_out = (x_in ? x_in : 0) + LayerNorm(t) ; xb = bf16(x_out).  x_in may be NULL or alias x_out. */
int vsc_ln_residual_f32(const float *t_dev, const float *gamma_dev, const float *beta_dev,
                        const float *x_in_dev, float *x_out_dev, uint16_t *xb_dev, int64_t rows,
                        int32_t width, float eps, void *stream);
/* The same update with the LayerNorm input produced in place by a GEMM whose tile owns whole rows:
 * x_out = (x_in ? x_in : 0) + LayerNorm(A[m,k] W[n,k]^T + bias) ; xb = bf16(x_out).  n in {128, 256, 512},
 * k % 32 == 0 (torch2scripts.py:297-300, 361-362: Swin-V2 res-post-norm and the PatchMerging norm). */
int vsc_gemm_ln_bf16(const uint16_t *a_dev, const uint16_t *w_dev, const float *bias_dev,
                     const float *gamma_dev, const float *beta_dev, const float *x_in_dev,
                     float *x_out_dev, uint16_t *xb_dev, int64_t m, int32_t n, int32_t k, float eps,
                     void *stream);
/* The whole MLP of a Swin-V2 block in one kernel (c = 128, 256 or 512), in place on the residual stream:
 *   x += LayerNorm(GELU(xb W1[4c,c]^T + b1) W2[c,4c]^T + b2) * gamma + beta ;  xb = bf16(x)
 * (Mlp + norm2 + residual of SwinTransformerBlock.forward, torch2scripts.py:18-35, 190-215, 297-300) -- the hidden activations
 * [m, 4c] never reach memory.  w2p_dev is fc2.weight in the kernel's contraction order, as made by
 * vsc_swin_mlp_permute_hidden_f32 (host arrays of c * 4c floats, once per model load) and then converted to bf16: for
 * c = 128 / 256 the hidden axis of every row reordered inside its 32-blocks, for c = 512 additionally chunk-major
 * [64 chunks][512][32] (one wave per SIMD with the whole register file: csrc/swin_mlp512.hip; m < 2^21 rows per call). */
int vsc_swin_mlp_bf16(const uint16_t *w1_dev, const float *b1_dev, const uint16_t *w2p_dev, const float *b2_dev,
                      const float *gamma_dev, const float *beta_dev, float *x_dev, uint16_t *xb_dev, int64_t m, int32_t c,
                      float eps, void *stream);
int vsc_swin_mlp_permute_hidden_f32(const float *w2_host, float *w2p_host, int32_t c);
/* The whole second half of a Swin-V2 block in one kernel (c = 128, 256 or 512), in place on the residual stream
 * (SwinTransformerBlock.forward after the window attention, torch2scripts.py:284-300):
 *   x1 = x + LayerNorm(att Wp[c,c]^T + bp) * gamma1 + beta1 ;  x = x1 + LayerNorm(GELU(bf16(x1) W1^T + b1) W2^T + b2) * gamma2 + beta2 ;  xb = bf16(x)
 * att_dev [m, c] bf16 is the attention output; xb_dev is written only (the MLP's input is formed in registers).  w2p_dev as for
 * vsc_swin_mlp_bf16.  At c = 512 x1 passes through x_dev once as fp32 between the two halves; m < 2^21. */
int vsc_swin_proj_mlp_bf16(const uint16_t *att_dev, const uint16_t *wp_dev, const float *bp_dev, const float *gamma1_dev, const float *beta1_dev,
                           const uint16_t *w1_dev, const float *b1_dev, const uint16_t *w2p_dev, const float *b2_dev, const float *gamma2_dev,
                           const float *beta2_dev, float *x_dev, uint16_t *xb_dev, int64_t m, int32_t c, float eps, void *stream);
/* ... and the NEXT block's qkv Linear behind it (c = 512 only -- the stage with 18 blocks, torch2scripts.py:284-300 followed by the
 * next block's WindowAttention.forward first line, :120-125):   qkv_next = bf16(x) Wq[3c,c]^T + bq   with x the block's output.
 * The bf16 shadow never leaves the registers (no xb_dev), the next block's qkv launch and its read of the shadow are gone.
 * bq_dev [3c] = (q_bias | 0 | v_bias).  Same bits as vsc_swin_proj_mlp_bf16 followed by vsc_gemm_bf16 on its shadow. */
int vsc_swin_proj_mlp_qkv_bf16(const uint16_t *att_dev, const uint16_t *wp_dev, const float *bp_dev, const float *gamma1_dev, const float *beta1_dev,
                               const uint16_t *w1_dev, const float *b1_dev, const uint16_t *w2p_dev, const float *b2_dev, const float *gamma2_dev,
                               const float *beta2_dev, const uint16_t *wq_dev, const float *bq_dev, float *x_dev, uint16_t *qkv_next_dev,
                               int64_t m, int32_t c, float eps, void *stream);
/* PatchMerging gather on bf16 tokens [frames, res, res, c] -> [frames*(res/2)^2, 4c] */
int vsc_merge_gather_bf16(const uint16_t *xb_dev, uint16_t *out_dev, int64_t frames, int32_t res,
                          int32_t c, void *stream);

/* Measurement aid for vsc_swin_mlp_bf16 at c = 512: buf_dev (NULL: off) receives, per workgroup and wave, eight uint32 -- shader
 * cycles spent waiting for the wave's own LDS-DMA pieces, at the chunk barrier, in the chunk's work, before the epilogue, in the
 * epilogue -- when the timing variant of the kernel runs (vsc_set_option("VSC_SWIN_MLP_ABL", "5")); [ceil(m / 128)][4][8]. */
int vsc_debug_mlp512_timing(uint32_t *buf_dev);
/* Measurement aid: one wave spins for `ticks` shader cycles (s_memtime) and stores the elapsed count. */
int vsc_debug_spin_ticks(uint64_t ticks, uint64_t *out_dev, void *stream);

/* ------------------------------------------------------------------------
 * Matching track: the fp32 convolution layers of the pair classifier (timm mobilenetv3_small_100) and the refinement
 * net (timm hrnet_w18 features + 1x1 fuse head) that the reference runs as TorchScript modules on similarity maps
 *   VSC22-Matching-Track-1st/infer/infer_matching.py:158-175 (match_classify), :177-204 (match_refine),
 *   train/models.py:6-40 (ClassifyModel, HRnet).
 * Activations are NHWC float32; BatchNorm is folded into weight / bias by the host (vsc_hip/cnn.py).
 * ------------------------------------------------------------------------ */
enum { VSC_ACT_NONE = 0, VSC_ACT_RELU = 1, VSC_ACT_HARDSWISH = 2, VSC_ACT_HARDSIGMOID = 3, VSC_ACT_GELU = 4 };

/* floats per packed weight row: cin * kh * kw rounded up to a multiple of 32 */
int vsc_conv_packed_k(int32_t cin, int32_t kh, int32_t kw);
/* w_dev [cout, k] float32 with k = (kh, kw, cin) order (torch's [cout, cin, kh, kw] permuted to [cout, kh, kw, cin]) ->
 * packed_dev [cout, vsc_conv_packed_k]: rows zero-padded to a multiple of 32 floats for the fp32 MFMA tiles; done once per layer. */
int vsc_conv_pack_weight_f32(const float *w_dev, float *packed_dev, int32_t cout, int32_t k, void *stream);
/* out[n, ho, wo, 0:cout] (row stride ldo) = act(conv2d(x[n, h, w, 0:cin] (row stride ldx), W) + bias [+ res[.., 0:cout]
 * (row stride ldr)]) -- torch.nn.functional.conv2d(stride, padding) + the fused tail of a BN/ReLU/residual block.
 * Layers that materialise their patch matrix share ONE grow-only scratch buffer per device: issue the convolutions of a
 * device from one stream at a time (calls from several host threads are serialised on an internal mutex, their kernels
 * are not ordered against each other).
 * Arithmetic: fp32 products with fp32 accumulation on the fp32 matrix pipe.  These 3 x 3 / stride 1 / pad 1 layers instead multiply
 * bf16 triples (x = x1 + x2 + x3, six products per multiply, fp32 accumulation: the same error against float64 as the fp32 pipe):
 *     cin == 20 with cout <= 20, cin == 36 with cout <= 36           (cin as passed: HRNet's 18 channels come padded to 20)
 *     cin == 64 with cout <= 64, cin == 256 with cout <= 32          dense rows (ldx == cin), >= 65 536 output pixels
 *     cin == 72 or 144 with 48 <= cout <= 160                        dense rows, 8 192 <= pixels, <= 16 M input elements
 * each only with 16-byte aligned x / weights / out / res / bias (anything else takes the fp32 tile kernels); VSC_CONV_X3=0 or
 * VSC_CONV_DIRECT=0 (vsc_set_option) switch all of them off; vsc_conv_last_pipe() tells which pipe a call took.  The 72- / 144-
 * channel form keeps its split operands in a second per-device scratch buffer: the one-stream-at-a-time rule above covers it too. */
int vsc_conv2d_f32(const float *x_dev, int64_t n, int32_t h, int32_t w, int32_t cin, int32_t ldx, const float *w_packed_dev,
                   const float *bias_dev, int32_t cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad,
                   const float *res_dev, int32_t ldr, int32_t act, float *out_dev, int32_t ldo, void *stream);
/* depthwise convolution: w_dev [c, kh * kw], x / out dense NHWC */
int vsc_dwconv2d_f32(const float *x_dev, int64_t n, int32_t h, int32_t w, int32_t c, const float *w_dev,
                     const float *bias_dev, int32_t kh, int32_t kw, int32_t stride, int32_t pad, int32_t act,
                     float *out_dev, void *stream);
/* out[n, c] = mean over the hw positions of x[n, hw, c] (AdaptiveAvgPool2d(1)) */
int vsc_global_avgpool_f32(const float *x_dev, int64_t n, int32_t hw, int32_t c, float *out_dev, void *stream);
/* x[n, hw, c] *= scale[n, c] (squeeze-excite gate) */
int vsc_channel_scale_f32(float *x_dev, const float *scale_dev, int64_t n, int32_t hw, int32_t c, void *stream);
/* One squeeze-excite block in place (timm SqueezeExcite: x * gate_fn(conv_expand(act(conv_reduce(x.mean((2, 3)))))), as one launch for
 * maps whose image fits LDS ((hw c + 2 c + cr) * 4 <= 150 KiB; larger ones: vsc_global_avgpool_f32 + vsc_conv2d_f32 x 2 +
 * vsc_channel_scale_f32): x[n, hw, c] *= act2(W2 act1(W1 mean_hw(x) + b1) + b2), W1 [cr, c] / W2 [c, cr] as packed by
 * vsc_conv_pack_weight_f32 (1 x 1 kernels), biases may be null.  c a multiple of 4. */
int vsc_se_block_f32(float *x_dev, int64_t n, int32_t hw, int32_t c, const float *w1_packed_dev, const float *b1_dev, int32_t cr,
                     const float *w2_packed_dev, const float *b2_dev, int32_t act1, int32_t act2, void *stream);
/* out[n, y, x, coff + ch] = act((accumulate ? out : 0) + src[n, y / factor, x / factor, ch]) for the h x w output grid:
 * nn.Upsample(scale_factor, 'nearest') fused with the sum of an HRNet fuse layer or with torch.cat along channels. */
int vsc_upsample_add_f32(const float *src_dev, int64_t n, int32_t h, int32_t w, int32_t c, int32_t factor, float *out_dev,
                         int32_t ldo, int32_t coff, int32_t accumulate, int32_t act, void *stream);
/* One HRNet fuse node in one pass (timm HighResolutionModule.forward: y = relu(sum_j fuse_layers[i][j](x[j])), the sum taken in
 * order of j): out[n, y, x, ch] = act(((base[n, y, x, ch] + up(src0)) + up(src1)) + up(src2)), up = nearest upsampling by a
 * power-of-two factor, absent sources null.  base [n, h, w, ldb >= c] may be null or alias out [n, h, w, ldo >= c]; srcK
 * [n, h / factorK, w / factorK, c] dense.  c, ldb, ldo multiples of 4, operands 16-byte aligned. */
int vsc_upsample_sum_f32(const float *base_dev, int32_t ldb, const float *src0_dev, int32_t factor0, const float *src1_dev,
                         int32_t factor1, const float *src2_dev, int32_t factor2, int64_t n, int32_t h, int32_t w, int32_t c,
                         int32_t act, float *out_dev, int32_t ldo, void *stream);

/* Diagnostic (bench.py: FLOPs per pipe): which pipe the last vsc_conv2d_f32 call of this process ran on -- 0 the fp32 matrix / vector
 * pipe, 1 the bf16 matrix pipe on split operands (3 x 3 / stride 1 layers with 20, 36, 64, 72, 144 input channels and the 256 -> <= 32
 * transition, unless VSC_CONV_X3=0 / VSC_CONV_DIRECT=0: six bf16 products per multiply, fp32-level error). */
int vsc_conv_last_pipe(void);

/* fp32 multi-head self-attention for short sequences: out[t, h*dh:(h+1)*dh] = softmax(q k^T / sqrt(dh)) v per head, qkv
 * [tokens, 3 * heads * head_dim] float32 as q | k | v column blocks.  Used by the video-score head (BERT encoder over <= 258
 * tokens, infer/extract_query_feats.py:165-173), whose sigmoid gate is compared with 1e-3 and therefore runs in fp32. */
int vsc_attention_f32(const float *qkv_dev, float *out_dev, int32_t tokens, int32_t heads, int32_t head_dim, void *stream);
/* The same for `seqs` sequences of the same length stored back to back (qkv [seqs * tokens, 3 * heads * head_dim]): the video-score
 * heads of a group of query videos in one launch per layer (their Linears and LayerNorms are row-wise and take all rows at once). */
int vsc_attention_f32_batch(const float *qkv_dev, float *out_dev, int32_t tokens, int32_t heads, int32_t head_dim, int32_t seqs,
                            void *stream);
/* ... and for sequences of different lengths (the video-score heads of a group of query videos of any lengths in one launch per
 * layer; train_vid_score/video/model.py: one video per forward): sequence z = rows row_offsets_dev[z] .. row_offsets_dev[z + 1] of
 * qkv_dev / out_dev (int32 [seqs + 1] on the device), max_tokens = the longest sequence.  A row's result is the same bits as in
 * vsc_attention_f32 on its sequence alone. */
int vsc_attention_f32_varlen(const float *qkv_dev, float *out_dev, const int32_t *row_offsets_dev, int32_t seqs, int32_t max_tokens,
                             int32_t heads, int32_t head_dim, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VSC_HIP_H */
