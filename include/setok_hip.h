/*
 * setok_hip.h — C ABI of libsetok_hip.so, the MI355X (gfx950) implementation of the SeTok
 * `encode_images` hot path.
 *
 * The reference (ChocoWu/SeTok) has NO native / FFI interface for this path: it is 100 % Python
 * over stock torch ops.  The boundary a maintainer binds is therefore the
 * set of torch-op groups of the reference's hot path; each entry point below cites the reference
 * lines (relative to /root/reference/) whose arithmetic it replaces.  The Python host that mirrors
 * the reference's `SetokTokenizer` / `build_vision_tower` / `encode_images` surface
 * (setok_amd/) calls these through ctypes; INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it, allocates
 *     nothing, and synchronises nothing — outputs and workspaces are caller-allocated (the context
 *     entry points are the documented exception: they own the weights; setok_encode synchronises only on request);
 *   - `dtype` selects the activation/weight element type: SETOK_F32 (parity mode; fp32 MFMA,
 *     exact fma chains), SETOK_BF16 (throughput mode; bf16 MFMA, fp32 accumulation) or — since ABI 9 —
 *     SETOK_F16 (IEEE half, fp16 MFMA at the bf16 rate, fp32 accumulation: what the reference's inference
 *     loader and its non-`--bf16` training launches cast the tower to, src/model/builder.py:43,135-136,
 *     src/train/train_setokim.py:326,348,374).  The 16-bit kernels are written ONCE against "a 16-bit float
 *     element with fp32 accumulation" and compiled twice: libsetok_hip.so serves SETOK_F32 + SETOK_BF16,
 *     libsetok_hip_f16.so (the same sources under -DSETOK_HALF, the same exported names) serves SETOK_F32 +
 *     SETOK_F16; each refuses the other's 16-bit code, a host binds the one(s) it needs (dlopen with
 *     RTLD_LOCAL: both export this header's names; the Python host routes by tensor dtype, setok_amd/_lib.py);
 *     biases, LayerNorm affine parameters, scores and distances are always fp32;
 *   - matrices are row-major and dense unless a leading dimension is given;
 *   - return value: 0 on success, a negative SETOK_E* code otherwise; setok_last_error() returns
 *     a thread-local human-readable message for the last failure.
 */
#ifndef SETOK_HIP_H
#define SETOK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SETOK_ABI_VERSION 9

enum { SETOK_F32 = 0, SETOK_BF16 = 1, SETOK_F16 = 2 };
enum { SETOK_ACT_NONE = 0, SETOK_ACT_QUICK_GELU = 1, SETOK_ACT_GELU_ERF = 2 };
enum { SETOK_OK = 0, SETOK_EINVAL = -1, SETOK_ELAUNCH = -2, SETOK_EUNSUPPORTED = -3 };

int setok_abi_version(void);
const char* setok_last_error(void);
/* Name of the device the library sees ("" if none), its CU count; both cheap, host-side only. */
int setok_device_info(char* name_host, int name_cap, int* cu_count_host);

/* Launch profiler (measurement aid, off by default; process-wide): between start and stop every setok_linear / setok_linear_ln /
 * setok_cluster_dpc_knn call is timed by a pair of HIP events on its stream — attached to the call's first and last kernel dispatch
 * (hipExtLaunchKernelGGL: the dispatches' own start / end timestamps, no marker packets) where the call launches the LDS-DMA GEMM kernels or
 * the single-launch clustering, recorded as markers around the call otherwise.  stop waits for them and returns the number of records written:
 * kind 0 = bf16 GEMM, 1 = fp32 GEMM (work = FLOPs), 2 = clustering (work = Gram FLOPs); cls = act | residual << 2 | folded LayerNorm << 3;
 * bytes = algorithmic bytes of the call; ms = event duration.  Call start / stop while no other library call is in flight. */
int setok_profile_start(void);
int setok_profile_stop(int* kind, int* cls, double* work, double* bytes, float* ms, int cap);
/* Between start and stop: pause != 0 stops attaching events to launches (the records so far stay), 0 resumes.  An event pair per launch costs ~4 us of device
 * time: a caller timing many steps probes some of them. */
int setok_profile_pause(int pause);

/* ---- the whole path behind one call (host-language-neutral entry) -------------------------------------------------
 * `SetokTokenizer.forward` (src/model/setok/tokenizer.py:157-182): tower (clip_encoder.py:50-62, HF CLIP ViT hidden_states[select_layer],
 * feature_select :40-48) -> + PositionalEncoding2D (:164-168) -> cluster_dpc_knn (:174) -> group_encoding (:177-178) -> inter_encoder
 * (:179) -> out (:180).  A context owns device copies of the weights in compute layout; setok_encode runs the path on the caller's
 * stream into caller-allocated buffers.  These are the only entry points that allocate (create / load / ready); setok_encode itself is
 * asynchronous: the data-dependent shapes of the ragged stages (per-image token counts) are read on the DEVICE, the host reads them after the
 * call when it wants them (see below).
 * Contexts are independent: no global mutable state, one context per (device, stream) in use at a time. */
typedef struct setok_ctx setok_ctx;

typedef struct setok_config {
    /* tower: the HF CLIP vision config behind clip_encoder.py:35 */
    int image_size, patch_size, hidden_size, intermediate_size, num_hidden_layers, num_attention_heads;
    float layer_norm_eps;
    int select_layer;            /* mm_vision_select_layer (clip_encoder.py:41): index into hidden_states, negative from the end */
    int select_cls_patch;        /* mm_vision_select_feature: 0 = 'patch' (drop the class token, :43), 1 = 'cls_patch' */
    /* head: ctor kwargs of SetokTokenizer (tokenizer.py:14-34); hidden_dim == tower hidden_size */
    int token_feat_dim, nheads, dim_feedforward, inner_cluster_layers, intra_cluster_layers, min_cluster_num;
    float threshold;
    int dtype;                   /* SETOK_BF16 / SETOK_F16 (throughput mode; F16 in libsetok_hip_f16.so) or SETOK_F32 (parity mode) */
    int fold_layernorm;          /* 16-bit modes only: fold layer_norm1 / layer_norm2 of the tower into the q|k|v / fc1 GEMMs */
} setok_config;

int setok_create(const setok_config* cfg, setok_ctx** out);
void setok_destroy(setok_ctx* ctx);
const char* setok_ctx_error(const setok_ctx* ctx);

/* One parameter by its name in the reference's state dict (`image_feature_encoder.vision_tower.` + HF CLIPVisionModel names, with or
 * without HF 4.x's `vision_model.`; `inner_encoder.*`, `inter_encoder.*`, `out.*`), plus `position_embedding.table`: the (N, C)
 * PositionalEncoding2D table of module.py:118-146 in the compute dtype (the reference builds it on the host in fp32 and casts; passing it
 * keeps it bit-identical to the host's).  `ptr` is a DEVICE pointer to a dense tensor of `dtype` and `shape`; the data is copied
 * (matrices to the compute dtype, vectors to fp32) on `stream`. */
int setok_load_weight(setok_ctx* ctx, void* stream, const char* name, const void* ptr, int dtype, const int64_t* shape, int ndim);

/* After the last setok_load_weight: checks that every parameter the configuration needs is there (the error names the first missing
 * one) and builds the fused q|k|v matrices, the padded patch matrix and the folded LayerNorm operands. */
int setok_weights_ready(setok_ctx* ctx, void* stream);

/* Bytes of 256-byte-aligned device workspace setok_encode needs for a batch of B images.
 * STREAM-ORDER CONTRACT of the workspace (and of every output buffer of setok_encode): the call's launches read and write it on `stream`, and
 * since ABI 8 the counts_host form RETURNS WHILE THE LAST OF THEM ARE STILL RUNNING.  The buffers therefore belong to `stream` until work enqueued
 * behind the call on that stream has run: reuse on the same stream needs nothing; reuse, reading (hipMemcpyAsync on another or a non-blocking
 * stream) or freeing from ANY OTHER stream or from the host needs an event recorded on `stream` after the call (or hipStreamSynchronize(stream)).
 * hipFree / hipFreeAsync on another stream without that order is a use-after-free, exactly as for any other asynchronous launch. */
int64_t setok_encode_workspace_bytes(const setok_ctx* ctx, int B);

/* images (B, 3, image_size, image_size) in the compute dtype -> tokens: packed (sum_b L_b, token_feat_dim) rows, image b owning rows
 * [sum_{b' < b} L_b', + L_b) (capacity B * N rows; rows past sum_b L_b are never written), counts (B) int32 on the device, idx_cluster (B, N)
 * int64, score (B, N) fp32, index_down (B, N) int64 (-1 padded).  k / threshold == 0 select the configured defaults (the reference's
 * truthiness rule, tokenizer.py:171-172); noise / token_mask as in setok_cluster_dpc_knn (NULL = none).  The optional stage_* outputs
 * receive pointers INTO the workspace (valid until its next use): x = features + positions (B*N, C), group (sum L, C), inter (sum L, C).
 *
 * ASYNCHRONOUS (since ABI 5): no host synchronisation between the stages — the ragged stages are launched at their worst-case
 * size and read the per-image token counts on the device — so with counts_host == NULL the call only enqueues work on `stream` (it can be
 * captured into a hipGraph) and the host reads `counts` whenever it needs shapes.  counts_host != NULL (B ints; total_tokens_host optional) is
 * the convenience form: ONE host wait at the END of the call fills them.  Since ABI 8 that wait is for the COUNTS only (they are final behind
 * the clustering stage and are copied to the host there): when the call returns, its remaining launches — the head, about 1 ms of device time at
 * batch 256 — may still be queued or running, so that the caller's next launches (the projector) queue up behind them and the device never
 * idles.  `tokens`, `idx_cluster`, `score`, `index_down` and the stage pointers are complete IN STREAM ORDER like the outputs of every other call
 * of this library: work enqueued on `stream` sees them; a host that reads them directly synchronises the stream first (a blocking hipMemcpy on
 * the null stream does). */
int setok_encode(setok_ctx* ctx, void* stream, const void* images, int B, int k, float threshold, const float* noise,
                 const float* token_mask, void* workspace, int64_t workspace_bytes, void* tokens, int32_t* counts,
                 int64_t* idx_cluster, float* score, int64_t* index_down, int32_t* counts_host, int64_t* total_tokens_host,
                 void** stage_x, void** stage_group, void** stage_inter);

/* ---- dense layers ------------------------------------------------------------------------ */

/* C[M,N] = act(A[M,K] · W[N,K]^T + bias[N]) + residual[M,N]          (bias, residual optional)
 * Replaces every nn.Linear on the path: module.py:40,43 (Mlp), :63,71 (Attention qkv/proj),
 * tokenizer.py:180 (`out`), multimodal_projector/builder.py:37-59 (mm_in_projector) and the HF CLIP
 * q/k/v/out_proj/fc1/fc2 linears reached from clip_encoder.py:59.  `act` fuses nn.GELU (exact erf,
 * module.py:41) or CLIP's quick_gelu.  A and W have element type `dtype`; C and residual have
 * `out_dtype` (C may alias residual).  lda/ldc are row strides in elements; K % 64 == 0 (bf16) or
 * K % 16 == 0 (fp32).  `batch` > 1 runs independent problems with the given element strides
 * (bias shared). */
int setok_linear(void* stream, int dtype, int out_dtype, const void* A, int64_t lda, const void* W,
                 const float* bias, const void* residual, void* C, int64_t ldc, int M, int N, int K,
                 int act, int batch, int64_t strideA, int64_t strideW, int64_t strideC);

/* y[r,:] = LayerNorm(x[r,:]) * gamma + beta   — nn.LayerNorm (module.py:81,83; HF CLIP
 * pre_layrnorm / layer_norm1 / layer_norm2).  Statistics in fp32.  x may alias y. */
int setok_layernorm(void* stream, int dtype, const void* x, const float* gamma, const float* beta,
                    void* y, int rows, int C, float eps);

/* y = act(x) elementwise (n elements) — nn.GELU placed after a LayerNorm in the `mlp{N}x_gelu_Norm`
 * projector (multimodal_projector/builder.py:48-58), where it cannot be fused into a GEMM epilogue. */
int setok_activation(void* stream, int dtype, const void* x, void* y, int64_t n, int act);

/* Training-mode dropout of the head's Block: nn.Dropout(proj_drop) after the attention projection (module.py:59,72), after the Mlp's activation
 * and after its fc2 (module.py:36,44,45); proj_drop = 0.2 by default (tokenizer.py:26).
 *   y[i] = residual[i] + (keep_i ? x[i] / (1 - p) : 0),   c = offset + i,  keep_i = 16-bit slice (c & 3) of hash(seed, c >> 2) >= p * 2^16
 *   (residual may be NULL; y may alias x or residual; 16-byte aligned operands take the vector kernel, others an element-wise one: same mask)
 * The mask is a pure function of (seed, offset + i) (one SplitMix64 finaliser per four consecutive counters): the backward pass applies the same call to the
 * incoming gradient instead of storing masks, and a step is reproducible from its seed.  Bernoulli(1 - p) like the reference's masks, not
 * bit-equal to torch's Philox stream.  n elements of `dtype`. */
int setok_dropout(void* stream, int dtype, const void* x, const void* residual, void* y, int64_t n, float p, uint64_t seed, uint64_t offset);
/* y = drop(act(x)) in ONE pass: Mlp.forward's `self.drop(self.act(self.fc1(x)))` (module.py:41,44) in training mode.  The same mask as
 * setok_dropout(seed, offset); act(x) is rounded to `dtype` before the mask is applied, so the result is bit-identical to setok_activation
 * followed by setok_dropout in place.  ABI 6. */
int setok_activation_dropout(void* stream, int dtype, const void* x, void* y, int64_t n, int act, float p, uint64_t seed, uint64_t offset);

/* Block-diagonal ("varlen") multi-head self-attention over contiguous row segments.
 * qkv: (rows, 3*H*Dh) laid out [q | k | v], heads inside — the layout both the fused `qkv` Linear
 * of module.py:63 and a concatenated HF q/k/v projection produce.  Row r attends to the rows of
 * its own segment [seg_offsets[s], seg_offsets[s+1]) only:
 *   - ViT tower: segments = images (T rows each)                       clip_encoder.py:59 (HF eager attention)
 *   - inner_encoder: segments = clusters (member tokens only)          tokenizer.py:147-150, module.py:61-73
 *   - inter_encoder: segments = images (L_i cluster tokens each)       tokenizer.py:179
 * out: (rows, H*Dh) = softmax(q k^T * scale) v, heads concatenated (module.py:70).
 * seg_offsets: int32[n_segs+1] on the device; if NULL, uniform segments of `seg_len` rows. */
int setok_attention(void* stream, int dtype, const void* qkv, const int32_t* seg_offsets, int n_segs,
                    int seg_len, void* out, int rows, int H, int Dh, float scale);

/* Multi-head CROSS-attention of uniform query groups to ragged key/value segments — the Q-Former of the
 * reconstruction decoder (cfg 3): BertSelfAttention.forward's cross branch, module.py:283-286 (keys / values from the
 * encoder states), :303 (q k^T), :342 (/ sqrt(d_h), passed as `scale`), :343-345 (+ mask), :348 (softmax), :360-364.
 * The reference pads every image's L_i tokens to a common L and adds (1 - m) * -10000 (module.py:849, 962-973); exp of a
 * score lowered by 10000 is exactly 0 in fp32, so attending to the UNPADDED segment is the same arithmetic.
 * q:   (n_segs * q_len, >= H*Dh) rows, stride ldq; query row r belongs to segment r / q_len.
 * k,v: rows of stride ldkv (typically two column windows of one fused [k | v] projection buffer);
 *      segment s owns rows [kv_offsets[s], kv_offsets[s+1]), at most max_kv of them (int32[n_segs+1], device);
 *      kv_offsets == NULL means uniform segments of max_kv rows.
 * out: (n_segs * q_len, H*Dh) rows of stride ldo. */
int setok_cross_attention(void* stream, int dtype, const void* q, int64_t ldq, const void* k, const void* v,
                          int64_t ldkv, const int32_t* kv_offsets, int n_segs, int q_len, int max_kv, void* out,
                          int64_t ldo, int H, int Dh, float scale);

/* ---- ViT tower glue (HF CLIPVisionEmbeddings reached from clip_encoder.py:59) ------------ */

/* im2col for the stride-p patch conv: images (B,3,H,W) -> patches (B*g*g, Kpad) with column index
 * c*p*p + py*p + px (the flattening of conv weight (C,3,p,p)); columns >= 3*p*p are zero. */
int setok_patchify(void* stream, int dtype, const void* images, void* patches, int B, int H, int W,
                   int p, int Kpad);
/* tokens[b,0,:] = cls + pos[0]; tokens[b,1+i,:] = patch_embed[b,i,:] + pos[1+i]  (embeddings.forward) */
int setok_vit_assemble(void* stream, int dtype, const void* patch_embed, const void* cls, const void* pos,
                       void* tokens, int B, int N, int C);

/* ---- LayerNorm folded into the consuming Linear (bf16 throughput mode) ------------------------------------------------------
 * HF CLIP's encoder layer (reached from clip_encoder.py:59) computes q/k/v = Linear(layer_norm1(h)) and fc1(layer_norm2(h)); the
 * reference's Block does the same with norm1 / norm2 (module.py:88,98).  With W' = gamma * W (rounded to bf16), c = W' 1, b' = b + W beta:
 *     LN(h) W^T + b = rstd_r * (h W'^T - mean_r * c + b' / rstd_r)
 * so the GEMM reads the raw residual stream h and no normalised copy of it is ever written.  The fp32 parity mode keeps the separate
 * setok_layernorm. */

/* stats row r (8 floats): [0], [1] the compact activation-side MFMA fragment of (-mean_r, 1 / rstd_r) (two bf16 pairs: two-way splits),
 * [2] and [4] rstd_r = 1 / sqrt(var + eps), [5] mean_r, the rest 0.  Two-pass fp32 statistics in setok_layernorm's order; rows x C, `dtype`. */
int setok_row_stats(void* stream, int dtype, const void* x, float* stats, int rows, int C, float eps);

/* Once per weight load: W (N, K) bf16, gamma / beta (K) fp32, bias (N) fp32 or NULL ->
 * w_gamma (N, K) bf16, w_colsum (N) fp32, bias_folded (N) fp32, col_frag (N, 4) fp32-sized words: the weight-side MFMA fragment of
 * (w_colsum[n], bias_folded[n]) as 8 bf16. */
int setok_ln_fold(void* stream, const void* W, const float* gamma, const float* beta, const float* bias, void* w_gamma,
                  float* w_colsum, float* bias_folded, float* col_frag, int N, int K);

/* C[M,N] = act(LN(A)[M,K] . W[N,K]^T + bias) from w_gamma / col_frag of setok_ln_fold and the row statistics of A (bf16 in, bf16 out;
 * K % 64 == 0, N % 64 == 0, lda / ldc multiples of 8). */
int setok_linear_ln(void* stream, const void* A, int64_t lda, const void* w_gamma, const float* col_frag, const float* row_stats,
                    void* C, int64_t ldc, int M, int N, int K, int act);

/* ---- SeTok head glue --------------------------------------------------------------------- */

/* x[b,i,:] = hidden[b, i+skip, :] + pos2d[i,:] : feature_select's `[:, 1:]` (clip_encoder.py:43,
 * skip = 1 for 'patch', 0 for 'cls_patch') fused with the PositionalEncoding2D add
 * (tokenizer.py:164-168).  hidden: (B, N+skip, C); pos2d: (N, C) in `dtype`; the sum is rounded
 * once to `dtype`, as the reference's `x + pos_emb` is. */
int setok_select_add_pos(void* stream, int dtype, const void* hidden, const void* pos2d, void* x,
                         int B, int N, int C, int skip);

/* cluster_dpc_knn (tokenizer.py:78-121), batched over B images, each exactly as the reference's
 * per-image call:  D = cdist(x,x)/sqrt(C) (:82) [token_mask :84-86]; density from the k nearest
 * (self included) (:88-90) + noise*1e-6 (:91) [* token_mask :93-94]; delta with the row-j-max
 * quirk (:96-99); score = delta*density (:101); centres = {score > threshold} (:103) else the
 * min_cluster_num best scores, ascending by index (:104-107); idx_cluster = argmin over centre
 * rows, centres own themselves (:111-119).
 *   x:          (B, N, C) `dtype`                      noise, token_mask: (B, N) fp32 or NULL
 *   idx_cluster:(B, N) int64 out                       score: (B, N) fp32 out  (reference: (1,N) per image)
 *   index_down: (B, N) int64 out, first counts[b] entries valid, rest -1
 *   counts:     (B) int32 out  = L_b
 *   dist_ws:    fp32 workspace for the scaled distance matrix, vec_ws: fp32 workspace (density, row max, delta, spare) — sizes from
 *               setok_cluster_workspace; both may be NULL when it reports 0 (bf16, N <= 256, C % 64 == 0: the whole call is ONE launch,
 *               one workgroup per image, the distance matrix lives in MFMA accumulators and never reaches memory).
 * Requires N <= 1024, 1 <= k <= N, min_cluster_num <= N. */
int setok_cluster_dpc_knn(void* stream, int dtype, const void* x, int B, int N, int C, int k,
                          float threshold, int min_cluster_num, const float* noise,
                          const float* token_mask, int64_t* idx_cluster, float* score,
                          int64_t* index_down, int32_t* counts, float* dist_ws, float* vec_ws);

/* Workspace of setok_cluster_dpc_knn for a problem shape, in floats (0 = not needed). */
int setok_cluster_workspace(int dtype, int B, int N, int C, int64_t* dist_floats, int64_t* vec_floats);

/* Stable counting sort of each image's tokens by cluster id (`labels.unique()` order ==
 * ascending label, tokenizer.py:141-143) plus the segment tables the ragged stages need:
 *   perm:        (B*N) int32  — sorted position -> source row (b*N + i)
 *   seg_offsets: (total+1) int32, total = sum_b counts[b]: cluster segments in sorted-row space
 *   img_offsets: (B+1) int32 — prefix sum of counts (segments of the inter-encoder / ragged output)
 * Capacity of seg_offsets must be B*N+1. */
int setok_cluster_sort(void* stream, const int64_t* idx_cluster, const int32_t* counts, int B, int N,
                       int32_t* perm, int32_t* seg_offsets, int32_t* img_offsets);

/* out[p,:] = x[perm[p],:]   (the `x[m]` gathers of tokenizer.py:150, all clusters at once) */
int setok_gather_rows(void* stream, int dtype, const void* x, const int32_t* perm, void* out, int rows, int C);

/* out[s,:] = mean over rows [seg_offsets[s], seg_offsets[s+1]) of h   (tokenizer.py:151).
 * n_segs_dev: device int32 holding the segment count (= img_offsets[B]); max_segs = launch bound. */
int setok_segment_mean(void* stream, int dtype, const void* h, const int32_t* seg_offsets,
                       const int32_t* n_segs_dev, int max_segs, void* out, int C);

/* ---- after the path: prepare_inputs_labels_for_multimodal (setokim_arch.py:213-355) ------------------------------------
 * The reference walks the batch in Python: strips padding by the mask (:258-259), cuts every sequence at its
 * IMAGE_TOKEN_INDEX placeholders, embeds the text pieces, interleaves them with the images' (L_i, D) token matrices
 * (:273-303), truncates (:311-314) and pads to the batch maximum (:317-339).  Here it is three asynchronous steps on device
 * buffers; the host reads `seq_len` once in between to size the outputs (max over the batch, :317).
 *
 * img_offsets: int32[n_images + 1], row offsets of the packed image tokens (image i owns rows [off[i], off[i+1])).
 * Images are consumed in batch order, one per placeholder; a sequence WITHOUT a placeholder still consumes one (:264-271). */

/* Step 1.  seq_len[b] = tokens kept by the mask - placeholders + rows of the sequence's images, truncated to max_length
 * (<= 0: no limit); img_start[b] = index of its first image; status (int32[4]): [0] = 1 if the batch needs more than n_images
 * images (the reference raises IndexError at image_features[cur_image_idx]), [1] = images needed, [2] = 1 if a kept id other than
 * the placeholder lies outside [0, vocab) (the reference's embed_tokens raises IndexError, :273; vocab = 0 disables the check),
 * [3] = flat position b*T + t of the first such id (-1 if none).  attention_mask: uint8 (B,T), NULL = all kept (:250-251).
 * count_ws: int32[3*B] scratch. */
int setok_splice_lengths(void* stream, const int64_t* input_ids, const uint8_t* attention_mask, int B, int T,
                         int64_t image_token_index, int64_t vocab, const int32_t* img_offsets, int n_images, int max_length,
                         int32_t* seq_len, int32_t* img_start, int32_t* status, int32_t* count_ws);

/* Step 2.  For every output position (b, p), p < max_len: src (int32) = embedding-table row (token id) | -(image-token row + 1)
 * | INT32_MIN for a zero padding row; new_labels (NULL iff labels is NULL, :341-342): the token's label, ignore_index on
 * image rows (:293) and padding (:319), target_token_index mapped to ignore_index (:344); new_mask (uint8, optional);
 * new_position_ids (int64, optional; 0..len-1 inside the kept range, 0 in the padding, :321,337).  left_pad selects
 * tokenizer_padding_side == "left" (:324-330). */
int setok_splice_plan(void* stream, const int64_t* input_ids, const uint8_t* attention_mask, const int64_t* labels, int B, int T,
                      int64_t image_token_index, int64_t ignore_index, int64_t target_token_index,
                      const int32_t* img_offsets, const int32_t* seq_len, const int32_t* img_start, int max_len, int left_pad,
                      int32_t* src, int64_t* new_labels, uint8_t* new_mask, int64_t* new_position_ids);

/* Step 3.  out[r, :] = embed_table[src[r]] | image_tokens[-(src[r] + 1)] | 0, r < rows = B * max_len: embed_tokens (:266,284)
 * fused with the concatenations and the zero padding (:296-303, 324-333).  D * sizeof(dtype) must be a multiple of 16.
 * A src the operands cannot serve (>= vocab; an image-token row >= image_token_rows or with image_tokens == NULL, e.g. an
 * IMAGE_TOKEN_INDEX in a text-only call) is never turned into an address: the row is zero-filled and, with `status` (int32[2], optional,
 * device), status[0] = 1 and status[1] = the first such row — where the reference's embed_tokens raises IndexError (:273). */
int setok_splice_rows(void* stream, int dtype, const int32_t* src, const void* embed_table, int vocab, const void* image_tokens,
                      int64_t image_token_rows, void* out, int64_t rows, int D, int32_t* status);

/* Backward of step 3 (the reference gets it from autograd: stage 2 trains mm_in_projector THROUGH the splice, scripts/pretrain_mm_proj.sh:40,
 * setokim_arch.py:290-293): d_image_tokens[-(src[r] + 1)] = d_out[r] (rows no output position consumed — truncation — are zero);
 * d_embed (fp32 (vocab, D), optional, ACCUMULATED into: the caller zeroes it) += d_out[r] at src[r] >= 0 — hardware fp32 atomics, the
 * summation order over a repeated token id is not fixed.  Either output may be NULL. */
int setok_splice_rows_bwd(void* stream, int dtype, const int32_t* src, const void* d_out, int64_t rows, int D,
                          void* d_image_tokens, int64_t image_token_rows, float* d_embed, int vocab);

/* ---- reconstruction decoder: the output the reference never defines ---------------------------------------------
 * SetokDeTokenizer.forward ends at decoder_norm and returns None (src/model/setok/detokenizer.py:101-120) while SeTok.forward hands the result
 * to a pixel-space loss as an image (src/model/setok/model.py:75-76,91).  The pixel head = one setok_linear (decoder_embed_dim -> patch^2 * 3 per
 * query) + this rearrangement: image[b, c, h*p + pi, w*p + qi] = patches[(b*gh + h)*gw + w, (pi*p + qi)*3 + c]; ld = row stride of `patches`
 * in elements (>= 3 p^2: the GEMM output may be padded). */
int setok_unpatchify(void* stream, int dtype, const void* patches, int64_t ld, void* image, int B, int gh, int gw, int p);

/* out[0] = mean over the n elements of (pred - target)^2 (kind 0: WeightedMSELoss without a mask, src/model/loss/mse.py:9-19) or |pred - target|
 * (kind 1: the pixel term of the GAN loss, src/model/loss/discriminator.py:161,170).  fp32 accumulation in two fixed-order stages (no atomics).
 * ws: >= 1024 floats of scratch. */
int setok_pixel_loss(void* stream, int dtype, const void* pred, const void* target, int64_t n, int kind, float* ws, float* out);

/* ---- training step of the trainable head --------------------------------------------------------------
 * The reference trains through torch autograd (src/train/setok_trainer.py / train_setokim.py drive `loss.backward()`); the tower is
 * frozen (clip_encoder.py:50, unfreeze_mm_vision_tower=False) and cluster_dpc_knn is no_grad (tokenizer.py:79), so the backward
 * pass covers group_encoding / inter_encoder / out (tokenizer.py:147-180) and the Block / Attention / Mlp of module.py:29-100.
 * Its GEMMs are setok_linear calls (dX = dY W: A = dY, W = W^T;  dW = dY^T X: A = dY^T, W = X^T, fp32 out); the entry points below
 * are everything else.  All deterministic (no atomics).  `ws` arguments are caller-allocated fp32 scratch. */

/* out[c * ldo + r] = x[r * ldx + c] for r < rows, c < cols; out rows are zero-filled for r in [rows, ldo) (pads the contraction
 * dimension of the following GEMM to its K granule).  chunk > 0 (a divisor of ldo): the padded row range is cut into ldo / chunk
 * pieces stored one after the other, each a (cols, chunk) matrix — the operand layout of a split-K batched dW GEMM whose
 * fp32 partial products are then summed in a fixed order (setok_colsum over the batch).  colsum_partial (optional): fp32
 * [ceil(ldo / 64) * cols]; row b receives the column sums of x over rows [64 b, 64 b + 64) — summing these rows (setok_colsum, fp32)
 * gives the bias gradient without another pass over dY. */
int setok_transpose(void* stream, int dtype, const void* x, int64_t ldx, int rows, int cols, void* out, int64_t ldo, int chunk,
                    float* colsum_partial);

/* out[c] (+)= sum_r x[r, c] (bias gradients).  ws: fp32[ws_rows * cols], ws_rows >= 1 (more rows = more parallelism). */
int setok_colsum(void* stream, int dtype, const void* x, int rows, int cols, float* out, int accumulate, float* ws, int ws_rows);

/* Backward of nn.LayerNorm (module.py:81,83): dx = rstd (g - mean(g) - xhat mean(g xhat)) [+ res], g = dy * gamma;
 * dgamma (+)= sum_rows dy * xhat, dbeta (+)= sum_rows dy.  dx may be NULL (first layer: only the parameter gradients are needed);
 * `accumulate` serves the norm1 shared by a Block's attention sub-layers (module.py:87-88).  ws: fp32[ws_rows * C], ws_rows >= 2. */
int setok_layernorm_bwd(void* stream, int dtype, const void* x, const void* dy, const float* gamma, float eps, int rows, int C,
                        void* dx, const void* res, float* dgamma, float* dbeta, int accumulate, float* ws, int ws_rows);

/* Backward of nn.GELU (exact erf, module.py:41): dx = dy * (Phi(pre) + pre * phi(pre)). */
int setok_gelu_bwd(void* stream, int dtype, const void* pre, const void* dy, void* dx, int64_t n);
/* dx = gelu'(pre) * drop(dy): the backward of `drop(act(fc1 x))` in ONE pass — bit-identical to setok_dropout on the gradient (in place) followed by
 * setok_gelu_bwd.  ABI 6. */
int setok_gelu_bwd_dropout(void* stream, int dtype, const void* pre, const void* dy, void* dx, int64_t n, float p, uint64_t seed, uint64_t offset);

/* Backward of setok_attention (module.py:61-73) over the same segments: dqkv laid out [dq | dk | dv] like qkv.
 * out / dout: (rows, H*Dh).  ws: fp32[2 * rows * H] (log-sum-exp and do.o per row and head). */
int setok_attention_bwd(void* stream, int dtype, const void* qkv, const int32_t* seg_offsets, int n_segs, int seg_len,
                        const void* out, const void* dout, void* dqkv, int rows, int H, int Dh, float scale, float* ws);

/* Backward of setok_segment_mean (tokenizer.py:151): drows[r, :] = dseg[s, :] / n_s for every member row r of segment s. */
int setok_segment_mean_bwd(void* stream, int dtype, const void* dseg, const int32_t* seg_offsets, const int32_t* n_segs_dev,
                           int max_segs, void* drows, int C);

/* torch.optim.AdamW step on fp32 master parameters (decoupled weight decay, bias correction by `step` >= 1), gradient pre-scaled
 * by grad_scale (1 / world_size after a sum all-reduce); param_lp (optional, dtype lp_dtype) receives the rounded copy the next
 * forward uses. */
int setok_adamw(void* stream, int lp_dtype, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_lp,
                int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale);

/* ---- LLM prefill of BASELINE config 5 (last row) -------------------------------------------------------------
 * SetokimLlamaForCausalLM.forward (src/model/language_model/setokim_llama.py:130-143) hands the spliced embeddings to
 * `self.model` — HuggingFace `transformers` LlamaModel (third party, pinned 4.46.3 by the reference) — and `self.lm_head`.
 * Its Linears are setok_linear calls; these are the other pieces of LlamaDecoderLayer.forward (eager path). */

/* LlamaRMSNorm.forward: y = weight * (x * rsqrt(mean(x^2) + eps)).to(dtype) — statistics in fp32, the normalised value rounded
 * to the activation dtype before the weight multiply. */
int setok_rmsnorm(void* stream, int dtype, const void* x, const float* weight, void* y, int rows, int C, float eps);

/* apply_rotary_pos_emb (rotate_half convention, default rope: inv_freq = theta^(-2i/Dh), cos / sin in fp32 rounded to dtype) in
 * place on the q and k thirds of qkv: (rows, 3*H*Dh) laid out [q | k | v]; position_ids: int64[rows]. */
int setok_rope(void* stream, int dtype, void* qkv, const int64_t* position_ids, int rows, int H, int Dh, float theta);
/* ... with grouped-query attention (LlamaConfig.num_key_value_heads = Hkv < H, H % Hkv == 0: Llama-2-70B, Llama-3, Mistral): rows of
 * (H + 2*Hkv)*Dh elements [q: H heads | k: Hkv heads | v: Hkv heads]; the H + Hkv heads of q and k are rotated.  Hkv == H is setok_rope. */
int setok_rope_gqa(void* stream, int dtype, void* qkv, const int64_t* position_ids, int rows, int H, int Hkv, int Dh, float theta);

/* LlamaMLP's act_fn(gate_proj(x)) * up_proj(x) on a fused (rows, 2*F) buffer [gate | up] -> (rows, F); act_fn = SiLU. */
int setok_swiglu(void* stream, int dtype, const void* gate_up, void* out, int64_t rows, int F);
/* ... on a buffer of INTERLEAVED pairs: gate_up_pairs (rows, 2 F) with (gate_j, up_j) in columns 2 j, 2 j + 1 — the output layout of a Linear whose weight
 * rows are interleaved the same way (setok_linear_swiglu's).  Same arithmetic, same bits as setok_swiglu on the de-interleaved buffer. */
int setok_swiglu_pairs(void* stream, int dtype, const void* gate_up_pairs, void* out, int64_t rows, int F);
/* out (M, F) = act_fn(A Wg^T) * (A Wu^T) in ONE launch: the gate|up Linear of LlamaMLP (HF modeling_llama.py LlamaMLP.forward, reached from
 * /root/reference/src/model/language_model/setokim_llama.py:130-143) with SwiGLU in the GEMM's epilogue — the (M, 2 F) intermediate is never written.
 * W_pairs (2 F, K): row 2 j = gate_proj.weight[j], row 2 j + 1 = up_proj.weight[j].  torch's 16-bit rounding points are kept (gate and up rounded to the
 * element type, the activation rounded, the product rounded), so the result equals setok_linear(W_pairs) followed by setok_swiglu_pairs bit for bit.
 * 16-bit element types; M % 256 == 0, (2 F) % 256 == 0, K % 64 == 0, K >= 128, else SETOK_EUNSUPPORTED (send those rows through the unfused pair). */
int setok_linear_swiglu(void* stream, int dtype, const void* A, int64_t lda, const void* W_pairs, void* out, int64_t ldo, int M, int F, int K);

/* Causal self-attention of LlamaAttention (eager_attention_forward: softmax(q k^T * scale + mask) v, fp32 softmax) over B
 * sequences of T rows of a fused [q | k | v] buffer (num_key_value_heads == num_attention_heads).  Query i of a sequence sees key
 * j iff j <= i and key_mask[b*T + j] != 0 (key_mask NULL = all tokens): the causal + padding mask HF builds from attention_mask.
 * A query that sees no token at all (padding before a sequence's first token) gets zeros (HF gives such rows an arbitrary uniform
 * mix; they are padding and masked out of the loss, setokim_llama.py:149-152). */
int setok_attention_causal(void* stream, int dtype, const void* qkv, const uint8_t* key_mask, void* out, int B, int T, int H, int Dh,
                           float scale);
/* ... with grouped-query attention (HF repeat_kv, modeling_llama.py: query head h reads key / value head h / (H / Hkv)): qkv rows of
 * (H + 2*Hkv)*Dh elements [q | k | v], out rows of H*Dh.  Hkv == H is setok_attention_causal (same kernels, same bits). */
int setok_attention_causal_gqa(void* stream, int dtype, const void* qkv, const uint8_t* key_mask, void* out, int B, int T, int H, int Hkv, int Dh,
                               float scale);

/* The language-model loss of SetokimLlamaForCausalLM.forward (setokim_llama.py:145-160): logits (B*T rows of V, row stride ld) are read as
 * fp32; position t predicts labels[t + 1]; positions with attention_mask[t + 1] == 0 (NULL = none) or labels[t + 1] == ignore_index are left
 * out; out[0] = mean cross entropy over the rest (NaN if none), out[1] = their number.  row_ws: 2*B*T floats of workspace.  Deterministic. */
int setok_lm_loss(void* stream, int dtype, const void* logits, int64_t ld, const int64_t* labels, const uint8_t* attention_mask, int B, int T,
                  int V, int ignore_index, float* row_ws, float* out);

#ifdef __cplusplus
}
#endif
#endif /* SETOK_HIP_H */
