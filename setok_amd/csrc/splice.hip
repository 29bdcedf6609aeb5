// splice.hip — prepare_inputs_labels_for_multimodal (src/model/setokim_arch.py:213-355) without its Python loops:
// the ragged splice of every image's L_i tokens into the text embeddings, on the device.
//
//   setok_splice_lengths : per sequence, tokens kept by the mask, image placeholders, index of its first image, new length
//                          (one host read of B ints follows: the padded width is the batch maximum, :317)
//   setok_splice_plan    : per output position, where its row comes from (embedding-table row / packed image-token row /
//                          zero padding) + new labels, attention mask, position ids
//   setok_splice_rows    : the row copies (16-byte pieces, one wave per output row) — the only step that moves real bytes:
//                          B x max_len x D elements read once and written once
#include "common.h"

namespace {

constexpr int SRC_PAD = INT32_MIN;

// n_valid[b], n_img[b], and the position of the sequence's first kept id that the embedding table cannot serve (T = none): a kept id
// that is not the placeholder must lie in [0, vocab) — the reference's `embed_tokens` raises IndexError otherwise (:273), e.g. for a
// TARGET_TOKEN_INDEX (-300) that leaked into input_ids; unchecked, a negative id would be decoded as an image-token row by splice_rows.
__global__ __launch_bounds__(256) void splice_count_kernel(const int64_t* __restrict__ ids, const uint8_t* __restrict__ mask, int T,
                                                           int64_t image_token, int64_t vocab, int B, int32_t* __restrict__ cnt) {
    __shared__ int sv, si, sbad;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) { sv = 0; si = 0; sbad = T; }
    __syncthreads();
    int v = 0, im = 0, bad = T;
    for (int t = threadIdx.x; t < T; t += 256) {
        const bool keep = mask ? mask[(int64_t)b * T + t] != 0 : true;
        if (keep) {
            const int64_t id = ids[(int64_t)b * T + t];
            ++v; im += id == image_token;
            if (vocab > 0 && id != image_token && (id < 0 || id >= vocab)) bad = min(bad, t);
        }
    }
    v = wave_sum_i(v); im = wave_sum_i(im);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&sv, v); atomicAdd(&si, im); }
    if (bad < T) atomicMin(&sbad, bad);
    __syncthreads();
    if (threadIdx.x == 0) { cnt[2 * b] = sv; cnt[2 * b + 1] = si; cnt[2 * B + b] = sbad; }
}

// One workgroup: img_start[b] = sum_{b' < b} max(n_img[b'], 1)  (a sequence without a placeholder still consumes an index,
// setokim_arch.py:264-271), new length = kept - placeholders + rows of its images, truncated (:311-314).
__global__ __launch_bounds__(256) void splice_len_kernel(const int32_t* __restrict__ cnt, const int32_t* __restrict__ img_offsets,
                                                         int n_images, int B, int T, int max_length, int32_t* __restrict__ seq_len,
                                                         int32_t* __restrict__ img_start, int32_t* __restrict__ status) {
    __shared__ int part[256];
    __shared__ int first_bad;
    const int tid = threadIdx.x;
    const int per = (B + 255) / 256, lo = tid * per, hi = min(lo + per, B);
    int c = 0;
    if (tid == 0) first_bad = B;
    __syncthreads();
    for (int b = lo; b < hi; ++b) { c += max(cnt[2 * b + 1], 1); if (cnt[2 * B + b] < T) atomicMin(&first_bad, b); }
    part[tid] = c;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < 256; ++t) { const int x = part[t]; part[t] = run; run += x; }
        status[0] = run > n_images ? 1 : 0;                         // the reference would raise IndexError at image_features[cur_image_idx]
        status[1] = run;
        status[2] = first_bad < B ? 1 : 0;                          // ... and at embed_tokens(id) for an id outside the table
        status[3] = first_bad < B ? first_bad * T + cnt[2 * B + first_bad] : -1;      // flat position of the first such id
    }
    __syncthreads();
    int s = part[tid];
    for (int b = lo; b < hi; ++b) {
        const int n = cnt[2 * b + 1];
        img_start[b] = s;
        int rows = 0;
        if (n > 0 && s + n <= n_images) rows = img_offsets[s + n] - img_offsets[s];
        int len = cnt[2 * b] - n + rows;
        if (max_length > 0) len = min(len, max_length);
        seq_len[b] = len;
        s += max(n, 1);
    }
}

// One workgroup per sequence.  src[b][p]: >= 0 embedding row (token id), < 0 packed image-token row -(r + 1), SRC_PAD zero row.
__global__ __launch_bounds__(256) void splice_plan_kernel(const int64_t* __restrict__ ids, const uint8_t* __restrict__ mask,
                                                          const int64_t* __restrict__ labels, int T, int64_t image_token,
                                                          int64_t ignore_index, int64_t target_index,
                                                          const int32_t* __restrict__ img_offsets, const int32_t* __restrict__ seq_len,
                                                          const int32_t* __restrict__ img_start, int max_len, int left_pad,
                                                          int32_t* __restrict__ src, int64_t* __restrict__ new_labels,
                                                          uint8_t* __restrict__ new_mask, int64_t* __restrict__ new_pos) {
    __shared__ int sc_img[256], sc_w[256];
    __shared__ int run_img, run_w;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int len = seq_len[b];
    const int shift = left_pad ? max_len - len : 0;                 // :322-337
    // padding first (defaults of :319-321), the kept range gets mask / position here as well
    for (int p = tid; p < max_len; p += 256) {
        const int q = p - shift;
        const bool in = q >= 0 && q < len;
        const int64_t o = (int64_t)b * max_len + p;
        if (!in) { src[o] = SRC_PAD; if (new_labels) new_labels[o] = ignore_index; }
        if (new_mask) new_mask[o] = in ? 1 : 0;
        if (new_pos) new_pos[o] = in ? q : 0;
    }
    if (tid == 0) { run_img = 0; run_w = 0; }
    __syncthreads();
    const int first_img = img_start[b];
    for (int t0 = 0; t0 < T; t0 += 256) {
        const int t = t0 + tid;
        bool keep = false, is_img = false;
        int64_t id = 0;
        if (t < T) {
            keep = mask ? mask[(int64_t)b * T + t] != 0 : true;
            id = ids[(int64_t)b * T + t];
            is_img = keep && id == image_token;
        }
        // exclusive scan of the image flags -> this placeholder's image index
        sc_img[tid] = is_img ? 1 : 0;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int v = tid >= d ? sc_img[tid - d] : 0;
            __syncthreads();
            sc_img[tid] += v;
            __syncthreads();
        }
        const int img_before = run_img + sc_img[tid] - (is_img ? 1 : 0);
        int w = 0, row0 = 0;
        if (is_img) { const int im = first_img + img_before; row0 = img_offsets[im]; w = img_offsets[im + 1] - row0; }
        else if (keep) w = 1;
        sc_w[tid] = w;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int v = tid >= d ? sc_w[tid - d] : 0;
            __syncthreads();
            sc_w[tid] += v;
            __syncthreads();
        }
        const int p0 = run_w + sc_w[tid] - w;                       // first output position of this token (before truncation / shift)
        if (keep) {
            if (is_img) {
                for (int r = 0; r < w && p0 + r < len; ++r) {
                    const int64_t o = (int64_t)b * max_len + shift + p0 + r;
                    src[o] = -(row0 + r + 1);
                    if (new_labels) new_labels[o] = ignore_index;   // :293
                }
            } else if (p0 < len) {
                const int64_t o = (int64_t)b * max_len + shift + p0;
                src[o] = (int32_t)id;
                if (new_labels) { const int64_t l = labels[(int64_t)b * T + t]; new_labels[o] = l == target_index ? ignore_index : l; }   // :344
            }
        }
        __syncthreads();
        if (tid == 255) { run_img += sc_img[255]; run_w += sc_w[255]; }
        __syncthreads();
    }
}

__global__ void splice_status_init_kernel(int32_t* status) { status[0] = 0; status[1] = INT32_MAX; }

// One wave per output row, 16-byte pieces.  An id the tables cannot serve (>= vocab, or an image-token row when no image tokens were
// handed over: e.g. an IMAGE_TOKEN_INDEX in a text-only call) never becomes an address: the row is zero-filled and, when `status` is given,
// status[0] = 1 and status[1] = the smallest such row (the reference's embed_tokens raises IndexError there, setokim_arch.py:273).
__global__ __launch_bounds__(256) void splice_rows_kernel(const int32_t* __restrict__ src, const char* __restrict__ embed,
                                                          const char* __restrict__ feats, char* __restrict__ out, int64_t rows,
                                                          int row_bytes, int vocab, int64_t feat_rows, int32_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const int s = src[r];
        const char* from = nullptr;
        bool bad = false;
        if (s >= 0) { if (s < vocab) from = embed + (int64_t)s * row_bytes; else bad = true; }
        else if (s != SRC_PAD) {
            const int64_t fr = -((int64_t)s + 1);
            if (feats && fr < feat_rows) from = feats + fr * row_bytes; else bad = true;
        }
        if (bad && status && lane == 0) { atomicExch(&status[0], 1); atomicMin(&status[1], (int)min(r, (int64_t)INT32_MAX)); }
        char* to = out + r * row_bytes;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        for (int o = lane * 16; o < row_bytes; o += 64 * 16)
            *reinterpret_cast<f32x4*>(to + o) = from ? *reinterpret_cast<const f32x4*>(from + o) : zero;
    }
}

// Backward of the row copies: d image_tokens[-(src[r] + 1)] = d out[r] (an image-token row is consumed by at most one output position,
// setokim_arch.py:290-293: plain copies, no conflicts; rows the truncation dropped keep the zeros the caller's memset left), and — when
// the embedding table is trained — d embed[src[r]] += d out[r] in fp32 (token ids repeat: hardware fp32 atomics, so the summation order of
// a repeated id's rows is not fixed; torch's own embedding backward makes the same trade).
template <typename T>
__global__ __launch_bounds__(256) void splice_rows_bwd_kernel(const int32_t* __restrict__ src, const T* __restrict__ dout, T* __restrict__ dfeats,
                                                              float* __restrict__ dembed, int64_t rows, int D, int vocab, int64_t feat_rows) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const int s = src[r];
        const T* from = dout + r * D;
        if (s >= 0) {
            if (dembed && s < vocab) {
                float* to = dembed + (int64_t)s * D;
                for (int c = lane; c < D; c += 64) unsafeAtomicAdd(to + c, (float)from[c]);
            }
        } else if (s != SRC_PAD && dfeats) {
            const int64_t fr = -((int64_t)s + 1);
            if (fr < feat_rows) {
                T* to = dfeats + fr * D;
                constexpr int V = 16 / sizeof(T);
                for (int c = lane * V; c < D; c += 64 * V)
                    *reinterpret_cast<f32x4*>(to + c) = *reinterpret_cast<const f32x4*>(from + c);
            }
        }
    }
}

}  // namespace

extern "C" int setok_splice_lengths(void* stream, const int64_t* input_ids, const uint8_t* attention_mask, int B, int T,
                                    int64_t image_token_index, int64_t vocab, const int32_t* img_offsets, int n_images, int max_length,
                                    int32_t* seq_len, int32_t* img_start, int32_t* status, int32_t* count_ws) {
    SETOK_CHECK_ARG(input_ids && img_offsets && seq_len && img_start && status && count_ws, "setok_splice_lengths: null operand");
    SETOK_CHECK_ARG(B > 0 && T > 0 && n_images >= 0 && vocab >= 0, "setok_splice_lengths: bad shape B=%d T=%d n_images=%d", B, T, n_images);
    hipStream_t s = (hipStream_t)stream;
    splice_count_kernel<<<B, 256, 0, s>>>(input_ids, attention_mask, T, image_token_index, vocab, B, count_ws);
    splice_len_kernel<<<1, 256, 0, s>>>(count_ws, img_offsets, n_images, B, T, max_length, seq_len, img_start, status);
    SETOK_CHECK_LAUNCH("setok_splice_lengths");
    return SETOK_OK;
}

extern "C" int setok_splice_plan(void* stream, const int64_t* input_ids, const uint8_t* attention_mask, const int64_t* labels,
                                 int B, int T, int64_t image_token_index, int64_t ignore_index, int64_t target_token_index,
                                 const int32_t* img_offsets, const int32_t* seq_len, const int32_t* img_start, int max_len,
                                 int left_pad, int32_t* src, int64_t* new_labels, uint8_t* new_mask, int64_t* new_position_ids) {
    SETOK_CHECK_ARG(input_ids && img_offsets && seq_len && img_start && src, "setok_splice_plan: null operand");
    SETOK_CHECK_ARG((labels == nullptr) == (new_labels == nullptr), "setok_splice_plan: labels and new_labels go together");
    SETOK_CHECK_ARG(B > 0 && T > 0 && max_len >= 0, "setok_splice_plan: bad shape");
    if (max_len == 0) return SETOK_OK;
    splice_plan_kernel<<<B, 256, 0, (hipStream_t)stream>>>(input_ids, attention_mask, labels, T, image_token_index, ignore_index,
                                                          target_token_index, img_offsets, seq_len, img_start, max_len, left_pad, src,
                                                          new_labels, new_mask, new_position_ids);
    SETOK_CHECK_LAUNCH("setok_splice_plan");
    return SETOK_OK;
}

extern "C" int setok_splice_rows(void* stream, int dtype, const int32_t* src, const void* embed_table, int vocab,
                                 const void* image_tokens, int64_t image_token_rows, void* out, int64_t rows, int D, int32_t* status) {
    SETOK_CHECK_ARG(src && embed_table && out, "setok_splice_rows: null operand");
    SETOK_CHECK_ARG(dtype == SETOK_BF16 || dtype == SETOK_F32, "setok_splice_rows: bad dtype %d", dtype);
    const int row_bytes = D * (dtype == SETOK_BF16 ? 2 : 4);
    SETOK_CHECK_ARG(rows >= 0 && D > 0 && row_bytes % 16 == 0 && vocab > 0 && image_token_rows >= 0, "setok_splice_rows: bad shape rows=%lld D=%d", (long long)rows, D);
    hipStream_t s = (hipStream_t)stream;
    if (status) splice_status_init_kernel<<<1, 1, 0, s>>>(status);
    if (rows == 0) return SETOK_OK;
    const int64_t want = (rows + 3) / 4;
    const int grid = (int)(want < 65536 * 4 ? want : 65536 * 4);
    splice_rows_kernel<<<grid, 256, 0, s>>>(src, (const char*)embed_table, (const char*)image_tokens, (char*)out, rows, row_bytes, vocab,
                                            image_tokens ? image_token_rows : 0, status);
    SETOK_CHECK_LAUNCH("setok_splice_rows");
    return SETOK_OK;
}

extern "C" int setok_splice_rows_bwd(void* stream, int dtype, const int32_t* src, const void* d_out, int64_t rows, int D,
                                     void* d_image_tokens, int64_t image_token_rows, float* d_embed, int vocab) {
    SETOK_CHECK_ARG(src && d_out, "setok_splice_rows_bwd: null operand");
    SETOK_CHECK_ARG(dtype == SETOK_BF16 || dtype == SETOK_F32, "setok_splice_rows_bwd: bad dtype %d", dtype);
    const int es = dtype == SETOK_BF16 ? 2 : 4;
    SETOK_CHECK_ARG(rows >= 0 && D > 0 && (D * es) % 16 == 0 && image_token_rows >= 0 && (!d_embed || vocab > 0), "setok_splice_rows_bwd: bad shape rows=%lld D=%d", (long long)rows, D);
    hipStream_t s = (hipStream_t)stream;
    if (d_image_tokens && image_token_rows > 0)
        SETOK_CHECK_ARG(hipMemsetAsync(d_image_tokens, 0, (size_t)image_token_rows * D * es, s) == hipSuccess, "setok_splice_rows_bwd: memset failed");
    if (rows == 0 || (!d_image_tokens && !d_embed)) return SETOK_OK;
    const int64_t want = (rows + 3) / 4;
    const int grid = (int)(want < 65536 * 4 ? want : 65536 * 4);
    if (dtype == SETOK_BF16)
        splice_rows_bwd_kernel<bf16><<<grid, 256, 0, s>>>(src, (const bf16*)d_out, (bf16*)d_image_tokens, d_embed, rows, D, vocab, image_token_rows);
    else
        splice_rows_bwd_kernel<float><<<grid, 256, 0, s>>>(src, (const float*)d_out, (float*)d_image_tokens, d_embed, rows, D, vocab, image_token_rows);
    SETOK_CHECK_LAUNCH("setok_splice_rows_bwd");
    return SETOK_OK;
}
