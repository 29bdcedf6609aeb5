// norm_attn.hip — LayerNorm and the generic (any head dim, any dtype) varlen attention.
#include "common.h"

// --------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, fp32 statistics (two-pass: mean, then centred variance).
// Bandwidth-bound: 16-byte accesses, row re-reads hit L1.
// --------------------------------------------------------------------------------------------
// The two per-element steps of every LayerNorm kernel here, with their fused multiply-adds written out and nothing else contracted: left to the
// compiler the generic kernel and a register-resident one rounded a row's centred sum of squares differently now and then (round 5: 10 rows of
// 7776 at C = 768 differed in one element) — a row alone (generic kernel) must equal the row in a batch.
__device__ inline float ln_sq_acc(float x, float mean, float q) {
#pragma clang fp contract(off)
    const float d = x - mean;
    return __builtin_fmaf(d, d, q);
}
__device__ inline float ln_apply(float x, float mean, float rstd, float g, float b) {
#pragma clang fp contract(off)
    const float t = (x - mean) * rstd;
    return __builtin_fmaf(t, g, b);
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* y,
                                                        int rows, int C, float eps, const int32_t* __restrict__ rows_dev) {
    constexpr int V = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rows_dev) rows = min(rows, *rows_dev);                     // a ragged stage's device-side row count (setok_encode)
    if (row >= rows) return;
    const T* xr = x + (int64_t)row * C;
    T* yr = y + (int64_t)row * C;
    float buf[V];
    float s = 0.f;
    for (int c = lane * V; c < C; c += 64 * V) {
        ld_vec<T>(xr + c, buf);
#pragma unroll
        for (int i = 0; i < V; ++i) s += buf[i];
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane * V; c < C; c += 64 * V) {
        ld_vec<T>(xr + c, buf);
#pragma unroll
        for (int i = 0; i < V; ++i) q = ln_sq_acc(buf[i], mean, q);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    for (int c = lane * V; c < C; c += 64 * V) {
        ld_vec<T>(xr + c, buf);
#pragma unroll
        for (int i = 0; i < V; ++i) buf[i] = ln_apply(buf[i], mean, rstd, gamma[c + i], beta[c + i]);
        st_vec<T>(yr + c, buf);
    }
}

// Rows of C = NCH * 64 * VEC elements (ViT-L: 1024 bf16 = 2 chunks per lane): the row stays in registers between the three passes, and a
// wave keeps ITS columns of gamma / beta in registers across the rows it walks (read per row they are 4x the row's own bytes through L1).
// Same arithmetic and summation order as the generic kernel above (results are bit-identical).
template <typename T, int NCH, int CT = NCH * 64 * Elem<T>::VEC>        // CT: the row length when it is not a whole number of 64-lane chunks (768, 1280: the
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const T* x, const float* __restrict__ gamma,      // decoder's widths) — the lanes past the end of the last chunk sit out
                                                             const float* __restrict__ beta, T* y, int rows, float eps,
                                                             const int32_t* __restrict__ rows_dev) {
    constexpr int V = Elem<T>::VEC, C = CT;
    static_assert(CT % V == 0 && CT <= NCH * 64 * V && CT > (NCH - 1) * 64 * V, "CT: a multiple of the vector width inside the last chunk");
    const int lane = threadIdx.x & 63;
    const int wave0 = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    if (rows_dev) rows = min(rows, *rows_dev);
    if (wave0 >= rows) return;
    const bool last_ok = ((NCH - 1) * 64 + lane) * V < CT;              // does this lane hold elements of the last chunk?
    float g[NCH][V], b[NCH][V];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const bool ok = k < NCH - 1 || last_ok;
            g[k][i] = ok ? gamma[(k * 64 + lane) * V + i] : 0.f; b[k][i] = ok ? beta[(k * 64 + lane) * V + i] : 0.f;
        }
    for (int row = wave0; row < rows; row += nwaves) {
        const T* xr = x + (int64_t)row * C;
        T* yr = y + (int64_t)row * C;
        float buf[NCH][V];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (k < NCH - 1 || last_ok) ld_vec<T>(xr + (k * 64 + lane) * V, buf[k]);
            else {
#pragma unroll
                for (int i = 0; i < V; ++i) buf[k][i] = 0.f;
            }
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int i = 0; i < V; ++i) if (k < NCH - 1 || last_ok) s += buf[k][i];
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int i = 0; i < V; ++i) if (k < NCH - 1 || last_ok) q = ln_sq_acc(buf[k][i], mean, q);
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
#pragma unroll
            for (int i = 0; i < V; ++i) buf[k][i] = ln_apply(buf[k][i], mean, rstd, g[k][i], b[k][i]);
            if (k < NCH - 1 || last_ok) st_vec<T>(yr + (k * 64 + lane) * V, buf[k]);
        }
    }
}

template <typename T, int NCH, int CT = NCH * 64 * Elem<T>::VEC>
static void launch_ln_rows(hipStream_t s, const void* x, const float* gamma, const float* beta, void* y, int rows, float eps, const int32_t* rows_dev) {
    const int grid = min(cdiv(rows, 4), 256 * 8);
    layernorm_rows_kernel<T, NCH, CT><<<grid, 256, 0, s>>>((const T*)x, gamma, beta, (T*)y, rows, eps, rows_dev);
}

// rows_dev: optional device-side row count (<= rows); rows beyond it are neither read nor written.
int setok_layernorm_dev(void* stream, int dtype, const void* x, const float* gamma, const float* beta,
                        void* y, int rows, int C, float eps, const int32_t* rows_dev) {
    SETOK_CHECK_ARG(x && y && gamma && beta, "setok_layernorm: null operand");
    SETOK_CHECK_ARG(rows >= 0 && C > 0 && C % 8 == 0, "setok_layernorm: C=%d must be a positive multiple of 8", C);
    if (rows == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(cdiv(rows, 4));
    if (dtype == SETOK_BF16 && (C == 512 || C == 768 || C == 1024 || C == 1280 || C == 1536 || C == 2048) && rows >= 1024) {
        if (C == 512) launch_ln_rows<bf16, 1>(s, x, gamma, beta, y, rows, eps, rows_dev);
        else if (C == 768) launch_ln_rows<bf16, 2, 768>(s, x, gamma, beta, y, rows, eps, rows_dev);         // round 5: the reconstruction decoder's width (cfg3: 49 LayerNorms of
        else if (C == 1280) launch_ln_rows<bf16, 3, 1280>(s, x, gamma, beta, y, rows, eps, rows_dev);       //  82 944 x 768 per step ran on the generic kernel at 4.2 TB/s)
        else if (C == 1024) launch_ln_rows<bf16, 2>(s, x, gamma, beta, y, rows, eps, rows_dev);
        else if (C == 1536) launch_ln_rows<bf16, 3>(s, x, gamma, beta, y, rows, eps, rows_dev);
        else launch_ln_rows<bf16, 4>(s, x, gamma, beta, y, rows, eps, rows_dev);
        SETOK_CHECK_LAUNCH("setok_layernorm");
        return SETOK_OK;
    }
    if (dtype == SETOK_BF16) layernorm_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)x, gamma, beta, (bf16*)y, rows, C, eps, rows_dev);
    else if (dtype == SETOK_F32) layernorm_kernel<float><<<grid, 256, 0, s>>>((const float*)x, gamma, beta, (float*)y, rows, C, eps, rows_dev);
    else return setok_fail(SETOK_EINVAL, "setok_layernorm: bad dtype %d", dtype);
    SETOK_CHECK_LAUNCH("setok_layernorm");
    return SETOK_OK;
}

extern "C" int setok_layernorm(void* stream, int dtype, const void* x, const float* gamma, const float* beta,
                               void* y, int rows, int C, float eps) {
    return setok_layernorm_dev(stream, dtype, x, gamma, beta, y, rows, C, eps, nullptr);
}

// --------------------------------------------------------------------------------------------
// The two small pieces of a LayerNorm that is folded into its consuming GEMM (setok_linear_ln):
//   row statistics {mean, rstd} — the first two passes of the kernels above, same summation order, no normalised copy written;
//   the weight fold, once per weight load:  W' = bf16(gamma * W),  c = W' 1 (fp32, ascending k),  b' = b + W beta.
// --------------------------------------------------------------------------------------------
// stats row = 8 floats: [0] = (-mean hi, lo) and [1] = (1 / rstd hi, lo) as bf16 pairs — the compact form of the activation-side fragment
// [mh, ml, mh, ml, sh, sl, sh, sl] (the GEMM duplicates the two words) — [2] = rstd (again), [3] = 0, [4] = rstd, [5] = mean, [6], [7] = 0
__device__ inline void write_row_stats(float* stats, int64_t row, float mean, float rstd) {
    const f32x4 fr = __builtin_bit_cast(f32x4, ln_row_frag(mean, rstd));
    f32x4* o = reinterpret_cast<f32x4*>(stats + 8 * row);
    const f32x4 head = {fr[0], fr[2], rstd, 0.f};             // (rstd once more as word 2: the ping-pong GEMM fetches fragment + rstd as ONE 16-byte load)
    o[0] = head;
    const f32x4 tail = {rstd, mean, 0.f, 0.f};
    o[1] = tail;
}

template <typename T>
__global__ __launch_bounds__(256) void row_stats_kernel(const T* __restrict__ x, float* __restrict__ stats, int rows, int C, float eps) {
    constexpr int V = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int wave0 = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (int row = wave0; row < rows; row += nwaves) {
        const T* xr = x + (int64_t)row * C;
        float buf[V];
        float s = 0.f;
        for (int c = lane * V; c < C; c += 64 * V) {
            ld_vec<T>(xr + c, buf);
#pragma unroll
            for (int i = 0; i < V; ++i) s += buf[i];
        }
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
        for (int c = lane * V; c < C; c += 64 * V) {
            ld_vec<T>(xr + c, buf);                                         // L1 hit
#pragma unroll
            for (int i = 0; i < V; ++i) { const float d = buf[i] - mean; q += d * d; }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
        if (lane == 0) write_row_stats(stats, row, mean, rstd);
    }
}

// rows of exactly CT bf16 (1024: the ViT-L width; 768 / 1280: the pixel decoder's and ViT-H's): the row stays in registers between the passes (as
// layernorm_rows_kernel<bf16, NCH, CT>); lanes past the end of the last 512-element chunk hold nothing.  (The 1024 instantiation keeps its round-2 name.)
template <int CT>
__device__ __forceinline__ void row_stats_rows(const bf16* __restrict__ x, float* __restrict__ stats, int rows, float eps) {
    constexpr int V = 8, NCH = (CT + 511) / 512, C = CT;
    static_assert(CT % 8 == 0, "whole 16-byte pieces");
    const int lane = threadIdx.x & 63;
    const bool last_ok = ((NCH - 1) * 64 + lane) * V < C;                        // does this lane hold a piece of the last chunk?
    const int wave0 = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (int row = wave0; row < rows; row += nwaves) {
        const bf16* xr = x + (int64_t)row * C;
        float buf[NCH][V];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (k + 1 < NCH || C % 512 == 0 || last_ok) ld_vec<bf16>(xr + (k * 64 + lane) * V, buf[k]);
            else {
#pragma unroll
                for (int i = 0; i < V; ++i) buf[k][i] = 0.f;
            }
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int i = 0; i < V; ++i) s += buf[k][i];
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k)
            if (k + 1 < NCH || C % 512 == 0 || last_ok) {
#pragma unroll
                for (int i = 0; i < V; ++i) { const float d = buf[k][i] - mean; q += d * d; }
            }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
        if (lane == 0) write_row_stats(stats, row, mean, rstd);
    }
}
__global__ __launch_bounds__(256) void row_stats_1024_kernel(const bf16* __restrict__ x, float* __restrict__ stats, int rows, float eps) { row_stats_rows<1024>(x, stats, rows, eps); }
template <int CT>
__global__ __launch_bounds__(256) void row_stats_ct_kernel(const bf16* __restrict__ x, float* __restrict__ stats, int rows, float eps) { row_stats_rows<CT>(x, stats, rows, eps); }

extern "C" int setok_row_stats(void* stream, int dtype, const void* x, float* stats, int rows, int C, float eps) {
    SETOK_CHECK_ARG(x && stats, "setok_row_stats: null operand");
    SETOK_CHECK_ARG(rows >= 0 && C > 0 && C % 8 == 0, "setok_row_stats: C=%d must be a positive multiple of 8", C);
    if (rows == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    const int grid = min(cdiv(rows, 4), 256 * 8);
    if (dtype == SETOK_BF16 && C == 1024) row_stats_1024_kernel<<<grid, 256, 0, s>>>((const bf16*)x, stats, rows, eps);
    else if (dtype == SETOK_BF16 && C == 768) row_stats_ct_kernel<768><<<grid, 256, 0, s>>>((const bf16*)x, stats, rows, eps);
    else if (dtype == SETOK_BF16 && C == 1280) row_stats_ct_kernel<1280><<<grid, 256, 0, s>>>((const bf16*)x, stats, rows, eps);
    else if (dtype == SETOK_BF16) row_stats_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)x, stats, rows, C, eps);
    else if (dtype == SETOK_F32) row_stats_kernel<float><<<grid, 256, 0, s>>>((const float*)x, stats, rows, C, eps);
    else return setok_fail(SETOK_EINVAL, "setok_row_stats: bad dtype %d", dtype);
    SETOK_CHECK_LAUNCH("setok_row_stats");
    return SETOK_OK;
}

// one wave per output row n of W (N, K)
__global__ __launch_bounds__(256) void ln_fold_kernel(const bf16* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ bias, bf16* __restrict__ Wg, float* __restrict__ colsum,
                                                      float* __restrict__ bias_folded, float* __restrict__ colfrag, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float c = 0.f, bb = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = (float)W[(int64_t)n * K + k];
        const bf16 wg = (bf16)(w * gamma[k]);
        Wg[(int64_t)n * K + k] = wg;
        c += (float)wg;
        bb += w * beta[k];
    }
    c = wave_sum(c); bb = wave_sum(bb);
    if (lane == 0) {
        const float bf = (bias ? bias[n] : 0.f) + bb;
        colsum[n] = c; bias_folded[n] = bf;
        reinterpret_cast<f32x4*>(colfrag)[n] = __builtin_bit_cast(f32x4, ln_col_frag(c, bf));
    }
}

extern "C" int setok_ln_fold(void* stream, const void* W, const float* gamma, const float* beta, const float* bias, void* w_gamma,
                             float* w_colsum, float* bias_folded, float* col_frag, int N, int K) {
    SETOK_CHECK_ARG(W && gamma && beta && w_gamma && w_colsum && bias_folded && col_frag && N > 0 && K > 0, "setok_ln_fold: bad argument");
    ln_fold_kernel<<<cdiv(N, 4), 256, 0, (hipStream_t)stream>>>((const bf16*)W, gamma, beta, bias, (bf16*)w_gamma, w_colsum, bias_folded, col_frag, N, K);
    SETOK_CHECK_LAUNCH("setok_ln_fold");
    return SETOK_OK;
}

// --------------------------------------------------------------------------------------------
// Generic varlen attention: one wave per (query row, head).  Exact-softmax (max-subtracted, fp32).
// This is the any-shape / parity-mode path (fp32, head dim 512 of the cluster encoders, ragged
// segments); the bf16 ViT shape has its own MFMA kernel in attn_vit.hip.
// --------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void attn_generic_kernel(const T* __restrict__ qbase, int64_t ldq, const T* __restrict__ kbase,
                                                          const T* __restrict__ vbase, int64_t ld,
                                                          const int32_t* __restrict__ seg_offsets, int n_segs, int seg_len,
                                                          int q_len, T* __restrict__ out, int64_t ldo, int rows, int H,
                                                          int Dh, float scale) {
    constexpr int V = Elem<T>::VEC;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* qs = sm;                 // Dh
    float* ps = sm + Dh;            // seg_len (upper bound on the segment length)
    const int row = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    int s0, s1;
    if (q_len > 0) {                                                // cross-attention: query row -> its segment's key rows
        const int seg = row / q_len;
        if (seg_offsets) { s0 = seg_offsets[seg]; s1 = seg_offsets[seg + 1]; }
        else { s0 = seg * seg_len; s1 = s0 + seg_len; }
        if (s1 - s0 > seg_len) s1 = s0 + seg_len;                   // host contract: seg_len bounds every segment
    } else if (seg_offsets) {
        int lo = 0, hi = n_segs;                                    // find s with off[s] <= row < off[s+1]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_offsets[mid] <= row) lo = mid; else hi = mid; }
        s0 = seg_offsets[lo]; s1 = seg_offsets[lo + 1];
        if (row >= s1 || row < s0) return;                          // row beyond the last segment
    } else {
        s0 = (row / seg_len) * seg_len; s1 = min(s0 + seg_len, rows);
    }
    const int n = s1 - s0;
    const T* qp = qbase + (int64_t)row * ldq + h * Dh;
    const T* kh = kbase + h * Dh;
    const T* vh = vbase + h * Dh;
    T* orow = out + (int64_t)row * ldo + h * Dh;
    float buf[V];
    if (n <= 64 && Dh % (64 * V) == 0) {
        // Short segments (the clusters of the SeTok head average ~7 tokens, the inter-encoder ~36): all 64 lanes
        // share every key — each lane owns a V-wide slice of the head dimension, K/V rows are read as coalesced
        // 16-byte pieces, the dot products are wave reductions, key j's probability lives in lane j.
        constexpr int MAXC = 2;                                     // head dim up to 2 * 64 * V (1024 for bf16)
        const int nc = Dh / (64 * V);
        float qreg[MAXC][V], oreg[MAXC][V];
        for (int c = 0; c < nc && c < MAXC; ++c) {
            ld_vec<T>(qp + (c * 64 + lane) * V, qreg[c]);
#pragma unroll
            for (int i = 0; i < V; ++i) oreg[c][i] = 0.f;
        }
        float sc = -INFINITY;
        for (int j = 0; j < n; ++j) {
            const T* kp = kh + (int64_t)(s0 + j) * ld;
            float acc = 0.f;
            for (int c = 0; c < nc && c < MAXC; ++c) {
                ld_vec<T>(kp + (c * 64 + lane) * V, buf);
#pragma unroll
                for (int i = 0; i < V; ++i) acc = fmaf(qreg[c][i], buf[i], acc);
            }
            acc = wave_sum(acc) * scale;
            if (lane == j) sc = acc;
        }
        const float mxs = wave_max(sc);
        const float e = (lane < n) ? expf(sc - mxs) : 0.f;
        const float inv = 1.0f / wave_sum(e);
        for (int j = 0; j < n; ++j) {
            const float p = __shfl(e, j, 64);
            const T* vp = vh + (int64_t)(s0 + j) * ld;
            for (int c = 0; c < nc && c < MAXC; ++c) {
                ld_vec<T>(vp + (c * 64 + lane) * V, buf);
#pragma unroll
                for (int i = 0; i < V; ++i) oreg[c][i] = fmaf(p, buf[i], oreg[c][i]);
            }
        }
        for (int c = 0; c < nc && c < MAXC; ++c) {
#pragma unroll
            for (int i = 0; i < V; ++i) oreg[c][i] *= inv;
            st_vec<T>(orow + (c * 64 + lane) * V, oreg[c]);
        }
        return;
    }
    for (int d = lane * V; d < Dh; d += 64 * V) {
        ld_vec<T>(qp + d, buf);
#pragma unroll
        for (int i = 0; i < V; ++i) qs[d + i] = buf[i];
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) {
        const T* kp = kh + (int64_t)(s0 + j) * ld;
        float acc = 0.f;
        for (int d = 0; d < Dh; d += V) {
            ld_vec<T>(kp + d, buf);
#pragma unroll
            for (int i = 0; i < V; ++i) acc = fmaf(qs[d + i], buf[i], acc);
        }
        acc *= scale;
        ps[j] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) { const float e = expf(ps[j] - mx); ps[j] = e; sum += e; }
    sum = wave_sum(sum);
    __syncthreads();
    const float inv = 1.0f / sum;
    for (int d = lane * V; d < Dh; d += 64 * V) {
        float o[V];
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = 0.f;
        const T* vp = vh + (int64_t)s0 * ld + d;
        for (int j = 0; j < n; ++j) {
            ld_vec<T>(vp + (int64_t)j * ld, buf);
            const float p = ps[j];
#pragma unroll
            for (int i = 0; i < V; ++i) o[i] = fmaf(p, buf[i], o[i]);
        }
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] *= inv;
        st_vec<T>(orow + d, o);
    }
}

int setok_attention_vit_bf16(hipStream_t s, const bf16* qkv, bf16* out, int n_imgs, int T, int H, int Dh, float scale);  // attn_vit.hip
int setok_attention_seg_bf16(hipStream_t s, const bf16* qkv, const int32_t* seg_offsets, int n_segs, int max_len, bf16* out, int rows, int H, int Dh, float scale);  // attn_seg.hip
int setok_cross_attention_bf16(hipStream_t s, const bf16* q, int64_t ldq, const bf16* k, const bf16* v, int64_t ldkv, const int32_t* kv_offsets,
                               int n_segs, int q_len, int max_kv, bf16* out, int64_t ldo, int H, float scale);           // attn_vit.hip

extern "C" int setok_attention(void* stream, int dtype, const void* qkv, const int32_t* seg_offsets, int n_segs,
                               int seg_len, void* out, int rows, int H, int Dh, float scale) {
    SETOK_CHECK_ARG(qkv && out, "setok_attention: null operand");
    SETOK_CHECK_ARG(rows >= 0 && H > 0 && Dh > 0 && Dh % 8 == 0, "setok_attention: bad H=%d Dh=%d", H, Dh);
    SETOK_CHECK_ARG(seg_len > 0, "setok_attention: seg_len (segment length / upper bound) must be > 0");
    SETOK_CHECK_ARG(seg_offsets == nullptr || n_segs > 0, "setok_attention: n_segs must be > 0 with seg_offsets");
    if (rows == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SETOK_BF16 && !seg_offsets && (Dh == 64 || Dh == 48) && rows % seg_len == 0) {
        const int rc = setok_attention_vit_bf16(s, (const bf16*)qkv, (bf16*)out, rows / seg_len, seg_len, H, Dh, scale);
        if (rc != SETOK_EUNSUPPORTED) return rc;
    }
    if (dtype == SETOK_BF16 && seg_offsets && Dh == 512) {
        const int rc = setok_attention_seg_bf16(s, (const bf16*)qkv, seg_offsets, n_segs, seg_len, (bf16*)out, rows, H, Dh, scale);
        if (rc != SETOK_EUNSUPPORTED) return rc;
    }
    const size_t smem = (size_t)(Dh + seg_len) * sizeof(float);
    SETOK_CHECK_ARG(smem <= 64 * 1024, "setok_attention: Dh + seg_len too large for the generic kernel");
    dim3 grid(rows, H);
    const int64_t C = (int64_t)H * Dh;
    if (dtype == SETOK_BF16) {
        const bf16* p = (const bf16*)qkv;
        attn_generic_kernel<bf16><<<grid, 64, smem, s>>>(p, 3 * C, p + C, p + 2 * C, 3 * C, seg_offsets, n_segs, seg_len, 0, (bf16*)out, C, rows, H, Dh, scale);
    } else if (dtype == SETOK_F32) {
        const float* p = (const float*)qkv;
        attn_generic_kernel<float><<<grid, 64, smem, s>>>(p, 3 * C, p + C, p + 2 * C, 3 * C, seg_offsets, n_segs, seg_len, 0, (float*)out, C, rows, H, Dh, scale);
    } else return setok_fail(SETOK_EINVAL, "setok_attention: bad dtype %d", dtype);
    SETOK_CHECK_LAUNCH("setok_attention");
    return SETOK_OK;
}

extern "C" int setok_cross_attention(void* stream, int dtype, const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv,
                                     const int32_t* kv_offsets, int n_segs, int q_len, int max_kv, void* out, int64_t ldo,
                                     int H, int Dh, float scale) {
    SETOK_CHECK_ARG(q && k && v && out, "setok_cross_attention: null operand");
    SETOK_CHECK_ARG(H > 0 && Dh > 0 && Dh % 8 == 0, "setok_cross_attention: bad H=%d Dh=%d", H, Dh);
    SETOK_CHECK_ARG(n_segs >= 0 && q_len > 0 && max_kv > 0, "setok_cross_attention: bad n_segs=%d q_len=%d max_kv=%d", n_segs, q_len, max_kv);
    SETOK_CHECK_ARG(ldq >= (int64_t)H * Dh && ldkv >= (int64_t)H * Dh && ldo >= (int64_t)H * Dh && ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 8 == 0,
                    "setok_cross_attention: bad leading dimensions");
    if (n_segs == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SETOK_BF16 && Dh == 64) {
        const int rc = setok_cross_attention_bf16(s, (const bf16*)q, ldq, (const bf16*)k, (const bf16*)v, ldkv, kv_offsets, n_segs, q_len,
                                                  max_kv, (bf16*)out, ldo, H, scale);
        if (rc != SETOK_EUNSUPPORTED) return rc;
    }
    const size_t smem = (size_t)(Dh + max_kv) * sizeof(float);
    SETOK_CHECK_ARG(smem <= 64 * 1024, "setok_cross_attention: Dh + max_kv too large for the generic kernel");
    const int rows = n_segs * q_len;
    dim3 grid(rows, H);
    if (dtype == SETOK_BF16)
        attn_generic_kernel<bf16><<<grid, 64, smem, s>>>((const bf16*)q, ldq, (const bf16*)k, (const bf16*)v, ldkv, kv_offsets, n_segs, max_kv,
                                                         q_len, (bf16*)out, ldo, rows, H, Dh, scale);
    else if (dtype == SETOK_F32)
        attn_generic_kernel<float><<<grid, 64, smem, s>>>((const float*)q, ldq, (const float*)k, (const float*)v, ldkv, kv_offsets, n_segs,
                                                          max_kv, q_len, (float*)out, ldo, rows, H, Dh, scale);
    else return setok_fail(SETOK_EINVAL, "setok_cross_attention: bad dtype %d", dtype);
    SETOK_CHECK_LAUNCH("setok_cross_attention");
    return SETOK_OK;
}
