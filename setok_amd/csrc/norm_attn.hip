// norm_attn.hip — LayerNorm and the generic (any head dim, any dtype) varlen attention.
#include "common.h"

// --------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, fp32 statistics (two-pass: mean, then centred variance).
// Bandwidth-bound: 16-byte accesses, row re-reads hit L1.
// --------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* y,
                                                        int rows, int C, float eps) {
    constexpr int V = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
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
        for (int i = 0; i < V; ++i) { const float d = buf[i] - mean; q += d * d; }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    for (int c = lane * V; c < C; c += 64 * V) {
        ld_vec<T>(xr + c, buf);
#pragma unroll
        for (int i = 0; i < V; ++i) buf[i] = (buf[i] - mean) * rstd * gamma[c + i] + beta[c + i];
        st_vec<T>(yr + c, buf);
    }
}

extern "C" int setok_layernorm(void* stream, int dtype, const void* x, const float* gamma, const float* beta,
                               void* y, int rows, int C, float eps) {
    SETOK_CHECK_ARG(x && y && gamma && beta, "setok_layernorm: null operand");
    SETOK_CHECK_ARG(rows >= 0 && C > 0 && C % 8 == 0, "setok_layernorm: C=%d must be a positive multiple of 8", C);
    if (rows == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(cdiv(rows, 4));
    if (dtype == SETOK_BF16) layernorm_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)x, gamma, beta, (bf16*)y, rows, C, eps);
    else if (dtype == SETOK_F32) layernorm_kernel<float><<<grid, 256, 0, s>>>((const float*)x, gamma, beta, (float*)y, rows, C, eps);
    else return setok_fail(SETOK_EINVAL, "setok_layernorm: bad dtype %d", dtype);
    SETOK_CHECK_LAUNCH("setok_layernorm");
    return SETOK_OK;
}

// --------------------------------------------------------------------------------------------
// Generic varlen attention: one wave per (query row, head).  Exact-softmax (max-subtracted, fp32).
// This is the any-shape / parity-mode path (fp32, head dim 512 of the cluster encoders, ragged
// segments); the bf16 ViT shape has its own MFMA kernel in attn_vit.hip.
// --------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void attn_generic_kernel(const T* __restrict__ qkv, const int32_t* __restrict__ seg_offsets,
                                                          int n_segs, int seg_len, T* __restrict__ out, int rows, int H,
                                                          int Dh, float scale) {
    constexpr int V = Elem<T>::VEC;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* qs = sm;                 // Dh
    float* ps = sm + Dh;            // seg_len (upper bound on the segment length)
    const int row = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    int s0, s1;
    if (seg_offsets) {
        int lo = 0, hi = n_segs;                                    // find s with off[s] <= row < off[s+1]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_offsets[mid] <= row) lo = mid; else hi = mid; }
        s0 = seg_offsets[lo]; s1 = seg_offsets[lo + 1];
        if (row >= s1 || row < s0) return;                          // row beyond the last segment
    } else {
        s0 = (row / seg_len) * seg_len; s1 = min(s0 + seg_len, rows);
    }
    const int n = s1 - s0;
    const int64_t ld = 3LL * H * Dh;
    const T* qp = qkv + (int64_t)row * ld + h * Dh;
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
            const T* kp = qkv + (int64_t)(s0 + j) * ld + (int64_t)H * Dh + h * Dh;
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
            const T* vp = qkv + (int64_t)(s0 + j) * ld + 2LL * H * Dh + h * Dh;
            for (int c = 0; c < nc && c < MAXC; ++c) {
                ld_vec<T>(vp + (c * 64 + lane) * V, buf);
#pragma unroll
                for (int i = 0; i < V; ++i) oreg[c][i] = fmaf(p, buf[i], oreg[c][i]);
            }
        }
        for (int c = 0; c < nc && c < MAXC; ++c) {
#pragma unroll
            for (int i = 0; i < V; ++i) oreg[c][i] *= inv;
            st_vec<T>(out + (int64_t)row * H * Dh + h * Dh + (c * 64 + lane) * V, oreg[c]);
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
        const T* kp = qkv + (int64_t)(s0 + j) * ld + (int64_t)H * Dh + h * Dh;
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
        const T* vp = qkv + (int64_t)s0 * ld + 2LL * H * Dh + h * Dh + d;
        for (int j = 0; j < n; ++j) {
            ld_vec<T>(vp + (int64_t)j * ld, buf);
            const float p = ps[j];
#pragma unroll
            for (int i = 0; i < V; ++i) o[i] = fmaf(p, buf[i], o[i]);
        }
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] *= inv;
        st_vec<T>(out + (int64_t)row * H * Dh + h * Dh + d, o);
    }
}

int setok_attention_vit_bf16(hipStream_t s, const bf16* qkv, bf16* out, int n_imgs, int T, int H, int Dh, float scale);  // attn_vit.hip

extern "C" int setok_attention(void* stream, int dtype, const void* qkv, const int32_t* seg_offsets, int n_segs,
                               int seg_len, void* out, int rows, int H, int Dh, float scale) {
    SETOK_CHECK_ARG(qkv && out, "setok_attention: null operand");
    SETOK_CHECK_ARG(rows >= 0 && H > 0 && Dh > 0 && Dh % 8 == 0, "setok_attention: bad H=%d Dh=%d", H, Dh);
    SETOK_CHECK_ARG(seg_len > 0, "setok_attention: seg_len (segment length / upper bound) must be > 0");
    SETOK_CHECK_ARG(seg_offsets == nullptr || n_segs > 0, "setok_attention: n_segs must be > 0 with seg_offsets");
    if (rows == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SETOK_BF16 && !seg_offsets && Dh == 64 && rows % seg_len == 0) {
        const int rc = setok_attention_vit_bf16(s, (const bf16*)qkv, (bf16*)out, rows / seg_len, seg_len, H, Dh, scale);
        if (rc != SETOK_EUNSUPPORTED) return rc;
    }
    const size_t smem = (size_t)(Dh + seg_len) * sizeof(float);
    SETOK_CHECK_ARG(smem <= 64 * 1024, "setok_attention: Dh + seg_len too large for the generic kernel");
    dim3 grid(rows, H);
    if (dtype == SETOK_BF16)
        attn_generic_kernel<bf16><<<grid, 64, smem, s>>>((const bf16*)qkv, seg_offsets, n_segs, seg_len, (bf16*)out, rows, H, Dh, scale);
    else if (dtype == SETOK_F32)
        attn_generic_kernel<float><<<grid, 64, smem, s>>>((const float*)qkv, seg_offsets, n_segs, seg_len, (float*)out, rows, H, Dh, scale);
    else return setok_fail(SETOK_EINVAL, "setok_attention: bad dtype %d", dtype);
    SETOK_CHECK_LAUNCH("setok_attention");
    return SETOK_OK;
}
