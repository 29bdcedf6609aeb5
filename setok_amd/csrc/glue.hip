// glue.hip — bandwidth-bound layout kernels around the GEMMs: patch extraction, token assembly,
// feature_select + positional add, row gather, segmented mean.  All use 16-byte accesses.
#include "common.h"

// images (B,3,H,W) -> patches (B*g*g, Kpad), column = c*p*p + py*p + px, zero padded to Kpad.
// A thread owns ONE column of the patch matrix: its (c, py, px) and the image offset they mean are computed once; the workgroup then walks
// over the patches (a wave-uniform row index: scalar arithmetic), so the per-element work is one add, one load, one store — the first
// version decomposed a 64-bit flat index with six runtime divisions per element (177 us for 256 images of 224^2 against 45).
// VW = 2: a thread owns two adjacent columns (even p: both in one image row, 4-byte aligned when W is even) — half the memory instructions.
template <typename T, int VW>
__global__ __launch_bounds__(1024) void patchify_kernel(const T* __restrict__ img, T* __restrict__ out, int B, int H, int W, int p, int Kpad) {
    constexpr int MAXC = 4 / VW;                                   // column groups per thread: Kpad <= 4096 (patches up to 36 x 36 pixels)
    const int gh = H / p, gw = W / p, K = 3 * p * p;
    int64_t off[MAXC]; bool live[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; ++j) {
        const int col = (threadIdx.x + j * blockDim.x) * VW;
        live[j] = col < K;
        const int px = col % p, py = (col / p) % p, c = col / (p * p);
        off[j] = live[j] ? ((int64_t)c * H + py) * W + px : 0;
    }
    const int npatch = B * gh * gw;
    for (int pr = blockIdx.x; pr < npatch; pr += gridDim.x) {
        const int gx = pr % gw, gy = (pr / gw) % gh, b = pr / (gw * gh);
        const T* src = img + ((int64_t)b * 3 * H + gy * p) * W + gx * p;
#pragma unroll
        for (int j = 0; j < MAXC; ++j) {
            const int col = (threadIdx.x + j * blockDim.x) * VW;
            if (col >= Kpad) continue;
            T* dst = out + (int64_t)pr * Kpad + col;
            if constexpr (VW == 2) {
                typedef T pair_t __attribute__((ext_vector_type(2)));
                pair_t v = {T(0.f), T(0.f)};
                if (live[j]) v = *reinterpret_cast<const pair_t*>(src + off[j]);
                *reinterpret_cast<pair_t*>(dst) = v;
            } else {
                Elem<T>::st(dst, live[j] ? Elem<T>::ld(src + off[j]) : 0.f);
            }
        }
    }
}

extern "C" int setok_patchify(void* stream, int dtype, const void* images, void* patches, int B, int H, int W, int p, int Kpad) {
    SETOK_CHECK_ARG(images && patches, "setok_patchify: null operand");
    SETOK_CHECK_ARG(B > 0 && p > 0 && H % p == 0 && W % p == 0 && Kpad >= 3 * p * p, "setok_patchify: bad shape");
    SETOK_CHECK_ARG(Kpad <= 4096, "setok_patchify: Kpad=%d above 4096 (a patch of more than 36 x 36 pixels)", Kpad);
    const int npatch = B * (H / p) * (W / p);
    const bool pairs = p % 2 == 0 && W % 2 == 0 && Kpad % 2 == 0;     // (px, px + 1) share an image row and are 4-byte aligned (8-byte in fp32)
    const int groups = pairs ? Kpad / 2 : Kpad;
    const int threads = groups >= 1024 ? 1024 : ((groups + 63) & ~63);
    const int grid = npatch < 256 * 16 ? npatch : 256 * 16;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SETOK_BF16) {
        if (pairs) patchify_kernel<bf16, 2><<<grid, threads, 0, s>>>((const bf16*)images, (bf16*)patches, B, H, W, p, Kpad);
        else patchify_kernel<bf16, 1><<<grid, threads, 0, s>>>((const bf16*)images, (bf16*)patches, B, H, W, p, Kpad);
    } else if (dtype == SETOK_F32) {
        if (pairs) patchify_kernel<float, 2><<<grid, threads, 0, s>>>((const float*)images, (float*)patches, B, H, W, p, Kpad);
        else patchify_kernel<float, 1><<<grid, threads, 0, s>>>((const float*)images, (float*)patches, B, H, W, p, Kpad);
    }
    else return setok_fail(SETOK_EINVAL, "setok_patchify: bad dtype %d", dtype);
    SETOK_CHECK_LAUNCH("setok_patchify");
    return SETOK_OK;
}

// tokens[b,0] = cls + pos[0]; tokens[b,1+i] = pe[b,i] + pos[1+i]
template <typename T>
__global__ void vit_assemble_kernel(const T* __restrict__ pe, const T* __restrict__ cls, const T* __restrict__ pos,
                                    T* __restrict__ tok, int B, int N, int C) {
    constexpr int V = Elem<T>::VEC;
    const int cv = C / V;
    const int64_t total = (int64_t)B * (N + 1) * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * V;
        const int64_t r = i / cv;
        const int t = (int)(r % (N + 1)), b = (int)(r / (N + 1));
        float a[V], q[V];
        if (t == 0) ld_vec<T>(cls + c, a); else ld_vec<T>(pe + ((int64_t)b * N + t - 1) * C + c, a);
        ld_vec<T>(pos + (int64_t)t * C + c, q);
#pragma unroll
        for (int j = 0; j < V; ++j) a[j] += q[j];
        st_vec<T>(tok + r * C + c, a);
    }
}

extern "C" int setok_vit_assemble(void* stream, int dtype, const void* patch_embed, const void* cls, const void* pos,
                                  void* tokens, int B, int N, int C) {
    SETOK_CHECK_ARG(patch_embed && cls && pos && tokens, "setok_vit_assemble: null operand");
    SETOK_CHECK_ARG(B > 0 && N > 0 && C % 8 == 0, "setok_vit_assemble: bad shape");
    const int64_t total = (int64_t)B * (N + 1) * (C / 4);
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SETOK_BF16) vit_assemble_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)patch_embed, (const bf16*)cls, (const bf16*)pos, (bf16*)tokens, B, N, C);
    else if (dtype == SETOK_F32) vit_assemble_kernel<float><<<grid, 256, 0, s>>>((const float*)patch_embed, (const float*)cls, (const float*)pos, (float*)tokens, B, N, C);
    else return setok_fail(SETOK_EINVAL, "setok_vit_assemble: bad dtype %d", dtype);
    SETOK_CHECK_LAUNCH("setok_vit_assemble");
    return SETOK_OK;
}

// x[b,i] = hidden[b, i+skip] + pos2d[i]
template <typename T>
__global__ void select_add_pos_kernel(const T* __restrict__ hid, const T* __restrict__ pos, T* __restrict__ x,
                                      int B, int N, int C, int skip) {
    constexpr int V = Elem<T>::VEC;
    const int cv = C / V;
    const int64_t total = (int64_t)B * N * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * V;
        const int64_t r = i / cv;
        const int t = (int)(r % N), b = (int)(r / N);
        float a[V], q[V];
        ld_vec<T>(hid + ((int64_t)b * (N + skip) + t + skip) * C + c, a);
        ld_vec<T>(pos + (int64_t)t * C + c, q);
#pragma unroll
        for (int j = 0; j < V; ++j) a[j] += q[j];
        st_vec<T>(x + r * C + c, a);
    }
}

extern "C" int setok_select_add_pos(void* stream, int dtype, const void* hidden, const void* pos2d, void* x,
                                    int B, int N, int C, int skip) {
    SETOK_CHECK_ARG(hidden && pos2d && x, "setok_select_add_pos: null operand");
    SETOK_CHECK_ARG(B > 0 && N > 0 && C % 8 == 0 && (skip == 0 || skip == 1), "setok_select_add_pos: bad shape");
    const int64_t total = (int64_t)B * N * (C / 4);
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SETOK_BF16) select_add_pos_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)hidden, (const bf16*)pos2d, (bf16*)x, B, N, C, skip);
    else if (dtype == SETOK_F32) select_add_pos_kernel<float><<<grid, 256, 0, s>>>((const float*)hidden, (const float*)pos2d, (float*)x, B, N, C, skip);
    else return setok_fail(SETOK_EINVAL, "setok_select_add_pos: bad dtype %d", dtype);
    SETOK_CHECK_LAUNCH("setok_select_add_pos");
    return SETOK_OK;
}

// out[p] = x[perm[p]]
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ x, const int32_t* __restrict__ perm, T* __restrict__ out, int rows, int C) {
    constexpr int V = Elem<T>::VEC;
    const int cv = C / V;
    const int64_t total = (int64_t)rows * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * V;
        const int64_t r = i / cv;
        float a[V];
        ld_vec<T>(x + (int64_t)perm[r] * C + c, a);
        st_vec<T>(out + r * C + c, a);
    }
}

extern "C" int setok_gather_rows(void* stream, int dtype, const void* x, const int32_t* perm, void* out, int rows, int C) {
    SETOK_CHECK_ARG(x && perm && out, "setok_gather_rows: null operand");
    SETOK_CHECK_ARG(rows >= 0 && C % 8 == 0, "setok_gather_rows: bad shape");
    if (rows == 0) return SETOK_OK;
    const int64_t total = (int64_t)rows * (C / 4);
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SETOK_BF16) gather_rows_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)x, perm, (bf16*)out, rows, C);
    else if (dtype == SETOK_F32) gather_rows_kernel<float><<<grid, 256, 0, s>>>((const float*)x, perm, (float*)out, rows, C);
    else return setok_fail(SETOK_EINVAL, "setok_gather_rows: bad dtype %d", dtype);
    SETOK_CHECK_LAUNCH("setok_gather_rows");
    return SETOK_OK;
}

// out[s] = mean_{r in [off[s], off[s+1])} h[r]      one block per segment, fp32 accumulation in row order
template <typename T>
__global__ void segment_mean_kernel(const T* __restrict__ h, const int32_t* __restrict__ off, const int32_t* __restrict__ n_segs,
                                    T* __restrict__ out, int C) {
    constexpr int V = Elem<T>::VEC;
    const int s = blockIdx.x;
    if (s >= *n_segs) return;
    const int r0 = off[s], r1 = off[s + 1];
    const float n = (float)(r1 - r0);
    for (int c = threadIdx.x * V; c < C; c += blockDim.x * V) {
        float acc[V], a[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
        for (int r = r0; r < r1; ++r) {
            ld_vec<T>(h + (int64_t)r * C + c, a);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += a[j];
        }
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] /= n;
        st_vec<T>(out + (int64_t)s * C + c, acc);
    }
}

extern "C" int setok_segment_mean(void* stream, int dtype, const void* h, const int32_t* seg_offsets,
                                  const int32_t* n_segs_dev, int max_segs, void* out, int C) {
    SETOK_CHECK_ARG(h && seg_offsets && n_segs_dev && out, "setok_segment_mean: null operand");
    SETOK_CHECK_ARG(max_segs >= 0 && C % 8 == 0, "setok_segment_mean: bad shape");
    if (max_segs == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    const int threads = C / 8 >= 256 ? 256 : (C / 8 >= 128 ? 128 : 64);
    if (dtype == SETOK_BF16) segment_mean_kernel<bf16><<<max_segs, threads, 0, s>>>((const bf16*)h, seg_offsets, n_segs_dev, (bf16*)out, C);
    else if (dtype == SETOK_F32) segment_mean_kernel<float><<<max_segs, threads, 0, s>>>((const float*)h, seg_offsets, n_segs_dev, (float*)out, C);
    else return setok_fail(SETOK_EINVAL, "setok_segment_mean: bad dtype %d", dtype);
    SETOK_CHECK_LAUNCH("setok_segment_mean");
    return SETOK_OK;
}

template <typename T>
__global__ void activation_kernel(const T* x, T* y, int64_t n, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        Elem<T>::st(y + i, act_apply(Elem<T>::ld(x + i), act));
}

extern "C" int setok_activation(void* stream, int dtype, const void* x, void* y, int64_t n, int act) {
    SETOK_CHECK_ARG(x && y && n >= 0, "setok_activation: bad operand");
    SETOK_CHECK_ARG(act >= SETOK_ACT_NONE && act <= SETOK_ACT_GELU_ERF, "setok_activation: bad act %d", act);
    if (n == 0) return SETOK_OK;
    const int grid = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SETOK_BF16) activation_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)x, (bf16*)y, n, act);
    else if (dtype == SETOK_F32) activation_kernel<float><<<grid, 256, 0, s>>>((const float*)x, (float*)y, n, act);
    else return setok_fail(SETOK_EINVAL, "setok_activation: bad dtype %d", dtype);
    SETOK_CHECK_LAUNCH("setok_activation");
    return SETOK_OK;
}

// ---- training-mode dropout of the head's Block (module.py:36,44,45,59,72: nn.Dropout(proj_drop) after the attention projection, after the Mlp's
// activation and after its fc2; proj_drop = 0.2 by default, tokenizer.py:26) -------------------------------------------------------------------
// out[i] = residual[i] + (keep_i ? x[i] / (1 - p) : 0).
// Counter-based: the mask of element i is a pure function of (seed, c = offset + i) — the backward pass regenerates it instead of storing it
// (d/dx = the same mask and scale on the incoming gradient), and a step is reproducible from its seed.  Round 4: ONE SplitMix64 finaliser per
// FOUR consecutive counters (word = hash(seed, c >> 2); keep_c = 16-bit slice (c & 3) of the word >= p * 2^16) instead of one per element — the
// 64-bit multiplies of a hash per element made the kernel compute-bound at 40 % of its memory roofline (2.2 ms of a 70 ms cfg4 step).  p is
// quantised to 2^-16 (0.2 -> 0.19999695).  Not torch's Philox stream: masks are Bernoulli(1 - p) like the reference's, not bit-equal to them.
template <typename T>
__global__ void dropout_kernel(const T* x, const T* res, T* y, int64_t n, float scale, unsigned thresh16, unsigned long long seed, unsigned long long offset) {
    constexpr int V = Elem<T>::VEC;                                      // 16 bytes per thread and step
    const int64_t nvec = n / V;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i0 = v * V;
        float xv[V], rv[V], ov[V];
        ld_vec<T>(x + i0, xv);
        if (res) ld_vec<T>(res + i0, rv);
        const unsigned long long c0 = offset + (unsigned long long)i0;
        unsigned long long grp = c0 >> 2, word = dropout_word(seed, grp);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const unsigned long long c = c0 + (unsigned long long)e;
            if ((c >> 2) != grp) { grp = c >> 2; word = dropout_word(seed, grp); }
            const bool keep = (unsigned)((word >> (16 * (unsigned)(c & 3))) & 0xffffu) >= thresh16;
            const float d = keep ? xv[e] * scale : 0.f;
            ov[e] = res ? rv[e] + d : d;
        }
        st_vec<T>(y + i0, ov);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - nvec * V)) {          // the last n mod V elements
        const int64_t i = nvec * V + threadIdx.x;
        const float d = dropout_keep(seed, offset + (unsigned long long)i, thresh16) ? Elem<T>::ld(x + i) * scale : 0.f;
        Elem<T>::st(y + i, res ? Elem<T>::ld(res + i) + d : d);
    }
}

// y = drop(act(x)) in one pass — Mlp.forward's `drop(act(fc1(x)))` (module.py:41,44) in training mode; the intermediate act(x) is rounded to `dtype`
// first, exactly as the two-launch form (setok_activation, then setok_dropout in place) rounds it: identical bits, one pass over the hidden less.
template <typename T>
__global__ void activation_dropout_kernel(const T* x, T* y, int64_t n, int act, float scale, unsigned thresh16, unsigned long long seed, unsigned long long offset) {
    constexpr int V = Elem<T>::VEC;
    const int64_t nvec = n / V;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i0 = v * V;
        float xv[V], ov[V];
        ld_vec<T>(x + i0, xv);
        const unsigned long long c0 = offset + (unsigned long long)i0;
        unsigned long long grp = c0 >> 2, word = dropout_word(seed, grp);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const unsigned long long c = c0 + (unsigned long long)e;
            if ((c >> 2) != grp) { grp = c >> 2; word = dropout_word(seed, grp); }
            const bool keep = (unsigned)((word >> (16 * (unsigned)(c & 3))) & 0xffffu) >= thresh16;
            const float a = (float)(T)act_apply(xv[e], act);                   // rounded like the stored activation of the two-launch form
            ov[e] = keep ? a * scale : 0.f;
        }
        st_vec<T>(y + i0, ov);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - nvec * V)) {
        const int64_t i = nvec * V + threadIdx.x;
        const float a = (float)(T)act_apply(Elem<T>::ld(x + i), act);
        Elem<T>::st(y + i, dropout_keep(seed, offset + (unsigned long long)i, thresh16) ? a * scale : 0.f);
    }
}

static inline unsigned dropout_thresh16(float p) {
    const double t = (double)p * 65536.0;
    return t >= 65535.0 ? 65535u : (unsigned)(t + 0.5);
}

extern "C" int setok_activation_dropout(void* stream, int dtype, const void* x, void* y, int64_t n, int act, float p, uint64_t seed, uint64_t offset) {
    SETOK_CHECK_ARG(x && y && n >= 0, "setok_activation_dropout: bad operand");
    SETOK_CHECK_ARG(act >= SETOK_ACT_NONE && act <= SETOK_ACT_GELU_ERF, "setok_activation_dropout: bad act %d", act);
    SETOK_CHECK_ARG(p >= 0.f && p < 1.f, "setok_activation_dropout: p=%g outside [0, 1)", (double)p);
    SETOK_CHECK_ARG((((size_t)x | (size_t)y) & 15) == 0, "setok_activation_dropout: operands must be 16-byte aligned");
    if (n == 0) return SETOK_OK;
    const int64_t nv = n / (dtype == SETOK_BF16 ? 8 : 4) + 1;
    const int grid = (int)((nv + 255) / 256 < 65536 ? (nv + 255) / 256 : 65536);
    const float scale = 1.0f / (1.0f - p);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SETOK_BF16) activation_dropout_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)x, (bf16*)y, n, act, scale, dropout_thresh16(p), seed, offset);
    else if (dtype == SETOK_F32) activation_dropout_kernel<float><<<grid, 256, 0, s>>>((const float*)x, (float*)y, n, act, scale, dropout_thresh16(p), seed, offset);
    else return setok_fail(SETOK_EINVAL, "setok_activation_dropout: bad dtype %d", dtype);
    SETOK_CHECK_LAUNCH("setok_activation_dropout");
    return SETOK_OK;
}

template <typename T>
__global__ void dropout_scalar_kernel(const T* x, const T* res, T* y, int64_t n, float scale, unsigned thresh16, unsigned long long seed, unsigned long long offset) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {      // operands not 16-byte aligned (a view)
        const float d = dropout_keep(seed, offset + (unsigned long long)i, thresh16) ? Elem<T>::ld(x + i) * scale : 0.f;
        Elem<T>::st(y + i, res ? Elem<T>::ld(res + i) + d : d);
    }
}

extern "C" int setok_dropout(void* stream, int dtype, const void* x, const void* residual, void* y, int64_t n, float p, uint64_t seed, uint64_t offset) {
    SETOK_CHECK_ARG(x && y && n >= 0, "setok_dropout: bad operand");
    SETOK_CHECK_ARG(p >= 0.f && p < 1.f, "setok_dropout: p=%g outside [0, 1)", (double)p);
    if (n == 0) return SETOK_OK;
    if ((((size_t)x | (size_t)y | (size_t)residual) & 15) != 0) {         // same mask, element by element
        const int g1 = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
        hipStream_t s1 = (hipStream_t)stream;
        const float sc = 1.0f / (1.0f - p);
        if (dtype == SETOK_BF16) dropout_scalar_kernel<bf16><<<g1, 256, 0, s1>>>((const bf16*)x, (const bf16*)residual, (bf16*)y, n, sc, dropout_thresh16(p), seed, offset);
        else if (dtype == SETOK_F32) dropout_scalar_kernel<float><<<g1, 256, 0, s1>>>((const float*)x, (const float*)residual, (float*)y, n, sc, dropout_thresh16(p), seed, offset);
        else return setok_fail(SETOK_EINVAL, "setok_dropout: bad dtype %d", dtype);
        SETOK_CHECK_LAUNCH("setok_dropout");
        return SETOK_OK;
    }
    const int64_t nv = n / (dtype == SETOK_BF16 ? 8 : 4) + 1;
    const int grid = (int)((nv + 255) / 256 < 65536 ? (nv + 255) / 256 : 65536);
    const unsigned thresh16 = dropout_thresh16(p);
    const float scale = 1.0f / (1.0f - p);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SETOK_BF16) dropout_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)x, (const bf16*)residual, (bf16*)y, n, scale, thresh16, seed, offset);
    else if (dtype == SETOK_F32) dropout_kernel<float><<<grid, 256, 0, s>>>((const float*)x, (const float*)residual, (float*)y, n, scale, thresh16, seed, offset);
    else return setok_fail(SETOK_EINVAL, "setok_dropout: bad dtype %d", dtype);
    SETOK_CHECK_LAUNCH("setok_dropout");
    return SETOK_OK;
}

// ---- reconstruction decoder, the output the reference never defines -------------------------------------------------
// SetokDeTokenizer.forward ends at decoder_norm and returns None (detokenizer.py:101-120) although SeTok.forward hands its result to a
// pixel-space loss as an IMAGE (`self.rec_loss(target, prediction, ...)`, model.py:75-76,91).  The head defined here is the standard one of
// a ViT pixel decoder: a Linear decoder_embed_dim -> patch^2 * 3 per query (a setok_linear call), this rearrangement, and the reference's own
// pixel terms as one scalar.
// unpatchify: patches (B * gh * gw, ld), row = one query, columns (pi, qi, c) with c fastest -> image (B, 3, gh * p, gw * p):
//   image[b, c, h * p + pi, w * p + qi] = patches[(b * gh + h) * gw + w, (pi * p + qi) * 3 + c]        ('n h w p q c -> n c (h p) (w q)')
template <typename T>
__global__ __launch_bounds__(256) void unpatchify_kernel(const T* __restrict__ patches, int64_t ld, T* __restrict__ img, int B, int gh, int gw, int p) {
    const int W = gw * p, Hh = gh * p;
    const int64_t n = (int64_t)B * 3 * Hh * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % Hh), c = (int)((i / ((int64_t)W * Hh)) % 3), b = (int)(i / ((int64_t)3 * W * Hh));
        const int h = y / p, pi = y - h * p, w = x / p, qi = x - w * p;
        img[i] = patches[((int64_t)(b * gh + h) * gw + w) * ld + (pi * p + qi) * 3 + c];
    }
}

extern "C" int setok_unpatchify(void* stream, int dtype, const void* patches, int64_t ld, void* image, int B, int gh, int gw, int p) {
    SETOK_CHECK_ARG(patches && image, "setok_unpatchify: null operand");
    SETOK_CHECK_ARG(B > 0 && gh > 0 && gw > 0 && p > 0 && ld >= (int64_t)p * p * 3, "setok_unpatchify: bad shape B=%d grid=%dx%d p=%d ld=%lld", B, gh, gw, p, (long long)ld);
    const int64_t n = (int64_t)B * 3 * gh * p * gw * p;
    const int grid = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SETOK_BF16) unpatchify_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)patches, ld, (bf16*)image, B, gh, gw, p);
    else if (dtype == SETOK_F32) unpatchify_kernel<float><<<grid, 256, 0, s>>>((const float*)patches, ld, (float*)image, B, gh, gw, p);
    else return setok_fail(SETOK_EINVAL, "setok_unpatchify: bad dtype %d", dtype);
    SETOK_CHECK_LAUNCH("setok_unpatchify");
    return SETOK_OK;
}

// pixel loss: out[0] = mean_i f(pred_i - target_i), f = square (kind 0: WeightedMSELoss without a mask, loss/mse.py:9-19 — the mean over
// (C, H, W) and then over the batch of equally sized images is the mean over all elements) or abs (kind 1: the pixel term of the GAN loss,
// loss/discriminator.py:161,170).  fp32 accumulation, two fixed-order stages (no atomics): bit-identical from run to run.
constexpr int PL_BLOCKS = 1024;
template <typename T>
__global__ __launch_bounds__(256) void pixel_loss_partial_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t n, int kind, float* __restrict__ part) {
    __shared__ float sw[4];
    const int64_t per = (n + PL_BLOCKS - 1) / PL_BLOCKS, lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    float acc = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const float d = Elem<T>::ld(a + i) - Elem<T>::ld(b + i);
        acc += kind == 0 ? d * d : fabsf(d);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
__global__ __launch_bounds__(256) void pixel_loss_final_kernel(const float* __restrict__ part, int64_t n, float* __restrict__ out) {
    __shared__ float sw[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < PL_BLOCKS; i += 256) acc += part[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((sw[0] + sw[1]) + (sw[2] + sw[3])) / (float)n;
}

extern "C" int setok_pixel_loss(void* stream, int dtype, const void* pred, const void* target, int64_t n, int kind, float* ws, float* out) {
    SETOK_CHECK_ARG(pred && target && ws && out, "setok_pixel_loss: null operand");
    SETOK_CHECK_ARG(n > 0 && (kind == 0 || kind == 1), "setok_pixel_loss: bad n=%lld kind=%d", (long long)n, kind);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SETOK_BF16) pixel_loss_partial_kernel<bf16><<<PL_BLOCKS, 256, 0, s>>>((const bf16*)pred, (const bf16*)target, n, kind, ws);
    else if (dtype == SETOK_F32) pixel_loss_partial_kernel<float><<<PL_BLOCKS, 256, 0, s>>>((const float*)pred, (const float*)target, n, kind, ws);
    else return setok_fail(SETOK_EINVAL, "setok_pixel_loss: bad dtype %d", dtype);
    pixel_loss_final_kernel<<<1, 256, 0, s>>>(ws, n, out);
    SETOK_CHECK_LAUNCH("setok_pixel_loss");
    return SETOK_OK;
}
