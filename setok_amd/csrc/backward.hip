// backward.hip — the pieces of the head's backward pass that are not GEMMs (training step of the trainable
// SeTok head — inner_encoder, inter_encoder, out; the clustering is no_grad, tokenizer.py:79, and the tower is frozen).
//
// The GEMMs of the backward pass go through setok_linear:  dX = dY W  is  setok_linear(A = dY, W = W^T),  dW = dY^T X  is
// setok_linear(A = dY^T, W = X^T, fp32 out) — so the only extra ingredient they need is a transpose (setok_transpose, which also
// zero-pads the contraction dimension to the GEMM's K granule).  Everything here is deterministic: no atomics, fixed reduction
// orders, so two runs of a step give bit-identical gradients.
#include "common.h"

namespace {

// ---- out[c * ldo + r] = x[r * ldx + c]; rows r in [rows, ldo) of the output are written as zeros.  With chunk > 0 the padded row
//      range is cut into ldo / chunk chunks stored one after the other, each as a (cols, chunk) matrix: the split-K operand layout of
//      a batched dW GEMM (out[((r / chunk) * cols + c) * chunk + r % chunk]).
//      64 x 64 tiles through LDS, 16-byte global accesses on both sides (V = 8 bf16 / 4 fp32 per access).  Optionally the tile's column
//      sums over its 64 rows go to colsum[blockIdx.x][c] (fp32, fixed order): the bias gradient falls out of the pass that transposes dY.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ x, int64_t ldx, int rows, int cols, T* __restrict__ out,
                                                        int64_t ldo, int chunk, float* __restrict__ colsum) {
    constexpr int V = Elem<T>::VEC;                // elements per 16-byte access
    constexpr int CPR = 64 / V;                    // 16-byte pieces per 64-element tile row
    constexpr int RPP = 256 / CPR;                 // tile rows covered per pass of the 256 threads
    __shared__ T tile[64][64 + 2];                 // +2 elements: the transposed 2-/4-byte reads of 8 (4) consecutive rows spread over the banks
    __shared__ float csum[RPP][64];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const int pc = tid % CPR, pr = tid / CPR;
    const bool vec_in = (ldx % V == 0) && (c0 + 64 <= cols);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    for (int i = pr; i < 64; i += RPP) {
        const int r = r0 + i, c = c0 + pc * V;
        float v[V];
        if (r < rows && vec_in) ld_vec<T>(x + (int64_t)r * ldx + c, v);
        else {
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] = (r < rows && c + e < cols) ? Elem<T>::ld(x + (int64_t)r * ldx + c + e) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < V; ++e) { tile[i][pc * V + e] = (T)v[e]; acc[e] += v[e]; }
    }
    if (colsum) {
#pragma unroll
        for (int e = 0; e < V; ++e) csum[pr][pc * V + e] = acc[e];
    }
    __syncthreads();
    if (colsum && tid < 64 && c0 + tid < cols) {
        float sum = 0.f;
        for (int k = 0; k < RPP; ++k) sum += csum[k][tid];
        colsum[(int64_t)blockIdx.x * cols + c0 + tid] = sum;
    }
    // output: row c of the transposed matrix holds 64 consecutive r: CPR pieces of V elements
    const bool vec_out = (ldo % V == 0) && (chunk == 0 || chunk % V == 0);
    for (int i = pr; i < 64; i += RPP) {
        const int c = c0 + i, r = r0 + pc * V;
        if (c >= cols || r >= ldo) continue;
        float v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = (float)tile[pc * V + e][i];
        if (vec_out && r + V <= ldo) {
            T* dst = chunk > 0 ? out + ((int64_t)(r / chunk) * cols + c) * chunk + r % chunk : out + (int64_t)c * ldo + r;
            st_vec<T>(dst, v);
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int rr = r + e;
                if (rr < ldo) {
                    T* dst = chunk > 0 ? out + ((int64_t)(rr / chunk) * cols + c) * chunk + rr % chunk : out + (int64_t)c * ldo + rr;
                    Elem<T>::st(dst, v[e]);
                }
            }
        }
    }
}

// ---- column sums, two deterministic stages -----------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int rows, int cols, int rows_per_chunk,
                                                             float* __restrict__ partial) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(r0 + rows_per_chunk, rows);
    if (c >= cols) return;
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += Elem<T>::ld(x + (int64_t)r * cols + c);
    partial[(int64_t)blockIdx.y * cols + c] = s;
}

// Round 5: 64 columns per workgroup, the chunks dealt to four quarter-sums per column (chunk k -> quarter k & 3, each an ascending chain with four
// loads in flight), combined in the fixed order ((q0 + q1) + (q2 + q3)) [+ out]: deterministic.  One thread walking all (up to 256) chunks of its
// column took 24 us per call, 38 calls per training step (profiles/r05_bench_cfg4_kernel_stats.csv).
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int chunks, int cols, float* __restrict__ out,
                                                           int accumulate) {
    __shared__ float q[4][64];
    const int cl = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (c < cols) {
        int k = kq;
        for (; k + 4 < chunks; k += 8) { s0 += partial[(int64_t)k * cols + c]; s1 += partial[(int64_t)(k + 4) * cols + c]; }
        if (k < chunks) s0 += partial[(int64_t)k * cols + c];
    }
    q[kq][cl] = s0 + s1;
    __syncthreads();
    if (kq == 0 && c < cols) {
        const float s = (q[0][cl] + q[1][cl]) + (q[2][cl] + q[3][cl]);
        out[c] = accumulate ? out[c] + s : s;
    }
}

// ---- LayerNorm backward: one wave per row; the row, its statistics and the three reductions live in registers ---------------------
//   xhat = (x - mean) * rstd,  g = dy * gamma,  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ res]
//   per workgroup: partial dgamma = sum_rows dy * xhat, partial dbeta = sum_rows dy   (reduced over workgroups by colsum_final)
constexpr int LN_MAXC = 4;                       // register chunks per lane: C <= 64 * VEC * 4
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                                                            float eps, int rows, int C, T* __restrict__ dx, const T* __restrict__ res,
                                                            float* __restrict__ pg, float* __restrict__ pb) {
    constexpr int V = Elem<T>::VEC;
    extern __shared__ float red[];               // 4 waves x 2 x C
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nc = (C + 64 * V - 1) / (64 * V);
    float ag[LN_MAXC][V], ab[LN_MAXC][V];
#pragma unroll
    for (int k = 0; k < LN_MAXC; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) { ag[k][i] = 0.f; ab[k][i] = 0.f; }
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const T* xr = x + (int64_t)row * C;
        const T* dr = dy + (int64_t)row * C;
        float xv[LN_MAXC][V], dv[LN_MAXC][V];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXC; ++k) {
            const int c = (k * 64 + lane) * V;
            if (k < nc && c < C) {
                ld_vec<T>(xr + c, xv[k]); ld_vec<T>(dr + c, dv[k]);
#pragma unroll
                for (int i = 0; i < V; ++i) s += xv[k][i];
            }
        }
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXC; ++k) {
            const int c = (k * 64 + lane) * V;
            if (k < nc && c < C) {
#pragma unroll
                for (int i = 0; i < V; ++i) { const float d = xv[k][i] - mean; q += d * d; }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXC; ++k) {
            const int c = (k * 64 + lane) * V;
            if (k < nc && c < C) {
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float xh = (xv[k][i] - mean) * rstd, g = dv[k][i] * gamma[c + i];
                    xv[k][i] = xh;
                    s1 += g; s2 += g * xh;
                    ag[k][i] += dv[k][i] * xh; ab[k][i] += dv[k][i];
                    dv[k][i] = g;
                }
            }
        }
        const float c1 = wave_sum(s1) / (float)C, c2 = wave_sum(s2) / (float)C;
        if (dx) {
#pragma unroll
            for (int k = 0; k < LN_MAXC; ++k) {
                const int c = (k * 64 + lane) * V;
                if (k < nc && c < C) {
                    float o[V], rr[V];
                    if (res) ld_vec<T>(res + (int64_t)row * C + c, rr);
#pragma unroll
                    for (int i = 0; i < V; ++i) o[i] = rstd * (dv[k][i] - c1 - xv[k][i] * c2) + (res ? rr[i] : 0.f);
                    st_vec<T>(dx + (int64_t)row * C + c, o);
                }
            }
        }
    }
    // the 4 waves' partial sums -> one row of the workgroup-partial arrays
#pragma unroll
    for (int k = 0; k < LN_MAXC; ++k) {
        const int c = (k * 64 + lane) * V;
        if (k < nc && c < C) {
#pragma unroll
            for (int i = 0; i < V; ++i) { red[(wave * 2 + 0) * C + c + i] = ag[k][i]; red[(wave * 2 + 1) * C + c + i] = ab[k][i]; }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float g = 0.f, b = 0.f;
        for (int w = 0; w < 4; ++w) { g += red[(w * 2 + 0) * C + c]; b += red[(w * 2 + 1) * C + c]; }
        pg[(int64_t)blockIdx.x * C + c] = g;
        pb[(int64_t)blockIdx.x * C + c] = b;
    }
}

// ---- exact-erf GELU backward ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ pre, const T* __restrict__ dy, T* __restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = Elem<T>::ld(pre + i);
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
        Elem<T>::st(dx + i, Elem<T>::ld(dy + i) * (cdf + x * pdf));
    }
}

// dx = gelu'(pre) * drop(dy): the backward of Mlp's `drop(act(fc1 x))` in one pass.  drop(dy) is rounded to `dtype` first, as the two-launch form
// (setok_dropout on the gradient in place, then setok_gelu_bwd) rounds it: identical bits.
template <typename T>
__global__ void gelu_bwd_dropout_kernel(const T* __restrict__ pre, const T* dy, T* dx, int64_t n, float scale, unsigned thresh16,        // dx may BE dy (training.py): no __restrict__ on the pair
                                        unsigned long long seed, unsigned long long offset) {
    constexpr int V = Elem<T>::VEC;
    const int64_t nvec = n / V;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i0 = v * V;
        float pv[V], gv[V], ov[V];
        ld_vec<T>(pre + i0, pv);
        ld_vec<T>(dy + i0, gv);
        const unsigned long long c0 = offset + (unsigned long long)i0;
        unsigned long long grp = c0 >> 2, word = dropout_word(seed, grp);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const unsigned long long c = c0 + (unsigned long long)e;
            if ((c >> 2) != grp) { grp = c >> 2; word = dropout_word(seed, grp); }
            const bool keep = (unsigned)((word >> (16 * (unsigned)(c & 3))) & 0xffffu) >= thresh16;
            const float g = (float)(T)(keep ? gv[e] * scale : 0.f);
            const float x = pv[e];
            const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
            const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
            ov[e] = g * (cdf + x * pdf);
        }
        st_vec<T>(dx + i0, ov);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - nvec * V)) {
        const int64_t i = nvec * V + threadIdx.x;
        const float g = (float)(T)(dropout_keep(seed, offset + (unsigned long long)i, thresh16) ? Elem<T>::ld(dy + i) * scale : 0.f);
        const float x = Elem<T>::ld(pre + i);
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
        Elem<T>::st(dx + i, g * (cdf + x * pdf));
    }
}

// ---- varlen attention backward, generic (any head dim up to 64 * VEC * 2) ------------------------------------------------------------
// Forward: s_ij = scale q_i.k_j, p_ij = softmax_j, o_i = sum_j p_ij v_j.  With D_i = do_i.o_i:
//   ds_ij = p_ij (do_i.v_j - D_i),  dq_i = scale sum_j ds_ij k_j,  dk_j = scale sum_i ds_ij q_i,  dv_j = sum_i p_ij do_i.
// Kernel A: one wave per (query row i, head): log-sum-exp and D_i (kept for kernel B) and dq_i.
// Kernel B: one wave per (key row j, head): dk_j, dv_j by a loop over the segment's queries.  No atomics.
constexpr int AT_MAXC = 2;
template <typename T>
__device__ inline void seg_of(const int32_t* seg_offsets, int n_segs, int seg_len, int rows, int row, int& s0, int& s1) {
    if (seg_offsets) {
        int lo = 0, hi = n_segs;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_offsets[mid] <= row) lo = mid; else hi = mid; }
        s0 = seg_offsets[lo]; s1 = seg_offsets[lo + 1];
    } else { s0 = (row / seg_len) * seg_len; s1 = min(s0 + seg_len, rows); }
}

template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_q_kernel(const T* __restrict__ qkv, const int32_t* __restrict__ seg_offsets, int n_segs, int seg_len,
                                                        const T* __restrict__ o, const T* __restrict__ dout, T* __restrict__ dqkv,
                                                        float* __restrict__ lse, float* __restrict__ dsum, int rows, int H, int Dh, float scale,
                                                        int skip_long) {
    constexpr int V = Elem<T>::VEC;
    const int row = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    int s0, s1;
    seg_of<T>(seg_offsets, n_segs, seg_len, rows, row, s0, s1);
    if (row < s0 || row >= s1) return;
    if (skip_long && s1 - s0 > 32) return;                              // the segment-owning MFMA kernels' share (attn_seg_bwd.hip)
    const int64_t C = (int64_t)H * Dh, ld = 3 * C;
    const int nc = (Dh + 64 * V - 1) / (64 * V);
    float q[AT_MAXC][V], dq[AT_MAXC][V], dO[AT_MAXC][V], buf[V];
    float D = 0.f;
#pragma unroll
    for (int c = 0; c < AT_MAXC; ++c) {
        const int d = (c * 64 + lane) * V;
#pragma unroll
        for (int i = 0; i < V; ++i) { q[c][i] = 0.f; dq[c][i] = 0.f; dO[c][i] = 0.f; }
        if (c < nc && d < Dh) {
            ld_vec<T>(qkv + (int64_t)row * ld + h * Dh + d, q[c]);
            ld_vec<T>(dout + (int64_t)row * C + h * Dh + d, dO[c]);
            ld_vec<T>(o + (int64_t)row * C + h * Dh + d, buf);
#pragma unroll
            for (int i = 0; i < V; ++i) D += dO[c][i] * buf[i];
        }
    }
    D = wave_sum(D);
    auto dot_k = [&](const T* base, const float (&a)[AT_MAXC][V]) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < AT_MAXC; ++c) {
            const int d = (c * 64 + lane) * V;
            if (c < nc && d < Dh) {
                ld_vec<T>(base + d, buf);
#pragma unroll
                for (int i = 0; i < V; ++i) acc = fmaf(a[c][i], buf[i], acc);
            }
        }
        return wave_sum(acc);
    };
    float m = -INFINITY, l = 0.f;
    for (int j = s0; j < s1; ++j) {
        const float s = dot_k(qkv + (int64_t)j * ld + C + h * Dh, q) * scale;
        const float mn = fmaxf(m, s);
        l = l * expf(m - mn) + expf(s - mn);
        m = mn;
    }
    const float L = m + logf(l);
    for (int j = s0; j < s1; ++j) {
        const T* kp = qkv + (int64_t)j * ld + C + h * Dh;
        const float s = dot_k(kp, q) * scale;
        const float p = expf(s - L);
        const float dp = dot_k(qkv + (int64_t)j * ld + 2 * C + h * Dh, dO);
        const float ds = p * (dp - D) * scale;
#pragma unroll
        for (int c = 0; c < AT_MAXC; ++c) {
            const int d = (c * 64 + lane) * V;
            if (c < nc && d < Dh) {
                ld_vec<T>(kp + d, buf);
#pragma unroll
                for (int i = 0; i < V; ++i) dq[c][i] = fmaf(ds, buf[i], dq[c][i]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < AT_MAXC; ++c) {
        const int d = (c * 64 + lane) * V;
        if (c < nc && d < Dh) st_vec<T>(dqkv + (int64_t)row * ld + h * Dh + d, dq[c]);
    }
    if (lane == 0) { lse[(int64_t)row * H + h] = L; dsum[(int64_t)row * H + h] = D; }
}

template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_kv_kernel(const T* __restrict__ qkv, const int32_t* __restrict__ seg_offsets, int n_segs, int seg_len,
                                                         const T* __restrict__ dout, T* __restrict__ dqkv, const float* __restrict__ lse,
                                                         const float* __restrict__ dsum, int rows, int H, int Dh, float scale, int skip_long) {
    constexpr int V = Elem<T>::VEC;
    const int row = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    int s0, s1;
    seg_of<T>(seg_offsets, n_segs, seg_len, rows, row, s0, s1);
    if (row < s0 || row >= s1) return;
    if (skip_long && s1 - s0 > 32) return;
    const int64_t C = (int64_t)H * Dh, ld = 3 * C;
    const int nc = (Dh + 64 * V - 1) / (64 * V);
    float k[AT_MAXC][V], v[AT_MAXC][V], dk[AT_MAXC][V], dv[AT_MAXC][V], qb[AT_MAXC][V], ob[AT_MAXC][V];
#pragma unroll
    for (int c = 0; c < AT_MAXC; ++c) {
        const int d = (c * 64 + lane) * V;
#pragma unroll
        for (int i = 0; i < V; ++i) { k[c][i] = 0.f; v[c][i] = 0.f; dk[c][i] = 0.f; dv[c][i] = 0.f; }
        if (c < nc && d < Dh) {
            ld_vec<T>(qkv + (int64_t)row * ld + C + h * Dh + d, k[c]);
            ld_vec<T>(qkv + (int64_t)row * ld + 2 * C + h * Dh + d, v[c]);
        }
    }
    for (int i = s0; i < s1; ++i) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int c = 0; c < AT_MAXC; ++c) {
            const int d = (c * 64 + lane) * V;
            if (c < nc && d < Dh) {
                ld_vec<T>(qkv + (int64_t)i * ld + h * Dh + d, qb[c]);
                ld_vec<T>(dout + (int64_t)i * C + h * Dh + d, ob[c]);
#pragma unroll
                for (int e = 0; e < V; ++e) { a = fmaf(qb[c][e], k[c][e], a); b = fmaf(ob[c][e], v[c][e], b); }
            }
        }
        const float s = wave_sum(a) * scale, dp = wave_sum(b);
        const float p = expf(s - lse[(int64_t)i * H + h]);
        const float ds = p * (dp - dsum[(int64_t)i * H + h]) * scale;
#pragma unroll
        for (int c = 0; c < AT_MAXC; ++c) {
            const int d = (c * 64 + lane) * V;
            if (c < nc && d < Dh) {
#pragma unroll
                for (int e = 0; e < V; ++e) { dk[c][e] = fmaf(ds, qb[c][e], dk[c][e]); dv[c][e] = fmaf(p, ob[c][e], dv[c][e]); }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < AT_MAXC; ++c) {
        const int d = (c * 64 + lane) * V;
        if (c < nc && d < Dh) {
            st_vec<T>(dqkv + (int64_t)row * ld + C + h * Dh + d, dk[c]);
            st_vec<T>(dqkv + (int64_t)row * ld + 2 * C + h * Dh + d, dv[c]);
        }
    }
}

// ---- d(mean over a segment): every member row gets d_seg / n -------------------------------------------------------------------------
template <typename T>
__global__ void segment_mean_bwd_kernel(const T* __restrict__ dseg, const int32_t* __restrict__ off, const int32_t* __restrict__ n_segs,
                                        T* __restrict__ drows, int C) {
    constexpr int V = Elem<T>::VEC;
    const int s = blockIdx.x;
    if (s >= *n_segs) return;
    const int r0 = off[s], r1 = off[s + 1];
    const float n = (float)(r1 - r0);
    for (int c = threadIdx.x * V; c < C; c += blockDim.x * V) {
        float a[V];
        ld_vec<T>(dseg + (int64_t)s * C + c, a);
#pragma unroll
        for (int j = 0; j < V; ++j) a[j] /= n;
        for (int r = r0; r < r1; ++r) st_vec<T>(drows + (int64_t)r * C + c, a);
    }
}

// ---- AdamW on fp32 master weights, optional low-precision copy for the next forward --------------------------------------------------
template <typename T>
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, T* __restrict__ p_lp,
                             int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        float w = p[i] * (1.0f - lr * wd);                          // decoupled weight decay (torch.optim.AdamW)
        w -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
        p[i] = w;
        if (p_lp) Elem<T>::st(p_lp + i, w);
    }
}

}  // namespace

int setok_attention_bwd_seg_bf16(hipStream_t s, const bf16* qkv, const int32_t* seg_offsets, int n_segs, const bf16* out, const bf16* dout,
                                 bf16* dqkv, float* lse_ws, float* d_ws, int H, int Dh, float scale);       // attn_seg_bwd.hip

#define DISPATCH_T(NAME, CALL_BF16, CALL_F32)                                  \
    if (dtype == SETOK_BF16) { CALL_BF16; }                                    \
    else if (dtype == SETOK_F32) { CALL_F32; }                                 \
    else return setok_fail(SETOK_EINVAL, NAME ": bad dtype %d", dtype);

extern "C" int setok_transpose(void* stream, int dtype, const void* x, int64_t ldx, int rows, int cols, void* out, int64_t ldo, int chunk,
                               float* colsum_partial) {
    SETOK_CHECK_ARG(x && out, "setok_transpose: null operand");
    SETOK_CHECK_ARG(rows >= 0 && cols > 0 && ldx >= cols && ldo >= rows, "setok_transpose: bad shape rows=%d cols=%d ldx=%lld ldo=%lld", rows, cols,
                    (long long)ldx, (long long)ldo);
    SETOK_CHECK_ARG(chunk == 0 || (chunk > 0 && ldo % chunk == 0), "setok_transpose: ldo=%lld is not a multiple of chunk=%d", (long long)ldo, chunk);
    if (ldo == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(cdiv((int)ldo, 64), cdiv(cols, 64));
    DISPATCH_T("setok_transpose", (transpose_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)x, ldx, rows, cols, (bf16*)out, ldo, chunk, colsum_partial)),
               (transpose_kernel<float><<<grid, 256, 0, s>>>((const float*)x, ldx, rows, cols, (float*)out, ldo, chunk, colsum_partial)));
    SETOK_CHECK_LAUNCH("setok_transpose");
    return SETOK_OK;
}

extern "C" int setok_colsum(void* stream, int dtype, const void* x, int rows, int cols, float* out, int accumulate, float* ws, int ws_rows) {
    SETOK_CHECK_ARG(x && out && ws, "setok_colsum: null operand");
    SETOK_CHECK_ARG(rows >= 0 && cols > 0 && ws_rows >= 1, "setok_colsum: bad shape");
    hipStream_t s = (hipStream_t)stream;
    int chunks = cdiv(max(rows, 1), rows > 16384 ? 512 : 32);        // enough workgroups to fill the chip also for short inputs (per-tile partial sums)
    if (chunks > ws_rows) chunks = ws_rows;
    const int rpc = cdiv(max(rows, 1), chunks);
    chunks = cdiv(max(rows, 1), rpc);
    dim3 grid(cdiv(cols, 256), chunks);
    DISPATCH_T("setok_colsum", (colsum_partial_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)x, rows, cols, rpc, ws)),
               (colsum_partial_kernel<float><<<grid, 256, 0, s>>>((const float*)x, rows, cols, rpc, ws)));
    colsum_final_kernel<<<cdiv(cols, 64), 256, 0, s>>>(ws, chunks, cols, out, accumulate);
    SETOK_CHECK_LAUNCH("setok_colsum");
    return SETOK_OK;
}

extern "C" int setok_layernorm_bwd(void* stream, int dtype, const void* x, const void* dy, const float* gamma, float eps, int rows, int C,
                                   void* dx, const void* res, float* dgamma, float* dbeta, int accumulate, float* ws, int ws_rows) {
    SETOK_CHECK_ARG(x && dy && gamma && dgamma && dbeta && ws, "setok_layernorm_bwd: null operand");
    const int V = dtype == SETOK_BF16 ? 8 : 4;
    SETOK_CHECK_ARG(rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * V * LN_MAXC, "setok_layernorm_bwd: C=%d unsupported (max %d)", C, 64 * V * LN_MAXC);
    SETOK_CHECK_ARG(ws_rows >= 2, "setok_layernorm_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    int nb = cdiv(rows, 4);
    if (nb > ws_rows / 2) nb = ws_rows / 2;
    if (nb > 256) nb = 256;
    float* pg = ws; float* pb = ws + (int64_t)nb * C;
    const size_t smem = (size_t)8 * C * sizeof(float);
    SETOK_CHECK_ARG(smem <= 64 * 1024, "setok_layernorm_bwd: C too large");
    DISPATCH_T("setok_layernorm_bwd",
               (layernorm_bwd_kernel<bf16><<<nb, 256, smem, s>>>((const bf16*)x, (const bf16*)dy, gamma, eps, rows, C, (bf16*)dx, (const bf16*)res, pg, pb)),
               (layernorm_bwd_kernel<float><<<nb, 256, smem, s>>>((const float*)x, (const float*)dy, gamma, eps, rows, C, (float*)dx, (const float*)res, pg, pb)));
    colsum_final_kernel<<<cdiv(C, 64), 256, 0, s>>>(pg, nb, C, dgamma, accumulate);
    colsum_final_kernel<<<cdiv(C, 64), 256, 0, s>>>(pb, nb, C, dbeta, accumulate);
    SETOK_CHECK_LAUNCH("setok_layernorm_bwd");
    return SETOK_OK;
}

extern "C" int setok_gelu_bwd_dropout(void* stream, int dtype, const void* pre, const void* dy, void* dx, int64_t n, float p, uint64_t seed, uint64_t offset) {
    SETOK_CHECK_ARG(pre && dy && dx && n >= 0, "setok_gelu_bwd_dropout: bad operand");
    SETOK_CHECK_ARG(p >= 0.f && p < 1.f, "setok_gelu_bwd_dropout: p=%g outside [0, 1)", (double)p);
    SETOK_CHECK_ARG((((size_t)pre | (size_t)dy | (size_t)dx) & 15) == 0, "setok_gelu_bwd_dropout: operands must be 16-byte aligned");
    if (n == 0) return SETOK_OK;
    const int64_t nv = n / (dtype == SETOK_BF16 ? 8 : 4) + 1;
    const int grid = (int)((nv + 255) / 256 < 65536 ? (nv + 255) / 256 : 65536);
    const double t = (double)p * 65536.0;
    const unsigned thresh16 = t >= 65535.0 ? 65535u : (unsigned)(t + 0.5);
    const float scale = 1.0f / (1.0f - p);
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T("setok_gelu_bwd_dropout", (gelu_bwd_dropout_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)pre, (const bf16*)dy, (bf16*)dx, n, scale, thresh16, seed, offset)),
               (gelu_bwd_dropout_kernel<float><<<grid, 256, 0, s>>>((const float*)pre, (const float*)dy, (float*)dx, n, scale, thresh16, seed, offset)));
    SETOK_CHECK_LAUNCH("setok_gelu_bwd_dropout");
    return SETOK_OK;
}

extern "C" int setok_gelu_bwd(void* stream, int dtype, const void* pre, const void* dy, void* dx, int64_t n) {
    SETOK_CHECK_ARG(pre && dy && dx && n >= 0, "setok_gelu_bwd: bad operand");
    if (n == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    const int grid = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    DISPATCH_T("setok_gelu_bwd", (gelu_bwd_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)pre, (const bf16*)dy, (bf16*)dx, n)),
               (gelu_bwd_kernel<float><<<grid, 256, 0, s>>>((const float*)pre, (const float*)dy, (float*)dx, n)));
    SETOK_CHECK_LAUNCH("setok_gelu_bwd");
    return SETOK_OK;
}

extern "C" int setok_attention_bwd(void* stream, int dtype, const void* qkv, const int32_t* seg_offsets, int n_segs, int seg_len, const void* out,
                                   const void* dout, void* dqkv, int rows, int H, int Dh, float scale, float* ws) {
    SETOK_CHECK_ARG(qkv && out && dout && dqkv && ws, "setok_attention_bwd: null operand");
    const int V = dtype == SETOK_BF16 ? 8 : 4;
    SETOK_CHECK_ARG(rows >= 0 && H > 0 && Dh > 0 && Dh % 8 == 0 && Dh <= 64 * V * AT_MAXC, "setok_attention_bwd: bad H=%d Dh=%d", H, Dh);
    SETOK_CHECK_ARG(seg_len > 0 && (seg_offsets == nullptr || n_segs > 0), "setok_attention_bwd: bad segments");
    if (rows == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    float* lse = ws; float* dsum = ws + (int64_t)rows * H;
    int skip_long = 0;
    if (dtype == SETOK_BF16 && seg_offsets && Dh == 512 && seg_len > 32) {      // long segments: the segment-owning MFMA kernels
        const int rc = setok_attention_bwd_seg_bf16(s, (const bf16*)qkv, seg_offsets, n_segs, (const bf16*)out, (const bf16*)dout, (bf16*)dqkv, lse, dsum,
                                                    H, Dh, scale);
        if (rc == SETOK_OK) skip_long = 1; else if (rc != SETOK_EUNSUPPORTED) return rc;
    }
    dim3 grid(rows, H);
    DISPATCH_T("setok_attention_bwd",
               (attn_bwd_q_kernel<bf16><<<grid, 64, 0, s>>>((const bf16*)qkv, seg_offsets, n_segs, seg_len, (const bf16*)out, (const bf16*)dout, (bf16*)dqkv, lse, dsum, rows, H, Dh, scale, skip_long),
                attn_bwd_kv_kernel<bf16><<<grid, 64, 0, s>>>((const bf16*)qkv, seg_offsets, n_segs, seg_len, (const bf16*)dout, (bf16*)dqkv, lse, dsum, rows, H, Dh, scale, skip_long)),
               (attn_bwd_q_kernel<float><<<grid, 64, 0, s>>>((const float*)qkv, seg_offsets, n_segs, seg_len, (const float*)out, (const float*)dout, (float*)dqkv, lse, dsum, rows, H, Dh, scale, 0),
                attn_bwd_kv_kernel<float><<<grid, 64, 0, s>>>((const float*)qkv, seg_offsets, n_segs, seg_len, (const float*)dout, (float*)dqkv, lse, dsum, rows, H, Dh, scale, 0)));
    SETOK_CHECK_LAUNCH("setok_attention_bwd");
    return SETOK_OK;
}

extern "C" int setok_segment_mean_bwd(void* stream, int dtype, const void* dseg, const int32_t* seg_offsets, const int32_t* n_segs_dev, int max_segs,
                                      void* drows, int C) {
    SETOK_CHECK_ARG(dseg && seg_offsets && n_segs_dev && drows, "setok_segment_mean_bwd: null operand");
    SETOK_CHECK_ARG(max_segs >= 0 && C % 8 == 0, "setok_segment_mean_bwd: bad shape");
    if (max_segs == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    const int threads = C / 8 >= 256 ? 256 : (C / 8 >= 128 ? 128 : 64);
    DISPATCH_T("setok_segment_mean_bwd",
               (segment_mean_bwd_kernel<bf16><<<max_segs, threads, 0, s>>>((const bf16*)dseg, seg_offsets, n_segs_dev, (bf16*)drows, C)),
               (segment_mean_bwd_kernel<float><<<max_segs, threads, 0, s>>>((const float*)dseg, seg_offsets, n_segs_dev, (float*)drows, C)));
    SETOK_CHECK_LAUNCH("setok_segment_mean_bwd");
    return SETOK_OK;
}

extern "C" int setok_adamw(void* stream, int lp_dtype, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_lp, int64_t n,
                           float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale) {
    SETOK_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "setok_adamw: bad operand");
    if (n == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    const int grid = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    const int dtype = param_lp ? lp_dtype : SETOK_F32;
    DISPATCH_T("setok_adamw",
               (adamw_kernel<bf16><<<grid, 256, 0, s>>>(param, grad, exp_avg, exp_avg_sq, (bf16*)param_lp, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale)),
               (adamw_kernel<float><<<grid, 256, 0, s>>>(param, grad, exp_avg, exp_avg_sq, (float*)param_lp, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale)));
    SETOK_CHECK_LAUNCH("setok_adamw");
    return SETOK_OK;
}
