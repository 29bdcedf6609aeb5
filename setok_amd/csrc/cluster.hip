// cluster.hip — DPC-kNN dynamic clustering (tokenizer.py:78-121), batched over images, no host sync.
//
// Per image:  Gram (MFMA, gemm.hip) -> scaled distance rows + row max + kNN density (one wave per
// row, the row lives in registers) -> delta/score (one wave per row) -> centre selection + ordered
// compaction (one workgroup per image) -> nearest-centre assignment (one thread per token).
// The N x N fp32 distance matrix of an image (256 KiB at N=256, 1.3 MiB at N=576) is written once
// and re-read twice; at these sizes it never leaves L2 / Infinity Cache.
//
// Integer outputs must be bit-exact w.r.t. the reference wherever its own fp32 decisions are not
// rounding-fragile, so every quantity follows the reference's formula: d = sqrt(max(|a|^2 + |b|^2
// - 2 a.b, 0)) / sqrt(C)  (torch.cdist's matmul form, :82), density = exp(-mean_k(d^2)) (:90),
// mask[i,j] = rho_j > rho_i and the row-j-max quirk (:96-99), score = delta * rho (:101), first-min
// argmin over centre rows (:111-113), centres own themselves (:117-119).
#include "common.h"

constexpr int MAXPL = 16;          // values per lane -> N <= 1024

extern "C" int setok_linear(void*, int, int, const void*, int64_t, const void*, const float*, const void*, void*,
                            int64_t, int, int, int, int, int, int64_t, int64_t, int64_t);

// --------------------------------------------------------------------------------------------
// bf16 Gram matrices G_b = X_b X_b^T (fp32 out) + their diagonals, one launch for the whole batch.
// Workgroup (4 waves) = (image, 64-row strip, 256-column chunk); K runs over the channels in tiles of 64 through two
// 40 KiB LDS stages fed by LDS-DMA (lane-linear image, so the 16-B-slot swizzle sits on the source address — same scheme as
// gemm_persist.hip); a wave owns 64 columns of the chunk for all 64 rows (2 x 2 MFMA tiles, 64 accumulator registers).
// The generic batched GEMM ran this shape (N x N x C per image, fp32 out) at 215 TFLOP/s: 4 tiles per image, two barriers per
// k-step, operands staged through VGPRs.  The diagonal block's wave also writes |x_i|^2 = G_ii into vec[b][3][i].
// --------------------------------------------------------------------------------------------
namespace {
constexpr int GR_ROWS = 64, GR_COLS = 256, GR_K = 64;
constexpr int GR_STAGE = (GR_ROWS + GR_COLS) * GR_K * 2;      // 40 KiB
constexpr int GR_BOFF = GR_ROWS * GR_K * 2;                   // 8 KiB

__device__ inline int gr_swz(int row) { return ((row >> 1) & 1) | (((row >> 4) & 1) << 1) | (((row >> 3) & 1) << 2); }

__global__ __launch_bounds__(256, 2) void gram_bf16_kernel(const bf16* __restrict__ X, float* __restrict__ G, float* __restrict__ vec,
                                                           int B, int N, int C, int strips, int chunks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, hi = lane >> 5;
    // blockIdx.x -> (image, strip, chunk): the workgroups of one image sit on ONE XCD (blockIdx % 8) so that the image's rows are
    // fetched from HBM once and shared through that XCD's L2.
    const int per_img = strips * chunks;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int b = (q / per_img) * 8 + xcd, rem = q % per_img;
    const int strip = rem / chunks, chunk = rem % chunks;
    if (b >= B) return;                                           // the grid is padded to a multiple of 8 images
    const int r0 = strip * GR_ROWS, c0 = chunk * GR_COLS;
    const bf16* Xb = X + (int64_t)b * N * C;

    const bf16* a_src[2]; const bf16* b_src[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int p = i * 256 + tid, row = p >> 3, kc = (p & 7) ^ gr_swz(row);
        if (i < 2) a_src[i] = Xb + (int64_t)min(r0 + row, N - 1) * C + kc * 8;
        b_src[i] = Xb + (int64_t)min(c0 + row, N - 1) * C + kc * 8;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem) + wave * 1024;
    auto dma16 = [&](const bf16* ptr, unsigned lds_dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(ptr), "s"(lds_dst) : "memory");
    };
    auto issue = [&](int stage, int k0) {
        const unsigned sb = lds0 + stage * GR_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(a_src[i] + k0, sb + i * 4096);
#pragma unroll
        for (int i = 0; i < 8; ++i) dma16(b_src[i] + k0, sb + GR_BOFF + i * 4096);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = C / GR_K;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of K-tile kt
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // everyone's; everyone is done with the other stage
        if (kt + 1 < nk) issue((kt + 1) & 1, (kt + 1) * GR_K);
        const char* Ab = smem + (kt & 1) * GR_STAGE;
        const char* Bb = Ab + GR_BOFF;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ra = t * 32 + frow, rb = wave * 64 + t * 32 + frow;
                af[t] = *reinterpret_cast<const bf16x8*>(Ab + ra * 128 + (((ks * 2 + hi) ^ gr_swz(ra)) << 4));
                bfr[t] = *reinterpret_cast<const bf16x8*>(Bb + rb * 128 + (((ks * 2 + hi) ^ gr_swz(rb)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }
    // D layout of v_mfma_f32_32x32x16: lane holds column (lane & 31), rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5): for a fixed
    // register 32 lanes cover 32 consecutive columns of one row -> 128-byte segments.
    float* Gb = G + (int64_t)b * N * N;
    float* norms = vec + ((int64_t)b * 4 + 3) * N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = c0 + wave * 64 + j * 32 + frow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (row < N && col < N) {
                    Gb[(int64_t)row * N + col] = acc[i][j][r];
                    if (row == col) norms[row] = acc[i][j][r];
                }
            }
        }
}
}  // namespace

// |x_i|^2 = G_ii (the same fma chain as every other Gram entry) -> vec[b][3][i]
__global__ void diag_kernel(const float* __restrict__ G, float* __restrict__ vec, int B, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, r = i % N;
    vec[((int64_t)b * 4 + 3) * N + r] = G[((int64_t)b * N + r) * N + r];
}

// k-th smallest (1-indexed) of the wave's values (all >= 0, so the uint bit pattern orders them) and the sum of squares of
// the k smallest: MSB-first radix select.  The count of one radix step is a wave ballot + scalar popcount per register slot,
// so the whole search runs on the scalar unit (the threshold, the counts and the decision are wave-uniform).
template <int PL>
__device__ inline float knn_sqsum(const float (&v)[MAXPL], int lane, int N, int k) {
    uint32_t T = 0;
    for (int bit = 30; bit >= 0; --bit) {
        const uint32_t cand = T | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int t = 0; t < PL; ++t)
            cnt += __builtin_popcountll(__ballot(lane + 64 * t < N && __float_as_uint(v[t]) < cand));
        if (cnt <= k - 1) T = cand;
    }
    const float kth = __uint_as_float(T);
    float s = 0.f; int cnt = 0;
#pragma unroll
    for (int t = 0; t < PL; ++t) {
        const bool in = lane + 64 * t < N && __float_as_uint(v[t]) < T;
        if (in) s += v[t] * v[t];
        cnt += __builtin_popcountll(__ballot(in));
    }
    s = wave_sum(s);
    return s + (float)(k - cnt) * (kth * kth);
}

// One wave per row i of image b.  G (in/out): Gram row -> scaled distance row.
// vec layout per image: [0] density, [1] row max, [2] delta, [3] norms (later: centre list as int32)
template <bool WITH_DENSITY, int PL>
__global__ __launch_bounds__(256) void dist_rows_kernel(float* __restrict__ G, float* __restrict__ vec, const float* __restrict__ noise,
                                                        int B, int N, int k, float sqrtC) {
    const int lane = threadIdx.x & 63;
    const int gr = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gr >= B * N) return;
    const int b = gr / N, i = gr % N;
    float* row = G + (int64_t)gr * N;
    const float* norms = vec + ((int64_t)b * 4 + 3) * N;
    const float ni = norms[i];
    float v[MAXPL];
    float mx = 0.f;
#pragma unroll
    for (int t = 0; t < PL; ++t) {
        const int j = lane + 64 * t;
        v[t] = 0.f;
        if (j < N) {
            const float d2 = fmaxf((ni + norms[j]) - 2.0f * row[j], 0.f);
            v[t] = sqrtf(d2) / sqrtC;
            row[j] = v[t];
            mx = fmaxf(mx, v[t]);
        }
    }
    mx = wave_max(mx);
    if (lane == 0) vec[((int64_t)b * 4 + 1) * N + i] = mx;
    if (WITH_DENSITY) {
        const float mean = knn_sqsum<PL>(v, lane, N, k) / (float)k;
        float rho = expf(-mean);
        if (noise) rho += noise[gr] * 1e-6f;
        if (lane == 0) vec[((int64_t)b * 4 + 0) * N + i] = rho;
    }
}

// token_mask path (:84-86, :93-94): gmax[b] = max of the raw distance matrix
__global__ void gmax_kernel(const float* __restrict__ vec, float* __restrict__ gmax, int N) {
    const int b = blockIdx.x;
    float m = 0.f;
    for (int j = threadIdx.x; j < N; j += 64) m = fmaxf(m, vec[((int64_t)b * 4 + 1) * N + j]);
    m = wave_max(m);
    if (threadIdx.x == 0) gmax[b] = m;
}

template <int PL>
__global__ __launch_bounds__(256) void masked_density_kernel(float* __restrict__ D, float* __restrict__ vec, const float* __restrict__ noise,
                                                             const float* __restrict__ tmask, const float* __restrict__ gmax,
                                                             int B, int N, int k) {
    const int lane = threadIdx.x & 63;
    const int gr = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gr >= B * N) return;
    const int b = gr / N, i = gr % N;
    float* row = D + (int64_t)gr * N;
    const float fill = gmax[b] + 1.0f;
    float v[MAXPL];
    float mx = 0.f;
#pragma unroll
    for (int t = 0; t < PL; ++t) {
        const int j = lane + 64 * t;
        v[t] = 0.f;
        if (j < N) {
            v[t] = (tmask[(int64_t)b * N + j] > 0.f) ? row[j] : fill;
            row[j] = v[t];
            mx = fmaxf(mx, v[t]);
        }
    }
    mx = wave_max(mx);
    const float mean = knn_sqsum<PL>(v, lane, N, k) / (float)k;
    float rho = expf(-mean);
    if (noise) rho += noise[gr] * 1e-6f;
    if (!(tmask[gr] > 0.f)) rho = 0.f;                                  // density * token_mask (:94)
    if (lane == 0) {
        vec[((int64_t)b * 4 + 1) * N + i] = mx;
        vec[((int64_t)b * 4 + 0) * N + i] = rho;
    }
}

// delta_i = min_j (rho_j > rho_i ? D_ij : rowmax_j);  score_i = delta_i * rho_i
__global__ __launch_bounds__(256) void score_kernel(const float* __restrict__ D, float* __restrict__ vec, float* __restrict__ score,
                                                    int B, int N) {
    const int lane = threadIdx.x & 63;
    const int gr = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gr >= B * N) return;
    const int b = gr / N, i = gr % N;
    const float* row = D + (int64_t)gr * N;
    const float* rho = vec + ((int64_t)b * 4 + 0) * N;
    const float* rmax = vec + ((int64_t)b * 4 + 1) * N;
    const float ri = rho[i];
    float m = INFINITY;
    for (int j = lane; j < N; j += 64) m = fminf(m, (rho[j] > ri) ? row[j] : rmax[j]);
    m = wave_min(m);
    if (lane == 0) {
        vec[((int64_t)b * 4 + 2) * N + i] = m;
        score[gr] = m * ri;
    }
}

// One workgroup per image: centres = {i : score_i > thr} in index order; if none, the
// min_cluster_num largest scores (ties: lower index first) in index order.
__global__ __launch_bounds__(256) void select_kernel(const float* __restrict__ score, float thr, int mcn, int N,
                                                     int64_t* __restrict__ index_down, int32_t* __restrict__ counts,
                                                     float* __restrict__ vec) {
    __shared__ float s[1024];
    __shared__ unsigned char flag[1024];
    __shared__ int part[256];
    __shared__ int total;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) total = 0;
    for (int i = tid; i < N; i += 256) s[i] = score[(int64_t)b * N + i];
    __syncthreads();
    int local = 0;
    for (int i = tid; i < N; i += 256) { const int f = s[i] > thr; flag[i] = f; local += f; }
    if (local) atomicAdd(&total, local);
    __syncthreads();
    if (total == 0) {
        for (int i = tid; i < N; i += 256) {
            const float si = s[i];
            int rank = 0;
            for (int j = 0; j < N; ++j) rank += (s[j] > si) || (s[j] == si && j < i);
            flag[i] = rank < mcn;
        }
    }
    __syncthreads();
    // ordered compaction: thread t owns the contiguous chunk [t*per, (t+1)*per)
    const int per = (N + 255) / 256;
    const int lo = tid * per, hi = min(lo + per, N);
    int c = 0;
    for (int i = lo; i < hi; ++i) c += flag[i];
    part[tid] = c;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < 256; ++t) { const int x = part[t]; part[t] = run; run += x; }
        total = run;
        counts[b] = run;
    }
    __syncthreads();
    int pos = part[tid];
    int32_t* centres = reinterpret_cast<int32_t*>(vec + ((int64_t)b * 4 + 3) * N);
    for (int i = lo; i < hi; ++i)
        if (flag[i]) { index_down[(int64_t)b * N + pos] = i; centres[pos] = i; ++pos; }
    for (int i = total + tid; i < N; i += 256) index_down[(int64_t)b * N + i] = -1;
}

// label_j = first argmin_c D[centre_c][j]; centres relabelled to their own position.
__global__ __launch_bounds__(256) void assign_kernel(const float* __restrict__ D, const float* __restrict__ vec,
                                                     const int32_t* __restrict__ counts, int64_t* __restrict__ idx, int N) {
    __shared__ int cs[1024];
    const int b = blockIdx.y;
    const int L = counts[b];
    const int32_t* centres = reinterpret_cast<const int32_t*>(vec + ((int64_t)b * 4 + 3) * N);
    for (int c = threadIdx.x; c < L; c += 256) cs[c] = centres[c];
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const float* Db = D + (int64_t)b * N * N;
    float best = INFINITY; int lab = 0;
    for (int c = 0; c < L; ++c) {
        const float d = Db[(int64_t)cs[c] * N + j];
        if (d < best) { best = d; lab = c; }
    }
    int lo = 0, hi = L - 1;                                           // is j itself a centre?
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (cs[mid] == j) { lab = mid; break; }
        if (cs[mid] < j) lo = mid + 1; else hi = mid - 1;
    }
    idx[(int64_t)b * N + j] = lab;
}

extern "C" int setok_cluster_dpc_knn(void* stream, int dtype, const void* x, int B, int N, int C, int k,
                                     float threshold, int min_cluster_num, const float* noise,
                                     const float* token_mask, int64_t* idx_cluster, float* score,
                                     int64_t* index_down, int32_t* counts, float* dist_ws, float* vec_ws) {
    SETOK_CHECK_ARG(x && idx_cluster && score && index_down && counts && dist_ws && vec_ws, "setok_cluster_dpc_knn: null operand");
    SETOK_CHECK_ARG(B > 0 && N > 0 && N <= 64 * MAXPL && C > 0, "setok_cluster_dpc_knn: need 0 < N <= %d (got N=%d)", 64 * MAXPL, N);
    SETOK_CHECK_ARG(k >= 1 && k <= N, "setok_cluster_dpc_knn: k=%d out of range (torch.topk would raise), N=%d", k, N);
    SETOK_CHECK_ARG(min_cluster_num >= 1 && min_cluster_num <= N, "setok_cluster_dpc_knn: min_cluster_num=%d out of range, N=%d", min_cluster_num, N);
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * N;
    const float sqrtC = (float)sqrt((double)C);
    if (dtype == SETOK_BF16 && C % GR_K == 0) {
        // Gram matrices + diagonals in one launch (bf16 throughput mode)
        static SetokDeviceOnce once;
        if (!once.run([] { return hipFuncSetAttribute((const void*)gram_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GR_STAGE) == hipSuccess; }))
            return setok_fail(SETOK_ELAUNCH, "setok_cluster_dpc_knn: cannot raise the dynamic LDS limit");
        const int strips = cdiv(N, GR_ROWS), chunks = cdiv(N, GR_COLS);
        gram_bf16_kernel<<<cdiv(B, 8) * 8 * strips * chunks, 256, 2 * GR_STAGE, s>>>((const bf16*)x, dist_ws, vec_ws, B, N, C, strips, chunks);
    } else {
        // fp32 parity mode: batched exact-f32 MFMA GEMM with A = W = X_b, then |x_i|^2 = G_ii
        int rc = setok_linear(stream, dtype, SETOK_F32, x, C, x, nullptr, nullptr, dist_ws, N, N, N, C, SETOK_ACT_NONE, B,
                              (int64_t)N * C, (int64_t)N * C, (int64_t)N * N);
        if (rc != SETOK_OK) return rc;
        diag_kernel<<<cdiv(rows, 256), 256, 0, s>>>(dist_ws, vec_ws, B, N);
    }
    const int pl = N <= 256 ? 4 : (N <= 576 ? 9 : MAXPL);             // register slots per lane holding one distance row
    const dim3 rg(cdiv(rows, 4));
    if (!token_mask) {
        if (pl == 4) dist_rows_kernel<true, 4><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, B, N, k, sqrtC);
        else if (pl == 9) dist_rows_kernel<true, 9><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, B, N, k, sqrtC);
        else dist_rows_kernel<true, MAXPL><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, B, N, k, sqrtC);
    } else {
        if (pl == 4) dist_rows_kernel<false, 4><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, B, N, k, sqrtC);
        else if (pl == 9) dist_rows_kernel<false, 9><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, B, N, k, sqrtC);
        else dist_rows_kernel<false, MAXPL><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, B, N, k, sqrtC);
        float* gmax = reinterpret_cast<float*>(counts);                 // B floats of scratch until select_kernel overwrites it
        gmax_kernel<<<B, 64, 0, s>>>(vec_ws, gmax, N);
        if (pl == 4) masked_density_kernel<4><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, token_mask, gmax, B, N, k);
        else if (pl == 9) masked_density_kernel<9><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, token_mask, gmax, B, N, k);
        else masked_density_kernel<MAXPL><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, token_mask, gmax, B, N, k);
    }
    score_kernel<<<cdiv(rows, 4), 256, 0, s>>>(dist_ws, vec_ws, score, B, N);
    select_kernel<<<B, 256, 0, s>>>(score, threshold, min_cluster_num, N, index_down, counts, vec_ws);
    assign_kernel<<<dim3(cdiv(N, 256), B), 256, 0, s>>>(dist_ws, vec_ws, counts, idx_cluster, N);
    SETOK_CHECK_LAUNCH("setok_cluster_dpc_knn");
    return SETOK_OK;
}

// --------------------------------------------------------------------------------------------
// stable counting sort by cluster id + segment tables
// --------------------------------------------------------------------------------------------
__global__ void prefix_counts_kernel(const int32_t* __restrict__ counts, int32_t* __restrict__ img_offsets, int B) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int run = 0;
        for (int b = 0; b < B; ++b) { img_offsets[b] = run; run += counts[b]; }
        img_offsets[B] = run;
    }
}

__global__ __launch_bounds__(256) void sort_kernel(const int64_t* __restrict__ idx, const int32_t* __restrict__ counts,
                                                   const int32_t* __restrict__ img_offsets, int B, int N,
                                                   int32_t* __restrict__ perm, int32_t* __restrict__ seg_offsets) {
    __shared__ int lab[1024];
    __shared__ int start[1025];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int L = counts[b];
    for (int i = tid; i < N; i += 256) lab[i] = (int)idx[(int64_t)b * N + i];
    for (int c = tid; c <= L; c += 256) start[c] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += 256) atomicAdd(&start[lab[i] + 1], 1);   // start[c+1] = size of cluster c
    __syncthreads();
    if (tid == 0) for (int c = 0; c < L; ++c) start[c + 1] += start[c];    // exclusive starts
    __syncthreads();
    for (int i = tid; i < N; i += 256) {
        const int l = lab[i];
        int rank = 0;
        for (int j = 0; j < i; ++j) rank += (lab[j] == l);
        perm[(int64_t)b * N + start[l] + rank] = b * N + i;
    }
    const int base = img_offsets[b];
    for (int c = tid; c < L; c += 256) seg_offsets[base + c] = b * N + start[c];
    if (b == B - 1 && tid == 0) seg_offsets[base + L] = B * N;
}

extern "C" int setok_cluster_sort(void* stream, const int64_t* idx_cluster, const int32_t* counts, int B, int N,
                                  int32_t* perm, int32_t* seg_offsets, int32_t* img_offsets) {
    SETOK_CHECK_ARG(idx_cluster && counts && perm && seg_offsets && img_offsets, "setok_cluster_sort: null operand");
    SETOK_CHECK_ARG(B > 0 && N > 0 && N <= 1024, "setok_cluster_sort: need 0 < N <= 1024");
    hipStream_t s = (hipStream_t)stream;
    prefix_counts_kernel<<<1, 64, 0, s>>>(counts, img_offsets, B);
    sort_kernel<<<B, 256, 0, s>>>(idx_cluster, counts, img_offsets, B, N, perm, seg_offsets);
    SETOK_CHECK_LAUNCH("setok_cluster_sort");
    return SETOK_OK;
}
