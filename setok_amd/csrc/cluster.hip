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
#include <atomic>
#include <stdlib.h>
#include <math.h>

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

__global__ __launch_bounds__(256, 2) void dpc_gram_bf16_kernel(const bf16* __restrict__ X, float* __restrict__ G, float* __restrict__ vec,
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


// --------------------------------------------------------------------------------------------
// The whole of cluster_dpc_knn for ONE image in ONE workgroup, N <= 256 tokens, bf16 (the BASELINE configuration: ViT-L/14-224).
//
// The N x N distance matrix never exists in memory: 8 waves hold it in their MFMA accumulators (wave w owns rows [32 w, 32 w + 32) x all
// 256 columns = 128 registers per lane) from the Gram product to the last use.  With the MFMA operands swapped (columns' fragment first) a
// lane holds, for row (lane & 15) of each of its two 16-row tiles, the columns t * 16 + 4 * (lane >> 4) + e (t < 16, e < 4): the four lanes
// {l, l + 16, l + 32, l + 48} own one row between them, so every per-row step (row max, the k-nearest radix select, delta, the
// nearest-centre argmin) is 64 register operations per lane and two lane exchanges — no LDS, no memory.
//
//   X (N x C bf16) streams HBM -> LDS once (LDS-DMA, K-tiles of 64 channels x 256 rows = 32 KiB, three stages, two in flight); the
//   same LDS tile feeds both MFMA operands (the Gram matrix is X X^T).
//   norms |x_j|^2 = the Gram diagonal (out of the accumulators, through 1 KiB of LDS)
//   d_ij = sqrt(max(|x_i|^2 + |x_j|^2 - 2 G_ij, 0)) / sqrt(C)   (torch.cdist's matmul form, tokenizer.py:82)            in place
//   row max, density from the k nearest (radix select on the float bit pattern, counts summed over the row's four lanes)   :88-94
//   delta / score against the other rows' density and row max (2 KiB of LDS)                                               :96-101
//   centres: score > threshold in index order, else the min_cluster_num best (one thread per token, ballots)               :103-107
//   assignment: argmin over the centre ROWS of column j (:111-113) = argmin over the centre COLUMNS of row j, because the matrix is
//   symmetric bit for bit (products commute, the k order of the accumulation is the same) — so a row's lanes find it in their own
//   registers; centres own themselves (:117-119).
// HBM traffic = x once + the three small outputs; one launch for the whole batch, no workspace.
// --------------------------------------------------------------------------------------------
namespace {
constexpr int FN = 256;                            // tokens per image the register-resident form holds
constexpr int F_STAGE = FN * 64 * 2;               // one K-tile: 256 rows x 128 B
constexpr int F_NST = 3;
constexpr int F_LDS = F_NST * F_STAGE + 6 * 1024 + 64;

struct FArgs {
    const bf16* x; const float* noise; const float* tmask;
    int64_t* idx; float* score; int64_t* index_down; int32_t* counts;
    int N, C, k, mcn;
    float thr, sqrtC, inv_sqrtC;
    int scale_by_mul;                              // sqrt(C) is a power of two: multiplying by 1 / sqrt(C) is the same rounding as dividing
    unsigned long long* tim;                       // SETOK_CLUSTER_TIMING=1: s_memtime at the phase boundaries of workgroup 0 (debug)
};

__device__ inline int f_swz(int row) { return (row >> 1) & 7; }

// The four lanes {l, l + 16, l + 32, l + 48} that own one row meet through gfx950's v_permlane16_swap / v_permlane32_swap: two exchanges in the
// vector ALU instead of two ds_bpermute round trips through the LDS crossbar (the radix select does one such meeting per bit).  The operand
// order of every addition is the one of the shuffle form (a + b with the partner's value second or first: the same fp32 result).
typedef unsigned u32x2c __attribute__((ext_vector_type(2)));
__device__ inline unsigned row4_add(unsigned x) {
    const u32x2c r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    const unsigned y = r[0] + r[1];
    const u32x2c q = __builtin_amdgcn_permlane32_swap(y, y, false, false);
    return q[0] + q[1];
}
__device__ inline float row4_addf(float x) {
    const u32x2c r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float y = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const u32x2c q = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ inline float row4_maxf(float x) {
    const u32x2c r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float y = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const u32x2c q = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ inline unsigned row4_minu(unsigned x) {
    const u32x2c r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    const unsigned y = min(r[0], r[1]);
    const u32x2c q = __builtin_amdgcn_permlane32_swap(y, y, false, false);
    return min(q[0], q[1]);
}
__device__ inline unsigned row4_maxu(unsigned x) {
    const u32x2c r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    const unsigned y = max(r[0], r[1]);
    const u32x2c q = __builtin_amdgcn_permlane32_swap(y, y, false, false);
    return max(q[0], q[1]);
}
__device__ inline float row4_minf(float x) {
    const u32x2c r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float y = fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const u32x2c q = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return fminf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}

template <bool MASKED>
__global__ __launch_bounds__(512) void dpc_fused_kernel(FArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_norm = reinterpret_cast<float*>(smem + F_NST * F_STAGE);      // [256]
    float* s_rho = s_norm + 256;                                           // [256]  density; -1 for rows >= N
    float* s_rmax = s_rho + 256;                                           // [256]  row max; +inf for rows >= N
    float* s_score = s_rmax + 256;                                         // [256]
    int* s_rank = reinterpret_cast<int*>(s_score + 256);                   // [256]  position of token i in the centre list
    float* s_raw = reinterpret_cast<float*>(s_rank + 256);                 // [256]  raw row max (token_mask path)
    unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(s_raw + 256);   // [4] centre flags; [4..7] token_mask bits

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int b = blockIdx.x, N = g.N;
    const bf16* Xb = g.x + (int64_t)b * N * g.C;

    // ---- Gram ---------------------------------------------------------------------------------------------------------------------
    unsigned src_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = i * 512 + tid, row = p >> 3, kc = (p & 7) ^ f_swz(row);
        src_off[i] = (unsigned)min(row, N - 1) * (unsigned)(g.C * 2) + kc * 16;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem) + wave * 1024;
    auto dma16 = [&](const char* base, unsigned off, unsigned lds_dst) {
        unsigned keep;
        const unsigned long long b64 = (unsigned long long)base;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);
        const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));
        const unsigned long long sb64 = (unsigned long long)lo | ((unsigned long long)hi32 << 32);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(sb64), "s"(lds_dst) : "memory");
    };
    auto issue = [&](int kt) {
        const unsigned sb = lds0 + (kt % F_NST) * F_STAGE;
        const char* base = reinterpret_cast<const char*>(Xb) + (size_t)kt * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(base, src_off[i], sb + i * 8192);
    };

    f32x4 acc[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[m][t][e] = 0.f;

    const bool stamp = g.tim && blockIdx.x == 0 && tid == 0;
    if (stamp) g.tim[0] = __builtin_amdgcn_s_memtime();
    const int nk = g.C / 64;
    issue(0);
    if (nk > 1) issue(1);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // K-tile kt has landed (this wave's pieces); kt + 1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");         // everyone's pieces; everyone is done reading stage (kt - 1) % 3
        if (kt + 2 < nk) issue(kt + 2);                                         // ... which is where K-tile kt + 2 goes
        const char* T = smem + (kt % F_NST) * F_STAGE;
        // Two k-steps of 32 channels; the 16 column fragments of a k-step in two batches of 8 that alternate between two register sets, so
        // that 8 ds_read_b128 are always in flight under the 16 MFMAs of the other batch (fetched two at a time just ahead of their use —
        // what hipcc makes of the plain loop — the wave stood at an lgkmcnt wait before every second MFMA pair).
        const char* rowp = T + (wave * 32 + l15) * 128;
        const char* colp = T + l15 * 128;
        const int sw = f_swz(l15);                                              // rows t * 16 + l15: the swizzle depends on l15 only
        auto slot = [&](int ks) { return ((ks * 4 + g4) ^ sw) << 4; };
        bf16x8 rf0[2], rf1[2], cA[8], cB[8];
        auto load_rows = [&](int ks, bf16x8 (&rf)[2]) {
#pragma unroll
            for (int m = 0; m < 2; ++m) rf[m] = *reinterpret_cast<const bf16x8*>(rowp + m * 2048 + slot(ks));
        };
        auto load_cols = [&](int ks, int half, bf16x8 (&cf)[8]) {
#pragma unroll
            for (int t = 0; t < 8; ++t) cf[t] = *reinterpret_cast<const bf16x8*>(colp + (half * 8 + t) * 2048 + slot(ks));
        };
        auto mma = [&](int half, const bf16x8 (&cf)[8], const bf16x8 (&rf)[2]) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                acc[0][half * 8 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cf[t], rf[0], acc[0][half * 8 + t], 0, 0, 0);
                acc[1][half * 8 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cf[t], rf[1], acc[1][half * 8 + t], 0, 0, 0);
            }
        };
        load_rows(0, rf0); load_cols(0, 0, cA);
        load_cols(0, 1, cB);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, cA, rf0);
        __builtin_amdgcn_sched_barrier(0);
        load_rows(1, rf1); load_cols(1, 0, cA);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, cB, rf0);
        __builtin_amdgcn_sched_barrier(0);
        load_cols(1, 1, cB);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, cA, rf1);
        mma(1, cB, rf1);
    }

    if (stamp) g.tim[1] = __builtin_amdgcn_s_memtime();
    // ---- norms = the diagonal: row i = 32 w + 16 m + l15 sits in tile t = 2 w + m at column offset l15 = 4 g4 + e ---------------------
    const int row0 = wave * 32 + l15, row1 = row0 + 16;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        float dv = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (t == 2 * wave + m && e == (l15 & 3)) dv = acc[m][t][e];
        if (g4 == (l15 >> 2)) s_norm[wave * 32 + m * 16 + l15] = dv;
    }
    __syncthreads();

    // ---- distances in place; raw row max ------------------------------------------------------------------------------------------------
    const float INF = __builtin_inff();
    const float n0 = s_norm[row0], n1 = s_norm[row1];
    float mx0 = 0.f, mx1 = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const f32x4 nj = *reinterpret_cast<const f32x4*>(s_norm + t * 16 + 4 * g4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = t * 16 + 4 * g4 + e < N;
            // v_sqrt_f32 itself (1 ulp): sqrtf() wraps it in a denormal rescale and two fma fix-ups for correct rounding — 17 instructions per
            // element, most of this phase — which the bf16 mode's contract (decisions equal wherever 64 ulps of d^2 cannot flip them) does not
            // need.  D stays symmetric bit for bit: the same instruction on the same operands.
            float d0 = __builtin_amdgcn_sqrtf(fmaxf((n0 + nj[e]) - 2.0f * acc[0][t][e], 0.f));
            float d1 = __builtin_amdgcn_sqrtf(fmaxf((n1 + nj[e]) - 2.0f * acc[1][t][e], 0.f));
            if (g.scale_by_mul) { d0 *= g.inv_sqrtC; d1 *= g.inv_sqrtC; } else { d0 = d0 / g.sqrtC; d1 = d1 / g.sqrtC; }
            acc[0][t][e] = ok ? d0 : INF;                                   // columns beyond N: never nearest, never a centre
            acc[1][t][e] = ok ? d1 : INF;
            mx0 = fmaxf(mx0, ok ? d0 : 0.f);
            mx1 = fmaxf(mx1, ok ? d1 : 0.f);
        }
    }
    mx0 = row4_maxf(mx0);
    mx1 = row4_maxf(mx1);

    if (stamp) g.tim[2] = __builtin_amdgcn_s_memtime();
    // ---- token_mask (:84-86): masked columns read (global max + 1) everywhere ------------------------------------------------------------
    constexpr bool masked = MASKED;
    float fill = 0.f;
    unsigned tmw[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};    // token_mask bits of this lane's columns: bit (t & 7) * 4 + e of word t >> 3
    bool tm_row0 = true, tm_row1 = true;
    if constexpr (MASKED) {
        if (g4 == 0) { s_raw[row0] = row0 < N ? mx0 : 0.f; s_raw[row1] = row1 < N ? mx1 : 0.f; }
        if (tid < 256) {
            const bool on = tid < N && g.tmask[(int64_t)b * N + tid] > 0.f;
            const unsigned long long bal = __ballot(on);
            if (lane == 0) s_mask[4 + wave] = bal;
        }
        __syncthreads();
        float gm = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) gm = fmaxf(gm, s_raw[lane + 64 * q]);
        gm = wave_max(gm);
        fill = gm + 1.0f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const unsigned nib = (unsigned)(s_mask[4 + (t >> 2)] >> ((t & 3) * 16 + 4 * g4)) & 0xfu;
            tmw[t >> 3] = (t & 7) == 0 ? nib : (tmw[t >> 3] | (nib << ((t & 7) * 4)));
        }
        tm_row0 = (s_mask[4 + (row0 >> 6)] >> (row0 & 63)) & 1ull;
        tm_row1 = (s_mask[4 + (row1 >> 6)] >> (row1 & 63)) & 1ull;
        mx0 = 0.f; mx1 = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = t * 16 + 4 * g4 + e < N;
                const bool on = (tmw[t >> 3] >> ((t & 7) * 4 + e)) & 1u;
                mx0 = fmaxf(mx0, ok ? (on ? acc[0][t][e] : fill) : 0.f);
                mx1 = fmaxf(mx1, ok ? (on ? acc[1][t][e] : fill) : 0.f);
            }
        mx0 = row4_maxf(mx0);
        mx1 = row4_maxf(mx1);
    }
    // value of (row m, t, e) as the reference's masked matrix holds it
    auto val = [&](int m, int t, int e) -> float {
        const float d = acc[m][t][e];
        if (!masked) return d;
        const bool on = (tmw[t >> 3] >> ((t & 7) * 4 + e)) & 1u;
        return (on || d == INF) ? d : fill;
    };

    // ---- density: mean of the squares of the k smallest of the row (self included), MSB-first radix select on the bit pattern ------------
    {
        const int k = g.k;
        // T = the largest bit pattern with (#values < T) <= k - 1, MSB first: the k-th smallest value.  A row is finished EARLY when a candidate
        // has exactly k values below it: the k smallest are then known as a set (`values < candidate`), which is all the sum needs — the
        // remaining low bits only matter when values tie at the k-th place.  Typically ~18 of the 31 steps.
        unsigned T0 = 0, T1 = 0;
        bool done0 = false, done1 = false;
        int top = 30;
        if constexpr (!MASKED) {
            // The leading bits the row's values (the self-distance 0 aside) have in common are decided without counting: where the common
            // prefix has a 1 only the self-distance lies below the candidate (1 <= k - 1), where it has a 0 all 256 values do (> k - 1) —
            // T starts as the prefix and the steps start at the highest bit in which two values of any row of the wave differ.  Distances of
            // high-dimensional features concentrate (d ~ sqrt(2) sigma): typically 8-9 of the ~18 steps go.  Exact for any input: rows
            // with a second exact zero, or columns beyond N (+inf), simply have no common prefix.
            if (k >= 2) {
                unsigned lo0 = 0xffffffffu, lo1 = 0xffffffffu, hi0 = 0u, hi1 = 0u;
                const bool self_lane = g4 == (l15 >> 2);
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const bool ts0 = t == 2 * wave, ts1 = t == 2 * wave + 1;           // wave-uniform: the 16-column tile that holds the diagonal
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        unsigned p0 = __float_as_uint(acc[0][t][e]), p1 = __float_as_uint(acc[1][t][e]);
                        hi0 = max(hi0, p0); hi1 = max(hi1, p1);
                        if (ts0) { if (self_lane && e == (l15 & 3)) p0 = 0xffffffffu; }
                        if (ts1) { if (self_lane && e == (l15 & 3)) p1 = 0xffffffffu; }
                        lo0 = min(lo0, p0); lo1 = min(lo1, p1);
                    }
                }
                lo0 = row4_minu(lo0); lo1 = row4_minu(lo1); hi0 = row4_maxu(hi0); hi1 = row4_maxu(hi1);
                const unsigned df0 = lo0 ^ hi0, df1 = lo1 ^ hi1;
                const int hb0 = df0 ? 31 - __builtin_clz(df0) : -1, hb1 = df1 ? 31 - __builtin_clz(df1) : -1;
                T0 = hb0 >= 0 ? (hb0 >= 31 ? 0u : lo0 & ~((2u << hb0) - 1u)) : lo0;
                T1 = hb1 >= 0 ? (hb1 >= 31 ? 0u : lo1 & ~((2u << hb1) - 1u)) : lo1;
                const int hbm = max(hb0, hb1);
                top = 30;
                while (top > 0 && __ballot(hbm >= top) == 0ull) --top;
            }
        }
        for (int bit = top; bit >= 0; --bit) {
            const unsigned c0 = T0 | (1u << bit), c1 = T1 | (1u << bit);
            // #values < candidate: the sign bit of (pattern - candidate) (both below 2^31) is shifted into a 32-bit register per element
            // (v_sub + v_alignbit: no carry flag, hence none of the wait states a v_cmp / v_addc chain needs) and popcounted per 32 elements
            // Eight elements per accumulator register, sixteen independent registers: a single register per 32 elements is a serial chain of
            // dependent v_alignbit (measured: ~8 cycles per instruction, 4 k cycles per bit with two waves per SIMD).
            int q0 = 0, q1 = 0;
#pragma unroll
            for (int tp = 0; tp < 8; ++tp) {
                unsigned w0 = 0, w1 = 0;
#pragma unroll
                for (int t = tp * 2; t < tp * 2 + 2; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        w0 = __builtin_amdgcn_alignbit(w0, __float_as_uint(val(0, t, e)) - c0, 31);
                        w1 = __builtin_amdgcn_alignbit(w1, __float_as_uint(val(1, t, e)) - c1, 31);
                    }
                q0 += __builtin_popcount(w0); q1 += __builtin_popcount(w1);
            }
            {
                const unsigned both = row4_add((unsigned)q0 | ((unsigned)q1 << 16));     // both rows' counts (<= 256 each) in one exchange
                q0 = (int)(both & 0xffffu); q1 = (int)(both >> 16);
            }
            if (!done0) { if (q0 <= k - 1) T0 = c0; else if (q0 == k) { T0 = c0; done0 = true; } }
            if (!done1) { if (q1 <= k - 1) T1 = c1; else if (q1 == k) { T1 = c1; done1 = true; } }
            if (__ballot(!(done0 && done1)) == 0ull) break;
        }
        float s0 = 0.f, s1 = 0.f; int q0 = 0, q1 = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v0 = val(0, t, e), v1 = val(1, t, e);
                const bool in0 = __float_as_uint(v0) < T0, in1 = __float_as_uint(v1) < T1;
                s0 += in0 ? v0 * v0 : 0.f; q0 += in0;
                s1 += in1 ? v1 * v1 : 0.f; q1 += in1;
            }
        s0 = row4_addf(s0);
        s1 = row4_addf(s1);
        {
            const unsigned both = row4_add((unsigned)q0 | ((unsigned)q1 << 16));
            q0 = (int)(both & 0xffffu); q1 = (int)(both >> 16);
        }
        // values tied AT the k-th place (q < k) enter with the k-th value; a row finished early has q == k and its T is a candidate, not a
        // value (its square may overflow: 0 * inf): no tie term then
        const float kth0 = __uint_as_float(T0), kth1 = __uint_as_float(T1);
        const float tie0 = q0 < k ? (float)(k - q0) * (kth0 * kth0) : 0.f, tie1 = q1 < k ? (float)(k - q1) * (kth1 * kth1) : 0.f;
        float rho0 = expf(-((s0 + tie0) / (float)k));
        float rho1 = expf(-((s1 + tie1) / (float)k));
        if (g.noise) {
            if (row0 < N) rho0 += g.noise[(int64_t)b * N + row0] * 1e-6f;
            if (row1 < N) rho1 += g.noise[(int64_t)b * N + row1] * 1e-6f;
        }
        if (masked) { if (!tm_row0) rho0 = 0.f; if (!tm_row1) rho1 = 0.f; }        // density * token_mask (:94)
        if (g4 == 0) {
            s_rho[row0] = row0 < N ? rho0 : -1.0f;  s_rmax[row0] = row0 < N ? mx0 : INF;
            s_rho[row1] = row1 < N ? rho1 : -1.0f;  s_rmax[row1] = row1 < N ? mx1 : INF;
        }
    }
    __syncthreads();

    if (stamp) g.tim[3] = __builtin_amdgcn_s_memtime();
    // ---- delta_i = min_j (rho_j > rho_i ? D_ij : rowmax_j) (:96-99), score = delta * rho (:101) -----------------------------------------------
    {
        const float r0 = s_rho[row0], r1 = s_rho[row1];
        float dm0 = INF, dm1 = INF;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const f32x4 rj = *reinterpret_cast<const f32x4*>(s_rho + t * 16 + 4 * g4);
            const f32x4 mj = *reinterpret_cast<const f32x4*>(s_rmax + t * 16 + 4 * g4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dm0 = fminf(dm0, rj[e] > r0 ? val(0, t, e) : mj[e]);
                dm1 = fminf(dm1, rj[e] > r1 ? val(1, t, e) : mj[e]);
            }
        }
        dm0 = row4_minf(dm0);
        dm1 = row4_minf(dm1);
        if (g4 == 0) {
            const float sc0 = dm0 * r0, sc1 = dm1 * r1;
            s_score[row0] = sc0; s_score[row1] = sc1;
            if (row0 < N) g.score[(int64_t)b * N + row0] = sc0;
            if (row1 < N) g.score[(int64_t)b * N + row1] = sc1;
        }
    }
    __syncthreads();

    if (stamp) g.tim[4] = __builtin_amdgcn_s_memtime();
    // ---- centres (:103-107): one thread per token ----------------------------------------------------------------------------------------------------
    if (tid < 256) {
        const float si = tid < N ? s_score[tid] : 0.f;
        const unsigned long long bal = __ballot(tid < N && si > g.thr);
        if (lane == 0) s_mask[wave] = bal;
    }
    __syncthreads();
    const bool none = (s_mask[0] | s_mask[1] | s_mask[2] | s_mask[3]) == 0ull;      // uniform
    __syncthreads();
    if (none && tid < 256) {                                   // the min_cluster_num largest scores, ties to the lower index, in index order
        const float si = tid < N ? s_score[tid] : 0.f;
        int rank = 0;
        for (int j = 0; j < N; ++j) { const float sj = s_score[j]; rank += (sj > si) || (sj == si && j < tid); }
        const unsigned long long bal = __ballot(tid < N && rank < g.mcn);
        if (lane == 0) s_mask[wave] = bal;
    }
    __syncthreads();
    const unsigned long long cm0 = s_mask[0], cm1 = s_mask[1], cm2 = s_mask[2], cm3 = s_mask[3];
    const int L = __builtin_popcountll(cm0) + __builtin_popcountll(cm1) + __builtin_popcountll(cm2) + __builtin_popcountll(cm3);
    if (tid < 256) {
        const int w = tid >> 6;
        const unsigned long long mine = w == 0 ? cm0 : (w == 1 ? cm1 : (w == 2 ? cm2 : cm3));
        int pos = __builtin_popcountll(mine & ((1ull << (tid & 63)) - 1ull));
        if (w > 0) pos += __builtin_popcountll(cm0);
        if (w > 1) pos += __builtin_popcountll(cm1);
        if (w > 2) pos += __builtin_popcountll(cm2);
        s_rank[tid] = pos;
        if ((mine >> (tid & 63)) & 1ull) g.index_down[(int64_t)b * N + pos] = tid;
        if (tid >= L && tid < N) g.index_down[(int64_t)b * N + tid] = -1;
        if (tid == 0) g.counts[b] = L;
    }
    __syncthreads();

    if (stamp) g.tim[5] = __builtin_amdgcn_s_memtime();
    // ---- assignment (:111-119): first argmin over the centres, read along the token's OWN row (the matrix is symmetric) -------------------------------------
    {
        float b0 = INF, b1 = INF; int j0 = 0x7fffffff, j1 = 0x7fffffff;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const unsigned long long wsel = (t >> 2) == 0 ? cm0 : ((t >> 2) == 1 ? cm1 : ((t >> 2) == 2 ? cm2 : cm3));
            const unsigned nib = (unsigned)(wsel >> ((t & 3) * 16 + 4 * g4)) & 0xfu;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool c = (nib >> e) & 1u;
                const int j = t * 16 + 4 * g4 + e;
                const float v0 = acc[0][t][e], v1 = acc[1][t][e];             // D[centre][token] is the RAW distance when the token is unmasked
                if (c && v0 < b0) { b0 = v0; j0 = j; }
                if (c && v1 < b1) { b1 = v1; j1 = j; }
            }
        }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float ob0 = __shfl_xor(b0, o, 64), ob1 = __shfl_xor(b1, o, 64);
            const int oj0 = __shfl_xor(j0, o, 64), oj1 = __shfl_xor(j1, o, 64);
            if (ob0 < b0 || (ob0 == b0 && oj0 < j0)) { b0 = ob0; j0 = oj0; }
            if (ob1 < b1 || (ob1 == b1 && oj1 < j1)) { b1 = ob1; j1 = oj1; }
        }
        if (g4 == 0) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int row = m ? row1 : row0;
                if (row >= N) continue;
                const int jb = m ? j1 : j0;
                const bool tm_on = m ? tm_row1 : tm_row0;
                int lab = ((masked && !tm_on) || jb >= FN) ? 0 : s_rank[jb];                 // a masked token's column reads `fill` in every centre row: first index
                const unsigned long long wsel = (row >> 6) == 0 ? cm0 : ((row >> 6) == 1 ? cm1 : ((row >> 6) == 2 ? cm2 : cm3));
                if ((wsel >> (row & 63)) & 1ull) lab = s_rank[row];             // a centre owns itself
                g.idx[(int64_t)b * N + row] = lab;
            }
        }
    }
    if (stamp) g.tim[6] = __builtin_amdgcn_s_memtime();
}
}  // namespace

// --------------------------------------------------------------------------------------------
// 256 < N <= 576 tokens (BASELINE config 4: ViT-L/14-336, 24 x 24 patches), bf16: ONE launch, no N x N workspace.
//
// 331 k distances do not fit the registers of one CU (131 k per CU), so an image is cut into STRIPS of 128 rows: a workgroup holds its strip
// (128 rows x all 576 columns) in its MFMA accumulators — wave w owns rows [16 w, 16 w + 16) of the strip = 36 tiles of 16 x 16 = 144 registers per
// lane, a row again shared by the four lanes {l, l + 16, l + 32, l + 48} — and runs every per-row step on it exactly as dpc_fused_kernel does
// (distances in place, row max, radix-select density, delta / score, nearest-centre argmin along the token's own row).  What a strip needs from the
// OTHER strips of its image is three vectors of N floats, exchanged through global memory inside the launch:
//   |x_j|^2 of all columns      none: the 36 diagonal 16 x 16 blocks are multiplied redundantly by every workgroup (+12 % MFMAs; the same
//                                instruction on the same operands as the block in its owner's accumulators: bit-identical, D stays symmetric)
//   rho_j, rowmax_j              exchange 1 (after the density)          tokenizer.py:96-99 compares against every other token
//   score_j                      exchange 2 (after delta)                tokenizer.py:103-107 ranks all tokens; every workgroup then selects the
//                                                                        centres itself (identical results), strip 0 writes index_down / counts
//   (token_mask: the global maximum of the raw distances, exchange 0)    tokenizer.py:84-86
// An exchange = plain stores -> workgroup barrier -> ONE agent-scope release + a relaxed counter increment; the readers poll the counter (relaxed),
// take ONE agent-scope acquire, and read with plain loads (cdna guide, Guideline 16).  Every strip writes whole 128-byte lines of its own; the
// counters live in a line of their own.
// Deadlock freedom without any assumption on dispatch order: the grid is PERSISTENT (at most one workgroup per USABLE CU — the device's CU count
// cut down to the stream's CU mask — so all of it is co-resident) and pulls (image, strip) items from Q queues in image-major order.  A queue has at
// most ONE incompletely pulled image at any time, whose pulled strips (<= S - 1 of them) wait in their holders; every other held item belongs to a
// completely pulled image, whose S holders are all running and need nobody else.  So at most Q (S - 1) workgroups can be waiting for an item nobody
// has pulled, and the launch requires grid > Q (S - 1): Q = min(8, (grid - 1) / (S - 1)), which is >= 1 whenever grid >= S (checked on the host; a
// device with fewer than S usable CUs is refused).  The round-3 code fixed Q = 8 without that check: a 32-CU partition at S = 5 could hang.
// With Q = 8 a queue is an XCD's (images b = x mod 8, blocks x mod 8 — observed placement, a speed matter only): the strips of an image re-read its
// x through ONE L2.
// --------------------------------------------------------------------------------------------
namespace {
constexpr int SN = 576, SNT = SN / 16, SROWS = 128;
constexpr int S_STAGE = SN * 128;                       // one K-tile: 576 rows x 64 channels x 2 B = 72 KiB
constexpr int S_LDS = 2 * S_STAGE + 256;               // + the item word
constexpr int S_HDR = 32;                               // workspace header (ints): [0..7] queue heads
constexpr int S_IMG = 3 * SN + 8 * 32 + 32;             // per image (floats): rho, rmax, score; 8 lines of one strip's raw maximum each; one line of counters

struct SArgs {
    const bf16* x; const float* noise; const float* tmask;
    int64_t* idx; float* score; int64_t* index_down; int32_t* counts;
    float* ws;
    int B, N, C, k, mcn, strips;
    float thr, sqrtC, inv_sqrtC;
    int scale_by_mul;
    int nq;                                                 // number of item queues, 1..8: grid > nq (strips - 1)
};

__device__ inline void strip_publish(int* counter) {          // every thread's plain stores of the exchange are issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ inline void strip_wait(int* counter, int target) {
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(4);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <bool MASKED>
__global__ __launch_bounds__(512) void dpc_strip_kernel(SArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_item = reinterpret_cast<int*>(smem + 2 * S_STAGE);
    // after the Gram product the two stages are free: the per-token vectors live in stage 0
    float* s_norm = reinterpret_cast<float*>(smem);                         // [640]
    float* s_rho = s_norm + 640;                                            // [640]  density; -1 for tokens >= N
    float* s_rmax = s_rho + 640;                                            // [640]  row max; +inf for tokens >= N
    float* s_score = s_rmax + 640;                                          // [640]
    int* s_rank = reinterpret_cast<int*>(s_score + 640);                    // [640]  position of token i in the centre list
    float* s_red = reinterpret_cast<float*>(s_rank + 640);                  // [16]
    unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(s_red + 16);   // [0..9] centre flags, [16..25] token_mask bits

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int N = g.N, S = g.strips;
    const float INF = __builtin_inff();
    int* heads = reinterpret_cast<int*>(g.ws);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem) + wave * 1024;
    const int nq = g.nq;
    const int home = blockIdx.x % nq;

    for (;;) {
        // ---- next (image, strip): the home queue first, then the others ------------------------------------------------------------------------
        if (tid == 0) {
            int it = -1;
            for (int o = 0; o < nq && it < 0; ++o) {
                int x = home + o;
                if (x >= nq) x -= nq;
                const int per = (g.B - x + nq - 1) / nq;                     // images x, x + nq, ... < B
                if (per <= 0) continue;
                if (__hip_atomic_load(heads + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= per * S) continue;
                const int q = __hip_atomic_fetch_add(heads + x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (q < per * S) it = ((x + nq * (q / S)) << 4) | (q % S);
            }
            *s_item = it;
        }
        __syncthreads();
        const int item = *s_item;
        if (item < 0) return;
        const int b = item >> 4, strip = item & 15;
        const bf16* Xb = g.x + (int64_t)b * N * g.C;
        float* wimg = g.ws + S_HDR + (int64_t)b * S_IMG;
        float* w_rho = wimg, *w_rmax = wimg + SN, *w_score = wimg + 2 * SN, *w_raw = wimg + 3 * SN;
        int* w_cnt = reinterpret_cast<int*>(wimg + 3 * SN + 8 * 32);

        // ---- Gram strip: rows [128 strip, + 128) x all columns; the diagonal blocks of ALL column tiles on the side -------------------------
        // LDS-DMA source: piece p = 512 i + tid -> row p >> 3 = 64 i + (tid >> 3), 16-byte slot (p & 7) ^ swizzle(row); the swizzle (row >> 1) & 7
        // does not depend on i (64 i >> 1 is a multiple of 8), so ONE lane offset serves all nine pieces and the row advance of 64 rows goes into
        // the wave-uniform base; only when N < 576 the last pieces' rows are clamped to N - 1 (per-lane offsets, computed on the fly).
        const int prow = tid >> 3;
        const unsigned poff = (unsigned)prow * (unsigned)(g.C * 2) + (unsigned)(((tid & 7) ^ f_swz(prow)) << 4);
        auto dma16 = [&](const char* base, unsigned off, unsigned lds_dst) {
            unsigned keep;
            const unsigned long long b64 = (unsigned long long)base;
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);
            const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));
            const unsigned long long sb64 = (unsigned long long)lo | ((unsigned long long)hi32 << 32);
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(off), "s"(sb64), "s"(lds_dst) : "memory");
        };
        const bool full = N == SN;                          // uniform
        auto issue = [&](int kt) {
            const unsigned sb = lds0 + (kt & 1) * S_STAGE;
            const char* base = reinterpret_cast<const char*>(Xb) + (size_t)kt * 128;
            if (full) {
#pragma unroll 1
                for (int i = 0; i < 9; ++i) dma16(base + (size_t)i * 64 * (size_t)(g.C * 2), poff, sb + i * 8192);
            } else {
#pragma unroll 1
                for (int i = 0; i < 9; ++i) {
                    const unsigned off = (unsigned)min(prow + 64 * i, N - 1) * (unsigned)(g.C * 2) + (unsigned)(((tid & 7) ^ f_swz(prow)) << 4);
                    dma16(base, off, sb + i * 8192);
                }
            }
        };
        f32x4 acc[SNT];
        f32x4 accd[5];                                    // diagonal blocks of column tiles wave, wave + 8, ... (< 36)
#pragma unroll
        for (int t = 0; t < SNT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) accd[i][e] = 0.f;
        const int nk = g.C / 64;
        const int lrow = min(strip * SROWS + wave * 16 + l15, SN - 1);            // this lane's row of the LDS tile (rows >= N hold row N - 1: finite, unused)
        issue(0);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // this wave's pieces of K-tile kt
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");         // everyone's; everyone is done reading the other stage
            if (kt + 1 < nk) issue(kt + 1);
            const char* T = smem + (kt & 1) * S_STAGE;
            const char* rowp = T + lrow * 128;
            const char* colp = T + l15 * 128;
            const int sw = f_swz(l15), swr = f_swz(lrow);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 rf = *reinterpret_cast<const bf16x8*>(rowp + (((ks * 4 + g4) ^ swr) << 4));
                const int so = ((ks * 4 + g4) ^ sw) << 4;
#pragma unroll
                for (int bt = 0; bt < 6; ++bt) {                                  // the 36 column fragments in six batches of six
                    bf16x8 cf[6];
#pragma unroll
                    for (int u = 0; u < 6; ++u) cf[u] = *reinterpret_cast<const bf16x8*>(colp + (bt * 6 + u) * 2048 + so);
#pragma unroll
                    for (int u = 0; u < 6; ++u) acc[bt * 6 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cf[u], rf, acc[bt * 6 + u], 0, 0, 0);
                }
                // the diagonal blocks of column tiles wave, wave + 8, ... : their own fragment reads (a wave-uniform tile offset), no branches
                {
                    bf16x8 df[5];
#pragma unroll
                    for (int i = 0; i < 5; ++i) df[i] = *reinterpret_cast<const bf16x8*>(colp + min(wave + 8 * i, SNT - 1) * 2048 + so);
#pragma unroll
                    for (int i = 0; i < 5; ++i) accd[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[i], df[i], accd[i], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                   // every wave is done with the stages: stage 0 now holds the per-token vectors

        // ---- norms = the Gram diagonal, for every column -----------------------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int t = wave + 8 * i;
            if (t < SNT) {
                const int c = l15 & 3;
                const float dv = c == 0 ? accd[i][0] : (c == 1 ? accd[i][1] : (c == 2 ? accd[i][2] : accd[i][3]));
                if (g4 == (l15 >> 2)) s_norm[t * 16 + l15] = dv;
            }
        }
        __syncthreads();

        // ---- distances in place; raw row max ----------------------------------------------------------------------------------------------------
        const int row = strip * SROWS + wave * 16 + l15;   // token index of this lane's row
        const bool rvalid = row < N;
        const int tself = strip * 8 + wave;                // the column tile that holds the diagonal of this wave's rows
        const float ni = s_norm[min(row, SN - 1)];
        float mx = 0.f;
        // (the item loop makes N and the lane's column indices loop invariants: hoisted, the 144 column-validity masks and indices are 288 + 144
        // registers that live across the whole kernel and spill — an opaque copy per item keeps them where they are used)
        int Nrt = N, c4 = 4 * g4;
        asm volatile("" : "+s"(Nrt));
        asm volatile("" : "+v"(c4));
#pragma unroll
        for (int t = 0; t < SNT; ++t) {
            const f32x4 nj = *reinterpret_cast<const f32x4*>(s_norm + t * 16 + 4 * g4);
            if (t % 6 == 5) __builtin_amdgcn_sched_barrier(0);               // (keeps the 36 vector reads from being issued in one cluster)
            if ((t + 1) * 16 <= Nrt) {                                        // a whole tile of real columns (all of them at N = 576): no masks
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d = __builtin_amdgcn_sqrtf(fmaxf((ni + nj[e]) - 2.0f * acc[t][e], 0.f));
                    if (g.scale_by_mul) d *= g.inv_sqrtC; else d = d / g.sqrtC;
                    acc[t][e] = d;
                    mx = fmaxf(mx, d);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool ok = t * 16 + c4 + e < Nrt;
                    float d = __builtin_amdgcn_sqrtf(fmaxf((ni + nj[e]) - 2.0f * acc[t][e], 0.f));
                    if (g.scale_by_mul) d *= g.inv_sqrtC; else d = d / g.sqrtC;
                    acc[t][e] = ok ? d : INF;                                 // columns beyond N: never nearest, never a centre
                    mx = fmaxf(mx, ok ? d : 0.f);
                }
            }
        }
        mx = row4_maxf(mx);

        // ---- token_mask (:84-86): masked columns read (global max + 1) everywhere -------------------------------------------------------------------
        constexpr bool masked = MASKED;
        float fill = 0.f;
        unsigned tmw[5] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};   // bit (t & 7) * 4 + e of word t >> 3
        bool tm_row = true;
        if constexpr (MASKED) {
            float sm = rvalid ? mx : 0.f;                                        // the strip's raw maximum
            sm = wave_max(sm);
            if (lane == 0) s_red[wave] = sm;
            for (int i = tid; i < 640; i += 512) {
                const bool on = i < N && g.tmask[(int64_t)b * N + i] > 0.f;
                const unsigned long long bal = __ballot(on);
                if (lane == 0) s_mask[16 + (i >> 6)] = bal;
            }
            __syncthreads();
            if (tid == 0) {
                float m8 = 0.f;
                for (int w = 0; w < 8; ++w) m8 = fmaxf(m8, s_red[w]);
                w_raw[strip * 32] = m8;
            }
            strip_publish(w_cnt + 0);
            strip_wait(w_cnt + 0, S);
            float gm = 0.f;
            for (int q = 0; q < S; ++q) gm = fmaxf(gm, w_raw[q * 32]);
            fill = gm + 1.0f;
#pragma unroll
            for (int t = 0; t < SNT; ++t) {
                const unsigned nib = (unsigned)(s_mask[16 + (t >> 2)] >> ((t & 3) * 16 + 4 * g4)) & 0xfu;
                tmw[t >> 3] = (t & 7) == 0 ? nib : (tmw[t >> 3] | (nib << ((t & 7) * 4)));
            }
            tm_row = rvalid && ((s_mask[16 + (row >> 6)] >> (row & 63)) & 1ull);
            mx = 0.f;
#pragma unroll
            for (int t = 0; t < SNT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = acc[t][e];                                  // columns beyond N hold +inf
                    const bool on = (tmw[t >> 3] >> ((t & 7) * 4 + e)) & 1u;
                    mx = fmaxf(mx, d == INF ? 0.f : (on ? d : fill));
                }
            mx = row4_maxf(mx);
        }
        auto val = [&](int t, int e) -> float {              // the entry as the reference's masked matrix holds it
            const float d = acc[t][e];
            if (!masked) return d;
            const bool on = (tmw[t >> 3] >> ((t & 7) * 4 + e)) & 1u;
            return (on || d == INF) ? d : fill;
        };

        // ---- density: mean of the squares of the k smallest of the row (self included): MSB-first radix select on the bit pattern ---------------
        float rho;
        {
            const int k = g.k;
            unsigned T0 = 0;
            bool done = false;
            int top = 30;
            if constexpr (!MASKED) {
                if (k >= 2) {                              // the leading bits all of the row's values (the self-distance aside) share are decided without counting
                    unsigned lo = 0xffffffffu, hi = 0u;
                    const bool self_lane = g4 == (l15 >> 2);
#pragma unroll
                    for (int t = 0; t < SNT; ++t) {
                        const bool ts = t == tself;         // wave-uniform
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            unsigned p0 = __float_as_uint(acc[t][e]);
                            hi = max(hi, p0);
                            if (ts) { if (self_lane && e == (l15 & 3)) p0 = 0xffffffffu; }
                            lo = min(lo, p0);
                        }
                    }
                    lo = row4_minu(lo); hi = row4_maxu(hi);
                    const unsigned df = lo ^ hi;
                    const int hb = df ? 31 - __builtin_clz(df) : -1;
                    T0 = hb >= 0 ? (hb >= 31 ? 0u : lo & ~((2u << hb) - 1u)) : lo;
                    top = 30;
                    while (top > 0 && __ballot(hb >= top) == 0ull) --top;
                }
            }
            for (int bit = top; bit >= 0; --bit) {
                const unsigned c0 = T0 | (1u << bit);
                int q = 0;
#pragma unroll
                for (int tp = 0; tp < SNT / 2; ++tp) {      // eight elements per count register, eighteen independent registers
                    unsigned w0 = 0;
#pragma unroll
                    for (int t = tp * 2; t < tp * 2 + 2; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e) w0 = __builtin_amdgcn_alignbit(w0, __float_as_uint(val(t, e)) - c0, 31);
                    q += __builtin_popcount(w0);
                }
                q = (int)row4_add((unsigned)q);
                if (!done) { if (q <= k - 1) T0 = c0; else if (q == k) { T0 = c0; done = true; } }
                if (__ballot(!done) == 0ull) break;
            }
            float s0 = 0.f; int q = 0;
#pragma unroll
            for (int t = 0; t < SNT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v0 = val(t, e);
                    const bool in0 = __float_as_uint(v0) < T0;
                    s0 += in0 ? v0 * v0 : 0.f; q += in0;
                }
            s0 = row4_addf(s0);
            q = (int)row4_add((unsigned)q);
            const float kth = __uint_as_float(T0);
            const float tie = q < k ? (float)(k - q) * (kth * kth) : 0.f;
            rho = expf(-((s0 + tie) / (float)k));
            if (g.noise && rvalid) rho += g.noise[(int64_t)b * N + row] * 1e-6f;
            if (masked && !tm_row) rho = 0.f;                                  // density * token_mask (:94)
            if (g4 == 0 && rvalid) { w_rho[row] = rho; w_rmax[row] = mx; }
        }
        // ---- exchange 1: every strip's rho / row max ---------------------------------------------------------------------------------------------
        strip_publish(w_cnt + 1);
        strip_wait(w_cnt + 1, S);
        for (int i = tid; i < 640; i += 512) { s_rho[i] = i < N ? w_rho[i] : -1.0f; s_rmax[i] = i < N ? w_rmax[i] : INF; }
        __syncthreads();

        // ---- delta_i = min_j (rho_j > rho_i ? D_ij : rowmax_j) (:96-99), score = delta * rho (:101) -----------------------------------------------
        {
            float dm = INF;
#pragma unroll
            for (int t = 0; t < SNT; ++t) {
                const f32x4 rj = *reinterpret_cast<const f32x4*>(s_rho + t * 16 + 4 * g4);
                const f32x4 mj = *reinterpret_cast<const f32x4*>(s_rmax + t * 16 + 4 * g4);
#pragma unroll
                for (int e = 0; e < 4; ++e) dm = fminf(dm, rj[e] > rho ? val(t, e) : mj[e]);
            }
            dm = row4_minf(dm);
            if (g4 == 0 && rvalid) {
                const float sc = dm * rho;
                w_score[row] = sc;
                g.score[(int64_t)b * N + row] = sc;
            }
        }
        // ---- exchange 2: every strip's scores ------------------------------------------------------------------------------------------------------
        strip_publish(w_cnt + 2);
        strip_wait(w_cnt + 2, S);
        for (int i = tid; i < 640; i += 512) s_score[i] = i < N ? w_score[i] : 0.f;
        __syncthreads();

        // ---- centres (:103-107): every workgroup of the image selects them itself (same inputs, same result) -------------------------------------
        for (int i = tid; i < 640; i += 512) {
            const unsigned long long bal = __ballot(i < N && s_score[i] > g.thr);
            if (lane == 0) s_mask[i >> 6] = bal;
        }
        __syncthreads();
        unsigned long long any = 0ull;
#pragma unroll
        for (int w = 0; w < 10; ++w) any |= s_mask[w];
        const bool none = any == 0ull;                       // uniform
        __syncthreads();
        if (none) {                                          // the min_cluster_num largest scores, ties to the lower index, in index order
            for (int i = tid; i < 640; i += 512) {
                const float si = i < N ? s_score[i] : 0.f;
                int rank = 0;
                for (int j = 0; j < N; ++j) { const float sj = s_score[j]; rank += (sj > si) || (sj == si && j < i); }
                const unsigned long long bal = __ballot(i < N && rank < g.mcn);
                if (lane == 0) s_mask[i >> 6] = bal;
            }
        }
        __syncthreads();
        int L = 0;
#pragma unroll
        for (int w = 0; w < 10; ++w) L += __builtin_popcountll(s_mask[w]);
        for (int i = tid; i < 640; i += 512) {
            const int w = i >> 6;
            const unsigned long long mine = s_mask[w];
            int pos = __builtin_popcountll(mine & ((1ull << (i & 63)) - 1ull));
            for (int v = 0; v < w; ++v) pos += __builtin_popcountll(s_mask[v]);
            s_rank[i] = pos;
            if (strip == 0) {
                if ((mine >> (i & 63)) & 1ull) g.index_down[(int64_t)b * N + pos] = i;
                if (i >= L && i < N) g.index_down[(int64_t)b * N + i] = -1;
            }
        }
        if (strip == 0 && tid == 0) g.counts[b] = L;
        __syncthreads();

        // ---- assignment (:111-119): first argmin over the centres, read along the token's OWN row (the matrix is symmetric) ---------------------
        {
            float bd = INF; int code = 0x7fffffff;                               // code = 4 t + e: a literal per element, the column index is formed once at the end
#pragma unroll
            for (int t = 0; t < SNT; ++t) {
                const unsigned nib = (unsigned)(s_mask[t >> 2] >> ((t & 3) * 16 + 4 * g4)) & 0xfu;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool c = (nib >> e) & 1u;
                    const float v0 = acc[t][e];                                  // D[centre][token] is the RAW distance when the token is unmasked
                    if (c && v0 < bd) { bd = v0; code = 4 * t + e; }             // (ascending column order within the lane: the first minimum wins)
                }
            }
            int jb = code == 0x7fffffff ? 0x7fffffff : (code >> 2) * 16 + 4 * g4 + (code & 3);
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float ob = __shfl_xor(bd, o, 64);
                const int oj = __shfl_xor(jb, o, 64);
                if (ob < bd || (ob == bd && oj < jb)) { bd = ob; jb = oj; }
            }
            if (g4 == 0 && rvalid) {
                int lab = ((masked && !tm_row) || jb >= SN) ? 0 : s_rank[jb];
                if ((s_mask[row >> 6] >> (row & 63)) & 1ull) lab = s_rank[row];    // a centre owns itself
                g.idx[(int64_t)b * N + row] = lab;
            }
        }
        __syncthreads();                                    // before the next item overwrites stage 0 and the item word
    }
}
}  // namespace

// |x_i|^2 = G_ii (the same fma chain as every other Gram entry) -> vec[b][3][i]
__global__ void dpc_diag_kernel(const float* __restrict__ G, float* __restrict__ vec, int B, int N) {
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
__global__ __launch_bounds__(256) void dpc_dist_rows_kernel(float* __restrict__ G, float* __restrict__ vec, const float* __restrict__ noise,
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
__global__ void dpc_gmax_kernel(const float* __restrict__ vec, float* __restrict__ gmax, int N) {
    const int b = blockIdx.x;
    float m = 0.f;
    for (int j = threadIdx.x; j < N; j += 64) m = fmaxf(m, vec[((int64_t)b * 4 + 1) * N + j]);
    m = wave_max(m);
    if (threadIdx.x == 0) gmax[b] = m;
}

template <int PL>
__global__ __launch_bounds__(256) void dpc_masked_density_kernel(float* __restrict__ D, float* __restrict__ vec, const float* __restrict__ noise,
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
__global__ __launch_bounds__(256) void dpc_score_kernel(const float* __restrict__ D, float* __restrict__ vec, float* __restrict__ score,
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
__global__ __launch_bounds__(256) void dpc_select_kernel(const float* __restrict__ score, float thr, int mcn, int N,
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
__global__ __launch_bounds__(256) void dpc_assign_kernel(const float* __restrict__ D, const float* __restrict__ vec,
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

static bool fused_disabled() {                      // SETOK_CLUSTER_FUSED=0: the multi-kernel path for every shape (A/B runs, tests of that path);
    const char* e = getenv("SETOK_CLUSTER_FUSED");  // read per call so that one process can exercise both
    return e && e[0] == '0';
}

extern "C" int setok_cluster_workspace(int dtype, int B, int N, int C, int64_t* dist_floats, int64_t* vec_floats) {
    SETOK_CHECK_ARG(dist_floats && vec_floats && B >= 0 && N > 0 && C > 0, "setok_cluster_workspace: bad argument");
    const bool fused = dtype == SETOK_BF16 && C % 64 == 0 && N <= FN && !fused_disabled();
    const bool strips = dtype == SETOK_BF16 && C % 64 == 0 && N > FN && N <= SN && !fused_disabled();
    *dist_floats = (fused || strips) ? 0 : (int64_t)B * N * N;                 // the N x N matrix exists in memory only on the multi-kernel path
    *vec_floats = fused ? 0 : (strips ? (int64_t)S_HDR + (int64_t)B * S_IMG : (int64_t)B * 4 * N);
    return SETOK_OK;
}

extern "C" int setok_cluster_dpc_knn(void* stream, int dtype, const void* x, int B, int N, int C, int k,
                                     float threshold, int min_cluster_num, const float* noise,
                                     const float* token_mask, int64_t* idx_cluster, float* score,
                                     int64_t* index_down, int32_t* counts, float* dist_ws, float* vec_ws) {
    SETOK_CHECK_ARG(x && idx_cluster && score && index_down && counts, "setok_cluster_dpc_knn: null operand");
    SETOK_CHECK_ARG(B > 0 && N > 0 && N <= 64 * MAXPL && C > 0, "setok_cluster_dpc_knn: need 0 < N <= %d (got N=%d)", 64 * MAXPL, N);
    SETOK_CHECK_ARG(k >= 1 && k <= N, "setok_cluster_dpc_knn: k=%d out of range (torch.topk would raise), N=%d", k, N);
    SETOK_CHECK_ARG(min_cluster_num >= 1 && min_cluster_num <= N, "setok_cluster_dpc_knn: min_cluster_num=%d out of range, N=%d", min_cluster_num, N);
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * N;
    const float sqrtC = (float)sqrt((double)C);
    // algorithmic bytes: x read once; idx_cluster (int64), score (fp32), index_down (int64, <= N) written
    const bool one_launch = dtype == SETOK_BF16 && C % 64 == 0 && N <= FN && !fused_disabled();
    SetokProfScope prof(s, SETOK_PROF_CLUSTER, 0, 2.0 * B * (double)N * N * C, (double)B * ((double)N * C * (dtype == SETOK_BF16 ? 2 : 4) + N * 8.0 + N * 4.0 + N * 8.0),
                        one_launch);      // the single launch carries the profiler's timestamps itself
    if (one_launch) {
        // the whole call in one launch, one workgroup per image, no workspace (the BASELINE configuration)
        static SetokDeviceOnce once_f;
        if (!once_f.run([] { return hipFuncSetAttribute((const void*)dpc_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS) == hipSuccess &&
                                    hipFuncSetAttribute((const void*)dpc_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS) == hipSuccess; }))
            return setok_fail(SETOK_ELAUNCH, "setok_cluster_dpc_knn: cannot raise the dynamic LDS limit");
        int ex = 0;
        const float mant = frexpf(sqrtC, &ex);
        static const bool timing = [] { const char* e = getenv("SETOK_CLUSTER_TIMING"); return e && e[0] == '1'; }();
        static unsigned long long* tim = nullptr;
        if (timing && !tim && hipMalloc(&tim, 8 * 8) != hipSuccess) tim = nullptr;
        FArgs a{(const bf16*)x, noise, token_mask, idx_cluster, score, index_down, counts, N, C, k, min_cluster_num, threshold, sqrtC, 1.0f / sqrtC,
                mant == 0.5f ? 1 : 0, timing ? tim : nullptr};
        const hipEvent_t e0 = setok_prof_start_event(), e1 = setok_prof_stop_event();
        if (token_mask) setok_launch(dpc_fused_kernel<true>, dim3(B), dim3(512), F_LDS, s, e0, e1, a); else setok_launch(dpc_fused_kernel<false>, dim3(B), dim3(512), F_LDS, s, e0, e1, a);
        SETOK_CHECK_LAUNCH("setok_cluster_dpc_knn(fused)");
        if (timing && tim) {
            unsigned long long h[8];
            if (hipMemcpy(h, tim, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
                fprintf(stderr, "[cluster timing] B=%d N=%d C=%d  cycles of workgroup 0: gram %llu, distances %llu, density %llu, delta %llu, centres %llu, assignment %llu, total %llu\n",
                        B, N, C, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[6] - h[0]);
        }
        return SETOK_OK;
    }
    if (dtype == SETOK_BF16 && C % 64 == 0 && N > FN && N <= SN && !fused_disabled()) {
        // 256 < N <= 576 (cfg4): one launch of a persistent grid, a workgroup per (image, 128-row strip), no N x N workspace
        SETOK_CHECK_ARG(vec_ws, "setok_cluster_dpc_knn: this shape needs the vector workspace (setok_cluster_workspace)");
        static SetokDeviceOnce once_s;
        if (!once_s.run([] { return hipFuncSetAttribute((const void*)dpc_strip_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS) == hipSuccess &&
                                    hipFuncSetAttribute((const void*)dpc_strip_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS) == hipSuccess; }))
            return setok_fail(SETOK_ELAUNCH, "setok_cluster_dpc_knn: cannot raise the dynamic LDS limit");
        const int strips = cdiv(N, SROWS);
        if (hipMemsetAsync(vec_ws, 0, ((size_t)S_HDR + (size_t)B * S_IMG) * 4, s) != hipSuccess)             // queue heads + exchange counters
            return setok_fail(SETOK_ELAUNCH, "setok_cluster_dpc_knn: cannot clear the exchange workspace");
        int ex = 0;
        const float mant = frexpf(sqrtC, &ex);
        // persistent grid: never more workgroups than USABLE CUs (147 KiB of LDS each: one per CU, all co-resident).  The attribute is the device's
        // (a CPX partition reports its own 32); a CU mask on the stream or the process (hipExtStreamCreateWithCUMask, ROC_GLOBAL_CU_MASK) cuts it down.
        int ncu = 256, dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        // The CU mask is a host-side property of the stream: queried under capture too (round 5, ADVICE r04: a graph captured on — or for — a
        // CU-masked stream was sized from the device's CU count and could launch more spin-waiting workgroups than can be co-resident).  Should the
        // runtime refuse the query while capturing, the smallest mask an earlier eager call saw on this device stands in (GraphedEncode's warm-up
        // call runs on the capture's stream just before the capture).
        static std::atomic<int> seen_masked[64];                          // per device: 0 = nothing seen
        {
            uint32_t mask[32] = {0};
            int bits = 0;
            if (hipExtStreamGetCUMask(s, 32, mask) == hipSuccess) {
                for (int i = 0; i < 32; ++i) bits += __builtin_popcount(mask[i]);
                if (bits > 0 && bits < ncu) {
                    ncu = bits;
                    if (dev >= 0 && dev < 64) { int prev = seen_masked[dev].load(); while ((prev == 0 || bits < prev) && !seen_masked[dev].compare_exchange_weak(prev, bits)) {} }
                }
            } else {
                (void)hipGetLastError();
                hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
                const bool capturing = hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
                const int prev = (dev >= 0 && dev < 64) ? seen_masked[dev].load() : 0;
                if (capturing && prev > 0 && prev < ncu) ncu = prev;
            }
        }
        if (const char* e = getenv("SETOK_STRIP_GRID")) {                 // tests / small-device rehearsal: cap the grid (read per call)
            const int cap = atoi(e);
            if (cap > 0 && cap < ncu) ncu = cap;
        }
        const int grid = B * strips < ncu ? B * strips : ncu;
        if (grid < strips)                                                // fewer co-resident workgroups than one image has strips: its exchanges could never complete
            return setok_fail(SETOK_ELAUNCH, "setok_cluster_dpc_knn: %d usable workgroups cannot hold the %d strips of an image (N=%d)", grid, strips, N);
        int nq = strips > 1 ? (grid - 1) / (strips - 1) : 8;             // grid > nq (strips - 1): see the deadlock-freedom note above the kernel
        nq = nq > 8 ? 8 : nq;
        SArgs a{(const bf16*)x, noise, token_mask, idx_cluster, score, index_down, counts, vec_ws, B, N, C, k, min_cluster_num, strips, threshold, sqrtC,
                1.0f / sqrtC, mant == 0.5f ? 1 : 0, nq};
        if (token_mask) dpc_strip_kernel<true><<<grid, 512, S_LDS, s>>>(a); else dpc_strip_kernel<false><<<grid, 512, S_LDS, s>>>(a);
        SETOK_CHECK_LAUNCH("setok_cluster_dpc_knn(strips)");
        return SETOK_OK;
    }
    SETOK_CHECK_ARG(dist_ws && vec_ws, "setok_cluster_dpc_knn: this shape needs the distance workspace (setok_cluster_workspace)");
    if (dtype == SETOK_BF16 && C % GR_K == 0) {
        // Gram matrices + diagonals in one launch (bf16 throughput mode)
        static SetokDeviceOnce once;
        if (!once.run([] { return hipFuncSetAttribute((const void*)dpc_gram_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GR_STAGE) == hipSuccess; }))
            return setok_fail(SETOK_ELAUNCH, "setok_cluster_dpc_knn: cannot raise the dynamic LDS limit");
        const int strips = cdiv(N, GR_ROWS), chunks = cdiv(N, GR_COLS);
        dpc_gram_bf16_kernel<<<cdiv(B, 8) * 8 * strips * chunks, 256, 2 * GR_STAGE, s>>>((const bf16*)x, dist_ws, vec_ws, B, N, C, strips, chunks);
    } else {
        // fp32 parity mode: batched exact-f32 MFMA GEMM with A = W = X_b, then |x_i|^2 = G_ii
        int rc = setok_linear(stream, dtype, SETOK_F32, x, C, x, nullptr, nullptr, dist_ws, N, N, N, C, SETOK_ACT_NONE, B,
                              (int64_t)N * C, (int64_t)N * C, (int64_t)N * N);
        if (rc != SETOK_OK) return rc;
        dpc_diag_kernel<<<cdiv(rows, 256), 256, 0, s>>>(dist_ws, vec_ws, B, N);
    }
    const int pl = N <= 256 ? 4 : (N <= 576 ? 9 : MAXPL);             // register slots per lane holding one distance row
    const dim3 rg(cdiv(rows, 4));
    if (!token_mask) {
        if (pl == 4) dpc_dist_rows_kernel<true, 4><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, B, N, k, sqrtC);
        else if (pl == 9) dpc_dist_rows_kernel<true, 9><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, B, N, k, sqrtC);
        else dpc_dist_rows_kernel<true, MAXPL><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, B, N, k, sqrtC);
    } else {
        if (pl == 4) dpc_dist_rows_kernel<false, 4><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, B, N, k, sqrtC);
        else if (pl == 9) dpc_dist_rows_kernel<false, 9><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, B, N, k, sqrtC);
        else dpc_dist_rows_kernel<false, MAXPL><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, B, N, k, sqrtC);
        float* gmax = reinterpret_cast<float*>(counts);                 // B floats of scratch until dpc_select_kernel overwrites it
        dpc_gmax_kernel<<<B, 64, 0, s>>>(vec_ws, gmax, N);
        if (pl == 4) dpc_masked_density_kernel<4><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, token_mask, gmax, B, N, k);
        else if (pl == 9) dpc_masked_density_kernel<9><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, token_mask, gmax, B, N, k);
        else dpc_masked_density_kernel<MAXPL><<<rg, 256, 0, s>>>(dist_ws, vec_ws, noise, token_mask, gmax, B, N, k);
    }
    dpc_score_kernel<<<cdiv(rows, 4), 256, 0, s>>>(dist_ws, vec_ws, score, B, N);
    dpc_select_kernel<<<B, 256, 0, s>>>(score, threshold, min_cluster_num, N, index_down, counts, vec_ws);
    dpc_assign_kernel<<<dim3(cdiv(N, 256), B), 256, 0, s>>>(dist_ws, vec_ws, counts, idx_cluster, N);
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
    // the entries after the last cluster's: every one = B * N, i.e. the segments [total, B * N) are EMPTY — a consumer that does not know the
    // number of clusters (setok_encode: no host read of the counts) launches for B * N segments and the surplus ones exit at once
    if (b == B - 1) for (int i = base + L + tid; i <= B * N; i += 256) seg_offsets[i] = B * N;
}

extern "C" int setok_cluster_sort(void* stream, const int64_t* idx_cluster, const int32_t* counts, int B, int N,
                                  int32_t* perm, int32_t* seg_offsets, int32_t* img_offsets) {
    SETOK_CHECK_ARG(idx_cluster && counts && perm && seg_offsets && img_offsets, "setok_cluster_sort: null operand");
    SETOK_CHECK_ARG(B > 0 && N > 0 && N <= 1024, "setok_cluster_sort: need 0 < N <= 1024");
    hipStream_t s = (hipStream_t)stream;
    prefix_counts_kernel<<<1, 64, 0, s>>>(counts, img_offsets, B);
    sort_kernel<<<B, 256, 0, s>>>(idx_cluster, counts, img_offsets, B, N, perm, seg_offsets);
    setok_prof_rows_changed();                                 // img_offsets[B] is the device-side row count of the ragged stages behind this call
    SETOK_CHECK_LAUNCH("setok_cluster_sort");
    return SETOK_OK;
}
