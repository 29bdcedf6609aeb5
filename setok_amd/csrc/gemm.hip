// gemm.hip — C[M,N] = act(A[M,K] · W[N,K]^T + bias) + residual   on gfx950 matrix cores.
//
// Replaces every nn.Linear on the reference's path (see include/setok_hip.h: setok_linear).
// Both operands are K-contiguous ("B^T input"), which is what MFMA fragments want: lane l of a
// 32x32x16 bf16 MFMA holds 8 consecutive k of row (l & 31), k-group (l >> 5).
//
//  * bf16 kernel  (throughput mode): 128x128x64 block tile, 4 waves (2x2), each wave a 64x64 sub-tile
//    = 2x2 MFMA tiles of v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  Global -> registers -> LDS
//    staging, LDS double-buffered, one barrier per K tile.  LDS rows are 128 B with a 16-B-slot XOR
//    swizzle (slot ^= row & 7) so a ds_read_b128 lane group is <= 2-way conflicted
//    (cdna_hip_programming.md T2).
//  * the big bf16 Linear layers (>= 48 output tiles of 256x256) go to the persistent direct-to-LDS kernel
//    in gemm_persist.hip, every smaller bf16 -> bf16 problem with N % 64 == 0 to that file's 64x64 eight-stage
//    LDS-DMA kernel (a handful of images is latency-bound: 257 x 3072 x 1024 takes 9 us there against 36 us
//    here; one-image encode 9.0 -> 3.8 ms); the kernel below serves fp32 outputs (Gram matrix), batches and
//    odd N.  The persistent and the 64x64 kernel give the same bits for the same row (bias as accumulator init, ascending k,
//    v_mfma_f32_16x16x32_bf16): a sample alone equals the sample inside a batch.  The kernel below keeps the 32x32x16 shape; for a
//    given layer it is either always or never the one that runs (the choice depends on N / output type / batching, not on M).
//  * fp32 kernel  (parity mode): 64x64x16 block tile, 4 waves, v_mfma_f32_32x32x2_f32 — bit-for-bit a
//    k-ordered fmaf chain, so results do not depend on tile geometry.
#include "common.h"
#include <stdlib.h>

struct GemmArgs {
    const void* A; const void* W; const float* bias; const void* res; void* C;
    int64_t lda, ldc, sA, sW, sC;
    int M, N, K, act;
    const int32_t* m_dev;        // optional DEVICE-side row count (<= M): rows beyond it are neither read nor written (setok_encode's ragged stages)
};
__device__ inline int rows_of(const GemmArgs& g) { return g.m_dev ? min(g.M, *g.m_dev) : g.M; }

template <typename TO> __device__ inline float ld_out(const TO* p) { return Elem<TO>::ld(p); }
template <typename TO> __device__ inline float round_to(float v) { return (float)(TO)v; }

// --------------------------------------------------------------------------------------------
// bf16 128x128x64
// --------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 64;

__device__ inline int lds_off_bf16(int row, int chunk) {      // byte offset of a 16-B chunk (8 bf16)
    return row * (BK * 2) + ((chunk ^ (row & 7)) << 4);
}

template <typename TO, bool FAST_ACT>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * BK * 2];   // 64 KiB
    constexpr int STAGE = (BM + BN) * BK * 2, BOFF = BM * BK * 2;       // per-buffer bytes; B tile offset

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int Mrt = rows_of(g);
    if (m0 >= Mrt) return;                                               // (whole workgroup: before any barrier)
    const bf16* A = (const bf16*)g.A + (int64_t)blockIdx.z * g.sA;
    const bf16* W = (const bf16*)g.W + (int64_t)blockIdx.z * g.sW;
    TO* C = (TO*)g.C + (int64_t)blockIdx.z * g.sC;
    const TO* R = g.res ? (const TO*)g.res + (int64_t)blockIdx.z * g.sC : nullptr;

    // staging map: chunk c = tid + 256*p  ->  row = c >> 3, kchunk = c & 7
    const bf16* a_src[4]; const bf16* b_src[4]; int st_off[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = tid + 256 * p, row = c >> 3, kc = c & 7;
        const int ar = min(m0 + row, max(Mrt - 1, 0)), br = min(n0 + row, g.N - 1);
        a_src[p] = A + (int64_t)ar * g.lda + kc * 8;
        b_src[p] = W + (int64_t)br * g.K + kc * 8;
        st_off[p] = lds_off_bf16(row, kc);
    }

    // The bias is the accumulators' initial value and the residual is added to the ROUNDED Linear output — exactly
    // the arithmetic of the persistent kernel (gemm_persist.hip), so a row's result does not depend on which of the
    // two kernels (i.e. on how many rows the batch has) computed it.
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            const float bv = (g.bias && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
        }

    uint4 ra[4], rb[4];
    auto load_global = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ra[p] = *reinterpret_cast<const uint4*>(a_src[p] + k0);
            rb[p] = *reinterpret_cast<const uint4*>(b_src[p] + k0);
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<uint4*>(smem + buf * STAGE + st_off[p]) = ra[p];
            *reinterpret_cast<uint4*>(smem + buf * STAGE + BOFF + st_off[p]) = rb[p];
        }
    };

    const int nk = g.K / BK;
    load_global(0);
    store_lds(0);
    __syncthreads();

    const int frow = lane & 31, fk = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const char* Ab = smem + buf * STAGE;
        const char* Bb = Ab + BOFF;
        if (kt + 1 < nk) load_global((kt + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ar = wm * 64 + t * 32 + frow, br = wn * 64 + t * 32 + frow;
                af[t] = *reinterpret_cast<const bf16x8*>(Ab + lds_off_bf16(ar, ks * 2 + fk));
                bfr[t] = *reinterpret_cast<const bf16x8*>(Bb + lds_off_bf16(br, ks * 2 + fk));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_lds(buf ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= Mrt) continue;
                float v = acc[i][j][r];
                if (g.act == SETOK_ACT_QUICK_GELU) v = FAST_ACT ? v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.45546696f * v)) : v / (1.0f + expf(-1.702f * v));
                else if (g.act == SETOK_ACT_GELU_ERF) v = FAST_ACT ? gelu_erf_fast(v) : 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                const int64_t o = (int64_t)row * g.ldc + col;
                if (R) v = round_to<TO>(v) + ld_out<TO>(R + o);           // Linear output rounded to TO first (no-op for fp32)
                Elem<TO>::st(C + o, v);
            }
        }
}

int setok_gemm_persist_bf16(hipStream_t s, const bf16* A, int64_t lda, const bf16* W, const float* bias, const bf16* res,
                            bf16* C, int64_t ldc, int M, int N, int K, int act, const float* ln_stats, const float* ln_colsum, const int32_t* m_dev);    // gemm_persist.hip
int setok_gemm_small_bf16(hipStream_t s, const bf16* A, int64_t lda, const bf16* W, const float* bias, const bf16* res,
                          bf16* C, int64_t ldc, int M, int N, int K, int act, const float* ln_stats, const float* ln_colsum, const int32_t* m_dev);      // gemm_persist.hip
int setok_gemm_persist_f32_batched(hipStream_t s, const bf16* A, int64_t lda, const bf16* W, float* C, int64_t ldc, int M, int N, int K, int batch,
                                   int64_t sA, int64_t sW, int64_t sC);               // gemm_persist.hip

// --------------------------------------------------------------------------------------------
// fp32 64x64x16  (exact f32 MFMA; parity mode)
// --------------------------------------------------------------------------------------------
constexpr int FM = 64, FN = 64, FK = 16, FLD = FM + 4;

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ float At[2][FK][FLD];
    __shared__ float Bt[2][FK][FLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * FM, n0 = blockIdx.x * FN;
    const int Mrt = rows_of(g);
    if (m0 >= Mrt) return;
    const float* A = (const float*)g.A + (int64_t)blockIdx.z * g.sA;
    const float* W = (const float*)g.W + (int64_t)blockIdx.z * g.sW;
    float* C = (float*)g.C + (int64_t)blockIdx.z * g.sC;
    const float* R = g.res ? (const float*)g.res + (int64_t)blockIdx.z * g.sC : nullptr;

    const int srow = tid >> 2, skq = tid & 3;
    const float* a_src = A + (int64_t)min(m0 + srow, max(Mrt - 1, 0)) * g.lda + skq * 4;
    const float* b_src = W + (int64_t)min(n0 + srow, g.N - 1) * g.K + skq * 4;

    // Blocked summation: each 64-deep K chunk is an exact fma chain from zero, chunks are then added —
    // the error growth of a blocked CPU sgemm rather than of one K-long serial chain.
    f32x16 acc, tot;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; tot[r] = 0.f; }

    f32x4 ra, rb;
    auto load_global = [&](int k0) {
        ra = *reinterpret_cast<const f32x4*>(a_src + k0);
        rb = *reinterpret_cast<const f32x4*>(b_src + k0);
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            At[buf][skq * 4 + j][srow] = ra[j];
            Bt[buf][skq * 4 + j][srow] = rb[j];
        }
    };
    const int nk = g.K / FK;
    load_global(0);
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_global((kt + 1) * FK);
#pragma unroll
        for (int ks = 0; ks < FK / 2; ++ks) {
            const int k = ks * 2 + (lane >> 5);
            const float a = At[buf][k][wm * 32 + (lane & 31)];
            const float b = Bt[buf][k][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if ((kt & 3) == 3) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { tot[r] += acc[r]; acc[r] = 0.f; }
        }
        if (kt + 1 < nk) store_lds(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += tot[r];
    const int col = n0 + wn * 32 + (lane & 31);
    if (col >= g.N) return;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= Mrt) continue;
        float v = acc[r] + bv;
        if (g.act == SETOK_ACT_QUICK_GELU) v = v / (1.0f + expf(-1.702f * v));
        else if (g.act == SETOK_ACT_GELU_ERF) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        const int64_t o = (int64_t)row * g.ldc + col;
        if (R) v += R[o];
        C[o] = v;
    }
}

// --------------------------------------------------------------------------------------------
static bool g_force_small_tiles = false;   // test hook: SETOK_GEMM_SMALL_TILES=1 keeps every bf16 GEMM on the 128x128 kernel

// setok_linear with an optional DEVICE-side row count `m_dev` (one int32, <= M): the launch is sized for M rows, tiles beyond *m_dev exit at once
// and no row beyond it is read or written — how setok_encode runs its ragged stages (the rows of all images' cluster tokens) without the host
// knowing their number.  The kernel is chosen from M exactly as without m_dev, and all bf16 kernels share their arithmetic per output row, so
// the results do not depend on it.
int setok_linear_dev(void* stream, int dtype, int out_dtype, const void* A, int64_t lda, const void* W,
                     const float* bias, const void* residual, void* C, int64_t ldc, int M, int N, int K,
                     int act, int batch, int64_t strideA, int64_t strideW, int64_t strideC, const int32_t* m_dev) {
    SETOK_CHECK_ARG(A && W && C, "setok_linear: null operand");
    SETOK_CHECK_ARG(M >= 0 && N > 0 && K > 0 && batch >= 1, "setok_linear: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
    SETOK_CHECK_ARG(act >= SETOK_ACT_NONE && act <= SETOK_ACT_GELU_ERF, "setok_linear: bad act %d", act);
    SETOK_CHECK_ARG(lda >= K && ldc >= N, "setok_linear: lda/ldc too small");
    if (M == 0) return SETOK_OK;
    GemmArgs g{A, W, bias, residual, C, lda, ldc, strideA, strideW, strideC, M, N, K, act, m_dev};
    hipStream_t s = (hipStream_t)stream;
    const double es_in = dtype == SETOK_BF16 ? 2.0 : 4.0, es_out = out_dtype == SETOK_BF16 ? 2.0 : 4.0;
    {
        static const bool env_small = [] { const char* e = getenv("SETOK_GEMM_SMALL_TILES"); return e && e[0] == '1'; }();
        g_force_small_tiles = env_small;
    }
    static const int persist_min = [] { const char* e = getenv("SETOK_GEMM_PERSIST_MINTILES"); return e ? atoi(e) : 90; }();   // test hook; round 4: 48 -> 90 (the pair-wise small-tile kernel moved the crossover: profiles/r04_mintiles.log)
    static const int small_max = [] { const char* e = getenv("SETOK_GEMM_SMALL64_MAXTILES"); return e ? atoi(e) : 0x7fffffff; }();
    // the launches of gemm_persist.hip (every bf16 -> bf16 problem with aligned rows, unless a test hook says otherwise) carry the profiler's timestamps themselves
    const bool lds_dma_path = dtype == SETOK_BF16 && out_dtype == SETOK_BF16 && batch == 1 && K % BK == 0 && lda % 8 == 0 && N % 64 == 0 && ldc % 8 == 0 && !g_force_small_tiles &&
                              ((K >= 192 && cdiv(M, 256) * cdiv(N, 256) >= persist_min) || cdiv(M, BM) * cdiv(N, BN) <= small_max);
    SetokProfScope prof(s, dtype == SETOK_BF16 ? SETOK_PROF_GEMM_BF16 : SETOK_PROF_GEMM_F32, act | (residual ? 4 : 0), 2.0 * M * N * K * batch,
                        batch * (((double)M * K + (double)N * K) * es_in + (double)M * N * es_out * (residual ? 2 : 1)),   // A, W (+ residual) read once, C written once
                        lds_dma_path);
    if (m_dev && prof.idx >= 0) setok_prof_rows(s, prof.idx, m_dev, M, batch * (double)N * K * es_in);
    if (dtype == SETOK_BF16) {
        SETOK_CHECK_ARG(K % BK == 0, "setok_linear(bf16): K=%d must be a multiple of %d", K, BK);
        SETOK_CHECK_ARG(lda % 8 == 0, "setok_linear(bf16): lda must be a multiple of 8");
        // big problems (>= 90 tiles of 256x256; measured crossover against the 64x64 kernel: 60 and 80 tiles are faster there, 96 here): persistent direct-to-LDS kernel (gemm_persist.hip)
        if (out_dtype == SETOK_BF16 && batch == 1 && N % 64 == 0 && K >= 192 && ldc % 8 == 0 && cdiv(M, 256) * cdiv(N, 256) >= persist_min && !g_force_small_tiles)
            return setok_gemm_persist_bf16(s, (const bf16*)A, lda, (const bf16*)W, bias, (const bf16*)residual, (bf16*)C, ldc, M, N, K, act, nullptr, nullptr, m_dev);
        // fp32-out batched problems without bias / activation / residual and with enough tiles (weight-gradient partial products)
        if (!m_dev && out_dtype == SETOK_F32 && !bias && !residual && act == SETOK_ACT_NONE && N % 64 == 0 && K >= 192 && ldc % 4 == 0 &&
            strideA % 8 == 0 && strideW % 8 == 0 && strideC % 4 == 0 && cdiv(M, 256) * cdiv(N, 256) * batch >= 96 && !g_force_small_tiles)
            return setok_gemm_persist_f32_batched(s, (const bf16*)A, lda, (const bf16*)W, (float*)C, ldc, M, N, K, batch, strideA, strideW, strideC);
        // every other bf16 -> bf16 problem with N % 64 == 0: 64 x 64 tiles, eight-stage LDS-DMA pipeline (gemm_persist.hip).
        // SETOK_GEMM_SMALL64_MAXTILES=<n> (test hook) limits it to problems of at most n 128 x 128 tiles.
        {
            if (out_dtype == SETOK_BF16 && batch == 1 && N % 64 == 0 && ldc % 8 == 0 && cdiv(M, BM) * cdiv(N, BN) <= small_max && !g_force_small_tiles)
                return setok_gemm_small_bf16(s, (const bf16*)A, lda, (const bf16*)W, bias, (const bf16*)residual, (bf16*)C, ldc, M, N, K, act, nullptr, nullptr, m_dev);
        }
        dim3 grid(cdiv(N, BN), cdiv(M, BM), batch);
        if (out_dtype == SETOK_BF16) gemm_bf16_kernel<bf16, true><<<grid, 256, 0, s>>>(g);
        else if (out_dtype == SETOK_F32) gemm_bf16_kernel<float, false><<<grid, 256, 0, s>>>(g);
        else return setok_fail(SETOK_EINVAL, "setok_linear: bad out_dtype %d", out_dtype);
    } else if (dtype == SETOK_F32) {
        SETOK_CHECK_ARG(out_dtype == SETOK_F32, "setok_linear(f32): out_dtype must be f32");
        SETOK_CHECK_ARG(K % FK == 0, "setok_linear(f32): K=%d must be a multiple of %d", K, FK);
        SETOK_CHECK_ARG(lda % 4 == 0, "setok_linear(f32): lda must be a multiple of 4");
        dim3 grid(cdiv(N, FN), cdiv(M, FM), batch);
        gemm_f32_kernel<<<grid, 256, 0, s>>>(g);
    } else {
        return setok_fail(SETOK_EINVAL, "setok_linear: bad dtype %d", dtype);
    }
    SETOK_CHECK_LAUNCH("setok_linear");
    return SETOK_OK;
}

extern "C" int setok_linear(void* stream, int dtype, int out_dtype, const void* A, int64_t lda, const void* W,
                            const float* bias, const void* residual, void* C, int64_t ldc, int M, int N, int K,
                            int act, int batch, int64_t strideA, int64_t strideW, int64_t strideC) {
    return setok_linear_dev(stream, dtype, out_dtype, A, lda, W, bias, residual, C, ldc, M, N, K, act, batch, strideA, strideW, strideC, nullptr);
}

// LayerNorm folded into the consuming Linear (bf16 throughput mode; gemm_persist.hip explains the algebra).  Same kernel choice as
// setok_linear makes for a bf16 -> bf16 problem, so a row's result does not depend on the batch it is computed in.
extern "C" int setok_linear_ln(void* stream, const void* A, int64_t lda, const void* w_gamma, const float* col_frag, const float* row_stats,
                               void* C, int64_t ldc, int M, int N, int K, int act) {
    SETOK_CHECK_ARG(A && w_gamma && col_frag && row_stats && C, "setok_linear_ln: null operand");
    SETOK_CHECK_ARG(M >= 0 && N > 0 && K > 0, "setok_linear_ln: bad shape M=%d N=%d K=%d", M, N, K);
    SETOK_CHECK_ARG(act >= SETOK_ACT_NONE && act <= SETOK_ACT_GELU_ERF, "setok_linear_ln: bad act %d", act);
    SETOK_CHECK_ARG(K % BK == 0 && N % 64 == 0 && lda >= K && ldc >= N && lda % 8 == 0 && ldc % 8 == 0,
                    "setok_linear_ln: needs K %% 64 == 0, N %% 64 == 0, 16-byte aligned rows (M=%d N=%d K=%d)", M, N, K);
    if (M == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    SetokProfScope prof(s, SETOK_PROF_GEMM_BF16, act | 8, 2.0 * M * N * K, ((double)M * K + (double)N * K) * 2.0 + (double)M * N * 2.0 + (double)M * 32.0, true);
    static const int persist_min = [] { const char* e = getenv("SETOK_GEMM_PERSIST_MINTILES"); return e ? atoi(e) : 90; }();
    if (K >= 192 && cdiv(M, 256) * cdiv(N, 256) >= persist_min)
        return setok_gemm_persist_bf16(s, (const bf16*)A, lda, (const bf16*)w_gamma, nullptr, nullptr, (bf16*)C, ldc, M, N, K, act, row_stats, col_frag, nullptr);
    return setok_gemm_small_bf16(s, (const bf16*)A, lda, (const bf16*)w_gamma, nullptr, nullptr, (bf16*)C, ldc, M, N, K, act, row_stats, col_frag, nullptr);
}
