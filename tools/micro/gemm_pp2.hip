// gemm_pp2.hip — EXPERIMENT (round 3), PERSISTENT form of gemm_pp.hip (one workgroup per CU walks tiles; the quarter-tile request stream runs across
// tile boundaries; epilogue through 32 KiB of LDS of its own with 16-byte row stores).  gemm_pp.hip — EXPERIMENT (round 3): the CDNA guide's 8-phase "ping-pong" main loop with M-split quarter tiles, as a stand-alone plain GEMM
// C[M,N] = A[M,K] W[N,K]^T (bf16 in, bf16 out), to be compared on one box with the library's persistent kernel (tools/bench_gemm_steady.py).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/gemm_pp.hip -o /tmp/gemm_pp && /tmp/gemm_pp
//
// Same LDS image, fragment layout and MFMA shape as csrc/gemm_persist.hip (256 x 256 x 64 tiles, 8 waves = 2 (M) x 4 (N), wave tile 128 x 64,
// v_mfma_f32_16x16x32_bf16, 16-byte-slot XOR swizzle, operands by global_load_lds_dwordx4 from inline asm).  What differs is the schedule:
//   * a K-tile is four PHASES per wave — the quadrants (row half i, column half j) of its 128 x 64 output in the order (0,0) (0,1) (1,1) (1,0) —
//     each phase = a LOAD slot (the fragments of the quadrant for BOTH k-steps: 4 of W and / or 8 of A ds_read_b128, + 2 LDS-DMA requests)
//     and an MFMA slot (16 MFMAs), every slot closed by s_barrier;
//   * the two wave rows run ONE SLOT APART (waves 4-7 execute one extra barrier first): on every SIMD one wave multiplies while its partner reads
//     fragments and stands at the address unit — the matrix pipe never has two claimants and never none;
//   * operands arrive in QUARTER tiles (A rows 0-127 / 128-255, W rows 0-127 / 128-255 of a stage: 16 KiB = 2 requests per lane, 128-byte rows),
//     one quarter per phase, issued up to 1.5 K-tiles ahead: a quarter of stage s is refilled (for K-tile t + 2) as soon as ITS last reader of
//     K-tile t is through — A quarters after phase 2, W quarters after phase 3 — so three quarters are always in flight and the only wait of a
//     K-tile is a counted vmcnt(2) one slot before the first read.
// Issue order per K-tile t (stage s = t & 1):  phase 0: A_hi(t+1) -> s^1   phase 1: W_lo(t+1) -> s^1   phase 2: W_hi(t+1) -> s^1   phase 3: A_lo(t+2) -> s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <chrono>
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int TM = 256, TK = 64, STAGE = 64 * 1024, BOFF = 32 * 1024, QT = 16 * 1024;
__device__ inline int swz(int row) { return (row >> 1) & 7; }

struct Args { const bf16* A; const bf16* W; bf16* C; int M, N, K, tilesM, tilesN; unsigned long long* tim; };


template <int GRP>
__device__ __forceinline__ void body(const Args& g, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 3;                                       // wr == GRP
    const int l15 = lane & 15, g4 = lane >> 4;
    const int nk = g.K / TK;
    const int G = gridDim.x;
    const int num_tiles = g.tilesM * g.tilesN;
    auto tile_of = [&](int round, int& m0, int& n0) -> bool {
        int L;
        if ((G & 7) == 0) L = round * G + (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
        else L = round * G + blockIdx.x;
        if (L >= num_tiles) return false;
        constexpr int GM = 8;
        const int per = GM * g.tilesN, group = L / per, first_m = group * GM;
        const int gm = min(g.tilesM - first_m, GM), in = L - group * per;
        m0 = (first_m + in % gm) * TM;
        n0 = (in / gm) * 256;
        return true;
    };
    const int prow = tid >> 3;
    const unsigned kc16 = (unsigned)(((tid & 7) ^ swz(prow)) << 4);
    const unsigned a_off = (unsigned)prow * (unsigned)(g.K * 2) + kc16;
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
    int m0, n0, nm0 = 0, nn0 = 0, round = 0;
    if (!tile_of(0, m0, n0)) return;
    bool has_next = false;
    const char *a_cur, *w_cur, *a_nxt = nullptr, *w_nxt = nullptr;
    a_cur = reinterpret_cast<const char*>(g.A + (size_t)m0 * g.K);
    w_cur = reinterpret_cast<const char*>(g.W + (size_t)n0 * g.K);
    int cnt = 0;                                                   // position in the K-tile stream: stage = (cnt + u) & 1 for the tile's K-tile u
    // quarter q (0 = A_lo, 1 = A_hi, 2 = W_lo, 3 = W_hi) of the CURRENT tile's K-tile u; u >= nk runs into the next tile's K-tile u - nk
    auto issue_quarter = [&](int q, int u) {
        const char* at = a_cur; const char* wt = w_cur; int kk = u;
        if (u >= nk) { if (!has_next) return; at = a_nxt; wt = w_nxt; kk = u - nk; }
        const unsigned sb = lds0 + ((cnt + u) & 1) * STAGE + (q >> 1) * BOFF + (q & 1) * QT;
        const char* base = ((q >> 1) ? wt : at) + (size_t)(q & 1) * 128 * (size_t)(g.K * 2) + (size_t)kk * 128;
        dma16(base, a_off, sb);
        dma16(base + (size_t)64 * (size_t)(g.K * 2), a_off, sb + 8192);
    };

    f32x4 acc[8][4];
    bf16x8 A0[2][4], A1[2][4], W0[2][2], W1[2][2];                  // [k-step][tile]
    auto rd_a = [&](const char* T, int ih, int ks, bf16x8 (&dst)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = GRP * 128 + ih * 64 + t * 16 + l15;
            dst[t] = *reinterpret_cast<const bf16x8*>(T + r * 128 + (((ks * 4 + g4) ^ swz(r)) << 4));
        }
    };
    auto rd_w = [&](const char* T, int jh, bf16x8 (&dst)[2][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wc * 64 + jh * 32 + j * 16 + l15;
                dst[ks][j] = *reinterpret_cast<const bf16x8*>(T + BOFF + r * 128 + (((ks * 4 + g4) ^ swz(r)) << 4));
            }
    };
    auto mma8 = [&](int ih, int jh, const bf16x8 (&a)[4], const bf16x8 (&w)[2]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[ih * 4 + t][jh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], a[t], acc[ih * 4 + t][jh * 2 + j], 0, 0, 0);
    };
    auto lgk0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); };
    auto bar = [&]() { asm volatile("s_barrier" ::: "memory"); };

    // ---- prologue of the stream: K-tile 0 entirely + the W quarters of K-tile 1 ---------------------------------------------------------------
    has_next = tile_of(1, nm0, nn0);
    if (has_next) { a_nxt = reinterpret_cast<const char*>(g.A + (size_t)nm0 * g.K); w_nxt = reinterpret_cast<const char*>(g.W + (size_t)nn0 * g.K); }
    issue_quarter(0, 0); issue_quarter(1, 0); issue_quarter(2, 0); issue_quarter(3, 0); issue_quarter(2, 1); issue_quarter(3, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    bar();
    if (GRP == 1) bar();                                            // the second wave row runs one slot behind
    bool a1_ahead = false;                                          // the A quarters of this tile's K-tile 1 were requested in the previous tile's epilogue
    constexpr int NSTORE = 16;

    for (;;) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][j][e] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            const char* T = smem + ((cnt + kt) & 1) * STAGE;
            const bool skipA = a1_ahead && kt == 0;                 // (uniform)
            // phase 0
            if (!skipA) issue_quarter(0, kt + 1);
            rd_w(T, 0, W0); rd_a(T, 0, 0, A0[0]);
            bar();
            rd_a(T, 0, 1, A0[1]);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mma8(0, 0, A0[0], W0[0]);
            lgk0();
            mma8(0, 0, A0[1], W0[1]);
            __builtin_amdgcn_s_setprio(0);
            bar();
            // phase 1
            if (!skipA) issue_quarter(1, kt + 1);
            rd_w(T, 1, W1); rd_a(T, 1, 0, A1[0]);
            lgk0();
            bar();
            __builtin_amdgcn_s_setprio(1);
            mma8(0, 1, A0[0], W1[0]);
            mma8(0, 1, A0[1], W1[1]);
            __builtin_amdgcn_s_setprio(0);
            bar();
            // phase 2
            issue_quarter(2, kt + 2);
            rd_a(T, 1, 1, A1[1]);
            bar();
            lgk0();
            __builtin_amdgcn_s_setprio(1);
            mma8(1, 1, A1[0], W1[0]);
            mma8(1, 1, A1[1], W1[1]);
            __builtin_amdgcn_s_setprio(0);
            bar();
            // phase 3 (+ the one wait of the K-tile, one slot before the next K-tile's first reader)
            issue_quarter(3, kt + 2);
            const bool issued = kt + 2 < nk || has_next;           // the two W quarters of this K-tile's phases 2 / 3 exist
            auto wait_next = [&]() {
                if (!issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (skipA) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");      // [A(1) quarters][16 stores][W(2) quarters]: the stores may still fly
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            };
            if (GRP == 1) wait_next();
            bar();
            __builtin_amdgcn_s_setprio(1);
            mma8(1, 0, A1[0], W0[0]);
            mma8(1, 0, A1[1], W0[1]);
            __builtin_amdgcn_s_setprio(0);
            if (GRP == 0) wait_next();
            bar();
        }
        // ---- tile boundary: the A quarters of the next tile's K-tile 1 go out BEFORE this tile's stores (their stage — the one of the K-tile just
        //      finished — is free), so that no operand request of the next K-tiles ever queues behind a store ---------------------------------------
        cnt += nk;
        if (has_next) {
            const char* ac = a_cur; const char* wc_ = w_cur;
            a_cur = a_nxt; w_cur = w_nxt;                           // issue_quarter(q, 1) now addresses the next tile's K-tile 1 -> stage (cnt + 1) & 1
            issue_quarter(0, 1); issue_quarter(1, 1);
            a_cur = ac; w_cur = wc_;
        }
        // epilogue: one 32-row MFMA pass at a time through this wave's private 4 KiB of staging, 16-byte row stores
        {
            char* stg = smem + 2 * STAGE + wave * 4096;
            const int slot = lane & 7, lrow = lane >> 3;
            bf16* cw = g.C + (size_t)(m0 + GRP * 128) * g.N + (n0 + wc * 64);
#pragma unroll
            for (int h = 0; h < 4; ++h) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bf16x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (bf16)acc[2 * h + tt][j][e];
                        const int srow = tt * 16 + l15;
                        *reinterpret_cast<bf16x4*>(stg + srow * 128 + (((j * 2 + (g4 >> 1)) ^ (srow & 7)) << 4) + 8 * (g4 & 1)) = v;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bf16x8 ov[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + lrow;
                    ov[it] = *reinterpret_cast<const bf16x8*>(stg + row * 128 + ((slot ^ (row & 7)) << 4));
                }
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    *reinterpret_cast<bf16x8*>(cw + (size_t)(h * 32 + it * 8 + lrow) * g.N + slot * 8) = ov[it];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the staging rows are rewritten by the next pass
            }
        }
        if (!has_next) break;
        a1_ahead = true;
        m0 = nm0; n0 = nn0; a_cur = a_nxt; w_cur = w_nxt; ++round;
        has_next = tile_of(round + 1, nm0, nn0);
        if (has_next) { a_nxt = reinterpret_cast<const char*>(g.A + (size_t)nm0 * g.K); w_nxt = reinterpret_cast<const char*>(g.W + (size_t)nn0 * g.K); }
    }
    if (GRP == 0) bar();                                            // matches the second row's last barrier
}

__global__ __launch_bounds__(512) void gemm_pp_kernel(Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) body<0>(g, smem); else body<1>(g, smem);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void run(int M, int N, int K, double seconds) {
    std::vector<bf16> hA((size_t)M * K), hW((size_t)N * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    const bool zeros = getenv("GEMM_PP_ZEROS") != nullptr;        // all-zero operands: the clock is not power-managed down, cycles show
    for (auto& v : hA) v = (bf16)(zeros ? 0.f : rnd());
    const float ws = 1.0f / sqrtf((float)K);
    for (auto& v : hW) v = (bf16)(zeros ? 0.f : rnd() * ws * 1.7f);
    bf16 *dA, *dW, *dC;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    unsigned long long* dT = nullptr;
    CK(hipMalloc(&dT, 32 * 8)); CK(hipMemset(dT, 0, 32 * 8));
    Args g{dA, dW, dC, M, N, K, (M + 255) / 256, (N + 255) / 256, dT};
    CK(hipFuncSetAttribute((const void*)gemm_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE + 32768));
    const int grid = g.tilesM * g.tilesN < 256 ? g.tilesM * g.tilesN : 256;
    gemm_pp_kernel<<<grid, 512, 2 * STAGE + 32768>>>(g);
    CK(hipDeviceSynchronize());
    // correctness on a sample of outputs (fp64 reference from the bf16 operands)
    std::vector<bf16> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int q = 0; q < 4000; ++q) {
        const int r = (int)(((unsigned long long)q * 2654435761ull) % (unsigned)M), c = (int)(((unsigned long long)q * 40503ull + 17) % (unsigned)N);
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)(float)hA[(size_t)r * K + k] * (double)(float)hW[(size_t)c * K + k];
        const double err = fabs((double)(float)hC[(size_t)r * N + c] - ref) / (fabs(ref) + 0.05);
        if (err > worst) worst = err;
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double ms = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) gemm_pp_kernel<<<grid, 512, 2 * STAGE + 32768>>>(g);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms = t / 20;
    }
    printf("gemm_pp2 M=%d N=%d K=%d: %8.1f us  %7.1f TF   worst sampled rel err %.2e %s\n", M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, worst,
           worst < 2e-2 ? "OK" : "WRONG");
    fflush(stdout);
    CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dC));
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    run(512, 512, 512, 0.2);
    run(8192, 8192, 8192, secs);
    run(4096, 4096, 4096, secs);
    run(65792, 3072, 1024, secs);
    run(65792, 4096, 1024, secs);
    run(65792, 1024, 4096, secs);
    return 0;
}
