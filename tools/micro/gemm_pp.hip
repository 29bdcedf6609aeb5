// gemm_pp.hip — EXPERIMENT (round 3): the CDNA guide's 8-phase "ping-pong" main loop with M-split quarter tiles, as a stand-alone plain GEMM
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
__device__ __forceinline__ void body(const Args& g, char* smem, int m0, int n0) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 3;                                       // wr == GRP
    const int l15 = lane & 15, g4 = lane >> 4;
    const int nk = g.K / TK;
    // ---- LDS-DMA addressing: one lane offset per operand, quarter / piece advance in the wave-uniform base ---------------------------------------
    const int prow = tid >> 3;
    const unsigned kc16 = (unsigned)(((tid & 7) ^ swz(prow)) << 4);
    const unsigned a_off = (unsigned)prow * (unsigned)(g.K * 2) + kc16;      // lda = K, ldw = K (dense)
    const char* a_tile = reinterpret_cast<const char*>(g.A + (size_t)m0 * g.K);
    const char* w_tile = reinterpret_cast<const char*>(g.W + (size_t)n0 * g.K);
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
    // quarter q: 0 = A_lo, 1 = A_hi, 2 = W_lo, 3 = W_hi of K-tile kt -> stage kt & 1
    auto issue_quarter = [&](int q, int kt) {
        if (kt >= nk) return;                                       // (uniform)
        const unsigned sb = lds0 + (kt & 1) * STAGE + (q >> 1) * BOFF + (q & 1) * QT;
        const char* base = ((q >> 1) ? w_tile : a_tile) + (size_t)(q & 1) * 128 * (size_t)(g.K * 2) + (size_t)kt * 128;
        dma16(base, a_off, sb);
        dma16(base + (size_t)64 * (size_t)(g.K * 2), a_off, sb + 8192);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][j][e] = 0.f;

    // fragments: both row halves and both column halves have registers of their own (96), so that the reads spread evenly over the phases:
    //   load slot 0: W0 (4) + A0[k-step 0] (4)      MFMA slot 0 starts with the 4 reads of A0[k-step 1] (they land under its first 8 MFMAs)
    //   load slot 1: W1 (4) + A1[k-step 0] (4)      load slot 2: A1[k-step 1] (4)      load slot 3: none
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
    auto mma8 = [&](int ih, int jh, const bf16x8 (&a)[4], const bf16x8 (&w)[2]) {      // one k-step of a quadrant
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[ih * 4 + t][jh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], a[t], acc[ih * 4 + t][jh * 2 + j], 0, 0, 0);
    };
    auto lgk0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); };
#ifdef PP_TIMING
    unsigned long long tw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tb[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // per slot of the K-tile: cycles of work / at the barrier
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    int slot_id = 0;
    auto bar = [&]() {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_barrier" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        tw[slot_id & 7] += t1 - tlast; tb[slot_id & 7] += t2 - t1; tlast = t2; ++slot_id;
    };
#else
    auto bar = [&]() { asm volatile("s_barrier" ::: "memory"); };
#endif

    // ---- prologue: K-tiles 0 and 1 entirely; the first read needs K-tile 0 -----------------------------------------------------------------------
    // Requests per K-tile t from here on, all for K-tile t + 2 into the CURRENT stage s = t & 1, each pair in a light load slot:
    //   phase 2: W_lo, W_hi(t+2)   (this stage's W quarters are free once phase 1's reads have retired; W0 stays in registers for phase 3)
    //   phase 3: A_lo, A_hi(t+2)   (its A quarters once phase 2's have)
    // so every quarter has a whole K-tile (>= 9 slots, ~1.2 us: the LDS-DMA round trip under load is ~1 us) to land, the heavy load slots
    // (phases 0 and 1: 8 fragment reads each) carry no request at all, and the one wait of a K-tile is a counted vmcnt(8).
#ifdef PP_CLUSTER
    issue_quarter(0, 0); issue_quarter(1, 0); issue_quarter(2, 0); issue_quarter(3, 0);
    issue_quarter(2, 1); issue_quarter(3, 1); issue_quarter(0, 1); issue_quarter(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    // SPREAD (kept): one quarter per phase — phase 0: A_lo(t+1) -> s^1, phase 1: A_hi(t+1) -> s^1, phase 2: W_lo(t+2) -> s, phase 3: W_hi(t+2) -> s;
    // the wait is vmcnt(4).  (CLUSTER — both W quarters in phase 2, both A quarters in phase 3, a whole K-tile of flight, vmcnt(8) — measured
    // 20 % SLOWER on all-zero operands: four requests in one slot hold the wave ~100 cycles each, and the K-tile period is bound by the per-CU
    // LDS-DMA rate (~57 GB/s), not by the request latency.)
    issue_quarter(0, 0); issue_quarter(1, 0); issue_quarter(2, 0); issue_quarter(3, 0); issue_quarter(2, 1); issue_quarter(3, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    bar();
    if (GRP == 1) bar();                                            // the second wave row runs one slot behind
#ifdef PP_TIMING
    slot_id = 0; tlast = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 8; ++i) { tw[i] = 0; tb[i] = 0; }
#endif

    for (int kt = 0; kt < nk; ++kt) {
        const char* T = smem + (kt & 1) * STAGE;
        // phase 0
#ifndef PP_CLUSTER
        issue_quarter(0, kt + 1);
#endif
        rd_w(T, 0, W0); rd_a(T, 0, 0, A0[0]);
        bar();
        rd_a(T, 0, 1, A0[1]);                                       // lands under the first 8 MFMAs
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");          // everything but those 4
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        mma8(0, 0, A0[0], W0[0]);
        lgk0();
        mma8(0, 0, A0[1], W0[1]);
        __builtin_amdgcn_s_setprio(0);
        bar();
        // phase 1
#ifndef PP_CLUSTER
        issue_quarter(1, kt + 1);
#endif
        rd_w(T, 1, W1); rd_a(T, 1, 0, A1[0]);
        lgk0();                                                     // the last readers of this stage's W quarters retire BEFORE the barrier: phase 2 refills them
        bar();
        __builtin_amdgcn_s_setprio(1);
        mma8(0, 1, A0[0], W1[0]);
        mma8(0, 1, A0[1], W1[1]);
        __builtin_amdgcn_s_setprio(0);
        bar();
        // phase 2
#ifdef PP_CLUSTER
        issue_quarter(2, kt + 2); issue_quarter(3, kt + 2);
#else
        issue_quarter(2, kt + 2);
#endif
        rd_a(T, 1, 1, A1[1]);
        lgk0();                                                     // likewise the A quarters: phase 3 refills them
        bar();
        __builtin_amdgcn_s_setprio(1);
        mma8(1, 1, A1[0], W1[0]);
        mma8(1, 1, A1[1], W1[1]);
        __builtin_amdgcn_s_setprio(0);
        bar();
        // phase 3: the wait for K-tile kt + 1 sits one slot before its first reader (wave row 0 reads at the next slot boundary):
        //          row 0 waits at the end of its MFMA slot, row 1 at the end of its LOAD slot — the same barrier for both
#ifdef PP_CLUSTER
        issue_quarter(0, kt + 2); issue_quarter(1, kt + 2);
#define PP_WAIT "s_waitcnt vmcnt(8)"
#else
        issue_quarter(3, kt + 2);
#define PP_WAIT "s_waitcnt vmcnt(4)"
#endif
        if (GRP == 1) { if (kt + 2 < nk) asm volatile(PP_WAIT ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        bar();
        __builtin_amdgcn_s_setprio(1);
        mma8(1, 0, A1[0], W0[0]);
        mma8(1, 0, A1[1], W0[1]);
        __builtin_amdgcn_s_setprio(0);
        if (GRP == 0) { if (kt + 2 < nk) asm volatile(PP_WAIT ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        bar();
    }
#ifdef PP_TIMING
    if (g.tim && blockIdx.x == 0 && (threadIdx.x & 255) == 0)
        for (int i = 0; i < 8; ++i) { g.tim[GRP * 16 + i] = tw[i]; g.tim[GRP * 16 + 8 + i] = tb[i]; }
#endif
    if (GRP == 0) bar();                                            // matches the second row's last barrier

    // ---- epilogue (plain): a lane holds row l15 of each 16-row tile, 4 consecutive columns per accumulator -> 8-byte stores ------------------
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int row = m0 + GRP * 128 + t * 16 + l15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wc * 64 + j * 16 + 4 * g4;
            if (row < g.M && col < g.N) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (bf16)acc[t][j][e];
                *reinterpret_cast<bf16x4*>(g.C + (size_t)row * g.N + col) = v;
            }
        }
    }
}

__global__ __launch_bounds__(512) void gemm_pp_kernel(Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware tile order: 32 consecutive ids per XCD = an 8 (M) x 4 (N) patch
    const int G = gridDim.x;
    int L = blockIdx.x;
    if ((G & 7) == 0) L = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    constexpr int GM = 8;
    const int per = GM * g.tilesN, group = L / per, first_m = group * GM;
    const int gm = min(g.tilesM - first_m, GM), in = L - group * per;
    const int m0 = (first_m + in % gm) * TM, n0 = (in / gm) * 256;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) body<0>(g, smem, m0, n0); else body<1>(g, smem, m0, n0);
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
    CK(hipFuncSetAttribute((const void*)gemm_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
    const int grid = g.tilesM * g.tilesN;
    gemm_pp_kernel<<<grid, 512, 2 * STAGE>>>(g);
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
        for (int i = 0; i < 20; ++i) gemm_pp_kernel<<<grid, 512, 2 * STAGE>>>(g);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms = t / 20;
    }
#ifdef PP_TIMING
    {
        unsigned long long h[32];
        CK(hipMemcpy(h, dT, sizeof(h), hipMemcpyDeviceToHost));
        const double nkt = K / 64.0;
        static const char* names[8] = {"load0", "mma0", "load1", "mma1", "load2", "mma2", "load3", "mma3"};
        for (int gq = 0; gq < 2; ++gq) {
            printf("  row %d cycles per K-tile (work + barrier wait):", gq);
            double tot = 0;
            for (int i = 0; i < 8; ++i) { printf(" %s %.0f+%.0f", names[i], h[gq * 16 + i] / nkt, h[gq * 16 + 8 + i] / nkt); tot += (h[gq * 16 + i] + h[gq * 16 + 8 + i]) / nkt; }
            printf("  = %.0f\n", tot);
        }
    }
#endif
    printf("gemm_pp M=%d N=%d K=%d: %8.1f us  %7.1f TF   worst sampled rel err %.2e %s\n", M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, worst,
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
