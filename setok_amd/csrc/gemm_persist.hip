// gemm_persist.hip — persistent bf16 GEMM for the large Linear layers (the dominant kernel of the path).
//
//   C[M,N] = act(A[M,K] · W[N,K]^T + bias) (+ residual, torch-bf16 semantics: the Linear output is rounded
//   to bf16 first, then the residual is added and the sum rounded again)
//
// One workgroup (8 waves, 2 per SIMD) per CU walks output tiles of 256 x (64*NT) columns:
//   * operands go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip).  The LDS image is
//     lane-linear, so the 16-B-slot XOR swizzle (slot ^= row & 7, cdna guide T2) is applied to the
//     per-lane SOURCE address and to the ds_read_b128 address (rule 21).  Two 64 KiB stages.
//   * the K-tiles of consecutive output tiles form ONE continuous stream: while the last K-tile of a
//     tile is multiplied, the first K-tile of the workgroup's NEXT tile is already in flight, so tile
//     boundaries pay neither a prologue latency nor a pipeline drain.  Barriers are raw s_barrier +
//     explicit s_waitcnt (a __syncthreads() would drain the in-flight LDS-DMA, cdna guide §5).
//   * MFMA operands are swapped (A-operand = W fragment, B-operand = activation fragment): a lane then
//     holds, for ONE output row, 4 consecutive columns per accumulator quad.  The epilogue adds the
//     bias and applies the activation in registers, stages bf16 through the LDS stage that has just
//     been consumed (XOR-swizzled 16-B slots), and writes full 128-byte row segments with 16-byte
//     stores; the residual is fetched with coalesced 16-byte loads issued before the staging pass.
//   * tile order: the 32 workgroups resident on one XCD (private 4 MiB L2) cover an 8 (M) x 4 (N)
//     patch of tiles in every round and share operand panels (cdna guide T1).
//   * NT = 4 (256x256 tiles) is the workhorse; NT = 1 (256x64 tiles, 8 waves stacked along M) finishes
//     the <1-round remainder of M so that the big launch runs an exact number of rounds (no tail).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TM = 256, TK = 64;
constexpr int STAGE = 64 * 1024;                 // bytes per pipeline stage (A tile at 0, W tile at BOFF)
constexpr int BOFF = TM * TK * 2;                // 32 KiB
constexpr int LDS_BYTES = 2 * STAGE + 8 * 256;    // + a 256-byte bias row per wave

struct PArgs {
    const bf16* A; const bf16* W; const float* bias; const bf16* res; bf16* C;
    int64_t lda, ldc;
    int M, N, K, tilesM, tilesN, dbg, stagger;
    unsigned long long* tim;     // debug: per-block {main-loop, epilogue, wait-at-first-ktile} cycle sums
};

__device__ inline void s_barrier_lgkm() {         // LDS ops of this wave retired, then the workgroup barrier
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 16-byte-slot swizzle of a 128-byte LDS row: a ds_read_b128 is served in four 16-lane groups
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32), one 256-byte bank row per cycle.  A group reads 16 rows of one
// 32-row MFMA tile at one logical slot; with slot ^= swz(row) its 16 accesses fall on 16 different physical
// (row parity, slot) positions of the bank row -> conflict-free (row & 7 would be 2-way).
__device__ inline int swz(int row) { return ((row >> 1) & 1) | (((row >> 4) & 1) << 1) | (((row >> 3) & 1) << 2); }

template <int N_>
__device__ inline void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

template <int NT, int ACT>
__global__ __launch_bounds__(512) void gemm_persist_kernel(PArgs g) {
    constexpr int WNC = NT, WMC = 8 / NT;         // wave grid (N x M)
    constexpr int WR = TM / WMC, MI = WR / 32;    // rows per wave, 32-row MFMA tiles per wave
    constexpr int HALVES = MI >= 2 ? 2 : 1;       // epilogue passes (staging fits ONE 64 KiB stage)
    constexpr int RP = WR / HALVES;               // rows per wave per pass
    constexpr int TNB = 64 * NT;                  // block tile width

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNC, wn = wave % WNC;
    const int frow = lane & 31, hi = lane >> 5;
    const int nk = g.K / TK;
    const int num_tiles = g.tilesM * g.tilesN;
    const int G = gridDim.x;

    // tile id for (round, block): XCD x (= blockIdx % 8) owns 32 consecutive ids per round
    auto tile_of = [&](int round, int& m0, int& n0) -> bool {
        int L;
        if ((G & 7) == 0) L = round * G + (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
        else L = round * G + blockIdx.x;
        if (L >= num_tiles) return false;
        constexpr int GM = 8;
        const int per = GM * g.tilesN, group = L / per, first_m = group * GM;
        const int gm = min(g.tilesM - first_m, GM), in = L - group * per;
        m0 = (first_m + in % gm) * TM;
        n0 = (in / gm) * TNB;
        return true;
    };

    const bf16* a_src[4]; const bf16* b_src[NT];
    auto set_src = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = i * 512 + tid, row = p >> 3, kc = (p & 7) ^ swz(row);
            a_src[i] = g.A + (int64_t)min(m0 + row, g.M - 1) * g.lda + kc * 8;
            if (i < NT) b_src[i] = g.W + (int64_t)min(n0 + row, g.N - 1) * g.K + kc * 8;
        }
    };
    auto issue_loads = [&](int stage, int k0) {
        char* sb = smem + stage * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + k0),
                                             (__attribute__((address_space(3))) void*)(sb + i * 8192), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NT; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + k0),
                                             (__attribute__((address_space(3))) void*)(sb + BOFF + i * 8192), 16, 0, 0);
    };

    const unsigned scratch_lds = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + 2 * STAGE) + wave * 256);
    auto lds_dma4 = [&](const void* ptr, unsigned lds_dst) {      // 4 bytes per lane -> LDS[lds_dst + 4*lane]
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(ptr), "s"(lds_dst) : "memory");
    };
    // The bias of a tile (this wave's 64 columns) is DMA'd into the wave's 256-byte scratch row one K-tile
    // before the tile starts and read back with ds_read (lgkmcnt domain).  An ordinary global load whose
    // result is first used by the tile's first MFMA makes hipcc emit `s_waitcnt vmcnt(0)` there, which
    // drains the freshly issued operand loads once per tile (seen in the .s).  Inline asm because a 9th
    // LDS-DMA builtin per iteration exceeds what hipcc's waitcnt pass tracks and also forces vmcnt(0);
    // an op the compiler does not count can only make ITS waits stricter (vmcnt retires in order).
    auto bias_issue = [&](int n0_) {
        if (g.bias) lds_dma4(g.bias + min(n0_ + wn * 64 + lane, g.N - 1), scratch_lds);
    };

    // De-synchronise the workgroups: all tiles take the same time, so without this every CU reaches its
    // epilogue at the same instant and the 32 MB of C-tile stores of a round hit the fabric as one burst
    // that no CU can overlap with compute.  A one-off start delay of (hash(block) % 16) * stagger/16 spreads the
    // epilogues over the tile period for the rest of the launch.
    if (g.stagger > 0) {
        const int slots = ((blockIdx.x * 37) & 15) * g.stagger;      // units of one s_sleep(32) ~= 2048 cycles ~= 1 us
        for (int i = 0; i < slots; ++i) __builtin_amdgcn_s_sleep(32);
    }

    int m0, n0, round = 0;
    if (!tile_of(0, m0, n0)) return;
    set_src(m0, n0);
    bias_issue(n0);
    issue_loads(0, 0);
    int cnt = 0;                                   // position in the K-tile stream (stage = cnt & 1)
    int pend = 0;                                  // stores of the previous epilogue still allowed in flight (-1: unknown)
    constexpr int NSTORE = (WR / 8);               // 16-byte stores per lane per tile

    // The bias is the accumulators' initial value (fp32, added before the single bf16 rounding).
    f32x4 bvn[2][4];
    auto bias_read = [&]() {
        const char* row = smem + 2 * STAGE + wave * 256;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                bvn[j][q4] = g.bias ? *reinterpret_cast<const f32x4*>(row + (j * 32 + 8 * q4 + 4 * hi) * 4) : z;
            }
    };

    unsigned long long t_main = 0, t_epi = 0, t_first = 0, t_bar = 0;
    for (;;) {
        const unsigned long long ts0 = g.tim ? __builtin_amdgcn_s_memtime() : 0;
        f32x16 acc[MI][2];
        auto init_acc = [&]() {
            bias_read();
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = bvn[j][r >> 2][r & 3];
        };

        int nm0 = 0, nn0 = 0;
        const bool has_next = tile_of(round + 1, nm0, nn0);
        // Operand loads of the current K-tile (and, on a tile's first K-tile, its bias row) have landed.  Across a
        // tile boundary they are all OLDER than the previous epilogue's NSTORE stores (vmcnt retires in order), so
        // those stores may stay in flight and drain under this tile's MFMAs.
        auto top_wait = [&](bool first) {
            if (first) {
                const unsigned long long w0 = g.tim ? __builtin_amdgcn_s_memtime() : 0;
                if (pend == NSTORE) wait_vm<NSTORE>(); else wait_vm<0>();
                if (g.tim) t_first += __builtin_amdgcn_s_memtime() - w0;
                init_acc();
            }
            else {
                const unsigned long long w0 = g.tim ? __builtin_amdgcn_s_memtime() : 0;
                wait_vm<0>();
                if (g.tim) t_first += __builtin_amdgcn_s_memtime() - w0;
            }
        };

        // One K-tile: 4 k-steps of {6 ds_read_b128, 8 MFMA}.  The (4 + NT) LDS-DMA loads of the NEXT K-tile are
        // issued two per k-step BETWEEN the fragment reads and the MFMAs, so their issue cost hides behind the
        // matrix pipe instead of delaying the first MFMA after the barrier (cdna guide: "the per-phase interleave
        // is the lever").
        auto multiply = [&](int nstage, int nk0, bool do_load) {
            const char* Ab = smem + (cnt & 1) * STAGE;
            const char* Bb = Ab + BOFF;
            char* sb = smem + nstage * STAGE + wave * 1024;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 wf[2], af[MI];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int r = wn * 64 + t * 32 + frow;
                    wf[t] = *reinterpret_cast<const bf16x8*>(Bb + r * 128 + (((ks * 2 + hi) ^ swz(r)) << 4));
                }
#pragma unroll
                for (int t = 0; t < MI; ++t) {
                    const int r = wm * WR + t * 32 + frow;
                    af[t] = *reinterpret_cast<const bf16x8*>(Ab + r * 128 + (((ks * 2 + hi) ^ swz(r)) << 4));
                }
                if (do_load) {                                // two of the next K-tile's LDS-DMA loads per k-step (measured: 4+4 in
                                                              // the first two k-steps trades vmcnt wait for a longer barrier wait, -8 %)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[ks] + nk0),
                                                     (__attribute__((address_space(3))) void*)(sb + ks * 8192), 16, 0, 0);
                    if (ks < NT)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[ks < NT ? ks : 0] + nk0),
                                                         (__attribute__((address_space(3))) void*)(sb + BOFF + ks * 8192), 16, 0, 0);
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
            ++cnt;
        };

        for (int kt = 0; kt + 1 < nk; ++kt) {
            top_wait(kt == 0);                                     // this wave's pieces of the current K-tile have landed
            { const unsigned long long w0 = g.tim ? __builtin_amdgcn_s_memtime() : 0;
              s_barrier_lgkm();                                    // everyone's have; everyone is done with the other stage
              if (g.tim) t_bar += __builtin_amdgcn_s_memtime() - w0; }
            multiply((cnt + 1) & 1, (kt + 1) * TK, true);
        }
        // ---- last K-tile of this tile: the NEXT tile's first K-tile, the bias and the first residual rows are
        //      requested now, so their latency hides behind these 32 MFMAs ---------------------------------------
        top_wait(nk == 1);
        s_barrier_lgkm();
        {   // every address is computed (and any spilled value reloaded) BEFORE the first load is issued, so that
            // no compiler-inserted `s_waitcnt vmcnt(0)` lands behind freshly issued operand loads
            const float* bp = g.bias ? g.bias + min(nn0 + wn * 64 + lane, g.N - 1) : nullptr;
            if (has_next) set_src(nm0, nn0);
            asm volatile("" : "+v"(bp) :: "memory");
            if (has_next && g.bias) lds_dma4(bp, scratch_lds);
        }
        const int slot = lane & 7;
        const int col = n0 + wn * 64 + slot * 8;
        const bool col_ok = col < g.N;
        constexpr int NLD = RP / 8;
        bf16x8 rv[HALVES][NLD];
        auto load_residual = [&](int h) {
            if (g.res && !(g.dbg & 2)) {
#pragma unroll
                for (int it = 0; it < NLD; ++it) {
                    const int grow = m0 + wm * WR + h * RP + it * 8 + (lane >> 3);
                    if (grow < g.M && col_ok) rv[h][it] = *reinterpret_cast<const bf16x8*>(g.res + (int64_t)grow * g.ldc + col);
                }
            }
        };
        multiply((cnt + 1) & 1, 0, has_next);
        const unsigned long long ts1 = g.tim ? __builtin_amdgcn_s_memtime() : 0;
        load_residual(0);
        const bool interior = (m0 + TM <= g.M) && (n0 + TNB <= g.N) && !(g.dbg & 1);

        // ---- epilogue: stage through the stage just consumed ((cnt-1)&1); the other one is receiving
        //      the next tile's first K-tile --------------------------------------------------------
        char* stg = smem + ((cnt - 1) & 1) * STAGE + wave * (RP * 128);
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
            if (h == 0) s_barrier_lgkm();   // every wave's MFMA operand reads of this stage are done (a wave only ever
                                            // touches its OWN staging rows afterwards, and LDS ops of one wave execute in order)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                    for (int ii = 0; ii < MI / HALVES; ++ii) {
                        const int i = h * (MI / HALVES) + ii;
                        bf16x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = acc[i][j][q4 * 4 + e];
                            if (ACT == SETOK_ACT_QUICK_GELU) x = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.45546696f * x));   // x*sigmoid(1.702x); 1.702*log2(e)
                            else if (ACT == SETOK_ACT_GELU_ERF) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
                            v[e] = (bf16)x;
                        }
                        const int row = ii * 32 + frow;
                        *reinterpret_cast<bf16x4*>(stg + row * 128 + (((j * 4 + q4) ^ (row & 7)) << 4) + 8 * hi) = v;
                    }
            if (h + 1 < HALVES) load_residual(h + 1);            // requested before this pass's stores (vmcnt retires in order)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // a wave re-reads only its own staging rows
            constexpr int CH = NLD >= 4 ? NLD / 2 : NLD;           // two chunks keep the register footprint low
#pragma unroll
            for (int c0 = 0; c0 < NLD; c0 += CH) {
                bf16x8 ov[CH];
#pragma unroll
                for (int it = 0; it < CH; ++it) {
                    const int row = (c0 + it) * 8 + (lane >> 3);
                    ov[it] = *reinterpret_cast<const bf16x8*>(stg + row * 128 + ((slot ^ (row & 7)) << 4));
                }
#pragma unroll
                for (int it = 0; it < CH; ++it) {
                    const int grow = m0 + wm * WR + h * RP + (c0 + it) * 8 + (lane >> 3);
                    if (grow < g.M && col_ok && !(g.dbg & 1)) {
                        bf16x8 v = ov[it];
                        if (g.res && !(g.dbg & 2)) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] + (float)rv[h][c0 + it][e]);
                        }
                        *reinterpret_cast<bf16x8*>(g.C + (int64_t)grow * g.ldc + col) = v;
                    }
                }
            }
        }
        if (g.tim) { const unsigned long long ts2 = __builtin_amdgcn_s_memtime(); t_main += ts1 - ts0; t_epi += ts2 - ts1; }
        if (!has_next) break;
        pend = interior ? NSTORE : -1;
        m0 = nm0; n0 = nn0; ++round;
    }
    if (g.tim && tid == 0) { g.tim[blockIdx.x * 4 + 0] = t_main; g.tim[blockIdx.x * 4 + 1] = t_epi; g.tim[blockIdx.x * 4 + 2] = t_first; g.tim[blockIdx.x * 4 + 3] = ((unsigned long long)(round + 1) << 40) | t_bar; }
}

// --------------------------------------------------------------------------------------------
// Tail kernel: the < 1-round remainder of M (p*256 rows) as 256 x 64 tiles, ONE tile per workgroup.
// These launches are latency-bound (few workgroups, 16-64 K-tiles each), so the 40 KiB stages are
// triple-buffered: two K-tiles of operands are in flight while one is multiplied.
// --------------------------------------------------------------------------------------------
constexpr int TSTAGE = 40 * 1024;                // A 32 KiB + W 8 KiB
constexpr int TAIL_LDS = 3 * TSTAGE + 256;       // + bias row

template <int ACT>
__global__ __launch_bounds__(512) void gemm_tail_kernel(PArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, hi = lane >> 5;
    const int nk = g.K / TK;
    const int m0 = (blockIdx.x / g.tilesN) * TM, n0 = (blockIdx.x % g.tilesN) * 64;
    float* sbias = reinterpret_cast<float*>(smem + 3 * TSTAGE);
    if (tid < 64) sbias[tid] = g.bias ? g.bias[min(n0 + tid, g.N - 1)] : 0.f;

    const bf16* a_src[4]; const bf16* b_src;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = i * 512 + tid, row = p >> 3, kc = (p & 7) ^ swz(row);
        a_src[i] = g.A + (int64_t)min(m0 + row, g.M - 1) * g.lda + kc * 8;
        if (i == 0) b_src = g.W + (int64_t)min(n0 + row, g.N - 1) * g.K + kc * 8;
    }
    // LDS-DMA through inline asm: with two K-tiles (10 loads per lane) in flight hipcc's waitcnt pass runs out of
    // tracked LDS-DMA slots (8) and falls back to `s_waitcnt vmcnt(0)` in front of every ds_read, which serialises the
    // whole pipeline (measured: 2.4 us per K-tile instead of 0.6).  All waits in this loop are the explicit ones below.
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem) + wave * 1024;
    auto dma16 = [&](const bf16* ptr, unsigned lds_dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(ptr), "s"(lds_dst) : "memory");
    };
    auto issue = [&](int stage, int k0, int i) {            // piece i of a K-tile: A chunk i (+ the W chunk with i == 0)
        const unsigned sb = lds0 + stage * TSTAGE;
        dma16(a_src[i] + k0, sb + i * 8192);
        if (i == 0) dma16(b_src + k0, sb + BOFF);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) issue(0, 0, i);
    if (nk > 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue(1, TK, i);
    }
    f32x16 acc[2];
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) wait_vm<5>(); else wait_vm<0>();   // K-tile kt has landed; kt+1 (5 loads per lane) may still fly
        s_barrier_lgkm();
        if (kt == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = sbias[j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi];
        }
        const char* Ab = smem + (kt % 3) * TSTAGE;
        const char* Bb = Ab + BOFF;
        const bool more = kt + 2 < nk;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 wf[2], af;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int r = t * 32 + frow;
                wf[t] = *reinterpret_cast<const bf16x8*>(Bb + r * 128 + (((ks * 2 + hi) ^ swz(r)) << 4));
            }
            const int r = wave * 32 + frow;
            af = *reinterpret_cast<const bf16x8*>(Ab + r * 128 + (((ks * 2 + hi) ^ swz(r)) << 4));
            if (more) issue((kt + 2) % 3, (kt + 2) * TK, ks);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af, acc[j], 0, 0, 0);
        }
    }
    // epilogue: 32 rows x 64 columns per wave, staged through stage 0 (every load has landed)
    const int slot = lane & 7, col = n0 + slot * 8;
    const bool col_ok = col < g.N;
    bf16x8 rv[4];
    if (g.res) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int grow = m0 + wave * 32 + it * 8 + (lane >> 3);
            if (grow < g.M && col_ok) rv[it] = *reinterpret_cast<const bf16x8*>(g.res + (int64_t)grow * g.ldc + col);
        }
    }
    s_barrier_lgkm();
    char* stg = smem + wave * (32 * 128);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[j][q4 * 4 + e];
                if (ACT == SETOK_ACT_QUICK_GELU) x = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.45546696f * x));
                else if (ACT == SETOK_ACT_GELU_ERF) x = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
                v[e] = (bf16)x;
            }
            *reinterpret_cast<bf16x4*>(stg + frow * 128 + (((j * 4 + q4) ^ (frow & 7)) << 4) + 8 * hi) = v;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3);
        const int grow = m0 + wave * 32 + row;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + row * 128 + ((slot ^ (row & 7)) << 4));
        if (grow < g.M && col_ok) {
            if (g.res) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] + (float)rv[it][e]);
            }
            *reinterpret_cast<bf16x8*>(g.C + (int64_t)grow * g.ldc + col) = v;
        }
    }
}

int launch_tail(hipStream_t s, const PArgs& g, int act) {
    static bool attr_set = false;
    if (!attr_set) {
        bool ok = hipFuncSetAttribute((const void*)gemm_tail_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)gemm_tail_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)gemm_tail_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS) == hipSuccess;
        if (!ok) return setok_fail(SETOK_ELAUNCH, "setok_linear: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    const int grid = g.tilesM * g.tilesN;
    if (act == SETOK_ACT_NONE) gemm_tail_kernel<0><<<grid, 512, TAIL_LDS, s>>>(g);
    else if (act == SETOK_ACT_QUICK_GELU) gemm_tail_kernel<1><<<grid, 512, TAIL_LDS, s>>>(g);
    else gemm_tail_kernel<2><<<grid, 512, TAIL_LDS, s>>>(g);
    SETOK_CHECK_LAUNCH("setok_linear(tail)");
    return SETOK_OK;
}

template <int NT>
int launch_nt(hipStream_t s, const PArgs& g, int act, int n_cu) {
    static bool attr_set = false;
    if (!attr_set) {
        bool ok = hipFuncSetAttribute((const void*)gemm_persist_kernel<NT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)gemm_persist_kernel<NT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess;
        ok = ok && hipFuncSetAttribute((const void*)gemm_persist_kernel<NT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) == hipSuccess;
        if (!ok) return setok_fail(SETOK_ELAUNCH, "setok_linear: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    const int tiles = g.tilesM * g.tilesN;
    const int grid = tiles < n_cu ? tiles : n_cu;
    if (act == SETOK_ACT_NONE) gemm_persist_kernel<NT, 0><<<grid, 512, LDS_BYTES, s>>>(g);
    else if (act == SETOK_ACT_QUICK_GELU) gemm_persist_kernel<NT, 1><<<grid, 512, LDS_BYTES, s>>>(g);
    else gemm_persist_kernel<NT, 2><<<grid, 512, LDS_BYTES, s>>>(g);
    SETOK_CHECK_LAUNCH("setok_linear(persistent)");
    return SETOK_OK;
}

int cu_count() {
    static int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    return n;
}

}  // namespace

// Called by setok_linear (gemm.hip) for bf16 -> bf16 problems with >= 96 tiles of 256x256.
int setok_gemm_persist_bf16(hipStream_t s, const bf16* A, int64_t lda, const bf16* W, const float* bias, const bf16* res,
                            bf16* C, int64_t ldc, int M, int N, int K, int act) {
    const int ncu = cu_count();
    const int tilesM = cdiv(M, TM), tilesN = cdiv(N, 256);
    // Peel off p <= 2 trailing M-tiles when that leaves the main launch an exact number of rounds: the
    // remainder (p*256 rows) runs as 256x64 tiles — a fraction of a round instead of a full extra one.
    int p = 0;
    const int T = tilesM * tilesN, r = T % ncu;
    if (T > ncu && r != 0 && r % tilesN == 0 && r / tilesN <= 2) p = r / tilesN;
    const int tm_main = tilesM - p;
    static const int dbg = [] { const char* e = getenv("SETOK_GEMM_DEBUG"); return e ? atoi(e) : 0; }();
    if (dbg & 4) p = 0;
    static const int stg = [] { const char* e = getenv("SETOK_GEMM_STAGGER"); return e ? atoi(e) : 0; }();
    static const bool timing = [] { const char* e = getenv("SETOK_GEMM_TIMING"); return e && e[0] == '1'; }();
    static unsigned long long* tim = nullptr;
    if (timing && !tim) { if (hipMalloc(&tim, 256 * 4 * 8) != hipSuccess) tim = nullptr; }
    PArgs g{A, W, bias, res, C, lda, ldc, (tilesM - p) * TM < M ? (tilesM - p) * TM : M, N, K, tilesM - p, tilesN, dbg, stg, timing ? tim : nullptr};
    int rc = launch_nt<4>(s, g, act, ncu);
    if (timing && tim) {
        unsigned long long h[256 * 4];
        if (hipMemcpy(h, tim, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            double a = 0, b = 0, c = 0, r = 0; int nb = ncu < tilesN * (tilesM - p) ? ncu : tilesN * (tilesM - p);
            double bar = 0;
            for (int i = 0; i < nb; ++i) { a += h[i * 4]; b += h[i * 4 + 1]; c += h[i * 4 + 2]; r += (double)(h[i * 4 + 3] >> 40); bar += (double)(h[i * 4 + 3] & ((1ull << 40) - 1)); }
            fprintf(stderr, "[gemm timing] M=%d N=%d K=%d tiles/block=%.2f  per tile: main %.0f cyc (vmcnt waits %.0f, barrier waits %.0f), epilogue %.0f cyc\n",
                    M, N, K, r / nb, a / r, c / r, bar / r, b / r);
        }
    }
    if (rc != SETOK_OK || p == 0) return rc;
    const int m_off = tm_main * TM;
    PArgs t{A + (int64_t)m_off * lda, W, bias, res ? res + (int64_t)m_off * ldc : nullptr, C + (int64_t)m_off * ldc,
            lda, ldc, M - m_off, N, K, cdiv(M - m_off, TM), cdiv(N, 64), dbg, 0, nullptr};
    return launch_tail(s, t, act);
}
