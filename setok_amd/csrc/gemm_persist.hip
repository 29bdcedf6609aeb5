// gemm_persist.hip — persistent bf16 GEMM for the large Linear layers (the dominant kernel of the path).
//
//   C[M,N] = act(A[M,K] · W[N,K]^T + bias) (+ residual, torch-bf16 semantics: the Linear output is rounded
//   to bf16 first, then the residual is added and the sum rounded again)
//
// One workgroup (8 waves, 2 per SIMD) per CU walks output tiles of 256 x 256:
//   * operands go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip), every one issued from inline asm so that
//     no wait in the kernel is the compiler's guess.  The LDS image is lane-linear, so the 16-B-slot XOR swizzle is applied to
//     the per-lane SOURCE address and to the ds_read_b128 address.  Two 64 KiB stages.
//   * the K-tiles of consecutive output tiles form ONE continuous stream: while the last K-tile of a tile is multiplied, the
//     first K-tile of the workgroup's NEXT tile is already in flight, and its second one is requested before the epilogue's
//     first store, so tile boundaries pay neither a prologue latency nor a pipeline drain.  Barriers are raw s_barrier +
//     explicit counted s_waitcnt (a __syncthreads() would drain the in-flight LDS-DMA).
//   * MFMA operands are swapped (A-operand = W fragment, B-operand = activation fragment): a lane then holds, for ONE output
//     row, 4 consecutive columns per accumulator quad.  The accumulators start at the bias (scalar loads); the epilogue applies
//     the activation in registers, transposes bf16 through 32 KiB of LDS of its own (XOR-swizzled 16-B slots, one 32-row MFMA
//     tile per pass) and writes full 128-byte row segments with 16-byte stores; the residual is fetched with coalesced 16-byte
//     loads one pass ahead.
//   * tile order: the 32 workgroups resident on one XCD (private 4 MiB L2) cover an 8 (M) x 4 (N) patch of tiles in every
//     round and share operand panels.
//   * a second kernel (256 x 64 tiles, 8 waves stacked along M, three 40 KiB stages) finishes the < 1-round remainder of M so
//     that the big launch runs an exact number of rounds (no tail).
// The chip is POWER-bound on this kernel (measured with rocm-smi under load: 1.37-1.38 kW of the 1.4 kW cap, sclk 1.73-1.95 GHz on
// random operands vs 2.39 GHz / 0.95 kW on all-zero operands), so removing stall cycles converts only partly into speed: the
// tile-boundary change above cut 9 % of the cycles per tile and 3 % of the time.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int TM = 256, TK = 64;
constexpr int STAGE = 64 * 1024;                 // bytes per pipeline stage (A tile at 0, W tile at BOFF)
constexpr int BOFF = TM * TK * 2;                // 32 KiB

struct PArgs {
    const bf16* A; const bf16* W; const float* bias; const bf16* res; bf16* C;
    int64_t lda, ldc;
    int M, N, K, tilesM, tilesN, dbg;
    unsigned long long* tim;     // debug: per-block {main-loop, epilogue, wait-at-first-ktile} cycle sums
    const float* zero_bias;      // 64 zeros: a null bias is scalar-loaded like a real one
    float* Cf;                   // F32B variant: fp32 output, `nbatch` independent problems (split-K partial products of a weight gradient)
    int64_t sA, sW, sC;          //   element strides between the batch members
    int nbatch;
    const float* ln_stats;       // LNK variant: per row of A 8 floats {row fragment (4 dwords), rstd, mean, 0, 0} (setok_row_stats): the LayerNorm folded into this GEMM
    const float* ln_colsum;      //   per column 4 dwords: the column fragment of c[n] = sum_k W'[n][k] and b'[n] = b[n] + sum_k W[n][k] beta[k] (setok_ln_fold)
    const int32_t* m_dev;        // optional device-side row count (<= M): tiles beyond it are never visited (setok_linear_dev)
    int rem_rows;                // ping-pong kernel only: rows [M, M + rem_rows) — the remainder behind the whole 256-row tiles — are finished by the same launch
    int rem_tiles;               //   as small tiles (gemm_tail_tile), one per workgroup 0 .. rem_tiles - 1, after the workgroup's last 256 x 256 tile
    int rem_shape;               //   64: 64 x 64 tiles, 32: 32 x 32 tiles (what tail_shape_for picks for the remainder as a launch of its own)
};

__device__ inline void s_barrier_lgkm() {         // LDS ops of this wave retired, then the workgroup barrier
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// MFMA shape: v_mfma_f32_16x16x32_bf16.  A register-resident loop of it sustains 2080 TFLOP/s on random bf16 operands under the socket's
// power management against 1855 for v_mfma_f32_32x32x16_bf16 (759 W vs 879 W on all-zero operands at the same 2456 TFLOP/s: half the
// accumulator traffic per FLOP) — tools/micro/mfma_peak.hip, profiles/r01_mfma_peak.log; and this GEMM is power-bound.
// Fragment of a 16-row tile: lane l reads row (l & 15), 16-byte slot ks * 4 + (l >> 4) of the 128-byte LDS row (8 consecutive k).
// 16-byte-slot swizzle: a ds_read_b128 is served in four 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32), one 256-byte bank
// row per cycle.  A group reads the 16 rows of a tile, 8 of them at slot c and 8 at slot c + 1; with slot ^= (row >> 1) & 7 the eight
// rows of either parity fall on eight different slots for every c -> conflict-free.
__device__ inline int swz(int row) { return (row >> 1) & 7; }
typedef __attribute__((ext_vector_type(4))) float f32x4;

// --------------------------------------------------------------------------------------------
// LayerNorm folded into the consuming GEMM (bf16 throughput mode).  With W' = gamma * W:
//     LN(x) W^T + b  =  rstd_r * ( x W'^T  -  mean_r * c  +  b' / rstd_r ),      c = W' 1,   b' = b + W beta
// so the GEMM streams the RAW residual stream x through the unchanged LDS-DMA operand path (no normalised copy of x is ever written or
// read: the separate LayerNorm pass was 5 % of the encode step), the accumulators START at the rank-2 correction
// (-mean_r) c_n + (1 / rstd_r) b'_n — formed by the matrix pipe itself from two-way bf16 splits (8 k-slots of one 16x16x32 MFMA) —
// and the epilogue multiplies by rstd_r.  Both GEMM kernels build the fragments with the functions below, so a row's result does not
// depend on which kernel produced it.
// --------------------------------------------------------------------------------------------
// The fragments themselves (8 bf16 = 16 bytes per row / per column) are written once by setok_row_stats / setok_ln_fold (common.h:
// ln_row_frag / ln_col_frag); a GEMM lane only loads them — the k-slots 0-7 of a 16x16x32 operand live in lanes 0-15, the other lanes hold zeros.
__device__ inline bf16x8 ln_frag_lane(const f32x4 raw, int g4) {
    f32x4 v = raw;
    if (g4 != 0) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
    return __builtin_bit_cast(bf16x8, v);
}

template <int N_>
__device__ inline void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// --------------------------------------------------------------------------------------------
// The tile boundary.  On gfx950 stores and loads retire through ONE in-order counter (vmcnt).  An earlier version of this
// kernel transposed the epilogue through the operand stage it had just consumed and requested the next tile's second K-tile
// after the stores; s_memtime showed 11 k of 75 k cycles per tile (K = 1024) going to the stores — 3 k of issue and 7 k of
// `vmcnt` wait for operands queued behind them.  Hence
//   * the epilogue has its OWN 32 KiB of LDS (8 waves x 32 rows x 128 B), so the stage freed by a tile's last K-tile
//     receives the NEXT tile's second K-tile BEFORE the first store is issued: at the next tile's start two K-tiles are in
//     flight ahead of the stores, the first two waits are counted (`vmcnt(NSTORE + 8)`, `vmcnt(NSTORE)`) and the stores drain
//     under 64 MFMAs per wave;
//   * to make room the bias row left LDS: it is read with scalar loads (lgkmcnt domain, no vmcnt interaction at all).
// --------------------------------------------------------------------------------------------
constexpr int MAIN_LDS = 2 * STAGE + 8 * 4096;   // 163840: all of the CU's LDS
__device__ float kZeroBias[64];                    // stands in for a null bias (scalar-loaded like a real one)

// F32B = false: the bf16-out kernel of every large Linear.  F32B = true: fp32 output written straight from the accumulators (the outputs
// are weight-gradient sized, store efficiency is irrelevant) for a BATCH of problems sharing M, N, K — the split-K partial products of
// dW = dY^T X (training.py), which by themselves have too few output tiles to fill the chip.
// RESK: the launch has a residual (its own instantiation: the residual rows' registers and code do not burden the kernels without one).
// LNK: the LayerNorm of the A rows is folded in (see above): accumulators start at the rank-2 correction, the epilogue scales by rstd.
template <int ACT, bool F32B, bool RESK = false, bool LNK = false>
__global__ __launch_bounds__(512) void gemm_persist_kernel(PArgs g) {
    static_assert(!(LNK && (F32B || RESK)), "the folded LayerNorm feeds qkv / fc1: bf16 out, no residual");
    constexpr int WNC = 4, WR = 128, MI = 4, TNB = 256;   // MI: 32-row epilogue passes per wave
    constexpr int MT = 8, NT = 4;                           // 16 x 16 MFMA tiles per wave: 8 along M (128 rows) x 4 along N (64 columns)
    constexpr int NL = 8;                          // LDS-DMA ops per lane per K-tile
    constexpr int NSTORE = WR / 8;                 // 16-byte stores per lane per (interior) tile
    constexpr int NLN = LNK ? 5 : 0;               // LNK: loads per lane per tile (one column fragment, two compact row fragments, two rstd)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNC, wn = wave % WNC;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int nk = g.K / TK;
    // rows of this launch: the host's M, or the device-side count of a ragged stage (uniform: one scalar load)
    const int Mrt = g.m_dev ? min(g.M, __builtin_amdgcn_readfirstlane(*g.m_dev)) : g.M;
    const int tilesMrt = g.m_dev ? (Mrt + TM - 1) / TM : g.tilesM;
    const int tiles_per_problem = tilesMrt * g.tilesN;
    const int num_tiles = tiles_per_problem * (F32B ? g.nbatch : 1);
    const int G = gridDim.x;
    int bz = 0, nbz = 0;                           // batch member of the current / next tile (F32B)

    // tile id for (round, block): XCD x (= blockIdx % 8) owns 32 consecutive ids per round = an 8 (M) x 4 (N) patch
    auto tile_of = [&](int round, int& m0, int& n0, int& b) -> bool {
        int L;
        if ((G & 7) == 0) L = round * G + (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
        else L = round * G + blockIdx.x;
        if (L >= num_tiles) return false;
        b = 0;
        if (F32B) { b = L / tiles_per_problem; L -= b * tiles_per_problem; }
        constexpr int GM = 8;
        const int per = GM * g.tilesN, group = L / per, first_m = group * GM;
        const int gm = min(tilesMrt - first_m, GM), in = L - group * per;
        m0 = (first_m + in % gm) * TM;
        n0 = (in / gm) * TNB;
        return true;
    };

    // LDS-DMA source addresses: a wave-uniform 64-bit base (SGPR pair: the tile's first row, advanced by the K offset with scalar adds)
    // + a 32-bit byte offset per lane and piece (row inside the tile, clamped at the matrix edge, and the swizzled 16-byte slot): half
    // the address registers of 64-bit per-lane pointers for the address unit to read (qkv / fc1 +4 % sustained), 8 VGPRs less.
    unsigned a_off[4], b_off[4];
    const char* a_base; const char* b_base;
    auto set_src = [&](int m0, int n0, int b) {
        const bf16* Ab_ = F32B ? g.A + (int64_t)b * g.sA : g.A;
        const bf16* Wb_ = F32B ? g.W + (int64_t)b * g.sW : g.W;
        a_base = reinterpret_cast<const char*>(Ab_ + (int64_t)m0 * g.lda);
        b_base = reinterpret_cast<const char*>(Wb_ + (int64_t)n0 * g.K);
        // rows / slots from an opaque copy of the thread id: computed outside the tile loop these eight values live across the main loop and, in the
        // kernel with a residual, came back from scratch one load + vmcnt(0) at a time at every tile boundary
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = i * 512 + tid_e, row = p >> 3, kc = (p & 7) ^ swz(row);
            a_off[i] = (unsigned)min(row, Mrt - 1 - m0) * (unsigned)(g.lda * 2) + kc * 16;       // < 256 rows x lda x 2 bytes
            b_off[i] = (unsigned)min(row, g.N - 1 - n0) * (unsigned)(g.K * 2) + kc * 16;
        }
    };
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem) + wave * 1024;
    auto dma16 = [&](const char* base, unsigned off, unsigned lds_dst) {
        unsigned keep;
        const unsigned long long b64 = (unsigned long long)base;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);          // (readfirstlane returns int: widen as unsigned)
        const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));
        const unsigned long long sb64 = (unsigned long long)lo | ((unsigned long long)hi32 << 32);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(sb64), "s"(lds_dst) : "memory");
    };
    auto issue_ktile = [&](int stage, int k0) {                      // a whole K-tile at once (tile boundaries only)
        const unsigned sb = lds0 + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(a_base + k0 * 2, a_off[i], sb + i * 8192);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(b_base + k0 * 2, b_off[i], sb + BOFF + i * 8192);
    };

    // The bias of a tile is fetched ONE TILE AHEAD (4 floats per lane: columns j * 16 + (lane & 15) of the wave's 64), before the previous
    // tile's last K-tile — i.e. ahead of that tile's store burst.  Read at the top of the tile (scalar loads then) it cost ~4 k cycles per
    // tile: the request sat behind the stores of all 256 CUs.  The accumulators then start at the bias through 32 MFMAs on a fragment that
    // holds the bias split into three bf16 parts (hi + mid + lo = the fp32 value exactly) against a fragment of ones: acc = 0 + hi + mid + lo
    // is exact, so the arithmetic (bias first, then the products in ascending k) is unchanged bit for bit.
    float nb[NT];
#ifdef SETOK_HALF
    f32x4 nbq[NT];
#endif
    // LNK: a lane fetches ONE column fragment (16 bytes: column g4 * 16 + l15 of the wave's 64) and the compact form (8 bytes) of TWO row
    // fragments (rows (2 g4 + i) * 16 + l15 of the wave's 128) — 8 registers instead of 48 for the tile's 12 fragments; at the tile's start the
    // lanes that feed the MFMA's k-slots 0-7 (lanes 0-15) collect them from the other 16-lane groups with ds_bpermute.
    f32x4 ncw; float2 nrw[2];
    float nrs[2], ers[2];                          // rstd of the same two rows: of the NEXT tile (fetched with the fragments) / of the CURRENT one
    auto load_bias = [&](int n0_, int m0_) {
        if constexpr (!F32B && !LNK) {
            const float* bp = g.bias ? g.bias + min(n0_ + wn * 64, g.N - 64) : g.zero_bias;
#ifdef SETOK_HALF
            // fp16 build: the accumulators start at the fp32 bias DIRECTLY (a lane's quad = columns j * 16 + 4 g4 .. + 3, as in the ping-pong and small-tile
            // kernels).  The matrix-pipe start of the bf16 build needs an EXACT split of an fp32 value into 16-bit parts: bf16 has fp32's exponent range, fp16
            // does not — the third part of a bias of order 1 is 2^-22, a subnormal half with two bits left — and a start value one fp32 ulp off flipped the
            // fp16 rounding of 8e-5 of this kernel's outputs against the other kernels' (tools/fuzz_gpu.py ... f16, seed 7).
#pragma unroll
            for (int j = 0; j < NT; ++j) nbq[j] = *reinterpret_cast<const f32x4*>(bp + (g.bias ? j * 16 + 4 * g4 : 0));
#else
#pragma unroll
            for (int j = 0; j < NT; ++j) nb[j] = bp[j * 16 + l15];
#endif
        }
    };
    // They are fetched in the MIDDLE of the previous tile's epilogue (after its second pass), so that their L2 round trip (1-2 us under
    // load) has the rest of the epilogue to complete: fetched at the tile's own start the round trip was exposed (-7 % on the qkv shape);
    // as 12 full fragments per lane a tile ahead they spilled 40-64 registers (-17 %).  NLN extra entries sit in the vector-memory queue
    // between the stores of passes 1 and 2: the counted waits at the next tile's start allow for them.
    auto load_ln_frags = [&](int n0_, int m0_) {
        if constexpr (LNK) {
            ncw = *reinterpret_cast<const f32x4*>(g.ln_colsum + 4 * (int64_t)(min(n0_ + wn * 64, g.N - 64) + g4 * 16 + l15));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* st = g.ln_stats + 8 * (int64_t)min(m0_ + wm * WR + (2 * g4 + i) * 16 + l15, Mrt - 1);
                nrw[i] = *reinterpret_cast<const float2*>(st);
                nrs[i] = st[4];
            }
        }
    };

    int m0, n0, round = 0;
    if (!tile_of(0, m0, n0, bz)) return;
    set_src(m0, n0, bz);
    load_bias(n0, m0);
    load_ln_frags(n0, m0);                         // (the first tile's: ahead of everything)
    issue_ktile(0, 0);
    issue_ktile(1, TK);
    int cnt = 0;                                   // position in the K-tile stream (stage = cnt & 1)
    int pend = 0;                                  // NSTORE: the previous epilogue issued exactly NSTORE stores per lane; else unknown (<= NSTORE)

    unsigned long long t_main = 0, t_epi = 0, t_first = 0, t_bar = 0;
    for (;;) {
        const unsigned long long ts0 = g.tim ? __builtin_amdgcn_s_memtime() : 0;
        int nm0 = 0, nn0 = 0;
        const bool has_next = tile_of(round + 1, nm0, nn0, nbz);

        f32x4 acc[MT][NT];

        // One K-tile: 2 k-steps of {12 ds_read_b128, 32 MFMA 16x16x32}; with LOAD the 8 LDS-DMA loads of K-tile `k_next` (this tile's
        // next one, or the next tile's first) go out between the MFMAs of the first k-step.
        auto multiply = [&](auto load_tag, bool do_load, int k_next) {
            constexpr bool LOAD = decltype(load_tag)::value;
            const char* Ab = smem + (cnt & 1) * STAGE;
            const char* Bb = Ab + BOFF;
            const unsigned sb = lds0 + ((cnt + 1) & 1) * STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 wf[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int r = wn * 64 + t * 16 + l15;
                    wf[t] = *reinterpret_cast<const bf16x8*>(Bb + r * 128 + (((ks * 4 + g4) ^ swz(r)) << 4));
                }
                // The whole next K-tile is requested during the first half of this one, ONE request after every fourth MFMA (64 cycles):
                // requested early they have the second half to land before the vmcnt wait at the top of the next K-tile (+1..5 % against
                // requests spread over the whole K-tile), and a vector-memory instruction holds its wave until the address unit has taken
                // it (16 cycles per 1 KiB request, all eight waves asking), so several in a row stalled the MFMA stream behind them
                // (spread out: +1..4 % on the ViT shapes, +5..8 % at 4096^3 / 8192^3, alternating runs on one box).
#pragma unroll
                for (int half = 0; half < 2; ++half) {                     // the activation fragments in two batches of four (register budget)
                    bf16x8 af[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int r = wm * WR + (half * 4 + t) * 16 + l15;
                        af[t] = *reinterpret_cast<const bf16x8*>(Ab + r * 128 + (((ks * 4 + g4) ^ swz(r)) << 4));
                    }
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[half * 4 + t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[t], acc[half * 4 + t][j], 0, 0, 0);
                        if (LOAD && ks == 0) {
                            if (do_load) {
                                __builtin_amdgcn_sched_barrier(0);
                                if (half == 0) dma16(a_base + k_next * 2, a_off[t], sb + t * 8192);
                                else dma16(b_base + k_next * 2, b_off[t], sb + BOFF + t * 8192);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                    __builtin_amdgcn_s_setprio(0);
                }
            }
            ++cnt;
        };
        using yes = std::true_type; using no = std::false_type;

        // In flight at this point, oldest first: K-tile 0, K-tile 1, the previous epilogue's stores (nk >= 3: dispatch condition)
        // [LNK: + the NLN fragment loads of this tile between them].
        {
            const unsigned long long w0 = g.tim ? __builtin_amdgcn_s_memtime() : 0;
            if (pend == NSTORE) wait_vm<NSTORE + NL + NLN>(); else wait_vm<NL>();       // K-tile 0 has landed
            if (g.tim) t_first += __builtin_amdgcn_s_memtime() - w0;
        }
        s_barrier_lgkm();
        if constexpr (LNK) {   // ---- accumulators start at (-mean_r) c_n + (1 / rstd_r) b'_n ---------------------------------------------------
            auto from_group = [&](float v, int grp) {              // lane (l15, *) reads lane (l15, grp)
                return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((l15 + 16 * grp) << 2, __builtin_bit_cast(int, v)));
            };
            bf16x8 cfr[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = from_group(ncw[e], j);
                cfr[j] = ln_frag_lane(v, g4);
            }
            ers[0] = nrs[0]; ers[1] = nrs[1];                     // kept for this tile's epilogue: a load there would wait for the operand DMA in flight
            f32x4 z;
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = 0.f;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const float d0 = from_group(nrw[t & 1].x, t >> 1), d1 = from_group(nrw[t & 1].y, t >> 1);
                const f32x4 v = {d0, d0, d1, d1};                  // (-mean hi, lo) twice, (1 / rstd hi, lo) twice
                const bf16x8 rfr = ln_frag_lane(v, g4);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cfr[j], rfr, z, 0, 0, 0);
            }
        } else {   // ---- accumulators start at the bias (fp32, added before the single bf16 rounding) -------------------------------------------
#ifdef SETOK_HALF
            if constexpr (F32B) {
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int j = 0; j < NT; ++j) { f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[t][j] = z; }
            } else {
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[t][j] = nbq[j];
            }
        }
#else
            bf16x8 ones;
#pragma unroll
            for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bf16x8 bw;
#pragma unroll
                for (int e = 0; e < 8; ++e) bw[e] = (bf16)0.0f;
                if constexpr (!F32B) {
                    // three-way exact split by truncation (top 16 bits of the fp32 pattern each time); only the k-group-0 lanes carry it
                    const float b = nb[j];
                    const float hi1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) & 0xffff0000u);
                    const float r1 = b - hi1;
                    const float hi2 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & 0xffff0000u);
                    const float r2 = r1 - hi2;
                    const bool fin = __builtin_isfinite(b);
                    if (g4 == 0) {
                        bw[0] = __builtin_bit_cast(bf16, (unsigned short)(__builtin_bit_cast(unsigned, b) >> 16));
                        bw[1] = fin ? __builtin_bit_cast(bf16, (unsigned short)(__builtin_bit_cast(unsigned, r1) >> 16)) : (bf16)0.0f;
                        bw[2] = fin ? __builtin_bit_cast(bf16, (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16)) : (bf16)0.0f;
                    }
                }
                f32x4 z;
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = 0.f;
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw, ones, z, 0, 0, 0);
            }
        }
#endif
        multiply(no{}, false, 0);                                                 // K-tile 1 was requested at the previous tile boundary
        {
            const unsigned long long w0 = g.tim ? __builtin_amdgcn_s_memtime() : 0;
            if (pend == NSTORE) wait_vm<NSTORE + NLN>(); else wait_vm<0>();       // K-tile 1 has landed; the stores may still fly
            if (g.tim) t_first += __builtin_amdgcn_s_memtime() - w0;
        }
        for (int kt = 1; kt + 1 < nk; ++kt) {
            s_barrier_lgkm();                                                      // everyone's pieces have landed; everyone is done with the other stage
            multiply(yes{}, true, (kt + 1) * TK);
            wait_vm<0>();                                                          // this wave's pieces of K-tile kt + 1
        }
        // ---- tile boundary.  Order of the vector-memory queue from here (it retires in order):
        //        residual rows of pass 0 | next tile's K-tile 0 (during the last multiply) | next tile's K-tile 1 |
        //        residual rows of pass 1 | stores of pass 0 | residual 2 | stores 1 | residual 3 | stores 2 | stores 3
        //      so the next tile waits vmcnt(NSTORE + 8) for its first K-tile and vmcnt(NSTORE) for its second, and no wait
        //      in here asks for a store to have completed.
        const bool interior = (m0 + TM <= Mrt) && (n0 + TNB <= g.N) && !(g.dbg & 1);
        constexpr bool use_res = RESK;
        // The lane-only parts of the store / residual addresses are recomputed per tile from an opaque copy of the lane id: hoisted out of
        // the tile loop they are ~30 registers that live across the main loop, get spilled, and come back one scratch load per store
        // (epilogue 6 k -> 15 k cycles per tile).
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int slot = lane_e & 7, lrow = lane_e >> 3;
        const int col = n0 + wn * 64 + slot * 8;
        const bool col_ok = col < g.N;
        char* stg = smem + 2 * STAGE + wave * 4096;
        // C / residual addresses = wave-uniform base of the wave's 128 x 64 block + uniform (h, it) row offset + ONE 32-bit lane offset: the
        // 16 stores and 16 residual loads of a tile share a single offset register and go out in SGPR-base form.
        const int64_t wave_elem = (int64_t)(m0 + wm * WR) * g.ldc + (n0 + wn * 64);
        char* c_wave = reinterpret_cast<char*>(g.C + wave_elem);
        const char* r_wave = reinterpret_cast<const char*>(g.res + (RESK ? wave_elem : 0));
        const unsigned lane_off = (unsigned)(lrow * (int)g.ldc + slot * 8) * 2u;
        const unsigned row8 = (unsigned)g.ldc * 16u;                          // bytes between consecutive `it` (8 rows)
        bf16x8 rv[1][4];
        auto load_residual = [&](auto int_tag, int h) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int grow = m0 + wm * WR + h * 32 + it * 8 + lrow;
                if (decltype(int_tag)::value || (grow < Mrt && col_ok))
                    rv[0][it] = *reinterpret_cast<const bf16x8*>(r_wave + (size_t)(h * 4 + it) * row8 + lane_off);
            }
        };
        s_barrier_lgkm();
        if (has_next) { set_src(nm0, nn0, nbz); if (!RESK && !LNK) load_bias(nn0, nm0); }       // addresses first, loads after: no reload lands behind a DMA
        if (use_res) { if (interior) load_residual(yes{}, 0); else load_residual(no{}, 0); }

        multiply(yes{}, has_next, 0);                                              // last K-tile; the next tile's first one goes out
        const unsigned long long ts1 = g.tim ? __builtin_amdgcn_s_memtime() : 0;
        s_barrier_lgkm();                                                          // every wave is done with the stage just multiplied: it
                                                                                   // receives the next tile's SECOND K-tile inside pass 0
        // One 32-row MFMA tile per pass through this wave's private 4 KiB of staging.
        if (RESK && has_next) load_bias(nn0, nm0);   // (with a residual the registers are tighter during the last K-tile: fetched here, still ahead of the stores)
        auto epilogue = [&](auto res_tag, auto int_tag) {
            constexpr bool RES = decltype(res_tag)::value, INT = decltype(int_tag)::value;
#pragma unroll
            for (int h = 0; h < MI; ++h) {
                float er_pass[2] = {1.f, 1.f};                            // LNK: rstd of this pass's rows (t = 2 h + tt): held by the lanes of 16-lane group h
                if constexpr (LNK) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
                        er_pass[tt] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((l15 + 16 * h) << 2, __builtin_bit_cast(int, ers[tt])));
                }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)                            // the two 16-row MFMA tiles of this 32-row pass
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        bf16x4 v;
                        f32x4 xs = acc[2 * h + tt][j];
                        if constexpr (LNK) xs = xs * er_pass[tt];            // one row, four columns: packed multiplies
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = xs[e];
                            if (ACT == SETOK_ACT_QUICK_GELU) x = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.45546696f * x));   // x*sigmoid(1.702x); 1.702*log2(e)
                            else if (ACT == SETOK_ACT_GELU_ERF) x = gelu_erf_fast(x);
                            v[e] = (bf16)x;
                        }
                        // row tt * 16 + l15 of the pass, columns j * 16 + 4 * g4 .. + 3: 16-byte slot j * 2 + (g4 >> 1), its half g4 & 1
                        const int srow = tt * 16 + l15;
                        *reinterpret_cast<bf16x4*>(stg + srow * 128 + (((j * 2 + (g4 >> 1)) ^ (srow & 7)) << 4) + 8 * (g4 & 1)) = v;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // a wave re-reads only its own staging rows
                bf16x8 ov[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + lrow;
                    ov[it] = *reinterpret_cast<const bf16x8*>(stg + row * 128 + ((slot ^ (row & 7)) << 4));
                }
                if (RES) {
#pragma unroll
                    for (int it = 0; it < 4; ++it)
#pragma unroll
                        for (int e = 0; e < 8; ++e) ov[it][e] = (bf16)((float)ov[it][e] + (float)rv[0][it][e]);
                }
                if (h == 0) {
                    asm volatile("" ::: "memory");
                    if (has_next) issue_ktile((cnt + 1) & 1, TK);
                }
                if (RES && h + 1 < MI) load_residual(int_tag, h + 1);     // requested before this pass's stores (vmcnt retires in order)
                if (LNK && h == 1 && has_next) load_ln_frags(nn0, nm0);  // the next tile's fragments: into the registers the first two passes freed
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int grow = m0 + wm * WR + h * 32 + it * 8 + lrow;
                    if (INT || (grow < Mrt && col_ok && !(g.dbg & 1)))
                        *reinterpret_cast<bf16x8*>(c_wave + (size_t)(h * 4 + it) * row8 + lane_off) = ov[it];
                }
            }
        };
        if constexpr (F32B) {
            // fp32 out: a lane holds, for output row l15 of each 16-row tile, 4 consecutive columns per accumulator -> 16-byte stores
            if (has_next) issue_ktile((cnt + 1) & 1, TK);
            float* Cb = g.Cf + (int64_t)bz * g.sC;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int grow = m0 + wm * WR + t * 16 + l15;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int c = n0 + wn * 64 + j * 16 + 4 * g4;
                    if (grow < Mrt && c < g.N) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[t][j][e];
                        *reinterpret_cast<f32x4*>(Cb + (int64_t)grow * g.ldc + c) = v;
                    }
                }
            }
        } else {
            if (use_res) { if (interior) epilogue(yes{}, yes{}); else epilogue(yes{}, no{}); }
            else { if (interior) epilogue(no{}, yes{}); else epilogue(no{}, no{}); }
        }
        if (g.tim) { const unsigned long long ts2 = __builtin_amdgcn_s_memtime(); t_main += ts1 - ts0; t_epi += ts2 - ts1; }
        if (!has_next) break;
        pend = (interior && !F32B) ? NSTORE : -1;
        m0 = nm0; n0 = nn0; bz = nbz; ++round;
    }
    if (g.tim && tid == 0) { g.tim[blockIdx.x * 4 + 0] = t_main; g.tim[blockIdx.x * 4 + 1] = t_epi; g.tim[blockIdx.x * 4 + 2] = t_first; g.tim[blockIdx.x * 4 + 3] = ((unsigned long long)(round + 1) << 40) | t_bar; }
}

// --------------------------------------------------------------------------------------------
// Tail kernel: the < 1-round remainder of M (p*256 rows; the ViT's 257 tokens per image leave one 256-row slab after every
// exact number of rounds).  A handful of tiles cannot fill 256 CUs, so this launch is pure latency: with 256 x 64 tiles and three
// stages it took 24 us (16 K-tiles at 1.5 us each) for 0.4 % of the GEMM's work — 7 % of its time.  Hence small tiles and a deep
// pipeline: 64 x 64 tiles (4 waves, 32 x 32 = 2 x 2 MFMA tiles of 16 x 16 each), EIGHT 16 KiB stages, seven K-tiles of operands in flight, every
// wait counted.  Same arithmetic per output element as the main kernel (bias as accumulator init, ascending k, one bf16 rounding,
// round-then-add residual): a row's result does not depend on which kernel produced it.
// --------------------------------------------------------------------------------------------
constexpr int TT = 64;                           // tail tile edge of the default shape
constexpr int TNS = 8;                           // stages of the default shape
__host__ __device__ constexpr int tail_stage(int ttm, int ttn) { return (ttm + ttn) * TK * 2; }                     // 64 x 64: A 8 KiB + W 8 KiB
__host__ __device__ constexpr int tail_lds(int ttm, int ttn, int ns) { return ns * tail_stage(ttm, ttn) + 2560; }   // + bias row (+ LNK: row rstds, column / row fragments)

// TTM x TTN: rows x columns per tile — 64 x 64; 64 x 32 or 32 x 32 for launches with too few 64 x 64 tiles to occupy the chip (two / four times
// the workgroups, each streaming 3/4 / 1/2 of the bytes: these launches are bound by the latency of one workgroup's K loop); NS: pipeline stages
// (8, or 4 = 66 KiB so that TWO workgroups fit a CU when a launch has more tiles than CUs).  The arithmetic per output element does not depend on
// any of them.
// One tile, by the four waves (threads 0-255) of a workgroup: the whole of gemm_tail_kernel, and the remainder rows of a ping-pong launch (there the
// second wave row has ended: s_barrier counts the surviving waves only).
template <int ACT, bool LNK, int TTM, int TTN, int NS>
__device__ __forceinline__ void gemm_tail_tile(const PArgs& g, const int tile, char* smem) {
    constexpr int NI = TTM / 32, NJ = TTN / 32;             // 16 x 16 MFMA tiles per wave (wave = TTM / 2 rows x TTN / 2 columns)
    constexpr int LPT = NI + NJ;                            // LDS-DMA loads per lane per K-tile: TTM / 32 of A, TTN / 32 of W
    constexpr int STG = tail_stage(TTM, TTN);
    static_assert(!LNK || TTM <= 64, "the folded LayerNorm's row values have 64 slots beside the bias row");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int nk = g.K / TK;
    const int m0 = (tile / g.tilesN) * TTM, n0 = (tile % g.tilesN) * TTN;
    const int Mrt = g.m_dev ? min(g.M, __builtin_amdgcn_readfirstlane(*g.m_dev)) : g.M;
    if (m0 >= Mrt) return;                                  // a ragged stage's tiles beyond its device-side row count (before any barrier)
    float* sbias = reinterpret_cast<float*>(smem + NS * STG);
    if (tid < TTN) sbias[tid] = g.bias ? g.bias[min(n0 + tid, g.N - 1)] : 0.f;
    if constexpr (LNK) {                                    // sbias[64..127] row rstds, then the column fragments and the row fragments (16 B each)
        if (tid < 64) {
            const float* st = g.ln_stats + 8 * (int64_t)min(m0 + min(tid, TTM - 1), Mrt - 1);
            sbias[64 + tid] = st[4];
            if (tid < TTN) reinterpret_cast<f32x4*>(sbias + 128)[tid] = *reinterpret_cast<const f32x4*>(g.ln_colsum + 4 * (int64_t)min(n0 + tid, g.N - 1));
            const f32x4 rf = {st[0], st[0], st[1], st[1]};        // compact (-mean hi, lo), (1 / rstd hi, lo) -> the 8 k-slots
            reinterpret_cast<f32x4*>(sbias + 384)[tid] = rf;
        }
    }

    const bf16* a_src[NI]; const bf16* b_src[NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = i * 256 + tid, row = p >> 3, kc = (p & 7) ^ swz(row);
        a_src[i] = g.A + (int64_t)min(m0 + row, Mrt - 1) * g.lda + kc * 8;
    }
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int p = i * 256 + tid, row = p >> 3, kc = (p & 7) ^ swz(row);
        b_src[i] = g.W + (int64_t)min(n0 + row, g.N - 1) * g.K + kc * 8;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem) + wave * 1024;
    auto dma16 = [&](const bf16* ptr, unsigned lds_dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(ptr), "s"(lds_dst) : "memory");
    };
    auto issue = [&](int kt) {                              // LPT loads per lane per K-tile
        const unsigned sb = lds0 + (kt % NS) * STG;
        const int k0 = kt * TK;
#pragma unroll
        for (int i = 0; i < NI; ++i) dma16(a_src[i] + k0, sb + i * 4096);
#pragma unroll
        for (int i = 0; i < NJ; ++i) dma16(b_src[i] + k0, sb + TTM * TK * 2 + i * 4096);
    };
    // The K loop used to be latency-serial per K-tile (4 waves, one per SIMD: counted wait -> barrier release ~80 cycles -> ds_read latency -> a
    // handful of MFMAs): ~590 cycles per K-tile at the 32 x 32 shape against 64 of MFMA issue — and the launches of one image are exactly this loop
    // (tools/latency.py).  Round 4, eight-stage shapes: the K-tiles go through the loop in PAIRS — one counted wait, one barrier, one round of
    // fragment-read latency per TWO K-tiles (the compiler hoists the pair's reads above its MFMAs; NS - 4 K-tiles stay in flight beyond the pair).
    // M = 257: fc2 (K = 4096) 17.9 -> 13.6 us, q|k|v 7.7 -> 6.8, proj 6.2 -> 5.6; one image 1.82 -> 1.67 ms (profiles/r04_small_gemm_pairs.log).
    // A second version that also read the NEXT pair's fragments under the current pair's MFMAs (two register sets, the pair's stages refilled one
    // barrier earlier) was slower than this one (fc2 15.8 us, one image 1.75 ms) and is not kept.  The MFMA order — K-tile kt, then kt + 1, each
    // k-step 0 then 1 — is unchanged: bits identical to every other bf16 GEMM kernel.
    constexpr bool PAIR = NS >= 6;                          // (six stages: the 128 x 64 shape, 24 KiB per stage — two K-tiles in flight beyond the pair)
    for (int kt = 0; kt < (PAIR ? NS - 2 : NS - 1) && kt < nk; ++kt) issue(kt);

    f32x4 acc[NI][NJ];                                      // this wave's (TTM / 2) x (TTN / 2): NI x NJ MFMA tiles of 16 x 16
    auto init_acc = [&]() {
        if constexpr (LNK) {                                // the same fragments, the same instruction as the persistent kernel: identical bits
            f32x4 z;
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = 0.f;
            bf16x8 cfr[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) cfr[j] = ln_frag_lane(reinterpret_cast<const f32x4*>(sbias + 128)[wn * (TTN / 2) + j * 16 + l15], g4);
#pragma unroll
            for (int t = 0; t < NI; ++t) {
                const bf16x8 rfr = ln_frag_lane(reinterpret_cast<const f32x4*>(sbias + 384)[wm * (TTM / 2) + t * 16 + l15], g4);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cfr[j], rfr, z, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float b = sbias[wn * (TTN / 2) + j * 16 + 4 * g4 + e];
#pragma unroll
                    for (int t = 0; t < NI; ++t) acc[t][j][e] = b;
                }
        }
    };
    auto read_frags = [&](int kt, bf16x8 (&af)[2][NI], bf16x8 (&wf)[2][NJ]) {
        const char* Ab = smem + (kt % NS) * STG;
        const char* Bb = Ab + TTM * TK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int t = 0; t < NI; ++t) {
                const int ra = wm * (TTM / 2) + t * 16 + l15;
                af[ks][t] = *reinterpret_cast<const bf16x8*>(Ab + ra * 128 + (((ks * 4 + g4) ^ swz(ra)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int rb = wn * (TTN / 2) + j * 16 + l15;
                wf[ks][j] = *reinterpret_cast<const bf16x8*>(Bb + rb * 128 + (((ks * 4 + g4) ^ swz(rb)) << 4));
            }
        }
    };
    auto mma_ktile = [&](const bf16x8 (&af)[2][NI], const bf16x8 (&wf)[2][NJ]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < NI; ++t)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][j], af[ks][t], acc[t][j], 0, 0, 0);
    };
    if constexpr (PAIR) {
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            // K-tiles kt and kt + 1 have landed for this wave: at most the LPT * min(NS - 4, nk - 2 - kt) younger loads may still fly
            switch (min(NS - 4, nk - 2 - kt)) {
                case 4: wait_vm<LPT * 4>(); break;
                case 3: wait_vm<LPT * 3>(); break;
                case 2: wait_vm<LPT * 2>(); break;
                case 1: wait_vm<LPT>(); break;
                default: wait_vm<0>(); break;
            }
            s_barrier_lgkm();                               // everyone's pieces; everyone is done with the stages of K-tiles kt - 2, kt - 1 ...
            if (kt + NS - 2 < nk) issue(kt + NS - 2);       // ... which are the stages K-tiles kt + NS - 2, kt + NS - 1 go to
            if (kt + NS - 1 < nk) issue(kt + NS - 1);
            if (kt == 0) init_acc();
            bf16x8 af0[2][NI], wf0[2][NJ], af1[2][NI], wf1[2][NJ];
            read_frags(kt, af0, wf0);
            read_frags(kt + 1, af1, wf1);
            mma_ktile(af0, wf0);
            mma_ktile(af1, wf1);
        }
        if (kt < nk) {                                      // an odd number of K-tiles: the last one alone (nothing younger is in flight)
            wait_vm<0>();
            s_barrier_lgkm();
            if (kt == 0) init_acc();
            bf16x8 af0[2][NI], wf0[2][NJ];
            read_frags(kt, af0, wf0);
            mma_ktile(af0, wf0);
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            // K-tile kt has landed for this wave: at most the LPT * min(NS - 2, nk - 1 - kt) younger loads may still fly
            switch (min(NS - 2, nk - 1 - kt)) {
                case 2: wait_vm<LPT * 2>(); break;
                case 1: wait_vm<LPT>(); break;
                default: wait_vm<0>(); break;
            }
            s_barrier_lgkm();                               // everyone's pieces; everyone is done with the stage of K-tile kt - 1 ...
            if (kt + NS - 1 < nk) issue(kt + NS - 1);       // ... which is the stage K-tile kt + NS - 1 goes to
            if (kt == 0) init_acc();
            bf16x8 af0[2][NI], wf0[2][NJ];
            read_frags(kt, af0, wf0);
            mma_ktile(af0, wf0);
        }
    }
    // epilogue: the TTM x TTN tile is transposed through stage 0 (every load has landed; the barrier orders the last reads); rows of 2 * TTN bytes
    s_barrier_lgkm();
    char* stg = smem;
    constexpr int RB = TTN * 2, NSL = TTN / 8;              // bytes / 16-byte slots per staged row
#pragma unroll
    for (int t = 0; t < NI; ++t)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[t][j][e];
                if constexpr (LNK) x *= sbias[64 + wm * (TTM / 2) + t * 16 + l15];
                if (ACT == SETOK_ACT_QUICK_GELU) x = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.45546696f * x));
                else if (ACT == SETOK_ACT_GELU_ERF) x = gelu_erf_fast(x);
                v[e] = (bf16)x;
            }
            const int row = wm * (TTM / 2) + t * 16 + l15;          // columns wn * TTN / 2 + j * 16 + 4 * g4 .. + 3
            const int sl = wn * (NSL / 2) + j * 2 + (g4 >> 1);
            *reinterpret_cast<bf16x4*>(stg + row * RB + ((sl ^ (row & (NSL - 1))) << 4) + 8 * (g4 & 1)) = v;
        }
    s_barrier_lgkm();
    const int slot = tid & (NSL - 1), col = n0 + slot * 8;
    constexpr int RPP = 256 / NSL;                          // rows per pass of the 256 threads
    constexpr int PASSES = TTM > RPP ? TTM / RPP : 1;
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
        const int row = it * RPP + tid / NSL;
        if (row >= TTM) continue;
        const int grow = m0 + row;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + row * RB + ((slot ^ (row & (NSL - 1))) << 4));
        if (grow < Mrt && col < g.N && !(g.dbg & 1)) {
            if (g.res && !(g.dbg & 2)) {
                const bf16x8 rv = *reinterpret_cast<const bf16x8*>(g.res + (int64_t)grow * g.ldc + col);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)v[e] + (float)rv[e]);
            }
            *reinterpret_cast<bf16x8*>(g.C + (int64_t)grow * g.ldc + col) = v;
        }
    }
}

template <int ACT, bool LNK = false, int TTM = 64, int TTN = 64, int NS = TNS>
__global__ __launch_bounds__(256) void gemm_tail_kernel(PArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_tail_tile<ACT, LNK, TTM, TTN, NS>(g, blockIdx.x, smem);
    if constexpr (TTM == 128) {
        // the 128 x 64 shape is launched for whole 128-row tiles that fill the chip in ONE round (eight images' fc2: 2048 x 1024 -> 256 tiles); the few rows behind
        // them (8 of 2056) are finished here as 32 x 32 tiles by workgroups 0 .. rem_tiles - 1, like the ping-pong kernel's remainder
        if (g.rem_rows > 0 && (int)blockIdx.x < g.rem_tiles) {
            s_barrier_lgkm();                               // everyone is out of the big tile's staging rows
            PArgs t = g;
            t.A = g.A + (int64_t)g.M * g.lda;
            t.C = g.C + (int64_t)g.M * g.ldc;
            if (g.res) t.res = g.res + (int64_t)g.M * g.ldc;
            t.M = g.rem_rows; t.tilesM = (g.rem_rows + 31) / 32; t.tilesN = g.N / 32; t.m_dev = nullptr; t.rem_rows = 0;
            gemm_tail_tile<ACT, LNK, 32, 32, TNS>(t, blockIdx.x, smem);
        }
    }
}

// Shape of the small-tile launch for a problem of M x N outputs (host side): how many workgroups the default 64 x 64 tiles give against the CUs.
// SETOK_GEMM_SMALL_SHAPE=0 keeps the default everywhere (A/B runs).
struct TailShape { int ttm, ttn, ns; };
// (`whole`: the caller launches the WHOLE problem with this shape — not a remainder — has no folded LayerNorm and no device-side row count: the 128 x 64 shape may be picked)
static TailShape tail_shape_for(int M, int N, int ncu, bool whole = false) {
    static const bool off = [] { const char* e = getenv("SETOK_GEMM_SMALL_SHAPE"); return e && e[0] == '0'; }();
    if (off) return {64, 64, TNS};
    const int t64 = cdiv(M, TT) * cdiv(N, 64);
    // Round 5: more 64 x 64 tiles than CUs, but the whole 128-row tiles of 64 columns fit ONE round and the rows behind them are few: 128 x 64 tiles (twice the MFMAs
    // per K-tile under the same latency chain) + the remainder inside the launch.  Eight images: fc2 59 -> 47 us, proj 19 -> 10 us, the step 5.0 -> 4.58 ms; with HALF
    // a round of such tiles (four images: 128) the two-per-CU 64 x 64 launch stays ahead (3.65 against 4.17 ms).  SETOK_GEMM_SHAPE_128=0: off (A/B).
    static const bool s128 = [] { const char* e = getenv("SETOK_GEMM_SHAPE_128"); return !(e && e[0] == '0'); }();
    if (s128 && whole && N % 64 == 0 && t64 > ncu) {
        const int main_tiles = (M / 128) * (N / 64), rem = M % 128;
        if (main_tiles > ncu / 2 && main_tiles <= ncu && (rem == 0 || cdiv(rem, 32) * (N / 32) <= main_tiles)) return {128, 64, 6};
    }
    if (t64 * 4 <= ncu && N % 32 == 0) return {32, 32, TNS};          // a quarter of the chip: four times the workgroups
#ifndef SETOK_NO_TWO_PER_CU_3232
    if (N % 32 == 0 && cdiv(M, 32) * cdiv(N, 32) <= 2 * ncu) return {32, 32, TNS};   // round 4: up to TWO 32 x 32 workgroups per CU (66.5 KiB of LDS each) in one round
#endif
    if (t64 * 2 <= ncu && N % 32 == 0) return {64, 32, TNS};          // half the chip: twice the workgroups
    if (t64 > ncu && t64 <= 2 * ncu) return {64, 64, 4};              // between one and two rounds: two workgroups per CU, ONE round
    return {64, 64, TNS};
}

template <int TTM, int TTN, int NS>
static int launch_tail_shape(hipStream_t s, const PArgs& g, int act, hipEvent_t e0, hipEvent_t e1) {
    constexpr bool LN_OK = TTM <= 64;                       // (the 128 x 64 shape has no folded-LayerNorm form)
    static SetokDeviceOnce once;
    if (!once.run([] {
            bool ok = true;
            const void* fns[] = {(const void*)gemm_tail_kernel<0, false, TTM, TTN, NS>, (const void*)gemm_tail_kernel<1, false, TTM, TTN, NS>,
                                 (const void*)gemm_tail_kernel<2, false, TTM, TTN, NS>};
            for (const void* f : fns) ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, tail_lds(TTM, TTN, NS)) == hipSuccess;
            if constexpr (LN_OK) {
                const void* lfns[] = {(const void*)gemm_tail_kernel<0, true, TTM, TTN, NS>, (const void*)gemm_tail_kernel<1, true, TTM, TTN, NS>,
                                      (const void*)gemm_tail_kernel<2, true, TTM, TTN, NS>};
                for (const void* f : lfns) ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, tail_lds(TTM, TTN, NS)) == hipSuccess;
            }
            return ok; }))
        return setok_fail(SETOK_ELAUNCH, "setok_linear: cannot raise the dynamic LDS limit");
    constexpr int LDS = tail_lds(TTM, TTN, NS);
    const int grid = g.tilesM * g.tilesN;
    const dim3 gr(grid), bl(256);
    if (g.ln_stats) {
        if constexpr (LN_OK) {
            if (act == SETOK_ACT_NONE) setok_launch(gemm_tail_kernel<0, true, TTM, TTN, NS>, gr, bl, LDS, s, e0, e1, g);
            else if (act == SETOK_ACT_QUICK_GELU) setok_launch(gemm_tail_kernel<1, true, TTM, TTN, NS>, gr, bl, LDS, s, e0, e1, g);
            else setok_launch(gemm_tail_kernel<2, true, TTM, TTN, NS>, gr, bl, LDS, s, e0, e1, g);
        } else return setok_fail(SETOK_EINVAL, "setok_linear_ln: no %d x %d small-tile kernel with a folded LayerNorm", TTM, TTN);
    } else if (act == SETOK_ACT_NONE) setok_launch(gemm_tail_kernel<0, false, TTM, TTN, NS>, gr, bl, LDS, s, e0, e1, g);
    else if (act == SETOK_ACT_QUICK_GELU) setok_launch(gemm_tail_kernel<1, false, TTM, TTN, NS>, gr, bl, LDS, s, e0, e1, g);
    else setok_launch(gemm_tail_kernel<2, false, TTM, TTN, NS>, gr, bl, LDS, s, e0, e1, g);
    SETOK_CHECK_LAUNCH("setok_linear(tail)");
    return SETOK_OK;
}

// g.tilesM / g.tilesN must have been computed for `sh` (cdiv(M, sh.ttm), cdiv(N, sh.ttn)).
int launch_tail(hipStream_t s, const PArgs& g, int act, hipEvent_t e0, hipEvent_t e1, TailShape sh) {
    if (sh.ttm == 128) return launch_tail_shape<128, 64, 6>(s, g, act, e0, e1);
    if (sh.ttm == 32) return launch_tail_shape<32, 32, TNS>(s, g, act, e0, e1);
    if (sh.ttn == 32) return launch_tail_shape<64, 32, TNS>(s, g, act, e0, e1);
    if (sh.ns == 4) return launch_tail_shape<64, 64, 4>(s, g, act, e0, e1);
    return launch_tail_shape<64, 64, TNS>(s, g, act, e0, e1);
}

// --------------------------------------------------------------------------------------------
// The PING-PONG form of the persistent kernel (round 3) — same tiles, same LDS image, same fragment / accumulator layout, same arithmetic per
// output element (the results are bit-identical to gemm_persist_kernel's), a different SCHEDULE of the K loop:
//   * a K-tile is four phases per wave — the quadrants (row half i, column half j) of its 128 x 64 output in the order (0,0) (0,1) (1,1) (1,0) —
//     each a LOAD slot (fragment reads, one quarter-tile of LDS-DMA requests) and an MFMA slot (16 MFMAs), every slot closed by s_barrier;
//   * the two wave rows run ONE SLOT APART (waves 4-7 execute one extra barrier in front of every tile's K loop, waves 0-3 one behind it — per TILE
//     since round 5, so that both rows reach the epilogue together; PP_RESYNC): on every SIMD one wave multiplies while its partner reads
//     fragments and stands at the address unit — the matrix pipe never has two claimants, and the lock-step in which gemm_persist_kernel's two
//     waves of a SIMD both fetch and then both multiply (matrix pipe busy 47 % of a launch) is gone by construction;
//   * operands arrive in QUARTER tiles (A rows 0-127 / 128-255, W rows 0-127 / 128-255 of a stage: 16 KiB = 2 requests per lane, full 128-byte
//     rows), one quarter per phase: phase 0: A_lo(t+1), phase 1: A_hi(t+1), phase 2: W_lo(t+2), phase 3: W_hi(t+2) — a quarter of the current stage
//     is refilled as soon as its last reader of this K-tile has retired (W after phase 1: the column fragments of phase 3 stay in registers;
//     A after phase 2) — and the stream runs across tile boundaries; the one wait of a K-tile is a counted vmcnt one slot before the next
//     K-tile's first reader.  The A quarters of a tile's SECOND K-tile go out before the previous tile's stores, so no request queues behind a store.
// Same-box A/B against gemm_persist_kernel (tools/micro/gemm_pp2.hip, plain GEMM, random operands): 65792 x 3072 x 1024 +7 %, 65792 x 4096 x 1024
// +10 %, 8192^3 +3 %; on all-zero operands (clock not power-managed) +19 % / +15 % / +18 %: at 1.8 PFLOP/s the K-tile period is the per-CU LDS-DMA
// rate (64 KiB per ~1.1 us).  Used for launches without a residual and with N a multiple of 256 (q|k|v, fc1, the head's / projector's plain
// Linears); the residual launches stay on gemm_persist_kernel (no gain at K = 4096, and no registers for the residual rows beside 96 fragment
// registers).  SETOK_GEMM_PP=0 switches it off (A/B runs).
// --------------------------------------------------------------------------------------------
#ifdef PP_TIMING             // -DPP_TIMING + SETOK_GEMM_TIMING=1: wave 0's cycles per tile in the K loop / the epilogue ("vmcnt waits" column: the accumulator start)
#define PP_TIMING_ON 1
#else
#define PP_TIMING_ON 0
#endif
#ifndef PP_GM
#define PP_GM 8            // M-tiles per group of the tile order (cfg2, same box: 4 = 8; 16: -1 %; 32: -4.7 %)
#endif
#ifndef PP_NT_STORE
#define PP_NT_STORE 1       // 1: the C tile leaves as streaming stores (`nt`); 0: ordinary stores (rounds 1-4).  Round 5, same box, alternating runs: q|k|v +4...5 %,
#endif                      // out-projection + residual +7 %, fc1 + quick_gelu +3 %, cfg2 step -1.7 %: the tile's 128 KiB no longer pass through the L2 the operand panels live in
// cache-policy bits of the ping-pong kernel's memory instructions as A/B switches (the vendor's tuned kernels carry such a digit per operand):
// a bit mask 1 = sc0, 2 = nt, 4 = sc1 for the A requests (PP_PA), the W requests (PP_PW) and the C stores (PP_PC)
#ifndef PP_PA
#define PP_PA 0
#endif
#ifndef PP_PW
#define PP_PW 0
#endif
#ifndef PP_PC
#define PP_PC 2
#endif
#define PP_POLSTR_0 ""
#define PP_POLSTR_1 " sc0"
#define PP_POLSTR_2 " nt"
#define PP_POLSTR_3 " sc0 nt"
#define PP_POLSTR_4 " sc1"
#define PP_POLSTR_5 " sc0 sc1"
#define PP_POLSTR_6 " sc1 nt"
#define PP_POLSTR_7 " sc0 sc1 nt"
#define PP_POLCAT(n) PP_POLSTR_##n
#define PP_POLSTR(n) PP_POLCAT(n)
#define PP_POL_A PP_POLSTR(PP_PA)
#define PP_POL_W PP_POLSTR(PP_PW)
#define PP_POL_C PP_POLSTR(PP_PC)
#ifndef PP_PC_RES
#define PP_PC_RES PP_PC     // the same for the launches with a residual (proj / fc2: 135 MB of output per launch against 404 / 539 MB — the one class write-through stores did not slow down)
#endif
#define PP_POL_C_RES PP_POLSTR(PP_PC_RES)
// PP_NT_STORE's inline-assembly stores carry hand-placed wait states for the gfx940+ "VALU write of the data registers of a > 64-bit store" hazard,
// PP_MERGE_REM relies on s_barrier counting only the surviving waves of a workgroup, and the K loop on v_mfma_f32_16x16x32_bf16 / global_load_lds_dwordx4 /
// one in-order vmcnt for loads and stores: all of it is gfx950 behaviour the compiler cannot check for another target.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "gemm_persist.hip is written for gfx950 (MI355X) only: the hand-scheduled waits, hazards and barriers in it are not valid elsewhere"
#endif
#ifndef PP_PHASES
#define PP_PHASES 2         // phases (LOAD slot + MFMA slot) per K-tile: 4 = the quadrants, 16 MFMAs each (rounds 3-5); 2 = the row halves, 32 MFMAs each (round 6: same box,
                            // alternating builds, cfg2 41.63 -> 40.77 ms, GEMM fraction 0.486 -> 0.4985, every class +1.5 ... +3 %; identical bits)
#endif
#ifndef PP_PH2_SPLIT
#define PP_PH2_SPLIT 0      // PP_PHASES == 2 only. 1: LOAD A's second k-step A fragments are read BEHIND the slot's barrier, under the first 16 MFMAs (a shorter LOAD A slot) — A/B
#endif
#ifndef PP_PH2_READS_FIRST
#define PP_PH2_READS_FIRST 0 // PP_PHASES == 2 only. 1: a LOAD slot's LDS-DMA requests go out BEHIND its fragment reads — A/B
#endif
#ifndef PP_RESYNC
#define PP_RESYNC 1         // 1: the wave rows' one-slot offset is set up and taken back per tile (both epilogues at the same time); 0: once per launch (rounds 3-4)
#endif
#ifndef PP_LN_PACK
#define PP_LN_PACK 1        // 1: the folded LayerNorm's start values are 6 loads per wave — the column fragments in EVERY lane group, row block t's fragment + rstd as ONE
#endif                      // 16-byte load in lane group t / 2, the start MFMA of block t masking the other groups; 0: 14 loads, fragments in lanes 0-15 (rounds 3-4)
#ifndef PP_MERGE_REM
#define PP_MERGE_REM 1      // 1: a launch finishes its remainder rows itself (see the end of gemm_pp_body); 0: they are a launch of the small-tile kernel (rounds 3-4)
#endif
#ifndef PP_NO_EDGE
#define PP_NO_EDGE 1        // 1: the host hands this kernel whole 256-row tiles only (the remainder rows go to the small-tile kernel, a device-side M to gemm_persist_kernel)
#endif
template <int ACT, bool LNK, bool RESK, int GRP>
__device__ __forceinline__ void gemm_pp_body(const PArgs& g, char* smem) {
    static_assert(!(LNK && RESK), "the folded LayerNorm feeds q|k|v / fc1: no residual");
    constexpr int QT = 16 * 1024;
    constexpr bool SWG = ACT == SETOK_ACT_SWIGLU_PAIRS;       // round 6: act_fn(gate) * up in the epilogue; the tile's 256 columns are 128 (gate, up) pairs -> 128 output columns
    static_assert(!(SWG && (LNK || RESK)), "the SwiGLU epilogue has no folded LayerNorm and no residual");
    constexpr int NSTORE = SWG ? 8 : 16;                    // 16-byte stores per lane and tile
    // entries of the vector-memory queue a tile's epilogue puts BEHIND the A quarters of the next tile's K-tile 1: the next tile's start values
    // (bias / LN fragments: 4 / 5 ordinary loads) and, with a residual, the residual rows of passes 1-3 (12 loads); + the 16 stores
    constexpr int NAUX = (LNK ? (PP_LN_PACK ? 6 : 14) : 4) + (RESK ? 12 : 0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3;                                // wave row == GRP
    const int l15 = lane & 15, g4 = lane >> 4;
    const int nk = g.K / TK;
    const int Mrt = (!PP_NO_EDGE && g.m_dev) ? min(g.M, __builtin_amdgcn_readfirstlane(*g.m_dev)) : g.M;
    const int tilesMrt = (!PP_NO_EDGE && g.m_dev) ? (Mrt + TM - 1) / TM : g.tilesM;
    const int num_tiles = tilesMrt * g.tilesN;
    const int G = gridDim.x;
    auto tile_of = [&](int round, int& m0, int& n0) -> bool {
        int L;
        if ((G & 7) == 0) L = round * G + (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
        else L = round * G + blockIdx.x;
        if (L >= num_tiles) return false;
        constexpr int GM = PP_GM;
        const int per = GM * g.tilesN, group = L / per, first_m = group * GM;
        const int gm = min(tilesMrt - first_m, GM), in = L - group * per;
        m0 = (first_m + in % gm) * TM;
        n0 = (in / gm) * 256;
        return true;
    };
    // LDS-DMA source: piece p = 512 i + tid of a quarter -> row 64 i + (tid >> 3), 16-byte slot (tid & 7) ^ swizzle(row); the swizzle does not
    // depend on i, so ONE lane offset per operand serves every piece and the row advance goes into the wave-uniform base.  Only a tile that
    // crosses the end of M clamps rows (per-lane offsets formed on the fly there).
    const int prow = tid >> 3;
    const unsigned kc16 = (unsigned)(((tid & 7) ^ swz(prow)) << 4);
    const unsigned a_off = (unsigned)prow * (unsigned)(g.lda * 2) + kc16;
    const unsigned w_off = (unsigned)prow * (unsigned)(g.K * 2) + kc16;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem) + wave * 1024;
    // (PP_POL_A / PP_POL_W: cache-policy bits of the operand requests — "", " nt", " sc0", " sc1" and their combinations — as A/B switches)
    auto dma16_pol = [&](auto is_w, const char* base, unsigned off, unsigned lds_dst) {
        unsigned keep;
        const unsigned long long b64 = (unsigned long long)base;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);
        const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));
        const unsigned long long sb64 = (unsigned long long)lo | ((unsigned long long)hi32 << 32);
        if constexpr (decltype(is_w)::value)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" PP_POL_W "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(off), "s"(sb64), "s"(lds_dst) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" PP_POL_A "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(off), "s"(sb64), "s"(lds_dst) : "memory");
    };
    auto dma16 = [&](const char* base, unsigned off, unsigned lds_dst) { dma16_pol(std::false_type{}, base, off, lds_dst); };
    auto dma16w = [&](const char* base, unsigned off, unsigned lds_dst) { dma16_pol(std::true_type{}, base, off, lds_dst); };
    int m0, n0, nm0 = 0, nn0 = 0, round = 0;
    if (!tile_of(0, m0, n0)) return;
    bool has_next = tile_of(1, nm0, nn0);
    int cnt = 0;                                            // K-tiles consumed before this tile: K-tile u of the tile lives in stage (cnt + u) & 1
    // Per tile, once: the byte addresses of its first A row / first W row (wave-uniform 64-bit values) and whether it crosses the end of M; a
    // request then costs a handful of scalar adds (computed per request from m0 / n0 the 64-bit multiplies were ~35 scalar instructions each,
    // eight times per K-tile, inside the load slots).
    const size_t a_half = (size_t)128 * (size_t)(g.lda * 2), a_piece = (size_t)64 * (size_t)(g.lda * 2);
    const size_t w_half = (size_t)128 * (size_t)(g.K * 2), w_piece = (size_t)64 * (size_t)(g.K * 2);
    const char *a_cur, *w_cur, *a_nxt = nullptr, *w_nxt = nullptr;
    bool edge_cur, edge_nxt = false;
    int lim_cur, lim_nxt = 0;                               // last valid row inside an edge tile
    auto tile_ptrs = [&](int tm0, int tn0, const char*& a, const char*& w, bool& edge, int& lim) {
        a = reinterpret_cast<const char*>(g.A + (int64_t)tm0 * g.lda);
        w = reinterpret_cast<const char*>(g.W + (int64_t)tn0 * g.K);
        edge = tm0 + TM > Mrt;
        lim = Mrt - 1 - tm0;
    };
    tile_ptrs(m0, n0, a_cur, w_cur, edge_cur, lim_cur);
    if (has_next) tile_ptrs(nm0, nn0, a_nxt, w_nxt, edge_nxt, lim_nxt);
    // quarter q (0 = A_lo, 1 = A_hi, 2 = W_lo, 3 = W_hi) of K-tile u (< nk) of a tile given by its pointers -> stage `stage`
    auto issue_at = [&](int q, int u, int stage, const char* at, const char* wt, bool edge, int lim) {
        const unsigned sb = lds0 + stage * STAGE + (q >> 1) * BOFF + (q & 1) * QT;
        if (q >> 1) {
            const char* base = wt + (q & 1) * w_half + (size_t)u * 128;
            dma16w(base, w_off, sb);
            dma16w(base + w_piece, w_off, sb + 8192);
        } else if (PP_NO_EDGE || !edge) {
            const char* base = at + (q & 1) * a_half + (size_t)u * 128;
            dma16(base, a_off, sb);
            dma16(base + a_piece, a_off, sb + 8192);
        } else {                                            // the tile crosses the end of M: rows clamped to the last one (never stored)
            const char* base = at + (size_t)u * 128;
            const int r0 = (q & 1) * 128;
            dma16(base, (unsigned)min(r0 + prow, lim) * (unsigned)(g.lda * 2) + kc16, sb);
            dma16(base, (unsigned)min(r0 + 64 + prow, lim) * (unsigned)(g.lda * 2) + kc16, sb + 8192);
        }
    };
    // ... of the CURRENT tile's K-tile u; u >= nk runs into the next tile's K-tile u - nk (the stream does not stop at tile boundaries)
    auto issue_quarter = [&](int q, int u) {
        if (u < nk) issue_at(q, u, (cnt + u) & 1, a_cur, w_cur, edge_cur, lim_cur);
        else if (has_next) issue_at(q, u - nk, (cnt + u) & 1, a_nxt, w_nxt, edge_nxt, lim_nxt);
    };

    f32x4 acc[8][4];
    bf16x8 A0[2][4], A1[2][4], W0[2][2], W1[2][2];          // [k-step][tile]: both row halves and both column halves have registers of their own
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
                const int r = wn * 64 + jh * 32 + j * 16 + l15;
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
    auto bar = [&]() { asm volatile("s_barrier" ::: "memory"); };

    // what the accumulators of a tile start from: the bias (4 consecutive columns per accumulator register quad), or the folded LayerNorm's
    // operand fragments — the same bits gemm_persist_kernel feeds its start MFMAs, fetched lane by lane instead of three loads + 40 ds_bpermute
    // (measured: no difference in time; kept because it is the shorter code and needs no cross-lane traffic at the tile start)
    f32x4 nbv[4];
    float nrs[2], ers[2];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x4 ncd[4]; f32x2 nrd[8];                             // the operand fragments of the start MFMAs as the lanes hold them (k-slots 0-7 live in lanes 0-15)
    f32x4 npk[2];                                           // PP_LN_PACK: words 0-2 of the statistics records of row blocks 2 g4, 2 g4 + 1 (fragment, fragment, rstd)
    // The loads are inline assembly and the wait for them is explicit (start_ready): as ordinary loads the compiler waited for them at the loop
    // header with the counts of the kernel-entry path (`vmcnt(0)` for the last one) — on the loop's back edge that is a full drain of the
    // previous tile's sixteen stores, ~2.5 k ticks per tile.  
    auto ld16 = [&](f32x4& dst, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); };
    auto ld8 = [&](f32x2& dst, const float* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); };
    auto ld4 = [&](float& dst, const float* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); };
    auto load_start = [&](int n0_, int m0_) {
        if constexpr (LNK) {
            // Every lane fetches its own operand registers: lanes 0-15 the fragments, the others zeros (one 256-byte line of zeros for all of
            // them) — 12 loads instead of 3, and the tile start has no cross-lane traffic (40 ds_bpermute + 48 selects per wave before).
#if PP_LN_PACK
            // Round 5 (second half): 6 loads instead of 14 in an epilogue whose length is its vector-memory instruction count.  The start MFMA contracts over 32
            // k-slots of which 8 carry the rank-2 correction: the COLUMN fragment sits in all four lane groups (the same address for the four lanes of a column),
            // the ROW fragment of block t only in group t / 2 (the MFMA of block t zeroes the other groups' operand) — the same eight products, and one 16-byte
            // load per row brings fragment and rstd (setok_row_stats writes rstd as word 2 of the record as well).
#pragma unroll
            for (int j = 0; j < 4; ++j) ld16(ncd[j], g.ln_colsum + 4 * (int64_t)(n0_ + wn * 64 + j * 16 + l15));
#pragma unroll
            for (int i = 0; i < 2; ++i) ld16(npk[i], g.ln_stats + 8 * (int64_t)(m0_ + GRP * 128 + (2 * g4 + i) * 16 + l15));
#else
            const bool own = g4 == 0;
            const float* zr = g.zero_bias;
#pragma unroll
            for (int j = 0; j < 4; ++j) ld16(ncd[j], own ? g.ln_colsum + 4 * (int64_t)(n0_ + wn * 64 + j * 16 + l15) : zr + 4 * l15);
#pragma unroll
            for (int t = 0; t < 8; ++t) ld8(nrd[t], own ? g.ln_stats + 8 * (int64_t)(m0_ + GRP * 128 + t * 16 + l15) : zr + 2 * l15);
#pragma unroll
            for (int i = 0; i < 2; ++i) ld4(nrs[i], g.ln_stats + 8 * (int64_t)(m0_ + GRP * 128 + (2 * g4 + i) * 16 + l15) + 4);
#endif
        } else {
            const float* bp = (g.bias ? g.bias + n0_ : g.zero_bias) + (g.bias ? wn * 64 : 0) + 4 * g4;
#pragma unroll
            for (int j = 0; j < 4; ++j) ld16(nbv[j], bp + (g.bias ? j * 16 : 0));
        }
    };
    // ... and the wait: behind the start values of a tile the queue holds the previous tile's 16 stores (+ with a residual the rows of passes 2 and 3,
    // 8 loads); at the kernel's entry the stream start has already waited for them.  The values pass through
    // the statement, so nothing that reads them can be scheduled above it.
    constexpr int NBEHIND = NSTORE + (RESK ? 8 : 0);
    auto start_ready = [&]() {
        if constexpr (LNK)
#if PP_LN_PACK
            asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(ncd[0]), "+v"(ncd[1]), "+v"(ncd[2]), "+v"(ncd[3]), "+v"(npk[0]), "+v"(npk[1]) : [n] "n"(NBEHIND) : "memory");
#else
            asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(ncd[0]), "+v"(ncd[1]), "+v"(ncd[2]), "+v"(ncd[3]), "+v"(nrd[0]), "+v"(nrd[1]), "+v"(nrd[2]), "+v"(nrd[3]),
                         "+v"(nrd[4]), "+v"(nrd[5]), "+v"(nrd[6]), "+v"(nrd[7]), "+v"(nrs[0]), "+v"(nrs[1]) : [n] "n"(NBEHIND) : "memory");
#endif
        else
            asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(nbv[0]), "+v"(nbv[1]), "+v"(nbv[2]), "+v"(nbv[3]) : [n] "n"(NBEHIND) : "memory");
    };

    // (Round 5: the 14 start-value loads per wave of the folded LayerNorm — 112 of a tile's ~270 vector-memory instructions in an epilogue that is bound
    //  by the CU's address unit — were replaced by three LDS-DMA pieces per wave into the wave's idle epilogue staging, read back in pass 0.  Two
    //  findings: a counted vmcnt wait alone does NOT make a wave's own LDS-DMA data readable — it takes the barrier behind it that the K loop has
    //  anyway (without it: old bytes now and then) — and with that barrier the change is worth nothing: 42.98 against 42.85 ms.  Not kept.)
    // ---- start of the stream: the first tile's start values, K-tile 0 entirely + the W quarters of K-tile 1 --------------------------------------
    load_start(n0, m0);
    issue_quarter(0, 0); issue_quarter(1, 0); issue_quarter(2, 0); issue_quarter(3, 0); issue_quarter(2, 1); issue_quarter(3, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    bar();
#if !PP_RESYNC
    if (GRP == 1) bar();                                    // the second wave row runs one slot behind
#endif
    int ahead = 0;                                          // 1: the A quarters of this tile's K-tile 1 went out in the previous tile's epilogue, ahead of
                                                            // exactly NSTORE + NAUX other entries of the vector-memory queue; 2: ahead of an unknown number
#ifdef PP_TIMING
    unsigned long long t_init = 0, t_k = 0, t_epi = 0;
    unsigned long long t_kt0 = 0, t_kt1 = 0, t_kt2 = 0, t_kt3 = 0, t_ktr = 0, t_ktl = 0;     // K-tiles 0..3, the middle ones, the last one
    unsigned long long t_p0 = 0, t_p1 = 0, t_p2 = 0, t_p3 = 0, t_pre = 0;                    // epilogue: before pass 0, passes 0..3
    unsigned long long t_s[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                      // the eight barrier-to-barrier slots of a tile's LAST K-tile
#ifndef PPT_KT
#define PPT_KT (nk - 1)       // which K-tile of a tile the slot stamps are taken in (-DPPT_KT=6: a steady-state one)
#endif
#define PPT_SLOT(i) if (kt == PPT_KT) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); t_s[i] += now_ - tsl; tsl = now_; }
#else
#define PPT_SLOT(i)
#endif
    for (;;) {
#ifdef PP_TIMING
        const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- accumulator start --------------------------------------------------------------------------------------------------------------------
        start_ready();
#ifdef PP_ABL_NOINIT
        if constexpr (false) {
#else
        if constexpr (LNK) {
#endif
#if PP_LN_PACK
            ers[0] = npk[0][2]; ers[1] = npk[1][2];
#else
            ers[0] = nrs[0]; ers[1] = nrs[1];
#endif
            asm volatile("" : "+v"(ers[0]), "+v"(ers[1]));   // values of THIS point: without it the epilogue's first use waits `vmcnt(0)` — for the operand DMA in flight
            f32x4 z;
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
#if PP_LN_PACK
                const bool mine = g4 == (t >> 1);                                        // this lane group carries row block t's fragment
                const float f0 = mine ? npk[t & 1][0] : 0.f, f1 = mine ? npk[t & 1][1] : 0.f;
                const f32x4 v = {f0, f0, f1, f1};
#else
                const f32x4 v = {nrd[t][0], nrd[t][0], nrd[t][1], nrd[t][1]};            // (-mean hi, lo) twice, (1 / rstd hi, lo) twice
#endif
                const bf16x8 rfr = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ncd[j]), rfr, z, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][j] = nbv[j];          // bias first, then the products in ascending k: gemm_persist_kernel's order
        }
#ifdef PP_TIMING
        const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
#if PP_RESYNC
        // The one-slot offset between the wave rows is set up PER TILE and taken back at the tile's end (row 0's closing barrier below), so that both
        // rows run their epilogues at the same time.  Round 5: with the offset set up once per launch, row 1's last barrier of a tile paired with
        // row 0's FIRST barrier of the next tile — row 1 stood at it through the whole of row 0's epilogue, and row 0 then stood at its second
        // barrier through the whole of row 1's: the two epilogues of every tile ran one after the other with the matrix pipe idle (s_memtime:
        // row 0's first K-tile of a tile 7-17 k ticks against 3.2 k for any other, row 1's last slot 4-12 k against 0.45 k).
        if (GRP == 1) bar();
#endif
        for (int kt = 0; kt < nk; ++kt) {
#ifdef PP_TIMING
            const unsigned long long tkt0 = __builtin_amdgcn_s_memtime();
            unsigned long long tsl = tkt0;
#endif
            const char* T = smem + ((cnt + kt) & 1) * STAGE;
            const bool skipA = ahead != 0 && kt == 0;        // (uniform)
#if PP_PHASES == 2
            // ---- two phases of 32 MFMAs (round 6).  A slot of 16 MFMAs is 256 cycles of matrix-pipe time; the partner row's LOAD slot beside it — two LDS-DMA requests,
            // up to eight fragment reads, their latency, the barrier — measures ~360, of which ~165 do not depend on how much the slot loads.  With the row
            // halves as phases — (0,0)+(0,1), then (1,1)+(1,0) — a LOAD slot carries twice the requests and reads against 512 cycles of MFMAs: the fixed part is paid
            // four times per K-tile instead of eight.  Same products into the same accumulators in the same order (k-step 0, then 1): identical bits.
            // LOAD A: both A quarters of K-tile kt + 1 -> the other stage; every W fragment of this K-tile and the first row half's A fragments
#if !PP_PH2_READS_FIRST
            if (!skipA) { issue_quarter(0, kt + 1); issue_quarter(1, kt + 1); }
#endif
#if PP_PH2_SPLIT
            rd_w(T, 0, W0); rd_w(T, 1, W1); rd_a(T, 0, 0, A0[0]);
            lgk0();                                         // the last readers of this stage's W quarters retire BEFORE the barrier: LOAD B refills them
            bar();
            PPT_SLOT(0)
            rd_a(T, 0, 1, A0[1]);                           // lands under the first 16 MFMAs (this stage's A quarters are refilled a K-tile later: no hazard)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mma8(0, 0, A0[0], W0[0]);
            mma8(0, 1, A0[0], W1[0]);
            lgk0();
            mma8(0, 0, A0[1], W0[1]);
            mma8(0, 1, A0[1], W1[1]);
            __builtin_amdgcn_s_setprio(0);
#else
            rd_w(T, 0, W0); rd_w(T, 1, W1); rd_a(T, 0, 0, A0[0]); rd_a(T, 0, 1, A0[1]);
#if PP_PH2_READS_FIRST
            __builtin_amdgcn_sched_barrier(0);
            if (!skipA) { issue_quarter(0, kt + 1); issue_quarter(1, kt + 1); }     // A/B: the requests BEHIND the fragment reads (their latency under the address unit's queue)
#endif
            lgk0();                                         // the last readers of this stage's W quarters retire BEFORE the barrier: LOAD B refills them
            bar();
            PPT_SLOT(0)
            __builtin_amdgcn_s_setprio(1);
            mma8(0, 0, A0[0], W0[0]);
            mma8(0, 0, A0[1], W0[1]);
            mma8(0, 1, A0[0], W1[0]);
            mma8(0, 1, A0[1], W1[1]);
            __builtin_amdgcn_s_setprio(0);
#endif
            bar();
            PPT_SLOT(1)
            // LOAD B: both W quarters of K-tile kt + 2 -> this stage; the second row half's A fragments; the one wait of the K-tile (as in the four-phase form:
            // row 1 at the end of its LOAD slot, row 0 at the end of its MFMA slot — the same barrier)
#if !PP_PH2_READS_FIRST
            issue_quarter(2, kt + 2); issue_quarter(3, kt + 2);
#endif
            rd_a(T, 1, 0, A1[0]); rd_a(T, 1, 1, A1[1]);    // (never behind the barrier: the other row's next LOAD A refills this stage's A quarters while this row multiplies)
#if PP_PH2_READS_FIRST
            __builtin_amdgcn_sched_barrier(0);
            issue_quarter(2, kt + 2); issue_quarter(3, kt + 2);
#endif
            lgk0();                                         // likewise this stage's A quarters (refilled by the next K-tile's LOAD A, or the epilogue)
            const bool issued = kt + 2 < nk || has_next;    // the two W quarters of this K-tile exist
            auto wait_next = [&]() {
                if (!issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (skipA && ahead == 1) wait_vm<4 + NSTORE + NAUX>();
                else wait_vm<4>();
            };
            if (GRP == 1) wait_next();
            bar();
            PPT_SLOT(2)
            __builtin_amdgcn_s_setprio(1);
            mma8(1, 1, A1[0], W1[0]);
            mma8(1, 1, A1[1], W1[1]);
            mma8(1, 0, A1[0], W0[0]);
            mma8(1, 0, A1[1], W0[1]);
            __builtin_amdgcn_s_setprio(0);
            if (GRP == 0) wait_next();
            bar();
            PPT_SLOT(3)
#else
            // phase 0
            if (!skipA) issue_quarter(0, kt + 1);
            rd_w(T, 0, W0); rd_a(T, 0, 0, A0[0]);
            bar();
            PPT_SLOT(0)
            rd_a(T, 0, 1, A0[1]);                           // lands under the first 8 MFMAs
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mma8(0, 0, A0[0], W0[0]);
            lgk0();
            mma8(0, 0, A0[1], W0[1]);
            __builtin_amdgcn_s_setprio(0);
            bar();
            PPT_SLOT(1)
            // phase 1
            if (!skipA) issue_quarter(1, kt + 1);
            rd_w(T, 1, W1); rd_a(T, 1, 0, A1[0]);
            lgk0();                                         // the last readers of this stage's W quarters retire BEFORE the barrier: phase 2 refills them
            bar();
            PPT_SLOT(2)
            __builtin_amdgcn_s_setprio(1);
            mma8(0, 1, A0[0], W1[0]);
            mma8(0, 1, A0[1], W1[1]);
            __builtin_amdgcn_s_setprio(0);
            bar();
            PPT_SLOT(3)
            // phase 2
            issue_quarter(2, kt + 2);
            rd_a(T, 1, 1, A1[1]);
            lgk0();                                         // likewise this stage's A quarters (refilled by the next K-tile's phases 0 / 1, or the epilogue)
            bar();
            PPT_SLOT(4)
            __builtin_amdgcn_s_setprio(1);
            mma8(1, 1, A1[0], W1[0]);
            mma8(1, 1, A1[1], W1[1]);
            __builtin_amdgcn_s_setprio(0);
            bar();
            PPT_SLOT(5)
            // phase 3 + the one wait of the K-tile, one slot before the next K-tile's first reader (wave row 0 reads at the next slot boundary:
            // row 0 waits at the end of its MFMA slot, row 1 at the end of its LOAD slot — the same barrier for both)
            issue_quarter(3, kt + 2);
            const bool issued = kt + 2 < nk || has_next;    // the two W quarters of this K-tile's phases 2 / 3 exist
            auto wait_next = [&]() {
                if (!issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (skipA && ahead == 1) wait_vm<4 + NSTORE + NAUX>();      // [A(1) quarters][stores, start-value loads][W(2) quarters]: only the first must have landed
                else wait_vm<4>();
            };
            if (GRP == 1) wait_next();
            bar();
            PPT_SLOT(6)
            __builtin_amdgcn_s_setprio(1);
            mma8(1, 0, A1[0], W0[0]);
            mma8(1, 0, A1[1], W0[1]);
            __builtin_amdgcn_s_setprio(0);
            if (GRP == 0) wait_next();
            bar();
            PPT_SLOT(7)
#endif
#ifdef PP_TIMING
            {
                const unsigned long long d = __builtin_amdgcn_s_memtime() - tkt0;
                if (kt == nk - 1) t_ktl += d; else if (kt == 0) t_kt0 += d; else if (kt == 1) t_kt1 += d; else if (kt == 2) t_kt2 += d; else if (kt == 3) t_kt3 += d; else t_ktr += d;
            }
#endif
        }
        // ---- tile boundary ------------------------------------------------------------------------------------------------------------------------
#if PP_RESYNC
        if (GRP == 0) bar();                                // pairs with row 1's last barrier of the tile: the rows enter the epilogue together
#endif
#ifdef PP_TIMING
        const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
#endif
        cnt += nk;
        const bool interior = PP_NO_EDGE || m0 + TM <= Mrt;
        auto issue_next_a1 = [&]() {                        // the A quarters of the next tile's K-tile 1 -> the stage of the K-tile just finished, BEFORE the stores
            if (has_next) {
                issue_at(0, 1, (cnt + 1) & 1, a_nxt, w_nxt, edge_nxt, lim_nxt);
                issue_at(1, 1, (cnt + 1) & 1, a_nxt, w_nxt, edge_nxt, lim_nxt);
            }
        };
        if constexpr (!RESK) issue_next_a1();
        {
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int slot = lane_e & 7, lrow = lane_e >> 3;
            char* stg = smem + 2 * STAGE + wave * 4096;
            // (SwiGLU: the wave's 64 columns are 32 (gate, up) pairs = 32 output columns; an output row piece is 64 bytes = 4 slots, a pass's 32 rows two store rounds of 16 rows)
            const int64_t wave_elem = (int64_t)(m0 + GRP * 128) * g.ldc + (SWG ? n0 / 2 + wn * 32 : n0 + wn * 64);
            char* c_wave = reinterpret_cast<char*>(g.C + wave_elem);
            const char* r_wave = reinterpret_cast<const char*>(g.res + (RESK ? wave_elem : 0));
            const unsigned lane_off = SWG ? (unsigned)((lane_e >> 2) * (int)g.ldc + (lane_e & 3) * 8) * 2u : (unsigned)(lrow * (int)g.ldc + slot * 8) * 2u;
            const unsigned row8 = (unsigned)g.ldc * (SWG ? 32u : 16u);           // bytes between consecutive store rounds (8 rows; SwiGLU: 16 rows)
#if PP_NT_STORE
            const unsigned long long c_wave_s = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)c_wave) |
                                                ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)c_wave >> 32)) << 32);
#endif
            // With a residual: the rows of pass h + 1 are requested inside pass h (ordinary loads: between the epilogue's own loads and stores no
            // LDS-DMA request is issued, so the compiler's counted waits see the queue as it is); pass 0's rows are requested here, FIRST, and the
            // next tile's A quarters go out only after pass 0 has them — requested the other way round, the wait for the rows would also wait for
            // quarters that were asked for a moment ago.
            bf16x8 rv[4];
            // (The residual rows stay ORDINARY loads although the streaming stores below are inline assembly the compiler's wait counting cannot see: its
            //  counts are then too small — a wait for pass h + 1's rows also waits for pass h's stores — which is safe and costs nothing measurable.
            //  Assembly loads with hand-counted waits were built: the same speed, and the -DPP_TIMING build of that form faulted; not kept.)
            auto load_residual = [&](int h) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int grow = m0 + GRP * 128 + h * 32 + it * 8 + lrow;
                    if (interior || grow < Mrt) rv[it] = *reinterpret_cast<const bf16x8*>(r_wave + (size_t)(h * 4 + it) * row8 + lane_off);
                }
            };

            if constexpr (RESK) load_residual(0);
#ifdef PP_TIMING
            unsigned long long tp = __builtin_amdgcn_s_memtime();
            t_pre += tp - ts2;
#endif
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                float er_pass[2] = {1.f, 1.f};
                if constexpr (LNK) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
                        er_pass[tt] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((l15 + 16 * h) << 2, __builtin_bit_cast(int, ers[tt])));
                }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // Everything on PAIRS (register-aligned halves of the accumulator quad): the rstd scaling, the activation's multiplies / add
                        // and the conversion issue as packed instructions.  Left to itself the compiler paired elements 1 and 2 of the quad and
                        // patched the bf16 words together with v_perm / v_alignbit (10 instructions per quad instead of 4).
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        bf16x4 v;
                        const f32x4 xq = acc[2 * h + tt][j];
                        if constexpr (SWG) {
                            // the quad is (gate, up, gate, up) of output columns j * 8 + 2 g4, + 1: two outputs, 4 bytes, into a 64-byte staging row whose 16-byte slot is
                            // XOR-ed with (row >> 1) & 3 (16 rows 64 bytes apart would meet in 4 banks).  swiglu16: torch's rounding points (gate and up to the element
                            // type, the activation, the product) — the bits of setok_swiglu_pairs on the unfused GEMM's output
                            bf16x2 o2;
                            o2[0] = swiglu16<bf16>(xq[0], xq[1]); o2[1] = swiglu16<bf16>(xq[2], xq[3]);
                            const int srow = tt * 16 + l15;
                            *reinterpret_cast<bf16x2*>(stg + srow * 64 + ((j ^ ((srow >> 1) & 3)) << 4) + 4 * g4) = o2;
                            continue;
                        }
                        if constexpr (ACT == SETOK_ACT_GELU_ERF) {           // (element by element as in the other kernels: on pairs the compiler contracts
                            f32x4 xs = xq;                                   //  the polynomial differently and the last bit moves)
#ifndef PP_ABL_NOSCALE
                            if constexpr (LNK) xs = xs * er_pass[tt];
#endif
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = (bf16)gelu_erf_fast(xs[e]);
                        } else
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {
                            f32x2 x2 = {xq[e], xq[e + 1]};
#ifndef PP_ABL_NOSCALE
                            if constexpr (LNK) x2 = x2 * f32x2{er_pass[tt], er_pass[tt]};
#endif
                            if constexpr (ACT == SETOK_ACT_QUICK_GELU) {
                                // x * sigmoid(1.702 x), 1.702 log2(e) = 2.45546696: the same five operations per element as in the other kernels
                                const f32x2 t = x2 * f32x2{-2.45546696f, -2.45546696f};
                                const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + f32x2{1.0f, 1.0f};
                                x2 = x2 * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
                            }
                            const bf16x2 b2 = __builtin_convertvector(x2, bf16x2);
                            v[e] = b2[0]; v[e + 1] = b2[1];
                        }
                        const int srow = tt * 16 + l15;
                        *reinterpret_cast<bf16x4*>(stg + srow * 128 + (((j * 2 + (g4 >> 1)) ^ (srow & 7)) << 4) + 8 * (g4 & 1)) = v;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bf16x8 ov[4];
                if constexpr (SWG) {
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int row = it * 16 + (lane_e >> 2);
                        ov[it] = *reinterpret_cast<const bf16x8*>(stg + row * 64 + (((lane_e & 3) ^ ((row >> 1) & 3)) << 4));
                    }
                } else {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + lrow;
                    ov[it] = *reinterpret_cast<const bf16x8*>(stg + row * 128 + ((slot ^ (row & 7)) << 4));
                }
                }
                if constexpr (RESK) {
#pragma unroll
                    for (int it = 0; it < 4; ++it)
#pragma unroll
                        for (int e = 0; e < 8; ++e) ov[it][e] = (bf16)((float)ov[it][e] + (float)rv[it][e]);     // round, THEN add the residual (torch's bf16 semantics)
                    if (h == 0) { asm volatile("" ::: "memory"); issue_next_a1(); }
                    if (h + 1 < 4) load_residual(h + 1);          // (all four passes' rows at the top of the epilogue instead: no difference in time; requested
                                                                  //  inside the tile's last K-tile, the rows need 64 registers the K loop does not have: -14 %)
                }
#ifndef PP_START_PASS
#define PP_START_PASS 0     // (A/B: requested in pass 2 instead, the loads are not back when the next tile starts: fc1 with the LayerNorm folded in -3 %)
#endif
                // the next tile's start values, into registers the K loop's fragments freed.  Unconditional (after the last tile nn0 / nm0 still name
                // a tile of this workgroup): a conditional load would make the values loop-carried and keep their registers live across the K loop.
                if (h == PP_START_PASS) load_start(nn0, nm0);
#pragma unroll
                for (int it = 0; it < (SWG ? 2 : 4); ++it) {
                    const int grow = m0 + GRP * 128 + h * 32 + (SWG ? it * 16 + (lane_e >> 2) : it * 8 + lrow);
                    if (interior || grow < Mrt) {
#if PP_NT_STORE
                        if constexpr (PP_NT_STORE == 1 || (PP_NT_STORE == 2 && !RESK) || (PP_NT_STORE == 3 && RESK)) {
                            // streaming store in SGPR-base form (the builtin falls back to 64-bit per-lane addresses)
                            // (s_nop 1: a VALU write of the data registers of a > 64-bit store needs two wait states behind it on gfx940+ — the compiler's hazard
                            //  recognizer pads its own stores, it cannot know that this statement is one: without the pad the erf-GELU instantiation, which reuses
                            //  the registers at once, stored the next element's intermediates — profiles/r05_nt_store_hazard.log)
                            if constexpr (RESK)
                                asm volatile("global_store_dwordx4 %0, %1, %2" PP_POL_C_RES "\n\ts_nop 1" :: "v"(lane_off), "v"(ov[it]), "s"(c_wave_s + (unsigned long long)(h * 4 + it) * row8) : "memory");
                            else
                                asm volatile("global_store_dwordx4 %0, %1, %2" PP_POL_C "\n\ts_nop 1" :: "v"(lane_off), "v"(ov[it]), "s"(c_wave_s + (unsigned long long)(h * (SWG ? 2 : 4) + it) * row8) : "memory");
                        } else
#endif
                        *reinterpret_cast<bf16x8*>(c_wave + (size_t)(h * (SWG ? 2 : 4) + it) * row8 + lane_off) = ov[it];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the staging rows are rewritten by the next pass)
#ifdef PP_TIMING
                { const unsigned long long tq = __builtin_amdgcn_s_memtime(); const unsigned long long d = tq - tp; tp = tq;
                  if (h == 0) t_p0 += d; else if (h == 1) t_p1 += d; else if (h == 2) t_p2 += d; else t_p3 += d; }
#endif
            }
        }
#ifdef PP_TIMING
        { const unsigned long long ts3 = __builtin_amdgcn_s_memtime(); t_init += ts1 - ts0; t_k += ts2 - ts1; t_epi += ts3 - ts2; }
        if (!has_next && g.tim && tid == 0) { g.tim[blockIdx.x * 4 + 0] = t_k; g.tim[blockIdx.x * 4 + 1] = t_epi; g.tim[blockIdx.x * 4 + 2] = t_init; g.tim[blockIdx.x * 4 + 3] = (unsigned long long)(round + 1) << 40; }
        if (!has_next && g.tim && (tid == 0 || tid == 256)) {          // the fine split: wave 0 (row 0) and wave 4 (row 1), behind the coarse table
            unsigned long long* f = g.tim + 256 * 4 + (blockIdx.x * 2 + (tid >> 8)) * 20;
            f[0] = t_kt0; f[1] = t_kt1; f[2] = t_kt2; f[3] = t_kt3; f[4] = t_ktr; f[5] = t_ktl; f[6] = t_pre; f[7] = t_p0; f[8] = t_p1; f[9] = t_p2; f[10] = t_p3; f[11] = (unsigned long long)(round + 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) f[12 + i] = t_s[i];
        }
#endif
        if (!has_next) break;
        ahead = interior ? 1 : 2;
        m0 = nm0; n0 = nn0; ++round;
        a_cur = a_nxt; w_cur = w_nxt; edge_cur = edge_nxt; lim_cur = lim_nxt;
        has_next = tile_of(round + 1, nm0, nn0);
        if (has_next) tile_ptrs(nm0, nn0, a_nxt, w_nxt, edge_nxt, lim_nxt);
    }
#if !PP_RESYNC
    if (GRP == 0) bar();                                    // matches the second row's last barrier
#endif
#if PP_MERGE_REM
    // ---- the remainder rows (round 5) ---------------------------------------------------------------------------------------------------------------
    // M = 65792 = 257 tile rows: 257 * N / 256 tiles never divide by 256 CUs, so the last tile row was a launch of its own on the small-tile kernel
    // (8-9 us + the ~5 us between two dependent dispatches, four times per ViT layer).  Here the first wave row of workgroups 0 .. rem_tiles - 1 runs
    // that kernel's tile function behind its last 256 x 256 tile — same arithmetic, same bits; the second wave row has ended by then or ends without
    // another barrier (a barrier counts the surviving waves only).
    if constexpr (GRP == 0) {
        if (g.rem_rows > 0) {
            bar();                                          // this row's four waves are out of their epilogues: wave 0's staging rows hold the small tile's bias rows
            if ((int)blockIdx.x < g.rem_tiles) {
                const int RT = g.rem_shape;                 // 64 (256 rows x 3072 / 4096 columns) or 32 (x 1024 columns; a few rows): tail_shape_for's choice
                PArgs t = g;
                t.A = g.A + (int64_t)g.M * g.lda;
                t.C = g.C + (int64_t)g.M * g.ldc;
                if constexpr (RESK) t.res = g.res + (int64_t)g.M * g.ldc;
                if constexpr (LNK) t.ln_stats = g.ln_stats + 8 * (int64_t)g.M;
                t.M = g.rem_rows; t.tilesM = (g.rem_rows + RT - 1) / RT; t.tilesN = g.N / RT; t.m_dev = nullptr;
                if (RT == 64) gemm_tail_tile<ACT, LNK, 64, 64, TNS>(t, blockIdx.x, smem);
                else gemm_tail_tile<ACT, LNK, 32, 32, TNS>(t, blockIdx.x, smem);
            }
        }
    }
#endif
}

template <int ACT, bool LNK, bool RESK = false>
__global__ __launch_bounds__(512) void gemm_pp_kernel(PArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef PP_TIMING           // the constant-rate (100 MHz) clock every workgroup of the chip reads the same: when did this workgroup start, and when did its first wave row end
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef PP_STAGGER_US       // A/B (round 6): every other workgroup of an XCD starts a residual launch late, so that the two halves' epilogues — each a burst of residual rows the
                           // whole chip asks for at once — do not coincide
    if (RESK && ((blockIdx.x >> 3) & 1)) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)(PP_STAGGER_US * 100)) __builtin_amdgcn_s_sleep(8);
    }
#endif
    if (wave < 4) gemm_pp_body<ACT, LNK, RESK, 0>(g, smem); else gemm_pp_body<ACT, LNK, RESK, 1>(g, smem);
#ifdef PP_TIMING
    if (g.tim && (threadIdx.x == 0 || threadIdx.x == 256)) {
        unsigned long long* f = g.tim + 256 * 4 + 256 * 2 * 20 + blockIdx.x * 3;
        if (threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); f[0] = rt0; f[1] = __builtin_amdgcn_s_memrealtime(); }
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); f[2] = __builtin_amdgcn_s_memrealtime(); }
    }
#endif
}

static bool pp_enabled() {
    static const bool on = [] { const char* e = getenv("SETOK_GEMM_PP"); return !(e && e[0] == '0'); }();
    return on;
}

// Does this main launch go to the ping-pong kernel?
static bool pp_takes(const PArgs& g) {
    static const bool pp_res = [] { const char* e = getenv("SETOK_GEMM_PP_RES"); return !(e && e[0] == '0'); }();     // A/B: residual launches on the old kernel
    const bool res = g.res && !(g.dbg & 2);
    return (!res || pp_res) && !g.Cf && g.N % 256 == 0 && g.K >= 128 && (PP_TIMING_ON || !g.tim) && pp_enabled() && (!PP_NO_EDGE || (g.M % TM == 0 && !g.m_dev));
}

int launch_main(hipStream_t s, const PArgs& g, int act, int n_cu, hipEvent_t e0, hipEvent_t e1) {
    static SetokDeviceOnce once;
    if (!once.run([] {
            bool ok = true;
            const void* fns[] = {(const void*)gemm_persist_kernel<0, false, false>, (const void*)gemm_persist_kernel<1, false, false>,
                                 (const void*)gemm_persist_kernel<2, false, false>, (const void*)gemm_persist_kernel<0, false, true>,
                                 (const void*)gemm_persist_kernel<1, false, true>, (const void*)gemm_persist_kernel<2, false, true>,
                                 (const void*)gemm_persist_kernel<0, false, false, true>, (const void*)gemm_persist_kernel<1, false, false, true>,
                                 (const void*)gemm_persist_kernel<2, false, false, true>};
            for (const void* f : fns) ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, MAIN_LDS) == hipSuccess;
            return ok; }))
        return setok_fail(SETOK_ELAUNCH, "setok_linear: cannot raise the dynamic LDS limit");
    const int tiles = g.tilesM * g.tilesN;
    const int grid = tiles < n_cu ? tiles : n_cu;
    const bool res = g.res && !(g.dbg & 2);
    const dim3 gr(grid), bl(512);
    if (pp_takes(g)) {                                      // the ping-pong schedule (whole tiles)
        static SetokDeviceOnce once_pp;
        if (!once_pp.run([] {
                bool ok = true;
                const void* fns[] = {(const void*)gemm_pp_kernel<SETOK_ACT_SWIGLU_PAIRS, false>,
                                     (const void*)gemm_pp_kernel<0, false>, (const void*)gemm_pp_kernel<1, false>, (const void*)gemm_pp_kernel<2, false>,
                                     (const void*)gemm_pp_kernel<0, true>, (const void*)gemm_pp_kernel<1, true>, (const void*)gemm_pp_kernel<2, true>,
                                     (const void*)gemm_pp_kernel<0, false, true>, (const void*)gemm_pp_kernel<1, false, true>, (const void*)gemm_pp_kernel<2, false, true>};
                for (const void* f : fns) ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, MAIN_LDS) == hipSuccess;
                return ok; }))
            return setok_fail(SETOK_ELAUNCH, "setok_linear: cannot raise the dynamic LDS limit");
        if (g.ln_stats) {
            if (act == SETOK_ACT_NONE) setok_launch(gemm_pp_kernel<0, true>, gr, bl, MAIN_LDS, s, e0, e1, g);
            else if (act == SETOK_ACT_QUICK_GELU) setok_launch(gemm_pp_kernel<1, true>, gr, bl, MAIN_LDS, s, e0, e1, g);
            else setok_launch(gemm_pp_kernel<2, true>, gr, bl, MAIN_LDS, s, e0, e1, g);
        } else if (res) {
            if (act == SETOK_ACT_NONE) setok_launch(gemm_pp_kernel<0, false, true>, gr, bl, MAIN_LDS, s, e0, e1, g);
            else if (act == SETOK_ACT_QUICK_GELU) setok_launch(gemm_pp_kernel<1, false, true>, gr, bl, MAIN_LDS, s, e0, e1, g);
            else setok_launch(gemm_pp_kernel<2, false, true>, gr, bl, MAIN_LDS, s, e0, e1, g);
        } else if (act == SETOK_ACT_SWIGLU_PAIRS) setok_launch(gemm_pp_kernel<SETOK_ACT_SWIGLU_PAIRS, false>, gr, bl, MAIN_LDS, s, e0, e1, g);
        else if (act == SETOK_ACT_NONE) setok_launch(gemm_pp_kernel<0, false>, gr, bl, MAIN_LDS, s, e0, e1, g);
        else if (act == SETOK_ACT_QUICK_GELU) setok_launch(gemm_pp_kernel<1, false>, gr, bl, MAIN_LDS, s, e0, e1, g);
        else setok_launch(gemm_pp_kernel<2, false>, gr, bl, MAIN_LDS, s, e0, e1, g);
        SETOK_CHECK_LAUNCH("setok_linear(ping-pong)");
        return SETOK_OK;
    }
    if (g.ln_stats) {
        if (act == SETOK_ACT_NONE) setok_launch(gemm_persist_kernel<0, false, false, true>, gr, bl, MAIN_LDS, s, e0, e1, g);
        else if (act == SETOK_ACT_QUICK_GELU) setok_launch(gemm_persist_kernel<1, false, false, true>, gr, bl, MAIN_LDS, s, e0, e1, g);
        else setok_launch(gemm_persist_kernel<2, false, false, true>, gr, bl, MAIN_LDS, s, e0, e1, g);
        SETOK_CHECK_LAUNCH("setok_linear_ln(persistent)");
        return SETOK_OK;
    }
    if (act == SETOK_ACT_NONE) { if (res) setok_launch(gemm_persist_kernel<0, false, true, false>, gr, bl, MAIN_LDS, s, e0, e1, g); else setok_launch(gemm_persist_kernel<0, false, false, false>, gr, bl, MAIN_LDS, s, e0, e1, g); }
    else if (act == SETOK_ACT_QUICK_GELU) { if (res) setok_launch(gemm_persist_kernel<1, false, true, false>, gr, bl, MAIN_LDS, s, e0, e1, g); else setok_launch(gemm_persist_kernel<1, false, false, false>, gr, bl, MAIN_LDS, s, e0, e1, g); }
    else { if (res) setok_launch(gemm_persist_kernel<2, false, true, false>, gr, bl, MAIN_LDS, s, e0, e1, g); else setok_launch(gemm_persist_kernel<2, false, false, false>, gr, bl, MAIN_LDS, s, e0, e1, g); }
    SETOK_CHECK_LAUNCH("setok_linear(persistent)");
    return SETOK_OK;
}

int cu_count() {                                    // of the CURRENT device
    static SetokPerDevice<int> cache;
    int n = 256;
    cache.get(n, [](int& v) {
        int dev = 0;
        return hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0;
    });
    return n;
}

const float* zero_bias() {                          // the current device's copy of kZeroBias (a __device__ symbol has one address per device)
    static SetokPerDevice<const float*> cache;
    const float* p = nullptr;
    cache.get(p, [](const float*& v) { void* q = nullptr; if (hipGetSymbolAddress(&q, HIP_SYMBOL(kZeroBias)) != hipSuccess) return false; v = (const float*)q; return true; });
    return p;
}

}  // namespace

// Called by setok_linear (gemm.hip) for bf16 -> bf16 problems with >= 48 tiles of 256x256.
int setok_gemm_persist_bf16(hipStream_t s, const bf16* A, int64_t lda, const bf16* W, const float* bias, const bf16* res,
                            bf16* C, int64_t ldc, int M, int N, int K, int act, const float* ln_stats, const float* ln_colsum, const int32_t* m_dev) {
    const int ncu = cu_count();
    const int tilesM = cdiv(M, TM), tilesN = cdiv(N, 256);
    // Peel off p <= 2 trailing M-tiles when that leaves the main launch an exact number of rounds: the
    // remainder (p*256 rows) runs as 256x64 tiles — a fraction of a round instead of a full extra one.
    int p = 0;
    const int T = tilesM * tilesN, r = T % ncu;
    if (T > ncu && r != 0 && r % tilesN == 0 && r / tilesN <= 2) p = r / tilesN;
    // SETOK_GEMM_DEBUG: bit 2 = no M-tail peel, bit 3 = everything through the 64x64 kernel (both leave the results unchanged).  Bits 0 / 1
    // (skip the C stores / the residual: ablation runs for the cycle breakdown in DESIGN.md) change the results and exist only in builds
    // with -DSETOK_GEMM_ABLATION.
    static const int dbg = [] {
        const char* e = getenv("SETOK_GEMM_DEBUG");
        int v = e ? atoi(e) : 0;
#ifndef SETOK_GEMM_ABLATION
        v &= ~3;
#endif
        return v;
    }();
    if (p == 0 && M % TM != 0 && tilesM > 1 && !m_dev && N % 256 == 0 && pp_enabled()) p = 1;   // the ping-pong kernel takes whole 256-row tiles only: the ragged last one goes to the small-tile kernel
    if (dbg & 4 || m_dev) p = 0;                                    // a device-side row count: no host-side split of M
    if (dbg & 8) p = tilesM;                                        // experiment: everything through the deep-pipeline 64x64 kernel
    const int tm_main = tilesM - p;
    static const bool timing = [] { const char* e = getenv("SETOK_GEMM_TIMING"); return e && e[0] == '1'; }();
    static unsigned long long* tim = nullptr;
    if (timing && !tim) { if (hipMalloc(&tim, (256 * 4 + 256 * 2 * 20 + 256 * 3) * 8) != hipSuccess) tim = nullptr; else (void)hipMemset(tim, 0, (256 * 4 + 256 * 2 * 20 + 256 * 3) * 8); }
    PArgs g{A, W, bias, res, C, lda, ldc, (tilesM - p) * TM < M ? (tilesM - p) * TM : M, N, K, tilesM - p, tilesN, dbg, timing ? tim : nullptr, nullptr, nullptr, 0, 0, 0, 1,
            ln_stats, ln_colsum, m_dev};
    if (!bias || ln_stats) {
        const float* zb = zero_bias();
        if (!zb) return setok_fail(SETOK_ELAUNCH, "setok_linear: cannot resolve the zero-bias symbol");
        g.zero_bias = zb;
    }
    // the profiler's timestamps ride on the dispatches themselves: start of the first launch, end of the last (common.h)
    const bool has_main = (tilesM - p) > 0;
    // The remainder rows inside the main launch (end of gemm_pp_body): when the main launch is a ping-pong one and the remainder's small tiles — of the
    // shape tail_shape_for would pick, 64 x 64 or 32 x 32 — are no more than its workgroups.  SETOK_GEMM_MERGE_REM=0: a launch
    // of their own as before (A/B runs; the results are the same bits either way).
    static const bool merge_rem = [] { const char* e = getenv("SETOK_GEMM_MERGE_REM"); return !(e && e[0] == '0'); }();
    bool merged = false;
    if (PP_MERGE_REM && merge_rem && has_main && p > 0 && pp_takes(g)) {
        const int rem = M - tm_main * TM;
        const TailShape sh = tail_shape_for(rem, N, ncu);
        const int grid = tm_main * tilesN < ncu ? tm_main * tilesN : ncu;
        int shape = sh.ttm == sh.ttn && sh.ns == TNS ? sh.ttm : 64;          // (the two-per-CU and 64 x 32 shapes exist as launches only)
        if (cdiv(rem, shape) * (N / shape) > grid) shape = 64;               // more 32 x 32 tiles than workgroups (128 rows x 3072 columns at 336^2: 384): one 64 x 64 tile per
        const int rem_tiles = cdiv(rem, shape) * (N / shape);                //   workgroup inside the launch still beats a launch of its own; the bits do not depend on the shape
        if (rem_tiles <= grid) { g.rem_rows = rem; g.rem_tiles = rem_tiles; g.rem_shape = shape; merged = true; }
    }
    int rc = has_main ? launch_main(s, g, act, ncu, setok_prof_start_event(), p == 0 || merged ? setok_prof_stop_event() : nullptr) : SETOK_OK;
    if (timing && tim) {
        unsigned long long h[256 * 4];
        if (hipMemcpy(h, tim, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            double a = 0, b = 0, c = 0, r = 0; int nb = ncu < tilesN * (tilesM - p) ? ncu : tilesN * (tilesM - p);
            double bar = 0;
            for (int i = 0; i < nb; ++i) { a += h[i * 4]; b += h[i * 4 + 1]; c += h[i * 4 + 2]; r += (double)(h[i * 4 + 3] >> 40); bar += (double)(h[i * 4 + 3] & ((1ull << 40) - 1)); }
            fprintf(stderr, "[gemm timing] M=%d N=%d K=%d tiles/block=%.2f  per tile: main %.0f cyc (vmcnt waits %.0f, barrier waits %.0f), epilogue %.0f cyc\n",
                    M, N, K, r / nb, a / r, c / r, bar / r, b / r);
#ifdef PP_TIMING
            {
                static unsigned long long hr[256 * 3];
                if (hipMemcpy(hr, tim + 256 * 4 + 256 * 2 * 20, sizeof(hr), hipMemcpyDeviceToHost) == hipSuccess) {
                    unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0; double es = 0;
                    for (int i = 0; i < nb; ++i) {
                        const unsigned long long st = hr[i * 3], en = hr[i * 3 + 1] > hr[i * 3 + 2] ? hr[i * 3 + 1] : hr[i * 3 + 2];
                        s0 = st < s0 ? st : s0; s1 = st > s1 ? st : s1; e0 = en < e0 ? en : e0; e1 = en > e1 ? en : e1; es += (double)en;
                    }
                    {
                        double xs[8] = {0}, xe[8] = {0}, xl[8] = {0}; int xn[8] = {0};
                        for (int i = 0; i < nb; ++i) {
                            const unsigned long long en = hr[i * 3 + 1] > hr[i * 3 + 2] ? hr[i * 3 + 1] : hr[i * 3 + 2];
                            xs[i & 7] += (double)(hr[i * 3] - s0); xe[i & 7] += (double)(en - s0); xn[i & 7]++; if ((double)(en - s0) > xl[i & 7]) xl[i & 7] = (double)(en - s0);
                        }
                        fprintf(stderr, "[gemm timing span] per XCD (workgroup & 7), mean start / mean end / last end after the first start:");
                        for (int x = 0; x < 8; ++x) if (xn[x]) fprintf(stderr, "  %d: %.0f / %.0f / %.0f", x, xs[x] / xn[x], xe[x] / xn[x], xl[x]);
                        fprintf(stderr, "\n");
                    }
                    fprintf(stderr, "[gemm timing span] %d workgroups (100 MHz ticks = 10 ns): first start -> last end %llu; starts spread over %llu; ends spread over %llu (mean end %.0f before the last)\n",
                            nb, e1 - s0, s1 - s0, e1 - e0, (double)e1 - es / nb);
                }
            }
            static unsigned long long hf[256 * 2 * 20];
            if (hipMemcpy(hf, tim + 256 * 4, sizeof(hf), hipMemcpyDeviceToHost) == hipSuccess) {
                for (int row = 0; row < 2; ++row) {
                    double s[11] = {0}, sl[8] = {0}, rr = 0;
                    for (int i = 0; i < nb; ++i) { const unsigned long long* f = hf + (i * 2 + row) * 20; for (int k = 0; k < 11; ++k) s[k] += (double)f[k]; rr += (double)f[11]; for (int k = 0; k < 8; ++k) sl[k] += (double)f[12 + k]; }
                    if (rr > 0)
                        fprintf(stderr, "[gemm timing fine] wave row %d: K-tile 0 %.0f, 1 %.0f, 2 %.0f, 3 %.0f, middle (each of %d) %.0f, last %.0f | epilogue: pre %.0f, pass0 %.0f, pass1 %.0f, pass2 %.0f, pass3 %.0f\n",
                                row, s[0] / rr, s[1] / rr, s[2] / rr, s[3] / rr, K / TK - 5, (K / TK > 5 ? s[4] / rr / (K / TK - 5) : 0.0), s[5] / rr, s[6] / rr, s[7] / rr, s[8] / rr, s[9] / rr, s[10] / rr);
                    if (rr > 0)
                        fprintf(stderr, "[gemm timing last] wave row %d: slots of the last K-tile %.0f %.0f %.0f %.0f %.0f %.0f %.0f %.0f\n", row, sl[0] / rr, sl[1] / rr, sl[2] / rr, sl[3] / rr, sl[4] / rr, sl[5] / rr, sl[6] / rr, sl[7] / rr);
                }
            }
#endif
        }
    }
    if (rc != SETOK_OK || p == 0 || merged) return rc;
    const int m_off = tm_main * TM;
    const TailShape sh = tail_shape_for(M - m_off, N, ncu);          // the remainder rows: a launch of its own, far from filling the chip
    PArgs t{A + (int64_t)m_off * lda, W, bias, res ? res + (int64_t)m_off * ldc : nullptr, C + (int64_t)m_off * ldc,
            lda, ldc, M - m_off, N, K, cdiv(M - m_off, sh.ttm), cdiv(N, sh.ttn), dbg, nullptr, nullptr, nullptr, 0, 0, 0, 1,
            ln_stats ? ln_stats + 8 * (int64_t)m_off : nullptr, ln_colsum, nullptr};
    return launch_tail(s, t, act, has_main ? nullptr : setok_prof_start_event(), setok_prof_stop_event(), sh);
}

// Called by setok_linear for SMALL bf16 -> bf16 problems (a handful of images: too few 128 x 128 tiles to fill 256 CUs): the deep-pipelined
// 64 x 64 kernel over the whole problem.
// act_fn(gate) * up in the epilogue of the gate|up GEMM (setok_linear_swiglu, llama.hip): W holds (gate_j, up_j) as rows 2 j, 2 j + 1; C is (M, N / 2).  Whole
// 256-row tiles of the ping-pong kernel only (the caller sends the rows behind them through setok_linear + setok_swiglu_pairs: the same bits).
int setok_gemm_swiglu_bf16(hipStream_t s, const bf16* A, int64_t lda, const bf16* W, bf16* C, int64_t ldc, int M, int N, int K) {
    if (M % TM != 0 || N % 256 != 0 || K % TK != 0 || K < 128 || !pp_enabled() || PP_TIMING_ON) return SETOK_EUNSUPPORTED;
    const float* zb = zero_bias();
    if (!zb) return setok_fail(SETOK_ELAUNCH, "setok_linear_swiglu: cannot resolve the zero-bias symbol");
    PArgs g{A, W, nullptr, nullptr, C, lda, ldc, M, N, K, M / TM, N / 256, 0, nullptr, zb, nullptr, 0, 0, 0, 1, nullptr, nullptr, nullptr};
    return launch_main(s, g, SETOK_ACT_SWIGLU_PAIRS, cu_count(), setok_prof_start_event(), setok_prof_stop_event());
}

int setok_gemm_small_bf16(hipStream_t s, const bf16* A, int64_t lda, const bf16* W, const float* bias, const bf16* res,
                          bf16* C, int64_t ldc, int M, int N, int K, int act, const float* ln_stats, const float* ln_colsum, const int32_t* m_dev) {
    const TailShape sh = tail_shape_for(M, N, cu_count(), !ln_stats && !m_dev);
    PArgs t{A, W, bias, res, C, lda, ldc, M, N, K, cdiv(M, sh.ttm), cdiv(N, sh.ttn), 0, nullptr, nullptr, nullptr, 0, 0, 0, 1, ln_stats, ln_colsum, m_dev};
    if (sh.ttm == 128) {                                    // whole 128-row tiles as the grid, the rows behind them inside the launch (gemm_tail_kernel)
        t.M = M / 128 * 128; t.tilesM = M / 128;
        t.rem_rows = M % 128; t.rem_tiles = cdiv(M % 128, 32) * (N / 32); t.rem_shape = 32;
    }
    const hipEvent_t e0 = setok_prof_start_event();
    return launch_tail(s, t, act, e0, setok_prof_stop_event(), sh);
}

// Called by setok_linear for bf16 -> fp32 batched problems (no bias / activation / residual): the split-K partial products of a weight
// gradient.  Same kernel, F32B variant.
int setok_gemm_persist_f32_batched(hipStream_t s, const bf16* A, int64_t lda, const bf16* W, float* C, int64_t ldc, int M, int N, int K, int batch,
                                   int64_t sA, int64_t sW, int64_t sC) {
    static SetokDeviceOnce once;
    if (!once.run([] { return hipFuncSetAttribute((const void*)gemm_persist_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, MAIN_LDS) == hipSuccess; }))
        return setok_fail(SETOK_ELAUNCH, "setok_linear: cannot raise the dynamic LDS limit");
    const float* zb = zero_bias();
    if (!zb) return setok_fail(SETOK_ELAUNCH, "setok_linear: cannot resolve the zero-bias symbol");
    const int tilesM = cdiv(M, TM), tilesN = cdiv(N, 256);
    PArgs g{A, W, nullptr, nullptr, nullptr, lda, ldc, M, N, K, tilesM, tilesN, 0, nullptr, zb, C, sA, sW, sC, batch, nullptr, nullptr, nullptr};
    const int tiles = tilesM * tilesN * batch, ncu = cu_count();
    gemm_persist_kernel<0, true><<<tiles < ncu ? tiles : ncu, 512, MAIN_LDS, s>>>(g);
    SETOK_CHECK_LAUNCH("setok_linear(persistent, fp32 batched)");
    return SETOK_OK;
}
