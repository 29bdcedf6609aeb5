// attn_vit.hip — bf16 MFMA self-attention for the ViT tower (uniform segments of T tokens, head dim 64).
//
// One workgroup per (image, head).  The head's K and V (T x 64 bf16 each, T <= 608) are staged ONCE
// in LDS (K with a 16-B-slot XOR swizzle for conflict-free ds_read_b128; V row-major, consumed through
// the gfx950 hardware-transposing ds_read_b64_tr_b16), then every wave owns 32 query rows and walks
// the keys in tiles of 32 with an online softmax that never leaves registers:
//   S^T = K Q^T  (v_mfma_f32_32x32x16_bf16, A = K fragment from LDS, B = Q fragment from HBM, kept in VGPRs)
//        -> lane (q = lane & 31, hi = lane >> 5) holds 16 of the tile's 32 keys for ITS query row,
//           so the row max / row sum are in-lane reductions plus one exchange with lane ^ 32;
//   O^T += V^T P^T  (A = V^T fragment by two transposing reads, B = P packed to bf16 in place — the
//           k-slot order of P's registers is matched by the order of the two V reads, so P needs no
//           cross-lane movement at all).
// HBM traffic per (image, head): read Q, K, V once, write O once — the algorithmic minimum.
#include "common.h"
#if defined(ATTN_NT) && (ATTN_NT & 4)     // A/B switch: the K / V staging requests as streaming loads
#define ATTN_KV_NT " nt"
#else
#define ATTN_KV_NT ""
#endif

namespace {

int cu_count() {                       // of the CURRENT device
    static SetokPerDevice<int> cache;
    int n = 256;
    cache.get(n, [](int& v) {
        int dev = 0;
        return hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0;
    });
    return n;
}

#ifndef ATTN_STAGE_SBASE
#define ATTN_STAGE_SBASE 1
#endif
#ifndef ATTN_O_SWAP
#define ATTN_O_SWAP 1
#endif
#ifndef ATTN_DEFER_MAX
#define ATTN_DEFER_MAX 8      // round 6: the running row maximum is raised — and the 32 output accumulators rescaled — only when some row's tile maximum exceeds it by more than this many
#endif                        // powers of two (0 = at every rise, rounds 1-5).  The branch is per WAVE: with 32 rows a tile raised SOME row's maximum almost every time, so nearly every
                              // tile paid 16 packed multiplies + an exp; with the threshold a query tile rescales once, after its first key tile.  Probabilities then reach 2^8 instead
                              // of 1 — nothing for an fp32 sum, a bf16 / fp16 P operand or the final division; the result's bits change within rounding.
#ifndef ATTN_SHORT_TAIL
#define ATTN_SHORT_TAIL 1     // 1 (round 6): a last key tile with at most 8 valid keys (T = 257 / 577: ONE — every ViT with a class token; T = 197: 5) runs a short form of
#endif                        // the softmax step — 4 of the 16 score registers, one of the two PV k-steps — instead of exponentiating 15 masked scores per lane
constexpr int ROWB = 128;              // bytes per K / V row in LDS: 64 dims; a 48-dim head leaves two 16-byte slots of each row unused

typedef __attribute__((ext_vector_type(4))) short short4v;

__device__ inline bf16x8 pack8(const float* p) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)p[i];
    return v;
}

// CROSS = false: self-attention over a fused [q | k | v] buffer, T rows per image (the ViT tower, the Q-Former's
// query self-attention).  CROSS = true: the Tq query rows of image b (buffer qkv, row stride ldq) attend to the ragged key
// / value segment [kv_offsets[b], kv_offsets[b+1]) of kptr / vptr (row stride ldkv) — the Q-Former's cross-attention to
// the image's cluster tokens (module.py:283-286), the additive -10000 mask of the padded reference realised as a segment.
// DH = 64 or 48 (the decoder's 768 / 16 heads): 48 runs three QK^T k-steps instead of four and stores 48 of the 64 output
// columns of the two PV tiles (the LDS slots past the head hold a copy of the head's first chunk: they only reach the unstored columns).
// HP (experiment, round 4): heads per workgroup.  HP = 2: a workgroup takes a PAIR of adjacent heads of an image — both heads' K / V
// staged (2 x the LDS: one workgroup per CU at T = 257), a wave loads its query tile for both heads up front (a row piece of 256 contiguous
// bytes) and runs the two heads one after the other.  Same arithmetic per (head, query tile): identical bits.  Measured slower; kept behind
// SETOK_ATTN_HEADPAIR=1 as the evidence (docs/PERF_NOTES.md E.5).
template <int NW, bool CROSS, int DH, int HP = 1>
__global__ __launch_bounds__(NW * 64) void attn_vit_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                           int T, int H, float scale_log2e, int64_t ldq,
                                                           const bf16* __restrict__ kptr, const bf16* __restrict__ vptr,
                                                           int64_t ldkv, const int32_t* __restrict__ kv_offsets, int Tq_,
                                                           int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // Workgroups are dealt to the 8 XCDs round-robin by linear id.  With 8 | images all heads of one image are put on ONE XCD
    // (image b on XCD b % 8): a 48-dim head is 96 bytes of a q|k|v row, so neighbouring heads share 128-byte lines, and lines
    // fetched by one XCD's L2 are not visible to another's.
    int h = blockIdx.x * HP, b = blockIdx.y;
    if ((gridDim.y & 7) == 0 && DH != 64) {
        const int id = blockIdx.y * gridDim.x + blockIdx.x, xcd = id & 7, k = id >> 3;
        b = xcd + 8 * (k / (int)gridDim.x);
        h = k % (int)gridDim.x;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = H * DH;
    int kv0 = 0;
    if constexpr (CROSS) { kv0 = kv_offsets[b]; T = min(kv_offsets[b + 1] - kv0, T); }     // T: this image's key count (<= the bound passed in)
    const int Tp = (T + 31) & ~31;
    const int Tq = CROSS ? Tq_ : T;
    const int64_t ld = CROSS ? ldq : 3LL * C;                       // query row stride
    const int64_t ldk = CROSS ? ldkv : 3LL * C;                     // key / value row stride
    const bf16* base = qkv + (int64_t)b * Tq * ld + h * DH;         // this image's queries
    const bf16* kbase = CROSS ? kptr + (int64_t)kv0 * ldk + h * DH : base + C;
    const bf16* vbase = CROSS ? vptr + (int64_t)kv0 * ldk + h * DH : base + 2 * C;
    if (CROSS && T <= 0) return;

    // ---- stage K (swizzled) and V (row-major) of this head in LDS by LDS-DMA: every 16-byte piece of the head's
    //      K and V (2 x Tp x 8 pieces) is requested up front, so the workgroup pays ONE memory round trip instead of one
    //      per loop iteration of a load->ds_write loop (measured: staging 13.5 k -> cycles per workgroup).  The LDS image is
    //      lane-linear, so K's slot swizzle is applied to the source address; padding rows re-read row T-1 (finite
    //      values; their scores are masked to -inf and their probabilities are exactly 0).
    {
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const int npieces = Tp * 8;                                  // 16-byte pieces per operand; Tp*8 is a multiple of 64
        // A request = 8 rows x 8 pieces; (p0 >> 3) is a multiple of 8, so a lane's row inside the request, its piece and K's swizzle are the same for
        // every request: ONE 32-bit lane offset per operand, the row advance goes into a wave-uniform 64-bit base (scalar adds).  Formed per lane
        // and per request (64-bit multiplies, clamps) the addresses were 68 vector instructions per pair of requests — a third of what a wave
        // spends on a whole query tile's key loop.  Only a request that reaches past row T - 1 still clamps per lane.
        const int r8 = lane >> 3, c8 = lane & 7, kc8 = c8 ^ r8;
        const unsigned koff = (unsigned)r8 * (unsigned)(ldk * 2) + (unsigned)((kc8 < DH / 8 ? kc8 : 0) << 4);
        const unsigned voff = (unsigned)r8 * (unsigned)(ldk * 2) + (unsigned)((c8 < DH / 8 ? c8 : 0) << 4);
        auto dma_s = [&](const char* base, unsigned off, unsigned dst) {
            unsigned keep;
            const unsigned long long b64 = (unsigned long long)base;
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);
            const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));
            const unsigned long long sb64 = (unsigned long long)lo | ((unsigned long long)hi32 << 32);
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ATTN_KV_NT "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(off), "s"(sb64), "s"(dst) : "memory");
        };
#pragma unroll
        for (int hh = 0; hh < HP; ++hh) {
        const bf16* kbase_h = kbase + hh * DH;
        const bf16* vbase_h = vbase + hh * DH;
        const unsigned ldsh = lds0 + (unsigned)hh * 2u * (unsigned)Tp * ROWB;
        for (int p0 = wave_u * 64; p0 < npieces; p0 += NW * 64) {    // this wave's 64 consecutive pieces
            const int key0 = p0 >> 3;
            if (ATTN_STAGE_SBASE && key0 + 8 <= T) {                  // (uniform) all eight rows exist
                const size_t rbytes = (size_t)key0 * (size_t)(ldk * 2);
                dma_s(reinterpret_cast<const char*>(kbase_h) + rbytes, koff, ldsh + p0 * 16);
                dma_s(reinterpret_cast<const char*>(vbase_h) + rbytes, voff, ldsh + Tp * ROWB + p0 * 16);
                continue;
            }
            const int p = p0 + lane, key = p >> 3, c = p & 7;
            const int64_t roff = (int64_t)min(key, T - 1) * ldk;
            const int kc = c ^ (key & 7);                               // physical slot c of row `key` holds logical chunk c ^ (key & 7)
            const bf16* ksrc = kbase_h + roff + ((kc < DH / 8 ? kc : 0) << 3);   // chunks past the head are never read back: fetch something valid
            const bf16* vsrc = vbase_h + roff + ((c < DH / 8 ? c : 0) << 3);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ATTN_KV_NT "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(ksrc), "s"(ldsh + p0 * 16) : "memory");
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ATTN_KV_NT "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vsrc), "s"(ldsh + Tp * ROWB + p0 * 16) : "memory");
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    const int qi = lane & 31, hi = lane >> 5;
    const int nq = (Tq + 31) >> 5, nkv = Tp >> 5;
    const float defer_raw = (float)ATTN_DEFER_MAX / scale_log2e;        // the threshold in raw score units
    // transposing-read address pattern: lanes 4j+p of a 16-lane group supply row j, 4-column piece p
    const int g16 = lane >> 4, i16 = lane & 15;
    const int tr_row = (i16 >> 2) + 4 * (g16 >> 1);            // + 4*hi
    const int tr_col = (g16 & 1) * 16 + (i16 & 3) * 4;         // column of this lane's 8-byte piece

    // (The class-token tile — T = 32k + 1 — costs wave 0 a second pass.  Splitting that tile's keys across the waves and
    // merging partial softmaxes through LDS was tried twice: no gain, the kernel is bound by the total number of
    // (q-tile, kv-tile) units per SIMD, and the second workgroup on the CU fills the idle waves' issue slots.)
    // Query tiles -> (workgroup z of gridDim.z, wave): tile t is taken by wave t / QS of workgroup t % QS.  QS = 1 (every launch with enough (image,
    // head) pairs to fill the chip): wave w takes tiles w, w + NW, ...  A handful of images (round 4): QS > 1 workgroups per (image, head), each
    // staging the head's K / V (L2 hits after the first) and taking every QS-th tile — 16 workgroups of nine tiles each cannot occupy 256 CUs.
    // A tile's arithmetic does not depend on who runs it: identical bits for every QS.
    const int QS = gridDim.z;
    for (int qt = wave * QS + (int)blockIdx.z; qt < nq; qt += NW * QS) {
        const int q = qt * 32 + qi;
        const bf16* qp = base + (int64_t)min(q, Tq - 1) * ld + hi * 8;
        constexpr int NKS = DH / 16;
        bf16x8 qfh[HP][NKS];
#pragma unroll
        for (int hh = 0; hh < HP; ++hh)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
#if defined(ATTN_NT) && (ATTN_NT & 2)
                qfh[hh][ks] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(qp + hh * DH + ks * 16));     // A/B: read-once rows as streaming loads
#else
                qfh[hh][ks] = *reinterpret_cast<const bf16x8*>(qp + hh * DH + ks * 16);
#endif
            }
#pragma unroll
        for (int hh = 0; hh < HP; ++hh) {
        const bf16x8 (&qf)[NKS] = qfh[hh];
        const char* Ks = lds + (size_t)hh * 2 * Tp * ROWB;
        const char* Vs = Ks + (size_t)Tp * ROWB;
        const int hcur = h + hh;

        f32x16 o[2];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;

        for (int kt = 0; kt < nkv; ++kt) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const int krow = kt * 32 + qi;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + krow * ROWB + (((ks * 2 + hi) ^ (krow & 7)) << 4));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
            }
            // s[r]: key = kt*32 + (r&3) + 8*(r>>2) + 4*hi, query = q.  The running max is kept in raw score units;
            // softmax scale and log2(e) are folded into one fma per element: p = exp2(s*c - m*c).
            float t[16];
            float mx = -INFINITY;
            const bool tail = (kt == nkv - 1) && (Tp != T);
#if ATTN_SHORT_TAIL
            if (tail && T - kt * 32 <= 8) {               // (wave-uniform)
                // The tile's valid keys are kt * 32 + r + 4 hi, r = 0 .. 3 (registers 0-3 of the score tile): the other twelve registers hold masked scores whose
                // probabilities are exactly 0 — the general form below exponentiates them, adds the zeros to the row sum and multiplies V's rows 16-31 by them.
                // Same operations in the same order on the four live registers, so the bits do not change: 9 of a T = 257 query tile's 81 (tile, tile) units
                // cost a quarter of the vector instructions and 6 instead of 8 MFMAs.
                const int nvalid = T - kt * 32;
#pragma unroll
                for (int r = 0; r < 4; ++r) { t[r] = (r + 4 * hi < nvalid) ? s[r] : -INFINITY; mx = fmaxf(mx, t[r]); }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run, mx);
                if (ATTN_DEFER_MAX ? __any(mx > m_run + defer_raw) : !__all(m_new == m_run)) {
                    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
                    l_run *= alpha;
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
                    m_run = m_new;
                }
                const float mc = m_run * scale_log2e;
                float ls = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { t[r] = __builtin_amdgcn_exp2f(fmaf(t[r], scale_log2e, -mc)); ls += t[r]; }
                l_run += ls;
#pragma unroll
                for (int r = 4; r < 8; ++r) t[r] = 0.f;
                const bf16x8 p0 = pack8(t);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const char* vb = Vs + (kt * 32 + tr_row) * ROWB + (d * 32 + tr_col) * 2;
                    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(vb));
                    const short4v hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(vb + 8 * ROWB));
                    union { short s8[8]; bf16x8 v; } u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { u.s8[j] = lo[j]; u.s8[4 + j] = hi4[j]; }
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u.v, p0, o[d], 0, 0, 0);
                }
                continue;
            }
#endif
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                t[r] = s[r];
                if (tail) { const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi; if (key >= T) t[r] = -INFINITY; }
                mx = fmaxf(mx, t[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            if (ATTN_DEFER_MAX ? __any(mx > m_run + defer_raw) : !__all(m_new == m_run)) {      // wave-uniform (see ATTN_DEFER_MAX)
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
                m_run = m_new;
            }
            const float mc = m_run * scale_log2e;
            float ls = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { t[r] = __builtin_amdgcn_exp2f(fmaf(t[r], scale_log2e, -mc)); ls += t[r]; }
            l_run += ls;
            const bf16x8 p0 = pack8(t), p1 = pack8(t + 8);
#pragma unroll
            for (int d = 0; d < 2; ++d) {
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const char* vb = Vs + (kt * 32 + k2 * 16 + tr_row) * ROWB + (d * 32 + tr_col) * 2;
                    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(vb));
                    const short4v hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(vb + 8 * ROWB));
                    union { short s8[8]; bf16x8 v; } u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { u.s8[j] = lo[j]; u.s8[4 + j] = hi4[j]; }
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u.v, k2 == 0 ? p0 : p1, o[d], 0, 0, 0);
                }
            }
        }
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        if constexpr (!CROSS && DH == 64 && ATTN_O_SWAP) {
            // The lane pair (q, q + 32) holds the query's 64 output dims in alternating 4-dim chunks (chunk j = d * 4 + r4: dims 8 j + 4 hi .. + 3).
            // One v_permlane32_swap per register exchanges the upper lanes' chunk j with the lower lanes' chunk j + 4: afterwards lane q owns dims
            // 0-31 and lane q + 32 dims 32-63 in 16-byte pieces — 4 stores of 16 bytes per lane instead of 8 of 8.
            unsigned cw[8][2];
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    bf16x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (bf16)(o[d][r4 * 4 + j] * inv);
                    const uint2 u2 = __builtin_bit_cast(uint2, v);
                    cw[d * 4 + r4][0] = u2.x; cw[d * 4 + r4][1] = u2.y;
                }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const auto r = __builtin_amdgcn_permlane32_swap(cw[j][w], cw[j + 4][w], false, false);
                    cw[j][w] = r[0]; cw[j + 4][w] = r[1];
                }
            if (q < Tq) {
                bf16* op = out + ((int64_t)b * Tq + q) * (int64_t)C + hcur * DH + hi * 32;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint4 piece = {cw[j][0], cw[j][1], cw[j + 4][0], cw[j + 4][1]};
#if defined(ATTN_NT) && (ATTN_NT & 1)
                    { typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                      __builtin_nontemporal_store(u32x4_t{piece.x, piece.y, piece.z, piece.w}, reinterpret_cast<u32x4_t*>(op + 8 * j)); }                                 // A/B: streaming stores of the output rows
#else
                    *reinterpret_cast<uint4*>(op + 8 * j) = piece;
#endif
                }
            }
        } else if (q < Tq) {
            bf16* op = out + ((int64_t)b * Tq + q) * (CROSS ? ldo : (int64_t)C) + hcur * DH;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    if (d * 32 + 8 * r4 >= DH) continue;                          // DH = 48: the second tile's upper half does not exist
                    bf16x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (bf16)(o[d][r4 * 4 + j] * inv);
                    *reinterpret_cast<bf16x4*>(op + d * 32 + 8 * r4 + 4 * hi) = v;   // dims (r&3) + 8*(r>>2) + 4*hi
                }
        }
        }                                                            // hh
    }
}

// --------------------------------------------------------------------------------------------
// Row-resident form with 16-query tiles (v_mfma_f32_16x16x32_bf16): the score row block of a tile is NKT x 8 floats per lane (72 at T = 257),
// the kernel fits 128 VGPRs — 8-wave workgroups, two per CU, FOUR waves per SIMD (the 32-query form above: ~200 VGPRs, two per SIMD, and the
// counters show what that costs: vector and matrix instructions busy 42 % + 23 % of the time, nothing overlapped, a third of the cycles idle).
//   S^T tile (16 keys x 16 queries) = K Q^T: A = K fragment (lane: key l & 15, dims 8 (l >> 4) .. + 7 of the 32-dim step), B = Q fragment (lane: query
//   l & 15, the same dims), D: lane holds keys 4 (l >> 4) .. + 3 of ITS query — row max / sum are in-lane plus one meeting of the four lanes of a query.
//   O^T tile (16 dims x 16 queries) += V^T P^T over 32 keys: B = P of two adjacent key tiles packed in place ([tile 2m: 4 keys | tile 2m + 1: 4 keys]
//   = this lane's 8 k-slots), A = V^T by two transposing reads whose rows are exactly those keys.
// LDS: K rows swizzled as the GEMM's operand stages (slot ^ (row >> 1) & 7: conflict-free for lane -> row l & 15), V rows by 32-byte pairs
// (slot ^ 2 ((row >> 1) & 3): the transposing read of 8 rows x 32 bytes then covers 64 banks once; row-major it was a two-way conflict).
// --------------------------------------------------------------------------------------------
__device__ inline float quad_maxf(float x) {       // over the four lanes {l, l ^ 16, l ^ 32, l ^ 48}
    typedef unsigned u32x2c __attribute__((ext_vector_type(2)));
    const u32x2c r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float y = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const u32x2c q = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ inline float quad_addf(float x) {
    typedef unsigned u32x2c __attribute__((ext_vector_type(2)));
    const u32x2c r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float y = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const u32x2c q = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

template <int NW, int NKT>
__global__ __launch_bounds__(NW * 64, 4) void attn_vit_row16_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, int T, int H, float scale_log2e) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int DH = 64, Tp = NKT * 32, NK16 = 2 * NKT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = H * DH;
    const int64_t ld = 3LL * C;
    const int h = blockIdx.x, b = blockIdx.y;
    const bf16* base = qkv + (int64_t)b * T * ld + h * DH;
    const bf16* kbase = base + C;
    const bf16* vbase = base + 2 * C;
    {   // K and V of the head -> LDS by LDS-DMA, every request up front; the swizzles are applied to the source address (the LDS image is lane-linear)
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const int r8 = lane >> 3, c8 = lane & 7;
        const unsigned rowoff = (unsigned)r8 * (unsigned)(ld * 2);
        const unsigned koff0 = rowoff + (unsigned)((c8 ^ (r8 >> 1)) << 4);                   // rows 16 k .. + 7:  (row >> 1) & 7 = r8 >> 1
        const unsigned koff1 = rowoff + (unsigned)((c8 ^ (4 + (r8 >> 1))) << 4);             // rows 16 k + 8 .. + 15
        const unsigned voff = rowoff + (unsigned)((c8 ^ ((r8 >> 1) << 1)) << 4);
        auto dma_s = [&](const char* src, unsigned off, unsigned dst) {
            unsigned keep;
            const unsigned long long b64 = (unsigned long long)src;
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);
            const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));
            const unsigned long long sb64 = (unsigned long long)lo | ((unsigned long long)hi32 << 32);
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(off), "s"(sb64), "s"(dst) : "memory");
        };
        for (int req = wave_u; req < Tp / 8; req += NW) {
            const int key0 = req * 8;
            if (key0 + 8 <= T) {
                const size_t rbytes = (size_t)key0 * (size_t)(ld * 2);
                dma_s(reinterpret_cast<const char*>(kbase) + rbytes, (req & 1) ? koff1 : koff0, lds0 + req * 1024);
                dma_s(reinterpret_cast<const char*>(vbase) + rbytes, voff, lds0 + Tp * ROWB + req * 1024);
                continue;
            }
            const int key = key0 + r8;                                          // a request reaching past row T - 1: padding rows re-read row T - 1
            const int64_t roff = (int64_t)min(key, T - 1) * ld;
            const bf16* ksrc = kbase + roff + ((c8 ^ ((key >> 1) & 7)) << 3);
            const bf16* vsrc = vbase + roff + ((c8 ^ (((key >> 1) & 3) << 1)) << 3);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(ksrc), "s"(lds0 + req * 1024) : "memory");
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vsrc), "s"(lds0 + Tp * ROWB + req * 1024) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const char* Ks = lds;
    const char* Vs = lds + (size_t)Tp * ROWB;
    const int qc = lane & 15, g = lane >> 4;
    const int nq = (T + 15) >> 4;
    const int QS = gridDim.z;
    const int wv = (wave + h + b) % NW;                               // who takes the odd tiles rotates: co-resident workgroups load different SIMDs
    // transposing read: lanes 4 j + p of a 16-lane group supply row j, 8-byte piece p (4 dims); lane i receives dim i of rows 0 .. 3
    const int i16 = lane & 15;
    const int tr_row = 4 * g + (i16 >> 2);
    const int tr_sw = ((2 * (g & 1) + (i16 >> 3)) & 3) << 1;          // ((row >> 1) & 3) << 1 of that row (the key steps are multiples of 16: no change)
    const int tr_sub = (i16 & 3) >> 1, tr_b8 = (i16 & 1) * 8;
    for (int qt = wv * QS + (int)blockIdx.z; qt < nq; qt += NW * QS) {
        const int q = qt * 16 + qc;
        const bf16* qp = base + (int64_t)min(q, T - 1) * ld + g * 8;
        bf16x8 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 32);
        // ---- pass 1: S^T = K Q^T
        f32x4 s[NK16];
        int qc_k = qc;                                          // (opaque per tile: the K fragments do not depend on the query tile; hoisted they would not fit)
        asm volatile("" : "+v"(qc_k));
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int kt = 0; kt < NK16; ++kt) {
                const int krow = kt * 16 + qc_k;
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + krow * ROWB + (((ks * 4 + g) ^ ((krow >> 1) & 7)) << 4));
                if (ks == 0) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0], z, 0, 0, 0);
                } else
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1], s[kt], 0, 0, 0);
            }
        // s[kt][i]: key = 16 kt + 4 g + i, query = q.  Keys >= T live in the last two key tiles (Tp - 32 < T)
        if (Tp != T) {
#pragma unroll
            for (int kt = NK16 - 2; kt < NK16; ++kt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (kt * 16 + 4 * g + i >= T) s[kt][i] = -INFINITY;
        }
        // ---- pass 2: the exact row maximum
        float mx4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kt = 0; kt < NK16; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) mx4[i] = fmaxf(mx4[i], s[kt][i]);
        const float mx = quad_maxf(fmaxf(fmaxf(mx4[0], mx4[1]), fmaxf(mx4[2], mx4[3])));
        const float mc = mx * scale_log2e;
        // ---- pass 3: p = exp2(s c - m c), row sum in four chains, P packed to bf16
        float ls[4] = {0.f, 0.f, 0.f, 0.f};
        bf16x8 pk[NKT];
#pragma unroll
        for (int m = 0; m < NKT; ++m) {
            float t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                t[i] = __builtin_amdgcn_exp2f(fmaf(s[2 * m + (i >> 2)][i & 3], scale_log2e, -mc));
                ls[i & 3] += t[i];
            }
            pk[m] = pack8(t);
        }
        const float inv = 1.0f / quad_addf((ls[0] + ls[1]) + (ls[2] + ls[3]));
        // ---- pass 4: O^T = V^T P^T, four 16-dim tiles
        f32x4 o[4];
#pragma unroll
        for (int m = 0; m < NKT; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const char* vb = Vs + (m * 32 + tr_row) * ROWB + (((2 * t + tr_sub) ^ tr_sw) << 4) + tr_b8;
                const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(vb));
                const short4v hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(vb + 16 * ROWB));
                typedef short short8v __attribute__((ext_vector_type(8)));
                const bf16x8 vv = __builtin_bit_cast(bf16x8, (short8v)__builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
                if (m == 0) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vv, pk[0], z, 0, 0, 0);
                } else
                    o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vv, pk[m], o[t], 0, 0, 0);
            }
        // ---- store: lane (query, g) holds dims 16 t + 4 g .. + 3 of its row
        if (q < T) {
            bf16* op = out + ((int64_t)b * T + q) * (int64_t)C + h * DH + 4 * g;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf16x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (bf16)(o[t][i] * inv);
                *reinterpret_cast<bf16x4*>(op + 16 * t) = v;
            }
        }
    }
}

int split_for(int pairs, int nq) {          // few (image, head) pairs: query tiles of a pair over several workgroups (SETOK_ATTN_QSPLIT=n forces n)
    const char* e = getenv("SETOK_ATTN_QSPLIT");            // (read per call: tests compare the forms in one process)
    const int forced = e ? atoi(e) : 0;
    const int slots = 2 * cu_count();
    int qs = 1;
    if (forced > 0) qs = forced;
    else if (pairs * 4 <= slots) qs = slots / pairs;
    if (qs > nq) qs = nq;
    return qs < 1 ? 1 : qs;
}

template <int NKT>
int launch_row16(hipStream_t s, const bf16* qkv, bf16* out, int n_imgs, int T, int H, float scale) {
    constexpr int NW = 8;
    const size_t smem = (size_t)NKT * 32 * ROWB * 2;
    static SetokDeviceOnce once;
    if (!once.run([] { return hipFuncSetAttribute((const void*)attn_vit_row16_kernel<NW, NKT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }))
        return setok_fail(SETOK_ELAUNCH, "attn_vit: cannot raise dynamic LDS limit");
    const int qs = split_for(H * n_imgs, (T + 15) >> 4);
    attn_vit_row16_kernel<NW, NKT><<<dim3(H, n_imgs, qs), NW * 64, smem, s>>>(qkv, out, T, H, scale * 1.44269504088896340736f);
    SETOK_CHECK_LAUNCH("setok_attention(vit bf16, row-resident, 16-query tiles)");
    return SETOK_OK;
}

template <int NW, int DH>
int launch(hipStream_t s, const bf16* qkv, bf16* out, int n_imgs, int T, int H, float scale) {
    const int Tp = (T + 31) & ~31;
    const size_t smem = (size_t)Tp * ROWB * 2;
    static SetokDeviceOnce once;                   // one per instantiation; per device inside
    if (!once.run([] { return hipFuncSetAttribute((const void*)attn_vit_kernel<NW, false, DH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }))
        return setok_fail(SETOK_ELAUNCH, "attn_vit: cannot raise dynamic LDS limit");
    // few (image, head) pairs: split the query tiles of a pair over QS workgroups so that the launch reaches ~2 workgroups per CU (SETOK_ATTN_QSPLIT=n forces n)
    const int nq = (T + 31) >> 5;
    int qs = 1;
    {
        const char* e = getenv("SETOK_ATTN_QSPLIT");            // (read per call: tests compare the forms in one process)
        const int forced = e ? atoi(e) : 0;
        const int pairs = H * n_imgs, slots = 2 * cu_count();
        if (forced > 0) qs = forced;
        else if (pairs * 4 <= slots) qs = slots / pairs;
        if (qs > nq) qs = nq;
        if (qs < 1) qs = 1;
    }
    if constexpr (DH == 64 && NW == 8) {                             // experiment: a workgroup per (image, PAIR of heads)
        const char* hp = getenv("SETOK_ATTN_HEADPAIR");
        if (hp && hp[0] == '1' && H % 2 == 0 && 2 * smem <= 160 * 1024) {
            static SetokDeviceOnce once2;
            if (!once2.run([] { return hipFuncSetAttribute((const void*)attn_vit_kernel<NW, false, DH, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }))
                return setok_fail(SETOK_ELAUNCH, "attn_vit: cannot raise dynamic LDS limit");
            attn_vit_kernel<NW, false, DH, 2><<<dim3(H / 2, n_imgs, qs), NW * 64, 2 * smem, s>>>(qkv, out, T, H, scale * 1.44269504088896340736f, 0, nullptr, nullptr,
                                                                                         0, nullptr, 0, 0);
            SETOK_CHECK_LAUNCH("setok_attention(vit bf16, head pairs)");
            return SETOK_OK;
        }
    }
    attn_vit_kernel<NW, false, DH><<<dim3(H, n_imgs, qs), NW * 64, smem, s>>>(qkv, out, T, H, scale * 1.44269504088896340736f, 0, nullptr, nullptr,
                                                                          0, nullptr, 0, 0);
    SETOK_CHECK_LAUNCH("setok_attention(vit bf16)");
    return SETOK_OK;
}

template <int NW>
int launch_cross(hipStream_t s, const bf16* q, int64_t ldq, const bf16* k, const bf16* v, int64_t ldkv, const int32_t* kv_offsets,
                 int n_segs, int q_len, int max_kv, bf16* out, int64_t ldo, int H, float scale) {
    const int Tp = (max_kv + 31) & ~31;
    const size_t smem = (size_t)Tp * ROWB * 2;
    static SetokDeviceOnce once;
    if (!once.run([] { return hipFuncSetAttribute((const void*)attn_vit_kernel<NW, true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; }))
        return setok_fail(SETOK_ELAUNCH, "cross attention: cannot raise dynamic LDS limit");
    attn_vit_kernel<NW, true, 64><<<dim3(H, n_segs), NW * 64, smem, s>>>(q, out, max_kv, H, scale * 1.44269504088896340736f, ldq, k, v, ldkv,
                                                                     kv_offsets, q_len, ldo);
    SETOK_CHECK_LAUNCH("setok_cross_attention(bf16 mfma)");
    return SETOK_OK;
}

}  // namespace

int setok_attention_vit_bf16(hipStream_t s, const bf16* qkv, bf16* out, int n_imgs, int T, int H, int Dh, float scale) {
    if ((Dh != 64 && Dh != 48) || T < 1 || (size_t)((T + 31) & ~31) * ROWB * 2 > 160 * 1024) return SETOK_EUNSUPPORTED;
    const int nq = (T + 31) >> 5;
    // K + V of more than 80 KiB leave room for one workgroup per CU only: then a wave per query tile (up to 12) instead of 8 waves taking
    // two rounds (T = 324, the decoder's 18 x 18 queries: 11 tiles on 8 waves = 2 rounds with 5 idle waves in the second)
    const bool one_wg = (size_t)((T + 31) & ~31) * ROWB * 2 > 80 * 1024;
    if (Dh == 48) {                                                   // the reconstruction decoder's ViT blocks (768 / 16 heads)
        if (one_wg && nq >= 9 && nq <= 12) {
            if (nq == 9) return launch<9, 48>(s, qkv, out, n_imgs, T, H, scale);
            if (nq == 10) return launch<10, 48>(s, qkv, out, n_imgs, T, H, scale);
            if (nq == 11) return launch<11, 48>(s, qkv, out, n_imgs, T, H, scale);
            return launch<12, 48>(s, qkv, out, n_imgs, T, H, scale);
        }
        if (nq >= 8) return launch<8, 48>(s, qkv, out, n_imgs, T, H, scale);
        if (nq >= 4) return launch<4, 48>(s, qkv, out, n_imgs, T, H, scale);
        return launch<1, 48>(s, qkv, out, n_imgs, T, H, scale);
    }
    {   // 129 <= T <= 288 (ViT-L/14-224: 257, ViT-B/16-224: 197), SETOK_ATTN_ROW=1: the row-resident kernel with 16-query tiles.  OPT-IN: per layer it is
        // 171 vs 186 us at 256 images and 5.2 vs 7.4 us at one, but inside the encode step that is 0.3 % (44.50 vs 44.63 ms) and 2 % (1.47 vs 1.50 ms),
        // and its bits differ from the online-softmax kernel's (the exact row maximum instead of a running one) — every golden drift statistic of the
        // step would have to be re-pinned for that.  The choice never depends on the batch: an image's bits must not.
        const char* e = getenv("SETOK_ATTN_ROW");
        if (Dh == 64 && nq >= 5 && nq <= 9 && e && e[0] == '1') {
            switch (nq) {
                case 5: return launch_row16<5>(s, qkv, out, n_imgs, T, H, scale);
                case 6: return launch_row16<6>(s, qkv, out, n_imgs, T, H, scale);
                case 7: return launch_row16<7>(s, qkv, out, n_imgs, T, H, scale);
                case 8: return launch_row16<8>(s, qkv, out, n_imgs, T, H, scale);
                default: return launch_row16<9>(s, qkv, out, n_imgs, T, H, scale);
            }
        }
    }
    // T = 257 (ViT-L/14-224): 8 waves x 32 rows + the class-token row as a second pass of wave 0.  8-wave workgroups
    // fit two per CU (16 waves at 125 VGPRs), 9-wave ones only one: measured 202 vs 242 us per layer.  (Splitting the
    // class-token tile's keys across the 8 waves and merging partial softmaxes through LDS was tried: no gain — the
    // kernel is bound by aggregate VALU/LDS issue, not by the longest wave.)
    if (one_wg && nq >= 13) return launch<12, 64>(s, qkv, out, n_imgs, T, H, scale);   // T = 577 (19 tiles, one workgroup per CU): 12 waves 372 us, 16: 373, 10: 405, 8: 394 (round 3); re-swept after round 6's key-loop changes: 12: 332, 16: 333, 14: 357, 10: 367
    if (nq >= 9) return launch<8, 64>(s, qkv, out, n_imgs, T, H, scale);
    if (nq >= 7) return launch<7, 64>(s, qkv, out, n_imgs, T, H, scale);
    if (nq >= 4) return launch<4, 64>(s, qkv, out, n_imgs, T, H, scale);
    return launch<1, 64>(s, qkv, out, n_imgs, T, H, scale);
}

// Q-Former cross-attention (module.py:283-286,303,342-364): head dim 64, ragged key segments of at most max_kv rows.
int setok_cross_attention_bf16(hipStream_t s, const bf16* q, int64_t ldq, const bf16* k, const bf16* v, int64_t ldkv, const int32_t* kv_offsets,
                               int n_segs, int q_len, int max_kv, bf16* out, int64_t ldo, int H, float scale) {
    if (!kv_offsets || max_kv < 1 || (size_t)((max_kv + 31) & ~31) * ROWB * 2 > 160 * 1024) return SETOK_EUNSUPPORTED;
    const int nq = (q_len + 31) >> 5;
    if (nq >= 8) return launch_cross<8>(s, q, ldq, k, v, ldkv, kv_offsets, n_segs, q_len, max_kv, out, ldo, H, scale);
    if (nq >= 4) return launch_cross<4>(s, q, ldq, k, v, ldkv, kv_offsets, n_segs, q_len, max_kv, out, ldo, H, scale);
    return launch_cross<1>(s, q, ldq, k, v, ldkv, kv_offsets, n_segs, q_len, max_kv, out, ldo, H, scale);
}
