// attn_seg.hip — bf16 MFMA attention over RAGGED segments at head dim 512: the `Attention` of the SeTok head's cluster encoders
// (module.py:61-73 with dim 1024 / 2 heads; inner_encoder: one segment per cluster, a handful of tokens each, tokenizer.py:147-150;
// inter_encoder: one segment per image, its L_i cluster tokens, tokenizer.py:179).
//
// The generic kernel (norm_attn.hip) gives every (query row, head) its own wave, which re-reads the segment's K and V once per
// row: (1 + 2n) KB per row and head — 2 GB through L2 per call at cfg2.  Here a workgroup (or, for short segments, a wave) owns a
// (segment, head) and reads its Q, K, V once; tiles are aligned to the segment's first row, so the arithmetic of a segment does not
// depend on its position in the batch (bit-exact batch invariance).  The big-segment kernel:
//   * 4 waves split the 512 head dims, 128 each.  For a (32-query, 32-key) tile every wave multiplies its 128-dim slice
//     (8 x v_mfma_f32_32x32x16_bf16, K and Q fragments straight from global memory — a lane's fragment is 16 contiguous bytes of
//     one row), the four partial S^T tiles meet in LDS and every wave sums them in the same order (identical softmax in all four);
//   * S^T = K Q^T, so a lane owns one query row: online softmax in registers, P packed to bf16 in place as the B operand (the
//     attn_vit.hip scheme);
//   * O^T += V^T P^T on the wave's own 128 output dims: its V slice of the key tile (32 x 256 B) is DMA'd into wave-private LDS and
//     read back through the hardware-transposing ds_read_b64_tr_b16.
// Segment traffic: Q, K, V read once per 32-query tile, O written once.
#include "common.h"

namespace {

constexpr int SD = 512;                 // head dim
constexpr int WD = 128;                 // dims per wave
constexpr int VROW = WD * 2;            // bytes per V row slice in LDS

typedef __attribute__((ext_vector_type(4))) short short4v;

__device__ inline bf16x8 pack8s(const float* p) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)p[i];
    return v;
}

// ---- segments of more than 32 rows (the inter encoder: an image's L_i cluster tokens): one workgroup per (segment, head); query and
//      key tiles are aligned to the segment's first row, so a segment's arithmetic never depends on where it sits in the batch ----------
__device__ __forceinline__ void attn_seg_big_segment(const bf16* __restrict__ qkv, const int32_t* __restrict__ seg_offsets, bf16* __restrict__ out, int H,
                                                     float scale_log2e, const int s, const int h, float (*Sp)[16 * 64], char (*Vs)[32 * VROW]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = seg_offsets[s];
    const int n = seg_offsets[s + 1] - r0;
    if (n <= 32) return;                                                          // the small-segment kernel's share
    const int nt = (n + 31) >> 5;
    const int64_t C = (int64_t)H * SD, ld = 3 * C;
    const bf16* qb = qkv + (int64_t)r0 * ld + h * SD + wave * WD;
    const bf16* kb = qb + C;
    const bf16* vb = qb + 2 * C;
    const int qi = lane & 31, hi = lane >> 5;
    const int g16 = lane >> 4, i16 = lane & 15;
    const int tr_row = (i16 >> 2) + 4 * (g16 >> 1);
    const int tr_col = (g16 & 1) * 16 + (i16 & 3) * 4;
    const unsigned vlds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)&Vs[0][0]) + wave * (32 * VROW);
    float* mySp = &Sp[wave][0];

    for (int qt = 0; qt < nt; ++qt) {
        const int q = qt * 32 + qi;
        const bf16* qp = qb + (int64_t)min(q, n - 1) * ld + hi * 8;
        bf16x8 qf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
        f32x16 o[4];
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;

        for (int kt = 0; kt < nt; ++kt) {
            // this wave's V slice of the key tile -> LDS (8 x 1 KiB; piece p = 64 i + lane: row p >> 4, 16-byte column p & 15); the previous
            // tile's transposing reads of the same region have retired (lgkmcnt(0) below)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int p = i * 64 + lane, key = kt * 32 + (p >> 4), c = p & 15;
                const bf16* src = vb + (int64_t)min(key, n - 1) * ld + c * 8;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(vlds + i * 1024) : "memory");
            }
            f32x16 sp;
#pragma unroll
            for (int r = 0; r < 16; ++r) sp[r] = 0.f;
            const bf16* kp = kb + (int64_t)min(kt * 32 + qi, n - 1) * ld + hi * 8;
            bf16x8 kf[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(kp + ks * 16);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) sp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], sp, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) mySp[r * 64 + lane] = sp[r];
            __syncthreads();
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = (Sp[0][r * 64 + lane] + Sp[1][r * 64 + lane]) + (Sp[2][r * 64 + lane] + Sp[3][r * 64 + lane]);
            __syncthreads();                                           // everyone has read the partials before the next tile overwrites them
            // t[r]: key = kt*32 + (r&3) + 8*(r>>2) + 4*hi, query = q
            float mx = -INFINITY;
            const bool tail = (kt == nt - 1) && (n & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (tail) { const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi; if (key >= n) t[r] = -INFINITY; }
                mx = fmaxf(mx, t[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            if (!__all(m_new == m_run)) {
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
                m_run = m_new;
            }
            const float mc = m_run * scale_log2e;
            float ls = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { t[r] = __builtin_amdgcn_exp2f(fmaf(t[r], scale_log2e, -mc)); ls += t[r]; }
            l_run += ls;
            const bf16x8 p0 = pack8s(t), p1 = pack8s(t + 8);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the V slice has landed (wave-private: no barrier needed)
            const char* Vw = &Vs[wave][0];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const char* va = Vw + (k2 * 16 + tr_row) * VROW + (d * 32 + tr_col) * 2;
                    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(va));
                    const short4v hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(va + 8 * VROW));
                    union { short s8[8]; bf16x8 v; } u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { u.s8[j] = lo[j]; u.s8[4 + j] = hi4[j]; }
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u.v, k2 == 0 ? p0 : p1, o[d], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the transposing reads are done before the next tile's DMA lands
        }
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        if (q < n) {
            bf16* op = out + ((int64_t)r0 + q) * C + h * SD + wave * WD;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    bf16x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (bf16)(o[d][r4 * 4 + j] * inv);
                    *reinterpret_cast<bf16x4*>(op + d * 32 + 8 * r4 + 4 * hi) = v;
                }
        }
    }
}

// SPW = 1: one workgroup per (segment, head) — the inter encoder's B segments.  SPW = 64: the workgroup looks at 64 consecutive segment lengths with ONE
// coalesced load per wave and runs the (rare) long ones among them — the inner encoder's worst-case launch: B * N segments, ~8 % of them not empty and
// hardly any longer than 32 rows; as one workgroup per segment that was 131 072 workgroups that load two words and exit, 56 us per call at batch 256.
template <int SPW>
__global__ __launch_bounds__(256) void attn_seg_big_kernel(const bf16* __restrict__ qkv, const int32_t* __restrict__ seg_offsets, int n_segs,
                                                           bf16* __restrict__ out, int H, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) float Sp[4][16 * 64];                 // the waves' partial S^T tiles, [reg][lane]
    __shared__ __attribute__((aligned(16))) char Vs[4][32 * VROW];                // wave-private V slices of the current key tile
    if constexpr (SPW == 1) {
        attn_seg_big_segment(qkv, seg_offsets, out, H, scale_log2e, blockIdx.x, blockIdx.y, Sp, Vs);
    } else {
        static_assert(SPW == 64, "one lane per segment length");
        const int sl = blockIdx.x * 64 + (threadIdx.x & 63);
        const int nl = sl < n_segs ? seg_offsets[sl + 1] - seg_offsets[sl] : 0;
        unsigned long long m = __ballot(nl > 32);                                 // (the same mask in all four waves: uniform control flow around the barriers)
        while (m) {
            const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
            m &= m - 1;
            attn_seg_big_segment(qkv, seg_offsets, out, H, scale_log2e, blockIdx.x * 64 + b, blockIdx.y, Sp, Vs);
            __syncthreads();                                                      // the next segment reuses Sp / Vs
        }
    }
}

// ---- segments of at most 32 rows (the inner encoder's clusters: 7 tokens on average, 3 at 336^2): ONE WAVE per (segment, head), four
//      independent waves per workgroup, no barrier anywhere.  The wave runs the whole 512-dim contraction of the single 32 x 32 score tile
//      (32 MFMAs, in the same dim order and with the same four-way grouping as the big kernel's LDS sum), one softmax, and the PV product
//      in four 128-dim passes through its private 8 KiB of LDS.
// (164 registers = three waves per SIMD.  Asked to fit four / five — amdgpu_waves_per_eu, 106 / 102 registers, no spills — the kernel is 2 % / 8 % SLOWER inside
//  the step: it is not short of waves; its 64 fragment loads per task touch 32 lines of 128 bytes each for 32 used bytes per line.)
__global__ __launch_bounds__(256) void attn_seg_small_kernel(const bf16* __restrict__ qkv, const int32_t* __restrict__ seg_offsets, int n_segs,
                                                             bf16* __restrict__ out, int H, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) char Vs[4][32 * VROW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = blockIdx.x * 4 + wave, h = blockIdx.y;
    if (s >= n_segs) return;
    const int r0 = seg_offsets[s];
    const int n = seg_offsets[s + 1] - r0;
    if (n <= 0 || n > 32) return;
    const int64_t C = (int64_t)H * SD, ld = 3 * C;
    const bf16* qb = qkv + (int64_t)r0 * ld + h * SD;
    const int qi = lane & 31, hi = lane >> 5;
    const int g16 = lane >> 4, i16 = lane & 15;
    const int tr_row = (i16 >> 2) + 4 * (g16 >> 1);
    const int tr_col = (g16 & 1) * 16 + (i16 & 3) * 4;
    const unsigned vlds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)&Vs[0][0]) + wave * (32 * VROW);
    const bf16* rowp = qb + (int64_t)min(qi, n - 1) * ld + hi * 8;              // this lane's row, as a query and as a key
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = 0.f;
    f32x16 part[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {                                                 // the four 128-dim slices, summed as (0 + 1) + (2 + 3) below
        bf16x8 qf[8], kf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(rowp + w * WD + ks * 16);
            kf[ks] = *reinterpret_cast<const bf16x8*>(rowp + C + w * WD + ks * 16);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) part[w][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) part[w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], part[w], 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        t[r] = (part[0][r] + part[1][r]) + (part[2][r] + part[3][r]);
        const int key = (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (key >= n) t[r] = -INFINITY;
        mx = fmaxf(mx, t[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * scale_log2e;
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { t[r] = __builtin_amdgcn_exp2f(fmaf(t[r], scale_log2e, -mc)); ls += t[r]; }
    const float inv = 1.0f / (ls + __shfl_xor(ls, 32, 64));
    const bf16x8 p0 = pack8s(t), p1 = pack8s(t + 8);
    const char* Vw = &Vs[wave][0];
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        const bf16* vb = qb + 2 * C + w * WD;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = i * 64 + lane, key = p >> 4, c = p & 15;
            const bf16* src = vb + (int64_t)min(key, n - 1) * ld + c * 8;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(vlds + i * 1024) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x16 o[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const char* va = Vw + (k2 * 16 + tr_row) * VROW + (d * 32 + tr_col) * 2;
                const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(va));
                const short4v hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(va + 8 * VROW));
                union { short s8[8]; bf16x8 v; } u;
#pragma unroll
                for (int j = 0; j < 4; ++j) { u.s8[j] = lo[j]; u.s8[4 + j] = hi4[j]; }
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u.v, k2 == 0 ? p0 : p1, o[d], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // reads done before the next pass's DMA overwrites the slice
        if (qi < n) {
            bf16* op = out + ((int64_t)r0 + qi) * C + h * SD + w * WD;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    bf16x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (bf16)(o[d][r4 * 4 + j] * inv);
                    *reinterpret_cast<bf16x4*>(op + d * 32 + 8 * r4 + 4 * hi) = v;
                }
        }
    }
}

}  // namespace

// setok_attention's bf16 path for head dim 512 over ragged segments (the SeTok head).
int setok_attention_seg_bf16(hipStream_t s, const bf16* qkv, const int32_t* seg_offsets, int n_segs, int max_len, bf16* out, int rows, int H, int Dh, float scale) {
    if (Dh != SD || !seg_offsets || n_segs <= 0 || rows <= 0) return SETOK_EUNSUPPORTED;
    // every segment belongs to exactly one of the two kernels (n <= 32 / n > 32); each exits at once on the other's segments
    attn_seg_small_kernel<<<dim3(cdiv(n_segs, 4), H), 256, 0, s>>>(qkv, seg_offsets, n_segs, out, H, scale * 1.44269504088896340736f);
    if (max_len > 32) {
        if (n_segs > 2048) attn_seg_big_kernel<64><<<dim3(cdiv(n_segs, 64), H), 256, 0, s>>>(qkv, seg_offsets, n_segs, out, H, scale * 1.44269504088896340736f);
        else attn_seg_big_kernel<1><<<dim3(n_segs, H), 256, 0, s>>>(qkv, seg_offsets, n_segs, out, H, scale * 1.44269504088896340736f);
    }
    SETOK_CHECK_LAUNCH("setok_attention(segments bf16)");
    return SETOK_OK;
}
