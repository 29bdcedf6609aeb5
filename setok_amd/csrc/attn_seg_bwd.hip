// attn_seg_bwd.hip — bf16 MFMA backward of the head's 2 x 512 attention for segments of more than 32 rows (the inter encoder: an image's L_i
// cluster tokens; up to a few hundred at 336^2).  The generic backward (backward.hip) gives every row a wave and re-reads the segment per
// row — 10 of the 11 ms the training step of cfg4 spends in attention backward are these long segments.  Same decomposition as the forward
// kernel (attn_seg.hip): a workgroup per (segment, head), 4 waves x 128 head dims, partial score tiles summed through LDS in a fixed order,
// tiles aligned to the segment's first row (bit-exact batch invariance), no atomics.
//   kernel Q : per 32-query tile — D_q = dO_q . O_q, log-sum-exp (pass 1), then (pass 2) S^T and dP^T = V dO^T tiles with the QUERY in the
//              lane, dS^T = P^T (dP^T - D) scale, dQ^T += K^T dS^T (K^T through the transposing LDS reads).  Writes lse / D for kernel KV.
//   kernel KV: per 32-key tile — S and dP = dO V^T tiles with the KEY in the lane (operands swapped), P, dS; dV^T += dO^T P,
//              dK^T += Q^T dS (dO^T and Q^T through the transposing reads).
// With s = scale q.k:  ds = p (dp - D) scale,  dq = sum_k ds k,  dk = sum_q ds q,  dv = sum_q p do.
#include "common.h"

namespace {

constexpr int SD = 512, WD = 128, VROW = WD * 2;
typedef __attribute__((ext_vector_type(4))) short short4v;

__device__ inline bf16x8 pk8(const float* p) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)p[i];
    return v;
}

// wave-private 32 x 128 slice (rows r0 + t*32 .., dims of this wave) of a row-major matrix -> LDS (8 x 1 KiB LDS-DMA pieces)
__device__ inline void stage_slice(const bf16* base, int64_t ld, int t, int n, unsigned lds, int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int p = i * 64 + lane, row = t * 32 + (p >> 4), c = p & 15;
        const bf16* src = base + (int64_t)min(row, n - 1) * ld + c * 8;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds + i * 1024) : "memory");
    }
}

// acc^T[d-tile][lane column] += X^T (through transposing reads of the wave's LDS slice) * packed lane-owned values (p0: rows 0-15, p1: 16-31)
__device__ inline void acc_xT(f32x16 (&o)[4], const char* Xw, int tr_row, int tr_col, const bf16x8& p0, const bf16x8& p1) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const char* va = Xw + (k2 * 16 + tr_row) * VROW + (d * 32 + tr_col) * 2;
            const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(va));
            const short4v hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(va + 8 * VROW));
            union { short s8[8]; bf16x8 v; } u;
#pragma unroll
            for (int j = 0; j < 4; ++j) { u.s8[j] = lo[j]; u.s8[4 + j] = hi4[j]; }
            o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u.v, k2 == 0 ? p0 : p1, o[d], 0, 0, 0);
        }
    }
}

__device__ inline void store_T(bf16* op, const f32x16 (&o)[4], int hi) {            // o^T tiles -> 128 dims of one row
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            bf16x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (bf16)o[d][r4 * 4 + j];
            *reinterpret_cast<bf16x4*>(op + d * 32 + 8 * r4 + 4 * hi) = v;
        }
}

// 8 k-steps of the wave's 128-dim slice: acc += A-rows . B-rows^T; both operands are "row fragments" (lane = row & 31, 16 bytes at hi * 8)
__device__ inline void rows_dot(f32x16& acc, const bf16* a_row, const bf16x8 (&b)[8]) {
    bf16x8 a[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) a[ks] = *reinterpret_cast<const bf16x8*>(a_row + ks * 16);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], b[ks], acc, 0, 0, 0);
}

__device__ inline void zero16(f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_seg_bwd_q_kernel(const bf16* __restrict__ qkv, const int32_t* __restrict__ seg_offsets,
                                                             const bf16* __restrict__ out, const bf16* __restrict__ dout, bf16* __restrict__ dqkv,
                                                             float* __restrict__ lse_ws, float* __restrict__ d_ws, int H, float scale) {
    __shared__ __attribute__((aligned(16))) float Sp[4][16 * 64];
    __shared__ __attribute__((aligned(16))) float Dp[4][16 * 64];
    __shared__ __attribute__((aligned(16))) char Ks[4][32 * VROW];
    const int s = blockIdx.x, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = seg_offsets[s];
    const int n = seg_offsets[s + 1] - r0;
    if (n <= 32) return;
    const int nt = (n + 31) >> 5;
    const float c = scale * 1.44269504088896340736f;
    const int64_t C = (int64_t)H * SD, ld = 3 * C;
    const bf16* qb = qkv + (int64_t)r0 * ld + h * SD + wave * WD;
    const bf16* kb = qb + C;
    const bf16* vb = qb + 2 * C;
    const bf16* ob = out + (int64_t)r0 * C + h * SD + wave * WD;
    const bf16* gb = dout + (int64_t)r0 * C + h * SD + wave * WD;
    const int qi = lane & 31, hi = lane >> 5;
    const int g16 = lane >> 4, i16 = lane & 15;
    const int tr_row = (i16 >> 2) + 4 * (g16 >> 1);
    const int tr_col = (g16 & 1) * 16 + (i16 & 3) * 4;
    const unsigned klds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)&Ks[0][0]) + wave * (32 * VROW);

    for (int qt = 0; qt < nt; ++qt) {
        const int q = qt * 32 + qi, qc = min(q, n - 1);
        bf16x8 qf[8], gf[8];
        float dpart = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(qb + (int64_t)qc * ld + hi * 8 + ks * 16);
            gf[ks] = *reinterpret_cast<const bf16x8*>(gb + (int64_t)qc * C + hi * 8 + ks * 16);
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(ob + (int64_t)qc * C + hi * 8 + ks * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) dpart = fmaf((float)gf[ks][e], (float)of[e], dpart);
        }
        dpart += __shfl_xor(dpart, 32, 64);                           // the two lanes of a query cover the wave's 128 dims
        Sp[wave][lane] = dpart;
        __syncthreads();
        const float Dq = (Sp[0][lane] + Sp[1][lane]) + (Sp[2][lane] + Sp[3][lane]);
        __syncthreads();

        // ---- pass 1: log-sum-exp of the query's row of scores
        float m_run = -INFINITY, l_run = 0.f;
        for (int kt = 0; kt < nt; ++kt) {
            f32x16 sp; zero16(sp);
            rows_dot(sp, kb + (int64_t)min(kt * 32 + qi, n - 1) * ld + hi * 8, qf);
#pragma unroll
            for (int r = 0; r < 16; ++r) Sp[wave][r * 64 + lane] = sp[r];
            __syncthreads();
            float mx = -INFINITY, t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                t[r] = (Sp[0][r * 64 + lane] + Sp[1][r * 64 + lane]) + (Sp[2][r * 64 + lane] + Sp[3][r * 64 + lane]);
                if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= n) t[r] = -INFINITY;
                mx = fmaxf(mx, t[r]);
            }
            __syncthreads();
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            float ls = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) ls += __builtin_amdgcn_exp2f((t[r] - m_new) * c);
            l_run = l_run * __builtin_amdgcn_exp2f((m_run - m_new) * c) + ls;
            m_run = m_new;
        }
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float lse2 = m_run * c + __builtin_amdgcn_logf(l_tot);   // log2-domain log-sum-exp: p = exp2(s c - lse2)
        if (wave == 0 && hi == 0 && q < n) { lse_ws[((int64_t)r0 + q) * H + h] = lse2; d_ws[((int64_t)r0 + q) * H + h] = Dq; }

        // ---- pass 2: dQ^T += K^T dS^T over the key tiles
        f32x16 dq[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) zero16(dq[d]);
        for (int kt = 0; kt < nt; ++kt) {
            stage_slice(kb, ld, kt, n, klds, lane);
            const int kr = min(kt * 32 + qi, n - 1);
            f32x16 sp, dp; zero16(sp); zero16(dp);
            rows_dot(sp, kb + (int64_t)kr * ld + hi * 8, qf);
            rows_dot(dp, vb + (int64_t)kr * ld + hi * 8, gf);
#pragma unroll
            for (int r = 0; r < 16; ++r) { Sp[wave][r * 64 + lane] = sp[r]; Dp[wave][r * 64 + lane] = dp[r]; }
            __syncthreads();
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sv = (Sp[0][r * 64 + lane] + Sp[1][r * 64 + lane]) + (Sp[2][r * 64 + lane] + Sp[3][r * 64 + lane]);
                const float dv = (Dp[0][r * 64 + lane] + Dp[1][r * 64 + lane]) + (Dp[2][r * 64 + lane] + Dp[3][r * 64 + lane]);
                const bool live = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi < n;
                const float p = live ? __builtin_amdgcn_exp2f(fmaf(sv, c, -lse2)) : 0.f;
                t[r] = p * (dv - Dq) * scale;
            }
            __syncthreads();
            const bf16x8 p0 = pk8(t), p1 = pk8(t + 8);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the K slice has landed (wave-private)
            acc_xT(dq, &Ks[wave][0], tr_row, tr_col, p0, p1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // reads done before the next tile's DMA
        }
        if (q < n) store_T(dqkv + ((int64_t)r0 + q) * ld + h * SD + wave * WD, dq, hi);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
constexpr int KV_LDS = 2 * 4 * 16 * 64 * 4 + 2 * 4 * 32 * VROW;        // Sp, Dp, Q slices, dO slices = 98304 bytes

__global__ __launch_bounds__(256) void attn_seg_bwd_kv_kernel(const bf16* __restrict__ qkv, const int32_t* __restrict__ seg_offsets,
                                                              const bf16* __restrict__ dout, bf16* __restrict__ dqkv, const float* __restrict__ lse_ws,
                                                              const float* __restrict__ d_ws, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float (*Sp)[16 * 64] = reinterpret_cast<float (*)[16 * 64]>(smem);
    float (*Dp)[16 * 64] = reinterpret_cast<float (*)[16 * 64]>(smem + 4 * 16 * 64 * 4);
    char* Qs = smem + 2 * 4 * 16 * 64 * 4;
    char* Gs = Qs + 4 * 32 * VROW;
    const int s = blockIdx.x, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = seg_offsets[s];
    const int n = seg_offsets[s + 1] - r0;
    if (n <= 32) return;
    const int nt = (n + 31) >> 5;
    const float c = scale * 1.44269504088896340736f;
    const int64_t C = (int64_t)H * SD, ld = 3 * C;
    const bf16* qb = qkv + (int64_t)r0 * ld + h * SD + wave * WD;
    const bf16* kb = qb + C;
    const bf16* vb = qb + 2 * C;
    const bf16* gb = dout + (int64_t)r0 * C + h * SD + wave * WD;
    const int ki = lane & 31, hi = lane >> 5;
    const int g16 = lane >> 4, i16 = lane & 15;
    const int tr_row = (i16 >> 2) + 4 * (g16 >> 1);
    const int tr_col = (g16 & 1) * 16 + (i16 & 3) * 4;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    const unsigned qlds = lds_base + 2 * 4 * 16 * 64 * 4 + wave * (32 * VROW);
    const unsigned glds = qlds + 4 * 32 * VROW;
    const char* Qw = Qs + wave * (32 * VROW);
    const char* Gw = Gs + wave * (32 * VROW);

    for (int kt = 0; kt < nt; ++kt) {
        const int k = kt * 32 + ki, kc = min(k, n - 1);
        bf16x8 kf[8], vf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(kb + (int64_t)kc * ld + hi * 8 + ks * 16);
            vf[ks] = *reinterpret_cast<const bf16x8*>(vb + (int64_t)kc * ld + hi * 8 + ks * 16);
        }
        f32x16 dk[4], dv[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) { zero16(dk[d]); zero16(dv[d]); }
        for (int qt = 0; qt < nt; ++qt) {
            stage_slice(qb, ld, qt, n, qlds, lane);
            stage_slice(gb, C, qt, n, glds, lane);
            const int qr = min(qt * 32 + ki, n - 1);                    // as an A-operand row: lane & 31 indexes the QUERY here
            f32x16 sp, dp; zero16(sp); zero16(dp);
            rows_dot(sp, qb + (int64_t)qr * ld + hi * 8, kf);           // S[query (registers)][key (lane)]
            rows_dot(dp, gb + (int64_t)qr * C + hi * 8, vf);            // dP[query][key] = dO_q . V_k
#pragma unroll
            for (int r = 0; r < 16; ++r) { Sp[wave][r * 64 + lane] = sp[r]; Dp[wave][r * 64 + lane] = dp[r]; }
            __syncthreads();
            float tp[16], td[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sv = (Sp[0][r * 64 + lane] + Sp[1][r * 64 + lane]) + (Sp[2][r * 64 + lane] + Sp[3][r * 64 + lane]);
                const float dvv = (Dp[0][r * 64 + lane] + Dp[1][r * 64 + lane]) + (Dp[2][r * 64 + lane] + Dp[3][r * 64 + lane]);
                const int qq = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const bool live = qq < n && k < n;
                const int64_t wi = ((int64_t)r0 + min(qq, n - 1)) * H + h;
                const float p = live ? __builtin_amdgcn_exp2f(fmaf(sv, c, -lse_ws[wi])) : 0.f;
                tp[r] = p;
                td[r] = p * (dvv - d_ws[wi]) * scale;
            }
            __syncthreads();
            const bf16x8 p0 = pk8(tp), p1 = pk8(tp + 8), s0 = pk8(td), s1 = pk8(td + 8);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's Q and dO slices have landed
            acc_xT(dv, Gw, tr_row, tr_col, p0, p1);                     // dV^T += dO^T P
            acc_xT(dk, Qw, tr_row, tr_col, s0, s1);                     // dK^T += Q^T dS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (k < n) {
            store_T(dqkv + ((int64_t)r0 + k) * ld + C + h * SD + wave * WD, dk, hi);
            store_T(dqkv + ((int64_t)r0 + k) * ld + 2 * C + h * SD + wave * WD, dv, hi);
        }
    }
}

}  // namespace

// Long segments (> 32 rows) of setok_attention_bwd's bf16 / head dim 512 case; the generic kernels keep the short ones.
int setok_attention_bwd_seg_bf16(hipStream_t s, const bf16* qkv, const int32_t* seg_offsets, int n_segs, const bf16* out, const bf16* dout,
                                 bf16* dqkv, float* lse_ws, float* d_ws, int H, int Dh, float scale) {
    if (Dh != SD || !seg_offsets || n_segs <= 0) return SETOK_EUNSUPPORTED;
    static SetokDeviceOnce once;
    if (!once.run([] { return hipFuncSetAttribute((const void*)attn_seg_bwd_kv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, KV_LDS) == hipSuccess; }))
        return setok_fail(SETOK_ELAUNCH, "setok_attention_bwd: cannot raise the dynamic LDS limit");
    attn_seg_bwd_q_kernel<<<dim3(n_segs, H), 256, 0, s>>>(qkv, seg_offsets, out, dout, dqkv, lse_ws, d_ws, H, scale);
    attn_seg_bwd_kv_kernel<<<dim3(n_segs, H), 256, KV_LDS, s>>>(qkv, seg_offsets, dout, dqkv, lse_ws, d_ws, H, scale);
    SETOK_CHECK_LAUNCH("setok_attention_bwd(segments bf16)");
    return SETOK_OK;
}
