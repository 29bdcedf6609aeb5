// llama.hip — what the LLM prefill of BASELINE config 5 needs beyond setok_linear: the pieces of HuggingFace LlamaModel.forward that
// SetokimLlamaForCausalLM.forward (src/model/language_model/setokim_llama.py:130-139) runs on the spliced embeddings.
//   setok_rmsnorm           LlamaRMSNorm (fp32 statistics, normalised value rounded to the activation dtype BEFORE the weight multiply)
//   setok_rope              apply_rotary_pos_emb on the q and k thirds of a fused [q | k | v] buffer (rotate_half convention, cos / sin in fp32
//                           rounded to the activation dtype, every elementwise product / sum rounded like the eager bf16 graph)
//   setok_swiglu            act_fn(gate) * up on a fused [gate | up] buffer
//   setok_attention_causal  causal + key-padding-masked attention over uniform sequences: MFMA kernel for bf16 / head dim 128 (keys and
//                           values streamed through LDS in tiles of 32, online softmax in registers), generic wave-per-row kernel otherwise
#include "common.h"

namespace {

template <typename T> __device__ inline float rnd(float v) { return (float)(T)v; }      // one rounding to the activation dtype

// ---- RMSNorm ---------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, int rows, int C,
                                                      float eps) {
    constexpr int V = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + (int64_t)row * C;
    float buf[V], s = 0.f;
    for (int c = lane * V; c < C; c += 64 * V) {
        ld_vec<T>(xr + c, buf);
#pragma unroll
        for (int i = 0; i < V; ++i) s += buf[i] * buf[i];
    }
    const float rstd = rsqrtf(wave_sum(s) / (float)C + eps);
    for (int c = lane * V; c < C; c += 64 * V) {
        ld_vec<T>(xr + c, buf);
#pragma unroll
        for (int i = 0; i < V; ++i) buf[i] = rnd<T>(w[c + i]) * rnd<T>(buf[i] * rstd);   // weight * hidden.to(input_dtype)
        st_vec<T>(y + (int64_t)row * C + c, buf);
    }
}

// Rows of NCH * 64 * VEC elements (Llama hidden 4096 bf16 = 8 chunks per lane): the row stays in registers between the two passes and a wave
// keeps its columns of the weight (already rounded to the activation dtype) across the rows it walks.  Same arithmetic and summation order.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, int rows, float eps) {
    constexpr int V = Elem<T>::VEC, C = NCH * 64 * V;
    const int lane = threadIdx.x & 63;
    const int wave0 = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    float wr[NCH][V];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int i = 0; i < V; ++i) wr[k][i] = rnd<T>(w[(k * 64 + lane) * V + i]);
    for (int row = wave0; row < rows; row += nwaves) {
        const T* xr = x + (int64_t)row * C;
        float buf[NCH][V], s = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) ld_vec<T>(xr + (k * 64 + lane) * V, buf[k]);
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int i = 0; i < V; ++i) s += buf[k][i] * buf[k][i];
        const float rstd = rsqrtf(wave_sum(s) / (float)C + eps);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
#pragma unroll
            for (int i = 0; i < V; ++i) buf[k][i] = wr[k][i] * rnd<T>(buf[k][i] * rstd);
            st_vec<T>(y + (int64_t)row * C + (k * 64 + lane) * V, buf[k]);
        }
    }
}

// ---- rotary embedding on q and k of [q | k | v] rows -----------------------------------------------------------------------------------------
// One thread per (row, chunk of V dims of the first half): cos / sin of its V angles are computed ONCE and reused for the row's 2H heads
// (the angle depends on the position and the dim only), every head costs two 16-byte loads and two 16-byte stores.  Grouped-query attention:
// H query heads, Hkv <= H key heads, rows of (H + 2 Hkv) Dh elements.
template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(T* __restrict__ qkv, const int64_t* __restrict__ pos, int rows, int H, int Hkv, int Dh, float log2_theta) {
    constexpr int V = Elem<T>::VEC;
    const int half = Dh >> 1, chunks = half / V;
    const int64_t total = (int64_t)rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % chunks);
        const int64_t r = i / chunks;
        const float p = (float)pos[r];
        float c[V], s[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float ang = p * (1.0f / exp2f(log2_theta * (float)(2 * (ch * V + e)) / (float)Dh));
            c[e] = rnd<T>(cosf(ang)); s[e] = rnd<T>(sinf(ang));
        }
        T* base = qkv + r * (int64_t)((H + 2 * Hkv) * Dh) + ch * V;
        for (int hh = 0; hh < H + Hkv; ++hh) {                            // q heads, then k heads: contiguous in [q | k | v]
            T* q = base + (int64_t)hh * Dh;
            float x1[V], x2[V], o1[V], o2[V];
            ld_vec<T>(q, x1); ld_vec<T>(q + half, x2);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                o1[e] = rnd<T>(x1[e] * c[e]) + rnd<T>(-x2[e] * s[e]);      // q * cos + rotate_half(q) * sin, each op rounded
                o2[e] = rnd<T>(x2[e] * c[e]) + rnd<T>(x1[e] * s[e]);
            }
            st_vec<T>(q, o1); st_vec<T>(q + half, o2);
        }
    }
}

// scalar fallback for head dims whose half is not a multiple of the vector width
template <typename T>
__global__ void rope_scalar_kernel(T* __restrict__ qkv, const int64_t* __restrict__ pos, int rows, int H, int Hkv, int Dh, float log2_theta) {
    const int half = Dh >> 1;
    const int HR = H + Hkv;
    const int64_t total = (int64_t)rows * HR * half;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % half);
        const int hh = (int)((i / half) % HR);
        const int64_t r = i / ((int64_t)half * HR);
        const float ang = (float)pos[r] * (1.0f / exp2f(log2_theta * (float)(2 * d) / (float)Dh));
        const float c = rnd<T>(cosf(ang)), s = rnd<T>(sinf(ang));
        T* p = qkv + r * (int64_t)((H + 2 * Hkv) * Dh) + (int64_t)hh * Dh + d;
        const float x1 = (float)p[0], x2 = (float)p[half];
        p[0] = (T)(rnd<T>(x1 * c) + rnd<T>(-x2 * s));
        p[half] = (T)(rnd<T>(x2 * c) + rnd<T>(x1 * s));
    }
}

// ---- SwiGLU: V elements per thread, 16-byte accesses -----------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void swiglu_kernel(const T* __restrict__ gu, T* __restrict__ out, int64_t rows, int F) {
    constexpr int V = Elem<T>::VEC;
    const int chunks = F / V;
    const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / chunks; const int c = (int)(i % chunks) * V;
        float g[V], u[V];
        ld_vec<T>(gu + r * 2 * F + c, g); ld_vec<T>(gu + r * 2 * F + F + c, u);
#pragma unroll
        for (int e = 0; e < V; ++e) g[e] = (float)swiglu16<T>(g[e], u[e]);
        st_vec<T>(out + r * F + c, g);
    }
}

// ... on INTERLEAVED (gate_j, up_j) pairs: the layout the fused gate|up GEMM takes its weights in (setok_linear_swiglu), for the rows it leaves to the unfused pair
template <typename T>
__global__ __launch_bounds__(256) void swiglu_pairs_kernel(const T* __restrict__ gu, T* __restrict__ out, int64_t rows, int F) {
    constexpr int V = Elem<T>::VEC;                                   // V outputs per thread from 2 V inputs: two 16-byte loads, one 16-byte store
    const int chunks = F / V;
    const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / chunks; const int c = (int)(i % chunks) * V;
        float a[V], b[V], o[V];
        ld_vec<T>(gu + r * 2 * F + 2 * c, a); ld_vec<T>(gu + r * 2 * F + 2 * c + V, b);
#pragma unroll
        for (int e = 0; e < V / 2; ++e) { o[e] = (float)swiglu16<T>(a[2 * e], a[2 * e + 1]); o[V / 2 + e] = (float)swiglu16<T>(b[2 * e], b[2 * e + 1]); }
        st_vec<T>(out + r * F + c, o);
    }
}

// ---- generic causal attention: one wave per (query row, head) ------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void attn_causal_generic_kernel(const T* __restrict__ qkv, const uint8_t* __restrict__ kmask, T* __restrict__ out,
                                                                 int Tn, int H, int Hkv, int Dh, float scale) {
    extern __shared__ float ps[];                                      // Tn scores
    const int row = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int b = row / Tn, i = row % Tn;
    const int64_t C = (int64_t)H * Dh, ld = (int64_t)(H + 2 * Hkv) * Dh;
    const int hk = h / (H / Hkv);                                      // grouped-query attention: H / Hkv query heads share a key / value head (repeat_kv)
    const int64_t KO = C + (int64_t)hk * Dh - (int64_t)h * Dh, VO = C + (int64_t)Hkv * Dh + (int64_t)hk * Dh - (int64_t)h * Dh;   // from `base` to the key / value head
    const T* base = qkv + (int64_t)b * Tn * ld + h * Dh;
    float mx = -INFINITY;
    for (int j = lane; j <= i; j += 64) {
        float acc = -INFINITY;
        if (!kmask || kmask[(int64_t)b * Tn + j]) {
            acc = 0.f;
            for (int d = 0; d < Dh; ++d) acc = fmaf((float)base[(int64_t)i * ld + d], (float)base[(int64_t)j * ld + KO + d], acc);
            acc *= scale;
        }
        ps[j] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j <= i; j += 64) { const float e = mx == -INFINITY ? 0.f : expf(ps[j] - mx); ps[j] = e; sum += e; }
    sum = wave_sum(sum);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    for (int d = lane; d < Dh; d += 64) {
        float o = 0.f;
        for (int j = 0; j <= i; ++j) o = fmaf(rnd<T>(ps[j] * inv), (float)base[(int64_t)j * ld + VO + d], o);   // probabilities cast to dtype (HF)
        out[((int64_t)b * Tn + i) * C + h * Dh + d] = (T)o;
    }
}

// ---- bf16 MFMA causal attention, head dim 128 --------------------------------------------------------------------------------------------------
constexpr int CD = 128;                  // head dim
constexpr int CROW = CD * 2;             // bytes per K / V row in LDS
constexpr int CQ = 128;                  // queries per workgroup (4 waves x 32)
typedef __attribute__((ext_vector_type(4))) short short4v;

__device__ inline bf16x8 pack8c(const float* p) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)p[i];
    return v;
}

__global__ __launch_bounds__(256) void attn_causal_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ kmask, bf16* __restrict__ out,
                                                          int Tn, int H, int Hkv, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) char Ks[2][32 * CROW];       // K tile, 16-byte slots XOR-swizzled by (row & 15)
    __shared__ __attribute__((aligned(16))) char Vs[2][32 * CROW];       // V tile, row-major (hardware-transposing reads)
    const int qb = (int)gridDim.x - 1 - (int)blockIdx.x;                 // heavy (late) query blocks first: the causal triangle
    const int h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qi = lane & 31, hi = lane >> 5;
    const int64_t C = (int64_t)H * CD, ld = (int64_t)(H + 2 * Hkv) * CD;
    const bf16* base = qkv + (int64_t)b * Tn * ld + h * CD;
    const int hk = h / (H / Hkv);                                        // grouped-query attention: the key / value head of this query head
    const int64_t KO = C + (int64_t)(hk - h) * CD, VO = C + (int64_t)Hkv * CD + (int64_t)(hk - h) * CD;
    const uint8_t* km = kmask ? kmask + (int64_t)b * Tn : nullptr;
    const int q0 = qb * CQ + wave * 32;                                  // this wave's first query
    const int q = q0 + qi;
    const bf16* qp = base + (int64_t)min(q, Tn - 1) * ld + hi * 8;
    bf16x8 qf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
    f32x16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    constexpr float NEG = -1.0e30f;                                      // finite sentinel: a query may meet only masked keys first (left padding)
    float m_run = NEG, l_run = 0.f;
    const int g16 = lane >> 4, i16 = lane & 15;
    const int tr_row = (i16 >> 2) + 4 * (g16 >> 1);
    const int tr_col = (g16 & 1) * 16 + (i16 & 3) * 4;
    const unsigned klds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)&Ks[0][0]) + wave * 1024;
    const unsigned vlds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)&Vs[0][0]) + wave * 1024;
    const int kend = min(qb * CQ + CQ, Tn);                              // keys this block can see: [0, kend)
    const int nkt = (kend + 31) >> 5;
    // (round 6) The counters said 31 vector instructions per MFMA in this kernel — 490 per (wave, key tile) against ~70 in the ViT attention: per tile every thread formed
    // four 64-bit source addresses with clamps, and every score went through the mask logic although all but the diagonal and the last tile of a block are unmasked.
    // Whole tiles: a wave-uniform 64-bit base (scalar adds per tile) + two constant 32-bit lane offsets per operand; unmasked tiles: no mask arithmetic (same bits).
    unsigned koffs[2], voffs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = i * 256 + tid, row = p >> 4, c = p & 15;
        koffs[i] = (unsigned)row * (unsigned)(ld * 2) + (unsigned)((c ^ (row & 15)) << 4);
        voffs[i] = (unsigned)row * (unsigned)(ld * 2) + (unsigned)(c << 4);
    }
    auto dma_sb = [&](const char* sbase, unsigned off, unsigned dst) {
        unsigned keep;
        const unsigned long long b64 = (unsigned long long)sbase;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);
        const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));
        const unsigned long long sb64 = (unsigned long long)lo | ((unsigned long long)hi32 << 32);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(sb64), "s"(dst) : "memory");
    };
    auto stage = [&](int kt, int buf) {                                  // K and V tile kt -> LDS buffer buf: 512 pieces of 16 B each, 2 per thread
        if (kt * 32 + 32 <= Tn) {                                        // (uniform) all 32 rows exist
            const char* tb = reinterpret_cast<const char*>(base) + (size_t)kt * 32 * (size_t)(ld * 2);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                dma_sb(tb + KO * 2, koffs[i], klds + buf * (32 * CROW) + i * 4096);
                dma_sb(tb + VO * 2, voffs[i], vlds + buf * (32 * CROW) + i * 4096);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = i * 256 + tid, row = p >> 4, c = p & 15;
            const bf16* src = base + (int64_t)min(kt * 32 + row, Tn - 1) * ld;
            const bf16* ksrc = src + KO + ((c ^ (row & 15)) << 3);       // physical slot c of a row holds logical chunk c ^ (row & 15)
            const bf16* vsrc = src + VO + (c << 3);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(ksrc), "s"(klds + buf * (32 * CROW) + i * 4096) : "memory");
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vsrc), "s"(vlds + buf * (32 * CROW) + i * 4096) : "memory");
        }
    };
    stage(0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this thread's pieces of tile kt
        __syncthreads();                                                 // everyone's; everyone is done with the other buffer
        if (kt + 1 < nkt) stage(kt + 1, buf ^ 1);
        const int k0 = kt * 32;
        if (k0 > q0 + 31) continue;                                      // wave-uniform: the whole tile lies in this wave's future
        // key-padding mask of the tile as a 32-bit set (bit j = key k0 + j is a token)
        unsigned kbits = 0xffffffffu;
        if (km) kbits = (unsigned)__ballot(lane < 32 && k0 + lane < Tn && km[min(k0 + lane, Tn - 1)] != 0);
        else if (k0 + 32 > Tn) kbits = (unsigned)__ballot(lane < 32 && k0 + lane < Tn);
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const char* Kb = &Ks[buf][0];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kb + qi * CROW + (((ks * 2 + hi) ^ (qi & 15)) << 4));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
        }
        float t[16];
        float mx = NEG;
        const bool diag = k0 + 31 > q0;                                  // some key of the tile may lie after some query of the wave
        const bool plain = !diag && kbits == 0xffffffffu;                // (wave-uniform) every key of the tile is a token every query of the wave may see
        if (plain) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { t[r] = s[r]; mx = fmaxf(mx, t[r]); }
        } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kj = (r & 3) + 8 * (r >> 2) + 4 * hi;              // key index inside the tile
            t[r] = s[r];
            if (!((kbits >> kj) & 1u) || (diag && k0 + kj > q)) t[r] = NEG;
            mx = fmaxf(mx, t[r]);
        }
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
        if (plain) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { t[r] = __builtin_amdgcn_exp2f(fmaf(t[r], scale_log2e, -mc)); ls += t[r]; }
        } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) { t[r] = t[r] <= NEG ? 0.f : __builtin_amdgcn_exp2f(fmaf(t[r], scale_log2e, -mc)); ls += t[r]; }
        }
        l_run += ls;
        const bf16x8 p0 = pack8c(t), p1 = pack8c(t + 8);
        const char* Vb = &Vs[buf][0];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const char* va = Vb + (k2 * 16 + tr_row) * CROW + (d * 32 + tr_col) * 2;
                const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(va));
                const short4v hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(va + 8 * CROW));
                union { short s8[8]; bf16x8 v; } u;
#pragma unroll
                for (int j = 0; j < 4; ++j) { u.s8[j] = lo[j]; u.s8[4 + j] = hi4[j]; }
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u.v, k2 == 0 ? p0 : p1, o[d], 0, 0, 0);
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;                  // a query without any visible token (padding before the first token): zeros
    if (q < Tn) {
        bf16* op = out + ((int64_t)b * Tn + q) * C + h * CD;
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

}  // namespace

// ---- language-model loss -------------------------------------------------------------------------------------------------------------------
// setokim_llama.py:145-160: logits.float(); position t predicts token t + 1; positions whose NEXT token is padding (attention_mask[t + 1] == 0)
// are dropped, CrossEntropyLoss() then ignores label -100 and averages over the rest.  One workgroup per (sequence, position): a dropped /
// ignored position costs nothing (its logits row is never read); the others make two passes over their row (max, then sum of exponentials:
// the re-read hits L2) in fp32.  A label outside [0, V) that is not the ignore index yields NaN (torch raises there).
template <typename T>
__global__ __launch_bounds__(256) void lm_loss_rows_kernel(const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                           const uint8_t* __restrict__ amask, int Tn, int V, int ignore_index,
                                                           float* __restrict__ loss_row, float* __restrict__ valid_row) {
    constexpr int VE = Elem<T>::VEC;
    const int row = blockIdx.x, t = row % Tn, tid = threadIdx.x;
    __shared__ float red[4];
    bool valid = t + 1 < Tn;
    int64_t target = 0;
    if (valid) {
        target = labels[row + 1];
        valid = (!amask || amask[row + 1] != 0) && target != (int64_t)ignore_index;
    }
    if (!valid) {
        if (tid == 0) { loss_row[row] = 0.f; valid_row[row] = 0.f; }
        return;
    }
    const T* x0 = logits + (int64_t)row * ld;
    // 16-byte loads need a 16-byte-aligned start: a vocabulary resized for the image tokens (32002, 32003: initialize_vision_tokenizer) gives
    // contiguous logits rows that start anywhere, so a row is read as [scalar head up to the next 16-byte boundary | vectors | scalar tail]
    const int head = min(V, (int)(((16u - (unsigned)((size_t)x0 & 15u)) & 15u) / sizeof(T)));
    const T* x = x0 + head;
    const int Vb = V - head;
    const int nvec = Vb / VE;
    float buf[VE];
    float m = -INFINITY;
    if (tid < head) m = Elem<T>::ld(x0 + tid);
    for (int c = tid; c < nvec; c += 256) {
        ld_vec<T>(x + (int64_t)c * VE, buf);
#pragma unroll
        for (int i = 0; i < VE; ++i) m = fmaxf(m, buf[i]);
    }
    for (int c = nvec * VE + tid; c < Vb; c += 256) m = fmaxf(m, Elem<T>::ld(x + c));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    if (tid < head) sum = expf(Elem<T>::ld(x0 + tid) - m);
    for (int c = tid; c < nvec; c += 256) {
        ld_vec<T>(x + (int64_t)c * VE, buf);
#pragma unroll
        for (int i = 0; i < VE; ++i) sum += expf(buf[i] - m);
    }
    for (int c = nvec * VE + tid; c < Vb; c += 256) sum += expf(Elem<T>::ld(x + c) - m);
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) {
        const float tot = (red[0] + red[1]) + (red[2] + red[3]);
        const bool in_range = target >= 0 && target < (int64_t)V;
        loss_row[row] = in_range ? (m + logf(tot)) - Elem<T>::ld(x0 + target) : NAN;
        valid_row[row] = 1.f;
    }
}

// fixed-order mean of the per-position losses: thread i sums rows i, i + 256, ... then a tree over the 256 partial sums
__global__ __launch_bounds__(256) void lm_loss_reduce_kernel(const float* __restrict__ loss_row, const float* __restrict__ valid_row, int rows,
                                                             float* __restrict__ out) {
    __shared__ float sl[256], sc[256];
    const int tid = threadIdx.x;
    float a = 0.f, c = 0.f;
    for (int r = tid; r < rows; r += 256) { a += loss_row[r]; c += valid_row[r]; }
    sl[tid] = a; sc[tid] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { sl[tid] += sl[tid + o]; sc[tid] += sc[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { out[0] = sl[0] / sc[0]; out[1] = sc[0]; }        // 0 / 0 = NaN when nothing is valid: what torch's mean over an empty set gives
}

#define LL_DISPATCH(NAME, CALL_BF16, CALL_F32)                                 \
    if (dtype == SETOK_BF16) { CALL_BF16; }                                    \
    else if (dtype == SETOK_F32) { CALL_F32; }                                 \
    else return setok_fail(SETOK_EINVAL, NAME ": bad dtype %d", dtype);

extern "C" int setok_rmsnorm(void* stream, int dtype, const void* x, const float* weight, void* y, int rows, int C, float eps) {
    SETOK_CHECK_ARG(x && weight && y, "setok_rmsnorm: null operand");
    SETOK_CHECK_ARG(rows >= 0 && C > 0 && C % 8 == 0, "setok_rmsnorm: C=%d must be a positive multiple of 8", C);
    if (rows == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SETOK_BF16 && (C == 4096 || C == 5120 || C == 2048) && rows >= 1024) {
        const int grid = min(cdiv(rows, 4), 256 * 8);
        if (C == 4096) rmsnorm_rows_kernel<bf16, 8><<<grid, 256, 0, s>>>((const bf16*)x, weight, (bf16*)y, rows, eps);
        else if (C == 5120) rmsnorm_rows_kernel<bf16, 10><<<grid, 256, 0, s>>>((const bf16*)x, weight, (bf16*)y, rows, eps);
        else rmsnorm_rows_kernel<bf16, 4><<<grid, 256, 0, s>>>((const bf16*)x, weight, (bf16*)y, rows, eps);
        SETOK_CHECK_LAUNCH("setok_rmsnorm");
        return SETOK_OK;
    }
    LL_DISPATCH("setok_rmsnorm", (rmsnorm_kernel<bf16><<<cdiv(rows, 4), 256, 0, s>>>((const bf16*)x, weight, (bf16*)y, rows, C, eps)),
                (rmsnorm_kernel<float><<<cdiv(rows, 4), 256, 0, s>>>((const float*)x, weight, (float*)y, rows, C, eps)));
    SETOK_CHECK_LAUNCH("setok_rmsnorm");
    return SETOK_OK;
}

extern "C" int setok_rope_gqa(void* stream, int dtype, void* qkv, const int64_t* position_ids, int rows, int H, int Hkv, int Dh, float theta) {
    SETOK_CHECK_ARG(qkv && position_ids, "setok_rope: null operand");
    SETOK_CHECK_ARG(rows >= 0 && H > 0 && Hkv > 0 && H % Hkv == 0 && Dh > 0 && Dh % 2 == 0 && theta > 0.f, "setok_rope: bad shape (H=%d Hkv=%d Dh=%d)", H, Hkv, Dh);
    if (rows == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    const float l2 = log2f(theta);
    const int V = dtype == SETOK_BF16 ? 8 : 4;
    if ((Dh / 2) % V == 0) {
        const int64_t total = (int64_t)rows * (Dh / 2 / V);
        const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
        LL_DISPATCH("setok_rope", (rope_kernel<bf16><<<grid, 256, 0, s>>>((bf16*)qkv, position_ids, rows, H, Hkv, Dh, l2)),
                    (rope_kernel<float><<<grid, 256, 0, s>>>((float*)qkv, position_ids, rows, H, Hkv, Dh, l2)));
    } else {
        const int64_t total = (int64_t)rows * (H + Hkv) * (Dh / 2);
        const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
        LL_DISPATCH("setok_rope", (rope_scalar_kernel<bf16><<<grid, 256, 0, s>>>((bf16*)qkv, position_ids, rows, H, Hkv, Dh, l2)),
                    (rope_scalar_kernel<float><<<grid, 256, 0, s>>>((float*)qkv, position_ids, rows, H, Hkv, Dh, l2)));
    }
    SETOK_CHECK_LAUNCH("setok_rope");
    return SETOK_OK;
}

extern "C" int setok_rope(void* stream, int dtype, void* qkv, const int64_t* position_ids, int rows, int H, int Dh, float theta) {
    return setok_rope_gqa(stream, dtype, qkv, position_ids, rows, H, H, Dh, theta);
}

extern "C" int setok_swiglu(void* stream, int dtype, const void* gate_up, void* out, int64_t rows, int F) {
    SETOK_CHECK_ARG(gate_up && out && rows >= 0 && F > 0 && F % 8 == 0, "setok_swiglu: bad operand (F must be a multiple of 8)");
    if (rows == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = rows * (F / (dtype == SETOK_BF16 ? 8 : 4));
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    LL_DISPATCH("setok_swiglu", (swiglu_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)gate_up, (bf16*)out, rows, F)),
                (swiglu_kernel<float><<<grid, 256, 0, s>>>((const float*)gate_up, (float*)out, rows, F)));
    SETOK_CHECK_LAUNCH("setok_swiglu");
    return SETOK_OK;
}

extern "C" int setok_swiglu_pairs(void* stream, int dtype, const void* gate_up_pairs, void* out, int64_t rows, int F) {
    SETOK_CHECK_ARG(gate_up_pairs && out && rows >= 0 && F > 0 && F % 8 == 0, "setok_swiglu_pairs: bad operand (F must be a multiple of 8)");
    if (rows == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = rows * (F / (dtype == SETOK_BF16 ? 8 : 4));
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    LL_DISPATCH("setok_swiglu_pairs", (swiglu_pairs_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)gate_up_pairs, (bf16*)out, rows, F)),
                (swiglu_pairs_kernel<float><<<grid, 256, 0, s>>>((const float*)gate_up_pairs, (float*)out, rows, F)));
    SETOK_CHECK_LAUNCH("setok_swiglu_pairs");
    return SETOK_OK;
}

int setok_gemm_swiglu_bf16(hipStream_t s, const bf16* A, int64_t lda, const bf16* W, bf16* C, int64_t ldc, int M, int N, int K);   // gemm_persist.hip

extern "C" int setok_linear_swiglu(void* stream, int dtype, const void* A, int64_t lda, const void* W_pairs, void* out, int64_t ldo, int M, int F, int K) {
    SETOK_CHECK_ARG(A && W_pairs && out, "setok_linear_swiglu: null operand");
    SETOK_CHECK_ARG(M >= 0 && F > 0 && K > 0 && lda >= K && ldo >= F && lda % 8 == 0 && ldo % 8 == 0, "setok_linear_swiglu: bad shape M=%d F=%d K=%d", M, F, K);
    if (dtype != SETOK_BF16) return setok_fail(SETOK_EUNSUPPORTED, "setok_linear_swiglu: 16-bit element types only (the caller runs setok_linear + setok_swiglu_pairs in fp32)");
    if (M == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    SetokProfScope prof(s, SETOK_PROF_GEMM_BF16, 0, 2.0 * M * (2.0 * F) * K, ((double)M * K + 2.0 * F * K) * 2.0 + (double)M * F * 2.0, true);
    const int rc = setok_gemm_swiglu_bf16(s, (const bf16*)A, lda, (const bf16*)W_pairs, (bf16*)out, ldo, M, 2 * F, K);
    if (rc == SETOK_EUNSUPPORTED)
        return setok_fail(SETOK_EUNSUPPORTED, "setok_linear_swiglu: needs M %% 256 == 0, (2 F) %% 256 == 0, K %% 64 == 0, K >= 128 (M=%d F=%d K=%d): "
                                              "send other rows through setok_linear + setok_swiglu_pairs (identical bits)", M, F, K);
    return rc;
}

extern "C" int setok_attention_causal_gqa(void* stream, int dtype, const void* qkv, const uint8_t* key_mask, void* out, int B, int T, int H, int Hkv,
                                          int Dh, float scale) {
    SETOK_CHECK_ARG(qkv && out, "setok_attention_causal: null operand");
    SETOK_CHECK_ARG(B >= 0 && T > 0 && H > 0 && Hkv > 0 && H % Hkv == 0 && Dh > 0 && Dh % 8 == 0,
                    "setok_attention_causal: bad shape B=%d T=%d H=%d Hkv=%d Dh=%d", B, T, H, Hkv, Dh);
    if (B == 0) return SETOK_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SETOK_BF16 && Dh == CD) {
        attn_causal_kernel<<<dim3(cdiv(T, CQ), H, B), 256, 0, s>>>((const bf16*)qkv, key_mask, (bf16*)out, T, H, Hkv, scale * 1.44269504088896340736f);
        SETOK_CHECK_LAUNCH("setok_attention_causal(bf16 mfma)");
        return SETOK_OK;
    }
    const size_t smem = (size_t)T * sizeof(float);
    SETOK_CHECK_ARG(smem <= 64 * 1024, "setok_attention_causal: T=%d too long for the generic kernel", T);
    dim3 grid(B * T, H);
    LL_DISPATCH("setok_attention_causal",
                (attn_causal_generic_kernel<bf16><<<grid, 64, smem, s>>>((const bf16*)qkv, key_mask, (bf16*)out, T, H, Hkv, Dh, scale)),
                (attn_causal_generic_kernel<float><<<grid, 64, smem, s>>>((const float*)qkv, key_mask, (float*)out, T, H, Hkv, Dh, scale)));
    SETOK_CHECK_LAUNCH("setok_attention_causal");
    return SETOK_OK;
}

extern "C" int setok_attention_causal(void* stream, int dtype, const void* qkv, const uint8_t* key_mask, void* out, int B, int T, int H, int Dh,
                                      float scale) {
    return setok_attention_causal_gqa(stream, dtype, qkv, key_mask, out, B, T, H, H, Dh, scale);
}

extern "C" int setok_lm_loss(void* stream, int dtype, const void* logits, int64_t ld, const int64_t* labels, const uint8_t* attention_mask, int B,
                             int T, int V, int ignore_index, float* row_ws, float* out) {
    SETOK_CHECK_ARG(logits && labels && row_ws && out, "setok_lm_loss: null operand");
    SETOK_CHECK_ARG(B >= 0 && T > 0 && V > 0 && ld >= V, "setok_lm_loss: bad shape B=%d T=%d V=%d", B, T, V);
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * T;
    if (rows > 0) {
        LL_DISPATCH("setok_lm_loss", (lm_loss_rows_kernel<bf16><<<rows, 256, 0, s>>>((const bf16*)logits, ld, labels, attention_mask, T, V, ignore_index, row_ws, row_ws + rows)),
                    (lm_loss_rows_kernel<float><<<rows, 256, 0, s>>>((const float*)logits, ld, labels, attention_mask, T, V, ignore_index, row_ws, row_ws + rows)));
        SETOK_CHECK_LAUNCH("setok_lm_loss(rows)");
    }
    lm_loss_reduce_kernel<<<1, 256, 0, s>>>(row_ws, row_ws + rows, rows, out);
    SETOK_CHECK_LAUNCH("setok_lm_loss(reduce)");
    return SETOK_OK;
}
