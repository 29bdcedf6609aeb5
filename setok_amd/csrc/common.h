// common.h — shared device/host helpers for libsetok_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/setok_hip.h"

// The 16-bit element type.  Every 16-bit kernel of the library is written against "a 16-bit float element, fp32 accumulation" under the name
// `bf16` — the type of the BASELINE metric — and nothing in them depends on the exponent width except the places marked SETOK_HALF
// (the two-way splits that feed fp32 start values to the matrix pipe; the round-2 GEMM's start at the bias, which is an exact three-way split in bf16 only).  The library is compiled TWICE from these sources:
//   libsetok_hip.so       bf16 = __bf16,    v_mfma_f32_*_bf16, serves SETOK_F32 + SETOK_BF16
//   libsetok_hip_f16.so   bf16 = _Float16,  v_mfma_f32_*_f16 (the same rate), serves SETOK_F32 + SETOK_F16   (-DSETOK_HALF; round 6)
// — the reference's inference loader and its non-bf16 launches run the tower in torch.float16 (src/model/builder.py:43,135-136,
// src/train/train_setokim.py:326).  In the fp16 build the dtype code the sources compare against (SETOK_BF16) IS SETOK_F16, so that
// library refuses bf16 buffers just as this one refuses fp16 ones.
#ifdef SETOK_HALF
typedef _Float16 bf16;
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 bf16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 bf16x2;
#define SETOK_BF16 SETOK_F16
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define WAVE 64

// ---- error reporting (host) -----------------------------------------------------------------
char* setok_err_buf();                      // thread-local, defined in capi.hip
int setok_fail(int code, const char* fmt, ...);

#define SETOK_CHECK_ARG(cond, ...) \
    do { if (!(cond)) return setok_fail(SETOK_EINVAL, __VA_ARGS__); } while (0)

#define SETOK_CHECK_LAUNCH(what)                                                              \
    do {                                                                                      \
        hipError_t e__ = hipGetLastError();                                                   \
        if (e__ != hipSuccess) return setok_fail(SETOK_ELAUNCH, "%s: %s", what, hipGetErrorString(e__)); \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- optional launch profiler (capi.hip): HIP events on the launch stream around the GEMM and clustering entry points ------------------
// Off by default (one relaxed load per call).  bench.py switches it on around its timed region: `roofline.achieved` is the algorithmic
// work of these launches / their event durations, measured where the launches really happen — inside the library, whichever host calls it.
enum { SETOK_PROF_GEMM_BF16 = 0, SETOK_PROF_GEMM_F32 = 1, SETOK_PROF_CLUSTER = 2 };
bool setok_prof_on();
int setok_prof_begin(hipStream_t s, int kind, int cls, double work, double bytes, bool attach);    // cls: act | residual << 2 | layernorm << 3; returns the record's index
void setok_prof_end(hipStream_t s, int index);
void setok_prof_rows(hipStream_t s, int index, const int32_t* rows_dev, int rows_full, double bytes_fixed);   // the launch processes *rows_dev of the rows_full its record was priced for
void setok_prof_rows_changed();                        // a kernel that writes device-side row counts was launched: the profiler re-reads them for the next launch
// A scope opened with attach = true does not put marker events on the stream: its first launch takes setok_prof_start_event() and its last
// setok_prof_stop_event() through hipExtLaunchKernelGGL, so the timestamps are the dispatches' own (start of the first kernel, end of the
// last) and no barrier packet sits between the launches — marker events around each of the ~110 GEMM calls of a cfg2 step cost 0.55 ms of a
// 45.7 ms step.  Whatever such a scope did not hand out by its end is recorded as a marker then (a path that launches the plain way).
hipEvent_t setok_prof_start_event();                 // of the innermost open attach-scope of this thread, once; else nullptr
hipEvent_t setok_prof_stop_event();
struct SetokProfScope {                              // scopes may nest (the fp32 clustering calls setok_linear for its Gram matrices)
    hipStream_t s; int idx;
    SetokProfScope(hipStream_t s_, int kind, int cls, double work, double bytes, bool attach = false)
        : s(s_), idx(setok_prof_on() ? setok_prof_begin(s_, kind, cls, work, bytes, attach) : -1) {}
    ~SetokProfScope() { if (idx >= 0) setok_prof_end(s, idx); }
};
// kernel<<<grid, block, lds, s>>>(args...), with the dispatch's own start / stop timestamps going to the events when there are any
#if defined(__HIPCC__)
#include <hip/hip_ext.h>
template <typename K, typename... A>
static inline void setok_launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, hipEvent_t e0, hipEvent_t e1, A... args) {
    if (e0 || e1) hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)lds, s, e0, e1, 0u, args...);
    else kernel<<<grid, block, lds, s>>>(args...);
}
#endif

// ---- per-device one-time setup (host) ---------------------------------------------------------
// hipFuncSetAttribute (dynamic-LDS limit) is a property of (function, DEVICE); a process-wide `static bool` would skip it on a second
// device of the same process and races between threads.  One atomic bit per device: the setup is idempotent, so two threads doing it at
// the same time is harmless; nothing else in the library keeps mutable global state.
struct SetokDeviceOnce {
    unsigned long long done = 0;            // bit d: done on device d (devices >= 64 redo it on every call)
    template <class F>
    bool run(F&& setup) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (dev >= 0 && dev < 64 && ((__atomic_load_n(&done, __ATOMIC_ACQUIRE) >> dev) & 1ull)) return true;
        if (!setup()) return false;
        if (dev >= 0 && dev < 64) __atomic_fetch_or(&done, 1ull << dev, __ATOMIC_RELEASE);
        return true;
    }
};

// A small per-device cache of one value (device symbol address, CU count): slot d holds device d's value.
template <class T>
struct SetokPerDevice {
    T val[64] = {};
    unsigned long long have = 0;
    template <class F>
    bool get(T& out, F&& query) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        const bool slot = dev >= 0 && dev < 64;
        if (slot && ((__atomic_load_n(&have, __ATOMIC_ACQUIRE) >> dev) & 1ull)) { out = val[dev]; return true; }
        if (!query(out)) return false;
        if (slot) { val[dev] = out; __atomic_fetch_or(&have, 1ull << dev, __ATOMIC_RELEASE); }
        return true;
    }
};

// ---- element access templated on the storage type ------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VEC = 4;                       // elements per 16-byte access
    __device__ static inline float ld(const float* p) { return *p; }
    __device__ static inline void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16> {
    static constexpr int VEC = 8;
    __device__ static inline float ld(const bf16* p) { return (float)*p; }
    __device__ static inline void st(bf16* p, float v) { *p = (bf16)v; }
};

// 16-byte vector load/store of VEC elements as floats
template <typename T> __device__ inline void ld_vec(const T* p, float* out);
template <> __device__ inline void ld_vec<float>(const float* p, float* out) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
}
template <> __device__ inline void ld_vec<bf16>(const bf16* p, float* out) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
}
template <typename T> __device__ inline void st_vec(T* p, const float* in);
template <> __device__ inline void st_vec<float>(float* p, const float* in) {
    f32x4 v = {in[0], in[1], in[2], in[3]};
    *reinterpret_cast<f32x4*>(p) = v;
}
template <> __device__ inline void st_vec<bf16>(bf16* p, const float* in) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)in[i];
    *reinterpret_cast<bf16x8*>(p) = v;
}

// ---- wave-level reductions (64 lanes) -------------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Exact-erf GELU for the bf16 GEMM epilogues: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below one bf16 ulp of the result) on
// v_rcp / v_exp — libm's erff cost 29 k cycles per 256 x 256 tile in the epilogue against 9.5 k for quick_gelu.  The negative branch uses
// 0.5 * poly * e directly (no 1 - (1 - small) cancellation).  The fp32 parity kernels keep erff.
__device__ inline float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float half_tail = 0.5f * poly * __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);      // 0.5 * (1 - erf(|z|))
    return x * (x >= 0.f ? 1.0f - half_tail : half_tail);
}

// ---- the counter-based dropout mask (glue.hip: setok_dropout explains it): one SplitMix64 finaliser per four consecutive counters ----------
__device__ inline unsigned long long dropout_word(unsigned long long seed, unsigned long long group) {
    unsigned long long z = seed + group * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
__device__ inline bool dropout_keep(unsigned long long seed, unsigned long long ctr, unsigned thresh16) {
    return (unsigned)((dropout_word(seed, ctr >> 2) >> (16 * (unsigned)(ctr & 3))) & 0xffffu) >= thresh16;
}


// ---- LayerNorm folded into the consuming GEMM: the rank-2 start of the accumulators as two-way bf16 splits (gemm_persist.hip) ----------
#ifdef SETOK_HALF
__device__ inline void split2(float x, bf16& hi, bf16& lo) {           // fp16 build: x = hi + lo + O(2^-22 |x|), both by rounding (|x| < 65504: row means, 1 / rstd, column sums)
    hi = (bf16)x;
    lo = (bf16)(x - (float)hi);
}
#else
__device__ inline void split2(float x, bf16& hi, bf16& lo) {           // x = hi + lo + O(2^-16 |x|), both by truncation
    const unsigned u = __builtin_bit_cast(unsigned, x);
    hi = __builtin_bit_cast(bf16, (unsigned short)(u >> 16));
    const float r = x - __builtin_bit_cast(float, u & 0xffff0000u);
    lo = __builtin_bit_cast(bf16, (unsigned short)(__builtin_bit_cast(unsigned, r) >> 16));
}
#endif
__device__ inline bf16x8 ln_col_frag(float c, float b) {                // the W-side operand's k-slots 0-7 for one output column
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (bf16)0.0f;
    {
        bf16 ch, cl, bh, bl;
        split2(c, ch, cl); split2(b, bh, bl);
        f[0] = ch; f[1] = ch; f[2] = cl; f[3] = cl; f[4] = bh; f[5] = bh; f[6] = bl; f[7] = bl;
    }
    return f;
}
__device__ inline bf16x8 ln_row_frag(float mean, float rstd) {          // the activation-side operand's k-slots 0-7 for one output row
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (bf16)0.0f;
    {
        bf16 mh, ml, sh, sl;
        split2(-mean, mh, ml); split2(1.0f / rstd, sh, sl);
        f[0] = mh; f[1] = ml; f[2] = mh; f[3] = ml; f[4] = sh; f[5] = sl; f[6] = sh; f[7] = sl;
    }
    return f;
}

// SwiGLU on one (gate, up) pair of fp32 accumulator values with torch's 16-bit rounding points (HF LlamaMLP in the element type: gate_proj / up_proj outputs rounded,
// act_fn's result rounded, the product rounded): the ONE definition behind the fused GEMM epilogue (gemm_persist.hip) and setok_swiglu / setok_swiglu_pairs (llama.hip), so a
// row's bits do not depend on which of them produced it.  16-bit types: v_exp_f32 / v_rcp_f32 (1 ulp-class, far below the type's rounding); fp32 keeps expf and the division.
template <typename T>
__device__ inline T swiglu16(float gate, float up) {
    const float g = (float)(T)gate, u = (float)(T)up;
    float s;
    if constexpr (sizeof(T) == 4) s = g / (1.0f + expf(-g));
    else s = g * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * g));
    return (T)((float)(T)s * u);
}
constexpr int SETOK_ACT_SWIGLU_PAIRS = 3;      // internal (not an `act` of setok_linear): the ping-pong GEMM's SwiGLU epilogue, reached through setok_linear_swiglu only

__device__ inline float act_apply(float v, int act) {
    if (act == SETOK_ACT_QUICK_GELU) return v / (1.0f + expf(-1.702f * v));
    if (act == SETOK_ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    return v;
}
