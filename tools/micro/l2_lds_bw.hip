// l2_lds_bw.hip — PROBE (round 4): how many bytes per clock can ONE CU pull from L2 into LDS (global_load_lds_dwordx4, 1 KiB per wave-request) or
// into registers (global_load_dwordx4), as a function of the number of waves issuing and of the requests each wave keeps in flight?  The GEMM's
// 256 x 256 x 64 K-tile needs 64 KiB per 2048 MFMA cycles = 32 B/clk/CU at the full matrix rate; the K loops of rounds 2-4 all settle at 21-25 B/clk.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/l2_lds_bw.hip -o tools/micro/l2_lds_bw && tools/micro/l2_lds_bw
//
// One workgroup per CU (the LDS size forces it), every workgroup of an XCD (blockIdx % 8) walks the SAME 2 MiB window (L2-resident after the first
// pass; windows of different XCDs are disjoint) in 1 KiB pieces, rows of 128 bytes like a GEMM operand panel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int WINDOW = 2 << 20;

template <int DEPTH, bool TO_LDS>
__global__ __launch_bounds__(1024) void bw_kernel(const char* src, int iters, float* sink, unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const char* win = src + (size_t)(blockIdx.x & 7) * WINDOW;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem) + wave * (DEPTH * 1024);
    // piece p of this wave: offset ((p * nw + wave) * 1024 + something per block) mod WINDOW
    unsigned pos = ((unsigned)(blockIdx.x >> 3) * 37u + (unsigned)wave) * 1024u;
    const unsigned stride = (unsigned)nw * 1024u;
    const unsigned voff = (unsigned)lane * 16u;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const unsigned long long b = (unsigned long long)(win + (pos & (WINDOW - 1)));
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
            const unsigned long long sb = (unsigned long long)lo | ((unsigned long long)hi << 32);
            if (TO_LDS) {
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sb), "s"(lds0 + d * 1024) : "memory");
            } else {
                f32x4 t;
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(t) : "v"(voff), "s"(sb) : "memory");
                asm volatile("" :: "v"(t));
            }
            pos += stride;
        }
        // keep DEPTH requests in flight on average: wait for the older half before issuing the next batch
        if (DEPTH >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH / 2) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (TO_LDS) acc[0] = *reinterpret_cast<float*>(smem + tid * 4);
    if (tid == 0) { cycles[blockIdx.x] = t1 - t0; }
    if (acc[0] == 12345.678f) sink[0] = acc[0];
}


// GEMM-shaped access: a workgroup streams a 256-row operand panel K-tile by K-tile — per K-tile 256 rows x 128 bytes, a wave-request = 8 rows x 128 B,
// row pitch `pitch` bytes — the 32 workgroups of an XCD walk 4 panels (8 share one: L2-resident after the first pass).  Question: does the ROW PITCH
// (2 KiB for K = 1024, 8 KiB for K = 4096, 16 KiB for 8192^3) decide how many L2 channels a K-tile's 256 lines fall on?
template <int DEPTH>
__global__ __launch_bounds__(256) void panel_kernel(const char* src, int pitch, int ktiles, int passes, unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t panel_bytes = (size_t)256 * pitch;
    const char* pan = src + ((size_t)(blockIdx.x & 7) * 4 + ((blockIdx.x >> 3) & 3)) * panel_bytes;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem) + wave * (DEPTH * 1024);
    const unsigned voff = (unsigned)(tid & 63) / 8u * (unsigned)pitch + ((unsigned)(tid & 7) << 4);        // 8 rows x 8 slots of 16 B
    int d = 0;
    for (int ps = 0; ps < passes; ++ps)
        for (int t = 0; t < ktiles; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned long long b = (unsigned long long)(pan + (size_t)(i * 32 + wave * 8) * pitch + (size_t)t * 128);
                const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
                const unsigned long long sb = (unsigned long long)lo | ((unsigned long long)hi << 32);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sb), "s"(lds0 + (d % DEPTH) * 1024) : "memory");
                if (++d % (DEPTH / 2) == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH / 2) : "memory");
            }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) cycles[blockIdx.x] = 0;
}

static void run_panel(const char* d_src, int pitch, unsigned long long* d_cyc, int ncu) {
    constexpr int DEPTH = 16;
    const int ktiles = 16, passes = 64;
    const size_t lds = (size_t)4 * DEPTH * 1024;
    CK(hipFuncSetAttribute((const void*)panel_kernel<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        panel_kernel<DEPTH><<<ncu, 256, lds>>>(d_src, pitch, ktiles, passes, d_cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes_per_cu = (double)passes * ktiles * 256 * 128.0;
    printf("panel walk, row pitch %6d B (K-tile = 256 rows x 128 B, 16 KiB in flight per wave): %7.1f GB/s per CU  %6.2f TB/s chip\n", pitch,
           bytes_per_cu / (best * 1e-3) / 1e9, bytes_per_cu * ncu / (best * 1e-3) / 1e12);
    fflush(stdout);
}

template <int DEPTH, bool TO_LDS>
static void run(const char* d_src, int nw, float* sink, unsigned long long* d_cyc, int ncu) {
    const int iters = 4096 / DEPTH;
    const size_t lds = TO_LDS ? (size_t)nw * DEPTH * 1024 : 64;
    if (lds > 160 * 1024) return;
    CK(hipFuncSetAttribute((const void*)bw_kernel<DEPTH, TO_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        bw_kernel<DEPTH, TO_LDS><<<ncu, nw * 64, lds>>>(d_src, iters, sink, d_cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    std::vector<unsigned long long> cyc(ncu);
    CK(hipMemcpy(cyc.data(), d_cyc, ncu * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (auto c : cyc) mean += (double)c; mean /= ncu;
    const double bytes_per_cu = (double)nw * iters * DEPTH * 1024.0;
    printf("%s waves %2d  in flight/wave %2d (%3d KiB/CU): %7.1f GB/s per CU  %6.2f TB/s chip   %5.1f B/clk/CU (s_memtime-free cycle counter: %.0f cycles)\n",
           TO_LDS ? "LDS-DMA " : "to VGPRs", nw, DEPTH, nw * DEPTH, bytes_per_cu / (best * 1e-3) / 1e9, bytes_per_cu * ncu / (best * 1e-3) / 1e12,
           bytes_per_cu / mean, mean);
    fflush(stdout);
}

int main() {
    int ncu = 256; { hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); ncu = p.multiProcessorCount; }
    char* d_src; float* sink; unsigned long long* d_cyc;
    const size_t src_bytes = (size_t)32 * 256 * (16384 + 256);        // 32 panels at the largest pitch
    CK(hipMalloc(&d_src, src_bytes)); CK(hipMemset(d_src, 1, src_bytes));
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&d_cyc, ncu * 8));
    for (int pitch : {2048, 2048 + 128, 2048 + 256, 6144, 8192, 8192 + 128, 16384, 16384 + 128}) run_panel(d_src, pitch, d_cyc, ncu);
    if (getenv("PANEL_ONLY")) return 0;
    for (int nw : {4, 8, 16}) {
        run<2, true>(d_src, nw, sink, d_cyc, ncu);
        run<4, true>(d_src, nw, sink, d_cyc, ncu);
        run<8, true>(d_src, nw, sink, d_cyc, ncu);
        run<16, true>(d_src, nw, sink, d_cyc, ncu);
        run<32, true>(d_src, nw, sink, d_cyc, ncu);
    }
    for (int nw : {4, 8, 16}) {
        run<4, false>(d_src, nw, sink, d_cyc, ncu);
        run<16, false>(d_src, nw, sink, d_cyc, ncu);
    }
    return 0;
}
