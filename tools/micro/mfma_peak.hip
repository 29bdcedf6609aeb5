// mfma_peak.hip — the MFMA rate the chip sustains with NO operand traffic at all (operands and accumulators stay in registers), on all-zero
// and on random bf16 operands: the power-capped ceiling any GEMM on this part lives under.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(512) void mfma_loop(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = src[(tid * 6 + i) & 0xffff];
    for (int i = 0; i < 2; ++i) b[i] = src[(tid * 6 + 4 + i) & 0xffff];
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[tid] = s;
}

typedef __attribute__((ext_vector_type(4))) float f32x4;
// the same register footprint with v_mfma_f32_16x16x32_bf16 (32 accumulator tiles of 4 registers): half the accumulator traffic per FLOP,
// twice the operand reads — does the shape change the power-managed rate?
__global__ __launch_bounds__(512) void mfma_loop16(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = src[(tid * 12 + i) & 0xffff];
    for (int i = 0; i < 4; ++i) b[i] = src[(tid * 12 + 8 + i) & 0xffff];
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[tid] = s;
}

int main(int argc, char** argv) {
    const int iters = 20000, blocks = 256, threads = 512;
    std::vector<unsigned short> h(65536 * 8);
    bf16x8* src; float* out;
    hipMalloc(&src, h.size() * 2); hipMalloc(&out, blocks * threads * 4);
    const bool shape16 = argc > 2;                          // ./mfma_peak smi 16 : the 16x16x32 shape (iters halved: same FLOPs per launch)
    for (int mode = 0; mode < 2; ++mode) {
        srand(1);
        for (auto& v : h) {                                   // mode 0: zeros; mode 1: random bf16 in [-1, 1)
            if (mode == 0) { v = 0; continue; }
            float f = (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16);
        }
        hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&] { if (shape16) mfma_loop16<<<blocks, threads>>>(src, out, iters / 2); else mfma_loop<<<blocks, threads>>>(src, out, iters); };
        for (int rep = 0; rep < 2; ++rep) launch();             // warm-up (and heat-up)
        hipDeviceSynchronize();
        const int reps = 1200;                                 // ~ several seconds: long enough for the power controller to settle
        hipEventRecord(e0);
        for (int rep = 0; rep < reps; ++rep) launch();
        hipEventRecord(e1);
        if (argc > 1) { system("sleep 1; rocm-smi --showclocks --showpower | grep -E 'sclk|Socket'"); }   // sampled WHILE the queue drains
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)reps * blocks * (threads / 64) * iters * 8.0 * 32 * 32 * 16 * 2;
        printf("%s %s operands: %.1f TFLOP/s (%.2f s)\n", shape16 ? "16x16x32" : "32x32x16", mode ? "random" : "zero  ", flops / (ms * 1e-3) / 1e12, ms * 1e-3);
        fflush(stdout);
    }
    return 0;
}
