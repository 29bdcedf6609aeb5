// launch_gap.hip — what does a dependent kernel boundary cost on this chip, by what the two kernels are?  (round 6)
// The step timeline (profiles/r05_step_timeline.txt) shows 4.5 us behind every GEMM launch, 6.5-9 us in front of one and 0.0 between small
// kernels.  This probe launches N dependent kernels back to back on one stream; every workgroup spins until `spin_us` after ITS start (100 MHz
// s_memrealtime), so a launch lasts spin_us + its start ramp, and (total / N - spin_us) is what the boundary costs by pattern.
//   hipcc --offload-arch=gfx950 -O3 -o launch_gap launch_gap.hip && ./launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode: 0 no stores, 1 plain, 2 nt, 3 sc1, 4 sc0 sc1
__global__ void probe(unsigned* out, int words_per_wg, int spin_ticks, int mode, int use_scratch) {
    extern __shared__ char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (lds && threadIdx.x == 0 && spin_ticks < 0) lds[0] = 1;
    unsigned* p = out + (size_t)blockIdx.x * words_per_wg;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 v = {blockIdx.x, threadIdx.x, 3u, 4u};
    for (int i = threadIdx.x * 4; i + 3 < words_per_wg; i += blockDim.x * 4) {
        unsigned* q = p + i;
        if (mode == 1) *reinterpret_cast<u4*>(q) = v;
        else if (mode == 2) __builtin_nontemporal_store(v, reinterpret_cast<u4*>(q));
        else if (mode == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(q), "v"(v) : "memory");
        else if (mode == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(q), "v"(v) : "memory");
    }
    if (use_scratch) {                                   // a private array with a dynamic index: the kernel gets a scratch segment
        volatile unsigned a[64];
        for (int i = 0; i < 64; ++i) a[i] = i * threadIdx.x;
        if (a[(threadIdx.x + spin_ticks) & 63] == 0xdeadbeef) out[0] = 1;
    }
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin_ticks) __builtin_amdgcn_s_sleep(4);
}

struct K { int grid, block, lds, words, mode, scratch; const char* name; };

int main() {
    int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    CK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    unsigned* buf; CK(hipMalloc(&buf, 256u << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int BIG = 160 * 1024;
    const K S = {ncu, 256, 0, 0, 0, 0, "small(256 thr, no LDS)"};
    const K SM = {ncu * 16, 256, 0, 0, 0, 0, "small x16 WGs"};
    const K L = {ncu, 512, BIG, 0, 0, 0, "big(512 thr, 160 KiB LDS)"};
    const K L64 = {ncu, 512, 64 * 1024, 0, 0, 0, "512 thr, 64 KiB LDS"};
    const K T512 = {ncu, 512, 0, 0, 0, 0, "512 thr, no LDS"};
    const K Lp = {ncu, 512, BIG, 32768, 1, 0, "big + 128 KiB plain stores / WG"};
    const K Lnt = {ncu, 512, BIG, 32768, 2, 0, "big + 128 KiB nt stores / WG"};
    const K Lsc1 = {ncu, 512, BIG, 32768, 3, 0, "big + 128 KiB sc1 stores / WG"};
    const K Lsc01 = {ncu, 512, BIG, 32768, 4, 0, "big + 128 KiB sc0 sc1 stores / WG"};
    const K Lscr = {ncu, 512, BIG, 0, 0, 1, "big + scratch"};
    const K Sscr = {ncu, 256, 0, 0, 0, 1, "small + scratch"};
    struct Pat { K a, b; };
    const std::vector<Pat> pats = {{S, S}, {SM, SM}, {T512, T512}, {L64, L64}, {L, L}, {S, L}, {Lscr, Lscr}, {Sscr, Sscr}, {Lscr, S}, {Lp, Lp}, {Lnt, Lnt}, {Lsc1, Lsc1}, {Lsc01, Lsc01}, {Lp, S}, {Lsc1, S}};
    const int N = 200;
    for (int spin_us : {20, 100}) {
        printf("---- every workgroup lasts %d us from its own start; N = %d dependent launches, pattern A B A B ...\n", spin_us, N);
        for (const Pat& p : pats) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < N; ++i) {
                    const K& k = (i & 1) ? p.b : p.a;
                    hipLaunchKernelGGL(probe, dim3(k.grid), dim3(k.block), k.lds, s, buf, k.words, spin_us * 100, k.mode, k.scratch);
                }
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("  %7.2f us per launch beyond the spin   A = %-36s B = %s\n", best * 1000.f / N - spin_us, p.a.name, p.b.name);
        }
    }
    return 0;
}
