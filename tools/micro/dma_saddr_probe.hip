// dma_saddr_probe.hip — what does global_load_lds_dwordx4 accept as the per-lane VGPR offset in SGPR-base (saddr) mode?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dma_saddr_probe.hip -o tools/micro/dma_saddr_probe && tools/micro/dma_saddr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const char* base, unsigned off0, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
    const unsigned lane = threadIdx.x;
    const unsigned off = off0 + lane * 16;
    const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds;
    const unsigned long long b64 = (unsigned long long)base;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(off), "s"(b64), "s"(ldsb) : "memory");
    __syncthreads();
    out[lane] = lds[lane * 4];
}
int main() {
    const size_t n = 64ull << 20;                       // 64 MiB of dwords holding their own byte offset
    std::vector<unsigned> h(n / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 4);
    char* d; unsigned* out;
    hipMalloc(&d, n); hipMalloc(&out, 256);
    hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
    const unsigned offs[] = {0u, 1u << 12, (1u << 19) - 1024, 1u << 19, (1u << 20) - 1024, 1u << 20, 1u << 21, 1u << 23, 1u << 25};
    for (unsigned o : offs) {
        probe<<<1, 64>>>(d, o, out);
        hipError_t e = hipDeviceSynchronize();
        unsigned r[64]; hipMemcpy(r, out, 256, hipMemcpyDeviceToHost);
        printf("offset %10u: %s  lane0 read %u (want %u), lane63 read %u (want %u)\n", o, e == hipSuccess ? "ok " : hipGetErrorString(e), r[0], o, r[63], o + 63 * 16);
        if (e != hipSuccess) break;
    }
    return 0;
}
