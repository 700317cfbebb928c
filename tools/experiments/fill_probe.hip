// What a CU can pull from L2 per clock, by path (round 4, after the K-tile ring left the 64x64 GEMM tier at ~22 B/clk/CU):
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction) into a ring of LDS slots
//   mode 1: global_load_dwordx4 into registers (16 B per lane), consumed by an integer fold
//   mode 2: both, alternating (half of the bytes each)
// Every workgroup (W waves) streams its own L2-resident window (<= 2 MiB per XCD: 32 CUs x 64 KiB) over and over.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/fill_probe.hip -o fill_probe && ./fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, uint32_t* __restrict__ sink, int iters, int window) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* base = src + (size_t)blockIdx.x * window;
    uint32_t acc = 0;
    // per wave: 8 pieces of 1 KiB per iteration
    for (int it = 0; it < iters; ++it) {
        const int off = ((it * nw + wave) * 8192) % window;
        uint4 r[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const char* g = base + ((off + p * 1024) % window) + lane * 16;
            if (MODE == 0 || (MODE == 2 && (p & 1) == 0)) glds16(g, lds + (wave * 8 + p) * 1024);
            else r[p] = *(const uint4*)g;
        }
        if (MODE != 0) {
#pragma unroll
            for (int p = 0; p < 8; ++p)
                if (MODE == 1 || (p & 1)) acc += r[p].x ^ r[p].y ^ r[p].z ^ r[p].w;
        }
        if (MODE != 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE != 1) acc += *(volatile uint32_t*)(lds + (threadIdx.x * 4) % 1024);
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
double run(const char* d, uint32_t* sink, int ncu, int waves, int wg_per_cu, int iters, int window) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int lds = MODE == 1 ? 0 : waves * 8 * 1024;
    hipLaunchKernelGGL(probe<MODE>, dim3(ncu * wg_per_cu), dim3(waves * 64), lds, 0, d, sink, 10, window);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<MODE>, dim3(ncu * wg_per_cu), dim3(waves * 64), lds, 0, d, sink, iters, window);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)ncu * wg_per_cu * waves * iters * 8192.0;
    return bytes / (ms * 1e-3);
}

int main() {
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    int mhz = 2400;
    hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, 0);
    const int window = 64 * 1024;
    char* d; uint32_t* sink;
    hipMalloc(&d, (size_t)ncu * 4 * window + 65536);
    hipMemset(d, 1, (size_t)ncu * 4 * window + 65536);
    hipMalloc(&sink, 64);
    const double ghz = mhz / 1e6;
    for (int wg = 1; wg <= 2; ++wg)
        for (int waves : {2, 4, 8}) {
            const double a = run<0>(d, sink, ncu, waves, wg, 4000, window), b = run<1>(d, sink, ncu, waves, wg, 4000, window), c = run<2>(d, sink, ncu, waves, wg, 4000, window);
            printf("wg/CU %d waves/wg %d: LDS-DMA %.2f TB/s (%.1f B/clk/CU) | to registers %.2f TB/s (%.1f) | half / half %.2f TB/s (%.1f)   [nominal %.2f GHz]\n",
                   wg, waves, a / 1e12, a / ncu / (ghz * 1e9), b / 1e12, b / ncu / (ghz * 1e9), c / 1e12, c / ncu / (ghz * 1e9), ghz);
        }
    return 0;
}
