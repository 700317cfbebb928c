// What the GEMM's operand access pattern costs on the L2 -> LDS path (round 5).  tools/experiments/fill_probe.hip measured 41 B/clk/CU for four
// waves issuing CONTIGUOUS 1 KiB LDS-DMA pieces; the GEMM tiles run at ~22 B/clk/CU.  A GEMM piece is 8 rows x 128 B of a row-major
// [rows, K] operand: 8 cache lines whose addresses are one row stride (2 KiB at K = 1024, 8 KiB at K = 4096) apart.  This probe issues
// pieces of that shape — the workgroups of an XCD stream one shared, L2-resident [256 rows x K] panel, 64 columns (128 B) per step like a K loop — against
// the contiguous form, for row strides 128 B (= contiguous K-panel layout), 2 KiB, 2 KiB + 128 B (padded rows), 8 KiB, 8 KiB + 128 B.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/fill_stride_probe.hip -o /tmp/fsp && /tmp/fsp && /tmp/fsp private
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// workgroup = W waves; per step the workgroup fetches ROWS rows x 128 B (ROWS / 8 pieces, dealt to the waves), then moves 128 B along K
template <int ROWS>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, uint32_t* __restrict__ sink, int iters, int stride, int ksteps, size_t panel, int priv) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // shared: one panel per XCD (workgroups are dealt to the 8 XCDs round-robin): L2-resident, shared like a GEMM's W panel;
    // private: a panel per workgroup (128 MiB+ in all: streams from the Infinity Cache / HBM like first-touch operands)
    const char* base = src + (size_t)(priv ? blockIdx.x : (blockIdx.x & 7)) * panel;
    constexpr int NP = ROWS / 8;                 // pieces per step
    const int ppw = NP / nw;                      // pieces per wave and step
    const int r8 = lane >> 3, s8 = lane & 7;
    for (int it = 0; it < iters; ++it) {
        const int k = it % ksteps;
        char* slot = lds + (it & 1) * ROWS * 128;
        for (int q = 0; q < ppw; ++q) {
            const int piece = q * nw + wave;
            const char* g = base + (size_t)(piece * 8 + r8) * stride + k * 128 + ((s8 ^ r8) << 4);
            glds16(g, slot + piece * 1024);
        }
        if (ppw == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ppw == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (ppw == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t acc = *(volatile uint32_t*)(lds + (threadIdx.x * 4) % 1024);
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int priv = argc > 1 && argv[1][0] == 'p';
    printf("%s panels\n", priv ? "PRIVATE (one per workgroup: Infinity Cache / HBM)" : "SHARED (one per XCD: L2-resident)");
    int ncu = 256, mhz = 2400;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, 0);
    const double ghz = mhz / 1e6;
    constexpr int ROWS = 256;
    const size_t maxpanel = (size_t)ROWS * (8192 + 128);
    char* d; uint32_t* sink;
    hipMalloc(&d, (size_t)ncu * 2 * maxpanel + 65536);
    hipMemset(d, 1, (size_t)ncu * 2 * maxpanel + 65536);
    hipMalloc(&sink, 64);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int wg = 1; wg <= 2; ++wg)
        for (int waves : {4, 8, 16}) {
            if (waves == 16) continue;
            printf("wg/CU %d, %2d waves, %d rows x 128 B per step:", wg, waves, ROWS);
            for (int stride : {128, 2048, 2048 + 128, 8192, 8192 + 128}) {
                const int ksteps = stride == 128 ? 1 : (stride / 128 > 64 ? 64 : stride / 128 - (stride % 2048 ? 1 : 0));
                const size_t panel = (size_t)ROWS * stride;
                const int iters = 4000;
                hipLaunchKernelGGL(probe<ROWS>, dim3(ncu * wg), dim3(waves * 64), 2 * ROWS * 128, 0, d, sink, 50, stride, ksteps, panel, priv);
                hipEventRecord(a);
                hipLaunchKernelGGL(probe<ROWS>, dim3(ncu * wg), dim3(waves * 64), 2 * ROWS * 128, 0, d, sink, iters, stride, ksteps, panel, priv);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms = 0;
                hipEventElapsedTime(&ms, a, b);
                const double bytes = (double)ncu * wg * iters * ROWS * 128.0;
                printf("  stride %5d: %5.1f B/clk/CU", stride, bytes / (ms * 1e-3) / ncu / (ghz * 1e9));
            }
            printf("   [%.2f GHz nominal]\n", ghz);
        }
    return 0;
}
