// How fast can one CU get a 128 KiB output tile out?  256 workgroups x 16 waves (the GEMM's geometry); every wave stores 32 x 1 KiB
// (8 full 128-byte rows per instruction, 16 B per lane) — plain, non-temporal, through a buffer descriptor with various cache bits,
// and 8-byte stores for comparison — `iters` tiles back to back.  Reports bytes / clock / CU at the nominal 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

template <int MODE>
__global__ __launch_bounds__(1024) void probe(uint32_t* out, int iters, int ld_words) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int prow = lane >> 3, pslot = lane & 7;
    u32x4_t v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (int it = 0; it < iters; ++it) {
        // tile (blockIdx, it): 256 rows x 256 bf16 (512 B per row); wave (wm, wn) owns rows wm*64.., columns wn*64 bf16 = 128 B
        const size_t tile_row0 = ((size_t)it * gridDim.x + blockIdx.x) * 256;
        const int wm = wave >> 2, wn = wave & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const size_t row = tile_row0 + wm * 64 + i * 16 + h * 8 + prow;
                uint32_t* p = out + row * ld_words + wn * 32 + pslot * 4;
                v.x += 1;
                if constexpr (MODE == 0) *(u32x4_t*)p = v;
                else if constexpr (MODE == 1) __builtin_nontemporal_store(v, (u32x4_t*)p);
                else if constexpr (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
                else if constexpr (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
                else if constexpr (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
                else if constexpr (MODE == 5) { ((uint2*)p)[0] = make_uint2(v.x, v.y); ((uint2*)p)[1] = make_uint2(v.z, v.w); }
            }
    }
}

template <int MODE>
static void run(const char* name, uint32_t* buf, int ncu, int iters, int ld_words) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(ncu), dim3(1024), 0, 0, buf, iters, ld_words);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<MODE>, dim3(ncu), dim3(1024), 0, 0, buf, iters, ld_words);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)ncu * iters * 131072.0;
    printf("{\"store\": \"%s\", \"iters\": %d, \"ms\": %.4f, \"TB_per_s\": %.2f, \"B_per_clk_per_CU_at_2.4GHz\": %.1f, \"us_per_tile\": %.2f}\n", name, iters, ms,
           bytes / ms / 1e9, bytes / ncu / (ms * 1e-3 * 2.4e9), ms * 1e3 / iters);
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int ncu = prop.multiProcessorCount;
    for (int iters : {1, 8, 64}) {
        const int ld_words = 512;   // 2 KiB rows (N = 1024 bf16): the tile's 512-byte row segments are strided like the GEMM's output
        uint32_t* buf;
        if (hipMalloc((void**)&buf, (size_t)ncu * iters * 256 * ld_words * 4) != hipSuccess) return 2;
        run<0>("plain dwordx4", buf, ncu, iters, ld_words);
        run<1>("nontemporal dwordx4", buf, ncu, iters, ld_words);
        run<2>("sc1 dwordx4", buf, ncu, iters, ld_words);
        run<3>("sc0 sc1 dwordx4", buf, ncu, iters, ld_words);
        run<4>("sc0 dwordx4", buf, ncu, iters, ld_words);
        run<5>("plain 2 x dwordx2", buf, ncu, iters, ld_words);
        (void)hipFree(buf);
    }
    // per-CU limit or memory-side limit?  the same 8-tile stream from fewer workgroups (one per CU, spread over the XCDs by the dispatcher)
    for (int g : {8, 32, 64, 128}) {
        const int iters = 8, ld_words = 512;
        uint32_t* buf;
        if (hipMalloc((void**)&buf, (size_t)g * iters * 256 * ld_words * 4) != hipSuccess) return 2;
        char name[64];
        snprintf(name, sizeof name, "plain dwordx4, %d workgroups", g);
        run<0>(name, buf, g, iters, ld_words);
        snprintf(name, sizeof name, "nontemporal dwordx4, %d workgroups", g);
        run<1>(name, buf, g, iters, ld_words);
        (void)hipFree(buf);
    }
    return 0;
}
