// How many independent VALU instructions hide behind an MFMA issued by the SAME wave, as a function of waves per SIMD?
//   for W in {1,2,3,4} waves/SIMD, MFMA shape in {16x16x32, 32x32x16}, F fillers per MFMA in 0..8 (plain v_fma_f32, or every
//   third one a v_exp_f32): cycles per MFMA per SIMD (wall) and the MFMA-pipe utilisation it implies.
// Development probe (not part of the library):  hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o /tmp/ip && /tmp/ip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int SHAPE, int F, int EXPS>   // SHAPE 0: 16x16x32 (8 independent accumulators), 1: 32x32x16 (4 accumulators)
__global__ __launch_bounds__(1024) void probe(float* out, int iters) {
    bf16x8_t a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (threadIdx.x % 13 + e)); b[e] = (__bf16)(0.02f * (threadIdx.x % 7 + e) - 0.05f); }
    f32x4_t c4[8];
    f32x16_t c16[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) c4[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) c16[i][e] = 0.f;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (float)(threadIdx.x % 11 + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if constexpr (SHAPE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c4[m]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c16[m & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < F; ++f) {
                float& r = x[(m + f) & 7];
                if (EXPS && (f % 3) == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(r));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(0.999f), "v"(0.0001f));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c4[i][0] + x[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += c16[i][0];
    if (s == 123456.789f) out[0] = s;
}

template <int SHAPE, int F, int EXPS>
static void run(int ncu, float* sink, double ghz) {
    const int iters = 2000;
    for (int w = 1; w <= 4; ++w) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((probe<SHAPE, F, EXPS>), dim3(ncu), dim3(256 * w), 0, 0, sink, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<SHAPE, F, EXPS>), dim3(ncu), dim3(256 * w), 0, 0, sink, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n_mfma = (double)iters * 8 * w;                  // per SIMD
        const double cyc = ms * 1e-3 * ghz * 1e9 / n_mfma;
        printf("{\"shape\": \"%s\", \"fillers\": %d, \"exp_every_3rd\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"cycles_per_mfma_at_%.1fGHz\": %.1f}\n",
               SHAPE ? "32x32x16" : "16x16x32", F, EXPS, w, ms, ghz, cyc);
    }
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int ncu = prop.multiProcessorCount;
    float* sink;
    hipMalloc(&sink, 64);
    const double ghz = 2.4;   // nominal; operands are tiny constants, so the part is not power-limited here
#define R(S, F, E) run<S, F, E>(ncu, sink, ghz);
    R(0, 0, 0) R(0, 1, 0) R(0, 2, 0) R(0, 3, 0) R(0, 4, 0) R(0, 6, 0) R(0, 3, 1) R(0, 6, 1)
    R(1, 0, 0) R(1, 2, 0) R(1, 4, 0) R(1, 5, 0) R(1, 6, 0) R(1, 8, 0) R(1, 6, 1)
    return 0;
}
