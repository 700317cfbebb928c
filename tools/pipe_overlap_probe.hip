// Do the matrix pipe and the vector ALU of ONE SIMD overlap when the MFMAs and the VALU work come from DIFFERENT waves?
// 512-thread workgroups (2 waves per SIMD, 1 workgroup per CU): waves 0-3 run an MFMA-only loop, waves 4-7 a VALU loop shaped like
// the attention softmax (per 36 MFMAs: 32 v_exp_f32 + ~70 plain VALU).  Modes: both / MFMA waves only / VALU waves only, and a
// same-wave interleave for comparison.  Development probe:
//   hipcc --offload-arch=gfx950 -O3 tools/pipe_overlap_probe.hip -o /tmp/pop && /tmp/pop
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int MODE>   // 0: MFMA waves + VALU waves, 1: MFMA waves only, 2: VALU waves only, 3: every wave does both (same-wave interleave, half the iterations each)
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    bf16x8_t a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (threadIdx.x % 13 + e)); b[e] = (__bf16)(0.02f * (threadIdx.x % 7 + e) - 0.05f); }
    f32x4_t acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float x[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = -0.01f * (float)(threadIdx.x % 11 + i);
    const bool do_mfma = MODE == 3 || (MODE != 2 && wave < 4);
    const bool do_valu = MODE == 3 || (MODE != 1 && wave >= 4);
    const int n = MODE == 3 ? iters / 2 : iters;
    for (int it = 0; it < n; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 9; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        if (do_valu) {
            float mx = x[0];
#pragma unroll
            for (int i = 0; i < 32; ++i) mx = fmaxf(mx, x[i]);
#pragma unroll
            for (int i = 0; i < 32; ++i) x[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[i], 0.18f, -mx * 0.18f)) - 1.0f;
#pragma unroll
            for (int i = 0; i < 32; i += 2) { const float t = x[i] + x[i + 1]; x[i] = t * 0.5f - 0.3f; }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) s += acc[i][0];
#pragma unroll
    for (int i = 0; i < 32; ++i) s += x[i];
    if (s == 123456.789f) out[0] = s;
}

template <class F>
static float time_ms(F&& f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) f();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int ncu = prop.multiProcessorCount, iters = 4000;
    float* sink;
    hipMalloc(&sink, 64);
    const float both = time_ms([&] { hipLaunchKernelGGL(probe<0>, dim3(ncu), dim3(512), 0, 0, sink, iters); });
    const float mfma = time_ms([&] { hipLaunchKernelGGL(probe<1>, dim3(ncu), dim3(512), 0, 0, sink, iters); });
    const float valu = time_ms([&] { hipLaunchKernelGGL(probe<2>, dim3(ncu), dim3(512), 0, 0, sink, iters); });
    const float same = time_ms([&] { hipLaunchKernelGGL(probe<3>, dim3(ncu), dim3(512), 0, 0, sink, iters); });
    printf("{\"iters\": %d, \"mfma_waves_only_ms\": %.3f, \"valu_waves_only_ms\": %.3f, \"both_on_partner_waves_ms\": %.3f, \"sum_ms\": %.3f, "
           "\"every_wave_does_both_half_iters_ms\": %.3f, \"note\": \"36 MFMAs 16x16x32 vs 32 v_exp + ~100 VALU per iteration; 2 waves per SIMD\"}\n",
           iters, mfma, valu, both, mfma + valu, same);
    return 0;
}
