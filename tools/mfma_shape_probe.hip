// Which MFMA shape sustains more flops on THIS box under its power limit?  Register-only loops, no memory traffic:
//   v_mfma_f32_16x16x32_bf16 (4 passes, 4 acc regs) vs v_mfma_f32_32x32x16_bf16 (8 passes, 16 acc regs),
//   at 1 / 2 / 4 waves per SIMD, on zero and on pseudo-random operands (power, and so the clock, depends on the data).
// Development probe, not part of the library:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o /tmp/mfma_shape_probe && /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int SHAPE, int NFRAG>   // SHAPE 0: 16x16x32 with 8 accumulators; 1: 32x32x16 with 4 accumulators.  NFRAG distinct operand pairs
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters, int random) {
    bf16x8_t a[NFRAG], b[NFRAG];
#pragma unroll
    for (int f = 0; f < NFRAG; ++f)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t h = hash32((blockIdx.x * 256 + threadIdx.x) * 64 + f * 16 + e);
            const float va = random ? ((float)(h & 0xffff) / 32768.f - 1.f) : 0.f;
            const float vb = random ? ((float)(h >> 16) / 32768.f - 1.f) : 0.f;
            a[f][e] = (__bf16)va;
            b[f][e] = (__bf16)(vb * 0.05f);
        }
    float s = 0.f;
    if constexpr (SHAPE == 0) {
        f32x4_t acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i % NFRAG]), "v"(b[(i / 2) % NFRAG]));   // asm: hipcc's
                // own allocation of this loop puts D and C in different (overlapping) AGPR ranges and fills the loop with v_accvgpr moves
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x16_t acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i % NFRAG]), "v"(b[(i / 2) % NFRAG]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][e];
    }
    if (s == 123456.789f) out[0] = s;
}

template <class F>
static float time_ms(F&& f, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0);
        f();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        sum += ms;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    (void)best;
    return sum / reps;   // mean: under a power limit the sustained rate is what matters
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int ncu = prop.multiProcessorCount;
    float* sink;
    hipMalloc(&sink, 64);
    printf("{\"device\": \"%s\", \"cus\": %d, \"rows\": [\n", prop.name, ncu);
    const int iters = 40000;
    bool first = true;
    for (int random = 0; random < 2; ++random)
        for (int wps = 1; wps <= 4; wps *= 2) {      // waves per SIMD = workgroups (4 waves) per CU
            const int wgs = ncu * wps;
            for (int shape = 0; shape < 2; ++shape) {
                const double per_wave = shape == 0 ? iters * 8.0 * (2.0 * 16 * 16 * 32) : iters * 4.0 * (2.0 * 32 * 32 * 16);
                float ms;
                if (shape == 0) ms = time_ms([&] { hipLaunchKernelGGL((mfma_kernel<0, 4>), dim3(wgs), dim3(256), 0, 0, sink, iters, random); }, 6);
                else ms = time_ms([&] { hipLaunchKernelGGL((mfma_kernel<1, 4>), dim3(wgs), dim3(256), 0, 0, sink, iters, random); }, 6);
                const double tf = (double)wgs * 4 * per_wave / (ms * 1e-3) / 1e12;
                printf("%s  {\"shape\": \"%s\", \"waves_per_simd\": %d, \"data\": \"%s\", \"ms\": %.3f, \"tflops\": %.1f}", first ? "" : ",\n",
                       shape == 0 ? "16x16x32" : "32x32x16", wps, random ? "random" : "zero", ms, tf);
                first = false;
                fflush(stdout);
            }
        }
    printf("\n]}\n");
    return 0;
}
