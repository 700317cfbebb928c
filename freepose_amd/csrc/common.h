// freepose_amd — shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// Wave = 64 lanes everywhere; no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bf16 bits; all conversions below are explicit RNE

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define FP_WAVE 64

// ---- bf16 <-> f32 (round-to-nearest-even; NaN kept quiet) -----------------------------------
__host__ __device__ __forceinline__ float bf2f(bf16_t h) {
    union { uint32_t u; float f; } v;
    v.u = ((uint32_t)h) << 16;
    return v.f;
}
__host__ __device__ __forceinline__ bf16_t f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    // gfx950 has a hardware RNE f32->bf16 convert (v_cvt_pk_bf16_f32); identical to the software form below for every
    // finite input, branch-free (the software NaN test costs a divergent branch per element in epilogues).
    return __builtin_bit_cast(unsigned short, static_cast<__bf16>(f));
#else
    union { uint32_t u; float f; } v;
    v.f = f;
    uint32_t u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
#endif
}
// round an f32 to the nearest bf16 value but keep it in f32 (models a bf16 module boundary)
__host__ __device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
    bf16x2_hw v;
    v[0] = static_cast<__bf16>(lo);
    v[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(uint32_t, v);  // one v_cvt_pk_bf16_f32
}
__device__ __forceinline__ float lo_bf(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// ---- wave-level helpers ----------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// async global -> LDS copy, 16 B per lane, LDS image is lane-linear from the wave-uniform base
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ---- host-side error plumbing (C-ABI: int status + fp_last_error()) --------------------------
#ifdef __cplusplus
extern "C" const char* fp_last_error(void);
#endif
void fp_set_error(const char* fmt, ...);

#define FP_OK 0
#define FP_ERR_INVALID 1
#define FP_ERR_HIP 2
#define FP_ERR_STATE 3

#define FP_HIP(call)                                                                       \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            fp_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                         __LINE__);                                                        \
            return FP_ERR_HIP;                                                             \
        }                                                                                  \
    } while (0)

#define FP_REQUIRE(cond, ...)       \
    do {                            \
        if (!(cond)) {              \
            fp_set_error(__VA_ARGS__); \
            return FP_ERR_INVALID;  \
        }                           \
    } while (0)

#define FP_LAUNCH_CHECK()                                                          \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            fp_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), \
                         __FILE__, __LINE__);                                      \
            return FP_ERR_HIP;                                                     \
        }                                                                          \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// experiment toggles (fp_set_option in the C ABI); a value < 0 means "use the built-in default"
enum { FP_OPT_GEMM_VARIANT = 0, FP_OPT_ATTN_SLOTS = 1, FP_OPT_COUNT = 8 };
int fp_opt_get(int key, int dflt);
