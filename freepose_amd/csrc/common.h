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
// IEEE square root.  `__fsqrt_rn` is NOT one on this target: the HIP headers map it to __ocml_native_sqrt_f32 = a bare v_sqrt_f32
// (1 ulp), and a norm that lands within that ulp of a bf16 rounding boundary then rounds the other way — every element of the row moves
// (about one row in 16 000; found by tests/test_gpu_fuzz.py at 53 357 x 768).  `sqrtf` compiles to v_sqrt_f32 + the correction
// steps (correctly rounded: -fhip-fp32-correctly-rounded-divide-sqrt, on by default), which is what the C oracle's sqrtf is.
__device__ __forceinline__ float fp_sqrt_rn(float x) { return sqrtf(x); }

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
    bf16x2_hw v;
    v[0] = static_cast<__bf16>(lo);
    v[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(uint32_t, v);  // one v_cvt_pk_bf16_f32
}
__device__ __forceinline__ float lo_bf(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// packed 16-bit helpers (one VALU instruction each; hipcc scalarises the generic vector forms of sub_sat / min)
__device__ __forceinline__ uint32_t pk_sub_sat_u16(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// ---- wave-level helpers ----------------------------------------------------------------------
// Value held by lane (l ^ mask), mask = 32, 16, 8, 4, 2, 1, fetched on the VALU only (gfx950): v_permlane32/16_swap for
// the wave halves / 16-lane rows, DPP inside a row (row_ror:8 is l^8 within 16 lanes; row_half_mirror then
// quad_perm[3,2,1,0] is l^7^3 = l^4).  __shfl_xor compiles to ds_bpermute_b32 — an LDS-crossbar round trip with a wait
// per butterfly step, which made the row-dot kernels (bank scan, template scoring, LayerNorm) latency-bound.
template <int MASK>
__device__ __forceinline__ float lane_xor(float v) {
    const int i = __float_as_int(v);
    if constexpr (MASK == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(i, i, false, false);
        return __int_as_float((threadIdx.x & 32) ? r[0] : r[1]);
    } else if constexpr (MASK == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(i, i, false, false);
        return __int_as_float((threadIdx.x & 16) ? r[0] : r[1]);
    } else if constexpr (MASK == 8) {
        return __int_as_float(__builtin_amdgcn_mov_dpp(i, 0x128, 0xf, 0xf, true));       // row_ror:8
    } else if constexpr (MASK == 4) {
        const int m = __builtin_amdgcn_mov_dpp(i, 0x141, 0xf, 0xf, true);                  // row_half_mirror: l^7
        return __int_as_float(__builtin_amdgcn_mov_dpp(m, 0x1b, 0xf, 0xf, true));         // quad_perm [3,2,1,0]: ^3
    } else if constexpr (MASK == 2) {
        return __int_as_float(__builtin_amdgcn_mov_dpp(i, 0x4e, 0xf, 0xf, true));         // quad_perm [2,3,0,1]
    } else {
        return __int_as_float(__builtin_amdgcn_mov_dpp(i, 0xb1, 0xf, 0xf, true));         // quad_perm [1,0,3,2]
    }
}
// xor-butterfly reductions, masks 32 -> 1 (the pairing order is part of the fp32 "dot64" contract with the oracle)
__device__ __forceinline__ float wave_sum(float v) {
    v += lane_xor<32>(v);
    v += lane_xor<16>(v);
    v += lane_xor<8>(v);
    v += lane_xor<4>(v);
    v += lane_xor<2>(v);
    v += lane_xor<1>(v);
    return v;
}
// Four xor-butterfly sums at once: returns, on every lane of 16-lane row u = lane >> 4, wave_sum(v_u) — bit-identical to
// four separate wave_sum calls (only commutativity of + is used): one v_permlane32_swap gives each wave half its own and
// its partner's value for two of the rows, one v_permlane16_swap repeats that inside the halves, the last four steps run
// on the single remaining value.  15 VALU instructions instead of 56.
__device__ __forceinline__ float wave_sum4(float v0, float v1, float v2, float v3) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_int(v0), __float_as_int(v2), false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_int(v1), __float_as_int(v3), false, false);
    const float t02 = __int_as_float(a[0]) + __int_as_float(a[1]);   // lanes 0-31: row 0, lanes 32-63: row 2
    const float t13 = __int_as_float(b[0]) + __int_as_float(b[1]);   // lanes 0-31: row 1, lanes 32-63: row 3
    const auto c = __builtin_amdgcn_permlane16_swap(__float_as_int(t02), __float_as_int(t13), false, false);
    float w = __int_as_float(c[0]) + __int_as_float(c[1]);           // 16-lane row u holds row u's partial
    w += lane_xor<8>(w);
    w += lane_xor<4>(w);
    w += lane_xor<2>(w);
    w += lane_xor<1>(w);
    return w;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, lane_xor<32>(v));
    v = fmaxf(v, lane_xor<16>(v));
    v = fmaxf(v, lane_xor<8>(v));
    v = fmaxf(v, lane_xor<4>(v));
    v = fmaxf(v, lane_xor<2>(v));
    v = fmaxf(v, lane_xor<1>(v));
    return v;
}

// async global -> LDS copy, 16 B per lane, LDS image is lane-linear from the wave-uniform base
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// the same copy through a buffer descriptor: wave-uniform base, per-lane 32-bit byte offset (VGPR), wave-uniform byte offset
// (SGPR): a loop that only advances the uniform offset spends no vector ALU work on addresses.  The descriptor type exists only
// in the device pass, hence the guard (the host pass needs just the declaration to emit the kernel stubs).
__device__ __forceinline__ void glds16_buf(const void* base, unsigned lane_off, unsigned wave_off, void* lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, lane_off, wave_off, 0, 0);
#else
    (void)base; (void)lane_off; (void)wave_off; (void)lds_wave_base;
#endif
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

// Raise a kernel's dynamic-LDS limit once per DEVICE (function attributes are per device; a host may hold one context per GPU in
// one process).  `seen` is a per-call-site bit mask of the devices already done.
#define FP_DYN_LDS_ONCE(fn, bytes)                                                                              \
    do {                                                                                                        \
        static unsigned long long seen__ = 0ull;                                                                \
        int dev__ = 0;                                                                                          \
        FP_HIP(hipGetDevice(&dev__));                                                                           \
        if (dev__ >= 64 || !((seen__ >> dev__) & 1ull)) {                                                       \
            FP_HIP(hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes))); \
            if (dev__ < 64) seen__ |= 1ull << dev__;                                                            \
        }                                                                                                       \
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

#ifdef FP_LAB
// LAB BUILD ONLY (libfreepose_hip_lab.so, loaded by tools/): process-global experiment toggles set through fp_lab_set_option or the
// FP_* environment variables; a value < 0 means "use the built-in default".  The product library has neither: its three run-time
// options ("ln_fused", "raster_tiled", "gemm_row_split") live in the fp_ctx (fp_ctx_set_option).
enum { FP_OPT_GEMM_VARIANT = 0, FP_OPT_ATTN_SLOTS = 1, FP_OPT_GEMM_DBG = 2, FP_OPT_ATTN_VARIANT = 3, FP_OPT_TOPK_SELECT = 4, FP_OPT_GEMM_RING = 5, FP_OPT_GEMM_SK = 6, FP_OPT_GEMM_SK_GRID = 7, FP_OPT_GEMM_STREAM_MB = 8, FP_OPT_RASTER_DBG = 9, FP_OPT_COUNT = 12 };
int fp_opt_get(int key, int dflt);
#endif
