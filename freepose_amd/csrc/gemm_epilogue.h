// Fused GEMM epilogues shared by the GEMM kernels (same accumulator layout: acc[TC][TR], R operand = permuted rows in the
// MFMA A slot, so a lane owns 4*TR consecutive R indices for each of its TC C-operand rows).
#pragma once
#include "gemm_bf16.h"

namespace fp_gemm {

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
// exact-erf GELU with erf from Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the bf16 output ulp):
// erf(z) = 1 - (a1 t + .. + a5 t^5) exp(-z^2), t = 1/(1 + p z), z >= 0; odd extension.  ~14 VALU ops vs ~40 for erff.
__device__ __forceinline__ float gelu_erf_poly(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float erfabs = fmaf(-poly * t, e, 1.0f);      // erf(|x|/sqrt2)
    const float erfv = copysignf(erfabs, x);
    return 0.5f * x * (1.0f + erfv);
}

template <int BM, int BN, int WM, int WN, int EPI, int VAR, int TC, int TR>
__device__ __forceinline__ void epilogue(const FpGemmArgs& p, f32x4_t (&acc)[TC][TR], int m0, int n0, int wm, int wn,
                                         int li, int lg) {
    constexpr bool TRANS = (EPI == FP_EPI_VT);
    constexpr int TM = BM / WM / 16;
    constexpr int TN = BN / WN / 16;
    if constexpr (!TRANS) {
        // lane owns, for each of its TM token rows, 4*TN consecutive output features
        // (a wave's 16*TN columns are handled as NG independent groups of 64 columns, each with the permuted-row map
        //  a*16 + 4f + b, so a lane owns 16 consecutive features per group)
        constexpr int RUN = 16;
        static_assert(TN % 4 == 0, "wave tile width must be a multiple of 64 columns");
        constexpr int NG = TN / 4;
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
        const int nb = n0 + wn * (16 * TN) + grp * 64 + lg * RUN;
        if (nb < p.N) {
            float bias[RUN], gam[RUN];
            {
                const uint4* bp = (const uint4*)(p.bias + nb);
                uint4 b0 = bp[0], b1 = bp[1];
                const uint32_t w[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) { bias[2 * e] = lo_bf(w[e]); bias[2 * e + 1] = hi_bf(w[e]); }
            }
            if constexpr (EPI == FP_EPI_BIAS_LS_RES) {
                const uint4* gp = (const uint4*)(p.gamma + nb);
                uint4 g0 = gp[0], g1 = gp[1];
                const uint32_t w[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) { gam[2 * e] = lo_bf(w[e]); gam[2 * e + 1] = hi_bf(w[e]); }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm * (16 * TM) + 16 * i + li;
                if (m >= p.M) continue;
                float v[RUN];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][grp * 4 + j][r] + bias[4 * j + r];
                size_t orow = (size_t)m;
                if constexpr (EPI == FP_EPI_BIAS_GELU) {
#pragma unroll
                    for (int e = 0; e < RUN; ++e) v[e] = (VAR & 4) ? gelu_erf_poly(rbf(v[e])) : gelu_erf(rbf(v[e]));
                } else if constexpr (EPI == FP_EPI_BIAS_LS_RES) {
                    const uint4* rp = (const uint4*)(p.resid + (size_t)m * p.ldr + nb);
                    uint4 r0 = rp[0], r1 = rp[1];
                    const uint32_t w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        // reference rounding points: linear out -> bf16, *gamma -> bf16, +resid -> bf16
                        v[2 * e] = lo_bf(w[e]) + rbf(gam[2 * e] * rbf(v[2 * e]));
                        v[2 * e + 1] = hi_bf(w[e]) + rbf(gam[2 * e + 1] * rbf(v[2 * e + 1]));
                    }
                } else if constexpr (EPI == FP_EPI_PATCH) {
                    const int b = m / p.P, pp = m - b * p.P;
                    orow = (size_t)b * p.npad + p.tok_off + pp;
                    const uint4* pp4 = (const uint4*)(p.pos + (size_t)pp * p.N + nb);
                    uint4 q0 = pp4[0], q1 = pp4[1];
                    const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[2 * e] = rbf(v[2 * e]) + lo_bf(w[e]);
                        v[2 * e + 1] = rbf(v[2 * e + 1]) + hi_bf(w[e]);
                    }
                }
                uint4 o0, o1;
                o0.x = pack_bf2(v[0], v[1]);   o0.y = pack_bf2(v[2], v[3]);
                o0.z = pack_bf2(v[4], v[5]);   o0.w = pack_bf2(v[6], v[7]);
                o1.x = pack_bf2(v[8], v[9]);   o1.y = pack_bf2(v[10], v[11]);
                o1.z = pack_bf2(v[12], v[13]); o1.w = pack_bf2(v[14], v[15]);
                uint4* op = (uint4*)(p.C + orow * p.ldc + nb);
                op[0] = o0;
                op[1] = o1;
            }
        }
        }
    } else {
        // transposed V store: lane owns, for each of its TN features, 4*TM consecutive tokens
        constexpr int RUN = 4 * TM;
        static_assert(RUN % 16 == 0, "token runs are stored in 16-token groups");
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + wn * (16 * TN) + 16 * i + li;
            if (n >= p.N) continue;
            const float bias = p.bias ? bf2f(p.bias[n]) : 0.f;
            const int h = n >> 6, d = n & 63;
#pragma unroll
            for (int half = 0; half < RUN / 16; ++half) {
                const int m16 = m0 + wm * (16 * TM) + lg * RUN + half * 16;
                if (m16 >= p.M) continue;
                const int b = m16 / p.npad, t = m16 - b * p.npad;
                float v[16];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][half * 4 + j][r] + bias;
                uint4 o0, o1;
                o0.x = pack_bf2(v[0], v[1]);   o0.y = pack_bf2(v[2], v[3]);
                o0.z = pack_bf2(v[4], v[5]);   o0.w = pack_bf2(v[6], v[7]);
                o1.x = pack_bf2(v[8], v[9]);   o1.y = pack_bf2(v[10], v[11]);
                o1.z = pack_bf2(v[12], v[13]); o1.w = pack_bf2(v[14], v[15]);
                uint4* op = (uint4*)(p.C + (((size_t)b * p.heads + h) * 64 + d) * p.npad + t);
                op[0] = o0;
                op[1] = o1;
            }
        }
    }
}

}  // namespace fp_gemm
