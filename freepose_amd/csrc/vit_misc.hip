// HBM-bound helper kernels of the ViT path (K0, K2, K8, K9 of SURVEY.md §2.3), gfx950 only.
//   im2col_norm   : ImageNet normalise (dino.py:12,16) + 14x14 patch unfold -> GEMM A operand
//   token_init    : cls+pos / register tokens / zero pad rows of the token buffer
//   layernorm     : LayerNorm(eps) over D, one wave per row, optional row gather (final norm + slice,
//                   dino.py:23-30)
//   posembed_aa   : bicubic antialias resize of the 37x37 pos-embed grid (hub DINOv2
//                   interpolate_pos_encoding; public algorithm of torch upsample_bicubic2d_aa)
//   ffa           : mask any-pool 14x14 -> masked mean over patches -> optional L2 normalise
//                   (scripts/extract_retrieval_features.py:51-57, extract_proposals_ground.py:129-134)
#include "internal.h"
#include "gemm_bf16.h"

namespace {

// ---------------------------------------------------------------------------------------------
// im2col + normalise.  images [B,3,H,W] bf16 in [0,1]  ->  A [B*P, KP] bf16, k = c*ps*ps + dy*ps + dx
// Rounding points follow torchvision Normalize on a bf16 tensor: sub (round) then div (round).
__global__ void im2col_norm_kernel(const bf16_t* __restrict__ img, bf16_t* __restrict__ A, int B, int H,
                                   int W, int ps, int KP, float m0, float m1, float m2, float s0, float s1,
                                   float s2) {
    const int gw = W / ps, gh = H / ps, P = gw * gh;
    const int K = 3 * ps * ps;
    const int chunks = KP / 8;
    const long total = (long)B * P * chunks;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % chunks);
        const long row = idx / chunks;
        const int b = (int)(row / P), pp = (int)(row % P);
        const int py = pp / gw, px = pp % gw;
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            float v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = ch * 8 + e + u;
                float val = 0.f;
                if (k < K) {
                    const int c = k / (ps * ps), rem = k % (ps * ps);
                    const int dy = rem / ps, dx = rem % ps;
                    const float x = bf2f(img[(((size_t)b * 3 + c) * H + (py * ps + dy)) * W + px * ps + dx]);
                    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
                    const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
                    val = rbf(rbf(x - mean) / sd);
                }
                v[u] = val;
            }
            w[e / 2] = pack_bf2(v[0], v[1]);
        }
        *(uint4*)(A + (size_t)row * KP + ch * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// token buffer rows that the patch GEMM does not write: row 0 = cls + pos[0], rows 1..nreg = register
// tokens (no pos-embed), rows [n_tok, npad) = 0
__global__ void token_init_kernel(bf16_t* __restrict__ X, const bf16_t* __restrict__ cls,
                                  const bf16_t* __restrict__ pos0, const bf16_t* __restrict__ reg, int nreg,
                                  int n_tok, int npad, int D) {
    const int b = blockIdx.y;
    const int nrows = 1 + nreg + (npad - n_tok);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nrows * D; i += gridDim.x * blockDim.x) {
        const int r = i / D, c = i % D;
        int row;
        bf16_t v;
        if (r == 0) { row = 0; v = f2bf(bf2f(cls[c]) + bf2f(pos0[c])); }
        else if (r <= nreg) { row = r; v = reg[(r - 1) * D + c]; }
        else { row = n_tok + (r - 1 - nreg); v = 0; }
        X[((size_t)b * npad + row) * D + c] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per output row. in_row = (r / rows_per_b) * in_stride_b + in_off + r % rows_per_b
// L2 = 1: the row is written F.normalize()d (pose_estimator.py:85: the estimator normalises the template features it scores) — the
// bf16-rounded LayerNorm outputs a lane holds ARE the canonical dot64 layout of l2norm_rows_kernel (lane l owns elements (c*64+l)*8+e),
// so norm, rounding points and quotient are that kernel's, bit for bit, without a second pass over the features.
template <int MAXC, int L2 = 0>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ X, bf16_t* __restrict__ Y,
                                                        const bf16_t* __restrict__ gamma,
                                                        const bf16_t* __restrict__ beta, int rows, int D,
                                                        float eps, int rows_per_b, int in_stride_b, int in_off) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    const int nch = D / 8;
    for (int r = wave; r < rows; r += nwave) {
        const size_t ir = (size_t)(r / rows_per_b) * in_stride_b + in_off + (r % rows_per_b);
        const bf16_t* xr = X + ir * D;
        float v[MAXC][8];
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                const uint4 q = *(const uint4*)(xr + ch * 8);
                const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[c][2 * e] = lo_bf(w[e]); v[c][2 * e + 1] = hi_bf(w[e]); }
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += v[c][e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
            }
        }
        const float mean = wave_sum(sum) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (lane + 64 * c < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; sq += d * d; }
            }
        const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
        bf16_t* yr = Y + (size_t)r * D;
        uint32_t keep[MAXC][4];
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                const uint4 g = *(const uint4*)(gamma + ch * 8), bb = *(const uint4*)(beta + ch * 8);
                const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, bw[4] = {bb.x, bb.y, bb.z, bb.w};
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = (v[c][2 * e] - mean) * rstd * lo_bf(gw[e]) + lo_bf(bw[e]);
                    const float b = (v[c][2 * e + 1] - mean) * rstd * hi_bf(gw[e]) + hi_bf(bw[e]);
                    o[e] = pack_bf2(a, b);
                }
                if (L2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        keep[c][e] = o[e];
                        const float lo = lo_bf(o[e]), hi = hi_bf(o[e]);
                        acc = __fmaf_rn(lo, lo, acc);
                        acc = __fmaf_rn(hi, hi, acc);
                    }
                } else {
                    *(uint4*)(yr + ch * 8) = make_uint4(o[0], o[1], o[2], o[3]);
                }
            }
        }
        if (L2) {
            acc = wave_sum(acc);
            const float n = fmaxf(rbf(fp_sqrt_rn(acc)), 1e-12f);
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int ch = lane + 64 * c;
                if (ch < nch) {
                    uint32_t o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pack_bf2(__fdiv_rn(lo_bf(keep[c][e]), n), __fdiv_rn(hi_bf(keep[c][e]), n));
                    *(uint4*)(yr + ch * 8) = make_uint4(o[0], o[1], o[2], o[3]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bicubic (a = -0.5) antialias resize of the patch pos-embed grid, fp32 math, bf16 in/out.
// src [G*G, D] (row-major grid, channel-last), dst [gh*gw, D]
__device__ __forceinline__ float cubic_aa(float x) {
    const float a = -0.5f;
    x = fabsf(x);
    if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
    if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
    return 0.f;
}
__device__ __forceinline__ void aa_window(int i, int in, int out, int& xmin, int& xsize, float& center,
                                          float& invscale) {
    const float scale = (float)in / (float)out;
    const float support = scale >= 1.f ? 2.f * scale : 2.f;
    invscale = scale >= 1.f ? 1.f / scale : 1.f;
    center = scale * ((float)i + 0.5f);
    xmin = max((int)(center - support + 0.5f), 0);
    xsize = min((int)(center + support + 0.5f), in) - xmin;
}
constexpr int AA_MAX_TAPS = 192;
__global__ void posembed_aa_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int G, int gh,
                                   int gw, int D) {
    const int op = blockIdx.x;  // output patch
    const int oy = op / gw, ox = op % gw;
    int ymin, ysize, xmin, xsize;
    float cy, iy, cx, ix;
    aa_window(oy, G, gh, ymin, ysize, cy, iy);
    aa_window(ox, G, gw, xmin, xsize, cx, ix);
    // window weights in LDS: 2 * support + 1 taps, support = 2 * G / g when shrinking (a 37-cell grid to one cell: 150 taps)
    __shared__ float wy[AA_MAX_TAPS], wx[AA_MAX_TAPS];
    if (threadIdx.x == 0) {
        float sy = 0.f, sx = 0.f;
        for (int j = 0; j < ysize; ++j) { wy[j] = cubic_aa(((float)(j + ymin) - cy + 0.5f) * iy); sy += wy[j]; }
        for (int j = 0; j < xsize; ++j) { wx[j] = cubic_aa(((float)(j + xmin) - cx + 0.5f) * ix); sx += wx[j]; }
        for (int j = 0; j < ysize; ++j) wy[j] /= sy;
        for (int j = 0; j < xsize; ++j) wx[j] /= sx;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        // horizontal pass first, then vertical (order of torch's separable CPU kernel)
        float acc = 0.f;
        for (int jy = 0; jy < ysize; ++jy) {
            float row = 0.f;
            for (int jx = 0; jx < xsize; ++jx)
                row += wx[jx] * bf2f(src[((size_t)(ymin + jy) * G + xmin + jx) * D + c]);
            acc += wy[jy] * row;
        }
        dst[(size_t)op * D + c] = f2bf(acc);
    }
}

// ---------------------------------------------------------------------------------------------
// FFA descriptor. feats [B,P,D] bf16, mask u8 [B,Hm,Wm] (Hm = gh*cell, Wm = gw*cell) or patch mask u8 [B,P] when cell == 1;
// out [B,D] bf16 = feat[mask].mean(0) (scripts/extract_retrieval_features.py:51-57).
// Canonical summation order (round 6; the CPU checker restates it step for step): the P patches are cut into FFA_NB = 32
// blocks of BL = ceil(P / 32) consecutive patch indices; within a block the masked rows are added in ascending patch order to an fp32
// accumulator that starts at 0; the 32 block sums are then added in ascending block order, ((s0 + s1) + s2) + ...; divide by the mask
// count, round to bf16.  (Rounds 1-5 used one accumulator over all P patches: a dependent chain of 1 369 additions that one crop —
// the video query — paid 0.25 ms for.  torch's own order for the reference's mean is unspecified; the oracle is pinned to it to
// 1 bf16 ulp either way.)
//   ffa_cellmask_kernel : mask [Hm,Wm] -> patch mask [P] (any pixel of the cell x cell block: cv2.resize(INTER_AREA) > 0), one
//                         workgroup per cell row, the row's `cell` mask lines staged in LDS by coalesced reads
//   ffa_kernel          : one workgroup per (64-column slab, crop), thread (j, c2) sums block j of channel pair c2
constexpr int FFA_NB = 32;

__global__ __launch_bounds__(256) void ffa_cellmask_kernel(const uint8_t* __restrict__ mask, uint8_t* __restrict__ pm, int gh, int gw,
                                                           int cell) {
    extern __shared__ uint8_t rows[];   // [cell][Wm]
    const int b = blockIdx.y, py = blockIdx.x, Wm = gw * cell;
    const uint8_t* src = mask + ((size_t)b * gh * cell + (size_t)py * cell) * Wm;   // the cell row's lines are contiguous
    for (int i = threadIdx.x; i < cell * Wm; i += blockDim.x) rows[i] = src[i];
    __syncthreads();
    for (int px = threadIdx.x; px < gw; px += blockDim.x) {
        int any = 0;
        for (int dy = 0; dy < cell; ++dy)
            for (int dx = 0; dx < cell; ++dx) any |= rows[dy * Wm + px * cell + dx];
        pm[(size_t)b * gh * gw + (size_t)py * gw + px] = any ? 1 : 0;
    }
}

// `arrive` != null (small batches: the video / image queries, with a normalised descriptor asked for): the LAST column-slab workgroup of
// a crop to arrive (one agent-scope counter per crop) normalises the finished row with l2norm_rows_kernel's arithmetic, so a query's
// FFA is two launches (cell mask, this) instead of three — each costs ~5 us of launch gap at one crop.  (Round 6 also measured the mask
// pooling INSIDE this kernel — every slab workgroup reading the crop's 268 KB mask: 34 us against 30 for the three launches.  Dropped.)
// Same sums in the same order either way.
__global__ __launch_bounds__(1024) void ffa_kernel(const bf16_t* __restrict__ feats, const uint8_t* __restrict__ pmask,
                                                   bf16_t* __restrict__ out, float* __restrict__ out_f32, int P, int D,
                                                   bf16_t* __restrict__ out_norm, int* __restrict__ arrive) {
    extern __shared__ uint8_t pm[];  // [P]
    __shared__ int cnt_s, last_s;
    __shared__ float2 part[FFA_NB][32];
    const int b = blockIdx.y;
    if (threadIdx.x == 0) { cnt_s = 0; last_s = 0; }
    __syncthreads();
    int local = 0;
    for (int pidx = threadIdx.x; pidx < P; pidx += blockDim.x) {
        const uint8_t m = pmask[(size_t)b * P + pidx] ? 1 : 0;
        pm[pidx] = m;
        local += m;
    }
    if (local) atomicAdd(&cnt_s, local);
    __syncthreads();
    const int cnt = cnt_s;
    const int c2l = threadIdx.x & 31, j = threadIdx.x >> 5;
    const int c2 = blockIdx.x * 32 + c2l;              // pair of channels
    const bool live = c2 * 2 < D;
    const int BL = (P + FFA_NB - 1) / FFA_NB;
    const int p_lo = j * BL, p_hi = min(P, p_lo + BL);
    float a0 = 0.f, a1 = 0.f;
    if (live) {
        const uint32_t* fp = (const uint32_t*)(feats + (size_t)b * P * D) + c2;
        // the additions keep their order; the loads do not wait for the mask test, so FFA_UNR rows are in flight per thread
        constexpr int FFA_UNR = 16;      // (32 in flight: 0.5 us less for one crop, 23 % more for 256 — measured, not kept)
        for (int p0 = p_lo; p0 < p_hi; p0 += FFA_UNR) {
            uint32_t w[FFA_UNR];
#pragma unroll
            for (int u = 0; u < FFA_UNR; ++u) w[u] = (p0 + u < p_hi) ? fp[(size_t)(p0 + u) * (D / 2)] : 0u;
#pragma unroll
            for (int u = 0; u < FFA_UNR; ++u)
                if (p0 + u < p_hi && pm[p0 + u]) {
                    a0 += lo_bf(w[u]);
                    a1 += hi_bf(w[u]);
                }
        }
    }
    part[j][c2l] = make_float2(a0, a1);
    __syncthreads();
    if (j == 0 && live) {
        float2 t = part[0][c2l];
        for (int k = 1; k < FFA_NB; ++k) { t.x += part[k][c2l].x; t.y += part[k][c2l].y; }
        // mean of a bf16 tensor: fp32 accumulate, divide, round to bf16 (0/0 -> NaN like the reference)
        const float m0 = t.x / (float)cnt, m1 = t.y / (float)cnt;
        if (out) ((uint32_t*)(out + (size_t)b * D))[c2] = pack_bf2(m0, m1);
        if (out_f32) { out_f32[(size_t)b * D + 2 * c2] = rbf(m0); out_f32[(size_t)b * D + 2 * c2 + 1] = rbf(m1); }
    }
    if (!arrive || !out_norm) return;
    // ---- the last slab workgroup of this crop normalises the finished row (hand-off recipe of the CDNA guide, §6 G16: every wave drains
    // its stores, barrier, ONE lane releases at agent scope and arrives on the counter; the last arriver acquires at agent scope, barrier,
    // plain loads).  Nobody waits for anybody: no residency assumption.  The last arriver puts the counter back to zero for the next call.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int prev = __hip_atomic_fetch_add(&arrive[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == (int)gridDim.x - 1) {
            __hip_atomic_store(&arrive[b], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            last_s = 1;
        }
    }
    __syncthreads();
    if (!last_s || threadIdx.x >= 64) return;
    // one wave: l2norm_rows_kernel's arithmetic (lane l takes elements (c*64 + l)*8 + e, fmaf chain, xor-butterfly), 16-byte accesses
    const int lane = threadIdx.x;
    const bf16_t* xr = out + (size_t)b * D;
    float acc = 0.f;
    for (int base = lane * 8; base < D; base += 512) {
        const uint4 q = *(const uint4*)(xr + base);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float lo = lo_bf(w[e]), hi = hi_bf(w[e]); acc = __fmaf_rn(lo, lo, acc); acc = __fmaf_rn(hi, hi, acc); }
    }
    acc = wave_sum(acc);
    const float n = fmaxf(rbf(fp_sqrt_rn(acc)), 1e-12f);
    for (int base = lane * 8; base < D; base += 512) {
        const uint4 q = *(const uint4*)(xr + base);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf2(__fdiv_rn(lo_bf(w[e]), n), __fdiv_rn(hi_bf(w[e]), n));
        *(uint4*)(out_norm + (size_t)b * D + base) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// F.normalize(x, dim=-1) on bf16 rows with the reference's rounding points:
//   n = bf16(sqrt(sum x^2)) ; y = bf16(x / max(n, eps))
// sum order: lane l takes elements (c*64 + l)*8 + e, fmaf chain, then xor-butterfly (canonical, see oracle)
__global__ __launch_bounds__(64) void l2norm_rows_kernel(const bf16_t* __restrict__ X, bf16_t* __restrict__ Y,
                                                         int rows, int D) {
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= rows) return;
    const bf16_t* xr = X + (size_t)r * D;
    float acc = 0.f;
    for (int base = lane * 8; base < D; base += 512) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float v = bf2f(xr[base + e]); acc = __fmaf_rn(v, v, acc); }
    }
    acc = wave_sum(acc);
    const float n = fmaxf(rbf(fp_sqrt_rn(acc)), 1e-12f);
    for (int base = lane * 8; base < D; base += 512) {
#pragma unroll
        for (int e = 0; e < 8; ++e) Y[(size_t)r * D + base + e] = f2bf(__fdiv_rn(bf2f(xr[base + e]), n));
    }
}

// the same arithmetic with the row held in registers (one 16-byte load and one 16-byte store per lane and chunk, four rows per
// workgroup): the pre-normalised template store passes T*P = 540 000 rows through this once per mesh (in place)
template <int NCH>
__global__ __launch_bounds__(256) void l2norm_rows_vec_kernel(const bf16_t* X, bf16_t* Y, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    uint4 a[NCH];
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int base = (c * 64 + lane) * 8;
        a[c] = make_uint4(0, 0, 0, 0);
        if (base < D) a[c] = *(const uint4*)(X + (size_t)r * D + base);
        const uint32_t w[4] = {a[c].x, a[c].y, a[c].z, a[c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
            acc = __fmaf_rn(lo, lo, acc);
            acc = __fmaf_rn(hi, hi, acc);
        }
    }
    acc = wave_sum(acc);
    const float n = fmaxf(rbf(fp_sqrt_rn(acc)), 1e-12f);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int base = (c * 64 + lane) * 8;
        if (base >= D) continue;
        const uint32_t w[4] = {a[c].x, a[c].y, a[c].z, a[c].w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = __fdiv_rn(__uint_as_float(w[e] << 16), n), hi = __fdiv_rn(__uint_as_float(w[e] & 0xffff0000u), n);
            o[e] = (__float_as_uint(rbf(lo)) >> 16) | (__float_as_uint(rbf(hi)) & 0xffff0000u);
        }
        *(uint4*)(Y + (size_t)r * D + base) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm folded into the consuming GEMM (gemm_bf16.h FP_EPI_LN_*): what is left of LN1 / LN2 are per-row statistics.
// Both kernels write, per row, the init-MFMA operand record of gemm_bf16.h — {sh, sl, sh, -mh, -ml, -mh, 0, 0}: sigma = sqrt(var + eps)
// and -mean as two-piece bf16 splits (x ~ xh + xl, xh = bf16(x), xl = bf16(x - xh): 16 mantissa bits) — and rstd = 1 / sigma (fp32).
// row_stats_kernel: statistics straight from the rows (block 0, whose input comes from the patch-embed scatter + token init);
// two-pass variance like layernorm_kernel.  One wave per row.
__device__ __forceinline__ uint4 ln_row_record(float mean, float sigma) { return fp_ln_row_record(mean, sigma); }   // gemm_bf16.h
template <int MAXC>
__global__ __launch_bounds__(256) void row_stats_kernel(const bf16_t* __restrict__ X, uint4* __restrict__ mfrag, float* __restrict__ rstd_out,
                                                        int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    const int nch = D / 8;
    for (int r = wave; r < rows; r += nwave) {
        const bf16_t* xr = X + (size_t)r * D;
        float v[MAXC][8];
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lane + 64 * c;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
            if (ch < nch) {
                const uint4 q = *(const uint4*)(xr + ch * 8);
                const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[c][2 * e] = lo_bf(w[e]); v[c][2 * e + 1] = hi_bf(w[e]); }
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += v[c][e];
            }
        }
        const float mean = wave_sum(sum) / (float)D;
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (lane + 64 * c < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; sq += d * d; }
            }
        const float sigma = __fsqrt_rn(wave_sum(sq) / (float)D + eps);
        if (lane == 0) { mfrag[r] = ln_row_record(mean, sigma); rstd_out[r] = __builtin_amdgcn_rcpf(sigma); }

    }
}

// stats_finalize_kernel: (mean, rstd) from the per-64-column partial sums the producing GEMM's epilogue wrote
// (part[nb][m] = (sum x, sum x^2) of row m over columns 64 nb .. 64 nb + 63; FP_EPI_LS_RES_STATS), added in block order.
__global__ __launch_bounds__(256) void stats_finalize_kernel(const float2* __restrict__ part, uint4* __restrict__ mfrag,
                                                             float* __restrict__ rstd_out, int rows, int nb, float inv_d, float eps, int part_ld) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    uint4 rec;
    float rstd;
    fp_ln_finalize_row(part, (size_t)part_ld, nb, r, inv_d, eps, rec, rstd);
    mfrag[r] = rec;
    rstd_out[r] = rstd;
}

// ln_fold_kernel (once per weight load): W' = bf16(W diag(gamma)),  cs[n] = sum_k W'[n,k],  b'[n] = bias[n] + sum_k W[n,k] beta[k];
// written as the per-feature init-MFMA record {b'h, b'h, b'l, ch, ch, cl, 0, 0} (two-piece bf16 splits).
// One wave per output feature n; the column sum runs over the ROUNDED W' — it must cancel what the MFMAs accumulate.
// Output features n < n_scaled additionally carry the factor row_scale (W' = bf16(W gamma row_scale), b' = (b + W beta) row_scale; one
// rounding, like the fold itself): the ViT passes the q rows of the qkv layer with log2(e) / sqrt(head_dim), so the attention kernel's
// S^T accumulators are base-2 exponents and its softmax spends no multiply-add per element (attention.hip, q_prescaled).
__global__ __launch_bounds__(256) void ln_fold_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ gamma,
                                                      const bf16_t* __restrict__ beta, const bf16_t* __restrict__ bias,
                                                      bf16_t* __restrict__ Wf, uint4* __restrict__ cfrag, int N, int K, int n_scaled,
                                                      float row_scale) {
    const int lane = threadIdx.x & 63;
    const int n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (n >= N) return;
    const float rs = n < n_scaled ? row_scale : 1.0f;   // x 1.0f is exact: unscaled rows keep their bits
    float cs = 0.f, wb = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        const uint4 w = *(const uint4*)(W + (size_t)n * K + k), g = *(const uint4*)(gamma + k), b = *(const uint4*)(beta + k);
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w}, gw[4] = {g.x, g.y, g.z, g.w}, bw[4] = {b.x, b.y, b.z, b.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float f0 = rbf(lo_bf(ww[e]) * lo_bf(gw[e]) * rs), f1 = rbf(hi_bf(ww[e]) * hi_bf(gw[e]) * rs);
            o[e] = pack_bf2(f0, f1);
            cs += f0;
            cs += f1;
            wb = __fmaf_rn(lo_bf(ww[e]), lo_bf(bw[e]), wb);
            wb = __fmaf_rn(hi_bf(ww[e]), hi_bf(bw[e]), wb);
        }
        *(uint4*)(Wf + (size_t)n * K + k) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    cs = wave_sum(cs);
    wb = wave_sum(wb);
    if (lane == 0) {
        const float bp = ((bias ? bf2f(bias[n]) : 0.f) + wb) * rs;
        const float bh = rbf(bp), bl = rbf(bp - bh), ch = rbf(cs), cl = rbf(cs - ch);
        cfrag[n] = make_uint4(pack_bf2(bh, bh), pack_bf2(bl, ch), pack_bf2(ch, cl), 0u);
    }
}

}  // namespace

int fp_row_stats(const bf16_t* X, uint4* ms, float* rstd, int rows, int D, float eps, hipStream_t s) {
    FP_REQUIRE(D % 8 == 0 && D <= 8 * 64 * 3, "row_stats: D=%d unsupported", D);
    const int blocks = std::min(cdiv(rows, 4), 256 * 8);
    if (D <= 512) hipLaunchKernelGGL(row_stats_kernel<1>, dim3(blocks), dim3(256), 0, s, X, ms, rstd, rows, D, eps);
    else if (D <= 1024) hipLaunchKernelGGL(row_stats_kernel<2>, dim3(blocks), dim3(256), 0, s, X, ms, rstd, rows, D, eps);
    else hipLaunchKernelGGL(row_stats_kernel<3>, dim3(blocks), dim3(256), 0, s, X, ms, rstd, rows, D, eps);
    FP_LAUNCH_CHECK();
    return FP_OK;
}

int fp_stats_finalize(const float2* part, uint4* ms, float* rstd, int rows, int D, float eps, hipStream_t s, int part_ld) {
    FP_REQUIRE(D % 64 == 0, "stats_finalize: D=%d must be a multiple of 64", D);
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, s, part, ms, rstd, rows, D / 64, 1.0f / (float)D, eps, part_ld > 0 ? part_ld : rows);
    FP_LAUNCH_CHECK();
    return FP_OK;
}

int fp_ln_fold(const bf16_t* W, const bf16_t* gamma, const bf16_t* beta, const bf16_t* bias, bf16_t* Wf, uint4* cb, int N, int K,
               int n_scaled, float row_scale, hipStream_t s) {
    FP_REQUIRE(W && gamma && beta && bias && Wf && cb, "ln_fold: null argument (W / LayerNorm gamma, beta / bias / outputs)");
    FP_REQUIRE(N > 0 && K % 8 == 0, "ln_fold: N=%d, K=%d (K must be a multiple of 8)", N, K);
    hipLaunchKernelGGL(ln_fold_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, W, gamma, beta, bias, Wf, cb, N, K, n_scaled, row_scale);
    FP_LAUNCH_CHECK();
    return FP_OK;
}

int fp_im2col_norm(const bf16_t* img, bf16_t* A, int B, int H, int W, int ps, int KP, hipStream_t s) {
    FP_REQUIRE(H % ps == 0 && W % ps == 0 && KP % 8 == 0 && KP >= 3 * ps * ps, "im2col: bad shape");
    const long total = (long)B * (H / ps) * (W / ps) * (KP / 8);
    const int blocks = (int)std::min<long>((total + 255) / 256, 256 * 16);
    // torchvision Normalize builds mean/std tensors in the image dtype (bf16)
    hipLaunchKernelGGL(im2col_norm_kernel, dim3(blocks), dim3(256), 0, s, img, A, B, H, W, ps, KP, rbf(0.485f),
                       rbf(0.456f), rbf(0.406f), rbf(0.229f), rbf(0.224f), rbf(0.225f));
    FP_LAUNCH_CHECK();
    return FP_OK;
}

int fp_token_init(bf16_t* X, const bf16_t* cls, const bf16_t* pos0, const bf16_t* reg, int nreg, int B,
                  int n_tok, int npad, int D, hipStream_t s) {
    const int nrows = 1 + nreg + (npad - n_tok);
    hipLaunchKernelGGL(token_init_kernel, dim3(cdiv(nrows * D, 256), B), dim3(256), 0, s, X, cls, pos0, reg, nreg,
                       n_tok, npad, D);
    FP_LAUNCH_CHECK();
    return FP_OK;
}

int fp_layernorm(const bf16_t* X, bf16_t* Y, const bf16_t* gamma, const bf16_t* beta, int rows, int D, float eps,
                 int rows_per_b, int in_stride_b, int in_off, hipStream_t s, int l2_normalize) {
    FP_REQUIRE(D % 8 == 0 && D <= 8 * 64 * 3, "layernorm: D=%d unsupported", D);
    if (rows_per_b <= 0) { rows_per_b = rows; in_stride_b = 0; in_off = 0; }
    const int blocks = std::min(cdiv(rows, 4), 256 * 8);
#define FP_LN(C, L) hipLaunchKernelGGL((layernorm_kernel<C, L>), dim3(blocks), dim3(256), 0, s, X, Y, gamma, beta, rows, D, eps, rows_per_b, in_stride_b, in_off)
    if (D <= 512) { if (l2_normalize) FP_LN(1, 1); else FP_LN(1, 0); }
    else if (D <= 1024) { if (l2_normalize) FP_LN(2, 1); else FP_LN(2, 0); }
    else { if (l2_normalize) FP_LN(3, 1); else FP_LN(3, 0); }
#undef FP_LN
    FP_LAUNCH_CHECK();
    return FP_OK;
}

int fp_posembed_aa(const bf16_t* src, bf16_t* dst, int G, int gh, int gw, int D, hipStream_t s) {
    FP_REQUIRE(G > 0 && gh > 0 && gw > 0, "posembed: bad grid");
    FP_REQUIRE(4.0f * G / gh + 2 < AA_MAX_TAPS && 4.0f * G / gw + 2 < AA_MAX_TAPS, "posembed: downscale factor too large (grid %d -> %d x %d)", G, gh, gw);
    hipLaunchKernelGGL(posembed_aa_kernel, dim3(gh * gw), dim3(256), 0, s, src, dst, G, gh, gw, D);
    FP_LAUNCH_CHECK();
    return FP_OK;
}

int fp_ffa_pool(const bf16_t* feats, const uint8_t* mask, bf16_t* out, float* out_f32, int B, int P, int D, int gh,
                int gw, int cell, uint8_t* pm_scratch, hipStream_t s, bf16_t* out_norm, int* arrive) {
    FP_REQUIRE(gh * gw == P && D % 2 == 0 && cell >= 1, "ffa: bad shape P=%d gh=%d gw=%d", P, gh, gw);
    FP_REQUIRE(P <= 60000 && cell * cell * gw <= 60000, "ffa: P=%d cell=%d too large for the LDS staging", P, cell);
    const uint8_t* pm = mask;
    if (cell > 1) {
        FP_REQUIRE(pm_scratch, "ffa: no patch-mask scratch");
        hipLaunchKernelGGL(ffa_cellmask_kernel, dim3(gh, B), dim3(256), (size_t)cell * cell * gw, s, mask, pm_scratch, gh, gw, cell);
        FP_LAUNCH_CHECK();
        pm = pm_scratch;
    }
    FP_REQUIRE(!arrive || (out && out_norm && D % 8 == 0), "ffa: the fused normalisation needs both bf16 buffers and D %% 8 == 0");
    hipLaunchKernelGGL(ffa_kernel, dim3(cdiv(D, 64), B), dim3(1024), P, s, feats, pm, out, out_f32, P, D, out_norm, arrive);
    FP_LAUNCH_CHECK();
    return FP_OK;
}

int fp_l2norm_rows(const bf16_t* X, bf16_t* Y, int rows, int D, hipStream_t s) {
    FP_REQUIRE(D % 8 == 0, "l2norm: D must be a multiple of 8");
    if (rows == 0) return FP_OK;
    const int nch = cdiv(D, 512);
    const bool aligned = (((uintptr_t)X | (uintptr_t)Y) & 15) == 0;
    if (nch <= 3 && aligned) {   // in-place (X == Y) is fine: a wave reads its whole row before it writes
        if (nch == 1) hipLaunchKernelGGL(l2norm_rows_vec_kernel<1>, dim3(cdiv(rows, 4)), dim3(256), 0, s, X, Y, rows, D);
        else if (nch == 2) hipLaunchKernelGGL(l2norm_rows_vec_kernel<2>, dim3(cdiv(rows, 4)), dim3(256), 0, s, X, Y, rows, D);
        else hipLaunchKernelGGL(l2norm_rows_vec_kernel<3>, dim3(cdiv(rows, 4)), dim3(256), 0, s, X, Y, rows, D);
    } else
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(rows), dim3(64), 0, s, X, Y, rows, D);
    FP_LAUNCH_CHECK();
    return FP_OK;
}
