// Pose-side kernels (K14, K16, K17, K18 of SURVEY.md §2.3), gfx950 only.  Integer / byte streaming work.
//   crop_resize_pad : CropResizePad.__call__ (src/utils/bbox_utils.py:20-56) with torch's nearest
//                     index rule  src = min(floorf(dst * float(1/scale)), in-1)
//   geodesic_select : DinoOnlinePoseEstimator.geodesic_distance + np.where(dists < n)
//                     (src/pipeline/estimators/online_pose_estimator.py:25-34,55-56)
//   depth_extents   : mask/bbox of a rendered depth map (renderer.py:112-119, template.py:73-78) and the
//                     x/y extents of depthmap_to_pointcloud (src/pipeline/utils.py:122-145,157-158)
#include "internal.h"

namespace {

struct CropParam {
    int x0, y0, cw, ch;     // crop window (after extension / clipping)
    int h1, w1;             // size after the first nearest resize
    int pad_t, pad_l;       // centre padding (0 when the crop is square)
    int S_h, S_w;           // size before the final resize
    float inv1, inv2;       // float(1/scale) of both interpolate calls
    int out;                // final side (must equal target)
};

// one thread per box: integer/float32/float64 arithmetic exactly as the Python reference evaluates it
__global__ void crop_params_kernel(const int32_t* __restrict__ boxes, int n, int H, int W, float ext, int ext_is_zero,
                                   int target, CropParam* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x0 = boxes[4 * i + 0], y0 = boxes[4 * i + 1], x1 = boxes[4 * i + 2], y1 = boxes[4 * i + 3];
    const int bw = x1 - x0, bh = y1 - y0;
    if (ext_is_zero) {  // int arithmetic: max(0, x0 - 0*bw) ...
        x0 = max(0, x0); x1 = min(W, x1); y0 = max(0, y0); y1 = min(H, y1);
    } else {            // float32 tensor arithmetic, assignment into an int tensor truncates toward zero
        const float ew = ext * (float)bw, eh = ext * (float)bh;
        const float fx0 = (float)x0 - ew, fx1 = (float)x1 + ew, fy0 = (float)y0 - eh, fy1 = (float)y1 + eh;
        x0 = (fx0 > 0.f) ? (int)fx0 : 0;          // Python max(0, t): t only if t > 0
        x1 = (fx1 < (float)W) ? (int)fx1 : W;     // Python min(w, t): t only if t < w
        y0 = (fy0 > 0.f) ? (int)fy0 : 0;
        y1 = (fy1 < (float)H) ? (int)fy1 : H;
    }
    CropParam p;
    p.x0 = x0; p.y0 = y0; p.cw = x1 - x0; p.ch = y1 - y0;
    const int side = max(p.cw, p.ch);
    // `target_max / int_tensor` is Tensor.__rtruediv__ = reciprocal(tensor) * scalar in float32 (two roundings)
    const float scale_f32 = __fmul_rn(__frcp_rn((float)side), (float)target);
    const double scale = (double)scale_f32;               // .item()
    p.h1 = (int)floor((double)p.ch * scale);
    p.w1 = (int)floor((double)p.cw * scale);
    p.inv1 = (float)(1.0 / scale);
    const double ratio = (double)p.w1 / (double)p.h1;
    if (ratio != 1.0) {
        p.pad_t = max((target - p.h1) / 2, 0);   // Python // on non-negative values
        p.pad_l = max((target - p.w1) / 2, 0);
        if (target - p.h1 < 0) p.pad_t = 0;
        if (target - p.w1 < 0) p.pad_l = 0;
        p.S_h = target; p.S_w = target;
    } else {
        p.pad_t = 0; p.pad_l = 0; p.S_h = p.h1; p.S_w = p.w1;
    }
    const double scale2 = (double)target / (double)p.S_h;
    p.out = (int)floor((double)p.S_h * scale2);
    p.inv2 = (float)(1.0 / scale2);
    out[i] = p;
}

// torch's CPU nearest source index for F.interpolate(scale_factor=s): ATen uses its nearest_idx kernel (identity when
// out == in, idx >> 1 when out == 2*in) only when out_h + out_w <= 128 (`small`); the generic kernel used otherwise —
// always, at the pipeline's 420 px — applies  min(floorf(dst * float(1/s)), in - 1)  unconditionally.
__device__ __forceinline__ int nearest_src(int dst, int in, int out, float inv_scale, bool small) {
    if (small && out == in) return dst;
    if (small && out == 2 * in) return dst >> 1;
    return min((int)floorf((float)dst * inv_scale), in - 1);
}

constexpr int CROP_ROWS = 8;

template <int SRC_U8, int OUT_BF16>
__global__ __launch_bounds__(256) void crop_kernel(const void* __restrict__ images, int n_img, int C, int H, int W,
                                                   const CropParam* __restrict__ params, int target,
                                                   const uint8_t* __restrict__ masks, int mask_mode,
                                                   void* __restrict__ outp) {
    // u8 sources: the 256 possible values of u8/255 (as float(double/255.0) or as fp32 division, whichever the caller's
    // reference path uses) are tabulated once per block instead of dividing per pixel and channel
    __shared__ float lut[256];
    if (SRC_U8 != 0) {
        const int t = threadIdx.x;
        lut[t] = (SRC_U8 == 1) ? (float)((double)t / 255.0) : __fdiv_rn((float)t, 255.0f);
        __syncthreads();
    }
    const int i = blockIdx.y;
    const CropParam p = params[i];
    const int img = (n_img == 1) ? 0 : i;
    const bool geom = (p.out == target) && p.cw > 0 && p.ch > 0;
    const bool small1 = (p.h1 + p.w1) <= 128, small2 = (2 * p.out) <= 128;
    // a block owns a strip of CROP_ROWS output rows; a thread walks (row, pixel pair) items of the strip
    const int npair = (target + 1) >> 1;
    const int row0 = blockIdx.x * CROP_ROWS;
    const int nrow = min(CROP_ROWS, target - row0);
    for (int item = threadIdx.x; item < nrow * npair; item += blockDim.x) {
        const int oy = row0 + item / npair;
        const int ox0 = (item % npair) * 2;
        bool valid[2] = {false, false};
        int ys = 0, xs[2] = {0, 0};
        if (geom) {
            const int y2 = nearest_src(oy, p.S_h, p.out, p.inv2, small2);
            const int y1 = y2 - p.pad_t;
            if (y1 >= 0 && y1 < p.h1) {
                ys = p.y0 + nearest_src(y1, p.ch, p.h1, p.inv1, small1);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (ox0 + k >= target) continue;
                    const int x2 = nearest_src(ox0 + k, p.S_w, p.out, p.inv2, small2);
                    const int x1 = x2 - p.pad_l;
                    if (x1 >= 0 && x1 < p.w1) {
                        valid[k] = true;
                        xs[k] = p.x0 + nearest_src(x1, p.cw, p.w1, p.inv1, small1);
                    }
                }
            }
        }
        float m[2] = {1.f, 1.f};
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (valid[k] && masks && mask_mode != 0) m[k] = masks[((size_t)i * H + ys) * W + xs[k]] ? 1.f : 0.f;
        const bool pair = ox0 + 1 < target;
        for (int c = 0; c < C; ++c) {
            float v[2] = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!valid[k]) continue;
                if (mask_mode == 2) v[k] = m[k];
                else {
                    if (SRC_U8 != 0) v[k] = lut[((const uint8_t*)images)[(((size_t)img * H + ys) * W + xs[k]) * C + c]];
                    else v[k] = ((const float*)images)[(((size_t)img * C + c) * H + ys) * W + xs[k]];
                    v[k] *= m[k];
                }
            }
            const size_t o = (((size_t)i * C + c) * target + oy) * target + ox0;
            if (OUT_BF16) {
                bf16_t* op = (bf16_t*)outp + o;
                if (pair && (o & 1) == 0) *(uint32_t*)op = pack_bf2(v[0], v[1]);
                else { op[0] = f2bf(v[0]); if (pair) op[1] = f2bf(v[1]); }
            } else {
                float* op = (float*)outp + o;
                op[0] = v[0];
                if (pair) op[1] = v[1];
            }
        }
    }
}

// The renders' case (u8 HWC source with 3 channels, no mask, even target): the nearest source column of every output column and the
// source row of the strip's rows are tabulated once per block in LDS, a thread then owns FOUR consecutive output pixels of a row in
// all three channels — 12 byte loads in flight, the u8/255 table read in the output type, 4-/8-byte stores.  Same index rule and same
// values as crop_kernel (bit-identical; the generic kernel keeps float sources, masks and odd targets).
template <int SRC_U8, int OUT_BF16>
__global__ __launch_bounds__(256) void crop_rgb_kernel(const uint8_t* __restrict__ images, int n_img, int H, int W,
                                                       const CropParam* __restrict__ params, int target, void* __restrict__ outp) {
    __shared__ float lut[256];
    __shared__ __align__(8) short xs_s[2048 + 4];
    __shared__ int ys_s[CROP_ROWS];
    const int t = threadIdx.x;
    lut[t] = (SRC_U8 == 1) ? (float)((double)t / 255.0) : __fdiv_rn((float)t, 255.0f);
    if (OUT_BF16) lut[t] = __uint_as_float((uint32_t)f2bf(lut[t]));      // the table in the output type (bits in the low half)
    const int i = blockIdx.y;
    const CropParam p = params[i];
    const int img = (n_img == 1) ? 0 : i;
    const bool geom = (p.out == target) && p.cw > 0 && p.ch > 0;
    const bool small1 = (p.h1 + p.w1) <= 128, small2 = (2 * p.out) <= 128;
    const int row0 = blockIdx.x * CROP_ROWS;
    const int nrow = min(CROP_ROWS, target - row0);
    const int ngrp = (target + 3) >> 2;
    for (int ox = t; ox < ngrp * 4; ox += blockDim.x) {
        int xs = -1;
        if (geom && ox < target) {
            const int x1 = nearest_src(ox, p.S_w, p.out, p.inv2, small2) - p.pad_l;
            if (x1 >= 0 && x1 < p.w1) xs = p.x0 + nearest_src(x1, p.cw, p.w1, p.inv1, small1);
        }
        xs_s[ox] = (short)xs;
    }
    if (t < nrow) {
        int ys = -1;
        if (geom) {
            const int y1 = nearest_src(row0 + t, p.S_h, p.out, p.inv2, small2) - p.pad_t;
            if (y1 >= 0 && y1 < p.h1) ys = p.y0 + nearest_src(y1, p.ch, p.h1, p.inv1, small1);
        }
        ys_s[t] = ys;
    }
    __syncthreads();
    const uint8_t* base = images + (size_t)img * H * W * 3;
    const size_t plane = (size_t)target * target;
    for (int item = t; item < nrow * ngrp; item += blockDim.x) {
        const int r = item / ngrp, g = item - r * ngrp;
        const int oy = row0 + r, ox0 = g * 4;
        const int ys = ys_s[r];
        const uint2 xq = *(const uint2*)&xs_s[ox0];
        const int xs[4] = {(int)(short)(xq.x & 0xffff), (int)(short)(xq.x >> 16), (int)(short)(xq.y & 0xffff), (int)(short)(xq.y >> 16)};
        float v[3][4];
        const uint8_t* rowp = base + (size_t)max(ys, 0) * W * 3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = ys >= 0 && xs[k] >= 0;
            const uint8_t* px = rowp + max(xs[k], 0) * 3;
            const int b0 = px[0], b1 = px[1], b2 = px[2];
            v[0][k] = ok ? lut[b0] : 0.f; v[1][k] = ok ? lut[b1] : 0.f; v[2][k] = ok ? lut[b2] : 0.f;
        }
        const int nval = min(4, target - ox0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t o = ((size_t)i * 3 + c) * plane + (size_t)oy * target + ox0;
            if (OUT_BF16) {
                bf16_t* op = (bf16_t*)outp + o;        // (target even, ox0 % 4 == 0: o is even -> 4-byte aligned)
                const uint32_t lo = (__float_as_uint(v[c][0]) & 0xffffu) | (__float_as_uint(v[c][1]) << 16);
                const uint32_t hi = (__float_as_uint(v[c][2]) & 0xffffu) | (__float_as_uint(v[c][3]) << 16);
                if (nval == 4) {
                    if ((o & 3) == 0) *(uint2*)op = make_uint2(lo, hi);
                    else { ((uint32_t*)op)[0] = lo; ((uint32_t*)op)[1] = hi; }
                } else {
                    ((uint32_t*)op)[0] = lo;             // nval == 2 (target even)
                }
            } else {
                float* op = (float*)outp + o;
                if (nval == 4) {
                    if ((o & 3) == 0) *(float4*)op = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
                    else { *(float2*)op = make_float2(v[c][0], v[c][1]); *(float2*)(op + 2) = make_float2(v[c][2], v[c][3]); }
                } else {
                    *(float2*)op = make_float2(v[c][0], v[c][1]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
struct Rot9 { double v[9]; };   // the previous rotation travels as a kernel argument (no staging copy, no host sync)
__global__ void geodesic_flags_kernel(const double* __restrict__ grid, int G, const Rot9 Rprev,
                                      double thresh_deg, int32_t* __restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G) return;
    const double* Rp = Rprev.v;
    double R[9], D[9];
    for (int k = 0; k < 9; ++k) R[k] = grid[(size_t)i * 9 + k];
    // D = R_i * Rp^T
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) D[3 * r + c] = R[3 * r] * Rp[3 * c] + R[3 * r + 1] * Rp[3 * c + 1] + R[3 * r + 2] * Rp[3 * c + 2];
    const double cosv = 0.5 * (D[0] + D[4] + D[8] - 1.0);
    const double a = D[7] - D[5], b = D[2] - D[6], c = D[3] - D[1];
    const double sinv = 0.5 * sqrt(a * a + b * b + c * c);
    const double ang = atan2(sinv, cosv) * 57.29577951308232;
    flags[i] = ang < thresh_deg ? 1 : 0;
}
// single-block ordered compaction of flags -> ascending indices
__global__ __launch_bounds__(1024) void compact_flags_kernel(const int32_t* __restrict__ flags, int G,
                                                             int32_t* __restrict__ out_idx, int* __restrict__ out_n) {
    __shared__ int wsum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int base = 0; base < G; base += 1024) {
        const int i = base + threadIdx.x;
        const int f = (i < G) ? flags[i] : 0;
        int x = f;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        int pre = carry;
        for (int k = 0; k < w; ++k) pre += wsum[k];
        if (f) out_idx[pre + x - 1] = i;
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < 16; ++k) t += wsum[k]; carry += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) *out_n = carry;
}

// ---------------------------------------------------------------------------------------------
// one block per view: count, bbox of (depth > 0) with the <100 px fallback square, fp64 extents.
// X = ((x - cx) / fx) * z: the quotient depends only on the column (row for Y), so it is tabulated once per block in
// LDS and the per-pixel work is two fp64 multiplies — bit-identical to dividing per pixel, ~10x less fp64 work.
constexpr int EXT_THREADS = 1024;
__global__ __launch_bounds__(EXT_THREADS) void depth_extents_kernel(const float* __restrict__ depth, int Hh, int W, double fx,
                                                                    double fy, double cx, double cy, double* __restrict__ out) {
    extern __shared__ double tab[];   // [W] column factors, then [Hh] row factors
    double* ax = tab;
    double* ay = tab + W;
    for (int x = threadIdx.x; x < W; x += blockDim.x) ax[x] = ((double)x - cx) / fx;
    for (int y = threadIdx.x; y < Hh; y += blockDim.x) ay[y] = ((double)y - cy) / fy;
    __syncthreads();
    const int v = blockIdx.x;
    const float* d = depth + (size_t)v * Hh * W;
    int cnt = 0, xmin = 1 << 30, ymin = 1 << 30, xmax = -1, ymax = -1;
    double Xmin = 1e300, Xmax = -1e300, Ymin = 1e300, Ymax = -1e300;
    auto pixel = [&](float z, int x, int y) {
        if (z != 0.f) {  // depthmap_to_pointcloud keeps every row that is not all-zero
            const double X = ax[x] * (double)z, Y = ay[y] * (double)z;
            Xmin = fmin(Xmin, X); Xmax = fmax(Xmax, X); Ymin = fmin(Ymin, Y); Ymax = fmax(Ymax, Y);
            if (z > 0.f) { ++cnt; xmin = min(xmin, x); xmax = max(xmax, x); ymin = min(ymin, y); ymax = max(ymax, y); }
        }
    };
    const int npx = Hh * W;
    if ((W & 3) == 0 && (((size_t)v * npx) & 3) == 0) {   // rows are whole float4s: 16-B loads, one div per four pixels
        const float4* d4 = (const float4*)d;
        for (int i4 = threadIdx.x; i4 < npx / 4; i4 += blockDim.x) {
            const float4 z = d4[i4];
            const int i = i4 * 4, y = i / W, x = i - y * W;
            pixel(z.x, x, y); pixel(z.y, x + 1, y); pixel(z.z, x + 2, y); pixel(z.w, x + 3, y);
        }
    } else {
        for (int i = threadIdx.x; i < npx; i += blockDim.x) {
            const int y = i / W, x = i - y * W;
            pixel(d[i], x, y);
        }
    }
    __shared__ int si[5][EXT_THREADS];
    __shared__ double sd[4][EXT_THREADS];
    const int t = threadIdx.x;
    si[0][t] = cnt; si[1][t] = xmin; si[2][t] = ymin; si[3][t] = xmax; si[4][t] = ymax;
    sd[0][t] = Xmin; sd[1][t] = Xmax; sd[2][t] = Ymin; sd[3][t] = Ymax;
    __syncthreads();
    for (int s = EXT_THREADS / 2; s > 0; s >>= 1) {
        if (t < s) {
            si[0][t] += si[0][t + s];
            si[1][t] = min(si[1][t], si[1][t + s]); si[2][t] = min(si[2][t], si[2][t + s]);
            si[3][t] = max(si[3][t], si[3][t + s]); si[4][t] = max(si[4][t], si[4][t + s]);
            sd[0][t] = fmin(sd[0][t], sd[0][t + s]); sd[1][t] = fmax(sd[1][t], sd[1][t + s]);
            sd[2][t] = fmin(sd[2][t], sd[2][t + s]); sd[3][t] = fmax(sd[3][t], sd[3][t + s]);
        }
        __syncthreads();
    }
    if (t == 0) {
        int c = si[0][0], bx0 = si[1][0], by0 = si[2][0], bx1 = si[3][0], by1 = si[4][0];
        if (c < 100) {  // mask[105:315, 105:315] = True  (renderer.py:116-117, template.py:75-77)
            const int lo = 105, hx = min(315, W) - 1, hy = min(315, Hh) - 1;
            if (c == 0) { bx0 = lo; by0 = lo; bx1 = hx; by1 = hy; }
            else { bx0 = min(bx0, lo); by0 = min(by0, lo); bx1 = max(bx1, hx); by1 = max(by1, hy); }
        }
        double* o = out + (size_t)v * 8;
        o[0] = (double)bx0; o[1] = (double)by0; o[2] = (double)bx1; o[3] = (double)by1;
        o[4] = sd[1][0] > -1e299 ? sd[1][0] - sd[0][0] : 0.0;
        o[5] = sd[3][0] > -1e299 ? sd[3][0] - sd[2][0] : 0.0;
        o[6] = (double)c; o[7] = 0.0;
    }
}

// ---------------------------------------------------------------------------------------------
// RoIAlign forward (torchvision.ops.roi_align, aligned=False) as TrackingRefiner crops the photo with it
// (src/pipeline/refiner_utils.py:127-132: output 518x518, sampling_ratio=2, spatial_scale 1).  Restated from the
// published algorithm (torchvision/csrc/ops/cpu/roi_align_kernel.cpp); torchvision is not installable here, so parity with
// it is pinned only by known-answer tests.  All arithmetic in fp32 in the order written (oracle: same).
__device__ __forceinline__ float roi_bilinear(const float* __restrict__ in, int H, int W, float y, float x) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.f;
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    return w1 * in[y_low * W + x_low] + w2 * in[y_low * W + x_high] + w3 * in[y_high * W + x_low] + w4 * in[y_high * W + x_high];
}

__global__ __launch_bounds__(256) void roi_align_kernel(const float* __restrict__ images, const float* __restrict__ rois,
                                                        int n, int C, int H, int W, int PH, int PW, int sampling,
                                                        float spatial_scale, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (roi, c, ph, pw), pw fastest
    if (idx >= (size_t)n * C * PH * PW) return;
    const int pw = (int)(idx % PW), ph = (int)((idx / PW) % PH), c = (int)((idx / ((size_t)PW * PH)) % C);
    const int r = (int)(idx / ((size_t)PW * PH * C));
    const float* roi = rois + (size_t)r * 5;
    const int b = (int)roi[0];
    const float roi_start_w = roi[1] * spatial_scale, roi_start_h = roi[2] * spatial_scale;
    const float roi_end_w = roi[3] * spatial_scale, roi_end_h = roi[4] * spatial_scale;
    const float roi_w = fmaxf(roi_end_w - roi_start_w, 1.0f), roi_h = fmaxf(roi_end_h - roi_start_h, 1.0f);
    const float bin_h = roi_h / (float)PH, bin_w = roi_w / (float)PW;
    const int grid_h = sampling > 0 ? sampling : (int)ceilf(roi_h / (float)PH);
    const int grid_w = sampling > 0 ? sampling : (int)ceilf(roi_w / (float)PW);
    const float count = (float)max(grid_h * grid_w, 1);
    const float* in = images + ((size_t)b * C + c) * H * W;
    float acc = 0.f;
    for (int iy = 0; iy < grid_h; ++iy) {
        const float y = roi_start_h + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)grid_h;
        for (int ix = 0; ix < grid_w; ++ix) {
            const float x = roi_start_w + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)grid_w;
            acc += roi_bilinear(in, H, W, y, x);
        }
    }
    out[idx] = acc / count;
}

}  // namespace

int fp_crop_resize_pad_launch(const void* images, int src_u8, int n_img, int C, int H, int W, const int32_t* boxes,
                              int n, float bbox_extend, int target, const uint8_t* masks, int mask_mode, void* out,
                              int out_bf16, CropParam* params, hipStream_t s) {
    hipLaunchKernelGGL(crop_params_kernel, dim3(cdiv(n, 64)), dim3(64), 0, s, boxes, n, H, W, bbox_extend,
                       bbox_extend == 0.f ? 1 : 0, target, params);
    FP_LAUNCH_CHECK();
    dim3 grid(cdiv(target, CROP_ROWS), n), block(256);
    if ((src_u8 == 1 || src_u8 == 2) && C == 3 && mask_mode == 0 && (target & 1) == 0 && target <= 2048 && W <= 32767) {
#define FP_CROP_RGB(U, O) hipLaunchKernelGGL((crop_rgb_kernel<U, O>), grid, block, 0, s, (const uint8_t*)images, n_img, H, W, params, target, out)
        if (src_u8 == 1) { if (out_bf16) FP_CROP_RGB(1, 1); else FP_CROP_RGB(1, 0); }
        else { if (out_bf16) FP_CROP_RGB(2, 1); else FP_CROP_RGB(2, 0); }
#undef FP_CROP_RGB
        FP_LAUNCH_CHECK();
        return FP_OK;
    }
#define FP_CROP(U, O) hipLaunchKernelGGL((crop_kernel<U, O>), grid, block, 0, s, images, n_img, C, H, W, params, target, masks, mask_mode, out)
    if (src_u8 == 1) { if (out_bf16) FP_CROP(1, 1); else FP_CROP(1, 0); }
    else if (src_u8 == 2) { if (out_bf16) FP_CROP(2, 1); else FP_CROP(2, 0); }
    else { if (out_bf16) FP_CROP(0, 1); else FP_CROP(0, 0); }
#undef FP_CROP
    FP_LAUNCH_CHECK();
    return FP_OK;
}

#include "../../include/freepose_hip.h"

extern "C" int fp_crop_resize_pad(fp_ctx* ctx, const void* d_images, int src_fmt, int n_img, int C, int H, int W,
                                  const int32_t* d_boxes, int n, float bbox_extend, int target, const uint8_t* d_masks,
                                  int mask_mode, void* d_out, int out_fmt, void* stream) {
    FP_REQUIRE(ctx && d_images && d_boxes && d_out, "crop_resize_pad: null argument");
    FP_REQUIRE(n_img == 1 || n_img == n, "crop_resize_pad: n_img must be 1 or n");
    FP_REQUIRE(mask_mode == 0 || d_masks, "crop_resize_pad: mask_mode %d needs masks", mask_mode);
    if (n == 0) return FP_OK;
    CropParam* params;
    int rc;
    if ((rc = ctx->get("crop.params", (size_t)n * sizeof(CropParam), (void**)&params))) return rc;
    return fp_crop_resize_pad_launch(d_images, src_fmt, n_img, C, H, W, d_boxes, n, bbox_extend, target, d_masks,
                                     mask_mode, d_out, out_fmt, params, (hipStream_t)stream);
}

extern "C" int fp_geodesic_select(fp_ctx* ctx, const double* d_grid, int G, const double* h_R, double thresh_deg,
                                  int32_t* d_out_idx, int* h_n, void* stream) {
    FP_REQUIRE(ctx && d_grid && h_R && d_out_idx && h_n, "geodesic_select: null argument");
    hipStream_t s = (hipStream_t)stream;
    char* ws;
    int rc;
    if ((rc = ctx->get("geo.ws", (size_t)G * 4 + 256, (void**)&ws))) return rc;
    int* dn = (int*)(ws + 128);
    int32_t* flags = (int32_t*)(ws + 256);
    Rot9 Rd;
    for (int i = 0; i < 9; ++i) Rd.v[i] = h_R[i];
    hipLaunchKernelGGL(geodesic_flags_kernel, dim3(cdiv(G, 256)), dim3(256), 0, s, d_grid, G, Rd, thresh_deg, flags);
    FP_LAUNCH_CHECK();
    hipLaunchKernelGGL(compact_flags_kernel, dim3(1), dim3(1024), 0, s, flags, G, d_out_idx, dn);
    FP_LAUNCH_CHECK();
    FP_HIP(hipMemcpyAsync(h_n, dn, sizeof(int), hipMemcpyDeviceToHost, s));
    FP_HIP(hipStreamSynchronize(s));   // the one wait of the call: the caller sizes its render batch with *h_n
    return FP_OK;
}

extern "C" int fp_depth_extents(fp_ctx* ctx, const float* d_depth, int Hn, int Hh, int W, float fx, float fy, float cx,
                                float cy, double* d_out, void* stream) {
    FP_REQUIRE(ctx && d_depth && d_out, "depth_extents: null argument");
    if (Hn == 0) return FP_OK;
    hipLaunchKernelGGL(depth_extents_kernel, dim3(Hn), dim3(EXT_THREADS), (size_t)(Hh + W) * sizeof(double), (hipStream_t)stream, d_depth, Hh, W, (double)fx,
                       (double)fy, (double)cx, (double)cy, d_out);
    FP_LAUNCH_CHECK();
    return FP_OK;
}

extern "C" int fp_generate_rotations(int n, double* out) {
    FP_REQUIRE(n > 0 && out, "generate_rotations: bad argument");
    // super-Fibonacci spiral on SO(3) (pose_estimator.py:121-147); scalar-last quaternion -> matrix
    const double phi = sqrt(2.0), psi = 1.533751168755204288118041, PI = 3.14159265358979323846;
    for (int i = 0; i < n; ++i) {
        const double s = i + 0.5, r = sqrt(s / n), Rr = sqrt(1.0 - s / n);
        const double al = 2.0 * PI * s / phi, be = 2.0 * PI * s / psi;
        double x = r * sin(al), y = r * cos(al), z = Rr * sin(be), w = Rr * cos(be);
        const double nn = sqrt(x * x + y * y + z * z + w * w);
        x /= nn; y /= nn; z /= nn; w /= nn;
        double* M = out + (size_t)i * 9;
        const double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w, xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
        M[0] = x2 - y2 - z2 + w2; M[1] = 2 * (xy - zw);      M[2] = 2 * (xz + yw);
        M[3] = 2 * (xy + zw);     M[4] = -x2 + y2 - z2 + w2; M[5] = 2 * (yz - xw);
        M[6] = 2 * (xz - yw);     M[7] = 2 * (yz + xw);      M[8] = -x2 - y2 + z2 + w2;
    }
    return FP_OK;
}

extern "C" int fp_roi_align(fp_ctx* ctx, const float* d_images, int n_img, int C, int H, int W, const float* d_rois, int n,
                            int pooled_h, int pooled_w, int sampling_ratio, float spatial_scale, float* d_out, void* stream) {
    FP_REQUIRE(ctx && d_images && d_out, "roi_align: null argument");
    FP_REQUIRE(n_img > 0 && C > 0 && H > 0 && W > 0 && pooled_h > 0 && pooled_w > 0, "roi_align: bad shape");
    if (n == 0) return FP_OK;
    FP_REQUIRE(d_rois, "roi_align: null rois");
    const size_t total = (size_t)n * C * pooled_h * pooled_w;
    hipLaunchKernelGGL(roi_align_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_images,
                       d_rois, n, C, H, W, pooled_h, pooled_w, sampling_ratio, spatial_scale, d_out);
    FP_LAUNCH_CHECK();
    return FP_OK;
}
