// Pose-hypothesis rasteriser (K15 of SURVEY.md §2.3), gfx950 only.  Replaces pyrender's OpenGL path used by
// MeshRenderer.render_from_poses (src/pipeline/retrieval/renderer.py:70-95): one mesh, Hn poses, pinhole
// intrinsics, ambient-only shading, no face culling, rgb u8 + metric eye-depth f32 (0 = background).
//
// Arithmetic contract (shared with oracle/fp_oracle.c, every step fp32 IEEE unless noted):
//   vertex : s = scale*v ; Xc = fma(R00,sx, fma(R01,sy, fma(R02,sz, tx))) (same for Yc, Zc)
//            iz = 1/Zc ; u = fma(fx, Xc*iz, cx) ; v = fma(fy, Yc*iz, cy)
//            fixed point 24.8 : xi = rint(u*256), yi = rint(v*256)      (image coords, y down)
//   near plane (znear = 0.05, renderer.py:62-67 / pyrender's default): a triangle with all three Zc <= znear is dropped; one that
//   STRADDLES the plane is rasterised in homogeneous coordinates, which clips it exactly at z = znear without generating geometry:
//       vertex k -> (xk, yk, wk) = (pixel x * z, pixel y * z, z): for Zc > znear from the snapped fixed-point pixel (xi / 256 * Zc, so
//       edges shared with ordinary triangles coincide), else xk = fma(fx, Xc, cx * Zc), yk likewise (stored as float bits in xi / yi);
//       in double: a0 = y1 w2 - y2 w1, b0 = x2 w1 - x1 w2, c0 = x1 y2 - x2 y1 (cyclic for 1, 2), det = x0 a0 + y0 b0 + w0 c0;
//       pixel (px, py): X = px + 0.5, Y = py + 0.5, e_i = (a_i X + b_i Y) + c_i, all negated when det < 0;
//       covered iff e0, e1, e2 >= 0, S = (e0 + e1) + e2 > 0 and z = |det| / S > znear; depth = (float) z;
//       barycentrics (float)(e_i / S) are perspective-correct; colour / uv interpolate with them directly, textures at level 0.
//   coverage: sample (256 px + 128, 256 py + 128); int64 edge functions; orientation normalised by
//            swapping v1,v2 when the doubled area is negative (SKIP_CULL_FACES); top-left rule on ties
//   depth  : b_i = float(E_i)/float(area2) ; izp = fma(b2,iz2, fma(b1,iz1, b0*iz0)) ; depth = 1/izp
//   visibility: 64-bit key (depth bits << 32 | triangle id), atomic MIN -> nearest depth, lowest id on
//            exact ties; order independent, hence deterministic
//   base colour of the visible fragment, q_i = b_i*iz_i:
//            vertex colours : cv = fma(q2,c2, fma(q1,c1, q0*c0)) * depth                      (0..255 units)
//            textured mesh  : per-corner UV, U = fma(q2,u2, fma(q1,u1, q0*u0)) * depth (same for V).  The texture is filtered
//                             as stored (u8 values 0..255, the image's gamma space — a GL_RGBA8 texture):
//              level k      : max(1, tw>>k) x max(1, th>>k); level k+1 = 2x2 box of level k, (a+b+c+d+2)>>2 per channel (built on
//                             the host at upload; second row / column clamped when the source size is 1)
//              bilinear(k)  : x = U*wk - 0.5, y = (1-V)*hk - 0.5 (v = 1 at the image's first row), (x0,y0) = floor, REPEAT wrap,
//                             top = fma(wx, t01-t00, t00), bot = fma(wx, t11-t10, t10), val = fma(wy, bot-top, top)
//              level of detail (filter 1, default: GL_LINEAR_MIPMAP_LINEAR with ANALYTIC derivatives — what pyrender's sampler
//                             asks GL for): g?_i = d(b_i)/d(px|py) * iz_i are constants of the triangle;
//                             dU/dx = fma(gx2,u2-U, fma(gx1,u1-U, gx0*(u0-U))) * depth (same for V, y);
//                             rho2 = max((dU/dx*tw)^2 + (dV/dx*th)^2, (dU/dy*tw)^2 + (dV/dy*th)^2);
//                             rho2 <= 1 (magnification) or a single level: bilinear(0); else e = exponent(rho2), m = mantissa,
//                             lg = LOG2P(m-1) (degree-5 polynomial in fmaf steps, |err| < 2e-5), l0 = e>>1,
//                             fr = 0.5*((e&1) + lg); l0 >= last: bilinear(last); else val = fma(fr, bilinear(l0+1)-bilinear(l0), bilinear(l0))
//                             filter 0: bilinear(0) always
//              shade 1      : lin = sRGB->linear of val/255 by linear interpolation in DEC[256] (i = min(int(val),254),
//                             lin = fma(val-i, DEC[i+1]-DEC[i], DEC[i])): sample, THEN decode (a shader's srgb_to_linear(texture()));
//                             c = lin*Kd.      shade 0: c255 = val*Kd
//   output : shade 1 (default, "gamma"): u8 = #{k in 1..255 : THR[k] <= a*c}, THR[k] = ((k-.5)/255)^2.2, i.e.
//            round(255 (a c)^(1/2.2)) found by table search so that host oracle and device agree bit for bit (vertex: c = cv/255)
//            shade 0 ("linear", the round-1 rule): (u8) min(255, a*cv + 0.5)                  (textured: cv = c255)
//            a = ambient light factor: 2 by default (renderer.py:53-55), 5 in tracking_refiner.py:33.
//   pyrender's fragment shader is third-party and absent from /root/reference: the shading rule is STATED here, not pinned
//   (DESIGN.md §5).  A GPU derives the level of detail from 2x2-quad finite differences with a few fractional bits; the analytic
//   derivative used here is the quantity those approximate.
// Launch shape: vertex kernel (Hn x V threads), triangle kernel (Hn x F threads; small triangles are
// rasterised by their thread, large ones by a whole wave via a queue), resolve kernel (Hn x pixels).
#include "../../include/freepose_hip.h"
#include "internal.h"

#include <algorithm>

struct fp_mesh {
    fp_ctx* ctx = nullptr;
    float* verts = nullptr;    // [V,3]
    int32_t* faces = nullptr;  // [F,3]
    int32_t* perm = nullptr;   // [F] slot -> face id, Morton order of the object-space centroids (tiled path: coherent 64-triangle chunks)
    int32_t* fsort = nullptr;  // [F,3] faces[perm[slot]]: the corner ids in slot order (read coalesced by the bin / tile kernels)
    std::vector<int32_t> h_vmap;   // host copy of vmap (per-vertex attributes are stored by device id)
    int32_t* vmap = nullptr;   // [V] caller's vertex id -> device vertex id (vertices are stored in order of first use by the Morton-sorted
                               // triangles, so the 3 x 64 vertex gathers of a chunk hit a few cache lines instead of one line each)
    uint8_t* colors = nullptr; // [V,4] rgba (a unused)
    float* uv = nullptr;       // [F,3,2] per-corner texture coordinates (textured meshes)
    uint8_t* tex = nullptr;    // rgba diffuse texture, all mip levels back to back (level k at texel offset lev_off[k])
    int nlev = 0;
    uint32_t lev_off[16] = {};
    int filter = 1;            // 1 = trilinear mip-maps, 0 = bilinear level 0
    int cull = 0;              // 1 = back faces are not drawn (renderer.py:63-66: render without SKIP_CULL_FACES), 0 = both sides
    float* tables = nullptr;   // DEC[256] sRGB->linear, THR[256] gamma-encode thresholds
    int th = 0, tw = 0;
    float kd[3] = {1.f, 1.f, 1.f};   // material diffuse factor (MTL Kd / glTF baseColorFactor)
    int shade = 1;             // 1 = gamma rule, 0 = linear (see the contract above)
    int V = 0, F = 0;
    float ambient = 2.0f;      // scene ambient light factor: 2 in renderer.py:53-55, 5 in tracking_refiner.py:33
};

namespace {

struct SVert { int xi, yi; float iz, zc; };
struct RasterView { float R[9]; float t[3]; };

constexpr float ZNEAR = 0.05f;
constexpr int BIG_AREA = 256;  // bbox pixels above which a triangle goes to the wave-per-triangle queue

__global__ void raster_vertex_kernel(const float* __restrict__ verts, int V, const float* __restrict__ poses, int Hn,
                                     float scale, float fx, float fy, float cx, float cy, SVert* __restrict__ sv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (i >= V) return;
    const float* P = poses + (size_t)h * 16;
    const float sx = scale * verts[3 * i], sy = scale * verts[3 * i + 1], sz = scale * verts[3 * i + 2];
    const float Xc = fmaf(P[0], sx, fmaf(P[1], sy, fmaf(P[2], sz, P[3])));
    const float Yc = fmaf(P[4], sx, fmaf(P[5], sy, fmaf(P[6], sz, P[7])));
    const float Zc = fmaf(P[8], sx, fmaf(P[9], sy, fmaf(P[10], sz, P[11])));
    SVert o;
    o.zc = Zc;
    if (Zc > ZNEAR) {
        const float iz = 1.0f / Zc;
        const float u = fmaf(fx, Xc * iz, cx), v = fmaf(fy, Yc * iz, cy);
        // clamp far-off-screen coordinates so the fixed-point products stay inside int64
        const float uc = fminf(fmaxf(u, -30000.f), 30000.f), vc = fminf(fmaxf(v, -30000.f), 30000.f);
        o.xi = (int)rintf(uc * 256.0f);
        o.yi = (int)rintf(vc * 256.0f);
        o.iz = iz;
    } else {   // behind the near plane: homogeneous pixel coordinates for straddling triangles (contract in the header)
        o.xi = (int)__float_as_uint(fmaf(fx, Xc, cx * Zc));
        o.yi = (int)__float_as_uint(fmaf(fy, Yc, cy * Zc));
        o.iz = 0.f;
    }
    sv[(size_t)h * V + i] = o;
}

struct TriSetup {
    int x0, y0, x1, y1, x2, y2;
    float iz0, iz1, iz2;
    long long area2;
    int bx0, by0, bx1, by1;  // pixel bbox, inclusive, clipped
    int i0, i1, i2;          // vertex ids after orientation normalisation
    bool ok;
    bool swapped;            // corners 1 and 2 were exchanged (per-corner attributes follow)
    bool strad;              // straddles the near plane: homogeneous rasterisation over the whole frame (struct Strad)
    bool small;              // spans < 128 px in x and y: every edge function at a candidate pixel fits 32 bits (tri_cover32)
};

// homogeneous edge functions of a triangle that straddles the near plane (contract in the header)
struct Strad { double a[3], b[3], c[3], det; };
__device__ __forceinline__ void strad_vertex(const SVert& v, double& x, double& y, double& w) {
    w = (double)v.zc;
    if (v.zc > ZNEAR) {
        x = ((double)v.xi * (1.0 / 256.0)) * w;
        y = ((double)v.yi * (1.0 / 256.0)) * w;
    } else {
        x = (double)__uint_as_float((unsigned)v.xi);
        y = (double)__uint_as_float((unsigned)v.yi);
    }
}
__device__ __forceinline__ Strad strad_setup(const SVert* __restrict__ sv, const int32_t* __restrict__ faces, int f) {
    double x[3], y[3], w[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) strad_vertex(sv[faces[3 * f + k]], x[k], y[k], w[k]);
    Strad q;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        q.a[k] = y[k1] * w[k2] - y[k2] * w[k1];
        q.b[k] = x[k2] * w[k1] - x[k1] * w[k2];
        q.c[k] = x[k1] * y[k2] - x[k2] * y[k1];
    }
    q.det = (x[0] * q.a[0] + y[0] * q.b[0]) + w[0] * q.c[0];
    return q;
}
// coverage, depth and perspective-correct barycentrics of pixel (px, py)
__device__ __forceinline__ bool strad_pixel(const Strad& q, int px, int py, float& d, float& b0, float& b1, float& b2) {
    const double X = (double)px + 0.5, Y = (double)py + 0.5;
    double e0 = (q.a[0] * X + q.b[0] * Y) + q.c[0];
    double e1 = (q.a[1] * X + q.b[1] * Y) + q.c[1];
    double e2 = (q.a[2] * X + q.b[2] * Y) + q.c[2];
    double det = q.det;
    if (det < 0.0) { e0 = -e0; e1 = -e1; e2 = -e2; det = -det; }
    if (!(det > 0.0) || e0 < 0.0 || e1 < 0.0 || e2 < 0.0) return false;
    const double S = (e0 + e1) + e2;
    if (!(S > 0.0)) return false;
    const double z = det / S;
    if (!(z > (double)ZNEAR)) return false;
    d = (float)z;
    b0 = (float)(e0 / S); b1 = (float)(e1 / S); b2 = (float)(e2 / S);
    return true;
}

__device__ __forceinline__ bool topleft(int dx, int dy) { return (dy < 0) || (dy == 0 && dx > 0); }

// set-up from the three screen-space vertices already in registers (callers that batch their gathers load them first)
__device__ __forceinline__ TriSetup tri_setup_v(SVert a, SVert b, SVert c, int i0, int i1, int i2, int W, int Hh) {
    TriSetup t;
    const int nfront = (a.zc > ZNEAR) + (b.zc > ZNEAR) + (c.zc > ZNEAR);
    t.ok = nfront == 3;
    t.strad = nfront == 1 || nfront == 2;
    if (t.strad) {   // near-plane straddler: candidate pixels = the whole frame, original corner order, no fixed-point set-up
        t.ok = true; t.swapped = false; t.area2 = 1; t.small = false;
        t.x0 = t.y0 = t.x1 = t.y1 = t.x2 = t.y2 = 0;
        t.iz0 = t.iz1 = t.iz2 = 0.f;
        t.i0 = i0; t.i1 = i1; t.i2 = i2;
        t.bx0 = 0; t.by0 = 0; t.bx1 = W - 1; t.by1 = Hh - 1;
        return t;
    }
    long long area2 = (long long)(b.xi - a.xi) * (c.yi - a.yi) - (long long)(b.yi - a.yi) * (c.xi - a.xi);
    t.swapped = area2 < 0;
    if (area2 < 0) {
        SVert tmp = b; b = c; c = tmp;
        int ti = i1; i1 = i2; i2 = ti;
        area2 = -area2;
    }
    t.area2 = area2;
    if (area2 == 0) t.ok = false;
    t.x0 = a.xi; t.y0 = a.yi; t.x1 = b.xi; t.y1 = b.yi; t.x2 = c.xi; t.y2 = c.yi;
    t.iz0 = a.iz; t.iz1 = b.iz; t.iz2 = c.iz;
    t.i0 = i0; t.i1 = i1; t.i2 = i2;
    const int mnx = min(a.xi, min(b.xi, c.xi)), mxx = max(a.xi, max(b.xi, c.xi));
    const int mny = min(a.yi, min(b.yi, c.yi)), mxy = max(a.yi, max(b.yi, c.yi));
    // pixel p is a candidate when its centre 256p+128 lies in [mn, mx]
    t.bx0 = max(0, (mnx - 128 + 255) >> 8);
    t.by0 = max(0, (mny - 128 + 255) >> 8);
    t.bx1 = min(W - 1, (mxx - 128) >> 8);
    t.by1 = min(Hh - 1, (mxy - 128) >> 8);
    if (t.bx1 < t.bx0 || t.by1 < t.by0) t.ok = false;
    t.small = (mxx - mnx) < 32768 && (mxy - mny) < 32768;
    return t;
}
__device__ __forceinline__ TriSetup tri_setup(const SVert* __restrict__ sv, const int32_t* __restrict__ faces, int f,
                                              int W, int Hh) {
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    return tri_setup_v(sv[i0], sv[i1], sv[i2], i0, i1, i2, W, Hh);
}

// edge functions at the pixel centre; returns coverage and the three (non-negative) weights
__device__ __forceinline__ bool tri_cover(const TriSetup& t, int px, int py, long long& w0, long long& w1, long long& w2) {
    const long long sx = (long long)px * 256 + 128, sy = (long long)py * 256 + 128;
    // E_ab(p) = (bx-ax)(py-ay) - (by-ay)(px-ax); positive inside for area2 > 0
    w0 = (long long)(t.x2 - t.x1) * (sy - t.y1) - (long long)(t.y2 - t.y1) * (sx - t.x1);  // opposite v0
    w1 = (long long)(t.x0 - t.x2) * (sy - t.y2) - (long long)(t.y0 - t.y2) * (sx - t.x2);  // opposite v1
    w2 = (long long)(t.x1 - t.x0) * (sy - t.y0) - (long long)(t.y1 - t.y0) * (sx - t.x0);  // opposite v2
    if (w0 < 0 || w1 < 0 || w2 < 0) return false;
    if (w0 == 0 && !topleft(t.x2 - t.x1, t.y2 - t.y1)) return false;
    if (w1 == 0 && !topleft(t.x0 - t.x2, t.y0 - t.y2)) return false;
    if (w2 == 0 && !topleft(t.x1 - t.x0, t.y1 - t.y0)) return false;
    return true;
}

// The same edge functions for a `small` triangle at one of its candidate pixels (centre inside the vertex bounding box): every
// difference is below 2^15 in magnitude, so the products fit 24 x 24 -> 32-bit multiplies (v_mul_i32_i24, full rate; a 64-bit product
// is four quarter-rate multiplies) and the sums 32 bits: the SAME integers as tri_cover, hence the same coverage, the same float
// conversions and the same depth bits — what the dense meshes' ~1-pixel triangles spend their time on.
__device__ __forceinline__ bool tri_cover32(const TriSetup& t, int px, int py, int& w0, int& w1, int& w2) {
    const int sx = px * 256 + 128, sy = py * 256 + 128;
    w0 = __mul24(t.x2 - t.x1, sy - t.y1) - __mul24(t.y2 - t.y1, sx - t.x1);
    w1 = __mul24(t.x0 - t.x2, sy - t.y2) - __mul24(t.y0 - t.y2, sx - t.x2);
    w2 = __mul24(t.x1 - t.x0, sy - t.y0) - __mul24(t.y1 - t.y0, sx - t.x0);
    if ((w0 | w1 | w2) < 0) return false;
    if (w0 == 0 && !topleft(t.x2 - t.x1, t.y2 - t.y1)) return false;
    if (w1 == 0 && !topleft(t.x0 - t.x2, t.y0 - t.y2)) return false;
    if (w2 == 0 && !topleft(t.x1 - t.x0, t.y1 - t.y0)) return false;
    return true;
}
// b_i = float(w_i) / float(area2), the IEEE quotient, WITHOUT its ~12-instruction expansion per weight: the reciprocal of the area is
// taken once per triangle in double (rd = RN(1 / fa)), each weight is then one fp64 multiply rounded to float.  The double product is
// within 2^-52 (relative) of the exact quotient; a quotient of two 24-bit significands is never closer than 2^-49 to a rounding
// midpoint of the float grid (|x/n - m| >= 1 / (N k) for a 25-bit midpoint significand k), so rounding the product gives the
// correctly rounded quotient: the bits of the division (and of the oracle).
__device__ __forceinline__ double tri_rcp_area(const TriSetup& t) { return 1.0 / (double)(float)(int)t.area2; }
__device__ __forceinline__ float tri_depth32(const TriSetup& t, double rd, int w0, int w1, int w2, float& b0, float& b1, float& b2) {
    b0 = (float)((double)(float)w0 * rd); b1 = (float)((double)(float)w1 * rd); b2 = (float)((double)(float)w2 * rd);
    const float izp = fmaf(b2, t.iz2, fmaf(b1, t.iz1, b0 * t.iz0));
    return 1.0f / izp;
}

__device__ __forceinline__ float tri_depth(const TriSetup& t, long long w0, long long w1, long long w2, float& b0,
                                           float& b1, float& b2) {
    const float fa = (float)t.area2;
    b0 = (float)w0 / fa; b1 = (float)w1 / fa; b2 = (float)w2 / fa;
    const float izp = fmaf(b2, t.iz2, fmaf(b1, t.iz1, b0 * t.iz0));
    return 1.0f / izp;
}

__device__ __forceinline__ void tri_pixel(const TriSetup& t, int f, int px, int py, unsigned long long* __restrict__ zb,
                                          int W) {
    long long w0, w1, w2;
    if (!tri_cover(t, px, py, w0, w1, w2)) return;
    float b0, b1, b2;
    const float d = tri_depth(t, w0, w1, w2, b0, b1, b2);
    if (!(d > 0.f)) return;
    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)f;
    // (a read-before-atomic "cannot win" test was measured here: the scattered 8-byte loads cost more than the L2 atomics
    //  they save — 2.4 -> 3.5 ms per 576 views — so the atomic is issued unconditionally)
    atomicMin(&zb[(size_t)py * W + px], key);
}

// Back face (renderer.py:63-66, cull_faces = True -> GL_CULL_FACE with counter-clockwise front faces): in this frame (x right, y down,
// z forward) a counter-clockwise-from-outside triangle that faces the camera has a NEGATIVE screen-space area, i.e. tri_setup swapped
// its corners; a straddler is classified by the sign of its homogeneous determinant (same sign as the area when all w > 0).
__device__ __forceinline__ bool back_facing(const TriSetup& t, const SVert* __restrict__ sv, const int32_t* __restrict__ faces, int f) {
    if (t.strad) return strad_setup(sv, faces, f).det > 0.0;
    return !t.swapped;
}

__device__ __forceinline__ void strad_pixel_global(const Strad& q, int f, int px, int py, unsigned long long* __restrict__ zb, int W) {
    float d, b0, b1, b2;
    if (!strad_pixel(q, px, py, d, b0, b1, b2)) return;
    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)f;
    atomicMin(&zb[(size_t)py * W + px], key);
}

// ---- shading of the visible fragment (shared by both visibility strategies; arithmetic = the contract in the header) ----
struct ShadeArgs {
    const uint8_t* colors;   // [V,4] or null
    const float* uv;         // [F,3,2] or null
    const uint8_t* tex;      // rgba mip chain or null
    int th, tw;
    float kd0, kd1, kd2;
    float ambient;
    int shade;
    int nlev, filter;
    uint32_t lev_off[16];
};

// largest k in [0,255] with thr[k] <= x (thr[0] = 0); the pow() estimate only decides how many table steps are taken
__device__ __forceinline__ uint8_t encode_gamma(float x, const float* thr) {
    if (!(x > 0.f)) return 0;
    int k = (int)(255.0f * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(fminf(x, 1.0f)) * 0.45454545f) + 0.5f);
    k = min(max(k, 0), 255);
    while (k < 255 && thr[k + 1] <= x) ++k;
    while (k > 0 && thr[k] > x) --k;
    return (uint8_t)k;
}
__device__ __forceinline__ int wrapi(int a, int n) { const int m = a % n; return m < 0 ? m + n : m; }

// bilinear sample of one mip level (values in 0..255 units)
__device__ __forceinline__ void bilinear_level(const uint8_t* __restrict__ lev, int w, int h, float U, float Vv, float (&out)[3]) {
    float x = fmaf(U, (float)w, -0.5f), y = fmaf(1.0f - Vv, (float)h, -0.5f);
    x = fminf(fmaxf(x, -1.0e6f), 1.0e6f); y = fminf(fmaxf(y, -1.0e6f), 1.0e6f);
    if (!(x == x)) x = 0.f;
    if (!(y == y)) y = 0.f;
    const float xf = floorf(x), yf = floorf(y);
    const float wx = x - xf, wy = y - yf;
    const int x0 = wrapi((int)xf, w), x1 = wrapi((int)xf + 1, w);
    const int y0 = wrapi((int)yf, h), y1 = wrapi((int)yf + 1, h);
    const uint32_t e00 = *(const uint32_t*)(lev + ((size_t)y0 * w + x0) * 4), e01 = *(const uint32_t*)(lev + ((size_t)y0 * w + x1) * 4);
    const uint32_t e10 = *(const uint32_t*)(lev + ((size_t)y1 * w + x0) * 4), e11 = *(const uint32_t*)(lev + ((size_t)y1 * w + x1) * 4);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float t00 = (float)((e00 >> (8 * c)) & 255), t01 = (float)((e01 >> (8 * c)) & 255);
        const float t10 = (float)((e10 >> (8 * c)) & 255), t11 = (float)((e11 >> (8 * c)) & 255);
        const float top = fmaf(wx, t01 - t00, t00), bot = fmaf(wx, t11 - t10, t10);
        out[c] = fmaf(wy, bot - top, top);
    }
}
// LOG2P(t) ~ log2(1 + t) on [0,1)
__device__ __forceinline__ float log2p(float t) {
    float a = 0.045148879289627075f;
    a = fmaf(a, t, -0.19357527792453766f);
    a = fmaf(a, t, 0.41560569405555725f);
    a = fmaf(a, t, -0.7090963125228882f);
    a = fmaf(a, t, 1.441917061805725f);
    return a * t;
}

// tab: DEC[256] then THR[256] (LDS copy)
// per-triangle shading attributes, loaded by the caller (so a batch of pixels can have all its gathers in flight at once):
// vertex colours as rgba words of the triangle's ORIGINAL corners (face order) or the six per-corner uv floats
struct TriAttr { uint32_t cw[3]; float tc[6]; };
__device__ __forceinline__ TriAttr load_attr(const ShadeArgs& s, int f, int v0, int v1, int v2) {
    TriAttr a;
    a.cw[0] = a.cw[1] = a.cw[2] = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < 6; ++k) a.tc[k] = 0.f;
    if (s.uv) {
        const float2* tc = (const float2*)(s.uv + (size_t)f * 6);
        const float2 t0 = tc[0], t1 = tc[1], t2 = tc[2];
        a.tc[0] = t0.x; a.tc[1] = t0.y; a.tc[2] = t1.x; a.tc[3] = t1.y; a.tc[4] = t2.x; a.tc[5] = t2.y;
    } else if (s.colors) {
        const uint32_t* cw = (const uint32_t*)s.colors;
        a.cw[0] = cw[v0]; a.cw[1] = cw[v1]; a.cw[2] = cw[v2];
    }
    return a;
}

// `at`: attributes in FACE order (corner k of faces[3f + k]); t.swapped says whether set-up exchanged corners 1 and 2
__device__ __forceinline__ void shade_fragment(const ShadeArgs& s, const TriSetup& t, const TriAttr& at, float q0, float q1, float q2,
                                               float dd, const float* tab, uint8_t (&out)[3], bool lod0 = false) {
    const float* dec = tab;
    const float* thr = tab + 256;
    if (s.uv) {
        const float* tc = at.tc;
        const int k1 = t.swapped ? 2 : 1, k2 = t.swapped ? 1 : 2;
        const float u0 = tc[0], u1 = tc[2 * k1], u2 = tc[2 * k2];
        const float v0 = tc[1], v1 = tc[2 * k1 + 1], v2 = tc[2 * k2 + 1];
        const float U = fmaf(q2, u2, fmaf(q1, u1, q0 * u0)) * dd;
        const float Vv = fmaf(q2, v2, fmaf(q1, v1, q0 * v0)) * dd;
        int l0 = 0;
        bool two = false;
        float fr = 0.f;
        if (s.filter && s.nlev > 1 && !lod0) {
            const float fa = (float)t.area2;
            // d(w_i)/d(px) = -256 (y_b - y_a), d(w_i)/d(py) = 256 (x_b - x_a) of the edge opposite corner i
            const float gx0 = (float)(-(long long)(t.y2 - t.y1) * 256) / fa * t.iz0, gy0 = (float)((long long)(t.x2 - t.x1) * 256) / fa * t.iz0;
            const float gx1 = (float)(-(long long)(t.y0 - t.y2) * 256) / fa * t.iz1, gy1 = (float)((long long)(t.x0 - t.x2) * 256) / fa * t.iz1;
            const float gx2 = (float)(-(long long)(t.y1 - t.y0) * 256) / fa * t.iz2, gy2 = (float)((long long)(t.x1 - t.x0) * 256) / fa * t.iz2;
            const float dux = fmaf(gx2, u2 - U, fmaf(gx1, u1 - U, gx0 * (u0 - U))) * dd * (float)s.tw;
            const float dvx = fmaf(gx2, v2 - Vv, fmaf(gx1, v1 - Vv, gx0 * (v0 - Vv))) * dd * (float)s.th;
            const float duy = fmaf(gy2, u2 - U, fmaf(gy1, u1 - U, gy0 * (u0 - U))) * dd * (float)s.tw;
            const float dvy = fmaf(gy2, v2 - Vv, fmaf(gy1, v1 - Vv, gy0 * (v0 - Vv))) * dd * (float)s.th;
            float r2 = fmaxf(fmaf(dux, dux, dvx * dvx), fmaf(duy, duy, dvy * dvy));
            if (!(r2 == r2)) r2 = 0.f;
            if (r2 > 1.0f) {
                const uint32_t rb = __float_as_uint(r2);
                const int e = (int)(rb >> 23) - 127;
                const float m = __uint_as_float((rb & 0x7fffffu) | 0x3f800000u);
                const float lg = log2p(m - 1.0f);
                l0 = e >> 1;
                fr = 0.5f * ((float)(e & 1) + lg);
                if (l0 >= s.nlev - 1) { l0 = s.nlev - 1; fr = 0.f; } else two = true;
            }
        }
        float val[3];
        bilinear_level(s.tex + (size_t)s.lev_off[l0] * 4, max(1, s.tw >> l0), max(1, s.th >> l0), U, Vv, val);
        if (two) {
            float hi[3];
            bilinear_level(s.tex + (size_t)s.lev_off[l0 + 1] * 4, max(1, s.tw >> (l0 + 1)), max(1, s.th >> (l0 + 1)), U, Vv, hi);
#pragma unroll
            for (int c = 0; c < 3; ++c) val[c] = fmaf(fr, hi[c] - val[c], val[c]);
        }
        const float kd[3] = {s.kd0, s.kd1, s.kd2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (s.shade) {
                const int ii = min(max((int)val[c], 0), 254);
                const float lin = fmaf(val[c] - (float)ii, dec[ii + 1] - dec[ii], dec[ii]);
                out[c] = encode_gamma(s.ambient * (lin * kd[c]), thr);
            } else {
                out[c] = (uint8_t)fmaxf(fminf(s.ambient * (val[c] * kd[c]) + 0.5f, 255.0f), 0.f);
            }
        }
    } else {
        const uint32_t w0 = at.cw[0], w1 = at.cw[t.swapped ? 2 : 1], w2 = at.cw[t.swapped ? 1 : 2];   // (white = 255 when the mesh has no colours)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float c0 = (float)((w0 >> (8 * c)) & 255u), c1 = (float)((w1 >> (8 * c)) & 255u), c2 = (float)((w2 >> (8 * c)) & 255u);
            const float cv = fmaf(q2, c2, fmaf(q1, c1, q0 * c0)) * dd;
            if (s.shade) out[c] = encode_gamma(s.ambient * (cv * (1.0f / 255.f)), thr);
            else out[c] = (uint8_t)fmaxf(fminf(s.ambient * cv + 0.5f, 255.0f), 0.f);
        }
    }
}

// resolve one pixel whose visibility key is `key`: depth and colour of the winning triangle.  `t` / `at`: the winner's set-up and
// attributes (loaded by the caller; ignored for a background pixel)
__device__ __forceinline__ void resolve_pixel_v(const SVert* __restrict__ sv, const int32_t* __restrict__ faces, const ShadeArgs& s,
                                                const float* tab, unsigned long long key, const TriSetup& t, const TriAttr& at, int px,
                                                int py, float& d, uint8_t (&out)[3]) {
    d = 0.f;
    out[0] = out[1] = out[2] = 0;
    if (key == ~0ull) return;
    const int f = (int)(unsigned)(key & 0xffffffffu);
    d = __uint_as_float((unsigned)(key >> 32));
    float b0, b1, b2;
    if (t.strad) {   // near-plane straddler: true (3-D) barycentrics, level-0 texture
        const Strad q = strad_setup(sv, faces, f);
        float ds;
        strad_pixel(q, px, py, ds, b0, b1, b2);
        shade_fragment(s, t, at, b0, b1, b2, 1.0f, tab, out, true);
        return;
    }
    // the winner's weights again (three IEEE quotients: per PIXEL that is cheaper than the fp64 reciprocal the raster pass amortises
    // over a triangle); its depth is the key's — the same bits the raster pass computed from the same weights
    const float fa = (float)t.area2;
    if (t.small) {
        int w0, w1, w2;
        tri_cover32(t, px, py, w0, w1, w2);
        b0 = (float)w0 / fa; b1 = (float)w1 / fa; b2 = (float)w2 / fa;
    } else {
        long long w0, w1, w2;
        tri_cover(t, px, py, w0, w1, w2);
        b0 = (float)w0 / fa; b1 = (float)w1 / fa; b2 = (float)w2 / fa;
    }
    shade_fragment(s, t, at, b0 * t.iz0, b1 * t.iz1, b2 * t.iz2, d, tab, out);
}
__device__ __forceinline__ void resolve_pixel(const SVert* __restrict__ sv, const int32_t* __restrict__ faces, const ShadeArgs& s,
                                              const float* tab, unsigned long long key, int px, int py, int W, int Hh, float& d,
                                              uint8_t (&out)[3]) {
    if (key == ~0ull) { d = 0.f; out[0] = out[1] = out[2] = 0; return; }
    const int f = (int)(unsigned)(key & 0xffffffffu);
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    const TriAttr at = load_attr(s, f, i0, i1, i2);
    const TriSetup t = tri_setup_v(sv[i0], sv[i1], sv[i2], i0, i1, i2, W, Hh);
    resolve_pixel_v(sv, faces, s, tab, key, t, at, px, py, d, out);
}

__global__ __launch_bounds__(256) void raster_tri_kernel(const SVert* __restrict__ sv_all, const int32_t* __restrict__ faces,
                                                         const int32_t* __restrict__ perm, int V, int F, int W, int Hh,
                                                         unsigned long long* __restrict__ zb_all,
                                                         int* __restrict__ queue, int* __restrict__ qcount, int qcap, int cull) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (slot >= F) return;
    const int f = perm[slot];     // Morton order of the centroids: a wave's 64 triangles (and their atomics) are neighbours on the screen whatever the file's face order
    const SVert* sv = sv_all + (size_t)h * V;
    unsigned long long* zb = zb_all + (size_t)h * W * Hh;
    const TriSetup t = tri_setup(sv, faces, f, W, Hh);
    if (!t.ok) return;
    if (cull && back_facing(t, sv, faces, f)) return;
    const int area = (t.bx1 - t.bx0 + 1) * (t.by1 - t.by0 + 1);
    if (area > BIG_AREA || t.strad) {
        const int slot = atomicAdd(qcount, 1);
        if (slot < qcap) { queue[2 * slot] = h; queue[2 * slot + 1] = f; return; }
        // queue full: fall through and rasterise here (slow but correct)
    }
    if (t.strad) {
        const Strad q = strad_setup(sv, faces, f);
        for (int py = 0; py < Hh; ++py)
            for (int px = 0; px < W; ++px) strad_pixel_global(q, f, px, py, zb, W);
        return;
    }
    for (int py = t.by0; py <= t.by1; ++py)
        for (int px = t.bx0; px <= t.bx1; ++px) tri_pixel(t, f, px, py, zb, W);
}

// one wave per queued (view, triangle): lanes stride over the bbox pixels
__global__ __launch_bounds__(256) void raster_big_kernel(const SVert* __restrict__ sv_all, const int32_t* __restrict__ faces,
                                                         int V, int W, int Hh, unsigned long long* __restrict__ zb_all,
                                                         const int* __restrict__ queue, const int* __restrict__ qcount,
                                                         int qcap) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    const int n = min(*qcount, qcap);
    for (int q = wave; q < n; q += nwave) {
        const int h = queue[2 * q], f = queue[2 * q + 1];
        const TriSetup t = tri_setup(sv_all + (size_t)h * V, faces, f, W, Hh);
        unsigned long long* zb = zb_all + (size_t)h * W * Hh;
        const int bw = t.bx1 - t.bx0 + 1, bh = t.by1 - t.by0 + 1;
        if (t.strad) {   // wave-uniform
            const Strad sq = strad_setup(sv_all + (size_t)h * V, faces, f);
            for (int i = lane; i < bw * bh; i += 64) strad_pixel_global(sq, f, i % bw, i / bw, zb, W);
            continue;
        }
        for (int i = lane; i < bw * bh; i += 64) tri_pixel(t, f, t.bx0 + i % bw, t.by0 + i / bw, zb, W);
    }
}

__global__ __launch_bounds__(256) void raster_resolve_kernel(const SVert* __restrict__ sv_all,
                                                             const int32_t* __restrict__ faces, ShadeArgs sh,
                                                             const float* __restrict__ tables, int V, int W, int Hh,
                                                             const unsigned long long* __restrict__ zb_all,
                                                             uint8_t* __restrict__ rgb, float* __restrict__ depth) {
    __shared__ float tab[512];
    tab[threadIdx.x] = tables[threadIdx.x];
    tab[threadIdx.x + 256] = tables[threadIdx.x + 256];
    __syncthreads();
    const int h = blockIdx.y;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= W * Hh) return;
    const unsigned long long key = zb_all[(size_t)h * W * Hh + pix];
    const int py = pix / W, px = pix - py * W;
    float d;
    uint8_t out[3];
    resolve_pixel(sv_all + (size_t)h * V, faces, sh, tab, key, px, py, W, Hh, d, out);
    depth[(size_t)h * W * Hh + pix] = d;
    uint8_t* o = rgb + ((size_t)h * W * Hh + pix) * 3;
    o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
}

// ---------------------------------------------------------------------------------------------------------------------
// Tiled path (images up to 704 px, any triangle count): the visibility keys of one screen tile live in LDS, so there is no global
// visibility buffer (no 8 B/pixel clear, no L2 atomics, no resolve read) — LDS triangle binning + per-wave depth test.
//   upload      : the triangles are put in Morton order of their object-space centroids once per mesh (`perm`: slot -> face id), so
//                 64 consecutive slots are neighbours on the surface and, under any pose, on the screen.  The visibility key still
//                 carries the ORIGINAL face id: results do not depend on the order.
//   bin kernel  : one wave per chunk of 64 slots.  Per (view, slot) the tile range of the triangle's clipped pixel bbox as four
//                 nibbles (16 bit), and per chunk the OR of its triangles' tile masks (64 bit, tiles <= 8 x 8) by a wave reduction.
//   tile kernel : one workgroup per (view, tile), all tiles of a view on one XCD (its screen-space vertices stay in that L2).
//                 The chunk masks are read 2048 at a time (coalesced) and the chunks that touch the tile compacted into an LDS hit
//                 list; each WAVE then takes hit chunks on its own (one triangle per lane, the next chunks' reads in flight): the
//                 candidate pixels of the chunk's triangles are flattened over the wave's lanes and depth-tested into LDS with
//                 ds_min_u64 — 32-bit edge functions for triangles below 128 px (all of a dense mesh's), a whole-wave loop for the
//                 rest and for near-plane straddlers.  A tile no chunk touches is written as background at once.  Otherwise the tile's pixels are
//                 resolved (same colour / depth arithmetic as the global path), packed into their own LDS slot, and flushed row by
//                 row as whole dwords.  The epilogue also reduces what fp_depth_extents would compute from the depth image
//                 (count, pixel bbox, fp64 cloud extents: min / max / integer sums, order-independent, hence bit-identical) to one
//                 partial record per tile; raster_extents_kernel folds the <= 64 records of a view.  The depth image itself is
//                 optional: the pose hot path needs only the extents (pose_estimator.py:104-112) and the crops.
// Same tri_setup / tri_cover / tri_depth, same candidate pixel set (bbox ∩ tile over all tiles = bbox), same 64-bit key and
// an order-independent minimum: the output is bit-identical to the global-buffer path and to the oracle.
constexpr int BIN_CHUNK = 64;
constexpr int BIG_TILE_AREA = 32;       // candidate pixels in the tile above which a triangle is handed to the whole wave
constexpr int HITS_ROUND = 2048;         // chunk masks examined per round of the tile kernel (8 per thread; 16-bit ids relative to the round's base)
constexpr uint16_t TBOX_NONE = 0x000f;   // tx0 = 15 > tx1 = 0: overlaps nothing
constexpr int PART_N = 10;               // doubles per (view, tile) extents record

// Workgroup -> (item, view) with all items of a view on ONE XCD: workgroup L of a launch runs on XCD L % 8 (observed; a speed hint
// only, MI355X_MICROARCH.md "Workgroup dispatch"), so within a group of 8 views the linear id walks the views fastest.
__device__ __forceinline__ void view_major_ids(int nx, int Hn, int& item, int& h) {
    const int L = blockIdx.y * nx + blockIdx.x;
    const int full = Hn & ~7, per_group = 8 * nx;
    if (L < (full >> 3) * per_group) {
        const int g = L / per_group, r = L - g * per_group;
        h = g * 8 + (r & 7); item = r >> 3;
    } else {
        const int r = L - (full >> 3) * per_group;
        h = full + r / nx; item = r - (r / nx) * nx;
    }
}

__device__ __forceinline__ unsigned long long wave_or64(unsigned long long m) {
    unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { lo |= __shfl_xor(lo, off, 64); hi |= __shfl_xor(hi, off, 64); }
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void raster_bin_kernel(const SVert* __restrict__ sv_all, const int32_t* __restrict__ faces,
                                                         const int32_t* __restrict__ perm, const int32_t* __restrict__ fsort, int V,
                                                         int F, int W, int Hh, int T, int Hn, int nchunk,
                                                         uint16_t* __restrict__ tbox, unsigned long long* __restrict__ cmask, int cull) {
    int item, h;
    view_major_ids(gridDim.x, Hn, item, h);
    const int lane = threadIdx.x & 63;
    const int chunk = item * 4 + (threadIdx.x >> 6);
    if (chunk >= nchunk) return;                         // wave-uniform
    const int slot = chunk * BIN_CHUNK + lane;
    uint16_t box = TBOX_NONE;
    unsigned long long m = 0ull;
    if (slot < F) {
        const SVert* sv = sv_all + (size_t)h * V;
        const int i0 = fsort[3 * slot], i1 = fsort[3 * slot + 1], i2 = fsort[3 * slot + 2];   // the slot's corners, read in slot order
        const TriSetup t = tri_setup_v(sv[i0], sv[i1], sv[i2], i0, i1, i2, W, Hh);
        if (t.ok && !(cull && back_facing(t, sv, faces, perm[slot]))) {
            const int tx0 = t.bx0 / T, ty0 = t.by0 / T, tx1 = t.bx1 / T, ty1 = t.by1 / T;
            box = (uint16_t)(tx0 | (ty0 << 4) | (tx1 << 8) | (ty1 << 12));
            const unsigned long long row = ((1ull << (tx1 - tx0 + 1)) - 1ull) << tx0;    // <= 8 tiles per row
            for (int ty = ty0; ty <= ty1; ++ty) m |= row << (ty * 8);
        }
    }
    tbox[((size_t)h * nchunk + chunk) * BIN_CHUNK + lane] = box;
    m = wave_or64(m);
    if (lane == 0) cmask[(size_t)h * nchunk + chunk] = m;
}

__device__ __forceinline__ void strad_pixel_tile(const Strad& q, int f, int px, int py, unsigned long long* tile, int X0, int Y0, int T) {
    float d, b0, b1, b2;
    if (!strad_pixel(q, px, py, d, b0, b1, b2)) return;
    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)f;
    unsigned long long* slot = &tile[(py - Y0) * T + (px - X0)];
    if (key >= *(volatile unsigned long long*)slot) return;
    atomicMin(slot, key);
}

__device__ __forceinline__ void tile_pixel(const TriSetup& t, int f, int px, int py, unsigned long long* tile, int X0, int Y0, int T) {
    float b0, b1, b2, d;
    if (t.small) {
        int w0, w1, w2;
        if (!tri_cover32(t, px, py, w0, w1, w2)) return;
        d = tri_depth32(t, tri_rcp_area(t), w0, w1, w2, b0, b1, b2);   // (three IEEE quotients instead: measured 5 % slower at 1 280 triangles)
    } else {
        long long w0, w1, w2;
        if (!tri_cover(t, px, py, w0, w1, w2)) return;
        d = tri_depth(t, w0, w1, w2, b0, b1, b2);
    }
    if (!(d > 0.f)) return;
    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)f;
    unsigned long long* slot = &tile[(py - Y0) * T + (px - X0)];
    // stored keys only ever decrease, so a plain (possibly stale) read bounds the current value from above: a key that does
    // not beat it cannot change the slot (LDS reads are cheap; the same test on the global buffer is a loss, see tri_pixel)
    if (key >= *(volatile unsigned long long*)slot) return;
    atomicMin(slot, key);
}

struct ExtAcc {
    int cnt, xmin, ymin, xmax, ymax;
    double Xmin, Xmax, Ymin, Ymax;
};
__device__ __forceinline__ double shfl_xor_f64(double v, int off) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __shfl_xor((unsigned)b, off, 64), hi = __shfl_xor((unsigned)(b >> 32), off, 64);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// the set-up a lane holds, handed to the whole wave (lane `src` is wave-uniform: v_readlane, no memory round trip)
__device__ __forceinline__ int rl_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ float rl_f(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ TriSetup bcast_setup(const TriSetup& t, int src) {
    TriSetup r;
    r.x0 = rl_i(t.x0, src); r.y0 = rl_i(t.y0, src); r.x1 = rl_i(t.x1, src); r.y1 = rl_i(t.y1, src); r.x2 = rl_i(t.x2, src); r.y2 = rl_i(t.y2, src);
    r.iz0 = rl_f(t.iz0, src); r.iz1 = rl_f(t.iz1, src); r.iz2 = rl_f(t.iz2, src);
    const unsigned lo = (unsigned)rl_i((int)(unsigned)(unsigned long long)t.area2, src), hi = (unsigned)rl_i((int)(unsigned)((unsigned long long)t.area2 >> 32), src);
    r.area2 = (long long)(((unsigned long long)hi << 32) | lo);
    r.bx0 = rl_i(t.bx0, src); r.by0 = rl_i(t.by0, src); r.bx1 = rl_i(t.bx1, src); r.by1 = rl_i(t.by1, src);
    r.i0 = rl_i(t.i0, src); r.i1 = rl_i(t.i1, src); r.i2 = rl_i(t.i2, src);
    const int flags = rl_i((t.ok ? 1 : 0) | (t.swapped ? 2 : 0) | (t.strad ? 4 : 0) | (t.small ? 8 : 0), src);
    r.ok = flags & 1; r.swapped = flags & 2; r.strad = flags & 4; r.small = flags & 8;
    return r;
}

// what a lane needs of its triangle in a hit chunk: level 1 = coalesced reads in slot order (tile box, corner ids, face id),
// level 2 = the gathered screen-space vertices.  The chunk loop keeps the level-1 reads of chunk i+2 and the gathers of chunk i+1 in
// flight while chunk i is rasterised (a wave otherwise walks its ~30 chunks of an 82 k-triangle mesh one memory round trip at a time).
struct ChunkL1 { unsigned box; int i0, i1, i2, f; };
struct ChunkL2 { SVert a, b, c; };
constexpr int RES_BATCH = 2;             // pixels a thread resolves together (their gathers are issued back to back)

// A lane's own (small) triangle against its <= 32 candidate pixels in the tile, in two passes: coverage of every candidate first (a few
// integer operations each) into a bit mask, then depth + LDS minimum for the covered ones only.  The wave waits for its slowest lane in
// both loops, so the expensive part now runs max-over-lanes(COVERED pixels) times (~1-4 for the ~1.4-pixel triangles of an 82 k mesh)
// instead of max-over-lanes(candidates) (~12): the chunk loop was 2/3 of the kernel on that mesh.
__device__ __forceinline__ void tile_tri_small(const TriSetup& t, int f, int x0, int y0, int x1, int y1, unsigned long long* tile,
                                               int X0, int Y0, int T) {
    const int A0 = t.x2 - t.x1, B0 = t.y2 - t.y1, A1 = t.x0 - t.x2, B1 = t.y0 - t.y2, A2 = t.x1 - t.x0, B2 = t.y1 - t.y0;
    const bool tl0 = topleft(A0, B0), tl1 = topleft(A1, B1), tl2 = topleft(A2, B2);
    const int bw = x1 - x0 + 1;
    unsigned mask = 0u;
    int k = 0;
    for (int py = y0; py <= y1; ++py) {
        const int sy = py * 256 + 128;
        for (int px = x0; px <= x1; ++px, ++k) {
            const int sx = px * 256 + 128;
            const int w0 = __mul24(A0, sy - t.y1) - __mul24(B0, sx - t.x1);
            const int w1 = __mul24(A1, sy - t.y2) - __mul24(B1, sx - t.x2);
            const int w2 = __mul24(A2, sy - t.y0) - __mul24(B2, sx - t.x0);
            const bool in = (w0 | w1 | w2) >= 0 && (w0 != 0 || tl0) && (w1 != 0 || tl1) && (w2 != 0 || tl2);
            mask |= (in ? 1u : 0u) << k;
        }
    }
    if (!mask) return;
    const double rd = tri_rcp_area(t);
    const float rbw = 1.0f / (float)bw;
    while (mask) {
        const int kk = __ffs((int)mask) - 1;
        mask &= mask - 1u;
        const int ry = (int)(((float)kk + 0.5f) * rbw), rx = kk - ry * bw;      // kk / bw for kk < 32, bw <= 32
        const int px = x0 + rx, py = y0 + ry;
        const int sx = px * 256 + 128, sy = py * 256 + 128;
        const int w0 = __mul24(A0, sy - t.y1) - __mul24(B0, sx - t.x1);
        const int w1 = __mul24(A1, sy - t.y2) - __mul24(B1, sx - t.x2);
        const int w2 = __mul24(A2, sy - t.y0) - __mul24(B2, sx - t.x0);
        float b0, b1, b2;
        const float d = tri_depth32(t, rd, w0, w1, w2, b0, b1, b2);
        if (!(d > 0.f)) continue;
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)f;
        unsigned long long* slot = &tile[(py - Y0) * T + (px - X0)];
        if (key < *(volatile unsigned long long*)slot) atomicMin(slot, key);
    }
}

__global__ __launch_bounds__(256) void raster_tile_kernel(const SVert* __restrict__ sv_all, const int32_t* __restrict__ faces,
                                                          const int32_t* __restrict__ perm, const int32_t* __restrict__ fsort, ShadeArgs sh,
                                                          const float* __restrict__ tables, int V, int F, int W, int Hh, int T,
                                                          int ntx, int Hn, const uint16_t* __restrict__ tbox,
                                                          const unsigned long long* __restrict__ cmask, int nchunk,
                                                          uint8_t* __restrict__ rgb, float* __restrict__ depth,
                                                          double* __restrict__ part, double fx, double fy, double cx, double cy,
                                                          int dbg_arg, unsigned long long* dbg_buf) {
#ifdef FP_LAB   // lab build: ablation bits for tools/raster_ablate.py (1 = no rasterisation, 2 = no shading, 4 = no flush, 8 = no mask scan),
    // bits 8.. = the per-lane / whole-wave threshold (0 = the product's), dbg_buf = per-phase shader-clock sums of all workgroups
    const int dbg = dbg_arg & 255;
    const int big_area = (dbg_arg >> 8) ? min(dbg_arg >> 8, 32) : BIG_TILE_AREA;   // (the per-lane coverage mask has 32 bits)
    unsigned long long tstamp = __builtin_readcyclecounter();
#define FP_RASTER_PHASE(k)                                                                                   \
    do {                                                                                                     \
        if (dbg_buf && threadIdx.x == 0) {                                                                   \
            const unsigned long long now__ = __builtin_readcyclecounter();                                   \
            atomicAdd(&dbg_buf[k], now__ - tstamp);                                                          \
            tstamp = now__;                                                                                  \
        }                                                                                                    \
    } while (0)
#else
    constexpr int dbg = 0;
    constexpr int big_area = BIG_TILE_AREA;
    (void)dbg_arg; (void)dbg_buf;
#define FP_RASTER_PHASE(k) do {} while (0)
#endif
    extern __shared__ unsigned long long tile[];   // [T*T] visibility keys | DEC/THR tables (512 floats) | hit list | column / row factors
    float* tab = (float*)(tile + T * T);
    unsigned short* hits = (unsigned short*)(tab + 512);
    double* ax = (double*)(hits + HITS_ROUND);
    double* ay = ax + T;
    __shared__ int nhit_s, total_s;
    __shared__ double red_d[4][4];
    __shared__ int red_i[4][5];
    int tidx, h;
    view_major_ids(gridDim.x, Hn, tidx, h);
    const int ty = tidx / ntx, tx = tidx - ty * ntx;
    const int X0 = tx * T, Y0 = ty * T;
    const int X1 = min(W - 1, X0 + T - 1), Y1 = min(Hh - 1, Y0 + T - 1);
    const SVert* sv = sv_all + (size_t)h * V;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    tab[threadIdx.x] = tables[threadIdx.x];
    tab[threadIdx.x + 256] = tables[threadIdx.x + 256];
    if (threadIdx.x < T) ax[threadIdx.x] = ((double)(X0 + (int)threadIdx.x) - cx) / fx;
    else if ((int)threadIdx.x < 2 * T) ay[threadIdx.x - T] = ((double)(Y0 + (int)threadIdx.x - T) - cy) / fy;
    for (int p = threadIdx.x; p < T * T; p += blockDim.x) tile[p] = ~0ull;
    if (threadIdx.x == 0) { nhit_s = 0; total_s = 0; }
    __syncthreads();
    FP_RASTER_PHASE(0);                                // init
    const int tbit = ty * 8 + tx;
    const unsigned long long* cm = cmask + (size_t)h * nchunk;
    const uint16_t* tb = tbox + (size_t)h * nchunk * BIN_CHUNK;
    for (int base = 0; base < ((dbg & 8) ? 0 : nchunk); base += HITS_ROUND) {
        // ---- which of the next 2048 chunks touch this tile: coalesced mask reads, wave-compacted into the hit list
        unsigned long long mk[HITS_ROUND / 256];
#pragma unroll
        for (int k = 0; k < HITS_ROUND / 256; ++k) {                      // the round's four mask loads in flight together
            const int c = base + k * 256 + (int)threadIdx.x;
            mk[k] = c < nchunk ? cm[c] : 0ull;
        }
#pragma unroll
        for (int k = 0; k < HITS_ROUND / 256; ++k) {
            const int c = base + k * 256 + (int)threadIdx.x;
            const bool hit = (mk[k] >> tbit) & 1ull;
            const unsigned long long bm = __ballot(hit);
            if (bm) {                                                     // wave-uniform
                int wbase = 0;
                if (lane == 0) wbase = atomicAdd(&nhit_s, __popcll(bm));
                wbase = __shfl(wbase, 0, 64);
                if (hit) hits[wbase + __popcll(bm & ((1ull << lane) - 1ull))] = (unsigned short)(c - base);
            }
        }
        __syncthreads();
        FP_RASTER_PHASE(1);                            // mask scan
        const int n = (dbg & 1) ? 0 : nhit_s;
        // ---- every wave takes hit chunks on its own: one triangle per lane, software-pipelined over the wave's chunks
        auto level1 = [&](int i, ChunkL1& c) {
            c.box = TBOX_NONE; c.i0 = c.i1 = c.i2 = 0; c.f = 0;
            if (i < n) {
                const int slot = (base + (int)hits[i]) * BIN_CHUNK + lane;
                c.box = tb[slot];                                        // (TBOX_NONE for slots >= F)
                if (slot < F) { c.i0 = fsort[3 * slot]; c.i1 = fsort[3 * slot + 1]; c.i2 = fsort[3 * slot + 2]; c.f = perm[slot]; }
            }
        };
        auto overlaps = [&](unsigned box) {
            const int bx0 = box & 15, by0 = (box >> 4) & 15, bx1 = (box >> 8) & 15, by1 = (box >> 12) & 15;
            return bx0 <= tx && tx <= bx1 && by0 <= ty && ty <= by1;
        };
        auto level2 = [&](const ChunkL1& c, ChunkL2& g) {   // (lanes whose triangle misses the tile read vertex 0: one cached line, no branch, no zero fill)
            const bool o = overlaps(c.box);
            g.a = sv[o ? c.i0 : 0]; g.b = sv[o ? c.i1 : 0]; g.c = sv[o ? c.i2 : 0];
        };
        ChunkL1 c0, c1, c2;
        ChunkL2 g0, g1;
        level1(wave, c0);
        level1(wave + 4, c1);
        level2(c0, g0);
        for (int i = wave; i < n; i += 4) {
            level1(i + 8, c2);
            level2(c1, g1);
            // set-up on every lane (a wave executes it once whatever the number of live lanes; a branch only added register traffic)
            const int f = c0.f;
            const TriSetup t = tri_setup_v(g0.a, g0.b, g0.c, c0.i0, c0.i1, c0.i2, W, Hh);
            const int x0 = max(t.bx0, X0), y0 = max(t.by0, Y0), x1 = min(t.bx1, X1), y1 = min(t.by1, Y1);
            const bool mine = overlaps(c0.box) && t.ok && x0 <= x1 && y0 <= y1;
            // the wave loop: many candidate pixels, edge functions beyond 32 bits, near-plane straddlers
            const bool big = mine && ((x1 - x0 + 1) * (y1 - y0 + 1) > big_area || !t.small || t.strad);
            if (mine && !big) tile_tri_small(t, f, x0, y0, x1, y1, tile, X0, Y0, T);
            // triangles with many candidate pixels in this tile, and near-plane straddlers: the whole wave strides over them, one triangle
            // at a time, the owner lane's set-up broadcast by v_readlane.  (Round 6 also measured the candidates of a chunk FLATTENED over
            // the lanes — counts prefix-summed, owner found by binary search, set-up fetched by ds_bpermute: 14 permutes per item cost what
            // the idle lanes did; 2.32 / 3.03 / 6.29 ms against 1.70 / 3.13 / 5.26 at 1 280 / 81 920 / 327 680 triangles.  Not shipped.)
            unsigned long long bm = __ballot(big);
            while (bm) {
                const int src = __ffsll((long long)bm) - 1;
                bm &= bm - 1;
                const int fb = rl_i(f, src);
                const TriSetup tbg = bcast_setup(t, src);          // (was a second set-up from memory: two dependent gathers per triangle)
                const int ax0 = max(tbg.bx0, X0), ay0 = max(tbg.by0, Y0), ax1 = min(tbg.bx1, X1), ay1 = min(tbg.by1, Y1);
                const int bw = ax1 - ax0 + 1, np = bw * (ay1 - ay0 + 1);
                if (tbg.strad) {
                    const Strad sq = strad_setup(sv, faces, fb);
                    for (int j = lane; j < np; j += 64) strad_pixel_tile(sq, fb, ax0 + j % bw, ay0 + j / bw, tile, X0, Y0, T);
                    continue;
                }
                for (int j = lane; j < np; j += 64) tile_pixel(tbg, fb, ax0 + j % bw, ay0 + j / bw, tile, X0, Y0, T);
            }
            c0 = c1; c1 = c2; g0 = g1;
        }
#ifdef FP_LAB
        if (dbg_buf && lane == 0 && wave == 0) atomicAdd(&dbg_buf[6], (unsigned long long)(__builtin_readcyclecounter() - tstamp));   // wave 0's own chunk loop
        if (dbg_buf && threadIdx.x == 0) atomicAdd(&dbg_buf[7], (unsigned long long)n);
#endif
        __syncthreads();
        FP_RASTER_PHASE(2);                            // chunk loop (until the slowest wave is done)
        if (threadIdx.x == 0) { total_s += n; nhit_s = 0; }
        __syncthreads();
    }
    const int tw = X1 - X0 + 1, th = Y1 - Y0 + 1;
    const bool touched = total_s > 0;                  // workgroup-uniform
    ExtAcc e;
    e.cnt = 0; e.xmin = 1 << 30; e.ymin = 1 << 30; e.xmax = -1; e.ymax = -1;
    e.Xmin = 1e300; e.Xmax = -1e300; e.Ymin = 1e300; e.Ymax = -1e300;
    if (touched) {
        // ---- resolve: same arithmetic as raster_resolve_kernel; the result replaces the pixel's key in its own LDS slot.
        // RES_BATCH pixels per thread at a time: keys -> corner ids -> vertices + attributes, each level's loads issued for the whole
        // batch before the first is used (a pixel at a time, the three dependent gathers bound the whole kernel: 2 ms per 576 views
        // whatever the triangle count)
        for (int pb = threadIdx.x; pb < tw * th; pb += blockDim.x * RES_BATCH) {
            unsigned long long key[RES_BATCH];
            int lxs[RES_BATCH], lys[RES_BATCH], ids[RES_BATCH][3];
            SVert va[RES_BATCH], vb[RES_BATCH], vc[RES_BATCH];
            TriAttr at[RES_BATCH];
#pragma unroll
            for (int k = 0; k < RES_BATCH; ++k) {
                const int p = pb + k * (int)blockDim.x;
                key[k] = ~0ull; lxs[k] = lys[k] = 0;
                if (p < tw * th) { lys[k] = p / tw; lxs[k] = p - lys[k] * tw; key[k] = tile[lys[k] * T + lxs[k]]; }
            }
#pragma unroll
            for (int k = 0; k < RES_BATCH; ++k) {
                ids[k][0] = ids[k][1] = ids[k][2] = 0;
                if (key[k] != ~0ull) {
                    const int f = (int)(unsigned)(key[k] & 0xffffffffu);
                    ids[k][0] = faces[3 * f]; ids[k][1] = faces[3 * f + 1]; ids[k][2] = faces[3 * f + 2];
                }
            }
#pragma unroll
            for (int k = 0; k < RES_BATCH; ++k) {
                if (key[k] != ~0ull) {
                    va[k] = sv[ids[k][0]]; vb[k] = sv[ids[k][1]]; vc[k] = sv[ids[k][2]];
                    at[k] = load_attr(sh, (int)(unsigned)(key[k] & 0xffffffffu), ids[k][0], ids[k][1], ids[k][2]);
                } else {
                    va[k] = vb[k] = vc[k] = SVert{0, 0, 0.f, 0.f};
                    at[k] = TriAttr{};
                }
            }
#pragma unroll
            for (int k = 0; k < RES_BATCH; ++k) {
                const int p = pb + k * (int)blockDim.x;
                if (p >= tw * th) continue;
                const int lx = lxs[k], ly = lys[k], px = X0 + lx, py = Y0 + ly;
                float d = 0.f;
                uint8_t out[3] = {0, 0, 0};
                if (key[k] != ~0ull && !(dbg & 2)) {
                    const TriSetup t = tri_setup_v(va[k], vb[k], vc[k], ids[k][0], ids[k][1], ids[k][2], W, Hh);
                    resolve_pixel_v(sv, faces, sh, tab, key[k], t, at[k], px, py, d, out);
                }
                tile[ly * T + lx] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)out[0] | ((unsigned)out[1] << 8) | ((unsigned)out[2] << 16);
                if (d != 0.f) {   // fp_depth_extents: every non-zero depth joins the cloud, every positive one the mask
                    const double X = ax[lx] * (double)d, Y = ay[ly] * (double)d;
                    e.Xmin = fmin(e.Xmin, X); e.Xmax = fmax(e.Xmax, X); e.Ymin = fmin(e.Ymin, Y); e.Ymax = fmax(e.Ymax, Y);
                    if (d > 0.f) { ++e.cnt; e.xmin = min(e.xmin, px); e.xmax = max(e.xmax, px); e.ymin = min(e.ymin, py); e.ymax = max(e.ymax, py); }
                }
            }
        }
    } else {
        for (int p = threadIdx.x; p < T * T; p += blockDim.x) tile[p] = 0ull;
    }
    __syncthreads();
    FP_RASTER_PHASE(3);                                // resolve
    // ---- flush: depth rows as floats; rgb rows as whole dwords (byte stores only for a row's unaligned ends)
    if (dbg & 4) return;
    if (depth) {
        for (int p = threadIdx.x; p < tw * th; p += blockDim.x) {
            const int ly = p / tw, lx = p - ly * tw;
            depth[((size_t)h * Hh + Y0 + ly) * W + X0 + lx] = __uint_as_float((unsigned)(tile[ly * T + lx] >> 32));
        }
    }
    if (((W * 3) & 3) == 0 && (tw & 3) == 0 && (T & 3) == 0) {
        // rows start on a dword (W * 3 and X0 * 3 are multiples of 4) and hold whole groups of four pixels = three dwords: the
        // packed results' low words are spliced with shifts, one 12-byte store per lane, adjacent lanes adjacent
        const int nq = tw >> 2;
        for (int q = threadIdx.x; q < th * nq; q += blockDim.x) {
            const int ly = q / nq, qx = q - ly * nq;
            const unsigned long long* src = &tile[ly * T + 4 * qx];
            const unsigned p0 = (unsigned)src[0], p1 = (unsigned)src[1], p2 = (unsigned)src[2], p3 = (unsigned)src[3];
            uint3 o;
            o.x = (p0 & 0xffffffu) | (p1 << 24);
            o.y = ((p1 >> 8) & 0xffffu) | (p2 << 16);
            o.z = ((p2 >> 16) & 0xffu) | (p3 << 8);
            *(uint3*)(rgb + (((size_t)h * Hh + Y0 + ly) * W + X0 + 4 * qx) * 3) = o;
        }
    } else {
        const int ndw = (tw * 3 + 3) / 4 + 1;                          // dwords that can hold a row's bytes at any alignment
        for (int q = threadIdx.x; q < th * ndw; q += blockDim.x) {
            const int ly = q / ndw, j = q - ly * ndw;
            const size_t start = (((size_t)h * Hh + Y0 + ly) * W + X0) * 3;   // first byte of the tile row in the rgb buffer
            const size_t a = (start & ~(size_t)3) + (size_t)j * 4;            // this thread's aligned dword
            const long long o0 = (long long)a - (long long)start;              // row-relative offset of its first byte
            if (o0 >= (long long)tw * 3) continue;
            unsigned v = 0;
            bool in[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long long o = o0 + k;
                in[k] = o >= 0 && o < (long long)tw * 3;
                if (in[k]) {
                    const int pix = (int)o / 3, ch = (int)o - pix * 3;
                    v |= (unsigned)((tile[ly * T + pix] >> (8 * ch)) & 255ull) << (8 * k);
                }
            }
            if (in[0] && in[3]) *(uint32_t*)(rgb + a) = v;
            else {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (in[k]) rgb[a + k] = (uint8_t)(v >> (8 * k));
            }
        }
    }
    FP_RASTER_PHASE(4);                                // flush
    if (!part) return;
    // ---- the tile's extents record
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        e.cnt += __shfl_xor(e.cnt, off, 64);
        e.xmin = min(e.xmin, __shfl_xor(e.xmin, off, 64)); e.ymin = min(e.ymin, __shfl_xor(e.ymin, off, 64));
        e.xmax = max(e.xmax, __shfl_xor(e.xmax, off, 64)); e.ymax = max(e.ymax, __shfl_xor(e.ymax, off, 64));
        e.Xmin = fmin(e.Xmin, shfl_xor_f64(e.Xmin, off)); e.Xmax = fmax(e.Xmax, shfl_xor_f64(e.Xmax, off));
        e.Ymin = fmin(e.Ymin, shfl_xor_f64(e.Ymin, off)); e.Ymax = fmax(e.Ymax, shfl_xor_f64(e.Ymax, off));
    }
    if (lane == 0) {
        red_i[wave][0] = e.cnt; red_i[wave][1] = e.xmin; red_i[wave][2] = e.ymin; red_i[wave][3] = e.xmax; red_i[wave][4] = e.ymax;
        red_d[wave][0] = e.Xmin; red_d[wave][1] = e.Xmax; red_d[wave][2] = e.Ymin; red_d[wave][3] = e.Ymax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            red_i[0][0] += red_i[w][0];
            red_i[0][1] = min(red_i[0][1], red_i[w][1]); red_i[0][2] = min(red_i[0][2], red_i[w][2]);
            red_i[0][3] = max(red_i[0][3], red_i[w][3]); red_i[0][4] = max(red_i[0][4], red_i[w][4]);
            red_d[0][0] = fmin(red_d[0][0], red_d[w][0]); red_d[0][1] = fmax(red_d[0][1], red_d[w][1]);
            red_d[0][2] = fmin(red_d[0][2], red_d[w][2]); red_d[0][3] = fmax(red_d[0][3], red_d[w][3]);
        }
        double* o = part + ((size_t)h * gridDim.x + tidx) * PART_N;
        for (int k = 0; k < 5; ++k) o[k] = (double)red_i[0][k];
        for (int k = 0; k < 4; ++k) o[5 + k] = red_d[0][k];
    }
}

// folds the per-tile records of a view into fp_depth_extents' row (and the int32 box CropResizePad takes): same rule for views with
// fewer than 100 mask pixels (renderer.py:116-117, template.py:75-77)
__global__ __launch_bounds__(64) void raster_extents_kernel(const double* __restrict__ part, int Hn, int ntile, int W, int Hh,
                                                            double* __restrict__ ext, int32_t* __restrict__ boxes) {
    const int v = blockIdx.x, lane = threadIdx.x;        // one wave per view, one tile record per lane (<= 64 tiles), butterfly reduction
    int c = 0, bx0 = 1 << 30, by0 = 1 << 30, bx1 = -1, by1 = -1;
    double Xmin = 1e300, Xmax = -1e300, Ymin = 1e300, Ymax = -1e300;
    for (int t = lane; t < ntile; t += 64) {
        const double* p = part + ((size_t)v * ntile + t) * PART_N;
        c += (int)p[0];
        bx0 = min(bx0, (int)p[1]); by0 = min(by0, (int)p[2]); bx1 = max(bx1, (int)p[3]); by1 = max(by1, (int)p[4]);
        Xmin = fmin(Xmin, p[5]); Xmax = fmax(Xmax, p[6]); Ymin = fmin(Ymin, p[7]); Ymax = fmax(Ymax, p[8]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        c += __shfl_xor(c, off, 64);
        bx0 = min(bx0, __shfl_xor(bx0, off, 64)); by0 = min(by0, __shfl_xor(by0, off, 64));
        bx1 = max(bx1, __shfl_xor(bx1, off, 64)); by1 = max(by1, __shfl_xor(by1, off, 64));
        Xmin = fmin(Xmin, shfl_xor_f64(Xmin, off)); Xmax = fmax(Xmax, shfl_xor_f64(Xmax, off));
        Ymin = fmin(Ymin, shfl_xor_f64(Ymin, off)); Ymax = fmax(Ymax, shfl_xor_f64(Ymax, off));
    }
    if (lane != 0) return;
    if (c < 100) {
        const int lo = 105, hx = min(315, W) - 1, hy = min(315, Hh) - 1;
        if (c == 0) { bx0 = lo; by0 = lo; bx1 = hx; by1 = hy; }
        else { bx0 = min(bx0, lo); by0 = min(by0, lo); bx1 = max(bx1, hx); by1 = max(by1, hy); }
    }
    if (ext) {
        double* o = ext + (size_t)v * 8;
        o[0] = (double)bx0; o[1] = (double)by0; o[2] = (double)bx1; o[3] = (double)by1;
        o[4] = Xmax > -1e299 ? Xmax - Xmin : 0.0;
        o[5] = Ymax > -1e299 ? Ymax - Ymin : 0.0;
        o[6] = (double)c; o[7] = 0.0;
    }
    if (boxes) { boxes[4 * v] = bx0; boxes[4 * v + 1] = by0; boxes[4 * v + 2] = bx1; boxes[4 * v + 3] = by1; }
}

// int32 boxes of an extents table (global-buffer path of fp_rasterize_extents)
__global__ void ext_boxes_kernel(const double* __restrict__ ext, int Hn, int32_t* __restrict__ boxes) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= Hn) return;
    for (int k = 0; k < 4; ++k) boxes[4 * v + k] = (int)ext[(size_t)v * 8 + k];
}

// vertex stage export (fp_project_vertices)
__global__ void raster_export_kernel(const SVert* __restrict__ sv, const int32_t* __restrict__ vmap, int V, size_t n,
                                     int32_t* __restrict__ xy, float* __restrict__ zc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (view, caller's vertex id)
    if (i >= n) return;
    const size_t h = i / (size_t)V;
    const SVert v = sv[h * (size_t)V + (size_t)vmap[i - h * (size_t)V]];
    const bool front = v.zc > ZNEAR;     // behind the near plane the fields hold homogeneous coordinates (straddler path), not pixels
    xy[2 * i] = front ? v.xi : 0; xy[2 * i + 1] = front ? v.yi : 0;
    zc[i] = v.zc;
}

}  // namespace

static int mesh_tables(fp_mesh* m) {
    // DEC[i] = sRGB -> linear of i/255 ; THR[k] = ((k - 0.5)/255)^2.2 (THR[0] = 0): double precision, rounded to float once
    float tab[512];
    for (int i = 0; i < 256; ++i) {
        const double sv = (double)i / 255.0;
        tab[i] = (float)(sv <= 0.04045 ? sv / 12.92 : pow((sv + 0.055) / 1.055, 2.4));
        tab[256 + i] = i == 0 ? 0.f : (float)pow(((double)i - 0.5) / 255.0, 2.2);
    }
    FP_HIP(hipMalloc((void**)&m->tables, sizeof(tab)));
    FP_HIP(hipMemcpy(m->tables, tab, sizeof(tab), hipMemcpyHostToDevice));
    return FP_OK;
}

extern "C" int fp_mesh_destroy(fp_mesh* m);
namespace {
// frees a half-built mesh when an upload step fails (FP_HIP / FP_REQUIRE return early); release() on success
struct MeshGuard {
    fp_mesh* m;
    ~MeshGuard() { if (m) (void)fp_mesh_destroy(m); }
    fp_mesh* release() { fp_mesh* r = m; m = nullptr; return r; }
};
}  // namespace

static int mesh_geometry(fp_ctx* ctx, const float* h_verts, int V, const int32_t* h_faces, int F, fp_mesh** out) {
    FP_REQUIRE(ctx && h_verts && h_faces && out && V > 0 && F > 0, "mesh_upload: bad argument");
    for (int i = 0; i < 3 * F; ++i) FP_REQUIRE(h_faces[i] >= 0 && h_faces[i] < V, "mesh_upload: face index out of range");
    fp_mesh* m = new fp_mesh();
    MeshGuard guard{m};
    m->ctx = ctx; m->V = V; m->F = F;
    FP_HIP(hipMalloc((void**)&m->verts, (size_t)V * 12));
    FP_HIP(hipMalloc((void**)&m->faces, (size_t)F * 12));
    {   // Morton order of the triangle centroids (10 bits per axis over the vertex bounding box), ties by face id
        float lo[3] = {h_verts[0], h_verts[1], h_verts[2]}, hi[3] = {h_verts[0], h_verts[1], h_verts[2]};
        for (int i = 0; i < V; ++i)
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], h_verts[3 * i + a]); hi[a] = std::max(hi[a], h_verts[3 * i + a]); }
        auto spread = [](uint32_t x) {   // 10 bits -> every third bit
            x &= 0x3ffu; x = (x | (x << 16)) & 0x030000ffu; x = (x | (x << 8)) & 0x0300f00fu; x = (x | (x << 4)) & 0x030c30c3u; x = (x | (x << 2)) & 0x09249249u;
            return x;
        };
        std::vector<std::pair<uint32_t, int32_t>> key((size_t)F);
        for (int f = 0; f < F; ++f) {
            uint32_t code = 0;
            for (int a = 0; a < 3; ++a) {
                const float c = (h_verts[3 * h_faces[3 * f] + a] + h_verts[3 * h_faces[3 * f + 1] + a] + h_verts[3 * h_faces[3 * f + 2] + a]) * (1.0f / 3.0f);
                const float span = hi[a] - lo[a];
                float u = span > 0.f ? (c - lo[a]) / span : 0.f;
                if (!(u >= 0.f)) u = 0.f;                // (NaN vertices land in cell 0; they are dropped by the raster set-up anyway)
                const uint32_t q = (uint32_t)std::min(1023.0f, u * 1024.0f);
                code |= spread(q) << a;
            }
            key[f] = {code, f};
        }
        std::sort(key.begin(), key.end());
        std::vector<int32_t> perm((size_t)F);
        for (int f = 0; f < F; ++f) perm[f] = key[f].second;
        // device vertex ids: order of first use along the sorted triangles (unreferenced vertices last).  Purely internal — outputs carry
        // no vertex id, fp_project_vertices maps back — but a chunk's vertices are then neighbours in the per-view vertex array: the
        // gathers of the bin / tile / resolve stages fetched one 128-byte line per 16-byte vertex before (3 x 64 lines per chunk).
        std::vector<int32_t> vmap((size_t)V, -1);
        int next = 0;
        for (int k = 0; k < F; ++k)
            for (int a = 0; a < 3; ++a) { int32_t& id = vmap[h_faces[3 * (size_t)perm[k] + a]]; if (id < 0) id = next++; }
        for (int i = 0; i < V; ++i) if (vmap[i] < 0) vmap[i] = next++;
        std::vector<float> verts((size_t)V * 3);
        for (int i = 0; i < V; ++i)
            for (int a = 0; a < 3; ++a) verts[3 * (size_t)vmap[i] + a] = h_verts[3 * (size_t)i + a];
        std::vector<int32_t> faces((size_t)F * 3), fsort((size_t)F * 3);
        for (size_t i = 0; i < (size_t)F * 3; ++i) faces[i] = vmap[h_faces[i]];
        for (int k = 0; k < F; ++k)
            for (int a = 0; a < 3; ++a) fsort[3 * (size_t)k + a] = faces[3 * (size_t)perm[k] + a];
        FP_HIP(hipMemcpy(m->verts, verts.data(), (size_t)V * 12, hipMemcpyHostToDevice));
        FP_HIP(hipMemcpy(m->faces, faces.data(), (size_t)F * 12, hipMemcpyHostToDevice));
        FP_HIP(hipMalloc((void**)&m->perm, (size_t)F * 4));
        FP_HIP(hipMemcpy(m->perm, perm.data(), (size_t)F * 4, hipMemcpyHostToDevice));
        FP_HIP(hipMalloc((void**)&m->fsort, (size_t)F * 12));
        FP_HIP(hipMemcpy(m->fsort, fsort.data(), (size_t)F * 12, hipMemcpyHostToDevice));
        FP_HIP(hipMalloc((void**)&m->vmap, (size_t)V * 4));
        FP_HIP(hipMemcpy(m->vmap, vmap.data(), (size_t)V * 4, hipMemcpyHostToDevice));
        m->h_vmap = std::move(vmap);
    }
    int rc = mesh_tables(m);
    if (rc) return rc;
    *out = guard.release();
    return FP_OK;
}

extern "C" int fp_mesh_upload(fp_ctx* ctx, const float* h_verts, int V, const int32_t* h_faces, int F,
                              const uint8_t* h_colors, fp_mesh** out) {
    fp_mesh* m = nullptr;
    int rc = mesh_geometry(ctx, h_verts, V, h_faces, F, &m);
    if (rc) return rc;
    MeshGuard guard{m};
    if (h_colors) {
        std::vector<uint8_t> rgba((size_t)V * 4, 255);
        for (int i = 0; i < V; ++i) {
            const size_t d = (size_t)m->h_vmap[i];                   // vertex attributes live at the device id
            rgba[4 * d] = h_colors[3 * i]; rgba[4 * d + 1] = h_colors[3 * i + 1]; rgba[4 * d + 2] = h_colors[3 * i + 2];
        }
        FP_HIP(hipMalloc((void**)&m->colors, (size_t)V * 4));
        FP_HIP(hipMemcpy(m->colors, rgba.data(), (size_t)V * 4, hipMemcpyHostToDevice));
    }
    *out = guard.release();
    return FP_OK;
}

extern "C" int fp_mesh_upload_textured(fp_ctx* ctx, const float* h_verts, int V, const int32_t* h_faces, int F,
                                       const float* h_uv, const uint8_t* h_texture, int th, int tw, const float* h_kd3,
                                       fp_mesh** out) {
    FP_REQUIRE(h_uv && h_texture && th > 0 && tw > 0 && th <= 16384 && tw <= 16384, "mesh_upload_textured: bad texture argument");
    fp_mesh* m = nullptr;
    int rc = mesh_geometry(ctx, h_verts, V, h_faces, F, &m);
    if (rc) return rc;
    MeshGuard guard{m};
    // level 0 + the box-filtered chain down to 1 x 1 (contract in the header), rgba, back to back
    int lw[16], lh[16], n = 0;
    size_t total = 0;
    for (int w = tw, h = th;; w = w > 1 ? w >> 1 : 1, h = h > 1 ? h >> 1 : 1) {
        lw[n] = w; lh[n] = h; m->lev_off[n] = (uint32_t)total; total += (size_t)w * h; ++n;
        if ((w == 1 && h == 1) || n == 16) break;
    }
    m->nlev = n;
    std::vector<uint8_t> rgba(total * 4, 255);
    for (size_t i = 0; i < (size_t)th * tw; ++i) { rgba[4 * i] = h_texture[3 * i]; rgba[4 * i + 1] = h_texture[3 * i + 1]; rgba[4 * i + 2] = h_texture[3 * i + 2]; }
    for (int k = 1; k < n; ++k) {
        const uint8_t* src = rgba.data() + (size_t)m->lev_off[k - 1] * 4;
        uint8_t* dst = rgba.data() + (size_t)m->lev_off[k] * 4;
        const int sw = lw[k - 1], sh = lh[k - 1];
        for (int y = 0; y < lh[k]; ++y)
            for (int x = 0; x < lw[k]; ++x) {
                const int x0 = std::min(2 * x, sw - 1), x1 = std::min(2 * x + 1, sw - 1), y0 = std::min(2 * y, sh - 1), y1 = std::min(2 * y + 1, sh - 1);
                for (int c = 0; c < 3; ++c)
                    dst[((size_t)y * lw[k] + x) * 4 + c] = (uint8_t)((src[((size_t)y0 * sw + x0) * 4 + c] + src[((size_t)y0 * sw + x1) * 4 + c] +
                                                                     src[((size_t)y1 * sw + x0) * 4 + c] + src[((size_t)y1 * sw + x1) * 4 + c] + 2) >> 2);
            }
    }
    FP_HIP(hipMalloc((void**)&m->tex, rgba.size()));
    FP_HIP(hipMemcpy(m->tex, rgba.data(), rgba.size(), hipMemcpyHostToDevice));
    FP_HIP(hipMalloc((void**)&m->uv, (size_t)F * 24));
    FP_HIP(hipMemcpy(m->uv, h_uv, (size_t)F * 24, hipMemcpyHostToDevice));
    m->th = th; m->tw = tw;
    if (h_kd3) { m->kd[0] = h_kd3[0]; m->kd[1] = h_kd3[1]; m->kd[2] = h_kd3[2]; }
    *out = guard.release();
    return FP_OK;
}
extern "C" int fp_mesh_destroy(fp_mesh* m) {
    if (!m) return FP_OK;
    if (m->verts) (void)hipFree(m->verts);
    if (m->faces) (void)hipFree(m->faces);
    if (m->perm) (void)hipFree(m->perm);
    if (m->fsort) (void)hipFree(m->fsort);
    if (m->vmap) (void)hipFree(m->vmap);
    if (m->colors) (void)hipFree(m->colors);
    if (m->uv) (void)hipFree(m->uv);
    if (m->tex) (void)hipFree(m->tex);
    if (m->tables) (void)hipFree(m->tables);
    delete m;
    return FP_OK;
}

static ShadeArgs shade_args(const fp_mesh* m) {
    ShadeArgs a;
    a.colors = m->colors; a.uv = m->uv; a.tex = m->tex; a.th = m->th; a.tw = m->tw;
    a.kd0 = m->kd[0]; a.kd1 = m->kd[1]; a.kd2 = m->kd[2];
    a.ambient = m->ambient; a.shade = m->shade;
    a.nlev = m->nlev; a.filter = m->filter;
    for (int k = 0; k < 16; ++k) a.lev_off[k] = m->lev_off[k];
    return a;
}

extern "C" int fp_project_vertices(fp_ctx* ctx, const fp_mesh* mesh, const float* d_poses, int Hn, float scale, float fx,
                                   float fy, float cx, float cy, int32_t* d_xy, float* d_zc, void* stream) {
    FP_REQUIRE(ctx && mesh && d_poses && d_xy && d_zc, "project_vertices: null argument");
    if (Hn == 0) return FP_OK;
    hipStream_t s = (hipStream_t)stream;
    const int V = mesh->V;
    SVert* sv;
    int rc;
    if ((rc = ctx->get("raster.sv", (size_t)Hn * V * sizeof(SVert), (void**)&sv))) return rc;
    hipLaunchKernelGGL(raster_vertex_kernel, dim3(cdiv(V, 256), Hn), dim3(256), 0, s, mesh->verts, V, d_poses, Hn, scale,
                       fx, fy, cx, cy, sv);
    FP_LAUNCH_CHECK();
    const size_t n = (size_t)Hn * V;
    hipLaunchKernelGGL(raster_export_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sv, mesh->vmap, V, n, d_xy, d_zc);
    FP_LAUNCH_CHECK();
    return FP_OK;
}

// d_depth, d_ext, d_boxes: each may be null (at least rgb is always written)
static int rasterize_impl(fp_ctx* ctx, const fp_mesh* mesh, const float* d_poses, int Hn, float scale, float fx, float fy, float cx,
                          float cy, int W, int Hh, uint8_t* d_rgb, float* d_depth, double* d_ext, int32_t* d_boxes, hipStream_t s) {
    const int V = mesh->V, F = mesh->F;
    SVert* sv;
    int rc;
    if ((rc = ctx->get("raster.sv", (size_t)Hn * V * sizeof(SVert), (void**)&sv))) return rc;
    hipLaunchKernelGGL(raster_vertex_kernel, dim3(cdiv(V, 256), Hn), dim3(256), 0, s, mesh->verts, V, d_poses, Hn, scale,
                       fx, fy, cx, cy, sv);
    FP_LAUNCH_CHECK();
    const bool want_ext = d_ext || d_boxes;

    // tile edge: 64 px up to 512-px images, else the smallest multiple of 8 that covers the image with 8 x 8 tiles
    const int side = W > Hh ? W : Hh;
    const int T = side <= 512 ? 64 : (cdiv(side, 8) + 7) / 8 * 8;
    // fp_ctx_set_option(ctx, "raster_tiled", v): -1 / unset = tiled for images of up to 8 x 8 tiles of <= 88 px and meshes of up to
    // 131 072 triangles, 0 = global visibility buffer, 1 = tiled.  Measured, 576 views @420^2 with boxes + extents (ms; tiled | global,
    // profiles/r06_raster_perf.log): 1 280 triangles 1.64 | 2.57, 20 480: 1.65 | 2.70, 81 920: 2.28 | 2.40, 327 680: 5.04 | 2.81 —
    // both are bound by vector instructions, the tiled path per triangle (set-up in the bin AND the tile kernel, ~1.6 tiles per chunk of
    // 64), the global path per fragment (L2 atomics).  Rounds 1-5 stopped at 32 768 triangles.
    const int mode = ctx->opt_raster_tiled;
    const bool tiled = T <= 88 && (mode == 1 || (mode != 0 && F <= 131072));
    if (tiled) {
        const int ntx = cdiv(W, T), nty = cdiv(Hh, T), ntile = ntx * nty, nchunk = cdiv(F, BIN_CHUNK);
        uint16_t* tbox;
        unsigned long long* cmask;
        double* part = nullptr;
        if ((rc = ctx->get("raster.tbox", (size_t)Hn * nchunk * BIN_CHUNK * 2, (void**)&tbox))) return rc;
        if ((rc = ctx->get("raster.cmask", (size_t)Hn * nchunk * 8, (void**)&cmask))) return rc;
        if (want_ext && (rc = ctx->get("raster.part", (size_t)Hn * ntile * PART_N * 8, (void**)&part))) return rc;
        hipLaunchKernelGGL(raster_bin_kernel, dim3(cdiv(nchunk, 4), Hn), dim3(256), 0, s, sv, mesh->faces, mesh->perm, mesh->fsort, V, F, W, Hh, T, Hn,
                           nchunk, tbox, cmask, mesh->cull);
        FP_LAUNCH_CHECK();
        unsigned long long* dbg_buf = nullptr;
#ifdef FP_LAB
        const int raster_dbg = fp_opt_get(FP_OPT_RASTER_DBG, 0);
        if (raster_dbg & (1 << 30)) {                      // phase clocks: summed over all workgroups into "raster.dbg" (fp_lab_read_buffer)
            if ((rc = ctx->get("raster.dbg", 64, (void**)&dbg_buf))) return rc;
            FP_HIP(hipMemsetAsync(dbg_buf, 0, 64, s));
        }
#else
        const int raster_dbg = 0;
#endif
        const size_t lds = (size_t)T * T * 8 + 2048 + HITS_ROUND * 2 + (size_t)2 * T * 8;   // keys + DEC/THR tables + hit list + column / row factors
        FP_DYN_LDS_ONCE(raster_tile_kernel, 88 * 88 * 8 + 2048 + HITS_ROUND * 2 + 2 * 88 * 8);
        hipLaunchKernelGGL(raster_tile_kernel, dim3(ntile, Hn), dim3(256), lds, s, sv, mesh->faces, mesh->perm, mesh->fsort, shade_args(mesh), mesh->tables,
                           V, F, W, Hh, T, ntx, Hn, tbox, cmask, nchunk, d_rgb, d_depth, part, (double)fx, (double)fy, (double)cx, (double)cy, raster_dbg & ~(1 << 30), dbg_buf);
        FP_LAUNCH_CHECK();
        if (want_ext) {
            hipLaunchKernelGGL(raster_extents_kernel, dim3(Hn), dim3(64), 0, s, part, Hn, ntile, W, Hh, d_ext, d_boxes);
            FP_LAUNCH_CHECK();
        }
        return FP_OK;
    }

    // global visibility-buffer path (large images; A/B reference)
    unsigned long long* zb;
    int* queue;
    const int qcap = 1 << 20;
    if (!d_depth && (rc = ctx->get("raster.depth", (size_t)Hn * W * Hh * 4, (void**)&d_depth))) return rc;   // the extents are taken from it
    if ((rc = ctx->get("raster.zb", (size_t)Hn * W * Hh * 8, (void**)&zb))) return rc;
    if ((rc = ctx->get("raster.queue", (size_t)qcap * 8 + 64, (void**)&queue))) return rc;
    int* qcount = queue + 2 * qcap;
    FP_HIP(hipMemsetAsync(zb, 0xff, (size_t)Hn * W * Hh * 8, s));
    FP_HIP(hipMemsetAsync(qcount, 0, 4, s));
    hipLaunchKernelGGL(raster_tri_kernel, dim3(cdiv(F, 256), Hn), dim3(256), 0, s, sv, mesh->faces, mesh->perm, V, F, W, Hh, zb,
                       queue, qcount, qcap, mesh->cull);
    FP_LAUNCH_CHECK();
    hipLaunchKernelGGL(raster_big_kernel, dim3(1024), dim3(256), 0, s, sv, mesh->faces, V, W, Hh, zb, queue, qcount, qcap);
    FP_LAUNCH_CHECK();
    hipLaunchKernelGGL(raster_resolve_kernel, dim3(cdiv(W * Hh, 256), Hn), dim3(256), 0, s, sv, mesh->faces, shade_args(mesh),
                       mesh->tables, V, W, Hh, zb, d_rgb, d_depth);
    FP_LAUNCH_CHECK();
    if (want_ext) {
        double* ext = d_ext;
        if (!ext && (rc = ctx->get("raster.ext", (size_t)Hn * 64, (void**)&ext))) return rc;
        if ((rc = fp_depth_extents(ctx, d_depth, Hn, Hh, W, fx, fy, cx, cy, ext, s))) return rc;
        if (d_boxes) {
            hipLaunchKernelGGL(ext_boxes_kernel, dim3(cdiv(Hn, 64)), dim3(64), 0, s, ext, Hn, d_boxes);
            FP_LAUNCH_CHECK();
        }
    }
    return FP_OK;
}

extern "C" int fp_rasterize(fp_ctx* ctx, const fp_mesh* mesh, const float* d_poses, int Hn, float scale, float fx,
                            float fy, float cx, float cy, int W, int Hh, uint8_t* d_rgb, float* d_depth, void* stream) {
    FP_REQUIRE(ctx && mesh && d_poses && d_rgb && d_depth, "rasterize: null argument");
    FP_REQUIRE(W > 0 && Hh > 0 && W <= 8192 && Hh <= 8192, "rasterize: bad image size");
    if (Hn == 0) return FP_OK;
    return rasterize_impl(ctx, mesh, d_poses, Hn, scale, fx, fy, cx, cy, W, Hh, d_rgb, d_depth, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int fp_rasterize_extents(fp_ctx* ctx, const fp_mesh* mesh, const float* d_poses, int Hn, float scale, float fx,
                                    float fy, float cx, float cy, int W, int Hh, uint8_t* d_rgb, float* d_depth, double* d_ext,
                                    int32_t* d_boxes, void* stream) {
    FP_REQUIRE(ctx && mesh && d_poses && d_rgb && (d_ext || d_boxes), "rasterize_extents: null argument");
    FP_REQUIRE(W > 0 && Hh > 0 && W <= 8192 && Hh <= 8192, "rasterize_extents: bad image size");
    if (Hn == 0) return FP_OK;
    return rasterize_impl(ctx, mesh, d_poses, Hn, scale, fx, fy, cx, cy, W, Hh, d_rgb, d_depth, d_ext, d_boxes, (hipStream_t)stream);
}

extern "C" int fp_mesh_set_ambient(fp_mesh* mesh, float ambient) {
    FP_REQUIRE(mesh && ambient >= 0.f, "mesh_set_ambient: bad argument");
    mesh->ambient = ambient;
    return FP_OK;
}
extern "C" int fp_mesh_set_cull(fp_mesh* mesh, int mode) {
    FP_REQUIRE(mesh && (mode == 0 || mode == 1), "mesh_set_cull: mode must be 0 (both sides) or 1 (back faces culled)");
    mesh->cull = mode;
    return FP_OK;
}
extern "C" int fp_mesh_set_filter(fp_mesh* mesh, int mode) {
    FP_REQUIRE(mesh && (mode == 0 || mode == 1), "mesh_set_filter: mode must be 0 (bilinear level 0) or 1 (trilinear mip-maps)");
    mesh->filter = mode;
    return FP_OK;
}
extern "C" int fp_mesh_set_shading(fp_mesh* mesh, int mode) {
    FP_REQUIRE(mesh && (mode == 0 || mode == 1), "mesh_set_shading: mode must be 0 (linear) or 1 (gamma)");
    mesh->shade = mode;
    return FP_OK;
}
